"""The parity tests of test_gpu_parity.py / test_gpu_edges.py once more, with every engine created under PTX_FLAG_NARROW_IDS: resident
batches then carry the narrow mirror of the id / side columns (32-bit ids, both sides in one byte) and ptx_merge launches the kernel builds
that read it (ptx_merge_kernel_n, ..._rest_n, ..._many_n, ..._diag_n).  Same inputs, same oracle, same bit-exact bar: the mirror is a
second encoding of the same rows, so nothing may differ — documents, digests, error statuses and the row a failing log is blamed on,
elem_rank and the resolved references the patch-stream replay / cursors / change() read.

The tests are the other modules' own functions, collected here under this module's `eng` fixture; PTX_NARROW=1 in the environment (the
library's tuning override, read by ptx_create) also turns the flag on for the engines those tests create themselves."""
import os

import pytest

import test_gpu_edges as E
import test_gpu_parity as P
from peritext_amd import abi

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _narrow_env():
    old = os.environ.get("PTX_NARROW")
    os.environ["PTX_NARROW"] = "1"
    yield
    if old is None:
        del os.environ["PTX_NARROW"]
    else:
        os.environ["PTX_NARROW"] = old


@pytest.fixture(scope="module")
def eng(_narrow_env):
    import torch

    assert torch.cuda.is_available(), "GPU tests need a real MI355X"
    from peritext_amd.engine import Engine

    e = Engine(0, flags=abi.FLAG_NARROW_IDS)
    yield e
    e.close()


golden = E.golden  # the edge-case fixture of test_gpu_edges.py


def test_the_narrow_kernels_are_the_ones_launched(eng):
    assert eng.flags() & abi.FLAG_NARROW_IDS
    assert eng.kernel_name() == "ptx_merge_kernel_n"
    from peritext_amd.engine import Engine

    with Engine(0) as e2:  # the environment override reaches engines created without the flag
        assert e2.flags() & abi.FLAG_NARROW_IDS


# everything in the two modules that runs ptx_merge (and what reads its outputs: replay, cursors, generate + merge, append, split launch)
_SKIP = {
    "test_native_library_is_loaded",  # asserts the wide kernel's name
    "test_digest_allgather_in_the_c_abi_single_rank",
    "test_root_maps_on_the_device",  # ptx_root_map reads the wire columns only
}
for _mod in (P, E):
    for _name in dir(_mod):
        if _name.startswith("test_") and _name not in _SKIP:
            globals()[_name] = getattr(_mod, _name)
del _mod, _name
