"""The emulation suites once more over the NARROW mirror of the id / side columns (PTX_FLAG_NARROW_IDS): the emulation driver then packs
every batch the way the library's ptx_narrow_pack_kernel does (32-bit ids counter << 12 | actorRank, both sides in one byte) and runs the
kNarrow build of the kernel body.  Same oracle, same bit-exact bar — documents, digests, statuses and blamed rows, elem_rank, the resolved
references the replay / cursor / change() emulations read.  The tests are the other modules' own functions, collected here."""
import ctypes
import os

import pytest

import helpers as H
import test_emu_change as C_
import test_emu_parity as P
import test_emu_patches as R

pytestmark = pytest.mark.skipif(not os.path.exists(H.EMU_LIB), reason="tests/emu/libperitext_emu.so not built (run __graft_entry__.build())")


@pytest.fixture(scope="module", autouse=True)
def _narrow():
    lib = H._emu(H.EMU_LIB)
    lib.ptx_emu_get_narrow.restype = ctypes.c_int
    was = lib.ptx_emu_get_narrow()
    lib.ptx_emu_set_narrow(1)
    yield
    lib.ptx_emu_set_narrow(was)


def test_the_switch_is_on_and_ids_that_do_not_fit_are_not_found():
    lib = H._emu(H.EMU_LIB)
    assert lib.ptx_emu_get_narrow() == 1
    # a reference to an element whose counter is beyond 20 bits, an actor rank beyond 12: "List element not found", not an alias of a real id
    from peritext_amd import wire

    log = [
        {"actor": "a", "seq": 1, "deps": {}, "startOp": 1, "ops": [{"action": "makeList", "obj": None, "key": "text", "opId": "1@a"},
                                                                      {"action": "set", "obj": "1@a", "elemId": None, "insert": True, "value": "x", "opId": "2@a"},
                                                                      {"action": "set", "obj": "1@a", "elemId": "2@a", "insert": True, "value": "y", "opId": "3@a"}]},
    ]
    b = wire.encode_docs([[log]])
    for bad in ((2 + (1 << 20)) << 32, (2 << 32) | (1 << 12)):  # would alias 2@a if the narrow id simply dropped the high bits
        b2 = wire.encode_docs([[log]])
        assert int(b2.ref_a[2]) == 2 << 32
        b2.ref_a[2] = bad
        assert int(H.emu_merge(b2).logs["status"][0]) == 1
    assert int(H.emu_merge(b).logs["status"][0]) == 0


for _mod in (P, R, C_):
    for _name in dir(_mod):
        if _name.startswith("test_"):
            globals()[_name] = getattr(_mod, _name)
del _mod, _name
