"""bench.py's own N > 1 branch on ONE GPU: the driver's 8-GPU run launches `python -m torch.distributed.run --nproc-per-node N bench.py
--gpus N`; here the same launcher starts two (and three) ranks that share GPU 0.  torch's side channel (the 128-byte communicator id, the barriers, the max
over ranks of the timings) falls back to gloo because two NCCL ranks cannot share a device; the data-path collective — ptx_comm_init → ptx_allgather_digests →
ptx_count_converged_digests, all inside the library, on the step's stream — runs as it does on a multi-GPU node, with tests/fake_rccl bound through
PTX_RCCL_LIB (ptx_comm_use_library) in place of RCCL.  Real RCCL over xGMI at N > 1 remains unmeasured here (SCALE_rNN.json is the driver's to produce).
Stands for the reference's convergence assert over all documents (reference/test/fuzz.ts:277-278) on a sharded batch."""
import json
import os
import subprocess
import sys

import pytest

import helpers as H

pytestmark = pytest.mark.gpu
FAKE = os.path.join(H.ROOT, "tests", "fake_rccl", "librccl.so.1")


@pytest.mark.parametrize("n_ranks,docs", [(2, 2048), (3, 1000)])
def test_bench_runs_its_n_gt_1_branch_with_ranks_sharing_one_gpu(n_ranks, docs):
    if not os.path.exists(FAKE):
        pytest.skip("tests/fake_rccl/librccl.so.1 not built (run __graft_entry__.build())")
    env = dict(os.environ, PTX_RCCL_LIB=FAKE, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_ranks), "--master-addr", "127.0.0.1", "--master-port", str(29511 + n_ranks),
           os.path.join(H.ROOT, "bench.py"), "--gpus", str(n_ranks), "--docs", str(docs), "--steps", "2", "--warmup", "1", "--no-cpu", "--no-extras", "--sustain-s", "0"]
    p = subprocess.run(cmd, cwd=H.ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, (p.stdout[-2000:], p.stderr[-4000:])
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]  # rank 0 prints ONE JSON line
    out = json.loads(lines[0])
    assert out["n_gpus"] == n_ranks and out["steps"] == 2 and out["scaling"] == "strong"
    assert out["docs_total"] == docs and out["docs_converged"] == docs
    assert ("x%d" % n_ranks) in out["config"]["parallelism"] and "ptx_allgather_digests" in out["config"]["parallelism"]
    assert out["config"]["docs_this_gpu"] in (docs // n_ranks, docs // n_ranks + 1) and out["config"]["docs_total"] == docs
    assert out["value"] > 0 and out["roofline"]["frac"] > 0
