"""The N > 1 collective path of the C ABI on ONE GPU: two / three processes share GPU 0, each owns a block of the
documents, and ptx_comm_init(…, n) → ptx_allgather_digests → ptx_count_converged_digests run between them — equal blocks (the direct
path), unequal blocks (pack → padded all-gather → ptx_compact_digests_kernel) and PTX_FLAG_PAD_GATHER on equal blocks.  RCCL itself
is replaced by tests/fake_rccl (shared memory between the processes; the library dlopens "librccl.so.1", the workers get that
directory first on LD_LIBRARY_PATH and never import torch): what is tested is the library's code on both sides of the ncclAllGather
call.  Real RCCL over xGMI at N > 1 is what `bench.py --gpus N` runs on a multi-GPU node; it has not run on hardware here.
Replaces the reference's convergence assert over all documents (reference/test/fuzz.ts:277-278) for a sharded batch."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import helpers as H
from peritext_amd import abi, shard, wire

pytestmark = pytest.mark.gpu
FAKE_DIR = os.path.join(H.ROOT, "tests", "fake_rccl")
WORKER = os.path.join(H.ROOT, "tests", "shard_rank_worker.py")


def _docs(n_docs, broken):
    """n_docs documents (3 replicas each) cut from the committed fixtures; document `broken` has one replica that lacks its last changes."""
    docs = []
    for name in ("ptxgen_config4_600.json", "ptxgen_mini.json", "ptxgen_rich_700.json", "ptxgen_config3_512.json"):
        with open(os.path.join(H.GOLDEN, name)) as f:
            docs += [d["logs"] for d in json.load(f)["docs"] if len(d["logs"]) == 3]
    assert len(docs) >= 4
    docs = [docs[i % len(docs)] for i in range(n_docs)]
    if broken is not None:
        d = docs[broken]
        docs[broken] = [d[0], d[1][:-2], d[2]]
    return docs


def _run_ranks(tmp_path, docs, n_ranks, flags=0, rounds=1):
    if not os.path.exists(os.path.join(FAKE_DIR, "librccl.so.1")):
        pytest.skip("tests/fake_rccl/librccl.so.1 not built (run __graft_entry__.build())")
    docs_file, id_file = str(tmp_path / "docs.json"), str(tmp_path / "comm.id")
    with open(docs_file, "w") as f:
        json.dump({"docs": docs}, f)
    env = dict(os.environ, LD_LIBRARY_PATH=FAKE_DIR + ":" + os.environ.get("LD_LIBRARY_PATH", ""))
    procs = [subprocess.Popen([sys.executable, WORKER, str(r), str(n_ranks), id_file, docs_file, str(flags), str(rounds)], cwd=H.ROOT, env=env, stdout=subprocess.PIPE,
                              stderr=subprocess.PIPE, text=True) for r in range(n_ranks)]
    outs = []
    for p in procs:
        o, e = p.communicate(timeout=600)
        assert p.returncode == 0, e[-3000:]
        outs.append(json.loads(o.strip().splitlines()[-1]))
    return outs


def _expected(docs, n_ranks):
    """Digests of every replica log, rank-major, each rank's block encoded on its own as the rank does (value / url ids are tables of the batch, so a
    digest is comparable inside one rank's batch only — which is all the convergence check needs: the replicas of a document share a rank)."""
    import torch  # (first, like the other GPU suites: this process may run them too)

    assert torch.cuda.is_available(), "GPU tests need a real MI355X"
    from peritext_amd.engine import Engine

    dgs = []
    with Engine(0) as e:
        for r in range(n_ranks):
            first, count = shard.doc_range(len(docs), r, n_ranks)
            if count:
                dgs.append(e.apply_materialize(wire.encode_docs(docs[first:first + count])).logs["digest"])
    dg = np.concatenate(dgs)
    conv = int((dg.reshape(-1, 3, 2) == dg.reshape(-1, 3, 2)[:, :1, :]).all(axis=(1, 2)).sum())
    return ["%016x%016x" % (int(a), int(b)) for a, b in dg], conv


@pytest.mark.parametrize("n_docs,n_ranks,flags", [(12, 2, 0), (11, 2, 0), (12, 2, abi.FLAG_PAD_GATHER), (10, 3, 0)])
def test_digest_allgather_between_processes(tmp_path, n_docs, n_ranks, flags):
    """Every rank ends up with every rank's digests, rank-major, and counts the same converged documents; one document (owned by the LAST
    rank) has a replica that lags behind, so the count is n_docs - 1 on every rank.  (11 documents over 2 ranks and 10 over 3 are
    unequal blocks: the padded path; 12 over 2 is the direct path, and again through the padded one under PTX_FLAG_PAD_GATHER.)"""
    docs = _docs(n_docs, broken=n_docs - 1)
    want, want_conv = _expected(docs, n_ranks)
    assert want_conv == n_docs - 1
    outs = _run_ranks(tmp_path, docs, n_ranks, flags=flags, rounds=2)
    sizes = [shard.doc_range(n_docs, r, n_ranks)[1] * 3 for r in range(n_ranks)]
    if flags == 0 and n_docs % n_ranks:
        assert len(set(sizes)) > 1  # really unequal blocks
    for r, o in enumerate(outs):
        assert o["fake_rccl_loaded"], "the worker bound the real RCCL: LD_LIBRARY_PATH did not take"
        assert o["counts"] == sizes
        first = sum(sizes[:r])
        for rnd in o["rounds"]:  # the second round reuses the communicator's cached block table (no host sync in the step)
            assert rnd["own"] == want[first:first + sizes[r]]
            assert rnd["gathered"] == want, (r, flags)
            assert rnd["converged"] == want_conv


def test_a_flipped_digest_is_seen_by_every_rank(tmp_path):
    """The lagging replica sits on rank 0 this time; both ranks must see it in the gathered array."""
    docs = _docs(11, broken=0)
    want, want_conv = _expected(docs, 2)
    outs = _run_ranks(tmp_path, docs, 2)
    assert want_conv == 10 and all(o["rounds"][0]["converged"] == 10 and o["rounds"][0]["gathered"] == want for o in outs)
    healthy, _ = _expected(_docs(11, broken=None), 2)
    assert healthy != want and healthy[3:] == want[3:]


@pytest.mark.skipif(not H.have_node(), reason="node not installed")
def test_digest_allgather_between_node_processes(tmp_path):
    """The same through the JS/TS host: two node processes, MergeEngine.commInit / convergedDocs (N-API mergeAndGather), unequal blocks."""
    addon = os.path.join(H.ROOT, "peritext_amd", "node", "peritext_node.node")
    if not os.path.exists(addon) or not os.path.exists(os.path.join(FAKE_DIR, "librccl.so.1")):
        pytest.skip("N-API addon or tests/fake_rccl not built")
    docs = _docs(11, broken=10)
    want, want_conv = _expected(docs, 2)
    docs_file, id_file = str(tmp_path / "docs.json"), str(tmp_path / "comm.id")
    with open(docs_file, "w") as f:
        json.dump({"docs": docs}, f)
    env = dict(os.environ, LD_LIBRARY_PATH=FAKE_DIR + ":" + os.environ.get("LD_LIBRARY_PATH", ""))
    driver = os.path.join(H.ROOT, "tests", "node_host_check.js")
    procs = [subprocess.Popen([H.NODE, driver, "commrank", str(r), "2", id_file, docs_file], cwd=H.ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
             for r in range(2)]
    for r, p in enumerate(procs):
        o, e = p.communicate(timeout=600)
        assert p.returncode == 0, e[-3000:]
        out = json.loads(o.strip().splitlines()[-1])
        assert out["ok"] and out["counts"] == [18, 15] and out["gathered"] == want and out["converged"] == want_conv == 10 and out["total"] == 11
        assert all(st == 0 for st in out["statuses"])
