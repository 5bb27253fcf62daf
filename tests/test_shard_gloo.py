"""The N>1 path on CPU: world_size-2 `gloo` job (torch.distributed.run, 127.0.0.1) — document sharding, the
digests-only all-gather and the global convergence count (peritext_amd/shard.py; bench.py uses the same functions
over RCCL)."""
import json
import os
import re
import socket
import subprocess
import sys

import pytest

import helpers as H
from peritext_amd import shard


def test_doc_range_partitions_every_document_once():
    for n in (0, 1, 7, 8, 64, 65536 + 3):
        for world in (1, 2, 3, 8):
            spans = [shard.doc_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and sum(c for _, c in spans) == n
            for (f0, c0), (f1, _) in zip(spans, spans[1:]):
                assert f0 + c0 == f1
            assert max(c for _, c in spans) - min(c for _, c in spans) <= 1


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("corrupt,n_docs", [(0, 12), (1, 12), (1, 11)])
def test_world_size_2_gloo_digest_allgather(corrupt, n_docs):
    """n_docs = 11: the two ranks own 6 and 5 documents — unequal all-gather blocks (gathered padded, then compacted)."""
    fixture = "ptxgen_mini.json"  # 12 documents x 3 replicas
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(H.ROOT, "tests", "gloo_worker.py"), fixture, str(corrupt), str(n_docs)]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    p = subprocess.run(cmd, cwd=H.ROOT, capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    rows = [json.loads(m) for m in re.findall(r"RESULT (\{[^{}]*\})", p.stdout)]  # two ranks may share a line
    assert sorted(r["rank"] for r in rows) == [0, 1]
    assert n_docs <= len(json.load(open(os.path.join(H.GOLDEN, fixture)))["docs"])
    for r in rows:
        assert r["world"] == 2 and r["total"] == n_docs
        assert r["converged"] == n_docs - corrupt  # every rank sees the same global count
    assert sum(r["count"] for r in rows) == n_docs and {r["first"] for r in rows} == {0, (n_docs + 1) // 2}
    assert sum(r["local_converged"] for r in rows) == n_docs - corrupt
