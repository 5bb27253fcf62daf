"""Worker of tests/test_shard_gloo.py: one rank of a world_size-N gloo job on CPU.  Each rank owns a contiguous block
of the fixture's documents, holds the digests of its replica logs (computed from the committed oracle output with
canon.digest — the same value the kernel writes, which the GPU tests check), all-gathers them and reports the global
convergence count.  One rank flips a digest bit to prove divergence is seen by EVERY rank."""
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers as H  # noqa: E402
from peritext_amd import canon, shard, wire  # noqa: E402


def main():
    fixture, corrupt = sys.argv[1], int(sys.argv[2])
    n_use = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    gen = json.load(open(os.path.join(H.GOLDEN, fixture)))
    docs = gen["docs"][:n_use] if n_use else gen["docs"]
    replicas = len(docs[0]["logs"])
    first, count = shard.doc_range(len(docs), rank, world)
    counts = [shard.doc_range(len(docs), r, world)[1] * replicas for r in range(world)]  # blocks may differ by one document
    mine = docs[first : first + count]
    batch = wire.encode_docs([d["logs"] for d in mine])
    rows = []
    log = 0
    for d in mine:
        for exp in d["expected"]:
            dd = batch.log_doc[log]
            value_ix = {v: i for i, v in enumerate(batch.values)}
            url_ix = {u: i for i, u in enumerate(batch.urls)}
            crank = {c: i for i, c in enumerate(batch.doc_comments[dd])}
            ev, es, ec = canon.canonical_from_spans(exp["spans"], exp["text"], value_ix, url_ix, crank)
            n_elems = int(np.count_nonzero(batch.action[int(batch.log_off[log]) : int(batch.log_off[log + 1])] == 1))
            rows.append(canon.digest(ev, es, ec, n_elems))
            log += 1
    dg = torch.from_numpy(np.asarray(rows, dtype=np.uint64).view(np.int64).reshape(-1, 2).copy())
    if corrupt and rank == world - 1:
        dg[1, 0] ^= 1  # second replica of this rank's first document now disagrees
    conv, total = shard.global_convergence(dg, replicas, dist, None, counts)
    local = int(shard.converged_docs(dg, replicas))
    out = {"rank": rank, "world": world, "first": first, "count": count, "converged": int(conv), "total": int(total), "local_converged": local}
    print("RESULT " + json.dumps(out), flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
