"""One rank of tests/test_gpu_shard_ranks.py: a process of its own (no torch: the real RCCL is not in the process; the test puts
tests/fake_rccl first on LD_LIBRARY_PATH so that the library's dlopen("librccl.so.1") binds the stand-in) that owns a contiguous block
of the documents, merges it on GPU 0 and takes part in the digest all-gather of the C ABI.
    python tests/shard_rank_worker.py <rank> <n_ranks> <id_file> <docs.json> <flags> <rounds>
Prints one JSON line: this rank's digests, the gathered digests, the device-side count of converged documents, per round."""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from peritext_amd import abi, shard, wire  # noqa: E402
from peritext_amd.engine import Engine  # noqa: E402


def main():
    rank, n_ranks, id_file, docs_file, flags, rounds = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4], int(sys.argv[5]), int(sys.argv[6])
    with open(docs_file) as f:
        docs = json.load(f)["docs"]
    replicas = len(docs[0])
    first, count = shard.doc_range(len(docs), rank, n_ranks)
    counts = [shard.doc_range(len(docs), r, n_ranks)[1] * replicas for r in range(n_ranks)]
    eng = Engine(0, flags=flags)
    # the 128-byte id travels over the host's own channel (here: a file), as ptx_comm_unique_id documents
    if rank == 0:
        uid = eng.comm_unique_id()
        with open(id_file + ".tmp", "wb") as f:
            f.write(uid)
        os.replace(id_file + ".tmp", id_file)
    else:
        t0 = time.time()
        while not os.path.exists(id_file):
            if time.time() - t0 > 60:
                raise SystemExit("rank 0 never published the communicator id")
            time.sleep(0.01)
        with open(id_file, "rb") as f:
            uid = f.read()
    comm = eng.comm_init(uid, rank, n_ranks)
    assert eng.lib.ptx_comm_n_ranks(comm) == n_ranks and eng.lib.ptx_comm_rank(comm) == rank
    total = sum(counts)
    gathered = C.c_void_p()
    conv = C.c_void_p()
    assert eng.lib.ptx_device_alloc(eng.ctx, total * 16 + 16, C.byref(gathered)) == 0
    assert eng.lib.ptx_device_alloc(eng.ctx, 8, C.byref(conv)) == 0
    out = {"rank": rank, "counts": counts, "rounds": []}
    batch = wire.encode_docs(docs[first:first + count]) if count else None
    db = eng.upload(batch) if count else None
    dr = eng.alloc_result(db) if count else None
    for rnd in range(rounds):
        if count:
            eng.merge(db, dr)
        if count:
            eng.allgather_digests(comm, dr, counts, gathered.value)
        eng.count_converged_digests(gathered.value, total, replicas, conv.value)
        eng.sync()
        host = np.zeros((total, 2), dtype=np.uint64)
        assert eng.lib.ptx_device_read(eng.ctx, gathered, host.ctypes.data_as(C.c_void_p), total * 16) == 0
        c = np.zeros(1, dtype=np.uint64)
        assert eng.lib.ptx_device_read(eng.ctx, conv, c.ctypes.data_as(C.c_void_p), 8) == 0
        own = eng.download_logs(dr, eng.n_logs(db))["digest"] if count else np.zeros((0, 2), dtype=np.uint64)
        out["rounds"].append({"own": ["%016x%016x" % (int(a), int(b)) for a, b in own], "gathered": ["%016x%016x" % (int(a), int(b)) for a, b in host],
                              "converged": int(c[0])})
    try:
        fake = C.CDLL("librccl.so.1")
        out["fake_rccl_loaded"] = hasattr(fake, "ptx_fake_rccl_calls")
    except OSError:
        out["fake_rccl_loaded"] = False
    eng.comm_destroy(comm)
    eng.lib.ptx_device_free(eng.ctx, gathered)
    eng.lib.ptx_device_free(eng.ctx, conv)
    if count:
        eng.free_result(dr)
        eng.free_batch(db)
    eng.close()
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
