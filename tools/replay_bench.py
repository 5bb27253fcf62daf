#!/usr/bin/env python3
"""Throughput of the patch-stream replay (ptx_replay_kernel) on a PTXGEN batch (GPU box only).
    python tools/replay_bench.py [--config config4] [--unique 8] [--docs 2048] [--check]
Prints one JSON line: ops replayed per second, patches per second, kernel ms (HIP events inside the library)."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402
from peritext_amd import wire  # noqa: E402
from peritext_amd.engine import Engine  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="config4")
    ap.add_argument("--unique", type=int, default=8)
    ap.add_argument("--docs", type=int, default=2048)
    ap.add_argument("--ops", type=int, default=None)
    ap.add_argument("--check", action="store_true", help="compare the streams of the unique documents with the oracle's")
    args = ap.parse_args()
    docs = bench.gen_unique_docs(args.config, args.unique, 4242, ops=args.ops)
    batch = wire.encode_docs([d["logs"] for d in docs])
    copies = max(1, args.docs // args.unique)
    eng = Engine(0)
    db = eng.upload(batch, copies=copies)
    dr = eng.alloc_result(db)
    eng.merge(db, dr)
    eng.sync()
    t0 = time.time()
    pat = eng.replay_patches(db, dr)
    wall = time.time() - t0
    n_logs = batch.n_logs * copies
    ops = batch.counted_ops() * copies
    n_pat = int(pat.logs["n_patches"].sum())
    bad = np.flatnonzero(pat.logs["status"] != 0)
    assert len(bad) == 0, "logs without a stream: %s status %s n_patches %s launches %d" % (bad[:8], pat.logs["status"][bad[:8]], pat.logs["n_patches"][bad[:8]], pat.launches)
    out = {"config": args.config, "logs": n_logs, "ops": ops, "patches": n_pat, "launches": pat.launches, "kernel_ms": pat.kernel_ms,
           "ops_per_s": ops / pat.kernel_ms * 1e3, "patches_per_s": n_pat / pat.kernel_ms * 1e3, "us_per_log": pat.kernel_ms * 1e3 / n_logs,
           "wall_s_incl_download": wall}
    if args.check:
        import helpers as H
        exp = H.oracle_apply([d["logs"] for d in docs], patches=True)
        H.check_patch_streams(batch, pat, exp)  # the first copy of every unique document
        out["checked_logs"] = batch.n_logs
    print(json.dumps(out), flush=True)
    eng.free_result(dr)
    eng.free_batch(db)
    eng.close()


if __name__ == "__main__":
    main()
