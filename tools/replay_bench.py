#!/usr/bin/env python3
"""Throughput of the patch-stream replay (ptx_replay_kernel) on a PTXGEN batch (GPU box only).
    python tools/replay_bench.py [--config config4] [--docs 2048] [--check 2]
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
from peritext_amd import wire, workloads  # noqa: E402
from peritext_amd.engine import Engine  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="config4")
    ap.add_argument("--docs", type=int, default=2048)
    ap.add_argument("--ops", type=int, default=None)
    ap.add_argument("--lib", default=None, help="a candidate build of the library")
    ap.add_argument("--flags", type=int, default=0, help="ptx_create flags (8 = PTX_FLAG_REPLAY_LDS_ONLY)")
    ap.add_argument("--check", type=int, default=0, help="compare the streams of the first CHECK documents with the oracle's (needs node)")
    args = ap.parse_args()
    c = workloads.gen_config(args.config, ops=args.ops)
    eng = Engine(0, flags=args.flags, lib_path=args.lib if (args.lib is None or os.path.isabs(args.lib)) else os.path.join(ROOT, args.lib))
    db, info = eng.generate(c["replicas"], c["ops_per_log"], c["mix"], c["mark_types"], args.docs, 4242, list_cap=2048)
    dr = eng.alloc_result(db)
    eng.merge(db, dr)
    eng.sync()
    t0 = time.time()
    pat = eng.replay_patches(db, dr)
    wall = time.time() - t0
    n_logs = eng.n_logs(db)
    ops = n_logs * c["ops_per_log"]
    n_pat = int(pat.logs["n_patches"].sum())
    bad = np.flatnonzero(pat.logs["status"] != 0)
    assert len(bad) == 0, "logs without a stream: %s status %s n_patches %s launches %d" % (bad[:8], pat.logs["status"][bad[:8]], pat.logs["n_patches"][bad[:8]], pat.launches)
    out = {"lib": os.path.basename(args.lib or "product"), "flags": args.flags, "config": args.config, "logs": n_logs, "ops": ops, "patches": n_pat, "launches": pat.launches, "kernel_ms": pat.kernel_ms,
           "ops_per_s": ops / pat.kernel_ms * 1e3, "patches_per_s": n_pat / pat.kernel_ms * 1e3, "us_per_log": pat.kernel_ms * 1e3 / n_logs,
           "wall_s_incl_download": wall}
    if args.check:
        import helpers as H
        hb, hinfo = eng.generate(c["replicas"], c["ops_per_log"], c["mix"], c["mark_types"], args.check, 4242, list_cap=2048)
        actors_t, comments_t, log_doc_t = wire.generated_tables(args.check, c["replicas"], hinfo["n_comments"])
        batch = eng.download_batch(hb, wire.GEN_VALUES, wire.GEN_URLS, log_doc_t, actors_t, comments_t)
        eng.free_batch(hb)
        docs_logs = [[wire.decode_changes(batch, d * c["replicas"] + r) for r in range(c["replicas"])] for d in range(args.check)]
        exp = H.oracle_apply(docs_logs, patches=True)
        H.check_patch_streams(batch, pat, exp)  # the first documents of the resident batch (same rows, same offsets)
        out["checked_logs"] = batch.n_logs
    print(json.dumps(out), flush=True)
    eng.free_result(dr)
    eng.free_batch(db)
    eng.close()


if __name__ == "__main__":
    main()
