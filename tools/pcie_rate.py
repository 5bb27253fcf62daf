#!/usr/bin/env python3
"""The rate a caller sees who hands over HOST buffers (ptx_batch_upload + ptx_merge + ptx_result_download: the C ABI's ptx_apply_materialize path), beside the
resident rate bench.py reports as `value` (GPU box only).  Pageable host memory, one stream, no overlap of copies and compute — the plain path of the boundary.
    python tools/pcie_rate.py [--docs 8192]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from peritext_amd import abi, wire, workloads  # noqa: E402
from peritext_amd.engine import Engine  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--docs", type=int, default=8192)
    ap.add_argument("--config", default="config4")
    args = ap.parse_args()
    c = workloads.gen_config(args.config)
    with Engine(0, flags=abi.FLAG_NO_ELEM_RANK) as e:
        db, info = e.generate(c["replicas"], c["ops_per_log"], c["mix"], c["mark_types"], args.docs, 2024, list_cap=1536)
        actors_t, comments_t, log_doc_t = wire.generated_tables(args.docs, c["replicas"], info["n_comments"])
        hb = e.download_batch(db, wire.GEN_VALUES, wire.GEN_URLS, log_doc_t, actors_t, comments_t)  # the same documents as host buffers
        e.free_batch(db)
        ops = hb.n_logs * c["ops_per_log"]
        in_bytes = sum(getattr(hb, k).nbytes for k in ("op_id", "ref_a", "ref_b", "payload", "action", "mark_type", "side_a", "side_b", "log_off", "chg_off", "chg_hdr", "chg_env"))
        out = {"config": args.config, "docs": args.docs, "replica_logs": hb.n_logs, "ops": ops, "host_input_bytes": in_bytes}
        for rep in range(2):
            t0 = time.time()
            d = e.upload(hb)
            e.sync()
            t1 = time.time()
            dr = e.alloc_result(d)
            e.merge(d, dr)
            e.sync()
            t2 = time.time()
            res = e.download(d, dr)
            t3 = time.time()
            assert (res.logs["status"] == 0).all()
            out_bytes = res.logs.nbytes + res.values.nbytes + res.spans.nbytes + res.cintervals.nbytes
            out["run%d" % rep] = {"upload_s": t1 - t0, "merge_s": t2 - t1, "download_s": t3 - t2, "result_bytes": out_bytes,
                                  "upload_GBps": in_bytes / (t1 - t0) / 1e9, "ops_per_s_host_to_host": ops / (t3 - t0), "ops_per_s_merge_only": ops / (t2 - t1)}
            e.free_result(dr)
            e.free_batch(d)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
