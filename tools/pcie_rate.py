#!/usr/bin/env python3
"""The rate a caller sees who hands over HOST buffers (ptx_batch_upload + ptx_merge + ptx_result_download: the C ABI's ptx_apply_materialize path), beside the
resident rate bench.py reports as `value` (GPU box only).  Pageable host input, one stream, no overlap of copies and compute — the plain path of the boundary;
since ABI 7 the result rows come down compact into pinned memory (round 4: one row per op, 2.4 GB for 100 M ops).
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


def measure(docs=8192, config="config4", reps=3):
    """Upload of host buffers, merge, download of the COMPACT result rows (ABI 7), each timed around the C call itself (the Python driver's numpy copies of the
    result are not what a caller of the C ABI pays; they are reported beside it)."""
    import ctypes as C

    c = workloads.gen_config(config)
    with Engine(0, flags=abi.FLAG_NO_ELEM_RANK) as e:
        db, info = e.generate(c["replicas"], c["ops_per_log"], c["mix"], c["mark_types"], docs, 2024, list_cap=1536)
        actors_t, comments_t, log_doc_t = wire.generated_tables(docs, c["replicas"], info["n_comments"])
        hb = e.download_batch(db, wire.GEN_VALUES, wire.GEN_URLS, log_doc_t, actors_t, comments_t)  # the same documents as host buffers
        e.free_batch(db)
        ops = hb.n_logs * c["ops_per_log"]
        in_bytes = sum(getattr(hb, k).nbytes for k in ("op_id", "ref_a", "ref_b", "payload", "action", "mark_type", "side_a", "side_b", "log_off", "chg_off", "chg_hdr", "chg_env"))
        out = {"config": config, "docs": docs, "replica_logs": hb.n_logs, "ops": ops, "host_input_bytes": in_bytes, "abi": abi.PTX_ABI_VERSION, "runs": []}
        for _ in range(reps):
            t0 = time.time()
            d = e.upload(hb)
            e.sync()
            t1 = time.time()
            dr = e.alloc_result(d)
            e.merge(d, dr)
            e.sync()
            t2 = time.time()
            res = abi.ptx_result()
            e._check(e.lib.ptx_result_download(e.ctx, d, dr, C.byref(res)))
            t3 = time.time()
            nl = int(res.n_logs)
            rows = (int(res.value_off[nl]), int(res.span_off[nl]), int(res.cint_off[nl]))
            out_bytes = nl * 48 + 3 * 8 * (nl + 1) + rows[0] * 4 + rows[1] * 8 + rows[2] * 12
            e.lib.ptx_result_free(C.byref(res))
            t4 = time.time()
            full = e.download(d, dr)  # the same through the Python driver (its numpy copies included)
            t5 = time.time()
            assert (full.logs["status"] == 0).all()
            out["runs"].append({"upload_s": t1 - t0, "merge_s": t2 - t1, "download_s": t3 - t2, "download_python_driver_s": t5 - t4, "result_bytes": out_bytes,
                                "result_rows": {"values": rows[0], "spans": rows[1], "cintervals": rows[2]}, "capacity_bytes_round4": nl * 48 + hb.n_ops * (4 + 8 + 12),
                                "upload_GBps": in_bytes / (t1 - t0) / 1e9, "ops_per_s_host_to_host": ops / (t3 - t0), "ops_per_s_merge_only": ops / (t2 - t1)})
            e.free_result(dr)
            e.free_batch(d)
        best = max(out["runs"], key=lambda r: r["ops_per_s_host_to_host"])
        out["best"] = best
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--docs", type=int, default=8192)
    ap.add_argument("--config", default="config4")
    args = ap.parse_args()
    print(json.dumps(measure(args.docs, args.config)))


if __name__ == "__main__":
    main()
