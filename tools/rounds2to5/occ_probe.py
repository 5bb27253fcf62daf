#!/usr/bin/env python3
"""Logs per CU against throughput on the SAME logs: one generated batch merged under several LDS windows per log (ptx_set_launch_shape), i.e. 10 / 9 / 8 / 7 ... logs
resident per CU (the CU hands out LDS in 1 280-byte granules: tools/micro/occupancy_query.hip), for one or more builds of the library.  GPU box; no torch.
    python tools/occ_probe.py --ops 3200 --docs 16384 --lds 16640,19200,23040   (granules of 1 280 B: 9 / 8 / 7 logs per CU) [--lib peritext_amd/lib/exp_w8.so ...]"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from peritext_amd import abi, workloads  # noqa: E402
from peritext_amd.engine import Engine  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="config4")
    ap.add_argument("--ops", type=int, default=None)
    ap.add_argument("--docs", type=int, default=16384)
    ap.add_argument("--lds", default="0")
    ap.add_argument("--threads", default="0")
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--lib", nargs="*", default=[None])
    ap.add_argument("--flags", type=int, default=abi.FLAG_NO_ELEM_RANK)
    ap.add_argument("--list-cap", type=int, default=2048)
    args = ap.parse_args()
    c = workloads.gen_config(args.config, ops=args.ops)
    for lib in args.lib:
        path = lib if (lib is None or os.path.isabs(lib)) else os.path.join(ROOT, lib)
        for t in [int(x) for x in args.threads.split(",")]:
            for lds in [int(x) for x in args.lds.split(",")]:
                with Engine(0, flags=args.flags, lib_path=path) as e:
                    e.set_launch_shape(t, lds)
                    db, _ = e.generate(c["replicas"], c["ops_per_log"], c["mix"], c["mark_types"], args.docs, 2024, list_cap=args.list_cap)
                    dr = e.alloc_result(db)
                    e.merge(db, dr)
                    e.sync()
                    ms = min(e.merge_timed(db, dr, args.iters) / args.iters for _ in range(2))
                    logs = e.download_logs(dr, e.n_logs(db))
                    n_logs = e.n_logs(db)
                    shape = e.launch_shape(db)
                    row = {"lib": os.path.basename(lib or "product"), "config": args.config, "ops": c["ops_per_log"], "docs": args.docs, "launch": shape,
                           "logs_per_cu_by_lds": (160 * 1024) // max(1280, (shape[1] + 1279) // 1280 * 1280), "ms": ms, "Gops_s": n_logs * c["ops_per_log"] / ms / 1e6,
                           "us_per_log_per_cu": ms * 1e3 * 256 / n_logs, "lds_high": int(logs["reserved"][:, 0].max()), "ok": bool(int(logs["status"].max()) == 0),
                           "digest_xor": "%016x" % int(abs(int(logs["digest"].astype("uint64").sum())) & 0xFFFFFFFFFFFFFFFF)}
                    print(json.dumps(row), flush=True)
                    e.free_result(dr)
                    e.free_batch(db)


if __name__ == "__main__":
    main()
