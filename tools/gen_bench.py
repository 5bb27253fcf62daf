#!/usr/bin/env python3
"""Throughput of the on-device change() generator (ptx_gen_kernel) + merge of what it made (GPU box only).
    python tools/gen_bench.py [--config config4] [--docs 2048] [--list-cap 2048]
Prints one JSON line: documents / ops generated per second, kernel ms, and the merge of the generated (all distinct) batch."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from peritext_amd import abi, workloads as H  # noqa: E402
from peritext_amd.engine import Engine  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="config4")
    ap.add_argument("--docs", type=int, default=2048)
    ap.add_argument("--ops", type=int, default=None)
    ap.add_argument("--seed", type=int, default=2024)
    ap.add_argument("--list-cap", type=int, default=0)
    args = ap.parse_args()
    c = H.gen_config(args.config, ops=args.ops)
    eng = Engine(0, flags=abi.FLAG_NO_ELEM_RANK)
    h, info = eng.generate(c["replicas"], c["ops_per_log"], c["mix"], c["mark_types"], args.docs, args.seed, list_cap=args.list_cap)
    n_logs = args.docs * c["replicas"]
    ops = n_logs * c["ops_per_log"]
    dr = eng.alloc_result(h)
    eng.merge(h, dr)
    eng.sync()
    ms = eng.merge_timed(h, dr, 5) / 5
    logs = eng.download_logs(dr, n_logs)
    assert int(logs["status"].max()) == 0
    d = logs["digest"].reshape(args.docs, c["replicas"], 2)
    converged = int((d == d[:, :1, :]).all(axis=(1, 2)).sum())
    out = {"config": args.config, "docs": args.docs, "logs": n_logs, "ops": ops, "gen_kernel_ms": info["kernel_ms"], "docs_per_s": args.docs / info["kernel_ms"] * 1e3,
           "ops_generated_per_s": ops / info["kernel_ms"] * 1e3, "launch": eng.launch_shape(h), "merge_ms": ms, "merge_Gops_s": ops / ms / 1e6,
           "docs_converged": converged, "distinct_digests": len({(int(x[0]), int(x[1])) for x in d[:, 0, :]})}
    print(json.dumps(out), flush=True)
    eng.free_result(dr)
    eng.free_batch(h)
    eng.close()


if __name__ == "__main__":
    main()
