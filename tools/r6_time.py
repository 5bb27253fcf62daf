#!/usr/bin/env python3
"""One build, one process: generate a BASELINE config's batch on the device, merge it, print the launch time (HIP events, the library's own stream).
Every build measured this way sees the same allocation sequence in a fresh process, i.e. the same device addresses: engines that share a process do not
(round 6: the first engine of tools/lib_ab.py measured 8.23 ms where the same library measured 8.44 as the first of three).
    python tools/r6_time.py --lib peritext_amd/lib/exp_x.so --config config4 --docs 65536 [--digest]"""
import argparse
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from peritext_amd import abi, workloads  # noqa: E402
from peritext_amd.engine import Engine  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lib", default=None)
    ap.add_argument("--config", default="config4")
    ap.add_argument("--docs", type=int, default=65536)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--flags", type=int, default=abi.FLAG_NO_ELEM_RANK)
    ap.add_argument("--list-cap", type=int, default=2048)
    ap.add_argument("--ops", type=int, default=None, help="ops per log instead of the configuration's own")
    ap.add_argument("--threads", type=int, default=0, help="force the threads per log (ptx_set_launch_shape)")
    ap.add_argument("--regen", type=int, default=1, help="generate the batch this many times (each at other device addresses; the earlier ones stay allocated) and time each")
    args = ap.parse_args()
    lib = args.lib and (args.lib if os.path.isabs(args.lib) else os.path.join(ROOT, args.lib))
    g = workloads.gen_config(args.config, ops=args.ops)
    with Engine(0, flags=args.flags, lib_path=lib) as e:
      if args.threads:
        e.set_launch_shape(args.threads, 0)
      for _regen in range(args.regen):
        db, ginfo = e.generate(g["replicas"], g["ops_per_log"], g["mix"], g["mark_types"], args.docs, 2024, list_cap=args.list_cap)
        dr = e.alloc_result(db)
        e.merge(db, dr)
        e.sync()
        n = e.n_logs(db)
        logs = e.download_logs(dr, n)
        h = hashlib.sha1()
        for k in ("status", "digest", "n_spans", "n_visible"):
            h.update(logs[k].tobytes())
        ms = [e.merge_timed(db, dr, args.iters) / args.iters for _ in range(args.rounds)]
        print(json.dumps({"build": os.path.basename(lib or "libperitext_hip.so"), "config": args.config, "docs": args.docs, "kernel_ms": [round(x, 4) for x in ms], "gen_ms": round(ginfo["kernel_ms"], 3), "min_ms": round(min(ms), 4),
                          "kernel": e.batch_kernel_name(db), "launch": e.launch_shape(db), "max_status": int(logs["status"].max()), "results_sha1": h.hexdigest()[:16]}))


if __name__ == "__main__":
    main()
