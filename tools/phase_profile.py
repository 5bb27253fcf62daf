#!/usr/bin/env python3
"""Per-phase cycle breakdown + launch-shape sweep of ptx_merge_kernel on a device-generated PTXGEN batch (GPU box only).
    python tools/phase_profile.py [--config config4] [--docs 8192] [--threads 128,192,256] [--flags 1] [--no-phases]"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from peritext_amd import abi, workloads  # noqa: E402
from peritext_amd.engine import Engine  # noqa: E402

PHASES = {0: "P0 admission", 1: "P1 row loop", 11: "P1 tail (census, dup, scan)", 2: "P3a index+parents", 12: "P3b checks+scatter", 3: "P3c child order", 4: "P3d after()",
          14: "P3d successor words", 16: "P3d jumping rounds", 17: "P3d positions", 18: "unpark marks", 5: "P4 tombstones", 6: "P5a values", 13: "P5a mark intervals",
          7: "P5c comments", 8: "P5b trees + P6 spans", 10: "end"}
ORDER = [0, 1, 11, 2, 12, 3, 4, 14, 16, 17, 18, 5, 6, 13, 7, 8]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="config4")
    ap.add_argument("--docs", type=int, default=8192)
    ap.add_argument("--ops", type=int, default=None)
    ap.add_argument("--threads", default="0", help="threads per log to sweep (0 = the library's own choice)")
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--stop-after", type=int, default=0, help="diagnostic builds (-DPTX_DIAG): truncate the kernel after the phase with this stamp index")
    ap.add_argument("--lib", default=None, help="experimental build of libperitext_hip.so")
    ap.add_argument("--no-phases", action="store_true")
    ap.add_argument("--no-check", action="store_true")
    ap.add_argument("--flags", type=int, default=abi.FLAG_NO_ELEM_RANK, help="ptx_create flags (1 no elem_rank, 2 no admission)")
    ap.add_argument("--list-cap", type=int, default=2048)
    args = ap.parse_args()
    if args.lib and not os.path.isabs(args.lib):
        args.lib = os.path.join(ROOT, args.lib)
    c = workloads.gen_config(args.config, ops=args.ops)
    for t in [int(x) for x in args.threads.split(",")]:
        eng = Engine(0, flags=args.flags, lib_path=args.lib)
        if t:
            eng.set_launch_shape(t, 0)
        if args.stop_after:  # -DPTX_DIAG builds only (the symbol is not part of the ABI): the diagnostic kernel leaves after a phase
            import ctypes
            eng.lib.ptx_diag_stop_after.argtypes = [ctypes.c_void_p, ctypes.c_uint32]
            assert eng.lib.ptx_diag_stop_after(eng.ctx, args.stop_after) == 0
        db, info = eng.generate(c["replicas"], c["ops_per_log"], c["mix"], c["mark_types"], args.docs, 2024, list_cap=args.list_cap)
        n_logs = eng.n_logs(db)
        ops = n_logs * c["ops_per_log"]
        dr = eng.alloc_result(db)
        eng.merge(db, dr)
        eng.sync()
        ms = eng.merge_timed(db, dr, args.iters) / args.iters
        cyc = [0] * 16 if args.no_phases else eng.phase_cycles(db, dr)
        logs = eng.download_logs(dr, n_logs)
        assert args.no_check or int(logs["status"].max()) == 0
        tot = sum(cyc) or 1
        row = {"lib": os.path.basename(args.lib or "default"), "config": args.config, "threads": t, "flags": args.flags, "ms": ms, "Gops_s": ops / ms / 1e6,
               "us_per_log_per_cu": ms * 1e3 * 256 / n_logs, "lds_high": int(logs["reserved"][:, 0].max()), "launch": eng.launch_shape(db),
               "cycles_per_log": tot / n_logs, "gen_ms": info["kernel_ms"],
               "phases": {PHASES.get(k, str(k)): round(cyc[k] / n_logs) for k in ORDER if k < len(cyc) and cyc[k]}}
        print(json.dumps(row), flush=True)
        eng.free_result(dr)
        eng.free_batch(db)
        eng.close()


if __name__ == "__main__":
    main()
