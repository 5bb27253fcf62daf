#!/usr/bin/env python3
"""Per-phase cycle breakdown + launch-shape sweep of ptx_merge_kernel on a PTXGEN batch (GPU box only).
    python tools/phase_profile.py [--config config4] [--unique 16] [--docs 2048] [--threads 128,256,512,1024]"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from peritext_amd import abi, wire  # noqa: E402
from peritext_amd.engine import Engine  # noqa: E402

PHASES = ["P1 classify", "P2 index+lists", "P3a buckets", "P3b child order", "P3c tour+rank", "P4 tombstones", "P5a values+intervals",
          "P5b LWW trees", "P5c comments", "P6 spans+digest"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="config4")
    ap.add_argument("--unique", type=int, default=16)
    ap.add_argument("--docs", type=int, default=2048)
    ap.add_argument("--ops", type=int, default=None)
    ap.add_argument("--threads", default="128,256,512,1024")
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--lib", default=None, help="experimental build of libperitext_hip.so")
    ap.add_argument("--variants", default="0", help="PTX_VARIANT values to sweep (0 = 128-VGPR kernel, 6, 8)")
    ap.add_argument("--no-phases", action="store_true")
    ap.add_argument("--no-check", action="store_true")
    ap.add_argument("--flags", type=int, default=abi.FLAG_NO_ELEM_RANK, help="ptx_create flags (1 no elem_rank, 2 no admission)")
    args = ap.parse_args()
    if args.lib and not os.path.isabs(args.lib):
        args.lib = os.path.join(ROOT, args.lib)
    docs = bench.gen_unique_docs(args.config, args.unique, 4242, ops=args.ops)
    batch = wire.encode_docs([d["logs"] for d in docs])
    copies = max(1, args.docs // args.unique)
    out = {"config": args.config, "logs": batch.n_logs * copies, "ops": batch.counted_ops() * copies, "shapes": []}
    for t, var in [(int(x), int(v)) for v in args.variants.split(",") for x in args.threads.split(",")]:
        if var and t > (512 if var == 8 else 256):
            continue
        os.environ["PTX_THREADS"] = str(t)
        os.environ["PTX_VARIANT"] = str(var)
        eng = Engine(0, flags=args.flags, lib_path=args.lib)
        db = eng.upload(batch, copies=copies)
        dr = eng.alloc_result(db)
        eng.merge(db, dr)
        eng.sync()
        ms = eng.merge_timed(db, dr, args.iters) / args.iters
        cyc = [0] * 16 if args.no_phases else eng.phase_cycles(db, dr)
        logs = eng.download_logs(dr, eng.n_logs(db))
        assert args.no_check or int(logs["status"].max()) == 0
        tot = sum(cyc) or 1
        row = {"lib": os.path.basename(args.lib or "default"), "variant": var, "threads": t, "ms": ms, "Gops_s": out["ops"] / ms / 1e6, "us_per_log_per_cu": ms * 1e3 * 256 / out["logs"],
               "lds_high": int(logs["reserved"][:, 0].max()), "launch": eng.launch_shape(db), "cycles_per_log": tot / out["logs"],
               "phases": {PHASES[k] if k < len(PHASES) else str(k): round(cyc[k] / out["logs"]) for k in range(len(cyc)) if cyc[k]}}
        out["shapes"].append(row)
        print(json.dumps(row), flush=True)
        eng.free_result(dr)
        eng.free_batch(db)
        eng.close()
    del os.environ["PTX_THREADS"]


if __name__ == "__main__":
    main()
