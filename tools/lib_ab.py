#!/usr/bin/env python3
"""Same-box, same-call A/B of two builds of libperitext_hip.so (GPU box; no torch: starts in under a second, so a gpurun call costs ~10 s).
  1. parity of build B: every committed PTXGEN fixture against the oracle's output (tests/helpers.check_generated);
  2. the same generated batch resident under both builds, ptx_merge timed with HIP events on each engine's stream, alternating A / B;
     statuses, digests and row counts of every log compared between the builds.
    python tools/lib_ab.py --b peritext_amd/lib/exp_park.so --docs 65536 > gpurun_out/lib_ab.json"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers as H  # noqa: E402
from peritext_amd import abi, workloads  # noqa: E402
from peritext_amd.engine import Engine  # noqa: E402

FIXTURES = ["ptxgen_mini.json", "ptxgen_config2.json", "ptxgen_config3_512.json", "ptxgen_config4_600.json", "ptxgen_rich_700.json", "ptxgen_rich_2600.json",
            "ptxgen_config5_8192.json", "ptxgen_mini_10actors.json"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--a", default=None, help="build A (default: peritext_amd/lib/libperitext_hip.so)")
    ap.add_argument("--b", required=True, help="build B")
    ap.add_argument("--docs", type=int, default=8192)
    ap.add_argument("--config", default="config4")
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--rounds", type=int, default=2)
    ap.add_argument("--flags", type=int, default=abi.FLAG_NO_ELEM_RANK)
    ap.add_argument("--no-parity", action="store_true")
    args = ap.parse_args()
    libs = [args.a and os.path.join(ROOT, args.a), os.path.join(ROOT, args.b)]
    names = [os.path.basename(p or "libperitext_hip.so") for p in libs]
    t0 = time.time()
    out = {"builds": names, "parity_b": {}, "timing": []}

    def say(msg):
        print("[%5.1fs] %s" % (time.time() - t0, msg), file=sys.stderr, flush=True)

    if not args.no_parity:
        with Engine(0, lib_path=libs[1]) as e:
            for name in FIXTURES:
                with open(os.path.join(H.GOLDEN, name)) as f:
                    gen = json.load(f)
                try:
                    H.check_generated(gen, e.apply_materialize)
                    out["parity_b"][name] = "ok"
                except Exception as ex:  # noqa: BLE001
                    out["parity_b"][name] = "FAIL: " + str(ex).splitlines()[0][:200]
                say("parity of %s on %s: %s" % (names[1], name, out["parity_b"][name]))
    g = workloads.gen_config(args.config)
    engs = [Engine(0, flags=args.flags, lib_path=p) for p in libs]
    state = []
    for e in engs:
        db, _ = e.generate(g["replicas"], g["ops_per_log"], g["mix"], g["mark_types"], args.docs, 2024, list_cap=2048)
        dr = e.alloc_result(db)
        e.merge(db, dr)
        e.sync()
        state.append((db, dr, e.n_logs(db), e.launch_shape(db)))
    logs = [e.download_logs(dr, n) for e, (db, dr, n, _) in zip(engs, state)]
    same = bool((logs[0]["status"] == logs[1]["status"]).all() and (logs[0]["digest"] == logs[1]["digest"]).all() and (logs[0]["n_spans"] == logs[1]["n_spans"]).all()
                and (logs[0]["n_visible"] == logs[1]["n_visible"]).all() and int(logs[0]["status"].max()) == 0)
    out["identical_results"] = same
    out["logs_compared"] = int(state[0][2])
    say("results of the two builds identical over %d logs: %s" % (state[0][2], same))
    for rnd in range(args.rounds):
        for k, e in enumerate(engs):
            db, dr, n_logs, shape = state[k]
            ms = e.merge_timed(db, dr, args.iters) / args.iters
            row = {"round": rnd, "build": names[k], "kernel_ms": ms, "docs": args.docs, "config": args.config, "launch": shape,
                   "lds_high": int(logs[k]["reserved"][:, 0].max()), "ops_per_s": n_logs * g["ops_per_log"] / (ms * 1e-3)}
            out["timing"].append(row)
            say(json.dumps(row))
    print(json.dumps(out))
    bad = [k for k, v in out["parity_b"].items() if v != "ok"]
    sys.exit(1 if bad or not same else 0)


if __name__ == "__main__":
    main()
