#!/usr/bin/env python3
"""Same-box, same-call A/B of two builds of libperitext_hip.so (GPU box; no torch: starts in under a second, so a gpurun call costs ~10 s).
  1. parity of build B: every committed PTXGEN fixture against the oracle's output (tests/helpers.check_generated);
  2. the same generated batch resident under both builds, ptx_merge timed with HIP events on each engine's stream, alternating A / B;
     statuses, digests and row counts of every log compared between the builds.
    python tools/lib_ab.py --b peritext_amd/lib/exp_x.so [peritext_amd/lib/exp_y.so ...] --docs 65536 > gpurun_out/lib_ab.json
Experimental builds: hipcc with the product's flags (__graft_entry__.py) plus -D<macro>=<value> (the tuning macros at the top of merge_core.h), e.g.
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -mllvm -amdgpu-atomic-optimizer-strategy=None -DPTX_S=8 -o peritext_amd/lib/exp_s8.so peritext_amd/csrc/peritext_hip.hip
(16 s each on the build container; the .so files travel to the GPU box with the snapshot)."""
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
    ap.add_argument("--b", required=True, nargs="+", help="build(s) B: each is checked against the fixtures and timed against build A")
    ap.add_argument("--docs", type=int, default=8192)
    ap.add_argument("--config", default="config4")
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--rounds", type=int, default=2)
    ap.add_argument("--flags", type=int, default=abi.FLAG_NO_ELEM_RANK)
    ap.add_argument("--no-parity", action="store_true")
    ap.add_argument("--threads", type=int, default=0, help="force the threads per log of every build (0 = the library's choice)")
    args = ap.parse_args()
    libs = [args.a and os.path.join(ROOT, args.a)] + [os.path.join(ROOT, b) for b in args.b]
    names = [os.path.basename(p or "libperitext_hip.so") for p in libs]
    t0 = time.time()
    out = {"builds": names, "parity_b": {}, "timing": []}

    def say(msg):
        print("[%5.1fs] %s" % (time.time() - t0, msg), file=sys.stderr, flush=True)

    for k in range(1, len(libs) if not args.no_parity else 0):
        with Engine(0, lib_path=libs[k]) as e:
            for name in FIXTURES:
                with open(os.path.join(H.GOLDEN, name)) as f:
                    gen = json.load(f)
                try:
                    H.check_generated(gen, e.apply_materialize)
                    verdict = "ok"
                except Exception as ex:  # noqa: BLE001
                    verdict = "FAIL: " + str(ex).splitlines()[0][:200]
                out["parity_b"]["%s %s" % (names[k], name)] = verdict
                say("parity of %s on %s: %s" % (names[k], name, verdict))
    g = workloads.gen_config(args.config)
    engs = [Engine(0, flags=args.flags, lib_path=p) for p in libs]
    for e in engs:
        if args.threads:
            e.set_launch_shape(args.threads, 0)
    state = []
    for e in engs:
        db, _ = e.generate(g["replicas"], g["ops_per_log"], g["mix"], g["mark_types"], args.docs, 2024, list_cap=2048)
        dr = e.alloc_result(db)
        e.merge(db, dr)
        e.sync()
        state.append((db, dr, e.n_logs(db), e.launch_shape(db)))
    logs = [e.download_logs(dr, n) for e, (db, dr, n, _) in zip(engs, state)]
    same = int(logs[0]["status"].max()) == 0
    for k in range(1, len(libs)):
        same_k = bool((logs[0]["status"] == logs[k]["status"]).all() and (logs[0]["digest"] == logs[k]["digest"]).all() and (logs[0]["n_spans"] == logs[k]["n_spans"]).all()
                      and (logs[0]["n_visible"] == logs[k]["n_visible"]).all())
        say("results of %s identical with build A's over %d logs: %s" % (names[k], state[0][2], same_k))
        same = same and same_k
    out["identical_results"] = same
    out["logs_compared"] = int(state[0][2])
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
