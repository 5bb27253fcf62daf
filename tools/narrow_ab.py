#!/usr/bin/env python3
"""Same-box A/B of the two encodings of the id / side columns the merge kernel can read (GPU box, no torch: starts in seconds):
the wire columns (64-bit ids) against the narrow mirror (PTX_FLAG_NARROW_IDS: 32-bit ids, both sides in one byte).
  1. parity: every committed PTXGEN fixture through an engine with the flag, against the oracle's output (tests/helpers.check_generated);
  2. timing: one generated config batch, ptx_merge timed with HIP events on the engine's stream, mirror off / on / off / on on the SAME
     resident batch; digests, statuses and row counts of the two encodings compared.
    python tools/narrow_ab.py --docs 8192 --iters 10 > gpurun_out/narrow_ab.json"""
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
    ap.add_argument("--docs", type=int, default=8192)
    ap.add_argument("--config", default="config4")
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--rounds", type=int, default=2)
    ap.add_argument("--no-parity", action="store_true")
    args = ap.parse_args()
    t0 = time.time()
    out = {"parity": {}, "timing": []}
    if not args.no_parity:
        with Engine(0, flags=abi.FLAG_NARROW_IDS) as e:
            assert e.flags() & abi.FLAG_NARROW_IDS
            for name in FIXTURES:
                with open(os.path.join(H.GOLDEN, name)) as f:
                    gen = json.load(f)
                try:
                    H.check_generated(gen, e.apply_materialize)
                    out["parity"][name] = "ok"
                except Exception as ex:  # noqa: BLE001
                    out["parity"][name] = "FAIL: " + str(ex).splitlines()[0][:200]
                print("[%5.1fs] narrow parity %s: %s" % (time.time() - t0, name, out["parity"][name]), file=sys.stderr, flush=True)
    g = workloads.gen_config(args.config)
    with Engine(0, flags=abi.FLAG_NO_ELEM_RANK) as e:
        db, info = e.generate(g["replicas"], g["ops_per_log"], g["mix"], g["mark_types"], args.docs, 2024, list_cap=2048)
        dr = e.alloc_result(db)
        n_logs = e.n_logs(db)
        ref = None
        for rnd in range(args.rounds):
            for on in (False, True):
                e.narrow_mirror(db, on)
                assert e.has_narrow_mirror(db) == on
                e.merge(db, dr)
                e.sync()
                ms = e.merge_timed(db, dr, args.iters) / args.iters
                logs = e.download_logs(dr, n_logs)
                assert int(logs["status"].max()) == 0
                if ref is None:
                    ref = logs
                same = bool((logs["digest"] == ref["digest"]).all() and (logs["n_spans"] == ref["n_spans"]).all() and (logs["n_visible"] == ref["n_visible"]).all())
                row = {"round": rnd, "id_columns": "narrow mirror" if on else "wire columns", "kernel_ms": ms, "docs": args.docs, "config": args.config,
                       "ops_per_s": n_logs * g["ops_per_log"] / (ms * 1e-3), "same_digests_as_first": same}
                out["timing"].append(row)
                print("[%5.1fs] %s" % (time.time() - t0, json.dumps(row)), file=sys.stderr, flush=True)
        e.free_result(dr)
        e.free_batch(db)
    print(json.dumps(out))
    bad = [k for k, v in out["parity"].items() if v != "ok"] + [r for r in out["timing"] if not r["same_digests_as_first"]]
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
