#!/usr/bin/env python3
"""The command tools/pmc_traffic.sh profiles: the bench workload resident in HBM, the calibration stream twice, the merge
three times (with causal admission, as bench.py times it).  Prints one JSON line describing the workload."""
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
    ap.add_argument("--docs", type=int, default=65536)
    ap.add_argument("--seed", type=int, default=2024)
    args = ap.parse_args()
    c = workloads.gen_config(args.config)
    eng = Engine(0, flags=abi.FLAG_NO_ELEM_RANK)
    db, _ = eng.generate(c["replicas"], c["ops_per_log"], c["mix"], c["mark_types"], args.docs, args.seed, list_cap=2048)
    dr = eng.alloc_result(db)
    known = [eng.calib_stream(db) for _ in range(2)][0]
    for _ in range(3):
        eng.merge(db, dr)
    eng.sync()
    logs = eng.download_logs(dr, eng.n_logs(db))
    assert int(logs["status"].max()) == 0
    print("TRAFFIC_RUN " + json.dumps({"n_logs": eng.n_logs(db), "rows": eng.n_ops(db), "n_changes": eng.n_changes(db), "calib_known_bytes": known,
                                       "launch": list(eng.launch_shape(db)), "V": int(logs["n_visible"].sum()), "S": int(logs["n_spans"].sum()), "T": int(logs["n_cintervals"].sum())}), flush=True)
    eng.free_result(dr)
    eng.free_batch(db)
    eng.close()


if __name__ == "__main__":
    main()
