#!/usr/bin/env python3
"""Fixture maker (test infrastructure): expected outputs of the hand-written edge-case logs of tests/helpers.py, produced by the
reference itself (oracle/_ref, types erased by oracle/build_ref.js) in THIS container and committed, so that the GPU twins of the
emulation tests (tests/test_gpu_edges.py) need neither node nor /root/reference on the GPU box.

    python tests/make_edge_golden.py          # writes tests/golden/edge_cases_ref.json
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import helpers as H  # noqa: E402


def main():
    with open(os.path.join(H.GOLDEN, "ptxgen_mini.json")) as f:
        mini = json.load(f)
    cursor_docs = [d["logs"] for d in mini["docs"][:3]]
    out = {
        "impl": "ref",
        "edge": H.oracle_apply(H.edge_case_docs(), impl="ref"),
        "huge_bucket": H.oracle_apply([[H.huge_bucket_log()]], impl="ref"),
        "boundary": H.oracle_apply(H.boundary_docs(), impl="ref", patches=True),
        "unsynced": H.oracle_apply(H.unsynced_docs(), impl="ref", patches=True),
        "cursors": [[{"text": e["text"], "cursorAt": e["cursorAt"], "cursorResolve": e["cursorResolve"]} for e in d]
                    for d in H.oracle_apply(cursor_docs, impl="ref", cursors=True)],
    }
    with open(os.path.join(H.GOLDEN, "edge_cases_ref.json"), "w") as f:
        json.dump(out, f)
    print("wrote", os.path.join(H.GOLDEN, "edge_cases_ref.json"))


if __name__ == "__main__":
    main()
