#!/usr/bin/env python3
"""Makes tests/golden/rootmap_ref.json: the documents of helpers.root_map_docs() and what the REFERENCE itself (type-erased build
oracle/_ref, `--impl ref`) answers for them — getRoot() of every replica, or the RangeError it throws.  Needs node and
/root/reference (the build container); the GPU box only reads the committed fixture."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import helpers as H  # noqa: E402

if __name__ == "__main__":
    docs = H.root_map_docs()
    exp = H.oracle_apply(docs, impl="ref", roots=True)
    assert exp == H.oracle_apply(docs, impl="oracle", roots=True), "oracle and reference disagree"
    # change() calls with InputOperations on map objects: the Changes the reference returns, or its error
    cdocs, calls, actors = H.root_map_change_calls()
    made, l = [], 0
    for logs in cdocs:
        for log in logs:
            got = {}
            for impl in ("ref", "oracle"):
                try:
                    got[impl] = {"changes": H.oracle_change([[log]], [calls[l]], [actors[l]], impl=impl)}
                except RuntimeError as e:
                    got[impl] = {"error": str(e).replace("Symbol(_root)", "_root")}
            assert got["ref"] == got["oracle"], (l, got)
            made.append(got["ref"])
            l += 1
    with open(os.path.join(H.GOLDEN, "rootmap_ref.json"), "w") as f:
        json.dump({"impl": "ref", "docs": docs, "expected": exp, "change": {"docs": cdocs, "calls": calls, "actors": actors, "made": made}}, f, indent=1, sort_keys=True)
    print("wrote rootmap_ref.json:", sum(len(d) for d in exp), "replica logs,", len(made), "replicas with change() calls")
