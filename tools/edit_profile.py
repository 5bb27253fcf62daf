#!/usr/bin/env python3
"""Where the time of ONE resident edit goes (GPU box): a two-replica document of --ops ops stays in HBM; every edit is what the JS host's replica().change() does through
the addon — result_alloc, merge, sync, ptx_change, download of the made batch, append on the device, frees — each stage timed on the host (the calls are synchronous
or followed by a sync here).   python tools/edit_profile.py --ops 4096 --edits 200"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers as H  # noqa: E402
from peritext_amd import wire  # noqa: E402
from peritext_amd.engine import Engine  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ops", type=int, default=4096)
    ap.add_argument("--edits", type=int, default=200)
    ap.add_argument("--lib", default=None)
    args = ap.parse_args()
    gen = H.oracle_gen("rich", 1, 5, args.ops, 2)
    docs = [d["logs"] for d in gen["docs"]]
    batch = wire.encode_docs(docs, extra_comments=[[]])
    actors = [docs[0][0][0]["actor"]] * 0
    t = {k: 0.0 for k in ("result_alloc", "merge+sync", "change", "download_made", "append_device", "frees")}
    with Engine(0, lib_path=args.lib) as e:
        db = e.upload(batch)
        names = batch.doc_actors[0]
        for i in range(args.edits + 20):
            if i == 20:
                t = {k: 0.0 for k in t}
            c0 = time.perf_counter()
            dr = e.alloc_result(db)
            c1 = time.perf_counter()
            e.merge(db, dr)
            e.sync()
            c2 = time.perf_counter()
            calls = [[[{"path": ["text"], "action": "insert", "index": 3 + (i % 7), "values": ["k"]}]], []]
            ops = wire.encode_input_ops(batch, calls, [names[0], names[1]])
            c2b = time.perf_counter()
            made, st = e.change(db, dr, ops)
            assert not st.any()
            c3 = time.perf_counter()
            e.download_batch(made, batch.values, batch.urls, batch.log_doc, batch.doc_actors, batch.doc_comments, batch.keys, batch.map_values)
            c4 = time.perf_counter()
            after = e.append_device(db, made)
            c5 = time.perf_counter()
            e.free_result(dr)
            e.free_batch(made)
            e.free_batch(db)
            db = after
            c6 = time.perf_counter()
            t["result_alloc"] += c1 - c0
            t["merge+sync"] += c2 - c1
            t["change"] += c3 - c2b
            t["download_made"] += c4 - c3
            t["append_device"] += c5 - c4
            t["frees"] += c6 - c5
        out = {k: round(1e3 * v / args.edits, 4) for k, v in t.items()}
        out["total_ms_per_edit"] = round(sum(out.values()), 4)
        out["ops"] = args.ops
        print(json.dumps(out))


if __name__ == "__main__":
    main()
