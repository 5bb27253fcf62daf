#!/usr/bin/env python3
"""The extra legs of bench.py, in a process of their own (no torch; nothing here is the bench's `value`):
  patch_replay  ptx_replay_patches on 2 048 documents of the bench's config: ops replayed per second (kernel time measured inside the library);
  experiments   the bench's workload (same generator arguments, so the same documents) under the product build, then under every experimental build
                peritext_amd/lib/exp_*.so (__graft_entry__.EXPERIMENTS) and under other launch shapes of the product build (PTX_THREADS): kernel ms per
                launch (HIP events on the engine's stream), and whether statuses / digests / row counts of EVERY log equal the product build's.
Every leg records its own failure instead of raising.  One JSON line on stdout.
    python tools/bench_extras.py --config config4 --docs 65536 [--iters 10]"""
import argparse
import glob
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from peritext_amd import abi, workloads  # noqa: E402
from peritext_amd.engine import Engine  # noqa: E402

T0 = time.time()


def say(msg):
    print("[extras %6.1fs] %s" % (time.time() - T0, msg), file=sys.stderr, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="config4")
    ap.add_argument("--docs", type=int, default=65536)
    ap.add_argument("--first-doc", type=int, default=0)
    ap.add_argument("--ops", type=int, default=None)
    ap.add_argument("--seed", type=int, default=2024)
    ap.add_argument("--list-cap", type=int, default=2048)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--no-admission", action="store_true")
    ap.add_argument("--threads", default="128,256", help="launch shapes of the product build to time beside its own choice")
    args = ap.parse_args()
    g = workloads.gen_config(args.config, ops=args.ops)
    gen_args = (g["replicas"], g["ops_per_log"], g["mix"], g["mark_types"])
    out = {"patch_replay": None, "experiments": None}

    def stream_checksum(pat):
        """Order-sensitive checksum over the records every log really produced (the rows past a log's count are capacity, not data)."""
        import numpy as np

        caps = np.diff(pat.patch_off.astype(np.int64))
        n = pat.logs["n_patches"].astype(np.int64)
        valid = (np.arange(int(caps.sum()), dtype=np.int64) - np.repeat(pat.patch_off[:-1].astype(np.int64), caps)) < np.repeat(n, caps)
        rows = pat.patches[: len(valid)][valid]
        w = np.arange(1, len(rows) + 1, dtype=np.uint64)
        mix = (rows["row"].astype(np.uint64) * np.uint64(0x9E3779B1) + rows["kind"].astype(np.uint64) * np.uint64(0x85EBCA77) + rows["a"].astype(np.uint64) * np.uint64(0xC2B2AE3D)
               + rows["b"].astype(np.uint64) * np.uint64(0x27D4EB2F))
        return int(n.sum()), int((mix * w).sum() & np.uint64(0xFFFFFFFFFFFFFFFF))

    def replay_rate(lib):
        try:
            with Engine(args.device, lib_path=lib) as e:  # with elem_rank: the replay reads it
                docs = min(2048, args.docs)
                db, _ = e.generate(*gen_args, docs, args.seed, first_doc=args.first_doc, list_cap=args.list_cap)
                dr = e.alloc_result(db)
                e.merge(db, dr)
                e.sync()
                pat = e.replay_patches(db, dr)
                ops = e.n_logs(db) * g["ops_per_log"]
                n_pat = int(pat.logs["n_patches"].sum())
                r = {"build": os.path.basename(lib or "libperitext_hip.so"), "docs": docs, "ops": ops, "patches": n_pat, "kernel_ms": pat.kernel_ms, "launches": pat.launches,
                     "every_log_has_a_stream": bool(int(pat.logs["status"].max()) == 0), "ops_per_s": ops / pat.kernel_ms * 1e3, "patches_per_s": n_pat / pat.kernel_ms * 1e3,
                     "_sum": stream_checksum(pat)}
                e.free_result(dr)
                e.free_batch(db)
                return r
        except Exception as ex:  # noqa: BLE001
            return {"build": os.path.basename(lib or "libperitext_hip.so"), "error": str(ex)[:300]}

    out["patch_replay"] = replay_rate(None)
    say("patch_replay %s" % json.dumps({k: v for k, v in out["patch_replay"].items() if k != "_sum"}))
    # the same under every experimental build (exp_nopark carries the replay as it was measured before, PTX_REPLAY_V1; exp_replay_* other searches)
    others = []
    for lib in sorted(glob.glob(os.path.join(ROOT, "peritext_amd", "lib", "exp_*.so"))):
        other = replay_rate(lib)
        if "_sum" in other and "_sum" in out["patch_replay"]:
            other["same_streams_by_checksum"] = other["_sum"] == out["patch_replay"]["_sum"]
        other.pop("_sum", None)
        others.append(other)
        say("patch_replay %s" % json.dumps(other))
    out["patch_replay"].pop("_sum", None)
    out["patch_replay"]["other_builds"] = others

    flags = abi.FLAG_NO_ELEM_RANK | (abi.FLAG_NO_ADMISSION if args.no_admission else 0)
    variants = [("product build", None, 0)]
    variants += [(os.path.basename(p)[:-3], p, 0) for p in sorted(glob.glob(os.path.join(ROOT, "peritext_amd", "lib", "exp_*.so")))]
    variants += [("product build, %d threads per log" % int(t), None, int(t)) for t in args.threads.split(",") if t]
    rows, ref = [], None
    for name, lib, threads in variants:
        row = {"name": name}
        try:
            e = Engine(args.device, flags=flags, lib_path=lib)
            if threads:
                e.set_launch_shape(threads, 0)
            db, _ = e.generate(*gen_args, args.docs, args.seed, first_doc=args.first_doc, list_cap=args.list_cap)
            dr = e.alloc_result(db)
            e.merge(db, dr)
            e.sync()
            row["kernel_ms"] = e.merge_timed(db, dr, args.iters) / args.iters
            row["launch"] = list(e.launch_shape(db))
            if ref is None:  # the product build: where a log's residency goes (thread-0 cycle stamps of the diagnostic build of the same kernel body)
                try:
                    cyc = e.phase_cycles(db, dr)
                    names = ["P0+P1 admission+rows", "P2", "P3a+P3b buckets", "P3c child order", "P3d tour+rank", "P4 tombstones", "P5a values+intervals", "P5c comments",
                             "P5b+P6 LWW trees+spans", "P6 tail"]
                    row["phase_cycles_per_log"] = {(names[k] if k < len(names) else str(k)): round(cyc[k] / e.n_logs(db)) for k in range(len(cyc)) if cyc[k]}
                    e.merge(db, dr)  # the diagnostic launch wrote the same rows; run the product kernel once more before they are read
                    e.sync()
                except Exception as ex:  # noqa: BLE001
                    row["phase_cycles_per_log"] = {"error": str(ex)[:200]}
            lo = e.download_logs(dr, e.n_logs(db))
            if ref is None:
                ref = lo
                row["every_log_ok"] = bool(int(lo["status"].max()) == 0)
            else:
                row["identical_results"] = bool((lo["status"] == ref["status"]).all() and (lo["digest"] == ref["digest"]).all() and (lo["n_spans"] == ref["n_spans"]).all()
                                                and (lo["n_visible"] == ref["n_visible"]).all())
            e.free_result(dr)
            e.free_batch(db)
            e.close()
        except Exception as ex:  # noqa: BLE001
            row["error"] = str(ex)[:300]
        rows.append(row)
        say(json.dumps(row))
        if ref is None:
            break  # nothing to compare the variants with
    # how long ONE log takes when it has a CU to itself, and one full set of resident logs per CU (chain latency against contention)
    probes = []
    try:
        with Engine(args.device, flags=flags) as e:
            for docs in (85, 683, 5461):  # x 3 replicas = 255 / 2 049 / 16 383 logs: ~1, ~8, ~64 per CU
                if docs > args.docs:
                    continue
                db, _ = e.generate(*gen_args, docs, args.seed, first_doc=args.first_doc, list_cap=args.list_cap)
                dr = e.alloc_result(db)
                e.merge(db, dr)
                e.sync()
                ms = e.merge_timed(db, dr, args.iters) / args.iters
                probes.append({"logs": e.n_logs(db), "kernel_ms": ms, "us_per_log_per_cu": ms * 1e3 * 256 / e.n_logs(db)})
                e.free_result(dr)
                e.free_batch(db)
    except Exception as ex:  # noqa: BLE001
        probes.append({"error": str(ex)[:300]})
    out["occupancy_probe"] = probes
    say("occupancy_probe %s" % json.dumps(probes))
    out["experiments"] = {"workload": "%s, %d docs (the bench's own documents)" % (args.config, args.docs), "launches_each": args.iters, "same_box_same_call": True, "variants": rows}
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
