#!/usr/bin/env python3
"""The extra legs of bench.py, in a process of their own (no torch; nothing here is the bench's `value`):
  baseline_configs  EVERY other BASELINE.json configuration on this GPU — #2 (1 024 docs x 256 ops as specified, and a GPU-filling multiple), #3 (8 192 docs x
                    1 024 ops, and a GPU-filling multiple), #5 (8 192-op logs with link / comment marks and 50 % deletes: the share of one of eight GPUs and
                    twice that) — each with the kernel's launch duration (HIP events on the engine's stream), the SURVEY 8(d) roofline fraction of that launch
                    (B_alg = sum of 32 N + 4 V + 8 S + 16 T + 16 over its logs), the launch shape, and --parity-docs documents of the resident batch
                    checked against the oracle on the host (whole logs: decoded spans, raw rows, digests);
  patch_replay      ptx_replay_patches on 2 048 documents of the bench's config: ops replayed per second (kernel time measured in the library);
  phase_cycles      where a log's residency goes (thread-0 cycle stamps of the diagnostic build of the same kernel body), loaded and with a CU to itself;
  pipeline          change() -> merge -> convergence as a two-stage pipeline: ptx_generate of batch k + 1 (one engine, its stream) runs while ptx_merge of batch k and
                    the digest comparison run on a second engine's stream — against the same batches one after the other on one engine;
  host_to_host      the boundary's one-shot call with HOST buffers: upload, merge, download of the compact result rows (tools/pcie_rate.py);
  many              documents of 5 and 8 actors under causal admission (ptx_merge_kernel_many) beside 3-actor documents prepared the same way;
  candidate         when a candidate build peritext_amd/lib/exp_<name>.so stands beside the product (__graft_entry__.CANDIDATE): the bench's workload under
                    it, same box, same call, every log's status / digest / row counts compared with the product's.
Every leg records its own failure instead of raising.  One JSON line on stdout.
    python tools/bench_extras.py --config config4 --docs 65536 [--iters 10]"""
import argparse
import glob
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from peritext_amd import abi, wire, workloads  # noqa: E402
from peritext_amd.engine import Engine  # noqa: E402

T0 = time.time()
HBM_PEAK = 8.0e12
PHASES = {0: "P0 admission", 1: "P1 row loop", 11: "P1 tail (census, dup, scan)", 2: "P3a index+parents", 12: "P3b checks+scatter", 3: "P3c child order", 4: "P3d after()",
          14: "P3d successor words", 16: "P3d jumping rounds", 17: "P3d positions", 18: "unpark marks", 5: "P4 tombstones", 6: "P5a values", 13: "P5a mark intervals",
          7: "P5c comments", 8: "P5b trees + P6 spans", 10: "end"}
ORDER = [0, 1, 11, 2, 12, 3, 4, 14, 16, 17, 18, 5, 6, 13, 7, 8]
# (config, documents): the specified size first, then sizes that fill the GPU
BASELINE_LEGS = [("config2", 1024), ("config2", 524288), ("config3", 8192), ("config3", 196608), ("config5", 32768), ("config5", 65536),
                 ("rich4k", 16384)]  # (round 6: not a BASELINE configuration — the `rich` mix, 55 / 10 / 20 / 15 with all four mark types, at 3 x 4 096 ops: documents that hold ~2 000 characters)


def say(msg):
    print("[extras %6.1fs] %s" % (time.time() - T0, msg), file=sys.stderr, flush=True)


def oracle_spans(docs_logs):
    """Expected {spans, text} of every replica log through oracle/peritext_oracle.js (whole logs, Patch[] bookkeeping off), one node process per 4 logs."""
    node = shutil.which("node")
    if node is None:
        raise RuntimeError("node (the oracle runtime) is not on this box")
    flat = [(d, r) for d in range(len(docs_logs)) for r in range(len(docs_logs[d]))]
    parts = [flat[p::8] for p in range(min(8, len(flat)))]
    td = tempfile.mkdtemp(prefix="ptxextra_")
    jobs = []
    for p, mine in enumerate(parts):
        inp, outp = os.path.join(td, "in%d.json" % p), os.path.join(td, "out%d.json" % p)
        with open(inp, "w") as f:
            json.dump({"docs": [{"logs": [docs_logs[d][r]]} for d, r in mine]}, f)
        jobs.append((subprocess.Popen([node, os.path.join(ROOT, "oracle", "cli.js"), "apply", "--impl", "oracle", "--no-patches", "--in", inp, "--out", outp], cwd=ROOT), outp))
    expected = [[None] * len(l) for l in docs_logs]
    for (pr, outp), mine in zip(jobs, parts):
        if pr.wait() != 0:
            raise RuntimeError("oracle run failed")
        with open(outp) as f:
            o = json.load(f)
        for (d, r), e in zip(mine, o["docs"]):
            expected[d][r] = e["expected"][0]
    shutil.rmtree(td, ignore_errors=True)
    return expected


def config_leg(args, name, docs, flags, replicas=None, parity_docs=None):
    """One BASELINE configuration resident on the GPU: kernel ms, SURVEY 8(d) fraction, launch shape, oracle parity of a few of its documents."""
    import helpers

    g = workloads.gen_config(name, replicas=replicas)
    gen_args = (g["replicas"], g["ops_per_log"], g["mix"], g["mark_types"])
    row = {"config": name, "docs": docs, "replicas": g["replicas"], "ops_per_log": g["ops_per_log"], "mix_ins_del_add_rem": g["mix"], "causal_admission": not args.no_admission}
    try:
        with Engine(args.device, flags=flags) as e:
            list_cap = max(args.list_cap, g["ops_per_log"] // 2 + 512, g["replicas"] * 640 if g["replicas"] > 4 else 0, g["ops_per_log"] * 3 // 4 if g["mix"][0] > 50 else 0)  # the element list of a document, held on chip while it is generated (it grows with the replicas: every one of them makes ops_per_log ops)
            db, info = e.generate(*gen_args, docs, args.seed, list_cap=list_cap)
            n_logs, rows = e.n_logs(db), e.n_ops(db)
            dr = e.alloc_result(db)
            e.merge(db, dr)
            e.sync()
            iters = max(args.iters, 3)
            ms = min(e.merge_timed(db, dr, iters) / iters for _ in range(2))
            logs = e.download_logs(dr, n_logs)
            ok = bool(int(logs["status"].max()) == 0)
            V, S, T = int(logs["n_visible"].sum()), int(logs["n_spans"].sum()), int(logs["n_cintervals"].sum())
            alg = 32 * rows + 4 * V + 8 * S + 16 * T + 16 * n_logs
            env = 0 if args.no_admission else abi.envelope_bytes(e.n_changes(db), g["replicas"])
            threads, lds = e.launch_shape(db)
            row.update({"replica_logs": n_logs, "ops": n_logs * g["ops_per_log"], "kernel": e.batch_kernel_name(db), "kernel_ms": ms, "ops_per_s": n_logs * g["ops_per_log"] / (ms * 1e-3), "every_log_ok": ok,
                        "algorithmic_bytes": alg, "roofline_GBps": alg / (ms * 1e-3) / 1e9, "roofline_frac": alg / (ms * 1e-3) / HBM_PEAK,
                        "with_envelope_frac": (alg + env) / (ms * 1e-3) / HBM_PEAK, "launch": {"threads_per_log": threads, "lds_bytes_per_log": lds,
                                                                                                 "logs_per_cu_by_lds": int((160 * 1024) // max(512, (lds + 511) // 512 * 512))},
                        "lds_high": int(logs["reserved"][:, 0].max()), "visible_chars_per_log": V / n_logs, "gen_kernel_ms": info["kernel_ms"]})
            # parity: documents of the RESIDENT batch (regenerated one by one with the same generator arguments, so the same documents) against the oracle
            n_parity = args.parity_docs if parity_docs is None else min(parity_docs, args.parity_docs)
            if n_parity > 0 and shutil.which("node"):
                rng = np.random.default_rng(args.seed + docs)
                # half drawn at random, half the documents that show the most text (a 50 %-deletes configuration leaves most documents nearly empty: random
                # draws alone would hardly exercise span and comment-interval rows)
                half = min(n_parity, docs) // 2
                vis_doc = logs["n_visible"].reshape(-1, g["replicas"])[:, 0]
                rich = [int(x) for x in np.argsort(-vis_doc.astype(np.int64), kind="stable")[:half]]
                rest = [int(x) for x in rng.permutation(docs) if int(x) not in set(rich)][: min(n_parity, docs) - len(rich)]
                pick = sorted(rich + rest)
                ones, docs_logs = [], []
                for d in pick:
                    hb, hinfo = e.generate(*gen_args, 1, args.seed, first_doc=d, list_cap=list_cap)
                    actors_t, comments_t, log_doc_t = wire.generated_tables(1, g["replicas"], hinfo["n_comments"])
                    one = e.download_batch(hb, wire.GEN_VALUES, wire.GEN_URLS, log_doc_t, actors_t, comments_t)
                    e.free_batch(hb)
                    ones.append(one)
                    docs_logs.append([wire.decode_changes(one, r) for r in range(g["replicas"])])
                expected = oracle_spans(docs_logs)
                for d, one, exp in zip(pick, ones, expected):
                    sub = e.download_range(db, dr, d * g["replicas"], g["replicas"])
                    for r in range(g["replicas"]):
                        helpers.check_log(one, sub, r, exp[r])
                row["parity"] = {"documents_checked": len(pick), "visible_chars_checked": int(sum(int(vis_doc[d]) for d in pick)), "against": "oracle/peritext_oracle.js (whole logs): decoded spans, raw rows, digests of the resident batch's result rows"}
            e.free_result(dr)
            e.free_batch(db)
    except Exception as ex:  # noqa: BLE001
        row["error"] = str(ex)[:300]
    return row


def typed_essay(n_ops, seed, actor="essay"):
    """A long single-author document made on the host: mostly typing at the cursor, some jumps, a fifth deletes — Changes of 1..12 ops (no oracle here:
    the parity of such logs is the test-suite's, tests/test_gpu_biglog.py; this leg only times them)."""
    import random

    rng = random.Random(seed)
    ctr, seq = 1, 1
    log = [{"actor": actor, "seq": seq, "deps": {}, "startOp": 1, "ops": [{"opId": "1@%s" % actor, "action": "makeList", "obj": "_root", "key": "text"}]}]
    alive, cursor, made = [], "_head", 1
    while made < n_ops:
        ops = []
        for _ in range(min(rng.randint(1, 12), n_ops - made)):
            ctr += 1
            oid = "%d@%s" % (ctr, actor)
            if alive and rng.random() < 0.2:
                k = rng.randrange(len(alive))
                alive[k], alive[-1] = alive[-1], alive[k]
                ops.append({"opId": oid, "action": "del", "obj": "1@%s" % actor, "elemId": alive.pop()})
            else:
                if alive and rng.random() < 0.1:
                    cursor = alive[rng.randrange(len(alive))]
                ops.append({"opId": oid, "action": "set", "obj": "1@%s" % actor, "elemId": cursor, "insert": True, "value": chr(97 + ctr % 26)})
                alive.append(oid)
                cursor = oid
            made += 1
        seq += 1
        log.append({"actor": actor, "seq": seq, "deps": {}, "startOp": ctr - len(ops) + 1, "ops": ops})
    return log


def biglog_leg(args):
    """Documents beyond one CU's LDS (biglog_core.h, the HBM-staged kernel inside the same ptx_merge): a 100 000-op essay and a 40 001-row all-marks log,
    each alone in a batch and both together — ms per merge (HIP events around ptx_merge), ops/s."""
    import helpers

    from peritext_amd import wire

    rows = []
    essay = typed_essay(100000, 7)
    marks = helpers.synthetic_marks_log(6000, 34000, 9)
    for name, docs in (("essay_100k", [[essay]]), ("marks_40k", [[marks]]), ("both", [[essay], [marks]])):
        batch = wire.encode_docs(docs)
        with Engine(args.device, flags=abi.FLAG_NO_ELEM_RANK) as e:
            db = e.upload(batch)
            dr = e.alloc_result(db)
            e.merge(db, dr)
            e.sync()
            ms = min(e.merge_timed(db, dr, 3) / 3 for _ in range(2))
            logs = e.download_logs(dr, batch.n_logs)
            rows.append({"batch": name, "rows": int(batch.n_ops), "elements": [int(x) for x in logs["n_elems"]], "visible": [int(x) for x in logs["n_visible"]],
                         "every_log_ok": bool(int(logs["status"].max()) == 0), "hbm_staged": bool((logs["reserved"][:, 0] == 0).all()), "ms": ms,
                         "ops_per_s": batch.n_ops / ms * 1e3})
            e.free_result(dr)
            e.free_batch(db)
    return {"kernel": "ptx_merge_big_grid_kernel: a cooperative launch of up to 64 workgroups per log of 16 384 rows or more (working set in HBM scratch; smaller ones: ptx_merge_big_kernel, one 1 024-thread workgroup each)", "legs": rows}


def many_actor_leg(args, target_logs=48960):
    """Documents with MORE than three actors under causal admission: config-4-shaped documents of 5 and 8 replicas beside 3-replica ones,
    all made on the device (round 5: the generator holds up to 8 replicas) — the same number of replica logs per leg, every document distinct.  Per leg: the
    kernel build the library chose (ptx_merge_kernel_many: the one-pass admission walk up to seven actors; _many_wide: eight to fifteen), its ms, the SURVEY 8(d)
    fraction, the per-log cost against three replicas, and --parity-docs documents of the resident batch against the oracle."""
    rows = []
    for R in (3, 5, 8):
        row = config_leg(args, args.config, target_logs // R, abi.FLAG_NO_ELEM_RANK, replicas=R)
        if "kernel_ms" in row:
            row["us_per_log_per_cu"] = row["kernel_ms"] * 1e3 * 256 / row["replica_logs"]
        rows.append(row)
        say("many_actor %s" % json.dumps(row))
    base = rows[0].get("us_per_log_per_cu")
    for r in rows:
        if base and "us_per_log_per_cu" in r:
            r["per_log_cost_vs_3_replicas"] = r["us_per_log_per_cu"] / base
    return {"workload": "%s documents generated on the device, %d replica logs per leg" % (args.config, target_logs), "legs": rows}


def pipeline_leg(args, gen_args, flags, batches=6, docs=8192):
    """generate -> merge -> converged count over `batches` batches of `docs` documents: serial on one engine, then generator and merger on two engines (two HIP
    streams of one device) in two host threads (the ctypes calls release the GIL).  Handles made by one engine are merged by the other: a resident batch is plain
    device memory.  The count is taken on the host from the 16-byte digests of the result rows (no torch in this process)."""
    import queue
    import threading

    replicas = gen_args[0]

    def converged(e, db, dr):
        dg = e.download_logs(dr, e.n_logs(db))["digest"].reshape(-1, replicas, 2)
        return int((dg == dg[:, :1, :]).all(axis=(1, 2)).sum())

    def consume(e, db):
        dr = e.alloc_result(db)
        e.merge(db, dr)
        e.sync()
        c = converged(e, db, dr)
        e.free_result(dr)
        return c

    ops = batches * docs * replicas * gen_args[1]
    with Engine(args.device, flags=flags) as g, Engine(args.device, flags=flags) as m:
        db, _ = g.generate(*gen_args, 256, args.seed, list_cap=args.list_cap)  # warm both engines
        consume(m, db)
        g.free_batch(db)
        t0 = time.time()
        conv_serial = 0
        for k in range(batches):
            db, _ = g.generate(*gen_args, docs, args.seed, first_doc=args.first_doc + k * docs, list_cap=args.list_cap)
            conv_serial += consume(g, db)
            g.free_batch(db)
        serial_s = time.time() - t0
        ready, done, errs = queue.Queue(maxsize=2), queue.Queue(), []

        def produce():
            try:
                for k in range(batches):
                    while not done.empty():
                        g.free_batch(done.get())
                    db, _ = g.generate(*gen_args, docs, args.seed, first_doc=args.first_doc + k * docs, list_cap=args.list_cap)
                    ready.put(db)
            except Exception as ex:  # noqa: BLE001
                errs.append(str(ex)[:300])
            ready.put(None)

        t0 = time.time()
        th = threading.Thread(target=produce)
        th.start()
        conv_pipe = 0
        while True:
            db = ready.get()
            if db is None:
                break
            conv_pipe += consume(m, db)
            done.put(db)
        th.join()
        pipe_s = time.time() - t0
        while not done.empty():
            g.free_batch(done.get())
    if errs:
        raise RuntimeError(errs[0])
    return {"batches": batches, "docs_per_batch": docs, "ops": ops, "serial_s": serial_s, "pipelined_s": pipe_s, "serial_ops_per_s": ops / serial_s, "pipelined_ops_per_s": ops / pipe_s,
            "docs_converged": conv_pipe, "same_count_serial": conv_pipe == conv_serial, "stages": "ptx_generate (engine A) || ptx_merge + digest comparison (engine B)"}


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
    ap.add_argument("--parity-docs", type=int, default=32)
    ap.add_argument("--legs", default="configs,replay,biglog,pipeline,host,many,phases,candidate,probe,jshost")
    args = ap.parse_args()
    legs = set(args.legs.split(","))
    g = workloads.gen_config(args.config, ops=args.ops)
    gen_args = (g["replicas"], g["ops_per_log"], g["mix"], g["mark_types"])
    flags = abi.FLAG_NO_ELEM_RANK | (abi.FLAG_NO_ADMISSION if args.no_admission else 0)
    out = {}

    if "configs" in legs:
        out["baseline_configs"] = []
        for name, docs in BASELINE_LEGS:
            row = config_leg(args, name, docs, flags, parity_docs=8 if name == "rich4k" else None)  # (a 4 096-op log that keeps its text is minutes of oracle time)
            out["baseline_configs"].append(row)
            say("baseline_configs %s" % json.dumps(row))

    if "replay" in legs:
        try:
            with Engine(args.device) as e:  # with elem_rank: the replay reads it
                docs = min(4096, args.docs)  # (12 288 logs: three full rounds of workgroups at 16 resident logs per CU)
                db, _ = e.generate(*gen_args, docs, args.seed, first_doc=args.first_doc, list_cap=args.list_cap)
                dr = e.alloc_result(db)
                e.merge(db, dr)
                e.sync()
                pat = e.replay_patches(db, dr)
                ops = e.n_logs(db) * g["ops_per_log"]
                n_pat = int(pat.logs["n_patches"].sum())
                out["patch_replay"] = {"docs": docs, "ops": ops, "patches": n_pat, "kernel_ms": pat.kernel_ms, "launches": pat.launches,
                                       "every_log_has_a_stream": bool(int(pat.logs["status"].max()) == 0), "ops_per_s": ops / pat.kernel_ms * 1e3,
                                       "patches_per_s": n_pat / pat.kernel_ms * 1e3}
                e.free_result(dr)
                e.free_batch(db)
        except Exception as ex:  # noqa: BLE001
            out["patch_replay"] = {"error": str(ex)[:300]}
        say("patch_replay %s" % json.dumps(out["patch_replay"]))

    if "jshost" in legs:
        # the TypeScript-facing entry point end to end — Change[] JSON (reference/src/micromerge.ts:60-71) -> MergeEngine.applyChanges -> spans,
        # one Node thread, on documents the device generator made (downloaded and decoded to the reference's Change objects here); and one resident edit session
        try:
            node = shutil.which("node")
            if node is None:
                raise RuntimeError("node is not on this box")
            jdocs = 24
            with Engine(args.device, flags=flags) as e:
                db, ginfo = e.generate(*gen_args, jdocs, args.seed, list_cap=args.list_cap)
                actors, comments, log_doc = wire.generated_tables(jdocs, g["replicas"], ginfo["n_comments"])
                hb = e.download_batch(db, wire.GEN_VALUES, wire.GEN_URLS, log_doc, actors, comments)
                e.free_batch(db)
            logs = [wire.decode_changes(hb, l) for l in range(hb.n_logs)]
            docs_json = [logs[d * g["replicas"]:(d + 1) * g["replicas"]] for d in range(jdocs)]
            with tempfile.NamedTemporaryFile("w", suffix=".json", delete=False) as f:
                json.dump(docs_json, f)
                path = f.name
            try:
                p = subprocess.run([node, os.path.join(ROOT, "tools", "js_host_bench.js"), path, "3"], capture_output=True, text=True, timeout=600)
                if p.returncode != 0:
                    raise RuntimeError(p.stderr[-300:])
                row = json.loads(p.stdout.strip().splitlines()[-1])
            finally:
                os.unlink(path)
            row["config"] = args.config
            p = subprocess.run([node, os.path.join(ROOT, "tests", "node_host_check.js"), "resident-edit", "200"], capture_output=True, text=True, timeout=600)
            if p.returncode == 0:
                ed = json.loads(p.stdout.strip().splitlines()[-1])
                row["resident_edit_session"] = {"edits": ed["edits"], "ms_per_change": ed["msPerResidentChange"], "whole_document_uploads_after_setup": ed["wholeDocumentUploadsAfterSetup"]}
            out["js_host_end_to_end"] = row
        except Exception as ex:  # noqa: BLE001
            out["js_host_end_to_end"] = {"error": str(ex)[:300]}
        say("js_host_end_to_end %s" % json.dumps(out["js_host_end_to_end"]))

    if "biglog" in legs:
        try:
            out["biglog"] = biglog_leg(args)
        except Exception as ex:  # noqa: BLE001
            out["biglog"] = {"error": str(ex)[:300]}
        say("biglog %s" % json.dumps(out["biglog"]))

    if "pipeline" in legs:
        try:
            out["generate_merge_pipeline"] = pipeline_leg(args, gen_args, flags)
        except Exception as ex:  # noqa: BLE001
            out["generate_merge_pipeline"] = {"error": str(ex)[:300]}
        say("generate_merge_pipeline %s" % json.dumps(out["generate_merge_pipeline"]))

    if "host" in legs:
        # the boundary's one-shot call with HOST buffers (ptx_batch_upload -> ptx_merge -> ptx_result_download: ptx_apply_materialize), 8 192 documents: never `value`
        try:
            import pcie_rate

            m = pcie_rate.measure(min(8192, args.docs), args.config, reps=3)
            out["host_to_host"] = {k: m[k] for k in ("config", "docs", "replica_logs", "ops", "host_input_bytes", "abi", "best")}
        except Exception as ex:  # noqa: BLE001
            out["host_to_host"] = {"error": str(ex)[:300]}
        say("host_to_host %s" % json.dumps(out["host_to_host"]))

    if "many" in legs:
        try:
            out["many_actor_documents"] = many_actor_leg(args)
        except Exception as ex:  # noqa: BLE001
            out["many_actor_documents"] = {"error": str(ex)[:300]}
        say("many_actor_documents %s" % json.dumps(out["many_actor_documents"]))

    ref = None
    if "phases" in legs or "candidate" in legs:
        variants = [("product build", None)]
        if "candidate" in legs:
            variants += [(os.path.basename(p)[:-3], p) for p in sorted(glob.glob(os.path.join(ROOT, "peritext_amd", "lib", "exp_*.so"))) if not p.endswith(("exp_diag.so", "exp_base.so"))]
        rows = []
        for name, lib in variants:
            row = {"name": name}
            try:
                with Engine(args.device, flags=flags, lib_path=lib) as e:
                    db, _ = e.generate(*gen_args, args.docs, args.seed, first_doc=args.first_doc, list_cap=args.list_cap)
                    dr = e.alloc_result(db)
                    e.merge(db, dr)
                    e.sync()
                    row["kernel_ms"] = e.merge_timed(db, dr, args.iters) / args.iters
                    row["launch"] = list(e.launch_shape(db))
                    if ref is None and "phases" in legs:
                        cyc = e.phase_cycles(db, dr)
                        row["phase_cycles_per_log"] = {PHASES.get(k, str(k)): round(cyc[k] / e.n_logs(db)) for k in ORDER if k < len(cyc) and cyc[k]}
                        e.merge(db, dr)  # the diagnostic launch wrote the same rows; run the product kernel once more before they are read
                        e.sync()
                    lo = e.download_logs(dr, e.n_logs(db))
                    if ref is None:
                        ref = lo
                        row["every_log_ok"] = bool(int(lo["status"].max()) == 0)
                    else:
                        row["identical_results"] = bool((lo["status"] == ref["status"]).all() and (lo["digest"] == ref["digest"]).all() and (lo["n_spans"] == ref["n_spans"]).all()
                                                        and (lo["n_visible"] == ref["n_visible"]).all())
                    e.free_result(dr)
                    e.free_batch(db)
            except Exception as ex:  # noqa: BLE001
                row["error"] = str(ex)[:300]
            rows.append(row)
            say(json.dumps(row))
        out["same_workload"] = {"workload": "%s, %d docs (the bench's own documents)" % (args.config, args.docs), "launches_each": args.iters, "same_box_same_call": True, "builds": rows}

    if "probe" in legs:
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
                    row = {"logs": e.n_logs(db), "kernel_ms": ms, "us_per_log_per_cu": ms * 1e3 * 256 / e.n_logs(db)}
                    if docs == 85:
                        cyc = e.phase_cycles(db, dr)
                        row["phase_cycles_per_log_alone"] = {PHASES.get(k, str(k)): round(cyc[k] / e.n_logs(db)) for k in ORDER if k < len(cyc) and cyc[k]}
                    probes.append(row)
                    e.free_result(dr)
                    e.free_batch(db)
        except Exception as ex:  # noqa: BLE001
            probes.append({"error": str(ex)[:300]})
        out["occupancy_probe"] = probes
        say("occupancy_probe %s" % json.dumps(probes))
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
