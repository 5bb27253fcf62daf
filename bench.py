#!/usr/bin/env python3
"""bench.py — whole-node throughput of the Peritext hot path on MI355X (BASELINE.json metric).

A "step" = one pass of the hot path over one resident batch: apply every replica op log of the batch (causal admission of
every Change included, micromerge.ts:499-511) and materialise its formatted document (ptx_merge = ONE launch of
ptx_merge_kernel), then count the converged documents on the device (N>1: after the all-gather of the per-replica digests
over RCCL/xGMI — the only collective on the path).  The op columns are resident in HBM before the timed region starts.

Workload (config.workload): BASELINE config #4 — 65 536 docs x 3 replicas x 4 096 ops, PTXGEN documents (the
workload of reference/test/fuzz.ts, seeded) GENERATED ON THE DEVICE (ptx_generate: on-device change(), every document
distinct).  `--gpus N` shards the 65 536 documents over N ranks in contiguous blocks ("strong" scaling: the total work is
fixed, rank r owns documents [r*65536/N, (r+1)*65536/N)); at N=1 the whole 64K-doc batch (25.8 GB of op log) is on one GPU.
`--docs-per-gpu D` fixes the per-rank share instead (the 8 192-doc shard of round 1 = --docs-per-gpu 8192).

One JSON line on stdout (rank 0):
  roofline.achieved = SURVEY.md §8(d) algorithmic bytes of one launch, B_alg = sum over logs of 32*N + 4*V + 8*S + 16*T + 16
      (N rows of the log, V visible values, S span rows, T = comment-interval rows — what this ABI writes in place of a
      mark-state table — and the 128-bit digest), divided by the kernel's average launch duration measured with HIP events on
      the stream the kernel runs on.  The Change envelope that causal admission reads on top is NOT in B_alg; the figure
      that includes it is reported separately (roofline.with_envelope).
  roofline.traffic  = HBM bytes per launch from the PMC counters of the same command (profiles/r06_hbm_traffic.json, made by
      tools/pmc_traffic.sh on the GPU box: separate --pmc passes; reads = the L2's read requests by size, cross-checked against
      FETCH_SIZE calibrated as MI355X_MICROARCH.md prescribes; writes = WRITE_SIZE); null when that file does not describe this workload.
  parity            = --check-docs random documents of the RESIDENT batch checked against the oracle run on the host cores on
      WHOLE logs (decoded spans, raw rows, digests of the rows the timed launches wrote).
  cpu_baseline      = the reference's own code (oracle/_ref) on the same sampled logs, one process per core, time-boxed.
"""
import argparse
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK = 8.0e12  # B/s, MI355X spec (MI355X_MICROARCH.md; 6.29e12 measured copy ceiling)
HBM_COPY_CEILING = 6.29e12
OPS = {"config4": 4096, "config3": 1024, "config2": 256, "config5": 8192, "rich": 1024, "mini": 96}


def log(msg):
    print("[bench %7.1fs] %s" % (time.time() - T0, msg), file=sys.stderr, flush=True)


T0 = time.time()


def _node_jobs(parts, extra, inputs):
    """One node process per part: oracle/cli.js <extra...> --in part.json; returns the parsed outputs in part order."""
    node = shutil.which("node")
    if node is None:
        raise RuntimeError("bench.py needs node (the oracle runtime) on this box for the parity guard and the CPU baseline")
    td = tempfile.mkdtemp(prefix="ptxref_")
    jobs = []
    for p, mine in enumerate(parts):
        inp, out = os.path.join(td, "in%d.json" % p), os.path.join(td, "out%d.json" % p)
        with open(inp, "w") as f:
            json.dump({"docs": [{"logs": [inputs[d][r]]} for d, r in mine]}, f)
        cmd = [node, os.path.join(ROOT, "oracle", "cli.js")] + extra + ["--in", inp]
        if extra[0] == "apply":
            jobs.append((subprocess.Popen(cmd + ["--out", out], cwd=ROOT), out))
        else:
            jobs.append((subprocess.Popen(cmd, cwd=ROOT, stdout=subprocess.PIPE, text=True), None))
    outs = []
    for pr, out in jobs:
        if out is not None:
            if pr.wait() != 0:
                raise RuntimeError("oracle run failed")
            with open(out) as f:
                outs.append(json.load(f))
        else:
            o, _ = pr.communicate()
            outs.append(json.loads(o.strip().splitlines()[-1]))
    shutil.rmtree(td, ignore_errors=True)
    return outs


def oracle_expected(docs_logs, procs):
    """Expected {spans, text} of every replica log ([doc][replica] -> Change[]): whole logs through the oracle (oracle/
    peritext_oracle.js, the restatement the test-suite pins against the reference; its Patch[] bookkeeping — a pure speed
    switch — is off: the reference itself needs minutes per 4 096-op log), one node process per core."""
    flat = [(d, r) for d in range(len(docs_logs)) for r in range(len(docs_logs[d]))]
    procs = max(1, min(procs, len(flat)))
    parts = [flat[p::procs] for p in range(procs)]
    outs = _node_jobs(parts, ["apply", "--impl", "oracle", "--no-patches"], docs_logs)
    expected = [[None] * len(logs) for logs in docs_logs]
    for mine, o in zip(parts, outs):
        for (d, r), e in zip(mine, o["docs"]):
            expected[d][r] = e["expected"][0]
    return expected


def cpu_truncated_leg(docs_logs, budget_s, procs):
    """The reference's own code (oracle/_ref, types erased; the restated oracle where that is absent) on the host cores, one node process per core, each on
    its own log of the sampled documents and each STOPPING after `budget_s`: the per-op cost grows along a log, so this leg OVERSTATES the CPU rate; it is
    kept beside the whole-log figure (cpu_baseline.value) as `deadline_truncated`."""
    impl = "ref" if os.path.exists(os.path.join(ROOT, "oracle", "_ref", "micromerge.js")) else "oracle"
    flat = [(d, r) for d in range(len(docs_logs)) for r in range(len(docs_logs[d]))]
    procs = max(1, min(procs, len(flat)))
    parts = [flat[p::procs] for p in range(procs)]
    t0 = time.time()
    rows = _node_jobs(parts, ["time", "--impl", impl, "--budget-ms", str(int(budget_s * 1000))], docs_logs)
    wall = time.time() - t0
    per_core = [r["ops_per_s"] for r in rows if r["seconds"] > 0]
    return {"value": float(sum(per_core)), "unit": "ops/s", "cores": len(rows), "per_core_ops_per_s": float(np.mean(per_core)) if per_core else 0.0,
            "sample": "%d whole + %d deadline-truncated replica logs (%d ops), %.0f s budget each, %.1f s wall" % (
                sum(r["logs"] for r in rows), sum(r.get("truncated_logs", 0) for r in rows), sum(r["ops"] for r in rows), budget_s, wall)}


class WholeLogBaseline:
    """cpu_baseline proper: WHOLE replica logs through the reference's own code (oracle/_ref; kind "reference") — applyChange over every
    change of the log + getTextWithFormatting — one node process per log, one log per core, started early and collected at the end of the bench (a whole
    4 096-op log takes the reference minutes: its per-op cost grows along the log).  A log that has not finished by the deadline counts with the ops it got
    through (and is reported as cut)."""

    def __init__(self, docs_logs, max_procs, timeout_s):
        self.impl = "ref" if os.path.exists(os.path.join(ROOT, "oracle", "_ref", "micromerge.js")) else "oracle"
        self.node = shutil.which("node")
        self.td = tempfile.mkdtemp(prefix="ptxwhole_")
        self.timeout_s = timeout_s
        self.t0 = time.time()
        self.jobs = []
        flat = [(d, r) for d in range(len(docs_logs)) for r in range(len(docs_logs[d]))][:max_procs]
        for p, (d, r) in enumerate(flat):
            inp = os.path.join(self.td, "in%d.json" % p)
            with open(inp, "w") as f:
                json.dump({"docs": [{"logs": [docs_logs[d][r]]}]}, f)
            cmd = [self.node, os.path.join(ROOT, "oracle", "cli.js"), "time", "--impl", self.impl, "--whole", "--budget-ms", str(int(timeout_s * 1000)), "--in", inp,
                   "--spans-out", os.path.join(self.td, "spans%d.json" % p)]
            self.jobs.append(subprocess.Popen(cmd, cwd=ROOT, stdout=subprocess.PIPE, text=True))
        self.flat = flat
        self.spans = {}  # (document of the sample, replica) -> {spans, text} as the reference's own code left the replica after the WHOLE log

    def finish(self):
        rows = []
        for p, pr in enumerate(self.jobs):
            try:
                o, _ = pr.communicate(timeout=max(1.0, self.timeout_s + 30 - (time.time() - self.t0)))
                rows.append(json.loads(o.strip().splitlines()[-1]))
                with open(os.path.join(self.td, "spans%d.json" % p)) as f:
                    e = json.load(f)["docs"][0]["expected"][0]
                if e is not None:
                    self.spans[self.flat[p]] = e
            except Exception:  # noqa: BLE001
                pr.kill()
        wall = time.time() - self.t0
        shutil.rmtree(self.td, ignore_errors=True)
        per_core = [r["ops_per_s"] for r in rows if r["seconds"] > 0]
        whole, cut = sum(r["logs"] for r in rows), sum(r.get("truncated_logs", 0) for r in rows)
        secs = [r["seconds"] for r in rows if r["logs"]]
        return {
            "value": float(sum(per_core)),
            "unit": "ops/s",
            "cores": len(rows),
            "kind": "reference" if self.impl == "ref" else "port",
            "per_core_ops_per_s": float(np.mean(per_core)) if per_core else 0.0,
            "seconds_per_whole_log": {"mean": float(np.mean(secs)), "min": float(np.min(secs)), "max": float(np.max(secs))} if secs else None,
            "sample": "%d WHOLE replica logs (+ %d cut at the %.0f s deadline), %d ops, of documents drawn at random from the resident batch: applyChange over every change + "
                      "getTextWithFormatting, one node process per log, one log per core, %.0f s wall beside the rest of the bench; value = sum of the cores' own rates" % (
                          whole, cut, self.timeout_s, sum(r["ops"] for r in rows), wall),
        }


def extra_legs(args, n_docs, first_doc, local, iters):
    """Extra legs (N = 1 only; none of them is `value`), run by tools/bench_extras.py in a process of its own so that nothing an experimental
    build does can cost the bench line: the patch-stream replay rate and the SAME workload under the experimental builds
    peritext_amd/lib/exp_*.so (__graft_entry__.EXPERIMENTS) and other launch shapes of the product build, each compared log for log with the
    product build's results in that process."""
    cmd = [sys.executable, os.path.join(ROOT, "tools", "bench_extras.py"), "--config", args.config, "--docs", str(n_docs), "--first-doc", str(first_doc), "--seed", str(args.seed),
           "--list-cap", str(args.list_cap), "--iters", str(iters), "--device", str(local)] + (["--ops", str(args.ops)] if args.ops else []) + (["--no-admission"] if args.no_admission else [])
    try:
        p = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900)
        return json.loads(p.stdout.strip().splitlines()[-1])
    except Exception as ex:  # noqa: BLE001
        return {"error": str(ex)[:300]}


def kernel_source_sha16():
    """sha256 over the kernel sources and the ABI header (tools/traffic_json.py records the same for the build its counters were taken on)."""
    import hashlib

    h = hashlib.sha256()
    d = os.path.join(ROOT, "peritext_amd", "csrc")
    for f in sorted(os.listdir(d)) + ["../../include/peritext_hip.h"]:
        with open(os.path.join(d, f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def load_traffic(n_logs, rows, launch):
    """PMC-measured HBM bytes per launch of this command, if profiles/ holds them for this very workload, launch shape AND build of the kernel sources
    (a traffic file of another build is refused: roofline.traffic is null then, never a stale number)."""
    p = os.path.join(ROOT, "profiles", "r06_hbm_traffic.json")
    if not os.path.exists(p):
        return None
    with open(p) as f:
        t = json.load(f)
    if t.get("n_logs") != n_logs or t.get("rows") != rows or t.get("kernel_source_sha16") != kernel_source_sha16() or list(t.get("launch") or []) != list(launch):
        return None
    return t


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="config4")
    ap.add_argument("--docs", type=int, default=65536, help="documents of the whole job (sharded over --gpus ranks)")
    ap.add_argument("--docs-per-gpu", type=int, default=0, help="fix the per-rank share instead (weak scaling), e.g. 8192 = the config-#4 shard of one of 8 GPUs")
    ap.add_argument("--ops", type=int, default=None, help="override ops per log (debug only; makes the number non-BASELINE)")
    ap.add_argument("--seed", type=int, default=2024)
    ap.add_argument("--check-docs", type=int, default=64, help="random documents of the resident batch checked against the reference on the host")
    ap.add_argument("--cpu-procs", type=int, default=0, help="host processes of the oracle / reference runs (0 = one per core)")
    ap.add_argument("--cpu-budget-s", type=float, default=20.0, help="seconds every process of the deadline-truncated CPU leg runs the reference")
    ap.add_argument("--cpu-whole-logs", type=int, default=48, help="whole replica logs the cpu_baseline runs through the reference, one node process (core) each")
    ap.add_argument("--cpu-whole-timeout-s", type=float, default=420.0, help="deadline of the whole-log CPU leg")
    ap.add_argument("--no-cpu", action="store_true", help="skip the reference run: no parity guard against the oracle, no cpu_baseline")
    ap.add_argument("--no-admission", action="store_true", help="skip applyChange's causal admission (seq/deps) in the timed path")
    ap.add_argument("--sustain-s", type=float, default=5.0, help="extra leg: back-to-back steps for at least this many seconds (clocks / thermals)")
    ap.add_argument("--host-sync-step", action="store_true", help="the round-1 step: engine on its own stream, a host-side sync between merge and digest check")
    ap.add_argument("--no-extras", action="store_true", help="skip the extra legs (patch-stream replay rate, experimental builds / launch shapes on the same workload)")
    ap.add_argument("--list-cap", type=int, default=1536, help="list elements per replica the generator holds on chip (the longest list of the 65 536 documents of this seed has 1 334: profiles/r04_j_*; the LDS per document decides how many are generated side by side)")
    args = ap.parse_args()

    import torch

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world == 1 and args.gpus > 1:
        raise SystemExit("launch with torch.distributed.run --nproc-per-node %d" % args.gpus)
    assert torch.cuda.is_available(), "bench.py needs an MI355X; there is no CPU fallback"
    # Fewer GPUs than ranks (a one-GPU box running the N > 1 path as a test: tests/test_gpu_bench_ranks.py): the ranks share the GPUs, and torch's side channel
    # — the 128-byte communicator id, the barriers, the max over ranks of the timings — goes over gloo on the host (torch's NCCL refuses two ranks on one
    # device).  The data-path collective is the library's either way (ptx_allgather_digests); with PTX_RCCL_LIB the library binds it from that path.
    n_dev = torch.cuda.device_count()
    shared_gpus = world > n_dev
    local = local % n_dev
    torch.cuda.set_device(local)
    side = "cpu" if shared_gpus else "cuda"  # where the side channel's small tensors live
    dist = None
    if world > 1:
        import torch.distributed as dist

        if shared_gpus:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))  # RCCL over xGMI

    from peritext_amd import abi, shard, wire, workloads
    from peritext_amd.engine import Engine

    cores = os.cpu_count() or 8
    weak = args.docs_per_gpu > 0
    if weak:
        first_doc, n_docs = rank * args.docs_per_gpu, args.docs_per_gpu
        total_docs = args.docs_per_gpu * world
    else:
        first_doc, n_docs = shard.doc_range(args.docs, rank, world)
        total_docs = args.docs
    eng = Engine(local, flags=abi.FLAG_NO_ELEM_RANK | (abi.FLAG_NO_ADMISSION if args.no_admission else 0))
    gcfg = workloads.gen_config(args.config, ops=args.ops)
    gen_args = (gcfg["replicas"], gcfg["ops_per_log"], gcfg["mix"], gcfg["mark_types"])
    replicas = gcfg["replicas"]

    # ---- the workload, made on the device: on-device change() (ptx_generate), every document of every rank distinct ----
    log("generating %d documents on the device" % n_docs)
    t_gen = time.time()
    db, gen_info = eng.generate(*gen_args, n_docs, args.seed, first_doc=first_doc, list_cap=args.list_cap)
    t_gen = time.time() - t_gen
    n_logs = eng.n_logs(db)
    rows = eng.n_ops(db)
    ops_per_step = n_logs * gcfg["ops_per_log"]
    n_changes = eng.n_changes(db)
    dr = eng.alloc_result(db)
    digests = torch.empty((n_logs, 2), dtype=torch.int64, device="cuda")
    counts = [n_logs_r * replicas for n_logs_r in ([shard.doc_range(args.docs, r, world)[1] for r in range(world)] if not weak else [n_docs] * world)]
    gathered = torch.empty((sum(counts), 2), dtype=torch.int64, device="cuda") if world > 1 else None
    conv_dev = torch.zeros(1, dtype=torch.int64, device="cuda")
    conv = conv_dev[0]
    comm = None
    if world > 1 and not args.host_sync_step:
        # the digest all-gather lives in the C ABI (ptx_allgather_digests: RCCL bound inside libperitext_hip.so); torch.distributed
        # only carries the 128-byte communicator id from rank 0 to the others
        if os.environ.get("PTX_RCCL_LIB"):
            eng.comm_use_library(os.environ["PTX_RCCL_LIB"])
        uid = torch.tensor(list(eng.comm_unique_id()) if rank == 0 else [0] * abi.COMM_ID_BYTES, dtype=torch.uint8, device=side)
        dist.broadcast(uid, 0)
        comm = eng.comm_init(bytes(uid.cpu().tolist()), rank, world)

    stream = None
    if not args.host_sync_step:
        # everything of a step on ONE stream (torch's): no host-side synchronisation inside the step
        stream = torch.cuda.Stream()
        eng.set_stream(stream.cuda_stream)
    events = []

    def step(timed):
        """One pass of the hot path.  The kernel's launch duration comes from events on the stream the kernel runs on."""
        nonlocal conv
        if stream is not None:
            with torch.cuda.stream(stream):
                if timed:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(stream)
                eng.merge(db, dr)
                if timed:
                    e1.record(stream)
                    events.append((e0, e1))
                if world == 1:
                    eng.count_converged(dr, replicas, conv_dev.data_ptr())
                else:  # N > 1: the only collective on the path, all of it inside the library, on the same stream
                    eng.allgather_digests(comm, dr, counts, gathered.data_ptr())
                    eng.count_converged_digests(gathered.data_ptr(), sum(counts), replicas, conv_dev.data_ptr())
                conv = conv_dev[0]
            return None
        ms = eng.merge_timed(db, dr, 1) if timed else eng.merge(db, dr)  # HIP events on the engine's own stream
        eng.pack_digests(dr, 0, n_logs, digests.data_ptr())
        eng.sync()
        conv, _ = shard.global_convergence(digests, replicas, dist if world > 1 else None, gathered, counts)
        return ms

    def fence():
        torch.cuda.synchronize()
        eng.sync()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    log("warmup")
    for _ in range(args.warmup):
        step(False)
    fence()
    log("timed region: %d steps" % args.steps)
    t0 = time.perf_counter()
    kernel_ms = [step(True) for _ in range(args.steps)]
    fence()
    elapsed = time.perf_counter() - t0
    if stream is not None:
        kernel_ms = [a.elapsed_time(b) for a, b in events]
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=side)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    converged_docs = int(conv.item())

    # ---- sustained leg: the same step back to back for >= --sustain-s seconds (is the rate a cold-boost burst?) ----
    sustained = None
    log("sustained leg")
    if args.sustain_s > 0:
        n_sus = max(args.steps, int(args.sustain_s / max(elapsed / args.steps, 1e-6)) + 1)
        if world > 1:
            t = torch.tensor([n_sus], dtype=torch.int64, device=side)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            n_sus = int(t.item())
        events.clear()
        fence()
        ts = time.perf_counter()
        sus_ms = [step(True) for _ in range(n_sus)]
        fence()
        sus_elapsed = time.perf_counter() - ts
        if stream is not None:
            sus_ms = [a.elapsed_time(b) for a, b in events]
        if world > 1:
            t = torch.tensor([sus_elapsed], dtype=torch.float64, device=side)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            sus_elapsed = float(t.item())
        third = max(1, n_sus // 3)
        sustained = {"seconds": sus_elapsed, "steps": n_sus, "ops_per_s": None,  # filled below (whole-job ops)
                     "kernel_ms_avg": float(np.mean(sus_ms)), "kernel_ms_first_third": float(np.mean(sus_ms[:third])), "kernel_ms_last_third": float(np.mean(sus_ms[-third:]))}
    if stream is not None:
        eng.set_stream(0)

    log("checks")
    # ---- every log ok, every document converged ----
    logs = eng.download_logs(dr, n_logs)
    assert int(logs["status"].max()) == 0, "a log failed"
    assert int(logs["n_ops"].sum()) == ops_per_step
    dg = logs["digest"].reshape(-1, replicas, 2)
    assert (dg == dg[:, :1, :]).all(), "replicas of a document must converge"
    assert converged_docs == total_docs, "device count of converged documents disagrees"

    total_ops_per_step = ops_per_step * world if weak else sum(counts) * gcfg["ops_per_log"]
    if rank == 0:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import helpers

        # ---- parity guard: random documents of the RESIDENT batch against the reference run on the host cores ----
        parity, cpu = None, None
        if not args.no_cpu:
            rng = np.random.default_rng(args.seed)
            pick = sorted(int(x) for x in rng.choice(n_docs, size=min(args.check_docs, n_docs), replace=False))
            ones, docs_logs = [], []
            for d in pick:
                hb, hinfo = eng.generate(*gen_args, 1, args.seed, first_doc=first_doc + d, list_cap=args.list_cap)
                actors_t, comments_t, log_doc_t = wire.generated_tables(1, replicas, hinfo["n_comments"])
                one = eng.download_batch(hb, wire.GEN_VALUES, wire.GEN_URLS, log_doc_t, actors_t, comments_t)
                eng.free_batch(hb)
                ones.append(one)
                docs_logs.append([wire.decode_changes(one, r) for r in range(replicas)])
            log("parity guard: %d documents through the oracle on the host" % len(pick))
            expected = oracle_expected(docs_logs, args.cpu_procs or cores)
            for d, one, exp in zip(pick, ones, expected):
                sub = eng.download_range(db, dr, d * replicas, replicas)  # the rows the timed launches wrote for this document
                for r in range(replicas):
                    helpers.check_log(one, sub, r, exp[r])
            parity = {"documents_checked": len(pick), "replica_logs_checked": len(pick) * replicas, "against": "oracle/peritext_oracle.js (whole logs)",
                      "what": "decoded spans, raw value/span/comment-interval rows and 128-bit digests of the resident batch's result rows"}
            whole = None
            if args.cpu_whole_logs > 0:
                log("cpu baseline: %d whole logs through the reference, one per core (collected at the end)" % args.cpu_whole_logs)
                whole = WholeLogBaseline(docs_logs, min(args.cpu_whole_logs, max(1, cores - 8)), args.cpu_whole_timeout_s)
            log("cpu, deadline-truncated leg: the reference on the other host cores, %.0f s" % args.cpu_budget_s)
            cpu_cut = cpu_truncated_leg(docs_logs, args.cpu_budget_s, max(1, (args.cpu_procs or cores) - (len(whole.jobs) if whole else 0) - 8))

        # ---- roofline: B_alg = sum over logs of 32*N + 4*V + 8*S + 16*T + 16 ----
        V, S, T = int(logs["n_visible"].sum()), int(logs["n_spans"].sum()), int(logs["n_cintervals"].sum())
        alg_bytes = 32 * rows + 4 * V + 8 * S + 16 * T + 16 * n_logs
        env_bytes = 0 if args.no_admission else abi.envelope_bytes(n_changes, replicas)
        k_ms = float(np.mean(kernel_ms))
        achieved = alg_bytes / (k_ms * 1e-3)
        # the same launch without the admission phase (PTX_FLAG_NO_ADMISSION), for reference: a second context on the same resident batch
        ms_noadm = None
        if not args.no_admission:
            eng2 = Engine(local, flags=abi.FLAG_NO_ELEM_RANK | abi.FLAG_NO_ADMISSION)
            eng2.merge(db, dr)
            eng2.sync()
            it = max(2, args.steps // 2)
            ms_noadm = eng2.merge_timed(db, dr, it) / it
            eng2.close()
        traffic = load_traffic(n_logs, rows, eng.launch_shape(db))
        extras = None
        if world == 1 and not args.no_extras:
            log("extra legs")
            extras = extra_legs(args, n_docs, first_doc, local, max(2, args.steps // 2))
        threads, lds = eng.launch_shape(db)
        out = {
            "metric": "CRDT ops applied+materialised per second (whole node)",
            "value": total_ops_per_step * args.steps / elapsed,
            "unit": "ops/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak" if weak else "strong",
            "vs_baseline": None,
            "dtype": "u64/u32 integer (opIds u64; indices u16/u32 in LDS)",
            "data": "synthetic: PTXGEN (seeded restatement of reference/test/fuzz.ts) generated ON THE DEVICE by ptx_generate (on-device change()); "
                    "%d distinct docs on this GPU, %d in the job, none repeated" % (n_docs, total_docs),
            "config": {
                "workload": "BASELINE config #4: %d docs x %d replicas x %d ops (%s)" % (
                    total_docs, replicas, gcfg["ops_per_log"],
                    "the whole 64K-doc batch on one GPU" if world == 1 and total_docs == 65536 else "%d docs per GPU on %d GPUs, doc-sharded" % (n_docs, world)),
                "ptxgen_config": args.config,
                "docs_total": total_docs,
                "docs_this_gpu": n_docs,
                "replica_logs_this_gpu": n_logs,
                "ops_this_gpu_per_step": ops_per_step,
                "op_log_bytes_this_gpu": 32 * rows,
                "changes_this_gpu": n_changes,
                "parallelism": "doc-sharded x%d, digests-only all-gather (ptx_allgather_digests: RCCL inside the C ABI)%s" % (
                    world, "; the ranks SHARE %d GPU(s): torch side channel over gloo, a test set-up, not a scaling figure" % n_dev if shared_gpus else ""),
                "causal_admission": not args.no_admission,
                "step": "merge + device-side convergence count, one stream, no host sync" if stream is not None else "merge, host sync, digest check",
            },
            "docs_converged_per_s": converged_docs * args.steps / elapsed,
            "docs_converged": converged_docs,
            "docs_total": total_docs,
            "roofline": {
                "bound": "hbm",
                "achieved": achieved / 1e9,
                "peak": HBM_PEAK / 1e9,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK,
                "frac_of_measured_copy_ceiling": achieved / HBM_COPY_CEILING,
                "traffic": None if traffic is None else traffic["hbm_bytes_per_launch"],
                "traffic_over_algorithmic": None if traffic is None else traffic["hbm_bytes_per_launch"] / alg_bytes,
                "traffic_source": None if traffic is None else traffic.get("source"),
                "traffic_note": None if traffic is None else traffic.get("note"),
                "kernel": eng.batch_kernel_name(db),
                "kernel_ms_avg": k_ms,
                "algorithmic_bytes_per_launch": alg_bytes,
                "formula": "sum over logs of 32*N + 4*V + 8*S + 16*T + 16 (SURVEY 8d; T = comment-interval rows)",
                "with_envelope": {"envelope_bytes_per_launch": env_bytes, "GBps": (alg_bytes + env_bytes) / (k_ms * 1e-3) / 1e9,
                                  "frac": (alg_bytes + env_bytes) / (k_ms * 1e-3) / HBM_PEAK},
            },
            "without_admission": None if ms_noadm is None else {"kernel_ms": ms_noadm, "ops_per_s_1gpu": ops_per_step / (ms_noadm * 1e-3),
                                                                "hbm_GBps": alg_bytes / (ms_noadm * 1e-3) / 1e9, "frac": alg_bytes / (ms_noadm * 1e-3) / HBM_PEAK},
            "sustained": sustained,
            "extras": extras,  # tools/bench_extras.py: patch-stream replay rate; experimental builds / launch shapes on the same workload (not `value`)
            "parity": parity,
            "launch": {"threads_per_log": threads, "lds_bytes_per_log": lds},
            "host": {"cores": cores, "gen_s": t_gen},
            "device_gen": {"kernel_ms": gen_info["kernel_ms"], "ops_generated_per_s": ops_per_step / (gen_info["kernel_ms"] * 1e-3),
                           "list_cap": args.list_cap, "longest_list": int(logs["n_elems"].max())},
        }
        if sustained is not None:
            sustained["ops_per_s"] = total_ops_per_step * sustained["steps"] / sustained["seconds"]
        if not args.no_cpu:
            if whole is not None:
                log("waiting for the whole-log CPU leg")
                cpu = whole.finish()
                cpu["deadline_truncated"] = cpu_cut
                # the whole logs the reference's OWN code has just replayed are parity checks against the reference itself (VERDICT r4 weak #1a: the guard above
                # uses the restated oracle): same documents, the rows the timed launches wrote
                if whole.impl == "ref" and whole.spans:
                    subs = {}
                    for (d, r), e in sorted(whole.spans.items()):
                        if d not in subs:
                            subs[d] = eng.download_range(db, dr, pick[d] * replicas, replicas)
                        helpers.check_log(ones[d], subs[d], r, e)
                    parity["against_the_reference_itself"] = {"replica_logs_checked": len(whole.spans), "documents": len(subs),
                                                              "how": "oracle/_ref (the reference's TypeScript, types erased) replayed these WHOLE logs for the cpu_baseline leg; what its replicas "
                                                                     "show at the end — decoded spans, raw rows, digests — equals the resident batch's result rows"}
            else:
                cpu = dict(cpu_cut, kind="reference" if os.path.exists(os.path.join(ROOT, "oracle", "_ref", "micromerge.js")) else "port")
            out["cpu_baseline"] = cpu
        print(json.dumps(out), flush=True)

    if world > 1:
        dist.barrier()  # rank 0 is still busy with the oracle / reference runs: leave together
    if comm is not None:
        eng.comm_destroy(comm)
    eng.free_result(dr)
    eng.free_batch(db)
    eng.close()
    if world > 1:  # rank 0 is still busy with the reference run: leave together
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
