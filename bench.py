#!/usr/bin/env python3
"""bench.py — whole-node throughput of the Peritext hot path on MI355X (BASELINE.json metric).

A "step" = one pass of the hot path over one resident batch: apply every replica op log of the batch and
materialise its formatted document (ptx_merge = ONE launch of ptx_merge_kernel), pack the per-replica
digests, (N>1: RCCL all-gather of the digests over xGMI), count converged documents on the device.
The op columns are resident in HBM before the timed region starts (PCIe upload is reported separately).

Workload (config.workload): BASELINE config #4 — 64K docs x 3 replicas x 4096 ops sharded over 8 GPUs =
8192 docs x 3 replicas per GPU ("weak" scaling: per-GPU work is fixed, N GPUs process N x 8192 docs).
The batch is PTXGEN documents (SURVEY.md §8d generator = the workload of reference/test/fuzz.ts, seeded), by default
GENERATED ON THE DEVICE (ptx_generate: on-device change(), every document of every rank distinct; change for change
the documents oracle/ptxgen.js makes — rank 0 re-checks one against the oracle in every run).  --oracle-gen takes
`--unique` documents from the oracle's own change() on the host instead and tiles them to 8192 docs in HBM.

One JSON line on stdout (rank 0).  `roofline.achieved` = algorithmic bytes of one launch
(32 B per op row + 32 B header per log + the Change envelope when causal admission is on, read; 4 B per visible value + 8 B per span + 12 B per comment interval + 48 B result
row per log written) / the kernel's average launch duration measured with HIP events on the stream
the kernel runs on.  `cpu_baseline` = the reference's own code (oracle/_ref, types erased) or, where
that is absent, the oracle port, timed on this box's host cores on a bounded sample of the same logs.
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

HBM_PEAK = 8.0e12  # B/s, MI355X spec (guide: 6.29e12 measured copy ceiling)
HBM_COPY_CEILING = 6.29e12


def gen_unique_docs(config, n_docs, seed, ops=None, procs=None):
    """PTXGEN documents from the oracle CLI, `procs` node processes in parallel; returns list of doc dicts."""
    node = shutil.which("node")
    if node is None:
        raise RuntimeError("bench.py needs node (the oracle/generator runtime) on this box")
    procs = max(1, min(procs or (os.cpu_count() or 8), n_docs, 64))
    td = tempfile.mkdtemp(prefix="ptxbench_")
    per = (n_docs + procs - 1) // procs
    jobs = []
    for p in range(procs):
        first = p * per
        cnt = min(per, n_docs - first)
        if cnt <= 0:
            break
        out = os.path.join(td, "g%d.json" % p)
        cmd = [node, os.path.join(ROOT, "oracle", "cli.js"), "gen", "--config", config, "--docs", str(cnt), "--first", str(first), "--seed", str(seed), "--out", out]
        if ops:
            cmd += ["--ops", str(ops)]
        jobs.append((subprocess.Popen(cmd, cwd=ROOT), out))
    docs = []
    for pr, out in jobs:
        if pr.wait() != 0:
            raise RuntimeError("oracle generator failed")
        with open(out) as f:
            docs += json.load(f)["docs"]
    shutil.rmtree(td, ignore_errors=True)
    return docs


def cpu_baseline(docs, budget_s, procs):
    """Time the reference's CPU path (applyChange over the whole log + getTextWithFormatting) on a sample:
    one node process per core, each on its own documents, each stopping after `budget_s`."""
    node = shutil.which("node")
    impl = "ref" if os.path.exists(os.path.join(ROOT, "oracle", "_ref", "micromerge.js")) else "oracle"
    procs = max(1, min(procs, len(docs)))
    td = tempfile.mkdtemp(prefix="ptxcpu_")
    jobs = []
    for p in range(procs):
        mine = docs[p::procs]
        inp = os.path.join(td, "in%d.json" % p)
        with open(inp, "w") as f:
            json.dump({"docs": [{"logs": d["logs"]} for d in mine]}, f)
        cmd = [node, os.path.join(ROOT, "oracle", "cli.js"), "time", "--in", inp, "--impl", impl, "--budget-ms", str(int(budget_s * 1000))]
        jobs.append(subprocess.Popen(cmd, cwd=ROOT, stdout=subprocess.PIPE, text=True))
    t0 = time.time()
    rows = []
    for pr in jobs:
        out, _ = pr.communicate()
        rows.append(json.loads(out.strip().splitlines()[-1]))
    wall = time.time() - t0
    shutil.rmtree(td, ignore_errors=True)
    ops = sum(r["ops"] for r in rows)
    logs = sum(r["logs"] for r in rows)
    cut = sum(r.get("truncated_logs", 0) for r in rows)
    per_core = [r["ops_per_s"] for r in rows if r["seconds"] > 0]
    return {
        "value": float(sum(per_core)),
        "unit": "ops/s",
        "cores": len(rows),
        "kind": "reference" if impl == "ref" else "port",
        "per_core_ops_per_s": float(np.mean(per_core)) if per_core else 0.0,
        "sample": "%d whole + %d deadline-truncated replica logs (%d ops) of the same PTXGEN documents, applyChange over every change + "
        "getTextWithFormatting, one node process per core, %.0f s budget each, %.1f s wall; per-op cost grows along a log, so truncated "
        "logs OVERSTATE the CPU rate" % (logs, cut, ops, budget_s, wall),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="config4")
    ap.add_argument("--docs-per-gpu", type=int, default=8192)
    ap.add_argument("--unique", type=int, default=64, help="unique PTXGEN documents per GPU (tiled to --docs-per-gpu)")
    ap.add_argument("--ops", type=int, default=None, help="override ops per log (debug only; makes the number non-BASELINE)")
    ap.add_argument("--seed", type=int, default=2024)
    ap.add_argument("--cpu-budget-s", type=float, default=30.0)
    ap.add_argument("--cpu-procs", type=int, default=16)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-admission", action="store_true", help="skip applyChange's causal admission (seq/deps) in the timed path")
    ap.add_argument("--oracle-gen", action="store_true", help="take the op logs from the oracle's generator on the host (--unique documents per GPU, "
                    "tiled in HBM) instead of generating them on the GPU (ptx_generate: on-device change(), every document distinct; the default)")
    ap.add_argument("--fused-step", action="store_true", help="EXPERIMENTAL (not yet measured): run the engine on a torch stream and count the converged "
                    "documents with one library kernel, so that a step has no host-side synchronisation")
    ap.add_argument("--list-cap", type=int, default=2048, help="--device-gen: list elements per replica held on chip")
    args = ap.parse_args()
    args.device_gen = not args.oracle_gen

    import torch

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d" % args.gpus)
    assert torch.cuda.is_available(), "bench.py needs an MI355X; there is no CPU fallback"
    torch.cuda.set_device(local)
    dist = None
    if world > 1:
        import torch.distributed as dist

        dist.init_process_group("nccl", device_id=torch.device("cuda", local))  # RCCL over xGMI

    from peritext_amd import abi, shard, wire
    from peritext_amd.engine import Engine

    cores = os.cpu_count() or 8
    eng = Engine(local, flags=abi.FLAG_NO_ELEM_RANK | (abi.FLAG_NO_ADMISSION if args.no_admission else 0))
    gen_info = None
    if args.device_gen:
        # ---- workload made on the device: on-device change() (ptx_generate), every document of every rank distinct ----
        from peritext_amd import workloads

        gcfg = workloads.gen_config(args.config, ops=args.ops)
        gen_args = (gcfg["replicas"], gcfg["ops_per_log"], gcfg["mix"], gcfg["mark_types"])
        t_gen = time.time()
        db, gen_info = eng.generate(*gen_args, args.docs_per_gpu, args.seed, first_doc=rank * args.docs_per_gpu, list_cap=args.list_cap)
        t_gen = time.time() - t_gen
        t_up, copies, replicas = 0.0, 1, gcfg["replicas"]
        n_logs = eng.n_logs(db)
        ops_per_step = n_logs * gcfg["ops_per_log"]
        n_changes_rank = eng.n_changes(db)
        max_actors = replicas
        # a few documents of the same stream on the host, for the oracle check and the CPU baseline leg
        docs = []
        if rank == 0:
            n_host = min(args.cpu_procs, args.docs_per_gpu)
            hb, hinfo = eng.generate(*gen_args, n_host, args.seed, first_doc=rank * args.docs_per_gpu, list_cap=args.list_cap)
            actors_t, comments_t, log_doc_t = wire.generated_tables(n_host, replicas, hinfo["n_comments"])
            host_batch = eng.download_batch(hb, wire.GEN_VALUES, wire.GEN_URLS, log_doc_t, actors_t, comments_t)
            eng.free_batch(hb)
            docs = [{"logs": [wire.decode_changes(host_batch, d * replicas + r) for r in range(replicas)]} for d in range(n_host)]
    else:
        # ---- workload: unique documents of this rank, tiled in HBM ----
        assert args.docs_per_gpu % args.unique == 0, "--docs-per-gpu must be a multiple of --unique"
        copies = args.docs_per_gpu // args.unique
        t_gen = time.time()
        docs = gen_unique_docs(args.config, args.unique, args.seed + 7919 * rank, ops=args.ops, procs=max(1, cores // max(world, 1)))
        t_gen = time.time() - t_gen
        replicas = len(docs[0]["logs"])
        batch = wire.encode_docs([d["logs"] for d in docs])
        ops_unique = batch.counted_ops()
        t_up = time.time()
        db = eng.upload(batch, copies=copies)
        eng.sync()
        t_up = time.time() - t_up
        n_logs = eng.n_logs(db)
        ops_per_step = ops_unique * copies  # counted ops (makeList rows excluded), this rank
        n_changes_rank = int(batch.chg_off[-1]) * copies
        max_actors = batch.max_actors
    dr = eng.alloc_result(db)
    n_docs = n_logs // replicas
    digests = torch.empty((n_logs, 2), dtype=torch.int64, device="cuda")
    gathered = torch.empty((world * n_logs, 2), dtype=torch.int64, device="cuda") if world > 1 else None
    conv = torch.zeros((), dtype=torch.int64, device="cuda")

    fused_stream = None
    fused_events = []
    if args.fused_step:
        # experimental: everything of a step on ONE stream (torch's), no host sync inside the step; the kernel's duration comes
        # from torch events around the launch, which now see the stream the kernel runs on
        fused_stream = torch.cuda.Stream()
        eng.set_stream(fused_stream.cuda_stream)
        conv_dev = torch.zeros(1, dtype=torch.int64, device="cuda")

    def step(timed):
        """One pass of the hot path.  Returns the kernel's launch duration in ms when `timed`."""
        nonlocal conv
        if fused_stream is not None:
            with torch.cuda.stream(fused_stream):
                if timed:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(fused_stream)
                eng.merge(db, dr)
                if timed:
                    e1.record(fused_stream)
                    fused_events.append((e0, e1))
                if world == 1:
                    eng.count_converged(dr, replicas, conv_dev.data_ptr())
                    conv = conv_dev[0]
                else:
                    eng.pack_digests(dr, 0, n_logs, digests.data_ptr())
                    conv, _ = shard.global_convergence(digests, replicas, dist, gathered)
            return None
        ms = None
        if timed:
            ms = eng.merge_timed(db, dr, 1)  # HIP events on the engine's stream around the one launch
        else:
            eng.merge(db, dr)
        eng.pack_digests(dr, 0, n_logs, digests.data_ptr())
        eng.sync()
        # N > 1: the only collective on the path — RCCL all-gather of the digests (peritext_amd/shard.py)
        conv, _ = shard.global_convergence(digests, replicas, dist if world > 1 else None, gathered)
        return ms

    for _ in range(args.warmup):
        step(False)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    eng.sync()
    t0 = time.perf_counter()
    kernel_ms = []
    for _ in range(args.steps):
        kernel_ms.append(step(True))
    torch.cuda.synchronize()
    eng.sync()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if fused_stream is not None:
        kernel_ms = [a.elapsed_time(b) for a, b in fused_events]
        eng.set_stream(0)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    converged_docs = int(conv.item())

    # ---- parity guard inside the bench: every log ok; rank 0 re-checks one document against the oracle ----
    logs = eng.download_logs(dr, n_logs)
    assert int(logs["status"].max()) == 0, "a log failed"
    assert int(logs["n_ops"].sum()) == ops_per_step

    if rank == 0:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import helpers

        one = wire.encode_docs([docs[0]["logs"]])
        res1 = eng.apply_materialize(one)
        expected0 = docs[0]["expected"] if "expected" in docs[0] else helpers.oracle_apply([docs[0]["logs"]])[0]
        for r_ in range(replicas):
            helpers.check_log(one, res1, r_, expected0[r_])

        # algorithmic bytes of ONE launch on this rank (with this ABI's row sizes)
        rows = eng.n_ops(db)
        n_changes = n_changes_rank
        env_bytes = 0 if args.no_admission else n_changes * (12 + 4 * max_actors)  # chg_actor, chg_seq, chg_nops, chg_deps row
        # the same launch without the admission phase (PTX_FLAG_NO_ADMISSION), for reference: a second engine on the same batch
        ms_noadm = None
        if not args.no_admission:
            eng2 = Engine(local, flags=abi.FLAG_NO_ELEM_RANK | abi.FLAG_NO_ADMISSION)
            if args.device_gen:
                db2, _ = eng2.generate(*gen_args, args.docs_per_gpu, args.seed, first_doc=rank * args.docs_per_gpu, list_cap=args.list_cap)
            else:
                db2 = eng2.upload(batch, copies=copies)
            dr2 = eng2.alloc_result(db2)
            eng2.merge(db2, dr2)
            eng2.sync()
            ms_noadm = eng2.merge_timed(db2, dr2, max(2, args.steps // 2)) / max(2, args.steps // 2)
            eng2.free_result(dr2)
            eng2.free_batch(db2)
            eng2.close()
        alg_bytes = env_bytes + 32 * rows + 32 * n_logs + 4 * int(logs["n_visible"].sum()) + 8 * int(logs["n_spans"].sum()) + 12 * int(logs["n_cintervals"].sum()) + 48 * n_logs
        k_ms = float(np.mean(kernel_ms))
        achieved = alg_bytes / (k_ms * 1e-3)
        total_ops = ops_per_step * world * args.steps
        out = {
            "metric": "CRDT ops applied+materialised per second (whole node)",
            "value": total_ops / elapsed,
            "unit": "ops/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u32/u64 integer (opIds u64, indices u16/u32 in LDS)",
            "data": ("synthetic: PTXGEN (seeded restatement of reference/test/fuzz.ts) generated ON THE DEVICE by ptx_generate (on-device change()); "
                     "%d distinct docs per GPU, none repeated" % n_docs) if args.device_gen else
                    "synthetic: PTXGEN (seeded restatement of reference/test/fuzz.ts) via the oracle's change(); %d unique docs per GPU tiled x%d in HBM" % (args.unique, copies),
            "config": {
                "workload": "BASELINE config #4 shard: %d docs x %d replicas x %d ops per GPU (64K docs x 3 x 4096 at 8 GPUs)"
                % (n_docs, replicas, (args.ops or {"config4": 4096, "config3": 1024, "config2": 256, "config5": 8192, "rich": 1024, "mini": 96}[args.config])),
                "ptxgen_config": args.config,
                "replica_logs_per_gpu": n_logs,
                "ops_per_gpu_per_step": ops_per_step,
                "op_log_bytes_per_gpu": 32 * rows,
                "parallelism": "doc-sharded x%d, digests-only all-gather" % world,
                "causal_admission": not args.no_admission,
                "fused_step": bool(args.fused_step),
                "changes_per_gpu_per_step": n_changes,
            },
            "docs_converged_per_s": converged_docs * args.steps / elapsed,
            "docs_converged": converged_docs,
            "docs_total": n_docs * world,
            "roofline": {
                "bound": "hbm",
                "achieved": achieved / 1e9,
                "peak": HBM_PEAK / 1e9,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK,
                "frac_of_measured_copy_ceiling": achieved / HBM_COPY_CEILING,
                "traffic": None,  # PMC counters cannot be read from inside the timed run; measured separately for this command:
                "traffic_profile": "profiles/r01_x_v40_hbm_traffic_pmc.txt: FETCH_SIZE + WRITE_SIZE per launch = 0.96 x the algorithmic bytes",
                "kernel": eng.kernel_name(),
                "kernel_ms_avg": k_ms,
                "algorithmic_bytes_per_launch": alg_bytes,
                "envelope_bytes_per_launch": env_bytes,
            },
            "without_admission": None if ms_noadm is None else {"kernel_ms": ms_noadm, "ops_per_s_1gpu": ops_per_step / (ms_noadm * 1e-3),
                                                                "hbm_GBps": (alg_bytes - env_bytes) / (ms_noadm * 1e-3) / 1e9},
            "launch": dict(zip(("threads_per_log", "lds_bytes_per_log"), eng.launch_shape(db))),
            "host": {"cores": cores, "gen_s": t_gen, "upload_s": t_up, "upload_GBps": None if args.device_gen else 32 * rows / copies / max(t_up, 1e-9) / 1e9},
            "device_gen": None if gen_info is None else {"kernel_ms": gen_info["kernel_ms"], "ops_generated_per_s": ops_per_step / (gen_info["kernel_ms"] * 1e-3),
                                                        "launch_shape": list(eng.launch_shape(db))},
        }
        if not args.no_cpu:
            out["cpu_baseline"] = cpu_baseline(docs, args.cpu_budget_s, min(args.cpu_procs, cores))
        print(json.dumps(out), flush=True)

    eng.free_result(dr)
    eng.free_batch(db)
    eng.close()
    if world > 1:
        dist.barrier()  # rank 0 is still busy with the CPU baseline leg: leave together
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
