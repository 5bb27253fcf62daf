"""Multi-GPU layout of the hot path: documents shard across ranks, all replicas of a document stay
on one rank, no data-path collective; the only exchange is an all-gather of the per-replica 128-bit digests so that
every rank can state global convergence (the reference's `assert.deepStrictEqual(leftText, rightText)`,
test/fuzz.ts:277-278, for the whole batch).  `torch.distributed` backend "nccl" is RCCL over xGMI on the GPU box;
the same code runs over gloo on CPU tensors in the tests.  Plumbing only — merges happen in engine.py.

`doc_range` is what every host uses for the partition.  `allgather_digests` / `global_convergence` are the torch.distributed TWIN of the C ABI's
ptx_allgather_digests / ptx_count_converged_digests: the default bench step and the hosts use the C functions (RCCL bound inside the library; executed
between processes in tests/test_gpu_shard_ranks.py); the twin serves the world-size-2 gloo CPU test (tests/test_shard_gloo.py) and `bench.py --host-sync-step`."""


def doc_range(n_docs, rank, world):
    """Contiguous block of documents owned by `rank`: [first, first + count).  Blocks differ by at most one doc."""
    base, extra = divmod(n_docs, world)
    first = rank * base + min(rank, extra)
    return first, base + (1 if rank < extra else 0)


def converged_docs(digests, replicas):
    """digests: integer tensor [n_logs, 2] (two u64 halves reinterpreted as int64), logs grouped by document
    (`replicas` consecutive logs per document).  Returns a 0-d tensor: documents whose replicas all agree."""
    d = digests.view(-1, replicas, 2)
    return (d == d[:, :1, :]).all(dim=2).all(dim=1).sum()


def allgather_digests(digests, dist=None, out=None, counts=None):
    """All-gather of the per-rank digest tensors -> [sum(counts), 2] (rank-major).  `dist` is torch.distributed (already
    initialised) or None for a single process.  `counts[r]` = rows of rank r (doc_range gives blocks that differ by one
    document when the documents do not divide evenly): unequal blocks are gathered padded to the largest and compacted."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return digests
    import torch

    world = dist.get_world_size()
    n = digests.shape[0]
    if counts is None:
        counts = [n] * world
    assert len(counts) == world and counts[dist.get_rank()] == n, "counts must list every rank's rows"
    tail = tuple(digests.shape[1:])
    if all(c == n for c in counts):
        if out is None:
            out = torch.empty((world * n,) + tail, dtype=digests.dtype, device=digests.device)
        dist.all_gather_into_tensor(out, digests.contiguous())
        return out
    width = max(counts)
    mine = torch.zeros((width,) + tail, dtype=digests.dtype, device=digests.device)
    mine[:n] = digests
    padded = torch.empty((world * width,) + tail, dtype=digests.dtype, device=digests.device)
    dist.all_gather_into_tensor(padded, mine)
    parts = [padded[r * width:r * width + counts[r]] for r in range(world)]
    if out is None:
        return torch.cat(parts)
    torch.cat(parts, out=out)
    return out


def global_convergence(digests, replicas, dist=None, out=None, counts=None):
    """(converged documents, total documents) over all ranks, identical on every rank."""
    g = allgather_digests(digests, dist, out, counts)
    return converged_docs(g, replicas), g.shape[0] // replicas
