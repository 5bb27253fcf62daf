"""Thin ctypes driver over the C ABI (include/peritext_hip.h).  Plumbing only: every merge goes
through libperitext_hip.so on a gfx950 device; nothing here computes a result on the CPU."""
import ctypes as C
import os

import numpy as np

from . import abi, wire


class PtxError(RuntimeError):
    def __init__(self, status, message):
        super().__init__("peritext_hip status %d: %s" % (status, message))
        self.status = status


def _batch_struct(b):
    s = abi.ptx_batch()
    s.n_logs = b.n_logs
    s.n_ops = b.n_ops
    p = lambda a, t: a.ctypes.data_as(t)  # noqa: E731
    s.log_off = p(b.log_off, abi.u64p)
    s.op_id = p(b.op_id, abi.u64p)
    s.ref_a = p(b.ref_a, abi.u64p)
    s.ref_b = p(b.ref_b, abi.u64p)
    s.payload = p(b.payload, abi.u32p)
    s.action = p(b.action, abi.u8p)
    s.mark_type = p(b.mark_type, abi.u8p)
    s.side_a = p(b.side_a, abi.u8p)
    s.side_b = p(b.side_b, abi.u8p)
    if b.chg_off is not None:  # a batch without the Change envelope (e.g. downloaded from wrapped device columns): NULL = no admission
        s.chg_off = p(b.chg_off, abi.u64p)
        s.chg_hdr = p(b.chg_hdr, abi.u32p)
        s.chg_env = p(b.chg_env, abi.u16p)
        if b.chg_env_hi is not None:
            s.chg_env_hi = p(b.chg_env_hi, abi.u16p)
        s.max_actors = b.max_actors
    if b.log_hdr is not None and len(b.log_hdr):
        s.log_hdr = b.log_hdr.ctypes.data_as(C.POINTER(abi.ptx_log_hdr))
    return s


def _copy_result(res):
    """ptx_result (library-owned host memory, COMPACT rows since ABI 7) -> wire.Results (numpy copies)."""
    nl, nr = int(res.n_logs), int(res.n_rows)

    def arr(ptr, dtype, n):
        if n == 0:
            return np.zeros(0, dtype=dtype)
        buf = (C.c_char * (n * np.dtype(dtype).itemsize)).from_address(C.addressof(ptr.contents))
        return np.frombuffer(buf, dtype=dtype, count=n).copy()

    voff, soff, coff = (arr(p, np.uint64, nl + 1) if nl else np.zeros(1, dtype=np.uint64) for p in (res.value_off, res.span_off, res.cint_off))
    return wire.Results(
        logs=arr(res.logs, abi.LOG_RESULT_DTYPE, nl),
        values=arr(res.values, np.uint32, int(voff[nl])),
        spans=arr(res.spans, abi.SPAN_DTYPE, int(soff[nl])),
        cintervals=arr(res.cintervals, abi.CINTERVAL_DTYPE, int(coff[nl])),
        elem_rank=arr(res.elem_rank, np.uint32, nr) if res.elem_rank else np.zeros(0, dtype=np.uint32),
        value_off=voff,
        span_off=soff,
        cint_off=coff,
    )


class Engine:
    """One device context (one HIP stream).  One Engine per process/GPU in multi-GPU runs."""

    def __init__(self, device=0, lib_path=None, flags=0):
        self.lib = abi.load_library(lib_path)
        ctx = C.c_void_p()
        st = self.lib.ptx_create(device, flags, C.byref(ctx))
        if st != 0:
            raise PtxError(st, (self.lib.ptx_last_error(None) or b"").decode())
        self.ctx = ctx
        self.device = device

    def _check(self, st):
        if st != 0:
            raise PtxError(st, (self.lib.ptx_last_error(self.ctx) or b"").decode())

    def set_launch_shape(self, threads_per_log=0, lds_bytes_per_log=0):
        """Threads per log / LDS window per log of the batches made resident after this call (0 = the library's choice)."""
        self._check(self.lib.ptx_set_launch_shape(self.ctx, threads_per_log, lds_bytes_per_log))

    def close(self):
        if self.ctx:
            self.lib.ptx_destroy(self.ctx)
            self.ctx = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # ---- the drop-in call ----
    def apply_materialize(self, batch):
        """Upload, merge, download: wire.Batch -> wire.Results (ptx_apply_materialize)."""
        s = _batch_struct(batch)
        res = abi.ptx_result()
        self._check(self.lib.ptx_apply_materialize(self.ctx, C.byref(s), C.byref(res)))
        try:
            return _copy_result(res)
        finally:
            self.lib.ptx_result_free(C.byref(res))

    # ---- staged form ----
    def upload(self, batch, copies=1):
        s = _batch_struct(batch)
        h = C.c_void_p()
        self._check(self.lib.ptx_batch_upload_tiled(self.ctx, C.byref(s), copies, C.byref(h)))
        return h

    def wrap_device(self, n_logs, n_ops, ptrs, log_hdr_ptr=0):
        """Adopt caller-owned DEVICE columns (ptx_batch_wrap_device): `ptrs` maps the nine column names of
        ptx_batch (log_off, op_id, ref_a, ref_b, payload, action, mark_type, side_a, side_b) to device addresses,
        e.g. torch tensors' data_ptr().  Nothing is copied; the log headers are computed on the device unless
        `log_hdr_ptr` (device address of n_logs ptx_log_hdr rows) is given."""
        s = abi.ptx_batch()
        s.n_logs = n_logs
        s.n_ops = n_ops
        for name, typ in (("log_off", abi.u64p), ("op_id", abi.u64p), ("ref_a", abi.u64p), ("ref_b", abi.u64p), ("payload", abi.u32p),
                          ("action", abi.u8p), ("mark_type", abi.u8p), ("side_a", abi.u8p), ("side_b", abi.u8p)):
            setattr(s, name, C.cast(C.c_void_p(int(ptrs[name])), typ))
        if log_hdr_ptr:
            s.log_hdr = C.cast(C.c_void_p(int(log_hdr_ptr)), C.POINTER(abi.ptx_log_hdr))
        h = C.c_void_p()
        self._check(self.lib.ptx_batch_wrap_device(self.ctx, C.byref(s), C.byref(h)))
        return h

    def append(self, dbatch, more):
        """Streaming append: a new resident batch whose log l = log l of `dbatch` + log l of the wire.Batch `more`."""
        s = _batch_struct(more)
        h = C.c_void_p()
        self._check(self.lib.ptx_batch_append(self.ctx, dbatch, C.byref(s), C.byref(h)))
        return h

    def resolve_cursors(self, dbatch, dresult, q_log, q_kind, q_arg):
        """Micromerge.resolveCursor / getCursor (micromerge.ts:465-477) for many replicas on the device (ptx_resolve_cursors):
        query q = (log, abi.CURSOR_RESOLVE, elemId as counter << 32 | actorRank) -> visible index, or (log, abi.CURSOR_GET, visible
        index) -> elemId.  Returns (out u64[n], status u32[n])."""
        ql = np.ascontiguousarray(q_log, dtype=np.uint32)
        qk = np.ascontiguousarray(q_kind, dtype=np.uint8)
        qa = np.ascontiguousarray(q_arg, dtype=np.uint64)
        n = len(ql)
        out = np.zeros(max(n, 1), dtype=np.uint64)
        status = np.zeros(max(n, 1), dtype=np.uint32)
        self._check(self.lib.ptx_resolve_cursors(self.ctx, dbatch, dresult, n, ql.ctypes.data_as(abi.u32p), qk.ctypes.data_as(abi.u8p), qa.ctypes.data_as(abi.u64p),
                                                 out.ctypes.data_as(abi.u64p), status.ctypes.data_as(abi.u32p)))
        return out[:n], status[:n]

    def append_device(self, dbatch, more_dbatch):
        """The same with `more` already resident (e.g. the batch Engine.change made)."""
        h = C.c_void_p()
        self._check(self.lib.ptx_batch_append_device(self.ctx, dbatch, more_dbatch, C.byref(h)))
        return h

    def change(self, dbatch, dresult, ops):
        """Micromerge.change for many replicas at once (ptx_change): `ops` = wire.InputOps (index-based InputOperations per
        log, grouped into the Changes to make) resolved against the replica states `dresult` = merge of `dbatch` holds.
        Returns (resident batch of the new Changes only, status per log as a numpy array)."""
        s = abi.ptx_input_ops()
        s.n_logs, s.max_actors = len(ops.chg_off) - 1, ops.max_actors
        p = lambda a, t: a.ctypes.data_as(t)  # noqa: E731
        s.chg_off, s.op_off = p(ops.chg_off, abi.u64p), p(ops.op_off, abi.u64p)
        s.action, s.mark_type = p(ops.action, abi.u8p), p(ops.mark_type, abi.u8p)
        s.index, s.count, s.payload = p(ops.index, abi.u32p), p(ops.count, abi.u32p), p(ops.payload, abi.u32p)
        s.values, s.n_values, s.actor = p(ops.values, abi.u32p), len(ops.values), p(ops.actor, abi.u32p)
        status = np.zeros(max(s.n_logs, 1), dtype=np.uint32)
        h = C.c_void_p()
        self._check(self.lib.ptx_change(self.ctx, dbatch, dresult, C.byref(s), C.byref(h), status.ctypes.data_as(abi.u32p)))
        return h, status[: s.n_logs]

    def free_batch(self, h):
        self.lib.ptx_batch_free(self.ctx, h)

    def alloc_result(self, dbatch):
        h = C.c_void_p()
        self._check(self.lib.ptx_result_alloc(self.ctx, dbatch, C.byref(h)))
        return h

    def free_result(self, h):
        self.lib.ptx_dresult_free(self.ctx, h)

    def merge(self, dbatch, dresult):
        self._check(self.lib.ptx_merge(self.ctx, dbatch, dresult))

    def merge_timed(self, dbatch, dresult, iters):
        """Milliseconds (HIP events on the engine's stream) of `iters` back-to-back merges."""
        ms = C.c_float()
        self._check(self.lib.ptx_merge_timed(self.ctx, dbatch, dresult, iters, C.byref(ms)))
        return float(ms.value)

    def root_map(self, dbatch):
        """The map objects of every replica of a resident batch (ptx_root_map: getRoot(), micromerge.ts:443-449): wire.RootMaps."""
        m = abi.ptx_root_maps()
        self._check(self.lib.ptx_root_map(self.ctx, dbatch, C.byref(m)))
        try:
            n = int(m.n_logs)
            off = np.ctypeslib.as_array(m.entry_off, shape=(n + 1,)).astype(np.uint64).copy()
            logs = np.frombuffer(C.string_at(m.logs, max(n, 1) * C.sizeof(abi.ptx_root_log)), dtype=abi.ROOT_LOG_DTYPE)[:n].copy()
            total = int(off[-1])
            ent = np.frombuffer(C.string_at(m.entries, total * C.sizeof(abi.ptx_root_entry)), dtype=abi.ROOT_ENTRY_DTYPE).copy() if total else np.zeros(0, dtype=abi.ROOT_ENTRY_DTYPE)
            return wire.RootMaps(entry_off=off, logs=logs, entries=ent)
        finally:
            self.lib.ptx_root_maps_free(C.byref(m))

    def phase_cycles(self, dbatch, dresult, n=32):
        """Diagnostic: shader-clock cycles per phase of merge_core.h, summed over all workgroups."""
        out = (C.c_uint64 * n)()
        self._check(self.lib.ptx_merge_phase_cycles(self.ctx, dbatch, dresult, out, n))
        return [int(x) for x in out]

    def set_stream(self, hip_stream):
        """Run this engine on the caller's HIP stream (an int: e.g. torch.cuda.current_stream().cuda_stream); 0 = its own again."""
        self._check(self.lib.ptx_set_stream(self.ctx, C.c_void_p(hip_stream or None)))

    def count_converged(self, dresult, replicas, count_device_ptr):
        """Documents whose `replicas` logs all carry the same digest, into a u64 in device memory (stream-ordered, no host sync)."""
        self._check(self.lib.ptx_count_converged(self.ctx, dresult, replicas, C.c_void_p(count_device_ptr)))

    def calib_stream(self, dbatch):
        """Diagnostic: stream the batch's columns once (kernel ptx_calib_stream_kernel); returns the bytes that stream is."""
        n = C.c_uint64()
        self._check(self.lib.ptx_calib_stream(self.ctx, dbatch, C.byref(n)))
        return int(n.value)

    # ---- multi-GPU: the digest all-gather over RCCL, inside the library ----
    def comm_use_library(self, path):
        """Bind the collective library from `path` (before the first comm_* call of the process) instead of the process's librccl.so.1."""
        self._check(self.lib.ptx_comm_use_library(self.ctx, os.fsencode(path)))

    def comm_unique_id(self):
        """128 bytes (ncclUniqueId) rank 0 makes and hands to the other ranks."""
        buf = (C.c_uint8 * abi.COMM_ID_BYTES)()
        self._check(self.lib.ptx_comm_unique_id(self.ctx, buf))
        return bytes(buf)

    def comm_init(self, unique_id, rank, n_ranks):
        buf = (C.c_uint8 * abi.COMM_ID_BYTES).from_buffer_copy(unique_id)
        h = C.c_void_p()
        self._check(self.lib.ptx_comm_init(self.ctx, buf, rank, n_ranks, C.byref(h)))
        return h

    def comm_destroy(self, comm):
        self.lib.ptx_comm_destroy(self.ctx, comm)

    def allgather_digests(self, comm, dresult, counts, out_device_ptr):
        """Digests of every rank's result -> [sum(counts), 2] u64 at `out_device_ptr` (rank-major), on the engine's stream."""
        c = np.ascontiguousarray(counts, dtype=np.uint32)
        if len(c) != self.lib.ptx_comm_n_ranks(comm):
            raise ValueError("allgather_digests: counts must hold one entry per rank of the communicator (%d)" % self.lib.ptx_comm_n_ranks(comm))
        self._check(self.lib.ptx_allgather_digests(self.ctx, comm, dresult, c.ctypes.data_as(abi.u32p), C.c_void_p(out_device_ptr)))

    def count_converged_digests(self, digests_device_ptr, n_logs, replicas, count_device_ptr):
        self._check(self.lib.ptx_count_converged_digests(self.ctx, C.c_void_p(digests_device_ptr), n_logs, replicas, C.c_void_p(count_device_ptr)))

    def sync(self):
        self._check(self.lib.ptx_sync(self.ctx))

    def download(self, dbatch, dresult):
        res = abi.ptx_result()
        self._check(self.lib.ptx_result_download(self.ctx, dbatch, dresult, C.byref(res)))
        try:
            return _copy_result(res)
        finally:
            self.lib.ptx_result_free(C.byref(res))

    def download_range(self, dbatch, dresult, first_log, n_logs):
        """Result rows of logs [first_log, first_log + n_logs) only, rebased to row 0 (a sample of a large resident batch)."""
        res = abi.ptx_result()
        self._check(self.lib.ptx_result_download_range(self.ctx, dbatch, dresult, first_log, n_logs, C.byref(res)))
        try:
            return _copy_result(res)
        finally:
            self.lib.ptx_result_free(C.byref(res))

    def replay_patches(self, dbatch, dresult, first_row=None):
        """The Patch[] stream every applyChange of every log would have returned (reference/src/micromerge.ts:499),
        as wire.Patches; `dresult` = the merge of the same batch (engine created without FLAG_NO_ELEM_RANK).
        first_row[l]: only the records of log l's rows from there on (ptx_replay_patches_from: the Changes appended to a resident replica)."""
        p = abi.ptx_patches()
        if first_row is None:
            self._check(self.lib.ptx_replay_patches(self.ctx, dbatch, dresult, C.byref(p)))
        else:
            first = np.ascontiguousarray(first_row, dtype=np.uint32)
            if len(first) != self.n_logs(dbatch):
                raise ValueError("first_row needs one entry per replica log")
            self._check(self.lib.ptx_replay_patches_from(self.ctx, dbatch, dresult, first.ctypes.data_as(C.c_void_p), C.byref(p)))
        try:
            n = int(p.n_logs)
            off = np.ctypeslib.as_array(p.patch_off, shape=(n + 1,)).copy() if n else np.zeros(1, dtype=np.uint64)
            logs = np.frombuffer(C.string_at(p.logs, n * C.sizeof(abi.ptx_patch_log)), dtype=abi.PATCH_LOG_DTYPE).copy() if n else np.zeros(0, dtype=abi.PATCH_LOG_DTYPE)
            total = int(off[n]) if n else 0
            if total:  # (not C.string_at: its size is a C int, and the streams of a large batch exceed 2 GiB)
                rows = np.ctypeslib.as_array(C.cast(p.patches, C.POINTER(C.c_uint8)), shape=(total * C.sizeof(abi.ptx_patch),)).view(abi.PATCH_DTYPE).copy()
            else:
                rows = np.zeros(0, dtype=abi.PATCH_DTYPE)
            return wire.Patches(patch_off=off, logs=logs, patches=rows, kernel_ms=float(p.kernel_ms), launches=int(p.launches))
        finally:
            self.lib.ptx_patches_free(C.byref(p))

    def generate(self, replicas, ops_per_log, mix, mark_types, n_docs, seed, first_doc=0, list_cap=0, initial_text=""):
        """On-device change(): n_docs PTXGEN documents (oracle/ptxgen.js, i.e. the workload of reference/test/fuzz.ts)
        generated straight into HBM.  Returns (resident batch handle, {"kernel_ms", "n_comments"})."""
        cfg = abi.ptx_gen_config()
        cfg.replicas, cfg.ops_per_log, cfg.n_mark_types = replicas, ops_per_log, len(mark_types)
        for i in range(4):
            cfg.mix[i] = mix[i]
        for i, t in enumerate(mark_types):
            cfg.mark_types[i] = t
        cfg.seed, cfg.first_doc, cfg.n_docs, cfg.list_cap = seed, first_doc, n_docs, list_cap
        cfg.initial_text = initial_text.encode("ascii")
        h = C.c_void_p()
        info = abi.ptx_gen_info()
        self._check(self.lib.ptx_generate(self.ctx, C.byref(cfg), C.byref(h), C.byref(info)))
        try:
            nc = np.ctypeslib.as_array(info.n_comments, shape=(n_docs,)).copy() if n_docs else np.zeros(0, dtype=np.uint32)
            return h, {"kernel_ms": float(info.kernel_ms), "n_comments": nc}
        finally:
            self.lib.ptx_gen_info_free(C.byref(info))

    def download_batch(self, dbatch, values=None, urls=None, log_doc=None, doc_actors=None, doc_comments=None, keys=None, map_values=None):
        """Copy a resident batch back as a wire.Batch (the decode tables are the caller's: a generated batch uses
        wire.GEN_VALUES / GEN_URLS / generated_tables)."""
        hb = abi.ptx_host_batch()
        self._check(self.lib.ptx_batch_download(self.ctx, dbatch, C.byref(hb)))
        try:
            b = hb.b
            L, T = int(b.n_logs), int(b.n_ops)

            def arr(ptr, dtype, n):
                return np.frombuffer(C.string_at(ptr, n * np.dtype(dtype).itemsize), dtype=dtype).copy() if n else np.zeros(0, dtype=dtype)

            log_off = arr(b.log_off, np.uint64, L + 1)
            has_env = bool(b.chg_off)
            chg_off = arr(b.chg_off, np.uint64, L + 1) if has_env else None
            NC = int(chg_off[-1]) if has_env else 0
            return wire.Batch(
                log_off, arr(b.op_id, np.uint64, T), arr(b.ref_a, np.uint64, T), arr(b.ref_b, np.uint64, T), arr(b.payload, np.uint32, T),
                arr(b.action, np.uint8, T), arr(b.mark_type, np.uint8, T), arr(b.side_a, np.uint8, T), arr(b.side_b, np.uint8, T),
                chg_off, arr(b.chg_hdr, np.uint32, NC) if has_env else None,
                arr(b.chg_env, np.uint16, NC * abi.env_stride(b.max_actors)) if has_env else None, int(b.max_actors), arr(b.log_hdr, abi.LOG_HDR_DTYPE, L), values or [], urls or [], log_doc or [], doc_actors or [], doc_comments or [], keys or [], map_values or [],
                chg_env_hi=arr(b.chg_env_hi, np.uint16, NC * abi.env_stride(b.max_actors)) if has_env and bool(b.chg_env_hi) else None)
        finally:
            self.lib.ptx_host_batch_free(C.byref(hb))

    def download_logs(self, dresult, n_logs):
        out = np.zeros(n_logs, dtype=abi.LOG_RESULT_DTYPE)
        self._check(self.lib.ptx_result_download_logs(self.ctx, dresult, out.ctypes.data_as(C.POINTER(abi.ptx_log_result)), n_logs))
        return out

    def pack_digests(self, dresult, first, count, dst_device_ptr):
        self._check(self.lib.ptx_pack_digests(self.ctx, dresult, first, count, C.c_void_p(dst_device_ptr)))

    def n_logs(self, dbatch):
        return int(self.lib.ptx_batch_n_logs(dbatch))

    def n_ops(self, dbatch):
        return int(self.lib.ptx_batch_n_ops(dbatch))

    def n_changes(self, dbatch):
        return int(self.lib.ptx_batch_n_changes(dbatch))

    def launch_shape(self, dbatch):
        """(threads per workgroup, dynamic LDS bytes per workgroup) the library chose for this batch."""
        t, l = C.c_uint32(), C.c_uint32()
        self.lib.ptx_batch_launch_shape(dbatch, C.byref(t), C.byref(l))
        return int(t.value), int(l.value)

    def max_ops_per_log(self):
        return int(self.lib.ptx_max_ops_per_log(self.ctx))

    def kernel_name(self):
        return self.lib.ptx_kernel_name().decode()

    def batch_kernel_name(self, dbatch):
        """The build of the merge kernel ptx_merge launches for this resident batch (its name in a rocprofv3 trace)."""
        return self.lib.ptx_batch_kernel_name(self.ctx, dbatch).decode()
