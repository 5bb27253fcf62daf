/*
 * peritext_hip.h — C ABI of the MI355X (gfx950) batch CRDT merge engine for Peritext's hot path.
 *
 * The reference (inkandswitch/peritext) has no FFI: its boundary is the TypeScript module API
 *     Micromerge.applyChange(change)            reference/src/micromerge.ts:499
 *     Micromerge.getTextWithFormatting(path)    reference/src/micromerge.ts:516 -> src/peritext.ts:337
 * called per replica by bridge.ts:253/288 and by the test harness (test/micromerge.ts:54-79,
 * test/fuzz.ts:198-205).  This header is what an N-API addon binds instead (INTEGRATION.md shows the
 * stub): ONE call applies MANY replica op logs and materialises the formatted documents.
 *
 * Conventions
 *   - plain C, plain pointers and sizes; no C++/torch types; every function returns a ptx_status
 *     (0 = ok) except the accessors; the last failure text is kept per context (ptx_last_error).
 *   - a context owns one HIP stream and every device allocation made through it; contexts are
 *     independent, one context must not be used from two threads at once.
 *   - per-LOG failures (the reference's throw sites, micromerge.ts:503,:507,:752) do not abort the
 *     batch: they are reported in ptx_log_result.status and that log's outputs are empty.
 *
 * Wire format of the op log (SoA, 32 bytes per op)
 *   A batch holds n_logs replica-logs.  A replica-log is the sequence of internal Operations
 *   (Change.ops flattened, micromerge.ts:60-71,:204-212) one replica applied, IN ITS APPLICATION
 *   ORDER.  Ops of log l are rows log_off[l] .. log_off[l+1]-1 of nine columns:
 *     op_id   u64  (counter << 32) | actorRank     "counter@actor" (micromerge.ts:488); actorRank =
 *                  rank of the actor string in UTF-16 code-unit order among the doc's actors, which
 *                  makes integer order == compareOpIds order (micromerge.ts:812-827)
 *     ref_a   u64  insert: elemId it was inserted after, 0 = HEAD (:153, :348); delete: elemId (:165);
 *                  mark: start.elemId (peritext.ts:17-21,:28)
 *     ref_b   u64  mark: end.elemId (peritext.ts:30); otherwise 0
 *     payload u32  insert: value id (index into the caller's string table); addMark link: url id;
 *                  add/removeMark comment: DOC-LOCAL comment id (rank among the document's comment ids, shared by
 *                  all replicas of the document) whose numeric order equals the code-unit order of the id strings
 *                  (peritext.ts:318 keeps the array sorted by id)
 *     action  u8   PTX_ACT_*      mark_type u8  PTX_MARK_* (schema.ts:125 ALL_MARKS order)
 *     side_a  u8   PTX_SIDE_* of start        side_b u8  PTX_SIDE_* of end
 *   Ops on the map objects (set / del / makeMap on the root map or a nested map) are PTX_ACT_MAPSET / PTX_ACT_MAPDEL rows: the text path
 *   ignores them, ptx_root_map resolves them; anything else is PTX_ACT_NOP.
 *
 * Output (all per log, canonical — two replicas have deep-equal getTextWithFormatting output iff
 * their canonical outputs, and therefore their digests, are equal):
 *     values     u32[n_visible]   value ids of the visible elements in document order
 *     spans      {start, attr}[n_spans]   maximal runs of visible elements with equal marks
 *                (peritext.ts:438-455): start = visible index of the first element, attr = flags<<28 | link url id
 *     cintervals {id, start, end}[n_cintervals]  for every comment id the maximal visible ranges
 *                [start,end) on which that comment is present, sorted by (id, start).  A span's
 *                `comment` array is the set of ids whose interval covers it (present iff
 *                PTX_ATTR_COMMENT is set — `comment: []` is a real state).
 *     digest     2 x u64 multiset hash of the above (peritext_amd/canon.py restates it).  Compared only with digests of the SAME build of the library
 *                (replicas of a document, ranks of a job): the per-item mixing function is not part of the ABI (round 6 replaced it, csrc/merge_core.h)
 *   Output rows of log l start at row log_off[l] of each output array (a log never produces more
 *   rows than it has ops), so no cross-log compaction or device-side allocation is needed.
 */
#ifndef PERITEXT_HIP_H
#define PERITEXT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PTX_ABI_VERSION 7u

/* Operation.action (micromerge.ts:150-212, peritext.ts:25-65) */
enum {
    PTX_ACT_MAKELIST = 0,   /* creates the text list; exactly one per log, ignored by the kernels */
    PTX_ACT_INSERT = 1,     /* {action:"set", insert:true}  -> applyListInsert  micromerge.ts:614 */
    PTX_ACT_DELETE = 2,     /* {action:"del", elemId}       -> applyListUpdate  micromerge.ts:677 */
    PTX_ACT_ADDMARK = 3,    /* applyAddRemoveMark peritext.ts:154 */
    PTX_ACT_REMOVEMARK = 4,
    PTX_ACT_NOP = 5,        /* an op the engine does not model: no effect */
    /* ops on MAP objects (the root map or a nested map), micromerge.ts:572-602: last-writer-wins per (object, key).  No effect on the
     * text path (ptx_merge ignores them); ptx_root_map resolves them.  ref_a = the map's object id (0 = the root map, else the opId
     * of the makeMap that created it), ref_b = key id (batch-wide string table).  The text list's own PTX_ACT_MAKELIST row is such a
     * write too (root map, its key in ref_b, kind PTX_MAPV_LIST). */
    PTX_ACT_MAPSET = 6,     /* {action:"set", key, value} / makeMap / makeList: mark_type = PTX_MAPV_*, payload = value id (scalars) */
    PTX_ACT_MAPDEL = 7      /* {action:"del", key} */
};
/* what a PTX_ACT_MAPSET row writes (its mark_type byte) / what ptx_root_entry.kind says */
enum { PTX_MAPV_SCALAR = 0, PTX_MAPV_MAP = 1, PTX_MAPV_LIST = 2, PTX_MAPV_DELETED = 3 };

/* markType, in ALL_MARKS order (schema.ts:125) */
enum { PTX_MARK_STRONG = 0, PTX_MARK_EM = 1, PTX_MARK_COMMENT = 2, PTX_MARK_LINK = 3 };

/* BoundaryPosition.type (peritext.ts:17-21) */
enum { PTX_SIDE_BEFORE = 0, PTX_SIDE_AFTER = 1, PTX_SIDE_START_OF_TEXT = 2, PTX_SIDE_END_OF_TEXT = 3 };

/* span attr: flags in the top 4 bits, link url id in the low 28 (0 when no link) */
#define PTX_ATTR_STRONG 0x10000000u
#define PTX_ATTR_EM 0x20000000u
#define PTX_ATTR_LINK 0x40000000u
#define PTX_ATTR_COMMENT 0x80000000u /* key `comment` present (possibly []) */
#define PTX_ATTR_ID_MASK 0x0fffffffu

/* Change envelope: chg_hdr word and chg_env row stride (u16 entries, a multiple of 4 so that rows are 8-byte aligned) */
#define PTX_CHG_ACTOR_SHIFT 20u
#define PTX_CHG_NOPS 0x000FFFFFu
#define PTX_ENV_STRIDE(max_actors) ((1u + (uint32_t)(max_actors) + 3u) & ~3u)
#define PTX_ENV_SATURATED 65535u

/* elem_rank: bit 31 marks a tombstone, the low 31 bits are the document position */
#define PTX_RANK_TOMBSTONE 0x80000000u
#define PTX_RANK_MASK 0x7fffffffu

typedef int32_t ptx_status;
enum {
    PTX_OK = 0,
    /* per-log (ptx_log_result.status): mirror the reference's throw sites */
    PTX_ERR_ELEM_NOT_FOUND = 1,  /* RangeError "List element not found"      micromerge.ts:752 */
    PTX_ERR_SEQ_GAP = 2,         /* RangeError "Expected sequence number"    micromerge.ts:503 */
    PTX_ERR_MISSING_DEP = 3,     /* RangeError "Missing dependency"          micromerge.ts:507 */
    PTX_ERR_DUPLICATE_OP = 4,    /* same opId twice in one log (the seq check makes this impossible upstream) */
    PTX_ERR_CAPACITY = 5,        /* log beyond what this build holds (ptx_merge: beyond the HBM-staged path's bounds or a forced LDS window; the on-chip-only entry points: beyond one CU's LDS) */
    PTX_ERR_BAD_OP = 6,          /* malformed row: unknown action / mark type / comment id beyond the header's n_comment_ids */
    PTX_ERR_INDEX_OOB = 7,       /* RangeError "List index out of bounds"    micromerge.ts:804 (ptx_change only) */
    /* call-level */
    PTX_ERR_INVALID_ARG = 100,
    PTX_ERR_HIP = 101,           /* a HIP runtime call failed; see ptx_last_error */
    PTX_ERR_NO_DEVICE = 102,
    PTX_ERR_OOM = 103
};

/* Per-log census, part of the wire format: it lets the kernel size its on-chip lists and id bitmaps
 * before it has seen a row, so the op columns are streamed ONCE.  The encoder that flattens Change.ops
 * (micromerge.ts:60-71) knows these numbers for free.  The kernel verifies them against the rows
 * (a wrong header is PTX_ERR_BAD_OP, never a wrong result); a batch may omit them (log_hdr = NULL),
 * then the library computes them with a small pre-pass kernel when the batch becomes resident. */
typedef struct ptx_log_hdr {
    uint32_t n_ins;       /* PTX_ACT_INSERT rows */
    uint32_t n_del;       /* PTX_ACT_DELETE rows */
    uint32_t n_mark[4];   /* add/removeMark rows per PTX_MARK_* */
    uint32_t max_counter; /* largest counter of any op_id of the log */
    uint32_t max_actor;   /* largest actorRank of any op_id of the log */
    uint32_t n_comment_ids; /* comment payloads of the log are < n_comment_ids: largest doc-local comment id the log uses + 1
                               (0 without comment ops).  Comment ids are ranks over ALL replicas of the document, so a replica
                               that has seen only some of the comments still carries the document's ranks (an upper bound
                               is accepted: it only sizes the per-id tables) */
    uint32_t reserved;
} ptx_log_hdr;

/* One batch of replica-logs.  All pointers are HOST pointers for ptx_batch_upload /
 * ptx_apply_materialize and DEVICE pointers for ptx_batch_wrap_device. */
typedef struct ptx_batch {
    uint32_t n_logs;
    uint32_t reserved;
    uint64_t n_ops;            /* == log_off[n_logs] */
    const uint64_t* log_off;   /* [n_logs + 1] */
    const uint64_t* op_id;     /* [n_ops] */
    const uint64_t* ref_a;     /* [n_ops] */
    const uint64_t* ref_b;     /* [n_ops] */
    const uint32_t* payload;   /* [n_ops] */
    const uint8_t* action;     /* [n_ops] */
    const uint8_t* mark_type;  /* [n_ops] */
    const uint8_t* side_a;     /* [n_ops] */
    const uint8_t* side_b;     /* [n_ops] */
    /* optional causal envelope (Change headers, micromerge.ts:60-71); NULL = skip causal admission.
     * When present (host batches: ptx_batch_upload / ptx_apply_materialize) the kernel admits every change
     * exactly like applyChange (micromerge.ts:499-511): seq == clock[actor] + 1 (PTX_ERR_SEQ_GAP) and
     * clock[a] >= deps[a] for all a (PTX_ERR_MISSING_DEP); 4 + 2 * PTX_ENV_STRIDE(max_actors) extra bytes are read per
     * change (12 B up to three actors).  chg_off[l]..chg_off[l+1]-1 are the changes of log l in application order. */
    const uint64_t* chg_off;   /* [n_logs + 1] or NULL */
    const uint32_t* chg_hdr;   /* [n_changes] actorRank << PTX_CHG_ACTOR_SHIFT | ops in the change (consecutive rows of the log) */
    const uint16_t* chg_env;   /* [n_changes * PTX_ENV_STRIDE(max_actors)] one row per change: seq, deps[0 .. max_actors) (0 = none),
                                  zero padding.  The LOW 16 bits of every value when the batch carries chg_env_hi; without it the
                                  values themselves, saturated at 65 535 (PTX_ENV_SATURATED) — a log of up to 65 533 changes can
                                  never admit such a change: the reference's RangeError either way (micromerge.ts:501-509) */
    uint32_t max_actors;       /* actors of a document (deps entries per change) */
    uint32_t reserved2;
    const ptx_log_hdr* log_hdr; /* [n_logs] or NULL (the library computes it) */
    /* optional wide envelope (ABI 6): the HIGH 16 bits of every chg_env value, same shape; NULL = every seq / dep of the batch fits 16
     * bits.  seq / deps are plain numbers in the reference (micromerge.ts:499-511): a replica that typed one change per keystroke
     * (bridge.ts:535) passes 65 535 changes of one actor.  Encoders emit the column whenever some value of the batch exceeds 65 534;
     * values are then EXACT (lo | hi << 16, no saturation).  A log whose rows of it are not all zero, or that holds more than 65 533
     * changes, is merged by the HBM-staged kernel (32-bit admission table); a log of more than 65 533 changes in a batch WITHOUT the
     * column cannot be represented and reports PTX_ERR_CAPACITY — never a spurious PTX_ERR_SEQ_GAP. */
    const uint16_t* chg_env_hi; /* [n_changes * PTX_ENV_STRIDE(max_actors)] or NULL */
} ptx_batch;

typedef struct ptx_span {
    uint32_t start; /* visible index of the first element of the span */
    uint32_t attr;  /* PTX_ATTR_* | link url id */
} ptx_span;

typedef struct ptx_cinterval {
    uint32_t id;    /* doc-local comment id */
    uint32_t start; /* visible index, inclusive */
    uint32_t end;   /* visible index, exclusive */
} ptx_cinterval;

typedef struct ptx_log_result {
    uint32_t status;        /* PTX_OK or a per-log PTX_ERR_* */
    uint32_t n_ops;         /* ops applied (the makeList and NOP rows excluded) */
    uint32_t n_elems;       /* list elements incl. tombstones */
    uint32_t n_visible;     /* rows of `values` */
    uint32_t n_spans;       /* rows of `spans` */
    uint32_t n_cintervals;  /* rows of `cintervals` */
    uint32_t reserved[2];   /* [0] diagnostic: LDS bytes the log needed; [1] on a per-row failure (status 1-4, 6): the row of the log at
                               which the reference's sequential replay would have thrown (the first op of the change for seq / deps
                               failures), else 0xffffffff */
    uint64_t digest[2];
} ptx_log_result;

/* Host-side view of a result set (returned by ptx_result_download[_range] / ptx_apply_materialize).  ABI 7: the rows are COMPACT — only the rows that
 * exist come down (a 4 096-op fuzz log shows a few dozen characters; round 4 downloaded one row per op: 2.4 GB for 100 M ops whose output is ~1 MB) —
 * gathered on the device into dense arrays and copied into pinned host memory:
 *     values[value_off[l] .. value_off[l + 1])          the n_visible value ids of log l, in document order
 *     spans[span_off[l] .. span_off[l + 1])             its n_spans span rows
 *     cintervals[cint_off[l] .. cint_off[l + 1])        its n_cintervals comment-interval rows
 * (a failed log has none).  elem_rank, where the context produces it, keeps one entry per op ROW of the downloaded logs (row r of log l at
 * log_off[l] - log_off[first log] + r).  Owned by the library until ptx_result_free. */
typedef struct ptx_result {
    uint32_t n_logs;
    uint32_t reserved;
    uint64_t n_rows;                  /* op rows of the downloaded logs (entries of elem_rank) */
    const ptx_log_result* logs;       /* [n_logs] */
    const uint32_t* values;           /* [value_off[n_logs]] */
    const ptx_span* spans;            /* [span_off[n_logs]] */
    const ptx_cinterval* cintervals;  /* [cint_off[n_logs]] */
    const uint32_t* elem_rank;        /* [n_rows] or NULL (PTX_FLAG_NO_ELEM_RANK).  Per op row: document position (incl. tombstones) of the
                                         element an INSERT row created (what findListElement(...).index
                                         would return, micromerge.ts:731), | PTX_RANK_TOMBSTONE when the
                                         element is deleted; 0xffffffff for other rows.  Enough to resolve
                                         cursors (getCursor / resolveCursor, micromerge.ts:465-477) on the host */
    const uint64_t* value_off;        /* [n_logs + 1] exclusive prefix sum of logs[].n_visible */
    const uint64_t* span_off;         /* [n_logs + 1] ... of logs[].n_spans */
    const uint64_t* cint_off;         /* [n_logs + 1] ... of logs[].n_cintervals */
    void* owner;
} ptx_result;

/* ---- incremental patch streams (what applyChange RETURNS, micromerge.ts:499 -> Patch[]) ----
 * One record per patch, in the replica's application order.  `row` = the op row of the log that produced it.
 *   PTX_PATCH_MAKELIST        the makeList op itself                          micromerge.ts:575
 *   PTX_PATCH_INSERT          a = visible index, b = marks of the new char (PTX_ATTR_* | link url id, as ptx_span.attr;
 *                             PTX_ATTR_COMMENT = key `comment` present)       micromerge.ts:661-671, peritext.ts:328
 *   PTX_PATCH_INSERT_COMMENT  follows its INSERT record: a = one doc-local comment id of the new char's marks
 *                             (ids in no particular order; the reference lists them sorted by id string)
 *   PTX_PATCH_DELETE          a = visible index, b = count (1)                micromerge.ts:696-703
 *   PTX_PATCH_ADDMARK / PTX_PATCH_REMOVEMARK  a = startIndex, b = endIndex (visible, exclusive); markType and attrs
 *                             are those of op `row`                           peritext.ts:251-281 */
enum { PTX_PATCH_MAKELIST = 0, PTX_PATCH_INSERT = 1, PTX_PATCH_DELETE = 2, PTX_PATCH_ADDMARK = 3, PTX_PATCH_REMOVEMARK = 4, PTX_PATCH_INSERT_COMMENT = 5 };
typedef struct ptx_patch {
    uint32_t row;
    uint32_t kind; /* PTX_PATCH_* */
    uint32_t a;
    uint32_t b;
} ptx_patch;
typedef struct ptx_patch_log {
    uint32_t status;    /* PTX_OK, or the log's merge status (no stream for a log the reference would have thrown on),
                           or PTX_ERR_CAPACITY (working set beyond the on-chip memory) */
    uint32_t n_patches; /* records of this log */
} ptx_patch_log;
/* Host-side view of the patch streams of a batch: records of log l at patches[patch_off[l] .. patch_off[l] + logs[l].n_patches).
 * Owned by the library until ptx_patches_free. */
typedef struct ptx_patches {
    uint32_t n_logs;
    uint32_t launches;          /* launches of the replay kernel it took: 1 (a log that outgrows the guessed capacity continues in an overflow extent; the records are
                                   packed to exact offsets on the device); 2 only when the overflow arena itself ran out */
    float kernel_ms;            /* duration of the last launch (HIP events) */
    uint32_t reserved;
    const uint64_t* patch_off;  /* [n_logs + 1] */
    const ptx_patch_log* logs;  /* [n_logs] */
    const ptx_patch* patches;   /* [patch_off[n_logs]] */
    void* owner;
} ptx_patches;

typedef struct ptx_ctx ptx_ctx;          /* device context: stream + allocations */
typedef struct ptx_dbatch ptx_dbatch;    /* a batch resident in HBM */
typedef struct ptx_dresult ptx_dresult;  /* result buffers resident in HBM */

/* ptx_create flags */
#define PTX_FLAG_NO_ELEM_RANK 1u /* do not produce ptx_result.elem_rank (saves 4 B/op of HBM writes) */
#define PTX_FLAG_NO_ADMISSION 2u /* ignore the Change envelope (chg_*) even when the batch carries it: no seq / deps checks */
#define PTX_FLAG_REPLAY_LDS_ONLY 8u /* ptx_replay_patches keeps its whole working set in LDS even where it exceeds 5.5 KB per log (by default the per-slot link urls and the
                                      tables of applied ops then live in global memory: four times the logs per CU); tuning / A-B */
#define PTX_FLAG_PAD_GATHER 4u   /* ptx_allgather_digests always takes its padded path (pack, all-gather of max(counts) pairs per rank, compact on the
                                    device) even when every rank holds the same number of logs: same result; lets a one-GPU host exercise that path */

/* ---- lifecycle ---- */
uint32_t ptx_abi_version(void);
/* device_ordinal: HIP device index.  Fails with PTX_ERR_NO_DEVICE when no gfx950 GPU is visible:
 * there is NO CPU fallback behind this ABI. */
ptx_status ptx_create(int device_ordinal, uint32_t flags, ptx_ctx** out);
void ptx_destroy(ptx_ctx* ctx);
/* Launch shape of the batches made resident AFTER this call (upload / append / generate / wrap): threads per replica log (a
 * multiple of 64, <= 1024) and the LDS window per log in bytes; 0 = the library's own choice by log size (the default).  A tuning
 * knob: results do not depend on it; a window too small for a log is that log's PTX_ERR_CAPACITY. */
ptx_status ptx_set_launch_shape(ptx_ctx* ctx, uint32_t threads_per_log, uint32_t lds_bytes_per_log);
const char* ptx_last_error(const ptx_ctx* ctx); /* ctx may be NULL: last ptx_create failure */

/* ---- the drop-in call: replaces a loop of applyChange(...) + getTextWithFormatting(["text"]) ---- */
/* Upload `batch` (host pointers), run the merge, download the results, synchronise. */
ptx_status ptx_apply_materialize(ptx_ctx* ctx, const ptx_batch* batch, ptx_result* out);
void ptx_result_free(ptx_result* res);

/* ---- staged form (what bench.py and the multi-GPU driver use: inputs resident in HBM) ---- */
ptx_status ptx_batch_upload(ptx_ctx* ctx, const ptx_batch* host, ptx_dbatch** out);
/* Build a resident batch made of `copies` back-to-back copies of `host` (distinct HBM addresses,
 * used to scale a synthetic batch to BASELINE sizes without regenerating it). */
ptx_status ptx_batch_upload_tiled(ptx_ctx* ctx, const ptx_batch* host, uint32_t copies, ptx_dbatch** out);
/* Adopt caller-owned DEVICE pointers (e.g. torch tensors' data_ptr); nothing is copied or freed. */
ptx_status ptx_batch_wrap_device(ptx_ctx* ctx, const ptx_batch* device, ptx_dbatch** out);
/* Streaming append: a NEW resident batch whose log l = log l of `base` followed by log l of `more` (host
 * pointers; the changes that arrived since, in application order; same n_logs, empty logs allowed).  Rows are copied
 * device to device; `base` stays valid.  `more` must be encoded with the tables of `base` (actor ranks, comment ranks,
 * value / url ids); both batches carry the Change envelope with the same max_actors, or neither does. */
ptx_status ptx_batch_append(ptx_ctx* ctx, const ptx_dbatch* base, const ptx_batch* more, ptx_dbatch** out);
void ptx_batch_free(ptx_ctx* ctx, ptx_dbatch* b);
uint32_t ptx_batch_n_logs(const ptx_dbatch* b);
uint64_t ptx_batch_n_ops(const ptx_dbatch* b);
uint64_t ptx_batch_n_changes(const ptx_dbatch* b); /* rows of the Change envelope (0 = the batch has none) */
/* Launch shape the library derived from the batch's log headers: threads per workgroup (= per log) and
 * dynamic LDS bytes per workgroup (the largest log's working set; it sets how many logs share a CU).  When at most a
 * tenth of the logs need more LDS than would let one more log share a CU, those are merged in a second launch of
 * their own and the figure reported here is that of the main launch. */
void ptx_batch_launch_shape(const ptx_dbatch* b, uint32_t* threads, uint32_t* lds_bytes);

ptx_status ptx_result_alloc(ptx_ctx* ctx, const ptx_dbatch* b, ptx_dresult** out);
void ptx_dresult_free(ptx_ctx* ctx, ptx_dresult* r);

/* Enqueue the merge of every log of `b` on the context's stream (asynchronous). */
ptx_status ptx_merge(ptx_ctx* ctx, const ptx_dbatch* b, ptx_dresult* r);
/* Same, `iters` times back to back, bracketed by HIP events on the context's stream:
 * *ms_total = elapsed milliseconds of the `iters` launches (for roofline accounting). */
ptx_status ptx_merge_timed(ptx_ctx* ctx, const ptx_dbatch* b, ptx_dresult* r, uint32_t iters, float* ms_total);
/* Diagnostic: run the merge once with per-phase cycle stamps (thread 0 of every workgroup) and return
 * the shader-clock cycles summed over all workgroups per phase: cycles[k], k < n <= 16, phases in the
 * order of merge_core.h (P1 classify, P2 index, P3a buckets, P3b child order, P3c tour+ranking, P4
 * tombstones, P5a values+mark intervals, P5b LWW, P5c comments, P6 spans).  Not for timed runs. */
ptx_status ptx_merge_phase_cycles(ptx_ctx* ctx, const ptx_dbatch* b, ptx_dresult* r, uint64_t* cycles, uint32_t n);
/* Diagnostic (profiling runs only): stream every op column — and the Change envelope when the batch has one — exactly once
 * with the element widths the merge kernel uses (kernel `ptx_calib_stream_kernel`); *bytes_read = the bytes that stream is,
 * 32 per op row + the envelope.  It calibrates the HBM-traffic PMC counters on a known byte count in this library's own
 * access pattern (tools/pmc_traffic.sh).  Synchronises. */
ptx_status ptx_calib_stream(ptx_ctx* ctx, const ptx_dbatch* b, uint64_t* bytes_read);
ptx_status ptx_sync(ptx_ctx* ctx);
/* Run the context's launches and copies on the caller's HIP stream (e.g. torch's current stream: everything stays
 * stream-ordered with the caller's own kernels and no host-side synchronisation is needed between them).  `hip_stream`
 * = a hipStream_t; NULL returns to the context's own stream.  The stream stays the caller's. */
ptx_status ptx_set_stream(ptx_ctx* ctx, void* hip_stream);
/* Documents whose `replicas` consecutive logs all have the same digest (= converged, test/fuzz.ts:277-278), counted on the
 * device: *count_device (a u64 in DEVICE memory) is overwritten.  n_logs must be a multiple of `replicas`. */
ptx_status ptx_count_converged(ptx_ctx* ctx, const ptx_dresult* r, uint32_t replicas, uint64_t* count_device);

ptx_status ptx_result_download(ptx_ctx* ctx, const ptx_dbatch* b, const ptx_dresult* r, ptx_result* out);
/* The same for the logs [first_log, first_log + n_logs) only (e.g. a sample of a large resident batch): out->n_logs = n_logs,
 * out->n_rows = their rows, row r of the k-th log of the range at index (log_off[first_log + k] - log_off[first_log]) + r. */
ptx_status ptx_result_download_range(ptx_ctx* ctx, const ptx_dbatch* b, const ptx_dresult* r, uint32_t first_log, uint32_t n_logs, ptx_result* out);
/* Only the per-log rows (status, counts, digests): [n_logs] ptx_log_result into caller memory. */
ptx_status ptx_result_download_logs(ptx_ctx* ctx, const ptx_dresult* r, ptx_log_result* out, uint32_t n_logs);
/* Device address of the per-log result rows (for a collective on the digests); never freed by the caller. */
const ptx_log_result* ptx_dresult_logs_device(const ptx_dresult* r);
/* Pack the digests of logs [first, first+count) as 2*count u64 into DEVICE memory `dst`
 * (e.g. a torch tensor that is then all-gathered with RCCL), on the context's stream. */
ptx_status ptx_pack_digests(ptx_ctx* ctx, const ptx_dresult* r, uint32_t first, uint32_t count, uint64_t* dst_device);

/* ---- multi-GPU: the digest all-gather (the reference's convergence assert, test/fuzz.ts:277-278, for a sharded batch) ----
 * Documents shard across ranks (one process per GPU), every replica of a document on one rank; the only exchange is an
 * all-gather of the per-replica 128-bit digests over RCCL (xGMI inside a node) so that every rank can state global convergence.
 * RCCL is loaded at run time (librccl.so.1) by the first of these calls: a single-GPU host never needs it. */
#define PTX_COMM_ID_BYTES 128 /* sizeof(ncclUniqueId) */
typedef struct ptx_comm ptx_comm;
/* Optional, before the first ptx_comm_* call of the process: bind the collective library (the five RCCL entry points this ABI uses) from `path`
 * instead of the process's librccl.so.1 — a host that ships its own RCCL build; the test-suite's shared-memory stand-in, which lets several ranks
 * share one GPU.  PTX_ERR_INVALID_ARG once a library is bound. */
ptx_status ptx_comm_use_library(ptx_ctx* ctx, const char* path);
/* Rank 0 makes the id (ncclGetUniqueId) and hands it to the other ranks over the host's own channel. */
ptx_status ptx_comm_unique_id(ptx_ctx* ctx, uint8_t id[PTX_COMM_ID_BYTES]);
/* Collective over all ranks (ncclCommInitRank) on the context's device. */
ptx_status ptx_comm_init(ptx_ctx* ctx, const uint8_t id[PTX_COMM_ID_BYTES], uint32_t rank, uint32_t n_ranks, ptx_comm** out);
void ptx_comm_destroy(ptx_ctx* ctx, ptx_comm* comm);
uint32_t ptx_comm_n_ranks(const ptx_comm* comm); /* 0 for NULL: hosts check the length of `counts` against it */
uint32_t ptx_comm_rank(const ptx_comm* comm);
/* All-gather of the digests of every rank's result: counts[r] = replica logs of rank r (host array [n_ranks]; counts[rank] must be
 * the logs of `r`), out_device = [sum(counts)] x 2 u64 in DEVICE memory, rank-major.  Enqueued on the context's stream; ranks
 * whose blocks differ in size are gathered padded and compacted on the device. */
ptx_status ptx_allgather_digests(ptx_ctx* ctx, ptx_comm* comm, const ptx_dresult* r, const uint32_t* counts, uint64_t* out_device);
/* Documents whose `replicas` consecutive digest pairs are all equal, over a gathered digest array in DEVICE memory (n_logs pairs):
 * *count_device (a u64 in DEVICE memory) is overwritten.  A failed log has the digest {0, 0} and never converges with a good one;
 * per-log statuses stay with the rank that owns the log. */
ptx_status ptx_count_converged_digests(ptx_ctx* ctx, const uint64_t* digests_device, uint64_t n_logs, uint32_t replicas, uint64_t* count_device);

/* Device memory for hosts that have no HIP binding of their own (the N-API addon): scratch for out_device / count_device above.
 * ptx_device_read copies `bytes` back to the host after everything enqueued on the context's stream has completed. */
ptx_status ptx_device_alloc(ptx_ctx* ctx, uint64_t bytes, void** out_device);
void ptx_device_free(ptx_ctx* ctx, void* device);
ptx_status ptx_device_read(ptx_ctx* ctx, const void* device, void* host, uint64_t bytes);

/* ---- patch streams ---- */
/* Replay every log of `b` in application order and return the Patch[] stream each applyChange would have returned.
 * `r` must be the result of ptx_merge on the same batch, produced WITH elem_rank (no PTX_FLAG_NO_ELEM_RANK) and
 * complete (the call synchronises).  Capacity is guessed (2 records per op) and the launch repeated once with exact
 * sizes when a log produced more. */
ptx_status ptx_replay_patches(ptx_ctx* ctx, const ptx_dbatch* b, const ptx_dresult* r, ptx_patches* out);
/* The same for the TAIL of every log: first_row[l] (host array [n_logs]; NULL = 0 everywhere) is the first row of log l whose records are wanted —
 * what the applyChange calls of the Changes appended since the last call return (a host that keeps its replicas resident: ptx_batch_append, ptx_merge,
 * this).  The rows before are replayed for their state only: no record of theirs is written, counted or downloaded; `row` in the records stays the row
 * in the log.  first_row[l] >= the log's rows: an empty stream. */
ptx_status ptx_replay_patches_from(ptx_ctx* ctx, const ptx_dbatch* b, const ptx_dresult* r, const uint32_t* first_row, ptx_patches* out);
void ptx_patches_free(ptx_patches* p);

/* ---- the map objects of a replica: getRoot() (micromerge.ts:443-449) ----
 * applyOp on a map object (micromerge.ts:572-602) keeps, per key, the op with the largest opId (compareOpIds): last writer wins;
 * "del" removes the key, makeMap / makeList put a child object there.  ptx_root_map resolves, per replica log, every (object, key)
 * the log's PTX_ACT_MAPSET / PTX_ACT_MAPDEL / PTX_ACT_MAKELIST rows write: one ptx_root_entry per pair = the winning row.  An op
 * on an object that does not exist when it is applied (micromerge.ts:538-540 "Object does not exist"), or a key op on a list
 * object, fails the log with PTX_ERR_ELEM_NOT_FOUND (first failing row in first_bad_row).  Entries of log l are
 * entries[entry_off[l] .. entry_off[l] + logs[l].n_entries), in no particular order.  Owned by the library until ptx_root_maps_free. */
typedef struct ptx_root_entry {
    uint64_t obj;    /* the map: 0 = root, else the opId of its makeMap */
    uint32_t key;    /* key id */
    uint32_t row;    /* the winning row of the log (its op_id is the child's object id for PTX_MAPV_MAP / PTX_MAPV_LIST) */
    uint32_t kind;   /* PTX_MAPV_*: PTX_MAPV_DELETED = the key is absent */
    uint32_t value;  /* payload of the winning row (PTX_MAPV_SCALAR: value id) */
} ptx_root_entry;
typedef struct ptx_root_log {
    uint32_t status;        /* PTX_OK, PTX_ERR_ELEM_NOT_FOUND, PTX_ERR_CAPACITY (more map ops than the on-chip table holds) */
    uint32_t n_entries;
    uint32_t first_bad_row; /* 0xFFFFFFFF when status == PTX_OK */
    uint32_t reserved;
} ptx_root_log;
typedef struct ptx_root_maps {
    uint32_t n_logs;
    uint32_t reserved;
    const uint64_t* entry_off;     /* [n_logs + 1] */
    const ptx_root_log* logs;      /* [n_logs] */
    const ptx_root_entry* entries;
    void* owner;
} ptx_root_maps;
ptx_status ptx_root_map(ptx_ctx* ctx, const ptx_dbatch* b, ptx_root_maps* out);
void ptx_root_maps_free(ptx_root_maps* m);

/* ---- on-device change(): op logs generated in HBM ----
 * The workload of the reference's fuzzer (test/fuzz.ts:115-205) in its seeded form PTXGEN (oracle/ptxgen.js): per
 * document `replicas` replicas; every step a random replica makes one change() (micromerge.ts:308-441: insert /
 * delete / addMark / removeMark given by visible indexes, resolved to element ids incl. lookAfterTombstones :762-805
 * and changeMark peritext.ts:458-501), then two random replicas exchange what the other lacks with applyChange (:499),
 * retrying in the reference's order on the causal RangeError; a full sync ends the document.  Document `first_doc + i`
 * of seed `seed` is, change for change, the one oracle/ptxgen.js generates for (seed, first_doc + i).
 * String tables of a generated batch are fixed: insert payload = the character's code, link payload = letter index of
 * "<A-Z>.com", comment payload = rank of "comment-<k>" in string order among the document's n_comments ids,
 * actors "doc1".."doc<replicas>". */
typedef struct ptx_gen_config {
    uint32_t replicas;      /* 1..8 */
    uint32_t ops_per_log;   /* ops of every replica log (the makeList row comes on top) */
    uint32_t mix[4];        /* percent of insert, delete, addMark, removeMark steps */
    uint32_t n_mark_types;  /* 0..4 */
    uint8_t mark_types[4];  /* PTX_MARK_* the mark steps draw from, in the workload's order */
    uint32_t seed;
    uint32_t first_doc;
    uint32_t n_docs;
    uint32_t list_cap;      /* list elements (incl. tombstones) per replica the on-chip state holds; 0 = ops_per_log + 8 (always enough) */
    char initial_text[16];  /* NUL-terminated ASCII; "" = "ABCDE" (generateDocs.ts:11-42) */
} ptx_gen_config;
typedef struct ptx_gen_info {
    uint32_t n_docs;
    float kernel_ms;            /* duration of the generator launch (HIP events) */
    const uint32_t* n_comments; /* [n_docs] */
    void* owner;
} ptx_gen_info;
/* Generate n_docs documents (n_docs * replicas logs, document-major) as a resident batch, envelope and headers
 * included, ready for ptx_merge.  PTX_ERR_CAPACITY: some document outgrew list_cap (or the LDS). */
ptx_status ptx_generate(ptx_ctx* ctx, const ptx_gen_config* cfg, ptx_dbatch** out, ptx_gen_info* info);
void ptx_gen_info_free(ptx_gen_info* info);
/* ---- cursors: Micromerge.getCursor / resolveCursor (micromerge.ts:465-477) for many replicas ----
 * Query q names a replica log and is either
 *   PTX_CURSOR_RESOLVE  arg = the cursor's elemId (counter << 32 | actorRank)  ->  out = visible elements BEFORE that element
 *                       (findListElement(...).visible, :731-755; the element itself may be a tombstone — a cursor survives the
 *                       deletion of its character); PTX_ERR_ELEM_NOT_FOUND for an id that names no list element (:752)
 *   PTX_CURSOR_GET      arg = a visible index  ->  out = elemId of the index-th visible element (getListElementId, :762-805);
 *                       PTX_ERR_INDEX_OOB past the end (:804)
 * `r` = ptx_merge of `b` WITH elem_rank, complete.  All arrays are HOST memory, one entry per query; status_out[q] may also be
 * the log's merge status (a replica the reference threw on has no cursors) or PTX_ERR_CAPACITY.  The call synchronises. */
enum { PTX_CURSOR_RESOLVE = 0, PTX_CURSOR_GET = 1 };
ptx_status ptx_resolve_cursors(ptx_ctx* ctx, const ptx_dbatch* b, const ptx_dresult* r, uint32_t n_queries, const uint32_t* q_log, const uint8_t* q_kind,
                               const uint64_t* q_arg, uint64_t* out, uint32_t* status_out);

/* ---- change(): caller-supplied InputOperations made into Changes on the device ----
 * Replaces `doc.change(ops: InputOperation[])` (micromerge.ts:308-441; InputOperation :133-148) for MANY replicas at once:
 * log l of `base` is a replica's op log as applied so far, the InputOperations of log l are resolved against THAT replica's
 * state (getListElementId incl. lookAfterTombstones :762-805, changeMark peritext.ts:458-501) into id-based ops and returned as
 * new Changes {actor, seq = clock + 1, deps = the clock before, startOp = maxOp + 1, ops} — one Change per entry of chg_off.
 * Columns (HOST pointers), one row per InputOperation:
 *   action      PTX_IN_INSERT {index, values}   PTX_IN_DELETE {index, count}   PTX_IN_ADDMARK / PTX_IN_REMOVEMARK {startIndex,
 *               endIndex, markType, attrs}      PTX_IN_MAKELIST {key: "text"} (only on a log without one)
 *   index       insert / delete: index            marks: startIndex
 *   count       insert: number of values          delete: count        marks: endIndex
 *   payload     insert: position of its first value in `values`        link: url id        comment: doc-local comment id
 *   mark_type   PTX_MARK_*
 * `actor[l]` = actorRank of the replica behind log l (ranks as in the op ids of `base`).  Ids follow the tables of `base`: a new
 * comment id must already have its rank (the caller encodes the document with the ids it is about to use). */
enum { PTX_IN_INSERT = 0, PTX_IN_DELETE = 1, PTX_IN_ADDMARK = 2, PTX_IN_REMOVEMARK = 3, PTX_IN_MAKELIST = 4,
       /* on a MAP object (micromerge.ts:400-425): {action: "set", key, value} / makeMap / makeList -> one PTX_ACT_MAPSET row, {action: "del",
        * key} -> one PTX_ACT_MAPDEL row.  index = the map object the host resolved the path to (getObjectIdForPath, :446-463): 0 = the
        * root map, PTX_IN_OBJ_NEW | k = the object made by row k of THIS log's output (a makeMap earlier in the same call), else counter
        * << 12 | actor rank of the makeMap that created it; count = key id, mark_type = PTX_MAPV_*, payload = value id */
       PTX_IN_MAPSET = 5, PTX_IN_MAPDEL = 6 };
#define PTX_IN_OBJ_NEW 0x80000000u
typedef struct ptx_input_ops {
    uint32_t n_logs;            /* == logs of the base batch */
    uint32_t max_actors;        /* actors of a document = row stride of the deps the new Changes carry (must equal the base
                                   batch's when that carries the envelope) */
    const uint64_t* chg_off;    /* [n_logs + 1] Changes to make per log (a log may make none) */
    const uint64_t* op_off;     /* [n_changes + 1] InputOperations per Change */
    const uint8_t* action;      /* [n_input_ops] PTX_IN_* */
    const uint8_t* mark_type;   /* [n_input_ops] */
    const uint32_t* index;      /* [n_input_ops] */
    const uint32_t* count;      /* [n_input_ops] */
    const uint32_t* payload;    /* [n_input_ops] */
    const uint32_t* values;     /* [n_values] value ids of the inserts */
    uint64_t n_values;
    const uint32_t* actor;      /* [n_logs] */
} ptx_input_ops;
/* `merged` = ptx_merge of `base` WITH elem_rank, complete (the call synchronises).  *made = a resident batch of n_logs logs
 * holding ONLY the new Changes (rows + envelope + headers), ready for ptx_batch_append_device / ptx_batch_download.
 * status_out[l] (caller memory, [n_logs]): PTX_OK, the log's merge status (a broken replica cannot make changes),
 * PTX_ERR_INDEX_OOB (the reference's RangeError, micromerge.ts:804), PTX_ERR_BAD_OP (no text list / a second makeList /
 * unknown action) or PTX_ERR_CAPACITY; a failed log makes NO change (the reference leaves the replica half-mutated: the ops
 * before the throw stay applied although no Change was returned — not reproduced). */
ptx_status ptx_change(ptx_ctx* ctx, const ptx_dbatch* base, const ptx_dresult* merged, const ptx_input_ops* in, ptx_dbatch** made, uint32_t* status_out);
/* Streaming append with `more` already resident (e.g. the output of ptx_change): log l of *out = log l of `base` + log l of `more`. */
ptx_status ptx_batch_append_device(ptx_ctx* ctx, const ptx_dbatch* base, const ptx_dbatch* more, ptx_dbatch** out);

/* Copy a resident batch back to the host (columns, envelope, headers; library-owned until ptx_host_batch_free). */
typedef struct ptx_host_batch {
    ptx_batch b;
    void* owner;
} ptx_host_batch;
ptx_status ptx_batch_download(ptx_ctx* ctx, const ptx_dbatch* b, ptx_host_batch* out);
void ptx_host_batch_free(ptx_host_batch* hb);

/* ---- introspection ---- */
/* Largest number of ops a log may have and still be merged ENTIRELY ON CHIP (the LDS kernel: one CU's 160 KB, 16-bit indices).  A longer log — the
 * reference has no bound, micromerge.ts:614-672 — is merged in the same ptx_merge call by the HBM-staged kernel (working set in device scratch the
 * library sizes per log, 32-bit indices; an order of magnitude slower per op), up to 2^26 rows and an id keyspace (max counter + 1) x (actors) of 2^30;
 * PTX_ERR_CAPACITY beyond that.  The editor-facing entry points follow (rounds 5-6): ptx_resolve_cursors on a log of any length; ptx_change too (its element list in
 * global scratch where one CU's LDS cannot hold it); ptx_replay_patches while the replay's bitmaps fit one CU's LDS (about 70 000 list elements, any number of
 * rows: a wide build with 32-bit ranks and boundary slots) — PTX_ERR_CAPACITY for a log beyond that. */
uint32_t ptx_max_ops_per_log(const ptx_ctx* ctx);
/* Name of the kernel the merge launches (to find it in a rocprofv3 trace): the family's ... */
const char* ptx_kernel_name(void);
/* ... and the build of it ptx_merge launches for THIS resident batch under this context: "ptx_merge_kernel", "ptx_merge_kernel_w7" (the same body held to the
 * scalar registers of seven waves per SIMD: launched when the batch's LDS window lets more than 24 waves share a CU) or "ptx_merge_kernel_many" (admission of
 * documents with more than three actors); a split batch's few large logs run as "..._rest*" / "ptx_merge_big_kernel" beside it. */
const char* ptx_batch_kernel_name(const ptx_ctx* ctx, const ptx_dbatch* b);

#ifdef __cplusplus
}
#endif
#endif /* PERITEXT_HIP_H */
