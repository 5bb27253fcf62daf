/*
 * Host emulation of the merge kernel's logic — TEST TOOLING ONLY (see merge_core.h header).
 *
 * Compiles peritext_amd/csrc/merge_core.h with -DPTX_EMU: every PTX_FOR becomes a sequential loop
 * (optionally reversed, to expose any dependence on iteration order), barriers vanish, atomics are
 * plain read-modify-writes.  Built into tests/emu/libperitext_emu.so by __graft_entry__.build() and
 * loaded ONLY by the CPU test-suite (tests/test_emu_*.py).  It is deliberately NOT part of
 * libperitext_hip.so and exports nothing the product ABI declares.
 */
#define PTX_EMU 1
#define PTX_PLATFORM_HEADER "../../tests/emu/ptx_platform_emu.h" /* resolved from peritext_amd/csrc/, where the #include stands */
#include <stdlib.h>
#include <string.h>
int ptx_emu_reverse = 0;
unsigned long long ptx_emu_exact_walks = 0;
extern "C" unsigned long long ptx_emu_exact_walk_count() { return ptx_emu_exact_walks; }
#include "../../peritext_amd/csrc/merge_core.h"
#include "../../peritext_amd/csrc/biglog_core.h"

/* LDS of the next log: not zero-initialised on the GPU either; a sanitizer build forgets the padding marks of the log before */
static size_t ptx_emu_lds_size = 0;
static void ptx_emu_lds_fill(uint8_t* lds, size_t bytes) {
    if (bytes) ptx_emu_lds_size = bytes;
#if defined(__SANITIZE_ADDRESS__)
    ASAN_UNPOISON_MEMORY_REGION(lds, ptx_emu_lds_size);
#endif
    if (bytes) memset(lds, 0xA5, bytes);
}
#include "../../peritext_amd/csrc/replay_core.h"
#include "../../peritext_amd/csrc/gen_core.h"
#include "../../peritext_amd/csrc/change_core.h"
#include "../../peritext_amd/csrc/cursor_core.h"
#include "../../peritext_amd/csrc/rootmap_core.h"

static int emu_merge_impl(const ptx_batch* b, ptx_log_result* res, uint32_t* values, ptx_span* spans, ptx_cinterval* cints, uint32_t* rank,
                          uint32_t lds_bytes, int reverse, int admission, uint32_t* refs, int lean = 0);

extern "C" int ptx_emu_merge(const ptx_batch* b, ptx_log_result* res, uint32_t* values, ptx_span* spans,
                             ptx_cinterval* cints, uint32_t* rank, uint32_t lds_bytes, int reverse) {
    return emu_merge_impl(b, res, values, spans, cints, rank, lds_bytes, reverse, 0, nullptr);
}
/* the same with causal admission over the batch's Change envelope (chg_* columns) */
extern "C" int ptx_emu_merge_admit(const ptx_batch* b, ptx_log_result* res, uint32_t* values, ptx_span* spans,
                                   ptx_cinterval* cints, uint32_t* rank, uint32_t lds_bytes, int reverse) {
    return emu_merge_impl(b, res, values, spans, cints, rank, lds_bytes, reverse, 1, nullptr);
}

/* the same with the resolved references of the delete / mark rows (what ptx_replay_patches reads beside elem_rank) */
extern "C" int ptx_emu_merge_refs(const ptx_batch* b, ptx_log_result* res, uint32_t* values, ptx_span* spans, ptx_cinterval* cints, uint32_t* rank,
                                  uint32_t* refs, uint32_t lds_bytes, int reverse, int admission) {
    return emu_merge_impl(b, res, values, spans, cints, rank, lds_bytes, reverse, admission, refs);
}

/* the LEAN build of the body (16-bit id keys, no elem_rank / resolved references: what ptx_merge_kernel_lean* are made of) for every log that qualifies */
extern "C" int ptx_emu_merge_lean(const ptx_batch* b, ptx_log_result* res, uint32_t* values, ptx_span* spans, ptx_cinterval* cints, uint32_t lds_bytes, int reverse,
                                  int admission) {
    return emu_merge_impl(b, res, values, spans, cints, nullptr, lds_bytes, reverse, admission, nullptr, 1);
}

/* (round 6) the high halves of the mark rows' boundary slots (PtxMergeArgs.out_refs_hi / ptx_dresult.refs_hi): a buffer the test sets before a merge through the
 * HBM-staged path and leaves set for the replay / change() calls that read it; NULL = the library's behaviour for a result without the column */
static uint32_t* ptx_emu_refs_hi = nullptr;
extern "C" void ptx_emu_set_refs_hi(uint32_t* p) { ptx_emu_refs_hi = p; }

static int emu_merge_impl(const ptx_batch* b, ptx_log_result* res, uint32_t* values, ptx_span* spans, ptx_cinterval* cints, uint32_t* rank,
                          uint32_t lds_bytes, int reverse, int admission, uint32_t* refs, int lean) {
    PtxMergeArgs A;
    memset(&A, 0, sizeof(A));
    A.log_off = b->log_off;
    A.op_id = b->op_id;
    A.ref_a = b->ref_a;
    A.ref_b = b->ref_b;
    A.payload = b->payload;
    A.action = b->action;
    A.mark_type = b->mark_type;
    A.side_a = b->side_a;
    A.side_b = b->side_b;
    A.chg_off = admission ? b->chg_off : nullptr;
    A.chg_hdr = b->chg_hdr;
    A.chg_env = b->chg_env;
    A.chg_env_hi = b->chg_env_hi;
    A.max_actors = b->max_actors;
    A.clocks = nullptr;
    A.stop_after = 0;
    A.div_magic = 0;
    A.log_index = nullptr;
    A.res = res;
    A.out_values = values;
    A.out_spans = spans;
    A.out_cints = cints;
    A.out_rank = rank;
    A.out_refs = refs;
    A.n_logs = b->n_logs;
    A.lds_bytes = lds_bytes;
    ptx_log_hdr* hdr = nullptr;
    if (b->log_hdr) {
        A.log_hdr = b->log_hdr;
    } else { /* what the library's census pre-pass does on the device */
        hdr = (ptx_log_hdr*)calloc(b->n_logs ? b->n_logs : 1, sizeof(ptx_log_hdr));
        for (uint32_t l = 0; l < b->n_logs; ++l) {
            const uint64_t b0 = b->log_off[l], b1 = b->log_off[l + 1];
            ptx_census_rows(b->op_id + b0, b->action + b0, b->mark_type + b0, b->payload + b0, b1 - b0, &hdr[l]);
        }
        A.log_hdr = hdr;
    }
    uint8_t* lds = (uint8_t*)aligned_alloc(64, (size_t)lds_bytes + 64);
    if (!lds) return 1;
    ptx_emu_reverse = reverse;
    for (uint32_t l = 0; l < b->n_logs; ++l) {
        ptx_emu_lds_fill(lds, lds_bytes); /* LDS is not zero-initialised on the GPU either */
        const uint64_t ks = ((uint64_t)A.log_hdr[l].max_counter + 1) * ((uint64_t)A.log_hdr[l].max_actor + 1);
        if (lean && !rank && !refs && ks <= 65536u && !(admission && b->max_actors > 3)) ptx_merge_log<0, 0, false, true>(A, l, lds); /* (the host's own rule: wants_lean) */
        else if (admission && b->max_actors >= 8u && b->max_actors <= 15u) ptx_merge_log<2, 0>(A, l, lds); /* (the host's own rule: launch_merge) */
        else ptx_merge_log<1, 0>(A, l, lds);
    }
    ptx_emu_lds_fill(lds, 0);
    free(lds);
    free(hdr);
    return 0;
}

/* the HBM-staged path for logs beyond one CU's LDS (biglog_core.h): every log of the batch through ptx_big_merge_log, its working set in a host buffer
 * sized by ptx_big_need (slack > 0 adds bytes the kernel must not need; < 0 takes some away: the log must then report PTX_ERR_CAPACITY) */
extern "C" int ptx_emu_merge_big(const ptx_batch* b, ptx_log_result* res, uint32_t* values, ptx_span* spans, ptx_cinterval* cints, uint32_t* rank, int reverse, int admission,
                                 long long slack, uint32_t* refs) {
    PtxMergeArgs A;
    memset(&A, 0, sizeof(A));
    A.out_refs = refs;
    A.out_refs_hi = refs ? ptx_emu_refs_hi : nullptr;
    A.log_off = b->log_off;
    A.op_id = b->op_id;
    A.ref_a = b->ref_a;
    A.ref_b = b->ref_b;
    A.payload = b->payload;
    A.action = b->action;
    A.mark_type = b->mark_type;
    A.side_a = b->side_a;
    A.side_b = b->side_b;
    A.chg_off = admission ? b->chg_off : nullptr;
    A.chg_hdr = b->chg_hdr;
    A.chg_env = b->chg_env;
    A.chg_env_hi = b->chg_env_hi;
    A.max_actors = b->max_actors;
    A.res = res;
    A.out_values = values;
    A.out_spans = spans;
    A.out_cints = cints;
    A.out_rank = rank;
    A.n_logs = b->n_logs;
    ptx_log_hdr* hdr = (ptx_log_hdr*)calloc(b->n_logs ? b->n_logs : 1, sizeof(ptx_log_hdr));
    for (uint32_t l = 0; l < b->n_logs; ++l) {
        const uint64_t b0 = b->log_off[l], b1 = b->log_off[l + 1];
        if (b->log_hdr) hdr[l] = b->log_hdr[l];
        else ptx_census_rows(b->op_id + b0, b->action + b0, b->mark_type + b0, b->payload + b0, b1 - b0, &hdr[l]);
    }
    A.log_hdr = hdr;
    ptx_emu_reverse = reverse;
    uint8_t* lds = (uint8_t*)aligned_alloc(64, 4096);
    for (uint32_t l = 0; l < b->n_logs; ++l) {
        const uint64_t C = A.chg_off ? A.chg_off[l + 1] - A.chg_off[l] : 0;
        const long long need = (long long)ptx_big_need(b->log_off[l + 1] - b->log_off[l], hdr[l], C, b->max_actors, PTX_NTHREADS) + slack;
        const uint64_t bytes = need > 64 ? (uint64_t)need : 64;
        uint8_t* win = (uint8_t*)aligned_alloc(64, (bytes + 63) & ~63ull);
        memset(win, 0xA5, bytes);
        memset(lds, 0xA5, 4096);
        /* (the same text either way: with the permuted loop order the header lives in the scratch slice, as in the cooperative multi-workgroup launch) */
        if (reverse == 2) ptx_big_merge_log<true>(A, l, win, bytes, nullptr);
        else ptx_big_merge_log<false>(A, l, win, bytes, lds);
        free(win);
    }
    free(lds);
    free(hdr);
    return 0;
}

/* LDS bound the host uses to size the launch (tests check it against the measured high-water mark) */
extern "C" uint64_t ptx_emu_lds_need(uint64_t N, uint64_t n, uint64_t D, uint64_t K, uint64_t Kc, uint64_t ks, uint64_t Kid) {
    return ptx_lds_need(N, n, D, K, Kc, ks, Kid);
}

/* patch-stream replay (replay_core.h) over the merge results `res` / `rank` of the same batch; patch_off = capacity
 * offsets [n_logs + 1] */
/* arena_cap > 0: `patches` holds arena_cap more records behind patch_off[n_logs] for the overflow extents; ext_off[n_logs] says where a log's extent starts (~0: none) */
extern "C" int ptx_emu_replay_arena(const ptx_batch* b, const ptx_log_result* res, const uint32_t* rank, const uint32_t* refs, const uint64_t* patch_off, ptx_patch* patches,
                                    ptx_patch_log* plogs, uint32_t lds_bytes, int reverse, const uint32_t* first_row /* NULL: whole streams */, uint64_t arena_cap, uint64_t* ext_off) {
    PtxReplayArgs A;
    unsigned long long arena_next = 0;
    A.first_row = first_row;
    A.arena_next = arena_cap ? &arena_next : nullptr;
    A.arena_base = b->n_logs ? patch_off[b->n_logs] : 0;
    A.arena_cap = arena_cap;
    A.ext_off = arena_cap ? ext_off : nullptr;
    A.log_off = b->log_off;
    A.op_id = b->op_id;
    A.ref_a = b->ref_a;
    A.ref_b = b->ref_b;
    A.payload = b->payload;
    A.action = b->action;
    A.mark_type = b->mark_type;
    A.side_a = b->side_a;
    A.side_b = b->side_b;
    A.res = res;
    A.elem_rank = rank;
    A.refs = refs;
    A.refs_hi = ptx_emu_refs_hi;
    A.patch_off = patch_off;
    A.patches = patches;
    A.plogs = plogs;
    A.n_logs = b->n_logs;
    A.lds_bytes = lds_bytes;
    ptx_log_hdr* hdr = nullptr;
    if (b->log_hdr) {
        A.log_hdr = b->log_hdr;
    } else {
        hdr = (ptx_log_hdr*)calloc(b->n_logs ? b->n_logs : 1, sizeof(ptx_log_hdr));
        for (uint32_t l = 0; l < b->n_logs; ++l) {
            const uint64_t b0 = b->log_off[l], b1 = b->log_off[l + 1];
            ptx_census_rows(b->op_id + b0, b->action + b0, b->mark_type + b0, b->payload + b0, b1 - b0, &hdr[l]);
        }
        A.log_hdr = hdr;
    }
    /* bit 8 of `reverse`: the per-slot link urls and the op tables in "global" memory, as the library does for working sets above 5.5 KB */
    bool gwin = (reverse & 256) != 0;
    reverse &= 255;
    /* the library's rule: a batch with a log beyond 16-bit ranks / slots / rows takes the wide build (its tables in global memory) — when the slots' high halves are there */
    bool wide = false, wide_slots = false;
    for (uint32_t l = 0; l < b->n_logs; ++l) {
        wide = wide || ptx_replay_wants_wide(b->log_off[l + 1] - b->log_off[l], A.log_hdr[l]);
        wide_slots = wide_slots || A.log_hdr[l].n_ins > 32766u;
    }
    if (wide && wide_slots && !ptx_emu_refs_hi) wide = false;
    gwin = gwin || wide;
    const uint64_t n_ops = b->n_logs ? b->log_off[b->n_logs] : 0;
    (void)n_ops;
    uint64_t* win_off = (uint64_t*)calloc((size_t)b->n_logs + 1, 8);
    for (uint32_t l = 0; l < b->n_logs; ++l) win_off[l + 1] = win_off[l] + ptx_replay_win_units_hdr(A.log_hdr[l], wide);
    A.win_off = win_off;
    A.win_scratch = gwin ? (uint16_t*)malloc(2 * win_off[b->n_logs] + 16) : nullptr;
    if (gwin) memset(A.win_scratch, 0xA5, 2 * win_off[b->n_logs] + 16);
    uint8_t* lds = (uint8_t*)aligned_alloc(64, (size_t)lds_bytes + 64);
    if (!lds) return 1;
    ptx_emu_reverse = reverse;
    for (uint32_t l = 0; l < b->n_logs; ++l) {
        ptx_emu_lds_fill(lds, lds_bytes);
        if (wide) ptx_replay_log<0, true, true>(A, l, lds);
        else if (gwin) ptx_replay_log<0, true>(A, l, lds);
        else ptx_replay_log<0, false>(A, l, lds);
    }
    ptx_emu_lds_fill(lds, 0);
    free(lds);
    free(hdr);
    free(A.win_scratch);
    free(win_off);
    return 0;
}
extern "C" int ptx_emu_replay_from(const ptx_batch* b, const ptx_log_result* res, const uint32_t* rank, const uint32_t* refs, const uint64_t* patch_off, ptx_patch* patches,
                                   ptx_patch_log* plogs, uint32_t lds_bytes, int reverse, const uint32_t* first_row /* NULL: whole streams */) {
    return ptx_emu_replay_arena(b, res, rank, refs, patch_off, patches, plogs, lds_bytes, reverse, first_row, 0, nullptr);
}
extern "C" int ptx_emu_replay(const ptx_batch* b, const ptx_log_result* res, const uint32_t* rank, const uint32_t* refs, const uint64_t* patch_off, ptx_patch* patches,
                              ptx_patch_log* plogs, uint32_t lds_bytes, int reverse) {
    return ptx_emu_replay_from(b, res, rank, refs, patch_off, patches, plogs, lds_bytes, reverse, nullptr);
}
extern "C" uint64_t ptx_emu_replay_lds_need(uint64_t n, uint64_t K, uint64_t Kc, uint64_t ks, uint64_t Kid) { return ptx_replay_lds_need(n, K, Kc, ks, Kid); }
extern "C" uint64_t ptx_emu_replay_lds_need_wide(uint64_t n, uint64_t K, uint64_t Kc, uint64_t Kid) { return ptx_replay_lds_need(n, K, Kc, 0, Kid, true, true); }

/* on-device change() / PTXGEN (gen_core.h): generate n_docs documents into caller-allocated capacity-layout columns
 * (rows_per_log rows per log, R logs per doc); the envelope is left at capacity stride, n_changes says how much is used */
extern "C" int ptx_emu_generate(PtxGenArgs* A, int reverse) {
    A->lds_bytes = 160 * 1024;
    A->ctab = calloc((size_t)A->n_docs * A->R * A->rows_per_log + 1, ptx_gen_change_bytes(A->R));
    A->known = (uint16_t*)calloc((size_t)A->n_docs * A->R * A->rows_per_log + 1, sizeof(uint16_t));
    uint8_t* lds = (uint8_t*)aligned_alloc(64, (size_t)A->lds_bytes + 64);
    if (!lds || !A->ctab || !A->known) return 1;
    ptx_emu_reverse = reverse;
    for (uint32_t d = 0; d < A->n_docs; ++d) {
        ptx_emu_lds_fill(lds, A->lds_bytes);
        if (A->R <= 4) ptx_gen_doc<0, 4>(*A, d, lds);
        else ptx_gen_doc<0, PTX_GEN_MAX_R>(*A, d, lds);
    }
    ptx_emu_lds_fill(lds, 0);
    free(lds);
    free(A->ctab);
    free(A->known);
    A->ctab = nullptr;
    A->known = nullptr;
    return 0;
}

/* change() for caller-supplied InputOperations (change_core.h) over the merge results `res` / `rank` of the base batch;
 * the caller allocates the capacity-layout output (out_off = rows per log, known from the InputOperations) */
extern "C" int ptx_emu_change(const ptx_batch* b, const ptx_log_result* res, const uint32_t* rank, const uint32_t* refs, const ptx_input_ops* in, const uint64_t* out_off, uint64_t* o_op_id,
                              uint64_t* o_ref_a, uint64_t* o_ref_b, uint32_t* o_payload, uint8_t* o_action, uint8_t* o_mark_type, uint8_t* o_side_a, uint8_t* o_side_b,
                              uint32_t* o_chg_hdr, uint16_t* o_chg_env, uint16_t* o_chg_env_hi, uint32_t* any_wide, uint32_t* status, uint32_t* rows_made, uint32_t* chgs_made,
                              uint32_t lds_bytes, int reverse) {
    PtxChangeArgs A;
    memset(&A, 0, sizeof(A));
    A.log_off = b->log_off;
    A.op_id = b->op_id;
    A.ref_a = b->ref_a;
    A.ref_b = b->ref_b;
    A.action = b->action;
    A.mark_type = b->mark_type;
    A.side_a = b->side_a;
    A.side_b = b->side_b;
    A.res = res;
    A.elem_rank = rank;
    A.refs = refs;
    A.refs_hi = ptx_emu_refs_hi;
    A.chg_off = b->chg_off;
    A.chg_hdr = b->chg_hdr;
    A.max_actors = in->max_actors;
    A.in_chg_off = in->chg_off;
    A.in_op_off = in->op_off;
    A.in_action = in->action;
    A.in_mark_type = in->mark_type;
    A.in_index = in->index;
    A.in_count = in->count;
    A.in_payload = in->payload;
    A.in_values = in->values;
    A.actor = in->actor;
    A.out_off = out_off;
    A.o_op_id = o_op_id;
    A.o_ref_a = o_ref_a;
    A.o_ref_b = o_ref_b;
    A.o_payload = o_payload;
    A.o_action = o_action;
    A.o_mark_type = o_mark_type;
    A.o_side_a = o_side_a;
    A.o_side_b = o_side_b;
    A.o_chg_hdr = o_chg_hdr;
    A.o_chg_env = o_chg_env;
    A.o_chg_env_hi = o_chg_env_hi;
    A.any_wide = any_wide;
    A.status = status;
    A.rows_made = rows_made;
    A.chgs_made = chgs_made;
    A.n_logs = b->n_logs;
    A.lds_bytes = lds_bytes;
    ptx_log_hdr* hdr = (ptx_log_hdr*)calloc(b->n_logs ? b->n_logs : 1, sizeof(ptx_log_hdr));
    for (uint32_t l = 0; l < b->n_logs; ++l) {
        const uint64_t b0 = b->log_off[l], b1 = b->log_off[l + 1];
        ptx_census_rows(b->op_id + b0, b->action + b0, b->mark_type + b0, b->payload + b0, b1 - b0, &hdr[l]);
    }
    A.log_hdr = hdr;
    /* the library's rule (ptx_change): the element list of a log that does not fit the LDS lives in a slice of global scratch */
    uint64_t* list_off = (uint64_t*)calloc((size_t)b->n_logs + 1, 8);
    for (uint32_t l = 0; l < b->n_logs; ++l) {
        list_off[l + 1] = list_off[l];
        uint64_t grow = 0;
        for (uint64_t q = in->op_off[in->chg_off[l]]; q < in->op_off[in->chg_off[l + 1]]; ++q) grow += in->action[q] == PTX_IN_INSERT ? in->count[q] : 0u;
        const uint64_t n_l = b->log_off[l + 1] > b->log_off[l] ? hdr[l].n_ins : 0;
        if (in->chg_off[l + 1] > in->chg_off[l] && ptx_change_lds_need(n_l, grow, 0, in->max_actors) > lds_bytes) list_off[l + 1] += ptx_change_list_words(n_l, grow);
    }
    uint32_t* list = list_off[b->n_logs] ? (uint32_t*)aligned_alloc(64, (4 * list_off[b->n_logs] + 63) & ~63ull) : nullptr;
    if (list) memset(list, 0xA5, 4 * list_off[b->n_logs]);
    A.list_scratch = list;
    A.list_off = list_off;
    uint8_t* lds = (uint8_t*)aligned_alloc(64, (size_t)lds_bytes + 64);
    if (!lds) return 1;
    ptx_emu_reverse = reverse;
    for (uint32_t l = 0; l < b->n_logs; ++l) {
        ptx_emu_lds_fill(lds, lds_bytes);
        ptx_change_log<0>(A, l, lds);
    }
    ptx_emu_lds_fill(lds, 0);
    free(lds);
    free(hdr);
    free(list);
    free(list_off);
    return 0;
}

/* cursor resolution (cursor_core.h) over the merge results of the batch; queries already grouped by log by the caller */
extern "C" int ptx_emu_cursors(const ptx_batch* b, const ptx_log_result* res, const uint32_t* rank, uint32_t n_groups, const uint32_t* g_log, const uint64_t* g_off,
                               const uint32_t* perm, const uint8_t* kind, const uint64_t* arg, uint64_t* out, uint32_t* status, uint32_t lds_bytes, int reverse) {
    PtxCursorArgs A;
    memset(&A, 0, sizeof(A));
    A.log_off = b->log_off;
    A.op_id = b->op_id;
    A.action = b->action;
    A.res = res;
    A.elem_rank = rank;
    A.q_group_log = g_log;
    A.q_group_off = g_off;
    A.q_perm = perm;
    A.q_kind = kind;
    A.q_arg = arg;
    A.out = out;
    A.status = status;
    A.n_groups = n_groups;
    A.lds_bytes = lds_bytes;
    ptx_log_hdr* hdr = (ptx_log_hdr*)calloc(b->n_logs ? b->n_logs : 1, sizeof(ptx_log_hdr));
    for (uint32_t l = 0; l < b->n_logs; ++l) {
        const uint64_t b0 = b->log_off[l], b1 = b->log_off[l + 1];
        ptx_census_rows(b->op_id + b0, b->action + b0, b->mark_type + b0, b->payload + b0, b1 - b0, &hdr[l]);
    }
    A.log_hdr = hdr;
    uint8_t* lds = (uint8_t*)aligned_alloc(64, (size_t)lds_bytes + 64);
    if (!lds) return 1;
    ptx_emu_reverse = reverse;
    for (uint32_t g = 0; g < n_groups; ++g) {
        ptx_emu_lds_fill(lds, lds_bytes);
        ptx_cursor_group<0>(A, g, lds);
    }
    ptx_emu_lds_fill(lds, 0);
    free(lds);
    free(hdr);
    return 0;
}

/* the map objects of a replica (rootmap_core.h): the caller sizes the entry rows (entry_off) by the map rows of every log */
extern "C" int ptx_emu_root_map(const ptx_batch* b, const uint64_t* entry_off, ptx_root_entry* entries, ptx_root_log* rlogs, uint32_t lds_bytes, int reverse) {
    PtxRootArgs A;
    A.log_off = b->log_off;
    A.op_id = b->op_id;
    A.ref_a = b->ref_a;
    A.ref_b = b->ref_b;
    A.payload = b->payload;
    A.action = b->action;
    A.mark_type = b->mark_type;
    A.entry_off = entry_off;
    A.entries = entries;
    A.rlogs = rlogs;
    A.n_logs = b->n_logs;
    A.lds_bytes = lds_bytes;
    uint8_t* lds = (uint8_t*)aligned_alloc(64, (((size_t)lds_bytes + 63) & ~(size_t)63) + 64);
    if (!lds) return 1;
    ptx_emu_reverse = reverse;
    for (uint32_t l = 0; l < b->n_logs; ++l) {
        ptx_emu_lds_fill(lds, lds_bytes);
        ptx_rootmap_log<0>(A, l, lds);
    }
    ptx_emu_lds_fill(lds, 0);
    free(lds);
    return 0;
}
