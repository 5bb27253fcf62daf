/*
 * addon.cc — N-API binding of the C ABI in include/peritext_hip.h (the TypeScript/JS host's door to the GPU).
 *
 * The reference has no FFI: bridge.ts:253/288 and the test harness (test/micromerge.ts:54-79) call
 * Micromerge.applyChange / getTextWithFormatting on JS objects.  This addon is what a maintainer adds instead
 * (INTEGRATION.md): index.js flattens `Change[]` into the SoA op-log columns, this file hands their ArrayBuffers
 * to ptx_apply_materialize and copies the canonical result rows back into typed arrays.  Nothing is computed
 * here; libperitext_hip.so is dlopen()ed at run time, so the addon builds with plain g++ and no ROCm headers:
 *     g++ -O2 -shared -fPIC -I/usr/include/node -Iinclude peritext_amd/node/addon.cc -ldl -o peritext_amd/node/peritext_node.node
 *
 * Exports:  open(libPath) -> abiVersion      create(device, flags) -> ctx (external)      destroy(ctx)
 *           applyMaterialize(ctx, batch) -> {logs:Uint32Array(12/log), values:Uint32Array, spans:Uint32Array(2/row),
 *                                            cintervals:Uint32Array(3/row), valueOff / spanOff / cintOff: BigUint64Array(nLogs + 1) (the rows are compact: ABI 7),
 *                                            elemRank:Uint32Array|null}
 *           generate(ctx, cfg)  change(ctx, batch, inputOps) -> {batch, status}  maxOpsPerLog(ctx)  kernelName()
 * Errors of the library surface as JS exceptions (Error with the library's message); per-LOG failures stay in
 * logs[12*l] (status) and are turned into RangeError by index.js, mirroring micromerge.ts:503,:507,:752.
 */
#include <dlfcn.h>
#include <node_api.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <string>
#include <vector>

#include "../../include/peritext_hip.h"

namespace {

struct Lib {
    void* handle = nullptr;
    uint32_t (*abi_version)(void) = nullptr;
    ptx_status (*create)(int, uint32_t, ptx_ctx**) = nullptr;
    void (*destroy)(ptx_ctx*) = nullptr;
    const char* (*last_error)(const ptx_ctx*) = nullptr;
    ptx_status (*apply_materialize)(ptx_ctx*, const ptx_batch*, ptx_result*) = nullptr;
    void (*result_free)(ptx_result*) = nullptr;
    uint32_t (*max_ops_per_log)(const ptx_ctx*) = nullptr;
    const char* (*kernel_name)(void) = nullptr;
    /* staged form, used when the caller also wants the Patch[] streams */
    ptx_status (*batch_upload)(ptx_ctx*, const ptx_batch*, ptx_dbatch**) = nullptr;
    void (*batch_free)(ptx_ctx*, ptx_dbatch*) = nullptr;
    ptx_status (*result_alloc)(ptx_ctx*, const ptx_dbatch*, ptx_dresult**) = nullptr;
    void (*dresult_free)(ptx_ctx*, ptx_dresult*) = nullptr;
    ptx_status (*merge)(ptx_ctx*, const ptx_dbatch*, ptx_dresult*) = nullptr;
    ptx_status (*sync)(ptx_ctx*) = nullptr;
    ptx_status (*result_download)(ptx_ctx*, const ptx_dbatch*, const ptx_dresult*, ptx_result*) = nullptr;
    ptx_status (*replay_patches)(ptx_ctx*, const ptx_dbatch*, const ptx_dresult*, ptx_patches*) = nullptr;
    /* resident replicas: the logs stay in HBM between calls, new Changes are appended, only their Patch[] records come back */
    ptx_status (*batch_append)(ptx_ctx*, const ptx_dbatch*, const ptx_batch*, ptx_dbatch**) = nullptr;
    ptx_status (*replay_patches_from)(ptx_ctx*, const ptx_dbatch*, const ptx_dresult*, const uint32_t*, ptx_patches*) = nullptr;
    uint32_t (*batch_n_logs)(const ptx_dbatch*) = nullptr;
    void (*patches_free)(ptx_patches*) = nullptr;
    /* on-device change(): op logs generated in HBM */
    ptx_status (*generate)(ptx_ctx*, const ptx_gen_config*, ptx_dbatch**, ptx_gen_info*) = nullptr;
    void (*gen_info_free)(ptx_gen_info*) = nullptr;
    ptx_status (*batch_download)(ptx_ctx*, const ptx_dbatch*, ptx_host_batch*) = nullptr;
    void (*host_batch_free)(ptx_host_batch*) = nullptr;
    /* change(): caller-supplied InputOperations */
    ptx_status (*change)(ptx_ctx*, const ptx_dbatch*, const ptx_dresult*, const ptx_input_ops*, ptx_dbatch**, uint32_t*) = nullptr;
    ptx_status (*batch_append_device)(ptx_ctx*, const ptx_dbatch*, const ptx_dbatch*, ptx_dbatch**) = nullptr;
    /* multi-GPU: the digest all-gather (RCCL inside the library) */
    ptx_status (*comm_unique_id)(ptx_ctx*, uint8_t*) = nullptr;
    ptx_status (*comm_init)(ptx_ctx*, const uint8_t*, uint32_t, uint32_t, ptx_comm**) = nullptr;
    void (*comm_destroy)(ptx_ctx*, ptx_comm*) = nullptr;
    uint32_t (*comm_n_ranks)(const ptx_comm*) = nullptr;
    ptx_status (*allgather_digests)(ptx_ctx*, ptx_comm*, const ptx_dresult*, const uint32_t*, uint64_t*) = nullptr;
    ptx_status (*count_converged_digests)(ptx_ctx*, const uint64_t*, uint64_t, uint32_t, uint64_t*) = nullptr;
    ptx_status (*result_download_logs)(ptx_ctx*, const ptx_dresult*, ptx_log_result*, uint32_t) = nullptr;
    ptx_status (*resolve_cursors)(ptx_ctx*, const ptx_dbatch*, const ptx_dresult*, uint32_t, const uint32_t*, const uint8_t*, const uint64_t*, uint64_t*, uint32_t*) = nullptr;
    /* the map objects of a replica: getRoot() */
    ptx_status (*root_map)(ptx_ctx*, const ptx_dbatch*, ptx_root_maps*) = nullptr;
    void (*root_maps_free)(ptx_root_maps*) = nullptr;
    ptx_status (*device_alloc)(ptx_ctx*, uint64_t, void**) = nullptr;
    void (*device_free)(ptx_ctx*, void*) = nullptr;
    ptx_status (*device_read)(ptx_ctx*, const void*, void*, uint64_t) = nullptr;
} L;

#define NAPI_OK(call)                                                        \
    do {                                                                     \
        if ((call) != napi_ok) {                                             \
            napi_throw_error(env, nullptr, "N-API call failed: " #call);     \
            return nullptr;                                                  \
        }                                                                    \
    } while (0)

napi_value throw_msg(napi_env env, const char* msg) {
    napi_throw_error(env, nullptr, msg);
    return nullptr;
}

template <class F>
bool sym(F& fn, const char* name) {
    fn = (F)dlsym(L.handle, name);
    return fn != nullptr;
}

napi_value Open(napi_env env, napi_callback_info info) {
    size_t argc = 1;
    napi_value argv[1];
    NAPI_OK(napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
    char path[4096];
    size_t len = 0;
    if (argc < 1 || napi_get_value_string_utf8(env, argv[0], path, sizeof(path), &len) != napi_ok) return throw_msg(env, "open(libPath): string expected");
    if (!L.handle) {
        L.handle = dlopen(path, RTLD_NOW | RTLD_LOCAL);
        if (!L.handle) {
            char msg[4600];
            snprintf(msg, sizeof(msg), "cannot load %s: %s (build it with __graft_entry__.build(); there is no CPU fallback)", path, dlerror());
            return throw_msg(env, msg);
        }
        bool ok = sym(L.abi_version, "ptx_abi_version") && sym(L.create, "ptx_create") && sym(L.destroy, "ptx_destroy") &&
                  sym(L.last_error, "ptx_last_error") && sym(L.apply_materialize, "ptx_apply_materialize") && sym(L.result_free, "ptx_result_free") &&
                  sym(L.max_ops_per_log, "ptx_max_ops_per_log") && sym(L.kernel_name, "ptx_kernel_name") && sym(L.batch_upload, "ptx_batch_upload") &&
                  sym(L.batch_free, "ptx_batch_free") && sym(L.result_alloc, "ptx_result_alloc") && sym(L.dresult_free, "ptx_dresult_free") &&
                  sym(L.merge, "ptx_merge") && sym(L.sync, "ptx_sync") && sym(L.result_download, "ptx_result_download") &&
                  sym(L.replay_patches, "ptx_replay_patches") && sym(L.batch_append, "ptx_batch_append") && sym(L.replay_patches_from, "ptx_replay_patches_from") &&
                  sym(L.batch_n_logs, "ptx_batch_n_logs") && sym(L.patches_free, "ptx_patches_free") && sym(L.generate, "ptx_generate") &&
                  sym(L.gen_info_free, "ptx_gen_info_free") && sym(L.batch_download, "ptx_batch_download") && sym(L.host_batch_free, "ptx_host_batch_free") && sym(L.change, "ptx_change") &&
                  sym(L.comm_unique_id, "ptx_comm_unique_id") && sym(L.comm_init, "ptx_comm_init") && sym(L.comm_destroy, "ptx_comm_destroy") && sym(L.comm_n_ranks, "ptx_comm_n_ranks") &&
                  sym(L.allgather_digests, "ptx_allgather_digests") && sym(L.count_converged_digests, "ptx_count_converged_digests") &&
                  sym(L.result_download_logs, "ptx_result_download_logs") && sym(L.root_map, "ptx_root_map") && sym(L.root_maps_free, "ptx_root_maps_free") && sym(L.device_alloc, "ptx_device_alloc") && sym(L.device_free, "ptx_device_free") &&
                  sym(L.device_read, "ptx_device_read") && sym(L.resolve_cursors, "ptx_resolve_cursors") && sym(L.batch_append_device, "ptx_batch_append_device");
        if (!ok) {
            dlclose(L.handle);
            L.handle = nullptr;
            return throw_msg(env, "libperitext_hip.so lacks a symbol include/peritext_hip.h declares");
        }
        if (L.abi_version() != PTX_ABI_VERSION) return throw_msg(env, "libperitext_hip.so ABI version mismatch");
    }
    napi_value v;
    NAPI_OK(napi_create_uint32(env, L.abi_version(), &v));
    return v;
}

napi_value Create(napi_env env, napi_callback_info info) {
    if (!L.handle) return throw_msg(env, "call open(libPath) first");
    size_t argc = 2;
    napi_value argv[2];
    NAPI_OK(napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
    int32_t device = 0;
    uint32_t flags = 0;
    if (argc > 0) napi_get_value_int32(env, argv[0], &device);
    if (argc > 1) napi_get_value_uint32(env, argv[1], &flags);
    ptx_ctx* ctx = nullptr;
    const ptx_status st = L.create(device, flags, &ctx);
    if (st != PTX_OK) {
        char msg[1024];
        snprintf(msg, sizeof(msg), "ptx_create failed (status %d): %s", st, L.last_error(nullptr));
        return throw_msg(env, msg);
    }
    napi_value ext;
    NAPI_OK(napi_create_external(env, ctx, nullptr, nullptr, &ext));
    return ext;
}

ptx_ctx* ctx_of(napi_env env, napi_value v) {
    void* p = nullptr;
    if (napi_get_value_external(env, v, &p) != napi_ok) return nullptr;
    return (ptx_ctx*)p;
}

napi_value Destroy(napi_env env, napi_callback_info info) {
    size_t argc = 1;
    napi_value argv[1];
    NAPI_OK(napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
    ptx_ctx* ctx = argc ? ctx_of(env, argv[0]) : nullptr;
    if (ctx && L.handle) L.destroy(ctx);
    return nullptr;
}

/* typed-array property -> raw pointer + element count; expected element size is checked */
bool column(napi_env env, napi_value obj, const char* name, size_t elem, const void** ptr, size_t* count, bool optional = false) {
    napi_value v;
    bool has = false;
    if (napi_has_named_property(env, obj, name, &has) != napi_ok || !has) {
        *ptr = nullptr;
        *count = 0;
        return optional;
    }
    if (napi_get_named_property(env, obj, name, &v) != napi_ok) return false;
    bool is_ta = false;
    napi_is_typedarray(env, v, &is_ta);
    if (!is_ta) {
        *ptr = nullptr;
        *count = 0;
        return optional;
    }
    napi_typedarray_type type;
    size_t length = 0, offset = 0;
    void* data = nullptr;
    napi_value ab;
    if (napi_get_typedarray_info(env, v, &type, &length, &data, &ab, &offset) != napi_ok) return false;
    size_t es = 0;
    switch (type) {
        case napi_uint8_array: es = 1; break;
        case napi_uint16_array: es = 2; break;
        case napi_uint32_array: es = 4; break;
        case napi_biguint64_array: es = 8; break;
        default: return false;
    }
    if (es != elem) return false;
    *ptr = data;
    *count = length;
    return true;
}

napi_value make_u32(napi_env env, const void* src, size_t count) {
    napi_value ab, ta;
    void* data = nullptr;
    if (napi_create_arraybuffer(env, count * 4, &data, &ab) != napi_ok) return nullptr;
    if (count) memcpy(data, src, count * 4);
    if (napi_create_typedarray(env, napi_uint32_array, count, ab, 0, &ta) != napi_ok) return nullptr;
    return ta;
}

/* JS WireBatch (typed arrays, index.d.ts) -> ptx_batch pointing into their ArrayBuffers; false + pending exception on a malformed one */
bool read_batch(napi_env env, napi_value b, ptx_batch* out) {
    ptx_batch& pb = *out;
    memset(&pb, 0, sizeof(pb));
    size_t n_off = 0, n = 0, m = 0;
    const void* p = nullptr;
    if (!column(env, b, "logOff", 8, &p, &n_off)) return throw_msg(env, "batch.logOff must be a BigUint64Array"), false;
    pb.log_off = (const uint64_t*)p;
    if (n_off == 0) return throw_msg(env, "batch.logOff needs n_logs + 1 entries"), false;
    pb.n_logs = (uint32_t)(n_off - 1);
    pb.n_ops = pb.log_off[pb.n_logs];
    struct Col { const char* name; size_t elem; const void** dst; } cols[] = {
        {"opId", 8, (const void**)&pb.op_id},     {"refA", 8, (const void**)&pb.ref_a},         {"refB", 8, (const void**)&pb.ref_b},
        {"payload", 4, (const void**)&pb.payload}, {"action", 1, (const void**)&pb.action},     {"markType", 1, (const void**)&pb.mark_type},
        {"sideA", 1, (const void**)&pb.side_a},    {"sideB", 1, (const void**)&pb.side_b},
    };
    for (const Col& c : cols) {
        if (!column(env, b, c.name, c.elem, c.dst, &n) || n != pb.n_ops) {
            char msg[128];
            snprintf(msg, sizeof(msg), "batch.%s: typed array of %zu-byte elements with n_ops entries expected", c.name, c.elem);
            return throw_msg(env, msg), false;
        }
    }
    /* optional: per-log census (sizeof(ptx_log_hdr) / 4 = 10 u32 per log) — otherwise the library computes it on the device */
    if (column(env, b, "logHdr", 4, &p, &m, true) && p) {
        if (m != (size_t)pb.n_logs * (sizeof(ptx_log_hdr) / 4)) return throw_msg(env, "batch.logHdr: Uint32Array with 10 entries per log expected"), false;
        pb.log_hdr = (const ptx_log_hdr*)p;
    }
    /* optional: the Change envelope -> causal admission on the device (chgOff + chgHdr + chgEnv + maxActors, or none) */
    {
        size_t n_co = 0, n_h = 0, n_e = 0;
        const void *co = nullptr, *ch = nullptr, *ce = nullptr;
        uint32_t max_actors = 0;
        napi_value mv;
        bool has = false;
        if (napi_has_named_property(env, b, "maxActors", &has) == napi_ok && has && napi_get_named_property(env, b, "maxActors", &mv) == napi_ok)
            napi_get_value_uint32(env, mv, &max_actors);
        if (max_actors && column(env, b, "chgOff", 8, &co, &n_co, true) && co && column(env, b, "chgHdr", 4, &ch, &n_h, true) && ch &&
            column(env, b, "chgEnv", 2, &ce, &n_e, true) && ce) {
            const uint64_t nc = n_co == (size_t)pb.n_logs + 1 ? ((const uint64_t*)co)[pb.n_logs] : ~0ull;
            if (nc != n_h || nc * PTX_ENV_STRIDE(max_actors) != n_e) return throw_msg(env, "batch.chg*: inconsistent Change envelope columns"), false;
            pb.chg_off = (const uint64_t*)co;
            pb.chg_hdr = (const uint32_t*)ch;
            pb.chg_env = (const uint16_t*)ce;
            pb.max_actors = max_actors;
            /* optional wide column (ptx_batch.chg_env_hi): the high halves of chgEnv's values, same shape */
            size_t n_hi = 0;
            const void* hi = nullptr;
            if (column(env, b, "chgEnvHi", 2, &hi, &n_hi, true) && hi) {
                if (n_hi != n_e) return throw_msg(env, "batch.chgEnvHi: Uint16Array of the shape of chgEnv expected"), false;
                pb.chg_env_hi = (const uint16_t*)hi;
            }
        }
    }
    return true;
}


napi_value make_typed(napi_env env, napi_typedarray_type type, size_t elem, const void* src, size_t count) {
    napi_value ab, ta;
    void* data = nullptr;
    if (napi_create_arraybuffer(env, count * elem, &data, &ab) != napi_ok) return nullptr;
    if (count && src) memcpy(data, src, count * elem);
    if (napi_create_typedarray(env, type, count, ab, 0, &ta) != napi_ok) return nullptr;
    return ta;
}

/* ptx_batch in host memory -> JS WireBatch columns (copies) */
napi_value batch_to_js(napi_env env, const ptx_batch& b) {
    napi_value batch, v;
    if (napi_create_object(env, &batch) != napi_ok) return nullptr;
    const size_t Lg = b.n_logs, T = (size_t)b.n_ops, NC = b.chg_off ? (size_t)b.chg_off[Lg] : 0;
    struct { const char* name; napi_typedarray_type type; size_t elem; const void* src; size_t count; } cols[] = {
        {"logOff", napi_biguint64_array, 8, b.log_off, Lg + 1}, {"opId", napi_biguint64_array, 8, b.op_id, T}, {"refA", napi_biguint64_array, 8, b.ref_a, T},
        {"refB", napi_biguint64_array, 8, b.ref_b, T}, {"payload", napi_uint32_array, 4, b.payload, T}, {"action", napi_uint8_array, 1, b.action, T},
        {"markType", napi_uint8_array, 1, b.mark_type, T}, {"sideA", napi_uint8_array, 1, b.side_a, T}, {"sideB", napi_uint8_array, 1, b.side_b, T},
        {"logHdr", napi_uint32_array, 4, b.log_hdr, Lg * (sizeof(ptx_log_hdr) / 4)}, {"chgOff", napi_biguint64_array, 8, b.chg_off, Lg + 1},
        {"chgHdr", napi_uint32_array, 4, b.chg_hdr, NC}, {"chgEnv", napi_uint16_array, 2, b.chg_env, NC * PTX_ENV_STRIDE(b.max_actors)},
        {"chgEnvHi", napi_uint16_array, 2, b.chg_env_hi, b.chg_env_hi ? NC * PTX_ENV_STRIDE(b.max_actors) : 0},
    };
    for (auto& col : cols) {
        if (!col.src && col.count) continue;
        v = make_typed(env, col.type, col.elem, col.src, col.count);
        if (v) napi_set_named_property(env, batch, col.name, v);
    }
    napi_create_uint32(env, b.max_actors, &v);
    napi_set_named_property(env, batch, "maxActors", v);
    napi_create_uint32(env, b.n_logs, &v);
    napi_set_named_property(env, batch, "nLogs", v);
    napi_create_double(env, (double)b.n_ops, &v);
    napi_set_named_property(env, batch, "nOps", v);
    return batch;
}

/* the rows of a ptx_result (compact since ABI 7: log l's values at [valueOff[l], valueOff[l + 1]) ...) as typed arrays on `obj` */
static void result_rows_to_js(napi_env env, const ptx_result& res, napi_value obj) {
    napi_value v;
    static_assert(sizeof(ptx_log_result) == 48, "ptx_log_result layout");
    const size_t nl = (size_t)res.n_logs, n_off = nl + 1;
    v = make_u32(env, res.logs, nl * 12);
    if (v) napi_set_named_property(env, obj, "logs", v);
    v = make_u32(env, res.values, (size_t)res.value_off[nl]);
    if (v) napi_set_named_property(env, obj, "values", v);
    v = make_u32(env, res.spans, (size_t)res.span_off[nl] * 2);
    if (v) napi_set_named_property(env, obj, "spans", v);
    v = make_u32(env, res.cintervals, (size_t)res.cint_off[nl] * 3);
    if (v) napi_set_named_property(env, obj, "cintervals", v);
    const char* names[3] = {"valueOff", "spanOff", "cintOff"};
    const uint64_t* offs[3] = {res.value_off, res.span_off, res.cint_off};
    for (int k = 0; k < 3; ++k) {
        napi_value ab, ta;
        void* data = nullptr;
        if (napi_create_arraybuffer(env, n_off * 8, &data, &ab) == napi_ok && napi_create_typedarray(env, napi_biguint64_array, n_off, ab, 0, &ta) == napi_ok) {
            memcpy(data, offs[k], n_off * 8);
            napi_set_named_property(env, obj, names[k], ta);
        }
    }
    if (res.elem_rank) {
        v = make_u32(env, res.elem_rank, (size_t)res.n_rows);
        if (v) napi_set_named_property(env, obj, "elemRank", v);
    }
}

/* ptx_result (+ the patch streams) -> the JS result object; both are freed here */
napi_value result_to_js(napi_env env, ptx_result& res, ptx_patches* patp) {
    napi_value out;
    NAPI_OK(napi_create_object(env, &out));
    napi_value v;
    result_rows_to_js(env, res, out);
    L.result_free(&res);
    if (patp) {
        ptx_patches& pat = *patp;
        static_assert(sizeof(ptx_patch) == 16 && sizeof(ptx_patch_log) == 8, "ptx_patch layout");
        napi_value ab, ta;
        void* data = nullptr;
        const size_t n_off = (size_t)pat.n_logs + 1;
        if (napi_create_arraybuffer(env, n_off * 8, &data, &ab) == napi_ok && napi_create_typedarray(env, napi_biguint64_array, n_off, ab, 0, &ta) == napi_ok) {
            memcpy(data, pat.patch_off, n_off * 8);
            napi_set_named_property(env, out, "patchOff", ta);
        }
        v = make_u32(env, pat.logs, (size_t)pat.n_logs * 2);
        if (v) napi_set_named_property(env, out, "patchLogs", v);
        v = make_u32(env, pat.patches, (size_t)pat.patch_off[pat.n_logs] * 4);
        if (v) napi_set_named_property(env, out, "patches", v);
        L.patches_free(&pat);
    }
    return out;
}

napi_value ApplyMaterialize(napi_env env, napi_callback_info info) {
    if (!L.handle) return throw_msg(env, "call open(libPath) first");
    size_t argc = 3;
    napi_value argv[3];
    NAPI_OK(napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
    if (argc < 2) return throw_msg(env, "applyMaterialize(ctx, batch[, wantPatches])");
    bool want_patches = false;
    if (argc > 2) napi_get_value_bool(env, argv[2], &want_patches);
    ptx_ctx* ctx = ctx_of(env, argv[0]);
    if (!ctx) return throw_msg(env, "applyMaterialize: bad context");
    ptx_batch pb;
    if (!read_batch(env, argv[1], &pb)) return nullptr;
    ptx_result res;
    ptx_patches pat;
    memset(&pat, 0, sizeof(pat));
    ptx_status st;
    if (!want_patches) {
        st = L.apply_materialize(ctx, &pb, &res);
    } else {
        /* the same in stages, the results stay in HBM for the replay that produces what every applyChange returns */
        ptx_dbatch* db = nullptr;
        ptx_dresult* dr = nullptr;
        st = L.batch_upload(ctx, &pb, &db);
        if (st == PTX_OK) st = L.result_alloc(ctx, db, &dr);
        if (st == PTX_OK) st = L.merge(ctx, db, dr);
        if (st == PTX_OK) st = L.sync(ctx);
        if (st == PTX_OK) st = L.result_download(ctx, db, dr, &res);
        if (st == PTX_OK) {
            st = L.replay_patches(ctx, db, dr, &pat);
            if (st != PTX_OK) L.result_free(&res);
        }
        if (dr) L.dresult_free(ctx, dr);
        if (db) L.batch_free(ctx, db);
    }
    if (st != PTX_OK) {
        char msg[1024];
        snprintf(msg, sizeof(msg), "ptx_apply_materialize failed (status %d): %s", st, L.last_error(ctx));
        return throw_msg(env, msg);
    }
    return result_to_js(env, res, want_patches ? &pat : nullptr);
}


/* ---- resident replicas (INTEGRATION.md "Resident replicas"): residentUpload(ctx, batch) -> handle; residentAppend(ctx, handle, more) -> new handle (the
 *      old one is released); residentApply(ctx, handle, wantPatches[, firstRow: Uint32Array]) -> the result object of applyMaterialize, the patch streams
 *      from firstRow[l] on; residentFree(ctx, handle) ---- */
ptx_dbatch* dbatch_of(napi_env env, napi_value v) {
    void* p = nullptr;
    if (napi_get_value_external(env, v, &p) != napi_ok) return nullptr;
    return (ptx_dbatch*)p;
}
napi_value lib_error(napi_env env, ptx_ctx* ctx, const char* what, ptx_status st) {
    char msg[1024];
    snprintf(msg, sizeof(msg), "%s failed (status %d): %s", what, st, L.last_error(ctx));
    return throw_msg(env, msg);
}
napi_value ResidentUpload(napi_env env, napi_callback_info info) {
    if (!L.handle) return throw_msg(env, "call open(libPath) first");
    size_t argc = 2;
    napi_value argv[2];
    NAPI_OK(napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
    ptx_ctx* ctx = argc > 1 ? ctx_of(env, argv[0]) : nullptr;
    if (!ctx) return throw_msg(env, "residentUpload(ctx, batch)");
    ptx_batch pb;
    if (!read_batch(env, argv[1], &pb)) return nullptr;
    ptx_dbatch* db = nullptr;
    const ptx_status st = L.batch_upload(ctx, &pb, &db);
    if (st != PTX_OK) return lib_error(env, ctx, "ptx_batch_upload", st);
    napi_value ext;
    NAPI_OK(napi_create_external(env, db, nullptr, nullptr, &ext));
    return ext;
}
napi_value ResidentAppend(napi_env env, napi_callback_info info) {
    if (!L.handle) return throw_msg(env, "call open(libPath) first");
    size_t argc = 3;
    napi_value argv[3];
    NAPI_OK(napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
    ptx_ctx* ctx = argc > 2 ? ctx_of(env, argv[0]) : nullptr;
    ptx_dbatch* base = ctx ? dbatch_of(env, argv[1]) : nullptr;
    if (!base) return throw_msg(env, "residentAppend(ctx, handle, moreBatch)");
    ptx_batch pb;
    if (!read_batch(env, argv[2], &pb)) return nullptr;
    ptx_dbatch* db = nullptr;
    const ptx_status st = L.batch_append(ctx, base, &pb, &db);
    if (st != PTX_OK) return lib_error(env, ctx, "ptx_batch_append", st);
    L.batch_free(ctx, base);
    napi_value ext;
    NAPI_OK(napi_create_external(env, db, nullptr, nullptr, &ext));
    return ext;
}
napi_value ResidentApply(napi_env env, napi_callback_info info) {
    if (!L.handle) return throw_msg(env, "call open(libPath) first");
    size_t argc = 4;
    napi_value argv[4];
    NAPI_OK(napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
    ptx_ctx* ctx = argc > 1 ? ctx_of(env, argv[0]) : nullptr;
    ptx_dbatch* db = ctx ? dbatch_of(env, argv[1]) : nullptr;
    if (!db) return throw_msg(env, "residentApply(ctx, handle[, wantPatches[, firstRow]])");
    bool want_patches = false;
    if (argc > 2) napi_get_value_bool(env, argv[2], &want_patches);
    const uint32_t* first = nullptr;
    if (argc > 3) {
        bool is_ta = false;
        napi_is_typedarray(env, argv[3], &is_ta);
        if (is_ta) {
            napi_typedarray_type type;
            size_t length = 0, offset = 0;
            void* data = nullptr;
            napi_value ab;
            if (napi_get_typedarray_info(env, argv[3], &type, &length, &data, &ab, &offset) != napi_ok || type != napi_uint32_array || length != L.batch_n_logs(db))
                return throw_msg(env, "residentApply: firstRow must be a Uint32Array with one entry per replica log");
            first = (const uint32_t*)data;
        }
    }
    ptx_dresult* dr = nullptr;
    ptx_result res;
    ptx_patches pat;
    memset(&pat, 0, sizeof(pat));
    ptx_status st = L.result_alloc(ctx, db, &dr);
    if (st == PTX_OK) st = L.merge(ctx, db, dr);
    if (st == PTX_OK) st = L.sync(ctx);
    if (st == PTX_OK) st = L.result_download(ctx, db, dr, &res);
    if (st == PTX_OK && want_patches) {
        st = L.replay_patches_from(ctx, db, dr, first, &pat);
        if (st != PTX_OK) L.result_free(&res);
    }
    if (dr) L.dresult_free(ctx, dr);
    if (st != PTX_OK) return lib_error(env, ctx, "residentApply (ptx_merge / ptx_replay_patches_from)", st);
    return result_to_js(env, res, want_patches ? &pat : nullptr);
}
napi_value ResidentFree(napi_env env, napi_callback_info info) {
    size_t argc = 2;
    napi_value argv[2];
    NAPI_OK(napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
    ptx_ctx* ctx = argc > 1 ? ctx_of(env, argv[0]) : nullptr;
    ptx_dbatch* db = ctx ? dbatch_of(env, argv[1]) : nullptr;
    if (db && L.handle) L.batch_free(ctx, db);
    return nullptr;
}

uint32_t u32_prop(napi_env env, napi_value obj, const char* name, uint32_t dflt) {
    napi_value v;
    bool has = false;
    uint32_t out = dflt;
    if (napi_has_named_property(env, obj, name, &has) == napi_ok && has && napi_get_named_property(env, obj, name, &v) == napi_ok) napi_get_value_uint32(env, v, &out);
    return out;
}

/* generate(ctx, cfg): ptx_generate (on-device change(), the fuzzer workload) + merge of what it made, all resident; returns the
 * op-log columns (ptx_batch_download), the comment-id counts per document and the merge result */
napi_value Generate(napi_env env, napi_callback_info info) {
    if (!L.handle) return throw_msg(env, "call open(libPath) first");
    size_t argc = 2;
    napi_value argv[2];
    NAPI_OK(napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
    ptx_ctx* ctx = argc > 1 ? ctx_of(env, argv[0]) : nullptr;
    if (!ctx) return throw_msg(env, "generate(ctx, cfg)");
    napi_value c = argv[1];
    ptx_gen_config cfg;
    memset(&cfg, 0, sizeof(cfg));
    cfg.replicas = u32_prop(env, c, "replicas", 1);
    cfg.ops_per_log = u32_prop(env, c, "opsPerLog", 0);
    cfg.seed = u32_prop(env, c, "seed", 1);
    cfg.first_doc = u32_prop(env, c, "firstDoc", 0);
    cfg.n_docs = u32_prop(env, c, "nDocs", 1);
    cfg.list_cap = u32_prop(env, c, "listCap", 0);
    napi_value arr;
    uint32_t len = 0;
    if (napi_get_named_property(env, c, "mix", &arr) != napi_ok || napi_get_array_length(env, arr, &len) != napi_ok || len != 4) return throw_msg(env, "cfg.mix: [insert, delete, addMark, removeMark] percent");
    for (uint32_t i = 0; i < 4; ++i) {
        napi_value e;
        napi_get_element(env, arr, i, &e);
        napi_get_value_uint32(env, e, &cfg.mix[i]);
    }
    if (napi_get_named_property(env, c, "markTypes", &arr) != napi_ok || napi_get_array_length(env, arr, &len) != napi_ok || len > 4) return throw_msg(env, "cfg.markTypes: up to four PTX_MARK_* codes");
    cfg.n_mark_types = len;
    for (uint32_t i = 0; i < len; ++i) {
        napi_value e;
        uint32_t t = 0;
        napi_get_element(env, arr, i, &e);
        napi_get_value_uint32(env, e, &t);
        cfg.mark_types[i] = (uint8_t)t;
    }
    {
        napi_value tv;
        bool has = false;
        size_t tl = 0;
        if (napi_has_named_property(env, c, "initialText", &has) == napi_ok && has && napi_get_named_property(env, c, "initialText", &tv) == napi_ok)
            napi_get_value_string_utf8(env, tv, cfg.initial_text, sizeof(cfg.initial_text), &tl);
    }
    ptx_dbatch* db = nullptr;
    ptx_dresult* dr = nullptr;
    ptx_gen_info gi;
    ptx_host_batch hb;
    ptx_result res;
    memset(&hb, 0, sizeof(hb));
    ptx_status st = L.generate(ctx, &cfg, &db, &gi);
    if (st != PTX_OK) {
        char msg[1024];
        snprintf(msg, sizeof(msg), "ptx_generate failed (status %d): %s", st, L.last_error(ctx));
        return throw_msg(env, msg);
    }
    st = L.batch_download(ctx, db, &hb);
    if (st == PTX_OK) st = L.result_alloc(ctx, db, &dr);
    if (st == PTX_OK) st = L.merge(ctx, db, dr);
    if (st == PTX_OK) st = L.sync(ctx);
    bool have_res = false;
    if (st == PTX_OK) {
        st = L.result_download(ctx, db, dr, &res);
        have_res = st == PTX_OK;
    }
    if (dr) L.dresult_free(ctx, dr);
    L.batch_free(ctx, db);
    if (st != PTX_OK) {
        char msg[1024];
        snprintf(msg, sizeof(msg), "generate: merge of the generated batch failed (status %d): %s", st, L.last_error(ctx));
        if (hb.owner) L.host_batch_free(&hb);
        L.gen_info_free(&gi);
        return throw_msg(env, msg);
    }
    napi_value out, batch, result, v;
    NAPI_OK(napi_create_object(env, &out));
    NAPI_OK(napi_create_object(env, &result));
    batch = batch_to_js(env, hb.b);
    if (!batch) return throw_msg(env, "generate: cannot build the batch object");
    napi_set_named_property(env, out, "batch", batch);
    v = make_u32(env, gi.n_comments, gi.n_docs);
    if (v) napi_set_named_property(env, out, "nComments", v);
    napi_create_double(env, (double)gi.kernel_ms, &v);
    napi_set_named_property(env, out, "kernelMs", v);
    if (have_res) {
        result_rows_to_js(env, res, result);
        L.result_free(&res);
        napi_set_named_property(env, out, "result", result);
    }
    L.host_batch_free(&hb);
    L.gen_info_free(&gi);
    return out;
}

/* change(ctx, batch, inputOps): Micromerge.change for many replicas (ptx_change).  `batch` = the replica logs applied so far
 * (WireBatch with the Change envelope), inputOps = {chgOff, opOff: BigUint64Array, action, markType: Uint8Array, index, count,
 * payload, values, actor: Uint32Array, maxActors}.  Upload, merge, ptx_change, download of what was made.
 * Returns {batch: WireBatch columns of the new Changes only, status: Uint32Array per log}. */
napi_value Change(napi_env env, napi_callback_info info) {
    if (!L.handle) return throw_msg(env, "call open(libPath) first");
    size_t argc = 3;
    napi_value argv[3];
    NAPI_OK(napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
    ptx_ctx* ctx = argc > 2 ? ctx_of(env, argv[0]) : nullptr;
    if (!ctx) return throw_msg(env, "change(ctx, batch | resident handle, inputOps)");
    /* the replicas: a host batch (encoded, uploaded and merged here), or the handle of a RESIDENT batch (residentUpload / residentAppend): then nothing of the
     * document goes up — the InputOperations are resolved against the logs in HBM, the Changes made are appended to them on the device and the new handle
     * comes back beside the made rows (the old one is released) */
    napi_valuetype vt;
    const bool resident = napi_typeof(env, argv[1], &vt) == napi_ok && vt == napi_external;
    ptx_batch pb;
    memset(&pb, 0, sizeof(pb));
    ptx_dbatch* rdb = resident ? dbatch_of(env, argv[1]) : nullptr;
    if (resident) {
        if (!rdb) return throw_msg(env, "change: bad resident handle");
        pb.n_logs = L.batch_n_logs(rdb);
    } else if (!read_batch(env, argv[1], &pb)) return nullptr;
    napi_value io = argv[2];
    ptx_input_ops in;
    memset(&in, 0, sizeof(in));
    size_t n_co = 0, n_oo = 0, n = 0, n_ops = 0, n_act = 0;
    const void* p = nullptr;
    if (!column(env, io, "chgOff", 8, &p, &n_co) || n_co != (size_t)pb.n_logs + 1) return throw_msg(env, "inputOps.chgOff: BigUint64Array with n_logs + 1 entries expected");
    in.chg_off = (const uint64_t*)p;
    if (!column(env, io, "opOff", 8, &p, &n_oo) || n_oo != (size_t)in.chg_off[pb.n_logs] + 1) return throw_msg(env, "inputOps.opOff: BigUint64Array with n_changes + 1 entries expected");
    in.op_off = (const uint64_t*)p;
    n_ops = (size_t)in.op_off[n_oo - 1];
    struct Col { const char* name; size_t elem; const void** dst; } cols[] = {
        {"action", 1, (const void**)&in.action}, {"markType", 1, (const void**)&in.mark_type}, {"index", 4, (const void**)&in.index},
        {"count", 4, (const void**)&in.count},   {"payload", 4, (const void**)&in.payload},
    };
    for (const Col& c : cols)
        if (!column(env, io, c.name, c.elem, c.dst, &n) || n != n_ops) return throw_msg(env, "inputOps: a column does not have one entry per InputOperation");
    if (!column(env, io, "values", 4, &p, &n)) return throw_msg(env, "inputOps.values must be a Uint32Array");
    in.values = (const uint32_t*)p;
    in.n_values = n;
    if (!column(env, io, "actor", 4, &p, &n_act) || n_act != pb.n_logs) return throw_msg(env, "inputOps.actor: Uint32Array with one entry per log expected");
    in.actor = (const uint32_t*)p;
    in.n_logs = pb.n_logs;
    in.max_actors = u32_prop(env, io, "maxActors", pb.max_actors);
    if (resident && in.max_actors == 0) return throw_msg(env, "change on a resident handle: inputOps.maxActors is needed");
    ptx_dbatch *db = nullptr, *made = nullptr;
    ptx_dresult* dr = nullptr;
    ptx_host_batch hb;
    memset(&hb, 0, sizeof(hb));
    napi_value status_arr = nullptr;
    void* status_data = nullptr;
    {
        napi_value ab;
        if (napi_create_arraybuffer(env, (size_t)pb.n_logs * 4, &status_data, &ab) != napi_ok ||
            napi_create_typedarray(env, napi_uint32_array, pb.n_logs, ab, 0, &status_arr) != napi_ok)
            return throw_msg(env, "change: cannot allocate the status array");
    }
    ptx_dbatch* after = nullptr;
    ptx_status st = resident ? PTX_OK : L.batch_upload(ctx, &pb, &db);
    if (resident) db = rdb;
    if (st == PTX_OK) st = L.result_alloc(ctx, db, &dr);
    if (st == PTX_OK) st = L.merge(ctx, db, dr);
    if (st == PTX_OK) st = L.sync(ctx);
    if (st == PTX_OK) st = L.change(ctx, db, dr, &in, &made, (uint32_t*)status_data);
    if (st == PTX_OK) st = L.batch_download(ctx, made, &hb);
    if (st == PTX_OK && resident) st = L.batch_append_device(ctx, db, made, &after); /* the replicas after the edit: still resident, nothing re-uploaded */
    if (dr) L.dresult_free(ctx, dr);
    if (db && !resident) L.batch_free(ctx, db);
    if (made) L.batch_free(ctx, made);
    if (st != PTX_OK) {
        char msg[1024];
        snprintf(msg, sizeof(msg), "ptx_change failed (status %d): %s", st, L.last_error(ctx));
        if (hb.owner) L.host_batch_free(&hb);
        return throw_msg(env, msg);
    }
    napi_value out, batch = batch_to_js(env, hb.b);
    L.host_batch_free(&hb);
    if (!batch || napi_create_object(env, &out) != napi_ok) {
        if (after) L.batch_free(ctx, after); /* the session keeps its old handle (rdb): still valid */
        return throw_msg(env, "change: cannot build the result object");
    }
    napi_set_named_property(env, out, "batch", batch);
    napi_set_named_property(env, out, "status", status_arr);
    if (resident) {
        /* the new handle first, the old batch only once the result object holds it: a failure here must leave the session with a live handle */
        napi_value ext;
        if (napi_create_external(env, after, nullptr, nullptr, &ext) != napi_ok || napi_set_named_property(env, out, "handle", ext) != napi_ok) {
            L.batch_free(ctx, after);
            return throw_msg(env, "change: cannot hand out the resident batch");
        }
        L.batch_free(ctx, rdb); /* superseded by `after` */
    }
    return out;
}

/* rootMap(ctx, batch): Micromerge.getRoot() for every replica log of the batch (ptx_root_map): upload, resolve.  Returns
 * {entryOff: BigUint64Array [n_logs + 1], logs: Uint32Array (status, n_entries, first_bad_row, 0 per log), entries: Uint32Array
 * (obj lo, obj hi, key, row, kind, value per entry)}. */
napi_value RootMap(napi_env env, napi_callback_info info) {
    if (!L.handle) return throw_msg(env, "call open(libPath) first");
    size_t argc = 2;
    napi_value argv[2];
    NAPI_OK(napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
    if (argc < 2) return throw_msg(env, "rootMap(ctx, batch)");
    ptx_ctx* ctx = ctx_of(env, argv[0]);
    if (!ctx) return throw_msg(env, "rootMap: bad context");
    ptx_batch pb;
    if (!read_batch(env, argv[1], &pb)) return nullptr;
    ptx_dbatch* db = nullptr;
    ptx_root_maps rm;
    memset(&rm, 0, sizeof(rm));
    ptx_status st = L.batch_upload(ctx, &pb, &db);
    if (st == PTX_OK) st = L.root_map(ctx, db, &rm);
    if (db) L.batch_free(ctx, db);
    if (st != PTX_OK) {
        char msg[1024];
        snprintf(msg, sizeof(msg), "ptx_root_map failed (status %d): %s", st, L.last_error(ctx));
        return throw_msg(env, msg);
    }
    static_assert(sizeof(ptx_root_entry) == 24 && sizeof(ptx_root_log) == 16, "ptx_root_entry / ptx_root_log layout");
    napi_value out, v, ab, ta;
    NAPI_OK(napi_create_object(env, &out));
    void* data = nullptr;
    const size_t n_off = (size_t)rm.n_logs + 1;
    if (napi_create_arraybuffer(env, n_off * 8, &data, &ab) == napi_ok && napi_create_typedarray(env, napi_biguint64_array, n_off, ab, 0, &ta) == napi_ok) {
        memcpy(data, rm.entry_off, n_off * 8);
        napi_set_named_property(env, out, "entryOff", ta);
    }
    v = make_u32(env, rm.logs, (size_t)rm.n_logs * 4);
    if (v) napi_set_named_property(env, out, "logs", v);
    v = make_u32(env, rm.entries, (size_t)rm.entry_off[rm.n_logs] * 6);
    if (v) napi_set_named_property(env, out, "entries", v);
    L.root_maps_free(&rm);
    return out;
}

/* cursors(ctx, batch, {log: Uint32Array, kind: Uint8Array, arg: BigUint64Array}): Micromerge.getCursor / resolveCursor for many replicas
 * (ptx_resolve_cursors): upload, merge, resolve.  Returns {out: BigUint64Array, status: Uint32Array}, one entry per query. */
napi_value Cursors(napi_env env, napi_callback_info info) {
    if (!L.handle) return throw_msg(env, "call open(libPath) first");
    size_t argc = 3;
    napi_value argv[3];
    NAPI_OK(napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
    ptx_ctx* ctx = argc > 2 ? ctx_of(env, argv[0]) : nullptr;
    if (!ctx) return throw_msg(env, "cursors(ctx, batch | resident handle, queries)");
    napi_valuetype vt;
    const bool resident = napi_typeof(env, argv[1], &vt) == napi_ok && vt == napi_external; /* the logs already in HBM: nothing is encoded or uploaded */
    ptx_batch pb;
    memset(&pb, 0, sizeof(pb));
    ptx_dbatch* rdb = resident ? dbatch_of(env, argv[1]) : nullptr;
    if (resident && !rdb) return throw_msg(env, "cursors: bad resident handle");
    if (!resident && !read_batch(env, argv[1], &pb)) return nullptr;
    const void *ql = nullptr, *qk = nullptr, *qa = nullptr;
    size_t n = 0, n2 = 0, n3 = 0;
    if (!column(env, argv[2], "log", 4, &ql, &n) || !column(env, argv[2], "kind", 1, &qk, &n2) || !column(env, argv[2], "arg", 8, &qa, &n3) || n != n2 || n != n3)
        return throw_msg(env, "cursors: queries = {log: Uint32Array, kind: Uint8Array, arg: BigUint64Array} of one length");
    std::vector<uint64_t> out(n ? n : 1);
    std::vector<uint32_t> status(n ? n : 1);
    ptx_dbatch* db = nullptr;
    ptx_dresult* dr = nullptr;
    ptx_status st = resident ? PTX_OK : L.batch_upload(ctx, &pb, &db);
    if (resident) db = rdb;
    if (st == PTX_OK) st = L.result_alloc(ctx, db, &dr);
    if (st == PTX_OK) st = L.merge(ctx, db, dr);
    if (st == PTX_OK) st = L.sync(ctx);
    if (st == PTX_OK) st = L.resolve_cursors(ctx, db, dr, (uint32_t)n, (const uint32_t*)ql, (const uint8_t*)qk, (const uint64_t*)qa, out.data(), status.data());
    std::string err = st != PTX_OK ? L.last_error(ctx) : "";
    if (dr) L.dresult_free(ctx, dr);
    if (db && !resident) L.batch_free(ctx, db);
    if (st != PTX_OK) return throw_msg(env, ("ptx_resolve_cursors failed: " + err).c_str());
    napi_value o, v;
    NAPI_OK(napi_create_object(env, &o));
    v = make_typed(env, napi_biguint64_array, 8, out.data(), n);
    if (v) napi_set_named_property(env, o, "out", v);
    v = make_u32(env, status.data(), n);
    if (v) napi_set_named_property(env, o, "status", v);
    return o;
}

/* commUniqueId(ctx) -> Uint8Array(128): rank 0 makes it, the host's own channel carries it to the other ranks */
napi_value CommUniqueId(napi_env env, napi_callback_info info) {
    if (!L.handle) return throw_msg(env, "call open(libPath) first");
    size_t argc = 1;
    napi_value argv[1];
    NAPI_OK(napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
    ptx_ctx* ctx = argc ? ctx_of(env, argv[0]) : nullptr;
    if (!ctx) return throw_msg(env, "commUniqueId(ctx)");
    uint8_t id[PTX_COMM_ID_BYTES];
    if (L.comm_unique_id(ctx, id) != PTX_OK) return throw_msg(env, L.last_error(ctx));
    return make_typed(env, napi_uint8_array, 1, id, PTX_COMM_ID_BYTES);
}

/* commInit(ctx, id: Uint8Array(128), rank, nRanks) -> comm (external); collective over all ranks */
napi_value CommInit(napi_env env, napi_callback_info info) {
    if (!L.handle) return throw_msg(env, "call open(libPath) first");
    size_t argc = 4;
    napi_value argv[4];
    NAPI_OK(napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
    ptx_ctx* ctx = argc > 3 ? ctx_of(env, argv[0]) : nullptr;
    if (!ctx) return throw_msg(env, "commInit(ctx, id, rank, nRanks)");
    napi_typedarray_type type;
    size_t length = 0, offset = 0;
    void* data = nullptr;
    napi_value ab;
    if (napi_get_typedarray_info(env, argv[1], &type, &length, &data, &ab, &offset) != napi_ok || type != napi_uint8_array || length != PTX_COMM_ID_BYTES)
        return throw_msg(env, "commInit: id must be the Uint8Array(128) commUniqueId() made on rank 0");
    uint32_t rank = 0, n = 1;
    napi_get_value_uint32(env, argv[2], &rank);
    napi_get_value_uint32(env, argv[3], &n);
    ptx_comm* comm = nullptr;
    if (L.comm_init(ctx, (const uint8_t*)data, rank, n, &comm) != PTX_OK) return throw_msg(env, L.last_error(ctx));
    napi_value ext;
    NAPI_OK(napi_create_external(env, comm, nullptr, nullptr, &ext));
    return ext;
}

napi_value CommDestroy(napi_env env, napi_callback_info info) {
    size_t argc = 2;
    napi_value argv[2];
    NAPI_OK(napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
    ptx_ctx* ctx = argc > 1 ? ctx_of(env, argv[0]) : nullptr;
    void* p = nullptr;
    if (ctx && L.handle && napi_get_value_external(env, argv[1], &p) == napi_ok && p) L.comm_destroy(ctx, (ptx_comm*)p);
    return nullptr;
}

/* mergeAndGather(ctx, comm, batch, counts: Uint32Array(nRanks), replicas): this rank's replica logs are merged, the per-replica
 * digests of ALL ranks gathered on the device (ptx_allgather_digests) and the converged documents of the whole job counted there.
 * Returns {logs: Uint32Array (this rank's ptx_log_result rows), gathered: BigUint64Array (2 per replica log, rank-major), converged}. */
napi_value MergeAndGather(napi_env env, napi_callback_info info) {
    if (!L.handle) return throw_msg(env, "call open(libPath) first");
    size_t argc = 5;
    napi_value argv[5];
    NAPI_OK(napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
    ptx_ctx* ctx = argc > 4 ? ctx_of(env, argv[0]) : nullptr;
    void* cp = nullptr;
    if (!ctx || napi_get_value_external(env, argv[1], &cp) != napi_ok || !cp) return throw_msg(env, "mergeAndGather(ctx, comm, batch, counts, replicas)");
    ptx_batch pb;
    if (!read_batch(env, argv[2], &pb)) return nullptr;
    napi_typedarray_type type;
    size_t n_ranks = 0, offset = 0;
    void* counts = nullptr;
    napi_value ab;
    if (napi_get_typedarray_info(env, argv[3], &type, &n_ranks, &counts, &ab, &offset) != napi_ok || type != napi_uint32_array || n_ranks == 0)
        return throw_msg(env, "mergeAndGather: counts must be a Uint32Array with one entry per rank");
    if (n_ranks != L.comm_n_ranks((const ptx_comm*)cp)) return throw_msg(env, "mergeAndGather: counts.length differs from the communicator's rank count");
    uint32_t replicas = 1;
    napi_get_value_uint32(env, argv[4], &replicas);
    uint64_t total = 0;
    for (size_t k = 0; k < n_ranks; ++k) total += ((const uint32_t*)counts)[k];
    ptx_dbatch* db = nullptr;
    ptx_dresult* dr = nullptr;
    void *d_gather = nullptr, *d_count = nullptr;
    std::vector<ptx_log_result> logs(pb.n_logs ? pb.n_logs : 1);
    std::vector<uint64_t> gathered(total * 2 + 1);
    uint64_t converged = 0;
    ptx_status st = L.batch_upload(ctx, &pb, &db);
    if (st == PTX_OK) st = L.result_alloc(ctx, db, &dr);
    if (st == PTX_OK) st = L.device_alloc(ctx, total * 16 + 16, &d_gather);
    if (st == PTX_OK) st = L.device_alloc(ctx, 8, &d_count);
    if (st == PTX_OK) st = L.merge(ctx, db, dr);
    if (st == PTX_OK) st = L.allgather_digests(ctx, (ptx_comm*)cp, dr, (const uint32_t*)counts, (uint64_t*)d_gather);
    if (st == PTX_OK) st = L.count_converged_digests(ctx, (const uint64_t*)d_gather, total, replicas, (uint64_t*)d_count);
    if (st == PTX_OK) st = L.result_download_logs(ctx, dr, logs.data(), pb.n_logs);
    if (st == PTX_OK) st = L.device_read(ctx, d_gather, gathered.data(), total * 16);
    if (st == PTX_OK) st = L.device_read(ctx, d_count, &converged, 8);
    std::string err = st != PTX_OK ? L.last_error(ctx) : "";
    if (d_gather) L.device_free(ctx, d_gather);
    if (d_count) L.device_free(ctx, d_count);
    if (dr) L.dresult_free(ctx, dr);
    if (db) L.batch_free(ctx, db);
    if (st != PTX_OK) return throw_msg(env, ("mergeAndGather failed: " + err).c_str());
    napi_value out, v;
    NAPI_OK(napi_create_object(env, &out));
    v = make_u32(env, logs.data(), (size_t)pb.n_logs * 12);
    if (v) napi_set_named_property(env, out, "logs", v);
    v = make_typed(env, napi_biguint64_array, 8, gathered.data(), (size_t)total * 2);
    if (v) napi_set_named_property(env, out, "gathered", v);
    napi_create_double(env, (double)converged, &v);
    napi_set_named_property(env, out, "converged", v);
    return out;
}

napi_value MaxOpsPerLog(napi_env env, napi_callback_info info) {
    size_t argc = 1;
    napi_value argv[1];
    NAPI_OK(napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr));
    ptx_ctx* ctx = argc ? ctx_of(env, argv[0]) : nullptr;
    napi_value v;
    NAPI_OK(napi_create_uint32(env, L.handle ? L.max_ops_per_log(ctx) : 0, &v));
    return v;
}

napi_value KernelName(napi_env env, napi_callback_info) {
    napi_value v;
    NAPI_OK(napi_create_string_utf8(env, L.handle ? L.kernel_name() : "", NAPI_AUTO_LENGTH, &v));
    return v;
}

napi_value Init(napi_env env, napi_value exports) {
    struct { const char* name; napi_callback fn; } fns[] = {
        {"open", Open}, {"create", Create}, {"destroy", Destroy}, {"applyMaterialize", ApplyMaterialize}, {"residentUpload", ResidentUpload}, {"residentAppend", ResidentAppend}, {"residentApply", ResidentApply}, {"residentFree", ResidentFree}, {"generate", Generate}, {"change", Change}, {"cursors", Cursors}, {"rootMap", RootMap}, {"commUniqueId", CommUniqueId}, {"commInit", CommInit}, {"commDestroy", CommDestroy}, {"mergeAndGather", MergeAndGather},
        {"maxOpsPerLog", MaxOpsPerLog}, {"kernelName", KernelName},
    };
    for (auto& f : fns) {
        napi_value fn;
        if (napi_create_function(env, f.name, NAPI_AUTO_LENGTH, f.fn, nullptr, &fn) != napi_ok) return nullptr;
        if (napi_set_named_property(env, exports, f.name, fn) != napi_ok) return nullptr;
    }
    return exports;
}

}  // namespace

NAPI_MODULE(NODE_GYP_MODULE_NAME, Init)
