"""ctypes mirror of include/peritext_hip.h (the C ABI of libperitext_hip.so).

The structures here must stay byte-identical to the header; tests/test_abi.py checks sizes and that
every symbol the header declares is exported.  There is no CPU fallback: `load_library()` raises if
the shared object is missing, and `ptx_create` fails when no gfx950 device is visible.
"""
import ctypes as C
import os

PTX_ABI_VERSION = 7

# Operation.action (reference/src/micromerge.ts:150-212, src/peritext.ts:25-65)
ACT_MAKELIST, ACT_INSERT, ACT_DELETE, ACT_ADDMARK, ACT_REMOVEMARK, ACT_NOP, ACT_MAPSET, ACT_MAPDEL = range(8)
MAPV_SCALAR, MAPV_MAP, MAPV_LIST, MAPV_DELETED = range(4)  # what a map row writes / ptx_root_entry.kind
# markType in ALL_MARKS order (reference/src/schema.ts:125)
MARK_STRONG, MARK_EM, MARK_COMMENT, MARK_LINK = range(4)
MARK_NAMES = ["strong", "em", "comment", "link"]
# BoundaryPosition.type (reference/src/peritext.ts:17-21)
SIDE_BEFORE, SIDE_AFTER, SIDE_START_OF_TEXT, SIDE_END_OF_TEXT = range(4)
SIDE_NAMES = ["before", "after", "startOfText", "endOfText"]

ATTR_STRONG = 0x10000000
ATTR_EM = 0x20000000
ATTR_LINK = 0x40000000
ATTR_COMMENT = 0x80000000
ATTR_ID_MASK = 0x0FFFFFFF
RANK_TOMBSTONE = 0x80000000
RANK_MASK = 0x7FFFFFFF

FLAG_NO_ELEM_RANK = 1
FLAG_NO_ADMISSION = 2
FLAG_PAD_GATHER = 4
FLAG_REPLAY_LDS_ONLY = 8
COMM_ID_BYTES = 128

PTX_OK = 0
ERR_ELEM_NOT_FOUND = 1
ERR_SEQ_GAP = 2
ERR_MISSING_DEP = 3
ERR_DUPLICATE_OP = 4
ERR_CAPACITY = 5
ERR_BAD_OP = 6
ERR_INDEX_OOB = 7
ERR_INVALID_ARG = 100
ERR_HIP = 101
ERR_NO_DEVICE = 102
ERR_OOM = 103

STATUS_NAMES = {
    0: "ok",
    1: "RangeError: List element not found",
    2: "RangeError: Expected sequence number",
    3: "RangeError: Missing dependency",
    4: "duplicate opId",
    5: "log exceeds on-chip capacity",
    6: "malformed op row",
    7: "RangeError: List index out of bounds",
}



CHG_ACTOR_SHIFT = 20
CHG_NOPS = 0x000FFFFF
ENV_SATURATED = 65535


def env_stride(max_actors):
    """u16 entries of one chg_env row (PTX_ENV_STRIDE): seq + deps[max_actors], padded to a multiple of 4."""
    return (1 + int(max_actors) + 3) & ~3


def envelope_bytes(n_changes, max_actors):
    """Bytes of the Change envelope the admission phase reads: chg_hdr (u32) + one chg_env row (u16 x env_stride) per change."""
    return n_changes * (4 + 2 * env_stride(max_actors))


u8p = C.POINTER(C.c_uint8)
u16p = C.POINTER(C.c_uint16)
u32p = C.POINTER(C.c_uint32)
u64p = C.POINTER(C.c_uint64)


class ptx_log_hdr(C.Structure):
    _fields_ = [
        ("n_ins", C.c_uint32),
        ("n_del", C.c_uint32),
        ("n_mark", C.c_uint32 * 4),
        ("max_counter", C.c_uint32),
        ("max_actor", C.c_uint32),
        ("n_comment_ids", C.c_uint32),
        ("reserved", C.c_uint32),
    ]


class ptx_batch(C.Structure):
    _fields_ = [
        ("n_logs", C.c_uint32),
        ("reserved", C.c_uint32),
        ("n_ops", C.c_uint64),
        ("log_off", u64p),
        ("op_id", u64p),
        ("ref_a", u64p),
        ("ref_b", u64p),
        ("payload", u32p),
        ("action", u8p),
        ("mark_type", u8p),
        ("side_a", u8p),
        ("side_b", u8p),
        ("chg_off", u64p),
        ("chg_hdr", u32p),
        ("chg_env", u16p),
        ("max_actors", C.c_uint32),
        ("reserved2", C.c_uint32),
        ("log_hdr", C.POINTER(ptx_log_hdr)),
        ("chg_env_hi", u16p),
    ]


class ptx_span(C.Structure):
    _fields_ = [("start", C.c_uint32), ("attr", C.c_uint32)]


class ptx_cinterval(C.Structure):
    _fields_ = [("id", C.c_uint32), ("start", C.c_uint32), ("end", C.c_uint32)]


class ptx_log_result(C.Structure):
    _fields_ = [
        ("status", C.c_uint32),
        ("n_ops", C.c_uint32),
        ("n_elems", C.c_uint32),
        ("n_visible", C.c_uint32),
        ("n_spans", C.c_uint32),
        ("n_cintervals", C.c_uint32),
        ("reserved", C.c_uint32 * 2),
        ("digest", C.c_uint64 * 2),
    ]


class ptx_result(C.Structure):
    _fields_ = [
        ("n_logs", C.c_uint32),
        ("reserved", C.c_uint32),
        ("n_rows", C.c_uint64),
        ("logs", C.POINTER(ptx_log_result)),
        ("values", u32p),
        ("spans", C.POINTER(ptx_span)),
        ("cintervals", C.POINTER(ptx_cinterval)),
        ("elem_rank", u32p),
        ("value_off", C.POINTER(C.c_uint64)),  # ABI 7: the rows are compact, log l's at [off[l], off[l + 1])
        ("span_off", C.POINTER(C.c_uint64)),
        ("cint_off", C.POINTER(C.c_uint64)),
        ("owner", C.c_void_p),
    ]


# Patch[] streams (what applyChange returns, reference/src/micromerge.ts:499)
PATCH_MAKELIST, PATCH_INSERT, PATCH_DELETE, PATCH_ADDMARK, PATCH_REMOVEMARK, PATCH_INSERT_COMMENT = range(6)


class ptx_patch(C.Structure):
    _fields_ = [("row", C.c_uint32), ("kind", C.c_uint32), ("a", C.c_uint32), ("b", C.c_uint32)]


class ptx_patch_log(C.Structure):
    _fields_ = [("status", C.c_uint32), ("n_patches", C.c_uint32)]


class ptx_root_entry(C.Structure):
    _fields_ = [("obj", C.c_uint64), ("key", C.c_uint32), ("row", C.c_uint32), ("kind", C.c_uint32), ("value", C.c_uint32)]


class ptx_root_log(C.Structure):
    _fields_ = [("status", C.c_uint32), ("n_entries", C.c_uint32), ("first_bad_row", C.c_uint32), ("reserved", C.c_uint32)]


class ptx_root_maps(C.Structure):
    _fields_ = [
        ("n_logs", C.c_uint32),
        ("reserved", C.c_uint32),
        ("entry_off", C.POINTER(C.c_uint64)),
        ("logs", C.POINTER(ptx_root_log)),
        ("entries", C.POINTER(ptx_root_entry)),
        ("owner", C.c_void_p),
    ]


class ptx_patches(C.Structure):
    _fields_ = [
        ("n_logs", C.c_uint32),
        ("launches", C.c_uint32),
        ("kernel_ms", C.c_float),
        ("reserved", C.c_uint32),
        ("patch_off", u64p),
        ("logs", C.POINTER(ptx_patch_log)),
        ("patches", C.POINTER(ptx_patch)),
        ("owner", C.c_void_p),
    ]


class ptx_gen_config(C.Structure):
    _fields_ = [
        ("replicas", C.c_uint32),
        ("ops_per_log", C.c_uint32),
        ("mix", C.c_uint32 * 4),
        ("n_mark_types", C.c_uint32),
        ("mark_types", C.c_uint8 * 4),
        ("seed", C.c_uint32),
        ("first_doc", C.c_uint32),
        ("n_docs", C.c_uint32),
        ("list_cap", C.c_uint32),
        ("initial_text", C.c_char * 16),
    ]


class ptx_gen_info(C.Structure):
    _fields_ = [("n_docs", C.c_uint32), ("kernel_ms", C.c_float), ("n_comments", u32p), ("owner", C.c_void_p)]


# InputOperation.action of ptx_change (reference/src/micromerge.ts:133-148)
IN_INSERT, IN_DELETE, IN_ADDMARK, IN_REMOVEMARK, IN_MAKELIST, IN_MAPSET, IN_MAPDEL = range(7)
IN_OBJ_NEW = 0x80000000  # ptx_input_ops.index of a map op: the object made by row k of this log's output
CURSOR_RESOLVE, CURSOR_GET = 0, 1


class ptx_input_ops(C.Structure):
    _fields_ = [
        ("n_logs", C.c_uint32),
        ("max_actors", C.c_uint32),
        ("chg_off", u64p),
        ("op_off", u64p),
        ("action", u8p),
        ("mark_type", u8p),
        ("index", u32p),
        ("count", u32p),
        ("payload", u32p),
        ("values", u32p),
        ("n_values", C.c_uint64),
        ("actor", u32p),
    ]


class ptx_host_batch(C.Structure):
    _fields_ = [("b", ptx_batch), ("owner", C.c_void_p)]


# numpy dtypes with the same layout
import numpy as np  # noqa: E402

LOG_RESULT_DTYPE = np.dtype(
    [
        ("status", "<u4"),
        ("n_ops", "<u4"),
        ("n_elems", "<u4"),
        ("n_visible", "<u4"),
        ("n_spans", "<u4"),
        ("n_cintervals", "<u4"),
        ("reserved", "<u4", (2,)),
        ("digest", "<u8", (2,)),
    ]
)
LOG_HDR_DTYPE = np.dtype([("n_ins", "<u4"), ("n_del", "<u4"), ("n_mark", "<u4", (4,)), ("max_counter", "<u4"), ("max_actor", "<u4"),
                          ("n_comment_ids", "<u4"), ("reserved", "<u4")])
SPAN_DTYPE = np.dtype([("start", "<u4"), ("attr", "<u4")])
CINTERVAL_DTYPE = np.dtype([("id", "<u4"), ("start", "<u4"), ("end", "<u4")])
PATCH_DTYPE = np.dtype([("row", "<u4"), ("kind", "<u4"), ("a", "<u4"), ("b", "<u4")])
PATCH_LOG_DTYPE = np.dtype([("status", "<u4"), ("n_patches", "<u4")])
ROOT_ENTRY_DTYPE = np.dtype([("obj", "<u8"), ("key", "<u4"), ("row", "<u4"), ("kind", "<u4"), ("value", "<u4")])
ROOT_LOG_DTYPE = np.dtype([("status", "<u4"), ("n_entries", "<u4"), ("first_bad_row", "<u4"), ("reserved", "<u4")])

# every function include/peritext_hip.h declares: name -> (restype, argtypes)
vp = C.c_void_p
FUNCTIONS = {
    "ptx_abi_version": (C.c_uint32, []),
    "ptx_create": (C.c_int32, [C.c_int, C.c_uint32, C.POINTER(vp)]),
    "ptx_destroy": (None, [vp]),
    "ptx_last_error": (C.c_char_p, [vp]),
    "ptx_apply_materialize": (C.c_int32, [vp, C.POINTER(ptx_batch), C.POINTER(ptx_result)]),
    "ptx_result_free": (None, [C.POINTER(ptx_result)]),
    "ptx_batch_upload": (C.c_int32, [vp, C.POINTER(ptx_batch), C.POINTER(vp)]),
    "ptx_batch_upload_tiled": (C.c_int32, [vp, C.POINTER(ptx_batch), C.c_uint32, C.POINTER(vp)]),
    "ptx_batch_wrap_device": (C.c_int32, [vp, C.POINTER(ptx_batch), C.POINTER(vp)]),
    "ptx_batch_append": (C.c_int32, [vp, vp, C.POINTER(ptx_batch), C.POINTER(vp)]),
    "ptx_batch_free": (None, [vp, vp]),
    "ptx_batch_n_logs": (C.c_uint32, [vp]),
    "ptx_batch_n_ops": (C.c_uint64, [vp]),
    "ptx_batch_n_changes": (C.c_uint64, [vp]),
    "ptx_batch_launch_shape": (None, [vp, u32p, u32p]),
    "ptx_result_alloc": (C.c_int32, [vp, vp, C.POINTER(vp)]),
    "ptx_dresult_free": (None, [vp, vp]),
    "ptx_set_launch_shape": (C.c_int32, [vp, C.c_uint32, C.c_uint32]),
    "ptx_merge": (C.c_int32, [vp, vp, vp]),
    "ptx_merge_timed": (C.c_int32, [vp, vp, vp, C.c_uint32, C.POINTER(C.c_float)]),
    "ptx_merge_phase_cycles": (C.c_int32, [vp, vp, vp, u64p, C.c_uint32]),
    "ptx_calib_stream": (C.c_int32, [vp, vp, u64p]),
    "ptx_sync": (C.c_int32, [vp]),
    "ptx_set_stream": (C.c_int32, [vp, vp]),
    "ptx_count_converged": (C.c_int32, [vp, vp, C.c_uint32, vp]),
    "ptx_result_download": (C.c_int32, [vp, vp, vp, C.POINTER(ptx_result)]),
    "ptx_result_download_range": (C.c_int32, [vp, vp, vp, C.c_uint32, C.c_uint32, C.POINTER(ptx_result)]),
    "ptx_result_download_logs": (C.c_int32, [vp, vp, C.POINTER(ptx_log_result), C.c_uint32]),
    "ptx_dresult_logs_device": (vp, [vp]),
    "ptx_pack_digests": (C.c_int32, [vp, vp, C.c_uint32, C.c_uint32, vp]),
    "ptx_comm_use_library": (C.c_int32, [vp, C.c_char_p]),
    "ptx_comm_unique_id": (C.c_int32, [vp, u8p]),
    "ptx_comm_init": (C.c_int32, [vp, u8p, C.c_uint32, C.c_uint32, C.POINTER(vp)]),
    "ptx_comm_destroy": (None, [vp, vp]),
    "ptx_comm_n_ranks": (C.c_uint32, [vp]),
    "ptx_comm_rank": (C.c_uint32, [vp]),
    "ptx_allgather_digests": (C.c_int32, [vp, vp, vp, u32p, vp]),
    "ptx_count_converged_digests": (C.c_int32, [vp, vp, C.c_uint64, C.c_uint32, vp]),
    "ptx_device_alloc": (C.c_int32, [vp, C.c_uint64, C.POINTER(vp)]),
    "ptx_device_free": (None, [vp, vp]),
    "ptx_device_read": (C.c_int32, [vp, vp, vp, C.c_uint64]),
    "ptx_replay_patches": (C.c_int32, [vp, vp, vp, C.POINTER(ptx_patches)]),
    "ptx_replay_patches_from": (C.c_int32, [vp, vp, vp, vp, C.POINTER(ptx_patches)]),
    "ptx_patches_free": (None, [C.POINTER(ptx_patches)]),
    "ptx_root_map": (C.c_int32, [vp, vp, C.POINTER(ptx_root_maps)]),
    "ptx_root_maps_free": (None, [C.POINTER(ptx_root_maps)]),
    "ptx_generate": (C.c_int32, [vp, C.POINTER(ptx_gen_config), C.POINTER(vp), C.POINTER(ptx_gen_info)]),
    "ptx_gen_info_free": (None, [C.POINTER(ptx_gen_info)]),
    "ptx_resolve_cursors": (C.c_int32, [vp, vp, vp, C.c_uint32, u32p, u8p, u64p, u64p, u32p]),
    "ptx_change": (C.c_int32, [vp, vp, vp, C.POINTER(ptx_input_ops), C.POINTER(vp), u32p]),
    "ptx_batch_append_device": (C.c_int32, [vp, vp, vp, C.POINTER(vp)]),
    "ptx_batch_download": (C.c_int32, [vp, vp, C.POINTER(ptx_host_batch)]),
    "ptx_host_batch_free": (None, [C.POINTER(ptx_host_batch)]),
    "ptx_max_ops_per_log": (C.c_uint32, [vp]),
    "ptx_kernel_name": (C.c_char_p, []),
    "ptx_batch_kernel_name": (C.c_char_p, [vp, vp]),
}

LIB_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib")
LIB_PATH = os.path.join(LIB_DIR, "libperitext_hip.so")
_lib = None


def load_library(path=None):
    """dlopen libperitext_hip.so and type every export.  Raises (never falls back) when it is missing."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise RuntimeError(
            "peritext_amd: %s is missing — build it with `python -c 'import __graft_entry__ as g; g.build()'`; "
            "there is no CPU fallback for the merge path" % p
        )
    lib = C.CDLL(p)
    for name, (res, args) in FUNCTIONS.items():
        fn = getattr(lib, name)  # AttributeError if the export is missing: loud by design
        fn.restype = res
        fn.argtypes = args
    if lib.ptx_abi_version() != PTX_ABI_VERSION:
        raise RuntimeError("peritext_amd: ABI version mismatch")
    if path is None:
        _lib = lib
    return lib
