"""Patch[] streams: the replay logic of peritext_amd/csrc/replay_core.h compiled with -DPTX_EMU (tests/emu,
test tooling only) against what the reference's applyChange returns (micromerge.ts:499): golden fixtures made with the
type-erased reference itself (oracle/gen_patch_golden.js --impl ref), the reference's 46 test cases and traces through
the oracle, the A.6 quirks, and the accumulatePatches property of reference/test/accumulatePatches.ts.  The GPU tests
(test_gpu_parity.py) repeat the fixture comparisons through ptx_replay_patches on a real MI355X."""
import json
import os

import numpy as np
import pytest

import helpers as H
from peritext_amd import abi, wire

pytestmark = pytest.mark.skipif(not os.path.exists(H.EMU_LIB), reason="tests/emu/libperitext_emu.so not built (run __graft_entry__.build())")

PATCH_GOLDEN = ["patches_mini.json", "patches_rich_300.json"]


def _load(name):
    with open(os.path.join(H.GOLDEN, name)) as f:
        return json.load(f)


_check_streams = H.check_patch_streams
accumulate = H.accumulate_patches


@pytest.mark.parametrize("name", PATCH_GOLDEN)
@pytest.mark.parametrize("reverse", [0, 1, 2])
@pytest.mark.parametrize("gwin", [False, True])
def test_golden_patch_streams(name, reverse, gwin):
    """Fixtures produced by the reference itself: every patch of every replica log, in order, deep-equal."""
    g = _load(name)
    assert g["impl"] == "ref"
    batch = wire.encode_docs([d["logs"] for d in g["docs"]])
    res = H.emu_merge(batch, lds_bytes=160 * 1024, reverse=reverse)
    pat = H.emu_replay(batch, res, reverse=reverse, gwin=gwin)
    n = _check_streams(batch, pat, [d["expected"] for d in g["docs"]])
    assert n == batch.n_logs


def test_kat_and_trace_patch_streams():
    """The reference's 46 test cases and 9 traces: stream of every replica log == the oracle's (which the same 46
    cases and the differential fuzz against oracle/_ref pin, patches included)."""
    if not H.have_node():
        pytest.skip("node not installed")
    cases = H.load_kat()
    docs = [[r["log"] for r in c["replicas"]] for c in cases] + [t["logs"] for t in _load("reference_traces.json")]
    batch = wire.encode_docs(docs)
    expected = H.oracle_apply(docs, patches=True)
    for reverse in (0, 1):
        res = H.emu_merge(batch, reverse=reverse)
        ok = [[e for e in exp if "error" not in e] for exp in expected]
        assert all(len(a) == len(b) for a, b in zip(ok, expected)), "no KAT / trace log fails"
        for gwin in (False, True):
            _check_streams(batch, H.emu_replay(batch, res, reverse=reverse, gwin=gwin), expected)


def _mini_doc(ops, first_text="ABCDE"):
    from test_emu_parity import _mini_doc as m
    return m(ops, first_text=first_text)


def test_patch_quirks_against_oracle():
    """SURVEY A.6 corners in the patch stream: remove of an absent comment (undefined -> [] IS a change), re-adding an
    active mark (no patch), an end slot met before the start (defines a slot, later patches break there), same-slot
    marks (run to the end of the text), endOfText + later insert inheriting marks, comment add/remove/re-add chains."""
    if not H.have_node():
        pytest.skip("node not installed")
    el = lambda i: "%d@a" % (i + 2)  # noqa: E731
    bf = lambda i: {"type": "before", "elemId": el(i)}  # noqa: E731
    af = lambda i: {"type": "after", "elemId": el(i)}  # noqa: E731
    docs = [
        [_mini_doc([{"action": "removeMark", "markType": "comment", "attrs": {"id": "c1"}, "start": bf(1), "end": af(3)},
                    {"action": "removeMark", "markType": "comment", "attrs": {"id": "c1"}, "start": bf(0), "end": af(4)}])],
        [_mini_doc([{"action": "addMark", "markType": "strong", "start": bf(1), "end": bf(3)},
                    {"action": "addMark", "markType": "strong", "start": bf(0), "end": bf(4)},
                    {"action": "removeMark", "markType": "strong", "start": bf(2), "end": {"type": "endOfText"}}])],
        [_mini_doc([{"action": "addMark", "markType": "link", "attrs": {"url": "u"}, "start": bf(3), "end": af(1)},
                    {"action": "addMark", "markType": "link", "attrs": {"url": "v"}, "start": bf(0), "end": af(4)},
                    {"action": "addMark", "markType": "link", "attrs": {"url": "v"}, "start": bf(1), "end": af(2)}])],
        [_mini_doc([{"action": "addMark", "markType": "em", "start": bf(2), "end": bf(2)},
                    {"action": "set", "insert": True, "elemId": el(4), "value": "!"},
                    {"action": "del", "elemId": el(3)},
                    {"action": "removeMark", "markType": "em", "start": bf(3), "end": bf(4)}])],
        [_mini_doc([{"action": "addMark", "markType": "em", "start": {"type": "before", "elemId": "99@zz"}, "end": af(2)},
                    {"action": "addMark", "markType": "em", "start": bf(0), "end": af(4)}])],
        [_mini_doc([{"action": "addMark", "markType": "comment", "attrs": {"id": "c2"}, "start": bf(0), "end": af(2)},
                    {"action": "addMark", "markType": "comment", "attrs": {"id": "c1"}, "start": bf(1), "end": af(4)},
                    {"action": "removeMark", "markType": "comment", "attrs": {"id": "c2"}, "start": bf(1), "end": af(1)},
                    {"action": "addMark", "markType": "comment", "attrs": {"id": "c2"}, "start": bf(0), "end": af(4)},
                    {"action": "set", "insert": True, "elemId": el(1), "value": "x"},
                    {"action": "set", "insert": True, "elemId": el(0), "value": "y"},
                    {"action": "addMark", "markType": "comment", "attrs": {"id": "c1"}, "start": bf(0), "end": af(4)}])],
        [_mini_doc([{"action": "del", "elemId": el(i)} for i in range(5)] + [{"action": "del", "elemId": el(0)},
                    {"action": "addMark", "markType": "strong", "start": bf(0), "end": bf(4)},
                    {"action": "set", "insert": True, "elemId": el(2), "value": "z"}])],
        [_mini_doc([], first_text="")],
    ]
    expected = H.oracle_apply(docs, patches=True)
    batch = wire.encode_docs(docs)
    for reverse in (0, 1):
        res = H.emu_merge(batch, reverse=reverse)
        pat = H.emu_replay(batch, res, reverse=reverse)
        _check_streams(batch, pat, expected)
    # the quirks really are in the expected streams
    assert [p["action"] for p in expected[0][0]["patches"]].count("removeMark") >= 2
    assert sum(1 for p in expected[1][0]["patches"] if p["action"] == "addMark") == 3  # the second add only patches the two new ends


@pytest.mark.parametrize("config,docs,ops", [("mini", 6, None), ("rich", 1, 500), ("config4", 1, 700), ("config5", 1, 900)])
def test_live_oracle_patch_streams_and_accumulate(config, docs, ops):
    """Fresh PTXGEN documents: streams equal the oracle's, and replaying a stream per character
    (test/accumulatePatches.ts) reproduces the batch result getTextWithFormatting gives for the same log."""
    if not H.have_node():
        pytest.skip("node not installed")
    g = H.oracle_gen(config, seed=23, docs=docs, ops=ops)
    dl = [d["logs"] for d in g["docs"]]
    expected = H.oracle_apply(dl, patches=True)
    batch = wire.encode_docs(dl)
    res = H.emu_merge(batch, lds_bytes=160 * 1024)
    pat = H.emu_replay(batch, res)
    _check_streams(batch, pat, expected)
    pat_g = H.emu_replay(batch, res, gwin=True)  # per-slot urls, op tables and comment id tables in global memory: the same records
    assert np.array_equal(pat_g.patch_off, pat.patch_off) and np.array_equal(pat_g.patches, pat.patches) and np.array_equal(pat_g.logs, pat.logs)
    for log in range(batch.n_logs):
        got = accumulate(wire.decode_patches(batch, pat, log, with_rows=True))
        want = wire.decode_spans(batch, res, log)
        assert H.norm_spans(got) == H.norm_spans(want), "log %d: accumulated patches != batch spans" % log


def test_failed_logs_have_no_stream_and_capacity_is_reported():
    docs = [
        [_mini_doc([{"action": "del", "elemId": "77@zz"}])],
        [_mini_doc([{"action": "set", "insert": True, "elemId": "6@a", "value": "ok"}])],
    ]
    batch = wire.encode_docs(docs)
    res = H.emu_merge(batch)
    pat = H.emu_replay(batch, res)
    assert int(pat.logs[0]["status"]) == abi.ERR_ELEM_NOT_FOUND and int(pat.logs[0]["n_patches"]) == 0
    assert int(pat.logs[1]["status"]) == 0 and int(pat.logs[1]["n_patches"]) == 7  # makeList + 5 chars + 1
    with pytest.raises(ValueError, match="List element not found"):
        wire.decode_patches(batch, pat, 0)
    # a too-small record capacity: the count is still exact, the status says the rows are truncated
    small = H.emu_replay(batch, res, cap=3)
    assert int(small.logs[1]["status"]) == abi.ERR_CAPACITY and int(small.logs[1]["n_patches"]) == 7
    # a too-small on-chip working set
    tiny = H.emu_replay(batch, res, lds_bytes=256)
    assert int(tiny.logs[1]["status"]) == abi.ERR_CAPACITY and int(tiny.logs[1]["n_patches"]) == 0


def test_replay_lds_bound():
    import ctypes as C
    lib = C.CDLL(H.EMU_LIB)
    lib.ptx_emu_replay_lds_need.restype = C.c_uint64
    lib.ptx_emu_replay_lds_need.argtypes = [C.c_uint64] * 5
    # a config-4 log (1250 inserts, 1600 marks, 400 comment ops, 8.7 k ids) replays within a quarter of the LDS of a CU
    assert lib.ptx_emu_replay_lds_need(1250, 1600, 400, 8736, 400) < 40 * 1024
    assert lib.ptx_emu_replay_lds_need(0, 0, 0, 0, 0) < 2048


def test_patch_streams_random_workloads():
    """Seeded random workload definitions (mix, mark types, replicas, log length): patch streams equal the oracle's, in both
    iteration orders of the emulated parallel loops."""
    if not H.have_node():
        pytest.skip("node not installed")
    rng = np.random.default_rng(7_2024)
    all_marks = ["strong", "em", "comment", "link"]
    for trial in range(8):
        cuts = np.sort(rng.integers(0, 101, size=3))
        mix = (int(cuts[0]), int(cuts[1] - cuts[0]), int(cuts[2] - cuts[1]), int(100 - cuts[2]))
        marks = tuple(all_marks[i] for i in rng.permutation(4)[: int(rng.integers(1, 5))])
        replicas = int(rng.integers(1, 5))
        ops = int(rng.integers(30, 400))
        g = H.oracle_gen("mini", seed=int(rng.integers(1, 1 << 30)), docs=2, ops=ops, replicas=replicas, mix=mix, marks=marks)
        dl = [d["logs"] for d in g["docs"]]
        expected = H.oracle_apply(dl, patches=True)
        batch = wire.encode_docs(dl)
        for reverse in (0, 1):
            res = H.emu_merge(batch, lds_bytes=160 * 1024, reverse=reverse)
            pat = H.emu_replay(batch, res, reverse=reverse)
            _check_streams(batch, pat, expected)


def test_patch_streams_with_op_counters_beyond_the_dense_key_range():
    """The replay keeps the LWW winners of strong / em per slot as dense op-id keys where the log's id space fits 16 bits, and as rows (op ids read
    back from the columns) where it does not: the same documents with every counter moved up by 70 000 take the second path and must give the very
    same streams (a patch carries no op id)."""
    g = _load("patches_rich_300.json")
    docs = [d["logs"] for d in g["docs"]]
    wide = wire.encode_docs(H.shift_counters(docs, 70000))
    assert (int(wide.log_hdr["max_counter"].max()) + 1) * (int(wide.log_hdr["max_actor"].max()) + 1) > 65535
    res = H.emu_merge(wide, lds_bytes=160 * 1024)
    assert _check_streams(wide, H.emu_replay(wide, res), [d["expected"] for d in g["docs"]]) == wide.n_logs


@pytest.mark.parametrize("gwin", [False, True])
def test_tail_streams_are_the_suffix_of_the_whole_stream(gwin):
    """ptx_replay_patches_from: the records of the rows from first_row[l] on are exactly the tail of the log's whole stream (same records, same order),
    whatever the cut — row 0, inside a change, the last row, past the end."""
    g = _load("patches_rich_300.json")
    batch = wire.encode_docs([d["logs"] for d in g["docs"]])
    res = H.emu_merge(batch, lds_bytes=160 * 1024)
    whole = H.emu_replay(batch, res, gwin=gwin)
    sizes = np.diff(batch.log_off.astype(np.int64))
    rng = np.random.default_rng(5)
    for cut in ("zero", "random", "last", "past"):
        first = {"zero": np.zeros_like(sizes), "random": rng.integers(0, np.maximum(sizes, 1)), "last": np.maximum(sizes - 1, 0), "past": sizes + 3}[cut]
        tail = H.emu_replay(batch, res, gwin=gwin, first_row=first)
        for log in range(batch.n_logs):
            a = whole.patches[int(whole.patch_off[log]):int(whole.patch_off[log]) + int(whole.logs["n_patches"][log])]
            b = tail.patches[int(tail.patch_off[log]):int(tail.patch_off[log]) + int(tail.logs["n_patches"][log])]
            want = a[a["row"] >= first[log]]
            assert tail.logs["status"][log] == whole.logs["status"][log]
            assert np.array_equal(b, want), (cut, log, int(first[log]))


@pytest.mark.parametrize("gwin", [False, True])
def test_overflow_extents_give_the_same_streams(gwin):
    """ptx_replay_patches launches the replay ONCE: a log that outgrows the capacity guessed for it takes one overflow extent from an arena (an atomic bump)
    and goes on writing there; the two parts are packed afterwards.  With 40 records of capacity every log of the fixture overflows: the packed streams
    equal the ones replayed into ample room; with an arena too small for all of them, the logs that got no extent report PTX_ERR_CAPACITY with exact counts."""
    g = _load("patches_rich_300.json")
    batch = wire.encode_docs([d["logs"] for d in g["docs"]])
    res = H.emu_merge(batch)
    ref = H.emu_replay(batch, res, gwin=gwin)
    pat, ext = H.emu_replay_with_arena(batch, res, cap=40, arena=200000, gwin=gwin)
    assert np.all(ext[:batch.n_logs, 0] != np.uint64(0xFFFFFFFFFFFFFFFF)) and np.all(pat.logs["status"] == 0)
    assert np.array_equal(pat.logs["n_patches"], ref.logs["n_patches"])
    for log in range(batch.n_logs):
        assert np.array_equal(pat.patches[int(pat.patch_off[log]):int(pat.patch_off[log + 1])], ref.patches[int(ref.patch_off[log]):int(ref.patch_off[log]) + int(ref.logs[log]["n_patches"])])
    assert H.check_patch_streams(batch, pat, [d["expected"] for d in g["docs"]]) == batch.n_logs
    starved, ext2 = H.emu_replay_with_arena(batch, res, cap=40, arena=9000, gwin=gwin)
    got = ext2[:batch.n_logs, 0] != np.uint64(0xFFFFFFFFFFFFFFFF)
    assert got.any() and not got.all()
    assert np.array_equal(starved.logs["n_patches"][~got], ref.logs["n_patches"][~got]) and np.all(starved.logs["status"][~got] == abi.ERR_CAPACITY)
    assert np.all(starved.logs["status"][got] == 0)
    # a log that starts quietly (one record per row) and then produces many per row outgrows the first extent (sized from the rate so far) and takes a second
    late = wire.encode_docs([[H.synthetic_marks_log(400, 300, 5, max_span=400)]])
    lres = H.emu_merge(late, lds_bytes=160 * 1024)
    lref = H.emu_replay(late, lres, cap=200000, gwin=gwin)
    lpat, lext = H.emu_replay_with_arena(late, lres, cap=40, arena=400000, gwin=gwin)
    assert int(lext[0, 1]) != 0xFFFFFFFFFFFFFFFF and int(lpat.logs[0]["status"]) == 0, (lext, lref.logs)
    n = int(lref.logs[0]["n_patches"])
    assert int(lpat.logs[0]["n_patches"]) == n and n > 40 + int(lext[0, 2])
    assert np.array_equal(lpat.patches[:n], lref.patches[:n])


@pytest.mark.parametrize("gwin", [False, True])
def test_marks_that_arrive_after_larger_op_ids(gwin):
    """The replay decides compareOpIds per slot without a per-slot winner: an op loses where an earlier-APPLIED op of its type with a larger opId covers.  In the
    fuzzer's logs few ops meet a larger id; here three actors mark the same text concurrently and a replica applies them in descending id order, so that almost
    every op has many larger ones applied before it (and every link / comment state is met): streams against the oracle, op for op."""
    if not H.have_node():
        pytest.skip("node not installed")
    docs = H.concurrent_marks_docs()
    expected = H.oracle_apply(docs, patches=True)
    if os.path.isdir(os.path.join(os.path.dirname(H.GOLDEN), "..", "oracle", "_ref")):  # the type-erased reference itself says the same (build container)
        ref = H.oracle_apply(docs, patches=True, impl="ref")
        assert [[H.norm_patches(r["patches"]) for r in d] for d in ref] == [[H.norm_patches(r["patches"]) for r in d] for d in expected]
    batch = wire.encode_docs(docs)
    for reverse in (0, 1, 2):
        res = H.emu_merge(batch, reverse=reverse, lds_bytes=160 * 1024)
        assert np.all(res.logs["status"] == 0)
        pat = H.emu_replay(batch, res, reverse=reverse, gwin=gwin)
        _check_streams(batch, pat, expected)
