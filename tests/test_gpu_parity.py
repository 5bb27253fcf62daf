"""Parity tests proper: the HIP path (libperitext_hip.so on a real MI355X, through the C ABI) against
the oracle — committed golden fixtures, live oracle runs on fresh seeds, the reference's 46 known-answer
cases and 9 traces, edge cases, and size-independent properties at BASELINE sizes."""
import json
import os

import numpy as np
import pytest

import helpers as H
from peritext_amd import abi, canon, wire

pytestmark = pytest.mark.gpu

GOLDEN_GEN = ["ptxgen_mini.json", "ptxgen_config2.json", "ptxgen_config3_512.json", "ptxgen_config4_600.json", "ptxgen_rich_700.json", "ptxgen_rich_2600.json",
              "ptxgen_config5_8192.json", "ptxgen_mini_10actors.json"]


@pytest.fixture(scope="module")
def eng():
    import torch

    assert torch.cuda.is_available(), "GPU tests need a real MI355X"
    from peritext_amd.engine import Engine

    e = Engine(0)  # raises if libperitext_hip.so is missing or no gfx950 is visible: no fallback
    yield e
    e.close()


def _load(name):
    with open(os.path.join(H.GOLDEN, name)) as f:
        return json.load(f)


def test_native_library_is_loaded(eng):
    assert os.path.samefile(eng.lib._name, abi.LIB_PATH)
    maps = open("/proc/self/maps").read()
    assert "libperitext_hip.so" in maps
    assert "libperitext_emu" not in maps  # the CPU emulation must never be in a GPU test process
    assert eng.kernel_name().startswith("ptx_merge_kernel")
    assert eng.max_ops_per_log() >= 4102


def test_kat_literals(eng):
    """Every replica log of the reference's 46 test cases -> the reference's expectedResult literal."""
    cases = H.load_kat()
    batch = wire.encode_docs([[r["log"] for r in c["replicas"]] for c in cases])
    res = eng.apply_materialize(batch)
    log = 0
    for c in cases:
        for r in c["replicas"]:
            want = c.get("expected", r["spans"])
            assert H.norm_spans(wire.decode_spans(batch, res, log)) == H.norm_spans(want), c["title"]
            log += 1
    assert log == 92


@pytest.mark.parametrize("name", GOLDEN_GEN)
def test_golden_ptxgen(eng, name):
    """Committed oracle output: decoded spans, raw canonical rows, digests — bit-exact."""
    H.check_generated(_load(name), eng.apply_materialize)


def test_reference_traces(eng):
    traces = _load("reference_traces.json")
    batch = wire.encode_docs([t["logs"] for t in traces])
    res = eng.apply_materialize(batch)
    log = 0
    for t in traces:
        for _ in t["logs"]:
            assert H.norm_spans(wire.decode_spans(batch, res, log)) == H.norm_spans(t["spans"]), t["name"]
            log += 1
    lm = [i for i, t in enumerate(traces) if t["name"] == "links-minimal.json"][0]
    assert traces[lm]["spans"][0]["text"] == "ABC9ee09150DE"


def test_staged_api_equals_one_shot_and_tiling(eng):
    gen = _load("ptxgen_mini.json")
    batch = wire.encode_docs([d["logs"] for d in gen["docs"]])
    one = eng.apply_materialize(batch)
    db = eng.upload(batch, copies=3)
    dr = eng.alloc_result(db)
    try:
        assert eng.n_logs(db) == 3 * batch.n_logs and eng.n_ops(db) == 3 * batch.n_ops
        eng.merge(db, dr)
        eng.merge(db, dr)  # idempotent: same buffers, same result
        got = eng.download(db, dr)
        logs_only = eng.download_logs(dr, 3 * batch.n_logs)
    finally:
        eng.free_result(dr)
        eng.free_batch(db)
    n = batch.n_ops
    for k in range(3):
        sl = slice(k * batch.n_logs, (k + 1) * batch.n_logs)
        assert (got.logs[sl] == one.logs).all() and (logs_only[sl] == one.logs).all()
        for log in range(batch.n_logs):
            a, b = wire.canonical_of_log(batch, one, log)[0], wire.canonical_of_log(batch, got, k * batch.n_logs + log)[0]  # (compact rows since ABI 7: by value_off)
            assert (a == b).all()
    tiled = batch.tile(3)
    for log in (0, batch.n_logs + 1, 3 * batch.n_logs - 1):
        assert wire.decode_spans(tiled, got, log) == wire.decode_spans(batch, one, log % batch.n_logs)


def test_compact_result_rows_of_large_and_small_ranges(eng):
    """ABI 7: ptx_result rows are compact (gathered on the device by the per-log offsets, copied into pinned memory).  A download of more than 65 536 op rows
    (offsets first, dense arrays of exactly the totals) and the downloads of single logs (one allocation sized by the rows, one wait) must return the same
    rows, and both the reference-made fixture's documents; the offsets are the exclusive prefix sums of the per-log counts."""
    gen = _load("ptxgen_rich_2600.json")
    batch = wire.encode_docs([d["logs"] for d in gen["docs"]])
    copies = 70000 // batch.n_ops + 1
    db = eng.upload(batch, copies=copies)
    dr = eng.alloc_result(db)
    try:
        assert eng.n_ops(db) > 65536
        eng.merge(db, dr)
        whole = eng.download(db, dr)
        n = eng.n_logs(db)
        for name, cnt in (("value_off", "n_visible"), ("span_off", "n_spans"), ("cint_off", "n_cintervals")):
            off = getattr(whole, name)
            assert int(off[0]) == 0 and (np.diff(off.astype(np.int64)) == whole.logs[cnt].astype(np.int64)).all(), name
        assert len(whole.values) == int(whole.value_off[n]) and len(whole.spans) == int(whole.span_off[n]) and len(whole.cintervals) == int(whole.cint_off[n])
        tiled = batch.tile(copies)
        exp = [e for d in gen["docs"] for e in d["expected"]]
        for log in (0, 1, batch.n_logs, n // 2, n - 1):
            one = eng.download_range(db, dr, log, 1)
            assert (one.logs[0] == whole.logs[log]).all()
            for a, b in zip(wire.canonical_of_log(tiled, whole, log), wire.canonical_of_log(tiled, one, 0)):
                assert a.tobytes() == b.tobytes()
            H.check_log(tiled, whole, log, exp[log % batch.n_logs])
        some = eng.download_range(db, dr, 3, 4)  # a range in the middle: offsets rebased to its first log
        for k in range(4):
            for a, b in zip(wire.canonical_of_log(tiled, whole, 3 + k), wire.canonical_of_log(tiled, some, k)):
                assert a.tobytes() == b.tobytes()
        assert int(eng.download_range(db, dr, 2, 0).value_off[0]) == 0  # an empty range
    finally:
        eng.free_result(dr)
        eng.free_batch(db)


def test_wrap_device_columns(eng):
    """ptx_batch_wrap_device: op columns that already live in HBM (here torch tensors) are merged in place — no
    copy, log headers from the device census — with the same result rows as the upload path."""
    import torch

    gen = _load("ptxgen_config3_512.json")
    batch = wire.encode_docs([d["logs"] for d in gen["docs"]])
    want = eng.apply_materialize(batch)
    cols = {}
    for name in ("log_off", "op_id", "ref_a", "ref_b", "payload", "action", "mark_type", "side_a", "side_b"):
        a = getattr(batch, name)
        as_signed = {np.dtype("uint64"): np.int64, np.dtype("uint32"): np.int32, np.dtype("uint8"): np.uint8}[a.dtype]
        cols[name] = torch.from_numpy(a.view(as_signed).copy()).cuda()
    torch.cuda.synchronize()
    db = eng.wrap_device(batch.n_logs, batch.n_ops, {k: v.data_ptr() for k, v in cols.items()})
    dr = eng.alloc_result(db)
    try:
        eng.merge(db, dr)
        got = eng.download(db, dr)
    finally:
        eng.free_result(dr)
        eng.free_batch(db)
    assert (got.logs["status"] == 0).all() and (got.logs["digest"] == want.logs["digest"]).all()
    for log in range(batch.n_logs):
        assert wire.decode_spans(batch, got, log) == wire.decode_spans(batch, want, log)


def test_elem_rank_is_document_position(eng):
    """elem_rank of an insert row == findListElement(...).index in the oracle's final metadata order:
    checked through the property that visible elements sorted by rank spell the document text."""
    gen = _load("ptxgen_rich_700.json")
    batch = wire.encode_docs([d["logs"] for d in gen["docs"]])
    res = eng.apply_materialize(batch)
    log = 0
    for d in gen["docs"]:
        for exp in d["expected"]:
            b0, b1 = int(batch.log_off[log]), int(batch.log_off[log + 1])
            rk = res.elem_rank[b0:b1].copy()
            rk[rk != 0xFFFFFFFF] &= abi.RANK_MASK
            ins = np.flatnonzero(batch.action[b0:b1] == abi.ACT_INSERT)
            assert sorted(rk[ins].tolist()) == list(range(len(ins)))  # a permutation of 0..n-1
            assert (rk[np.setdiff1d(np.arange(b1 - b0), ins)] == 0xFFFFFFFF).all()
            dele = set()
            key = {int(batch.op_id[b0 + i]): i for i in range(b1 - b0)}
            for i in np.flatnonzero(batch.action[b0:b1] == abi.ACT_DELETE):
                dele.add(key[int(batch.ref_a[b0 + i])])
            alive = [i for i in ins[np.argsort(rk[ins])] if i not in dele]
            assert [batch.values[int(batch.payload[b0 + i])] for i in alive] == exp["text"]
            log += 1


def test_log_header_census_paths(eng):
    """Batches without ptx_log_hdr get the device census pre-pass; lying headers are rejected per log."""
    gen = _load("ptxgen_mini.json")
    batch = wire.encode_docs([d["logs"] for d in gen["docs"]])
    a = eng.apply_materialize(batch)
    hdr = batch.log_hdr
    batch.log_hdr = None
    b = eng.apply_materialize(batch)
    assert (a.logs == b.logs).all() and (a.logs["status"] == 0).all()
    bad = hdr.copy()
    bad["n_ins"][0] += 1
    bad["max_counter"][1] -= 2
    batch.log_hdr = bad
    c = eng.apply_materialize(batch)
    assert [int(x) for x in c.logs["status"][:3]] == [abi.ERR_BAD_OP, abi.ERR_BAD_OP, 0]
    assert (c.logs[2:] == a.logs[2:]).all()


@pytest.mark.skipif(not H.have_node(), reason="node (oracle runtime) not installed")
def test_causal_admission_on_device(eng):
    """applyChange's admission rule (seq contiguity + deps, micromerge.ts:499-511) runs in the kernel whenever the
    batch carries the Change envelope: mutated logs fail with the reference's RangeError class (live oracle replay),
    intact logs are untouched, and PTX_FLAG_NO_ADMISSION switches the check off."""
    import copy

    from peritext_amd.engine import Engine
    from test_emu_parity import _expected_status

    gen = _load("ptxgen_mini.json")
    base = gen["docs"][0]["logs"][1]
    drop = copy.deepcopy(base)
    del drop[3]
    skip = copy.deepcopy(base)
    skip[5]["seq"] += 1
    dep = copy.deepcopy(base)
    other = [a for a in {c["actor"] for c in dep} if a != dep[2]["actor"]][0]
    dep[2]["deps"][other] = 10 ** 6
    logs = [drop, skip, dep, base]
    exp = H.oracle_apply([[l] for l in logs])
    want = [_expected_status(e[0]) for e in exp]
    assert want == [abi.ERR_SEQ_GAP, abi.ERR_SEQ_GAP, abi.ERR_MISSING_DEP, 0]
    batch = wire.encode_docs([[l] for l in logs])
    res = eng.apply_materialize(batch)
    assert [int(x) for x in res.logs["status"]] == want
    H.check_log(batch, res, 3, exp[3][0])
    with Engine(0, flags=abi.FLAG_NO_ADMISSION) as e2:
        res2 = e2.apply_materialize(batch)
        assert int(res2.logs["status"][1]) == 0 and int(res2.logs["status"][2]) == 0  # envelope ignored: the ops themselves are fine
        assert (res2.logs[3]["digest"] == res.logs[3]["digest"]).all() and int(res2.logs[3]["status"]) == 0


def test_admission_fast_check_passes_every_valid_log_on_the_device(eng):
    """The one-pass admission check (packed 16-bit relative clocks per wave, the per-wave numbers validated after the pass) is the
    part of P0 the CPU emulation cannot play (its waves have one lane): on the device no VALID log may fall back to the exact
    walk — the diagnostic build counts those in slot 15 of the phase clocks — while every log with a perturbed envelope must."""
    gen = _load("ptxgen_config4_600.json")
    base = wire.encode_docs([d["logs"] for d in gen["docs"]]).tile(40)
    db = eng.upload(base)
    dr = eng.alloc_result(db)
    try:
        eng.merge(db, dr)
        assert eng.phase_cycles(db, dr)[15] == 0
        assert int(eng.download_logs(dr, base.n_logs)["status"].max()) == 0
    finally:
        eng.free_result(dr)
        eng.free_batch(db)
    es = abi.env_stride(base.max_actors)
    rng = np.random.default_rng(5)
    env = base.chg_env.copy().reshape(-1, es)
    for log in range(base.n_logs):
        c = int(rng.integers(int(base.chg_off[log]), int(base.chg_off[log + 1])))
        env[c, 0] = np.uint16(int(env[c, 0]) + 1)  # every log: one seq too large
    base.chg_env = env.reshape(-1)
    db = eng.upload(base)
    dr = eng.alloc_result(db)
    try:
        eng.merge(db, dr)
        assert eng.phase_cycles(db, dr)[15] == base.n_logs
        assert (eng.download_logs(dr, base.n_logs)["status"] == abi.ERR_SEQ_GAP).all()
    finally:
        eng.free_result(dr)
        eng.free_batch(db)


@pytest.mark.skipif(not H.have_node(), reason="node (oracle runtime) not installed")
def test_many_actor_documents(eng):
    """> 4 actors per document: the library launches the kernel build that carries the table-based admission."""
    gen = H.oracle_gen("mini", 2, 3, None, 6)
    batch, res = H.check_generated(gen, eng.apply_materialize)
    assert batch.max_actors > 4
    broken = [c for c in gen["docs"][0]["logs"][2]]
    del broken[4]
    b2 = wire.encode_docs([[broken], gen["docs"][0]["logs"]])
    r2 = eng.apply_materialize(b2)
    assert int(r2.logs["status"][0]) in (abi.ERR_SEQ_GAP, abi.ERR_MISSING_DEP) and (r2.logs["status"][1:] == 0).all()


def test_error_statuses(eng):
    from test_emu_parity import _mini_doc

    docs = [
        [_mini_doc([{"action": "set", "insert": True, "elemId": "77@zz", "value": "x"}])],
        [_mini_doc([{"action": "del", "elemId": "77@zz"}])],
        [_mini_doc([{"action": "set", "insert": True, "elemId": "6@a", "value": "ok"}])],
    ]
    batch = wire.encode_docs(docs)
    res = eng.apply_materialize(batch)
    assert [int(s) for s in res.logs["status"]] == [abi.ERR_ELEM_NOT_FOUND, abi.ERR_ELEM_NOT_FOUND, 0]
    assert wire.decode_spans(batch, res, 2) == [{"text": "ABCDEok", "marks": {}}]


def test_empty_batch_and_empty_logs(eng):
    from test_emu_parity import _mini_doc

    batch = wire.encode_docs([[_mini_doc([], first_text="")], [[]]])  # a doc with no text, a log with no changes
    res = eng.apply_materialize(batch)
    assert [int(s) for s in res.logs["status"]] == [0, 0]
    assert wire.decode_spans(batch, res, 0) == [] and wire.decode_spans(batch, res, 1) == []
    empty = wire.encode_docs([])
    res = eng.apply_materialize(empty)
    assert len(res.logs) == 0


@pytest.mark.skipif(not H.have_node(), reason="node (oracle runtime) not installed")
@pytest.mark.parametrize("config,docs,ops,seed", [("mini", 64, None, 5), ("config2", 64, None, 6), ("config3", 8, None, 7), ("config4", 2, None, 8), ("rich", 3, None, 9), ("config5", 1, 4000, 10)])
def test_live_oracle_fresh_seeds(eng, config, docs, ops, seed):
    """The oracle generates fresh traces on this box; the HIP path must reproduce them bit-exactly
    (config4 here is the FULL BASELINE shape: 3 replicas x 4096 ops per document)."""
    H.check_generated(H.oracle_gen(config, docs, seed, ops), eng.apply_materialize)


@pytest.mark.skipif(not H.have_node(), reason="node (oracle runtime) not installed")
def test_baseline_size_properties(eng):
    """At BASELINE batch shape (thousands of logs of 4096 ops, tiled from unique documents): every log ok,
    the three replicas of every document converge to one digest, copies of a document agree, distinct
    documents differ, and a sampled log is bit-exact against the oracle."""
    gen = H.oracle_gen("config4", 8, 99)
    batch = wire.encode_docs([d["logs"] for d in gen["docs"]])
    copies = 64  # 8 docs x 3 replicas x 64 = 1536 logs x 4096 ops = 6.3 M ops, 201 MB of op log
    db = eng.upload(batch, copies=copies)
    dr = eng.alloc_result(db)
    try:
        eng.merge(db, dr)
        logs = eng.download_logs(dr, copies * batch.n_logs)
        got = eng.download(db, dr)
    finally:
        eng.free_result(dr)
        eng.free_batch(db)
    assert (logs["status"] == 0).all()
    dg = logs["digest"].reshape(copies, len(gen["docs"]), 3, 2)
    assert (dg == dg[:, :, :1, :]).all(), "replicas of a document must converge"
    assert (dg == dg[:1]).all(), "copies of a document must agree"
    assert len({tuple(x) for x in dg[0, :, 0, :].tolist()}) == len(gen["docs"])
    assert int(logs["n_ops"].sum()) == copies * batch.counted_ops()
    tiled = batch.tile(copies)
    for log in (0, 17 * batch.n_logs + 5, copies * batch.n_logs - 1):
        d, r = divmod(log % batch.n_logs, 3)
        H.check_log(tiled, got, log, gen["docs"][d]["expected"][r])


# ---- Patch[] streams: ptx_replay_patches through the C ABI ----
PATCH_GOLDEN = ["patches_mini.json", "patches_rich_300.json"]


def _streams(eng, batch):
    db = eng.upload(batch)
    dr = eng.alloc_result(db)
    try:
        eng.merge(db, dr)
        eng.sync()
        res = eng.download(db, dr)
        return res, eng.replay_patches(db, dr)
    finally:
        eng.free_result(dr)
        eng.free_batch(db)


@pytest.mark.parametrize("name", PATCH_GOLDEN)
def test_golden_patch_streams(eng, name):
    """Fixtures made by the reference itself (oracle/gen_patch_golden.js --impl ref): every patch every applyChange
    returns, in order, deep-equal."""
    g = _load(name)
    assert g["impl"] == "ref"
    batch = wire.encode_docs([d["logs"] for d in g["docs"]])
    res, pat = _streams(eng, batch)
    assert H.check_patch_streams(batch, pat, [d["expected"] for d in g["docs"]]) == batch.n_logs
    assert pat.kernel_ms > 0
    assert pat.launches == 1
    assert np.array_equal(pat.patch_off[1:] - pat.patch_off[:-1], np.where(pat.logs["status"] == 0, pat.logs["n_patches"], 0).astype(pat.patch_off.dtype))  # packed to exact offsets
    if name == "patches_rich_300.json":
        # more than two records per op: these logs outgrow the capacity the library guesses, take an overflow extent inside the ONE launch and are packed afterwards
        rows = np.diff(batch.log_off.astype(np.int64))
        assert np.any(pat.logs["n_patches"].astype(np.int64) > 2 * rows + 16)


def test_patch_streams_when_device_memory_is_short(eng, monkeypatch):
    """the one-launch replay asks for its guessed capacities, an arena of the same size and a packed copy; where the device has no room for the
    arena the capacities alone must do (a log that outgrows its own is replayed again with exact sizes: two launches), and where it has none for the whole packed
    copy the streams are packed and downloaded a range of logs at a time.  Both are played through the library's test switches on the reference-made fixture whose
    logs overflow the guess; the streams must equal the reference's, patch for patch."""
    g = _load("patches_rich_300.json")
    batch = wire.encode_docs([d["logs"] for d in g["docs"]])
    expected = [d["expected"] for d in g["docs"]]
    _, whole = _streams(eng, batch)
    monkeypatch.setenv("PTX_REPLAY_NO_ARENA", "1")
    _, pat = _streams(eng, batch)
    assert pat.launches == 2 and H.check_patch_streams(batch, pat, expected) == batch.n_logs
    monkeypatch.delenv("PTX_REPLAY_NO_ARENA")
    monkeypatch.setenv("PTX_REPLAY_PACK_RECORDS", "1")  # (never smaller than the longest log's stream: a log at a time)
    _, pat = _streams(eng, batch)
    assert pat.launches == 1 and H.check_patch_streams(batch, pat, expected) == batch.n_logs
    assert np.array_equal(pat.patch_off, whole.patch_off) and np.array_equal(pat.patches, whole.patches)
    monkeypatch.setenv("PTX_REPLAY_PACK_RECORDS", str(int(whole.patch_off[-1]) // 3))
    _, pat = _streams(eng, batch)
    assert np.array_equal(pat.patch_off, whole.patch_off) and np.array_equal(pat.patches, whole.patches)


def test_marks_that_arrive_after_larger_op_ids(eng):
    """GPU twin of test_emu_patches.py's: three actors mark one text concurrently and a replica applies them in descending id order — almost every op meets
    larger ids applied before it (the replay's table scan instead of per-slot winners), every link / comment state is met."""
    if not H.have_node():
        pytest.skip("node not installed")
    docs = H.concurrent_marks_docs()
    expected = H.oracle_apply(docs, patches=True)
    batch = wire.encode_docs(docs)
    res, pat = _streams(eng, batch)
    assert H.check_patch_streams(batch, pat, expected) == batch.n_logs


def test_patch_streams_with_op_counters_beyond_the_dense_key_range(eng):
    """The replay keeps the LWW winners of strong / em per slot as dense op-id keys where the log's id space fits 16 bits and as rows where it
    does not: the reference-made fixture with every counter moved up by 70 000 takes the second path and must give the same streams."""
    g = _load("patches_rich_300.json")
    wide = wire.encode_docs(H.shift_counters([d["logs"] for d in g["docs"]], 70000))
    res, pat = _streams(eng, wide)
    assert (res.logs["status"] == 0).all()
    assert H.check_patch_streams(wide, pat, [d["expected"] for d in g["docs"]]) == wide.n_logs


def test_patch_streams_kat_traces_and_failed_logs(eng):
    if not H.have_node():
        pytest.skip("node not installed")
    cases = H.load_kat()
    bad = H.mini_doc([{"action": "del", "elemId": "77@zz"}])
    docs = [[r["log"] for r in c["replicas"]] for c in cases] + [t["logs"] for t in _load("reference_traces.json")]
    expected = H.oracle_apply(docs, patches=True)
    batch = wire.encode_docs(docs + [[bad]])
    res, pat = _streams(eng, batch)
    H.check_patch_streams(batch, pat, expected)
    last = batch.n_logs - 1
    assert int(pat.logs[last]["status"]) == abi.ERR_ELEM_NOT_FOUND and int(pat.logs[last]["n_patches"]) == 0
    with pytest.raises(ValueError, match="List element not found"):
        wire.decode_patches(batch, pat, last)


@pytest.mark.parametrize("config,docs,ops,seed", [("mini", 8, None, 31), ("rich", 1, 800, 32), ("config4", 1, None, 33), ("config5", 1, 1500, 34)])
def test_patch_streams_live_oracle_and_accumulate(eng, config, docs, ops, seed):
    """Fresh PTXGEN documents (config4: full 4 096-op logs): streams equal the oracle's, and replaying a stream per
    character (reference/test/accumulatePatches.ts) gives the batch result of the same log."""
    if not H.have_node():
        pytest.skip("node not installed")
    g = H.oracle_gen(config, seed=seed, docs=docs, ops=ops)
    dl = [d["logs"] for d in g["docs"]]
    expected = H.oracle_apply(dl, patches=True)
    batch = wire.encode_docs(dl)
    res, pat = _streams(eng, batch)
    H.check_patch_streams(batch, pat, expected)
    for log in range(batch.n_logs):
        got = H.accumulate_patches(wire.decode_patches(batch, pat, log, with_rows=True))
        assert H.norm_spans(got) == H.norm_spans(wire.decode_spans(batch, res, log)), "log %d" % log


def test_tail_patch_streams_are_the_suffix_of_the_whole_stream(eng):
    """ptx_replay_patches_from (what a host with resident replicas asks for after appending Changes): the records from first_row[l] on = the tail of the
    log's whole stream, for cuts at row 0, inside the log, at the last row and past the end."""
    g = _load("patches_rich_300.json")
    batch = wire.encode_docs([d["logs"] for d in g["docs"]])
    db = eng.upload(batch)
    dr = eng.alloc_result(db)
    try:
        eng.merge(db, dr)
        eng.sync()
        whole = eng.replay_patches(db, dr)
        sizes = np.diff(batch.log_off.astype(np.int64))
        rng = np.random.default_rng(11)
        for first in (np.zeros_like(sizes), rng.integers(0, np.maximum(sizes, 1)), np.maximum(sizes - 1, 0), sizes + 5):
            tail = eng.replay_patches(db, dr, first_row=first)
            assert np.array_equal(tail.logs["status"], whole.logs["status"])
            for log in range(batch.n_logs):
                a = whole.patches[int(whole.patch_off[log]):int(whole.patch_off[log]) + int(whole.logs["n_patches"][log])]
                b = tail.patches[int(tail.patch_off[log]):int(tail.patch_off[log]) + int(tail.logs["n_patches"][log])]
                assert np.array_equal(b, a[a["row"] >= first[log]]), (log, int(first[log]))
        with pytest.raises(ValueError):
            eng.replay_patches(db, dr, first_row=[0])
    finally:
        eng.free_result(dr)
        eng.free_batch(db)


def test_patch_streams_need_elem_rank_and_handle_empty(eng):
    from peritext_amd.engine import Engine, PtxError

    batch = wire.encode_docs([[H.mini_doc([])]])
    e2 = Engine(0, flags=abi.FLAG_NO_ELEM_RANK)
    try:
        db = e2.upload(batch)
        dr = e2.alloc_result(db)
        e2.merge(db, dr)
        with pytest.raises(PtxError, match="elem_rank"):
            e2.replay_patches(db, dr)
        e2.free_result(dr)
        e2.free_batch(db)
    finally:
        e2.close()
    empty = wire.encode_docs([])
    res, pat = _streams(eng, empty)
    assert len(pat.logs) == 0 and len(pat.patches) == 0


# ---- on-device change(): ptx_generate through the C ABI ----
def _generate(eng, cfg, n_docs, seed, first_doc=0, list_cap=0):
    h, info = eng.generate(cfg["replicas"], cfg["ops_per_log"], cfg["mix"], cfg["mark_types"], n_docs, seed, first_doc=first_doc, list_cap=list_cap)
    actors, comments, log_doc = wire.generated_tables(n_docs, cfg["replicas"], info["n_comments"])
    batch = eng.download_batch(h, wire.GEN_VALUES, wire.GEN_URLS, log_doc, actors, comments)
    return h, batch, info


def _check_changes(batch, docs_logs):
    norm = lambda x: json.loads(json.dumps(x, sort_keys=True))  # noqa: E731
    log = 0
    for logs in docs_logs:
        for want in logs:
            got = wire.decode_changes(batch, log)
            assert len(got) == len(want), "log %d: %d changes, expected %d" % (log, len(got), len(want))
            for i, (x, y) in enumerate(zip(got, want)):
                assert norm(x) == norm(y), "log %d change %d" % (log, i)
            log += 1
    assert log == batch.n_logs


@pytest.mark.parametrize("name,cfg", [("ptxgen_mini.json", "mini"), ("ptxgen_config3_512.json", "config3"), ("ptxgen_config4_600.json", "config4"),
                                       ("ptxgen_rich_700.json", "rich")])
def test_generate_reproduces_the_ptxgen_fixtures_and_merges_them(eng, name, cfg):
    """Given only (config, seed, document index) the device produces, change for change, the logs the oracle's change() made
    (committed fixtures); the resident batch then merges to the fixtures' spans without visiting the host."""
    g = _load(name)
    c = H.gen_config(cfg, ops=g["cfg"]["opsPerLog"], replicas=g["cfg"]["replicas"])
    h, batch, info = _generate(eng, c, len(g["docs"]), g["seed"], first_doc=g["docs"][0]["docIndex"])
    try:
        assert info["kernel_ms"] > 0
        _check_changes(batch, [d["logs"] for d in g["docs"]])
        dr = eng.alloc_result(h)
        eng.merge(h, dr)
        eng.sync()
        res = eng.download(h, dr)
        eng.free_result(dr)
        log = 0
        for d in g["docs"]:
            digests = set()
            for exp in d["expected"]:
                assert int(res.logs[log]["status"]) == 0
                assert H.norm_spans(wire.decode_spans(batch, res, log)) == H.norm_spans(exp["spans"]), "log %d" % log
                digests.add((int(res.logs[log]["digest"][0]), int(res.logs[log]["digest"][1])))
                log += 1
            assert len(digests) == 1  # the replicas of a document converge
    finally:
        eng.free_batch(h)


def test_generate_full_config4_document_against_live_oracle(eng):
    if not H.have_node():
        pytest.skip("node not installed")
    g = H.oracle_gen("config4", seed=41, docs=1)
    h, batch, info = _generate(eng, H.gen_config("config4"), 1, 41)
    try:
        _check_changes(batch, [d["logs"] for d in g["docs"]])
    finally:
        eng.free_batch(h)


@pytest.mark.skipif(not H.have_node(), reason="node (oracle runtime) not installed")
@pytest.mark.parametrize("cfg,replicas,ops,docs,seed", [("mini", 5, 200, 4, 95), ("rich", 8, 500, 2, 96), ("config4", 6, 900, 1, 98)])
def test_generate_documents_of_five_to_eight_replicas(eng, cfg, replicas, ops, docs, seed):
    """Round 5: the generator holds up to 8 replicas per document (three actor bits per key, four words of dependencies per change).  Change for Change the
    oracle's PTXGEN; the resident batch is then admitted by the many-actor builds and merges to the oracle's spans; every replica of a document converges."""
    g = H.oracle_gen(cfg, seed=seed, docs=docs, ops=ops, replicas=replicas)
    h, batch, info = _generate(eng, H.gen_config(cfg, ops=ops, replicas=replicas), docs, seed, list_cap=4096)
    try:
        _check_changes(batch, [d["logs"] for d in g["docs"]])
        assert eng.batch_kernel_name(h) == ("ptx_merge_kernel_many" if replicas <= 7 else "ptx_merge_kernel_many_wide")
        dr = eng.alloc_result(h)
        eng.merge(h, dr)
        eng.sync()
        res = eng.download(h, dr)
        eng.free_result(dr)
        log = 0
        for d in g["docs"]:
            for exp in d["expected"]:
                H.check_log(batch, res, log, exp)
                log += 1
        dg = res.logs["digest"].reshape(docs, replicas, 2)
        assert (dg == dg[:, :1, :]).all()
    finally:
        eng.free_batch(h)


def test_generate_capacity_and_arguments(eng):
    from peritext_amd.engine import PtxError

    c = H.gen_config("mini")
    with pytest.raises(PtxError, match="list_cap"):
        _generate(eng, c, 4, 5, list_cap=8)
    with pytest.raises(PtxError, match="replicas"):
        eng.generate(9, 16, [25, 25, 25, 25], [0], 1, 1)
    h, batch, info = _generate(eng, c, 0, 1)
    assert batch.n_logs == 0
    eng.free_batch(h)


def test_split_launch_when_a_few_logs_need_more_lds(eng):
    """A batch of many small logs and a few large ones is merged in two launches (the dynamic LDS size is per launch: sized for
    the large logs it would cost every log a share of the CU); results are those of the oracle for every log of both groups."""
    big, small = _load("ptxgen_rich_2600.json"), _load("ptxgen_config4_600.json")
    docs = [d for d in big["docs"]] + [d for _ in range(12) for d in small["docs"]]
    batch = wire.encode_docs([d["logs"] for d in docs])
    db = eng.upload(batch)
    dr = eng.alloc_result(db)
    try:
        eng.merge(db, dr)
        eng.sync()
        res = eng.download(db, dr)
        threads, lds_main = eng.launch_shape(db)
        lds_high = res.logs["reserved"][:, 0]
        n_big = sum(len(d["logs"]) for d in big["docs"])
        assert int(lds_high[:n_big].max()) > lds_main >= int(lds_high[n_big:].max()), "the large logs ran in a launch of their own"
        log = 0
        for d in docs:
            for exp in d["expected"]:
                H.check_log(batch, res, log, exp)
                log += 1
    finally:
        eng.free_result(dr)
        eng.free_batch(db)


def test_streaming_append_to_resident_logs(eng):
    """the changes that arrived since are appended to the resident logs device to device; the grown batch is, column for
    column, the batch of the whole logs, and merges to the oracle's spans with the same digests and Patch[] streams."""
    g = _load("ptxgen_rich_700.json")
    docs = [d["logs"] for d in g["docs"]]
    full = wire.encode_docs(docs)
    nch = np.diff(full.chg_off.astype(np.int64))
    head, tail = wire.split_batch(full, nch * 2 // 3)
    assert 0 < head.n_ops < full.n_ops and head.n_ops + tail.n_ops == full.n_ops
    db_full = eng.upload(full)
    db_head = eng.upload(head)
    db_grown = eng.append(db_head, tail)
    try:
        assert eng.n_ops(db_grown) == full.n_ops and eng.n_changes(db_grown) == int(full.chg_off[-1])
        got = eng.download_batch(db_grown)
        for k in ("log_off", "op_id", "ref_a", "ref_b", "payload", "action", "mark_type", "side_a", "side_b", "chg_off", "chg_hdr", "chg_env"):
            assert np.array_equal(getattr(got, k), getattr(full, k)), k
        assert np.array_equal(got.log_hdr, full.log_hdr)
        out = {}
        for name, db in (("full", db_full), ("grown", db_grown)):
            dr = eng.alloc_result(db)
            eng.merge(db, dr)
            eng.sync()
            out[name] = (eng.download(db, dr), eng.replay_patches(db, dr))
            eng.free_result(dr)
        assert np.array_equal(out["grown"][0].logs["digest"], out["full"][0].logs["digest"])
        log = 0
        for d in g["docs"]:
            for exp in d["expected"]:
                H.check_log(full, out["grown"][0], log, exp)
                a, b = wire.decode_patches(full, out["full"][1], log), wire.decode_patches(full, out["grown"][1], log)
                assert H.norm_patches(a) == H.norm_patches(b)
                log += 1
    finally:
        for h in (db_full, db_head, db_grown):
            eng.free_batch(h)


def test_admission_fast_check_agrees_with_a_sequential_replay(eng):
    """GPU twin of the emulation test: the one-pass admission check (64-lane waves, 256 changes per wave step) against a sequential
    replay of the envelope, on full config-4-sized logs with one envelope word perturbed at random, untouched logs in between."""
    from test_emu_parity import _sequential_admission

    h, info = eng.generate(3, 4096, [25, 25, 25, 25], [0, 1, 3, 2], 40, 123, list_cap=2048)
    actors_t, comments_t, log_doc_t = wire.generated_tables(40, 3, info["n_comments"])
    batch = eng.download_batch(h, wire.GEN_VALUES, wire.GEN_URLS, log_doc_t, actors_t, comments_t)
    eng.free_batch(h)
    es = abi.env_stride(batch.max_actors)
    rng = np.random.default_rng(5)
    env = batch.chg_env.copy().reshape(-1, es)
    for log in range(batch.n_logs):
        if log % 7 == 0:
            continue
        c = int(rng.integers(int(batch.chg_off[log]), int(batch.chg_off[log + 1])))
        col = int(rng.integers(0, 1 + batch.max_actors))
        env[c, col] = np.uint16(max(0, min(65535, int(env[c, col]) + int(rng.choice([-2, -1, 1, 2, 40000])))))
    batch.chg_env = env.reshape(-1)
    res = eng.apply_materialize(batch)
    want = [_sequential_admission(batch, log) for log in range(batch.n_logs)]
    assert [int(x) for x in res.logs["status"]] == want
    assert want.count(0) >= 18 and want.count(abi.ERR_SEQ_GAP) > 5 and want.count(abi.ERR_MISSING_DEP) > 5


@pytest.mark.skipif(not H.have_node(), reason="node (oracle runtime) not installed")
@pytest.mark.parametrize("replicas", [4, 5, 7, 8, 11, 12, 15])
def test_admission_walk_of_four_to_fifteen_actors(eng, replicas):
    """Round 5, GPU twin of the emulation test: documents of four to fifteen actors are admitted by a one-pass walk (64-lane waves, the vector clock in 4 / 6 / 8
    packed words: DPP prefix sums, v_perm selects, packed 16-bit arithmetic) — pass / fail exactly like a sequential replay on 2 000-op logs with one envelope
    word perturbed at random; the documents themselves against the oracle; the failing row of a dropped change comes from the table, as before."""
    from test_emu_parity import _sequential_admission

    gen = H.oracle_gen("mini", 3, 60 + replicas, 2000, replicas)
    base, res0 = H.check_generated(gen, eng.apply_materialize)
    assert base.max_actors == replicas
    batch = base.tile(6)
    env = batch.chg_env.copy().reshape(-1, abi.env_stride(replicas))
    rng = np.random.default_rng(replicas)
    for log in range(batch.n_logs):
        if log % 5 == 0:
            continue
        c = int(rng.integers(int(batch.chg_off[log]), int(batch.chg_off[log + 1])))
        col = int(rng.integers(0, 1 + replicas))
        env[c, col] = np.uint16(max(0, min(65535, int(env[c, col]) + int(rng.choice([-2, -1, 1, 2, 40000])))))
    batch.chg_env = env.reshape(-1)
    db = eng.upload(batch)
    try:
        assert eng.batch_kernel_name(db) == ("ptx_merge_kernel_many" if replicas <= 7 else "ptx_merge_kernel_many_wide")
    finally:
        eng.free_batch(db)
    res = eng.apply_materialize(batch)
    want = [_sequential_admission(batch, log) for log in range(batch.n_logs)]
    assert [int(x) for x in res.logs["status"]] == want
    assert want.count(0) >= batch.n_logs // 5 and want.count(abi.ERR_SEQ_GAP) > 3 and want.count(abi.ERR_MISSING_DEP) > 3
    ok = [l for l, w in enumerate(want) if w == 0]
    assert all((res.logs["digest"][l] == res0.logs["digest"][l % base.n_logs]).all() for l in ok)
