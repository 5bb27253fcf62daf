"""GPU twins of the emulation-only suites (VERDICT r1 'parity gaps'): the hand-written edge cases, the duplicate-op and
huge-bucket paths, unsynced replicas, cursors, and EVERY launch shape of ptx_merge_kernel (64..512 threads per log, the
48 KB and the 160 KB LDS windows) on reference-made fixtures — through the C ABI on a real MI355X.

Expected outputs come from committed fixtures made by the reference itself in the build container
(tests/make_edge_golden.py, oracle/cli.js gen --impl ref): no node and no /root/reference needed on the GPU box."""
import json
import os

import numpy as np
import pytest

import helpers as H
from peritext_amd import abi, wire

pytestmark = pytest.mark.gpu


def _load(name):
    with open(os.path.join(H.GOLDEN, name)) as f:
        return json.load(f)


@pytest.fixture(scope="module")
def eng():
    import torch

    assert torch.cuda.is_available(), "GPU tests need a real MI355X"
    from peritext_amd.engine import Engine

    e = Engine(0)
    yield e
    e.close()


@pytest.fixture(scope="module")
def golden():
    g = _load("edge_cases_ref.json")
    assert g["impl"] == "ref"
    return g


def _streams(eng, batch):
    db = eng.upload(batch)
    dr = eng.alloc_result(db)
    try:
        eng.merge(db, dr)
        eng.sync()
        return eng.download(db, dr), eng.replay_patches(db, dr)
    finally:
        eng.free_result(dr)
        eng.free_batch(db)


def test_edge_cases(eng, golden):
    """removeMark comment -> `comment: []`, zero-width marks, unknown boundary element, endOfText, empty
    document, everything deleted, add/remove/add of one comment id."""
    docs = H.edge_case_docs()
    batch = wire.encode_docs(docs)
    res = eng.apply_materialize(batch)
    for log, exp in enumerate(golden["edge"]):
        H.check_log(batch, res, log, exp[0])
    assert golden["edge"][0][0]["spans"][1]["marks"] == {"comment": []}


def test_boundaries_changemark_never_generates(eng, golden):
    """helpers.boundary_docs on the device: startOfText as a start and as an end, endOfText as a start, `after` start on an inclusive mark, `before` end on a
    link — spans and the patch streams of ptx_replay_patches against what the type-erased reference gave (edge_cases_ref.json)."""
    docs = H.boundary_docs()
    batch = wire.encode_docs(docs)
    assert int((batch.side_a == 2).sum()) >= 3 and int((batch.side_b == 2).sum()) >= 2
    db = eng.upload(batch)
    dr = eng.alloc_result(db)
    try:
        eng.merge(db, dr)
        eng.sync()
        res = eng.download(db, dr)
        pat = eng.replay_patches(db, dr)
    finally:
        eng.free_result(dr)
        eng.free_batch(db)
    for log, exp in enumerate(golden["boundary"]):
        H.check_log(batch, res, log, exp[0])
    assert H.check_patch_streams(batch, pat, golden["boundary"]) == len(docs)


def test_huge_sibling_bucket(eng, golden):
    log = H.huge_bucket_log()
    batch = wire.encode_docs([[log]])
    res = eng.apply_materialize(batch)
    H.check_log(batch, res, 0, golden["huge_bucket"][0][0])
    if H.have_node():  # beyond PTX_HUGE_BUCKET (256) children of HEAD: the bitmap-ranked path, against a live run of the oracle
        big = H.huge_bucket_log(n_head=300)
        b2 = wire.encode_docs([[big]])
        H.check_log(b2, eng.apply_materialize(b2), 0, H.oracle_apply([[big]])[0][0])


def test_duplicate_op_id(eng):
    batch = wire.encode_docs(H.duplicate_op_docs())
    res = eng.apply_materialize(batch)
    assert [int(x) for x in res.logs["status"]] == [abi.ERR_DUPLICATE_OP, 0]
    assert wire.decode_spans(batch, res, 1) == [{"text": "ABCDEx", "marks": {}}]


def test_element_that_only_appears_later(eng):
    docs = [[H.mini_doc([{"action": "del", "elemId": "9@a"}, {"action": "set", "insert": True, "elemId": "6@a", "value": "x"},
                         {"action": "set", "insert": True, "elemId": "6@a", "value": "y"}])]]
    batch = wire.encode_docs(docs)
    res = eng.apply_materialize(batch)
    assert int(res.logs["status"][0]) == abi.ERR_ELEM_NOT_FOUND


def test_wide_id_keys_give_the_same_documents(eng, golden):
    """Twin of the emulation test: counters moved up by 70 000 (32-bit id keys in P3a), same documents."""
    with open(os.path.join(H.GOLDEN, "ptxgen_mini.json")) as f:
        gen = json.load(f)
    docs = [d["logs"] for d in gen["docs"][:4]] + H.more_deletes_than_inserts_docs()
    base, wide = wire.encode_docs(docs), wire.encode_docs(H.shift_counters(docs, 70000))
    r0, r1 = eng.apply_materialize(base), eng.apply_materialize(wide)
    assert (r0.logs["status"] == r1.logs["status"]).all()
    for log in range(len(r0.logs["status"])):
        if int(r0.logs["status"][log]) == 0:
            assert wire.decode_spans(wide, r1, log) == wire.decode_spans(base, r0, log)
    for log, exp in enumerate(e for d in gen["docs"][:4] for e in d["expected"]):
        H.check_log(base, r0, log, exp)


def test_more_deletes_than_inserts(eng):
    """Twin of the emulation test: deletes beyond slot n are resolved in their own loop; order check of both kinds of deletes."""
    batch = wire.encode_docs(H.more_deletes_than_inserts_docs())
    res = eng.apply_materialize(batch)
    assert [int(x) for x in res.logs["status"]] == [0, abi.ERR_ELEM_NOT_FOUND, abi.ERR_ELEM_NOT_FOUND]
    assert wire.decode_spans(batch, res, 0) == [{"text": "!", "marks": {}}]
    assert [int(x) for x in res.logs["reserved"][1:, 1]] == [15, 15]


def test_unsynced_replicas_with_different_comment_sets(eng, golden):
    """replicas that have seen different subsets of the document's comments (prefixes of replica logs; two
    replicas that each know one comment the other lacks) — spans, digests and Patch[] streams."""
    docs = H.unsynced_docs()
    batch = wire.encode_docs(docs)
    assert (batch.log_hdr["n_comment_ids"] > batch.log_hdr["n_mark"][:, abi.MARK_COMMENT]).any()
    res, pat = _streams(eng, batch)
    log = 0
    for exp in golden["unsynced"]:
        for e in exp:
            H.check_log(batch, res, log, e)
            log += 1
    H.check_patch_streams(batch, pat, golden["unsynced"])
    hdr = batch.log_hdr
    batch.log_hdr = None  # the device census finds the same id space
    res2 = eng.apply_materialize(batch)
    assert (res2.logs["digest"] == res.logs["digest"]).all() and (res2.logs["status"] == 0).all()
    bad = hdr.copy()
    l = int(np.flatnonzero(hdr["n_comment_ids"] > 1)[0])
    bad["n_comment_ids"][l] -= 1
    batch.log_hdr = bad
    res3 = eng.apply_materialize(batch)
    assert int(res3.logs["status"][l]) == abi.ERR_BAD_OP and (np.delete(res3.logs["status"], l) == 0).all()


def test_cursors_from_elem_rank(eng, golden):
    """getCursor / resolveCursor (micromerge.ts:465-477) from the elem_rank column the GPU produced."""
    gen = _load("ptxgen_mini.json")
    docs = [d["logs"] for d in gen["docs"][:3]]
    batch = wire.encode_docs(docs)
    res = eng.apply_materialize(batch)
    log = 0
    for d in golden["cursors"]:
        for e in d:
            assert [wire.get_cursor(batch, res, log, i) for i in range(len(e["text"]))] == e["cursorAt"]
            for elem, idx in e["cursorResolve"].items():
                assert wire.resolve_cursor(batch, res, log, elem) == idx, elem
            with pytest.raises(ValueError):
                wire.get_cursor(batch, res, log, len(e["text"]))
            log += 1


def test_cursors_resolved_on_the_device(eng, golden):
    """ptx_resolve_cursors: getCursor / resolveCursor (micromerge.ts:465-477) answered by the device for every visible
    index and every element ever inserted of three documents — the reference's own answers — plus the two RangeErrors."""
    gen = _load("ptxgen_mini.json")
    docs = [d["logs"] for d in gen["docs"][:3]]
    batch = wire.encode_docs(docs)
    db = eng.upload(batch)
    dr = eng.alloc_result(db)
    try:
        eng.merge(db, dr)
        eng.sync()
        q_log, q_kind, q_arg, want = H.cursor_queries(batch, golden["cursors"])
        out, status = eng.resolve_cursors(db, dr, q_log, q_kind, q_arg)
        H.check_cursor_answers(q_kind, want, out, status)
        assert len(want) > 300
        o0, s0 = eng.resolve_cursors(db, dr, [], [], [])
        assert len(o0) == 0
    finally:
        eng.free_result(dr)
        eng.free_batch(db)
    # the 8 192-op config-5 log (512-thread merge shape, 3 112 elements, one visible): every element resolves to 0 or 1
    g5 = _load("ptxgen_config5_8192.json")
    b5 = wire.encode_docs([d["logs"] for d in g5["docs"]])
    db = eng.upload(b5)
    dr = eng.alloc_result(db)
    try:
        eng.merge(db, dr)
        eng.sync()
        res = eng.download(db, dr)
        ins = np.flatnonzero(b5.action == abi.ACT_INSERT)
        out, status = eng.resolve_cursors(db, dr, [0] * len(ins), [abi.CURSOR_RESOLVE] * len(ins), b5.op_id[ins])
        assert (status == 0).all()
        assert [int(x) for x in out] == [wire.resolve_cursor(b5, res, 0, "%d@doc1" % (int(i) >> 32)) for i in b5.op_id[ins]]
    finally:
        eng.free_result(dr)
        eng.free_batch(db)


@pytest.mark.skipif(not H.have_node(), reason="node (oracle runtime) not installed")
@pytest.mark.parametrize("config,docs,ops,replicas,seed", [("mini", 6, None, None, 71), ("rich", 3, 160, None, 72), ("config4", 2, 220, None, 73), ("rich", 2, 120, 4, 74)])
def test_redealt_logs_against_the_oracle(eng, config, docs, ops, replicas, seed):
    """GPU twin of test_emu_partial_orders.py: random causally closed subsets of a document's changes in random linear extensions of the
    causal order — valid inputs of applyChange no replica of the generator ever applied — merged (causal admission on), replayed into patch
    streams and queried for cursors on the device, all against a live oracle replay."""
    import random

    gen = H.oracle_gen(config, docs=docs, seed=seed, ops=ops, replicas=replicas)
    rng = random.Random(seed)
    dealt = [H.redeal_logs(d["logs"], rng, 8) for d in gen["docs"]]
    expected = H.oracle_apply(dealt, patches=True, cursors=True)
    batch = wire.encode_docs(dealt)
    db = eng.upload(batch)
    dr = eng.alloc_result(db)
    try:
        eng.merge(db, dr)
        eng.sync()
        res = eng.download(db, dr)
        log = 0
        for exps in expected:
            for exp in exps:
                assert "error" not in exp, exp.get("error")
                H.check_log(batch, res, log, exp)
                log += 1
        assert log == batch.n_logs >= 4 * docs
        assert H.check_patch_streams(batch, eng.replay_patches(db, dr), expected) == batch.n_logs
        q_log, q_kind, q_arg, want = H.cursor_queries(batch, expected)
        out, status = eng.resolve_cursors(db, dr, q_log, q_kind, q_arg)
        H.check_cursor_answers(q_kind, want, out, status)
    finally:
        eng.free_result(dr)
        eng.free_batch(db)


def test_ten_actor_document_string_order(eng):
    """a1: "7@doc10" < "7@doc2" (compareOpIds compares actor STRINGS, micromerge.ts:826): a 10-replica fixture made by the
    reference; ranks follow the string order and the many-actor admission build runs."""
    g = _load("ptxgen_mini_10actors.json")
    assert g["impl"] == "ref"
    batch, res = H.check_generated(g, eng.apply_materialize)
    assert batch.doc_actors[0][:3] == ["doc1", "doc10", "doc2"] and batch.max_actors == 10


# ---- every launch shape ----
SHAPE_FIXTURES = ["ptxgen_config4_600.json", "ptxgen_rich_2600.json", "ptxgen_config5_8192.json", "ptxgen_mini_10actors.json"]


@pytest.mark.parametrize("lds", [0, 48 * 1024, 160 * 1024])
@pytest.mark.parametrize("threads", [64, 128, 192, 256, 512])
def test_every_launch_shape(threads, lds):
    """The merge kernel under every workgroup size the library ever picks (and the two LDS windows of config #5 / the CU
    maximum), with and without causal admission: reference-made fixtures incl. the full 8 192-op config-5 log, bit-exact."""
    from peritext_amd.engine import Engine

    for flags in (0, abi.FLAG_NO_ADMISSION):
        with Engine(0, flags=flags) as e:
            e.set_launch_shape(threads, lds)
            for name in SHAPE_FIXTURES:
                g = _load(name)
                batch = wire.encode_docs([d["logs"] for d in g["docs"]])
                db = e.upload(batch)
                dr = e.alloc_result(db)
                try:
                    t, l = e.launch_shape(db)
                    assert t == threads and (not lds or l == lds)
                    e.merge(db, dr)
                    res = e.download(db, dr)
                finally:
                    e.free_result(dr)
                    e.free_batch(db)
                log = 0
                for d in g["docs"]:
                    for exp in d["expected"]:
                        H.check_log(batch, res, log, exp)
                        log += 1


LEAN_FIXTURES = ["ptxgen_mini.json", "ptxgen_config2.json", "ptxgen_config3_512.json", "ptxgen_config4_600.json", "ptxgen_rich_700.json", "ptxgen_rich_2600.json"]


@pytest.mark.parametrize("threads", [64, 128, 192, 256])
def test_lean_builds_of_the_merge_kernel(threads):
    """Round 5: ptx_merge_kernel_lean64 / 128 / 192 — the builds for batches of 16-bit id keys merged without elem_rank (PTX_FLAG_NO_ELEM_RANK; the
    workgroup size a compile-time constant, a one-wave log without s_barrier).  The library must CHOOSE them for such batches (the kernel's name is
    asserted) and they must reproduce the reference-made fixtures bit for bit, with and without causal admission; a batch with wide id keys, or a
    context that wants elem_rank, keeps the general build."""
    from peritext_amd.engine import Engine

    for flags in (abi.FLAG_NO_ELEM_RANK, abi.FLAG_NO_ELEM_RANK | abi.FLAG_NO_ADMISSION):
        with Engine(0, flags=flags) as e:
            e.set_launch_shape(threads, 0)
            seen = set()
            for name in LEAN_FIXTURES:
                g = _load(name)
                batch = wire.encode_docs([d["logs"] for d in g["docs"]])
                db = e.upload(batch)
                dr = e.alloc_result(db)
                try:
                    # (the lean builds are held to 96 scalar registers like ptx_merge_kernel_w7: chosen where the LDS window lets more than 24 waves share a CU)
                    t, l = e.launch_shape(db)
                    waves_per_cu = (160 * 1024 // (-(-l // 1280) * 1280)) * (t // 64)
                    # (round 6: the four-wave build has no register cap — 8K-op logs and documents that keep their text, whose LDS window bounds the resident logs — and is taken whatever the window)
                    want = "ptx_merge_kernel_lean%d" % threads if waves_per_cu > 24 or threads == 256 else "ptx_merge_kernel"
                    assert e.batch_kernel_name(db) == want, name
                    seen.add(want)
                    e.merge(db, dr)
                    res = e.download(db, dr)
                finally:
                    e.free_result(dr)
                    e.free_batch(db)
                log = 0
                for d in g["docs"]:
                    for exp in d["expected"]:
                        H.check_log(batch, res, log, exp)
                        log += 1
            assert "ptx_merge_kernel_lean%d" % threads in seen
            # wide id keys (counters moved up by 70 000): not a lean batch, same documents
            gen = _load("ptxgen_mini.json")
            docs = [d["logs"] for d in gen["docs"][:4]]
            base, wide = wire.encode_docs(docs), wire.encode_docs(H.shift_counters(docs, 70000))
            db = e.upload(wide)
            try:
                assert "lean" not in e.batch_kernel_name(db)
            finally:
                e.free_batch(db)
            r0, r1 = e.apply_materialize(base), e.apply_materialize(wide)
            assert (r0.logs["status"] == 0).all() and (r0.logs["digest"] == r1.logs["digest"]).all()
    with Engine(0) as e:  # elem_rank wanted: the general build
        e.set_launch_shape(threads, 0)
        db = e.upload(wire.encode_docs([d["logs"] for d in _load("ptxgen_config4_600.json")["docs"]]))
        try:
            assert "lean" not in e.batch_kernel_name(db)
        finally:
            e.free_batch(db)


def test_default_shapes_cover_256_and_512_threads(eng):
    """What the library picks on its own: 192 threads up to 4 608 ops, 256 up to 12 288 (config #5's 8 192-op logs run 8 % faster as four waves than as eight,
    profiles/r04_i_*), 512 beyond."""
    g5 = _load("ptxgen_config5_8192.json")
    b5 = wire.encode_docs([d["logs"] for d in g5["docs"]])
    db = eng.upload(b5)
    try:
        assert eng.launch_shape(db)[0] == 256
    finally:
        eng.free_batch(db)
    H.check_generated(g5, eng.apply_materialize)
    if H.have_node():
        g = H.oracle_gen("config5", 1, 12, 5600)  # 5 601 rows -> the 256-thread shape too
        b = wire.encode_docs([d["logs"] for d in g["docs"]])
        db = eng.upload(b)
        try:
            assert eng.launch_shape(db)[0] == 256
        finally:
            eng.free_batch(db)
        H.check_generated(g, eng.apply_materialize)
    # 13 000-op logs made on the device: 512 threads by default; the same batch under a forced 256-thread shape gives the same digests (every forced shape is
    # checked against reference-made fixtures by test_every_launch_shape above)
    from peritext_amd import workloads
    from peritext_amd.engine import Engine

    c = workloads.gen_config("config5", ops=13000)
    digests = []
    for threads in (0, 256):
        with Engine(0) as e:
            e.set_launch_shape(threads, 0)
            db, _ = e.generate(c["replicas"], c["ops_per_log"], c["mix"], c["mark_types"], 4, 77)
            try:
                assert e.launch_shape(db)[0] == (512 if threads == 0 else 256)
                dr = e.alloc_result(db)
                e.merge(db, dr)
                res = e.download(db, dr)
                assert (res.logs["status"] == 0).all()
                digests.append(res.logs["digest"].copy())
                e.free_result(dr)
            finally:
                e.free_batch(db)
    assert np.array_equal(digests[0], digests[1])


def test_capacity_status_when_the_lds_window_is_too_small():
    from peritext_amd.engine import Engine

    g = _load("ptxgen_config4_600.json")
    batch = wire.encode_docs([g["docs"][0]["logs"]])
    with Engine(0) as e:
        e.set_launch_shape(0, 2048)
        res = e.apply_materialize(batch)
    assert (res.logs["status"] == abi.ERR_CAPACITY).all()


def test_config5_patch_stream_golden(eng):
    """Patch[] stream of the full 8 192-op config-5 log against the reference-made fixture (oracle/gen_patch_golden.js --from)."""
    g = _load("ptxgen_config5_8192.json")
    p = _load("patches_config5_8192.json")
    assert p["impl"] == "ref" and p["from"] == "ptxgen_config5_8192.json"
    batch = wire.encode_docs([d["logs"] for d in g["docs"]])
    res, pat = _streams(eng, batch)
    H.check_patch_streams(batch, pat, [d["expected"] for d in p["docs"]])


def _canonical_patches(pat):
    """Records with the comment ids of one insert patch (kind INSERT_COMMENT, emitted through an atomic counter) in id order."""
    off = pat.patch_off.astype(np.int64)
    valid = np.concatenate([np.arange(off[l], off[l] + n) for l, n in enumerate(pat.logs["n_patches"].astype(np.int64))] or [np.zeros(0, np.int64)])
    rec = pat.patches[valid]  # (the gaps between the logs' capacities are not written)
    is_c = rec["kind"] == abi.PATCH_INSERT_COMMENT
    start = np.arange(len(rec))
    start[is_c] = 0
    start = np.maximum.accumulate(start)  # a comment record sorts with the run that follows its insert record
    order = np.lexsort((np.where(is_c, rec["a"], 0), is_c, start))
    return rec[order]


def test_replay_with_global_winner_arrays_equals_the_lds_only_form():
    """Logs whose replay working set exceeds 5.5 KB replay with the per-slot link urls, the op tables and the comment ops' id tables in global memory:
    same records as the all-LDS form (PTX_FLAG_REPLAY_LDS_ONLY) on 2 048-op config-4 logs."""
    from peritext_amd import workloads
    from peritext_amd.engine import Engine

    c = workloads.gen_config("config4", ops=2048)
    out = []
    for flags in (0, abi.FLAG_REPLAY_LDS_ONLY):
        with Engine(0, flags=flags) as e:
            db, _ = e.generate(c["replicas"], c["ops_per_log"], c["mix"], c["mark_types"], 24, 99, list_cap=2048)
            dr = e.alloc_result(db)
            e.merge(db, dr)
            e.sync()
            pat = e.replay_patches(db, dr)
            assert (pat.logs["status"] == 0).all() and int(pat.logs["n_patches"].sum()) > 24 * 3 * 2048
            out.append(pat)
            e.free_result(dr)
            e.free_batch(db)
    a, b = out
    assert np.array_equal(a.logs, b.logs)
    assert np.array_equal(_canonical_patches(a), _canonical_patches(b))


def test_set_stream_and_count_converged(eng):
    """ptx_set_stream + ptx_count_converged: the engine runs on the caller's HIP stream (here a torch stream) and counts the
    converged documents on the device, no host synchronisation in between."""
    import torch

    g = _load("ptxgen_config4_600.json")
    docs = [d["logs"] for d in g["docs"]]
    docs[1] = [docs[1][0], docs[1][1], docs[1][2][:-3]]  # one document whose third replica lags: not converged
    batch = wire.encode_docs(docs)
    replicas = 3
    db = eng.upload(batch, copies=5)
    dr = eng.alloc_result(db)
    stream = torch.cuda.Stream()
    count = torch.full((1,), -1, dtype=torch.int64, device="cuda")
    try:
        eng.set_stream(stream.cuda_stream)
        with torch.cuda.stream(stream):
            eng.merge(db, dr)
            eng.count_converged(dr, replicas, count.data_ptr())
            doubled = count * 2  # a torch kernel on the same stream sees the count without any host sync
        stream.synchronize()
        logs = eng.download_logs(dr, eng.n_logs(db))
        dg = logs["digest"].reshape(-1, replicas, 2)
        want = int(((dg == dg[:, :1, :]).all(axis=(1, 2)) & (logs["status"].reshape(-1, replicas) == 0).all(axis=1)).sum())
        assert want == 5 * (len(docs) - 1)
        assert int(count.item()) == want and int(doubled.item()) == 2 * want
    finally:
        eng.set_stream(0)
        eng.free_result(dr)
        eng.free_batch(db)
    # back on its own stream the engine still works
    H.check_generated(_load("ptxgen_mini.json"), eng.apply_materialize)


def test_digest_allgather_in_the_c_abi_single_rank(eng):
    """ptx_comm_* / ptx_allgather_digests / ptx_count_converged_digests (SURVEY §8b,e): RCCL bound inside the library, here a
    communicator of ONE rank (the box has one GPU; N ranks run the same code in bench.py --gpus N): the gathered array is this
    rank's digests, the device count equals the host's."""
    import torch

    g = _load("ptxgen_config4_600.json")
    docs = [d["logs"] for d in g["docs"]]
    docs[2] = [docs[2][0], docs[2][1][:-2], docs[2][2]]  # one document that has not converged
    batch = wire.encode_docs(docs)
    replicas = 3
    db = eng.upload(batch, copies=7)
    dr = eng.alloc_result(db)
    n_logs = eng.n_logs(db)
    comm = eng.comm_init(eng.comm_unique_id(), 0, 1)
    gathered = torch.zeros((n_logs, 2), dtype=torch.int64, device="cuda")
    count = torch.full((1,), -1, dtype=torch.int64, device="cuda")
    try:
        eng.merge(db, dr)
        eng.allgather_digests(comm, dr, [n_logs], gathered.data_ptr())
        eng.count_converged_digests(gathered.data_ptr(), n_logs, replicas, count.data_ptr())
        eng.sync()
        logs = eng.download_logs(dr, n_logs)
        assert (gathered.cpu().numpy().view(np.uint64) == logs["digest"]).all()
        dg = logs["digest"].reshape(-1, replicas, 2)
        want = int((dg == dg[:, :1, :]).all(axis=(1, 2)).sum())
        assert want == 7 * (len(docs) - 1) and int(count.item()) == want
    finally:
        eng.comm_destroy(comm)
        eng.free_result(dr)
        eng.free_batch(db)


def test_padded_gather_path_with_real_rccl_single_rank():
    """PTX_FLAG_PAD_GATHER: pack -> all-gather of max(counts) pairs -> ptx_compact_digests_kernel, here with the REAL RCCL and a communicator of one rank
    (the multi-process runs of the same path are in test_gpu_shard_ranks.py, over the test-suite's RCCL stand-in)."""
    import torch
    from peritext_amd.engine import Engine

    g = _load("ptxgen_config4_600.json")
    batch = wire.encode_docs([d["logs"] for d in g["docs"]])
    with Engine(0, flags=abi.FLAG_PAD_GATHER) as e:
        db = e.upload(batch, copies=3)
        dr = e.alloc_result(db)
        n_logs = e.n_logs(db)
        comm = e.comm_init(e.comm_unique_id(), 0, 1)
        gathered = torch.zeros((n_logs, 2), dtype=torch.int64, device="cuda")
        count = torch.full((1,), -1, dtype=torch.int64, device="cuda")
        try:
            e.merge(db, dr)
            for _ in range(2):  # the second call reuses the cached block table
                e.allgather_digests(comm, dr, [n_logs], gathered.data_ptr())
            e.count_converged_digests(gathered.data_ptr(), n_logs, 3, count.data_ptr())
            e.sync()
            logs = e.download_logs(dr, n_logs)
            assert (gathered.cpu().numpy().view(np.uint64) == logs["digest"]).all() and int(count.item()) == n_logs // 3
            with pytest.raises(ValueError):
                e.allgather_digests(comm, dr, [n_logs, 3], gathered.data_ptr())  # one count per rank of the communicator
        finally:
            e.comm_destroy(comm)
            e.free_result(dr)
            e.free_batch(db)


def test_malformed_rows_are_named(eng):
    """GPU twin: the row pass only flags malformed rows while it streams (and keeps every store inside the log's LDS window when a
    header understates the rows); the first failing row is named by a second pass."""
    H.check_malformed_rows(eng.apply_materialize)


def test_root_maps_on_the_device(eng):
    """getRoot() (micromerge.ts:443-449; last writer wins per key, :572-602) of replicas whose changes write the root map and nested
    maps concurrently, against the reference's own answers (rootmap_ref.json: made by the type-erased reference)."""
    g = _load("rootmap_ref.json")
    assert g["impl"] == "ref" and g["docs"] == H.root_map_docs()

    def root_fn(batch):
        db = eng.upload(batch)
        try:
            return eng.root_map(db)
        finally:
            eng.free_batch(db)

    batch, rm = H.check_root_maps(root_fn, eng.apply_materialize, g["expected"])
    # a text-only batch: every replica's root holds just the text list
    mini = _load("ptxgen_mini.json")
    b2 = wire.encode_docs([d["logs"] for d in mini["docs"]])
    rm2 = root_fn(b2)
    assert (rm2.logs["status"] == 0).all() and all(wire.decode_root(b2, rm2, l) == {"text": {"$list": True}} for l in range(b2.n_logs))
