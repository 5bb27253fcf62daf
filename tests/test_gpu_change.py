"""change() for caller-supplied InputOperations through the C ABI on a real MI355X (ptx_change / ptx_batch_append_device;
SURVEY §8 a13): every Micromerge.change(InputOperation[]) call of the reference's own test file yields the reference's Change
(tests/golden/kat_change_scripts.json, made by the type-erased reference), the replicas end on the reference's spans."""
import numpy as np
import pytest

import change_script as CS
import helpers as H
from peritext_amd import abi, wire

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    import torch

    assert torch.cuda.is_available(), "GPU tests need a real MI355X"
    from peritext_amd.engine import Engine

    e = Engine(0)
    yield e
    e.close()


class GpuBackend:
    def __init__(self, eng):
        self.eng = eng

    def change(self, batch, ops):
        e = self.eng
        db = e.upload(batch)
        dr = e.alloc_result(db)
        made_h = None
        try:
            e.merge(db, dr)
            e.sync()
            made_h, status = e.change(db, dr, ops)
            made = e.download_batch(made_h, batch.values, batch.urls, batch.log_doc, batch.doc_actors, batch.doc_comments, batch.keys, batch.map_values)
        finally:
            if made_h is not None:
                e.free_batch(made_h)
            e.free_result(dr)
            e.free_batch(db)
        return made, status

    def spans(self, batch):
        return self.eng.apply_materialize(batch)


def test_reference_test_file_change_calls(eng):
    assert CS.run_scripts(CS.load_scripts(), GpuBackend(eng)) == 125


def test_change_calls_on_map_objects(eng):
    """Micromerge.change with InputOperations on map objects (micromerge.ts:400-425) through ptx_change on the device: the Changes the
    reference itself returned (rootmap_ref.json); what was made, appended to the replicas, gives the reference's getRoot()."""
    import json
    import os

    with open(os.path.join(H.GOLDEN, "rootmap_ref.json")) as f:
        g = json.load(f)
    batch, made = H.check_map_change_calls(GpuBackend(eng).change, g["change"])


def test_statuses_and_device_append(eng):
    """RangeError 'List index out of bounds' (micromerge.ts:804) and misuse per log; what was made is appended device to device
    (ptx_batch_append_device) and the grown batch merges to the expected documents."""
    base = H.mini_doc([])
    docs = [[base] for _ in range(8)]
    batch = wire.encode_docs(docs)
    T = ["text"]
    calls = [
        [[{"path": T, "action": "insert", "index": 6, "values": ["x"]}]],
        [[{"path": T, "action": "delete", "index": 3, "count": 3}]],
        [[{"path": T, "action": "addMark", "markType": "link", "attrs": {"url": "u"}, "startIndex": 0, "endIndex": 0}]],
        [[{"path": T, "action": "addMark", "markType": "strong", "startIndex": 5, "endIndex": 7}]],
        [[{"path": [], "action": "makeList", "key": "text"}]],
        [[{"path": T, "action": "insert", "index": 5, "values": ["!", "?"]}, {"path": T, "action": "delete", "index": 0, "count": 1}]],
        [],
        [[{"path": T, "action": "addMark", "markType": "strong", "startIndex": 0, "endIndex": 9}]],
    ]
    ops = wire.encode_input_ops(batch, calls, ["a"] * 8)
    db = eng.upload(batch)
    dr = eng.alloc_result(db)
    eng.merge(db, dr)
    eng.sync()
    made_h, status = eng.change(db, dr, ops)
    grown_h = eng.append_device(db, made_h)
    dr2 = eng.alloc_result(grown_h)
    try:
        assert [int(s) for s in status] == [abi.ERR_INDEX_OOB] * 4 + [abi.ERR_BAD_OP, 0, 0, 0]
        assert eng.n_ops(made_h) == 4 and eng.n_changes(made_h) == 2
        eng.merge(grown_h, dr2)
        eng.sync()
        grown = eng.download_batch(grown_h, batch.values, batch.urls, batch.log_doc, batch.doc_actors, batch.doc_comments)
        res = eng.download(grown_h, dr2)
    finally:
        for h in (dr, dr2):
            eng.free_result(h)
        for h in (db, made_h, grown_h):
            eng.free_batch(h)
    assert (res.logs["status"] == 0).all()
    assert wire.decode_spans(grown, res, 5) == [{"text": "BCDE!?", "marks": {}}]
    assert wire.decode_spans(grown, res, 7) == [{"text": "ABCDE", "marks": {"strong": {"active": True}}}]
    assert wire.decode_spans(grown, res, 0) == [{"text": "ABCDE", "marks": {}}]
    ch = wire.decode_changes(grown, 7)[-1]
    assert ch["ops"][0]["end"] == {"type": "endOfText"} and ch["seq"] == 3 and ch["deps"] == {"a": 2} and ch["startOp"] == 7


def test_change_on_generated_replicas_and_empty_base(eng):
    """InputOperations on top of device-generated PTXGEN replicas (tombstones with defined after-slots, comments, links): the made
    batch appended on the device merges with converging digests when every replica receives every change; and a document is
    started from NOTHING (makeList + insert on an empty log)."""
    cfg = H.gen_config("mini")
    h, info = eng.generate(cfg["replicas"], cfg["ops_per_log"], cfg["mix"], cfg["mark_types"], 16, 77)
    actors_t, comments_t, log_doc_t = wire.generated_tables(16, cfg["replicas"], info["n_comments"])
    batch = eng.download_batch(h, wire.GEN_VALUES, wire.GEN_URLS, log_doc_t, actors_t, comments_t)
    dr = eng.alloc_result(h)
    eng.merge(h, dr)
    eng.sync()
    res = eng.download(h, dr)
    T = ["text"]
    calls, actors = [], []
    for log in range(batch.n_logs):
        n = int(res.logs[log]["n_visible"])
        r = log % cfg["replicas"]
        c = [{"path": T, "action": "insert", "index": n // 2, "values": ["x", "y"]}, {"path": T, "action": "delete", "index": n // 3, "count": 1},
             {"path": T, "action": "addMark", "markType": "link", "attrs": {"url": "B.com"}, "startIndex": 0, "endIndex": max(1, n // 2)}]
        calls.append([c] if r == 0 else [])  # replica doc1 of every document edits
        actors.append("doc%d" % (r + 1))
    ops = wire.encode_input_ops(batch, calls, actors)
    made_h, status = eng.change(h, dr, ops)
    assert (status == 0).all()
    made = eng.download_batch(made_h, batch.values, batch.urls, batch.log_doc, batch.doc_actors, batch.doc_comments)
    # deliver doc1's change to the other replicas of its document: `more` = the same change on every log of the document
    deliver = []
    for d in range(16):
        ch = wire.decode_changes(made, d * cfg["replicas"], text_obj="1@doc1")
        assert len(ch) == 1 and ch[0]["actor"] == "doc1"
        deliver.append([ch] * cfg["replicas"])
    more = wire.encode_docs(deliver, extra_actors=batch.doc_actors, extra_comments=batch.doc_comments, text_objs=["1@doc1"] * 16)
    assert (more.action != abi.ACT_NOP).all()
    # same tables as the generated batch: value / url ids of a generated batch are fixed (wire.GEN_*), the encoder interns its own
    vmap = np.array([wire.GEN_VALUES.index(v) for v in more.values], dtype=np.uint32)
    ins = more.action == abi.ACT_INSERT
    more.payload[ins] = vmap[more.payload[ins]]
    lnk = (more.action == abi.ACT_ADDMARK) & (more.mark_type == abi.MARK_LINK)
    more.payload[lnk] = np.array([wire.GEN_URLS.index(more.urls[int(p)]) for p in more.payload[lnk]], dtype=np.uint32)
    grown_h = eng.append(h, more)
    dr2 = eng.alloc_result(grown_h)
    try:
        eng.merge(grown_h, dr2)
        eng.sync()
        logs = eng.download_logs(dr2, batch.n_logs)
        assert (logs["status"] == 0).all()
        dg = logs["digest"].reshape(16, cfg["replicas"], 2)
        assert (dg == dg[:, :1, :]).all(), "every replica of a document received doc1's change: they converge again"
        assert (logs["n_visible"] == res.logs["n_visible"] + 1).all()
    finally:
        eng.free_result(dr)
        eng.free_result(dr2)
        for x in (h, made_h, grown_h):
            eng.free_batch(x)
    # a document started from nothing
    empty = wire.encode_docs([[[]]], extra_actors=[["zed"]])
    ops0 = wire.encode_input_ops(empty, [[[{"path": [], "action": "makeList", "key": "text"}, {"path": T, "action": "insert", "index": 0, "values": list("hey")}]]], ["zed"])
    made0, st0 = GpuBackend(eng).change(empty, ops0)
    assert int(st0[0]) == 0
    ch0 = wire.decode_changes(made0, 0)[0]
    assert ch0["seq"] == 1 and ch0["startOp"] == 1 and [o["action"] for o in ch0["ops"]] == ["makeList", "set", "set", "set"]
    b1 = wire.encode_docs([[[ch0]]])
    r1 = eng.apply_materialize(b1)
    assert wire.decode_spans(b1, r1, 0) == [{"text": "hey", "marks": {}}]
