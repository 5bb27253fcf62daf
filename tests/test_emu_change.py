"""change() for caller-supplied InputOperations (peritext_amd/csrc/change_core.h) on the CPU emulation: every
Micromerge.change(InputOperation[]) call of the reference's own test file (tests/golden/kat_change_scripts.json, made by the
type-erased reference: oracle/run_reference_tests.js --impl ref --dump-scripts) must yield, Change for Change, what the
reference returned, and the replicas must end on the reference's spans.  tests/test_gpu_change.py repeats it through the C ABI."""
import os

import numpy as np
import pytest

import change_script as CS
import helpers as H
from peritext_amd import abi, wire

pytestmark = pytest.mark.skipif(not os.path.exists(H.EMU_LIB), reason="tests/emu/libperitext_emu.so not built (run __graft_entry__.build())")


class EmuBackend:
    def __init__(self, reverse=0):
        self.reverse = reverse

    def change(self, batch, ops):
        res = H.emu_merge(batch, reverse=self.reverse, admission=True)
        made, status = H.emu_change(batch, res, ops, reverse=self.reverse)
        return made, status

    def spans(self, batch):
        res = H.emu_merge(batch, reverse=self.reverse, admission=True)
        return res


@pytest.mark.parametrize("reverse", [0, 1, 2])
def test_reference_test_file_change_calls(reverse):
    """All 46 cases of reference/test/micromerge.ts in lockstep: 125 change() calls (makeList, insert, delete, addMark, removeMark
    incl. comments, links, tombstone boundaries) -> the reference's Changes; final spans = the reference's."""
    cases = CS.load_scripts()
    n_calls = CS.run_scripts(cases, EmuBackend(reverse))
    assert n_calls == 125


def test_index_out_of_bounds_and_misuse_statuses():
    """The reference's RangeError 'List index out of bounds' (micromerge.ts:804) and type misuse, per log, without touching the
    neighbours; a failed log makes no change."""
    base = H.mini_doc([])  # "ABCDE", actor a
    docs = [[base] for _ in range(8)]
    batch = wire.encode_docs(docs)
    T = ["text"]
    calls = [
        [[{"path": T, "action": "insert", "index": 6, "values": ["x"]}]],                                # index-1 = 5 is past the end
        [[{"path": T, "action": "delete", "index": 3, "count": 3}]],                                      # third delete finds nothing
        [[{"path": T, "action": "addMark", "markType": "link", "attrs": {"url": "u"}, "startIndex": 0, "endIndex": 0}]],  # after(elem[-1])
        [[{"path": T, "action": "addMark", "markType": "strong", "startIndex": 5, "endIndex": 7}]],    # start out of bounds
        [[{"path": [], "action": "makeList", "key": "text"}]],                                           # a second text list
        [[{"path": T, "action": "insert", "index": 5, "values": ["!", "?"]}, {"path": T, "action": "delete", "index": 0, "count": 1}]],
        [],
        [[{"path": T, "action": "addMark", "markType": "strong", "startIndex": 0, "endIndex": 9}]],    # inclusive end past the text: endOfText
    ]
    ops = wire.encode_input_ops(batch, calls, ["a"] * 8)
    res = H.emu_merge(batch, admission=True)
    made, status = H.emu_change(batch, res, ops)
    assert [int(s) for s in status] == [abi.ERR_INDEX_OOB] * 4 + [abi.ERR_BAD_OP, 0, 0, 0]
    assert [int(made.log_off[l + 1] - made.log_off[l]) for l in range(8)] == [0, 0, 0, 0, 0, 3, 0, 1]
    grown = H.concat_batches(batch, made)
    r2 = H.emu_merge(grown, admission=True)
    assert (r2.logs["status"] == 0).all()
    assert wire.decode_spans(grown, r2, 5) == [{"text": "BCDE!?", "marks": {}}]
    assert wire.decode_spans(grown, r2, 7) == [{"text": "ABCDE", "marks": {"strong": {"active": True}}}]
    ch = wire.decode_changes(made, 7, text_obj="1@a")[0]
    assert ch["ops"][0]["end"] == {"type": "endOfText"} and ch["seq"] == 3 and ch["deps"] == {"a": 2} and ch["startOp"] == 7
    if H.have_node():  # the oracle agrees on the two that succeed
        exp = H.oracle_apply([[wire.decode_changes(grown, 5)], [wire.decode_changes(grown, 7)]])
        assert H.norm_spans(exp[0][0]["spans"]) == H.norm_spans(wire.decode_spans(grown, r2, 5))
        assert H.norm_spans(exp[1][0]["spans"]) == H.norm_spans(wire.decode_spans(grown, r2, 7))


@pytest.mark.skipif(not H.have_node(), reason="node (oracle runtime) not installed")
def test_change_on_generated_replicas_matches_the_oracle():
    """InputOperations on top of PTXGEN replica states (tombstones with defined after-slots, comments, links): the same calls on
    the oracle's replicas give the same Changes."""
    gen = H._load_golden("ptxgen_mini.json")
    docs = [d["logs"] for d in gen["docs"][:6]]
    exp0 = H.oracle_apply(docs)
    rng = np.random.default_rng(7)
    calls, actors = [], []
    T = ["text"]
    for d, logs in enumerate(docs):
        for r in range(len(logs)):
            n = len(exp0[d][r]["text"])
            c = []
            for _ in range(3):
                k = int(rng.integers(0, 4))
                if k == 0 or n < 2:
                    c.append({"path": T, "action": "insert", "index": int(rng.integers(0, n + 1)), "values": ["Q", "R"]})
                    n += 2
                elif k == 1:
                    i = int(rng.integers(0, n - 1))
                    c.append({"path": T, "action": "delete", "index": i, "count": 1})
                    n -= 1
                else:
                    s = int(rng.integers(0, n))
                    e = int(rng.integers(s + 1, n + 1))
                    mt = ["strong", "em", "link", "comment"][int(rng.integers(0, 4))]
                    op = {"path": T, "action": "addMark" if k == 2 else "removeMark", "markType": mt, "startIndex": s, "endIndex": e}
                    if mt == "link" and k == 2:
                        op["attrs"] = {"url": "Z.com"}
                    if mt == "comment":
                        op["attrs"] = {"id": "comment-new"}
                    c.append(op)
            calls.append([c])
            actors.append("doc%d" % (r + 1))
    batch = wire.encode_docs(docs, extra_comments=[["comment-new"]] * len(docs))
    ops = wire.encode_input_ops(batch, calls, actors)
    res = H.emu_merge(batch, admission=True)
    made, status = H.emu_change(batch, res, ops)
    assert (status == 0).all()
    want = H.oracle_change(docs, calls, actors)
    log = 0
    for d, logs in enumerate(docs):
        for r in range(len(logs)):
            got = wire.decode_changes(made, log, text_obj=CS.text_obj_of(logs[r]))
            assert CS.norm_change(got[0]) == CS.norm_change(want[log]), (d, r)
            log += 1


@pytest.mark.skipif(not H.have_node(), reason="node (oracle runtime) not installed")
def test_change_on_a_replica_past_65535_changes():
    """seq = clock + 1 and deps = the clock are plain numbers (micromerge.ts:314-327): a replica that has made 65 600 changes (most of them empty:
    change() with no ops still bumps seq) makes change 65 601 with the exact values — the made batch then carries the wide column.
    Expected: the type-erased reference's own change()."""
    log = [H.mini_doc([])[0]] + [{"actor": "a", "seq": k, "deps": {}, "startOp": 7, "ops": []} for k in range(2, 65601)]
    batch = wire.encode_docs([[log]])
    assert batch.chg_env_hi is not None
    calls = [[[{"path": ["text"], "action": "insert", "index": 2, "values": ["x", "y"]}], [{"path": ["text"], "action": "addMark", "markType": "strong", "startIndex": 1, "endIndex": 4}]]]
    want = H.oracle_change([[log]], calls, ["a"], impl="ref")
    ops = wire.encode_input_ops(batch, calls, ["a"])
    res = H.emu_merge(batch, admission=False)  # (admission of such a log is the HBM-staged kernel's: tests/test_emu_biglog.py)
    made, status = H.emu_change(batch, res, ops)
    assert int(status[0]) == 0 and made.chg_env_hi is not None
    got = wire.decode_changes(made, 0, text_obj="1@a")
    assert [c["seq"] for c in got] == [65601, 65602] and got == want
    # ... and the replica after the two changes is admitted by the HBM-staged kernel
    after = H.concat_batches(batch, made)
    r = H.emu_merge_big(after, admission=True)
    assert int(r.logs["status"][0]) == 0 and int(r.logs["n_visible"][0]) == 7
