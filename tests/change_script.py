"""Runner of the reference's change()/applyChange call scripts (tests/golden/kat_change_scripts.json) over any backend that
offers change(batch, input_ops) -> (made batch, status) and spans(batch) -> results: the CPU emulation
(tests/test_emu_change.py) and the C ABI on the GPU (tests/test_gpu_change.py)."""
import json
import os

import helpers as H
from peritext_amd import wire


def load_scripts():
    with open(os.path.join(H.GOLDEN, "kat_change_scripts.json")) as f:
        g = json.load(f)
    assert g["impl"] == "ref"
    return g["cases"]


def text_obj_of(log):
    for ch in log:
        for op in ch["ops"]:
            if op["action"] == "makeList":
                return op["opId"]
    return None


def norm_change(ch):
    """A Change in comparable form: the reference's ROOT / HEAD Symbols vanish in JSON, the decoder writes "_root" / "_head"."""
    out = {"actor": ch["actor"], "seq": ch["seq"], "deps": {k: v for k, v in ch["deps"].items() if v}, "ops": [],
           # a Change without ops uses no opId: its startOp says nothing (applyChange: maxOp = max(maxOp, startOp - 1), micromerge.ts:511)
           # and the wire format does not carry it
           "startOp": ch["startOp"] if ch["ops"] else None}
    for op in ch["ops"]:
        o = {k: v for k, v in op.items() if not (k == "obj" and v == wire.ROOT) and not (k == "elemId" and v == wire.HEAD)}
        out["ops"].append(json.loads(json.dumps(o, sort_keys=True)))
    return out


def comment_ids(case):
    ids = set()
    for e in case["events"]:
        for op in e["change"]["ops"]:
            if op.get("markType") == "comment":
                ids.add(op["attrs"]["id"])
    return sorted(ids)


def run_scripts(cases, backend):
    """All cases in lockstep: step k = the k-th call of every case; the change() calls of a step are ONE backend.change call.
    Returns the number of change() calls checked."""
    logs = [[[] for _ in c["actors"]] for c in cases]  # case -> replica -> Change[] applied so far
    checked = 0
    for k in range(max(len(c["events"]) for c in cases)):
        active = [i for i, c in enumerate(cases) if k < len(c["events"])]
        changers = [i for i in active if cases[i]["events"][k]["kind"] == "change"]
        if changers:
            docs = [logs[i] for i in changers]
            batch = wire.encode_docs(docs, extra_actors=[cases[i]["actors"] for i in changers], extra_comments=[comment_ids(cases[i]) for i in changers])
            per_log, actors = [], []
            for i in changers:
                e = cases[i]["events"][k]
                for r, a in enumerate(cases[i]["actors"]):
                    per_log.append([e["ops"]] if r == e["replica"] else [])
                    actors.append(a)
            ops = wire.encode_input_ops(batch, per_log, actors)
            made, status = backend.change(batch, ops)
            assert (status == 0).all(), [cases[i]["title"] for i in changers]
            log = 0
            for i in changers:
                e = cases[i]["events"][k]
                for r in range(len(cases[i]["actors"])):
                    if r == e["replica"]:
                        got = wire.decode_changes(made, log, text_obj=text_obj_of(logs[i][r]))
                        assert len(got) == 1, cases[i]["title"]
                        assert norm_change(got[0]) == norm_change(e["change"]), "%s: call %d" % (cases[i]["title"], k)
                        logs[i][r].append(got[0])  # the replica goes on with what the DEVICE made
                        checked += 1
                    else:
                        assert made.log_off[log + 1] == made.log_off[log]
                    log += 1
        for i in active:
            e = cases[i]["events"][k]
            if e["kind"] == "apply":
                logs[i][e["replica"]].append(e["change"])
    batch = wire.encode_docs(logs)
    res = backend.spans(batch)
    log = 0
    for i, c in enumerate(cases):
        for r in range(len(c["actors"])):
            if c["spans"][r] is not None:
                assert H.norm_spans(wire.decode_spans(batch, res, log)) == H.norm_spans(c["spans"][r]), c["title"]
            log += 1
    return checked
