"""On-device change(): the generator logic of peritext_amd/csrc/gen_core.h compiled with -DPTX_EMU (tests/emu,
test tooling only) against the oracle's PTXGEN (oracle/ptxgen.js — Micromerge.change of reference/src/micromerge.ts:308-441
driven by the workload of reference/test/fuzz.ts): every Change of every replica log deep-equal (actor, seq, deps, startOp,
ops with their element ids, boundary positions and attrs), on every committed PTXGEN fixture and on fresh seeds.  The GPU
tests (test_gpu_parity.py) repeat it through ptx_generate on a real MI355X."""
import json
import os

import numpy as np
import pytest

import helpers as H
from peritext_amd import abi, wire

pytestmark = pytest.mark.skipif(not os.path.exists(H.EMU_LIB), reason="tests/emu/libperitext_emu.so not built (run __graft_entry__.build())")

norm = lambda x: json.loads(json.dumps(x, sort_keys=True))  # noqa: E731


def check_generated_logs(batch, docs_logs):
    log = 0
    for logs in docs_logs:
        for want in logs:
            got = wire.decode_changes(batch, log)
            assert len(got) == len(want), "log %d: %d changes, expected %d" % (log, len(got), len(want))
            for i, (x, y) in enumerate(zip(got, want)):
                assert norm(x) == norm(y), "log %d change %d: %r != %r" % (log, i, x, y)
            log += 1
    assert log == batch.n_logs


@pytest.mark.parametrize("name,cfg", [("ptxgen_mini.json", "mini"), ("ptxgen_config2.json", "config2"), ("ptxgen_config3_512.json", "config3"),
                                       ("ptxgen_config4_600.json", "config4"), ("ptxgen_rich_700.json", "rich")])
def test_generator_reproduces_the_committed_ptxgen_fixtures(name, cfg):
    """The fixtures hold the Change logs the oracle's change() produced (seed, config and ops in the file): the generator,
    given only (config, seed, document index), must produce the same logs."""
    with open(os.path.join(H.GOLDEN, name)) as f:
        g = json.load(f)
    c = H.gen_config(cfg, ops=g["cfg"]["opsPerLog"], replicas=g["cfg"]["replicas"])
    first = g["docs"][0]["docIndex"]
    batch, status = H.emu_generate(c, len(g["docs"]), g["seed"], first_doc=first)
    assert not status.any()
    check_generated_logs(batch, [d["logs"] for d in g["docs"]])


def test_decode_changes_inverts_encode_docs():
    with open(os.path.join(H.GOLDEN, "ptxgen_rich_700.json")) as f:
        g = json.load(f)
    docs = [d["logs"] for d in g["docs"]]
    check_generated_logs(wire.encode_docs(docs), docs)


@pytest.mark.parametrize("cfg,docs,ops,seed", [("mini", 24, None, 77), ("rich", 2, None, 78), ("config4", 1, None, 79), ("config5", 1, 2000, 80)])
def test_generator_against_live_oracle_and_merge(cfg, docs, ops, seed):
    """Fresh seeds (config4: full 4 096-op logs): same logs as the oracle; and the generated batch, merged, gives the spans the
    oracle's replicas hold — the whole device pipeline generate -> merge against the reference's change + getTextWithFormatting."""
    if not H.have_node():
        pytest.skip("node not installed")
    g = H.oracle_gen(cfg, seed=seed, docs=docs, ops=ops)
    batch, status = H.emu_generate(H.gen_config(cfg, ops=ops), docs, seed)
    assert not status.any()
    check_generated_logs(batch, [d["logs"] for d in g["docs"]])
    res = H.emu_merge(batch, lds_bytes=160 * 1024, admission=True)
    log = 0
    for d in g["docs"]:
        for exp in d["expected"]:
            assert int(res.logs[log]["status"]) == 0
            assert H.norm_spans(wire.decode_spans(batch, res, log)) == H.norm_spans(exp["spans"]), "log %d" % log
            log += 1


def test_generator_capacity_and_single_replica():
    c = H.gen_config("mini")
    batch, status = H.emu_generate(c, 2, 5, list_cap=8)  # lists outgrow 8 elements
    assert (status == abi.ERR_CAPACITY).all()
    c1 = H.gen_config("config2", ops=64)
    batch, status = H.emu_generate(c1, 3, 9)
    assert not status.any() and batch.n_logs == 3 and int(batch.log_off[-1]) == 3 * 65
    if H.have_node():
        g = H.oracle_gen("config2", seed=9, docs=3, ops=64)
        check_generated_logs(batch, [d["logs"] for d in g["docs"]])


@pytest.mark.parametrize("cfg,replicas,ops,docs,seed", [("mini", 2, None, 8, 91), ("mini", 4, 160, 6, 92), ("rich", 4, 400, 2, 93), ("config3", 2, 300, 3, 94),
                                                         ("mini", 5, 200, 4, 95), ("rich", 8, 500, 2, 96), ("mini", 7, 150, 3, 97), ("config4", 6, 900, 1, 98)])
def test_generator_other_replica_counts(cfg, replicas, ops, docs, seed):
    """2 to 8 replicas (the sync pairs, the pending-queue order and the deps rows change with R; round 5: five to eight replicas — three actor bits per key, four
    words of dependencies per change, documents that ptx_merge_kernel_many / _many_wide admit)."""
    if not H.have_node():
        pytest.skip("node not installed")
    g = H.oracle_gen(cfg, seed=seed, docs=docs, ops=ops, replicas=replicas)
    batch, status = H.emu_generate(H.gen_config(cfg, ops=ops, replicas=replicas), docs, seed)
    assert not status.any()
    check_generated_logs(batch, [d["logs"] for d in g["docs"]])
    res = H.emu_merge(batch, lds_bytes=160 * 1024, admission=True)
    assert not res.logs["status"].any()
    d = res.logs["digest"].reshape(docs, replicas, 2)
    assert (d == d[:, :1, :]).all()  # every replica of a document converges


def test_generator_random_workloads():
    """Seeded random workload definitions (mix, mark types and their order, replicas, log length, initial text): the device
    logic and the oracle's change() must agree on every one."""
    if not H.have_node():
        pytest.skip("node not installed")
    rng = np.random.default_rng(20240923)
    all_marks = ["strong", "em", "comment", "link"]
    for trial in range(10):
        cuts = np.sort(rng.integers(0, 101, size=3))
        mix = [int(cuts[0]), int(cuts[1] - cuts[0]), int(cuts[2] - cuts[1]), int(100 - cuts[2])]
        marks = [all_marks[i] for i in rng.permutation(4)[: int(rng.integers(0, 5))]]
        replicas = int(rng.integers(1, 5))
        ops = int(rng.integers(20, 260))
        text = "".join(chr(int(c)) for c in rng.integers(97, 123, size=int(rng.integers(1, 9))))
        seed = int(rng.integers(1, 1 << 30))
        g = H.oracle_gen("mini", seed=seed, docs=3, ops=ops, replicas=replicas, mix=tuple(mix), marks=tuple(marks), initial_text=text)
        cfg = {"replicas": replicas, "ops_per_log": ops, "mix": mix, "mark_types": [abi.MARK_NAMES.index(m) for m in marks], "initial_text": text}
        batch, status = H.emu_generate(cfg, 3, seed)
        assert not status.any(), (trial, cfg)
        check_generated_logs(batch, [d["logs"] for d in g["docs"]])


@pytest.mark.parametrize("cfg,ops", [("mini", None), ("config5", 1500)])
def test_generator_list_capacity_is_exact(cfg, ops):
    """list_cap = the largest element count any replica reaches is enough; one less is reported, never silently wrong.  (config5:
    ONE replica per document, comments — the comment ranks at the end live in the dead lists, which are small then.)"""
    c = H.gen_config(cfg, ops=ops)
    batch, status = H.emu_generate(c, 6, 21)
    assert not status.any()
    hdr = wire.census(batch.log_off, batch.op_id, batch.action, batch.mark_type)
    per_doc = hdr["n_ins"].reshape(6, c["replicas"]).max(axis=1)
    need = int(per_doc.max())
    ok, st_ok = H.emu_generate(c, 6, 21, list_cap=need)
    assert not st_ok.any() and np.array_equal(ok.op_id, batch.op_id)
    _, st_small = H.emu_generate(c, 6, 21, list_cap=need - 1)
    assert [int(s) for s in st_small] == [abi.ERR_CAPACITY if int(n) == need else 0 for n in per_doc]


def test_replay_of_a_batch_without_headers_and_of_a_generated_batch():
    """The patch replay derives the log headers itself when the batch has none, and runs on generator output (fixed string
    tables) as on encoder output."""
    if not H.have_node():
        pytest.skip("node not installed")
    g = H.oracle_gen("mini", seed=61, docs=5)
    dl = [d["logs"] for d in g["docs"]]
    expected = H.oracle_apply(dl, patches=True)
    enc = wire.encode_docs(dl)
    enc.log_hdr = None
    gen, status = H.emu_generate(H.gen_config("mini"), 5, 61)
    assert not status.any()
    for batch in (enc, gen):
        res = H.emu_merge(batch)
        pat = H.emu_replay(batch, res)
        H.check_patch_streams(batch, pat, expected)
