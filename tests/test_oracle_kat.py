"""The oracle against the reference's golden vectors: runs on CPU (needs node)."""
import json
import os

import pytest

import helpers as H

pytestmark = pytest.mark.skipif(not H.have_node(), reason="node (the oracle's runtime) is not installed")


def test_oracle_reproduces_reference_expected_results():
    """Every testConcurrentWrites case of reference/test/micromerge.ts: both replicas' logs, applied to a
    fresh oracle replica with applyChange, flatten to the reference's expectedResult literal."""
    cases = H.load_kat()
    docs = [[r["log"] for r in c["replicas"]] for c in cases]
    got = H.oracle_apply(docs)
    n_lit = 0
    for c, exp in zip(cases, got):
        for r, e in zip(c["replicas"], exp):
            want = c.get("expected", r["spans"])
            if want is None:
                continue
            assert H.norm_spans(e["spans"]) == H.norm_spans(want), c["title"]
            n_lit += "expected" in c
    assert len(cases) == 46 and n_lit >= 60  # 31 two-replica cases carry an expectedResult literal


def test_reference_test_file_passes_against_oracle_when_reference_is_mounted():
    """Replay the reference's own mocha file (types erased in memory) against the oracle: 46/46."""
    if not os.path.exists("/root/reference/test/micromerge.ts"):
        pytest.skip("/root/reference not mounted (GPU box)")
    out = H.run_node(["oracle/run_reference_tests.js"])
    assert "46 passed, 0 failed" in out


def test_oracle_matches_erased_reference_differentially():
    """oracle vs the reference itself (oracle/_ref) on seeded PTXGEN traces: changes, patches, spans."""
    if not os.path.exists(os.path.join(H.ROOT, "oracle", "_ref", "micromerge.js")):
        pytest.skip("oracle/_ref not built (needs /root/reference)")
    out = H.run_node(["oracle/diff_fuzz.js", "--docs", "60", "--seed", "123"])
    assert " 0 mismatches" in out


def test_reference_traces_converge_and_match_survey():
    """The 9 saved traces (committed as op logs only): forward and reversed delivery converge; config #1
    (links-minimal) gives the result recorded in SURVEY.md Appendix B."""
    with open(os.path.join(H.GOLDEN, "reference_traces.json")) as f:
        traces = json.load(f)
    docs = [t["logs"] for t in traces]
    got = H.oracle_apply(docs)
    for t, exp in zip(traces, got):
        spans = [H.norm_spans(e["spans"]) for e in exp]
        assert all(s == spans[0] for s in spans), t["name"]
        assert H.norm_spans(t["spans"]) == spans[0], t["name"]
    lm = [t for t in traces if t["name"] == "links-minimal.json"][0]
    assert lm["spans"] == [{"text": "ABC9ee09150DE", "marks": {"link": {"url": "https://inkandswitch.com/pushpin"}}}]
