"""Canonical output form and its 128-bit digest — the host-side restatement of what the kernel emits.

`canonical_from_spans` turns a `FormatSpanWithText[]` (what the reference's getTextWithFormatting
returns, reference/src/peritext.ts:337-395) plus the element values into the engine's canonical triple
(values, spans, comment intervals); `digest` hashes such a triple exactly like
peritext_amd/csrc/merge_core.h (`ptx_digest_item`).  Tests use both to check the HIP path's raw
arrays and digests against the oracle, not just the decoded JSON.
"""
import numpy as np

from . import abi

_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)
_GOLD = np.uint64(0x9E3779B97F4A7C15)
_SALT2 = np.uint64(0xD6E8FEB86659FD93)


def _fmix64(x):
    x = x.astype(np.uint64)
    x = x ^ (x >> np.uint64(30))
    x = x * _M1
    x = x ^ (x >> np.uint64(27))
    x = x * _M2
    x = x ^ (x >> np.uint64(31))
    return x


def _items(tag, a, b, c):
    a = np.asarray(a, dtype=np.uint64)
    b = np.asarray(b, dtype=np.uint64)
    c = np.asarray(c, dtype=np.uint64)
    with np.errstate(over="ignore"):
        x = (np.uint64(tag) << np.uint64(60)) ^ (a << np.uint64(32)) ^ b
        y = _fmix64(x) ^ (c * _GOLD)
        h1 = _fmix64(y).sum(dtype=np.uint64)
        h2 = _fmix64(y ^ _SALT2).sum(dtype=np.uint64)
    return int(h1), int(h2)


def digest(values, spans, cintervals, n_elems):
    """values: u32[V]; spans: sequence of (start, attr); cintervals: sequence of (id, start, end)."""
    mask = (1 << 64) - 1
    h1 = h2 = 0
    V = len(values)
    S = len(spans)
    I = len(cintervals)

    def add(t):
        nonlocal h1, h2
        h1 = (h1 + t[0]) & mask
        h2 = (h2 + t[1]) & mask

    if V:
        add(_items(1, np.arange(V), np.asarray(values), np.zeros(V)))
    if S:
        sp = np.asarray([(int(s[0]), int(s[1])) for s in spans], dtype=np.uint64).reshape(-1, 2)
        add(_items(2, np.arange(S), sp[:, 0], sp[:, 1]))
    if I:
        ci = np.asarray([(int(c[0]), int(c[1]), int(c[2])) for c in cintervals], dtype=np.uint64).reshape(-1, 3)
        add(_items(3, ci[:, 0], ci[:, 1], ci[:, 2]))
    add(_items(4, [0], [V], [S]))
    add(_items(4, [1], [I], [n_elems]))
    return h1, h2


def canonical_from_spans(spans, text, value_ix, url_ix, comment_rank):
    """Canonical triple of a reference-style result.

    spans: FormatSpanWithText[]; text: the visible element values in order (doc.root.text);
    value_ix / url_ix: string -> id tables of the batch; comment_rank: comment id string -> doc-local rank.
    Span boundaries are recovered at element granularity by matching accumulated text lengths.
    """
    values = [value_ix[v] for v in text]
    rows = []
    per_span_ids = []
    pos = 0
    for sp in spans:
        start = pos
        need = len(sp["text"])
        got = 0
        while got < need:  # (empty-string values would make this ambiguous; the generators never emit them)
            got += len(text[pos])
            pos += 1
        if got != need:
            raise ValueError("span text does not align with element values")
        m = sp["marks"]
        attr = 0
        if "strong" in m:
            attr |= abi.ATTR_STRONG
        if "em" in m:
            attr |= abi.ATTR_EM
        if "link" in m:
            attr |= abi.ATTR_LINK | url_ix[m["link"]["url"]]
        ids = None
        if "comment" in m:
            attr |= abi.ATTR_COMMENT
            ids = [comment_rank[c["id"]] for c in m["comment"]]
        rows.append((start, attr))
        per_span_ids.append(set(ids) if ids is not None else set())
    if pos != len(text):
        raise ValueError("spans do not cover the text")
    # comment presence intervals: maximal runs of consecutive spans containing the id
    cints = []
    all_ids = sorted(set().union(*per_span_ids)) if per_span_ids else []
    for c in all_ids:
        run = None
        for k, ids in enumerate(per_span_ids):
            here = c in ids
            if here and run is None:
                run = rows[k][0]
            if not here and run is not None:
                cints.append((c, run, rows[k][0]))
                run = None
        if run is not None:
            cints.append((c, run, len(text)))
    return values, rows, cints
