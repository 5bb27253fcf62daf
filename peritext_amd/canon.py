"""Canonical output form and its 128-bit digest — the host-side restatement of what the kernel emits.

`canonical_from_spans` turns a `FormatSpanWithText[]` (what the reference's getTextWithFormatting
returns, reference/src/peritext.ts:337-395) plus the element values into the engine's canonical triple
(values, spans, comment intervals); `digest` hashes such a triple exactly like
peritext_amd/csrc/merge_core.h (`ptx_digest_item`).  Tests use both to check the HIP path's raw
arrays and digests against the oracle, not just the decoded JSON.
"""
import numpy as np

from . import abi

_U32 = np.uint32


def _rotl(x, r):
    return (x << _U32(r)) | (x >> _U32(32 - r))


def _qr(a, b, c, d):
    """The ChaCha quarter round over four uint32 arrays (merge_core.h PTX_DIGEST_QR)."""
    a = a + b
    d = _rotl(d ^ a, 16)
    c = c + d
    b = _rotl(b ^ c, 12)
    a = a + b
    d = _rotl(d ^ a, 8)
    c = c + d
    b = _rotl(b ^ c, 7)
    return a, b, c, d


def _items(tag, a, b, c):
    """Sum over the items (tag, a[i], b[i], c[i]) of their 128-bit hashes, as two 64-bit halves (merge_core.h ptx_digest_item: four quarter rounds, the
    words' roles rotated from round to round; 32-bit arithmetic that wraps)."""
    a = np.asarray(a, dtype=np.uint64).astype(_U32)
    b = np.asarray(b, dtype=np.uint64).astype(_U32)
    c = np.asarray(c, dtype=np.uint64).astype(_U32)
    with np.errstate(over="ignore"):
        x0 = a ^ _U32(0x9E3779B9)
        x1 = b ^ _U32(0x85EBCA6B)
        x2 = c ^ _U32(0xC2B2AE35)
        x3 = np.full(a.shape, (int(tag) | (int(tag) << 16)) ^ 0x165667B1, dtype=_U32)
        x0, x1, x2, x3 = _qr(x0, x1, x2, x3)
        x1, x2, x3, x0 = _qr(x1, x2, x3, x0)
        x2, x3, x0, x1 = _qr(x2, x3, x0, x1)
        x3, x0, x1, x2 = _qr(x3, x0, x1, x2)
        h1 = (x0.astype(np.uint64) | (x1.astype(np.uint64) << np.uint64(32))).sum(dtype=np.uint64)
        h2 = (x2.astype(np.uint64) | (x3.astype(np.uint64) << np.uint64(32))).sum(dtype=np.uint64)
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
