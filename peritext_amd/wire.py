"""Host-side wire codec: Peritext `Change` JSON  <->  the SoA op-log batch of include/peritext_hip.h.

Input side mirrors the reference's types (reference/src/micromerge.ts:60-71 `Change`, :204-212
`Operation`; src/peritext.ts:17-65 boundary positions and mark ops): a *replica log* is the list of
`Change` objects one replica applied, in application order — exactly what a loop of
`doc.applyChange(change)` (micromerge.ts:499) would be fed.  Output side rebuilds what
`doc.getTextWithFormatting(["text"])` (micromerge.ts:516) returns: `FormatSpanWithText[]`.

Encoding rules the wrapper owns:
  * actor strings -> rank in UTF-16 code-unit order within the doc, so that integer order of
    (counter << 32 | rank) equals compareOpIds (micromerge.ts:812-827, JS string `<`);
  * ROOT / HEAD -> 0 (JSON drops the reference's Symbols: a `makeList` without `obj`, an inserting
    `set` without `elemId`; the strings "_root"/"_head" are accepted too);
  * inserted values (arbitrary strings, e.g. " is great!" at test/micromerge.ts:202) -> ids in one
    batch-wide string table; link urls -> ids in one batch-wide table;
  * comment ids -> DOC-LOCAL dense ranks in code-unit order (peritext.ts:318 keeps arrays id-sorted).
"""
import json
import re
from dataclasses import dataclass, field

import numpy as np

from . import abi

_ID_RE = re.compile(r"^([0-9]+)@(.*)$", re.S)
ROOT = "_root"
HEAD = "_head"


def _u16key(s):
    """Sort key reproducing JS string comparison (UTF-16 code units)."""
    return s.encode("utf-16-be", "surrogatepass")


def split_op_id(op_id):
    m = _ID_RE.match(op_id)
    if not m:
        raise ValueError("Invalid operation ID: %r" % (op_id,))
    return int(m.group(1)), m.group(2)


@dataclass
class Batch:
    """A batch of replica logs in wire form plus the tables needed to decode the results."""

    log_off: np.ndarray
    op_id: np.ndarray
    ref_a: np.ndarray
    ref_b: np.ndarray
    payload: np.ndarray
    action: np.ndarray
    mark_type: np.ndarray
    side_a: np.ndarray
    side_b: np.ndarray
    # causal envelope (one entry per Change): chg_hdr = actor << 20 | nops, chg_env rows of env_stride(max_actors) u16 = seq, deps[...]
    chg_off: np.ndarray
    chg_hdr: np.ndarray
    chg_env: np.ndarray
    max_actors: int
    # per-log census (include/peritext_hip.h ptx_log_hdr); None = let the library compute it
    log_hdr: np.ndarray = None
    # decode tables
    values: list = field(default_factory=list)  # value id -> string
    urls: list = field(default_factory=list)  # url id -> string
    log_doc: list = field(default_factory=list)  # log index -> doc index
    doc_actors: list = field(default_factory=list)  # doc -> [actor strings in rank order]
    doc_comments: list = field(default_factory=list)  # doc -> [comment id strings in rank order]
    keys: list = field(default_factory=list)  # key id -> string (keys of the map objects: PTX_ACT_MAPSET / MAPDEL / MAKELIST rows, ref_b)
    map_values: list = field(default_factory=list)  # value id -> JSON text of the value a PTX_ACT_MAPSET row sets
    # the wide envelope column (ptx_batch.chg_env_hi): high halves of chg_env's values, None while every seq / dep fits 16 bits
    chg_env_hi: np.ndarray = None
    # a document with several list objects (encode_docs(list_keys=...)): device log l merges the list under root key log_list[l] of replica log_replica[l]
    log_list: list = None
    log_replica: list = None

    @property
    def n_logs(self):
        return len(self.log_off) - 1

    # the envelope's fields, unpacked (views for decoding and tests; the device reads chg_hdr / chg_env)
    @property
    def chg_actor(self):
        return self.chg_hdr >> np.uint32(abi.CHG_ACTOR_SHIFT)

    @property
    def chg_nops(self):
        return self.chg_hdr & np.uint32(abi.CHG_NOPS)

    @property
    def chg_seq(self):
        return self._env32().reshape(-1, abi.env_stride(self.max_actors))[:, 0]

    @property
    def chg_deps(self):
        """[n_changes, max_actors]"""
        return self._env32().reshape(-1, abi.env_stride(self.max_actors))[:, 1:1 + self.max_actors]

    def _env32(self):
        """The envelope's values as u32: exact with the wide column, else the 16-bit column (saturated at 65 535)."""
        v = self.chg_env.astype(np.uint32)
        return v if self.chg_env_hi is None else v | (self.chg_env_hi.astype(np.uint32) << np.uint32(16))

    @property
    def n_ops(self):
        return int(self.log_off[-1])

    def counted_ops(self, log=None):
        """Ops the metric counts: the ops of the text list (insert / delete / addMark / removeMark)."""
        a = self.action if log is None else self.action[int(self.log_off[log]) : int(self.log_off[log + 1])]
        return int(np.count_nonzero((a >= abi.ACT_INSERT) & (a <= abi.ACT_REMOVEMARK)))

    def tile(self, copies):
        """`copies` back-to-back copies of this batch (same content, distinct rows)."""
        n = self.n_ops
        offs = [self.log_off[:-1] + k * n for k in range(copies)]
        log_off = np.concatenate(offs + [np.array([copies * n], dtype=np.uint64)]).astype(np.uint64)
        rep = lambda a: np.tile(a, copies)  # noqa: E731
        if self.chg_off is None:  # a batch without the Change envelope (e.g. downloaded from a wrapped device batch)
            chg_off = chg_hdr = chg_env = None
            chg_env_hi = None
        else:
            nc = int(self.chg_off[-1])
            chg_off = np.concatenate([self.chg_off[:-1] + k * nc for k in range(copies)] + [np.array([copies * nc], dtype=np.uint64)]).astype(np.uint64)
            chg_hdr, chg_env = rep(self.chg_hdr), rep(self.chg_env)
            chg_env_hi = None if self.chg_env_hi is None else rep(self.chg_env_hi)
        hdr = None if self.log_hdr is None else np.tile(self.log_hdr, copies)
        return Batch(
            log_off, rep(self.op_id), rep(self.ref_a), rep(self.ref_b), rep(self.payload), rep(self.action),
            rep(self.mark_type), rep(self.side_a), rep(self.side_b), chg_off, chg_hdr, chg_env, self.max_actors, hdr, self.values, self.urls,
            self.log_doc * copies, self.doc_actors, self.doc_comments, keys=self.keys, map_values=self.map_values, chg_env_hi=chg_env_hi,
        )


def census(log_off, op_id, action, mark_type, payload=None):
    """ptx_log_hdr rows (abi.LOG_HDR_DTYPE) of a batch: what the encoder knows for free about every log."""
    n_logs = len(log_off) - 1
    hdr = np.zeros(n_logs, dtype=abi.LOG_HDR_DTYPE)
    if n_logs == 0 or len(op_id) == 0:
        return hdr
    lens = np.diff(log_off.astype(np.int64))
    lix = np.repeat(np.arange(n_logs), lens)
    hdr["n_ins"] = np.bincount(lix[action == abi.ACT_INSERT], minlength=n_logs)
    hdr["n_del"] = np.bincount(lix[action == abi.ACT_DELETE], minlength=n_logs)
    is_mark = (action == abi.ACT_ADDMARK) | (action == abi.ACT_REMOVEMARK)
    for t in range(4):
        hdr["n_mark"][:, t] = np.bincount(lix[is_mark & (mark_type == t)], minlength=n_logs)
    mc = np.zeros(n_logs, dtype=np.uint32)
    ma = np.zeros(n_logs, dtype=np.uint32)
    np.maximum.at(mc, lix, (op_id >> np.uint64(32)).astype(np.uint32))
    np.maximum.at(ma, lix, (op_id & np.uint64(0xFFFFFFFF)).astype(np.uint32))
    hdr["max_counter"] = mc
    hdr["max_actor"] = ma
    if payload is not None:
        # comment ids are ranks over the whole document: a log's id space = its largest comment payload + 1
        is_c = is_mark & (mark_type == abi.MARK_COMMENT)
        nid = np.zeros(n_logs, dtype=np.uint32)
        np.maximum.at(nid, lix[is_c], payload[is_c].astype(np.uint32) + np.uint32(1))
        hdr["n_comment_ids"] = nid
    return hdr


def pack_envelope(actor, seq, nops, deps, max_actors):
    """(chg_hdr u32[n], chg_env u16[n * env_stride]) from per-change actor ranks, seqs, op counts and a [n, max_actors] deps
    matrix.  seq / deps beyond 16 bits saturate at 65 535: a log holds at most 65 533 changes, so such a change can never be
    admitted — the reference's RangeError either way."""
    actor, nops = np.asarray(actor, dtype=np.uint64), np.asarray(nops, dtype=np.uint64)
    if len(actor) and (int(actor.max()) > 4095 or int(nops.max()) > abi.CHG_NOPS):
        raise ValueError("a document has at most 4096 actors and a change at most %d ops" % abi.CHG_NOPS)
    hdr = ((actor << np.uint64(abi.CHG_ACTOR_SHIFT)) | nops).astype(np.uint32)
    es = abi.env_stride(max_actors)
    env = np.zeros((len(actor), es), dtype=np.uint16)
    if len(actor):
        env[:, 0] = np.minimum(np.asarray(seq, dtype=np.uint64), abi.ENV_SATURATED).astype(np.uint16)
        env[:, 1:1 + max_actors] = np.minimum(np.asarray(deps, dtype=np.uint64).reshape(len(actor), max_actors), abi.ENV_SATURATED).astype(np.uint16)
    return hdr, env.reshape(-1)


def pack_envelope_wide(actor, seq, nops, deps, max_actors):
    """(chg_hdr, chg_env, chg_env_hi): as pack_envelope while every seq / dep is at most 65 534 (chg_env_hi = None); otherwise the
    values are split EXACTLY into low and high halves (include/peritext_hip.h ptx_batch.chg_env_hi) — seq / deps are plain numbers in
    the reference (micromerge.ts:499-511), a replica that made one change per keystroke passes 65 535."""
    seq = np.asarray(seq, dtype=np.uint64)
    deps = np.asarray(deps, dtype=np.uint64).reshape(len(seq), max_actors)
    top = max(int(seq.max()) if len(seq) else 0, int(deps.max()) if deps.size else 0)
    if top < abi.ENV_SATURATED:
        hdr, env = pack_envelope(actor, seq, nops, deps, max_actors)
        return hdr, env, None
    if top > 0xFFFFFFFF:
        raise ValueError("seq / deps beyond 32 bits")
    hdr, _ = pack_envelope(actor, np.zeros(len(seq), np.uint64), nops, np.zeros_like(deps), max_actors)
    es = abi.env_stride(max_actors)
    v = np.zeros((len(seq), es), dtype=np.uint32)
    v[:, 0] = seq
    v[:, 1:1 + max_actors] = deps
    return hdr, (v & np.uint32(0xFFFF)).astype(np.uint16).reshape(-1), (v >> np.uint32(16)).astype(np.uint16).reshape(-1)


def _pack(ctr, rank):
    return (int(ctr) << 32) | int(rank)


def resolve_list_path(log, path):
    """The list object a replica's root shows under `path` once it has applied `log`: every key of a map holds the write with the LARGEST opId
    (reference/src/micromerge.ts:572-602: a write is kept iff compareOpIds says its id is larger than the key's current one), so two replicas that made
    a list under one key concurrently both show the same one.  Returns its opId, or None when the path does not end at a list (a deleted key, a scalar,
    a map)."""
    win = {}
    for ch in log:
        for op in ch["ops"]:
            if "key" in op and "elemId" not in op and op["action"] in ("set", "del", "makeMap", "makeList"):
                obj = op.get("obj")
                k = (ROOT if obj is None else obj, op["key"])
                ctr, actor = split_op_id(op["opId"])
                cur = win.get(k)
                if cur is None or (ctr, _u16key(actor)) > cur[0]:
                    win[k] = ((ctr, _u16key(actor)), op)
    cur = ROOT
    for i, key in enumerate(path):
        w = win.get((cur, key))
        if w is None or w[1]["action"] != ("makeList" if i == len(path) - 1 else "makeMap"):
            return None
        cur = w[1]["opId"]
    return cur


def encode_docs(docs, extra_actors=None, extra_comments=None, text_objs=None, list_keys=("text",)):
    """docs: list of docs; a doc is a list of replica logs; a replica log is a list of Change dicts.

    list_keys (round 5): the LIST objects of the root map to merge, by key.  The engine merges one list object per device log; a document that holds several
    (micromerge.ts:589: makeList under any key; :534-571: applyOp takes any list object) becomes one device log per (replica, key), in that order — each with the
    replica's whole Change envelope (causal admission is the replica's, not the list's) and, of the list ops, those of ITS list: the ops on the replica's other
    list objects are rows without effect there.  `Batch.log_list[l]` names the key of device log l, `Batch.log_replica[l]` its replica within the document.
    The default merges the list under "text" alone, as the reference's editor does (bridge.ts).  A key with dots — "meta.notes" — names a list NESTED in map
    objects by its path (the reference's OperationPath ["meta", "notes"], micromerge.ts:178-196): the list the first makeList of key "notes" made in the map
    the first makeMap of key "meta" made in the root map.

    All replicas of a doc share actor ranks and comment-id ranks, so their digests are comparable.
    extra_actors / extra_comments: per doc, actor names / comment ids that get a rank although no change of the batch uses them
    yet (replicas about to make their first change, comment ids a later InputOperation will introduce: ranks are positions in
    the document's sorted id list, so they must be reserved before the rows that use them are made).
    text_objs: per doc, the opId of the text list when the logs of this batch do not hold its makeList (a batch of newly arrived
    changes for Engine.append / ptx_batch_append); None = found in the log.
    """
    values, value_ix = [], {}
    urls, url_ix = [], {}
    keys, key_ix = ["text"], {"text": 0}  # key 0 of every batch: the text list's (rows made on the device use it)
    mvals, mval_ix = [], {}

    def intern(table, index, v):
        if v not in index:
            index[v] = len(table)
            table.append(v)
        return index[v]

    cols = {k: [] for k in ("op_id", "ref_a", "ref_b", "payload", "action", "mark_type", "side_a", "side_b")}
    log_off = [0]
    chg_off = [0]
    chg_actor, chg_seq, chg_nops, chg_deps_rows = [], [], [], []
    log_doc, doc_actors, doc_comments = [], [], []
    log_list, log_replica = [], []
    max_actors = 1
    for d, logs in enumerate(docs):
        actors, comments = set(), set()
        for log in logs:
            for ch in log:
                actors.add(ch["actor"])
                for a in (ch.get("deps") or {}):
                    actors.add(a)
                for op in ch["ops"]:
                    actors.add(split_op_id(op["opId"])[1])
                    # ids that are only ever REFERENCED (possibly unknown elements) need a rank too
                    refs = [op.get("elemId"), (op.get("start") or {}).get("elemId"), (op.get("end") or {}).get("elemId")]
                    for ref in refs:
                        if isinstance(ref, str) and ref not in (HEAD, ROOT):
                            actors.add(split_op_id(ref)[1])
                    if op.get("markType") == "comment":
                        comments.add(op["attrs"]["id"])
                    if isinstance(op.get("obj"), str) and op["obj"] not in (HEAD, ROOT):
                        actors.add(split_op_id(op["obj"])[1])
        actors.update((extra_actors or [[]] * len(docs))[d])
        comments.update((extra_comments or [[]] * len(docs))[d])
        actor_list = sorted(actors, key=_u16key)
        arank = {a: i for i, a in enumerate(actor_list)}
        comment_list = sorted(comments, key=_u16key)
        crank = {c: i for i, c in enumerate(comment_list)}
        doc_actors.append(actor_list)
        doc_comments.append(comment_list)
        max_actors = max(max_actors, len(actor_list))

        def enc_id(s):
            if s is None or s == HEAD or s == ROOT:
                return 0
            ctr, actor = split_op_id(s)
            return _pack(ctr, arank[actor])

        for rep_ix, lkey in ((r_, k_) for r_ in range(len(logs)) for k_ in list_keys):
            log = logs[rep_ix]
            text_obj = text_objs[d] if text_objs else None
            other_lists = set()  # the replica's list objects that are not this device log's: their ops are rows without effect here
            want_path = tuple(lkey.split("."))  # the list's path through the map objects (one key: a list of the root map)
            path_of = {}  # map / list object -> the keys that lead to it from the root map, as the ops of this log made them
            # Round 6: WHICH list the path names is decided as the reference decides it — the last-writer-wins winner of every key on the way
            # (two replicas that made a list under one key concurrently show the same one).  None (the path ends at no list: a deleted key ...) or a
            # seeded text_obj (Changes appended to a resident log): the first object made under the path, as before.
            resolved = resolve_list_path(log, want_path) if text_obj is None else None
            nrows = 0
            for ch in log:
                chg_actor.append(arank[ch["actor"]])
                chg_seq.append(int(ch["seq"]))
                chg_nops.append(len(ch["ops"]))
                chg_deps_rows.append({arank[a]: int(v) for a, v in (ch.get("deps") or {}).items()})
                for op in ch["ops"]:
                    act = op["action"]
                    row = dict(op_id=enc_id(op["opId"]), ref_a=0, ref_b=0, payload=0, action=abi.ACT_NOP, mark_type=0, side_a=0, side_b=0)
                    obj = op.get("obj")
                    on_root = obj is None or obj == ROOT
                    if act in ("makeMap", "makeList") and "key" in op and (on_root or obj in path_of):
                        path_of.setdefault(op["opId"], (() if on_root else path_of[obj]) + (op["key"],))
                    if act == "makeList" and on_root and op.get("key") == lkey and text_obj is None and (resolved is None or op["opId"] == resolved):
                        row.update(action=abi.ACT_MAKELIST, ref_b=intern(keys, key_ix, lkey))  # also a write of the root map's key
                        text_obj = op["opId"]
                    elif text_obj is not None and obj == text_obj:
                        if act == "set" and op.get("insert"):
                            v = op["value"]
                            if not isinstance(v, str):
                                raise ValueError("Expected value inserted into text to be a string")
                            if v not in value_ix:
                                value_ix[v] = len(values)
                                values.append(v)
                            row.update(action=abi.ACT_INSERT, ref_a=enc_id(op.get("elemId")), payload=value_ix[v])
                        elif act == "del" and "elemId" in op:
                            row.update(action=abi.ACT_DELETE, ref_a=enc_id(op["elemId"]))
                        elif act in ("addMark", "removeMark"):
                            mt = abi.MARK_NAMES.index(op["markType"])
                            st, en = op["start"], op["end"]
                            row.update(
                                action=abi.ACT_ADDMARK if act == "addMark" else abi.ACT_REMOVEMARK,
                                mark_type=mt,
                                side_a=abi.SIDE_NAMES.index(st["type"]),
                                side_b=abi.SIDE_NAMES.index(en["type"]),
                                ref_a=enc_id(st.get("elemId")),
                                ref_b=enc_id(en.get("elemId")),
                            )
                            if mt == abi.MARK_LINK and act == "addMark":
                                u = op["attrs"]["url"]
                                if u not in url_ix:
                                    url_ix[u] = len(urls)
                                    urls.append(u)
                                row["payload"] = url_ix[u]
                            elif mt == abi.MARK_COMMENT:
                                row["payload"] = crank[op["attrs"]["id"]]
                    elif "key" in op and "elemId" not in op and act in ("set", "del", "makeMap", "makeList"):
                        # an op on a MAP object (the root map or a nested one), micromerge.ts:572-602: last writer wins per (object, key)
                        row.update(ref_a=enc_id(obj), ref_b=intern(keys, key_ix, op["key"]))
                        if act == "del":
                            row["action"] = abi.ACT_MAPDEL
                        else:
                            row["action"] = abi.ACT_MAPSET
                            row["mark_type"] = abi.MAPV_MAP if act == "makeMap" else abi.MAPV_LIST if act == "makeList" else abi.MAPV_SCALAR
                            if act == "set":
                                row["payload"] = intern(mvals, mval_ix, json.dumps(op.get("value"), sort_keys=True, ensure_ascii=False, separators=(",", ":")))
                            if act == "makeList":
                                if text_obj is None and len(want_path) > 1 and path_of.get(op["opId"]) == want_path and (resolved is None or op["opId"] == resolved):
                                    text_obj = op["opId"]  # the nested list this device log merges: its makeList stays a write of its map's key
                                else:
                                    other_lists.add(op["opId"])
                    elif obj in other_lists and (act in ("addMark", "removeMark") or "elemId" in op or op.get("insert")):
                        pass  # an op on ANOTHER list object of this replica (merged by its own device log when its key is in list_keys): PTX_ACT_NOP here
                    elif act in ("addMark", "removeMark") or "elemId" in op or op.get("insert"):
                        # A list op on an object that NO makeList of this log created: the reference throws RangeError("Object does not exist")
                        # (micromerge.ts:538) — rejected here.  (An op on a list object the log did create under a key that is not in list_keys is a row
                        # without effect, the branch above: the reference's checks on THAT list — an unknown element, micromerge.ts:752 — are made only
                        # when the caller names the key in list_keys and so has the list merged; INTEGRATION.md "Several list objects per document".)
                        raise ValueError("list op %s on object %r, which no earlier makeList of this log created" % (op.get("opId"), obj))
                    for k, v in row.items():
                        cols[k].append(v)
                    nrows += 1
            log_off.append(log_off[-1] + nrows)
            chg_off.append(len(chg_actor))
            log_doc.append(d)
            log_list.append(lkey)
            log_replica.append(rep_ix)
    deps = np.zeros((len(chg_actor), max_actors), dtype=np.uint32)
    for i, row in enumerate(chg_deps_rows):
        for a, v in row.items():
            deps[i, a] = v
    u64 = lambda x: np.asarray(x, dtype=np.uint64)  # noqa: E731
    chg_hdr, chg_env, chg_env_hi = pack_envelope_wide(chg_actor, chg_seq, chg_nops, deps, max_actors)
    hdr = census(u64(log_off), u64(cols["op_id"]), np.asarray(cols["action"], dtype=np.uint8), np.asarray(cols["mark_type"], dtype=np.uint8),
                 np.asarray(cols["payload"], dtype=np.uint32))
    return Batch(
        log_hdr=hdr,
        log_off=u64(log_off), op_id=u64(cols["op_id"]), ref_a=u64(cols["ref_a"]), ref_b=u64(cols["ref_b"]),
        payload=np.asarray(cols["payload"], dtype=np.uint32), action=np.asarray(cols["action"], dtype=np.uint8),
        mark_type=np.asarray(cols["mark_type"], dtype=np.uint8), side_a=np.asarray(cols["side_a"], dtype=np.uint8),
        side_b=np.asarray(cols["side_b"], dtype=np.uint8), chg_off=u64(chg_off),
        chg_hdr=chg_hdr, chg_env=chg_env, max_actors=max_actors,
        values=values, urls=urls, log_doc=log_doc, doc_actors=doc_actors, doc_comments=doc_comments, keys=keys, map_values=mvals, chg_env_hi=chg_env_hi,
        log_list=log_list if tuple(list_keys) != ("text",) else None, log_replica=log_replica if tuple(list_keys) != ("text",) else None,
    )


@dataclass
class RootMaps:
    """Host view of ptx_root_map: per log the winning row of every (map object, key) its ops write."""

    entry_off: np.ndarray  # u64 [n_logs + 1]
    logs: np.ndarray  # ROOT_LOG_DTYPE [n_logs]
    entries: np.ndarray  # ROOT_ENTRY_DTYPE


def decode_root(batch, rm, log):
    """getRoot() of the replica behind `log` (micromerge.ts:443-449) as JSON: nested dicts for the maps, {"$list": True} where a
    list object hangs (the text list: its content is getTextWithFormatting's business), the set values elsewhere.  Maps that no
    key points at any more (their makeMap lost, or was overwritten) are not reachable, as in the reference."""
    b0 = int(batch.log_off[log])
    e0 = int(rm.entry_off[log])
    ent = rm.entries[e0:e0 + int(rm.logs["n_entries"][log])]
    by_obj = {}
    for e in ent:
        by_obj.setdefault(int(e["obj"]), []).append(e)
    keys = batch.keys or ["text"]  # batches made on the device (ptx_generate) hold the text list's makeList only: key id 0

    def build(obj):
        out = {}
        for e in by_obj.get(obj, []):
            kind = int(e["kind"])
            if kind == abi.MAPV_DELETED:
                continue
            k = keys[int(e["key"])]
            if kind == abi.MAPV_MAP:
                out[k] = build(int(batch.op_id[b0 + int(e["row"])]))
            elif kind == abi.MAPV_LIST:
                out[k] = {"$list": True}
            else:
                out[k] = json.loads(batch.map_values[int(e["value"])])
        return out

    return build(0)


@dataclass
class InputOps:
    """Index-based InputOperations (reference/src/micromerge.ts:133-148) as the columns of ptx_input_ops: per log the Changes
    to make, per Change its input ops."""

    chg_off: np.ndarray
    op_off: np.ndarray
    action: np.ndarray
    mark_type: np.ndarray
    index: np.ndarray
    count: np.ndarray
    payload: np.ndarray
    values: np.ndarray
    actor: np.ndarray
    max_actors: int


def _is_map_input(op):
    """An InputOperation on a map object (micromerge.ts:109-131: makeMap / set / del, makeList of another key than the text's)."""
    a = op.get("action")
    if a in ("makeMap", "set", "del"):
        return True
    return a == "makeList" and not (list(op.get("path", [])) == [] and op.get("key") == "text")


def map_children_of_log(batch, log):
    """metadata[CHILDREN] of every map object of the replica behind `log` (micromerge.ts:585-596): (object, key id) -> (child object
    id packed counter << 12 | actor rank, kind).  A makeMap / makeList registers its child when it wins its key AT THE TIME it is
    applied; later winners of the key that are no makeMap leave the entry alone — so this is a replay of the log's map rows in order."""
    b0, b1 = int(batch.log_off[log]), int(batch.log_off[log + 1])
    last, children = {}, {}
    for i in range(b0, b1):
        a = int(batch.action[i])
        if a not in (abi.ACT_MAKELIST, abi.ACT_MAPSET, abi.ACT_MAPDEL):
            continue
        obj = 0 if a == abi.ACT_MAKELIST else int(batch.ref_a[i])
        obj = ((obj >> 32) << 12) | (obj & 0xFFF)
        key = (obj, int(batch.ref_b[i]) & 0xFFFFFFFF)
        oid = int(batch.op_id[i])
        if key not in last or last[key] < oid:
            last[key] = oid
            kind = abi.MAPV_LIST if a == abi.ACT_MAKELIST else int(batch.mark_type[i]) if a == abi.ACT_MAPSET else abi.MAPV_DELETED
            if kind in (abi.MAPV_MAP, abi.MAPV_LIST):
                children[key] = (((oid >> 32) << 12) | (oid & 0xFFF), kind)
    return children


def encode_input_ops(batch, per_log, actors):
    """per_log[l] = list of change() calls of the replica behind log l, each a list of InputOperation dicts in the reference's
    shape ({path, action: "insert", index, values} / {action: "delete", index, count} / {action: "addMark" | "removeMark",
    startIndex, endIndex, markType, attrs?} / {path: [], action: "makeList", key: "text"} / on map objects {path, action: "makeMap" |
    "makeList" | "del", key} and {path, action: "set", key, value}: their paths are resolved here, micromerge.ts:446-463); actors[l] =
    that replica's actor id.
    New inserted strings / urls extend batch.values / batch.urls; comment ids and actors must already have their rank in
    `batch` (encode_docs(..., extra_actors=, extra_comments=))."""
    value_ix = {v: i for i, v in enumerate(batch.values)}
    url_ix = {u: i for i, u in enumerate(batch.urls)}
    if not batch.keys:
        batch.keys.append("text")
    key_ix = {k: i for i, k in enumerate(batch.keys)}
    mval_ix = {v: i for i, v in enumerate(batch.map_values)}

    def intern(table, ix, v):
        if v not in ix:
            ix[v] = len(table)
            table.append(v)
        return ix[v]

    chg_off, op_off = [0], [0]
    action, mark_type, index, count, payload, values, actor = [], [], [], [], [], [], []
    for l, calls in enumerate(per_log):
        d = batch.log_doc[l]
        actor.append(batch.doc_actors[d].index(actors[l]))
        crank = {c: i for i, c in enumerate(batch.doc_comments[d])}
        children = map_children_of_log(batch, l) if any(_is_map_input(op) for ops in calls for op in ops) else {}
        made = 0  # rows this log's calls have made so far
        for ops in calls:
            for op in ops:
                a = op["action"]
                if _is_map_input(op):
                    # getObjectIdForPath (micromerge.ts:446-463): down the CHILDREN of the map objects, from the root
                    obj = 0
                    for elem in op.get("path", []):
                        child = children.get((obj, key_ix.get(elem, -1)))
                        if child is None:
                            raise ValueError("Child not found: %s in %s" % (elem, op.get("path")))
                        if child[1] != abi.MAPV_MAP:
                            raise ValueError("Object %s in path %r is a list" % (elem, op.get("path")))
                        obj = child[0]
                    k = intern(batch.keys, key_ix, op["key"])
                    if a == "del":
                        row = (abi.IN_MAPDEL, 0, obj, k, 0)
                    else:
                        kind = abi.MAPV_MAP if a == "makeMap" else abi.MAPV_LIST if a == "makeList" else abi.MAPV_SCALAR
                        pl = intern(batch.map_values, mval_ix, json.dumps(op.get("value"), sort_keys=True, ensure_ascii=False, separators=(",", ":"))) if a == "set" else 0
                        row = (abi.IN_MAPSET, kind, obj, k, pl)
                        if kind != abi.MAPV_SCALAR:
                            children[(obj, k)] = (abi.IN_OBJ_NEW | made, kind)  # the newest op of the replica: it wins its key
                    made += 1
                elif a == "makeList":
                    if list(op.get("path", [])) != [] or op.get("key") != "text":
                        raise ValueError("only the text list of the root map is supported")
                    row = (abi.IN_MAKELIST, 0, 0, 0, 0)
                    made += 1
                elif list(op.get("path", [])) != ["text"]:
                    raise ValueError("Only the text list is supported: %r" % (op.get("path"),))
                elif a == "insert":
                    first = len(values)
                    for v in op["values"]:
                        if not isinstance(v, str):
                            raise ValueError("Expected value inserted into text to be a string")
                        if v not in value_ix:
                            value_ix[v] = len(batch.values)
                            batch.values.append(v)
                        values.append(value_ix[v])
                    row = (abi.IN_INSERT, 0, int(op["index"]), len(op["values"]), first)
                    made += len(op["values"])
                elif a == "delete":
                    row = (abi.IN_DELETE, 0, int(op["index"]), int(op["count"]), 0)
                    made += int(op["count"])
                elif a in ("addMark", "removeMark"):
                    mt = abi.MARK_NAMES.index(op["markType"])
                    pl = 0
                    if mt == abi.MARK_LINK and a == "addMark":
                        u = op["attrs"]["url"]
                        if u not in url_ix:
                            url_ix[u] = len(batch.urls)
                            batch.urls.append(u)
                        pl = url_ix[u]
                    elif mt == abi.MARK_COMMENT:
                        pl = crank[op["attrs"]["id"]]
                    row = (abi.IN_ADDMARK if a == "addMark" else abi.IN_REMOVEMARK, mt, int(op["startIndex"]), int(op["endIndex"]), pl)
                    made += 1
                else:
                    raise ValueError("unsupported InputOperation action %r" % (a,))
                for lst, v in zip((action, mark_type, index, count, payload), row):
                    lst.append(v)
            op_off.append(len(action))
        chg_off.append(len(op_off) - 1)
    u32 = lambda x: np.asarray(x, dtype=np.uint32)  # noqa: E731
    return InputOps(np.asarray(chg_off, dtype=np.uint64), np.asarray(op_off, dtype=np.uint64), np.asarray(action, dtype=np.uint8), np.asarray(mark_type, dtype=np.uint8),
                    u32(index), u32(count), u32(payload), u32(values), u32(actor), max(batch.max_actors, max((len(a) for a in batch.doc_actors), default=1)))


@dataclass
class Results:
    """Host view of a merge result: numpy arrays, row r of log l at log_off[l] + r."""

    logs: np.ndarray  # LOG_RESULT_DTYPE [n_logs]
    values: np.ndarray  # u32 [n_rows]
    spans: np.ndarray  # SPAN_DTYPE [n_rows]
    cintervals: np.ndarray  # CINTERVAL_DTYPE [n_rows]
    elem_rank: np.ndarray  # u32 [n_rows]
    ref_slots: np.ndarray = None  # u32 [n_rows], emulation only: the merge's resolved references of the delete / mark rows (the library keeps them on the device)
    ref_slots_hi: np.ndarray = None  # u32 [n_rows], emulation only: the high halves of the mark rows' boundary slots (logs of more than 32 766 list elements; ptx_dresult.refs_hi)
    # ABI 7: the library's rows are COMPACT — log l's values at values[value_off[l] : value_off[l + 1]] etc.; None = the capacity layout (a log's rows at its
    # own row offset batch.log_off[l]: what the kernels write on the device, and what the test-suite's emulation returns)
    value_off: np.ndarray = None
    span_off: np.ndarray = None
    cint_off: np.ndarray = None


def canonical_of_log(batch, res, log):
    """(values u32[], spans [(start, attr)], cintervals [(id, s, e)]) of one log, as plain arrays."""
    r = res.logs[log]
    if res.value_off is None:
        bv = bs = bc = int(batch.log_off[log])
    else:
        bv, bs, bc = int(res.value_off[log]), int(res.span_off[log]), int(res.cint_off[log])
    v = res.values[bv : bv + int(r["n_visible"])]
    s = res.spans[bs : bs + int(r["n_spans"])]
    c = res.cintervals[bc : bc + int(r["n_cintervals"])]
    return v, s, c


def decode_spans(batch, res, log):
    """FormatSpanWithText[] of one log, i.e. what getTextWithFormatting(["text"]) returns
    (reference/src/peritext.ts:35-38, :135-137).  Raises RangeError-like ValueError on a failed log."""
    r = res.logs[log]
    if int(r["status"]) != 0:
        raise ValueError(abi.STATUS_NAMES.get(int(r["status"]), "error %d" % int(r["status"])))
    v, s, c = canonical_of_log(batch, res, log)
    comments = batch.doc_comments[batch.log_doc[log]]
    out = []
    n_vis = len(v)
    for k in range(len(s)):
        start = int(s[k]["start"])
        end = int(s[k + 1]["start"]) if k + 1 < len(s) else n_vis
        attr = int(s[k]["attr"])
        marks = {}
        if attr & abi.ATTR_STRONG:
            marks["strong"] = {"active": True}
        if attr & abi.ATTR_EM:
            marks["em"] = {"active": True}
        if attr & abi.ATTR_COMMENT:
            ids = [int(ci["id"]) for ci in c if int(ci["start"]) <= start < int(ci["end"])]
            marks["comment"] = [{"id": comments[i]} for i in sorted(ids)]
        if attr & abi.ATTR_LINK:
            marks["link"] = {"url": batch.urls[attr & abi.ATTR_ID_MASK]}
        out.append({"text": "".join(batch.values[int(x)] for x in v[start:end]), "marks": marks})
    return out


def split_batch(batch, first_changes):
    """Cut every log after its first `first_changes[l]` changes: (head, tail) Batches sharing the tables of `batch` — what a
    replica had applied at some point, and what arrived since (the input of Engine.append / ptx_batch_append)."""
    parts = ([], [])
    rows = ([0], [0])
    chgs = ([0], [0])
    for l in range(batch.n_logs):
        b0, b1 = int(batch.log_off[l]), int(batch.log_off[l + 1])
        c0, c1 = int(batch.chg_off[l]), int(batch.chg_off[l + 1])
        k = min(max(int(first_changes[l]), 0), c1 - c0)
        cut = b0 + int(batch.chg_nops[c0:c0 + k].sum())
        for part, (r0, r1, q0, q1) in enumerate(((b0, cut, c0, c0 + k), (cut, b1, c0 + k, c1))):
            parts[part].append((r0, r1, q0, q1))
            rows[part].append(rows[part][-1] + r1 - r0)
            chgs[part].append(chgs[part][-1] + q1 - q0)
    out = []
    for part in (0, 1):
        ridx = np.concatenate([np.arange(r0, r1) for r0, r1, _, _ in parts[part]]).astype(np.int64) if parts[part] else np.zeros(0, dtype=np.int64)
        cidx = np.concatenate([np.arange(q0, q1) for _, _, q0, q1 in parts[part]]).astype(np.int64) if parts[part] else np.zeros(0, dtype=np.int64)
        out.append(Batch(
            np.asarray(rows[part], dtype=np.uint64), batch.op_id[ridx], batch.ref_a[ridx], batch.ref_b[ridx], batch.payload[ridx], batch.action[ridx],
            batch.mark_type[ridx], batch.side_a[ridx], batch.side_b[ridx], np.asarray(chgs[part], dtype=np.uint64), batch.chg_hdr[cidx],
            batch.chg_env.reshape(-1, abi.env_stride(batch.max_actors))[cidx].reshape(-1), batch.max_actors, None,
            batch.values, batch.urls, batch.log_doc, batch.doc_actors, batch.doc_comments,
            chg_env_hi=None if batch.chg_env_hi is None else batch.chg_env_hi.reshape(-1, abi.env_stride(batch.max_actors))[cidx].reshape(-1)))
    return out[0], out[1]


def decode_changes(batch, log, text_obj=None):
    """Change[] of one log — the inverse of encode_docs for the ops of the text list (reference/src/micromerge.ts:60-71
    Change, :150-212 Operation, src/peritext.ts:25-65 mark ops), in the JSON-portable form of the traces
    (ROOT / HEAD as "_root" / "_head"), ops on the map objects included.  Needs the Change envelope; PTX_ACT_NOP rows cannot be restored.
    text_obj: opId of the text list when the log does not hold its makeList (a batch of newly made Changes only)."""
    if batch.chg_off is None:
        raise ValueError("the batch carries no Change envelope")
    d = batch.log_doc[log]
    actors, comments = batch.doc_actors[d], batch.doc_comments[d]
    b0 = int(batch.log_off[log])
    c0, c1 = int(batch.chg_off[log]), int(batch.chg_off[log + 1])

    def oid(v):
        v = int(v)
        return "%d@%s" % (v >> 32, actors[v & 0xFFFFFFFF])

    out = []
    row = b0
    c_actor, c_nops, c_seq, c_deps = batch.chg_actor, batch.chg_nops, batch.chg_seq, batch.chg_deps
    for c in range(c0, c1):
        nops = int(c_nops[c])
        deps = {}
        for a in range(batch.max_actors):
            v = int(c_deps[c, a])
            if v:
                deps[actors[a]] = v
        ops = []
        for i in range(row, row + nops):
            act = int(batch.action[i])
            op = {"opId": oid(batch.op_id[i])}
            if act == abi.ACT_MAKELIST:
                op.update(action="makeList", obj=ROOT, key="text")
                text_obj = op["opId"]
            elif act == abi.ACT_INSERT:
                ref = int(batch.ref_a[i])
                op.update(action="set", obj=text_obj, elemId=oid(ref) if ref else HEAD, insert=True, value=batch.values[int(batch.payload[i])])
            elif act == abi.ACT_DELETE:
                op.update(action="del", obj=text_obj, elemId=oid(batch.ref_a[i]))
            elif act in (abi.ACT_ADDMARK, abi.ACT_REMOVEMARK):
                mt = int(batch.mark_type[i])
                sa, sb = int(batch.side_a[i]), int(batch.side_b[i])
                start = {"type": abi.SIDE_NAMES[sa]}
                if sa in (abi.SIDE_BEFORE, abi.SIDE_AFTER):
                    start["elemId"] = oid(batch.ref_a[i])
                end = {"type": abi.SIDE_NAMES[sb]}
                if sb in (abi.SIDE_BEFORE, abi.SIDE_AFTER):
                    end["elemId"] = oid(batch.ref_b[i])
                op.update(action="addMark" if act == abi.ACT_ADDMARK else "removeMark", obj=text_obj, start=start, end=end, markType=abi.MARK_NAMES[mt])
                if mt == abi.MARK_LINK and act == abi.ACT_ADDMARK:
                    op["attrs"] = {"url": batch.urls[int(batch.payload[i])]}
                elif mt == abi.MARK_COMMENT:
                    op["attrs"] = {"id": comments[int(batch.payload[i])]}
            elif act in (abi.ACT_MAPSET, abi.ACT_MAPDEL):  # an op on a map object (the root map or a nested one)
                ref = int(batch.ref_a[i])
                op.update(obj=oid(ref) if ref else ROOT, key=batch.keys[int(batch.ref_b[i])])
                if act == abi.ACT_MAPDEL:
                    op["action"] = "del"
                else:
                    kind = int(batch.mark_type[i])
                    op["action"] = "makeMap" if kind == abi.MAPV_MAP else "makeList" if kind == abi.MAPV_LIST else "set"
                    if kind == abi.MAPV_SCALAR:
                        op["value"] = json.loads(batch.map_values[int(batch.payload[i])])
            else:
                raise ValueError("row %d of log %d is not an op this engine models" % (i - b0, log))
            ops.append(op)
        start_op = int(batch.op_id[row]) >> 32 if nops else 0
        out.append({"actor": actors[int(c_actor[c])], "seq": int(c_seq[c]), "deps": deps, "startOp": start_op, "ops": ops})
        row += nops
    return out


# ---- batches made by the on-device generator (ptx_generate / gen_core.h): fixed string tables ----
GEN_VALUES = [chr(i) for i in range(128)]               # payload of an insert = the character's code
GEN_URLS = [chr(65 + i) + ".com" for i in range(26)]    # payload of a link addMark = the letter


def generated_tables(n_docs, replicas, n_comments):
    """Decode tables of a generated batch: actors "doc1".., comment ids "comment-<k>" ranked in string order."""
    actors = [["doc%d" % (i + 1) for i in range(replicas)] for _ in range(n_docs)]
    comments = [sorted(("comment-%d" % k for k in range(int(c))), key=_u16key) for c in n_comments]
    log_doc = [d for d in range(n_docs) for _ in range(replicas)]
    return actors, comments, log_doc


class Patches:
    """Patch streams of a batch (ptx_replay_patches): records of log l = patches[patch_off[l] : patch_off[l] + logs[l].n_patches]."""

    def __init__(self, patch_off, logs, patches, kernel_ms=0.0, launches=1):
        self.patch_off, self.logs, self.patches, self.kernel_ms, self.launches = patch_off, logs, patches, kernel_ms, launches

    def of_log(self, log):
        b0 = int(self.patch_off[log])
        return self.patches[b0:b0 + int(self.logs[log]["n_patches"])]


def _marks_of_attr(batch, attr, comment_ids, comments):
    marks = {}
    if attr & abi.ATTR_STRONG:
        marks["strong"] = {"active": True}
    if attr & abi.ATTR_EM:
        marks["em"] = {"active": True}
    if attr & abi.ATTR_COMMENT:
        marks["comment"] = [{"id": comments[i]} for i in sorted(comment_ids)]
    if attr & abi.ATTR_LINK:
        marks["link"] = {"url": batch.urls[attr & abi.ATTR_ID_MASK]}
    return marks


def decode_patches(batch, pat, log, with_rows=False):
    """Patch[] of one log in application order, in the reference's shapes (reference/src/micromerge.ts:214-222 Patch,
    :661-671 insert, :696-703 delete, src/peritext.ts:251-281 add/removeMark).  The makeList patch (the op itself,
    micromerge.ts:575) is reduced to {"action": "makeList"}.  Raises on a log without a stream.
    with_rows: add "_row" (the op row that produced the patch) and, on comment patches, "_commentId" — the reference's
    removeMark patch does not say WHICH comment went away (peritext.ts:262 attaches attrs to addMark only)."""
    st = int(pat.logs[log]["status"])
    if st != 0:
        raise ValueError(abi.STATUS_NAMES.get(st, "error %d" % st))
    b0 = int(batch.log_off[log])
    comments = batch.doc_comments[batch.log_doc[log]]
    rows = pat.of_log(log)
    out = []
    k = 0
    while k < len(rows):
        r = rows[k]
        kind, row, a, b = int(r["kind"]), int(r["row"]), int(r["a"]), int(r["b"])
        k += 1
        if kind == abi.PATCH_MAKELIST:
            out.append({"action": "makeList"})
        elif kind == abi.PATCH_INSERT:
            ids = []
            while k < len(rows) and int(rows[k]["kind"]) == abi.PATCH_INSERT_COMMENT:
                ids.append(int(rows[k]["a"]))
                k += 1
            out.append({"path": ["text"], "action": "insert", "index": a, "values": [batch.values[int(batch.payload[b0 + row])]],
                        "marks": _marks_of_attr(batch, b, ids, comments)})
        elif kind == abi.PATCH_DELETE:
            out.append({"path": ["text"], "action": "delete", "index": a, "count": b})
        elif kind in (abi.PATCH_ADDMARK, abi.PATCH_REMOVEMARK):
            mt = int(batch.mark_type[b0 + row])
            p = {"action": "addMark" if kind == abi.PATCH_ADDMARK else "removeMark", "markType": abi.MARK_NAMES[mt], "path": ["text"],
                 "startIndex": a, "endIndex": b}
            if kind == abi.PATCH_ADDMARK and mt == abi.MARK_LINK:
                p["attrs"] = {"url": batch.urls[int(batch.payload[b0 + row]) & abi.ATTR_ID_MASK]}
            elif kind == abi.PATCH_ADDMARK and mt == abi.MARK_COMMENT:
                p["attrs"] = {"id": comments[int(batch.payload[b0 + row])]}
            if with_rows and mt == abi.MARK_COMMENT:
                p["_commentId"] = comments[int(batch.payload[b0 + row])]
            out.append(p)
        else:
            raise ValueError("unknown patch kind %d" % kind)
        if with_rows:
            out[-1]["_row"] = row
    return out


def prosemirror_doc(spans):
    """prosemirrorDocFromCRDT (reference/src/bridge.ts:394-414, marks :369-391) as the JSON prosemirror-model's Node.toJSON() gives:
    doc > paragraph > text nodes, marks in ALL_MARKS order (schema.ts:125; = the rank order of the schema's mark table), attrs only
    for the types that declare some (comment {id}, link {url}; schema.ts:45-96), adjacent spans whose ProseMirror marks are equal
    (e.g. `comment: []` next to no comment key) joined into one text node as Fragment.fromArray does, the single empty span of an
    empty document -> an empty paragraph (:399-401).  The rules the reference's own source decides are pinned by
    tests/golden/pm_docs.json (oracle/gen_pm_golden.js reads them from the reference's schema.ts); ProseMirror's toJSON / joining
    rules are restated from its documented behaviour (not in this image): PARITY UNPINNED for those."""
    if len(spans) == 1 and spans[0]["text"] == "":
        return {"type": "doc", "content": [{"type": "paragraph"}]}
    text = []
    for s in spans:
        if s["text"] == "":
            raise ValueError("Empty text nodes are not allowed")  # what prosemirror-model's schema.text("") throws
        marks = []
        for t in abi.MARK_NAMES:
            v = s["marks"].get(t)
            if v is None:
                continue
            if isinstance(v, list):
                marks += [{"type": t, "attrs": {"id": one["id"]}} for one in v]
            elif t == "link":
                marks.append({"type": t, "attrs": {"url": v["url"]}})
            else:
                marks.append({"type": t})
        if text and text[-1].get("marks", []) == marks:
            text[-1]["text"] += s["text"]
            continue
        node = {"type": "text"}
        if marks:
            node["marks"] = marks
        node["text"] = s["text"]
        text.append(node)
    paragraph = {"type": "paragraph"}
    if text:
        paragraph["content"] = text
    return {"type": "doc", "content": [paragraph]}


def _elements(batch, res, log):
    """(op_id, rank, deleted) of every list element of a log, from the elem_rank output column."""
    b0, b1 = int(batch.log_off[log]), int(batch.log_off[log + 1])
    rk = res.elem_rank[b0:b1]
    ins = np.flatnonzero(batch.action[b0:b1] == abi.ACT_INSERT)
    return batch.op_id[b0:b1][ins], (rk[ins] & abi.RANK_MASK).astype(np.int64), (rk[ins] & abi.RANK_TOMBSTONE) != 0


def resolve_cursor(batch, res, log, elem_id):
    """Micromerge.resolveCursor (reference/src/micromerge.ts:475-477): visible index of a cursor's element =
    visible elements before it (the element itself may be a tombstone).  Raises like findListElement (:752)."""
    d = batch.log_doc[log]
    ctr, actor = split_op_id(elem_id)
    if actor not in batch.doc_actors[d]:
        raise ValueError("List element not found")
    want = np.uint64(_pack(ctr, batch.doc_actors[d].index(actor)))
    ids, rank, dead = _elements(batch, res, log)
    hit = np.flatnonzero(ids == want)
    if len(hit) == 0:
        raise ValueError("List element not found")
    r = rank[hit[0]]
    return int(np.count_nonzero((rank < r) & ~dead))


def get_cursor(batch, res, log, index):
    """Micromerge.getCursor (reference/src/micromerge.ts:465-473): the elemId of the visible element at `index`."""
    d = batch.log_doc[log]
    ids, rank, dead = _elements(batch, res, log)
    alive = np.flatnonzero(~dead)
    order = alive[np.argsort(rank[alive])]
    if not 0 <= index < len(order):
        raise ValueError("List index out of bounds: %d" % index)
    v = int(ids[order[index]])
    return "%d@%s" % (v >> 32, batch.doc_actors[d][v & 0xFFFFFFFF])


# ---- on-disk form of a batch (SURVEY.md §5 "checkpoint / resume": the reference only has JSON.stringify dumps of
# Change objects, test/fuzz.ts:16-20; replaying a saved op log = re-running the merge) ----
_COLUMNS = ("log_off", "op_id", "ref_a", "ref_b", "payload", "action", "mark_type", "side_a", "side_b", "chg_off", "chg_hdr", "chg_env", "chg_env_hi")


def save_batch(path, batch):
    """Write a Batch as one .npz: the SoA columns as they go to the device + the decode tables as JSON."""
    import json

    meta = {"format": "peritext-soa-oplog", "abi": abi.PTX_ABI_VERSION, "max_actors": batch.max_actors, "values": batch.values, "urls": batch.urls, "keys": batch.keys, "map_values": batch.map_values,
            "log_doc": batch.log_doc, "doc_actors": batch.doc_actors, "doc_comments": batch.doc_comments}
    arrays = {k: getattr(batch, k) for k in _COLUMNS if getattr(batch, k) is not None}  # an envelope-less batch has no chg_* columns
    if batch.log_hdr is not None:
        arrays["log_hdr"] = batch.log_hdr
    np.savez_compressed(path, meta=np.frombuffer(json.dumps(meta).encode("utf-8"), dtype=np.uint8), **arrays)


def load_batch(path):
    import json

    with np.load(path) as z:
        meta = json.loads(bytes(z["meta"]).decode("utf-8"))
        if meta.get("format") != "peritext-soa-oplog" or meta.get("abi") not in (5, 6, abi.PTX_ABI_VERSION):  # (the file format has not changed since ABI 5: v6 added the optional chg_env_hi column, v7 changed ptx_result only)
            raise ValueError("not a peritext SoA op-log file of ABI %d" % abi.PTX_ABI_VERSION)
        cols = {k: (z[k] if k in z.files else None) for k in _COLUMNS}
        hdr = z["log_hdr"] if "log_hdr" in z.files else None
    return Batch(log_hdr=hdr, max_actors=int(meta["max_actors"]), values=meta["values"], urls=meta["urls"], log_doc=meta["log_doc"],
                 doc_actors=meta["doc_actors"], doc_comments=meta["doc_comments"], keys=meta.get("keys", []), map_values=meta.get("map_values", []), **cols)
