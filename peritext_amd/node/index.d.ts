/**
 * Types of the MI355X batch merge engine's TypeScript/JS host.  The hot-path types are the reference's own
 * (reference/src/micromerge.ts:25-71,:133-212; src/peritext.ts:17-65,:35-38,:125-137): a maintainer can import
 * them from the reference instead — they are restated here so that the package type-checks on its own.
 */
export type ActorId = string
export type OperationId = string /** `${counter}@${actorId}` (micromerge.ts:35) */
export type Clock = Record<ActorId, number>
export type MarkType = "strong" | "em" | "comment" | "link"

export type BoundaryPosition =
    | { type: "before"; elemId: OperationId }
    | { type: "after"; elemId: OperationId }
    | { type: "startOfText" }
    | { type: "endOfText" }

export interface InsertOperation { opId: OperationId; action: "set"; obj: OperationId; elemId?: OperationId | "_head"; insert: true; value: string }
export interface DeleteOperation { opId: OperationId; action: "del"; obj: OperationId; elemId: OperationId }
export interface MakeListOperation { opId: OperationId; action: "makeList"; obj?: OperationId | "_root"; key: string }
export interface AddMarkOperation { opId: OperationId; action: "addMark"; obj: OperationId; start: BoundaryPosition; end: BoundaryPosition; markType: MarkType; attrs?: { url?: string; id?: string } }
export interface RemoveMarkOperation { opId: OperationId; action: "removeMark"; obj: OperationId; start: BoundaryPosition; end: BoundaryPosition; markType: MarkType; attrs?: { id?: string } }
export type Operation = InsertOperation | DeleteOperation | MakeListOperation | AddMarkOperation | RemoveMarkOperation | { opId: OperationId; action: string; [k: string]: unknown }

/** micromerge.ts:133-148 — what Micromerge.change(ops) takes: index-based operations on the text list (the only path this host serves) */
export type InputOperation =
    | { path: []; action: "makeList"; key: "text" }
    /** on a map object — the root map (path []) or a map nested in it (path = the keys down from the root), micromerge.ts:109-131 */
    | { path: string[]; action: "makeMap" | "makeList"; key: string }
    | { path: string[]; action: "set"; key: string; value: JsonValue }
    | { path: string[]; action: "del"; key: string }
    | { path: ["text"]; action: "insert"; index: number; values: string[] }
    | { path: ["text"]; action: "delete"; index: number; count: number }
    | { path: ["text"]; action: "addMark"; startIndex: number; endIndex: number; markType: MarkType; attrs?: { url?: string; id?: string } }
    | { path: ["text"]; action: "removeMark"; startIndex: number; endIndex: number; markType: MarkType; attrs?: { id?: string } }

/** micromerge.ts:60-71 */
export interface Change { actor: ActorId; seq: number; deps: Clock; startOp: number; ops: Operation[] }

export type MarkMap = { strong?: { active: true }; em?: { active: true }; link?: { url: string }; comment?: Array<{ id: string }> }
/** peritext.ts:35-38 */
export interface FormatSpanWithText { text: string; marks: MarkMap }

/** SoA op log of include/peritext_hip.h (one row per internal Operation, 32 bytes) + decode tables. */
export interface WireBatch {
    nLogs: number; nOps: number
    logOff: BigUint64Array; opId: BigUint64Array; refA: BigUint64Array; refB: BigUint64Array
    payload: Uint32Array; action: Uint8Array; markType: Uint8Array; sideA: Uint8Array; sideB: Uint8Array
    logHdr?: Uint32Array
    /** Change envelope (micromerge.ts:60-71) -> applyChange's causal admission runs on the device */
    chgOff?: BigUint64Array; maxActors?: number
    /** what the device reads: chgHdr = actorRank << 20 | nops, chgEnv rows of ((1 + maxActors + 3) & ~3) u16 = seq, deps[...] */
    chgHdr?: Uint32Array; chgEnv?: Uint16Array
    /** the wide envelope column (ptx_batch.chg_env_hi): high halves of chgEnv's values, present once some seq / dep exceeds 65534 */
    chgEnvHi?: Uint16Array
    /** the same unpacked (filled by encodeDocs / unpackEnvelope; decodeChanges and decodePatches read these) */
    chgActor?: Uint32Array; chgSeq?: Uint32Array; chgNops?: Uint32Array; chgDeps?: Uint32Array
    values: string[]; urls: string[]; logDoc: number[]; docActors: string[][]; docComments: string[][]
    /** keys of the map objects (ref_b of the PTX_ACT_MAPSET / MAPDEL / MAKELIST rows) and the JSON text of the values they set (payload) */
    keys?: string[]; mapValues?: string[]
    /** encodeDocs(listKeys): device log l merges the list under root key logList[l] ("a.b": the list under key b of the map under root key a — an OperationPath) of replica logReplica[l] of its document */
    logList?: string[]; logReplica?: number[]
}
export interface WireResult {
    logs: Uint32Array; values: Uint32Array; spans: Uint32Array; cintervals: Uint32Array; elemRank?: Uint32Array
    /** ABI 7: the rows are compact — log l's values at values[valueOff[l] .. valueOff[l + 1]), span rows (2 u32 each) from spanOff[l], comment intervals (3 u32 each) from cintOff[l] */
    valueOff: BigUint64Array; spanOff: BigUint64Array; cintOff: BigUint64Array
    /** with applyMaterialize(batch, true): ptx_patches (patch_off, {status, n_patches} per log, 4 u32 per record) */
    patchOff?: BigUint64Array; patchLogs?: Uint32Array; patches?: Uint32Array
}

/** micromerge.ts:214-222 — what applyChange returns (the makeList patch, the raw op upstream, is {action: "makeList"}) */
export type Patch =
    | { action: "makeList" }
    | { path: ["text"]; action: "insert"; index: number; values: string[]; marks: MarkMap }
    | { path: ["text"]; action: "delete"; index: number; count: number }
    | { path: ["text"]; action: "addMark"; markType: MarkType; startIndex: number; endIndex: number; attrs?: { url?: string; id?: string } }
    | { path: ["text"]; action: "removeMark"; markType: MarkType; startIndex: number; endIndex: number }

/** the columns of ptx_input_ops (include/peritext_hip.h) */
export interface WireInputOps {
    chgOff: BigUint64Array; opOff: BigUint64Array; action: Uint8Array; markType: Uint8Array
    index: Uint32Array; count: Uint32Array; payload: Uint32Array; values: Uint32Array; actor: Uint32Array; maxActors: number
}

/** Per-replica handle with the reference's calls (Micromerge.applyChange :499, change :308, getTextWithFormatting :516). */
export interface ReplicaHandle {
    /** causal admission (seq / deps) is checked HERE: throws RangeError like micromerge.ts:501-509 and leaves the replica untouched;
     *  the change itself is queued (all handles of an engine are merged in one launch); returns [] — see getPatches() */
    applyChange(change: Change): Patch[]
    /** Micromerge.change (micromerge.ts:308): InputOperations resolved against this replica's state on the device (ptx_change);
     *  throws RangeError("List index out of bounds") like :804.  Needs engine.replica(docId, actorId). */
    change(ops: InputOperation[]): { change: Change; patches: Patch[] }
    /** micromerge.ts:465-477, resolved on the device (ptx_resolve_cursors); RangeError like :804 / :752 */
    getCursor(path: ["text"], index: number): { objectId: OperationId | null; elemId: OperationId }
    resolveCursor(cursor: { objectId?: OperationId | null; elemId: OperationId }): number
    /** entry c = the Patch[] the reference's applyChange(c-th change) returns */
    getPatches(): Patch[][]
    /** throws RangeError("List element not found" | …) exactly where the reference's applyChange would have thrown */
    getTextWithFormatting(path: ["text"]): FormatSpanWithText[]
    /** Micromerge.getRoot() (micromerge.ts:443-449): the root map and the maps nested in it, resolved on the device (ptx_root_map: last
     *  writer wins per key, :572-602); list objects appear as { $list: true }.  RangeError("Object does not exist") like :538-540 */
    getRoot(): RootJson
}
/** a map object as JSON: nested maps, the values set, { $list: true } where a list object (the text) hangs */
export type RootJson = { [key: string]: RootJson | JsonValue | { $list: true } }
export type JsonValue = string | number | boolean | null | JsonValue[] | { [key: string]: JsonValue }
/** ptx_root_map's raw answer (include/peritext_hip.h ptx_root_maps) */
export interface WireRootMaps { entryOff: BigUint64Array; logs: Uint32Array; entries: Uint32Array }

export class MergeEngine {
    /** resident (default true): the logs of a document's replica() handles stay in HBM between calls — a read encodes and uploads only the Changes that
     *  arrived since (ptx_batch_append), merges the resident logs and fetches the Patch[] records of the new rows only (ptx_replay_patches_from).
     *  false: every read encodes, uploads and replays the whole log of every handle. */
    constructor(opts?: { device?: number; libPath?: string; addonPath?: string; resident?: boolean })
    /** what the resident replicas cost so far: documents uploaded whole (first read, or a new actor re-ranked its op ids), appends, rows sent to the device */
    readonly stats: { residentUploads: number; residentAppends: number; rowsUploaded: number; residentChanges: number; residentChangeMs: number; residentCursorCalls: number }
    close(): void
    applyMaterialize(batch: WireBatch, wantPatches?: boolean): WireResult
    /** docs -> replica logs -> changes in application order  =>  spans per replica log */
    /** without opts: the spans of the list under "text" per replica; with opts.listKeys (several list objects per document, round 5): every replica's entry is
     *  {key: getTextWithFormatting([key])} for the root keys named */
    applyChanges(docs: Change[][][], opts?: { listKeys: string[] }): FormatSpanWithText[][][] | Array<Array<{ [key: string]: FormatSpanWithText[] }>>
    /** spans as applyChanges + patches[doc][replica][change] = what applyChange(change) returns (micromerge.ts:499) */
    applyChangesWithPatches(docs: Change[][][]): { spans: FormatSpanWithText[][][]; patches: Patch[][][][] }
    /** getRoot() of every replica: roots[doc][replica] (ptx_root_map) */
    roots(docs: Change[][][]): RootJson[][]
    /** on-device change(): edit histories generated on the GPU (ptx_generate; the documents of oracle/ptxgen.js for the same seed) and merged there */
    generate(cfg: { replicas: number; opsPerLog: number; mix: [number, number, number, number]; markTypes: MarkType[]; seed: number; nDocs: number; firstDoc?: number; listCap?: number; initialText?: string }):
        { docs: Change[][][]; spans: FormatSpanWithText[][][]; kernelMs: number; batch: WireBatch }
    digests(docs: Change[][][]): Array<[bigint, bigint]>
    /** multi-GPU, one process per GPU: RCCL communicator (id made on rank 0, carried by the host's own channel) */
    commUniqueId(): Uint8Array
    commInit(id: Uint8Array, rank: number, nRanks: number): unknown
    commDestroy(comm: unknown): void
    /** merge this rank's documents, all-gather the digests of all ranks on the device (ptx_allgather_digests), count the converged documents of the job */
    convergedDocs(docs: Change[][][], comm: unknown, counts: number[], replicas: number): { converged: number; total: number; digests: Array<[bigint, bigint]>; statuses: number[] }
    /** Micromerge.change for many replicas in one call: calls[d][r] = change() calls of replica r of document d */
    changeMany(docs: Change[][][], calls: InputOperation[][][][], actors: ActorId[][], opts?: { extraComments?: string[][] }): { changes: Change[][][]; status: number[][] }
    replica(docId?: number | string, actorId?: ActorId): ReplicaHandle
    flush(wantPatches?: boolean): void
}
export function encodeDocs(docs: Change[][][], opts?: { extraActors?: ActorId[][]; extraComments?: string[][]; textObjs?: Array<OperationId | null>; listKeys?: string[] }): WireBatch
export function encodeInputOps(batch: WireBatch, perLog: InputOperation[][][], actors: ActorId[]): WireInputOps
export function packEnvelope(batch: WireBatch): WireBatch
export function unpackEnvelope(batch: WireBatch): WireBatch
export function decodeSpans(batch: WireBatch, res: WireResult, log: number): FormatSpanWithText[]
export function decodePatches(batch: WireBatch, res: WireResult, log: number): Patch[][]
export function decodeChanges(batch: WireBatch, log: number, textObjOfLog?: OperationId | null): Change[]
export function decodeRoot(batch: WireBatch, rm: WireRootMaps, log: number): RootJson
/** bridge.ts:394-414 prosemirrorDocFromCRDT, as the Node.toJSON() form of the document (parity unpinned: no ProseMirror in the build image) */
export function prosemirrorDocFromSpans(spans: FormatSpanWithText[]): { type: "doc"; content: Array<{ type: "paragraph"; content?: Array<{ type: "text"; text: string; marks?: Array<{ type: MarkType; attrs?: Record<string, string> }> }> }> }
export function census(batch: WireBatch): Uint32Array
