#!/usr/bin/env node
"use strict"
/*
 * "COMPILES" THE REFERENCE ITSELF — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * The reference is TypeScript and this image has no tsc.  This script is the recipe that turns the
 * reference's own hot-path sources, read from where they lie under /root/reference,
 *      src/micromerge.ts   src/peritext.ts   (+ the markSpec table of src/schema.ts:45-96)
 * into runnable Node-12 JavaScript by ERASING types only (no logic is rewritten): output goes to
 * oracle/_ref/ which is git-ignored (it is a build artefact of the reference, like a .so) but does
 * travel to the GPU box with gpurun.  Nothing from /root/reference is ever committed.
 *
 * What oracle/_ref is used for:
 *   - differential fuzzing of oracle/peritext_oracle.js against the real reference logic
 *     (oracle/diff_fuzz.js; also the generator of tests/golden/ref_fuzz_*.json);
 *   - the `"kind": "reference"` cpu_baseline of bench.py (the reference's own code timed on the GPU
 *     box's host cores).
 *
 * Every rewrite below is an exact-text rule that must match the stated number of times; if the
 * reference ever changes shape this fails loudly instead of producing a silently different program.
 *
 * Usage: node oracle/build_ref.js [--ref /root/reference] [--out oracle/_ref]
 */
const fs = require("fs")
const path = require("path")
const vm = require("vm")

const argv = process.argv.slice(2)
function flag(name, dflt) {
    const i = argv.indexOf(name)
    return i >= 0 ? argv[i + 1] : dflt
}
const refRoot = flag("--ref", "/root/reference")
const outDir = flag("--out", path.join(__dirname, "_ref"))

function die(msg) {
    console.error("build_ref: " + msg)
    process.exit(2)
}
if (!fs.existsSync(path.join(refRoot, "src", "micromerge.ts"))) die("reference not found under " + refRoot)

/* ---- generic passes ---- */

/** Drop `import ...` statements (single- or multi-line). */
function dropImports(src) {
    return src.replace(/^import\s[\s\S]*?from\s+"[^"]+"\s*;?\s*$/gm, "")
}

/** Drop type-level declarations: `type X = ...`, `interface X {...}` (exported or not, any indent). */
function dropTypeDecls(src) {
    const lines = src.split("\n")
    const out = []
    let i = 0
    const startsDecl = l => /^\s*(export\s+)?(type\s+\w+(<[^=]*>)?\s*=|interface\s+\w+)/.test(l)
    while (i < lines.length) {
        if (!startsDecl(lines[i])) {
            out.push(lines[i])
            i++
            continue
        }
        let depth = 0
        let j = i
        for (;;) {
            const l = lines[j].replace(/\/\*.*?\*\//g, "").replace(/\/\/.*$/, "").replace(/=>/g, "")
            for (const ch of l) {
                if (ch === "{" || ch === "(" || ch === "[" || ch === "<") depth++
                else if (ch === "}" || ch === ")" || ch === "]" || ch === ">") depth--
            }
            const t = l.trim()
            const next = j + 1 < lines.length ? lines[j + 1].trim() : ""
            const continues =
                depth > 0 ||
                /[=|&?:,]$/.test(t) ||
                t === "" && depth > 0 ||
                /^[|&?:]/.test(next) ||
                /^(\/\/|\/\*|\*)/.test(next) && depth > 0
            if (!continues) break
            j++
        }
        i = j + 1
    }
    return out.join("\n")
}

function applyRules(src, rules, file) {
    for (const rule of rules) {
        const from = rule[0]
        const to = rule[1]
        const want = rule.length > 2 ? rule[2] : 1
        let count = 0
        if (from instanceof RegExp) {
            src = src.replace(from, (...m) => {
                count++
                return typeof to === "function" ? to(...m) : to.replace(/\$(\d)/g, (_, d) => m[+d])
            })
        } else {
            let idx = src.indexOf(from)
            while (idx >= 0) {
                count++
                src = src.slice(0, idx) + to + src.slice(idx + from.length)
                idx = src.indexOf(from, idx + to.length)
            }
        }
        if (count !== want) die(`${file}: rule ${String(from).slice(0, 70)} matched ${count}x, expected ${want}x`)
    }
    return src
}

/* ---- micromerge.ts ---- */
const mmRules = [
    ["export default class Micromerge {", "class Micromerge {"],
    ['    public static contentKey: CONTENT_KEY = "text"', '    static contentKey = "text"'],
    ["    public actorId: string\n", "\n"],
    ["    private seq: number = 0", "    seq = 0"],
    ["    private maxOp: number = 0", "    maxOp = 0"],
    ["    public clock: Record<string, number> = {}", "    clock = {}"],
    ["    private objects: Record<ObjectId, JsonComposite> & Record<typeof ROOT, Record<string, Json>> = {", "    objects = {"],
    ["    private metadata: Record<ObjectId, Metadata> = {", "    metadata = {"],
    ["    constructor(actorId: string = uuid.v4()) {", "    constructor(actorId = uuid.v4()) {"],
    ["    get root(): Record<string, Json> {", "    get root() {"],
    ["    public getRoot<T extends Record<string, Json>>(): Partial<T> {", "    getRoot() {"],
    ["return this.objects[ROOT] as T", "return this.objects[ROOT]"],
    ["    public change(ops: Array<InputOperation>): {\n        change: Change\n        patches: Patch[]\n    } {", "    change(ops) {"],
    ["const change: Change = {", "const change = {"],
    ["const patchesForChange: Patch[] = []", "const patchesForChange = []"],
    ['    getObjectIdForPath(path: InputOperation["path"]): ObjectId {', "    getObjectIdForPath(path) {"],
    ["let objectId: ObjectId = ROOT", "let objectId = ROOT"],
    ["const meta: Metadata = this.metadata[objectId]", "const meta = this.metadata[objectId]"],
    ["const childId: ObjectId | undefined = meta[CHILDREN][pathElem]", "const childId = meta[CHILDREN][pathElem]"],
    ["    public getCursor(path: OperationPath, index: number): Cursor {", "    getCursor(path, index) {"],
    ["    public resolveCursor(cursor: Cursor): number {", "    resolveCursor(cursor) {"],
    [
        '    private makeNewOp(\n        change: Change,\n        op: DistributiveOmit<Operation, "opId">,\n    ): { opId: OperationId; patches: Patch[] } {',
        "    makeNewOp(change, op) {",
    ],
    ["    applyChange(change: Change): Patch[] {", "    applyChange(change) {"],
    ["    public getTextWithFormatting(path: OperationPath): Array<FormatSpanWithText> {", "    getTextWithFormatting(path) {"],
    ["    private applyOp = (op: Operation): Patch[] => {", "    applyOp = (op) => {"],
    ["    private applyListInsert(op: InsertOperation): Patch[] {", "    applyListInsert(op) {"],
    ["    private applyListUpdate(op: DeleteOperation): Patch[] {", "    applyListUpdate(op) {"],
    [
        "    private findListElement(\n        objectId: ObjectId,\n        elemId: ElemId,\n    ): {\n        index: number\n        visible: number\n    } {",
        "    findListElement(objectId, elemId) {",
    ],
    [
        "export function getListElementId(\n    meta: Metadata,\n    index: number,\n    options?: { lookAfterTombstones: boolean },\n): OperationId {",
        "function getListElementId(meta, index, options) {",
    ],
    ["let latestIndexAfterTombstone: number | undefined", "let latestIndexAfterTombstone"],
    ["if (options?.lookAfterTombstones) {", "if (options && options.lookAfterTombstones) {"],
    ["export function compareOpIds(id1: OperationId, id2: OperationId): -1 | 0 | 1 {", "function compareOpIds(id1, id2) {"],
    /* `unreachable` is only declared in globals.d.ts:10, never defined; keep the calls, define a thrower */
]
const mmHeader = `"use strict"
/* GENERATED by oracle/build_ref.js from reference/src/micromerge.ts — types erased, logic untouched. DO NOT COMMIT. */
const uuid = { v4() { return "actor-" + Math.random().toString(16).slice(2) } }
function unreachable(x) { throw new ReferenceError("unreachable is not defined (globals.d.ts:10) " + String(x)) }
const __peritext = require("./peritext")
const changeMark = (...a) => __peritext.changeMark(...a)
const applyAddRemoveMark = (...a) => __peritext.applyAddRemoveMark(...a)
const getActiveMarksAtIndex = (...a) => __peritext.getActiveMarksAtIndex(...a)
const getTextWithFormatting = (...a) => __peritext.getTextWithFormatting(...a)
`
const mmFooter = `
/* assign onto the existing exports object: peritext.js already holds a reference to it (circular require) */
Object.assign(module.exports, { default: Micromerge, Micromerge, getListElementId, compareOpIds, ROOT, HEAD, CHILDREN })
`

/* ---- peritext.ts ---- */
const ptRules = [
    [
        "export function applyAddRemoveMark(op: MarkOperation, object: Json, metadata: ListMetadata): Patch[] {",
        "function applyAddRemoveMark(op, object, metadata) {",
    ],
    ["const patches: Patch[] = []", "const patches = []"],
    ["    ]).flat() as Positions;", "    ]).flat();"],
    ["let currentOps = new Set<MarkOperation>()", "let currentOps = new Set()"],
    ['let opState: MarkOpState = "BEFORE"', 'let opState = "BEFORE"'],
    ["let partialPatch: PartialPatch | undefined", "let partialPatch"],
    ["const objLength = object.length as number", "const objLength = object.length"],
    [
        "function calculateOpsForPosition(\n    op: MarkOperation, currentOps: Set<MarkOperation>,\n    side: MarkOpsPosition,\n    elMeta: ListItemMetadata,\n    opState: MarkOpState): [opState: MarkOpState, newOps?: Set<MarkOperation>] {",
        "function calculateOpsForPosition(op, currentOps, side, elMeta, opState) {",
    ],
    ["function beginPartialPatch(\n    op: MarkOperation,\n    startIndex: number\n): PartialPatch {", "function beginPartialPatch(op, startIndex) {"],
    ["const partialPatch: PartialPatch = {", "const partialPatch = {"],
    [
        "function finishPartialPatch(partialPatch: PartialPatch, endIndex: number, length: number): Patch | undefined {",
        "function finishPartialPatch(partialPatch, endIndex, length) {",
    ],
    ["const patch = { ...partialPatch, endIndex: Math.min(endIndex, length) } as AddMarkOperationInput | RemoveMarkOperationInput", "const patch = { ...partialPatch, endIndex: Math.min(endIndex, length) }"],
    ["export function opsToMarks(ops: Set<MarkOperation>): MarkMap {", "function opsToMarks(ops) {"],
    ["const markMap: MarkMap = {}", "const markMap = {}"],
    ["const opIdMap: Record<MarkType, OperationId> = {}", "const opIdMap = {}"],
    [
        'if (op.action === "addMark" && !markMap[op.markType]?.find(c => c.id === op.attrs.id)) {',
        'if (op.action === "addMark" && !(markMap[op.markType] === undefined || markMap[op.markType] === null ? undefined : markMap[op.markType].find(c => c.id === op.attrs.id))) {',
    ],
    ["export function getActiveMarksAtIndex(metadata: ListMetadata, index: number): MarkMap {", "function getActiveMarksAtIndex(metadata, index) {"],
    [
        "export function getTextWithFormatting(text: Json, metadata: ListMetadata): Array<FormatSpanWithText> {",
        "function getTextWithFormatting(text, metadata) {",
    ],
    ["const spans: FormatSpanWithText[] = []", "const spans = []"],
    ["let characters: string[] = []", "let characters = []"],
    ["let marks: MarkMap = {}", "let marks = {}"],
    ["let newMarks: MarkMap | undefined", "let newMarks"],
    ["opsToMarks(metadata[index - 1].markOpsAfter!)", "opsToMarks(metadata[index - 1].markOpsAfter)"],
    ["characters.push(text[visible] as string)", "characters.push(text[visible])"],
    [
        'function findClosestMarkOpsToLeft(args: {\n    index: number\n    side: "before" | "after"\n    metadata: ListMetadata\n}): Set<MarkOperation> {',
        "function findClosestMarkOpsToLeft(args) {",
    ],
    ["let ops = new Set<MarkOperation>()", "let ops = new Set()"],
    ["return new Set(metadata[index].markOpsBefore!)", "return new Set(metadata[index].markOpsBefore)"],
    [
        "export function addCharactersToSpans(args: {\n    characters: string[]\n    marks: MarkMap\n    spans: FormatSpanWithText[]\n}): void {",
        "function addCharactersToSpans(args) {",
    ],
    [
        'export function changeMark(\n    inputOp: AddMarkOperationInput | RemoveMarkOperationInput,\n    objId: ObjectId,\n    meta: ListMetadata,\n    obj: Json[] | (Json[] & Record<string, Json>)): DistributiveOmit<AddMarkOperation | RemoveMarkOperation, "opId"> {',
        "function changeMark(inputOp, objId, meta, obj) {",
    ],
    ["let start: BoundaryPosition", "let start"],
    ["let end: BoundaryPosition", "let end"],
    [
        'const partialOp: DistributiveOmit<AddMarkOperation | RemoveMarkOperation, "opId"> = {',
        "const partialOp = {",
    ],
]
const ptHeader = `"use strict"
/* GENERATED by oracle/build_ref.js from reference/src/peritext.ts — types erased, logic untouched. DO NOT COMMIT. */
const { isEqual, sortBy } = require("./lodash_pair")
const { markSpec } = require("./schema")
const __mm = require("./micromerge") /* circular: resolved lazily, as the ES-module original does */
const Micromerge = { get contentKey() { return __mm.default.contentKey } }
const compareOpIds = (a, b) => __mm.compareOpIds(a, b)
const getListElementId = (m, i, o) => __mm.getListElementId(m, i, o)
`
const ptFooter = `
module.exports = { applyAddRemoveMark, opsToMarks, getActiveMarksAtIndex, getTextWithFormatting, addCharactersToSpans, changeMark }
`

/* ---- schema.ts: the two flags the CRDT reads from markSpec (schema.ts:45-96) + what the ProseMirror doc build depends on
 *      (bridge.ts:369-414): the attribute names of every mark type, ALL_MARKS (:125) and the key order of the schema's mark table
 *      (demoMarkSpec :99-121 = markSpec + two demo-only types; ProseMirror ranks marks by that order) ---- */
function buildSchema(src) {
    const spec = {}
    for (const t of ["strong", "em", "comment", "link"]) {
        const m = new RegExp("\\n    " + t + ": \\{([\\s\\S]*?)\\n    \\},").exec(src)
        if (!m) die("schema.ts: markSpec entry not found: " + t)
        const inc = /inclusive:\s*(true|false)/.exec(m[1])
        const multi = /allowMultiple:\s*(true|false)/.exec(m[1])
        if (!inc || !multi) die("schema.ts: flags not found for " + t)
        const attrs = /attrs:\s*\{([\s\S]*?)\n        \},/.exec(m[1])
        const names = attrs ? (attrs[1].match(/([A-Za-z_]+):\s*\{\}/g) || []).map(x => x.split(":")[0]) : []
        spec[t] = { inclusive: inc[1] === "true", allowMultiple: multi[1] === "true", attrs: names }
    }
    const all = /export const ALL_MARKS = \[([^\]]*)\]/.exec(src)
    if (!all) die("schema.ts: ALL_MARKS not found")
    const allMarks = (all[1].match(/"([a-z]+)"/g) || []).map(x => x.replace(/"/g, ""))
    const specKeys = []
    const body = /export const markSpec = \{([\s\S]*?)\n\} as const/.exec(src)
    if (!body) die("schema.ts: markSpec body not found")
    for (const m of body[1].matchAll ? body[1].matchAll(/\n    ([a-zA-Z]+): \{/g) : []) specKeys.push(m[1])
    const demo = /export const demoMarkSpec = \{([\s\S]*?)\n\}\n/.exec(src)
    if (!demo || !/\.\.\.markSpec,/.test(demo[1])) die("schema.ts: demoMarkSpec does not start with ...markSpec")
    const demoKeys = []
    for (const m of demo[1].matchAll(/\n    ([a-zA-Z]+): \{/g)) demoKeys.push(m[1])
    return (
        '"use strict"\n/* GENERATED by oracle/build_ref.js from reference/src/schema.ts:45-125 (flags, attribute names, mark order). DO NOT COMMIT. */\n' +
        "module.exports = { markSpec: " + JSON.stringify(spec) + ", ALL_MARKS: " + JSON.stringify(allMarks) + ", markRankOrder: " + JSON.stringify(specKeys.concat(demoKeys)) + " }\n"
    )
}

/*
 * lodash isEqual / sortBy (peritext.ts:2): the system ships Debian's per-method packages
 * lodash.isequal / lodash.sortby 4.17.21 == the version pinned in the reference's package-lock.json.
 * Use them when present so that the reference runs on its real dependency; otherwise fall back to
 * the oracle's structural stand-ins.
 */
const lodashPair = `"use strict"
/* GENERATED by oracle/build_ref.js. DO NOT COMMIT. */
let isEqual, sortBy, real = true
try {
    isEqual = require("lodash.isequal")
    sortBy = require("lodash.sortby")
} catch (e) {
    real = false
    const O = require("../peritext_oracle")
    isEqual = O.deepEqual
    sortBy = (list, key) => list.map((v, i) => [v, i]).sort((p, q) => (key(p[0]) < key(q[0]) ? -1 : key(p[0]) > key(q[0]) ? 1 : p[1] - q[1])).map(p => p[0])
}
module.exports = { isEqual, sortBy, real }
`

function build(file, rules, header, footer) {
    let src = fs.readFileSync(path.join(refRoot, "src", file), "utf8")
    src = dropImports(src)
    src = dropTypeDecls(src)
    src = applyRules(src, rules, file)
    src = src.replace(/^export const /gm, "const ")
    const js = header + src + footer
    try {
        new vm.Script(js, { filename: file + " (erased)" })
    } catch (e) {
        fs.mkdirSync(outDir, { recursive: true })
        fs.writeFileSync(path.join(outDir, file.replace(/\.ts$/, ".failed.js")), js)
        die(file + ": erased source does not parse: " + e.message + "\n" + String(e.stack).split("\n").slice(0, 6).join("\n"))
    }
    return js
}

fs.mkdirSync(outDir, { recursive: true })
fs.writeFileSync(path.join(outDir, "schema.js"), buildSchema(fs.readFileSync(path.join(refRoot, "src", "schema.ts"), "utf8")))
fs.writeFileSync(path.join(outDir, "lodash_pair.js"), lodashPair)
fs.writeFileSync(path.join(outDir, "peritext.js"), build("peritext.ts", ptRules, ptHeader, ptFooter))
fs.writeFileSync(path.join(outDir, "micromerge.js"), build("micromerge.ts", mmRules, mmHeader, mmFooter))
const mm = require(path.resolve(outDir, "micromerge.js"))
if (typeof mm.default !== "function" || typeof mm.compareOpIds !== "function") die("built module does not export Micromerge")
console.log("build_ref: wrote " + outDir + "/{micromerge,peritext,schema,lodash_pair}.js (real lodash: " + require(path.resolve(outDir, "lodash_pair.js")).real + ")")
