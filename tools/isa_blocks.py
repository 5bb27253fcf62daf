#!/usr/bin/env python3
"""Static view of a kernel's gfx950 assembly (hipcc -S): basic blocks with their VALU / SALU / LDS / VMEM / branch instruction
counts and the loops (backward branches) they belong to.  The merge kernel is VALU-issue bound, so the instruction count of its
loop bodies is the number to push down between GPU runs.  Usage: isa_blocks.py file.s kernel_name [min_instrs]"""
import re
import sys


def classify(op):
    if op.startswith("v_"):
        return "valu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith(("s_cbranch", "s_branch")):
        return "br"
    if op.startswith(("s_waitcnt", "s_nop", "s_barrier")):
        return "wait"
    if op.startswith("s_load") or op.startswith("s_buffer_load"):
        return "smem"
    if op.startswith("s_"):
        return "salu"
    return "other"


def main():
    path, kern = sys.argv[1], sys.argv[2]
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith(kern + ":"))
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    blocks = []  # (label, line, counts, branch targets)
    cur = {"label": "entry", "line": start, "n": {}, "tg": []}
    order = {}
    for i in range(start + 1, end):
        l = lines[i]
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            blocks.append(cur)
            cur = {"label": m.group(1), "line": i, "n": {}, "tg": []}
            continue
        s = l.strip()
        if not s or s.startswith((";", ".", "//")):
            continue
        op = s.split()[0]
        k = classify(op)
        cur["n"][k] = cur["n"].get(k, 0) + 1
        if k == "br":
            t = s.split()[-1]
            cur["tg"].append(t)
    blocks.append(cur)
    for bi, b in enumerate(blocks):
        order[b["label"]] = bi
    # loops: backward branch from block j to block i <= j
    loops = []
    for bi, b in enumerate(blocks):
        for t in b["tg"]:
            if t in order and order[t] <= bi:
                loops.append((order[t], bi))
    loops.sort(key=lambda x: (x[0], -x[1]))
    print("%d blocks, %d loops" % (len(blocks), len(loops)))
    tot = {}
    for b in blocks:
        for k, v in b["n"].items():
            tot[k] = tot.get(k, 0) + v
    print("static totals", tot)
    minn = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    for (a, z) in loops:
        n = {}
        for b in blocks[a : z + 1]:
            for k, v in b["n"].items():
                n[k] = n.get(k, 0) + v
        depth = sum(1 for (a2, z2) in loops if a2 <= a and z2 >= z) - 1
        if sum(n.values()) >= minn:
            print("%sloop %s..%s (asm lines %d-%d, %d blocks): valu %d salu %d lds %d vmem %d br %d wait %d" % ("  " * depth, blocks[a]["label"], blocks[z]["label"], blocks[a]["line"] + 1, blocks[z + 1]["line"] if z + 1 < len(blocks) else end, z - a + 1, n.get("valu", 0), n.get("salu", 0), n.get("lds", 0), n.get("vmem", 0), n.get("br", 0), n.get("wait", 0)))


if __name__ == "__main__":
    main()
