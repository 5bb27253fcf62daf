#!/usr/bin/env python3
"""Register / occupancy summary of the kernels of peritext_hip.hip as hipcc reports them (build container, no GPU needed).
    python tools/kres.py [-DX=1 ...]      -> one line per kernel: VGPRs, SGPRs, waves per SIMD, spills, code bytes"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as G  # noqa: E402


def main():
    out = "/tmp/kres_%d.so" % os.getpid()
    cmd = [G.HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-Rpass-analysis=kernel-resource-usage"] + G.HIP_FLAGS + sys.argv[1:] + [
        "-o", out, os.path.join(G.CSRC, "peritext_hip.hip")]
    p = subprocess.run(cmd, capture_output=True, text=True)
    if p.returncode:
        print(p.stderr[-3000:])
        sys.exit(1)
    cur, rows = None, {}
    for line in p.stderr.splitlines():
        m = re.search(r"remark:\s+(Function Name|TotalSGPRs|VGPRs|AGPRs|Occupancy \[waves/SIMD\]|VGPRs Spill|SGPRs Spill|ScratchSize \[bytes/lane\]):\s*(\S+)", line)
        if not m:
            continue
        if m.group(1) == "Function Name":
            cur = m.group(2)
            rows[cur] = {}
        elif cur:
            rows[cur][m.group(1)] = m.group(2)
    for k, r in rows.items():
        if not k.startswith("ptx_"):
            continue
        print("%-26s VGPRs %3s  SGPRs %3s  waves/SIMD %s  spill v%s s%s scratch %s" % (k, r.get("VGPRs"), r.get("TotalSGPRs"), r.get("Occupancy [waves/SIMD]"), r.get("VGPRs Spill"),
                                                                                   r.get("SGPRs Spill"), r.get("ScratchSize [bytes/lane]")))
    os.remove(out)


if __name__ == "__main__":
    main()
