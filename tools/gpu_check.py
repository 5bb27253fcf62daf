#!/usr/bin/env python3
"""Parity spot-check of an experimental library build / kernel variant against the committed golden fixtures (GPU box).
    python tools/gpu_check.py --lib peritext_amd/lib/exp_u2.so --variants 0,5,6 --threads 256"""
import argparse
import json
import os
import sys
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers as H  # noqa: E402
from peritext_amd.engine import Engine  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lib", default=None)
    ap.add_argument("--variants", default="0")
    ap.add_argument("--threads", default="256")
    args = ap.parse_args()
    lib = args.lib if (args.lib is None or os.path.isabs(args.lib)) else os.path.join(ROOT, args.lib)
    names = ["ptxgen_mini.json", "ptxgen_config2.json", "ptxgen_config3_512.json", "ptxgen_config4_600.json", "ptxgen_rich_700.json", "ptxgen_rich_2600.json"]
    for var in args.variants.split(","):
        for t in args.threads.split(","):
            eng = Engine(0, lib_path=lib)
            eng.set_launch_shape(int(t), 0)
            for name in names:
                gen = json.load(open(os.path.join(H.GOLDEN, name)))
                try:
                    _, res = H.check_generated(gen, eng.apply_materialize)
                    print("ok   lib=%s variant=%s threads=%s %s" % (os.path.basename(lib or "default"), var, t, name), flush=True)
                except Exception as e:  # noqa: BLE001
                    print("FAIL lib=%s variant=%s threads=%s %s: %s" % (os.path.basename(lib or "default"), var, t, name, str(e).splitlines()[0][:200]), flush=True)
            eng.close()


if __name__ == "__main__":
    main()
