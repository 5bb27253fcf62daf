#!/usr/bin/env python3
"""Same-box, same-call A/B of the generator kernel (ptx_generate on 65 536 config-4 documents, kernel ms) between the product and other builds of the library:
    python tools/gen_ab.py peritext_amd/lib/exp_<name>.so [...]        (GPU box; tools/build_rev.sh makes a revision's build)"""
import os
import sys

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from peritext_amd import abi, workloads  # noqa: E402
from peritext_amd.engine import Engine  # noqa: E402

c = workloads.gen_config("config4")
libs = [None] + [p if os.path.isabs(p) else os.path.join(ROOT, p) for p in sys.argv[1:]]
for rnd in range(3):
    for lib in libs:
        e = Engine(0, flags=abi.FLAG_NO_ELEM_RANK, lib_path=lib)
        h, info = e.generate(c["replicas"], c["ops_per_log"], c["mix"], c["mark_types"], 65536, 2024, list_cap=1536)
        print(os.path.basename(lib or "product"), round(info["kernel_ms"], 1), flush=True)
        e.free_batch(h)
        e.close()
