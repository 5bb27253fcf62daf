"""The synthetic workloads of BASELINE.json / SURVEY.md §8(d) ("PTXGEN" configs #2..#5 + two small ones used by the tests) as
parameters of the on-device generator (Engine.generate / ptx_generate).  The same table lives in oracle/ptxgen.js CONFIGS (the
oracle side); tests/test_emu_generate.py checks that both sides produce the same documents from it."""
from . import abi

# name -> (replicas, ops per log, mix % [insert, delete, addMark, removeMark], mark types in the config's order)
CONFIGS = {
    "config2": (1, 256, [70, 30, 0, 0], []),
    "config3": (1, 1024, [40, 20, 25, 15], ["strong", "em"]),
    "config4": (3, 4096, [25, 25, 25, 25], ["strong", "em", "link", "comment"]),
    "config5": (1, 8192, [20, 50, 20, 10], ["link", "comment"]),
    "rich": (3, 1024, [55, 10, 20, 15], ["strong", "em", "link", "comment"]),
    # the same mix at the headline's log length: documents that HOLD text (about two thousand visible characters, hundreds of spans) — bench.py's extras leg `rich4k`
    "rich4k": (3, 4096, [55, 10, 20, 15], ["strong", "em", "link", "comment"]),
    "mini": (3, 96, [25, 25, 25, 25], ["strong", "em", "link", "comment"]),
}


def gen_config(name, ops=None, replicas=None):
    r, n, mix, marks = CONFIGS[name]
    return {"replicas": replicas or r, "ops_per_log": ops or n, "mix": list(mix), "mark_types": [abi.MARK_NAMES.index(m) for m in marks]}
