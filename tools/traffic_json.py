#!/usr/bin/env python3
"""Turn the two PMC passes of tools/pmc_traffic.sh into profiles/r06_hbm_traffic.json (what bench.py reports as roofline.traffic).
FETCH_SIZE / WRITE_SIZE are in KB; the read side is calibrated on ptx_calib_stream_kernel, whose byte count is known."""
import hashlib
import json
import os
import re
import sys

out = sys.argv[1]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernel_source_sha16():
    """Identity of the build the counters were taken on: sha256 over the kernel sources and the ABI header (bench.py recomputes it and refuses a traffic
    file of another build)."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "peritext_amd", "csrc")
    for f in sorted(os.listdir(d)) + ["../../include/peritext_hip.h"]:
        with open(os.path.join(d, f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def per_launch(path, kernel, counter, launches):
    """Counter sum of a kernel over the run / the launches tools/traffic_run.py made (a merge whose batch is split in two
    dispatches — a few logs with a larger LDS window — counts as ONE launch)."""
    for line in open(path):
        if line.startswith(kernel) and (" " + counter + " ") in line:
            m = re.search(r"sum=([0-9.e+]+)", line)
            return float(m.group(1)) / launches
    return None


run = None
for line in open(out + "/fetch.log"):
    if line.startswith("TRAFFIC_RUN "):
        run = json.loads(line[len("TRAFFIC_RUN "):])
f_merge = per_launch(out + "/fetch.txt", "ptx_merge_kernel", "FETCH_SIZE", 3)
f_calib = per_launch(out + "/fetch.txt", "ptx_calib_stream_kernel", "FETCH_SIZE", 2)
w_merge = per_launch(out + "/write.txt", "ptx_merge_kernel", "WRITE_SIZE", 3)
factor = run["calib_known_bytes"] / (f_calib * 1024.0)  # true bytes per counted byte on the read side
fetch_calibrated = f_merge * 1024.0 * factor
write = w_merge * 1024.0
# the exact count: read requests of the L2 to the fabric by size
sizes = {}
try:
    for k, b in (("32B", 32), ("64B", 64), ("128B", 128)):
        sizes[k] = per_launch(out + "/size.txt", "ptx_merge_kernel", "TCC_EA0_RDREQ_%s_sum" % k, 3)
    fetch = sum(sizes[k] * b for k, b in (("32B", 32), ("64B", 64), ("128B", 128)))
    calib_exact = sum(per_launch(out + "/size.txt", "ptx_calib_stream_kernel", "TCC_EA0_RDREQ_%s_sum" % k, 2) * b for k, b in (("32B", 32), ("64B", 64), ("128B", 128)))
except (OSError, TypeError):
    sizes, fetch, calib_exact = None, fetch_calibrated, None
print(json.dumps({
    "kernel_source_sha16": kernel_source_sha16(), "launch": run.get("launch"),
    "n_logs": run["n_logs"], "rows": run["rows"], "n_changes": run["n_changes"],
    "fetch_size_kb_per_launch": f_merge, "write_size_kb_per_launch": w_merge,
    "calibration": {"kernel": "ptx_calib_stream_kernel", "known_bytes": run["calib_known_bytes"], "fetch_size_kb": f_calib, "true_bytes_per_counted_byte": factor},
    "read_requests_per_launch": sizes, "fetch_bytes_from_fetch_size_calibrated": fetch_calibrated,
    "calib_stream_bytes_from_request_sizes": calib_exact,
    "fetch_bytes_per_launch": fetch, "write_bytes_per_launch": write, "hbm_bytes_per_launch": fetch + write,
    "source": "rocprofv3 --pmc, separate passes, --kernel-trace only, over tools/traffic_run.py: reads = TCC_EA0_RDREQ_{32B,64B,128B}_sum x their sizes "
              "(cross-check: FETCH_SIZE calibrated on a known byte count, and the same request counters on that calibration stream); writes = WRITE_SIZE. "
              "These are requests of the L2 to the fabric: reads served by the Infinity Cache are included",
}))
