import sys; sys.path.insert(0,'/root/repo')
from peritext_amd import workloads
from peritext_amd.engine import Engine
c = workloads.gen_config("config4")
e = Engine(0)
for cap in (2048, 1536, 1408, 2048, 1536):
    db, info = e.generate(c["replicas"], c["ops_per_log"], c["mix"], c["mark_types"], 32768, 4242, list_cap=cap)
    print("list_cap", cap, "kernel_ms", round(info["kernel_ms"],1), "G ops/s", round(32768*3*4096/info["kernel_ms"]/1e6,3), flush=True)
    e.free_batch(db)
e.close()
