import sys,os,json
ROOT=os.environ.get("GRAFT_REPO_ROOT","/root/repo"); sys.path.insert(0,ROOT)
from peritext_amd import abi, workloads
from peritext_amd.engine import Engine
c=workloads.gen_config("config4")
for rnd in range(3):
    for lib in (None, os.path.join(ROOT,"peritext_amd/lib/exp_genold.so"), os.path.join(ROOT,"peritext_amd/lib/exp_nor8.so")):
        e=Engine(0, flags=abi.FLAG_NO_ELEM_RANK, lib_path=lib)
        h,info=e.generate(c["replicas"],c["ops_per_log"],c["mix"],c["mark_types"],65536,2024,list_cap=1536)
        print(os.path.basename(lib or "product"), round(info["kernel_ms"],1), flush=True)
        e.free_batch(h); e.close()
