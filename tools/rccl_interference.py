"""Diagnostic: does an initialised RCCL communicator in the process change the cost of the streaming export? (search time of
partialorder_14 before / after init_process_group, with and without streaming)"""
import importlib, os, sys, time
sys.path.insert(0, '.')
import torch, torch.distributed as dist
st = importlib.import_module("stcsp-solver_amd")
torch.cuda.set_device(0)
dev = torch.device("cuda:0")
m = st.Model.from_name("partialorder_14")
def run(tag, flags=0):
    e = st.Engine(m, flags=flags)
    best = 1e9
    for _ in range(8):
        r = e.solve(); best = min(best, r.counters.seconds_search)
    print(tag, "search ms %.3f export %.3f" % (best * 1e3, r.counters.seconds_export * 1e3), flush=True)
    e.close()
run("no rccl, streaming")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
t = torch.ones(4, device=dev); dist.all_reduce(t); torch.cuda.synchronize()
run("rccl initialised, streaming")
os.environ["STCSP_STREAM_EXPORT"] = "0"
run("rccl initialised, no streaming")
dist.destroy_process_group()
