"""Where the time goes in the sharded (candidate/commit) pipeline, on one GPU with a size-1 RCCL group."""
import importlib, os, sys, time
sys.path.insert(0, '.')
import torch, torch.distributed as dist
st = importlib.import_module("stcsp-solver_amd")
sh = importlib.import_module("stcsp-solver_amd.sharded")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
torch.cuda.set_device(0)
dev = torch.device("cuda:0")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
name = sys.argv[1] if len(sys.argv) > 1 else "partialorder_14"
m = st.Model.from_name(name)
eng = st.Engine(m, flags=st.F_STEPPED)
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    rounds = sh.solve_sharded(eng, 0, 1, dev)
    torch.cuda.synchronize(); print(f"solve_sharded: {rounds} supersteps {(time.perf_counter()-t0)*1e3:.2f} ms")
# instrumented copy of the loop
T = dict(expand=0.0, outbox=0.0, meta=0.0, a2a=0.0, commit=0.0)
csw = eng.candidate_bytes() // 4
eng.begin()
def lap(key, t):
    torch.cuda.synchronize(); T[key] += time.perf_counter() - t
cands = 0
while True:
    t = time.perf_counter(); left = eng.expand_local(); lap("expand", t)
    t = time.perf_counter(); ptr, cnt = eng.outbox(0); lap("outbox", t)
    t = time.perf_counter()
    meta = torch.tensor([cnt, left, eng.sets_count()], dtype=torch.int64, device=dev)
    allm = [torch.empty_like(meta)]; dist.all_gather(allm, meta); allm = [x.tolist() for x in allm]; lap("meta", t)
    t = time.perf_counter()
    send = sh._view(ptr, cnt * csw, dev); recv = torch.empty(cnt * csw, dtype=torch.int32, device=dev)
    dist.all_to_all_single(recv, send, [cnt * csw], [cnt * csw]); lap("a2a", t)
    t = time.perf_counter(); eng.commit(recv.data_ptr() if cnt else 0, cnt); lap("commit", t)
    cands += cnt
    if left == 0 and cnt == 0: break
eng.finish()
print(name, "candidates", cands, "bytes/candidate", csw * 4, {k: round(v * 1e3, 3) for k, v in T.items()}, "ms; total", round(sum(T.values()) * 1e3, 2))
e2 = st.Engine(m, flags=st.F_NO_EXPORT); e2.solve(); e2.solve(); print("fused solve ms", e2.counters().seconds_search * 1e3)
dist.destroy_process_group()
