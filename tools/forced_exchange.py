"""Superstep cost on one GPU: a group of ONE rank with STCSP_FORCE_CANDIDATES=1 sends every leaf through the exchange and k_commit,
like a shard of a multi-GPU run does with the leaves it does not own. Native loop over the RCCL transport (libstcsp_rccl.so),
no torch. usage: python tools/forced_exchange.py [workload] [repeats]"""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
st = importlib.import_module("stcsp-solver_amd")
name = sys.argv[1] if len(sys.argv) > 1 else "partialorder_14"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
m = st.Model.from_name(name)
t = st.RcclTransport(st.rccl_unique_id(), 0, 1, 0)
for forced in ("0", "1"):
    os.environ["STCSP_FORCE_CANDIDATES"] = forced
    e = st.Engine(m, rank=0, world=1, flags=st.F_STEPPED | st.F_NO_EXPORT)
    best = None
    for _ in range(reps):
        t0 = time.perf_counter()
        s = st.solve_sharded_native(e, t.ptr)
        dt = time.perf_counter() - t0
        if best is None or dt < best[0]:
            best = (dt, s)
    c = e.counters()
    dt, s = best
    print(f"{name} forced={forced}: {dt*1e3:8.3f} ms  {c.search_nodes/dt/1e6:7.1f} M nodes/s  supersteps {s['supersteps']}  rounds {c.levels}  "
          f"candidates {s['candidates_sent']}  per superstep {dt*1e3/s['supersteps']:.3f} ms = expand {s['seconds_expand']*1e3/s['supersteps']:.3f} + pack {s['seconds_pack']*1e3/s['supersteps']:.3f} "
          f"+ collectives {s['seconds_collectives']*1e3/s['supersteps']:.3f} + commit {s['seconds_commit']*1e3/s['supersteps']:.3f}", flush=True)
    e.close()
