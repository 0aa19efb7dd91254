"""Synthetic random-binary stCSP (BASELINE.json config 4) on the GPU: parity at shapes the
CPU oracle finishes, then a time-boxed 64 vars x |D|=32 run next to the CPU port."""
import ctypes as C, importlib, sys, time
sys.path.insert(0, '.')
st = importlib.import_module("stcsp-solver_amd")
olib = C.CDLL('oracle/libstcsp_oracle.so')
st.bind_engine_api(olib, "stcsp_oracle")
class Ref(st.EngineBase):
    _prefix = "stcsp_oracle"
    def __init__(self, model, **o): super().__init__(olib, model, **o)
for (n,d,m,s,seed) in [(8,4,14,2,1),(16,8,95,4,3),(16,8,88,4,4),(16,8,80,4,5)]:
    mod = st.Model(text=st.instances.synthetic(n,d,m,s,seed))
    o = Ref(mod); a = o.automaton(o.solve()).traverse().renumber()
    e = st.Engine(mod); r = e.solve(); ae = e.automaton(r).traverse().renumber()
    print((n,d,m,s,seed), "SAME" if a.canonical()==ae.canonical() else "DIFF", "gpu nodes", r.counters.search_nodes, "fails", r.counters.fails, "%.4fs"%r.counters.seconds_search, flush=True)
box = float(sys.argv[1]) if len(sys.argv) > 1 else 10.0
for (n,d,m,s,seed,batch) in [(32,8,167,4,7,0),(64,32,602,6,20261003,0),(64,32,602,6,20261003,262144)]:
    mod = st.Model(text=st.instances.synthetic(n,d,m,s,seed))
    t=time.time(); e = st.Engine(mod, time_limit_s=box, batch_nodes=batch); tc=time.time()-t
    r = e.solve(); c = r.counters
    print((n,d,m,s,seed,batch), f"create {tc:.2f}s vars {mod.n_vars} cons {mod.n_constraints} trunc {r.truncated} nodes {c.search_nodes} fails {c.fails} leaves {c.leaves} states {r.n_states} search {c.seconds_search:.2f}s -> {c.search_nodes/c.seconds_search/1e6:.2f} M nodes/s levels {c.levels} wrev {c.wave_revisions} sweeps {c.sweeps}", flush=True)
    e.close()
    if batch == 0:
        o = Ref(mod, time_limit_s=box); t=time.time(); ro = o.solve(); dt=time.time()-t
        print("   cpu port:", f"trunc {ro.truncated} nodes {ro.counters.search_nodes} fails {ro.counters.fails} {dt:.2f}s -> {ro.counters.search_nodes/dt/1e3:.1f} k nodes/s", flush=True)
