"""Diagnostic: the wide-domain test models one by one with timings (engine vs oracle/ref_dfs.cpp)."""
import importlib, sys, time, ctypes as C
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
st = importlib.import_module("stcsp-solver_amd")
from conftest import finish
from test_wide_gpu import WIDE
lib = C.CDLL('oracle/libstcsp_oracle.so'); st.bind_engine_api(lib, "stcsp_oracle")
class Ref(st.EngineBase):
    _prefix = "stcsp_oracle"
    def __init__(s, m, **o): super().__init__(lib, m, **o)
for name, text in sorted(WIDE.items()):
    m = st.Model(text=text)
    o = Ref(m); ro = o.solve(); ao, _ = finish(o, ro)
    try:
        e = st.Engine(m); t = time.time(); r = e.solve(); dt = time.time() - t
        t = time.time(); r = e.solve(); dt = time.time() - t
        a, _ = finish(e, r)
        print(name, 'OK' if a.canonical() == ao.canonical() else 'MISMATCH', 'states', a.n_live_states, ao.n_live_states, 'edges', a.n_live_edges, ao.n_live_edges,
              'nodes', r.counters.search_nodes, ro.counters.search_nodes, 'fails', r.counters.fails, ro.counters.fails, 'skipped', r.counters.skipped_revisions, '%.2f ms' % (dt * 1e3), flush=True)
    except Exception as ex:
        print(name, 'ERROR', ex, flush=True)
