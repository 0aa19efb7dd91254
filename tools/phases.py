"""Diagnostic: per-phase cycle shares of k_expand (needs csrc/libstcsp_hip_phases.so, built with -DSTCSP_PHASES)."""
import ctypes as C, importlib, os, sys
sys.path.insert(0, '.')
st = importlib.import_module("stcsp-solver_amd")
lib = C.CDLL(str(st.CSRC / os.environ.get("STCSP_PHASES_LIB", "libstcsp_hip_phases.so"))); st.bind_engine_api(lib)
class E(st.EngineBase):
    def __init__(self, m, **o): super().__init__(lib, m, **o)
for name in sys.argv[1:] or ["partialorder_14"]:
    m = st.Model(text=st.instances.synthetic(64, 32, 602, 6, 20261003)) if name == "synth" else st.Model.from_name(name)
    e = E(m, flags=st.F_NO_EXPORT, time_limit_s=2.0 if name == "synth" else 0.0); e.solve(); e.solve()
    c = e.counters(); print(name, "nodes", c.search_nodes, "search s", c.seconds_search)
