"""Per-launch profile of k_expand (STCSP_DEBUG=2 STCSP_BURST=1): microseconds and nodes per round."""
import importlib, os, sys
os.environ["STCSP_DEBUG"] = "2"; os.environ["STCSP_BURST"] = "1"
sys.path.insert(0, '.')
st = importlib.import_module("stcsp-solver_amd")
m = st.Model.from_name(sys.argv[1] if len(sys.argv) > 1 else "partialorder_14")
e = st.Engine(m, flags=st.F_NO_EXPORT | st.F_PROFILE)
os.environ["STCSP_DEBUG"] = "0"
e.solve()
print("---- second solve", file=sys.stderr)
os.environ["STCSP_DEBUG"] = "2"
e.solve()
