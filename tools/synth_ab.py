"""A/B of two engine builds on the synthetic 64x32 time-boxed run (STCSP_HIP_LIB selects the build)."""
import importlib, os, sys, time
sys.path.insert(0, '.')
st = importlib.import_module("stcsp-solver_amd")
box = float(sys.argv[1]) if len(sys.argv) > 1 else 5.0
for (n, d, m, s, seed, batch) in [(32, 8, 167, 4, 7, 0), (64, 32, 602, 6, 20261003, 0)]:
    mod = st.Model(text=st.instances.synthetic(n, d, m, s, seed))
    e = st.Engine(mod, time_limit_s=box, batch_nodes=batch, flags=st.F_NO_EXPORT)
    r = e.solve(); c = r.counters
    print(os.environ.get("STCSP_HIP_LIB", "default"), (n, d, m, s), f"trunc {r.truncated} nodes {c.search_nodes} fails {c.fails} search {c.seconds_search:.3f}s -> {c.search_nodes/c.seconds_search/1e6:.2f} M nodes/s rounds {c.levels}", flush=True)
    e.close()
