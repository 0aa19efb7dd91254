"""Throughput over the partialorder family (bigger members than the reference ships)."""
import importlib, sys, time
sys.path.insert(0, '.')
st = importlib.import_module("stcsp-solver_amd")
for n in [10, 12, 14, 15, 16, 17, 18]:
    m = st.Model(text=st.instances.partialorder(n))
    e = st.Engine(m, flags=st.F_NO_EXPORT)
    best = None
    for _ in range(3):
        c = e.solve().counters
        if best is None or c.seconds_search < best[0]: best = (c.seconds_search, c.search_nodes, c.levels, c.leaves)
    r = e.export()
    print(f"partialorder_{n}: nodes {best[1]} leaves {best[3]} states {r.n_states} edges {r.n_edges} rounds {best[2]} search {best[0]*1e3:.2f} ms -> {best[1]/best[0]/1e6:.1f} M nodes/s; export {r.counters.seconds_export*1e3:.1f} ms", flush=True)
    e.close()
