"""Launch batch size (frontier nodes taken per round) vs throughput on big members of the partialorder family."""
import importlib, sys
sys.path.insert(0, '.')
st = importlib.import_module("stcsp-solver_amd")
for n in [int(a) for a in sys.argv[1:]] or [14, 16, 18]:
    m = st.Model(text=st.instances.partialorder(n))
    for batch in [65536, 131072, 262144, 524288, 1048576]:
        e = st.Engine(m, flags=st.F_NO_EXPORT, batch_nodes=batch)
        best = None
        for _ in range(3):
            c = e.solve().counters
            if best is None or c.seconds_search < best[0]: best = (c.seconds_search, c.search_nodes, c.levels)
        print(f"partialorder_{n} batch {batch}: rounds {best[2]} search {best[0]*1e3:.2f} ms -> {best[1]/best[0]/1e6:.1f} M nodes/s", flush=True)
        e.close()
