"""Per-node work counters of one solve."""
import importlib, sys
sys.path.insert(0, '.')
st = importlib.import_module("stcsp-solver_amd")
for name in sys.argv[1:]:
    e = st.Engine(st.Model.from_name(name), flags=st.F_NO_EXPORT)
    e.solve(); c = e.solve().counters; n = c.search_nodes
    print(f"{name}: nodes {n} fails {c.fails} leaves {c.leaves} | per node: item revisions {c.revisions/n:.1f} sweeps {c.sweeps/n:.2f} wavefront revisions {c.wave_revisions/n:.2f} "
          f"tuple evaluations {c.evaluations/n:.0f} skipped {c.skipped_revisions/n:.2f} | rounds {c.levels} search {c.seconds_search*1e3:.2f} ms")
