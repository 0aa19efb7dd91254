import importlib, sys
sys.path.insert(0, '.')
st = importlib.import_module("stcsp-solver_amd")
mod = st.Model(text=st.instances.synthetic(64, 32, 602, 6, 20261003))
e = st.Engine(mod, time_limit_s=2.0, flags=st.F_NO_EXPORT)
c = e.solve().counters
n = c.search_nodes
print(f"nodes {n} fails {c.fails} leaves {c.leaves} per node: item revisions {c.revisions/n:.1f} sweeps {c.sweeps/n:.2f} wave revisions {c.wave_revisions/n:.2f} tuple evals {c.evaluations/n:.1f} skipped {c.skipped_revisions/n:.2f}; rounds {c.levels}; {n/c.seconds_search/1e6:.2f} M nodes/s")
