"""The synthetic 64 x 32 instance (BASELINE config 4), time-boxed: one warm solve (grows the arena), one measured.
Prints one JSON line. usage: python tools/synth_bench.py [seconds]"""
import importlib, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
st = importlib.import_module("stcsp-solver_amd")
box = float(sys.argv[1]) if len(sys.argv) > 1 else 2.0
n, d, m, s, seed = 64, 32, 602, 6, 20261003
mod = st.Model(text=st.instances.synthetic(n, d, m, s, seed))
e = st.Engine(mod, time_limit_s=box, flags=st.F_NO_EXPORT | st.F_PROFILE)
e.solve()
c = e.solve().counters
p = mod.problem.contents
print(json.dumps({"workload": f"synthetic {n}x{d}, {m}+{s} constraints, seed {seed}", "time_box_s": box, "nodes": c.search_nodes, "fails": c.fails,
                  "leaves": c.leaves, "seconds_search": c.seconds_search, "nodes_per_s": c.search_nodes / c.seconds_search,
                  "seconds_expand_kernel": c.seconds_expand_kernel, "launches": c.expand_launches, "bytes_per_node": 2 * p.n_vars * p.prefix_k * 4,
                  "per_node": {"item_revisions": c.revisions / c.search_nodes, "sweeps": c.sweeps / c.search_nodes}}))
