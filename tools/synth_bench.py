"""The synthetic 64 x 32 instance (BASELINE config 4): one warm solve (grows the arena), one measured. Prints one JSON line.
usage: python tools/synth_bench.py [seconds]           time-boxed (throughput: what bench.py reports)
       python tools/synth_bench.py --nodes N           node-bounded (max_search_nodes = N): the profiler passes -- every pass
                                                       expands about the same nodes, and `nodes_all_solves` is what the counters
                                                       of a pass (which cover the warm solve too) are divided by"""
import importlib, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
st = importlib.import_module("stcsp-solver_amd")
bounded = len(sys.argv) > 2 and sys.argv[1] == "--nodes"
box = 0.0 if bounded else (float(sys.argv[1]) if len(sys.argv) > 1 else 2.0)
max_nodes = int(sys.argv[2]) if bounded else 0
n, d, m, s, seed = 64, 32, 602, 6, 20261003
mod = st.Model(text=st.instances.synthetic(n, d, m, s, seed))
e = st.Engine(mod, time_limit_s=box, max_search_nodes=max_nodes, flags=st.F_NO_EXPORT | st.F_PROFILE)
c0 = e.solve().counters
warm_nodes, warm_revs = c0.search_nodes, c0.revisions
c = e.solve().counters
p = mod.problem.contents
print(json.dumps({"workload": f"synthetic {n}x{d}, {m}+{s} constraints, seed {seed}", "time_box_s": box or None, "max_search_nodes": max_nodes or None,
                  "nodes": c.search_nodes, "nodes_all_solves": warm_nodes + c.search_nodes, "fails": c.fails,
                  "leaves": c.leaves, "seconds_search": c.seconds_search, "nodes_per_s": c.search_nodes / c.seconds_search,
                  "seconds_expand_kernel": c.seconds_expand_kernel, "launches": c.expand_launches, "bytes_per_node": 2 * p.n_vars * p.prefix_k * 4,
                  "per_node": {"item_revisions": c.revisions / c.search_nodes, "sweeps": c.sweeps / c.search_nodes,
                               "item_revisions_all_solves": (warm_revs + c.revisions) / (warm_nodes + c.search_nodes)}}))
