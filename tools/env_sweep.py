"""Diagnostic: search time of one instance under several settings of the engine's tuning variables.
usage: env_sweep.py <instance> VAR=a,b,c [VAR2=x,y ...]   (cartesian product; best of 5 solves each)"""
import importlib, itertools, os, sys
sys.path.insert(0, '.')
st = importlib.import_module("stcsp-solver_amd")
name = sys.argv[1]
axes = [(a.split("=")[0], a.split("=")[1].split(",")) for a in sys.argv[2:]]
m = st.Model.from_name(name)
for combo in itertools.product(*[v for _, v in axes]):
    for (k, _), v in zip(axes, combo):
        if v == "-":
            os.environ.pop(k, None)
        else:
            os.environ[k] = v
    e = st.Engine(m, flags=st.F_NO_EXPORT)
    best, rounds = 1e9, 0
    for _ in range(5):
        r = e.solve()
        c = r.counters
        best = min(best, c.seconds_search)
        rounds = c.levels
    print(name, dict(zip([k for k, _ in axes], combo)), "search ms %.3f" % (best * 1e3), "nodes", c.search_nodes, "rounds", rounds, flush=True)
    e.close()
