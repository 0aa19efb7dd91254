"""A/B of STCSP_FRESH_INIT (engine.hip fresh_init): search time, nodes and parity against the golden automaton.
usage: fresh_sweep.py <instances...>"""
import importlib, json, os, sys
sys.path.insert(0, '.')
st = importlib.import_module("stcsp-solver_amd")
golden = json.load(open("tests/golden/reference_golden.json"))
for name in sys.argv[1:]:
    m = st.Model.from_name(name)
    for mode in ["0", "1"]:
        os.environ["STCSP_FRESH_INIT"] = mode
        e = st.Engine(m)
        r = e.solve()
        a = e.automaton(r).traverse().renumber()
        g = golden[name]
        ok = a.canonical_sha256() == g["canonical_sha256"] and r.counters.dominance == g["dom"] and (r.counters.search_nodes == g["search"] or g["fail"] > 0)
        e.close()
        e = st.Engine(m, flags=st.F_NO_EXPORT)
        best = 1e9
        for _ in range(6):
            c = e.solve().counters
            best = min(best, c.seconds_search)
        print(f"{name:24s} fresh_init={mode} {'ok ' if ok else 'MISMATCH'} search {best*1e3:8.3f} ms nodes {c.search_nodes} rounds {c.levels} revisions/node {c.revisions/c.search_nodes:.1f} wave revs/node {c.wave_revisions/c.search_nodes:.2f}", flush=True)
        e.close()
