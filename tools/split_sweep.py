"""A/B of STCSP_SPLIT_WIDE (cset.cpp split_wide): search time, nodes and parity against the golden automaton per mode.
usage: split_sweep.py "<modes>" <instances...>"""
import importlib, json, os, sys
sys.path.insert(0, '.')
st = importlib.import_module("stcsp-solver_amd")
golden = json.load(open("tests/golden/reference_golden.json"))
modes = sys.argv[1].split(",")
for name in sys.argv[2:]:
    m = st.Model.from_name(name)
    for mode in modes:
        os.environ["STCSP_SPLIT_WIDE"] = mode
        e = st.Engine(m)
        r = e.solve()
        a = e.automaton(r).traverse().renumber()
        ok = a.canonical_sha256() == golden[name]["canonical_sha256"] and r.counters.dominance == golden[name]["dom"]
        e.close()
        e = st.Engine(m, flags=st.F_NO_EXPORT)
        best = 1e9
        for _ in range(6):
            c = e.solve().counters
            best = min(best, c.seconds_search)
        print(f"{name:24s} split={mode} {'ok ' if ok else 'MISMATCH'} search {best*1e3:8.3f} ms nodes {c.search_nodes} fails {c.fails} rounds {c.levels} "
              f"wave revs/node {c.wave_revisions/c.search_nodes:.2f} skipped/node {c.skipped_revisions/c.search_nodes:.2f}", flush=True)
        e.close()
