"""Markdown table of all 26 reference instances: parity and search time (best of 3 solves)."""
import importlib, json, sys
sys.path.insert(0, '.')
st = importlib.import_module("stcsp-solver_amd")
gold = json.load(open('tests/golden/reference_golden.json'))
print("| instance | search nodes | states (table) | edges | rounds | search ms | M nodes/s | canonical sha256 |")
print("|---|---|---|---|---|---|---|---|")
for n in st.instances.REFERENCE_EXAMPLES:
    m = st.Model.from_name(n)
    e = st.Engine(m)
    best = None
    for _ in range(3):
        r = e.solve(); c = r.counters
        if best is None or c.seconds_search < best: best = c.seconds_search
    a = e.automaton(r).import_flags(e.postprocess()).renumber()
    g = gold[n]
    ok = a.canonical_sha256() == g['canonical_sha256']
    print(f"| {n} | {c.search_nodes:,} | {a.n_live_states:,} ({r.n_states:,}) | {a.n_live_edges:,} | {c.levels} | {best*1e3:.2f} | {c.search_nodes/best/1e6:.1f} | {'= reference' if ok else 'DIFFERS'} |", flush=True)
    e.close()
