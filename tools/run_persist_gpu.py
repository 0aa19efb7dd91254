"""Persistent-mode check: parity + timing for a list of instances (no budgets => k_persist)."""
import importlib, json, sys, time
sys.path.insert(0, '.')
st = importlib.import_module("stcsp-solver_amd")
gold = json.load(open('tests/golden/reference_golden.json'))
for n in sys.argv[1:]:
    m = st.Model.from_name(n)
    e = st.Engine(m)
    t = time.time(); r = e.solve(); dt = time.time() - t
    a = e.automaton(r).traverse().renumber()
    g = gold[n]; c = r.counters
    print(f"{n:22s} {'OK ' if a.canonical_sha256()==g['canonical_sha256'] else 'BAD'} search={c.seconds_search*1e3:8.3f} ms launches={c.levels} nodes={c.search_nodes} (ref {g['search']}) fails={c.fails} table={r.n_states} (ref {g['node']})", flush=True)
    t = time.time(); r = e.solve(); c = r.counters
    print(f"{'':22s} 2nd solve search={c.seconds_search*1e3:8.3f} ms launches={c.levels}", flush=True)
    e.close()
