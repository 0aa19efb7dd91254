import importlib, json, sys, time
sys.path.insert(0, '.')
st = importlib.import_module("stcsp-solver_amd")
gold = json.load(open('tests/golden/reference_golden.json'))
probes = json.load(open('tests/golden/reference_probes.json'))
names = sys.argv[1:] or st.instances.REFERENCE_EXAMPLES
for n in names:
    m = st.Model.from_name(n)
    t = time.time()
    e = st.Engine(m, time_limit_s=100)
    r = e.solve()
    dt = time.time() - t
    a = e.automaton(r).traverse().renumber()
    g = gold[n]; c = r.counters
    ok = (a.canonical_sha256() == g['canonical_sha256'])
    print(f"{n:24s} {'OK ' if ok else 'BAD'} trunc={r.truncated} {dt:7.2f}s search={c.seconds_search:7.3f}s nodes={c.search_nodes} (ref {g['search']}) fails={c.fails} table={r.n_states} (ref {g['node']}) evals={c.evaluations} wrev={c.wave_revisions} skipped={c.skipped_revisions} levels={c.levels}", flush=True)
    e.close()
for k, p in probes.items():
    if k.startswith('_'): continue
    m = st.Model(text=p['text'])
    e = st.Engine(m)
    r = e.solve()
    a = e.automaton(r).traverse().renumber()
    print("probe", k, [m.n_vars, m.n_constraints, r.counters.dominance, r.n_states, r.counters.fails], p['stats'], a.n_live_states, a.n_live_edges, r.n_constraint_sets, flush=True)
