"""Extra fuzzing on the GPU beyond the seeds the test-suite covers: engine vs the reference restatement on random models
(tests/fuzz_models.py). usage: fuzz_more.py <first seed> <count> [prefix_k] [wide]     (wide: domains of 33..128 values, WideGen)"""
import importlib, os, sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
st = importlib.import_module("stcsp-solver_amd")
import ctypes as C  # noqa: E402
from fuzz_models import random_model, random_wide_model  # noqa: E402

first, count = int(sys.argv[1]), int(sys.argv[2])
k = int(sys.argv[3]) if len(sys.argv) > 3 else 2
wide = len(sys.argv) > 4 and sys.argv[4] == "wide"
_lib = C.CDLL(str(st.CSRC.parent.parent / "oracle" / "libstcsp_oracle.so"))
st.bind_engine_api(_lib, "stcsp_oracle")


class RefOracle(st.EngineBase):  # oracle/ref_dfs.cpp (a checker: this tool is test infrastructure, like tests/)
    _prefix = "stcsp_oracle"

    def __init__(self, model, **o):
        super().__init__(_lib, model, **o)


bad = checked = refused = trees = 0
t0 = time.time()
for seed in range(first, first + count):
    text = random_wide_model(seed) if wide else random_model(seed)
    m = st.Model(text=text, prefix_k=k)
    o = RefOracle(m, time_limit_s=2.0 if wide else 0.0); ro = o.solve()
    if ro.truncated:  # (a support search of the restatement over ~100^3 tuples)
        refused += 1
        continue
    ao = o.automaton(ro); ao.traverse(); ao.renumber()
    try:
        e = st.Engine(m)
    except st.StcspError as ex:
        refused += 1
        continue
    re_ = e.solve(); ae = e.automaton(re_); ae.traverse(); ae.renumber()
    ok = ae.canonical() == ao.canonical() and re_.counters.dominance == ro.counters.dominance
    same_tree = True
    if ro.counters.fails == 0 and (not wide or re_.counters.fails == 0):
        same_tree = (re_.n_states, re_.counters.search_nodes) == (ro.n_states, ro.counters.search_nodes)
    checked += 1
    if ok and not same_tree and os.environ.get("STCSP_SPLIT_WIDE") == "2":
        # (tuning mode: every conditional constraint runs as its guarded branches -- GAC per branch is weaker than GAC on the
        # constraint, so a search the reference finishes without a failure may meet failing nodes; the automaton is what counts)
        trees += 1
        print(f"TREE seed {seed}: {re_.counters.search_nodes} nodes / {re_.counters.fails} fails against {ro.counters.search_nodes} / 0", flush=True)
    elif not (ok and same_tree):
        bad += 1
        print(f"MISMATCH seed {seed}\n{text}", flush=True)
    e.close(); o.close()
    if (seed - first) % 200 == 199:
        print(f"... {seed - first + 1} models, {bad} mismatches, {time.time() - t0:.0f} s", flush=True)
print(f"checked {checked}, refused / skipped {refused}, mismatches {bad}" + (f", same automaton but another search tree {trees}" if trees else ""))
sys.exit(1 if bad else 0)
