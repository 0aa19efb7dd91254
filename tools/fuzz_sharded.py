"""Random models (tests/fuzz_models.py, narrow and wide) through the NATIVE sharded loop -- stcsp_engine_solve_sharded with 2 or 3
HIP shards on one GPU over the in-process transport -- against oracle/ref_dfs.cpp. usage: fuzz_sharded.py <first seed> <count> [world] [wide]"""
import importlib, sys, time, ctypes as C
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
st = importlib.import_module("stcsp-solver_amd")
from fuzz_models import random_model, random_wide_model  # noqa: E402

first, count = int(sys.argv[1]), int(sys.argv[2])
world = int(sys.argv[3]) if len(sys.argv) > 3 else 2
wide = len(sys.argv) > 4 and sys.argv[4] == "wide"
_lib = C.CDLL(str(st.CSRC.parent.parent / "oracle" / "libstcsp_oracle.so"))
st.bind_engine_api(_lib, "stcsp_oracle")


class RefOracle(st.EngineBase):  # the checker (test infrastructure)
    _prefix = "stcsp_oracle"

    def __init__(self, model, **o):
        super().__init__(_lib, model, **o)


bad = checked = skipped = 0
t0 = time.time()
for seed in range(first, first + count):
    text = random_wide_model(seed) if wide else random_model(seed)
    m = st.Model(text=text)
    o = RefOracle(m, time_limit_s=2.0); ro = o.solve()
    if ro.truncated:
        skipped += 1
        continue
    ao = o.automaton(ro).traverse().renumber()
    try:
        engines = [st.Engine(m, rank=r, world=world) for r in range(world)]
    except st.StcspError:
        skipped += 1
        continue
    g = st.LocalGroup(world)
    g.solve(engines, budget_rounds=1, share_per_rank=2)
    results = [e.export() for e in engines]
    h, merged = st.merge_shards(results)
    a = st.Automaton(m, merged).traverse().renumber()
    ok = a.canonical() == ao.canonical() and merged.counters.dominance == ro.counters.dominance
    checked += 1
    if not ok:
        bad += 1
        print(f"MISMATCH seed {seed}\n{text}", flush=True)
    for e in engines:
        e.close()
    g.close(); o.close()
    if (seed - first) % 100 == 99:
        print(f"... {seed - first + 1} models, {bad} mismatches, {time.time() - t0:.0f} s", flush=True)
print(f"checked {checked}, skipped {skipped}, mismatches {bad}")
sys.exit(1 if bad else 0)
