"""Diagnostic: the stepping interface (STCSP_F_STEPPED, one shard) driven without torch: search time with and without the
streaming export (is a slow-down of the stepped bench the driver's or the engine's?)"""
import importlib, os, sys
sys.path.insert(0, '.')
st = importlib.import_module("stcsp-solver_amd")
m = st.Model.from_name(sys.argv[1] if len(sys.argv) > 1 else "partialorder_14")
ENVS = [{}, {"STCSP_STREAM_EXPORT": "0"}, {"STCSP_STREAM_POLL": "0"},
        {"STCSP_BURST": "32"}, {"STCSP_BATCH": "262144"},
        {"STCSP_BATCH": "262144", "STCSP_BURST": "32"}]
for env in ENVS:
    for k in ("STCSP_STREAM_EXPORT", "STCSP_STREAM_POLL", "STCSP_BURST", "STCSP_BATCH"):
        os.environ.pop(k, None)
    os.environ.update(env)
    e = st.Engine(m, flags=st.F_STEPPED)
    best = 1e9
    for _ in range(8):
        e.begin()
        while True:
            left = e.expand_local()
            ptr, cnt = e.outbox(0)
            e.commit(0, 0)
            if left == 0 and cnt == 0:
                break
        e.finish()
        r = e.export()
        c = e.counters()
        best = min(best, c.seconds_search)
    print(env or "streaming", "search ms %.3f export ms %.3f edges %d" % (best * 1e3, r.counters.seconds_export * 1e3, r.n_edges), flush=True)
    e.close()
