"""Device post-processing (stcsp_engine_postprocess) vs its host twin (postproc.cpp) -- wall times."""
import importlib, sys, time
sys.path.insert(0, '.')
st = importlib.import_module("stcsp-solver_amd")
for name in sys.argv[1:] or ["partialorder_14", "digitinvader9"]:
    m = st.Model.from_name(name)
    e = st.Engine(m)
    r = e.solve()
    for label, kw in (("traverse", {}), ("traverse+a", {"adversarial": 5}), ("traverse+a+z", {"adversarial": 5, "adversarial2": (5, 6)})):
        e.postprocess(**kw)
        t = time.perf_counter(); post = e.postprocess(**kw); dev_ms = (time.perf_counter() - t) * 1e3
        t = time.perf_counter(); a = e.automaton(r); build_ms = (time.perf_counter() - t) * 1e3
        t = time.perf_counter()
        a.traverse()
        if "adversarial" in kw: a.adversarial(5)
        if "adversarial2" in kw: a.adversarial2(5, 6)
        host_ms = (time.perf_counter() - t) * 1e3
        print(f"{name:18s} {label:13s} states {r.n_states} edges {r.n_edges} device {dev_ms:.2f} ms (engine {post.seconds*1e3:.2f} ms, rounds {list(post.rounds)}) "
              f"host passes {host_ms:.2f} ms (+ CSR build {build_ms:.1f} ms)  adver {post.adver1} {post.adver2}", flush=True)
