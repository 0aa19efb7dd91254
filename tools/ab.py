"""A/B of engine builds: best-of-N search time per instance for every library given.
usage: python tools/ab.py "lib1.so lib2.so" "inst1 inst2 synth" [repeats]   (libs relative to stcsp-solver_amd/csrc;
each (lib, instance) pair runs in its own process: STCSP_HIP_LIB is read once per process)"""
import importlib, os, subprocess, sys
if sys.argv[1] == "--one":
    sys.path.insert(0, '.')
    st = importlib.import_module("stcsp-solver_amd")
    name, reps = sys.argv[2], int(sys.argv[3])
    if name == "synth":
        m = st.Model(text=st.instances.synthetic(64, 32, 602, 6, 20261003))
        e = st.Engine(m, time_limit_s=2.0, flags=st.F_NO_EXPORT | st.F_PROFILE)
        e.solve()  # the first time-boxed solve grows the frontier arena to its working size
        c = e.solve().counters
        print(f"synth64x32: {c.search_nodes / c.seconds_search / 1e6:8.2f} M nodes/s  rounds {c.levels}  kernel {c.seconds_expand_kernel:.3f} s of {c.seconds_search:.3f}", flush=True)
    else:
        m = st.Model.from_name(name)
        e = st.Engine(m, flags=st.F_NO_EXPORT | st.F_PROFILE)
        best = 1e9
        for _ in range(reps):
            e.solve(); c = e.counters(); best = min(best, c.seconds_search)
        print(f"{name}: best {best*1e3:9.3f} ms  {c.search_nodes / best / 1e6:8.2f} M nodes/s  rounds {c.levels}  kernel {c.seconds_expand_kernel*1e3:.3f} ms  nodes {c.search_nodes}", flush=True)
else:
    libs, insts = sys.argv[1].split(), sys.argv[2].split()
    reps = sys.argv[3] if len(sys.argv) > 3 else "6"
    for inst in insts:
        for lib in libs:
            print(f"{lib:28s} ", end="", flush=True)
            subprocess.run([sys.executable, __file__, "--one", inst, reps], env=dict(os.environ, STCSP_HIP_LIB=f"stcsp-solver_amd/csrc/{lib}"), check=False)
