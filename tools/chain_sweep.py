"""Sweep the chain policy (expansions per slot and launch) on one instance."""
import importlib, os, subprocess, sys
if len(sys.argv) > 1 and sys.argv[1] == "--one":
    sys.path.insert(0, '.')
    st = importlib.import_module("stcsp-solver_amd")
    m = st.Model.from_name(sys.argv[2])
    e = st.Engine(m, flags=st.F_NO_EXPORT | st.F_PROFILE)
    best = 1e9
    for _ in range(6):
        e.solve(); c = e.counters(); best = min(best, c.seconds_search)
    print(f"{sys.argv[2]} small={os.environ.get('STCSP_CHAIN_SMALL')} big={os.environ.get('STCSP_CHAIN_BIG')} thresh={os.environ.get('STCSP_CHAIN_THRESH')}: "
          f"best {best*1e3:.3f} ms  rounds {c.levels} launches {c.expand_launches} kernel {c.seconds_expand_kernel*1e3:.3f} ms nodes {c.search_nodes}", flush=True)
else:
    name = sys.argv[1] if len(sys.argv) > 1 else "partialorder_14"
    for small, big, th in [(1, 1, 0), (8, 2, 4096), (8, 1, 4096), (4, 2, 4096), (16, 2, 4096), (8, 2, 16384), (8, 4, 4096), (16, 4, 8192), (32, 2, 2048), (8, 3, 8192)]:
        env = dict(os.environ, STCSP_CHAIN_SMALL=str(small), STCSP_CHAIN_BIG=str(big), STCSP_CHAIN_THRESH=str(th))
        subprocess.run([sys.executable, __file__, "--one", name], env=env, check=False)
