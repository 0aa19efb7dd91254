"""End-to-end wall time of the command line on generated instances (process start -> parse -> engine
-> device passes -> solutions.dot), run from a scratch directory under gpurun_out/."""
import importlib, os, subprocess, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
inst = importlib.import_module("stcsp-solver_amd.instances")
work = os.path.join(REPO, "gpurun_out", "cli"); os.makedirs(work, exist_ok=True); os.chdir(work)
cli = os.path.join(REPO, "stcsp-solver_amd", "csrc", "stcsp")
for name in sys.argv[1:] or ["partialorder_14", "digitinvader9", "juggling_b5_f6"]:
    open(f"{name}.csp", "w").write(inst.by_name(name))
    for flags in ([], ["-s"], ["-s", f"--binary={name}.bin"], ["-a"]):
        best = 1e9
        for _ in range(3):
            t = time.perf_counter(); r = subprocess.run([cli, *flags, f"{name}.csp"], capture_output=True, text=True); best = min(best, time.perf_counter() - t)
        sizes = {f: os.path.getsize(f) for f in ("solutions.dot", f"{name}.bin") if os.path.exists(f)}
        print(f"{name:16s} {' '.join(flags):28s} wall {best:.3f} s  stdout: {r.stdout.strip()!r}  {sizes}", flush=True)
        for f in sizes: os.remove(f)
