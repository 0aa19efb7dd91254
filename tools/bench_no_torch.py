"""The headline workload through the C-ABI from a process that never loads PyTorch: the engine then runs on the HIP
runtime a C++ host linked against /opt/rocm gets (what the INTEGRATION.md binding and the `stcsp` CLI use), not on the
one PyTorch bundles. bench.py starts this as a child process outside its timed region and puts the line into its own
(`config.without_pytorch_runtime`). usage: python tools/bench_no_torch.py [workload] [steps] [warmup]"""
import ctypes as C
import importlib
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
assert "torch" not in sys.modules
st = importlib.import_module("stcsp-solver_amd")
name = sys.argv[1] if len(sys.argv) > 1 else "partialorder_14"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
warmup = int(sys.argv[3]) if len(sys.argv) > 3 else 5
m = st.Model.from_name(name)
e = st.Engine(m, flags=st.F_PROFILE)
rtv = C.c_int(0)
try:
    C.CDLL("libamdhip64.so.7").hipRuntimeGetVersion(C.byref(rtv))  # (the copy the engine library already loaded)
except OSError:
    pass
for _ in range(warmup):
    e.solve()
t0 = time.perf_counter()
nodes = 0
s_search = s_export = 0.0
for _ in range(steps):
    r = e.solve()
    nodes += r.counters.search_nodes
    s_search += r.counters.seconds_search
    s_export += r.counters.seconds_export
dt = time.perf_counter() - t0
a = e.automaton(r).traverse().renumber()
gold = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "reference_golden.json")))
assert "torch" not in sys.modules
print(json.dumps({"workload": name, "steps": steps, "warmup": warmup, "nodes_per_s": nodes / dt, "ms_per_step": dt / steps * 1e3,
                  "search_ms": s_search / steps * 1e3, "export_ms": s_export / steps * 1e3, "hip_runtime_version": rtv.value,
                  "canonical_sha256": a.canonical_sha256(), "parity_ok": a.canonical_sha256() == gold.get(name, {}).get("canonical_sha256"),
                  "pytorch_loaded": False}))
