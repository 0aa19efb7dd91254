"""Repeated time-boxed sharded solves of the synthetic 64 x 32 instance with 2 HIP-engine shards on ONE GPU (the setup that
exposed the late-workgroup race fixed by Plan::gate): prints per run the ranks' search nodes or the device error."""
import json, os, socket, subprocess, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
def free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p
def launch(world, name, env):
    port = free_port()
    procs = [subprocess.Popen([sys.executable, f"{REPO}/tests/_sharded_worker.py", str(r), str(world), str(port), name, "hip", "/tmp/merged.json"],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=dict(os.environ, **env)) for r in range(world)]
    logs = [p.communicate(timeout=200)[0] for p in procs]
    bad = [l for l in logs if "device reported" in l]
    if bad:
        import re
        print("  FAIL:", re.findall(r"device reported[^\n]*", bad[0])[0][:400])
        print("\n".join([l for l in bad[0].splitlines() if "act " in l or "planner history" in l or "cursor check" in l or "host:" in l][-75:]))
    else:
        rc = [p.returncode for p in procs]
        print("  ok rc", rc, (json.load(open("/tmp/merged.json"))["rank_nodes"] if all(r == 0 for r in rc) else [l[-300:] for l in logs]))
for label, env in [("default", {})]:
    for i in range(6):
        print(label, i, flush=True)
        launch(3 if "world3" in label else 2, "synth:64,32,602,6,20261003", dict(env, STCSP_TEST_TIME_LIMIT="1.0"))
