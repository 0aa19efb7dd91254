"""A/B of one environment switch on the current library: best-of-N search time per instance with and without it.
usage: python tools/ab_env.py VAR=value "inst1 inst2" [repeats]"""
import os, subprocess, sys
var, insts = sys.argv[1], sys.argv[2].split()
reps = sys.argv[3] if len(sys.argv) > 3 else "6"
k, v = var.split("=")
for inst in insts:
    for env in ({}, {k: v}):
        print(f"{(k + '=' + v) if env else 'default':28s} ", end="", flush=True)
        subprocess.run([sys.executable, os.path.join(os.path.dirname(__file__), "ab.py"), "--one", inst, reps], env=dict(os.environ, **env), check=False)
