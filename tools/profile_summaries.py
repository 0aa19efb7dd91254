"""Turn the raw output of tools/profile_final.sh (gpurun_out/prof_final/) into the summaries committed
under profiles/: kernel stats csv, instruction-mix / wave-cycle json, HBM traffic json."""
import glob, json, os, shutil, subprocess, sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RAW = os.path.join(REPO, "gpurun_out", "prof_final")
OUT = os.path.join(REPO, "profiles")
TAG = sys.argv[1] if len(sys.argv) > 1 else "r01_chain_p14"
NOTE = sys.argv[2] if len(sys.argv) > 2 else "round-1 final engine"


def newest(pattern):
    files = sorted(glob.glob(pattern, recursive=True), key=os.path.getmtime)
    return files[-1]


def summ(p):
    return json.loads(subprocess.check_output([sys.executable, os.path.join(REPO, "tools", "pmc_summary.py"), os.path.join(RAW, f"pmc_{p}")]))


shutil.copy(newest(os.path.join(RAW, "stats", "**", "*_kernel_stats.csv")), os.path.join(OUT, f"{TAG}_kernel_stats.csv"))
f, w, i, b = summ("FETCH_SIZE"), summ("WRITE_SIZE"), summ("SQ_INSTS_VALU"), summ("SQ_BUSY_CYCLES")
bench = json.load(open(os.path.join(RAW, "pmc_FETCH_SIZE.json")))
nodes, leaves = bench["config"]["nodes_per_step"], bench["config"]["leaves_per_step"]
src = f"tools/profile_final.sh ({NOTE})"
wc = i["SQ_WAVE_CYCLES"]["sum"]
json.dump({"_source": src + ": rocprofv3 --pmc <list> --kernel-trace -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline (one cold solve of "
                      "partialorder_14); sums over the k_expand dispatches, tools/pmc_summary.py",
           "pass_insts": i, "pass_busy": b, "nodes": nodes,
           "per_node": {k: v["sum"] / nodes for k, v in i.items() if k.startswith("SQ_INSTS")},
           "wave_cycles_per_node_x4": wc * 4 / nodes,
           "share_of_wave_cycles": {k: b[k]["sum"] / wc for k in ("SQ_WAIT_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_ANY")}},
          open(os.path.join(OUT, f"{TAG}_pmc.json"), "w"), indent=1)
fetch_b, write_b = f["FETCH_SIZE"]["sum"] * 1024, w["WRITE_SIZE"]["sum"] * 1024
alg = nodes * 496 + leaves * 260
json.dump({"_source": src + ": separate passes `rocprofv3 --pmc FETCH_SIZE --kernel-trace` and `--pmc WRITE_SIZE --kernel-trace` over `python3 bench.py "
                      "--steps 1 --warmup 0 --no-cpu-baseline` (one solve of partialorder_14), summed over all k_expand dispatches of the solve. "
                      "FETCH_SIZE / WRITE_SIZE are KiB. gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE tallies 128-B requests at 64 B, so "
                      "reads are doubled; our loads are 4 B per lane (a 248-B block per wavefront), which the guide calls uncalibrated -- the doubled "
                      "figure is the upper bound, the raw one the lower.",
           "kernel": "k_expand", "fetch_size_kib": f["FETCH_SIZE"]["sum"], "write_size_kib": w["WRITE_SIZE"]["sum"],
           "hbm_bytes_per_solve_raw": fetch_b + write_b, "hbm_bytes_per_solve_corrected": 2 * fetch_b + write_b,
           "algorithmic_bytes_per_solve": alg, "traffic_over_algorithmic": (2 * fetch_b + write_b) / alg,
           "note": "bench.py divides hbm_bytes_per_solve_corrected by its own launches per solve to report roofline.traffic per launch. Writes exceed "
                   "the algorithmic figure because node records are 272 B (16-B header + 62 words, padded to 16 B) against 248 B of block, edge "
                   "records 144 B against 132 B, and records that are not 128-B aligned touch partial lines."},
          open(os.path.join(OUT, f"{TAG}_traffic.json"), "w"), indent=1)
d = json.load(open(os.path.join(OUT, f"{TAG}_pmc.json")))
print("per node", {k: round(v, 1) for k, v in d["per_node"].items()}, "wave cycles", round(d["wave_cycles_per_node_x4"]), {k: round(v, 3) for k, v in d["share_of_wave_cycles"].items()})
print(open(os.path.join(OUT, f"{TAG}_kernel_stats.csv")).read().split("\n")[1])
sb = json.load(open(os.path.join(RAW, "stats_bench.json")))
print("bench under rocprofv3:", sb["value"], sb["ms_per_step"], sb["roofline"]["avg_launch_us"], sb["roofline"]["launches"])
