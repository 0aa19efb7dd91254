"""Turn the raw output of tools/profile_r02.sh (gpurun_out/prof_r02/) into the summaries committed under
profiles/: kernel stats csv, instruction-mix / wave-cycle json, HBM traffic json -- partialorder_14 (headline),
digitinvader9 and the synthetic 64 x 32 instance."""
import glob, json, os, shutil, subprocess, sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RAW = os.path.join(REPO, "gpurun_out", "prof_r02")
OUT = os.path.join(REPO, "profiles")
NOTE = sys.argv[1] if len(sys.argv) > 1 else "round-2 engine"


def newest(pattern):
    return sorted(glob.glob(pattern, recursive=True), key=os.path.getmtime)[-1]


def summ(d):
    return json.loads(subprocess.check_output([sys.executable, os.path.join(REPO, "tools", "pmc_summary.py"), os.path.join(RAW, d)]))


def bench_line(path):
    for line in open(path):
        if line.startswith("{"):
            return json.loads(line)
    raise SystemExit(f"no JSON line in {path}")


src = f"tools/profile_r02.sh ({NOTE})"
# ---- partialorder_14
shutil.copy(newest(os.path.join(RAW, "stats_p14", "**", "*_kernel_stats.csv")), os.path.join(OUT, "r02_p14_kernel_stats.csv"))
f, w, i, b = summ("pmc_FETCH_SIZE"), summ("pmc_WRITE_SIZE"), summ("pmc_SQ_INSTS_VALU"), summ("pmc_SQ_BUSY_CYCLES")
bench = bench_line(os.path.join(RAW, "pmc_FETCH_SIZE.json"))
nodes, leaves = bench["config"]["nodes_per_step"], bench["config"]["leaves_per_step"]
wc = i["SQ_WAVE_CYCLES"]["sum"]
json.dump({"_source": src + ": rocprofv3 --pmc <list> --kernel-trace -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-other-workloads "
                      "(one cold solve of partialorder_14); sums over the k_expand dispatches, tools/pmc_summary.py",
           "pass_insts": i, "pass_busy": b, "nodes": nodes,
           "per_node": {k: v["sum"] / nodes for k, v in i.items() if k.startswith("SQ_INSTS")},
           "wave_cycles_per_node_x4": wc * 4 / nodes,
           "share_of_wave_cycles": {k: b[k]["sum"] / wc for k in ("SQ_WAIT_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_SCA")
                                    if k in b}},
          open(os.path.join(OUT, "r02_p14_pmc.json"), "w"), indent=1)
fetch_b, write_b = f["FETCH_SIZE"]["sum"] * 1024, w["WRITE_SIZE"]["sum"] * 1024
alg = nodes * bench["roofline"]["bytes_per_node"] + leaves * bench["roofline"]["bytes_per_leaf"]
json.dump({"_source": src + ": separate passes `rocprofv3 --pmc FETCH_SIZE --kernel-trace` and `--pmc WRITE_SIZE --kernel-trace` over `python3 bench.py "
                      "--steps 1 --warmup 0 --no-cpu-baseline --no-other-workloads` (one solve of partialorder_14), summed over all k_expand dispatches "
                      "of the solve. FETCH_SIZE / WRITE_SIZE are KiB. gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE tallies 128-B requests at "
                      "64 B, so reads are doubled; our loads are 4 B per lane (a 248-B block per wavefront), which the guide calls uncalibrated -- the "
                      "doubled figure is the upper bound, the raw one the lower.",
           "kernel": "k_expand", "fetch_size_kib": f["FETCH_SIZE"]["sum"], "write_size_kib": w["WRITE_SIZE"]["sum"],
           "hbm_bytes_per_solve_raw": fetch_b + write_b, "hbm_bytes_per_solve_corrected": 2 * fetch_b + write_b,
           "algorithmic_bytes_per_solve": alg, "traffic_over_algorithmic": (2 * fetch_b + write_b) / alg,
           "note": "bench.py divides hbm_bytes_per_solve_corrected by its own launches per solve to report roofline.traffic per launch."},
          open(os.path.join(OUT, "r02_p14_traffic.json"), "w"), indent=1)
sb = bench_line(os.path.join(RAW, "stats_p14.json"))
json.dump(sb, open(os.path.join(OUT, "r02_p14_bench_under_rocprof.json"), "w"), indent=1)


def trace_summary(run, bench):
    """k_expand dispatch durations of the kernel trace: all of them (what --stats averages: cold solves with their
    pool growth and the no-op launches past the end of a burst included) and the timed steps only."""
    import csv
    rows = list(csv.DictReader(open(newest(os.path.join(RAW, run, "**", "*_kernel_trace.csv")))))
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if "k_expand" in r["Kernel_Name"]]
    n_timed = bench["roofline"]["launches"]
    timed = d[-n_timed:]
    copies = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if "copyBuffer" in r["Kernel_Name"]]
    return {"k_expand_dispatches": len(d), "avg_us_all": sum(d) / len(d), "timed_dispatches": len(timed), "avg_us_timed_steps": sum(timed) / len(timed),
            "kernel_ms_per_timed_solve": sum(timed) / bench["steps"] / 1e3, "bench_avg_launch_us_hip_events": bench["roofline"]["avg_launch_us"],
            "blit_copy_kernels": len(copies), "blit_copy_avg_us": (sum(copies) / len(copies)) if copies else 0.0,
            "bench_value": bench["value"], "bench_search_only_nodes_per_s": bench["search_only_nodes_per_s"]}


ts = {"_source": src + ": per-dispatch durations from the kernel traces of the two --stats runs. `streaming`: the bench command as is; under the "
                   "tracer the D2H copies of the streaming export run as blit kernels (__amd_rocclr_copyBuffer) on the CUs instead of on the SDMA "
                   "engines, k_expand slows from ~107 to ~150 us per launch (HSA_ENABLE_SDMA=0 gives the same figure without the tracer). "
                   "`no_streaming`: STCSP_STREAM_EXPORT=0, the kernel by itself.",
      "streaming": trace_summary("stats_p14", sb)}
if os.path.isdir(os.path.join(RAW, "stats_p14_nostream")):
    sbn = bench_line(os.path.join(RAW, "stats_p14_nostream.json"))
    shutil.copy(newest(os.path.join(RAW, "stats_p14_nostream", "**", "*_kernel_stats.csv")), os.path.join(OUT, "r02_p14_nostream_kernel_stats.csv"))
    ts["no_streaming"] = trace_summary("stats_p14_nostream", sbn)
json.dump(ts, open(os.path.join(OUT, "r02_p14_kernel_trace_summary.json"), "w"), indent=1)
print("trace summary", json.dumps({k: ({a: (round(b, 1) if isinstance(b, float) else b) for a, b in v.items()} if isinstance(v, dict) else "") for k, v in ts.items() if k != "_source"}))
d = json.load(open(os.path.join(OUT, "r02_p14_pmc.json")))
print("p14 per node", {k: round(v, 1) for k, v in d["per_node"].items()}, "wave cycles", round(d["wave_cycles_per_node_x4"]),
      {k: round(v, 3) for k, v in d["share_of_wave_cycles"].items()})
print(open(os.path.join(OUT, "r02_p14_kernel_stats.csv")).read().split("\n")[1])
print("bench under rocprofv3:", sb["value"], sb["search_only_nodes_per_s"], sb["ms_per_step"], sb["roofline"]["avg_launch_us"], sb["roofline"]["launches"])
# ---- digitinvader9
shutil.copy(newest(os.path.join(RAW, "stats_d9", "**", "*_kernel_stats.csv")), os.path.join(OUT, "r02_d9_kernel_stats.csv"))
i9 = summ("pmc_d9")
b9 = bench_line(os.path.join(RAW, "pmc_d9.json"))
n9 = b9["config"]["nodes_per_step"]
json.dump({"_source": src + ": rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVES SQ_WAVE_CYCLES --kernel-trace -- python3 bench.py "
                      "--workload digitinvader9 --steps 1 --warmup 0 (one cold solve), sums over the k_expand dispatches",
           "pass_insts": i9, "nodes": n9, "per_node": {k: v["sum"] / n9 for k, v in i9.items() if k.startswith("SQ_INSTS")},
           "wave_cycles_per_node_x4": i9["SQ_WAVE_CYCLES"]["sum"] * 4 / n9,
           "bench_line_of_the_stats_run": bench_line(os.path.join(RAW, "stats_d9.json"))},
          open(os.path.join(OUT, "r02_d9_pmc.json"), "w"), indent=1)
print("d9 per node", {k: round(v["sum"] / n9, 1) for k, v in i9.items() if k.startswith("SQ_INSTS")})
print(open(os.path.join(OUT, "r02_d9_kernel_stats.csv")).read().split("\n")[1])
# ---- synthetic 64 x 32
shutil.copy(newest(os.path.join(RAW, "stats_synth", "**", "*_kernel_stats.csv")), os.path.join(OUT, "r02_synth_kernel_stats.csv"))
json.dump(bench_line(os.path.join(RAW, "stats_synth.json")), open(os.path.join(OUT, "r02_synth_bench_under_rocprof.json"), "w"), indent=1)
print(open(os.path.join(OUT, "r02_synth_kernel_stats.csv")).read().split("\n")[1])
