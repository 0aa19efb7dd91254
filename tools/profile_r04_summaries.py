"""Turn the raw output of tools/profile_r04.sh (gpurun_out/prof_r04/) into the summaries committed under profiles/:
kernel stats csv, instruction-mix / wait-share json, HBM traffic json (tagged with the engine source hash bench.py checks),
and the per-dispatch table the kernel-trace summary is computed from -- partialorder_14 (headline), digitinvader9 and the
synthetic 64 x 32 instance."""
import csv, glob, importlib, json, os, shutil, subprocess, sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
RAW = os.path.join(REPO, "gpurun_out", "prof_r04")
OUT = os.path.join(REPO, "profiles")
st = importlib.import_module("stcsp-solver_amd")
SHA = st.engine_source_sha()
HEAD = subprocess.run(["git", "-C", REPO, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()


def newest(pattern):
    return sorted(glob.glob(pattern, recursive=True), key=os.path.getmtime)[-1]


def summ(d, sub="k_expand"):
    return json.loads(subprocess.check_output([sys.executable, os.path.join(REPO, "tools", "pmc_summary.py"), os.path.join(RAW, d), sub]))


def bench_line(path):
    for line in open(path):
        if line.startswith("{"):
            return json.loads(line)
    raise SystemExit(f"no JSON line in {path}")


src = f"tools/profile_r04.sh (engine source sha {SHA}, git {HEAD})"
tag = {"engine_source_sha": SHA, "git_head": HEAD}
SHARES = ("SQ_WAIT_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_LDS")


def pmc_json(name, insts, busy, nodes, source, extra=None):
    wc = insts["SQ_WAVE_CYCLES"]["sum"]
    d = {"_source": source, **tag, "pass_insts": insts, "pass_busy": busy, "nodes_of_the_instruction_pass": nodes,
         "per_node": {k: v["sum"] / nodes for k, v in insts.items() if k.startswith("SQ_INSTS")},
         "instructions_per_node": sum(v["sum"] for k, v in insts.items() if k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_SMEM", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR")) / nodes,
         "wave_cycles_per_node_x4": wc * 4 / nodes}
    if busy:
        # the two passes are separate runs (time-boxed workloads differ in length): shares are taken against the busy pass's own
        # total where it has one, else against the instruction pass's wave cycles
        d["share_of_wave_cycles"] = {k: busy[k]["sum"] / wc for k in SHARES if k in busy}
    if extra:
        d.update(extra)
    json.dump(d, open(os.path.join(OUT, name), "w"), indent=1)
    return d


# ---- partialorder_14
shutil.copy(newest(os.path.join(RAW, "stats_p14", "**", "*_kernel_stats.csv")), os.path.join(OUT, "r04_p14_kernel_stats.csv"))
f, w, i, b = summ("pmc_FETCH_SIZE"), summ("pmc_WRITE_SIZE"), summ("pmc_SQ_INSTS_VALU"), summ("pmc_SQ_BUSY_CYCLES")
bench = bench_line(os.path.join(RAW, "pmc_FETCH_SIZE.json"))
nodes, leaves = bench["config"]["nodes_per_step"], bench["config"]["leaves_per_step"]
d = pmc_json("r04_p14_pmc.json", i, b, nodes,
             src + ": rocprofv3 --pmc <list> --kernel-trace -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-other-workloads "
                   "(two solves of partialorder_14: the timed one and the compacting-export comparison; counters summed over all k_expand dispatches "
                   "and divided by the nodes of both)", None)
# (bench.py solves the instance twice per run since round 3: the timed step and the STCSP_STREAM_EXPORT=0 comparison)
solves = 2
for k in d["per_node"]:
    d["per_node"][k] /= solves
d["instructions_per_node"] /= solves
d["wave_cycles_per_node_x4"] /= solves
d["solves_in_the_run"] = solves
json.dump(d, open(os.path.join(OUT, "r04_p14_pmc.json"), "w"), indent=1)
fetch_b, write_b = f["FETCH_SIZE"]["sum"] * 1024 / solves, w["WRITE_SIZE"]["sum"] * 1024 / solves
alg = nodes * bench["roofline"]["bytes_per_node"] + leaves * bench["roofline"]["bytes_per_leaf"]
json.dump({"_source": src + ": separate passes `rocprofv3 --pmc FETCH_SIZE --kernel-trace` and `--pmc WRITE_SIZE --kernel-trace` over `python3 bench.py "
                      "--steps 1 --warmup 0 --no-cpu-baseline --no-other-workloads` (two solves of partialorder_14 per run, halved), summed over all k_expand "
                      "dispatches. FETCH_SIZE / WRITE_SIZE are KiB. gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE tallies 128-B requests at "
                      "64 B, so reads are doubled; our loads are 4 B per lane (a 248-B block per wavefront), which the guide calls uncalibrated -- the "
                      "doubled figure is the upper bound, the raw one the lower.",
           **tag, "kernel": "k_expand", "fetch_size_kib": f["FETCH_SIZE"]["sum"] / solves, "write_size_kib": w["WRITE_SIZE"]["sum"] / solves,
           "hbm_bytes_per_solve_raw": fetch_b + write_b, "hbm_bytes_per_solve_corrected": 2 * fetch_b + write_b,
           "algorithmic_bytes_per_solve": alg, "traffic_over_algorithmic": (2 * fetch_b + write_b) / alg,
           "note": "bench.py divides hbm_bytes_per_solve_corrected by its own launches per solve to report roofline.traffic per launch, and only when "
                   "engine_source_sha equals the hash of the sources it runs on."},
          open(os.path.join(OUT, "r04_p14_traffic.json"), "w"), indent=1)
sb = bench_line(os.path.join(RAW, "stats_p14.json"))
json.dump(sb, open(os.path.join(OUT, "r04_p14_bench_under_rocprof.json"), "w"), indent=1)


def trace_summary(run, bench, table_name, trailing_solves=1):
    """k_expand dispatch durations of the kernel trace: all of them (what --stats averages: cold solves with their pool growth
    and the launches past the end of a burst included) and the timed steps only. The per-dispatch durations of the timed steps
    are written out too (profiles/<table_name>), so the averages can be recomputed without the raw trace."""
    rows = list(csv.DictReader(open(newest(os.path.join(RAW, run, "**", "*_kernel_trace.csv")))))
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if "k_expand" in r["Kernel_Name"]]
    # bench.py: warmup + steps timed-style solves, then ONE more solve (compacting-export comparison); every solve is one burst of 32 launches
    per_solve = bench["roofline"]["launches"] // bench["steps"]
    n_timed = bench["roofline"]["launches"]
    # (with STCSP_STREAM_EXPORT=0 set by the caller bench.py skips the comparison solve since round 4: trailing_solves = 0)
    timed = d[-(n_timed + trailing_solves * per_solve):(-trailing_solves * per_solve) or None]
    with open(os.path.join(OUT, table_name), "w") as fh:
        fh.write("# k_expand dispatches of the timed steps, in order: step, launch within the step, duration_us\n")
        for k, us in enumerate(timed):
            fh.write(f"{k // per_solve},{k % per_solve},{us:.3f}\n")
    copies = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if "copyBuffer" in r["Kernel_Name"]]
    return {"k_expand_dispatches": len(d), "avg_us_all": sum(d) / len(d), "timed_dispatches": len(timed), "avg_us_timed_steps": sum(timed) / len(timed),
            "kernel_ms_per_timed_solve": sum(timed) / bench["steps"] / 1e3, "bench_avg_launch_us_hip_events": bench["roofline"]["avg_launch_us"],
            "per_dispatch_table": "profiles/" + table_name,
            "blit_copy_kernels": len(copies), "blit_copy_avg_us": (sum(copies) / len(copies)) if copies else 0.0,
            "bench_value": bench["value"], "bench_search_only_nodes_per_s": bench["search_only_nodes_per_s"]}


ts = {"_source": src + ": per-dispatch durations from the kernel traces of the two --stats runs. `streaming`: the bench command as is; under the "
                   "tracer the D2H copies of the streaming export run as blit kernels (__amd_rocclr_copyBuffer) on the CUs instead of on the SDMA "
                   "engines and slow k_expand. `no_streaming`: STCSP_STREAM_EXPORT=0, the kernel by itself.", **tag,
      "streaming": trace_summary("stats_p14", sb, "r04_p14_timed_dispatches.csv")}
if os.path.isdir(os.path.join(RAW, "stats_p14_nostream")):
    sbn = bench_line(os.path.join(RAW, "stats_p14_nostream.json"))
    shutil.copy(newest(os.path.join(RAW, "stats_p14_nostream", "**", "*_kernel_stats.csv")), os.path.join(OUT, "r04_p14_nostream_kernel_stats.csv"))
    ts["no_streaming"] = trace_summary("stats_p14_nostream", sbn, "r04_p14_nostream_timed_dispatches.csv", trailing_solves=0)
json.dump(ts, open(os.path.join(OUT, "r04_p14_kernel_trace_summary.json"), "w"), indent=1)
print("trace summary", json.dumps({k: ({a: (round(b, 1) if isinstance(b, float) else b) for a, b in v.items()} if isinstance(v, dict) else "") for k, v in ts.items() if k not in ("_source", "engine_source_sha", "git_head")}))
print("p14 per node", {k: round(v, 1) for k, v in d["per_node"].items()}, "instructions", round(d["instructions_per_node"]), "wave cycles", round(d["wave_cycles_per_node_x4"]),
      {k: round(v, 3) for k, v in d.get("share_of_wave_cycles", {}).items()})
print(open(os.path.join(OUT, "r04_p14_kernel_stats.csv")).read().split("\n")[1])
print("bench under rocprofv3:", sb["value"], sb["search_only_nodes_per_s"], sb["ms_per_step"], sb["roofline"]["avg_launch_us"], sb["roofline"]["launches"])
# ---- digitinvader9
shutil.copy(newest(os.path.join(RAW, "stats_d9", "**", "*_kernel_stats.csv")), os.path.join(OUT, "r04_d9_kernel_stats.csv"))
i9, b9c = summ("pmc_d9_SQ_INSTS_VALU"), summ("pmc_d9_SQ_BUSY_CYCLES")
b9 = bench_line(os.path.join(RAW, "pmc_d9_SQ_INSTS_VALU.json"))
n9 = b9["config"]["nodes_per_step"] * 2  # two solves per run (see above)
d9 = pmc_json("r04_d9_pmc.json", i9, b9c, n9,
              src + ": rocprofv3 --pmc <list> --kernel-trace -- python3 bench.py --workload digitinvader9 --steps 1 --warmup 0 (two solves), separate instruction "
                    "and wait/busy passes, sums over the k_expand dispatches",
              {"bench_line_of_the_stats_run": bench_line(os.path.join(RAW, "stats_d9.json"))})
print("d9 per node", {k: round(v, 1) for k, v in d9["per_node"].items()}, {k: round(v, 3) for k, v in d9.get("share_of_wave_cycles", {}).items()})
print(open(os.path.join(OUT, "r04_d9_kernel_stats.csv")).read().split("\n")[1])
# ---- synthetic 64 x 32
shutil.copy(newest(os.path.join(RAW, "stats_synth", "**", "*_kernel_stats.csv")), os.path.join(OUT, "r04_synth_kernel_stats.csv"))
json.dump(bench_line(os.path.join(RAW, "stats_synth.json")), open(os.path.join(OUT, "r04_synth_bench_under_rocprof.json"), "w"), indent=1)
isy, bsy = summ("pmc_synth_SQ_INSTS_VALU"), summ("pmc_synth_SQ_BUSY_CYCLES")
nsy = bench_line(os.path.join(RAW, "pmc_synth_SQ_INSTS_VALU.json"))["nodes_all_solves"]
nsy_busy = bench_line(os.path.join(RAW, "pmc_synth_SQ_BUSY_CYCLES.json"))["nodes_all_solves"]
revs_per_node = {k: bench_line(os.path.join(RAW, f"pmc_synth_{k}.json"))["per_node"]["item_revisions_all_solves"]
                 for k in ("SQ_INSTS_VALU", "SQ_BUSY_CYCLES", "TCC_HIT_sum", "FETCH_SIZE", "WRITE_SIZE")}
tcc, fs, ws = summ("pmc_synth_TCC_HIT_sum"), summ("pmc_synth_FETCH_SIZE"), summ("pmc_synth_WRITE_SIZE")
n_tcc = bench_line(os.path.join(RAW, "pmc_synth_TCC_HIT_sum.json"))["nodes_all_solves"]
n_fs = bench_line(os.path.join(RAW, "pmc_synth_FETCH_SIZE.json"))["nodes_all_solves"]
n_ws = bench_line(os.path.join(RAW, "pmc_synth_WRITE_SIZE.json"))["nodes_all_solves"]
# every pass is its own run of two NODE-bounded solves (warm + measured, max_search_nodes = 40 M each); the counters cover both and
# are divided by the nodes the engine itself counted over both (nodes_all_solves of that pass)
extra = {"note": "tools/synth_bench.py --nodes 40000000: two node-bounded solves per pass; every per-node figure = counter sum of the pass / "
                 "nodes_all_solves of the SAME pass (engine-counted). item_revisions_per_node_by_pass shows that the passes did the same work per node.",
         "item_revisions_per_node_by_pass": revs_per_node,
         "busy_pass_nodes": nsy_busy,
         "share_of_wave_cycles_busy_pass_scaled": {k: bsy[k]["sum"] / (isy["SQ_WAVE_CYCLES"]["sum"] * nsy_busy / nsy) for k in SHARES if k in bsy},
         "l2": {"TCC_HIT_sum": tcc["TCC_HIT_sum"]["sum"], "TCC_MISS_sum": tcc["TCC_MISS_sum"]["sum"], "TCC_REQ_sum": tcc["TCC_REQ_sum"]["sum"],
                "hit_rate": tcc["TCC_HIT_sum"]["sum"] / max(1.0, tcc["TCC_HIT_sum"]["sum"] + tcc["TCC_MISS_sum"]["sum"]),
                "requests_per_node": tcc["TCC_REQ_sum"]["sum"] / n_tcc, "nodes": n_tcc},
         "hbm": {"fetch_bytes_per_node_raw": fs["FETCH_SIZE"]["sum"] * 1024 / n_fs, "write_bytes_per_node": ws["WRITE_SIZE"]["sum"] * 1024 / n_ws,
                 "algorithmic_bytes_per_node": 2 * 70 * 2 * 4}}
dsy = pmc_json("r04_synth_pmc.json", isy, None, nsy,
               src + ": rocprofv3 --pmc <list> --kernel-trace -- python3 tools/synth_bench.py --nodes 40000000 (synthetic 64 x 32, 602 + 6 constraints, seed 20261003), "
                     "separate passes for instructions, wait/busy shares, L2 (TCC) and HBM (FETCH_SIZE / WRITE_SIZE); sums over the k_expand dispatches", extra)
print("synth per node", {k: round(v, 1) for k, v in dsy["per_node"].items()}, "L2", {k: (round(v, 3) if isinstance(v, float) else v) for k, v in extra["l2"].items()},
      {k: round(v, 3) for k, v in extra["share_of_wave_cycles_busy_pass_scaled"].items()})
print(open(os.path.join(OUT, "r04_synth_kernel_stats.csv")).read().split("\n")[1])
