#!/usr/bin/env python3
"""bench.py -- search-tree nodes/sec of the MI355X engine on partialorder_14.csp.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A "step" is one complete solve of the instance as SURVEY.md section 8(d) defines the metric: from
`solve` entry (the compiled problem is resident on the device) until the raw automaton is available
on the HOST -- propagation + search, the ok/fail fixpoint (reference src/solveralgorithm.cpp:857-874,
904-909), compaction of the live edges and the device-to-host copy of the C-ABI result arrays.
`value` = nodes / that whole time; the search-only rate (automaton left in HBM) is reported beside it
as `search_only_nodes_per_s`.  The unit is the search-tree node = one propagation-to-fixpoint +
classification, the engine's analogue of one solverSolveRe call (reference
src/solveralgorithm.cpp:733); each implementation counts its own tree.

The automaton of the last timed step is checked against the reference's recorded canonical sha256
(tests/golden/reference_golden.json): a mismatch makes bench.py exit non-zero.

One JSON line on stdout (rank 0).  Extra objects:
  roofline     dominant kernel k_expand: algorithmic bytes per launch / average launch duration,
               HIP events on the engine's stream, against the 8 TB/s HBM3E peak
  cpu_baseline oracle/ref_dfs.cpp (the reference's algorithm restated, 1 thread) time-boxed on
               the same instance on this box's host cores
  config.other_workloads  (outside the timed region) the other BASELINE configs on this GPU:
               digitinvader9, juggling_b5_f6, the time-boxed synthetic 64 x 32 instance
"""
import argparse
import ctypes as C
import importlib
import json
import os
import sys
import time
from pathlib import Path

REPO = Path(__file__).resolve().parent
sys.path.insert(0, str(REPO))

WORKLOAD = "partialorder_14"
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
SYNTH = (64, 32, 602, 6, 20261003)  # BASELINE config 4: 64 vars x |D| = 32, 602 point + 6 next table constraints
REFERENCE_P14_NODES_PER_S = 10.9e3  # the reference itself, survey VM, 1 core


def cpu_baseline(st, name, seconds, threads):
    """Time-boxed runs of the reference-faithful CPU restatement (kind = "port"): one thread (the reference is
    single-threaded), and `threads` workers over the automaton's states (oracle/ref_dfs.cpp struct Shared)."""
    path = REPO / "oracle" / "libstcsp_oracle.so"
    lib = C.CDLL(str(path))
    st.bind_engine_api(lib, "stcsp_oracle")
    lib.stcsp_oracle_solve_parallel.argtypes = [C.c_void_p, C.c_int, C.POINTER(st.Result)]

    class Ref(st.EngineBase):
        _prefix = "stcsp_oracle"

        def __init__(self, model, **o):
            super().__init__(lib, model, **o)

    m = st.Model.from_name(name)
    o = Ref(m, time_limit_s=seconds)
    t0 = time.time()
    r = o.solve()
    wall = time.time() - t0
    nodes = r.counters.search_nodes
    out = {"value": nodes / wall, "unit": "search-tree nodes/s", "cores": 1, "kind": "port",
           "sample": f"first {wall:.1f} s of the DFS on {name} ({nodes} nodes, "
                     f"{'truncated' if r.truncated else 'complete'}); oracle/ref_dfs.cpp, 1 thread. The reference itself "
                     f"(unbuildable on this box) measured {REFERENCE_P14_NODES_PER_S:.0f} nodes/s on partialorder_14 on the "
                     "survey VM (1 core; it leaks 16 kB per leaf, half of its time is page faults)",
           "host_cpus": os.cpu_count()}
    if threads > 1:
        op = Ref(m, time_limit_s=seconds)
        t0 = time.time()
        op._check(lib.stcsp_oracle_solve_parallel(op._h, threads, C.byref(op.result)))
        wall = time.time() - t0
        rp = op.result
        out["all_cores"] = {"value": rp.counters.search_nodes / wall, "unit": "search-tree nodes/s", "cores": threads, "kind": "port",
                            "sample": f"{wall:.1f} s on {name} ({rp.counters.search_nodes} nodes, {'truncated' if rp.truncated else 'complete'}): "
                                      f"the same restatement with {threads} workers over the automaton's states (shared state table, "
                                      "ok/fail by fixpoint)"}
    return out


def alg_bytes(model, res_sig_len, nodes, leaves):
    """B_node = 2*N*K*W*4 (read the parent block, write the child block)
    + per leaf: key probe + key store + edge record."""
    p = model.problem.contents
    N, K = p.n_vars, p.prefix_k
    b_node = 2 * N * K * 1 * 4
    b_leaf = 4 * (res_sig_len + 1) * 2 + 8 + 4 * N
    return nodes * b_node + leaves * b_leaf, b_node, b_leaf


def other_workload(st, golden, name, device, repeats=3):
    """One of the other BASELINE configs, full solve (search + export to the host), best of `repeats`."""
    m = st.Model.from_name(name)
    e = st.Engine(m, device=device, flags=st.F_PROFILE)
    e.solve()
    best = None
    for _ in range(repeats):
        t0 = time.perf_counter()
        r = e.solve()
        dt = time.perf_counter() - t0
        if best is None or dt < best[0]:
            c = r.counters
            best = (dt, c.search_nodes, c.leaves, c.seconds_search, c.seconds_expand_kernel, c.expand_launches, c.levels)
    dt, nodes, leaves, s_search, k_time, launches, levels = best
    a = e.automaton(r).traverse().renumber()
    sha = a.canonical_sha256()
    ab, b_node, b_leaf = alg_bytes(m, r.sig_len, nodes, leaves)
    out = {"nodes": nodes, "ms_per_solve": dt * 1e3, "nodes_per_s": nodes / dt, "search_only_ms": s_search * 1e3,
           "search_only_nodes_per_s": nodes / s_search if s_search > 0 else None, "launch_rounds": int(levels),
           "alg_bytes": ab, "bytes_per_node": b_node, "bytes_per_leaf": b_leaf,
           "k_expand_GBps": ab / k_time / 1e9 if k_time > 0 else None,
           "hbm_frac": ab / k_time / 1e9 / HBM_PEAK_GBS if k_time > 0 else None,
           "states": a.n_live_states, "edges": a.n_live_edges, "canonical_sha256": sha,
           "parity_ok": sha == golden[name]["canonical_sha256"]}
    e.close()
    return out


def synthetic_workload(st, device, seconds):
    """BASELINE config 4, time-boxed (nobody reaches a leaf on this instance: SURVEY 8(d))."""
    n, d, mp, s, seed = SYNTH
    m = st.Model(text=st.instances.synthetic(n, d, mp, s, seed))
    e = st.Engine(m, device=device, flags=st.F_NO_EXPORT | st.F_PROFILE, time_limit_s=seconds)
    e.solve()  # the first time-boxed solve grows the frontier arena to its working size (GBs of hipMalloc + copies)
    r = e.solve()
    c = r.counters
    p = m.problem.contents
    b_node = 2 * p.n_vars * p.prefix_k * 4
    out = {"shape": f"{n} vars x |D|={d}, {mp} point + {s} next table constraints, seed {seed}", "time_box_s": seconds,
           "nodes": c.search_nodes, "fails": c.fails, "leaves": c.leaves, "nodes_per_s": c.search_nodes / c.seconds_search,
           "bytes_per_node": b_node,
           "k_expand_GBps": c.search_nodes * b_node / c.seconds_expand_kernel / 1e9 if c.seconds_expand_kernel > 0 else None,
           "hbm_frac": c.search_nodes * b_node / c.seconds_expand_kernel / 1e9 / HBM_PEAK_GBS if c.seconds_expand_kernel > 0 else None,
           "launch_rounds": int(c.levels)}
    e.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default=WORKLOAD)
    ap.add_argument("--batch", type=int, default=0)
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-threads", type=int, default=0, help="workers of the all-cores CPU leg (default: min(host cpus, 16), the box's CPU share)")
    ap.add_argument("--no-other-workloads", action="store_true")
    ap.add_argument("--synthetic-seconds", type=float, default=2.0)
    ap.add_argument("--stepped", action="store_true",
                    help="N=1 only: run the sharded pipeline (size-1 RCCL group, STCSP_F_STEPPED) -- the N>1 code path on one GPU")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    st = importlib.import_module("stcsp-solver_amd")
    golden = json.loads((REPO / "tests" / "golden" / "reference_golden.json").read_text())
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device(f"cuda:{local_rank}")
    stepped = world > 1 or args.stepped
    if stepped:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29519")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    # a fresh checkout has no built libraries: rank 0 compiles them once, the others wait
    if rank == 0 and (not (st.CSRC / "libstcsp_hip.so").exists() or not (REPO / "oracle" / "libstcsp_oracle.so").exists()):
        import __graft_entry__
        __graft_entry__.build()
    if world > 1:
        dist.barrier()

    model = st.Model.from_name(args.workload)
    flags = st.F_PROFILE | (st.F_STEPPED if args.stepped and world == 1 else 0)
    eng = st.Engine(model, device=local_rank, rank=rank, world=world, flags=flags, batch_nodes=args.batch)
    sh = importlib.import_module("stcsp-solver_amd.sharded") if stepped else None

    def one_step():
        """One full solve: search + ok-fixpoint + compaction + copy to the host (per shard when sharded)."""
        if not stepped:
            res = eng.solve()
            return res.counters, res
        sh.solve_sharded(eng, rank, world, dev)
        res = eng.export()  # this shard's states and raw edges on the host (the merge on rank 0 is outside the step)
        return eng.counters(), res

    def barrier():
        if stepped:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        one_step()
    barrier()
    t0 = time.perf_counter()
    nodes = leaves = revs = evals = wrevs = sweeps = 0
    k_time = s_search = s_export = 0.0
    k_launches = 0
    for _ in range(args.steps):
        c, res = one_step()
        nodes += c.search_nodes
        leaves += c.leaves
        revs += c.revisions
        evals += c.evaluations
        wrevs += c.wave_revisions
        sweeps += c.sweeps
        k_time += c.seconds_expand_kernel
        k_launches += c.expand_launches
        s_search += c.seconds_search
        s_export += res.counters.seconds_export
    barrier()
    elapsed = time.perf_counter() - t0
    levels = c.levels
    if stepped:
        t = torch.tensor([elapsed, s_search], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, s_search = t.tolist()
        t = torch.tensor([nodes, leaves], dtype=torch.int64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        nodes, leaves = t.tolist()

    # parity of the last timed step (outside the timed region)
    check = {}
    if not stepped:
        a = eng.automaton(res).traverse().renumber()
        check = {"states": a.n_live_states, "edges": a.n_live_edges, "canonical_sha256": a.canonical_sha256()}
    else:
        try:  # gather the shards, merge, canonical hash
            merged = sh.gather_and_merge(st, eng, rank, world, dev)
            if rank == 0:
                _, mres = merged
                a = st.Automaton(model, mres).traverse().renumber()
                check = {"states": a.n_live_states, "edges": a.n_live_edges, "canonical_sha256": a.canonical_sha256()}
        except Exception as ex:
            check = {"error": f"{type(ex).__name__}: {ex}"}
    parity_ok = True
    if rank == 0 and args.workload in golden:
        check["golden_sha256"] = golden[args.workload]["canonical_sha256"]
        parity_ok = check.get("canonical_sha256") == check["golden_sha256"]
        check["ok"] = parity_ok

    if rank == 0:
        S = res.sig_len
        ab, b_node, b_leaf = alg_bytes(model, S, nodes, leaves)
        per_launch_bytes = ab / max(k_launches, 1)
        avg_launch_s = k_time / max(k_launches, 1)
        achieved = per_launch_bytes / avg_launch_s / 1e9 if avg_launch_s > 0 else 0.0
        # HBM traffic of the same kernel: FETCH_SIZE + WRITE_SIZE of every k_expand dispatch of one solve from
        # this round's separate --pmc passes (tools/profile_r02.sh; a PMC run cannot be part of the timed
        # bench), gfx950-corrected, per launch like `achieved`; null when no profile of this engine is committed
        traffic = None
        tf = REPO / "profiles" / "r02_p14_traffic.json"
        if args.workload == WORKLOAD and not stepped and tf.exists() and k_launches:
            traffic = json.loads(tf.read_text())["hbm_bytes_per_solve_corrected"] / (k_launches / args.steps)
        p = model.problem.contents
        cfg = {"workload": f"{args.workload}.csp (generated by stcsp-solver_amd/instances.py; "
                           f"{p.n_vars} vars incl. aux, prefix K={p.prefix_k}, whole frontier resident in HBM)",
               "timed_region": "solve entry -> raw automaton on the host (search, with the edge log and state keys streamed to pinned host arrays "
                               "on a second stream while it runs, + ok-fixpoint + the last chunks; compaction + D2H when a state failed)",
               "nodes_per_step": nodes // args.steps, "leaves_per_step": leaves // args.steps,
               "launch_rounds_per_step": int(levels),
               "per_node": {"item_revisions": revs / max(nodes, 1), "tuple_evaluations": evals / max(nodes, 1),
                            "wavefront_revisions": wrevs / max(nodes, 1), "sweeps": sweeps / max(nodes, 1)},
               "sharding": "none" if not stepped else f"state-owner x{world}"}
        if world == 1 and not args.stepped and not args.no_other_workloads:
            others = {}
            for name in ("digitinvader9", "juggling_b5_f6"):
                try:
                    others[name] = other_workload(st, golden, name, local_rank)
                    parity_ok = parity_ok and others[name]["parity_ok"]
                except Exception as ex:  # the headline line must survive
                    others[name] = {"error": f"{type(ex).__name__}: {ex}"}
                    parity_ok = False
            try:
                others["synthetic_64x32"] = synthetic_workload(st, local_rank, args.synthetic_seconds)
            except Exception as ex:
                others["synthetic_64x32"] = {"error": f"{type(ex).__name__}: {ex}"}
            cfg["other_workloads"] = others
        out = {
            "metric": "search-tree nodes/sec on partialorder_14.csp",
            "value": nodes / elapsed,
            "unit": "search-tree nodes/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "u32",
            "data": "synthetic",
            "config": cfg,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "traffic_source": "profiles/r02_p14_traffic.json: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over this command "
                                           "(tools/profile_r02.sh), gfx950-corrected, divided by this run's launches per solve" if traffic is not None else None,
                         "kernel": "k_expand",
                         "alg_bytes_per_launch": per_launch_bytes, "avg_launch_us": avg_launch_s * 1e6,
                         "launches": k_launches, "bytes_per_node": b_node, "bytes_per_leaf": b_leaf,
                         "kernel_time_share": k_time / elapsed if elapsed > 0 else None},
            "search_only_nodes_per_s": nodes / s_search if s_search > 0 else None,
            "search_ms": s_search / args.steps * 1e3,
            "export_ms": s_export / args.steps * 1e3,
            "parity": check,
        }
        if not args.no_cpu_baseline and world == 1 and not args.stepped:
            out["cpu_baseline"] = cpu_baseline(st, args.workload, args.cpu_seconds, args.cpu_threads or min(os.cpu_count() or 1, 16))
        print(json.dumps(out), flush=True)
    if stepped:
        ok = torch.tensor([1 if parity_ok else 0], dtype=torch.int64, device=dev)
        dist.broadcast(ok, 0)
        parity_ok = bool(ok.item())
        dist.destroy_process_group()
    if not parity_ok:
        print("bench.py: PARITY FAILURE -- the automaton differs from the reference's recorded canonical sha256", file=sys.stderr)
        sys.exit(3)


if __name__ == "__main__":
    main()
