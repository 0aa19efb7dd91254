#!/usr/bin/env python3
"""bench.py -- search-tree nodes/sec of the MI355X engine on partialorder_14.csp.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A "step" is one complete solve of the instance: propagation + search from the root until the
whole automaton (state table + edge log) is resident in HBM.  The problem is already compiled
and resident on the device when the timed region starts; copying the automaton to the host is
NOT in `value` (it is reported as `export_ms`).  The unit is the search-tree node = one
propagation-to-fixpoint + classification, the engine's analogue of one solverSolveRe call
(reference src/solveralgorithm.cpp:733); each implementation counts its own tree.

One JSON line on stdout (rank 0).  Extra objects:
  roofline     dominant kernel k_expand: algorithmic bytes per launch / average launch duration,
               HIP events on the engine's stream, against the 8 TB/s HBM3E peak
  cpu_baseline oracle/ref_dfs.cpp (the reference's algorithm restated, 1 thread) time-boxed on
               the same instance on this box's host cores
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


def cpu_baseline(st, name, seconds):
    """Time-boxed run of the reference-faithful CPU restatement (kind = "port")."""
    path = REPO / "oracle" / "libstcsp_oracle.so"
    lib = C.CDLL(str(path))
    st.bind_engine_api(lib, "stcsp_oracle")

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
    return {"value": nodes / wall, "unit": "search-tree nodes/s", "cores": 1, "kind": "port",
            "sample": f"first {wall:.1f} s of the DFS on {name} ({nodes} nodes, "
                      f"{'truncated' if r.truncated else 'complete'}); oracle/ref_dfs.cpp, 1 thread",
            "host_cpus": os.cpu_count()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default=WORKLOAD)
    ap.add_argument("--batch", type=int, default=0)
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--stepped", action="store_true",
                    help="N=1 only: run the sharded pipeline (size-1 RCCL group, STCSP_F_STEPPED) -- the N>1 code path on one GPU")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    st = importlib.import_module("stcsp-solver_amd")
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
    flags = st.F_NO_EXPORT | st.F_PROFILE | (st.F_STEPPED if args.stepped and world == 1 else 0)
    eng = st.Engine(model, device=local_rank, rank=rank, world=world, flags=flags, batch_nodes=args.batch)

    def one_step():
        if not stepped:
            return eng.solve().counters  # the engine reads its counters once, at the end of the solve
        from importlib import import_module
        sh = import_module("stcsp-solver_amd.sharded")
        sh.solve_sharded(eng, rank, world, dev)
        return eng.counters()

    def barrier():
        if stepped:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        one_step()
    barrier()
    t0 = time.perf_counter()
    nodes = leaves = revs = evals = wrevs = sweeps = 0
    k_time = 0.0
    k_launches = 0
    for _ in range(args.steps):
        c = one_step()
        nodes += c.search_nodes
        leaves += c.leaves
        revs += c.revisions
        evals += c.evaluations
        wrevs += c.wave_revisions
        sweeps += c.sweeps
        k_time += c.seconds_expand_kernel
        k_launches += c.expand_launches
    barrier()
    elapsed = time.perf_counter() - t0
    levels = c.levels
    if stepped:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()
        t = torch.tensor([nodes, leaves], dtype=torch.int64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        nodes, leaves = t.tolist()

    # automaton export cost (PCIe D2H + ok-fixpoint), outside the timed region
    export_ms = None
    check = {}
    if not stepped:
        t1 = time.perf_counter()
        res = eng.export()
        export_first_ms = (time.perf_counter() - t1) * 1e3  # includes pinning the result buffers
        t1 = time.perf_counter()
        res = eng.export()
        export_ms = (time.perf_counter() - t1) * 1e3        # steady state (buffers exist)
        if os.environ.get("STCSP_DEBUG"):
            print("[bench] export python-side ms", export_ms, "engine-side ms", res.counters.seconds_export * 1e3, file=sys.stderr)
        a = eng.automaton(res).traverse().renumber()
        check = {"states": a.n_live_states, "edges": a.n_live_edges, "canonical_sha256": a.canonical_sha256()}

    if stepped:
        # parity of the sharded run (outside the timed region): gather the shards, merge, canonical hash
        try:
            from importlib import import_module
            sh = import_module("stcsp-solver_amd.sharded")
            merged = sh.gather_and_merge(st, eng, rank, world)
            if rank == 0:
                _, mres = merged
                a = st.Automaton(model, mres).traverse().renumber()
                check = {"states": a.n_live_states, "edges": a.n_live_edges, "canonical_sha256": a.canonical_sha256()}
        except Exception as ex:  # the throughput line must survive a failure of the check
            check = {"error": f"{type(ex).__name__}: {ex}"}

    if rank == 0:
        p = model.problem.contents
        N, K = p.n_vars, p.prefix_k
        S = res.sig_len if not stepped else model.n_vars  # sharded runs: upper bound for the per-leaf key bytes
        # B_node = 2*N*K*W*4 (read the parent block, write the child block)
        # + per leaf: key probe + key store + edge record
        b_node = 2 * N * K * 1 * 4
        b_leaf = 4 * (S + 1) * 2 + 8 + 4 * N
        alg_bytes = nodes * b_node + leaves * b_leaf
        per_launch_bytes = alg_bytes / max(k_launches, 1)
        avg_launch_s = k_time / max(k_launches, 1)
        achieved = per_launch_bytes / avg_launch_s / 1e9 if avg_launch_s > 0 else 0.0
        # HBM traffic of the same kernel from the committed PMC passes (bench.py cannot collect
        # counters itself): profiles/r01_chain_p14_traffic.json holds FETCH_SIZE + WRITE_SIZE summed
        # over every k_expand dispatch of one solve (gfx950-corrected); per launch = / launches per solve
        traffic = None
        tf = REPO / "profiles" / "r01_chain_p14_traffic.json"
        if args.workload == WORKLOAD and not stepped and tf.exists() and k_launches:
            traffic = json.loads(tf.read_text())["hbm_bytes_per_solve_corrected"] / (k_launches / args.steps)
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
            "config": {"workload": f"{args.workload}.csp (generated by stcsp-solver_amd/instances.py; "
                                   f"{N} vars incl. aux, prefix K={K}, whole frontier resident in HBM)",
                       "nodes_per_step": nodes // args.steps, "leaves_per_step": leaves // args.steps,
                       "launch_rounds_per_step": int(levels),
                       "per_node": {"item_revisions": revs / max(nodes, 1), "tuple_evaluations": evals / max(nodes, 1),
                                    "wavefront_revisions": wrevs / max(nodes, 1), "sweeps": sweeps / max(nodes, 1)},
                       "sharding": "none" if not stepped else f"state-owner x{world}"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "kernel": "k_expand",
                         "alg_bytes_per_launch": per_launch_bytes, "avg_launch_us": avg_launch_s * 1e6,
                         "launches": k_launches, "bytes_per_node": b_node, "bytes_per_leaf": b_leaf,
                         "kernel_time_share": k_time / elapsed if elapsed > 0 else None},
            "export_ms": export_ms,
            "export_first_ms": export_first_ms if not stepped else None,
            "parity": check,
        }
        if not args.no_cpu_baseline and world == 1 and not args.stepped:
            out["cpu_baseline"] = cpu_baseline(st, args.workload, args.cpu_seconds)
        print(json.dumps(out), flush=True)
    if stepped:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
