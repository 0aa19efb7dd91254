#!/usr/bin/env python3
"""bench.py -- search-tree nodes/sec of the MI355X engine on partialorder_14.csp.

    python bench.py --gpus N --steps K --warmup W
N > 1 either way: under a launcher that sets RANK / LOCAL_RANK / WORLD_SIZE (python -m torch.distributed.run
--nnodes=1 --nproc-per-node N ... bench.py --gpus N ...), or plainly -- bench.py then starts its own N rank
processes (fresh children created BEFORE this process makes any GPU call, one per LOCAL_RANK), waits for them
and prints rank 0's line.

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
SYNTH_SHAPE = SYNTH  # --synthetic-shape (tests shrink it: the CPU stand-in engine manages ~3 k nodes/s on 64 x 32)
REFERENCE_P14_NODES_PER_S = 10.9e3  # the reference itself, survey VM, 1 core


def free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(args, argv):
    """`bench.py --gpus N` without a launcher: become the parent of N rank processes.  The parent never
    touches the GPU (counting devices does not initialise it) and never execs: the ranks are ordinary
    child processes; rank 0's stdout is this process's stdout.  A rank that dies takes the others with it
    (they would wait in a collective for ever), and the exit code is the worst of the ranks'."""
    import subprocess
    n = args.gpus
    if HOOKS["engine"] is None:
        import torch
        have = torch.cuda.device_count()
        if have < n:
            print(f"bench.py: --gpus {n} but this process sees {have} GPU(s)", file=sys.stderr)
            return 2
    env0 = dict(os.environ, WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()),
                LOCAL_WORLD_SIZE=str(n), STCSP_BENCH_SELF_LAUNCHED="1")
    env0.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    procs = []
    for r in range(n):
        env = dict(env0, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, str(HOOKS["entry"])] + argv, env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rcs = [None] * n
    deadline = None
    while any(rc is None for rc in rcs):
        for i, pr in enumerate(procs):
            if rcs[i] is None:
                rcs[i] = pr.poll()
        if any(rc not in (None, 0) for rc in rcs) and deadline is None:
            deadline = time.time() + 20.0  # the others get a moment to fail on their own (agreed errors), then go
        if deadline is not None and time.time() > deadline:
            for i, pr in enumerate(procs):
                if rcs[i] is None:
                    pr.kill()  # exactly the children started above
                    rcs[i] = pr.wait()
        time.sleep(0.05)
    bad = [rc for rc in rcs if rc != 0]
    return 0 if not bad else max(abs(rc) for rc in bad) or 1


def host_cpu_share():
    """(CPUs this process may run on, workers for the all-cores CPU legs, where that number comes from).  A GPU box hands a
    one-GPU job a SHARE of its host cores: the affinity mask (os.sched_getaffinity) or the cgroup's CPU quota (cpu.max) say
    so when the box enforces it; this pool enforces it with neither (round 4: the mask showed all 256 CPUs and 64 workers ran
    3.7x SLOWER than 16 -- the cores are shared with the other tenants), its documented share is 16 cores per GPU, which is
    therefore the cap."""
    try:
        mask = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        mask = os.cpu_count() or 1
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = max(1, int(int(q) / int(per)))
    except (OSError, ValueError):
        pass
    share = min(mask, quota) if quota else mask
    workers = max(1, min(share, 16))
    why = (f"affinity mask {mask} CPUs, cgroup quota {quota if quota else 'none'}; capped at the pool's documented share of 16 host cores per GPU")
    return share, workers, why


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
           "host_cpus": os.cpu_count(), "host_cpus_in_affinity_mask": host_cpu_share()[0], "all_cores_workers_from": host_cpu_share()[2]}
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
    widest = max(p.var_ub[i] - p.var_lb[i] + 1 for i in range(N))
    W = 1 if widest <= 32 else (2 if widest <= 64 else 4)  # bitset words per (variable, time point): the engine's W (dev_wide.hpp)
    b_node = 2 * N * K * W * 4
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


def synthetic_model(st):
    n, d, mp, s, seed = SYNTH_SHAPE
    return st.Model(text=st.instances.synthetic(n, d, mp, s, seed))


def synthetic_cpu(st, seconds, threads):
    """north_star's pair for the synthetic shape: the reference-faithful CPU restatement (oracle/ref_dfs.cpp, kind
    "port") time-boxed on the same instance on this box's host cores -- 1 thread, then `threads` workers."""
    lib = C.CDLL(str(REPO / "oracle" / "libstcsp_oracle.so"))
    st.bind_engine_api(lib, "stcsp_oracle")
    lib.stcsp_oracle_solve_parallel.argtypes = [C.c_void_p, C.c_int, C.POINTER(st.Result)]

    class Ref(st.EngineBase):
        _prefix = "stcsp_oracle"

        def __init__(self, model, **o):
            super().__init__(lib, model, **o)

    m = synthetic_model(st)
    o = Ref(m, time_limit_s=seconds)
    t0 = time.time()
    r = o.solve()
    wall = time.time() - t0
    out = {"one_core": {"nodes_per_s": r.counters.search_nodes / wall, "nodes": r.counters.search_nodes, "seconds": wall, "cores": 1},
           "kind": "port", "note": "oracle/ref_dfs.cpp (the reference's interval propagation + DFS restated), time-boxed, same instance; "
                                   "the reference itself reached no leaf in 25 min on this shape and prints nothing on timeout (BASELINE.md)"}
    if threads > 1:
        op = Ref(m, time_limit_s=seconds)
        t0 = time.time()
        op._check(lib.stcsp_oracle_solve_parallel(op._h, threads, C.byref(op.result)))
        wall = time.time() - t0
        out["all_cores"] = {"nodes_per_s": op.result.counters.search_nodes / wall, "nodes": op.result.counters.search_nodes, "seconds": wall,
                            "cores": threads, "note": "workers share the automaton's states; this instance never leaves its root state, so the "
                                                      "extra workers find nothing to take (the reference's unit of work is the state)"}
    return out


def synthetic_workload(st, device, seconds, cpu_seconds=0.0, cpu_threads=1):
    """BASELINE config 4, time-boxed (nobody reaches a leaf on this instance: SURVEY 8(d))."""
    n, d, mp, s, seed = SYNTH_SHAPE
    m = synthetic_model(st)
    e = st.Engine(m, device=device, flags=st.F_NO_EXPORT | st.F_PROFILE, time_limit_s=seconds)
    e.solve()  # the first time-boxed solve grows the frontier arena to its working size (GBs of hipMalloc + copies)
    r = e.solve()
    c = r.counters
    p = m.problem.contents
    b_node = 2 * p.n_vars * p.prefix_k * 4
    out = {"shape": f"{n} vars x |D|={d}, {mp} point + {s} next table constraints, seed {seed}", "time_box_s": seconds,
           "nodes": c.search_nodes, "fails": c.fails, "leaves": c.leaves, "nodes_per_s": c.search_nodes / c.seconds_search,
           "bytes_per_node": b_node, "item_revisions_per_node": c.revisions / max(c.search_nodes, 1),
           "k_expand_GBps": c.search_nodes * b_node / c.seconds_expand_kernel / 1e9 if c.seconds_expand_kernel > 0 else None,
           "hbm_frac": c.search_nodes * b_node / c.seconds_expand_kernel / 1e9 / HBM_PEAK_GBS if c.seconds_expand_kernel > 0 else None,
           "launch_rounds": int(c.levels)}
    e.close()
    if cpu_seconds > 0:
        try:
            out["cpu_baseline"] = synthetic_cpu(st, cpu_seconds, cpu_threads)
        except Exception as ex:  # the GPU figure must survive
            out["cpu_baseline"] = {"error": f"{type(ex).__name__}: {ex}"}
    return out


# Injection points for tests/bench_cpu_entry.py (the launcher / N-rank pipeline test on a box without GPUs): a stand-in
# engine class, the process-group backend that goes with it, the script the launcher starts per rank, and a label that
# ends up in the output line.  bench.py itself never sets them: run as `python bench.py` the engine is the HIP engine.
HOOKS = {"engine": None, "backend": "nccl", "entry": Path(__file__).resolve(), "label": None}


def make_engine_factory(st, local_rank):
    if HOOKS["engine"] is not None:
        return lambda model, **kw: HOOKS["engine"](model, **kw)
    return lambda model, **kw: st.Engine(model, device=local_rank, **kw)


def sharded_workload(st, sh, dist, torch, make_engine, model, rank, world, dev, label, time_limit_s=0.0, repeats=2, expect=None, knobs=None, run=None):
    """One workload through solve_sharded on all ranks (N>1 line): whole-job nodes/s with per-rank search nodes,
    donated / adopted open nodes, supersteps and the wall time spent in collectives."""
    eng = make_engine(model, rank=rank, world=world, time_limit_s=time_limit_s,
                      flags=(st.F_STEPPED if world == 1 else 0) | st.F_NO_EXPORT)
    best = None
    for _ in range(repeats):
        stats = {}
        dist.barrier()
        t0 = time.perf_counter()
        if run is not None:
            run(eng, stats)  # the native loop (stcsp_engine_solve_sharded)
        else:
            sh.solve_sharded(eng, rank, world, dev, stats=stats, **(knobs or {}))
        if dev.type == "cuda":
            torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        c = eng.counters()
        mine = torch.tensor([dt, c.search_nodes, c.fails, c.leaves, stats["nodes_donated"], stats["nodes_adopted"],
                             stats["seconds_collectives"], c.seconds_search], dtype=torch.float64, device=dev)
        allr = torch.empty(world * mine.numel(), dtype=torch.float64, device=dev)
        dist.all_gather_into_tensor(allr, mine)
        rows = allr.view(world, -1).tolist()
        wall = max(r[0] for r in rows)
        if best is None or wall < best["seconds"]:
            nodes = int(sum(r[1] for r in rows))
            best = {"workload": label, "seconds": wall, "nodes": nodes, "nodes_per_s": nodes / wall,
                    "rank_search_nodes": [int(r[1]) for r in rows], "fails": int(sum(r[2] for r in rows)), "leaves": int(sum(r[3] for r in rows)),
                    "nodes_donated": [int(r[4]) for r in rows], "nodes_adopted": [int(r[5]) for r in rows],
                    "supersteps": stats["supersteps"], "ms_in_collectives": [r[6] * 1e3 for r in rows],
                    "time_box_s": time_limit_s or None}
    if expect is not None and not time_limit_s:
        best["expected_nodes"] = expect
        best["nodes_ok"] = best["nodes"] == expect
    eng.close()
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default=WORKLOAD)
    ap.add_argument("--batch", type=int, default=0)
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-threads", type=int, default=0, help="workers of the all-cores CPU legs (default: min(affinity mask, cgroup CPU quota, 16 = the pool's share per GPU))")
    ap.add_argument("--no-other-workloads", action="store_true")
    ap.add_argument("--synthetic-seconds", type=float, default=2.0)
    ap.add_argument("--synthetic-cpu-seconds", type=float, default=5.0)
    ap.add_argument("--synthetic-shape", default="", help="n,d,point constraints,next constraints,seed (default: BASELINE config 4 = 64,32,602,6,20261003)")
    ap.add_argument("--scalable-workload", default="partialorder_18",
                    help="N>1 (and --stepped) only: a second, larger instance through the sharded pipeline, outside the timed region")
    ap.add_argument("--budget-rounds", type=int, default=8, help="sharded runs: launch rounds per superstep once a rank holds enough open nodes to share")
    ap.add_argument("--share-per-rank", type=int, default=64, help="sharded runs: ... 'enough' = this many open nodes per rank")
    ap.add_argument("--python-driver", action="store_true",
                    help="sharded runs: drive the supersteps from stcsp-solver_amd/sharded.py over torch.distributed instead of the "
                         "native loop (stcsp_engine_solve_sharded over the RCCL transport of libstcsp_rccl.so)")
    ap.add_argument("--stepped", action="store_true",
                    help="N=1 only: run the sharded pipeline (size-1 RCCL group, STCSP_F_STEPPED) -- the N>1 code path on one GPU")
    args = ap.parse_args()

    if args.synthetic_shape:
        global SYNTH_SHAPE
        SYNTH_SHAPE = tuple(int(x) for x in args.synthetic_shape.split(","))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args, sys.argv[1:]))

    import torch
    import torch.distributed as dist

    st = importlib.import_module("stcsp-solver_amd")
    golden = json.loads((REPO / "tests" / "golden" / "reference_golden.json").read_text())
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    cpu_test = HOOKS["engine"] is not None
    if not cpu_test and not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    if cpu_test:
        dev = torch.device("cpu")
    else:
        torch.cuda.set_device(local_rank)
        dev = torch.device(f"cuda:{local_rank}")
    stepped = world > 1 or args.stepped or cpu_test
    if stepped:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29519")
        if cpu_test:
            dist.init_process_group(HOOKS["backend"], rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    # a fresh checkout has no built libraries: rank 0 compiles them once, the others wait
    if rank == 0 and (not (st.CSRC / "libstcsp_hip.so").exists() or not (REPO / "oracle" / "libstcsp_oracle.so").exists()):
        import __graft_entry__
        __graft_entry__.build()
    if world > 1:
        dist.barrier()

    make_engine = make_engine_factory(st, local_rank)
    model = st.Model.from_name(args.workload)
    flags = st.F_PROFILE | (st.F_STEPPED if stepped and world == 1 else 0)
    eng = make_engine(model, rank=rank, world=world, flags=flags, batch_nodes=args.batch)
    sh = importlib.import_module("stcsp-solver_amd.sharded") if stepped else None
    sstats = {}
    knobs = dict(budget_rounds=args.budget_rounds, share_per_rank=args.share_per_rank)
    # Sharded runs of the HIP engine: the superstep loop runs INSIDE the engine library (include/stcsp_sharded.h) over the RCCL
    # transport -- one more communicator next to torch's, its unique id broadcast through the process group; torch.distributed
    # is left with the barriers around the timed region and the final gather. (--python-driver: the loop of sharded.py.)
    native = stepped and not cpu_test and not args.python_driver
    transport = None
    native_note = None
    if native:
        # every rank has to end up with the same driver: the ranks agree (all-reduce) on whether the RCCL transport came up
        # everywhere; if it did not on some rank, all of them take the torch-driven loop and the line says why
        err = None
        try:
            have_id = torch.ones(1, dtype=torch.int32, device=dev)
            uid = torch.zeros(st.RCCL_ID_BYTES, dtype=torch.uint8, device=dev)
            if rank == 0:
                try:
                    uid = torch.frombuffer(bytearray(st.rccl_unique_id()), dtype=torch.uint8).to(dev)
                except Exception as ex:  # noqa: BLE001
                    err, have_id[0] = ex, 0
            dist.broadcast(have_id, 0)
            dist.broadcast(uid, 0)
            if int(have_id.item()):
                transport = st.RcclTransport(bytes(uid.cpu().tolist()), rank, world, local_rank)
        except Exception as ex:  # noqa: BLE001
            err = err or ex
        ok = torch.tensor([1 if transport is not None else 0], dtype=torch.int32, device=dev)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if not int(ok.item()):
            native = False
            native_note = f"the RCCL transport of libstcsp_rccl.so did not come up on every rank ({type(err).__name__ if err else 'a peer'}: {err}): torch-driven loop"
            if transport is not None:
                transport.close()
                transport = None

    def run_native(engine, stats):
        stats.update(st.solve_sharded_native(engine, transport.ptr, **knobs))

    def one_step():
        """One full solve: search + ok-fixpoint + compaction + copy to the host (per shard when sharded)."""
        if not stepped:
            res = eng.solve()
            return res.counters, res
        if native:
            run_native(eng, sstats)
        else:
            sh.solve_sharded(eng, rank, world, dev, stats=sstats, **knobs)
        res = eng.export()  # this shard's states and raw edges on the host (the merge on rank 0 is outside the step)
        return eng.counters(), res

    def barrier():
        if stepped:
            dist.barrier()
        if dev.type == "cuda":
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        one_step()
    barrier()
    t0 = time.perf_counter()
    nodes = leaves = revs = evals = wrevs = sweeps = 0
    k_time = s_search = s_export = s_coll = 0.0
    k_launches = 0
    supersteps = donated = adopted = 0
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
        if stepped:
            s_coll += sstats.get("seconds_collectives", 0.0)
            supersteps += sstats.get("supersteps", 0)
            donated += sstats.get("nodes_donated", 0)
            adopted += sstats.get("nodes_adopted", 0)
    barrier()
    elapsed = time.perf_counter() - t0
    levels = c.levels
    rank_rows = None
    if stepped:
        t = torch.tensor([elapsed, s_search], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, s_search = t.tolist()
        mine = torch.tensor([nodes, leaves, donated, adopted, int(s_coll * 1e6)], dtype=torch.int64, device=dev)
        allr = torch.empty(world * mine.numel(), dtype=torch.int64, device=dev)
        dist.all_gather_into_tensor(allr, mine)
        rank_rows = allr.view(world, -1).tolist()
        nodes = sum(r[0] for r in rank_rows)
        leaves = sum(r[1] for r in rank_rows)

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
    if not stepped:
        # the streamed export (edge log shipped while the search runs, ordered by kernel boundaries: DESIGN 4.4) against
        # the compacting export of one more solve of the same engine (everything copied after the search has ended)
        prev = os.environ.get("STCSP_STREAM_EXPORT")
        if prev is not None and prev.strip() == "0":
            # the caller switched streaming off for the whole run (tools/profile_*.sh): both solves took the compacting path
            check["export_paths"] = {"skipped": "STCSP_STREAM_EXPORT=0 was set by the caller: the timed steps already used the compacting export"}
        else:
            os.environ["STCSP_STREAM_EXPORT"] = "0"
            try:
                r2 = eng.solve()
                a2 = eng.automaton(r2).traverse().renumber()
                check["export_paths"] = {"streamed_sha256": check.get("canonical_sha256"), "compacting_sha256": a2.canonical_sha256(),
                                         "equal": a2.canonical_sha256() == check.get("canonical_sha256")}
                parity_ok = parity_ok and check["export_paths"]["equal"]
            finally:
                if prev is None:
                    del os.environ["STCSP_STREAM_EXPORT"]
                else:
                    os.environ["STCSP_STREAM_EXPORT"] = prev

    # N>1 (and --stepped): workloads that CAN scale, through the same sharded pipeline, outside the timed region
    scalable = {}
    if stepped and not args.no_other_workloads:
        try:
            if args.scalable_workload:
                exp = {"partialorder_16": 6112188, "partialorder_18": 27678644}.get(args.scalable_workload)
                scalable[args.scalable_workload] = sharded_workload(st, sh, dist, torch, make_engine, st.Model.from_name(args.scalable_workload), rank, world,
                                                                    dev, args.scalable_workload, repeats=2, expect=exp, knobs=knobs, run=run_native if native else None)
            if args.synthetic_seconds > 0:
                n, d, mp, s_, seed = SYNTH_SHAPE
                scalable["synthetic_64x32" if SYNTH_SHAPE == SYNTH else "synthetic"] = sharded_workload(st, sh, dist, torch, make_engine, synthetic_model(st), rank, world, dev,
                                                               f"synthetic {n} vars x |D|={d}, {mp} + {s_} constraints, seed {seed}",
                                                               time_limit_s=args.synthetic_seconds, repeats=2, knobs=knobs, run=run_native if native else None)
        except (sh.ShardedSolveError, st.StcspError) as ex:  # agreed on every rank (python driver / native loop)
            scalable["error"] = f"{type(ex).__name__}: {ex}"
        # a scalable workload that failed, or expanded another number of nodes than the unsharded search does, fails the run
        if "error" in scalable or any(isinstance(v, dict) and v.get("nodes_ok") is False for v in scalable.values()):
            parity_ok = False

    if rank == 0:
        S = res.sig_len
        ab, b_node, b_leaf = alg_bytes(model, S, nodes, leaves)
        per_launch_bytes = ab / max(k_launches * world, 1)  # (k_launches: this rank's; every rank runs its own launches)
        avg_launch_s = k_time / max(k_launches, 1)
        achieved = per_launch_bytes / avg_launch_s / 1e9 if avg_launch_s > 0 else 0.0
        # HBM traffic of the same kernel: FETCH_SIZE + WRITE_SIZE of every k_expand dispatch of one solve from this round's
        # separate --pmc passes (tools/profile_r04.sh; a PMC run cannot be part of the timed bench), gfx950-corrected, per
        # launch like `achieved`; null when the committed profile was not taken on THIS engine source (hash mismatch)
        traffic = traffic_note = None
        tf = REPO / "profiles" / "r04_p14_traffic.json"
        if args.workload == WORKLOAD and not stepped and tf.exists() and k_launches:
            tj = json.loads(tf.read_text())
            if tj.get("engine_source_sha") == st.engine_source_sha():
                traffic = tj["hbm_bytes_per_solve_corrected"] / (k_launches / args.steps)
                traffic_note = ("profiles/r04_p14_traffic.json (engine source sha %s): separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over this "
                                "command, gfx950-corrected, divided by this run's launches per solve" % tj["engine_source_sha"])
            else:
                traffic_note = f"profiles/r04_p14_traffic.json was measured on another engine source ({tj.get('engine_source_sha')}): not quoted"
        p = model.problem.contents
        cfg = {"workload": f"{args.workload}.csp (token-identical to the reference's examples/{args.workload}.csp, regenerated by stcsp-solver_amd/instances.py; "
                           f"{p.n_vars} vars incl. aux, prefix K={p.prefix_k}, whole frontier resident in HBM)",
               "timed_region": "solve entry -> raw automaton on the host (search, with the edge log and state keys streamed to pinned host arrays "
                               "on a second stream while it runs, + ok-fixpoint + the last chunks; compaction + D2H when a state failed)",
               "nodes_per_step": nodes // args.steps, "leaves_per_step": leaves // args.steps,
               "launch_rounds_per_step": int(levels),
               "per_node": {"item_revisions": revs / max(nodes, 1) * (world if stepped else 1), "tuple_evaluations": evals / max(nodes, 1) * (world if stepped else 1),
                            "wavefront_revisions": wrevs / max(nodes, 1) * (world if stepped else 1), "sweeps": sweeps / max(nodes, 1) * (world if stepped else 1)},
               "sharding": "none" if not stepped else f"state-owner x{world}"}
        if stepped:
            cfg["superstep_driver"] = ("native: stcsp_engine_solve_sharded over the RCCL transport (libstcsp_rccl.so: ncclAllGather for the count table, "
                                       "grouped ncclSend / ncclRecv on the engine's stream)" if native else "python: stcsp-solver_amd/sharded.py over torch.distributed")
            if native_note:
                cfg["superstep_driver"] += " -- " + native_note
            if world > 1:
                cfg["multi_gpu_note"] = "the builder's pool has one GPU per box: every N > 1 figure comes from the driver's run, none was measured while building"
        if cpu_test:
            cfg["engine"] = HOOKS["label"] or "TEST ONLY: injected stand-in engine (not a measurement of the product)"
        if stepped:
            cfg["sharded"] = {"rank_search_nodes_per_step": [r[0] // args.steps for r in rank_rows],
                              "nodes_donated": [r[2] for r in rank_rows], "nodes_adopted": [r[3] for r in rank_rows],
                              "supersteps_per_step": supersteps / args.steps,
                              "ms_in_collectives_per_step": [r[4] / 1e3 / args.steps for r in rank_rows],
                              "scalable_workloads": scalable}
        if world == 1 and not stepped and not args.no_other_workloads:
            others = {}
            for name in ("digitinvader9", "juggling_b5_f6"):
                try:
                    others[name] = other_workload(st, golden, name, local_rank)
                    parity_ok = parity_ok and others[name]["parity_ok"]
                except Exception as ex:  # the headline line must survive
                    others[name] = {"error": f"{type(ex).__name__}: {ex}"}
                    parity_ok = False
            try:
                others["synthetic_64x32"] = synthetic_workload(st, local_rank, args.synthetic_seconds,
                                                               0.0 if args.no_cpu_baseline else args.synthetic_cpu_seconds,
                                                               args.cpu_threads or host_cpu_share()[1])
            except Exception as ex:
                others["synthetic_64x32"] = {"error": f"{type(ex).__name__}: {ex}"}
            cfg["other_workloads"] = others
            # the same workload from a process WITHOUT PyTorch's HIP runtime (the runtime a C++ host linked against /opt/rocm
            # gets: INTEGRATION.md's binding, the stcsp CLI); outside the timed region, this process idle meanwhile
            try:
                import subprocess
                r_ = subprocess.run([sys.executable, str(REPO / "tools" / "bench_no_torch.py"), args.workload, str(args.steps), str(min(args.warmup, 5))],
                                    capture_output=True, text=True, timeout=300)
                line = [ln for ln in r_.stdout.splitlines() if ln.startswith("{")]
                cfg["without_pytorch_runtime"] = json.loads(line[-1]) if line else {"error": (r_.stderr or r_.stdout)[-400:]}
                if line and not cfg["without_pytorch_runtime"].get("parity_ok", False) and args.workload in golden:
                    parity_ok = False
            except Exception as ex:  # noqa: BLE001 -- the headline line must survive
                cfg["without_pytorch_runtime"] = {"error": f"{type(ex).__name__}: {ex}"}
        out = {
            "metric": f"search-tree nodes/sec on {args.workload}.csp",
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
            "data": "reference example (regenerated)",
            "config": cfg,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_note,
                         "kernel": "k_expand",
                         "alg_bytes_per_launch": per_launch_bytes, "avg_launch_us": avg_launch_s * 1e6,
                         "launches": k_launches, "bytes_per_node": b_node, "bytes_per_leaf": b_leaf,
                         "kernel_time_share": k_time / elapsed if elapsed > 0 else None},
            "search_only_nodes_per_s": nodes / s_search if s_search > 0 else None,
            "search_ms": s_search / args.steps * 1e3,
            "export_ms": s_export / args.steps * 1e3,
            "parity": check,
        }
        if not args.no_cpu_baseline and world == 1 and not stepped:
            out["cpu_baseline"] = cpu_baseline(st, args.workload, args.cpu_seconds, args.cpu_threads or host_cpu_share()[1])
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
