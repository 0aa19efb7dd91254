"""Multi-shard driver: one process per GPU, the open search frontier sharded across them.

The reference is a single-threaded program (no threads, no MPI/NCCL), so
there is no reference call pattern to follow here.  The unit of data parallelism is the open
search node (one call of solverSolveRe, reference src/solveralgorithm.cpp:733); the only shared
structure is the automaton's state table, sharded by owner = hash(constraint set, signature) % world.
Per superstep every shard

  1. expands its own open nodes (`stcsp_engine_expand_local`) -- until its frontier is dry, or, when
     it holds enough nodes to share, for a bounded number of launch rounds (`set_expand_budget`),
  2. all-gathers a small table of counts (= termination test, load picture, outbox sizes),
  3. REDISTRIBUTES open nodes when the load picture is lopsided: the shards above the mean donate
     their oldest (shallowest) open nodes to the shards below it (`donate` / `adopt`; the branch case
     of the search, src/solveralgorithm.cpp:911-939, is the work that moves).  The same rule scatters
     the work at the start: the root lives on one shard, which expands for a few rounds and then
     shares what it has,
  4. exchanges the leaf successor candidates with an all-to-all-v (RCCL over xGMI: `torch.distributed`
     backend "nccl" is RCCL on ROCm; every peer is one direct xGMI hop, messages are KB..MB, so this
     is latency- not bandwidth-bound) and commits what it received: lookup-or-insert in its table
     shard, edge log, new open nodes (`stcsp_engine_commit`),

until no shard has open nodes or candidates.  Constraint-set definitions (a handful per run) travel
as int32 tensors when some shard's count moves; the final gather of the shards' automata on rank 0
uses tensor collectives too.  PyTorch is used for device buffers and the process group only.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch
import torch.distributed as dist


class _DevWords:
    """Zero-copy view of engine-owned device memory for torch (no torch types cross the ABI)."""

    def __init__(self, ptr: int, n_words: int):
        self.__cuda_array_interface__ = {"shape": (n_words,), "typestr": "<i4", "data": (ptr, False), "version": 2}


def _view(ptr: int, n_words: int, device: torch.device) -> torch.Tensor:
    if n_words == 0:
        return torch.empty(0, dtype=torch.int32, device=device)
    if device.type == "cuda":
        return torch.as_tensor(_DevWords(ptr, n_words), device=device)
    buf = (C.c_int32 * n_words).from_address(ptr)
    return torch.frombuffer(buf, dtype=torch.int32)


def plan_transfers(left, min_gain: int = 1):
    """Deterministic redistribution plan from the all-gathered open-node counts (every rank computes the
    same one): shards above the mean give their surplus to the shards below it, largest surplus to
    largest deficit first.  Returns send[i][j] = nodes rank i ships to rank j (all zero when the
    frontier is balanced enough: the poorest shard has at least half the mean, or there is less than
    `min_gain` to move)."""
    world = len(left)
    send = [[0] * world for _ in range(world)]
    total = sum(left)
    if world == 1 or total == 0:
        return send
    mean = total // world
    if mean == 0 or min(left) * 2 >= mean:
        return send
    surplus = sorted(((left[r] - mean, r) for r in range(world) if left[r] > mean), reverse=True)
    deficit = sorted(((mean - left[r], r) for r in range(world) if left[r] < mean), reverse=True)
    si = di = 0
    surplus = [list(x) for x in surplus]
    deficit = [list(x) for x in deficit]
    while si < len(surplus) and di < len(deficit):
        n = min(surplus[si][0], deficit[di][0])
        if n >= min_gain:
            send[surplus[si][1]][deficit[di][1]] += n
        surplus[si][0] -= n
        deficit[di][0] -= n
        if surplus[si][0] == 0:
            si += 1
        if deficit[di][0] == 0:
            di += 1
    return send


def _exchange(send: torch.Tensor, send_counts, recv_counts, words: int, comm_dev, device):
    """all-to-all-v of fixed-size records (`words` int32 each)."""
    if comm_dev != device:
        send = send.to(comm_dev)
    recv = torch.empty(sum(recv_counts) * words, dtype=torch.int32, device=comm_dev)
    dist.all_to_all_single(recv, send, [c * words for c in recv_counts], [c * words for c in send_counts])
    if comm_dev != device:
        recv = recv.to(device)
    return recv


def _exchange_sets(engine, rank: int, world: int, blob_lens, comm_dev):
    """Everyone learns every shard's constraint sets: padded all-gather of the int32 blobs."""
    err = None
    try:
        blob = engine.sets_blob()
    except Exception as ex:  # noqa: BLE001 -- take part in the collective first, raise afterwards (the caller agrees on it)
        err, blob = ex, []
    mine = torch.tensor(blob, dtype=torch.int32, device=comm_dev)
    width = max(max(blob_lens), mine.numel(), 1)
    padded = torch.zeros(width, dtype=torch.int32, device=comm_dev)
    padded[: mine.numel()] = mine
    everyone = torch.empty(world * width, dtype=torch.int32, device=comm_dev)
    dist.all_gather_into_tensor(everyone, padded)
    everyone = everyone.cpu()
    if err is not None:
        raise err
    for r in range(world):
        if r != rank:
            engine.sets_import(everyone[r * width: r * width + blob_lens[r]].tolist())


class ShardedSolveError(RuntimeError):
    """Raised on EVERY rank of a sharded solve when any rank's engine failed (the ranks agree on the status before
    each data exchange, so a failure never leaves the peers blocked in a collective)."""


def _agree(ok: bool, comm_dev, what: str, local_error: Exception | None, rank: int):
    """One tiny all-reduce(MAX) of a status word: every rank leaves with the same verdict."""
    t = torch.tensor([0 if ok else 1], dtype=torch.int32, device=comm_dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    if int(t.item()):
        if local_error is not None:
            raise ShardedSolveError(f"rank {rank}: {what} failed: {local_error}") from local_error
        raise ShardedSolveError(f"rank {rank}: a peer failed in {what}")


def solve_sharded(engine, rank: int, world: int, device: torch.device, stage_through_host: bool = False,
                  max_rounds: int = 1_000_000, budget_rounds: int = 8, share_per_rank: int = 64, stats: dict | None = None):
    """Run the sharded search to completion on this rank.  `engine` is an engine-shaped object
    (the HIP `Engine` created with rank/world; tests substitute the CPU frontier model).
    `device` is where the engine's record buffers live ("cuda:N" or "cpu");
    `stage_through_host` moves the exchanges through CPU tensors (gloo with a GPU engine).
    `budget_rounds` / `share_per_rank`: expand_local hands control back after that many launch rounds
    once it holds at least share_per_rank * world open nodes (so that there is something to share).
    Returns the number of supersteps; `stats` (optional dict) receives the redistribution totals and
    the wall time spent in collectives.

    Failure protocol (a rank must never raise alone while its peers sit in a collective): every engine
    call is caught; the error word travels with the superstep's count table (meta), with a status
    all-reduce between `donate` and the node exchange, and with a last all-reduce after `finish` -- so
    all ranks raise ShardedSolveError together, at the same point."""
    import time
    csw = engine.candidate_bytes() // 4
    nsw = engine.node_bytes() // 4
    comm_dev = torch.device("cpu") if (stage_through_host or device.type == "cpu") else device
    pending: Exception | None = None  # an engine error of this rank that the peers have not heard of yet
    t_coll = 0.0

    def guarded(fn, *a):
        nonlocal pending
        if pending is not None:
            return None
        try:
            return fn(*a)
        except Exception as ex:  # noqa: BLE001 -- anything: the peers must be told
            pending = ex
            return None

    guarded(engine.set_expand_budget, budget_rounds if world > 1 else 0, share_per_rank * world)
    guarded(engine.begin)
    # every shard starts from the same registry (the model's own sets, plus whatever earlier solves
    # on these engines exchanged), so the definitions only travel once somebody's count moves
    last_sets = [guarded(engine.sets_count) or 0] * world
    rounds = 0
    moved = received = 0
    keep = None
    W = world + 4  # per-rank row of the meta table: candidates per peer, open nodes, set count, set blob length, status
    while True:
        rounds += 1
        if rounds > max_rounds and pending is None:
            pending = RuntimeError("sharded solve did not terminate")
        left = guarded(engine.expand_local)
        outs = [guarded(engine.outbox, p) for p in range(world)]  # (ptr, record count) per peer
        n_sets = guarded(engine.sets_count)
        blob_len = guarded(lambda: len(engine.sets_blob()))
        if pending is not None:
            row = [0] * world + [0, 0, 0, 1]
        else:
            row = [c for _, c in outs] + [left, n_sets, blob_len, 0]
        t0 = time.perf_counter()
        meta = torch.tensor(row, dtype=torch.int64, device=comm_dev)
        gathered = torch.empty(world * W, dtype=torch.int64, device=comm_dev)
        dist.all_gather_into_tensor(gathered, meta)
        flat = gathered.tolist()  # one device-to-host copy for the whole table
        t_coll += time.perf_counter() - t0
        allmeta = [flat[r * W:(r + 1) * W] for r in range(world)]
        if any(m[world + 3] for m in allmeta):
            bad = [r for r in range(world) if allmeta[r][world + 3]]
            if pending is not None:
                raise ShardedSolveError(f"rank {rank}: engine failed in superstep {rounds}: {pending}") from pending
            raise ShardedSolveError(f"rank {rank}: rank(s) {bad} failed in superstep {rounds}")
        sets_now = [m[world + 1] for m in allmeta]
        if sets_now != last_sets:  # somebody met a new constraint set: everyone learns all of them
            t0 = time.perf_counter()
            err = None
            try:
                _exchange_sets(engine, rank, world, [m[world + 2] for m in allmeta], comm_dev)
            except Exception as ex:  # noqa: BLE001 -- sets_import failed after the all-gather: agree below
                err = ex
            _agree(err is None, comm_dev, "the constraint-set exchange", err, rank)
            t_coll += time.perf_counter() - t0
            last_sets = [engine.sets_count()] * world  # after the import every shard knows the union
        # ---- frontier redistribution (open nodes): same plan on every rank
        lefts = [m[world] for m in allmeta]
        plan = plan_transfers(lefts, min_gain=1)
        n_send = plan[rank]
        n_recv = [plan[p][rank] for p in range(world)]
        nodes_recv = None
        if any(any(row) for row in plan):
            want = sum(n_send)
            err = None
            ptr = got = 0
            if want:
                try:
                    ptr, got = engine.donate(want)
                    if got != want:  # (cannot happen: `left` counted these nodes a moment ago)
                        raise RuntimeError(f"planned to donate {want} open nodes, the engine gave {got}")
                except Exception as ex:  # noqa: BLE001
                    err = ex
            t0 = time.perf_counter()
            _agree(err is None, comm_dev, "donate", err, rank)  # before the exchange: nobody waits for a rank that raised
            send = _view(ptr, got * nsw, device)
            nodes_recv = _exchange(send, n_send, n_recv, nsw, comm_dev, device)
            t_coll += time.perf_counter() - t0
            moved += got
            received += sum(n_recv)
        # ---- leaf successor candidates
        send_counts = [c for _, c in outs]
        recv_counts = [allmeta[p][rank] for p in range(world)]
        if sum(send_counts) == 0:
            send = torch.empty(0, dtype=torch.int32, device=device)
        elif all(outs[p + 1][0] == outs[p][0] + outs[p][1] * csw * 4 for p in range(world - 1)):
            send = _view(outs[0][0], sum(send_counts) * csw, device)  # the engine packs peers back to back
        else:
            send = torch.cat([_view(p, c * csw, device) for p, c in outs])
        t0 = time.perf_counter()
        recv = _exchange(send, send_counts, recv_counts, csw, comm_dev, device)
        if device.type == "cuda":
            torch.cuda.current_stream(device).synchronize()  # (not the whole device: the engine's export streams keep copying)
        t_coll += time.perf_counter() - t0
        # an error from here on is reported with the next superstep's meta row (or the final agreement)
        guarded(engine.commit, recv.data_ptr() if recv.numel() else 0, sum(recv_counts))
        if nodes_recv is not None and nodes_recv.numel():
            guarded(engine.adopt, nodes_recv.data_ptr(), sum(n_recv))
        total_open = sum(lefts)
        total_cands = sum(sum(m[:world]) for m in allmeta)
        if total_open == 0 and total_cands == 0:
            break
        keep = (recv, nodes_recv)  # the engine reads the received records asynchronously: keep them alive for the superstep
    del keep
    guarded(engine.finish)
    t0 = time.perf_counter()
    _agree(pending is None, comm_dev, "the last superstep", pending, rank)
    t_coll += time.perf_counter() - t0
    if stats is not None:
        stats.update(supersteps=rounds, nodes_donated=moved, nodes_adopted=received, seconds_collectives=t_coll)
    return rounds


# ------------------------------------------------------------------ final gather (tensor collectives)
_SCALARS = ("n_states", "n_edges", "sig_len", "n_vars", "n_sig_vars", "n_until", "n_until_cons", "n_constraint_sets",
            "root_final", "truncated")
_ARRAYS = (("state_cid", np.int32), ("state_sig", np.int32), ("state_fail", np.uint8), ("edge_src", np.int64),
           ("edge_dst", np.int64), ("edge_values", np.int32), ("var_is_signature", np.uint8))


def result_to_numpy(res) -> dict:
    """Copy a stcsp_result into numpy arrays."""
    ns, ne, sl, nv = res.n_states, res.n_edges, res.sig_len, res.n_vars
    as_np = lambda p, n, dt: np.ctypeslib.as_array(p, shape=(n,)).astype(dt, copy=True) if n else np.zeros(0, dt)  # noqa: E731
    d = dict(n_states=ns, n_edges=ne, sig_len=sl, n_vars=nv, n_sig_vars=res.n_sig_vars, n_until=res.n_until,
             n_until_cons=res.n_until_cons, n_constraint_sets=res.n_constraint_sets, root_final=res.root_final,
             truncated=res.truncated, counters=res.counters.as_dict(),
             state_cid=as_np(res.state_cid, ns, np.int32), state_sig=as_np(res.state_sig, ns * sl, np.int32),
             state_fail=as_np(res.state_fail, ns, np.uint8), edge_src=as_np(res.edge_src, ne, np.int64),
             edge_dst=as_np(res.edge_dst, ne, np.int64), edge_values=as_np(res.edge_values, ne * nv, np.int32),
             var_is_signature=as_np(res.var_is_signature, nv, np.uint8))
    return d


def numpy_to_result(d: dict, Result, Counters):
    """Inverse of result_to_numpy; the returned struct borrows the arrays in `d`."""
    r = Result()
    for k in _SCALARS:
        setattr(r, k, int(d[k]))
    ptr = lambda a, t: a.ctypes.data_as(C.POINTER(t))  # noqa: E731
    r.state_cid = ptr(d["state_cid"], C.c_int32)
    r.state_sig = ptr(d["state_sig"], C.c_int32)
    r.state_fail = ptr(d["state_fail"], C.c_uint8)
    r.edge_src = ptr(d["edge_src"], C.c_int64)
    r.edge_dst = ptr(d["edge_dst"], C.c_int64)
    r.edge_values = ptr(d["edge_values"], C.c_int32)
    r.var_is_signature = ptr(d["var_is_signature"], C.c_uint8)
    c = Counters()
    for k, v in d["counters"].items():
        setattr(c, k, v)
    r.counters = c
    return r


def _gather_bytes(buf: np.ndarray, rank: int, world: int, comm_dev):
    """Variable-length gather of a byte buffer on rank 0: sizes by all-gather, payload by dist.gather of
    equally padded tensors (only rank 0 receives)."""
    size = torch.tensor([buf.size], dtype=torch.int64, device=comm_dev)
    sizes = torch.empty(world, dtype=torch.int64, device=comm_dev)
    dist.all_gather_into_tensor(sizes, size)
    sizes = sizes.tolist()
    width = max(max(sizes), 1)
    mine = torch.zeros(width, dtype=torch.uint8, device=comm_dev)
    if buf.size:
        mine[: buf.size] = torch.from_numpy(buf).to(comm_dev)
    parts = [torch.empty(width, dtype=torch.uint8, device=comm_dev) for _ in range(world)] if rank == 0 else None
    dist.gather(mine, parts, dst=0)
    if rank != 0:
        return None
    return [parts[r][: sizes[r]].cpu().numpy() for r in range(world)]


def gather_and_merge(stcsp, engine, rank: int, world: int, device: torch.device | None = None):
    """Export every shard, gather on rank 0 and merge (host post-processing runs there, like the
    reference's single process).  Every shard's result travels as ONE byte tensor (a small int64
    header + its arrays back to back).  Returns (merge handle, merged Result) on rank 0, else None."""
    comm_dev = device if (device is not None and device.type == "cuda" and dist.get_backend() == "nccl") else torch.device("cpu")
    mine = result_to_numpy(engine.export())
    ckeys = sorted(mine["counters"])
    head = np.array([mine[k] for k in _SCALARS] + [len(ckeys)] + [mine[a].size for a, _ in _ARRAYS], dtype=np.int64)
    cvals = np.array([float(mine["counters"][k]) for k in ckeys], dtype=np.float64)
    payload = np.concatenate([head.view(np.uint8), cvals.view(np.uint8)] + [mine[a].view(np.uint8) for a, _ in _ARRAYS])
    parts = _gather_bytes(payload, rank, world, comm_dev)
    if rank != 0:
        return None
    dicts = []
    for raw in parts:
        raw = np.ascontiguousarray(raw)
        nh = len(_SCALARS) + 1 + len(_ARRAYS)
        h = raw[: nh * 8].view(np.int64)
        d = {k: int(h[i]) for i, k in enumerate(_SCALARS)}
        off = nh * 8
        nck = int(h[len(_SCALARS)])
        cv = raw[off: off + nck * 8].view(np.float64)
        off += nck * 8
        d["counters"] = {k: (float(v) if k.startswith("seconds") else int(v)) for k, v in zip(ckeys, cv)}
        for j, (a, dt) in enumerate(_ARRAYS):
            n = int(h[len(_SCALARS) + 1 + j])
            nbytes = n * np.dtype(dt).itemsize
            d[a] = raw[off: off + nbytes].copy().view(dt)
            off += nbytes
        dicts.append(d)
    results = [numpy_to_result(p, stcsp.Result, stcsp.Counters) for p in dicts]
    h, merged = stcsp.merge_shards(results)
    merged._keepalive = (dicts, results)
    return h, merged
