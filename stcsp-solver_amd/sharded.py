"""Multi-shard driver: one process per GPU, the open search frontier sharded by state owner.

The reference is a single-threaded program (no threads, no MPI/NCCL), so
there is no reference call pattern to follow here.  The unit of data parallelism is the open
search node; the only shared structure is the automaton's state table, sharded by
owner = hash(constraint set, signature) % world.  Per superstep every shard

  1. expands its own open nodes until only leaf successor candidates remain
     (`stcsp_engine_expand_local`; bisection children never leave the GPU that produced them),
  2. exchanges the candidates with an all-to-all-v (RCCL over xGMI: `torch.distributed` backend
     "nccl" is RCCL on ROCm; every peer is one direct xGMI hop, messages are KB..MB, so this is
     latency- not bandwidth-bound and one exchange per *time step* of the automaton, not per
     tree level, is what keeps it off the critical path),
  3. commits what it received: lookup-or-insert in its table shard, edge log, new open nodes
     (`stcsp_engine_commit`),

until no shard has open nodes or candidates.  A tiny all-gather of counts doubles as the
termination test; constraint-set definitions (a handful per run) are all-gathered when a shard
meets a new one.  PyTorch is used for device buffers and the process group only.
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


def solve_sharded(engine, rank: int, world: int, device: torch.device, stage_through_host: bool = False,
                  max_rounds: int = 1_000_000):
    """Run the sharded search to completion on this rank.  `engine` is an engine-shaped object
    (the HIP `Engine` created with rank/world; tests substitute the CPU frontier model).
    `device` is where the engine's candidate buffers live ("cuda:N" or "cpu");
    `stage_through_host` moves the all-to-all through CPU tensors (gloo with a GPU engine).
    Returns the number of supersteps."""
    csw = engine.candidate_bytes() // 4
    engine.begin()
    comm_dev = torch.device("cpu") if (stage_through_host or device.type == "cpu") else device
    # every shard starts from the same registry (the model's own sets, plus whatever earlier solves
    # on these engines exchanged), so the definitions only travel once somebody's count moves
    last_sets = [engine.sets_count()] * world
    rounds = 0
    while True:
        rounds += 1
        if rounds > max_rounds:
            raise RuntimeError("sharded solve did not terminate")
        left = engine.expand_local()
        outs = [engine.outbox(p) for p in range(world)]  # (ptr, record count) per peer
        n_sets = engine.sets_count()
        meta = torch.tensor([c for _, c in outs] + [left, n_sets], dtype=torch.int64, device=comm_dev)
        gathered = torch.empty(world * (world + 2), dtype=torch.int64, device=comm_dev)
        dist.all_gather_into_tensor(gathered, meta)
        flat = gathered.tolist()  # one device-to-host copy for the whole table
        allmeta = [flat[r * (world + 2):(r + 1) * (world + 2)] for r in range(world)]
        sets_now = [m[world + 1] for m in allmeta]
        if sets_now != last_sets:  # somebody met a new constraint set: everyone learns all of them
            blobs = [None] * world
            dist.all_gather_object(blobs, engine.sets_blob())
            for r, b in enumerate(blobs):
                if r != rank:
                    engine.sets_import(b)
            last_sets = sets_now
        send_counts = [c for _, c in outs]
        recv_counts = [allmeta[p][rank] for p in range(world)]
        if sum(send_counts) == 0:
            send = torch.empty(0, dtype=torch.int32, device=device)
        elif all(outs[p + 1][0] == outs[p][0] + outs[p][1] * csw * 4 for p in range(world - 1)):
            send = _view(outs[0][0], sum(send_counts) * csw, device)  # the engine packs peers back to back
        else:
            send = torch.cat([_view(p, c * csw, device) for p, c in outs])
        if comm_dev != device:
            send = send.to(comm_dev)
        recv = torch.empty(sum(recv_counts) * csw, dtype=torch.int32, device=comm_dev)
        dist.all_to_all_single(recv, send, [c * csw for c in recv_counts], [c * csw for c in send_counts])
        if comm_dev != device:
            recv = recv.to(device)
        if device.type == "cuda":
            torch.cuda.synchronize(device)
        engine.commit(recv.data_ptr() if recv.numel() else 0, sum(recv_counts))
        total_open = sum(m[world] for m in allmeta)
        total_cands = sum(sum(m[:world]) for m in allmeta)
        if total_open == 0 and total_cands == 0:
            break
    engine.finish()
    return rounds


def result_to_numpy(res) -> dict:
    """Copy a stcsp_result into picklable numpy arrays (to gather shards on one rank)."""
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
    for k in ("n_states", "n_edges", "sig_len", "n_vars", "n_sig_vars", "n_until", "n_until_cons", "n_constraint_sets",
              "root_final", "truncated"):
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


def gather_and_merge(stcsp, engine, rank: int, world: int):
    """Export every shard, gather on rank 0 and merge (host post-processing runs there, like the
    reference's single process).  Returns (merge handle, merged Result) on rank 0, else None."""
    mine = result_to_numpy(engine.export())
    parts = [None] * world
    dist.all_gather_object(parts, mine)
    if rank != 0:
        return None
    results = [numpy_to_result(p, stcsp.Result, stcsp.Counters) for p in parts]
    h, merged = stcsp.merge_shards(results)
    merged._keepalive = (parts, results)
    return h, merged
