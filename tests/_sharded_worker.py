"""Worker for the multi-process tests of the sharded driver (one process per shard)."""
import ctypes as C
import importlib
import json
import os
import sys
from pathlib import Path

REPO = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(REPO))


def load_model(st, spec, prefix_k=2):
    """`spec`: an instance name, "file:<path to a .csp>" or "synth:n,d,m,s,seed" (instances.synthetic)."""
    if spec.startswith("file:"):
        return st.Model(text=Path(spec[5:]).read_text(), prefix_k=prefix_k)
    if spec.startswith("synth:"):
        n, d, m, s, seed = (int(x) for x in spec[6:].split(","))
        return st.Model(text=st.instances.synthetic(n, d, m, s, seed), prefix_k=prefix_k)
    return st.Model.from_name(spec, prefix_k=prefix_k)


class Faulty:
    """Fault injection for the failure-protocol tests: the wrapped engine raises at the n-th call of one method
    (STCSP_TEST_FAULT="method,rank[,n]"); everything else passes through."""

    def __init__(self, eng, method, nth):
        self._eng, self._method, self._nth, self._seen = eng, method, nth, 0

    def __getattr__(self, name):
        attr = getattr(self._eng, name)
        if name != self._method:
            return attr

        def boom(*a, **k):
            self._seen += 1
            if self._seen >= self._nth:
                raise RuntimeError(f"injected fault in {name}")
            return attr(*a, **k)

        return boom


def run(rank, world, port, name, backend_kind, out_path, prefix_k=2):
    # redistribution knobs for the tests (small instances must share early to exercise the path)
    budget = dict(budget_rounds=int(os.environ.get("STCSP_TEST_BUDGET_ROUNDS", "8")),
                  share_per_rank=int(os.environ.get("STCSP_TEST_SHARE", "64")))
    stats = {}
    import torch
    import torch.distributed as dist

    st = importlib.import_module("stcsp-solver_amd")
    sh = importlib.import_module("stcsp-solver_amd.sharded")
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    if backend_kind == "hip-nccl":
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda:0"))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    m = load_model(st, name, prefix_k)
    if backend_kind == "fmodel":
        # CPU stand-in for the HIP engine (tests only): oracle/frontier_model.cpp
        lib = C.CDLL(str(REPO / "oracle" / "libstcsp_oracle.so"))
        st.bind_engine_api(lib, "stcsp_fmodel")

        class FModel(st.EngineBase):
            _prefix = "stcsp_fmodel"

            def __init__(self, model, **o):
                super().__init__(lib, model, **o)

        eng = FModel(m, rank=rank, world=world)
        fault = os.environ.get("STCSP_TEST_FAULT")
        if fault:
            f = fault.split(",")
            if int(f[1]) == rank:
                eng = Faulty(eng, f[0], int(f[2]) if len(f) > 2 else 1)
        device = torch.device("cpu")
        rounds = sh.solve_sharded(eng, rank, world, device, stats=stats, **budget)
    elif backend_kind == "hip-nccl":
        # the production N>1 code path on one GPU: RCCL process group of size 1, candidates stay in
        # HBM (all_to_all_single on views of the engine's outbox), STCSP_F_STEPPED forces the
        # candidate/commit pipeline although world == 1
        import time
        assert world == 1
        eng = st.Engine(m, device=0, rank=0, world=1, flags=st.F_STEPPED)
        device = torch.device("cuda:0")
        rounds = sh.solve_sharded(eng, rank, world, device)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        rounds = sh.solve_sharded(eng, rank, world, device)
        torch.cuda.synchronize()
        stepped_ms = (time.perf_counter() - t0) * 1e3
    else:
        # the real HIP engine, every shard on GPU 0, all-to-all staged through host (gloo)
        eng = st.Engine(m, device=0, rank=rank, world=world, time_limit_s=float(os.environ.get("STCSP_TEST_TIME_LIMIT", "0")))
        device = torch.device("cuda:0")
        rounds = sh.solve_sharded(eng, rank, world, device, stage_through_host=True, stats=stats, **budget)
    # every rank's own search-node count (tensor collective)
    mine = torch.tensor([eng.counters().search_nodes, stats.get("nodes_donated", 0), stats.get("nodes_adopted", 0)], dtype=torch.int64)
    everyone = torch.empty(3 * world, dtype=torch.int64)
    if backend_kind == "hip-nccl":
        mine, everyone = mine.cuda(), everyone.cuda()
    dist.all_gather_into_tensor(everyone, mine)
    everyone = everyone.cpu().view(world, 3).tolist()
    merged = sh.gather_and_merge(st, eng, rank, world, device)
    if rank == 0:
        h, res = merged
        a = st.Automaton(m, res).traverse().renumber()
        out = dict(states=a.n_live_states, edges=a.n_live_edges, sha=a.canonical_sha256(), rounds=rounds,
                   table=res.n_states, dom=res.counters.dominance, nodes=res.counters.search_nodes,
                   sets=res.n_constraint_sets, fails=res.counters.fails, rank_nodes=[e[0] for e in everyone],
                   donated=[e[1] for e in everyone], adopted=[e[2] for e in everyone], canonical=a.canonical() if a.n_states <= 4096 else None)
        if backend_kind == "hip-nccl":
            out["stepped_ms"] = stepped_ms
        if os.environ.get("STCSP_TEST_ADVERSARIAL"):
            # solverSolve's tail with -a (solveralgorithm.cpp:972-985) on the merged automaton
            b = st.Automaton(m, res).traverse()
            out["adver1"] = b.adversarial(5)
            b.renumber()
            out["adv_empty"] = b.canonical().endswith("EMPTY\n")
        Path(out_path).write_text(json.dumps(out))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    rank, world, port, name, kind, out = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5], sys.argv[6]
    run(rank, world, port, name, kind, out)
