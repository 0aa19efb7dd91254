"""stcsp_engine_solve_sharded (include/stcsp_sharded.h): the superstep loop inside the engine library, no Python and no
torch between two bursts of k_expand.  The in-process transport (ranks = host threads, records by hipMemcpyAsync) runs
2, 3 and 4 shards on one GPU; the RCCL transport (libstcsp_rccl.so) runs with a communicator of size 1 -- all one GPU
allows.  Same checks as the torch-driven loop: the reference's canonical automaton and `dom`, node conservation, what is
donated is adopted, and a failing rank ends every rank."""
import pytest

from conftest import finish

pytestmark = pytest.mark.gpu

SHARE = dict(budget_rounds=1, share_per_rank=2)


def run_local(stcsp, model, world, knobs=None, **opts):
    engines = [stcsp.Engine(model, rank=r, world=world, flags=stcsp.F_STEPPED if world == 1 else 0, **opts) for r in range(world)]
    g = stcsp.LocalGroup(world)
    stats = g.solve(engines, **(knobs or {}))
    results = [e.export() for e in engines]
    h, merged = stcsp.merge_shards(results)
    a = stcsp.Automaton(model, merged).traverse().renumber()
    nodes = [e.counters().search_nodes for e in engines]
    return a, merged, stats, nodes, engines, g


@pytest.mark.parametrize("world,name", [(2, "juggling_b4_f5"), (3, "digitinvader3"), (2, "partialorder_10"), (4, "partialorder_12"),
                                        (2, "juggling_b4_f4_nosym")])
def test_native_sharded_matches_golden(stcsp, golden, world, name):
    m = stcsp.Model.from_name(name)
    a, merged, stats, nodes, engines, g = run_local(stcsp, m, world)
    gold = golden[name]
    assert (a.n_live_states, a.n_live_edges, a.canonical_sha256()) == (gold["states"], gold["edges"], gold["canonical_sha256"])
    assert merged.counters.dominance == gold["dom"]
    if gold["fail"] == 0:
        assert sum(nodes) == gold["search"]
    assert sum(s["nodes_donated"] for s in stats) == sum(s["nodes_adopted"] for s in stats)
    assert sum(s["candidates_sent"] for s in stats) == sum(s["candidates_received"] for s in stats) > 0


@pytest.mark.parametrize("world", [2, 3])
def test_native_sharded_redistribution_no_leaf(stcsp, FrontierModel, world):
    """A refutation-only member of the synthetic family (no leaf is ever reached: without redistribution every shard but
    the root's would stay idle): every shard searches, the sum is the unsharded tree."""
    inst = stcsp.instances
    m = stcsp.Model(text=inst.synthetic(24, 8, 125, 4, 7))
    f = FrontierModel(m)
    r1 = f.solve()
    a, merged, stats, nodes, engines, g = run_local(stcsp, m, world, knobs=SHARE)
    assert all(n > 0 for n in nodes), nodes
    assert sum(nodes) == r1.counters.search_nodes
    assert sum(s["nodes_donated"] for s in stats) == sum(s["nodes_adopted"] for s in stats) > 0


def test_native_sharded_headline_two_shards(stcsp, golden):
    name = "partialorder_14"
    m = stcsp.Model.from_name(name)
    a, merged, stats, nodes, engines, g = run_local(stcsp, m, 2)
    gold = golden[name]
    assert (a.n_live_states, a.n_live_edges, a.canonical_sha256()) == (gold["states"], gold["edges"], gold["canonical_sha256"])
    assert sum(nodes) == gold["search"] and all(n > 0 for n in nodes)


def test_native_sharded_wide_domains(stcsp, RefOracle):
    from test_wide_gpu import WIDE
    m = stcsp.Model(text=WIDE["hull_next_sum"])
    o = RefOracle(m)
    ro = o.solve()
    ao, _ = finish(o, ro)
    a, merged, stats, nodes, engines, g = run_local(stcsp, m, 2, knobs=SHARE)
    assert a.canonical_sha256() == ao.canonical_sha256() and merged.counters.dominance == ro.counters.dominance


@pytest.mark.parametrize("call,rank", [("commit", 1), ("expand_local", 0), ("donate", 0), ("adopt", 1)])
def test_native_sharded_failure_ends_every_rank(stcsp, monkeypatch, call, rank):
    """An engine call that fails on ONE rank: every rank returns an error at the same superstep (threads all end)."""
    inst = stcsp.instances
    m = stcsp.Model(text=inst.synthetic(24, 8, 125, 4, 7)) if call in ("donate", "adopt") else stcsp.Model.from_name("partialorder_10")
    monkeypatch.setenv("STCSP_FAULT", f"{call},{rank},{1 if call in ('donate', 'adopt') else 2}")
    engines = [stcsp.Engine(m, rank=r, world=2) for r in range(2)]
    monkeypatch.delenv("STCSP_FAULT")
    g = stcsp.LocalGroup(2)
    with pytest.raises(stcsp.StcspError):
        g.solve(engines, **SHARE)
    assert all(ex is not None for ex in g.errors), g.errors
    assert "injected fault" in str(g.errors[rank]) and "failed in" in str(g.errors[1 - rank])


@pytest.mark.parametrize("name", ["juggling_b4_f5", "partialorder_12"])
def test_native_sharded_rccl_world1(stcsp, golden, monkeypatch, name):
    """The RCCL transport (ncclAllGather for the count table, grouped ncclSend / ncclRecv on the engine's stream) with a
    communicator of size 1. A single shard owns every state and would commit every leaf in place: STCSP_FORCE_CANDIDATES
    sends them all through the exchange (RCCL send / recv to itself) and k_commit instead."""
    m = stcsp.Model.from_name(name)
    monkeypatch.setenv("STCSP_FORCE_CANDIDATES", "1")
    e = stcsp.Engine(m, rank=0, world=1, flags=stcsp.F_STEPPED)
    monkeypatch.delenv("STCSP_FORCE_CANDIDATES")
    t = stcsp.RcclTransport(stcsp.rccl_unique_id(), 0, 1, 0)
    st = stcsp.solve_sharded_native(e, t.ptr)
    r = e.export()
    h, merged = stcsp.merge_shards([r])
    a = stcsp.Automaton(m, merged).traverse().renumber()
    gold = golden[name]
    assert (a.n_live_states, a.n_live_edges, a.canonical_sha256()) == (gold["states"], gold["edges"], gold["canonical_sha256"])
    assert st["candidates_sent"] == st["candidates_received"] == e.counters().leaves > 0  # every leaf travelled
    t.close()


@pytest.mark.parametrize("name,flags", [("juggling_b4_f6", ["-s"]), ("digitinvader4", ["-s", "-a"]), ("partialorder_11", ["-s"])])
def test_cli_shards_writes_the_same_files(stcsp, tmp_path, name, flags):
    """`stcsp --shards=N` (N host threads, one engine each, in-process transport) against the unsharded command line: the
    same statistics fields and byte-identical solutions.dot (the writer orders edges by label)."""
    import subprocess
    exe = stcsp.CSRC / "stcsp"
    src = tmp_path / f"{name}.csp"
    src.write_text(stcsp.instances.by_name(name))
    outs = []
    for extra in ([], ["--shards=2"], ["--shards=3"]):
        d = tmp_path / ("one" if not extra else extra[0].replace("=", ""))
        d.mkdir()
        r = subprocess.run([str(exe), *flags, *extra, str(src)], cwd=d, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr
        line = r.stdout.strip().split("\n")[-1]
        head = r.stdout[: r.stdout.rfind(line)]
        fields = line.split("\t")
        outs.append((head, fields[1:6], (d / "solutions.dot").read_bytes()))
    assert outs[0][0] == outs[1][0] == outs[2][0]          # "adver1: ..." part
    assert outs[0][1][:3] == outs[1][1][:3] == outs[2][1][:3]  # var con dom
    assert outs[0][2] == outs[1][2] == outs[2][2]


def test_native_sharded_resolve_on_the_same_engines(stcsp, golden):
    """The -t loop of the reference re-solves one model (src/solver.cpp:295-349): the same engines and the same transport group
    run the sharded search again and again (the table's generation, the plan mirror and the transport's barriers all start over)."""
    name = "digitinvader3"
    m = stcsp.Model.from_name(name)
    engines = [stcsp.Engine(m, rank=r, world=3) for r in range(3)]
    g = stcsp.LocalGroup(3)
    gold = golden[name]
    for _ in range(3):
        g.solve(engines, budget_rounds=1, share_per_rank=2)
        h, merged = stcsp.merge_shards([e.export() for e in engines])
        a = stcsp.Automaton(m, merged).traverse().renumber()
        assert (a.n_live_states, a.n_live_edges, a.canonical_sha256()) == (gold["states"], gold["edges"], gold["canonical_sha256"])
        assert sum(e.counters().search_nodes for e in engines) == gold["search"]
