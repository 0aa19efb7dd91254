"""Parity tests proper: the HIP engine, called through the C-ABI, against the oracle and the
reference's recorded golden values. Run on the GPU box: pytest -m gpu."""
import os

import pytest

from conftest import finish

pytestmark = pytest.mark.gpu

import json
from pathlib import Path

PROBES = json.loads((Path(__file__).resolve().parent / "golden" / "reference_probes.json").read_text())


def all_examples():
    import importlib
    return importlib.import_module("stcsp-solver_amd").instances.REFERENCE_EXAMPLES


@pytest.mark.parametrize("name", all_examples())
def test_engine_matches_reference_golden(stcsp, golden, name):
    """Canonical automaton (states keyed by (set, signature), final flags, labelled edges)
    bit-identical to the reference's: sha256 of the canonical text equals the recorded value."""
    m = stcsp.Model.from_name(name)
    e = stcsp.Engine(m)
    r = e.solve()
    a, _ = finish(e, r)
    g = golden[name]
    assert a.n_live_states == g["states"]
    assert a.n_live_edges == g["edges"]
    assert a.canonical_sha256() == g["canonical_sha256"]
    assert r.counters.dominance == g["dom"]          # order-independent
    assert r.truncated == 0
    if g["fail"] == 0:
        # table size (failed look-ahead states included) and the search tree itself coincide with
        # the reference's whenever the reference never fails
        assert r.n_states == g["node"]
        assert r.counters.search_nodes == g["search"]


@pytest.mark.parametrize("probe", ["until", "arr", "at", "misc", "adversarial"])
def test_engine_feature_probes(stcsp, RefOracle, probe):
    """until / arr (+ out-of-range `valid`) / @ (4 constraint sets) / abs,not,-> / adversarial:
    the reference's stats line columns and the oracle's canonical automaton."""
    p = PROBES[probe]
    m = stcsp.Model(text=p["text"])
    e = stcsp.Engine(m)
    r = e.solve()
    assert [m.n_vars, m.n_constraints, r.counters.dominance, r.n_states, r.counters.fails] == p["stats"]
    a, _ = finish(e, r)
    o = RefOracle(m)
    ao, _ = finish(o, o.solve())
    assert a.canonical() == ao.canonical()
    if "constraint_sets" in p:
        assert r.n_constraint_sets == p["constraint_sets"]
    if probe == "adversarial":
        e1 = stcsp.Engine(m)
        a1, adv = finish(e1, e1.solve(), adversarial="a")
        assert adv == p["adver1"] and (a1.n_live_states, a1.n_live_edges) == (p["adver1_live_states"], p["adver1_live_edges"])
        e2 = stcsp.Engine(m)
        a2, adv2 = finish(e2, e2.solve(), adversarial="z")
        assert adv2 == p["adver2"] and a2.canonical().endswith("EMPTY\n")


def test_engine_resolve_and_limits(stcsp, golden):
    """solve() may be called repeatedly (the reference's -t loop re-solves the model); node and
    time limits truncate cleanly."""
    m = stcsp.Model.from_name("partialorder_10")
    e = stcsp.Engine(m)
    for _ in range(3):
        a, _ = finish(e, e.solve())
        assert a.canonical_sha256() == golden["partialorder_10"]["canonical_sha256"]
    e2 = stcsp.Engine(m, max_search_nodes=2000, batch_nodes=256)
    r = e2.solve()
    assert r.truncated == 1 and r.counters.search_nodes >= 2000


def test_engine_small_batches_depth_first(stcsp, golden):
    """A tiny launch batch forces the chunked, depth-first segment stack (bounded frontier
    memory) instead of level-synchronous expansion: same automaton."""
    for name in ["juggling_b4_f5_nosym", "digitinvader3", "partialorder_10"]:
        m = stcsp.Model.from_name(name)
        e = stcsp.Engine(m, batch_nodes=64)
        a, _ = finish(e, e.solve())
        assert a.canonical_sha256() == golden[name]["canonical_sha256"], name


def test_engine_prefix_k3(stcsp, golden):
    m = stcsp.Model.from_name("juggling_b4_f5", prefix_k=3)
    e = stcsp.Engine(m)
    a, _ = finish(e, e.solve())
    assert a.canonical_sha256() == golden["juggling_b4_f5"]["canonical_sha256"]


def test_engine_rejects_wide_domains(stcsp):
    """Aux variables of / and % get [INT_MIN, INT_MAX] (solveralgorithm.cpp:316-322): outside the
    bitset path -> a clean STCSP_E_UNSUPPORTED, never a crash."""
    m = stcsp.Model(text="var x:[0,3]; var y:[1,3]; var z:[0,3]; z == next (x / y);")
    with pytest.raises(stcsp.StcspError) as ex:
        stcsp.Engine(m)
    assert ex.value.code == -2


@pytest.mark.parametrize("name", ["juggling_b4_f5", "digitinvader2", "juggling_b4_f4_nosym"])
def test_engine_matches_oracle_text(stcsp, RefOracle, name):
    """Same seeded input through oracle/ref_dfs.cpp and the engine: identical canonical text."""
    m = stcsp.Model.from_name(name)
    o = RefOracle(m)
    ao, _ = finish(o, o.solve())
    e = stcsp.Engine(m)
    ae, _ = finish(e, e.solve())
    assert ae.canonical() == ao.canonical()


@pytest.mark.parametrize("shape", [(8, 4, 14, 2, 1), (16, 8, 95, 4, 3), (16, 8, 88, 4, 4), (16, 8, 80, 4, 5)])
def test_engine_synthetic_parity(stcsp, RefOracle, shape):
    """BASELINE.json config 4 family (random extensional binary constraints through `arr`
    lookups, some under `next`) at shapes the CPU oracle finishes: these instances exercise the
    FAILING branches the shipped examples (fail = 0) never hit."""
    m = stcsp.Model(text=stcsp.instances.synthetic(*shape))
    o = RefOracle(m)
    ao, _ = finish(o, o.solve())
    e = stcsp.Engine(m)
    r = e.solve()
    ae, _ = finish(e, r)
    assert ae.canonical() == ao.canonical()
    assert r.counters.fails > 0 or shape[0] == 8


@pytest.mark.parametrize("name,shape", [("partialorder_10", None), ("partialorder_13", None), ("partialorder_14", None),
                                        ("synth95", (16, 8, 95, 4, 3)), ("synth88", (16, 8, 88, 4, 4)), ("synth125", (24, 8, 125, 4, 7))])
def test_engine_big_workgroup_kernel_parity(stcsp, RefOracle, golden, monkeypatch, capfd, name, shape):
    """The big-workgroup variant of k_expand (one 1024-thread workgroup per CU around ONE LDS copy of a LITE program: what the
    synthetic 64 x 32 family and partialorder_18 run under) forced onto programs small enough for the CPU oracle
    (STCSP_BIG=2): same automaton as oracle/ref_dfs.cpp / the recorded reference values, same search tree as the regular
    kernel."""
    m = stcsp.Model(text=stcsp.instances.synthetic(*shape)) if shape else stcsp.Model.from_name(name)
    monkeypatch.setenv("STCSP_BIG", "0")
    e0 = stcsp.Engine(m)
    r0 = e0.solve()
    a0, _ = finish(e0, r0)
    monkeypatch.setenv("STCSP_BIG", "2")
    monkeypatch.setenv("STCSP_DEBUG", "1")
    capfd.readouterr()
    e = stcsp.Engine(m)
    r = e.solve()
    assert "big-workgroup kernel" in capfd.readouterr().err
    monkeypatch.delenv("STCSP_DEBUG")
    a, _ = finish(e, r)
    if shape:
        o = RefOracle(m)
        ao, _ = finish(o, o.solve())
        assert a.canonical() == ao.canonical()
        assert r.counters.fails > 0
    else:
        assert a.canonical_sha256() == golden[name]["canonical_sha256"]
    assert a.canonical_sha256() == a0.canonical_sha256()
    assert (r.counters.search_nodes, r.counters.fails, r.counters.leaves, r.n_states) == \
           (r0.counters.search_nodes, r0.counters.fails, r0.counters.leaves, r0.n_states)


def test_engine_synthetic_64x32_timebox(stcsp):
    """64 vars x |D| = 32, 602 point + 6 stream constraints: a time-boxed throughput run (no
    implementation reaches a leaf quickly). Exercises 4-register blocks, the global-memory
    program path, the chunked depth-first frontier, arena growth and the time limit."""
    m = stcsp.Model(text=stcsp.instances.synthetic(64, 32, 602, 6, 20261003))
    assert m.n_vars == 70
    e = stcsp.Engine(m, time_limit_s=2.0)
    r = e.solve()
    assert r.truncated == 1
    assert r.counters.search_nodes > 1_000_000
    assert r.counters.fails > 0


@pytest.mark.parametrize("blocks", [1, 3, 37])
def test_engine_tiny_grids_take_every_slot(stcsp, golden, monkeypatch, blocks):
    """Slots beyond a wavefront's first are dealt by ticket from sixteen interleaved counters (k_expand): with fewer workgroups
    than counters (STCSP_BLOCKS: a grid of 1, 3 or 37 workgroups instead of the resident 1,536) every slot must still be taken
    exactly once -- same automaton and node count as the reference."""
    monkeypatch.setenv("STCSP_BLOCKS", str(blocks))
    for name in ["partialorder_11", "digitinvader3"]:
        m = stcsp.Model.from_name(name)
        e = stcsp.Engine(m)
        r = e.solve()
        a, _ = finish(e, r)
        assert a.canonical_sha256() == golden[name]["canonical_sha256"], name
        assert r.counters.search_nodes == golden[name]["search"] or golden[name]["fail"] > 0, name


def test_engine_pool_growth_paths(stcsp, golden, monkeypatch):
    """Start with tiny device pools (STCSP_SMALL_POOLS): the frontier arena, edge log, state pool
    and hash table all have to grow (realloc / rehash between launch bursts) several times."""
    monkeypatch.setenv("STCSP_SMALL_POOLS", "1")
    for name in ["juggling_b4_f5_nosym", "digitinvader3", "partialorder_11"]:
        m = stcsp.Model.from_name(name)
        e = stcsp.Engine(m)
        r = e.solve()
        a, _ = finish(e, r)
        assert a.canonical_sha256() == golden[name]["canonical_sha256"], name
        assert r.n_states == golden[name]["node"] or golden[name]["fail"] > 0


def test_engine_streaming_export_equals_compacting_export(stcsp, golden, monkeypatch):
    """The edge log leaves the device in chunks WHILE the search runs (engine.hip stream_edges); the result is
    used as is when no state fails (the shipped examples) and dropped for the compacting export when one does
    (synthetic instances). Tiny chunks (one per launch round), small pools (the log is reallocated under the
    stream) and the switched-off path all give the same automaton."""
    names = ["partialorder_11", "digitinvader3", "juggling_b4_f5_nosym"]
    synth = stcsp.instances.synthetic(16, 8, 88, 4, 4)
    def run(m, **kw):
        e = stcsp.Engine(m, **kw)
        r = e.solve()
        a, _ = finish(e, r)
        a2, _ = finish(e, e.export())  # a second export of the same solve
        assert a.canonical_sha256() == a2.canonical_sha256()
        return a.canonical_sha256(), r.counters.dominance, r.counters.fails
    monkeypatch.setenv("STCSP_STREAM_EXPORT", "0")
    want = {n: run(stcsp.Model.from_name(n)) for n in names}
    want_synth = run(stcsp.Model(text=synth))
    assert want_synth[2] > 0
    for n in names:
        assert want[n][0] == golden[n]["canonical_sha256"]
    monkeypatch.delenv("STCSP_STREAM_EXPORT")
    for chunk, small, batch in [("1", "0", 0), ("64", "1", 64), ("100000000", "0", 0)]:
        monkeypatch.setenv("STCSP_STREAM_CHUNK", chunk)
        if small == "1":
            monkeypatch.setenv("STCSP_SMALL_POOLS", "1")
        kw = {"batch_nodes": batch} if batch else {}
        for n in names:
            assert run(stcsp.Model.from_name(n), **kw) == want[n], (n, chunk)
        assert run(stcsp.Model(text=synth), **kw) == want_synth, chunk
        monkeypatch.delenv("STCSP_SMALL_POOLS", raising=False)


MANY_SETS = [
    # `first x` inside arithmetic: one constraint set per captured value (33 = 1 + |D(x)|)
    ("var x:[0,31]; var y:[0,31]; var z:[0,1]; y + z == first x; next z == 1 - z;", 33),
    # two captured variables, one of them pinned by its own `first` constraint (1 + 21 * 2)
    ("var x:[0,20]; var w:[0,3]; var y:[0,24]; var z:[0,1]; y == first x + first w + z; next z == 1 - z; first w <= 1;", 43),
]


@pytest.mark.parametrize("text,n_sets", MANY_SETS)
def test_engine_many_constraint_sets_never_stop_the_device(stcsp, RefOracle, monkeypatch, text, n_sets):
    """SURVEY 8(f) row 1 (constraintTranslate per leaf, src/constraint.cpp:466-548 / solveralgorithm.cpp:755-805):
    sets whose captured `first` variables span few value tuples are translated ahead of need, so the device never
    stops for the host (translation_stops == 0); switched off, the same model stops once per batch of new sets.
    Either way: the reference-faithful oracle's automaton and its number of constraint sets."""
    m = stcsp.Model(text=text)
    o = RefOracle(m)
    ro = o.solve()
    ao, _ = finish(o, ro)
    assert ro.n_constraint_sets == n_sets
    e = stcsp.Engine(m)
    r = e.solve()
    a, _ = finish(e, r)
    assert a.canonical() == ao.canonical()
    assert r.n_constraint_sets == n_sets
    assert r.counters.translation_stops == 0
    assert r.counters.dominance == ro.counters.dominance  # order-independent
    monkeypatch.setenv("STCSP_PRETRANSLATE", "0")
    e2 = stcsp.Engine(m)
    r2 = e2.solve()
    a2, _ = finish(e2, r2)
    assert a2.canonical() == ao.canonical() and r2.n_constraint_sets == n_sets
    assert r2.counters.translation_stops >= 1


def test_engine_array_validity_semantics(stcsp, RefOracle):
    """Out-of-range array lookups (reference `valid` flag), also under branches that are not
    taken: tabulated by the host evaluator and checked against the oracle."""
    from test_oracle import ARR_EDGE_CASES
    for text in ARR_EDGE_CASES:
        m = stcsp.Model(text=text)
        o = RefOracle(m)
        ao, _ = finish(o, o.solve())
        e = stcsp.Engine(m)
        ae, _ = finish(e, e.solve())
        assert ae.canonical() == ao.canonical(), text


def test_engine_batch_shrinks_at_arena_soft_limit(stcsp, golden, monkeypatch):
    """An automatic launch batch (256 k nodes) halves itself instead of growing the frontier arena
    past its soft limit (explosive searches: memory ~ depth x batch); same automaton."""
    monkeypatch.setenv("STCSP_ARENA_SOFT_MB", "4")
    for name in ["partialorder_14", "digitinvader5"]:
        m = stcsp.Model.from_name(name)
        e = stcsp.Engine(m)
        r = e.solve()
        a, _ = finish(e, r)
        assert a.canonical_sha256() == golden[name]["canonical_sha256"], name
        assert r.counters.search_nodes == golden[name]["search"]
    # partialorder_16 (frontier of several hundred thousand nodes): the limited run really uses smaller rounds
    big = stcsp.Model(text=stcsp.instances.partialorder(16))
    rl = stcsp.Engine(big).solve()
    monkeypatch.delenv("STCSP_ARENA_SOFT_MB")
    ru = stcsp.Engine(big).solve()
    assert (rl.counters.search_nodes, rl.n_states) == (ru.counters.search_nodes, ru.n_states) == (6112188, 130048)
    assert ru.counters.levels < rl.counters.levels


# ---- device tabulation (k_tabulate): constraints whose tuple space is too big for the host (2^22 .. 2^28 tuples)
# Eight variables over [0,7] = 2^24 tuples of the initial product (what the bitmap is tabulated over); six unary
# constraints keep the search at 2^6 * 8 * 8 leaves so that the reference-faithful oracle finishes in seconds.
TAB_HEAD = "".join(f"var x{i} : [0, 7];\n" for i in range(8)) + "arr T : {3, 1, 4, 1, 5, 9, 2, 6};\n" + \
           "".join(f"x{i} < 2;\n" for i in range(1, 7))
TAB_ARR_IF = TAB_HEAD + "T[x0 + x1 * 7] + (if (x1 gt x2) then (x3 + x7) else (x4 - x7)) + x5 * x6 - x7 >= 2;\n"  # x0 + 7 x1 leaves the array for x1 = 1, x0 > 0


def deep_sum(depth):
    """x0 + (x1 + (x2 + ... )) nested `depth` levels: the postfix program needs an operand stack of depth + 1."""
    expr = "x7"
    for k in range(depth):
        expr = f"x{k % 7} + ({expr})"
    return expr


TAB_DEEP = TAB_HEAD + f"{deep_sum(36)} >= 6;\n"


@pytest.mark.parametrize("text,label", [(TAB_ARR_IF, "array + if"), (TAB_DEEP, "operand stack deeper than k_tabulate's 32 registers")])
def test_device_tabulated_bitmap_equals_host_evaluation_and_oracle(stcsp, RefOracle, monkeypatch, text, label):
    """k_tabulate ran deep programs on a wrapping 32-entry stack, and nothing compared a device-
    tabulated bitmap (incl. the array / `valid` path) with the host evaluator. Same model three ways: device tabulation
    on, device tabulation off (the constraint is interpreted by the wavefront revision), reference-faithful oracle."""
    m = stcsp.Model(text=text)
    o = RefOracle(m)
    ao, _ = finish(o, o.solve())
    shas = {}
    for tab in ("1", "0"):
        monkeypatch.setenv("STCSP_DEVICE_TABULATE", tab)
        e = stcsp.Engine(m)
        r = e.solve()
        a, _ = finish(e, r)
        shas[tab] = a.canonical_sha256()
        assert a.canonical() == ao.canonical(), f"{label}: STCSP_DEVICE_TABULATE={tab}"
        e.close()
    assert shas["1"] == shas["0"]


# ---- constraint-set translation beyond the old 4,096-tuple cap (reference src/constraint.cpp:466-548,
# :551-576; per-leaf use src/solveralgorithm.cpp:755-805). Three captured variables span 9 * 16 * 32 = 4,608 value tuples and
# every tuple translates to a constraint set of its own (the reference folds `first a` to a constant, not the product around
# it): 4,609 sets, 9,217 states, 73,728 + ... edges. The oracle (ref_dfs.cpp) re-translates at every leaf like the reference.
MANY_SETS_BIG = """var a : [0, 8];
var b : [0, 15];
var c : [0, 31];
var y : [0, 3];
var z : [0, 1];
y + first a * 64 + first b * 4 + first c >= 0;
next z == z;
next a == 0;
next b == 0;
next c == 0;
"""


def test_engine_thousands_of_constraint_sets_without_stopping(stcsp, RefOracle, monkeypatch):
    m = stcsp.Model(text=MANY_SETS_BIG)
    o = RefOracle(m)
    ro = o.solve()
    ao, _ = finish(o, ro)
    assert ro.n_constraint_sets == 4609
    # translated ahead of need, transitions looked up in a table indexed by the captured tuple: the device never stops
    e = stcsp.Engine(m)
    r = e.solve()
    assert r.counters.translation_stops == 0
    assert r.n_constraint_sets == 4609
    a, _ = finish(e, r)
    assert (a.n_live_states, a.n_live_edges) == (ao.n_live_states, ao.n_live_edges)
    assert a.canonical_sha256() == ao.canonical_sha256()
    assert r.counters.dominance == ro.counters.dominance
    e.close()
    # without the ahead-of-need translation the host is asked per batch of unknown tuples: each stop serves up to 4,096
    # requests (many of them for the same tuple), so at most a few dozen stops for 4,608 tuples; same automaton
    monkeypatch.setenv("STCSP_PRETRANSLATE", "0")
    e2 = stcsp.Engine(m)
    r2 = e2.solve()
    assert 1 <= r2.counters.translation_stops <= 32
    a2, _ = finish(e2, r2)
    assert a2.canonical_sha256() == ao.canonical_sha256()


def test_engine_merges_sets_that_differ_only_in_the_array(stcsp, RefOracle):
    """constraintNodeEq (src/constraint.cpp:551-561) ignores the array a node indexes: two translations that differ only
    there are ONE constraint set to the reference, the one its DFS met first. Same automaton on the engine (whose
    ahead-of-need translation meets the captured tuples in the DFS's leaf order)."""
    from test_oracle import TWO_ARRAYS
    m = stcsp.Model(text=TWO_ARRAYS)
    o = RefOracle(m)
    ao, _ = finish(o, o.solve())
    e = stcsp.Engine(m)
    r = e.solve()
    a, _ = finish(e, r)
    assert a.canonical() == ao.canonical()
    assert r.counters.translation_stops == 0


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["0", "1", "2"])
def test_engine_wide_conditionals_as_guarded_branches(stcsp, golden, monkeypatch, mode):
    """cset.cpp split_wide: a conditional constraint too wide for a tuple bitmap (juggling `_nosym`: 11 and 13 variables of 6-7
    values) runs as the conjunction of its guarded branches in the compiled program -- same solutions, so the same automaton as the
    reference, and the same search tree wherever the reference never fails. STCSP_SPLIT_WIDE=0 keeps the constraint interpreted,
    2 also splits every conditional constraint that has a bitmap (digitinvader's `next D == if ...`)."""
    monkeypatch.setenv("STCSP_SPLIT_WIDE", mode)
    for name in ["juggling_b5_f5_nosym", "juggling_b6_f6_nosym", "digitinvader3", "juggling_b4_f5"]:
        m = stcsp.Model.from_name(name)
        e = stcsp.Engine(m)
        r = e.solve()
        a, _ = finish(e, r)
        g = golden[name]
        assert (a.n_live_states, a.n_live_edges, a.canonical_sha256()) == (g["states"], g["edges"], g["canonical_sha256"]), name
        assert r.counters.dominance == g["dom"], name
        assert r.counters.search_nodes == g["search"] or g["fail"] > 0, name


@pytest.mark.gpu
def test_engine_fresh_states_start_with_the_new_point_dirty_only(stcsp, golden):
    """A state opened under the constraint set of the state it comes from starts with the dirty seed N*K + 1: only the items that
    read the new time point (and until / `first` items) are revised at its first node (dev_propagate.hpp process_node). The
    fixpoint is the same as with every item dirty: same search tree as the reference on models that open tens of thousands of
    states under one set, with chains (states entered inside a slot) and with 64-node batches (states entered from the frontier)."""
    for name, kw in [("partialorder_12", {}), ("digitinvader6", {}), ("digitinvader4", {"batch_nodes": 64}), ("juggling_b5_f6", {"batch_nodes": 64})]:
        m = stcsp.Model.from_name(name)
        e = stcsp.Engine(m, **kw)
        r = e.solve()
        a, _ = finish(e, r)
        g = golden[name]
        assert a.canonical_sha256() == g["canonical_sha256"], name
        assert (r.counters.search_nodes, r.n_states, r.counters.fails) == (g["search"], g["node"], 0), name
