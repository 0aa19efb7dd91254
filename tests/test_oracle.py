"""The oracle is pinned before it is trusted: oracle/ref_dfs.cpp must reproduce the reference's
recorded outputs -- canonical sha256 of the automaton AND the reference's own counters (stats
line dom/node/fail, search nodes, arc revisions, validate calls) -- and oracle/frontier_model.cpp
(the scalar model of the build's algorithm, running the product's compiled constraint program)
must agree with it on the automaton."""
import hashlib
import json
import os
from pathlib import Path

import pytest

from conftest import finish
from canon import canon_sha256

REPO = Path(__file__).resolve().parents[1]
PROBES = json.loads((REPO / "tests" / "golden" / "reference_probes.json").read_text())

FAST = ["juggling_b4_f4", "juggling_b4_f5", "juggling_b4_f6", "juggling_b5_f5", "juggling_b6_f6",
        "juggling_b4_f4_nosym", "juggling_b4_f5_nosym", "juggling_b5_f5_nosym", "juggling_b4_f6_nosym",
        "digitinvader1", "digitinvader2", "digitinvader3", "digitinvader4", "partialorder_10", "partialorder_11"]
SLOW = ["juggling_b5_f6", "juggling_b5_f6_nosym", "juggling_b6_f6_nosym", "digitinvader5", "digitinvader6",
        "digitinvader7", "digitinvader8", "digitinvader9", "partialorder_12", "partialorder_13", "partialorder_14"]
slow = pytest.mark.skipif(not os.environ.get("STCSP_SLOW"), reason="minutes of CPU; set STCSP_SLOW=1")


def check_ref(stcsp, RefOracle, golden, name, tmp_path):
    m = stcsp.Model.from_name(name)
    o = RefOracle(m)
    r = o.solve()
    a, _ = finish(o, r)
    g, c = golden[name], r.counters
    got = dict(var=m.n_vars, con=m.n_constraints, dom=c.dominance, node=r.n_states, fail=c.fails,
               search=c.search_nodes, revisions=c.revisions, validate=c.evaluations,
               states=a.n_live_states, edges=a.n_live_edges, canonical_sha256=a.canonical_sha256())
    assert got == g
    # the dot writer + the survey's normative canonicaliser give the same hash as the C++ one
    dot = tmp_path / "solutions.dot"
    a.write_dot(str(dot))
    sha, ns, ne = canon_sha256(str(dot))
    assert (sha, ns, ne) == (g["canonical_sha256"], g["states"], g["edges"])
    assert dot.read_text().splitlines()[0] == f"# Number of nodes = {g['node']}"


@pytest.mark.parametrize("name", FAST)
def test_ref_oracle_reproduces_reference(stcsp, RefOracle, golden, name, tmp_path):
    check_ref(stcsp, RefOracle, golden, name, tmp_path)


@slow
@pytest.mark.parametrize("name", SLOW)
def test_ref_oracle_reproduces_reference_slow(stcsp, RefOracle, golden, name, tmp_path):
    check_ref(stcsp, RefOracle, golden, name, tmp_path)


@pytest.mark.parametrize("name", ["juggling_b4_f4", "juggling_b4_f5", "juggling_b5_f5", "juggling_b4_f4_nosym",
                                  "juggling_b4_f5_nosym", "digitinvader1", "digitinvader2", "digitinvader3"])
def test_frontier_model_matches_reference(stcsp, FrontierModel, golden, name):
    m = stcsp.Model.from_name(name)
    f = FrontierModel(m)
    r = f.solve()
    a, _ = finish(f, r)
    g = golden[name]
    assert (a.n_live_states, a.n_live_edges, a.canonical_sha256()) == (g["states"], g["edges"], g["canonical_sha256"])
    assert r.counters.dominance == g["dom"]


@pytest.mark.parametrize("probe", ["until", "arr", "at", "misc", "adversarial"])
def test_feature_probes(stcsp, RefOracle, FrontierModel, probe):
    """until / arr / @ / abs,not,-> / adversarial: reference stats line + recorded dot facts."""
    p = PROBES[probe]
    m = stcsp.Model(text=p["text"])
    o = RefOracle(m)
    r = o.solve()
    assert [m.n_vars, m.n_constraints, r.counters.dominance, r.n_states, r.counters.fails] == p["stats"]
    a, _ = finish(o, r)
    if "live_states" in p:
        assert a.n_live_states == p["live_states"]
    if "live_edges" in p:
        assert a.n_live_edges == p["live_edges"]
    if "root_final" in p:
        assert a.canonical().splitlines()[2].split()[3] == str(p["root_final"])
    if "constraint_sets" in p:
        assert r.n_constraint_sets == p["constraint_sets"]
    if "edge_labels" in p:
        labels = {tuple(int(x) for x in ln.split()[3:]) for ln in a.canonical().splitlines() if ln.startswith("E ")}
        assert labels == {tuple(l) for l in p["edge_labels"]}
    # the frontier model agrees on the automaton
    f = FrontierModel(m)
    af, _ = finish(f, f.solve())
    assert af.canonical() == a.canonical()
    if probe == "adversarial":
        o1 = RefOracle(m)
        a1, adv = finish(o1, o1.solve(), adversarial="a")
        assert adv == p["adver1"] and (a1.n_live_states, a1.n_live_edges) == (p["adver1_live_states"], p["adver1_live_edges"])
        o2 = RefOracle(m)
        a2, adv2 = finish(o2, o2.solve(), adversarial="z")
        assert adv2 == p["adver2"] and a2.canonical().endswith("EMPTY\n")


def test_prefix_k3_gives_same_automaton(stcsp, RefOracle, FrontierModel, golden):
    """Look-ahead strength does not change the automaton (K=2 and K=3 agree)."""
    m = stcsp.Model.from_name("juggling_b4_f5", prefix_k=3)
    for cls in (RefOracle, FrontierModel):
        e = cls(m)
        a, _ = finish(e, e.solve())
        assert a.canonical_sha256() == golden["juggling_b4_f5"]["canonical_sha256"]


def test_oracle_time_box(stcsp, RefOracle):
    m = stcsp.Model.from_name("partialorder_12")
    o = RefOracle(m, max_search_nodes=5000)
    r = o.solve()
    assert r.truncated == 1 and 5000 <= r.counters.search_nodes <= 5100


ARR_EDGE_CASES = [
    # index out of range makes the comparison false (valid = false), so i is pruned to {0,1}
    "arr T:{1,2}; var i:[0,3]; var v:[0,3]; v == T[i]; next i == (i + 1) % 2;",
    # out-of-range lookup in the branch that is NOT taken must not invalidate the tuple
    "arr T:{5,6,7}; var i:[0,4]; var v:[0,9]; v == if i lt 3 then T[i] else i; next i == i;",
    # ... and `and` / `or` short-circuits guard it the same way
    "arr T:{0,1}; var i:[0,3]; var b:[0,1]; b == ((i lt 2) and (T[i] eq 1)); next i == (i + b) % 4;",
    "arr T:{0,1}; var i:[0,3]; var b:[0,1]; b == ((i ge 2) or (T[i] eq 1)); next i == (i + 1) % 4;",
]


@pytest.mark.parametrize("text", ARR_EDGE_CASES)
def test_array_validity_semantics(stcsp, RefOracle, FrontierModel, text):
    """solverValidateRe's `valid` flag (solveralgorithm.cpp:344-352, 396-397): the compiled
    postfix program + liveness guards (frontier model) must agree with the tree-walking
    restatement of the reference on out-of-range array lookups."""
    m = stcsp.Model(text=text)
    o = RefOracle(m)
    ao, _ = finish(o, o.solve())
    f = FrontierModel(m)
    af, _ = finish(f, f.solve())
    assert af.canonical() == ao.canonical()
    assert ao.n_live_states > 1


@pytest.mark.parametrize("name", ["juggling_b4_f5", "partialorder_10", "digitinvader2", "juggling_b4_f4_nosym"])
def test_parallel_restatement_matches_reference_golden(stcsp, oracle_lib, RefOracle, golden, name):
    """bench.py's all-cores CPU leg (oracle/ref_dfs.cpp struct Shared: workers over the automaton's states):
    the reference's recorded automaton and its order-independent counters."""
    import ctypes as C
    oracle_lib.stcsp_oracle_solve_parallel.argtypes = [C.c_void_p, C.c_int, C.POINTER(stcsp.Result)]
    m = stcsp.Model.from_name(name)
    o = RefOracle(m)
    for threads in (1, 4):
        o._check(oracle_lib.stcsp_oracle_solve_parallel(o._h, threads, C.byref(o.result)))
        r = o.result
        a = o.automaton(r).traverse().renumber()
        g = golden[name]
        assert (a.n_live_states, a.n_live_edges, a.canonical_sha256()) == (g["states"], g["edges"], g["canonical_sha256"])
        assert r.counters.dominance == g["dom"]
        if g["fail"] == 0:
            assert r.n_states == g["node"] and r.counters.search_nodes == g["search"]


# ---- the product's ahead-of-need constraint-set translation (cset.cpp SetManager::pretranslate, linked into the oracle
# library with the frontier model) on the CPU: what gets translated, which sets get a tuple-indexed transition table, and
# that a model solved AFTER it gives the reference's automaton and set count
PRETRANSLATE_CASES = [
    # (model, sets after pretranslation, sets with a direct table, direct-table entries)
    ("name:partialorder_10", 2, 1, 10 * 2 ** 10 * 2),   # 20,480 captured tuples, cut to 2 by the unary `first` constraints
    ("name:partialorder_14", 2, 1, 14 * 2 ** 14 * 2),   # 458,752 -> 2
    ("var x:[0,31]; var y:[0,31]; var z:[0,1]; y + z == first x; next z == 1 - z;", 33, 1, 32),
    ("var x:[0,20]; var w:[0,3]; var y:[0,24]; var z:[0,1]; y == first x + first w + z; next z == 1 - z; first w <= 1;", 43, 1, 84),
]


@pytest.mark.parametrize("spec,n_sets,n_direct,entries", PRETRANSLATE_CASES)
def test_pretranslation_on_the_cpu(stcsp, oracle_lib, RefOracle, FrontierModel, spec, n_sets, n_direct, entries):
    import ctypes as C
    oracle_lib.stcsp_fmodel_pretranslate.argtypes = [C.c_void_p, C.c_longlong, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_longlong)]
    m = stcsp.Model.from_name(spec[5:]) if spec.startswith("name:") else stcsp.Model(text=spec)
    f = FrontierModel(m)
    ns, nd, te = C.c_int(), C.c_int(), C.c_longlong()
    added = oracle_lib.stcsp_fmodel_pretranslate(f._h, 65536, 16384, C.byref(ns), C.byref(nd), C.byref(te))
    assert added >= 1
    assert (ns.value, nd.value, te.value) == (n_sets, n_direct, entries)
    if not spec.startswith("name:partialorder_14"):  # (the scalar model needs minutes for partialorder_14)
        r = f.solve()
        a = f.automaton(r).traverse().renumber()
        o = RefOracle(m)
        ro = o.solve()
        ao = o.automaton(ro).traverse().renumber()
        assert a.canonical_sha256() == ao.canonical_sha256()
        assert r.n_constraint_sets == ro.n_constraint_sets


# Two translated constraint sets that differ ONLY in the array a node indexes are one set to the reference
# (constraintNodeEq, src/constraint.cpp:551-561, compares token / num / var and the shape, not the array): the set its DFS
# translated first stands for both.  The product's set registry must merge them too (content hash over what set_eq compares)
# and its ahead-of-need translation must meet the tuples in the DFS's leaf order, or the automaton differs.
TWO_ARRAYS = ("arr A:{0,1}; arr B:{1,0}; var p:[0,1]; var q:[0,1]; var z:[0,1]; "
              "((first p) and A[z]) == 0; ((first q) and B[z]) == 0; next p == p; next q == q;")


@pytest.mark.parametrize("pretranslate", [False, True])
def test_sets_that_differ_only_in_the_array_are_one_set(stcsp, oracle_lib, RefOracle, FrontierModel, pretranslate):
    import ctypes as C
    m = stcsp.Model(text=TWO_ARRAYS)
    o = RefOracle(m)
    ro = o.solve()
    ao, _ = finish(o, ro)
    f = FrontierModel(m)
    if pretranslate:
        oracle_lib.stcsp_fmodel_pretranslate.argtypes = [C.c_void_p, C.c_longlong, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_longlong)]
        ns, nd, te = C.c_int(), C.c_int(), C.c_longlong()
        assert oracle_lib.stcsp_fmodel_pretranslate(f._h, 65536, 16384, C.byref(ns), C.byref(nd), C.byref(te)) >= 1
        assert ns.value == 4  # initial, {}, {c'}, {c1', c2'} (no leaf ever shows p = q = 1: the DFS never translates it) -- not 5
    r = f.solve()
    a, _ = finish(f, r)
    assert ro.n_constraint_sets == 3 and r.n_constraint_sets == (4 if pretranslate else 3)
    assert a.canonical() == ao.canonical()
    # the merged set really is the one over B (z = 1 on the self-loops of BOTH one-constraint states), as in the reference
    assert "E 2 2 0 1 1 0 1" in ao.canonical() and "E 3 3 1 0 1 1 0" in ao.canonical()


def test_wide_conditional_constraints_split_into_guarded_branches(stcsp, FrontierModel, golden, monkeypatch):
    """cset.cpp split_wide on the CPU (the frontier model runs the product's compiler): juggling_b4_f5_nosym's 9-variable
    `A == if B0 eq 1 then next B0 else if ...` (6^9 tuples: more than the host tabulates) becomes five guarded branches of at most
    6^6 tuples. Same automaton as the reference either way, a fraction of the tuple evaluations with the branches."""
    name = "juggling_b4_f5_nosym"
    g = golden[name]
    evals = {}
    for mode in ["0", "1"]:
        monkeypatch.setenv("STCSP_SPLIT_WIDE", mode)
        f = FrontierModel(stcsp.Model.from_name(name))
        r = f.solve()
        a, _ = finish(f, r)
        assert (a.n_live_states, a.n_live_edges, a.canonical_sha256()) == (g["states"], g["edges"], g["canonical_sha256"]), mode
        assert r.counters.dominance == g["dom"]
        evals[mode] = r.counters.evaluations
    assert evals["1"] * 3 < evals["0"]
