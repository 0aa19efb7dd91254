"""Domains of more than 32 values (W = ceil(|D| / 32) bitset words per variable and time point; the
reference's intervals take any width, src/variable.h:19-20, and aux variables of arithmetic under next / fby get hull
bounds, src/solveralgorithm.cpp:182-184): the engine's W = 2 (<= 64 values) and W = 4 (<= 128) kernels (dev_wide.hpp)
against oracle/ref_dfs.cpp through the C-ABI -- canonical automaton, `dom`, and the search tree itself whenever neither
side ever fails."""
import pytest

from conftest import finish
from fuzz_models import random_wide_model

pytestmark = pytest.mark.gpu

WIDE = {
    # the review's example: the aux variables of `next (x + y)` get the hull [0, 40] = 41 values
    "hull_next_sum": "var x:[0,20]; var y:[0,20]; var z:[0,40]; z == next (x + y); x + y <= 30;",
    "counter100": "var x:[0,99]; var y:[0,99]; first x == 0; next x == (x + 7) % 100; y == 99 - x;",
    "neg_w4": "var a:[-50,60]; var b:[0,3]; first a == -50; next a == (if (a ge 55) then -50 else (a + b + 1)); b != 2;",
    "fby_hull": "var x:[0,40]; var y:[0,2]; var z:[0,45]; z == (0 fby (x + y)); x <= 3 * y + 30; x >= 28;",
    "w2_until": "var x:[0,50]; var g:[0,1]; var y:[0,1]; first x == 0; next x == (if (x lt 50) then (x + 1) else x); y == (x ge 50); g until y;",
    "w4_two_wide": "var x:[0,99]; var y:[0,99]; var s:[0,1]; first x == 3; next x == y; y <= x + 2; y >= x - 1; y <= 12; s == ((x + y) % 2);",
    "w2_mult": "var x:[0,7]; var y:[0,7]; var p:[0,49]; p == x * y; first x == 1; next x == (if (p gt 20) then 1 else (x + 1)) ; y <= x;",
    "w4_edge_128": "var x:[-1,126]; var y:[0,1]; first x == -1; next x == (if (x ge 126) then -1 else (x + 3 - y));",
}


def compare(stcsp, RefOracle, text, prefix_k=2, **opts):
    m = stcsp.Model(text=text, prefix_k=prefix_k)
    o = RefOracle(m, time_limit_s=20.0)
    ro = o.solve()
    assert not ro.truncated
    ao, _ = finish(o, ro)
    e = stcsp.Engine(m, **opts)
    r = e.solve()
    a, _ = finish(e, r)
    assert a.canonical() == ao.canonical()
    assert r.counters.dominance == ro.counters.dominance
    if ro.counters.fails == 0 and r.counters.fails == 0:
        assert (r.n_states, r.counters.search_nodes) == (ro.n_states, ro.counters.search_nodes)
    return m, r, ro


@pytest.mark.parametrize("name", sorted(WIDE))
def test_wide_models_match_reference_restatement(stcsp, RefOracle, name):
    m, r, ro = compare(stcsp, RefOracle, WIDE[name])
    assert max(hi - lo + 1 for lo, hi in m.var_bounds()) > 32
    assert r.n_states > 3


@pytest.mark.parametrize("name", ["hull_next_sum", "neg_w4"])
def test_wide_models_small_batches_and_pools(stcsp, RefOracle, monkeypatch, name):
    """Depth-first segment stack (64-node batches) and every pool growing, under the wide kernels."""
    monkeypatch.setenv("STCSP_SMALL_POOLS", "1")
    compare(stcsp, RefOracle, WIDE[name], batch_nodes=64)


@pytest.mark.parametrize("name", ["w4_two_wide", "fby_hull"])
def test_wide_models_prefix_k3(stcsp, RefOracle, name):
    compare(stcsp, RefOracle, WIDE[name], prefix_k=3)


def test_domains_beyond_128_values_are_refused(stcsp):
    m = stcsp.Model(text="var x:[0,128]; var y:[0,1]; next x == x + y;")
    with pytest.raises(stcsp.StcspError) as ex:
        stcsp.Engine(m)
    assert ex.value.code == -2 and "128" in str(ex.value)


@pytest.mark.parametrize("block", range(3))
def test_fuzz_wide_domains(stcsp, RefOracle, block):
    """Random models with one or two variables of 33..128 values (tests/fuzz_models.py WideGen)."""
    checked = w4 = nontrivial = 0
    for seed in range(block * 100, (block + 1) * 100):
        text = random_wide_model(seed)
        m = stcsp.Model(text=text)
        o = RefOracle(m, time_limit_s=2.0)
        ro = o.solve()
        if ro.truncated:
            continue  # (a support search of the restatement over 100^3 tuples: not a case for a test)
        ao, _ = finish(o, ro)
        try:
            e = stcsp.Engine(m)
        except stcsp.StcspError as ex:
            assert ex.code == -2, f"seed {seed}: {ex}\n{text}"  # hull bounds of an aux variable beyond 128 values
            continue
        r = e.solve()
        a, _ = finish(e, r)
        assert a.canonical() == ao.canonical(), f"seed {seed}\n{text}"
        assert r.counters.dominance == ro.counters.dominance, f"seed {seed}\n{text}"
        if ro.counters.fails == 0 and r.counters.fails == 0:
            assert (r.n_states, r.counters.search_nodes) == (ro.n_states, ro.counters.search_nodes), f"seed {seed}\n{text}"
        checked += 1
        w4 += max(hi - lo + 1 for lo, hi in m.var_bounds()) > 64
        nontrivial += a.n_live_states > 3
        e.close()
    assert checked >= 80 and w4 >= 20 and nontrivial >= 10


@pytest.mark.parametrize("name", ["hull_next_sum", "neg_w4"])
def test_wide_models_two_hip_shards_one_gpu(stcsp, RefOracle, tmp_path, name):
    """Chunked blocks through the sharded pipeline (candidate records, commit, frontier redistribution) with two
    HIP-engine shards on one GPU."""
    from test_sharded import launch, SHARE
    f = tmp_path / f"{name}.csp"
    f.write_text(WIDE[name])
    m = stcsp.Model(text=WIDE[name])
    o = RefOracle(m)
    ro = o.solve()
    ao, _ = finish(o, ro)
    r = launch(2, f"file:{f}", "hip", tmp_path, env=SHARE)
    assert r["sha"] == ao.canonical_sha256() and r["dom"] == ro.counters.dominance
    assert sum(r["donated"]) == sum(r["adopted"])
