"""Random models over the whole constraint language (tests/fuzz_models.py): every implementation
must produce the same canonical automaton and the same order-independent counter.

  CPU : oracle/ref_dfs.cpp (the reference's algorithm restated, pinned to the reference's golden
        values in test_oracle.py) vs oracle/frontier_model.cpp (scalar model of the engine's
        frontier algorithm and propagators)
  GPU : the HIP engine through the C-ABI vs oracle/ref_dfs.cpp
"""
import pytest

from conftest import finish
from fuzz_models import random_model


def solve_canonical(stcsp, cls, text, prefix_k=2, **opts):
    m = stcsp.Model(text=text, prefix_k=prefix_k)
    e = cls(m, **opts)
    r = e.solve()
    a, _ = finish(e, r)
    return m, r, a


@pytest.mark.parametrize("block", range(4))
def test_fuzz_frontier_model_matches_reference_restatement(stcsp, RefOracle, FrontierModel, block):
    n_nontrivial = 0
    for seed in range(block * 150, (block + 1) * 150):
        text = random_model(seed)
        _, r, a = solve_canonical(stcsp, RefOracle, text)
        _, rf, af = solve_canonical(stcsp, FrontierModel, text)
        assert af.canonical() == a.canonical(), f"seed {seed}\n{text}"
        assert rf.counters.dominance == r.counters.dominance, f"seed {seed}\n{text}"
        if r.counters.fails == 0:
            assert (rf.n_states, rf.counters.search_nodes) == (r.n_states, r.counters.search_nodes), f"seed {seed}\n{text}"
        n_nontrivial += a.n_live_states > 3
    assert n_nontrivial >= 15  # the generator keeps producing real automata


@pytest.mark.gpu
@pytest.mark.parametrize("block", range(10))
def test_fuzz_engine_matches_reference_restatement(stcsp, RefOracle, block):
    checked = 0
    for seed in range(block * 100, (block + 1) * 100):
        text = random_model(seed)
        _, r, a = solve_canonical(stcsp, RefOracle, text)
        try:
            _, re_, ae = solve_canonical(stcsp, stcsp.Engine, text)
        except stcsp.StcspError as ex:
            assert ex.code == -2, f"seed {seed}: {ex}\n{text}"  # only "domain wider than 32 values" may be refused
            continue
        assert ae.canonical() == a.canonical(), f"seed {seed}\n{text}"
        assert re_.counters.dominance == r.counters.dominance, f"seed {seed}\n{text}"
        if r.counters.fails == 0:
            assert (re_.n_states, re_.counters.search_nodes) == (r.n_states, r.counters.search_nodes), f"seed {seed}\n{text}"
        checked += 1
    assert checked >= 90


@pytest.mark.parametrize("k", [1, 3])
def test_fuzz_frontier_model_other_windows(stcsp, RefOracle, FrontierModel, k):
    """Look-ahead windows other than the reference's default (-k1: no look-ahead at all, -k3)."""
    checked = 0
    for seed in range(200):
        text = random_model(seed)
        _, r, a = solve_canonical(stcsp, RefOracle, text, prefix_k=k, time_limit_s=5)
        if r.truncated:
            continue
        _, rf, af = solve_canonical(stcsp, FrontierModel, text, prefix_k=k)
        assert af.canonical() == a.canonical(), f"k {k} seed {seed}\n{text}"
        assert rf.counters.dominance == r.counters.dominance, f"k {k} seed {seed}\n{text}"
        checked += 1
    assert checked >= 180


@pytest.mark.gpu
@pytest.mark.parametrize("k", [1, 3, 4])
def test_fuzz_engine_other_windows(stcsp, RefOracle, k):
    checked = 0
    for seed in range(150):
        text = random_model(seed)
        _, r, a = solve_canonical(stcsp, RefOracle, text, prefix_k=k, time_limit_s=5)
        if r.truncated:
            continue
        try:
            _, re_, ae = solve_canonical(stcsp, stcsp.Engine, text, prefix_k=k)
        except stcsp.StcspError as ex:
            assert ex.code == -2, f"k {k} seed {seed}: {ex}\n{text}"
            continue
        assert ae.canonical() == a.canonical(), f"k {k} seed {seed}\n{text}"
        assert re_.counters.dominance == r.counters.dominance, f"k {k} seed {seed}\n{text}"
        checked += 1
    assert checked >= 130


@pytest.mark.gpu
def test_fuzz_engine_device_passes(stcsp):
    """The device post-processing against its host twin on random automata (until flags, failing
    branches, several constraint sets), incl. both adversarial passes on every pair of variables."""
    from test_postproc_gpu import host_and_device
    done = 0
    for seed in range(1000, 1150):
        m = stcsp.Model(text=random_model(seed))
        try:
            e = stcsp.Engine(m)
        except stcsp.StcspError:
            continue
        r = e.solve()
        if r.n_edges == 0 or r.n_edges > 5000:
            continue
        host_and_device(stcsp, e, r)
        host_and_device(stcsp, e, r, adv=seed % m.n_vars)
        host_and_device(stcsp, e, r, adv2=(seed % m.n_vars, (seed // 3) % m.n_vars))
        done += 1
    assert done >= 40
