"""graphTraverse / adversarialTraverse / adversarialTraverse2 on the
device (stcsp_engine_postprocess) against their host twins (postproc.cpp, themselves pinned to the
reference's -a / -z probes in test_engine_gpu.py / test_oracle.py) and against the golden values.
Run on the GPU box: pytest -m gpu."""
import importlib
import json
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu

PROBES = json.loads((Path(__file__).resolve().parent / "golden" / "reference_probes.json").read_text())


def all_examples():
    return importlib.import_module("stcsp-solver_amd").instances.REFERENCE_EXAMPLES


def host_and_device(stcsp, e, r, adv=-1, adv2=None):
    """Run the passes twice over the same exported automaton: host twin and device kernels."""
    h = e.automaton(r)
    h.traverse()
    h1 = h.adversarial(adv) if adv >= 0 else -1
    h2 = h.adversarial2(*adv2) if adv2 else -1
    post = e.postprocess(adversarial=adv, adversarial2=adv2)
    assert (post.n_states, post.n_edges) == (r.n_states, r.n_edges)
    d = e.automaton(r).import_flags(post)
    assert (post.adver1, post.adver2) == (h1, h2)
    hv, hf, ha = h.flags()
    dv, df, da = d.flags()
    assert hv == dv and hf == df            # the fixpoints are order-independent: flags bit-identical
    if not adv2:
        assert ha == da                     # deterministic edge removal
    else:
        # adversarialTraverse2 also thins the edges of states it later invalidates, in worklist
        # order; those states are unreachable afterwards -- compare the edges of kept states
        src = [r.edge_src[k] for k in range(r.n_edges)]
        assert all(ha[k] == da[k] for k in range(r.n_edges) if hv[src[k]])
    h.renumber()
    d.renumber()
    assert d.canonical() == h.canonical()
    return h, d, post


@pytest.mark.parametrize("name", all_examples())
def test_device_traverse_matches_golden(stcsp, golden, name):
    m = stcsp.Model.from_name(name)
    e = stcsp.Engine(m)
    r = e.solve()
    post = e.postprocess()
    a = e.automaton(r).import_flags(post).renumber()
    g = golden[name]
    assert (a.n_live_states, a.n_live_edges) == (g["states"], g["edges"])
    assert a.canonical_sha256() == g["canonical_sha256"]
    assert (post.adver1, post.adver2) == (-1, -1)


@pytest.mark.parametrize("name", ["juggling_b4_f5", "juggling_b5_f6", "digitinvader3", "digitinvader5", "partialorder_10",
                                  "juggling_b4_f5_nosym"])
def test_device_adversarial_passes_match_host(stcsp, name):
    """-a / -z with the reference's hard-coded indices (5; 5,6) and with every other choice of
    variables that has a domain: same flags, same canonical automaton as the host twin."""
    m = stcsp.Model.from_name(name)
    e = stcsp.Engine(m)
    r = e.solve()
    n = m.n_vars
    host_and_device(stcsp, e, r)
    for v in sorted({5, 0, 1, 2, n - 1, n // 2}):
        host_and_device(stcsp, e, r, adv=v)
    for op, ava in [(5, 6), (0, 1), (1, 0), (2, n - 1), (n - 1, 3), (4, 4)]:
        host_and_device(stcsp, e, r, adv2=(op, ava))
    host_and_device(stcsp, e, r, adv=5, adv2=(5, 6))   # the reference's `-a -z`
    # export() again after the passes: the automaton itself is untouched
    r2 = e.export()
    assert (r2.n_states, r2.n_edges) == (r.n_states, r.n_edges)


def test_device_passes_reference_probes(stcsp):
    """The reference's own -a / -z / until probes."""
    p = PROBES["adversarial"]
    m = stcsp.Model(text=p["text"])
    e = stcsp.Engine(m)
    r = e.solve()
    _, d, post = host_and_device(stcsp, e, r, adv=5)
    assert post.adver1 == p["adver1"] and (d.n_live_states, d.n_live_edges) == (p["adver1_live_states"], p["adver1_live_edges"])
    _, d, post = host_and_device(stcsp, e, r, adv2=(5, 6))
    assert post.adver2 == p["adver2"] and d.canonical().endswith("EMPTY\n")
    p = PROBES["until"]
    m = stcsp.Model(text=p["text"])
    e = stcsp.Engine(m)
    r = e.solve()
    _, d, post = host_and_device(stcsp, e, r)
    assert (d.n_live_states, d.n_live_edges) == (p["live_states"], p["live_edges"]) and r.root_final == p["root_final"]
    assert post.rounds[0] >= 1


@pytest.mark.parametrize("text", [
    # until chains: validity has to travel backwards over several edges (multi-round traverse)
    "var x:[0,3]; var y:[0,1]; var g:[0,1]; first x == 0; next x == if (x lt 3) then x + 1 else x; y == (x eq 3); g until y;",
    "var x:[0,5]; var y:[0,1]; var g:[0,1]; first x == 0; next x >= x; next x <= x + 1; y == (x eq 5); g until y; g == 1;",
    # a game where the opponent's choice e decides: the adversarial fixpoints cascade over states
    "var a:[0,0]; var b:[0,0]; var c:[0,0]; var d:[0,0]; var f:[0,0]; var e:[0,1]; var p:[0,2]; var s:[0,4]; "
    "first s == 0; next s == if (e eq (p % 2)) then (if (s lt 4) then s + 1 else s) else s; s <= 3;",
])
def test_device_passes_cascades(stcsp, text):
    m = stcsp.Model(text=text)
    e = stcsp.Engine(m)
    r = e.solve()
    host_and_device(stcsp, e, r)
    for v in range(m.n_vars):
        host_and_device(stcsp, e, r, adv=v)
    host_and_device(stcsp, e, r, adv2=(5, 6) if m.n_vars > 6 else (0, 1))
    host_and_device(stcsp, e, r, adv=1, adv2=(0, 1))


@pytest.mark.parametrize("shape", [(16, 8, 88, 4, 4), (16, 8, 80, 4, 5)])
def test_device_passes_synthetic(stcsp, shape):
    """Failing branches + wide domains (config 4 family)."""
    inst = importlib.import_module("stcsp-solver_amd").instances
    n, d, mm, s, seed = shape
    m = stcsp.Model(text=inst.synthetic(n, d, mm, s, seed))
    e = stcsp.Engine(m)
    r = e.solve()
    host_and_device(stcsp, e, r)
    for v in (0, 3, n - 1):
        host_and_device(stcsp, e, r, adv=v)
    host_and_device(stcsp, e, r, adv2=(0, 1))
    host_and_device(stcsp, e, r, adv2=(n - 1, 2))


def test_device_postprocess_call_order(stcsp):
    """STCSP_E_STATE before a solve / with F_NO_EXPORT until export() ran; bad indices are rejected."""
    m = stcsp.Model.from_name("juggling_b4_f4")
    e = stcsp.Engine(m, flags=stcsp.F_NO_EXPORT)
    with pytest.raises(stcsp.StcspError) as ex:
        e.postprocess()
    assert ex.value.code == -6
    e.solve()
    with pytest.raises(stcsp.StcspError):
        e.postprocess()
    r = e.export()
    e.postprocess()
    with pytest.raises(stcsp.StcspError) as ex:
        e.postprocess(adversarial=m.n_vars)
    assert ex.value.code == -1
    host_and_device(stcsp, e, r, adv=5)


@pytest.mark.parametrize("name", ["juggling_b4_f6", "digitinvader4", "partialorder_11"])
def test_engine_files_are_reproducible(stcsp, RefOracle, tmp_path, name):
    """State numbering and edge order on the GPU depend on wavefront scheduling; after
    order_by_label() the written files do not: identical bytes for different launch batch sizes,
    and identical to what the depth-first CPU restatement writes."""
    m = stcsp.Model.from_name(name)
    blobs = []
    for k, opts in enumerate(({}, {"batch_nodes": 64}, {"batch_nodes": 1024})):
        e = stcsp.Engine(m, **opts)
        r = e.solve()
        a = e.automaton(r).import_flags(e.postprocess()).order_by_label().renumber()
        a.write_dot(str(tmp_path / f"{k}.dot"))
        a.write_binary(str(tmp_path / f"{k}.bin"))
        blobs.append(((tmp_path / f"{k}.dot").read_bytes(), (tmp_path / f"{k}.bin").read_bytes()))
    assert blobs[0] == blobs[1] == blobs[2]
    o = RefOracle(m)
    ao = o.automaton(o.solve()).traverse().order_by_label().renumber()
    ao.write_dot(str(tmp_path / "o.dot"))
    assert (tmp_path / "o.dot").read_bytes() == blobs[0][0]
