"""The C-ABI libraries load and export every symbol the headers declare (no compute calls)."""
import ctypes as C
import re
import subprocess
from pathlib import Path

REPO = Path(__file__).resolve().parents[1]


def declared(header):
    text = (REPO / "include" / header).read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(stcsp_[a-z0-9_]+)\s*\(", text)))


def test_engine_library_exports_every_declared_symbol(stcsp):
    subprocess.run(["make", "-C", str(stcsp.CSRC), "libstcsp_hip.so"], check=True, capture_output=True)
    lib = C.CDLL(str(stcsp.CSRC / "libstcsp_hip.so"))  # links libamdhip64: loads without a GPU
    names = declared("stcsp_engine.h")
    assert set(names) == set(stcsp.ENGINE_SYMBOLS)
    for n in names:
        assert hasattr(lib, n), n


def test_sharded_header_symbols_are_exported(stcsp):
    """include/stcsp_sharded.h: the superstep loop and the in-process transport live in libstcsp_hip.so, the RCCL transport in
    libstcsp_rccl.so (which loads without a GPU too: it links librccl and libamdhip64)."""
    subprocess.run(["make", "-C", str(stcsp.CSRC), "libstcsp_hip.so", "libstcsp_rccl.so"], check=True, capture_output=True)
    names = declared("stcsp_sharded.h")
    assert set(names) == set(stcsp.SHARDED_SYMBOLS_HIP) | set(stcsp.SHARDED_SYMBOLS_RCCL)
    hip = C.CDLL(str(stcsp.CSRC / "libstcsp_hip.so"))
    for n in stcsp.SHARDED_SYMBOLS_HIP:
        assert hasattr(hip, n), n
    rccl = C.CDLL(str(stcsp.CSRC / "libstcsp_rccl.so"))
    for n in stcsp.SHARDED_SYMBOLS_RCCL:
        assert hasattr(rccl, n), n
    assert C.sizeof(stcsp.ShardedOptions) == 24 and C.sizeof(stcsp.ShardedStats) == 5 * 8 + 4 * 8


def test_host_library_exports_every_declared_symbol(stcsp):
    lib = stcsp.host_lib()
    names = declared("stcsp_host.h")
    assert set(names) == set(stcsp.HOST_SYMBOLS)
    for n in names:
        assert hasattr(lib, n), n


def test_engine_fails_loudly_without_gpu(stcsp):
    """No CPU fallback: on a machine without a GPU, creating an engine is an error."""
    import torch
    if torch.cuda.is_available():
        return
    import pytest
    m = stcsp.Model.from_name("juggling_b4_f4")
    with pytest.raises(stcsp.StcspError) as e:
        stcsp.Engine(m)
    assert e.value.code == -3  # STCSP_E_DEVICE


def test_abi_struct_sizes(stcsp):
    # mirrors of include/stcsp_engine.h (LP64)
    assert C.sizeof(stcsp.Node) == 24
    assert C.sizeof(stcsp.Options) == 40
    assert C.sizeof(stcsp.Counters) == 8 * 8 + 2 * 8 + 8 + 8 + 24 + 8  # + translation_stops
    assert C.sizeof(stcsp.PostOptions) == 16
    assert C.sizeof(stcsp.PostResult) == 72


def test_automaton_flags_roundtrip(stcsp, RefOracle):
    """stcsp_automaton_import_flags / stcsp_automaton_flags (the hand-over point of the device
    post-processing) on the host: flags exported from one automaton reproduce it in another."""
    m = stcsp.Model(text="var x:[0,1]; var y:[0,1]; x until y;")
    o = RefOracle(m)
    r = o.solve()
    a = o.automaton(r)
    a.traverse()
    valid, final, alive = a.flags()
    assert len(valid) == r.n_states and len(alive) == r.n_edges

    class P:  # shaped like PostResult
        state_valid, state_final, edge_alive = valid, final, alive

    b = o.automaton(r).import_flags(P)
    assert b.flags() == (valid, final, alive)
    assert a.renumber().canonical() == b.renumber().canonical()
