"""solutions.dot writer and the compact binary automaton format,
on oracle results (CPU)."""
import os

import pytest

from conftest import finish


def dot_lines(a, path):
    a.write_dot(str(path))
    return path.read_text().splitlines()


@pytest.mark.parametrize("name", ["juggling_b4_f5", "digitinvader3", "juggling_b4_f5_nosym", "partialorder_10"])
def test_binary_roundtrip(stcsp, RefOracle, tmp_path, name):
    m = stcsp.Model.from_name(name)
    o = RefOracle(m)
    a, _ = finish(o, o.solve())
    p = tmp_path / "a.bin"
    a.write_binary(str(p))
    b = stcsp.Automaton.read_binary(str(p))
    assert b.canonical() == a.canonical()
    assert (b.n_states, b.n_live_states, b.n_live_edges) == (a.n_states, a.n_live_states, a.n_live_edges)
    la, lb = dot_lines(a, tmp_path / "a.dot"), dot_lines(b, tmp_path / "b.dot")
    assert la[:4] == lb[:4]                      # header comment lines + digraph opener
    assert sorted(la) == sorted(lb)              # same vertex / edge lines, printed ids preserved
    # the binary form is several times smaller than the text
    assert os.path.getsize(p) * 2 < os.path.getsize(tmp_path / "a.dot")


def test_binary_empty_and_corrupt(stcsp, RefOracle, tmp_path):
    text = ("var d0:[0,0]; var d1:[0,0]; var d2:[0,0]; var d3:[0,0]; var d4:[0,0]; var e:[0,1]; var p:[0,1]; var s:[0,2]; "
            "first s == 0; next s == if (e eq p) then 0 else 2; s <= 1;")
    m = stcsp.Model(text=text)
    o = RefOracle(m)
    a, adv = finish(o, o.solve(), adversarial="z")
    assert adv == 0 and a.canonical().endswith("EMPTY\n")
    p = tmp_path / "e.bin"
    a.write_binary(str(p))
    b = stcsp.Automaton.read_binary(str(p))
    assert b.canonical() == a.canonical()
    raw = p.read_bytes()
    (tmp_path / "bad1.bin").write_bytes(b"NOTMAGIC" + raw[8:])
    (tmp_path / "bad2.bin").write_bytes(raw[: len(raw) // 2])
    # a header that claims 2^32 states / 2^36 edges in a file of a few hundred bytes must be rejected
    # before anything is allocated from those counts (header: magic 8, u32 x 6, table u64, ns u64, ne u64)
    import struct
    hostile = bytearray(raw)
    struct.pack_into("<QQ", hostile, 8 + 24 + 8, 1 << 32, 1 << 36)
    (tmp_path / "bad3.bin").write_bytes(bytes(hostile))
    for bad in ("bad1.bin", "bad2.bin", "bad3.bin", "missing.bin"):
        with pytest.raises(stcsp.StcspError):
            stcsp.Automaton.read_binary(str(tmp_path / bad))


def test_cli_option_parsing(stcsp, tmp_path):
    """Option handling of the stcsp command line (reference getopt string "b:e:cv:l:stk:m:az",
    src/solver.cpp:211): bad values and unknown letters are rejected before any device is touched."""
    import subprocess
    exe = stcsp.CSRC / "stcsp"
    if not exe.exists():
        subprocess.run(["make", "-C", str(stcsp.CSRC), "stcsp"], check=True, capture_output=True)
    f = tmp_path / "m.csp"
    f.write_text("var x:[0,1];\nx == 1;\n")
    run = lambda *a: subprocess.run([str(exe), *a], capture_output=True, text=True, timeout=60)  # noqa: E731
    r = run("-k0", str(f))
    assert r.returncode == 1 and "Invalid argument" in r.stderr
    r = run("-k", "abc", str(f))
    assert r.returncode == 1 and "Invalid argument" in r.stderr
    r = run("-sq", str(f))
    assert r.returncode == 1 and "Unknown argument: q" in r.stderr
    r = run(str(f), "-k")
    assert r.returncode == 1 and "needs a value" in r.stderr
    r = run()
    assert r.returncode == 0 and "No constraints!" in r.stdout


def test_dot_format_lines(stcsp, RefOracle, tmp_path):
    """Line formats of solverOut / vertexOut / edgeOut (src/solveralgorithm.cpp:709-730,
    src/graph.cpp:41-101) incl. negative values (digitinvader domains start at -1)."""
    m = stcsp.Model.from_name("digitinvader2")
    o = RefOracle(m)
    a, _ = finish(o, o.solve())
    lines = dot_lines(a, tmp_path / "s.dot")
    assert lines[0] == f"# Number of nodes = {a.n_states}"
    assert lines[1].startswith("# ") and lines[2].startswith("# ") and lines[3] == 'digraph "StCSP" {' and lines[-1] == "}"
    import re
    vertex = re.compile(r'^\d+ \[shape=(double)?circle, label="(\d+: (-?\d+(, -?\d+)*)?|0: S)"\];$')
    edge = re.compile(r'^\d+ -> \d+ \[label="-?\d+(, -?\d+)*"\];$')
    body = lines[4:-1]
    assert body and all(vertex.match(l) or edge.match(l) for l in body)
    assert sum(1 for l in body if edge.match(l)) == a.n_live_edges
    assert sum(1 for l in body if vertex.match(l)) == a.n_live_states
    assert any("-1" in l for l in body)
    n_vals = {len(l.split('"')[1].split(", ")) for l in body if edge.match(l)}
    assert n_vals == {m.n_vars}


@pytest.mark.parametrize("name", ["juggling_b4_f5", "digitinvader3", "partialorder_10"])
def test_label_order_makes_files_reproducible(stcsp, RefOracle, FrontierModel, tmp_path, name):
    """Two implementations that number states and order edges differently (depth-first restatement
    vs the frontier algorithm) write byte-identical files after order_by_label()."""
    m = stcsp.Model.from_name(name)
    files = []
    for k, cls in enumerate((RefOracle, FrontierModel)):
        e = cls(m)
        r = e.solve()
        a = e.automaton(r).traverse().order_by_label().renumber()
        a.write_dot(str(tmp_path / f"{k}.dot"))
        a.write_binary(str(tmp_path / f"{k}.bin"))
        files.append(((tmp_path / f"{k}.dot").read_bytes(), (tmp_path / f"{k}.bin").read_bytes()))
    assert files[0][0] == files[1][0]
    assert files[0][1] == files[1][1]
    # and it is still the same automaton
    assert stcsp.Automaton.read_binary(str(tmp_path / "1.bin")).canonical() == a.canonical()


def test_label_order_renumbers_constraint_sets(stcsp, RefOracle, FrontierModel, tmp_path):
    """Four constraint sets (`@` probe): set ids are printed in the order the output walk meets them."""
    text = "var x:[0,3]; var y:[0,3]; first x == 0; next x == (x + 1) % 4; y == x@2;"
    m = stcsp.Model(text=text)
    outs = []
    for cls in (RefOracle, FrontierModel):
        e = cls(m)
        a = e.automaton(e.solve()).traverse().order_by_label().renumber()
        p = tmp_path / f"{cls.__name__}.dot"
        a.write_dot(str(p))
        outs.append(p.read_text())
    assert outs[0] == outs[1]
    import re
    cids = [int(x) for x in re.findall(r'label="(\d+): ', outs[0])]
    seen = []
    for c in cids:
        if c not in seen:
            seen.append(c)
    assert seen == list(range(len(seen))) and len(seen) == 4
