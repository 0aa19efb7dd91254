"""Generators for the stCSP benchmark instance families.

The reference ships 26 example inputs (`examples/*.csp`: partialorder_{10..14},
juggling_b{4,5,6}_f{4,5,6}[_nosym], digitinvader{1..9}).  They are three parametric model
families; this module regenerates them from their parameters so that tests and the bench run on
a GPU box where the reference tree does not exist, and so that larger members of each family
(e.g. partialorder_16) can be produced.  `tests/test_frontend.py::test_generated_instances_equal_reference_examples` checks, when
/root/reference is present, that every generated text lexes to exactly the same token stream as
the shipped example of the same name (same statements, same order => same model, same
constraint queue order).

Also here: the synthetic "N vars x |D|, random binary table constraints" family that
BASELINE.json's config 4 names, with a self-contained
splitmix64 generator so every implementation regenerates the same file from the same seed.
"""
from __future__ import annotations

import re

# ------------------------------------------------------------------ partialorder_N
# "first giveTo < c": the shipped files use c = 4,5,5,6,6 for N = 10..14
_PO_FIRST_BOUND = {10: 4, 11: 5, 12: 5, 13: 6, 14: 6}


def partialorder(n: int, first_bound: int | None = None) -> str:
    if first_bound is None:
        first_bound = _PO_FIRST_BOUND.get(n, (n - 2) // 2)
    out = ["var succ : [0, 1];", f"var giveTo : [0, {n - 1}];"]
    out += [f"var seen{i} : [0, 1];" for i in range(n)]
    out += ["", f"first giveTo < {first_bound};"]
    for i in range(n):
        out += [f"first seen{i} == 0;", f"next seen{i} == seen{i} or (giveTo eq {i});"]
    out += ["", "first succ == 0;", ""]
    out += ["succ >= (" + " and ".join(f"seen{i}" for i in range(n)) + ");", "next succ >= succ;"]
    return "\n".join(out) + "\n"


# ------------------------------------------------------------------ juggling_bB_fF[_nosym]
def juggling(balls: int, maxh: int, nosym: bool = False, first_last: bool | None = None) -> str:
    """B balls, throw heights 0..F.  The shipped juggling_b4_f6.csp states its `first`
    (symmetry-breaking) constraints last; every other file states them first."""
    if first_last is None:
        first_last = (balls, maxh, nosym) == (4, 6, False)
    decl = [f"var A : [0, {maxh}];"] + [f"var B{i} : [0, {maxh}];" for i in range(balls)]
    first = ["first B0 == 1;"] + [f"first B{i} < first B{i + 1};" for i in range(balls - 1)]
    nxt = [f"next B{i} == if B{i} eq 1 then A else (B{i} - 1);" for i in range(balls)]
    diff = [f"B{i} != B{j};" for i in range(balls) for j in range(i + 1, balls)]
    last = ["A == if B0 eq 1 then next B0"]
    last += [f"else if B{i} eq 1 then next B{i}" for i in range(1, balls)]
    last += ["else 0;"]
    blocks = [decl]
    if not nosym and not first_last:
        blocks.append(first)
    blocks += [nxt, diff, last]
    if not nosym and first_last:
        blocks.append(first)
    return "\n\n".join("\n".join(b) for b in blocks) + "\n"


# ------------------------------------------------------------------ digitinvaderN
def digitinvader(n: int, slots: int = 6) -> str:
    out = [f"var I : [0, {n}];"]
    out += [f"var D{i} : [-1, {n}];" for i in range(slots)]
    out += [f"var A{i} : [0, 1];" for i in range(slots)]
    out += ["var MISS : [0, 1];", "var GAMEOVER : [0, 1];", ""]
    out += [f"first D{i} == -1;" for i in range(slots - 1)]
    out += [f"D{slots - 1} == " + " fby ".join(str(v) for v in range(n + 1)) + f" fby D{slots - 1};"]
    for i in range(slots):
        out += [f"A{i} == " + " and ".join([f"I ne D{j}" for j in range(i)] + [f"I eq D{i}"]) + ";"]
    out += ["", "MISS == (" + " + ".join(f"A{i}" for i in range(slots)) + ") eq 0;", "GAMEOVER == D0 ne -1 and MISS;", ""]
    for i in range(slots - 1):
        cond = " or ".join(["MISS"] + [f"A{j}" for j in range(i + 1)])
        out += [f"next D{i} == if GAMEOVER then -1 else if {cond} then D{i + 1} else D{i};"]
    return "\n".join(out)


# ------------------------------------------------------------------ by-name access
def by_name(name: str) -> str:
    """Text of the instance the reference ships as examples/<name>.csp (or a larger member of
    the same family, e.g. partialorder_16)."""
    name = name[:-4] if name.endswith(".csp") else name
    m = re.fullmatch(r"partialorder_(\d+)", name)
    if m:
        return partialorder(int(m.group(1)))
    m = re.fullmatch(r"juggling_b(\d+)_f(\d+)(_nosym)?", name)
    if m:
        return juggling(int(m.group(1)), int(m.group(2)), bool(m.group(3)))
    m = re.fullmatch(r"digitinvader(\d+)", name)
    if m:
        return digitinvader(int(m.group(1)))
    m = re.fullmatch(r"synthetic_n(\d+)_d(\d+)_m(\d+)_s(\d+)_seed(\d+)", name)
    if m:
        n, d, mm, s, seed = (int(x) for x in m.groups())
        return synthetic(n, d, mm, s, seed)
    raise KeyError(name)


REFERENCE_EXAMPLES = (
    [f"partialorder_{n}" for n in range(10, 15)]
    + [f"juggling_b{b}_f{f}{s}" for (b, f) in [(4, 4), (4, 5), (4, 6), (5, 5), (5, 6), (6, 6)] for s in ("", "_nosym")]
    + [f"digitinvader{n}" for n in range(1, 10)]
)


# ------------------------------------------------------------------ synthetic random binary stCSP
class SplitMix64:
    """Self-contained counter-based RNG (not a language-library RNG) so the same seed gives the
    same file in every implementation."""

    def __init__(self, seed: int):
        self.s = seed & 0xFFFFFFFFFFFFFFFF

    def next(self) -> int:
        self.s = (self.s + 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
        z = self.s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
        return z ^ (z >> 31)

    def below(self, n: int) -> int:
        return self.next() % n


def synthetic(n_vars: int, dom: int, n_point: int, n_next: int, seed: int, tightness_permille: int = 300) -> str:
    """N variables x0..x{N-1} : [0, dom-1]; `n_point` extensional binary constraints
    `T[xa*dom + xb] == 1;` on distinct random pairs and `n_next` stream constraints
    `T[xa*dom + next xb] == 1;`, each table with `tightness_permille`/1000 forbidden tuples.
    All expressible in the reference DSL (arr lookup, stcsp.y:171; solveralgorithm.cpp:344-352)."""
    rng = SplitMix64(seed)
    out = [f"var x{i} : [0, {dom - 1}];" for i in range(n_vars)]
    pairs = set()
    cons = []

    def table():
        return [0 if rng.below(1000) < tightness_permille else 1 for _ in range(dom * dom)]

    k = 0
    while len(cons) < n_point:
        a, b = rng.below(n_vars), rng.below(n_vars)
        if a == b or (a, b) in pairs or (b, a) in pairs:
            continue
        pairs.add((a, b))
        out.append(f"arr T{k} : {{" + ", ".join(map(str, table())) + "};")
        cons.append(f"T{k}[x{a} * {dom} + x{b}] == 1;")
        k += 1
    used_next = set()
    while len(used_next) < n_next:
        a, b = rng.below(n_vars), rng.below(n_vars)
        if b in used_next:
            continue
        used_next.add(b)
        out.append(f"arr T{k} : {{" + ", ".join(map(str, table())) + "};")
        cons.append(f"T{k}[x{a} * {dom} + next x{b}] == 1;")
        k += 1
    return "\n".join(out + [""] + cons) + "\n"
