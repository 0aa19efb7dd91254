"""Front end (hand-written lexer/parser/normaliser): model shape against the reference's recorded
`var con` columns and header lines, language corner cases, error behaviour."""
import os
from pathlib import Path

import pytest

REF_EXAMPLES = Path("/root/reference/examples")


def test_model_shape_matches_reference_counts(stcsp, golden):
    """`var` / `con` of the reference's stats line (solveralgorithm.cpp:1001) for all 26 examples."""
    for name, g in golden.items():
        m = stcsp.Model.from_name(name)
        assert (m.n_vars, m.n_constraints) == (g["var"], g["con"]), name


def test_normalisation_worked_examples(stcsp):
    # next B0 == ... becomes _V0 == next(B0) + _V0 == if ...
    m = stcsp.Model.from_name("juggling_b4_f4")
    assert m.var_names[:5] == ["A", "B0", "B1", "B2", "B3"]
    assert m.var_names[5:] == [f"_V{i}" for i in range(8)]
    assert m.constraint_string(4) == "_V0 == next(B0)"
    assert m.constraint_string(5).startswith("_V0 == if ((B0 eq 1)) then (A) else ((B0 - 1))")
    # D5 == 0 fby 1 fby 2 fby D5 (digitinvader2): the exact aux-variable / constraint order
    m = stcsp.Model.from_name("digitinvader2")
    got = [m.constraint_string(i) for i in range(5, 17)]
    assert got == ["_V0 == 0", "_V1 == 1", "_V2 == 2", "first(_V3) == first(_V2)", "D5 == next(_V3)", "_V4 == _V3",
                   "first(_V5) == first(_V1)", "_V4 == next(_V5)", "_V6 == _V5", "first(_V7) == first(_V0)",
                   "_V6 == next(_V7)", "D5 == _V7"]
    assert m.var_bounds()[m.var_names.index("_V3")] == (-1, 2)


def test_normalisation_fixtures(stcsp):
    """Vectors that do not come from the product: tests/golden/frontend_normalised.json holds the normalised
    constraint lists and auxiliary-variable bounds derived by hand from the reference's constraintNormalise
    (src/solveralgorithm.cpp:60-332) for its quirk cases (`not` bounds, @ after next, nested fby, the sign cases
    of `*`, ...). The oracle and the engine both consume what this front end produces, so this is what keeps the
    fuzz agreement from being front-end-against-itself."""
    import json
    doc = json.loads((Path(__file__).resolve().parent / "golden" / "frontend_normalised.json").read_text())
    assert len(doc["cases"]) >= 20
    for case in doc["cases"]:
        m = stcsp.Model(text=case["text"])
        got = [m.constraint_string(i) for i in range(m.n_constraints)]
        assert got == case["constraints"], case["name"]
        aux = [[n, lo, hi] for n, (lo, hi) in zip(m.var_names, m.var_bounds()) if n.startswith("_V")]
        assert aux == case["aux"], case["name"]


def test_lexer_corner_cases(stcsp):
    ok = "var x:[0,3]; // comment\nvar y : [ -1 , 2 ];\n' quote comment\n/* block */ x - 1 == y; x <= 3;"
    m = stcsp.Model(text=ok)
    assert m.n_vars == 2 and m.n_constraints == 2
    # `x -1` lexes as IDENT CONST(-1): syntax error (stcsp.l:77), reported with the line number
    with pytest.raises(stcsp.StcspError) as e:
        stcsp.Model(text="var x:[0,3];\nx -1 == 0;")
    assert "Line 2: syntax error" in str(e.value)
    # keywords only win at equal length: `lta` is an identifier
    m = stcsp.Model(text="var lta:[0,1]; lta == 1;")
    assert m.var_names == ["lta"]
    # else-arm is a unary_expression: `else x + 1` parses as (if..else x) + 1 (stcsp.y:165)
    m = stcsp.Model(text="var x:[0,3]; var y:[0,9]; y == if x eq 0 then 1 else x + 1;")
    assert m.constraint_string(0) == "y == (if ((x eq 0)) then (1) else (x) + 1)"
    # fby is right associative, @ binds looser than fby
    m = stcsp.Model(text="var x:[0,3]; var y:[0,3]; y == x@2;")
    assert m.n_vars == 3 and m.constraint_string(0) == "_V0 == (x @ 2)"


def test_error_behaviour(stcsp):
    with pytest.raises(stcsp.StcspError):           # obj parses but is rejected (solver.cpp:154-156)
        stcsp.Model(text="var a:[0,1]; obj a;")
    with pytest.raises(stcsp.StcspError) as e:      # undefined identifier (solver.cpp:33-36)
        stcsp.Model(text="var a:[0,1]; b == 1;")
    assert "has not been defined" in str(e.value)
    with pytest.raises(stcsp.StcspError) as e:      # empty domain (variable.cpp:15-18)
        stcsp.Model(text="var a:[2,1];")
    assert "Invalid domain" in str(e.value)
    with pytest.raises(stcsp.StcspError):
        stcsp.Model(text="var a:[0,1]; a == ;")
    # tautologies are dropped at build time (solveralgorithm.cpp:19), aux constraints are not
    m = stcsp.Model(text="var a:[0,1]; 1 == 1; a == next 1;")
    assert m.n_constraints == 1


@pytest.mark.skipif(not REF_EXAMPLES.exists(), reason="reference tree not present (GPU box)")
def test_generated_instances_equal_reference_examples(stcsp):
    """instances.py regenerates the reference's examples: identical models (same variables,
    bounds, arrays and constraint trees in the same order) as parsing the shipped files."""
    for name in stcsp.instances.REFERENCE_EXAMPLES:
        a = stcsp.Model.from_name(name)
        b = stcsp.Model(path=str(REF_EXAMPLES / f"{name}.csp"))
        assert a.var_names == b.var_names and a.var_bounds() == b.var_bounds(), name
        assert a.n_constraints == b.n_constraints, name
        for i in range(a.n_constraints):
            assert a.constraint_string(i) == b.constraint_string(i), (name, i)
