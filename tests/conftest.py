import ctypes as C
import importlib
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

REPO = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(REPO))
sys.path.insert(0, str(REPO / "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")
    config.addinivalue_line("markers", "slow: long CPU run, enabled with STCSP_SLOW=1")


st = importlib.import_module("stcsp-solver_amd")


def _ensure(path: Path, make_dir: Path, target: str):
    if not path.exists():
        subprocess.run(["make", "-C", str(make_dir), target], check=True, capture_output=True)
    return path


@pytest.fixture(scope="session")
def stcsp():
    _ensure(st.CSRC / "libstcsp_host.so", st.CSRC, "libstcsp_host.so")
    _ensure(st.CSRC / "libstcsp_hip.so", st.CSRC, "libstcsp_hip.so")  # hipcc cross-compiles without a GPU
    return st


@pytest.fixture(scope="session")
def oracle_lib():
    """The CPU oracle (tests only)."""
    path = _ensure(REPO / "oracle" / "libstcsp_oracle.so", REPO / "oracle", "libstcsp_oracle.so")
    lib = C.CDLL(str(path))
    st.bind_engine_api(lib, "stcsp_oracle")
    st.bind_engine_api(lib, "stcsp_fmodel")
    lib.stcsp_fmodel_propagate.argtypes = [C.c_void_p, C.c_int, C.c_uint32, C.POINTER(C.c_uint32)]
    return lib


@pytest.fixture(scope="session")
def RefOracle(oracle_lib):
    class RefOracle(st.EngineBase):
        """oracle/ref_dfs.cpp: the reference's algorithm restated."""
        _prefix = "stcsp_oracle"

        def __init__(self, model, **o):
            super().__init__(oracle_lib, model, **o)

    return RefOracle


@pytest.fixture(scope="session")
def FrontierModel(oracle_lib):
    class FrontierModel(st.EngineBase):
        """oracle/frontier_model.cpp: scalar model of the build's frontier algorithm."""
        _prefix = "stcsp_fmodel"

        def __init__(self, model, **o):
            super().__init__(oracle_lib, model, **o)

    return FrontierModel


@pytest.fixture(scope="session")
def golden():
    return json.loads((REPO / "tests" / "golden" / "reference_golden.json").read_text())


def finish(engine, result, adversarial=None):
    """solverSolve's tail (reference src/solveralgorithm.cpp:972-985) on an engine result."""
    a = engine.automaton(result)
    a.traverse()
    adv = None
    if adversarial == "a":
        adv = a.adversarial(5)
    elif adversarial == "z":
        adv = a.adversarial2(5, 6)
    a.renumber()
    return a, adv
