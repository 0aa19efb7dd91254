"""Tests-side entry of bench.py for boxes without GPUs: the same launcher, rendezvous, sharded pipeline and N>1 output
line, driven with the CPU frontier model (oracle/frontier_model.cpp) over gloo.  The stand-in is injected from HERE
(bench.HOOKS); bench.py itself contains no way of replacing the HIP engine."""
import ctypes as C
import importlib
import sys
from pathlib import Path

REPO = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(REPO))

import bench  # noqa: E402

st = importlib.import_module("stcsp-solver_amd")
_lib = C.CDLL(str(REPO / "oracle" / "libstcsp_oracle.so"))
st.bind_engine_api(_lib, "stcsp_fmodel")


class FModel(st.EngineBase):
    _prefix = "stcsp_fmodel"

    def __init__(self, model, **o):
        super().__init__(_lib, model, **o)


bench.HOOKS.update(engine=FModel, backend="gloo", entry=Path(__file__).resolve(),
                   label="TEST ONLY: oracle/frontier_model.cpp over gloo (launcher / pipeline check without GPUs; not a measurement of the product)")

if __name__ == "__main__":
    bench.main()
