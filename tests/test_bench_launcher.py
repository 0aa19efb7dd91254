"""bench.py's own N-rank launcher (`python bench.py --gpus N` without torchrun): the parent starts N rank processes
BEFORE touching any GPU, waits, prints rank 0's line. On CPU the ranks run the oracle's frontier model over gloo,
injected by tests/bench_cpu_entry.py (bench.py has no switch for it), so the launcher, the rendezvous, the sharded
pipeline, the N>1 output line and the exit codes are exercised without GPUs."""
import json
import subprocess
import sys
from pathlib import Path

import pytest

REPO = Path(__file__).resolve().parents[1]


def run_bench(*argv, timeout=300, entry="bench.py"):
    return subprocess.run([sys.executable, str(REPO / entry), *argv], capture_output=True, text=True, timeout=timeout)


CPU_ENTRY = "tests/bench_cpu_entry.py"


def last_json(stdout):
    lines = [ln for ln in stdout.splitlines() if ln.startswith("{")]
    assert lines, stdout
    return json.loads(lines[-1])


@pytest.mark.parametrize("world", [2, 3])
def test_bench_starts_its_own_ranks(oracle_lib, golden, world):
    r = run_bench("--gpus", str(world), "--workload", "partialorder_10", "--steps", "1", "--warmup", "0",
                  "--scalable-workload", "juggling_b4_f5", "--synthetic-seconds", "0", entry=CPU_ENTRY)
    assert r.returncode == 0, r.stderr[-2000:]
    d = last_json(r.stdout)
    assert d["n_gpus"] == world
    assert d["config"]["sharding"] == f"state-owner x{world}"
    assert d["parity"]["ok"] and d["parity"]["canonical_sha256"] == golden["partialorder_10"]["canonical_sha256"]
    sh = d["config"]["sharded"]
    assert len(sh["rank_search_nodes_per_step"]) == world and all(n > 0 for n in sh["rank_search_nodes_per_step"])
    assert sum(sh["rank_search_nodes_per_step"]) == d["config"]["nodes_per_step"] == 55636
    assert sum(sh["nodes_donated"]) == sum(sh["nodes_adopted"])
    extra = sh["scalable_workloads"]["juggling_b4_f5"]
    assert extra["nodes"] == 327 and len(extra["rank_search_nodes"]) == world
    assert "TEST ONLY" in d["config"]["engine"]


def test_bench_time_boxed_synthetic_through_the_sharded_pipeline(oracle_lib):
    """The N>1 line carries the time-boxed synthetic workload through solve_sharded (here a 24 x 8 member of the family
    that never reaches a leaf either: the CPU stand-in manages ~3 k nodes/s on 64 x 32): every rank gets work."""
    r = run_bench("--gpus", "2", "--workload", "juggling_b4_f5", "--steps", "1", "--warmup", "0",
                  "--scalable-workload", "", "--synthetic-seconds", "20.0", "--synthetic-shape", "24,8,125,4,7",
                  "--budget-rounds", "1", "--share-per-rank", "2", entry=CPU_ENTRY)
    assert r.returncode == 0, r.stderr[-2000:]
    d = last_json(r.stdout)
    s = d["config"]["sharded"]["scalable_workloads"]["synthetic"]
    assert s["time_box_s"] == 20.0 and s["leaves"] == 0
    assert all(n > 0 for n in s["rank_search_nodes"]), s
    assert sum(s["nodes_donated"]) == sum(s["nodes_adopted"]) > 0


def test_bench_refuses_more_gpus_than_visible():
    """No GPU in the CPU container: --gpus 2 must exit non-zero with a message (not print an n_gpus = 1 line)."""
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("this box has the GPUs")
    r = run_bench("--gpus", "2", "--steps", "1", "--warmup", "0", timeout=120)
    assert r.returncode != 0
    assert "sees" in r.stderr and not any(ln.startswith("{") for ln in r.stdout.splitlines())


def test_bench_world_mismatch_is_an_error():
    import os
    r = subprocess.run([sys.executable, str(REPO / CPU_ENTRY), "--gpus", "1"], capture_output=True, text=True, timeout=120,
                       env=dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0"))
    assert r.returncode != 0 and "WORLD_SIZE" in r.stderr


def test_bench_has_no_switch_that_replaces_the_engine():
    """The benchmark script must not be able to put the checker in the product's place by itself."""
    src = (REPO / "bench.py").read_text()
    assert "--test-engine" not in src and "fmodel" not in src
    r = run_bench("--test-engine", "fmodel", timeout=120)
    assert r.returncode != 0 and "unrecognized arguments" in r.stderr
