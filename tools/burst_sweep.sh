#!/bin/bash
# launches per burst under both HIP runtimes: bench.py (PyTorch's bundled runtime) and tools/bench_no_torch.py (/opt/rocm's)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r4
for b in 8 16 32 64; do
  echo "== torch runtime, STCSP_BURST=$b"
  STCSP_BURST=$b python3 bench.py --no-cpu-baseline --no-other-workloads | python3 -c "
import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(round(d['value']/1e6,1), 'M  ms/step', round(d['ms_per_step'],3), 'search', round(d['search_ms'],3), 'export', round(d['export_ms'],3), d['parity']['ok'])"
  echo "== rocm 7.2 runtime, STCSP_BURST=$b"
  STCSP_BURST=$b python3 tools/bench_no_torch.py partialorder_14 20 5 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print(round(d['nodes_per_s']/1e6,1), 'M  ms/step', round(d['ms_per_step'],3), 'search', round(d['search_ms'],3), 'export', round(d['export_ms'],3), d['parity_ok'])"
done
