#!/bin/bash
# partialorder_14 through the C-ABI from a process without PyTorch (ROCm 7.2's own HIP runtime) under a few knobs
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r4
run() { echo "== $*"; env "$@" python3 tools/bench_no_torch.py partialorder_14 20 5 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print({k: (round(v,3) if isinstance(v,float) else v) for k,v in d.items() if k in ('nodes_per_s','ms_per_step','search_ms','export_ms','hip_runtime_version','parity_ok')})"; }
run STCSP_DUMMY=1
run STCSP_PLAN_MIRROR=0
run STCSP_BURST=2
run STCSP_BURST=3
run STCSP_BURST=6
run STCSP_BURST=8
run STCSP_BURST=32
