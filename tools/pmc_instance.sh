#!/bin/bash
# instruction mix of one solve of an instance: tools/pmc_instance.sh <instance>
cd "$GRAFT_REPO_ROOT"; R=$PWD; export TMPDIR=/tmp; OUT=$R/gpurun_out/pmc_$1; rm -rf $OUT; mkdir -p $OUT; cd /tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $OUT/raw -- python3 $R/bench.py --workload $1 --steps 1 --warmup 0 --no-cpu-baseline --no-other-workloads > $OUT/bench.json 2> $OUT/err.txt
python3 $R/tools/pmc_summary.py $OUT/raw
python3 -c "
import json;d=json.load(open('$OUT/bench.json'));print('nodes', d['config']['nodes_per_step'], 'ms', d['ms_per_step'])"
