#!/bin/bash
# Round-end profile of bench.py's workload: kernel-trace stats + separate PMC passes (never combined
# with other trace domains). Outputs under gpurun_out/prof_final/.
set -e
cd "$GRAFT_REPO_ROOT"
R=$PWD
export TMPDIR=/tmp
OUT=$R/gpurun_out/prof_final
mkdir -p $OUT
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/stats_bench.json 2> $OUT/stats.err
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVES SQ_WAVE_CYCLES" "SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE"; do
  name=$(echo $pass | cut -d' ' -f1)
  rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $OUT/pmc_$name -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $OUT/pmc_$name.json 2> $OUT/pmc_$name.err
  echo "pass $name done"
done
find $OUT -name "*.csv" | head -30
