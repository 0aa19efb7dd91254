#!/bin/bash
# Round-4 profiles: rocprofv3 kernel-trace stats + separate PMC passes (never combined with other trace domains) for the
# headline workload (partialorder_14), digitinvader9 and the synthetic 64 x 32 instance. New in round 4: the synthetic PMC passes
# are NODE-bounded (max_search_nodes = 40 M per solve) instead of time-boxed: every pass expands about the same nodes and is
# divided by its own engine-counted total (tools/synth_bench.py nodes_all_solves), so the per-node figures are reproducible. Raw output under
# gpurun_out/prof_r04/; tools/profile_r04_summaries.py turns it into the files committed under profiles/.
set -e
cd "$GRAFT_REPO_ROOT"
R=$PWD
export TMPDIR=/tmp
OUT=$R/gpurun_out/prof_r04
rm -rf $OUT; mkdir -p $OUT
cd /tmp
B="python3 $R/bench.py --no-cpu-baseline --no-other-workloads"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_p14 -- $B --steps 5 --warmup 2 > $OUT/stats_p14.json 2> $OUT/stats_p14.err
echo "stats p14 done"
# the same with the export's D2H traffic out of the way (under the tracer the copies of the streaming export run as blit
# kernels on the CUs, which slows k_expand): the kernel by itself
STCSP_STREAM_EXPORT=0 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_p14_nostream -- $B --steps 5 --warmup 2 > $OUT/stats_p14_nostream.json 2> $OUT/stats_p14_nostream.err
echo "stats p14 (no streaming) done"
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_WAVE_CYCLES" "SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE"; do
  name=$(echo $pass | cut -d' ' -f1)
  rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $OUT/pmc_$name -- $B --steps 1 --warmup 0 > $OUT/pmc_$name.json 2> $OUT/pmc_$name.err
  echo "pass $name done"
done
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_d9 -- $B --workload digitinvader9 --steps 3 --warmup 1 > $OUT/stats_d9.json 2> $OUT/stats_d9.err
echo "stats d9 done"
for pass in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_WAVE_CYCLES" "SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE"; do
  name=$(echo $pass | cut -d' ' -f1)
  rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $OUT/pmc_d9_$name -- $B --workload digitinvader9 --steps 1 --warmup 0 > $OUT/pmc_d9_$name.json 2> $OUT/pmc_d9_$name.err
  echo "d9 pass $name done"
done
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_synth -- python3 $R/tools/synth_bench.py 2.0 > $OUT/stats_synth.json 2> $OUT/stats_synth.err
echo "stats synth done"
for pass in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_WAVE_CYCLES" "SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  name=$(echo $pass | cut -d' ' -f1)
  rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $OUT/pmc_synth_$name -- python3 $R/tools/synth_bench.py --nodes 40000000 > $OUT/pmc_synth_$name.json 2> $OUT/pmc_synth_$name.err
  echo "synth pass $name done"
done
find $OUT -name "*_kernel_stats.csv" | head
