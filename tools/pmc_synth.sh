#!/bin/bash
# instruction mix, wait / busy shares and L2 traffic of the synthetic 64 x 32 instance (time-boxed): separate --pmc passes
cd "$GRAFT_REPO_ROOT"; R=$PWD; export TMPDIR=/tmp; OUT=$R/gpurun_out/pmc_synth; rm -rf $OUT; mkdir -p $OUT; cd /tmp
BOX=${1:-0.3}
i=0
for pass in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_WAVE_CYCLES" \
            "SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
            "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA_RDREQ_sum" \
            "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $OUT/pass$i -- python3 $R/tools/synth_bench.py $BOX > $OUT/pass$i.json 2> $OUT/pass$i.err
  echo "== pass $i: $pass"; python3 $R/tools/pmc_summary.py $OUT/pass$i | python3 -c "
import json,sys; d=json.load(sys.stdin); print({k: v['sum'] for k,v in d.items()})"; tail -1 $OUT/pass$i.json | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('nodes', d['nodes'], 'launches', d['launches'], 'Mnodes/s', d['nodes_per_s']/1e6, d['per_node'])"
done
