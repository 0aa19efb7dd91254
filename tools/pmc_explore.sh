#!/bin/bash
# counter exploration for one workload (separate --pmc passes, kernel-trace only). usage: pmc_explore.sh <workload> "<pass 1 counters>" "<pass 2>" ...
cd "$GRAFT_REPO_ROOT"; R=$PWD; export TMPDIR=/tmp; OUT=$R/gpurun_out/pmc_explore; rm -rf $OUT; mkdir -p $OUT; cd /tmp
W=$1; shift
rocprofv3 --list-avail > $OUT/avail.txt 2>&1 || rocprofv3 -L > $OUT/avail.txt 2>&1
i=0
for pass in "$@"; do
  i=$((i+1))
  rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $OUT/pass$i -- python3 $R/bench.py --workload $W --steps 1 --warmup 0 --no-cpu-baseline --no-other-workloads > $OUT/pass$i.json 2> $OUT/pass$i.err
  echo "== pass $i: $pass"; python3 $R/tools/pmc_summary.py $OUT/pass$i | python3 -c "
import json,sys; d=json.load(sys.stdin); print({k: v['sum'] for k,v in d.items()})"
done
