#!/bin/bash
# end-to-end wall time of the command line on generated instances (parse -> engine -> passes -> dot)
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/cli && cd gpurun_out/cli
for inst in partialorder_14 digitinvader9 juggling_b5_f6; do
  python3 -c "
import importlib,sys
sys.path.insert(0,'../..')
inst=importlib.import_module('stcsp-solver_amd.instances')
open('$inst.csp','w').write(inst.by_name('$inst'))"
  for flags in "" "-s" "-s --binary=$inst.bin"; do
    s=$(date +%s.%N)
    ../../stcsp-solver_amd/csrc/stcsp $flags $inst.csp > out.txt 2> err.txt
    e=$(date +%s.%N)
    echo "$inst [$flags] wall $(echo "$e - $s" | bc) s : $(cat out.txt | tr '\n' ' ')"
  done
  ls -la solutions.dot $inst.bin 2>/dev/null | awk '{print $5, $9}'
  rm -f solutions.dot $inst.bin
done
