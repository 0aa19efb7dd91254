#!/bin/bash
# usage: tools/variant_compare.sh "<instances>" <lib1> <lib2> ...   (libs relative to stcsp-solver_amd/csrc)
insts=$1; shift
for inst in $insts; do
  for lib in "$@"; do
    echo -n "$lib: "
    STCSP_HIP_LIB=stcsp-solver_amd/csrc/$lib python tools/chain_sweep.py --one $inst
  done
done
