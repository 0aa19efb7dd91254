#!/bin/bash
# usage: tools/variant_compare.sh <instance> <lib1> <lib2> ...   (libs relative to stcsp-solver_amd/csrc)
inst=$1; shift
libs=("$@")
for lib in "${libs[@]}"; do
  for cfg in "4 2 4096" "8 2 4096" "1 1 0"; do
    set -- $cfg
    echo "== $lib chain $cfg"
    STCSP_HIP_LIB=stcsp-solver_amd/csrc/$lib STCSP_CHAIN_SMALL=$1 STCSP_CHAIN_BIG=$2 STCSP_CHAIN_THRESH=$3 python tools/chain_sweep.py --one $inst
  done
done
