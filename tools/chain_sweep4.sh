#!/bin/bash
for inst in "$@"; do
  for cfg in "4 2 4096" "8 2 4096" "6 3 4096" "4 3 4096" "8 4 4096" "4 2 16384" "8 2 16384" "6 2 8192"; do
    set -- $cfg
    STCSP_CHAIN_SMALL=$1 STCSP_CHAIN_BIG=$2 STCSP_CHAIN_THRESH=$3 python tools/chain_sweep.py --one $inst
  done
done
