#!/bin/bash
for inst in "$@"; do
  for cfg in "1 1 0" "4 2 4096" "4 1 4096" "4 4 4096"; do
    set -- $cfg
    STCSP_CHAIN_SMALL=$1 STCSP_CHAIN_BIG=$2 STCSP_CHAIN_THRESH=$3 python tools/chain_sweep.py --one $inst
  done
done
