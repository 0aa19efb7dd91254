#!/bin/bash
for cfg in "4 2 4096" "4 4 4096" "8 4 8192" "8 8 8192" "16 8 16384" "4 2 16384"; do
  set -- $cfg
  echo -n "chain $cfg: "
  STCSP_CHAIN_SMALL=$1 STCSP_CHAIN_BIG=$2 STCSP_CHAIN_THRESH=$3 python tools/synth_ab.py 2 | tail -1
done
