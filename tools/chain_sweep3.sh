#!/bin/bash
for inst in "$@"; do
  for heavy in 150000 300000 500000 800000; do
    STCSP_CHAIN_HEAVY=$heavy python tools/chain_sweep.py --one $inst | sed "s/^/heavy=$heavy /"
  done
done
