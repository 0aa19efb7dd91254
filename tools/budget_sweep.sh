#!/bin/bash
for inst in digitinvader9 digitinvader7 digitinvader5; do
  for b in 256 1024 4096 16384; do
    echo -n "bitmap budget $b: "; STCSP_BUDGET_BITMAP=$b python tools/chain_sweep.py --one $inst
  done
done
for inst in juggling_b6_f6_nosym juggling_b5_f6 juggling_b4_f6; do
  for b in 64 256 1024; do
    echo -n "code budget $b: "; STCSP_BUDGET_CODE=$b python tools/chain_sweep.py --one $inst
  done
done
