#!/bin/bash
# A/B of the sibling hysteresis of tri_quad_kernel (KRYST_ILU_HYST = 1 .. 4, read per apply): alternating processes, several rounds.
# usage: tq_hyst.sh [rounds=3]
R=${1:-3}
cd /root/repo
for r in $(seq $R); do
  for H in 1 2 3 4; do
    for g in 128 256 512; do echo -n "hyst $H "; KRYST_ILU_HYST=$H python tools/ilu_only.py $g 30 true 2>&1 | grep apply | sed 's/ WAVE=default//'; done
  done
done
