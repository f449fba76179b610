#!/bin/bash
# A/B of two builds of the library on the grid ILU(0) apply (tri_quad_kernel): alternating processes, several rounds.   usage: tq_ab.sh libA libB [rounds=4]
A=$1; B=$2; R=${3:-4}
for r in $(seq $R); do
  for L in $A $B; do
    for g in 128 256 512; do echo -n "$(basename $L) "; KRYST_HIP_LIB=$L python tools/ilu_only.py $g 30 true 2>&1 | grep apply | sed 's/.*WAVE=default: //'; done
  done
done
