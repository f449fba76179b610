#!/bin/bash
# Copy what tools/profile_round.sh left under gpurun_out/round into profiles/rNN (run here, after the gpurun call).  usage: collect_round.sh r04
P=profiles/${1:-r04}; O=gpurun_out/round
cp $O/bench_n1.json $O/bench_default_kernel_by_size.csv $O/configs_256.jsonl $O/ilu_general.jsonl $P/
find $O/bench_prof -name "*kernel_stats.csv" -exec cp {} $P/bench_default_kernel_stats.csv \;
for f in spmv256_default spmv256_plain spmv256_varcoef spmv512_default spmv512_plain spmv512_varcoef; do cp $O/${f}_traffic.json $P/; done
cp $O/spmv_traffic.json profiles/spmv_traffic.json
find $O/ilu_prof_256 -name "*kernel_stats.csv" -exec cp {} $P/ilu_apply256_kernel_stats.csv \;
find $O/ilu_prof_512 -name "*kernel_stats.csv" -exec cp {} $P/ilu_apply512_kernel_stats.csv \;
find $O/box_prof_96 -name "*kernel_stats.csv" -exec cp {} $P/box_apply96_kernel_stats.csv \;
