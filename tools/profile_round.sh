#!/bin/bash
# Evidence for profiles/: PMC traffic of the SpMV kernel (separate --pmc passes), kernel-trace stats of the default
# bench.py command, the bench line itself and the per-config table.  Run on the GPU box:  bash tools/profile_round.sh
R=/root/repo
O=$R/gpurun_out/round
rm -rf $O && mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for g in 256 512; do
  timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_f$g -o f -- python3 $R/tools/spmv_only.py $g 5 1 > /dev/null 2>&1 || exit 1
  timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_w$g -o w -- python3 $R/tools/spmv_only.py $g 5 1 > /dev/null 2>&1 || exit 1
  python3 $R/tools/pmc_traffic.py $O/pmc_f$g $O/pmc_w$g $g $O/spmv${g}_traffic.json > /dev/null || exit 1
done
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench_prof -o bench -- python3 $R/bench.py > $O/bench_prof.log 2>&1 || exit 1
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench256_prof -o bench -- python3 $R/bench.py --grid 256 --no-cpu-baseline > $O/bench256_prof.log 2>&1 || exit 1
cd $R
timeout -k 10 600 python3 bench.py > $O/bench_n1.json 2> $O/bench_n1.err || exit 1
timeout -k 10 600 python3 bench.py --grid 256 --no-cpu-baseline > $O/bench_256.json 2> $O/bench_256.err || exit 1
timeout -k 10 900 python3 tools/bench_configs.py 256 64 > $O/configs_256.jsonl 2> $O/configs.err || exit 1
echo ROUND_OK
