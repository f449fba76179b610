#!/bin/bash
# Evidence for profiles/rNN: PMC traffic of the SpMV kernel in both storage forms (separate --pmc passes), kernel-trace stats of
# the default bench.py command (plus per-size averages: the default command measures 512^3 and 256^3), the bench line itself
# and the per-config table.  Run on the GPU box:  bash tools/profile_round.sh
R=/root/repo
O=$R/gpurun_out/round
rm -rf $O && mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for g in 256 512; do
  for form in default plain; do
    if [ $form = plain ]; then export KRYST_SPMV_COMPRESS=0; else unset KRYST_SPMV_COMPRESS; fi
    timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_f${g}_$form -o f -- python3 $R/tools/spmv_only.py $g 5 1 > /dev/null 2>&1 || exit 1
    timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_w${g}_$form -o w -- python3 $R/tools/spmv_only.py $g 5 1 > /dev/null 2>&1 || exit 1
    python3 $R/tools/pmc_traffic.py $O/pmc_f${g}_$form $O/pmc_w${g}_$form $g $O/spmv${g}_${form}_traffic.json $O/spmv_traffic.json $form > /dev/null || exit 1
  done
done
unset KRYST_SPMV_COMPRESS
# the variable-coefficient operator in its default form (CSR-DIA)
for g in 256 512; do
  timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_f${g}_varcoef -o f -- python3 $R/tools/spmv_only.py $g 5 1 varcoef > /dev/null 2>&1 || exit 1
  timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_w${g}_varcoef -o w -- python3 $R/tools/spmv_only.py $g 5 1 varcoef > /dev/null 2>&1 || exit 1
  python3 $R/tools/pmc_traffic.py $O/pmc_f${g}_varcoef $O/pmc_w${g}_varcoef $g $O/spmv${g}_varcoef_traffic.json $O/spmv_traffic.json varcoef > /dev/null || exit 1
done
cp $O/spmv_traffic.json $R/profiles/spmv_traffic.json   # (on the GPU box only: copy gpurun_out/round/spmv_traffic.json into profiles/ after the call)
KRYST_BENCH_LIVE_TRAFFIC=0 timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench_prof -o bench -- python3 $R/bench.py > $O/bench_prof.log 2>&1 || exit 1
python3 $R/tools/kernel_by_size.py $O/bench_prof $O/bench_default_kernel_by_size.csv || exit 1
for g in 256 512; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ilu_prof_$g -o ilu -- python3 $R/tools/ilu_only.py $g 20 true > $O/ilu_$g.log 2>&1 || exit 1
done
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/box_prof_96 -o box -- python3 $R/tools/box_only.py 96 20 > $O/box_96.log 2>&1 || exit 1
cd $R
timeout -k 10 900 python3 bench.py > $O/bench_n1.json 2> $O/bench_n1.err || exit 1
timeout -k 10 900 python3 tools/bench_configs.py 256 64 > $O/configs_256.jsonl 2> $O/configs.err || exit 1
timeout -k 10 600 python3 tools/ilu_general.py 96 2000000 > $O/ilu_general.jsonl 2> $O/ilu_general.err || exit 1
echo ROUND_OK
