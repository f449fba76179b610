#!/bin/bash
# ILU evidence for profiles/rNN: kernel stats of the apply at 256^3 / 512^3, PMC passes at 512^3, the config table.
R=/root/repo; O=$R/gpurun_out/ilu_round
rm -rf $O && mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for g in 256 512; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$g -o ilu -- python3 $R/tools/ilu_only.py $g 20 true > $O/ilu_$g.log 2>&1 || exit 1
done
bash $R/tools/ilu_pmc.sh 512 > $O/pmc512.log 2>&1 || exit 1
cp $R/gpurun_out/ilu_pmc_512/summary.txt $O/ilu_apply512_pmc_summary.txt
cd $R
timeout -k 10 600 python3 tools/bench_configs.py 256 64 > $O/configs_256.jsonl 2> $O/configs.err || exit 1
echo ILU_ROUND_OK
