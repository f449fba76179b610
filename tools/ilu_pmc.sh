#!/bin/bash
# PMC passes (separate runs) over the ILU apply alone.  usage: bash tools/ilu_pmc.sh [grid=512]
R=/root/repo; G=${1:-512}; O=$R/gpurun_out/ilu_pmc_$G
rm -rf $O && mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum; do
  timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d $O/$c -o p -- python3 $R/tools/ilu_only.py $G 5 true > $O/$c.log 2>&1 || { echo "pass $c failed"; tail -3 $O/$c.log; }
done
python3 $R/tools/pmc_kernels.py $O/FETCH_SIZE $O/WRITE_SIZE $O/TCC_HIT_sum $O/TCC_MISS_sum $O/TCC_EA0_RDREQ_sum $O/TCC_EA0_WRREQ_sum --match tri_ > $O/summary.txt
cat $O/summary.txt
