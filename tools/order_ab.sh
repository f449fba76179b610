mkdir -p gpurun_out/r3j
o=gpurun_out/r3j/ab_order_forms.txt; : > $o
for g in 512 256; do
  python tools/spmv_ab.py $g poisson 7 20 -- "" "KRYST_SPMV_ORDER=0" >> $o 2>&1
  python tools/spmv_ab.py $g varcoef 7 20 -- "" "KRYST_SPMV_ORDER=0" >> $o 2>&1
  KRYST_SPMV_COMPRESS=0 python tools/spmv_ab.py $g poisson 7 20 -- "" "KRYST_SPMV_ORDER=0" >> $o 2>&1
  KRYST_SPMV_COMPRESS=1 KRYST_SPMV_DIA=0 python tools/spmv_ab.py $g varcoef 7 20 -- "" "KRYST_SPMV_ORDER=0" >> $o 2>&1
  KRYST_SPMV_COMPRESS=2 python tools/spmv_ab.py $g poisson 7 20 -- "" "KRYST_SPMV_ORDER=0" >> $o 2>&1
done
cat $o
