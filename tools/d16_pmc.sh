cd /tmp && export TMPDIR=/tmp
R=/root/repo
for grp in 1 8; do
  export KRYST_SPMV_GROUP=$grp
  timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_f$grp -o f -- python3 $R/tools/spmv_only.py 512 5 1 > /dev/null 2>&1 || exit 1
  timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_w$grp -o w -- python3 $R/tools/spmv_only.py 512 5 1 > /dev/null 2>&1 || exit 1
  python3 $R/tools/pmc_traffic.py $R/gpurun_out/pmc_f$grp $R/gpurun_out/pmc_w$grp 512 $R/gpurun_out/traffic512_g$grp.json | grep "read_bytes_per_launch\|write_bytes_per_launch\|factor"
done
