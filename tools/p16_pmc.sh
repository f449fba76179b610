cd /tmp && export TMPDIR=/tmp
R=/root/repo
for cfg in "0 8 8" "0 1 1" "0 8 1" "1 1 8"; do
  set -- $cfg
  export KRYST_SPMV_SWIZZLE=$1 KRYST_SPMV_GROUP=$2 KRYST_SPMV_PATTERN_TPW=$3
  rm -rf $R/gpurun_out/pq
  timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pq -o f -- python3 $R/tools/spmv_only.py 512 4 1 > /dev/null 2>&1 || exit 1
  python3 - "$cfg" <<'PY'
import csv, glob, sys
f = glob.glob('/root/repo/gpurun_out/pq/**/*_counter_collection.csv', recursive=True)[0]
v = [float(r['Counter_Value']) for r in csv.DictReader(open(f)) if 'spmv_pattern' in r['Kernel_Name'] and r['Counter_Name'] == 'FETCH_SIZE']
print("swizzle/group/tpw", sys.argv[1], "read GB per launch %.2f" % (sum(v) / len(v) * 1024 * 2 / 1e9))
PY
done
