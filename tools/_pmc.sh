R=/root/repo; O=$R/gpurun_out/pmcq; rm -rf $O; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for st in 1 0; do
export KRYST_SPMV_STAGE=$st
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/f$st -o f -- python3 $R/tools/spmv_only.py 512 5 1 > /dev/null 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/w$st -o w -- python3 $R/tools/spmv_only.py 512 5 1 > /dev/null 2>&1 || exit 1
python3 $R/tools/pmc_traffic.py $O/f$st $O/w$st 512 $O/t$st.json > /dev/null || exit 1
python3 -c "import json; d=json.load(open('$O/t$st.json')); print('stage $st', d['kernel'][:50], d['read_bytes_per_launch']/1e9, d['write_bytes_per_launch']/1e9)"
done
timeout -k 10 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/h1 -o h -- python3 $R/tools/spmv_only.py 512 5 1 > /dev/null 2>&1
python3 - <<PY
import csv,glob,collections
f=glob.glob('$O/h1/**/*_counter_collection.csv',recursive=True)[0]
acc=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if 'spmv' in r['Kernel_Name']: acc[r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in acc.items(): print(k, sum(v)/len(v))
PY
