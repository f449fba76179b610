cd /tmp && export TMPDIR=/tmp
R=/root/repo
export KRYST_SPMV_GROUP=8
i=0
for set in "SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "GRBM_GUI_ACTIVE SQ_WAVES SQ_WAIT_INST_LDS SQ_LDS_ADDR_CONFLICT"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/sq$i -o s -- python3 $R/tools/spmv_only.py 512 3 1 > /dev/null 2>&1 || echo "set $i failed"
done
python3 - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob('/root/repo/gpurun_out/sq*/')):
    for f in glob.glob(d + '**/*_counter_collection.csv', recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if 'spmv_dict' in r['Kernel_Name']:
                acc[r['Counter_Name']].append(float(r['Counter_Value']))
        for k, v in acc.items():
            print(k, sum(v) / len(v))
PY
