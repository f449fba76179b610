cd /tmp && export TMPDIR=/tmp
R=/root/repo
G=${1:-512}
i=0
rm -rf $R/gpurun_out/sq*
for set in "VALUBusy SALUBusy" "LdsUtil MemUnitStalled" "TA_BUSY_avr GRBM_GUI_ACTIVE" "TCP_TOTAL_CACHE_ACCESSES TCP_TCC_READ_REQ TCP_PENDING_STALL_CYCLES" "TCP_UTCL1_TRANSLATION_MISS TCP_UTCL1_REQUEST TCP_GATE_EN1" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_BUSY_CYCLES"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/sq$i -o s -- python3 $R/tools/spmv_only.py $G 3 1 > /dev/null 2>&1 || echo "set $i failed: $set"
done
python3 - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob('/root/repo/gpurun_out/sq*/')):
    for f in glob.glob(d + '**/*_counter_collection.csv', recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if 'spmv_' in r['Kernel_Name']:
                acc[r['Counter_Name']].append(float(r['Counter_Value']))
        for k, v in acc.items():
            print(k, sum(v) / len(v))
PY
