#!/bin/bash
# one multi-rank case of tests/test_gpu_z_multirank_shim.py by hand: mr_case.sh <P> <N> <kind> <ranks per process> [light=1]
P=$1; N=$2; KIND=$3; PER=${4:-1}; LIGHT=${5:-1}
cd /root/repo
D=$(mktemp -d /tmp/mr_XXXX)
export KRYST_RCCL_LIB=/root/repo/tests/shim/librccl_shim.so KRYST_STENCIL_HOST=0 KRYST_MR_LIGHT=$LIGHT GPU_MAX_HW_QUEUES=8 HSA_ENABLE_IPC_MODE_LEGACY=0
pids=()
for ((r = 0; r < P; r += PER)); do
  ranks=$r; for ((t = 1; t < PER; ++t)); do ranks="$ranks,$((r + t))"; done
  python3 tests/multirank_worker.py $ranks $P $D $N $KIND > $D/log_$r.txt 2>&1 &
  pids+=($!)
done
rc=0
for p in "${pids[@]}"; do wait $p || rc=1; done
grep -h "MISMATCH\|Error\|RANK_OK" $D/log_*.txt | sort | uniq -c | head -20
echo "case $P $N $KIND per=$PER rc=$rc"
rm -rf $D
exit 0
