#!/bin/bash
# the 8-rank, 4 x 2 rank-thread case of tests/test_gpu_z_multirank_shim.py, N times: does the mailbox set-up ever fall back, and why?
cd /root/repo
for i in $(seq 1 ${1:-12}); do
  D=$(mktemp -d /tmp/mr_XXXX)
  export KRYST_RCCL_LIB=/root/repo/tests/shim/librccl_shim.so KRYST_STENCIL_HOST=0 KRYST_MR_LIGHT=1 GPU_MAX_HW_QUEUES=8 HSA_ENABLE_IPC_MODE_LEGACY=0 KRYST_IPC_DEBUG=1
  pids=()
  for r in 0 2 4 6; do python3 tests/multirank_worker.py $r,$((r+1)) 8 $D 4000 random > $D/log_$r.txt 2>&1 & pids+=($!); done
  for p in "${pids[@]}"; do wait $p; done
  echo "run $i: $(cat $D/log_*.txt | grep -c RANK_OK) ranks ok; $(grep -h "fell back\|hipIpc\|could not export" $D/log_*.txt | sort | uniq -c | head -6 | tr '\n' ';')"
  rm -rf $D
done
