#!/bin/bash
# Rehearsal of bench.py's N > 1 flow on a ONE-GPU box: P ranks share device 0 and tests/shim/librccl_shim.so stands in for RCCL
# (real RCCL refuses two ranks per GPU).  Checks the launch plumbing, the distributed CG path and the JSON line, not the speed.
# usage: bench_rehearsal.sh <P> <launcher: torch|socket> [grid]
P=${1:-2}; L=${2:-torch}; G=${3:-128}
cd /root/repo
export KRYST_RCCL_LIB=/root/repo/tests/shim/librccl_shim.so KRYST_BENCH_DEVICE=0 HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 500 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node $P --master-addr 127.0.0.1 --master-port 29533 \
    bench.py --gpus $P --steps 40 --warmup 5 --grid $G --launcher $L --phase-iters 10
