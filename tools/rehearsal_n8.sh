#!/bin/bash
# Config 4 AS WRITTEN, rehearsed on a ONE-GPU box: Jacobi-PCG on the 512^3 Poisson system row-partitioned over EIGHT ranks (eight k-slabs of
# 64 planes, 16.8 M rows per rank; the whole problem fits one 288 GB GPU), through bench.py's own N > 1 flow.  The eight ranks are 4
# processes x 2 rank threads (a GPU box admits at most 6 processes on its card) and tests/shim/librccl_shim.so stands in for RCCL (real RCCL
# refuses two ranks per GPU).  Both launchers; bench.py itself times both scalar-reduce paths and every halo form and reports whether their
# residuals are bit-identical.  The 1-GPU run of the same iterations is printed beside it: the residuals must agree to 1e-12 relative.
# Checks plumbing, partition and bits -- not speed.   usage: tools/rehearsal_n8.sh [grid=512] [steps=20]
G=${1:-512}; K=${2:-20}
cd /root/repo
O=gpurun_out/rehearsal; mkdir -p $O
export KRYST_RCCL_LIB=/root/repo/tests/shim/librccl_shim.so KRYST_BENCH_DEVICE=0 HSA_ENABLE_IPC_MODE_LEGACY=0 GPU_MAX_HW_QUEUES=8 KRYST_BENCH_WATCHDOG_S=900
[ -f $KRYST_RCCL_LIB ] && [ ! tests/shim/rccl_shim.cpp -nt $KRYST_RCCL_LIB ] || /opt/rocm/bin/hipcc -O2 -std=c++17 -fPIC -shared -x hip --offload-arch=gfx950 tests/shim/rccl_shim.cpp -o $KRYST_RCCL_LIB -I/opt/rocm/include -lrt || exit 1
for L in torch socket; do
  timeout -k 10 1000 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port $((29600 + RANDOM % 300)) \
      bench.py --gpus 8 --solver pcg --steps $K --warmup 5 --grid $G --launcher $L --phase-iters 10 --ranks-per-process 2 \
      > $O/rehearsal_n8_${G}_$L.json 2> $O/rehearsal_n8_${G}_$L.err || { tail -30 $O/rehearsal_n8_${G}_$L.err; exit 1; }
done
# five PROCESSES (with the launcher's own process six have the GPU open: the most a GPU box admits), one rank each: the form in which the
# peer-store halo exchange is available (rank threads that share a device are refused, kryst_amd/csrc/dist.cpp: ipc_map_peers); k-slabs of
# 102 / 103 planes
timeout -k 10 1000 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 5 --master-addr 127.0.0.1 --master-port $((29600 + RANDOM % 300)) \
    bench.py --gpus 5 --solver pcg --steps $K --warmup 5 --grid $G --launcher torch --phase-iters 10 \
    > $O/rehearsal_n5_${G}_torch.json 2> $O/rehearsal_n5_${G}_torch.err || { tail -30 $O/rehearsal_n5_${G}_torch.err; exit 1; }
unset KRYST_RCCL_LIB KRYST_BENCH_DEVICE
timeout -k 10 600 python3 bench.py --gpus 1 --solver pcg --steps $K --warmup 5 --grid $G --no-256 --no-configs --no-cpu-baseline --phase-iters 10 \
    > $O/rehearsal_n1_${G}.json 2> $O/rehearsal_n1_${G}.err || { tail -30 $O/rehearsal_n1_${G}.err; exit 1; }
python3 - $O $G <<'PY'
import json, sys
o, g = sys.argv[1], sys.argv[2]
def load(path):                      # (gloo prints its connection messages on stdout too: the bench line is the one that starts with a brace)
    return json.loads([ln for ln in open(path) if ln.startswith("{")][-1])
one = load(f"{o}/rehearsal_n1_{g}.json")
r1 = one["config"]["final_residual"]
for ranks, l in ((8, "torch"), (8, "socket"), (5, "torch")):
    d = load(f"{o}/rehearsal_n{ranks}_{g}_{l}.json")
    r8 = d["config"]["final_residual"]
    print(json.dumps({"ranks": ranks, "launcher": d["config"]["launcher"], "final_residual": r8, "final_residual_1_gpu": r1, "relative_difference": abs(r8 - r1) / r1,
                      "within_1e-12": abs(r8 - r1) / r1 <= 1e-12, "scalar_reduce": d["scalar_reduce"], "value_on_one_shared_gpu": d["value"],
                      "phase_ms": d["phase_ms"]}))
PY
echo REHEARSAL_OK
