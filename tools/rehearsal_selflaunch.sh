#!/bin/bash
# Round 5: config 4 AS WRITTEN through bench.py's SELF-LAUNCH (`python3 bench.py --gpus N`, started plainly like the driver starts --gpus 1), rehearsed
# on a ONE-GPU box: Jacobi-PCG on the 512^3 Poisson system over EIGHT ranks (4 processes x 2 rank threads; tests/shim/librccl_shim.so stands in for
# RCCL, which refuses two ranks per GPU) and over FIVE processes (the form in which the peer-store halo exchange and the mailboxes are the library's
# defaults), GMRES(30) + Jacobi timed beside CG / PCG in every line.  The 1-GPU run of the same iterations beside it: residuals within 1e-12 relative.
# Checks plumbing, partition, defaults and bits -- not speed.   usage: tools/rehearsal_selflaunch.sh [grid=512] [steps=20]
G=${1:-512}; K=${2:-20}
cd /root/repo
O=gpurun_out/rehearsal5; mkdir -p $O
export KRYST_RCCL_LIB=/root/repo/tests/shim/librccl_shim.so KRYST_BENCH_DEVICE=0 HSA_ENABLE_IPC_MODE_LEGACY=0 GPU_MAX_HW_QUEUES=8 KRYST_BENCH_WATCHDOG_S=900
unset WORLD_SIZE RANK LOCAL_RANK MASTER_ADDR MASTER_PORT
[ -f $KRYST_RCCL_LIB ] && [ ! tests/shim/rccl_shim.cpp -nt $KRYST_RCCL_LIB ] || /opt/rocm/bin/hipcc -O2 -std=c++17 -fPIC -shared -x hip --offload-arch=gfx950 tests/shim/rccl_shim.cpp -o $KRYST_RCCL_LIB -I/opt/rocm/include -lrt || exit 1
for S in pcg cg; do
  timeout -k 10 1000 python3 bench.py --gpus 8 --ranks-per-process 2 --solver $S --steps $K --warmup 5 --grid $G --phase-iters 10 --gmres-steps 30 \
      > $O/selflaunch_n8_${G}_$S.json 2> $O/selflaunch_n8_${G}_$S.err || { tail -30 $O/selflaunch_n8_${G}_$S.err; exit 1; }
done
timeout -k 10 1000 python3 bench.py --gpus 5 --solver pcg --steps $K --warmup 5 --grid $G --phase-iters 10 --gmres-steps 30 \
    > $O/selflaunch_n5_${G}_pcg.json 2> $O/selflaunch_n5_${G}_pcg.err || { tail -30 $O/selflaunch_n5_${G}_pcg.err; exit 1; }
unset KRYST_RCCL_LIB KRYST_BENCH_DEVICE
for S in pcg cg; do
  timeout -k 10 600 python3 bench.py --gpus 1 --solver $S --steps $K --warmup 5 --grid $G --no-256 --no-configs --no-cpu-baseline --phase-iters 10 --gmres-steps 30 \
      > $O/selflaunch_n1_${G}_$S.json 2> $O/selflaunch_n1_${G}_$S.err || { tail -30 $O/selflaunch_n1_${G}_$S.err; exit 1; }
done
python3 - $O $G <<'PY'
import json, sys
o, g = sys.argv[1], sys.argv[2]
def load(path):
    lines = [ln for ln in open(path) if ln.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), (path, len(lines))      # the JSON line is the ONLY thing on stdout
    return json.loads(lines[0])
one = {s: load(f"{o}/selflaunch_n1_{g}_{s}.json") for s in ("pcg", "cg")}
for ranks, s in ((8, "pcg"), (8, "cg"), (5, "pcg")):
    d = load(f"{o}/selflaunch_n{ranks}_{g}_{s}.json")
    r8, r1 = d["config"]["final_residual"], one[s]["config"]["final_residual"]
    g8, g1 = d["gmres30_jacobi"], one[s]["gmres30_jacobi"]
    print(json.dumps({"ranks": ranks, "solver": s, "launcher": d["config"]["launcher"], "final_residual": r8, "final_residual_1_gpu": r1,
                      "relative_difference": abs(r8 - r1) / r1, "within_1e-12": abs(r8 - r1) / r1 <= 1e-12,
                      "gmres30_jacobi": {"value_on_one_shared_gpu": g8.get("value"), "final_residual": g8.get("final_residual"), "final_residual_1_gpu": g1.get("final_residual"),
                                         "relative_difference": abs(g8["final_residual"] - g1["final_residual"]) / g1["final_residual"] if "final_residual" in g8 else None},
                      "scalar_reduce": d["scalar_reduce"], "value_on_one_shared_gpu": d["value"], "phase_ms": d["phase_ms"]}))
PY
echo REHEARSAL_OK
