cd /root/repo
export KRYST_RCCL_LIB=/root/repo/tests/shim/librccl_shim.so KRYST_BENCH_DEVICE=0 HSA_ENABLE_IPC_MODE_LEGACY=0 GPU_MAX_HW_QUEUES=8 KRYST_BENCH_WATCHDOG_S=600
unset WORLD_SIZE RANK LOCAL_RANK MASTER_ADDR MASTER_PORT
for N in 2 4; do
  timeout -k 10 500 python3 bench.py --gpus $N --solver cg --steps 20 --warmup 5 --grid 512 --phase-iters 10 --gmres-steps 30 > gpurun_out/reh_n${N}_512.json 2> gpurun_out/reh_n${N}_512.err || { tail -20 gpurun_out/reh_n${N}_512.err; exit 1; }
  python3 - $N <<'PY'
import json, sys
n = sys.argv[1]
d = json.loads(open(f"/root/repo/gpurun_out/reh_n{n}_512.json").read().strip().splitlines()[-1])
print(n, "ranks:", d["value"], d["config"]["final_residual"], d["scalar_reduce"].get("chosen") if isinstance(d.get("scalar_reduce"), dict) else None, json.dumps(d["phase_ms"][0])[:200])
PY
done
