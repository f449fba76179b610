"""bench.py's N > 1 flow rehearsed on the one GPU a test box has: two ranks started by torch.distributed.run exactly as the driver
starts them, both on device 0 (KRYST_BENCH_DEVICE) with the shared-memory stand-in for RCCL (real RCCL refuses two ranks per GPU).
Checks the launch plumbing, the row-partitioned CG session, both scalar all-reduce paths and the shape of the JSON line -- not speed."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIM = os.path.join(ROOT, "tests", "shim", "librccl_shim.so")


@pytest.mark.gpu
@pytest.mark.parametrize("launcher", ["torch", "socket"])
def test_bench_with_two_ranks_on_one_gpu(launcher):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_gpu_z_multirank_shim import _build_shim
    _build_shim()
    env = dict(os.environ, KRYST_RCCL_LIB=SHIM, KRYST_BENCH_DEVICE="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    port = "29541" if launcher == "torch" else "29551"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", port,
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "3", "--grid", "64", "--phase-iters", "5", "--launcher", launcher]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 20 and d["value"] > 0 and d["scaling"] == "strong" and d["unit"] == "iterations/s"
    assert d["config"]["partition"] == "2 k-slab(s)" and d["config"]["grid"] == 64
    assert len(d["phase_ms"]) == 2 and all("spmv" in p and "reduce" in p for p in d["phase_ms"])
    sr = d["scalar_reduce"]
    assert sr["path"] in ("ipc", "rccl") and sr["value_rccl"] > 0 and sr["value_ipc"] > 0 and sr["ipc"] == "bit-identical residual"
    assert d["roofline"]["frac"] <= 1.0 and d["roofline_csr"]["frac"] <= 1.0 and "cpu_baseline" not in d
