#!/bin/bash
# quick check of the box-stencil kernels on the GPU box: parity tests, then the timings that matter
timeout -k 10 300 python -m pytest tests/test_gpu_0_parity.py tests/test_gpu_fullsize_oracle.py -m gpu -x -q -k "box or ilup or fill" 2>&1 | tail -2
timeout -k 10 200 python tools/box_shapes.py 4096x7x8,96x96x96,16x96x96 | cut -c1-170
python - <<PY
import sys; sys.path.insert(0, ".")
import kryst_amd as K
ctx = K.Context(0)
a = K.CsrMatrix.stencil7(128, "poisson", ctx=ctx)
pc = K.Ilup(1).setup(a)
n = a.nrows(); r = ctx.vec(n).fill_splitmix(3); z = ctx.vec(n)
print("Ilup(1) 128^3 apply ms", min(pc.bench_apply(r, z, 10) for _ in range(3)), pc.ilu_info()["form"])
PY
