#!/usr/bin/env python3
"""Step-level timeline of one block hop of the forward 16 x 16 wavefront solve: for a producer block A and its east neighbour B
(trace build, see tq_trace.py) prints, per step, when A's quadrants finished it, when A's exporter stored it, when B's poller
delivered it and when B's quadrants finished theirs.  usage: tq_steps.py N JA,KA JB,KB [steps=40]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import kryst_amd as K
from kryst_amd import _ffi
N = int(sys.argv[1]); nbj = (N + 15) // 16
A = [int(v) for v in sys.argv[2].split(",")]; B = [int(v) for v in sys.argv[3].split(",")]
nsteps = int(sys.argv[4]) if len(sys.argv) > 4 else 40
ctx = K.Context(0)
a = K.CsrMatrix.stencil7(N, "aniso", ctx=ctx)
pc = K.TrueIlu0().setup(a)
n = a.nrows()
r = ctx.vec(n).fill_splitmix(3); z = ctx.vec(n)
lib = _ffi.lib()
lib.kryst_debug_tq_select.argtypes = [C.c_int32, C.c_int32]
lib.kryst_debug_tq_steps.argtypes = [C.POINTER(C.c_longlong)]
assert lib.kryst_debug_tq_select(A[1] * nbj + A[0], B[1] * nbj + B[0]) == 0
for _ in range(3):
    lib.kryst_pc_apply(pc.h, r.h, z.h); ctx.synchronize()
buf = (C.c_longlong * 3072)()
assert lib.kryst_debug_tq_steps(buf) == 0
t = np.array(buf, dtype=np.int64).reshape(3, 2, 4, 128).astype(np.float64) / 100.0
steps, deliv, exp = t[0], t[1], t[2]
t0 = steps[0, 0, 0]
print(f"N {N}: producer block {A}, consumer block {B}; us after the producer's q0 step 0")
print("step | A.q0   A.q1   A.q2   A.q3 | A.exp(e of q1) A.exp(e of q3) | B.deliv(w of q0) B.deliv(w of q2) | B.q0   B.q1   B.q2   B.q3")
for s in range(nsteps):
    f = lambda x: f"{x - t0:6.2f}" if x > 0 else "   -  "
    print(f"{s:4d} | " + " ".join(f(steps[0, q, s]) for q in range(4)) + " | " + f(exp[0, 0, s]) + "        " + f(exp[0, 1, s]) + "        | " +
          f(deliv[1, 0, s]) + "           " + f(deliv[1, 1, s]) + "           | " + " ".join(f(steps[1, q, s]) for q in range(4)))
