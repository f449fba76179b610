"""CGS and TFQMR: the HIP CgsSolver / TfqmrSolver against the oracle's restatements of
src/solver/cgs.rs:58-135 and src/solver/tfqmr.rs:64-221, bit for bit in the library's reduction order."""
import numpy as np
import pytest

import kryst_amd as K
from oracle import oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    return K.Context(0)


@pytest.fixture(scope="module")
def rs():
    T, V, F = K.reduce_spec()
    return O.Reduce.tiled(T, V, F)


def to_dev(ctx, a):
    return K.CsrMatrix.from_csr(a.nrows, a.ncols, a.row_ptr, a.col_idx, a.vals, ctx=ctx)


def check(res, st, s, x, nan_ok=False):
    assert st.iterations == res.iterations and st.converged == res.converged
    assert st.final_residual == res.final_residual or (nan_ok and np.isnan(st.final_residual) and np.isnan(res.final_residual))
    assert np.array_equal(s.residual_history, res.history, equal_nan=nan_ok)
    assert np.array_equal(x, res.x, equal_nan=nan_ok)


def test_cgs_reference_known_answer(ctx):
    # src/solver/cgs.rs:155-188
    ao = O.Csr.from_dense([[10.0, 2, 0, 0, 0], [3, 15, 4, 0, 0], [0, -2, 8, 1, 0], [0, 0, 1, 7, 3], [0, 0, 0, 2, 12]])
    xt = np.arange(1.0, 6.0); b = ao.spmv(xt)
    x = np.zeros(5)
    st = K.CgsSolver(1e-10, 200).solve(to_dev(ctx, ao), None, b, x)
    assert st.converged and np.all(np.abs(x - xt) <= 1e-6)


@pytest.mark.parametrize("kind,N", [("convdiff", 8), ("poisson", 12), ("aniso", 10)])
@pytest.mark.parametrize("method", ["cgs", "tfqmr"])
def test_bit_exact(ctx, rs, method, kind, N):
    a = O.stencil7(N, kind)
    b = a.spmv(np.linspace(0.5, 1.5, a.nrows))
    d = to_dev(ctx, a)
    cls = K.CgsSolver if method == "cgs" else K.TfqmrSolver
    for tol, mx in ((1e-8, 300), (1e-30, 7), (1e-2, 300)):
        x0 = np.linspace(-1.0, 1.0, a.nrows)
        res = O.solve(method, a, b, x0=x0, tol=tol, max_iters=mx, rs=rs)
        s = cls(tol, mx); x = x0.copy()
        st = s.solve(d, K.Jacobi().setup(d), b, x)          # pc is ignored by both (cgs.rs:59, tfqmr.rs:66)
        check(res, st, s, x)
    ser = O.solve(method, a, b, tol=1e-8, max_iters=300)
    s = cls(1e-8, 300); x = np.zeros(a.nrows)
    st = s.solve(d, None, b, x)
    if method == "cgs":                                      # vs the strict serial fold: same count, close history
        assert st.iterations == ser.iterations
        assert np.max(np.abs(np.array(ser.history) - np.array(s.residual_history))) <= 1e-9 * np.linalg.norm(b)
        assert np.linalg.norm(b - a.spmv(x)) <= 1e-7 * np.linalg.norm(b)


def test_edges(ctx, rs):
    a = O.stencil7(6, "convdiff"); d = to_dev(ctx, a)
    b = a.spmv(np.ones(a.nrows))
    for method, cls in (("cgs", K.CgsSolver), ("tfqmr", K.TfqmrSolver)):
        # zero right-hand side with a zero / nonzero guess; max_iters = 0
        for x0 in (np.zeros(a.nrows), np.ones(a.nrows)):
            res = O.solve(method, a, np.zeros(a.nrows), x0=x0, tol=1e-8, max_iters=20, rs=rs)
            s = cls(1e-8, 20); x = x0.copy()
            check(res, s.solve(d, None, np.zeros(a.nrows), x), s, x, nan_ok=True)
        res = O.solve(method, a, b, tol=1e-8, max_iters=0, rs=rs)
        s = cls(1e-8, 0); x = np.zeros(a.nrows)
        check(res, s.solve(d, None, b, x), s, x)
    # CGS exact solve on a diagonal system: rho underflows to the breakdown test (cgs.rs:80-82) on both sides alike
    a2 = O.Csr.from_dense(np.diag([2.0, 2.0, 2.0])); b2 = np.array([1.0, 2.0, 3.0])
    for method, cls in (("cgs", K.CgsSolver), ("tfqmr", K.TfqmrSolver)):
        res = O.solve(method, a2, b2, tol=1e-12, max_iters=10, rs=rs)
        s = cls(1e-12, 10); x = np.zeros(3)
        check(res, s.solve(to_dev(ctx, a2), None, b2, x), s, x, nan_ok=True)
    # TFQMR sigma == 0 exit (tfqmr.rs:116-121): A y orthogonal to r_tld at the first step
    a3 = O.Csr.from_dense([[0.0, 1.0], [-1.0, 0.0]]); b3 = np.array([1.0, 1.0])
    res = O.solve("tfqmr", a3, b3, tol=1e-12, max_iters=10, rs=rs)
    assert res.iterations == 1 and not res.converged
    s = K.TfqmrSolver(1e-12, 10); x = np.zeros(2)
    check(res, s.solve(to_dev(ctx, a3), None, b3, x), s, x)


def test_device_vectors_session_and_ksp(ctx, rs):
    a = O.stencil7(8, "convdiff"); d = to_dev(ctx, a)
    b = a.spmv(np.ones(a.nrows))
    for method, cls, kind in (("cgs", K.CgsSolver, K.SolverKind.Cgs), ("tfqmr", K.TfqmrSolver, K.SolverKind.Tfqmr)):
        res = O.solve(method, a, b, tol=1e-9, max_iters=40, rs=rs)
        seen = []
        s = cls(1e-9, 40).with_monitor(lambda i, r: seen.append(r))
        bv, xv = K.DeviceVec(ctx, b), K.DeviceVec(ctx, np.zeros(a.nrows))
        check(res, s.solve(d, None, bv, xv), s, xv.to_host())
        assert seen == list(res.history)
        x = np.zeros(a.nrows)
        st = K.KspContext(kind, d, tol=1e-9, max_it=40).solve_context(b, x)
        assert st.iterations == res.iterations and np.array_equal(x, res.x)
        # stepping session: exactly 9 iterations in two batches
        ref = O.solve(method, a, b, tol=0.0, max_iters=9, rs=rs)
        xs = K.DeviceVec(ctx, np.zeros(a.nrows))
        sess = K.Session(method, d, None, K.DeviceVec(ctx, b), xs, tol=0.0, max_iters=9)
        sess.step(4); sess.step(5)
        st = sess.end()
        assert st.iterations == 9 and np.array_equal(xs.to_host(), ref.x)
