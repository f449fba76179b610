"""FGMRES: the HIP FgmresSolver::solve_flex against the oracle's restatement of
src/solver/fgmres.rs:114-340, bit for bit in the library's reduction order (iterations, stats, history, x)."""
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
    assert st.final_residual == res.final_residual
    assert np.array_equal(s.residual_history, res.history, equal_nan=nan_ok)
    assert np.array_equal(x, res.x, equal_nan=nan_ok)


def test_reference_known_answer(ctx):
    # src/solver/fgmres.rs:532-552 through the mirrored API
    a = K.CsrMatrix.from_csr(2, 2, [0, 2, 4], [0, 1, 0, 1], [2.0, 1.0, 1.0, 3.0], ctx=ctx)
    xt = np.array([1.0, 2.0]); b = np.array([4.0, 7.0])
    x = np.zeros(2)
    st = K.FgmresSolver(1e-10, 100, 25).solve_flex(a, K.Jacobi().setup(a), b, x)
    assert st.converged and np.all(np.abs(x - xt) < 1e-6)
    x = np.zeros(2)
    st = K.FgmresSolver(1e-10, 100, 25).solve_flex(a, None, b, x)
    assert st.converged and np.all(np.abs(x - xt) < 1e-6)


@pytest.mark.parametrize("orthog", [K.Orthog.Classical, K.Orthog.Modified])
@pytest.mark.parametrize("pcname", ["none", "jacobi", "ilu0"])
@pytest.mark.parametrize("restart", [3, 9, 30])
def test_fgmres_bit_exact(ctx, rs, orthog, pcname, restart):
    a = O.stencil7(8, "convdiff")
    b = a.spmv(np.ones(a.nrows))
    d = to_dev(ctx, a)
    opc = {"none": lambda: None, "jacobi": lambda: O.Pc.jacobi(a), "ilu0": lambda: O.Pc.ilu0_compat(a)}[pcname]()
    kpc = {"none": lambda: None, "jacobi": lambda: K.Jacobi().setup(d), "ilu0": lambda: K.Ilu0().setup(d)}[pcname]()
    res = O.solve("fgmres", a, b, pc=opc, tol=1e-8, max_iters=70, restart=restart, rs=rs, orthog=int(orthog))
    s = K.FgmresSolver(1e-8, 70, restart).with_orthog(orthog)
    x = np.zeros(a.nrows)
    st = s.solve_flex(d, kpc, b, x)
    check(res, st, s, x)
    assert st.final_residual == np.sqrt(O.dot(b, b, rs))          # the stats quirk: the INITIAL residual norm


def test_fgmres_large_batches_and_serial_agreement(ctx, rs):
    # restart 40: dot/axpy batches of 8, 4, 2 and 1 all occur; vs the strict serial fold: equal counts, close history
    a = O.stencil7(12, "convdiff")
    b = a.spmv(np.linspace(0.5, 1.5, a.nrows))
    d = to_dev(ctx, a)
    res = O.solve("fgmres", a, b, pc=O.Pc.jacobi(a), tol=1e-10, max_iters=120, restart=40, rs=rs)
    s = K.FgmresSolver(1e-10, 120, 40); x = np.zeros(a.nrows)
    st = s.solve_flex(d, K.Jacobi().setup(d), b, x)
    check(res, st, s, x)
    ser = O.solve("fgmres", a, b, pc=O.Pc.jacobi(a), tol=1e-10, max_iters=120, restart=40)
    assert ser.iterations == st.iterations
    assert np.max(np.abs(np.array(ser.history) - np.array(s.residual_history))) <= 1e-12 * np.linalg.norm(b)
    assert np.linalg.norm(b - a.spmv(x)) <= 1e-9 * np.linalg.norm(b)


def test_fgmres_edges(ctx, rs):
    a = O.stencil7(6, "poisson"); d = to_dev(ctx, a)
    b = a.spmv(np.ones(a.nrows))
    # zero right-hand side: beta == 0 early return (:141-143)
    s = K.FgmresSolver(1e-8, 50, 10); x = np.zeros(a.nrows)
    st = s.solve_flex(d, None, np.zeros(a.nrows), x)
    assert st.iterations == 0 and st.converged and st.final_residual == 0.0 and s.residual_history == [] and not x.any()
    # max_iters == 0: the loop is never entered
    res = O.solve("fgmres", a, b, tol=1e-8, max_iters=0, restart=10, rs=rs)
    s = K.FgmresSolver(1e-8, 0, 10); x = np.zeros(a.nrows)
    check(res, s.solve_flex(d, None, b, x), s, x)
    # iteration cap inside a cycle (Convergence::check reports stop at the cap), preallocate on/off, nonzero x0
    for pre in (False, True):
        for mx in (4, 10, 13):
            x0 = np.linspace(-1, 1, a.nrows)
            res = O.solve("fgmres", a, b, x0=x0, tol=1e-30, max_iters=mx, restart=5, rs=rs, preallocate=pre)
            s = K.FgmresSolver(1e-30, mx, 5).with_preallocate(pre); x = x0.copy()
            check(res, s.solve_flex(d, None, b, x), s, x)
    # happy breakdown: identity operator -> w - h v == 0 at the first step
    ai = O.Csr.from_dense(np.eye(5)); di = to_dev(ctx, ai); bi = np.arange(1.0, 6.0)
    res = O.solve("fgmres", ai, bi, tol=1e-10, max_iters=12, restart=4, rs=rs)
    s = K.FgmresSolver(1e-10, 12, 4); x = np.zeros(5)
    check(res, s.solve_flex(di, None, bi, x), s, x)
    assert np.array_equal(x, bi)
    # a loose happy tolerance (with_haptol) zeroes v_{j+1} while the iteration goes on (:259-261): the reference then
    # divides 0/0 in the back-substitution (:313, no pivot guard) and returns NaNs -- and so do we, at the same step
    res = O.solve("fgmres", a, b, tol=1e-12, max_iters=40, restart=8, rs=rs, haptol=0.5)
    s = K.FgmresSolver(1e-12, 40, 8).with_haptol(0.5); x = np.zeros(a.nrows)
    check(res, s.solve_flex(d, None, b, x), s, x, nan_ok=True)
    assert np.isnan(res.x).all()
    res = O.solve("fgmres", a, b, tol=1e-12, max_iters=40, restart=8, rs=rs, haptol=1e-3)
    s = K.FgmresSolver(1e-12, 40, 8).with_haptol(1e-3); x = np.zeros(a.nrows)
    check(res, s.solve_flex(d, None, b, x), s, x, nan_ok=True)
    with pytest.raises(K.KError):
        K.FgmresSolver(1e-8, 10, 0).solve_flex(d, None, b, np.zeros(a.nrows))


def test_fgmres_device_vectors_monitor_and_ksp(ctx, rs):
    a = O.stencil7(8, "aniso"); d = to_dev(ctx, a)
    b = a.spmv(np.ones(a.nrows))
    res = O.solve("fgmres", a, b, pc=O.Pc.jacobi(a), tol=1e-9, max_iters=80, restart=20, rs=rs)
    seen = []
    s = K.FgmresSolver(1e-9, 80, 20).with_monitor(lambda i, r: seen.append((i, r)))
    bv, xv = K.DeviceVec(ctx, b), K.DeviceVec(ctx, np.zeros(a.nrows))
    st = s.solve_flex(d, K.Jacobi().setup(d), bv, xv)
    check(res, st, s, xv.to_host())
    assert seen == [(i + 1, r) for i, r in enumerate(res.history)]
    x = np.zeros(a.nrows)
    st = K.KspContext(K.SolverKind.Fgmres, d, pc=None, flex_pc=K.Jacobi().setup(d), tol=1e-9, max_it=80, restart=20).solve_context(b, x)
    assert st.iterations == res.iterations and np.array_equal(x, res.x)
