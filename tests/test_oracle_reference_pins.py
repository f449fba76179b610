"""Pins the CPU oracle against every known-answer test the reference holds for the hot path
(SURVEY.md section 8c).  Each test cites the reference test it re-encodes (paths relative to /root/reference).
The reference's dense test matrices are fed as CSR with every entry stored, which reproduces its dense
row loop (src/core/wrappers.rs:31-36) term for term."""
import numpy as np
import pytest

from oracle import oracle as O

DENSE2 = [[4.0, 1.0], [1.0, 3.0]]
DENSE3 = [[4.0, 1.0, 0.0], [1.0, 3.0, 1.0], [0.0, 1.0, 2.0]]
DENSE4 = [[4.0, 1.0, 0.0, 0.0], [1.0, 3.0, 1.0, 0.0], [0.0, 1.0, 2.0, 1.0], [0.0, 0.0, 1.0, 3.0]]
X2 = [0.09090909090909091, 0.6363636363636364]          # src/solver/cg.rs:317


def rel_error(x, xt):                                    # tests/preconditioner_integration.rs:60-64
    return np.sqrt(((x - xt) ** 2).sum() / (xt ** 2).sum())


# ---- src/matrix/sparse.rs:121-144
def test_identity_spmv():
    m = O.Csr(3, 3, [0, 1, 2, 3], [0, 1, 2], [1.0, 1.0, 1.0])
    x = np.array([2.0, 3.0, 5.0])
    assert np.array_equal(m.spmv(x), x)


def test_simple_pattern():
    m = O.Csr(2, 3, [0, 2, 4], [0, 1, 1, 2], [1.0, 2.0, 3.0, 4.0])
    assert np.array_equal(m.spmv(np.ones(3)), [3.0, 7.0])


def test_csr_preconditions_like_new_checked():            # sparse.rs:36-42
    with pytest.raises(ValueError):
        O.Csr(2, 2, [0, 2, 3], [1, 0, 1], [1.0, 2.0, 3.0])   # unsorted row
    with pytest.raises(ValueError):
        O.Csr(2, 2, [0, 1, 2], [0, 2], [1.0, 2.0])           # column out of range


# ---- tests/core_dense.rs:15-47
def test_dot_and_norm():
    x, y = np.array([1.0, 2.0, 3.0]), np.array([4.0, -5.0, 6.0])
    assert abs(O.dot(x, y) - 12.0) < 1e-12
    assert abs(O.norm(x) - np.sqrt(14.0)) < 1e-12
    for rs in (O.Reduce.tiled(), O.Reduce.tiled(T=64, V=1, F=64)):
        assert abs(O.dot(x, y, rs) - 12.0) < 1e-12
        assert abs(O.norm(x, rs) - np.sqrt(14.0)) < 1e-12


def test_matvec_small_dense():
    rng = np.random.default_rng(5)
    a = rng.random((5, 5)); x = rng.random(5)
    y = O.Csr.from_dense(a).spmv(x)
    for i in range(5):
        exp = 0.0
        for j in range(5):
            exp = exp + a[i, j] * x[j]
        assert y[i] == exp                                 # same order => bit-identical


# ---- src/solver/cg.rs:309-415
@pytest.mark.parametrize("single", [False, True])
def test_cg_solves_simple_spd(single):
    r = O.solve("cg", O.Csr.from_dense(DENSE2), [1.0, 2.0], tol=1e-10, max_iters=20, single_reduction=single)
    assert r.converged and np.all(np.abs(r.x - X2) < 1e-8)


@pytest.mark.parametrize("single", [False, True])
def test_cg_solves_spd3(single):
    a = O.Csr.from_dense(DENSE3)
    b = a.spmv([1.0, 2.0, 3.0])
    r = O.solve("cg", a, b, tol=1e-10, max_iters=100, single_reduction=single)
    assert r.converged and np.linalg.norm(b - a.spmv(r.x)) <= 1e-8


# ---- src/solver/pcg.rs:253-275
def test_pcg_single_reduction_equivalence():
    a = O.Csr.from_dense(DENSE2)
    r1 = O.solve("pcg", a, [1.0, 2.0], pc=O.Pc.identity(), tol=1e-10, max_iters=20)
    r2 = O.solve("pcg", a, [1.0, 2.0], pc=O.Pc.identity(), tol=1e-10, max_iters=20, single_reduction=True)
    assert r2.converged and np.all(np.abs(r1.x - r2.x) < 1e-8) and np.all(np.abs(r2.x - X2) < 1e-8)


# ---- src/solver/gmres.rs:438-528
def _gm4():
    a = O.Csr.from_dense(DENSE4)
    xt = np.array([1.0, 2.0, 3.0, 4.0])
    return a, a.spmv(xt), xt


def test_gmres_solves_well_conditioned_nonsym():
    a, b, xt = _gm4()
    r = O.solve("gmres", a, b, tol=1e-10, max_iters=100, restart=4)
    assert r.converged and np.all(np.abs(r.x - xt) < 1e-8)


def test_gmres_with_jacobi_left():
    a, b, xt = _gm4()
    r = O.solve("gmres", a, b, pc=O.Pc.jacobi(a), tol=1e-10, max_iters=100, restart=4)
    assert r.converged and np.all(np.abs(r.x - xt) < 1e-8)


def test_gmres_with_jacobi_right():
    a, b, xt = _gm4()
    r = O.solve("gmres", a, b, pc=O.Pc.jacobi(a), tol=1e-10, max_iters=100, restart=4, side=O.SIDE_RIGHT)
    assert np.linalg.norm(a.spmv(r.x) - b) < 1e-2        # gmres.rs:521-527: convergence NOT asserted


# ---- src/solver/bicgstab.rs:303-328
def test_bicgstab_nonsym_3x3():
    a = np.array([[4.0 if i == j else (i + 2 * j) + 1.0 for j in range(3)] for i in range(3)])
    xt = np.array([1.0, 2.0, 3.0])
    A = O.Csr.from_dense(a)
    r = O.solve("bicgstab", A, A.spmv(xt), tol=1e-10, max_iters=100)
    assert r.converged and np.all(np.abs(r.x - xt) < 1e-8)


# ---- src/preconditioner/chebyshev.rs:184-206 (finite-only in the reference)
def test_chebyshev_identity_and_diagonal():
    z = O.apply_chebyshev(O.Csr.from_dense([[1.0, 0.0], [0.0, 1.0]]), [2.0, 3.0], 1.0, 1.0, 1)
    assert np.array_equal(z, [2.0, 3.0])                   # degenerate interval copies r (chebyshev.rs:88-92)
    z = O.apply_chebyshev(O.Csr.from_dense([[2.0, 0.0], [0.0, 3.0]]), [1.0, 1.0], 2.0, 3.0, 1)
    assert np.all(np.isfinite(z))
    assert np.array_equal(z, [(2.0 - 2.5) / 0.5, (3.0 - 2.5) / 0.5])   # m == 1: v1 unscaled (chebyshev.rs:112-116)


def test_chebyshev_trait_apply_is_a_stub():               # chebyshev.rs:68-70
    with pytest.raises(O.KrylovError) as e:
        O.Pc.chebyshev_stub().apply([1.0, 2.0])
    assert e.value.code == O.SOLVE_ERROR


# ---- src/preconditioner/ilup.rs:202-228
def test_ilup_identity_and_tridiag():
    a = O.Csr.from_dense([[1.0, 0.0], [0.0, 1.0]])
    assert np.array_equal(O.Pc.ilup0(a).apply([2.0, 3.0]), [2.0, 3.0])
    a = O.Csr.from_dense([[2.0, -1.0, 0.0], [-1.0, 2.0, -1.0], [0.0, -1.0, 2.0]])
    assert np.all(np.isfinite(O.Pc.ilup0(a).apply([1.0, 2.0, 3.0])))


# ---- src/preconditioner/ilu.rs:59-122 -- dense transcription of the reference code, checked against the sparse form
def _ilu0_dense_as_written(a):
    n = len(a)
    l = np.zeros((n, n)); u = np.zeros((n, n))
    for i in range(n):
        u[i, i] = a[i, i]
        for j in range(i + 1, n):
            if a[i, j] != 0.0:
                u[i, j] = a[i, j]
        l[i, i] = 1.0
        for j in range(i + 1, n):
            if a[j, i] != 0.0:
                l[j, i] = a[j, i] / u[i, i]
        for j in range(i + 1, n):
            for k in range(i + 1, n):
                if a[j, k] != 0.0:
                    v = a[j, k] - l[j, i] * u[i, k]
                    if v != 0.0:
                        if k >= j:
                            u[j, k] = v
                        else:
                            l[j, k] = v
    return l, u


def _ilu0_dense_apply(l, u, x):
    y = np.array(x, dtype=float); n = len(y)
    for i in range(n):
        for j in range(i):
            y[i] = y[i] - l[i, j] * y[j]
    for i in range(n - 1, -1, -1):
        for j in range(i + 1, n):
            y[i] = y[i] - u[i, j] * y[j]
    return y


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_ilu0_compat_matches_dense_transcription(seed):
    rng = np.random.default_rng(seed)
    n = 9
    a = rng.random((n, n)) * (rng.random((n, n)) < 0.5) + np.diag(2.0 + rng.random(n))
    l, u = _ilu0_dense_as_written(a)
    x = rng.random(n)
    for keep in (True, False):
        z = O.Pc.ilu0_compat(O.Csr.from_dense(a, keep_zeros=keep)).apply(x)
        assert np.array_equal(z, _ilu0_dense_apply(l, u, x))


# ---- tests/preconditioner_integration.rs:81-179
def _ill_cond(n, kappa):
    d = np.ones(n); d[-1] = kappa
    return O.Csr.from_dense(np.diag(d)), np.ones(n)


def test_cg_with_jacobi_smoke():                           # :82-95 (PCG called with pc=None)
    a, b = _ill_cond(5, 1e6)
    assert np.array_equal(O.Pc.jacobi(a).apply(b), [1.0, 1.0, 1.0, 1.0, 1e-6])
    assert O.solve("pcg", a, b, tol=1e-6, max_iters=1000).converged


def test_gmres_with_ilu0_smoke():                          # :100-109 (pc built but not passed)
    a, b = _ill_cond(5, 1e4)
    O.Pc.ilu0_compat(a)
    assert O.solve("gmres", a, b, tol=1e-6, max_iters=1000, restart=4).converged


def test_pcg_with_jacobi_illcond():                        # :114-122
    a, b = _ill_cond(5, 1e6)
    assert O.solve("pcg", a, b, pc=O.Pc.jacobi(a), tol=1e-6, max_iters=1000).converged


@pytest.mark.parametrize("keep", [True, False])
def test_spd_jacobi_pcg_converges(keep):                   # :127-138
    n = 10
    a = O.Csr.from_dense(O.tridiag(n, -1.0, 2.0, -1.0), keep_zeros=keep)
    b = a.spmv(np.ones(n))
    r = O.solve("pcg", a, b, pc=O.Pc.jacobi(a), tol=1e-12, max_iters=n)
    assert r.converged and rel_error(r.x, np.ones(n)) < 1e-10 and r.iterations <= n


def test_spd_no_pc_cg_converges():                         # :143-151
    n = 10
    a = O.Csr.from_dense(O.tridiag(n, -1.0, 2.0, -1.0))
    r = O.solve("pcg", a, a.spmv(np.ones(n)), tol=1e-12, max_iters=n)
    assert r.converged and rel_error(r.x, np.ones(n)) < 1e-10


def test_nonsym_no_pc_gmres_converges():                   # :156-164
    n = 10
    a = O.Csr.from_dense(O.tridiag(n, -1.0, 2.0, 0.5))
    r = O.solve("gmres", a, a.spmv(np.ones(n)), tol=1e-12, max_iters=100, restart=10)
    assert r.converged and rel_error(r.x, np.ones(n)) < 1e-10


def test_nonsym_left_ilu0_gmres_converges():               # :169-179
    n = 10
    a = O.Csr.from_dense(O.tridiag(n, -1.0, 2.0, 0.5))
    r = O.solve("gmres", a, a.spmv(np.ones(n)), pc=O.Pc.ilu0_compat(a), tol=1e-12, max_iters=100, restart=10,
                side=O.SIDE_LEFT)
    assert r.converged and rel_error(r.x, np.ones(n)) < 1e-10
    # the reference's non-standard Left variant needs two restart cycles on this system
    assert r.iterations == 20


# ---- src/utils/convergence.rs:25: the iteration cap reports converged = true
def test_iteration_cap_reports_converged():
    a = O.stencil7(6)
    r = O.solve("cg", a, a.spmv(np.ones(a.nrows)), tol=1e-30, max_iters=3)
    assert r.iterations == 3 and r.converged
    r = O.solve("pcg", a, a.spmv(np.ones(a.nrows)), pc=O.Pc.jacobi(a), tol=1e-30, max_iters=3)
    assert r.iterations == 3 and r.converged


# ---- cg.rs:168-174 / pcg.rs:162-172 error returns
def test_indefinite_matrix_error():
    a = O.Csr.from_dense([[1.0, 0.0], [0.0, -1.0]])
    for m in ("cg", "pcg"):
        with pytest.raises(O.KrylovError) as e:
            O.solve(m, a, [0.0, 1.0], tol=1e-10, max_iters=10)
        assert e.value.code == O.INDEFINITE_MATRIX


# ---- src/preconditioner/ilut.rs:185-211 and the level-of-fill path of ilup.rs
def test_ilut_identity_and_tridiag():
    a = O.Csr.from_dense([[1.0, 0.0], [0.0, 1.0]])
    assert np.array_equal(O.Pc.ilut(a, 2, 1e-12).apply([2.0, 3.0]), [2.0, 3.0])
    a = O.Csr.from_dense([[2.0, -1.0, 0.0], [-1.0, 2.0, -1.0], [0.0, -1.0, 2.0]])
    assert np.all(np.isfinite(O.Pc.ilut(a, 3, 1e-12).apply([1.0, 2.0, 3.0])))


def test_ilup_dense_as_written_equals_sparse_restatement_at_fill_0():
    a = O.stencil7(5, "convdiff")
    r = O.splitmix64_uniform(3, a.nrows)
    assert np.array_equal(O.Pc.ilup(a, 0).apply(r), O.Pc.ilup0(a).apply(r))
    # with fill >= 1 eliminations happen: on a tridiagonal matrix (no fill-in possible) Ilup(1) is the exact LU
    t = O.Csr.from_dense(O.tridiag(12, -1.0, 2.0, -1.0), keep_zeros=False)
    x = np.arange(1.0, 13.0)
    assert np.allclose(O.Pc.ilup(t, 1).apply(t.spmv(x)), x, rtol=1e-12)


# ---- src/solver/fgmres.rs:537-552 fgmres_equiv_to_gmres_on_fixed_pc
def test_fgmres_on_fixed_jacobi():
    a = O.Csr.from_dense([[2.0, 1.0], [1.0, 3.0]])
    xt = np.array([1.0, 2.0])
    r = O.solve("fgmres", a, a.spmv(xt), pc=O.Pc.jacobi(a), tol=1e-10, max_iters=100, restart=25)
    assert r.converged and np.all(np.abs(r.x - xt) < 1e-6)
    # quirk: final_residual reports the INITIAL residual norm (fgmres.rs:171,339)
    assert r.final_residual == np.linalg.norm(a.spmv(xt))
    r0 = O.solve("fgmres", a, np.zeros(2), tol=1e-10, max_iters=100, restart=25)        # beta == 0 early return (:141-143)
    assert r0.converged and r0.iterations == 0 and r0.final_residual == 0.0


# ---- src/solver/cgs.rs:155-188 cgs_solves_large_well_conditioned_nonsym
def test_cgs_reference_known_answer():
    a = O.Csr.from_dense([[10.0, 2, 0, 0, 0], [3, 15, 4, 0, 0], [0, -2, 8, 1, 0], [0, 0, 1, 7, 3], [0, 0, 0, 2, 12]])
    xt = np.arange(1.0, 6.0)
    r = O.solve("cgs", a, a.spmv(xt), tol=1e-10, max_iters=200)
    assert r.converged and np.all(np.abs(r.x - xt) <= 1e-6)


# ---- src/solver/tfqmr.rs:241-259: the reference's only TFQMR test is #[ignore] ("may not pass in all environments").
# The restatement agrees: as written the method does NOT solve that 2x2 system, so there is nothing to pin beyond the
# structural facts below (x0 discarded :72, rho == 0 early return :81-83, stop at the cap after the FIRST substep).
def test_tfqmr_as_written_structure():
    a = O.Csr.from_dense([[2.0, 1.0], [3.0, 4.0]])
    b = a.spmv(np.array([1.0, 2.0]))
    r1 = O.solve("tfqmr", a, b, x0=np.array([5.0, -3.0]), tol=1e-10, max_iters=500)
    r2 = O.solve("tfqmr", a, b, tol=1e-10, max_iters=500)
    assert np.array_equal(r1.x, r2.x) and r1.iterations == r2.iterations          # the initial guess is ignored
    assert not (np.all(np.abs(r2.x - [1.0, 2.0]) < 1e-3) and r2.converged)        # what the ignored test would assert
    r0 = O.solve("tfqmr", a, np.zeros(2), x0=np.array([1.0, 1.0]), tol=1e-10, max_iters=5)
    assert r0.converged and r0.iterations == 0 and not r0.x.any()
    a5 = O.stencil7(4, "convdiff"); b5 = a5.spmv(np.ones(a5.nrows))
    r = O.solve("tfqmr", a5, b5, tol=1e-30, max_iters=3)
    assert r.iterations == 3 and r.converged and len(r.history) == 5              # 2 + 2 + 1 residual estimates


# ---- src/preconditioner/approxinv.rs:384-443: the apply side of the SPAI tests, with the inverse rows those tests expect
# from setup (diagonal / identity cases are exact; setup itself -- faer's least squares -- is not restated)
def test_approxinv_apply_with_given_rows():
    d = O.Csr.from_dense(np.diag([0.5, 1.0 / 3.0, 0.25]), keep_zeros=False)           # :384-397 inverse of diag(2,3,4)
    z = O.Pc.approx_inverse(d).apply(np.array([2.0, 3.0, 4.0]))
    assert np.array_equal(z, [1.0, 1.0, 1.0])
    eye = O.Csr.from_dense(np.eye(4), keep_zeros=False)                               # :428-443 identity
    x = np.array([1.0, 2.0, 3.0, 4.0])
    assert np.array_equal(O.Pc.approx_inverse(eye).apply(x), x)
