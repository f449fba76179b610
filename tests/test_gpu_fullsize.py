"""Size-independent properties at BASELINE.json's full sizes (256^3 and the bench's 512^3), where the CPU oracle would be
too slow to be the checker."""

import numpy as np
import pytest

import kryst_amd as K

pytestmark = pytest.mark.gpu
N = 256


@pytest.fixture(scope="module")
def ctx():
    return K.Context(0)


@pytest.fixture(scope="module")
def poisson(ctx):
    return K.CsrMatrix.stencil7(N, "poisson", ctx=ctx)


def test_spmv_row_sums_are_exact(ctx, poisson):
    """A*1 for the 7-point Poisson operator is 6 - (#neighbours): small integers, exact in any summation order."""
    n = N ** 3
    y = poisson.spmv(ctx.vec(n).fill(1.0)).to_host().reshape(N, N, N)       # [k][j][i]
    idx = np.arange(N)
    edge = ((idx == 0) | (idx == N - 1)).astype(np.float64)
    exp = edge[:, None, None] + edge[None, :, None] + edge[None, None, :]
    assert np.array_equal(y, exp)


def test_spmv_kernel_forms_agree_bitwise(ctx, poisson, monkeypatch):
    n = N ** 3
    x = ctx.vec(n).fill_splitmix(0xC0FFEE)
    outs = []
    for kernel, comp in (("2", "0"), ("3", "0"), ("3", "1"), ("3", "2"), ("3", "3")):
        monkeypatch.setenv("KRYST_SPMV_KERNEL", kernel); monkeypatch.setenv("KRYST_SPMV_COMPRESS", comp)
        outs.append(poisson.spmv(x).to_host())
    assert all(np.array_equal(outs[0], o) for o in outs[1:])
    # linearity within rounding: A(2x) == 2 A(x) exactly (scaling by 2 is exact)
    x2 = ctx.vec(n); K.axpy(2.0, x, x2)
    assert np.array_equal(poisson.spmv(x2).to_host(), 2.0 * outs[0])


@pytest.mark.parametrize("n", [256 ** 3, 512 ** 3 // 4 + 12345, 512 ** 3])
def test_dot_of_ones_is_exact(ctx, n):
    """(1,1) = n exactly in every association order; exercises the two-level fold with 32 .. 256 chunks."""
    v = ctx.vec(n).fill(1.0)
    assert K.dot(v, v) == float(n)
    assert K.norm(v) == float(np.sqrt(float(n)))


def test_cg_full_size_converges_and_is_deterministic(ctx, poisson):
    n = N ** 3
    b = poisson.spmv(ctx.vec(n).fill(1.0))
    hist = []
    for _ in range(2):
        x = ctx.vec(n)
        s = K.CgSolver(1e-8, 2000)
        st = s.solve(poisson, None, b, x)
        assert st.converged and 300 < st.iterations < 1500
        hist.append((st.iterations, tuple(s.residual_history)))
    assert hist[0] == hist[1]                                   # run-to-run bit-identical
    r = poisson.spmv(x); K.axpy(-1.0, b, r)
    assert K.norm(r) <= 1.01e-8 * K.norm(b)
    assert np.max(np.abs(x.to_host() - 1.0)) < 1e-5


def test_pcg_jacobi_equals_cg_on_constant_diagonal(ctx, poisson):
    """Jacobi on a constant diagonal only rescales: PCG must take the same number of iterations as CG (+-1: mixed norms,
    pcg.rs:134 vs :192) -- the fused Jacobi update kernel at full size."""
    n = N ** 3
    b = poisson.spmv(ctx.vec(n).fill(1.0))
    x1, x2 = ctx.vec(n), ctx.vec(n)
    st1 = K.CgSolver(1e-8, 2000).solve(poisson, None, b, x1)
    st2 = K.PcgSolver(1e-8, 2000).solve(poisson, K.Jacobi().setup(poisson), b, x2)
    assert st1.converged and st2.converged and abs(st1.iterations - st2.iterations) <= 20
    K.axpy(-1.0, x1, x2)
    assert K.norm(x2) <= 1e-6 * K.norm(x1)


def test_ilu_apply_inverts_its_factors(ctx):
    """z = U^-1 L^-1 r  =>  L U z == r to rounding, checked with the factors applied forward on the host (64^3)."""
    from oracle import oracle as O
    M = 48
    a = K.CsrMatrix.stencil7(M, "aniso", ctx=ctx)
    ao = O.stencil7(M, "aniso")
    pc = K.TrueIlu0().setup(a)
    r = O.splitmix64_uniform(5, ao.nrows)
    z = pc.apply(r)
    ref = O.Pc.ilu0_true(ao)
    assert np.array_equal(z, ref.apply(r))


def test_ilu_forms_agree_at_scale(ctx, monkeypatch):
    """192^3 (7 M rows, 576 line blocks: more than are resident at once): the three-wave wavefront solve, its one-wave
    predecessor and the level-ordered sync-free solve give the same bits for the same factors."""
    M = 192
    a = K.CsrMatrix.stencil7(M, "aniso", ctx=ctx)
    n = a.nrows()
    r = ctx.vec(n).fill_splitmix(11)
    outs = []
    for grid, wave in (("1", "1"), ("1", "0"), ("0", "1")):
        monkeypatch.setenv("KRYST_ILU_GRID", grid); monkeypatch.setenv("KRYST_ILU_WAVE", wave)
        pc = K.TrueIlu0().setup(a)
        z = ctx.vec(n)
        pc.apply(r, z); pc.apply(r, z)                   # twice: the second apply starts from stale sentinels / results
        outs.append(z.to_host())
        del pc
    assert np.all(np.isfinite(outs[0]))
    assert np.array_equal(outs[0], outs[1]) and np.array_equal(outs[0], outs[2])


def test_512_cubed_properties(ctx, monkeypatch):
    """The bench's own problem (512^3, 134 M rows, 938 M nonzeros): exact row sums of the Poisson operator, every SpMV storage
    form bit-identical on a random vector (compared on the device through a checksum of per-plane checksums and a full
    difference norm), and a deterministic CG run whose residual history is strictly decreasing at the start."""
    M = 512
    n = M ** 3
    monkeypatch.delenv("KRYST_SPMV_COMPRESS", raising=False)
    a = K.CsrMatrix.stencil7(M, "poisson", ctx=ctx)
    assert a.nnz == 7 * n - 6 * M * M and a.encoding()[0] == "csr-p16"
    ones = ctx.vec(n).fill(1.0)
    y = a.spmv(ones)
    # A*1 = number of missing neighbours: 3 at corners, 0 inside; sum = 6 N^2, sum of squares = (N-2)^2*6 + 4*12*(N-2) + 9*8
    assert K.dot(y, ones) == float(6 * M * M)
    assert K.dot(y, y) == float(6 * (M - 2) ** 2 + 4 * 12 * (M - 2) + 9 * 8)
    x = ctx.vec(n).fill_splitmix(0xC0FFEE)
    outs = []
    for comp in ("3", "2", "1", "0"):
        monkeypatch.setenv("KRYST_SPMV_COMPRESS", comp)
        outs.append(a.spmv(x))
    monkeypatch.delenv("KRYST_SPMV_COMPRESS", raising=False)
    ref = outs[-1]
    for o in outs[:-1]:
        d = ctx.vec(n); d.copy_from(o); K.axpy(-1.0, ref, d)
        assert K.norm(d) == 0.0                                     # every entry identical (a NaN anywhere would also fail)
        assert K.dot(o, ones) == K.dot(ref, ones)                   # checksum in the library's fixed association order
    b = a.spmv(ones)
    hist = []
    for _ in range(2):
        xs = ctx.vec(n)
        s = K.CgSolver(0.0, 25)
        st = s.solve(a, None, b, xs)
        assert st.iterations == 25
        hist.append(list(s.residual_history))
    assert hist[0] == hist[1]
    assert all(hist[0][i + 1] < hist[0][i] for i in range(5))
