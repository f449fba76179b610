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
    """192^3 (7 M rows, 576 8 x 8 line blocks: more than are resident at once): the 16 x 16 wavefront solve, the 8 x 8 three-wave
    solve, its one-wave predecessor and the level-ordered sync-free solve give the same bits for the same factors."""
    M = 192
    a = K.CsrMatrix.stencil7(M, "aniso", ctx=ctx)
    n = a.nrows()
    r = ctx.vec(n).fill_splitmix(11)
    outs = []
    for grid, wave in (("1", "2"), ("1", "1"), ("1", "0"), ("0", "1")):
        monkeypatch.setenv("KRYST_ILU_GRID", grid); monkeypatch.setenv("KRYST_ILU_WAVE", wave)
        pc = K.TrueIlu0().setup(a)
        z = ctx.vec(n)
        pc.apply(r, z); pc.apply(r, z)                   # twice: the second apply starts from stale sentinels / results
        outs.append(z.to_host())
        del pc
    assert np.all(np.isfinite(outs[0]))
    assert all(np.array_equal(outs[0], o) for o in outs[1:])


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


# ------------------------------------------------------------------------------------------------ BASELINE configs 3, 4 and 5 at full size
def _true_residual(ctx, a, b, x):
    """||b - A x|| with the device's own SpMV, subtraction (`bi - ax`, cg.rs:123 / gmres.rs:224) and norm."""
    ax = a.spmv(x)
    r = ctx.vec(a.nrows())
    K.check(K.lib().kryst_sub(b.h, ax.h, r.h))
    return K.norm(r)


def test_config3_gmres30_left_jacobi_convdiff_256(ctx):
    """BASELINE configs[2]: GmresSolver(30) Left + Jacobi on the 256^3 upwind convection-diffusion operator, tol 1e-8, max 600
    (src/solver/gmres.rs:216-402).  At this size the oracle cannot be the checker, so: two runs give the same bits; the
    reported final_residual IS the true residual ||b - A x|| of the returned x (gmres.rs:387-393 recomputes it after the last
    cycle), recomputed here with the device SpMV; `converged` is exactly `final_residual < tol * res0` (:394); the iteration
    count is a whole number of cycles or the cap.  (As written, Left-preconditioned GMRES orthogonalises against Z with
    Z[0] = M^-1 v0 -- DESIGN.md section 2 -- and does not reach 1e-8 on this operator within 600 iterations: the reference's
    behaviour, reproduced, not a defect of the port; the bit-level check against the oracle is the 64^3 test below.)"""
    n = N ** 3
    a = K.CsrMatrix.stencil7(N, "convdiff", ctx=ctx)
    b = a.spmv(ctx.vec(n).fill(1.0))
    res0 = K.norm(b)
    pc = K.Jacobi().setup(a)
    runs = []
    for _ in range(2):
        x = ctx.vec(n)
        s = K.GmresSolver(30, 1e-8, 600)
        st = s.solve(a, pc, b, x)
        runs.append((st.iterations, st.converged, st.final_residual, tuple(s.residual_history)))
    assert runs[0] == runs[1]                                              # run-to-run bit-identical
    its, conv, fin, hist = runs[0]
    assert its == 600 or its % 30 == 0 or conv
    true_res = _true_residual(ctx, a, b, x)
    assert abs(true_res - fin) <= 1e-12 * res0, (true_res, fin)          # tolerance: 1e-12 relative to ||b|| (north_star)
    assert conv == (fin < 1e-8 * res0)
    assert np.isfinite(fin) and fin < res0                                 # it does make progress
    assert len(hist) == its and all(np.isfinite(hist))


def test_config3_first_cycles_match_the_oracle_bitwise_64(ctx):
    """The same solver at 64^3 (262 144 rows), where the oracle can follow: two restart cycles of GMRES(30) Left + Jacobi on
    the convection-diffusion operator -- every residual-history entry, the iteration count and x, bit for bit."""
    from oracle import oracle as O
    M = 64
    ao = O.stencil7(M, "convdiff")
    a = K.CsrMatrix.stencil7(M, "convdiff", ctx=ctx)
    b = ao.spmv(np.ones(ao.nrows))
    T, V, F = K.reduce_spec()
    ref = O.solve("gmres", ao, b, pc=O.Pc.jacobi(ao), tol=1e-8, max_iters=60, restart=30, side=O.SIDE_LEFT, rs=O.Reduce.tiled(T, V, F))
    s = K.GmresSolver(30, 1e-8, 60); x = np.zeros(ao.nrows)
    st = s.solve(a, K.Jacobi().setup(a), b, x)
    assert (st.iterations, st.converged, st.final_residual) == (ref.iterations, ref.converged, ref.final_residual)
    assert np.array_equal(np.array(s.residual_history), ref.history) and np.array_equal(x, ref.x)


def test_config5_bicgstab_true_ilu0_aniso_256(ctx, monkeypatch):
    """BASELINE configs[4] as the labelled extension (the reference's BiCgStabSolver ignores pc, bicgstab.rs:70): right-
    preconditioned BiCGStab + textbook ILU(0) on the 256^3 anisotropic operator, absolute tol 1e-8 ||b|| (bicgstab.rs:69-293).
    Converges; two runs give the same bits; the true residual of the returned x meets the tolerance; and the wavefront
    triangular solve it runs on is checked at this size against the plane kernels (one launch per hyperplane, no
    inter-workgroup waits: an independent implementation of the same row order): identical bits for z = U^-1 L^-1 r."""
    n = N ** 3
    a = K.CsrMatrix.stencil7(N, "aniso", ctx=ctx)
    b = a.spmv(ctx.vec(n).fill(1.0))
    bn = K.norm(b)
    pc = K.TrueIlu0().setup(a)
    runs = []
    for _ in range(2):
        x = ctx.vec(n)
        s = K.BiCgStabRightPcSolver(1e-8 * bn, 1000)
        st = s.solve(a, pc, b, x)
        runs.append((st.iterations, st.converged, st.final_residual, tuple(s.residual_history)))
    assert runs[0] == runs[1]
    its, conv, fin, _ = runs[0]
    assert conv and 10 < its < 1000
    assert _true_residual(ctx, a, b, x) <= 1.5e-8 * bn                    # the recurrence residual met 1e-8 ||b||; the true one follows it
    assert np.max(np.abs(x.to_host() - 1.0)) < 1e-4
    r = ctx.vec(n).fill_splitmix(77)
    z_wave = ctx.vec(n); pc.apply(r, z_wave)
    monkeypatch.setenv("KRYST_ILU_PLANES", "1")
    pc2 = K.TrueIlu0().setup(a)
    z_planes = ctx.vec(n); pc2.apply(r, z_planes)
    d = ctx.vec(n); d.copy_from(z_wave); K.axpy(-1.0, z_planes, d)
    assert K.norm(d) == 0.0 and np.isfinite(K.norm(z_wave))


@pytest.mark.gpu
def test_config4_jacobi_pcg_512_cubed_on_one_gpu(ctx):
    """BASELINE configs[3]'s workload -- Jacobi-PCG (pcg.rs:114-222) on the 512^3 Poisson system, tol 1e-8, max 3000 -- on ONE GPU
    (its 8-way partition needs 8 GPUs; the partitioned code path is covered at small sizes by the multi-rank tests).  Size-independent
    properties: it converges below the cap, two solves give the same history bit for bit, the iteration count is close to
    unpreconditioned CG's (the Jacobi-scaled system has the same Krylov spaces; PCG's mixed-norm test -- res0 = sqrt|r.z| = ||r0||/sqrt 6,
    pcg.rs:134 vs :192 -- is sqrt 6 stricter), and the TRUE residual meets the tolerance the solver reports."""
    M = 512
    n = M ** 3
    a = K.CsrMatrix.stencil7(M, "poisson", ctx=ctx)
    ones = ctx.vec(n).fill(1.0)
    b = a.spmv(ones)
    bn = K.norm(b)
    pc = K.Jacobi().setup(a)
    runs = []
    for _ in range(2):
        x = ctx.vec(n)
        s = K.PcgSolver(1e-8, 3000)
        st = s.solve(a, pc, b, x)
        runs.append((st.iterations, st.converged, st.final_residual, list(s.residual_history)))
    assert runs[0] == runs[1]
    its, conv, fr, hist = runs[0]
    assert conv and 100 < its < 3000 and len(hist) == its + 1
    assert hist[0] == bn                                       # the history holds ||r|| (pcg.rs:140,192); x0 = 0, so r0 = b
    assert fr <= 1e-8 * bn / np.sqrt(6.0) * (1.0 + 1e-9)       # ... while the test is against res0 = sqrt(|r0 . D^-1 r0|) = ||b|| / sqrt 6 (pcg.rs:134)
    assert hist[-2] > 1e-8 * bn / np.sqrt(6.0)                 # and it stopped at the first iterate that meets it
    tr = _true_residual(ctx, a, b, x)
    assert tr <= 1e-8 * bn and abs(tr - fr) <= 1e-3 * fr + 1e-12 * bn                          # the recurrence residual is the true one
    xc = ctx.vec(n)
    sc = K.CgSolver(1e-8, 3000)
    stc = sc.solve(a, None, b, xc)
    assert stc.converged and abs(stc.iterations - its) <= 60, (stc.iterations, its)
    K.axpy(-1.0, ones, x)
    assert K.norm(x) <= 1e-3 * np.sqrt(n)                                                      # x* = 1: error <= kappa * 1e-8


@pytest.mark.gpu
def test_cg_forms_bit_identical_at_512_cubed(ctx, monkeypatch):
    """Round 5: at 512^3 CG / PCG take, by default, the direction pass inside the staged-window SpMV (spmv_pattern_fuse_kernel) with x updated in
    batches of 8 / 7 iterations from a ring of direction vectors (XBatchOp) -- a form the small parity cases only reach through environment knobs.
    Here the DEFAULT path at the size that selects it, for an iteration count that ends inside a batch: residual history and x must be bit for bit
    those of the unfused form (KRYST_CG_FUSE_P=0), which the oracle pins at 256^3 and in bench.py's parity_at_size, and of a second default run."""
    grid, iters = 512, 43
    a = K.CsrMatrix.stencil7(grid, "poisson", ctx=ctx)
    n = a.nrows()
    b = a.spmv(ctx.vec(n).fill(1.0))
    pcj = K.Jacobi().setup(a)
    for cls, pc in ((K.CgSolver, None), (K.PcgSolver, pcj)):
        runs = []
        for fuse in ("0", None, None):
            if fuse is None:
                monkeypatch.delenv("KRYST_CG_FUSE_P", raising=False)
            else:
                monkeypatch.setenv("KRYST_CG_FUSE_P", fuse)
            x = ctx.vec(n)
            s = cls(0.0, iters)
            st = s.solve(a, pc, b, x)
            runs.append((st.iterations, np.array(s.residual_history), x.to_host()))
        assert runs[0][0] == iters
        for r in runs[1:]:
            assert r[0] == runs[0][0] and np.array_equal(r[1], runs[0][1]) and np.array_equal(r[2], runs[0][2]), cls.__name__


@pytest.mark.gpu
def test_sizes_at_the_index_limits(ctx, monkeypatch):
    """648^3 = 272 M rows is beyond the CSR-P16 kernel's 32-bit byte offsets (2^28 rows): the operator must fall back to CSR-D16 and say
    so, and every form must still give the exact row sums of the Poisson operator (A 1 = number of missing neighbours) and agree bit for
    bit; 768^3 (3.2 G entries) exceeds the int32 entry indexing and must be refused with an argument error, not attempted."""
    with pytest.raises(K.KError) as e:
        K.CsrMatrix.stencil7(768, "poisson", ctx=ctx)
    assert e.value.code == 102
    M = 648
    n = M ** 3
    assert n > 1 << 28 and 7 * n - 6 * M * M < (1 << 31) - 16
    monkeypatch.delenv("KRYST_SPMV_COMPRESS", raising=False)
    a = K.CsrMatrix.stencil7(M, "poisson", ctx=ctx)
    assert a.encoding()[0] == "csr-d16"
    ones = ctx.vec(n).fill(1.0)
    x = ctx.vec(n).fill_splitmix(0xC0FFEE)
    ref = None
    for comp in ("3", "1", "0"):
        monkeypatch.setenv("KRYST_SPMV_COMPRESS", comp)
        y = a.spmv(ones)
        assert K.dot(y, ones) == float(6 * M * M)
        assert K.dot(y, y) == float(6 * (M - 2) ** 2 + 4 * 12 * (M - 2) + 9 * 8)
        z = a.spmv(x)
        if ref is None:
            ref = z
        else:
            d = ctx.vec(n); d.copy_from(z); K.axpy(-1.0, ref, d)
            assert K.norm(d) == 0.0
        del y
