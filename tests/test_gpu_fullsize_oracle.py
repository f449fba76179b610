"""Full-size parity AGAINST THE ORACLE (VERDICT r03 "missing" 3): BASELINE.json's 256^3 configurations solved on the GPU and by the
CPU restatement of the reference (oracle/kryst_oracle.c, 16 OpenMP threads for the pointwise loops; test infrastructure), whole residual
histories compared -- BIT FOR BIT in the library's dot order (the library's tree is one admissible execution of the reference's Rayon
reduce, whose association is unspecified, src/core/wrappers.rs:92-100), and against the reference's strict serial left fold
(--no-default-features, wrappers.rs:101-107): equal iteration counts and histories within SERIAL_TOL * ||r0||.
north_star's 1e-12 holds up to 64^3 (tests/test_golden.py, test_gpu_0_parity.py); at 256^3 a serial fold over 16.8 M terms carries a
rounding error of its own of up to n * eps = 1.9e-9 relative per inner product (the tiled tree's bound is ~30 eps), and the two
association orders measured here differ by 1.7e-11 * ||r0|| over config 2's 581 iterations (printed by the tests; profiles/r04/).  The
bound below is therefore 1e-10, stated as what it is: the distance between two legal executions of the reference, not an error of the
port -- the port's own arithmetic is pinned by the bit-for-bit half.  The oracle is the checker; nothing here is timed."""
import os

import numpy as np
import pytest

import kryst_amd as K

pytestmark = pytest.mark.gpu
N = 256
SERIAL_TOL = 1e-10          # 256^3, against the strict serial fold (see the module docstring); BiCGStab: 1e-9


@pytest.fixture(scope="module")
def ctx():
    return K.Context(0)


def _oracle_system(kind):
    from oracle import oracle as O
    O.set_threads(min(len(os.sched_getaffinity(0)), 16))
    rp, ci, va = K.host_stencil7(N, kind)
    a = O.Csr(N ** 3, N ** 3, rp, ci, va, check=False)
    return O, a, a.spmv(np.ones(a.nrows))


def _compare(name, gpu_hist, gpu_stats, gpu_x, ref_tiled, ref_serial, serial_tol, serial_entries=None):
    h = np.array(gpu_hist)
    # the library's association tree (kryst_reduce_spec): everything bit for bit
    assert (gpu_stats.iterations, bool(gpu_stats.converged), gpu_stats.final_residual) == (ref_tiled.iterations, ref_tiled.converged, ref_tiled.final_residual), name
    assert len(h) == len(ref_tiled.history) and np.array_equal(h, ref_tiled.history), name
    assert np.array_equal(gpu_x, ref_tiled.x), name
    # the reference's --no-default-features fold: same iteration count, history within the stated tolerance of ||r0||
    assert ref_serial.iterations == gpu_stats.iterations, name
    m = len(h) if serial_entries is None else min(len(h), serial_entries)
    dev = float(np.max(np.abs(h[:m] - ref_serial.history[:m])) / ref_serial.history[0])
    growth = ", ".join(f"{k}: {float(np.max(np.abs(h[:k] - ref_serial.history[:k])) / ref_serial.history[0]):.1e}" for k in (10, 25, 50, len(h)) if k <= len(h))
    print(f"[full-size parity] {name}: {gpu_stats.iterations} iterations, {len(h)} history entries bit-identical to the tiled-order oracle; "
          f"max |history - serial-fold history| / ||r0|| over the first {m} entries = {dev:.3e} (bound {serial_tol:.0e}); by entries {{{growth}}}")
    assert dev <= serial_tol, (name, dev)
    return dev


def test_config2_cg_256_cubed_full_history_equals_the_oracle(ctx, monkeypatch):
    """BASELINE config 2 as written: unpreconditioned CG on 256^3 Poisson, tol 1e-8, max 2000 (cg.rs:114-288) -- all ~580 iterations.
    Then the SAME solve through the form bench.py's 512^3 headline takes (round 5: the direction pass inside the staged-window SpMV, x updated
    in batches of 8 iterations from a ring of direction vectors), forced here at 256^3: the same iterations, history and x -- the convergence
    falls inside a batch (581 = 72 x 8 + 5)."""
    O, a, b = _oracle_system("poisson")
    T, V, F = K.reduce_spec()
    ga = K.CsrMatrix.stencil7(N, "poisson", ctx=ctx)
    gb = ga.spmv(ctx.vec(N ** 3).fill(1.0))
    assert np.array_equal(gb.to_host(), b)
    s = K.CgSolver(1e-8, 2000)
    x = ctx.vec(N ** 3)
    st = s.solve(ga, None, gb, x)
    assert st.converged and st.iterations > 300
    ref_t = O.solve("cg", a, b, tol=1e-8, max_iters=2000, rs=O.Reduce.tiled(T, V, F))
    ref_s = O.solve("cg", a, b, tol=1e-8, max_iters=2000, rs=O.Reduce.serial())
    _compare("config 2: CG 256^3 to 1e-8", s.residual_history, st, x.to_host(), ref_t, ref_s, SERIAL_TOL)
    monkeypatch.setenv("KRYST_CG_FUSE_P", "1")
    s2 = K.CgSolver(1e-8, 2000)
    x2 = ctx.vec(N ** 3)
    st2 = s2.solve(ga, None, gb, x2)
    _compare("config 2 through the fused SpMV with x in batches", s2.residual_history, st2, x2.to_host(), ref_t, ref_s, SERIAL_TOL)


def test_config4_jacobi_pcg_256_cubed_hundred_iterations_equal_the_oracle(ctx):
    """Config 4's solver (Jacobi-PCG, pcg.rs:114-222) at the size ONE rank of the 8-way 512^3 partition holds: 100 iterations, tol 0."""
    O, a, b = _oracle_system("poisson")
    T, V, F = K.reduce_spec()
    ga = K.CsrMatrix.stencil7(N, "poisson", ctx=ctx)
    gb = ga.spmv(ctx.vec(N ** 3).fill(1.0))
    s = K.PcgSolver(0.0, 100)
    x = ctx.vec(N ** 3)
    st = s.solve(ga, K.Jacobi().setup(ga), gb, x)
    assert st.iterations == 100
    ref_t = O.solve("pcg", a, b, pc=O.Pc.jacobi(a), tol=0.0, max_iters=100, rs=O.Reduce.tiled(T, V, F))
    ref_s = O.solve("pcg", a, b, pc=O.Pc.jacobi(a), tol=0.0, max_iters=100, rs=O.Reduce.serial())
    _compare("config 4's solver: Jacobi-PCG 256^3, 100 iterations", s.residual_history, st, x.to_host(), ref_t, ref_s, SERIAL_TOL)


def test_config5_bicgstab_256_cubed_hundred_iterations_equal_the_oracle(ctx):
    """Config 5's solver as the reference has it (BiCGStab, pc ignored, bicgstab.rs:69-293) on the 256^3 anisotropic operator: 100
    iterations.  Bit-identical in the library's dot order; against the serial fold BiCGStab's recurrences amplify the association
    rounding, so the 1e-9 * ||r0|| of DESIGN section 2 is asserted on the first 10 entries and the growth over the 100 is printed."""
    O, a, b = _oracle_system("aniso")
    T, V, F = K.reduce_spec()
    ga = K.CsrMatrix.stencil7(N, "aniso", ctx=ctx)
    gb = ga.spmv(ctx.vec(N ** 3).fill(1.0))
    assert np.array_equal(gb.to_host(), b)
    s = K.BiCgStabSolver(0.0, 100)
    x = ctx.vec(N ** 3)
    st = s.solve(ga, None, gb, x)
    ref_t = O.solve("bicgstab", a, b, tol=0.0, max_iters=100, rs=O.Reduce.tiled(T, V, F))
    ref_s = O.solve("bicgstab", a, b, tol=0.0, max_iters=100, rs=O.Reduce.serial())
    # (far from convergence BiCGStab is not contractive: after 100 iterations at tol 0 the two association orders are 7.6e-4 * ||r0|| apart -- measured,
    # printed below -- while the library-order history is bit-identical; the bound is asserted where a bound means something, on the first entries)
    _compare("config 5's solver: BiCGStab 256^3 anisotropic, 100 iterations", s.residual_history, st, x.to_host(), ref_t, ref_s, 1e-9, serial_entries=10)


def test_box_stencil_27_point_80_cubed_ilu0_apply_and_gmres_equal_the_oracle(ctx):
    """SURVEY 8 row f-2 at a size where the box-stencil wavefront solve (tri_box.h) runs as it does in production -- 11 x 10 blocks of
    lines, the loaders' coalesced path, pollers three blocks deep: a 27-point operator with random unsymmetric coefficients on 80 x 72 x 75
    (432 000 rows, 11.4 M entries), true ILU(0) factored on the device; the apply and 30 iterations of right-preconditioned GMRES(10)
    bit for bit against the oracle (ilup.rs:138-167 triangular solves, gmres.rs:216-402)."""
    from oracle import oracle as O
    import scipy.sparse as sp
    O.set_threads(min(len(os.sched_getaffinity(0)), 16))
    rng = np.random.default_rng(80)
    Ni, Nj, Nk = 80, 72, 75
    n = Ni * Nj * Nk
    idx = np.arange(n)
    i, j, k = idx % Ni, (idx // Ni) % Nj, idx // (Ni * Nj)
    rows, cols, vals = [], [], []
    for dk in (-1, 0, 1):
        for dj in (-1, 0, 1):
            for di in (-1, 0, 1):
                if (dk, dj, di) == (0, 0, 0):
                    continue
                ok = (i + di >= 0) & (i + di < Ni) & (j + dj >= 0) & (j + dj < Nj) & (k + dk >= 0) & (k + dk < Nk)
                r = idx[ok]
                rows.append(r); cols.append(r + di + Ni * dj + Ni * Nj * dk); vals.append(-rng.uniform(0.2, 1.0, len(r)))
    rows, cols, vals = np.concatenate(rows), np.concatenate(cols), np.concatenate(vals)
    dsum = np.ones(n)
    np.add.at(dsum, rows, np.abs(vals))
    m = sp.coo_matrix((np.concatenate([vals, dsum]), (np.concatenate([rows, idx]), np.concatenate([cols, idx]))), shape=(n, n)).tocsr()
    m.sort_indices()
    ao = O.Csr(n, n, m.indptr, m.indices, m.data)
    a = K.CsrMatrix.from_csr(n, n, m.indptr, m.indices, m.data, ctx=ctx)
    pc = K.TrueIlu0().setup(a)
    ref = O.Pc.ilu0_true(ao)
    info = pc.ilu_info()
    assert info["form"].startswith("box wavefront") and info["box"] == [Ni, Nj, Nk] and info["streams"] == [13, 13] and info["regular"], info
    for seed in (1, 2):
        r = np.random.default_rng(seed).standard_normal(n)
        ctx.poison_lds()        # the first version of the kernel read LDS it had not written (rows -1, -2 of its ring): right or wrong by what was there
        assert np.array_equal(pc.apply(r), ref.apply(r))
    assert pc.ilu_info()["form"] == info["form"], "the wavefront solve gave up"
    b = ao.spmv(np.ones(n))
    res = O.solve("gmres", ao, b, pc=ref, tol=1e-30, max_iters=30, restart=10, side=O.SIDE_RIGHT, rs=O.Reduce.tiled(*K.reduce_spec()))
    g = K.GmresSolver(10, 1e-30, 30); g.preconditioning = K.Preconditioning.Right
    x = np.zeros(n)
    st = g.solve(a, pc, b, x)
    assert st.iterations == res.iterations and st.final_residual == res.final_residual and np.array_equal(x, res.x)
    print(f"[full-size parity] 27-point {Ni}x{Nj}x{Nk}: true ILU(0) apply and {st.iterations} GMRES(10) iterations bit-identical to the oracle "
          f"(final residual {st.final_residual:.6e})")
