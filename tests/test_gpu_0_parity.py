"""GPU parity tests proper: the HIP path (through the C ABI) against the CPU oracle on the same inputs.

Bar: BIT-EXACT.  SpMV and every pointwise kernel reproduce the reference's operation order; inner products are
compared with the oracle in the library's own association order (kryst_reduce_spec -> oracle REDUCE_TILED), which
is one admissible execution of the reference's Rayon reduce (src/core/wrappers.rs:92-100).  Against the oracle's
strict serial fold (the --no-default-features reference) iteration counts must be equal and residual histories
agree to the tolerance written in each test.
"""
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


def random_csr(rng, nrows, ncols, row_len):
    """row_len: callable -> length of each row (may be 0)."""
    rp = [0]; ci = []; va = []
    for _ in range(nrows):
        k = min(int(row_len()), ncols)
        cols = np.sort(rng.choice(ncols, size=k, replace=False)) if k else np.array([], dtype=np.int64)
        ci.extend(cols.tolist()); va.extend(rng.standard_normal(k).tolist()); rp.append(len(ci))
    return O.Csr(nrows, ncols, rp, ci, va)


def random_csr_fast(rng, nrows, ncols, max_len):
    """rows of 0..max_len distinct random columns (duplicates of the draw removed), vectorised for large matrices."""
    rp = [0]; ci = []; va = []
    lens = rng.integers(0, max_len + 1, size=nrows)
    for k in lens:
        cols = np.unique(rng.integers(0, ncols, size=int(k)))
        ci.append(cols); rp.append(rp[-1] + len(cols))
    ci = np.concatenate(ci) if ci else np.array([], dtype=np.int64)
    return O.Csr(nrows, ncols, rp, ci, rng.standard_normal(len(ci)))


# ------------------------------------------------------------------------------------------------ SpMV
def test_reference_spmv_known_answers(ctx):
    # src/matrix/sparse.rs:121-144
    m = K.CsrMatrix.from_csr(3, 3, [0, 1, 2, 3], [0, 1, 2], [1.0, 1.0, 1.0], ctx=ctx)
    x = np.array([2.0, 3.0, 5.0]); y = np.zeros(3)
    m.spmv(x, y)
    assert np.array_equal(y, x)
    m = K.CsrMatrix.from_csr(2, 3, [0, 2, 4], [0, 1, 1, 2], [1.0, 2.0, 3.0, 4.0], ctx=ctx)
    y = np.zeros(2)
    m.spmv(np.ones(3), y)
    assert np.array_equal(y, [3.0, 7.0])


def test_from_csr_rejects_what_new_checked_rejects(ctx):
    with pytest.raises(K.KError):
        K.CsrMatrix.from_csr(2, 2, [0, 2, 3], [1, 0, 1], [1.0, 2.0, 3.0], ctx=ctx)     # unsorted
    with pytest.raises(K.KError):
        K.CsrMatrix.from_csr(2, 2, [0, 1, 2], [0, 2], [1.0, 2.0], ctx=ctx)             # out of range
    m = K.CsrMatrix.from_csr(2, 3, [0, 2, 4], [0, 1, 1, 2], [1.0, 2.0, 3.0, 4.0], ctx=ctx)
    with pytest.raises(K.KError):
        m.spmv(np.ones(2))                                                              # sparse.rs:57 assert_eq!


@pytest.mark.parametrize("kind", ["poisson", "aniso", "convdiff", "varcoef"])
@pytest.mark.parametrize("N", [1, 2, 7, 16, 33])
def test_spmv_stencil_bit_exact(ctx, kind, N):
    a = O.stencil7(N, kind)
    x = O.splitmix64_uniform(0xC0FFEE, a.ncols) - 0.5
    assert np.array_equal(to_dev(ctx, a).spmv(x), a.spmv(x))


def test_spmv_general_matrices_bit_exact(ctx):
    rng = np.random.default_rng(7)
    cases = [
        random_csr(rng, 1, 1, lambda: 1),
        random_csr(rng, 1000, 777, lambda: rng.integers(0, 12)),                 # ragged, empty rows, non-square
        random_csr(rng, 513, 513, lambda: rng.integers(0, 3)),                   # tile boundary + 1
        random_csr(rng, 600, 9000, lambda: rng.choice([0, 1, 5, 4000, 8000])),   # rows longer than an LDS window
        random_csr(rng, 1536, 1536, lambda: 40),                                 # several windows per tile
        O.Csr.from_dense(rng.standard_normal((70, 70))),                         # the reference's dense test shape
        O.Csr(5, 5, [0, 0, 0, 0, 0, 0], [], []),                                 # empty matrix
    ]
    for a in cases:
        x = rng.standard_normal(a.ncols)
        assert np.array_equal(to_dev(ctx, a).spmv(x), a.spmv(x)), (a.nrows, a.ncols, a.nnz)


@pytest.mark.parametrize("kernel,compress", [("2", "0"), ("3", "0"), ("3", "1"), ("3", "2"), ("3", "3")])
def test_spmv_kernel_forms_bit_exact(ctx, kernel, compress, monkeypatch):
    """The products-in-LDS wave kernel, the rows kernel, the rows kernel with CSR-D8 index compression and the fully
    dictionary-coded CSR-D16 kernel and the row-pattern CSR-P16 kernel must all reproduce the oracle bit for bit (and
    therefore each other), also where compression does not apply."""
    monkeypatch.setenv("KRYST_SPMV_KERNEL", kernel)
    monkeypatch.setenv("KRYST_SPMV_COMPRESS", compress)
    rng = np.random.default_rng(21)
    cases = [O.stencil7(19, "convdiff"), O.stencil7(40, "poisson"),
             random_csr(rng, 1000, 777, lambda: rng.integers(0, 12)),                 # > 256 distinct offsets: plain path
             random_csr(rng, 600, 9000, lambda: rng.choice([0, 1, 5, 4000, 8000])),   # rows longer than a window
             O.Csr.from_dense(rng.standard_normal((70, 70))),                         # 139 offsets: compressible
             O.Csr.from_dense(O.tridiag(1500, -1.0, 2.0, 0.5), keep_zeros=False),
             O.Csr(5, 5, [0, 0, 0, 0, 0, 0], [], [])]
    for a in cases:
        x = rng.standard_normal(a.ncols)
        assert np.array_equal(to_dev(ctx, a).spmv(x), a.spmv(x)), (kernel, compress, a.nrows, a.nnz)
    # fused inner products through a solver, device-generated operator (codes come from the generator kernel)
    a = K.CsrMatrix.stencil7(24, "aniso", ctx=ctx)
    ao = O.stencil7(24, "aniso")
    b = ao.spmv(np.ones(ao.nrows))
    T, V, F = K.reduce_spec()
    res = O.solve("bicgstab", ao, b, tol=1e-7 * np.linalg.norm(b), max_iters=200, rs=O.Reduce.tiled(T, V, F))
    s = K.BiCgStabSolver(1e-7 * np.linalg.norm(b), 200); xx = np.zeros(ao.nrows)
    st = s.solve(a, None, b, xx)
    assert st.iterations == res.iterations and np.array_equal(xx, res.x) and np.array_equal(np.array(s.residual_history), res.history)


def test_spmv_diagonal_streams_form_bit_exact(ctx, monkeypatch):
    """CSR-DIA (one value stream per diagonal, absent entries marked by a NaN payload): built whenever an operator has at most 32
    well-filled diagonals (KRYST_SPMV_DIA=2: also beside the more compact D16 / P16 forms, which this test switches off).  Cases:
    1-, 3-, 5-, 7-, 9-, 11-, 16-, 27- and 32-diagonal operators (the exact-count kernels and the batched one), boxes whose first tiles start
    before x[0] and whose last reach past x's end, stored zeros, signed zeros, inf / NaN values AND inf / NaN in x next to absent
    entries (an absent entry must contribute nothing, not 0 * x), empty rows, a stored value with the marker's own bits (the
    operator must then NOT take this form), 33 diagonals and sparsely filled diagonals (not this form either)."""
    import scipy.sparse as sp
    monkeypatch.setenv("KRYST_SPMV_DIA", "2"); monkeypatch.setenv("KRYST_SPMV_COMPRESS", "1")
    rng = np.random.default_rng(44)

    def banded(n, offs, keep=1.0, vals=None):
        rows, cols, vs = [], [], []
        for o in offs:
            i = np.arange(max(0, -o), min(n, n - o))
            m = rng.random(len(i)) < keep
            rows.append(i[m]); cols.append(i[m] + o); vs.append(rng.standard_normal(int(m.sum())) if vals is None else np.full(int(m.sum()), vals))
        m = sp.csr_matrix((np.concatenate(vs), (np.concatenate(rows), np.concatenate(cols))), shape=(n, n)); m.sort_indices()
        return O.Csr(n, n, m.indptr, m.indices, m.data)

    def box27(N):                                                                                     # 27-point operator on an N^3 box, random values
        one = sp.diags([np.ones(N - 1), np.ones(N), np.ones(N - 1)], [-1, 0, 1])
        m = sp.kron(one, sp.kron(one, one)).tocsr(); m.sort_indices()
        m.data = rng.standard_normal(m.nnz)
        return O.Csr(N ** 3, N ** 3, m.indptr, m.indices, m.data)

    yes = [banded(1, [0]), banded(700, [0]), banded(2000, [-1, 0, 1]), banded(5000, [-70, -1, 0, 1, 70]), O.stencil7(21, "varcoef"), O.stencil7(33, "poisson"),
           banded(3000, [-300, -30, -1, 0, 1, 30, 300], keep=0.97),                               # randomly missing couplings
           banded(4099, [-1200, -35, -34, -1, 0, 1, 34, 35, 1200]), banded(2500, list(range(-5, 6))), banded(3333, list(range(-8, 8))),
           banded(6000, list(range(-13, 14))), banded(5000, list(range(-16, 16))), box27(24)]            # 27 and 32 diagonals; a 27-point box with random coefficients
    for a in yes:
        d = to_dev(ctx, a)
        assert d.encoding()[0] == "csr-dia", (a.nrows, a.nnz)
        for trial in range(2):
            x = rng.standard_normal(a.ncols)
            if trial == 1 and a.ncols > 10:
                x[rng.integers(0, a.ncols, 5)] = [np.inf, -np.inf, np.nan, 0.0, -0.0]             # next to absent entries at the box's faces
            got, want = d.spmv(x), a.spmv(x)
            ok = ~np.isnan(want)
            assert np.array_equal(got, want, equal_nan=True) and np.array_equal(np.signbit(got[ok]), np.signbit(want[ok])), (a.nrows, a.nnz, trial)
    # special values among the stored entries; an explicitly stored zero is an entry (0 * inf = NaN), an absent one is not
    sv = banded(1500, [-40, -1, 0, 1, 40])
    sv.vals[rng.integers(0, sv.nnz, 40)] = rng.choice([0.0, -0.0, np.inf, -np.inf, np.nan, 5e-324, -5e-324], 40)
    d = to_dev(ctx, sv)
    assert d.encoding()[0] == "csr-dia"
    x = rng.standard_normal(1500); x[7] = np.inf; x[900] = np.nan
    assert np.array_equal(d.spmv(x), sv.spmv(x), equal_nan=True)
    # NOT this form: the marker's bit pattern among the values, 17 diagonals, thinly filled diagonals, an empty matrix
    mk = banded(1000, [-1, 0, 1])
    mk.vals[123] = np.frombuffer(np.uint64(0x7FF8D1A0D1A0D1A0).tobytes(), dtype=np.float64)[0]
    for a in (mk, banded(3000, list(range(-16, 17))), banded(4000, [-1000, -10, 0, 10, 1000], keep=0.5), O.Csr(5, 5, [0, 0, 0, 0, 0, 0], [], [])):
        d = to_dev(ctx, a)
        assert d.encoding()[0] != "csr-dia", (a.nrows, a.nnz)
        x = rng.standard_normal(a.ncols)
        assert np.array_equal(d.spmv(x), a.spmv(x), equal_nan=True)
    # fused inner products through a solver on the diagonal-streams form
    ao = O.stencil7(24, "aniso"); a = K.CsrMatrix.stencil7(24, "aniso", ctx=ctx)
    assert a.encoding()[0] == "csr-dia"
    b = ao.spmv(np.ones(ao.nrows))
    T, V, F = K.reduce_spec()
    res = O.solve("bicgstab", ao, b, tol=1e-7 * np.linalg.norm(b), max_iters=200, rs=O.Reduce.tiled(T, V, F))
    s = K.BiCgStabSolver(1e-7 * np.linalg.norm(b), 200); xx = np.zeros(ao.nrows)
    st = s.solve(a, None, b, xx)
    assert st.iterations == res.iterations and np.array_equal(xx, res.x) and np.array_equal(np.array(s.residual_history), res.history)


@pytest.mark.parametrize("compress,dia", [("0", "0"), ("1", "0"), ("1", "2"), ("2", "0"), ("3", "0")])
def test_spmv_slab_order_of_the_tiles_bit_exact(ctx, compress, dia, monkeypatch):
    """The slab order of the tiles (csr_create.hip: build_tile_order -- plane-structured operators hand their tiles to the XCDs segment by
    segment of the plane) changes which workgroup computes a tile, nothing else: y, the fused inner products and whole solves are bit
    for bit those of the natural order and of the oracle, in every storage form (KRYST_SPMV_ORDER=2: all kernels), on cubes, on boxes
    whose planes are not a whole number of tiles (uneven shares: empty slots), with two passes of segments, and on operators that must
    NOT get an order (entries far from 0 and +-b; small planes).  The thresholds are lowered so that 10^5-row cases qualify."""
    import scipy.sparse as sp
    monkeypatch.setenv("KRYST_SPMV_COMPRESS", compress); monkeypatch.setenv("KRYST_SPMV_DIA", dia)
    monkeypatch.setenv("KRYST_SPMV_ORDER", "2"); monkeypatch.setenv("KRYST_SPMV_ORDER_MIN_PLANE", "8192")
    rng = np.random.default_rng(77)

    def box7(ni, nj, nk, values="random"):
        e = lambda n: sp.diags([np.ones(n - 1), np.ones(n), np.ones(n - 1)], [-1, 0, 1])
        pat = (sp.kron(sp.identity(nk), sp.kron(sp.identity(nj), e(ni))) + sp.kron(sp.identity(nk), sp.kron(e(nj), sp.identity(ni))) +
               sp.kron(e(nk), sp.identity(nj * ni))).tocsr()
        pat.sort_indices()
        vals = rng.standard_normal(pat.nnz) if values == "random" else rng.choice([-1.0, 6.0, 0.5], pat.nnz)
        return O.Csr(pat.shape[0], pat.shape[1], pat.indptr, pat.indices, vals)

    cases = [(box7(150, 150, 9), True), (box7(128, 128, 7, "few"), True), (box7(100, 250, 6), True), (box7(40, 40, 40), False)]
    for a, qualifies in cases:
        monkeypatch.setenv("KRYST_SPMV_ORDER", "2")
        d = to_dev(ctx, a)
        info = d.tile_order()
        assert (info["plane_rows"] > 0) == qualifies and info["in_use"] == qualifies, info
        if qualifies:
            ntiles = (a.nrows + 511) // 512
            assert info["slots"] >= ntiles and info["slots8"] >= info["slots"] and info["slots"] % 8 == 0 and info["slots8"] % 64 == 0
        x = rng.standard_normal(a.ncols)
        want = a.spmv(x)
        assert np.array_equal(d.spmv(x), want), (compress, dia, a.nrows)
        monkeypatch.setenv("KRYST_SPMV_ORDER", "0")              # read per launch: the same operator in natural order
        assert not d.tile_order()["in_use"]
        assert np.array_equal(d.spmv(x), want)
        monkeypatch.setenv("KRYST_SPMV_ORDER", "1")              # the default policy: CSR-DIA always, plain CSR from 262144-row planes, the coded forms never
        assert d.tile_order()["in_use"] == (qualifies and d.encoding()[0] == "csr-dia")
    # two passes of segments (16 segments of 2304 rows per plane of 192 x 192: an XCD walks segment x, then segment x + 8), the
    # generator-made operator, fused inner products in a solve
    monkeypatch.setenv("KRYST_SPMV_ORDER", "2"); monkeypatch.setenv("KRYST_SPMV_ORDER_SEG_ROWS", "2560")
    N = 192
    for kind in ("varcoef", "poisson"):
        monkeypatch.setenv("KRYST_SPMV_ORDER", "2")              # (KRYST_SPMV_ORDER=0 at creation: no order is built)
        a = K.CsrMatrix.stencil7(N, kind, ctx=ctx)
        n = a.nrows()
        assert a.tile_order()["plane_rows"] == N * N and n // 512 <= a.tile_order()["slots"] <= 1.03 * n / 512     # even shares
        xs = rng.standard_normal(n)
        monkeypatch.setenv("KRYST_SPMV_ORDER", "2"); y2 = a.spmv(xs)
        monkeypatch.setenv("KRYST_SPMV_ORDER", "0"); y0 = a.spmv(xs)
        assert np.array_equal(y2, y0)
        b = a.spmv(np.ones(n))
        out = []
        for order in ("2", "0"):
            monkeypatch.setenv("KRYST_SPMV_ORDER", order)
            s = K.CgSolver(1e-9, 25); xx = np.zeros(n)
            st = s.solve(a, None, b, xx)
            out.append((st.iterations, tuple(s.residual_history), xx.tobytes()))
        assert out[0] == out[1], kind


def test_spmv_pattern_kernel_with_staged_window_bit_exact(ctx, rs, monkeypatch):
    """spmv_pattern_stage_kernel (CSR-P16 operators whose bases are (far, -n, -1, 0, +1, +n, far) with one even n <= 1024: the near
    operands of a run of 4 tiles come out of an LDS window filled by LDS-DMA loads) against the oracle and against
    spmv_pattern_kernel (KRYST_SPMV_STAGE=0), bit for bit: boxes with lines of 8 ... 1024 points, first / last runs whose window or
    far operands reach outside x (clamped requests, masked operands), a last run of fewer than 4 tiles and a last tile of fewer than
    512 rows, inf / NaN in x next to absent entries (an absent entry must contribute nothing), odd line lengths (NOT this kernel),
    far offsets that differ between bases, the fused inner products of CG and BiCGStab, and the generator-made operator."""
    import scipy.sparse as sp
    rng = np.random.default_rng(123)

    def box7(ni, nj, nk, vals=(6.0, -1.0, -1.5, -0.25)):
        e = lambda n: sp.diags([np.ones(n - 1), np.ones(n - 1)], [-1, 1])
        I = sp.identity
        m = (vals[0] * I(ni * nj * nk) + vals[1] * sp.kron(I(nk), sp.kron(I(nj), e(ni))) + vals[2] * sp.kron(I(nk), sp.kron(e(nj), I(ni))) +
             vals[3] * sp.kron(e(nk), I(nj * ni))).tocsr()
        m.sort_indices()
        return O.Csr(m.shape[0], m.shape[1], m.indptr, m.indices, m.data)

    boxes = [(8, 9, 40), (34, 7, 33), (130, 12, 5), (1024, 3, 3), (40, 40, 40), (64, 64, 9), (9, 16, 16), (1026, 3, 3)]
    for ni, nj, nk in boxes:
        a = box7(ni, nj, nk)
        d = to_dev(ctx, a)
        assert d.encoding()[0] == "csr-p16", (ni, nj, nk)
        monkeypatch.setenv("KRYST_SPMV_STAGE", "1")
        info = d.pattern_info()
        assert info["line"] == (ni if ni % 2 == 0 and 8 <= ni <= 1024 else 0) and info["staged"] == (info["line"] > 0) and info["uniform_far"] == (info["line"] > 0), info
        for trial in range(2):
            x = rng.standard_normal(a.ncols)
            if trial == 1:
                x[rng.integers(0, a.ncols, 6)] = [np.inf, -np.inf, np.nan, 0.0, -0.0, 1e308]
                x[0] = np.nan; x[-1] = np.inf                                                      # next to the absent -1 / +1 entries of the corners
            want = a.spmv(x)
            out = []
            for stage in ("1", "0"):
                monkeypatch.setenv("KRYST_SPMV_STAGE", stage)
                assert d.pattern_info()["staged"] == (stage == "1" and info["line"] > 0)
                out.append(d.spmv(x))
            ok = ~np.isnan(want)
            assert np.array_equal(out[0], want, equal_nan=True) and np.array_equal(out[1], want, equal_nan=True), (ni, nj, nk, trial)
            assert np.array_equal(np.signbit(out[0][ok]), np.signbit(want[ok]))
    monkeypatch.delenv("KRYST_SPMV_STAGE")
    # fused inner products: whole solves on a host-built box (one base, uniform far offsets) and on generator-made operators
    a = box7(34, 20, 11); d = to_dev(ctx, a)
    b = a.spmv(np.ones(a.nrows))
    for name, cls, tol in (("cg", K.CgSolver, 1e-9), ("bicgstab", K.BiCgStabSolver, 1e-7 * np.linalg.norm(b))):
        res = O.solve(name, a, b, tol=tol, max_iters=60, rs=rs)
        s = cls(tol, 60); x = np.zeros(a.nrows)
        st = s.solve(d, None, b, x)
        _check_solver(res, st, s, x)
    for N, kind in ((26, "aniso"), (64, "convdiff")):
        ao = O.stencil7(N, kind); ag = K.CsrMatrix.stencil7(N, kind, ctx=ctx)
        x = rng.standard_normal(ao.nrows)
        assert np.array_equal(ag.spmv(x), ao.spmv(x)), (N, kind)


def test_csr_placement_tries_keep_the_operator_bit_exact(ctx, monkeypatch):
    """KRYST_CSR_PLACEMENT_TRIES (bench_streams.hip: csr_place): the creation copies (row_ptr, col, val) to further homes, times the traffic
    skeleton on each and keeps the fastest -- whichever home wins, the operator is the same: plain-CSR SpMV, download and a CG solve bit for bit,
    through kryst_csr_create and through the device generator."""
    monkeypatch.setenv("KRYST_CSR_PLACEMENT_TRIES", "4")
    monkeypatch.setenv("KRYST_SPMV_COMPRESS", "0")
    rng = np.random.default_rng(77)
    T, V, F = K.reduce_spec()
    rs = O.Reduce.tiled(T, V, F)
    for ao, make in ((O.stencil7(24, "convdiff"), lambda: K.CsrMatrix.stencil7(24, "convdiff", ctx=ctx)),
                     (random_csr_fast(rng, 20000, 20000, 8), None)):
        a = make() if make else to_dev(ctx, ao)
        info = a.placement_info()
        assert info["tries"] == 4 and 0 <= info["chosen"] < 4 and len(info["skeleton_ms"]) == 4 and all(m > 0 for m in info["skeleton_ms"]), info
        x = rng.standard_normal(ao.ncols)
        assert np.array_equal(a.spmv(x), ao.spmv(x))
        rp, ci, va = a.download()
        assert np.array_equal(rp, ao.row_ptr) and np.array_equal(ci, ao.col_idx) and np.array_equal(va, ao.vals)
    ao = O.stencil7(24, "poisson"); a = K.CsrMatrix.stencil7(24, "poisson", ctx=ctx)
    b = ao.spmv(np.ones(ao.nrows))
    res = O.solve("cg", ao, b, tol=1e-9, max_iters=300, rs=rs)
    s = K.CgSolver(1e-9, 300); xx = np.zeros(ao.nrows)
    st = s.solve(a, None, b, xx)
    assert st.iterations == res.iterations and np.array_equal(xx, res.x) and np.array_equal(np.array(s.residual_history), res.history)
    monkeypatch.setenv("KRYST_CSR_PLACEMENT_TRIES", "1")
    assert K.CsrMatrix.stencil7(12, "poisson", ctx=ctx).placement_info()["tries"] == 1


@pytest.mark.parametrize("slots", ["2", "4", "7"])
@pytest.mark.parametrize("nt,align", [("0", "0"), ("1", "0"), ("1", "1")])
def test_spmv_plain_kernel_settings_bit_exact(ctx, slots, nt, align, monkeypatch):
    """spmv_wave_kernel with every window size (the launcher picks 7 or 4 pair slots by vector size), nontemporal stream loads
    and line-aligned windows, on ragged / empty / long rows and through the fused inner products (CG: one quantity,
    BiCGStab: two) -- bit for bit the oracle."""
    monkeypatch.setenv("KRYST_SPMV_KERNEL", "2"); monkeypatch.setenv("KRYST_SPMV_COMPRESS", "0")
    monkeypatch.setenv("KRYST_SPMV_SLOTS", slots); monkeypatch.setenv("KRYST_SPMV_NT", nt); monkeypatch.setenv("KRYST_SPMV_ALIGN", align)
    rng = np.random.default_rng(int(slots) * 100 + int(nt) * 10 + int(align))
    cases = [O.stencil7(19, "convdiff"), O.stencil7(40, "poisson"),                  # 125 tiles: several runs per XCD
             random_csr(rng, 1, 1, lambda: 1),
             random_csr(rng, 1000, 777, lambda: rng.integers(0, 12)),
             random_csr(rng, 513, 513, lambda: rng.integers(0, 3)),                   # tile boundary + 1, many empty rows
             random_csr(rng, 600, 9000, lambda: rng.choice([0, 1, 5, 4000, 8000])),   # rows longer than a window
             random_csr(rng, 5000, 5000, lambda: rng.choice([0, 0, 0, 300])),         # whole 128-row slices without entries
             random_csr(rng, 1536, 1536, lambda: 40),                                 # several windows per slice
             random_csr_fast(rng, 20000, 20000, 8),                                  # 40 tiles, ragged
             O.Csr.from_dense(rng.standard_normal((70, 70))),
             O.Csr(5, 5, [0, 0, 0, 0, 0, 0], [], []),
             O.Csr(1030, 4, [0] * 1031, [], [])]                                       # three tiles, no entry at all
    for a in cases:
        x = rng.standard_normal(a.ncols)
        assert np.array_equal(to_dev(ctx, a).spmv(x), a.spmv(x)), (slots, nt, align, a.nrows, a.nnz)
    T, V, F = K.reduce_spec()
    rs = O.Reduce.tiled(T, V, F)
    for kind, N in (("poisson", 20), ("aniso", 24)):
        ao = O.stencil7(N, kind)
        a = to_dev(ctx, ao)
        b = ao.spmv(np.ones(ao.nrows))
        res = O.solve("cg", ao, b, tol=1e-9, max_iters=300, rs=rs)
        s = K.CgSolver(1e-9, 300); xx = np.zeros(ao.nrows)
        st = s.solve(a, None, b, xx)
        assert st.iterations == res.iterations and np.array_equal(xx, res.x) and np.array_equal(np.array(s.residual_history), res.history)
        tol = 1e-7 * np.linalg.norm(b)
        res = O.solve("bicgstab", ao, b, tol=tol, max_iters=200, rs=rs)
        s = K.BiCgStabSolver(tol, 200); xx = np.zeros(ao.nrows)
        st = s.solve(a, None, b, xx)
        assert st.iterations == res.iterations and np.array_equal(xx, res.x) and np.array_equal(np.array(s.residual_history), res.history)


@pytest.mark.parametrize("level", ["2", "3"])
def test_spmv_value_dictionary_cases(ctx, level, monkeypatch):
    """CSR-D16 (offset AND value dictionary): applies with <= 256 distinct value bit patterns, else falls back.
    CSR-P16 (row patterns) on top: applies to the few-pattern cases below, falls back to D16 / D8 / plain otherwise."""
    monkeypatch.setenv("KRYST_SPMV_COMPRESS", level)
    rng = np.random.default_rng(99)

    def with_values(a, pool):
        v = rng.choice(pool, size=a.nnz)
        return O.Csr(a.nrows, a.ncols, a.row_ptr, a.col_idx, v)

    banded = lambda n, offs: O.Csr.from_dense(sum(np.diag(np.ones(n - abs(o)), o) for o in offs), keep_zeros=False)      # noqa: E731
    special = np.array([0.0, -0.0, 1.0, -1.0, np.inf, -np.inf, np.nan, 5e-324, 1.7976931348623157e308, 0.1, 1 / 3])
    cases = [
        with_values(banded(700, range(-20, 21)), rng.standard_normal(256)),          # exactly 256 distinct values: D16
        with_values(banded(700, range(-20, 21)), rng.standard_normal(257)),          # 257: CSR-D8 (offsets only)
        with_values(banded(1300, [-600, -3, -1, 0, 1, 2, 600]), special),            # signed zeros, inf, nan, denormal
        with_values(random_csr(rng, 1000, 777, lambda: rng.integers(0, 12)), [2.0, -1.0]),    # > 256 offsets: plain CSR
        with_values(O.Csr.from_dense(np.ones((3, 2000))), np.arange(1.0, 100.0)),    # rows longer than a 1016-entry pass
        with_values(O.Csr.from_dense(np.ones((130, 250))), [0.5, 0.25]),             # 250 entries per row, many windows
        with_values(banded(4, [0]), [7.0]),
        banded(5000, [-70, -1, 0, 1, 70]),                                           # 9 row patterns, several tiles
        with_values(banded(3000, [-2, 0, 5]), [1.5, -0.0, np.nan]),                  # 3^3 value patterns x boundary shapes
        O.Csr.from_dense(np.ones((600, 9))),                                         # one pattern per row (offsets differ): > KR_PMAX
        O.Csr.from_dense(np.triu(np.ones((40, 40)))),                                # 40 patterns, 820 table entries
        O.Csr.from_dense(np.triu(np.ones((46, 46)))),                                # 1081 table entries: > KR_TMAX, falls back
        O.Csr(1030, 1030, [0] * 513 + [1] * 518, [512], [3.0]),                       # empty rows, one entry in the second tile
    ]
    for a in cases:
        x = rng.standard_normal(a.ncols)
        got, ref = to_dev(ctx, a).spmv(x), a.spmv(x)
        assert np.array_equal(got, ref, equal_nan=True), (a.nrows, a.ncols, a.nnz)
        assert np.array_equal(np.signbit(got), np.signbit(ref))


def test_spmv_device_generator_matches_host(ctx):
    for kind in ("poisson", "aniso", "convdiff", "varcoef"):
        a = K.CsrMatrix.stencil7(12, kind, ctx=ctx)
        ref = O.stencil7(12, kind)
        rp, ci, va = a.download()
        assert np.array_equal(rp, ref.row_ptr) and np.array_equal(ci, ref.col_idx) and np.array_equal(va, ref.vals)
        hrp, hci, hva = K.host_stencil7(12, kind)
        assert np.array_equal(hrp, ref.row_ptr) and np.array_equal(hci, ref.col_idx) and np.array_equal(hva, ref.vals)


def test_operator_download_of_a_large_operator_returns_what_was_uploaded(ctx):
    """kryst_csr_download at a size where the copies are tens of megabytes (the set-up paths of Ilup / Ilut / the host ILU(0) loop start
    with it): row pointers widened from the device's int32, columns and values byte for byte, empty rows included."""
    threads = 0
    rng = np.random.default_rng(77 + threads)
    n = 1_300_003                                                                   # row pointers: 5.2 MB = one full piece + a ragged one
    i = np.arange(n, dtype=np.int64)
    cols = np.sort((i[:, None] * 3 + np.array([0, 1001, 20011, 300007])[None, :]) % n, axis=1)
    keep = np.arange(4)[None, :] < (i % 5)[:, None]                                 # 0 .. 4 entries per row
    ci = cols[keep].astype(np.int32); row_ptr = np.concatenate(([0], np.cumsum(keep.sum(axis=1))))
    va = rng.standard_normal(len(ci))
    assert len(ci) * 12 > 16 << 20
    d = K.CsrMatrix.from_csr(n, n, row_ptr, ci, va, ctx=ctx)
    rp, gc, gv = d.download()
    assert rp.dtype == np.int64 and np.array_equal(rp, row_ptr) and np.array_equal(gc, ci)
    assert np.array_equal(gv.view(np.uint64), va.view(np.uint64))


@pytest.mark.parametrize("N", [1, 2, 3, 8, 17, 40, 70])
def test_spmv_on_generator_made_operators_bit_exact(ctx, rs, N):
    """Operators written by the device generator (ids, codes and tables come from stencil7_gen_kernel, not from the host builder) run
    the CSR-P16 kernel's generator instances (two gathers for x[row-1 .. row+2], the fused dot's p taken from the diagonal's gather):
    the oracle's bits with and without the fused (p, Ap) partials, boxes smaller than a tile and several tiles large."""
    for kind in ("poisson", "aniso", "convdiff"):
        ao = O.stencil7(N, kind)
        a = K.CsrMatrix.stencil7(N, kind, ctx=ctx)
        assert a.encoding()[0] == "csr-p16"
        x = O.splitmix64_uniform(0xC0FFEE + N, ao.ncols) - 0.5
        assert np.array_equal(a.spmv(x), ao.spmv(x)), (kind, N)
        b = ao.spmv(np.ones(ao.nrows))
        ref = O.solve("cg", ao, b, tol=1e-10, max_iters=12, rs=rs)                 # the fused-dot instances (CENTER = 3)
        s = K.CgSolver(1e-10, 12); xx = np.zeros(ao.nrows)
        st = s.solve(a, None, b, xx)
        assert st.iterations == ref.iterations and np.array_equal(xx, ref.x) and np.array_equal(np.array(s.residual_history), ref.history), (kind, N)


@pytest.mark.parametrize("N", [5, 21, 40])
def test_variable_coefficient_operator_every_form_and_solver(ctx, rs, N, monkeypatch):
    """The variable-coefficient 7-point operator (kind "varcoef": no two rows alike, so neither the value dictionary nor the row
    patterns apply): its most compact form is CSR-DIA (seven value streams); CSR-DIA, CSR-D8 and plain CSR SpMV, CG, Jacobi-PCG and
    BiCGStab + true ILU(0) (device-side factorisation, wavefront solve without repeating coefficient chunks) give the oracle's bits."""
    ao = O.stencil7(N, "varcoef")
    a = K.CsrMatrix.stencil7(N, "varcoef", ctx=ctx)
    assert a.encoding()[:2] == ("csr-dia", 7)
    x = O.splitmix64_uniform(0xC0FFEE, ao.ncols) - 0.5
    want = ao.spmv(x)
    for comp, dia, name in (("3", "1", "csr-dia"), ("1", "1", "csr-dia"), ("1", "0", "csr-d8"), ("0", "1", "csr")):
        monkeypatch.setenv("KRYST_SPMV_COMPRESS", comp); monkeypatch.setenv("KRYST_SPMV_DIA", dia)
        assert a.encoding()[0] == name
        assert np.array_equal(a.spmv(x), want), (comp, dia)
        h = to_dev(ctx, ao)                                                    # host-built operator: same encodings found by kryst_csr_create
        assert np.array_equal(h.spmv(x), want), (comp, dia)
        assert h.encoding()[0] == name or (ao.nrows <= 512 and comp == "3"), (comp, dia)     # (<= 512 rows: every row is its own CSR-P16 pattern)
    monkeypatch.delenv("KRYST_SPMV_COMPRESS"); monkeypatch.delenv("KRYST_SPMV_DIA")
    b = ao.spmv(np.ones(ao.nrows))
    atol = 1e-9 * float(np.linalg.norm(b))                                     # BiCGStab's tolerance is absolute (bicgstab.rs)
    for name, mk, pc_k, pc_o, kw in (("cg", lambda: K.CgSolver(1e-9, 400), None, None, dict(tol=1e-9, max_iters=400)),
                                     ("pcg", lambda: K.PcgSolver(1e-9, 400), lambda: K.Jacobi().setup(a), O.Pc.jacobi, dict(tol=1e-9, max_iters=400)),
                                     ("bicgstab_rpc", lambda: K.BiCgStabRightPcSolver(atol, 200), lambda: K.TrueIlu0().setup(a), O.Pc.ilu0_true,
                                      dict(tol=atol, max_iters=200))):
        ref = O.solve(name, ao, b, pc=pc_o(ao) if pc_o else None, rs=rs, **kw)
        s = mk(); xx = np.zeros(ao.nrows)
        st = s.solve(a, pc_k() if pc_k else None, b, xx)
        assert (st.iterations, st.converged) == (ref.iterations, ref.converged), name
        assert np.array_equal(np.array(s.residual_history), ref.history) and np.array_equal(xx, ref.x), name
    r = O.splitmix64_uniform(5, ao.nrows) - 0.5
    for kpc, ofn in ((K.TrueIlu0, O.Pc.ilu0_true), (K.Ilu0, O.Pc.ilu0_compat), (lambda: K.Ilup(0), O.Pc.ilup0)):
        assert np.array_equal(kpc().setup(a).apply(r), ofn(ao).apply(r))


# ------------------------------------------------------------------------------------------------ BLAS-1
@pytest.mark.parametrize("n", [1, 2, 3, 511, 512, 513, 1024, 100003, 1 << 20])
def test_dot_norm_bit_exact_in_library_order(ctx, rs, n):
    rng = np.random.default_rng(n)
    x, y = rng.standard_normal(n), rng.standard_normal(n)
    dx, dy = ctx.vec(x), ctx.vec(y)
    assert K.dot(dx, dy) == O.dot(x, y, rs)
    assert K.norm(dx) == O.norm(x, rs)
    # against the strict serial fold: rounding-level agreement (tolerance: 1e-12 relative to sum |x_i y_i|)
    assert abs(K.dot(dx, dy) - O.dot(x, y)) <= 1e-12 * np.abs(x * y).sum()


def test_reference_dot_and_norm_known_answers(ctx):
    # tests/core_dense.rs:37-47
    x, y = ctx.vec([1.0, 2.0, 3.0]), ctx.vec([4.0, -5.0, 6.0])
    assert abs(K.dot(x, y) - 12.0) < 1e-12 and abs(K.norm(x) - np.sqrt(14.0)) < 1e-12
    with pytest.raises(K.KError):
        K.dot(x, ctx.vec(4))                                                            # wrappers.rs:91 assert_eq!


def test_pointwise_updates_bit_exact(ctx):
    rng = np.random.default_rng(3)
    n = 70001
    x, y = rng.standard_normal(n), rng.standard_normal(n)
    al = 0.3712345678901234
    dx, dy = ctx.vec(x), ctx.vec(y)
    K.axpy(al, dx, dy)
    assert np.array_equal(dy.to_host(), y + al * x)            # cg.rs:208 `*xj + alpha * pj`
    dy.upload(y)
    K.aypx(al, dx, dy)
    assert np.array_equal(dy.to_host(), x + al * y)            # cg.rs:275 `rj + beta * *pj`
    v = ctx.vec(n).fill_splitmix(0x5EED, 10)
    assert np.array_equal(v.to_host(), O.splitmix64_uniform(0x5EED, n + 10)[10:])


# ------------------------------------------------------------------------------------------------ preconditioners
def test_jacobi_setup_apply_bit_exact(ctx):
    a = O.stencil7(9, "convdiff")
    d = to_dev(ctx, a)
    r = O.splitmix64_uniform(1, a.nrows)
    assert np.array_equal(K.Jacobi().setup(d).apply(r), O.Pc.jacobi(a).apply(r))
    # zero diagonal -> inv 0 (jacobi.rs:69-71); missing diagonal likewise
    a2 = O.Csr(3, 3, [0, 1, 2, 4], [1, 1, 0, 2], [2.0, 0.0, 1.0, 4.0])
    assert np.array_equal(K.Jacobi().setup(to_dev(ctx, a2)).apply([1.0, 1.0, 1.0]), [0.0, 0.0, 0.25])


@pytest.mark.parametrize("m", [0, 1, 2, 5])
def test_apply_chebyshev_bit_exact(ctx, m):
    a = O.stencil7(10)
    r = O.splitmix64_uniform(2, a.nrows)
    z = np.zeros(a.nrows)
    K.apply_chebyshev(to_dev(ctx, a), r, z, 0.2, 11.9, m)
    assert np.array_equal(z, O.apply_chebyshev(a, r, 0.2, 11.9, m))
    z2 = np.zeros(a.nrows)
    K.apply_chebyshev(to_dev(ctx, a), r, z2, 1.0, 1.0, 3)       # degenerate interval copies r (chebyshev.rs:88-92)
    assert np.array_equal(z2, r)


def test_chebyshev_trait_apply_is_a_stub(ctx):
    a = to_dev(ctx, O.stencil7(4))
    with pytest.raises(K.KError) as e:
        K.Chebyshev(3, 0.1, 12.0).setup(a).apply(np.ones(64))
    assert e.value.code == 2                                    # KError::SolveError, chebyshev.rs:68-70


ILU_MODES = [("compat", K.Ilu0, O.Pc.ilu0_compat), ("ilup0", K.Ilup, O.Pc.ilup0), ("true", K.TrueIlu0, O.Pc.ilu0_true)]


@pytest.mark.parametrize("mode", [0, 1, 2])
def test_ilu_apply_bit_exact(ctx, mode):
    """Level-scheduled triangular solve == the reference's row-sequential loops (ilu.rs:105-122, ilup.rs:138-167)."""
    _, kcls, ofn = ILU_MODES[mode]
    rng = np.random.default_rng(11 + mode)
    mats = [O.stencil7(5, "convdiff"), O.stencil7(17, "aniso"), O.stencil7(30, "poisson"),
            O.Csr.from_dense(O.tridiag(10, -1.0, 2.0, 0.5), keep_zeros=False)]
    d = rng.random((40, 40)) * (rng.random((40, 40)) < 0.2) + np.diag(3.0 + rng.random(40))
    mats.append(O.Csr.from_dense(d, keep_zeros=False))
    mats.append(O.Csr.from_dense(d, keep_zeros=True))                   # stored zeros are skipped (`!= T::zero()`)
    for a in mats:
        r = rng.standard_normal(a.nrows)
        z = kcls().setup(to_dev(ctx, a)).apply(r)
        assert np.array_equal(z, ofn(a).apply(r)), (mode, a.nrows)


@pytest.mark.parametrize("syncfree", ["0", "1"])
def test_triangular_solve_forms_bit_exact(ctx, syncfree, monkeypatch):
    """One launch per dependency level (KRYST_ILU_SYNCFREE=0) and the sync-free single launch (default) give the oracle's bits,
    for ELL factors (<= 4 entries per row), CSR factors (longer rows), tiny levels sharing a wave, and a several-tile grid."""
    monkeypatch.setenv("KRYST_ILU_SYNCFREE", syncfree)
    rng = np.random.default_rng(77)
    dense = rng.random((300, 300)) * (rng.random((300, 300)) < 0.05) + np.diag(5.0 + rng.random(300))
    cases = [(K.TrueIlu0(), O.Pc.ilu0_true, O.stencil7(24, "aniso")),                           # ELL, 70 levels, 13 824 rows
             (K.Ilu0(), O.Pc.ilu0_compat, O.Csr.from_dense(dense, keep_zeros=False)),           # CSR rows, irregular levels
             (K.Ilup(2), lambda a: O.Pc.ilup(a, 2), O.stencil7(9, "convdiff")),                 # fill-in: long rows
             (K.Ilup(0), O.Pc.ilup0, O.Csr.from_dense(O.tridiag(700, -1.0, 2.5, -0.5), keep_zeros=False))]   # 700 levels of one row
    for kpc, ofn, a in cases:
        pc = kpc.setup(to_dev(ctx, a)); ref = ofn(a)
        for seed in (1, 2):
            r = O.splitmix64_uniform(seed, a.nrows) - 0.5
            assert np.array_equal(pc.apply(r), ref.apply(r)), (syncfree, a.nrows)


def test_measurement_hooks_of_round_4_leave_the_operator_alone(ctx):
    """kryst_bench_csr_skeleton streams the operator's own CSR arrays (bench.py: roofline_csr.stream_skeleton) and writes garbage into y only;
    kryst_csr_halo_mode on an operator that is not row-partitioned is a no-op that reports the RCCL mode; kryst_device_count sees the GPU."""
    a = O.stencil7(12, "varcoef")
    d = to_dev(ctx, a)
    x = ctx.vec(a.nrows).fill_splitmix(11)
    before = d.spmv(x).to_host()
    y = ctx.vec(a.nrows)
    ms = d.bench_csr_skeleton(x, y, reps=3)
    assert ms > 0.0
    assert np.array_equal(d.spmv(x).to_host(), before) and np.array_equal(before, a.spmv(x.to_host()))
    assert d.halo_mode("peer") == "rccl" and d.halo_mode("rccl") == "rccl"
    from kryst_amd._ffi import device_count
    assert device_count() >= 1


@pytest.mark.parametrize("pipe", ["free8", "free4", "free1", "free8-wide-first-level", "free8-wide-level-in-the-middle", "1", "0"])
def test_narrow_level_runs_of_a_deep_factor_bit_exact(ctx, pipe, monkeypatch):
    """The one-workgroup run kernels for deep, narrow factors (tri_run_free_kernel with 16 / 4 / 1 waves -- no barriers, values handed over
    through an LDS ring of the last 4 096 positions, older ones gathered from the vector --, tri_run_pipe_kernel / tri_run_kernel, KRYST_ILU_SYNCFREE=0) on a factor of
    40 000 rows and a thousand levels: dependencies in the previous level and tens of thousands of positions back, rows longer than the eight
    entries held in registers, a level wider than the workgroup (1 500 independent rows) -- the oracle's bits.  (Round 4 rebuilt the
    pipelined kernel three ways against this test -- LDS window, lazily waited stores, counted look-ahead, a prefetching workgroup: all
    bit-exact, none faster, DESIGN.md section 8 -- and kept the round-3 kernel.)"""
    import scipy.sparse as sp
    monkeypatch.setenv("KRYST_ILU_SYNCFREE", "0")
    wide = pipe.endswith("-wide-first-level")           # 3 000 independent rows: a level kernel of its own, the run starts behind it (operands "before the run")
    if pipe.startswith("free"):
        monkeypatch.setenv("KRYST_ILU_RUN_FREE", "1"); monkeypatch.setenv("KRYST_ILU_FREE_WAVES", pipe[4])
    else:
        monkeypatch.setenv("KRYST_ILU_RUN_FREE", "0"); monkeypatch.setenv("KRYST_ILU_RUN_PIPE", pipe)
    rng = np.random.default_rng(2024)
    n, free = 40000, (3000 if wide else 1500)
    rows = np.repeat(np.arange(free, n), 9)
    near = rows + rng.integers(-300, 301, len(rows))
    far = rng.integers(0, n, len(rows))                                   # one entry in ten couples to ANY row: dependencies far below the window
    cols = np.clip(np.where(rng.random(len(rows)) < 0.1, far, near), 0, n - 1)
    vals = rng.uniform(-1.0, 1.0, len(rows))
    chain = n
    if pipe.endswith("-wide-level-in-the-middle"):
        # 2 500 more rows that all hang on ONE row half way down the chain: a level of > 2 048 rows between two runs of narrow levels
        # (a level kernel of its own; the second run's first chunks take their operands "from before the run")
        rows = np.concatenate([rows, np.arange(n, n + 2500)]); cols = np.concatenate([cols, np.full(2500, 20000)]); vals = np.concatenate([vals, rng.uniform(-1.0, 1.0, 2500)])
        n += 2500
    m = sp.csr_matrix((vals, (rows, cols)), shape=(n, n)); m.sum_duplicates()
    dense_rows = rng.choice(np.arange(free, chain), 50, replace=False)    # rows with ~30 entries: longer than the held eight
    extra = sp.csr_matrix((rng.uniform(-1.0, 1.0, 50 * 24), (np.repeat(dense_rows, 24), np.clip(np.repeat(dense_rows, 24) + rng.integers(-2000, 2001, 50 * 24), 0, chain - 1))), shape=(n, n))
    m = (m + extra).tocsr(); m.sum_duplicates()
    m = m - sp.diags(m.diagonal()) + sp.diags(np.asarray(abs(m).sum(axis=1)).ravel() + 1.0)
    m = m.tocsr(); m.sort_indices(); m.eliminate_zeros()
    a = O.Csr(n, n, m.indptr, m.indices, m.data)
    d = to_dev(ctx, a)
    for kpc, ofn in ((K.TrueIlu0(), O.Pc.ilu0_true), (K.Ilu0(), O.Pc.ilu0_compat)):
        pc = kpc.setup(d); ref = ofn(a)
        info = pc.ilu_info()
        assert info["form"].startswith("level") and min(info["levels"]) > 200, info
        for seed in (1, 2):
            r = O.splitmix64_uniform(seed, n) - 0.5
            assert np.array_equal(pc.apply(r), ref.apply(r)), (pipe, seed)


@pytest.mark.parametrize("tune", ["16897", "513", "8705"])
def test_chains_of_virtual_rows_with_waves_that_could_run_ahead_bit_exact(ctx, tune, monkeypatch):
    """tri_run_free_kernel on factors whose rows are all longer than its eight operands (a 27-point operator on a 41 x 30 x 19 box through the
    level-ordered forms: 13 entries per row = chains of two virtual rows, the partial sum handed over through the LDS ring) and whose
    dependency cones are narrow: a wave whose rows need nothing of a slow wave's chunk can get far ahead.  Before the waves were held within
    eight chunks of each other one of them came round the ring and overwrote a partial sum that was still waited for (NaNs once the poll
    budget ran out) -- with the gated loop (tune 513) every time, with the default now and then.  All three forms of the loop."""
    import scipy.sparse as sp
    monkeypatch.setenv("KRYST_ILU_BOX", "0"); monkeypatch.setenv("KRYST_ILU_FREE_TUNE", tune)
    t = lambda n: sp.diags([np.ones(n - 1), np.ones(n), np.ones(n - 1)], [-1, 0, 1])
    rng = np.random.default_rng(3)
    m = (sp.identity(41 * 30 * 19) * 28.0 - sp.kron(t(19), sp.kron(t(30), t(41)))).tocsr(); m.sort_indices()
    m.data = m.data * rng.uniform(0.5, 1.5, len(m.data))
    n = m.shape[0]
    a = O.Csr(n, n, m.indptr, m.indices, m.data)
    pc = K.TrueIlu0().setup(to_dev(ctx, a)); ref = O.Pc.ilu0_true(a)
    assert pc.ilu_info()["form"].startswith("level"), pc.ilu_info()
    for seed in (5, 6):
        r = np.random.default_rng(seed).standard_normal(n)
        assert np.array_equal(pc.apply(r), ref.apply(r)), (tune, seed)


@pytest.mark.parametrize("grid_path", ["quad", "1", "wave0", "0"])
def test_structured_grid_triangular_solve_bit_exact(ctx, grid_path, monkeypatch):
    """Factors of 7-point / 5-point operators on an Ni x Nj x Nk box are solved by the pipelined wavefront kernel
    (KRYST_ILU_GRID=1, default): boxes whose sides are not multiples of the 8 x 8 line block, thin and 2-D boxes, and banded
    matrices that LOOK like a grid but wrap around line ends (must be recognised and take the general path).  Both paths
    give the oracle's bits."""
    import scipy.sparse as sp
    # "quad": 16 x 16 lines per workgroup (tri_quad.h, the default); "1": the 8 x 8 three-wave kernel; "wave0": its one-wave
    # predecessor (KRYST_ILU_WAVE=0); "0": level-ordered forms
    monkeypatch.setenv("KRYST_ILU_GRID", "0" if grid_path == "0" else "1")
    monkeypatch.setenv("KRYST_ILU_WAVE", {"wave0": "0", "1": "1"}.get(grid_path, "2"))
    rng = np.random.default_rng(5)

    def box(Ni, Nj, Nk, wrap=False):
        def lap(n, w):
            return sp.diags([-w * np.ones(n - 1), 2 * w * np.ones(n), -0.5 * w * np.ones(n - 1)], [-1, 0, 1])
        if wrap:                                          # plain bands: entries cross line ends -> not a grid operator
            n = Ni * Nj * Nk
            offs = [o for o in (-Ni * Nj, -Ni, -1, 0, 1, Ni, Ni * Nj) if abs(o) < n]
            m = sp.diags([(-1.0 if o else 7.0) * np.ones(n - abs(o)) for o in offs], offs).tocsr()
        else:
            m = (sp.kron(sp.eye(Nk), sp.kron(sp.eye(Nj), lap(Ni, 1.0))) + sp.kron(sp.eye(Nk), sp.kron(lap(Nj, 0.7), sp.eye(Ni)))
                 + sp.kron(lap(Nk, 0.3), sp.kron(sp.eye(Nj), sp.eye(Ni)))).tocsr()
        m.sort_indices(); m.eliminate_zeros()
        return O.Csr(m.shape[0], m.shape[1], m.indptr, m.indices, m.data)

    cases = [box(5, 3, 2), box(9, 8, 8), box(16, 17, 9), box(40, 9, 20), box(7, 20, 1), box(64, 1, 1), box(6, 5, 4, wrap=True),
             box(75, 24, 17), box(23, 16, 16),          # long lines in full blocks: the predicate-free interior chunks
             box(1, 9, 9), box(2, 17, 9), box(3, 8, 16),   # lines shorter than the poller's two-row window
             box(300, 8, 16),                           # the neighbour ring and the LDS stage wrap many times
             box(40, 32, 16), box(37, 16, 32), box(66, 16, 16), box(33, 48, 33)]   # whole 16 x 16 blocks: the 64-byte result groups
                                                        # of tri_quad.h (four lanes per line), also with Ni not a multiple of 8
    for a in cases:
        d = to_dev(ctx, a)
        for kpc, ofn in ((K.TrueIlu0(), O.Pc.ilu0_true), (K.Ilu0(), O.Pc.ilu0_compat), (K.Ilup(0), O.Pc.ilup0)):
            pc = kpc.setup(d); ref = ofn(a)
            for _ in range(2):
                r = rng.standard_normal(a.nrows)
                assert np.array_equal(pc.apply(r), ref.apply(r)), (grid_path, a.nrows)


def _box_operator(rng, Ni, Nj, Nk, keep, drop=0.0, unsym=True, zeros=0.0):
    """A stencil operator inside the 3 x 3 x 3 cube on an Ni x Nj x Nk box, natural ordering: `keep(dk, dj, di)` selects the couplings (27-point: all),
    a fraction `drop` of the couplings is removed at random, the values are random (unsymmetric), the diagonal dominates."""
    import scipy.sparse as sp
    n = Ni * Nj * Nk
    idx = np.arange(n)
    i, j, k = idx % Ni, (idx // Ni) % Nj, idx // (Ni * Nj)
    rows, cols, vals = [], [], []
    for dk in (-1, 0, 1):
        for dj in (-1, 0, 1):
            for di in (-1, 0, 1):
                if (dk, dj, di) == (0, 0, 0) or not keep(dk, dj, di):
                    continue
                ok = (i + di >= 0) & (i + di < Ni) & (j + dj >= 0) & (j + dj < Nj) & (k + dk >= 0) & (k + dk < Nk)
                if drop > 0.0:
                    ok &= rng.random(n) >= drop
                r = idx[ok]
                rows.append(r); cols.append(r + di + Ni * dj + Ni * Nj * dk)
                v = -rng.uniform(0.2, 1.0, len(r)) if unsym else -np.ones(len(r))
                if zeros > 0.0:
                    v[rng.random(len(v)) < zeros] = 0.0                    # stored zeros: part of the pattern, never kept in a factor
                vals.append(v)
    rows, cols, vals = np.concatenate(rows), np.concatenate(cols), np.concatenate(vals)
    dsum = np.ones(n)
    np.add.at(dsum, rows, np.abs(vals))
    m = sp.coo_matrix((np.concatenate([vals, dsum]), (np.concatenate([rows, idx]), np.concatenate([cols, idx]))), shape=(n, n)).tocsr()
    m.sort_indices()
    assert zeros == 0.0 or (m.data == 0.0).any()                           # the zeros are stored
    return O.Csr(n, n, m.indptr, m.indices, m.data)


@pytest.mark.parametrize("form", ["wave", "box", "levels"])
def test_box_stencil_triangular_solve_bit_exact(ctx, form, monkeypatch):
    """Factors of operators inside the 3 x 3 x 3 cube that are NOT 7-point operators (27-point, 19-point, 2-D 9-point, boxes with randomly
    missing couplings) are laid out as 13 natural-order coefficient streams per factor and solved by the box kernels (KRYST_ILU_BOX=1,
    default; round 4) -- hyperplanes i + 2 j + 4 k -- instead of the level-ordered forms: the oracle's bits either way, for true ILU(0),
    Ilu0 as written and Ilup(0), on boxes of any shape; a band matrix that wraps around line ends is not a box operator."""
    import scipy.sparse as sp
    monkeypatch.setenv("KRYST_ILU_BOX", {"wave": "2", "box": "1", "levels": "0"}[form])
    rng = np.random.default_rng(27)
    all27 = lambda dk, dj, di: True
    p19 = lambda dk, dj, di: abs(dk) + abs(dj) + abs(di) <= 2
    p9_2d = lambda dk, dj, di: dk == 0
    cases = [(_box_operator(rng, 5, 4, 3, all27), True), (_box_operator(rng, 9, 8, 8, all27), True), (_box_operator(rng, 17, 9, 10, all27), True),
             (_box_operator(rng, 12, 11, 7, p19), True), (_box_operator(rng, 40, 25, 1, p9_2d), True), (_box_operator(rng, 3, 3, 3, all27), True),
             (_box_operator(rng, 10, 9, 9, all27, drop=0.3), True), (_box_operator(rng, 33, 5, 6, all27, drop=0.05), True),
             (_box_operator(rng, 41, 30, 19, all27), True), (_box_operator(rng, 7, 26, 17, p19, drop=0.1), True),
             (_box_operator(rng, 13, 12, 11, all27, drop=0.1, zeros=0.1), True)]
    # wrap-around bands with the offsets of a 6 x 5 x 4 box: entries cross line ends -> level-ordered forms
    n = 6 * 5 * 4
    offs = sorted({di + 6 * dj + 30 * dk for dk in (-1, 0, 1) for dj in (-1, 0, 1) for di in (-1, 0, 1)})
    m = sp.diags([(-0.03 if o else 5.0) * np.ones(n - abs(o)) for o in offs], offs).tocsr(); m.sort_indices()
    cases.append((O.Csr(n, n, m.indptr, m.indices, m.data), False))
    for case, (a, is_box) in enumerate(cases):
        d = to_dev(ctx, a)
        # streams written on the device from the factor values (gen_box_fill_kernel) / by the host path (build_box)
        monkeypatch.setenv("KRYST_ILU_DEVICE_SETUP", "0" if case % 2 else "1")
        for kpc, ofn in ((K.TrueIlu0(), O.Pc.ilu0_true), (K.Ilu0(), O.Pc.ilu0_compat), (K.Ilup(0), O.Pc.ilup0), (K.Ilup(1), lambda m: O.Pc.ilup(m, 1))):
            pc = kpc.setup(d); ref = ofn(a)
            info = pc.ilu_info()
            if isinstance(kpc, K.Ilup) and kpc.fill == 1:                  # fill entries leave the 3 x 3 x 3 cube
                assert np.array_equal(pc.apply(np.ones(a.nrows)), ref.apply(np.ones(a.nrows)))
                continue
            assert info["form"].startswith("box") == (is_box and form != "levels"), (form, a.nrows, info)
            if is_box and form != "levels":
                assert info["form"].startswith("box wavefront") == (form == "wave"), (form, info)
                # streams without any entry are neither streamed nor subtracted; a factor with entries missing INSIDE the box is not "regular"
                if case == 1 and isinstance(kpc, K.TrueIlu0):
                    assert info["streams"] == [13, 13] and info["regular"], info
                if case == 3 and isinstance(kpc, K.TrueIlu0):
                    assert info["streams"] == [9, 9] and info["regular"], info          # 19-point: no corner couplings
                if case == 6:
                    assert not info["regular"], info                                    # 30 % of the couplings dropped at random
            for rep in range(2):
                r = rng.standard_normal(a.nrows)
                if rep == 0:
                    ctx.poison_lds()                               # (LDS is not cleared between kernels: NaNs there must not reach a result)
                assert np.array_equal(pc.apply(r), ref.apply(r)), (form, a.nrows)
            assert pc.ilu_info()["form"] == info["form"], (form, a.nrows, "the wavefront solve gave up and fell back")


def test_lines_longer_than_the_chunk_flag_table(ctx, monkeypatch):
    """A box whose lines have more 8-step chunks than the 16 x 16 kernel's flag table holds (TQ_SKIPMAX = 544 chunks, Ni > ~4 330):
    the kernel runs without repeat flags and must not request coefficient chunks past the line's end (needs(m) read
    the table's last entry for every m beyond it, so the last block fetched up to four chunks beyond its coefficient array)."""
    import scipy.sparse as sp
    monkeypatch.setenv("KRYST_ILU_GRID", "1"); monkeypatch.setenv("KRYST_ILU_WAVE", "2")
    Ni, Nj, Nk = 4400, 32, 32

    def lap(n, w):
        return sp.diags([-w * np.ones(n - 1), 2 * w * np.ones(n), -0.5 * w * np.ones(n - 1)], [-1, 0, 1])
    m = (sp.kron(sp.eye(Nk), sp.kron(sp.eye(Nj), lap(Ni, 1.0))) + sp.kron(sp.eye(Nk), sp.kron(lap(Nj, 0.7), sp.eye(Ni)))
         + sp.kron(lap(Nk, 0.3), sp.kron(sp.eye(Nj), sp.eye(Ni)))).tocsr()
    m.sort_indices(); m.eliminate_zeros()
    a = O.Csr(m.shape[0], m.shape[1], m.indptr, m.indices, m.data)
    d = to_dev(ctx, a)
    r = np.random.default_rng(9).standard_normal(a.nrows)
    for kpc, ofn in ((K.TrueIlu0(), O.Pc.ilu0_true), (K.Ilu0(), O.Pc.ilu0_compat)):
        assert np.array_equal(kpc.setup(d).apply(r), ofn(a).apply(r))


def test_repeated_coefficient_chunks_are_skipped_bit_exact(ctx, monkeypatch, capfd):
    """tri_quad.h does not request a coefficient chunk that repeats chunk - 3's bits (flags from setup).  On a constant-coefficient
    box most chunks carry the flag (reported by KRYST_ILU_VERBOSE); on a box with random coefficients none does; the solve gives the
    oracle's bits either way and the same bits as with KRYST_ILU_DEDUP=0."""
    import re
    import scipy.sparse as sp
    monkeypatch.setenv("KRYST_ILU_WAVE", "2"); monkeypatch.setenv("KRYST_ILU_GRID", "1"); monkeypatch.setenv("KRYST_ILU_VERBOSE", "1")
    rng = np.random.default_rng(11)
    a = O.stencil7(96, "aniso")                                            # 96^3: 6 x 6 blocks of 16 x 16 lines, 14 chunks each (the factor settles ~15 rows in)
    d = to_dev(ctx, a)
    ref = O.Pc.ilu0_true(a)
    outs = {}
    for dedup in ("1", "0"):
        monkeypatch.setenv("KRYST_ILU_DEDUP", dedup)
        capfd.readouterr()
        pc = K.TrueIlu0().setup(d)
        err = capfd.readouterr().err
        counts = [int(m.group(1)) for m in re.finditer(r"(\d+) of \d+ coefficient chunks repeat", err)]
        if dedup == "1":
            assert len(counts) == 2 and min(counts) > 0, err               # both factors have repeating chunks away from the low faces
        else:
            assert counts == [], err
        r = rng.standard_normal(a.nrows)
        rng = np.random.default_rng(11)                                    # the same right-hand side for both settings
        outs[dedup] = pc.apply(r)
        assert np.array_equal(outs[dedup], ref.apply(r)), dedup
    assert np.array_equal(outs["1"], outs["0"])
    # random coefficients: nothing repeats
    monkeypatch.setenv("KRYST_ILU_DEDUP", "1")
    n1 = 32
    idx = np.arange(n1 ** 3).reshape(n1, n1, n1)
    rows, cols, vals = [np.arange(n1 ** 3)], [np.arange(n1 ** 3)], [rng.uniform(6.5, 8.0, n1 ** 3)]
    for lo, hi in ((idx[:, :, :-1], idx[:, :, 1:]), (idx[:, :-1, :], idx[:, 1:, :]), (idx[:-1, :, :], idx[1:, :, :])):
        for a_, b_ in ((lo.ravel(), hi.ravel()), (hi.ravel(), lo.ravel())):
            rows.append(a_); cols.append(b_); vals.append(rng.uniform(-1.0, 1.0, len(a_)))
    m = sp.csr_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(n1 ** 3, n1 ** 3)); m.sort_indices()
    b = O.Csr(n1 ** 3, n1 ** 3, m.indptr, m.indices, m.data)
    capfd.readouterr()
    pc = K.TrueIlu0().setup(to_dev(ctx, b))
    err = capfd.readouterr().err
    counts = [int(mm.group(1)) for mm in re.finditer(r"(\d+) of \d+ coefficient chunks repeat", err)]
    assert counts == [0, 0], err
    r = rng.standard_normal(b.nrows)
    assert np.array_equal(pc.apply(r), O.Pc.ilu0_true(b).apply(r))


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_structured_grid_solve_random_boxes_and_missing_entries(ctx, seed, monkeypatch):
    """Random boxes, random coefficients and randomly MISSING couplings (an entry that is absent must contribute nothing, not
    0 * neighbour): the wavefront kernels agree with the oracle bit for bit for every ILU flavour."""
    import scipy.sparse as sp
    rng = np.random.default_rng(100 + seed)
    for _ in range(6):
        Ni, Nj, Nk = (int(v) for v in rng.integers(2, 41, 3))
        n = Ni * Nj * Nk
        idx = np.arange(n).reshape(Nk, Nj, Ni)
        rows, cols, vals = [np.arange(n)], [np.arange(n)], [rng.uniform(6.5, 8.0, n)]
        for lo, hi in ((idx[:, :, :-1], idx[:, :, 1:]), (idx[:, :-1, :], idx[:, 1:, :]), (idx[:-1, :, :], idx[1:, :, :])):
            lo, hi = lo.ravel(), hi.ravel()
            for a_, b_ in ((lo, hi), (hi, lo)):
                keep = rng.random(len(a_)) > 0.25                      # a quarter of the couplings is missing, independently per direction
                rows.append(a_[keep]); cols.append(b_[keep]); vals.append(rng.uniform(-1.0, 1.0, int(keep.sum())))
        m = sp.csr_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(n, n))
        m.sort_indices()
        a = O.Csr(n, n, m.indptr, m.indices, m.data)
        d = to_dev(ctx, a)
        r = rng.standard_normal(n)
        for wave in ("2", "1", "0"):
            monkeypatch.setenv("KRYST_ILU_WAVE", wave)
            for kpc, ofn in ((K.TrueIlu0(), O.Pc.ilu0_true), (K.Ilu0(), O.Pc.ilu0_compat), (K.Ilup(0), O.Pc.ilup0)):
                assert np.array_equal(kpc.setup(d).apply(r), ofn(a).apply(r)), (seed, (Ni, Nj, Nk), wave)


@pytest.mark.parametrize("wave", ["1", "2"])
def test_wavefront_give_up_path_falls_back_to_plane_kernels(ctx, rs, wave, monkeypatch):
    """The wavefront triangular solve waits for neighbour blocks and so relies on in-order workgroup dispatch.  With a poll
    budget of ONE empty poll every block gives up at once (NaN results on the device): the host must notice (mapped give-up
    word), discard the result and repeat the work with the plane kernels (one launch per hyperplane, no inter-workgroup
    waits) -- the caller sees the oracle's bits, from pc.apply and from whole solves, and never a NaN."""
    monkeypatch.setenv("KRYST_ILU_POLL_BUDGET", "1")
    monkeypatch.setenv("KRYST_ILU_WAVE", wave)                     # the 8 x 8 (tri_wave.h) and the 16 x 16 (tri_quad.h) kernel
    ao = O.stencil7(24, "aniso")
    a = to_dev(ctx, ao)
    r = O.splitmix64_uniform(7, ao.nrows) - 0.5
    for kpc, ofn in ((K.TrueIlu0(), O.Pc.ilu0_true), (K.Ilu0(), O.Pc.ilu0_compat)):
        pc = kpc.setup(a); ref = ofn(ao)
        assert np.array_equal(pc.apply(r), ref.apply(r))          # gives up, repeated with the plane kernels
        assert np.array_equal(pc.apply(2.0 * r), ref.apply(2.0 * r))   # stays on the plane kernels
    b = ao.spmv(np.ones(ao.nrows))
    pc = K.TrueIlu0().setup(a)                                      # a fresh preconditioner: the give-up happens inside the solve
    res = O.solve("pcg", ao, b, pc=O.Pc.ilu0_true(ao), tol=1e-9, max_iters=200, rs=rs)
    s = K.PcgSolver(1e-9, 200); x = np.zeros(ao.nrows)
    st = s.solve(a, pc, b, x)
    assert st.iterations == res.iterations and np.array_equal(x, res.x)
    pc = K.Ilu0().setup(a)
    res = O.solve("gmres", ao, b, pc=O.Pc.ilu0_compat(ao), tol=1e-9, max_iters=100, restart=10, side=O.SIDE_RIGHT, rs=rs)
    g = K.GmresSolver(10, 1e-9, 100); g.preconditioning = K.Preconditioning.Right; x = np.zeros(ao.nrows)
    st = g.solve(a, pc, b, x)
    assert st.iterations == res.iterations and np.array_equal(x, res.x)
    # a stepping session cannot repeat itself: it reports the abandoned apply as a SolveError
    pc = K.TrueIlu0().setup(a)
    sess = K.Session("pcg", a, pc, K.DeviceVec(ctx, b), K.DeviceVec(ctx, np.zeros(ao.nrows)), tol=1e-9, max_iters=50)
    sess.step(50)
    with pytest.raises(K.KError) as e:
        sess.end()
    assert e.value.code == 2


def test_box_wavefront_give_up_path_falls_back_to_hyperplane_launches(ctx, rs, monkeypatch):
    """tri_box.h's pollers wait for neighbour blocks like tri_wave.h's: with a poll budget of one empty poll every block gives up; the host
    notices, repeats the apply with one launch per hyperplane i + 2 j + 4 k and stays there -- the oracle's bits throughout."""
    monkeypatch.setenv("KRYST_ILU_POLL_BUDGET", "1")
    monkeypatch.setenv("KRYST_ILU_BOX", "2")
    rng = np.random.default_rng(5)
    ao = _box_operator(rng, 20, 18, 17, lambda dk, dj, di: True)
    a = to_dev(ctx, ao)
    r = rng.standard_normal(ao.nrows)
    pc = K.TrueIlu0().setup(a); ref = O.Pc.ilu0_true(ao)
    assert pc.ilu_info()["form"].startswith("box wavefront")
    assert np.array_equal(pc.apply(r), ref.apply(r))              # gives up, repeated with the hyperplane kernels
    assert pc.ilu_info()["form"].startswith("box planes")
    assert np.array_equal(pc.apply(2.0 * r), ref.apply(2.0 * r))
    b = ao.spmv(np.ones(ao.nrows))
    pc = K.TrueIlu0().setup(a)                                      # a fresh preconditioner: the give-up happens inside the solve
    res = O.solve("gmres", ao, b, pc=O.Pc.ilu0_true(ao), tol=1e-10, max_iters=100, restart=10, side=O.SIDE_RIGHT, rs=rs)
    g = K.GmresSolver(10, 1e-10, 100); g.preconditioning = K.Preconditioning.Right; x = np.zeros(ao.nrows)
    st = g.solve(a, pc, b, x)
    assert st.iterations == res.iterations and np.array_equal(x, res.x)
    assert pc.ilu_info()["form"].startswith("box planes")


@pytest.mark.parametrize("seed", [1, 2])
def test_device_side_grid_setup_equals_host_setup(ctx, seed, monkeypatch):
    """Grid operators are factored on the device (grid_setup_on_device: coefficient streams pulled out of the CSR arrays, compat /
    Ilup(0) quotients pointwise, textbook ILU(0) as a recurrence over hyperplanes); KRYST_ILU_DEVICE_SETUP=0 keeps the host loops.
    Both must give the oracle's bits -- random boxes, random coefficients, randomly MISSING couplings and missing diagonals'
    neighbours, stored zeros -- and report the same zero pivot."""
    import scipy.sparse as sp
    rng = np.random.default_rng(500 + seed)
    for _ in range(4):
        Ni, Nj, Nk = (int(v) for v in rng.integers(3, 30, 3))
        if rng.random() < 0.25:
            Nk = 1                                                             # 2-D operators too
        n = Ni * Nj * Nk
        idx = np.arange(n).reshape(Nk, Nj, Ni)
        rows, cols, vals = [np.arange(n)], [np.arange(n)], [rng.uniform(6.5, 8.0, n)]
        for lo, hi in ((idx[:, :, :-1], idx[:, :, 1:]), (idx[:, :-1, :], idx[:, 1:, :]), (idx[:-1, :, :], idx[1:, :, :])):
            lo, hi = lo.ravel(), hi.ravel()
            for a_, b_ in ((lo, hi), (hi, lo)):
                keep = rng.random(len(a_)) > 0.2
                v = rng.uniform(-1.0, 1.0, int(keep.sum()))
                v[rng.random(len(v)) < 0.05] = 0.0                             # stored zeros: eliminated like any entry, never kept
                rows.append(a_[keep]); cols.append(b_[keep]); vals.append(v)
        m = sp.coo_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(n, n)).tocsr()
        m.sort_indices()
        a = O.Csr(n, n, m.indptr, m.indices, m.data)
        d = to_dev(ctx, a)
        r = rng.standard_normal(n)
        for kpc, ofn in ((K.TrueIlu0, O.Pc.ilu0_true), (K.Ilu0, O.Pc.ilu0_compat), (lambda: K.Ilup(0), O.Pc.ilup0)):
            ref = ofn(a).apply(r)
            for dev in ("1", "0"):
                monkeypatch.setenv("KRYST_ILU_DEVICE_SETUP", dev)
                assert np.array_equal(kpc().setup(d).apply(r), ref), (seed, (Ni, Nj, Nk), dev)
    # a zero pivot in the middle of a grid: u_dd(row) = a_dd - l * u = 0 exactly; both paths name the same row
    Ni, Nj, Nk = 5, 4, 3
    n = Ni * Nj * Nk
    main = np.full(n, 4.0)
    m = sp.diags([main, -np.ones(n - 1), -np.ones(n - 1)], [0, -1, 1]).tolil()
    for i in range(Ni, n, Ni):
        m[i, i - 1] = 0.0; m[i - 1, i] = 0.0                                   # no coupling across line ends
    m = m.tocsr(); m.eliminate_zeros()
    m = (m + sp.diags([-np.ones(n - Ni), -np.ones(n - Ni)], [-Ni, Ni])).tolil()
    for k in range(1, Nk):
        for i in range(Ni):
            m[k * Ni * Nj + i, k * Ni * Nj + i - Ni] = 0.0; m[k * Ni * Nj + i - Ni, k * Ni * Nj + i] = 0.0
    m = m.tocsr(); m.eliminate_zeros()
    m = m.tolil()
    m[1, 1] = 0.25                                                             # row 1: u_11 = 0.25 - (-1/4)(-1) = 0
    m = m.tocsr(); m.sort_indices()
    d = to_dev(ctx, O.Csr(n, n, m.indptr, m.indices, m.data))
    for dev in ("1", "0"):
        monkeypatch.setenv("KRYST_ILU_DEVICE_SETUP", dev)
        with pytest.raises(K.KError) as e:
            K.TrueIlu0().setup(d)
        assert e.value.code == 5 and e.value.row == 1, dev


def test_device_side_general_factorisation_equals_host_and_oracle(ctx, monkeypatch):
    """True ILU(0) of operators that are NOT 7-point boxes is eliminated on the device too (ilu0_ikj_syncfree_kernel: one lane per row,
    rows concurrent along the dependency graph, the host loop's operations in the host loop's order); KRYST_ILU_DEVICE_SETUP=0 keeps
    the sequential host loop.  Both give the oracle's bits on random sparse operators (unsymmetric patterns, long and empty lower
    parts, deep and shallow dependency graphs, a band matrix that only looks like a grid), name the same zero-pivot row, and a
    starved poll budget falls back to the host loop without changing a bit."""
    import scipy.sparse as sp
    rng = np.random.default_rng(2024)

    def rand_op(n, per_row, band=None):
        rows = np.repeat(np.arange(n), per_row)
        if band:
            cols = np.clip(rows + rng.integers(-band, band + 1, len(rows)), 0, n - 1)
        else:
            cols = rng.integers(0, n, len(rows))
        m = sp.csr_matrix((rng.uniform(-1.0, 1.0, len(rows)), (rows, cols)), shape=(n, n))
        m.sum_duplicates()
        m = m - sp.diags(m.diagonal()) + sp.diags(np.asarray(abs(m).sum(axis=1)).ravel() + 1.0 + rng.random(n))
        m = m.tocsr(); m.sort_indices()
        return O.Csr(n, n, m.indptr, m.indices, m.data)

    n_w = 6 * 5 * 4
    offs = [o for o in (-30, -6, -1, 0, 1, 6, 30)]
    wrap = sp.diags([(-1.0 if o else 7.0) * np.ones(n_w - abs(o)) for o in offs], offs).tocsr(); wrap.sort_indices()
    cases = [rand_op(50, 3), rand_op(3000, 6), rand_op(20000, 9, band=40), rand_op(7000, 25, band=300), rand_op(1500, 2), rand_op(600, 90),          # (rows longer than a wave: host loop)
             O.Csr(n_w, n_w, wrap.indptr, wrap.indices, wrap.data), O.Csr.from_dense(O.tridiag(900, -1.0, 2.5, -0.5), keep_zeros=False)]
    for a in cases:
        d = to_dev(ctx, a)
        ref = O.Pc.ilu0_true(a)
        r = rng.standard_normal(a.nrows)
        want = ref.apply(r)
        for dev, budget in (("1", None), ("0", None), ("1", "1")):
            monkeypatch.setenv("KRYST_ILU_DEVICE_SETUP", dev)
            if budget:
                monkeypatch.setenv("KRYST_ILU_SETUP_POLL_BUDGET", budget)                      # gives up at once -> host loop
            else:
                monkeypatch.delenv("KRYST_ILU_SETUP_POLL_BUDGET", raising=False)
            assert np.array_equal(K.TrueIlu0().setup(d).apply(r), want), (a.nrows, a.nnz, dev, budget)
    monkeypatch.delenv("KRYST_ILU_SETUP_POLL_BUDGET", raising=False)
    # zero pivot: u_11 = 1 - 1 * 1 = 0 is met when row 2 eliminates with row 1; rows beyond it must not change which row is named
    n = 40
    m = sp.lil_matrix((n, n))
    for i in range(n):
        m[i, i] = 4.0
        if i > 0:
            m[i, i - 1] = 1.0
        if i + 1 < n:
            m[i, i + 1] = 1.0
    m[0, 0] = 1.0; m[0, 1] = 1.0; m[1, 0] = 1.0; m[1, 1] = 1.0
    m = m.tocsr(); m.sort_indices()
    d = to_dev(ctx, O.Csr(n, n, m.indptr, m.indices, m.data))
    for dev in ("1", "0"):
        monkeypatch.setenv("KRYST_ILU_DEVICE_SETUP", dev)
        with pytest.raises(K.KError) as e:
            K.TrueIlu0().setup(d)
        assert e.value.code == 5 and e.value.row == 1, dev


def test_measurement_hooks_describe_what_the_apply_runs(ctx, monkeypatch):
    """kryst_pc_ilu_info / kryst_bench_pc_apply (bench.py prices the triangular solve with them): the form an ILU-family apply runs,
    its chunk bookkeeping (a constant-coefficient box repeats most coefficient chunks, a variable-coefficient one none), dependency
    levels of a level-ordered factor, and a timing loop that leaves the same z as a single apply."""
    for kind, repeats in (("aniso", True), ("varcoef", False)):
        a = K.CsrMatrix.stencil7(64, kind, ctx=ctx)
        pc = K.TrueIlu0().setup(a)
        info = pc.ilu_info()
        assert info["form"].startswith("grid 16x16") and info["box"] == [64, 64, 64]
        assert info["chunks"][0] == info["chunks"][1] == 16 * 4 * ((64 + 14 + 7) // 8) and info["bytes_per_chunk"] == [12288, 16384]
        assert (min(info["chunks_not_requested"]) > 0) == repeats
        r = ctx.vec(a.nrows()).fill_splitmix(5)
        z1 = pc.apply(r).to_host()
        z2 = ctx.vec(a.nrows())
        assert pc.bench_apply(r, z2, 3) > 0.0 and np.array_equal(z2.to_host(), z1)
    rng = np.random.default_rng(8)
    d = rng.random((200, 200)) * (rng.random((200, 200)) < 0.05) + np.diag(4.0 + rng.random(200))
    pc = K.TrueIlu0().setup(to_dev(ctx, O.Csr.from_dense(d, keep_zeros=False)))
    info = pc.ilu_info()
    assert info["form"] == "level-ordered" and min(info["levels"]) >= 1 and info["chunks"] == [0, 0]
    with pytest.raises(K.KError):
        K.Jacobi().setup(K.CsrMatrix.stencil7(8, "poisson", ctx=ctx)).ilu_info()


def test_plane_kernels_bit_exact(ctx, monkeypatch):
    """KRYST_ILU_PLANES=1: the fallback of the wavefront solve on its own, ragged boxes included."""
    monkeypatch.setenv("KRYST_ILU_PLANES", "1")
    rng = np.random.default_rng(11)
    for N, kind in ((9, "poisson"), (17, "convdiff"), (20, "aniso")):
        ao = O.stencil7(N, kind)
        a = to_dev(ctx, ao)
        r = rng.standard_normal(ao.nrows)
        for kpc, ofn in ((K.TrueIlu0(), O.Pc.ilu0_true), (K.Ilu0(), O.Pc.ilu0_compat), (K.Ilup(0), O.Pc.ilup0)):
            assert np.array_equal(kpc.setup(a).apply(r), ofn(ao).apply(r)), (N, kind)


def test_ilu_apply_twice_reuses_graph(ctx):
    a = O.stencil7(20)
    pc = K.Ilu0().setup(to_dev(ctx, a))
    ref = O.Pc.ilu0_compat(a)
    for seed in (1, 2, 3):
        r = O.splitmix64_uniform(seed, a.nrows)
        assert np.array_equal(pc.apply(r), ref.apply(r))


def test_refactoring_reuses_pooled_device_blocks_and_trim_returns_them(ctx):
    """Preconditioner::setup is called per matrix, repeatedly (ilup.rs:77-134): the device blocks of a destroyed ILU-family preconditioner go to
    the context's size-keyed pool and the next set-up of the same operator takes them back -- blocks that are NOT cleared in between, so every
    byte a set-up relies on must be written by it (tests/conftest.py sets KRYST_DEV_POOL_POISON: every block that enters the pool is filled with
    0xFF bytes); kryst_ctx_trim gives the pool back."""
    import os
    assert os.environ.get("KRYST_DEV_POOL_POISON") == "1"
    rng = np.random.default_rng(21)
    ctx.trim()
    cases = [(O.stencil7(40, "aniso"), lambda: K.CsrMatrix.stencil7(40, "aniso", ctx=ctx)),      # 7-point grid: device set-up, blocked layouts (> 1 MiB blocks)
             (random_csr_fast(rng, 60000, 60000, 9), None)]                                      # general operator: level-ordered factors, operand streams
    for ao, make in cases:
        if make is None:                                                                        # (a diagonally dominant copy: the factorisation must not break down)
            import scipy.sparse as sp
            m = sp.csr_matrix((ao.vals, ao.col_idx, ao.row_ptr), shape=(ao.nrows, ao.ncols))
            m = (m - sp.diags(m.diagonal()) + sp.diags(np.asarray(abs(m).sum(axis=1)).ravel() + 1.0)).tocsr(); m.sort_indices()
            ao = O.Csr(m.shape[0], m.shape[1], m.indptr, m.indices, m.data)
        a = make() if make else to_dev(ctx, ao)
        r = rng.standard_normal(ao.nrows)
        ref = O.Pc.ilu0_true(ao).apply(r)
        for round_ in range(3):
            pc = K.TrueIlu0().setup(a)
            assert np.array_equal(pc.apply(r), ref), round_
            del pc                                                                              # (its blocks go to the pool, poisoned)
        freed = ctx.trim()
        assert freed > 0, "nothing was pooled"
        pc = K.TrueIlu0().setup(a)                                                              # and after the trim: fresh blocks, same bits
        assert np.array_equal(pc.apply(r), ref)
        del pc
    assert ctx.trim() >= 0


# ------------------------------------------------------------------------------------------------ solvers
def _check_solver(res, stats, solver, x, exact=True):
    assert stats.iterations == res.iterations and stats.converged == res.converged
    h = np.array(solver.residual_history)
    assert len(h) == len(res.history)
    if exact:
        assert np.array_equal(h, res.history), np.max(np.abs(h - res.history) / np.abs(res.history))
        assert stats.final_residual == res.final_residual
        assert np.array_equal(x, res.x)


@pytest.mark.parametrize("defer_x", ["1", "0"])
@pytest.mark.parametrize("N", [4, 12, 24])
@pytest.mark.parametrize("norm", [K.CgNormType.Unpreconditioned, K.CgNormType.Natural, K.CgNormType.NoNorm])
def test_cg_bit_exact(ctx, rs, N, norm, defer_x, monkeypatch):
    """defer_x: x += alpha p riding on the direction pass (the default; solvers.hip) or done where the reference does it -- same bits."""
    monkeypatch.setenv("KRYST_CG_DEFER_X", defer_x)
    a = O.stencil7(N)
    b = a.spmv(np.ones(a.nrows))
    res = O.solve("cg", a, b, tol=1e-8, max_iters=400, norm_type=int(norm), rs=rs)
    s = K.CgSolver(1e-8, 400).with_norm(norm)
    x = np.zeros(a.nrows)
    st = s.solve(to_dev(ctx, a), None, b, x)
    _check_solver(res, st, s, x)


def test_cg_vs_serial_fold_reference(ctx):
    """Against the strict left-fold reference build (--no-default-features): equal iteration counts, residual
    history within 1e-12 relative to the initial residual (north_star's tolerance), first iterations within 1e-12
    of their own value."""
    a = O.stencil7(24)
    b = a.spmv(np.ones(a.nrows))
    res = O.solve("cg", a, b, tol=1e-8, max_iters=400)
    s = K.CgSolver(1e-8, 400)
    x = np.zeros(a.nrows)
    st = s.solve(to_dev(ctx, a), None, b, x)
    h = np.array(s.residual_history)
    assert st.iterations == res.iterations and st.converged
    assert np.max(np.abs(h[:10] - res.history[:10]) / res.history[:10]) < 1e-12
    assert np.max(np.abs(h - res.history)) <= 1e-12 * res.history[0]
    assert np.linalg.norm(x - res.x) / np.linalg.norm(res.x) < 1e-12


@pytest.mark.parametrize("defer_x", ["1", "0"])
@pytest.mark.parametrize("pcname", ["none", "identity", "jacobi"])
@pytest.mark.parametrize("norm", [K.CgNormType.Preconditioned, K.CgNormType.Unpreconditioned, K.CgNormType.Natural])
def test_pcg_bit_exact(ctx, rs, pcname, norm, defer_x, monkeypatch):
    monkeypatch.setenv("KRYST_CG_DEFER_X", defer_x)
    a = O.stencil7(14, "aniso")
    b = O.splitmix64_uniform(0x5EED, a.nrows)
    d = to_dev(ctx, a)
    opc = {"none": None, "identity": O.Pc.identity(), "jacobi": O.Pc.jacobi(a)}[pcname]
    kpc = {"none": None, "identity": K.IdentityPc().setup(d), "jacobi": K.Jacobi().setup(d)}[pcname]
    res = O.solve("pcg", a, b, pc=opc, tol=1e-9, max_iters=500, norm_type=int(norm), rs=rs)
    s = K.PcgSolver(1e-9, 500).with_norm(norm)
    x = np.zeros(a.nrows)
    st = s.solve(d, kpc, b, x)
    _check_solver(res, st, s, x)


@pytest.mark.parametrize("defer_x,xbatch", [("1", "1"), ("1", "2"), ("1", "8"), ("0", "1")])
def test_deferred_x_update_on_every_exit_path(ctx, rs, defer_x, xbatch, monkeypatch):
    """x += alpha p travels with the direction pass (p is read once per iteration).  Every way out of CG / PCG must leave x as the
    reference does: convergence, the iteration cap (1, 2, 7 iterations -- `converged` true, x includes the last alpha p), p.Ap <= 0
    (x untouched by that iteration), a non-pointwise preconditioner (ILU: the unfused PCG path), a stepping session ended after k
    iterations, and an initial guess.  xbatch > 1: the direction vectors in a ring, x paid in batches of that many iterations (XBatchOp)."""
    monkeypatch.setenv("KRYST_CG_DEFER_X", defer_x); monkeypatch.setenv("KRYST_CG_X_BATCH", xbatch)
    a = O.stencil7(9, "aniso"); d = to_dev(ctx, a)
    b = O.splitmix64_uniform(0xD0E, a.nrows)
    x0 = O.splitmix64_uniform(0xABC, a.nrows)
    for cap in (1, 2, 7, 300):
        for name, cls, opc, kpc in (("cg", K.CgSolver, None, None), ("pcg", K.PcgSolver, O.Pc.jacobi(a), K.Jacobi().setup(d)),
                                    ("pcg", K.PcgSolver, O.Pc.ilu0_true(a), K.TrueIlu0().setup(d))):
            res = O.solve(name, a, b, pc=opc, x0=x0, tol=1e-10, max_iters=cap, rs=rs)
            s = cls(1e-10, cap); x = x0.copy()
            st = s.solve(d, kpc, b, x)
            _check_solver(res, st, s, x)
    # p.Ap <= 0 in the second iteration: diag(1, -1, 2) from b = (1, 1, 1): x keeps the first iteration's update only
    ind = O.Csr(3, 3, [0, 1, 2, 3], [0, 1, 2], [1.0, -1.0, 2.0]); bi = np.ones(3)
    for name, cls in (("cg", K.CgSolver), ("pcg", K.PcgSolver)):
        s = cls(1e-12, 10); x = np.zeros(3); xin = x.copy()
        with pytest.raises(K.KError) as e:
            s.solve(to_dev(ctx, ind), None, bi, x)
        assert e.value.code == 3 and np.array_equal(x, xin)                  # IndefiniteMatrix; on Err the reference never writes x back
    # a stepping session ended after k of 1000 allowed iterations: x holds exactly k updates
    for method, opc, kpc in (("cg", None, None), ("pcg", O.Pc.jacobi(a), K.Jacobi().setup(d))):
        for k in (1, 3, 6):
            res = O.solve(method, a, b, pc=opc, tol=1e-30, max_iters=k, rs=rs)
            xs, bs = K.DeviceVec(ctx, np.zeros(a.nrows)), K.DeviceVec(ctx, b)
            sess = K.Session(method, d, kpc, bs, xs, tol=1e-30, max_iters=1000)
            sess.step(k)
            sess.end()
            assert np.array_equal(xs.to_host(), res.x), (method, k)


@pytest.mark.parametrize("fuse,xbatch", [("1", "1"), ("1", "4"), ("1", "3"), ("1", "8"), ("0", "1"), ("0", "2"), ("0", "3"), ("0", "8")])
@pytest.mark.parametrize("N,kind,T", [(10, "aniso", "2"), (16, "poisson", "2"), (16, "poisson", "4"), (32, "convdiff", "2"), (40, "poisson", "4")])
def test_direction_pass_inside_the_spmv_on_every_exit_path(ctx, rs, fuse, xbatch, N, kind, T, monkeypatch):
    """Round 5: on stencil operators in their staged CSR-P16 form CG / PCG form p = z + beta p_old INSIDE the next iteration's SpMV
    (spmv.hip: spmv_pattern_fuse_kernel) and the deferred x += alpha p rides on the same pass -- two direction vectors alternate, the x
    update of iteration k is paid by the fused SpMV of iteration k + 1 or, when none follows, by a flush at the end.  KRYST_CG_FUSE_P=0 is
    the unfused form.  Both must leave iteration counts, histories and x as the oracle does on every way out: convergence (the device runs
    ahead of the host: fused SpMVs enqueued behind the final iteration must pay the owed update exactly once), the iteration cap (1, 2, 3,
    8 -- the flush), an initial guess, Jacobi / identity / ILU preconditioners, stepping sessions in several steps, runs of 2 and of 4 tiles,
    grids whose last run is partial.  xbatch > 1: x is updated in BATCHES -- the direction vectors of the last m iterations stay in a ring and
    x += alpha_i p_i for i = k - m + 1 .. k happens in one pass every m iterations (XBatchOp), a partial batch at the end (caps and session
    lengths that are not multiples of m, solves that end inside a batch).  fuse = 0 with xbatch > 1: the UNFUSED form with the ring -- the direction
    pass writes p_new = z + beta p_old out of place (CgDirectionRingOp) and never touches x."""
    monkeypatch.setenv("KRYST_CG_FUSE_P", fuse); monkeypatch.setenv("KRYST_SPMV_FUSE_T", T); monkeypatch.setenv("KRYST_CG_X_BATCH", xbatch)
    a = O.stencil7(N, kind)
    d = K.CsrMatrix.stencil7(N, kind, ctx=ctx) if N % 4 == 0 else to_dev(ctx, a)
    assert d.encoding()[0] == "csr-p16" and d.pattern_info()["staged"]
    b = O.splitmix64_uniform(0xD0E + N, a.nrows)
    x0 = O.splitmix64_uniform(0xABC, a.nrows)
    sym = kind != "convdiff"
    pcs = [("cg", K.CgSolver, None, None), ("pcg", K.PcgSolver, O.Pc.jacobi(a), K.Jacobi().setup(d)), ("pcg", K.PcgSolver, None, None),
           ("pcg", K.PcgSolver, O.Pc.identity(), K.IdentityPc().setup(d))]
    if N <= 16:
        pcs.append(("pcg", K.PcgSolver, O.Pc.ilu0_true(a), K.TrueIlu0().setup(d)))
    for cap in (1, 2, 3, 4, 5, 8, 9, 400):
        for name, cls, opc, kpc in pcs:
            res = O.solve(name, a, b, pc=opc, x0=x0, tol=1e-9, max_iters=cap, rs=rs, raise_on_error=False)
            s = cls(1e-9, cap); x = x0.copy()
            try:
                st, code = s.solve(d, kpc, b, x), 0
            except K.KError as e:                           # (the unsymmetric operator: IndefiniteMatrix / IndefinitePreconditioner must match too)
                st, code = e.stats, e.code
            assert code == res.code, (name, cap, code, res.code)
            _check_solver(res, st, s, x, exact=(code == 0))
    if sym:
        for method, opc, kpc in (("cg", None, None), ("pcg", O.Pc.jacobi(a), K.Jacobi().setup(d))):
            for steps in ((1,), (2, 1), (3, 4), (1, 1, 1, 5), (4, 4), (5, 6, 2)):
                k = sum(steps)
                res = O.solve(method, a, b, pc=opc, tol=1e-30, max_iters=k, rs=rs)
                xs, bs = K.DeviceVec(ctx, np.zeros(a.nrows)), K.DeviceVec(ctx, b)
                with K.Session(method, d, kpc, bs, xs, tol=1e-30, max_iters=1000) as sess:
                    for q in steps:
                        sess.step(q)
                    st = sess.end()
                assert st.iterations == k and np.array_equal(xs.to_host(), res.x) and np.array_equal(np.array(sess.residual_history), res.history), (method, steps)
        # a session that converges in the middle of a step: the iterations enqueued behind the last one must not touch x again
        res = O.solve("cg", a, b, tol=1e-6, max_iters=1000, rs=rs)
        xs, bs = K.DeviceVec(ctx, np.zeros(a.nrows)), K.DeviceVec(ctx, b)
        with K.Session("cg", d, None, bs, xs, tol=1e-6, max_iters=1000) as sess:
            sess.step(res.iterations + 7)
            st = sess.end()
        assert st.iterations == res.iterations and st.converged and np.array_equal(xs.to_host(), res.x)


def test_pcg_with_chebyshev_extension_bit_exact(ctx, rs):
    a = O.stencil7(10)
    b = a.spmv(np.ones(a.nrows))
    d = to_dev(ctx, a)
    res = O.solve("pcg", a, b, pc=O.Pc.chebyshev(a, 0.3, 11.8, 3), tol=1e-8, max_iters=200, rs=rs, raise_on_error=False)
    s = K.PcgSolver(1e-8, 200)
    x = np.zeros(a.nrows)
    try:
        st = s.solve(d, K.ChebyshevPc(3, 0.3, 11.8).setup(d), b, x)
        code = 0
    except K.KError as e:
        st, code = e.stats, e.code
    assert code == res.code
    _check_solver(res, st, s, x, exact=(code == 0))


def test_initial_guess_is_honoured(ctx, rs):
    a = O.stencil7(8)
    b = a.spmv(np.ones(a.nrows))
    x0 = O.splitmix64_uniform(9, a.nrows)
    res = O.solve("cg", a, b, x0=x0, tol=1e-10, max_iters=300, rs=rs)
    s = K.CgSolver(1e-10, 300)
    x = x0.copy()
    st = s.solve(to_dev(ctx, a), None, b, x)
    _check_solver(res, st, s, x)


def test_iteration_cap_reports_converged_true(ctx, rs):
    # convergence.rs:25
    a = O.stencil7(8)
    b = a.spmv(np.ones(a.nrows))
    for cls, m in ((K.CgSolver, "cg"), (K.PcgSolver, "pcg")):
        s = cls(1e-30, 5)
        x = np.zeros(a.nrows)
        st = s.solve(to_dev(ctx, a), None, b, x)
        res = O.solve(m, a, b, tol=1e-30, max_iters=5, rs=rs)
        assert st.iterations == 5 and st.converged
        _check_solver(res, st, s, x)


def test_indefinite_matrix_error_and_x_untouched(ctx):
    a = K.CsrMatrix.from_csr(2, 2, [0, 1, 2], [0, 1], [1.0, -1.0], ctx=ctx)
    for cls in (K.CgSolver, K.PcgSolver):
        x = np.array([0.25, 0.5])
        with pytest.raises(K.KError) as e:
            cls(1e-10, 10).solve(a, None, np.array([0.0, 1.0]), x)
        assert e.value.code == 3                              # KError::IndefiniteMatrix (cg.rs:168-174)
        assert np.array_equal(x, [0.25, 0.5])                 # Err: `*x = ...` is never reached


def test_monitor_and_history(ctx):
    a = O.stencil7(6)
    b = a.spmv(np.ones(a.nrows))
    seen = []
    s = K.CgSolver(1e-8, 100).with_monitor(lambda i, r: seen.append((i, r)))
    x = np.zeros(a.nrows)
    st = s.solve(to_dev(ctx, a), None, b, x)
    assert [i for i, _ in seen] == list(range(st.iterations + 1))
    assert [r for _, r in seen] == s.residual_history
    n1 = len(s.residual_history)
    s.solve(to_dev(ctx, a), None, b, x)                         # residual_history accumulates across solves (cg.rs:140)
    assert len(s.residual_history) > n1
    s.clear_history()
    assert s.residual_history == []


@pytest.mark.parametrize("solver", ["cg", "pcg", "gmres"])
def test_monitor_is_live_and_in_order(ctx, rs, solver):
    """with_monitor (cg.rs:84-88): the reference calls the monitor inside the loop (cg.rs:137-140,260-263).  Here the host fires
    it from its poll loop while the device iterates: callbacks arrive in order, each exactly once, equal to the residual
    history, all before solve() returns -- and the early ones while later iterations have not happened yet (the value a
    callback sees in the device's progress record lags behind the end of the solve)."""
    import time
    N = 48
    ao = O.stencil7(N)
    a = to_dev(ctx, ao)
    b = ao.spmv(np.ones(ao.nrows))
    seen, stamps = [], []
    def mon(i, r):
        seen.append((i, r)); stamps.append(time.perf_counter())
    if solver == "gmres":
        s = K.GmresSolver(5, 1e-10, 400).with_monitor(mon); pc = None
    elif solver == "pcg":
        s = K.PcgSolver(1e-12, 400).with_monitor(mon); pc = K.Jacobi().setup(a)
    else:
        s = K.CgSolver(1e-12, 400).with_monitor(mon); pc = None
    s.check_every = 1 if solver == "cg" else 0
    x = np.zeros(ao.nrows)
    t0 = time.perf_counter()
    st = s.solve(a, pc, b, x)
    t1 = time.perf_counter()
    first = 1 if solver == "gmres" else 0                       # GMRES numbers its history entries from 1
    assert [i for i, _ in seen] == list(range(first, first + len(s.residual_history)))
    assert [r for _, r in seen] == s.residual_history
    assert st.iterations > 40 and len(seen) >= st.iterations
    assert all(t0 <= t <= t1 for t in stamps)
    # live: the first tenth of the callbacks was delivered before the last tenth of the solve's wall time began
    k = max(1, len(stamps) // 10)
    assert stamps[k] < t0 + 0.9 * (t1 - t0), "monitor callbacks were replayed after the solve instead of fired during it"


def test_one_open_solve_per_context(ctx, rs):
    """The scalar state of a solve lives in per-context scratch (kryst_hip.h, Contexts): while a stepping session is open a
    second solve or session on the same context is refused with KRYST_ERR_BUSY, inner products, SpMV and vector updates stay
    legal and do not disturb the session, and a second context solves concurrently."""
    ao = O.stencil7(12)
    a = to_dev(ctx, ao)
    b = ao.spmv(np.ones(ao.nrows))
    ref = O.solve("cg", ao, b, tol=1e-10, max_iters=200, rs=rs)
    bv, xv = K.DeviceVec(ctx, b), K.DeviceVec(ctx, np.zeros(ao.nrows))
    sess = K.Session("cg", a, None, bv, xv, tol=1e-10, max_iters=200)
    sess.step(3)
    with pytest.raises(K.KError) as e:
        K.CgSolver(1e-10, 200).solve(a, None, b, np.zeros(ao.nrows))
    assert e.value.code == 104
    with pytest.raises(K.KError) as e:
        K.Session("cg", a, None, bv, K.DeviceVec(ctx, np.zeros(ao.nrows)), tol=1e-10, max_iters=5)
    assert e.value.code == 104
    u = K.DeviceVec(ctx, b)
    d = K.dot(u, u)                                               # legal between steps; result slot of its own
    assert d == O.dot(b, b, rs)
    assert np.array_equal(a.spmv(np.ones(ao.nrows)), b)
    ctx2 = K.Context(0)                                           # a second context: independent scratch
    a2 = K.CsrMatrix.from_csr(ao.nrows, ao.ncols, ao.row_ptr, ao.col_idx, ao.vals, ctx=ctx2)
    x2 = np.zeros(ao.nrows)
    st2 = K.CgSolver(1e-10, 200).solve(a2, None, b, x2)
    assert st2.iterations == ref.iterations and np.array_equal(x2, ref.x)
    sess.step(200)
    st = sess.end()
    assert st.iterations == ref.iterations and np.array_equal(np.array(sess.residual_history), ref.history)
    assert np.array_equal(xv.to_host(), ref.x)
    x3 = np.zeros(ao.nrows)                                       # the context is free again
    assert K.CgSolver(1e-10, 200).solve(a, None, b, x3).iterations == ref.iterations


def test_zero_pivot_reports_its_row(ctx):
    """KError::ZeroPivot(row) (src/error.rs:15-16): the row travels through kryst_hip_last_error_row()."""
    d = np.array([[2.0, 1.0, 0.0, 0.0], [1.0, 0.5, 1.0, 0.0], [0.0, 1.0, 3.0, 1.0], [0.0, 0.0, 1.0, 2.0]])   # u_11 = 0.5 - 1*1/2 = 0
    a = to_dev(ctx, O.Csr.from_dense(d, keep_zeros=False))
    with pytest.raises(K.KError) as e:
        K.TrueIlu0().setup(a)
    assert e.value.code == 5 and e.value.row == 1


def test_huge_iteration_cap_does_not_allocate_a_huge_history(ctx, rs):
    """max_iters = 10^9 ("unbounded"): the history buffer is capped (2^22 entries), the count is still exact."""
    ao = O.stencil7(8)
    b = ao.spmv(np.ones(ao.nrows))
    ref = O.solve("cg", ao, b, tol=1e-10, max_iters=10 ** 9, rs=rs)
    s = K.CgSolver(1e-10, 10 ** 9); x = np.zeros(ao.nrows)
    st = s.solve(to_dev(ctx, ao), None, b, x)
    assert st.iterations == ref.iterations and np.array_equal(np.array(s.residual_history), ref.history) and np.array_equal(x, ref.x)


@pytest.mark.parametrize("kind,N", [("convdiff", 8), ("aniso", 12), ("poisson", 5)])
def test_bicgstab_bit_exact(ctx, rs, kind, N):
    a = O.stencil7(N, kind)
    b = a.spmv(np.ones(a.nrows))
    tol = 1e-8 * np.linalg.norm(b)                              # absolute tolerance (bicgstab.rs:98,189,281)
    res = O.solve("bicgstab", a, b, tol=tol, max_iters=300, rs=rs)
    s = K.BiCgStabSolver(tol, 300)
    x = np.zeros(a.nrows)
    st = s.solve(to_dev(ctx, a), K.Jacobi().setup(to_dev(ctx, a)), b, x)     # pc is ignored (bicgstab.rs:70)
    _check_solver(res, st, s, x)


def test_bicgstab_edge_exits(ctx, rs):
    a = O.stencil7(6, "convdiff")
    d = to_dev(ctx, a)
    b = a.spmv(np.ones(a.nrows))
    # already converged at entry (bicgstab.rs:98-102), iteration cap (converged=false), s-norm early exit
    for tol, mx, x0 in ((1e6, 50, None), (1e-30, 7, None), (1e-3, 50, None)):
        res = O.solve("bicgstab", a, b, tol=tol, max_iters=mx, rs=rs)
        s = K.BiCgStabSolver(tol, mx)
        x = np.zeros(a.nrows)
        st = s.solve(d, None, b, x)
        _check_solver(res, st, s, x)
    # exact solution in one step on a diagonal system -> s-norm exit
    a2 = O.Csr.from_dense(np.diag([2.0, 2.0, 2.0]))
    res = O.solve("bicgstab", a2, [1.0, 2.0, 3.0], tol=1e-12, max_iters=10, rs=rs)
    s = K.BiCgStabSolver(1e-12, 10)
    x = np.zeros(3)
    st = s.solve(to_dev(ctx, a2), None, np.array([1.0, 2.0, 3.0]), x)
    _check_solver(res, st, s, x)


@pytest.mark.parametrize("side", [K.Preconditioning.NoPc, K.Preconditioning.Left, K.Preconditioning.Right, K.Preconditioning.LeftTextbook])
@pytest.mark.parametrize("restart", [5, 30])
def test_gmres_bit_exact(ctx, rs, side, restart):
    a = O.stencil7(8, "convdiff")
    b = a.spmv(np.ones(a.nrows))
    d = to_dev(ctx, a)
    opc = None if side == K.Preconditioning.NoPc else O.Pc.jacobi(a)
    kpc = None if side == K.Preconditioning.NoPc else K.Jacobi().setup(d)
    res = O.solve("gmres", a, b, pc=opc, tol=1e-8, max_iters=90, restart=restart, side=int(side), rs=rs)
    s = K.GmresSolver(restart, 1e-8, 90).with_preconditioning(side)
    x = np.zeros(a.nrows)
    st = s.solve(d, kpc, b, x)
    _check_solver(res, st, s, x)


@pytest.mark.parametrize("mode", [0, 1, 2])
def test_solvers_with_ilu_bit_exact(ctx, rs, mode):
    _, kcls, ofn = ILU_MODES[mode]
    a = O.stencil7(10, "aniso")
    b = a.spmv(np.ones(a.nrows))
    d = to_dev(ctx, a)
    kpc, opc = kcls().setup(d), ofn(a)
    # PCG (the symmetric modes are SPD preconditioners; compat is not symmetric: errors must match too)
    res = O.solve("pcg", a, b, pc=opc, tol=1e-9, max_iters=200, rs=rs, raise_on_error=False)
    s = K.PcgSolver(1e-9, 200); x = np.zeros(a.nrows)
    try:
        st, code = s.solve(d, kpc, b, x), 0
    except K.KError as e:
        st, code = e.stats, e.code
    assert code == res.code
    _check_solver(res, st, s, x, exact=(code == 0))
    # GMRES-Left (tests/preconditioner_integration.rs:169-179 shape) and the right-preconditioned BiCGStab extension
    an = O.stencil7(9, "convdiff"); bn = an.spmv(np.ones(an.nrows)); dn = to_dev(ctx, an)
    kpc, opc = kcls().setup(dn), ofn(an)
    res = O.solve("gmres", an, bn, pc=opc, tol=1e-9, max_iters=60, restart=20, side=O.SIDE_LEFT, rs=rs)
    s = K.GmresSolver(20, 1e-9, 60); x = np.zeros(an.nrows)
    st = s.solve(dn, kpc, bn, x)
    _check_solver(res, st, s, x)
    # ... and the textbook Left extension (precond_side 3) with the same factors
    res = O.solve("gmres", an, bn, pc=opc, tol=1e-9, max_iters=60, restart=20, side=O.SIDE_LEFT_TEXTBOOK, rs=rs)
    s = K.GmresSolver(20, 1e-9, 60).with_preconditioning(K.Preconditioning.LeftTextbook); x = np.zeros(an.nrows)
    st = s.solve(dn, kpc, bn, x)
    _check_solver(res, st, s, x)
    tol = 1e-9 * np.linalg.norm(bn)
    res = O.solve("bicgstab_rpc", an, bn, pc=opc, tol=tol, max_iters=100, rs=rs)
    s = K.BiCgStabRightPcSolver(tol, 100); x = np.zeros(an.nrows)
    st = s.solve(dn, kpc, bn, x)
    _check_solver(res, st, s, x)


def test_nonsym_left_ilu0_gmres_reference_test(ctx, rs):
    # tests/preconditioner_integration.rs:169-179: needs two restart cycles = 20 iterations
    n = 10
    ao = O.Csr.from_dense(O.tridiag(n, -1.0, 2.0, 0.5), keep_zeros=False)
    a = to_dev(ctx, ao)
    b = ao.spmv(np.ones(n)); x = np.zeros(n)
    st = K.GmresSolver(10, 1e-12, 100).with_preconditioning(K.Preconditioning.Left).solve(a, K.Ilu0().setup(a), b, x)
    assert st.converged and st.iterations == 20 and np.linalg.norm(x - 1) / np.sqrt(n) < 1e-10


def test_gmres_reference_known_answers(ctx):
    # src/solver/gmres.rs:438-528 through the mirrored API
    A = [[4.0, 1.0, 0.0, 0.0], [1.0, 3.0, 1.0, 0.0], [0.0, 1.0, 2.0, 1.0], [0.0, 0.0, 1.0, 3.0]]
    ao = O.Csr.from_dense(A)
    a = to_dev(ctx, ao)
    xt = np.array([1.0, 2.0, 3.0, 4.0])
    b = ao.spmv(xt)
    x = np.zeros(4)
    st = K.GmresSolver(4, 1e-10, 100).solve(a, None, b, x)
    assert st.converged and np.all(np.abs(x - xt) < 1e-8)
    x = np.zeros(4)
    st = K.GmresSolver(4, 1e-10, 100).solve(a, K.Jacobi().setup(a), b, x)
    assert st.converged and np.all(np.abs(x - xt) < 1e-8)
    x = np.zeros(4)
    K.GmresSolver(4, 1e-10, 100).with_preconditioning(K.Preconditioning.Right).solve(a, K.Jacobi().setup(a), b, x)
    assert np.linalg.norm(ao.spmv(x) - b) < 1e-2


def test_gmres_happy_breakdown_paths(ctx, rs):
    # identity system: h[1][0] == 0 at the first step in every branch (gmres.rs:99-101, 300-303, 332-335)
    ao = O.Csr.from_dense(np.eye(6))
    a = to_dev(ctx, ao)
    b = np.arange(1.0, 7.0)
    for side in (0, 1, 2, 3):
        opc = None if side == 0 else O.Pc.jacobi(ao)
        kpc = None if side == 0 else K.Jacobi().setup(a)
        res = O.solve("gmres", ao, b, pc=opc, tol=1e-10, max_iters=12, restart=4, side=side, rs=rs)
        s = K.GmresSolver(4, 1e-10, 12).with_preconditioning(K.Preconditioning(side))
        x = np.zeros(6)
        st = s.solve(a, kpc, b, x)
        assert st.iterations == res.iterations and st.converged == res.converged, (side, st, res)
        assert np.array_equal(x, res.x) and st.final_residual == res.final_residual


def test_reference_solver_known_answers_via_api(ctx):
    # cg.rs:310-323, pcg.rs:253-275, bicgstab.rs:316-328, tests/preconditioner_integration.rs:127-164
    a2 = to_dev(ctx, O.Csr.from_dense([[4.0, 1.0], [1.0, 3.0]]))
    exp = [0.09090909090909091, 0.6363636363636364]
    for solver in (K.CgSolver(1e-10, 20), K.CgSolver(1e-10, 20).with_single_reduction(True), K.PcgSolver(1e-10, 20)):
        x = np.zeros(2)
        pc = K.IdentityPc().setup(a2) if isinstance(solver, K.PcgSolver) else None
        st = solver.solve(a2, pc, np.array([1.0, 2.0]), x)
        assert st.converged and np.all(np.abs(x - exp) < 1e-8)
    dense = np.array([[4.0 if i == j else (i + 2 * j) + 1.0 for j in range(3)] for i in range(3)])
    ao = O.Csr.from_dense(dense)
    xt = np.array([1.0, 2.0, 3.0]); x = np.zeros(3)
    st = K.BiCgStabSolver(1e-10, 100).solve(to_dev(ctx, ao), None, ao.spmv(xt), x)
    assert st.converged and np.all(np.abs(x - xt) < 1e-8)
    n = 10
    ao = O.Csr.from_dense(O.tridiag(n, -1.0, 2.0, -1.0), keep_zeros=False)
    a = to_dev(ctx, ao)
    b = ao.spmv(np.ones(n)); x = np.zeros(n)
    st = K.PcgSolver(1e-12, n).solve(a, K.Jacobi().setup(a), b, x)
    assert st.converged and st.iterations <= n and np.linalg.norm(x - 1) / np.sqrt(n) < 1e-10
    ao = O.Csr.from_dense(O.tridiag(n, -1.0, 2.0, 0.5), keep_zeros=False)
    b = ao.spmv(np.ones(n)); x = np.zeros(n)
    st = K.GmresSolver(10, 1e-12, 100).solve(to_dev(ctx, ao), None, b, x)
    assert st.converged and np.linalg.norm(x - 1) / np.sqrt(n) < 1e-10


def test_device_resident_solve_and_roundtrip_property(ctx):
    """Size-independent property at a size the oracle would be slow on: b = A*1 => CG returns x ~ 1 with
    ||b - A x|| <= tol * ||b|| (checked with the device SpMV)."""
    N = 96
    a = K.CsrMatrix.stencil7(N, "poisson", ctx=ctx)
    n = N ** 3
    ones = ctx.vec(n).fill(1.0)
    b = a.spmv(ones)
    x = ctx.vec(n)
    s = K.CgSolver(1e-8, 2000)
    st = s.solve(a, None, b, x)
    assert st.converged and 0 < st.iterations < 2000
    r = a.spmv(x)
    K.axpy(-1.0, b, r)
    assert K.norm(r) <= 1.0001e-8 * K.norm(b)
    assert abs(s.residual_history[-1] - st.final_residual) == 0.0
    assert np.max(np.abs(x.to_host() - 1.0)) < 1e-6


def test_ksp_context_factory(ctx, rs):
    # src/context/ksp_context.rs:88-148: one level of dispatch, a fresh solver per call
    a = O.stencil7(8, "convdiff")
    b = a.spmv(np.ones(a.nrows))
    d = to_dev(ctx, a)
    for kind, method, side in ((K.SolverKind.Cg, "cg", 1), (K.SolverKind.Pcg, "pcg", 1), (K.SolverKind.GmresLeft, "gmres", 1),
                               (K.SolverKind.GmresRight, "gmres", 2), (K.SolverKind.Bicgstab, "bicgstab", 1)):
        res = O.solve(method, a, b, pc=O.Pc.jacobi(a), tol=1e-8, max_iters=60, restart=30, side=side, rs=rs, raise_on_error=False)
        x = np.zeros(a.nrows)
        try:
            st, code = K.KspContext(kind, d, K.Jacobi().setup(d), 1e-8, 60, 30).solve_context(b, x), 0
        except K.KError as e:
            st, code = e.stats, e.code
        assert code == res.code and st.iterations == res.iterations and st.converged == res.converged
        if code == 0:
            assert np.array_equal(x, res.x)


@pytest.mark.parametrize("radius", [0.5, 3.0, 7.5, 1e9])
def test_cg_trust_region_exit_bit_exact(ctx, rs, radius):
    # cg.rs:177-202 (Steihaug-Toint): stop on the radius with x += max_step * p, converged = false
    a = O.stencil7(10)
    b = a.spmv(np.ones(a.nrows))
    res = O.solve("cg", a, b, tol=1e-8, max_iters=200, radius=radius, rs=rs)
    s = K.CgSolver(1e-8, 200).with_radius(radius)
    x = np.zeros(a.nrows)
    st = s.solve(to_dev(ctx, a), None, b, x)
    _check_solver(res, st, s, x)
    if radius < 1e8:
        assert not st.converged and abs(np.linalg.norm(x) - radius) < 1e-9 * radius


@pytest.mark.parametrize("norm", [K.CgNormType.Unpreconditioned, K.CgNormType.Natural])
def test_cg_objective_target_exit_bit_exact(ctx, rs, norm):
    # cg.rs:231-252: obj = 0.5 x.Ax - x.b, stop (converged = true) once obj <= target
    a = O.stencil7(10)
    b = a.spmv(np.ones(a.nrows))
    obj_final = -0.5 * float(np.ones(a.nrows) @ b)            # at the solution x = 1
    for target in (0.9 * obj_final, 0.999999 * obj_final, 2.0 * obj_final):
        res = O.solve("cg", a, b, tol=1e-8, max_iters=200, obj_target=target, norm_type=int(norm), rs=rs)
        s = K.CgSolver(1e-8, 200).with_obj_target(target).with_norm(norm)
        x = np.zeros(a.nrows)
        st = s.solve(to_dev(ctx, a), None, b, x)
        _check_solver(res, st, s, x)
    # PcgSolver carries the same fields but its solve never reads them (pcg.rs:114-222)
    res = O.solve("pcg", a, b, tol=1e-8, max_iters=200, rs=rs)
    s = K.PcgSolver(1e-8, 200).with_radius(0.1).with_obj_target(1e30)
    x = np.zeros(a.nrows)
    _check_solver(res, s.solve(to_dev(ctx, a), None, b, x), s, x)


@pytest.mark.parametrize("fill", [0, 1, 2, 3])
def test_ilup_level_of_fill_bit_exact(ctx, rs, fill):
    """Ilup::new(p) (ilup.rs:77-167): the sparse-row setup of the library against the reference's dense algorithm as
    written (oracle), then the level-scheduled apply against its row-sequential apply -- bit for bit."""
    rng = np.random.default_rng(40 + fill)
    d = rng.random((60, 60)) * (rng.random((60, 60)) < 0.12) + np.diag(4.0 + rng.random(60))
    for a in (O.stencil7(6, "convdiff"), O.stencil7(7, "aniso"), O.Csr.from_dense(d, keep_zeros=False)):
        r = rng.standard_normal(a.nrows)
        opc = O.Pc.ilup(a, fill)
        z = K.Ilup(fill).setup(to_dev(ctx, a)).apply(r)
        assert np.array_equal(z, opc.apply(r)), (fill, a.nrows)
    a = O.stencil7(6, "convdiff"); b = a.spmv(np.ones(a.nrows)); dev = to_dev(ctx, a)
    res = O.solve("gmres", a, b, pc=O.Pc.ilup(a, fill), tol=1e-10, max_iters=60, restart=20, side=O.SIDE_LEFT, rs=rs)
    s = K.GmresSolver(20, 1e-10, 60); x = np.zeros(a.nrows)
    _check_solver(res, s.solve(dev, K.Ilup(fill).setup(dev), b, x), s, x)


@pytest.mark.parametrize("threads,block", [(4, 8), (16, 32), (3, 1), (16, 2048)])
def test_ilup_row_pipeline_over_host_threads_bit_exact(ctx, threads, block, monkeypatch):
    """Ilup(p >= 1) is eliminated by a pipeline of host threads (round 4: rows dealt out in blocks, a thread waits on a per-row flag before it uses
    another thread's pivot row).  Small operators take one thread by default; here the pipeline is forced onto them with blocks of 1 .. 32 rows,
    so that nearly every pivot row is another thread's -- against the oracle's dense algorithm (ilup.rs:77-167) bit for bit, fill 1 .. 3, on
    stencils, a random sparse matrix and one with stored zeros and cancellations; and at 40^3 the 16-thread result against the one-thread one."""
    monkeypatch.setenv("KRYST_ILUP_THREADS", str(threads))
    monkeypatch.setenv("KRYST_ILUP_BLOCK", str(block))
    rng = np.random.default_rng(77)
    d = rng.random((90, 90)) * (rng.random((90, 90)) < 0.1) + np.diag(4.0 + rng.random(90))
    z = np.array(O.stencil7(5, "poisson").to_dense()); z[z == -1.0] = np.where(rng.random((z == -1.0).sum()) < 0.3, 0.0, -1.0)   # zeros: entries that never take part
    for a in (O.stencil7(9, "convdiff"), O.stencil7(8, "poisson"), O.Csr.from_dense(d, keep_zeros=False), O.Csr.from_dense(z, keep_zeros=False)):
        for fill in (1, 2, 3):
            r = rng.standard_normal(a.nrows)
            assert np.array_equal(K.Ilup(fill).setup(to_dev(ctx, a)).apply(r), O.Pc.ilup(a, fill).apply(r)), (threads, block, fill, a.nrows)
    big = K.CsrMatrix.stencil7(40, "aniso", ctx=ctx)
    r = np.random.default_rng(3).standard_normal(big.nrows())
    got = K.Ilup(1).setup(big).apply(r)
    monkeypatch.setenv("KRYST_ILUP_THREADS", "1")
    assert np.array_equal(got, K.Ilup(1).setup(big).apply(r)), (threads, block)


@pytest.mark.parametrize("fill,droptol", [(7, 1e-12), (3, 1e-12), (4, 0.6), (1, 0.0), (0, 0.0)])
def test_ilut_bit_exact(ctx, rs, fill, droptol):
    """Ilut::new(fill, droptol) (ilut.rs:80-150): rows truncated to the `fill` largest entries keep their sorted
    (descending magnitude) order, which is the order the apply subtracts them in."""
    rng = np.random.default_rng(50 + fill)
    d = rng.standard_normal((50, 50)) * (rng.random((50, 50)) < 0.2) + np.diag(5.0 + rng.random(50))
    for a in (O.stencil7(6, "convdiff"), O.Csr.from_dense(d, keep_zeros=False)):
        r = rng.standard_normal(a.nrows)
        z = K.Ilut(fill, droptol).setup(to_dev(ctx, a)).apply(r)
        assert np.array_equal(z, O.Pc.ilut(a, fill, droptol).apply(r)), (fill, droptol, a.nrows)
    if fill == 4:                                                   # 110 592 rows: the set-up's row loops run on all host cores (round 4)
        big = O.stencil7(48, "aniso")
        r = rng.standard_normal(big.nrows)
        assert np.array_equal(K.Ilut(fill, 1e-3).setup(to_dev(ctx, big)).apply(r), O.Pc.ilut(big, fill, 1e-3).apply(r))
    # reference tests ilut.rs:185-211
    ident = to_dev(ctx, O.Csr.from_dense([[1.0, 0.0], [0.0, 1.0]]))
    assert np.array_equal(K.Ilut(2, 1e-12).setup(ident).apply(np.array([2.0, 3.0])), [2.0, 3.0])
    a = O.stencil7(6, "convdiff"); b = a.spmv(np.ones(a.nrows)); dev = to_dev(ctx, a)
    res = O.solve("gmres", a, b, pc=O.Pc.ilut(a, fill, droptol), tol=1e-10, max_iters=40, restart=20, side=O.SIDE_LEFT, rs=rs,
                  raise_on_error=False)
    s = K.GmresSolver(20, 1e-10, 40); x = np.zeros(a.nrows)
    _check_solver(res, s.solve(dev, K.Ilut(fill, droptol).setup(dev), b, x), s, x)


def test_zero_rhs_and_exact_guess_edge_cases(ctx, rs):
    """b = 0 / x0 already exact: the reference divides by a zero initial residual (NaN comparisons are false) or hits
    p.Ap = 0 -> Err(IndefiniteMatrix); whatever it does, the device must do the same."""
    a = O.stencil7(6)
    d = to_dev(ctx, a)
    n = a.nrows
    for b, x0 in ((np.zeros(n), np.zeros(n)), (a.spmv(np.ones(n)), np.ones(n))):
        for method, cls in (("cg", K.CgSolver), ("pcg", K.PcgSolver), ("bicgstab", K.BiCgStabSolver)):
            res = O.solve(method, a, b, x0=x0, tol=1e-8, max_iters=20, rs=rs, raise_on_error=False)
            s = cls(1e-8, 20); x = x0.copy()
            try:
                st, code = s.solve(d, None, b, x), 0
            except K.KError as e:
                st, code = e.stats, e.code
            assert code == res.code, (method, code, res.code)
            assert st.iterations == res.iterations and st.converged == res.converged
            assert np.array_equal(np.array(s.residual_history), res.history, equal_nan=True)
            if code == 0:
                assert np.array_equal(x, res.x, equal_nan=True)


def test_max_iters_zero(ctx, rs):
    a = O.stencil7(5); b = a.spmv(np.ones(a.nrows)); d = to_dev(ctx, a)
    for method, mk in (("cg", lambda: K.CgSolver(1e-8, 0)), ("pcg", lambda: K.PcgSolver(1e-8, 0)),
                       ("bicgstab", lambda: K.BiCgStabSolver(1e-8, 0)), ("gmres", lambda: K.GmresSolver(5, 1e-8, 0))):
        res = O.solve(method, a, b, tol=1e-8, max_iters=0, restart=5, rs=rs)
        s = mk(); x = np.zeros(a.nrows)
        st = s.solve(d, None, b, x)
        assert (st.iterations, st.converged, st.final_residual) == (res.iterations, res.converged, res.final_residual), method
        assert np.array_equal(x, res.x)


# ------------------------------------------------------------------------------------------------ SPAI apply (§8 f-4)
def test_approx_inverse_apply_and_solve(ctx, rs):
    """ApproxInv::apply with given inverse rows (approxinv.rs:268-298) = the sparse-row product on the device; inside PCG and
    right-preconditioned GMRES it must match the oracle bit for bit."""
    a = O.stencil7(8, "poisson")
    d = to_dev(ctx, a)
    # a crude sparse approximate inverse on A's own pattern: M = D^-1 (2I - A D^-1) restricted to the pattern (symmetric here)
    dinv = 1.0 / np.array([a.vals[a.row_ptr[i]:a.row_ptr[i + 1]][list(a.col_idx[a.row_ptr[i]:a.row_ptr[i + 1]]).index(i)] for i in range(a.nrows)])
    rows = []
    for i in range(a.nrows):
        row = []
        for k in range(a.row_ptr[i], a.row_ptr[i + 1]):
            j = int(a.col_idx[k])
            v = (2.0 * dinv[i] if i == j else 0.0) - dinv[i] * a.vals[k] * dinv[j]
            row.append((j, v))
        rows.append(row)
    kpc = K.ApproxInv(rows, ctx=ctx).setup(d)
    m = O.Csr(a.nrows, a.nrows, a.row_ptr, a.col_idx, [v for row in rows for _, v in row])
    opc = O.Pc.approx_inverse(m)
    r = np.linspace(-1.0, 1.0, a.nrows)
    assert np.array_equal(kpc.apply(r), opc.apply(r))
    b = a.spmv(np.ones(a.nrows))
    res = O.solve("pcg", a, b, pc=opc, tol=1e-10, max_iters=200, rs=rs)
    s = K.PcgSolver(1e-10, 200); x = np.zeros(a.nrows)
    _check_solver(res, s.solve(d, kpc, b, x), s, x)
    assert res.converged and res.iterations < O.solve("cg", a, b, tol=1e-10, max_iters=200, rs=rs).iterations
    res = O.solve("gmres", a, b, pc=opc, tol=1e-10, max_iters=60, restart=20, side=O.SIDE_RIGHT, rs=rs)
    s = K.GmresSolver(20, 1e-10, 60).with_preconditioning(K.Preconditioning.Right); x = np.zeros(a.nrows)
    _check_solver(res, s.solve(d, kpc, b, x), s, x)
    with pytest.raises(K.KError):
        K.ApproxInv(rows[:5], ctx=ctx).setup(d)


@pytest.mark.parametrize("seed", [1, 2, 3, 4])
def test_spmv_random_stencil_matrices_all_forms(ctx, seed, monkeypatch):
    """Matrices assembled from a few random stencils with boundary truncation: many row patterns that are sub-sequences of
    a few bases (the CSR-P16 base + mask builder), adjacent rows with equal and with different bases, bases longer than the
    unroll factor, masks with holes.  Every storage form must equal the oracle bit for bit."""
    rng = np.random.default_rng(seed)
    n = int(rng.integers(600, 2500))
    nst = int(rng.integers(1, 6))
    stencils = []
    for _ in range(nst):
        k = int(rng.integers(1, 15))
        offs = np.sort(rng.choice(np.arange(-40, 41), size=k, replace=False))
        vals = rng.choice([1.0, -1.0, 0.5, 4.0, -0.25, 1e-3], size=k)
        drop = rng.random(k) < 0.15                      # entries that some rows additionally lack (masks with holes)
        stencils.append((offs, vals, drop))
    rp = [0]; ci = []; va = []
    sid = 0
    for i in range(n):
        if rng.random() < 0.02:
            sid = int(rng.integers(0, nst))              # runs of rows with the same stencil
        offs, vals, drop = stencils[sid]
        holes = rng.random() < 0.1
        for o, v, d in zip(offs, vals, drop):
            c = i + int(o)
            if 0 <= c < n and not (holes and d):
                ci.append(c); va.append(v)
        rp.append(len(ci))
    a = O.Csr(n, n, rp, ci, va)
    x = rng.standard_normal(n)
    ref = a.spmv(x)
    for level in ("0", "1", "2", "3"):
        monkeypatch.setenv("KRYST_SPMV_COMPRESS", level)
        assert np.array_equal(to_dev(ctx, a).spmv(x), ref), (seed, level)
    monkeypatch.setenv("KRYST_SPMV_COMPRESS", "3")
    d = to_dev(ctx, a)
    name, npat, ntab = d.encoding()
    assert name == "csr-p16" and npat >= nst and ntab <= 2048


@pytest.mark.parametrize("seed", [11, 12, 13, 14, 15, 16])
def test_solvers_on_random_sparse_matrices_bit_exact(ctx, rs, seed):
    """Every solver on random diagonally dominant sparse matrices (unstructured pattern: the plain-CSR kernels carry the
    SpMVs, several tiles, ragged rows), symmetric and not, against the oracle bit for bit."""
    rng = np.random.default_rng(seed)
    n = int(rng.integers(700, 1800))
    dens = 6.0 / n
    m = (rng.random((n, n)) < dens) * rng.standard_normal((n, n))
    if seed % 2 == 0:
        m = 0.5 * (m + m.T)                                          # symmetric: CG / PCG apply
    m = m + np.diag(np.abs(m).sum(axis=1) + 1.0 + rng.random(n))
    a = O.Csr.from_dense(m, keep_zeros=False)
    d = to_dev(ctx, a)
    assert d.encoding()[0] == "csr"
    b = a.spmv(rng.standard_normal(n))
    jac_o, jac_k = O.Pc.jacobi(a), K.Jacobi().setup(d)
    runs = [("gmres", lambda: K.GmresSolver(15, 1e-10, 90).with_preconditioning(K.Preconditioning.Right), jac_o, jac_k,
             dict(tol=1e-10, max_iters=90, restart=15, side=O.SIDE_RIGHT)),
            ("fgmres", lambda: K.FgmresSolver(1e-10, 90, 15), jac_o, jac_k, dict(tol=1e-10, max_iters=90, restart=15)),
            ("bicgstab", lambda: K.BiCgStabSolver(1e-9, 200), None, None, dict(tol=1e-9, max_iters=200)),
            ("cgs", lambda: K.CgsSolver(1e-10, 200), None, None, dict(tol=1e-10, max_iters=200))]
    if seed % 2 == 0:
        runs += [("cg", lambda: K.CgSolver(1e-10, 300), None, None, dict(tol=1e-10, max_iters=300)),
                 ("pcg", lambda: K.PcgSolver(1e-10, 300), jac_o, jac_k, dict(tol=1e-10, max_iters=300))]
    for method, mk, opc, kpc, kw in runs:
        res = O.solve(method, a, b, pc=opc, rs=rs, **kw)
        s = mk(); x = np.zeros(n)
        st = s.solve(d, kpc, b, x)
        _check_solver(res, st, s, x)                   # parity is the point: whether the reference's variant converges is its business


def test_pc_enum_constructor(ctx, rs):
    """PC<T> (pc_context.rs:36-76) with the constructor the reference lacks, through KspContext (ksp_context.rs:88-148)."""
    a = O.stencil7(8, "convdiff"); d = to_dev(ctx, a)
    b = a.spmv(np.ones(a.nrows))
    for cfg, opc in ((K.PC.Jacobi(), O.Pc.jacobi(a)), (K.PC.Ilu0(), O.Pc.ilu0_compat(a)), (K.PC.Ilup(1), O.Pc.ilup(a, 1)),
                     (K.PC.Ilut(4, 1e-3), O.Pc.ilut(a, 4, 1e-3))):
        res = O.solve("gmres", a, b, pc=opc, tol=1e-9, max_iters=60, restart=20, side=O.SIDE_LEFT, rs=rs)
        x = np.zeros(a.nrows)
        st = K.KspContext(K.SolverKind.GmresLeft, d, pc=cfg.build(d), tol=1e-9, max_it=60, restart=20).solve_context(b, x)
        assert st.iterations == res.iterations and np.array_equal(x, res.x), cfg
    with pytest.raises(K.KError) as e:
        K.PC("AMG").build(d)
    assert e.value.code == 6
    with pytest.raises(K.KError) as e:                      # the Chebyshev trait object builds, its apply is the reference's stub
        K.PC.Chebyshev(3, 0.1, 12.0).build(d).apply(np.ones(a.nrows))
    assert e.value.code == 2
