"""The host-side factorisations on plain host arrays (kryst_host_ilup / kryst_host_ilut / kryst_host_levels: kryst_amd/csrc/host_factor.cpp,
the code kryst_pc_ilup / kryst_pc_ilut run between download and upload) against the oracle's dense restatements of Ilup::setup
(ilup.rs:77-134) and Ilut::setup (ilut.rs:80-117) -- without a GPU.  The row pipeline must give the bits of the one-thread loop for ANY thread
count and block size, including which zero pivot it reports."""
import numpy as np
import pytest
import scipy.sparse as sp

import kryst_amd as K
from kryst_amd._ffi import lib
from oracle import oracle as O


def random_dd(n, density, seed, nonsym=True):
    rng = np.random.default_rng(seed)
    m = sp.random(n, n, density=density, random_state=rng, data_rvs=lambda k: rng.uniform(-1.0, 1.0, k)).tocsr()
    if not nonsym:
        m = (m + m.T).tocsr()
    m.setdiag(0.0); m.eliminate_zeros()
    m = (m + sp.diags(np.asarray(abs(m).sum(axis=1)).ravel() + 1.0)).tocsr()
    m.sort_indices()
    return m


def poisson(N, kind="poisson"):
    rp, ci, va = K.host_stencil7(N, kind)
    return sp.csr_matrix((va, ci, rp), shape=(N ** 3, N ** 3))


def oracle_rows(pc, n):
    """The oracle's TRIROWS factors in the host entry points' shape: strictly-lower L, strictly-upper U, diagonal (1.0 where U keeps none)."""
    lp, lc, lv, up, uc, uv = pc.tri_rows()
    dg = np.ones(n)
    keep = np.ones(len(uc), dtype=bool)
    nup = np.zeros(n + 1, dtype=np.int64)
    for i in range(n):
        seen = False
        for k in range(up[i], up[i + 1]):
            if uc[k] == i and not seen:
                dg[i] = uv[k]; keep[k] = False; seen = True
            elif uc[k] == i:
                keep[k] = False
        nup[i + 1] = nup[i] + int(keep[up[i]:up[i + 1]].sum())
    return lp, lc, lv, nup, uc[keep], uv[keep], dg


def same(got, ref):
    lp, lc, lv, up, uc, uv, dg = got
    rlp, rlc, rlv, rup, ruc, ruv, rdg = ref
    assert np.array_equal(lp, rlp) and np.array_equal(up, rup)
    assert np.array_equal(lc.astype(np.int64), rlc) and np.array_equal(uc.astype(np.int64), ruc)
    assert np.array_equal(lv, rlv) and np.array_equal(uv, ruv) and np.array_equal(dg, rdg)


CASES = [("random 150", random_dd(150, 0.04, 1)), ("random symmetric 90", random_dd(90, 0.06, 2, nonsym=False)), ("poisson 6^3", poisson(6)),
         ("convdiff 5^3", poisson(5, "convdiff")), ("aniso 7^3", poisson(7, "aniso"))]


@pytest.mark.parametrize("name,m", CASES, ids=[c[0] for c in CASES])
@pytest.mark.parametrize("fill", [1, 2, 3])
def test_host_ilup_equals_the_oracle_for_every_thread_count_and_block_size(name, m, fill):
    n = m.shape[0]
    ref = oracle_rows(O.Pc.ilup(O.Csr(n, n, m.indptr, m.indices, m.data), fill), n)
    for threads in (1, 3, 16):
        for block in (1, 2, 7, 64, 2048):
            same(K.host_ilup(m.indptr, m.indices, m.data, fill, threads=threads, block=block), ref)


def test_host_ilup_stored_zeros_and_missing_diagonals():
    """`!= 0.0` tests (ilup.rs:106,117,129) decide on VALUES: a stored zero never starts an elimination (and never counts as a level-0 entry), and
    a row whose diagonal is not kept divides by nothing (dg = 1.0)."""
    n = 14
    a = np.zeros((n, n))
    rng = np.random.default_rng(5)
    for i in range(n - 1):
        a[i, i] = 4.0 + i
        for j in (i - 3, i - 1, i + 1, i + 2):
            if 0 <= j < n - 1:
                a[i, j] = rng.uniform(-1.0, 1.0)
    pat = np.abs(a) > 0
    for i in range(2, n - 1):
        pat[i, i - 2] = True                                    # explicit stored zeros on the second sub-diagonal ...
    pat[3, 9] = pat[4, 11] = True                               # ... and two in the upper part
    pat[n - 1, :] = False                                       # the last row stores nothing at all: no kept diagonal, U row empty
    rows, cols = np.nonzero(pat)
    m = sp.csr_matrix((a[rows, cols], (rows, cols)), shape=(n, n)); m.sort_indices()
    assert (m.data == 0.0).sum() >= 10 and m.indptr[n] == m.indptr[n - 1]
    for fill in (1, 2, 3):
        ref = oracle_rows(O.Pc.ilup(O.Csr(n, n, m.indptr, m.indices, m.data), fill), n)
        assert ref[6][n - 1] == 1.0
        for threads, block in ((1, 2048), (3, 2), (4, 5), (16, 1)):
            same(K.host_ilup(m.indptr, m.indices, m.data, fill, threads=threads, block=block), ref)


def zero_pivot_matrix(n, j):
    """Tridiagonal-plus matrix whose row j has no lower entry and a zero diagonal: row j + 1 (a_{j+1,j} != 0) divides by u_jj = 0."""
    a = np.zeros((n, n))
    for i in range(n):
        a[i, i] = 4.0
        if i > 0:
            a[i, i - 1] = -1.0
        if i + 1 < n:
            a[i, i + 1] = -1.0
        if i > 4:
            a[i, i - 5] = -0.5
    a[j, j] = 0.0
    for k in range(j):
        a[j, k] = 0.0
    m = sp.csr_matrix(a); m.sort_indices()
    return m


@pytest.mark.parametrize("threads", [1, 3, 16])
def test_host_ilup_zero_pivot_in_every_position_of_a_block(threads):
    """ilup.rs:108-110 returns at the FIRST row that meets a zero u_jj; the pipeline must report that row's j whichever thread found which pivot
    first -- the zero pivot placed at every position of a block (block = 8 rows), and two of them at once."""
    n, block = 40, 8
    for j in range(0, 2 * block + 2):
        m = zero_pivot_matrix(n, j)
        with pytest.raises(O.KrylovError):
            O.Pc.ilup(O.Csr(n, n, m.indptr, m.indices, m.data), 1)
        with pytest.raises(K.KError) as e:
            K.host_ilup(m.indptr, m.indices, m.data, 1, threads=threads, block=block)
        assert e.value.code == 2 and lib().kryst_hip_last_error_row() == j, (j, e.value, lib().kryst_hip_last_error_row())
    # two zero pivots: the lower row's wins
    m = zero_pivot_matrix(n, 23).tolil()
    m[9, 9] = 0.0
    for k in range(9):
        m[9, k] = 0.0
    m = m.tocsr(); m.eliminate_zeros(); m.sort_indices()
    with pytest.raises(K.KError):
        K.host_ilup(m.indptr, m.indices, m.data, 2, threads=threads, block=4)
    assert lib().kryst_hip_last_error_row() == 9


@pytest.mark.parametrize("name,m", CASES, ids=[c[0] for c in CASES])
def test_host_ilut_equals_the_oracle(name, m):
    n = m.shape[0]
    for fill, droptol in ((2, 0.0), (4, 1e-3), (3, 0.3), (50, 0.0), (0, 0.0)):
        ref = oracle_rows(O.Pc.ilut(O.Csr(n, n, m.indptr, m.indices, m.data), fill, droptol), n)
        for threads in (0, 1, 3, 16):
            same(K.host_ilut(m.indptr, m.indices, m.data, fill, droptol, threads=threads), ref)


def test_host_factorisations_drop_halo_columns():
    """A rank's block of a row-partitioned operator carries halo columns numbered n + slot: they are outside the block factor."""
    m = random_dd(60, 0.08, 9)
    n = 40
    blk = m[:n].tocsr()                                         # 40 rows, 60 columns: columns >= 40 are "halo slots"
    sq = m[:n, :n].tocsr(); sq.sort_indices()
    same(K.host_ilup(blk.indptr, blk.indices, blk.data, 2, threads=3, block=4), K.host_ilup(sq.indptr, sq.indices, sq.data, 2))
    same(K.host_ilut(blk.indptr, blk.indices, blk.data, 3, 1e-2, threads=3), K.host_ilut(sq.indptr, sq.indices, sq.data, 3, 1e-2))


def test_host_levels_are_the_longest_dependency_chains():
    m = random_dd(300, 0.02, 11)
    lo = sp.tril(m, -1).tocsr(); lo.sort_indices()
    up = sp.triu(m, 1).tocsr(); up.sort_indices()
    for t, fwd in ((lo, True), (up, False)):
        lvl, nl = K.host_levels(t.indptr, t.indices, forward=fwd)
        ref = np.zeros(300, dtype=np.int64)
        order = range(300) if fwd else range(299, -1, -1)
        for i in order:
            deps = t.indices[t.indptr[i]:t.indptr[i + 1]]
            ref[i] = 0 if len(deps) == 0 else ref[deps].max() + 1
        assert np.array_equal(lvl, ref) and nl == ref.max() + 1
    # a 7-point grid: level = i + j + k (forward), 3N - 3 - (i + j + k) (backward): 3N - 2 levels each
    N = 6
    g = poisson(N)
    lo = sp.tril(g, -1).tocsr(); lo.sort_indices()
    lvl, nl = K.host_levels(lo.indptr, lo.indices, forward=True)
    idx = np.arange(N ** 3)
    assert nl == 3 * N - 2 and np.array_equal(lvl, idx % N + (idx // N) % N + idx // (N * N))
    with pytest.raises(K.KError):
        K.host_levels(g.indptr, g.indices, forward=True)         # not strictly lower


def test_host_entry_points_reject_bad_rows():
    with pytest.raises(K.KError):
        K.host_ilup([0, 2, 1], [0, 1], [1.0, 2.0], 1)          # lengths disagree with row_ptr
    rp = np.array([0, 1, 0], dtype=np.int64)                   # decreasing row_ptr
    ci = np.zeros(1, dtype=np.int32); va = np.ones(1)
    h = K._ffi.Handle()
    import ctypes as C
    assert lib().kryst_host_ilup(2, rp.ctypes.data_as(K._ffi.c_i64p), ci.ctypes.data_as(K._ffi.c_i32p), va.ctypes.data_as(K._ffi.c_dp), 1, 1, 0, C.byref(h)) == 102
    # an empty operator is fine
    got = K.host_ilup([0], [], [], 1)
    assert len(got[0]) == 1 and len(got[6]) == 0


@pytest.mark.parametrize("san", ["thread", "address,undefined"])
def test_sanitizer_tier_is_clean(san):
    """`make -C kryst_amd/csrc san SAN=...`: the host-side concurrency (Ilup row pipeline, parallel Ilut rows, host pool + janitor thread, counted
    shared mappings with fake handles) built for the CPU with the sanitizer and run against the oracle -- exit code 0, no report."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run(["make", "-C", os.path.join(root, "kryst_amd", "csrc"), "san", f"SAN={san}"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "test_host_san ok" in r.stdout and "Sanitizer" not in r.stdout + r.stderr and "runtime error" not in r.stdout + r.stderr
