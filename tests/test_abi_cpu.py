"""CPU-side checks of the C-ABI library: it loads, exports every symbol include/kryst_hip.h declares, its
host-only helpers agree with the oracle's independent generators, and compute calls fail loudly without a GPU."""
import os
import re
import numpy as np
import pytest

import kryst_amd as K
from kryst_amd import _ffi
from oracle import oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    txt = open(os.path.join(ROOT, "include", "kryst_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(kryst_[a-z0-9_]+)\s*\(", txt)))


def test_every_declared_symbol_is_exported_and_bound():
    names = _declared()
    assert len(names) >= 50
    L = K.lib()
    for nm in names:
        assert hasattr(L, nm), f"{nm} declared in kryst_hip.h but not exported"
        assert nm in _ffi.SIGNATURES, f"{nm} has no ctypes signature"
    assert set(_ffi.SIGNATURES) <= set(names)
    assert K.lib().kryst_hip_abi_version() == 5


def test_reduce_spec():
    T, V, F = K.reduce_spec()
    assert T % 64 == 0 and F % 64 == 0 and V >= 1


@pytest.mark.parametrize("kind", ["poisson", "aniso", "convdiff", "varcoef"])
def test_host_stencil_matches_oracle_generator(kind):
    N = 6
    for k_lo, k_hi in ((0, N), (0, 2), (2, 5), (5, 6)):
        rp, ci, va = K.host_stencil7(N, kind, k_lo, k_hi)
        ref = O.stencil7(N, kind, k_lo, k_hi)
        assert np.array_equal(rp, ref.row_ptr) and np.array_equal(ci, ref.col_idx) and np.array_equal(va, ref.vals)
    n = N ** 3
    assert K.host_stencil7(N, kind)[0][-1] == 7 * n - 6 * N * N        # nnz = 7n - 6N^2


def test_variable_coefficient_operator_is_a_symmetric_m_matrix():
    """kind "varcoef" (DESIGN.md section 5): per-edge weights in [0.5, 1.5) from splitmix64, a_ij = a_ji = -w, diagonal = sum of the
    six incident weights (1.0 for a neighbour outside the box): symmetric, weakly diagonally dominant, strictly at the faces, SPD;
    every row differs from every other (nothing for a value dictionary or a row-pattern table to find)."""
    import scipy.sparse as sp
    N = 7
    rp, ci, va = K.host_stencil7(N, "varcoef")
    m = sp.csr_matrix((va, ci, rp), shape=(N ** 3, N ** 3))
    assert abs(m - m.T).max() == 0.0
    off = m - sp.diags(m.diagonal())
    assert off.data.max() < -0.5 + 1e-15 and off.data.min() >= -1.5
    slack = m.diagonal() - np.asarray(abs(off).sum(axis=1)).ravel()
    assert slack.min() >= -1e-12 and slack.max() <= 3.0 + 1e-12 and (slack > 0.5).sum() == N ** 3 - (N - 2) ** 3
    assert np.linalg.eigvalsh(m.toarray()).min() > 0.0
    assert len(np.unique(va)) >= 4 * N ** 3 - 3 * N * N - 10                   # ~ one value per edge plus one per row


def test_partition_rows_and_halo_plan():
    N, P = 8, 4
    offs = K.partition_rows(N ** 3, P, N * N)
    assert list(offs) == [0, 128, 256, 384, 512]
    offs3 = K.partition_rows(N ** 3, 3, N * N)
    assert offs3[0] == 0 and offs3[-1] == N ** 3 and all(o % (N * N) == 0 for o in offs3)
    for r in range(P):
        rp, ci, _ = K.host_stencil7(N, "poisson", int(offs[r]) // (N * N), int(offs[r + 1]) // (N * N))
        counts, cols = K.halo_recv_plan(r, P, offs, rp, ci)
        exp_lo = np.arange(offs[r] - N * N, offs[r]) if r > 0 else np.array([], dtype=np.int64)
        exp_hi = np.arange(offs[r + 1], offs[r + 1] + N * N) if r < P - 1 else np.array([], dtype=np.int64)
        assert np.array_equal(cols, np.concatenate([exp_lo, exp_hi]))
        assert counts.sum() == len(cols) and counts[r] == 0


def test_compute_fails_loudly_without_gpu():
    import subprocess, sys
    # in a child: a GPU box must not run this assertion path
    code = ("import kryst_amd as K\n"
            "try:\n    K.Context(0)\n    print('HAVE_GPU')\n"
            "except K.KError as e:\n    print('KERROR', e.code)\n")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=ROOT).stdout
    assert "HAVE_GPU" in out or "KERROR 100" in out, out


def test_cpp_mirror_compiles_against_the_abi(tmp_path):
    """include/kryst_hip.hpp (the C++ mirror of MatVec / Preconditioner / LinearSolver) + its test program build and link."""
    import subprocess
    out = tmp_path / "test_mirror"
    r = subprocess.run(["g++", "-std=c++17", "-Wall", "-I" + os.path.join(ROOT, "include"),
                        os.path.join(ROOT, "tests", "cpp", "test_mirror.cpp"), "-o", str(out),
                        "-L" + os.path.join(ROOT, "kryst_amd", "lib"), "-lkryst_hip", "-Wl,-rpath,/opt/rocm/lib"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]


def test_matrix_market_reader(tmp_path):
    """kryst_host_read_matrix_market (host only) against scipy.io.mmread on general / symmetric / skew / pattern / integer files,
    a file with duplicates and an unsupported one."""
    import scipy.io, scipy.sparse as sp
    import kryst_amd as K
    rng = np.random.default_rng(5)
    g = sp.random(37, 23, density=0.2, random_state=1, format="coo")
    s = sp.random(19, 19, density=0.3, random_state=2, format="coo"); s = (s + s.T).tocoo()
    cases = {"general": (g, {}), "symmetric": (s, {"symmetry": "symmetric"}),
             "pattern": (g, {"field": "pattern"}), "integer": ((g * 100).astype(np.int64), {"field": "integer"})}
    for name, (m, kw) in cases.items():
        path = tmp_path / f"{name}.mtx"
        scipy.io.mmwrite(str(path), m, **kw)
        ref = sp.csr_matrix(scipy.io.mmread(str(path)))
        ref.sort_indices()
        nr, nc, rp, ci, va = K.read_matrix_market(path)
        assert (nr, nc) == ref.shape and np.array_equal(rp, ref.indptr) and np.array_equal(ci, ref.indices)
        assert np.array_equal(va, ref.data.astype(np.float64)), name
    (tmp_path / "skew.mtx").write_text("%%MatrixMarket matrix coordinate real skew-symmetric\n% comment\n3 3 2\n2 1 1.5\n3 1 -2\n")
    nr, nc, rp, ci, va = K.read_matrix_market(tmp_path / "skew.mtx")
    dense = np.zeros((3, 3))
    for i in range(3):
        dense[i, ci[rp[i]:rp[i + 1]]] = va[rp[i]:rp[i + 1]]
    assert np.array_equal(dense, [[0, -1.5, 2.0], [1.5, 0, 0], [-2.0, 0, 0]])
    (tmp_path / "dup.mtx").write_text("%%MatrixMarket matrix coordinate real general\n2 2 3\n1 1 1.0\n2 2 4.0\n1 1 0.5\n")
    nr, nc, rp, ci, va = K.read_matrix_market(tmp_path / "dup.mtx")
    assert list(rp) == [0, 1, 2] and list(ci) == [0, 1] and list(va) == [1.5, 4.0]
    (tmp_path / "arr.mtx").write_text("%%MatrixMarket matrix array real general\n1 1\n1.0\n")
    with pytest.raises(K.KError):
        K.read_matrix_market(tmp_path / "arr.mtx")
    with pytest.raises(K.KError):
        K.read_matrix_market(tmp_path / "missing.mtx")
    # a header that promises more entries than memory holds must come back as an error, not as a C++ exception across the ABI
    (tmp_path / "huge.mtx").write_text("%%MatrixMarket matrix coordinate real general\n3 3 4611686018427387904\n1 1 1.0\n")
    with pytest.raises(K.KError):
        K.read_matrix_market(tmp_path / "huge.mtx")


def test_petsc_binary_reader(tmp_path):
    """kryst_host_read_petsc_binary on a file written the way PETSc's MatView (binary viewer) writes AIJ matrices."""
    import scipy.sparse as sp
    import kryst_amd as K
    m = sp.random(41, 29, density=0.15, random_state=3, format="csr"); m.sort_indices()
    path = tmp_path / "a.petsc"
    with open(path, "wb") as f:
        np.array([1211216, m.shape[0], m.shape[1], m.nnz], dtype=">i4").tofile(f)
        np.diff(m.indptr).astype(">i4").tofile(f)
        m.indices.astype(">i4").tofile(f)
        m.data.astype(">f8").tofile(f)
    nr, nc, rp, ci, va = K.read_petsc_binary(path)
    assert (nr, nc) == m.shape and np.array_equal(rp, m.indptr) and np.array_equal(ci, m.indices) and np.array_equal(va, m.data)
    with open(tmp_path / "vec.petsc", "wb") as f:
        np.array([1211214, 5], dtype=">i4").tofile(f); np.zeros(5, dtype=">f8").tofile(f)     # a Vec, not a Mat
    with pytest.raises(K.KError):
        K.read_petsc_binary(tmp_path / "vec.petsc")
    with open(tmp_path / "short.petsc", "wb") as f:
        np.array([1211216, 3, 3, 4, 2, 1], dtype=">i4").tofile(f)
    with pytest.raises(K.KError):
        K.read_petsc_binary(tmp_path / "short.petsc")
