"""The library's fixed inner-product tree, transcribed a THIRD time -- in numpy, from the specification in include/kryst_hip.h
(kryst_reduce_spec) and DESIGN.md section 4.2, without looking at the HIP kernels (common.h, ew.h) or the C oracle (kryst_oracle.c):

    tile = T * V elements; thread t of a tile folds its V products in index order from 0.0 (elements past n contribute nothing);
    64-lane xor butterfly (v = v + v[lane ^ off], off = 32 .. 1); the tile's T / 64 wave values are folded serially -> one partial
    per tile.  The partials are folded in chunks of F: thread t takes partial c F + t (0.0 past the end), butterfly, serial over the
    F / 64 waves; with more than one chunk, thread t then folds chunk values t, t + F, ... in ascending order from 0.0, and the same
    butterfly / serial fold gives the result.  Rank results are folded in rank order.

The oracle's KRO_REDUCE_TILED mode (which the GPU is compared with bit for bit in the -m gpu tier) must equal it bit for bit: the tree
the parity tests rest on is pinned by its written specification, not only by two implementations agreeing with each other."""
import numpy as np
import pytest

import kryst_amd as K
from oracle import oracle as O


def butterfly(v):
    """v: (..., 64) -> (...,): what lane 0 holds after the xor butterfly (every lane holds the same bits: IEEE addition commutes)."""
    idx = np.arange(64)
    for off in (32, 16, 8, 4, 2, 1):
        v = v + v[..., idx ^ off]
    return v[..., 0]


def block_fold(threads):
    """threads: (nblocks, nthreads) -> (nblocks,): butterfly per 64-lane wave, then the waves serially."""
    nb, nt = threads.shape
    waves = butterfly(threads.reshape(nb, nt // 64, 64))
    s = waves[:, 0].copy()
    for w in range(1, nt // 64):
        s = s + waves[:, w]
    return s


def tiled_dot(x, y, T, V, F, part_off=None):
    if part_off is not None:                                  # several ranks: each slice on its own, then total = r0; total = total + r_p
        parts = [tiled_dot(x[part_off[p]:part_off[p + 1]], y[part_off[p]:part_off[p + 1]], T, V, F) for p in range(len(part_off) - 1)]
        total = parts[0]
        for r in parts[1:]:
            total = total + r
        return total
    n = len(x)
    tile = T * V
    ntiles = max(1, -(-n // tile))
    prod = np.zeros(ntiles * tile)
    prod[:n] = x * y                                          # one rounding per product (no FMA)
    pv = prod.reshape(ntiles, T, V)
    live = (np.arange(ntiles * tile) < n).reshape(ntiles, T, V)
    acc = np.zeros((ntiles, T))
    for e in range(V):                                        # the thread's own elements in index order, from 0.0
        acc = np.where(live[:, :, e], acc + pv[:, :, e], acc)
    partial = block_fold(acc)
    nch = -(-ntiles // F) if ntiles > F else 1
    pad = np.zeros(nch * F)
    pad[:ntiles] = partial
    chunk = block_fold(pad.reshape(nch, F))
    if nch == 1:
        return float(chunk[0])
    th = np.zeros(F)
    for j in range(nch):                                      # thread t: chunks t, t + F, ... ascending, from 0.0
        t = j % F
        th[t] = th[t] + chunk[j]
    return float(block_fold(th.reshape(1, F))[0])


@pytest.mark.parametrize("n", [1, 2, 63, 511, 512, 513, 5000, 70001, 600000, 1100003])
def test_tiled_tree_of_the_oracle_equals_its_specification(n):
    T, V, F = K.reduce_spec()
    assert (T, V, F) == (256, 2, 1024)                        # the published constants (include/kryst_hip.h)
    rng = np.random.default_rng(n)
    x = rng.standard_normal(n) * np.exp(rng.uniform(-8, 8, n))
    y = rng.standard_normal(n)
    rs = O.Reduce.tiled(T, V, F)
    assert O.dot(x, y, rs) == tiled_dot(x, y, T, V, F)
    assert O.norm(x, rs) == float(np.sqrt(tiled_dot(x, x, T, V, F)))
    if n >= 5000:                                             # rank slices: each in its own tree, folded in rank order
        offs = np.array([0, n // 3, n // 3 + 1, n // 2, n], dtype=np.int64)
        assert O.dot(x, y, O.Reduce.tiled(T, V, F, part_off=offs)) == tiled_dot(x, y, T, V, F, offs)
