"""The schedule of the box-stencil wavefront solve (kryst_amd/csrc/tri_box.h), restated in a few lines of Python and checked exhaustively on small
boxes -- no GPU: every row's 13 possible dependencies are where the kernel looks for them, early enough.

A workgroup owns a parallelogram of 8 x 8 grid lines: lane (jl, kl) of block (J, K) walks line jj = 8 J + jl - kl, kk = 8 K + kl, row ii at step
t = ii + 2 (jl + kl) + 1.  A dependency (ii + di, jj + dj, kk + dk), (dk, dj, di) lexicographically negative, must be
  * in the same block, on the lane the kernel permutes from -- (jl-1, kl) for (dj, dk) = (-1, 0); (jl, kl-1) for (+1, -1); (jl-1, kl-1) for
    (0, -1); (jl-2, kl-1) for (-1, -1); the lane itself for (0, 0) -- and finished at least 1 / 1 / 3 / 5 / 1 steps (for di = +1) before; or
  * one of the 25 external lines of a block that is dispatched earlier (J + 2 K smaller): lines 0-7 = lanes (7, e) of block (J-1, K), 8-14 =
    its lanes (6, e - 8), 15-24 = lines jj = 8 J - 1 + (e - 15) of row kk = 8 K - 1, i.e. lanes (6, 7), (7, 7) of block (J, K-1) and (0..7, 7) of
    block (J+1, K-1) -- and the poller's first-use step r + sigma(e) for its row r is not later than the step the lane asks for it."""
import itertools

import pytest

LOWER = [(dk, dj, di) for dk in (-1, 0, 1) for dj in (-1, 0, 1) for di in (-1, 0, 1) if (dk, dj, di) < (0, 0, 0)]
NEIGHBOUR_LANE = {(-1, 0): (-1, 0, 1), (1, -1): (0, -1, 1), (0, -1): (-1, -1, 3), (-1, -1): (-2, -1, 5), (0, 0): (0, 0, 1)}   # (dj, dk) -> (d jl, d kl, age of row ii + 1)


def block_and_lane(jj, kk):
    K, kl = divmod(kk, 8)
    J, jl = divmod(jj + kl, 8)
    return J, K, jl, kl


def external_line(J, K, e):
    """(jj, kk, sigma) of external line e of block (J, K) -- tri_box.h, the poller."""
    if e < 8:
        return 8 * J - 1 - e, 8 * K + e, 2 * e
    if e < 15:
        return 8 * J - 2 - (e - 8), 8 * K + (e - 8), 2 * (e - 8) + 2
    m = e - 15
    return 8 * J - 1 + m, 8 * K - 1, (0 if m <= 1 else 2 * m - 4)


def external_column(jl, kl, which):
    """the ring column lane (jl, kl) reads for neighbour line `which` (1: (jj-1, kk), 2: (jj+1, kk-1), 3: (jj, kk-1), 4: (jj-1, kk-1)); -1: a lane of its own wave"""
    if which == 1:
        return kl if jl == 0 else -1
    if which == 2:
        return 17 + jl if kl == 0 else -1
    if which == 3:
        return 16 + jl if kl == 0 else (kl - 1 if jl == 0 else -1)
    return 15 + jl if kl == 0 else (7 + kl if jl == 0 else (kl - 1 if jl == 1 else -1))


@pytest.mark.parametrize("dims", [(5, 9, 9), (4, 17, 10), (3, 8, 25), (6, 23, 1), (3, 3, 3)])
def test_every_dependency_is_where_the_kernel_looks_for_it(dims):
    Ni, Nj, Nk = dims
    nbj, nbk = (Nj + 6) // 8 + 1, (Nk + 7) // 8
    for kk, jj in itertools.product(range(Nk), range(Nj)):
        J, K, jl, kl = block_and_lane(jj, kk)
        assert 0 <= J < nbj and 0 <= K < nbk and 8 * J + jl - kl == jj and 8 * K + kl == kk
        skew = 2 * (jl + kl) + 1
        for dk, dj, di in LOWER:
            j2, k2 = jj + dj, kk + dk
            if not (0 <= j2 < Nj and 0 <= k2 < Nk):
                continue                                    # no such row: the stream holds +0.0 there
            which = {(-1, 0): 1, (1, -1): 2, (0, -1): 3, (-1, -1): 4, (0, 0): 0}[(dj, dk)]
            J2, K2, jl2, kl2 = block_and_lane(j2, k2)
            d_jl, d_kl, age = NEIGHBOUR_LANE[(dj, dk)]
            if (J2, K2) == (J, K):
                # a lane of the same wave, and the one the kernel permutes from; its row ii + 1 is `age` steps old when this lane is at row ii
                assert (jl2, kl2) == (jl + d_jl, kl + d_kl), (dims, jj, kk, dj, dk)
                assert which == 0 or 2 * (jl2 + kl2) + 1 == skew - age - 1      # row ii + 1 of that line was finished at step t - age
                assert external_column(jl, kl, which) == -1 if which else True
            else:
                # another block: dispatched earlier, one of the three the poller gates on, and the line is the external line the lane reads
                assert (J2, K2) in ((J - 1, K), (J, K - 1), (J + 1, K - 1)), (dims, jj, kk, dj, dk)
                assert J2 + 2 * K2 < J + 2 * K
                e = external_column(jl, kl, which)
                assert e >= 0, (dims, jj, kk, dj, dk)
                ej, ek, sigma = external_line(J, K, e)
                assert (ej, ek) == (j2, k2), (dims, jj, kk, dj, dk, e)
                # the lane asks for row ii + 1 of that line at step ii + skew; the poller has it from step (ii + 1) + sigma on
                assert sigma <= skew - 1
                # ... and the producing lane is one that writes through: jl >= 6 or kl == 7
                assert jl2 >= 6 or kl2 == 7
    # the hyperplane order the kernels rely on: every lower offset of the cube lies on an earlier plane i + 2 j + 4 k
    assert all(di + 2 * dj + 4 * dk <= -1 for dk, dj, di in LOWER)
