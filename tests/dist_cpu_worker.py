"""One rank of the CPU (gloo) rehearsal of the row-partitioned Krylov path.

Every rank owns a k-slab of a 7-point grid and uses the PRODUCT's host-side planning code
(kryst_host_partition_rows, kryst_host_stencil7, kryst_host_halo_recv_plan -- the same functions
kryst_csr_create_dist / kryst_csr_create_stencil7 run before touching the GPU) plus gloo in the place of RCCL:
  * send lists are learnt by exchanging the recv lists (what csr_create_dist does with ncclSend/ncclRecv),
  * each SpMV exchanges halo values, then multiplies the local rows (local column numbering: owned -> c - lo,
    halo -> nloc + slot),
  * each inner product = local value in the library's tile order, all-gather, fold in rank order.
The distributed CG / PCG must reproduce the serial oracle run with the same rank partition of the dots bit for bit,
and the residual history of the serial strict-fold oracle to 1e-12 relative to the initial residual (north_star's
"residuals matching CPU reference to 1e-12 relative").
"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import kryst_amd as K                      # noqa: E402  (host-only helpers; no GPU call is made)
from oracle import oracle as O             # noqa: E402


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    N, kind, method = int(sys.argv[1]), sys.argv[2], sys.argv[3]
    dist.init_process_group("gloo", rank=rank, world_size=world)
    T, V, F = K.reduce_spec()
    n = N ** 3
    offs = K.partition_rows(n, world, N * N)
    lo, hi = int(offs[rank]), int(offs[rank + 1])
    nloc = hi - lo
    rp, ci, va = K.host_stencil7(N, kind, lo // (N * N), hi // (N * N))
    counts, cols = K.halo_recv_plan(rank, world, offs, rp, ci)
    # --- send side: every owner learns which rows the others need
    allreq = [None] * world
    dist.all_gather_object(allreq, (counts.tolist(), cols.tolist()))
    send_rows = {}
    for p in range(world):
        if p == rank:
            continue
        pc, pcols = allreq[p]
        off = sum(pc[:rank])
        send_rows[p] = np.array(pcols[off:off + pc[rank]], dtype=np.int64) - lo
        assert np.all((send_rows[p] >= 0) & (send_rows[p] < nloc))
    recv_off = np.concatenate([[0], np.cumsum(counts)])
    # --- local column numbering
    owned = (ci >= lo) & (ci < hi)
    lcol = np.where(owned, ci - lo, 0)
    if len(cols):
        lcol[~owned] = nloc + np.searchsorted(cols, ci[~owned])
        assert np.array_equal(cols[lcol[~owned] - nloc], ci[~owned])
    a_loc = O.Csr(nloc, nloc + len(cols), rp, lcol, va, check=False)
    rs_loc = O.Reduce.tiled(T, V, F)

    def halo_exchange(x):
        halo = np.zeros(len(cols))
        reqs = []
        bufs = {}
        for p in range(world):
            if p == rank:
                continue
            if len(send_rows[p]):
                reqs.append(dist.isend(torch.from_numpy(np.ascontiguousarray(x[send_rows[p]])), dst=p))
            if counts[p]:
                bufs[p] = torch.empty(int(counts[p]), dtype=torch.float64)
                reqs.append(dist.irecv(bufs[p], src=p))
        for r in reqs:
            r.wait()
        for p, b in bufs.items():
            halo[recv_off[p]:recv_off[p + 1]] = b.numpy()
        return halo

    def spmv(x):
        return a_loc.spmv(np.concatenate([x, halo_exchange(x)]))

    def gdot(u, v):
        mine = torch.tensor([O.dot(u, v, rs_loc)], dtype=torch.float64)
        parts = [torch.empty(1, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(parts, mine)
        tot = float(parts[0][0])
        for p in range(1, world):                      # fold in rank order (rank_fold_kernel)
            tot = tot + float(parts[p][0])
        return tot

    # b = A*1, x0 = 0
    b = spmv(np.ones(nloc))
    x = np.zeros(nloc)
    tol, max_iters = 1e-9, 400
    inv = None
    if method == "pcg":
        dpos = [np.flatnonzero(lcol[rp[i]:rp[i + 1]] == i)[0] + rp[i] for i in range(nloc)]
        inv = 1.0 / va[dpos]
    r = b - spmv(x)
    z = inv * r if inv is not None else r
    p = z.copy()
    rz = gdot(r, z)
    res0 = np.sqrt(abs(rz)) if method == "pcg" else np.sqrt(rz)
    hist = [np.sqrt(gdot(r, r))]
    iters = 0
    for i in range(1, max_iters + 1):
        ap = spmv(p)
        alpha = rz / gdot(p, ap)
        x = x + alpha * p
        r = r - alpha * ap
        z = inv * r if inv is not None else r
        rz_new = gdot(r, z)
        rr = gdot(r, r) if method == "pcg" else rz_new
        res = np.sqrt(rr)
        hist.append(res)
        iters = i
        if res / res0 <= tol or i >= max_iters:
            break
        beta = rz_new / rz
        p = z + beta * p
        rz = rz_new
    # --- compare with the serial oracle (rank 0 gathers x)
    xs = [None] * world
    dist.all_gather_object(xs, x)
    if rank == 0:
        full = O.stencil7(N, kind)
        bfull = full.spmv(np.ones(n))
        rs = O.Reduce.tiled(T, V, F, part_off=offs)
        pcf = O.Pc.jacobi(full) if method == "pcg" else None
        ref = O.solve(method, full, bfull, pc=pcf, tol=tol, max_iters=max_iters, rs=rs)
        xg = np.concatenate(xs)
        assert ref.iterations == iters, (ref.iterations, iters)
        assert np.array_equal(np.array(hist), ref.history), "history differs from the partition-aware oracle"
        assert np.array_equal(xg, ref.x), "solution differs from the partition-aware oracle"
        ser = O.solve(method, full, bfull, pc=pcf, tol=tol, max_iters=max_iters)
        assert ser.iterations == iters
        assert np.max(np.abs(np.array(hist) - ser.history)) <= 1e-12 * ser.history[0]
        print(f"DIST_OK world={world} N={N} {kind} {method} iterations={iters}")
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
