"""One rank of the shared-GPU multi-rank test (tests/test_gpu_z_multirank_shim.py): P processes on ONE GPU run the library's
real distributed code path with tests/shim/librccl_shim.so standing in for RCCL.  Results are written to
<outdir>/rank<r>.npz; rank 0 compares them with the partition-aware oracle bit for bit."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import kryst_amd as K                      # noqa: E402


def random_system(n):
    """A symmetric, strictly diagonally dominant random sparse operator (about 7 entries per row, couplings of any
    distance: every rank needs halo entries from every other one); the same on every rank and in the test."""
    import scipy.sparse as sp
    rng = np.random.default_rng(1234 + n)
    b = sp.random(n, n, density=3.0 / n, random_state=rng, data_rvs=lambda k: rng.uniform(-1.0, 1.0, k)).tocsr()
    m = (b + b.T).tocsr()
    m.setdiag(0.0); m.eliminate_zeros()
    m = (m + sp.diags(np.asarray(abs(m).sum(axis=1)).ravel() + 1.0)).tocsr()
    m.sort_indices()
    return m


_T0 = time.time()
THREADED = False


def stage(rank, name):
    """Flushed stage marker: when a rank dies, the last marker in its log names the stage."""
    print(f"[rank {rank} +{time.time() - _T0:7.3f}s] {name}", flush=True)


def mixed_system(N):
    """The N^3 Poisson operator plus three symmetric couplings between row 0 and rows n-1, n-3, n-7: the last rank then sends a
    NON-contiguous list to rank 0 while every other send list is a contiguous run -- ranks that disagree on `send_contiguous`
    (the early halo start must be switched off on all of them alike).  Same on every rank and in the test."""
    import scipy.sparse as sp
    rp, ci, va = K.host_stencil7(N, "poisson")
    n = N ** 3
    m = sp.csr_matrix((va, ci, rp), shape=(n, n)).tolil()
    for j, w in ((n - 1, -0.25), (n - 3, -0.5), (n - 7, -0.125)):
        m[0, j] = w; m[j, 0] = w
        m[0, 0] += abs(w); m[j, j] += abs(w)
    m = m.tocsr(); m.sort_indices()
    return m


def main():
    ranks, P, outdir, N = [int(r) for r in sys.argv[1].split(",")], int(sys.argv[2]), sys.argv[3], int(sys.argv[4])
    kind = sys.argv[5]
    if len(ranks) == 1:
        return run_rank(ranks[0], P, outdir, N, kind)
    # several ranks of ONE process, a host thread each (a GPU box admits at most 6 processes on its card: 8 ranks = 4 x 2)
    import threading
    global THREADED
    THREADED = True
    errs = []

    def guarded(r):
        try:
            run_rank(r, P, outdir, N, kind)
        except BaseException as e:                      # noqa: BLE001  (the process must exit non-zero, with the stage in its log)
            import traceback
            traceback.print_exc()
            errs.append((r, e))
            sys.stdout.flush(); sys.stderr.flush()
            os._exit(3)                                 # the sibling would otherwise wait for this rank inside a collective

    ts = [threading.Thread(target=guarded, args=(r,)) for r in ranks]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    if errs:
        sys.exit(3)


def run_rank(rank, P, outdir, N, kind):
    stage(rank, "start")
    idfile = os.path.join(outdir, "uid.bin")
    if rank == 0:
        uid = K.Context.unique_id()
        with open(idfile + ".tmp", "wb") as f:
            f.write(uid)
        os.rename(idfile + ".tmp", idfile)
    else:
        for _ in range(6000):
            if os.path.exists(idfile):
                break
            time.sleep(0.01)
        uid = open(idfile, "rb").read()
    stage(rank, "ctx_create_dist")
    ctx = K.Context(0, rank, P, uid)
    # what the library chose by itself (mailboxes when they work on every rank), then the RCCL all-gather path: the reference results below
    # are made with RCCL + RCCL, the hipIpc forms are compared with them further down
    default_scalar = ctx.scalar_reduce("query")
    assert ctx.scalar_reduce("rccl") == "rccl"
    stage(rank, "operator")
    if kind in ("random", "mixed"):                   # general operator: halo entries from any rank / one non-contiguous send list
        m = random_system(N) if kind == "random" else mixed_system(N)
        N = m.shape[0]
        offs = K.partition_rows(N, P, 1) if kind == "random" else K.partition_rows(N, P, round(N ** (2 / 3)))
        sub = m[int(offs[rank]):int(offs[rank + 1])].tocsr()
        sub.sort_indices()
        a = K.CsrMatrix.from_csr_dist(ctx, N, offs, sub.indptr, sub.indices, sub.data)
    else:
        a = K.CsrMatrix.stencil7(N, kind, ctx=ctx)    # device generator or (KRYST_STENCIL_HOST=1) kryst_csr_create_dist
    nloc = a.nrows()
    default_halo = a.halo_mode("query")
    assert a.halo_mode("rccl") == "rccl"
    stage(rank, f"spmv nloc={nloc} encoding={a.encoding()[0]} defaults: scalar {default_scalar}, halo {default_halo}")
    b = a.spmv(ctx.vec(nloc).fill(1.0))
    out = {"b": b.to_host(), "nloc": np.array([nloc]), "default_scalar_ipc": np.array([1 if default_scalar == "ipc" else 0]),
           "default_halo_peer": np.array([1 if default_halo == "peer" else 0])}
    stage(rank, "norm")
    bn = K.norm(b)
    out["bnorm"] = np.array([bn])
    stage(rank, "all_reduce")
    assert ctx.all_reduce(float(rank + 1)) == P * (P + 1) / 2
    stage(rank, "jacobi_setup")
    pcj = K.Jacobi().setup(a)
    light = os.environ.get("KRYST_MR_LIGHT") == "1"       # larger grids (many tiles per rank): CG, Jacobi-PCG, BiCGStab and the session only
    runs = [("cg", K.CgSolver(1e-9, 300), None), ("pcg", K.PcgSolver(1e-9, 300), pcj),
            ("bicgstab", K.BiCgStabSolver(1e-9 * bn, 300), None),
            ("gmres", K.GmresSolver(10, 1e-9, 40).with_preconditioning(K.Preconditioning.Left), pcj),
            ("fgmres", K.FgmresSolver(1e-9, 40, 12), pcj), ("cgs", K.CgsSolver(1e-9, 60), None),
            ("tfqmr", K.TfqmrSolver(1e-9, 30), None)]
    if light:
        runs = runs[:3]
    for name, s, pc in runs:
        stage(rank, "solve " + name)
        x = ctx.vec(nloc)
        st = s.solve(a, pc, b, x)
        out[name + "_x"] = x.to_host()
        out[name + "_hist"] = np.array(s.residual_history)
        out[name + "_stats"] = np.array([st.iterations, float(st.converged), st.final_residual])
    # Chebyshev needs the halo exchange inside a preconditioner; GMRES-right and the right-preconditioned BiCGStab extension
    stage(rank, "chebyshev_setup")
    cheb = K.ChebyshevPc(3, 1.0, 11.5).setup(a)
    extra = [("gmres_cheb", K.GmresSolver(8, 1e-9, 32).with_preconditioning(K.Preconditioning.Left), cheb),
             ("gmres_right", K.GmresSolver(8, 1e-9, 40).with_preconditioning(K.Preconditioning.Right), pcj),
             ("gmres_ltb", K.GmresSolver(8, 1e-9, 40).with_preconditioning(K.Preconditioning.LeftTextbook), pcj),      # the labelled extension (side 3)
             ("bicg_rpc", K.BiCgStabRightPcSolver(1e-9 * bn, 120), pcj)]
    if light:
        extra = []
    for name, s, pc in extra:
        stage(rank, "solve " + name)
        x = ctx.vec(nloc)
        st = s.solve(a, pc, b, x)
        out[name + "_x"] = x.to_host()
        out[name + "_hist"] = np.array(s.residual_history)
        out[name + "_stats"] = np.array([st.iterations, float(st.converged), st.final_residual])
    # the ILU family factors each rank's own diagonal block (halo columns dropped): its apply
    r = ctx.vec(nloc).fill_splitmix(7)
    out["ilu_r"] = r.to_host()
    for nm, mk in (() if light else (("true", K.TrueIlu0), ("compat", K.Ilu0), ("ilup0", lambda: K.Ilup(0)), ("ilup1", lambda: K.Ilup(1)),
                                     ("ilut", lambda: K.Ilut(4, 1e-3)))):
        stage(rank, "ilu setup " + nm)
        pc_ = mk().setup(a)
        stage(rank, "ilu apply " + nm)
        z_ = pc_.apply(r)
        stage(rank, "ilu download " + nm)
        out["ilu_z_" + nm] = z_.to_host()
        del pc_, z_
    # a stepping session like bench.py's
    stage(rank, "session")
    x = ctx.vec(nloc)
    sess = K.Session("cg", a, None, b, x, tol=0.0, max_iters=25)
    sess.step(5); sess.step(20)
    st = sess.end()
    out["sess_stats"] = np.array([st.iterations, float(st.converged), st.final_residual])
    out["sess_x"] = x.to_host()
    # the mailbox path of the scalar all-reduce (kryst_ctx_scalar_reduce: hipIpc-mapped cells written and polled by the kernel that
    # finishes the local fold): every solver again, bit for bit the results of the RCCL all-gather path above
    stage(rank, "scalar_reduce ipc")
    mode = ctx.scalar_reduce("ipc")
    out["ipc_active"] = np.array([1 if mode == "ipc" else 0])
    if mode != "ipc":
        from kryst_amd._ffi import lib as _lib
        print(f"[rank {rank}] scalar_reduce('ipc') fell back to {mode}: {_lib().kryst_hip_last_error().decode()}", flush=True)
    if mode == "ipc":
        for name, s, pc in runs + extra:
            stage(rank, "ipc solve " + name)
            s.clear_history()                                   # (residual_history accumulates over solves, like the reference's Vec)
            x = ctx.vec(nloc)
            st = s.solve(a, pc, b, x)
            assert np.array_equal(x.to_host(), out[name + "_x"]), ("ipc", name)
            assert np.array_equal(np.array(s.residual_history), out[name + "_hist"]), ("ipc", name)
            assert (st.iterations, float(st.converged), st.final_residual) == tuple(out[name + "_stats"]), ("ipc", name)
        stage(rank, "ipc session")
        x = ctx.vec(nloc)
        with K.Session("cg", a, None, b, x, tol=0.0, max_iters=25) as sess:
            sess.step(5); sess.step(20)
            sess.end()
        assert np.array_equal(x.to_host(), out["sess_x"]), "ipc session"
        assert ctx.scalar_reduce("rccl") == "rccl"
    # the halo exchange by direct peer stores (kryst_csr_halo_mode: push kernel into the neighbours' hipIpc-mapped landing buffers, epoch
    # stamps, pull kernel in front of the boundary tiles): every solver again on the RCCL scalar path and once more with BOTH hipIpc paths
    # together, bit for bit the results above
    stage(rank, "halo_mode peer")
    hmode = a.halo_mode("peer")
    if THREADED:                 # rank threads that share one device are refused (dist.cpp: ipc_map_peers) -- on every rank alike
        assert hmode == "rccl", hmode
    out["peer_active"] = np.array([1 if hmode == "peer" else 0])
    if hmode == "peer":
        b2 = a.spmv(ctx.vec(nloc).fill(1.0))
        assert np.array_equal(b2.to_host(), out["b"]), "peer spmv"
        for both in (False, True):
            if both and ctx.scalar_reduce("ipc") != "ipc":
                break
            for name, s, pc in runs + extra:
                stage(rank, ("peer+ipc solve " if both else "peer solve ") + name)
                s.clear_history()
                x = ctx.vec(nloc)
                st = s.solve(a, pc, b, x)
                xh, hh = x.to_host(), np.array(s.residual_history)
                if not np.array_equal(xh, out[name + "_x"]) or not np.array_equal(hh, out[name + "_hist"]):      # say what differs before failing
                    bad = np.flatnonzero(xh != out[name + "_x"])
                    hb = np.flatnonzero(hh[:len(out[name + "_hist"])] != out[name + "_hist"][:len(hh)])
                    print(f"[rank {rank}] PEER MISMATCH {name} both={both}: {len(bad)} of {len(xh)} x entries differ (first {bad[:4]}), history lengths "
                          f"{len(hh)} / {len(out[name + '_hist'])}, first differing history entry {hb[:1]}", flush=True)
                assert np.array_equal(xh, out[name + "_x"]), ("peer", both, name)
                assert np.array_equal(hh, out[name + "_hist"]), ("peer", both, name)
                assert (st.iterations, float(st.converged), st.final_residual) == tuple(out[name + "_stats"]), ("peer", both, name)
            stage(rank, "peer session")
            x = ctx.vec(nloc)
            with K.Session("cg", a, None, b, x, tol=0.0, max_iters=25) as sess:
                sess.step(5); sess.step(20)
                sess.end()
            assert np.array_equal(x.to_host(), out["sess_x"]), ("peer session", both)
        ctx.scalar_reduce("rccl")
        # messages pushed from the second stream instead of the compute stream (the form large planes take)
        os.environ["KRYST_HALO_INLINE_BYTES"] = "0"
        stage(rank, "peer solve cg (push on the second stream)")
        s = runs[0][1]; s.clear_history()
        x = ctx.vec(nloc)
        s.solve(a, None, b, x)
        assert np.array_equal(x.to_host(), out["cg_x"]), "peer, second stream"
        os.environ.pop("KRYST_HALO_INLINE_BYTES", None)
        assert a.halo_mode("rccl") == "rccl"
    stage(rank, "barrier")
    ctx.barrier()
    np.savez(os.path.join(outdir, f"rank{rank}.npz"), **out)
    print(f"RANK_OK {rank}")


if __name__ == "__main__":
    main()
