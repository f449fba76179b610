"""A SECOND, independent transcription of the reference's quirkiest code, written directly from the Rust source (not from
oracle/kryst_oracle.c) in plain Python floats (IEEE double, one rounding per operation, no FMA), dense like the reference's own
tests: restarted GMRES with its None / Left / Right branches (src/solver/gmres.rs:216-402, helpers :65-105,:154-192), Ilu0's
setup and apply (src/preconditioner/ilu.rs:59-122), Jacobi (jacobi.rs:53-95), the dense row loop (src/core/wrappers.rs:27-38)
and the serial dot / norm folds (wrappers.rs:101-107,120-126); since round 4 also the three solvers the BASELINE configs run --
CgSolver (cg.rs:114-288, every CgNormType and both Indefinite errors), PcgSolver (pcg.rs:114-222, mixed norms included) and
BiCgStabSolver (bicgstab.rs:69-293, absolute tolerance, silent breakdown breaks) with Convergence::check (convergence.rs:18-34), the
section-8f solvers FgmresSolver::solve_flex (fgmres.rs:114-340), CgsSolver (cgs.rs:58-135) and TfqmrSolver as written
(tfqmr.rs:64-221), and apply_chebyshev (chebyshev.rs:83-159), Ilup::new(p) (ilup.rs:77-167) and Ilut (ilut.rs:80-150).

The C oracle (serial-fold mode) must agree with it BIT FOR BIT -- iteration counts, converged flags, final residuals and every
entry of x -- on the reference's own test systems (gmres.rs:438-528, tests/preconditioner_integration.rs:16-57,126-179) and on
a few seeded dense systems.  Two transcriptions by different routes agreeing to the last bit is the strongest pin the
reference allows: its own tests only hold answers to 1e-8 .. 1e-10."""
import math

import numpy as np
import pytest

from oracle import oracle as O

EPS = 1e-14            # gmres.rs:233


# ------------------------------------------------------------------------------------------------ wrappers.rs
def matvec(a, x):                                   # wrappers.rs:27-38: y[i] = 0; y[i] = y[i] + a[i][j] * x[j], j ascending
    y = []
    for row in a:
        s = 0.0
        for aij, xj in zip(row, x):
            s = s + aij * xj
        y.append(s)
    return y


def dot(x, y):                                      # wrappers.rs:101-107 (no-rayon build): fold(0, acc + x_i * y_i)
    acc = 0.0
    for xi, yi in zip(x, y):
        acc = acc + xi * yi
    return acc


def norm(x):                                        # wrappers.rs:120-126
    acc = 0.0
    for xi in x:
        acc = acc + xi * xi
    return math.sqrt(acc)


# ------------------------------------------------------------------------------------------------ preconditioners
class Jacobi:                                       # jacobi.rs:53-95
    def __init__(self, a):
        n = len(a)
        self.inv = []
        for i in range(n):
            e = [0.0] * n
            e[i] = 1.0
            d = matvec(a, e)[i]                     # diag via A e_i (:56-67)
            self.inv.append(1.0 / d if d != 0.0 else 0.0)      # :69-71

    def apply(self, x):
        return [self.inv[i] * x[i] for i in range(len(x))]      # :84-86


class Ilu0:                                         # ilu.rs:59-122, exactly as written (dense, every quirk included)
    def __init__(self, a):
        n = len(a)
        l = [[0.0] * n for _ in range(n)]
        u = [[0.0] * n for _ in range(n)]
        for i in range(n):
            u[i][i] = a[i][i]                       # :66
            for j in range(i + 1, n):               # :68-72
                if a[i][j] != 0.0:
                    u[i][j] = a[i][j]
            l[i][i] = 1.0                           # :74
            for j in range(i + 1, n):               # :76-80
                if a[j][i] != 0.0:
                    l[j][i] = a[j][i] / u[i][i]
            for j in range(i + 1, n):               # :82-96: "Schur complement update" from the ORIGINAL a, overwriting
                for k in range(i + 1, n):
                    if a[j][k] != 0.0:
                        v = a[j][k] - l[j][i] * u[i][k]
                        if v != 0.0:
                            if k >= j:
                                u[j][k] = v
                            else:
                                l[j][k] = v
        self.l, self.u = l, u

    def apply(self, x):                             # :107-121: forward with L (unit diagonal never divided), backward with U
        y = list(x)                                 # (NO division by u_ii: the reference never divides)
        n = len(x)
        for i in range(n):
            for j in range(i):
                y[i] = y[i] - self.l[i][j] * y[j]
        for i in reversed(range(n)):
            for j in range(i + 1, n):
                y[i] = y[i] - self.u[i][j] * y[j]
        return y


# ------------------------------------------------------------------------------------------------ gmres.rs
def givens(h, g, cs, sn, j):                        # gmres.rs:154-176
    for i in range(j):
        temp = cs[i] * h[i][j] + sn[i] * h[i + 1][j]
        h[i + 1][j] = -sn[i] * h[i][j] + cs[i] * h[i + 1][j]
        h[i][j] = temp
    h_kk, h_k1k = h[j][j], h[j + 1][j]
    r = math.sqrt(h_kk * h_kk + h_k1k * h_k1k)
    if abs(r) < EPS:
        cs[j], sn[j] = 1.0, 0.0
    else:
        cs[j], sn[j] = h_kk / r, h_k1k / r
    h[j][j] = cs[j] * h_kk + sn[j] * h_k1k
    h[j + 1][j] = 0.0
    temp = cs[j] * g[j] + sn[j] * g[j + 1]
    g[j + 1] = -sn[j] * g[j] + cs[j] * g[j + 1]
    g[j] = temp


def back_substitution(h, g, m):                     # gmres.rs:180-192
    y = [0.0] * m
    for i in reversed(range(m)):
        y[i] = g[i]
        for j in range(i + 1, m):
            y[i] = y[i] - h[i][j] * y[j]
        y[i] = y[i] / h[i][i] if abs(h[i][i]) > EPS else 0.0
    return y


def mgs2(z, basis, h, j):                           # the double modified Gram-Schmidt every branch repeats (:83-96,:286-298,:318-330)
    for i in range(j + 1):
        h[i][j] = dot(z, basis[i])
        z = [zk - h[i][j] * bk for zk, bk in zip(z, basis[i])]
    for i in range(j + 1):
        tmp = dot(z, basis[i])
        h[i][j] = h[i][j] + tmp
        z = [zk - tmp * bk for zk, bk in zip(z, basis[i])]
    return z


def gmres(a, pc, side, b, x0, restart, tol, max_iters):     # gmres.rs:216-402
    n = len(b)
    xk = list(x0)
    r0 = [bi - axi for axi, bi in zip(matvec(a, xk), b)]     # :221-227
    beta = norm(r0)
    res0 = beta
    iterations, final_residual, converged = 0, beta, False
    n_outer = -(-max_iters // restart)                       # div_ceil :231
    iteration = 0
    res0_in = res0                                           # what the in-cycle test divides by (the extension below changes it)
    for outer in range(n_outer):
        v, zb = [], []
        r0_norm = beta
        if side == "left_textbook" and pc is not None:       # LABELLED EXTENSION, not in the reference (written from Saad's Alg. 9.4 in the
            z0 = pc.apply(r0)                                # reference's frame, without looking at the C oracle): v0 = M^-1 r0 / ||M^-1 r0||
            r0_norm = norm(z0)
            v.append([zi / r0_norm for zi in z0])
            if outer == 0:
                res0_in = r0_norm
        elif side == "left" and pc is not None:              # :239-246
            v0 = [ri / r0_norm for ri in r0]
            v.append(v0)
            zb.append(pc.apply(v0))
        elif side == "right" and pc is not None:             # :247-260
            z0 = pc.apply(r0)
            r0_norm = norm(z0)
            v0 = [zi / r0_norm for zi in z0]
            v.append(v0)
            zb.append(pc.apply(v0))
            beta = r0_norm
        else:                                                # :261-265
            v.append([ri / r0_norm for ri in r0])
        h = [[0.0] * restart for _ in range(restart + 1)]
        g = [0.0] * (restart + 1)
        g[0] = r0_norm
        cs, sn = [0.0] * restart, [0.0] * restart
        m = 0
        happy = False
        for j in range(restart):
            iteration += 1
            if side == "left" and pc is not None:            # :279-307: orthogonalise against Z, push v_{j+1} into BOTH bases
                z = pc.apply(matvec(a, v[j]))
                z = mgs2(z, zb, h, j)
                h[j + 1][j] = norm(z)
                if abs(h[j + 1][j]) < EPS:
                    happy = True
                    break                                    # leaves the loop BEFORE the rotation and `m = j + 1`
                vj1 = [zi / h[j + 1][j] for zi in z]
                v.append(vj1)
                zb.append(list(vj1))
            elif side == "right" and pc is not None:         # :308-343
                w2 = matvec(a, pc.apply(v[j]))
                w2 = mgs2(w2, v, h, j)
                h[j + 1][j] = norm(w2)
                if abs(h[j + 1][j]) < EPS:
                    happy = True
                    break
                vj1 = [wi / h[j + 1][j] for wi in w2]
                v.append(vj1)
                zb.append(pc.apply(vj1))
            elif side == "left_textbook" and pc is not None:     # extension: z = M^-1 A v_j against V; breakdown handled as arnoldi does
                z = mgs2(pc.apply(matvec(a, v[j])), v, h, j)
                h[j + 1][j] = norm(z)
                if abs(h[j + 1][j]) < EPS:
                    happy = True
                else:
                    v.append([zi / h[j + 1][j] for zi in z])
            else:                                            # arnoldi :65-105: the rotation still runs after a happy breakdown
                w = mgs2(matvec(a, v[j]), v, h, j)
                h[j + 1][j] = norm(w)
                if abs(h[j + 1][j]) < EPS:
                    happy = True
                else:
                    v.append([wi / h[j + 1][j] for wi in w])
            givens(h, g, cs, sn, j)                          # :347
            res_norm = abs(g[j + 1])
            rel = res_norm / res0_in                         # Convergence::check, convergence.rs:18-34 (res0_in is res0 in the reference's arms)
            stop = rel <= tol or iteration >= max_iters
            iterations, final_residual, converged = iteration, res_norm, stop
            m = j + 1
            if (stop and converged) or happy:
                break
        y = back_substitution([row[:m] for row in h[:m]], g[:m], m)      # :358-361
        basis = zb if (side == "right" and pc is not None) else v       # :363-386
        for j in range(m):
            xk = [xi + y[j] * bj for xi, bj in zip(xk, basis[j])]
        r0 = [bi - axi for axi, bi in zip(matvec(a, xk), b)]             # :388-391
        beta = norm(r0)
        final_residual = beta                                            # :393
        converged = beta < tol * res0                                    # :394
        if converged or iteration >= max_iters:
            break
    return xk, iterations, final_residual, converged


# ------------------------------------------------------------------------------------------------ cg.rs / pcg.rs / bicgstab.rs / convergence.rs
# (round 4) The three solvers the BASELINE configs run, transcribed from the Rust source like GMRES above -- without looking at the C oracle.
class Indefinite(Exception):
    """KError::IndefiniteMatrix (code 3) / KError::IndefinitePreconditioner (code 4), src/error.rs:6-19."""
    def __init__(self, code, stats):
        super().__init__(code)
        self.code, self.stats = code, stats


def conv_check(tol, max_iters, res_norm, res0, i):  # convergence.rs:18-34
    rel = res_norm / res0
    converged = (rel <= tol) or (i >= max_iters)
    return converged, (i, res_norm, converged)


def cg(a, b, x0, tol, max_iters, norm_type="unpreconditioned"):      # cg.rs:114-288 (no radius, no obj_target: both default to None, :58-59)
    n = len(b)
    x = list(x0)
    ax = matvec(a, x)
    r = [bi - axi for axi, bi in zip(ax, b)]        # :120-125 `bi - ax`
    p = list(r)                                     # :126
    rsq = dot(r, r)                                 # :127
    res0 = math.sqrt(rsq)                           # :128
    stats = (0, res0, False)                        # :129
    dp = {"preconditioned": dot(r, r), "unpreconditioned": dot(r, r), "natural": dot(r, p), "none": 0.0}[norm_type]   # :131-136
    hist = [math.sqrt(dp)]                          # :140
    for i in range(1, max_iters + 1):               # :141
        ap = matvec(a, p)                           # :143-144
        p_dot_ap = dot(p, ap)                       # :164 (single_reduction false by default; the fused form has the same order)
        if p_dot_ap <= 0.0:                         # :168-174
            raise Indefinite(3, (i, math.sqrt(dot(r, r)), False))
        alpha = rsq / p_dot_ap                      # :175
        x = [xj + alpha * pj for xj, pj in zip(x, p)]          # :217-219
        r = [rj - alpha * apj for rj, apj in zip(r, ap)]       # :220-222
        rsq_new = dot(r, r)                         # :223
        if norm_type in ("preconditioned", "unpreconditioned"):
            res_norm = math.sqrt(rsq_new)           # :225-226
        elif norm_type == "natural":
            res_norm = math.sqrt(abs(dot(r, p)))    # :227 (the OLD p)
        else:
            res_norm = 0.0
        if rsq_new / rsq < 0.0:                     # :254-259
            raise Indefinite(4, (i, res_norm, False))
        hist.append(res_norm)                       # :263
        stop, stats = conv_check(tol, max_iters, res_norm, res0, i)      # :264-265
        if stop and stats[2]:                       # :266-269
            return x, stats, hist
        beta = rsq_new / rsq                        # :270
        p = [rj + beta * pj for pj, rj in zip(p, r)]           # :280-282
        rsq = rsq_new                               # :284
    return x, stats, hist                           # :286-287


def pcg(a, pc, b, x0, tol, max_iters, norm_type="unpreconditioned"):  # pcg.rs:114-222
    x = list(x0)
    ax = matvec(a, x)
    r = [bi - axi for axi, bi in zip(ax, b)]        # :119-124
    z = pc.apply(r) if pc is not None else list(r)  # :126-131
    p = list(z)                                     # :132
    rz = dot(r, z)                                  # :133
    res0 = math.sqrt(abs(rz))                       # :134  (NOT the norm the loop tests with: mixed norms)
    stats = (0, res0, False)
    dp = {"preconditioned": dot(z, z), "unpreconditioned": dot(r, r), "natural": dot(r, z), "none": 0.0}[norm_type]      # :137-142
    hist = [math.sqrt(dp)]                          # :146 (no abs at iteration 0)

    def res_of(r_, z_):                             # :190-195 / :164-169
        if norm_type == "preconditioned":
            return math.sqrt(dot(z_, z_))
        if norm_type == "unpreconditioned":
            return math.sqrt(dot(r_, r_))
        if norm_type == "natural":
            return math.sqrt(abs(dot(r_, z_)))
        return 0.0
    for i in range(max_iters):                      # :147
        ap = matvec(a, p)                           # :149-150
        p_dot_ap = dot(p, ap)                       # :159
        if p_dot_ap <= 0.0:                         # :162-172
            raise Indefinite(3, (i + 1, res_of(r, z), False))
        alpha = rz / p_dot_ap                       # :173
        x = [xj + alpha * pj for xj, pj in zip(x, p)]          # :175-177
        r = [rj - alpha * apj for rj, apj in zip(r, ap)]       # :179-181
        z = pc.apply(r) if pc is not None else list(r)         # :183-187
        rz_new = dot(r, z)                          # :188
        res_norm = res_of(r, z)                     # :190-195
        hist.append(res_norm)                       # :199
        stop, stats = conv_check(tol, max_iters, res_norm, res0, i + 1)  # :200-201
        if stop and stats[2]:                       # :202-205
            return x, stats, hist
        beta = rz_new / rz                          # :206
        if beta < 0.0:                              # :208-213
            raise Indefinite(4, (i + 1, res_norm, False))
        p = [zj + beta * pj for pj, zj in zip(p, z)]           # :215-217
        rz = rz_new                                 # :218
    return x, stats, hist                           # :220-221


def bicgstab(a, b, x0, tol, max_iters):             # bicgstab.rs:69-293 (pc ignored, :70; no residual history in the reference)
    eps = 2.220446049250313e-16                     # T::epsilon()
    x = list(x0)
    ax = matvec(a, x)
    r = [bi - axi for axi, bi in zip(ax, b)]        # :75-77
    r_hat = list(r)                                 # :78
    rho_prev = alpha = omega_prev = 1.0             # :79-81
    v = [0.0] * len(b)                              # :82
    p = list(r)                                     # :83
    res0 = norm(r)                                  # :95
    stats = (0, res0, False)                        # :97
    hist = [res0]                                   # (the port records what the reference computes: res0, then r_norm / s_norm per iteration)
    if res0 <= tol:                                 # :98-102 ABSOLUTE tolerance
        return x, (0, res0, True), hist
    for i in range(1, max_iters + 1):               # :103
        rho = dot(r_hat, r)                         # :115
        if abs(rho) < eps:                          # :117-119 `break`: the stats of the previous iteration stay
            break
        beta = 0.0 if i == 1 else (rho / rho_prev) * (alpha / omega_prev)      # :120-124
        p = [rj + beta * (pj - omega_prev * vj) for pj, rj, vj in zip(p, r, v)]   # :139-141
        v = matvec(a, p)                            # :144-146
        alpha_den = dot(r_hat, v)                   # :159
        if abs(alpha_den) < eps:                    # :161-163
            break
        alpha = rho / alpha_den                     # :164
        s = [rj - alpha * vj for rj, vj in zip(r, v)]          # :174
        s_norm = norm(s)                            # :187
        if s_norm <= tol:                           # :189-206
            x = [xj + alpha * pj for xj, pj in zip(x, p)]
            hist.append(s_norm)
            return x, (i, s_norm, True), hist
        t = matvec(a, s)                            # :208-209
        omega_num = dot(t, s)                       # :221
        omega_den = dot(t, t)                       # :233
        if abs(omega_den) < eps:                    # :235-237
            break
        omega = omega_num / omega_den               # :238
        x = [xj + alpha * pj + omega * sj for xj, pj, sj in zip(x, p, s)]      # :251-253, evaluated left to right
        r = [sj - omega * tj for sj, tj in zip(s, t)]          # :264
        r_norm = norm(r)                            # :278
        stats = (i, r_norm, r_norm <= tol)          # :280
        hist.append(r_norm)
        if r_norm <= tol:                           # :281-284
            return x, stats, hist
        if abs(omega) < eps:                        # :285-287
            break
        rho_prev, omega_prev = rho, omega           # :288-289
    return x, stats, hist                           # :291-292


# ------------------------------------------------------------------------------------------------ the comparison
def dense_csr(a):
    return O.Csr.from_dense(np.array(a, dtype=np.float64))             # every entry stored: the dense row loop, zeros included


def tridiag(n, lo, di, up):
    a = [[0.0] * n for _ in range(n)]
    for i in range(n):
        a[i][i] = di
        if i > 0:
            a[i][i - 1] = lo
        if i + 1 < n:
            a[i][i + 1] = up
    return a


A4 = [[4.0, 1.0, 0.0, 0.0], [1.0, 3.0, 1.0, 0.0], [0.0, 1.0, 2.0, 1.0], [0.0, 0.0, 1.0, 3.0]]        # gmres.rs:441-450
SIDES = {"none": O.SIDE_NONE, "left": O.SIDE_LEFT, "right": O.SIDE_RIGHT, "left_textbook": O.SIDE_LEFT_TEXTBOOK}


def _cases():
    rng = np.random.default_rng(2024)
    out = [("gmres.rs 4x4", A4, [1.0, 2.0, 3.0, 4.0], 4, 1e-10, 100),
           ("integration nonsym tridiag 10", tridiag(10, -1.0, 2.0, 0.5), [1.0] * 10, 10, 1e-12, 100),   # preconditioner_integration.rs:41-57
           ("integration spd tridiag 10", tridiag(10, -1.0, 2.0, -1.0), [1.0] * 10, 10, 1e-12, 100)]
    for n, restart in ((7, 3), (12, 5), (9, 9)):                         # seeded dense systems, diagonally dominant, restarts that bite
        m = rng.uniform(-1.0, 1.0, (n, n))
        m[rng.random((n, n)) < 0.4] = 0.0
        m += np.diag(np.abs(m).sum(axis=1) + 1.0)
        out.append((f"random {n} restart {restart}", m.tolist(), rng.uniform(-2.0, 2.0, n).tolist(), restart, 1e-11, 60))
    return out


@pytest.mark.parametrize("name,a,x_true,restart,tol,max_iters", _cases(), ids=[c[0] for c in _cases()])
@pytest.mark.parametrize("side,pcname", [("none", None), ("left", "jacobi"), ("right", "jacobi"), ("left", "ilu0"), ("right", "ilu0"),
                                         ("left_textbook", "jacobi"), ("left_textbook", "ilu0")])
def test_c_oracle_equals_the_independent_transcription_bitwise(name, a, x_true, restart, tol, max_iters, side, pcname):
    b = matvec(a, x_true)
    pc_py = {None: None, "jacobi": Jacobi, "ilu0": Ilu0}[pcname]
    pc_py = pc_py(a) if pc_py else None
    x_py, its, fin, conv = gmres(a, pc_py, side, b, [0.0] * len(b), restart, tol, max_iters)
    ao = dense_csr(a)
    pc_c = {None: None, "jacobi": O.Pc.jacobi, "ilu0": O.Pc.ilu0_compat}[pcname]
    ref = O.solve("gmres", ao, np.array(b), pc=pc_c(ao) if pc_c else None, tol=tol, max_iters=max_iters, restart=restart,
                  side=SIDES[side], rs=O.SERIAL)
    assert (ref.iterations, ref.converged) == (its, conv), (name, side, pcname)
    assert ref.final_residual == fin, (name, side, pcname, ref.final_residual, fin)
    assert np.array_equal(ref.x, np.array(x_py)), (name, side, pcname, np.max(np.abs(ref.x - np.array(x_py))))


def test_textbook_left_gmres_extension_properties():
    """The labelled extension (kro_gmres side 3; Preconditioning.LeftTextbook on the device) pinned by what it must be, since the reference
    has no such arm: (a) with the identity as preconditioner it IS the unpreconditioned solver, bit for bit (z = w, ||M^-1 r0|| = ||r0||);
    (b) on the reference's own Left + Ilu0 integration case (tests/preconditioner_integration.rs:169-179), where the reference's Left arm
    needs two restart cycles = 20 iterations, it converges inside the first cycle; (c) it solves every case to the tolerance."""
    class Identity:
        def apply(self, r):
            return list(r)
    for name, a, x_true, restart, tol, max_iters in _cases():
        b = matvec(a, x_true)
        x0 = [0.0] * len(b)
        plain = gmres(a, None, "none", b, x0, restart, tol, max_iters)
        ident = gmres(a, Identity(), "left_textbook", b, x0, restart, tol, max_iters)
        assert plain == ident, name
        for pc in (Jacobi(a), Ilu0(a)):
            x, its, fin, conv = gmres(a, pc, "left_textbook", b, x0, restart, tol, max_iters)
            assert conv and norm([bi - axi for axi, bi in zip(matvec(a, x), b)]) <= 10 * tol * norm(b), (name, type(pc).__name__, its, fin)
    a = tridiag(10, -1.0, 2.0, 0.5)
    b = matvec(a, [1.0] * 10)
    _, its_ref, _, conv_ref = gmres(a, Ilu0(a), "left", b, [0.0] * 10, 10, 1e-12, 100)
    _, its_txt, _, conv_txt = gmres(a, Ilu0(a), "left_textbook", b, [0.0] * 10, 10, 1e-12, 100)
    assert conv_ref and its_ref == 20 and conv_txt and its_txt <= 10, (its_ref, its_txt)


def test_ilu0_factors_and_apply_bitwise():
    """Ilu0::setup / apply (ilu.rs:59-122) alone, on matrices where its "Schur complement update from the original a" differs
    visibly from a textbook ILU(0): dense-ish rows, zeros inside the band, an unsymmetric pattern."""
    rng = np.random.default_rng(7)
    mats = [tridiag(10, -1.0, 2.0, 0.5), A4]
    for n in (6, 11):
        m = rng.uniform(-1.0, 1.0, (n, n))
        m[rng.random((n, n)) < 0.5] = 0.0
        m += np.diag(np.abs(m).sum(axis=1) + 1.0)
        mats.append(m.tolist())
    for a in mats:
        pc = Ilu0(a)
        ref = O.Pc.ilu0_compat(dense_csr(a))
        for _ in range(3):
            r = rng.standard_normal(len(a)).tolist()
            assert np.array_equal(ref.apply(np.array(r)), np.array(pc.apply(r)))


def test_reference_expectations_hold_for_the_transcription():
    """The transcription itself satisfies the reference's own assertions (gmres.rs:452-460,484-491,517-527 and
    tests/preconditioner_integration.rs:158-179) -- so a disagreement above could never be blamed on a wrong reading here."""
    b = matvec(A4, [1.0, 2.0, 3.0, 4.0])
    x, _, _, conv = gmres(A4, None, "none", b, [0.0] * 4, 4, 1e-10, 100)
    assert conv and all(abs(xi - ei) < 1e-8 for xi, ei in zip(x, [1.0, 2.0, 3.0, 4.0]))
    x, _, _, conv = gmres(A4, Jacobi(A4), "left", b, [0.0] * 4, 4, 1e-10, 100)
    assert conv and all(abs(xi - ei) < 1e-8 for xi, ei in zip(x, [1.0, 2.0, 3.0, 4.0]))
    x, _, _, _ = gmres(A4, Jacobi(A4), "right", b, [0.0] * 4, 4, 1e-10, 100)
    assert norm([ai - bi for ai, bi in zip(matvec(A4, x), b)]) < 1e-2
    an = tridiag(10, -1.0, 2.0, 0.5)
    bn = matvec(an, [1.0] * 10)
    x, its, _, conv = gmres(an, Ilu0(an), "left", bn, [0.0] * 10, 10, 1e-12, 100)
    rel = math.sqrt(sum((xi - 1.0) ** 2 for xi in x) / 10.0)
    assert conv and rel < 1e-10 and its == 20                            # two cycles: SURVEY 3.3


# ------------------------------------------------------------------------------------------------ CG / PCG / BiCGStab against the C oracle
def poisson3d_dense(N):
    """The build's 7-point Poisson operator on an N^3 grid as a dense list of lists (the reference's solvers see dense rows)."""
    n = N ** 3
    a = [[0.0] * n for _ in range(n)]
    for k in range(N):
        for j in range(N):
            for i in range(N):
                r = i + N * (j + N * k)
                a[r][r] = 6.0
                for d, ok in ((-1, i > 0), (1, i < N - 1), (-N, j > 0), (N, j < N - 1), (-N * N, k > 0), (N * N, k < N - 1)):
                    if ok:
                        a[r][r + d] = -1.0
    return a


def _krylov_cases():
    rng = np.random.default_rng(4242)
    out = [("cg.rs 2x2", [[4.0, 1.0], [1.0, 3.0]], [1.0, 2.0], True),                                      # cg.rs:310-323
           ("cg.rs 3x3", [[4.0, 1.0, 0.0], [1.0, 3.0, 1.0], [0.0, 1.0, 2.0]], [1.0, 2.0, 3.0], True),      # cg.rs:326-356
           ("bicgstab.rs 3x3", [[4.0, 1.0, 0.0], [2.0, 3.0, 1.0], [0.0, 1.0, 2.0]], [1.0, 2.0, 3.0], False),
           ("spd tridiag 10", tridiag(10, -1.0, 2.0, -1.0), matvec(tridiag(10, -1.0, 2.0, -1.0), [1.0] * 10), True),
           ("nonsym tridiag 10", tridiag(10, -1.0, 2.0, 0.5), matvec(tridiag(10, -1.0, 2.0, 0.5), [1.0] * 10), False),
           ("poisson 4^3", poisson3d_dense(4), matvec(poisson3d_dense(4), [1.0] * 64), True)]
    for n in (6, 17, 40):                              # seeded dense systems: SPD (B^T B + I) and diagonally dominant unsymmetric
        bm = rng.uniform(-1.0, 1.0, (n, n))
        spd = (bm.T @ bm + np.eye(n)).tolist()
        out.append((f"random spd {n}", spd, rng.uniform(-2.0, 2.0, n).tolist(), True))
        m = rng.uniform(-1.0, 1.0, (n, n))
        m[rng.random((n, n)) < 0.5] = 0.0
        m += np.diag(np.abs(m).sum(axis=1) + 1.0)
        out.append((f"random nonsym {n}", m.tolist(), rng.uniform(-2.0, 2.0, n).tolist(), False))
    return out


NORMS = {"preconditioned": O.NORM_PRECONDITIONED, "unpreconditioned": O.NORM_UNPRECONDITIONED, "natural": O.NORM_NATURAL, "none": O.NORM_NONE}


def _same(ref, x, stats, hist, what):
    assert (ref.iterations, ref.converged) == (stats[0], stats[2]), (what, ref, stats)
    assert ref.final_residual == stats[1], (what, ref.final_residual, stats[1])
    assert np.array_equal(ref.x, np.array(x)), (what, np.max(np.abs(ref.x - np.array(x))))
    assert len(ref.history) == len(hist) and np.array_equal(ref.history, np.array(hist)), what


@pytest.mark.parametrize("name,a,b,spd", _krylov_cases(), ids=[c[0] for c in _krylov_cases()])
def test_cg_pcg_bicgstab_of_the_c_oracle_equal_the_independent_transcription_bitwise(name, a, b, spd):
    """CgSolver (cg.rs:114-288, every CgNormType), PcgSolver (pcg.rs:114-222; no pc, Jacobi, Ilu0; every norm type) and BiCgStabSolver
    (bicgstab.rs:69-293) of the C oracle in serial-fold mode against the transcription above: iteration counts, flags, final residual, the
    whole residual history and every entry of x, bit for bit -- on the reference's own test systems, tridiagonal systems, the build's 4^3
    Poisson operator and seeded dense systems, with tolerances that are met, iteration caps that bite (`converged = true` at the cap,
    convergence.rs:25) and tol = 0."""
    ao = dense_csr(a)
    n = len(b)
    x0 = [0.0] * n
    for tol, cap in ((1e-10, 200), (1e-6, 3), (0.0, 7)):
        if spd:
            for nt in NORMS:
                x, st, hist = cg(a, b, x0, tol, cap, nt)
                _same(O.solve("cg", ao, np.array(b), tol=tol, max_iters=cap, norm_type=NORMS[nt], rs=O.SERIAL), x, st, hist, (name, "cg", nt, tol, cap))
            for pcname, pc_py, pc_c in ((None, None, None), ("jacobi", Jacobi, O.Pc.jacobi), ("ilu0", Ilu0, O.Pc.ilu0_compat)):
                for nt in NORMS:
                    try:
                        x, st, hist = pcg(a, pc_py(a) if pc_py else None, b, x0, tol, cap, nt)
                        code = 0
                    except Indefinite as e:                               # (Ilu0 as written is not symmetric: beta < 0 can happen)
                        code, st = e.code, e.stats
                    ref = O.solve("pcg", ao, np.array(b), pc=pc_c(ao) if pc_c else None, tol=tol, max_iters=cap, norm_type=NORMS[nt], rs=O.SERIAL,
                                  raise_on_error=False)
                    assert ref.code == code, (name, "pcg", pcname, nt, ref.code, code)
                    if code == 0:
                        _same(ref, x, st, hist, (name, "pcg", pcname, nt, tol, cap))
                    else:
                        assert (ref.iterations, ref.final_residual, ref.converged) == st, (name, "pcg", pcname, nt)
        bn = norm(b)
        x, st, hist = bicgstab(a, b, x0, tol * bn, cap)                    # ABSOLUTE tolerance (bicgstab.rs:98,189,281)
        _same(O.solve("bicgstab", ao, np.array(b), tol=tol * bn, max_iters=cap, rs=O.SERIAL), x, st, hist, (name, "bicgstab", tol, cap))


def test_indefinite_matrix_and_breakdown_paths_bitwise():
    """cg.rs:168-174 / pcg.rs:162-172 (p.Ap <= 0 -> KError::IndefiniteMatrix with the stats of that iteration) and BiCGStab's silent
    `break`s (bicgstab.rs:117-119: rho = 0 with r_hat orthogonal to r after one step of a rotation-like operator): same codes, same stats."""
    a = [[1.0, 0.0, 0.0], [0.0, -2.0, 0.0], [0.0, 0.0, 3.0]]
    b = [1.0, 1.0, 1.0]
    ao = dense_csr(a)
    for solver, method in ((lambda: cg(a, b, [0.0] * 3, 1e-10, 50), "cg"), (lambda: pcg(a, None, b, [0.0] * 3, 1e-10, 50), "pcg")):
        try:
            solver()
            raised = None
        except Indefinite as e:
            raised = e
        ref = O.solve(method, ao, np.array(b), tol=1e-10, max_iters=50, rs=O.SERIAL, raise_on_error=False)
        assert raised is not None and ref.code == raised.code
        assert (ref.iterations, ref.final_residual, ref.converged) == raised.stats, (method, ref, raised.stats)
    rot = [[0.0, 1.0], [-1.0, 0.0]]                                        # A r is orthogonal to r: alpha_den = <r_hat, A r> = 0 at i = 1
    br = [1.0, 0.0]
    x, st, hist = bicgstab(rot, br, [0.0, 0.0], 1e-12, 20)
    _same(O.solve("bicgstab", dense_csr(rot), np.array(br), tol=1e-12, max_iters=20, rs=O.SERIAL), x, st, hist, "bicgstab breakdown")


# ------------------------------------------------------------------------------------------------ chebyshev.rs / ilup.rs / ilut.rs
def chebyshev_t(m, x):                              # chebyshev.rs:143-159
    if m == 0:
        return 1.0
    if m == 1:
        return x
    t0, t1 = 1.0, x
    for _ in range(2, m + 1):
        t2 = 2.0 * x * t1 - t0
        t0, t1 = t1, t2
    return t1


def apply_chebyshev(a, r, alpha, beta, m):          # chebyshev.rs:83-140
    if abs(beta - alpha) < 2.220446049250313e-16:   # :88-92
        return list(r)
    v0 = list(r)
    c = (beta + alpha) / 2.0                        # :99
    d = (beta - alpha) / 2.0                        # :100
    tau = 1.0 / chebyshev_t(m, (0.0 - c) / d)       # :102
    v1 = matvec(a, v0)                              # :104
    v1 = [(v1[i] - c * v0[i]) / d for i in range(len(r))]      # :105-107
    if m == 0:                                      # :108-111 (after the wasted matvec)
        return v0
    if m == 1:                                      # :112-116: v1 UNSCALED by tau
        return v1
    for _k in range(2, m + 1):                      # :118-125
        v2 = matvec(a, v1)
        v2 = [(2.0 * (v2[i] - c * v1[i]) / d) - v0[i] for i in range(len(r))]
        v0, v1 = v1, v2
    return [tau * x for x in v1]                    # :127-139


class Ilup:                                         # ilup.rs:77-167, dense work arrays exactly as written
    def __init__(self, a, fill):
        n = len(a)
        UMAX = 2 ** 64 - 1
        level = [[0 if a[i][j] != 0.0 else UMAX for j in range(n)] for i in range(n)]      # :84-91
        w = [list(row) for row in a]                # :93-98
        self.l = [([], []) for _ in range(n)]
        self.u = [([], []) for _ in range(n)]
        for i in range(n):                          # :100
            for j in range(i):                      # :102
                if w[i][j] != 0.0 and level[i][j] <= fill:          # :103
                    u_jj = w[j][j]
                    if u_jj == 0.0:
                        raise ZeroDivisionError(j)  # :106-108 KError::SolveError
                    lij = w[i][j] / u_jj            # :109
                    self.l[i][0].append(j); self.l[i][1].append(lij)
                    for k in range(j + 1, n):       # :113
                        if w[j][k] != 0.0:
                            new_level = min(UMAX, min(UMAX, level[i][j] + level[j][k]) + 1)      # saturating adds :115
                            if new_level <= fill:
                                w[i][k] = w[i][k] - lij * w[j][k]               # :117-118
                                level[i][k] = min(level[i][k], new_level)       # :119
            for k in range(i, n):                   # :126-131
                if w[i][k] != 0.0 and level[i][k] <= fill:
                    self.u[i][0].append(k); self.u[i][1].append(w[i][k])

    def apply(self, r):                             # :138-167 (shared with Ilut, ilut.rs:121-150)
        return tri_apply(self.l, self.u, r)


def tri_apply(l, u, r):
    n = len(r)
    y = [0.0] * n
    for i in range(n):
        s = r[i]
        for j, v in zip(*l[i]):
            s = s - v * y[j]
        y[i] = s
    z = [0.0] * n
    for i in reversed(range(n)):
        s = y[i]
        for j, v in zip(*u[i]):
            if j > i:
                s = s - v * z[j]
        z[i] = s / u[i][1][u[i][0].index(i)] if i in u[i][0] else s
    return z


class Ilut:                                         # ilut.rs:80-117: no elimination at all -- drop, keep the `fill` largest, split
    def __init__(self, a, fill, droptol):
        n = len(a)
        self.l, self.u = [], []
        for i in range(n):
            row = [(j, a[i][j]) for j in range(n) if a[i][j] != 0.0]           # :88-94
            row = [(j, v) for j, v in row if abs(v) >= droptol]                # :96
            if len(row) > fill:                                                # :98-101 (sort_by is stable; descending magnitude)
                row = sorted(row, key=lambda e: -abs(e[1]))[:fill]
            self.l.append(([j for j, _ in row if j < i], [v for j, v in row if j < i]))
            self.u.append(([j for j, _ in row if j >= i], [v for j, v in row if j >= i]))

    def apply(self, r):
        return tri_apply(self.l, self.u, r)


def test_chebyshev_ilup_ilut_of_the_c_oracle_equal_the_independent_transcription_bitwise():
    """apply_chebyshev (chebyshev.rs:83-159: degenerate interval, m = 0 after a wasted matvec, m = 1 unscaled, m >= 2 scaled by tau),
    Ilup::new(p) (ilup.rs:77-167: level-of-fill bookkeeping with saturating adds, fill 0 .. 3) and Ilut::new(fill, droptol)
    (ilut.rs:80-150: no elimination, stable magnitude sort) of the C oracle against the transcriptions above, bit for bit."""
    rng = np.random.default_rng(99)
    mats = [tridiag(10, -1.0, 2.0, 0.5), A4, poisson3d_dense(3)]
    for n in (8, 14):
        m = rng.uniform(-1.0, 1.0, (n, n))
        m[rng.random((n, n)) < 0.6] = 0.0
        m += np.diag(np.abs(m).sum(axis=1) + 1.0)
        mats.append(m.tolist())
    for a in mats:
        ao = dense_csr(a)
        n = len(a)
        r = rng.standard_normal(n).tolist()
        for m_deg in (0, 1, 2, 3, 6):
            for lo, hi in ((0.5, 7.5), (1.0, 1.0)):
                assert np.array_equal(O.apply_chebyshev(ao, np.array(r), lo, hi, m_deg), np.array(apply_chebyshev(a, r, lo, hi, m_deg))), (n, m_deg, lo, hi)
            assert O.chebyshev_t(m_deg, -1.37) == chebyshev_t(m_deg, -1.37)
        sparse = O.Csr.from_dense(np.array(a, dtype=np.float64), keep_zeros=False)
        for fill in (0, 1, 2, 3):
            assert np.array_equal(O.Pc.ilup(sparse, fill).apply(np.array(r)), np.array(Ilup(a, fill).apply(r))), (n, "ilup", fill)
        for fill, droptol in ((2, 0.0), (4, 1e-3), (3, 0.3), (100, 0.0)):
            assert np.array_equal(O.Pc.ilut(sparse, fill, droptol).apply(np.array(r)), np.array(Ilut(a, fill, droptol).apply(r))), (n, "ilut", fill, droptol)


# ------------------------------------------------------------------------------------------------ cgs.rs / tfqmr.rs
def cgs(a, b, x0, tol, max_iters):                  # cgs.rs:58-135 (pc ignored, :59)
    eps = 2.220446049250313e-16
    n = len(b)
    x = list(x0)
    ax = matvec(a, x)
    r = [bi - axi for bi, axi in zip(b, ax)]        # :64-69
    r_tld = list(r)                                 # :70
    p = list(r)                                     # :71
    q = [0.0] * n                                   # :72
    u = [0.0] * n                                   # :73
    rho = dot(r_tld, r)                             # :74
    rho_old = 0.0
    res0 = norm(r)                                  # :76
    stats = (0, res0, False)
    hist = []                                       # (the port records res_norm per iteration; nothing at iteration 0)
    for i in range(1, max_iters + 1):               # :78
        if abs(rho) < eps:                          # :80-82
            break
        if i == 1:                                  # :83-86
            u = list(r)
            p = list(u)
        else:
            beta = rho / rho_old                    # :88
            q_old, p_old = list(q), list(p)
            u = [rj + beta * qo for rj, qo in zip(r, q_old)]                       # :93-95
            p = [uj + beta * (qo + beta * po) for uj, qo, po in zip(u, q_old, p_old)]   # :97-99
        v = matvec(a, p)                            # :102-104
        alpha = rho / dot(r_tld, v)                 # :106
        q = [uj - alpha * vj for uj, vj in zip(u, v)]          # :108-110
        x = [xj + alpha * (uj + qj) for xj, uj, qj in zip(x, u, q)]    # :112-114  `*xj += alpha * (u + q)`
        upq = [uj + qj for uj, qj in zip(u, q)]     # :116-119
        w = matvec(a, upq)                          # :120-121
        r = [rj - alpha * wj for rj, wj in zip(r, w)]          # :122-124
        res_norm = norm(r)                          # :125
        hist.append(res_norm)
        stop, stats = conv_check(tol, max_iters, res_norm, res0, i)    # :127-128
        if stop and stats[2]:                       # :129-132
            return x, stats, hist
        rho_old = rho                               # :133
        rho = dot(r_tld, r)                         # :134
    return x, stats, hist


def tfqmr(a, b, tol, max_iters):                    # tfqmr.rs:64-221, as written (x0 is discarded, :72; the reference's only test of it is #[ignore])
    n = len(b)
    x = [0.0] * n                                   # :72
    r = list(b)                                     # :75
    r_tld = list(r)                                 # :77
    rho = dot(r, r_tld)                             # :80
    if rho == 0.0:                                  # :81-83
        return x, (0, norm(r), True), []
    w = list(r); y = list(r)                        # :96-97  (w is never read again)
    u = [0.0] * n; d = [0.0] * n                    # :98-99
    psi_old = eta_old = 0.0                         # :100-101
    tau = norm(r)                                   # :102
    res0 = tau                                      # :103
    stats = (0, res0, False)                        # :104
    hist = []
    if tau == 0.0:                                  # :105-107
        return x, (0, 0.0, True), hist
    dpold = tau                                     # :109
    for k in range(1, max_iters + 1):               # :110
        v = matvec(a, y)                            # :112-114
        sigma = dot(r_tld, v)                       # :117
        if sigma == 0.0 or not math.isfinite(sigma):        # :118-123
            return x, (k, norm(r), False), hist
        alpha = rho / sigma                         # :124
        if alpha == 0.0 or not math.isfinite(alpha):        # :125-130
            return x, (k, norm(r), False), hist
        u = [ri - alpha * vi for ri, vi in zip(r, v)]       # :133-135
        q = [ui - alpha * vi for ui, vi in zip(u, v)]       # :138-141
        t = [ui + qi for ui, qi in zip(u, q)]       # :144-147
        au = matvec(a, t)                           # :148-149
        r = [ri - alpha * aui for ri, aui in zip(r, au)]    # :151-153
        dp = norm(r)                                # :154
        tau_m0 = math.sqrt(dp * dpold)              # :155
        tau_local = tau_m0                          # :156
        for m in range(2):                          # :158
            if m == 0:
                norm_u_m, tau_for_m, u_m = dp, tau_m0, u            # :159-161,:164
            else:
                norm_u_m, tau_for_m, u_m = norm(q), tau_local, q    # :162-164
            psi = norm_u_m / tau_for_m              # :167
            c_m = 1.0 / math.sqrt(1.0 + psi * psi)  # :168
            eta = c_m * c_m * alpha                 # :169
            cf = 0.0 if (alpha == 0.0 or k == 1) else psi_old * psi_old * eta_old / alpha      # :172-176
            d = [umi + cf * di for umi, di in zip(u_m, d)]          # :177-179
            x = [xi + eta * di for xi, di in zip(x, d)]             # :182-184
            dpest = math.sqrt(float(2 * k + m + 2)) * tau_for_m      # :187
            stop, stats = conv_check(tol, max_iters, dpest, res0, k)     # :188-189
            hist.append(dpest)
            psi_old, eta_old = psi, eta             # :190-191
            tau_local = tau_for_m * psi * c_m       # :192
            if stop:                                # :193-198
                return x, (k, dpest, True), hist
        r = list(u)                                 # :205
        rho_new = dot(r_tld, r)                     # :206
        beta = rho_new / rho                        # :207
        rho = rho_new                               # :208
        y = [ui + beta * (qi + beta * yi) for ui, qi, yi in zip(u, q, y)]       # :212
        dpold = dp                                  # :214
    return x, (max_iters, norm(r), stats[2]), hist   # :217-219 (converged keeps the last check's value)


@pytest.mark.parametrize("name,a,b,spd", _krylov_cases(), ids=[c[0] for c in _krylov_cases()])
def test_cgs_tfqmr_of_the_c_oracle_equal_the_independent_transcription_bitwise(name, a, b, spd):
    """CgsSolver (cgs.rs:58-135) and TfqmrSolver exactly as written (tfqmr.rs:64-221) of the C oracle, serial-fold mode, against the
    transcriptions above: stats, residual estimates and x bit for bit."""
    ao = dense_csr(a)
    n = len(b)
    for tol, cap in ((1e-10, 60), (1e-4, 3), (0.0, 6)):
        x, st, hist = cgs(a, b, [0.0] * n, tol, cap)
        _same(O.solve("cgs", ao, np.array(b), tol=tol, max_iters=cap, rs=O.SERIAL), x, st, hist, (name, "cgs", tol, cap))
        x, st, hist = tfqmr(a, b, tol, cap)
        _same(O.solve("tfqmr", ao, np.array(b), tol=tol, max_iters=cap, rs=O.SERIAL), x, st, hist, (name, "tfqmr", tol, cap))


# ------------------------------------------------------------------------------------------------ fgmres.rs
def fgmres(a, pc, b, x0, tol, max_iters, restart, orthog="classical", haptol=1e-12, preallocate=False):     # fgmres.rs:114-340
    n = len(b)
    x = list(x0)
    tmp = matvec(a, x)
    r = [ri - ai for ri, ai in zip(b, tmp)]         # :133-138
    beta = norm(r)                                  # :139
    if beta == 0.0:                                 # :140-142
        return x, (0, 0.0, True), []
    nb = max_iters if preallocate else restart      # :144-165 (sizes only matter through `m` below)
    v = [None] * (nb + 1 + 16)
    z = [None] * (nb + 16)
    h = [[0.0] * max(nb, restart) for _ in range(nb + 1 + 16)]
    cs = [0.0] * (nb + 16); sn = [0.0] * (nb + 16)
    sv = [0.0] * (nb + 1 + 16)
    sv[0] = beta                                    # :166
    v[0] = [ri / beta for ri in r]                  # :167-169
    total = 0
    res_norm_outer = beta                           # :171 (never updated: the stats' final_residual at :339)
    stats = (0, res_norm_outer, False)
    hist = []
    while total < max_iters:                        # :175
        m = min(max_iters, restart) if preallocate else min(restart, max_iters - total)      # :202
        converged = False
        steps = m
        for j in range(m):                          # :206
            z[j] = list(v[j])                       # :208
            if pc is not None:
                z[j] = pc.apply(v[j])               # :209-211
            w = matvec(a, z[j])                     # :213-214
            hcol = [0.0] * (j + 2)
            for i in range(j + 1):                  # :219-221 / :229-231: ALL dots first (classical in both arms)
                hcol[i] = dot(w, v[i])
            for i in range(j + 1):                  # :222-226 / :232-236
                w = [wi - hcol[i] * vi for wi, vi in zip(w, v[i])]
            if orthog == "modified":                # :238-245
                for i in range(j + 1):
                    corr = dot(w, v[i])
                    if abs(corr) > 1e-10:
                        w = [wi - corr * vi for wi, vi in zip(w, v[i])]
            h[j + 1][j] = norm(w)                   # :248
            for i in range(j + 1):
                h[i][j] = hcol[i]                   # :249
            hapbnd = haptol * abs(sv[j])            # :251
            if not (abs(h[j + 1][j]) < hapbnd):     # :252-259
                wn = h[j + 1][j]
                v[j + 1] = [wi / wn for wi in w]
            else:
                v[j + 1] = [0.0] * n
            for i in range(j):                      # :261-265
                temp = cs[i] * h[i][j] + sn[i] * h[i + 1][j]
                h[i + 1][j] = -sn[i] * h[i][j] + cs[i] * h[i + 1][j]
                h[i][j] = temp
            h1, h2 = h[j][j], h[j + 1][j]           # :267-276
            denom = math.sqrt(h1 * h1 + h2 * h2)
            c, s_ = (1.0, 0.0) if denom == 0.0 else (h1 / denom, h2 / denom)
            cs[j], sn[j] = c, s_
            temp = c * sv[j] + s_ * sv[j + 1]       # :279-281
            sv[j + 1] = -s_ * sv[j] + c * sv[j + 1]
            sv[j] = temp
            h[j][j] = c * h[j][j] + s_ * h[j + 1][j]    # :282-283
            h[j + 1][j] = 0.0
            res_norm = abs(sv[j + 1])               # :284
            total += 1                              # :285
            hist.append(res_norm)                   # :290
            stop, stats = conv_check(tol, max_iters, res_norm, sv[0], total)     # :291 (against the ROTATED s[0])
            if stop:                                # :293-299
                stats = (total, res_norm, stats[2])
                steps = j + 1
                converged = True
                break
        k = steps                                   # :302-311: no pivot guard
        y = [0.0] * k
        for i in reversed(range(k)):
            acc = sv[i]
            for l in range(i + 1, k):
                acc = acc - h[i][l] * y[l]
            y[i] = acc / h[i][i]
        for i, yi in enumerate(y):                  # :349-353
            x = [xi + yi * zi for xi, zi in zip(x, z[i])]
        tmp = matvec(a, x)                          # :314-319
        r_new = [ri - ai for ri, ai in zip(b, tmp)]
        rn = norm(r_new)                            # :320
        if rn < tol or converged:                   # :321-326  (ABSOLUTE tol for the true residual)
            stats = (total, rn, True)
            break
        beta = rn                                   # :328-331
        v[0] = [ri / beta for ri in r_new]
        sv = [0.0] * (restart + 1 + 16)             # :332-334
        sv[0] = beta
    return x, (total, res_norm_outer, stats[2]), hist       # :339-340


class FixedJacobi:                                  # a FlexiblePreconditioner that happens not to change (preconditioner/mod.rs:16-19)
    def __init__(self, a):
        self.inv = [1.0 / a[i][i] for i in range(len(a))]

    def apply(self, x):
        return [d * xi for d, xi in zip(self.inv, x)]


@pytest.mark.parametrize("name,a,b,spd", _krylov_cases(), ids=[c[0] for c in _krylov_cases()])
def test_fgmres_of_the_c_oracle_equals_the_independent_transcription_bitwise(name, a, b, spd):
    """FgmresSolver::solve_flex (fgmres.rs:114-340) of the C oracle in serial-fold mode against the transcription above: classical and
    modified orthogonalisation, restarts that bite, preallocate, with and without a (Jacobi) flexible preconditioner; the quirks are the
    reference's (convergence against the rotated s[0], the INITIAL residual as final_residual, no pivot guard)."""
    ao = dense_csr(a)
    n = len(b)
    for pc_py, pc_c in ((None, None), (FixedJacobi(a), O.Pc.jacobi(ao))):
        for orth, oc in (("classical", 0), ("modified", 1)):
            for tol, cap, restart, pre in ((1e-10, 60, 5, False), (1e-6, 7, 3, False), (1e-9, 12, 4, True)):
                x, st, hist = fgmres(a, pc_py, b, [0.0] * n, tol, cap, restart, orth, 1e-12, pre)
                ref = O.solve("fgmres", ao, np.array(b), pc=pc_c, tol=tol, max_iters=cap, restart=restart, orthog=oc, haptol=1e-12, preallocate=pre,
                              rs=O.SERIAL, raise_on_error=False)
                if not all(math.isfinite(v_) for v_ in x):              # (a happy breakdown without convergence ends in NaNs in the reference too)
                    assert not np.all(np.isfinite(ref.x)), (name, orth, tol, cap, restart, pre)
                    continue
                _same(ref, x, st, hist, (name, "fgmres", orth, tol, cap, restart, pre, pc_py is not None))


def test_approx_inverse_apply_bitwise():
    """ApproxInv::apply (approxinv.rs:268-298): y_i = sum over the row's stored (j, m_ij) in stored order of m_ij * x_j, from 0.0 --
    here with rows whose entries are NOT in ascending column order (inv_rows keeps whatever order setup produced)."""
    rng = np.random.default_rng(3)
    n = 9
    rows = []
    for i in range(n):
        cols = rng.permutation(n)[: rng.integers(0, 6)].tolist()
        rows.append([(int(j), float(rng.uniform(-2.0, 2.0))) for j in cols])
    x = rng.standard_normal(n).tolist()
    y = []
    for row in rows:
        acc = 0.0
        for j, mij in row:
            acc = acc + mij * x[j]
        y.append(acc)
    ptr = np.cumsum([0] + [len(r) for r in rows])
    col = np.array([j for r in rows for j, _ in r], dtype=np.int64)
    val = np.array([v for r in rows for _, v in r])
    m = O.Csr(n, n, ptr, col, val, check=False)                            # stored order kept: unsorted columns are the point
    assert np.array_equal(O.Pc.approx_inverse(m).apply(np.array(x)), np.array(y))
