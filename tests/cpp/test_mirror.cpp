// The reference's own solver / preconditioner tests, re-written against the C++ mirror (include/kryst_hip.hpp) so that
// they read like the originals (file:line of each original given).  Runs on the GPU through the C ABI.
#include <cassert>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include "kryst_hip.hpp"

using namespace kryst;

#define REQUIRE(cond) do { if (!(cond)) { std::fprintf(stderr, "REQUIRE failed: %s (%s:%d)\n", #cond, __FILE__, __LINE__); std::exit(1); } } while (0)

// dense rows -> CSR with every entry stored (reproduces the reference's dense row loop, wrappers.rs:31-36)
static HipCsrMatrix dense(const std::vector<std::vector<double>>& d) {
    const size_t n = d.size(), m = d[0].size();
    std::vector<size_t> rp(n + 1), ci; Vec va;
    for (size_t i = 0; i < n; ++i) { for (size_t j = 0; j < m; ++j) { ci.push_back(j); va.push_back(d[i][j]); } rp[i + 1] = ci.size(); }
    return HipCsrMatrix::from_csr(n, m, rp, ci, va);
}
static HipCsrMatrix tridiag(size_t n, double lo, double di, double up) {       // tests/preconditioner_integration.rs:16-57
    std::vector<size_t> rp(n + 1), ci; Vec va;
    for (size_t i = 0; i < n; ++i) {
        if (i > 0) { ci.push_back(i - 1); va.push_back(lo); }
        ci.push_back(i); va.push_back(di);
        if (i + 1 < n) { ci.push_back(i + 1); va.push_back(up); }
        rp[i + 1] = ci.size();
    }
    return HipCsrMatrix::from_csr(n, n, rp, ci, va);
}
static double rel_error(const Vec& x, const Vec& t) {                           // :60-64
    double num = 0, den = 0;
    for (size_t i = 0; i < x.size(); ++i) { num += (x[i] - t[i]) * (x[i] - t[i]); den += t[i] * t[i]; }
    return std::sqrt(num / den);
}

int main() {
    {   // src/matrix/sparse.rs:121-144
        auto m = HipCsrMatrix::from_csr(3, 3, {0, 1, 2, 3}, {0, 1, 2}, {1.0, 1.0, 1.0});
        Vec x{2.0, 3.0, 5.0}, y(3, 0.0);
        m.spmv(x, y);
        REQUIRE(y == x);
        auto m2 = HipCsrMatrix::from_csr(2, 3, {0, 2, 4}, {0, 1, 1, 2}, {1.0, 2.0, 3.0, 4.0});
        Vec y2(2, 0.0);
        m2.spmv(Vec{1.0, 1.0, 1.0}, y2);
        REQUIRE((y2 == Vec{3.0, 7.0}));
        bool threw = false;
        try { HipCsrMatrix::from_csr(2, 2, {0, 2, 3}, {1, 0, 1}, {1.0, 2.0, 3.0}); } catch (const KError& e) { threw = e.code == KError::CsrError; }
        REQUIRE(threw);                                          // new_checked rejects unsorted columns
    }
    {   // src/solver/cg.rs:310-323 cg_solves_simple_spd  +  :359-379 single-reduction equivalence
        auto a = dense({{4.0, 1.0}, {1.0, 3.0}});
        Vec b{1.0, 2.0}, x{0.0, 0.0};
        CgSolver solver(1e-10, 20);
        auto stats = solver.solve(a, nullptr, b, x);
        const Vec expected{0.09090909090909091, 0.6363636363636364};
        for (size_t i = 0; i < 2; ++i) REQUIRE(std::fabs(x[i] - expected[i]) < 1e-8);
        REQUIRE(stats.converged);
        Vec xs{0.0, 0.0};
        CgSolver single(1e-10, 20); single.with_single_reduction(true);
        REQUIRE(single.solve(a, nullptr, b, xs).converged);
        for (size_t i = 0; i < 2; ++i) REQUIRE(std::fabs(x[i] - xs[i]) < 1e-8);
    }
    {   // src/solver/pcg.rs:253-275 with IdentityPC
        auto a = dense({{4.0, 1.0}, {1.0, 3.0}});
        IdentityPC pc; pc.setup(a);
        Vec b{1.0, 2.0}, x{0.0, 0.0};
        PcgSolver s(1e-10, 20);
        REQUIRE(s.solve(a, &pc, b, x).converged);
        REQUIRE(std::fabs(x[0] - 0.09090909090909091) < 1e-8 && std::fabs(x[1] - 0.6363636363636364) < 1e-8);
    }
    {   // src/solver/gmres.rs:439-528
        auto a = dense({{4, 1, 0, 0}, {1, 3, 1, 0}, {0, 1, 2, 1}, {0, 0, 1, 3}});
        const Vec x_true{1, 2, 3, 4};
        Vec b(4); a.matvec(x_true, b);
        Vec x(4, 0.0);
        GmresSolver solver(4, 1e-10, 100);
        REQUIRE(solver.solve(a, nullptr, b, x).converged);
        for (size_t i = 0; i < 4; ++i) REQUIRE(std::fabs(x[i] - x_true[i]) < 1e-8);
        Jacobi pc; pc.setup(a);
        Vec x2(4, 0.0);
        GmresSolver s2(4, 1e-10, 100);
        REQUIRE(s2.solve(a, &pc, b, x2).converged);
        for (size_t i = 0; i < 4; ++i) REQUIRE(std::fabs(x2[i] - x_true[i]) < 1e-8);
        Vec x3(4, 0.0), ax(4);
        GmresSolver s3(4, 1e-10, 100); s3.with_preconditioning(Preconditioning::Right);
        s3.solve(a, &pc, b, x3);
        a.matvec(x3, ax);
        double rn = 0; for (size_t i = 0; i < 4; ++i) rn += (ax[i] - b[i]) * (ax[i] - b[i]);
        REQUIRE(std::sqrt(rn) < 1e-2);                           // gmres.rs:521-527: convergence not asserted
    }
    {   // src/solver/fgmres.rs:532-552 fgmres_equiv_to_gmres_on_fixed_pc
        auto a = dense({{2.0, 1.0}, {1.0, 3.0}});
        const Vec x_true{1.0, 2.0};
        Vec b(2); a.matvec(x_true, b);
        Jacobi pc; pc.setup(a);
        Vec x(2, 0.0);
        FgmresSolver fg(1e-10, 100, 25);
        auto st = fg.solve_flex(a, &pc, b, x);
        REQUIRE(st.converged);
        for (size_t i = 0; i < 2; ++i) REQUIRE(std::fabs(x[i] - x_true[i]) < 1e-6);
        Vec xm(2, 0.0);
        FgmresSolver fm(1e-10, 100, 25); fm.with_orthog(Orthog::Modified).with_haptol(1e-14);
        REQUIRE(fm.solve_flex(a, nullptr, b, xm).converged && std::fabs(xm[0] - 1.0) < 1e-6 && std::fabs(xm[1] - 2.0) < 1e-6);
    }
    {   // src/solver/cgs.rs:155-188 cgs_solves_large_well_conditioned_nonsym
        auto a = dense({{10, 2, 0, 0, 0}, {3, 15, 4, 0, 0}, {0, -2, 8, 1, 0}, {0, 0, 1, 7, 3}, {0, 0, 0, 2, 12}});
        const Vec x_true{1.0, 2.0, 3.0, 4.0, 5.0};
        Vec b(5); a.matvec(x_true, b);
        Vec x(5, 0.0);
        CgsSolver cgs(1e-10, 200);
        REQUIRE(cgs.solve(a, nullptr, b, x).converged);
        for (size_t i = 0; i < 5; ++i) REQUIRE(std::fabs(x[i] - x_true[i]) <= 1e-6);
        TfqmrSolver tf(1e-10, 3);                                // tfqmr.rs:241 is #[ignore]: only the call shape is exercised
        Vec xt(5, 7.0);
        auto st = tf.solve(a, nullptr, b, xt);
        // (as written the recurrence breaks down or stops at the cap on this system, depending on the dot association)
        REQUIRE(st.iterations == 3 && tf.residual_history.size() >= 4 && tf.residual_history.size() <= 5);
        REQUIRE(std::fabs(tf.residual_history[0] - 286.00529883) < 1e-6);
    }
    {   // src/solver/bicgstab.rs:303-328
        std::vector<std::vector<double>> d(3, std::vector<double>(3));
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) d[i][j] = (i == j) ? 4.0 : (double)(i + 2 * j) + 1.0;
        auto a = dense(d);
        const Vec x_true{1.0, 2.0, 3.0};
        Vec b(3); a.matvec(x_true, b);
        Vec x(3, 0.0);
        BiCgStabSolver solver(1e-10, 100);
        auto stats = solver.solve(a, nullptr, b, x);
        for (size_t i = 0; i < 3; ++i) REQUIRE(std::fabs(x[i] - x_true[i]) < 1e-8);
        REQUIRE(stats.converged);
    }
    {   // tests/preconditioner_integration.rs:127-138 spd_jacobi_pcg_converges, :156-164, :169-179
        const size_t n = 10;
        auto a = tridiag(n, -1.0, 2.0, -1.0);
        const Vec x_true(n, 1.0);
        Vec b(n); a.matvec(x_true, b);
        Jacobi pc; pc.setup(a);
        PcgSolver solver(1e-12, n);
        Vec x(n, 0.0);
        auto stats = solver.solve(a, &pc, b, x);
        REQUIRE(stats.converged && rel_error(x, x_true) < 1e-10 && stats.iterations <= n);
        auto an = tridiag(n, -1.0, 2.0, 0.5);
        Vec bn(n); an.matvec(x_true, bn);
        Vec xn(n, 0.0);
        GmresSolver g(10, 1e-12, 100);
        REQUIRE(g.solve(an, nullptr, bn, xn).converged && rel_error(xn, x_true) < 1e-10);
        Ilu0 ilu; ilu.setup(an);
        Vec xl(n, 0.0);
        GmresSolver gl(10, 1e-12, 100); gl.with_preconditioning(Preconditioning::Left);
        auto sl = gl.solve(an, &ilu, bn, xl);
        REQUIRE(sl.converged && rel_error(xl, x_true) < 1e-10 && sl.iterations == 20);     // two cycles
    }
    {   // src/preconditioner/approxinv.rs:384-397 / :428-443, apply side: the exact inverse rows of diag(2,3,4) and of the identity
        ApproxInv inv({{{0, 0.5}}, {{1, 1.0 / 3.0}}, {{2, 0.25}}});
        auto a = dense({{2.0, 0.0, 0.0}, {0.0, 3.0, 0.0}, {0.0, 0.0, 4.0}});
        inv.setup(a);
        Vec z(3, 0.0);
        inv.apply(Vec{2.0, 3.0, 4.0}, z);
        REQUIRE(z[0] == 1.0 && z[1] == 1.0 && z[2] == 1.0);
        Vec x(3, 0.0);
        PcgSolver pcg(1e-12, 10);
        REQUIRE(pcg.solve(a, &inv, Vec{2.0, 3.0, 4.0}, x).converged && std::fabs(x[0] - 1.0) < 1e-12 && std::fabs(x[2] - 1.0) < 1e-12);
    }
    {   // src/preconditioner/chebyshev.rs:184-206 and the stub :68-70 ; ilup.rs:202-212
        auto a = dense({{2.0, 0.0}, {0.0, 3.0}});
        Vec z(2, 0.0);
        apply_chebyshev(a, Vec{1.0, 1.0}, z, 2.0, 3.0, 1);
        REQUIRE(std::isfinite(z[0]) && std::isfinite(z[1]));
        Chebyshev c(3, 0.1, 12.0); c.setup(a);
        bool threw = false;
        try { c.apply(Vec{1.0, 1.0}, z); } catch (const KError& e) { threw = e.code == KError::SolveError; }
        REQUIRE(threw);
        auto id = dense({{1.0, 0.0}, {0.0, 1.0}});
        Ilup p(0); p.setup(id);
        Vec zi(2, 0.0);
        p.apply(Vec{2.0, 3.0}, zi);
        REQUIRE(std::fabs(zi[0] - 2.0) < 1e-12 && std::fabs(zi[1] - 3.0) < 1e-12);
    }
    {   // cg.rs:168-174: Err(IndefiniteMatrix), x untouched ; monitor + residual_history (cg.rs:137-140,260-263)
        auto a = dense({{1.0, 0.0}, {0.0, -1.0}});
        Vec x{0.25, 0.5};
        bool threw = false;
        try { CgSolver(1e-10, 10).solve(a, nullptr, Vec{0.0, 1.0}, x); } catch (const KError& e) { threw = e.code == KError::IndefiniteMatrix; }
        REQUIRE(threw && x[0] == 0.25 && x[1] == 0.5);
        auto p = HipCsrMatrix::stencil7(6, 0);
        Vec ones(216, 1.0), b(216), xx(216, 0.0);
        p.matvec(ones, b);
        std::vector<size_t> its;
        CgSolver s(1e-8, 100); s.with_monitor([&](size_t i, double) { its.push_back(i); });
        auto st = s.solve(p, nullptr, b, xx);
        REQUIRE(st.converged && its.size() == st.iterations + 1 && its.front() == 0 && its.back() == st.iterations);
        REQUIRE(s.residual_history.size() == its.size());
    }
    {   // src/context/ksp_context.rs:88-148 (KspContext::solve_context) + pc_context.rs:36-76 (PC<T>) with the constructor it lacks;
        // tests/preconditioner_integration.rs:126-179: tridiag(10) through the factory, every kind on the accelerated path
        auto an = tridiag(10, -1.0, 2.0, -1.0);
        Vec x_true(10, 1.0), bn(10);
        an.matvec(x_true, bn);
        for (SolverKind k : {SolverKind::Cg, SolverKind::Pcg, SolverKind::GmresLeft, SolverKind::GmresRight, SolverKind::Bicgstab, SolverKind::Fgmres}) {
            KspContext ksp(k, an, PC::Jacobi().build(an), k == SolverKind::Bicgstab ? 1e-12 : 1e-10, 200, 10, PC::Jacobi().build(an));
            Vec x(10, 0.0);
            auto st = ksp.solve_context(bn, x);
            // right-preconditioned GMRES as written re-normalises by ||M^-1 r|| and stalls near 1e-6 on this system (the oracle: 103
            // iterations, error 6.9e-6, converged = false; the reference's own test only asks for a residual below 1e-2, gmres.rs:524)
            const double bar = k == SolverKind::GmresRight ? 1e-4 : 1e-6;
            if (rel_error(x, x_true) >= bar) std::fprintf(stderr, "KspContext kind %d: rel. error %.3e after %zu iterations\n", (int)k, rel_error(x, x_true), st.iterations);
            REQUIRE(rel_error(x, x_true) < bar);
            // (the converged flag of right-preconditioned / flexible GMRES and of BiCGStab's breakdown exits follows the
            // reference's own quirks -- gmres.rs:526 "Do not assert stats.converged" -- so only the kinds below assert it)
            if (k == SolverKind::Cg || k == SolverKind::Pcg || k == SolverKind::GmresLeft) REQUIRE(st.converged);
        }
        KspContext ilu(SolverKind::GmresLeft, an, PC::Ilut(10, 1e-3).build(an), 1e-10, 200, 10);
        Vec xi(10, 0.0);
        REQUIRE(ilu.solve_context(bn, xi).converged && rel_error(xi, x_true) < 1e-8);
        bool threw = false;
        try { KspContext q(SolverKind::Qmr, an, nullptr, 1e-8, 10); Vec xq(10, 0.0); q.solve_context(bn, xq); } catch (const KError& e) { threw = e.code == KError::Unsupported; }
        REQUIRE(threw);
        threw = false;
        try { PC{PC::AMGKind}.build(an); } catch (const KError& e) { threw = e.code == KError::Unsupported; }
        REQUIRE(threw);
    }
    {   // KError::ZeroPivot(row) (error.rs:15-16) carries its row; a second solve on a busy context is refused, not corrupted
        auto z = HipCsrMatrix::from_csr(3, 3, {0, 2, 5, 7}, {0, 1, 0, 1, 2, 1, 2}, {2.0, 1.0, 1.0, 0.5, 1.0, 1.0, 3.0});
        TrueIlu0 t;
        bool threw = false;
        try { t.setup(z); } catch (const KError& e) { threw = e.code == KError::ZeroPivot && e.row == 1; }
        REQUIRE(threw);
    }
    std::printf("CPP_MIRROR_OK\n");
    return 0;
}
