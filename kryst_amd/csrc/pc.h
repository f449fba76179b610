// Preconditioner objects (Preconditioner<M,V>, src/preconditioner/mod.rs:8-13).
#pragma once
#include "csr.h"

enum { KR_PC_IDENTITY = 1, KR_PC_JACOBI = 2, KR_PC_ILU = 3, KR_PC_CHEB_STUB = 6, KR_PC_CHEB = 7, KR_PC_SPAI = 9 };

struct kryst_pc_s {
    kryst_ctx_t ctx = nullptr;
    int kind = 0;
    kryst_csr_t a = nullptr;          // borrowed: the operator the factors refer to
    int64_t n = 0;
    double* d_inv_diag = nullptr;     // JACOBI
    // ILU kinds: factor values on A's pattern + level schedule (precond.hip)
    int ilu_mode = 0;
    double* d_lfac = nullptr; double* d_ufac = nullptr;
    int divide_diag = 0;
    int32_t* d_lvl_rows_l = nullptr; int32_t* d_lvl_rows_u = nullptr;   // rows ordered by level
    std::vector<int32_t> lvl_off_l, lvl_off_u;                          // level offsets (host)
    int32_t* d_lvl_off_l = nullptr; int32_t* d_lvl_off_u = nullptr;
    double* d_work = nullptr;
    int32_t* d_sync = nullptr;        // grid-barrier words of the persistent triangular solve
    // Chebyshev
    double cheb_alpha = 0, cheb_beta = 0; int64_t cheb_degree = 0;
    double* d_v0 = nullptr; double* d_v1 = nullptr; double* d_v2 = nullptr;
};

namespace kr {
// z <- M^-1 r on ctx->s_main (device pointers, padded vectors).  `done`: device flag that turns kernels into no-ops.
int32_t pc_apply_dev(kryst_pc_t pc, const double* r, double* z, const int* done);
// Call after the stream has been synchronised: KRYST_OK, or KRYST_SOLVE_ERROR when an apply since the last check was abandoned
// by the device (the ILU wavefront solve's give-up path, tri_wave.h); the preconditioner has then switched itself to kernels
// that cannot stall, pc_fell_back() reports that switch once, and the caller repeats the work.
int32_t pc_health(kryst_pc_t pc);
bool pc_fell_back(kryst_pc_t pc);
int32_t chebyshev_dev(kryst_csr_t a, const double* r, double* z, double alpha, double beta, int64_t m,
                      double* v0, double* v1, double* v2, const int* done);
}
