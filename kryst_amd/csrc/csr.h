// Device CSR operator (CsrMatrix<f64>, src/matrix/sparse.rs:22-46) and the SpMV launcher.
#pragma once
#include "dist.h"

namespace kr {
constexpr int KR_DIA_MAX = 32;     // CSR-DIA: at most this many diagonals (27-point box stencils fit)
constexpr unsigned long long KR_DIA_ABSENT = 0x7FF8D1A0D1A0D1A0ull;   // "no entry here": a quiet NaN whose payload no arithmetic produces
}

struct kryst_csr_s {
    kryst_ctx_t ctx = nullptr;
    int64_t nrows = 0;        // local rows
    int64_t ncols = 0;        // global columns
    int64_t xlen = 0;         // length the SpMV input vector must have (ncols, or local rows when distributed)
    int64_t nnz = 0;          // local
    int32_t* d_row_ptr = nullptr;   // nrows+1 (+pad)
    int32_t* d_col = nullptr;       // nnz (+8 pad, zero): LOCAL column index; >= nloc means halo slot
    double*  d_val = nullptr;       // nnz (+8 pad, zero)
    uint8_t* d_code = nullptr;      // CSR-D8: nnz (+32 pad) codes into d_dict, or nullptr (> 256 distinct col-row offsets)
    int32_t* d_dict = nullptr;      // 256 offsets: col = row + d_dict[code]
    uint16_t* d_code16 = nullptr;   // CSR-D16: nnz (+32 pad) entries  (value code << 8) | offset code, or nullptr
    double* d_vdict = nullptr;      // 256 values: val = d_vdict[code >> 8]  (<= 256 distinct value bit patterns)
    // CSR-P16: one 16-bit pattern id per ROW; a pattern is the row's whole sequence of (col - row, value) pairs
    uint16_t* d_pid = nullptr;      // nrows (+ KR_TILE pad) pattern ids, or nullptr
    uint32_t* d_pmeta = nullptr;    // per pattern, 2 words: (first table entry of its base | base length << 16, presence mask)
    int32_t* d_poff = nullptr;      // table: col - row
    double* d_pval = nullptr;       // table: value
    int32_t npat = 0, ntab = 0, pat_unroll = 8; bool pat_single = false;
    bool pat_far_interior = false;  // ... at least in the rows of interior tiles (generator-made distributed operator)
    int32_t pat_far_lo = 0, pat_far_hi = 0; bool pat_far_uniform = false;   // table positions 0 / 6 are the same offsets in every base (a whole box on one rank)
    int32_t pat_stage_n = 0;       // > 0: every base is (far, -n, -1, 0, +1, +n, far) with one even n <= 1024 -- the near operands of a run of tiles can be staged in LDS (spmv_pattern_stage_kernel)
    bool pat_diag3 = false;        // stencil generator: every row has its diagonal, at table position 3 of its base   // table entries padded per pattern to a multiple of pat_unroll
    // CSR-DIA: operators with at most KR_DIA_MAX = 32 distinct (col - row) offsets that fill their diagonals (any stencil on a structured
    // grid, variable coefficients included): the values as ONE stream per diagonal in natural row order, absent entries marked by
    // a NaN payload no stored value may carry -- no row pointers, no per-entry codes (8 D + 16 bytes per row with x and y)
    double* d_dia = nullptr;        // dia_nd streams of dia_stride doubles: d_dia[d * dia_stride + row]
    int64_t dia_stride = 0;
    int32_t dia_nd = 0;
    int32_t dia_off[kr::KR_DIA_MAX] = {0};      // col - row of diagonal d (a halo slot's local column for the halo diagonals), in the rows' stored order
    int32_t dia_min = 0, dia_max = 0;   // over the diagonals that address x itself
    int64_t ntiles = 0;
    // slab order of the tiles for plane-structured operators (csr_create.hip: build_tile_order), or nullptr: slot -> tile (-1: none);
    // the first order_slots1 entries for runs of 1 slot per XCD, then order_slots8 entries for runs of 8
    int32_t* d_tile_order = nullptr; int64_t order_slots1 = 0, order_slots8 = 0, order_plane = 0;   // order_plane: b, rows per plane
    int slots = 7;            // SpMV pair slots per lane (picked from the average nnz of a 128-row slice)
    // distributed
    bool dist = false;
    std::vector<int64_t> row_offsets;
    kr::HaloPlan plan;
    int32_t* d_tiles_interior = nullptr; int64_t n_interior = 0;
    int64_t interior_first = -1;    // >= 0: the interior tiles are the contiguous range [interior_first, interior_first + n_interior)
    int32_t* d_tiles_boundary = nullptr; int64_t n_boundary = 0;
    bool send_contiguous = false;   // every send list is a contiguous run of local rows (k-slab stencils)
    bool halo_early_ok = false;     // ... on EVERY rank (agreed at creation): the solvers may start the exchange of a new direction vector
                                    // behind the pass that writes it -- a per-iteration exchange that all ranks issue or none does
    bool halo_pushed_inline = false;            // (peer stores) the push in flight was enqueued on the compute stream itself
    const double* halo_started_for = nullptr;   // the halo exchange of this input vector is already in flight (halo_begin: a solver started it early)
    // where the CSR arrays live (bench_streams.hip: csr_place): homes tried, the one kept, the traffic skeleton's ms on each
    int32_t placement_tries = 1, placement_chosen = 0; double placement_ms[8] = {0};
};

namespace kr {

constexpr int KR_PMAX = 512;        // CSR-P16 limits: patterns and (padded) table entries held in LDS: 2 + 12 + 24 KiB at most
constexpr int KR_TMAX = 2048;

// y <- A x on ctx->s_main.  nq = 0: plain.  nq = 1: also tile partials of sum d[i]*y[i] into partial array 0.
// nq = 2: additionally sum y[i]*y[i] into partial array 1.  `done` (device flag) makes the launch a no-op when set.
int32_t launch_spmv(kryst_csr_t a, const double* x, double* y, int nq, const double* dvec, const int* done);
// CG / PCG with the direction pass inside the SpMV (spmv.hip: spmv_pattern_fuse_kernel): whether the operator can take it, and the launch
bool spmv_can_fuse_direction(kryst_csr_t a);
int32_t launch_spmv_fused(kryst_csr_t a, const double* z, const double* p_old, double* p_new, double* xvec, double* y, int nq,
                          const double* alpha, const double* beta, const long long* xpend, long long it, const int* done);
// Distributed operators: start the halo exchange of x NOW (everything enqueued on the compute stream so far is waited for, nothing
// later) and remember it, so that the next launch_spmv(a, x, ...) does not start it again.  A solver calls it after the launch that
// wrote the rows the neighbours need and before the launches that write the rest (send_contiguous operators only).
int32_t halo_begin(kryst_csr_t a, const double* x);
// switch the operator's halo exchange to direct peer stores (collective; KRYST_UNSUPPORTED on every rank when a rank cannot map a peer's
// landing buffer or a neighbour relation is one-way) / back to RCCL
int32_t halo_peer_setup(kryst_csr_t a);
int32_t halo_peer_selftest(kryst_csr_t a);   // one checked exchange over the fresh mappings; agreed verdict (KRYST_OK / KRYST_UNSUPPORTED on every rank)
void halo_peer_destroy(kryst_csr_t a);
int32_t csr_place(kryst_csr_t a);            // KRYST_CSR_PLACEMENT_TRIES: the fastest of up to K homes for (row_ptr, col, val), by the traffic skeleton (bench_streams.hip)
int32_t halo_default_mode(kryst_csr_t a);    // at creation: peer stores when they work on every rank, else RCCL (KRYST_HALO_MODE=rccl: RCCL)
// the tile ranges [lo, hi) whose rows are sent to neighbours, merged and ascending (send_contiguous operators)
void halo_send_tiles(kryst_csr_t a, std::vector<std::pair<int64_t, int64_t>>& ranges);

}  // namespace kr
