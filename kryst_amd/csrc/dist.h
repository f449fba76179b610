// RCCL-over-xGMI layer that replaces the reference's MPI placeholder (src/parallel/mpi_comm.rs).
// One process per GPU; librccl.so.1 is bound at run time with dlopen so that a single-GPU user never needs it
// and so that a host process that already carries an RCCL (e.g. torch's) shares that copy.
#pragma once
#include "common.h"

namespace kr {

int32_t comm_unique_id(void* out128);
int32_t comm_init(kryst_ctx_t ctx, const void* uid128);
void    comm_destroy(kryst_ctx_t ctx);
// all-gather of `count` doubles per rank on ctx->s_main: recv[p*count + i]
int32_t comm_all_gather(kryst_ctx_t ctx, const double* send, double* recv, int count);
int32_t comm_all_gather_i64(kryst_ctx_t ctx, const int64_t* send, int64_t* recv, int count, hipStream_t s);
// grouped neighbour exchange: for every peer p, send send_counts[p] items at send + send_off[p] and receive
// recv_counts[p] items at recv + recv_off[p]; elem_bytes is 8 (double or int64)
int32_t comm_exchange(kryst_ctx_t ctx, const void* send, const int64_t* send_counts, const int64_t* send_off,
                      void* recv, const int64_t* recv_counts, const int64_t* recv_off, bool is_double, hipStream_t s);

// Mailbox path of the scalar all-reduce (see kryst_ctx_s): collective over the context's ranks (one all-gather of the IPC
// handles).  On success ctx->ipc_on is set on EVERY rank, or on none (the outcome is agreed through an all-gather).
int32_t ipc_reduce_setup(kryst_ctx_t ctx);
int32_t ipc_reduce_selftest(kryst_ctx_t ctx);          // solvers.hip: one checked reduction over the fresh mailboxes (KRYST_OK / KRYST_UNSUPPORTED on every rank)
// one allocation of every rank mapped into this process (hipIpc between processes, directly between ranks of one process); collective,
// agreed outcome (KRYST_OK everywhere or KRYST_UNSUPPORTED everywhere); `opened` collects the mappings to close with hipIpcCloseMemHandle
void ipc_close_shared(void* ptr);                       // closes a mapping ipc_map_peers opened (shared between the ranks of a process, counted)
// *sibling (optional): some peer is a rank of this very process and is addressed by its raw device pointer (no mapping keeps its memory alive)
int32_t ipc_map_peers(kryst_ctx_t ctx, void* mine, std::vector<void*>& peers, std::vector<void*>& opened, bool same_device_siblings, bool* sibling = nullptr);
void    ipc_reduce_destroy(kryst_ctx_t ctx);

// Halo exchange by direct peer stores ("direct peer writes over xGMI / IPC-mapped buffers"; replaces the neighbour exchange
// the reference's MpiComm leaves as a TODO, src/parallel/mpi_comm.rs:133-143): every rank owns a LANDING buffer in fine-grained device
// memory that its neighbours map (hipIpc; directly between ranks of one process) -- two parities x total_recv doubles, then one 16-byte
// stamp cell per sender.  An exchange = one push kernel on the sender (the rows each neighbour needs, stored straight into that
// neighbour's landing buffer [epoch parity] with system-scope write-through stores; the workgroup that finishes last writes the epoch
// into the neighbours' stamp cells) and one pull kernel on the receiver's compute stream in front of the boundary tiles (polls its own
// stamp cells for the epoch, copies landing[parity] into the plan's d_halo).  No ncclSend / ncclRecv, no pack kernel, no event between
// the receive and the compute stream.  Two parities suffice when every neighbour relation is mutual (checked at set-up): a rank pushes
// exchange e + 2 only after it has consumed its neighbours' e + 1, which they push after they have consumed e.
struct HaloPushSeg { double* dst; unsigned long long* stamp; int64_t dst_stride; int64_t src; int64_t count; };   // dst: peer landing + my offset there;
                                                                     // src: first local row (contiguous lists) or first entry of d_send_idx
struct HaloPullSeg { const unsigned long long* stamp; int64_t off; int64_t count; };    // my stamp cell of that sender, its range in d_halo
struct HaloPeer {
    bool on = false;
    double* landing = nullptr;           // fine-grained: 2 * stride doubles of data, then 2 doubles (one 16-byte cell) per rank
    int64_t stride = 0;                  // doubles per parity (total_recv rounded up to a multiple of 2, at least 2)
    std::vector<void*> opened;           // peer mappings to close
    HaloPushSeg* d_push = nullptr; int npush = 0; int64_t push_max = 0;      // longest segment (grid size)
    HaloPullSeg* d_pull = nullptr; int npull = 0; int64_t pull_max = 0;
    unsigned int* d_ticket = nullptr;    // the push kernel's "last workgroup" counter
    unsigned long long epoch = 0;        // exchanges issued on this operator so far (the same number on every rank)
    bool sibling = false;                // a rank thread of this process stores into `landing` by raw pointer: freed with the context, not the operator
    bool failed = false;                 // the test exchange did not arrive intact (agreed across ranks): never switched on again
};

// Halo plan of a row-partitioned operator (host side; also exported as kryst_host_halo_recv_plan)
struct HaloPlan {
    int nranks = 1, rank = 0;
    int64_t row_lo = 0, row_hi = 0;
    std::vector<int64_t> recv_counts, recv_off, send_counts, send_off;
    std::vector<int64_t> recv_cols;      // global columns, grouped by owner, ascending inside a group
    int64_t total_recv = 0, total_send = 0;
    int32_t* d_send_idx = nullptr;       // local rows to pack, total_send
    double*  d_sendbuf = nullptr;        // total_send
    double*  d_halo = nullptr;           // total_recv
    HaloPeer peer;                       // the exchange by direct peer stores (off unless kryst_csr_halo_mode switched it on)
};

// recv side from the local rows (global column ids); returns the plan's recv_* fields
void halo_recv_plan(int rank, int nranks, const int64_t* row_offsets, int64_t nloc, const int64_t* row_ptr,
                    const int64_t* col_global, HaloPlan* plan);
// slot of global column c in the concatenated recv list (binary search inside the owner's group)
int64_t halo_slot(const HaloPlan& plan, const int64_t* row_offsets, int64_t c);

}  // namespace kr
