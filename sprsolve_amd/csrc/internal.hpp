// Internal declarations shared by the translation units of libsprsolve_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <ctime>
#include <mutex>
#include <vector>

#include "../../include/sprsolve_hip.h"
#include "scalar.hpp"

namespace sprs {

constexpr int BLOCK = 256;     // threads per workgroup (4 wavefronts of 64)
constexpr uint32_t UNI2 = 0x40000000u;  // 128-row block descriptors (spmv_dict.hip), rb bit 30: every row has the SAME code sequence
constexpr uint32_t SEAM2 = 0x20000000u; // ... bit 29 (with UNI2, full 128-row blocks): all rows but one or two; rb's low bits then describe those (spmv_dict.hip)
constexpr int MAX_GRID = 4096; // upper bound of the streaming-kernel grid (= max partials per reduction)

// device-side status word of a running solve
enum : int { ST_RUNNING = 0, ST_CONVERGED = 1, ST_RESTART = 2, ST_BREAKDOWN = 3, ST_INVALID_PC = 4,
             ST_COMM_TIMEOUT = 5 };   // a peer's hand-off never arrived in this rank's mailbox (device.hpp, mbox_sum)

// Hand-off of a fused reduction whose FINAL value is computed inside the producing launch (device.hpp,
// finalize_last_block): the workgroup that arrives last re-reduces all partials in the library's fixed order and writes
// the scalars to out0 / out1.  Used by the distributed solves, whose consumer is an ncclAllReduce of those scalars: one
// stream operation per hand-off instead of a finalize launch plus the collective.  counter == nullptr: no finalize (the
// consumer kernel re-reduces the partials itself, the single-GPU hand-off).
// Peer-to-peer mailboxes of a communicator (dist.hip, p2p_setup; SURVEY §8e's deterministic all-reduce): every rank owns one
// mailbox of MB_BYTES in uncached device memory, mapped into every other rank of the node (hipIpcOpenMemHandle).  A hand-off's
// producer writes its reduced scalars into EVERY rank's mailbox — entry [slot][parity][source rank], eight 8-byte granules
// {32 data bits, 32-bit tag}, each one naturally aligned store, so a granule is seen whole or not at all whatever the link —
// and the consumer kernels of all ranks sum the `world` entries in rank order: bit-identical on every rank, no stream operation.
constexpr int MB_SLOTS = 4, MB_RANKS = 8, MB_GRAN = 8;
constexpr size_t MB_ENTRY = MB_GRAN * 8;                                  // 64 B: two 16-byte cells (out0, out1) as 8 granules
constexpr size_t MB_BYTES = (size_t)MB_SLOTS * 2 * MB_RANKS * MB_ENTRY;   // 4 KiB
inline size_t mb_offset(int slot, int parity) { return ((size_t)slot * 2 + (size_t)parity) * MB_RANKS * MB_ENTRY; }
struct P2pBox {
    unsigned long long peer[MB_RANKS];   // rank q's mailbox as mapped on this device (own rank: the local allocation)
    int world, rank;
};

struct Fin {
    unsigned int *counter = nullptr;   // agent-scope arrival counter, zero between launches
    const void *base0 = nullptr, *base1 = nullptr;   // first partial of the WHOLE reduction (an earlier launch may have written the head)
    void *out0 = nullptr, *out1 = nullptr;           // the reduced values (16-byte slots of the solver's `red` buffer)
    int P = 0;                         // partials of the whole reduction
    // peer-to-peer hand-off: the last workgroup also posts the values into every rank's mailbox (tag != 0)
    const P2pBox *box = nullptr;
    unsigned int tag = 0;              // this hand-off's sequence number on its slot (never 0)
    unsigned int mb_off = 0;           // byte offset of [slot][parity] inside a mailbox
};

}  // namespace sprs

struct sprs_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    int num_cu = 256;
    int grid = 512;      // blocks launched by streaming / reduction kernels (multiple of 8; 2 per CU, set in sprs_ctx_create)
    int spmv_grid = -1;    // workgroups of the persistent SpMV grid; -1 = auto (4 per CU)
    // SpMV placement: -1 = auto (measured on MI355X, profiles/r01_tuning.md): matrices whose stream fits the 256 MiB
    // Infinity Cache run best with one contiguous chunk of row blocks per XCD (x stays in that XCD's L2); HBM-bound
    // ones run best round-robin (all XCDs sweep the same region, x re-reads are served by the Infinity Cache).
    int xcd_chunk = -1;
    // dictionary-compressed SpMV stream: -1 auto (the most compact the matrix qualifies for), 0 plain CSR,
    // 1 offset codes + values, 2 (offset, value) pair codes.  Read at handle creation (what is built) and at launch (what is used).
    int spmv_dict = -1;
    int spmv_wide = -1;   // f64 pair codes: two rows per lane (16-byte gathers); -1 / 1 on, 0 off
    int spmv_wideload = -1; // plain-CSR stream, f64: 16 bytes per lane per stream load, 3 workgroups per CU on HBM-sized matrices; 0 = the 4/8-byte kernel.  Read at creation and at launch
    int spmv_eqrows = -1; // plain-CSR stream: blocks of equal-length rows take their extents from the descriptor (no row_ptr read); read at creation
    int spmv_period = -1;  // XCD-period walk for matrices with a far band (3-D stencils): -1 automatic = the f64 pair-code stream only, 1 = the offset-code stream too, 0 = off.  Read at creation
    int spmv_triple = -1;  // f64 pair codes, uniform blocks: columns c - 1 and c + 1 read from column c's loads; 0 = off.  Read at creation
    int spmv_seam = -1;    // f64 pair codes: blocks that are uniform but for one or two adjacent rows lacking one slot run the uniform path; 0 = off.  Read at creation
    int spmv_tile = -1;    // f64 compressed streams: LDS x-window tiles for the near columns of uniform stencil runs (spmv_tile_kernel, spmv_tile_off_kernel): -1 automatic = vectors of 44 MiB and more (tile_wanted), 1 = every matrix that has such runs, 0 = off.  Read at creation; 0 also at launch
    int spmv_chain = -1;   // f64 pair codes, 3-D stencils: plane-streaming chains (spmv_chain_kernel): -1 automatic = wherever the tile plan is wanted and the chains fill the chip, 1 = wherever chains exist, 0 = off.  Read at creation; 0 also at launch
    int spmv_fuse = -1;    // BiCGStab (f64, no preconditioner, one GPU) on a handle whose SpMV runs through chains: K3 formed inside K4 and K1 inside K2 (krylov.hip, "fused SpMV input"); 0 = off.  Read per solve
    int ew_chunk = -1;     // fused recurrence kernels walk one contiguous eighth of the vectors per XCD: -1 automatic (fused_chunked), 0 / 1
    int stream_nt = -1;    // fused recurrence kernels access their vectors with non-temporal loads / stores: -1 auto (by vector size), 0 / 1
    int spmv_uniform = -1; // ... and blocks whose rows all repeat one code sequence read neither codes nor row_ptr; read at creation
    int halo_overlap = 1;  // distributed SpMV: run the halo-free rows while the halo travels
    int gs_graph = 0;    // Gauss-Seidel: 1 = replay a sweep's level launches from a hipGraph (tests/test_gpu_gauss_seidel.py; no gain measured, r01_tuning.md)
    int p2p_allreduce = -1; // distributed hand-offs through the peer-to-peer mailboxes instead of ncclAllReduce: -1 / 1 wherever the communicator has them, 0 = RCCL.  Read at communicator creation (0: no mailboxes are set up) and per solve
    int p2p_timeout_ms = 20000; // a consumer that has polled its mailbox this long gives up (status ST_COMM_TIMEOUT -> SPRS_ERR_RCCL)
    int poll = 16;       // iterations between host polls of the device status word
    double *d_part = nullptr;  // reduction partials for the stand-alone vecalg entry points
    double *d_scal = nullptr;  // small device result buffer
    double *h_scal = nullptr;  // pinned host mirror
    mutable char err[512] = {0};
    // When both are set, the next SpMV kernel launch records them as its OWN begin / end (hipExtLaunchKernelGGL): the
    // solver's profile then times the kernel exactly as rocprofv3 does, without the dispatch gap two hipEventRecord
    // calls around the launch would include (1104 vs 1089 us on the cfg-5 plain stream).  Set and cleared by
    // KrylovBase::spmv under the context mutex.
    hipEvent_t prof_start = nullptr, prof_stop = nullptr;
    // Serialises every entry point that uses per-context or per-handle scratch (reduction partials, the pinned
    // scalar mirror, the host-slice staging buffers) or that must see its own results on the single stream: the
    // reference shares `&M` between threads (bicg_stab.rs:17-18 `T: Send + Sync`, mat.rs:156-161), so concurrent
    // sprs_mul_vec_* / solves on handles of ONE context must be safe.  They are: they queue behind this mutex (the
    // GPU runs one stream anyway); callers that want concurrency create one context per thread.  Recursive because
    // the blocking entry points nest (a solve calls norm2_host, mul_vec_dot calls reduce_partials_host).
    mutable std::recursive_mutex mu;
};
struct CtxLock {
    std::unique_lock<std::recursive_mutex> lk;
    explicit CtxLock(const sprs_ctx *c) { if (c) lk = std::unique_lock<std::recursive_mutex>(c->mu); }
};

#define SPRS_HIP_TRY(ctx, expr)                                                                     \
    do {                                                                                            \
        hipError_t e__ = (expr);                                                                    \
        if (e__ != hipSuccess) {                                                                    \
            snprintf((ctx)->err, sizeof((ctx)->err), "%s:%d: %s -> %s", __FILE__, __LINE__, #expr,  \
                     hipGetErrorString(e__));                                                       \
            return SPRS_ERR_HIP;                                                                    \
        }                                                                                           \
    } while (0)

// Launch of an SpMV kernel on the context's stream; timed by the launch itself when a profile is being taken.
#define SPRS_LAUNCH_SPMV(c, kernel, grid, ...)                                                                            \
    do {                                                                                                                 \
        if ((c)->prof_start && (c)->prof_stop)                                                                           \
            hipExtLaunchKernelGGL(kernel, dim3(grid), dim3(sprs::BLOCK), 0, (c)->stream, (c)->prof_start, (c)->prof_stop, 0, __VA_ARGS__); \
        else                                                                                                             \
            hipLaunchKernelGGL(kernel, dim3(grid), dim3(sprs::BLOCK), 0, (c)->stream, __VA_ARGS__);                      \
    } while (0)

// SPRS_CREATE_TRACE=1: stage times of handle creation on stderr (diagnostics; profiles/r03_tuning.md §4)
struct CreateTrace {
    bool on; double t0;
    static double now() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6; }
    CreateTrace() : on(getenv("SPRS_CREATE_TRACE") != nullptr), t0(now()) {}
    void lap(const char *what) { if (!on) return; const double t = now(); fprintf(stderr, "[sprs create] %-28s %8.2f ms\n", what, t - t0); t0 = t; }
};

#define SPRS_TRY(expr)                       \
    do {                                     \
        int s__ = (expr);                    \
        if (s__ != SPRS_OK) return s__;      \
    } while (0)

// RCCL communicator of one rank (dist.hip).  RCCL is resolved with dlopen at first use, so the
// single-GPU library has no link-time dependency on it.
struct sprs_comm {
    sprs_ctx *ctx = nullptr;
    void *nccl = nullptr;   // ncclComm_t
    int world = 1, rank = 0;
    // peer-to-peer mailboxes (sprs::P2pBox): set up collectively at creation when every rank could map every other's
    bool p2p = false;
    void *mbox = nullptr;                 // this rank's mailbox (uncached device memory, MB_BYTES)
    void *peer_map[8] = {nullptr};        // the other ranks' mailboxes as opened here (hipIpcOpenMemHandle); [rank] = mbox
    sprs::P2pBox *d_box = nullptr;        // device copy of the table
    unsigned int seq[4] = {0, 0, 0, 0};   // hand-offs issued per slot: the host recurrences are replicated, so every rank counts alike
};

// Row-partition metadata of a distributed CSR operator: which local x entries each peer needs
// (packed and sent before every SpMV) and where the entries received from each peer land in
// the halo tail [n_local, n_ext) of the extended x vector.
// Runs of TILE_B consecutive full uniform 128-row blocks that share one pattern whose NEAR columns (|col - row| <= TILE_W - 2)
// are taken from an LDS window of x (spmv_dict.hip); the blocks outside those runs stay with the per-block walk (same launch)
struct sprs_tile_plan {
    int32_t *list = nullptr;       // device: {first 128-row block, first row} of each tile, eight per-XCD sections in row order
    int32_t *xstart = nullptr;     // device, 9 entries: section bounds within list
    int32_t *left = nullptr;       // device: the other blocks, in the walk order they had (128-row blocks: pair stream; 64-row blocks: offset stream)
    int n_tile = 0, n_left = 0;
    int ul = 0, fl = 0, fh = 0;    // pattern shape: slots, leading far slots, trailing far slots
    int w = 0;                     // half-width of the x window (512, or 1536 for the pair stream's long-line grids)
    int32_t off[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    double val[8] = {0, 0, 0, 0, 0, 0, 0, 0};      // pair stream: the pattern's values
    std::vector<int32_t> h_list, h_xstart;         // host copies of list / xstart (the distributed operator cuts its interior plan from them)
};

// Plane-streaming CHAINS of the f64 pair-code stream (spmv_chain.hip, round 4): a pattern with exactly one far slot a side at
// -Pf / +Pf (a 3-D stencil's plane neighbours).  A chain = tiles of CH_B uniform 128-row blocks at rows ts, ts + Pf, ts + 2 Pf, ...
// (each within 127 rows of that, on the 128-row block grid); a workgroup walks a segment of a chain keeping the x windows of three
// consecutive tiles in LDS, so the +-Pf operands come from the neighbouring tiles' windows: no far load at all.
struct sprs_chain_plan {
    int32_t *tiles = nullptr;      // device, int4 per tile: {first 128-row block, first row, first row of the NEXT window (the chain's next tile, or ts + Pf), 0}; segments contiguous
    int32_t *segs = nullptr;       // device, int2 per segment: {first tile, tiles}
    int32_t *xstart = nullptr;     // device, 9 entries: per-XCD ranges of segs
    int32_t *left = nullptr;       // device: the 128-row blocks outside the chains, in the per-block kernel's walk order
    int n_tile = 0, n_seg = 0, n_chain = 0, n_left = 0;
    int ul = 0, tri = 0;           // pattern slots; tri: the near slots are (.., c - 1, c, c + 1, ..) around an even centre, the others even (16-byte LDS reads)
    int32_t off[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    double val[8] = {0, 0, 0, 0, 0, 0, 0, 0};
};

struct sprs_dist_info {
    sprs_comm *comm = nullptr;
    int64_t n_local = 0, n_ext = 0;
    std::vector<int> peer;
    std::vector<int64_t> send_off, recv_off;   // n_peers + 1 each (elements)
    int32_t *send_idx = nullptr;               // device: local indices to pack, grouped by peer
    void *send_buf = nullptr;                  // device: packed values
    // north_star's literal exchange: ncclAllGather of every rank's (padded) x slice into ag_buf, column
    // indices address ag_buf = [rank 0 slice | rank 1 slice | ...]; ag_slice == 0 selects the sparse halo
    int64_t ag_slice = 0;
    void *ag_buf = nullptr;
    // overlap of the halo exchange with the SpMV of the rows that need no halo entry
    int32_t *order_int = nullptr, *order_bnd = nullptr;   // device: interior / boundary row blocks
    int32_t n_int = 0, n_bnd = 0;
    // the same split in units of the 128-row blocks of the two-rows-per-lane kernel (null when it does not apply)
    int32_t *order_int_w = nullptr, *order_bnd_w = nullptr;
    sprs_tile_plan tile_int;       // LDS-window tiles that hold interior rows only + the interior blocks outside them (the interior launch)
    bool tile_int_off = false;     // ... of the offset-code stream (else the pair-code stream)
    int32_t n_int_w = 0, n_bnd_w = 0;
    hipStream_t comm_stream = nullptr;
    hipEvent_t ev_pack = nullptr, ev_halo = nullptr;
};

// Dictionary-compressed copy of the (col_idx, val) stream (spmv_dict.hip): one byte per nnz indexing the
// table of distinct (col - row) offsets or, for real matrices with few distinct values, the table of distinct
// (offset, value) pairs.  Built at handle creation when the matrix qualifies.
struct sprs_dict {
    uint8_t *idx_code = nullptr;   // device, nnz (+ pad): code of col - row
    uint8_t *pair_code = nullptr;  // device, nnz (+ pad): code of the (col - row, value) pair, or null
    int32_t *off_tab = nullptr;    // device, 256 entries: col - row per offset code
    int32_t *pair_off = nullptr;   // device, 256 entries: col - row per pair code
    void *pair_val = nullptr;      // device, 256 entries of T: value per pair code
    void *rowval = nullptr;        // device, nrows of T (complex scalars only): the value of each row's offset-0 entry — pair code 255 means "this row's value" (spmv_dict.hip, cpair stage)
    int n_off = 0, n_val = 0, n_pair = 0;
    void *wide_desc = nullptr;     // device: descriptors of the 128-row blocks of the two-rows-per-lane kernel (f64 pair codes)
    int n_wide = 0;
    int32_t *wide_order = nullptr; // device: XCD-period schedule of the 128-row blocks (null = natural order)
    int32_t *off_order = nullptr;  // device: the same schedule for the 64-row blocks of the offset-code stream
    int64_t period = 0;            // the far band it folds over (rows)
    int64_t max_off = -1;          // largest |col - row| of the matrix (-1: unknown)
    // tile plans of spmv_tile_kernel (f64, HBM-sized stencil-like matrices): one for the pair-code stream, one for the
    // offset-code stream (values per entry) when that is the stream the handle multiplies with
    sprs_tile_plan tile_pair, tile_off;
    sprs_chain_plan chain_pair;    // plane-streaming chains of the pair-code stream (preferred over tile_pair where it exists; knob "spmv_chain")
    void *owide_desc = nullptr;    // device: 128-row descriptors of the offset-code stream (uniform / seam blocks marked on its codes)
    int n_owide = 0;
    void *off_desc = nullptr;      // device: copy of blk_desc for the offset-code stream with the uniform blocks flagged (bit 30, nn = row length)
    int n_off_uniform = 0;
};

struct sprs_csr {
    sprs_ctx *ctx = nullptr;
    int dtype = 0;               // sprs::DT_D / DT_Z / DT_S / DT_C
    int64_t nrows = 0, ncols = 0, nnz = 0;
    int32_t *row_ptr = nullptr;  // device
    int32_t *col_idx = nullptr;  // device
    void *val = nullptr;         // device, T
    bool owns_arrays = true;
    bool was_csc = false;        // created from CSC arrays (GaussSeidel::new rejects those, gauss_seidel.rs:22-26)
    int32_t *rowblk = nullptr;   // device: n_rowblk+1 row starts, bit31 set on vector-mode blocks
    int32_t n_rowblk = 0;
    void *blk_desc = nullptr;      // device: one 16-byte {ra, rb|flag, pa, nn} descriptor per row block
    void *blk_desc_eq = nullptr;   // device: the plain-CSR kernel's copy, equal-length blocks flagged (rb bit 30, row length in nn >> 16)
    int32_t n_eq_blocks = 0;
    void *tail = nullptr;          // device, 64 B: zero-padded copy of the last 4-entry group of col_idx (16 B) and val (32 B) for spmv_wide_kernel; null = that kernel does not apply       // ... how many are flagged (their row_ptr entries are not read)
    // scratch for the host-slice trait entry points (lazily allocated)
    void *x_tmp = nullptr, *y_tmp = nullptr;
    double *part = nullptr;      // partials for mul_vec_dot
    sprs_dist_info *dist = nullptr;   // non-null: row block of a matrix partitioned over ranks
    sprs_dict *dict = nullptr;        // non-null: dictionary-compressed stream available (spmv_dict.hip)
};

struct sprs_diag {
    sprs_ctx *ctx = nullptr;
    int t_dtype = 0;    // T (sprs::DT_*)
    int v_complex = 0;  // V is complex (same precision as T)
    size_t n = 0;
    void *dinv = nullptr;  // device, V
    void *in_tmp = nullptr, *out_tmp = nullptr;
};

namespace sprs {

// ---- spmv.hip
// y = A x.  dot_mode 0: none; 1: part0[b] = sum conj(u_i) y_i; 2: part0 = sum conj(y_i) y_i, part1 = sum conj(y_i) u_i.
// status (device int*, may be null): kernels return immediately when *status != 0.
// conj_x: gather conj(x[col]) instead of x[col] (CSMINRES: A * conj(q), cs_minres.rs:99-101, without materialising conj(q)).
template <class T>
int launch_spmv(const sprs_csr *A, const T *x, T *y, int dot_mode, const T *u, T *part0, T *part1, const int *status,
                bool conj_x = false, const Fin *fin = nullptr);
int tile_blocks();   // 128-row blocks per LDS-window tile (spmv_dict.hip)
bool tile_plan_used(const sprs_csr *A);   // the SpMV of this handle runs through its tile plan
bool chain_plan_used(const sprs_csr *A);  // ... through its plane-streaming chains (spmv_chain.hip)
int build_rowblocks(sprs_csr *A, const int32_t *host_row_ptr);   // host_row_ptr == null: row_ptr lives in HBM only (summaries first)
int validate_cols_device(const sprs_csr *A);   // SPRS_INVALID_ARGUMENT if any col_idx is outside [0, ncols)
int spmv_num_partials(const sprs_csr *A);  // workgroups launch_spmv uses == partials it writes
// per-row-block column span (device kernel + D2H): lo/hi sized n_rowblk
int rowblk_spans(const sprs_csr *A, std::vector<int32_t> &lo, std::vector<int32_t> &hi);
// SpMV over a subset of the row blocks (order[0..count)); writes `subset_grid(count)` partials
template <class T>
int launch_spmv_subset(const sprs_csr *A, const int32_t *order, int count, const T *x, T *y, int dot_mode, const T *u,
                       T *part0, T *part1, const int *status, bool conj_x, const Fin *fin = nullptr);
int spmv_subset_grid(const sprs_csr *A, int count);
// ---- spmv_dict.hip
// SPRS_OK also when the matrix does not qualify (A->dict stays null); blk / blk_pa: the row blocks just built (first row / first entry)
int build_dict(sprs_csr *A, bool has_vector_blocks, const std::vector<int32_t> &blk, const std::vector<int32_t> &blk_pa);
void free_dict(sprs_csr *A);
int dict_mode(const sprs_csr *A);                      // 0 plain, 1 offsets, 2 offsets + values: what launch_spmv will use
// Should the fused recurrence kernels of a solve on A give every XCD one contiguous eighth of the vectors (spmv.hip)?
bool fused_chunked(const sprs_csr *A);

template <class T>
int launch_spmv_dict(const sprs_csr *A, int mode, const int32_t *order, int count, int g, int xcd_chunk, const T *x, T *y,
                     int dot_mode, const T *u, T *part0, T *part1, const int *status, bool conj_x, const Fin &fin);

// ---- blas1.hip  (all on ctx->stream, asynchronous)
template <class T, class S> int launch_axpy(sprs_ctx *c, size_t n, S a, const T *x, T *y);
template <class T> int launch_axpby(sprs_ctx *c, size_t n, T a, const T *x, T b, T *y);
template <class T> int launch_scale(sprs_ctx *c, size_t n, T a, T *x);
template <class T> int launch_rscale(sprs_ctx *c, size_t n, Real<T> a, T *x);
template <class T> int launch_conj(sprs_ctx *c, size_t n, const T *in, T *out);
template <class T, class V> int launch_diag_apply(sprs_ctx *c, size_t n, const V *dinv, const T *in, T *out);
template <class V> int launch_diag_inv(sprs_ctx *c, size_t n, const V *diag, V *dinv);
// reductions: blocking, result returned to the host
// comm != null: the locally reduced value is all-reduced over the ranks before it is returned
template <class T> int dot_host(sprs_ctx *c, size_t n, const T *x, const T *y, bool conj, T *out, sprs_comm *comm = nullptr);
template <class T> int norm2_host(sprs_ctx *c, size_t n, const T *x, Real<T> *out, sprs_comm *comm = nullptr);
// reduce `P` partials of T (or of double when T_is_real_partials) with the library's fixed order; blocking
template <class T> int reduce_partials_host(sprs_ctx *c, const T *part, int P, T *out, sprs_comm *comm = nullptr);

// ---- dist.hip
// exchange the halo entries of the extended vector x (local part [0,n_local) already in place)
template <class T> int halo_exchange(const sprs_csr *A, T *x_ext);
// split form used for overlap: pack + exchange on the communication stream (returns immediately),
// then make the compute stream wait for the halo
template <class T> int halo_begin(const sprs_csr *A, T *x_ext);
int halo_wait(const sprs_csr *A);
template <class T>
int dist_spmv(const sprs_csr *A, T *x_ext, T *y, int dot_mode, const T *u, T *part0, T *part1, const int *status, bool conj_x,
              const Fin *fin = nullptr);   // fin: handed to the LAST launch of the operator (P and base pointers span all of them)
int allreduce_sum(sprs_comm *comm, void *dev, size_t count, bool f32 = false);   // in place, on the ctx stream; count elements of f64 (or f32)

inline int grid_for(const sprs_ctx *c) {
    int g = c->grid;
    if (g < 8) g = 8;
    if (g > MAX_GRID) g = MAX_GRID;
    return g & ~7;
}

// Non-temporal vector accesses — operand loads and result stores of the fused recurrence kernels, the y stores of the
// pair-code SpMV: knob "stream_nt" (1 / 0), automatic (-1) from the size of one vector.  Measured cross-over: 64 MB
// (-0.9 %; at 50 MB the vectors of a solve live in the 256 MiB Infinity Cache: -3 %) to 80 MB (+3.6 %; 100-200 MB: +11 %),
// profiles/r02_tuning.md §20.
inline bool stream_loads_nt(const sprs_ctx *c, size_t vector_bytes) {
    if (c->stream_nt >= 0) return c->stream_nt != 0;
    return vector_bytes >= (size_t)72 << 20;
}
// the whole stream + three vectors fit the 256 MiB Infinity Cache (spmv.hip: XCD-chunked walks for these, round-robin otherwise)
inline bool is_cache_resident(const sprs_csr *A) {
    const double s = (double)dtype_size(A->dtype);
    return (double)A->nnz * (s + 4) + 3.0 * A->nrows * s < 192.0 * 1024 * 1024;
}

// LDS-window tiles of the compressed-stream SpMVs (spmv_dict.hip): knob "spmv_tile" 1 / 0, automatic (-1) from the size of one
// vector.  Measured cross-over on 500 x 500 x nz slabs (per-block kernel -> tiles, SpMV us in the solve; profiles/r03_tuning.md
// §9): 16 MB 15.6 -> 31.9, 32 MB 22.1 -> 30.3, 50 MB 36.5 -> 33.6, 100 MB 71.8 -> 64.6, 200 MB 139 -> 110, 400 MB 278 -> 213.
inline bool tile_wanted(const sprs_ctx *c, size_t vector_bytes) {
    if (c->spmv_tile >= 0) return c->spmv_tile != 0;
    return vector_bytes >= (size_t)44 << 20;
}

// Grid of a grid-stride streaming pass over `work` tiles: no more workgroups than the context's grid, and as few as make
// every workgroup take the same number of trips (1953 tiles on 512 workgroups are 4 trips for most and 3 for some —
// 496 workgroups take 4 each and the pass ends 5 % sooner; at HBM sizes the difference vanishes).  Multiple of 8 (XCDs).
inline int balanced_grid(const sprs_ctx *c, int64_t work) {
    const int g0 = grid_for(c);
    if (work <= g0) return (int)(work < 1 ? 1 : work);
    const int64_t trips = (work + g0 - 1) / g0;
    int64_t g = (work + trips - 1) / trips;
    g = (g + 7) & ~(int64_t)7;
    return (int)(g > g0 ? g0 : g);
}

}  // namespace sprs
