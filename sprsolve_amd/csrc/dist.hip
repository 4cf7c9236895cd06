// Multi-GPU layer: one process per GPU, the CSR matrix row-partitioned over the
// ranks, collectives issued from the C++ recurrence on the solver's stream through RCCL.
//
//  * before every SpMV: halo exchange of exactly the remote x entries the local rows reference
//    (a pack kernel + grouped ncclSend/ncclRecv with each neighbour).  For a z-slab partition of
//    a 7-point stencil this is one plane (2 MB) per neighbour instead of the 400 MB a full
//    all-gather of x would move per SpMV;
//  * per dot product / norm: the per-workgroup partials are reduced locally in the library's
//    fixed order, then one ncclAllReduce(sum) of the 1..4 scalars.  RCCL returns bit-identical
//    results on all ranks, so all ranks take the same convergence / restart / breakdown branch.
//
// xGMI is point-to-point (7 links per GPU): halo traffic goes only to the ranks that own
// referenced columns, so each transfer uses the direct link to that peer.
// RCCL is bound with dlopen so that libsprsolve_hip.so has no link-time dependency on it (and
// shares whichever librccl.so.1 the process — e.g. PyTorch — has already loaded).
#include <dlfcn.h>
#include <cstdlib>
#include <mutex>
#include <rccl/rccl.h>
#include <rocprim/device/device_scan.hpp>

#include "device.hpp"

namespace {

struct Rccl {
    void *so = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclCommCount) CommCount = nullptr;   // optional (evidence only)
    bool ok = false;
};

void rccl_load(Rccl &r);
Rccl &rccl() {
    static Rccl r;
    static std::once_flag once;     // two contexts created from two host threads must not race the dlopen
    std::call_once(once, [] { rccl_load(r); });
    return r;
}
void rccl_load(Rccl &r) {
    // SPRS_RCCL_LIB selects another build of the library (the tests point it at a shared-memory
    // stand-in so that several ranks can share the single GPU of a test box)
    if (const char *alt = getenv("SPRS_RCCL_LIB")) r.so = dlopen(alt, RTLD_NOW | RTLD_LOCAL);
    for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
        if (r.so) break;
        r.so = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
    }
    if (!r.so) return;
#define SPRS_SYM(field, sym) r.field = reinterpret_cast<decltype(r.field)>(dlsym(r.so, sym))
    SPRS_SYM(GetUniqueId, "ncclGetUniqueId");
    SPRS_SYM(CommInitRank, "ncclCommInitRank");
    SPRS_SYM(CommDestroy, "ncclCommDestroy");
    SPRS_SYM(AllReduce, "ncclAllReduce");
    SPRS_SYM(AllGather, "ncclAllGather");
    SPRS_SYM(Send, "ncclSend");
    SPRS_SYM(Recv, "ncclRecv");
    SPRS_SYM(GroupStart, "ncclGroupStart");
    SPRS_SYM(GroupEnd, "ncclGroupEnd");
    SPRS_SYM(GetErrorString, "ncclGetErrorString");
    SPRS_SYM(CommCount, "ncclCommCount");
#undef SPRS_SYM
    r.ok = r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.AllReduce && r.AllGather && r.Send && r.Recv && r.GroupStart &&
           r.GroupEnd && r.GetErrorString;
}

#define SPRS_NCCL_TRY(ctx, expr)                                                                        \
    do {                                                                                                \
        ncclResult_t e__ = (expr);                                                                      \
        if (e__ != ncclSuccess) {                                                                       \
            snprintf((ctx)->err, sizeof((ctx)->err), "%s:%d: %s -> %s", __FILE__, __LINE__, #expr,      \
                     rccl().GetErrorString(e__));                                                       \
            return SPRS_ERR_RCCL;                                                                       \
        }                                                                                               \
    } while (0)

// One round trip through the mailboxes at communicator creation (p2p_setup): every rank posts {rank + 1, 2 rank + 1} with the
// self-test's tag into every rank's mailbox and sums what arrives in its own.  result[0] = 1 when all `world` entries arrived in
// time and the sums are the expected ones — otherwise the communicator keeps ncclAllReduce for its hand-offs.
__global__ __launch_bounds__(sprs::BLOCK) void mbox_selftest_kernel(const sprs::P2pBox *box, unsigned int tag, unsigned int mb_off,
                                                                    const unsigned long long *own_entries, unsigned long long timeout,
                                                                    int *result) {
    using namespace sprs;
    Fin f;
    f.box = box; f.tag = tag; f.mb_off = mb_off;
    const int world = box->world, rank = box->rank;
    mbox_post<double, double>(f, (double)(rank + 1), (double)(2 * rank + 1));
    double a = 0.0, b = 0.0;
    const bool ok = mbox_sum2<double, double>(MboxSrc{own_entries, world, tag, timeout}, a, b);
    if (threadIdx.x == 0) result[0] = (ok && a == 0.5 * world * (world + 1) && b == (double)world * world) ? 1 : 0;
}

template <class T>
__global__ __launch_bounds__(sprs::BLOCK) void pack_kernel(int64_t n, const int32_t *__restrict__ idx,
                                                           const T *__restrict__ x, T *__restrict__ buf) {
    for (int64_t i = (int64_t)blockIdx.x * sprs::BLOCK + threadIdx.x; i < n; i += (int64_t)gridDim.x * sprs::BLOCK)
        buf[i] = x[idx[i]];
}

}  // namespace

namespace sprs {

int allreduce_sum(sprs_comm *comm, void *dev, size_t count, bool f32) {
    if (!comm) return SPRS_OK;
    sprs_ctx *c = comm->ctx;
    SPRS_NCCL_TRY(c, rccl().AllReduce(dev, dev, count, f32 ? ncclFloat : ncclDouble, ncclSum, (ncclComm_t)comm->nccl, c->stream));
    return SPRS_OK;
}

template <class T>
static int halo_issue(const sprs_csr *A, T *x, hipStream_t pack_stream, hipStream_t xfer_stream) {
    const sprs_dist_info *D = A->dist;
    sprs_ctx *c = A->ctx;
    const int64_t n_send = D->send_off.back();
    T *buf = reinterpret_cast<T *>(D->send_buf);
    if (n_send > 0) {
        int g = (int)std::min<int64_t>((n_send + BLOCK - 1) / BLOCK, 1024);
        hipLaunchKernelGGL((pack_kernel<T>), dim3(g), dim3(BLOCK), 0, pack_stream, n_send, D->send_idx, x, buf);
        SPRS_HIP_TRY(c, hipGetLastError());
    }
    if (xfer_stream != pack_stream) {
        SPRS_HIP_TRY(c, hipEventRecord(D->ev_pack, pack_stream));
        SPRS_HIP_TRY(c, hipStreamWaitEvent(xfer_stream, D->ev_pack, 0));
    }
    constexpr size_t W = sizeof(T) / sizeof(Real<T>);   // complex travels as 2 reals
    const ncclDataType_t NT_ = sizeof(Real<T>) == 4 ? ncclFloat : ncclDouble;
    ncclComm_t comm = (ncclComm_t)D->comm->nccl;
    SPRS_NCCL_TRY(c, rccl().GroupStart());
    for (size_t p = 0; p < D->peer.size(); ++p) {
        const int64_t ns = D->send_off[p + 1] - D->send_off[p], nr = D->recv_off[p + 1] - D->recv_off[p];
        if (ns > 0) SPRS_NCCL_TRY(c, rccl().Send(buf + D->send_off[p], (size_t)ns * W, NT_, D->peer[p], comm, xfer_stream));
        if (nr > 0) SPRS_NCCL_TRY(c, rccl().Recv(x + D->n_local + D->recv_off[p], (size_t)nr * W, NT_, D->peer[p], comm, xfer_stream));
    }
    SPRS_NCCL_TRY(c, rccl().GroupEnd());
    if (xfer_stream != pack_stream) SPRS_HIP_TRY(c, hipEventRecord(D->ev_halo, xfer_stream));
    return SPRS_OK;
}

template <class T>
int halo_exchange(const sprs_csr *A, T *x) {
    const sprs_dist_info *D = A->dist;
    if (!D || D->peer.empty()) return SPRS_OK;
    return halo_issue<T>(A, x, A->ctx->stream, A->ctx->stream);
}

// overlapped form: the pack runs on the compute stream (it reads the just-produced x), the
// transfer on the communication stream; halo_wait() makes the compute stream wait for it
template <class T>
int halo_begin(const sprs_csr *A, T *x) {
    const sprs_dist_info *D = A->dist;
    if (!D || D->peer.empty()) return SPRS_OK;
    return halo_issue<T>(A, x, A->ctx->stream, D->comm_stream);
}
int halo_wait(const sprs_csr *A) {
    const sprs_dist_info *D = A->dist;
    if (!D || D->peer.empty()) return SPRS_OK;
    SPRS_HIP_TRY(A->ctx, hipStreamWaitEvent(A->ctx->stream, D->ev_halo, 0));
    return SPRS_OK;
}
template int halo_begin<double>(const sprs_csr *, double *);
template int halo_begin<cplx>(const sprs_csr *, cplx *);
template int halo_begin<float>(const sprs_csr *, float *);
template int halo_begin<cplxf>(const sprs_csr *, cplxf *);
template int halo_exchange<double>(const sprs_csr *, double *);
template int halo_exchange<cplx>(const sprs_csr *, cplx *);
template int halo_exchange<float>(const sprs_csr *, float *);
template int halo_exchange<cplxf>(const sprs_csr *, cplxf *);

// y = A_local x_ext with the halo exchange; overlapped with the interior rows when the operator was
// split at creation.  Dot partials of the two launches are concatenated (spmv_num_partials()).
template <class T>
int dist_spmv(const sprs_csr *A, T *x_ext, T *y, int dot_mode, const T *u, T *part0, T *part1, const int *status,
              bool conj_x, const Fin *fin) {
    const sprs_dist_info *D = A->dist;
    if (D->ag_slice > 0) {
        // literal north_star exchange: all-gather the x slices of all ranks, then multiply
        sprs_ctx *c = A->ctx;
        constexpr size_t W = sizeof(T) / sizeof(Real<T>);
        SPRS_NCCL_TRY(c, rccl().AllGather(x_ext, D->ag_buf, (size_t)D->ag_slice * W, sizeof(Real<T>) == 4 ? ncclFloat : ncclDouble,
                                          (ncclComm_t)D->comm->nccl, c->stream));
        return launch_spmv<T>(A, reinterpret_cast<const T *>(D->ag_buf), y, dot_mode, u, part0, part1, status, conj_x, fin);
    }
    if (!D->order_int) {
        SPRS_TRY(halo_exchange<T>(A, x_ext));
        return launch_spmv<T>(A, x_ext, y, dot_mode, u, part0, part1, status, conj_x, fin);
    }
    SPRS_TRY(halo_begin<T>(A, x_ext));
    SPRS_TRY(launch_spmv_subset<T>(A, D->order_int, D->n_int, x_ext, y, dot_mode, u, part0, part1, status, conj_x));
    SPRS_TRY(halo_wait(A));
    // the boundary launch starts after the interior one has finished (same stream): its last workgroup finalizes the
    // concatenated partials of both
    const int off = spmv_subset_grid(A, D->n_int);
    return launch_spmv_subset<T>(A, D->order_bnd, D->n_bnd, x_ext, y, dot_mode, u, part0 ? part0 + off : nullptr,
                                 part1 ? part1 + off : nullptr, status, conj_x, fin);
}
template int dist_spmv<double>(const sprs_csr *, double *, double *, int, const double *, double *, double *, const int *, bool, const Fin *);
template int dist_spmv<cplx>(const sprs_csr *, cplx *, cplx *, int, const cplx *, cplx *, cplx *, const int *, bool, const Fin *);
template int dist_spmv<float>(const sprs_csr *, float *, float *, int, const float *, float *, float *, const int *, bool, const Fin *);
template int dist_spmv<cplxf>(const sprs_csr *, cplxf *, cplxf *, int, const cplxf *, cplxf *, cplxf *, const int *, bool, const Fin *);

}  // namespace sprs

using namespace sprs;

namespace {
template <class T>
int dist_csr_create(sprs_comm *comm, int64_t n_local, int64_t n_ext, int64_t nnz, const int32_t *d_rp,
                    const int32_t *d_ci, const T *d_val, int adopt, int n_peers, const int32_t *peer_rank,
                    const int64_t *send_off, const int32_t *send_idx_dev, const int64_t *recv_off, sprs_csr **out) {
    if (!comm || !out || n_peers < 0 || n_ext < n_local) return SPRS_INVALID_ARGUMENT;
    if (n_peers > 0 && (!peer_rank || !send_off || !recv_off)) return SPRS_INVALID_ARGUMENT;
    sprs_ctx *c = comm->ctx;
    // host-side validation of the exchange plan: the pack kernel and the receives index with it
    for (int p = 0; p < n_peers; ++p) {
        if (peer_rank[p] < 0 || peer_rank[p] >= comm->world) return SPRS_INVALID_ARGUMENT;
        if (send_off[p + 1] < send_off[p] || recv_off[p + 1] < recv_off[p]) return SPRS_INVALID_ARGUMENT;
    }
    if (n_peers > 0 && (send_off[0] != 0 || recv_off[0] != 0 || recv_off[n_peers] != n_ext - n_local))
        return SPRS_INVALID_ARGUMENT;
    if (n_peers == 0 && n_ext != n_local) return SPRS_INVALID_ARGUMENT;
    const int64_t n_send = n_peers ? send_off[n_peers] : 0;
    if (n_send > 0) {
        if (!send_idx_dev) return SPRS_INVALID_ARGUMENT;
        std::vector<int32_t> h((size_t)n_send);
        SPRS_HIP_TRY(c, hipMemcpy(h.data(), send_idx_dev, sizeof(int32_t) * (size_t)n_send, hipMemcpyDeviceToHost));
        for (int32_t v : h)
            if (v < 0 || v >= n_local) return SPRS_INVALID_ARGUMENT;
    }
    sprs_csr *A = nullptr;
    int st;
    if constexpr (dtype_of<T>::value == DT_Z) st = sprs_csr_create_dev_z(c, n_local, n_ext, nnz, d_rp, d_ci, (const sprs_c64 *)d_val, adopt, &A);
    else if constexpr (dtype_of<T>::value == DT_C) st = sprs_csr_create_dev_c(c, n_local, n_ext, nnz, d_rp, d_ci, (const sprs_c32 *)d_val, adopt, &A);
    else if constexpr (dtype_of<T>::value == DT_S) st = sprs_csr_create_dev_s(c, n_local, n_ext, nnz, d_rp, d_ci, d_val, adopt, &A);
    else st = sprs_csr_create_dev_d(c, n_local, n_ext, nnz, d_rp, d_ci, d_val, adopt, &A);
    if (st != SPRS_OK) return st;
    sprs_dist_info *D = new sprs_dist_info();
    D->comm = comm; D->n_local = n_local; D->n_ext = n_ext;
    D->peer.assign(peer_rank, peer_rank + n_peers);
    D->send_off.assign(1, 0); D->recv_off.assign(1, 0);
    if (n_peers) { D->send_off.assign(send_off, send_off + n_peers + 1); D->recv_off.assign(recv_off, recv_off + n_peers + 1); }
    A->dist = D;
    if (n_send > 0) {
        if (hipMalloc((void **)&D->send_idx, sizeof(int32_t) * (size_t)n_send) != hipSuccess ||
            hipMalloc(&D->send_buf, sizeof(T) * (size_t)n_send) != hipSuccess ||
            hipMemcpy(D->send_idx, send_idx_dev, sizeof(int32_t) * (size_t)n_send, hipMemcpyDeviceToDevice) != hipSuccess) {
            sprs_csr_destroy(A);
            return SPRS_ERR_HIP;
        }
    }
    // interior / boundary split of the row blocks: a block is "boundary" if any of its rows reads
    // the halo tail (column >= n_local).  Interior blocks are multiplied while the halo travels.
    if (n_peers > 0 && c->halo_overlap) {
        std::vector<int32_t> lo, hi;
        st = rowblk_spans(A, lo, hi);
        if (st != SPRS_OK) { sprs_csr_destroy(A); return st; }
        std::vector<int32_t> oi, ob, oiw, obw;
        const bool wide = A->dict && A->dict->wide_desc;
        if (wide) {
            // classify PAIRS of consecutive 64-row blocks (= the 128-row blocks of the two-rows-per-lane kernel) so
            // that both kernels can run the same split: a pair is boundary if either half reads the halo
            for (int j = 0; 2 * j < A->n_rowblk; ++j) {
                const int b0 = 2 * j, b1 = std::min(2 * j + 1, A->n_rowblk - 1);
                const bool bnd = hi[b0] >= n_local || hi[b1] >= n_local;
                (bnd ? obw : oiw).push_back(j);
                for (int b = b0; b <= b1; ++b) (bnd ? ob : oi).push_back(b);
            }
        } else {
            for (int b = 0; b < A->n_rowblk; ++b) (hi[b] >= n_local ? ob : oi).push_back(b);
        }
        if (!oi.empty() && !ob.empty() && ob.size() * 2 <= (size_t)A->n_rowblk) {
            if (wide) {
                bool okw = hipMalloc((void **)&D->order_int_w, sizeof(int32_t) * oiw.size()) == hipSuccess &&
                           hipMalloc((void **)&D->order_bnd_w, sizeof(int32_t) * obw.size()) == hipSuccess &&
                           hipMemcpy(D->order_int_w, oiw.data(), sizeof(int32_t) * oiw.size(), hipMemcpyHostToDevice) == hipSuccess &&
                           hipMemcpy(D->order_bnd_w, obw.data(), sizeof(int32_t) * obw.size(), hipMemcpyHostToDevice) == hipSuccess;
                if (!okw) { sprs_csr_destroy(A); return SPRS_ERR_HIP; }
                D->n_int_w = (int32_t)oiw.size(); D->n_bnd_w = (int32_t)obw.size();
            }
            bool ok = hipMalloc((void **)&D->order_int, sizeof(int32_t) * oi.size()) == hipSuccess &&
                      hipMalloc((void **)&D->order_bnd, sizeof(int32_t) * ob.size()) == hipSuccess &&
                      hipMemcpy(D->order_int, oi.data(), sizeof(int32_t) * oi.size(), hipMemcpyHostToDevice) == hipSuccess &&
                      hipMemcpy(D->order_bnd, ob.data(), sizeof(int32_t) * ob.size(), hipMemcpyHostToDevice) == hipSuccess &&
                      hipStreamCreateWithFlags(&D->comm_stream, hipStreamNonBlocking) == hipSuccess &&
                      hipEventCreateWithFlags(&D->ev_pack, hipEventDisableTiming) == hipSuccess &&
                      hipEventCreateWithFlags(&D->ev_halo, hipEventDisableTiming) == hipSuccess;
            if (!ok) { sprs_csr_destroy(A); return SPRS_ERR_HIP; }
            D->n_int = (int32_t)oi.size(); D->n_bnd = (int32_t)ob.size();
            // LDS-window tiles (spmv_dict.hip) of the stream this handle multiplies with: the interior launch keeps the tiles
            // that hold interior rows only (the last plane of a slab can look exactly like the stencil — its halo columns sit
            // one plane behind the local rows — and must wait for the halo) and walks the other interior blocks one by one
            if (A->dict && sprs::tile_plan_used(A)) {
                const bool off_stream = sprs::dict_mode(A) == 1;
                const sprs_tile_plan &TP = off_stream ? A->dict->tile_off : A->dict->tile_pair;
                const int nw = (A->n_rowblk + 1) / 2, TB = sprs::tile_blocks();
                std::vector<char> bnd128((size_t)nw, 0), in_int_tile((size_t)nw, 0);
                for (int j = 0; j < nw; ++j) {
                    const int b0 = 2 * j, b1 = std::min(2 * j + 1, A->n_rowblk - 1);
                    bnd128[(size_t)j] = hi[b0] >= n_local || hi[b1] >= n_local;
                }
                std::vector<int32_t> list, xstart(9, 0), left;
                for (int xq = 0; xq < 8; ++xq) {
                    xstart[(size_t)xq] = (int32_t)(list.size() / 2);
                    for (int t = TP.h_xstart[(size_t)xq]; t < TP.h_xstart[(size_t)xq + 1]; ++t) {
                        const int b0 = TP.h_list[(size_t)2 * t];
                        bool inside = true;
                        for (int q = 0; q < TB; ++q) inside = inside && !bnd128[(size_t)(b0 + q)];
                        if (!inside) continue;
                        list.push_back(b0); list.push_back(TP.h_list[(size_t)2 * t + 1]);
                        for (int q = 0; q < TB; ++q) in_int_tile[(size_t)(b0 + q)] = 1;
                    }
                }
                xstart[8] = (int32_t)(list.size() / 2);
                if (off_stream) { for (int32_t b : oi) if (!in_int_tile[(size_t)(b / 2)]) left.push_back(b); }
                else { for (int32_t j : oiw) if (!in_int_tile[(size_t)j]) left.push_back(j); }
                if (list.size() / 2 >= 8 && (off_stream || wide)) {
                    sprs_tile_plan &TI = D->tile_int;
                    bool okt = hipMalloc((void **)&TI.list, sizeof(int32_t) * list.size()) == hipSuccess &&
                               hipMalloc((void **)&TI.xstart, sizeof(int32_t) * 9) == hipSuccess &&
                               hipMalloc((void **)&TI.left, sizeof(int32_t) * std::max<size_t>(left.size(), 1)) == hipSuccess &&
                               hipMemcpy(TI.list, list.data(), sizeof(int32_t) * list.size(), hipMemcpyHostToDevice) == hipSuccess &&
                               hipMemcpy(TI.xstart, xstart.data(), sizeof(int32_t) * 9, hipMemcpyHostToDevice) == hipSuccess &&
                               (left.empty() || hipMemcpy(TI.left, left.data(), sizeof(int32_t) * left.size(), hipMemcpyHostToDevice) == hipSuccess);
                    if (!okt) { sprs_csr_destroy(A); return SPRS_ERR_HIP; }
                    TI.n_tile = (int)(list.size() / 2); TI.n_left = (int)left.size();
                    TI.ul = TP.ul; TI.fl = TP.fl; TI.fh = TP.fh; TI.w = TP.w;
                    for (int t = 0; t < 8; ++t) { TI.off[t] = TP.off[t]; TI.val[t] = TP.val[t]; }
                    D->tile_int_off = off_stream;
                }
            }
        }
    }
    *out = A;
    return SPRS_OK;
}
template <class T>
int dist_csr_create_allgather(sprs_comm *comm, int64_t n_local, int64_t slice, int64_t nnz, const int32_t *d_rp,
                              const int32_t *d_ci, const T *d_val, int adopt, sprs_csr **out) {
    if (!comm || !out || slice < n_local || n_local < 0) return SPRS_INVALID_ARGUMENT;
    if ((int64_t)comm->world * slice >= INT32_MAX) return SPRS_INVALID_ARGUMENT;
    sprs_ctx *c = comm->ctx;
    sprs_csr *A = nullptr;
    const int64_t n_ext = (int64_t)comm->world * slice;
    int st;
    if constexpr (dtype_of<T>::value == DT_Z) st = sprs_csr_create_dev_z(c, n_local, n_ext, nnz, d_rp, d_ci, (const sprs_c64 *)d_val, adopt, &A);
    else if constexpr (dtype_of<T>::value == DT_C) st = sprs_csr_create_dev_c(c, n_local, n_ext, nnz, d_rp, d_ci, (const sprs_c32 *)d_val, adopt, &A);
    else if constexpr (dtype_of<T>::value == DT_S) st = sprs_csr_create_dev_s(c, n_local, n_ext, nnz, d_rp, d_ci, d_val, adopt, &A);
    else st = sprs_csr_create_dev_d(c, n_local, n_ext, nnz, d_rp, d_ci, d_val, adopt, &A);
    if (st != SPRS_OK) return st;
    sprs_dist_info *D = new sprs_dist_info();
    D->comm = comm; D->n_local = n_local; D->n_ext = n_ext; D->ag_slice = slice;
    D->send_off.assign(1, 0); D->recv_off.assign(1, 0);
    A->dist = D;
    if (hipMalloc(&D->ag_buf, sizeof(T) * (size_t)n_ext) != hipSuccess ||
        hipMemset(D->ag_buf, 0, sizeof(T) * (size_t)n_ext) != hipSuccess) { sprs_csr_destroy(A); return SPRS_ERR_HIP; }
    *out = A;
    return SPRS_OK;
}
// ---------------------------------------------------------------------------------------------
// Exchange plan from GLOBAL column indices, behind the C ABI (a Rust host has no numpy): the plan that
// sprsolve_amd/partition.py derives on the host, derived here on the device and over RCCL.
//   1. mark[g] = 1 for every remote column g the local rows reference (one byte per global column);
//   2. exclusive scan of mark -> the position of g in the sorted list of needed columns == its offset in the halo
//      tail (peers ascending, indices ascending within a peer == global ascending order);
//   3. columns renumbered in place: owned g -> g - r0, remote g -> n_local + pos[g];
//   4. the needed list is split by owner on the host (it is the halo, small), the per-peer counts travel in one
//      ncclAllGather, the index lists in one group of ncclSend / ncclRecv; what arrives is what this rank must pack.
// Temporary device memory: 5 bytes per GLOBAL column (cfg 5: 250 MB of the 288 GB), freed before returning.
// PLAN_TRY frees the temporaries on every error path.
#define PLAN_TRY(expr) do { const int st__ = (expr); if (st__ != SPRS_OK) { cleanup(); return st__; } } while (0)
#define PLAN_HIP(expr) do { const hipError_t e__ = (expr); if (e__ != hipSuccess) { snprintf(c->err, sizeof(c->err), "%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(e__)); cleanup(); return SPRS_ERR_HIP; } } while (0)
#define PLAN_NCCL(expr) do { const ncclResult_t e__ = (expr); if (e__ != ncclSuccess) { snprintf(c->err, sizeof(c->err), "%s:%d: %s -> %s", __FILE__, __LINE__, #expr, rccl().GetErrorString(e__)); cleanup(); return SPRS_ERR_RCCL; } } while (0)

__global__ __launch_bounds__(sprs::BLOCK) void plan_mark_kernel(int64_t nnz, const int32_t *__restrict__ col, int32_t r0, int32_t r1,
                                                                int32_t n_global, uint8_t *__restrict__ mark, int *__restrict__ bad) {
    int local = 0;
    for (int64_t k = (int64_t)blockIdx.x * sprs::BLOCK + threadIdx.x; k < nnz; k += (int64_t)gridDim.x * sprs::BLOCK) {
        const int32_t g = col[k];
        if (g < 0 || g >= n_global) { local = 1; continue; }
        if (g < r0 || g >= r1) mark[g] = 1;      // benign race: every writer stores the same byte
    }
    if (local) atomicOr(bad, 1);
}
// range check alone (the all-gather numbering renumbers after the ranks have agreed that every input is valid)
__global__ __launch_bounds__(sprs::BLOCK) void plan_check_kernel(int64_t nnz, const int32_t *__restrict__ col, int32_t n_global, int *__restrict__ bad) {
    int local = 0;
    for (int64_t k = (int64_t)blockIdx.x * sprs::BLOCK + threadIdx.x; k < nnz; k += (int64_t)gridDim.x * sprs::BLOCK) {
        const int32_t g = col[k];
        local |= (g < 0) | (g >= n_global);
    }
    if (local) atomicOr(bad, 1);
}
__global__ __launch_bounds__(sprs::BLOCK) void plan_compact_kernel(int32_t n_global, const uint8_t *__restrict__ mark,
                                                                   const int32_t *__restrict__ pos, int32_t *__restrict__ uniq) {
    for (int64_t g = (int64_t)blockIdx.x * sprs::BLOCK + threadIdx.x; g < n_global; g += (int64_t)gridDim.x * sprs::BLOCK)
        if (mark[g]) uniq[pos[g]] = (int32_t)g;
}
__global__ __launch_bounds__(sprs::BLOCK) void plan_renumber_kernel(int64_t nnz, const int32_t *__restrict__ col_in, int32_t *__restrict__ col_out,
                                                                    int32_t r0, int32_t r1, const int32_t *__restrict__ pos) {
    const int32_t n_local = r1 - r0;
    for (int64_t k = (int64_t)blockIdx.x * sprs::BLOCK + threadIdx.x; k < nnz; k += (int64_t)gridDim.x * sprs::BLOCK) {
        const int32_t g = col_in[k];
        col_out[k] = (g < r0 || g >= r1) ? n_local + pos[g] : g - r0;
    }
}
// all-gather numbering: [rank 0 slice | rank 1 slice | ...], slices padded to `slice`
__global__ __launch_bounds__(sprs::BLOCK) void plan_renumber_ag_kernel(int64_t nnz, const int32_t *__restrict__ col_in, int32_t *__restrict__ col_out,
                                                                       int world, const int64_t *__restrict__ starts, int64_t slice,
                                                                       int *__restrict__ bad) {
    int local = 0;
    for (int64_t k = (int64_t)blockIdx.x * sprs::BLOCK + threadIdx.x; k < nnz; k += (int64_t)gridDim.x * sprs::BLOCK) {
        const int64_t g = col_in[k];
        if (g < 0 || g >= starts[world]) { local = 1; continue; }
        int lo = 0, hi = world;                       // owner = last rank with starts[rank] <= g
        while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (starts[mid] <= g) lo = mid; else hi = mid; }
        col_out[k] = (int32_t)((int64_t)lo * slice + (g - starts[lo]));
    }
    if (local) atomicOr(bad, 1);
}
__global__ __launch_bounds__(sprs::BLOCK) void plan_shift_kernel(int64_t n, int32_t *__restrict__ idx, int32_t r0) {
    for (int64_t k = (int64_t)blockIdx.x * sprs::BLOCK + threadIdx.x; k < n; k += (int64_t)gridDim.x * sprs::BLOCK) idx[k] -= r0;
}
static inline int plan_grid(int64_t n) { return (int)std::max<int64_t>(1, std::min<int64_t>((n + sprs::BLOCK - 1) / sprs::BLOCK, 2048)); }

// Collective error contract: the function is collective over the communicator, so a rank must never leave it while the
// others are still heading for an exchange.  Everything a rank can find wrong BY ITSELF (arguments, an out-of-range
// column, an allocation) only sets `err`; the flags travel with the counts in the first ncclAllGather and every rank
// returns the largest one; the two later rank-local failure points (the send list's allocation, the handle's creation)
// are agreed on the same way before anyone proceeds.  The caller's column array (adopt != 0) is renumbered only after the
// first two agreements.  An RCCL call that fails is a communicator failure: it is returned at once.
template <class T>
int dist_csr_create_global(sprs_comm *comm, const int64_t *row_starts, int64_t nnz, const int32_t *d_rp, int32_t *d_ci_global,
                           const T *d_val, int adopt, int exchange, sprs_csr **out) {
    if (!comm || !out) return SPRS_INVALID_ARGUMENT;         // no communicator / nowhere to report: nothing collective can be done
    *out = nullptr;
    sprs_ctx *c = comm->ctx;
    const int world = comm->world, rank = comm->rank;
    CtxLock lock(c);
    int err = SPRS_OK;                                         // this rank's own verdict so far
    auto fail = [&](int code) { if (err == SPRS_OK) err = code; };
    if (!row_starts || !d_rp || nnz < 0 || (nnz > 0 && (!d_ci_global || !d_val)) || (exchange != 0 && exchange != 1)) fail(SPRS_INVALID_ARGUMENT);
    if (err == SPRS_OK) {
        for (int r = 0; r < world; ++r)
            if (row_starts[r + 1] < row_starts[r]) fail(SPRS_INVALID_ARGUMENT);
        if (row_starts[0] != 0 || row_starts[world] >= INT32_MAX) fail(SPRS_INVALID_ARGUMENT);
    }
    const int32_t n_global = err == SPRS_OK ? (int32_t)row_starts[world] : 0, r0 = err == SPRS_OK ? (int32_t)row_starts[rank] : 0,
                  r1 = err == SPRS_OK ? (int32_t)row_starts[rank + 1] : 0;
    const int64_t n_local = r1 - r0;
    if (hipSetDevice(c->device) != hipSuccess) fail(SPRS_ERR_HIP);
    uint8_t *mark = nullptr; int32_t *pos = nullptr, *uniq = nullptr, *send_idx = nullptr, *counts = nullptr, *col_copy = nullptr;
    int64_t *d_starts = nullptr; void *scan_tmp = nullptr; int *d_bad = nullptr;
    auto cleanup = [&]() {
        for (void *q : {(void *)mark, (void *)pos, (void *)uniq, (void *)send_idx, (void *)counts, (void *)d_starts, scan_tmp, (void *)d_bad})
            if (q) (void)hipFree(q);
        if (col_copy) (void)hipFree(col_copy);
    };
    // rank-local HIP step: a failure is recorded, the text kept, and the remaining local steps are skipped
#define PLAN_LOCAL(expr) do { if (err == SPRS_OK) { const hipError_t e__ = (expr); if (e__ != hipSuccess) { snprintf(c->err, sizeof(c->err), "%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(e__)); err = SPRS_ERR_HIP; } } } while (0)
    // agreement: all-gather one status word per rank, everybody leaves with the largest
    std::vector<int32_t> h_flags((size_t)world);
    auto agree = [&](int mine, int *verdict) -> int {
        int32_t *d_flags = counts;       // [world + 1] scratch at the head of `counts` is free at every call site
        const int32_t m = mine;
        if (hipMemcpyAsync(d_flags + world, &m, sizeof(int32_t), hipMemcpyHostToDevice, c->stream) != hipSuccess) return SPRS_ERR_HIP;
        if (rccl().AllGather(d_flags + world, d_flags, 1, ncclInt32, (ncclComm_t)comm->nccl, c->stream) != ncclSuccess) return SPRS_ERR_RCCL;
        if (hipMemcpyAsync(h_flags.data(), d_flags, sizeof(int32_t) * (size_t)world, hipMemcpyDeviceToHost, c->stream) != hipSuccess) return SPRS_ERR_HIP;
        if (hipStreamSynchronize(c->stream) != hipSuccess) return SPRS_ERR_HIP;
        int v = SPRS_OK;
        for (int q = 0; q < world; ++q) v = std::max(v, (int)h_flags[(size_t)q]);
        *verdict = v;
        return SPRS_OK;
    };
    // `counts` carries every exchange of this function; without it this rank cannot even take part in the agreement, and
    // its peers would wait for it: that one allocation failing is reported at once (as any RCCL failure is)
    if (hipMalloc((void **)&counts, sizeof(int32_t) * ((size_t)world * (size_t)(world + 2) + 8)) != hipSuccess) {
        snprintf(c->err, sizeof(c->err), "%s:%d: hipMalloc of the plan's exchange buffer failed", __FILE__, __LINE__);
        cleanup();
        return SPRS_ERR_HIP;
    }
    PLAN_LOCAL(hipMalloc((void **)&d_bad, sizeof(int)));
    PLAN_LOCAL(hipMemsetAsync(d_bad, 0, sizeof(int), c->stream));
    int32_t *col_out = d_ci_global;
    if (!adopt && nnz > 0) {      // the caller keeps its global indices: renumber into a private copy the handle will own
        PLAN_LOCAL(hipMalloc((void **)&col_copy, sizeof(int32_t) * (size_t)nnz));
        col_out = col_copy;
    }
    int bad = 0, verdict = SPRS_OK;
    if (exchange == 1) {
        int64_t slice = 0;
        if (err == SPRS_OK) {
            for (int r = 0; r < world; ++r) slice = std::max(slice, row_starts[r + 1] - row_starts[r]);
            slice += slice & 1;                        // every slice stays 16-byte aligned
            if ((int64_t)world * slice >= INT32_MAX) fail(SPRS_INVALID_ARGUMENT);
        }
        PLAN_LOCAL(hipMalloc((void **)&d_starts, sizeof(int64_t) * (size_t)(world + 1)));
        PLAN_LOCAL(hipMemcpyAsync(d_starts, row_starts, sizeof(int64_t) * (size_t)(world + 1), hipMemcpyHostToDevice, c->stream));
        // range check only (output to a scratch-free pass): the renumbering itself waits for the agreement
        if (err == SPRS_OK && nnz > 0) {
            hipLaunchKernelGGL(plan_check_kernel, dim3(plan_grid(nnz)), dim3(BLOCK), 0, c->stream, nnz, d_ci_global, n_global, d_bad);
            PLAN_LOCAL(hipGetLastError());
        }
        PLAN_LOCAL(hipMemcpyAsync(&bad, d_bad, sizeof(int), hipMemcpyDeviceToHost, c->stream));
        PLAN_LOCAL(hipStreamSynchronize(c->stream));
        if (err == SPRS_OK && bad) fail(SPRS_INVALID_ARGUMENT);
        { const int st = agree(err, &verdict); if (st != SPRS_OK) { cleanup(); return st; } }
        if (verdict != SPRS_OK) { cleanup(); return err != SPRS_OK ? err : verdict; }
        int st = SPRS_OK;
        if (nnz > 0) {
            hipLaunchKernelGGL(plan_renumber_ag_kernel, dim3(plan_grid(nnz)), dim3(BLOCK), 0, c->stream, nnz, d_ci_global, col_out, world,
                               d_starts, slice, d_bad);
            if (hipGetLastError() != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess) st = SPRS_ERR_HIP;
        }
        // a private copy is handed over to the handle (adopt = 0 semantics: copied again inside; keep it simple and correct)
        if (st == SPRS_OK) st = dist_csr_create_allgather<T>(comm, n_local, slice, nnz, d_rp, col_out, d_val, adopt, out);
        { const int s2 = agree(st, &verdict); if (s2 != SPRS_OK) { if (*out) { sprs_csr_destroy(*out); *out = nullptr; } cleanup(); return s2; } }
        if (verdict != SPRS_OK && *out) { sprs_csr_destroy(*out); *out = nullptr; }
        cleanup();
        return st != SPRS_OK ? st : verdict;
    }
    // ---- 1. mark, 2. scan, compact
    PLAN_LOCAL(hipMalloc((void **)&mark, (size_t)n_global + 16));
    PLAN_LOCAL(hipMalloc((void **)&pos, sizeof(int32_t) * ((size_t)n_global + 1)));
    PLAN_LOCAL(hipMemsetAsync(mark, 0, (size_t)n_global + 16, c->stream));
    if (err == SPRS_OK && nnz > 0) {
        hipLaunchKernelGGL(plan_mark_kernel, dim3(plan_grid(nnz)), dim3(BLOCK), 0, c->stream, nnz, d_ci_global, r0, r1, n_global, mark, d_bad);
        PLAN_LOCAL(hipGetLastError());
    }
    size_t tmp_bytes = 0;
    PLAN_LOCAL(rocprim::exclusive_scan(nullptr, tmp_bytes, mark, pos, (int32_t)0, (size_t)n_global + 1, rocprim::plus<int32_t>(), c->stream));
    PLAN_LOCAL(hipMalloc(&scan_tmp, tmp_bytes ? tmp_bytes : 16));
    PLAN_LOCAL(rocprim::exclusive_scan(scan_tmp, tmp_bytes, mark, pos, (int32_t)0, (size_t)n_global + 1, rocprim::plus<int32_t>(), c->stream));
    int32_t n_halo = 0;
    PLAN_LOCAL(hipMemcpyAsync(&n_halo, pos + n_global, sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));   // mark[n_global] == 0: total
    PLAN_LOCAL(hipMemcpyAsync(&bad, d_bad, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    PLAN_LOCAL(hipStreamSynchronize(c->stream));
    if (err == SPRS_OK && bad) fail(SPRS_INVALID_ARGUMENT);
    if (err != SPRS_OK) n_halo = 0;
    // the renumbered columns n_local + pos[g] are int32 as well
    if (err == SPRS_OK && n_local + (int64_t)n_halo >= INT32_MAX) fail(SPRS_INVALID_ARGUMENT);
    std::vector<int32_t> h_uniq((size_t)n_halo);
    if (n_halo > 0) {
        PLAN_LOCAL(hipMalloc((void **)&uniq, sizeof(int32_t) * (size_t)n_halo));
        if (err == SPRS_OK) {
            hipLaunchKernelGGL(plan_compact_kernel, dim3(plan_grid(n_global)), dim3(BLOCK), 0, c->stream, n_global, mark, pos, uniq);
            PLAN_LOCAL(hipGetLastError());
        }
        PLAN_LOCAL(hipMemcpyAsync(h_uniq.data(), uniq, sizeof(int32_t) * (size_t)n_halo, hipMemcpyDeviceToHost, c->stream));
        PLAN_LOCAL(hipStreamSynchronize(c->stream));
    }
    // ---- 3. who needs what: counts (and this rank's verdict so far) in one all-gather
    std::vector<int32_t> need(world + 1, 0), need_off(world + 1, 0);
    if (err == SPRS_OK) {
        int p = 0;
        for (int32_t i = 0; i < n_halo; ++i) {
            while (h_uniq[(size_t)i] >= row_starts[p + 1]) ++p;      // ascending ids: owners ascend too
            need[p]++;
        }
        for (int q = 0; q < world; ++q) need_off[q + 1] = need_off[q] + need[q];
    }
    need[world] = err;                                               // the extra element: this rank's status
    const size_t W1 = (size_t)world + 1;
    int32_t *d_mine = counts + W1 * (size_t)world;                   // [world + 1], behind the gathered matrix
    if (hipMemcpyAsync(d_mine, need.data(), sizeof(int32_t) * W1, hipMemcpyHostToDevice, c->stream) != hipSuccess) { cleanup(); return SPRS_ERR_HIP; }
    PLAN_NCCL(rccl().AllGather(d_mine, counts, W1, ncclInt32, (ncclComm_t)comm->nccl, c->stream));
    std::vector<int32_t> M(W1 * (size_t)world);                      // M[q * (world + 1) + p]: rank q needs that many entries of rank p; [.. + world]: rank q's status
    if (hipMemcpyAsync(M.data(), counts, sizeof(int32_t) * M.size(), hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
        hipStreamSynchronize(c->stream) != hipSuccess) { cleanup(); return SPRS_ERR_HIP; }
    for (int q = 0; q < world; ++q) verdict = std::max(verdict, (int)M[(size_t)q * W1 + (size_t)world]);
    // every rank checks EVERY pair with the same data, so all reach the same conclusion
    if (verdict == SPRS_OK)
        for (int q = 0; q < world && verdict == SPRS_OK; ++q)
            for (int p = 0; p < world; ++p) {
                const int32_t cnt = M[(size_t)q * W1 + (size_t)p];
                if (cnt < 0 || (int64_t)cnt > row_starts[p + 1] - row_starts[p] || (p == q && cnt != 0)) { verdict = SPRS_INVALID_ARGUMENT; break; }
            }
    if (verdict != SPRS_OK) { cleanup(); return err != SPRS_OK ? err : verdict; }
    std::vector<int64_t> send_off_all(world + 1, 0);
    for (int q = 0; q < world; ++q) send_off_all[q + 1] = send_off_all[q] + (q == rank ? 0 : M[(size_t)q * W1 + (size_t)rank]);
    const int64_t n_send = send_off_all[world];
    // ---- 4. the send list's buffer is the last rank-local allocation in front of an exchange: agree on it
    int st_local = SPRS_OK;
    if (n_send > 0 && hipMalloc((void **)&send_idx, sizeof(int32_t) * (size_t)n_send) != hipSuccess) st_local = SPRS_ERR_HIP;
    { const int st = agree(st_local, &verdict); if (st != SPRS_OK) { cleanup(); return st; } }
    if (verdict != SPRS_OK) { cleanup(); return st_local != SPRS_OK ? st_local : verdict; }
    // ---- 5. renumber the columns (in place when adopted: only now, with every rank's input accepted), exchange the lists
    if (nnz > 0) hipLaunchKernelGGL(plan_renumber_kernel, dim3(plan_grid(nnz)), dim3(BLOCK), 0, c->stream, nnz, d_ci_global, col_out, r0, r1, pos);
    PLAN_NCCL(rccl().GroupStart());
    for (int q = 0; q < world; ++q) {
        if (q == rank) continue;
        if (need[q] > 0) PLAN_NCCL(rccl().Send(uniq + need_off[q], (size_t)need[q], ncclInt32, q, (ncclComm_t)comm->nccl, c->stream));
        const int64_t cnt = send_off_all[q + 1] - send_off_all[q];
        if (cnt > 0) PLAN_NCCL(rccl().Recv(send_idx + send_off_all[q], (size_t)cnt, ncclInt32, q, (ncclComm_t)comm->nccl, c->stream));
    }
    PLAN_NCCL(rccl().GroupEnd());
    int st = SPRS_OK;
    if (n_send > 0) hipLaunchKernelGGL(plan_shift_kernel, dim3(plan_grid(n_send)), dim3(BLOCK), 0, c->stream, n_send, send_idx, r0);   // global -> local
    if (hipGetLastError() != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess) st = SPRS_ERR_HIP;
    // ---- one peer list for both directions
    std::vector<int32_t> peers; std::vector<int64_t> s_off(1, 0), r_off(1, 0);
    for (int q = 0; q < world; ++q) {
        const int64_t ns = send_off_all[q + 1] - send_off_all[q], nr = q == rank ? 0 : need[q];
        if (ns == 0 && nr == 0) continue;
        peers.push_back(q); s_off.push_back(s_off.back() + ns); r_off.push_back(r_off.back() + nr);
    }
    // send_idx is grouped by ascending peer with no gaps: exactly the layout dist_csr_create expects
    if (st == SPRS_OK)
        st = dist_csr_create<T>(comm, n_local, n_local + n_halo, nnz, d_rp, col_out, d_val, adopt, (int)peers.size(), peers.data(),
                                s_off.data(), send_idx, r_off.data(), out);
    // the handle exists on all ranks or on none: a solve on it is collective too
    { const int s2 = agree(st, &verdict); if (s2 != SPRS_OK) { if (*out) { sprs_csr_destroy(*out); *out = nullptr; } cleanup(); return s2; } }
    if (verdict != SPRS_OK && *out) { sprs_csr_destroy(*out); *out = nullptr; }
    cleanup();
    return st != SPRS_OK ? st : verdict;
#undef PLAN_LOCAL
}
#undef PLAN_TRY
#undef PLAN_HIP
#undef PLAN_NCCL
}  // namespace

extern "C" {

int sprs_dist_csr_info(const sprs_csr *A, int64_t *n_local, int64_t *n_ext, int *n_peers, int64_t *send_entries, int64_t *recv_entries) {
    if (!A || !A->dist) return SPRS_INVALID_ARGUMENT;
    const sprs_dist_info *D = A->dist;
    if (n_local) *n_local = D->n_local;
    if (n_ext) *n_ext = D->n_ext;
    if (n_peers) *n_peers = (int)D->peer.size();
    if (send_entries) *send_entries = D->ag_slice > 0 ? D->ag_slice : D->send_off.back();
    if (recv_entries) *recv_entries = D->ag_slice > 0 ? D->ag_slice * (D->comm->world - 1) : D->recv_off.back();
    return SPRS_OK;
}
int sprs_dist_csr_peers(const sprs_csr *A, int cap, int32_t *peer_rank, int64_t *send_off, int64_t *recv_off) {
    if (!A || !A->dist || cap < (int)A->dist->peer.size() || (cap > 0 && (!peer_rank || !send_off || !recv_off))) return SPRS_INVALID_ARGUMENT;
    const sprs_dist_info *D = A->dist;
    CtxLock lock(A->ctx);          // host-side plan data only; the lock orders the read with a concurrent destroy / create on the context
    for (size_t p = 0; p < D->peer.size(); ++p) peer_rank[p] = D->peer[p];
    if (send_off) for (size_t p = 0; p < D->send_off.size(); ++p) send_off[p] = D->send_off[p];
    if (recv_off) for (size_t p = 0; p < D->recv_off.size(); ++p) recv_off[p] = D->recv_off[p];
    return SPRS_OK;
}
int sprs_dist_csr_send_idx(const sprs_csr *A, int64_t cap, int32_t *send_idx_host) {
    if (!A || !A->dist || !send_idx_host) return SPRS_INVALID_ARGUMENT;
    const sprs_dist_info *D = A->dist;
    const int64_t n = D->send_off.back();
    if (cap < n) return SPRS_INVALID_ARGUMENT;
    if (n > 0) {
        sprs_ctx *c = A->ctx;
        CtxLock lock(c);
        SPRS_HIP_TRY(c, hipSetDevice(c->device));
        SPRS_HIP_TRY(c, hipMemcpyAsync(send_idx_host, D->send_idx, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToHost, c->stream));
        SPRS_HIP_TRY(c, hipStreamSynchronize(c->stream));
    }
    return SPRS_OK;
}

int sprs_comm_unique_id(void *id128) {
    if (!id128) return SPRS_INVALID_ARGUMENT;
    if (!rccl().ok) return SPRS_ERR_RCCL;
    ncclUniqueId id;
    if (rccl().GetUniqueId(&id) != ncclSuccess) return SPRS_ERR_RCCL;
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId size");
    memcpy(id128, &id, 128);
    return SPRS_OK;
}

static int p2p_setup(sprs_comm *c);
static void p2p_release(sprs_comm *c);
int sprs_comm_create(sprs_ctx *ctx, int world, int rank, const void *id128, sprs_comm **out) {
    if (!ctx || !out || !id128 || world < 1 || rank < 0 || rank >= world) return SPRS_INVALID_ARGUMENT;
    *out = nullptr;
    if (!rccl().ok) {
        snprintf(ctx->err, sizeof(ctx->err), "librccl.so.1 could not be loaded: %s", dlerror());
        return SPRS_ERR_RCCL;
    }
    SPRS_HIP_TRY(ctx, hipSetDevice(ctx->device));
    ncclUniqueId id;
    memcpy(&id, id128, 128);
    ncclComm_t nc = nullptr;
    SPRS_NCCL_TRY(ctx, rccl().CommInitRank(&nc, world, id, rank));
    sprs_comm *c = new sprs_comm();
    c->ctx = ctx; c->nccl = nc; c->world = world; c->rank = rank;
    if (const int st = p2p_setup(c)) { p2p_release(c); (void)rccl().CommDestroy(nc); delete c; return st; }
    *out = c;
    return SPRS_OK;
}

// Peer-to-peer mailboxes (internal.hpp, P2pBox): COLLECTIVE.  Every rank allocates its mailbox in uncached device memory, exports
// it (hipIpcGetMemHandle), the 64-byte handles travel in one ncclAllGather together with a "so far so good" byte, every rank opens
// the others' (same node: hipIpcOpenMemHandle maps the peer's HBM over xGMI; the dmabuf IPC mode HSA_ENABLE_IPC_MODE_LEGACY=0 is
// what this pool's driver supports) and a last one-word all-reduce agrees on the outcome: either EVERY rank has every mailbox
// mapped and the hand-offs may use them, or none does and they stay on ncclAllReduce.  Nothing here is an error: a rank that cannot
// export or map (another node, an IPC-less driver) just votes no.
static void p2p_release(sprs_comm *c) {
    for (int q = 0; q < 8; ++q)
        if (c->peer_map[q] && q != c->rank) (void)hipIpcCloseMemHandle(c->peer_map[q]);
    for (auto &q : c->peer_map) q = nullptr;
    if (c->d_box) (void)hipFree(c->d_box);
    if (c->mbox) (void)hipFree(c->mbox);
    c->d_box = nullptr; c->mbox = nullptr; c->p2p = false;
}
static int p2p_setup(sprs_comm *c) {
    sprs_ctx *ctx = c->ctx;
    if (ctx->p2p_allreduce == 0 || c->world > MB_RANKS) return SPRS_OK;
    struct Card { hipIpcMemHandle_t h; unsigned char ok; unsigned char pad[7]; };
    static_assert(sizeof(Card) == 72, "handle card");
    Card mine; memset(&mine, 0, sizeof(mine));
    bool ok = true;
    if (hipExtMallocWithFlags(&c->mbox, MB_BYTES, hipDeviceMallocUncached) != hipSuccess) { c->mbox = nullptr; ok = hipMalloc(&c->mbox, MB_BYTES) == hipSuccess; }
    ok = ok && hipMemsetAsync(c->mbox, 0, MB_BYTES, ctx->stream) == hipSuccess && hipStreamSynchronize(ctx->stream) == hipSuccess;
    if (ok && c->world > 1) ok = hipIpcGetMemHandle(&mine.h, c->mbox) == hipSuccess;
    (void)hipGetLastError();
    mine.ok = ok ? 1 : 0;
    std::vector<Card> all((size_t)c->world);
    void *d_cards = nullptr;
    SPRS_HIP_TRY(ctx, hipMalloc(&d_cards, sizeof(Card) * ((size_t)c->world + 1)));
    struct Free { void *p; ~Free() { if (p) (void)hipFree(p); } } guard{d_cards};
    Card *d_mine = reinterpret_cast<Card *>(d_cards) + c->world;
    SPRS_HIP_TRY(ctx, hipMemcpyAsync(d_mine, &mine, sizeof(Card), hipMemcpyHostToDevice, ctx->stream));
    SPRS_NCCL_TRY(ctx, rccl().AllGather(d_mine, d_cards, sizeof(Card), ncclInt8, (ncclComm_t)c->nccl, ctx->stream));
    SPRS_HIP_TRY(ctx, hipMemcpyAsync(all.data(), d_cards, sizeof(Card) * (size_t)c->world, hipMemcpyDeviceToHost, ctx->stream));
    SPRS_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    for (int q = 0; q < c->world; ++q) ok = ok && all[(size_t)q].ok;
    if (ok) {
        c->peer_map[c->rank] = c->mbox;
        for (int q = 0; q < c->world && ok; ++q) {
            if (q == c->rank) continue;
            ok = hipIpcOpenMemHandle(&c->peer_map[q], all[(size_t)q].h, hipIpcMemLazyEnablePeerAccess) == hipSuccess;
            if (!ok) c->peer_map[q] = nullptr;
        }
        (void)hipGetLastError();
    }
    // the vote: how many ranks could NOT map everything
    double *d_vote = reinterpret_cast<double *>(d_cards);
    const double mine_bad = ok ? 0.0 : 1.0;
    SPRS_HIP_TRY(ctx, hipMemcpyAsync(d_vote, &mine_bad, sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    SPRS_NCCL_TRY(ctx, rccl().AllReduce(d_vote, d_vote, 1, ncclDouble, ncclSum, (ncclComm_t)c->nccl, ctx->stream));
    double bad = 1.0;
    SPRS_HIP_TRY(ctx, hipMemcpyAsync(&bad, d_vote, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    SPRS_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (bad != 0.0) { p2p_release(c); return SPRS_OK; }
    P2pBox box; memset(&box, 0, sizeof(box));
    for (int q = 0; q < c->world; ++q) box.peer[q] = (unsigned long long)reinterpret_cast<uintptr_t>(c->peer_map[q]);
    box.world = c->world; box.rank = c->rank;
    SPRS_HIP_TRY(ctx, hipMalloc((void **)&c->d_box, sizeof(P2pBox)));
    SPRS_HIP_TRY(ctx, hipMemcpyAsync(c->d_box, &box, sizeof(P2pBox), hipMemcpyHostToDevice, ctx->stream));
    // self-test: one real round trip (every rank has mapped every mailbox by now: the vote above was a barrier).  Its granules carry
    // a tag no solve reaches (2^31 - 1 hand-offs on one slot) in the last slot's second half, so nothing has to be cleaned up.
    {
        int *d_res = reinterpret_cast<int *>(reinterpret_cast<char *>(d_cards) + 16);
        SPRS_HIP_TRY(ctx, hipMemsetAsync(d_res, 0, sizeof(int), ctx->stream));
        const unsigned int off = (unsigned int)mb_offset(MB_SLOTS - 1, 1);
        hipLaunchKernelGGL(mbox_selftest_kernel, dim3(1), dim3(BLOCK), 0, ctx->stream, c->d_box, 0x7fffffffu, off,
                           reinterpret_cast<const unsigned long long *>(reinterpret_cast<const char *>(c->mbox) + off),
                           (unsigned long long)3000 * 100000ull, d_res);
        SPRS_HIP_TRY(ctx, hipGetLastError());
        int res = 0;
        SPRS_HIP_TRY(ctx, hipMemcpyAsync(&res, d_res, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
        SPRS_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        const double failed = res == 1 ? 0.0 : 1.0;
        SPRS_HIP_TRY(ctx, hipMemcpyAsync(d_vote, &failed, sizeof(double), hipMemcpyHostToDevice, ctx->stream));
        SPRS_NCCL_TRY(ctx, rccl().AllReduce(d_vote, d_vote, 1, ncclDouble, ncclSum, (ncclComm_t)c->nccl, ctx->stream));
        double any = 1.0;
        SPRS_HIP_TRY(ctx, hipMemcpyAsync(&any, d_vote, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        SPRS_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        if (any != 0.0) { p2p_release(c); return SPRS_OK; }      // some rank did not see every post: the hand-offs stay on ncclAllReduce
    }
    c->p2p = true;
    return SPRS_OK;
}

int sprs_comm_p2p(const sprs_comm *comm, int *enabled_out) {
    if (!comm || !enabled_out) return SPRS_INVALID_ARGUMENT;
    *enabled_out = comm->p2p ? 1 : 0;
    return SPRS_OK;
}

int sprs_comm_destroy(sprs_comm *comm) {
    if (!comm) return SPRS_OK;
    if (comm->ctx) (void)hipStreamSynchronize(comm->ctx->stream);
    p2p_release(comm);
    if (comm->nccl && rccl().ok) (void)rccl().CommDestroy((ncclComm_t)comm->nccl);
    delete comm;
    return SPRS_OK;
}

int sprs_comm_count(const sprs_comm *comm, int *count_out) {
    if (!comm || !count_out) return SPRS_INVALID_ARGUMENT;
    *count_out = 0;
    if (!rccl().ok || !rccl().CommCount) return SPRS_ERR_RCCL;
    SPRS_NCCL_TRY(comm->ctx, rccl().CommCount((ncclComm_t)comm->nccl, count_out));   // what RCCL itself says, not comm->world
    return SPRS_OK;
}

int sprs_comm_allreduce_sum_f64(sprs_comm *comm, double *dev, size_t count) {
    if (!comm || !dev) return SPRS_INVALID_ARGUMENT;
    SPRS_TRY(allreduce_sum(comm, dev, count, false));
    SPRS_HIP_TRY(comm->ctx, hipStreamSynchronize(comm->ctx->stream));
    return SPRS_OK;
}

// `reps` back-to-back all-reduces of `count` doubles on the context's stream between two events: what ONE hand-off of the
// distributed recurrence costs on this communicator (bench.py's `allreduce_us`)
int sprs_comm_allreduce_timed_f64(sprs_comm *comm, double *dev, size_t count, int reps, double *us_out) {
    if (!comm || !dev || !us_out || reps < 1) return SPRS_INVALID_ARGUMENT;
    sprs_ctx *c = comm->ctx;
    CtxLock lock(c);
    SPRS_HIP_TRY(c, hipSetDevice(c->device));
    hipEvent_t e0 = nullptr, e1 = nullptr;
    struct Ev { hipEvent_t &a, &b; ~Ev() { if (a) (void)hipEventDestroy(a); if (b) (void)hipEventDestroy(b); } } guard{e0, e1};
    SPRS_HIP_TRY(c, hipEventCreate(&e0));
    SPRS_HIP_TRY(c, hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) SPRS_TRY(allreduce_sum(comm, dev, count, false));
    SPRS_HIP_TRY(c, hipEventRecord(e0, c->stream));
    for (int i = 0; i < reps; ++i) SPRS_TRY(allreduce_sum(comm, dev, count, false));
    SPRS_HIP_TRY(c, hipEventRecord(e1, c->stream));
    SPRS_HIP_TRY(c, hipEventSynchronize(e1));
    float ms = 0.f;
    SPRS_HIP_TRY(c, hipEventElapsedTime(&ms, e0, e1));
    *us_out = (double)ms * 1e3 / reps;
    return SPRS_OK;
}

#define SPRS_DIST_API(X, T, CT)                                                                                          \
    int sprs_dist_csr_create_dev_##X(sprs_comm *comm, int64_t n_local, int64_t n_ext, int64_t nnz, const int32_t *rp,    \
                                     const int32_t *ci, const CT *val, int adopt, int n_peers, const int32_t *peer_rank, \
                                     const int64_t *send_off, const int32_t *send_idx_dev, const int64_t *recv_off,      \
                                     sprs_csr **out) {                                                                   \
        try { return dist_csr_create<T>(comm, n_local, n_ext, nnz, rp, ci, (const T *)val, adopt, n_peers, peer_rank,    \
                                        send_off, send_idx_dev, recv_off, out); }                                        \
        catch (...) { return SPRS_ERR_HIP; }                                                                             \
    }                                                                                                                    \
    int sprs_dist_csr_create_allgather_dev_##X(sprs_comm *comm, int64_t n_local, int64_t slice, int64_t nnz,              \
                                               const int32_t *rp, const int32_t *ci, const CT *val, int adopt,            \
                                               sprs_csr **out) {                                                         \
        try { return dist_csr_create_allgather<T>(comm, n_local, slice, nnz, rp, ci, (const T *)val, adopt, out); }      \
        catch (...) { return SPRS_ERR_HIP; }                                                                             \
    }                                                                                                                    \
    /* y_local = A_local * x_ext after exchanging the halo of x_ext (device vector of n_ext elements whose first        \
       n_local entries are this rank's slice of x) */                                                                    \
    int sprs_dist_csr_create_global_dev_##X(sprs_comm *comm, const int64_t *row_starts, int64_t nnz, const int32_t *rp,   \
                                            int32_t *ci_global, const CT *val, int adopt, int exchange, sprs_csr **out) { \
        try { return dist_csr_create_global<T>(comm, row_starts, nnz, rp, ci_global, (const T *)val, adopt, exchange, out); } \
        catch (...) { return SPRS_ERR_HIP; }                                                                             \
    }                                                                                                                    \
    int sprs_dist_mul_vec_dev_##X(const sprs_csr *A, CT *x_ext, CT *y_local) {                                           \
        if (!A || !A->dist || A->dtype != dtype_of<T>::value) return SPRS_INVALID_ARGUMENT;                              \
        return dist_spmv<T>(A, (T *)x_ext, (T *)y_local, 0, nullptr, nullptr, nullptr, nullptr, false, nullptr);                  \
    }
SPRS_DIST_API(d, double, double)
SPRS_DIST_API(z, cplx, sprs_c64)
SPRS_DIST_API(s, float, float)
SPRS_DIST_API(c, cplxf, sprs_c32)

}  // extern "C"
