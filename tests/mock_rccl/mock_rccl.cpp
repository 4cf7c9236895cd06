// TEST INFRASTRUCTURE ONLY — a stand-in for librccl.so.1 that lets several ranks share ONE GPU.
//
// RCCL refuses two ranks on the same device, and the test boxes have a single GPU; the driver's
// 2/4/8-GPU run is the only place the real library carries traffic between ranks.  To exercise the
// library's distributed code path (exchange plan -> pack -> send/recv -> SpMV -> all-reduce ->
// identical branch decisions) with REAL kernels and REAL separate processes before that run, this
// file implements the nine RCCL entry points libsprsolve_hip.so binds, host-staged through POSIX
// shared memory.  Selected with SPRS_RCCL_LIB=<path>; never used outside tests/.
//
// Semantics kept: in-stream ordering (every call synchronises its stream), grouped send/recv matched
// FIFO per ordered pair of ranks, sum all-reduce in rank order (bit-identical on all ranks).
#include <hip/hip_runtime.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <unistd.h>
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <ctime>
#include <vector>

typedef enum { ncclSuccess = 0, ncclUnhandledCudaError = 1, ncclSystemError = 2, ncclInternalError = 3, ncclInvalidArgument = 4 } ncclResult_t;
typedef enum { ncclInt8 = 0, ncclUint8 = 1, ncclInt32 = 2, ncclUint32 = 3, ncclInt64 = 4, ncclUint64 = 5, ncclFloat16 = 6, ncclFloat32 = 7, ncclFloat64 = 8 } ncclDataType_t;
typedef enum { ncclSum = 0 } ncclRedOp_t;
typedef struct { char internal[128]; } ncclUniqueId;

namespace {
constexpr int MAXR = 8;
constexpr size_t BOX = 16u << 20;       // bytes per ordered (src,dst) mailbox
struct Shared {
    std::atomic<int> ready, arrived, generation;
    std::atomic<uint64_t> sent[MAXR][MAXR], taken[MAXR][MAXR];   // message counters per ordered pair
    std::atomic<uint64_t> box_len[MAXR][MAXR];
    double red[MAXR][64];
};
struct Comm {
    int n, rank;
    Shared *sh;
    char *boxes;
    size_t map_bytes;
    char name[64];
};
struct Op { bool send; void *buf; size_t bytes; int peer; hipStream_t st; };
thread_local bool g_grouped = false;
thread_local std::vector<Op> g_ops;
thread_local Comm *g_comm = nullptr;

size_t esize(ncclDataType_t t) { return t == ncclFloat64 || t == ncclInt64 || t == ncclUint64 ? 8 : (t == ncclFloat32 || t == ncclInt32 || t == ncclUint32 ? 4 : (t == ncclFloat16 ? 2 : 1)); }
char *box(Comm *c, int src, int dst) { return c->boxes + ((size_t)src * MAXR + dst) * BOX; }

void barrier(Comm *c) {
    Shared *s = c->sh;
    const int gen = s->generation.load();
    if (s->arrived.fetch_add(1) + 1 == c->n) { s->arrived.store(0); s->generation.fetch_add(1); }
    else while (s->generation.load() == gen) usleep(20);
}

ncclResult_t run_ops(Comm *c, std::vector<Op> &ops) {
    // sends first (one message in flight per ordered pair: wait until the previous one was taken)
    for (auto &o : ops) if (o.send) {
        if (o.bytes > BOX) return ncclInvalidArgument;
        if (hipStreamSynchronize(o.st) != hipSuccess) return ncclUnhandledCudaError;
        auto &sent = c->sh->sent[c->rank][o.peer]; auto &taken = c->sh->taken[c->rank][o.peer];
        while (sent.load() != taken.load()) usleep(20);
        if (hipMemcpy(box(c, c->rank, o.peer), o.buf, o.bytes, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
        c->sh->box_len[c->rank][o.peer].store(o.bytes);
        sent.fetch_add(1);
    }
    for (auto &o : ops) if (!o.send) {
        auto &sent = c->sh->sent[o.peer][c->rank]; auto &taken = c->sh->taken[o.peer][c->rank];
        while (sent.load() == taken.load()) usleep(20);
        if (c->sh->box_len[o.peer][c->rank].load() != o.bytes) { fprintf(stderr, "mock_rccl: size mismatch %d<-%d\n", c->rank, o.peer); return ncclInvalidArgument; }
        if (hipStreamSynchronize(o.st) != hipSuccess) return ncclUnhandledCudaError;
        if (hipMemcpy(o.buf, box(c, o.peer, c->rank), o.bytes, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
        taken.fetch_add(1);
    }
    return ncclSuccess;
}
}  // namespace

extern "C" {

const char *ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "no error" : "mock_rccl error"; }

ncclResult_t ncclGetUniqueId(ncclUniqueId *id) {
    memset(id, 0, sizeof(*id));
    snprintf(id->internal, sizeof(id->internal), "sprsmock_%d_%ld", (int)getpid(), (long)time(nullptr));
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(void **comm, int nranks, ncclUniqueId id, int rank) {
    if (nranks > MAXR) return ncclInvalidArgument;
    Comm *c = new Comm();
    c->n = nranks; c->rank = rank;
    snprintf(c->name, sizeof(c->name), "/%.48s", id.internal);
    c->map_bytes = sizeof(Shared) + (size_t)MAXR * MAXR * BOX;
    int fd = -1;
    if (rank == 0) {
        fd = shm_open(c->name, O_CREAT | O_RDWR, 0600);
        if (fd < 0 || ftruncate(fd, (off_t)c->map_bytes) != 0) return ncclSystemError;
    } else {
        for (int tries = 0; tries < 30000 && fd < 0; ++tries) { fd = shm_open(c->name, O_RDWR, 0600); if (fd < 0) usleep(1000); }
        if (fd < 0) return ncclSystemError;
        // wait until rank 0 has sized the object
        for (int tries = 0; tries < 30000; ++tries) { off_t len = lseek(fd, 0, SEEK_END); if ((size_t)len >= c->map_bytes) break; usleep(1000); }
    }
    void *p = mmap(nullptr, c->map_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) return ncclSystemError;
    c->sh = (Shared *)p; c->boxes = (char *)p + sizeof(Shared);
    if (rank == 0) { c->sh->ready.store(1); }   // shm_open memory is zero-filled: counters start at 0
    else while (c->sh->ready.load() != 1) usleep(100);
    barrier(c);
    *comm = c;
    return ncclSuccess;
}

ncclResult_t ncclCommCount(const void *comm, int *count) { *count = ((const Comm *)comm)->n; return ncclSuccess; }
ncclResult_t ncclCommDestroy(void *comm) {
    Comm *c = (Comm *)comm;
    if (!c) return ncclSuccess;
    if (c->rank == 0) shm_unlink(c->name);
    munmap(c->sh, c->map_bytes);
    delete c;
    return ncclSuccess;
}

ncclResult_t ncclAllReduce(const void *send, void *recv, size_t count, ncclDataType_t t, ncclRedOp_t, void *comm, hipStream_t st) {
    Comm *c = (Comm *)comm;
    const size_t bytes = count * esize(t);
    if (bytes > sizeof(c->sh->red[0]) || (t != ncclFloat64 && t != ncclFloat32)) return ncclInvalidArgument;
    if (hipStreamSynchronize(st) != hipSuccess) return ncclUnhandledCudaError;
    if (hipMemcpy(c->sh->red[c->rank], send, bytes, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
    barrier(c);
    double outd[64]; float outf[128];
    if (t == ncclFloat64) { for (size_t i = 0; i < count; ++i) { double s = 0; for (int r = 0; r < c->n; ++r) s += c->sh->red[r][i]; outd[i] = s; } }
    else { for (size_t i = 0; i < count; ++i) { float s = 0; for (int r = 0; r < c->n; ++r) s += ((float *)c->sh->red[r])[i]; outf[i] = s; } }
    barrier(c);   // everyone has read the slots before anyone overwrites them in the next call
    if (hipMemcpy(recv, t == ncclFloat64 ? (void *)outd : (void *)outf, bytes, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
    return ncclSuccess;
}

ncclResult_t ncclAllGather(const void *send, void *recv, size_t count, ncclDataType_t t, void *comm, hipStream_t st) {
    Comm *c = (Comm *)comm;
    const size_t bytes = count * esize(t);
    if (bytes > BOX) return ncclInvalidArgument;
    if (hipStreamSynchronize(st) != hipSuccess) return ncclUnhandledCudaError;
    if (hipMemcpy(box(c, c->rank, c->rank), send, bytes, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
    barrier(c);
    for (int r = 0; r < c->n; ++r)
        if (hipMemcpy((char *)recv + (size_t)r * bytes, box(c, r, r), bytes, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
    barrier(c);
    return ncclSuccess;
}

ncclResult_t ncclGroupStart() { g_grouped = true; g_ops.clear(); return ncclSuccess; }
ncclResult_t ncclGroupEnd() {
    g_grouped = false;
    ncclResult_t r = g_comm ? run_ops(g_comm, g_ops) : ncclSuccess;
    g_ops.clear();
    return r;
}
ncclResult_t ncclSend(const void *buf, size_t count, ncclDataType_t t, int peer, void *comm, hipStream_t st) {
    g_comm = (Comm *)comm;
    Op o{true, const_cast<void *>(buf), count * esize(t), peer, st};
    if (g_grouped) { g_ops.push_back(o); return ncclSuccess; }
    std::vector<Op> one{o};
    return run_ops(g_comm, one);
}
ncclResult_t ncclRecv(void *buf, size_t count, ncclDataType_t t, int peer, void *comm, hipStream_t st) {
    g_comm = (Comm *)comm;
    Op o{false, buf, count * esize(t), peer, st};
    if (g_grouped) { g_ops.push_back(o); return ncclSuccess; }
    std::vector<Op> one{o};
    return run_ops(g_comm, one);
}

}  // extern "C"
