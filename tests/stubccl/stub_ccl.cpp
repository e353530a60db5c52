// TEST INFRASTRUCTURE ONLY -- a stand-in for librccl.so so that the N > 1 code paths of libtbnn
// (tbnn_comm_create(world >= 2), tbnn_gather_samples, tbnn_set_row_shard) execute on a ONE-GPU box, where RCCL
// itself refuses two ranks on one device.  libtbnn resolves its collective library with dlopen; TBNN_RCCL_LIB points
// it here.  Semantics are RCCL's for the six entry points libtbnn binds (in-order collectives on the caller's
// stream, all-reduce = sum in rank order, all-gather = rank-major), implemented the slow obvious way: every rank
// drains its stream, copies its operand into its slot of a POSIX shared-memory segment, the ranks meet at a
// barrier, every rank reads all slots and uploads the result.  Host-synchronous, so never a performance statement.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <atomic>
#include <cstdio>
#include <cstring>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

namespace {
constexpr size_t SLOT_BYTES = 8u << 20;      // per-rank operand slot (configs[3]'s gradient row is 331 KB)
constexpr int MAX_WORLD = 8;
struct Seg {
    std::atomic<int> joined;
    std::atomic<int> count;
    std::atomic<int> sense;
    char pad[52];
    char slot[MAX_WORLD][SLOT_BYTES];
};
struct StubComm { Seg* seg; int world, rank, local_sense; char name[64]; };

bool spin_until(const std::atomic<int>& v, int want, double seconds) {
    timespec t0; clock_gettime(CLOCK_MONOTONIC, &t0);
    while (v.load(std::memory_order_acquire) != want) {
        timespec t; clock_gettime(CLOCK_MONOTONIC, &t);
        if ((t.tv_sec - t0.tv_sec) + 1e-9 * (t.tv_nsec - t0.tv_nsec) > seconds) return false;
        usleep(50);
    }
    return true;
}
// sense-reversing barrier over the segment; bounded (a rank that died must not hang the test box)
bool barrier(StubComm* c) {
    c->local_sense ^= 1;
    if (c->seg->count.fetch_add(1, std::memory_order_acq_rel) == c->world - 1) {
        c->seg->count.store(0, std::memory_order_relaxed);
        c->seg->sense.store(c->local_sense, std::memory_order_release);
        return true;
    }
    return spin_until(c->seg->sense, c->local_sense, 120.0);
}
size_t dtype_bytes(ncclDataType_t t) { return t == ncclFloat ? 4 : t == ncclDouble ? 8 : 0; }
std::atomic<int> g_ids{0};
std::atomic<int> g_allreduce_calls{0};
}  // namespace

extern "C" {
// test hook: all-reduce calls this process has made (the row-sharded chain must issue exactly one per fused pass)
int stubccl_allreduce_calls() { return g_allreduce_calls.load(); }
const char* ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "ok" : "stub collective library error"; }

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
    memset(id, 0, sizeof(*id));
    snprintf(id->internal, sizeof(id->internal), "/tbnn_stubccl_%d_%d", (int)getpid(), g_ids.fetch_add(1));
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* out, int world, ncclUniqueId id, int rank) {
    if (world < 1 || world > MAX_WORLD || rank < 0 || rank >= world) return ncclInvalidArgument;
    StubComm* c = new StubComm();
    c->world = world; c->rank = rank; c->local_sense = 0;
    strncpy(c->name, id.internal, sizeof(c->name) - 1);
    int fd = -1;
    if (rank == 0) {
        fd = shm_open(c->name, O_CREAT | O_RDWR, 0600);
        if (fd < 0 || ftruncate(fd, sizeof(Seg)) != 0) { delete c; return ncclSystemError; }
    } else {
        for (int tries = 0; tries < 200000 && fd < 0; ++tries) {      // rank 0 creates it; wait up to ~100 s
            fd = shm_open(c->name, O_RDWR, 0600);
            struct stat st;
            if (fd >= 0 && (fstat(fd, &st) != 0 || (size_t)st.st_size < sizeof(Seg))) { close(fd); fd = -1; }
            if (fd < 0) usleep(500);
        }
        if (fd < 0) { delete c; return ncclSystemError; }
    }
    void* p = mmap(nullptr, sizeof(Seg), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) { delete c; return ncclSystemError; }
    c->seg = (Seg*)p;                                                  // a fresh segment is zero-filled
    c->seg->joined.fetch_add(1, std::memory_order_acq_rel);
    if (!spin_until(c->seg->joined, world, 120.0)) { munmap(p, sizeof(Seg)); delete c; return ncclSystemError; }
    if (rank == 0) shm_unlink(c->name);                                // everyone has it mapped: the name can go
    *out = (ncclComm_t)c;
    return ncclSuccess;
}

ncclResult_t ncclCommCount(const ncclComm_t comm, int* count) {
    if (!comm || !count) return ncclInvalidArgument;
    *count = ((StubComm*)comm)->seg->joined.load(std::memory_order_acquire);      // ranks that really joined the segment
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm) {
    StubComm* c = (StubComm*)comm;
    if (c) { munmap(c->seg, sizeof(Seg)); delete c; }
    return ncclSuccess;
}

ncclResult_t ncclAllGather(const void* send, void* recv, size_t count, ncclDataType_t dt, ncclComm_t comm, hipStream_t st) {
    StubComm* c = (StubComm*)comm;
    const size_t b = count * dtype_bytes(dt);
    if (!b || b > SLOT_BYTES) return ncclInvalidArgument;
    if (hipStreamSynchronize(st) != hipSuccess) return ncclUnhandledCudaError;
    if (hipMemcpy(c->seg->slot[c->rank], send, b, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
    if (!barrier(c)) return ncclSystemError;
    for (int r = 0; r < c->world; ++r)
        if (hipMemcpy((char*)recv + (size_t)r * b, c->seg->slot[r], b, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
    if (!barrier(c)) return ncclSystemError;
    return ncclSuccess;
}

ncclResult_t ncclAllReduce(const void* send, void* recv, size_t count, ncclDataType_t dt, ncclRedOp_t op, ncclComm_t comm,
                           hipStream_t st) {
    StubComm* c = (StubComm*)comm;
    const size_t b = count * dtype_bytes(dt);
    if (!b || b > SLOT_BYTES || op != ncclSum) return ncclInvalidArgument;
    g_allreduce_calls.fetch_add(1);
    if (hipStreamSynchronize(st) != hipSuccess) return ncclUnhandledCudaError;
    if (hipMemcpy(c->seg->slot[c->rank], send, b, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
    if (!barrier(c)) return ncclSystemError;
    char* acc = new char[b];
    memcpy(acc, c->seg->slot[0], b);
    for (int r = 1; r < c->world; ++r) {                               // fixed rank order: every rank gets the same bits
        if (dt == ncclFloat) { float* a = (float*)acc; const float* s = (const float*)c->seg->slot[r]; for (size_t i = 0; i < count; ++i) a[i] += s[i]; }
        else { double* a = (double*)acc; const double* s = (const double*)c->seg->slot[r]; for (size_t i = 0; i < count; ++i) a[i] += s[i]; }
    }
    const hipError_t e = hipMemcpy(recv, acc, b, hipMemcpyHostToDevice);
    delete[] acc;
    if (e != hipSuccess) return ncclUnhandledCudaError;
    if (!barrier(c)) return ncclSystemError;
    return ncclSuccess;
}
}  // extern "C"
