// TEST INFRASTRUCTURE ONLY -- a stand-in for librccl.so that lets ONE GPU run the library's multi-rank and
// multi-device call sequences (mcd_api.hip: ncclCommInitRank / ncclCommInitAll, ncclGroupStart/End, ncclAllReduce on the
// catalogue's streams) with REAL shards and REAL kernels.  RCCL itself refuses two ranks on one device ("invalid usage"),
// so on the single-GPU development box the code that runs with star_begin > 0, per-shard background sums, re-run signals
// crossing ranks etc. would otherwise never execute on a device.  Selected with MCD_RCCL_LIBRARY=<this .so>; never loaded
// by the product otherwise.  It is NOT a performance model: results are staged through host memory (POSIX shared memory
// between processes) and summed in rank order.
//
// Between PROCESSES the all-reduce is asynchronous like the real one (a stand-in that synchronises its stream
// hides missing waits between the library's two streams and reuse of a buffer a collective still reads): the call only
// enqueues, in stream order, a device-to-host copy into pinned memory, a host function (hipLaunchHostFunc) that meets the
// other ranks in shared memory and sums, and the copy back.  The stream stays blocked while a peer is missing -- exactly
// how a real all-reduce behaves when a rank never arrives, which is what the library's collective deadline
// (include/mcd.h) is tested against: FAKE_RCCL_HANG_AT_CALL=k makes this rank's k-th all-reduce sit for
// FAKE_RCCL_HANG_MS (default 30000) before it proceeds.  The single-process clique (ncclGroupStart/End over several
// local devices) keeps the synchronous form.
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <sys/mman.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <mutex>
#include <random>
#include <string>
#include <thread>
#include <vector>

namespace {

constexpr size_t kMaxCount = 1 << 16;          // doubles per all-reduce (B x W outputs of the tests)
constexpr int kMaxRanks = 8;

struct Shared {                                  // one per multi-process communicator, in POSIX shared memory
    std::atomic<int> arrived;
    std::atomic<int> generation;
    std::atomic<int> attached;
    double data[kMaxRanks][kMaxCount];
};

struct LocalGroup {                              // one per ncclCommInitAll clique
    int n = 0;
};

struct Op { ncclComm_t comm; const void* send; void* recv; size_t count; hipStream_t stream; };
thread_local int g_group_depth = 0;
thread_local std::vector<Op> g_ops;

}  // namespace

struct ncclComm {
    int rank = 0, n_ranks = 1, device = 0;
    Shared* shm = nullptr;                       // multi-process
    std::string shm_name;
    LocalGroup* local = nullptr;                 // single-process clique
    double* h_in = nullptr;                      // pinned staging of the asynchronous form (one all-reduce at a time per
    double* h_out = nullptr;                     // communicator, as with the real library)
    long calls = 0;
};

namespace {

bool barrier(Shared* s, int n) {
    const int gen = s->generation.load();
    if (s->arrived.fetch_add(1) + 1 == n) {
        s->arrived.store(0);
        s->generation.fetch_add(1);
        return true;
    }
    const auto t0 = std::chrono::steady_clock::now();
    while (s->generation.load() == gen) {
        if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(120)) return false;
        std::this_thread::yield();
    }
    return true;
}

struct PendingReduce { ncclComm* c; size_t count; long hang_ms; };

// runs on a runtime thread, in stream order, between the two copies: no HIP calls in here
void host_reduce(void* arg) {
    PendingReduce* p = static_cast<PendingReduce*>(arg);
    ncclComm* c = p->c;
    const size_t count = p->count;
    for (long waited = 0; waited < p->hang_ms; waited += 10) std::this_thread::sleep_for(std::chrono::milliseconds(10));
    std::memcpy(c->shm->data[c->rank], c->h_in, count * sizeof(double));
    bool ok = barrier(c->shm, c->n_ranks);
    for (size_t i = 0; i < count; ++i) c->h_out[i] = 0.0;
    for (int r = 0; ok && r < c->n_ranks; ++r)
        for (size_t i = 0; i < count; ++i) c->h_out[i] += c->shm->data[r][i];  // rank order: the same bits on every rank
    ok = ok && barrier(c->shm, c->n_ranks);                                    // nobody overwrites a slot still being read
    if (!ok)
        for (size_t i = 0; i < count; ++i) c->h_out[i] = std::numeric_limits<double>::quiet_NaN();   // a peer never came
    delete p;
}

ncclResult_t reduce_processes(ncclComm_t c, const void* send, void* recv, size_t count, hipStream_t stream) {
    if (count > kMaxCount) return ncclInvalidArgument;
    if (hipSetDevice(c->device) != hipSuccess) return ncclUnhandledCudaError;
    if (!c->h_in) {
        if (hipHostMalloc((void**)&c->h_in, kMaxCount * sizeof(double), hipHostMallocDefault) != hipSuccess ||
            hipHostMalloc((void**)&c->h_out, kMaxCount * sizeof(double), hipHostMallocDefault) != hipSuccess)
            return ncclUnhandledCudaError;
    }
    ++c->calls;
    long hang_ms = 0;
    if (const char* at = std::getenv("FAKE_RCCL_HANG_AT_CALL"))
        if (std::atol(at) == c->calls) {
            const char* ms = std::getenv("FAKE_RCCL_HANG_MS");
            hang_ms = ms ? std::atol(ms) : 30000;
        }
    PendingReduce* p = new PendingReduce{c, count, hang_ms};
    if (hipMemcpyAsync(c->h_in, send, count * sizeof(double), hipMemcpyDeviceToHost, stream) != hipSuccess ||
        hipLaunchHostFunc(stream, host_reduce, p) != hipSuccess ||
        hipMemcpyAsync(recv, c->h_out, count * sizeof(double), hipMemcpyHostToDevice, stream) != hipSuccess)
        return ncclUnhandledCudaError;
    return ncclSuccess;
}

ncclResult_t reduce_local(std::vector<Op>& ops) {
    if (ops.empty()) return ncclSuccess;
    const size_t count = ops[0].count;
    std::vector<double> sum(count, 0.0), tmp(count);
    std::vector<Op*> by_rank(ops.size(), nullptr);
    for (Op& o : ops) {
        if (o.count != count || o.comm->rank < 0 || o.comm->rank >= (int)ops.size()) return ncclInvalidArgument;
        by_rank[o.comm->rank] = &o;
    }
    for (Op* o : by_rank) {
        if (!o) return ncclInvalidUsage;                                      // a clique member did not join the group
        if (hipSetDevice(o->comm->device) != hipSuccess || hipStreamSynchronize(o->stream) != hipSuccess) return ncclUnhandledCudaError;
        if (hipMemcpy(tmp.data(), o->send, count * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
        for (size_t i = 0; i < count; ++i) sum[i] += tmp[i];
    }
    for (Op* o : by_rank)
        if (hipMemcpy(o->recv, sum.data(), count * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
    return ncclSuccess;
}

}  // namespace

extern "C" {

ncclResult_t ncclGetVersion(int* v) { *v = 29999; return ncclSuccess; }         // recognisably not a real RCCL
const char* ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "no error" : "fake rccl: failure (see tests/fake_rccl)"; }

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
    std::memset(id, 0, sizeof *id);
    std::random_device rd;
    std::snprintf(id->internal, sizeof id->internal, "/mcd_fake_rccl_%d_%08x", (int)getpid(), (unsigned)rd());
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* out, int n_ranks, ncclUniqueId id, int rank) {
    if (n_ranks < 1 || n_ranks > kMaxRanks || rank < 0 || rank >= n_ranks) return ncclInvalidArgument;
    ncclComm* c = new ncclComm();
    c->rank = rank; c->n_ranks = n_ranks;
    (void)hipGetDevice(&c->device);
    c->shm_name = id.internal;
    const int fd = shm_open(c->shm_name.c_str(), O_CREAT | O_RDWR, 0600);
    if (fd < 0 || ftruncate(fd, sizeof(Shared)) != 0) { delete c; return ncclSystemError; }
    void* p = mmap(nullptr, sizeof(Shared), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) { delete c; return ncclSystemError; }
    c->shm = static_cast<Shared*>(p);                                         // fresh shared memory is zero-filled
    c->shm->attached.fetch_add(1);
    const auto t0 = std::chrono::steady_clock::now();                          // "implicitly synchronises with other ranks"
    while (c->shm->attached.load() < n_ranks) {
        if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(120)) { delete c; return ncclSystemError; }
        std::this_thread::yield();
    }
    *out = c;
    return ncclSuccess;
}

ncclResult_t ncclCommInitAll(ncclComm_t* comms, int ndev, const int* devlist) {
    if (ndev < 1 || ndev > kMaxRanks) return ncclInvalidArgument;
    LocalGroup* g = new LocalGroup();
    g->n = ndev;
    for (int i = 0; i < ndev; ++i) {
        ncclComm* c = new ncclComm();
        c->rank = i; c->n_ranks = ndev; c->device = devlist ? devlist[i] : i; c->local = g;
        comms[i] = c;
    }
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t c) {
    if (!c) return ncclSuccess;
    if (c->shm) {
        munmap(c->shm, sizeof(Shared));
        if (c->rank == 0) shm_unlink(c->shm_name.c_str());
    }
    if (c->h_in) (void)hipHostFree(c->h_in);
    if (c->h_out) (void)hipHostFree(c->h_out);
    if (c->local && c->rank == 0) delete c->local;
    delete c;
    return ncclSuccess;
}

ncclResult_t ncclCommCount(const ncclComm_t c, int* n) { *n = c->n_ranks; return ncclSuccess; }
ncclResult_t ncclCommUserRank(const ncclComm_t c, int* r) { *r = c->rank; return ncclSuccess; }

ncclResult_t ncclGroupStart() { ++g_group_depth; return ncclSuccess; }
ncclResult_t ncclGroupEnd() {
    if (g_group_depth <= 0) return ncclInvalidUsage;
    if (--g_group_depth > 0) return ncclSuccess;
    std::vector<Op> ops;
    ops.swap(g_ops);
    return reduce_local(ops);
}

ncclResult_t ncclAllReduce(const void* send, void* recv, size_t count, ncclDataType_t type, ncclRedOp_t op, ncclComm_t c,
                           hipStream_t stream) {
    if (!c || type != ncclDouble || op != ncclSum) return ncclInvalidArgument;
    if (c->shm) return reduce_processes(c, send, recv, count, stream);
    if (g_group_depth > 0) { g_ops.push_back(Op{c, send, recv, count, stream}); return ncclSuccess; }
    if (c->n_ranks == 1) {                                                    // a clique of one: nothing to add
        if (send != recv && hipMemcpyAsync(recv, send, count * sizeof(double), hipMemcpyDeviceToDevice, stream) != hipSuccess)
            return ncclUnhandledCudaError;
        return ncclSuccess;
    }
    return ncclInvalidUsage;                                                  // several local devices need a group
}

}  // extern "C"
