// rccl_stub.cpp -- TEST DOUBLE of the RCCL entry points librsrl_hip.so calls (test infrastructure; LD_PRELOADed by tests/test_gpu_rccl_stub.py).
//
// Why: RCCL refuses ranks that share a device, and the build box has ONE GPU, so the N > 1 RCCL path of the shared-W exchange (ncclCommInitAll /
// ncclCommInitRank, one all-reduce of the fixed-point delta table per batch-step, grouped for single-thread hosts) has never executed in any form.
// This double implements the eleven symbols functionally for ranks that live in ONE process -- an all-reduce really sums the ranks' buffers, in rank
// order, through host memory -- and RECORDS every call, so a test can (a) run G = 8 ranks end to end on one device and compare with the unsharded
// run bit for bit, and (b) assert the call protocol a real RCCL needs: for a single-thread group every rank's all-reduce of a batch-step sits
// inside ONE ncclGroupStart / ncclGroupEnd pair, each rank exactly once, in place, ncclInt64 for the dense table.
//
// Semantics kept from NCCL: a collective outside a group blocks until every rank of the communicator has called it (ranks = threads here); inside
// a group the calls are queued and executed at the outermost ncclGroupEnd, which fails (ncclInvalidUsage) unless every rank of the communicator
// has queued exactly one call with equal count / type.  Nothing here is asynchronous: buffers are read after a hipStreamSynchronize of the rank's
// stream and written back before the call returns, so stream capture is not supported (the tests run with RSRL_NO_GRAPH=1).
//   build: g++ -shared -fPIC -O2 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include rccl_stub.cpp -L/opt/rocm/lib -lamdhip64 -o librccl_stub.so   (rsrl_amd/_build.build_rccl_stub)
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <condition_variable>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

namespace {
struct Group;
struct Call { struct ncclComm* comm; const void* send; void* recv; size_t count; ncclDataType_t type; hipStream_t stream; };
struct Group {
    int n = 0;
    std::vector<struct ncclComm*> ranks;
    // rendezvous of un-grouped collectives (one thread per rank)
    std::vector<Call> posted; int arrived = 0; uint64_t generation = 0; ncclResult_t last = ncclSuccess;
};
}  // namespace
struct ncclComm { Group* g; int rank; int device; };

namespace {
std::mutex mu;
std::condition_variable cv;
std::map<std::string, Group*> by_id;
std::string log_text;
uint64_t next_id = 1;
thread_local int depth = 0;
thread_local std::vector<Call> queued;

void logf(const char* fmt, ...) {
    char buf[256];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof(buf), fmt, ap); va_end(ap);
    log_text += buf; log_text += '\n';
}
size_t elem_size(ncclDataType_t t) {
    switch (t) { case ncclInt64: case ncclUint64: case ncclFloat64: return 8; case ncclFloat: case ncclInt32: case ncclUint32: return 4; default: return 0; }
}
// sum over the ranks, ascending, through host memory; `calls` holds exactly one call per rank (any order)
ncclResult_t reduce_all(Group* g, std::vector<Call>& calls) {
    const size_t count = calls[0].count; const ncclDataType_t type = calls[0].type; const size_t es = elem_size(type);
    if (!es || (type != ncclInt64 && type != ncclFloat)) return ncclInvalidArgument;
    std::vector<const Call*> by_rank((size_t)g->n, nullptr);
    for (const Call& c : calls) {
        if (c.comm->g != g || c.count != count || c.type != type || by_rank[(size_t)c.comm->rank]) return ncclInvalidUsage;
        by_rank[(size_t)c.comm->rank] = &c;
    }
    for (const Call* c : by_rank) if (!c) return ncclInvalidUsage;
    std::vector<char> acc(count * es, 0), tmp(count * es);
    for (int r = 0; r < g->n; ++r) {
        const Call* c = by_rank[(size_t)r];
        if (hipSetDevice(c->comm->device) != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess ||
            hipMemcpy(tmp.data(), c->send, count * es, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
        if (type == ncclInt64) { auto* a = reinterpret_cast<int64_t*>(acc.data()); auto* t = reinterpret_cast<const int64_t*>(tmp.data()); for (size_t i = 0; i < count; ++i) a[i] += t[i]; }
        else { auto* a = reinterpret_cast<float*>(acc.data()); auto* t = reinterpret_cast<const float*>(tmp.data()); for (size_t i = 0; i < count; ++i) a[i] += t[i]; }
    }
    for (int r = 0; r < g->n; ++r) {
        const Call* c = by_rank[(size_t)r];
        if (hipSetDevice(c->comm->device) != hipSuccess || hipMemcpy(c->recv, acc.data(), count * es, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
    }
    return ncclSuccess;
}
}  // namespace

extern "C" {
ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
    std::lock_guard<std::mutex> l(mu);
    memset(id, 0, sizeof(*id));
    snprintf(id->internal, sizeof(id->internal), "rccl-stub-%llu", (unsigned long long)next_id++);
    logf("GetUniqueId");
    return ncclSuccess;
}
ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId id, int rank) {
    if (!comm || nranks < 1 || rank < 0 || rank >= nranks) return ncclInvalidArgument;
    std::unique_lock<std::mutex> l(mu);
    const std::string key(id.internal, strnlen(id.internal, sizeof(id.internal)));
    Group*& g = by_id[key];
    if (!g) { g = new Group; g->n = nranks; g->ranks.assign((size_t)nranks, nullptr); }
    if (g->n != nranks || g->ranks[(size_t)rank]) return ncclInvalidUsage;
    int dev = 0; (void)hipGetDevice(&dev);
    *comm = new ncclComm{g, rank, dev};
    g->ranks[(size_t)rank] = *comm;
    logf("CommInitRank n=%d rank=%d dev=%d", nranks, rank, dev);
    // like the real thing: returns once every rank has joined
    cv.notify_all();
    cv.wait(l, [&] { for (auto* c : g->ranks) if (!c) return false; return true; });
    return ncclSuccess;
}
ncclResult_t ncclCommInitAll(ncclComm_t* comms, int ndev, const int* devlist) {
    if (!comms || ndev < 1) return ncclInvalidArgument;
    std::lock_guard<std::mutex> l(mu);
    Group* g = new Group; g->n = ndev; g->ranks.assign((size_t)ndev, nullptr);
    for (int r = 0; r < ndev; ++r) { comms[r] = new ncclComm{g, r, devlist ? devlist[r] : r}; g->ranks[(size_t)r] = comms[r]; }
    logf("CommInitAll n=%d", ndev);
    return ncclSuccess;
}
ncclResult_t ncclCommDestroy(ncclComm_t comm) { std::lock_guard<std::mutex> l(mu); logf("CommDestroy rank=%d", comm ? comm->rank : -1); return ncclSuccess; }
ncclResult_t ncclCommAbort(ncclComm_t comm) { std::lock_guard<std::mutex> l(mu); logf("CommAbort rank=%d", comm ? comm->rank : -1); return ncclSuccess; }
ncclResult_t ncclCommCount(const ncclComm_t comm, int* count) { if (!comm || !count) return ncclInvalidArgument; *count = comm->g->n; return ncclSuccess; }
ncclResult_t ncclCommUserRank(const ncclComm_t comm, int* rank) { if (!comm || !rank) return ncclInvalidArgument; *rank = comm->rank; return ncclSuccess; }
const char* ncclGetErrorString(ncclResult_t r) {
    switch (r) { case ncclSuccess: return "no error"; case ncclInvalidUsage: return "invalid usage (rccl_stub: a group that did not hold one call per rank?)";
                 case ncclInvalidArgument: return "invalid argument"; case ncclUnhandledCudaError: return "unhandled hip error"; default: return "rccl_stub error"; }
}
ncclResult_t ncclGroupStart(void) { if (depth++ == 0) { std::lock_guard<std::mutex> l(mu); logf("GroupStart"); } return ncclSuccess; }
ncclResult_t ncclGroupEnd(void) {
    if (depth <= 0) return ncclInvalidUsage;
    if (--depth > 0) return ncclSuccess;
    std::vector<Call> calls; calls.swap(queued);
    std::lock_guard<std::mutex> l(mu);
    ncclResult_t rc = ncclSuccess;
    // the queued calls, communicator by communicator (one collective per communicator and group here)
    while (!calls.empty() && rc == ncclSuccess) {
        Group* g = calls[0].comm->g;
        std::vector<Call> mine, rest;
        for (const Call& c : calls) (c.comm->g == g ? mine : rest).push_back(c);
        rc = (int)mine.size() == g->n ? reduce_all(g, mine) : ncclInvalidUsage;      // a single thread must queue EVERY rank's call before the group ends
        calls.swap(rest);
    }
    logf("GroupEnd rc=%d", (int)rc);
    return rc;
}
ncclResult_t ncclAllReduce(const void* sendbuff, void* recvbuff, size_t count, ncclDataType_t datatype, ncclRedOp_t op, ncclComm_t comm, hipStream_t stream) {
    if (!comm || !sendbuff || !recvbuff || op != ncclSum) return ncclInvalidArgument;
    const Call call{comm, sendbuff, recvbuff, count, datatype, stream};
    std::unique_lock<std::mutex> l(mu);
    logf("AllReduce rank=%d count=%zu type=%d inplace=%d grouped=%d", comm->rank, count, (int)datatype, sendbuff == recvbuff ? 1 : 0, depth > 0 ? 1 : 0);
    if (depth > 0) { l.unlock(); queued.push_back(call); return ncclSuccess; }
    Group* g = comm->g;
    if (g->n == 1) { std::vector<Call> one{call}; return reduce_all(g, one); }
    // un-grouped: rendezvous of the ranks' threads; the last one to arrive reduces for all
    const uint64_t gen = g->generation;
    g->posted.push_back(call);
    if (++g->arrived == g->n) {
        g->last = reduce_all(g, g->posted);
        g->posted.clear(); g->arrived = 0; g->generation += 1;
        cv.notify_all();
        return g->last;
    }
    cv.wait(l, [&] { return g->generation != gen; });
    return g->last;
}
// ---- the test's window on what was called
int rccl_stub_log(char* out, int n) {
    std::lock_guard<std::mutex> l(mu);
    const int len = (int)log_text.size();
    if (out && n > 0) { const int m = len < n - 1 ? len : n - 1; memcpy(out, log_text.data(), (size_t)m); out[m] = 0; }
    return len;
}
void rccl_stub_log_clear(void) { std::lock_guard<std::mutex> l(mu); log_text.clear(); }
// capability probe (dlsym from rsrl_hip_group_create): this library admits ranks that share a device
int rccl_stub_allows_shared_device(void) { return 1; }
}
