// kmg_group.hip -- several GPUs behind the C ABI (include/kmeans_hip.h, "a group of devices").
//
// The reference is single-device (core/src/lib.rs:38-65: one adapter, one queue).  This layer is ImageProcessor::new over a
// device list: one kmg_processor + compute stream + RCCL rank per device, driven SPMD -- every rank runs the same sequence of
// kmg_lloyd_* calls on its row band and meets the others in the path's one exchange step, ncclAllReduce(sum) of the k x 4
// int64 accumulators.  In a one-process group each rank is a worker thread of the library; with one process per
// GPU the calling thread is the rank.  Everything below the collectives is the public device-pointer API of this library:
// the group adds no arithmetic of its own (the loopback exchange used for tests on a one-GPU box is integer add / max).
//
// RCCL is dlopen'ed on first use: a single-GPU host never loads it, and a process that already maps a copy (PyTorch's
// librccl.so.1) keeps exactly that one.

#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>          // types and prototypes only: every call goes through the table below
#include <stddef.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "kmg_internal.h"
#include "kmg_kernels.h"
#include "kmg_table.h"

using namespace kmg;

#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess)                                                                  \
            return fail(e_ == hipErrorOutOfMemory ? KMG_ERR_OUT_OF_MEMORY : KMG_ERR_HIP,       \
                        "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

#define KMG_TRY(expr)                                                                          \
    do {                                                                                       \
        const int rc_ = (expr);                                                                \
        if (rc_ != KMG_OK) return rc_;                                                         \
    } while (0)

namespace {

// ---------------------------------------------------------------------------------------------
// RCCL at run time
// ---------------------------------------------------------------------------------------------
struct Rccl {
    void *handle = nullptr;
    std::string path;
    decltype(&ncclGetVersion) GetVersion = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclCommAbort) CommAbort = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclBroadcast) Broadcast = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
};

std::mutex g_rccl_mu;
Rccl g_rccl;

// The copy the process already maps wins (RTLD_NOLOAD by soname: PyTorch ships its own librccl.so.1 and two copies in one
// process would each open the fabric); then KMG_RCCL_LIBRARY; then the loader's search path and ROCm's install directory.
int rccl_load(const Rccl **out)
{
    std::lock_guard<std::mutex> lock(g_rccl_mu);
    if (!g_rccl.handle) {
        void *h = nullptr;
        std::string from;
        for (const char *name : {"librccl.so.1", "librccl.so"})
            if (!h && (h = dlopen(name, RTLD_NOW | RTLD_NOLOAD))) from = std::string(name) + " (already mapped)";
        if (!h)
            if (const char *env = getenv("KMG_RCCL_LIBRARY"))
                if ((h = dlopen(env, RTLD_NOW | RTLD_LOCAL))) from = env;
        for (const char *name : {"librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"})
            if (!h && (h = dlopen(name, RTLD_NOW | RTLD_LOCAL))) from = name;
        if (!h) return fail(KMG_ERR_UNSUPPORTED, "RCCL is needed for a group of more than one rank and librccl.so.1 could not be loaded: %s", dlerror());
        Rccl r;
        r.handle = h;
        r.path = from;
        bool ok = true;
#define KMG_RCCL_SYM(field, name) ok = ok && (r.field = reinterpret_cast<decltype(r.field)>(dlsym(h, name))) != nullptr
        KMG_RCCL_SYM(GetVersion, "ncclGetVersion");
        KMG_RCCL_SYM(GetUniqueId, "ncclGetUniqueId");
        KMG_RCCL_SYM(CommInitRank, "ncclCommInitRank");
        KMG_RCCL_SYM(CommInitAll, "ncclCommInitAll");
        KMG_RCCL_SYM(CommDestroy, "ncclCommDestroy");
        KMG_RCCL_SYM(CommAbort, "ncclCommAbort");
        KMG_RCCL_SYM(AllReduce, "ncclAllReduce");
        KMG_RCCL_SYM(AllGather, "ncclAllGather");
        KMG_RCCL_SYM(Broadcast, "ncclBroadcast");
        KMG_RCCL_SYM(GroupStart, "ncclGroupStart");
        KMG_RCCL_SYM(GroupEnd, "ncclGroupEnd");
        KMG_RCCL_SYM(GetErrorString, "ncclGetErrorString");
#undef KMG_RCCL_SYM
        if (!ok) return fail(KMG_ERR_UNSUPPORTED, "%s lacks an RCCL entry point this library needs", from.c_str());
        g_rccl = r;
        if (log_debug()) fprintf(stderr, "[kmeans_hip] RCCL from %s\n", from.c_str());
    }
    *out = &g_rccl;
    return KMG_OK;
}

#define NCCL_TRY(rccl_, expr)                                                                                                \
    do {                                                                                                                     \
        const ncclResult_t n_ = (expr);                                                                                      \
        if (n_ != ncclSuccess) return fail(KMG_ERR_HIP, "%s failed: %s (%s:%d)", #expr, (rccl_)->GetErrorString(n_), __FILE__, __LINE__); \
    } while (0)

// ---------------------------------------------------------------------------------------------
// host-side rendezvous of the ranks of one process
// ---------------------------------------------------------------------------------------------
struct Barrier {
    std::mutex mu;
    std::condition_variable cv;
    uint32_t n = 1, waiting = 0;
    uint64_t generation = 0;
    bool broken = false;
    // false: another rank gave up (its error is the call's result)
    bool wait()
    {
        std::unique_lock<std::mutex> lock(mu);
        if (broken) return false;
        const uint64_t gen = generation;
        if (++waiting == n) {
            waiting = 0;
            generation += 1;
            cv.notify_all();
            return true;
        }
        cv.wait(lock, [&] { return generation != gen || broken; });
        return !broken;
    }
    void abort()
    {
        std::lock_guard<std::mutex> lock(mu);
        broken = true;
        cv.notify_all();
    }
    void reset()
    {
        std::lock_guard<std::mutex> lock(mu);
        broken = false;
        waiting = 0;
    }
};

struct Worker {
    std::thread th;
    std::mutex mu;
    std::condition_variable cv;
    std::function<int()> task;
    bool has_task = false, done = false, quit = false;
    int rc = KMG_OK;
    std::string err;
};

}  // namespace

struct GroupRank {
    kmg_group *g = nullptr;
    uint32_t idx = 0, rank = 0;      // local index, rank in the world
    int device = 0;
    kmg_processor *p = nullptr;
    hipStream_t st = nullptr;        // compute stream: kernels and, unless stated otherwise, the collectives
    hipStream_t st_comm = nullptr;   // KMG_GROUP_OVERLAP: the all-reduce of the sums beside the label pass
    hipEvent_t ev_sums = nullptr, ev_done = nullptr, ev_comm = nullptr;
    hipStream_t comm_last = nullptr; // the stream this rank's communicator was last used on (its operations stay ordered)
    ncclComm_t comm = nullptr;       // written once, at creation (worker threads read it without a lock)
    bool comm_aborted = false;       // ncclCommAbort has taken it (kmg_group::abort_mu)
    void *lb_tmp = nullptr;          // loopback: the reduction's result before it replaces the operand
    size_t lb_cap = 0;
    // host-buffer calls: the band, its output, the shrunk working image (grow-only, kept between calls)
    void *d_in = nullptr, *d_out = nullptr, *d_small = nullptr;
    size_t in_cap = 0, out_cap = 0, small_cap = 0;
    Worker *w = nullptr;
};

struct kmg_group {
    kmg_group_options opt;
    uint32_t n_local = 0, first_rank = 0, world = 0;
    bool loopback = false, force = false, collectives = false;
    const Rccl *rccl = nullptr;
    int rccl_version = 0;
    std::vector<GroupRank> ranks;
    Barrier barrier;                 // the local ranks' host rendezvous (loopback exchange, host-buffer calls)
    std::vector<const void *> lb_ptrs;
    std::mutex call_mu;              // one group operation at a time: the ranks' collectives must pair up
    std::mutex abort_mu;             // group_abort: two ranks may fail at once
    std::atomic<bool> broken{false}; // a rank failed while a collective may have been in flight: the communicators are gone
};

namespace {

// ---------------------------------------------------------------------------------------------
// running one function on every local rank
// ---------------------------------------------------------------------------------------------
void worker_main(GroupRank *r)
{
    (void)hipSetDevice(r->device);
    Worker &w = *r->w;
    for (;;) {
        std::function<int()> task;
        {
            std::unique_lock<std::mutex> lock(w.mu);
            w.cv.wait(lock, [&] { return w.has_task || w.quit; });
            if (w.quit) return;
            task.swap(w.task);
            w.has_task = false;
        }
        const int rc = task();                                      // (run_all's task: catches everything)
        {
            std::lock_guard<std::mutex> lock(w.mu);
            w.rc = rc;
            w.err = rc == KMG_OK ? "" : kmg_last_error();
            w.done = true;
        }
        w.cv.notify_all();
    }
}

// A rank that fails while a collective may be in flight leaves the others waiting inside RCCL or at the barrier: release
// them -- the communicators are aborted and the group is finished (every later call returns KMG_ERR_HIP; kmeans_hip.h).
void group_abort(kmg_group *g)
{
    std::lock_guard<std::mutex> lock(g->abort_mu);
    g->broken.store(true);
    g->barrier.abort();
    if (g->rccl)
        for (GroupRank &r : g->ranks)
            if (r.comm && !r.comm_aborted) { (void)g->rccl->CommAbort(r.comm); r.comm_aborted = true; }
}

// what a failing rank does for the others: with a collective possibly in flight the group is aborted; a call without one
// (the host-buffer calls over row bands: upload, shrink, output pass, download) only releases the ranks that wait at THIS
// call's barrier -- run_all resets it at the next call and the group stays usable, like a kmg_processor after a failed call.
void rank_failed(kmg_group *g, bool collective)
{
    if (collective && g->collectives) group_abort(g);
    else g->barrier.abort();
}

int guarded(const std::function<int(GroupRank &)> &fn, GroupRank &r) noexcept
{
    try { return fn(r); } catch (...) { return abi_trap(); }
}

// fn(rank) on every local rank -- on the calling thread for a group of one, else on the ranks' worker threads, side by
// side -- and the first failure (with its message on the CALLING thread) as the result.  `collective`: fn may issue RCCL calls.
int run_all(kmg_group *g, bool collective, const std::function<int(GroupRank &)> &fn)
{
    if (g->broken.load()) return fail(KMG_ERR_HIP, "the group is broken: an earlier call failed on one of its ranks while a collective was in flight");
    g->barrier.reset();
    if (g->n_local == 1) {
        GroupRank &r = g->ranks[0];
        int rc = hipSetDevice(r.device) == hipSuccess ? KMG_OK : fail(KMG_ERR_HIP, "hipSetDevice(%d) failed", r.device);
        if (rc == KMG_OK) rc = guarded(fn, r);
        if (rc != KMG_OK) {
            const std::string msg = kmg_last_error();
            rank_failed(g, collective);
            return fail(rc, "%s", msg.c_str());
        }
        return rc;
    }
    for (GroupRank &r : g->ranks) {
        Worker &w = *r.w;
        std::lock_guard<std::mutex> lock(w.mu);
        GroupRank *rp = &r;
        w.task = [rp, &fn, g, collective]() {
            const int rc = guarded(fn, *rp);
            if (rc != KMG_OK) {
                // keep this rank's message: releasing the others must not replace it
                const std::string msg = kmg_last_error();
                rank_failed(g, collective);
                return fail(rc, "%s", msg.c_str());
            }
            return rc;
        };
        w.done = false;
        w.has_task = true;
        w.cv.notify_all();
    }
    int rc = KMG_OK;
    std::string err;
    for (GroupRank &r : g->ranks) {
        Worker &w = *r.w;
        std::unique_lock<std::mutex> lock(w.mu);
        w.cv.wait(lock, [&] { return w.done; });
        // (the rank that failed FIRST released the others, whose own failures are consequences: prefer a message that does
        // not speak of another rank)
        if (w.rc != KMG_OK && (rc == KMG_OK || err.find("another rank failed") != std::string::npos)) { rc = w.rc; err = w.err; }
    }
    if (rc != KMG_OK) return fail(rc, "%s", err.c_str());
    return KMG_OK;
}

// ---------------------------------------------------------------------------------------------
// the exchange steps
// ---------------------------------------------------------------------------------------------
struct LbPtrs {
    const void *p[KMG_MAX_DEVICES];
    uint32_t n;
};

template <typename T, bool MAX>
__global__ __launch_bounds__(256) void k_lb_reduce(LbPtrs in, T *__restrict__ out, size_t count)
{
    const size_t stride = (size_t)gridDim.x * 256u;
    for (size_t i = (size_t)blockIdx.x * 256u + threadIdx.x; i < count; i += stride) {
        T v = static_cast<const T *>(in.p[0])[i];
        for (uint32_t r = 1; r < in.n; ++r) {
            const T o = static_cast<const T *>(in.p[r])[i];
            v = MAX ? (o > v ? o : v) : (T)(v + o);
        }
        out[i] = v;
    }
}

enum class Op { SumI64, MaxU64, SumU32 };

size_t op_bytes(Op op) { return op == Op::SumU32 ? 4u : 8u; }

// KMG_GROUP_LOOPBACK: the ranks are threads of this process and may share a device.  Every rank publishes its operand,
// all meet, every rank reduces all operands into a buffer of its own, all meet again (nobody still reads an operand) and the
// result replaces the operand.  Host rendezvous and stream synchronisation per exchange: for tests, not for speed.
int lb_rendezvous(GroupRank &r, hipStream_t st)
{
    HIP_TRY(hipStreamSynchronize(st));
    if (!r.g->barrier.wait()) return fail(KMG_ERR_HIP, "another rank failed");
    return KMG_OK;
}

int lb_allreduce(GroupRank &r, void *buf, size_t count, Op op, hipStream_t st)
{
    kmg_group *g = r.g;
    const size_t bytes = count * op_bytes(op);
    if (r.lb_cap < bytes) {
        if (r.lb_tmp) { HIP_TRY(hipStreamSynchronize(st)); HIP_TRY(hipFree(r.lb_tmp)); r.lb_tmp = nullptr; r.lb_cap = 0; }
        HIP_TRY(hipMalloc(&r.lb_tmp, bytes));
        r.lb_cap = bytes;
    }
    g->lb_ptrs[r.idx] = buf;
    KMG_TRY(lb_rendezvous(r, st));
    LbPtrs in;
    in.n = g->n_local;
    for (uint32_t i = 0; i < g->n_local; ++i) in.p[i] = g->lb_ptrs[i];
    const uint32_t grid = (uint32_t)std::min<size_t>((count + 255u) / 256u, 2048u);
    switch (op) {
    case Op::SumI64: hipLaunchKernelGGL((k_lb_reduce<long long, false>), dim3(grid), dim3(256), 0, st, in, (long long *)r.lb_tmp, count); break;
    case Op::MaxU64: hipLaunchKernelGGL((k_lb_reduce<unsigned long long, true>), dim3(grid), dim3(256), 0, st, in, (unsigned long long *)r.lb_tmp, count); break;
    case Op::SumU32: hipLaunchKernelGGL((k_lb_reduce<uint32_t, false>), dim3(grid), dim3(256), 0, st, in, (uint32_t *)r.lb_tmp, count); break;
    }
    HIP_TRY(hipGetLastError());
    KMG_TRY(lb_rendezvous(r, st));
    HIP_TRY(hipMemcpyAsync(buf, r.lb_tmp, bytes, hipMemcpyDeviceToDevice, st));
    return KMG_OK;
}

// a communicator's operations stay ordered on the device when they move to another stream
int comm_on(GroupRank &r, hipStream_t st)
{
    if (r.comm_last && r.comm_last != st) {
        HIP_TRY(hipEventRecord(r.ev_comm, r.comm_last));
        HIP_TRY(hipStreamWaitEvent(st, r.ev_comm, 0));
    }
    r.comm_last = st;
    return KMG_OK;
}

// in place, on `st`
int allreduce(GroupRank &r, void *buf, size_t count, Op op, hipStream_t st)
{
    kmg_group *g = r.g;
    if (!g->collectives) return KMG_OK;
    if (g->loopback) return lb_allreduce(r, buf, count, op, st);
    if (g->broken.load()) return fail(KMG_ERR_HIP, "another rank failed");      // (its communicator may be gone)
    KMG_TRY(comm_on(r, st));
    const ncclDataType_t dt = op == Op::SumI64 ? ncclInt64 : op == Op::MaxU64 ? ncclUint64 : ncclUint32;
    NCCL_TRY(g->rccl, g->rccl->AllReduce(buf, buf, count, dt, op == Op::MaxU64 ? ncclMax : ncclSum, r.comm, st));
    return KMG_OK;
}

// Rank q's share of `buf` is bytes [off(q), off(q + 1)), off(q) = unit * floor(units * q / world): after the call every rank
// holds every share.  Equal shares: one in-place ncclAllGather; otherwise one broadcast per owner.
int allgather_shares(GroupRank &r, void *buf, uint64_t units, size_t unit, hipStream_t st)
{
    kmg_group *g = r.g;
    if (!g->collectives) return KMG_OK;
    const uint32_t world = g->world;
    auto off = [&](uint32_t q) { return (size_t)((units * q) / world) * unit; };
    uint8_t *base = static_cast<uint8_t *>(buf);
    if (g->loopback) {
        g->lb_ptrs[r.idx] = buf;
        KMG_TRY(lb_rendezvous(r, st));
        for (uint32_t q = 0; q < world; ++q)
            if (q != r.rank && off(q + 1) > off(q))
                HIP_TRY(hipMemcpyAsync(base + off(q), static_cast<const uint8_t *>(g->lb_ptrs[q - g->first_rank]) + off(q), off(q + 1) - off(q),
                                       hipMemcpyDeviceToDevice, st));
        KMG_TRY(lb_rendezvous(r, st));
        return KMG_OK;
    }
    if (g->broken.load()) return fail(KMG_ERR_HIP, "another rank failed");
    KMG_TRY(comm_on(r, st));
    if (units % world == 0) {
        const size_t share = off(1);
        NCCL_TRY(g->rccl, g->rccl->AllGather(base + (size_t)r.rank * share, base, share, ncclUint8, r.comm, st));
        return KMG_OK;
    }
    NCCL_TRY(g->rccl, g->rccl->GroupStart());
    for (uint32_t q = 0; q < world; ++q)
        if (off(q + 1) > off(q))
            NCCL_TRY(g->rccl, g->rccl->Broadcast(base + off(q), base + off(q), off(q + 1) - off(q), ncclUint8, (int)q, r.comm, st));
    NCCL_TRY(g->rccl, g->rccl->GroupEnd());
    return KMG_OK;
}

// every rank's `bytes` at `send` -> all ranks' blocks in rank order at `recv` (world x bytes); `send` must not lie inside `recv`
int allgather_blocks(GroupRank &r, const void *send, void *recv, size_t bytes, hipStream_t st)
{
    kmg_group *g = r.g;
    uint8_t *out = static_cast<uint8_t *>(recv);
    if (!g->collectives) return hipMemcpyAsync(out, send, bytes, hipMemcpyDeviceToDevice, st) == hipSuccess ? KMG_OK : fail(KMG_ERR_HIP, "copy failed");
    if (g->loopback) {
        g->lb_ptrs[r.idx] = send;
        KMG_TRY(lb_rendezvous(r, st));
        for (uint32_t q = 0; q < g->world; ++q)
            HIP_TRY(hipMemcpyAsync(out + (size_t)q * bytes, g->lb_ptrs[q - g->first_rank], bytes, hipMemcpyDeviceToDevice, st));
        KMG_TRY(lb_rendezvous(r, st));
        return KMG_OK;
    }
    if (g->broken.load()) return fail(KMG_ERR_HIP, "another rank failed");
    KMG_TRY(comm_on(r, st));
    NCCL_TRY(g->rccl, g->rccl->AllGather(send, recv, bytes, ncclUint8, r.comm, st));
    return KMG_OK;
}

// The sharded initialisation's pick: every rank offers, per image, the largest key of its band and the colour of the pixel that key
// names; all offers are gathered (ONE collective per centroid instead of a MAX all-reduce of the key and a SUM all-reduce of the
// colour), and every rank takes the colour of the largest key -- the reference's tie rule is in the key (image-wide pixel index).
struct InitOffer { unsigned long long key; uint32_t colour, valid; };
__global__ void k_init_offer(const unsigned long long *__restrict__ keys, const uint32_t *__restrict__ colours2, InitOffer *__restrict__ offers, uint32_t n_images)
{
    const uint32_t im = blockIdx.x * blockDim.x + threadIdx.x;
    if (im >= n_images) return;
    InitOffer o;
    o.key = keys[im]; o.colour = colours2[2u * im]; o.valid = colours2[2u * im + 1u];
    offers[im] = o;
}
__global__ void k_init_take(const InitOffer *__restrict__ gathered, uint32_t world, uint32_t n_images, unsigned long long *__restrict__ keys,
                            uint32_t *__restrict__ colours2)
{
    const uint32_t im = blockIdx.x * blockDim.x + threadIdx.x;
    if (im >= n_images) return;
    InitOffer best = gathered[im];
    for (uint32_t q = 1; q < world; ++q) {
        const InitOffer o = gathered[(size_t)q * n_images + im];
        // (a band without the pixel its own key names does not occur; a band without pixels offers key 0, valid 0)
        if (o.valid && (!best.valid || o.key > best.key)) best = o;
    }
    keys[im] = best.key;
    colours2[2u * im] = best.colour;
    colours2[2u * im + 1u] = best.valid ? 1u : 0u;
}

int grow(void **ptr, size_t *cap, size_t bytes, hipStream_t st)
{
    if (*cap >= bytes) return KMG_OK;
    if (*ptr) { HIP_TRY(hipStreamSynchronize(st)); HIP_TRY(hipFree(*ptr)); *ptr = nullptr; *cap = 0; }
    HIP_TRY(hipMalloc(ptr, bytes));
    *cap = bytes;
    return KMG_OK;
}

void band_of(uint32_t height, uint32_t rank, uint32_t world, uint32_t *r0, uint32_t *r1)
{
    *r0 = (uint32_t)(((uint64_t)rank * height) / world);
    *r1 = (uint32_t)(((uint64_t)(rank + 1) * height) / world);
}

}  // namespace

// ---------------------------------------------------------------------------------------------
// kmg_group: creation
// ---------------------------------------------------------------------------------------------
extern "C" void kmg_default_group_options(kmg_group_options *opt)
try {
    if (!opt) return;
    memset(opt, 0, sizeof *opt);
    opt->struct_size = sizeof(kmg_group_options);
    kmg_default_options(&opt->processor);
}
KMG_ABI_CATCH_VOID

extern "C" int kmg_group_unique_id(uint8_t id[KMG_UNIQUE_ID_BYTES])
try {
    static_assert(sizeof(ncclUniqueId) == KMG_UNIQUE_ID_BYTES, "KMG_UNIQUE_ID_BYTES");
    if (!id) return fail(KMG_ERR_INVALID_ARGUMENT, "id is NULL");
    const Rccl *rccl = nullptr;
    KMG_TRY(rccl_load(&rccl));
    ncclUniqueId u;
    NCCL_TRY(rccl, rccl->GetUniqueId(&u));
    memcpy(id, &u, sizeof u);
    return KMG_OK;
}
KMG_ABI_CATCH

static int group_create_impl(const kmg_group_options *opt, const uint8_t *id, uint32_t first_rank, uint32_t world, kmg_group **out)
{
    if (!out) return fail(KMG_ERR_INVALID_ARGUMENT, "out is NULL");
    *out = nullptr;
    kmg_group_options o;
    kmg_default_group_options(&o);
    if (opt) {
        // (kmg_options, the last member, grew by `strategy` in round 6: a caller compiled against the previous header passes the old size)
        constexpr uint32_t kOldSize = (uint32_t)(offsetof(kmg_group_options, processor) + offsetof(kmg_options, strategy));
        if (opt->struct_size != sizeof(kmg_group_options) && opt->struct_size != kOldSize)
            return fail(KMG_ERR_INVALID_ARGUMENT, "kmg_group_options.struct_size mismatch");
        memcpy(&o, opt, opt->struct_size);
        o.struct_size = sizeof(kmg_group_options);
    }
    if (o.n_devices > KMG_MAX_DEVICES) return fail(KMG_ERR_INVALID_ARGUMENT, "more than KMG_MAX_DEVICES = %d devices", KMG_MAX_DEVICES);
    int count = 0;
    const hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0)
        return fail(KMG_ERR_NO_DEVICE, "no HIP device available (%s); libkmeans_hip has no CPU path",
                    e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
    if (o.n_devices == 0) {
        o.n_devices = (uint32_t)std::min(count, KMG_MAX_DEVICES);
        for (uint32_t i = 0; i < o.n_devices; ++i) o.devices[i] = (int32_t)i;
    }
    const bool loopback = (o.flags & KMG_GROUP_LOOPBACK) != 0;
    for (uint32_t i = 0; i < o.n_devices; ++i) {
        if (o.devices[i] < 0 || o.devices[i] >= count) return fail(KMG_ERR_NO_DEVICE, "device %d out of range (%d devices)", o.devices[i], count);
        for (uint32_t j = 0; j < i && !loopback; ++j)
            if (o.devices[j] == o.devices[i])
                return fail(KMG_ERR_INVALID_ARGUMENT, "device %d is listed twice (RCCL takes one rank per device; KMG_GROUP_LOOPBACK allows it)", o.devices[i]);
    }
    if (id == nullptr) { first_rank = 0; world = o.n_devices; }
    if (world == 0 || first_rank + o.n_devices > world) return fail(KMG_ERR_INVALID_ARGUMENT, "ranks %u..%u do not fit a world of %u", first_rank, first_rank + o.n_devices, world);
    if (loopback && world != o.n_devices) return fail(KMG_ERR_INVALID_ARGUMENT, "KMG_GROUP_LOOPBACK is for the ranks of ONE process");

    kmg_group *g = new (std::nothrow) kmg_group();
    if (!g) return fail(KMG_ERR_OUT_OF_MEMORY, "host allocation failed");
    g->opt = o;
    g->n_local = o.n_devices; g->first_rank = first_rank; g->world = world;
    g->loopback = loopback;
    g->force = (o.flags & KMG_GROUP_FORCE_COLLECTIVES) != 0;
    g->collectives = world > 1 || g->force;
    g->barrier.n = g->n_local;
    g->lb_ptrs.assign(g->n_local, nullptr);
    g->ranks.resize(g->n_local);
    struct Undo { kmg_group *g; ~Undo() { if (g) kmg_group_destroy(g); } } undo{g};
    for (uint32_t i = 0; i < g->n_local; ++i) {
        GroupRank &r = g->ranks[i];
        r.g = g; r.idx = i; r.rank = first_rank + i; r.device = o.devices[i];
        kmg_options po = o.processor;
        po.struct_size = sizeof(kmg_options);
        po.device = r.device;
        KMG_TRY(kmg_processor_create_ex(&po, &r.p));
        HIP_TRY(hipSetDevice(r.device));
        HIP_TRY(hipStreamCreateWithFlags(&r.st, hipStreamNonBlocking));
        HIP_TRY(hipStreamCreateWithFlags(&r.st_comm, hipStreamNonBlocking));
        HIP_TRY(hipEventCreateWithFlags(&r.ev_sums, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&r.ev_done, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&r.ev_comm, hipEventDisableTiming));
    }
    if (g->collectives && !g->loopback) {
        KMG_TRY(rccl_load(&g->rccl));
        (void)g->rccl->GetVersion(&g->rccl_version);
        if (id == nullptr) {
            std::vector<ncclComm_t> comms(g->n_local);
            std::vector<int> devs(g->n_local);
            for (uint32_t i = 0; i < g->n_local; ++i) devs[i] = g->ranks[i].device;
            NCCL_TRY(g->rccl, g->rccl->CommInitAll(comms.data(), (int)g->n_local, devs.data()));
            for (uint32_t i = 0; i < g->n_local; ++i) g->ranks[i].comm = comms[i];
        } else {
            ncclUniqueId u;
            memcpy(&u, id, sizeof u);
            NCCL_TRY(g->rccl, g->rccl->GroupStart());
            for (uint32_t i = 0; i < g->n_local; ++i) {
                HIP_TRY(hipSetDevice(g->ranks[i].device));
                NCCL_TRY(g->rccl, g->rccl->CommInitRank(&g->ranks[i].comm, (int)world, u, (int)g->ranks[i].rank));
            }
            NCCL_TRY(g->rccl, g->rccl->GroupEnd());
        }
    }
    if (g->n_local > 1)
        for (GroupRank &r : g->ranks) {
            r.w = new Worker();
            r.w->th = std::thread(worker_main, &r);
        }
    undo.g = nullptr;
    *out = g;
    return KMG_OK;
}

extern "C" int kmg_group_create(const kmg_group_options *opt, kmg_group **out)
try { return group_create_impl(opt, nullptr, 0, 0, out); }
KMG_ABI_CATCH

extern "C" int kmg_group_create_rank(const kmg_group_options *opt, const uint8_t id[KMG_UNIQUE_ID_BYTES], uint32_t first_rank,
                                     uint32_t world, kmg_group **out)
try {
    if (!id) return fail(KMG_ERR_INVALID_ARGUMENT, "id is NULL (kmg_group_unique_id on rank 0, handed to every process)");
    return group_create_impl(opt, id, first_rank, world, out);
}
KMG_ABI_CATCH

extern "C" void kmg_group_destroy(kmg_group *g)
try {
    if (!g) return;
    for (GroupRank &r : g->ranks) {
        if (r.w) {
            { std::lock_guard<std::mutex> lock(r.w->mu); r.w->quit = true; }
            r.w->cv.notify_all();
            if (r.w->th.joinable()) r.w->th.join();
            delete r.w;
        }
        (void)hipSetDevice(r.device);
        if (r.st) (void)hipStreamSynchronize(r.st);
        if (r.st_comm) (void)hipStreamSynchronize(r.st_comm);
        if (r.comm && g->rccl && !r.comm_aborted) (void)g->rccl->CommDestroy(r.comm);
        for (void *ptr : {r.lb_tmp, r.d_in, r.d_out, r.d_small})
            if (ptr) (void)hipFree(ptr);
        if (r.ev_sums) (void)hipEventDestroy(r.ev_sums);
        if (r.ev_done) (void)hipEventDestroy(r.ev_done);
        if (r.ev_comm) (void)hipEventDestroy(r.ev_comm);
        if (r.st) (void)hipStreamDestroy(r.st);
        if (r.st_comm) (void)hipStreamDestroy(r.st_comm);
        if (r.p) kmg_processor_destroy(r.p);
    }
    delete g;
}
KMG_ABI_CATCH_VOID

extern "C" int kmg_group_info(kmg_group *g, uint32_t *n_local, uint32_t *first_rank, uint32_t *world, int *rccl_version)
try {
    if (!g) return fail(KMG_ERR_INVALID_ARGUMENT, "group is NULL");
    if (n_local) *n_local = g->n_local;
    if (first_rank) *first_rank = g->first_rank;
    if (world) *world = g->world;
    if (rccl_version) *rccl_version = g->rccl_version;
    return KMG_OK;
}
KMG_ABI_CATCH

extern "C" kmg_processor *kmg_group_processor(kmg_group *g, uint32_t i) { return g && i < g->n_local ? g->ranks[i].p : nullptr; }
extern "C" void *kmg_group_stream(kmg_group *g, uint32_t i) { return g && i < g->n_local ? (void *)g->ranks[i].st : nullptr; }

// ---------------------------------------------------------------------------------------------
// kmg_group_lloyd: the Lloyd loop over row bands (modules.rs:763-840 + the exchange step), for ONE image or a BATCH of images
// that are each tiled over all ranks (BASELINE config 4 as north_star words it): the accumulators of the whole batch are one
// block of n_images x k x 4 int64 per rank, so an iteration costs ONE all-reduce whatever the number of images.
// ---------------------------------------------------------------------------------------------
namespace {
struct RankLloyd {                       // one image's band on one rank
    kmg_lloyd *s = nullptr;
    int64_t *d_acc = nullptr;            // k x 4 inside the rank's block: this band's sums, then the image's
    uint64_t *d_key = nullptr;           // init: the arg-max key (inside the rank's block, one per image)
    uint32_t *d_colour = nullptr;        // init: {colour of the winning pixel, 1} on its owner
    const uint8_t *band = nullptr;
    uint32_t *labels = nullptr;
    uint32_t row0 = 0, rows = 0, width = 0, height = 0;
    uint64_t n_local = 0, first = 0;
    bool prepared = false, table = false, cells = false;
    void *lab_t = nullptr, *ent_t = nullptr;
};
struct RankBlock {                       // what the images of one rank share: one device block, one collective per exchange
    void *blk = nullptr;
    int64_t *d_acc = nullptr;            // n_images x k x 4
    uint64_t *d_key = nullptr;           // n_images
    uint32_t *d_colour = nullptr;        // n_images x 2
    uint32_t *d_dummy = nullptr;         // one pixel: what a rank without rows binds in a cell-sharded loop
    void *d_offer = nullptr, *d_gathered = nullptr;   // init: n_images offers of this rank / world x n_images gathered
};
}  // namespace

struct kmg_group_lloyd {
    kmg_group *g = nullptr;
    uint32_t k = 0, n_images = 1, flags = 0;
    bool bound = false;
    std::vector<RankLloyd> r;            // [local rank][image]
    std::vector<RankBlock> blocks;       // [local rank]
    std::vector<uint8_t> active;         // [image]: still iterating (kmg_group_lloyd_run_batch); the same on every rank
    RankLloyd &at(uint32_t rank_idx, uint32_t image) { return r[(size_t)rank_idx * n_images + image]; }
};

namespace {

int rank_prepare(kmg_group_lloyd *gl, GroupRank &r, uint32_t image)
{
    RankLloyd &q = gl->at(r.idx, image);
    if (q.prepared) return KMG_OK;
    kmg_group *g = gl->g;
    q.cells = (gl->flags & KMG_GROUP_CELLS) != 0 && gl->k <= 256 && g->collectives;
    q.table = false;
    if (q.cells) {
        // once per image: the band's histogram, all-reduced into the image's; this rank's share of the colour cube
        const uint64_t n_total = (uint64_t)q.width * q.height;
        if (n_total > 0xFFFFFFFFull) return fail(KMG_ERR_UNSUPPORTED, "image has more than 2^32-1 pixels");
        // (a rank without rows still takes its share of the cube: the table of ONE dummy pixel, its count taken out again)
        const uint8_t *dummy = (const uint8_t *)gl->blocks[r.idx].d_dummy;
        if (q.n_local) KMG_TRY(kmg_lloyd_bind_image(q.s, q.band, q.n_local, r.st));
        else KMG_TRY(kmg_lloyd_bind_image(q.s, dummy, 1, r.st));
        void *hist = nullptr;
        uint64_t hist_bytes = 0;
        KMG_TRY(kmg_lloyd_histogram_buffer(q.s, &hist, &hist_bytes));
        if (!q.n_local) HIP_TRY(hipMemsetAsync(hist, 0, hist_bytes, r.st));
        KMG_TRY(allreduce(r, hist, hist_bytes / 4u, Op::SumU32, r.st));
        KMG_TRY(kmg_lloyd_rebuild_from_histogram(q.s, n_total, r.st));
        KMG_TRY(kmg_lloyd_set_cell_share(q.s, r.rank, g->world, r.st));
        void *ent = nullptr;
        uint64_t lab_bytes = 0, ent_bytes = 0;
        KMG_TRY(kmg_lloyd_table_buffers(q.s, &q.lab_t, &lab_bytes, &ent, &ent_bytes));
        q.ent_t = static_cast<uint8_t *>(ent) + sizeof(uint16_t) * (size_t)(kSubCells + kCells);   // the cells' pair entries (u32)
        q.table = true;
    } else if (q.n_local) {
        int strategy = 0;
        KMG_TRY(kmg_lloyd_prepare(q.s, q.band, q.n_local, q.labels != nullptr, &strategy, r.st));
        q.table = strategy == 1;
    }
    q.prepared = true;
    return KMG_OK;
}

// KMG_GROUP_FUSED_UPDATE with KMG_GROUP_CELLS: every rank needs a band with rows and a label map (its label pass carries the update)
bool fused_cells(const kmg_group_lloyd *gl, const RankLloyd &q)
{
    return (gl->flags & KMG_GROUP_FUSED_UPDATE) != 0 && q.cells && q.n_local != 0 && q.labels != nullptr && gl->k <= 256u;
}

bool image_active(const kmg_group_lloyd *gl, uint32_t image) { return gl->active.empty() || gl->active[image] != 0; }

// labels + sums of the current centroids for every (active) image of the batch, and ONE exchange of all their sums
int rank_pass(kmg_group_lloyd *gl, GroupRank &r)
{
    kmg_group *g = gl->g;
    const size_t acc_count = 4u * (size_t)gl->k;
    RankBlock &blk = gl->blocks[r.idx];
    for (uint32_t im = 0; im < gl->n_images; ++im) KMG_TRY(rank_prepare(gl, r, im));
    RankLloyd &q0 = gl->at(r.idx, 0);
    if (q0.cells && fused_cells(gl, q0)) {
        // (KMG_GROUP_FUSED_UPDATE: the sums are added to accumulators that the previous label pass left zero, and this band's
        // label pass updates the centroids from the all-reduced sums -- no hand-over launch, no update launch)
        KMG_TRY(kmg_lloyd_accumulate_into(q0.s, q0.band, q0.n_local, q0.d_acc, r.st));
        KMG_TRY(allreduce(r, q0.d_acc, acc_count, Op::SumI64, r.st));
        KMG_TRY(allgather_shares(r, q0.lab_t, kCells, kCellColours, r.st));
        KMG_TRY(allgather_shares(r, q0.ent_t, kCells, sizeof(uint32_t), r.st));
        return kmg_lloyd_labels_from_tables_update(q0.s, q0.band, q0.n_local, q0.labels, q0.d_acc, r.st);
    }
    if (q0.cells) {
        // (strong scaling of ONE image: kmg_group_lloyd_bind refuses KMG_GROUP_CELLS for a batch)
        const uint8_t *bound = q0.n_local ? q0.band : (const uint8_t *)blk.d_dummy;
        KMG_TRY(kmg_lloyd_assign_accumulate(q0.s, bound, q0.n_local ? q0.n_local : 1, nullptr, q0.d_acc, r.st));
        KMG_TRY(allreduce(r, q0.d_acc, acc_count, Op::SumI64, r.st));
        // every rank's share of the per-colour labels (512 per cell) and of the cells' pair entries -> all ranks, in place
        KMG_TRY(allgather_shares(r, q0.lab_t, kCells, kCellColours, r.st));
        KMG_TRY(allgather_shares(r, q0.ent_t, kCells, sizeof(uint32_t), r.st));
        if (q0.labels && q0.n_local) KMG_TRY(kmg_lloyd_labels_from_tables(q0.s, q0.band, q0.n_local, q0.labels, r.st));
        return KMG_OK;
    }
    // 1. the sums of every image's band (colour table: the cube pass; per-pixel scan: with its label map)
    bool labels_later = false;
    for (uint32_t im = 0; im < gl->n_images; ++im) {
        RankLloyd &q = gl->at(r.idx, im);
        if (!image_active(gl, im)) continue;                   // (its rows stay in the block, zero: the collective's shape is fixed)
        if (q.n_local == 0) { HIP_TRY(hipMemsetAsync(q.d_acc, 0, sizeof(int64_t) * acc_count, r.st)); continue; }
        if (q.table && q.labels) {
            // colour table: the sums come from the cube pass, the label map from a gather pass that feeds nothing in the loop
            KMG_TRY(kmg_lloyd_assign_accumulate(q.s, q.band, q.n_local, nullptr, q.d_acc, r.st));
            labels_later = true;
        } else {
            KMG_TRY(kmg_lloyd_assign_accumulate(q.s, q.band, q.n_local, q.labels, q.d_acc, r.st));
        }
    }
    // 2. ONE all-reduce of the batch's n_images x k x 4 sums (in line, or on the second stream beside the label passes)
    const bool beside = labels_later && (gl->flags & KMG_GROUP_OVERLAP) != 0 && g->collectives && !g->loopback;
    if (beside) {
        HIP_TRY(hipEventRecord(r.ev_sums, r.st));
        HIP_TRY(hipStreamWaitEvent(r.st_comm, r.ev_sums, 0));
        KMG_TRY(allreduce(r, blk.d_acc, acc_count * gl->n_images, Op::SumI64, r.st_comm));
        HIP_TRY(hipEventRecord(r.ev_done, r.st_comm));
    } else {
        KMG_TRY(allreduce(r, blk.d_acc, acc_count * gl->n_images, Op::SumI64, r.st));
    }
    // 3. the label maps of the colour-table images
    if (labels_later)
        for (uint32_t im = 0; im < gl->n_images; ++im) {
            RankLloyd &q = gl->at(r.idx, im);
            if (image_active(gl, im) && q.n_local && q.table && q.labels) KMG_TRY(kmg_lloyd_labels(q.s, q.band, q.n_local, q.labels, r.st));
        }
    if (beside) HIP_TRY(hipStreamWaitEvent(r.st, r.ev_done, 0));         // the compute stream waits for the collective, not the host
    return KMG_OK;
}

bool fused_update(const kmg_group_lloyd *gl) { return (gl->flags & KMG_GROUP_FUSED_UPDATE) != 0 && !gl->g->collectives; }

int rank_fused(kmg_group_lloyd *gl, GroupRank &r)
{
    for (uint32_t im = 0; im < gl->n_images; ++im) {
        RankLloyd &q = gl->at(r.idx, im);
        KMG_TRY(rank_prepare(gl, r, im));
        if (q.n_local) KMG_TRY(kmg_lloyd_assign_update(q.s, q.band, q.n_local, q.labels, q.d_acc, 1, r.st));
    }
    return KMG_OK;
}

int rank_update(kmg_group_lloyd *gl, GroupRank &r)
{
    // modules.rs:773-788, from the image's sums: identical on every rank
    for (uint32_t im = 0; im < gl->n_images; ++im)
        if (image_active(gl, im)) KMG_TRY(kmg_lloyd_update(gl->at(r.idx, im).s, gl->at(r.idx, im).d_acc, r.st));
    return KMG_OK;
}

int rank_prime(kmg_group_lloyd *gl, GroupRank &r)
{
    if (fused_update(gl)) return rank_fused(gl, r);
    if ((gl->flags & KMG_GROUP_FUSED_UPDATE) && (gl->flags & KMG_GROUP_CELLS)) {
        KMG_TRY(rank_prepare(gl, r, 0));
        RankLloyd &q = gl->at(r.idx, 0);
        if (!fused_cells(gl, q))
            return fail(KMG_ERR_INVALID_ARGUMENT, "KMG_GROUP_FUSED_UPDATE with KMG_GROUP_CELLS: every rank needs a band with rows and a label map, k <= 256");
        HIP_TRY(hipMemsetAsync(q.d_acc, 0, sizeof(int64_t) * 4u * gl->k, r.st));     // (from here on every label pass leaves it zero)
        return rank_pass(gl, r);
    }
    return rank_pass(gl, r);                                   // operations.rs:75-83
}

int rank_step(kmg_group_lloyd *gl, GroupRank &r)
{
    if (fused_update(gl)) return rank_fused(gl, r);
    if ((gl->flags & KMG_GROUP_FUSED_UPDATE) && (gl->flags & KMG_GROUP_CELLS)) return rank_pass(gl, r);   // (its label pass updates)
    KMG_TRY(rank_update(gl, r));
    return rank_pass(gl, r);                                   // modules.rs:793-800
}

// ChooseCentroidModule::compute for every image of the batch: an image that has converged at one of its every-check_period
// checks stops being updated (exactly like the single-image loop); its rows stay in the collective and carry nothing.
// iterations[image] = the reference's `current_iteration` when that image's loop stopped.
int rank_run(kmg_group_lloyd *gl, GroupRank &r, uint32_t *iterations)
{
    const kmg_options &o = gl->g->opt.processor;
    const size_t acc_count = 4u * (size_t)gl->k;
    // (`active` is shared by the local ranks: every rank reads the same convergence counts -- all updated from the same sums -- and
    // rank 0 alone writes it, between two meetings of the ranks)
    auto meet = [&]() -> int {
        if (gl->g->n_local > 1 && !gl->g->barrier.wait()) return fail(KMG_ERR_HIP, "another rank failed");
        return KMG_OK;
    };
    KMG_TRY(rank_pass(gl, r));
    std::vector<uint32_t> its(gl->n_images, 0u);
    uint32_t it = 0;
    for (it = 0; it < o.max_iterations; ++it) {                // modules.rs:769
        bool any = false;
        for (uint32_t im = 0; im < gl->n_images; ++im)
            if (image_active(gl, im)) { any = true; its[im] = it; }
        if (!any) break;
        KMG_TRY(rank_update(gl, r));
        KMG_TRY(rank_pass(gl, r));
        if (it > 0 && it % o.check_period == 0) {              // :802
            std::vector<uint8_t> done(gl->n_images, 0);
            for (uint32_t im = 0; im < gl->n_images; ++im) {
                if (!image_active(gl, im)) continue;
                uint32_t conv = 0;
                KMG_TRY(kmg_lloyd_converged_count(gl->at(r.idx, im).s, &conv, r.st));
                if (conv >= gl->k) {
                    done[im] = 1;
                    if (log_debug() && r.rank == 0) fprintf(stderr, "[kmeans_hip] image %u: we have convergence, checked at iteration %u (%u ranks)\n", im, it, gl->g->world);
                }
            }
            if (gl->n_images == 1u) { if (done[0]) break; continue; }
            KMG_TRY(meet());                                   // nobody still reads `active` for this iteration
            for (uint32_t im = 0; im < gl->n_images; ++im)
                if (done[im]) {
                    if (r.idx == 0) gl->active[im] = 0;
                    HIP_TRY(hipMemsetAsync(gl->at(r.idx, im).d_acc, 0, sizeof(int64_t) * acc_count, r.st));
                }
            KMG_TRY(meet());
        }
    }
    HIP_TRY(hipStreamSynchronize(r.st));
    if (iterations)
        for (uint32_t im = 0; im < gl->n_images; ++im)
            iterations[im] = gl->n_images == 1u ? (it < o.max_iterations ? it : o.max_iterations - 1) : its[im];
    return KMG_OK;
}

// PlusPlusInitModule::compute (modules.rs:946-1246) for images sharded in row bands: every rank runs the pass of its band;
// every rank offers the largest 64-bit key of its band (distance bits | image-wide position under the reference's tie rule) with the
// colour of the pixel it names, ONE all-gather brings the offers to every rank, every rank takes the largest and sets the same
// centroid.  A batch does this for all its images at once: n_images offers per collective.
int rank_init(kmg_group_lloyd *gl, GroupRank &r)
{
    RankBlock &blk = gl->blocks[r.idx];
    const uint32_t ni = gl->n_images;
    for (uint32_t im = 0; im < ni; ++im) gl->at(r.idx, im).prepared = false;   // the initialisation starts a new problem (and may bind the band itself)
    // centroid j of every image: the bands' keys are in d_key (or, j = 0, the same `initial` key on every rank)
    auto publish = [&](uint32_t j, bool gather) -> int {
        for (uint32_t im = 0; im < ni; ++im) {
            RankLloyd &q = gl->at(r.idx, im);
            // {colour of the pixel the key names, 1} on the band that owns that pixel, {0, 0} elsewhere
            KMG_TRY(kmg_lloyd_init_pick_band(q.s, q.n_local ? q.band : nullptr, q.n_local, q.first, q.d_key, q.d_colour, r.st));
        }
        if (gather) {
            // (every band names a pixel of its own: the offers differ, the largest key wins)
            const uint32_t grid = (ni + 63u) / 64u;
            hipLaunchKernelGGL(k_init_offer, dim3(grid), dim3(64), 0, r.st, (const unsigned long long *)blk.d_key, blk.d_colour, (InitOffer *)blk.d_offer, ni);
            HIP_TRY(hipGetLastError());
            KMG_TRY(allgather_blocks(r, blk.d_offer, blk.d_gathered, sizeof(InitOffer) * ni, r.st));
            hipLaunchKernelGGL(k_init_take, dim3(grid), dim3(64), 0, r.st, (const InitOffer *)blk.d_gathered, gl->g->collectives ? gl->g->world : 1u, ni,
                               (unsigned long long *)blk.d_key, blk.d_colour);
            HIP_TRY(hipGetLastError());
        } else {
            // (the same key on every rank: exactly one band owns the pixel, the sum of {colour, 1} is its colour)
            KMG_TRY(allreduce(r, blk.d_colour, 2u * ni, Op::SumU32, r.st));
        }
        for (uint32_t im = 0; im < ni; ++im) KMG_TRY(kmg_lloyd_set_centroid_rgba(gl->at(r.idx, im).s, j, gl->at(r.idx, im).d_colour, r.st));
        return KMG_OK;
    };
    std::vector<uint64_t> key0(ni);
    for (uint32_t im = 0; im < ni; ++im) key0[im] = kmg_init_first_key(gl->at(r.idx, im).width, gl->at(r.idx, im).height);   // plus_plus_init.wgsl:161-168 `initial`
    HIP_TRY(hipMemcpyAsync(blk.d_key, key0.data(), sizeof(uint64_t) * ni, hipMemcpyHostToDevice, r.st));
    HIP_TRY(hipStreamSynchronize(r.st));                       // (key0 lives on this stack frame)
    KMG_TRY(publish(0, false));
    for (uint32_t j = 1; j < gl->k; ++j) {
        for (uint32_t im = 0; im < ni; ++im) {
            RankLloyd &q = gl->at(r.idx, im);
            KMG_TRY(kmg_lloyd_init_step(q.s, q.n_local ? q.band : nullptr, q.n_local, q.first, j, q.d_key, r.st));
        }
        KMG_TRY(publish(j, true));                             // ONE all-gather per centroid (rounds 3-5: a MAX and a SUM all-reduce)
    }
    return KMG_OK;
}

void group_lloyd_free(kmg_group_lloyd *gl)
{
    if (!gl) return;
    for (uint32_t i = 0; i < gl->blocks.size(); ++i) {
        (void)hipSetDevice(gl->g->ranks[i].device);
        for (uint32_t im = 0; im < gl->n_images; ++im)
            if (gl->at(i, im).s) kmg_lloyd_destroy(gl->at(i, im).s);
        if (gl->blocks[i].blk) (void)hipFree(gl->blocks[i].blk);
    }
    delete gl;
}

int group_lloyd_new(kmg_group *g, uint32_t k, uint32_t n_images, kmg_group_lloyd **out)
{
    if (n_images == 0u || n_images > 4096u) return fail(KMG_ERR_INVALID_ARGUMENT, "a batch has 1 .. 4096 images");
    kmg_group_lloyd *gl = new (std::nothrow) kmg_group_lloyd();
    if (!gl) return fail(KMG_ERR_OUT_OF_MEMORY, "host allocation failed");
    gl->g = g; gl->k = k; gl->n_images = n_images;
    gl->r.resize((size_t)g->n_local * n_images);
    gl->blocks.resize(g->n_local);
    struct Undo { kmg_group_lloyd *gl; ~Undo() { group_lloyd_free(gl); } } undo{gl};
    for (uint32_t i = 0; i < g->n_local; ++i) {
        RankBlock &b = gl->blocks[i];
        HIP_TRY(hipSetDevice(g->ranks[i].device));
        const size_t acc_bytes = (sizeof(int64_t) * 4u * k * n_images + 255u) & ~(size_t)255u;
        const size_t key_bytes = (sizeof(uint64_t) * n_images + 255u) & ~(size_t)255u, col_bytes = (sizeof(uint32_t) * 2u * n_images + 255u) & ~(size_t)255u;
        const size_t offer_bytes = (sizeof(InitOffer) * n_images + 255u) & ~(size_t)255u;
        HIP_TRY(hipMalloc(&b.blk, acc_bytes + key_bytes + col_bytes + 256 + offer_bytes * (1u + g->world)));
        HIP_TRY(hipMemset(b.blk, 0, acc_bytes + key_bytes + col_bytes + 256 + offer_bytes * (1u + g->world)));
        uint8_t *base = static_cast<uint8_t *>(b.blk);
        b.d_acc = (int64_t *)base;
        b.d_key = (uint64_t *)(base + acc_bytes);
        b.d_colour = (uint32_t *)(base + acc_bytes + key_bytes);
        b.d_dummy = (uint32_t *)(base + acc_bytes + key_bytes + col_bytes);
        b.d_offer = base + acc_bytes + key_bytes + col_bytes + 256;
        b.d_gathered = base + acc_bytes + key_bytes + col_bytes + 256 + offer_bytes;
        for (uint32_t im = 0; im < n_images; ++im) {
            RankLloyd &q = gl->at(i, im);
            KMG_TRY(kmg_lloyd_create(g->ranks[i].p, k, &q.s));
            q.d_acc = b.d_acc + (size_t)im * 4u * k;
            q.d_key = b.d_key + im;
            q.d_colour = b.d_colour + 2u * im;
        }
    }
    undo.gl = nullptr;
    *out = gl;
    return KMG_OK;
}

// d_rgba / row0 / rows / d_labels: [image * n_local + local rank]; widths / heights: [image]
int group_lloyd_bind(kmg_group_lloyd *gl, const uint8_t *const *d_rgba, const uint32_t *row0, const uint32_t *rows, const uint32_t *widths,
                     const uint32_t *heights, uint32_t *const *d_labels, uint32_t flags)
{
    if (!gl || !d_rgba || !row0 || !rows || !widths || !heights) return fail(KMG_ERR_INVALID_ARGUMENT, "bad group_lloyd_bind arguments");
    const uint32_t nl = gl->g->n_local;
    if ((flags & KMG_GROUP_CELLS) && gl->n_images > 1u)
        return fail(KMG_ERR_INVALID_ARGUMENT, "KMG_GROUP_CELLS shards the cube pass of ONE image: not for a batch (its images' cube passes already fill the ranks)");
    for (uint32_t im = 0; im < gl->n_images; ++im) {
        if (!widths[im] || !heights[im]) return fail(KMG_ERR_INVALID_ARGUMENT, "image %u has zero width or height", im);
        if ((uint64_t)widths[im] * heights[im] > 0xFFFFFFFFull) return fail(KMG_ERR_UNSUPPORTED, "image %u has more than 2^32-1 pixels", im);
        for (uint32_t i = 0; i < nl; ++i) {
            const size_t e = (size_t)im * nl + i;
            if (rows[e] && !d_rgba[e]) return fail(KMG_ERR_INVALID_ARGUMENT, "band %u of image %u has rows but no pixels", i, im);
            if ((uint64_t)row0[e] + rows[e] > heights[im]) return fail(KMG_ERR_INVALID_ARGUMENT, "band %u of image %u leaves the image", i, im);
        }
    }
    gl->flags = flags;
    for (uint32_t im = 0; im < gl->n_images; ++im)
        for (uint32_t i = 0; i < nl; ++i) {
            const size_t e = (size_t)im * nl + i;
            RankLloyd &q = gl->at(i, im);
            q.band = d_rgba[e];
            q.labels = d_labels ? d_labels[e] : nullptr;
            q.row0 = row0[e]; q.rows = rows[e];
            q.width = widths[im]; q.height = heights[im];
            q.n_local = (uint64_t)rows[e] * widths[im];
            q.first = (uint64_t)row0[e] * widths[im];
            q.prepared = false;
        }
    gl->active.assign(gl->n_images > 1u ? gl->n_images : 0u, 1);
    gl->bound = true;
    return KMG_OK;
}

}  // namespace

extern "C" int kmg_group_lloyd_create(kmg_group *g, uint32_t k, kmg_group_lloyd **out)
try {
    if (!g || !out) return fail(KMG_ERR_INVALID_ARGUMENT, "bad group_lloyd_create arguments");
    *out = nullptr;
    std::lock_guard<std::mutex> lock(g->call_mu);
    return group_lloyd_new(g, k, 1u, out);
}
KMG_ABI_CATCH

extern "C" int kmg_group_lloyd_create_batch(kmg_group *g, uint32_t k, uint32_t n_images, kmg_group_lloyd **out)
try {
    if (!g || !out) return fail(KMG_ERR_INVALID_ARGUMENT, "bad group_lloyd_create_batch arguments");
    *out = nullptr;
    std::lock_guard<std::mutex> lock(g->call_mu);
    return group_lloyd_new(g, k, n_images, out);
}
KMG_ABI_CATCH

extern "C" void kmg_group_lloyd_destroy(kmg_group_lloyd *gl)
try {
    if (!gl) return;
    std::lock_guard<std::mutex> lock(gl->g->call_mu);
    group_lloyd_free(gl);
}
KMG_ABI_CATCH_VOID

extern "C" int kmg_group_lloyd_bind(kmg_group_lloyd *gl, const uint8_t *const *d_rgba, const uint32_t *row0, const uint32_t *rows,
                                    uint32_t width, uint32_t height, uint32_t *const *d_labels, uint32_t flags)
try {
    if (!gl) return fail(KMG_ERR_INVALID_ARGUMENT, "group_lloyd is NULL");
    if (gl->n_images != 1u) return fail(KMG_ERR_INVALID_ARGUMENT, "a batch is bound with kmg_group_lloyd_bind_batch");
    std::lock_guard<std::mutex> lock(gl->g->call_mu);
    return group_lloyd_bind(gl, d_rgba, row0, rows, &width, &height, d_labels, flags);
}
KMG_ABI_CATCH

extern "C" int kmg_group_lloyd_bind_batch(kmg_group_lloyd *gl, const uint8_t *const *d_rgba, const uint32_t *row0, const uint32_t *rows,
                                          const uint32_t *widths, const uint32_t *heights, uint32_t *const *d_labels, uint32_t flags)
try {
    if (!gl) return fail(KMG_ERR_INVALID_ARGUMENT, "group_lloyd is NULL");
    std::lock_guard<std::mutex> lock(gl->g->call_mu);
    return group_lloyd_bind(gl, d_rgba, row0, rows, widths, heights, d_labels, flags);
}
KMG_ABI_CATCH

static int group_lloyd_set_centroids(kmg_group_lloyd *gl, uint32_t image, const float *centroids4)
{
    if (!gl || !centroids4 || image >= gl->n_images) return fail(KMG_ERR_INVALID_ARGUMENT, "bad group_lloyd_set_centroids arguments");
    std::lock_guard<std::mutex> lock(gl->g->call_mu);
    for (uint32_t i = 0; i < gl->g->n_local; ++i) {
        HIP_TRY(hipSetDevice(gl->g->ranks[i].device));
        KMG_TRY(kmg_lloyd_set_centroids(gl->at(i, image).s, centroids4, gl->g->ranks[i].st));
    }
    return KMG_OK;
}

static int group_lloyd_get_centroids(kmg_group_lloyd *gl, uint32_t image, float *centroids4)
{
    if (!gl || !centroids4 || image >= gl->n_images) return fail(KMG_ERR_INVALID_ARGUMENT, "bad group_lloyd_get_centroids arguments");
    std::lock_guard<std::mutex> lock(gl->g->call_mu);
    HIP_TRY(hipSetDevice(gl->g->ranks[0].device));
    return kmg_lloyd_get_centroids(gl->at(0, image).s, centroids4, gl->g->ranks[0].st);      // (identical on every rank)
}

extern "C" int kmg_group_lloyd_set_centroids(kmg_group_lloyd *gl, const float *centroids4)
try { return group_lloyd_set_centroids(gl, 0u, centroids4); }
KMG_ABI_CATCH

extern "C" int kmg_group_lloyd_get_centroids(kmg_group_lloyd *gl, float *centroids4)
try { return group_lloyd_get_centroids(gl, 0u, centroids4); }
KMG_ABI_CATCH

extern "C" int kmg_group_lloyd_set_centroids_image(kmg_group_lloyd *gl, uint32_t image, const float *centroids4)
try { return group_lloyd_set_centroids(gl, image, centroids4); }
KMG_ABI_CATCH

extern "C" int kmg_group_lloyd_get_centroids_image(kmg_group_lloyd *gl, uint32_t image, float *centroids4)
try { return group_lloyd_get_centroids(gl, image, centroids4); }
KMG_ABI_CATCH

#define KMG_GROUP_CALL(name, body)                                                                     \
    extern "C" int name(kmg_group_lloyd *gl)                                                           \
    try {                                                                                              \
        if (!gl) return fail(KMG_ERR_INVALID_ARGUMENT, #name ": group_lloyd is NULL");                 \
        if (!gl->bound) return fail(KMG_ERR_INVALID_ARGUMENT, #name ": no bands (kmg_group_lloyd_bind)"); \
        std::lock_guard<std::mutex> lock(gl->g->call_mu);                                              \
        return run_all(gl->g, true, [gl](GroupRank &r) -> int { (void)gl; body; });                    \
    }                                                                                                  \
    KMG_ABI_CATCH

KMG_GROUP_CALL(kmg_group_lloyd_init, return rank_init(gl, r))
KMG_GROUP_CALL(kmg_group_lloyd_prime, return rank_prime(gl, r))
KMG_GROUP_CALL(kmg_group_lloyd_step, return rank_step(gl, r))
KMG_GROUP_CALL(kmg_group_lloyd_sync, HIP_TRY(hipStreamSynchronize(r.st)); return KMG_OK)

static int group_lloyd_run(kmg_group_lloyd *gl, uint32_t *iterations)
{
    if (!gl) return fail(KMG_ERR_INVALID_ARGUMENT, "kmg_group_lloyd_run: group_lloyd is NULL");
    if (!gl->bound) return fail(KMG_ERR_INVALID_ARGUMENT, "kmg_group_lloyd_run: no bands (kmg_group_lloyd_bind)");
    if (gl->flags & KMG_GROUP_FUSED_UPDATE)
        return fail(KMG_ERR_INVALID_ARGUMENT, "kmg_group_lloyd_run: KMG_GROUP_FUSED_UPDATE is for _prime / _step (the loop reads the convergence count between update and re-assignment)");
    std::lock_guard<std::mutex> lock(gl->g->call_mu);
    if (gl->n_images > 1u) gl->active.assign(gl->n_images, 1);
    std::vector<uint32_t> its((size_t)gl->g->n_local * gl->n_images, 0u);
    const int rc = run_all(gl->g, true, [&](GroupRank &r) { return rank_run(gl, r, &its[(size_t)r.idx * gl->n_images]); });
    if (gl->n_images > 1u) gl->active.assign(gl->n_images, 1);           // (_prime / _step after the loop act on every image again)
    KMG_TRY(rc);
    if (iterations)
        for (uint32_t im = 0; im < gl->n_images; ++im) iterations[im] = its[im];
    return KMG_OK;
}

extern "C" int kmg_group_lloyd_run(kmg_group_lloyd *gl, uint32_t *iterations)
try {
    if (gl && gl->n_images != 1u) return fail(KMG_ERR_INVALID_ARGUMENT, "a batch runs with kmg_group_lloyd_run_batch (one iteration count per image)");
    return group_lloyd_run(gl, iterations);
}
KMG_ABI_CATCH

extern "C" int kmg_group_lloyd_run_batch(kmg_group_lloyd *gl, uint32_t *iterations)
try { return group_lloyd_run(gl, iterations); }
KMG_ABI_CATCH

extern "C" kmg_lloyd *kmg_group_lloyd_member(kmg_group_lloyd *gl, uint32_t i, int *strategy)
try {
    if (!gl || i >= gl->blocks.size()) return nullptr;
    if (strategy) *strategy = gl->at(i, 0).table ? 1 : 0;
    return gl->at(i, 0).s;
}
KMG_ABI_CATCH_NULL

// ---------------------------------------------------------------------------------------------
// host-buffer API over a one-process group (ImageProcessor::{palette, find, reduce}, lib.rs:67-164)
// ---------------------------------------------------------------------------------------------
namespace {

// what the ranks of one call share
struct HostCall {
    const uint8_t *rgba = nullptr;
    uint32_t w = 0, h = 0;
    uint32_t k = 0;                      // centroids of the output pass (octree: <= color_count)
    int algo = 0, mode = 0;
    bool want_palette = false;           // k-means / octree palette extraction (palette, reduce) or a given palette (find)
    bool want_output = false;
    uint8_t *out = nullptr;
    uint32_t sw = 0, sh = 0;             // the shrunk working image (== w, h when nothing shrinks)
    bool shrink = false, sharded_kmeans = false;
    std::vector<uint8_t> shrunk;         // host copy of the working image when it was shrunk on the devices
    std::vector<float> c4;
    std::vector<std::array<uint8_t, 4>> colors;   // octree result
    kmg_group_lloyd *gl = nullptr;       // sharded full-resolution k-means
};

// k-means of a small device-resident image on one device (operations.rs:15-88 without the shrink)
int kmeans_small(kmg_processor *p, const uint8_t *d_img, uint32_t w, uint32_t h, uint32_t k, hipStream_t st, float *c4)
{
    kmg_lloyd *s = nullptr;
    KMG_TRY(kmg_lloyd_create(p, k, &s));
    int rc = kmg_lloyd_init_centroids(s, d_img, w, h, st);                      // operations.rs:73
    uint32_t it = 0;
    if (rc == KMG_OK) rc = kmg_lloyd_run(s, d_img, (uint64_t)w * h, nullptr, &it, st);   // operations.rs:85
    if (rc == KMG_OK) rc = kmg_lloyd_get_centroids(s, c4, st);
    kmg_lloyd_destroy(s);
    return rc;
}

int host_call_rank(kmg_group *g, HostCall &c, GroupRank &r)
{
    uint32_t r0, r1;
    band_of(c.h, r.rank, g->world, &r0, &r1);
    const uint32_t rows = r1 - r0;
    const size_t row_bytes = (size_t)c.w * 4u;
    const uint32_t halo = (c.shrink && rows && r1 < c.h) ? 1u : 0u;  // the bilinear shrink samples the row below a band's last
    auto meet = [&]() -> int {
        if (g->n_local > 1 && !g->barrier.wait()) return fail(KMG_ERR_HIP, "another rank failed");
        return KMG_OK;
    };
    if (rows) {
        KMG_TRY(grow(&r.d_in, &r.in_cap, (size_t)(rows + 1u) * row_bytes, r.st));
        HIP_TRY(copy_host_image(r.p, r.d_in, c.rgba + (size_t)r0 * row_bytes, (size_t)(rows + halo) * row_bytes, hipMemcpyHostToDevice, r.st));   // structures.rs:31-65
    }
    if (c.want_palette) {
        if (c.shrink) {
            // structures.rs:67-182: every rank shrinks the output rows whose first source row lies in its band
            uint32_t o0 = c.sh, o1 = 0;
            for (uint32_t gy = 0; gy < c.sh; ++gy) {
                const uint32_t y0 = resize_source_row(gy, c.h, c.sh);
                if (y0 >= r0 && y0 < r1) { o0 = std::min(o0, gy); o1 = std::max(o1, gy + 1u); }
            }
            if (o1 > o0) {
                const size_t bytes = (size_t)(o1 - o0) * c.sw * 4u;
                KMG_TRY(grow(&r.d_small, &r.small_cap, (size_t)c.sw * c.sh * 4u, r.st));
                HIP_TRY(launch_resize_band((const uint32_t *)r.d_in, c.w, c.h, r0, c.sw, c.sh, o0, o1 - o0, (uint32_t *)r.d_small, r.st));
                HIP_TRY(hipMemcpyAsync(c.shrunk.data() + (size_t)o0 * c.sw * 4u, r.d_small, bytes, hipMemcpyDeviceToHost, r.st));
                HIP_TRY(hipStreamSynchronize(r.st));
            }
            KMG_TRY(meet());
        }
        if (c.sharded_kmeans) {
            // full-resolution k-means over the bands: initialisation, loop, one all-reduce per iteration
            KMG_TRY(rank_init(c.gl, r));
            uint32_t it = 0;
            KMG_TRY(rank_run(c.gl, r, &it));
            if (r.idx == 0) KMG_TRY(kmg_lloyd_get_centroids(c.gl->at(0, 0).s, c.c4.data(), r.st));
        } else if (r.idx == 0) {
            // the working image is tiny (<= 256 x 256, or the whole small image): one device, launch-bound
            const uint8_t *host_img = c.shrink ? c.shrunk.data() : c.rgba;
            if (c.algo == KMG_ALGO_OCTREE) {
                c.colors = octree_sorted_palette(host_img, (uint64_t)c.sw * c.sh, c.k);            // lib.rs:288-331
                if (c.colors.empty()) return fail(KMG_ERR_INVALID_ARGUMENT, "the octree returned no colour");
                if (c.colors.size() > KMG_MAX_K) return fail(KMG_ERR_UNSUPPORTED, "the octree returned %zu colours, more than KMG_MAX_K = %u", c.colors.size(), KMG_MAX_K);
                c.k = (uint32_t)c.colors.size();
                c.c4.resize(4u * (size_t)c.k);
                KMG_TRY(kmg_palette_to_centroids(c.colors[0].data(), c.k, c.c4.data()));
            } else {
                const uint8_t *d_img = (const uint8_t *)r.d_in;          // a group of one holds the whole image already
                if (c.shrink || g->n_local > 1) {
                    const size_t bytes = (size_t)c.sw * c.sh * 4u;
                    KMG_TRY(grow(&r.d_small, &r.small_cap, bytes, r.st));
                    HIP_TRY(hipMemcpyAsync(r.d_small, host_img, bytes, hipMemcpyHostToDevice, r.st));
                    d_img = (const uint8_t *)r.d_small;
                }
                KMG_TRY(kmeans_small(r.p, d_img, c.sw, c.sh, c.k, r.st, c.c4.data()));
            }
        }
        KMG_TRY(meet());
    }
    if (c.want_output && rows) {
        // find_colors / dither_colors / meld_colors at full resolution (lib.rs:139-161): the Bayer index uses image rows
        KMG_TRY(grow(&r.d_out, &r.out_cap, (size_t)rows * row_bytes, r.st));
        KMG_TRY(kmg_dev_apply(r.p, (const uint8_t *)r.d_in, c.w, rows, r0, c.c4.data(), c.k, c.mode, (uint8_t *)r.d_out, r.st));
        HIP_TRY(copy_host_image(r.p, c.out + (size_t)r0 * row_bytes, r.d_out, (size_t)rows * row_bytes, hipMemcpyDeviceToHost, r.st));   // structures.rs:441-470
        HIP_TRY(hipStreamSynchronize(r.st));
    }
    return KMG_OK;
}

int host_call(kmg_group *g, HostCall &c)
{
    if (!g) return fail(KMG_ERR_INVALID_ARGUMENT, "group is NULL");
    if (!c.rgba) return fail(KMG_ERR_INVALID_ARGUMENT, "image pointer is NULL");
    if (c.w == 0 || c.h == 0) return fail(KMG_ERR_INVALID_ARGUMENT, "image has zero width or height");
    if ((uint64_t)c.w * c.h > 0xFFFFFFFFull) return fail(KMG_ERR_UNSUPPORTED, "image has more than 2^32-1 pixels");
    if (g->world != g->n_local) return fail(KMG_ERR_UNSUPPORTED, "the host-buffer calls of a group need all its ranks in one process");
    std::lock_guard<std::mutex> lock(g->call_mu);
    const kmg_options &o = g->opt.processor;
    c.sw = c.w; c.sh = c.h;
    if (c.want_palette) {
        const uint32_t m = c.algo == KMG_ALGO_OCTREE ? 128u : o.shrink_max_dim;       // lib.rs:293; structures.rs:23
        if (m && (c.w > m || c.h > m)) {
            c.shrink = true;
            kmg_resized_dims(c.w, c.h, m, &c.sw, &c.sh);
            c.shrunk.resize((size_t)c.sw * c.sh * 4u);
        }
        c.c4.resize(4u * (size_t)c.k);
        c.sharded_kmeans = c.algo == KMG_ALGO_KMEANS && !c.shrink && (uint64_t)c.w * c.h >= (1ull << 20) && g->n_local > 1;
    }
    struct Guard { kmg_group_lloyd *gl; ~Guard() { group_lloyd_free(gl); } } guard{nullptr};
    std::vector<void *> bands(g->n_local);
    if (c.sharded_kmeans) {
        // the bands' device buffers must exist before they are bound: size them here, on the calling thread
        std::vector<uint32_t> row0(g->n_local), rows(g->n_local);
        for (uint32_t i = 0; i < g->n_local; ++i) {
            GroupRank &r = g->ranks[i];
            uint32_t a, b;
            band_of(c.h, r.rank, g->world, &a, &b);
            row0[i] = a; rows[i] = b - a;
            HIP_TRY(hipSetDevice(r.device));
            if (b > a) KMG_TRY(grow(&r.d_in, &r.in_cap, (size_t)(b - a + 1u) * c.w * 4u, r.st));
            bands[i] = r.d_in;
        }
        KMG_TRY(group_lloyd_new(g, c.k, 1u, &c.gl));
        guard.gl = c.gl;
        KMG_TRY(group_lloyd_bind(c.gl, (const uint8_t *const *)bands.data(), row0.data(), rows.data(), &c.w, &c.h, nullptr, 0u));
    }
    // (only the sharded full-resolution k-means issues collectives: everything else is per-band work with host rendezvous)
    return run_all(g, c.sharded_kmeans, [&](GroupRank &r) { return host_call_rank(g, c, r); });
}

}  // namespace

extern "C" int kmg_group_palette(kmg_group *g, const uint8_t *rgba, uint32_t width, uint32_t height, uint32_t color_count, int algo,
                                 uint8_t *out_rgba, uint32_t *out_count)
try {
    if (color_count == 0) return fail(KMG_ERR_INVALID_ARGUMENT, "k must be an integer higher than 0");
    if (!out_rgba || !out_count) return fail(KMG_ERR_INVALID_ARGUMENT, "output pointer is NULL");
    if (algo != KMG_ALGO_KMEANS && algo != KMG_ALGO_OCTREE) return fail(KMG_ERR_INVALID_ARGUMENT, "unknown algorithm %d", algo);
    if (algo == KMG_ALGO_KMEANS && color_count > KMG_MAX_K) return fail(KMG_ERR_UNSUPPORTED, "k = %u exceeds KMG_MAX_K = %u", color_count, KMG_MAX_K);
    HostCall c;
    c.rgba = rgba; c.w = width; c.h = height; c.k = color_count; c.algo = algo; c.want_palette = true;
    KMG_TRY(host_call(g, c));
    if (algo == KMG_ALGO_OCTREE) {
        for (size_t i = 0; i < c.colors.size(); ++i) memcpy(out_rgba + 4 * i, c.colors[i].data(), 4);
        *out_count = (uint32_t)c.colors.size();
        return KMG_OK;
    }
    sorted_palette_of(c.c4.data(), color_count, out_rgba);     // lib.rs:255-286
    *out_count = color_count;
    return KMG_OK;
}
KMG_ABI_CATCH

extern "C" int kmg_group_find(kmg_group *g, const uint8_t *rgba, uint32_t width, uint32_t height, const uint8_t *palette_rgba,
                              uint32_t n_colors, int mode, uint8_t *out_rgba)
try {
    if (!palette_rgba || n_colors == 0) return fail(KMG_ERR_INVALID_ARGUMENT, "palette is empty");
    if (!out_rgba) return fail(KMG_ERR_INVALID_ARGUMENT, "output pointer is NULL");
    if (mode < KMG_MODE_REPLACE || mode > KMG_MODE_MELD) return fail(KMG_ERR_INVALID_ARGUMENT, "unknown mode %d", mode);
    if (n_colors > KMG_MAX_K) return fail(KMG_ERR_UNSUPPORTED, "k = %u exceeds KMG_MAX_K = %u", n_colors, KMG_MAX_K);
    HostCall c;
    c.rgba = rgba; c.w = width; c.h = height; c.k = n_colors; c.mode = mode; c.want_output = true; c.out = out_rgba;
    c.c4.resize(4u * (size_t)n_colors);
    KMG_TRY(kmg_palette_to_centroids(palette_rgba, n_colors, c.c4.data()));     // lib.rs:86-87
    return host_call(g, c);
}
KMG_ABI_CATCH

extern "C" int kmg_group_reduce(kmg_group *g, const uint8_t *rgba, uint32_t width, uint32_t height, uint32_t color_count, int algo,
                                int mode, uint8_t *out_rgba)
try {
    if (color_count == 0) return fail(KMG_ERR_INVALID_ARGUMENT, "k must be an integer higher than 0");
    if (!out_rgba) return fail(KMG_ERR_INVALID_ARGUMENT, "output pointer is NULL");
    if (algo != KMG_ALGO_KMEANS && algo != KMG_ALGO_OCTREE) return fail(KMG_ERR_INVALID_ARGUMENT, "unknown algorithm %d", algo);
    if (mode < KMG_MODE_REPLACE || mode > KMG_MODE_MELD) return fail(KMG_ERR_INVALID_ARGUMENT, "unknown mode %d", mode);
    if (algo == KMG_ALGO_KMEANS && color_count > KMG_MAX_K) return fail(KMG_ERR_UNSUPPORTED, "k = %u exceeds KMG_MAX_K = %u", color_count, KMG_MAX_K);
    HostCall c;
    c.rgba = rgba; c.w = width; c.h = height; c.k = color_count; c.algo = algo; c.mode = mode;
    c.want_palette = true; c.want_output = true; c.out = out_rgba;
    return host_call(g, c);
}
KMG_ABI_CATCH

extern "C" int kmg_group_reduce_batch(kmg_group *g, uint32_t n_images, const uint8_t *const *rgba, const uint32_t *widths,
                                      const uint32_t *heights, uint32_t color_count, int algo, int mode, uint8_t *const *out_rgba)
try {
    if (!g || !rgba || !widths || !heights || !out_rgba) return fail(KMG_ERR_INVALID_ARGUMENT, "bad group_reduce_batch arguments");
    if (g->world != g->n_local) return fail(KMG_ERR_UNSUPPORTED, "the host-buffer calls of a group need all its ranks in one process");
    std::lock_guard<std::mutex> lock(g->call_mu);
    // whole images per device, no exchange of any kind: a failure of one image must not leave the others' ranks waiting
    // anywhere, so the ranks report instead of aborting the group
    std::vector<int> rcs(g->n_local, KMG_OK);
    std::vector<std::string> errs(g->n_local);
    const int rc = run_all(g, false, [&](GroupRank &r) {
        for (uint32_t i = r.idx; i < n_images; i += g->n_local) {
            const int one = kmg_reduce(r.p, rgba[i], widths[i], heights[i], color_count, algo, mode, out_rgba[i]);
            if (one != KMG_OK && rcs[r.idx] == KMG_OK) { rcs[r.idx] = one; errs[r.idx] = "image " + std::to_string(i) + ": " + kmg_last_error(); }
        }
        return KMG_OK;
    });
    KMG_TRY(rc);
    for (uint32_t i = 0; i < g->n_local; ++i)
        if (rcs[i] != KMG_OK) return fail(rcs[i], "%s", errs[i].c_str());
    return KMG_OK;
}
KMG_ABI_CATCH
