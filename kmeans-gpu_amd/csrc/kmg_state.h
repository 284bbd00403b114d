// kmg_state.h -- the objects behind the opaque handles of include/kmeans_hip.h (kmg_processor, kmg_lloyd) and what the
// translation units of the C ABI share: kmg_processor.hip (processor, device blocks, host helpers), kmg_lloyd.hip (one Lloyd
// problem: colour table, initialisation, passes, loop), kmg_apply.hip (output passes), kmg_api.hip (the host-buffer calls of
// ImageProcessor::{palette, find, reduce}), kmg_group.hip (several devices).  Internal: nothing here is exported.
#pragma once

#include <hip/hip_runtime.h>

#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <exception>
#include <mutex>
#include <thread>
#include <new>
#include <vector>

#include "kmg_color.h"
#include "kmg_internal.h"
#include "kmg_kernels.h"
#include "kmg_octree.h"
#include "kmg_table.h"

using namespace kmg;

#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess)                                                                  \
            return fail(e_ == hipErrorOutOfMemory ? KMG_ERR_OUT_OF_MEMORY : KMG_ERR_HIP,       \
                        "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

// launch wrapper: when profiling is on, bracket the launch with HIP events on its own stream
// (events come from a pool that kmg_lloyd_profile_read recycles: no create/destroy in the hot loop)
#define PROF_LAUNCH(s_, id_, st_, expr)                                                        \
    do {                                                                                       \
        if ((s_)->prof & (1u << (id_))) {                                                      \
            ProfEvent pe_; pe_.id = (id_);                                                     \
            if ((s_)->pool.size() >= 2) {                                                      \
                pe_.e0 = (s_)->pool.back(); (s_)->pool.pop_back();                             \
                pe_.e1 = (s_)->pool.back(); (s_)->pool.pop_back();                             \
            } else {                                                                           \
                HIP_TRY(hipEventCreate(&pe_.e0)); HIP_TRY(hipEventCreate(&pe_.e1));            \
            }                                                                                  \
            HIP_TRY(hipEventRecord(pe_.e0, (st_)));                                            \
            HIP_TRY(expr);                                                                     \
            HIP_TRY(hipEventRecord(pe_.e1, (st_)));                                            \
            (s_)->events.push_back(pe_);                                                       \
        } else {                                                                               \
            HIP_TRY(expr);                                                                     \
        }                                                                                      \
    } while (0)

// ---------------------------------------------------------------------------------------------
// processor
// ---------------------------------------------------------------------------------------------
struct kmg_processor {
    int device;
    kmg_options opt;
    std::atomic<int> strategy{0};   // KMG_STRATEGY_* (kmg_options.strategy, kmg_processor_set_strategy)
    float *d_lut;            // 256 x f32: sRGB decode * 100, then 256 x f32: thresholds of the sRGB8 encode (k_meld)
    std::mutex mu;           // guards the lazily built static tables below
    CellBounds *d_bounds;    // kCells static cell bounds of the colour-table strategy
    CellBounds *d_sub_bounds;   // kSubCells static bounds of the 4x4x4 sub-cells
    float4 *d_lab_table;     // 2^24 x (L, a, b, C): Lab of every colour (256 MiB, built with d_bounds)
    float *d_sub_affine;     // sub_affine_bytes(): affine feature models per sub-cell (dominance test of k_cube_small / k_cube_one),
                             // built on the first colour-table pass with k <= 256
    bool affine_failed;      // ... or not at all (allocation failed: the pass runs without the test)
    std::vector<hipStream_t> idle_streams;   // streams of finished host-buffer calls, reused by the next ones (mu)
    // Device blocks the processor keeps between uses (mu): output-pass scratch, colour tables and workspaces of finished
    // kmg_lloyd objects.  A block is handed out again to the next request it is large enough for (block_take), so a second
    // image on a warm processor -- a frame loop, the two images per rank of BASELINE config 4 -- binds without a hipMalloc
    // (nine of them cost 1.4 ms per 8192^2 image in round 2, three times the kernels of the binding).
    std::vector<std::pair<void *, size_t>> idle_arenas;
    uint64_t n_block_malloc = 0, n_block_reuse = 0;   // block_take: fresh hipMallocs / blocks handed out again (kmg_debug_block_counts)
    hipMemPool_t pool;       // private stream-ordered pool for per-call scratch (never the device's default pool)
    // 1 MiB of page-locked host memory in 8 KiB slots (mu): where the loops read their few bytes back to (the convergence count
    // every check_period iterations, the centroids at the end: 22 us through pageable memory, 13 us through page-locked).  A
    // kmg_lloyd holds a slot for its lifetime; without one (memory not obtained, all slots taken) it reads back as before.
    void *h_page = nullptr;
    bool h_page_tried = false;
    std::vector<uint16_t> h_free;
};

constexpr size_t kHostPageBytes = (size_t)1 << 20, kHostSlotBytes = 8192;

void *host_slot_take(kmg_processor *p);
void host_slot_give(kmg_processor *p, void *slot);
// Device blocks the processor keeps between uses (kmg_processor.hip): smallest idle block that is large enough, else a fresh
// one / back to the idle list (the caller guarantees that no kernel still uses the block)
hipError_t block_take(kmg_processor *p, size_t bytes, void **ptr, size_t *cap);
void block_give(kmg_processor *p, void *ptr, size_t cap);

static inline size_t pad256(size_t bytes) { return (bytes + 255u) & ~(size_t)255u; }

struct ProfEvent { int id; hipEvent_t e0, e1; };

// colour table of a bound image (kmg_table.h)
struct ColourTable {
    const uint8_t *rgba = nullptr;   // the bound device buffer
    uint64_t n = 0;
    uint32_t *d_hist = nullptr;      // 2^24 counts, cell-major colour order
    int64_t *d_agg = nullptr;        // kCells x 4 per-cell sums of the image
    int64_t *d_sub_agg = nullptr;    // kSubCells x 4 per-sub-cell sums of the image
    uint8_t *d_occ = nullptr;        // 2^24 bits: colours the image has pixels of
    void *d_cell_work = nullptr;     // cube_work_bytes(): records the cube pass's stage kernel leaves for its scan kernel
    uint64_t *d_masks = nullptr;     // kCells x words candidate masks
    uint32_t *d_work = nullptr;      // kWorkWords: dense list of the occupied cells, then the hot cells (kmg_table.h)
    uint32_t n_hot = 0;              // host copy of the number of hot cells of the bound image
    uint32_t n_occ = 0;              // host copy of the number of occupied cells of the bound image (d_work[0])
    uint16_t *d_balance = nullptr;   // cube_balance_bytes(): the one-launch cube pass's deal of its tasks, two sets (kmg_table.h CubeBalance)
    const uint32_t *balance_work = nullptr;   // the work list the one-launch cube pass last walked with CubeBalance, and how often
    uint32_t balance_pass = 0;
    uint32_t *share_buf = nullptr;   // storage of d_work_share
    uint32_t *d_work_share = nullptr;   // 1 + kCells: this rank's share of the work list (kmg_lloyd_set_cell_share), or NULL = all of it
    bool tables_valid = false;       // label tables describe the CURRENT centroid table
    bool entries_valid = false;      // ... including the cells' pair entries / summaries (kmg_lloyd_run defers them to its last pass)
    void *d_colour_labels = nullptr; // 2^24 x u8 (k <= 256) or u16
    uint16_t *d_sub = nullptr;       // kSubCells 4x4x4 summaries (u16), kCells 8x8x8 summaries (u16), kCells pair entries (u32)
    // second set of label tables (kmg_lloyd_iterate): the cube pass of iteration t + 1 writes one set while the
    // label pass of iteration t still reads the other; d_colour_labels / d_sub are always the set written last
    void *d_colour_labels_alt = nullptr;
    uint16_t *d_sub_alt = nullptr;
    // farthest-point init over the colours (built on demand by the init entry points)
    uint32_t *d_tie = nullptr;       // 2^24: 1 + largest low half of the init key per colour, 0 = unoccupied
    float *d_cdist = nullptr;        // 2^24 running min-distance per colour
    void *d_init_cells = nullptr;    // init_scratch_bytes(): the passes' cell records and slots
    bool tie_valid = false;          // d_tie describes (rgba, n, tie_first)
    bool bound_by_init = false;      // the binding was made by the initialisation of the current problem
    bool bound_by_caller = false;    // kmg_lloyd_bind_image / kmg_lloyd_prepare: the caller vouches for the buffer's contents
    uint64_t tie_first = 0;
    // the three blocks all of the above are carved from (kmg_processor::idle_arenas): the tables proper, the second label-table
    // set of kmg_lloyd_iterate, the tables of the initialisation
    void *blk = nullptr, *blk_alt = nullptr, *blk_init = nullptr;
    size_t blk_cap = 0, blk_alt_cap = 0, blk_init_cap = 0;
};

struct kmg_lloyd {
    kmg_processor *p;
    uint32_t k;
    Centroid *d_cent;            // k
    int64_t *d_partials;         // 2048 x k x 4: partial sums of the per-pixel scan; between passes also scratch of whoever runs --
                                 // the key slots of a per-pixel initialisation, the sum / centroid rotation of the small-image loop
    int64_t *d_acc;              // k x 4 (used by kmg_lloyd_run)
    int64_t *d_acc_int;          // k x 4: where the cube pass accumulates; ZERO between passes (its last launch hands the sums
                                 // over and clears it, kmg_table.h CubeTail) -- no memset launch per pass
    bool acc_int_dirty;          // a pass was interrupted: clear d_acc_int before the next one
    uint32_t *d_nconv;           // 1
    void *h_slot = nullptr;      // kHostSlotBytes page-locked bytes for small read-backs (kmg_processor::h_page), or NULL
    unsigned long long *d_key;   // 1 (init arg-max of a sharded image)
    float *d_dist;               // init distance map, grown on demand (a block of its own)
    uint64_t dist_cap;
    size_t dist_blk_cap;
    void *ws;                    // the block d_cent .. d_key are carved from (kmg_processor::idle_arenas)
    size_t ws_cap;
    uint32_t last_rows;          // rows of d_partials written by the last assign pass
    bool init_colours;           // the running sharded init (kmg_lloyd_init_step) walks colours, not pixels
    uint32_t reserve_cus = 0;    // CUs the label pass leaves free (kmg_lloyd_reserve_cus)
    bool pooled;                 // workspace came from the stream-ordered pool of `pool_stream` (internal per-call objects)
    hipStream_t pool_stream;
    ColourTable tab;
    // kmg_lloyd_iterate: label passes run on a stream of their own, beside the next iteration's cube pass
    hipStream_t side;            // high-priority stream of the label passes (created on first use)
    hipEvent_t ev_cube;          // cube pass of the current iteration done (main stream -> side stream)
    hipEvent_t ev_lab[2];        // label pass reading table set i done (side stream -> main stream)
    bool lab_pending[2];
    int set;                     // event slot of the table set written last
    uint32_t prof;               // per-launch HIP-event timing: bit i = time kernel id i (kmg_lloyd_profile)
    std::vector<ProfEvent> events;
    std::vector<hipEvent_t> pool; // recycled timing events
};

static inline hipStream_t S(void *s) { return (hipStream_t)s; }

// What a cost model's decision is overridden with: +1 = colour table / candidate lists, -1 = per-pixel scan, 0 = the model decides
// (kmg_options.strategy; the tools build also honours the environment variable KMG_STRATEGY = brute | table)
static inline int forced_strategy(const kmg_processor *p)
{
    if (const char *e = KMG_TOOLS_ENV("KMG_STRATEGY")) {
        if (!strcmp(e, "brute")) return -1;
        if (!strcmp(e, "table")) return 1;
    }
    const int st = p->strategy.load(std::memory_order_relaxed) & 3;
    return st == KMG_STRATEGY_TABLE ? 1 : (st == KMG_STRATEGY_SCAN ? -1 : 0);
}

namespace {
struct DevBuf {
    void *ptr = nullptr;
    ~DevBuf() { if (ptr) (void)hipFree(ptr); }
    hipError_t alloc(size_t bytes) { return hipMalloc(&ptr, bytes); }
};

// scratch that lives for one call on one stream: stream-ordered allocation from the device's memory
// pool (a reused block after the first call instead of a ~0.1 ms hipMalloc + hipFree pair)
static hipError_t pool_alloc(kmg_processor *p, void **ptr, size_t bytes, hipStream_t stream)
{
    return p->pool ? hipMallocFromPoolAsync(ptr, bytes, p->pool, stream) : hipMallocAsync(ptr, bytes, stream);
}

struct StreamBuf {
    void *ptr = nullptr;
    hipStream_t st = nullptr;
    ~StreamBuf() { if (ptr) (void)hipFreeAsync(ptr, st); }
    hipError_t alloc(kmg_processor *p, size_t bytes, hipStream_t stream) { st = stream; return pool_alloc(p, &ptr, bytes, stream); }
};

// The scratch of one output pass (kmg_dev_apply): ONE block per call, taken from / returned to the processor's idle
// list and grown on demand, carved up by take().  Calls on one processor may run concurrently (examples/parallel.rs),
// so each holds its own block; the call synchronises its stream before it returns, so a returned block is idle.
// (Stream-ordered pool allocations were measured here first: a hipFreeAsync of a 16 MiB buffer takes up to 0.37 ms
// of host time on this runtime whatever the pool's release threshold -- more than the kernels of a replace pass.)
struct ArenaGuard {
    kmg_processor *p = nullptr;
    void *base = nullptr;
    size_t cap = 0, used = 0;
    hipError_t acquire(kmg_processor *proc, size_t bytes)
    {
        p = proc;
        return block_take(p, bytes, &base, &cap);
    }
    static size_t padded(size_t bytes) { return pad256(bytes); }
    void *take(size_t bytes)
    {
        void *r = (uint8_t *)base + used;
        used += padded(bytes);
        return used <= cap ? r : nullptr;
    }
    ~ArenaGuard() { if (base) block_give(p, base, cap); }
};

// the private stream of one host-buffer call (every call has its own, so calls on one processor run
// concurrently, examples/parallel.rs); taken from / returned to the processor's idle list
struct StreamGuard {
    kmg_processor *p = nullptr;
    hipStream_t st = nullptr;
    hipError_t acquire(kmg_processor *proc)
    {
        p = proc;
        {
            std::lock_guard<std::mutex> lock(p->mu);
            if (!p->idle_streams.empty()) { st = p->idle_streams.back(); p->idle_streams.pop_back(); return hipSuccess; }
        }
        return hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    }
    ~StreamGuard()
    {
        if (!st) return;
        std::lock_guard<std::mutex> lock(p->mu);
        p->idle_streams.push_back(st);
    }
};

}  // namespace

// kmg_lloyd.hip
int ensure_bounds(kmg_processor *p, hipStream_t st);
const float *affine_for(kmg_processor *p, uint32_t k, hipStream_t st);
int lloyd_create_impl(kmg_processor *p, uint32_t k, kmg_lloyd **out, hipStream_t pool_stream);
static inline size_t sub_table_bytes() { return sizeof(uint16_t) * (kSubCells + kCells) + sizeof(uint32_t) * kCells; }
