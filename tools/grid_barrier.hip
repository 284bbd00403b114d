// grid_barrier.hip -- what a device-wide barrier per pick would cost a single-launch farthest-point initialisation (VERDICT r05
// item 3): 256 co-resident workgroups (one per CU), per step every workgroup writes a 16-byte slot (its best key) and `dirty` bytes of
// other data (the distances of the cells the pick reached), all meet at a barrier in device memory, then every workgroup reads all
// 256 slots.  Variants of the barrier: one counter per step (every workgroup polls it), or a two-level one (one counter per XCD-sized
// group of 32 workgroups, then a counter of the 8 groups).  Reported: us per step.
// Third variant: no counter and no fence at all -- the slot itself carries the step number, thread t of every workgroup polls slot t
// until it shows this step (the dirty data is private to its workgroup: nothing else has to become visible).
//   hipcc -O3 -w --offload-arch=gfx950 -o /tmp/grid_barrier tools/grid_barrier.hip && /tmp/grid_barrier     (profiles/r06_grid_barrier.txt)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

constexpr int kGrid = 256, kBlock = 256;

template <int MODE>
__global__ __launch_bounds__(kBlock) void k_steps(unsigned long long *slots, unsigned int *counters, float *dirty, int dirty_floats,
                                                  int steps, unsigned long long *sink)
{
    __shared__ unsigned long long s_best;
    unsigned long long acc = 0;
    for (int step = 0; step < steps; ++step) {
        // the step's work: `dirty_floats` floats of this workgroup's own data change, its slot is written
        for (int i = threadIdx.x; i < dirty_floats; i += kBlock) dirty[(size_t)blockIdx.x * dirty_floats + i] += 1.0f;
        if (threadIdx.x == 0) {
            __hip_atomic_store(&slots[(step & 1) * kGrid + blockIdx.x], ((unsigned long long)(step + 1) << 32) | (blockIdx.x * 2654435761u), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        if (MODE == 2) {
            // the slots ARE the barrier: slot t shows this step's tag once workgroup t has written it
            unsigned long long v;
            do {
                v = __hip_atomic_load(&slots[(step & 1) * kGrid + threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } while ((unsigned)(v >> 32) != (unsigned)step + 1u);
            if (threadIdx.x == 0) s_best = 0;
            __syncthreads();
            atomicMax(&s_best, v);
            __syncthreads();
            acc += s_best;
            continue;
        }
        // ---- the barrier: release our writes, count in, wait for everybody, acquire ----
        if (threadIdx.x == 0) {
            __atomic_thread_fence(__ATOMIC_RELEASE);                 // (agent scope by default: this workgroup's writes reach memory)
            if (MODE == 0) {
                __hip_atomic_fetch_add(&counters[step], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                while (__hip_atomic_load(&counters[step], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)kGrid) __builtin_amdgcn_s_sleep(1);
            } else {
                const int group = blockIdx.x >> 5;
                unsigned int *c = counters + (size_t)step * 16;
                if (__hip_atomic_fetch_add(&c[group], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 31u)
                    __hip_atomic_fetch_add(&c[8], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                while (__hip_atomic_load(&c[8], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < 8u) __builtin_amdgcn_s_sleep(1);
            }
            __atomic_thread_fence(__ATOMIC_ACQUIRE);
        }
        __syncthreads();
        // every workgroup reduces the 256 slots (one per thread)
        const unsigned long long v = __hip_atomic_load(&slots[(step & 1) * kGrid + threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (threadIdx.x == 0) s_best = 0;
        __syncthreads();
        atomicMax(&s_best, v);
        __syncthreads();
        acc += s_best;
    }
    if (threadIdx.x == 0) sink[blockIdx.x] = acc;
}

int main()
{
    const int steps = 255;
    unsigned long long *slots, *sink;
    unsigned int *counters;
    float *dirty;
    hipMalloc(&slots, sizeof(unsigned long long) * 2 * kGrid);
    hipMalloc(&sink, sizeof(unsigned long long) * kGrid);
    hipMalloc(&counters, sizeof(unsigned int) * 16 * (steps + 1));
    hipMalloc(&dirty, sizeof(float) * kGrid * 4096);
    hipMemset(dirty, 0, sizeof(float) * kGrid * 4096);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 3; ++mode)
        for (int dirty_floats : {0, 4, 1024, 4096}) {             // 0 B / 16 B / 4 KiB / 16 KiB of dirty data per workgroup and step
            float best = 1e9f;
            for (int rep = 0; rep < 4; ++rep) {
                hipMemset(counters, 0, sizeof(unsigned int) * 16 * (steps + 1));
                hipMemset(slots, 0, sizeof(unsigned long long) * 2 * kGrid);
                hipDeviceSynchronize();
                hipEventRecord(e0, 0);
                if (mode == 2) hipLaunchKernelGGL(k_steps<2>, dim3(kGrid), dim3(kBlock), 0, 0, slots, counters, dirty, dirty_floats, steps, sink);
                else if (mode == 0) hipLaunchKernelGGL(k_steps<0>, dim3(kGrid), dim3(kBlock), 0, 0, slots, counters, dirty, dirty_floats, steps, sink);
                else hipLaunchKernelGGL(k_steps<1>, dim3(kGrid), dim3(kBlock), 0, 0, slots, counters, dirty, dirty_floats, steps, sink);
                hipEventRecord(e1, 0); hipEventSynchronize(e1);
                float ms = 0; hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) best = ms;
            }
            printf("%s barrier, %5d dirty bytes per workgroup and step: %.2f us per step (%d steps, %.3f ms)\n", mode == 2 ? "slots-as-flags" : (mode ? "two-level" : "one-counter"),
                   dirty_floats * 4, best * 1e3 / steps, steps, best);
        }
    return 0;
}
