// graph_gap.hip -- 255 dependent small launches (the shape of the initialisation of a small image): back to back on a stream
// against the same launches captured once into a hipGraph and replayed.  Does the graph shorten the kernel-to-kernel gap?
//   hipcc -O3 --offload-arch=gfx950 -o /tmp/graph_gap tools/graph_gap.hip && /tmp/graph_gap
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

__global__ __launch_bounds__(256) void k_step(float *buf, int n, int j, int work)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float x = buf[i] + (float)j;
    for (int w = 0; w < work; ++w) x = __builtin_fmaf(x, 0.999f, 0.001f);
    buf[i] = x;
}

int main()
{
    const int n = 43776, launches = 255;
    float *buf; hipMalloc(&buf, n * 4); hipMemset(buf, 0, n * 4);
    hipStream_t st; hipStreamCreate(&st);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int work : {0, 256, 1024}) {
        auto issue = [&] { for (int j = 0; j < launches; ++j) hipLaunchKernelGGL(k_step, dim3((n + 255) / 256), dim3(256), 0, st, buf, n, j, work); };
        issue(); hipStreamSynchronize(st);
        float ms_stream = 0, ms_graph = 0;
        hipEventRecord(e0, st);
        for (int r = 0; r < 5; ++r) issue();
        hipEventRecord(e1, st); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms_stream, e0, e1);
        hipGraph_t g; hipGraphExec_t ge;
        hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
        issue();
        hipStreamEndCapture(st, &g);
        hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
        hipGraphLaunch(ge, st); hipStreamSynchronize(st);
        hipEventRecord(e0, st);
        for (int r = 0; r < 5; ++r) hipGraphLaunch(ge, st);
        hipEventRecord(e1, st); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms_graph, e0, e1);
        printf("work %4d fma per thread: stream %.2f us per launch, graph %.2f us per launch\n", work,
               ms_stream * 1e3 / (5 * launches), ms_graph * 1e3 / (5 * launches));
        hipGraphExecDestroy(ge); hipGraphDestroy(g);
    }
    return 0;
}
