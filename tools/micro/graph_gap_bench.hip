// Micro-benchmark (round 4): what a hipGraph replay saves per launch boundary on a chain of dependent launches like the train
// step's (about 25 kernels, one fork / join onto a second stream), against plain stream launches.  The kernels do next to nothing, so
// the time per kernel IS the boundary.
//   hipcc --offload-arch=gfx950 -O3 -o graph_gap_bench graph_gap_bench.hip && ./graph_gap_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void k_busy(float* p, int spins) {
    for (int i = 0; i < spins; ++i) __builtin_amdgcn_s_sleep(8);
    if (threadIdx.x == 0 && p) p[blockIdx.x] += 1.f;
}
static int chain(hipStream_t s, hipStream_t side, hipEvent_t fork, hipEvent_t join, float* p, int n, int us, bool two) {
    for (int i = 0; i < n; ++i) {
        if (two && i == n / 2) {
            CK(hipEventRecord(fork, s));
            CK(hipStreamWaitEvent(side, fork, 0));
            hipLaunchKernelGGL(k_busy, dim3(96), dim3(256), 0, side, p + 4096, 8 * us);
            CK(hipEventRecord(join, side));
        }
        if (two && i == n - 2) CK(hipStreamWaitEvent(s, join, 0));
        hipLaunchKernelGGL(k_busy, dim3(160), dim3(256), 0, s, p, us);
    }
    return 0;
}
int main() {
    float* p; CK(hipMalloc(&p, 1 << 20)); CK(hipMemset(p, 0, 1 << 20));
    hipStream_t s, side; CK(hipStreamCreate(&s)); CK(hipStreamCreateWithPriority(&side, hipStreamNonBlocking, -1));
    hipEvent_t e0, e1, fork, join; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventCreateWithFlags(&fork, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&join, hipEventDisableTiming));
    const int n = 25, reps = 50;
    for (int two = 0; two < 2; ++two)
        for (int us : {0, 50}) {
            float ms_s = 0.f, ms_g = 0.f;
            for (int w = 0; w < 3; ++w) if (chain(s, side, fork, join, p, n, us, two)) return 1;
            CK(hipStreamSynchronize(s));
            CK(hipEventRecord(e0, s));
            for (int r = 0; r < reps; ++r) if (chain(s, side, fork, join, p, n, us, two)) return 1;
            CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms_s, e0, e1));
            hipGraph_t g; hipGraphExec_t ge;
            CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
            if (chain(s, side, fork, join, p, n, us, two)) return 1;
            CK(hipStreamEndCapture(s, &g));
            CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
            for (int w = 0; w < 3; ++w) CK(hipGraphLaunch(ge, s));
            CK(hipStreamSynchronize(s));
            CK(hipEventRecord(e0, s));
            for (int r = 0; r < reps; ++r) CK(hipGraphLaunch(ge, s));
            CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms_g, e0, e1));
            printf("%d kernels (%2d sleeps each)%s: stream launches %6.1f us per chain = %.2f us per kernel, graph replay %6.1f us = %.2f\n", n, us,
                   two ? " + a forked kernel on a second stream" : "", ms_s * 1e3 / reps, ms_s * 1e3 / reps / n, ms_g * 1e3 / reps, ms_g * 1e3 / reps / n);
            CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
        }
    return 0;
}
