// Microbenchmark: what does straight-line code that is executed ONCE cost on gfx950?
// k_straight<N>: N dependent v_fma in a row (8 B each, fully unrolled); k_loop<N>: the same count
// as a 64-instruction loop body.  Both timed with s_memtime inside the kernel (wave 0 of every block)
// and with HIP events around the launch.  Build: hipcc --offload-arch=gfx950 -O3 icache_bench.hip -o icache_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

template <int N>
__global__ __launch_bounds__(256) void k_straight(float* out, unsigned long long* cyc, float a, float b) {
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);
    float v = threadIdx.x;
#pragma unroll
    for (int i = 0; i < N; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v) : "v"(a), "v"(b));
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);
    out[blockIdx.x * blockDim.x + threadIdx.x] = v;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int N>
__global__ __launch_bounds__(256) void k_loop(float* out, unsigned long long* cyc, float a, float b) {
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);
    float v = threadIdx.x;
#pragma unroll 1
    for (int j = 0; j < N / 64; ++j) {
#pragma unroll
        for (int i = 0; i < 64; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v) : "v"(a), "v"(b));
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);
    out[blockIdx.x * blockDim.x + threadIdx.x] = v;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <typename F>
static void run(const char* name, F launch, unsigned long long* cyc, int nblk) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        launch();
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> h(nblk);
        hipMemcpy(h.data(), cyc, nblk * 8, hipMemcpyDeviceToHost);
        unsigned long long mx = 0, mn = ~0ull; double av = 0;
        for (auto c : h) { mx = c > mx ? c : mx; mn = c < mn ? c : mn; av += c; }
        printf("%-22s rep %d: event %.1f us; in-kernel ticks min %llu avg %.0f max %llu\n", name, rep, ms * 1e3, mn, av / nblk, mx);
    }
}

int main() {
    const int nblk = 512;
    float* out; unsigned long long* cyc;
    hipMalloc(&out, nblk * 256 * 4); hipMalloc(&cyc, nblk * 8);
    // memtime tick rate
    run("straight 512 (4 KB)", [&] { hipLaunchKernelGGL(k_straight<512>, dim3(nblk), dim3(256), 0, 0, out, cyc, 1.0001f, 0.5f); }, cyc, nblk);
    run("loop     512", [&] { hipLaunchKernelGGL(k_loop<512>, dim3(nblk), dim3(256), 0, 0, out, cyc, 1.0001f, 0.5f); }, cyc, nblk);
    run("straight 2048 (16 KB)", [&] { hipLaunchKernelGGL(k_straight<2048>, dim3(nblk), dim3(256), 0, 0, out, cyc, 1.0001f, 0.5f); }, cyc, nblk);
    run("loop     2048", [&] { hipLaunchKernelGGL(k_loop<2048>, dim3(nblk), dim3(256), 0, 0, out, cyc, 1.0001f, 0.5f); }, cyc, nblk);
    run("straight 4096 (32 KB)", [&] { hipLaunchKernelGGL(k_straight<4096>, dim3(nblk), dim3(256), 0, 0, out, cyc, 1.0001f, 0.5f); }, cyc, nblk);
    run("loop     4096", [&] { hipLaunchKernelGGL(k_loop<4096>, dim3(nblk), dim3(256), 0, 0, out, cyc, 1.0001f, 0.5f); }, cyc, nblk);
    run("straight 8192 (64 KB)", [&] { hipLaunchKernelGGL(k_straight<8192>, dim3(nblk), dim3(256), 0, 0, out, cyc, 1.0001f, 0.5f); }, cyc, nblk);
    run("loop     8192", [&] { hipLaunchKernelGGL(k_loop<8192>, dim3(nblk), dim3(256), 0, 0, out, cyc, 1.0001f, 0.5f); }, cyc, nblk);
    return 0;
}
