// Per-CU intake against the bytes a CU keeps in flight: one workgroup per CU, W waves, every wave issues D independent
// 16-byte-per-lane loads (1 KB per wave instruction, coalesced), waits for all of them, and repeats; footprints of 16 MB
// (MALL / L2 resident after the first pass) and 1 GB (HBM).  Prints GB/s per CU and B / cycle at the measured clock.
//   hipcc --offload-arch=gfx950 -O3 -o intake_bench intake_bench.hip && ./intake_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef unsigned u4 __attribute__((ext_vector_type(4)));

template <int D>
__global__ __launch_bounds__(1024) void k_intake(const u4* __restrict__ src, size_t n16, int iters, unsigned* sink, long long* cyc) {
    const int waves = blockDim.x >> 6, wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    // a CU's stream: chunks of (waves x D) KB, advancing through the footprint; different CUs start at different places
    size_t pos = ((size_t)blockIdx.x * 7919u * 64u) % n16;
    unsigned acc = 0;
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        u4 v[D];
#pragma unroll
        for (int d = 0; d < D; ++d) {
            size_t i = pos + ((size_t)(d * waves + wv) * 64 + lane);
            if (i >= n16) i -= n16;
            v[d] = __builtin_nontemporal_load(src + i);
        }
#pragma unroll
        for (int d = 0; d < D; ++d) acc += v[d].x ^ v[d].w;
        pos += (size_t)D * waves * 64;
        if (pos >= n16) pos -= n16;
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    if (acc == 0x12345678u) sink[0] = acc;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int D>
static void run(const u4* buf, size_t bytes, int waves, unsigned* sink, long long* cyc, const char* what) {
    const size_t n16 = bytes / 16;
    const int iters = 400;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k_intake<D>, dim3(256), dim3(64 * waves), 0, 0, buf, n16, 50, sink, cyc);
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_intake<D>, dim3(256), dim3(64 * waves), 0, 0, buf, n16, iters, sink, cyc);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double per_cu = (double)iters * D * waves * 1024.0;
    printf("%-6s waves %2d  loads in flight per wave %2d  (%3d KB per CU in flight)  %6.1f GB/s per CU  %5.2f TB/s chip\n", what, waves, D,
           D * waves, per_cu / (ms * 1e-3) / 1e9, per_cu * 256 / (ms * 1e-3) / 1e12);
}

int main() {
    const size_t big = (size_t)1 << 30, small = (size_t)16 << 20;
    u4* buf;
    unsigned* sink;
    long long* cyc;
    CK(hipMalloc(&buf, big));
    CK(hipMemset(buf, 1, big));
    CK(hipMalloc(&sink, 64));
    CK(hipMalloc(&cyc, 256 * 8));
    for (int pass = 0; pass < 2; ++pass) {
        const size_t bytes = pass ? big : small;
        const char* what = pass ? "1 GB" : "16 MB";
        for (int waves : {4, 8, 16}) {
            run<2>(buf, bytes, waves, sink, cyc, what);
            run<4>(buf, bytes, waves, sink, cyc, what);
            run<8>(buf, bytes, waves, sink, cyc, what);
            run<16>(buf, bytes, waves, sink, cyc, what);
        }
    }
    return 0;
}
