// Micro-benchmark: cycles per v_mfma_f32_32x32x16_bf16 for one wave per SIMD, with dependent / independent accumulators
// and with fillers (VALU / SALU / LDS reads) between the MFMAs.  hipcc --offload-arch=gfx950 -O3 x3_issue_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int CHAINS, int VALU, int SALU, int LDSR, int WAVES>
__global__ __launch_bounds__(64 * WAVES, 1) void k(long long* out, float* sink, int iters) {
    __shared__ __attribute__((aligned(16))) unsigned L[4096];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) L[i] = i * 2654435761u;
    __syncthreads();
    f32x16 acc[4];
    for (int c = 0; c < 4; ++c) for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(1.0f + threadIdx.x * 0.001f + e); b[e] = (__bf16)(0.5f + e * 0.25f); }
    float v[8];
    for (int e = 0; e < 8; ++e) v[e] = threadIdx.x + e;
    int sc = iters;
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    u4 lv = {0, 0, 0, 0};
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            acc[m % CHAINS] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[m % CHAINS], 0, 0, 0);
#pragma unroll
            for (int f = 0; f < VALU; ++f) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[f % 8]) : "v"(v[(f + 1) % 8]));
#pragma unroll
            for (int f = 0; f < SALU; ++f) asm volatile("s_add_i32 %0, %0, 1" : "+s"(sc));
#pragma unroll
            for (int f = 0; f < LDSR; ++f) {
                u4 t = *reinterpret_cast<const u4*>(&L[((threadIdx.x * 4) + 64 * f + 16 * m) & 4092]);
                asm volatile("" : "+v"(t));
                lv ^= t;
            }
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int c = 0; c < 4; ++c) for (int r = 0; r < 16; ++r) s += acc[c][r];
    for (int e = 0; e < 8; ++e) s += v[e];
    sink[blockIdx.x * blockDim.x + threadIdx.x] = s + sc + lv[0] + lv[1] + lv[2] + lv[3];
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
}
template <int CHAINS, int VALU, int SALU, int LDSR, int WAVES>
void run(const char* name, long long* d_out, float* sink) {
    const int iters = 200;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<CHAINS, VALU, SALU, LDSR, WAVES>), dim3(256), dim3(64 * WAVES), 0, 0, d_out, sink, iters);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<CHAINS, VALU, SALU, LDSR, WAVES>), dim3(256), dim3(64 * WAVES), 0, 0, d_out, sink, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long t; hipMemcpy(&t, d_out, 8, hipMemcpyDeviceToHost);
    const double per = (double)t / (iters * 16.0);
    printf("%-44s ticks/MFMA %6.1f   kernel %7.1f us  -> %.2f ticks/ns\n", name, per, ms * 1e3, (double)t / (ms * 1e6));
}
int main() {
    long long* d_out; float* sink;
    hipMalloc(&d_out, 64); hipMalloc(&sink, 256 * 512 * 4);
    run<1, 0, 0, 0, 4>("1 wave/SIMD, 1 chain", d_out, sink);
    run<4, 0, 0, 0, 4>("1 wave/SIMD, 4 chains", d_out, sink);
    run<1, 4, 0, 0, 4>("1 wave/SIMD, 1 chain + 4 VALU", d_out, sink);
    run<1, 6, 0, 0, 4>("1 wave/SIMD, 1 chain + 6 VALU", d_out, sink);
    run<1, 8, 0, 0, 4>("1 wave/SIMD, 1 chain + 8 VALU", d_out, sink);
    run<4, 6, 0, 0, 4>("1 wave/SIMD, 4 chains + 6 VALU", d_out, sink);
    run<1, 4, 2, 0, 4>("1 wave/SIMD, 1 chain + 4 VALU + 2 SALU", d_out, sink);
    run<1, 4, 0, 1, 4>("1 wave/SIMD, 1 chain + 4 VALU + 1 ds_read_b128", d_out, sink);
    run<1, 0, 0, 1, 4>("1 wave/SIMD, 1 chain + 1 ds_read_b128", d_out, sink);
    run<1, 0, 0, 0, 8>("2 waves/SIMD, 1 chain", d_out, sink);
    run<1, 6, 0, 0, 8>("2 waves/SIMD, 1 chain + 6 VALU", d_out, sink);
    run<1, 12, 0, 0, 8>("2 waves/SIMD, 1 chain + 12 VALU", d_out, sink);
    return 0;
}
