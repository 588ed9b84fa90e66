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
// Two waves per SIMD with DIFFERENT roles: waves 0-3 issue MFMAs back to back, waves 4-7 a stream of `KIND` instructions
// (0: independent v_add_f32, 1: one dependent v_add_f32 chain, 2: v_cvt_pk_bf16_f32 + shift + subtract (the operand split),
// 3: ds_write_b64).  Reports the cycles per instruction the second group achieves beside the first.
template <int KIND>
__global__ __launch_bounds__(512, 1) void k_roles(long long* out, float* sink, int iters) {
    __shared__ __attribute__((aligned(16))) unsigned L[8192];
    const int grp = threadIdx.x >> 8;
    f32x16 acc; for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(1.0f + threadIdx.x * 0.001f + e); b[e] = (__bf16)(0.5f + e * 0.25f); }
    float v[8];
    for (int e = 0; e < 8; ++e) v[e] = threadIdx.x + e * 0.37f;
    unsigned w = threadIdx.x;
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    if (grp == 0) {
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int m = 0; m < 16; ++m) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
    } else {
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int m = 0; m < 64; ++m) {
                if (KIND == 0) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[m % 8]) : "v"(v[(m + 3) % 8]));
                else if (KIND == 1) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[0]) : "v"(v[1]));
                else if (KIND == 2) {
                    asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(w) : "v"(v[m % 4]), "v"(v[4 + m % 4]));
                    asm volatile("v_lshlrev_b32 %0, 16, %1" : "=v"(w) : "v"(w));
                    asm volatile("v_sub_f32 %0, %0, %1" : "+v"(v[m % 4]) : "v"(w));
                } else {
                    asm volatile("ds_write_b64 %0, %1" :: "v"((unsigned)((threadIdx.x & 255) * 8 + (m & 7) * 2048)), "v"((unsigned long long)w) : "memory");
                }
            }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int r = 0; r < 16; ++r) s += acc[r];
    for (int e = 0; e < 8; ++e) s += v[e];
    sink[blockIdx.x * blockDim.x + threadIdx.x] = s + w + L[threadIdx.x];
    if (blockIdx.x == 0 && (threadIdx.x == 0 || threadIdx.x == 256)) out[1 + grp] = t1 - t0;
}
template <int KIND>
void run_roles(const char* name, long long* d_out, float* sink) {
    const int iters = 200;
    hipLaunchKernelGGL((k_roles<KIND>), dim3(256), dim3(512), 0, 0, d_out, sink, iters);
    hipLaunchKernelGGL((k_roles<KIND>), dim3(256), dim3(512), 0, 0, d_out, sink, iters);
    hipDeviceSynchronize();
    long long t[3]; hipMemcpy(t, d_out, 24, hipMemcpyDeviceToHost);
    const int per_iter = KIND == 2 ? 64 * 3 : 64;
    printf("%-52s MFMA wave: %5.1f ticks/MFMA   other wave: %5.1f ticks/instruction\n", name, (double)t[1] / (iters * 16.0),
           (double)t[2] / (iters * (double)per_iter));
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
    run_roles<0>("roles: MFMA wave + independent v_add_f32 wave", d_out, sink);
    run_roles<1>("roles: MFMA wave + dependent v_add_f32 chain", d_out, sink);
    run_roles<2>("roles: MFMA wave + cvt_pk / shift / sub (split)", d_out, sink);
    run_roles<3>("roles: MFMA wave + ds_write_b64", d_out, sink);
    return 0;
}
