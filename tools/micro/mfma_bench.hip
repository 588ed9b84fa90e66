// Microbenchmark: what does one SIMD sustain on v_mfma_f32_32x32x2_f32 in loops shaped like ours?
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }

// mode 0: registers only; 1: operands via ds_read_b128 from LDS each group; 2: + barrier pair and LDS store per 64 MFMAs
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed) {
    __shared__ __attribute__((aligned(16))) float As[128 * 36];
    __shared__ __attribute__((aligned(16))) float Bs[128 * 36];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, wm = wv >> 1, wn = wv & 1;
    for (int i = tid; i < 128 * 36; i += 256) { As[i] = seed * (i % 7); Bs[i] = seed * (i % 5); }
    __syncthreads();
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const float* la = As + (wm * 64 + (lane & 31)) * 36 + 4 * (lane >> 5);
    const float* lb = Bs + (wn * 64 + (lane & 31)) * 36 + 4 * (lane >> 5);
    float4 ra = make_float4(seed, seed * 2, seed * 3, seed * 4);
    for (int it = 0; it < iters; ++it) {
        if (MODE == 2) {
            *reinterpret_cast<float4*>(&As[(tid >> 3) * 36 + (tid & 7) * 4]) = ra;
            *reinterpret_cast<float4*>(&As[((tid >> 3) + 32) * 36 + (tid & 7) * 4]) = ra;
            *reinterpret_cast<float4*>(&As[((tid >> 3) + 64) * 36 + (tid & 7) * 4]) = ra;
            *reinterpret_cast<float4*>(&As[((tid >> 3) + 96) * 36 + (tid & 7) * 4]) = ra;
            *reinterpret_cast<float4*>(&Bs[(tid >> 3) * 36 + (tid & 7) * 4]) = ra;
            *reinterpret_cast<float4*>(&Bs[((tid >> 3) + 32) * 36 + (tid & 7) * 4]) = ra;
            *reinterpret_cast<float4*>(&Bs[((tid >> 3) + 64) * 36 + (tid & 7) * 4]) = ra;
            *reinterpret_cast<float4*>(&Bs[((tid >> 3) + 96) * 36 + (tid & 7) * 4]) = ra;
            __syncthreads();
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float4 a0, a1, q0, q1;
            if (MODE >= 1) {
                a0 = *reinterpret_cast<const float4*>(la + 8 * g);
                a1 = *reinterpret_cast<const float4*>(la + 32 * 36 + 8 * g);
                q0 = *reinterpret_cast<const float4*>(lb + 8 * g);
                q1 = *reinterpret_cast<const float4*>(lb + 32 * 36 + 8 * g);
            } else {
                a0 = ra; a1 = ra; q0 = ra; q1 = ra;
                asm volatile("" : "+v"(a0.x), "+v"(a1.x), "+v"(q0.x), "+v"(q1.x));
            }
            acc[0][0] = mfma32(a0.x, q0.x, acc[0][0]); acc[0][1] = mfma32(a0.x, q1.x, acc[0][1]);
            acc[1][0] = mfma32(a1.x, q0.x, acc[1][0]); acc[1][1] = mfma32(a1.x, q1.x, acc[1][1]);
            acc[0][0] = mfma32(a0.y, q0.y, acc[0][0]); acc[0][1] = mfma32(a0.y, q1.y, acc[0][1]);
            acc[1][0] = mfma32(a1.y, q0.y, acc[1][0]); acc[1][1] = mfma32(a1.y, q1.y, acc[1][1]);
            acc[0][0] = mfma32(a0.z, q0.z, acc[0][0]); acc[0][1] = mfma32(a0.z, q1.z, acc[0][1]);
            acc[1][0] = mfma32(a1.z, q0.z, acc[1][0]); acc[1][1] = mfma32(a1.z, q1.z, acc[1][1]);
            acc[0][0] = mfma32(a0.w, q0.w, acc[0][0]); acc[0][1] = mfma32(a0.w, q1.w, acc[0][1]);
            acc[1][0] = mfma32(a1.w, q0.w, acc[1][0]); acc[1][1] = mfma32(a1.w, q1.w, acc[1][1]);
        }
        if (MODE == 2) __syncthreads();
    }
    float s = 0.f;
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    out[blockIdx.x * 256 + tid] = s;
}

template <int MODE>
void run(const char* name, int blocks, int iters, float* out) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters, 1e-3f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters, 1e-3f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double mfma_per_simd = (double)iters * 64.0 * blocks / 256.0;   // one wave per SIMD per block
    const double tf = (double)blocks * 4 * iters * 64 * 4096.0 / (ms * 1e-3) / 1e12;
    printf("%-28s blocks %4d iters %5d: %8.1f us  %6.1f TFLOP/s  (%.0f ns per MFMA-slot, ideal 26.7 at 2.4 GHz)\n", name,
           blocks, iters, ms * 1e3, tf, ms * 1e6 / mfma_per_simd);
}

int main() {
    float* out; hipMalloc(&out, 4096 * 256 * 4);
    for (int blocks : {256, 512, 768}) {
        run<0>("regs only", blocks, 200, out);
        run<1>("ds_read_b128 operands", blocks, 200, out);
        run<2>("+ LDS store + 2 barriers", blocks, 200, out);
    }
    return 0;
}
