// Probe (round 4): a structured buffer with stride 16 addresses 2^32 sixteen-byte records (64 GB) with ONE 32-bit index per
// lane -- the row-indexed step's way past the 4 GB a raw buffer's byte offset reaches.
//   hipcc --offload-arch=gfx950 -O3 -o struct_buffer_probe struct_buffer_probe.hip && ./struct_buffer_probe [GiB=5]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ f32x4 struct_load_b128(__amdgpu_buffer_rsrc_t rsrc, int vindex, int voffset, int soffset, int aux) __asm("llvm.amdgcn.struct.ptr.buffer.load.v4f32");

__global__ void k_fill(float* p, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const float v = (float)(i % 16777213u);
        reinterpret_cast<f32x4*>(p)[i] = f32x4{v, v + 0.25f, v + 0.5f, v + 0.75f};
    }
}
// thread t reads record idx[t] through the structured buffer and through a plain 64-bit address
__global__ void k_probe(const float* p, unsigned n_rec, const unsigned* idx, int n, f32x4* via_buf, f32x4* via_ptr) {
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p), 16, (int)n_rec, 0x00020000);
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    via_buf[t] = struct_load_b128(rs, (int)idx[t], 0, 0, 0);
    via_ptr[t] = idx[t] < n_rec ? reinterpret_cast<const f32x4*>(p)[idx[t]] : f32x4{0.f, 0.f, 0.f, 0.f};
}
int main(int argc, char** argv) {
    const double gib = argc > 1 ? atof(argv[1]) : 5.0;
    const size_t n4 = (size_t)(gib * (1ull << 30)) / 16;
    float* p; CK(hipMalloc(&p, n4 * 16));
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, p, n4);
    const int n = 1 << 16;
    unsigned* h = (unsigned*)malloc(n * 4);
    srand(1);
    for (int i = 0; i < n; ++i) h[i] = (unsigned)(((unsigned long long)rand() * 65536ull + rand()) % (n4 + 1000));   // some past the end: zero
    h[0] = 0; h[1] = (unsigned)(n4 - 1); h[2] = (unsigned)n4; h[3] = (unsigned)((1ull << 28));     // first, last, first past the end, the 4 GB line
    unsigned* d; f32x4 *a, *b; CK(hipMalloc(&d, n * 4)); CK(hipMalloc(&a, n * 16)); CK(hipMalloc(&b, n * 16));
    CK(hipMemcpy(d, h, n * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_probe, dim3(n / 256), dim3(256), 0, 0, p, (unsigned)n4, d, n, a, b);
    CK(hipDeviceSynchronize());
    f32x4* ha = (f32x4*)malloc(n * 16); f32x4* hb = (f32x4*)malloc(n * 16);
    CK(hipMemcpy(ha, a, n * 16, hipMemcpyDeviceToHost)); CK(hipMemcpy(hb, b, n * 16, hipMemcpyDeviceToHost));
    int bad = 0, beyond4g = 0, past = 0;
    for (int i = 0; i < n; ++i) {
        if (h[i] >= (1u << 28) && h[i] < n4) ++beyond4g;
        if (h[i] >= n4) ++past;
        for (int e = 0; e < 4; ++e) if (ha[i][e] != hb[i][e]) { if (bad < 5) printf("mismatch at %d: record %u: %g vs %g\n", i, h[i], ha[i][e], hb[i][e]); ++bad; break; }
    }
    printf("%.1f GiB, %zu records: %d probes, %d beyond the 4 GB line, %d past the end (must read zero): %d mismatches\n", gib, n4, n, beyond4g, past, bad);
    return bad != 0;
}
