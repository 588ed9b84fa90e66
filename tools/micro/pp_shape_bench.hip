// Micro-benchmark (round 4): the planes x planes GEMM engine (csrc/gemm_pp.hip) on the shapes of the train step's fc1 product --
// what the step's x3 GEMMs could gain on pre-split planes of the resident matrix (no dropout mask, no row map here).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../distributed-vae_amd/csrc -I../../include -o pp_shape_bench pp_shape_bench.hip && ./pp_shape_bench
#include <stdarg.h>
#include <stdio.h>
#include "../../distributed-vae_amd/csrc/gemm_pp.hip"
namespace mmvae { void set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fprintf(stderr, "\n"); } }
using namespace mmvae;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
static int run(int M, int N, int K, int force, const char* what) {
    const int NP = 3;
    unsigned short *a, *b; float *slab, *out;
    const int64_t ae = NP * tp_plane_elems(M, K), be = NP * tp_plane_elems(N, K);
    CK(hipMalloc(&a, ae * 2)); CK(hipMalloc(&b, be * 2));
    const int64_t scr = (int64_t)16 * M * rup(N, 2);
    CK(hipMalloc(&slab, scr * 4)); CK(hipMalloc(&out, (int64_t)M * N * 4));
    CK(hipMemset(a, 0x3c, ae * 2)); CK(hipMemset(b, 0x3c, be * 2));     // bf16 0x3c3c = 0.0115: finite, non-trivial bit patterns
    TPlanes A = tp_make(a, M, K), Bp = tp_make(b, N, K);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) if (launch_pp_gemm(0, NP, A, Bp, M, N, nullptr, nullptr, false, false, out, N, N, nullptr, slab, scr, force)) return 1;
    CK(hipEventRecord(e0, 0));
    const int reps = 20;
    for (int i = 0; i < reps; ++i) launch_pp_gemm(0, NP, A, Bp, M, N, nullptr, nullptr, false, false, out, N, N, nullptr, slab, scr, force);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-46s M=%5d N=%4d K=%5d force=%3d  %7.1f us  (%.0f TFLOP/s fp32-equivalent on the real extents)\n", what, M, N, K, force, ms * 1e3 / reps,
           2.0 * M * N * K / (ms * 1e-3 / reps) / 1e12);
    hipFree(a); hipFree(b); hipFree(slab); hipFree(out);
    return 0;
}
int main() {
    // fc1 of two arms on one x tile: N = 2 x 128; K split so that the grid fills the chip (20 row tiles)
    run(5000, 256, 5000, 41, "fc1, both arms in one 256 x 256 tile, KS=4");
    run(5000, 256, 5000, 0, "fc1, both arms, automatic");
    run(5000, 128, 5000, 42, "fc1, one arm, 256 x 128 tile, KS=4");
    run(10000, 128, 5000, 42, "fc1, arms stacked along M, 256 x 128, KS=4");
    run(10000, 128, 5000, 32, "fc1, arms stacked along M, 256 x 128, KS=3");
    run(5000, 256, 5000, 1, "fc1 256 x 256 KS=1 (20 blocks: per-block rate)");
    return 0;
}
