// Micro-benchmark (round 4): the planes x planes GEMM engine (csrc/gemm_pp.hip) on the augmenter's layer shapes (default: the
// launcher's choice of tile / K split beside forced alternatives), on a shape from the command line (M N K force [planes]), or
// ("fc1") on the shapes of the train step's fc1 product -- what the step's x3 GEMMs could gain on pre-split planes of the
// resident matrix (no dropout mask, no row map here).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../distributed-vae_amd/csrc -I../../include -o pp_shape_bench pp_shape_bench.hip && ./pp_shape_bench
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "../../distributed-vae_amd/csrc/gemm_pp.hip"
namespace mmvae { void set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fprintf(stderr, "\n"); } }
using namespace mmvae;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
static int run(int M, int N, int K, int force, const char* what, int NP = 3) {
    unsigned short *a, *b; float *scratch, *out;
    const int64_t ae = NP * tp_plane_elems(M, K), be = NP * tp_plane_elems(N, K);
    CK(hipMalloc(&a, ae * 2)); CK(hipMalloc(&b, be * 2));
    const int64_t scr = pp_scratch_floats();
    CK(hipMalloc(&scratch, scr * 4)); CK(hipMalloc(&out, (int64_t)M * N * 4));
    CK(hipMemset(a, 0x3c, ae * 2)); CK(hipMemset(b, 0x3c, be * 2));     // bf16 0x3c3c = 0.0115: finite, non-trivial bit patterns
    TPlanes A = tp_make(a, M, K), Bp = tp_make(b, N, K);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto once = [&]() { return launch_pp_zero_flags(0, scratch) || launch_pp_gemm(0, NP, A, Bp, M, N, nullptr, nullptr, false, false, out, N, N, nullptr, scratch, scr, 0, force); };
    for (int i = 0; i < 3; ++i) if (once()) return 1;
    CK(hipEventRecord(e0, 0));
    const int reps = 20;
    for (int i = 0; i < reps; ++i) once();
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-46s NP=%d M=%5d N=%4d K=%5d force=%3d  %7.1f us  (%.0f TFLOP/s fp32-equivalent on the real extents)\n", what, NP, M, N, K, force, ms * 1e3 / reps,
           2.0 * M * N * K / (ms * 1e-3 / reps) / 1e12);
    hipFree(a); hipFree(b); hipFree(scratch); hipFree(out);
    return 0;
}
int main(int argc, char** argv) {
    if (argc >= 5) return run(atoi(argv[1]), atoi(argv[2]), atoi(argv[3]), atoi(argv[4]), "command line", argc > 5 ? atoi(argv[5]) : 3);
    if (argc == 2 && !strcmp(argv[1], "fc1")) {
        // fc1 of two arms on one x tile: N = 2 x 128; K split so that the grid fills the chip (20 row tiles)
        run(5000, 256, 5000, 41, "fc1, both arms in one 256 x 256 tile, KS=4");
        run(5000, 256, 5000, 0, "fc1, both arms, automatic");
        run(5000, 128, 5000, 42, "fc1, one arm, 256 x 128 tile, KS=4");
        run(10000, 128, 5000, 42, "fc1, arms stacked along M, 256 x 128, KS=4");
        run(10000, 128, 5000, 32, "fc1, arms stacked along M, 256 x 128, KS=3");
        run(5000, 256, 5000, 1, "fc1 256 x 256 KS=1 (20 blocks: per-block rate)");
        return 0;
    }
    // the augmenter's layers at the benchmark shape (A = 2, B = 5000, D = 5000, n_dim 500; the time includes the flag memset):
    // the launcher's choice, then forced alternatives (force = tile + 10 KS; tiles 1 .. 4 = 256 x 256, 256 x 128, 128 x 128, 160 x 256)
    struct L { const char* name; int M, N, K; int forces[6]; };
    const L layers[] = {
        {"aug fc1", 5000, 1000, 5000, {0, 31, 24, 34, 4, 22}},   {"aug fc2", 5000, 1000, 1000, {0, 2, 24, 21, 1, 4}},
        {"aug fc3", 5000, 500, 1000, {0, 32, 44, 41, 22, 24}},   {"aug fc4", 5000, 500, 500, {0, 3, 44, 24, 41, 22}},
        {"aug fc5", 10000, 100, 500, {0, 43, 42, 22, 2, 3}},     {"aug fc7", 10000, 500, 100, {0, 2, 4, 1, 3, 24}},
        {"aug fc8", 10000, 500, 500, {0, 2, 24, 4, 21, 22}},     {"aug fc9", 10000, 1000, 500, {0, 1, 4, 2, 24, 21}},
        {"aug fc10", 10000, 1000, 1000, {0, 1, 4, 2, 24, 21}},   {"aug fc11", 10000, 5000, 1000, {0, 1, 4, 2, 3, 1}},
    };
    for (const L& l : layers)
        for (int f = 0; f < 6; ++f)
            if (f == 0 || l.forces[f] != l.forces[f - 1]) run(l.M, l.N, l.K, l.forces[f], l.name);
    return 0;
}
