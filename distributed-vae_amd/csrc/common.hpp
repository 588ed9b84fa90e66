// Internal header of libmmvae_hip.so: shapes, workspace layout, and the gfx950 device helpers
// (fp32 MFMA tile products out of LDS, wave reductions, Philox) shared by every kernel file.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include "../../include/mmvae.h"
#include "tune.h"

namespace mmvae {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int WAVE = 64;
constexpr int ROWS32 = 32;   // row block of the small-layer / row-wise kernels
constexpr int NP = 128;      // padded width of every narrow (<=128) dimension in MFMA tiles
constexpr int MAP_PAD = 256;  // valid entries behind the row map's last one (scalar requests run up to two K tiles of 64 ahead)
constexpr int DW11_LD = 132;  // row stride of the dW11 slab: H weights + 1 bias column, H <= 128
constexpr int SMALL_LD = 256; // row stride of a small-layer dW slab: [N<=128][K+1<=256]

__host__ __device__ inline int cdiv(int a, int b) { return (a + b - 1) / b; }
__host__ __device__ inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }
__host__ __device__ inline int rup(int a, int b) { return cdiv(a, b) * b; }
__host__ __device__ inline int64_t imin64(int64_t a, int64_t b) { return a < b ? a : b; }

// ------------------------------------------------------------------------------------------
// Workspace layout (offsets in floats).  Authoritative copy; Python asks through
// mmvae_ws_offset().
// ------------------------------------------------------------------------------------------
struct Splits {
    int ks_fc1;   // split-K of fc1 forward
    int ns_fc11;  // column splits of the fused fc11 kernel
    int ks_dw;    // batch splits of the dW1 GEMM (and of both big dW GEMMs on the general path)
    int ks_dw11;  // batch splits of the dW11 GEMM on the fast path
    int ks_small; // batch splits of the batched small-layer dW GEMM
    int ks_gd10;  // gene splits of the d(d10) = dZ11 W11 GEMM (fast path; the fused general kernel uses ns_fc11)
};

constexpr int N_SMALL = 12;  // fc2 fc3 fc4 fc5 fcc musig fc6 fc7 fc8 fc9 fc10 + fc1.bias

struct Layout {
    int nblk32;   // ceil(B/32)
    int nblk64;   // ceil(B/64)
    Splits sp;
    // forward, saved for backward
    int64_t R[5];                  // R1..R4 [A,B,H], R5 [A,B,L]
    int nblkc;                       // ceil(B / CHAIN_ROWS)
    // cells per workgroup of the FORWARD chain launches (CHAIN_ROWS unless MMVAE_TUNE_CHAIN_ROWS_FWD asks for a smaller block:
    // measured, no gain) and their count
    int chain_rows_fwd, nblkf;
    int nblkl;                       // ceil(B / LAT_ROWS)
    int64_t bn_mean[5], bn_rstd[5];  // [A,W]
    int64_t bn_part[5];            // [A][nblk32][2][W]   (block mean, block M2)
    int64_t XLOW, CPROB, CC, YSOFT, CSMP, Y, MS, MU, LV, SS, ZIN;
    int64_t Dk[5];                 // D6 [A,B,L], D7..D10 [A,B,H]
    int64_t c_part, c_mean, c_iv;  // [A][nblk32][2][C], [A,C], [A,C]
    int64_t lat_part;              // [A][nblk32][2]  (kl sum, entropy sum)
    int64_t fc1_slab;              // [KS][A][B][NP]
    int n11;                       // loss partial slots per arm (zero-filled, a subset is written)
    int64_t fc11_part;             // [A][n11][2] (squared error sum, mismatch count)
    int64_t GD10_slab;             // [max(ns_fc11, ks_gd10)][A][B][H]
    int64_t DZ11;                  // [A][B][D]
    int64_t couple_part;           // [nblk32][2]  (pair distance sum, pair l2 sum)
    int64_t T_part, T;             // [nblk32][A][C], [A][C]
    // backward
    int64_t DZ[11];                // DZ[1..4] [A,B,H], DZ[5] [A,B,L], DZ[6] [A,B,L], DZ[7..10] [A,B,H]
    int64_t GZIN, GMS, GZC, G[6];  // G[5] [A,B,L] (grad wrt x_low); G[1..4] [A,B,H] grad wrt BN_i output
    int64_t bnb_part[6], bnb_sum[6];  // [A][nblk32][2][W], [A][2][W]   (index 1..5)
    int64_t dw1_slab;              // [KS][A][H][D]
    int64_t dw11_slab;             // [KS][A][D][DW11_LD]
    int64_t small_slab;            // [KS][A][N_SMALL][NP*SMALL_LD]
    int64_t xbits;                 // uint32 [A][B][ceil(D/32)] dropout keep-mask, bit-packed (fast path)
    // bf16 slice planes of the SMALL operands of the large GEMMs (fp32x3 engine, gemm_bf16.hip): [A][3][rows][cols] bf16,
    // zero-padded to whole tiles so that the GEMMs copy them into LDS without bounds checks or arithmetic
    int64_t pl_w1;                 // W1   [H -> 128][D -> rup 32]
    int64_t pl_w11;                // [W11 | b11]  [D -> rup 128][H + 1 -> 128]
    int64_t pl_dz1;                // dZ1  [B -> rup 256][H -> 128]
    int64_t pl_d10;                // [d10 | 1]  [B -> rup 256][H + 1 -> 128]
    // weights of the small layers for the fp32x3 form of the chain kernels (chain.hip): [A][PL_SMALL_SLOTS][3][128][128] bf16,
    // slot 0..3 fc2..fc5, 4..8 fc6..fc10 ([N][K] as in the parameters), 9..17 the same layers transposed ([K][N])
    int64_t pl_small;
    // exact batch sums (fixed-point accumulators, see acc_add below): [ACC_NSETS][A] sets of ACC_SET_FLOATS floats;
    // set k < 5: (sum, sum of squares) of BatchNorm k's input, set 5 + (l - 1): (sum G, sum G * xhat) of layer l's
    // BatchNorm backward, l = 1..5.  Sets 0..4 are zeroed by the first kernel of a forward pass, 5..9 by the first
    // kernel of a backward pass.
    int64_t acc;
    int64_t acc_end;               // end of the accumulator sets (the backward pass zeroes [ACC_BWD sets, acc_end))
    int64_t rowmap;                // uint32 [B + MAP_PAD]: element offsets of the batch's rows in the resident matrix (mmvae_train_step_rows)
    int64_t loss_scratch;          // small
    int64_t total;
};
// accumulator set geometry (device side below)
constexpr int PL_SMALL_SLOTS = 18;
constexpr int ACC_W = 128;                             // columns per set
constexpr int ACC_SET_I64 = 6 * ACC_W + 8;             // [6][ACC_W] slots + tail block, [0]: addends outside the window (-> NaN)
constexpr int ACC_SET_FLOATS = 2 * ACC_SET_I64;
constexpr int ACC_NSETS = 12;
// set indices: 0..4 BatchNorm 1..5 forward (sum, sum of squares), ACC_C the statistics of c for the coupling terms (both
// zeroed by the first kernel of a training forward pass), ACC_T the coupling's T sums (sum 1 only; zeroed by the coupling
// launcher, which may run more than once per forward pass), ACC_BWD + l - 1 the BatchNorm backward sums of layer l = 1..5
// (zeroed by the first kernel of a backward pass)
constexpr int ACC_C = 5, ACC_T = 6, ACC_BWD = 7;
inline int64_t acc_set_off(const Layout& L, int A, int set) { return L.acc + (int64_t)set * A * ACC_SET_FLOATS; }

Layout make_layout(const mmvae_dims& d, const mmvae_exec* ex);
Splits default_splits(const mmvae_dims& d, const mmvae_exec* ex);

// Per-arm parameter offsets (floats) -- mirrors mmvae_param_layout_t
struct POff {
    int64_t per_arm;
    int64_t o[MMVAE_N_PARAM_TENSORS];
    int64_t bn_per_arm;
    int64_t bn_mean[MMVAE_N_BN], bn_var[MMVAE_N_BN];
};
POff make_poff(const mmvae_dims& d);

void set_error(const char* fmt, ...);

#ifdef __HIPCC__
// ------------------------------------------------------------------------------------------
// Device helpers
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
__device__ __forceinline__ int wave_id() { return threadIdx.x >> 6; }

// Row of accumulator register r for this lane in a 32x32 MFMA result (col = lane & 31).
__device__ __forceinline__ int acc_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ f32x16 zero16() {
    f32x16 z;
#pragma unroll
    for (int i = 0; i < 16; ++i) z[i] = 0.f;
    return z;
}

// acc[32x32] += A[32 x K] * Bt[32 x K]^T.  Both tiles are LDS images with K contiguous
// (rows a_row0.. of At, rows b_row0.. of Bt).  `kgroups` groups of 8 k's.  Each lane fetches
// 4 consecutive k's with one ds_read_b128 and feeds 4 MFMAs: the k order inside a group is
// permuted identically for A and B, which leaves the dot product unchanged.
__device__ __forceinline__ void mma_nt(f32x16& acc, const float* At, int lda, int a_row0,
                                       const float* Bt, int ldb, int b_row0, int kgroups) {
    const int lane = lane_id();
    const float* pa = At + (a_row0 + (lane & 31)) * lda + 4 * (lane >> 5);
    const float* pb = Bt + (b_row0 + (lane & 31)) * ldb + 4 * (lane >> 5);
    // fragments of group g+1 are requested before the MFMAs of group g: LDS latency hides under them
    float4 a = *reinterpret_cast<const float4*>(pa);
    float4 b = *reinterpret_cast<const float4*>(pb);
    for (int g = 0; g < kgroups; ++g) {
        const int gn = (g + 1 < kgroups) ? g + 1 : g;
        const float4 an = *reinterpret_cast<const float4*>(pa + 8 * gn);
        const float4 bn = *reinterpret_cast<const float4*>(pb + 8 * gn);
        acc = mfma32(a.x, b.x, acc);
        acc = mfma32(a.y, b.y, acc);
        acc = mfma32(a.z, b.z, acc);
        acc = mfma32(a.w, b.w, acc);
        a = an;
        b = bn;
    }
}

// acc[32x32] += A[32 x K] * Bk[K x 32]; A as above (K contiguous), Bk an LDS image [k][n] with n
// contiguous (columns n0..n0+31).
__device__ __forceinline__ void mma_nn(f32x16& acc, const float* At, int lda, int a_row0,
                                       const float* Bk, int ldb, int n0, int kgroups) {
    const int lane = lane_id();
    const float* pa = At + (a_row0 + (lane & 31)) * lda + 4 * (lane >> 5);
    const float* pb = Bk + (4 * (lane >> 5)) * ldb + n0 + (lane & 31);
    float4 a = *reinterpret_cast<const float4*>(pa);
    float b0 = pb[0], b1 = pb[ldb], b2 = pb[2 * ldb], b3 = pb[3 * ldb];
    for (int g = 0; g < kgroups; ++g) {
        const int gn = (g + 1 < kgroups) ? g + 1 : g;
        const float4 an = *reinterpret_cast<const float4*>(pa + 8 * gn);
        const float* q = pb + (8 * gn) * ldb;
        const float n0_ = q[0], n1_ = q[ldb], n2_ = q[2 * ldb], n3_ = q[3 * ldb];
        acc = mfma32(a.x, b0, acc);
        acc = mfma32(a.y, b1, acc);
        acc = mfma32(a.z, b2, acc);
        acc = mfma32(a.w, b3, acc);
        a = an;
        b0 = n0_; b1 = n1_; b2 = n2_; b3 = n3_;
    }
}

// two adjacent 32-column tiles sharing the A fragments (n0 and n0 + 32); second tile optional
__device__ __forceinline__ void mma_nn2(f32x16& acc0, f32x16& acc1, bool second, const float* At, int lda,
                                        int a_row0, const float* Bk, int ldb, int n0, int kgroups) {
    const int lane = lane_id();
    const float* pa = At + (a_row0 + (lane & 31)) * lda + 4 * (lane >> 5);
    const float* pb = Bk + (4 * (lane >> 5)) * ldb + n0 + (lane & 31);
    const int o1 = second ? 32 : 0;
    float4 a = *reinterpret_cast<const float4*>(pa);
    float b0 = pb[0], b1 = pb[ldb], b2 = pb[2 * ldb], b3 = pb[3 * ldb];
    float c0 = pb[o1], c1 = pb[ldb + o1], c2 = pb[2 * ldb + o1], c3 = pb[3 * ldb + o1];
    for (int g = 0; g < kgroups; ++g) {
        const int gn = (g + 1 < kgroups) ? g + 1 : g;
        const float4 an = *reinterpret_cast<const float4*>(pa + 8 * gn);
        const float* q = pb + (8 * gn) * ldb;
        const float n0_ = q[0], n1_ = q[ldb], n2_ = q[2 * ldb], n3_ = q[3 * ldb];
        const float m0_ = q[o1], m1_ = q[ldb + o1], m2_ = q[2 * ldb + o1], m3_ = q[3 * ldb + o1];
        acc0 = mfma32(a.x, b0, acc0);
        if (second) acc1 = mfma32(a.x, c0, acc1);
        acc0 = mfma32(a.y, b1, acc0);
        if (second) acc1 = mfma32(a.y, c1, acc1);
        acc0 = mfma32(a.z, b2, acc0);
        if (second) acc1 = mfma32(a.z, c2, acc1);
        acc0 = mfma32(a.w, b3, acc0);
        if (second) acc1 = mfma32(a.w, c3, acc1);
        a = an;
        b0 = n0_; b1 = n1_; b2 = n2_; b3 = n3_;
        c0 = m0_; c1 = m1_; c2 = m2_; c3 = m3_;
    }
}

// acc[32x32] += Pk[K x 32]^T * Qk[K x 32]; both LDS images [k][.] with the non-k index contiguous.
// ksteps = K/2.
__device__ __forceinline__ void mma_tn(f32x16& acc, const float* Pk, int ldp, int m0,
                                       const float* Qk, int ldq, int n0, int ksteps) {
    const int lane = lane_id();
    const float* pa = Pk + (lane >> 5) * ldp + m0 + (lane & 31);
    const float* pb = Qk + (lane >> 5) * ldq + n0 + (lane & 31);
#pragma unroll 4
    for (int s = 0; s < ksteps; ++s) {
        acc = mfma32(pa[2 * s * ldp], pb[2 * s * ldq], acc);
    }
}

// Branch-free [1 x 4] load of a row-major matrix, zero outside [R, Cn].  Addresses are clamped and the
// result selected, so a tile's loads issue back to back (a per-element `if`, or even a uniform
// vec/scalar `if` around each load, makes hipcc wait vmcnt(0) per load).  VEC: ld % 4 == 0,
// Cn % 4 == 0, col % 4 == 0, base 16-byte aligned -- decide it once per tile, outside the loads.
template <bool VEC>
__device__ __forceinline__ float4 ldg4_t(const float* __restrict__ m, int64_t ld, int row, int col, int R, int Cn) {
    const bool rok = row < R;
    const float* p = m + (int64_t)(rok ? row : 0) * ld;
    float4 v;
    if (VEC) {
        const bool ok = rok && (col < Cn);
        v = *reinterpret_cast<const float4*>(p + (ok ? col : 0));
        v.x = ok ? v.x : 0.f; v.y = ok ? v.y : 0.f; v.z = ok ? v.z : 0.f; v.w = ok ? v.w : 0.f;
        return v;
    }
    const bool o0 = rok && (col < Cn), o1 = rok && (col + 1 < Cn), o2 = rok && (col + 2 < Cn), o3 = rok && (col + 3 < Cn);
    v.x = p[o0 ? col : 0]; v.y = p[o1 ? col + 1 : 0]; v.z = p[o2 ? col + 2 : 0]; v.w = p[o3 ? col + 3 : 0];
    v.x = o0 ? v.x : 0.f; v.y = o1 ? v.y : 0.f; v.z = o2 ? v.z : 0.f; v.w = o3 ? v.w : 0.f;
    return v;
}
struct VecTag { static constexpr bool value = true; };
struct ScalarTag { static constexpr bool value = false; };

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also waits vmcnt(0), i.e. for every
// global store the wave has in flight; after an epilogue that is several microseconds per barrier.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Wave-wide all-reduce without LDS traffic: four DPP steps inside each row of 16 lanes (quad_perm xor 1,
// quad_perm xor 2, row_half_mirror, row_mirror) and the gfx950 half-exchange instructions
// v_permlane16_swap / v_permlane32_swap across rows.  (__shfl_xor lowers to ds_bpermute_b32, ~100 cycles
// of dependent latency per step; the row-wise kernels chain ~15 reductions per cell.)
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
template <typename Op>
__device__ __forceinline__ float wave_allreduce(float v, Op op) {
    v = op(v, dpp_f<0xB1>(v));    // quad_perm [1,0,3,2]
    v = op(v, dpp_f<0x4E>(v));    // quad_perm [2,3,0,1]
    v = op(v, dpp_f<0x141>(v));   // row_half_mirror
    v = op(v, dpp_f<0x140>(v));   // row_mirror: every lane holds its 16-lane row's result
    {
        const unsigned u = __builtin_bit_cast(unsigned, v);
        const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
        v = op(__builtin_bit_cast(float, (unsigned)r[0]), __builtin_bit_cast(float, (unsigned)r[1]));
    }
    {
        const unsigned u = __builtin_bit_cast(unsigned, v);
        const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
        v = op(__builtin_bit_cast(float, (unsigned)r[0]), __builtin_bit_cast(float, (unsigned)r[1]));
    }
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
    return wave_allreduce(v, [](float a, float b) { return a + b; });
}
__device__ __forceinline__ float wave_max(float v) {
    return wave_allreduce(v, [](float a, float b) { return fmaxf(a, b); });
}
__device__ __forceinline__ int wave_min_i(int v) {
    const float f = wave_allreduce(__builtin_bit_cast(float, v), [](float a, float b) {
        const int x = __builtin_bit_cast(int, a), y = __builtin_bit_cast(int, b);
        return __builtin_bit_cast(float, x < y ? x : y);
    });
    return __builtin_bit_cast(int, f);
}

// ---- consumer-side recombination of per-row-block partials ---------------------------------
// Batch statistics cross every row block, so each BatchNorm used to cost a tiny finalize launch
// between two layers.  Instead every workgroup of the CONSUMING kernel recombines the producers'
// partials itself (125 KB out of L2 at B = 5000, W = 100; all workgroups compute bit-identical
// results because the order is fixed by (NT, nblk) alone).
//
// part: [nblk][2][W] = (block mean, block M2) over min(PR, B - PR blk) rows.  All NT threads call;
// thread t < W returns column t's (mean, M2) over the whole batch.  scratch: 3 * PART_MAXG * W floats
// of LDS the caller does not need until the next barrier it executes itself.
constexpr int PART_MAXG = 16;
constexpr int PART_BATCH = 16;
// rows per workgroup of the small-layer chain kernels (chain.hip) = rows per statistics partial they emit
#ifndef MMVAE_CHAIN_ROWS
#define MMVAE_CHAIN_ROWS 64
#endif
constexpr int CHAIN_ROWS = MMVAE_CHAIN_ROWS;
// cells per workgroup of the latent-block kernels (16 waves, one cell per wave at a time).  These kernels are
// VALU-bound, so what counts is cells per CU: 48 gives 105 workgroups per arm at B = 5000 -- one per CU, three cells
// per wave -- where 32 gave 314 workgroups on 256 CUs, i.e. 58 CUs with two (four cells per wave slot).
constexpr int LAT_ROWS = 48;
// The backward kernel runs beside the dW11 GEMM of the side stream, where smaller workgroups spread over all CUs
// measured faster in the step (61 against 70 us) although slower alone (32 against 24 us); with dW11 on 160 CUs
// (round 2, chain kernels on the split engine) 8 cells per workgroup again beat 16 (step 718 against 725 us; 32: 729).
#ifndef MMVAE_LAT_ROWS_BWD
#define MMVAE_LAT_ROWS_BWD 8
#endif
constexpr int LAT_ROWS_BWD = MMVAE_LAT_ROWS_BWD;
constexpr int LATB_NW = 8;    // waves per workgroup of the latent backward kernel
constexpr int LATB_NR = LAT_ROWS_BWD / LATB_NW;   // cells per wave, processed side by side
// The exact batch-sum accumulators (acc_add below) hold ACC_MAX_ADDENDS addends of the largest magnitude per column without a
// carry between their slots; a training batch may therefore have at most ACC_MAX_ADDENDS x (the fewest cells any producing
// workgroup adds at once) cells per rank.  Producers: the fc1 epilogue and the coupling (32-cell blocks), the chain kernels
// (CHAIN_ROWS, or the forward launches' smaller blocks, >= 8), the latent kernels (LAT_ROWS / LAT_ROWS_BWD).
constexpr int ACC_MAX_ADDENDS = 4096;
constexpr int ACC_MIN_PRODUCER_ROWS = LAT_ROWS_BWD < 8 ? LAT_ROWS_BWD : 8;
static_assert(ACC_MIN_PRODUCER_ROWS <= LAT_ROWS_BWD && ACC_MIN_PRODUCER_ROWS <= LAT_ROWS && ACC_MIN_PRODUCER_ROWS <= CHAIN_ROWS &&
              ACC_MIN_PRODUCER_ROWS <= 32 && ACC_MIN_PRODUCER_ROWS <= 8,
              "a producer of batch sums with fewer cells per workgroup lowers the batch-size cap of make_ctx: name it here");

template <bool VEC, int NT>
__device__ __forceinline__ void stats_from_partials_t(const float* __restrict__ part, int nblk, int B, int PR, int W,
                                                      float* scratch, float& mean_out, float& m2_out,
                                                      unsigned long long* st = nullptr) {
    auto tick = [&](int i) {
        if (st) {
            __builtin_amdgcn_sched_barrier(0);
            st[i] = __builtin_amdgcn_s_memtime();
            __builtin_amdgcn_s_waitcnt(0xC07F);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    tick(0);
    constexpr int E = VEC ? 4 : 1;
    const int units = 2 * W / E;   // load units per partial: the first W/E hold means, the rest M2s
    const int G = max(1, min(min(NT / units, PART_MAXG), nblk));
    const int t = threadIdx.x, u = t % units, g = t / units;
    if (g < G) {
        const bool is_mean = u * E < W;
        const int col = is_mean ? u * E : u * E - W;
        const float* p = part + u * E;
        float sh[E], s1[E], s2[E];
        float n = 0.f;
        // the group's first block mean is the shift: s2 - s1^2/n then loses nothing to cancellation.  It is the first
        // value of the first batch (a separate load for it would cost one more memory latency in front of the batch)
#pragma unroll
        for (int e = 0; e < E; ++e) { sh[e] = 0.f; s1[e] = 0.f; s2[e] = 0.f; }
        tick(1);
        // PART_BATCH loads in flight per pass, clamped and weighted instead of branched (a branch around
        // a load makes hipcc wait for every load separately: one memory latency per partial)
        for (int i0 = g; i0 < nblk; i0 += PART_BATCH * G) {
            float v[PART_BATCH][E];
#pragma unroll
            for (int j = 0; j < PART_BATCH; ++j) {
                const int i = min(i0 + j * G, nblk - 1);
                if constexpr (VEC) {
                    const float4 q = *reinterpret_cast<const float4*>(p + (int64_t)i * 2 * W);
                    v[j][0] = q.x; v[j][1] = q.y; v[j][2] = q.z; v[j][3] = q.w;
                } else {
                    v[j][0] = p[(int64_t)i * 2 * W];
                }
            }
            if (i0 == g) {
#pragma unroll
                for (int e = 0; e < E; ++e) sh[e] = is_mean ? v[0][e] : 0.f;
            }
#pragma unroll
            for (int j = 0; j < PART_BATCH; ++j) {
                const int i = i0 + j * G;
                const float nb = i < nblk ? (float)min(PR, B - PR * i) : 0.f;
                const float w = is_mean ? nb : (i < nblk ? 1.f : 0.f);
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    const float d = v[j][e] - sh[e];
                    s1[e] += w * d;
                    s2[e] += nb * d * d;
                }
                n += nb;
            }
        }
#pragma unroll
        for (int e = 0; e < E; ++e) {
            if (is_mean) {
                scratch[(g * 3 + 0) * W + col + e] = sh[e] + s1[e] / n;
                scratch[(g * 3 + 1) * W + col + e] = s2[e] - s1[e] * s1[e] / n;
            } else {
                scratch[(g * 3 + 2) * W + col + e] = s1[e];
            }
        }
    }
    tick(2);
    lds_barrier();
    tick(3);
    float n = 0.f, mean = 0.f, m2 = 0.f;
    if (t < W) {
        for (int k = 0; k < G; ++k) {   // Chan's pairwise update over the groups, fixed order
            const int cnt = (nblk - k + G - 1) / G;
            const float nb = (float)(PR * cnt - (((nblk - 1) % G == k) ? PR * nblk - B : 0));
            const float mg = scratch[(k * 3 + 0) * W + t];
            const float m2g = scratch[(k * 3 + 1) * W + t] + scratch[(k * 3 + 2) * W + t];
            const float nn = n + nb, dl = mg - mean;
            mean += dl * (nb / nn);
            m2 += m2g + dl * dl * (n * nb / nn);
            n = nn;
        }
    }
    mean_out = mean;
    m2_out = m2;
    tick(4);
}
template <int NT>
__device__ __forceinline__ void stats_from_partials(const float* __restrict__ part, int nblk, int B, int PR, int W,
                                                    float* scratch, float& mean_out, float& m2_out,
                                                    unsigned long long* st = nullptr) {
    if ((W & 3) == 0) stats_from_partials_t<true, NT>(part, nblk, B, PR, W, scratch, mean_out, m2_out, st);
    else stats_from_partials_t<false, NT>(part, nblk, B, PR, W, scratch, mean_out, m2_out, st);
}

// part: [nblk][n] plain partial sums -> thread t < n returns sum over blocks of part[.][t].
// scratch: NT * 4 doubles at most (G * n with G = NT / units).
template <bool VEC, int NT>
__device__ __forceinline__ float sums_from_partials_t(const float* __restrict__ part, int nblk, int n, double* scratch) {
    constexpr int E = VEC ? 4 : 1;
    const int units = n / E;
    const int G = max(1, min(min(NT / units, PART_MAXG), nblk));
    const int t = threadIdx.x, u = t % units, g = t / units;
    if (g < G) {
        double s[E];
#pragma unroll
        for (int e = 0; e < E; ++e) s[e] = 0.0;
        const float* p = part + u * E;
        for (int i0 = g; i0 < nblk; i0 += PART_BATCH * G) {
            float v[PART_BATCH][E];
#pragma unroll
            for (int j = 0; j < PART_BATCH; ++j) {
                const int i = min(i0 + j * G, nblk - 1);
                if constexpr (VEC) {
                    const float4 q = *reinterpret_cast<const float4*>(p + (int64_t)i * n);
                    v[j][0] = q.x; v[j][1] = q.y; v[j][2] = q.z; v[j][3] = q.w;
                } else {
                    v[j][0] = p[(int64_t)i * n];
                }
            }
#pragma unroll
            for (int j = 0; j < PART_BATCH; ++j) {
                const bool ok = i0 + j * G < nblk;
#pragma unroll
                for (int e = 0; e < E; ++e) s[e] += ok ? (double)v[j][e] : 0.0;
            }
        }
#pragma unroll
        for (int e = 0; e < E; ++e) scratch[g * n + u * E + e] = s[e];
    }
    lds_barrier();
    double r = 0.0;
    if (t < n)
        for (int k = 0; k < G; ++k) r += scratch[k * n + t];
    return (float)r;
}
template <int NT>
__device__ __forceinline__ float sums_from_partials(const float* __restrict__ part, int nblk, int n, double* scratch) {
    if ((n & 3) == 0) return sums_from_partials_t<true, NT>(part, nblk, n, scratch);
    return sums_from_partials_t<false, NT>(part, nblk, n, scratch);
}

// ReLU that keeps NaN (torch.relu does; fmaxf(NaN, 0) = 0 would turn a diverged hidden layer into zeros and the loss
// back into a finite number)
__device__ __forceinline__ float relu_keep_nan(float v) { return v < 0.f ? 0.f : v; }

// ---- fp32x3 engine (gemm_bf16.hip): the three bf16 slices of two fp32 values (low half: a, high half: b) ------------
__device__ __forceinline__ unsigned cvt_pk_bf16(float a, float b) {   // one v_cvt_pk_bf16_f32 (low half: a), round to nearest even
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ void split3(float a, float b, unsigned (&w)[3]) {
    // eleven VALU instructions per pair: 3 conversions, 4 half -> float, 4 exact subtractions
    w[0] = cvt_pk_bf16(a, b);
    a -= __uint_as_float(w[0] << 16);
    b -= __uint_as_float(w[0] & 0xFFFF0000u);
    w[1] = cvt_pk_bf16(a, b);
    a -= __uint_as_float(w[1] << 16);
    b -= __uint_as_float(w[1] & 0xFFFF0000u);
    w[2] = cvt_pk_bf16(a, b);
}

// ---- exact batch sums through fixed-point atomic accumulators ----------------------------------
// The alternative to the partial arrays above (production; MMVAE_TUNE_BN_PARTIALS switches back): a producing
// workgroup ADDS its block sums to one accumulator per column instead of storing them, so a consumer reads W numbers
// instead of W x (number of producing workgroups).  Floating-point atomics would make the result depend on the arrival
// order; these are integer adds, which commute: a block sum v (a double, |v| < 2^58) is written as the fixed-point number
// round-toward-zero(|v| * 2^84) = h * 2^92 + m * 2^46 + l (0 <= m, l < 2^46, h < 2^50), the three pieces get the sign of v
// and are added to three 64-bit slots with device-scope no-return atomics; 2^12 addends of the largest magnitude and
// either sign fit in the high slot and 2^17 in the others without a carry between the slots (more workgroups than that
// add to one column only for batches beyond 65 000 cells).  Window: block sums below 2^58 = 2.9e17 -- squares of
// activations of 6e7 in every cell of a 64-cell block --, resolution 2^-84 = 5e-26: the variance of a nearly dead unit
// (BatchNorm eps 1e-8 and below) and gradient sums of 1e-20 are still resolved.  A value outside the window (or NaN / Inf)
// counts in the set's flag word and every consumer then returns NaN (the reference's fp32 arithmetic overflows later, at
// squares of 3e38: a documented limit of this build).  The result is the exact sum of the block sums -- bit-identical
// from run to run and for every consumer -- and the consumers form mean and variance from it in fp64.
// Layout of a set: [6][ACC_W] slots (sum 1 high / middle / low, sum 2 high / middle / low; column-minor, so that the lanes
// of one atomic instruction -- one column each -- fall into as few cache lines as possible: the L2 retires an atomic
// request per line, and with one slot per line the same adds took 6 us per launch instead of 1), then the flag word.
__device__ __forceinline__ void acc_add(long long* __restrict__ set, int sum, int col, double v) {
    const double a = fabs(v);
    if (!(a < 0x1p58)) {
        __hip_atomic_fetch_add(set + 6 * ACC_W, 1ll, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    const double t = a * 0x1p84;
    const double h = floor(t * 0x1p-92);
    const double r = t - h * 0x1p92;             // exact: the low bits of t
    const double m = floor(r * 0x1p-46);
    const double l = floor(r - m * 0x1p46);
    long long ih = (long long)h, im = (long long)m, il = (long long)l;
    if (v < 0.0) { ih = -ih; im = -im; il = -il; }
    long long* p = set + (3 * sum) * ACC_W + col;
    if (ih) __hip_atomic_fetch_add(p, ih, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (im) __hip_atomic_fetch_add(p + ACC_W, im, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (il) __hip_atomic_fetch_add(p + 2 * ACC_W, il, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// column `col` of a set: both sums (NaN when the set's flag word is non-zero)
__device__ __forceinline__ void acc_get(const long long* __restrict__ set, int col, double& s1, double& s2) {
    long long q[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) q[k] = set[k * ACC_W + col];
    const long long bad = set[6 * ACC_W];
    s1 = ((double)q[0] * 0x1p92 + (double)q[1] * 0x1p46 + (double)q[2]) * 0x1p-84;
    s2 = ((double)q[3] * 0x1p92 + (double)q[4] * 0x1p46 + (double)q[5]) * 0x1p-84;
    if (bad) { s1 = __builtin_nan(""); s2 = s1; }
}
// (sum, sum of squares) over B rows -> (mean, M2) as stats_from_partials returns them
__device__ __forceinline__ void acc_mean_m2(const long long* __restrict__ set, int col, int B, float& mean, float& m2) {
    double s1, s2;
    acc_get(set, col, s1, s2);
    const double mu = s1 / (double)B;
    mean = (float)mu;
    m2 = (float)fmax(s2 - s1 * mu, 0.0);
    if (s1 != s1) m2 = mean;
}
// a block's (mean, M2) over nb rows -> its contribution to (sum, sum of squares)
__device__ __forceinline__ void acc_add_stats(long long* __restrict__ set, int col, float nb, float mean, float m2) {
    const double n = (double)nb, mu = (double)mean;
    acc_add(set, 0, col, n * mu);
    acc_add(set, 1, col, (double)m2 + n * mu * mu);
}
__device__ __forceinline__ void acc_add_sums(long long* __restrict__ set, int col, float s1, float s2) {
    acc_add(set, 0, col, (double)s1);
    acc_add(set, 1, col, (double)s2);
}
// zero n4 float4 of p, spread over the whole grid (first kernel of a pass: the accumulator sets it is going to fill)
__device__ __forceinline__ void grid_zero(float* __restrict__ p, int n4) {
    const int nt = blockDim.x * blockDim.y * blockDim.z;
    const int64_t nblk = (int64_t)gridDim.x * gridDim.y * gridDim.z;
    const int64_t blk = blockIdx.x + (int64_t)gridDim.x * (blockIdx.y + (int64_t)gridDim.y * blockIdx.z);
    for (int64_t i = blk * nt + threadIdx.x + blockDim.x * (threadIdx.y + blockDim.y * threadIdx.z); i < n4; i += nblk * nt)
        reinterpret_cast<float4*>(p)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
}

// ---- Philox4x32-10 -----------------------------------------------------------------------
struct u32x4 { uint32_t x, y, z, w; };

__host__ __device__ inline u32x4 philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                            uint32_t k0, uint32_t k1) {
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)M0 * c0;
        const uint64_t p1 = (uint64_t)M1 * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        const uint32_t n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        const uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += W0; k1 += W1;
    }
    return u32x4{c0, c1, c2, c3};
}

// Noise streams of the Philox mode.  Counter = (element group, stream id, step); key = seed.
enum { STREAM_XMASK = 0, STREAM_GUMBEL = 1, STREAM_STATE = 2, STREAM_SMASK = 3 };

struct NoiseDev {
    int mode;
    const uint8_t* x_mask;
    const float* u_gumbel;
    const float* u_state;
    const uint8_t* s_mask;
    uint32_t k0, k1;       // seed
    uint32_t step_lo, step_hi;
    uint32_t x_mlog2;      // input dropout: m = 1 << x_mlog2 random bits per element (see xmask_keep)
    uint32_t x_thr;        //   keep iff the element's m-bit field < x_thr, x_thr = round((1-p) 2^16) >> (16 - m) in [0, 2^m]
    uint32_t s_keep_thr;   // state dropout: keep iff u32 < thr (thr = (1-p) * 2^32, saturated)
};

__host__ __device__ inline uint32_t keep_threshold(float p_drop) {
    double k = (1.0 - (double)p_drop) * 4294967296.0;
    if (k >= 4294967295.0) return 0xFFFFFFFFu;
    if (k <= 0.0) return 0u;
    return (uint32_t)k;
}

__host__ __device__ inline uint32_t keep_threshold16(float p_drop) {
    double k = (1.0 - (double)p_drop) * 65536.0 + 0.5;
    if (k >= 65536.0) return 65536u;
    if (k <= 0.0) return 0u;
    return (uint32_t)k;
}

// 4 random words for elements [4*g, 4*g+3] of stream (arm, kind).
__device__ __forceinline__ u32x4 noise_words(const NoiseDev& nz, int arm, int kind, uint64_t group) {
    return philox4x32((uint32_t)group, (uint32_t)(group >> 32), (uint32_t)(arm * 4 + kind) ^ (nz.step_hi << 8),
                      nz.step_lo, nz.k0, nz.k1);
}
__device__ __forceinline__ float u01(uint32_t w) { return (float)(w >> 8) * (1.0f / 16777216.0f); }
__device__ __forceinline__ uint32_t pick(const u32x4& w, int i) {
    return i == 0 ? w.x : (i == 1 ? w.y : (i == 2 ? w.z : w.w));
}
// scalar access helpers (used by the row-wise kernels; the GEMM loaders use the vector form)
__device__ __forceinline__ float noise_uniform(const NoiseDev& nz, int arm, int kind, uint64_t idx) {
    const u32x4 w = noise_words(nz, arm, kind, idx >> 2);
    return u01(pick(w, (int)(idx & 3)));
}
__device__ __forceinline__ bool noise_keep(const NoiseDev& nz, int arm, int kind, uint64_t idx, uint32_t thr) {
    const u32x4 w = noise_words(nz, arm, kind, idx >> 2);
    const uint32_t v = pick(w, (int)(idx & 3));
    return thr == 0xFFFFFFFFu ? true : (v < thr);
}
// Input-dropout keep decision for gene `col` of cell `row`.  The B x D mask is the only noise large enough for the
// generator to cost time (25 M elements per arm per step: 21 us of Philox at 16 bits per element), so an element takes
// only as many random bits as the keep probability needs: m = 1, 2, 4, 8 or 16, the smallest with round((1-p) 2^16)
// a multiple of 2^(16-m) (p = 0.5: one bit per element; resolution of p: 2^-16).  One Philox call serves the 128/m
// consecutive genes [cg * 128/m, ...) of one row: counter (cg, row, stream ^ step_hi, step_lo); element i of the
// group owns bits [i m, (i+1) m) of the 128-bit output (word i m / 32), and is kept iff that field < x_thr.
__device__ __forceinline__ u32x4 xmask_words(const NoiseDev& nz, int arm, uint32_t row, uint32_t cg) {
    return philox4x32(cg, row, (uint32_t)(arm * 4 + STREAM_XMASK) ^ (nz.step_hi << 8), nz.step_lo, nz.k0, nz.k1);
}
__device__ __forceinline__ bool xmask_field_keep(const NoiseDev& nz, const u32x4& w, uint32_t i) {
    const uint32_t bit = i << nz.x_mlog2;
    const uint32_t f = (pick(w, (int)(bit >> 5)) >> (bit & 31u)) & (0xFFFFFFFFu >> (32 - (1 << nz.x_mlog2)));
    return f < nz.x_thr;
}
__device__ __forceinline__ bool xmask_keep(const NoiseDev& nz, int arm, int row, int col) {
    const int epg_log2 = 7 - nz.x_mlog2;
    const u32x4 w = xmask_words(nz, arm, (uint32_t)row, (uint32_t)col >> epg_log2);
    return xmask_field_keep(nz, w, (uint32_t)col & ((1u << epg_log2) - 1u));
}
// The bit-packed dropout keep-mask of x (k_make_xbits, gemm_fast.hip; also a role of the step's prologue launch,
// gemm_bf16.hip): threads first, first + stride, ... of the A * B * ceil(wpr / words-per-thread) work items.
__device__ __forceinline__ void make_xbits_range(const NoiseDev& nz, int A, int B, int D, int wpr, uint32_t* __restrict__ bits,
                                                 int64_t first, int64_t stride) {
    const uint32_t mlog2 = nz.x_mlog2, m = 1u << mlog2;
    const int wpt = nz.mode != 0 && m <= 4 ? (int)(4u >> mlog2) : 1;      // words per thread
    const int tpr = (wpr + wpt - 1) / wpt;                // threads per row
    const int64_t n = (int64_t)A * B * tpr;
    for (int64_t i = first; i < n; i += stride) {
        const int tq = (int)(i % tpr);
        const int64_t ar = i / tpr;
        const int row = (int)(ar % B), arm = (int)(ar / B);
        uint32_t* out = bits + ar * wpr;
        if (nz.mode == 0) {
            const uint8_t* mk = nz.x_mask + ((int64_t)arm * B + row) * D;
            for (int k = 0; k < wpt; ++k) {
                const int w = tq * wpt + k;
                if (w >= wpr) break;
                uint32_t word = 0;
                for (int j = 0; j < 32; ++j) {
                    const int col = 32 * w + j;
                    if (col < D && mk[col]) word |= (1u << j);
                }
                out[w] = word;
            }
            continue;
        }
        if (m <= 4) {
            // one call = genes [tq * 128/m, ...) = words tq * wpt .. + wpt - 1
            const u32x4 r = xmask_words(nz, arm, (uint32_t)row, (uint32_t)tq);
            for (int k = 0; k < wpt; ++k) {
                const int w = tq * wpt + k;
                if (w >= wpr) break;
                uint32_t word = 0;
                if (m == 1) {
                    const uint32_t f = pick(r, k);   // gene j of the word <-> bit j
                    word = nz.x_thr >= 2 ? 0xFFFFFFFFu : (nz.x_thr == 1 ? ~f : 0u);
                } else {
                    for (int j = 0; j < 32; ++j) word |= xmask_field_keep(nz, r, (uint32_t)(32 * k + j)) ? (1u << j) : 0u;
                }
                const int left = D - 32 * w;
                if (left < 32) word &= (1u << left) - 1u;
                out[w] = word;
            }
        } else {
            // 32 genes = m / 4 calls of 128 / m genes
            const int w = tq, ncall = (int)(m >> 2), epc = 32 / ncall;
            uint32_t word = 0;
            for (int j = 0; j < ncall; ++j) {
                const u32x4 r = xmask_words(nz, arm, (uint32_t)row, (uint32_t)(w * ncall + j));
                for (int e = 0; e < epc; ++e) word |= xmask_field_keep(nz, r, (uint32_t)e) ? (1u << (j * epc + e)) : 0u;
            }
            const int left = D - 32 * w;
            if (left < 32) word &= (1u << left) - 1u;
            out[w] = word;
        }
    }
}
#endif  // __HIPCC__

// ------------------------------------------------------------------------------------------
// Launch context handed to the stage launchers (host)
// ------------------------------------------------------------------------------------------
struct Ctx {
    mmvae_dims d;
    mmvae_hyper h;
    Layout lay;
    POff po;
    float* ws;
    hipStream_t stream;
    mmvae_exec ex;          // copy of the caller's execution context (zeros when the caller passed none)
    mmvae_exec* ex_out;     // the caller's own, for `early_recorded` (may be null)
    hipStream_t side() const { return reinterpret_cast<hipStream_t>(ex.side_stream); }
    hipEvent_t ev(int i) const { return reinterpret_cast<hipEvent_t>(ex.ev[i]); }
    int tune(int i) const { return ex.tune[i]; }
    // training-mode batch sums through the fixed-point accumulators (production) or the per-workgroup partial arrays
    bool use_acc() const { return !ex.tune[MMVAE_TUNE_BN_PARTIALS]; }
    // set by the launcher that zeroed [fc11_part, end of the forward accumulator sets) at the start of this call's forward
    // pass; launchers that find it unset (a stage replayed on its own) zero what they need themselves
    mutable bool fwd_zeroed = false;
    // set by launch_x3_planes when this call has written the small-layer weight planes (Layout::pl_small): the chain
    // launchers then take the fp32x3 form of their kernels
    mutable bool small_planes = false;
    // (through the coupling's T set: the fused step's coupling runs inside the decoder chain's launch and finds it zeroed; the
    // coupling's own launcher zeroes it again, it may run more than once per forward pass)
    int64_t fwd_zero_floats() const { return acc_set_off(lay, d.A, ACC_BWD) - lay.fc11_part; }
    int64_t bwd_zero_floats() const { return lay.acc_end - acc_set_off(lay, d.A, ACC_BWD); }
    mutable bool bwd_zeroed = false;   // set by the launcher of the first kernel of a backward pass (it zeroes that range)
    // mmvae_train_step_rows: the batch is rows x_rows[0 .. B) (device, int64) of the resident matrix [x_nrows][x_ld] that the
    // call's `x` points at; the head launch of the step turns them into the row map (Layout::rowmap) and sets rowmap_ready
    const int64_t* x_rows = nullptr;
    // ... and, optionally, its bf16 copy (same shape and leading dimension, in elements): the bf16 engine's large GEMMs and its
    // fused fc11 kernel read x from it and dZ11 travels as bf16 (dz16: the fused kernel of this call wrote it that way)
    const unsigned short* x16 = nullptr;
    mutable bool dz16 = false;
    int64_t x_ld = 0, x_nrows = 0;
    mutable bool rowmap_ready = false;
    // Fork events that ride on a kernel (hipExtLaunchKernel's stop event: the dispatch packet's own completion signal) instead
    // of a hipEventRecord behind it -- a recorded event is a barrier packet of its own, 6 - 7 us of idle main stream at every
    // fork.  The step's driver names the event the NEXT launch of a launcher that knows launch_k carries (stop_ev); the
    // launcher consumes it and sets stop_used, and the fork then only makes the side stream wait.
    mutable hipEvent_t stop_ev = nullptr;
    mutable bool stop_used = false;
    mutable bool couple_in_dec = false;  // the coupling ran as a role of the decoder chain's launch: main stream, nothing to wait for
    mutable bool fork_on_fc11 = false;   // EV_FORK rode on this call's fused fc11 kernel (do_backward's dW11 fork only waits)
};
#ifdef __HIPCC__
template <class K, class... Args>
inline void launch_k(const Ctx& c, K kernel, dim3 grid, dim3 block, unsigned shm, Args... args) {
    if (c.stop_ev) {
        hipExtLaunchKernelGGL(kernel, grid, block, shm, c.stream, nullptr, c.stop_ev, 0, args...);
        c.stop_ev = nullptr;
        c.stop_used = true;
    } else {
        hipLaunchKernelGGL(kernel, grid, block, shm, c.stream, args...);
    }
}
#endif
// events of mmvae_exec.ev by role
enum { EV_LAT = 0, EV_COUPLE, EV_FC11, EV_FORK, EV_JOIN, EV_DEC, EV_ENC, EV_SPARE /* unused */ };

#ifdef __HIPCC__
NoiseDev make_noise_dev(const mmvae_noise* nz, const mmvae_hyper& h);
#endif

// stage launchers (one per kernel family); each returns 0 or MMVAE_E_LAUNCH
int launch_fc1_fwd(const Ctx& c, const mmvae_noise* nz, const float* params, const float* x, int64_t xs);
int launch_bn_eval_stats(const Ctx& c, const float* bn_running);   // eval mode: all five layers, one launch
int launch_chain_fwd_enc(const Ctx& c, int layer /*2..5*/, const float* params, float* bn_running, int64_t* nbt);
int launch_chain_fwd_enc_eval(const Ctx& c, const float* params);   // eval mode: fc2..fc5 in one launch
int launch_lat_fwd(const Ctx& c, const mmvae_noise* nz, const float* params, float* bn_running, int64_t* nbt,
                   int32_t* labels = nullptr /*eval: argmax of c per cell and arm*/);
int launch_chain_fwd_dec(const Ctx& c, const float* params, bool with_couple = false /*the coupling terms as a role of the launch*/);
bool dec_couple_ok(const Ctx& c);
int launch_fc11_fused(const Ctx& c, const float* params, const float* x, int64_t xs, float* x_rec, int need_grad);
int launch_couple(const Ctx& c);
int launch_loss_finalize(const Ctx& c, float* loss_out, int mode = 0 /*1: the T sums only, 2: the scalars only*/);
int launch_chain_bwd_dec(const Ctx& c, const float* params, int nslab);
bool fc11_split_path(const Ctx& c, const float* params, const float* x, int64_t xs);
int launch_lat_bwd(const Ctx& c, const mmvae_noise* nz, const float* params);
int launch_chain_bwd_enc(const Ctx& c, int layer /*5..2*/, const float* params);
int launch_bn_bwd_apply1(const Ctx& c);
int launch_dw_big(const Ctx& c, const mmvae_noise* nz, const float* x, int64_t xs);
int launch_dw_small(const Ctx& c, int which = 3 /*bit0 decoder layers, bit1 encoder side*/);
// one product of the batched small-layer gradient GEMM: out[m][n] = sum_b P[b][m] * Q'[b][n]  (gemm_big.hip, gemm_bf16.hip)
struct TnDesc {
    const float* P; int64_t p_arm_stride; int ldp; int Mv;   // rows of out
    const float* Q; int64_t q_arm_stride; int ldq; int Nv;   // cols of out (before the ones column)
    int q_ones;            // 1: column Nv of Q is the constant 1 (bias gradient)
    int q_xmask;           // 1: Q is x, apply dropout keep-mask (scale applied by the reducer)
    const float* q_mean;   // != null: Q <- (Q - mean[n]) * rstd[n]   (BatchNorm-normalised input)
    const float* q_rstd;   //          arrays are [A][Nv]
    float* out; int64_t out_arm_stride; int64_t out_ks_stride; int ldo;   // out[ks][arm][m][n]
};

struct TnDescs { TnDesc d[N_SMALL]; };
int launch_dw_small_x3(const Ctx& c, const TnDescs& ts, int nsel);
struct AdamHost { float* p; float* m; float* v; int64_t step; float lr, b1, b2, eps, wd; int decoupled; };
// slabs -> grads; with `adam` (p != null) the Adam update is fused into the same pass
int launch_reduce_grads(const Ctx& c, float* grads, float grad_scale, const AdamHost* adam, bool dw11_fast,
                        int which = 3 /*bit0 fc11 tensors, bit1 the rest*/);
int launch_adam(int64_t n, float* p, const float* g, float* m, float* v, int64_t step, float lr, float b1,
                float b2, float eps, float wd, int decoupled, hipStream_t s);
bool fast_path_ok(const Ctx& c, const float* params, const float* x, int64_t xs);
int launch_make_xbits(const Ctx& c, const mmvae_noise* nz);
// first thing of a forward pass: zero the loss partial slots and the forward accumulator sets (folded into
// k_make_xbits when that runs, a fill otherwise)
int launch_forward_zero(const Ctx& c, bool with_xbits, const mmvae_noise* nz);
int launch_fc1_fwd_fast(const Ctx& c, const float* params, const float* x, int64_t xs);
int launch_fc1_epi(const Ctx& c, const float* params);
int launch_fc11_fast(const Ctx& c, const float* params, const float* x, int64_t xs, float* x_rec, int need_grad,
                     int which = 3);
int launch_dw_big_fast(const Ctx& c, const float* x, int64_t xs, int which /*bit0 dW1, bit1 dW11*/);
// bf16-operand variants of the five D x H GEMMs (gemm_bf16.hip; mmvae_hyper.gemm_bf16), same outputs / layouts
// gemm_bf16 == 2: fp32 operands split exactly into three bf16 slices each (six slice products per product: fp32-grade
// results on the bf16 matrix pipe); the same tile engine with three LDS planes per operand
// bits 8.. of gemm_bf16 (diagnostics): products that stay on the fp32 matrix instruction although gemm_bf16 & 0xFF == 2
// (1 fc1, 2 fc11 + d(d10), 4 dW1, 8 dW11)
inline bool split3_gemms(const Ctx& c, int op = 0) { return (c.h.gemm_bf16 & 0xFF) == 2 && c.d.H <= 124 && !((c.h.gemm_bf16 >> 8) & op); }
inline bool bf16_gemms(const Ctx& c, int op = 0) { return ((c.h.gemm_bf16 & 0xFF) == 1 || split3_gemms(c, op)) && c.d.H <= 124; }
int launch_x3_planes(const Ctx& c, const float* params, int which /*bit0 W1 + [W11|b11] + small layers, bit1 [d10|1], bit2 dZ1, bit3 small layers only, bit4 + keep-mask and zero fill (head of a training step)*/,
                     const mmvae_noise* nz = nullptr);
// true when launch_x3_planes(.., 1 | 16, nz) takes over k_make_xbits' work (the engines that have a k_presplit launch at the head of the step)
inline bool prologue_merged(const Ctx& c);
// the kernels that produce dZ1 / d10 write their slice planes themselves (no k_presplit launch for them)
// the chain kernels' own GEMMs on the fp32x3 engine: every layer within one 128 x 128 plane
// (also in the bf16 configuration: only its five D x H products round their operands, everything else stays fp32-grade)
inline bool chain_x3_ok(const Ctx& c) {
    return bf16_gemms(c) && c.d.C + c.d.S <= 128 && c.d.L <= 128 && !c.tune(MMVAE_TUNE_CHAIN_FP32);
}
inline bool prologue_merged(const Ctx& c) {
    return c.h.training && c.h.x_drop > 0.f && (split3_gemms(c) || chain_x3_ok(c)) && !c.tune(MMVAE_TUNE_PRESPLIT_ALL);
}
inline bool dec_chain_writes_planes(const Ctx& c) { return split3_gemms(c) && !c.tune(MMVAE_TUNE_PRESPLIT_ALL); }
// bf16 configuration on bf16 storage (mmvae_train_step_rows(data_bf16)): the NARROW operands of fc1 / dW1 (W1, dZ1) are read as
// bf16 too -- slice 0 of the planes the fp32x3 engine uses -- instead of fp32 rounded by every block tile
inline bool bf16_narrow_planes(const Ctx& c) {
    return (c.h.gemm_bf16 & 0xFF) == 1 && c.x16 != nullptr && c.d.H <= 124 && (c.d.D & 7) == 0 && !c.tune(MMVAE_TUNE_PRESPLIT_ALL) &&
           !c.tune(MMVAE_TUNE_BF16_NARROW_FP32);
}
inline bool bn_apply_writes_planes(const Ctx& c) {
    return (split3_gemms(c, 4) || bf16_narrow_planes(c)) && (c.d.H & 1) == 0 && !c.tune(MMVAE_TUNE_PRESPLIT_ALL);
}
int launch_fc1_fwd_bf16(const Ctx& c, const float* params, const float* x, int64_t xs);
int launch_fc11_bf16(const Ctx& c, const float* params, const float* x, int64_t xs, float* x_rec, int need_grad, int which);
int launch_dw_big_bf16(const Ctx& c, const float* x, int64_t xs, int which);
int launch_bf16_affine(hipStream_t s, bool relu, bool affine, const float* A, int lda, int M, const float* W, int ldw, int N,
                       int Kpad, const float* sc, const float* sh, float* C, int ldc, int ncols, int split3 = 0,
                       const unsigned short* w_planes = nullptr, int Np = 0, int Kp = 0,
                       float* scratch = nullptr, int64_t scratch_floats = 0 /*room for split-K slabs of layers with few tiles and a long K*/);
// fp32 [R][C] (row pitch ld) -> three bf16 slice planes [3][Rp][Cp], zero-padded (fp32x3 engine)
int launch_presplit_one(hipStream_t s, const float* src, int64_t ld, int R, int C, int Rp, int Cp, unsigned short* dst);
// planes x planes GEMM (gemm_pp.hip): tiled slice planes of a matrix X [R][K] -- NP planes (3 exact slices / 1 rounded), each
// [KT = ceil(K / 16)][Rp = R rounded up to 256][16] bf16, zero for k >= K
struct TPlanes { unsigned short* p; int64_t plane; int Rp, KT; };
inline int tp_rp(int R) { return rup(R, 256); }
inline int tp_kt(int K) { return cdiv(K, 16); }
inline int64_t tp_plane_elems(int R, int K) { return (int64_t)tp_kt(K) * tp_rp(R) * 16; }
inline TPlanes tp_make(unsigned short* p, int R, int K) { return TPlanes{p, tp_plane_elems(R, K), tp_rp(R), tp_kt(K)}; }
int launch_tp_from_f32(hipStream_t s, const float* src, int64_t ld, int R, int K, int NP, TPlanes dst, float* zero_flags_of_scratch = nullptr);
int launch_rp_from_f32(hipStream_t s, const float* src, int64_t ld, int R, int K, int NP, TPlanes dst);   // row-major planes [NP][Rp][KT][16]
constexpr int PP_MAX_WG = 256;                           // workgroups of a K-split k_pp_gemm launch at most (partial slots)
constexpr int PP_MAX_KS = 8;
constexpr int PP_FLAG_SLOTS = 16;                        // launches between two launch_pp_zero_flags
constexpr int PP_FLAG_WORDS = PP_FLAG_SLOTS * PP_MAX_WG;
int64_t pp_scratch_floats();
int launch_pp_zero_flags(hipStream_t s, float* scratch);
int launch_pp_gemm(hipStream_t s, int NP, TPlanes a, TPlanes b, int M, int N, const float* scale, const float* shift, bool affine, bool relu,
                   float* out32, int64_t ld32, int ncols32, const TPlanes* outp, float* scratch, int64_t scratch_floats, int flag_slot, int force = 0,
                   const unsigned* a_map = nullptr, int a_map_rows = 0);
int launch_pp_rowmap(hipStream_t s, const int64_t* rows, int n, int64_t n_rows, unsigned* out, int n_pad, float* zero_flags_of_scratch = nullptr);
// evaluation labels / consensus (consensus.hip)
int launch_classify(const float* cc, int64_t n_cells, int C, int32_t* labels, hipStream_t s);
int launch_confmat(const int32_t* labels, int A, int64_t n, int C, int64_t* counts, hipStream_t s);
int launch_consensus(const int64_t* counts, int npairs, int C, double* cm_norm, double* consensus, hipStream_t s);
int launch_dump_noise(const mmvae_dims& d, const mmvae_hyper& h, const mmvae_noise* nz, uint8_t* x_mask,
                      float* u_gumbel, float* u_state, uint8_t* s_mask, hipStream_t s);

}  // namespace mmvae
