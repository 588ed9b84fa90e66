// Fast paths of the three MFMA-bound stages (selected when D % 4 == 0 and H % 4 == 0 and the buffers
// are 16-byte aligned; gemm_big.hip keeps the fully general kernels).  What differs from the general
// kernels is the instruction stream around the MFMAs, not the arithmetic:
//   * every global load is unconditional (clamped address + select), so hipcc issues the whole
//     tile's loads back to back and waits once -- the general loaders branch per element and wait
//     vmcnt(0) per load, which serialises HBM latency;
//   * the dropout keep-mask is read from a bit-packed image built once per step (k_make_xbits)
//     instead of a Philox evaluation per float4 in both fc1 forward and dW1;
//   * 64x64 (fc1, dW) wave tiles: 4 accumulators per wave, 1 ds_read_b128 per 4 MFMAs.
#include "common.hpp"
#include <stdlib.h>

namespace mmvae {

#define HIP_LAUNCH_CHECK(what)                                                        \
    do {                                                                              \
        hipError_t e_ = hipGetLastError();                                            \
        if (e_ != hipSuccess) {                                                       \
            set_error("%s: %s", what, hipGetErrorString(e_));                         \
            return MMVAE_E_LAUNCH;                                                    \
        }                                                                             \
    } while (0)

__device__ __forceinline__ float4 zero4() { return make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ float4 sel4(bool ok, float4 v) { return ok ? v : zero4(); }
__device__ __forceinline__ float4 mask4(float4 v, uint32_t nib) {
    v.x = (nib & 1u) ? v.x : 0.f;
    v.y = (nib & 2u) ? v.y : 0.f;
    v.z = (nib & 4u) ? v.z : 0.f;
    v.w = (nib & 8u) ? v.w : 0.f;
    return v;
}

// ---------------------------------------------------------------------------------------------
// keep-mask bit image: bits[(arm*B + row) * wpr + w] bit i <-> gene 32 w + i (zero beyond D).  Same element ->
// random-bits mapping as the general kernels and mmvae_dump_noise (xmask_keep).  One thread per Philox call when a
// call covers whole words (m <= 4 bits per element: 4 / m words), else one thread per word (m / 4 calls).
// ---------------------------------------------------------------------------------------------
__global__ void k_make_xbits(NoiseDev nz, int A, int B, int D, int wpr, uint32_t* __restrict__ bits,
                             float* __restrict__ zero_p, int zero_n4) {
    grid_zero(zero_p, zero_n4);   // the step's loss partial slots and forward accumulator sets (launch_forward_zero)
    make_xbits_range(nz, A, B, D, wpr, bits, (int64_t)blockIdx.x * blockDim.x + threadIdx.x, (int64_t)gridDim.x * blockDim.x);
}

// =============================================================================================
// fc1 forward, fast: block tile 128 x 128, K tile 32, wave tile 64 x 64.  grid (ceil(B/128), KS, A)
// =============================================================================================
constexpr int V2_LD = 36;

template <bool USE_MASK>
__global__ __launch_bounds__(256) void k_fc1_fwd_v2(const float* __restrict__ x, int64_t x_arm_stride,
                                                    const float* __restrict__ params, int64_t per_arm, int64_t w_off,
                                                    const uint32_t* __restrict__ bits, int wpr,
                                                    float* __restrict__ slab, int A, int B, int D, int H, int KS,
                                                    int ablate) {
    __shared__ __attribute__((aligned(16))) float As[128 * V2_LD];
    __shared__ __attribute__((aligned(16))) float Bs[128 * V2_LD];
    const int arm = blockIdx.z, ks = blockIdx.y, b0 = blockIdx.x * 128;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, wm = wv >> 1, wn = wv & 1;
    const int l31 = lane & 31, hh = lane >> 5;
    const float* xa = x + (int64_t)arm * x_arm_stride;
    const float* W = params + (int64_t)arm * per_arm + w_off;
    const int nkt = cdiv(D, 32);
    const int kt0 = (int)(((int64_t)ks * nkt) / KS), kt1 = (int)(((int64_t)(ks + 1) * nkt) / KS);
    const int r0 = tid >> 3, c4 = tid & 7;

    const float* pa[4];
    const float* pb[4];
    const uint32_t* pm[4];
    bool okb[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int ra = min(b0 + r0 + 32 * i, B - 1);      // rows past B recompute row B-1; never stored
        pa[i] = xa + (int64_t)ra * D + c4 * 4;
        pm[i] = bits + ((int64_t)arm * B + ra) * wpr;
        const int rb = r0 + 32 * i;
        okb[i] = rb < H;
        pb[i] = W + (int64_t)min(rb, H - 1) * D + c4 * 4;
    }
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = zero16();

    float4 ra4[4], rb4[4];
    auto load_tiles = [&](int kt) {
        const bool colok = kt * 32 + c4 * 4 < D;
        const int koff = colok ? kt * 32 : 0;
        uint32_t wd[4];
        if (USE_MASK) {
#pragma unroll
            for (int i = 0; i < 4; ++i) wd[i] = pm[i][kt];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) ra4[i] = *reinterpret_cast<const float4*>(pa[i] + koff);
#pragma unroll
        for (int i = 0; i < 4; ++i) rb4[i] = *reinterpret_cast<const float4*>(pb[i] + koff);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float4 v = sel4(colok, ra4[i]);
            if (USE_MASK) v = mask4(v, wd[i] >> (c4 * 4));
            ra4[i] = v;
            rb4[i] = sel4(colok && okb[i], rb4[i]);
        }
    };
    if (kt0 < kt1) load_tiles(kt0);
    for (int kt = kt0; kt < kt1; ++kt) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            *reinterpret_cast<float4*>(&As[(r0 + 32 * i) * V2_LD + c4 * 4]) = ra4[i];
            *reinterpret_cast<float4*>(&Bs[(r0 + 32 * i) * V2_LD + c4 * 4]) = rb4[i];
        }
        __syncthreads();
        if (kt + 1 < kt1 && !(ablate & 2)) load_tiles(kt + 1);
        const float* la = As + (wm * 64 + l31) * V2_LD + 4 * hh;
        const float* lb = Bs + (wn * 64 + l31) * V2_LD + 4 * hh;
        if (!(ablate & 1))
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 a0 = *reinterpret_cast<const float4*>(la + 8 * g);
            const float4 a1 = *reinterpret_cast<const float4*>(la + 32 * V2_LD + 8 * g);
            const float4 q0 = *reinterpret_cast<const float4*>(lb + 8 * g);
            const float4 q1 = *reinterpret_cast<const float4*>(lb + 32 * V2_LD + 8 * g);
            acc[0][0] = mfma32(a0.x, q0.x, acc[0][0]); acc[0][1] = mfma32(a0.x, q1.x, acc[0][1]);
            acc[1][0] = mfma32(a1.x, q0.x, acc[1][0]); acc[1][1] = mfma32(a1.x, q1.x, acc[1][1]);
            acc[0][0] = mfma32(a0.y, q0.y, acc[0][0]); acc[0][1] = mfma32(a0.y, q1.y, acc[0][1]);
            acc[1][0] = mfma32(a1.y, q0.y, acc[1][0]); acc[1][1] = mfma32(a1.y, q1.y, acc[1][1]);
            acc[0][0] = mfma32(a0.z, q0.z, acc[0][0]); acc[0][1] = mfma32(a0.z, q1.z, acc[0][1]);
            acc[1][0] = mfma32(a1.z, q0.z, acc[1][0]); acc[1][1] = mfma32(a1.z, q1.z, acc[1][1]);
            acc[0][0] = mfma32(a0.w, q0.w, acc[0][0]); acc[0][1] = mfma32(a0.w, q1.w, acc[0][1]);
            acc[1][0] = mfma32(a1.w, q0.w, acc[1][0]); acc[1][1] = mfma32(a1.w, q1.w, acc[1][1]);
        }
        __syncthreads();
    }
    float* out = slab + (((int64_t)ks * A + arm) * B) * NP;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = b0 + wm * 64 + i * 32 + acc_row(r, lane);
                if (row < B) out[(int64_t)row * NP + wn * 64 + j * 32 + l31] = acc[i][j][r];
            }
}

// =============================================================================================
// fc1 forward for fc_dim = 100: the same tiles as k_fc1_fwd_v2, but a wave owns 32 rows x (3 MFMA column tiles
// + 4 leftover columns) instead of 64 x 64 of a 128-wide tile whose last 28 columns are padding.  The leftover
// columns 96..99 are plain FMAs on the A fragments the wave already holds (16 per 12 MFMAs): a quarter fewer
// MFMAs for the same result.  Columns >= 100 of the slab rows are not written (the epilogue never uses them).
// =============================================================================================
template <bool USE_MASK>
__global__ __launch_bounds__(256, 2) void k_fc1_fwd_v3(const float* __restrict__ x, int64_t x_arm_stride,
                                                    const float* __restrict__ params, int64_t per_arm, int64_t w_off,
                                                    const uint32_t* __restrict__ bits, int wpr,
                                                    float* __restrict__ slab, int A, int B, int D, int H, int KS,
                                                    int ablate) {
    __shared__ __attribute__((aligned(16))) float As[128 * V2_LD];
    __shared__ __attribute__((aligned(16))) float Bs[128 * V2_LD];
    const int arm = blockIdx.z, ks = blockIdx.y, b0 = blockIdx.x * 128;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int l31 = lane & 31, hh = lane >> 5;
    const float* xa = x + (int64_t)arm * x_arm_stride;
    const float* W = params + (int64_t)arm * per_arm + w_off;
    const int nkt = cdiv(D, 32);
    const int kt0 = (int)(((int64_t)ks * nkt) / KS), kt1 = (int)(((int64_t)(ks + 1) * nkt) / KS);
    const int r0 = tid >> 3, c4 = tid & 7;

    const float* pa[4];
    const float* pb[4];
    const uint32_t* pm[4];
    bool okb[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int ra = min(b0 + r0 + 32 * i, B - 1);      // rows past B recompute row B-1; never stored
        pa[i] = xa + (int64_t)ra * D + c4 * 4;
        pm[i] = bits + ((int64_t)arm * B + ra) * wpr;
        const int rb = r0 + 32 * i;
        okb[i] = rb < H;
        pb[i] = W + (int64_t)min(rb, H - 1) * D + c4 * 4;
    }
    f32x16 acc[3] = {zero16(), zero16(), zero16()};   // columns [32 j, 32 j + 32), rows [32 wv, 32 wv + 32)
    float lo[4] = {0.f, 0.f, 0.f, 0.f};               // columns 96..99 of row (lane & 31): this lane's k's only

    float4 ra4[4], rb4[4];
    auto load_tiles = [&](int kt) {
        const bool colok = kt * 32 + c4 * 4 < D;
        const int koff = colok ? kt * 32 : 0;
        uint32_t wd[4];
        if (USE_MASK) {
#pragma unroll
            for (int i = 0; i < 4; ++i) wd[i] = pm[i][kt];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) ra4[i] = *reinterpret_cast<const float4*>(pa[i] + koff);
#pragma unroll
        for (int i = 0; i < 4; ++i) rb4[i] = *reinterpret_cast<const float4*>(pb[i] + koff);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float4 v = sel4(colok, ra4[i]);
            if (USE_MASK) v = mask4(v, wd[i] >> (c4 * 4));
            ra4[i] = v;
            rb4[i] = sel4(colok && okb[i], rb4[i]);
        }
    };
    if (kt0 < kt1) load_tiles(kt0);
    for (int kt = kt0; kt < kt1; ++kt) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            *reinterpret_cast<float4*>(&As[(r0 + 32 * i) * V2_LD + c4 * 4]) = ra4[i];
            *reinterpret_cast<float4*>(&Bs[(r0 + 32 * i) * V2_LD + c4 * 4]) = rb4[i];
        }
        __syncthreads();
        if (kt + 1 < kt1 && !(ablate & 2)) load_tiles(kt + 1);
        const float* la = As + (wv * 32 + l31) * V2_LD + 4 * hh;
        const float* lb = Bs + l31 * V2_LD + 4 * hh;
        const float* ll = Bs + 96 * V2_LD + 4 * hh;      // rows 96..99 of W1: the same address for a whole half wave
        if (!(ablate & 1))
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 a = *reinterpret_cast<const float4*>(la + 8 * g);
            const float4 q0 = *reinterpret_cast<const float4*>(lb + 8 * g);
            const float4 q1 = *reinterpret_cast<const float4*>(lb + 32 * V2_LD + 8 * g);
            const float4 q2 = *reinterpret_cast<const float4*>(lb + 64 * V2_LD + 8 * g);
            const float4 w0 = *reinterpret_cast<const float4*>(ll + 8 * g);
            const float4 w1 = *reinterpret_cast<const float4*>(ll + V2_LD + 8 * g);
            const float4 w2 = *reinterpret_cast<const float4*>(ll + 2 * V2_LD + 8 * g);
            const float4 w3 = *reinterpret_cast<const float4*>(ll + 3 * V2_LD + 8 * g);
            acc[0] = mfma32(a.x, q0.x, acc[0]); acc[1] = mfma32(a.x, q1.x, acc[1]); acc[2] = mfma32(a.x, q2.x, acc[2]);
            acc[0] = mfma32(a.y, q0.y, acc[0]); acc[1] = mfma32(a.y, q1.y, acc[1]); acc[2] = mfma32(a.y, q2.y, acc[2]);
            acc[0] = mfma32(a.z, q0.z, acc[0]); acc[1] = mfma32(a.z, q1.z, acc[1]); acc[2] = mfma32(a.z, q2.z, acc[2]);
            acc[0] = mfma32(a.w, q0.w, acc[0]); acc[1] = mfma32(a.w, q1.w, acc[1]); acc[2] = mfma32(a.w, q2.w, acc[2]);
            // the four columns that do not fill a 32-wide MFMA tile: 16 VALU FMAs in the MFMAs' shadow
            lo[0] = fmaf(a.x, w0.x, lo[0]); lo[1] = fmaf(a.x, w1.x, lo[1]); lo[2] = fmaf(a.x, w2.x, lo[2]); lo[3] = fmaf(a.x, w3.x, lo[3]);
            lo[0] = fmaf(a.y, w0.y, lo[0]); lo[1] = fmaf(a.y, w1.y, lo[1]); lo[2] = fmaf(a.y, w2.y, lo[2]); lo[3] = fmaf(a.y, w3.y, lo[3]);
            lo[0] = fmaf(a.z, w0.z, lo[0]); lo[1] = fmaf(a.z, w1.z, lo[1]); lo[2] = fmaf(a.z, w2.z, lo[2]); lo[3] = fmaf(a.z, w3.z, lo[3]);
            lo[0] = fmaf(a.w, w0.w, lo[0]); lo[1] = fmaf(a.w, w1.w, lo[1]); lo[2] = fmaf(a.w, w2.w, lo[2]); lo[3] = fmaf(a.w, w3.w, lo[3]);
            __builtin_amdgcn_sched_barrier(0);   // keep the fragment registers of one k group at a time
        }
        __syncthreads();
    }
    float* out = slab + (((int64_t)ks * A + arm) * B) * NP;
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = b0 + wv * 32 + acc_row(r, lane);
            if (row < B) out[(int64_t)row * NP + j * 32 + l31] = acc[j][r];
        }
    // lanes l and l ^ 32 hold the two k halves of the same row
#pragma unroll
    for (int c = 0; c < 4; ++c) lo[c] += __shfl_xor(lo[c], 32, 64);
    {
        const int row = b0 + wv * 32 + l31;
        if (hh == 0 && row < B) *reinterpret_cast<float4*>(out + (int64_t)row * NP + 96) = make_float4(lo[0], lo[1], lo[2], lo[3]);
    }
}

// =============================================================================================
// TN over the batch, fast: out[m][n] = sum_b P[b][m] Q[b][n]; tile 128 x 128, 32 batch rows per step,
// wave tile 64 x 64.  grid (tiles_m * tiles_n, KS, A).  Q may carry the keep-mask (x) or a trailing
// ones column (bias gradient).
// =============================================================================================
constexpr int TN_LD = 132;

template <bool QMASK, bool QONES>
__global__ __launch_bounds__(256) void k_tn_v2(const float* __restrict__ P, int64_t p_arm, int ldp, int Mv,
                                               const float* __restrict__ Q, int64_t q_arm, int ldq, int Nv,
                                               const uint32_t* __restrict__ bits, int wpr, float* __restrict__ out,
                                               int64_t out_arm, int64_t out_ks, int ldo, int B, int KS, int tiles_n) {
    __shared__ __attribute__((aligned(16))) float Ps[32 * TN_LD];
    __shared__ __attribute__((aligned(16))) float Qs[32 * TN_LD];
    const int arm = blockIdx.z, ks = blockIdx.y;
    const int tm = blockIdx.x / tiles_n, tn = blockIdx.x % tiles_n;
    const int m0 = tm * 128, n0 = tn * 128;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, wm = wv >> 1, wn = wv & 1;
    const int l31 = lane & 31, hh = lane >> 5;
    const float* Pa = P + (int64_t)arm * p_arm;
    const float* Qa = Q + (int64_t)arm * q_arm;
    const int nbt = cdiv(B, 32);
    const int bt0 = (int)(((int64_t)ks * nbt) / KS), bt1 = (int)(((int64_t)(ks + 1) * nbt) / KS);
    const int rr = tid >> 5, c4 = tid & 31;
    const int pc = m0 + c4 * 4, qc = n0 + c4 * 4;
    const bool pok = pc < Mv, qok = qc < Nv;
    const bool qone = QONES && (qc == Nv);
    const int pcc = pok ? pc : 0, qcc = qok ? qc : 0;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = zero16();
    float4 rp[4], rq[4];
    auto load_tiles = [&](int bt) {
        uint32_t wd[4];
        int rows[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) rows[i] = bt * 32 + rr + 8 * i;
        if (QMASK) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                wd[i] = bits[((int64_t)arm * B + min(rows[i], B - 1)) * wpr + (qcc >> 5)];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
            rp[i] = *reinterpret_cast<const float4*>(Pa + (int64_t)min(rows[i], B - 1) * ldp + pcc);
#pragma unroll
        for (int i = 0; i < 4; ++i)
            rq[i] = *reinterpret_cast<const float4*>(Qa + (int64_t)min(rows[i], B - 1) * ldq + qcc);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            rp[i] = sel4(pok && rows[i] < B, rp[i]);      // zero P rows past the batch: their products vanish
            float4 v = sel4(qok, rq[i]);
            if (QMASK) v = mask4(v, wd[i] >> (qcc & 31));
            if (qone) v.x = 1.f;
            rq[i] = v;
        }
    };
    if (bt0 < bt1) load_tiles(bt0);
    for (int bt = bt0; bt < bt1; ++bt) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            *reinterpret_cast<float4*>(&Ps[(rr + 8 * i) * TN_LD + c4 * 4]) = rp[i];
            *reinterpret_cast<float4*>(&Qs[(rr + 8 * i) * TN_LD + c4 * 4]) = rq[i];
        }
        __syncthreads();
        if (bt + 1 < bt1) load_tiles(bt + 1);
        const float* la = Ps + hh * TN_LD + wm * 64 + l31;
        const float* lb = Qs + hh * TN_LD + wn * 64 + l31;
#pragma unroll 4
        for (int s = 0; s < 16; ++s) {
            const float a0 = la[2 * s * TN_LD], a1 = la[2 * s * TN_LD + 32];
            const float q0 = lb[2 * s * TN_LD], q1 = lb[2 * s * TN_LD + 32];
            acc[0][0] = mfma32(a0, q0, acc[0][0]);
            acc[0][1] = mfma32(a0, q1, acc[0][1]);
            acc[1][0] = mfma32(a1, q0, acc[1][0]);
            acc[1][1] = mfma32(a1, q1, acc[1][1]);
        }
        __syncthreads();
    }
    float* o = out + (int64_t)ks * out_ks + (int64_t)arm * out_arm;
    const int ncols = Nv + (QONES ? 1 : 0);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * 64 + i * 32 + acc_row(r, lane);
                const int n = n0 + wn * 64 + j * 32 + l31;
                if (m < Mv && n < ncols) o[(int64_t)m * ldo + n] = acc[i][j][r];
            }
}

// TN product with N = 100 + ones column (dW11 | db11 at fc_dim 100): a wave owns 32 rows x (3 MFMA column tiles + 5
// leftover columns done by plain FMAs on the P fragment it already holds); requires ldo % 4 == 0 and Nv == 100
__global__ __launch_bounds__(256, 2) void k_tn_v3n(const float* __restrict__ P, int64_t p_arm, int ldp, int Mv,
                                               const float* __restrict__ Q, int64_t q_arm, int ldq, int Nv,
                                               const uint32_t* __restrict__ bits, int wpr, float* __restrict__ out,
                                               int64_t out_arm, int64_t out_ks, int ldo, int B, int KS, int tiles_n) {
    __shared__ __attribute__((aligned(16))) float Ps[32 * TN_LD];
    __shared__ __attribute__((aligned(16))) float Qs[32 * TN_LD];
    const int arm = blockIdx.z, ks = blockIdx.y;
    const int tm = blockIdx.x / tiles_n, tn = blockIdx.x % tiles_n;
    const int m0 = tm * 128, n0 = tn * 128;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int l31 = lane & 31, hh = lane >> 5;
    const float* Pa = P + (int64_t)arm * p_arm;
    const float* Qa = Q + (int64_t)arm * q_arm;
    const int nbt = cdiv(B, 32);
    const int bt0 = (int)(((int64_t)ks * nbt) / KS), bt1 = (int)(((int64_t)(ks + 1) * nbt) / KS);
    const int rr = tid >> 5, c4 = tid & 31;
    const int pc = m0 + c4 * 4, qc = n0 + c4 * 4;
    const bool pok = pc < Mv, qok = qc < Nv;
    const bool qone = true && (qc == Nv);
    const int pcc = pok ? pc : 0, qcc = qok ? qc : 0;

    f32x16 acc[3] = {zero16(), zero16(), zero16()};   // rows [32 wv, 32 wv + 32), columns [32 j, 32 j + 32)
    float lo[5] = {0.f, 0.f, 0.f, 0.f, 0.f};          // columns 96..100 of row (lane & 31): this lane's batch rows only
    float4 rp[4], rq[4];
    auto load_tiles = [&](int bt) {
        uint32_t wd[4];
        int rows[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) rows[i] = bt * 32 + rr + 8 * i;
        if (false) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                wd[i] = bits[((int64_t)arm * B + min(rows[i], B - 1)) * wpr + (qcc >> 5)];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
            rp[i] = *reinterpret_cast<const float4*>(Pa + (int64_t)min(rows[i], B - 1) * ldp + pcc);
#pragma unroll
        for (int i = 0; i < 4; ++i)
            rq[i] = *reinterpret_cast<const float4*>(Qa + (int64_t)min(rows[i], B - 1) * ldq + qcc);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            rp[i] = sel4(pok && rows[i] < B, rp[i]);      // zero P rows past the batch: their products vanish
            float4 v = sel4(qok, rq[i]);
            if (false) v = mask4(v, wd[i] >> (qcc & 31));
            if (qone) v.x = 1.f;
            rq[i] = v;
        }
    };
    if (bt0 < bt1) load_tiles(bt0);
    for (int bt = bt0; bt < bt1; ++bt) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            *reinterpret_cast<float4*>(&Ps[(rr + 8 * i) * TN_LD + c4 * 4]) = rp[i];
            *reinterpret_cast<float4*>(&Qs[(rr + 8 * i) * TN_LD + c4 * 4]) = rq[i];
        }
        __syncthreads();
        if (bt + 1 < bt1) load_tiles(bt + 1);
        const float* la = Ps + hh * TN_LD + wv * 32 + l31;
        const float* lb = Qs + hh * TN_LD + l31;
        const float* ll = Qs + hh * TN_LD + 96;       // Q[b][96..100]: one address per half wave
#pragma unroll 4
        for (int s = 0; s < 16; ++s) {
            const float a = la[2 * s * TN_LD];
            const float q0 = lb[2 * s * TN_LD], q1 = lb[2 * s * TN_LD + 32], q2 = lb[2 * s * TN_LD + 64];
            const float4 w4 = *reinterpret_cast<const float4*>(ll + 2 * s * TN_LD);
            const float w5 = ll[2 * s * TN_LD + 4];
            acc[0] = mfma32(a, q0, acc[0]);
            acc[1] = mfma32(a, q1, acc[1]);
            acc[2] = mfma32(a, q2, acc[2]);
            // the five output columns (four of d10 and the ones column) that do not fill a 32-wide MFMA tile
            lo[0] = fmaf(a, w4.x, lo[0]); lo[1] = fmaf(a, w4.y, lo[1]); lo[2] = fmaf(a, w4.z, lo[2]); lo[3] = fmaf(a, w4.w, lo[3]);
            lo[4] = fmaf(a, w5, lo[4]);
        }
        __syncthreads();
    }
    float* o = out + (int64_t)ks * out_ks + (int64_t)arm * out_arm;
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + wv * 32 + acc_row(r, lane);
            if (m < Mv) o[(int64_t)m * ldo + j * 32 + l31] = acc[j][r];
        }
    // lanes l and l ^ 32 hold the even / odd batch rows of the same output row
#pragma unroll
    for (int c = 0; c < 5; ++c) lo[c] += __shfl_xor(lo[c], 32, 64);
    {
        const int m = m0 + wv * 32 + l31;
        if (hh == 0 && m < Mv) {
            *reinterpret_cast<float4*>(o + (int64_t)m * ldo + 96) = make_float4(lo[0], lo[1], lo[2], lo[3]);
            o[(int64_t)m * ldo + 100] = lo[4];
        }
    }
}

// TN product with M = 100 output rows (dW1 at fc_dim 100): as k_tn_v2, but a wave owns (3 MFMA row tiles + 4 leftover
// rows done by plain FMAs on the Q fragment it already holds) x 32 columns instead of 64 x 64 of a 128-row tile with
// 28 padding rows
template <bool QMASK>
__global__ __launch_bounds__(256, 2) void k_tn_v3m(const float* __restrict__ P, int64_t p_arm, int ldp, int Mv,
                                               const float* __restrict__ Q, int64_t q_arm, int ldq, int Nv,
                                               const uint32_t* __restrict__ bits, int wpr, float* __restrict__ out,
                                               int64_t out_arm, int64_t out_ks, int ldo, int B, int KS, int tiles_n) {
    __shared__ __attribute__((aligned(16))) float Ps[32 * TN_LD];
    __shared__ __attribute__((aligned(16))) float Qs[32 * TN_LD];
    const int arm = blockIdx.z, ks = blockIdx.y;
    const int tm = blockIdx.x / tiles_n, tn = blockIdx.x % tiles_n;
    const int m0 = tm * 128, n0 = tn * 128;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int l31 = lane & 31, hh = lane >> 5;
    const float* Pa = P + (int64_t)arm * p_arm;
    const float* Qa = Q + (int64_t)arm * q_arm;
    const int nbt = cdiv(B, 32);
    const int bt0 = (int)(((int64_t)ks * nbt) / KS), bt1 = (int)(((int64_t)(ks + 1) * nbt) / KS);
    const int rr = tid >> 5, c4 = tid & 31;
    const int pc = m0 + c4 * 4, qc = n0 + c4 * 4;
    const bool pok = pc < Mv, qok = qc < Nv;
    const bool qone = false && (qc == Nv);
    const int pcc = pok ? pc : 0, qcc = qok ? qc : 0;

    f32x16 acc[3] = {zero16(), zero16(), zero16()};   // rows [32 i, 32 i + 32), columns [32 wv, 32 wv + 32)
    float lo[4] = {0.f, 0.f, 0.f, 0.f};               // rows 96..99 at column (lane & 31): this lane's batch rows only
    float4 rp[4], rq[4];
    auto load_tiles = [&](int bt) {
        uint32_t wd[4];
        int rows[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) rows[i] = bt * 32 + rr + 8 * i;
        if (QMASK) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                wd[i] = bits[((int64_t)arm * B + min(rows[i], B - 1)) * wpr + (qcc >> 5)];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
            rp[i] = *reinterpret_cast<const float4*>(Pa + (int64_t)min(rows[i], B - 1) * ldp + pcc);
#pragma unroll
        for (int i = 0; i < 4; ++i)
            rq[i] = *reinterpret_cast<const float4*>(Qa + (int64_t)min(rows[i], B - 1) * ldq + qcc);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            rp[i] = sel4(pok && rows[i] < B, rp[i]);      // zero P rows past the batch: their products vanish
            float4 v = sel4(qok, rq[i]);
            if (QMASK) v = mask4(v, wd[i] >> (qcc & 31));
            if (qone) v.x = 1.f;
            rq[i] = v;
        }
    };
    if (bt0 < bt1) load_tiles(bt0);
    for (int bt = bt0; bt < bt1; ++bt) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            *reinterpret_cast<float4*>(&Ps[(rr + 8 * i) * TN_LD + c4 * 4]) = rp[i];
            *reinterpret_cast<float4*>(&Qs[(rr + 8 * i) * TN_LD + c4 * 4]) = rq[i];
        }
        __syncthreads();
        if (bt + 1 < bt1) load_tiles(bt + 1);
        const float* la = Ps + hh * TN_LD + l31;
        const float* lb = Qs + hh * TN_LD + wv * 32 + l31;
        const float* ll = Ps + hh * TN_LD + 96;       // P[b][96..99]: one address per half wave
#pragma unroll 4
        for (int s = 0; s < 16; ++s) {
            const float a0 = la[2 * s * TN_LD], a1 = la[2 * s * TN_LD + 32], a2 = la[2 * s * TN_LD + 64];
            const float q = lb[2 * s * TN_LD];
            const float4 p4 = *reinterpret_cast<const float4*>(ll + 2 * s * TN_LD);
            acc[0] = mfma32(a0, q, acc[0]);
            acc[1] = mfma32(a1, q, acc[1]);
            acc[2] = mfma32(a2, q, acc[2]);
            // the four output rows that do not fill a 32-row MFMA tile: VALU FMAs in the MFMAs' shadow
            lo[0] = fmaf(p4.x, q, lo[0]); lo[1] = fmaf(p4.y, q, lo[1]); lo[2] = fmaf(p4.z, q, lo[2]); lo[3] = fmaf(p4.w, q, lo[3]);
        }
        __syncthreads();
    }
    float* o = out + (int64_t)ks * out_ks + (int64_t)arm * out_arm;
    const int n = n0 + wv * 32 + l31;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + i * 32 + acc_row(r, lane);
            if (m < Mv && n < Nv) o[(int64_t)m * ldo + n] = acc[i][r];
        }
    // lanes l and l ^ 32 hold the even / odd batch rows of the same column
#pragma unroll
    for (int c = 0; c < 4; ++c) lo[c] += __shfl_xor(lo[c], 32, 64);
    if (hh == 0 && n < Nv) {
#pragma unroll
        for (int c = 0; c < 4; ++c)
            if (m0 + 96 + c < Mv) o[(int64_t)(m0 + 96 + c) * ldo + n] = lo[c];
    }
}

// =============================================================================================
// fc11 forward + reconstruction loss + dZ11, staggered with 16-byte global accesses (the default fast path).
//   z = d10 W11^T by MFMA, "A-stationary": every wave keeps the d10 rows of its 32 cells in registers for the whole
//   kernel (13 float4 at H = 100) and only the W11 tile moves through LDS (double buffered).  Every 4 x 4 block of the accumulator is transposed inside its lane quad (DPP), after which a lane
//   holds FOUR CONSECUTIVE GENES of one cell: x is read and dZ11 written 16 B per lane and every instruction
//   covers whole 128-byte lines (8 cells x 32 genes) -- 8 + 8 vector memory instructions per 64-gene step
//   instead of the 32 + 32 of 4-byte accesses (more than the 63 the vmcnt counter can track: those loads stalled at issue).
//   A 512-thread workgroup covers 256 cells.  Waves 4-7 run one epilogue behind waves 0-3: on every SIMD one
//   wave's epilogue (VALU + stores) runs beside its partner's MFMAs instead of both leaving the matrix pipe
//   idle together.  One barrier per step.
//   BIASK: the bias rides in the K padding (d10 fragment k = H is 1, W tile column H is b11; needs H % 8 == 4).
// grid (ceil(B/256), NS, A)
// =============================================================================================
// on entry v[e] on lane i of a quad is M[e][i]; on exit it is M[i][e]
__device__ __forceinline__ void quad_transpose4(float& v0, float& v1, float& v2, float& v3, bool b0, bool b1) {
    const float sA = b1 ? v0 : v2, sB = b1 ? v1 : v3;
    const float rA = dpp_f<0x4E>(sA), rB = dpp_f<0x4E>(sB);      // quad_perm [2,3,0,1]
    v0 = b1 ? rA : v0; v1 = b1 ? rB : v1; v2 = b1 ? v2 : rA; v3 = b1 ? v3 : rB;
    const float s0 = b0 ? v0 : v1, s1 = b0 ? v2 : v3;
    const float r0 = dpp_f<0xB1>(s0), r1 = dpp_f<0xB1>(s1);      // quad_perm [1,0,3,2]
    v0 = b0 ? r0 : v0; v1 = b0 ? v1 : r0; v2 = b0 ? r1 : v2; v3 = b0 ? v3 : r1;
}

// XREC: x_rec is written (forward / inference calls); the train step instantiates XREC = false, where dZ11 is
// always written.  ABL: timing experiments only (1 no MFMA, 2 no x loads, 4 no dZ11 stores); 0 in production --
// compile-time so that the step body stays one basic block.
template <int FZ_KG, bool EXACT, bool BIASK, bool XREC, int ABL>
__global__ __launch_bounds__(512, 2) void k_fc11_zt(const float* __restrict__ d10, const float* __restrict__ params,
                                                    int64_t per_arm, int64_t w_off, int64_t b_off,
                                                    const float* __restrict__ x, int64_t x_arm_stride,
                                                    float* __restrict__ x_rec, float* __restrict__ dz11,
                                                    float* __restrict__ part, int n11, float coef, int need_grad,
                                                    int A, int B, int D, int H, int ldk, unsigned long long* dbgc) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Wbuf = smem;                     // [2][64][ldk]
    float* red = smem + 2 * 64 * ldk;       // [16]
    const int arm = blockIdx.z, ns = blockIdx.y, NS = gridDim.y, b0 = blockIdx.x * 256;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int l31 = lane & 31, hh = lane >> 5;
    const bool lb0 = (lane & 1) != 0, lb1 = (lane & 2) != 0;
    const int KP = rup(H, 8), kg = KP / 8, nc4 = KP / 4, hc4 = H / 4;
    const bool late = (ABL & 16) ? wv < 4 : wv >= 4;
    const float* W = params + (int64_t)arm * per_arm + w_off;     // [D, H]
    const float* bias = params + (int64_t)arm * per_arm + b_off;
    const float* xa = x + (int64_t)arm * x_arm_stride;
    float* dza = dz11 + (int64_t)arm * B * D;
    float* xra = XREC ? x_rec + (int64_t)arm * B * D : nullptr;
    const bool do_grad = XREC ? (need_grad != 0) : true;
    const int bw = b0 + 32 * wv;
    const bool rows_full = b0 + 256 <= B;

    // ---- d10 fragments (MFMA A operand): cell = bw + (lane & 31), k = 8 g + 4 hh .. + 3
    float4 afr[FZ_KG];
    {
        const int row = bw + l31;
        const float* p = d10 + ((int64_t)arm * B + min(row, B - 1)) * H + 4 * hh;
#pragma unroll
        for (int g = 0; g < FZ_KG; ++g) {
            const int k0 = 8 * g + 4 * hh;
            const bool ok = row < B && k0 < H;
            const float4 v = *reinterpret_cast<const float4*>(p + (ok ? 8 * g : 0));
            afr[g] = sel4(ok, v);
            if (BIASK && k0 == H) afr[g].x = 1.f;
        }
    }
    const int ntall = cdiv(D, 64);
    const int t0 = (int)(((int64_t)ns * ntall) / NS), t1 = (int)(((int64_t)(ns + 1) * ntall) / NS);
    const int srow = tid >> 3, spart = tid & 7;
    // The W tile for step t+1 is requested one step ahead and lands in registers while the MFMAs / epilogue run;
    // masking and the bias column are applied when it is written to LDS (touching the loaded values any
    // earlier makes the compiler wait for the loads where they are issued).
    float4 wreg[4];
    float wbias = 0.f;
    int wj = 0;
    auto prefetch_w = [&](int t) {
        wj = t * 64 + srow;
        const float* p = W + (int64_t)min(wj, D - 1) * H;
        if (BIASK) wbias = bias[min(wj, D - 1)];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = spart + 8 * i;
            wreg[i] = *reinterpret_cast<const float4*>(p + (c < hc4 ? c * 4 : 0));
        }
    };
    auto store_w = [&](float* Ws) {
        const bool jok = wj < D;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = spart + 8 * i;
            float4 v = sel4(jok && c < hc4, wreg[i]);
            if (BIASK && c == hc4) v.x = jok ? wbias : 0.f;
            if (c < nc4) *reinterpret_cast<float4*>(&Ws[srow * ldk + c * 4]) = v;
        }
    };
    // after the quad transposes: register group q of a lane is cell cq = bw + 8 q + 4 hh + (lane & 3),
    // genes j0 + 32 c + 4 ((lane & 31) >> 2) .. + 3
    const int cbase = bw + 4 * hh + (l31 & 3);
    const int gcol = 4 * (l31 >> 2);
    const uint32_t lane_off = (uint32_t)cbase * (uint32_t)D + (uint32_t)gcol;   // interior: + wave-uniform part
    float se = 0.f;
    int mism = 0;   // per-lane count
    f32x16 z0 = zero16(), z1 = zero16();
    float4 xv[2][4];

    // one 16-byte piece (gene half c, cell group q) of tile t's x
    auto load_x1 = [&](int t, int c, int q, float4& dst, auto edge_tag) __attribute__((always_inline)) {
        constexpr bool EDGE = decltype(edge_tag)::value;
        if (ABL & 2) { dst = make_float4(0.f, 0.f, 0.f, 0.f); return; }
        if (!EDGE) {
            const float* xu = xa + ((int64_t)8 * q * D + t * 64 + 32 * c);      // wave-uniform
            dst = *reinterpret_cast<const float4*>(xu + lane_off);
        } else {
            const int cellq = min(cbase + 8 * q, B - 1), col = min(t * 64 + 32 * c + gcol, D - 4);
            dst = *reinterpret_cast<const float4*>(xa + (uint32_t)cellq * (uint32_t)D + (uint32_t)col);
        }
    };
    auto load_x = [&](int t, float4 (&dst)[2][4], auto edge_tag) __attribute__((always_inline)) {
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int q = 0; q < 4; ++q) load_x1(t, c, q, dst[c][q], edge_tag);
    };
    auto mfma_tile = [&](const float* Ws) __attribute__((always_inline)) {
        z0 = zero16();
        z1 = zero16();
        const float* pb = Ws + l31 * ldk + 4 * hh;
        float4 q0 = *reinterpret_cast<const float4*>(pb);
        float4 q1 = *reinterpret_cast<const float4*>(pb + 32 * ldk);
        if (!(ABL & 1))
#pragma unroll
        for (int g = 0; g < FZ_KG; ++g) {
            if (EXACT || g < kg) {
                const int gn = EXACT ? ((g + 1 < FZ_KG) ? g + 1 : g) : ((g + 1 < kg) ? g + 1 : g);
                const float4 n0 = *reinterpret_cast<const float4*>(pb + 8 * gn);
                const float4 n1 = *reinterpret_cast<const float4*>(pb + 32 * ldk + 8 * gn);
                const float4 a = afr[g];
                z0 = mfma32(a.x, q0.x, z0); z1 = mfma32(a.x, q1.x, z1);
                z0 = mfma32(a.y, q0.y, z0); z1 = mfma32(a.y, q1.y, z1);
                z0 = mfma32(a.z, q0.z, z0); z1 = mfma32(a.z, q1.z, z1);
                z0 = mfma32(a.w, q0.w, z0); z1 = mfma32(a.w, q1.w, z1);
                q0 = n0; q1 = n1;
            }
        }
    };
    // x_rec = relu(z + b); e = x_rec - x; dZ11 = coef * e where x_rec > 0 (nn_model.py:544-546 + autograd)
    auto epilogue = [&](int t, const float4 (&xs)[2][4], auto edge_tag) __attribute__((always_inline)) {
        constexpr bool EDGE = decltype(edge_tag)::value;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int col = t * 64 + 32 * c + gcol;
            float bq[4] = {0.f, 0.f, 0.f, 0.f};
            if (!BIASK) {
                const float4 b4 = *reinterpret_cast<const float4*>(bias + (EDGE ? min(col, D - 4) : col));
                bq[0] = b4.x; bq[1] = b4.y; bq[2] = b4.z; bq[3] = b4.w;
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float zz[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) zz[e] = (c == 0 ? z0[4 * q + e] : z1[4 * q + e]);
                quad_transpose4(zz[0], zz[1], zz[2], zz[3], lb0, lb1);
                const bool ok = !EDGE || ((cbase + 8 * q < B) && (col < D));
                const float xin[4] = {xs[c][q].x, xs[c][q].y, xs[c][q].z, xs[c][q].w};
                float xr[4], dzv[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    xr[e] = fmaxf(zz[e] + bq[e], 0.f);
                    const float er = xr[e] - xin[e];
                    dzv[e] = (xr[e] > 0.f) ? coef * er : 0.f;
                    se += ok ? er * er : 0.f;
                    // per-lane count (keeping 64 ballot masks alive spills SGPRs into VGPR lanes)
                    mism += (ok && ((xr[e] > 0.1f) != (xin[e] > 0.1f))) ? 1 : 0;
                }
                if (ok) {
                    if (!EDGE) {
                        const int64_t uo = (int64_t)8 * q * D + t * 64 + 32 * c;      // wave-uniform
                        if (XREC) *reinterpret_cast<float4*>(xra + uo + lane_off) = make_float4(xr[0], xr[1], xr[2], xr[3]);
                        if (do_grad && !(ABL & 4))
                            *reinterpret_cast<float4*>(dza + uo + lane_off) = make_float4(dzv[0], dzv[1], dzv[2], dzv[3]);
                    } else {
                        const uint32_t off = (uint32_t)(cbase + 8 * q) * (uint32_t)D + (uint32_t)col;
                        if (XREC) *reinterpret_cast<float4*>(xra + off) = make_float4(xr[0], xr[1], xr[2], xr[3]);
                        if (do_grad) *reinterpret_cast<float4*>(dza + off) = make_float4(dzv[0], dzv[1], dzv[2], dzv[3]);
                    }
                }
                // piece fence.  The accumulators are pinned here: integer adds are associative, so left alone the
                // optimiser turns the 32 mismatch increments of a step into one tree at its end and keeps every
                // compare result (and the x_rec / x values feeding it) alive until then; and the scheduler hoists
                // all compares to the top and spills their lane masks into VGPR lanes.
                asm volatile("" : "+v"(mism), "+v"(se));
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };

    if (t0 < t1) {
        prefetch_w(t0);
        store_w(Wbuf);
    }
    __syncthreads();
    int cur = 0;
    // ABL bit 3: shader-clock stamps per phase, summed over waves (early waves: counters 0-5, late: 8-13)
    constexpr bool STAMPS = (ABL & 8) != 0;
    unsigned long long ph[6] = {0, 0, 0, 0, 0, 0}, tprev = 0;
    auto stamp = [&](int i) {
        if (STAMPS) {
            __builtin_amdgcn_sched_barrier(0);
            const unsigned long long now = __builtin_amdgcn_s_memtime();
            __builtin_amdgcn_s_waitcnt(0xC07F);
            ph[i] += now - tprev;
            tprev = now;
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    if (STAMPS) { tprev = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_s_waitcnt(0xC07F); }
    // Every step body is straight-line per role: no condition around a load or a store (a conditional epilogue or
    // prefetch makes the compiler's vmcnt bookkeeping pessimistic at the join, and it then waits for the previous
    // step's dZ11 stores before touching the W tile).  The W prefetch of the last step re-reads a clamped row and
    // its LDS copy is never used; the late waves' first step (no tile behind them yet) is a separate instance.
    auto step = [&](int t, auto edge_tag, auto first_tag) __attribute__((always_inline)) {
        constexpr bool FIRST = decltype(first_tag)::value;
        const float* Ws = Wbuf + cur * 64 * ldk;
        // phase fences: without them the scheduler hoists epilogue work that only needs x (the mismatch
        // compares) to the top of the step, where it waits for the loads just issued, and sinks W loads behind
        // the dZ11 stores
        if (!late) {
            load_x(t, xv, edge_tag);
            __builtin_amdgcn_sched_barrier(0);
            stamp(0);
            mfma_tile(Ws);
            if (STAMPS) asm volatile("" :: "v"(z0[0]), "v"(z1[15]));
            __builtin_amdgcn_sched_barrier(0);
            stamp(1);
            prefetch_w(t + 1);
            __builtin_amdgcn_sched_barrier(0);
            stamp(2);
            epilogue(t, xv, edge_tag);
            __builtin_amdgcn_sched_barrier(0);
            stamp(3);
        } else {
            // W loads before the epilogue's stores: vmcnt retires in order
            prefetch_w(t + 1);
            __builtin_amdgcn_sched_barrier(0);
            stamp(2);
            if (!FIRST) epilogue(t - 1, xv, edge_tag);   // z still holds tile t-1
            __builtin_amdgcn_sched_barrier(0);
            stamp(3);
            // x for this tile goes into the registers the epilogue has just drained; it is not needed before the
            // next step's epilogue, and issuing it here (not at the top of the step with waves 0-3's loads) spreads
            // the CU's outstanding misses over the step
            load_x(t, xv, edge_tag);
            __builtin_amdgcn_sched_barrier(0);
            stamp(0);
            mfma_tile(Ws);
            if (STAMPS) asm volatile("" :: "v"(z0[0]), "v"(z1[15]));
            __builtin_amdgcn_sched_barrier(0);
            stamp(1);
        }
        store_w(Wbuf + (cur ^ 1) * 64 * ldk);
        stamp(4);
        lds_barrier();      // LDS only: the dZ11 / x_rec stores stay in flight
        stamp(5);
        cur ^= 1;
    };
    // first step (guarded body, valid for any tile), interior steps (all 256 cells and all 64 genes in range),
    // then the guarded ones
    const int t_mid = rows_full ? max(t0 + 1, min(t1, D / 64)) : t0 + 1;
    if (t0 < t1) step(t0, VecTag{}, VecTag{});
    for (int t = t0 + 1; t < t_mid; ++t) step(t, ScalarTag{}, ScalarTag{});      // ::value == false: interior body
    for (int t = max(t_mid, t0 + 1); t < t1; ++t) step(t, VecTag{}, ScalarTag{});    // guarded body
    if (late && t1 > t0) epilogue(t1 - 1, xv, VecTag{});
    if (STAMPS && lane == 0) {
        unsigned long long* dbg = dbgc + (late ? 8 : 0);
        for (int i = 0; i < 6; ++i) atomicAdd(dbg + i, ph[i]);
        atomicAdd(dbg + 6, 1ull);
    }
    se = wave_sum(se);
    const float mismf = wave_sum((float)mism);
    if (lane == 0) { red[wv * 2] = se; red[wv * 2 + 1] = mismf; }
    __syncthreads();
    if (tid == 0) {
        float* p = part + ((int64_t)arm * n11 + (int64_t)blockIdx.x * NS + ns) * 2;
        float s0 = 0.f, s1 = 0.f;
        for (int w = 0; w < 8; ++w) { s0 += red[w * 2]; s1 += red[w * 2 + 1]; }
        p[0] = s0;
        p[1] = s1;
    }
}

// =============================================================================================
// fc11 forward + reconstruction loss + dZ11 + d(d10), train step at fc_dim = 100.  k_fc11_zt with the d(d10) GEMM of
// k_gd10_v3 folded in: each wave keeps the dZ11 tile it has just produced in LDS and multiplies it with the W11 tile
// that is already there (3 MFMA column tiles + 4 VALU columns), accumulating its 32 cells' d(d10) over the gene
// range in registers.  k_fc11_zt alone leaves the matrix pipe about half idle (it is bound by the CU's memory queue);
// the fused kernel fills that time instead of launching a second GEMM that re-reads dZ11 (200 MB).  Three W buffers:
// the late waves multiply tile t-1 while tile t+1 is staged.  Output: d(d10) slabs [NS][A][B][H], summed by the
// decoder's backward prologue.
// =============================================================================================
typedef unsigned int bu32x4 __attribute__((ext_vector_type(4)));
constexpr int ZG_LD = 68;   // dZ11 tile row stride (ld / 4 odd: conflict-free b128 A-fragment reads)

template <int FZ_KG, bool EXACT, bool BIASK>
__global__ __launch_bounds__(512, 2) void k_fc11_zg(const float* __restrict__ d10, const float* __restrict__ params,
                                                    int64_t per_arm, int64_t w_off, int64_t b_off,
                                                    const float* __restrict__ x, int64_t x_arm_stride,
                                                    float* __restrict__ x_rec, float* __restrict__ dz11,
                                                    float* __restrict__ part, int n11, float coef, int need_grad,
                                                    int A, int B, int D, int H, int ldk, float* __restrict__ gd_slab) {
    constexpr bool XREC = false;
    constexpr int ABL = 0;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Wbuf = smem;                               // [3][64][ldk]: tiles t-1 (late waves' d(d10)), t, t+1 (being staged)
    float* DZall = smem + 3 * 64 * ldk;               // [8 waves][32 cells][ZG_LD]: each wave's dZ11 tile
    float* red = DZall + 8 * 32 * ZG_LD;              // [16]
    const int arm = blockIdx.z, ns = blockIdx.y, NS = gridDim.y, b0 = blockIdx.x * 256;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int l31 = lane & 31, hh = lane >> 5;
    const bool lb0 = (lane & 1) != 0, lb1 = (lane & 2) != 0;
    const int KP = rup(H, 8), kg = KP / 8, nc4 = KP / 4, hc4 = H / 4;
    const bool late = (ABL & 16) ? wv < 4 : wv >= 4;
    const float* W = params + (int64_t)arm * per_arm + w_off;     // [D, H]
    const float* bias = params + (int64_t)arm * per_arm + b_off;
    const float* xa = x + (int64_t)arm * x_arm_stride;
    float* dza = dz11 + (int64_t)arm * B * D;
    float* xra = XREC ? x_rec + (int64_t)arm * B * D : nullptr;
    const bool do_grad = XREC ? (need_grad != 0) : true;
    const int bw = b0 + 32 * wv;
    float* DZw = DZall + wv * (32 * ZG_LD);
    const bool rows_full = b0 + 256 <= B;

    // ---- d10 fragments (MFMA A operand): cell = bw + (lane & 31), k = 8 g + 4 hh .. + 3
    float4 afr[FZ_KG];
    {
        const int row = bw + l31;
        const float* p = d10 + ((int64_t)arm * B + min(row, B - 1)) * H + 4 * hh;
#pragma unroll
        for (int g = 0; g < FZ_KG; ++g) {
            const int k0 = 8 * g + 4 * hh;
            const bool ok = row < B && k0 < H;
            const float4 v = *reinterpret_cast<const float4*>(p + (ok ? 8 * g : 0));
            afr[g] = sel4(ok, v);
            if (BIASK && k0 == H) afr[g].x = 1.f;
        }
    }
    const int ntall = cdiv(D, 64);
    const int t0 = (int)(((int64_t)ns * ntall) / NS), t1 = (int)(((int64_t)(ns + 1) * ntall) / NS);
    const int srow = tid >> 3, spart = tid & 7;
    // The W tile for step t+1 is requested one step ahead and lands in registers while the MFMAs / epilogue run;
    // masking and the bias column are applied when it is written to LDS (touching the loaded values any
    // earlier makes the compiler wait for the loads where they are issued).
    float4 wreg[4];
    float wbias = 0.f;
    int wj = 0;
    auto prefetch_w = [&](int t) {
        wj = t * 64 + srow;
        const float* p = W + (int64_t)min(wj, D - 1) * H;
        if (BIASK) wbias = bias[min(wj, D - 1)];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = spart + 8 * i;
            wreg[i] = *reinterpret_cast<const float4*>(p + (c < hc4 ? c * 4 : 0));
        }
    };
    auto store_w = [&](float* Ws) {
        const bool jok = wj < D;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = spart + 8 * i;
            float4 v = sel4(jok && c < hc4, wreg[i]);
            if (BIASK && c == hc4) v.x = jok ? wbias : 0.f;
            if (c < nc4) *reinterpret_cast<float4*>(&Ws[srow * ldk + c * 4]) = v;
        }
    };
    // after the quad transposes: register group q of a lane is cell cq = bw + 8 q + 4 hh + (lane & 3),
    // genes j0 + 32 c + 4 ((lane & 31) >> 2) .. + 3
    const int cbase = bw + 4 * hh + (l31 & 3);
    const int gcol = 4 * (l31 >> 2);
    // interior accesses: buffer instructions with one 32-bit per-lane byte offset (VGPR) + a wave-uniform byte offset
    // (SGPR); flat 64-bit addresses cost a VGPR pair per (half, cell group) for loads and again for stores
    const uint32_t lane_boff = ((uint32_t)cbase * (uint32_t)D + (uint32_t)gcol) * 4u;   // B * D < 2^30
    const int arm_bytes = (int)((uint32_t)B * (uint32_t)D * 4u);
    const auto rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xa), 0, arm_bytes, 0x00020000);
    const auto rs_dz = __builtin_amdgcn_make_buffer_rsrc(dza, 0, arm_bytes, 0x00020000);
    float se = 0.f;
    int mism = 0;   // per-lane count
    f32x16 z0 = zero16(), z1 = zero16();
    float4 xv[2][4];

    // one 16-byte piece (gene half c, cell group q) of tile t's x
    auto load_x1 = [&](int t, int c, int q, float4& dst, auto edge_tag) __attribute__((always_inline)) {
        constexpr bool EDGE = decltype(edge_tag)::value;
        if (ABL & 2) { dst = make_float4(0.f, 0.f, 0.f, 0.f); return; }
        if (!EDGE) {
            const uint32_t so = ((uint32_t)(8 * q) * (uint32_t)D + (uint32_t)(t * 64 + 32 * c)) * 4u;   // wave-uniform
            dst = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, (int)lane_boff, (int)so, 0));
        } else {
            const int cellq = min(cbase + 8 * q, B - 1), col = min(t * 64 + 32 * c + gcol, D - 4);
            dst = *reinterpret_cast<const float4*>(xa + (uint32_t)cellq * (uint32_t)D + (uint32_t)col);
        }
    };
    auto load_x = [&](int t, float4 (&dst)[2][4], auto edge_tag) __attribute__((always_inline)) {
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int q = 0; q < 4; ++q) load_x1(t, c, q, dst[c][q], edge_tag);
    };
    auto mfma_tile = [&](const float* Ws) __attribute__((always_inline)) {
        z0 = zero16();
        z1 = zero16();
        const float* pb = Ws + l31 * ldk + 4 * hh;
        // no register double-buffering of the W fragments here (this kernel sits at the 256-VGPR cap): the partner
        // wave's MFMAs cover the LDS latency
#pragma unroll
        for (int g = 0; g < FZ_KG; ++g) {
            if (EXACT || g < kg) {
                const float4 q0 = *reinterpret_cast<const float4*>(pb + 8 * g);
                const float4 q1 = *reinterpret_cast<const float4*>(pb + 32 * ldk + 8 * g);
                const float4 a = afr[g];
                z0 = mfma32(a.x, q0.x, z0); z1 = mfma32(a.x, q1.x, z1);
                z0 = mfma32(a.y, q0.y, z0); z1 = mfma32(a.y, q1.y, z1);
                z0 = mfma32(a.z, q0.z, z0); z1 = mfma32(a.z, q1.z, z1);
                z0 = mfma32(a.w, q0.w, z0); z1 = mfma32(a.w, q1.w, z1);
            }
        }
    };
    // x_rec = relu(z + b); e = x_rec - x; dZ11 = coef * e where x_rec > 0 (nn_model.py:544-546 + autograd)
    auto epilogue = [&](int t, const float4 (&xs)[2][4], auto edge_tag) __attribute__((always_inline)) {
        constexpr bool EDGE = decltype(edge_tag)::value;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int col = t * 64 + 32 * c + gcol;
            float bq[4] = {0.f, 0.f, 0.f, 0.f};
            if (!BIASK) {
                const float4 b4 = *reinterpret_cast<const float4*>(bias + (EDGE ? min(col, D - 4) : col));
                bq[0] = b4.x; bq[1] = b4.y; bq[2] = b4.z; bq[3] = b4.w;
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float zz[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) zz[e] = (c == 0 ? z0[4 * q + e] : z1[4 * q + e]);
                quad_transpose4(zz[0], zz[1], zz[2], zz[3], lb0, lb1);
                const bool ok = !EDGE || ((cbase + 8 * q < B) && (col < D));
                const float xin[4] = {xs[c][q].x, xs[c][q].y, xs[c][q].z, xs[c][q].w};
                float xr[4], dzv[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    xr[e] = fmaxf(zz[e] + bq[e], 0.f);
                    const float er = xr[e] - xin[e];
                    dzv[e] = (ok && xr[e] > 0.f) ? coef * er : 0.f;
                    se += ok ? er * er : 0.f;
                    // per-lane count (keeping 64 ballot masks alive spills SGPRs into VGPR lanes)
                    mism += (ok && ((xr[e] > 0.1f) != (xin[e] > 0.1f))) ? 1 : 0;
                }
                // the wave's own dZ11 tile, [cell][gene], for its d(d10) MFMAs (zero where the piece is out of range)
                *reinterpret_cast<float4*>(&DZw[(8 * q + 4 * hh + (l31 & 3)) * ZG_LD + 32 * c + gcol]) =
                    make_float4(dzv[0], dzv[1], dzv[2], dzv[3]);
                if (ok) {
                    if (!EDGE) {
                        const uint32_t so = ((uint32_t)(8 * q) * (uint32_t)D + (uint32_t)(t * 64 + 32 * c)) * 4u;   // wave-uniform
                        __builtin_amdgcn_raw_buffer_store_b128(
                            __builtin_bit_cast(bu32x4, make_float4(dzv[0], dzv[1], dzv[2], dzv[3])), rs_dz, (int)lane_boff, (int)so, 0);
                        // A 128-bit buffer store reads its data registers over more than one cycle.  hipcc pads this
                        // hazard only for stores WITHOUT an SGPR offset; measured on gfx950 it exists with one too:
                        // the next piece's first VALU write into the same registers (zero wait states after the
                        // store in the generated code) corrupted the stored dZ11, nondeterministically and mostly
                        // under memory-pipe back-pressure (81k of 50M elements at the benchmark shape; 0 with the
                        // two wait states the compiler gives global stores).
                        asm volatile("s_nop 1");
                    } else {
                        const uint32_t off = (uint32_t)(cbase + 8 * q) * (uint32_t)D + (uint32_t)col;
                        if (XREC) *reinterpret_cast<float4*>(xra + off) = make_float4(xr[0], xr[1], xr[2], xr[3]);
                        if (do_grad) *reinterpret_cast<float4*>(dza + off) = make_float4(dzv[0], dzv[1], dzv[2], dzv[3]);
                    }
                }
                // piece fence.  The accumulators are pinned here: integer adds are associative, so left alone the
                // optimiser turns the 32 mismatch increments of a step into one tree at its end and keeps every
                // compare result (and the x_rec / x values feeding it) alive until then; and the scheduler hoists
                // all compares to the top and spills their lane masks into VGPR lanes.
                asm volatile("" : "+v"(mism), "+v"(se));
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };

    // d(d10) += dZ11 tile (this wave's 32 cells x 64 genes, from LDS) x W11 tile (64 genes x H): 3 MFMA column tiles
    // + columns 96..99 by plain FMAs, as k_gd10_v3
    f32x16 gacc[3] = {zero16(), zero16(), zero16()};
    float glo[4] = {0.f, 0.f, 0.f, 0.f};
    auto mfma_gd = [&](const float* Ws) __attribute__((always_inline)) {
        const float* pa = DZw + l31 * ZG_LD + 4 * hh;
        const float* pb = Ws + (4 * hh) * ldk + l31;
        const float* pl = Ws + (4 * hh) * ldk + 96;
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            const float4 a = *reinterpret_cast<const float4*>(pa + 8 * g);
            const float av[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float* q = pb + (8 * g + e) * ldk;
                const float q0 = q[0], q1 = q[32], q2 = q[64];
                const float4 w = *reinterpret_cast<const float4*>(pl + (8 * g + e) * ldk);
                gacc[0] = mfma32(av[e], q0, gacc[0]);
                gacc[1] = mfma32(av[e], q1, gacc[1]);
                gacc[2] = mfma32(av[e], q2, gacc[2]);
                glo[0] = fmaf(av[e], w.x, glo[0]); glo[1] = fmaf(av[e], w.y, glo[1]);
                glo[2] = fmaf(av[e], w.z, glo[2]); glo[3] = fmaf(av[e], w.w, glo[3]);
            }
            __builtin_amdgcn_sched_barrier(0);   // keep the fragment registers of one k group at a time
        }
    };

    if (t0 < t1) {
        prefetch_w(t0);
        store_w(Wbuf);
    }
    __syncthreads();
    int cur = 0;   // W buffer of the current tile; (cur + 1) % 3 is being staged, (cur + 2) % 3 holds the previous tile
    // Every step body is straight-line per role: no condition around a load or a store (a conditional epilogue or
    // prefetch makes the compiler's vmcnt bookkeeping pessimistic at the join, and it then waits for the previous
    // step's dZ11 stores before touching the W tile).  The W prefetch of the last step re-reads a clamped row and
    // its LDS copy is never used; the late waves' first step (no tile behind them yet) is a separate instance.
    auto step = [&](int t, auto edge_tag, auto first_tag) __attribute__((always_inline)) {
        constexpr bool FIRST = decltype(first_tag)::value;
        const float* Ws = Wbuf + cur * 64 * ldk;
        const int nxt = cur == 2 ? 0 : cur + 1, prv = cur == 0 ? 2 : cur - 1;
        // phase fences: without them the scheduler hoists epilogue work that only needs x (the mismatch
        // compares) to the top of the step, where it waits for the loads just issued, and sinks W loads behind
        // the dZ11 stores
        if (!late) {
            load_x(t, xv, edge_tag);
            __builtin_amdgcn_sched_barrier(0);
            mfma_tile(Ws);
            __builtin_amdgcn_sched_barrier(0);
            prefetch_w(t + 1);
            __builtin_amdgcn_sched_barrier(0);
            epilogue(t, xv, edge_tag);
            __builtin_amdgcn_sched_barrier(0);
            mfma_gd(Ws);
            __builtin_amdgcn_sched_barrier(0);
        } else {
            // W loads before the epilogue's stores: vmcnt retires in order
            prefetch_w(t + 1);
            __builtin_amdgcn_sched_barrier(0);
            if (!FIRST) {
                epilogue(t - 1, xv, edge_tag);   // z still holds tile t-1
                __builtin_amdgcn_sched_barrier(0);
                mfma_gd(Wbuf + prv * 64 * ldk);
            }
            __builtin_amdgcn_sched_barrier(0);
            load_x(t, xv, edge_tag);
            __builtin_amdgcn_sched_barrier(0);
            mfma_tile(Ws);
            __builtin_amdgcn_sched_barrier(0);
        }
        // tile t+1 goes where tile t-2 was: its last readers (the late waves' d(d10) of step t-1) are past a barrier
        store_w(Wbuf + nxt * 64 * ldk);
        lds_barrier();      // LDS only: the dZ11 stores stay in flight
        cur = nxt;
    };
    // first step (guarded body, valid for any tile), interior steps (all 256 cells and all 64 genes in range),
    // then the guarded ones
    const int t_mid = rows_full ? max(t0 + 1, min(t1, D / 64)) : t0 + 1;
    if (t0 < t1) step(t0, VecTag{}, VecTag{});
    for (int t = t0 + 1; t < t_mid; ++t) step(t, ScalarTag{}, ScalarTag{});      // ::value == false: interior body
    for (int t = max(t_mid, t0 + 1); t < t1; ++t) step(t, VecTag{}, ScalarTag{});    // guarded body
    if (late && t1 > t0) {
        epilogue(t1 - 1, xv, VecTag{});
        mfma_gd(Wbuf + (cur == 0 ? 2 : cur - 1) * 64 * ldk);   // cur has moved past the last tile
    }
    // ---- d(d10) partial of this gene range: slab [NS][A][B][H]
    {
        float* out = gd_slab + (((int64_t)ns * A + arm) * B) * H;
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = bw + acc_row(r, lane);
                if (row < B) out[(int64_t)row * H + j * 32 + l31] = gacc[j][r];
            }
#pragma unroll
        for (int c = 0; c < 4; ++c) glo[c] += __shfl_xor(glo[c], 32, 64);   // the two k halves of the same cell
        const int row = bw + l31;
        if (hh == 0 && row < B) *reinterpret_cast<float4*>(out + (int64_t)row * H + 96) = make_float4(glo[0], glo[1], glo[2], glo[3]);
    }
    se = wave_sum(se);
    const float mismf = wave_sum((float)mism);
    if (lane == 0) { red[wv * 2] = se; red[wv * 2 + 1] = mismf; }
    __syncthreads();
    if (tid == 0) {
        float* p = part + ((int64_t)arm * n11 + (int64_t)blockIdx.x * NS + ns) * 2;
        float s0 = 0.f, s1 = 0.f;
        for (int w = 0; w < 8; ++w) { s0 += red[w * 2]; s1 += red[w * 2 + 1]; }
        p[0] = s0;
        p[1] = s1;
    }
}

// =============================================================================================
// d(d10) = dZ11 W11: M = cells, N = H, K = genes.  Tile 128 x 128, K tile 32, wave tile 64 x 64.
// A = dZ11 (K contiguous, b128 fragment reads), B = W11 rows (h contiguous, b32 reads).
// grid (ceil(B/128), KS, A) -> slabs [KS][A][B][H]
// =============================================================================================
__global__ __launch_bounds__(256) void k_gd10_v2(const float* __restrict__ dz11, const float* __restrict__ params,
                                                 int64_t per_arm, int64_t w_off, float* __restrict__ slab, int A, int B,
                                                 int D, int H, int KS) {
    __shared__ __attribute__((aligned(16))) float As[128 * V2_LD];
    __shared__ __attribute__((aligned(16))) float Bs[32 * TN_LD];
    const int arm = blockIdx.z, ks = blockIdx.y, b0 = blockIdx.x * 128;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, wm = wv >> 1, wn = wv & 1;
    const int l31 = lane & 31, hh = lane >> 5;
    const float* Z = dz11 + (int64_t)arm * B * D;
    const float* W = params + (int64_t)arm * per_arm + w_off;     // [D, H]
    const int nkt = cdiv(D, 32);
    const int kt0 = (int)(((int64_t)ks * nkt) / KS), kt1 = (int)(((int64_t)(ks + 1) * nkt) / KS);
    const int r0 = tid >> 3, c4 = tid & 7;      // A staging
    const int rr = tid >> 5, bc4 = tid & 31;    // B staging
    const bool bok = bc4 * 4 < H;
    const float* pa[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) pa[i] = Z + (int64_t)min(b0 + r0 + 32 * i, B - 1) * D + c4 * 4;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = zero16();
    float4 ra4[4], rb4[4];
    auto load_tiles = [&](int kt) {
        const bool colok = kt * 32 + c4 * 4 < D;
        const int koff = colok ? kt * 32 : 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) ra4[i] = *reinterpret_cast<const float4*>(pa[i] + koff);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int j = kt * 32 + rr + 8 * i;
            const bool ok = bok && (j < D);
            rb4[i] = *reinterpret_cast<const float4*>(W + (int64_t)(ok ? j : 0) * H + (ok ? bc4 * 4 : 0));
            rb4[i] = sel4(ok, rb4[i]);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) ra4[i] = sel4(colok, ra4[i]);
    };
    if (kt0 < kt1) load_tiles(kt0);
    for (int kt = kt0; kt < kt1; ++kt) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            *reinterpret_cast<float4*>(&As[(r0 + 32 * i) * V2_LD + c4 * 4]) = ra4[i];
            *reinterpret_cast<float4*>(&Bs[(rr + 8 * i) * TN_LD + bc4 * 4]) = rb4[i];
        }
        __syncthreads();
        if (kt + 1 < kt1) load_tiles(kt + 1);
        const float* la = As + (wm * 64 + l31) * V2_LD + 4 * hh;
        const float* lb = Bs + (4 * hh) * TN_LD + wn * 64 + l31;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 a0 = *reinterpret_cast<const float4*>(la + 8 * g);
            const float4 a1 = *reinterpret_cast<const float4*>(la + 32 * V2_LD + 8 * g);
            const float* q = lb + (8 * g) * TN_LD;
            const float q00 = q[0], q01 = q[32], q10 = q[TN_LD], q11 = q[TN_LD + 32];
            const float q20 = q[2 * TN_LD], q21 = q[2 * TN_LD + 32], q30 = q[3 * TN_LD], q31 = q[3 * TN_LD + 32];
            acc[0][0] = mfma32(a0.x, q00, acc[0][0]); acc[0][1] = mfma32(a0.x, q01, acc[0][1]);
            acc[1][0] = mfma32(a1.x, q00, acc[1][0]); acc[1][1] = mfma32(a1.x, q01, acc[1][1]);
            acc[0][0] = mfma32(a0.y, q10, acc[0][0]); acc[0][1] = mfma32(a0.y, q11, acc[0][1]);
            acc[1][0] = mfma32(a1.y, q10, acc[1][0]); acc[1][1] = mfma32(a1.y, q11, acc[1][1]);
            acc[0][0] = mfma32(a0.z, q20, acc[0][0]); acc[0][1] = mfma32(a0.z, q21, acc[0][1]);
            acc[1][0] = mfma32(a1.z, q20, acc[1][0]); acc[1][1] = mfma32(a1.z, q21, acc[1][1]);
            acc[0][0] = mfma32(a0.w, q30, acc[0][0]); acc[0][1] = mfma32(a0.w, q31, acc[0][1]);
            acc[1][0] = mfma32(a1.w, q30, acc[1][0]); acc[1][1] = mfma32(a1.w, q31, acc[1][1]);
        }
        __syncthreads();
    }
    float* out = slab + (((int64_t)ks * A + arm) * B) * H;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = b0 + wm * 64 + i * 32 + acc_row(r, lane);
                const int col = wn * 64 + j * 32 + l31;
                if (row < B && col < H) out[(int64_t)row * H + col] = acc[i][j][r];
            }
}

// d(d10) for fc_dim = 100: as k_gd10_v2, but a wave owns 32 rows x (3 MFMA column tiles + 4 leftover columns done
// by plain FMAs on the A fragments it already holds) instead of 64 x 64 of a 128-wide tile with 28 padding columns
__global__ __launch_bounds__(256, 2) void k_gd10_v3(const float* __restrict__ dz11, const float* __restrict__ params,
                                                 int64_t per_arm, int64_t w_off, float* __restrict__ slab, int A, int B,
                                                 int D, int H, int KS) {
    __shared__ __attribute__((aligned(16))) float As[128 * V2_LD];
    __shared__ __attribute__((aligned(16))) float Bs[32 * TN_LD];
    const int arm = blockIdx.z, ks = blockIdx.y, b0 = blockIdx.x * 128;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int l31 = lane & 31, hh = lane >> 5;
    const float* Z = dz11 + (int64_t)arm * B * D;
    const float* W = params + (int64_t)arm * per_arm + w_off;     // [D, H]
    const int nkt = cdiv(D, 32);
    const int kt0 = (int)(((int64_t)ks * nkt) / KS), kt1 = (int)(((int64_t)(ks + 1) * nkt) / KS);
    const int r0 = tid >> 3, c4 = tid & 7;      // A staging
    const int rr = tid >> 5, bc4 = tid & 31;    // B staging
    const bool bok = bc4 * 4 < H;
    const float* pa[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) pa[i] = Z + (int64_t)min(b0 + r0 + 32 * i, B - 1) * D + c4 * 4;
    f32x16 acc[3] = {zero16(), zero16(), zero16()};   // columns [32 j, 32 j + 32), rows [32 wv, 32 wv + 32)
    float lo[4] = {0.f, 0.f, 0.f, 0.f};               // columns 96..99 of row (lane & 31): this lane's k's only
    float4 ra4[4], rb4[4];
    auto load_tiles = [&](int kt) {
        const bool colok = kt * 32 + c4 * 4 < D;
        const int koff = colok ? kt * 32 : 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) ra4[i] = *reinterpret_cast<const float4*>(pa[i] + koff);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int j = kt * 32 + rr + 8 * i;
            const bool ok = bok && (j < D);
            rb4[i] = *reinterpret_cast<const float4*>(W + (int64_t)(ok ? j : 0) * H + (ok ? bc4 * 4 : 0));
            rb4[i] = sel4(ok, rb4[i]);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) ra4[i] = sel4(colok, ra4[i]);
    };
    if (kt0 < kt1) load_tiles(kt0);
    for (int kt = kt0; kt < kt1; ++kt) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            *reinterpret_cast<float4*>(&As[(r0 + 32 * i) * V2_LD + c4 * 4]) = ra4[i];
            *reinterpret_cast<float4*>(&Bs[(rr + 8 * i) * TN_LD + bc4 * 4]) = rb4[i];
        }
        __syncthreads();
        if (kt + 1 < kt1) load_tiles(kt + 1);
        const float* la = As + (wv * 32 + l31) * V2_LD + 4 * hh;
        const float* lb = Bs + (4 * hh) * TN_LD + l31;
        const float* ll = Bs + (4 * hh) * TN_LD + 96;     // columns 96..99 of W11's rows: one address per half wave
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 a = *reinterpret_cast<const float4*>(la + 8 * g);
            const float* q = lb + (8 * g) * TN_LD;
            const float q00 = q[0], q01 = q[32], q02 = q[64];
            const float q10 = q[TN_LD], q11 = q[TN_LD + 32], q12 = q[TN_LD + 64];
            const float q20 = q[2 * TN_LD], q21 = q[2 * TN_LD + 32], q22 = q[2 * TN_LD + 64];
            const float q30 = q[3 * TN_LD], q31 = q[3 * TN_LD + 32], q32 = q[3 * TN_LD + 64];
            const float4 w0 = *reinterpret_cast<const float4*>(ll + (8 * g) * TN_LD);
            const float4 w1 = *reinterpret_cast<const float4*>(ll + (8 * g + 1) * TN_LD);
            const float4 w2 = *reinterpret_cast<const float4*>(ll + (8 * g + 2) * TN_LD);
            const float4 w3 = *reinterpret_cast<const float4*>(ll + (8 * g + 3) * TN_LD);
            acc[0] = mfma32(a.x, q00, acc[0]); acc[1] = mfma32(a.x, q01, acc[1]); acc[2] = mfma32(a.x, q02, acc[2]);
            acc[0] = mfma32(a.y, q10, acc[0]); acc[1] = mfma32(a.y, q11, acc[1]); acc[2] = mfma32(a.y, q12, acc[2]);
            acc[0] = mfma32(a.z, q20, acc[0]); acc[1] = mfma32(a.z, q21, acc[1]); acc[2] = mfma32(a.z, q22, acc[2]);
            acc[0] = mfma32(a.w, q30, acc[0]); acc[1] = mfma32(a.w, q31, acc[1]); acc[2] = mfma32(a.w, q32, acc[2]);
            // the four columns that do not fill a 32-wide MFMA tile: 16 VALU FMAs in the MFMAs' shadow
            lo[0] = fmaf(a.x, w0.x, lo[0]); lo[1] = fmaf(a.x, w0.y, lo[1]); lo[2] = fmaf(a.x, w0.z, lo[2]); lo[3] = fmaf(a.x, w0.w, lo[3]);
            lo[0] = fmaf(a.y, w1.x, lo[0]); lo[1] = fmaf(a.y, w1.y, lo[1]); lo[2] = fmaf(a.y, w1.z, lo[2]); lo[3] = fmaf(a.y, w1.w, lo[3]);
            lo[0] = fmaf(a.z, w2.x, lo[0]); lo[1] = fmaf(a.z, w2.y, lo[1]); lo[2] = fmaf(a.z, w2.z, lo[2]); lo[3] = fmaf(a.z, w2.w, lo[3]);
            lo[0] = fmaf(a.w, w3.x, lo[0]); lo[1] = fmaf(a.w, w3.y, lo[1]); lo[2] = fmaf(a.w, w3.z, lo[2]); lo[3] = fmaf(a.w, w3.w, lo[3]);
            __builtin_amdgcn_sched_barrier(0);   // keep the fragment registers of one k group at a time
        }
        __syncthreads();
    }
    float* out = slab + (((int64_t)ks * A + arm) * B) * H;
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = b0 + wv * 32 + acc_row(r, lane);
            if (row < B) out[(int64_t)row * H + j * 32 + l31] = acc[j][r];
        }
    // lanes l and l ^ 32 hold the two k halves of the same row
#pragma unroll
    for (int c = 0; c < 4; ++c) lo[c] += __shfl_xor(lo[c], 32, 64);
    {
        const int row = b0 + wv * 32 + l31;
        if (hh == 0 && row < B) *reinterpret_cast<float4*>(out + (int64_t)row * H + 96) = make_float4(lo[0], lo[1], lo[2], lo[3]);
    }
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
static inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

bool fast_path_ok(const Ctx& c, const float* params, const float* x, int64_t xs) {
    const mmvae_dims& d = c.d;
    return (d.D & 3) == 0 && (d.H & 3) == 0 && al16(params) && al16(x) && (xs & 3) == 0 && d.H >= 4 &&
           (int64_t)d.B * d.D < ((int64_t)1 << 30);
}

// true when forward used the fast fc11 kernels (d(d10) slab count ks_gd10) rather than the general fused kernel (ns_fc11)
bool fc11_split_path(const Ctx& c, const float* params, const float* x, int64_t xs) {
    return fast_path_ok(c, params, x, xs);
}

int launch_forward_zero(const Ctx& c, bool with_xbits, const mmvae_noise* nz) {
    if (with_xbits && c.h.training && c.h.x_drop > 0.f) {
        c.fwd_zeroed = true;   // k_make_xbits does it
        return launch_make_xbits(c, nz);
    }
    hipError_t e = hipMemsetAsync(c.ws + c.lay.fc11_part, 0, sizeof(float) * (size_t)c.fwd_zero_floats(), c.stream);
    if (e != hipSuccess) { set_error("memset: %s", hipGetErrorString(e)); return MMVAE_E_LAUNCH; }
    c.fwd_zeroed = true;
    return 0;
}

int launch_make_xbits(const Ctx& c, const mmvae_noise* nz) {
    if (!(c.h.training && c.h.x_drop > 0.f)) return 0;
    const mmvae_dims& d = c.d;
    NoiseDev nd = make_noise_dev(nz, c.h);
    const int wpr = cdiv(d.D, 32);
    const int wpt = nd.mode != 0 && nd.x_mlog2 <= 2 ? (int)(4u >> nd.x_mlog2) : 1;   // as in the kernel
    const int64_t n = (int64_t)d.A * d.B * cdiv(wpr, wpt);
    const int blocks = (int)imin64(4096, cdiv64(n, 256));
    hipLaunchKernelGGL(k_make_xbits, dim3(blocks), dim3(256), 0, c.stream, nd, d.A, d.B, d.D, wpr,
                       reinterpret_cast<uint32_t*>(c.ws + c.lay.xbits), c.ws + c.lay.fc11_part,
                       c.fwd_zeroed ? (int)(c.fwd_zero_floats() / 4) : 0);
    HIP_LAUNCH_CHECK("k_make_xbits");
    return 0;
}

int launch_fc1_fwd_fast(const Ctx& c, const float* params, const float* x, int64_t xs) {
    if (bf16_gemms(c, 1)) return launch_fc1_fwd_bf16(c, params, x, xs);
    const mmvae_dims& d = c.d;
    const bool use_mask = c.h.training && c.h.x_drop > 0.f;
    const int KS = c.lay.sp.ks_fc1;
    const int ablate = c.tune(MMVAE_TUNE_ABLATE);   // timing experiments only
    const int padlds = 0;
    dim3 grid(cdiv(d.B, 128), KS, d.A);
    const uint32_t* bits = reinterpret_cast<const uint32_t*>(c.ws + c.lay.xbits);
    if (d.H == 100) {
        if (use_mask)
            hipLaunchKernelGGL((k_fc1_fwd_v3<true>), grid, dim3(256), padlds, c.stream, x, xs, params, c.po.per_arm,
                               c.po.o[0], bits, cdiv(d.D, 32), c.ws + c.lay.fc1_slab, d.A, d.B, d.D, d.H, KS, ablate);
        else
            hipLaunchKernelGGL((k_fc1_fwd_v3<false>), grid, dim3(256), 0, c.stream, x, xs, params, c.po.per_arm,
                               c.po.o[0], bits, cdiv(d.D, 32), c.ws + c.lay.fc1_slab, d.A, d.B, d.D, d.H, KS, ablate);
        HIP_LAUNCH_CHECK("k_fc1_fwd_v3");
        return 0;
    }
    if (use_mask)
        hipLaunchKernelGGL((k_fc1_fwd_v2<true>), grid, dim3(256), padlds, c.stream, x, xs, params, c.po.per_arm, c.po.o[0],
                           bits, cdiv(d.D, 32), c.ws + c.lay.fc1_slab, d.A, d.B, d.D, d.H, KS, ablate);
    else
        hipLaunchKernelGGL((k_fc1_fwd_v2<false>), grid, dim3(256), 0, c.stream, x, xs, params, c.po.per_arm,
                           c.po.o[0], bits, cdiv(d.D, 32), c.ws + c.lay.fc1_slab, d.A, d.B, d.D, d.H, KS, ablate);
    HIP_LAUNCH_CHECK("k_fc1_fwd_v2");
    return 0;
}

int launch_fc11_fast(const Ctx& c, const float* params, const float* x, int64_t xs, float* x_rec, int need_grad,
                     int which /*bit0: x_rec/loss/dZ11 kernel, bit1: d(d10) GEMM*/) {
    // fp32x3: the fused train-step form (forward for gradients, no x_rec, fc_dim + 1 <= 112) has its own kernel; the other
    // forms of fc11 (x_rec wanted, forward only) run the fp32 matrix-instruction kernels below
    if (split3_gemms(c, 2) && need_grad && !x_rec && c.d.H + 1 <= 112 && !c.tune(MMVAE_TUNE_FC11_ZG_OFF) &&
        (int64_t)cdiv(c.d.B, 128) * c.lay.sp.ks_gd10 <= c.lay.n11)
        return launch_fc11_bf16(c, params, x, xs, x_rec, need_grad, which);
    if ((c.h.gemm_bf16 & 0xFF) == 1 && bf16_gemms(c)) return launch_fc11_bf16(c, params, x, xs, x_rec, need_grad, which);
    const mmvae_dims& d = c.d;
    const Layout& L = c.lay;
    const int ldk = rup(d.H, 8) + 4;
    const float coef = (float)(d.A > 1 ? d.A - 1 : 1) / (float)d.B;
    const int NS = L.sp.ns_fc11;
    // train step at fc_dim 100: d(d10) is folded into the fc11 kernel (k_fc11_zg), whose gene split count equals the
    // d(d10) kernel's so that the decoder backward sums the same number of slabs whichever forward ran
    const int zg_off = c.tune(MMVAE_TUNE_FC11_ZG_OFF);   // A/B timing
    const bool use_zg = need_grad && !x_rec && d.H == 100 && !zg_off &&
                        (int64_t)cdiv(d.B, 256) * L.sp.ks_gd10 <= L.n11;
    if ((which & 1) && use_zg) {
        hipError_t e = c.fwd_zeroed ? hipSuccess : hipMemsetAsync(c.ws + L.fc11_part, 0, sizeof(float) * 2 * (size_t)d.A * L.n11, c.stream);
        if (e != hipSuccess) { set_error("memset: %s", hipGetErrorString(e)); return MMVAE_E_LAUNCH; }
        const size_t shm = (size_t)(3 * 64 * ldk + 8 * 32 * ZG_LD + 16) * sizeof(float);
        // (152 KB of dynamic LDS: this runtime takes it without the hipFuncSetAttribute opt-in, and the library keeps no
        // per-process flag for having asked)
        hipLaunchKernelGGL((k_fc11_zg<13, true, true>), dim3(cdiv(d.B, 256), L.sp.ks_gd10, d.A), dim3(512), shm, c.stream,
                           c.ws + L.Dk[4], params, c.po.per_arm, c.po.o[26], c.po.o[27], x, xs, x_rec, c.ws + L.DZ11,
                           c.ws + L.fc11_part, L.n11, coef, need_grad, d.A, d.B, d.D, d.H, ldk, c.ws + L.GD10_slab);
        HIP_LAUNCH_CHECK("k_fc11_zg");
    }
    if (use_zg) return 0;
    if (which & 1) {
        // loss partials: the launch below fills a subset of the reserved slots
        hipError_t e = c.fwd_zeroed ? hipSuccess : hipMemsetAsync(c.ws + L.fc11_part, 0, sizeof(float) * 2 * (size_t)d.A * L.n11, c.stream);
        if (e != hipSuccess) { set_error("memset: %s", hipGetErrorString(e)); return MMVAE_E_LAUNCH; }
        const int ntall = cdiv(d.D, 64);
        const int kgv = rup(d.H, 8) / 8;
        {
            // one 512-thread workgroup per CU: split the gene range so that the grid fills the chip once
            const int nb = cdiv(d.B, 256);
            int nsz = max(1, min(min(256 / max(nb * d.A, 1), 16), ntall));
            while (nsz > 1 && (int64_t)nb * nsz > L.n11) --nsz;
            const size_t shm = (size_t)(2 * 64 * ldk + 16) * sizeof(float);
            dim3 grid(nb, nsz, d.A);
#define FZT_ARGS c.ws + L.Dk[4], params, c.po.per_arm, c.po.o[26], c.po.o[27], x, xs, x_rec, c.ws + L.DZ11,        \
                 c.ws + L.fc11_part, L.n11, coef, need_grad, d.A, d.B, d.D, d.H, ldk,                            \
                 reinterpret_cast<unsigned long long*>(c.ws + L.loss_scratch + 2048)
#define FZT_LAUNCH(KG, EX, BK, XR, AB) \
    hipLaunchKernelGGL((k_fc11_zt<KG, EX, BK, XR, AB>), grid, dim3(512), shm, c.stream, FZT_ARGS)
            const bool xr = x_rec != nullptr;
            // without x_rec the kernel always writes dZ11 (workspace), wanted or not: one variant fewer
            if (kgv == 13 && d.H == 100) {
                if (xr) FZT_LAUNCH(13, true, true, true, 0);
                else FZT_LAUNCH(13, true, true, false, 0);
            } else if (kgv == 16 && d.H == 128) {
                if (xr) FZT_LAUNCH(16, true, false, true, 0);
                else FZT_LAUNCH(16, true, false, false, 0);
            } else {
                if (xr) FZT_LAUNCH(16, false, false, true, 0);
                else FZT_LAUNCH(16, false, false, false, 0);
            }
#undef FZT_LAUNCH
#undef FZT_ARGS
            HIP_LAUNCH_CHECK("k_fc11_zt");
        }
    }
    if (need_grad && (which & 2)) {
        if (d.H == 100)
            hipLaunchKernelGGL(k_gd10_v3, dim3(cdiv(d.B, 128), L.sp.ks_gd10, d.A), dim3(256), 0, c.stream, c.ws + L.DZ11,
                               params, c.po.per_arm, c.po.o[26], c.ws + L.GD10_slab, d.A, d.B, d.D, d.H, L.sp.ks_gd10);
        else
        hipLaunchKernelGGL(k_gd10_v2, dim3(cdiv(d.B, 128), L.sp.ks_gd10, d.A), dim3(256), 0, c.stream, c.ws + L.DZ11,
                           params, c.po.per_arm, c.po.o[26], c.ws + L.GD10_slab, d.A, d.B, d.D, d.H, L.sp.ks_gd10);
        HIP_LAUNCH_CHECK("k_gd10_v2");
    }
    return 0;
}

int launch_dw_big_fast(const Ctx& c, const float* x, int64_t xs, int which) {
    if ((c.h.gemm_bf16 & 0xFF) == 2 && ((c.h.gemm_bf16 >> 8) & 12)) {     // diagnostics: one of the two on the fp32 matrix instruction
        int rc = 0;
        Ctx c0 = c;
        c0.h.gemm_bf16 = 0;
        if (which & 1) rc = split3_gemms(c, 4) ? launch_dw_big_bf16(c, x, xs, 1) : launch_dw_big_fast(c0, x, xs, 1);
        if (!rc && (which & 2)) rc = split3_gemms(c, 8) ? launch_dw_big_bf16(c, x, xs, 2) : launch_dw_big_fast(c0, x, xs, 2);
        return rc;
    }
    if (bf16_gemms(c)) return launch_dw_big_bf16(c, x, xs, which);
    const mmvae_dims& d = c.d;
    const Layout& L = c.lay;
    const bool use_mask = c.h.training && c.h.x_drop > 0.f;
    const int KS = L.sp.ks_dw;
    const uint32_t* bits = reinterpret_cast<const uint32_t*>(c.ws + L.xbits);
    const int wpr = cdiv(d.D, 32);
    if (which & 1) {   // dW1[h][d] = sum_b dZ1[b][h] x~[b][d]   -> slab [KS][A][H][D]
        const int tiles_n = cdiv(d.D, 128);
        dim3 grid(cdiv(d.H, 128) * tiles_n, KS, d.A);
        if (d.H == 100) {
            if (use_mask)
                hipLaunchKernelGGL((k_tn_v3m<true>), grid, dim3(256), 0, c.stream, c.ws + L.DZ[1],
                                   (int64_t)d.B * d.H, d.H, d.H, x, xs, d.D, d.D, bits, wpr, c.ws + L.dw1_slab,
                                   (int64_t)d.H * d.D, (int64_t)d.A * d.H * d.D, d.D, d.B, KS, tiles_n);
            else
                hipLaunchKernelGGL((k_tn_v3m<false>), grid, dim3(256), 0, c.stream, c.ws + L.DZ[1],
                                   (int64_t)d.B * d.H, d.H, d.H, x, xs, d.D, d.D, bits, wpr, c.ws + L.dw1_slab,
                                   (int64_t)d.H * d.D, (int64_t)d.A * d.H * d.D, d.D, d.B, KS, tiles_n);
            HIP_LAUNCH_CHECK("k_tn_v3m<dW1>");
        } else
        if (use_mask)
            hipLaunchKernelGGL((k_tn_v2<true, false>), grid, dim3(256), 0, c.stream, c.ws + L.DZ[1],
                               (int64_t)d.B * d.H, d.H, d.H, x, xs, d.D, d.D, bits, wpr, c.ws + L.dw1_slab,
                               (int64_t)d.H * d.D, (int64_t)d.A * d.H * d.D, d.D, d.B, KS, tiles_n);
        else
            hipLaunchKernelGGL((k_tn_v2<false, false>), grid, dim3(256), 0, c.stream, c.ws + L.DZ[1],
                               (int64_t)d.B * d.H, d.H, d.H, x, xs, d.D, d.D, bits, wpr, c.ws + L.dw1_slab,
                               (int64_t)d.H * d.D, (int64_t)d.A * d.H * d.D, d.D, d.B, KS, tiles_n);
        HIP_LAUNCH_CHECK("k_tn_v2<dW1>");
    }
    if (which & 2) {   // [dW11 | db11][j][h] = sum_b dZ11[b][j] [d10 | 1][b][h]   -> slab [KS][A][D][DW11_LD]
        const int tiles_n = cdiv(d.H + 1, 128);
        const int KS11 = L.sp.ks_dw11;
        dim3 grid(cdiv(d.D, 128) * tiles_n, KS11, d.A);
        if (d.H == 100)
            hipLaunchKernelGGL(k_tn_v3n, grid, dim3(256), 0, c.stream, c.ws + L.DZ11, (int64_t)d.B * d.D,
                               d.D, d.D, c.ws + L.Dk[4], (int64_t)d.B * d.H, d.H, d.H, bits, wpr, c.ws + L.dw11_slab,
                               (int64_t)d.D * DW11_LD, (int64_t)d.A * d.D * DW11_LD, DW11_LD, d.B, KS11, tiles_n);
        else
        hipLaunchKernelGGL((k_tn_v2<false, true>), grid, dim3(256), 0, c.stream, c.ws + L.DZ11, (int64_t)d.B * d.D,
                           d.D, d.D, c.ws + L.Dk[4], (int64_t)d.B * d.H, d.H, d.H, bits, wpr, c.ws + L.dw11_slab,
                           (int64_t)d.D * DW11_LD, (int64_t)d.A * d.D * DW11_LD, DW11_LD, d.B, KS11, tiles_n);
        HIP_LAUNCH_CHECK("k_tn_v2<dW11>");
    }
    return 0;
}

}  // namespace mmvae
