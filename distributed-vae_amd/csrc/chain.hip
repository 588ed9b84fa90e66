// Small-layer chains (every Linear whose dimensions are <= 128 x 256): a block of CHAIN_ROWS cells is
// carried through up to five layers inside one 4 x CHAIN_ROWS / 32-wave workgroup, activations staying in LDS, weights
// staged per layer.  MFMA fp32 32x32x2; wave (rt, ct) = (wave >> 2, wave & 3) owns rows [32 rt, 32 rt + 32) and
// output columns [32 ct, 32 ct + 32).
//
//   k_chain_fwd   encoder fc2..fc5 one layer per launch (BatchNorm needs the whole batch between
//                 layers: nn_model.py:265-268), decoder fc6..fc10 in one launch (:277-284)
//   k_chain_bwd   their autograd: dZ = G .* relu'(out); G_prev = dZ W; with the BatchNorm backward
//                 folded into the prologue and its batch sums emitted by the epilogue
//
// These launches are latency chains (load -> LDS -> 13 MFMAs -> store), not throughput kernels: what a workgroup
// does once -- recombining the batch statistics from the producer's per-workgroup partials, staging the layer's
// weights -- is shared by sixteen waves instead of four, and 128-row workgroups emit a quarter of the partials
// the next launch has to recombine (40 instead of 157 at B = 5000).
#include "common.hpp"
#include "couple.hpp"
#include <stdlib.h>
#include <type_traits>

namespace mmvae {

#define HIP_LAUNCH_CHECK(what)                                                        \
    do {                                                                              \
        hipError_t e_ = hipGetLastError();                                            \
        if (e_ != hipSuccess) {                                                       \
            set_error("%s: %s", what, hipGetErrorString(e_));                         \
            return MMVAE_E_LAUNCH;                                                    \
        }                                                                             \
    } while (0)

constexpr int CH_RT = CHAIN_ROWS / 32;     // 32-row tiles per workgroup
constexpr int CH_NT = 256 * CH_RT;         // threads per workgroup

struct FwdLayer {
    int64_t w_off, b_off;   // inside one arm's parameter segment
    int64_t out_off;        // workspace, [A,B,N]
    int K, N, act;          // act: 1 = ReLU, 0 = identity
    // eval-mode encoder chain: the next layer reads BatchNorm(out) with the statistics at these workspace offsets
    // ([A,N] each; -1: the next layer reads `out` as is).  The stored `out` stays un-normalised.
    int64_t obn_mean_off = -1, obn_rstd_off = -1;
    int pl_slot = -1;       // fp32x3 form: index of this layer's weight planes in Layout::pl_small
};
struct ChainFwdArgs {
    int nlayers;
    FwdLayer L[5];
    int64_t x_off;          // workspace, [A,B,K0]
    int K0;
    int64_t bn_mean_off, bn_rstd_off;   // [A,K0] or -1: input is BatchNorm(x)
    int64_t bn_part_off;                // >= 0 (training): [A][part_n][2][K0] partials of x to recombine here
    int part_n, part_rows;              // their count and the rows each covers
    int64_t run_mean_off, run_var_off, run_arm_stride;   // inside bn_running (updated by row block 0)
    int bn_idx;
    float bn_eps, bn_momentum;
    int64_t stats_part_off;             // [A][gridDim.x][2][N_last] or -1
    // accumulator sets (common.hpp acc_add; [A] sets each) that replace the two partial arrays: the input's batch sums
    // to read, the output's to add to; -1 = the partial arrays
    int64_t acc_in_off, acc_out_off;
    // >= 0 (fp32x3 engine, decoder chain): the launch also writes the bf16 slice planes of [out_last | 1] that fc11 and dW11
    // stage -- [A][3][planes_rows][128] bf16 at this workspace offset, zero outside [B][N_last + 1]
    int64_t planes_off;
    int planes_rows;
    int64_t wpl_off;        // fp32x3 form (k_chain_fwd<true>): workspace offset of Layout::pl_small
    int B, ld, wrows;
    int rows;               // cells per workgroup (<= CHAIN_ROWS, multiple of 8): Layout::chain_rows_fwd
    int nblk;               // row blocks per arm (the launch's grid.x, unless the launch has other roles beside the chain)
    int64_t per_arm;
    int ablate;   // timing experiments only (MMVAE_ABLATE_C)
    int64_t dbg_off;   // >= 0: workspace offset of a diagnostic stamp-counter block (bit 3 of ablate)
};

// stage W [N][K] (global, row-major) into LDS rows [0, rows_pad) x cols [0, cols_pad), zero padded.
// 8 threads per row (128-B segments), 128 rows per pass, four chunks in flight per thread, no per-element
// branches: all loads of a pass issue before the first LDS store waits.
template <bool VEC>
__device__ __forceinline__ void stage_w_t(float* Ws, int ld, const float* __restrict__ W, int N, int K,
                                          int rows_pad, int cols_pad) {
    const int c4n = cols_pad >> 2;
    const int part = threadIdx.x & 7, r0 = threadIdx.x >> 3;
    for (int cb = 0; cb < c4n; cb += 32) {
        for (int row = r0; row < rows_pad; row += CH_NT / 8) {
            float4 v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = ldg4_t<VEC>(W, K, row, (cb + part + 8 * j) * 4, N, K);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int c = cb + part + 8 * j;
                if (c < c4n) *reinterpret_cast<float4*>(&Ws[row * ld + c * 4]) = v[j];
            }
        }
    }
}
__device__ __forceinline__ void stage_w(float* Ws, int ld, const float* __restrict__ W, int N, int K,
                                        int rows_pad, int cols_pad) {
    const bool vec = (K & 3) == 0 && ((reinterpret_cast<uintptr_t>(W) & 15) == 0);
    if (vec) stage_w_t<true>(Ws, ld, W, N, K, rows_pad, cols_pad);
    else stage_w_t<false>(Ws, ld, W, N, K, rows_pad, cols_pad);
}

// Split form of stage_w for weights with K % 4 == 0 and a padded tile of at most 128 x 128: w_load only REQUESTS
// the tile (raw clamped loads, nothing touches the values), w_store masks it and writes it to LDS later.  The
// request for layer l+1 is issued before layer l's GEMM and lands under the GEMM and the epilogue, so a chain
// pays the weight latency once instead of once per layer.
constexpr int WQ_N = 4 * (128 / (CH_NT / 8));   // float4 per thread: 128 rows / (threads / 8 per row) passes x 4 chunks
__device__ __forceinline__ bool w_split_ok(const float* W, int K, int rows_pad, int cols_pad) {
    return (K & 3) == 0 && ((reinterpret_cast<uintptr_t>(W) & 15) == 0) && rows_pad <= 128 && cols_pad <= 128;
}
__device__ __forceinline__ void w_load(float4 (&wq)[WQ_N], const float* __restrict__ W, int N, int K) {
    const int part = threadIdx.x & 7, r0 = threadIdx.x >> 3;
#pragma unroll
    for (int p = 0; p < WQ_N / 4; ++p)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int row = r0 + (CH_NT / 8) * p, col = (part + 8 * j) * 4;
            const bool ok = row < N && col < K;
            wq[p * 4 + j] = *reinterpret_cast<const float4*>(W + (int64_t)(ok ? row : 0) * K + (ok ? col : 0));
        }
}
__device__ __forceinline__ void w_store(float* Ws, int ld, const float4 (&wq)[WQ_N], int N, int K, int rows_pad, int cols_pad) {
    const int part = threadIdx.x & 7, r0 = threadIdx.x >> 3, c4n = cols_pad >> 2;
#pragma unroll
    for (int p = 0; p < WQ_N / 4; ++p)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int row = r0 + (CH_NT / 8) * p, c = part + 8 * j;
            const bool ok = row < N && c * 4 < K;
            const float4 q = wq[p * 4 + j];
            const float4 v = make_float4(ok ? q.x : 0.f, ok ? q.y : 0.f, ok ? q.z : 0.f, ok ? q.w : 0.f);
            if (c < c4n && row < rows_pad) *reinterpret_cast<float4*>(&Ws[row * ld + c * 4]) = v;
        }
}

// ---- fp32x3 form of the chains' GEMMs (X3 = true; the fp32x3 engine of gemm_bf16.hip) ------------
// The 64 x N x K products of a chain launch are MFMA-throughput work for the two waves of a SIMD (52 dependent
// v_mfma_f32_32x32x2_f32 each: 3.2 us of a 13 us launch).  Here both operands live in LDS as three bf16 slice planes
// ([row][k], row pitch `ldp` dwords = K/2 rounded up to 8, + 4: conflict-free ds_read_b128) and a product is six
// v_mfma_f32_32x32x16_bf16 per 16 k -- 42 matrix instructions of 32 cycles instead of 52 of 64, with fp32-grade results.
// The weights come as planes from the step's k_presplit launch (Layout::pl_small: [A][slot][3][128][128] bf16, zero
// outside [N][K]); the activations are split where they are written to LDS (input tile, epilogue).
typedef __bf16 bf16x8c __attribute__((ext_vector_type(8)));
typedef unsigned u32x4c __attribute__((ext_vector_type(4)));
constexpr int PLS = 128 * 128;                     // bf16 elements of one global weight plane
constexpr int WP_N = 3 * ((128 * 16) / CH_NT);     // uint4 per thread: 3 planes x 128 rows x 16 sixteen-byte slots
__device__ __forceinline__ void wp_load(u32x4c (&wq)[WP_N], const unsigned short* __restrict__ Wg, int N, int K) {
    const int kc_n = rup(K, 16) >> 3;
#pragma unroll
    for (int pl = 0; pl < 3; ++pl)
#pragma unroll
        for (int i = 0; i < WP_N / 3; ++i) {
            const int idx = threadIdx.x + CH_NT * i, row = idx >> 4, kc = idx & 15;
            const bool ok = row < N && kc < kc_n;
            wq[pl * (WP_N / 3) + i] = *reinterpret_cast<const u32x4c*>(Wg + pl * PLS + (ok ? row : 0) * 128 + (ok ? kc : 0) * 8);
        }
}
__device__ __forceinline__ void wp_store(unsigned* Wp, int wpl, int ldp, const u32x4c (&wq)[WP_N], int N, int K) {
    const int kc_n = rup(K, 16) >> 3;
#pragma unroll
    for (int pl = 0; pl < 3; ++pl)
#pragma unroll
        for (int i = 0; i < WP_N / 3; ++i) {
            const int idx = threadIdx.x + CH_NT * i, row = idx >> 4, kc = idx & 15;
            if (row < N && kc < kc_n) *reinterpret_cast<u32x4c*>(Wp + pl * wpl + row * ldp + kc * 4) = wq[pl * (WP_N / 3) + i];
        }
}
// acc[32 x 32] += A[a_row0 + .][k] * B[b_row0 + .][k] over ksteps x 16 k, operands as three planes each (plane strides
// xpl / wpl dwords); the fragments of step s + 1 are requested before the MFMAs of step s
__device__ __forceinline__ void mma_nt_x3(f32x16& acc, const unsigned* Xp, int xpl, const unsigned* Wp, int wpl, int ldp,
                                          int a_row0, int b_row0, int ksteps) {
    const int lane = lane_id();
    const unsigned* pa = Xp + (a_row0 + (lane & 31)) * ldp + 4 * (lane >> 5);
    const unsigned* pb = Wp + (b_row0 + (lane & 31)) * ldp + 4 * (lane >> 5);
    bf16x8c a[3], b[3];
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) {
        a[pl] = __builtin_bit_cast(bf16x8c, *reinterpret_cast<const u32x4c*>(pa + pl * xpl));
        b[pl] = __builtin_bit_cast(bf16x8c, *reinterpret_cast<const u32x4c*>(pb + pl * wpl));
    }
    for (int st = 0; st < ksteps; ++st) {
        const int sn = (st + 1 < ksteps) ? st + 1 : st;
        bf16x8c an[3], bn[3];
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
            an[pl] = __builtin_bit_cast(bf16x8c, *reinterpret_cast<const u32x4c*>(pa + pl * xpl + 8 * sn));
            bn[pl] = __builtin_bit_cast(bf16x8c, *reinterpret_cast<const u32x4c*>(pb + pl * wpl + 8 * sn));
        }
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], acc, 0, 0, 0);   // smallest terms first
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], acc, 0, 0, 0);
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) { a[pl] = an[pl]; b[pl] = bn[pl]; }
    }
}

template <bool X3>
__device__ __forceinline__ void chain_fwd_body(const ChainFwdArgs& a_in, const float* __restrict__ params,
                                               float* __restrict__ ws, float* __restrict__ bn_running,
                                               int64_t* __restrict__ nbt) {
    // Copy the argument block into registers once.  Read in place, the kernarg segment may alias the
    // stores below as far as hipcc knows: it then re-loads fields after every store and waits vmcnt(0)
    // before each re-load, which serialises the epilogue's stores (measured: 47 % of the kernel).
    const ChainFwdArgs a = a_in;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    // fp32 form: Xs [CHAIN_ROWS][ld], Ws [wrows][ld] floats.  X3: three planes each, row pitch ld DWORDS (bf16 pairs)
    constexpr int NPL = X3 ? 3 : 1;
    float* Xs = smem;
    float* Ws = smem + NPL * CHAIN_ROWS * a.ld;
    float* mean_s = Ws + NPL * a.wrows * a.ld;      // [128]
    float* rstd_s = mean_s + 128;                   // [128]
    unsigned* const Xp = reinterpret_cast<unsigned*>(Xs);
    unsigned* const Wp = reinterpret_cast<unsigned*>(Ws);
    const int xpl = CHAIN_ROWS * a.ld, wpl = a.wrows * a.ld;     // dwords per plane (X3)
    const int arm = blockIdx.y, blk = blockIdx.x, b0 = blk * a.rows;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, rt = wv >> 2, ct = wv & 3;
    const int B = a.B, ld = a.ld;
    const int nvalid = min(a.rows, B - b0);
    const int Rlim = b0 + nvalid;            // rows of x beyond the block read as zero (the tile has CHAIN_ROWS rows)
    const float* P = params + (int64_t)arm * a.per_arm;
    const bool stamps = (a.ablate & 8) != 0 && a.dbg_off >= 0;
    unsigned long long ph[6] = {0, 0, 0, 0, 0, 0}, tprev = 0;
    auto stamp = [&](int i) {
        if (stamps) {
            __builtin_amdgcn_sched_barrier(0);
            const unsigned long long now = __builtin_amdgcn_s_memtime();
            __builtin_amdgcn_s_waitcnt(0xC07F);
            ph[i] += now - tprev;
            tprev = now;
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    if (stamps) { tprev = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_s_waitcnt(0xC07F); }

    // ---- requested before anything waits: the first layer's weights and the workgroup's input rows (with the batch sums
    //      below: one memory round trip in front of the first GEMM instead of three)
    float4 wq[X3 ? 1 : WQ_N];
    u32x4c wq3[X3 ? WP_N : 1];
    bool wq_valid = false;
    const unsigned short* const WPL = reinterpret_cast<const unsigned short*>(ws + (X3 ? a.wpl_off : 0)) + (int64_t)arm * PL_SMALL_SLOTS * 3 * PLS;
    {
        const FwdLayer L0 = a.L[0];
        if constexpr (X3) {
            wq_valid = true;
            wp_load(wq3, WPL + (int64_t)L0.pl_slot * 3 * PLS, L0.N, L0.K);
        } else {
            wq_valid = w_split_ok(P + L0.w_off, L0.K, rup(L0.N, 32), rup(L0.K, 8));
            if (wq_valid) w_load(wq, P + L0.w_off, L0.N, L0.K);
        }
    }
    const float* X = ws + a.x_off + (int64_t)arm * B * a.K0;
    const int xc4n = rup(a.K0, X3 ? 16 : 8) >> 2;
    const bool x_early = (a.K0 & 3) == 0 && xc4n <= 32;   // one pass of 16-byte loads covers the tile
    float4 xq[4];
    if (x_early) {
#pragma unroll
        for (int j = 0; j < 4; ++j) xq[j] = ldg4_t<true>(X, a.K0, b0 + (tid >> 3), ((tid & 7) + 8 * j) * 4, Rlim, a.K0);
    }
    // ---- statistics of the input's BatchNorm: recombined from the producer's partials (training) or
    //      the running buffers' values left in the workspace (eval); zero beyond K0
    if (a.bn_mean_off >= 0) {
        const int K0 = a.K0;
        float mean = 0.f, rstd = 0.f;
        if (a.bn_part_off >= 0) {
            float m2 = 0.f;
            unsigned long long st[5] = {0, 0, 0, 0, 0};
            if (a.acc_in_off >= 0) {
                if (stamps) { st[0] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_s_waitcnt(0xC07F); }
                if (tid < K0)
                    acc_mean_m2(reinterpret_cast<const long long*>(ws + a.acc_in_off) + (int64_t)arm * ACC_SET_I64, tid, B, mean, m2);
                if (stamps) {
                    asm volatile("" :: "v"(mean), "v"(m2));
                    st[1] = st[2] = st[3] = st[0];
                    st[4] = __builtin_amdgcn_s_memtime();
                    __builtin_amdgcn_s_waitcnt(0xC07F);
                }
            } else {
                stats_from_partials<CH_NT>(ws + a.bn_part_off + (int64_t)arm * a.part_n * 2 * K0, a.part_n, B, a.part_rows, K0,
                                           Ws, mean, m2, stamps ? st : nullptr);
            }
            if (stamps && lane == 0) {
                unsigned long long* dbg = reinterpret_cast<unsigned long long*>(ws + a.dbg_off);
                for (int i = 0; i < 4; ++i) atomicAdd(dbg + 8 + i, st[i + 1] - st[i]);
            }
            rstd = 1.0f / sqrtf(m2 / (float)B + a.bn_eps);
            if (blk == 0 && tid < K0) {
                ws[a.bn_mean_off + (int64_t)arm * K0 + tid] = mean;
                ws[a.bn_rstd_off + (int64_t)arm * K0 + tid] = rstd;
                if (bn_running) {
                    float* rm = bn_running + a.run_mean_off + arm * a.run_arm_stride;
                    float* rv = bn_running + a.run_var_off + arm * a.run_arm_stride;
                    rm[tid] = (1.f - a.bn_momentum) * rm[tid] + a.bn_momentum * mean;
                    rv[tid] = (1.f - a.bn_momentum) * rv[tid] + a.bn_momentum * (m2 / (float)max(B - 1, 1));
                }
                if (nbt && tid == 0) nbt[arm * MMVAE_N_BN + a.bn_idx] += 1;
            }
        } else if (tid < K0) {
            mean = ws[a.bn_mean_off + (int64_t)arm * K0 + tid];
            rstd = ws[a.bn_rstd_off + (int64_t)arm * K0 + tid];
        }
        if (tid < 128) { mean_s[tid] = tid < K0 ? mean : 0.f; rstd_s[tid] = tid < K0 ? rstd : 0.f; }
        lds_barrier();
    }
    // ---- input tile (optionally BatchNorm-normalised), zero padded to a multiple of 8 columns
    {
        const bool bn = a.bn_mean_off >= 0;
        const int c4n = xc4n;
        const bool vec = (a.K0 & 3) == 0;    // workspace regions are 256-B aligned, widths multiples of 4
        const int part = tid & 7, row = tid >> 3;      // 128 rows x 8 sixteen-byte parts
        auto stage_x = [&](auto tag) __attribute__((always_inline)) {
            constexpr bool V = decltype(tag)::value;
            for (int cb = 0; cb < c4n; cb += 32) {
                float4 v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = (V && x_early) ? xq[j] : ldg4_t<V>(X, a.K0, b0 + row, (cb + part + 8 * j) * 4, Rlim, a.K0);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int c = cb + part + 8 * j;
                    float4 o = v[j];
                    if (bn) {   // padded columns: (0 - 0) * 0 = 0; rows past the batch must stay zero
                        const bool rok = b0 + row < Rlim;
                        const int cc = min(c, 31) * 4;   // BatchNorm widths are <= 128
                        const float4 m4 = *reinterpret_cast<const float4*>(&mean_s[cc]);
                        const float4 r4 = *reinterpret_cast<const float4*>(&rstd_s[cc]);
                        o.x = rok ? (o.x - m4.x) * r4.x : 0.f;
                        o.y = rok ? (o.y - m4.y) * r4.y : 0.f;
                        o.z = rok ? (o.z - m4.z) * r4.z : 0.f;
                        o.w = rok ? (o.w - m4.w) * r4.w : 0.f;
                    }
                    if constexpr (X3) {
                        if (c < c4n) {
                            unsigned w0[3], w1[3];
                            split3(o.x, o.y, w0);
                            split3(o.z, o.w, w1);
#pragma unroll
                            for (int pl = 0; pl < 3; ++pl)
                                *reinterpret_cast<uint2*>(Xp + pl * xpl + row * ld + c * 2) = make_uint2(w0[pl], w1[pl]);
                        }
                    } else {
                        if (c < c4n) *reinterpret_cast<float4*>(&Xs[row * ld + c * 4]) = o;
                    }
                }
            }
        };
        if (vec) stage_x(VecTag{});
        else stage_x(ScalarTag{});
    }
    stamp(0);
    for (int l = 0; l < a.nlayers; ++l) {
        const FwdLayer Lr = a.L[l];
        const int K = Lr.K, N = Lr.N, KP = rup(K, 8), NPad = rup(N, 32);
        if constexpr (X3) wp_store(Wp, wpl, ld, wq3, N, K);
        else if (wq_valid) w_store(Ws, ld, wq, N, K, NPad, KP);      // requested during the previous layer
        else stage_w(Ws, ld, P + Lr.w_off, N, K, NPad, KP);
        const int col = ct * 32 + (lane & 31);
        const bool active = ct * 32 < NPad;
        const float bias = (active && col < N) ? P[Lr.b_off + col] : 0.f;   // requested before the barrier
        const bool obn = Lr.obn_mean_off >= 0;
        const float omean = (obn && active && col < N) ? ws[Lr.obn_mean_off + (int64_t)arm * N + col] : 0.f;
        const float orstd = (obn && active && col < N) ? ws[Lr.obn_rstd_off + (int64_t)arm * N + col] : 0.f;
        stamp(1);
        lds_barrier();
        stamp(2);
        wq_valid = false;
        if (l + 1 < a.nlayers) {   // next layer's weights travel while this layer computes
            const FwdLayer Ln = a.L[l + 1];
            if constexpr (X3) {
                wp_load(wq3, WPL + (int64_t)Ln.pl_slot * 3 * PLS, Ln.N, Ln.K);
            } else {
                wq_valid = w_split_ok(P + Ln.w_off, Ln.K, rup(Ln.N, 32), rup(Ln.K, 8));
                if (wq_valid) w_load(wq, P + Ln.w_off, Ln.N, Ln.K);
            }
        }
        f32x16 acc = zero16();
        if constexpr (X3) {
            if (active && !(a.ablate & 1)) mma_nt_x3(acc, Xp, xpl, Wp, wpl, ld, rt * 32, ct * 32, rup(K, 16) >> 4);
        } else {
            if (active && !(a.ablate & 1)) mma_nt(acc, Xs, ld, rt * 32, Ws, ld, ct * 32, KP / 8);
        }
        if (stamps) asm volatile("" :: "v"(acc[0]));
        stamp(3);
        lds_barrier();   // every wave has finished reading Xs / Ws
        stamp(2);
        const bool last = (l + 1 == a.nlayers);
        const bool need_x = !last || a.planes_off >= 0;
        float vals[16];
        if (active) {
            float* out = ws + Lr.out_off + (int64_t)arm * B * N;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = rt * 32 + acc_row(r, lane);
                float v = 0.f;
                if (col < N && row < nvalid) {
                    v = acc[r] + bias;
                    if (Lr.act) v = relu_keep_nan(v);
                    if (!(a.ablate & 4)) out[(int64_t)(b0 + row) * N + col] = v;
                }
                vals[r] = v;
                // next layer's input: zero beyond N (up to the next multiple of 8) and beyond nvalid
                const float xin = obn ? ((col < N && row < nvalid) ? (v - omean) * orstd : 0.f) : v;
                if constexpr (X3) {
                    // the even lane of a pair takes its neighbour's value (quad_perm [1,1,3,3]) and writes whole dwords: 16-bit
                    // LDS writes of both lanes doubled the epilogue's time.  Nothing reads the tile behind a single layer.
                    const float nbv = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, xin), 0xF5, 0xF, 0xF, false));
                    if (need_x && !(lane & 1) && col < rup(N, 16)) {
                        unsigned w3[3];
                        split3(xin, nbv, w3);
#pragma unroll
                        for (int pl = 0; pl < 3; ++pl) Xp[pl * xpl + row * ld + (col >> 1)] = w3[pl];
                    }
                } else {
                    if (col < rup(N, 8)) Xs[row * ld + col] = xin;
                }
            }
        }
        if (last && a.stats_part_off >= 0) {
            // per-workgroup column mean and M2 over the nvalid rows: each wave over its own (up to 32) rows --
            // lanes l and l^32 share a column --, then Chan's update over the four row tiles through LDS
            // (Ws is free: every wave is past the GEMM barrier)
            const int nv = max(0, min(32, nvalid - rt * 32));
            if (active) {
                float s = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) s += vals[r];
                s += __shfl_xor(s, 32, 64);
                const float mean = nv > 0 ? s / (float)nv : 0.f;
                float m2 = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = rt * 32 + acc_row(r, lane);
                    if (row < nvalid) { const float dl = vals[r] - mean; m2 += dl * dl; }
                }
                m2 += __shfl_xor(m2, 32, 64);
                if (lane < 32) { Ws[(rt * 2 + 0) * 128 + col] = mean; Ws[(rt * 2 + 1) * 128 + col] = m2; }
            }
            lds_barrier();
            if (rt == 0 && active && lane < 32 && col < N) {
                float n = 0.f, mu = 0.f, M2 = 0.f;
#pragma unroll
                for (int k = 0; k < CH_RT; ++k) {
                    const float nb = (float)max(0, min(32, nvalid - k * 32));
                    if (nb > 0.f) {
                        const float nn = n + nb, dl = Ws[(k * 2 + 0) * 128 + col] - mu;
                        mu += dl * (nb / nn);
                        M2 += Ws[(k * 2 + 1) * 128 + col] + dl * dl * (n * nb / nn);
                        n = nn;
                    }
                }
                if (a.acc_out_off >= 0) {
                    acc_add_stats(reinterpret_cast<long long*>(ws + a.acc_out_off) + (int64_t)arm * ACC_SET_I64, col, n, mu, M2);
                } else {
                    float* p = ws + a.stats_part_off + (((int64_t)arm * a.nblk + blk) * 2) * N;
                    p[col] = mu;
                    p[N + col] = M2;
                }
            }
        }
        stamp(4);
        lds_barrier();
        stamp(2);
    }
    if (a.planes_off >= 0) {
        // slice planes of [d10 | 1]: Xs holds the last layer's output (zero beyond nvalid rows); pairs of columns, lanes
        // along a row -> 256-byte store segments.  The last row block also writes the zero rows up to planes_rows.
        const int N = a.L[a.nlayers - 1].N;
        const int64_t plane = (int64_t)a.planes_rows * 128;
        unsigned short* pl = reinterpret_cast<unsigned short*>(ws + a.planes_off) + (int64_t)arm * 3 * plane;
        const int rows_here = blk == a.nblk - 1 ? a.planes_rows - b0 : a.rows;
        for (int i = tid; i < rows_here * 64; i += CH_NT) {
            const int r = i >> 6, c = (i & 63) * 2;
            unsigned w[3] = {0u, 0u, 0u};
            if constexpr (X3) {
                // Xp already holds the slices of the output (zero from N to the next multiple of 16); add the ones column
                if (r < nvalid) {
                    if (c < rup(N, 16)) {
#pragma unroll
                        for (int p = 0; p < 3; ++p) w[p] = Xp[p * xpl + r * ld + (c >> 1)];
                    }
                    if (c == N) w[0] |= 0x3F80u;              // bf16(1.0) in the low half (N even)
                    if (c + 1 == N) w[0] |= 0x3F800000u;      // ... in the high half (N odd)
                }
            } else {
                float v0 = 0.f, v1 = 0.f;
                if (r < nvalid) {
                    v0 = c < N ? Xs[r * ld + c] : (c == N ? 1.f : 0.f);
                    v1 = c + 1 < N ? Xs[r * ld + c + 1] : (c + 1 == N ? 1.f : 0.f);
                }
                split3(v0, v1, w);
            }
#pragma unroll
            for (int p = 0; p < 3; ++p) *reinterpret_cast<unsigned*>(pl + p * plane + (int64_t)(b0 + r) * 128 + c) = w[p];
        }
    }
    if (stamps && lane == 0) {
        unsigned long long* dbg = reinterpret_cast<unsigned long long*>(ws + a.dbg_off);
        for (int i = 0; i < 5; ++i) atomicAdd(dbg + i, ph[i]);
        atomicAdd(dbg + 5, 1ull);
    }
}
template <bool X3>
__global__ __launch_bounds__(CH_NT) void k_chain_fwd(const ChainFwdArgs a_in, const float* __restrict__ params,
                                                   float* __restrict__ ws, float* __restrict__ bn_running,
                                                   int64_t* __restrict__ nbt) {
    chain_fwd_body<X3>(a_in, params, ws, bn_running, nbt);
}

// The decoder chain of the fused train step with the coupling terms as a second ROLE of the same launch (fp32x3 form only):
// grid (max(row blocks, ceil(coupling blocks / 2)), A + 1); blockIdx.y < A: the chain's row block blockIdx.x of that arm;
// blockIdx.y == A: the two 256-thread halves of the workgroup take coupling row blocks 2 blockIdx.x and 2 blockIdx.x + 1
// (couple.hpp).  The chain's 158 workgroups (A = 2) leave 98 CUs idle; the coupling's 79 run there, and the T sums are
// complete three launches before the latent backward reads them -- no fork to the side stream behind the latent forward, no
// join in front of the latent backward, no fill of the T set (the step's head launch zeroes it with the forward sets).
struct CoupleRoleArgs {
    int nchain, ncouple;         // row blocks of the chain / 32-cell blocks of the coupling
    int B, C;
    float eps, lam;
    int64_t cc_off, csmp_off, c_mean_off, c_iv_off, couple_part_off, c_acc_off, t_acc_off;   // workspace offsets (floats)
};
template <int AT>
__device__ __forceinline__ void couple_role(const CoupleRoleArgs& cr_in, float* __restrict__ ws) {
    // (not inlined: the chain's code keeps the registers and the schedule it has as a kernel of its own)
    const CoupleRoleArgs cr = cr_in;
    if (2 * (int)blockIdx.x >= cr.ncouple) return;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int half = threadIdx.x >> 8, t256 = threadIdx.x & 255;
    constexpr int HALF = 4 * AT * CPL * 64 + AT * CPL * 64 + 8;      // shT, sh_iv, sh_red
    float* sh = smem + half * HALF;
    const int blk = 2 * blockIdx.x + half;
    couple_body<AT>(blk, blk < cr.ncouple, t256, cr.B, cr.C, cr.eps, cr.lam, ws + cr.cc_off, ws + cr.csmp_off, nullptr, 0,
                    reinterpret_cast<const long long*>(ws + cr.c_acc_off), ws + cr.c_mean_off, ws + cr.c_iv_off,
                    ws + cr.couple_part_off, nullptr, reinterpret_cast<long long*>(ws + cr.t_acc_off), sh,
                    sh + 4 * AT * CPL * 64 + AT * CPL * 64, sh + 4 * AT * CPL * 64, nullptr);
}
template <int AT>
__global__ __launch_bounds__(CH_NT) void k_chain_fwd_couple(const ChainFwdArgs a_in, const CoupleRoleArgs cr_in,
                                                          const float* __restrict__ params, float* __restrict__ ws) {
    if ((int)blockIdx.y == AT) {
        couple_role<AT>(cr_in, ws);
        return;
    }
    if ((int)blockIdx.x >= cr_in.nchain) return;
    chain_fwd_body<true>(a_in, params, ws, nullptr, nullptr);
}

// ---------------------------------------------------------------------------------------------
struct BwdLayer {
    int64_t w_off;      // [N][K]
    int64_t dz_off;     // workspace [A,B,N]: dZ of this layer (stored)
    int64_t act_off;    // workspace [A,B,N]: saved output of this layer (ReLU mask), -1 = identity
    int K, N;
    int pl_slot = -1;   // fp32x3 form: index of this layer's TRANSPOSED weight planes ([K][N]) in Layout::pl_small
};
struct ChainBwdArgs {
    int nlayers;
    BwdLayer L[5];          // backward order
    int64_t g_off;          // [nslab][A,B,N0] gradient w.r.t. the output of L[0] (after its BN if any)
    int nslab;
    int64_t slab_stride;
    int64_t bnb_part_off;   // [A][bnb_n][2][N0] per-workgroup sums (sum G, sum G*xhat), recombined here, or -1:
    int bnb_n;              // BN backward prologue
    int64_t bn_mean_off, bn_rstd_off;   // statistics of L[0]'s output, [A,N0]
    int64_t gout_off;       // [A,B,Klast]
    int64_t part_off;       // [A][gridDim.x][2][Klast] or -1: sums of gout and gout*xhat_prev
    int64_t rprev_off, rprev_mean_off, rprev_rstd_off;   // the BN input that produced the chain input
    // accumulator sets that replace bnb_part (read) and part (added to); -1 = the partial arrays.  zero_off / zero_n4:
    // float4s this launch zeroes first (the first launch of a backward pass: all backward accumulator sets)
    int64_t acc_in_off, acc_out_off, zero_off;
    int zero_n4;
    int64_t wpl_off;        // fp32x3 form (k_chain_bwd<true>): workspace offset of Layout::pl_small
    int B, ld, wrows;
    int64_t per_arm;
};

// X3: the fp32x3 form (see k_chain_fwd): the gradient tile as three planes [row][n], the weights TRANSPOSED as three
// planes [k][n] (slots 9.. of Layout::pl_small), so that G_prev[row][k] = sum_n dZ[row][n] W[n][k] is the same
// both-operands-k-contiguous product as the forward one, with the contraction over n.
// BatchNorm backward of one element and the sum of G * xhat: one definition, contraction off, so that every kernel that
// evaluates them (one launch per layer, one launch per chain: picked-up rows and register-resident rows) rounds alike
__device__ __forceinline__ float bn_bwd_val(float g, float av, float mm, float rr, float s1, float s2, float invB) {
#pragma clang fp contract(off)
    const float xh = (av - mm) * rr;
    const float t = g - s1 * invB;
    const float u = xh * (s2 * invB);
    return rr * (t - u);
}
__device__ __forceinline__ float bn_bwd_xhat_acc(float g, float rprev, float mu, float rs, float sum) {
#pragma clang fp contract(off)
    const float xh = (rprev - mu) * rs;
    const float p = g * xh;
    return sum + p;
}

template <bool X3>
__global__ __launch_bounds__(CH_NT) void k_chain_bwd(const ChainBwdArgs a_in, const float* __restrict__ params,
                                                   float* __restrict__ ws) {
    const ChainBwdArgs a = a_in;   // see k_chain_fwd: keep the argument block out of memory
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int NPL = X3 ? 3 : 1;
    float* Gs = smem;                              // [CHAIN_ROWS][ld]           (X3: three planes, pitch ld dwords)
    float* Ws = smem + NPL * CHAIN_ROWS * a.ld;    // [wrows][ld]  rows = n (output features), cols = k (input features)
    float* sums_s = Ws + NPL * a.wrows * a.ld;     // [2][128]: sum G, sum G*xhat over the batch
    unsigned* const Gp = reinterpret_cast<unsigned*>(Gs);
    unsigned* const Wp = reinterpret_cast<unsigned*>(Ws);
    const int xpl = CHAIN_ROWS * a.ld, wpl = a.wrows * a.ld;     // dwords per plane (X3)
    const int arm = blockIdx.y, blk = blockIdx.x, b0 = blk * CHAIN_ROWS;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, rt = wv >> 2, ct = wv & 3;
    const int B = a.B, ld = a.ld;
    const int nvalid = min(CHAIN_ROWS, B - b0);
    const float* P = params + (int64_t)arm * a.per_arm;

    if (a.zero_n4 > 0) grid_zero(ws + a.zero_off, a.zero_n4);
    if (a.bnb_part_off >= 0 && a.acc_in_off >= 0) {
        const int N = a.L[0].N;
        if (tid < 128) {
            double s1 = 0.0, s2 = 0.0;
            if (tid < N) acc_get(reinterpret_cast<const long long*>(ws + a.acc_in_off) + (int64_t)arm * ACC_SET_I64, tid, s1, s2);
            sums_s[tid] = (float)s1;
            sums_s[128 + tid] = (float)s2;
        }
        lds_barrier();
    } else if (a.bnb_part_off >= 0) {
        const int N = a.L[0].N;
        const float r = sums_from_partials<CH_NT>(ws + a.bnb_part_off + (int64_t)arm * a.bnb_n * 2 * N, a.bnb_n, 2 * N,
                                                  reinterpret_cast<double*>(Ws));
        // thread t < 2N holds sum t of [sum G | sum G*xhat]; spread to [2][128], zero beyond N
        if (tid < 256) sums_s[tid] = 0.f;
        lds_barrier();
        if (tid < 2 * N) sums_s[tid < N ? tid : 128 + tid - N] = r;
        lds_barrier();
    }
    // ---- prologue: dZ of the first (= last forward) layer
    {
        const BwdLayer L0 = a.L[0];
        const int N = L0.N, c4n = rup(N, X3 ? 16 : 8) >> 2;
        const bool has_act = L0.act_off >= 0, bnb = a.bnb_part_off >= 0;
        const float* G = ws + a.g_off + (int64_t)arm * B * N;
        const float* act = has_act ? ws + L0.act_off + (int64_t)arm * B * N : G;
        float* dz = ws + L0.dz_off + (int64_t)arm * B * N;
        const float* mu = bnb ? ws + a.bn_mean_off + (int64_t)arm * N : G;
        const float* rs = bnb ? ws + a.bn_rstd_off + (int64_t)arm * N : G;
        const float invB = 1.f / (float)B;
        const bool vec = (N & 3) == 0;
        const int part = tid & 7, row = tid >> 3;      // 128 rows x 8 sixteen-byte parts
        const bool rok = row < nvalid;
        auto stage_g = [&](auto tag) __attribute__((always_inline)) {
            constexpr bool V = decltype(tag)::value;
            for (int cb = 0; cb < c4n; cb += 32) {
                float4 gq[4], avq[4], mmq[4], rrq[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int col = (cb + part + 8 * j) * 4;
                    // split-K slabs of the producing GEMM: all (up to 16) requested before the first add -- a loop
                    // with a running sum waits for one memory latency per slab
                    float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
                    // (chunks of 4 for up to four slabs, of 16 beyond: clamped duplicates of the last slab cost a pass through
                    // the vector-memory pipe each -- 16 loads issued for the benchmark's 3 slabs were 5 x the needed ones)
                    auto add_slabs = [&](auto chunk) __attribute__((always_inline)) {
                        constexpr int CHK = decltype(chunk)::value;
                        for (int s0 = 0; s0 < a.nslab; s0 += CHK) {
                            float4 t[CHK];
#pragma unroll
                            for (int sl = 0; sl < CHK; ++sl)
                                t[sl] = ldg4_t<V>(G + (int64_t)min(s0 + sl, a.nslab - 1) * a.slab_stride, N, b0 + row, col, B, N);
#pragma unroll
                            for (int sl = 0; sl < CHK; ++sl) {
                                const bool on = s0 + sl < a.nslab;
                                g.x += on ? t[sl].x : 0.f; g.y += on ? t[sl].y : 0.f; g.z += on ? t[sl].z : 0.f; g.w += on ? t[sl].w : 0.f;
                            }
                        }
                    };
                    if (a.nslab == 1) g = ldg4_t<V>(G, N, b0 + row, col, B, N);
                    else if (a.nslab <= 4) add_slabs(std::integral_constant<int, 4>{});
                    else add_slabs(std::integral_constant<int, 16>{});
                    gq[j] = g;
                    avq[j] = ldg4_t<V>(act, N, b0 + row, col, B, N);
                    mmq[j] = ldg4_t<V>(mu, 0, 0, col, 1, N);
                    rrq[j] = ldg4_t<V>(rs, 0, 0, col, 1, N);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int c = cb + part + 8 * j, col = c * 4;
                    float4 g = gq[j];
                    const float4 av = avq[j];
                    if (bnb) {
                        const int cs = min(c, 31) * 4;   // BatchNorm widths are <= 128
                        const float4 m1 = *reinterpret_cast<const float4*>(&sums_s[cs]);
                        const float4 m2 = *reinterpret_cast<const float4*>(&sums_s[128 + cs]);
                        const float4 mm = mmq[j], rr = rrq[j];
                        g.x = bn_bwd_val(g.x, av.x, mm.x, rr.x, m1.x, m2.x, invB);
                        g.y = bn_bwd_val(g.y, av.y, mm.y, rr.y, m1.y, m2.y, invB);
                        g.z = bn_bwd_val(g.z, av.z, mm.z, rr.z, m1.z, m2.z, invB);
                        g.w = bn_bwd_val(g.w, av.w, mm.w, rr.w, m1.w, m2.w, invB);
                    }
                    float4 v;
                    v.x = (rok && col < N && (!has_act || av.x > 0.f)) ? g.x : 0.f;
                    v.y = (rok && col + 1 < N && (!has_act || av.y > 0.f)) ? g.y : 0.f;
                    v.z = (rok && col + 2 < N && (!has_act || av.z > 0.f)) ? g.z : 0.f;
                    v.w = (rok && col + 3 < N && (!has_act || av.w > 0.f)) ? g.w : 0.f;
                    if (c < c4n) {
                        if constexpr (X3) {
                            unsigned w0[3], w1[3];
                            split3(v.x, v.y, w0);
                            split3(v.z, v.w, w1);
#pragma unroll
                            for (int pl = 0; pl < 3; ++pl)
                                *reinterpret_cast<uint2*>(Gp + pl * xpl + row * ld + c * 2) = make_uint2(w0[pl], w1[pl]);
                        } else {
                            *reinterpret_cast<float4*>(&Gs[row * ld + col]) = v;
                        }
                        if (rok) {
                            float* o = dz + (int64_t)(b0 + row) * N + col;
                            if (V && col + 3 < N) *reinterpret_cast<float4*>(o) = v;
                            else {
                                if (col < N) o[0] = v.x;
                                if (col + 1 < N) o[1] = v.y;
                                if (col + 2 < N) o[2] = v.z;
                                if (col + 3 < N) o[3] = v.w;
                            }
                        }
                    }
                }
            }
        };
        if (vec) stage_g(VecTag{});
        else stage_g(ScalarTag{});
    }
    float4 wq[X3 ? 1 : WQ_N];
    u32x4c wq3[X3 ? WP_N : 1];
    bool wq_valid = false;
    const unsigned short* const WPL = reinterpret_cast<const unsigned short*>(ws + (X3 ? a.wpl_off : 0)) + (int64_t)arm * PL_SMALL_SLOTS * 3 * PLS;
    if constexpr (X3) wp_load(wq3, WPL + (int64_t)a.L[0].pl_slot * 3 * PLS, a.L[0].K, a.L[0].N);   // [K][N]: rows k, contraction n
    for (int l = 0; l < a.nlayers; ++l) {
        const BwdLayer Lr = a.L[l];
        const int K = Lr.K, N = Lr.N, NP8 = rup(N, 8), KPad = rup(K, 32);
        if constexpr (X3) wp_store(Wp, wpl, ld, wq3, K, N);
        else if (wq_valid) w_store(Ws, ld, wq, N, K, NP8, KPad);      // requested during the previous layer
        else stage_w(Ws, ld, P + Lr.w_off, N, K, NP8, KPad);
        lds_barrier();
        wq_valid = false;
        if (l + 1 < a.nlayers) {   // next layer's weights travel while this layer computes
            const BwdLayer Ln = a.L[l + 1];
            if constexpr (X3) {
                wp_load(wq3, WPL + (int64_t)Ln.pl_slot * 3 * PLS, Ln.K, Ln.N);
            } else {
                wq_valid = w_split_ok(P + Ln.w_off, Ln.K, rup(Ln.N, 8), rup(Ln.K, 32));
                if (wq_valid) w_load(wq, P + Ln.w_off, Ln.N, Ln.K);
            }
        }
        // the epilogue's global operands -- the saved activation that gates ReLU', or the BatchNorm input of the last
        // layer -- are requested here, in front of the GEMM: fetched after it they cost one exposed memory latency per
        // layer (five per decoder launch)
        const bool last = (l + 1 == a.nlayers);
        constexpr int NTI = X3 ? 1 : 2;   // X3: every width <= 128, one column tile per wave
        float pre[NTI][16];
        {
            const float* src = nullptr;
            if (!last) {
                const BwdLayer Ln = a.L[l + 1];
                src = Ln.act_off >= 0 ? ws + Ln.act_off + (int64_t)arm * B * K : nullptr;
            } else if (a.part_off >= 0) {
                src = ws + a.rprev_off + (int64_t)arm * B * K;
            }
#pragma unroll
            for (int ti = 0; ti < NTI; ++ti) {
                const int col = (ct + 4 * ti) * 32 + (lane & 31);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = rt * 32 + acc_row(r, lane);
                    const bool ok = src && (ct + 4 * ti) * 32 < KPad && col < K && row < nvalid;
                    pre[ti][r] = ok ? src[(int64_t)(b0 + row) * K + col] : (last ? 0.f : 1.f);
                }
            }
        }
        // input width K may reach 255 (fc6: K = C + S): up to 8 column tiles, two per wave
        f32x16 accs[NTI];
#pragma unroll
        for (int ti = 0; ti < NTI; ++ti) accs[ti] = zero16();
#pragma unroll
        for (int ti = 0; ti < NTI; ++ti) {
            const int cti = ct + 4 * ti;
            if constexpr (X3) {
                if (cti * 32 < KPad) mma_nt_x3(accs[ti], Gp, xpl, Wp, wpl, ld, rt * 32, cti * 32, rup(N, 16) >> 4);
            } else {
                if (cti * 32 < KPad) mma_nn(accs[ti], Gs, ld, rt * 32, Ws, ld, cti * 32, NP8 / 8);
            }
        }
        lds_barrier();
        float ps1[2] = {0.f, 0.f}, ps2[2] = {0.f, 0.f};
#pragma unroll
        for (int ti = 0; ti < NTI; ++ti) {
            const int cti = ct + 4 * ti;
            if (cti * 32 >= KPad) continue;
            const f32x16 acc = accs[ti];
            const int col = cti * 32 + (lane & 31);
            if (!last) {
                const BwdLayer Ln = a.L[l + 1];   // its N == this K
                float* dz = ws + Ln.dz_off + (int64_t)arm * B * K;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = rt * 32 + acc_row(r, lane);
                    const bool ok = col < K && row < nvalid;
                    const float v = (ok && pre[ti][r] > 0.f) ? acc[r] : 0.f;
                    if (ok) dz[(int64_t)(b0 + row) * K + col] = v;
                    if constexpr (X3) {   // pairs of columns through the even lane (see k_chain_fwd)
                        const float nbv = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xF5, 0xF, 0xF, false));
                        if (!(lane & 1) && col < rup(K, 16)) {
                            unsigned w3[3];
                            split3(v, nbv, w3);
#pragma unroll
                            for (int pl = 0; pl < 3; ++pl) Gp[pl * xpl + row * ld + (col >> 1)] = w3[pl];
                        }
                    } else {
                        Gs[row * ld + col] = v;
                    }
                }
            } else {
                float* go = ws + a.gout_off + (int64_t)arm * B * K;
                float s1 = 0.f, s2 = 0.f;
                const bool want = a.part_off >= 0;
                float mu = 0.f, rs = 0.f;
                if (want && col < K) {
                    mu = ws[a.rprev_mean_off + (int64_t)arm * K + col];
                    rs = ws[a.rprev_rstd_off + (int64_t)arm * K + col];
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = rt * 32 + acc_row(r, lane);
                    if (col < K && row < nvalid) {
                        go[(int64_t)(b0 + row) * K + col] = acc[r];
                        s1 += acc[r];
                        s2 = bn_bwd_xhat_acc(acc[r], pre[ti][r], mu, rs, s2);
                    }
                }
                if (want) {
                    s1 += __shfl_xor(s1, 32, 64);
                    s2 += __shfl_xor(s2, 32, 64);
                    ps1[ti] = s1;
                    ps2[ti] = s2;
                }
            }
        }
        if (last && a.part_off >= 0) {
            // per-workgroup sums over the row tiles through LDS (Ws is free after the GEMM barrier): [rt][2][256]
#pragma unroll
            for (int ti = 0; ti < NTI; ++ti) {
                const int col = (ct + 4 * ti) * 32 + (lane & 31);
                if (lane < 32) { Ws[(rt * 2 + 0) * 256 + col] = ps1[ti]; Ws[(rt * 2 + 1) * 256 + col] = ps2[ti]; }
            }
            lds_barrier();
            if (tid < K) {
                float t1 = 0.f, t2 = 0.f;
#pragma unroll
                for (int k = 0; k < CH_RT; ++k) { t1 += Ws[(k * 2 + 0) * 256 + tid]; t2 += Ws[(k * 2 + 1) * 256 + tid]; }
                if (a.acc_out_off >= 0) {
                    acc_add_sums(reinterpret_cast<long long*>(ws + a.acc_out_off) + (int64_t)arm * ACC_SET_I64, tid, t1, t2);
                } else {
                    float* p = ws + a.part_off + (((int64_t)arm * gridDim.x + blk) * 2) * K;
                    p[tid] = t1;
                    p[K + tid] = t2;
                }
            }
        }
        lds_barrier();
    }
}

// DZ1 = BNbackward(G1) .* relu'(R1).  grid (ceil(B/32), A): every row block reads the batch sums (accumulator set, or
// part: [A][npart][2][W] from fc2's backward, recombined here) and applies them to its 32 rows.
// planes != null (fp32x3 engine, W even): the kernel also writes the three bf16 slice planes of dZ1 that the dW1 GEMM
// stages ([A][3][Rp][128], zero outside [B][W]; grid ceil(Rp/32) row blocks) -- a k_presplit launch less per step.
__global__ __launch_bounds__(256) void k_bn_bwd_apply(const float* __restrict__ G, const float* __restrict__ R,
                                                      const float* __restrict__ mean, const float* __restrict__ rstd,
                                                      const float* __restrict__ part, int npart,
                                                      const long long* __restrict__ acc, float* __restrict__ DZ,
                                                      int B, int W, unsigned short* __restrict__ planes, int Rp) {
    __shared__ __attribute__((aligned(16))) double scratch[1024];
    __shared__ float sums_s[2][128], mu_s[128], rs_s[128];
    const int arm = blockIdx.y, blk = blockIdx.x, tid = threadIdx.x;
    if (acc) {
        if (tid < W) {
            double s1, s2;
            acc_get(acc + (int64_t)arm * ACC_SET_I64, tid, s1, s2);
            sums_s[0][tid] = (float)s1;
            sums_s[1][tid] = (float)s2;
        }
    } else {
        const float r = sums_from_partials<256>(part + (int64_t)arm * npart * 2 * W, npart, 2 * W, scratch);
        if (tid < 2 * W) sums_s[tid < W ? 0 : 1][tid < W ? tid : tid - W] = r;
    }
    if (tid < W) { mu_s[tid] = mean[arm * W + tid]; rs_s[tid] = rstd[arm * W + tid]; }
    lds_barrier();
    const float invB = 1.f / (float)B;
    const int nvalid = max(0, min(32, B - blk * 32));
    const int64_t base = ((int64_t)arm * B + (int64_t)blk * 32) * W;
    auto dz_of = [&](float gv, float rv, int col) {
        const float rs = rs_s[col];
        const float xh = (rv - mu_s[col]) * rs;
        const float g = rs * (gv - sums_s[0][col] * invB - xh * (sums_s[1][col] * invB));
        return rv > 0.f ? g : 0.f;
    };
    if (!planes) {
        for (int i = tid; i < nvalid * W; i += 256) DZ[base + i] = dz_of(G[base + i], R[base + i], i % W);
        return;
    }
    const int64_t plane = (int64_t)Rp * 128;
    unsigned short* pl = planes + (int64_t)arm * 3 * plane;
#pragma unroll
    for (int it = 0; it < 8; ++it) {               // 32 rows x 64 column pairs
        const int i = tid + 256 * it, r = i >> 6, c = (i & 63) * 2, row = blk * 32 + r;
        const bool ok = r < nvalid && c < W;
        const int64_t o = base + (int64_t)(ok ? r : 0) * W + (ok ? c : 0);
        float2 d = make_float2(0.f, 0.f);
        if (nvalid > 0) {
            const float2 gv = *reinterpret_cast<const float2*>(G + o), rv = *reinterpret_cast<const float2*>(R + o);
            if (ok) {
                d = make_float2(dz_of(gv.x, rv.x, c), dz_of(gv.y, rv.y, c + 1));
                *reinterpret_cast<float2*>(DZ + o) = d;
            }
        }
        if (row < Rp) {
            unsigned w[3];
            split3(d.x, d.y, w);
#pragma unroll
            for (int p = 0; p < 3; ++p) *reinterpret_cast<unsigned*>(pl + p * plane + (int64_t)row * 128 + c) = w[p];
        }
    }
}

// ---------------------------------------------------------------------------------------------
// host launchers
// ---------------------------------------------------------------------------------------------
// forward: activations and weights are K-contiguous rows -> ld = max width rounded to 8, + 4
// (ld/4 odd: conflict-free ds_read_b128); weights need rup(N,32) rows.
// backward: weight tile is [N rows][K cols] read along K by lane -> ld = rup(K,32) + 4, rup(N,8) rows.
// The weight tile doubles as scratch: statistics prologue 3 x 16 x 128 floats, batch-sum prologue 16 x 256 doubles,
// epilogue sums CH_RT x 2 x 256 floats.
constexpr int WS_SCRATCH = 8192;
static int fwd_ld(int maxdim) { return rup(maxdim, 8) + 4; }
static int bwd_ld(int maxdim) { return rup(maxdim, 32) + 4; }
// + 256 floats: the BatchNorm statistics (forward) / batch sums (backward) every row block recombines
static size_t chain_smem(int ld, int wrows) { return (size_t)(CHAIN_ROWS * ld + wrows * ld + 256) * sizeof(float); }
// fp32x3 form: three planes per operand, row pitch in dwords (bf16 pairs)
static int x3_ld(int maxdim) { return rup(maxdim, 16) / 2 + 4; }
static size_t chain_smem_x3(int ld, int wrows) { return (size_t)(3 * CHAIN_ROWS * ld + 3 * wrows * ld + 256) * sizeof(float); }
static int launch_fwd(const Ctx& c, ChainFwdArgs& a, int maxdim, const float* params, float* bn_running, int64_t* nbt, const char* what) {
    a.rows = c.lay.chain_rows_fwd;
    a.nblk = c.lay.nblkf;
    const dim3 grid(c.lay.nblkf, c.d.A);
    if (c.small_planes && chain_x3_ok(c) && maxdim <= 128) {
        a.ld = x3_ld(maxdim);
        a.wrows = 128;
        a.wpl_off = c.lay.pl_small;
        const size_t shm = chain_smem_x3(a.ld, a.wrows);
        hipLaunchKernelGGL(k_chain_fwd<true>, grid, dim3(CH_NT), shm, c.stream, a, params, c.ws, bn_running, nbt);
    } else {
        hipLaunchKernelGGL(k_chain_fwd<false>, grid, dim3(CH_NT), chain_smem(a.ld, a.wrows), c.stream, a, params, c.ws, bn_running, nbt);
    }
    hipError_t e_ = hipGetLastError();
    if (e_ != hipSuccess) { set_error("%s: %s", what, hipGetErrorString(e_)); return MMVAE_E_LAUNCH; }
    return 0;
}

static int launch_bwd(const Ctx& c, ChainBwdArgs& a, int maxdim, const float* params, const char* what) {
    const dim3 grid(c.lay.nblkc, c.d.A);
    if (c.small_planes && chain_x3_ok(c) && maxdim <= 128) {
        a.ld = x3_ld(maxdim);
        a.wrows = 128;
        a.wpl_off = c.lay.pl_small;
        const size_t shm = chain_smem_x3(a.ld, a.wrows);
        hipLaunchKernelGGL(k_chain_bwd<true>, grid, dim3(CH_NT), shm, c.stream, a, params, c.ws);
    } else {
        hipLaunchKernelGGL(k_chain_bwd<false>, grid, dim3(CH_NT), chain_smem(a.ld, a.wrows), c.stream, a, params, c.ws);
    }
    hipError_t e_ = hipGetLastError();
    if (e_ != hipSuccess) { set_error("%s: %s", what, hipGetErrorString(e_)); return MMVAE_E_LAUNCH; }
    return 0;
}

int launch_chain_fwd_enc(const Ctx& c, int layer, const float* params, float* bn_running, int64_t* nbt) {
    // layer in 2..5: out = relu(BN_{layer-1}(R_{layer-1}) W^T + b), statistics of the output
    const mmvae_dims& d = c.d;
    const Layout& L = c.lay;
    ChainFwdArgs a{};
    a.nlayers = 1;
    const int i = layer - 1;   // index into R / bn arrays of this layer's output
    const int N = (layer == 5) ? d.L : d.H;
    a.L[0] = FwdLayer{c.po.o[2 * (layer - 1)], c.po.o[2 * (layer - 1) + 1], L.R[i], d.H, N, 1};
    a.x_off = L.R[i - 1];
    a.K0 = d.H;
    a.bn_mean_off = L.bn_mean[i - 1];
    a.bn_rstd_off = L.bn_rstd[i - 1];
    // training: this launch recombines BN_{layer-1}'s partials itself (and row block 0 updates the running
    // buffers); eval: launch_bn_eval_stats has put the running statistics into the workspace
    // BN1's partials come from the fc1 epilogue (32-row blocks), the others from the previous chain launch
    a.bn_part_off = c.h.training ? L.bn_part[i - 1] : -1;
    a.part_n = (layer == 2) ? L.nblk32 : L.nblkf;      // (the partial-array form keeps chain_rows_fwd == CHAIN_ROWS)
    a.part_rows = (layer == 2) ? 32 : L.chain_rows_fwd;
    a.run_mean_off = c.po.bn_mean[i - 1];
    a.run_var_off = c.po.bn_var[i - 1];
    a.run_arm_stride = c.po.bn_per_arm;
    a.bn_idx = i - 1;
    a.bn_eps = c.h.eps;
    a.bn_momentum = c.h.bn_momentum;
    a.stats_part_off = L.bn_part[i];
    const bool acc = c.h.training && c.use_acc();
    a.acc_in_off = acc ? acc_set_off(L, d.A, i - 1) : -1;
    a.acc_out_off = acc ? acc_set_off(L, d.A, i) : -1;
    a.planes_off = -1;
    a.B = d.B;
    a.ld = fwd_ld(max(d.H, N));
    a.wrows = max(rup(N, 32), cdiv(WS_SCRATCH, a.ld));   // Ws doubles as scratch (statistics prologue / epilogue)
    a.per_arm = c.po.per_arm;
    a.ablate = c.tune(MMVAE_TUNE_ABLATE_C);
    a.dbg_off = L.loss_scratch + 2048;
    a.L[0].pl_slot = layer - 2;
    return launch_fwd(c, a, max(d.H, N), params, bn_running, nbt, "k_chain_fwd<enc>");
}

int launch_chain_fwd_enc_eval(const Ctx& c, const float* params) {
    const mmvae_dims& d = c.d;
    const Layout& L = c.lay;
    ChainFwdArgs a{};
    a.nlayers = 4;
    for (int layer = 2; layer <= 5; ++layer) {
        const int i = layer - 1;
        FwdLayer f{c.po.o[2 * (layer - 1)], c.po.o[2 * (layer - 1) + 1], L.R[i], d.H, (layer == 5) ? d.L : d.H, 1};
        if (layer < 5) { f.obn_mean_off = L.bn_mean[i]; f.obn_rstd_off = L.bn_rstd[i]; }   // BN5 belongs to the latent kernel
        a.L[layer - 2] = f;
    }
    a.x_off = L.R[0];
    a.K0 = d.H;
    a.bn_mean_off = L.bn_mean[0];
    a.bn_rstd_off = L.bn_rstd[0];
    a.bn_part_off = -1;
    a.stats_part_off = -1;
    a.acc_in_off = a.acc_out_off = -1;
    a.planes_off = -1;
    a.bn_eps = c.h.eps;
    a.B = d.B;
    a.ld = fwd_ld(max(d.H, d.L));
    a.wrows = rup(max(d.H, d.L), 32);
    a.per_arm = c.po.per_arm;
    a.ablate = 0;
    a.dbg_off = -1;
    for (int i = 0; i < 4; ++i) a.L[i].pl_slot = i;
    return launch_fwd(c, a, max(d.H, d.L), params, nullptr, nullptr, "k_chain_fwd<enc eval>");
}

int launch_chain_fwd_dec(const Ctx& c, const float* params, bool with_couple) {
    const mmvae_dims& d = c.d;
    const Layout& L = c.lay;
    ChainFwdArgs a{};
    a.nlayers = 5;
    a.L[0] = FwdLayer{c.po.o[16], c.po.o[17], L.Dk[0], d.C + d.S, d.L, 1};   // fc6
    a.L[1] = FwdLayer{c.po.o[18], c.po.o[19], L.Dk[1], d.L, d.H, 1};         // fc7
    a.L[2] = FwdLayer{c.po.o[20], c.po.o[21], L.Dk[2], d.H, d.H, 1};
    a.L[3] = FwdLayer{c.po.o[22], c.po.o[23], L.Dk[3], d.H, d.H, 1};
    a.L[4] = FwdLayer{c.po.o[24], c.po.o[25], L.Dk[4], d.H, d.H, 1};
    a.x_off = L.ZIN;
    a.K0 = d.C + d.S;
    a.bn_mean_off = a.bn_rstd_off = a.bn_part_off = -1;
    a.stats_part_off = -1;
    a.acc_in_off = a.acc_out_off = -1;
    // fp32x3 engine: the slice planes of [d10 | 1] (launch_x3_planes(.., 2) then has nothing to do)
    a.planes_off = dec_chain_writes_planes(c) ? L.pl_d10 : -1;
    a.planes_rows = rup(d.B, 256);
    a.B = d.B;
    a.ld = fwd_ld(max(max(d.H, d.L), d.C + d.S));
    a.wrows = rup(max(d.H, d.L), 32);
    a.per_arm = c.po.per_arm;
    a.ablate = c.tune(MMVAE_TUNE_ABLATE_C);
    a.dbg_off = L.loss_scratch + 2048;
    for (int i = 0; i < 5; ++i) a.L[i].pl_slot = 4 + i;
    if (with_couple) {
        // the coupling terms as a second role of this launch (k_chain_fwd_couple): fp32x3 form, accumulator sets, 2 .. 5 arms
        const int maxdim = max(max(d.H, d.L), d.C + d.S);
        a.rows = L.chain_rows_fwd;
        a.nblk = L.nblkf;
        a.ld = x3_ld(maxdim);
        a.wrows = 128;
        a.wpl_off = L.pl_small;
        CoupleRoleArgs cr{};
        cr.nchain = L.nblkf; cr.ncouple = L.nblk32; cr.B = d.B; cr.C = d.C; cr.eps = c.h.eps; cr.lam = c.h.lam;
        cr.cc_off = L.CC; cr.csmp_off = L.CSMP; cr.c_mean_off = L.c_mean; cr.c_iv_off = L.c_iv; cr.couple_part_off = L.couple_part;
        cr.c_acc_off = acc_set_off(L, d.A, ACC_C); cr.t_acc_off = acc_set_off(L, d.A, ACC_T);
        if (c.tune(MMVAE_TUNE_COUPLE_SIDE) == 2) cr.ncouple = 0;   // timing experiment: the role's workgroups exit at once (results wrong)
        const dim3 grid(max(L.nblkf, cdiv(L.nblk32, 2)), d.A + 1);
        const size_t shm_c = (size_t)2 * (5 * d.A * CPL * 64 + 8) * sizeof(float);
        const size_t shm = max(chain_smem_x3(a.ld, a.wrows), shm_c);
#define MMVAE_DEC_COUPLE(AT) hipLaunchKernelGGL(k_chain_fwd_couple<AT>, grid, dim3(CH_NT), shm, c.stream, a, cr, params, c.ws)
        switch (d.A) {
            case 2: MMVAE_DEC_COUPLE(2); break;
            case 3: MMVAE_DEC_COUPLE(3); break;
            case 4: MMVAE_DEC_COUPLE(4); break;
            default: MMVAE_DEC_COUPLE(5); break;
        }
#undef MMVAE_DEC_COUPLE
        hipError_t e_ = hipGetLastError();
        if (e_ != hipSuccess) { set_error("k_chain_fwd_couple: %s", hipGetErrorString(e_)); return MMVAE_E_LAUNCH; }
        return 0;
    }
    return launch_fwd(c, a, max(max(d.H, d.L), d.C + d.S), params, nullptr, nullptr, "k_chain_fwd<dec>");
}
// the fused train step may run the coupling terms inside the decoder chain's launch (see k_chain_fwd_couple).  Measured
// (A/B/A/B per arm count on one box, ms per step, role against side stream): A = 2 0.674 / 0.672, A = 3 0.901 / 0.896, A = 5
// 1.479 / 1.489 -- the two bubbles it removes from the main stream show in a rocprofv3 trace (5 - 6 us each) but not in the
// un-profiled step at two and three arms, where the combined launch is 3 us longer than the chain's own (its grid has a third
// row of workgroups); at five arms the chain's 395 workgroups already run in two rounds and the role fills the second.  So: the
// role from four arms up, the side stream below (MMVAE_TUNE_COUPLE_SIDE: 1 side stream always, 3 role always, 2 the role's
// launch with its workgroups exiting at once -- a timing experiment, results wrong).
bool dec_couple_ok(const Ctx& c) {
    const mmvae_dims& d = c.d;
    return c.h.training && c.use_acc() && c.small_planes && chain_x3_ok(c) && max(max(d.H, d.L), d.C + d.S) <= 128 && d.A >= 2 &&
           d.A <= 5 && d.C <= CPL * 64 && c.tune(MMVAE_TUNE_COUPLE_SIDE) != 1 && (d.A >= 4 || c.tune(MMVAE_TUNE_COUPLE_SIDE) >= 2);
}

int launch_chain_bwd_dec(const Ctx& c, const float* params, int nslab) {
    const mmvae_dims& d = c.d;
    const Layout& L = c.lay;
    ChainBwdArgs a{};
    a.nlayers = 5;
    a.L[0] = BwdLayer{c.po.o[24], L.DZ[10], L.Dk[4], d.H, d.H};          // fc10
    a.L[1] = BwdLayer{c.po.o[22], L.DZ[9], L.Dk[3], d.H, d.H};
    a.L[2] = BwdLayer{c.po.o[20], L.DZ[8], L.Dk[2], d.H, d.H};
    a.L[3] = BwdLayer{c.po.o[18], L.DZ[7], L.Dk[1], d.L, d.H};           // fc7: K = L
    a.L[4] = BwdLayer{c.po.o[16], L.DZ[6], L.Dk[0], d.C + d.S, d.L};     // fc6: K = C+S
    a.g_off = L.GD10_slab;
    a.nslab = nslab;
    a.slab_stride = (int64_t)d.A * d.B * d.H;
    a.bnb_part_off = -1;
    a.bn_mean_off = a.bn_rstd_off = -1;
    a.gout_off = L.GZIN;
    a.part_off = -1;
    a.rprev_off = a.rprev_mean_off = a.rprev_rstd_off = -1;
    a.acc_in_off = a.acc_out_off = -1;
    // the first launch of every backward pass: it zeroes the backward accumulator sets (the latent backward behind it is
    // their first producer)
    a.zero_off = acc_set_off(L, d.A, ACC_BWD);
    a.zero_n4 = c.use_acc() ? (int)(c.bwd_zero_floats() / 4) : 0;
    c.bwd_zeroed = a.zero_n4 > 0;
    a.B = d.B;
    a.ld = bwd_ld(max(max(d.H, d.L), d.C + d.S));
    a.wrows = max(rup(max(d.H, d.L), 8), cdiv(WS_SCRATCH, a.ld));   // Ws doubles as scratch of the epilogue's sums
    a.per_arm = c.po.per_arm;
    for (int i = 0; i < 5; ++i) a.L[i].pl_slot = 9 + 8 - i;   // fc10 .. fc6, transposed planes
    return launch_bwd(c, a, max(max(d.H, d.L), d.C + d.S), params, "k_chain_bwd<dec>");
}

int launch_chain_bwd_enc(const Ctx& c, int layer, const float* params) {
    // layer in 5..2: G[layer] is the gradient w.r.t. BN_layer's output
    const mmvae_dims& d = c.d;
    const Layout& L = c.lay;
    ChainBwdArgs a{};
    a.nlayers = 1;
    const int i = layer - 1;
    const int N = (layer == 5) ? d.L : d.H;
    a.L[0] = BwdLayer{c.po.o[2 * (layer - 1)], L.DZ[layer], L.R[i], d.H, N};
    a.g_off = L.G[layer];
    a.nslab = 1;
    a.slab_stride = 0;
    // BN_layer's backward sums: per workgroup of the latent backward (layer 5), else of the previous chain launch
    a.bnb_part_off = L.bnb_part[layer];
    a.bnb_n = (layer == 5) ? cdiv(d.B, LAT_ROWS_BWD) : L.nblkc;
    a.bn_mean_off = L.bn_mean[i];
    a.bn_rstd_off = L.bn_rstd[i];
    a.gout_off = L.G[layer - 1];
    a.part_off = L.bnb_part[layer - 1];
    a.rprev_off = L.R[i - 1];
    a.rprev_mean_off = L.bn_mean[i - 1];
    a.rprev_rstd_off = L.bn_rstd[i - 1];
    a.acc_in_off = c.use_acc() ? acc_set_off(L, d.A, ACC_BWD + layer - 1) : -1;
    a.acc_out_off = c.use_acc() ? acc_set_off(L, d.A, ACC_BWD + layer - 2) : -1;
    a.zero_off = 0;
    a.zero_n4 = 0;
    a.B = d.B;
    a.ld = bwd_ld(max(d.H, N));
    a.wrows = max(rup(N, 8), cdiv(WS_SCRATCH, a.ld));   // Ws doubles as scratch (batch-sum prologue / epilogue)
    a.per_arm = c.po.per_arm;
    a.L[0].pl_slot = 9 + layer - 2;
    return launch_bwd(c, a, max(d.H, N), params, "k_chain_bwd<enc>");
}

int launch_bn_bwd_apply1(const Ctx& c) {
    const mmvae_dims& d = c.d;
    const Layout& L = c.lay;
    // fp32x3 engine: this launch also writes the slice planes of dZ1 (launch_x3_planes(.., 4) then has nothing to do)
    const bool planes = bn_apply_writes_planes(c);
    const int Rp = rup(d.B, 256);
    hipLaunchKernelGGL(k_bn_bwd_apply, dim3(planes ? Rp / 32 : L.nblk32, d.A), dim3(256), 0, c.stream, c.ws + L.G[1],
                       c.ws + L.R[0], c.ws + L.bn_mean[0], c.ws + L.bn_rstd[0], c.ws + L.bnb_part[1], L.nblkc,
                       c.use_acc() ? reinterpret_cast<const long long*>(c.ws + acc_set_off(L, d.A, ACC_BWD)) : nullptr,
                       c.ws + L.DZ[1], d.B, d.H, planes ? reinterpret_cast<unsigned short*>(c.ws + L.pl_dz1) : nullptr, Rp);
    HIP_LAUNCH_CHECK("k_bn_bwd_apply");
    return 0;
}

}  // namespace mmvae
