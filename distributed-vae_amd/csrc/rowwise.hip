// Row-wise (one wavefront per cell) kernels and the small reductions between them:
//
//   (batch statistics: every consumer recombines the producers' per-row-block (mean, M2) partials itself,
//    common.hpp stats_from_partials -- BatchNorm1d incl. running statistics, nn_model.py:208-255, and
//    inv_var, nn_model.py:75-77; eval mode: k_stats_from_running)
//   k_lat_fwd         x_low = BN5(R5); c_prob = softmax(fcc x_low); c = softmax(c_prob/tau);
//                     Gumbel-softmax sample; state head; reparameterise; decoder input
//                     (nn_model.py:268-269, :337-351, :413-493)
//   k_couple          pairwise coupling terms over arms (nn_model.py:558-569)
//   k_loss_finalize   scalars of nn_model.py:542-598
//   k_lat_bwd         autograd of k_lat_fwd + coupling / entropy / KL terms
//   k_reduce          slabs -> flat gradient buffer;  k_adam: torch.optim.Adam(W) update
//
// Wave reductions only (no MFMA): these tensors are [B, <=128] and HBM/L2 resident.
#include "common.hpp"
#include "couple.hpp"
#include <type_traits>
#include <math.h>
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


NoiseDev make_noise_dev(const mmvae_noise* nz, const mmvae_hyper& h) {
    NoiseDev n{};
    n.mode = nz ? nz->mode : 0;
    if (nz) {
        n.x_mask = nz->x_mask; n.u_gumbel = nz->u_gumbel; n.u_state = nz->u_state; n.s_mask = nz->s_mask;
        n.k0 = (uint32_t)nz->seed; n.k1 = (uint32_t)(nz->seed >> 32);
        n.step_lo = (uint32_t)nz->offset; n.step_hi = (uint32_t)(nz->offset >> 32);
    }
    // smallest field width that represents the 16-bit keep threshold exactly (0 and 65536: one bit, thr 0 / 2)
    const uint32_t t16 = keep_threshold16(h.x_drop);
    uint32_t ml = 0;
    while (ml < 4 && (t16 & ((65536u >> (1u << ml)) - 1u)) != 0) ++ml;
    n.x_mlog2 = ml;
    n.x_thr = t16 >> (16u - (1u << ml));
    n.s_keep_thr = keep_threshold(h.s_drop);
    return n;
}

// eval mode: statistics come from the running buffers.  One launch for the five BatchNorm layers: grid (A, 5).
struct EvalStatArgs { int64_t run_mean[5], run_var[5], mean_out[5], rstd_out[5]; int W[5]; };
__global__ void k_stats_from_running(const float* __restrict__ bn_running, int64_t run_arm_stride, EvalStatArgs a, float eps,
                                     float* __restrict__ ws) {
    const int arm = blockIdx.x, layer = blockIdx.y, col = threadIdx.x;
    const int W = a.W[layer];
    if (col < W) {
        ws[a.mean_out[layer] + arm * W + col] = bn_running[a.run_mean[layer] + arm * run_arm_stride + col];
        ws[a.rstd_out[layer] + arm * W + col] = 1.0f / sqrtf(bn_running[a.run_var[layer] + arm * run_arm_stride + col] + eps);
    }
}

// ---------------------------------------------------------------------------------------------
struct LatArgs {
    int A, B, L, C, S;
    float tau, temp, eps, s_drop;
    int hard, training, eval_flag;
    int64_t per_arm, o_wc, o_bc, o_wms, o_bms;
    // workspace offsets
    int64_t R5, mean5, rstd5, XLOW, CPROB, CC, YSOFT, CSMP, Y, MS, MU, LV, SS, ZIN, c_part, lat_part;
    // backward only
    int64_t GZIN, GMS, GZC, G5, bnb_part5, T, c_mean, c_iv;
    float am1, beta, lam;
    int64_t dbg_off;   // >= 0: diagnostic stamp counters (MMVAE_ABLATE_L=8)
    int32_t* labels;   // non-null (eval): labels[arm * B + b] = argmax_k c, the `classify` of the consensus path
    // forward, training: BN5's partials [A][nblk][2][L] are recombined by every row block
    int64_t bn_part5, run_mean_off, run_var_off, run_arm_stride;
    int bn5_n;             // partials fc5's launch emitted (one per CHAIN_ROWS cells)
    int64_t acc_bn5, acc_bnb5;   // accumulator sets ([A] each) instead of bn_part5 (read) / bnb_part5 (added to); -1 = partials
    int64_t acc_c, acc_T;        // ... instead of c_part (added to by the forward kernels) / T (read by the backward kernels)
    float bn_momentum;
    // category subset of the pruning-time forward (nn_model.py:332-335: c = softmax(c_prob[:, mask] / tau) on the kept
    // categories, 0 elsewhere): bit k of cmask = category k is kept; use_mask == 0: all of them
    uint32_t cmask[4];
    int use_mask;
};
__device__ __forceinline__ bool cat_kept(const LatArgs& a, int col) {
    return !a.use_mask || ((a.cmask[(col >> 5) & 3] >> (col & 31)) & 1u) != 0u;
}

__device__ __forceinline__ float gumbel_u(const NoiseDev& nz, int arm, int B, int C, int b, int col) {
    if (nz.mode == 0) return nz.u_gumbel[((int64_t)arm * B + b) * C + col];
    return noise_uniform(nz, arm, STREAM_GUMBEL, (uint64_t)b * C + col);
}
__device__ __forceinline__ float state_u(const NoiseDev& nz, int arm, int B, int S, int b, int s) {
    if (nz.mode == 0) return nz.u_state[((int64_t)arm * B + b) * S + s];
    return noise_uniform(nz, arm, STREAM_STATE, (uint64_t)b * S + s);
}
__device__ __forceinline__ bool state_keep(const NoiseDev& nz, int arm, int B, int S, int b, int s) {
    if (nz.mode == 0) return nz.s_mask[((int64_t)arm * B + b) * S + s] != 0;
    return noise_keep(nz, arm, STREAM_SMASK, (uint64_t)b * S + s, nz.s_keep_thr);
}

// Stages fcc (transposed to [L][C]) and the state-head weights in LDS once per workgroup.
constexpr int LAT_NW = 16;   // waves per workgroup: LAT_ROWS / 16 cells each, one at a time
__device__ __forceinline__ void lat_stage_weights(float* WcT, float* Wm, const float* __restrict__ Wc,
                                                  const float* __restrict__ Wms, int L, int C, int S) {
    for (int i = threadIdx.x; i < C * L; i += blockDim.x) {
        const int col = i / L, k = i % L;
        WcT[k * C + col] = Wc[i];
    }
    for (int i = threadIdx.x; i < 2 * S * (L + C); i += blockDim.x) Wm[i] = Wms[i];
    __syncthreads();
}

// grid (ceil(B/LAT_ROWS), A), 1024 threads; wave w handles cells b0 + w, b0 + w + 16, ... -- all LAT_NR of them side
// by side: the per-cell work is one long dependency chain (three softmaxes = six wave reductions, the Gumbel
// transform, four state-head dot products), and a wave that walks its cells one after the other spends most of its
// time waiting on that chain (measured: 15 k cycles per cell with 4 waves per SIMD).
constexpr int LAT_NR = LAT_ROWS / LAT_NW;
static_assert(LAT_NR * LAT_NW == LAT_ROWS, "LAT_ROWS must be a multiple of the wave count");
__global__ __launch_bounds__(1024) void k_lat_fwd(const LatArgs a_in, const NoiseDev nz_in,
                                                  const float* __restrict__ params, float* __restrict__ ws,
                                                  float* __restrict__ bn_running, int64_t* __restrict__ nbt) {
    const LatArgs a = a_in;      // argument blocks into registers once (see k_chain_fwd)
    const NoiseDev nz = nz_in;
    extern __shared__ __attribute__((aligned(16))) float lat_smem[];
    // statistics scratch of the prologue, then the workgroup's c tile [LAT_ROWS][128] for the block statistics
    __shared__ __attribute__((aligned(16))) float sh_buf[LAT_ROWS * CPL * 64];
    __shared__ float sh_ps[8][CPL * 64];
    __shared__ float sh_red[LAT_NW][2], sh_bn5[2][64];
    const int arm = blockIdx.y, blk = blockIdx.x, b0 = blk * LAT_ROWS;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int B = a.B, L = a.L, C = a.C, S = a.S;
    const float* P = params + (int64_t)arm * a.per_arm;
    float* WcT = lat_smem;            // [L][C]
    float* Wms = lat_smem + C * L;    // [2S][L+C]
    const int64_t ab = (int64_t)arm * B;
    const float eps = a.eps;

    const bool stamps = a.dbg_off >= 0;
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

    // this wave's cells; their fc5 outputs are requested first, so that this latency, the weight staging and the
    // statistics partials all overlap
    int bb[LAT_NR];
    bool okr[LAT_NR];
    float r5[LAT_NR];
#pragma unroll
    for (int r = 0; r < LAT_NR; ++r) {
        bb[r] = b0 + wv + LAT_NW * r;
        okr[r] = bb[r] < B;   // wave-uniform
        r5[r] = lane < L ? ws[a.R5 + (ab + min(bb[r], B - 1)) * L + lane] : 0.f;
    }
    const bool vcol[CPL] = {lane < C, lane + 64 < C};
    float bcv[CPL];
#pragma unroll
    for (int t = 0; t < CPL; ++t) bcv[t] = vcol[t] ? P[a.o_bc + lane + 64 * t] : 0.f;
    lat_stage_weights(WcT, Wms, P + a.o_wc, P + a.o_wms, L, C, S);
    const float* bms = P + a.o_bms;   // [2S]

    if (a.bn_part5 >= 0) {   // training: BN5 batch statistics from fc5's per-row-block partials
        float mean, m2;
        if (a.acc_bn5 >= 0) {
            mean = m2 = 0.f;
            if ((int)threadIdx.x < L)
                acc_mean_m2(reinterpret_cast<const long long*>(ws + a.acc_bn5) + (int64_t)arm * ACC_SET_I64, threadIdx.x, B, mean, m2);
        } else {
            stats_from_partials<64 * LAT_NW>(ws + a.bn_part5 + (int64_t)arm * a.bn5_n * 2 * L, a.bn5_n, B, CHAIN_ROWS, L,
                                             sh_buf, mean, m2);
        }
        if (threadIdx.x < L) {
            const int t = threadIdx.x;
            const float rstd = 1.0f / sqrtf(m2 / (float)B + eps);
            sh_bn5[0][t] = mean;
            sh_bn5[1][t] = rstd;
            if (blk == 0) {
                ws[a.mean5 + arm * L + t] = mean;
                ws[a.rstd5 + arm * L + t] = rstd;
                if (bn_running) {
                    float* rm = bn_running + a.run_mean_off + arm * a.run_arm_stride;
                    float* rv = bn_running + a.run_var_off + arm * a.run_arm_stride;
                    rm[t] = (1.f - a.bn_momentum) * rm[t] + a.bn_momentum * mean;
                    rv[t] = (1.f - a.bn_momentum) * rv[t] + a.bn_momentum * (m2 / (float)max(B - 1, 1));
                }
                if (nbt && t == 0) nbt[arm * MMVAE_N_BN + 4] += 1;
            }
        }
        lds_barrier();
    }
    const float mu5 = lane < L ? (a.bn_part5 >= 0 ? sh_bn5[0][lane] : ws[a.mean5 + arm * L + lane]) : 0.f;
    const float rs5 = lane < L ? (a.bn_part5 >= 0 ? sh_bn5[1][lane] : ws[a.rstd5 + arm * L + lane]) : 0.f;
    stamp(0);   // weight staging + statistics loads

    // ---- x_low = BN5(R5)
    float xl[LAT_NR];
#pragma unroll
    for (int r = 0; r < LAT_NR; ++r) {
        xl[r] = lane < L ? (r5[r] - mu5) * rs5 : 0.f;
        if (okr[r] && lane < L) {
            ws[a.XLOW + (ab + bb[r]) * L + lane] = xl[r];
            ws[a.Y + (ab + bb[r]) * (L + C) + lane] = xl[r];
        }
    }
    // ---- zc = fcc(x_low); c_prob = softmax(zc)
    float z[LAT_NR][CPL];
#pragma unroll
    for (int r = 0; r < LAT_NR; ++r)
#pragma unroll
        for (int t = 0; t < CPL; ++t) z[r][t] = bcv[t];
    for (int k = 0; k < L; ++k) {
        float w[CPL];
#pragma unroll
        for (int t = 0; t < CPL; ++t) w[t] = vcol[t] ? WcT[k * C + lane + 64 * t] : 0.f;
#pragma unroll
        for (int r = 0; r < LAT_NR; ++r) {
            const float xk = __shfl(xl[r], k, 64);
#pragma unroll
            for (int t = 0; t < CPL; ++t) z[r][t] += xk * w[t];
        }
    }
    float m[LAT_NR], ssum[LAT_NR], e[LAT_NR][CPL];
    float cp[LAT_NR][CPL], cc[LAT_NR][CPL], lc[LAT_NR][CPL], ys[LAT_NR][CPL], cs[LAT_NR][CPL];
    // softmax over the valid columns of v (in place in e, sum in ssum)
    auto softmax_rows = [&](float (&v)[LAT_NR][CPL], float (&out)[LAT_NR][CPL]) {
#pragma unroll
        for (int r = 0; r < LAT_NR; ++r) {
            m[r] = -INFINITY;
#pragma unroll
            for (int t = 0; t < CPL; ++t) if (vcol[t]) m[r] = fmaxf(m[r], v[r][t]);
        }
#pragma unroll
        for (int r = 0; r < LAT_NR; ++r) m[r] = wave_max(m[r]);
#pragma unroll
        for (int r = 0; r < LAT_NR; ++r) {
            ssum[r] = 0.f;
#pragma unroll
            for (int t = 0; t < CPL; ++t) { e[r][t] = vcol[t] ? expf(v[r][t] - m[r]) : 0.f; ssum[r] += e[r][t]; }
        }
#pragma unroll
        for (int r = 0; r < LAT_NR; ++r) ssum[r] = wave_sum(ssum[r]);
#pragma unroll
        for (int r = 0; r < LAT_NR; ++r) {
            const float inv = 1.f / ssum[r];   // one division per cell, not per element: the kernel is VALU-bound
#pragma unroll
            for (int t = 0; t < CPL; ++t) out[r][t] = e[r][t] * inv;
        }
    };
    softmax_rows(z, cp);
    stamp(1);   // x_low, fcc, first softmax
    // ---- c = softmax(c_prob / tau)
    const float inv_tau = 1.f / a.tau, inv_temp = 1.f / a.temp;
    float tmp[LAT_NR][CPL];
#pragma unroll
    for (int r = 0; r < LAT_NR; ++r)
#pragma unroll
        for (int t = 0; t < CPL; ++t) tmp[r][t] = cat_kept(a, lane + 64 * t) ? cp[r][t] * inv_tau : -INFINITY;   // masked-out: exp(-inf) = 0
    softmax_rows(tmp, cc);
#pragma unroll
    for (int r = 0; r < LAT_NR; ++r)
#pragma unroll
        for (int t = 0; t < CPL; ++t) lc[r][t] = logf(cc[r][t] + eps);
    // ---- Gumbel-softmax sample
    bool hard = a.hard != 0;
    if (a.eval_flag) {
        hard = true;
#pragma unroll
        for (int r = 0; r < LAT_NR; ++r)
#pragma unroll
            for (int t = 0; t < CPL; ++t) ys[r][t] = cc[r][t];
    } else {
#pragma unroll
        for (int r = 0; r < LAT_NR; ++r)
#pragma unroll
            for (int t = 0; t < CPL; ++t) {
                tmp[r][t] = 0.f;
                if (vcol[t]) {
                    const float U = gumbel_u(nz, arm, B, C, min(bb[r], B - 1), lane + 64 * t);
                    const float g = -logf(-logf(U + eps) + eps);
                    tmp[r][t] = (lc[r][t] + g) * inv_temp;
                }
            }
        softmax_rows(tmp, ys);
    }
    if (hard) {
#pragma unroll
        for (int r = 0; r < LAT_NR; ++r) {
            float mv = -INFINITY;
#pragma unroll
            for (int t = 0; t < CPL; ++t) if (vcol[t]) mv = fmaxf(mv, ys[r][t]);
            mv = wave_max(mv);
            int cand = 1 << 30;
#pragma unroll
            for (int t = 0; t < CPL; ++t) if (vcol[t] && ys[r][t] == mv) cand = min(cand, lane + 64 * t);
            cand = wave_min_i(cand);
            if (a.labels && a.eval_flag && okr[r] && lane == 0) a.labels[ab + bb[r]] = cand;   // eval: ys == c
#pragma unroll
            for (int t = 0; t < CPL; ++t) {
                const float hv = (lane + 64 * t == cand) ? 1.f : 0.f;
                cs[r][t] = (hv - ys[r][t]) + ys[r][t];   // (y_hard - y).detach() + y, nn_model.py:492
            }
        }
    } else {
#pragma unroll
        for (int r = 0; r < LAT_NR; ++r)
#pragma unroll
            for (int t = 0; t < CPL; ++t) cs[r][t] = ys[r][t];
    }
    stamp(2);   // second softmax, Gumbel sample (noise)
    // ---- store; c goes to the workgroup tile for the block statistics (zero rows beyond the batch)
    float kl_acc = 0.f, ent_acc = 0.f;
#pragma unroll
    for (int r = 0; r < LAT_NR; ++r) {
#pragma unroll
        for (int t = 0; t < CPL; ++t) {
            const int col = lane + 64 * t;
            sh_buf[(wv + LAT_NW * r) * (CPL * 64) + col] = (okr[r] && vcol[t]) ? cc[r][t] : 0.f;
            if (okr[r] && vcol[t]) {
                const int64_t o = (ab + bb[r]) * C + col;
                ws[a.CPROB + o] = cp[r][t];
                ws[a.CC + o] = cc[r][t];
                ws[a.YSOFT + o] = ys[r][t];
                ws[a.CSMP + o] = cs[r][t];
                ws[a.Y + (ab + bb[r]) * (L + C) + L + col] = cs[r][t];
                ws[a.ZIN + (ab + bb[r]) * (C + S) + col] = cs[r][t];
                ent_acc += cc[r][t] * lc[r][t];
            }
        }
    }
    stamp(3);   // stores
    // ---- state head: [mu | sigma_pre] = y [Wmu; Wsigma]^T + b
    float mso[LAT_NR];
#pragma unroll
    for (int r = 0; r < LAT_NR; ++r) mso[r] = 0.f;
    for (int o = 0; o < 2 * S; ++o) {
        const float* w = Wms + (int64_t)o * (L + C);
        const float wl = lane < L ? w[lane] : 0.f;
        float wc[CPL], pr[LAT_NR];
#pragma unroll
        for (int t = 0; t < CPL; ++t) wc[t] = vcol[t] ? w[L + lane + 64 * t] : 0.f;
#pragma unroll
        for (int r = 0; r < LAT_NR; ++r) {
            pr[r] = xl[r] * wl;
#pragma unroll
            for (int t = 0; t < CPL; ++t) pr[r] += cs[r][t] * wc[t];
        }
        const float bo = bms[o];
#pragma unroll
        for (int r = 0; r < LAT_NR; ++r) {
            const float pv = wave_sum(pr[r]) + bo;
            if (lane == o) mso[r] = pv;
        }
    }
#pragma unroll
    for (int r = 0; r < LAT_NR; ++r) {
        if (okr[r] && lane < 2 * S) ws[a.MS + (ab + bb[r]) * 2 * S + lane] = mso[r];
        const float sg = __shfl(mso[r], (lane + S) & 63, 64);
        if (okr[r] && lane < S) {
            const int b = bb[r];
            const float mu = mso[r];
            const float var = 1.f / (1.f + expf(-sg));
            const float lv = logf(var + eps);
            const float sd = sqrtf(expf(lv));
            const float U = state_u(nz, arm, B, S, b, lane);
            const float sv = U * sd + mu;
            float sin_ = sv;
            if (a.training && a.s_drop > 0.f) sin_ = state_keep(nz, arm, B, S, b, lane) ? sv / (1.f - a.s_drop) : 0.f;
            ws[a.MU + (ab + b) * S + lane] = mu;
            ws[a.LV + (ab + b) * S + lane] = lv;
            ws[a.SS + (ab + b) * S + lane] = sv;
            ws[a.ZIN + (ab + b) * (C + S) + C + lane] = sin_;
            kl_acc += 1.f + lv - mu * mu - expf(lv);
        }
    }
    stamp(4);   // state head
    // ---- block partials: mean and M2 of c over the workgroup's cells, two passes over the LDS tile, 8 row groups
    kl_acc = wave_sum(kl_acc);
    ent_acc = wave_sum(ent_acc);
    if (lane == 0) { sh_red[wv][0] = kl_acc; sh_red[wv][1] = ent_acc; }
    lds_barrier();
    {
        const int col = threadIdx.x & (CPL * 64 - 1), g = threadIdx.x >> 7;   // 1024 threads = 8 groups x 128 columns
        const int nv = min(LAT_ROWS, B - b0);
        float v[LAT_ROWS / 8];
        float s1 = 0.f;
#pragma unroll
        for (int i = 0; i < LAT_ROWS / 8; ++i) { v[i] = sh_buf[(g + 8 * i) * (CPL * 64) + col]; s1 += v[i]; }   // rows beyond nv are 0
        sh_ps[g][col] = s1;
        lds_barrier();
        float tot = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) tot += sh_ps[k][col];
        const float mean = tot / (float)nv;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < LAT_ROWS / 8; ++i) { const float d = v[i] - mean; q += (g + 8 * i < nv) ? d * d : 0.f; }
        lds_barrier();   // every thread has read sh_ps
        sh_ps[g][col] = q;
        lds_barrier();
        if (g == 0 && col < C) {
            float m2 = 0.f;
#pragma unroll
            for (int k = 0; k < 8; ++k) m2 += sh_ps[k][col];
            if (a.acc_c >= 0) {
                acc_add_stats(reinterpret_cast<long long*>(ws + a.acc_c) + (int64_t)arm * ACC_SET_I64, col, (float)nv, mean, m2);
            } else {
                float* p = ws + a.c_part + (((int64_t)arm * gridDim.x + blk) * 2) * C;
                p[col] = mean;
                p[C + col] = m2;
            }
        }
    }
    if (threadIdx.x == 0) {
        float* p = ws + a.lat_part + ((int64_t)arm * gridDim.x + blk) * 2;
        float k0 = 0.f, k1 = 0.f;
#pragma unroll
        for (int w = 0; w < LAT_NW; ++w) { k0 += sh_red[w][0]; k1 += sh_red[w][1]; }
        p[0] = k0;
        p[1] = k1;
    }
    stamp(5);   // block reduction
    if (stamps && lane == 0) {
        unsigned long long* dbg = reinterpret_cast<unsigned long long*>(ws + a.dbg_off);
        for (int i = 0; i < 6; ++i) atomicAdd(dbg + i, ph[i]);
        atomicAdd(dbg + 6, 1ull);
    }
}

// ---------------------------------------------------------------------------------------------
// Half-wave layout of the latent forward (C <= 96, L <= 32, 2 S <= 32 -- the reference's 92 / 10 / 2):
// a cell owns 32 lanes x 3 registers (96 column slots for 92 categories instead of 64 x 2 = 128), a wave instruction
// therefore serves TWO cells, and the softmax reductions stop at 32 lanes (the last cross-half step is dropped).
// The kernel is VALU-bound (a wave instruction occupies its SIMD for four cycles): three registers for two cells
// instead of two for one is a quarter less element-wise work, and a reduction step now counts for two cells.
// 8 waves x 2 halves x 3 cells = the same LAT_ROWS cells per workgroup and the same partial layouts as k_lat_fwd.
// ---------------------------------------------------------------------------------------------
constexpr int LH_NW = 8, LH_CPL = 3, LH_NR = LAT_ROWS / (LH_NW * 2);
static_assert(LH_NR * LH_NW * 2 == LAT_ROWS, "LAT_ROWS must be a multiple of 16");
template <typename Op>
__device__ __forceinline__ float half_allreduce(float v, Op op) {   // wave_allreduce without the 32-lane swap
    v = op(v, dpp_f<0xB1>(v));
    v = op(v, dpp_f<0x4E>(v));
    v = op(v, dpp_f<0x141>(v));
    v = op(v, dpp_f<0x140>(v));
    const unsigned u = __builtin_bit_cast(unsigned, v);
    const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    return op(__builtin_bit_cast(float, (unsigned)r[0]), __builtin_bit_cast(float, (unsigned)r[1]));
}
__device__ __forceinline__ float half_sum(float v) { return half_allreduce(v, [](float a, float b) { return a + b; }); }
__device__ __forceinline__ float half_max(float v) { return half_allreduce(v, [](float a, float b) { return fmaxf(a, b); }); }
__device__ __forceinline__ int half_min_i(int v) {
    const float f = half_allreduce(__builtin_bit_cast(float, v), [](float a, float b) {
        const int x = __builtin_bit_cast(int, a), y = __builtin_bit_cast(int, b);
        return __builtin_bit_cast(float, x < y ? x : y);
    });
    return __builtin_bit_cast(int, f);
}

__global__ __launch_bounds__(64 * LH_NW) void k_lat_fwd_h(const LatArgs a_in, const NoiseDev nz_in,
                                                         const float* __restrict__ params, float* __restrict__ ws,
                                                         float* __restrict__ bn_running, int64_t* __restrict__ nbt) {
    constexpr int NR = LH_NR, CP = LH_CPL, NT = 64 * LH_NW;
    const LatArgs a = a_in;
    const NoiseDev nz = nz_in;
    extern __shared__ __attribute__((aligned(16))) float lat_smem[];
    __shared__ __attribute__((aligned(16))) float sh_buf[LAT_ROWS * 128];   // prologue scratch, then the c tile
    __shared__ float sh_ps[NT / 128][128];
    __shared__ float sh_red[LH_NW][2], sh_bn5[2][64];
    const int arm = blockIdx.y, blk = blockIdx.x, b0 = blk * LAT_ROWS;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int sub = lane & 31, hb = lane & 32, half = lane >> 5;
    const int B = a.B, L = a.L, C = a.C, S = a.S;
    const float* P = params + (int64_t)arm * a.per_arm;
    float* WcT = lat_smem;            // [L][C]
    float* Wms = lat_smem + C * L;    // [2S][L+C]
    const int64_t ab = (int64_t)arm * B;
    const float eps = a.eps;

    int slot[NR], bb[NR];
    bool okr[NR];
    float r5[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        slot[r] = r * (LH_NW * 2) + wv * 2 + half;
        bb[r] = b0 + slot[r];
        okr[r] = bb[r] < B;                   // per half-wave
        r5[r] = sub < L ? ws[a.R5 + (ab + min(bb[r], B - 1)) * L + sub] : 0.f;
    }
    bool vcol[CP];
    float bcv[CP];
#pragma unroll
    for (int t = 0; t < CP; ++t) { vcol[t] = sub + 32 * t < C; bcv[t] = vcol[t] ? P[a.o_bc + sub + 32 * t] : 0.f; }
    lat_stage_weights(WcT, Wms, P + a.o_wc, P + a.o_wms, L, C, S);
    const float* bms = P + a.o_bms;

    if (a.bn_part5 >= 0) {
        float mean, m2;
        if (a.acc_bn5 >= 0) {
            mean = m2 = 0.f;
            if ((int)threadIdx.x < L)
                acc_mean_m2(reinterpret_cast<const long long*>(ws + a.acc_bn5) + (int64_t)arm * ACC_SET_I64, threadIdx.x, B, mean, m2);
        } else {
            stats_from_partials<NT>(ws + a.bn_part5 + (int64_t)arm * a.bn5_n * 2 * L, a.bn5_n, B, CHAIN_ROWS, L, sh_buf, mean, m2);
        }
        if (threadIdx.x < L) {
            const int t = threadIdx.x;
            const float rstd = 1.0f / sqrtf(m2 / (float)B + eps);
            sh_bn5[0][t] = mean;
            sh_bn5[1][t] = rstd;
            if (blk == 0) {
                ws[a.mean5 + arm * L + t] = mean;
                ws[a.rstd5 + arm * L + t] = rstd;
                if (bn_running) {
                    float* rm = bn_running + a.run_mean_off + arm * a.run_arm_stride;
                    float* rv = bn_running + a.run_var_off + arm * a.run_arm_stride;
                    rm[t] = (1.f - a.bn_momentum) * rm[t] + a.bn_momentum * mean;
                    rv[t] = (1.f - a.bn_momentum) * rv[t] + a.bn_momentum * (m2 / (float)max(B - 1, 1));
                }
                if (nbt && t == 0) nbt[arm * MMVAE_N_BN + 4] += 1;
            }
        }
        lds_barrier();
    }
    const float mu5 = sub < L ? (a.bn_part5 >= 0 ? sh_bn5[0][sub] : ws[a.mean5 + arm * L + sub]) : 0.f;
    const float rs5 = sub < L ? (a.bn_part5 >= 0 ? sh_bn5[1][sub] : ws[a.rstd5 + arm * L + sub]) : 0.f;

    // ---- x_low = BN5(R5)
    float xl[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        xl[r] = sub < L ? (r5[r] - mu5) * rs5 : 0.f;
        if (okr[r] && sub < L) {
            ws[a.XLOW + (ab + bb[r]) * L + sub] = xl[r];
            ws[a.Y + (ab + bb[r]) * (L + C) + sub] = xl[r];
        }
    }
    // ---- zc = fcc(x_low); c_prob = softmax(zc)
    float z[NR][CP];
#pragma unroll
    for (int r = 0; r < NR; ++r)
#pragma unroll
        for (int t = 0; t < CP; ++t) z[r][t] = bcv[t];
    for (int k = 0; k < L; ++k) {
        float w[CP];
#pragma unroll
        for (int t = 0; t < CP; ++t) w[t] = vcol[t] ? WcT[k * C + sub + 32 * t] : 0.f;
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const float xk = __shfl(xl[r], hb + k, 64);   // this half's own cell
#pragma unroll
            for (int t = 0; t < CP; ++t) z[r][t] += xk * w[t];
        }
    }
    float m[NR], ssum[NR], e[NR][CP];
    float cp[NR][CP], cc[NR][CP], lc[NR][CP], ys[NR][CP], cs[NR][CP];
    auto softmax_rows = [&](float (&v)[NR][CP], float (&out)[NR][CP]) {
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            m[r] = -INFINITY;
#pragma unroll
            for (int t = 0; t < CP; ++t) if (vcol[t]) m[r] = fmaxf(m[r], v[r][t]);
        }
#pragma unroll
        for (int r = 0; r < NR; ++r) m[r] = half_max(m[r]);
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            ssum[r] = 0.f;
#pragma unroll
            for (int t = 0; t < CP; ++t) { e[r][t] = vcol[t] ? expf(v[r][t] - m[r]) : 0.f; ssum[r] += e[r][t]; }
        }
#pragma unroll
        for (int r = 0; r < NR; ++r) ssum[r] = half_sum(ssum[r]);
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const float inv = 1.f / ssum[r];
#pragma unroll
            for (int t = 0; t < CP; ++t) out[r][t] = e[r][t] * inv;
        }
    };
    softmax_rows(z, cp);
    // ---- c = softmax(c_prob / tau)
    const float inv_tau = 1.f / a.tau, inv_temp = 1.f / a.temp;
    float tmp[NR][CP];
#pragma unroll
    for (int r = 0; r < NR; ++r)
#pragma unroll
        for (int t = 0; t < CP; ++t) tmp[r][t] = cat_kept(a, sub + 32 * t) ? cp[r][t] * inv_tau : -INFINITY;   // masked-out: exp(-inf) = 0
    softmax_rows(tmp, cc);
#pragma unroll
    for (int r = 0; r < NR; ++r)
#pragma unroll
        for (int t = 0; t < CP; ++t) lc[r][t] = logf(cc[r][t] + eps);
    // ---- Gumbel-softmax sample
    bool hard = a.hard != 0;
    if (a.eval_flag) {
        hard = true;
#pragma unroll
        for (int r = 0; r < NR; ++r)
#pragma unroll
            for (int t = 0; t < CP; ++t) ys[r][t] = cc[r][t];
    } else {
#pragma unroll
        for (int r = 0; r < NR; ++r)
#pragma unroll
            for (int t = 0; t < CP; ++t) {
                tmp[r][t] = 0.f;
                if (vcol[t]) {
                    const float U = gumbel_u(nz, arm, B, C, min(bb[r], B - 1), sub + 32 * t);
                    const float g = -logf(-logf(U + eps) + eps);
                    tmp[r][t] = (lc[r][t] + g) * inv_temp;
                }
            }
        softmax_rows(tmp, ys);
    }
    if (hard) {
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            float mv = -INFINITY;
#pragma unroll
            for (int t = 0; t < CP; ++t) if (vcol[t]) mv = fmaxf(mv, ys[r][t]);
            mv = half_max(mv);
            int cand = 1 << 30;
#pragma unroll
            for (int t = 0; t < CP; ++t) if (vcol[t] && ys[r][t] == mv) cand = min(cand, sub + 32 * t);
            cand = half_min_i(cand);
            if (a.labels && a.eval_flag && okr[r] && sub == 0) a.labels[ab + bb[r]] = cand;    // eval: ys == c
#pragma unroll
            for (int t = 0; t < CP; ++t) {
                const float hv = (sub + 32 * t == cand) ? 1.f : 0.f;
                cs[r][t] = (hv - ys[r][t]) + ys[r][t];   // (y_hard - y).detach() + y, nn_model.py:492
            }
        }
    } else {
#pragma unroll
        for (int r = 0; r < NR; ++r)
#pragma unroll
            for (int t = 0; t < CP; ++t) cs[r][t] = ys[r][t];
    }
    // ---- store; c goes to the workgroup tile for the block statistics (zero rows beyond the batch)
    float kl_acc = 0.f, ent_acc = 0.f;
#pragma unroll
    for (int r = 0; r < NR; ++r) {
#pragma unroll
        for (int t = 0; t < CP; ++t) {
            const int col = sub + 32 * t;
            sh_buf[slot[r] * 128 + col] = (okr[r] && vcol[t]) ? cc[r][t] : 0.f;
            if (okr[r] && vcol[t]) {
                const int64_t o = (ab + bb[r]) * C + col;
                ws[a.CPROB + o] = cp[r][t];
                ws[a.CC + o] = cc[r][t];
                ws[a.YSOFT + o] = ys[r][t];
                ws[a.CSMP + o] = cs[r][t];
                ws[a.Y + (ab + bb[r]) * (L + C) + L + col] = cs[r][t];
                ws[a.ZIN + (ab + bb[r]) * (C + S) + col] = cs[r][t];
                ent_acc += cc[r][t] * lc[r][t];
            }
        }
    }
    // ---- state head: [mu | sigma_pre] = y [Wmu; Wsigma]^T + b
    float mso[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) mso[r] = 0.f;
    for (int o = 0; o < 2 * S; ++o) {
        const float* w = Wms + (int64_t)o * (L + C);
        const float wl = sub < L ? w[sub] : 0.f;
        float wc[CP], pr[NR];
#pragma unroll
        for (int t = 0; t < CP; ++t) wc[t] = vcol[t] ? w[L + sub + 32 * t] : 0.f;
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            pr[r] = xl[r] * wl;
#pragma unroll
            for (int t = 0; t < CP; ++t) pr[r] += cs[r][t] * wc[t];
        }
        const float bo = bms[o];
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const float pv = half_sum(pr[r]) + bo;
            if (sub == o) mso[r] = pv;
        }
    }
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        if (okr[r] && sub < 2 * S) ws[a.MS + (ab + bb[r]) * 2 * S + sub] = mso[r];
        const float sg = __shfl(mso[r], hb + ((sub + S) & 31), 64);
        if (okr[r] && sub < S) {
            const int b = bb[r];
            const float mu = mso[r];
            const float var = 1.f / (1.f + expf(-sg));
            const float lv = logf(var + eps);
            const float sd = sqrtf(expf(lv));
            const float U = state_u(nz, arm, B, S, b, sub);
            const float sv = U * sd + mu;
            float sin_ = sv;
            if (a.training && a.s_drop > 0.f) sin_ = state_keep(nz, arm, B, S, b, sub) ? sv / (1.f - a.s_drop) : 0.f;
            ws[a.MU + (ab + b) * S + sub] = mu;
            ws[a.LV + (ab + b) * S + sub] = lv;
            ws[a.SS + (ab + b) * S + sub] = sv;
            ws[a.ZIN + (ab + b) * (C + S) + C + sub] = sin_;
            kl_acc += 1.f + lv - mu * mu - expf(lv);
        }
    }
    // ---- block partials (as k_lat_fwd): two passes over the LDS tile, NT / 128 row groups
    kl_acc = wave_sum(kl_acc);
    ent_acc = wave_sum(ent_acc);
    if (lane == 0) { sh_red[wv][0] = kl_acc; sh_red[wv][1] = ent_acc; }
    lds_barrier();
    {
        constexpr int G = NT / 128, RPG = LAT_ROWS / G;
        const int col = threadIdx.x & 127, g = threadIdx.x >> 7;
        const int nv = min(LAT_ROWS, B - b0);
        const bool live = col < 32 * CP;              // columns the cells wrote
        float v[RPG];
        float s1 = 0.f;
#pragma unroll
        for (int i = 0; i < RPG; ++i) { v[i] = live ? sh_buf[(g + G * i) * 128 + col] : 0.f; s1 += v[i]; }
        sh_ps[g][col] = s1;
        lds_barrier();
        float tot = 0.f;
#pragma unroll
        for (int k = 0; k < G; ++k) tot += sh_ps[k][col];
        const float mean = tot / (float)nv;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < RPG; ++i) { const float d = v[i] - mean; q += (g + G * i < nv) ? d * d : 0.f; }
        lds_barrier();
        sh_ps[g][col] = q;
        lds_barrier();
        if (g == 0 && col < C) {
            float m2 = 0.f;
#pragma unroll
            for (int k = 0; k < G; ++k) m2 += sh_ps[k][col];
            if (a.acc_c >= 0) {
                acc_add_stats(reinterpret_cast<long long*>(ws + a.acc_c) + (int64_t)arm * ACC_SET_I64, col, (float)nv, mean, m2);
            } else {
                float* p = ws + a.c_part + (((int64_t)arm * gridDim.x + blk) * 2) * C;
                p[col] = mean;
                p[C + col] = m2;
            }
        }
    }
    if (threadIdx.x == 0) {
        float* p = ws + a.lat_part + ((int64_t)arm * gridDim.x + blk) * 2;
        float k0 = 0.f, k1 = 0.f;
#pragma unroll
        for (int w = 0; w < LH_NW; ++w) { k0 += sh_red[w][0]; k1 += sh_red[w][1]; }
        p[0] = k0;
        p[1] = k1;
    }
}

// ---------------------------------------------------------------------------------------------
// coupling: for every cell, over all arms.  u_a = log(c_a + eps) * iv_a
//   dist += sum_{a<b} |u_a - u_b|^2 ;  l2 += sum_{a<b} |c_smp_a - c_smp_b|^2
//   T_part[a][k] += G_a[k] * log(c_a[k] + eps),  G_a = (2 lam / B) (A u_a - sum_b u_b)
// grid (ceil(B/32)), 256 threads.
// ---------------------------------------------------------------------------------------------
// AT: the number of arms (a template parameter: every loop over arms is static, so that all loads of a batch of rows
// issue before the first use -- with a run-time arm count inside the row loop the kernel waited for every row in turn:
// 54 us).  c_acc / t_acc != null: the statistics of c come from the accumulator set the latent forward kernels added to,
// and this kernel adds its T sums to another (the latent backward reads W numbers; no reduction launch); else the
// partial arrays (c_part recombined here, T_part summed by k_loss_finalize).
template <int AT>
__global__ __launch_bounds__(256) void k_couple(int B, int C, float eps, float lam, const float* __restrict__ CCp,
                                                const float* __restrict__ CSMPp, const float* __restrict__ c_part, int c_n,
                                                const long long* __restrict__ c_acc, float* __restrict__ c_mean,
                                                float* __restrict__ c_iv, float* __restrict__ couple_part,
                                                float* __restrict__ T_part, long long* __restrict__ t_acc) {
    // (the arithmetic lives in couple.hpp: the decoder chain's launch of the fused train step runs it as one of its roles)
    __shared__ __attribute__((aligned(16))) float shT[4 * AT * CPL * 64];
    __shared__ float sh_red[8], sh_iv[AT * CPL * 64];
    __shared__ __attribute__((aligned(16))) float sh_scr[3 * PART_MAXG * CPL * 64];
    couple_body<AT>(blockIdx.x, true, threadIdx.x, B, C, eps, lam, CCp, CSMPp, c_part, c_n, c_acc, c_mean, c_iv, couple_part, T_part,
                    t_acc, shT, sh_red, sh_iv, sh_scr);
}

// ---------------------------------------------------------------------------------------------
// loss scalars (one block).  Sums the per-block partials in double.
// ---------------------------------------------------------------------------------------------
__device__ double block_sum_d(double v, double* sh) {
    const int tid = threadIdx.x;
    sh[tid] = v;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) sh[tid] += sh[tid + o];
        __syncthreads();
    }
    const double r = sh[0];
    __syncthreads();
    return r;
}

__global__ __launch_bounds__(256) void k_loss_finalize(int A, int B, int D, int C, float beta, float lam,
                                                       const float* __restrict__ fc11_part, int n11,
                                                       const float* __restrict__ lat_part, int nlat, int nblk,
                                                       const float* __restrict__ couple_part,
                                                       const float* __restrict__ T_part, float* __restrict__ T,
                                                       float* __restrict__ out, int mode) {
    // mode 0: block 0 the scalars, blocks 1.. the T sums; 1: T sums only (every block; they need the coupling kernel's
    // output alone and the latent backward waits for them); 2: scalars only (one block; they need fc11's loss partials)
    __shared__ double sh[256];
    const int tid = threadIdx.x;
    const int bx = mode == 1 ? (int)blockIdx.x + 1 : (int)blockIdx.x;
    if (bx > 0) {
        // blocks 1.. : T[a][k] = sum over row blocks of T_part[blk][a][k]  (8 block groups x 32 columns)
        const int c = tid & 31, g = tid >> 5, i = (bx - 1) * 32 + c, n = A * C;
        double s = 0.0;
        if (i < n)
            for (int b = g; b < nblk; b += 8) s += T_part[(int64_t)b * n + i];
        sh[g * 32 + c] = s;
        __syncthreads();
        if (g == 0 && i < n) {
            for (int k = 1; k < 8; ++k) s += sh[k * 32 + c];
            T[i] = (float)s;
        }
        return;
    }
    const double PI2 = 6.283185307179586;
    double sum_ind = 0.0, sum_ent = 0.0;
    for (int a = 0; a < A; ++a) {
        double se = 0.0, mm = 0.0, kl = 0.0, en = 0.0;
        for (int i = tid; i < n11; i += 256) {
            se += fc11_part[((int64_t)a * n11 + i) * 2];
            mm += fc11_part[((int64_t)a * n11 + i) * 2 + 1];
        }
        for (int i = tid; i < nlat; i += 256) {
            kl += lat_part[((int64_t)a * nlat + i) * 2];
            en += lat_part[((int64_t)a * nlat + i) * 2 + 1];
        }
        se = block_sum_d(se, sh); mm = block_sum_d(mm, sh); kl = block_sum_d(kl, sh); en = block_sum_d(en, sh);
        const double rec = 0.5 * se / B + 0.5 * (100.0 * mm / ((double)B * D));   // nn_model.py:544-546
        const double ll = se / ((double)B * D) + B * log(PI2);                     // :542
        const double klv = -0.5 * kl / B;                                          // :43-44
        if (tid == 0) {
            out[MMVAE_LOSS_REC0 + a] = (float)rec;
            out[MMVAE_LOSS_REC0 + A + a] = (float)klv;
            out[MMVAE_LOSS_REC0 + 2 * A + a] = (float)ll;
        }
        sum_ind += rec + beta * klv;
        sum_ent += en / B;
    }
    double ds = 0.0, l2 = 0.0;
    for (int i = tid; i < nblk; i += 256) { ds += couple_part[i * 2]; l2 += couple_part[i * 2 + 1]; }
    ds = block_sum_d(ds, sh) / B;
    l2 = block_sum_d(l2, sh) / B;
    const double npairs = A > 1 ? A * (A - 1) / 2.0 : 1.0;
    const double sum_c_ents = (A - 1) * sum_ent;   // every arm is in A-1 pairs
    const double joint = lam * ds + sum_c_ents + npairs * ((C / 2.0) * log(PI2) - 0.5 * log(2.0 * lam));   // :581-586
    const double total = (A > 1 ? A - 1 : 1) * sum_ind + joint;                                           // :587
    if (tid == 0) {
        out[MMVAE_LOSS_TOTAL] = (float)total;
        out[MMVAE_LOSS_JOINT] = (float)joint;
        out[MMVAE_LOSS_CENT] = (float)(sum_c_ents / npairs);
        out[MMVAE_LOSS_CDIST] = (float)(ds / npairs);
        out[MMVAE_LOSS_CL2] = (float)(l2 / npairs);
    }
}

// ---------------------------------------------------------------------------------------------
// backward of the latent block.  grid (ceil(B/LAT_ROWS_BWD), A), 64 * LATB_NW threads, one wave per cell at a time.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64 * LATB_NW) void k_lat_bwd(const LatArgs a_in, const NoiseDev nz_in,
                                                  const float* __restrict__ params, float* __restrict__ ws) {
    const LatArgs a = a_in;
    const NoiseDev nz = nz_in;
    extern __shared__ __attribute__((aligned(16))) float lat_smem[];
    __shared__ float sh_s[LATB_NW][2][64];
    const int arm = blockIdx.y, blk = blockIdx.x, b0 = blk * LAT_ROWS_BWD;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int A = a.A, B = a.B, L = a.L, C = a.C, S = a.S;
    const float* P = params + (int64_t)arm * a.per_arm;
    float* WcT = lat_smem;            // [L][C]
    float* Wms = lat_smem + C * L;    // [2S][L+C]
    lat_stage_weights(WcT, Wms, P + a.o_wc, P + a.o_wms, L, C, S);
    const int64_t ab = (int64_t)arm * B;
    const float eps = a.eps, invB = 1.f / (float)B;
    const float coefG = 2.f * a.lam * invB;

    float Tk[CPL], cmean[CPL], ivm[CPL], ivall[MMVAE_MAX_ARMS][CPL];
#pragma unroll
    for (int t = 0; t < CPL; ++t) {
        const int col = lane + 64 * t;
        Tk[t] = col < C ? ws[a.T + arm * C + col] : 0.f;
        if (a.acc_T >= 0 && col < C) {
            double s1, s2;
            acc_get(reinterpret_cast<const long long*>(ws + a.acc_T) + (int64_t)arm * ACC_SET_I64, col, s1, s2);
            Tk[t] = (float)s1;
        }
        cmean[t] = col < C ? ws[a.c_mean + arm * C + col] : 0.f;
        ivm[t] = col < C ? ws[a.c_iv + arm * C + col] : 0.f;
#pragma unroll
        for (int aa = 0; aa < MMVAE_MAX_ARMS; ++aa) ivall[aa][t] = (aa < A && col < C) ? ws[a.c_iv + aa * C + col] : 0.f;
    }
    float s1 = 0.f, s2 = 0.f;   // BN5 backward sums for column `lane` (< L)

    // The wave's LATB_NR cells side by side, every global load of both issued before the first use: beside the dW11
    // GEMM of the side stream this kernel's 4-byte loads queue behind the GEMM's in the CU's in-order memory pipe
    // (82 us in the step against 28 us alone when each cell's loads were issued and awaited one after the other).
    constexpr int NR = LATB_NR;
    int bb[NR];
    bool okr[NR];
    float gs_l[NR], mu_l[NR], lv_l[NR], sg_l[NR], xlow_l[NR];
    float cc[NR][CPL], ys[NR][CPL], gz[NR][CPL], cp[NR][CPL], call[NR][MMVAE_MAX_ARMS][CPL];
    const bool vcol[CPL] = {lane < C, lane + 64 < C};
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        bb[r] = b0 + wv + LATB_NW * r;
        okr[r] = bb[r] < B;                         // wave-uniform
        const int64_t b = min(bb[r], B - 1);        // cells beyond the batch recompute the last cell; nothing is stored
        gs_l[r] = lane < S ? ws[a.GZIN + (ab + b) * (C + S) + C + lane] : 0.f;
        mu_l[r] = lane < S ? ws[a.MU + (ab + b) * S + lane] : 0.f;
        lv_l[r] = lane < S ? ws[a.LV + (ab + b) * S + lane] : 0.f;
        sg_l[r] = lane < S ? ws[a.MS + (ab + b) * 2 * S + S + lane] : 0.f;
        xlow_l[r] = lane < L ? ws[a.XLOW + (ab + b) * L + lane] : 0.f;
#pragma unroll
        for (int t = 0; t < CPL; ++t) {
            const int col = lane + 64 * t;
            const int64_t o = (ab + b) * C + col;
            cc[r][t] = vcol[t] ? ws[a.CC + o] : 0.f;
            ys[r][t] = vcol[t] ? ws[a.YSOFT + o] : 0.f;
            cp[r][t] = vcol[t] ? ws[a.CPROB + o] : 0.f;
            gz[r][t] = vcol[t] ? ws[a.GZIN + (ab + b) * (C + S) + col] : 0.f;
#pragma unroll
            for (int aa = 0; aa < MMVAE_MAX_ARMS; ++aa)
                call[r][aa][t] = (aa < A && vcol[t]) ? ws[a.CC + ((int64_t)aa * B + b) * C + col] : 1.f;
        }
    }
    const float inv_temp = 1.f / a.temp, inv_tau = 1.f / a.tau, inv_bm1 = 1.f / (float)(B - 1);
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const int b = min(bb[r], B - 1);
        // ---- state head backward (lanes < S)
        float gms = 0.f;   // lane o < 2S: d loss / d MS[o]
        {
            float gmu = 0.f, gsig = 0.f;
            if (lane < S) {
                float gs = gs_l[r];
                if (a.training && a.s_drop > 0.f)
                    gs = state_keep(nz, arm, B, S, b, lane) ? gs / (1.f - a.s_drop) : 0.f;
                const float mu = mu_l[r], lv = lv_l[r], sg = sg_l[r];
                const float var = 1.f / (1.f + expf(-sg));
                const float U = state_u(nz, arm, B, S, b, lane);
                const float elv = expf(lv);
                gmu = gs + a.am1 * a.beta * mu * invB;
                const float glv = gs * U * 0.5f * sqrtf(elv) + a.am1 * a.beta * (-0.5f * invB) * (1.f - elv);
                const float gvar = glv / (var + eps);
                gsig = gvar * var * (1.f - var);
            }
            const float gsig_sh = __shfl(gsig, (lane - S) & 63, 64);   // lane S+s takes lane s's gsig
            if (lane < S) gms = gmu;
            else if (lane < 2 * S) gms = gsig_sh;
            if (okr[r] && lane < 2 * S) ws[a.GMS + (ab + b) * 2 * S + lane] = gms;
        }
        // ---- gy = gms [Wmu; Wsigma]
        float gxl = 0.f, gcs[CPL] = {0.f, 0.f};
        for (int o = 0; o < 2 * S; ++o) {
            const float go = __shfl(gms, o, 64);
            const float* w = Wms + (int64_t)o * (L + C);
            if (lane < L) gxl += go * w[lane];
#pragma unroll
            for (int t = 0; t < CPL; ++t) if (vcol[t]) gcs[t] += go * w[L + lane + 64 * t];
        }
        // ---- gradient w.r.t. the sample, through the Gumbel softmax to c
        float lc[CPL], gc[CPL], usum[CPL] = {0.f, 0.f}, rcc[CPL];
        float dot = 0.f;
#pragma unroll
        for (int t = 0; t < CPL; ++t) {
            lc[t] = gc[t] = 0.f;
            rcc[t] = 1.f / (cc[r][t] + eps);
            if (vcol[t]) {
                gcs[t] += gz[r][t];
                dot += ys[r][t] * gcs[t];
#pragma unroll
                for (int aa = 0; aa < MMVAE_MAX_ARMS; ++aa)
                    if (aa < A) {
                        const float l_aa = logf(call[r][aa][t] + eps);
                        usum[t] += l_aa * ivall[aa][t];
                        if (aa == arm) lc[t] = l_aa;      // this arm's own log c: the same value, computed once
                    }
            }
        }
        if (a.eval_flag) {
#pragma unroll
            for (int t = 0; t < CPL; ++t) gc[t] = gcs[t];
        } else {
            dot = wave_sum(dot);
#pragma unroll
            for (int t = 0; t < CPL; ++t) gc[t] = (ys[r][t] * (gcs[t] - dot) * inv_temp) * rcc[t];
        }
        // ---- coupling / entropy terms on c (nn_model.py:558-569)
        float dot2 = 0.f;
#pragma unroll
        for (int t = 0; t < CPL; ++t) {
            if (vcol[t]) {
                const float G = coefG * ((float)A * lc[t] * ivm[t] - usum[t]);
                gc[t] += (float)(A - 1) * (lc[t] + cc[r][t] * rcc[t]) * invB;
                gc[t] += G * ivm[t] * rcc[t];
                gc[t] += (Tk[t] * (-0.5f) * ivm[t] * ivm[t] * ivm[t]) * 2.f * (cc[r][t] - cmean[t]) * inv_bm1;
                dot2 += cc[r][t] * gc[t];
            } else {
                gc[t] = 0.f;
            }
        }
        dot2 = wave_sum(dot2);
        // ---- double softmax backward
        float gq[CPL], dot3 = 0.f;
#pragma unroll
        for (int t = 0; t < CPL; ++t) {
            gq[t] = cc[r][t] * (gc[t] - dot2) * inv_tau;
            dot3 += cp[r][t] * gq[t];
        }
        dot3 = wave_sum(dot3);
        float gzc[CPL];
#pragma unroll
        for (int t = 0; t < CPL; ++t) {
            gzc[t] = cp[r][t] * (gq[t] - dot3);
            if (okr[r] && vcol[t]) ws[a.GZC + (ab + b) * C + lane + 64 * t] = gzc[t];
        }
        // ---- g5 = gy[:, :L] + gzc Wc
        float g5 = 0.f;
        for (int k = 0; k < L; ++k) {
            float p = 0.f;
#pragma unroll
            for (int t = 0; t < CPL; ++t) if (vcol[t]) p += gzc[t] * WcT[k * C + lane + 64 * t];
            p = wave_sum(p);
            if (lane == k) g5 = gxl + p;
        }
        if (okr[r] && lane < L) {
            ws[a.G5 + (ab + b) * L + lane] = g5;
            s1 += g5;
            s2 += g5 * xlow_l[r];
        }
    }
    sh_s[wv][0][lane] = s1;
    sh_s[wv][1][lane] = s2;
    lds_barrier();
    if (threadIdx.x < L) {
        const int k = threadIdx.x;
        float* p = ws + a.bnb_part5 + (((int64_t)arm * gridDim.x + blk) * 2) * L;
        float k0 = 0.f, k1 = 0.f;
        for (int w = 0; w < LATB_NW; ++w) { k0 += sh_s[w][0][k]; k1 += sh_s[w][1][k]; }
        if (a.acc_bnb5 >= 0) {
            acc_add_sums(reinterpret_cast<long long*>(ws + a.acc_bnb5) + (int64_t)arm * ACC_SET_I64, k, k0, k1);
        } else {
            p[k] = k0;
            p[L + k] = k1;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Half-wave layout of the latent backward (same limits and reasons as k_lat_fwd_h): a cell owns 32 lanes x 3
// registers, a wave instruction serves two cells.  4 waves x 2 halves x 2 cells = LAT_ROWS_BWD cells per workgroup
// (the partial layout of k_lat_bwd).
// ---------------------------------------------------------------------------------------------
constexpr int LBH_NW = 4, LBH_NR = LAT_ROWS_BWD / (LBH_NW * 2);
static_assert(LBH_NR * LBH_NW * 2 == LAT_ROWS_BWD, "LAT_ROWS_BWD must be a multiple of 8");
__global__ __launch_bounds__(64 * LBH_NW) void k_lat_bwd_h(const LatArgs a_in, const NoiseDev nz_in,
                                                          const float* __restrict__ params, float* __restrict__ ws) {
    constexpr int NR = LBH_NR, CP = LH_CPL;
    const LatArgs a = a_in;
    const NoiseDev nz = nz_in;
    extern __shared__ __attribute__((aligned(16))) float lat_smem[];
    __shared__ float sh_s[LBH_NW][2][64];
    const int arm = blockIdx.y, blk = blockIdx.x, b0 = blk * LAT_ROWS_BWD;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int sub = lane & 31, hb = lane & 32, half = lane >> 5;
    const int A = a.A, B = a.B, L = a.L, C = a.C, S = a.S;
    const float* P = params + (int64_t)arm * a.per_arm;
    float* WcT = lat_smem;            // [L][C]
    float* Wms = lat_smem + C * L;    // [2S][L+C]
    lat_stage_weights(WcT, Wms, P + a.o_wc, P + a.o_wms, L, C, S);
    const int64_t ab = (int64_t)arm * B;
    const float eps = a.eps, invB = 1.f / (float)B;
    const float coefG = 2.f * a.lam * invB;

    bool vcol[CP];
    float Tk[CP], cmean[CP], ivm[CP], ivall[MMVAE_MAX_ARMS][CP];
#pragma unroll
    for (int t = 0; t < CP; ++t) {
        const int col = sub + 32 * t;
        vcol[t] = col < C;
        Tk[t] = vcol[t] ? ws[a.T + arm * C + col] : 0.f;
        if (a.acc_T >= 0 && vcol[t]) {
            double s1, s2;
            acc_get(reinterpret_cast<const long long*>(ws + a.acc_T) + (int64_t)arm * ACC_SET_I64, col, s1, s2);
            Tk[t] = (float)s1;
        }
        cmean[t] = vcol[t] ? ws[a.c_mean + arm * C + col] : 0.f;
        ivm[t] = vcol[t] ? ws[a.c_iv + arm * C + col] : 0.f;
#pragma unroll
        for (int aa = 0; aa < MMVAE_MAX_ARMS; ++aa) ivall[aa][t] = (aa < A && vcol[t]) ? ws[a.c_iv + aa * C + col] : 0.f;
    }
    float s1 = 0.f, s2 = 0.f;   // BN5 backward sums for column `sub` (< L), this half's cells

    int bb[NR];
    bool okr[NR];
    float gs_l[NR], mu_l[NR], lv_l[NR], sg_l[NR], xlow_l[NR];
    float cc[NR][CP], ys[NR][CP], gz[NR][CP], cp[NR][CP], call[NR][MMVAE_MAX_ARMS][CP];
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        bb[r] = b0 + r * (LBH_NW * 2) + wv * 2 + half;
        okr[r] = bb[r] < B;                         // per half-wave
        const int64_t b = min(bb[r], B - 1);        // cells beyond the batch recompute the last cell; nothing is stored
        gs_l[r] = sub < S ? ws[a.GZIN + (ab + b) * (C + S) + C + sub] : 0.f;
        mu_l[r] = sub < S ? ws[a.MU + (ab + b) * S + sub] : 0.f;
        lv_l[r] = sub < S ? ws[a.LV + (ab + b) * S + sub] : 0.f;
        sg_l[r] = sub < S ? ws[a.MS + (ab + b) * 2 * S + S + sub] : 0.f;
        xlow_l[r] = sub < L ? ws[a.XLOW + (ab + b) * L + sub] : 0.f;
#pragma unroll
        for (int t = 0; t < CP; ++t) {
            const int col = sub + 32 * t;
            const int64_t o = (ab + b) * C + col;
            cc[r][t] = vcol[t] ? ws[a.CC + o] : 0.f;
            ys[r][t] = vcol[t] ? ws[a.YSOFT + o] : 0.f;
            cp[r][t] = vcol[t] ? ws[a.CPROB + o] : 0.f;
            gz[r][t] = vcol[t] ? ws[a.GZIN + (ab + b) * (C + S) + col] : 0.f;
#pragma unroll
            for (int aa = 0; aa < MMVAE_MAX_ARMS; ++aa)
                call[r][aa][t] = (aa < A && vcol[t]) ? ws[a.CC + ((int64_t)aa * B + b) * C + col] : 1.f;
        }
    }
    const float inv_temp = 1.f / a.temp, inv_tau = 1.f / a.tau, inv_bm1 = 1.f / (float)(B - 1);
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const int b = min(bb[r], B - 1);
        // ---- state head backward (lanes sub < S of each half)
        float gms = 0.f;   // sub = o < 2S: d loss / d MS[o]
        {
            float gmu = 0.f, gsig = 0.f;
            if (sub < S) {
                float gs = gs_l[r];
                if (a.training && a.s_drop > 0.f)
                    gs = state_keep(nz, arm, B, S, b, sub) ? gs / (1.f - a.s_drop) : 0.f;
                const float mu = mu_l[r], lv = lv_l[r], sg = sg_l[r];
                const float var = 1.f / (1.f + expf(-sg));
                const float U = state_u(nz, arm, B, S, b, sub);
                const float elv = expf(lv);
                gmu = gs + a.am1 * a.beta * mu * invB;
                const float glv = gs * U * 0.5f * sqrtf(elv) + a.am1 * a.beta * (-0.5f * invB) * (1.f - elv);
                const float gvar = glv / (var + eps);
                gsig = gvar * var * (1.f - var);
            }
            const float gsig_sh = __shfl(gsig, hb + ((sub - S) & 31), 64);   // sub S+s takes sub s's gsig
            if (sub < S) gms = gmu;
            else if (sub < 2 * S) gms = gsig_sh;
            if (okr[r] && sub < 2 * S) ws[a.GMS + (ab + b) * 2 * S + sub] = gms;
        }
        // ---- gy = gms [Wmu; Wsigma]
        float gxl = 0.f, gcs[CP];
#pragma unroll
        for (int t = 0; t < CP; ++t) gcs[t] = 0.f;
        for (int o = 0; o < 2 * S; ++o) {
            const float go = __shfl(gms, hb + o, 64);
            const float* w = Wms + (int64_t)o * (L + C);
            if (sub < L) gxl += go * w[sub];
#pragma unroll
            for (int t = 0; t < CP; ++t) if (vcol[t]) gcs[t] += go * w[L + sub + 32 * t];
        }
        // ---- gradient w.r.t. the sample, through the Gumbel softmax to c
        float lc[CP], gc[CP], usum[CP], rcc[CP];
        float dot = 0.f;
#pragma unroll
        for (int t = 0; t < CP; ++t) {
            lc[t] = gc[t] = usum[t] = 0.f;
            rcc[t] = 1.f / (cc[r][t] + eps);
            if (vcol[t]) {
                gcs[t] += gz[r][t];
                dot += ys[r][t] * gcs[t];
#pragma unroll
                for (int aa = 0; aa < MMVAE_MAX_ARMS; ++aa)
                    if (aa < A) {
                        const float l_aa = logf(call[r][aa][t] + eps);
                        usum[t] += l_aa * ivall[aa][t];
                        if (aa == arm) lc[t] = l_aa;
                    }
            }
        }
        if (a.eval_flag) {
#pragma unroll
            for (int t = 0; t < CP; ++t) gc[t] = gcs[t];
        } else {
            dot = half_sum(dot);
#pragma unroll
            for (int t = 0; t < CP; ++t) gc[t] = (ys[r][t] * (gcs[t] - dot) * inv_temp) * rcc[t];
        }
        // ---- coupling / entropy terms on c (nn_model.py:558-569)
        float dot2 = 0.f;
#pragma unroll
        for (int t = 0; t < CP; ++t) {
            if (vcol[t]) {
                const float G = coefG * ((float)A * lc[t] * ivm[t] - usum[t]);
                gc[t] += (float)(A - 1) * (lc[t] + cc[r][t] * rcc[t]) * invB;
                gc[t] += G * ivm[t] * rcc[t];
                gc[t] += (Tk[t] * (-0.5f) * ivm[t] * ivm[t] * ivm[t]) * 2.f * (cc[r][t] - cmean[t]) * inv_bm1;
                dot2 += cc[r][t] * gc[t];
            } else {
                gc[t] = 0.f;
            }
        }
        dot2 = half_sum(dot2);
        // ---- double softmax backward
        float gq[CP], dot3 = 0.f;
#pragma unroll
        for (int t = 0; t < CP; ++t) {
            gq[t] = cc[r][t] * (gc[t] - dot2) * inv_tau;
            dot3 += cp[r][t] * gq[t];
        }
        dot3 = half_sum(dot3);
        float gzc[CP];
#pragma unroll
        for (int t = 0; t < CP; ++t) {
            gzc[t] = cp[r][t] * (gq[t] - dot3);
            if (okr[r] && vcol[t]) ws[a.GZC + (ab + b) * C + sub + 32 * t] = gzc[t];
        }
        // ---- g5 = gy[:, :L] + gzc Wc
        float g5 = 0.f;
        for (int k = 0; k < L; ++k) {
            float p = 0.f;
#pragma unroll
            for (int t = 0; t < CP; ++t) if (vcol[t]) p += gzc[t] * WcT[k * C + sub + 32 * t];
            p = half_sum(p);
            if (sub == k) g5 = gxl + p;
        }
        if (okr[r] && sub < L) {
            ws[a.G5 + (ab + b) * L + sub] = g5;
            s1 += g5;
            s2 += g5 * xlow_l[r];
        }
    }
    sh_s[wv][0][lane] = s1;
    sh_s[wv][1][lane] = s2;
    lds_barrier();
    if (threadIdx.x < L) {
        const int k = threadIdx.x;
        float* p = ws + a.bnb_part5 + (((int64_t)arm * gridDim.x + blk) * 2) * L;
        float k0 = 0.f, k1 = 0.f;
#pragma unroll
        for (int w = 0; w < LBH_NW; ++w) {
            k0 += sh_s[w][0][k] + sh_s[w][0][32 + k];      // the two halves hold different cells
            k1 += sh_s[w][1][k] + sh_s[w][1][32 + k];
        }
        if (a.acc_bnb5 >= 0) {
            acc_add_sums(reinterpret_cast<long long*>(ws + a.acc_bnb5) + (int64_t)arm * ACC_SET_I64, k, k0, k1);
        } else {
            p[k] = k0;
            p[L + k] = k1;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// slabs -> flat gradient buffer
// ---------------------------------------------------------------------------------------------
struct RedDesc {
    const float* slab; int64_t ks_stride, arm_stride; int ld, col0;
    int rows, cols;
    int64_t dst_off; int dst_ld;
    float scale;
    int ks;   // slabs to sum
};
constexpr int MAX_RED = 28;
struct RedDescs { RedDesc d[MAX_RED]; int first[MAX_RED + 1]; int n; };   // descriptor i owns blocks [first[i], first[i + 1]) of grid.x

struct AdamArgs {
    float* p; float* m; float* v;        // null p: gradients only
    float lr_bc1, inv_sqrt_bc2, b1, b2, eps, wd, lr;
    int decoupled;
};

// grid (blocks, ndesc, A): descriptor y, arm z.  With
// adam.p != null the Adam/AdamW update of the element is applied in the same pass (single-GPU step).
// HBM-bound (every slab element is read once): VEC threads own four consecutive columns (16-byte loads), the
// slab loads of an element are issued sixteen at a time before the first add (a `for k < KS` loop with a
// running sum waits for one memory latency per slab), and the index arithmetic is 32-bit.
__device__ __forceinline__ float adam_update1(const AdamArgs& adam, int64_t o, float gi) {
    float pi = adam.p[o];
    if (adam.wd != 0.f) {
        if (adam.decoupled) pi *= (1.f - adam.lr * adam.wd);
        else gi += adam.wd * pi;
    }
    const float mi = adam.b1 * adam.m[o] + (1.f - adam.b1) * gi;
    const float vi = adam.b2 * adam.v[o] + (1.f - adam.b2) * gi * gi;
    adam.m[o] = mi;
    adam.v[o] = vi;
    return pi - adam.lr_bc1 * (mi / (sqrtf(vi) * adam.inv_sqrt_bc2 + adam.eps));
}

template <bool VEC>
__device__ __forceinline__ void reduce_desc(const RedDesc& d, int KS, int arm, float* __restrict__ grads, int64_t per_arm,
                                            const AdamArgs& adam, uint32_t bx, uint32_t nbx) {
    constexpr int E = VEC ? 4 : 1;
    const uint32_t cpr = (uint32_t)d.cols / E;                       // work items per row
    const uint32_t n = (uint32_t)d.rows * cpr;
    const float* base = d.slab + (int64_t)arm * d.arm_stride + d.col0;
    for (uint32_t i = bx * blockDim.x + threadIdx.x; i < n; i += nbx * blockDim.x) {
        const uint32_t r = i / cpr, cidx = (i - r * cpr) * E;
        const float* p = base + (int64_t)r * d.ld + cidx;
        const int64_t o = (int64_t)arm * per_arm + d.dst_off + (int64_t)r * d.dst_ld + cidx;
        // the parameter and its moments are requested WITH the slabs (one memory round trip per item instead of two)
        float4 pi = make_float4(0.f, 0.f, 0.f, 0.f), mi = pi, vi = pi;
        if constexpr (VEC) {
            if (adam.p) {
                pi = *reinterpret_cast<const float4*>(adam.p + o);
                mi = *reinterpret_cast<const float4*>(adam.m + o);
                vi = *reinterpret_cast<const float4*>(adam.v + o);
            }
        }
        float s[E];
#pragma unroll
        for (int e = 0; e < E; ++e) s[e] = 0.f;
        // slab loads in flight per item: the smallest of 4 / 8 / 16 that covers KS (the clamped duplicates of the last slab
        // each cost a pass through the vector-memory pipe: 16 issued for 4 or 6 slabs were 2.7 - 4 x the loads needed);
        // the sum runs over the slabs in the same order whatever the chunk
        auto add_slabs = [&](auto chunk, int kbeg, int kend) __attribute__((always_inline)) {
            constexpr int CHK = decltype(chunk)::value;
            for (int k0 = kbeg; k0 < kend; k0 += CHK) {
                float v[CHK][E];
#pragma unroll
                for (int k = 0; k < CHK; ++k) {
                    const float* q = p + (int64_t)min(k0 + k, KS - 1) * d.ks_stride;
                    if constexpr (VEC) {
                        const float4 t = *reinterpret_cast<const float4*>(q);
                        v[k][0] = t.x; v[k][1] = t.y; v[k][2] = t.z; v[k][3] = t.w;
                    } else {
                        v[k][0] = *q;
                    }
                }
#pragma unroll
                for (int k = 0; k < CHK; ++k)
#pragma unroll
                    for (int e = 0; e < E; ++e) s[e] += (k0 + k < KS) ? v[k][e] : 0.f;
            }
        };
        // (whole sixteens first, then the smallest chunk that covers the rest: 20 slabs -- the small-layer products at the benchmark
        // shape -- are 16 + 4 loads, not 32)
        const int k16 = KS > 8 ? (KS & ~15) : 0, rest = KS - k16;
        if (k16 > 0) add_slabs(std::integral_constant<int, 16>{}, 0, k16);
        if (rest > 8) add_slabs(std::integral_constant<int, 16>{}, k16, KS);
        else if (rest > 4) add_slabs(std::integral_constant<int, 8>{}, k16, KS);
        else if (rest > 0) add_slabs(std::integral_constant<int, 4>{}, k16, KS);
        float g[E];
#pragma unroll
        for (int e = 0; e < E; ++e) g[e] = s[e] * d.scale;
        if constexpr (VEC) {
            *reinterpret_cast<float4*>(grads + o) = make_float4(g[0], g[1], g[2], g[3]);
            if (adam.p) {
                const float pin[4] = {pi.x, pi.y, pi.z, pi.w}, min_[4] = {mi.x, mi.y, mi.z, mi.w}, vin[4] = {vi.x, vi.y, vi.z, vi.w};
                float po[4], mo[4], vo[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float pe = pin[e], ge = g[e];
                    if (adam.wd != 0.f) {
                        if (adam.decoupled) pe *= (1.f - adam.lr * adam.wd);
                        else ge += adam.wd * pe;
                    }
                    mo[e] = adam.b1 * min_[e] + (1.f - adam.b1) * ge;
                    vo[e] = adam.b2 * vin[e] + (1.f - adam.b2) * ge * ge;
                    po[e] = pe - adam.lr_bc1 * (mo[e] / (sqrtf(vo[e]) * adam.inv_sqrt_bc2 + adam.eps));
                }
                *reinterpret_cast<float4*>(adam.m + o) = make_float4(mo[0], mo[1], mo[2], mo[3]);
                *reinterpret_cast<float4*>(adam.v + o) = make_float4(vo[0], vo[1], vo[2], vo[3]);
                *reinterpret_cast<float4*>(adam.p + o) = make_float4(po[0], po[1], po[2], po[3]);
            }
        } else {
            grads[o] = g[0];
            if (adam.p) adam.p[o] = adam_update1(adam, o, g[0]);
        }
    }
}

__global__ __launch_bounds__(256) void k_reduce(const RedDescs ds, float* __restrict__ grads, int64_t per_arm,
                                                const AdamArgs adam_in) {
    // grid (blocks of all descriptors, 1, A): large and small tensors in ONE launch -- the three D x H tensors want thousands of
    // workgroups, the 23 small ones sixteen each, and as two launches the small one (latency-bound) cost as much as the large
    int di = 0;
    while (di + 1 < ds.n && (int)blockIdx.x >= ds.first[di + 1]) ++di;
    const int arm = blockIdx.z;
    const uint32_t bx = blockIdx.x - ds.first[di], nbx = ds.first[di + 1] - ds.first[di];
    const RedDesc& dr = ds.d[di];
    const RedDesc d = {dr.slab, dr.ks_stride, dr.arm_stride, dr.ld, dr.col0, dr.rows, dr.cols, dr.dst_off, dr.dst_ld, dr.scale, dr.ks};
    const AdamArgs adam = adam_in;
    const int KS = d.ks;
    // 16-byte path: four-column groups aligned in every slab, in the gradient and in the parameter / moment buffers
    const bool vec = ((d.cols | d.ld | d.col0 | d.dst_ld) & 3) == 0 && ((d.ks_stride | d.arm_stride | d.dst_off | per_arm) & 3) == 0 &&
                     ((reinterpret_cast<uintptr_t>(d.slab) | reinterpret_cast<uintptr_t>(grads) |
                       reinterpret_cast<uintptr_t>(adam.p) | reinterpret_cast<uintptr_t>(adam.m) |
                       reinterpret_cast<uintptr_t>(adam.v)) & 15) == 0;
    if (vec) reduce_desc<true>(d, KS, arm, grads, per_arm, adam, bx, nbx);
    else reduce_desc<false>(d, KS, arm, grads, per_arm, adam, bx, nbx);
}

__global__ void k_adam(int64_t n, float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                       float* __restrict__ v, float lr_bc1, float inv_sqrt_bc2, float b1, float b2, float eps,
                       float wd, float lr, int decoupled) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float pi = p[i], gi = g[i];
        if (wd != 0.f) {
            if (decoupled) pi *= (1.f - lr * wd);
            else gi += wd * pi;
        }
        const float mi = b1 * m[i] + (1.f - b1) * gi;
        const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        const float denom = sqrtf(vi) * inv_sqrt_bc2 + eps;
        p[i] = pi - lr_bc1 * (mi / denom);
    }
}

__global__ void k_dump_noise(NoiseDev nz, int A, int B, int D, int C, int S, uint8_t* x_mask, float* u_gumbel,
                             float* u_state, uint8_t* s_mask) {
    const int64_t nx = (int64_t)A * B * D, ng = (int64_t)A * B * C, ns = (int64_t)A * B * S;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t i0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (x_mask)
        for (int64_t i = i0; i < nx; i += stride) {
            const int arm = (int)(i / ((int64_t)B * D));
            const int64_t e = i % ((int64_t)B * D);
            x_mask[i] = xmask_keep(nz, arm, (int)(e / D), (int)(e % D)) ? 1 : 0;
        }
    if (u_gumbel)
        for (int64_t i = i0; i < ng; i += stride) {
            const int arm = (int)(i / ((int64_t)B * C));
            u_gumbel[i] = noise_uniform(nz, arm, STREAM_GUMBEL, (uint64_t)(i % ((int64_t)B * C)));
        }
    for (int64_t i = i0; i < ns; i += stride) {
        const int arm = (int)(i / ((int64_t)B * S));
        if (u_state) u_state[i] = noise_uniform(nz, arm, STREAM_STATE, (uint64_t)(i % ((int64_t)B * S)));
        if (s_mask) s_mask[i] = noise_keep(nz, arm, STREAM_SMASK, (uint64_t)(i % ((int64_t)B * S)), nz.s_keep_thr) ? 1 : 0;
    }
}

// ---------------------------------------------------------------------------------------------
// host launchers
// ---------------------------------------------------------------------------------------------
static LatArgs make_lat_args(const Ctx& c) {
    const mmvae_dims& d = c.d;
    const Layout& L = c.lay;
    LatArgs a{};
    a.A = d.A; a.B = d.B; a.L = d.L; a.C = d.C; a.S = d.S;
    a.tau = c.h.tau; a.temp = c.h.temp; a.eps = c.h.eps; a.s_drop = c.h.s_drop;
    a.use_mask = (c.h.cat_mask[0] | c.h.cat_mask[1] | c.h.cat_mask[2] | c.h.cat_mask[3]) != 0u;
    for (int i = 0; i < 4; ++i) a.cmask[i] = c.h.cat_mask[i];
    a.hard = c.h.hard; a.training = c.h.training; a.eval_flag = c.h.eval_flag;
    a.per_arm = c.po.per_arm; a.o_wc = c.po.o[10]; a.o_bc = c.po.o[11]; a.o_wms = c.po.o[12]; a.o_bms = c.po.o[14];
    a.R5 = L.R[4]; a.mean5 = L.bn_mean[4]; a.rstd5 = L.bn_rstd[4];
    a.XLOW = L.XLOW; a.CPROB = L.CPROB; a.CC = L.CC; a.YSOFT = L.YSOFT; a.CSMP = L.CSMP; a.Y = L.Y; a.MS = L.MS;
    a.MU = L.MU; a.LV = L.LV; a.SS = L.SS; a.ZIN = L.ZIN; a.c_part = L.c_part; a.lat_part = L.lat_part;
    a.GZIN = L.GZIN; a.GMS = L.GMS; a.GZC = L.GZC; a.G5 = L.G[5]; a.bnb_part5 = L.bnb_part[5];
    a.T = L.T; a.c_mean = L.c_mean; a.c_iv = L.c_iv;
    a.am1 = (float)(d.A > 1 ? d.A - 1 : 1); a.beta = c.h.beta; a.lam = c.h.lam;
    a.bn_part5 = c.h.training ? L.bn_part[4] : -1;
    a.bn5_n = L.nblkf;   // (partial-array form: chain_rows_fwd == CHAIN_ROWS)
    a.acc_bn5 = (c.h.training && c.use_acc()) ? acc_set_off(L, d.A, 4) : -1;
    a.acc_bnb5 = c.use_acc() ? acc_set_off(L, d.A, ACC_BWD + 4) : -1;
    a.acc_c = (c.h.training && c.use_acc()) ? acc_set_off(L, d.A, ACC_C) : -1;
    a.acc_T = (c.h.training && c.use_acc()) ? acc_set_off(L, d.A, ACC_T) : -1;
    a.run_mean_off = c.po.bn_mean[4]; a.run_var_off = c.po.bn_var[4]; a.run_arm_stride = c.po.bn_per_arm;
    a.bn_momentum = c.h.bn_momentum;
    const int abl = c.tune(MMVAE_TUNE_ABLATE_L);
    a.dbg_off = (abl & 8) ? L.loss_scratch + 2048 : -1;
    return a;
}

// eval mode: BatchNorm `layer` (0..4) normalises with the running buffers; in training mode the kernel that
// consumes the layer recombines the batch statistics itself and nothing is launched here
int launch_bn_eval_stats(const Ctx& c, const float* bn_running) {
    const mmvae_dims& d = c.d;
    const Layout& L = c.lay;
    if (c.h.training) return 0;
    if (!bn_running) { set_error("eval-mode forward needs bn_running"); return MMVAE_E_BADARG; }
    EvalStatArgs a{};
    for (int layer = 0; layer < 5; ++layer) {
        a.run_mean[layer] = c.po.bn_mean[layer];
        a.run_var[layer] = c.po.bn_var[layer];
        a.mean_out[layer] = L.bn_mean[layer];
        a.rstd_out[layer] = L.bn_rstd[layer];
        a.W[layer] = (layer == 4) ? d.L : d.H;
    }
    hipLaunchKernelGGL(k_stats_from_running, dim3(d.A, 5), dim3(128), 0, c.stream, bn_running, c.po.bn_per_arm, a, c.h.eps, c.ws);
    HIP_LAUNCH_CHECK("k_stats_from_running");
    return 0;
}

int launch_lat_fwd(const Ctx& c, const mmvae_noise* nz, const float* params, float* bn_running, int64_t* nbt,
                   int32_t* labels) {
    LatArgs a = make_lat_args(c);
    a.labels = labels;
    NoiseDev nd = make_noise_dev(nz, c.h);
    const size_t shm = (size_t)(c.d.C * c.d.L + 2 * c.d.S * (c.d.L + c.d.C)) * sizeof(float);
    const int fullwave = 0;
    if (!fullwave && a.dbg_off < 0 && c.d.C <= 32 * LH_CPL && c.d.L <= 32 && 2 * c.d.S <= 32) {
        launch_k(c, k_lat_fwd_h, dim3(c.lay.nblkl, c.d.A), dim3(64 * LH_NW), shm, a, nd, params, c.ws, bn_running, nbt);
        HIP_LAUNCH_CHECK("k_lat_fwd_h");
        return 0;
    }
    hipLaunchKernelGGL(k_lat_fwd, dim3(c.lay.nblkl, c.d.A), dim3(64 * LAT_NW), shm, c.stream, a, nd, params, c.ws,
                       bn_running, nbt);
    HIP_LAUNCH_CHECK("k_lat_fwd");
    return 0;
}

int launch_couple(const Ctx& c) {
    const mmvae_dims& d = c.d;
    const Layout& L = c.lay;
    const bool acc = c.h.training && c.use_acc();
    long long* t_acc = acc ? reinterpret_cast<long long*>(c.ws + acc_set_off(L, d.A, ACC_T)) : nullptr;
    const long long* c_acc = acc ? reinterpret_cast<const long long*>(c.ws + acc_set_off(L, d.A, ACC_C)) : nullptr;
    if (acc) {   // the T sums start at zero every time the coupling runs (it may run more than once per forward pass)
        hipError_t e = hipMemsetAsync(t_acc, 0, sizeof(float) * (size_t)d.A * ACC_SET_FLOATS, c.stream);
        if (e != hipSuccess) { set_error("memset: %s", hipGetErrorString(e)); return MMVAE_E_LAUNCH; }
    }
#define MMVAE_COUPLE(AT)                                                                                               \
    hipLaunchKernelGGL(k_couple<AT>, dim3(L.nblk32), dim3(256), 0, c.stream, d.B, d.C, c.h.eps, c.h.lam, c.ws + L.CC,    \
                       c.ws + L.CSMP, c.ws + L.c_part, L.nblkl, c_acc, c.ws + L.c_mean, c.ws + L.c_iv,                    \
                       c.ws + L.couple_part, c.ws + L.T_part, t_acc)
    switch (d.A) {
        case 1: MMVAE_COUPLE(1); break;
        case 2: MMVAE_COUPLE(2); break;
        case 3: MMVAE_COUPLE(3); break;
        case 4: MMVAE_COUPLE(4); break;
        case 5: MMVAE_COUPLE(5); break;
        case 6: MMVAE_COUPLE(6); break;
        case 7: MMVAE_COUPLE(7); break;
        default: MMVAE_COUPLE(8); break;
    }
#undef MMVAE_COUPLE
    HIP_LAUNCH_CHECK("k_couple");
    return 0;
}

int launch_loss_finalize(const Ctx& c, float* loss_out, int mode) {
    const mmvae_dims& d = c.d;
    const Layout& L = c.lay;
    if (c.h.training && c.use_acc()) {   // the coupling kernel has added the T sums to their accumulator set itself
        if (mode == 1) return 0;
        mode = 2;
    }
    const int nT = cdiv(d.A * d.C, 32);
    hipLaunchKernelGGL(k_loss_finalize, dim3(mode == 0 ? 1 + nT : (mode == 1 ? nT : 1)), dim3(256), 0, c.stream, d.A, d.B, d.D, d.C,
                       c.h.beta, c.h.lam, c.ws + L.fc11_part, L.n11, c.ws + L.lat_part, L.nblkl, L.nblk32, c.ws + L.couple_part,
                       c.ws + L.T_part, c.ws + L.T, loss_out, mode);
    HIP_LAUNCH_CHECK("k_loss_finalize");
    return 0;
}

int launch_lat_bwd(const Ctx& c, const mmvae_noise* nz, const float* params) {
    LatArgs a = make_lat_args(c);
    NoiseDev nd = make_noise_dev(nz, c.h);
    const size_t shm = (size_t)(c.d.C * c.d.L + 2 * c.d.S * (c.d.L + c.d.C)) * sizeof(float);
    const int fullwave = 0;
    if (!fullwave && c.d.C <= 32 * LH_CPL && c.d.L <= 32 && 2 * c.d.S <= 32) {
        hipLaunchKernelGGL(k_lat_bwd_h, dim3(cdiv(c.d.B, LAT_ROWS_BWD), c.d.A), dim3(64 * LBH_NW), shm, c.stream, a, nd, params, c.ws);
        HIP_LAUNCH_CHECK("k_lat_bwd_h");
        return 0;
    }
    hipLaunchKernelGGL(k_lat_bwd, dim3(cdiv(c.d.B, LAT_ROWS_BWD), c.d.A), dim3(64 * LATB_NW), shm, c.stream, a, nd, params, c.ws);
    HIP_LAUNCH_CHECK("k_lat_bwd");
    return 0;
}

// which: bit 0 = fc11.weight / fc11.bias (final as soon as the dW11 GEMM is), bit 1 = everything else
int launch_reduce_grads(const Ctx& c, float* grads, float gscale, const AdamHost* ah, bool dw11_fast, int which) {
    const mmvae_dims& d = c.d;
    const Layout& L = c.lay;
    const int A = d.A, H = d.H, D = d.D, Ld = d.L, C = d.C, S = d.S;
    RedDescs ds{};
    int n = 0;
    const float xscale = (c.h.training && c.h.x_drop > 0.f) ? 1.f / (1.f - c.h.x_drop) : 1.f;
    // big: fc1.w, fc11.w, fc11.b (the general path's dW11 uses ks_dw slabs, the fast path's its own count)
    const int ks11 = dw11_fast ? L.sp.ks_dw11 : L.sp.ks_dw;
    ds.d[n++] = RedDesc{c.ws + L.dw1_slab, (int64_t)A * H * D, (int64_t)H * D, D, 0, H, D, c.po.o[0], D, gscale * xscale, L.sp.ks_dw};
    ds.d[n++] = RedDesc{c.ws + L.dw11_slab, (int64_t)A * D * DW11_LD, (int64_t)D * DW11_LD, DW11_LD, 0, D, H, c.po.o[26], H, gscale, ks11};
    ds.d[n++] = RedDesc{c.ws + L.dw11_slab, (int64_t)A * D * DW11_LD, (int64_t)D * DW11_LD, DW11_LD, H, D, 1, c.po.o[27], 1, gscale, ks11};
    const int nbig = n;
    const int64_t sks = (int64_t)A * N_SMALL * NP * SMALL_LD, sarm = (int64_t)N_SMALL * NP * SMALL_LD;
    auto small = [&](int i, int N, int K, int64_t w_off, int64_t b_off) {
        const float* s = c.ws + L.small_slab + (int64_t)i * NP * SMALL_LD;
        if (K > 0) ds.d[n++] = RedDesc{s, sks, sarm, SMALL_LD, 0, N, K, w_off, K, gscale, L.sp.ks_small};
        ds.d[n++] = RedDesc{s, sks, sarm, SMALL_LD, K, N, 1, b_off, 1, gscale, L.sp.ks_small};
    };
    small(0, H, H, c.po.o[2], c.po.o[3]);
    small(1, H, H, c.po.o[4], c.po.o[5]);
    small(2, H, H, c.po.o[6], c.po.o[7]);
    small(3, Ld, H, c.po.o[8], c.po.o[9]);
    small(4, C, Ld, c.po.o[10], c.po.o[11]);
    small(5, 2 * S, Ld + C, c.po.o[12], c.po.o[14]);   // fc_mu / fc_sigma are adjacent in the flat layout
    small(6, Ld, C + S, c.po.o[16], c.po.o[17]);
    small(7, H, Ld, c.po.o[18], c.po.o[19]);
    small(8, H, H, c.po.o[20], c.po.o[21]);
    small(9, H, H, c.po.o[22], c.po.o[23]);
    small(10, H, H, c.po.o[24], c.po.o[25]);
    small(11, H, 0, 0, c.po.o[1]);                     // fc1.b = column sums of dZ1
    AdamArgs aa{};
    if (ah && ah->p) {
        const double bc1 = 1.0 - pow((double)ah->b1, (double)ah->step);
        const double bc2 = 1.0 - pow((double)ah->b2, (double)ah->step);
        aa = AdamArgs{ah->p, ah->m, ah->v, (float)(ah->lr / bc1), (float)(1.0 / sqrt(bc2)), ah->b1, ah->b2, ah->eps,
                      ah->wd, ah->lr, ah->decoupled};
    }
    // which: 1 = the fc11 tensors (behind dW11, on whatever stream that ran), 2 = fc1.w and the small tensors, 3 = everything
    const int64_t big_elems = (int64_t)max(H, 1) * D;
    const int gx = (int)imin64(2048, cdiv64(big_elems / 4, 256));
    RedDescs out{};
    int no = 0, blocks = 0;
    auto add = [&](const RedDesc& r, int nb) { out.d[no] = r; out.first[no++] = blocks; blocks += nb; };
    if (which & 2) add(ds.d[0], gx);
    if (which & 1) { add(ds.d[1], gx); add(ds.d[2], cdiv(D, 256)); }
    if (which & 2)
        for (int i = nbig; i < n; ++i) add(ds.d[i], 16);
    out.first[no] = blocks;
    out.n = no;
    if (no) hipLaunchKernelGGL(k_reduce, dim3(blocks, 1, A), dim3(256), 0, c.stream, out, grads, c.po.per_arm, aa);
    HIP_LAUNCH_CHECK("k_reduce");
    return 0;
}

int launch_adam(int64_t n, float* p, const float* g, float* m, float* v, int64_t step, float lr, float b1, float b2,
                float eps, float wd, int decoupled, hipStream_t s) {
    const double bc1 = 1.0 - pow((double)b1, (double)step);
    const double bc2 = 1.0 - pow((double)b2, (double)step);
    const int blocks = (int)imin64(4096, cdiv64(n, 256));
    hipLaunchKernelGGL(k_adam, dim3(blocks), dim3(256), 0, s, n, p, g, m, v, (float)(lr / bc1), (float)(1.0 / sqrt(bc2)),
                       b1, b2, eps, wd, lr, decoupled);
    HIP_LAUNCH_CHECK("k_adam");
    return 0;
}

int launch_dump_noise(const mmvae_dims& d, const mmvae_hyper& h, const mmvae_noise* nz, uint8_t* x_mask,
                      float* u_gumbel, float* u_state, uint8_t* s_mask, hipStream_t s) {
    NoiseDev nd = make_noise_dev(nz, h);
    hipLaunchKernelGGL(k_dump_noise, dim3(2048), dim3(256), 0, s, nd, d.A, d.B, d.D, d.C, d.S, x_mask, u_gumbel, u_state,
                       s_mask);
    HIP_LAUNCH_CHECK("k_dump_noise");
    return 0;
}

}  // namespace mmvae
