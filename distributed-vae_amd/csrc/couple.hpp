// The pairwise coupling terms over arms (nn_model.py:558-569) as a device function: k_couple (rowwise.hip) runs it as a
// launch of its own, k_chain_fwd_couple (chain.hip) as a ROLE of the decoder chain's launch in the fused train step -- the
// decoder chain leaves a third of the CUs idle, and as part of that launch the coupling needs neither a fork to the side
// stream behind the latent forward nor a join in front of the latent backward (two 5 - 6 us bubbles on the main stream).
#pragma once
#include "common.hpp"

namespace mmvae {

constexpr int CPL = 2;   // categories per lane: C <= 128

// One 256-thread group (t256 = 0 .. 255; four waves) handles the 32 cells of row block `blk` (valid: blk exists -- a group
// without a block still takes part in the barriers).  Shared memory, per group: shT [4][AT][CPL * 64], sh_red [8],
// sh_iv [AT][CPL * 64], and for the partial-array path sh_scr [3 * PART_MAXG * CPL * 64].
//   u_a = log(c_a + eps) * iv_a ;  dist += sum_{a<b} |u_a - u_b|^2 ;  l2 += sum_{a<b} |c_smp_a - c_smp_b|^2
//   T[a][k] += G_a[k] * log(c_a[k] + eps),  G_a = (2 lam / B) (A u_a - sum_b u_b)
// AT: the number of arms (a template parameter: every loop over arms is static, so that all loads of a batch of rows
// issue before the first use).  c_acc / t_acc != null: the statistics of c come from the accumulator set the latent forward
// kernels added to, and the T sums are added to another (the latent backward reads W numbers); else the partial arrays.
template <int AT>
__device__ __forceinline__ void couple_body(int blk, bool valid, int t256, int B, int C, float eps, float lam,
                                            const float* __restrict__ CCp, const float* __restrict__ CSMPp,
                                            const float* __restrict__ c_part, int c_n, const long long* __restrict__ c_acc,
                                            float* __restrict__ c_mean, float* __restrict__ c_iv, float* __restrict__ couple_part,
                                            float* __restrict__ T_part, long long* __restrict__ t_acc, float* __restrict__ shT,
                                            float* __restrict__ sh_red, float* __restrict__ sh_iv, float* __restrict__ sh_scr) {
    constexpr int A = AT, W = CPL * 64;
    const int b0 = blk * 32;
    const int lane = t256 & 63, wv = t256 >> 6;
    // inv_var of every arm's c over the batch (nn_model.py:558-560): mean and unbiased variance; row block 0 keeps them
    // for the backward
    if (c_acc) {
        if (t256 < C) {
#pragma unroll
            for (int aa = 0; aa < A; ++aa) {
                float mean, m2;
                acc_mean_m2(c_acc + (int64_t)aa * ACC_SET_I64, t256, B, mean, m2);
                const float ivv = sqrtf(1.0f / (m2 / (float)(B - 1) + eps));
                sh_iv[aa * W + t256] = ivv;
                if (blk == 0 && valid) { c_mean[aa * C + t256] = mean; c_iv[aa * C + t256] = ivv; }
            }
        }
        lds_barrier();
    } else {
        for (int aa = 0; aa < A; ++aa) {
            float mean, m2;
            stats_from_partials<256>(c_part + (int64_t)aa * c_n * 2 * C, c_n, B, LAT_ROWS, C, sh_scr, mean, m2);
            if (t256 < C) {
                const float ivv = sqrtf(1.0f / (m2 / (float)(B - 1) + eps));
                sh_iv[aa * W + t256] = ivv;
                if (blk == 0 && valid) { c_mean[aa * C + t256] = mean; c_iv[aa * C + t256] = ivv; }
            }
            lds_barrier();
        }
    }
    float iv[AT][CPL], Tacc[AT][CPL];
    bool vc[CPL];
    int colc[CPL];
#pragma unroll
    for (int t = 0; t < CPL; ++t) {
        const int col = lane + 64 * t;
        vc[t] = col < C;
        colc[t] = min(col, C - 1);
#pragma unroll
        for (int aa = 0; aa < A; ++aa) { iv[aa][t] = vc[t] ? sh_iv[aa * W + col] : 0.f; Tacc[aa][t] = 0.f; }
    }
    float dist = 0.f, l2 = 0.f;
    const float coefG = 2.f * lam / (float)B;
    // a wave owns rows wv, wv + 4, ..., wv + 28 of the block; RB rows at a time, every load of the batch requested (clamped
    // addresses, no branches) before the first logarithm
    constexpr int RB = AT <= 2 ? 4 : (AT <= 4 ? 2 : 1);
    for (int i0 = 0; i0 < 8; i0 += RB) {
        float ccv[RB][AT][CPL], csv[RB][AT][CPL];
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
            const int bc = min(b0 + wv + 4 * (i0 + rb), B - 1);
#pragma unroll
            for (int aa = 0; aa < A; ++aa)
#pragma unroll
                for (int t = 0; t < CPL; ++t) {
                    const int64_t o = ((int64_t)aa * B + bc) * C + colc[t];
                    ccv[rb][aa][t] = CCp[o];
                    csv[rb][aa][t] = CSMPp[o];
                }
        }
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
            const bool rok = b0 + wv + 4 * (i0 + rb) < B;     // wave-uniform
            float u[AT][CPL], lc[AT][CPL], us[CPL];
#pragma unroll
            for (int t = 0; t < CPL; ++t) {
                us[t] = 0.f;
#pragma unroll
                for (int aa = 0; aa < A; ++aa) {
                    lc[aa][t] = (vc[t] && rok) ? logf(ccv[rb][aa][t] + eps) : 0.f;
                    u[aa][t] = lc[aa][t] * iv[aa][t];
                    us[t] += u[aa][t];
                }
            }
#pragma unroll
            for (int aa = 0; aa < A; ++aa) {
#pragma unroll
                for (int t = 0; t < CPL; ++t) {
                    const float G = coefG * ((float)A * u[aa][t] - us[t]);
                    Tacc[aa][t] += G * lc[aa][t];
                }
#pragma unroll
                for (int bb = aa + 1; bb < A; ++bb)
#pragma unroll
                    for (int t = 0; t < CPL; ++t) {
                        const float du = u[aa][t] - u[bb][t];
                        const float dc = (vc[t] && rok) ? csv[rb][aa][t] - csv[rb][bb][t] : 0.f;
                        dist += du * du;
                        l2 += dc * dc;
                    }
            }
        }
    }
    dist = wave_sum(dist);
    l2 = wave_sum(l2);
#pragma unroll
    for (int aa = 0; aa < A; ++aa)
#pragma unroll
        for (int t = 0; t < CPL; ++t) shT[(wv * AT + aa) * W + lane + 64 * t] = Tacc[aa][t];
    if (lane == 0) { sh_red[wv * 2] = dist; sh_red[wv * 2 + 1] = l2; }
    lds_barrier();
    if (valid) {
        for (int i = t256; i < A * C; i += 256) {
            const int aa = i / C, col = i % C;
            const float tsum = shT[(0 * AT + aa) * W + col] + shT[(1 * AT + aa) * W + col] + shT[(2 * AT + aa) * W + col] + shT[(3 * AT + aa) * W + col];
            if (t_acc) acc_add(t_acc + (int64_t)aa * ACC_SET_I64, 0, col, (double)tsum);
            else T_part[((int64_t)blk * A + aa) * C + col] = tsum;
        }
        if (t256 == 0) {
            couple_part[blk * 2] = sh_red[0] + sh_red[2] + sh_red[4] + sh_red[6];
            couple_part[blk * 2 + 1] = sh_red[1] + sh_red[3] + sh_red[5] + sh_red[7];
        }
    }
}

}  // namespace mmvae
