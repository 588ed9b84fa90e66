// extern "C" surface of libmmvae_hip.so (see include/mmvae.h) and the per-step launch sequence.
#include "common.hpp"
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

namespace mmvae {

static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// Split factors are chosen so that each large kernel's grid fills the chip's resident-workgroup
// slots once (256 CUs x workgroups per CU that its registers / LDS admit) without spilling into a
// second, mostly empty round: e.g. 624 workgroups on 512 slots take two rounds, 468 take one.
Splits default_splits(const mmvae_dims& d, const mmvae_exec* ex) {
    int g_split[6] = {0, 0, 0, 0, 0, 0};
    if (ex) for (int i = 0; i < 6; ++i) g_split[i] = ex->split[i] > 0 && ex->split[i] <= 64 ? ex->split[i] : 0;
    constexpr int CUS = 256;
    // smallest split whose grid fills whole rounds of the resident slots to >= 93 % (else the best-filling one); an exact
    // fill one or two steps further is preferred.  With A = 5 a single round would leave a fifth of the chip idle (200 k
    // workgroups on 512 slots: k = 2 -> 78 %, k = 5 -> 98 % in two rounds; fc1 64 -> 53 us per arm); 720 workgroups on
    // 768 slots measured 12 % slower than 768 (small-layer dW GEMMs).
    auto fit = [](int base_blocks, int slots, int cap) {
        const int base = base_blocks > 0 ? base_blocks : 1;
        auto eff = [&](int k) { const int n = base * k; return (double)n / (double)(((n + slots - 1) / slots) * slots); };
        int pick = 1;
        double best = 0.0;
        for (int k = 1; k <= cap; ++k) {
            if (eff(k) >= 0.93) { pick = k; break; }
            if (eff(k) > best + 1e-9) { best = eff(k); pick = k; }
        }
        for (int k = pick + 1; k <= cap && k <= pick + 2; ++k)
            if (eff(k) >= 0.995) return k;
        return pick;
    };
    Splits s;
    const int nb128 = cdiv(d.B, 128), nb64 = cdiv(d.B, 64);
    const bool fastdims = (d.D & 3) == 0 && (d.H & 3) == 0;
    // fc1 forward: fast kernel 128-row blocks, 3 workgroups / CU; general kernel 64-row blocks
    // (fc_dim 100: k_fc1_fwd_v3, two workgroups / CU)
    s.ks_fc1 = g_split[0] > 0 ? g_split[0] : (fastdims ? fit(nb128 * d.A, (d.H == 100 ? 2 : 3) * CUS, 16) : fit(nb64 * d.A, 4 * CUS, 16));
    s.ks_fc1 = min(s.ks_fc1, max(1, cdiv(d.D, 32)));
    // fc11 x_rec/loss/dZ11 kernel: 128-row blocks, 2 workgroups / CU (general fused kernel: 64-row blocks)
    s.ns_fc11 = g_split[1] > 0 ? g_split[1] : (fastdims ? fit(nb128 * d.A, 2 * CUS, 16) : fit(nb64 * d.A, 2 * CUS, 16));
    s.ns_fc11 = min(s.ns_fc11, max(1, cdiv(d.D, 64)));
    // dW1 / dW11: 128-gene tiles, 4 workgroups / CU
    s.ks_dw = g_split[2] > 0 ? g_split[2] : fit(cdiv(d.D, fastdims ? 128 : 64) * d.A, 4 * CUS, 16);
    s.ks_dw = min(s.ks_dw, max(1, cdiv(d.B, 32)));
    // dW11 runs on the side stream beside the latency-bound backward chain, which hides it: fewer, longer
    // workgroups (about 1.6 per CU) cost nothing there and halve the slabs the reduction has to read
    s.ks_dw11 = g_split[5] > 0 ? g_split[5] : fit(cdiv(d.D, 128) * d.A, 13 * CUS / 8, 16);
    s.ks_dw11 = min(s.ks_dw11, max(1, cdiv(d.B, 32)));
    if (!fastdims) s.ks_dw11 = s.ks_dw;
    s.ks_small = g_split[3] > 0 ? g_split[3] : fit(N_SMALL * d.A, 3 * CUS, 32);   // 3 workgroups / CU (136 VGPRs)
    s.ks_small = min(s.ks_small, max(1, cdiv(d.B, 32)));
    s.ks_gd10 = g_split[4] > 0 ? g_split[4] : fit(nb128 * d.A, (fastdims && d.H == 100 ? 2 : 3) * CUS, 16);   // fc_dim 100: k_gd10_v3, 2 / CU
    if (g_split[4] <= 0 && fastdims && d.H == 100) {
        // this is also k_fc11_zg's gene split, and that kernel fits one 256-cell workgroup per CU (152 KB of LDS): take
        // the smallest split whose grid fills whole rounds of the chip (A = 5: 2 -> 5, 200 -> 500 workgroups,
        // 609 -> 529 us; A = 2 keeps 6)
        const int wg = cdiv(d.B, 256) * d.A;
        int best = s.ks_gd10;
        double best_eff = 0.0;
        for (int ks = 1; ks <= 16; ++ks) {
            const int n = wg * ks;
            const double eff = (double)n / (double)(cdiv(n, CUS) * CUS);
            if (eff > best_eff + 0.02) { best_eff = eff; best = ks; }
            if (eff >= 0.93) { best = ks; break; }
        }
        s.ks_gd10 = best;
    }
    s.ks_gd10 = min(s.ks_gd10, max(1, cdiv(d.D, fastdims && d.H == 100 ? 64 : 32)));   // fc_dim 100: also k_fc11_zg's gene split
    // fp32x3 engine (gemm_bf16.hip): its kernels run ONE 512- or 256-thread workgroup per CU -- a pair of 128-row tiles
    // (fc1, dW1, dW11) or 128 cells (the fused fc11 kernel) -- so the splits fill 256 slots, not 512.  dW11 runs beside the
    // latency-bound backward chain, which needs CUs of its own (measured at A = 2: 873 us per step with the splits above,
    // 836 with these).
    if (ex && ex->tune[MMVAE_TUNE_ENGINE] == 2 && fastdims && d.H + 1 <= 112) {
        const int pairs_b = cdiv(nb128, 2), pairs_d = cdiv(cdiv(d.D, 128), 2);
        if (g_split[0] <= 0) s.ks_fc1 = min(fit(pairs_b * d.A, CUS, 16), max(1, cdiv(d.D, 32)));
        if (g_split[4] <= 0) s.ks_gd10 = min(fit(nb128 * d.A, CUS, 16), max(1, cdiv(d.D, 64)));
        if (g_split[2] <= 0) s.ks_dw = min(fit(pairs_d * d.A, CUS, 16), max(1, cdiv(d.B, 32)));
        // (dW11: at most three eighths of the CUs -- its 120 KB of LDS leave a CU no room for a chain workgroup, and the
        // backward chain's 79 A workgroups should still find a CU each in ONE round: A = 2, 3 batch splits 768 us, 2: 757)
        // (workgroup count nearest to 96: measured best at A = 2 (2 splits), A = 3 (2) and A = 5 (1) while the chain kernels
        // ran their own GEMMs on the fp32 matrix instruction.  With those on the split engine too (chain.hip, X3) the chain
        // is 40 us shorter and dW11 has to keep up: nearest to 140 -- A = 2: 4 splits 707 us per step, 3: 720, 2: 728,
        // 5: 722; A = 3: 2 splits 931, 3: 940; A = 5: 1 split 1527, 2: 1530)
        if (g_split[5] <= 0) {
            const int nwg = max(1, pairs_d * d.A);
            const bool chain_x3 = d.C + d.S <= 128 && d.L <= 128 && !ex->tune[MMVAE_TUNE_CHAIN_FP32];
            const int target = chain_x3 ? 140 : 3 * CUS / 8;
            s.ks_dw11 = min(max(1, (target + nwg / 2) / nwg), max(1, cdiv(d.B, 32)));
        }
        if (g_split[3] <= 0) s.ks_small = min(fit(cdiv(N_SMALL * d.A, 2), CUS, 32), max(1, cdiv(d.B, 32)));   // k_x3_small: a pair of products per block
    }
    // the bf16 configuration runs its small-layer gradient products on k_x3_small too
    if (ex && ex->tune[MMVAE_TUNE_ENGINE] == 1 && fastdims && d.H <= 124) {
        if (g_split[3] <= 0) s.ks_small = min(fit(cdiv(N_SMALL * d.A, 2), CUS, 32), max(1, cdiv(d.B, 32)));
        // dW11 beside the backward chain: about 160 of its two-per-CU workgroups (A = 2: 5 splits 701 us per step, 3: 688, 2: 680)
        if (g_split[5] <= 0) {
            const int nwg = max(1, cdiv(d.D, 128) * d.A);
            s.ks_dw11 = min(max(1, (5 * CUS / 8 + nwg / 2) / nwg), max(1, cdiv(d.B, 32)));
        }
    }
    return s;
}

POff make_poff(const mmvae_dims& d) {
    POff p{};
    const int64_t D = d.D, H = d.H, L = d.L, C = d.C, S = d.S;
    const int64_t sizes[MMVAE_N_PARAM_TENSORS] = {
        H * D, H, H * H, H, H * H, H, H * H, H, L * H, L, C * L, C, S * (L + C), S * (L + C), S, S,
        L * (C + S), L, H * L, H, H * H, H, H * H, H, H * H, H, D * H, D};
    int64_t off = 0;
    for (int t = 0; t < MMVAE_N_PARAM_TENSORS; ++t) {
        // fc_sigma.w directly follows fc_mu.w, fc_sigma.b directly follows fc_mu.b: the state head is
        // one [2S, L+C] matrix for the kernels
        if (t != 13 && t != 15) off = cdiv64(off, 4) * 4;
        p.o[t] = off;
        off += sizes[t];
    }
    p.per_arm = cdiv64(off, 64) * 64;
    const int64_t bn[MMVAE_N_BN] = {H, H, H, H, L, S};
    off = 0;
    for (int i = 0; i < MMVAE_N_BN; ++i) {
        p.bn_mean[i] = off; off += bn[i];
        p.bn_var[i] = off; off += bn[i];
    }
    p.bn_per_arm = off;
    return p;
}

Layout make_layout(const mmvae_dims& d, const mmvae_exec* ex) {
    Layout L{};
    const int64_t A = d.A, B = d.B, D = d.D, H = d.H, Ld = d.L, C = d.C, S = d.S;
    L.nblk32 = cdiv(d.B, 32);
    L.nblk64 = cdiv(d.B, 64);
    L.nblkc = cdiv(d.B, CHAIN_ROWS);
    // cells per workgroup of the forward chain launches.  Measured at A = 2, B = 5000 (round 3): 64 cells (158 workgroups) 0.693 ms per
    // step, 48 (210) 0.692, 40 (250: one per CU) 0.695, 32 (314) 0.753 -- a chain launch is a latency chain of fixed costs (statistics
    // read, weight planes, barriers, the exchange), not of per-row work, so smaller blocks on the idle CUs buy nothing.
    L.chain_rows_fwd = CHAIN_ROWS;
    L.nblkf = cdiv(d.B, CHAIN_ROWS);
    L.nblkl = cdiv(d.B, LAT_ROWS);
    L.sp = default_splits(d, ex);
    int64_t off = 0;
    auto take = [&](int64_t n) { const int64_t o = off; off += cdiv64(n, 64) * 64; return o; };
    const int64_t nb = L.nblk32;
    for (int i = 0; i < 5; ++i) {
        const int64_t W = (i == 4) ? Ld : H;
        L.R[i] = take(A * B * W);
        L.bn_mean[i] = take(A * W);
        L.bn_rstd[i] = take(A * W);
        L.bn_part[i] = take(A * nb * 2 * W);
    }
    L.XLOW = take(A * B * Ld); L.CPROB = take(A * B * C); L.CC = take(A * B * C); L.YSOFT = take(A * B * C);
    L.CSMP = take(A * B * C); L.Y = take(A * B * (Ld + C)); L.MS = take(A * B * 2 * S); L.MU = take(A * B * S);
    L.LV = take(A * B * S); L.SS = take(A * B * S); L.ZIN = take(A * B * (C + S));
    for (int i = 0; i < 5; ++i) L.Dk[i] = take(A * B * (i == 0 ? Ld : H));
    L.c_part = take(A * nb * 2 * C); L.c_mean = take(A * C); L.c_iv = take(A * C);
    L.lat_part = take(A * nb * 2);
    L.fc1_slab = take((int64_t)L.sp.ks_fc1 * A * B * NP);
    L.n11 = (L.nblk64 + 2) * (max(L.sp.ns_fc11, L.sp.ks_gd10) + 1) + cdiv(d.D, 64);
    L.fc11_part = take(A * (int64_t)L.n11 * 2 + 64);   // + diagnostic stamp counters
    L.acc = take((int64_t)ACC_NSETS * A * ACC_SET_FLOATS);   // directly behind fc11_part: one zero fill at the start of a forward pass
    L.acc_end = off;                                        // (the backward sets are the last ones: one zero fill at the start of a backward pass)
    L.GD10_slab = take((int64_t)max(L.sp.ns_fc11, L.sp.ks_gd10) * A * B * H);
    L.DZ11 = take(A * B * D);
    L.couple_part = take(nb * 2);
    L.T_part = take(nb * A * C); L.T = take(A * C);
    for (int i = 1; i <= 10; ++i) L.DZ[i] = take(A * B * ((i == 5 || i == 6) ? Ld : H));
    L.GZIN = take(A * B * (C + S)); L.GMS = take(A * B * 2 * S); L.GZC = take(A * B * C);
    for (int i = 1; i <= 5; ++i) {
        const int64_t W = (i == 5) ? Ld : H;
        L.G[i] = take(A * B * W);
        L.bnb_part[i] = take(A * (i == 5 ? (nb > cdiv(B, LAT_ROWS_BWD) ? nb : (int64_t)cdiv(B, LAT_ROWS_BWD)) : nb) * 2 * W);   // layer 5: the latent backward's partials
        L.bnb_sum[i] = take(A * 2 * W);
    }
    L.dw1_slab = take((int64_t)L.sp.ks_dw * A * H * D);
    L.dw11_slab = take((int64_t)max(L.sp.ks_dw, L.sp.ks_dw11) * A * D * DW11_LD);
    L.small_slab = take((int64_t)L.sp.ks_small * A * N_SMALL * NP * SMALL_LD);
    L.xbits = take(A * B * cdiv(d.D, 32));
    {   // slice planes (bf16: two per float)
        const int64_t Dk = cdiv64(D, 32) * 32, Dr = cdiv64(D, 128) * 128, Br = cdiv64(B, 256) * 256;
        L.pl_w1 = take(A * 3 * 128 * Dk / 2);
        L.pl_w11 = take(A * 3 * Dr * 128 / 2);
        L.pl_dz1 = take(A * 3 * Br * 128 / 2);
        L.pl_d10 = take(A * 3 * Br * 128 / 2);
        L.pl_small = take(A * (int64_t)PL_SMALL_SLOTS * 3 * 128 * 128 / 2);
    }
    L.rowmap = take(B + MAP_PAD);
    L.loss_scratch = take(4096);
    L.total = off;
    return L;
}

static int check_dims(const mmvae_dims* d) {
    if (!d) { set_error("dims is null"); return MMVAE_E_BADARG; }
    if (d->A < 1 || d->B < 1 || d->D < 1 || d->H < 1 || d->L < 1 || d->C < 1 || d->S < 1) {
        set_error("non-positive dimension (A=%d B=%d D=%d H=%d L=%d C=%d S=%d)",
                  d->A, d->B, d->D, d->H, d->L, d->C, d->S);
        return MMVAE_E_BADARG;
    }
    if (d->A > MMVAE_MAX_ARMS || d->H > 128 || d->C > 128 || d->L > 64 || 2 * d->S > 64 || d->L + d->C > 255 ||
        d->C + d->S > 255) {
        set_error("unsupported shape: need A<=%d, fc_dim<=128, n_categories<=128, lowD_dim<=64, state_dim<=32, "
                  "lowD+C<=255, C+S<=255 (got A=%d H=%d C=%d L=%d S=%d)",
                  MMVAE_MAX_ARMS, d->A, d->H, d->C, d->L, d->S);
        return MMVAE_E_UNSUPPORTED;
    }
    return 0;
}

static int make_ctx(Ctx& c, const mmvae_dims* d, const mmvae_hyper* h, void* ws, size_t ws_bytes, mmvae_exec* ex,
                    void* stream) {
    if (int rc = check_dims(d)) return rc;
    if (!h || !ws) { set_error("null hyper / workspace"); return MMVAE_E_BADARG; }
    if (h->training && d->B < 2) {   // a one-cell batch has no batch statistics; eval mode (running statistics) takes it
        set_error("training mode needs B >= 2 for the batch statistics (got B=%d)", d->B);
        return MMVAE_E_BADARG;
    }
    // the fixed-point batch-sum accumulators (common.hpp acc_add) hold 2^12 addends of the largest magnitude per column
    // without a carry between their slots; the producer with the fewest cells per workgroup is the latent backward kernel
    if (h->training && cdiv(d->B, ACC_MIN_PRODUCER_ROWS) > ACC_MAX_ADDENDS) {
        set_error("training mode takes at most %d cells per batch and rank (got B=%d): capacity of the exact batch-sum accumulators",
                  ACC_MAX_ADDENDS * ACC_MIN_PRODUCER_ROWS, d->B);
        return MMVAE_E_UNSUPPORTED;
    }
    c.d = *d;
    c.h = *h;
    if (h->cat_mask[0] | h->cat_mask[1] | h->cat_mask[2] | h->cat_mask[3]) {
        // bits beyond n_categories are ignored; at least one category must be kept
        uint32_t any = 0;
        for (int k = 0; k < d->C; ++k) any |= (h->cat_mask[k >> 5] >> (k & 31)) & 1u;
        if (!any) { set_error("cat_mask keeps none of the %d categories", d->C); return MMVAE_E_BADARG; }
    }
    c.ex_out = ex;
    if (ex) c.ex = *ex; else memset(&c.ex, 0, sizeof(c.ex));
    if (c.ex.side_stream) {
        for (int i = 0; i < MMVAE_N_EVENTS; ++i)
            if (!c.ex.ev[i]) { set_error("mmvae_exec: side_stream is set but ev[%d] is null", i); return MMVAE_E_BADARG; }
        if (c.ex.side_stream == stream) { set_error("mmvae_exec: side_stream must differ from the call's stream"); return MMVAE_E_BADARG; }
    }
    c.lay = make_layout(*d, &c.ex);
    c.po = make_poff(*d);
    if ((size_t)c.lay.total * sizeof(float) > ws_bytes) {
        set_error("workspace too small: need %zu bytes, got %zu", (size_t)c.lay.total * sizeof(float), ws_bytes);
        return MMVAE_E_WORKSPACE;
    }
    if (reinterpret_cast<uintptr_t>(ws) & 255) { set_error("workspace must be 256-byte aligned"); return MMVAE_E_BADARG; }
    c.ws = reinterpret_cast<float*>(ws);
    c.stream = reinterpret_cast<hipStream_t>(stream);
    return 0;
}

static int check_noise(const Ctx& c, const mmvae_noise* nz) {
    if (!nz) { set_error("noise descriptor is null"); return MMVAE_E_BADARG; }
    if (nz->mode == 0) {
        if (c.h.training && c.h.x_drop > 0.f && !nz->x_mask) { set_error("explicit noise: x_mask is null"); return MMVAE_E_BADARG; }
        if (!c.h.eval_flag && !nz->u_gumbel) { set_error("explicit noise: u_gumbel is null"); return MMVAE_E_BADARG; }
        if (!nz->u_state) { set_error("explicit noise: u_state is null"); return MMVAE_E_BADARG; }
        if (c.h.training && c.h.s_drop > 0.f && !nz->s_mask) { set_error("explicit noise: s_mask is null"); return MMVAE_E_BADARG; }
    } else if (nz->mode != 1) {
        set_error("noise mode must be 0 (explicit) or 1 (philox)");
        return MMVAE_E_BADARG;
    }
    if (c.h.x_drop < 0.f || c.h.x_drop >= 1.f || c.h.s_drop < 0.f || c.h.s_drop >= 1.f) {
        set_error("dropout probabilities must be in [0,1)");
        return MMVAE_E_BADARG;
    }
    return 0;
}

// Train step with a side stream (couple_done != null): the coupling kernel needs only the latent block's outputs
// and the loss scalars only the coupling and fc11 partials, so both run on the side stream -- the coupling beside
// the decoder chain and fc11, the finalisation (loss_out != null) beside the d(d10) GEMM.  *couple_done tells the
// caller that the loss is on its way (event EV_COUPLE) and do_loss must not launch anything.
static int fork_to_side(const Ctx& c, int ev) {
    if (hipEventRecord(c.ev(ev), c.stream) != hipSuccess || hipStreamWaitEvent(c.side(), c.ev(ev), 0) != hipSuccess) {
        set_error("stream fork failed");
        return MMVAE_E_LAUNCH;
    }
    return 0;
}
// the fork's event rode on the kernel in front of it (Ctx::stop_ev, launch_k): only the side stream's wait is left
static int fork_wait_only(const Ctx& c, int ev) {
    if (hipStreamWaitEvent(c.side(), c.ev(ev), 0) != hipSuccess) { set_error("stream fork failed"); return MMVAE_E_LAUNCH; }
    return 0;
}
static int record_on_side(const Ctx& c, int ev) {
    if (hipEventRecord(c.ev(ev), c.side()) != hipSuccess) { set_error("event record failed"); return MMVAE_E_LAUNCH; }
    return 0;
}
static int join_from_side(const Ctx& c, int ev) {
    if (hipStreamWaitEvent(c.stream, c.ev(ev), 0) != hipSuccess) { set_error("stream join failed"); return MMVAE_E_LAUNCH; }
    return 0;
}

static int do_forward(const Ctx& c, const mmvae_noise* nz, const float* params, float* bn_running, int64_t* nbt,
                      const float* x, int64_t xs, float* x_rec, int need_grad, bool* couple_done = nullptr,
                      float* loss_out = nullptr, bool latent_only = false, int32_t* labels = nullptr) {
    int rc;
    const bool fast = fast_path_ok(c, params, x, xs);
    // training: the loss partial slots and the forward accumulator sets start the pass at zero (inside k_make_xbits
    // when that runs); eval mode has no batch sums and the fc11 launchers zero their slots themselves
    const bool merged = fast && prologue_merged(c);   // keep-mask + zero fill inside the k_presplit launch below
    if (c.h.training) {
        if (!merged && (rc = launch_forward_zero(c, fast, nz))) return rc;
    } else if (fast && (rc = launch_make_xbits(c, nz))) {
        return rc;
    }
    if (fast) {
        if ((rc = launch_x3_planes(c, params, merged ? 17 : 1, nz))) return rc;   // fp32x3: slice planes of W1, [W11 | b11], the small layers
        if ((rc = launch_fc1_fwd_fast(c, params, x, xs))) return rc;
        if ((rc = launch_fc1_epi(c, params))) return rc;
    } else if ((rc = launch_fc1_fwd(c, nz, params, x, xs))) {
        return rc;
    }
    // batch statistics are recombined by the kernel that consumes each BatchNorm (no finalize launches);
    // eval mode copies the running statistics into the workspace instead
    if ((rc = launch_bn_eval_stats(c, bn_running))) return rc;
    if (!c.h.training && !c.tune(MMVAE_TUNE_EVAL_CHAIN_OFF)) {
        if ((rc = launch_chain_fwd_enc_eval(c, params))) return rc;
    } else {
        for (int layer = 2; layer <= 5; ++layer)
            if ((rc = launch_chain_fwd_enc(c, layer, params, bn_running, nbt))) return rc;
    }
    // fork events ride on the kernels in front of the forks (the latent forward here, the fused fc11 kernel below): a recorded
    // event is a barrier packet of its own, 6 - 7 us of idle main stream (round 3: 686 -> 681 us per step)
    const bool t_early = fast && couple_done && c.side() && loss_out && !latent_only && fc11_split_path(c, params, x, xs);
    c.stop_used = false;
    // the coupling terms as a role of the decoder chain's launch: no fork behind the latent forward at all
    const bool couple_role = t_early && dec_couple_ok(c);
    c.couple_in_dec = false;
    if (couple_done && c.side() && !latent_only && !couple_role) c.stop_ev = c.ev(EV_LAT);
    if ((rc = launch_lat_fwd(c, nz, params, bn_running, nbt, labels))) return rc;
    c.stop_ev = nullptr;
    const bool lat_rode = c.stop_used;
    c.stop_used = false;
    if (latent_only) return 0;   // evaluation labels need c only: no decoder, no fc11
    Ctx cs = c;
    cs.stream = c.side();
    // The fused step on the fast path (t_early): the coupling terms AND the T sums of the latent backward (which need nothing but
    // the coupling kernel's output) run on the side stream from here, beside the decoder chain and fc11; the loss scalars
    // (which need fc11's partials) follow dW11 on the side stream in do_backward -- no fork between fc11 and the backward
    // pass, and dW11 starts as soon as fc11 has finished: EV_FORK rides on the fc11 kernel.
    auto fc11_with_fork = [&]() -> int {
        if (need_grad) c.stop_ev = c.ev(EV_FORK);
        const int r = launch_fc11_fast(c, params, x, xs, x_rec, need_grad);
        c.stop_ev = nullptr;
        c.fork_on_fc11 = c.stop_used;
        c.stop_used = false;
        return r;
    };
    if (couple_role) {
        *couple_done = true;
        c.couple_in_dec = true;
        if ((rc = launch_chain_fwd_dec(c, params, true))) return rc;
        if (need_grad && (rc = launch_x3_planes(c, params, 2))) return rc;
        return fc11_with_fork();
    }
    if (couple_done && c.side()) {
        if ((rc = lat_rode ? fork_wait_only(c, EV_LAT) : fork_to_side(c, EV_LAT))) return rc;
        if ((rc = launch_couple(cs))) return rc;
        *couple_done = true;
        if (t_early) {
            if ((rc = launch_loss_finalize(cs, loss_out, 1))) return rc;
            if ((rc = record_on_side(c, EV_COUPLE))) return rc;
        }
    }
    if ((rc = launch_chain_fwd_dec(c, params))) return rc;
    if (fast && need_grad && (rc = launch_x3_planes(c, params, 2))) return rc;   // fp32x3: slice planes of [d10 | 1] (fc11, dW11)
    if (t_early) return fc11_with_fork();
    if (couple_done && *couple_done && (rc = record_on_side(c, EV_COUPLE))) return rc;
    if (fast) return launch_fc11_fast(c, params, x, xs, x_rec, need_grad);
    return launch_fc11_fused(c, params, x, xs, x_rec, need_grad);
}

static int do_loss(const Ctx& c, float* loss_out) {
    int rc;
    if ((rc = launch_couple(c))) return rc;
    return launch_loss_finalize(c, loss_out);
}

// scalars_out != null: the loss scalars are still to be computed (mmvae_train_step on the fast path; the T sums are
// already on the side stream, EV_COUPLE) -- behind dW11 on the side stream, or at the end of the main stream
static int do_backward(const Ctx& c, const mmvae_noise* nz, const float* params, const float* x, int64_t xs,
                       float grad_scale, float* grads, const AdamHost* adam = nullptr, bool wait_loss = false,
                       float* scalars_out = nullptr) {
    int rc;
    const bool fast = fast_path_ok(c, params, x, xs);
    // dW11 depends only on dZ11 and d10 (both final after forward): it runs on the side stream beside the backward chain, forked at
    // the START of the backward pass -- the later it starts, the more of it lands on the MFMA-bound dW1 (measured again in round 4,
    // profiles/r04_dw11_placement_sweep.txt: behind the decoder chain + 30 us per step, behind the latent backward + 15, not
    // forked + 75; fewer or more workgroups than the default split + 5 .. 10).
    bool forked = false;
    const bool use_side = fast && c.side();
    const bool early = use_side && !adam && c.ex.early_grad_event != nullptr;
    const bool side_red = use_side && adam;
    if (c.ex_out) c.ex_out->early_recorded = 0;
    Ctx cs = c;
    cs.stream = c.side();
    if (use_side) {
        if ((rc = (c.fork_on_fc11 ? fork_wait_only(c, EV_FORK) : fork_to_side(c, EV_FORK)))) return rc;
        c.fork_on_fc11 = false;
        if ((rc = launch_dw_big_fast(cs, x, xs, 2))) return rc;
        if (early) {
            // data parallel: fc11.weight / fc11.bias (47 % of the parameters) are final here; reduce their slabs now
            // and tell the caller, who starts their all-reduce beside the rest of backward
            if ((rc = launch_reduce_grads(cs, grads, grad_scale, nullptr, true, 1))) return rc;
            if (hipEventRecord(reinterpret_cast<hipEvent_t>(c.ex.early_grad_event), c.side()) != hipSuccess) {
                set_error("event record failed");
                return MMVAE_E_LAUNCH;
            }
            if (c.ex_out) c.ex_out->early_recorded = 1;
        }
        if (side_red) {
            // fused Adam: fc11.weight / fc11.bias (47 % of the parameters) are reduced and updated here, behind their GEMM on
            // the side stream -- nothing reads W11 again in this step, and the side stream is idle from here to the join
            if ((rc = launch_reduce_grads(cs, grads, grad_scale, adam, true, 1))) return rc;
        }
        if (scalars_out && (rc = launch_loss_finalize(cs, scalars_out, 2))) return rc;
        if ((rc = record_on_side(c, EV_JOIN))) return rc;
        forked = true;
    }
    // fp32x3 engine: a backward pass that is its own call writes the small layers' weight planes again (the fused step's
    // forward pass has left them in place)
    if (fast && !c.small_planes && (rc = launch_x3_planes(c, params, 8))) return rc;
    const int nslab = fc11_split_path(c, params, x, xs) ? c.lay.sp.ks_gd10 : c.lay.sp.ns_fc11;
    if ((rc = launch_chain_bwd_dec(c, params, nslab))) return rc;
    // T (sum of G log c, from the loss finalisation) is first needed here
    if (wait_loss && !c.couple_in_dec && (rc = join_from_side(c, EV_COUPLE))) return rc;
    if ((rc = launch_lat_bwd(c, nz, params))) return rc;
    for (int layer = 5; layer >= 2; --layer)
        if ((rc = launch_chain_bwd_enc(c, layer, params))) return rc;
    if ((rc = launch_bn_bwd_apply1(c))) return rc;
    if (fast) {
        if ((rc = launch_x3_planes(c, params, 4))) return rc;            // fp32x3: slice planes of dZ1 (dW1)
        if ((rc = launch_dw_big_fast(c, x, xs, forked ? 1 : 3))) return rc;
    } else if ((rc = launch_dw_big(c, nz, x, xs))) {
        return rc;
    }
    if ((rc = launch_dw_small(c))) return rc;
    const bool fc11_on_side = (early || side_red) && forked;
    if (forked && (rc = join_from_side(c, EV_JOIN))) return rc;
    if (scalars_out && !forked && (rc = launch_loss_finalize(c, scalars_out, 2))) return rc;
    return launch_reduce_grads(c, grads, grad_scale, adam, fast, fc11_on_side ? 2 : 3);
}

}  // namespace mmvae

using namespace mmvae;

extern "C" {

int mmvae_abi_version(void) { return 4; }
const char* mmvae_last_error_string(void) { return g_err; }
int mmvae_check_dims(const mmvae_dims* d) { return check_dims(d); }

int mmvae_param_layout(const mmvae_dims* d, mmvae_param_layout_t* out) {
    if (int rc = check_dims(d)) return rc;
    if (!out) { set_error("out is null"); return MMVAE_E_BADARG; }
    const POff p = make_poff(*d);
    const int64_t D = d->D, H = d->H, L = d->L, C = d->C, S = d->S;
    const int64_t rows[MMVAE_N_PARAM_TENSORS] = {H, H, H, H, H, H, H, H, L, L, C, C, S, S, S, S, L, L, H, H, H, H, H, H, H, H, D, D};
    const int64_t cols[MMVAE_N_PARAM_TENSORS] = {D, 1, H, 1, H, 1, H, 1, H, 1, L, 1, L + C, L + C, 1, 1, C + S, 1, L, 1, H, 1, H, 1, H, 1, H, 1};
    out->per_arm = p.per_arm;
    for (int t = 0; t < MMVAE_N_PARAM_TENSORS; ++t) { out->offset[t] = p.o[t]; out->rows[t] = rows[t]; out->cols[t] = cols[t]; }
    out->bn_per_arm = p.bn_per_arm;
    const int64_t bn[MMVAE_N_BN] = {H, H, H, H, L, S};
    for (int i = 0; i < MMVAE_N_BN; ++i) { out->bn_mean_offset[i] = p.bn_mean[i]; out->bn_var_offset[i] = p.bn_var[i]; out->bn_dim[i] = bn[i]; }
    return 0;
}

size_t mmvae_workspace_bytes(const mmvae_dims* d, const mmvae_exec* ex) {
    if (check_dims(d)) return 0;
    return (size_t)make_layout(*d, ex).total * sizeof(float);
}

int64_t mmvae_ws_offset(const mmvae_dims* d, const mmvae_exec* ex, int id) {
    if (check_dims(d)) return -1;
    const Layout L = make_layout(*d, ex);
    switch (id) {
        case MMVAE_WS_X_LOW: return L.XLOW;
        case MMVAE_WS_C_PROB: return L.CPROB;
        case MMVAE_WS_C: return L.CC;
        case MMVAE_WS_C_SMP: return L.CSMP;
        case MMVAE_WS_S_MEAN: return L.MU;
        case MMVAE_WS_S_LOGVAR: return L.LV;
        case MMVAE_WS_S_SMP: return L.SS;
        case MMVAE_WS_Y_SOFT: return L.YSOFT;
        case MMVAE_WS_R1: return L.R[0];
        case MMVAE_WS_R2: return L.R[1];
        case MMVAE_WS_R3: return L.R[2];
        case MMVAE_WS_R4: return L.R[3];
        case MMVAE_WS_R5: return L.R[4];
        case MMVAE_WS_D6: return L.Dk[0];
        case MMVAE_WS_D7: return L.Dk[1];
        case MMVAE_WS_D8: return L.Dk[2];
        case MMVAE_WS_D9: return L.Dk[3];
        case MMVAE_WS_D10: return L.Dk[4];
        case MMVAE_WS_ZIN: return L.ZIN;
        case MMVAE_WS_DZ11: return L.DZ11;
        case MMVAE_WS_DZ1: return L.DZ[1];
        case MMVAE_WS_GZIN: return L.GZIN;
        case MMVAE_WS_GZC: return L.GZC;
        case MMVAE_WS_G5: return L.G[5];
        case MMVAE_WS_BN_MEAN1: return L.bn_mean[0];
        case MMVAE_WS_GD10_SLAB: return L.GD10_slab;
        case MMVAE_WS_G1: case MMVAE_WS_G2: case MMVAE_WS_G3: case MMVAE_WS_G4: return L.G[1 + id - MMVAE_WS_G1];
        case MMVAE_WS_DZ2: case MMVAE_WS_DZ3: case MMVAE_WS_DZ4: case MMVAE_WS_DZ5: return L.DZ[2 + id - MMVAE_WS_DZ2];
        default: set_error("unknown workspace id %d", id); return -1;
    }
}

int mmvae_splits(const mmvae_dims* d, const mmvae_exec* ex, int32_t out[6]) {
    if (int rc = check_dims(d)) return rc;
    if (!out) { set_error("out is null"); return MMVAE_E_BADARG; }
    const Splits s = default_splits(*d, ex);
    out[0] = s.ks_fc1; out[1] = s.ns_fc11; out[2] = s.ks_dw; out[3] = s.ks_small; out[4] = s.ks_gd10; out[5] = s.ks_dw11;
    return 0;
}

int64_t mmvae_ws_debug_offset(const mmvae_dims* d, const mmvae_exec* ex) {
    if (check_dims(d)) return -1;
    return make_layout(*d, ex).loss_scratch + 2048;
}

int mmvae_forward(const mmvae_dims* d, const mmvae_hyper* h, const mmvae_noise* nz, const float* params,
                  float* bn_running, int64_t* nbt, const float* x, int64_t x_arm_stride, float* x_rec, int need_grad,
                  void* ws, size_t ws_bytes, mmvae_exec* ex, void* stream) {
    Ctx c;
    if (int rc = make_ctx(c, d, h, ws, ws_bytes, ex, stream)) return rc;
    if (!params || !x) { set_error("null params / x"); return MMVAE_E_BADARG; }
    if (int rc = check_noise(c, nz)) return rc;
    if (need_grad && !h->training) { set_error("need_grad requires training mode (batch statistics)"); return MMVAE_E_UNSUPPORTED; }
    return do_forward(c, nz, params, bn_running, nbt, x, x_arm_stride, x_rec, need_grad);
}

int mmvae_loss(const mmvae_dims* d, const mmvae_hyper* h, void* ws, size_t ws_bytes, float* loss_out, mmvae_exec* ex,
               void* stream) {
    Ctx c;
    if (int rc = make_ctx(c, d, h, ws, ws_bytes, ex, stream)) return rc;
    if (!loss_out) { set_error("loss_out is null"); return MMVAE_E_BADARG; }
    return do_loss(c, loss_out);
}

int mmvae_backward(const mmvae_dims* d, const mmvae_hyper* h, const mmvae_noise* nz, const float* params,
                   const float* x, int64_t x_arm_stride, float grad_scale, void* ws, size_t ws_bytes, float* grads,
                   mmvae_exec* ex, void* stream) {
    Ctx c;
    if (int rc = make_ctx(c, d, h, ws, ws_bytes, ex, stream)) return rc;
    if (!params || !x || !grads) { set_error("null params / x / grads"); return MMVAE_E_BADARG; }
    if (int rc = check_noise(c, nz)) return rc;
    if (!h->training) { set_error("backward requires training mode"); return MMVAE_E_UNSUPPORTED; }
    return do_backward(c, nz, params, x, x_arm_stride, grad_scale, grads);
}

int mmvae_adam_step(int64_t n, float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t step,
                    float lr, float beta1, float beta2, float adam_eps, float weight_decay, int decoupled,
                    void* stream) {
    if (n <= 0 || !params || !grads || !exp_avg || !exp_avg_sq || step < 1) {
        set_error("adam: bad argument");
        return MMVAE_E_BADARG;
    }
    return launch_adam(n, params, grads, exp_avg, exp_avg_sq, step, lr, beta1, beta2, adam_eps, weight_decay, decoupled,
                       reinterpret_cast<hipStream_t>(stream));
}

static int train_step_impl(Ctx& c, const mmvae_hyper* h, const mmvae_noise* nz, float* params,
                           float* bn_running, int64_t* nbt, const float* x, int64_t x_arm_stride,
                           float* grads, float* loss_out, int do_adam, float* exp_avg, float* exp_avg_sq, int64_t step,
                           float lr, float beta1, float beta2, float adam_eps, float weight_decay, int decoupled) {
    if (!params || !x || !grads || !loss_out) { set_error("null params / x / grads / loss_out"); return MMVAE_E_BADARG; }
    if (int rc = check_noise(c, nz)) return rc;
    if (!h->training) { set_error("train_step requires training mode"); return MMVAE_E_UNSUPPORTED; }
    int rc;
    bool side_loss = false;   // coupling (+ loss scalars on the fast path) already running on the side stream
    if ((rc = do_forward(c, nz, params, bn_running, nbt, x, x_arm_stride, nullptr, 1, &side_loss, loss_out))) return rc;
    const bool loss_on_side = side_loss && fc11_split_path(c, params, x, x_arm_stride);
    if (side_loss && !loss_on_side) {   // general path: coupling done on the side, finalise here
        if ((rc = join_from_side(c, EV_COUPLE))) return rc;
        if ((rc = launch_loss_finalize(c, loss_out))) return rc;
    } else if (!side_loss && (rc = do_loss(c, loss_out))) {
        return rc;
    }
    if (do_adam) {
        // the Adam update rides on the slab reduction (alignment gaps of the flat buffers hold zeros and
        // need no update)
        if (!exp_avg || !exp_avg_sq || step < 1) { set_error("adam state missing"); return MMVAE_E_BADARG; }
        const AdamHost ah{params, exp_avg, exp_avg_sq, step, lr, beta1, beta2, adam_eps, weight_decay, decoupled};
        return do_backward(c, nz, params, x, x_arm_stride, 1.f, grads, &ah, loss_on_side, loss_on_side ? loss_out : nullptr);
    }
    return do_backward(c, nz, params, x, x_arm_stride, 1.f, grads, nullptr, loss_on_side, loss_on_side ? loss_out : nullptr);
}

int mmvae_train_step(const mmvae_dims* d, const mmvae_hyper* h, const mmvae_noise* nz, float* params,
                     float* bn_running, int64_t* nbt, const float* x, int64_t x_arm_stride, void* ws, size_t ws_bytes,
                     float* grads, float* loss_out, int do_adam, float* exp_avg, float* exp_avg_sq, int64_t step,
                     float lr, float beta1, float beta2, float adam_eps, float weight_decay, int decoupled,
                     mmvae_exec* ex, void* stream) {
    Ctx c;
    if (int rc = make_ctx(c, d, h, ws, ws_bytes, ex, stream)) return rc;
    return train_step_impl(c, h, nz, params, bn_running, nbt, x, x_arm_stride, grads, loss_out, do_adam, exp_avg, exp_avg_sq, step,
                           lr, beta1, beta2, adam_eps, weight_decay, decoupled);
}

int mmvae_train_step_rows(const mmvae_dims* d, const mmvae_hyper* h, const mmvae_noise* nz, float* params,
                          float* bn_running, int64_t* nbt, const float* data, const uint16_t* data_bf16, int64_t ld, int64_t n_rows,
                          const int64_t* rows, void* ws, size_t ws_bytes, float* grads, float* loss_out, int do_adam, float* exp_avg,
                          float* exp_avg_sq, int64_t step, float lr, float beta1, float beta2, float adam_eps,
                          float weight_decay, int decoupled, mmvae_exec* ex, void* stream) {
    Ctx c;
    if (int rc = make_ctx(c, d, h, ws, ws_bytes, ex, stream)) return rc;
    if (!data || !rows || n_rows < 1 || ld < d->D) { set_error("train_step_rows: null data / rows, n_rows < 1 or ld < D"); return MMVAE_E_BADARG; }
    // what the row-indexed kernels take: the fp32x3 engine's fused step (its head launch builds the row map), 16-byte rows,
    // a matrix within reach of 32-bit byte offsets.  Everything else: gather the batch (mmvae_gather_rows) and call
    // mmvae_train_step.
    if ((ld & 3) || (reinterpret_cast<uintptr_t>(data) & 15) || n_rows * ld >= ((int64_t)1 << 30)) {
        set_error("train_step_rows: needs ld %% 4 == 0, 16-byte aligned data and n_rows * ld < 2^30 floats");
        return MMVAE_E_UNSUPPORTED;
    }
    const bool x3 = split3_gemms(c) && d->H + 1 <= 112, b16 = (h->gemm_bf16 & 0xFF) == 1 && bf16_gemms(c);
    if (!h->training || !(h->x_drop > 0.f) || !(x3 || b16) || !prologue_merged(c) || !fast_path_ok(c, params, data, 0) ||
        (int64_t)cdiv(d->B, 128) * c.lay.sp.ks_gd10 > c.lay.n11 || c.tune(MMVAE_TUNE_FC11_ZG_OFF) || ((h->gemm_bf16 >> 8) & 15)) {
        set_error("train_step_rows: only the fused training step of the fp32x3 / bf16 engines reads the batch through a row map");
        return MMVAE_E_UNSUPPORTED;
    }
    if (data_bf16) {   // bf16 storage: the bf16 engine reads x from the copy and keeps dZ11 as bf16
        if (!b16 || (d->D & 7) || (ld & 7) || (reinterpret_cast<uintptr_t>(data_bf16) & 15)) {
            set_error("train_step_rows: data_bf16 needs the bf16 engine, D %% 8 == 0, ld %% 8 == 0 and a 16-byte aligned copy");
            return MMVAE_E_UNSUPPORTED;
        }
        c.x16 = data_bf16;
    }
    c.x_rows = rows;
    c.x_ld = ld;
    c.x_nrows = n_rows;
    return train_step_impl(c, h, nz, params, bn_running, nbt, data, 0, grads, loss_out, do_adam, exp_avg, exp_avg_sq, step,
                           lr, beta1, beta2, adam_eps, weight_decay, decoupled);
}

int mmvae_eval_classify(const mmvae_dims* d, const mmvae_hyper* h, const float* params, const float* bn_running,
                        const float* x, int64_t x_arm_stride, void* ws, size_t ws_bytes, int32_t* labels,
                        int64_t* counts, mmvae_exec* ex, void* stream) {
    Ctx c;
    if (int rc = make_ctx(c, d, h, ws, ws_bytes, ex, stream)) return rc;
    if (!params || !x || !bn_running || !labels) { set_error("null params / x / bn_running / labels"); return MMVAE_E_BADARG; }
    if (h->training || !h->eval_flag) { set_error("eval_classify needs training = 0 and eval_flag = 1"); return MMVAE_E_UNSUPPORTED; }
    int rc;
    // eval mode reads the running statistics (nothing is written to bn_running) and draws no Gumbel noise; the state
    // sample the latent kernel also produces does not enter c: it takes the Philox stream of seed 0
    mmvae_noise nzp{};
    nzp.mode = 1;
    // the latent kernel's hard-sample argmax IS classify(c) in eval mode: the labels come out of it directly
    if ((rc = do_forward(c, &nzp, params, const_cast<float*>(bn_running), nullptr, x, x_arm_stride, nullptr, 0, nullptr,
                         nullptr, true, labels)))
        return rc;
    if (counts) return launch_confmat(labels, d->A, d->B, d->C, counts, c.stream);
    return 0;
}

int mmvae_classify(const float* c_probs, int64_t n_cells, int C, int32_t* labels, void* stream) {
    if (!c_probs || !labels || n_cells <= 0 || C <= 0) { set_error("classify: bad argument"); return MMVAE_E_BADARG; }
    return launch_classify(c_probs, n_cells, C, labels, reinterpret_cast<hipStream_t>(stream));
}

int mmvae_confmat_accumulate(const int32_t* labels, int A, int64_t n, int C, int64_t* counts, void* stream) {
    if (!labels || !counts || A < 1 || A > MMVAE_MAX_ARMS || n <= 0 || C <= 0) {
        set_error("confmat_accumulate: bad argument");
        return MMVAE_E_BADARG;
    }
    return launch_confmat(labels, A, n, C, counts, reinterpret_cast<hipStream_t>(stream));
}

int mmvae_consensus(const int64_t* counts, int npairs, int C, double* cm_norm, double* consensus, void* stream) {
    if (!counts || !consensus || npairs < 1 || C < 1) { set_error("consensus: bad argument"); return MMVAE_E_BADARG; }
    if (C > 128) { set_error("consensus: C > 128 unsupported"); return MMVAE_E_UNSUPPORTED; }
    return launch_consensus(counts, npairs, C, cm_norm, consensus, reinterpret_cast<hipStream_t>(stream));
}

int mmvae_debug_stage(const mmvae_dims* d, const mmvae_hyper* h, const mmvae_noise* nz, int stage,
                      const float* params, const float* x, int64_t x_arm_stride, void* ws, size_t ws_bytes,
                      float* grads, mmvae_exec* ex, void* stream) {
    Ctx c;
    if (int rc = make_ctx(c, d, h, ws, ws_bytes, ex, stream)) return rc;
    if (!params || !x) { set_error("null params / x"); return MMVAE_E_BADARG; }
    if (int rc = check_noise(c, nz)) return rc;
    // a stage is replayed on the state a complete forward / backward pass of the same engine left behind: its slice planes
    // (fp32x3 engine) are in place
    c.small_planes = chain_x3_ok(c) && fast_path_ok(c, params, x, x_arm_stride);
    switch (stage) {
        case 0:
            if (fast_path_ok(c, params, x, x_arm_stride)) {
                if (int rc = launch_fc1_fwd_fast(c, params, x, x_arm_stride)) return rc;
                return launch_fc1_epi(c, params);
            }
            return launch_fc1_fwd(c, nz, params, x, x_arm_stride);
        case 1:
            if (fast_path_ok(c, params, x, x_arm_stride)) return launch_fc11_fast(c, params, x, x_arm_stride, nullptr, 1);
            return launch_fc11_fused(c, params, x, x_arm_stride, nullptr, 1);
        case 2:
            if (fast_path_ok(c, params, x, x_arm_stride)) return launch_dw_big_fast(c, x, x_arm_stride, 3);
            return launch_dw_big(c, nz, x, x_arm_stride);
        case 9: return launch_make_xbits(c, nz);
        case 20: return launch_chain_fwd_enc(c, 3, params, nullptr, nullptr);   // one encoder layer (fc3)
        case 21: return launch_chain_bwd_enc(c, 3, params);
        // single kernels of the fast path (per-kernel roofline timing)
        case 10: case 11: case 12: case 13: case 14:
            if (!fast_path_ok(c, params, x, x_arm_stride)) { set_error("stage %d needs the fast path", stage); return MMVAE_E_UNSUPPORTED; }
            if (stage == 10) return launch_fc11_fast(c, params, x, x_arm_stride, nullptr, 1, 1);
            if (stage == 11) return launch_fc11_fast(c, params, x, x_arm_stride, nullptr, 1, 2);
            if (stage == 12) return launch_dw_big_fast(c, x, x_arm_stride, 1);
            if (stage == 13) return launch_dw_big_fast(c, x, x_arm_stride, 2);
            return launch_fc1_fwd_fast(c, params, x, x_arm_stride);
        case 3: return launch_dw_small(c);
        case 4: return launch_chain_fwd_dec(c, params);
        case 5: return launch_chain_bwd_dec(c, params, fc11_split_path(c, params, x, x_arm_stride) ? c.lay.sp.ks_gd10 : c.lay.sp.ns_fc11);
        case 6: return launch_lat_fwd(c, nz, params, nullptr, nullptr);
        case 7: return launch_lat_bwd(c, nz, params);
        case 8: if (!grads) { set_error("grads is null"); return MMVAE_E_BADARG; } return launch_reduce_grads(c, grads, 1.f, nullptr, fast_path_ok(c, params, x, x_arm_stride));
        default: set_error("unknown stage %d", stage); return MMVAE_E_BADARG;
    }
}

int mmvae_dump_noise(const mmvae_dims* d, const mmvae_hyper* h, const mmvae_noise* nz, uint8_t* x_mask,
                     float* u_gumbel, float* u_state, uint8_t* s_mask, void* stream) {
    if (int rc = check_dims(d)) return rc;
    if (!h || !nz) { set_error("null hyper / noise"); return MMVAE_E_BADARG; }
    return launch_dump_noise(*d, *h, nz, x_mask, u_gumbel, u_state, s_mask, reinterpret_cast<hipStream_t>(stream));
}

}  // extern "C"
