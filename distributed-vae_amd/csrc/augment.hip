// Augmenter forward in the training loop.
//
// The reference trainer passes every batch through a frozen, eval-mode generator before the VAE step
// (`xs = self.netA(x.expand(A,-1,-1), True, 0.1)[1]`, mmidas/cpl_mixvae.py:422-423, `netA.eval()` :184;
// `Augmenter_smartseq.forward`, mmidas/augmentation/udagan.py:281-329): an MLP
//   D -> D/5 -> D/5 -> n -> n -> [concat noise] n/5 -> (mu, sigma) -> z -> n/5 -> n -> n -> D/5 -> D/5 -> D
// with BatchNorm1d(affine=False, eps=1e-10) + ReLU after every layer but the last -- 13.6 M MACs per cell-arm, about
// 2.5x the VAE's own forward + backward.  In eval mode BatchNorm is a per-column scale and shift, so each layer is
// one GEMM with a fused epilogue:  out = relu(acc * scale + shift),  scale = rsqrt(var + eps), shift = (b - mean) scale.
//
// What this file does differently from the reference's call pattern:
//   * x is the same for every arm (x.expand): the layers in front of the noise injection (fc1..fc4 and the x part of
//     fc5) are computed once per cell, not once per cell-arm (a third of the FLOPs at A = 2);
//   * the concat (h4 | z) in front of fc5 is never materialised: fc5 = h4 W5[:, :n]^T + z W5[:, n:]^T, the first term
//     a GEMM over the cells, the second inside the row-wise latent kernel together with the noise Linear + BN + ELU,
//     the (mu, sigma) heads, the reparameterisation and fc6;
//   * weights are packed once (mmvae_aug_pack): rows padded to a multiple of 4 floats for 16-byte loads (D/5 = 1006
//     for the 5032-gene panel), BatchNorm folded into (scale, shift).
//
// GEMM kernel: C[M,N] = epi(A[M,K] W[N,K]^T), fp32 MFMA 32x32x2, 128 x 128 block tile, K tile 32, 4 waves of 64 x 64
// (four accumulators per wave, one ds_read_b128 per operand per four MFMAs), register prefetch of the next K tile,
// workgroup ids remapped so that an XCD walks a contiguous range of tiles (its L2 then sees one A panel at a time).
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

static inline int64_t pad4(int64_t n) { return (n + 3) & ~(int64_t)3; }


// grid (tiles_n * tiles_m), 256 threads = 2 x 2 waves of (BM/2) x (BN/2).  lda, ldw, ldc multiples of 4; rows of A
// beyond M are clamped (recomputed, never stored); columns N <= col < ldc of C are written as zeros (they are the K
// padding of the next layer).  BM, BN in {64, 128}: the launcher takes the largest tile that still gives every CU two
// workgroups (a 128 x 128 grid of the 5000 x 1000 layers is 320 workgroups for 256 CUs: 74 TF against 91-106 TF for
// the layers with thousands of tiles).
template <int BM, int BN, int BK, bool RELU, bool AFFINE>
__global__ __launch_bounds__(256, 3) void k_aug_gemm(const float* __restrict__ Ain, int lda, int M,
                                                     const float* __restrict__ W, int ldw, int N, int K,
                                                     const float* __restrict__ scale, const float* __restrict__ shift,
                                                     float* __restrict__ Cout, int ldc, int tiles_n, int tiles_m) {
    constexpr int TI = BM / 64, TJ = BN / 64;      // 32 x 32 accumulators per wave
    constexpr int TPR = BK / 4, RPP = 256 / TPR;   // threads per tile row (one float4 each), rows per pass
    constexpr int LA = BM / RPP, LB = BN / RPP;    // float4 loads per thread and K tile
    constexpr int LD = BK + 4;                     // LDS row stride: LD / 4 odd, conflict-free b128 fragment reads
    __shared__ __attribute__((aligned(16))) float As[BM * LD];
    __shared__ __attribute__((aligned(16))) float Bs[BN * LD];
    // XCD-aware tile order: workgroup i runs on XCD i % 8; give XCD x the tiles [x * per, (x + 1) * per)
    const int nwg = tiles_n * tiles_m;
    int wg = blockIdx.x;
    if (nwg % 8 == 0) wg = (wg & 7) * (nwg >> 3) + (wg >> 3);
    const int tn = wg % tiles_n, tm = wg / tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int l31 = lane & 31, hh = lane >> 5;
    const int wm = wv >> 1, wn = wv & 1;
    const int r0 = tid / TPR, c4 = tid % TPR;
    const int nkt = cdiv(K, BK);

    const float* pa[LA];
    const float* pb[LB];
    bool okb[LB];
#pragma unroll
    for (int i = 0; i < LA; ++i) pa[i] = Ain + (int64_t)min(m0 + r0 + RPP * i, M - 1) * lda + c4 * 4;
#pragma unroll
    for (int i = 0; i < LB; ++i) {
        const int rb = n0 + r0 + RPP * i;
        okb[i] = rb < N;
        pb[i] = W + (int64_t)min(rb, N - 1) * ldw + c4 * 4;
    }
    f32x16 acc[TI][TJ];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j) acc[i][j] = zero16();

    float4 ra4[LA], rb4[LB];
    auto load_tiles = [&](int kt) {
        const bool colok = kt * BK + c4 * 4 < K;     // K padded to 4: a float4 is all in or all out
        const int koff = colok ? kt * BK : 0;
#pragma unroll
        for (int i = 0; i < LA; ++i) ra4[i] = *reinterpret_cast<const float4*>(pa[i] + koff);
#pragma unroll
        for (int i = 0; i < LB; ++i) rb4[i] = *reinterpret_cast<const float4*>(pb[i] + koff);
#pragma unroll
        for (int i = 0; i < LA; ++i)
            if (!colok) ra4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int i = 0; i < LB; ++i)
            if (!(colok && okb[i])) rb4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    };
    // One LDS buffer per operand and two barriers per K tile.  Double-buffering the LDS tiles (one barrier) measured
    // slower at the benchmark shape (2.67 ms against 2.22 ms per call): twice the LDS per workgroup for no gain, the
    // second resident workgroup already covers the barrier.
    load_tiles(0);
    for (int kt = 0; kt < nkt; ++kt) {
#pragma unroll
        for (int i = 0; i < LA; ++i) *reinterpret_cast<float4*>(&As[(r0 + RPP * i) * LD + c4 * 4]) = ra4[i];
#pragma unroll
        for (int i = 0; i < LB; ++i) *reinterpret_cast<float4*>(&Bs[(r0 + RPP * i) * LD + c4 * 4]) = rb4[i];
        __syncthreads();
        if (kt + 1 < nkt) load_tiles(kt + 1);
        const float* la = As + (wm * (BM / 2) + l31) * LD + 4 * hh;
        const float* lb = Bs + (wn * (BN / 2) + l31) * LD + 4 * hh;
#pragma unroll
        for (int g = 0; g < BK / 8; ++g) {
            float4 a[TI], q[TJ];
#pragma unroll
            for (int i = 0; i < TI; ++i) a[i] = *reinterpret_cast<const float4*>(la + 32 * i * LD + 8 * g);
#pragma unroll
            for (int j = 0; j < TJ; ++j) q[j] = *reinterpret_cast<const float4*>(lb + 32 * j * LD + 8 * g);
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int j = 0; j < TJ; ++j) acc[i][j] = mfma32(a[i].x, q[j].x, acc[i][j]);
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int j = 0; j < TJ; ++j) acc[i][j] = mfma32(a[i].y, q[j].y, acc[i][j]);
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int j = 0; j < TJ; ++j) acc[i][j] = mfma32(a[i].z, q[j].z, acc[i][j]);
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int j = 0; j < TJ; ++j) acc[i][j] = mfma32(a[i].w, q[j].w, acc[i][j]);
        }
        __syncthreads();
    }
    // epilogue: lane l31 owns one column of each 32-wide tile
#pragma unroll
    for (int j = 0; j < TJ; ++j) {
        const int col = n0 + wn * (BN / 2) + j * 32 + l31;
        if (col >= ldc) continue;
        const bool real = col < N;
        const float sc = (AFFINE && real) ? scale[col] : 1.f;
        const float sh = (AFFINE && real) ? shift[col] : 0.f;
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * (BM / 2) + i * 32 + acc_row(r, lane);
                if (row < M) {
                    float v = acc[i][j][r] * sc + sh;
                    if (RELU) v = fmaxf(v, 0.f);
                    Cout[(int64_t)row * ldc + col] = real ? v : 0.f;
                }
            }
    }
}

// ---------------------------------------------------------------------------------------------
// packing (once per set of weights)
// ---------------------------------------------------------------------------------------------
// dst[r][c] = c < K ? src[r * src_ld + col0 + c] : 0,  r < N, c < dst_ld
__global__ void k_aug_pack_w(const float* __restrict__ src, int src_ld, int col0, int N, int K,
                             float* __restrict__ dst, int dst_ld) {
    const int64_t n = (int64_t)N * dst_ld;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int r = (int)(i / dst_ld), c = (int)(i % dst_ld);
        dst[i] = c < K ? src[(int64_t)r * src_ld + col0 + c] : 0.f;
    }
}
// BatchNorm (eval) folded with the Linear bias: y = ((acc + b) - mean) * rstd * gamma + beta
//   scale = gamma * rsqrt(var + eps),  shift = (b - mean) * scale + beta;  mean == null: scale = 1, shift = b
__global__ void k_aug_pack_affine(const float* __restrict__ bias, const float* __restrict__ mean,
                                  const float* __restrict__ var, const float* __restrict__ gamma,
                                  const float* __restrict__ beta, float eps, int n, float* __restrict__ scale,
                                  float* __restrict__ shift) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float b = bias ? bias[i] : 0.f;
    if (!mean) { scale[i] = 1.f; shift[i] = b; return; }
    const float sc = (gamma ? gamma[i] : 1.f) / sqrtf(var[i] + eps);
    scale[i] = sc;
    shift[i] = (b - mean[i]) * sc + (beta ? beta[i] : 0.f);
}

// ---------------------------------------------------------------------------------------------
// packed layout
// ---------------------------------------------------------------------------------------------
struct AugLayer {         // offsets in floats into the packed buffer
    int64_t w, sc, sh;
    int N, K, ldw;
    int64_t pl;           // the weight's three bf16 slice planes [3][Np][Kp] (fp32x3 engine; gemm_bf16.hip), Np = N padded to
    int Np, Kp;           // whole 128-row tiles, Kp = ldw padded to whole K tiles of 32
    int64_t tp;           // the same three slices as TILED planes (gemm_pp.hip: [3][ceil(ldw / 16)][N rounded up to 256][16] bf16)
};
struct AugPacked {
    AugLayer g[10];          // fc1 fc2 fc3 fc4 fc5[:, :n]  fc7 fc8 fc9 fc10 fc11
    // latent block
    int64_t noise_w, z_sc, z_sh;      // [NZ][NZ], [NZ], [NZ]
    int64_t w5b, sc5, sh5;            // [N5][NZ], [N5], [N5]
    int64_t wmu, scmu, shmu;          // [Z][N5], [Z], [Z]
    int64_t wsig, bsig;               // [Z][N5], [Z]
    int64_t w6, sc6, sh6;             // [N5][Z], [N5], [N5]
    int64_t total;
};

static AugPacked aug_packed_layout(const mmvae_aug_dims& d) {
    AugPacked p{};
    int64_t off = 0;
    auto take = [&](int64_t n) { const int64_t o = off; off += pad4(n); return o; };
    const int D = d.D, N1 = d.N1, N3 = d.N3, N5 = d.N5, Z = d.Z, NZ = d.NZ;
    const int NK[10][2] = {{N1, D}, {N1, N1}, {N3, N1}, {N3, N3}, {N5, N3}, {N3, N5}, {N3, N3}, {N1, N3}, {N1, N1}, {D, N1}};
    for (int i = 0; i < 10; ++i) {
        AugLayer& g = p.g[i];
        g.N = NK[i][0];
        g.K = NK[i][1];
        g.ldw = (int)pad4(g.K);
        g.w = take((int64_t)g.N * g.ldw);
        g.sc = take(g.N);
        g.sh = take(g.N);
        g.Np = (g.N + 127) / 128 * 128;
        g.Kp = (g.ldw + 31) / 32 * 32;
        g.pl = take((int64_t)3 * g.Np * g.Kp / 2);     // bf16: two per float
        g.tp = take(3 * tp_plane_elems(g.N, g.ldw) / 2);
    }
    p.noise_w = take((int64_t)NZ * NZ); p.z_sc = take(NZ); p.z_sh = take(NZ);
    p.w5b = take((int64_t)N5 * NZ); p.sc5 = take(N5); p.sh5 = take(N5);
    p.wmu = take((int64_t)Z * N5); p.scmu = take(Z); p.shmu = take(Z);
    p.wsig = take((int64_t)Z * N5); p.bsig = take(Z);
    p.w6 = take((int64_t)N5 * Z); p.sc6 = take(N5); p.sh6 = take(N5);
    p.total = off;
    return p;
}

struct AugWs {
    int64_t h1, h2, h3, h4, P, H6, h7, h8, h9, h10, total; int ld1, ld3, ld5;
    // planes x planes engine (gemm_pp.hip): the activations as tiled slice planes (three planes' room each; the bf16
    // configuration fills one), written by the producing layer's epilogue; scratch of the stream-K grid (flags + partial tiles)
    int64_t tx, t1, t2, t3, t4, t6, t7, t8, t9, t10, scratch, scratch_floats;
    int64_t rowmap; int rowmap_n;     // uint32 [trunk rows rounded up to 256]: the batch's rows in the resident matrix' planes (mmvae_augment_rows)
};
static AugWs aug_ws_layout(const mmvae_aug_dims& d, int trunk_rows) {
    AugWs w{};
    int64_t off = 0;
    auto take = [&](int64_t n) { const int64_t o = off; off += (n + 63) & ~(int64_t)63; return o; };
    w.ld1 = (int)pad4(d.N1); w.ld3 = (int)pad4(d.N3); w.ld5 = (int)pad4(d.N5);
    const int64_t T = trunk_rows, R = (int64_t)d.A * d.B;
    w.h1 = take(T * w.ld1); w.h2 = take(T * w.ld1); w.h3 = take(T * w.ld3); w.h4 = take(T * w.ld3); w.P = take(T * w.ld5);
    w.H6 = take(R * w.ld5); w.h7 = take(R * w.ld3); w.h8 = take(R * w.ld3); w.h9 = take(R * w.ld1); w.h10 = take(R * w.ld1);
    auto tpl = [&](int rows, int K) { return take(3 * tp_plane_elems(rows, K) / 2); };
    w.tx = tpl((int)T, d.D); w.t1 = tpl((int)T, d.N1); w.t2 = tpl((int)T, d.N1); w.t3 = tpl((int)T, d.N3); w.t4 = tpl((int)T, d.N3);
    w.t6 = tpl((int)R, d.N5); w.t7 = tpl((int)R, d.N3); w.t8 = tpl((int)R, d.N3); w.t9 = tpl((int)R, d.N1); w.t10 = tpl((int)R, d.N1);
    w.scratch_floats = pp_scratch_floats();
    w.scratch = take(w.scratch_floats);
    w.rowmap_n = tp_rp((int)T);
    w.rowmap = take(w.rowmap_n);
    w.total = off;
    return w;
}

constexpr int AL_NW = 8;   // waves per workgroup of the latent kernel
static size_t aug_latent_lds_bytes(const mmvae_aug_dims& d) {
    return sizeof(float) * ((size_t)d.NZ * (d.NZ + 1) + (size_t)d.N5 * (d.NZ + 1) + 2 * (size_t)d.Z * (d.N5 + 1) +
                            (size_t)d.N5 * (d.Z + 1) + AL_NW * 128);
}

static int aug_check_dims(const mmvae_aug_dims* d) {
    if (!d) { set_error("aug dims is null"); return MMVAE_E_BADARG; }
    if (d->A < 1 || d->A > MMVAE_MAX_ARMS || d->B < 1 || d->D < 4 || d->N1 < 1 || d->N3 < 1 || d->N5 < 1 || d->Z < 1 || d->NZ < 1) {
        set_error("aug dims: non-positive size");
        return MMVAE_E_BADARG;
    }
    if (d->D % 4 != 0) { set_error("augmenter: input_dim must be a multiple of 4 (16-byte row loads of x)"); return MMVAE_E_UNSUPPORTED; }
    if (d->N5 > 128 || d->NZ > 128 || d->Z > 64) {
        set_error("augmenter: n_dim/5 and noise_dim <= 128, latent_dim <= 64");
        return MMVAE_E_UNSUPPORTED;
    }
    if (aug_latent_lds_bytes(*d) > 160 * 1024) {
        set_error("augmenter: the latent block's weights (noise, fc5 noise part, fc_mu, fc_sigma, fc6) exceed 160 KB of LDS");
        return MMVAE_E_UNSUPPORTED;
    }
    return 0;
}

// ---------------------------------------------------------------------------------------------
// latent block, one wave per cell-arm row (udagan.py:289-305 / :314-322):
//   z  = elu(bnz(noise(scale * z0)))                        noise Linear has no bias; bnz is affine
//   h5 = relu(bn5(P[cell] + W5[:, n:] z + b5))              P = h4 W5[:, :n]^T from the GEMM
//   mu = bn_mu(fc_mu(h5)),  sigma = sigmoid(fc_sigma(h5)),  s = eps * sigma + mu      (aug_utils.py:51-65)
//   h6 = relu(bn6(fc6(s)))
// Lane j owns outputs j and j + 64; inputs are broadcast from a per-wave LDS row.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64 * AL_NW) void k_aug_latent(const float* __restrict__ pk, AugPacked L, int A, int B,
                                                          int N5, int Z, int NZ, int trunk_shared,
                                                          const float* __restrict__ P, int ld5,
                                                          const float* __restrict__ z0, const float* __restrict__ eps_n,
                                                          float zscale, float* __restrict__ s_out,
                                                          float* __restrict__ H6, TPlanes h6p, int h6_np) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* Wn = sm;                      // [NZ][NZ+1]
    float* W5 = Wn + NZ * (NZ + 1);      // [N5][NZ+1]
    float* Wm = W5 + N5 * (NZ + 1);      // [Z][N5+1]
    float* Ws = Wm + Z * (N5 + 1);       // [Z][N5+1]
    float* W6 = Ws + Z * (N5 + 1);       // [N5][Z+1]
    float* rowbuf = W6 + N5 * (Z + 1);   // [AL_NW][128]
    for (int i = threadIdx.x; i < NZ * NZ; i += blockDim.x) Wn[(i / NZ) * (NZ + 1) + i % NZ] = pk[L.noise_w + i];
    for (int i = threadIdx.x; i < N5 * NZ; i += blockDim.x) W5[(i / NZ) * (NZ + 1) + i % NZ] = pk[L.w5b + i];
    for (int i = threadIdx.x; i < Z * N5; i += blockDim.x) {
        Wm[(i / N5) * (N5 + 1) + i % N5] = pk[L.wmu + i];
        Ws[(i / N5) * (N5 + 1) + i % N5] = pk[L.wsig + i];
    }
    for (int i = threadIdx.x; i < N5 * Z; i += blockDim.x) W6[(i / Z) * (Z + 1) + i % Z] = pk[L.w6 + i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    float* rb = rowbuf + wv * 128;
    const int64_t R = (int64_t)A * B;
    const int j0 = lane, j1 = lane + 64;
    // per-lane constants
    const float zsc0 = j0 < NZ ? pk[L.z_sc + j0] : 0.f, zsh0 = j0 < NZ ? pk[L.z_sh + j0] : 0.f;
    const float zsc1 = j1 < NZ ? pk[L.z_sc + j1] : 0.f, zsh1 = j1 < NZ ? pk[L.z_sh + j1] : 0.f;
    const float s5c0 = j0 < N5 ? pk[L.sc5 + j0] : 0.f, s5h0 = j0 < N5 ? pk[L.sh5 + j0] : 0.f;
    const float s5c1 = j1 < N5 ? pk[L.sc5 + j1] : 0.f, s5h1 = j1 < N5 ? pk[L.sh5 + j1] : 0.f;
    const float smc = lane < Z ? pk[L.scmu + lane] : 0.f, smh = lane < Z ? pk[L.shmu + lane] : 0.f;
    const float bsg = lane < Z ? pk[L.bsig + lane] : 0.f;
    const float s6c0 = j0 < N5 ? pk[L.sc6 + j0] : 0.f, s6h0 = j0 < N5 ? pk[L.sh6 + j0] : 0.f;
    const float s6c1 = j1 < N5 ? pk[L.sc6 + j1] : 0.f, s6h1 = j1 < N5 ? pk[L.sh6 + j1] : 0.f;

    for (int64_t row = (int64_t)blockIdx.x * AL_NW + wv; row < R; row += (int64_t)gridDim.x * AL_NW) {
        const int64_t cell = trunk_shared ? row % B : row;
        // ---- noise branch
        if (j0 < NZ) rb[j0] = zscale * z0[row * NZ + j0];
        if (j1 < NZ) rb[j1] = zscale * z0[row * NZ + j1];
        float a0 = 0.f, a1 = 0.f;
        for (int k = 0; k < NZ; ++k) {
            const float v = rb[k];
            if (j0 < NZ) a0 += Wn[j0 * (NZ + 1) + k] * v;
            if (j1 < NZ) a1 += Wn[j1 * (NZ + 1) + k] * v;
        }
        a0 = a0 * zsc0 + zsh0;
        a1 = a1 * zsc1 + zsh1;
        a0 = a0 > 0.f ? a0 : expm1f(a0);   // F.elu, alpha = 1
        a1 = a1 > 0.f ? a1 : expm1f(a1);
        if (j0 < NZ) rb[j0] = a0;
        if (j1 < NZ) rb[j1] = a1;
        // ---- fc5 (noise part) + P, BN, ReLU
        float h0 = j0 < N5 ? P[cell * ld5 + j0] : 0.f, h1 = j1 < N5 ? P[cell * ld5 + j1] : 0.f;
        for (int k = 0; k < NZ; ++k) {
            const float v = rb[k];
            if (j0 < N5) h0 += W5[j0 * (NZ + 1) + k] * v;
            if (j1 < N5) h1 += W5[j1 * (NZ + 1) + k] * v;
        }
        h0 = fmaxf(h0 * s5c0 + s5h0, 0.f);
        h1 = fmaxf(h1 * s5c1 + s5h1, 0.f);
        if (j0 < N5) rb[j0] = h0;
        if (j1 < N5) rb[j1] = h1;
        // ---- heads + reparameterisation (lane < Z)
        float mu = 0.f, sg = 0.f;
        if (lane < Z)
            for (int k = 0; k < N5; ++k) {
                const float v = rb[k];
                mu += Wm[lane * (N5 + 1) + k] * v;
                sg += Ws[lane * (N5 + 1) + k] * v;
            }
        mu = mu * smc + smh;
        sg = 1.f / (1.f + expf(-(sg + bsg)));
        float sv = 0.f;
        if (lane < Z) {
            sv = eps_n[row * Z + lane] * sg + mu;
            s_out[row * Z + lane] = sv;
        }
        // every lane has read h5 (the loop above) before it is overwritten: same wave, program order
        if (lane < Z) rb[lane] = sv;
        // ---- fc6, BN, ReLU
        float g0 = 0.f, g1 = 0.f;
        for (int k = 0; k < Z; ++k) {
            const float v = rb[k];
            if (j0 < N5) g0 += W6[j0 * (Z + 1) + k] * v;
            if (j1 < N5) g1 += W6[j1 * (Z + 1) + k] * v;
        }
        const float o0 = j0 < N5 ? fmaxf(g0 * s6c0 + s6h0, 0.f) : 0.f, o1 = j1 < N5 ? fmaxf(g1 * s6c1 + s6h1, 0.f) : 0.f;
        if (h6_np) {
            // the planes x planes engine takes h6 as tiled slice planes (gemm_pp.hip): the even lane writes its column and its
            // right neighbour's as one dword per plane (no fp32 copy, no conversion launch)
            const float n0 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, o0), 0xF5, 0xF, 0xF, false));
            const float n1 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, o1), 0xF5, 0xF, 0xF, false));
            if (!(lane & 1)) {
                const float v[2] = {o0, o1}, nb[2] = {n0, n1};
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int n = lane + 64 * q;
                    if ((n >> 4) >= h6p.KT) continue;
                    const int64_t e = ((int64_t)(n >> 4) * h6p.Rp + row) * 16 + (n & 15);
                    if (h6_np == 3) {
                        unsigned w[3];
                        split3(v[q], nb[q], w);
#pragma unroll
                        for (int pl = 0; pl < 3; ++pl) *reinterpret_cast<unsigned*>(h6p.p + pl * h6p.plane + e) = w[pl];
                    } else {
                        *reinterpret_cast<unsigned*>(h6p.p + e) = cvt_pk_bf16(v[q], nb[q]);
                    }
                }
            }
        } else {
            if (j0 < ld5) H6[row * ld5 + j0] = o0;
            if (j1 < ld5) H6[row * ld5 + j1] = o1;
        }
    }
}

template <int BM, int BN, int BK>
static void aug_gemm_launch(hipStream_t s, bool relu, bool affine, const float* A, int lda, int M, const float* W, int ldw,
                            int N, int K, const float* sc, const float* sh, float* C, int ldc, int ncols) {
    const int tiles_n = cdiv(ncols, BN), tiles_m = cdiv(M, BM);
    dim3 grid(tiles_n * tiles_m), block(256);
    if (relu) hipLaunchKernelGGL((k_aug_gemm<BM, BN, BK, true, true>), grid, block, 0, s, A, lda, M, W, ldw, N, K, sc, sh, C, ldc, tiles_n, tiles_m);
    else if (affine) hipLaunchKernelGGL((k_aug_gemm<BM, BN, BK, false, true>), grid, block, 0, s, A, lda, M, W, ldw, N, K, sc, sh, C, ldc, tiles_n, tiles_m);
    else hipLaunchKernelGGL((k_aug_gemm<BM, BN, BK, false, false>), grid, block, 0, s, A, lda, M, W, ldw, N, K, sc, sh, C, ldc, tiles_n, tiles_m);
}

static int aug_gemm(hipStream_t s, int force_tile, bool relu, bool affine, const float* A, int lda, int M, const float* pk,
                    const AugLayer& g, float* C, int ldc, float* scratch = nullptr, int64_t scratch_floats = 0) {
    const int ncols = ldc < (int)pad4(g.N) ? ldc : (int)pad4(g.N);   // the K padding of the next layer is written too (zeros)
    const float* W = pk + g.w;
    const float* sc = pk + g.sc;
    const float* sh = pk + g.sh;
    if (force_tile == 99 || force_tile == 98)   // bf16 operands (mmvae_augment's gemm_bf16 = 1) or fp32 operands split into
        // three bf16 slices (gemm_bf16 = 2): the shared tile engine of gemm_bf16.hip, fp32 epilogue
        return launch_bf16_affine(s, relu, affine, A, lda, M, W, g.ldw, g.N, g.ldw, sc, sh, C, ldc, ncols, force_tile == 98,
                                  reinterpret_cast<const unsigned short*>(pk + g.pl), g.Np, g.Kp, scratch, scratch_floats);
    // the largest tile that still leaves two workgroups per CU (256 CUs); MMVAE_AUG_TILE=<BM><BN> code forces one
    const int force = force_tile;   // 11 12 21 22 (1 = 64, 2 = 128), 0 = automatic
    auto count = [&](int bm, int bn) { return (int64_t)cdiv(M, bm) * cdiv(ncols, bn); };
    int pick = 11;
    if (count(128, 128) >= 512) pick = 22;
    else if (count(64, 128) >= 512) pick = 12;
    else if (count(128, 64) >= 512) pick = 21;
    if (force) pick = force;
    // K tile 32.  BK = 64 (half the barriers, twice the LDS and prefetch registers) measured slower: 2.36 against 2.23 ms
    // per call at the benchmark shape; so did double-buffered LDS tiles.  The barrier is not what limits this kernel.
#define AG_ARGS s, relu, affine, A, lda, M, W, g.ldw, g.N, g.K, sc, sh, C, ldc, ncols
    switch (pick) {
        case 22: aug_gemm_launch<128, 128, 32>(AG_ARGS); break;
        case 12: aug_gemm_launch<64, 128, 32>(AG_ARGS); break;
        case 21: aug_gemm_launch<128, 64, 32>(AG_ARGS); break;
        default: aug_gemm_launch<64, 64, 32>(AG_ARGS); break;
    }
#undef AG_ARGS
    HIP_LAUNCH_CHECK("k_aug_gemm");
    return 0;
}

}  // namespace mmvae

using namespace mmvae;

extern "C" {

size_t mmvae_aug_packed_floats(const mmvae_aug_dims* d) {
    if (aug_check_dims(d)) return 0;
    return (size_t)aug_packed_layout(*d).total;
}

size_t mmvae_aug_workspace_bytes(const mmvae_aug_dims* d, int shared_x) {
    if (aug_check_dims(d)) return 0;
    return (size_t)aug_ws_layout(*d, shared_x ? d->B : d->A * d->B).total * sizeof(float);
}

int mmvae_aug_pack(const mmvae_aug_dims* d, const mmvae_aug_tensors* t, float* packed, void* stream) {
    if (int rc = aug_check_dims(d)) return rc;
    if (!t || !packed) { set_error("aug_pack: null argument"); return MMVAE_E_BADARG; }
    for (int i = 0; i < 11; ++i)
        if (!t->w[i] || !t->b[i]) { set_error("aug_pack: null weight or bias"); return MMVAE_E_BADARG; }
    for (int i = 0; i < 10; ++i)
        if (!t->bn_mean[i] || !t->bn_var[i]) { set_error("aug_pack: null BatchNorm buffer"); return MMVAE_E_BADARG; }
    if (!t->w_mu || !t->b_mu || !t->w_sigma || !t->b_sigma || !t->noise_w || !t->bnz_mean || !t->bnz_var ||
        !t->bnz_weight || !t->bnz_bias || !t->bn_mu_mean || !t->bn_mu_var) {
        set_error("aug_pack: null latent-block tensor");
        return MMVAE_E_BADARG;
    }
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const AugPacked L = aug_packed_layout(*d);
    const float eps = 1e-10f;   // udagan.py:230-276 (every BatchNorm1d but bnz)
    auto packw = [&](const float* src, int src_ld, int col0, int N, int K, int64_t dst, int dst_ld) {
        const int64_t n = (int64_t)N * dst_ld;
        hipLaunchKernelGGL(k_aug_pack_w, dim3((unsigned)imin64(4096, cdiv64(n, 256))), dim3(256), 0, s, src, src_ld, col0, N, K,
                           packed + dst, dst_ld);
    };
    auto packa = [&](const float* b, const float* m, const float* v, const float* ga, const float* be, float e, int n,
                     int64_t sc, int64_t sh) {
        hipLaunchKernelGGL(k_aug_pack_affine, dim3(cdiv(n, 256)), dim3(256), 0, s, b, m, v, ga, be, e, n, packed + sc, packed + sh);
    };
    // module layer index (w[], b[], bn_*[]): 0 fc1 .. 3 fc4, 4 fc5, 5 fc6, 6 fc7 .. 9 fc10, 10 fc11
    const int mod_of_g[10] = {0, 1, 2, 3, 4, 6, 7, 8, 9, 10};
    for (int i = 0; i < 10; ++i) {
        const AugLayer& g = L.g[i];
        const int mi = mod_of_g[i];
        const int src_ld = (mi == 4) ? d->N3 + d->NZ : g.K;
        packw(t->w[mi], src_ld, 0, g.N, g.K, g.w, g.ldw);
        // slice planes of the packed (zero-padded) weight for the fp32x3 engine
        if (int rc = launch_presplit_one(s, packed + g.w, g.ldw, g.N, g.ldw, g.Np, g.Kp, reinterpret_cast<unsigned short*>(packed + g.pl)))
            return rc;
        if (int rc = launch_tp_from_f32(s, packed + g.w, g.ldw, g.N, g.ldw, 3, tp_make(reinterpret_cast<unsigned short*>(packed + g.tp), g.N, g.ldw)))
            return rc;
        if (mi == 4) continue;                                  // fc5's affine is applied in the latent kernel
        if (mi == 10) packa(t->b[10], nullptr, nullptr, nullptr, nullptr, 0.f, g.N, g.sc, g.sh);
        else packa(t->b[mi], t->bn_mean[mi], t->bn_var[mi], nullptr, nullptr, eps, g.N, g.sc, g.sh);
    }
    packw(t->noise_w, d->NZ, 0, d->NZ, d->NZ, L.noise_w, d->NZ);
    packa(nullptr, t->bnz_mean, t->bnz_var, t->bnz_weight, t->bnz_bias, 1e-5f, d->NZ, L.z_sc, L.z_sh);   // nn.BatchNorm1d default eps
    packw(t->w[4], d->N3 + d->NZ, d->N3, d->N5, d->NZ, L.w5b, d->NZ);
    packa(t->b[4], t->bn_mean[4], t->bn_var[4], nullptr, nullptr, eps, d->N5, L.sc5, L.sh5);
    packw(t->w_mu, d->N5, 0, d->Z, d->N5, L.wmu, d->N5);
    packa(t->b_mu, t->bn_mu_mean, t->bn_mu_var, nullptr, nullptr, eps, d->Z, L.scmu, L.shmu);
    packw(t->w_sigma, d->N5, 0, d->Z, d->N5, L.wsig, d->N5);
    if (hipMemcpyAsync(packed + L.bsig, t->b_sigma, sizeof(float) * d->Z, hipMemcpyDeviceToDevice, s) != hipSuccess) {
        set_error("aug_pack: copy failed");
        return MMVAE_E_LAUNCH;
    }
    packw(t->w[5], d->Z, 0, d->N5, d->Z, L.w6, d->Z);
    packa(t->b[5], t->bn_mean[5], t->bn_var[5], nullptr, nullptr, eps, d->N5, L.sc6, L.sh6);
    HIP_LAUNCH_CHECK("aug_pack");
    return 0;
}

// xp: the resident matrix as tiled slice planes (mmvae_tp_planes) and the batch's rows in it -- the first layer's DMA then
// reads the rows in place (no gathered batch, no per-batch conversion) --, or null: x is the batch itself
static int augment_impl(const mmvae_aug_dims* d, const float* packed, const float* x, int64_t x_arm_stride, const TPlanes* xp,
                        int64_t xp_rows, const int64_t* rows, const float* z0,
                        const float* eps_n, float scale, void* ws, size_t ws_bytes, float* s_out, float* x_aug,
                        int gemm_bf16, const mmvae_exec* ex, void* stream) {
    if (int rc = aug_check_dims(d)) return rc;
    if (!packed || (!x && !xp) || !z0 || !eps_n || !ws || !s_out || !x_aug) { set_error("augment: null argument"); return MMVAE_E_BADARG; }
    const bool shared = x_arm_stride == 0;
    if (!shared && x_arm_stride != (int64_t)d->B * d->D) {
        set_error("augment: x must be [B,D] shared by the arms (stride 0) or contiguous [A,B,D]");
        return MMVAE_E_BADARG;
    }
    const int T = shared ? d->B : d->A * d->B, R = d->A * d->B;
    const AugWs W = aug_ws_layout(*d, T);
    if (ws_bytes < (size_t)W.total * sizeof(float)) { set_error("augment: workspace too small"); return MMVAE_E_WORKSPACE; }
    const AugPacked L = aug_packed_layout(*d);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    float* w = reinterpret_cast<float*>(ws);
    const int ft = gemm_bf16 == 2 ? 98 : gemm_bf16 ? 99 : (ex ? ex->tune[MMVAE_TUNE_AUG_TILE] : 0);
    int rc;
    const int tune_tile = ex ? ex->tune[MMVAE_TUNE_AUG_TILE] : 0;
    if (xp && !(gemm_bf16 && tune_tile != 90)) { set_error("augment_rows: needs the planes engine (gemm_bf16 1 or 2)"); return MMVAE_E_UNSUPPORTED; }
    if (gemm_bf16 && tune_tile != 90) {
        // planes x planes engine (gemm_pp.hip): every layer's epilogue writes the next layer's operand as tiled slice planes
        // (three exact slices: fp32x3; one rounded plane: the bf16 configuration).  MMVAE_AUG_TILE=90: the tile engine of
        // gemm_bf16.hip as before (A/B timing); 1..3 (+ 10 x workgroups): forced tile / grid for every layer.
        const int NP = gemm_bf16 == 2 ? 3 : 1;
        auto tpa = [&](int64_t off, int rows, int K) { return tp_make(reinterpret_cast<unsigned short*>(w + off), rows, K); };
        auto tpw = [&](int i) {
            return tp_make(const_cast<unsigned short*>(reinterpret_cast<const unsigned short*>(packed + L.g[i].tp)), L.g[i].N, L.g[i].ldw);
        };
        const TPlanes X = tpa(W.tx, T, d->D), H1 = tpa(W.t1, T, d->N1), H2 = tpa(W.t2, T, d->N1), H3 = tpa(W.t3, T, d->N3),
                      H4 = tpa(W.t4, T, d->N3), H6 = tpa(W.t6, R, d->N5), H7 = tpa(W.t7, R, d->N3), H8 = tpa(W.t8, R, d->N3),
                      H9 = tpa(W.t9, R, d->N1), H10 = tpa(W.t10, R, d->N1);
        float* const scr = w + W.scratch;
        unsigned* const rmap = reinterpret_cast<unsigned*>(w + W.rowmap);
        auto layer = [&](int i, const TPlanes& in, int M, bool relu, bool affine, float* out32, int64_t ld32, int nc32, const TPlanes* outp,
                         bool mapped = false) {
            // the input's K steps cover its real width rounded up to 16; the weight planes' cover ldw (>= K, zero beyond K)
            TPlanes b = tpw(i);
            TPlanes a = in;
            if (a.KT > b.KT) a.KT = b.KT; else b.KT = a.KT;     // (equal unless the producer's width was padded differently)
            return launch_pp_gemm(s, NP, a, b, M, L.g[i].N, packed + L.g[i].sc, packed + L.g[i].sh, affine, relu, out32, ld32, nc32, outp,
                                  scr, W.scratch_floats, i, tune_tile, mapped ? rmap : nullptr, mapped ? W.rowmap_n : 0);
        };
        // (the forward's first launch also zeroes the flag words of the K-split combines)
        if (xp) {
            if ((rc = launch_pp_rowmap(s, rows, T, xp_rows, rmap, W.rowmap_n, scr))) return rc;
            if ((rc = layer(0, *xp, T, true, true, nullptr, 0, 0, &H1, true))) return rc;
        } else {
            if ((rc = launch_tp_from_f32(s, x, d->D, T, d->D, NP, X, scr))) return rc;
            if ((rc = layer(0, X, T, true, true, nullptr, 0, 0, &H1))) return rc;
        }
        if ((rc = layer(1, H1, T, true, true, nullptr, 0, 0, &H2))) return rc;
        if ((rc = layer(2, H2, T, true, true, nullptr, 0, 0, &H3))) return rc;
        if ((rc = layer(3, H3, T, true, true, nullptr, 0, 0, &H4))) return rc;
        if ((rc = layer(4, H4, T, false, false, w + W.P, W.ld5, W.ld5, nullptr))) return rc;
        {
            const size_t shm = aug_latent_lds_bytes(*d);
            if (shm > 64 * 1024)
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_aug_latent), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
            // (two rows per wave; one workgroup per CU with five rows per wave measured 69 against 45 us: the rows' dot-product
            // loops, not the 43 KB of weights every workgroup stages, are what the kernel spends its time on)
            const int blocks = (int)imin64(cdiv64(R, AL_NW * 2), 1024);
            hipLaunchKernelGGL(k_aug_latent, dim3(blocks), dim3(64 * AL_NW), shm, s, packed, L, d->A, d->B, d->N5, d->Z, d->NZ,
                               shared ? 1 : 0, w + W.P, W.ld5, z0, eps_n, scale, s_out, w + W.H6, H6, NP);
            HIP_LAUNCH_CHECK("k_aug_latent");
        }
        if ((rc = layer(5, H6, R, true, true, nullptr, 0, 0, &H7))) return rc;
        if ((rc = layer(6, H7, R, true, true, nullptr, 0, 0, &H8))) return rc;
        if ((rc = layer(7, H8, R, true, true, nullptr, 0, 0, &H9))) return rc;
        if ((rc = layer(8, H9, R, true, true, nullptr, 0, 0, &H10))) return rc;
        return layer(9, H10, R, true, true, x_aug, d->D, d->D, nullptr);
    }
    // trunk: once per cell when the arms share x
    // (the first layer may split K into slabs: the buffers of the last two hidden layers, h9 and h10, are adjacent and not in use yet)
    if ((rc = aug_gemm(s, ft, true, true, x, d->D, T, packed, L.g[0], w + W.h1, W.ld1, w + W.h9, W.total - W.h9))) return rc;
    if ((rc = aug_gemm(s, ft, true, true, w + W.h1, W.ld1, T, packed, L.g[1], w + W.h2, W.ld1))) return rc;
    if ((rc = aug_gemm(s, ft, true, true, w + W.h2, W.ld1, T, packed, L.g[2], w + W.h3, W.ld3))) return rc;
    if ((rc = aug_gemm(s, ft, true, true, w + W.h3, W.ld3, T, packed, L.g[3], w + W.h4, W.ld3))) return rc;
    if ((rc = aug_gemm(s, ft, false, false, w + W.h4, W.ld3, T, packed, L.g[4], w + W.P, W.ld5))) return rc;
    {
        const size_t shm = aug_latent_lds_bytes(*d);
        if (shm > 64 * 1024)
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_aug_latent), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
        const int blocks = (int)imin64(cdiv64(R, AL_NW * 2), 1024);
        hipLaunchKernelGGL(k_aug_latent, dim3(blocks), dim3(64 * AL_NW), shm, s, packed, L, d->A, d->B, d->N5, d->Z, d->NZ,
                           shared ? 1 : 0, w + W.P, W.ld5, z0, eps_n, scale, s_out, w + W.H6, TPlanes{}, 0);
        HIP_LAUNCH_CHECK("k_aug_latent");
    }
    if ((rc = aug_gemm(s, ft, true, true, w + W.H6, W.ld5, R, packed, L.g[5], w + W.h7, W.ld3))) return rc;
    if ((rc = aug_gemm(s, ft, true, true, w + W.h7, W.ld3, R, packed, L.g[6], w + W.h8, W.ld3))) return rc;
    if ((rc = aug_gemm(s, ft, true, true, w + W.h8, W.ld3, R, packed, L.g[7], w + W.h9, W.ld1))) return rc;
    if ((rc = aug_gemm(s, ft, true, true, w + W.h9, W.ld1, R, packed, L.g[8], w + W.h10, W.ld1))) return rc;
    return aug_gemm(s, ft, true, true, w + W.h10, W.ld1, R, packed, L.g[9], x_aug, d->D);
}


int mmvae_augment(const mmvae_aug_dims* d, const float* packed, const float* x, int64_t x_arm_stride, const float* z0,
                  const float* eps_n, float scale, void* ws, size_t ws_bytes, float* s_out, float* x_aug,
                  int gemm_bf16, const mmvae_exec* ex, void* stream) {
    return augment_impl(d, packed, x, x_arm_stride, nullptr, 0, nullptr, z0, eps_n, scale, ws, ws_bytes, s_out, x_aug, gemm_bf16, ex, stream);
}

size_t mmvae_tp_planes_bytes(int64_t n_rows, int32_t K, int32_t n_planes) {
    if (n_rows < 1 || n_rows > ((int64_t)1 << 27) || K < 4 || (K & 3) || (n_planes != 1 && n_planes != 3)) return 0;
    if (tp_plane_elems((int)n_rows, K) * 2 >= ((int64_t)1 << 32)) return 0;      // (a lane's 32-bit offset spans one plane)
    return (size_t)n_planes * (size_t)tp_plane_elems((int)n_rows, K) * 2;
}

int mmvae_tp_planes(const float* src, int64_t ld, int64_t n_rows, int32_t K, int32_t n_planes, uint16_t* dst, void* stream) {
    if (!src || !dst || ld < K || !mmvae_tp_planes_bytes(n_rows, K, n_planes)) {
        set_error("tp_planes: needs ld >= K, K a multiple of 4, one or three planes and less than 4 GB per plane");
        return MMVAE_E_BADARG;
    }
    return launch_rp_from_f32(reinterpret_cast<hipStream_t>(stream), src, ld, (int)n_rows, K, n_planes, tp_make(dst, (int)n_rows, K));
}

int mmvae_augment_rows(const mmvae_aug_dims* d, const float* packed, const uint16_t* x_planes, int64_t n_rows, int32_t n_planes,
                       const int64_t* rows, const float* z0, const float* eps_n, float scale, void* ws, size_t ws_bytes,
                       float* s_out, float* x_aug, int gemm_bf16, const mmvae_exec* ex, void* stream) {
    if (int rc = aug_check_dims(d)) return rc;
    if (!x_planes || !rows || !mmvae_tp_planes_bytes(n_rows, d->D, n_planes)) { set_error("augment_rows: bad planes argument"); return MMVAE_E_BADARG; }
    if (n_planes != (gemm_bf16 == 2 ? 3 : 1) || !gemm_bf16) {
        set_error("augment_rows: the planes must be the engine's (three for gemm_bf16 = 2, one for 1)");
        return MMVAE_E_UNSUPPORTED;
    }
    const TPlanes xp = tp_make(const_cast<unsigned short*>(x_planes), (int)n_rows, d->D);
    return augment_impl(d, packed, nullptr, 0, &xp, n_rows, rows, z0, eps_n, scale, ws, ws_bytes, s_out, x_aug, gemm_bf16, ex, stream);
}


}  // extern "C"
