// The three MFMA-bound kernels of the train step (fp32 in / fp32 accumulate,
// v_mfma_f32_32x32x2_f32: bit-for-bit an fmaf chain, so results track the fp32 reference).
//
//   k_fc1_fwd     z1 slabs  = (x .* mask) W1^T             nn_model.py:264   (split over K = genes)
//   k_fc11_fused  x_rec = relu(d10 W11^T + b11); squared-error / mismatch partials; dZ11;
//                 d loss / d d10 slabs                     nn_model.py:287, :542-546 + their autograd
//   k_gemm_tn     out = P^T Q over the batch: dW1 = dZ1^T x~, [dW11 | db11] = dZ11^T [d10 | 1],
//                 and the batched small-layer dW/db        autograd of every nn.Linear
//
// HBM layout: x [B,D] (or [A,B,D]), dZ11 [A,B,D] row-major; weights PyTorch [out,in].
#include "common.hpp"

namespace mmvae {

// ---------------------------------------------------------------------------------------------
// masked x loader: 4 consecutive genes of one cell, zero outside [B,D]; mask = dropout keep-mask
// (explicit bytes or Philox).  The 1/(1-p) scale is applied by the consumer.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float4 load_x4(const float* __restrict__ x, int row, int col, int B, int D,
                                          bool vec_ok) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (row < B) {
        const float* p = x + (int64_t)row * D + col;
        if (vec_ok && col + 3 < D) {
            v = *reinterpret_cast<const float4*>(p);
        } else {
            if (col < D) v.x = p[0];
            if (col + 1 < D) v.y = p[1];
            if (col + 2 < D) v.z = p[2];
            if (col + 3 < D) v.w = p[3];
        }
    }
    return v;
}

__device__ __forceinline__ float4 apply_xmask(float4 v, const NoiseDev& nz, int use_mask, int arm, int row,
                                              int col, int B, int D, bool vec_ok) {
    if (!use_mask || row >= B || col >= D) return v;
    const int64_t idx = ((int64_t)arm * B + row) * D + col;
    if (nz.mode == 0) {
        const uint8_t* m = nz.x_mask + idx;
        if (vec_ok && col + 3 < D) {
            const uint32_t w = *reinterpret_cast<const uint32_t*>(m);
            v.x = (w & 0xFFu) ? v.x : 0.f;
            v.y = (w & 0xFF00u) ? v.y : 0.f;
            v.z = (w & 0xFF0000u) ? v.z : 0.f;
            v.w = (w & 0xFF000000u) ? v.w : 0.f;
        } else {
            v.x = m[0] ? v.x : 0.f;
            if (col + 1 < D) v.y = m[1] ? v.y : 0.f;
            if (col + 2 < D) v.z = m[2] ? v.z : 0.f;
            if (col + 3 < D) v.w = m[3] ? v.w : 0.f;
        }
    } else {
        // col is a multiple of 4 and a Philox group holds >= 8 consecutive genes: one call serves the four
        const uint32_t epg_log2 = 7u - nz.x_mlog2, i0 = (uint32_t)col & ((1u << epg_log2) - 1u);
        const u32x4 w = xmask_words(nz, arm, (uint32_t)row, (uint32_t)col >> epg_log2);
        v.x = xmask_field_keep(nz, w, i0) ? v.x : 0.f;
        if (col + 1 < D) v.y = xmask_field_keep(nz, w, i0 + 1) ? v.y : 0.f;
        if (col + 2 < D) v.z = xmask_field_keep(nz, w, i0 + 2) ? v.z : 0.f;
        if (col + 3 < D) v.w = xmask_field_keep(nz, w, i0 + 3) ? v.w : 0.f;
    }
    return v;
}

// generic [rows x 4] loader of a row-major matrix, zero outside
__device__ __forceinline__ float4 load_m4(const float* __restrict__ m, int64_t ld, int row, int col, int R,
                                          int Cn, bool vec_ok) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (row < R) {
        const float* p = m + (int64_t)row * ld + col;
        if (vec_ok && col + 3 < Cn) {
            v = *reinterpret_cast<const float4*>(p);
        } else {
            if (col < Cn) v.x = p[0];
            if (col + 1 < Cn) v.y = p[1];
            if (col + 2 < Cn) v.z = p[2];
            if (col + 3 < Cn) v.w = p[3];
        }
    }
    return v;
}

// =============================================================================================
// fc1 forward:  slab[ks][a][b][0..127] = sum_{k in split ks} (x[b][k] * m[a][b][k]) * W1[a][n][k]
// grid (ceil(B/64), KS, A), 256 threads.  Block tile 64 x 128, K tile 32, waves 2(M) x 2(N).
// =============================================================================================
constexpr int F1_BM = 64, F1_BK = 32, F1_LD = 36;

__global__ __launch_bounds__(256) void k_fc1_fwd(const float* __restrict__ x, int64_t x_arm_stride,
                                                 const float* __restrict__ params, int64_t per_arm,
                                                 int64_t w_off, NoiseDev nz, int use_mask,
                                                 float* __restrict__ slab, int A, int B, int D, int H,
                                                 int KS, int vec_ok) {
    __shared__ __attribute__((aligned(16))) float As[F1_BM * F1_LD];
    __shared__ __attribute__((aligned(16))) float Bs[NP * F1_LD];
    const int arm = blockIdx.z, ks = blockIdx.y, b0 = blockIdx.x * F1_BM;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int wm = wv >> 1, wn = wv & 1;
    const float* xa = x + (int64_t)arm * x_arm_stride;
    const float* W = params + (int64_t)arm * per_arm + w_off;   // [H, D]
    const int nkt = cdiv(D, F1_BK);
    const int kt0 = (int)(((int64_t)ks * nkt) / KS), kt1 = (int)(((int64_t)(ks + 1) * nkt) / KS);

    f32x16 acc0 = zero16(), acc1 = zero16();
    float4 ra[2], rb[4];
    // thread -> (row, c4) assignments
    auto load_tiles = [&](int kt) {
        const int k0 = kt * F1_BK;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int idx = tid + i * 256, row = idx >> 3, c4 = idx & 7;
            float4 v = load_x4(xa, b0 + row, k0 + c4 * 4, B, D, vec_ok);
            ra[i] = apply_xmask(v, nz, use_mask, arm, b0 + row, k0 + c4 * 4, B, D, vec_ok);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = tid + i * 256, row = idx >> 3, c4 = idx & 7;
            rb[i] = load_m4(W, D, row, k0 + c4 * 4, H, D, vec_ok);
        }
    };
    auto store_tiles = [&]() {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int idx = tid + i * 256, row = idx >> 3, c4 = idx & 7;
            *reinterpret_cast<float4*>(&As[row * F1_LD + c4 * 4]) = ra[i];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = tid + i * 256, row = idx >> 3, c4 = idx & 7;
            *reinterpret_cast<float4*>(&Bs[row * F1_LD + c4 * 4]) = rb[i];
        }
    };

    if (kt0 < kt1) load_tiles(kt0);
    for (int kt = kt0; kt < kt1; ++kt) {
        store_tiles();
        __syncthreads();
        if (kt + 1 < kt1) load_tiles(kt + 1);
        const float* pa = As + (wm * 32 + (lane & 31)) * F1_LD + 4 * (lane >> 5);
        const float* pb0 = Bs + (wn * 64 + (lane & 31)) * F1_LD + 4 * (lane >> 5);
        const float* pb1 = pb0 + 32 * F1_LD;
#pragma unroll
        for (int g = 0; g < F1_BK / 8; ++g) {
            const float4 a = *reinterpret_cast<const float4*>(pa + 8 * g);
            const float4 p = *reinterpret_cast<const float4*>(pb0 + 8 * g);
            const float4 q = *reinterpret_cast<const float4*>(pb1 + 8 * g);
            acc0 = mfma32(a.x, p.x, acc0); acc1 = mfma32(a.x, q.x, acc1);
            acc0 = mfma32(a.y, p.y, acc0); acc1 = mfma32(a.y, q.y, acc1);
            acc0 = mfma32(a.z, p.z, acc0); acc1 = mfma32(a.z, q.z, acc1);
            acc0 = mfma32(a.w, p.w, acc0); acc1 = mfma32(a.w, q.w, acc1);
        }
        __syncthreads();
    }
    float* out = slab + (((int64_t)ks * A + arm) * B) * NP;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = b0 + wm * 32 + acc_row(r, lane);
        if (row < B) {
            float* o = out + (int64_t)row * NP + wn * 64 + (lane & 31);
            o[0] = acc0[r];
            o[32] = acc1[r];
        }
    }
}

// fc1 epilogue: R1 = relu(scale * sum_ks slab + b1), per-block column mean / M2 for BatchNorm.
// grid (ceil(B/32), A), 256 threads: thread t -> float4 column group t & 31 (columns 4c..4c+3), rows
// (t >> 5) + 8 i, i < 4.  All KS x 4 slab loads of a thread are independent float4 loads.
template <int CH>   // slab loads in flight per row and thread: the smallest of 4 / 8 / 16 that covers KS (clamped duplicates of
                    // the last slab cost a pass through the vector-memory pipe each: 16 issued for KS = 6 was 2.7 x the loads)
__global__ __launch_bounds__(256) void k_fc1_epi(const float* __restrict__ slab, const float* __restrict__ params,
                                                 int64_t per_arm, int64_t b_off, float scale,
                                                 float* __restrict__ R1, float* __restrict__ part,
                                                 long long* __restrict__ acc, int A, int B, int H, int KS) {
    __shared__ float sh[8][NP];
    const int arm = blockIdx.y, blk = blockIdx.x, b0 = blk * 32;
    const int c4 = threadIdx.x & 31, rg = threadIdx.x >> 5;
    const int nvalid = min(32, B - b0);
    const float* bp = params + (int64_t)arm * per_arm + b_off;
    float bias[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) bias[j] = (c4 * 4 + j < H) ? bp[min(c4 * 4 + j, H - 1)] : 0.f;
    float4 v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = b0 + rg + 8 * i;
        const int rc = min(row, B - 1);
        // split-K slabs: sixteen requested before the first add (a running-sum loop waits for every load in turn)
        float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int k0 = 0; k0 < KS; k0 += CH) {
            float4 t[CH];
#pragma unroll
            for (int k = 0; k < CH; ++k)
                t[k] = *reinterpret_cast<const float4*>(slab + (((int64_t)min(k0 + k, KS - 1) * A + arm) * B + rc) * NP + c4 * 4);
#pragma unroll
            for (int k = 0; k < CH; ++k) {
                const bool on = k0 + k < KS;
                z.x += on ? t[k].x : 0.f; z.y += on ? t[k].y : 0.f; z.z += on ? t[k].z : 0.f; z.w += on ? t[k].w : 0.f;
            }
        }
        const bool ok = row < B;
        z.x = ok ? relu_keep_nan(scale * z.x + bias[0]) : 0.f;
        z.y = ok ? relu_keep_nan(scale * z.y + bias[1]) : 0.f;
        z.z = ok ? relu_keep_nan(scale * z.z + bias[2]) : 0.f;
        z.w = ok ? relu_keep_nan(scale * z.w + bias[3]) : 0.f;
        v[i] = z;
        if (ok) {
            float* o = R1 + ((int64_t)arm * B + row) * H + c4 * 4;
            if ((H & 3) == 0) {
                if (c4 * 4 < H) *reinterpret_cast<float4*>(o) = z;
            } else {
                if (c4 * 4 < H) o[0] = z.x;
                if (c4 * 4 + 1 < H) o[1] = z.y;
                if (c4 * 4 + 2 < H) o[2] = z.z;
                if (c4 * 4 + 3 < H) o[3] = z.w;
            }
        }
    }
    // column sums over the 32 rows: 4 rows per thread, 8 row groups through LDS
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int i = 0; i < 4; ++i) { s.x += v[i].x; s.y += v[i].y; s.z += v[i].z; s.w += v[i].w; }
    *reinterpret_cast<float4*>(&sh[rg][c4 * 4]) = s;
    __syncthreads();
    float4 mean;
    {
        float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            const float4 u = *reinterpret_cast<const float4*>(&sh[g][c4 * 4]);
            t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w;
        }
        const float inv = 1.f / (float)nvalid;
        mean = make_float4(t.x * inv, t.y * inv, t.z * inv, t.w * inv);
    }
    __syncthreads();
    float4 m2 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (b0 + rg + 8 * i < B) {
            const float dx = v[i].x - mean.x, dy = v[i].y - mean.y, dz = v[i].z - mean.z, dw = v[i].w - mean.w;
            m2.x += dx * dx; m2.y += dy * dy; m2.z += dz * dz; m2.w += dw * dw;
        }
    }
    *reinterpret_cast<float4*>(&sh[rg][c4 * 4]) = m2;
    __syncthreads();
    if (rg == 0) {
        float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            const float4 u = *reinterpret_cast<const float4*>(&sh[g][c4 * 4]);
            t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w;
        }
        if (acc) {
            // one column per lane for the atomic adds below (consecutive lanes -> consecutive slots: few cache lines per
            // instruction); this half wave has read sh[0..7] above, its LDS accesses stay in program order
            *reinterpret_cast<float4*>(&sh[0][c4 * 4]) = mean;
            *reinterpret_cast<float4*>(&sh[1][c4 * 4]) = t;
        } else {
            float* p = part + (((int64_t)arm * gridDim.x + blk) * 2) * H;
            const float mm[4] = {mean.x, mean.y, mean.z, mean.w}, tt[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (c4 * 4 + j < H) { p[c4 * 4 + j] = mm[j]; p[H + c4 * 4 + j] = tt[j]; }
        }
    }
    if (acc) {
        __syncthreads();
        if ((int)threadIdx.x < H)
            acc_add_stats(acc + (int64_t)arm * ACC_SET_I64, threadIdx.x, (float)nvalid, sh[0][threadIdx.x], sh[1][threadIdx.x]);
    }
}

// =============================================================================================
// fc11 fused.  grid (ceil(B/64), NS, A), 256 threads, waves 2 x 2.
// Per 64-gene tile: z = d10 W11^T (K = H padded to a multiple of 8) -> x_rec, loss partials, dZ11
// (global + LDS) -> gd10 += dZ11 W11.  LDS: d10 tile [64][LDK], W tile [64][LDK], dZ tile [64][68].
// =============================================================================================
constexpr int F11_BM = 64, F11_BN = 64, F11_LDZ = 68;

__global__ __launch_bounds__(256) void k_fc11_fused(const float* __restrict__ d10, const float* __restrict__ params,
                                                    int64_t per_arm, int64_t w_off, int64_t b_off,
                                                    const float* __restrict__ x, int64_t x_arm_stride,
                                                    float* __restrict__ x_rec, float* __restrict__ dz11,
                                                    float* __restrict__ gd10_slab, float* __restrict__ part,
                                                    float coef, int need_grad, int A, int B, int D, int H, int NS,
                                                    int ldk, int vec_ok, int n11) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Ds = smem;                       // [64][ldk]
    float* Ws = Ds + F11_BM * ldk;          // [64][ldk]
    float* Zs = Ws + F11_BN * ldk;          // [64][68]
    float* red = Zs + F11_BM * F11_LDZ;     // [8]
    const int arm = blockIdx.z, ns = blockIdx.y, b0 = blockIdx.x * F11_BM;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, wm = wv >> 1, wn = wv & 1;
    const int KP = rup(H, 8);
    const float* W = params + (int64_t)arm * per_arm + w_off;     // [D, H]
    const float* bias = params + (int64_t)arm * per_arm + b_off;  // [D]
    const float* xa = x + (int64_t)arm * x_arm_stride;
    const float* d10a = d10 + (int64_t)arm * B * H;
    const bool hvec = ((H & 3) == 0) && vec_ok;

    // d10 tile, zero padded
    for (int idx = tid; idx < F11_BM * (ldk / 4); idx += 256) {
        const int row = idx / (ldk / 4), c4 = idx % (ldk / 4);
        *reinterpret_cast<float4*>(&Ds[row * ldk + c4 * 4]) = load_m4(d10a, H, b0 + row, c4 * 4, B, H, hvec);
    }
    const int ntile = cdiv(D, F11_BN);
    const int t0 = (int)(((int64_t)ns * ntile) / NS), t1 = (int)(((int64_t)(ns + 1) * ntile) / NS);
    f32x16 g0 = zero16(), g1 = zero16();   // gd10 tile: rows wm*32.., cols wn*64 + {0,32}
    float se = 0.f, mism = 0.f;

    for (int t = t0; t < t1; ++t) {
        const int j0 = t * F11_BN;
        __syncthreads();   // previous tile's readers of Ws / Zs are done
        for (int idx = tid; idx < F11_BN * (ldk / 4); idx += 256) {
            const int row = idx / (ldk / 4), c4 = idx % (ldk / 4);
            *reinterpret_cast<float4*>(&Ws[row * ldk + c4 * 4]) = load_m4(W, H, j0 + row, c4 * 4, D, H, hvec);
        }
        __syncthreads();
        f32x16 z = zero16();
        mma_nt(z, Ds, ldk, wm * 32, Ws, ldk, wn * 32, KP / 8);
        // epilogue on the 32x32 tile of this wave
        const int col = j0 + wn * 32 + (lane & 31);
        const float bj = (col < D) ? bias[col] : 0.f;
        float xv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = b0 + wm * 32 + acc_row(r, lane);
            xv[r] = (row < B && col < D) ? xa[(int64_t)row * D + col] : 0.f;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int lrow = wm * 32 + acc_row(r, lane);
            const int row = b0 + lrow;
            const bool ok = (row < B) && (col < D);
            const float xr = fmaxf(z[r] + bj, 0.f);
            const float e = xr - xv[r];
            float dzv = 0.f;
            if (ok) {
                se += e * e;
                mism += ((xr > 0.1f) != (xv[r] > 0.1f)) ? 1.f : 0.f;
                dzv = (xr > 0.f) ? coef * e : 0.f;
                if (x_rec) x_rec[((int64_t)arm * B + row) * D + col] = xr;
                if (need_grad) dz11[((int64_t)arm * B + row) * D + col] = dzv;
            }
            Zs[lrow * F11_LDZ + wn * 32 + (lane & 31)] = dzv;
        }
        if (need_grad) {
            __syncthreads();
            // gd10[64 x 128] += dZ[64 x 64] * W[64 x H]   (K = the 64 genes of this tile)
            mma_nn(g0, Zs, F11_LDZ, wm * 32, Ws, ldk, wn * 64, F11_BN / 8);
            mma_nn(g1, Zs, F11_LDZ, wm * 32, Ws, ldk, wn * 64 + 32, F11_BN / 8);
        }
    }
    if (need_grad) {
        float* out = gd10_slab + (((int64_t)ns * A + arm) * B) * H;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = b0 + wm * 32 + acc_row(r, lane);
            const int c0 = wn * 64 + (lane & 31);
            if (row < B) {
                if (c0 < H) out[(int64_t)row * H + c0] = g0[r];
                if (c0 + 32 < H) out[(int64_t)row * H + c0 + 32] = g1[r];
            }
        }
    }
    se = wave_sum(se);
    mism = wave_sum(mism);
    __syncthreads();
    if (lane == 0) { red[wv * 2] = se; red[wv * 2 + 1] = mism; }
    __syncthreads();
    if (tid == 0) {
        float* p = part + ((int64_t)arm * n11 + (int64_t)blockIdx.x * NS + ns) * 2;
        p[0] = red[0] + red[2] + red[4] + red[6];
        p[1] = red[1] + red[3] + red[5] + red[7];
    }
}

// =============================================================================================
// TN GEMM over the batch:  out[m][n] = sum_b P[b][m] * Q[b][n]      (m < Mv, n < Nv)
// Block tile TA x TB (TA*TB = 8192 or 16384), K tile 32 batch rows, 4 waves 2 x 2.
// grid (tiles_m * tiles_n, KS, A * ndesc).  Operand transforms are selected per descriptor.
// =============================================================================================
// (TnDesc / TnDescs: common.hpp)

template <int TA, int TB>
__global__ __launch_bounds__(256) void k_gemm_tn(const TnDescs descs, int ndesc, NoiseDev nz, int B,
                                                 int KS, int vec_ok_flags) {
    constexpr int LDA = TA + 4, LDB = TB + 4, BK = 32;
    constexpr int MT = TA / 64, NT = TB / 64;   // 32x32 tiles per wave in each direction
    __shared__ __attribute__((aligned(16))) float Ps[BK * LDA];
    __shared__ __attribute__((aligned(16))) float Qs[BK * LDB];
    const int di = blockIdx.z % ndesc, arm = blockIdx.z / ndesc, ks = blockIdx.y;
    // field-by-field into registers: read in place, the kernarg block is re-loaded after every store
    // (hipcc cannot rule out aliasing); copied as a whole with a runtime index it lands in scratch
    const TnDesc& dr = descs.d[di];
    const TnDesc d = {dr.P, dr.p_arm_stride, dr.ldp, dr.Mv, dr.Q, dr.q_arm_stride, dr.ldq, dr.Nv, dr.q_ones, dr.q_xmask,
                      dr.q_mean, dr.q_rstd, dr.out, dr.out_arm_stride, dr.out_ks_stride, dr.ldo};
    const int tiles_n = cdiv(d.Nv + d.q_ones, TB);
    const int tm = blockIdx.x / tiles_n, tn = blockIdx.x % tiles_n;
    if (tm * TA >= d.Mv) return;
    const int m0 = tm * TA, n0 = tn * TB;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, wm = wv >> 1, wn = wv & 1;
    const float* P = d.P + (int64_t)arm * d.p_arm_stride;
    const float* Q = d.Q + (int64_t)arm * d.q_arm_stride;
    const bool pvec = (vec_ok_flags & 1) && ((d.ldp & 3) == 0);
    const bool qvec = (vec_ok_flags & 2) && ((d.ldq & 3) == 0);
    const int nbt = cdiv(B, BK);
    const int bt0 = (int)(((int64_t)ks * nbt) / KS), bt1 = (int)(((int64_t)(ks + 1) * nbt) / KS);

    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = zero16();

    constexpr int PA4 = BK * TA / 4 / 256, QB4 = BK * TB / 4 / 256;   // float4 per thread
    float4 rp[PA4], rq[QB4];
    const bool pv = pvec && ((d.Mv & 3) == 0);
    const bool qv = qvec && ((d.Nv & 3) == 0);
    auto load_tiles_t = [&](int bt, auto ptag, auto qtag) __attribute__((always_inline)) {
        constexpr bool PV = decltype(ptag)::value, QV = decltype(qtag)::value;
        const int r0 = bt * BK;
#pragma unroll
        for (int i = 0; i < PA4; ++i) {
            const int idx = tid + i * 256, row = idx / (TA / 4), c4 = idx % (TA / 4);
            rp[i] = ldg4_t<PV>(P, d.ldp, r0 + row, m0 + c4 * 4, B, d.Mv);
        }
#pragma unroll
        for (int i = 0; i < QB4; ++i) {
            const int idx = tid + i * 256, row = idx / (TB / 4), c4 = idx % (TB / 4);
            rq[i] = ldg4_t<QV>(Q, d.ldq, r0 + row, n0 + c4 * 4, B, d.Nv);
        }
#pragma unroll
        for (int i = 0; i < QB4; ++i) {
            const int idx = tid + i * 256, row = idx / (TB / 4), c4 = idx % (TB / 4);
            const int gr = r0 + row, gc = n0 + c4 * 4;
            float4 v = rq[i];
            if (d.q_xmask) v = apply_xmask(v, nz, 1, arm, gr, gc, B, d.Nv, qvec);
            if (d.q_mean) {   // uniform branch; clamped, unconditional loads inside
                const float* mu = d.q_mean + (int64_t)arm * d.Nv;
                const float* rs = d.q_rstd + (int64_t)arm * d.Nv;
                const bool rok = gr < B;
                const bool o0 = rok && gc < d.Nv, o1 = rok && gc + 1 < d.Nv, o2 = rok && gc + 2 < d.Nv,
                           o3 = rok && gc + 3 < d.Nv;
                const int i0 = o0 ? gc : 0, i1 = o1 ? gc + 1 : 0, i2 = o2 ? gc + 2 : 0, i3 = o3 ? gc + 3 : 0;
                const float m0_ = mu[i0], m1_ = mu[i1], m2_ = mu[i2], m3_ = mu[i3];
                const float s0_ = rs[i0], s1_ = rs[i1], s2_ = rs[i2], s3_ = rs[i3];
                v.x = o0 ? (v.x - m0_) * s0_ : 0.f;
                v.y = o1 ? (v.y - m1_) * s1_ : 0.f;
                v.z = o2 ? (v.z - m2_) * s2_ : 0.f;
                v.w = o3 ? (v.w - m3_) * s3_ : 0.f;
            }
            if (d.q_ones) {
                const bool rok = gr < B;
                v.x = (rok && gc == d.Nv) ? 1.f : v.x;
                v.y = (rok && gc + 1 == d.Nv) ? 1.f : v.y;
                v.z = (rok && gc + 2 == d.Nv) ? 1.f : v.z;
                v.w = (rok && gc + 3 == d.Nv) ? 1.f : v.w;
            }
            rq[i] = v;
        }
    };
    auto load_tiles = [&](int bt) __attribute__((always_inline)) {
        if (pv && qv) load_tiles_t(bt, VecTag{}, VecTag{});
        else if (pv) load_tiles_t(bt, VecTag{}, ScalarTag{});
        else if (qv) load_tiles_t(bt, ScalarTag{}, VecTag{});
        else load_tiles_t(bt, ScalarTag{}, ScalarTag{});
    };
    auto store_tiles = [&]() {
#pragma unroll
        for (int i = 0; i < PA4; ++i) {
            const int idx = tid + i * 256, row = idx / (TA / 4), c4 = idx % (TA / 4);
            *reinterpret_cast<float4*>(&Ps[row * LDA + c4 * 4]) = rp[i];
        }
#pragma unroll
        for (int i = 0; i < QB4; ++i) {
            const int idx = tid + i * 256, row = idx / (TB / 4), c4 = idx % (TB / 4);
            *reinterpret_cast<float4*>(&Qs[row * LDB + c4 * 4]) = rq[i];
        }
    };

    bool live[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
            live[i][j] = (m0 + wm * (TA / 2) + i * 32 < d.Mv) && (n0 + wn * (TB / 2) + j * 32 < d.Nv + d.q_ones);
    if (bt0 < bt1) load_tiles(bt0);
    if (TA == 128 && TB == 128 && d.Mv > 96 && d.Mv <= 100 && tm == 0) {
        // 100-row outputs (the H x H small layers): a wave owns (3 MFMA row tiles + 4 leftover rows done by plain FMAs
        // on the Q fragment it already holds) x 32 columns instead of 64 x 64 of a 128-row tile with 28 padding rows
        f32x16 a3[3] = {zero16(), zero16(), zero16()};
        float lo[4] = {0.f, 0.f, 0.f, 0.f};
        const int l31 = lane & 31, hh = lane >> 5;
        for (int bt = bt0; bt < bt1; ++bt) {
            store_tiles();
            __syncthreads();
            if (bt + 1 < bt1) load_tiles(bt + 1);
            const float* pa = Ps + hh * LDA + l31;
            const float* pb = Qs + hh * LDB + wv * 32 + l31;
            const float* pl = Ps + hh * LDA + 96;      // P[b][96..99]: one address per half wave
#pragma unroll 4
            for (int s = 0; s < BK / 2; ++s) {
                const float a0 = pa[2 * s * LDA], a1 = pa[2 * s * LDA + 32], a2 = pa[2 * s * LDA + 64];
                const float q = pb[2 * s * LDB];
                const float4 p4 = *reinterpret_cast<const float4*>(pl + 2 * s * LDA);
                a3[0] = mfma32(a0, q, a3[0]);
                a3[1] = mfma32(a1, q, a3[1]);
                a3[2] = mfma32(a2, q, a3[2]);
                lo[0] = fmaf(p4.x, q, lo[0]); lo[1] = fmaf(p4.y, q, lo[1]); lo[2] = fmaf(p4.z, q, lo[2]); lo[3] = fmaf(p4.w, q, lo[3]);
            }
            __syncthreads();
        }
        float* out = d.out + (int64_t)ks * d.out_ks_stride + (int64_t)arm * d.out_arm_stride;
        const int ncols = d.Nv + d.q_ones, n = n0 + wv * 32 + l31;
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = i * 32 + acc_row(r, lane);
                if (n < ncols) out[(int64_t)m * d.ldo + n] = a3[i][r];
            }
#pragma unroll
        for (int c = 0; c < 4; ++c) lo[c] += __shfl_xor(lo[c], 32, 64);   // even / odd batch rows of the same column
        if (hh == 0 && n < ncols) {
#pragma unroll
            for (int c = 0; c < 4; ++c)
                if (96 + c < d.Mv) out[(int64_t)(96 + c) * d.ldo + n] = lo[c];
        }
        return;
    }
    for (int bt = bt0; bt < bt1; ++bt) {
        store_tiles();
        __syncthreads();
        if (bt + 1 < bt1) load_tiles(bt + 1);
        const float* pa = Ps + (lane >> 5) * LDA + wm * (TA / 2) + (lane & 31);
        const float* pb = Qs + (lane >> 5) * LDB + wn * (TB / 2) + (lane & 31);
        // 32x32 sub-tiles that lie entirely in the zero padding (small layers: N = 10, K = 10 ...) are skipped
#pragma unroll 4
        for (int s = 0; s < BK / 2; ++s) {
            float av[MT], bv[NT];
#pragma unroll
            for (int i = 0; i < MT; ++i) av[i] = pa[2 * s * LDA + i * 32];
#pragma unroll
            for (int j = 0; j < NT; ++j) bv[j] = pb[2 * s * LDB + j * 32];
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    if (live[i][j]) acc[i][j] = mfma32(av[i], bv[j], acc[i][j]);
        }
        __syncthreads();
    }
    float* out = d.out + (int64_t)ks * d.out_ks_stride + (int64_t)arm * d.out_arm_stride;
    const int ncols = d.Nv + d.q_ones;
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * (TA / 2) + i * 32 + acc_row(r, lane);
                const int n = n0 + wn * (TB / 2) + j * 32 + (lane & 31);
                if (m < d.Mv && n < ncols) out[(int64_t)m * d.ldo + n] = acc[i][j][r];
            }
}

template __global__ void k_gemm_tn<128, 64>(const TnDescs, int, NoiseDev, int, int, int);
template __global__ void k_gemm_tn<64, 128>(const TnDescs, int, NoiseDev, int, int, int);
template __global__ void k_gemm_tn<128, 128>(const TnDescs, int, NoiseDev, int, int, int);

// ---------------------------------------------------------------------------------------------
// host launchers
// ---------------------------------------------------------------------------------------------
static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

#define HIP_LAUNCH_CHECK(what)                                                        \
    do {                                                                              \
        hipError_t e_ = hipGetLastError();                                            \
        if (e_ != hipSuccess) {                                                       \
            set_error("%s: %s", what, hipGetErrorString(e_));                         \
            return MMVAE_E_LAUNCH;                                                    \
        }                                                                             \
    } while (0)

int launch_fc1_fwd(const Ctx& c, const mmvae_noise* nz, const float* params, const float* x, int64_t xs) {
    const mmvae_dims& d = c.d;
    const int use_mask = (c.h.training && c.h.x_drop > 0.f) ? 1 : 0;
    NoiseDev nd = make_noise_dev(nz, c.h);
    const int vec_ok = ((d.D & 3) == 0) && aligned16(x) && aligned16(params) && ((xs & 3) == 0) &&
                       (nd.mode != 0 || !use_mask || ((reinterpret_cast<uintptr_t>(nd.x_mask) & 3) == 0));
    const int KS = c.lay.sp.ks_fc1;
    dim3 grid(cdiv(d.B, F1_BM), KS, d.A);
    hipLaunchKernelGGL(k_fc1_fwd, grid, dim3(256), 0, c.stream, x, xs, params, c.po.per_arm, c.po.o[0], nd,
                       use_mask, c.ws + c.lay.fc1_slab, d.A, d.B, d.D, d.H, KS, vec_ok);
    HIP_LAUNCH_CHECK("k_fc1_fwd");
    return launch_fc1_epi(c, params);
}

int launch_fc1_epi(const Ctx& c, const float* params) {
    const mmvae_dims& d = c.d;
    const int KS = c.lay.sp.ks_fc1;
    const float scale = (c.h.training && c.h.x_drop > 0.f) ? 1.f / (1.f - c.h.x_drop) : 1.f;
    long long* acc = c.h.training && c.use_acc() ? reinterpret_cast<long long*>(c.ws + acc_set_off(c.lay, d.A, 0)) : nullptr;
    const dim3 grid(c.lay.nblk32, d.A);
    // the smallest slab-load chunk (4 / 8 / 16 in flight per row) that covers the split count
    if (KS <= 4)
        hipLaunchKernelGGL(k_fc1_epi<4>, grid, dim3(256), 0, c.stream, c.ws + c.lay.fc1_slab, params, c.po.per_arm, c.po.o[1], scale,
                           c.ws + c.lay.R[0], c.ws + c.lay.bn_part[0], acc, d.A, d.B, d.H, KS);
    else if (KS <= 8)
        hipLaunchKernelGGL(k_fc1_epi<8>, grid, dim3(256), 0, c.stream, c.ws + c.lay.fc1_slab, params, c.po.per_arm, c.po.o[1], scale,
                           c.ws + c.lay.R[0], c.ws + c.lay.bn_part[0], acc, d.A, d.B, d.H, KS);
    else
        hipLaunchKernelGGL(k_fc1_epi<16>, grid, dim3(256), 0, c.stream, c.ws + c.lay.fc1_slab, params, c.po.per_arm, c.po.o[1], scale,
                           c.ws + c.lay.R[0], c.ws + c.lay.bn_part[0], acc, d.A, d.B, d.H, KS);
    HIP_LAUNCH_CHECK("k_fc1_epi");
    return 0;
}

int launch_fc11_fused(const Ctx& c, const float* params, const float* x, int64_t xs, float* x_rec, int need_grad) {
    const mmvae_dims& d = c.d;
    const int ldk = rup(d.H, 8) + 4;
    const size_t shm = (size_t)(F11_BM * ldk + F11_BN * ldk + F11_BM * F11_LDZ + 8) * sizeof(float);
    const float am1 = (float)(d.A > 1 ? d.A - 1 : 1);
    const float coef = am1 / (float)d.B;
    const int NS = c.lay.sp.ns_fc11;
    dim3 grid(c.lay.nblk64, NS, d.A);
    hipError_t e = c.fwd_zeroed ? hipSuccess : hipMemsetAsync(c.ws + c.lay.fc11_part, 0, sizeof(float) * 2 * (size_t)d.A * c.lay.n11, c.stream);
    if (e != hipSuccess) { set_error("memset: %s", hipGetErrorString(e)); return MMVAE_E_LAUNCH; }
    hipLaunchKernelGGL(k_fc11_fused, grid, dim3(256), shm, c.stream, c.ws + c.lay.Dk[4], params, c.po.per_arm,
                       c.po.o[26], c.po.o[27], x, xs, x_rec, c.ws + c.lay.DZ11, c.ws + c.lay.GD10_slab,
                       c.ws + c.lay.fc11_part, coef, need_grad, d.A, d.B, d.D, d.H, NS, ldk,
                       (int)(aligned16(params)), c.lay.n11);
    HIP_LAUNCH_CHECK("k_fc11_fused");
    return 0;
}

int launch_dw_big(const Ctx& c, const mmvae_noise* nz, const float* x, int64_t xs) {
    const mmvae_dims& d = c.d;
    const Layout& L = c.lay;
    NoiseDev nd = make_noise_dev(nz, c.h);
    const int use_mask = (c.h.training && c.h.x_drop > 0.f) ? 1 : 0;
    const int KS = L.sp.ks_dw;
    // dW1[h][d] = sum_b dZ1[b][h] * x~[b][d]          -> slab [KS][A][H][D]
    TnDescs t1{}, t2{};
    t1.d[0] = TnDesc{c.ws + L.DZ[1], (int64_t)d.B * d.H, d.H, d.H, x, xs, d.D, d.D, 0, use_mask, nullptr, nullptr,
                   c.ws + L.dw1_slab, (int64_t)d.H * d.D, (int64_t)d.A * d.H * d.D, d.D};
    // [dW11 | db11][j][h] = sum_b dZ11[b][j] * [d10 | 1][b][h]   -> slab [KS][A][D][DW11_LD]
    t2.d[0] = TnDesc{c.ws + L.DZ11, (int64_t)d.B * d.D, d.D, d.D, c.ws + L.Dk[4], (int64_t)d.B * d.H, d.H, d.H, 1, 0,
                   nullptr, nullptr, c.ws + L.dw11_slab, (int64_t)d.D * DW11_LD, (int64_t)d.A * d.D * DW11_LD, DW11_LD};
    const int vx = ((d.D & 3) == 0) && aligned16(x) && ((xs & 3) == 0) &&
                   (nd.mode != 0 || !use_mask || ((reinterpret_cast<uintptr_t>(nd.x_mask) & 3) == 0));
    {
        dim3 grid(cdiv(d.H, 128) * cdiv(d.D, 64), KS, d.A);
        hipLaunchKernelGGL((k_gemm_tn<128, 64>), grid, dim3(256), 0, c.stream, t1, 1, nd, d.B, KS, 1 | (vx ? 2 : 0));
        HIP_LAUNCH_CHECK("k_gemm_tn<dW1>");
    }
    {
        dim3 grid(cdiv(d.D, 64) * cdiv(d.H + 1, 128), KS, d.A);
        hipLaunchKernelGGL((k_gemm_tn<64, 128>), grid, dim3(256), 0, c.stream, t2, 1, nd, d.B, KS, 3);
        HIP_LAUNCH_CHECK("k_gemm_tn<dW11>");
    }
    return 0;
}

// which: bit 0 = decoder layers fc6..fc10 (their dZ exist once the decoder backward chain has run), bit 1 = encoder
// side (fc2..fc5, fcc, state head, fc1.bias: after the encoder backward chain).  The slab of a layer does not depend
// on the grouping, so the two halves can be launched separately (side stream) or together.
int launch_dw_small(const Ctx& c, int which) {
    const mmvae_dims& d = c.d;
    const Layout& L = c.lay;
    const int A = d.A, B = d.B, H = d.H, Ld = d.L, C = d.C, S = d.S;
    const int KS = L.sp.ks_small;
    TnDescs ts{}, all{};
    TnDesc* hd = all.d;
    NoiseDev nd{};   // unused
    const int64_t slab_ks = (int64_t)A * N_SMALL * NP * SMALL_LD;
    auto mk = [&](int i, int64_t dz, int N, int64_t xin, int K, int64_t mean, int64_t rstd) {
        hd[i] = TnDesc{c.ws + dz, (int64_t)B * N, N, N, c.ws + xin, (int64_t)B * K, K, K, 1, 0,
                       mean >= 0 ? c.ws + mean : nullptr, rstd >= 0 ? c.ws + rstd : nullptr,
                       c.ws + L.small_slab + (int64_t)i * NP * SMALL_LD, (int64_t)N_SMALL * NP * SMALL_LD, slab_ks,
                       SMALL_LD};
    };
    mk(0, L.DZ[2], H, L.R[0], H, L.bn_mean[0], L.bn_rstd[0]);   // fc2: input BN1(R1)
    mk(1, L.DZ[3], H, L.R[1], H, L.bn_mean[1], L.bn_rstd[1]);
    mk(2, L.DZ[4], H, L.R[2], H, L.bn_mean[2], L.bn_rstd[2]);
    mk(3, L.DZ[5], Ld, L.R[3], H, L.bn_mean[3], L.bn_rstd[3]);  // fc5
    mk(4, L.GZC, C, L.XLOW, Ld, -1, -1);                         // fcc: input x_low
    mk(5, L.GMS, 2 * S, L.Y, Ld + C, -1, -1);                    // [fc_mu; fc_sigma]
    mk(6, L.DZ[6], Ld, L.ZIN, C + S, -1, -1);                    // fc6
    mk(7, L.DZ[7], H, L.Dk[0], Ld, -1, -1);                      // fc7
    mk(8, L.DZ[8], H, L.Dk[1], H, -1, -1);
    mk(9, L.DZ[9], H, L.Dk[2], H, -1, -1);
    mk(10, L.DZ[10], H, L.Dk[3], H, -1, -1);
    mk(11, L.DZ[1], H, L.R[0], 0, -1, -1);                       // fc1.bias: only the ones column
    // N <= 128 rows; K + 1 <= 256 columns (one or two column tiles), see mmvae_check_dims
    int nsel = 0, tiles = 1;
    for (int i = 0; i < N_SMALL; ++i) {
        const bool dec = i >= 6 && i <= 10;
        if (!(which & (dec ? 1 : 2))) continue;
        ts.d[nsel++] = hd[i];
        tiles = max(tiles, cdiv(hd[i].Mv, 128) * cdiv(hd[i].Nv + 1, 128));
    }
    if (nsel == 0) return 0;
    if (bf16_gemms(c)) {        // fp32x3 engine -- and the bf16 configuration, whose small layers stay fp32-grade: the same
                                // products as pairs of 128 x 128 tiles of exact slice products on the bf16 matrix pipe
        bool fits = true;
        for (int i = 0; i < nsel; ++i) fits = fits && ts.d[i].Mv <= 128 && ts.d[i].Nv + ts.d[i].q_ones <= 128 && !ts.d[i].q_xmask;
        if (fits) return launch_dw_small_x3(c, ts, nsel);
    }
    dim3 grid(tiles, KS, A * nsel);
    hipLaunchKernelGGL((k_gemm_tn<128, 128>), grid, dim3(256), 0, c.stream, ts, nsel, nd, B, KS, 3);
    HIP_LAUNCH_CHECK("k_gemm_tn<small>");
    return 0;
}

}  // namespace mmvae
