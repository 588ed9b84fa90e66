// The five D x H GEMMs of the train step on the bf16 matrix pipe (v_mfma_f32_32x32x16_bf16), in two precisions:
//
//   * mmvae_hyper.gemm_bf16 == 1 -- BASELINE.json configs[2] ("bf16, DP over 8 GPUs"): operands ROUNDED to bf16;
//   * mmvae_hyper.gemm_bf16 == 2 -- "fp32x3", the library's fp32 engine: every fp32 operand split EXACTLY into three bf16
//     slices, six slice products per product (see Eng<NP> below; k_presplit, k_x3_gemm, k_x3_small, k_x3_fc11g; DESIGN.md
//     section 14).  It also serves the small-layer gradient products and the augmenter's layers.
//
// What follows describes the shared tile engine in its one-plane (bf16) form.  Operands are rounded to bf16 (round-to-nearest-even, v_cvt_pk_bf16_f32) on their way
// into LDS, products accumulate in fp32 on v_mfma_f32_32x32x16_bf16 (2.5 PFLOP/s dense: 16 x the fp32 matrix rate), and
// everything else of the step -- BatchNorm, the softmaxes, KL / coupling terms, the reconstruction-loss epilogue, the
// slab reduction, Adam, the parameters themselves -- stays fp32 (SURVEY.md section 7 "Numerical range").  With the
// matrix pipe 16 x faster these kernels are bound by how fast a CU can pull its tiles out of L2 / HBM, so they share ONE
// simple tile engine instead of five hand-scheduled instruction streams:
//
//   C[m][n] (+ epilogue) = sum_k A(m, k) * B(n, k),  block tile 128 x 128, 256 threads = 2 x 2 waves of 64 x 64
//   (2 x 2 MFMA tiles of 32 x 32), K tile 64 = 4 MFMA steps of 16; one LDS buffer per operand, [row][k] bf16 with k
//   contiguous (row stride 144 B), so a lane's MFMA fragment (8 consecutive k of one row) is one ds_read_b128.
//
// An operand is described by how (row, k) maps to memory:
//   KMAJOR   ptr[row * ld + k]   k contiguous in memory   (x, W1, d10, W11 rows, dZ11 rows)
//   KMINOR   ptr[k * ld + row]   rows contiguous          (the batch-reduced GEMMs: dZ1, x, dZ11, d10 as [b][.];
//                                                          W11 as [j][h] for d(d10)): staged as they lie, [k][row],
//                                                          and transposed by the LDS read (ds_read_b64_tr_b16)
// plus optional decorations: the bit-packed dropout keep-mask of x (k_make_xbits) and a ones column (bias gradient).
// fc11's bias is added in fp32 by the epilogue (it is not a GEMM operand of the reference either).
//
//   fc1 forward      C[cell][h]  = x~[cell][:] . W1[h][:]            A KMAJOR+mask, B KMAJOR     -> split-K slabs
//   fc11 + loss      z[cell][j]  = d10[cell][:] . W11[j][:] (+ b11) A KMAJOR, B KMAJOR          -> dZ11, loss partials
//   d(d10)           g[cell][h]  = dZ11[cell][:] . W11[:][h]         A KMAJOR, B KMINOR          -> gene-split slabs
//   dW1              G[h][d]     = dZ1[:][h] . x~[:][d]              A KMINOR, B KMINOR+mask     -> batch-split slabs
//   [dW11 | db11]    G[j][h]     = dZ11[:][j] . [d10 | 1][:][h]      A KMINOR, B KMINOR+ones col -> batch-split slabs
// The outputs have the layouts of the fp32 kernels (gemm_fast.hip), so k_fc1_epi, the decoder chain and k_reduce are
// shared.  Reference arithmetic: mmidas/nn_model.py:263-287 (fc1, fc11), :542-546 (loss), autograd of both.
#include "common.hpp"

namespace mmvae {

#define HIP_LAUNCH_CHECK(what)                                                        \
    do {                                                                              \
        hipError_t e_ = hipGetLastError();                                            \
        if (e_ != hipSuccess) {                                                       \
            set_error("%s: %s", what, hipGetErrorString(e_));                         \
            return MMVAE_E_LAUNCH;                                                    \
        }                                                                             \
    } while (0)

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4v __attribute__((ext_vector_type(4)));


constexpr int BT = 128;          // block tile (both ways)
constexpr int KT = 64;           // K tile of the one-plane engine (the fc11 kernels below use it directly)
constexpr int LDB = 36;          // its KMAJOR LDS image [row][k]: row stride in dwords, 64 bf16 = 32 dwords + 4 (16-byte aligned rows)
constexpr int LDK = 136;         // KMINOR LDS image [k][row]: k-row stride in bf16, 128 rows + 8 (8-byte aligned)

// The engine exists in two precisions, selected by the number of bf16 PLANES an operand has in LDS:
//   NP = 1  bf16 operands (BASELINE configs[2]): K tile 64, two K tiles in LDS.
//   NP = 3  fp32 operands, split exactly into three bf16 slices on their way into LDS, x = x1 + x2 + x3 with
//           x1 = bf16(x), x2 = bf16(x - x1), x3 = bf16(x - x1 - x2) (8 + 8 + 8 significand bits; both differences are exact
//           in fp32, and x3 is exact).  The product a b is formed from six of the nine slice products, a1 b1 + (a1 b2 + a2 b1)
//           + (a1 b3 + a2 b2 + a3 b1); each is exact in the fp32 accumulator's input (8 x 8 bits), the dropped ones are
//           <= 2^-26 |a b|, below the fp32 rounding of the accumulation itself.  Six v_mfma_f32_32x32x16_bf16 do the work
//           of eight v_mfma_f32_32x32x2_f32 in 6/64 of their time (the bf16 matrix rate is 16 x the fp32 one), so the
//           fp32 configuration's large GEMMs leave the matrix pipe's roof and become memory-bound like the bf16 ones.
//           K tile 32, one K tile in LDS (3 planes x 2 operands x 10 KB = 60 KB: two workgroups per CU).
template <int NP>
struct Eng {
    static constexpr int KT = NP == 1 ? 64 : 32;
    static constexpr int NQ = KT / 8;            // 16-byte pieces per thread per operand and K tile
    static constexpr int LDB = KT / 2 + 4;       // KMAJOR image row stride in dwords (36 / 20: conflict-free ds_read_b128)
    static constexpr int PLANE = BT * LDB;       // dwords of one plane's image (the KMINOR image, KT x LDK bf16, fits in it)
    // KMINOR image [k][row]: k-row pitch in bf16.  A transposing read (ds_read_b64_tr_b16) takes, per 16 lanes, 4 k rows x
    // 16 row-columns (8 bytes per lane), 32 lanes per LDS cycle: with a pitch of 68 dwords (136 bf16) k rows q and q + 1
    // land 4 banks apart and collide with the neighbouring lanes' 8-byte reads (PMC: SQ_LDS_BANK_CONFLICT = 31 % of the
    // busy cycles of dW1 / dW11); 80 dwords (160 bf16) puts them 16 banks apart -- the four k rows of both 16-lane
    // groups tile the 64 banks exactly -- and 32 x 160 bf16 is exactly one plane.  The one-plane engine (K tile 64)
    // has no room for that pitch and keeps 136.
    static constexpr int LDK = NP == 1 ? 136 : 160;
    static constexpr int NBUF = NP == 1 ? 2 : 1;
};

__device__ __forceinline__ unsigned pack_bf16(float a, float b) {
    bf16x2 v;
    v[0] = (__bf16)a;
    v[1] = (__bf16)b;
    return __builtin_bit_cast(unsigned, v);
}
// (cvt_pk_bf16 and split3 -- the three bf16 slices of two fp32 values -- live in common.hpp: the chain kernels write
// slice planes too)

struct Operand {
    const float* ptr;        // this arm's matrix
    int64_t ld;              // leading dimension (floats)
    int rows, K;             // rows (m or n) that exist in memory and k extent
    int kminor;              // 0: ptr[row * ld + k]; 1: ptr[k * ld + row]
    const uint32_t* bits;    // optional keep-mask of x: bit (col & 31) of bits[cell * wpr + (col >> 5)], cell / col = (row, k) or (k, row)
    int wpr;
    int ones_row;            // KMINOR: row index that reads as 1.0 for every k < K (bias gradient), or -1
    // fp32x3 engine only: the operand's three bf16 slice planes, prepared once per step by k_presplit for the SMALL operand
    // of each large GEMM (W1, W11, dZ1, [d10 | 1]; every block tile of the GEMM would otherwise split the same values
    // again): [3][R][C] bf16 in the orientation of the fp32 matrix, zero-padded to whole tiles
    const unsigned short* pl;
    int64_t pl_plane, pl_arm;    // elements per plane / per arm
    int pl_ld;                   // row pitch (elements)
    // row indirection (the batch is never materialised: mmvae_train_step_rows): logical row / k index i of the operand lies
    // at ptr[rowmap[i] .. + ld) instead of ptr[i * ld ..); rowmap: uint32 [map_n] element offsets, nrec: floats addressable
    // from ptr (< 2^30).  The keep-mask words stay indexed by the logical index.
    const unsigned* rowmap;
    int64_t nrec;
    int map_n;
    // bf16 storage (the bf16 configuration on a bf16 copy of x / a bf16 dZ11, mmvae_train_step_rows(data_bf16)): ptr addresses
    // 2-byte elements (same ld, in elements); read through oct_load / oct_store below.  rows, K, ld: multiples of 8.
    int src16;
};

// One K tile of an operand in flight: rows [r0, r0 + 128) x k [k0, k0 + KT).  `load` only REQUESTS the data (NQ
// float4 and their mask words per thread; nothing touches the values, so the requests of both operands issue back to
// back and land while the MFMAs of the previous tile run); `store` masks, rounds to bf16 (or splits into the three
// slices) and writes the LDS image(s) ([128][LDB] dwords, bf16 pairs along k, zero outside the operand).
template <bool BITS, int NQ = 8>
struct TileRegsT {
    float4 v[NQ];
    uint32_t wd[BITS ? NQ : 1];   // keep-mask words (only operands that carry a mask hold them)
};

// Loads go through buffer instructions: the operand's base sits in an SGPR descriptor and a piece is addressed by ONE
// 32-bit per-lane byte offset computed where it is used (two integer ops), so no 64-bit pointers stay live across the
// pipeline (sixteen of them per thread cost 32 VGPRs and pushed the kernel into scratch).  Offsets are clamped into the
// matrix; what must read as zero is selected at the LDS store.  Every operand has ld % 4 == 0 and < 4 GB per arm.
struct OperandDev {
    __amdgpu_buffer_rsrc_t rs, rb;   // matrix, mask words
    __amdgpu_buffer_rsrc_t rm;       // row map (row indirection; see Operand::rowmap)
    const unsigned* mp;              // the same map as a pointer: K-minor operands read it with scalar loads (map_request)
    int ld, rows, K, wpr, ones_row;
    bool bits;
    float4 bn_sub, bn_mul;           // BN = true (k_x3_small): this thread's four rows read (v - bn_sub) * bn_mul
};
template <bool KMINOR>
__device__ __forceinline__ OperandDev make_operand_dev(const Operand& o) {
    OperandDev d;
    const int64_t n = KMINOR ? (int64_t)o.K * o.ld : (int64_t)o.rows * o.ld;
    d.rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(o.ptr), 0, (int)((o.rowmap ? o.nrec : n) * (o.src16 ? 2 : 4)), 0x00020000);
    d.rm = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned*>(o.rowmap ? o.rowmap : reinterpret_cast<const unsigned*>(o.ptr)), 0,
                                             (int)((o.rowmap ? o.map_n : 1) * 4), 0x00020000);
    d.mp = o.rowmap;
    d.bits = o.bits != nullptr;
    const int64_t nb = (KMINOR ? (int64_t)o.K : (int64_t)o.rows) * o.wpr;
    d.rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(d.bits ? o.bits : reinterpret_cast<const uint32_t*>(o.ptr)), 0,
                                             (int)((d.bits ? nb : 1) * 4), 0x00020000);
    d.ld = (int)o.ld; d.rows = o.rows; d.K = o.K; d.wpr = o.wpr; d.ones_row = o.ones_row;
    return d;
}

// piece q of a tile (q < NQ).  KMAJOR: thread = row tid >> 3 (+ 32 per row pass p), 16-byte k quad tid & 7 (+ 8 h for the
// second half of a 64-wide tile): K tile 64 -> p = q >> 1, h = q & 1; K tile 32 -> p = q, h = 0.  KMINOR: rows 4 (tid & 31)
// .. + 3 of k = 2 (tid >> 5 + 8 p) + h with p = q >> 1, h = q & 1 (K tile 32: p < 2).
template <bool KMINOR, int NP>
__device__ __forceinline__ void piece_ph(int q, int& p, int& h) {
    if (!KMINOR && NP != 1) { p = q; h = 0; }
    else { p = q >> 1; h = q & 1; }
}
template <bool KMINOR, bool BITS = true, int NP = 1>
__device__ __forceinline__ void quad_load(TileRegsT<BITS, Eng<NP>::NQ>& t, const OperandDev& o, int r0, int k0, int kend, int q) {
    const int tid = threadIdx.x & 255;     // (the ping-pong kernel runs two 256-thread groups per workgroup)
    int p, h;
    piece_ph<KMINOR, NP>(q, p, h);
    int off, woff;
    if (!KMINOR) {
        const int row = r0 + (tid >> 3) + 32 * p;
        const int rc = min(row, o.rows - 1);
        const int k = k0 + ((tid & 7) + 8 * h) * 4;
        const bool ok = row < o.rows && k + 3 < kend && k + 3 < o.K;     // K % 4 == 0, k0 % 4 == 0
        off = rc * o.ld + (ok ? k : 0);
        woff = rc * o.wpr + ((ok ? k : 0) >> 5);
    } else {
        // row % 4 == 0 and ld % 4 == 0: the four floats stay inside the memory row whenever row < ld
        const int k = k0 + 2 * ((tid >> 5) + 8 * p) + h;
        const int kc = min(k, o.K - 1);
        const int row = r0 + (tid & 31) * 4;
        off = kc * o.ld + (row < o.ld ? row : 0);
        woff = kc * o.wpr + (min(row, o.rows - 1) >> 5);
    }
    t.v[q] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(o.rs, off * 4, 0, 0));
    if constexpr (BITS) t.wd[q] = o.bits ? (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(o.rb, woff * 4, 0, 0) : 0xFFFFFFFFu;
}

// The same request through a row map (Operand::rowmap).  ro[q]: the element offset of piece q's memory row, i.e. of the logical
// row r0 + (tid >> 3) + 32 p (KMAJOR: the same for every K tile, requested once by ro_init) or of k (KMINOR: a new set per K
// tile, from map_request / map_offsets below).
template <bool KMINOR, int NP>
__device__ __forceinline__ void quad_load_idx(TileRegsT<true, Eng<NP>::NQ>& t, const OperandDev& o, int r0, int k0, int kend, int q,
                                              const unsigned (&ro)[Eng<NP>::NQ]) {
    const int tid = threadIdx.x & 255;
    int p, h;
    piece_ph<KMINOR, NP>(q, p, h);
    int off, woff;
    if (!KMINOR) {
        const int row = r0 + (tid >> 3) + 32 * p;
        const int rc = min(row, o.rows - 1);
        const int k = k0 + ((tid & 7) + 8 * h) * 4;
        const bool ok = row < o.rows && k + 3 < kend && k + 3 < o.K;
        off = (int)ro[q] + (ok ? k : 0);
        woff = rc * o.wpr + ((ok ? k : 0) >> 5);
    } else {
        const int k = k0 + 2 * ((tid >> 5) + 8 * p) + h;
        const int kc = min(k, o.K - 1);
        const int row = r0 + (tid & 31) * 4;
        off = (int)ro[q] + (row < o.ld ? row : 0);
        woff = kc * o.wpr + (min(row, o.rows - 1) >> 5);
    }
    t.v[q] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(o.rs, off * 4, 0, 0));
    t.wd[q] = o.bits ? (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(o.rb, woff * 4, 0, 0) : 0xFFFFFFFFu;
}
// K-minor operand through its row map: piece q of a thread lies in memory row k = k0 + 2 ((tid >> 5) + 8 p) + h with p = q >> 1,
// h = q & 1, i.e. k0 + 4 wave + 16 p + {0, 1} for the lower half of a wave and + {2, 3} for the upper one: the map entries a
// wave needs for one K tile are KT / 16 aligned groups of four consecutive words, the same for all its lanes.  They are
// requested with SCALAR loads (s_load_dwordx4 through the constant cache) one K tile ahead and turned into the per-lane
// offsets ro[] with one select each -- nothing of the map goes through the vector memory pipe, which is what bounds the
// staging phase of these kernels (as vector loads -- one dword request per piece -- the map cost dW1 14 us of 77).  The map is
// padded with MAP_PAD valid entries behind its last one (Layout::rowmap): requests run up to two K tiles past the K range.
typedef unsigned mapw4 __attribute__((ext_vector_type(4)));
typedef const mapw4 __attribute__((address_space(4)))* map4_ptr;
template <int NP>
struct MapRegs { mapw4 s[Eng<NP>::KT / 16]; };
template <int NP>
__device__ __forceinline__ void map_request(MapRegs<NP>& mr, const OperandDev& o, int k0) {
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x & 255) >> 6);
    const unsigned* base = o.mp + __builtin_amdgcn_readfirstlane(k0) + 4 * wv;
#pragma unroll
    for (int p = 0; p < Eng<NP>::KT / 16; ++p) mr.s[p] = *(map4_ptr)(const void*)(base + 16 * p);
}
template <int NP>
__device__ __forceinline__ void map_offsets(unsigned (&ro)[Eng<NP>::NQ], const MapRegs<NP>& mr) {
    const bool hi = (threadIdx.x & 32) != 0;
#pragma unroll
    for (int q = 0; q < Eng<NP>::NQ; ++q) {
        const int p = q >> 1, h = q & 1;
        ro[q] = hi ? mr.s[p][2 + h] : mr.s[p][h];
    }
}
template <int NP>     // K-major operand: the offsets of this thread's rows, once per block tile
__device__ __forceinline__ void ro_init(unsigned (&ro)[Eng<NP>::NQ], const OperandDev& o, int r0) {
    const int tid = threadIdx.x & 255;
#pragma unroll
    for (int q = 0; q < Eng<NP>::NQ; ++q) {
        int p, h;
        piece_ph<false, NP>(q, p, h);
        ro[q] = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(o.rm, min(r0 + (tid >> 3) + 32 * p, o.rows - 1) * 4, 0, 0);
    }
}
// the indexed operand's offsets for its first K tile (and, K-minor, the request for the second one's)
template <bool KMINOR, int NP, int KSTEP>
__device__ __forceinline__ void idx_begin(unsigned (&ro)[Eng<NP>::NQ], MapRegs<NP>& mr, const OperandDev& o, int r0, int kfirst) {
    if constexpr (KMINOR) {
        map_request<NP>(mr, o, kfirst);
        map_offsets<NP>(ro, mr);
        map_request<NP>(mr, o, kfirst + KSTEP);
    } else ro_init<NP>(ro, o, r0);
}
// ... and for the K tile at kld, at the head of the staging phase that requests it
template <bool KMINOR, int NP, int KSTEP>
__device__ __forceinline__ void idx_next(unsigned (&ro)[Eng<NP>::NQ], MapRegs<NP>& mr, const OperandDev& o, int kld) {
    if constexpr (KMINOR) {
        map_offsets<NP>(ro, mr);
        map_request<NP>(mr, o, kld + KSTEP);
    }
}
template <bool KMINOR, bool BITS = true, int NP = 1, bool BN = false>
__device__ __forceinline__ void quad_store(unsigned* __restrict__ T, const TileRegsT<BITS, Eng<NP>::NQ>& t, const OperandDev& o, int r0,
                                           int k0, int kend, int q) {
    constexpr int LDBv = Eng<NP>::LDB, PLANE = Eng<NP>::PLANE;
    const int tid = threadIdx.x & 255;
    int p, h;
    piece_ph<KMINOR, NP>(q, p, h);
    float x[4];
    int idx;        // dword index of the 8-byte LDS store inside a plane
    if (!KMINOR) {
        const int kq = tid & 7, rr = tid >> 3;
        const int row = r0 + rr + 32 * p;
        const int k = k0 + (kq + 8 * h) * 4;
        const bool ok = row < o.rows && k + 3 < kend && k + 3 < o.K;
        // keep bit e of nib -> an all-ones / all-zeros word (v_bfe_i32) ANDed onto the value: two VALU instructions per
        // element and no compare -> SGPR -> select chains
        const int nib = ok ? (BITS ? (int)(t.wd[q] >> (k & 31)) : 0xF) : 0;
        const float4 v = t.v[q];
        const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) x[e] = __uint_as_float(__float_as_uint(vv[e]) & (unsigned)__builtin_amdgcn_sbfe(nib, e, 1));
        idx = (rr + 32 * p) * LDBv + (kq + 8 * h) * 2;
    } else {
        // natural layout [k][row] (rows contiguous, Eng<NP>::LDK bf16 per k): one 8-byte store per (k, four rows); the MFMA
        // fragments come out of it through the transposing LDS read (ds_read_b64_tr_b16, see frag8)
        const int r4 = (tid & 31) * 4, kp = tid >> 5;
        const int kl = 2 * (kp + 8 * p) + h, k = k0 + kl;
        const bool kok = k < kend && k < o.K;
        const int nrow = min(max(o.rows - (r0 + r4), 0), 4);          // rows of this piece that exist
        const int nib = kok ? ((BITS ? (int)(t.wd[q] >> ((r0 + r4) & 31)) : 0xF) & ((1 << nrow) - 1)) : 0;
        const float4 v = t.v[q];
        const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) x[e] = __uint_as_float(__float_as_uint(vv[e]) & (unsigned)__builtin_amdgcn_sbfe(nib, e, 1));
        if constexpr (BN) {     // BatchNorm-normalised input, as k_gemm_tn: (v - mean) * rstd; rows / k that do not exist stay 0
            const float sb[4] = {o.bn_sub.x, o.bn_sub.y, o.bn_sub.z, o.bn_sub.w}, ml[4] = {o.bn_mul.x, o.bn_mul.y, o.bn_mul.z, o.bn_mul.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) x[e] = ((nib >> e) & 1) ? (x[e] - sb[e]) * ml[e] : 0.f;
        }
        if (o.ones_row >= 0) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (kok && r0 + r4 + e == o.ones_row) x[e] = 1.f;
        }
        idx = (kl * Eng<NP>::LDK + r4) >> 1;     // LDK and r4 are multiples of 4 bf16
    }
    if constexpr (NP == 1) {
        uint2 w;
        w.x = pack_bf16(x[0], x[1]);
        w.y = pack_bf16(x[2], x[3]);
        *reinterpret_cast<uint2*>(&T[idx]) = w;
    } else {
        unsigned w0[3], w1[3];
        split3(x[0], x[1], w0);
        split3(x[2], x[3], w1);
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) *reinterpret_cast<uint2*>(&T[pl * PLANE + idx]) = make_uint2(w0[pl], w1[pl]);
    }
}

// ---- operands kept as bf16 in HBM (Operand::src16; one-plane engine, K tile 64) ----------------------------------------------
// A piece is 16 bytes = EIGHT elements, four pieces per thread and K tile (an fp32 operand: eight pieces of four): K-major
// piece p = row (tid >> 3) + 32 p, k = k0 + 8 (tid & 7) .. + 7; K-minor piece p = memory row k = k0 + (tid >> 4) + 16 p, rows
// 8 (tid & 15) .. + 7.  The values are what quad_store would have produced from the fp32 original (round to nearest even, at
// the producer instead of here), so they go to the LDS image as they are -- one 16-byte store, no conversion -- with the keep
// bits ANDed on as half-word masks.  IDX: through the row map; ro[p] is the element offset of piece p's memory row.
struct OctRegs { u32x4v v[4]; uint32_t wd[4]; };
template <bool KMINOR, bool IDX>
__device__ __forceinline__ void oct_load(OctRegs& t, const OperandDev& o, int r0, int k0, int kend, int p, const unsigned (&ro)[8]) {
    const int tid = threadIdx.x & 255;
    int off, woff;
    if (!KMINOR) {
        const int row = r0 + (tid >> 3) + 32 * p;
        const int rc = min(row, o.rows - 1);
        const int k = k0 + 8 * (tid & 7);
        const bool ok = row < o.rows && k + 7 < kend && k + 7 < o.K;         // K % 8 == 0, k0 % 8 == 0
        off = (IDX ? (int)ro[p] : rc * o.ld) + (ok ? k : 0);
        woff = rc * o.wpr + ((ok ? k : 0) >> 5);
    } else {
        const int k = k0 + (tid >> 4) + 16 * p;
        const int kc = min(k, o.K - 1);
        const int row = r0 + 8 * (tid & 15);                                 // ld % 8 == 0: the eight stay inside the memory row
        off = (IDX ? (int)ro[p] : kc * o.ld) + (row < o.ld ? row : 0);
        woff = kc * o.wpr + (min(row, o.rows - 1) >> 5);
    }
    t.v[p] = __builtin_bit_cast(u32x4v, __builtin_amdgcn_raw_buffer_load_b128(o.rs, off * 2, 0, 0));
    t.wd[p] = o.bits ? (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(o.rb, woff * 4, 0, 0) : 0xFFFFFFFFu;
}
template <bool KMINOR>
__device__ __forceinline__ void oct_store(unsigned* __restrict__ T, const OctRegs& t, const OperandDev& o, int r0, int k0, int kend, int p) {
    const int tid = threadIdx.x & 255;
    int byte, idx;      // keep bits of the eight elements; dword index of the 16-byte LDS store
    if (!KMINOR) {
        const int rr = (tid >> 3) + 32 * p, row = r0 + rr;
        const int k = k0 + 8 * (tid & 7);
        const bool ok = row < o.rows && k + 7 < kend && k + 7 < o.K;
        byte = ok ? (int)((t.wd[p] >> (k & 31)) & 0xFFu) : 0;
        idx = rr * LDB + 4 * (tid & 7);
    } else {
        const int kl = (tid >> 4) + 16 * p, k = k0 + kl;
        const int r8 = 8 * (tid & 15);
        const bool kok = k < kend && k < o.K;
        const int nrow = min(max(o.rows - (r0 + r8), 0), 8);
        byte = kok ? (int)((t.wd[p] >> ((r0 + r8) & 31)) & ((1u << nrow) - 1u)) : 0;
        idx = (kl * LDK + r8) >> 1;
    }
    u32x4v w;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const unsigned m = ((unsigned)__builtin_amdgcn_sbfe(byte, 2 * j, 1) & 0xFFFFu) | ((unsigned)__builtin_amdgcn_sbfe(byte, 2 * j + 1, 1) & 0xFFFF0000u);
        w[j] = t.v[p][j] & m;
    }
    *reinterpret_cast<u32x4v*>(&T[idx]) = w;
}
// row-map offsets of the four pieces: K-major once per block tile; K-minor per K tile from the scalar map registers (the
// memory rows of a wave's pieces are k0 + 4 wave + 16 p + quarter-wave: the same groups of four words as map_request reads)
__device__ __forceinline__ void oct_ro_init(unsigned (&ro)[8], const OperandDev& o, int r0) {
    const int tid = threadIdx.x & 255;
#pragma unroll
    for (int p = 0; p < 4; ++p) ro[p] = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(o.rm, min(r0 + (tid >> 3) + 32 * p, o.rows - 1) * 4, 0, 0);
}
__device__ __forceinline__ void oct_map_offsets(unsigned (&ro)[8], const MapRegs<1>& mr) {
    const int qq = (threadIdx.x >> 4) & 3;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const unsigned lo = (qq & 1) ? mr.s[p][1] : mr.s[p][0], hi = (qq & 1) ? mr.s[p][3] : mr.s[p][2];
        ro[p] = (qq & 2) ? hi : lo;
    }
}

template <bool KMINOR, bool BITS = true, int NP = 1>
__device__ __forceinline__ void tile_load(TileRegsT<BITS, Eng<NP>::NQ>& t, const OperandDev& o, int r0, int k0, int kend) {
#pragma unroll
    for (int q = 0; q < Eng<NP>::NQ; ++q) quad_load<KMINOR, BITS, NP>(t, o, r0, k0, kend, q);
}
template <bool KMINOR, bool BITS = true, int NP = 1>
__device__ __forceinline__ void tile_store(unsigned* __restrict__ T, const TileRegsT<BITS, Eng<NP>::NQ>& t, const OperandDev& o, int r0,
                                           int k0, int kend) {
#pragma unroll
    for (int q = 0; q < Eng<NP>::NQ; ++q) quad_store<KMINOR, BITS, NP>(T, t, o, r0, k0, kend, q);
}

// MFMA fragment of a 32-row tile (rows rb .. rb + 31 of the block tile), K step s (16 k's): lane l holds row l % 32,
// k = 16 s + 8 (l / 32) .. + 7.  KMAJOR image: one ds_read_b128.  KMINOR image ([k][row]): two transposing reads -- per
// group of 16 lanes the hardware reads a 4 (k) x 16 (rows) block and hands lane i of the group column i, i.e. four
// consecutive k of row i; lane 4 q + p supplies the address of block row q, columns 4 p .. 4 p + 3.
typedef short s16x4 __attribute__((ext_vector_type(4)));
template <bool MINOR, int LDBv = LDB, int LDKv = LDK>
__device__ __forceinline__ bf16x8 frag8(const unsigned* T, int rb, int s, int lane) {
    if (!MINOR) {
        return __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4v*>(T + (rb + (lane & 31)) * LDBv + 4 * (lane >> 5) + 8 * s));
    } else {
        const unsigned short* Tk = reinterpret_cast<const unsigned short*>(T);
        const int grp = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
        const unsigned short* a = Tk + (16 * s + 8 * (grp >> 1) + q) * LDKv + rb + 16 * (grp & 1) + 4 * p;
        typedef s16x4 __attribute__((address_space(3))) * lds_ptr;
        const s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)a);
        const s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(a + 4 * LDKv));
        typedef short s16x8 __attribute__((ext_vector_type(8)));
        s16x8 r;
        r[0] = v0[0]; r[1] = v0[1]; r[2] = v0[2]; r[3] = v0[3];
        r[4] = v1[0]; r[5] = v1[1]; r[6] = v1[2]; r[7] = v1[3];
        return __builtin_bit_cast(bf16x8, r);
    }
}

// A K tile of an operand that exists as slice planes: 3 planes x 128 x 32 bf16 = 1536 sixteen-byte pieces, six per thread
// of a 256-thread group; a piece is copied global -> register -> LDS untouched (no bounds, no arithmetic).
struct PlaneRegs { u32x4v v[6]; };
struct PlaneDev { __amdgpu_buffer_rsrc_t rs; int plane, ld; };
__device__ __forceinline__ PlaneDev make_plane_dev(const Operand& o, int arm) {
    PlaneDev d;
    d.rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(o.pl + (int64_t)arm * o.pl_arm), 0, (int)(3 * o.pl_plane * 2), 0x00020000);
    d.plane = (int)o.pl_plane; d.ld = o.pl_ld;
    return d;
}
template <bool KMINOR>
__device__ __forceinline__ void plane_load(PlaneRegs& t, const PlaneDev& o, int r0, int k0, int j) {
    const int tid = threadIdx.x & 255, pl = j >> 1, id = tid + 256 * (j & 1);
    int off;   // elements
    if (!KMINOR) off = pl * o.plane + (r0 + (id >> 2)) * o.ld + k0 + 8 * (id & 3);      // row id >> 2, k octet id & 3
    else off = pl * o.plane + (k0 + (id >> 4)) * o.ld + r0 + 8 * (id & 15);              // k id >> 4, row octet id & 15
    t.v[j] = __builtin_bit_cast(u32x4v, __builtin_amdgcn_raw_buffer_load_b128(o.rs, off * 2, 0, 0));
}
template <bool KMINOR>
__device__ __forceinline__ void plane_store(unsigned* __restrict__ T, const PlaneRegs& t, int j) {
    const int tid = threadIdx.x & 255, pl = j >> 1, id = tid + 256 * (j & 1);
    int idx;   // dwords
    if (!KMINOR) idx = pl * Eng<3>::PLANE + (id >> 2) * Eng<3>::LDB + 4 * (id & 3);
    else idx = pl * Eng<3>::PLANE + (((id >> 4) * Eng<3>::LDK + 8 * (id & 15)) >> 1);
    *reinterpret_cast<u32x4v*>(&T[idx]) = t.v[j];
}

// fp32 [R][C] (row pitch ld) -> three bf16 slice planes [3][Rp][Cp], zeros outside, column `ones_col` reads 1.0 on the
// rows that exist (the bias-gradient column of [d10 | 1]).  One thread: eight consecutive columns of one row.
struct SplitJob {
    const float* src; int64_t ld, src_arm;
    int R, C, Rp, Cp, ones_col;
    unsigned short* dst; int64_t dst_arm;
    const float* col_src; int64_t col_arm;   // column `ones_col` reads col_src[row] instead of 1.0 (the bias column of [W11 | b11])
    int tr;                                  // 1: the planes hold the TRANSPOSE -- element (r, c) = src[c * ld + r]; R, C are the planes' extents
};
// The head of a training step as one launch: besides the split jobs, the first xb.blocks blocks of every z-slice make the
// dropout keep-mask (k_make_xbits' work) and the whole grid zeroes the step's loss partial slots and forward accumulator
// sets -- one launch boundary and one tail less on the critical path.
struct XbitsJob {
    NoiseDev nz; int A, B, D, wpr; uint32_t* bits; int blocks; float* zero_p; int zero_n4;
    // mmvae_train_step_rows: the row map of the batch, map[b] = rows[b] * ld (element offset of cell b's row in the resident
    // matrix; indices outside [0, n_rows) are clamped as mmvae_gather_rows does) -- null: the batch is materialised
    const int64_t* rows; unsigned* map; int64_t rows_ld, n_rows;
};
struct SplitJobs { SplitJob j[24]; int first[25]; int n; XbitsJob xb; };   // job i owns blocks [first[i], first[i + 1]) of grid.x (behind xb.blocks)
__global__ __launch_bounds__(256) void k_presplit(const SplitJobs js) {
    if (js.xb.zero_n4 > 0) grid_zero(js.xb.zero_p, js.xb.zero_n4);
    if (js.xb.rows && blockIdx.z == 0) {
        for (int i = blockIdx.x * 256 + threadIdx.x; i < js.xb.B + MAP_PAD; i += gridDim.x * 256) {   // the pad reads row 0
            const int64_t r = i < js.xb.B ? js.xb.rows[i] : 0;
            js.xb.map[i] = (unsigned)((r < 0 ? 0 : (r >= js.xb.n_rows ? js.xb.n_rows - 1 : r)) * js.xb.rows_ld);
        }
    }
    if ((int)blockIdx.x < js.xb.blocks) {
        make_xbits_range(js.xb.nz, js.xb.A, js.xb.B, js.xb.D, js.xb.wpr, js.xb.bits,
                         ((int64_t)blockIdx.z * js.xb.blocks + blockIdx.x) * 256 + threadIdx.x, (int64_t)gridDim.z * js.xb.blocks * 256);
        return;
    }
    const int bx0 = (int)blockIdx.x - js.xb.blocks;
    int ji = 0;
    while (ji + 1 < js.n && bx0 >= js.first[ji + 1]) ++ji;
    const SplitJob& J = js.j[ji];
    const int bx = bx0 - js.first[ji], nbx = js.first[ji + 1] - js.first[ji];
    const float* src = J.src + (int64_t)blockIdx.z * J.src_arm;
    const float* col = J.col_src ? J.col_src + (int64_t)blockIdx.z * J.col_arm : nullptr;
    unsigned short* dst = J.dst + (int64_t)blockIdx.z * J.dst_arm;
    const int c8n = J.Cp >> 3;                      // eight-column pieces per row
    const int64_t plane = (int64_t)J.Rp * J.Cp;
    const bool vec = (J.ld & 3) == 0 && (J.C & 3) == 0;
    // a thread keeps its piece column and walks the rows (row pitch of the walk: 256 / c8n rows when c8n <= 256)
    const int n = J.Rp * c8n;
    for (int i = bx * 256 + threadIdx.x; i < n; i += nbx * 256) {
        const int r = i / c8n, c0 = (i - r * c8n) * 8;
        float v[8];
        if (J.tr) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (r < J.R && c0 + e < J.C) ? src[(int64_t)(c0 + e) * J.ld + r] : 0.f;
        } else if (r < J.R && vec && c0 + 8 <= J.C) {
            const float4 a = *reinterpret_cast<const float4*>(src + (int64_t)r * J.ld + c0);
            const float4 b = *reinterpret_cast<const float4*>(src + (int64_t)r * J.ld + c0 + 4);
            v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int c = c0 + e;
                v[e] = (r < J.R && c < J.C) ? src[(int64_t)r * J.ld + c] : ((r < J.R && c == J.ones_col) ? (col ? col[r] : 1.f) : 0.f);
            }
        }
        unsigned w[4][3];
#pragma unroll
        for (int e = 0; e < 4; ++e) split3(v[2 * e], v[2 * e + 1], w[e]);
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
            *reinterpret_cast<uint4*>(dst + pl * plane + (int64_t)r * J.Cp + c0) = make_uint4(w[0][pl], w[1][pl], w[2][pl], w[3][pl]);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// epilogues
// ---------------------------------------------------------------------------------------------------------------
struct SlabOut {       // C tile -> out[(ks * A + arm)][m][n], m < M, n < N
    float* out;
    int64_t ks_stride, arm_stride;
    int ld, M, N;
};
struct Fc11Out {       // z tile -> + bias, dZ11, loss partials (nn_model.py:286, :542-546 and their autograd)
    const float* bias; // [D], arm stride bias_arm
    const float* x;    // this arm's [B][D]
    float* dz;         // this arm's [B][D]
    float* x_rec;      // this arm's [B][D] or null
    float* part;       // this arm's loss partial slots [n11][2]
    float coef;
    int B, D;
    const unsigned* xmap;   // row indirection (k_x3_fc11g): cell b's row of x lies at x[xmap[b] ..), x_nrec floats addressable; null: x[b * D ..)
    int64_t x_nrec;
};

struct AffineOut {     // C tile -> out[m][n] = act(C * scale[n] + shift[n]) (the augmenter's Linear + folded BatchNorm + ReLU);
    const float *scale, *shift;   // columns N .. ld - 1 (the next layer's K padding) are written as zeros
    float* out;
    int ld, M, N, relu, affine;
};

struct GemmArgs {
    Operand a, b;
    int64_t a_arm, b_arm;        // arm strides of the operands (floats); mask bits: a_bits_arm / b_bits_arm (words)
    int64_t a_bits_arm, b_bits_arm;
    int64_t bias_arm;            // arm stride of fo.bias
    int M, N, K;
    int KS;                      // splits of the k range (grid.y) -- or of the n tiles when loop_n
    SlabOut so;
    Fc11Out fo;
    AffineOut ao;
    int64_t fo_arm, fo_x_arm;    // arm strides of dz / x_rec and of x (0: the arms share x)
    int n11;
    int A;
    long long* dbg;              // diagnostic phase counters (builds with -DX3_STAMPS only), else unused
};

// the MFMAs of one K tile: this wave's 64 x 64 of the block tile (NP = 3: six slice products per pair of fragments, the
// small ones first)
template <bool AMINOR, bool BMINOR, int NP = 1>
__device__ __forceinline__ void mfma_ktile(f32x16 (&acc)[2][2], const unsigned* As, const unsigned* Bs, int wm, int wn, int lane) {
    constexpr int LDBv = Eng<NP>::LDB, PLANE = Eng<NP>::PLANE, NS = Eng<NP>::KT / 16;
    // NP = 3: the K tile has two K steps; the fragments of the second are requested before the MFMAs of the first (a
    // ds_read that is waited for at once costs 64+ cycles -- tools/micro/x3_issue_bench.hip -- and hipcc sinks reads
    // towards their uses: the fence keeps them where they are)
    bf16x8 a[NP == 3 ? 2 : 1][2][NP], b[NP == 3 ? 2 : 1][2][NP];
    auto frags = [&](int s, int w) __attribute__((always_inline)) {
#pragma unroll
        for (int pl = 0; pl < NP; ++pl) {
            a[w][0][pl] = frag8<AMINOR, LDBv, Eng<NP>::LDK>(As + pl * PLANE, wm * 64, s, lane);
            a[w][1][pl] = frag8<AMINOR, LDBv, Eng<NP>::LDK>(As + pl * PLANE, wm * 64 + 32, s, lane);
            b[w][0][pl] = frag8<BMINOR, LDBv, Eng<NP>::LDK>(Bs + pl * PLANE, wn * 64, s, lane);
            b[w][1][pl] = frag8<BMINOR, LDBv, Eng<NP>::LDK>(Bs + pl * PLANE, wn * 64 + 32, s, lane);
        }
    };
    if constexpr (NP == 3) {
        frags(0, 0);
        frags(1, 1);
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const int w = NP == 3 ? s : 0;
        if constexpr (NP != 3) frags(s, 0);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                if constexpr (NP == 3) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[w][i][2], b[w][j][0], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[w][i][0], b[w][j][2], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[w][i][1], b[w][j][1], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[w][i][1], b[w][j][0], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[w][i][0], b[w][j][1], acc[i][j], 0, 0, 0);
                }
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[w][i][0], b[w][j][0], acc[i][j], 0, 0, 0);
            }
    }
}

// The tile epilogues: accumulator register r of tile (i, j) is row m0 + 64 wm + 32 i + acc_row(r), column
// n0 + 64 wn + 32 j + (lane & 31).
template <int EPI>
__device__ __forceinline__ void gemm_epilogue(const GemmArgs& g, const f32x16 (&acc)[2][2], int m0, int n0, int wm, int wn, int lane, int arm) {
    const int l31 = lane & 31;
    if (EPI == 1) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = n0 + 64 * wn + 32 * j + l31;
            if (col >= g.ao.ld) continue;
            const bool real = col < g.ao.N;
            const float sc = (g.ao.affine && real) ? g.ao.scale[col] : 1.f;
            const float sh = (g.ao.affine && real) ? g.ao.shift[col] : 0.f;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = m0 + 64 * wm + 32 * i + acc_row(r, lane);
                    if (row < g.ao.M) {
                        float v = acc[i][j][r] * sc + sh;
                        if (g.ao.relu) v = fmaxf(v, 0.f);
                        g.ao.out[(int64_t)row * g.ao.ld + col] = real ? v : 0.f;
                    }
                }
        }
        return;
    }
    float* out = g.so.out + (int64_t)blockIdx.y * g.so.ks_stride + (int64_t)arm * g.so.arm_stride;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = n0 + 64 * wn + 32 * j + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + 64 * wm + 32 * i + acc_row(r, lane);
                if (row < g.so.M && col < g.so.N) out[(int64_t)row * g.so.ld + col] = acc[i][j][r];
            }
        }
}

// C tile -> slab.  grid (tiles_m * tiles_n, KS, A): block (m tile, n tile) x k range ks x arm.  Tile t + 1 is written to
// the second LDS buffer while tile t is multiplied (one barrier per K tile); a piece's registers request tile t + 2 as
// soon as they have been written out for tile t + 1.
// IDX: 1 / 2 = the A / B operand through its row map (see k_x3_gemm); S16: bit 0 / bit 1 = the A / B operand is bf16 in memory
template <bool AMINOR, bool BMINOR, int EPI = 0, int IDX = 0, int S16 = 0>
__global__ __launch_bounds__(256, 2) void k_bf16_gemm(const GemmArgs g_in) {
    const GemmArgs g = g_in;
    __shared__ __attribute__((aligned(16))) unsigned As[2][BT * LDB];   // two K tiles in LDS: tile t is multiplied while tile
    __shared__ __attribute__((aligned(16))) unsigned Bs[2][BT * LDB];   // t + 1 is written and tile t + 2 is in flight
    const int arm = blockIdx.z, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, wm = wv >> 1, wn = wv & 1;
    Operand oa_h = g.a, ob_h = g.b;
    oa_h.ptr += (int64_t)arm * g.a_arm;
    ob_h.ptr += (int64_t)arm * g.b_arm;
    if (oa_h.bits) oa_h.bits += (int64_t)arm * g.a_bits_arm;
    if (ob_h.bits) ob_h.bits += (int64_t)arm * g.b_bits_arm;
    const OperandDev oa = make_operand_dev<AMINOR>(oa_h), ob = make_operand_dev<BMINOR>(ob_h);
    const int tiles_n = cdiv(g.N, BT);
    int wg = blockIdx.x;
    if (EPI == 1) {   // XCD-aware tile order: workgroup i runs on XCD i % 8; give XCD x a contiguous range of tiles
        const int nwg = gridDim.x;
        if (nwg % 8 == 0) wg = (wg & 7) * (nwg >> 3) + (wg >> 3);
    }
    const int m0 = (wg / tiles_n) * BT, n0 = (wg % tiles_n) * BT;
    const int nkt = cdiv(g.K, KT);
    const int kb = (int)(((int64_t)blockIdx.y * nkt) / g.KS) * KT;
    const int ke = min(g.K, (int)(((int64_t)(blockIdx.y + 1) * nkt) / g.KS) * KT);
    f32x16 acc[2][2] = {{zero16(), zero16()}, {zero16(), zero16()}};
    TileRegsT<true> ta, tb;
    OctRegs t16a, t16b;                // S16 (bit 0: A, bit 1: B): a bf16-source operand's pieces (ta / tb of that operand stay unused)
    unsigned ro[Eng<1>::NQ] = {};      // IDX: element offsets of the indexed operand's memory rows (see quad_load_idx)
    MapRegs<1> mr = {};
    static_assert(IDX == 0 || S16 == 0 || (S16 & IDX) != 0, "row map and bf16 storage: the indexed operand (x) is a bf16 one");
    constexpr bool A16 = (S16 & 1) != 0, B16 = (S16 & 2) != 0;
    // Software pipeline at the granularity of one 16-byte piece: a piece of tile t + 1 is rounded and written to LDS
    // and the SAME registers immediately request the piece of tile t + 2, so sixteen loads per thread are in flight
    // all the time and every load has a whole iteration (the other fifteen pieces, the MFMAs, the barrier) to land.
    auto stage = [&](unsigned* Ad, unsigned* Bd, int kst, int kld, auto load_tag) __attribute__((always_inline)) {
        constexpr bool LOAD = decltype(load_tag)::value;
        if constexpr (LOAD && IDX == 1 && AMINOR) {
            if constexpr (A16) { oct_map_offsets(ro, mr); map_request<1>(mr, oa, kld + KT); }
            else idx_next<AMINOR, 1, KT>(ro, mr, oa, kld);
        }
        if constexpr (LOAD && IDX == 2 && BMINOR) {
            if constexpr (B16) { oct_map_offsets(ro, mr); map_request<1>(mr, ob, kld + KT); }
            else idx_next<BMINOR, 1, KT>(ro, mr, ob, kld);
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            // (no run-time condition around a load: hipcc then waits for every load separately)
            if constexpr (A16) {
                if (q < 4) {
                    oct_store<AMINOR>(Ad, t16a, oa, m0, kst, ke, q);
                    if constexpr (LOAD) oct_load<AMINOR, IDX == 1>(t16a, oa, m0, kld, ke, q, ro);
                }
            } else {
                quad_store<AMINOR>(Ad, ta, oa, m0, kst, ke, q);
                if constexpr (LOAD) {
                    if constexpr (IDX == 1) quad_load_idx<AMINOR, 1>(ta, oa, m0, kld, ke, q, ro);
                    else quad_load<AMINOR>(ta, oa, m0, kld, ke, q);
                }
            }
            if constexpr (B16) {
                if (q < 4) {
                    oct_store<BMINOR>(Bd, t16b, ob, n0, kst, ke, q);
                    if constexpr (LOAD) oct_load<BMINOR, IDX == 2>(t16b, ob, n0, kld, ke, q, ro);
                }
            } else {
                quad_store<BMINOR>(Bd, tb, ob, n0, kst, ke, q);
                if constexpr (LOAD) {
                    if constexpr (IDX == 2) quad_load_idx<BMINOR, 1>(tb, ob, n0, kld, ke, q, ro);
                    else quad_load<BMINOR>(tb, ob, n0, kld, ke, q);
                }
            }
        }
    };
    // first K tile: the indexed operand's offsets, then every piece's request
    auto first = [&](auto a_tag) __attribute__((always_inline)) {
        constexpr bool ISA = decltype(a_tag)::value;
        constexpr bool MINOR = ISA ? AMINOR : BMINOR;
        constexpr bool IDXD = IDX == (ISA ? 1 : 2), IS16 = ISA ? A16 : B16;
        const OperandDev& o = ISA ? oa : ob;
        const int r0 = ISA ? m0 : n0;
        OctRegs& t16 = ISA ? t16a : t16b;
        if constexpr (IS16) {
            if constexpr (IDXD) {
                if constexpr (MINOR) { map_request<1>(mr, o, kb); oct_map_offsets(ro, mr); map_request<1>(mr, o, kb + KT); }
                else oct_ro_init(ro, o, r0);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) oct_load<MINOR, IDXD>(t16, o, r0, kb, ke, q, ro);
        } else if constexpr (IDXD) {
            idx_begin<MINOR, 1, KT>(ro, mr, o, r0, kb);
#pragma unroll
            for (int q = 0; q < Eng<1>::NQ; ++q) quad_load_idx<MINOR, 1>(ISA ? ta : tb, o, r0, kb, ke, q, ro);
        } else tile_load<MINOR>(ISA ? ta : tb, o, r0, kb, ke);
    };
    if (kb < ke) {
        first(VecTag{});
        first(ScalarTag{});
    }
    if (kb < ke) {
        if (kb + KT < ke) stage(As[0], Bs[0], kb, kb + KT, VecTag{});
        else stage(As[0], Bs[0], kb, kb, ScalarTag{});
    }
    __syncthreads();
    int cur = 0;
    for (int k0 = kb; k0 < ke; k0 += KT) {
        mfma_ktile<AMINOR, BMINOR>(acc, As[cur], Bs[cur], wm, wn, lane);
        if (k0 + 2 * KT < ke) stage(As[cur ^ 1], Bs[cur ^ 1], k0 + KT, k0 + 2 * KT, VecTag{});
        else if (k0 + KT < ke) stage(As[cur ^ 1], Bs[cur ^ 1], k0 + KT, k0 + KT, ScalarTag{});
        __syncthreads();
        cur ^= 1;
    }
    gemm_epilogue<EPI>(g, acc, m0, n0, wm, wn, lane, arm);
}

// The split (NP = 3) engine: fp32 operands as three bf16 planes each, six MFMAs per pair of fragments.  With six times the
// matrix instructions and eleven VALU instructions per pair of elements for the split, a block tile is no longer a pure
// memory stream: the matrix phase (48 MFMAs per wave and K tile of 32) and the staging phase (split + LDS writes + the
// next requests) are about equally long, and two independent workgroups per CU did not overlap them (measured: the sum of
// the phases, 93 us for fc1).  So ONE workgroup of 512 threads carries TWO block tiles in lock-step ping-pong: group 0
// (waves 0-3) multiplies its K tile t while group 1 (waves 4-7, one per SIMD beside a wave of group 0) stages its own,
// and at every workgroup barrier the roles swap -- each SIMD always has one wave on the matrix pipe and one on the
// VALU / LDS / memory pipes.  A group owns one LDS buffer set (3 planes x 2 operands x 10 KB = 60 KB; 120 KB per CU).
// grid (ceil(tiles / 2), KS, A): group g of block b owns tile 2 b + g.
// SHARE: the two tiles of a block are neighbours along m when the GEMM is one tile wide (fc1, d(d10), dW11: same n0,
// SHARE = 2) or along n when it is one tile high (dW1: same m0, SHARE = 1) -- the narrow operand's K tile is then the
// SAME for both groups, and what bounds these kernels is the bytes a CU takes in (about 13 B per cycle, measured with
// in-kernel stamps: a stage phase lasts as long as its loads take to issue).  Group 0 stages the shared operand for both
// (two LDS buffers, indexed by the K tile's parity; group 1 multiplies tile t while group 0 already stages tile t + 1), a
// quarter less traffic per pair of tiles.  LDS: two private + two shared buffer sets of 30 KB = 120 KB.
// SPL: the shared operand comes as slice planes (k_presplit) and is copied, not split again by every block.
// SHARE = 3: BOTH groups work on the SAME block tile and take alternate K tiles (group g: tiles g, g + 2, ...); group 1's
// accumulators are added to group 0's through LDS before the epilogue.  For GEMMs with fewer tiles than CUs (the augmenter's
// 500- and 1000-wide trunk layers at M = 5000: 40 to 160 tiles): twice the workgroups, half the K loop each.  With SPL the B
// operand comes from planes, copied by each group for its own K tiles.
// IDX: 1 / 2 = the A / B operand is read through its row map (Operand::rowmap: x of a batch that is never materialised)
// K tiles of the fp32 operand a group keeps in flight.  2 (-DX3_DEPTH=2) was measured and lost: the kernels sit at 234 - 242 of
// the 256 registers a thread of a 512-thread workgroup has, twenty more for the second tile spilled (2 - 40 registers) and
// stretched the schedule -- fc1 79 -> 123 us, dW1 80 -> 108 us
#ifndef X3_DEPTH
#define X3_DEPTH 1
#endif
static_assert(X3_DEPTH == 1 || X3_DEPTH == 2, "one or two K tiles in flight");
template <bool AMINOR, bool BMINOR, int EPI = 0, int SHARE = 0, bool SPL = false, int IDX = 0>
__global__ __launch_bounds__(512, 1) void k_x3_gemm(const GemmArgs g_in) {
    typedef Eng<3> E;
    constexpr int KTv = E::KT;
    const GemmArgs g = g_in;
    __shared__ __attribute__((aligned(16))) unsigned As[2][3 * E::PLANE];
    __shared__ __attribute__((aligned(16))) unsigned Bs[2][3 * E::PLANE];
    const int arm = blockIdx.z, grp = threadIdx.x >> 8, tid = threadIdx.x & 255, lane = tid & 63, wv = tid >> 6, wm = wv >> 1, wn = wv & 1;
    Operand oa_h = g.a, ob_h = g.b;
    oa_h.ptr += (int64_t)arm * g.a_arm;
    ob_h.ptr += (int64_t)arm * g.b_arm;
    if (oa_h.bits) oa_h.bits += (int64_t)arm * g.a_bits_arm;
    if (ob_h.bits) ob_h.bits += (int64_t)arm * g.b_bits_arm;
    const OperandDev oa = make_operand_dev<AMINOR>(oa_h), ob = make_operand_dev<BMINOR>(ob_h);
    // The two tiles of a block: neighbours along m for the same n tile when they share the B operand (SHARE = 2), along n
    // for the same m tile when they share A (SHARE = 1), consecutive tiles of the (m, n) list otherwise.  An odd count
    // leaves the last block's second group idle.  EPI = 1 (the augmenter's layers): XCD-aware block order -- workgroup i
    // runs on XCD i % 8; give XCD x a contiguous range of blocks (neighbouring n tiles read the same activation rows).
    const int tiles_m = cdiv(g.M, BT), tiles_n = cdiv(g.N, BT);
    int bi = blockIdx.x;
    if (EPI == 1 && (gridDim.x & 7) == 0) bi = (bi & 7) * (gridDim.x >> 3) + (bi >> 3);
    int tm, tn;
    if (SHARE == 2) { tn = bi % tiles_n; tm = 2 * (bi / tiles_n) + grp; }
    else if (SHARE == 1) { tm = bi % tiles_m; tn = 2 * (bi / tiles_m) + grp; }
    else if (SHARE == 3) { tm = bi / tiles_n; tn = bi % tiles_n; }
    else { const int wg = 2 * bi + grp; tm = wg / tiles_n; tn = wg % tiles_n; }
    const bool active = tm < tiles_m && tn < tiles_n;
    const int m0 = tm * BT, n0 = tn * BT;
    const int nkt = cdiv(g.K, KTv);
    const int kb = (int)(((int64_t)blockIdx.y * nkt) / g.KS) * KTv;
    const int ke = min(g.K, (int)(((int64_t)(blockIdx.y + 1) * nkt) / g.KS) * KTv);
    const int n_all = ke > kb ? cdiv(ke - kb, KTv) : 0;  // K tiles of this block
    // SHARE = 3: this group's share of them (tiles grp, grp + 2, ...); the phase loop runs for the larger share
    const int n = SHARE == 3 ? (n_all + 1) / 2 : n_all;
    const int n_mine = SHARE == 3 ? (n_all + 1 - grp) / 2 : n_all;
    constexpr int KSTEP = SHARE == 3 ? 2 * KTv : KTv;      // distance between consecutive K tiles of a group
    const int kfirst = SHARE == 3 ? kb + grp * KTv : kb;
    f32x16 acc[2][2] = {{zero16(), zero16()}, {zero16(), zero16()}};
    // X3_DEPTH K tiles of a group's fp32 operand are in flight (register slot = tile index % X3_DEPTH).  With one, a CU keeps
    // about 30 KB in flight and takes in 26 GB/s -- the latency-bound part of the intake curve (tools/micro/intake_bench.hip,
    // profiles/r03_intake_microbench.txt: 29 GB/s at 16 KB in flight, 46 at 32 - 64 KB from L2, 28 from HBM) --, but the second
    // tile's registers are not there (see X3_DEPTH above)
    constexpr int DEPTH = X3_DEPTH;
    // (only the fp32 operand -- the one that comes from HBM -- goes two deep: twenty registers; with the slice planes two deep
    // as well the kernel spilled 48 - 64 registers)
    TileRegsT<true, E::NQ> ta_[DEPTH], tb_[DEPTH];
    unsigned ro[E::NQ] = {};           // IDX: element offsets of the indexed operand's memory rows (see quad_load_idx)
    MapRegs<3> mr = {};
    PlaneRegs ps;
    constexpr bool APL = SPL && SHARE == 1, BPL = SPL && (SHARE == 2 || SHARE == 3);
    static_assert(IDX == 0 || SHARE != 3, "row indirection: not with the K-alternating form");
    const PlaneDev dp = make_plane_dev(SHARE == 1 ? g.a : g.b, arm);
    const bool do_a = SHARE != 1 || grp == 0, do_b = SHARE != 2 || grp == 0;   // which operands this group stages
    // one piece of each operand in turn; a piece's registers request the next tile as soon as they have been written out
    // kst: the tile whose registers (slot SLOT) are written out; kld: the tile they then request (DEPTH tiles on, LOAD); kpl:
    // the tile the plane registers request (one tile on, LOADP)
    auto stage = [&](unsigned* Ad, unsigned* Bd, int kst, int kld, int kpl, auto load_tag, auto loadp_tag, auto slot_tag) __attribute__((always_inline)) {
        constexpr bool LOAD = decltype(load_tag)::value, LOADP = decltype(loadp_tag)::value;
        constexpr int SLOT = decltype(slot_tag)::value;
        TileRegsT<true, E::NQ>& ta = ta_[SLOT];
        TileRegsT<true, E::NQ>& tb = tb_[SLOT];
        if constexpr (LOAD && IDX == 1) idx_next<AMINOR, 3, KSTEP>(ro, mr, oa, kld);
        if constexpr (LOAD && IDX == 2) idx_next<BMINOR, 3, KSTEP>(ro, mr, ob, kld);
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            if constexpr (APL) {
                if (do_a) {
                    plane_store<AMINOR>(Ad, ps, i);
                    if constexpr (LOADP) plane_load<AMINOR>(ps, dp, m0, kpl, i);
                }
            } else if (i < E::NQ) {
                if (do_a) {
                    quad_store<AMINOR, true, 3>(Ad, ta, oa, m0, kst, ke, i);
                    if constexpr (LOAD) {
                        if constexpr (IDX == 1) quad_load_idx<AMINOR, 3>(ta, oa, m0, kld, ke, i, ro);
                        else quad_load<AMINOR, true, 3>(ta, oa, m0, kld, ke, i);
                    }
                }
            }
            if constexpr (BPL) {
                if (do_b) {
                    plane_store<BMINOR>(Bd, ps, i);
                    if constexpr (LOADP) plane_load<BMINOR>(ps, dp, n0, kpl, i);
                }
            } else if (i < E::NQ) {
                if (do_b) {
                    quad_store<BMINOR, true, 3>(Bd, tb, ob, n0, kst, ke, i);
                    if constexpr (LOAD) {
                        if constexpr (IDX == 2) quad_load_idx<BMINOR, 3>(tb, ob, n0, kld, ke, i, ro);
                        else quad_load<BMINOR, true, 3>(tb, ob, n0, kld, ke, i);
                    }
                }
            }
        }
    };
    // the first DEPTH tiles' requests (tile d into slot d)
    auto first = [&](auto slot_tag, int k) __attribute__((always_inline)) {
        constexpr int SLOT = decltype(slot_tag)::value;
        TileRegsT<true, E::NQ>& ta = ta_[SLOT];
        TileRegsT<true, E::NQ>& tb = tb_[SLOT];
        if constexpr (APL) {
            if (SLOT == 0 && do_a) {
#pragma unroll
                for (int i = 0; i < 6; ++i) plane_load<AMINOR>(ps, dp, m0, k, i);
            }
        } else if (do_a) {
            if constexpr (IDX == 1) {
                if constexpr (SLOT == 0) idx_begin<AMINOR, 3, KSTEP>(ro, mr, oa, m0, k);
                else idx_next<AMINOR, 3, KSTEP>(ro, mr, oa, k);
#pragma unroll
                for (int q = 0; q < E::NQ; ++q) quad_load_idx<AMINOR, 3>(ta, oa, m0, k, ke, q, ro);
            } else tile_load<AMINOR, true, 3>(ta, oa, m0, k, ke);
        }
        if constexpr (BPL) {
            if (SLOT == 0 && do_b) {
#pragma unroll
                for (int i = 0; i < 6; ++i) plane_load<BMINOR>(ps, dp, n0, k, i);
            }
        } else if (do_b) {
            if constexpr (IDX == 2) {
                if constexpr (SLOT == 0) idx_begin<BMINOR, 3, KSTEP>(ro, mr, ob, n0, k);
                else idx_next<BMINOR, 3, KSTEP>(ro, mr, ob, k);
#pragma unroll
                for (int q = 0; q < E::NQ; ++q) quad_load_idx<BMINOR, 3>(tb, ob, n0, k, ke, q, ro);
            } else tile_load<BMINOR, true, 3>(tb, ob, n0, k, ke);
        }
    };
    if (active && n_mine > 0) {
        first(std::integral_constant<int, 0>{}, kfirst);
        if constexpr (DEPTH > 1) {
            if (n_mine > 1) first(std::integral_constant<int, DEPTH - 1>{}, kfirst + KSTEP);
        }
    }
#ifdef X3_STAMPS
    long long t_st = 0, t_mf = 0, t_bar = 0, t_ld = 0, t0 = __builtin_amdgcn_s_memtime(), t_begin = t0;
#endif
    // phase p: group g is at step q = p - g of its own sequence stage(0), mfma(0), stage(1), mfma(1), ...
    for (int p = 0; p <= 2 * n; ++p) {
        const int q = p - grp;
        if (active && q >= 0 && q < 2 * n_mine) {
            const int t = q >> 1, k0 = kfirst + t * KSTEP;
            unsigned* const Ad = As[SHARE == 1 ? (t & 1) : grp];
            unsigned* const Bd = Bs[SHARE == 2 ? (t & 1) : grp];
            if (q & 1) {
                mfma_ktile<AMINOR, BMINOR, 3>(acc, Ad, Bd, wm, wn, lane);
#ifdef X3_STAMPS
                asm volatile("" :: "v"(acc[0][0]), "v"(acc[1][1]));
                { const long long t1 = __builtin_amdgcn_s_memtime(); t_mf += t1 - t0; t0 = t1; }
#endif
            } else {
#ifdef X3_STAMPS
                __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): how long the stage waits for its operands
                { const long long t1 = __builtin_amdgcn_s_memtime(); t_ld += t1 - t0; t0 = t1; }
#endif
                // tile t + DEPTH / t + 1 exist: their requests go out with tile t's store
                const bool more = t + DEPTH < n_mine, morep = t + 1 < n_mine;
                const int kld = k0 + DEPTH * KSTEP, kpl = k0 + KSTEP;
                auto go = [&](auto slot_tag) __attribute__((always_inline)) {
                    if (more) stage(Ad, Bd, k0, kld, kpl, VecTag{}, VecTag{}, slot_tag);
                    else if (morep) stage(Ad, Bd, k0, k0, kpl, ScalarTag{}, VecTag{}, slot_tag);
                    else stage(Ad, Bd, k0, k0, k0, ScalarTag{}, ScalarTag{}, slot_tag);
                };
                if (DEPTH == 1 || !(t & 1)) go(std::integral_constant<int, 0>{});
                else go(std::integral_constant<int, DEPTH - 1>{});
#ifdef X3_STAMPS
                __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0): the LDS writes have been issued and accepted
                { const long long t1 = __builtin_amdgcn_s_memtime(); t_st += t1 - t0; t0 = t1; }
#endif
            }
        }
        __syncthreads();
#ifdef X3_STAMPS
        { const long long t1 = __builtin_amdgcn_s_memtime(); t_bar += t1 - t0; t0 = t1; }
#endif
    }
#ifdef X3_STAMPS
    if (g.dbg && lane == 0 && wv == 0 && blockIdx.x == 3 && blockIdx.y == 2 && blockIdx.z == 0) {
        g.dbg[grp * 8 + 0] = t_st; g.dbg[grp * 8 + 1] = t_mf; g.dbg[grp * 8 + 2] = t_bar; g.dbg[grp * 8 + 3] = n;
        g.dbg[grp * 8 + 4] = __builtin_amdgcn_s_memtime() - t_begin;
        g.dbg[grp * 8 + 5] = t_ld;
    }
#endif
    if constexpr (SHARE == 3) {
        // group 1's partial sums join group 0's through LDS (the operand images are dead: every wave is past the loop's last
        // barrier), one row-tile pair of accumulators at a time: 2 x 16 floats per lane = 32 KB
        float* red = reinterpret_cast<float*>(&As[0][0]);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if (grp == 1) {
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) red[(j * 16 + r) * 256 + tid] = acc[i][j][r];
            }
            __syncthreads();
            if (grp == 0) {
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][j][r] += red[(j * 16 + r) * 256 + tid];
            }
            __syncthreads();
        }
        if (active && grp == 0) gemm_epilogue<EPI>(g, acc, m0, n0, wm, wn, lane, arm);
        return;
    }
    if (active) gemm_epilogue<EPI>(g, acc, m0, n0, wm, wn, lane, arm);
}

// fc11 forward + bias + reconstruction loss + dZ11.  The product is computed TRANSPOSED, z^T = W11 d10^T: MFMA rows are
// genes and columns are cells, so a lane's four consecutive accumulator registers are four consecutive genes of one
// cell -- 16 contiguous bytes of x, dZ11 and x_rec.  grid (cell tiles, gene splits NS, A): a block keeps its 128 cells'
// d10 (both K tiles, bf16) in LDS and walks the 128-gene tiles of its gene range; the next W11 tile is requested from
// memory before the epilogue of the current one.  K = fc_dim <= 128.
__global__ __launch_bounds__(256, 2) void k_bf16_fc11(const GemmArgs g_in) {
    const GemmArgs g = g_in;
    __shared__ __attribute__((aligned(16))) unsigned Ws[2][BT * LDB];   // W11 tile: [gene][k], two K tiles
    __shared__ __attribute__((aligned(16))) unsigned Ds[2][BT * LDB];   // d10 tile: [cell][k], two K tiles
    __shared__ float red[8];
    const int arm = blockIdx.z, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, wm = wv >> 1, wn = wv & 1;
    const int l31 = lane & 31, hh = lane >> 5;
    Operand od_h = g.a, ow_h = g.b;                   // a: d10 [B][H]; b: W11 [D][H]
    od_h.ptr += (int64_t)arm * g.a_arm;
    ow_h.ptr += (int64_t)arm * g.b_arm;
    const OperandDev od = make_operand_dev<false>(od_h), ow = make_operand_dev<false>(ow_h);
    const int c0 = blockIdx.x * BT;                    // cells of this block
    const int tiles = cdiv(g.N, BT);                   // gene tiles
    const int t0 = (int)(((int64_t)blockIdx.y * tiles) / g.KS), t1 = (int)(((int64_t)(blockIdx.y + 1) * tiles) / g.KS);
    const int K = g.K;
    const float* xa = g.fo.x + (int64_t)arm * g.fo_x_arm;
    float* dza = g.fo.dz + (int64_t)arm * g.fo_arm;
    float* xra = g.fo.x_rec ? g.fo.x_rec + (int64_t)arm * g.fo_arm : nullptr;
    const float* bias = g.fo.bias + (int64_t)arm * g.bias_arm;
    const int B = g.fo.B, D = g.fo.D;
    {
        TileRegsT<false> t0r, t1r;
        tile_load<false, false>(t0r, od, c0, 0, K);
        tile_load<false, false>(t1r, od, c0, KT, K);
        tile_store<false, false>(Ds[0], t0r, od, c0, 0, K);
        tile_store<false, false>(Ds[1], t1r, od, c0, KT, K);
    }
    TileRegsT<false> w0, w1;
    if (t0 < t1) {
        tile_load<false, false>(w0, ow, t0 * BT, 0, K);
        tile_load<false, false>(w1, ow, t0 * BT, KT, K);
    }
    float se = 0.f;
    int mism = 0;
    for (int t = t0; t < t1; ++t) {
        const int j0 = t * BT;
        tile_store<false, false>(Ws[0], w0, ow, j0, 0, K);
        tile_store<false, false>(Ws[1], w1, ow, j0, KT, K);
        __syncthreads();
        if (t + 1 < t1) {
            tile_load<false, false>(w0, ow, j0 + BT, 0, K);
            tile_load<false, false>(w1, ow, j0 + BT, KT, K);
        }
        // the epilogue walks the four 32 x 32 pieces (i, j) of this wave's tile with the NEXT piece's x and bias in flight:
        // piece 0 is requested before the MFMAs, piece p + 1 before piece p is processed (140 -> 131 us; swapping
        // registers between the half-waves for 32-byte runs per lane, v_permlane32_swap, measured slower: 146 us)
        float4 xin[2][4], b4[2][4];
        auto request = [&](int pc, float4 (&X)[4], float4 (&Bq)[4]) __attribute__((always_inline)) {
            const int i = pc >> 1, j = pc & 1;
            const int cell = c0 + 64 * wn + 32 * j + l31;
            const int64_t rowoff = (int64_t)min(cell, B - 1) * D;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int gene = j0 + 64 * wm + 32 * i + 8 * q + 4 * hh;
                const int gc = min(gene, D - 4);                      // D % 4 == 0
                X[q] = *reinterpret_cast<const float4*>(xa + rowoff + gc);
                Bq[q] = *reinterpret_cast<const float4*>(bias + gc);
            }
        };
        request(0, xin[0], b4[0]);
        __builtin_amdgcn_sched_barrier(0);
        f32x16 acc[2][2] = {{zero16(), zero16()}, {zero16(), zero16()}};
        mfma_ktile<false, false>(acc, Ws[0], Ds[0], wm, wn, lane);
        if (K > KT) mfma_ktile<false, false>(acc, Ws[1], Ds[1], wm, wn, lane);
        // acc[i][j][4 q + e]: gene j0 + 64 wm + 32 i + 8 q + 4 hh + e, cell c0 + 64 wn + 32 j + (lane & 31)
#pragma unroll
        for (int pc = 0; pc < 4; ++pc) {
            const int i = pc >> 1, j = pc & 1;
            if (pc + 1 < 4) request(pc + 1, xin[(pc + 1) & 1], b4[(pc + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
            const int cell = c0 + 64 * wn + 32 * j + l31;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int gene = j0 + 64 * wm + 32 * i + 8 * q + 4 * hh;
                const bool ok = cell < B && gene < D;
                const float4 xq = xin[pc & 1][q], bq = b4[pc & 1][q];
                const float xv[4] = {xq.x, xq.y, xq.z, xq.w};
                const float bv[4] = {bq.x, bq.y, bq.z, bq.w};
                float xr[4], dz[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    xr[e] = fmaxf(acc[i][j][4 * q + e] + bv[e], 0.f);
                    const float er = xr[e] - xv[e];
                    dz[e] = xr[e] > 0.f ? g.fo.coef * er : 0.f;
                    se += ok ? er * er : 0.f;
                    mism += (ok && ((xr[e] > 0.1f) != (xv[e] > 0.1f))) ? 1 : 0;
                }
                if (ok) {
                    *reinterpret_cast<float4*>(dza + (int64_t)cell * D + gene) = make_float4(dz[0], dz[1], dz[2], dz[3]);
                    if (xra) *reinterpret_cast<float4*>(xra + (int64_t)cell * D + gene) = make_float4(xr[0], xr[1], xr[2], xr[3]);
                }
            }
            // piece fence: keeps the requests one piece ahead (all four up front cost 128 registers and spill)
            asm volatile("" : "+v"(mism), "+v"(se));
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
    }
    se = wave_sum(se);
    const float mf = wave_sum((float)mism);
    if (lane == 0) { red[wv * 2] = se; red[wv * 2 + 1] = mf; }
    __syncthreads();
    if (tid == 0) {
        float* p = g.fo.part + ((int64_t)arm * g.n11 + (int64_t)blockIdx.x * g.KS + blockIdx.y) * 2;
        p[0] = red[0] + red[2] + red[4] + red[6];
        p[1] = red[1] + red[3] + red[5] + red[7];
    }
}

// fc11 forward + bias + reconstruction loss + dZ11 + d(d10), train step: k_bf16_fc11 with the second GEMM folded in, so
// dZ11 is not read back from HBM (200 MB per step at the benchmark shape) and one launch disappears.  Wave w of the four
// owns cells [32 w, 32 w + 32) of the block's 128 and ALL 128 genes of the current gene tile (four 32 x 32 accumulators,
// computed and consumed two at a time).  The dZ11 piece a wave has just formed IS the next MFMA's operand: the
// accumulator layout has the cell on the lane and four genes per register group, which is the A operand of
// d(d10)[cell][h] += sum_gene dZ11[cell][gene] W11[gene][h] once the sixteen genes of a K step are taken in the order
// the registers hold them (lower half-wave: genes 0-3, 8-11; upper: 4-7, 12-15 of the block of 16) -- no LDS round trip,
// no lane movement.  The B operand uses the same order: W11[gene][h] for four consecutive genes and one h per lane comes
// out of the [gene][h] image that is in LDS anyway through the transposing read (two ds_read_b64_tr_b16).  The d(d10)
// accumulators (32 cells x 128 h per wave) live across the whole gene range and are written once, to the gene-split slab.
// S16 (bf16 storage): x is read from its bf16 copy (g.fo.x addresses 2-byte elements; the loss and dZ11 then see the rounded
// x) and dZ11 is WRITTEN as bf16 (g.fo.dz likewise) -- exactly the values the d(d10) product here and the dW11 GEMM take
// anyway, so only the loss's view of x changes; the kernel moves 2 x 2 + 2 bytes per cell and gene instead of 2 x 4 + 4.
#ifndef BF16FC_ABL
#define BF16FC_ABL 0     // timing ablations of k_bf16_fc11g (diagnostic builds only; results wrong): 1 no W11 reloads, 2 no x loads, 4 no dZ11 stores
#endif
// W16: W11 is read from slice 0 of its planes ([D -> rup 128][128] bf16, launch_x3_planes) in sixteen-byte pieces of eight elements
// that go to the LDS image as they are, instead of fp32 rounded by every block (half the bytes from L2, no conversion; the
// timing ablation prices the fp32 tiles at 13 of the kernel's 84 us: profiles/r04_bf16_fc11g_ablation.txt; 84 -> 75 us.  Requesting
// x one half tile ahead in the registers this frees was measured too: 75.1 against 74.5 us, not kept)
template <bool S16, bool W16 = false>
__global__ __launch_bounds__(256, 2) void k_bf16_fc11g(const GemmArgs g_in) {
    const GemmArgs g = g_in;
    __shared__ __attribute__((aligned(16))) unsigned Ws[2][BT * LDB];   // W11 tile: [gene][k = h], two K tiles
    __shared__ __attribute__((aligned(16))) unsigned Ds[2][BT * LDB];   // d10 tile: [cell][k = h], two K tiles
    __shared__ __attribute__((aligned(16))) float bias_s[BT];
    __shared__ float red[8];
    const int arm = blockIdx.z, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int l31 = lane & 31, hh = lane >> 5;
    Operand od_h = g.a, ow_h = g.b;                   // a: d10 [B][H]; b: W11 [D][H]
    od_h.ptr += (int64_t)arm * g.a_arm;
    ow_h.ptr += (int64_t)arm * g.b_arm;
    const OperandDev od = make_operand_dev<false>(od_h), ow = make_operand_dev<false>(ow_h);
    const int c0 = blockIdx.x * BT;
    const int tiles = cdiv(g.N, BT);
    const int t0 = (int)(((int64_t)blockIdx.y * tiles) / g.KS), t1 = (int)(((int64_t)(blockIdx.y + 1) * tiles) / g.KS);
    const int K = g.K;
    const int ksteps1 = K > KT ? cdiv(K - KT, 16) : 0;      // MFMA steps of the second K tile that hold any k < K
    const float* xa = g.fo.x + (int64_t)arm * g.fo_x_arm;
    float* dza = g.fo.dz + (int64_t)arm * g.fo_arm;
    const float* bias = g.fo.bias + (int64_t)arm * g.bias_arm;
    const int B = g.fo.B, D = g.fo.D;
    {
        TileRegsT<false> t0r, t1r;
        tile_load<false, false>(t0r, od, c0, 0, K);
        tile_load<false, false>(t1r, od, c0, KT, K);
        tile_store<false, false>(Ds[0], t0r, od, c0, 0, K);
        tile_store<false, false>(Ds[1], t1r, od, c0, KT, K);
    }
    TileRegsT<false> w0, w1;
    OctRegs wo0, wo1;
    const unsigned no_map[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    auto w_request = [&](int j) __attribute__((always_inline)) {
        if constexpr (W16) {
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                oct_load<false, false>(wo0, ow, j, 0, 2 * KT, p, no_map);
                oct_load<false, false>(wo1, ow, j, KT, 2 * KT, p, no_map);
            }
        } else {
            tile_load<false, false>(w0, ow, j, 0, K);
            tile_load<false, false>(w1, ow, j, KT, K);
        }
    };
    auto w_store = [&](int j) __attribute__((always_inline)) {
        if constexpr (W16) {
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                oct_store<false>(Ws[0], wo0, ow, j, 0, 2 * KT, p);
                oct_store<false>(Ws[1], wo1, ow, j, KT, 2 * KT, p);
            }
        } else {
            tile_store<false, false>(Ws[0], w0, ow, j, 0, K);
            tile_store<false, false>(Ws[1], w1, ow, j, KT, K);
        }
    };
    float bnext = 0.f;
    if (t0 < t1) {
        w_request(t0 * BT);
        if (tid < BT) bnext = bias[min(t0 * BT + tid, D - 1)];
    }
    f32x16 gd[4] = {zero16(), zero16(), zero16(), zero16()};   // d(d10)[cell = 32 wv + row][h = 32 nt + (lane & 31)]
    float se = 0.f;
    int mism = 0;
    const int cell = c0 + 32 * wv + l31;
    const int64_t rowoff = g.fo.xmap ? (int64_t)g.fo.xmap[min(cell, B - 1)] : (int64_t)min(cell, B - 1) * D;   // (row map: the batch is rows of the resident matrix)
    const unsigned short* Wk16[2] = {reinterpret_cast<const unsigned short*>(Ws[0]), reinterpret_cast<const unsigned short*>(Ws[1])};
    typedef short s16x4 __attribute__((ext_vector_type(4)));
    typedef s16x4 __attribute__((address_space(3))) * lds_ptr;
    // transposing-read address of this lane inside a (4 genes x 16 h) block: row (lane & 15) >> 2, columns 4 (lane & 3) ..
    const int tr_off = ((lane & 15) >> 2) * (2 * LDB) + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
    for (int t = t0; t < t1; ++t) {
        const int j0 = t * BT;
        w_store(j0);
        if (tid < BT) bias_s[tid] = bnext;
        __syncthreads();
        if (!(BF16FC_ABL & 1) && t + 1 < t1) {
            w_request(j0 + BT);
            if (tid < BT) bnext = bias[min(j0 + BT + tid, D - 1)];
        }
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            // x of the two gene sub-tiles of this half (requested before the MFMAs that produce their z)
            // S16: a lane's four genes of a group are 8 bytes; the two half-waves of a cell (lanes l, l + 32: genes 8 q + 0 .. 3
            // and 8 q + 4 .. 7) instead request SIXTEEN bytes each -- the lower lane the whole group q = 2 qp, the upper one the
            // whole group 2 qp + 1 -- and trade halves with v_permlane32_swap where the values are used (and the other way
            // round for the dZ11 store): half the vector-memory instructions, and every store covers an aligned 16 bytes
            // (as 8-byte stores dZ11 cost 1.5 x its bytes in partially written lines, PMC)
            float4 xin[2][4];
            u32x4v xraw[2][2];
#pragma unroll
            for (int gl = 0; gl < 2; ++gl) {
                if constexpr (S16) {
#pragma unroll
                    for (int qp = 0; qp < 2; ++qp) {
                        const int gene = j0 + 32 * (2 * half + gl) + 8 * (2 * qp + hh);                   // D % 8 == 0
                        if (BF16FC_ABL & 2) xraw[gl][qp] = u32x4v{0x3c003c00u, 0x3c003c00u, 0u, 0u};
                        else xraw[gl][qp] = *reinterpret_cast<const u32x4v*>(reinterpret_cast<const unsigned short*>(xa) + rowoff + min(gene, D - 8));
                    }
                } else {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int gene = j0 + 32 * (2 * half + gl) + 8 * q + 4 * hh;
                        xin[gl][q] = *reinterpret_cast<const float4*>(xa + rowoff + min(gene, D - 4));      // D % 4 == 0
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            f32x16 acc[2] = {zero16(), zero16()};       // z^T[gene sub-tile 2 half + gl][this wave's 32 cells]
#pragma unroll
            for (int s = 0; s < KT / 16; ++s) {
                const bf16x8 b = frag8<false>(Ds[0], 32 * wv, s, lane);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag8<false>(Ws[0], 32 * (2 * half), s, lane), b, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag8<false>(Ws[0], 32 * (2 * half + 1), s, lane), b, acc[1], 0, 0, 0);
            }
            for (int s = 0; s < ksteps1; ++s) {
                const bf16x8 b = frag8<false>(Ds[1], 32 * wv, s, lane);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag8<false>(Ws[1], 32 * (2 * half), s, lane), b, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag8<false>(Ws[1], 32 * (2 * half + 1), s, lane), b, acc[1], 0, 0, 0);
            }
#pragma unroll
            for (int gl = 0; gl < 2; ++gl) {
                const int gi = 2 * half + gl;
                // acc[gl][4 q + e]: gene j0 + 32 gi + 8 q + 4 hh + e of cell `cell`
                float dzr[16];
                if constexpr (S16) {
                    // lanes l / l + 32 hold (group 2 qp | group 2 qp + 1) whole: swap(lo, hi) hands every lane its own
                    // four genes of both groups
#pragma unroll
                    for (int qp = 0; qp < 2; ++qp) {
                        const auto s0 = __builtin_amdgcn_permlane32_swap(xraw[gl][qp][0], xraw[gl][qp][2], false, false);
                        const auto s1 = __builtin_amdgcn_permlane32_swap(xraw[gl][qp][1], xraw[gl][qp][3], false, false);
                        xin[gl][2 * qp] = make_float4(__uint_as_float(s0[0] << 16), __uint_as_float(s0[0] & 0xFFFF0000u),
                                                      __uint_as_float(s1[0] << 16), __uint_as_float(s1[0] & 0xFFFF0000u));
                        xin[gl][2 * qp + 1] = make_float4(__uint_as_float(s0[1] << 16), __uint_as_float(s0[1] & 0xFFFF0000u),
                                                          __uint_as_float(s1[1] << 16), __uint_as_float(s1[1] & 0xFFFF0000u));
                    }
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int gene = j0 + 32 * gi + 8 * q + 4 * hh;
                    const bool ok = cell < B && gene < D;
                    const float4 xq = xin[gl][q];
                    const float4 bq = *reinterpret_cast<const float4*>(&bias_s[32 * gi + 8 * q + 4 * hh]);
                    const float xv[4] = {xq.x, xq.y, xq.z, xq.w};
                    const float bv[4] = {bq.x, bq.y, bq.z, bq.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float xr = fmaxf(acc[gl][4 * q + e] + bv[e], 0.f);
                        const float er = xr - xv[e];
                        dzr[4 * q + e] = (ok && xr > 0.f) ? g.fo.coef * er : 0.f;
                        se += ok ? er * er : 0.f;
                        mism += (ok && ((xr > 0.1f) != (xv[e] > 0.1f))) ? 1 : 0;
                    }
                    if constexpr (!S16) {
                        if (ok) *reinterpret_cast<float4*>(dza + (int64_t)cell * D + gene) =
                            make_float4(dzr[4 * q], dzr[4 * q + 1], dzr[4 * q + 2], dzr[4 * q + 3]);
                    }
                }
                if constexpr (S16) {
#pragma unroll
                    for (int qp = 0; qp < 2; ++qp) {
                        const int qe = 2 * qp, qo = 2 * qp + 1;
                        const auto s0 = __builtin_amdgcn_permlane32_swap(pack_bf16(dzr[4 * qe], dzr[4 * qe + 1]), pack_bf16(dzr[4 * qo], dzr[4 * qo + 1]), false, false);
                        const auto s1 = __builtin_amdgcn_permlane32_swap(pack_bf16(dzr[4 * qe + 2], dzr[4 * qe + 3]), pack_bf16(dzr[4 * qo + 2], dzr[4 * qo + 3]), false, false);
                        u32x4v w;
                        w[0] = s0[0]; w[1] = s1[0]; w[2] = s0[1]; w[3] = s1[1];
                        const int gene = j0 + 32 * gi + 8 * (2 * qp + hh);
                        if (!(BF16FC_ABL & 4) && cell < B && gene < D) *reinterpret_cast<u32x4v*>(reinterpret_cast<unsigned short*>(dza) + (int64_t)cell * D + gene) = w;
                        if (BF16FC_ABL & 4) asm volatile("" :: "v"(w));
                    }
                }
                // d(d10) += dZ11 piece (registers) x W11 rows 32 gi .. + 31 (LDS): two K steps of sixteen genes
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    u32x4v au;
                    au[0] = pack_bf16(dzr[8 * c + 0], dzr[8 * c + 1]);
                    au[1] = pack_bf16(dzr[8 * c + 2], dzr[8 * c + 3]);
                    au[2] = pack_bf16(dzr[8 * c + 4], dzr[8 * c + 5]);
                    au[3] = pack_bf16(dzr[8 * c + 6], dzr[8 * c + 7]);
                    const bf16x8 afr = __builtin_bit_cast(bf16x8, au);
                    const int grow = 32 * gi + 16 * c + 4 * hh;            // first of this lane's two gene groups (second: + 8)
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) {
                        const unsigned short* a = Wk16[nt >> 1] + grow * (2 * LDB) + 32 * (nt & 1) + tr_off;
                        const s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)a);
                        const s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(a + 8 * (2 * LDB)));
                        typedef short s16x8 __attribute__((ext_vector_type(8)));
                        s16x8 r;
                        r[0] = v0[0]; r[1] = v0[1]; r[2] = v0[2]; r[3] = v0[3];
                        r[4] = v1[0]; r[5] = v1[1]; r[6] = v1[2]; r[7] = v1[3];
                        gd[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr, __builtin_bit_cast(bf16x8, r), gd[nt], 0, 0, 0);
                    }
                }
                asm volatile("" : "+v"(mism), "+v"(se));
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __syncthreads();
    }
    // ---- d(d10) partial of this gene range: slab [NS][A][B][H]; gd[nt][r]: cell row acc_row(r), h = 32 nt + (lane & 31)
    {
        const int H = g.K;
        float* out = g.so.out + (int64_t)blockIdx.y * g.so.ks_stride + (int64_t)arm * g.so.arm_stride;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const int h = 32 * nt + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = c0 + 32 * wv + acc_row(r, lane);
                if (row < B && h < H) out[(int64_t)row * H + h] = gd[nt][r];
            }
        }
    }
    se = wave_sum(se);
    const float mf = wave_sum((float)mism);
    if (lane == 0) { red[wv * 2] = se; red[wv * 2 + 1] = mf; }
    __syncthreads();
    if (tid == 0) {
        float* p = g.fo.part + ((int64_t)arm * g.n11 + (int64_t)blockIdx.x * g.KS + blockIdx.y) * 2;
        p[0] = red[0] + red[2] + red[4] + red[6];
        p[1] = red[1] + red[3] + red[5] + red[7];
    }
}

// fc11 forward + bias + reconstruction loss + dZ11 + d(d10) of the fp32x3 configuration (the train step's dominant kernel).
// Same mathematics and data flow as k_bf16_fc11g -- z^T = W11 d10^T on the matrix pipe with the cell on the lane, the
// dZ11 piece in the accumulator IS the next product's operand -- with every operand as three bf16 slices and six slice
// products per product:
//   * [W11 | b11] and [d10 | 1] arrive as slice planes (k_presplit, once per step): the bias rides as k = fc_dim, so the
//     accumulator already holds z + b and the epilogue loads no bias;
//   * ONE WAVE PER SIMD (256 threads = 4 waves x 32 cells, up to 512 VGPRs each).  Two waves per SIMD at 256 registers
//     were measured first: a wave's VALU instructions hardly issue while its partner runs MFMAs (epilogue 1 860 cycles
//     alone, 3 600 - 4 700 beside the partner), the two waves ran in the sum of their times, and 256 registers left the
//     compiler no room to request LDS fragments ahead of the MFMAs that consume them.  With 512 registers the wave
//     overlaps its own work instead: the z product of piece p + 1 (42 MFMAs, fragments requested one K step ahead) is
//     issued through the epilogue of piece p (about six VALU instructions fit behind every MFMA);
//   * a wave keeps its cells' d10 slices in registers for the whole kernel (7 K steps x 3 slices x 4 VGPRs);
//   * LDS holds only W11 tiles (64 genes x 128 k x 3 slices, XOR-swizzled, 48 KB), three of them, filled by LDS-DMA
//     (global_load_lds_dwordx4: no registers, no VALU) two tiles ahead; a tile serves the z product as the A operand
//     (ds_read_b128) and the d(d10) product as the B operand through transposing reads (ds_read_b64_tr_b16);
//   * dZ11 is split into its slices in registers between the two products (eleven VALU instructions per pair);
//   * cells and genes that do not exist have z = 0 (zero rows of the planes) and read x = 0 (buffer range): no masks.
// grid (ceil(B / 128), gene splits NS, A); fc_dim + 1 <= 112.
// W11 tile image in LDS: [gene][k], 128 bf16 = 16 sixteen-byte blocks per row, NO padding; block c of row r sits at block
// c ^ fw_swz(r).  The image is read two ways -- row per lane (ds_read_b128: 16 lanes x 16 B must hit 16 different
// 4-bank groups) and transposed (ds_read_b64_tr_b16: per LDS cycle 2 x 16 lanes = 4 rows x 2 blocks x 2 column groups) --
// and no row pitch serves both (60 dwords: transposing reads 2-way conflicts, 19 % of the kernel's busy cycles in the PMC
// pass; 80: row reads 4-way).  The XOR below makes both conflict-free: rows r .. r + 3 differ in bits 2-3 of the block,
// rows r, r + 4, r + 8, r + 12 in bits 0-1.
#ifndef FC11_ABL
#define FC11_ABL 0     // timing ablations of k_x3_fc11g (diagnostic builds only; results wrong): 1 no x loads, 2 no dZ11 stores,
#endif                 // 4 no W11 DMA in the loop, 8 no epilogue, 16 / 32 no MFMAs in stage 1 / 2, 64 no mismatch count
constexpr int FW_ROW = 64;                     // dwords per W11 image row
constexpr int FW_PLANE = 64 * FW_ROW;          // dwords per slice image of a 64-gene tile
constexpr int FW_TILE = 3 * FW_PLANE;
__device__ __forceinline__ int fw_swz(int row) { return ((row & 3) << 2) | ((row >> 2) & 3); }
__global__ __launch_bounds__(256, 1) void k_x3_fc11g(const GemmArgs g_in) {
    const GemmArgs g = g_in;
    __shared__ __attribute__((aligned(16))) unsigned Wl[3 * FW_TILE];        // three tiles
    __shared__ float red[8];
    const int arm = blockIdx.z, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int l31 = lane & 31, hh = lane >> 5;
    const unsigned short* Wp = g.b.pl + (int64_t)arm * g.b.pl_arm;      // [3][Dr][128]
    const unsigned short* Dp = g.a.pl + (int64_t)arm * g.a.pl_arm;      // [3][Br][128]
    const int64_t wplane = g.b.pl_plane, dplane = g.a.pl_plane;
    const int B = g.fo.B, D = g.fo.D, H = g.K;
    const int c0 = blockIdx.x * 128;
    const int tiles = cdiv(D, 64);
    const int t0 = (int)(((int64_t)blockIdx.y * tiles) / g.KS), t1 = (int)(((int64_t)(blockIdx.y + 1) * tiles) / g.KS);
    const int cell = c0 + 32 * wv + l31;
    const float* xa = g.fo.x + (int64_t)arm * g.fo_x_arm;
    float* dza = g.fo.dz + (int64_t)arm * g.fo_arm;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xa), 0, (int)((g.fo.xmap ? g.fo.x_nrec : (int64_t)B * D) * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rz = __builtin_amdgcn_make_buffer_rsrc(dza, 0, (int)((int64_t)B * D * 4), 0x00020000);
    // this lane's cell: its row of x (through the row map when the batch is not materialised) and of dZ11
    const int rowoff = g.fo.xmap ? (int)g.fo.xmap[min(cell, B - 1)] : min(cell, B - 1) * D;          // B * D < 2^30 (fast-path condition)
    const bool cell_ok = cell < B;
    int dzoff = min(cell, B - 1) * D * 4;          // byte offset of the lane's row of dZ11 (a register of its own: with the product
    asm volatile("" : "+v"(dzoff));                // inside the store's select hipcc branched around it, and a branch ends a region)
    int dlim = cell_ok ? D : 0;                    // genes this lane may touch (a per-lane bound instead of `cell_ok && gene < D`:
    asm volatile("" : "+v"(dlim));                 // a uniform-looking `cell_ok` became an exec-mask branch around the select)

    // LDS-DMA of W11 tile t into buffer (t - t0) % 3: 3 slices x 64 rows x 16 sixteen-byte blocks = 48 wave instructions of
    // 1 KB, twelve per wave; LDS slot (row, block b) takes the plane's block b ^ fw_swz(row) (blocks 14, 15: the planes'
    // zero padding).
    // Addressing is split into what depends on the lane (once, one VGPR: the byte offset of the lane's block inside the
    // first rows of a tile's plane; the swizzle term does not depend on j because rows advance by 16) and what is uniform
    // (per instruction: a 64-bit scalar base and the LDS address for M0, a few scalar adds).  Written with a pointer into
    // Wl and a per-lane source pointer, hipcc converted the generic LDS pointer to a local one (null check included) and
    // moved both halves through v_readfirstlane for every one of the twelve instructions: ~130 of the loop's ~1060
    // non-matrix instructions, in a kernel that is bound by instruction issue.
    const int wv_s = __builtin_amdgcn_readfirstlane(wv);
    const unsigned wl_base = (unsigned)(size_t)(__attribute__((address_space(3))) void*)(&Wl[0]);
    const int lrow = 4 * wv + (lane >> 4);                                  // the lane's row inside a 16-row group
    const unsigned dma_voff = 2u * ((unsigned)lrow * 128u + 8u * (unsigned)((lane & 15) ^ fw_swz(lrow)));
    auto dma = [&](int t) __attribute__((always_inline)) {
        const int buf = (t - t0) % 3;
        const unsigned lds0 = wl_base + 4u * (unsigned)(buf * FW_TILE) + 1024u * (unsigned)wv_s;
        const unsigned short* src0 = Wp + (int64_t)t * (64 * 128);
#pragma unroll
        for (int j = 0; j < 12; ++j) {
            // instruction j of this wave: slice j >> 2, rows 16 (j & 3) + 4 wv .. + 3 of the tile (one 1 KB LDS chunk)
            const unsigned short* src = src0 + (int64_t)(j >> 2) * wplane + (j & 3) * (16 * 128);
            const unsigned lds_addr = lds0 + 4u * (unsigned)((j >> 2) * FW_PLANE) + 4096u * (unsigned)(j & 3);
            // (inline assembly, not __builtin_amdgcn_global_load_lds: hipcc's wait-count pass treats the builtin as a store to
            // "some" LDS and puts s_waitcnt vmcnt(0) in front of the next LDS read -- which, with the x prefetch just issued,
            // stalled every tile for a full memory round trip.  The waits for the DMA are the counted ones in the loop.)
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
                         :: "s"(lds_addr), "v"(dma_voff), "s"(src) : "memory");
        }
    };
    if (t0 < t1) dma(t0);
    if (t0 + 1 < t1) dma(t0 + 1);
    // this wave's d10 slices: lane = cell, K step s holds k = 16 s + 8 hh .. + 7
    bf16x8 dfr[7][3];
    {
        const int cr = min(cell, (int)(dplane / 128) - 1);
#pragma unroll
        for (int s = 0; s < 7; ++s)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
                dfr[s][pl] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4v*>(Dp + (int64_t)pl * dplane + (int64_t)cr * 128 + 16 * s + 8 * hh));
    }
    f32x16 gd[4] = {zero16(), zero16(), zero16(), zero16()};   // d(d10)[cell = 32 wv + row][h = 32 nt + (lane & 31)]
    float seq[4] = {0.f, 0.f, 0.f, 0.f};
    int mism = 0;                       // wave-uniform: mismatches of the whole wave
    typedef s16x4 __attribute__((address_space(3))) * lds_ptr;
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    // transposing read: this lane supplies the address of row q = (lane & 15) >> 2 of a (4 genes x 16 h) block, columns
    // 4 (lane & 3) .. + 3 of the 16 that start at 16 ((lane >> 4) & 1): sixteen-byte block 2 ((lane >> 4) & 1) + (lane & 3) / 2
    // of the h tile, its half (lane & 1)
    const int tr_q = (lane & 15) >> 2, tr_blk = 2 * ((lane >> 4) & 1) + ((lane & 3) >> 1), tr_half = 4 * (lane & 1);
    const int npieces = 2 * (t1 - t0);
    // piece p: genes [64 t0 + 32 p, + 32), rows 32 (p & 1) of tile t0 + p / 2 in LDS buffer (p / 2) % 3
    auto w_rows = [&](int p) __attribute__((always_inline)) { return Wl + ((p >> 1) % 3) * FW_TILE + 32 * (p & 1) * FW_ROW; };
    const int a_swz = fw_swz(l31);      // (a piece starts at row 0 or 32 of the tile: the row's low four bits are the lane's)
    auto a_frag = [&](const unsigned* Wr, int s, int pl) __attribute__((always_inline)) {
        return __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4v*>(Wr + pl * FW_PLANE + l31 * FW_ROW + 4 * ((2 * s + hh) ^ a_swz)));
    };
    auto request_x = [&](float4 (&X)[4], int j0g, int q) __attribute__((always_inline)) {   // x of gene group q of the piece at j0g
        const int gene = j0g + 8 * q + 4 * hh;                      // D % 4 == 0: a float4 exists entirely or not at all
        int off = gene < dlim ? (rowoff + gene) * 4 : -16;
        asm("" : "+v"(off));       // (a plain select: hipcc otherwise branches around two copies of the load, and a branch
                                   // ends the region in which MFMAs and VALU instructions can be interleaved)
        if constexpr (FC11_ABL & 1) { X[q] = make_float4(0.f, 0.f, 0.f, 0.f); return; }
        if constexpr (FC11_ABL & 128) {   // the same requests in whole 128-byte lines (eight lanes per row; wrong data)
            const int cc = min(c0 + 32 * wv + (lane >> 3) + 8 * q, B - 1), gg = min(j0g + 4 * (lane & 7), D - 4);
            off = (cc * D + gg) * 4;
        }
        X[q] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rx, off, 0, 0));
    };
    // six slice products of one K step
    // (one accumulation chain: dependent MFMAs of this shape issue back to back, tools/micro/x3_issue_bench.hip)
    auto mfma6 = [&](f32x16& acc, const bf16x8 (&a)[3], int s) __attribute__((always_inline)) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], dfr[s][0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], dfr[s][2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], dfr[s][1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], dfr[s][0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], dfr[s][1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], dfr[s][0], acc, 0, 0, 0);
    };
    // ------------------------------------------------------------------------------------------------------------
    // One piece = stage 1 (the z product of the NEXT piece, 42 MFMAs in seven regions of one K step, with this piece's
    // epilogue cut into eight chunks of VALU work behind them) + stage 2 (this piece's d(d10) product, 48 MFMAs in four
    // regions).  Every LDS fragment is requested TWO regions before the MFMAs that consume it (a ds_read costs 64+ cycles
    // when it is waited for at once -- tools/micro/x3_issue_bench.hip -- and hipcc sinks reads towards their uses, so the
    // regions are fenced with sched_barrier and the requests placed by hand); the rings of fragments carry over from
    // piece to piece.  Accumulators live in the accumulator half of the register file (this file is compiled without
    // -amdgpu-mfma-vgpr-form), which leaves the architectural VGPRs to the fragments in flight.
    // ------------------------------------------------------------------------------------------------------------
    bf16x8 an[3][3];          // A fragments (W11 slices) of the z product: ring over K steps
    bf16x8 wq[2][2][3];       // transposed W11 fragments of the d(d10) product: ring over regions; [h tile of the pair][slice]
    auto z_frags = [&](const unsigned* Wr, int s) __attribute__((always_inline)) {
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) an[s % 3][pl] = a_frag(Wr, s, pl);
    };
    auto d_frags = [&](const unsigned short* Wt16, int r) __attribute__((always_inline)) {   // region r: K step r >> 1, h tiles 2 (r & 1) ..
        const int row0 = 16 * (r >> 1) + 4 * hh + tr_q;     // this lane's row of the first gene group (second: + 8)
        const int sw0 = fw_swz(row0), sw1 = fw_swz(row0 + 8);
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
                const int blk = 4 * (2 * (r & 1) + j) + tr_blk;     // sixteen-byte block of the row (h tile 2 (r & 1) + j)
                const unsigned short* base = Wt16 + pl * (2 * FW_PLANE) + tr_half;
                const s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(base + row0 * (2 * FW_ROW) + 8 * (blk ^ sw0)));
                const s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(base + (row0 + 8) * (2 * FW_ROW) + 8 * (blk ^ sw1)));
                s16x8 rr;
                rr[0] = v0[0]; rr[1] = v0[1]; rr[2] = v0[2]; rr[3] = v0[3];
                rr[4] = v1[0]; rr[5] = v1[1]; rr[6] = v1[2]; rr[7] = v1[3];
                wq[r & 1][j][pl] = __builtin_bit_cast(bf16x8, rr);
            }
    };
    float4 xa_[4], xb_[4];
    f32x16 acc_a = zero16();
    // The d10 slices are complete HERE, on every path into the main loop (they live in the accumulator half of the register
    // file from now on: only MFMAs read them).  With this wait inside the `if` below, the path around it -- a block without
    // tiles, which never runs the loop, but hipcc cannot know -- carried "d10 still in flight" to the loop header, and the
    // wait-count pass put vmcnt(24) / (22) / (7) in front of the first MFMAs of EVERY piece: "all but the seven youngest
    // vector-memory operations have completed" right behind the four x requests, i.e. a wait for the previous piece's stores
    // and for the W11 tile requested half a tile earlier -- 50 us of the kernel's 155 (no-load and no-store ablations).
    __builtin_amdgcn_s_waitcnt(0x0F70);            // vmcnt(0)
#pragma unroll
    for (int s7 = 0; s7 < 7; ++s7)
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) asm volatile("" : "+a"(dfr[s7][pl]));
    if (npieces > 0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) request_x(xa_, t0 * 64, q);
        __builtin_amdgcn_s_waitcnt(0x0F70);        // vmcnt(0): both first tiles and x
        // (hipcc's wait-count pass does not see that wait: without a use of the registers HERE it keeps "x may still be in
        // flight" alive around the loop's back edge)
#pragma unroll
        for (int q = 0; q < 4; ++q) asm volatile("" : "+v"(xa_[q].x), "+v"(xa_[q].y), "+v"(xa_[q].z), "+v"(xa_[q].w));
        __syncthreads();
        if (t0 + 2 < t1) dma(t0 + 2);
        // z of piece 0
        const unsigned* Wr = w_rows(0);
#pragma unroll
        for (int s = 0; s < 7; ++s) {
            bf16x8 a[3];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) a[pl] = a_frag(Wr, s, pl);
            mfma6(acc_a, a, s);
        }
        const unsigned* W1 = w_rows(npieces > 1 ? 1 : 0);
        z_frags(W1, 0);
        z_frags(W1, 1);
    }
#ifdef X3_STAMPS
    long long t_s1 = 0, t_dma = 0, t_s2 = 0, tw = 0, tp = __builtin_amdgcn_s_memtime(), tbeg = tp;
#define X3_ST(var) { const long long t1_ = __builtin_amdgcn_s_memtime(); var += t1_ - tp; tp = t1_; }
#else
#define X3_ST(var)
#endif
    // piece p: epilogue reads `acc` / `xin`, the z product of piece p + 1 goes to `accn`, its x to `xnx`
    // (ONE z accumulator: the piece copies its z out of the accumulator registers at its head -- the epilogue's VALU
    // instructions cannot read those anyway -- and the next piece's product goes into the same registers.  With two
    // accumulators swapped from piece to piece hipcc parked d(d10)'s first tile in VGPRs around every z product: 32 more
    // accumulator copies per tile.)
    auto piece = [&](int p, f32x16& accn, float4 (&xin)[4], float4 (&xnx)[4], int dma_tile) __attribute__((always_inline)) {
        const int j0g = t0 * 64 + 32 * p;
        const unsigned* Wn = w_rows(p + 1 < npieces ? p + 1 : p);
        const unsigned short* Wt16 = reinterpret_cast<const unsigned short*>(w_rows(p));
        f32x16 acc = accn;
        asm volatile("" : "+v"(acc));
        accn = zero16();
        unsigned au[2][3][4];          // dZ11 slices: K step c (sixteen genes in register order), slice, four dwords
        float dzq[4];
        // x of the next piece first: requested in front of this piece's stores, the wait for it (vector-memory instructions
        // retire in order) then does not include them
#pragma unroll
        for (int q = 0; q < 4; ++q) request_x(xnx, j0g + 32, q);
        // chunk 2 q: loss terms, dZ11 and its store for gene group q (acc[4 q + e] is gene j0g + 8 q + 4 hh + e of cell
        // `cell`); chunk 2 q + 1: its slices.  Eight VALU instructions per element; the mismatch count is kept on the scalar
        // unit (lane-mask population counts), outside the VALU's dependency chains.
        auto chunk = [&](int ch) __attribute__((always_inline)) {
            const int q = ch >> 1;
            if constexpr (FC11_ABL & 8) {
                if (ch & 1) for (int pl = 0; pl < 3; ++pl) { au[q >> 1][pl][2 * (q & 1)] = 0x3f803f80u; au[q >> 1][pl][2 * (q & 1) + 1] = 0x3f803f80u; }
                return;
            }
            if ((ch & 1) == 0) {
                const int gene = j0g + 8 * q + 4 * hh;
                const float xv[4] = {xin[q].x, xin[q].y, xin[q].z, xin[q].w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float a_ = acc[4 * q + e];
                    // max(z, 0) as ONE instruction: a signed-integer maximum of the bit pattern (fmaxf also canonicalises
                    // its input -- a second v_max_f32 per element)
                    const float xr = __int_as_float(max(__float_as_int(a_), 0));
                    const float er = xr - xv[e];
                    seq[q] = __builtin_fmaf(er, er, seq[q]);
                    const float d_ = g.fo.coef * er;
                    dzq[e] = a_ > 0.f ? d_ : 0.f;
                    if constexpr (!(FC11_ABL & 64))
                    mism += __builtin_popcountll(__builtin_amdgcn_ballot_w64(xr > 0.1f) ^ __builtin_amdgcn_ballot_w64(xv[e] > 0.1f));
                }
                // (always issued -- the counted wait at the end of a tile relies on it; what must not be written gets an offset
                // beyond the buffer's range, which the hardware drops)
                int soff = gene < dlim ? dzoff + gene * 4 : -16;
                asm("" : "+v"(soff));
                if constexpr (FC11_ABL & 256) {
                    const int cc = min(c0 + 32 * wv + (lane >> 3) + 8 * q, B - 1), gg = min(j0g + 4 * (lane & 7), D - 4);
                    soff = (cc * D + gg) * 4;
                }
                if constexpr (!(FC11_ABL & 2))
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4v, make_float4(dzq[0], dzq[1], dzq[2], dzq[3])), rz, soff, 0, 0);
            } else {
                unsigned w0[3], w1[3];
                split3(dzq[0], dzq[1], w0);
                split3(dzq[2], dzq[3], w1);
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) { au[q >> 1][pl][2 * (q & 1)] = w0[pl]; au[q >> 1][pl][2 * (q & 1) + 1] = w1[pl]; }
            }
        };
        // (this piece's x was requested a piece ago: one wait for all of it here, in front of the DMA's assembly -- see below)
#pragma unroll
        for (int q = 0; q < 4; ++q) asm volatile("" : "+v"(xin[q].x), "+v"(xin[q].y), "+v"(xin[q].z), "+v"(xin[q].w));
        // ---- stage 1: seven regions of six MFMAs; chunks 0 .. 4 of the epilogue behind them (the other three go behind the
        // first MFMAs of stage 2, which has no other VALU work: 42 MFMAs + 5 chunks against 48 MFMAs + 3 chunks)
#pragma unroll
        for (int s = 0; s < 7; ++s) {
            __builtin_amdgcn_sched_barrier(0);
            if (s + 2 < 7) z_frags(Wn, s + 2);
            if (s == 5) d_frags(Wt16, 0);           // (the first region of stage 2: its slot of the ring is free)
            if constexpr (!(FC11_ABL & 16)) mfma6(accn, an[s % 3], s);
            else { asm volatile("" :: "v"(an[s % 3][0]), "v"(an[s % 3][1]), "v"(an[s % 3][2])); }
            if (s < 5) chunk(s);
        }
        asm volatile("" : "+s"(mism));
        // the DMA of a later tile goes here, behind the last use of this piece's x: hipcc does not count the assembly's
        // instructions, so a wait it places for a register loaded BEFORE them also waits for them
#ifdef X3_STAMPS
        asm volatile("" :: "v"(accn), "v"(au[0][0][0]), "v"(au[1][2][3]));
        X3_ST(t_s1)
#endif
        if (!(FC11_ABL & 4) && dma_tile >= 0) dma(dma_tile);
#ifdef X3_STAMPS
        X3_ST(t_dma)
#endif
        // ---- stage 2: d(d10) += dZ11 piece (registers) x W11 rows of piece p (LDS, transposed)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            __builtin_amdgcn_sched_barrier(0);
            if (r + 1 < 4) d_frags(Wt16, r + 1);    // the other slot of the ring (region r - 1 is done with it)
            if (r >= 2) {
                // the first two K steps of the next piece's z product (piece p + 2: its tile has landed -- it is the tile of
                // piece p + 1 or the one the barrier at the end of the previous tile waited for)
                const unsigned* W2 = w_rows(p + 2 < npieces ? p + 2 : p);
                z_frags(W2, r - 2);
            }
            if (r == 0) { chunk(5); chunk(6); }       // (gene groups 2 and 3 feed K step 1 = regions 2 and 3)
            if (r == 1) chunk(7);
            const int c = r >> 1;
            bf16x8 af[3];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
                u32x4v u;
                u[0] = au[c][pl][0]; u[1] = au[c][pl][1]; u[2] = au[c][pl][2]; u[3] = au[c][pl][3];
                af[pl] = __builtin_bit_cast(bf16x8, u);
            }
            // product-major: consecutive MFMAs go to the two h tiles of the region (independent accumulators)
            const int n0 = 2 * (r & 1);
            if constexpr (FC11_ABL & 32) {
#pragma unroll
                for (int j = 0; j < 2; ++j) asm volatile("" :: "v"(wq[r & 1][j][0]), "v"(wq[r & 1][j][1]), "v"(wq[r & 1][j][2]), "v"(af[0]), "v"(af[1]), "v"(af[2]));
                continue;
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) gd[n0 + j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[2], wq[r & 1][j][0], gd[n0 + j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < 2; ++j) gd[n0 + j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], wq[r & 1][j][2], gd[n0 + j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < 2; ++j) gd[n0 + j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1], wq[r & 1][j][1], gd[n0 + j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < 2; ++j) gd[n0 + j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1], wq[r & 1][j][0], gd[n0 + j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < 2; ++j) gd[n0 + j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], wq[r & 1][j][1], gd[n0 + j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < 2; ++j) gd[n0 + j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], wq[r & 1][j][0], gd[n0 + j], 0, 0, 0);
        }
#ifdef X3_STAMPS
        asm volatile("" :: "v"(gd[0]), "v"(gd[3]));
        X3_ST(t_s2)
#endif
        __builtin_amdgcn_sched_barrier(0);
    };
    for (int t = t0; t < t1; ++t) {
        const int p = 2 * (t - t0);
        // the first piece of tile t also requests tile t + 2 (into the buffer of tile t - 1, which every wave left at the
        // barrier below; tiles t0 .. t0 + 2 were requested by the prologue)
        piece(p, acc_a, xa_, xb_, (t > t0 && t + 2 < t1) ? t + 2 : -1);
        piece(p + 1, acc_a, xb_, xa_, -1);
        // ---- end of a tile (two pieces): tile t + 2 is needed next (the z product runs one piece ahead).  It was requested
        // in front of eight younger vector-memory instructions (the second piece's 4 loads + 4 stores); vector-memory
        // instructions retire in order, so "at most four outstanding" (those stores) guarantees this wave's share of it has
        // landed, and the barrier that of the others.  The barrier also lets tile t's buffer be overwritten.
        __builtin_amdgcn_s_waitcnt(0x0F74);                // vmcnt(4)
        __syncthreads();
        X3_ST(tw)
    }
#ifdef X3_STAMPS
    if (g.dbg && lane == 0 && (wv == 0 || wv == 3) && blockIdx.x == 3 && blockIdx.y == 2 && blockIdx.z == 0) {
        long long* o = g.dbg + (wv ? 1 : 0) * 8;
        o[0] = t_s1; o[1] = t_dma; o[2] = t_s2; o[3] = npieces; o[4] = __builtin_amdgcn_s_memtime() - tbeg; o[5] = tw; o[6] = 0;
    }
#endif
    // ---- d(d10) partial of this gene range: slab [NS][A][B][H]; gd[nt][r]: cell row acc_row(r), h = 32 nt + (lane & 31)
    {
        float* out = g.so.out + (int64_t)blockIdx.y * g.so.ks_stride + (int64_t)arm * g.so.arm_stride;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const int h = 32 * nt + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = c0 + 32 * wv + acc_row(r, lane);
                if (row < B && h < H) out[(int64_t)row * H + h] = gd[nt][r];
            }
        }
    }
    const float se = wave_sum((seq[0] + seq[1]) + (seq[2] + seq[3]));
    if (lane == 0) { red[wv * 2] = se; red[wv * 2 + 1] = (float)mism; }
    __syncthreads();
    if (tid == 0) {
        float* pp = g.fo.part + ((int64_t)arm * g.n11 + (int64_t)blockIdx.x * g.KS + blockIdx.y) * 2;
        pp[0] = (red[0] + red[2]) + (red[4] + red[6]);
        pp[1] = (red[1] + red[3]) + (red[5] + red[7]);
    }
    // the loss finalisation sums all n11 slots of an arm: the first block of the arm clears the ones no block writes
    // (instead of a memset launch in front of the kernel)
    if (blockIdx.x == 0 && blockIdx.y == 0)
        for (int i = (int)gridDim.x * g.KS * 2 + tid; i < 2 * g.n11; i += 256) g.fo.part[(int64_t)arm * g.n11 * 2 + i] = 0.f;
}

// The small-layer weight / bias gradients (twelve products per arm, out[m][n] = sum_b P[b][m] Q'[b][n], m <= 128,
// n + 1 <= 128, K = batch) on the split engine: the ping-pong kernel with ONE PRODUCT PER GROUP -- the two groups of a block
// take products 2 b and 2 b + 1 of the (product, arm) list, each with its own operands (both batch-reduced, i.e. K-minor;
// Q optionally BatchNorm-normalised on load and extended by the ones column of the bias gradient, as k_gemm_tn does).
// grid (ceil(ndesc * A / 2), KS).  2.4 GFLOP in all: 57 us on the fp32 matrix instruction (pipe 28 % busy).
__global__ __launch_bounds__(512, 1) void k_x3_small(const TnDescs descs, int ndesc, int A, int B, int KS) {
    typedef Eng<3> E;
    constexpr int KTv = E::KT;
    __shared__ __attribute__((aligned(16))) unsigned As[2][3 * E::PLANE];
    __shared__ __attribute__((aligned(16))) unsigned Bs[2][3 * E::PLANE];
    const int grp = threadIdx.x >> 8, tid = threadIdx.x & 255, lane = tid & 63, wv = tid >> 6, wm = wv >> 1, wn = wv & 1;
    const int unit = 2 * blockIdx.x + grp;
    const bool active = unit < ndesc * A;
    const int di = active ? unit % ndesc : 0, arm = active ? unit / ndesc : 0;
    const TnDesc& dr = descs.d[di];
    const TnDesc d = {dr.P, dr.p_arm_stride, dr.ldp, dr.Mv, dr.Q, dr.q_arm_stride, dr.ldq, dr.Nv, dr.q_ones, dr.q_xmask,
                      dr.q_mean, dr.q_rstd, dr.out, dr.out_arm_stride, dr.out_ks_stride, dr.ldo};
    Operand oa_h{d.P + (int64_t)arm * d.p_arm_stride, d.ldp, d.Mv, B, 1, nullptr, 0, -1, nullptr, 0, 0, 0};
    Operand ob_h{d.Q + (int64_t)arm * d.q_arm_stride, d.ldq, d.Nv, B, 1, nullptr, 0, d.q_ones ? d.Nv : -1, nullptr, 0, 0, 0};
    OperandDev oa = make_operand_dev<true>(oa_h), ob = make_operand_dev<true>(ob_h);
    const bool bn = d.q_mean != nullptr;
    {
        const int r4 = (tid & 31) * 4;
        float sb[4] = {0.f, 0.f, 0.f, 0.f}, ml[4] = {1.f, 1.f, 1.f, 1.f};
        if (bn) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (r4 + e < d.Nv) { sb[e] = d.q_mean[(int64_t)arm * d.Nv + r4 + e]; ml[e] = d.q_rstd[(int64_t)arm * d.Nv + r4 + e]; }
        }
        ob.bn_sub = make_float4(sb[0], sb[1], sb[2], sb[3]);
        ob.bn_mul = make_float4(ml[0], ml[1], ml[2], ml[3]);
    }
    const int nkt = cdiv(B, KTv);
    const int kb = (int)(((int64_t)blockIdx.y * nkt) / KS) * KTv;
    const int ke = min(B, (int)(((int64_t)(blockIdx.y + 1) * nkt) / KS) * KTv);
    const int n = ke > kb ? cdiv(ke - kb, KTv) : 0;
    f32x16 acc[2][2] = {{zero16(), zero16()}, {zero16(), zero16()}};
    TileRegsT<false, E::NQ> ta, tb;
    unsigned* const Ad = As[grp];
    unsigned* const Bd = Bs[grp];
    auto stage = [&](int kst, int kld, auto load_tag) __attribute__((always_inline)) {
        constexpr bool LOAD = decltype(load_tag)::value;
#pragma unroll
        for (int i = 0; i < E::NQ; ++i) {
            quad_store<true, false, 3>(Ad, ta, oa, 0, kst, ke, i);
            if constexpr (LOAD) quad_load<true, false, 3>(ta, oa, 0, kld, ke, i);
            quad_store<true, false, 3, true>(Bd, tb, ob, 0, kst, ke, i);
            if constexpr (LOAD) quad_load<true, false, 3>(tb, ob, 0, kld, ke, i);
        }
    };
    if (active && n > 0) {
        tile_load<true, false, 3>(ta, oa, 0, kb, ke);
        tile_load<true, false, 3>(tb, ob, 0, kb, ke);
    }
    for (int p = 0; p <= 2 * n; ++p) {
        const int q = p - grp;
        if (active && q >= 0 && q < 2 * n) {
            const int k0 = kb + (q >> 1) * KTv;
            if (q & 1) mfma_ktile<true, true, 3>(acc, Ad, Bd, wm, wn, lane);
            else if (k0 + KTv < ke) stage(k0, k0 + KTv, VecTag{});
            else stage(k0, k0, ScalarTag{});
        }
        __syncthreads();
    }
    if (!active) return;
    float* out = d.out + (int64_t)blockIdx.y * d.out_ks_stride + (int64_t)arm * d.out_arm_stride;
    const int ncols = d.Nv + d.q_ones, l31 = lane & 31;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = 64 * wn + 32 * j + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = 64 * wm + 32 * i + acc_row(r, lane);
                if (row < d.Mv && col < ncols) out[(int64_t)row * d.ldo + col] = acc[i][j][r];
            }
        }
}

// ---------------------------------------------------------------------------------------------------------------
// host launchers (same workspace layouts and split factors as the fp32 fast path)
// ---------------------------------------------------------------------------------------------------------------
static Operand kmajor(const float* p, int64_t ld, int rows, int K) { return Operand{p, ld, rows, K, 0, nullptr, 0, -1, nullptr, 0, 0, 0, nullptr, 0, 0, 0}; }
static Operand kminor(const float* p, int64_t ld, int rows, int K) { return Operand{p, ld, rows, K, 1, nullptr, 0, -1, nullptr, 0, 0, 0, nullptr, 0, 0, 0}; }

// Slice planes of the four small operands (fp32x3 engine).  They are written by launch_x3_planes at fixed points of the
// step -- W1 and [W11 | b11] at the start of the forward pass, [d10 | 1] behind the decoder chain, dZ1 behind the encoder's
// backward chain -- and read by the GEMM launchers below, which only fill in the operand's plane fields.
static inline int rup_i(int a, int b) { return cdiv(a, b) * b; }
enum { PL_W1 = 0, PL_W11 = 1, PL_D10 = 2, PL_DZ1 = 3 };
struct PlaneGeom { int R, C, Rp, Cp, ones_col; int64_t ws_off; };
static PlaneGeom plane_geom(const Ctx& c, int kind) {
    const mmvae_dims& d = c.d;
    switch (kind) {
        case PL_W1:  return PlaneGeom{d.H, d.D, 128, rup_i(d.D, 32), -1, c.lay.pl_w1};                 // W1 [H][D]
        case PL_W11: return PlaneGeom{d.D, d.H, rup_i(d.D, 128), 128, d.H, c.lay.pl_w11};              // [W11 | b11] [D][H + 1]
        case PL_D10: return PlaneGeom{d.B, d.H, rup_i(d.B, 256), 128, d.H, c.lay.pl_d10};              // [d10 | 1] [B][H + 1]
        default:     return PlaneGeom{d.B, d.H, rup_i(d.B, 256), 128, -1, c.lay.pl_dz1};               // dZ1 [B][H]
    }
}
static void use_planes(const Ctx& c, Operand& o, int kind) {
    const PlaneGeom g = plane_geom(c, kind);
    o.pl = reinterpret_cast<const unsigned short*>(c.ws + g.ws_off);
    o.pl_plane = (int64_t)g.Rp * g.Cp; o.pl_arm = 3 * o.pl_plane; o.pl_ld = g.Cp;
}
static SplitJob plane_job(const Ctx& c, int kind, const float* src, int64_t ld, int64_t src_arm, const float* col_src = nullptr, int64_t col_arm = 0) {
    const PlaneGeom g = plane_geom(c, kind);
    return SplitJob{src, ld, src_arm, g.R, g.C, g.Rp, g.Cp, g.ones_col, reinterpret_cast<unsigned short*>(c.ws + g.ws_off),
                    3 * (int64_t)g.Rp * g.Cp, col_src, col_arm};
}
static int launch_presplit(hipStream_t s, int A, const SplitJob* jobs, int n, const XbitsJob* xb = nullptr) {
    SplitJobs js{};
    if (xb) js.xb = *xb;
    js.n = n;
    for (int i = 0; i < n; ++i) {   // a job gets the blocks its size asks for (small-layer planes: 8, W11: 320)
        js.j[i] = jobs[i];
        js.first[i + 1] = js.first[i] + (int)imin64(1024, cdiv64((int64_t)jobs[i].Rp * (jobs[i].Cp / 8), 256));
    }
    hipLaunchKernelGGL(k_presplit, dim3(js.xb.blocks + js.first[n], 1, A), dim3(256), 0, s, js);
    HIP_LAUNCH_CHECK("k_presplit");
    return 0;
}

// fp32x3: write the slice planes of the small operands (bit 0: W1 and [W11 | b11], from the parameters; bit 1: [d10 | 1];
// bit 2: dZ1) -- one small launch each time, ahead of the GEMMs that copy them into LDS.  No-op for the other engines.
int launch_x3_planes(const Ctx& c, const float* params, int which, const mmvae_noise* nz) {
    const bool x3 = split3_gemms(c);
    if (!x3 && !chain_x3_ok(c)) return 0;
    // bit 4 (the head of a training step's forward pass, dropout on): this launch also makes the keep-mask and zeroes the
    // loss partial slots and the forward accumulator sets -- k_make_xbits' work (launch_forward_zero) without its launch
    XbitsJob xb{};
    const bool head = (which & 16) && c.h.training && c.h.x_drop > 0.f && !c.tune(MMVAE_TUNE_PRESPLIT_ALL);
    if (head) {
        const mmvae_dims& dd = c.d;
        xb.nz = make_noise_dev(nz, c.h);
        xb.A = dd.A; xb.B = dd.B; xb.D = dd.D; xb.wpr = cdiv(dd.D, 32);
        xb.bits = reinterpret_cast<uint32_t*>(c.ws + c.lay.xbits);
        const int wpt = xb.nz.mode != 0 && xb.nz.x_mlog2 <= 2 ? (int)(4u >> xb.nz.x_mlog2) : 1;   // as in make_xbits_range
        xb.blocks = (int)imin64(2048, cdiv64((int64_t)dd.B * cdiv(xb.wpr, wpt), 256));
        xb.zero_p = c.ws + c.lay.fc11_part;
        xb.zero_n4 = (int)(c.fwd_zero_floats() / 4);
        c.fwd_zeroed = true;
        if (c.x_rows) {
            xb.rows = c.x_rows; xb.map = reinterpret_cast<unsigned*>(c.ws + c.lay.rowmap);
            xb.rows_ld = c.x_ld; xb.n_rows = c.x_nrows;
            c.rowmap_ready = true;
        }
    }
    if (!x3) which &= 9;   // bf16 configuration: only the chain kernels take planes
    const mmvae_dims& d = c.d;
    SplitJob jobs[24];
    int n = 0;
    if (which & 1 && x3) {
        jobs[n++] = plane_job(c, PL_W1, params + c.po.o[0], d.D, c.po.per_arm);
        jobs[n++] = plane_job(c, PL_W11, params + c.po.o[26], d.H, c.po.per_arm, params + c.po.o[27], c.po.per_arm);   // bias: column fc_dim
    } else if ((which & 1) && bf16_narrow_planes(c)) {
        // bf16 configuration on bf16 storage: slice 0 of W1's planes IS bf16(W1) -- fc1 reads its narrow operand from it in
        // sixteen-byte pieces of eight elements, like x, instead of fp32 rounded by every block tile (half the bytes)
        jobs[n++] = plane_job(c, PL_W1, params + c.po.o[0], d.D, c.po.per_arm);
        // ... and slice 0 of [W11 | b11]'s planes is bf16(W11) for the fused fc11 kernel (the bias column meets a zero of d10)
        jobs[n++] = plane_job(c, PL_W11, params + c.po.o[26], d.H, c.po.per_arm, params + c.po.o[27], c.po.per_arm);
    }
    if (which & 9) {   // bit 3: the small layers alone (a backward pass that is its own call)
        if (chain_x3_ok(c) && !c.small_planes) {   // the small layers' weights for the chain kernels: slot s = [N][K] of fc2..fc5, fc6..fc10
            const int H = d.H, L = d.L, CS = d.C + d.S;
            const int ti[9] = {2, 4, 6, 8, 16, 18, 20, 22, 24};                    // parameter tensor index of the weight
            const int nn[9] = {H, H, H, L, L, H, H, H, H}, kk[9] = {H, H, H, H, CS, L, H, H, H};
            unsigned short* base = reinterpret_cast<unsigned short*>(c.ws + c.lay.pl_small);
            for (int s = 0; s < 9; ++s) {
                jobs[n++] = SplitJob{params + c.po.o[ti[s]], kk[s], c.po.per_arm, nn[s], kk[s], 128, 128, -1,
                                     base + (int64_t)s * 3 * 128 * 128, (int64_t)PL_SMALL_SLOTS * 3 * 128 * 128, nullptr, 0, 0};
                // and [K][N] for the backward chain (its contraction runs over N)
                jobs[n++] = SplitJob{params + c.po.o[ti[s]], kk[s], c.po.per_arm, kk[s], nn[s], 128, 128, -1,
                                     base + (int64_t)(9 + s) * 3 * 128 * 128, (int64_t)PL_SMALL_SLOTS * 3 * 128 * 128, nullptr, 0, 1};
            }
            c.small_planes = true;
        }
    }
    if ((which & 2) && !dec_chain_writes_planes(c)) jobs[n++] = plane_job(c, PL_D10, c.ws + c.lay.Dk[4], d.H, (int64_t)d.B * d.H);
    if ((which & 4) && !bn_apply_writes_planes(c)) jobs[n++] = plane_job(c, PL_DZ1, c.ws + c.lay.DZ[1], d.H, (int64_t)d.B * d.H);
    return (n || head) ? launch_presplit(c.stream, d.A, jobs, n, head ? &xb : nullptr) : 0;
}

int launch_fc1_fwd_bf16(const Ctx& c, const float* params, const float* x, int64_t xs) {
    const mmvae_dims& d = c.d;
    const bool use_mask = c.h.training && c.h.x_drop > 0.f;
    GemmArgs g{};
    g.a = kmajor(x, d.D, d.B, d.D);
    g.a_arm = xs;
    if (use_mask) { g.a.bits = reinterpret_cast<const uint32_t*>(c.ws + c.lay.xbits); g.a.wpr = cdiv(d.D, 32); g.a_bits_arm = (int64_t)d.B * g.a.wpr; }
    g.b = kmajor(params + c.po.o[0], d.D, d.H, d.D);
    g.b_arm = c.po.per_arm;
    g.M = d.B; g.N = d.H; g.K = d.D; g.KS = c.lay.sp.ks_fc1; g.A = d.A;
    g.so = SlabOut{c.ws + c.lay.fc1_slab, (int64_t)d.A * d.B * NP, (int64_t)d.B * NP, NP, d.B, d.H};
    g.dbg = reinterpret_cast<long long*>(c.ws + c.lay.loss_scratch + 2048);
    if (c.x_rows) {   // the batch as rows of the resident matrix (mmvae_train_step_rows): x is read through the row map
        if (!c.rowmap_ready) { set_error("row-indexed batches need the fused step's head launch"); return MMVAE_E_UNSUPPORTED; }
        g.a.rowmap = reinterpret_cast<const unsigned*>(c.ws + c.lay.rowmap); g.a.nrec = c.x_nrows * c.x_ld; g.a.map_n = d.B;
        if (split3_gemms(c)) {
            use_planes(c, g.b, PL_W1);
            hipLaunchKernelGGL((k_x3_gemm<false, false, 0, 2, true, 1>), dim3(cdiv(cdiv(d.B, BT), 2), g.KS, d.A), dim3(512), 0, c.stream, g);
        } else if (c.x16) {         // bf16 storage: x from its bf16 copy
            g.a.ptr = reinterpret_cast<const float*>(c.x16); g.a.src16 = 1;
            if (bf16_narrow_planes(c)) {   // ... and W1 from slice 0 of its planes (launch_x3_planes): [128][rup(D, 32)] bf16, zero rows beyond H
                const PlaneGeom pg = plane_geom(c, PL_W1);
                g.b.ptr = reinterpret_cast<const float*>(c.ws + pg.ws_off); g.b.ld = pg.Cp; g.b.src16 = 1;
                g.b_arm = 3 * (int64_t)pg.Rp * pg.Cp / 2;       // arm stride in FLOATS of the pointer arithmetic (planes: 2-byte elements)
                hipLaunchKernelGGL((k_bf16_gemm<false, false, 0, 1, 3>), dim3(cdiv(d.B, BT) * cdiv(d.H, BT), g.KS, d.A), dim3(256), 0, c.stream, g);
            } else
                hipLaunchKernelGGL((k_bf16_gemm<false, false, 0, 1, 1>), dim3(cdiv(d.B, BT) * cdiv(d.H, BT), g.KS, d.A), dim3(256), 0, c.stream, g);
        } else
            hipLaunchKernelGGL((k_bf16_gemm<false, false, 0, 1>), dim3(cdiv(d.B, BT) * cdiv(d.H, BT), g.KS, d.A), dim3(256), 0, c.stream, g);
    } else if (split3_gemms(c)) {   // (fc_dim <= 124: one tile wide, the two tiles of a block share the W1 tile)
        use_planes(c, g.b, PL_W1);
        hipLaunchKernelGGL((k_x3_gemm<false, false, 0, 2, true>), dim3(cdiv(cdiv(d.B, BT), 2), g.KS, d.A), dim3(512), 0, c.stream, g);
    } else
        hipLaunchKernelGGL((k_bf16_gemm<false, false>), dim3(cdiv(d.B, BT) * cdiv(d.H, BT), g.KS, d.A), dim3(256), 0, c.stream, g);
    HIP_LAUNCH_CHECK("k_bf16_gemm<fc1>");
    return 0;
}

int launch_fc11_bf16(const Ctx& c, const float* params, const float* x, int64_t xs, float* x_rec, int need_grad, int which) {
    const mmvae_dims& d = c.d;
    const Layout& L = c.lay;
    const int NS = L.sp.ks_gd10;
    // forward for gradients without x_rec (the train step, mmvae_forward(need_grad) for backward): one fused kernel, and
    // the call for d(d10) (which & 2) has nothing left to do; with x_rec wanted (or MMVAE_TUNE_FC11_ZG_OFF) two kernels
    const bool fused = need_grad && !x_rec && !c.tune(MMVAE_TUNE_FC11_ZG_OFF);
    if (split3_gemms(c)) {
        // fp32x3: only the fused train-step kernel exists in this engine (fc_dim + 1 <= 112, 128 cells per block fit the
        // loss-partial slots); everything else runs the fp32 matrix-instruction kernels (the caller falls through)
        if (!(which & 1)) return 0;
        GemmArgs g{};
        g.a = kmajor(c.ws + L.Dk[4], d.H, d.B, d.H);
        g.b = kmajor(params + c.po.o[26], d.H, d.D, d.H);
        use_planes(c, g.b, PL_W11);
        use_planes(c, g.a, PL_D10);
        g.M = d.B; g.N = d.D; g.K = d.H; g.KS = NS; g.A = d.A; g.n11 = L.n11;
        g.fo = Fc11Out{params + c.po.o[27], x, c.ws + L.DZ11, nullptr, c.ws + L.fc11_part,
                       (float)(d.A > 1 ? d.A - 1 : 1) / (float)d.B, d.B, d.D, nullptr, 0};
        if (c.x_rows) {
            if (!c.rowmap_ready) { set_error("row-indexed batches need the fp32x3 engine's fused step"); return MMVAE_E_UNSUPPORTED; }
            g.fo.xmap = reinterpret_cast<const unsigned*>(c.ws + L.rowmap);
            g.fo.x_nrec = c.x_nrows * c.x_ld;
        }
        g.fo_arm = (int64_t)d.B * d.D;
        g.fo_x_arm = xs;
        g.so = SlabOut{c.ws + L.GD10_slab, (int64_t)d.A * d.B * d.H, (int64_t)d.B * d.H, d.H, d.B, d.H};
        g.dbg = reinterpret_cast<long long*>(c.ws + c.lay.loss_scratch + 2048);
        launch_k(c, k_x3_fc11g, dim3(cdiv(d.B, 128), NS, d.A), dim3(256), 0, g);
        HIP_LAUNCH_CHECK("k_x3_fc11g");
        return 0;
    }
    if (which & 1) {
        hipError_t e = c.fwd_zeroed ? hipSuccess : hipMemsetAsync(c.ws + L.fc11_part, 0, sizeof(float) * 2 * (size_t)d.A * L.n11, c.stream);
        if (e != hipSuccess) { set_error("memset: %s", hipGetErrorString(e)); return MMVAE_E_LAUNCH; }
        if ((int64_t)cdiv(d.B, BT) * NS > L.n11) { set_error("fc11 bf16: loss partial slots"); return MMVAE_E_LAUNCH; }
        GemmArgs g{};
        g.a = kmajor(c.ws + L.Dk[4], d.H, d.B, d.H);          // d10 [B][H]
        g.a_arm = (int64_t)d.B * d.H;
        g.b = kmajor(params + c.po.o[26], d.H, d.D, d.H);      // W11 [D][H]
        g.b_arm = c.po.per_arm;
        g.bias_arm = c.po.per_arm;
        g.M = d.B; g.N = d.D; g.K = d.H; g.KS = NS; g.A = d.A; g.n11 = L.n11;
        g.fo = Fc11Out{params + c.po.o[27], x, c.ws + L.DZ11, x_rec, c.ws + L.fc11_part,
                       (float)(d.A > 1 ? d.A - 1 : 1) / (float)d.B, d.B, d.D, nullptr, 0};
        if (c.x_rows) {
            if (!fused || !c.rowmap_ready) { set_error("row-indexed batches need the fused fc11 kernel of a training step"); return MMVAE_E_UNSUPPORTED; }
            g.fo.xmap = reinterpret_cast<const unsigned*>(c.ws + L.rowmap);
            g.fo.x_nrec = c.x_nrows * c.x_ld;
        }
        g.fo_arm = (int64_t)d.B * d.D;
        g.fo_x_arm = xs;
        if (fused) {   // train step: d(d10) comes out of the same launch (which & 2 is then a no-op)
            g.so = SlabOut{c.ws + L.GD10_slab, (int64_t)d.A * d.B * d.H, (int64_t)d.B * d.H, d.H, d.B, d.H};
            if (c.x16) {   // bf16 storage: x from its bf16 copy, dZ11 written as bf16 (dW11 below reads it that way)
                g.fo.x = reinterpret_cast<const float*>(c.x16);
                g.fo_arm = (int64_t)d.B * d.D / 2;          // (arm stride of dZ11 in floats: B * D two-byte elements)
                c.dz16 = true;
                if (bf16_narrow_planes(c) && KT == 64) {   // W11 from slice 0 of its planes (launch_x3_planes): [rup(D, 128)][128] bf16
                    const PlaneGeom pg = plane_geom(c, PL_W11);
                    g.b = kmajor(reinterpret_cast<const float*>(c.ws + pg.ws_off), pg.Cp, pg.Rp, pg.Cp);
                    g.b.src16 = 1;
                    g.b_arm = 3 * (int64_t)pg.Rp * pg.Cp / 2;       // arm stride in FLOATS of the pointer arithmetic (planes: 2-byte elements)
                    launch_k(c, k_bf16_fc11g<true, true>, dim3(cdiv(d.B, BT), NS, d.A), dim3(256), 0, g);
                } else
                    launch_k(c, k_bf16_fc11g<true>, dim3(cdiv(d.B, BT), NS, d.A), dim3(256), 0, g);
            } else
                launch_k(c, k_bf16_fc11g<false>, dim3(cdiv(d.B, BT), NS, d.A), dim3(256), 0, g);
            HIP_LAUNCH_CHECK("k_bf16_fc11g");
        } else {
            hipLaunchKernelGGL(k_bf16_fc11, dim3(cdiv(d.B, BT), NS, d.A), dim3(256), 0, c.stream, g);
            HIP_LAUNCH_CHECK("k_bf16_fc11");
        }
    }
    if (need_grad && (which & 2) && !fused) {
        GemmArgs g{};
        g.a = kmajor(c.ws + L.DZ11, d.D, d.B, d.D);            // dZ11 [B][D], k = gene
        g.a_arm = (int64_t)d.B * d.D;
        g.b = kminor(params + c.po.o[26], d.H, d.H, d.D);      // W11 [D][H] read as B[n = h][k = j]
        g.b_arm = c.po.per_arm;
        g.M = d.B; g.N = d.H; g.K = d.D; g.KS = NS; g.A = d.A;
        g.so = SlabOut{c.ws + L.GD10_slab, (int64_t)d.A * d.B * d.H, (int64_t)d.B * d.H, d.H, d.B, d.H};
        if (split3_gemms(c)) {
            use_planes(c, g.b, PL_W11);      // (its bias column is row fc_dim of the transposed operand: beyond N, never stored)
            hipLaunchKernelGGL((k_x3_gemm<false, true, 0, 2, true>), dim3(cdiv(cdiv(d.B, BT), 2), NS, d.A), dim3(512), 0, c.stream, g);
        } else
            hipLaunchKernelGGL((k_bf16_gemm<false, true>), dim3(cdiv(d.B, BT) * cdiv(d.H, BT), NS, d.A), dim3(256), 0, c.stream, g);
        HIP_LAUNCH_CHECK("k_bf16_gemm<gd10>");
    }
    return 0;
}

int launch_dw_big_bf16(const Ctx& c, const float* x, int64_t xs, int which) {
    const mmvae_dims& d = c.d;
    const Layout& L = c.lay;
    const bool use_mask = c.h.training && c.h.x_drop > 0.f;
    if (which & 1) {   // dW1[h][d] = sum_b dZ1[b][h] x~[b][d]
        GemmArgs g{};
        g.a = kminor(c.ws + L.DZ[1], d.H, d.H, d.B);
        g.a_arm = (int64_t)d.B * d.H;
        g.b = kminor(x, d.D, d.D, d.B);
        g.b_arm = xs;
        if (use_mask) { g.b.bits = reinterpret_cast<const uint32_t*>(c.ws + L.xbits); g.b.wpr = cdiv(d.D, 32); g.b_bits_arm = (int64_t)d.B * g.b.wpr; }
        g.M = d.H; g.N = d.D; g.K = d.B; g.KS = L.sp.ks_dw; g.A = d.A;
        g.dbg = reinterpret_cast<long long*>(c.ws + c.lay.loss_scratch + 2048);
        g.so = SlabOut{c.ws + L.dw1_slab, (int64_t)d.A * d.H * d.D, (int64_t)d.H * d.D, d.D, d.H, d.D};
        if (c.x_rows) {
            if (!c.rowmap_ready) { set_error("row-indexed batches need the fused step's head launch"); return MMVAE_E_UNSUPPORTED; }
            g.b.rowmap = reinterpret_cast<const unsigned*>(c.ws + L.rowmap); g.b.nrec = c.x_nrows * c.x_ld; g.b.map_n = d.B;
            if (split3_gemms(c)) {
                use_planes(c, g.a, PL_DZ1);
                hipLaunchKernelGGL((k_x3_gemm<true, true, 0, 1, true, 2>), dim3(cdiv(cdiv(d.D, BT), 2), g.KS, d.A), dim3(512), 0, c.stream, g);
            } else if (c.x16) {
                g.b.ptr = reinterpret_cast<const float*>(c.x16); g.b.src16 = 1;
                if (bf16_narrow_planes(c) && bn_apply_writes_planes(c)) {   // dZ1 from slice 0 of its planes (k_bn_bwd_apply): [rup(B, 256)][128] bf16
                    const PlaneGeom pg = plane_geom(c, PL_DZ1);
                    g.a.ptr = reinterpret_cast<const float*>(c.ws + pg.ws_off); g.a.ld = pg.Cp; g.a.src16 = 1;
                    g.a_arm = 3 * (int64_t)pg.Rp * pg.Cp / 2;
                    hipLaunchKernelGGL((k_bf16_gemm<true, true, 0, 2, 3>), dim3(cdiv(d.H, BT) * cdiv(d.D, BT), g.KS, d.A), dim3(256), 0, c.stream, g);
                } else
                    hipLaunchKernelGGL((k_bf16_gemm<true, true, 0, 2, 2>), dim3(cdiv(d.H, BT) * cdiv(d.D, BT), g.KS, d.A), dim3(256), 0, c.stream, g);
            } else
                hipLaunchKernelGGL((k_bf16_gemm<true, true, 0, 2>), dim3(cdiv(d.H, BT) * cdiv(d.D, BT), g.KS, d.A), dim3(256), 0, c.stream, g);
        } else if (split3_gemms(c)) {   // one tile high: the two tiles of a block share the dZ1 tile
            use_planes(c, g.a, PL_DZ1);
            hipLaunchKernelGGL((k_x3_gemm<true, true, 0, 1, true>), dim3(cdiv(cdiv(d.D, BT), 2), g.KS, d.A), dim3(512), 0, c.stream, g);
        } else
            hipLaunchKernelGGL((k_bf16_gemm<true, true>), dim3(cdiv(d.H, BT) * cdiv(d.D, BT), g.KS, d.A), dim3(256), 0, c.stream, g);
        HIP_LAUNCH_CHECK("k_bf16_gemm<dW1>");
    }
    if (which & 2) {   // [dW11 | db11][j][h] = sum_b dZ11[b][j] [d10 | 1][b][h]
        GemmArgs g{};
        g.a = kminor(c.ws + L.DZ11, d.D, d.D, d.B);
        g.a_arm = (int64_t)d.B * d.D;
        g.b = kminor(c.ws + L.Dk[4], d.H, d.H, d.B);
        g.b.ones_row = d.H;                                      // logical row H (not in memory) reads 1: the bias gradient
        g.b_arm = (int64_t)d.B * d.H;
        g.M = d.D; g.N = d.H + 1; g.K = d.B; g.KS = L.sp.ks_dw11; g.A = d.A;
        g.dbg = reinterpret_cast<long long*>(c.ws + c.lay.loss_scratch + 2048);
        g.so = SlabOut{c.ws + L.dw11_slab, (int64_t)d.A * d.D * DW11_LD, (int64_t)d.D * DW11_LD, DW11_LD, d.D, d.H + 1};
        if (split3_gemms(c)) {   // one tile wide: the two tiles of a block share the [d10 | 1] tile
            use_planes(c, g.b, PL_D10);
            hipLaunchKernelGGL((k_x3_gemm<true, true, 0, 2, true>), dim3(cdiv(cdiv(d.D, BT), 2), g.KS, d.A), dim3(512), 0, c.stream, g);
        } else if (c.dz16) {        // bf16 storage: the fused fc11 kernel of this step wrote dZ11 as bf16
            g.a.src16 = 1;
            g.a_arm = (int64_t)d.B * d.D / 2;
            hipLaunchKernelGGL((k_bf16_gemm<true, true, 0, 0, 1>), dim3(cdiv(d.D, BT) * cdiv(d.H + 1, BT), g.KS, d.A), dim3(256), 0, c.stream, g);
        } else
            hipLaunchKernelGGL((k_bf16_gemm<true, true>), dim3(cdiv(d.D, BT) * cdiv(d.H + 1, BT), g.KS, d.A), dim3(256), 0, c.stream, g);
        HIP_LAUNCH_CHECK("k_bf16_gemm<dW11>");
    }
    return 0;
}

int launch_dw_small_x3(const Ctx& c, const TnDescs& ts, int nsel) {
    const int KS = c.lay.sp.ks_small;
    hipLaunchKernelGGL(k_x3_small, dim3(cdiv(nsel * c.d.A, 2), KS), dim3(512), 0, c.stream, ts, nsel, c.d.A, c.d.B, KS);
    HIP_LAUNCH_CHECK("k_x3_small");
    return 0;
}

// C[M][ld] = act((A[M][K] . W[N][K]^T) * scale + shift): the augmenter's layers with bf16 operands (augment.hip)
int launch_presplit_one(hipStream_t s, const float* src, int64_t ld, int R, int C, int Rp, int Cp, unsigned short* dst) {
    const SplitJob j{src, ld, 0, R, C, Rp, Cp, -1, dst, 0, nullptr, 0};
    return launch_presplit(s, 1, &j, 1);
}

// sum of the split-K slabs of an augmenter layer + its folded BatchNorm / ReLU epilogue: out[m][n] = act(sum_ks slab[ks][m][n] *
// scale[n] + shift[n]) for n < N, zero for the padding columns N .. ncols - 1.  One float4 per thread (ld, ncols multiples of 4).
__global__ __launch_bounds__(256) void k_aug_slab_epi(const float* __restrict__ slab, int64_t ks_stride, int KS, const float* __restrict__ scale,
                                                      const float* __restrict__ shift, float* __restrict__ out, int ld, int M, int N,
                                                      int ncols, int relu, int affine) {
    const int c4n = ncols >> 2;
    const int64_t n = (int64_t)M * c4n;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int m = (int)(i / c4n), c = (int)(i - (int64_t)m * c4n) * 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int k = 0; k < KS; ++k) {   // (KS <= 4: the loads of all slabs are independent and issue together)
            const float4 t = *reinterpret_cast<const float4*>(slab + k * ks_stride + (int64_t)m * ld + c);
            v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
        }
        float r[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int col = c + e;
            const bool real = col < N;
            float y = r[e] * ((affine && real) ? scale[col] : 1.f) + ((affine && real) ? shift[col] : 0.f);
            if (relu) y = fmaxf(y, 0.f);
            r[e] = real ? y : 0.f;
        }
        *reinterpret_cast<float4*>(out + (int64_t)m * ld + c) = make_float4(r[0], r[1], r[2], r[3]);
    }
}

int launch_bf16_affine(hipStream_t s, bool relu, bool affine, const float* A, int lda, int M, const float* W, int ldw, int N,
                       int Kpad, const float* sc, const float* sh, float* C, int ldc, int ncols, int split3,
                       const unsigned short* w_planes, int Np, int Kp, float* scratch, int64_t scratch_floats) {
    GemmArgs g{};
    g.a = kmajor(A, lda, M, Kpad);        // rows are zero-padded to Kpad = pad4(K) floats on both sides
    g.b = kmajor(W, ldw, N, Kpad);
    g.M = M; g.N = ncols; g.K = Kpad; g.KS = 1; g.A = 1;
    g.ao = AffineOut{sc, sh, C, ldc, M, N, relu ? 1 : 0, affine ? 1 : 0};
    if (split3 && w_planes && Np >= cdiv(ncols, BT) * BT) {
        // fp32x3 with the weight's slice planes (written at pack time): the two tiles of a block are m neighbours of one
        // n tile and share its weight tile, copied from the planes; the activations are split on their way into LDS
        g.b.pl = w_planes; g.b.pl_plane = (int64_t)Np * Kp; g.b.pl_arm = 0; g.b.pl_ld = Kp;
        const int tiles = cdiv(M, BT) * cdiv(ncols, BT);
        // A layer with fewer tile pairs than CUs and a long K (the first layer at the benchmark shape: 5000 x 1000 x 5000 = 160
        // blocks on 256 CUs, 362 of the forward's 1 580 us): split K over KS blocks per pair -- the count in 2 .. 4 that needs
        // the fewest rounds of the chip per unit of K -- into slabs in the caller's scratch, and a small pass sums them and
        // applies the epilogue.
        {
            constexpr int N_CUS = 256;   // MI355X
            const int pairs = cdiv(cdiv(M, BT), 2) * cdiv(ncols, BT);
            int ks_best = 1;
            double cost_best = (double)cdiv(pairs, N_CUS);
            for (int ks = 2; ks <= 4; ++ks) {
                const double cost = (double)cdiv(pairs * ks, N_CUS) / ks;
                if (cost < 0.8 * cost_best && Kpad / ks >= 32 * Eng<3>::KT && (int64_t)ks * M * ldc <= scratch_floats) { ks_best = ks; cost_best = cost; }
            }
            if (scratch && ks_best > 1 && tiles > 192 && (ldc & 3) == 0 && (ncols & 3) == 0) {
                g.KS = ks_best;
                g.so = SlabOut{scratch, (int64_t)M * ldc, 0, ldc, M, N < ncols ? N : ncols};
                hipLaunchKernelGGL((k_x3_gemm<false, false, 0, 2, true>), dim3(pairs, ks_best, 1), dim3(512), 0, s, g);
                HIP_LAUNCH_CHECK("k_x3_gemm<affine, split K>");
                const int64_t items = (int64_t)M * (ncols >> 2);
                hipLaunchKernelGGL(k_aug_slab_epi, dim3((unsigned)imin64(4096, cdiv64(items, 256))), dim3(256), 0, s, scratch, (int64_t)M * ldc,
                                   ks_best, sc, sh, C, ldc, M, N, ncols, relu ? 1 : 0, affine ? 1 : 0);
                HIP_LAUNCH_CHECK("k_aug_slab_epi");
                return 0;
            }
        }
        // fewer tile pairs than half the CUs: one tile per block, the groups split K (trunk layers 1000 -> 500, 500 -> 500,
        // 500 -> 100 at M = 5000: 77 -> 51, 45 -> 32, 44 -> 31 us; at 320 tiles the lost sharing of the weight tile costs more:
        // 360 -> 385, 82 -> 92, 50 -> 62 us)
        if (tiles <= 192 && Kpad >= 4 * Eng<3>::KT)
            hipLaunchKernelGGL((k_x3_gemm<false, false, 1, 3, true>), dim3(tiles, 1, 1), dim3(512), 0, s, g);
        else
            hipLaunchKernelGGL((k_x3_gemm<false, false, 1, 2, true>), dim3(cdiv(cdiv(M, BT), 2) * cdiv(ncols, BT), 1, 1), dim3(512), 0, s, g);
    } else if (split3)
        hipLaunchKernelGGL((k_x3_gemm<false, false, 1, 0>), dim3(cdiv(cdiv(M, BT) * cdiv(ncols, BT), 2), 1, 1), dim3(512), 0, s, g);
    else {
        // bf16 operands: one tile per 256-thread block, two blocks per CU; the same split of a long K for layers that leave
        // slots of the chip empty (the first layer: 320 tiles on 512 slots)
        const int tiles = cdiv(M, BT) * cdiv(ncols, BT);
        constexpr int SLOTS = 512;
        int ks_best = 1;
        double cost_best = (double)cdiv(tiles, SLOTS);
        for (int ks = 2; ks <= 4; ++ks) {
            const double cost = (double)cdiv(tiles * ks, SLOTS) / ks;
            if (cost < 0.8 * cost_best && Kpad / ks >= 16 * KT && (int64_t)ks * M * ldc <= scratch_floats) { ks_best = ks; cost_best = cost; }
        }
        if (scratch && ks_best > 1 && (ldc & 3) == 0 && (ncols & 3) == 0) {
            g.KS = ks_best;
            g.so = SlabOut{scratch, (int64_t)M * ldc, 0, ldc, M, N < ncols ? N : ncols};
            hipLaunchKernelGGL((k_bf16_gemm<false, false>), dim3(tiles, ks_best, 1), dim3(256), 0, s, g);
            HIP_LAUNCH_CHECK("k_bf16_gemm<affine, split K>");
            const int64_t items = (int64_t)M * (ncols >> 2);
            hipLaunchKernelGGL(k_aug_slab_epi, dim3((unsigned)imin64(4096, cdiv64(items, 256))), dim3(256), 0, s, scratch, (int64_t)M * ldc,
                               ks_best, sc, sh, C, ldc, M, N, ncols, relu ? 1 : 0, affine ? 1 : 0);
            HIP_LAUNCH_CHECK("k_aug_slab_epi");
            return 0;
        }
        hipLaunchKernelGGL((k_bf16_gemm<false, false, 1>), dim3(tiles, 1, 1), dim3(256), 0, s, g);
    }
    HIP_LAUNCH_CHECK("k_bf16_gemm<affine>");
    return 0;
}

}  // namespace mmvae
