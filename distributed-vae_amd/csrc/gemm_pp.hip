// Planes x planes GEMM: C = epi(A B^T) with BOTH operands already in bf16 slice planes, tiled for LDS-DMA.
//
// The augmenter's layers (augment.hip; mmidas/augmentation/udagan.py:284-329) are plain large GEMMs.  On the tile engine of
// gemm_bf16.hip every column tile split the same fp32 activation tile into its three bf16 slices again (eleven VALU
// instructions per pair of elements), and the kernel needed its lock-step ping-pong to hide that staging behind the matrix
// phase: 0.32 of the engine's ceiling.  Here a layer's epilogue WRITES the next layer's operand as slice planes, so staging
// is a copy with no registers and no VALU work (global_load_lds_dwordx4), all eight waves of a workgroup multiply all the
// time, and the block tile is 256 x 256 (a CU takes in about 20 B per cycle from L2: 128 x 128 tiles of three planes need 32).
//
// "Tiled planes" (TP) of a matrix X [R][K]: NP planes (3: the exact slices x1 + x2 + x3 of the fp32x3 engine, 1: X rounded to
// bf16), each stored [KT = ceil(K / 16)][Rp = R rounded up to 256][16] bf16 -- the 16 k of one K step of one row are 32
// contiguous bytes and a (row block, K step) tile is ONE contiguous piece of memory, which is what a lane-linear LDS-DMA wants.
// Zero for k >= K (weights: also rows >= R); activation rows >= R are never written and never read back as data (a row of A
// only reaches its own row of C).
//
//   k_pp_gemm<NP, WM, WN, WAVES_M, WAVES_N>   512 threads = 8 waves as WAVES_M x WAVES_N, wave tile 32 WM x 32 WN; stage =
//       one K step (16 k) of both operands, all planes, in an LDS ring of NSTAGE buffers filled by LDS-DMA NSTAGE - 1 steps
//       ahead; ONE raw s_barrier per K step, counted s_waitcnt vmcnt (the DMAs stay in flight across the barrier).
//       Epilogue: scale / shift per column + ReLU, then fp32 row-major and / or the TP planes of the next layer.  With a K split the
//       KS workgroups of a tile combine their accumulators INSIDE the launch (all-to-all over write-through partial slots).
//   k_tp_from_f32   fp32 row-major -> TP planes (the first layer's input, the latent block's output)
#include "common.hpp"
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

typedef __bf16 bf16x8q __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4q __attribute__((ext_vector_type(4)));

struct PPArgs {
    const unsigned short* a; int64_t a_plane; int a_Rp;     // A planes [NP][KT][a_Rp][16]
    const unsigned* a_map;                                  // or null: logical row m of A is row a_map[m] of ROW-MAJOR planes [NP][a_Rp][KT][16]
                                                            // (k_rp_from_f32: a row's K steps are contiguous; tiles_m x BM entries)
    const unsigned short* b; int64_t b_plane; int b_Rp;     // B planes [NP][KT][b_Rp][16]
    int M, N, KT;                                           // real rows of A / of B, K steps
    int KT_a;                                               // row-mapped A: K steps per row of the row-major planes
    int KS;                                                 // split of the K steps: workgroup = tile x KS + part
    const float* scale; const float* shift;                 // per column n (affine), or unused
    int affine, relu;
    float* out32; int64_t ld32; int ncols32;                // fp32 out [M][ld32], columns < ncols32 written (n >= N: zero), or null
    unsigned short* outp; int64_t outp_plane; int outp_Rp, outp_KT;   // TP planes of the output, or null
    float* part;                                            // KS > 1: [workgroup][BM x BN] partial accumulators
    unsigned* flags;                                        // KS > 1: [workgroup], zero before the launch; 1 = that workgroup's partial is written
    int tiles_m, tiles_n;
};

#ifndef PP_ABL
#define PP_ABL 0      // timing ablations (diagnostic builds only; results wrong): 1 no MFMAs, 2 no LDS-DMA in the loop, 4 no fragment reads
#endif
// s_waitcnt lgkmcnt(CNT) with the fragments it makes valid as operands
template <int CNT, int NP>
__device__ __forceinline__ void wait_lgkm_tie(bf16x8q (&a)[NP]) {
    if constexpr (NP == 3) asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]) : "n"(CNT));
    else asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(a[0]) : "n"(CNT));
}
template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }

// value v of output element (m, n) and its right neighbour's (lane + 1) into the output planes: the even lane writes the pair
template <int NP>
__device__ __forceinline__ void tp_store_pair(const PPArgs& g, int m, int n, float v, float nb, bool even) {
    if (!even || m >= g.M || (n >> 4) >= g.outp_KT) return;
    const int64_t e = ((int64_t)(n >> 4) * g.outp_Rp + m) * 16 + (n & 15);
    if (NP == 3) {
        unsigned w[3];
        split3(v, nb, w);
#pragma unroll
        for (int p = 0; p < 3; ++p) *reinterpret_cast<unsigned*>(g.outp + p * g.outp_plane + e) = w[p];
    } else {
        *reinterpret_cast<unsigned*>(g.outp + e) = cvt_pk_bf16(v, nb);
    }
}

constexpr unsigned PP_MAX_POLLS = 1u << 22;     // x >= 64 clocks of s_sleep: seconds

template <int NP, int WM, int WN, int WAVES_M, int WAVES_N>
__global__ __launch_bounds__(512) void k_pp_gemm(const PPArgs g_in) {
    static_assert(WAVES_M * WAVES_N == 8, "eight waves");
    constexpr int BM = 32 * WM * WAVES_M, BN = 32 * WN * WAVES_N;
    static_assert(BM <= 256 && BN <= 256 && BM % 32 == 0 && BN % 64 == 0, "tile");
    constexpr int A_PLANE = BM * 8, B_PLANE = BN * 8;              // dwords: rows x 32 bytes
    constexpr int STAGE = NP * (A_PLANE + B_PLANE);
    constexpr int NSTAGE = (160 * 1024 / 4 - 4) / STAGE >= 6 ? 6 : (160 * 1024 / 4 - 4) / STAGE;   // 256 x 256: 3 (three planes) / 6 (one)
    static_assert(NSTAGE >= 3, "at least three stages");
    constexpr int PER = 2 * NP;                                    // DMA instructions per wave and stage
    extern __shared__ __attribute__((aligned(16))) unsigned lds[];  // the ONLY LDS object: [NSTAGE][STAGE] + one word (poll result)
    const PPArgs g = g_in;
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hh = lane >> 5;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6), wm = wv / WAVES_N, wn = wv % WAVES_N;
    // XCD-aware order: workgroup i runs on XCD i % 8; an XCD walks a contiguous range of (tile, part) pairs, tile-major with n fastest:
    // the parts of a tile are neighbours (the combine below waits only for workgroups that are dispatched next to this one),
    // workgroups with the same part of neighbouring tiles run in step and share their A / B panels in the XCD's L2
    const int nwg = gridDim.x;
    int bi = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7, x = bi & 7;
        bi = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bi >> 3);
    }
    const int tile = bi / g.KS, part = bi - tile * g.KS;
    const int tm = tile / g.tiles_n, tn = tile % g.tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    const int kt0 = (int)(((int64_t)part * g.KT) / g.KS), kt1 = (int)(((int64_t)(part + 1) * g.KT) / g.KS);
    const int nk = kt1 - kt0;
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) void*)(&lds[0]);
    // a wave's chunk of a tile plane: rows x 32 B / 8 waves; lane l moves bytes [16 l, 16 l + 16) of it
    constexpr int A_CHUNK = BM * 4, B_CHUNK = BN * 4;              // bytes
    // LDS image of a tile plane: [row][32 B], the two 16-byte halves of a row SWAPPED in rows with bit 3 set -- a ds_read_b128 serves
    // 16 lanes (16 rows, one half each) per pass and their sixteen 16-byte slots must differ modulo 256 B: rows r and r + 8 would
    // collide (2 r mod 16).  The DMA is lane-linear in LDS, so the swap is applied to the SOURCE address.
    // (a wave's chunk starts at row wv BM / 8 of the tile: 20 rows for the 160-row tile, so the row's bit 3 is not the lane's bit 4 there)
    const unsigned swz_a = (unsigned)((((lane & 1) ^ (((wv * (BM / 8) + (lane >> 1)) >> 3) & 1)) * 16) + (lane >> 1) * 32);
    const unsigned swz_b = (unsigned)((((lane & 1) ^ (((wv * (BN / 8) + (lane >> 1)) >> 3) & 1)) * 16) + (lane >> 1) * 32);
    unsigned voff_a = (unsigned)(wv * A_CHUNK) + swz_a;
    const unsigned voff_b = (unsigned)(wv * B_CHUNK) + swz_b;
    if (g.a_map) {
        // row-mapped A (the first layer reads the batch's rows out of the resident matrix' planes): the lane's row of the tile lies
        // at a_map[...] x 32 bytes of the K step's slab; the LDS side of the DMA stays lane-linear
        // (row-major planes: the 32-byte pieces a tile takes from scattered rows are a quarter of a 128-byte line each, and the next
        // three K steps find the rest of the line in L2; out of K-step-major planes every piece cost a line of its own from HBM)
        const int rl = wv * (BM / 8) + (lane >> 1);
        const unsigned row = (BM == 256 || lane < BM / 4) ? g.a_map[m0 + rl] : 0u;
        voff_a = row * (unsigned)(g.KT_a * 32) + (unsigned)(((lane & 1) ^ ((rl >> 3) & 1)) * 16);
    }
    auto dma = [&](int kt, int st) __attribute__((always_inline)) {
        const unsigned short* sa = g.a_map ? g.a + (int64_t)kt * 16 : g.a + ((int64_t)kt * g.a_Rp + m0) * 16;
        const unsigned short* sb = g.b + ((int64_t)kt * g.b_Rp + n0) * 16;
        const unsigned la = lds_base + 4u * (unsigned)(st * STAGE) + (unsigned)(wv * A_CHUNK);
        const unsigned lb = lds_base + 4u * (unsigned)(st * STAGE + NP * A_PLANE) + (unsigned)(wv * B_CHUNK);
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const unsigned short* src = sa + p * g.a_plane;
            const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane((int)(la + 4u * (unsigned)(p * A_PLANE)));   // (wave-uniform: an SGPR for M0)
            if (BM == 256 || lane < BM / 4)
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" :: "s"(dst), "v"(voff_a), "s"(src) : "memory");
        }
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const unsigned short* src = sb + p * g.b_plane;
            const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane((int)(lb + 4u * (unsigned)(p * B_PLANE)));
            if (BN == 256 || lane < BN / 4)
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" :: "s"(dst), "v"(voff_b), "s"(src) : "memory");
        }
    };
    f32x16 acc[WM][WN];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j) acc[i][j] = zero16();
#pragma unroll
    for (int s = 0; s < NSTAGE - 1; ++s)
        if (s < nk) dma(kt0 + s, s);
    const int hsw = (hh ^ ((l31 >> 3) & 1)) * 4;      // this lane's half of its row in the swizzled image (dwords)
    const int a_off = (wm * 32 * WM + l31) * 8 + hsw, b_off = NP * A_PLANE + (wn * 32 * WN + l31) * 8 + hsw;
    auto lds_frag = [&](const unsigned* p) __attribute__((always_inline)) {
        return __builtin_bit_cast(bf16x8q, *reinterpret_cast<const u32x4q*>(p));
    };
    auto mma = [&](f32x16& c, const bf16x8q (&a)[NP], const bf16x8q (&b)[NP]) __attribute__((always_inline)) {
        if constexpr ((PP_ABL & 1) != 0) { asm volatile("" :: "v"(a[0]), "v"(b[0]), "v"(a[NP - 1]), "v"(b[NP - 1])); return; }
        if constexpr (NP == 3) {   // six of the nine slice products, the small ones first (gemm_bf16.hip Eng<3>)
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], c, 0, 0, 0);
        }
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], c, 0, 0, 0);
    };
    auto barrier = [&]() __attribute__((always_inline)) {
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
    auto wait_behind = [&](int behind) __attribute__((always_inline)) {
        // a stage has landed once at most the stages issued behind it are outstanding (a wave's DMAs retire in issue order)
        if (behind >= 4) wait_vm<4 * PER>();
        else if (behind == 3) wait_vm<3 * PER>();
        else if (behind == 2) wait_vm<2 * PER>();
        else if (behind == 1) wait_vm<PER>();
        else wait_vm<0>();
    };
    {
        // all eight waves in step: one barrier per K step, the fragments read behind it
        int st = 0;
        for (int it = 0; it < nk; ++it) {
            wait_behind(min(NSTAGE - 2, nk - 1 - it));
            barrier();     // every wave's share of stage `it` is in LDS; every wave is done reading stage it - 1
            if (!(PP_ABL & 2) && it + NSTAGE - 1 < nk) dma(kt0 + it + NSTAGE - 1, st == 0 ? NSTAGE - 1 : st - 1);
            // The fragment reads are inline assembly with hand-counted waits: left to hipcc, each read sinks to its first use and is
            // waited for at once (nine exposed LDS latencies per step in the ISA), and with all eighteen requested up front its
            // wait-count pass emits lgkmcnt(0) before the first MFMA (the counter has four bits).  B and the first row tiles of A
            // are requested (at most 15 reads), a row tile's MFMAs wait only for the reads in front of them, and the remaining row
            // tiles are requested behind the first MFMAs.
            const unsigned sbase = lds_base + 4u * (unsigned)(st * STAGE);
            const unsigned va = sbase + 4u * (unsigned)a_off, vb = sbase + 4u * (unsigned)b_off;
            bf16x8q bf[WN][NP], af[WM][NP];
#pragma unroll
            for (int j = 0; j < WN; ++j)
#pragma unroll
                for (int p = 0; p < NP; ++p)
                    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(bf[j][p]) : "v"(vb), "n"((p * B_PLANE + j * 256) * 4));
            constexpr int PRE = (15 - WN * NP) / NP < WM ? (15 - WN * NP) / NP : WM;     // row tiles requested up front
            static_assert(PRE >= 1, "fragment reads");
#pragma unroll
            for (int i = 0; i < PRE; ++i)
#pragma unroll
                for (int p = 0; p < NP; ++p)
                    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(af[i][p]) : "v"(va), "n"((p * A_PLANE + i * 256) * 4));
#pragma unroll
            for (int i = 0; i < WM; ++i) {
                constexpr int LATE = WM - PRE;                    // row tiles requested behind the MFMAs of tiles 0 .. LATE - 1
                const int issued = i < LATE ? PRE + i : WM;       // row tiles requested when tile i's MFMAs start
                const int after = (issued - 1 - i) * NP;          // reads behind tile i's
                // (the fragment registers are operands of the wait: no MFMA that uses them moves in front of it)
                if (after >= 6) wait_lgkm_tie<6, NP>(af[i]);
                else if (after == 4) wait_lgkm_tie<4, NP>(af[i]);
                else if (after == 3) wait_lgkm_tie<3, NP>(af[i]);
                else if (after == 2) wait_lgkm_tie<2, NP>(af[i]);
                else if (after == 1) wait_lgkm_tie<1, NP>(af[i]);
                else wait_lgkm_tie<0, NP>(af[i]);
                if (i == 0) {
#pragma unroll
                    for (int j = 0; j < WN; ++j)
#pragma unroll
                        for (int p = 0; p < NP; ++p) asm volatile("" : "+v"(bf[j][p]));
                }
#pragma unroll
                for (int j = 0; j < WN; ++j) mma(acc[i][j], af[i], bf[j]);
                __builtin_amdgcn_sched_barrier(0);                // (hipcc otherwise moves the later tiles' waits in front of these MFMAs)
                if (i < LATE) {
#pragma unroll
                    for (int p = 0; p < NP; ++p)
                        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(af[PRE + i][p]) : "v"(va), "n"((p * A_PLANE + (PRE + i) * 256) * 4));
                }
            }
            st = st + 1 == NSTAGE ? 0 : st + 1;
        }
    }
    // ---- K split: the KS workgroups of a tile combine their accumulators inside the launch, all-to-all.  Accumulator tile t (of
    // the WM x WN a wave holds) is FINISHED by part t % KS: every part writes the tiles it does not finish to its slot with
    // write-through (sc1) stores, drains, one lane sets the slot's flag; then it polls its partners' flags (bounded: a partial
    // that never arrives poisons the output with NaN) and adds their pieces of ITS tiles with sc1 loads.  The partners are the
    // neighbouring workgroups in dispatch order and each publishes before it waits, so the grid need not be resident as a whole.
    // slot layout: [t][v][thread] sixteen-byte pieces -- a wave's store / load is one contiguous KB
    constexpr int NT = WM * WN;
    if (g.KS > 1) {
        constexpr int PART_FLOATS = BM * BN;
        const __amdgpu_buffer_rsrc_t rs_mine = __builtin_amdgcn_make_buffer_rsrc(g.part + (int64_t)bi * PART_FLOATS, 0, PART_FLOATS * 4, 0x00020000);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            if (t % g.KS == part) continue;
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                u32x4q w;
#pragma unroll
                for (int e = 0; e < 4; ++e) w[e] = __float_as_uint(acc[t / WN][t % WN][4 * v + e]);
                __builtin_amdgcn_raw_buffer_store_b128(w, rs_mine, ((t * 4 + v) * 512 + tid) * 16, 0, 16);   // aux 16 = sc1
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // every storing wave drains
        __syncthreads();
        if (tid == 0) __hip_atomic_store(g.flags + bi, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (int d = 1; d < g.KS; ++d) {
            const int c = part + d < g.KS ? part + d : part + d - g.KS;      // partner part
            const int cw = tile * g.KS + c;
            if (tid == 0) {
                unsigned ok = 0;
                for (unsigned i = 0; i < PP_MAX_POLLS; ++i) {
                    if (__hip_atomic_load(g.flags + cw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 1u) { ok = 1; break; }
                    __builtin_amdgcn_s_sleep(1);
                }
                lds[NSTAGE * STAGE] = ok;
            }
            __syncthreads();
            const bool ok = *reinterpret_cast<volatile unsigned*>(&lds[NSTAGE * STAGE]) != 0;
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");     // (no instruction: every load below is sc1 and stays behind the poll)
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(g.part + (int64_t)cw * PART_FLOATS, 0, PART_FLOATS * 4, 0x00020000);
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                if (t % g.KS != part) continue;
                u32x4q w[4];
#pragma unroll
                for (int v = 0; v < 4; ++v)
                    w[v] = __builtin_bit_cast(u32x4q, __builtin_amdgcn_raw_buffer_load_b128(rs, ((t * 4 + v) * 512 + tid) * 16, 0, 16));   // sc1
#pragma unroll
                for (int v = 0; v < 4; ++v)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[t / WN][t % WN][4 * v + e] += ok ? __uint_as_float(w[v][e]) : __builtin_nanf("");
            }
            __syncthreads();     // the poll word is rewritten for the next partner
        }
    }
    // ---- epilogue (of the accumulator tiles this part finishes): register r of tile (i, j) is row m0 + 32 (WM wm + i) + acc_row(r),
    // column n0 + 32 (WN wn + j) + (lane & 31)
    const bool even = !(lane & 1);
#pragma unroll
    for (int j = 0; j < WN; ++j) {
        const int n = n0 + 32 * (WN * wn + j) + l31;
        const bool real = n < g.N;
        const float sc = (g.affine && real) ? g.scale[n] : 1.f;
        const float sh = (g.affine && real) ? g.shift[n] : 0.f;
#pragma unroll
        for (int i = 0; i < WM; ++i) {
            if (g.KS > 1 && (i * WN + j) % g.KS != part) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + 32 * (WM * wm + i) + acc_row(r, lane);
                float v = acc[i][j][r] * sc + sh;
                if (g.relu) v = fmaxf(v, 0.f);
                v = real ? v : 0.f;
                if (g.out32 && m < g.M && n < g.ncols32) g.out32[(int64_t)m * g.ld32 + n] = v;
                if (g.outp) {
                    const float nb = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xF5, 0xF, 0xF, false));
                    tp_store_pair<NP>(g, m, n, v, nb, even);
                }
            }
        }
    }
}

// fp32 [R][ld] (columns < K) -> TP planes.  A block owns 32 rows x 8 K steps (128 k): its reads are whole 512-byte row segments
// (a thread: one float4), the slices go through an LDS tile in the output order, and its writes are 1 KB runs (32 rows x 32 B of
// one K step and plane) in 16-byte pieces.  grid (ceil(R / 32), ceil(KT / 8)).
template <int NP>
__global__ __launch_bounds__(256) void k_tp_from_f32(const float* __restrict__ src, int64_t ld, int R, int K, unsigned short* __restrict__ dst,
                                                     int64_t plane, int Rp, int KT, unsigned* __restrict__ zero_words, int n_zero) {
    // (the first launch of an augmenter forward also zeroes the flag words of the forward's K-split combines)
    if (zero_words && blockIdx.x == 0 && blockIdx.y == 0)
        for (int i = threadIdx.x; i < n_zero; i += 256) zero_words[i] = 0u;
    constexpr int KSTR = 32 * 8 + 8;                       // dwords per (plane, K step) of the tile: 32 rows x 8 dwords (+ pad)
    __shared__ __attribute__((aligned(16))) unsigned tile[NP * 8 * KSTR];
    const int t = threadIdx.x, row0 = blockIdx.x * 32, kt0 = blockIdx.y * 8;
    const bool vec = (ld & 3) == 0 && (reinterpret_cast<uintptr_t>(src) & 15) == 0;
    const int c = t & 31, k = kt0 * 16 + 4 * c;
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
        const int rl = pass * 8 + (t >> 5), row = row0 + rl;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (row < R) {
            const float* p = src + (int64_t)row * ld + k;
            if (vec && k + 4 <= K) v = *reinterpret_cast<const float4*>(p);
            else {
                if (k < K) v.x = p[0];
                if (k + 1 < K) v.y = p[1];
                if (k + 2 < K) v.z = p[2];
                if (k + 3 < K) v.w = p[3];
            }
        }
        unsigned w0[3], w1[3];
        if (NP == 3) { split3(v.x, v.y, w0); split3(v.z, v.w, w1); }
        else { w0[0] = cvt_pk_bf16(v.x, v.y); w1[0] = cvt_pk_bf16(v.z, v.w); }
#pragma unroll
        for (int pl = 0; pl < NP; ++pl)
            *reinterpret_cast<uint2*>(&tile[(pl * 8 + (c >> 2)) * KSTR + rl * 8 + (c & 3) * 2]) = make_uint2(w0[pl], w1[pl]);
    }
    __syncthreads();
#pragma unroll
    for (int pass = 0; pass < NP * 2; ++pass) {
        const int id = t + 256 * pass;                     // NP x 8 K steps x 64 sixteen-byte pieces
        const int pl = id >> 9, ktl = (id >> 6) & 7, r16 = id & 63, rl = r16 >> 1, half = r16 & 1;
        if (kt0 + ktl < KT && row0 + rl < R) {
            const u32x4q o = *reinterpret_cast<const u32x4q*>(&tile[(pl * 8 + ktl) * KSTR + rl * 8 + half * 4]);
            *reinterpret_cast<u32x4q*>(dst + pl * plane + ((int64_t)(kt0 + ktl) * Rp + row0 + rl) * 16 + half * 8) = o;
        }
    }
}

// fp32 [R][ld] (columns < K) -> ROW-MAJOR slice planes [NP][Rp][KT][16] (the layout of a row-mapped A operand): a thread owns four
// consecutive k of a row.  Elements with k >= K and rows >= R are not written (the caller zero-fills once).
template <int NP>
__global__ __launch_bounds__(256) void k_rp_from_f32(const float* __restrict__ src, int64_t ld, int R, int K, unsigned short* __restrict__ dst,
                                                     int64_t plane, int KT) {
    const int q4 = K >> 2;                                  // K % 4 == 0
    const int64_t n = (int64_t)R * q4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int64_t row = i / q4;
        const int k = (int)(i - row * q4) * 4;
        const float4 v = *reinterpret_cast<const float4*>(src + row * ld + k);
        unsigned w0[3], w1[3];
        if (NP == 3) { split3(v.x, v.y, w0); split3(v.z, v.w, w1); }
        else { w0[0] = cvt_pk_bf16(v.x, v.y); w1[0] = cvt_pk_bf16(v.z, v.w); }
#pragma unroll
        for (int pl = 0; pl < NP; ++pl)
            *reinterpret_cast<uint2*>(dst + pl * plane + (row * KT * 16 + k)) = make_uint2(w0[pl], w1[pl]);
    }
}
int launch_rp_from_f32(hipStream_t s, const float* src, int64_t ld, int R, int K, int NP, TPlanes dst) {
    if ((K & 3) || (ld & 3) || (reinterpret_cast<uintptr_t>(src) & 15)) { set_error("rp_from_f32: K, ld multiples of 4, 16-byte aligned rows"); return MMVAE_E_BADARG; }
    const unsigned blocks = (unsigned)imin64(cdiv64((int64_t)R * (K >> 2), 256), 16384);
    if (NP == 3) hipLaunchKernelGGL(k_rp_from_f32<3>, dim3(blocks), dim3(256), 0, s, src, ld, R, K, dst.p, dst.plane, dst.KT);
    else hipLaunchKernelGGL(k_rp_from_f32<1>, dim3(blocks), dim3(256), 0, s, src, ld, R, K, dst.p, dst.plane, dst.KT);
    HIP_LAUNCH_CHECK("k_rp_from_f32");
    return 0;
}

int launch_tp_from_f32(hipStream_t s, const float* src, int64_t ld, int R, int K, int NP, TPlanes dst, float* zero_flags_of_scratch) {
    const dim3 grid(cdiv(R, 32), cdiv(dst.KT, 8));
    unsigned* const zw = reinterpret_cast<unsigned*>(zero_flags_of_scratch);
    if (NP == 3) hipLaunchKernelGGL(k_tp_from_f32<3>, grid, dim3(256), 0, s, src, ld, R, K, dst.p, dst.plane, dst.Rp, dst.KT, zw, PP_FLAG_WORDS);
    else hipLaunchKernelGGL(k_tp_from_f32<1>, grid, dim3(256), 0, s, src, ld, R, K, dst.p, dst.plane, dst.Rp, dst.KT, zw, PP_FLAG_WORDS);
    HIP_LAUNCH_CHECK("k_tp_from_f32");
    return 0;
}

// the row map of a row-mapped A operand: out[i] = rows[i] clamped to the matrix, i < n; 0 for the padding entries up to n_pad
__global__ __launch_bounds__(256) void k_pp_rowmap(const int64_t* __restrict__ rows, int n, int64_t n_rows, unsigned* __restrict__ out, int n_pad,
                                                   unsigned* __restrict__ zero_words, int n_zero) {
    if (zero_words && blockIdx.x == 0)
        for (int j = threadIdx.x; j < n_zero; j += 256) zero_words[j] = 0u;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n_pad) return;
    int64_t r = i < n ? rows[i] : 0;
    r = r < 0 ? 0 : (r >= n_rows ? n_rows - 1 : r);
    out[i] = (unsigned)r;
}
int launch_pp_rowmap(hipStream_t s, const int64_t* rows, int n, int64_t n_rows, unsigned* out, int n_pad, float* zero_flags_of_scratch) {
    hipLaunchKernelGGL(k_pp_rowmap, dim3(cdiv(n_pad, 256)), dim3(256), 0, s, rows, n, n_rows, out, n_pad,
                       reinterpret_cast<unsigned*>(zero_flags_of_scratch), PP_FLAG_WORDS);
    HIP_LAUNCH_CHECK("k_pp_rowmap");
    return 0;
}

template <int NP, int WM, int WN, int WAVES_M, int WAVES_N>
static int pp_launch(hipStream_t s, PPArgs& g) {
    constexpr int BM = 32 * WM * WAVES_M, BN = 32 * WN * WAVES_N;
    constexpr int STAGE = NP * (BM + BN) * 8;
    constexpr int NSTAGE = (160 * 1024 / 4 - 4) / STAGE >= 6 ? 6 : (160 * 1024 / 4 - 4) / STAGE;
    constexpr size_t SHM = (size_t)NSTAGE * STAGE * 4 + 16;
    auto kern = k_pp_gemm<NP, WM, WN, WAVES_M, WAVES_N>;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)SHM);
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3(g.tiles_m * g.tiles_n * g.KS), dim3(512), SHM, s, g);
    HIP_LAUNCH_CHECK("k_pp_gemm");
    return 0;
}

// the scratch of launch_pp_gemm: PP_FLAG_WORDS flag words (PP_FLAG_SLOTS launches' worth: a caller zeroes them ONCE, with
// launch_pp_zero_flags -- or by the zero_flags_of_scratch argument of launch_tp_from_f32 / launch_pp_rowmap, whose kernels then do it on
// the side --, in front of up to PP_FLAG_SLOTS launches that each name their own slot), then the partial slots
int64_t pp_scratch_floats() { return PP_FLAG_WORDS + (int64_t)PP_MAX_WG * 256 * 256; }
int launch_pp_zero_flags(hipStream_t s, float* scratch) {
    if (hipMemsetAsync(scratch, 0, PP_FLAG_WORDS * sizeof(unsigned), s) != hipSuccess) { set_error("pp_gemm: flag memset failed"); return MMVAE_E_LAUNCH; }
    return 0;
}

// One layer.  Tile and K split: the (tile, KS) pair with the lowest modelled time (below).  `force` (MMVAE_AUG_TILE): force % 10 =
// 1 / 2 / 3 / 4 = 256 x 256 / 256 x 128 / 128 x 128 / 160 x 256 (0: by the model), force / 10 = KS (0: by the model).
int launch_pp_gemm(hipStream_t s, int NP, TPlanes a, TPlanes b, int M, int N, const float* scale, const float* shift, bool affine, bool relu,
                   float* out32, int64_t ld32, int ncols32, const TPlanes* outp, float* scratch, int64_t scratch_floats, int flag_slot, int force,
                   const unsigned* a_map, int a_map_rows) {
    if (a.KT != b.KT) { set_error("pp_gemm: operands disagree on the K steps (%d, %d)", a.KT, b.KT); return MMVAE_E_BADARG; }
    if (flag_slot < 0 || flag_slot >= PP_FLAG_SLOTS) { set_error("pp_gemm: flag slot %d", flag_slot); return MMVAE_E_BADARG; }
    PPArgs g{};
    g.a = a.p; g.a_plane = a.plane; g.a_Rp = a.Rp; g.a_map = a_map;
    g.KT_a = a.KT;
    if (a_map && (int64_t)a.Rp * a.KT * 32 >= ((int64_t)1 << 32)) { set_error("pp_gemm: a row-mapped operand spans 4 GB per plane at most"); return MMVAE_E_UNSUPPORTED; }
    g.b = b.p; g.b_plane = b.plane; g.b_Rp = b.Rp;
    g.M = M; g.N = N; g.KT = a.KT; g.KS = 1;
    g.scale = scale; g.shift = shift; g.affine = affine ? 1 : 0; g.relu = relu ? 1 : 0;
    g.out32 = out32; g.ld32 = ld32; g.ncols32 = ncols32;
    if (outp) { g.outp = outp->p; g.outp_plane = outp->plane; g.outp_Rp = outp->Rp; g.outp_KT = outp->KT; }
    constexpr int CUS = 256;
    const int64_t part_floats = scratch ? scratch_floats - PP_FLAG_WORDS : 0;
    // modelled time in us: rounds of the chip x K steps per block x the measured time of one K step of the tile (one MI355X,
    // benchmark shapes, every CU busy: 256 x 256 / 256 x 128 / 128 x 128 / 160 x 256 -- three planes 2.08 / 1.22 / 0.85 / 1.29, one
    // plane 0.60 / 0.33 / 0.22 / 0.38) + launch, prologue and epilogue; a K split adds its combine: (KS - 1) / KS of a tile written and
    // read per workgroup at ~45 GB/s each, and two flag round trips
    struct Cand { int bm, bn; double tk; };
    const Cand cands[4] = {{256, 256, NP == 3 ? 2.08 : 0.60}, {256, 128, NP == 3 ? 1.22 : 0.33}, {128, 128, NP == 3 ? 0.85 : 0.22},
                           {160, 256, NP == 3 ? 1.29 : 0.38}};
    const int f_tile = force % 10, f_ks = force / 10;
    if (f_tile > 4 || f_ks > PP_MAX_KS) { set_error("pp_gemm: bad forced tile code %d", force); return MMVAE_E_BADARG; }
    int best_t = -1, best_ks = 1;
    double best = 1e300;
    for (int t = 0; t < 4; ++t) {
        if (f_tile && t != f_tile - 1) continue;
        const int bm = cands[t].bm, bn = cands[t].bn;
        if (b.Rp < cdiv(N, bn) * bn || (a_map ? a_map_rows : a.Rp) < cdiv(M, bm) * bm) continue;
        const int64_t tiles = (int64_t)cdiv(M, bm) * cdiv(N, bn);
        // (a forced split that does not fit the partial slots: the largest that does)
        int f_fit = f_ks;
        while (f_fit > 1 && (tiles * f_fit > PP_MAX_WG || tiles * f_fit * bm * bn > part_floats)) --f_fit;
        for (int ks = 1; ks <= PP_MAX_KS; ++ks) {
            if (f_ks && ks != f_fit) continue;
            if (ks > 1 && (tiles * ks > PP_MAX_WG || tiles * ks * bm * bn > part_floats)) continue;
            if (!f_ks && ks > 1 && g.KT / ks < 4) continue;
            double cost = (double)cdiv64(tiles * ks, CUS) * (double)cdiv(g.KT, ks) * cands[t].tk + 8.0;
            if (ks > 1) cost += 3.0 + 2.0 * (double)(ks - 1) / ks * bm * bn * 4.0 / 45.0e3;
            if (cost < best) { best = cost; best_t = t; best_ks = ks; }
        }
    }
    if (best_t < 0) { set_error("pp_gemm: no tile / K split fits (M %d, N %d, forced %d)", M, N, force); return MMVAE_E_BADARG; }
    g.KS = best_ks;
    g.tiles_m = cdiv(M, cands[best_t].bm);
    g.tiles_n = cdiv(N, cands[best_t].bn);
    if (best_ks > 1) {
        g.part = scratch + PP_FLAG_WORDS;
        g.flags = reinterpret_cast<unsigned*>(scratch) + (int64_t)flag_slot * PP_MAX_WG;
    }
    if (NP == 3) {
        if (best_t == 0) return pp_launch<3, 4, 2, 2, 4>(s, g);
        if (best_t == 1) return pp_launch<3, 2, 2, 4, 2>(s, g);
        if (best_t == 2) return pp_launch<3, 2, 1, 2, 4>(s, g);
        return pp_launch<3, 5, 1, 1, 8>(s, g);
    }
    if (best_t == 0) return pp_launch<1, 4, 2, 2, 4>(s, g);
    if (best_t == 1) return pp_launch<1, 2, 2, 4, 2>(s, g);
    if (best_t == 2) return pp_launch<1, 2, 1, 2, 4>(s, g);
    return pp_launch<1, 5, 1, 1, 8>(s, g);
}

}  // namespace mmvae
