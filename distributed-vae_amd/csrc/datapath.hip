// Device-resident data path: the reference keeps the cells x genes matrix on the host
// and feeds batches through torch DataLoader workers, pinned memory and one H2D copy per step
// (mmidas/utils/dataloader.py:86-168).  50 000 x 5000 fp32 is 1 GB, 500 000 x 5000 is 10 GB: it lives in HBM here, a
// shuffled batch is a row gather on the device.
//
// k_gather_rows: out[i, :] = data[idx[i], :].  HBM-bound (reads and writes every byte once): a wave copies one row
// with 16-byte accesses (4-byte when the row stride or width is not a multiple of 4), 4 loads in flight per lane.
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

template <bool VEC>
__global__ __launch_bounds__(256) void k_gather_rows(const float* __restrict__ data, int64_t ld, int64_t n_rows,
                                                     const int64_t* __restrict__ idx, int64_t n, int D,
                                                     float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwave = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t i = wave; i < n; i += nwave) {
        int64_t r = idx[i];
        r = r < 0 ? 0 : (r >= n_rows ? n_rows - 1 : r);   // never read outside the matrix (the host validates indices)
        const float* src = data + r * ld;
        float* dst = out + i * (int64_t)D;
        if (VEC) {
            const float4* s4 = reinterpret_cast<const float4*>(src);
            float4* d4 = reinterpret_cast<float4*>(dst);
            const int n4 = D >> 2;
            for (int c0 = 0; c0 < n4; c0 += 256) {
                float4 v[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int c = c0 + lane + 64 * k;
                    v[k] = s4[min(c, n4 - 1)];
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int c = c0 + lane + 64 * k;
                    if (c < n4) d4[c] = v[k];
                }
            }
        } else {
            for (int c = lane; c < D; c += 64) dst[c] = src[c];
        }
    }
}


// dst[r][c] = bf16(src[r][c]) (round to nearest even, v_cvt_pk_bf16_f32: the rounding the one-plane engine applies to its
// operands on their way into LDS) for c < D rounded up to 4: the bf16 copy of a resident matrix, made once per data set.  Only
// the row's own columns are touched -- for a column-offset view of a wider matrix (ld > D, data pointing into the rows) the
// pitch behind the LAST row's columns is not part of the view's storage and is neither read nor written.
__global__ __launch_bounds__(256) void k_to_bf16(const float* __restrict__ src, int64_t ld, int64_t n_rows, int D, unsigned short* __restrict__ dst) {
    const int64_t quads = (D + 3) >> 2;                  // ld % 4 == 0 and ld >= D: the last quad of a row stays inside its pitch
    const int64_t n = n_rows * quads;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / quads, at = r * ld + 4 * (i - r * quads);
        const float4 v = *reinterpret_cast<const float4*>(src + at);
        typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
        bf16x2 a, b;
        a[0] = (__bf16)v.x; a[1] = (__bf16)v.y; b[0] = (__bf16)v.z; b[1] = (__bf16)v.w;
        *reinterpret_cast<uint2*>(dst + at) = make_uint2(__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b));
    }
}

}  // namespace mmvae

using namespace mmvae;

extern "C" int mmvae_gather_rows(const float* data, int64_t ld, int64_t n_rows, const int64_t* idx, int64_t n, int32_t D,
                                 float* out, void* stream) {
    if (!data || !idx || !out || n <= 0 || D <= 0 || n_rows <= 0 || ld < D) { set_error("gather_rows: bad argument"); return MMVAE_E_BADARG; }
    const bool vec = (D % 4 == 0) && (ld % 4 == 0) && ((uintptr_t)data % 16 == 0) && ((uintptr_t)out % 16 == 0);
    const int blocks = (int)imin64(cdiv64(n, 4), 4096);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (vec) hipLaunchKernelGGL((k_gather_rows<true>), dim3(blocks), dim3(256), 0, s, data, ld, n_rows, idx, n, D, out);
    else hipLaunchKernelGGL((k_gather_rows<false>), dim3(blocks), dim3(256), 0, s, data, ld, n_rows, idx, n, D, out);
    HIP_LAUNCH_CHECK("k_gather_rows");
    return 0;
}

extern "C" int mmvae_to_bf16(const float* src, int64_t ld, int64_t n_rows, int32_t D, uint16_t* dst, void* stream) {
    if (!src || !dst || n_rows <= 0 || D <= 0 || ld < D) { set_error("to_bf16: bad argument"); return MMVAE_E_BADARG; }
    if ((ld & 3) || (reinterpret_cast<uintptr_t>(src) & 15) || (reinterpret_cast<uintptr_t>(dst) & 7)) {
        set_error("to_bf16: needs ld %% 4 == 0, a 16-byte aligned source and an 8-byte aligned destination");
        return MMVAE_E_UNSUPPORTED;
    }
    const int64_t n = n_rows * (int64_t)((D + 3) >> 2);
    hipLaunchKernelGGL(k_to_bf16, dim3((unsigned)imin64(cdiv64(n, 256 * 4), 8192)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), src, ld, n_rows, D, dst);
    HIP_LAUNCH_CHECK("k_to_bf16");
    return 0;
}
