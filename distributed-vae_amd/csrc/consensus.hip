// Evaluation labels and the between-arm consensus on the device.
//
// Every epoch the reference re-runs the training set through the model in eval mode, copies the categorical
// probabilities c of every arm to the host, takes argmax (`classify`, mmidas/_utils.py:79-80), builds one
// C x C confusion matrix per arm pair with np.add.at (`compute_confmat`, :84-95), normalises it by
// max(row sum, column sum) (`confmat_normalize`, :98-100) and averages its diagonal (`confmat_mean`, :128-129)
// -- mmidas/cpl_mixvae.py:563-657.  Here the labels never leave the device: k_classify reads c from the workspace
// of the (encoder + latent block only) eval forward, k_confmat accumulates integer counts with atomics, and
// k_consensus does the normalisation and the mean in fp64 in numpy's summation order, so the result is bit-identical
// to the reference's host arithmetic (counts are integers; the only rounding is the division and the mean).
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

// labels[a][b] = argmax_k c[a][b][k], first maximum on ties (np.argmax).  One wave per cell; grid-stride.
__global__ __launch_bounds__(256) void k_classify(const float* __restrict__ cc, int64_t n_cells, int C,
                                                  int32_t* __restrict__ labels) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwave = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t cell = wave; cell < n_cells; cell += nwave) {
        const float* p = cc + cell * C;
        float best = -INFINITY;
        int arg = 1 << 30;
        for (int k = lane; k < C; k += 64) {
            const float v = p[k];
            if (v > best) { best = v; arg = k; }   // strictly greater: the lane keeps its first maximum
        }
        const float m = wave_max(best);
        const int cand = wave_min_i(best == m ? arg : (1 << 30));
        if (lane == 0) labels[cell] = cand;
    }
}

// counts[pair(a,b)][labels[a][i]][labels[b][i]] += 1 for a < b, pairs in the reference's loop order
// (cpl_mixvae.py:644-653: for a in range(A): for b in range(a+1, A)).  labels: [A][n].
// Cells of one batch pile onto a few (label, label) cells -- arms that agree put everything on the diagonal -- so
// global atomics serialise (14.7 us for 5000 cells x 1 pair).  With `lds_pairs` > 0 a workgroup first counts its cells
// in an LDS histogram (32-bit, lds_pairs x C x C) and then adds its non-zero cells to the global counts.
__global__ __launch_bounds__(256) void k_confmat(const int32_t* __restrict__ labels, int A, int64_t n, int C,
                                                 unsigned long long* __restrict__ counts, int lds_pairs) {
    extern __shared__ unsigned int hist[];
    const int npairs = A * (A - 1) / 2;
    const bool use_lds = lds_pairs >= npairs;
    const int cells = npairs * C * C;
    if (use_lds) {
        for (int i = threadIdx.x; i < cells; i += blockDim.x) hist[i] = 0u;
        __syncthreads();
    }
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        int pair = 0;
        for (int a = 0; a < A; ++a) {
            const int la = labels[(int64_t)a * n + i];
            for (int b = a + 1; b < A; ++b, ++pair) {
                const int lb = labels[(int64_t)b * n + i];
                if ((unsigned)la < (unsigned)C && (unsigned)lb < (unsigned)C) {
                    const int64_t cell = ((int64_t)pair * C + la) * C + lb;
                    if (use_lds) atomicAdd(&hist[cell], 1u);
                    else atomicAdd(counts + cell, 1ull);
                }
            }
        }
    }
    if (use_lds) {
        __syncthreads();
        for (int i = threadIdx.x; i < cells; i += blockDim.x) {
            const unsigned int v = hist[i];
            if (v) atomicAdd(counts + i, (unsigned long long)v);
        }
    }
}

// numpy's pairwise summation of n <= 128 doubles with element stride `st` (numpy/core/src/umath/loops_utils.h,
// pairwise_sum: < 8 elements sequentially; else eight running sums, combined as a balanced tree, then the tail)
__device__ double np_pairwise_sum(const double* a, int n, int st) {
    if (n < 8) {
        double res = 0.0;
        for (int i = 0; i < n; ++i) res += a[i * st];
        return res;
    }
    double r[8];
    for (int k = 0; k < 8; ++k) r[k] = a[k * st];
    int i = 8;
    for (; i < n - (n % 8); i += 8)
        for (int k = 0; k < 8; ++k) r[k] += a[(i + k) * st];
    double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; ++i) res += a[i * st];
    return res;
}

// grid (npairs), 128 threads, C <= 128.  maxes[j] = max(column sum j, row sum j); norm[i][j] = cm[i][j] / maxes[j]
// where maxes[j] != 0 else 0 (np.divide broadcasts `maxes` along the last axis); consensus = mean(diag(norm)).
__global__ __launch_bounds__(128) void k_consensus(const unsigned long long* __restrict__ counts, int C,
                                                   double* __restrict__ cm_norm, double* __restrict__ consensus) {
    __shared__ double sh_max[128], sh_diag[128];
    const int pair = blockIdx.x, j = threadIdx.x;
    const unsigned long long* cm = counts + (int64_t)pair * C * C;
    if (j < C) {
        unsigned long long cs = 0, rs = 0;   // integer sums are exact (and equal numpy's float sums below 2^53)
        for (int i = 0; i < C; ++i) { cs += cm[(int64_t)i * C + j]; rs += cm[(int64_t)j * C + i]; }
        const double mx = (double)(cs > rs ? cs : rs);
        sh_max[j] = mx;
        sh_diag[j] = mx != 0.0 ? (double)cm[(int64_t)j * C + j] / mx : 0.0;
    }
    __syncthreads();
    if (cm_norm) {
        double* out = cm_norm + (int64_t)pair * C * C;
        for (int e = j; e < C * C; e += blockDim.x) {
            const double mx = sh_max[e % C];
            out[e] = mx != 0.0 ? (double)cm[e] / mx : 0.0;
        }
    }
    if (j == 0) consensus[pair] = np_pairwise_sum(sh_diag, C, 1) / (double)C;
}

int launch_classify(const float* cc, int64_t n_cells, int C, int32_t* labels, hipStream_t s) {
    const int blocks = (int)imin64(2048, cdiv64(n_cells, 4));
    hipLaunchKernelGGL(k_classify, dim3(blocks), dim3(256), 0, s, cc, n_cells, C, labels);
    HIP_LAUNCH_CHECK("k_classify");
    return 0;
}

int launch_confmat(const int32_t* labels, int A, int64_t n, int C, int64_t* counts, hipStream_t s) {
    if (A < 2) return 0;
    const int npairs = A * (A - 1) / 2;
    const size_t shm = (size_t)npairs * C * C * sizeof(unsigned int);
    const bool lds = shm <= 64 * 1024;                      // 92 x 92 categories: one pair 34 KB (A = 2)
    // few, fat workgroups when counting in LDS (each flushes its whole histogram), many thin ones otherwise
    const int blocks = lds ? (int)imin64(32, cdiv64(n, 1024)) : (int)imin64(1024, cdiv64(n, 256));
    hipLaunchKernelGGL(k_confmat, dim3(blocks > 0 ? blocks : 1), dim3(256), lds ? shm : 0, s, labels, A, n, C,
                       reinterpret_cast<unsigned long long*>(counts), lds ? npairs : 0);
    HIP_LAUNCH_CHECK("k_confmat");
    return 0;
}

int launch_consensus(const int64_t* counts, int npairs, int C, double* cm_norm, double* consensus, hipStream_t s) {
    hipLaunchKernelGGL(k_consensus, dim3(npairs), dim3(128), 0, s, reinterpret_cast<const unsigned long long*>(counts), C,
                       cm_norm, consensus);
    HIP_LAUNCH_CHECK("k_consensus");
    return 0;
}

}  // namespace mmvae
