// Micro-benchmark (round 4): what ONE BatchNorm exchange costs inside a launch, in the two protocols a fused chain kernel can use.
//   hipcc --offload-arch=gfx950 -O3 -o xchg_bench xchg_bench.hip && ./xchg_bench [workgroups per arm=157] [arms=2] [layers=8] [iters=40] [delay_cycles=0]
//
// G = arms x (workgroups per arm) workgroups of 256 threads (four waves; wave w owns columns [32 w, 32 w + 32) of a 100-wide layer),
// all resident.  Per "layer" every workgroup adds its block sums of 100 columns to its arm's accumulator set and then needs the
// COMPLETE sums of its own columns before it can go on.
//   counter   round 3's protocol (with `replicas` copies of the set: workgroup b adds to copy b % replicas, the reader sums them)
//             round 3's protocol (chain.hip k_enc_fwd_fused): 6 slots per column by agent-scope atomics, s_waitcnt vmcnt(0),
//             __syncthreads, one atomic on the arm's done counter, one lane polls it, __syncthreads, the slots are read (sc1 loads)
//   counted   the count travels IN the data: every slot is (piece << 12) + 1, eight 36-bit pieces per column (two sums x four),
//             a wave polls the eight slots of its own columns until every low field reads (workgroups per arm); no counter, no
//             wait for the atomics' acknowledgement, no workgroup barrier
// `delay_cycles` of s_memtime spinning between the add and the poll stand in for the other arm's compute.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
constexpr int ACC_W = 128, COLS = 100;
constexpr unsigned MAX_POLLS = 1u << 20;

__device__ __forceinline__ void spin(unsigned long long cycles) {
    if (!cycles) return;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    while (__builtin_amdgcn_s_memtime() - t0 < cycles) __builtin_amdgcn_s_sleep(1);
}

// sets: [nsets][arms][SLOTS][ACC_W] int64, zeroed; done: [nsets][arms][32] u32
template <int MODE>
__global__ __launch_bounds__(256) void k_xchg(long long* sets, unsigned* done, int per_arm, int layers, int iters, unsigned long long delay,
                                              unsigned* bad, unsigned* abort_, int R) {
    constexpr int SLOTS = MODE == 0 ? 6 : 8;
    const int arms = gridDim.x / per_arm, arm = blockIdx.x / per_arm, blk = blockIdx.x % per_arm;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, col = wv * 32 + (lane & 31);
    const bool mine = lane < 32 && col < COLS;
    __shared__ unsigned sh_ok;
    unsigned nerr = 0;
    bool ok = true;
    for (int it = 0; it < iters && ok; ++it)
        for (int l = 0; l < layers && ok; ++l) {
            const size_t si = (size_t)(it * layers + l) * arms + arm;
            long long* set = sets + si * 16 * SLOTS * ACC_W;     // up to 16 replicas of a set (MODE 0: workgroup blk adds to replica blk % R)
            const long long v = (long long)(blk + col + 1);
            if (MODE == 0) {
                if (mine) {
                    long long* rs = set + (size_t)(blk % R) * SLOTS * ACC_W;
#pragma unroll
                    for (int s = 0; s < SLOTS; ++s) __hip_atomic_fetch_add(rs + s * ACC_W + col, v + s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                spin(delay);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                if (tid == 0) {
                    unsigned* d = done + si * 32;
                    __hip_atomic_fetch_add(d, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    bool o = false;
                    for (unsigned i = 0; i < MAX_POLLS; ++i) {
                        if (__hip_atomic_load(d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= (unsigned)per_arm) { o = true; break; }
                        __builtin_amdgcn_s_sleep(2);
                    }
                    if (!o) atomicExch(abort_, 1u);
                    sh_ok = o;
                }
                __syncthreads();
                ok = sh_ok != 0;
                if (mine && ok) {
                    const long long want0 = (long long)per_arm * (col + 1) + (long long)per_arm * (per_arm - 1) / 2;
                    long long q[SLOTS];
#pragma unroll
                    for (int s = 0; s < SLOTS; ++s) q[s] = 0;
                    for (int r = 0; r < R; ++r)
#pragma unroll
                        for (int s = 0; s < SLOTS; ++s)
                            q[s] += __hip_atomic_load(set + ((size_t)r * SLOTS + s) * ACC_W + col, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
                    for (int s = 0; s < SLOTS; ++s)
                        if (q[s] != want0 + (long long)per_arm * s) ++nerr;
                }
            } else {
                if (mine)
#pragma unroll
                    for (int s = 0; s < SLOTS; ++s)
                        __hip_atomic_fetch_add(set + s * ACC_W + col, ((s & 1 ? -(v + s) : (v + s)) << 12) + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                spin(delay);
                long long q[SLOTS];
                bool o = false;
                for (unsigned i = 0; i < MAX_POLLS; ++i) {
                    bool all = true;
                    if (mine) {
#pragma unroll
                        for (int s = 0; s < SLOTS; ++s) q[s] = __hip_atomic_load(set + s * ACC_W + col, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
                        for (int s = 0; s < SLOTS; ++s) all = all && ((int)(q[s] & 4095) == per_arm);
                    }
                    if (__all(all)) { o = true; break; }
                    __builtin_amdgcn_s_sleep(1);
                }
                if (!o) { atomicExch(abort_, 1u); ok = false; }
                if (mine && ok) {
                    const long long want0 = (long long)per_arm * (col + 1) + (long long)per_arm * (per_arm - 1) / 2;
#pragma unroll
                    for (int s = 0; s < SLOTS; ++s) {
                        const long long sum = (q[s] - (q[s] & 4095)) >> 12, w = want0 + (long long)per_arm * s;
                        if (sum != (s & 1 ? -w : w)) ++nerr;
                    }
                }
                // (a real kernel has a workgroup barrier per layer for the activation tile; none is needed for the exchange)
                ok = __shfl(ok ? 1 : 0, 0) != 0;
            }
        }
    if (nerr) atomicAdd(bad, nerr);
}

template <int MODE>
static int run(const char* name, int per_arm, int arms, int layers, int iters, unsigned long long delay, long long* sets, size_t set_bytes,
               unsigned* done, size_t done_bytes, unsigned* bad, unsigned* abort_, int R) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e30f;
    unsigned ab = 0, nb = 0;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipMemset(sets, 0, set_bytes)); CK(hipMemset(done, 0, done_bytes)); CK(hipMemset(bad, 0, 4)); CK(hipMemset(abort_, 0, 4));
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(k_xchg<MODE>, dim3(per_arm * arms), dim3(256), 0, 0, sets, done, per_arm, layers, iters, delay, bad, abort_, R);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms = 0.f;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep && ms < best) best = ms;
        unsigned a = 0, b = 0;
        CK(hipMemcpy(&a, abort_, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(&b, bad, 4, hipMemcpyDeviceToHost));
        ab |= a; nb += b;
    }
    printf("%-8s per_arm=%3d arms=%d replicas=%2d delay=%5llu cycles  %6.2f us per layer%s   wrong values: %u\n", name, per_arm, arms, R, delay,
           best * 1e3f / (iters * layers), ab ? "  (ABORTED)" : "", nb);
    return 0;
}

int main(int argc, char** argv) {
    const int per_arm = argc > 1 ? atoi(argv[1]) : 157, arms = argc > 2 ? atoi(argv[2]) : 2;
    const int layers = argc > 3 ? atoi(argv[3]) : 8, iters = argc > 4 ? atoi(argv[4]) : 40;
    const unsigned long long delay = argc > 5 ? strtoull(argv[5], nullptr, 10) : 0;
    int dev = 0, cus = 0, pc0 = 0, pc1 = 0;
    CK(hipGetDevice(&dev));
    CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&pc0, k_xchg<0>, 256, 0));
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&pc1, k_xchg<1>, 256, 0));
    if (per_arm * arms > cus * (pc0 < pc1 ? pc0 : pc1) / 2) { printf("grid may not be co-resident\n"); return 1; }
    const size_t nsets = (size_t)iters * layers * arms;
    const size_t set_bytes = nsets * 16 * 8 * ACC_W * 8, done_bytes = nsets * 32 * 4;
    long long* sets; unsigned *done, *bad, *abort_;
    CK(hipMalloc(&sets, set_bytes)); CK(hipMalloc(&done, done_bytes)); CK(hipMalloc(&bad, 128)); CK(hipMalloc(&abort_, 128));
    // clock rate of s_memtime: 100 MHz constant on this part, so `delay` is in 10 ns units
    for (int R = 1; R <= 16; R *= 2)
        if (run<0>("counter", per_arm, arms, layers, iters, delay, sets, set_bytes, done, done_bytes, bad, abort_, R)) return 1;
    if (run<1>("counted", per_arm, arms, layers, iters, delay, sets, set_bytes, done, done_bytes, bad, abort_, 1)) return 1;
    return 0;
}
