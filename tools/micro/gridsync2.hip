// Micro-benchmark (round 3): the grid barrier redone with the documented primitive (MI355X_MICROARCH.md, price list row
// "barrier-xcd"), for the question "can fc2..fc5 of the train-mode encoder chain be ONE launch with a barrier per BatchNorm?"
//   hipcc --offload-arch=gfx950 -O3 -o gridsync2 gridsync2.hip && ./gridsync2 [workgroups=158] [iters=200] [threads=512]
//
// Round 2's version (gridsync.hip) measured 24-26 us per barrier at 158 workgroups; it acquired at SYSTEM scope, had all 512
// threads execute __threadfence(), grouped by blockIdx & 7 instead of the real XCC_ID and polled a plain relaxed load.
//
// Variants (per iteration; G workgroups of NT threads, all resident; every spin is bounded and sets an abort word):
//   launch        one (almost) empty kernel per iteration, stream-ordered                      -> the launch boundary
//   xcd_fence     barrier-xcd as documented: per-XCC arrival counter keyed by the real XCC_ID (membership from a census
//                 behind one flat barrier at kernel start), the XCD's last arriver runs ONE agent-scope release fence
//                 (L2 write-back) and arrives on the top counter, the last of those bumps the eight per-XCC generation
//                 words; every workgroup polls ITS XCC's word (relaxed sc1 load + s_sleep) from ONE lane, then ONE
//                 agent-scope acquire fence, s_waitcnt vmcnt(0), __syncthreads()
//   xcd_atomic    the same arrival / generation tree with NO fences: legal when everything that crosses workgroups is
//                 written by agent-scope atomics and read by agent-scope (sc1) atomic loads -- which is exactly the
//                 BatchNorm exchange of the chain kernels (fixed-point accumulator sets, common.hpp acc_add / acc_get)
//   flat_atomic   one counter, no hierarchy, no fences
//   acc_exchange  xcd_atomic + the payload of one BatchNorm: every workgroup adds 2 sums x 3 slots x 100 columns by
//                 64-bit agent-scope atomics in front of the barrier and reads the 600 slots back with sc1 loads behind it
//                 (checked: the value read must be the exact sum of all workgroups' addends)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
constexpr unsigned MAX_POLLS = 1u << 20;
constexpr int ACC_COLS = 100, ACC_SLOTS = 6, ACC_W = 128;

typedef __attribute__((address_space(1))) unsigned gu32;
typedef __attribute__((address_space(1))) unsigned long long gu64;

struct Sync {
    unsigned* xcc_members;   // [8 * 32] census: workgroups per XCC (one counter per 128-byte line)
    unsigned* xcc_arrive;    // [8 * 32]
    unsigned* xcc_gen;       // [8 * 32]
    unsigned* top;           // [32]
    unsigned* flat;          // [32]
    unsigned* abort_;        // [32]
};

__device__ __forceinline__ unsigned ld_rlx(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned add_rlx(unsigned* p, unsigned v) { return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__device__ __forceinline__ bool wait_ge(const unsigned* p, unsigned target, unsigned* abort_) {
    for (unsigned i = 0; i < MAX_POLLS; ++i) {
        if (ld_rlx(p) >= target) return true;
        __builtin_amdgcn_s_sleep(2);
    }
    atomicExch(abort_, 1u);
    return false;
}
__device__ __forceinline__ int xcc_id() { return (int)(__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)) & 7u); }   // HW_REG_XCC_ID[3:0]

__global__ void k_empty(float* out) { if (threadIdx.x == 0 && blockIdx.x == 0) out[0] += 1.f; }

// MODE 0 xcd_fence, 1 xcd_atomic, 2 flat_atomic, 3 acc_exchange
template <int MODE>
__global__ __launch_bounds__(512) void k_sync(Sync s, int iters, unsigned long long* acc, float* part, unsigned* bad) {
    const int G = gridDim.x, tid = threadIdx.x;
    __shared__ unsigned sh_ok, sh_members;
    const int xcc = xcc_id();
    bool ok = true;
    // census + one flat barrier: how many workgroups share my XCC
    if (tid == 0) {
        add_rlx(s.xcc_members + xcc * 32, 1u);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        add_rlx(s.flat, 1u);
        ok = wait_ge(s.flat, (unsigned)G, s.abort_);
        sh_members = ld_rlx(s.xcc_members + xcc * 32);
        sh_ok = ok;
    }
    __syncthreads();
    ok = sh_ok != 0;
    const unsigned members = sh_members;
    unsigned nerr = 0;
    for (int it = 1; it <= iters && ok; ++it) {
        if (MODE == 0) {
            // "work" that other workgroups will read with plain loads: needs the release / acquire pair
            if (tid < 200) part[(size_t)blockIdx.x * 200 + tid] = (float)(it + tid);
        }
        if (MODE == 3) {
            // one BatchNorm's block sums: 600 64-bit atomics per workgroup (set alternates so that a set is never added to
            // while a slow workgroup still reads the previous iteration's)
            unsigned long long* set = acc + (size_t)(it & 1) * ACC_SLOTS * ACC_W;
            for (int i = tid; i < ACC_SLOTS * ACC_COLS; i += blockDim.x) {
                const int slot = i / ACC_COLS, col = i % ACC_COLS;
                __hip_atomic_fetch_add(set + slot * ACC_W + col, (unsigned long long)(blockIdx.x + col + 1), __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // every storing wave drains before the workgroup arrives
        __syncthreads();
        if (tid == 0) {
            if (MODE == 2) {
                add_rlx(s.flat, 1u);
                ok = wait_ge(s.flat, (unsigned)G * (unsigned)(it + 1), s.abort_);
            } else {
                const unsigned a = add_rlx(s.xcc_arrive + xcc * 32, 1u);
                if (a + 1 == members * (unsigned)it) {
                    // last workgroup of this XCD: every workgroup of the XCD has drained its stores into this L2
                    if (MODE == 0) {
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    }
                    // XCCs that have members: counted by the census (an empty XCC never arrives)
                    unsigned nx = 0;
                    for (int x = 0; x < 8; ++x) nx += ld_rlx(s.xcc_members + x * 32) != 0;
                    const unsigned t = add_rlx(s.top, 1u);
                    if (t + 1 == nx * (unsigned)it) {
                        for (int x = 0; x < 8; ++x)
                            __hip_atomic_store(s.xcc_gen + x * 32, (unsigned)it, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
                ok = wait_ge(s.xcc_gen + xcc * 32, (unsigned)it, s.abort_);
                if (MODE == 0) {
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
            }
            sh_ok = ok;
        }
        __syncthreads();
        ok = sh_ok != 0;
        if (MODE == 0 && ok) {
            // read a neighbour's record with plain loads (behind the acquire) and check it
            const int nb = (blockIdx.x + 97) % G;
            if (tid < 200 && part[(size_t)nb * 200 + tid] != (float)(it + tid)) ++nerr;
        }
        if (MODE == 3 && ok) {
            const unsigned long long* set = acc + (size_t)(it & 1) * ACC_SLOTS * ACC_W;
            const unsigned long long rounds = (unsigned long long)((it + 1) / 2);    // iterations that added to this set so far
            for (int i = tid; i < ACC_SLOTS * ACC_COLS; i += blockDim.x) {
                const int slot = i / ACC_COLS, col = i % ACC_COLS;
                const unsigned long long v = __hip_atomic_load(set + slot * ACC_W + col, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned long long want = rounds * ((unsigned long long)G * (col + 1) + (unsigned long long)G * (G - 1) / 2);
                if (v != want) ++nerr;
            }
        }
        if (MODE == 0) {
            // the record is overwritten next iteration: readers must be done -> second barrier in a real kernel; here the
            // neighbour check tolerates nothing, so keep iterations apart with a workgroup-local delay only when checking
        }
    }
    if (nerr) atomicAdd(bad, nerr);
}

template <int MODE>
static int run(const char* name, int G, int iters, int NT, Sync s, unsigned long long* acc, float* part, unsigned* bad, bool check) {
    auto reset = [&]() -> int {
        CK(hipMemset(s.xcc_members, 0, 8 * 32 * 4)); CK(hipMemset(s.xcc_arrive, 0, 8 * 32 * 4)); CK(hipMemset(s.xcc_gen, 0, 8 * 32 * 4));
        CK(hipMemset(s.top, 0, 128)); CK(hipMemset(s.flat, 0, 128)); CK(hipMemset(s.abort_, 0, 128));
        CK(hipMemset(acc, 0, 2 * ACC_SLOTS * ACC_W * 8)); CK(hipMemset(bad, 0, 4));
        return 0;
    };
    if (reset()) return 1;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k_sync<MODE>, dim3(G), dim3(NT), 0, 0, s, 3, acc, part, bad);
    CK(hipDeviceSynchronize());
    if (reset()) return 1;
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(k_sync<MODE>, dim3(G), dim3(NT), 0, 0, s, iters, acc, part, bad);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned ab = 0, nb = 0, mem[8 * 32];
    CK(hipMemcpy(&ab, s.abort_, 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(&nb, bad, 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(mem, s.xcc_members, sizeof(mem), hipMemcpyDeviceToHost));
    printf("%-13s G=%3d NT=%d  %7.2f us per iteration%s", name, G, NT, ms * 1e3f / iters, ab ? "   (ABORTED: a wait timed out)" : "");
    if (check) printf("   wrong values read: %u", nb);
    printf("   [workgroups per XCC:");
    for (int x = 0; x < 8; ++x) printf(" %u", mem[x * 32]);
    printf("]\n");
    return 0;
}

int main(int argc, char** argv) {
    const int G = argc > 1 ? atoi(argv[1]) : 158;
    const int iters = argc > 2 ? atoi(argv[2]) : 200;
    const int NT = argc > 3 ? atoi(argv[3]) : 512;
    int dev = 0, cus = 0, per_cu = 0;
    CK(hipGetDevice(&dev));
    CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_sync<3>, NT, 0));
    printf("CUs %d, resident workgroups per CU %d\n", cus, per_cu);
    if (G > cus * (per_cu > 1 ? per_cu - 1 : 1)) { printf("grid of %d workgroups may not be co-resident\n", G); return 1; }
    Sync s;
    unsigned long long* acc;
    float *part, *sink;
    unsigned* bad;
    CK(hipMalloc(&s.xcc_members, 8 * 32 * 4)); CK(hipMalloc(&s.xcc_arrive, 8 * 32 * 4)); CK(hipMalloc(&s.xcc_gen, 8 * 32 * 4));
    CK(hipMalloc(&s.top, 128)); CK(hipMalloc(&s.flat, 128)); CK(hipMalloc(&s.abort_, 128));
    CK(hipMalloc(&acc, 2 * ACC_SLOTS * ACC_W * 8)); CK(hipMalloc(&part, (size_t)2048 * 200 * 4)); CK(hipMalloc(&sink, 128));
    CK(hipMalloc(&bad, 128));
    CK(hipMemset(sink, 0, 128));
    {
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(k_empty, dim3(G), dim3(NT), 0, 0, sink);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(k_empty, dim3(G), dim3(NT), 0, 0, sink);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms = 0.f;
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-13s G=%3d NT=%d  %7.2f us per iteration\n", "launch", G, NT, ms * 1e3f / iters);
    }
    if (run<0>("xcd_fence", G, iters, NT, s, acc, part, bad, false)) return 1;
    if (run<1>("xcd_atomic", G, iters, NT, s, acc, part, bad, false)) return 1;
    if (run<2>("flat_atomic", G, iters, NT, s, acc, part, bad, false)) return 1;
    if (run<3>("acc_exchange", G, iters, NT, s, acc, part, bad, true)) return 1;
    return 0;
}
