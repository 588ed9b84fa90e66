// Data-parallel gradient exchange inside the library: ONE stream-ordered RCCL all-reduce (average) of the flat fp32
// gradient buffer per step, on the stream the step runs on (replaces the reference's FSDP gradient
// traffic, train.py:140-143, and the process-group bring-up of mmidas/_dist_utils.py:12-55 for this one collective).
// RCCL is resolved at run time (dlopen of librccl.so.1, the library PyTorch-ROCm itself uses) so that libmmvae_hip.so loads
// on hosts without it; the entry points fail with MMVAE_E_UNSUPPORTED there.
#include "common.hpp"
#include <dlfcn.h>
#include <string.h>
#include <mutex>

namespace mmvae {
struct Id128 { char b[128]; };      // ncclUniqueId (passed by value to ncclCommInitRank)
namespace {
struct Rccl {
    void* lib = nullptr;
    int (*GetUniqueId)(void*) = nullptr;
    int (*CommInitRank)(void**, int, Id128, int) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    bool ok = false;
};
Rccl& rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        // the RCCL this process has mapped already (PyTorch-ROCm ships its own librccl.so, possibly under another soname than
        // the system's): its symbols first, so that no second RCCL runtime ends up in the process; only then the names
        if (dlsym(RTLD_DEFAULT, "ncclCommInitRank") && dlsym(RTLD_DEFAULT, "ncclAllReduce")) r.lib = dlopen(nullptr, RTLD_NOW);
        if (!r.lib)
            for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
                r.lib = dlopen(name, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);
                if (r.lib) break;
            }
        if (!r.lib)
            for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
                r.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
                if (r.lib) break;
            }
        if (!r.lib) return;
        r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(dlsym(r.lib, "ncclGetUniqueId"));
        r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(dlsym(r.lib, "ncclCommInitRank"));
        r.AllReduce = reinterpret_cast<decltype(r.AllReduce)>(dlsym(r.lib, "ncclAllReduce"));
        r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(r.lib, "ncclCommDestroy"));
        r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(r.lib, "ncclGetErrorString"));
        r.ok = r.GetUniqueId && r.CommInitRank && r.AllReduce && r.CommDestroy;
    });
    return r;
}
int fail(const char* what, int rc) {
    Rccl& r = rccl();
    set_error("%s: %s", what, r.GetErrorString ? r.GetErrorString(rc) : "RCCL error");
    return MMVAE_E_LAUNCH;
}
int need_rccl() {
    if (rccl().ok) return 0;
    set_error("RCCL (librccl.so.1) is not available on this host");
    return MMVAE_E_UNSUPPORTED;
}
constexpr int NCCL_FLOAT32 = 7, NCCL_AVG = 4;   // ncclDataType_t / ncclRedOp_t values of the NCCL 2.10+ ABI RCCL implements
}  // namespace
}  // namespace mmvae

using namespace mmvae;

extern "C" {

int mmvae_dp_unique_id(uint8_t id[MMVAE_DP_ID_BYTES]) {
    if (!id) { set_error("id is null"); return MMVAE_E_BADARG; }
    if (int rc = need_rccl()) return rc;
    Id128 u;
    memset(&u, 0, sizeof(u));
    if (int rc = rccl().GetUniqueId(&u)) return fail("ncclGetUniqueId", rc);
    memcpy(id, &u, MMVAE_DP_ID_BYTES);
    return 0;
}

int mmvae_dp_init(const uint8_t id[MMVAE_DP_ID_BYTES], int rank, int world_size, void** comm) {
    if (!id || !comm || world_size < 1 || rank < 0 || rank >= world_size) { set_error("dp_init: bad argument"); return MMVAE_E_BADARG; }
    if (int rc = need_rccl()) return rc;
    Id128 u;
    memcpy(&u, id, MMVAE_DP_ID_BYTES);
    void* c = nullptr;
    if (int rc = rccl().CommInitRank(&c, world_size, u, rank)) return fail("ncclCommInitRank", rc);
    *comm = c;
    return 0;
}

int mmvae_allreduce_grads(void* comm, float* grads, int64_t n, void* stream) {
    if (!comm || !grads || n <= 0) { set_error("allreduce_grads: bad argument"); return MMVAE_E_BADARG; }
    if (int rc = need_rccl()) return rc;
    if (int rc = rccl().AllReduce(grads, grads, (size_t)n, NCCL_FLOAT32, NCCL_AVG, comm, reinterpret_cast<hipStream_t>(stream)))
        return fail("ncclAllReduce", rc);
    return 0;
}

int mmvae_dp_destroy(void* comm) {
    if (!comm) return 0;
    if (int rc = need_rccl()) return rc;
    if (int rc = rccl().CommDestroy(comm)) return fail("ncclCommDestroy", rc);
    return 0;
}

}  // extern "C"
