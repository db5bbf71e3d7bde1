// The one real exchange step of the path: the T-frame tower shards over the GPUs of a
// node, every rank encodes a contiguous block of frames and ONE all-gather of the 1024-wide visual tokens (before the
// 4096-wide projector: 4x fewer bytes; 1 MiB per rank at T = 16 on 8 GPUs) rebuilds [T, 256, Dv] in rank order =
// chronological order on every rank.  The collective is RCCL's ncclAllGather over xGMI, held behind the C ABI in an opaque
// teo_ctx: no torch type, no Python collective on the data path.
//
// RCCL is bound at run time (dlopen/dlsym of librccl.so, preferring the copy the process already holds -- torch ships its
// own -- so the HIP runtime and RCCL instances stay single): libteo_hip.so itself has no link-time dependency on RCCL and
// still loads on a box without it (teo_ctx_create then fails with TEO_ERR_UNSUPPORTED and a message).
//
// Reference counterpart: none -- the reference is single-GPU (scripts/eval_teochat.sh:9-10); the frames it would stack on
// one device (llava_arch.py:194, modeling_image.py:641-643) are what is sharded here.
#include <dlfcn.h>

#include <cstdio>
#include <mutex>

#include "ops.h"

namespace teo {

// the handful of RCCL declarations used (rccl.h: ncclUniqueId is 128 opaque bytes, ncclComm_t an opaque pointer)
struct NcclId { char internal[TEO_COMM_ID_BYTES]; };
typedef void* NcclComm;
enum { kNcclSuccess = 0, kNcclUint8 = 1, kNcclFloat32 = 7, kNcclBfloat16 = 9 };

struct Rccl {
    void* handle = nullptr;
    int (*GetUniqueId)(NcclId*) = nullptr;
    int (*CommInitRank)(NcclComm*, int, NcclId, int) = nullptr;
    int (*CommDestroy)(NcclComm) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, NcclComm, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    bool ok = false;
};

static char g_rccl_load_error[256] = "symbols missing";

static void rccl_bind(Rccl& r) {
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
    for (const char* n : names) {                       // the copy already mapped into the process first
        r.handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
        if (r.handle) break;
    }
    if (!r.handle) r.handle = dlopen(nullptr, RTLD_NOW);            // symbols of an already loaded copy under another name
    if (r.handle && !dlsym(r.handle, "ncclAllGather")) r.handle = nullptr;
    if (!r.handle) {
        // nothing mapped yet: load the system copy.  (A process that already holds an RCCL under a name this cannot see would
        // end up with two; say so instead of doing it silently.)
        for (int i = 0; !r.handle && i < 4; ++i) {
            dlerror();
            r.handle = dlopen(names[i], RTLD_NOW | RTLD_GLOBAL);
            if (!r.handle) {
                const char* e = dlerror();
                snprintf(g_rccl_load_error, sizeof(g_rccl_load_error), "%s", e ? e : "dlopen failed");
            } else {
                fprintf(stderr, "libteo_hip: no RCCL was mapped in this process; loaded %s\n", names[i]);
            }
        }
    }
    if (!r.handle) return;
    r.GetUniqueId = (int (*)(NcclId*))dlsym(r.handle, "ncclGetUniqueId");
    r.CommInitRank = (int (*)(NcclComm*, int, NcclId, int))dlsym(r.handle, "ncclCommInitRank");
    r.CommDestroy = (int (*)(NcclComm))dlsym(r.handle, "ncclCommDestroy");
    r.AllGather = (int (*)(const void*, void*, size_t, int, NcclComm, hipStream_t))dlsym(r.handle, "ncclAllGather");
    r.GetErrorString = (const char* (*)(int))dlsym(r.handle, "ncclGetErrorString");
    r.ok = r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.AllGather;
    if (!r.ok) snprintf(g_rccl_load_error, sizeof(g_rccl_load_error), "symbols missing");
}

static Rccl& rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] { rccl_bind(r); });
    return r;
}

static int rccl_fail(int rc, const char* what) {
    Rccl& r = rccl();
    set_error("%s: RCCL error %d (%s)", what, rc, r.GetErrorString ? r.GetErrorString(rc) : "?");
    return TEO_ERR_HIP;
}

static int need_rccl(const char* who) {
    if (rccl().ok) return TEO_OK;
    set_error("%s: librccl.so could not be loaded (%s)", who, g_rccl_load_error);
    return TEO_ERR_UNSUPPORTED;
}

}  // namespace teo

struct teo_ctx {
    int rank, world, device;
    int cu_count;
    size_t hbm_bytes;
    teo::NcclComm comm;
    teo_tune tune;          // the context's own block of performance knobs (teo_ctx_tune)
};

using namespace teo;

extern "C" {

int teo_comm_unique_id(void* out_id) {
    TEO_CHECK_ARG(out_id != nullptr, "teo_comm_unique_id: null output");
    const int rc0 = need_rccl("teo_comm_unique_id");
    if (rc0 != TEO_OK) return rc0;
    NcclId id;
    const int rc = rccl().GetUniqueId(&id);
    if (rc != kNcclSuccess) return rccl_fail(rc, "ncclGetUniqueId");
    memcpy(out_id, id.internal, TEO_COMM_ID_BYTES);
    return TEO_OK;
}

int teo_ctx_create(int rank, int world_size, const void* unique_id, int device, teo_ctx** out) {
    TEO_CHECK_ARG(out != nullptr, "teo_ctx_create: null output");
    *out = nullptr;
    TEO_CHECK_ARG(world_size >= 1 && rank >= 0 && rank < world_size, "teo_ctx_create: rank %d of %d", rank, world_size);
    TEO_CHECK_ARG(unique_id != nullptr || world_size == 1, "teo_ctx_create: a unique id is needed for world_size %d", world_size);
    hipError_t e = hipSetDevice(device);
    if (e != hipSuccess) return hip_fail(e, "teo_ctx_create: hipSetDevice");
    hipDeviceProp_t prop;
    e = hipGetDeviceProperties(&prop, device);
    if (e != hipSuccess) return hip_fail(e, "teo_ctx_create: hipGetDeviceProperties");
    const int rc0 = need_rccl("teo_ctx_create");
    if (rc0 != TEO_OK) return rc0;
    NcclId id;
    if (unique_id) {
        memcpy(id.internal, unique_id, TEO_COMM_ID_BYTES);
    } else {
        const int rc = rccl().GetUniqueId(&id);
        if (rc != kNcclSuccess) return rccl_fail(rc, "ncclGetUniqueId");
    }
    NcclComm comm = nullptr;
    const int rc = rccl().CommInitRank(&comm, world_size, id, rank);
    if (rc != kNcclSuccess) return rccl_fail(rc, "ncclCommInitRank");
    teo_ctx* c = new teo_ctx();
    c->rank = rank; c->world = world_size; c->device = device;
    c->cu_count = prop.multiProcessorCount;
    c->hbm_bytes = prop.totalGlobalMem;
    c->comm = comm;
    *out = c;
    return TEO_OK;
}

int teo_ctx_destroy(teo_ctx* ctx) {
    if (!ctx) return TEO_OK;
    int rc = kNcclSuccess;
    if (ctx->comm && rccl().ok) rc = rccl().CommDestroy(ctx->comm);
    delete ctx;
    return rc == kNcclSuccess ? TEO_OK : rccl_fail(rc, "ncclCommDestroy");
}

int teo_ctx_info(const teo_ctx* ctx, int* rank, int* world_size, int* cu_count, size_t* hbm_bytes) {
    TEO_CHECK_ARG(ctx != nullptr, "teo_ctx_info: null ctx");
    if (rank) *rank = ctx->rank;
    if (world_size) *world_size = ctx->world;
    if (cu_count) *cu_count = ctx->cu_count;
    if (hbm_bytes) *hbm_bytes = ctx->hbm_bytes;
    return TEO_OK;
}

teo_tune* teo_ctx_tune(teo_ctx* ctx) { return ctx ? &ctx->tune : nullptr; }

int teo_allgather_visual(teo_ctx* ctx, const void* d_local, void* d_out, int rows_per_rank, int dim, int dtype,
                         teo_stream_t stream) {
    TEO_CHECK_ARG(ctx != nullptr, "teo_allgather_visual: null ctx");
    TEO_CHECK_ARG(rows_per_rank >= 0 && dim > 0, "teo_allgather_visual: rows_per_rank %d dim %d", rows_per_rank, dim);
    TEO_CHECK_ARG(dtype == TEO_F32 || dtype == TEO_BF16 || dtype == TEO_F16, "teo_allgather_visual: dtype %d", dtype);      // 16-bit formats move as bits
    if (rows_per_rank == 0) return TEO_OK;
    TEO_CHECK_ARG(d_local != nullptr && d_out != nullptr, "teo_allgather_visual: null buffer");
    const size_t count = (size_t)rows_per_rank * dim;
    const int rc = rccl().AllGather(d_local, d_out, count, dtype == TEO_F32 ? kNcclFloat32 : kNcclBfloat16, ctx->comm,
                                    (hipStream_t)stream);
    if (rc != kNcclSuccess) return rccl_fail(rc, "ncclAllGather");
    return TEO_OK;
}

}  // extern "C"
