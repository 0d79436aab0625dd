// nsnp_comm.hip -- the rooted result gather of SURVEY.md 8(b) / 8(e) as a C-ABI entry on RCCL (optional path).
//
// The reference is single-device (PileupModel/predict.py:208); the only exchange of the sharded path is one rooted gather of the
// compact per-site calls at the very end.  The Python host side does it with one torch.distributed collective
// (nanosnp_amd/dist.py); this file is the same gather for callers that are not PyTorch processes: grouped ncclSend / ncclRecv to
// the root over xGMI, no padding copies, no ring all-reduce.  RCCL is resolved at run time from the process (the copy PyTorch loaded,
// if any) or from the loader path - the library has no link-time dependency on it and every other entry point works without it.
#include "nsnp_common.hpp"

#include <dlfcn.h>
#include <rccl/rccl.h>

namespace {

struct RcclApi {
    bool tried = false, ok = false;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
};
RcclApi g_rccl;

void* rccl_symbol(void*& handle, const char* name)
{
    void* p = dlsym(RTLD_DEFAULT, name);                    // the copy already in the process (PyTorch bundles its own)
    if (p) return p;
    if (!handle) handle = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!handle) handle = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    return handle ? dlsym(handle, name) : nullptr;
}

bool rccl_load()
{
    if (g_rccl.tried) return g_rccl.ok;
    g_rccl.tried = true;
    void* h = nullptr;
#define SYM(F) *(void**)(&g_rccl.F) = rccl_symbol(h, "nccl" #F)
    SYM(GetUniqueId); SYM(CommInitRank); SYM(CommDestroy); SYM(Send); SYM(Recv); SYM(GroupStart); SYM(GroupEnd);
#undef SYM
    g_rccl.ok = g_rccl.GetUniqueId && g_rccl.CommInitRank && g_rccl.CommDestroy && g_rccl.Send && g_rccl.Recv && g_rccl.GroupStart && g_rccl.GroupEnd;
    return g_rccl.ok;
}

}  // namespace

extern "C" int nsnp_comm_unique_id(uint8_t* id128)
{
    if (!id128) return NSNP_EINVAL;
    if (!rccl_load()) return NSNP_ENOTSUP;
    ncclUniqueId id;
    if (g_rccl.GetUniqueId(&id) != ncclSuccess) return NSNP_ECOMM;
    static_assert(sizeof(id) == 128, "ncclUniqueId is 128 bytes");
    memcpy(id128, &id, 128);
    return NSNP_OK;
}

extern "C" int nsnp_comm_init(nsnp_ctx* ctx, const uint8_t* id128, int rank, int world)
{
    if (!ctx || !id128 || world < 1 || rank < 0 || rank >= world) return NSNP_EINVAL;
    if (!rccl_load()) return NSNP_ENOTSUP;
    if (ctx->comm) return NSNP_EINVAL;                      // one communicator per context
    NSNP_HIP(ctx, hipSetDevice(ctx->device));
    ncclUniqueId id;
    memcpy(&id, id128, 128);
    ncclComm_t comm = nullptr;
    if (g_rccl.CommInitRank(&comm, world, id, rank) != ncclSuccess) return NSNP_ECOMM;
    ctx->comm = comm; ctx->comm_rank = rank; ctx->comm_world = world;
    return NSNP_OK;
}

extern "C" int nsnp_comm_destroy(nsnp_ctx* ctx)
{
    if (!ctx) return NSNP_EINVAL;
    if (ctx->comm && g_rccl.ok) (void)g_rccl.CommDestroy((ncclComm_t)ctx->comm);
    ctx->comm = nullptr; ctx->comm_world = 0;
    return NSNP_OK;
}

extern "C" int nsnp_gather_results(nsnp_ctx* ctx, const void* local, int64_t local_bytes, void* root_buf,
                                   const int64_t* byte_off, int root, void* stream)
{
    if (!ctx || !ctx->comm || local_bytes < 0 || root < 0 || root >= ctx->comm_world) return NSNP_EINVAL;
    const int rank = ctx->comm_rank, world = ctx->comm_world;
    hipStream_t s = (hipStream_t)stream;
    ncclComm_t comm = (ncclComm_t)ctx->comm;
    if (rank == root) {
        if (!byte_off || (!root_buf && byte_off[world] > 0)) return NSNP_EINVAL;
        if (byte_off[rank + 1] - byte_off[rank] != local_bytes) return NSNP_EINVAL;
        if (local_bytes > 0) {
            if (!local) return NSNP_EINVAL;
            NSNP_HIP(ctx, hipMemcpyAsync((char*)root_buf + byte_off[rank], local, (size_t)local_bytes, hipMemcpyDeviceToDevice, s));
        }
        if (g_rccl.GroupStart() != ncclSuccess) return NSNP_ECOMM;
        for (int r = 0; r < world; ++r) {
            const int64_t n = byte_off[r + 1] - byte_off[r];
            if (r == root || n <= 0) continue;
            if (g_rccl.Recv((char*)root_buf + byte_off[r], (size_t)n, ncclUint8, r, comm, s) != ncclSuccess) { (void)g_rccl.GroupEnd(); return NSNP_ECOMM; }
        }
        if (g_rccl.GroupEnd() != ncclSuccess) return NSNP_ECOMM;
    } else if (local_bytes > 0) {
        if (!local) return NSNP_EINVAL;
        if (g_rccl.GroupStart() != ncclSuccess) return NSNP_ECOMM;
        const ncclResult_t r1 = g_rccl.Send(local, (size_t)local_bytes, ncclUint8, root, comm, s);
        if (g_rccl.GroupEnd() != ncclSuccess || r1 != ncclSuccess) return NSNP_ECOMM;
    }
    return NSNP_OK;
}
