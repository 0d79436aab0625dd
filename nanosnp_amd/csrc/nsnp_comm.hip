// nsnp_comm.hip -- the rooted result gather of SURVEY.md 8(b) / 8(e) as a C-ABI entry on RCCL (optional path).
//
// The reference is single-device (PileupModel/predict.py:208); the only exchange of the sharded path is one rooted gather of the
// compact per-site calls at the very end.  The Python host side does it with one torch.distributed collective
// (nanosnp_amd/dist.py); this file is the same gather for callers that are not PyTorch processes: grouped ncclSend / ncclRecv to
// the root over xGMI, no padding copies, no ring all-reduce.  RCCL is resolved at run time from the process (the copy PyTorch loaded,
// if any) or from the loader path - the library has no link-time dependency on it and every other entry point works without it.
// A host that already owns an ncclComm_t hands it over with nsnp_comm_attach (nsnp_gather_results(ctx, rccl_comm, ..)).
// Hardware status: validated at world size 1 only (the development pool has one-GPU boxes); argument validation is covered
// on CPU by tests/test_abi.py.
#include "nsnp_common.hpp"

#include <dlfcn.h>
#include <mutex>
#include <rccl/rccl.h>

namespace {

struct RcclApi {
    bool ok = false;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
};
RcclApi g_rccl;
std::once_flag g_rccl_once;

// Every symbol comes from ONE library image: the copy already mapped into the process (PyTorch bundles its own librccl and a
// second copy would own a second set of communicator tables), found by asking the loader which object defines ncclCommInitRank
// and re-opening exactly that file; only a process without RCCL opens the loader path's librccl.
void rccl_resolve()
{
    void* h = nullptr;
    if (void* p = dlsym(RTLD_DEFAULT, "ncclCommInitRank")) {
        Dl_info info;
        if (dladdr(p, &info) && info.dli_fname) h = dlopen(info.dli_fname, RTLD_NOW | RTLD_NOLOAD);
    }
    if (!h) {
        // a copy loaded RTLD_LOCAL (torch's extension modules) is invisible to RTLD_DEFAULT: RTLD_NOLOAD finds it by soname
        const char* names[] = {"librccl.so.1", "librccl.so"};
        for (const char* n : names) if (!h) h = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
        for (const char* n : names) if (!h) h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    }
    if (!h) return;
#define SYM(F) *(void**)(&g_rccl.F) = dlsym(h, "nccl" #F)
    SYM(GetUniqueId); SYM(CommInitRank); SYM(CommDestroy); SYM(Send); SYM(Recv); SYM(GroupStart); SYM(GroupEnd);
#undef SYM
    g_rccl.ok = g_rccl.GetUniqueId && g_rccl.CommInitRank && g_rccl.CommDestroy && g_rccl.Send && g_rccl.Recv && g_rccl.GroupStart && g_rccl.GroupEnd;
}

bool rccl_load()
{
    std::call_once(g_rccl_once, rccl_resolve);
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
    ctx->comm = comm; ctx->comm_rank = rank; ctx->comm_world = world; ctx->comm_borrowed = false;
    return NSNP_OK;
}

extern "C" int nsnp_comm_attach(nsnp_ctx* ctx, void* rccl_comm, int rank, int world)
{
    if (!ctx || !rccl_comm || world < 1 || rank < 0 || rank >= world) return NSNP_EINVAL;
    if (!rccl_load()) return NSNP_ENOTSUP;                  // the caller's ncclComm_t must belong to the RCCL image resolved here
    if (ctx->comm) return NSNP_EINVAL;
    ctx->comm = rccl_comm; ctx->comm_rank = rank; ctx->comm_world = world; ctx->comm_borrowed = true;
    return NSNP_OK;
}

extern "C" int nsnp_comm_destroy(nsnp_ctx* ctx)
{
    if (!ctx) return NSNP_EINVAL;
    if (ctx->comm && !ctx->comm_borrowed && g_rccl.ok) {
        (void)hipSetDevice(ctx->device);                    // may run from a destructor at interpreter teardown, on any current device
        (void)g_rccl.CommDestroy((ncclComm_t)ctx->comm);
    }
    ctx->comm = nullptr; ctx->comm_world = 0; ctx->comm_borrowed = false;
    return NSNP_OK;
}

extern "C" int nsnp_gather_check(int rank, int world, int64_t local_bytes, const int64_t* byte_off, int root)
{
    // the argument rules of nsnp_gather_results that need neither a device nor a communicator (pure: callable anywhere)
    if (world < 1 || rank < 0 || rank >= world || root < 0 || root >= world || !byte_off || local_bytes < 0) return NSNP_EINVAL;
    if (byte_off[0] != 0) return NSNP_EINVAL;
    for (int r = 0; r < world; ++r) if (byte_off[r + 1] < byte_off[r]) return NSNP_EINVAL;
    if (byte_off[rank + 1] - byte_off[rank] != local_bytes) return NSNP_EINVAL;
    return NSNP_OK;
}

extern "C" int nsnp_gather_results(nsnp_ctx* ctx, const void* local, int64_t local_bytes, void* root_buf,
                                   const int64_t* byte_off, int root, void* stream)
{
    // Every rank checks the SAME things on the SAME table before anything is posted, so a bad call fails on all ranks alike
    // and no peer is left waiting in a send or receive whose partner returned early.
    if (!ctx || !ctx->comm) return NSNP_EINVAL;
    const int rank = ctx->comm_rank, world = ctx->comm_world;
    if (const int rc = nsnp_gather_check(rank, world, local_bytes, byte_off, root)) return rc;
    if (local_bytes > 0 && !local) return NSNP_EINVAL;
    if (rank == root && !root_buf && byte_off[world] > 0) return NSNP_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    ncclComm_t comm = (ncclComm_t)ctx->comm;
    if (rank == root) {
        if (local_bytes > 0)
            NSNP_HIP(ctx, hipMemcpyAsync((char*)root_buf + byte_off[rank], local, (size_t)local_bytes, hipMemcpyDeviceToDevice, s));
        if (world == 1) return NSNP_OK;
        if (g_rccl.GroupStart() != ncclSuccess) return NSNP_ECOMM;
        for (int r = 0; r < world; ++r) {
            const int64_t n = byte_off[r + 1] - byte_off[r];
            if (r == root || n <= 0) continue;
            if (g_rccl.Recv((char*)root_buf + byte_off[r], (size_t)n, ncclUint8, r, comm, s) != ncclSuccess) { (void)g_rccl.GroupEnd(); return NSNP_ECOMM; }
        }
        if (g_rccl.GroupEnd() != ncclSuccess) return NSNP_ECOMM;
    } else if (local_bytes > 0) {
        if (g_rccl.GroupStart() != ncclSuccess) return NSNP_ECOMM;
        const ncclResult_t r1 = g_rccl.Send(local, (size_t)local_bytes, ncclUint8, root, comm, s);
        if (g_rccl.GroupEnd() != ncclSuccess || r1 != ncclSuccess) return NSNP_ECOMM;
    }
    return NSNP_OK;
}
