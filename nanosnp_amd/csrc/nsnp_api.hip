// nsnp_api.hip -- context management and the C-ABI entry points of include/nanosnp.h.
#include "nsnp_common.hpp"

#include <new>

extern "C" int nsnp_version(void) { return 100; }

extern "C" const char* nsnp_strerror(int code)
{
    switch (code) {
    case NSNP_OK: return "ok";
    case NSNP_EINVAL: return "invalid argument";
    case NSNP_ENOMEM: return "out of memory";
    case NSNP_EHIP: return "HIP runtime error";
    case NSNP_ENOWEIGHTS: return "weights not loaded";
    case NSNP_EARCH: return "device is not gfx950";
    case NSNP_ESHAPE: return "unsupported model dimensions";
    default: return "unknown error";
    }
}

extern "C" int nsnp_last_hip_error(const nsnp_ctx* ctx, const char** text)
{
    if (!ctx) return -1;
    if (text) *text = hipGetErrorString(ctx->last_err);
    return (int)ctx->last_err;
}

extern "C" int nsnp_ctx_create(int device, nsnp_ctx** out)
{
    if (!out) return NSNP_EINVAL;
    *out = nullptr;
    nsnp_ctx* ctx = new (std::nothrow) nsnp_ctx();
    if (!ctx) return NSNP_ENOMEM;
    memset((void*)ctx, 0, sizeof(*ctx));
    ctx->device = device;
    ctx->last_err = hipSuccess;
    ctx->chunk_sites = 32768;
    hipDeviceProp_t prop;
    hipError_t e = hipSetDevice(device);
    if (e == hipSuccess) e = hipGetDeviceProperties(&prop, device);
    if (e != hipSuccess) { delete ctx; return NSNP_EHIP; }
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) { delete ctx; return NSNP_EARCH; }
    ctx->n_cu = prop.multiProcessorCount;
    *out = ctx;
    return NSNP_OK;
}

void nsnp_hap_free(nsnp_ctx* ctx);   // hap_forward.hip

static void free_ws(nsnp_ctx* ctx)
{
    if (ctx->ws_h0) (void)hipFree(ctx->ws_h0);
    if (ctx->ws_xp1) (void)hipFree(ctx->ws_xp1);
    if (ctx->ws_h1c) (void)hipFree(ctx->ws_h1c);
    ctx->ws_h0 = ctx->ws_xp1 = ctx->ws_h1c = nullptr;
}

extern "C" int nsnp_ctx_destroy(nsnp_ctx* ctx)
{
    if (!ctx) return NSNP_EINVAL;
    (void)hipSetDevice(ctx->device);
    (void)hipDeviceSynchronize();
    free_ws(ctx);
    if (ctx->pw.arena) (void)hipFree(ctx->pw.arena);
    if (ctx->sel_tmp) (void)hipFree(ctx->sel_tmp);
    nsnp_hap_free(ctx);
    delete ctx;
    return NSNP_OK;
}

extern "C" int nsnp_ctx_reserve(nsnp_ctx* ctx, int64_t max_sites)
{
    if (!ctx || max_sites <= 0) return NSNP_EINVAL;
    if (ctx->ws_h0 && ctx->chunk_sites == max_sites) return NSNP_OK;
    NSNP_HIP(ctx, hipSetDevice(ctx->device));
    NSNP_HIP(ctx, hipDeviceSynchronize());
    free_ws(ctx);
    ctx->chunk_sites = max_sites;
    const size_t n = (size_t)max_sites;
    NSNP_HIP(ctx, hipMalloc((void**)&ctx->ws_h0, n * PW * 128 * sizeof(float)));
    NSNP_HIP(ctx, hipMalloc((void**)&ctx->ws_xp1, 2 * n * PSTEPS1 * 256 * sizeof(float)));
    NSNP_HIP(ctx, hipMalloc((void**)&ctx->ws_h1c, n * 128 * sizeof(float)));
    return NSNP_OK;
}

extern "C" int nsnp_pileup_load_weights(nsnp_ctx* ctx, const float* const* host_tensors, int n_tensors)
{
    if (!ctx || !host_tensors || n_tensors < 24) return NSNP_EINVAL;
    for (int i = 0; i < 24; ++i) if (!host_tensors[i]) return NSNP_EINVAL;
    NSNP_HIP(ctx, hipSetDevice(ctx->device));
    return nsnp_pileup_pack_weights(ctx, host_tensors);
}

extern "C" int nsnp_pileup_forward(nsnp_ctx* ctx, const int32_t* x, int64_t N,
                                   float* gt_prob, float* zy_prob, void* stream)
{
    if (!ctx || N < 0 || (N > 0 && (!x || !gt_prob || !zy_prob))) return NSNP_EINVAL;
    return nsnp_pileup_forward_impl(ctx, x, nullptr, N, gt_prob, zy_prob, (hipStream_t)stream);
}

extern "C" int nsnp_pileup_forward_windows(nsnp_ctx* ctx, const int32_t* counts, const int64_t* center_idx,
                                           int64_t N, float* gt_prob, float* zy_prob, void* stream)
{
    if (!ctx || N < 0 || (N > 0 && (!counts || !center_idx || !gt_prob || !zy_prob))) return NSNP_EINVAL;
    return nsnp_pileup_forward_impl(ctx, counts, center_idx, N, gt_prob, zy_prob, (hipStream_t)stream);
}
