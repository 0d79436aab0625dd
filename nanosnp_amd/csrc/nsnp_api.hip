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
    case NSNP_ENOTSUP: return "optional component unavailable (RCCL not found)";
    case NSNP_ECOMM: return "RCCL call failed";
    default: return "unknown error";
    }
}

static hipError_t g_create_err = hipSuccess;   // error of the last failed nsnp_ctx_create (no ctx yet)

extern "C" int nsnp_last_hip_error(const nsnp_ctx* ctx, const char** text)
{
    const hipError_t e = ctx ? ctx->last_err : g_create_err;
    if (text) *text = hipGetErrorString(e);
    return (int)e;
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
    ctx->hap_chunk = 16384;
    ctx->precision = 0;          // exact fp32 MFMA is the default of all three model forwards (the reference computes in
    ctx->hap_precision = 0;      // fp32); 1 opts into the f16x3 split (3 fp16 MFMAs per product, ~1e-6 from fp32)
    ctx->cat_precision = 0;
    ctx->cat_conv_lds = 1;
    ctx->cat_conv_pix2 = 1;
    ctx->hap_b3x = 1;
    ctx->tok_fused = 0;         // three launches, no workgroup waits for another (mpileup_tokenise.hip)
    ctx->proj1_tiles = 4;
    ctx->fused_l1 = 1;
    ctx->l0_rs = 1;
    ctx->l1_rs = 1;
    ctx->l1_stagger = 0;
    ctx->head_rs = 1;
    ctx->rs_prio = 0;           // measured without effect, kept as an option
    hipDeviceProp_t prop;
    hipError_t e = hipSetDevice(device);
    if (e == hipSuccess) e = hipGetDeviceProperties(&prop, device);
    if (e != hipSuccess) { g_create_err = e; delete ctx; return NSNP_EHIP; }
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) { delete ctx; return NSNP_EARCH; }
    ctx->n_cu = prop.multiProcessorCount;
    *out = ctx;
    return NSNP_OK;
}

void nsnp_hap_free(nsnp_ctx* ctx);   // hap_forward.hip
int  nsnp_hap_reserve(nsnp_ctx* ctx);
void nsnp_cat_free(nsnp_ctx* ctx);   // cat_forward.hip
void nsnp_tok_free(nsnp_ctx* ctx);   // mpileup_tokenise.hip

// ---- per-kernel timing ----------------------------------------------------------------------------
constexpr size_t TIMER_MAX_PAIRS = 8192;

ScopedKernelTimer::ScopedKernelTimer(nsnp_ctx* c, int kernel, hipStream_t stream)
    : ctx(c), k(kernel), s(stream), stop_ev(nullptr), on(false)
{
    KernelTimer* t = c->timer;
    if (!t || !t->enabled || t->used[k] >= TIMER_MAX_PAIRS) return;
    if (t->used[k] == t->start[k].size()) {
        hipEvent_t a, b;
        if (hipEventCreate(&a) != hipSuccess) return;
        if (hipEventCreate(&b) != hipSuccess) { (void)hipEventDestroy(a); return; }
        t->start[k].push_back(a); t->stop[k].push_back(b);
    }
    const size_t i = t->used[k]++;
    (void)hipEventRecord(t->start[k][i], s);
    stop_ev = t->stop[k][i];
    on = true;
}
void ScopedKernelTimer::stop() { if (on) { (void)hipEventRecord(stop_ev, s); on = false; } }
ScopedKernelTimer::~ScopedKernelTimer() { stop(); }

extern "C" int nsnp_ctx_set_option(nsnp_ctx* ctx, const char* name, int64_t value)
{
    if (!ctx || !name) return NSNP_EINVAL;
    if (strcmp(name, "cat_precision") == 0) {
        if (value != 0 && value != 1 && value != 2) return NSNP_EINVAL;
        ctx->cat_precision = (int)value;
        return NSNP_OK;
    }
    if (strcmp(name, "tok_fused") == 0) {
        if (value != 0 && value != 1) return NSNP_EINVAL;
        ctx->tok_fused = (int)value;
        return NSNP_OK;
    }
    if (strcmp(name, "hap_b3x") == 0) {
        if (value != 0 && value != 1) return NSNP_EINVAL;
        ctx->hap_b3x = (int)value;
        return NSNP_OK;
    }
    if (strcmp(name, "cat_conv_pix2") == 0) {
        if (value != 0 && value != 1) return NSNP_EINVAL;
        ctx->cat_conv_pix2 = (int)value;
        return NSNP_OK;
    }
    if (strcmp(name, "cat_conv_lds") == 0) {
        if (value != 0 && value != 1) return NSNP_EINVAL;
        ctx->cat_conv_lds = (int)value;
        return NSNP_OK;
    }
    if (strcmp(name, "hap_precision") == 0) {
        if (value != 0 && value != 1 && value != 2) return NSNP_EINVAL;
        ctx->hap_precision = (int)value;
        return nsnp_hap_reserve(ctx);           // (bf16x3 activation images are 1.5 x the fp32 ones: the workspace grows here, synchronously, never in the forward)
    }
    if (strcmp(name, "hap_pass_sites") == 0) {
        if (value < 128 || value > 131072 || value % 128) return NSNP_EINVAL;
        // synchronous re-allocation when the weights are already loaded; when the new workspace cannot be had, the previous pass size and
        // its workspace come back (the context keeps working at the old size instead of failing every later forward with ENOMEM)
        const int64_t old = ctx->hap_chunk;
        ctx->hap_chunk = value;
        const int rc = nsnp_hap_reserve(ctx);
        if (rc) { ctx->hap_chunk = old; (void)nsnp_hap_reserve(ctx); }
        return rc;
    }
    if (strcmp(name, "pileup_precision") == 0) {
        if (value != 0 && value != 1 && value != 2) return NSNP_EINVAL;
        ctx->precision = (int)value;
        // the bf16x3 layout of H0 is 1.5 x the fp32 one: a workspace that is already reserved grows now (synchronous), not inside a forward
        if (value == 2 && ctx->ws_h0 && ctx->ws_h0_floats < 192) return nsnp_ctx_reserve(ctx, ctx->chunk_sites);
        return NSNP_OK;
    }
    if (strcmp(name, "fused_l1") == 0) {
        if (value != 0 && value != 1) return NSNP_EINVAL;
        ctx->fused_l1 = (int)value;
        return NSNP_OK;
    }
    if (strcmp(name, "l0_register_stationary") == 0) {
        if (value != 0 && value != 1) return NSNP_EINVAL;
        ctx->l0_rs = (int)value;
        return NSNP_OK;
    }
    if (strcmp(name, "l1_register_stationary") == 0) {
        if (value != 0 && value != 1 && value != 2) return NSNP_EINVAL;
        ctx->l1_rs = (int)value;
        return NSNP_OK;
    }
    if (strcmp(name, "l0_input_weights_in_lds") == 0) {
        if (value != 0 && value != 1) return NSNP_EINVAL;
        ctx->l0_wx_lds = (int)value;
        return NSNP_OK;
    }
    if (strcmp(name, "static_priority") == 0) {
        if (value < 0 || value > 3) return NSNP_EINVAL;
        ctx->rs_prio = (int)value;
        return NSNP_OK;
    }
    if (strcmp(name, "head_split") == 0) {
        if (value != 0 && value != 1) return NSNP_EINVAL;
        ctx->head_rs = (int)value;
        return NSNP_OK;
    }
    if (strcmp(name, "l1_stagger") == 0) {
        if (value != 0 && value != 1) return NSNP_EINVAL;
        ctx->l1_stagger = (int)value;
        return NSNP_OK;
    }
    if (strcmp(name, "l1_site_groups") == 0) {
        if (value != 0 && value != 1 && value != 2 && value != 4) return NSNP_EINVAL;
        ctx->l1_rs_groups = (int)value;
        return NSNP_OK;
    }
    if (strcmp(name, "l0_site_groups") == 0) {
        if (value != 0 && value != 1 && value != 2 && value != 4) return NSNP_EINVAL;
        ctx->l0_rs_groups = (int)value;
        return NSNP_OK;
    }
    if (strcmp(name, "fused_waves") == 0) {
        if (value != 0 && value != 4 && value != 8 && value != 12) return NSNP_EINVAL;
        ctx->fused_waves = (int)value;
        return NSNP_OK;
    }
    if (strcmp(name, "proj1_tiles") == 0) {
        if (value < 1 || value > 64) return NSNP_EINVAL;
        ctx->proj1_tiles = (int)value;
        return NSNP_OK;
    }
    if (strcmp(name, "recurrence_waves") == 0) {
        // every kernel the recurrence grid feeds has 1/2/4/8-wave builds only
        if (value != 0 && value != 1 && value != 2 && value != 4 && value != 8) return NSNP_EINVAL;
        ctx->force_wpb = (int)value;
        return NSNP_OK;
    }
    return NSNP_EINVAL;
}

extern "C" int nsnp_ctx_enable_timing(nsnp_ctx* ctx, int enable)
{
    if (!ctx) return NSNP_EINVAL;
    if (!ctx->timer) {
        ctx->timer = new (std::nothrow) KernelTimer();
        if (!ctx->timer) return NSNP_ENOMEM;
        for (int k = 0; k < NSNP_K_COUNT; ++k) ctx->timer->used[k] = 0;
    }
    ctx->timer->enabled = enable != 0;
    return NSNP_OK;
}

extern "C" int nsnp_ctx_read_timing(nsnp_ctx* ctx, int kernel, double* total_ms, int64_t* launches)
{
    if (!ctx || kernel < 0 || kernel >= NSNP_K_COUNT || !total_ms || !launches) return NSNP_EINVAL;
    *total_ms = 0.0; *launches = 0;
    KernelTimer* t = ctx->timer;
    if (!t) return NSNP_OK;
    for (size_t i = 0; i < t->used[kernel]; ++i) {
        NSNP_HIP(ctx, hipEventSynchronize(t->stop[kernel][i]));
        float ms = 0.f;
        NSNP_HIP(ctx, hipEventElapsedTime(&ms, t->start[kernel][i], t->stop[kernel][i]));
        *total_ms += ms; ++*launches;
    }
    t->used[kernel] = 0;   // events are kept for reuse
    return NSNP_OK;
}

static void free_ws(nsnp_ctx* ctx)
{
    if (ctx->ws_h0) (void)hipFree(ctx->ws_h0);
    if (ctx->ws_xp1) (void)hipFree(ctx->ws_xp1);
    if (ctx->ws_h1c) (void)hipFree(ctx->ws_h1c);
    ctx->ws_h0 = ctx->ws_xp1 = ctx->ws_h1c = nullptr;
    ctx->ws_h0_floats = 0;
}

extern "C" int nsnp_ctx_destroy(nsnp_ctx* ctx)
{
    if (!ctx) return NSNP_EINVAL;
    (void)hipSetDevice(ctx->device);
    (void)hipDeviceSynchronize();
    free_ws(ctx);
    if (ctx->pw.arena) (void)hipFree(ctx->pw.arena);
    if (ctx->pw16.arena) (void)hipFree(ctx->pw16.arena);
    if (ctx->pw16.l1f_bias) (void)hipFree(ctx->pw16.l1f_bias);
    if (ctx->pwb3.arena) (void)hipFree(ctx->pwb3.arena);
    if (ctx->sel_tmp) (void)hipFree(ctx->sel_tmp);
    nsnp_tok_free(ctx);
    (void)nsnp_comm_destroy(ctx);
    nsnp_hap_free(ctx);
    nsnp_cat_free(ctx);
    if (ctx->timer) {
        for (int k = 0; k < NSNP_K_COUNT; ++k) {
            for (hipEvent_t e : ctx->timer->start[k]) (void)hipEventDestroy(e);
            for (hipEvent_t e : ctx->timer->stop[k]) (void)hipEventDestroy(e);
        }
        delete ctx->timer;
    }
    delete ctx;
    return NSNP_OK;
}

extern "C" int nsnp_ctx_reserve(nsnp_ctx* ctx, int64_t max_sites)
{
    if (!ctx || max_sites <= 0) return NSNP_EINVAL;
    // H0: 512 B per site and step on the fp32 / f16x3 paths, 768 B (three bf16 planes) on the bf16x3 path: the larger layout is
    // allocated only for a context that has selected pileup_precision 2 (the option re-reserves when it is set: below)
    const int h0_floats = ctx->precision == 2 ? 192 : 128;
    if (ctx->ws_h0 && ctx->chunk_sites == max_sites && ctx->ws_h0_floats >= h0_floats) return NSNP_OK;
    NSNP_HIP(ctx, hipSetDevice(ctx->device));
    NSNP_HIP(ctx, hipDeviceSynchronize());
    free_ws(ctx);
    ctx->chunk_sites = max_sites;
    const size_t n = (size_t)max_sites;
    NSNP_HIP(ctx, hipMalloc((void**)&ctx->ws_h0, (n + 64) * PW * h0_floats * sizeof(float)));
    ctx->ws_h0_floats = h0_floats;
    // ws_xp1 (34.8 KB per site) only exists for the legacy unfused layer-1 path: nsnp_ctx_need_xp1 allocates it on first use
    NSNP_HIP(ctx, hipMalloc((void**)&ctx->ws_h1c, (n + 64) * 128 * sizeof(float)));
    return NSNP_OK;
}

int nsnp_ctx_need_xp1(nsnp_ctx* ctx)
{
    if (ctx->ws_xp1) return NSNP_OK;
    NSNP_HIP(ctx, hipMalloc((void**)&ctx->ws_xp1, 2 * (size_t)ctx->chunk_sites * PSTEPS1 * 256 * sizeof(float)));
    return NSNP_OK;
}

extern "C" int nsnp_pileup_load_weights(nsnp_ctx* ctx, const float* const* host_tensors, int n_tensors)
{
    if (!ctx || !host_tensors || n_tensors < 24) return NSNP_EINVAL;
    for (int i = 0; i < 24; ++i) if (!host_tensors[i]) return NSNP_EINVAL;
    NSNP_HIP(ctx, hipSetDevice(ctx->device));
    int rc = nsnp_pileup_pack_weights(ctx, host_tensors);
    if (!rc) rc = nsnp_pileup_pack_weights_f16(ctx, host_tensors);
    return rc ? rc : nsnp_pileup_pack_weights_bf16(ctx, host_tensors);
}

// one dispatch for the three arithmetic modes; post / post_written as in nsnp_common.hpp
static int pileup_forward_any(nsnp_ctx* ctx, const int32_t* x, const int64_t* center_idx, int64_t N, float* gt, float* zy,
                              const PostOut* post, bool* post_written, hipStream_t s)
{
    if (post_written) *post_written = false;
    if (ctx->precision == 1) return nsnp_pileup_forward_f16x3(ctx, x, center_idx, N, gt, zy, s);
    if (ctx->precision == 2) {
        const int rc = nsnp_pileup_forward_bf16x3(ctx, x, center_idx, N, gt, zy, post, s);
        if (!rc && post_written) *post_written = post && post->gt_arg;
        return rc;
    }
    return nsnp_pileup_forward_impl(ctx, x, center_idx, N, gt, zy, post, post_written, s);
}

extern "C" int nsnp_pileup_forward(nsnp_ctx* ctx, const int32_t* x, int64_t N,
                                   float* gt_prob, float* zy_prob, void* stream)
{
    if (!ctx || N < 0 || (N > 0 && (!x || !gt_prob || !zy_prob))) return NSNP_EINVAL;
    return pileup_forward_any(ctx, x, nullptr, N, gt_prob, zy_prob, nullptr, nullptr, (hipStream_t)stream);
}

extern "C" int nsnp_pileup_forward_windows(nsnp_ctx* ctx, const int32_t* counts, const int64_t* center_idx,
                                           int64_t N, float* gt_prob, float* zy_prob, void* stream)
{
    if (!ctx || N < 0 || (N > 0 && (!counts || !center_idx || !gt_prob || !zy_prob))) return NSNP_EINVAL;
    return pileup_forward_any(ctx, counts, center_idx, N, gt_prob, zy_prob, nullptr, nullptr, (hipStream_t)stream);
}

// call rows [n][13] float64 (position, gt argmax, zy argmax, gt max, zy max, the eight coverage channels of predict.py:63) -> the typed
// arrays the row formatter takes; one thread per site, 104 bytes read, 41 written - to the device or straight into pinned host memory
__global__ void k_rows_unpack(const double* __restrict__ rows, int64_t n, int64_t* __restrict__ pos, uint8_t* __restrict__ ga, uint8_t* __restrict__ za,
                              float* __restrict__ gm, float* __restrict__ zm, float* __restrict__ cov)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double* r = rows + i * 13;
    pos[i] = (int64_t)r[0];
    ga[i] = (uint8_t)r[1]; za[i] = (uint8_t)r[2];
    gm[i] = (float)r[3]; zm[i] = (float)r[4];
#pragma unroll
    for (int c = 0; c < 8; ++c) cov[i * 8 + c] = (float)r[5 + c];
}

extern "C" int nsnp_pileup_rows_unpack(nsnp_ctx* ctx, const double* rows, int64_t N, int64_t* pos, uint8_t* gt_arg, uint8_t* zy_arg,
                                       float* gt_max, float* zy_max, float* cov8, void* stream)
{
    if (!ctx || N < 0 || (N > 0 && (!rows || !pos || !gt_arg || !zy_arg || !gt_max || !zy_max || !cov8))) return NSNP_EINVAL;
    if (N == 0) return NSNP_OK;
    hipLaunchKernelGGL(k_rows_unpack, dim3((unsigned)NSNP_CDIV(N, (int64_t)256)), dim3(256), 0, (hipStream_t)stream, rows, N, pos, gt_arg, zy_arg, gt_max,
                       zy_max, cov8);
    NSNP_HIP(ctx, hipGetLastError());
    return NSNP_OK;
}

extern "C" int nsnp_pileup_forward_windows_calls(nsnp_ctx* ctx, const int32_t* counts, const int64_t* center_idx, int64_t N,
                                                 float* gt_prob, float* zy_prob, uint8_t* gt_arg, uint8_t* zy_arg,
                                                 float* gt_max, float* zy_max, void* stream)
{
    if (!ctx || N < 0 || (N > 0 && (!counts || !center_idx || !gt_prob || !zy_prob || !gt_arg || !zy_arg || !gt_max || !zy_max))) return NSNP_EINVAL;
    if (N == 0) return NSNP_OK;
    // the fp32 and bf16x3 heads kernels write argmax / max from the registers that hold the probabilities; the other paths (f16x3,
    // the one-wave heads kernel of round 1) run the forward and then nsnp_pileup_postprocess on the same stream
    const PostOut post{gt_arg, zy_arg, gt_max, zy_max};
    bool written = false;
    const int rc = pileup_forward_any(ctx, counts, center_idx, N, gt_prob, zy_prob, &post, &written, (hipStream_t)stream);
    if (rc) return rc;
    return written ? NSNP_OK : nsnp_pileup_postprocess(ctx, gt_prob, zy_prob, nullptr, N, gt_arg, zy_arg, gt_max, zy_max, nullptr, stream);
}
