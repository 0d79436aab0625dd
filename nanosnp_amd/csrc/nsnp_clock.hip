// nsnp_clock.hip -- nsnp_ctx_shader_clock: the shader clock the chip holds under a full-chip fp32 MFMA load.
// Boxes of the pool differ (2.12 - 2.38 GHz measured in round 3) and every MFMA fraction of a bench line is priced at the
// 2.4 GHz peak, so a line carries the clock it was measured at.  Two workgroups of four waves on every CU run a dependent
// stream of v_mfma_f32_16x16x4_f32 (the instruction of the fp32 recurrence kernels) for about a millisecond; thread 0 of every
// workgroup stamps s_memtime (shader cycles) and s_memrealtime (100 MHz) at its first and last instruction
// (MI355X_MICROARCH.md "DVFS give-back" item 6); clock = 100 MHz x sum(cycles) / sum(ticks).  Diagnostic only: no product
// path depends on it.
#include "nsnp_common.hpp"

namespace {

__global__ __launch_bounds__(256) void k_clock_probe(unsigned long long* __restrict__ out, int iters, float seed)
{
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    f32x4 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = f32x4{seed, 0.f, 0.f, 0.f};
    const float a = seed * (float)(threadIdx.x & 7), b = 1.0f - seed;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) {
        out[2 * blockIdx.x] = t1 - t0;
        out[2 * blockIdx.x + 1] = r1 - r0;
    }
    if (s == 12345.678f) out[0] = 0;          // keeps the MFMA stream alive
}

}  // namespace

extern "C" int nsnp_ctx_shader_clock(nsnp_ctx* ctx, double* mhz, void* stream)
{
    if (!ctx || !mhz) return NSNP_EINVAL;
    *mhz = 0.0;
    NSNP_HIP(ctx, hipSetDevice(ctx->device));
    const int wgs = 2 * (ctx->n_cu > 0 ? ctx->n_cu : 256);
    unsigned long long* d = nullptr;
    NSNP_HIP(ctx, hipMalloc((void**)&d, sizeof(unsigned long long) * 2 * wgs));
    hipStream_t s = (hipStream_t)stream;
    // 2 waves per SIMD x 4 MFMAs x 32 cycles per trip: 20000 trips = 5.1 M cycles = ~2 ms at 2.4 GHz
    for (int rep = 0; rep < 2; ++rep)          // the first launch lets the clock settle, the second is read
        hipLaunchKernelGGL(k_clock_probe, dim3(wgs), dim3(256), 0, s, d, 20000, 0.001f);
    std::vector<unsigned long long> h(2 * (size_t)wgs);
    hipError_t e = hipStreamSynchronize(s);
    if (e == hipSuccess) e = hipMemcpy(h.data(), d, sizeof(unsigned long long) * 2 * wgs, hipMemcpyDeviceToHost);
    (void)hipFree(d);
    if (e != hipSuccess) { ctx->last_err = e; return NSNP_EHIP; }
    double cyc = 0.0, ticks = 0.0;
    for (int i = 0; i < wgs; ++i) {
        if (h[2 * i] >= (1ull << 40) || h[2 * i + 1] >= (1ull << 40) || h[2 * i + 1] == 0) continue;    // a counter that wrapped
        cyc += (double)h[2 * i]; ticks += (double)h[2 * i + 1];
    }
    if (ticks <= 0.0) return NSNP_EHIP;
    *mhz = 100.0 * cyc / ticks;
    return NSNP_OK;
}
