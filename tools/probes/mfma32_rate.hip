// Probe: how busy can v_mfma_f32_16x16x4_f32 keep a SIMD's matrix pipe in the shape of the fp32 recurrence kernels?  gfx950.
// A workgroup = 4 waves (one per SIMD); WGS workgroups per CU share the SIMDs.  Per iteration ("step") a wave issues 84 MFMAs on 4
// accumulators (operands in registers), optionally the LSTM cell of 4 hidden units (5 exp2 + 5 rcp + ~12 plain VALU each), an LDS
// write + read of 16 B, and a workgroup barrier.  Reports wall time per MFMA and SIMD (build with -DITERS=40000: the default 400 steps
// make a 0.5-1.5 ms launch whose ramp is 10 % of it).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#ifndef ITERS
#define ITERS 400
#endif
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float sg(float x) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.442695f * x)); }
__device__ __forceinline__ float th(float x) { return __builtin_fmaf(-2.0f, __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(2.88539f * x)), 1.0f); }

template <bool CELL, bool BAR, bool LDSX>
__global__ __launch_bounds__(256, 3) void k(float* out, unsigned long long* cyc, int iters)
{
    __shared__ __attribute__((aligned(16))) float xch[2][16][72];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n = lane & 15, q = lane >> 4;
    float W[84];
#pragma unroll
    for (int i = 0; i < 84; ++i) W[i] = 0.001f * (float)((i * 7 + lane) % 13) - 0.006f;
    f32x4 hb[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) hb[j] = f32x4{0.1f, 0.2f, -0.1f, 0.05f};
    float c[4] = {0.f, 0.f, 0.f, 0.f};
    if (LDSX) { for (int i = threadIdx.x; i < 2 * 16 * 72; i += 256) (&xch[0][0][0])[i] = 0.01f; }
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        const int cur = it & 1;
        if (LDSX) {
#pragma unroll
            for (int j = 0; j < 4; ++j) hb[j] = *reinterpret_cast<const f32x4*>(&xch[cur ^ 1][n][4 * q + 16 * j]);
        }
        f32x4 acc[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) acc[u] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k2 = 0; k2 < 21; ++k2)
#pragma unroll
            for (int u = 0; u < 4; ++u)
                acc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(W[4 * k2 + u], hb[k2 & 3][(k2 >> 2) & 3], acc[u], 0, 0, 0);
        f32x4 hn;
        if (CELL) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float ig = sg(acc[u][0]), fg = sg(acc[u][1]), gg = th(acc[u][2]), og = sg(acc[u][3]);
                c[u] = __builtin_fmaf(fg, c[u], ig * gg);
                hn[u] = og * th(c[u]);
            }
        } else {
#pragma unroll
            for (int u = 0; u < 4; ++u) hn[u] = acc[u][0] + acc[u][1] + acc[u][2] + acc[u][3];
        }
        if (LDSX) *reinterpret_cast<f32x4*>(&xch[cur][n][16 * wave + 4 * q]) = hn;
        else {
#pragma unroll
            for (int j = 0; j < 4; ++j) hb[j] = hn * 0.5f;
        }
        if (BAR) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) s += hb[j][0] + c[j];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (lane == 0) cyc[blockIdx.x * 4 + wave] = t1 - t0;
}

template <bool CELL, bool BAR, bool LDSX>
void run(const char* name, int wgs_per_cu)
{
    const int iters = ITERS, grid = 256 * wgs_per_cu;
    float* out; unsigned long long* cyc;
    hipMalloc(&out, grid * 256 * sizeof(float)); hipMalloc(&cyc, grid * 4 * sizeof(unsigned long long));
    hipLaunchKernelGGL((k<CELL, BAR, LDSX>), dim3(grid), dim3(256), 0, 0, out, cyc, 10); hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0); hipLaunchKernelGGL((k<CELL, BAR, LDSX>), dim3(grid), dim3(256), 0, 0, out, cyc, iters); hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(grid * 4); hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    double a = 0; for (auto v : h) a += (double)v; a /= h.size() * iters;
    // wall time is the honest figure: with several workgroups per CU the age-ordered arbiter lets waves finish at different times, so
    // the mean per-wave s_memtime span understates the kernel's length
    const double ns = ms * 1e6 / ((double)iters * 84 * wgs_per_cu);
    printf("%-46s %d WG/CU: kernel %8.3f ms = %5.2f ns per MFMA and SIMD (13.5 = the pipe's pace at 2.37 GHz: %5.1f %%), mean wave span %7.0f cycles per step\n",
           name, wgs_per_cu, ms, ns, 100.0 * 13.5 / ns, a);
    hipFree(out); hipFree(cyc);
}
int main()
{
    for (int w = 1; w <= 3; ++w) run<false, false, false>("MFMA only", w);
    for (int w = 1; w <= 3; ++w) run<true, false, false>("MFMA + cell", w);
    for (int w = 1; w <= 3; ++w) run<true, true, false>("MFMA + cell + barrier", w);
    for (int w = 1; w <= 3; ++w) run<true, true, true>("MFMA + cell + LDS exchange + barrier", w);
    for (int w = 1; w <= 3; ++w) run<false, true, true>("MFMA + LDS exchange + barrier (no cell)", w);
    return 0;
}
