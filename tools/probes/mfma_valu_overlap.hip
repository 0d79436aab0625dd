// Probe: does VALU / transcendental work of one wave overlap the fp32 MFMAs of ANOTHER wave on the same SIMD (gfx950)?
// A workgroup = 8 waves = two per SIMD.  Waves 0-3 run a pure v_mfma_f32_16x16x4_f32 loop, waves 4-7 run a loop of one VALU
// operation type on 8 independent registers.  Each class is timed with s_memtime while the other class is still running
// (the other class's loop is made ~3x longer), and alone (the other class idle).  One workgroup per CU, 256 workgroups.
//   cycles per MFMA  beside <op>   vs alone (32 = the pipe's pace)
//   cycles per <op>  beside MFMAs  vs alone
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

enum { OP_NONE = 0, OP_FMA, OP_EXP, OP_RCP, OP_MIX, OP_PKFMA };
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int OP, int SHAPE, int PRIO>
__global__ __launch_bounds__(512) void k(float* out, unsigned long long* cyc, int it_mfma, int it_valu)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float s = 0.f;
    unsigned long long t0 = 0, t1 = 0;
    if (wave < 4) {
        float a[8], b[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) { a[i] = 0.001f * (float)((i * 7 + lane) % 13) - 0.006f; b[i] = 0.002f * (float)((i * 5 + lane) % 11) - 0.01f; }
        if (SHAPE == 16) {
            f32x4 acc[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) acc[u] = f32x4{0.f, 0.f, 0.f, 0.f};
            t0 = __builtin_amdgcn_s_memtime();
            for (int it = 0; it < it_mfma; ++it)
#pragma unroll
                for (int r = 0; r < 16; ++r)
#pragma unroll
                    for (int u = 0; u < 4; ++u) acc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[(r + u) & 7], b[r & 7], acc[u], 0, 0, 0);
            t1 = __builtin_amdgcn_s_memtime();
#pragma unroll
            for (int u = 0; u < 4; ++u) s += acc[u][0] + acc[u][3];
        } else if (SHAPE == 1632) {
            // v_mfma_f32_16x16x32_f16: the same 8 passes as 16x16x4_f32, on the fp16 matrix path
            typedef _Float16 h8 __attribute__((ext_vector_type(8)));
            h8 ha[4], hb[4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int e = 0; e < 8; ++e) { ha[i][e] = (_Float16)a[(i + e) & 7]; hb[i][e] = (_Float16)b[(i + e) & 7]; }
            f32x4 acc[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) acc[u] = f32x4{0.f, 0.f, 0.f, 0.f};
            t0 = __builtin_amdgcn_s_memtime();
            for (int it = 0; it < it_mfma; ++it)
#pragma unroll
                for (int r = 0; r < 16; ++r)
#pragma unroll
                    for (int u = 0; u < 4; ++u) acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha[(r + u) & 3], hb[r & 3], acc[u], 0, 0, 0);
            t1 = __builtin_amdgcn_s_memtime();
#pragma unroll
            for (int u = 0; u < 4; ++u) s += acc[u][0] + acc[u][3];
        } else {
            f32x16 acc[4];
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[u][r] = 0.f;
            t0 = __builtin_amdgcn_s_memtime();
            for (int it = 0; it < it_mfma; ++it)      // 32 MFMAs of 32x32x2 = the flops and pipe time of 64 of 16x16x4
#pragma unroll
                for (int r = 0; r < 8; ++r)
#pragma unroll
                    for (int u = 0; u < 4; ++u) acc[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(r + u) & 7], b[r & 7], acc[u], 0, 0, 0);
            t1 = __builtin_amdgcn_s_memtime();
#pragma unroll
            for (int u = 0; u < 4; ++u) s += acc[u][0] + acc[u][15];
        }
    } else {
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = 0.01f * (float)(lane + i) - 0.3f;
        if (PRIO) __builtin_amdgcn_s_setprio(3);
        t0 = __builtin_amdgcn_s_memtime();
        for (int it = 0; it < it_valu; ++it) {
#pragma unroll
            for (int r = 0; r < 8; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    if (OP == OP_FMA) v[i] = __builtin_fmaf(v[i], 0.999f, 0.001f);
                    else if (OP == OP_EXP) v[i] = __builtin_amdgcn_exp2f(v[i]) - 1.0f;           // exp + one plain op
                    else if (OP == OP_RCP) v[i] = __builtin_amdgcn_rcpf(v[i]) * 0.5f + 1.0f;    // rcp + one fma
                    else if (OP == OP_MIX) v[i] = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.442695f * v[i]));   // a sigmoid: mul, exp, add, rcp
                }
            if (OP == OP_PKFMA) {
#pragma unroll
                for (int r = 0; r < 8; ++r)
#pragma unroll
                    for (int i = 0; i < 8; i += 2) {
                        f32x2 x = {v[i], v[i + 1]};
                        x = __builtin_elementwise_fma(x, f32x2{0.999f, 0.998f}, f32x2{0.001f, 0.002f});
                        v[i] = x[0]; v[i + 1] = x[1];
                    }
            }
        }
        t1 = __builtin_amdgcn_s_memtime();
#pragma unroll
        for (int i = 0; i < 8; ++i) s += v[i];
    }
    out[blockIdx.x * 512 + threadIdx.x] = s;
    if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int OP, int SHAPE, int PRIO>
void once(int it_mfma, int it_valu, double& cyc_mfma, double& cyc_valu, float& ms)
{
    const int grid = 256;
    float* out; unsigned long long* cyc;
    hipMalloc(&out, grid * 512 * sizeof(float)); hipMalloc(&cyc, grid * 8 * sizeof(unsigned long long));
    hipLaunchKernelGGL((k<OP, SHAPE, PRIO>), dim3(grid), dim3(512), 0, 0, out, cyc, it_mfma, it_valu); hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0); hipLaunchKernelGGL((k<OP, SHAPE, PRIO>), dim3(grid), dim3(512), 0, 0, out, cyc, it_mfma, it_valu); hipEventRecord(e1); hipDeviceSynchronize();
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(grid * 8); hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    double am = 0, av = 0;
    for (int g = 0; g < grid; ++g) for (int w = 0; w < 8; ++w) (w < 4 ? am : av) += (double)h[g * 8 + w];
    cyc_mfma = it_mfma ? am / (grid * 4) / ((double)it_mfma * 64) : 0;
    cyc_valu = it_valu ? av / (grid * 4) / ((double)it_valu * 64) : 0;
    hipFree(out); hipFree(cyc);
}

template <int OP, int SHAPE, int PRIO>
void run(const char* name, double valu_cost_guess)
{
    // MFMA loop of it_mfma*64 MFMAs ~ it*2048 cycles; VALU loop of it_valu*64 ops ~ it*64*cost cycles
    const int base = 20000;
    const int vbase = (int)(base * 2048.0 / (64.0 * valu_cost_guess));
    double m_alone, v_alone, m_b, v_b, d; float ms;
    once<OP, SHAPE, PRIO>(base, 0, m_alone, d, ms);
    once<OP, SHAPE, PRIO>(0, vbase, d, v_alone, ms);
    once<OP, SHAPE, PRIO>(base, vbase * 4, m_b, d, ms);        // MFMA waves timed while the VALU waves are still busy
    once<OP, SHAPE, PRIO>(base * 4, vbase, d, v_b, ms);        // VALU waves timed while the MFMA waves are still busy
    printf("%s%s %-20s  per 16x16x4-equivalent MFMA: %5.1f cycles alone, %5.1f beside the op | per op: %5.2f cycles alone, %6.2f beside MFMAs\n",
           SHAPE == 16 ? "16x16x4   " : (SHAPE == 1632 ? "16x16x32f16" : "32x32x2   "), PRIO ? " VALU-wave-prio3" : "                ", name, m_alone, m_b, v_alone, v_b);
}
int main()
{
    run<OP_FMA, 16, 0>("v_fma_f32", 4.0);
    run<OP_PKFMA, 16, 0>("v_pk_fma_f32 (per 2)", 4.0);
    run<OP_EXP, 16, 0>("v_exp_f32 + v_sub", 12.0);
    run<OP_RCP, 16, 0>("v_rcp_f32 + v_fma", 12.0);
    run<OP_MIX, 16, 0>("sigmoid (4 ops)", 24.0);
    run<OP_FMA, 16, 1>("v_fma_f32", 4.0);
    run<OP_EXP, 16, 1>("v_exp_f32 + v_sub", 12.0);
    run<OP_MIX, 16, 1>("sigmoid (4 ops)", 24.0);
    run<OP_FMA, 1632, 0>("v_fma_f32", 4.0);
    run<OP_EXP, 1632, 0>("v_exp_f32 + v_sub", 12.0);
    run<OP_MIX, 1632, 0>("sigmoid (4 ops)", 24.0);
    run<OP_MIX, 1632, 1>("sigmoid (4 ops)", 24.0);
    run<OP_FMA, 32, 0>("v_fma_f32", 4.0);
    run<OP_MIX, 32, 0>("sigmoid (4 ops)", 24.0);
    run<OP_FMA, 32, 1>("v_fma_f32", 4.0);
    run<OP_MIX, 32, 1>("sigmoid (4 ops)", 24.0);
    return 0;
}
