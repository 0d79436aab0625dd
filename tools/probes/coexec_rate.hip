// Probe: do MFMA and VALU co-execute when they come from DIFFERENT waves of a SIMD?  Workgroups of 8 waves: waves 0-3 (one per
// SIMD) run only v_mfma_f32_16x16x32_f16, waves 4-7 (their SIMD partners) run only LSTM cells.  gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float rcp1p_exp2(float x) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x)); }

template <int MODE>   // 0: split roles, 1: MFMA waves only (others exit), 2: cell waves only, 3: every wave does both (mixed)
__global__ void k(float* out, int iters)
{
    const int wave = threadIdx.x >> 6;
    const bool mfma_role = MODE == 3 ? true : wave < 4, cell_role = MODE == 3 ? true : wave >= 4;
    if (MODE == 1 && !mfma_role) return;
    if (MODE == 2 && !cell_role) return;
    f32x4 acc[4]; h8 a[4], b; float c[4] = {0.1f, 0.2f, 0.3f, 0.4f}; float s = 0;
    for (int i = 0; i < 4; ++i) { acc[i] = f32x4{0.1f, -0.2f, 0.3f, 0.05f}; for (int j = 0; j < 8; ++j) a[i][j] = (_Float16)(0.01f * (i + j)); }
    for (int j = 0; j < 8; ++j) b[j] = (_Float16)(0.02f * j);
    f32x4 g[4]; for (int i = 0; i < 4; ++i) g[i] = f32x4{0.3f, -0.1f, 0.2f, 0.4f};
    for (int it = 0; it < iters; ++it) {
        if (mfma_role) {
#pragma unroll
            for (int r = 0; r < 8; ++r)
#pragma unroll
                for (int u = 0; u < 4; ++u) acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[u], b, acc[u], 0, 0, 0);
        }
        if (cell_role) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float ig = rcp1p_exp2(g[u][0]), fg = rcp1p_exp2(g[u][1]);
                const float gk = __builtin_fmaf(-5.77f, rcp1p_exp2(g[u][2]), 2.885f);
                const float og = rcp1p_exp2(g[u][3]);
                const float cn = __builtin_fmaf(fg, c[u], ig * gk);
                c[u] = cn;
                const float h = og * __builtin_fmaf(-2.0f, rcp1p_exp2(cn), 1.0f);
                g[u][0] = h * 0.5f; g[u][1] = -h; g[u][2] = h + 0.1f; g[u][3] = 0.3f - h;
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][3] + c[i] + g[i][1];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE> void run(const char* name)
{
    float* out; hipMalloc(&out, 512 * 512 * sizeof(float));
    const int iters = 4000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, out, 10); hipDeviceSynchronize();
    hipEventRecord(e0); hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, out, iters); hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-60s %.3f ms   cycles(2.0 GHz) per iteration = %.0f\n", name, ms, ms * 1e-3 * 2.0e9 / iters);
    hipFree(out);
}
int main()
{
    run<1>("waves 0-3: 32 MFMA per iteration, alone");
    run<2>("waves 4-7: 4 cells per iteration, alone");
    run<0>("waves 0-3 MFMA + waves 4-7 cells, one of each per SIMD");
    run<3>("all 8 waves: 32 MFMA + 4 cells each (mixed)");
    return 0;
}
