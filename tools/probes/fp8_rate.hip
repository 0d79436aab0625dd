// Probe: issue rate of fp8 vs fp16 MFMAs of the same shape on gfx950 (cycles per instruction per SIMD at 1/2/4 waves per SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
template <int MODE>
__global__ void k(float* out, int iters)
{
    f32x4 acc[4]; for (int i = 0; i < 4; ++i) acc[i] = f32x4{0, 0, 0, 0};
    h8 a, b; for (int j = 0; j < 8; ++j) { a[j] = (_Float16)(0.01f * j); b[j] = (_Float16)(0.02f * j); }
    long a8 = 0x3838383838383838L + threadIdx.x, b8 = 0x3030303030303030L;
    i32x8 a128 = {0x38383838, 0x38383838, 0x38383838, 0x38383838, 0x38383838, 0x38383838, 0x38383838, 0x38383838}, b128 = a128;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (MODE == 0) acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[u], 0, 0, 0);
                else if (MODE == 1) acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(a8, b8, acc[u], 0, 0, 0);
                else acc[u] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a128, b128, acc[u], 0, 0, 0, 0, 0, 0);
            }
        __builtin_amdgcn_sched_barrier(0);
    }
    float s = 0; for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE> void run(const char* name, int kdepth)
{
    float* out; hipMalloc(&out, 256 * 1024 * sizeof(float));
    const int iters = 4000;
    for (int wps = 1; wps <= 4; wps *= 2) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(256 * wps), 0, 0, out, 10); hipDeviceSynchronize();
        hipEventRecord(e0); hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(256 * wps), 0, 0, out, iters); hipEventRecord(e1); hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double cyc = ms * 1e-3 * 2.0e9 / iters / 32 / wps;
        printf("%-40s waves/SIMD=%d  %.1f cycles(2.0 GHz) per MFMA per SIMD = %.1f cycles per 32 of K\n", name, wps, cyc, cyc * 32 / kdepth);
    }
    hipFree(out);
}
int main()
{
    run<0>("v_mfma_f32_16x16x32_f16", 32);
    run<1>("v_mfma_f32_16x16x32_fp8_fp8", 32);
    run<2>("v_mfma_scale_f32_16x16x128_f8f6f4 (fp8)", 128);
    return 0;
}
