// Probe: issue rate of v_mfma_f32_16x16x4_f32 (the PileupModel recurrence kernels' instruction) against v_mfma_f32_32x32x2_f32 (the tile
// GEMM's).  Operands in registers with random-ish contents, N accumulators interleaved (dependent MFMAs N issue slots apart),
// 1..3 workgroups of four waves per CU.  Reports TFLOP/s from HIP events; 157.3 = the fp32 matrix peak at 2.4 GHz.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(256) void k16(float* out, int iters)
{
    f32x4 acc[NACC];
    float a[8], b[8];
    for (int i = 0; i < 8; ++i) { a[i] = 0.001f * ((threadIdx.x * 7 + i * 13) % 31) - 0.015f; b[i] = 0.002f * ((threadIdx.x * 5 + i * 11) % 29) - 0.03f; }
    for (int u = 0; u < NACC; ++u) acc[u] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int u = 0; u < NACC; ++u) acc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[(j + u) & 7], b[j], acc[u], 0, 0, 0);
    }
    float r = 0.f;
    for (int u = 0; u < NACC; ++u) r += acc[u][0] + acc[u][1] + acc[u][2] + acc[u][3];
    out[blockIdx.x * 256 + threadIdx.x] = r;
}

__global__ __launch_bounds__(256) void k32(float* out, int iters)
{
    f32x16 acc[4];
    float a[8], b[8];
    for (int i = 0; i < 8; ++i) { a[i] = 0.001f * ((threadIdx.x * 7 + i * 13) % 31) - 0.015f; b[i] = 0.002f * ((threadIdx.x * 5 + i * 11) % 29) - 0.03f; }
    for (int u = 0; u < 4; ++u) for (int i = 0; i < 16; ++i) acc[u][i] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int u = 0; u < 4; ++u) acc[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(j + u) & 7], b[j], acc[u], 0, 0, 0);
    }
    float r = 0.f;
    for (int u = 0; u < 4; ++u) for (int i = 0; i < 16; ++i) r += acc[u][i];
    out[blockIdx.x * 256 + threadIdx.x] = r;
}

template <typename F>
void run(const char* name, F launch, int grid, double flop_per_mfma, int mfma_per_iter)
{
    const int iters = 20000;
    launch(grid, 100); (void)hipDeviceSynchronize();
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0); launch(grid, iters); (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double flop = (double)grid * 4 * iters * mfma_per_iter * flop_per_mfma;
    printf("%-44s grid %4d: %8.2f ms  %6.1f TFLOP/s  (%.3f of 157.3)\n", name, grid, ms, flop / (ms * 1e-3) / 1e12, flop / (ms * 1e-3) / 1e12 / 157.3);
}

int main()
{
    float* out; (void)hipMalloc(&out, 4096 * 256 * 4);
    for (int grid : {256, 512, 768}) {
        run("16x16x4 f32, 4 accumulators interleaved", [&](int g, int it) { hipLaunchKernelGGL((k16<4>), dim3(g), dim3(256), 0, 0, out, it); }, grid, 2048.0, 32);
        run("16x16x4 f32, 2 accumulators interleaved", [&](int g, int it) { hipLaunchKernelGGL((k16<2>), dim3(g), dim3(256), 0, 0, out, it); }, grid, 2048.0, 16);
        run("16x16x4 f32, 8 accumulators interleaved", [&](int g, int it) { hipLaunchKernelGGL((k16<8>), dim3(g), dim3(256), 0, 0, out, it); }, grid, 2048.0, 64);
        run("32x32x2 f32, 4 accumulators interleaved", [&](int g, int it) { hipLaunchKernelGGL(k32, dim3(g), dim3(256), 0, 0, out, it); }, grid, 4096.0, 32);
    }
    return 0;
}
