// Probe: sustained fp32 MFMA rate of the two fp32 shapes on gfx950, operands in registers, no memory traffic.
//   v_mfma_f32_16x16x4_f32  (the PileupModel recurrence kernels)   2048 flop, 8 passes
//   v_mfma_f32_32x32x2_f32  (the tile-image GEMM of the HaplotypeModel) 4096 flop, 16 passes
// Both have the same nominal rate (256 flop / clock / CU-SIMD quarter); what differs is register-file traffic per flop and hence
// power, and the chip is power-limited at full MFMA load.  Reports TFLOP/s from HIP events over a ~0.3 s launch.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float noise(unsigned x)
{
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return __uint_as_float(0x3c000000u | (x & 0x00ffffffu)) * ((x >> 31) ? -1.f : 1.f);     // full-mantissa values around +-0.01 .. 0.03
}

template <int SHAPE, int NACC, bool NOISY>
__global__ __launch_bounds__(256) void k(float* out, int iters)
{
    const int lane = threadIdx.x & 63;
    float a[8], b[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        if (NOISY) { a[i] = noise(threadIdx.x * 16 + i); b[i] = noise(threadIdx.x * 16 + 8 + i + blockIdx.x * 4096); }
        else { a[i] = 0.001f * (float)((i * 7 + lane) % 13) - 0.006f; b[i] = 0.002f * (float)((i * 5 + lane) % 11) - 0.01f; }
    }
    float s = 0.f;
    if (SHAPE == 16) {
        f32x4 acc[NACC];
#pragma unroll
        for (int u = 0; u < NACC; ++u) acc[u] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int r = 0; r < 64 / NACC; ++r)
#pragma unroll
                for (int u = 0; u < NACC; ++u) acc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[(r + u) & 7], b[r & 7], acc[u], 0, 0, 0);
#pragma unroll
        for (int u = 0; u < NACC; ++u) s += acc[u][0] + acc[u][3];
    } else {
        f32x16 acc[4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[u][r] = 0.f;
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int r = 0; r < 8; ++r)
#pragma unroll
                for (int u = 0; u < 4; ++u) acc[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(r + u) & 7], b[r], acc[u], 0, 0, 0);
#pragma unroll
        for (int u = 0; u < 4; ++u) s += acc[u][0] + acc[u][15];
    }
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int SHAPE, int NACC, bool NOISY>
void run(int wgs_per_cu, int iters)
{
    const int grid = 256 * wgs_per_cu;
    float* out; hipMalloc(&out, grid * 256 * sizeof(float));
    hipLaunchKernelGGL((k<SHAPE, NACC, NOISY>), dim3(grid), dim3(256), 0, 0, out, iters); hipDeviceSynchronize();   // warm, clocks settle
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0); hipLaunchKernelGGL((k<SHAPE, NACC, NOISY>), dim3(grid), dim3(256), 0, 0, out, iters); hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flop = (double)grid * 4 * iters * (SHAPE == 16 ? 64 * 2048.0 : 32 * 4096.0);
    printf("%s  %d accumulators  %s operands  %d waves/SIMD  %8.2f ms  %7.1f TFLOP/s\n", SHAPE == 16 ? "16x16x4_f32" : "32x32x2_f32",
           SHAPE == 16 ? NACC : 4, NOISY ? "full-mantissa" : "few-bit      ", wgs_per_cu, ms, flop / (ms * 1e-3) / 1e12);
    hipFree(out);
}
int main()
{
    for (int w = 1; w <= 3; ++w) {
        run<16, 8, false>(w, 200000 / w);
        run<16, 4, false>(w, 200000 / w);
        run<16, 2, false>(w, 200000 / w);
        run<32, 4, false>(w, 200000 / w);
        run<16, 8, true>(w, 200000 / w);
        run<16, 4, true>(w, 200000 / w);
        run<32, 4, true>(w, 200000 / w);
    }
    return 0;
}
