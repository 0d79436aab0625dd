// Probe: what does the bf16 matrix pipe of this chip SUSTAIN - the rate the bf16x3 kernels should be priced against?  Bare loops of
// v_mfma_f32_16x16x32_bf16 / v_mfma_f32_32x32x16_bf16 on register operands (no LDS, no memory), every CU, 1 / 2 / 4 waves per SIMD, for
// ~40 ms each (long enough for the power management to settle), with operands of
//   zero      all operand bits 0 (the pipe's rate with nothing toggling)
//   random    random bf16 mantissas, exponents near 1 (a GEMM on real data)
//   split     the three planes of a bf16x3 split of random fp32 values: p0 random, p1 ~ 2^-8 p0, p2 ~ 2^-16 p0, used in the six-product
//             pattern of the bf16x3 kernels
// Reports TFLOP/s and the shader clock during the loop (s_memtime / s_memrealtime at 100 MHz).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <cstring>
typedef __bf16 b8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int SHAPE>   // 0: 16x16x32, 1: 32x32x16
__global__ __launch_bounds__(256) void k(const b8* __restrict__ ops, float* out, int iters, unsigned long long* clk)
{
    const int tid = blockIdx.x * 256 + threadIdx.x;
    b8 a[3], b[3];
#pragma unroll
    for (int p = 0; p < 3; ++p) { a[p] = ops[(size_t)(2 * p) * 64 + (threadIdx.x & 63)]; b[p] = ops[(size_t)(2 * p + 1) * 64 + (threadIdx.x & 63)]; }
    const unsigned long long t0 = __builtin_readcyclecounter(), r0 = wall_clock64();
    float s = 0.f;
    if (SHAPE == 0) {
        f32x4 acc[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        constexpr int WP[6] = {0, 1, 2, 0, 1, 0}, XP[6] = {2, 1, 0, 1, 0, 0};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int g = 0; g < 6; ++g)
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[WP[g]], b[XP[g]], acc[i], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][3];
    } else {
        f32x16 acc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        constexpr int WP[6] = {0, 1, 2, 0, 1, 0}, XP[6] = {2, 1, 0, 1, 0, 0};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int g = 0; g < 6; ++g)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[WP[g]], b[XP[g]], acc[i], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][15];
    }
    const unsigned long long t1 = __builtin_readcyclecounter(), r1 = wall_clock64();
    out[tid] = s;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

static uint16_t bf(float v) { uint32_t u; memcpy(&u, &v, 4); return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16); }
static float f32(uint16_t h) { uint32_t u = (uint32_t)h << 16; float v; memcpy(&v, &u, 4); return v; }

int main()
{
    const int n_ops = 6 * 64 * 8;
    uint16_t* h = (uint16_t*)malloc(n_ops * 2 * 3);
    // three operand sets: zero, random (all planes random), split
    for (int i = 0; i < n_ops; ++i) h[i] = 0;
    for (int i = 0; i < n_ops; ++i) h[n_ops + i] = bf(((float)rand() / RAND_MAX - 0.5f) * 2.f);
    for (int q = 0; q < 2; ++q)            // a-operand / b-operand
        for (int l = 0; l < 64 * 8; ++l) {
            const float v = ((float)rand() / RAND_MAX - 0.5f) * 2.f;
            const uint16_t p0 = bf(v); const float r1 = v - f32(p0); const uint16_t p1 = bf(r1); const uint16_t p2 = bf(r1 - f32(p1));
            h[2 * n_ops + (0 * 2 + q) * 512 + l] = p0; h[2 * n_ops + (1 * 2 + q) * 512 + l] = p1; h[2 * n_ops + (2 * 2 + q) * 512 + l] = p2;
        }
    uint16_t* d; hipMalloc(&d, n_ops * 2 * 3); hipMemcpy(d, h, n_ops * 2 * 3, hipMemcpyHostToDevice);
    float* out; hipMalloc(&out, 4096 * 256 * 4);
    unsigned long long* clk; hipMalloc(&clk, 4096 * 16);
    unsigned long long* hclk = (unsigned long long*)malloc(4096 * 16);
    const char* names[3] = {"zero", "random", "split"};
    for (int shape = 0; shape < 2; ++shape)
        for (int set = 0; set < 3; ++set)
            for (int wps : {1, 2, 4}) {
                const int grid = 256 * wps;            // 256-thread workgroups: wps per CU = wps waves per SIMD
                const int per_iter = shape == 0 ? 48 : 24;
                const double flop_per_mfma = 32768.0 / (shape == 0 ? 2 : 1);   // 16x16x32: 16384, 32x32x16: 32768
                int iters = 20000 / wps * (shape == 0 ? 1 : 1);
                const b8* ops = reinterpret_cast<const b8*>(d + (size_t)set * n_ops);
                auto launch = [&](int it) { if (shape == 0) hipLaunchKernelGGL(k<0>, dim3(grid), dim3(256), 0, 0, ops, out, it, clk);
                                            else hipLaunchKernelGGL(k<1>, dim3(grid), dim3(256), 0, 0, ops, out, it, clk); };
                launch(iters); hipDeviceSynchronize();
                hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
                hipEventRecord(e0); launch(iters * 4); hipEventRecord(e1); hipDeviceSynchronize();
                float ms; hipEventElapsedTime(&ms, e0, e1);
                hipMemcpy(hclk, clk, grid * 16, hipMemcpyDeviceToHost);
                double cyc = 0, real = 0; for (int i = 0; i < grid; ++i) { cyc += hclk[2 * i]; real += hclk[2 * i + 1]; }
                const double mhz = cyc / real * 100.0;
                const double flop = (double)grid * 4 * (iters * 4.0) * per_iter * flop_per_mfma;
                printf("%-10s %-7s %d waves/SIMD: %7.2f ms  %7.1f TFLOP/s = %.3f of 2500  clock %5.0f MHz  pipe busy %.3f\n", shape == 0 ? "16x16x32" : "32x32x16", names[set], wps, ms,
                       flop / (ms * 1e-3) / 1e12, flop / (ms * 1e-3) / 2.5e15, mhz, flop / (ms * 1e-3) / (1024.0 * 1024.0 * mhz * 1e6 * 1.0));
            }
    return 0;
}
