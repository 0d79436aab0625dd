// Probe: the inner loop of the bf16x3 layer-1 kernel (k_pileup_l1_b3) in miniature - weights of NT gate tiles (6 K blocks x 3 planes)
// stationary in registers, the B fragments (three planes per K block and 16-site group) read from LDS by every wave, six
// v_mfma_f32_16x16x32_bf16 per (tile, K block) - for two splits of the 16 gate tiles of a direction over the waves of a workgroup:
//   8 waves x 2 tiles (the kernel: two waves per SIMD, every B fragment read by eight waves, 3 reads per 12 MFMAs)
//   4 waves x 4 tiles (one wave per SIMD, 288 weight registers, every B fragment read by four waves, 3 reads per 24 MFMAs)
// No cells, no exchange, no global memory: what the matrix + LDS side alone sustains, and at which clock.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef __bf16 b8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NT, int WAVES, int G>      // tiles per wave, waves per workgroup, 16-site groups per workgroup
__global__ __launch_bounds__(64 * WAVES, 1) void k(const b8* __restrict__ w, float* out, int steps, unsigned long long* clk)
{
    extern __shared__ b8 lds[];          // [G][6 kb][3 planes][64 lanes] b8 = 16 B each: 18 KB per group
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    b8 W[NT][6][3];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int kb = 0; kb < 6; ++kb)
#pragma unroll
            for (int p = 0; p < 3; ++p) W[t][kb][p] = w[(((wave * NT + t) * 6 + kb) * 3 + p) * 64 + lane];
    for (int i = tid; i < G * 6 * 3 * 64; i += 64 * WAVES) lds[i] = w[(i * 7 + 13) % (16 * 6 * 3 * 64)];
    __syncthreads();
    f32x4 acc[G][NT];
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[g][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    const unsigned long long t0 = __builtin_readcyclecounter(), r0 = wall_clock64();
    constexpr int WP[6] = {0, 1, 2, 0, 1, 0}, XP[6] = {2, 1, 0, 1, 0, 0};
    for (int s = 0; s < steps; ++s) {
#pragma unroll
        for (int g = 0; g < G; ++g)
#pragma unroll
            for (int kb = 0; kb < 6; ++kb) {
                b8 bf[3];
#pragma unroll
                for (int p = 0; p < 3; ++p) bf[p] = lds[((g * 6 + kb) * 3 + p) * 64 + lane];
#pragma unroll
                for (int pr = 0; pr < 6; ++pr)
#pragma unroll
                    for (int t = 0; t < NT; ++t) acc[g][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(W[t][kb][WP[pr]], bf[XP[pr]], acc[g][t], 0, 0, 0);
            }
        __builtin_amdgcn_s_barrier();                  // (the kernel's per-step barrier)
    }
    const unsigned long long t1 = __builtin_readcyclecounter(), r1 = wall_clock64();
    float sum = 0;
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
        for (int t = 0; t < NT; ++t) sum += acc[g][t][0] + acc[g][t][3];
    out[blockIdx.x * 64 * WAVES + tid] = sum;
    if (tid == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int NT, int WAVES, int G>
void run(const char* name, const b8* w, float* out, unsigned long long* clk, unsigned long long* hclk)
{
    const int grid = 256, steps = 3000;
    const size_t lds = (size_t)G * 6 * 3 * 64 * 16;
    auto kern = k<NT, WAVES, G>;
    hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * WAVES), lds, 0, w, out, 200, clk); hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0); hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * WAVES), lds, 0, w, out, steps, clk); hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(hclk, clk, grid * 16, hipMemcpyDeviceToHost);
    double cyc = 0, real = 0; for (int i = 0; i < grid; ++i) { cyc += hclk[2 * i]; real += hclk[2 * i + 1]; }
    const double flop = (double)grid * WAVES * steps * G * 6 * 6 * NT * 16384.0;
    hipFuncAttributes fa; hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(kern));
    printf("%-34s %7.2f ms  %7.1f TFLOP/s = %.3f of 2500  clock %4.0f MHz  %d registers\n", name, ms, flop / (ms * 1e-3) / 1e12, flop / (ms * 1e-3) / 2.5e15, cyc / real * 100.0, fa.numRegs);
}
int main()
{
    const size_t n = (size_t)16 * 6 * 3 * 64;
    uint16_t* h = (uint16_t*)malloc(n * 16);
    for (size_t i = 0; i < n * 8; ++i) h[i] = (uint16_t)(0x3c00 + ((i * 40503u) & 0x3ff) + ((i % 3) ? 0 : 0x8000));
    b8* w; hipMalloc(&w, n * 16); hipMemcpy(w, h, n * 16, hipMemcpyHostToDevice);
    float* out; hipMalloc(&out, 256 * 512 * 4);
    unsigned long long* clk; hipMalloc(&clk, 256 * 16); unsigned long long* hclk = (unsigned long long*)malloc(256 * 16);
    for (int rep = 0; rep < 2; ++rep) {
        run<2, 8, 4>("8 waves x 2 tiles, 64 sites", w, out, clk, hclk);
        run<4, 4, 4>("4 waves x 4 tiles, 64 sites", w, out, clk, hclk);
        run<4, 4, 2>("4 waves x 4 tiles, 32 sites", w, out, clk, hclk);
        run<2, 8, 2>("8 waves x 2 tiles, 32 sites", w, out, clk, hclk);
    }
    return 0;
}
