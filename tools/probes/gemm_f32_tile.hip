// Probe: the fp32 tile GEMM loop of hap_gemm.hpp (v_mfma_f32_32x32x2_f32, K chunks of 16, operands through LDS, loads two chunks ahead)
// for several workgroup shapes - would larger tiles lift the fused LSTM step above 0.89 of the fp32 MFMA peak?  Same total work per
// launch as one step launch of the HaplotypeModel (rows 1024 x sites 32768 x K 800).  Reports ms and TFLOP/s against the 157.3 peak.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int LDK = 20;

template <int ROWS, int SITES, int RT, int CT, int MINB>
__global__ __launch_bounds__(64 * (ROWS / (32 * RT)) * (SITES / (32 * CT)), MINB) void k(const float* __restrict__ w, const float* __restrict__ x, float* out, int nk)
{
    constexpr int WR = ROWS / (32 * RT), WC = SITES / (32 * CT), WAVES = WR * WC;
    constexpr int NT = 64 * WAVES;
    __shared__ float As[2][ROWS][LDK];
    __shared__ float Bs[2][SITES][LDK];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
    const int wr = wave / WC, wc = wave % WC;
    const float* wt = w + (size_t)blockIdx.y * nk * (ROWS * 16);
    const float* xt = x + (size_t)blockIdx.x * nk * (SITES * 16);
    constexpr int APT = (ROWS * 4 + NT - 1) / NT, BPT = (SITES * 4 + NT - 1) / NT;      // 16-byte pieces per thread
    f32x4 ga[APT], gb[BPT];
    auto gload = [&](int kc) {
        const f32x4* pa = reinterpret_cast<const f32x4*>(wt + (size_t)kc * (ROWS * 16));
        const f32x4* pb = reinterpret_cast<const f32x4*>(xt + (size_t)kc * (SITES * 16));
#pragma unroll
        for (int i = 0; i < APT; ++i) if (tid + NT * i < ROWS * 4) ga[i] = pa[tid + NT * i];
#pragma unroll
        for (int i = 0; i < BPT; ++i) if (tid + NT * i < SITES * 4) gb[i] = pb[tid + NT * i];
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < APT; ++i) if (tid + NT * i < ROWS * 4) { const int p = tid + NT * i; *reinterpret_cast<f32x4*>(&As[buf][p >> 2][(p & 3) * 4]) = ga[i]; }
#pragma unroll
        for (int i = 0; i < BPT; ++i) if (tid + NT * i < SITES * 4) { const int p = tid + NT * i; *reinterpret_cast<f32x4*>(&Bs[buf][p >> 2][(p & 3) * 4]) = gb[i]; }
    };
    f32x16 acc[RT][CT];
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
        for (int j = 0; j < CT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    gload(0); lstore(0); if (nk > 1) gload(1);
    __syncthreads();
    for (int kc = 0; kc < nk; ++kc) {
        const int cur = kc & 1;
        f32x4 af[RT][2], bf[CT][2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) af[rt][h] = *reinterpret_cast<const f32x4*>(&As[cur][32 * RT * wr + 32 * rt + li][lh * 8 + h * 4]);
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) bf[ct][h] = *reinterpret_cast<const f32x4*>(&Bs[cur][32 * CT * wc + 32 * ct + li][lh * 8 + h * 4]);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (j == 1 && kc + 1 < nk) { __builtin_amdgcn_sched_barrier(0); lstore(cur ^ 1); __builtin_amdgcn_sched_barrier(0); }
            if (j == 4 && kc + 2 < nk) { __builtin_amdgcn_sched_barrier(0); gload(kc + 2); __builtin_amdgcn_sched_barrier(0); }
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int ct = 0; ct < CT; ++ct)
                    acc[rt][ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[rt][j >> 2][j & 3], bf[ct][j >> 2][j & 3], acc[rt][ct], 0, 0, 0);
        }
        __syncthreads();
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
        for (int j = 0; j < CT; ++j) s += acc[i][j][0] + acc[i][j][15];
    out[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * NT + tid] = s;
}

template <int ROWS, int SITES, int RT, int CT, int MINB>
void run(const char* name, const float* w, const float* x, float* out)
{
    constexpr int NT = 64 * (ROWS / (32 * RT)) * (SITES / (32 * CT));
    const int nk = 50, sites = 16384 * 2, rows = 1024;
    const dim3 grid(sites / SITES, rows / ROWS);
    auto kern = k<ROWS, SITES, RT, CT, MINB>;
    hipLaunchKernelGGL(kern, grid, dim3(NT), 0, 0, w, x, out, nk); hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    for (int r = 0; r < 10; ++r) hipLaunchKernelGGL(kern, grid, dim3(NT), 0, 0, w, x, out, nk);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10;
    const double flop = 2.0 * rows * (double)sites * (nk * 16);
    int nb = 0; hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kern, NT, 0);
    hipFuncAttributes fa; hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(kern));
    printf("%-46s %7.3f ms  %7.1f TFLOP/s = %.3f of 157.3   (%d workgroups of %d waves per CU, %d registers, %zu B LDS)\n", name, ms, flop / (ms * 1e-3) / 1e12,
           flop / (ms * 1e-3) / 157.3e12, nb, NT / 64, fa.numRegs, fa.sharedSizeBytes);
}
int main()
{
    float *w, *x, *out;
    const size_t wn = (size_t)8 * 50 * 256 * 16, xn = (size_t)256 * 50 * 128 * 16;
    hipMalloc(&w, wn * 4); hipMalloc(&x, xn * 4); hipMalloc(&out, (size_t)4096 * 1024 * 4);
    float* h = (float*)malloc(xn * 4);
    for (size_t i = 0; i < xn; ++i) h[i] = (float)((i * 2654435761u) % 2001) / 1000.f - 1.f;
    hipMemcpy(x, h, xn * 4, hipMemcpyHostToDevice);
    hipMemcpy(w, h, wn * 4, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 2; ++rep) {
        run<128, 128, 2, 2, 4>("128 x 128, 4 waves of 64 x 64 (the kernel)", w, x, out);
        run<128, 128, 2, 2, 3>("128 x 128, 4 waves of 64 x 64, 3 per CU", w, x, out);
        run<128, 128, 2, 2, 2>("128 x 128, 4 waves of 64 x 64, bounds 2", w, x, out);
        run<128, 256, 2, 2, 1>("128 x 256, 8 waves of 64 x 64, bounds 1", w, x, out);
        run<256, 128, 2, 2, 2>("256 x 128, 8 waves of 64 x 64", w, x, out);
        run<128, 256, 2, 2, 2>("128 x 256, 8 waves of 64 x 64", w, x, out);
        run<256, 256, 4, 2, 1>("256 x 256, 8 waves of 128 x 64", w, x, out);
        run<256, 256, 2, 4, 1>("256 x 256, 8 waves of 64 x 128", w, x, out);
        run<256, 128, 4, 2, 2>("256 x 128, 4 waves of 128 x 64", w, x, out);
        run<128, 256, 2, 4, 2>("128 x 256, 4 waves of 64 x 128", w, x, out);
        printf("\n");
    }
    return 0;
}
