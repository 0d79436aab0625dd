// Probe: SIMD cycles per LSTM "group" = 32 x v_mfma_f32_16x16x32_f16 + 4 cells (5 exp2, 5 rcp, 5 add, ~8 fma/mul, fp16 hi/lo split)
// at 1/2/4/8 waves per SIMD, for a few instruction orders.  gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float rcp1p_exp2(float x) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x)); }

template <int MODE>
__global__ void k(float* out, int iters)
{
    f32x4 acc[2][4]; h8 a[4], b; float c[4] = {0.1f, 0.2f, 0.3f, 0.4f};
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 8; ++j) a[i][j] = (_Float16)(0.01f * (i + j + threadIdx.x % 7));
    for (int j = 0; j < 8; ++j) b[j] = (_Float16)(0.02f * j);
    for (int s = 0; s < 2; ++s) for (int i = 0; i < 4; ++i) acc[s][i] = f32x4{0.1f, -0.2f, 0.3f, 0.05f};
    h4 nh = {0, 0, 0, 0}, nl = {0, 0, 0, 0};
    typedef float f32x16 __attribute__((ext_vector_type(16)));
    f32x16 big[2];
    for (int u = 0; u < 2; ++u) for (int i = 0; i < 16; ++i) big[u][i] = 0.01f * i;
    for (int it = 0; it < iters; ++it) {
        const int p = it & 1;
        // gemm of the next group: 32 MFMAs on 4 accumulators
        if (MODE >= 5 && MODE <= 7) {
            // the same flops as 32 x 16x16x32: 16 x v_mfma_f32_32x32x16_f16 on 2 accumulators of 16 registers
#pragma unroll
            for (int r = 0; r < 8; ++r)
#pragma unroll
                for (int u = 0; u < 2; ++u) big[u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[u], b, big[u], 0, 0, 0);
#pragma unroll
            for (int u = 0; u < 4; ++u) acc[p ^ 1][u] = f32x4{big[u & 1][u], big[u & 1][u + 4], big[u & 1][u + 8], big[u & 1][u + 12]};
        } else if (MODE != 2 && MODE != 8) {
#pragma unroll
            for (int r = 0; r < 8; ++r)
#pragma unroll
                for (int u = 0; u < 4; ++u) acc[p ^ 1][u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[u], b, r ? acc[p ^ 1][u] : f32x4{0, 0, 0, 0}, 0, 0, 0);
        }
        if (MODE == 8 || MODE == 9) {
            // one reciprocal for the four gates: P = (1+ei)(1+ef)(1+eg)(1+eo), r = 1/P
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float ai = 1.0f + __builtin_amdgcn_exp2f(fminf(acc[p][u][0], 30.0f));
                const float af = 1.0f + __builtin_amdgcn_exp2f(fminf(acc[p][u][1], 30.0f));
                const float ag = 1.0f + __builtin_amdgcn_exp2f(fminf(acc[p][u][2], 30.0f));
                const float ao = 1.0f + __builtin_amdgcn_exp2f(fminf(acc[p][u][3], 30.0f));
                const float pif = ai * af, pgo = ag * ao;
                const float r = __builtin_amdgcn_rcpf(pif * pgo);
                const float rif = r * pgo, rgo = r * pif;          // 1/(ai af), 1/(ag ao)
                const float ig = rif * af, fg = rif * ai, rg = rgo * ao, og = rgo * ag;
                const float gk = __builtin_fmaf(-5.77f, rg, 2.885f);
                const float cn = __builtin_fmaf(fg, c[u], ig * gk);
                c[u] = cn;
                const float h = og * __builtin_fmaf(-2.0f, rcp1p_exp2(cn), 1.0f);
                const _Float16 hi = (_Float16)h; const _Float16 lo = (_Float16)(h - (float)hi);
                nh[u] = hi; nl[u] = lo;
            }
            b = __builtin_shufflevector(nh, nl, 0, 1, 2, 3, 4, 5, 6, 7);
        } else
        if (MODE != 1 && MODE != 5) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float ig = rcp1p_exp2(acc[p][u][0]);
                const float fg = rcp1p_exp2(acc[p][u][1]);
                const float gk = __builtin_fmaf(-5.77f, rcp1p_exp2(acc[p][u][2]), 2.885f);
                const float og = rcp1p_exp2(acc[p][u][3]);
                const float cn = __builtin_fmaf(fg, c[u], ig * gk);
                c[u] = cn;
                const float h = og * __builtin_fmaf(-2.0f, rcp1p_exp2(cn), 1.0f);
                const _Float16 hi = (_Float16)h; const _Float16 lo = (_Float16)(h - (float)hi);
                nh[u] = hi; nl[u] = lo;
            }
            // feed h back so that nothing is dead
            b = __builtin_shufflevector(nh, nl, 0, 1, 2, 3, 4, 5, 6, 7);
        }
        if (MODE == 0) {
#pragma unroll
            for (int i = 0; i < 32; ++i) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, 3, 0); }
        }
        if (MODE == 7) {
#pragma unroll
            for (int i = 0; i < 16; ++i) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, 6, 0); }
        }
        if (MODE == 3) {    // trans grouped: MFMA, 1 trans, 2 valu
#pragma unroll
            for (int i = 0; i < 32; ++i) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x400, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, 2, 0); }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    float s = c[0] + c[1] + c[2] + c[3] + (float)nh[0] + (float)nl[1];
    for (int i = 0; i < 4; ++i) s += acc[0][i][0] + acc[1][i][1];
    s += big[0][3] + big[1][5];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
void run(const char* name)
{
    float* out; hipMalloc(&out, 256 * 2048 * 4 * sizeof(float));
    const int iters = 4000;
    for (int wps = 1; wps <= 8; wps *= 2) {
        const int blocks_per_cu = wps > 4 ? 2 : 1, threads = 256 * (wps > 4 ? 4 : wps);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k<MODE>, dim3(256 * blocks_per_cu), dim3(threads), 0, 0, out, 10);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(256 * blocks_per_cu), dim3(threads), 0, 0, out, iters);
        hipEventRecord(e1); hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-40s waves/SIMD=%d  %.3f ms  SIMD-cycles(2.0GHz) per group per wave = %.0f\n", name, wps, ms, ms * 1e-3 * 2.0e9 / iters / wps);
    }
    hipFree(out);
}

int main()
{

    run<4>("32 MFMA + 4 cells, compiler order");
    run<1>("32 MFMA only");
    run<2>("4 cells only");
    run<8>("4 cells only, shared rcp");
    run<9>("32 MFMA + 4 cells shared rcp");
    return 0;
    run<5>("16 MFMA 32x32x16 only");
    run<6>("16 MFMA 32x32x16 + 4 cells, compiler order");
    run<7>("16 MFMA 32x32x16 + 4 cells, M vvvvvv");
    return 0;
}
