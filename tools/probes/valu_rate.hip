// Probe: issue/throughput cost of v_exp_f32 / v_rcp_f32 / v_fma_f32 and their overlap with v_mfma_f32_16x16x32_f16
// on gfx950, at 1, 2 and 4 waves per SIMD.  hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ void k(float* out, int iters, long long* cyc)
{
    float v[8]; f32x4 acc[4]; h8 a, b;
    for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 0.001f + i;
    for (int i = 0; i < 4; ++i) acc[i] = f32x4{0, 0, 0, 0};
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.01f * i); b[i] = (_Float16)(0.02f * i); }
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {          // 8 independent exp
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = __builtin_amdgcn_exp2f(v[i]);
        } else if (MODE == 1) {   // 8 independent rcp
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = __builtin_amdgcn_rcpf(v[i]);
        } else if (MODE == 2) {   // 8 independent fma
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = __builtin_fmaf(v[i], 1.0001f, 0.5f);
        } else if (MODE == 3) {   // 4 mfma
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[i], 0, 0, 0);
        } else if (MODE == 4) {   // 4 mfma + 8 exp interleaved
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[i], 0, 0, 0);
                v[2 * i] = __builtin_amdgcn_exp2f(v[2 * i]);
                v[2 * i + 1] = __builtin_amdgcn_exp2f(v[2 * i + 1]);
            }
        } else if (MODE == 5) {   // 4 mfma + 4 exp + 4 fma
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[i], 0, 0, 0);
                v[2 * i] = __builtin_amdgcn_exp2f(v[2 * i]);
                v[2 * i + 1] = __builtin_fmaf(v[2 * i + 1], 1.0001f, 0.5f);
            }
        } else if (MODE == 6) {   // 4 mfma + 16 fma
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[i], 0, 0, 0);
                v[2 * i] = __builtin_fmaf(v[2 * i], 1.0001f, 0.5f);
                v[2 * i + 1] = __builtin_fmaf(v[2 * i + 1], 1.0001f, 0.5f);
                v[(2 * i + 2) & 7] = __builtin_fmaf(v[(2 * i + 2) & 7], 1.0001f, 0.5f);
                v[(2 * i + 3) & 7] = __builtin_fmaf(v[(2 * i + 3) & 7], 1.0001f, 0.5f);
            }
        } else if (MODE == 7) {   // 4 exp + 4 fma, no mfma
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                v[2 * i] = __builtin_amdgcn_exp2f(v[2 * i]);
                v[2 * i + 1] = __builtin_fmaf(v[2 * i + 1], 1.0001f, 0.5f);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int i = 0; i < 8; ++i) s += v[i];
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int MODE>
void run(const char* name, int per_iter)
{
    float* out; long long* cyc;
    hipMalloc(&out, 256 * 2048 * 4 * sizeof(float)); hipMalloc(&cyc, 8);
    const int iters = 20000;
    for (int wps = 1; wps <= 4; wps *= 2) {           // waves per SIMD: block = 256 * wps threads, one block per CU
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(256 * wps), 0, 0, out, 100, cyc);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(256 * wps), 0, 0, out, iters, cyc);
        hipEventRecord(e1); hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
        // readcyclecounter on gfx9 = s_memtime (100 MHz constant?) -> also report from wall time at an assumed 2.1 GHz
        printf("%-28s waves/SIMD=%d  %.3f ms  counter/iter=%.1f  wall-cycles(2.1GHz)/iter/wave-slot=%.1f  per-instr(per SIMD)=%.2f\n",
               name, wps, ms, (double)c / iters, ms * 1e-3 * 2.1e9 / iters, ms * 1e-3 * 2.1e9 / iters / (per_iter * wps));
    }
    hipFree(out); hipFree(cyc);
}

int main()
{
    run<0>("8 exp", 8); run<1>("8 rcp", 8); run<2>("8 fma", 8); run<3>("4 mfma16x16x32", 4);
    run<4>("4 mfma + 8 exp", 12); run<5>("4 mfma + 4 exp + 4 fma", 12); run<6>("4 mfma + 16 fma", 20); run<7>("4 exp + 4 fma", 8);
    return 0;
}
