// Probe: what costs the tile-image GEMM loop (hap_gemm.hpp) its last 16 % of the fp32 matrix pipe?  One chunk = 32 x
// v_mfma_f32_32x32x2_f32 on four accumulators per wave, fed as in the kernel:
//   REG   operands stay in registers (the pipe's own rate)
//   LDS   + the chunk's 8 ds_read_b128 fragment reads (16 KB LDS image per workgroup, rows 80 B apart)
//   BAR   + 4 ds_write_b128 and one workgroup barrier per chunk (no global memory)
//   GLD   + 4 global_load_dwordx4 per chunk (8 KB weight image + 8 KB input image per workgroup from a 64 MB L2/HBM-resident array)
//   KRN   the same with the kernel's reuse: the weight chunk shared by the workgroups of a row tile, the input chunk by those of a site tile
// Grid 512 = two workgroups per CU on average, 1024 = four (LDS would let four be resident: the placement is the dispatcher's).  Reports TFLOP/s from HIP events.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int LDK = 20;

// VAR (KRN modes only): 1 = s_setprio(workgroup slot on the CU & 3) for the whole kernel, 2 = one-time start skew of a quarter
// chunk-round per workgroup slot, 3 = both.  Hypothesis: the four workgroups of a CU share every SIMD's matrix pipe, run the same
// program and reach their per-chunk barrier / LDS phase together, so the pipe idles while all of them synchronise.
template <int MODE, int PAD, int VAR = 0>
__global__ __launch_bounds__(256, 2) void k(const float* __restrict__ g, float* out, int nk, size_t gmask)
{
    if (VAR) {
        const unsigned tg = __builtin_amdgcn_s_getreg((3 << 11) | (16 << 6) | 4) & 3u;     // HW_REG_HW_ID.tg_id: the workgroup's slot on its CU
        if (VAR & 1) { if (tg == 1) __builtin_amdgcn_s_setprio(1); else if (tg == 2) __builtin_amdgcn_s_setprio(2); else if (tg == 3) __builtin_amdgcn_s_setprio(3); }
        if (VAR & 2) { for (unsigned i = 0; i < tg; ++i) { __builtin_amdgcn_s_sleep(127); __builtin_amdgcn_s_sleep(127); __builtin_amdgcn_s_sleep(127); __builtin_amdgcn_s_sleep(127); } }
    }
    __shared__ float As[2][128][LDK];
    __shared__ float Bs[2][128][LDK];
    __shared__ float pad[PAD ? PAD : 1];       // PAD > 0: LDS ballast so that at most two workgroups fit a CU
    if (PAD && threadIdx.x == 9999) pad[threadIdx.x % PAD] = 1.f;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5, wr = wave >> 1, wc = wave & 1;
    const int crow = tid >> 1, cq = (tid & 1) * 2;
    for (int i = tid; i < 2 * 128 * LDK; i += 256) { (&As[0][0][0])[i] = 0.001f * (float)(i % 17) - 0.008f; (&Bs[0][0][0])[i] = 0.002f * (float)(i % 13) - 0.01f; }
    __syncthreads();
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    f32x4 af[2][2], bf[2][2];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) { af[rt][0] = f32x4{0.1f, 0.2f, -0.1f, 0.05f} * (float)(lane % 7); af[rt][1] = af[rt][0] * 0.5f; bf[rt][0] = af[rt][0] * 0.25f; bf[rt][1] = af[rt][0] * 0.125f; }
    f32x4 ga0 = af[0][0], ga1 = af[0][1], gb0 = bf[0][0], gb1 = bf[0][1];
    f32x4 ha0 = ga0, ha1 = ga1, hb0 = gb0, hb1 = gb1;
    const size_t gbase = ((size_t)blockIdx.x * 4099) * 2048;
    for (int kc = 0; kc < nk; ++kc) {
        const int cur = kc & 1;
        if (MODE == 3) {
            const f32x4* pa = reinterpret_cast<const f32x4*>(g + ((gbase + (size_t)kc * 4096) & gmask)) + crow * 4 + cq;
            ga0 = pa[0]; ga1 = pa[1]; gb0 = pa[512]; gb1 = pa[513];
        }
        if (MODE == 4 || MODE == 5) {   // the kernel's sharing: weight chunk by row tile (y), input chunk by site tile (x), 48 chunks, 8 KB each
            const int kk = kc % 48;
            const f32x4* pa = reinterpret_cast<const f32x4*>(g + ((size_t)(blockIdx.z * 8 + blockIdx.y) * 48 + kk) * 2048) + crow * 4 + cq;
            const f32x4* pb = reinterpret_cast<const f32x4*>(g + (size_t)(4 << 20) + ((size_t)(blockIdx.z * 32 + (blockIdx.x & 31)) * 48 + kk) * 2048) + crow * 4 + cq;
            if (MODE == 5) { ha0 = ga0; ha1 = ga1; hb0 = gb0; hb1 = gb1; }      // what arrives now is stored one iteration later
            ga0 = pa[0]; ga1 = pa[1]; gb0 = pb[0]; gb1 = pb[1];
        }
        if (MODE >= 1) {
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) {
                const float* p = &As[cur][64 * wr + 32 * rt + li][lh * 8];
                af[rt][0] = *reinterpret_cast<const f32x4*>(p); af[rt][1] = *reinterpret_cast<const f32x4*>(p + 4);
            }
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
                const float* p = &Bs[cur][64 * wc + 32 * ct + li][lh * 8];
                bf[ct][0] = *reinterpret_cast<const f32x4*>(p); bf[ct][1] = *reinterpret_cast<const f32x4*>(p + 4);
            }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int ct = 0; ct < 2; ++ct)
                    acc[rt][ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[rt][j >> 2][j & 3], bf[ct][j >> 2][j & 3], acc[rt][ct], 0, 0, 0);
        if (MODE >= 2) {
            *reinterpret_cast<f32x4*>(&As[cur ^ 1][crow][cq * 4]) = MODE == 5 ? ha0 : ga0;
            *reinterpret_cast<f32x4*>(&As[cur ^ 1][crow][cq * 4 + 4]) = MODE == 5 ? ha1 : ga1;
            *reinterpret_cast<f32x4*>(&Bs[cur ^ 1][crow][cq * 4]) = MODE == 5 ? hb0 : gb0;
            *reinterpret_cast<f32x4*>(&Bs[cur ^ 1][crow][cq * 4 + 4]) = MODE == 5 ? hb1 : gb1;
            __syncthreads();
        }
        if (MODE == 0) {   // keep the operands loop-carried so that nothing is hoisted
            af[0][0][0] += 1e-9f;
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) s += acc[i][j][0] + acc[i][j][15];
    out[((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 256 + tid] = s;
}

template <int MODE, int PAD, int VAR = 0>
void run(const char* name, const float* g, float* out, int grid, dim3 g3 = dim3(0, 0, 0))
{
    const dim3 gd = g3.x ? g3 : dim3(grid);
    const int nk = 4000;
    hipLaunchKernelGGL((k<MODE, PAD, VAR>), gd, dim3(256), 0, 0, g, out, 100, (size_t)(16u << 20) - 1); hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0); hipLaunchKernelGGL((k<MODE, PAD, VAR>), gd, dim3(256), 0, 0, g, out, nk, (size_t)(16u << 20) - 1); hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flop = (double)grid * 4 * nk * 32 * 4096.0;
    printf("%-48s grid %4d: %8.2f ms  %6.1f TFLOP/s\n", name, grid, ms, flop / (ms * 1e-3) / 1e12);
}
int main()
{
    float *g, *out; hipMalloc(&g, ((size_t)128 << 20)); hipMemset(g, 0, ((size_t)128 << 20)); hipMalloc(&out, 2048 * 256 * 4);
    for (int grid : {512, 1024}) {
        run<0, 0>("REG  operands in registers", g, out, grid);
        run<1, 0>("LDS  + 8 ds_read_b128 per chunk", g, out, grid);
        run<2, 0>("BAR  + 4 ds_write_b128 + barrier per chunk", g, out, grid);
        run<3, 0>("GLD  + 4 global_load_dwordx4 per chunk", g, out, grid);
        run<0, 9216>("REG, at most 2 workgroups per CU (LDS ballast)", g, out, grid);
        run<3, 9216>("GLD, at most 2 workgroups per CU (LDS ballast)", g, out, grid);
    }
    run<4, 0>("KRN  loads shared as in the kernel, grid (32, 8, 2)", g, out, 512, dim3(32, 8, 2));
    run<4, 0>("KRN  loads shared as in the kernel, grid (32, 8, 4)", g, out, 1024, dim3(32, 8, 4));
    run<4, 0>("KRN  loads shared as in the kernel, grid (128, 8, 2)", g, out, 2048, dim3(128, 8, 2));
    run<4, 0, 1>("KRN  + s_setprio(workgroup slot), grid (32, 8, 4)", g, out, 1024, dim3(32, 8, 4));
    run<4, 0, 2>("KRN  + start skew by workgroup slot, grid (32, 8, 4)", g, out, 1024, dim3(32, 8, 4));
    run<4, 0, 3>("KRN  + both, grid (32, 8, 4)", g, out, 1024, dim3(32, 8, 4));
    run<4, 0, 1>("KRN  + s_setprio(workgroup slot), grid (128, 8, 2)", g, out, 2048, dim3(128, 8, 2));
    run<4, 0, 2>("KRN  + start skew by workgroup slot, grid (128, 8, 2)", g, out, 2048, dim3(128, 8, 2));
    run<2, 0, 1>("BAR  + s_setprio(workgroup slot)", g, out, 1024);
    run<2, 0, 2>("BAR  + start skew by workgroup slot", g, out, 1024);
    run<5, 0>("KRN2 the same, global loads two chunks ahead, grid (32, 8, 2)", g, out, 512, dim3(32, 8, 2));
    run<5, 0>("KRN2 the same, global loads two chunks ahead, grid (32, 8, 4)", g, out, 1024, dim3(32, 8, 4));
    run<5, 13824>("KRN2 + at most 2 workgroups per CU, grid (32, 8, 2)", g, out, 512, dim3(32, 8, 2));
    run<4, 13824>("KRN  + at most 2 workgroups per CU, grid (32, 8, 2)", g, out, 512, dim3(32, 8, 2));
    return 0;
}
