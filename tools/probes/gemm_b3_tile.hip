// Probe: would a larger workgroup tile lift the bf16x3 tile GEMM (hap_gemm.hpp, AR = 2) above its 60 % MFMA-busy?  The kernel's loop in
// miniature - per K chunk of 16: weight planes (96 B per row) and fp32 activations (64 B per site) from an L2-resident array, the
// activations split into three bf16 planes on their way into LDS, one barrier, fragment reads, six products per (row tile, site tile) -
// for three shapes:
//   T128   128 rows x 128 sites, 4 waves of 64 x 64     (the kernel: 48 KB LDS, 24 MFMAs per wave and barrier)
//   T256R  256 rows x 128 sites, 4 waves of 128 x 64    (72 KB LDS, 48 MFMAs per wave and barrier, 128 accumulator registers)
//   T256W  256 rows x 128 sites, 8 waves of 64 x 64     (72 KB LDS, 24 MFMAs per wave and barrier, the split shared by 8 waves)
// Same total work (rows 1024 x sites 16384 x K 800 per launch).  Reports ms and executed TFLOP/s (6 MFMA products priced).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 b8_t __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void split3_b8(const f32x4& lo, const f32x4& hi, b8_t& p0, b8_t& p1, b8_t& p2)
{
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float v = j < 4 ? lo[j & 3] : hi[j & 3];
        const __bf16 a = (__bf16)v; const float r1 = v - (float)a; const __bf16 b = (__bf16)r1;
        p0[j] = a; p1[j] = b; p2[j] = (__bf16)(r1 - (float)b);
    }
}

// ROWS x SITES per workgroup, waves of (32 RT) rows x (32 CT) sites; MINB: workgroups per CU asked of the compiler
template <int ROWS, int SITES, int RT, int CT, int MINB>
__global__ __launch_bounds__(64 * (ROWS / (32 * RT)) * (SITES / (32 * CT)), MINB) void k(const float* __restrict__ w, const float* __restrict__ x, float* out, int nk)
{
    constexpr int WR = ROWS / (32 * RT), WC = SITES / (32 * CT), WAVES = WR * WC;
    constexpr int NT = 64 * WAVES;
    __shared__ float As[2][ROWS][24];
    __shared__ float Bs[2][SITES][24];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
    const int wr = wave / WC, wc = wave % WC;
    const float* wt = w + (size_t)blockIdx.y * nk * (ROWS * 24);
    const float* xt = x + (size_t)blockIdx.x * nk * (SITES * 16);
    // staging: A = ROWS x 96 B = ROWS * 6 pieces of 16 B; B = 128 x 64 B = 512 pieces of 16 B -> 256 "row halves" of 32 B
    constexpr int APT = (ROWS * 6 + NT - 1) / NT;             // A pieces of 16 B per thread
    constexpr int BPT = (SITES * 2 + NT - 1) / NT;            // B row halves (32 B = 8 activations) per thread
    f32x4 ga[APT], gb0[BPT], gb1[BPT];
    auto gload = [&](int kc) {
        const f32x4* pa = reinterpret_cast<const f32x4*>(wt + (size_t)kc * (ROWS * 24));
#pragma unroll
        for (int i = 0; i < APT; ++i) if (tid + NT * i < ROWS * 6) ga[i] = pa[tid + NT * i];
#pragma unroll
        for (int i = 0; i < BPT; ++i) if (tid + NT * i < SITES * 2) {
            const int h = tid + NT * i;
            const f32x4* pb = reinterpret_cast<const f32x4*>(xt + (size_t)kc * (SITES * 16)) + (h >> 1) * 4 + (h & 1) * 2; gb0[i] = pb[0]; gb1[i] = pb[1];
        }
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < APT; ++i) if (tid + NT * i < ROWS * 6) { const int p = tid + NT * i, r = p / 6, s = p % 6; reinterpret_cast<f32x4*>(&As[buf][r][0])[s ^ ((r >> 3) & 1)] = ga[i]; }
#pragma unroll
        for (int i = 0; i < BPT; ++i) if (tid + NT * i < SITES * 2) {
            const int h = tid + NT * i, crow = h >> 1;
            b8_t p0, p1, p2; split3_b8(gb0[i], gb1[i], p0, p1, p2);
            const int sw = (crow >> 3) & 1, hb = (h & 1) ^ sw;
            b8_t* rb = reinterpret_cast<b8_t*>(&Bs[buf][crow][0]);
            rb[hb] = p0; rb[2 + hb] = p1; rb[4 + hb] = p2;
        }
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
        f32x4 af[RT][3], bf[CT][3];
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            const int sl = (2 * p + (lh ^ ((li >> 3) & 1))) * 4;
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) af[rt][p] = *reinterpret_cast<const f32x4*>(&As[cur][32 * RT * wr + 32 * rt + li][sl]);
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) bf[ct][p] = *reinterpret_cast<const f32x4*>(&Bs[cur][32 * CT * wc + 32 * ct + li][sl]);
        }
        constexpr int WP[6] = {0, 1, 2, 0, 1, 0}, XP[6] = {2, 1, 0, 1, 0, 0};
#pragma unroll
        for (int g = 0; g < 6; ++g) {
            if (g == 1 && kc + 1 < nk) { __builtin_amdgcn_sched_barrier(0); lstore(cur ^ 1); __builtin_amdgcn_sched_barrier(0); }
            if (g == 3 && kc + 2 < nk) { __builtin_amdgcn_sched_barrier(0); gload(kc + 2); __builtin_amdgcn_sched_barrier(0); }
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int ct = 0; ct < CT; ++ct)
                    acc[rt][ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(b8_t, af[rt][WP[g]]), __builtin_bit_cast(b8_t, bf[ct][XP[g]]), acc[rt][ct], 0, 0, 0);
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
    for (int r = 0; r < 20; ++r) hipLaunchKernelGGL(kern, grid, dim3(NT), 0, 0, w, x, out, nk);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 20;
    const double flop = 2.0 * rows * (double)sites * (nk * 16) * 6;
    int nb = 0; hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kern, NT, 0);
    hipFuncAttributes fa; hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(kern));
    printf("%-46s %7.3f ms  %7.1f TFLOP/s executed = %.3f of 2500   (%d workgroups of %d waves per CU, %d registers, %zu B LDS)\n", name, ms, flop / (ms * 1e-3) / 1e12,
           flop / (ms * 1e-3) / 2.5e15, nb, NT / 64, fa.numRegs, fa.sharedSizeBytes);
}
int main()
{
    float *w, *x, *out;
    const size_t wn = (size_t)8 * 50 * 256 * 24, xn = (size_t)256 * 50 * 128 * 16;
    hipMalloc(&w, wn * 4); hipMalloc(&x, xn * 4); hipMalloc(&out, (size_t)4096 * 1024 * 4);
    float* h = (float*)malloc(xn * 4);
    for (size_t i = 0; i < xn; ++i) h[i] = (float)((i * 2654435761u) % 2001) / 1000.f - 1.f;
    hipMemcpy(x, h, xn * 4, hipMemcpyHostToDevice);
    uint16_t* hw = (uint16_t*)h;
    for (size_t i = 0; i < wn * 2; ++i) hw[i] = (uint16_t)(0x3c00 + ((i * 40503u) & 0x3ff) + (((i >> 4) % 3) ? 0 : 0x8000));     // bf16 near +-0.01..0.03
    hipMemcpy(w, hw, wn * 4, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 2; ++rep) {
        run<128, 128, 2, 2, 3>("128 x 128, 4 waves of 64 x 64 (the kernel)", w, x, out);
        run<128, 128, 2, 2, 2>("128 x 128, 4 waves of 64 x 64, 2 per CU", w, x, out);
        run<256, 128, 4, 2, 1>("256 x 128, 4 waves of 128 x 64", w, x, out);
        run<256, 128, 2, 2, 1>("256 x 128, 8 waves of 64 x 64", w, x, out);
        run<128, 256, 2, 2, 1>("128 x 256, 8 waves of 64 x 64", w, x, out);
        run<256, 128, 2, 4, 2>("256 x 128, 4 waves of 64 x 128, 2 per CU", w, x, out);
        run<128, 256, 4, 2, 2>("128 x 256, 4 waves of 128 x 64, 2 per CU", w, x, out);
        run<128, 256, 2, 4, 2>("128 x 256, 4 waves of 64 x 128 (2 x 2), 2 per CU", w, x, out);
        run<256, 256, 4, 2, 1>("256 x 256, 8 waves of 128 x 64", w, x, out);
        run<256, 256, 2, 4, 1>("256 x 256, 8 waves of 64 x 128", w, x, out);
        run<256, 256, 2, 2, 1>("256 x 256, 16 waves of 64 x 64", w, x, out);
        run<128, 128, 1, 2, 2>("128 x 128, 8 waves of 32 x 64", w, x, out);
        run<128, 128, 2, 1, 2>("128 x 128, 8 waves of 64 x 32", w, x, out);
        printf("\n");
    }
    return 0;
}
