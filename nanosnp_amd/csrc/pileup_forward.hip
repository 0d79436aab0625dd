// pileup_forward.hip -- PileupModel forward (2-layer BiLSTM H=64 over 33 positions + heads) as
// fp32 MFMA kernels for gfx950.
//
// Replaces LSTMNetwork.predict (PileupModel/model.py:114-119): BaseEncoder.forward
// (model.py:31-39), ForwardLayer.forward (model.py:66-73) and the two softmaxes, called from
// PileupModel/predict.py:49-51.  Not a translation: the reference runs nn.LSTM (cuDNN) on all 33
// steps of both layers and the dense layers on all 33 positions; here
//
//   K1 k_pileup_l0    layer-0 recurrence, one workgroup per (128-site tile, direction).  W_hh and
//                     the first 16 input channels live in LDS as MFMA A-operand images (80 KB, so
//                     two workgroups share a CU); each wave owns 16 sites and ALL 256 gate rows,
//                     so h_t never leaves registers: the accumulator layout of
//                     v_mfma_f32_16x16x4_f32 (lane = site + 16*q, regs = gates i,f,g,o of hidden
//                     unit 4*tile+q) is exactly the B-operand layout of the next step.
//   K2 k_pileup_proj1 layer-1 input projection W_ih1 . [h0_fwd(t); h0_bwd(t)] + b for the 17
//                     steps per direction that reach position 16 (only position 16 is consumed,
//                     model.py:68), a [N*17 x 128] x [128 x 256] GEMM with W_ih1 (128 KB) in LDS.
//   K3 k_pileup_l1    layer-1 recurrence, 17 steps, accumulators initialised from K2's output.
//   K4 k_pileup_head  output_proj(128->128) at t=16, tanh(dense 128->256), genotype(21) and
//                     zygosity(3) heads, softmax; weights streamed from L2.
//
// All dot products are exact-fp32 MFMA (v_mfma_f32_16x16x4_f32 == a k-ordered fmaf chain);
// sigmoid/tanh use v_exp_f32 / v_rcp_f32 (1 ulp).  Reduced schedule: 6.29 MFLOP/site executed
// vs 12.55 MFLOP/site in the reference schedule, results identical to fp32 rounding.
#include "nsnp_common.hpp"
#include "nsnp_lstm_cell.hpp"
#include "nsnp_devclock.hpp"

namespace {

constexpr float LOG2E = 1.4426950408889634f;

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c)
{
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ float sigmoid_f(float x)
{
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-LOG2E * x));
}
__device__ __forceinline__ float tanh_f(float x)
{
    // 1 - 2/(1+e^{2x}); saturates cleanly to +-1 through exp2 -> inf / 0
    return __builtin_fmaf(-2.0f, __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f((2.0f * LOG2E) * x)), 1.0f);
}

// LSTM cell: nsnp_lstm_cell.hpp (shared with hap_gemm.hpp and the bf16x3 kernels; gate rows pre-scaled by nsnp_pileup_pack_weights).
// c is updated in place, h' returned.
__device__ __forceinline__ float lstm_cell(float zi, float zf, float zg, float zo, float& c)
{
    float cn;
    const float h = nsnp_cell::lstm_cell(zi, zf, zg, zo, c, cn);
    c = cn;
    return h;
}

// acc[NT] += W(image rows, K-steps [4*J4B, 4*(J4B+J4N))) . b, with the weight image read 16 B per
// lane per 4 K-steps.  w points at image element [tile 0][j4 0][lane 0]; NJ4 is the image's j4
// extent and TB the first of the NT tiles to compute (acc is indexed from 0).
// Tiles are walked in groups of 4 so that dependent MFMAs on one accumulator are
// 4 issue slots apart (dependent latency 40 > issue 32 cycles).
template <int NT, int NJ4, int J4B, int J4N, int TB = 0, typename WP>
__device__ __forceinline__ void wave_gemm(WP w, int lane, const float* b, f32x4* acc)
{
#pragma unroll
    for (int j = 0; j < J4N; ++j) {
#pragma unroll
        for (int ig = 0; ig < NT; ig += 4) {
            f32x4 a[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) a[u] = w[((TB + ig + u) * NJ4 + (J4B + j)) * 64 + lane];
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int u = 0; u < 4; ++u) acc[ig + u] = mfma4(a[u][e], b[4 * j + e], acc[ig + u]);
            // keep hipcc from hoisting every later weight read above these MFMAs (it spills
            // hundreds of VGPRs otherwise); latency is covered by the other waves on the SIMD
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// two-tile variant (the heads' 32 padded rows): the generic path walks tiles in groups of 4
__device__ __forceinline__ void wave_gemm2x16(const f32x4* w, int lane, const float* b, f32x4* acc)
{
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const f32x4 a0 = w[(0 * 16 + j) * 64 + lane], a1 = w[(1 * 16 + j) * 64 + lane];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            acc[0] = mfma4(a0[e], b[4 * j + e], acc[0]);
            acc[1] = mfma4(a1[e], b[4 * j + e], acc[1]);
        }
    }
}

// LSTM cell update for 8 of the 16 hidden units a lane owns (one per tile)
__device__ __forceinline__ void lstm_pointwise8(const f32x4* acc, float* c, float* h)
{
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        h[i] = lstm_cell(acc[i][0], acc[i][1], acc[i][2], acc[i][3], c[i]);
    }
}

__device__ __forceinline__ void copy_to_lds(f32x4* dst, const float* __restrict__ src, int n_f32x4,
                                            int tid, int nthreads)
{
    const f32x4* s4 = reinterpret_cast<const f32x4*>(src);
    for (int i = tid; i < n_f32x4; i += nthreads) dst[i] = s4[i];
}

// ---------------------------------------------------------------------------------------------
// K1: layer 0.  grid = (ceil(N/128), 2 directions), block = 512 (8 waves x 16 sites).
// LDS: [W_hh image 64 KB][W_ih image (channels 0..15) 16 KB] = 80 KB -> 2 workgroups per CU.
// H0 layout: [site][t][dir][q][16] fp32 = what K2 reads back with 16-byte loads.
// ---------------------------------------------------------------------------------------------
constexpr int L0_WHH_F4 = 16 * 4 * 64;   // f32x4 elements
constexpr int L0_WIH_F4 = 16 * 1 * 64;
constexpr int L0_LDS_BYTES = (L0_WHH_F4 + L0_WIH_F4) * 16;

// WAVES (8, 4, 2 or 1) is picked by the launcher so that small batches still spread over all CUs:
// LDS admits two workgroups per CU whatever their size.
template <int WAVES>
__global__ __launch_bounds__(64 * WAVES, (WAVES >= 8 ? 4 : (WAVES == 4 ? 2 : 1))) void k_pileup_l0(
    const int32_t* __restrict__ x, const int64_t* __restrict__ center_idx, int64_t N,
    const float* __restrict__ whh0, const float* __restrict__ whh1,
    const float* __restrict__ wih0, const float* __restrict__ wih1,
    const float* __restrict__ wlast0, const float* __restrict__ wlast1,
    float* __restrict__ H0)
{
    extern __shared__ f32x4 lds[];
    const int dir = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q = lane >> 4;
    copy_to_lds(lds, dir ? whh1 : whh0, L0_WHH_F4, tid, 64 * WAVES);
    copy_to_lds(lds + L0_WHH_F4, dir ? wih1 : wih0, L0_WIH_F4, tid, 64 * WAVES);
    const float* __restrict__ wlast = dir ? wlast1 : wlast0;
    __syncthreads();

    const int64_t site = (int64_t)blockIdx.x * (16 * WAVES) + wave * 16 + (lane & 15);
    const bool live = site < N;
    const int64_t sc = live ? site : N - 1;
    // window base: gathered [N,33,18] or straight out of the per-column count matrix
    const int32_t* __restrict__ xs = center_idx ? x + (center_idx[sc] - PCENTER) * PC
                                                : x + sc * (PW * PC);
    const int klast = q < 2 ? 16 + q : 16;   // lanes q=2,3 feed the bias / a zero instead
    float* __restrict__ hout = H0 + (sc * PW * 2 + dir) * 64 + q * 16;

    float c[16], h[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { c[i] = 0.f; h[i] = 0.f; }

    int xi[5];
    {
        const int t0 = dir ? PW - 1 : 0;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) xi[kk] = xs[t0 * PC + 4 * kk + q];
        xi[4] = xs[t0 * PC + klast];
    }
    for (int s = 0; s < PW; ++s) {
        const int t = dir ? PW - 1 - s : s;
        float xb[4];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) xb[kk] = (float)xi[kk];   // predict.py:49 int -> float
        const float xl = q == 2 ? 1.0f : (q == 3 ? 0.0f : (float)xi[4]);
        if (s + 1 < PW) {   // prefetch the next position
            const int tn = dir ? t - 1 : t + 1;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) xi[kk] = xs[tn * PC + 4 * kk + q];
            xi[4] = xs[tn * PC + klast];
        }
        // two half-passes of 8 gate tiles each keep the live accumulators at 32 registers and let
        // the second half's MFMAs overlap the first half's sigmoid/tanh work
        float hn[16];
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            f32x4 acc[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (hf == 0) wave_gemm<8, 1, 0, 1, 0>(lds + L0_WHH_F4, lane, xb, acc);
            else         wave_gemm<8, 1, 0, 1, 8>(lds + L0_WHH_F4, lane, xb, acc);
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = mfma4(wlast[(hf * 8 + i) * 64 + lane], xl, acc[i]);
            __builtin_amdgcn_sched_barrier(0);
            if (s > 0) {
                if (hf == 0) wave_gemm<8, 4, 0, 4, 0>(lds, lane, h, acc);
                else         wave_gemm<8, 4, 0, 4, 8>(lds, lane, h, acc);
            }
            lstm_pointwise8(acc, c + hf * 8, hn + hf * 8);
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) h[i] = hn[i];
        if (live) {
            f32x4* o = reinterpret_cast<f32x4*>(hout + (int64_t)t * 128);
#pragma unroll
            for (int v = 0; v < 4; ++v) o[v] = f32x4{h[4 * v], h[4 * v + 1], h[4 * v + 2], h[4 * v + 3]};
        }
    }
}

// ---------------------------------------------------------------------------------------------
// K2: layer-1 input projection.  grid = (G, 2 directions) persistent, block = 1024 (16 waves).
// LDS: [W_ih1 image 128 KB][bias image 4 KB].  Rows m = site*17 + u, u = step index of
// direction d (t = u for fwd, 32-u for reverse).  Output Xp1[d][m][tile][q][4] is the
// accumulator image K3 starts each step from.
// ---------------------------------------------------------------------------------------------
constexpr int P1_W_F4 = 16 * 8 * 64;
constexpr int P1_B_F4 = 16 * 64;
constexpr int P1_LDS_BYTES = (P1_W_F4 + P1_B_F4) * 16;

__global__ __launch_bounds__(1024, 4) void k_pileup_proj1(
    const float* __restrict__ H0, int64_t N,
    const float* __restrict__ w0, const float* __restrict__ w1,
    const float* __restrict__ b0, const float* __restrict__ b1,
    float* __restrict__ Xp1)
{
    extern __shared__ f32x4 lds[];
    const int dir = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q = lane >> 4;
    copy_to_lds(lds, dir ? w1 : w0, P1_W_F4, tid, 1024);
    copy_to_lds(lds + P1_W_F4, dir ? b1 : b0, P1_B_F4, tid, 1024);
    __syncthreads();
    const int64_t M = N * PSTEPS1;
    const int64_t n_rt = NSNP_CDIV(M, 16);
    float* __restrict__ out = Xp1 + (int64_t)dir * M * 256;
    for (int64_t rt = (int64_t)blockIdx.x * 16 + wave; rt < n_rt; rt += (int64_t)gridDim.x * 16) {
        const int64_t m = rt * 16 + (lane & 15);
        const bool live = m < M;
        const int64_t mc = live ? m : M - 1;
        const int64_t site = mc / PSTEPS1;
        const int u = (int)(mc - site * PSTEPS1);
        const int t = dir ? PW - 1 - u : u;
        const f32x4* __restrict__ hin = reinterpret_cast<const f32x4*>(H0 + (site * PW + t) * 128 + q * 16);
        float b[32];
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const f32x4 a = hin[v], bb = hin[16 + v];   // fwd half, reverse half (+64 floats)
#pragma unroll
            for (int e = 0; e < 4; ++e) { b[4 * v + e] = a[e]; b[16 + 4 * v + e] = bb[e]; }
        }
        f32x4* o = reinterpret_cast<f32x4*>(out + mc * 256 + q * 4);
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            f32x4 acc[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = lds[P1_W_F4 + (hf * 8 + i) * 64 + lane];
            if (hf == 0) wave_gemm<8, 8, 0, 8, 0>(lds, lane, b, acc);
            else         wave_gemm<8, 8, 0, 8, 8>(lds, lane, b, acc);
            if (live) {
#pragma unroll
                for (int i = 0; i < 8; ++i) o[(hf * 8 + i) * 4] = acc[i];
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// K3: layer-1 recurrence, 17 steps.  grid = (ceil(N/128), 2), block = 512.  LDS: W_hh1 image 64 KB.
// Writes h1 at position 16 to H1c[site][dir][q][16].
// ---------------------------------------------------------------------------------------------
constexpr int L1_LDS_BYTES = L0_WHH_F4 * 16;

template <int WAVES>
__global__ __launch_bounds__(64 * WAVES, (WAVES >= 8 ? 4 : (WAVES == 4 ? 2 : 1))) void k_pileup_l1(
    const float* __restrict__ Xp1, int64_t N,
    const float* __restrict__ whh0, const float* __restrict__ whh1,
    float* __restrict__ H1c)
{
    extern __shared__ f32x4 lds[];
    const int dir = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q = lane >> 4;
    copy_to_lds(lds, dir ? whh1 : whh0, L0_WHH_F4, tid, 64 * WAVES);
    __syncthreads();
    const int64_t site = (int64_t)blockIdx.x * (16 * WAVES) + wave * 16 + (lane & 15);
    const bool live = site < N;
    const int64_t sc = live ? site : N - 1;
    const int64_t M = N * PSTEPS1;
    const f32x4* __restrict__ xin =
        reinterpret_cast<const f32x4*>(Xp1 + ((int64_t)dir * M + sc * PSTEPS1) * 256 + q * 4);
    float c[16], h[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { c[i] = 0.f; h[i] = 0.f; }
    for (int u = 0; u < PSTEPS1; ++u) {
        float hn[16];
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            f32x4 acc[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = xin[(int64_t)u * 64 + (hf * 8 + i) * 4];
            if (u > 0) {
                if (hf == 0) wave_gemm<8, 4, 0, 4, 0>(lds, lane, h, acc);
                else         wave_gemm<8, 4, 0, 4, 8>(lds, lane, h, acc);
            }
            lstm_pointwise8(acc, c + hf * 8, hn + hf * 8);
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) h[i] = hn[i];
    }
    if (live) {
        f32x4* o = reinterpret_cast<f32x4*>(H1c + site * 128 + dir * 64 + q * 16);
#pragma unroll
        for (int v = 0; v < 4; ++v) o[v] = f32x4{h[4 * v], h[4 * v + 1], h[4 * v + 2], h[4 * v + 3]};
    }
}

// ---------------------------------------------------------------------------------------------
// K1r / K23r: REGISTER-STATIONARY fp32 recurrence kernels (default).  K1 / K3 above give one wave 16 sites and ALL 256
// gate rows, so a 4096-site batch is only 512 waves for 1024 SIMDs and every wave re-reads the weight images from LDS
// each step.  Here the gate tiles are split over the waves of a workgroup instead: a wave keeps the A fragments of its
// tiles in VGPRs for the whole kernel (v_mfma_f32_16x16x4_f32 takes ONE VGPR per operand: 84 / 96 registers), and only
// h_t crosses waves, through a double-buffered LDS exchange row per site with one LDS-only barrier per step.  A batch of
// 4096 sites then is 2048 (layer 0) / 2048 (layer 1) waves, two per SIMD, and the matrix pipe - the bound of the fp32
// path - is fed from registers.  Every accumulator sees the same k-ordered MFMA chain as in K1 / K2 + K3, and the cell
// uses the same expressions, so the results are bit-identical to the LDS-image kernels (tests/test_gpu_pileup_forward.py).
//
// Exchange row of a site (layer-0 h_t and layer-1 h_t alike): 64 floats, the unit 16j + 4e + q at position
// 16j + 4q + e, rows 72 floats apart.  The B fragment of lane (site n, quarter q) for K-steps 4j .. 4j+3 (units
// 4(4j+e) + q) is then ONE ds_read_b128 at position 16j + 4q, the four units lane (n, q) of wave w leaves the cell with
// (tiles 4w + u) are ONE ds_write_b128 at 16w + 4q, and with an 18-slot row stride both hit 16 distinct 16-byte bank
// slots in every 16-lane group of the instruction (MI355X_MICROARCH.md, LDS table).
// ---------------------------------------------------------------------------------------------
constexpr int RS_XROW = 72;                  // floats per exchange row: 64 + 8 pad (18 slots of 16 B)
constexpr int RS_XSROW = 24;                 // floats per staged input row: 20 used (16 + 2 channels, 1, 0) + pad
constexpr int RS_H0ROW = 136;                // floats per staged h0 row: 128 + 8 pad (34 slots)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// K1r: layer 0.  grid = (ceil(N / (16 NSG)), 2 directions), block = 256: wave w owns gate tiles 4w .. 4w+3 (hidden units
// 16w .. 16w+15, all four gates) for NSG groups of 16 sites.
template <int NSG, bool WXL = false>
__global__ __launch_bounds__(256, (WXL ? 4 : (NSG == 1 ? 3 : 2))) void k_pileup_l0_rs32(
    const int32_t* __restrict__ x, const int64_t* __restrict__ center_idx, int64_t N,
    const float* __restrict__ whh0, const float* __restrict__ whh1,
    const float* __restrict__ wih0, const float* __restrict__ wih1,
    const float* __restrict__ wlast0, const float* __restrict__ wlast1,
    float* __restrict__ H0)
{
    __shared__ __attribute__((aligned(16))) float hx[2][16 * NSG][RS_XROW];
    __shared__ __attribute__((aligned(16))) float xx[2][16 * NSG * RS_XSROW];
    // WXL: the input-part fragments (20 of the 84 weight registers) stay in LDS instead, [wave][tile][lane] f32x4 + [wave][tile][lane] float:
    // the kernel then fits 128 VGPRs and FOUR workgroups share a SIMD set instead of three
    __shared__ __attribute__((aligned(16))) float wxl[WXL ? 4 * 4 * 64 * 5 : 4];
    const int dir = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 15, q = lane >> 4;
    NSNP_DEVCLK_START
    const int64_t base_site = (int64_t)blockIdx.x * (16 * NSG);

    // ---- this wave's four gate tiles -> registers (the images K1 stages in LDS) ----
    f32x4 Whh[4][4], Wih[4];
    float Wl[4];
    {
        const f32x4* __restrict__ ghh = reinterpret_cast<const f32x4*>(dir ? whh1 : whh0);     // [tile][j4 4][lane]
        const f32x4* __restrict__ gih = reinterpret_cast<const f32x4*>(dir ? wih1 : wih0);     // [tile][lane]
        const float* __restrict__ gl = dir ? wlast1 : wlast0;                                   // [tile][lane]
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int j = 0; j < 4; ++j) Whh[u][j] = ghh[((4 * wave + u) * 4 + j) * 64 + lane];
            Wih[u] = gih[(4 * wave + u) * 64 + lane];
            Wl[u] = gl[(4 * wave + u) * 64 + lane];
            if (WXL) {
                *reinterpret_cast<f32x4*>(&wxl[((wave * 4 + u) * 64 + lane) * 4]) = Wih[u];
                wxl[4 * 4 * 64 * 4 + (wave * 4 + u) * 64 + lane] = Wl[u];
            }
        }
    }

    // ---- input windows: the 18 counts of (site, t) are 72 contiguous bytes; thread i < 144 NSG / 16 ... moves one 8-byte
    // piece per step (9 pieces per site), converts it to fp32 (predict.py:49) and stores it where the B fragments want it:
    // xx[site][4 q + e] = channel 4 e + q (e < 4), xx[site][16 + q] = channels 16, 17, the constant 1 the bias rides on, 0.
    // One workgroup-wide copy per step instead of five scattered dword loads per lane in each of the four waves.
    constexpr int NPIECE = 9 * 16 * NSG;                       // 8-byte pieces per step
    constexpr int PPT = (NPIECE + 255) / 256;                  // pieces per thread
    const int32_t* xsrc[PPT]; int xdst[PPT][2]; bool xon[PPT];
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
        const int pi = tid + 256 * k;
        xon[k] = pi < NPIECE;
        const int row = xon[k] ? pi / 9 : 0, piece = xon[k] ? pi % 9 : 0;
        const int64_t site = base_site + row;
        const int64_t sc = site < N ? site : N - 1;
        xsrc[k] = (center_idx ? x + (center_idx[sc] - PCENTER) * PC : x + sc * (PW * PC)) + 2 * piece;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int ch = 2 * piece + h;                       // channel -> position in the staged row
            xdst[k][h] = row * RS_XSROW + (ch < 16 ? 4 * (ch & 3) + (ch >> 2) : 16 + (ch - 16));
        }
    }
    int2 xi[PPT];
    auto load_x = [&](int t) {
#pragma unroll
        for (int k = 0; k < PPT; ++k) if (xon[k]) xi[k] = *reinterpret_cast<const int2*>(xsrc[k] + t * PC);
    };
    auto stage_x = [&](int buf) {
#pragma unroll
        for (int k = 0; k < PPT; ++k) if (xon[k]) {
            xx[buf][xdst[k][0]] = (float)xi[k].x;              // predict.py:49 int -> float
            xx[buf][xdst[k][1]] = (float)xi[k].y;
        }
    };
    for (int i = tid; i < 2 * 16 * NSG; i += 256) {            // the constant-1 / zero tail of every staged row (both buffers)
        float* r = &xx[i / (16 * NSG)][(i % (16 * NSG)) * RS_XSROW];
        r[18] = 1.0f; r[19] = 0.0f;
    }
    load_x(dir ? PW - 1 : 0);
    stage_x(0);
    if (PW > 1) load_x(dir ? PW - 2 : 1);
    __syncthreads();

    float c[NSG][4];
#pragma unroll
    for (int sg = 0; sg < NSG; ++sg)
#pragma unroll
        for (int u = 0; u < 4; ++u) c[sg][u] = 0.f;

    // H0 rows leave through the exchange buffer one step late: thread (row = tid / 16, chunk = tid % 16 = 4 q' + j) moves the
    // 16 bytes at exchange position 16 j + 4 q' to H0 position 16 q' + 4 j, so 16 lanes write one 256-byte row [q][16]
    auto flush_h = [&](int buf, int t) {
        const int cid = tid & 15, qq = cid >> 2, jj = cid & 3;
#pragma unroll
        for (int k = 0; k < NSG; ++k) {
            const int row = (tid >> 4) + 16 * k;
            const int64_t site = base_site + row;
            if (site < N) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(&hx[buf][row][16 * jj + 4 * qq]);
                *reinterpret_cast<f32x4*>(H0 + ((site * PW + t) * 2 + dir) * 64 + 4 * cid) = v;
            }
        }
    };

    for (int s = 0; s < PW; ++s) {
        const int t = dir ? PW - 1 - s : s;
        const int cur = s & 1;
        // x_t of this step was staged during the previous one; x_{t+1} is in flight in registers and is stored behind the MFMAs.
        // (Issuing the flush of h_{t-1} and the staging of x_{t+1} from inside the burst of 84 MFMAs instead - what gave the tile GEMM
        // 14 % - was measured here without effect: 3.078 vs 3.079-3.12 ms at 131072 sites.)
        if (s > 0) flush_h(cur ^ 1, dir ? t + 1 : t - 1);
        if constexpr (NSG >= 2) {
            // Software pipeline over the site groups: the 84 MFMAs of group g+1 are issued AMONG the sigmoid / tanh work of group g
            // (one scheduling region, one transcendental and one plain vector instruction behind every MFMA: a v_mfma_f32_16x16x4_f32
            // holds the vector issue port for 8 of its 32 cycles), so a wave keeps the matrix pipe fed through its own cells.
            struct Frag { f32x4 xb; float xl; f32x4 hb[4]; };
            auto load_b = [&](int sg, Frag& f) {
                const float* xr = &xx[cur][(16 * sg + n) * RS_XSROW];
                f.xb = *reinterpret_cast<const f32x4*>(xr + 4 * q);
                f.xl = xr[16 + q];
                if (s > 0) {
                    const float* hr = &hx[cur ^ 1][16 * sg + n][4 * q];
#pragma unroll
                    for (int j = 0; j < 4; ++j) f.hb[j] = *reinterpret_cast<const f32x4*>(hr + 16 * j);
                }
            };
            auto gemm = [&](const Frag& f, f32x4* acc) {
#pragma unroll
                for (int u = 0; u < 4; ++u) acc[u] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int u = 0; u < 4; ++u) acc[u] = mfma4(Wih[u][e], f.xb[e], acc[u]);
#pragma unroll
                for (int u = 0; u < 4; ++u) acc[u] = mfma4(Wl[u], f.xl, acc[u]);
                if (s > 0) {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int e = 0; e < 4; ++e)
#pragma unroll
                            for (int u = 0; u < 4; ++u) acc[u] = mfma4(Whh[u][j][e], f.hb[j][e], acc[u]);
                }
            };
            auto cell = [&](int sg, const f32x4* acc) {
                f32x4 hn;
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    hn[u] = lstm_cell(acc[u][0], acc[u][1], acc[u][2], acc[u][3], c[sg][u]);
                }
                *reinterpret_cast<f32x4*>(&hx[cur][16 * sg + n][16 * wave + 4 * q]) = hn;
            };
            Frag fr[2];
            f32x4 acc[2][4];
            load_b(0, fr[0]);
            gemm(fr[0], acc[0]);
#pragma unroll
            for (int sg = 0; sg < NSG; ++sg) {
                __builtin_amdgcn_sched_barrier(0);
                if (sg + 1 < NSG) load_b(sg + 1, fr[(sg + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);
                if (sg + 1 < NSG) gemm(fr[(sg + 1) & 1], acc[(sg + 1) & 1]);
                cell(sg, acc[sg & 1]);
                if (sg + 1 < NSG) {
#pragma unroll
                    for (int i = 0; i < 84; ++i) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // one MFMA
                        __builtin_amdgcn_sched_group_barrier(0x400, 1, 0);      // one transcendental  (cell of the previous group)
                        __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);      // one other vector instruction
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        } else {
#pragma unroll
        for (int sg = 0; sg < NSG; ++sg) {
            // recurrent B fragments are requested one K block (4 K-steps) ahead of the MFMAs that use them, two in flight:
            // the LDS latency hides behind 16 MFMAs and only 8 registers hold fragments
            const float* hr = &hx[cur ^ 1][16 * sg + n][4 * q];
            f32x4 hb0, hb1;
            if (s > 0) { hb0 = *reinterpret_cast<const f32x4*>(hr); hb1 = *reinterpret_cast<const f32x4*>(hr + 16); }
            const float* xr = &xx[cur][(16 * sg + n) * RS_XSROW];
            const f32x4 xb = *reinterpret_cast<const f32x4*>(xr + 4 * q);
            const float xl = xr[16 + q];
            f32x4 acc[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) acc[u] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (WXL) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    Wih[u] = *reinterpret_cast<const f32x4*>(&wxl[((wave * 4 + u) * 64 + lane) * 4]);
                    Wl[u] = wxl[4 * 4 * 64 * 4 + (wave * 4 + u) * 64 + lane];
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int u = 0; u < 4; ++u) acc[u] = mfma4(Wih[u][e], xb[e], acc[u]);
#pragma unroll
            for (int u = 0; u < 4; ++u) acc[u] = mfma4(Wl[u], xl, acc[u]);
            if (s > 0) {
#define RS_KBLOCK(J, HB)                                                                               \
                _Pragma("unroll") for (int e = 0; e < 4; ++e)                                          \
                    _Pragma("unroll") for (int u = 0; u < 4; ++u) acc[u] = mfma4(Whh[u][J][e], HB[e], acc[u]);
                __builtin_amdgcn_sched_barrier(0);
                RS_KBLOCK(0, hb0)
                __builtin_amdgcn_sched_barrier(0);
                hb0 = *reinterpret_cast<const f32x4*>(hr + 32);
                RS_KBLOCK(1, hb1)
                __builtin_amdgcn_sched_barrier(0);
                hb1 = *reinterpret_cast<const f32x4*>(hr + 48);
                RS_KBLOCK(2, hb0)
                __builtin_amdgcn_sched_barrier(0);
                RS_KBLOCK(3, hb1)
#undef RS_KBLOCK
            }
            // cell: lane (n, q) holds the four gates of unit 4 (4 wave + u) + q in acc[u]
            f32x4 hn;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                hn[u] = lstm_cell(acc[u][0], acc[u][1], acc[u][2], acc[u][3], c[sg][u]);
            }
            *reinterpret_cast<f32x4*>(&hx[cur][16 * sg + n][16 * wave + 4 * q]) = hn;
        }
        }
        if (s + 1 < PW) {
            stage_x(cur ^ 1);                                                   // x of step s+1 (its buffer was last read in step s-1)
            if (s + 2 < PW) load_x(dir ? t - 2 : t + 2);
        }
        lds_barrier();
    }
    flush_h((PW - 1) & 1, dir ? 0 : PW - 1);
    NSNP_DEVCLK_STOP(0)
}

// K23r: layer 1, input projection FUSED into the recurrence (no Xp1 round trip: 70 KB/site less HBM traffic than K2 + K3).
// grid = (ceil(N / (16 NSG)), 2), block = 512: wave w owns gate tiles 2w, 2w+1 (W_ih1 K = 128 and W_hh1 K = 64: 96 VGPRs).
// LDS: the h0_t rows of the workgroup's sites copied from H0 one step ahead (double-buffered; the 16-byte chunk (d, q, c) of
// a row is stored at slot 16 d + 4 c + q so that the fragment reads are conflict-free) and the h1 exchange rows.
constexpr int rs32_l1_lds_bytes(int nsg) { return (2 * 16 * nsg * RS_H0ROW + 2 * 16 * nsg * RS_XROW) * 4; }

template <int NSG, bool STAGGER>
__global__ __launch_bounds__(512, 2) void k_pileup_l1_rs32(
    const float* __restrict__ H0, int64_t N,
    const float* __restrict__ wih0, const float* __restrict__ wih1,
    const float* __restrict__ whh0, const float* __restrict__ whh1,
    const float* __restrict__ bias0, const float* __restrict__ bias1,
    float* __restrict__ H1c)
{
    extern __shared__ f32x4 lds[];
    constexpr int NS = 16 * NSG;
    float* const h0s = reinterpret_cast<float*>(lds);                  // [2][NS][RS_H0ROW]
    float* const h1x = h0s + 2 * NS * RS_H0ROW;                         // [2][NS][RS_XROW]
    const int dir = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 15, q = lane >> 4;
    const int64_t base_site = (int64_t)blockIdx.x * NS;

    f32x4 Wih[2][8], Whh[2][4], bias[2];
    {
        const f32x4* __restrict__ gih = reinterpret_cast<const f32x4*>(dir ? wih1 : wih0);      // [tile][j4 8][lane]
        const f32x4* __restrict__ ghh = reinterpret_cast<const f32x4*>(dir ? whh1 : whh0);      // [tile][j4 4][lane]
        const f32x4* __restrict__ gb = reinterpret_cast<const f32x4*>(dir ? bias1 : bias0);     // [tile][lane]
#pragma unroll
        for (int u = 0; u < 2; ++u) {
#pragma unroll
            for (int j = 0; j < 8; ++j) Wih[u][j] = gih[((2 * wave + u) * 8 + j) * 64 + lane];
#pragma unroll
            for (int j = 0; j < 4; ++j) Whh[u][j] = ghh[((2 * wave + u) * 4 + j) * 64 + lane];
            bias[u] = gb[(2 * wave + u) * 64 + lane];
        }
    }

    // ---- h0 staging: a row is 32 chunks of 16 B; thread (row = tid / 32 + 16 k, chunk = tid % 32) moves NSG chunks per step ----
    const int cid = tid & 31;                                            // H0 order: 16 d + 4 q' + c
    const int slot = (cid & 16) + 4 * (cid & 3) + ((cid >> 2) & 3);      // LDS order: 16 d + 4 c + q'
    f32x4 sreg[NSG];
    auto load_h0 = [&](int t) {
#pragma unroll
        for (int k = 0; k < NSG; ++k) {
            const int64_t site = base_site + (tid >> 5) + 16 * k;
            const int64_t sc = site < N ? site : N - 1;
            sreg[k] = *reinterpret_cast<const f32x4*>(H0 + (sc * PW + t) * 128 + 4 * cid);
        }
    };
    auto store_h0 = [&](int buf) {
#pragma unroll
        for (int k = 0; k < NSG; ++k)
            *reinterpret_cast<f32x4*>(h0s + ((size_t)buf * NS + (tid >> 5) + 16 * k) * RS_H0ROW + 4 * slot) = sreg[k];
    };
    // Software pipeline: the input part of step u+1 (64 of a step's 96 MFMAs per site group) depends on nothing step u
    // computes, so it is issued in step u behind the recurrent part - beside the sigmoid / tanh work of the cells and in front
    // of the barrier, which every wave then reaches with its h1 rows long written.  Per accumulator the chain is still
    // bias -> 32 input K-steps -> 16 recurrent K-steps.  h0 rows are staged two steps ahead (buffer u % 2 holds step u).
    auto tpos = [&](int u) { return dir ? PW - 1 - u : u; };
    load_h0(tpos(0)); store_h0(0);
    load_h0(tpos(1)); store_h0(1);
    __syncthreads();
    load_h0(tpos(2));

    f32x4 acc[NSG][2];
    auto input_part = [&](int sg, int buf, f32x4& a0, f32x4& a1) {
        const float* r0 = h0s + ((size_t)buf * NS + 16 * sg + n) * RS_H0ROW + 4 * q;
        a0 = bias[0]; a1 = bias[1];
#pragma unroll
        for (int j = 0; j < 8; ++j) {                                     // K-steps 4j .. 4j+3: direction j / 4, chunk j % 4
            const f32x4 b4 = *reinterpret_cast<const f32x4*>(r0 + 64 * (j >> 2) + 16 * (j & 3));
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                a0 = mfma4(Wih[0][j][e], b4[e], a0);
                a1 = mfma4(Wih[1][j][e], b4[e], a1);
            }
        }
    };
    // Waves w and w + 4 share a SIMD and its matrix pipe and run in lockstep between barriers.  The second half of the
    // workgroup therefore issues a group's next-step input part BEFORE that group's recurrent part and cell, the first half
    // behind it: the sigmoid / tanh work of one wave then sits beside MFMAs of its partner (MI355X_MICROARCH.md, 'Two waves
    // per SIMD', item 9) instead of both idling the pipe together.
    const bool late = STAGGER && wave >= 4;
#pragma unroll
    for (int sg = 0; sg < NSG; ++sg) input_part(sg, 0, acc[sg][0], acc[sg][1]);
    // step 0 stores the rows of step 2 over those of step 0: every wave must have read them (two waves share a SIMD and the
    // older one can be a whole input part ahead; seen as wrong results of the 64-site variant once the cell got shorter)
    lds_barrier();

    float c[NSG][2];
#pragma unroll
    for (int sg = 0; sg < NSG; ++sg) c[sg][0] = c[sg][1] = 0.f;

    for (int s = 0; s < PSTEPS1; ++s) {
        const int cur = s & 1;
        const float* hrb = h1x + (size_t)(cur ^ 1) * NS * RS_XROW;       // h1_{s-1}
        float* hwb = h1x + (size_t)cur * NS * RS_XROW;                   // h1_s
#pragma unroll
        for (int sg = 0; sg < NSG; ++sg) {
            f32x4 nx0, nx1;
            if (late && s + 1 < PSTEPS1) input_part(sg, cur ^ 1, nx0, nx1);
            if (s > 0) {
                const float* r1 = hrb + (size_t)(16 * sg + n) * RS_XROW + 4 * q;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const f32x4 b4 = *reinterpret_cast<const f32x4*>(r1 + 16 * j);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        acc[sg][0] = mfma4(Whh[0][j][e], b4[e], acc[sg][0]);
                        acc[sg][1] = mfma4(Whh[1][j][e], b4[e], acc[sg][1]);
                    }
                }
            }
            // cell: lane (n, q) holds unit 4 (2 wave + u) + q, u = 0, 1 -> exchange positions 16 (w / 2) + 4 q + 2 (w % 2) + u
            float2 w2;
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const float h = lstm_cell(acc[sg][u][0], acc[sg][u][1], acc[sg][u][2], acc[sg][u][3], c[sg][u]);
                if (u == 0) w2.x = h; else w2.y = h;
            }
            *reinterpret_cast<float2*>(hwb + (size_t)(16 * sg + n) * RS_XROW + 16 * (wave >> 1) + 4 * q + 2 * (wave & 1)) = w2;
            if (s + 1 < PSTEPS1) {                                      // step s+1's input part (reads h0 of step s+1)
                if (late) { acc[sg][0] = nx0; acc[sg][1] = nx1; }
                else input_part(sg, cur ^ 1, acc[sg][0], acc[sg][1]);
            }
        }
        if (s + 2 < PSTEPS1) {
            store_h0(cur);                                               // h0 of step s+2 replaces h0 of step s (last read in step s-1)
            if (s + 3 < PSTEPS1) load_h0(tpos(s + 3));
        }
        lds_barrier();
    }
    // h1 at position 16 -> H1c[site][dir][q][16]: entries 2 wave, 2 wave + 1 of row q (read back from the exchange rows)
    const float* hfin = h1x + (size_t)((PSTEPS1 - 1) & 1) * NS * RS_XROW;
#pragma unroll
    for (int sg = 0; sg < NSG; ++sg) {
        const int64_t site = base_site + 16 * sg + n;
        if (site < N)
            *reinterpret_cast<float2*>(H1c + site * 128 + dir * 64 + q * 16 + 2 * wave) =
                *reinterpret_cast<const float2*>(hfin + (size_t)(16 * sg + n) * RS_XROW + 16 * (wave >> 1) + 4 * q + 2 * (wave & 1));
    }
}

// K23r4: the same fused layer 1 with FOUR gate tiles per wave and four waves per workgroup (default).  K23r's eight waves sit
// two to a SIMD and move in lockstep between the step barriers, so the matrix pipe idles whenever both are in their cells,
// at the barrier or waiting for fragments (82 % MFMA busy measured).  With W_ih1 + W_hh1 of four tiles in 192 registers
// (gfx950 gives a wave up to 512 unified VGPR/AGPR; two such waves fit a SIMD) a workgroup is one wave per SIMD, the second
// wave of a SIMD belongs to ANOTHER workgroup with its own barriers, and every B fragment read from LDS feeds four tiles
// instead of two.  Same chains per accumulator: bit-identical to K2 + K3.
constexpr int rs4_l1_lds_bytes(int nsg) { return (2 * 16 * nsg * RS_H0ROW + 2 * 16 * nsg * RS_XROW + 256) * 4; }

template <int NSG>
__global__ __launch_bounds__(256, 2) void k_pileup_l1_rs4(
    const float* __restrict__ H0, int64_t N,
    const float* __restrict__ wih0, const float* __restrict__ wih1,
    const float* __restrict__ whh0, const float* __restrict__ whh1,
    const float* __restrict__ bias0, const float* __restrict__ bias1,
    float* __restrict__ H1c)
{
    extern __shared__ f32x4 lds[];
    constexpr int NS = 16 * NSG;
    float* const h0s = reinterpret_cast<float*>(lds);                  // [2][NS][RS_H0ROW]
    float* const h1x = h0s + 2 * NS * RS_H0ROW;                         // [2][NS][RS_XROW]
    f32x4* const lbias = reinterpret_cast<f32x4*>(h1x + 2 * NS * RS_XROW);   // [16 tiles][4 q]
    const int dir = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 15, q = lane >> 4;
    NSNP_DEVCLK_START
    const int64_t base_site = (int64_t)blockIdx.x * NS;

    f32x4 Wih[4][8], Whh[4][4];
    {
        const f32x4* __restrict__ gih = reinterpret_cast<const f32x4*>(dir ? wih1 : wih0);      // [tile][j4 8][lane]
        const f32x4* __restrict__ ghh = reinterpret_cast<const f32x4*>(dir ? whh1 : whh0);      // [tile][j4 4][lane]
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int j = 0; j < 8; ++j) Wih[u][j] = gih[((4 * wave + u) * 8 + j) * 64 + lane];
#pragma unroll
            for (int j = 0; j < 4; ++j) Whh[u][j] = ghh[((4 * wave + u) * 4 + j) * 64 + lane];
        }
        if (tid < 64) lbias[tid] = reinterpret_cast<const f32x4*>(dir ? bias1 : bias0)[(tid >> 2) * 64 + 16 * (tid & 3)];
    }

    // ---- h0 staging: a row is 32 chunks of 16 B; thread (row = tid / 32 + 8 k, chunk = tid % 32) moves 2 NSG chunks per step ----
    const int cid = tid & 31;                                            // H0 order: 16 d + 4 q' + c
    const int slot = (cid & 16) + 4 * (cid & 3) + ((cid >> 2) & 3);      // LDS order: 16 d + 4 c + q'
    f32x4 sreg[2 * NSG];
    auto load_h0 = [&](int t) {
#pragma unroll
        for (int k = 0; k < 2 * NSG; ++k) {
            const int64_t site = base_site + (tid >> 5) + 8 * k;
            const int64_t sc = site < N ? site : N - 1;
            sreg[k] = *reinterpret_cast<const f32x4*>(H0 + (sc * PW + t) * 128 + 4 * cid);
        }
    };
    auto store_h0 = [&](int buf) {
#pragma unroll
        for (int k = 0; k < 2 * NSG; ++k)
            *reinterpret_cast<f32x4*>(h0s + ((size_t)buf * NS + (tid >> 5) + 8 * k) * RS_H0ROW + 4 * slot) = sreg[k];
    };
    load_h0(dir ? PW - 1 : 0);
    store_h0(0);
    if (PSTEPS1 > 1) load_h0(dir ? PW - 2 : 1);
    __syncthreads();

    float c[NSG][4];
#pragma unroll
    for (int sg = 0; sg < NSG; ++sg)
#pragma unroll
        for (int u = 0; u < 4; ++u) c[sg][u] = 0.f;

    // What a wave does between the step barrier and its first MFMA is on the critical path of its SIMD (its partner wave belongs to
    // another workgroup but runs the same program), so only the fragment reads stand there; the h0 staging runs two steps ahead
    // from INSIDE the burst of 192 MFMAs: the rows of step s+1 (loaded during step s-1) go to LDS behind K block 1, the loads of
    // step s+2 go out behind K block 4 into the same registers.
    for (int s = 0; s < PSTEPS1; ++s) {
        const int t = dir ? PW - 1 - s : s;
        const int cur = s & 1;
        const float* h0b = h0s + (size_t)cur * NS * RS_H0ROW;
        const float* hrb = h1x + (size_t)(cur ^ 1) * NS * RS_XROW;       // h1_{s-1}
        float* hwb = h1x + (size_t)cur * NS * RS_XROW;                   // h1_s
#pragma unroll
        for (int sg = 0; sg < NSG; ++sg) {
            const float* r0 = h0b + (size_t)(16 * sg + n) * RS_H0ROW + 4 * q;
            f32x4 acc[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) acc[u] = lbias[(4 * wave + u) * 4 + q];
#pragma unroll
            for (int j = 0; j < 8; ++j) {                                 // K-steps 4j .. 4j+3: direction j / 4, chunk j % 4
                if (sg == 0 && j == 2 && s + 1 < PSTEPS1) {
                    __builtin_amdgcn_sched_barrier(0);
                    store_h0(cur ^ 1);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (sg == 0 && j == 5 && s + 2 < PSTEPS1) {
                    __builtin_amdgcn_sched_barrier(0);
                    load_h0(dir ? t - 2 : t + 2);
                    __builtin_amdgcn_sched_barrier(0);
                }
                const f32x4 b4 = *reinterpret_cast<const f32x4*>(r0 + 64 * (j >> 2) + 16 * (j & 3));
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int u = 0; u < 4; ++u) acc[u] = mfma4(Wih[u][j][e], b4[e], acc[u]);
            }
            if (s > 0) {
                const float* r1 = hrb + (size_t)(16 * sg + n) * RS_XROW + 4 * q;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const f32x4 b4 = *reinterpret_cast<const f32x4*>(r1 + 16 * j);
#pragma unroll
                    for (int e = 0; e < 4; ++e)
#pragma unroll
                        for (int u = 0; u < 4; ++u) acc[u] = mfma4(Whh[u][j][e], b4[e], acc[u]);
                }
            }
            // cell: lane (n, q) holds unit 4 (4 wave + u) + q -> exchange positions 16 wave + 4 q + u
            f32x4 hn;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                hn[u] = lstm_cell(acc[u][0], acc[u][1], acc[u][2], acc[u][3], c[sg][u]);
            }
            *reinterpret_cast<f32x4*>(hwb + (size_t)(16 * sg + n) * RS_XROW + 16 * wave + 4 * q) = hn;
        }
        lds_barrier();
    }
    // h1 at position 16 -> H1c[site][dir][q][16]: entries 4 wave .. 4 wave + 3 of row q (read back from the exchange rows)
    const float* hfin = h1x + (size_t)((PSTEPS1 - 1) & 1) * NS * RS_XROW;
#pragma unroll
    for (int sg = 0; sg < NSG; ++sg) {
        const int64_t site = base_site + 16 * sg + n;
        if (site < N)
            *reinterpret_cast<f32x4*>(H1c + site * 128 + dir * 64 + q * 16 + 4 * wave) =
                *reinterpret_cast<const f32x4*>(hfin + (size_t)(16 * sg + n) * RS_XROW + 16 * wave + 4 * q);
    }
    NSNP_DEVCLK_STOP(1)
}

// ---------------------------------------------------------------------------------------------
// K4: heads.  block = 256 (4 waves x 16 sites), weights streamed from L2 (216 KB of images).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_pileup_head(
    const float* __restrict__ H1c, int64_t N,
    const float* __restrict__ proj_w, const float* __restrict__ proj_b,
    const float* __restrict__ dense_w, const float* __restrict__ dense_b,
    const float* __restrict__ head_w, const float* __restrict__ head_b,
    float* __restrict__ gt_prob, float* __restrict__ zy_prob)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q = lane >> 4;
    const int64_t site = ((int64_t)blockIdx.x * 4 + wave) * 16 + (lane & 15);
    const bool live = site < N;
    const int64_t sc = live ? site : N - 1;
    const f32x4* __restrict__ hin = reinterpret_cast<const f32x4*>(H1c + sc * 128 + q * 16);
    float b[32];
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        const f32x4 a = hin[v], bb = hin[16 + v];
#pragma unroll
        for (int e = 0; e < 4; ++e) { b[4 * v + e] = a[e]; b[16 + 4 * v + e] = bb[e]; }
    }
    // output_proj 128 -> 128 (model.py:37)
    f32x4 ap[8];
    {
        const f32x4* pb = reinterpret_cast<const f32x4*>(proj_b);
#pragma unroll
        for (int i = 0; i < 8; ++i) ap[i] = pb[i * 64 + lane];
        wave_gemm<8, 8, 0, 8>(reinterpret_cast<const f32x4*>(proj_w), lane, b, ap);
    }
    // dense 128 -> 256 + tanh (model.py:67)
    float b2[32];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int g = 0; g < 4; ++g) b2[4 * i + g] = ap[i][g];
    f32x4 ad[16];
    {
        const f32x4* db = reinterpret_cast<const f32x4*>(dense_b);
#pragma unroll
        for (int i = 0; i < 16; ++i) ad[i] = db[i * 64 + lane];
        wave_gemm<16, 8, 0, 8>(reinterpret_cast<const f32x4*>(dense_w), lane, b2, ad);
    }
    float b3[64];
#pragma unroll
    for (int i = 0; i < 16; ++i)
#pragma unroll
        for (int g = 0; g < 4; ++g) b3[4 * i + g] = tanh_f(ad[i][g]);
    // genotype (rows 0..20) and zygosity (rows 21..23) heads (model.py:69-70)
    f32x4 ah[2];
    {
        const f32x4* hb = reinterpret_cast<const f32x4*>(head_b);
        ah[0] = hb[lane]; ah[1] = hb[64 + lane];
        wave_gemm2x16(reinterpret_cast<const f32x4*>(head_w), lane, b3, ah);
    }
    // lane (site, q) holds rows 4q..4q+3 of tile 0 and rows 16+4q.. of tile 1.
    // genotype rows: tile0 all 16; tile1 rows 16..19 (q=0, g=0..3) and row 20 (q=1, g=0)
    // zygosity rows 21..23: tile1, q=1, g=1..3
    const float NEG = -3.0e38f;
    float g0[4], g1[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        g0[g] = ah[0][g];
        const bool is_gt = (q == 0) || (q == 1 && g == 0);
        g1[g] = is_gt ? ah[1][g] : NEG;
    }
    float mx = fmaxf(fmaxf(fmaxf(g0[0], g0[1]), fmaxf(g0[2], g0[3])), fmaxf(fmaxf(g1[0], g1[1]), fmaxf(g1[2], g1[3])));
    mx = fmaxf(mx, __shfl_xor(mx, 16));
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    float e0[4], e1[4], sum = 0.f;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        e0[g] = __expf(g0[g] - mx);
        e1[g] = g1[g] > -1.0e38f ? __expf(g1[g] - mx) : 0.f;
        sum += e0[g] + e1[g];
    }
    sum += __shfl_xor(sum, 16);
    sum += __shfl_xor(sum, 32);
    // zygosity softmax is local to lane q == 1
    const float z1 = ah[1][1], z2 = ah[1][2], z3 = ah[1][3];
    const float zm = fmaxf(z1, fmaxf(z2, z3));
    const float ez1 = __expf(z1 - zm), ez2 = __expf(z2 - zm), ez3 = __expf(z3 - zm);
    const float zs = ez1 + ez2 + ez3;
    if (live) {
        float* gp = gt_prob + site * NSNP_GT_CLASSES;
#pragma unroll
        for (int g = 0; g < 4; ++g) gp[4 * q + g] = e0[g] / sum;
        if (q == 0) {
#pragma unroll
            for (int g = 0; g < 4; ++g) gp[16 + g] = e1[g] / sum;
        }
        if (q == 1) {
            gp[20] = e1[0] / sum;
            float* zp = zy_prob + site * NSNP_ZY_CLASSES;
            zp[0] = ez1 / zs; zp[1] = ez2 / zs; zp[2] = ez3 / zs;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// K4r: heads with the output tiles split over the 8 waves of a workgroup (default).  K4 above walks 896 dependent-latency
// bound MFMAs per 16 sites in ONE wave with its weights streamed from L2 (36 us for a 4096-site batch whose matrix work is
// 3 us).  Here a workgroup still owns 16 sites at a time, but wave w computes output_proj tile w, dense tiles 2w, 2w+1 and
// (waves 0, 1) one head tile each, with exactly those weight fragments held in VGPRs across a persistent loop over site
// groups; the 128 / 256 intermediate features cross waves through LDS in the order the next product's B operand wants them
// (K-step j of lane (n, q) = accumulator register j % 4 of tile j / 4 of the same lane).  Chains per accumulator are those
// of K4, so the probabilities are bit-identical.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512, 2) void k_pileup_head_rs(
    const float* __restrict__ H1c, int64_t N,
    const float* __restrict__ proj_w, const float* __restrict__ proj_b,
    const float* __restrict__ dense_w, const float* __restrict__ dense_b,
    const float* __restrict__ head_w, const float* __restrict__ head_b,
    float* __restrict__ gt_prob, float* __restrict__ zy_prob,
    uint8_t* __restrict__ gt_arg, uint8_t* __restrict__ zy_arg, float* __restrict__ gt_max, float* __restrict__ zy_max)
{
    __shared__ float x1[32][64];          // output_proj results as dense K-steps
    __shared__ float x2[64][64];          // tanh(dense) as head K-steps
    __shared__ f32x4 x3[64];              // head tile 1 (wave 1) for the softmax in wave 0
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q = lane >> 4;
    // weights of this wave
    f32x4 Wp[8], Wd[2][8], Wh[16], bp, bd[2], bh;
    {
        const f32x4* pw = reinterpret_cast<const f32x4*>(proj_w);      // [8 tiles][8 j4][lane]
        const f32x4* dw = reinterpret_cast<const f32x4*>(dense_w);     // [16][8][lane]
        const f32x4* hw = reinterpret_cast<const f32x4*>(head_w);      // [2][16][lane]
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            Wp[j] = pw[(wave * 8 + j) * 64 + lane];
            Wd[0][j] = dw[((2 * wave) * 8 + j) * 64 + lane];
            Wd[1][j] = dw[((2 * wave + 1) * 8 + j) * 64 + lane];
        }
        bp = reinterpret_cast<const f32x4*>(proj_b)[wave * 64 + lane];
        bd[0] = reinterpret_cast<const f32x4*>(dense_b)[(2 * wave) * 64 + lane];
        bd[1] = reinterpret_cast<const f32x4*>(dense_b)[(2 * wave + 1) * 64 + lane];
        if (wave < 2) {
#pragma unroll
            for (int j = 0; j < 16; ++j) Wh[j] = hw[(wave * 16 + j) * 64 + lane];
            bh = reinterpret_cast<const f32x4*>(head_b)[wave * 64 + lane];
        }
    }
    const int64_t n_groups = NSNP_CDIV(N, 16);
    for (int64_t grp = blockIdx.x; grp < n_groups; grp += gridDim.x) {
        const int64_t site = grp * 16 + (lane & 15);
        const bool live = site < N;
        const int64_t sc = live ? site : N - 1;
        const f32x4* __restrict__ hin = reinterpret_cast<const f32x4*>(H1c + sc * 128 + q * 16);
        // output_proj 128 -> 128 (model.py:37): tile `wave`
        f32x4 ap = bp;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const f32x4 b4 = hin[(j >> 2) * 16 + (j & 3)];            // fwd half, then reverse half (+64 floats)
#pragma unroll
            for (int e = 0; e < 4; ++e) ap = mfma4(Wp[j][e], b4[e], ap);
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) x1[4 * wave + g][lane] = ap[g];
        __syncthreads();
        // dense 128 -> 256 + tanh (model.py:67): tiles 2 wave, 2 wave + 1
        f32x4 ad[2] = {bd[0], bd[1]};
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float b = x1[4 * j + e][lane];
                ad[0] = mfma4(Wd[0][j][e], b, ad[0]);
                ad[1] = mfma4(Wd[1][j][e], b, ad[1]);
            }
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int g = 0; g < 4; ++g) x2[4 * (2 * wave + u) + g][lane] = tanh_f(ad[u][g]);
        __syncthreads();
        // genotype (rows 0..20) and zygosity (rows 21..23) heads (model.py:69-70): tile 0 in wave 0, tile 1 in wave 1
        f32x4 ah = bh;
        if (wave < 2) {
#pragma unroll
            for (int j = 0; j < 16; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) ah = mfma4(Wh[j][e], x2[4 * j + e][lane], ah);
            if (wave == 1) x3[lane] = ah;
        }
        __syncthreads();
        if (wave == 0) {
            const f32x4 a1 = x3[lane];
            // lane (site, q) holds rows 4q..4q+3 of tile 0 and rows 16+4q.. of tile 1.
            // genotype rows: tile0 all 16; tile1 rows 16..19 (q=0, g=0..3) and row 20 (q=1, g=0); zygosity rows 21..23: tile1, q=1, g=1..3
            const float NEG = -3.0e38f;
            float g0[4], g1[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                g0[g] = ah[g];
                const bool is_gt = (q == 0) || (q == 1 && g == 0);
                g1[g] = is_gt ? a1[g] : NEG;
            }
            float mx = fmaxf(fmaxf(fmaxf(g0[0], g0[1]), fmaxf(g0[2], g0[3])), fmaxf(fmaxf(g1[0], g1[1]), fmaxf(g1[2], g1[3])));
            mx = fmaxf(mx, __shfl_xor(mx, 16));
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            float e0[4], e1[4], sum = 0.f;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                e0[g] = __expf(g0[g] - mx);
                e1[g] = g1[g] > -1.0e38f ? __expf(g1[g] - mx) : 0.f;
                sum += e0[g] + e1[g];
            }
            sum += __shfl_xor(sum, 16);
            sum += __shfl_xor(sum, 32);
            const float z1 = a1[1], z2 = a1[2], z3 = a1[3];
            const float zm = fmaxf(z1, fmaxf(z2, z3));
            const float ez1 = __expf(z1 - zm), ez2 = __expf(z2 - zm), ez3 = __expf(z3 - zm);
            const float zs = ez1 + ez2 + ez3;
            float p0[4], p1[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) { p0[g] = e0[g] / sum; p1[g] = e1[g] / sum; }
            const float pz1 = ez1 / zs, pz2 = ez2 / zs, pz3 = ez3 / zs;
            if (live) {
                float* gp = gt_prob + site * NSNP_GT_CLASSES;
#pragma unroll
                for (int g = 0; g < 4; ++g) gp[4 * q + g] = p0[g];
                if (q == 0) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) gp[16 + g] = p1[g];
                }
                if (q == 1) {
                    gp[20] = p1[0];
                    float* zp = zy_prob + site * NSNP_ZY_CLASSES;
                    zp[0] = pz1; zp[1] = pz2; zp[2] = pz3;
                }
            }
            if (gt_arg) {
                // predict.py:54-57 in the same launch: np.argmax / np.max over the stored probabilities (first maximum wins).
                // Lane (site, q) holds classes 4q .. 4q+3 (p0), q == 0 also 16..19, q == 1 also 20; the four q lanes of a site are
                // 16 lanes apart.
                float bv = p0[0]; int bi = 4 * q;
#pragma unroll
                for (int g = 1; g < 4; ++g) if (p0[g] > bv) { bv = p0[g]; bi = 4 * q + g; }
                if (q == 0) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) if (p1[g] > bv) { bv = p1[g]; bi = 16 + g; }
                }
                if (q == 1 && p1[0] > bv) { bv = p1[0]; bi = 20; }
#pragma unroll
                for (int o = 16; o <= 32; o <<= 1) {
                    const float ov = __shfl_xor(bv, o); const int oi = __shfl_xor(bi, o);
                    if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
                }
                if (live && q == 0) { gt_arg[site] = (uint8_t)bi; gt_max[site] = bv; }
                if (live && q == 1) {
                    float zb = pz1; int zi = 0;
                    if (pz2 > zb) { zb = pz2; zi = 1; }
                    if (pz3 > zb) { zb = pz3; zi = 2; }
                    zy_arg[site] = (uint8_t)zi; zy_max[site] = zb;
                }
            }
        }
        // (x1 / x2 / x3 of this group are not touched again before the next group's first barrier)
    }
}

// ---- postprocess: predict.py:54-65 --------------------------------------------------------------
__global__ void k_pileup_post(const float* __restrict__ gt, const float* __restrict__ zy,
                              const int32_t* __restrict__ x, int64_t N, uint8_t* gt_arg, uint8_t* zy_arg,
                              float* gt_max, float* zy_max, int32_t* depth)
{
    const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const float* g = gt + n * NSNP_GT_CLASSES;
    int bi = 0; float bv = g[0];
    for (int i = 1; i < NSNP_GT_CLASSES; ++i) if (g[i] > bv) { bv = g[i]; bi = i; }   // first max, as np.argmax
    gt_arg[n] = (uint8_t)bi; gt_max[n] = bv;
    const float* z = zy + n * NSNP_ZY_CLASSES;
    bi = 0; bv = z[0];
    for (int i = 1; i < NSNP_ZY_CLASSES; ++i) if (z[i] > bv) { bv = z[i]; bi = i; }
    zy_arg[n] = (uint8_t)bi; zy_max[n] = bv;
    if (depth) {
        const int32_t* c = x + n * (PW * PC) + PCENTER * PC;
        const int idx[8] = {0, 1, 2, 3, 9, 10, 11, 12};
        int d = 0;
        for (int k = 0; k < 8; ++k) { const int v = c[idx[k]]; if (v < 0) d -= v; }
        depth[n] = d;
    }
}

}  // namespace

// ---------------------------------------------------------------------------------------------
// host side: weight packing
// ---------------------------------------------------------------------------------------------
namespace {

inline int gate_row(int row)   // accumulator row -> torch gate-major row (H = 64)
{
    const int i = row >> 4, r = row & 15, qp = r >> 2, g = r & 3;
    return g * PH + 4 * i + qp;
}
struct MatRef { const float* w; int ld; const float* b0; const float* b1; };

float f_whh(const void* u, int row, int k) { const MatRef* m = (const MatRef*)u; return m->w[gate_row(row) * m->ld + k]; }
float f_wih0(const void* u, int row, int k) { const MatRef* m = (const MatRef*)u; return m->w[gate_row(row) * m->ld + k]; }
// K index in H0 / H1c storage order: k = 4*j + q, j = dir*16 + i  ->  feature dir*64 + 4*i + q
inline int h_store_col(int k) { const int j = k >> 2, q = k & 3; return (j >> 4) * 64 + 4 * (j & 15) + q; }
float f_wih1(const void* u, int row, int k) { const MatRef* m = (const MatRef*)u; return m->w[gate_row(row) * m->ld + h_store_col(k)]; }
float f_proj(const void* u, int row, int k) { const MatRef* m = (const MatRef*)u; return m->w[row * m->ld + h_store_col(k)]; }
// K index in accumulator order of the previous layer: k = 4*j + q, j = 4*i + g -> feature 16*i + 4*q + g
inline int acc_col(int k) { const int j = k >> 2, q = k & 3; return 16 * (j >> 2) + 4 * q + (j & 3); }
float f_dense(const void* u, int row, int k) { const MatRef* m = (const MatRef*)u; return m->w[row * m->ld + acc_col(k)]; }
struct HeadRef { const float* gt; const float* zy; };
float f_head(const void* u, int row, int k)
{
    const HeadRef* m = (const HeadRef*)u;
    if (row < 21) return m->gt[row * 256 + acc_col(k)];
    if (row < 24) return m->zy[(row - 21) * 256 + acc_col(k)];
    return 0.f;
}

}  // namespace

void nsnp_pack_image(float* img, int n_tiles, int n_j4, nsnp_wfun f, const void* user)
{
    for (int tile = 0; tile < n_tiles; ++tile)
        for (int j4 = 0; j4 < n_j4; ++j4)
            for (int lane = 0; lane < 64; ++lane)
                for (int e = 0; e < 4; ++e) {
                    const int row = 16 * tile + (lane & 15);
                    const int k = 4 * (4 * j4 + e) + (lane >> 4);
                    img[(((size_t)tile * n_j4 + j4) * 64 + lane) * 4 + e] = f(user, row, k);
                }
}

int nsnp_pileup_pack_weights(nsnp_ctx* ctx, const float* const* w)
{
    PileupWeightsDev& pw = ctx->pw;
    // arena layout (floats)
    const size_t n_whh = 16 * 4 * 256, n_wih0 = 16 * 1 * 256, n_wlast = 16 * 64, n_wih1 = 16 * 8 * 256,
                 n_b1 = 16 * 256, n_proj = 8 * 8 * 256, n_pb = 8 * 256, n_dense = 16 * 8 * 256, n_db = 16 * 256,
                 n_head = 2 * 16 * 256, n_hb = 2 * 256;
    const size_t total = 2 * (n_whh + n_wih0 + n_wlast + n_wih1 + 2 * n_b1 + n_whh) + n_proj + n_pb + n_dense + n_db + n_head + n_hb;
    std::vector<float> host(total);
    size_t off = 0;
    auto take = [&](size_t n) { float* p = host.data() + off; off += n; return p; };
    float* h_l0_whh[2]; float* h_l0_wih[2]; float* h_l0_wlast[2]; float* h_l1_wih[2]; float* h_l1_b[2]; float* h_l1_braw[2]; float* h_l1_whh[2];
    for (int d = 0; d < 2; ++d) {
        h_l0_whh[d] = take(n_whh); h_l0_wih[d] = take(n_wih0); h_l0_wlast[d] = take(n_wlast);
        h_l1_wih[d] = take(n_wih1); h_l1_b[d] = take(n_b1); h_l1_braw[d] = take(n_b1); h_l1_whh[d] = take(n_whh);
    }
    float* h_proj = take(n_proj); float* h_pb = take(n_pb); float* h_dense = take(n_dense); float* h_db = take(n_db);
    float* h_head = take(n_head); float* h_hb = take(n_hb);

    // The cell wants exp2(-log2 e z) for the gates i, f, o and exp2(-2 log2 e z) for g: the factor is folded into the gate's rows
    // of W_ih, W_hh and the biases here (PyTorch row order i f g o), so that lstm_cell() starts with the exponentials
    // (4 of its ~26 vector instructions per unit-step; fp32 MFMA and the vector ALU share the SIMD's lanes, section 4 of DESIGN.md)
    std::vector<float> scaled[16];
    const float* ws[16];
    for (int t = 0; t < 16; ++t) {
        const int layer = t / 8, kind = t % 4;      // w_ih, w_hh, b_ih, b_hh
        const size_t cols = kind == 0 ? (layer ? 2 * PH : PC) : (kind == 1 ? PH : 1);
        scaled[t].resize((size_t)4 * PH * cols);
        for (int r = 0; r < 4 * PH; ++r) {
            const float f = (r / PH == 2) ? -2.0f * LOG2E : -LOG2E;
            for (size_t k = 0; k < cols; ++k) scaled[t][r * cols + k] = f * w[t][r * cols + k];
        }
        ws[t] = scaled[t].data();
    }
    for (int d = 0; d < 2; ++d) {
        const float* const* l0 = ws + d * 4;        // w_ih, w_hh, b_ih, b_hh
        const float* const* l1 = ws + 8 + d * 4;
        MatRef m;
        m = MatRef{l0[1], PH, nullptr, nullptr};  nsnp_pack_image(h_l0_whh[d], 16, 4, f_whh, &m);
        m = MatRef{l0[0], PC, nullptr, nullptr};  nsnp_pack_image(h_l0_wih[d], 16, 1, f_wih0, &m);
        for (int tile = 0; tile < 16; ++tile)
            for (int lane = 0; lane < 64; ++lane) {
                const int tr = gate_row(16 * tile + (lane & 15)), kq = lane >> 4;
                float v = 0.f;
                if (kq < 2) v = l0[0][tr * PC + 16 + kq];
                else if (kq == 2) v = l0[2][tr] + l0[3][tr];   // b_ih + b_hh rides on a constant-1 input
                h_l0_wlast[d][tile * 64 + lane] = v;
            }
        m = MatRef{l1[0], 2 * PH, nullptr, nullptr};  nsnp_pack_image(h_l1_wih[d], 16, 8, f_wih1, &m);
        for (int tile = 0; tile < 16; ++tile)
            for (int lane = 0; lane < 64; ++lane)
                for (int g = 0; g < 4; ++g) {
                    const int tr = gate_row(16 * tile + 4 * (lane >> 4) + g);
                    h_l1_b[d][(tile * 64 + lane) * 4 + g] = l1[2][tr] + l1[3][tr];
                    h_l1_braw[d][(tile * 64 + lane) * 4 + g] = w[8 + d * 4 + 2][tr] + w[8 + d * 4 + 3][tr];     // unscaled: the f16x3 two-kernel path
                }
        m = MatRef{l1[1], PH, nullptr, nullptr};  nsnp_pack_image(h_l1_whh[d], 16, 4, f_whh, &m);
    }
    {
        MatRef m{w[16], 128, nullptr, nullptr};  nsnp_pack_image(h_proj, 8, 8, f_proj, &m);
        for (int tile = 0; tile < 8; ++tile) for (int lane = 0; lane < 64; ++lane) for (int g = 0; g < 4; ++g)
            h_pb[(tile * 64 + lane) * 4 + g] = w[17][16 * tile + 4 * (lane >> 4) + g];
        MatRef md{w[18], 128, nullptr, nullptr}; nsnp_pack_image(h_dense, 16, 8, f_dense, &md);
        for (int tile = 0; tile < 16; ++tile) for (int lane = 0; lane < 64; ++lane) for (int g = 0; g < 4; ++g)
            h_db[(tile * 64 + lane) * 4 + g] = w[19][16 * tile + 4 * (lane >> 4) + g];
        HeadRef hr{w[20], w[22]};  nsnp_pack_image(h_head, 2, 16, f_head, &hr);
        for (int tile = 0; tile < 2; ++tile) for (int lane = 0; lane < 64; ++lane) for (int g = 0; g < 4; ++g) {
            const int row = 16 * tile + 4 * (lane >> 4) + g;
            h_hb[(tile * 64 + lane) * 4 + g] = row < 21 ? w[21][row] : (row < 24 ? w[23][row - 21] : 0.f);
        }
    }
    if (pw.arena && pw.arena_bytes != total * sizeof(float)) { (void)hipFree(pw.arena); pw.arena = nullptr; }
    if (!pw.arena) {
        NSNP_HIP(ctx, hipMalloc((void**)&pw.arena, total * sizeof(float)));
        pw.arena_bytes = total * sizeof(float);
    }
    NSNP_HIP(ctx, hipMemcpy(pw.arena, host.data(), total * sizeof(float), hipMemcpyHostToDevice));
    auto dev = [&](const float* hp) { return pw.arena + (hp - host.data()); };
    for (int d = 0; d < 2; ++d) {
        pw.l0_whh[d] = dev(h_l0_whh[d]); pw.l0_wih[d] = dev(h_l0_wih[d]); pw.l0_wlast[d] = dev(h_l0_wlast[d]);
        pw.l1_wih[d] = dev(h_l1_wih[d]); pw.l1_bias[d] = dev(h_l1_b[d]); pw.l1_bias_raw[d] = dev(h_l1_braw[d]); pw.l1_whh[d] = dev(h_l1_whh[d]);
    }
    pw.proj_w = dev(h_proj); pw.proj_b = dev(h_pb); pw.dense_w = dev(h_dense); pw.dense_b = dev(h_db);
    pw.head_w = dev(h_head); pw.head_b = dev(h_hb);
    pw.loaded = true;
    return NSNP_OK;
}

// ---------------------------------------------------------------------------------------------
// launcher
// ---------------------------------------------------------------------------------------------
static int set_lds_attr_once(nsnp_ctx* ctx)
{
    if (ctx->attr_set) return NSNP_OK;
    NSNP_HIP(ctx, hipFuncSetAttribute((const void*)k_pileup_l0<8>, hipFuncAttributeMaxDynamicSharedMemorySize, L0_LDS_BYTES));
    NSNP_HIP(ctx, hipFuncSetAttribute((const void*)k_pileup_l0<4>, hipFuncAttributeMaxDynamicSharedMemorySize, L0_LDS_BYTES));
    NSNP_HIP(ctx, hipFuncSetAttribute((const void*)k_pileup_l0<2>, hipFuncAttributeMaxDynamicSharedMemorySize, L0_LDS_BYTES));
    NSNP_HIP(ctx, hipFuncSetAttribute((const void*)k_pileup_l0<1>, hipFuncAttributeMaxDynamicSharedMemorySize, L0_LDS_BYTES));
    NSNP_HIP(ctx, hipFuncSetAttribute((const void*)k_pileup_proj1, hipFuncAttributeMaxDynamicSharedMemorySize, P1_LDS_BYTES));
    NSNP_HIP(ctx, hipFuncSetAttribute((const void*)k_pileup_l1<8>, hipFuncAttributeMaxDynamicSharedMemorySize, L1_LDS_BYTES));
    NSNP_HIP(ctx, hipFuncSetAttribute((const void*)k_pileup_l1<4>, hipFuncAttributeMaxDynamicSharedMemorySize, L1_LDS_BYTES));
    NSNP_HIP(ctx, hipFuncSetAttribute((const void*)k_pileup_l1<2>, hipFuncAttributeMaxDynamicSharedMemorySize, L1_LDS_BYTES));
    NSNP_HIP(ctx, hipFuncSetAttribute((const void*)k_pileup_l1<1>, hipFuncAttributeMaxDynamicSharedMemorySize, L1_LDS_BYTES));
    NSNP_HIP(ctx, hipFuncSetAttribute((const void*)k_pileup_l1_rs4<1>, hipFuncAttributeMaxDynamicSharedMemorySize, rs4_l1_lds_bytes(1)));
    NSNP_HIP(ctx, hipFuncSetAttribute((const void*)k_pileup_l1_rs4<2>, hipFuncAttributeMaxDynamicSharedMemorySize, rs4_l1_lds_bytes(2)));
    NSNP_HIP(ctx, hipFuncSetAttribute((const void*)k_pileup_l1_rs32<4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, rs32_l1_lds_bytes(4)));
    NSNP_HIP(ctx, hipFuncSetAttribute((const void*)k_pileup_l1_rs32<2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, rs32_l1_lds_bytes(2)));
    NSNP_HIP(ctx, hipFuncSetAttribute((const void*)k_pileup_l1_rs32<4, false>, hipFuncAttributeMaxDynamicSharedMemorySize, rs32_l1_lds_bytes(4)));
    NSNP_HIP(ctx, hipFuncSetAttribute((const void*)k_pileup_l1_rs32<2, false>, hipFuncAttributeMaxDynamicSharedMemorySize, rs32_l1_lds_bytes(2)));
    ctx->attr_set = true;
    return NSNP_OK;
}

int nsnp_pileup_forward_impl(nsnp_ctx* ctx, const int32_t* x, const int64_t* center_idx,
                             int64_t N, float* gt, float* zy, const PostOut* post, bool* post_written, hipStream_t s)
{
    if (!ctx->pw.loaded) return NSNP_ENOWEIGHTS;
    if (N == 0) return NSNP_OK;
    int rc = set_lds_attr_once(ctx);
    if (rc) return rc;
    if (!ctx->ws_h0) { rc = nsnp_ctx_reserve(ctx, ctx->chunk_sites); if (rc) return rc; }
    const PileupWeightsDev& pw = ctx->pw;
    for (int64_t base = 0; base < N; base += ctx->chunk_sites) {
        const int64_t n = (N - base < ctx->chunk_sites) ? N - base : ctx->chunk_sites;
        const int32_t* xc = center_idx ? x : x + base * (PW * PC);
        const int64_t* cc = center_idx ? center_idx + base : nullptr;
        // waves per recurrence workgroup: the largest of 8/4/2/1 that still yields a workgroup for every
        // second CU (LDS admits two workgroups per CU whatever their size): a 4096-site batch runs as
        // 128 four-wave workgroups, which measured best when several batches are in flight on
        // different streams (single-stream latency would prefer 1-wave workgroups: see DESIGN.md)
        const int64_t waves_total = NSNP_CDIV(n, 16) * 2;
        int wpb = 8;
        while (wpb > 1 && waves_total / wpb < (int64_t)ctx->n_cu / 2) wpb >>= 1;
        if (ctx->force_wpb) wpb = ctx->force_wpb;
        const dim3 g_rec((unsigned)NSNP_CDIV(n, 16 * wpb), 2);
        if (ctx->l0_rs) {
            ScopedKernelTimer tm(ctx, NSNP_K_L0, s);
            // 16 sites per workgroup (three independent workgroups per SIMD set) measured best at every batch size; 32 / 64
            // sites (software-pipelined over the site groups) stay available as options
            int nsg = 1;
            if (ctx->l0_rs_groups) nsg = ctx->l0_rs_groups;
#define LAUNCH_RS(G) hipLaunchKernelGGL(k_pileup_l0_rs32<G>, dim3((unsigned)NSNP_CDIV(n, 16 * G), 2), dim3(256), 0, s, xc, cc, n, \
                           pw.l0_whh[0], pw.l0_whh[1], pw.l0_wih[0], pw.l0_wih[1], pw.l0_wlast[0], pw.l0_wlast[1], ctx->ws_h0)
            if (ctx->l0_wx_lds && nsg == 1)
                hipLaunchKernelGGL((k_pileup_l0_rs32<1, true>), dim3((unsigned)NSNP_CDIV(n, 16), 2), dim3(256), 0, s, xc, cc, n,
                                   pw.l0_whh[0], pw.l0_whh[1], pw.l0_wih[0], pw.l0_wih[1], pw.l0_wlast[0], pw.l0_wlast[1], ctx->ws_h0);
            else
            if (nsg == 4) LAUNCH_RS(4); else if (nsg == 2) LAUNCH_RS(2); else LAUNCH_RS(1);
#undef LAUNCH_RS
        } else
        { ScopedKernelTimer tm(ctx, NSNP_K_L0, s);
#define LAUNCH_L0(W) hipLaunchKernelGGL(k_pileup_l0<W>, g_rec, dim3(64 * W), L0_LDS_BYTES, s, xc, cc, n, \
                           pw.l0_whh[0], pw.l0_whh[1], pw.l0_wih[0], pw.l0_wih[1], pw.l0_wlast[0], pw.l0_wlast[1], ctx->ws_h0)
        if (wpb == 8) LAUNCH_L0(8); else if (wpb == 4) LAUNCH_L0(4); else if (wpb == 2) LAUNCH_L0(2); else LAUNCH_L0(1);
#undef LAUNCH_L0
        }
        if (ctx->l1_rs == 1) {
            ScopedKernelTimer tm(ctx, NSNP_K_L1, s);
            // two 4-wave workgroups per CU (registers), 16 sites each
            int g1 = 1;            // (the 32-site build needs 11 more registers than a wave may hold and spills)
            if (ctx->l1_rs_groups == 1 || ctx->l1_rs_groups == 2) g1 = ctx->l1_rs_groups;
#define LAUNCH_R4(G) hipLaunchKernelGGL(k_pileup_l1_rs4<G>, dim3((unsigned)NSNP_CDIV(n, 16 * G), 2), dim3(256), rs4_l1_lds_bytes(G), s, \
                           ctx->ws_h0, n, pw.l1_wih[0], pw.l1_wih[1], pw.l1_whh[0], pw.l1_whh[1], pw.l1_bias[0], pw.l1_bias[1], ctx->ws_h1c)
            if (g1 == 2) LAUNCH_R4(2); else LAUNCH_R4(1);
#undef LAUNCH_R4
        } else if (ctx->l1_rs == 2) {
            ScopedKernelTimer tm(ctx, NSNP_K_L1, s);
            // one 8-wave workgroup per CU (LDS + registers): 64 sites each when that still gives every CU one, else 32
            int g1 = NSNP_CDIV(n, 64) * 2 >= (int64_t)ctx->n_cu ? 4 : 2;
            if (ctx->l1_rs_groups) g1 = ctx->l1_rs_groups;
#define LAUNCH_R1(G) hipLaunchKernelGGL((k_pileup_l1_rs32<G, ST>), dim3((unsigned)NSNP_CDIV(n, 16 * G), 2), dim3(512), rs32_l1_lds_bytes(G), s, \
                           ctx->ws_h0, n, pw.l1_wih[0], pw.l1_wih[1], pw.l1_whh[0], pw.l1_whh[1], pw.l1_bias[0], pw.l1_bias[1], ctx->ws_h1c)
            if (ctx->l1_stagger) { constexpr bool ST = true; if (g1 == 4) LAUNCH_R1(4); else LAUNCH_R1(2); }
            else { constexpr bool ST = false; if (g1 == 4) LAUNCH_R1(4); else LAUNCH_R1(2); }
#undef LAUNCH_R1
        } else {
        { const int rx = nsnp_ctx_need_xp1(ctx); if (rx) return rx; }
        const int64_t n_rt = NSNP_CDIV(n * PSTEPS1, 16);
        // persistent workgroups: each loads the 128 KB weight image once and then walks
        // proj1_tiles 16-row tiles per wave, so the load is amortised even at small batches
        int64_t gp = NSNP_CDIV(n_rt, 16 * (int64_t)ctx->proj1_tiles);
        if (gp > ctx->n_cu) gp = ctx->n_cu;
        if (gp < 1) gp = 1;
        { ScopedKernelTimer tm(ctx, NSNP_K_PROJ1, s);
        hipLaunchKernelGGL(k_pileup_proj1, dim3((unsigned)gp, 2), dim3(1024), P1_LDS_BYTES, s, ctx->ws_h0, n,
                           pw.l1_wih[0], pw.l1_wih[1], pw.l1_bias[0], pw.l1_bias[1], ctx->ws_xp1); }
        { ScopedKernelTimer tm(ctx, NSNP_K_L1, s);
#define LAUNCH_L1(W) hipLaunchKernelGGL(k_pileup_l1<W>, g_rec, dim3(64 * W), L1_LDS_BYTES, s, ctx->ws_xp1, n, \
                           pw.l1_whh[0], pw.l1_whh[1], ctx->ws_h1c)
        if (wpb == 8) LAUNCH_L1(8); else if (wpb == 4) LAUNCH_L1(4); else if (wpb == 2) LAUNCH_L1(2); else LAUNCH_L1(1);
#undef LAUNCH_L1
        }
        }
        ScopedKernelTimer tm_head(ctx, NSNP_K_HEAD, s);
        if (ctx->head_rs) {
            int64_t gh = NSNP_CDIV(n, 16);
            if (gh > 2 * (int64_t)ctx->n_cu) gh = 2 * (int64_t)ctx->n_cu;          // persistent: two 8-wave workgroups per CU
            const PostOut po = post ? *post : PostOut{nullptr, nullptr, nullptr, nullptr};
            hipLaunchKernelGGL(k_pileup_head_rs, dim3((unsigned)gh), dim3(512), 0, s, ctx->ws_h1c, n,
                               pw.proj_w, pw.proj_b, pw.dense_w, pw.dense_b, pw.head_w, pw.head_b,
                               gt + base * NSNP_GT_CLASSES, zy + base * NSNP_ZY_CLASSES,
                               po.gt_arg ? po.gt_arg + base : nullptr, po.gt_arg ? po.zy_arg + base : nullptr,
                               po.gt_arg ? po.gt_max + base : nullptr, po.gt_arg ? po.zy_max + base : nullptr);
            if (post_written) *post_written = po.gt_arg != nullptr;
        } else
        hipLaunchKernelGGL(k_pileup_head, dim3((unsigned)NSNP_CDIV(n, 64)), dim3(256), 0, s, ctx->ws_h1c, n,
                           pw.proj_w, pw.proj_b, pw.dense_w, pw.dense_b, pw.head_w, pw.head_b,
                           gt + base * NSNP_GT_CLASSES, zy + base * NSNP_ZY_CLASSES);
    }
    NSNP_HIP(ctx, hipGetLastError());
    return NSNP_OK;
}

NSNP_DEVCLK_READER(nsnp_devclk_read_pileup)

extern "C" int nsnp_pileup_postprocess(nsnp_ctx* ctx, const float* gt_prob, const float* zy_prob,
                                       const int32_t* x, int64_t N, uint8_t* gt_arg, uint8_t* zy_arg,
                                       float* gt_max, float* zy_max, int32_t* depth, void* stream)
{
    if (!ctx || N < 0 || !gt_prob || !zy_prob || !gt_arg || !zy_arg || !gt_max || !zy_max || (depth && !x)) return NSNP_EINVAL;
    if (N == 0) return NSNP_OK;
    hipLaunchKernelGGL(k_pileup_post, dim3((unsigned)NSNP_CDIV(N, 256)), dim3(256), 0, (hipStream_t)stream,
                       gt_prob, zy_prob, x, N, gt_arg, zy_arg, gt_max, zy_max, depth);
    NSNP_HIP(ctx, hipGetLastError());
    return NSNP_OK;
}
