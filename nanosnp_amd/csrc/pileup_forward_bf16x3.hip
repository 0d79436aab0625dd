// pileup_forward_bf16x3.hip -- PileupModel forward with every fp32 matrix product evaluated on the bf16 matrix pipe at FULL
// fp32 operand width ("bf16x3", pileup_precision 2):
//
//     w = w0 + w1 + w2,   x = x0 + x1 + x2      (v0 = bf16(v), v1 = bf16(v - v0), v2 = bf16(v - v0 - v1), round to nearest even:
//                                                 8 + 8 + 8 significand bits and the fp32 exponent range - the split is EXACT for
//                                                 every finite fp32 whose third term does not underflow, |v| > 2^-110)
//     w.x ~= w0.x0 + (w0.x1 + w1.x0) + (w0.x2 + w1.x1 + w2.x0)        six v_mfma_f32_16x16x32_bf16, fp32 accumulation
//
// bf16 products are exact in the fp32 accumulator; the three dropped terms (w1.x2, w2.x1, w2.x2) are below 2^-24 |w0 x0| (each
// residual is at most half an ulp of the term above it: 2^-9 . 2^-17), i.e. at the rounding error fp32 arithmetic itself makes per
// product.  Unlike the opt-in f16x3 mode (two fp16 terms, 21-22 bits, fp16 range: pileup_forward_f16x3.hip) nothing here is
// narrower than the reference's fp32: operands keep 24 bits and the full exponent range, sums are fp32.  What the mode buys: the
// fp32 MFMA (v_mfma_f32_16x16x4_f32: 256 flop per cycle and SIMD, sharing the lanes with the vector ALU) is replaced by six bf16
// MFMAs per 32-deep K block at 2048 flop per cycle that leave the vector ALU to the LSTM cell: 2.7x less matrix-pipe time.
//
// Kernels (same reference functions as pileup_forward.hip: LSTMNetwork.predict, PileupModel/model.py:31-39,66-73,114-119):
//   k_pileup_l0_b3    layer-0 BiLSTM recurrence, 33 steps; 4 waves x 4 gate tiles, weights (3 planes) in 144 VGPRs, h through LDS
//   k_pileup_l1_b3    layer-1 input projection + recurrence, 17 steps per direction; 8 waves x 2 gate tiles, 144 VGPRs of weights
//   k_pileup_head_b3  output_proj at t = 16, tanh(dense), heads, softmax, argmax / max (predict.py:54-57)
// Intermediates: H0 [16-site group][t][dir][plane 3][K block 2][quarter 4][site 16][8] bf16 (768 B per site and step, written
// already split and in the LDS operand order, so layer 0 flushes and layer 1 stages it with linear 16-byte copies), H1c
// [site][dir][64 units] fp32.
#include "nsnp_common.hpp"
#include "nsnp_lstm_cell.hpp"
#include <type_traits>

namespace {

typedef __bf16 b8 __attribute__((ext_vector_type(8)));
typedef __bf16 b4 __attribute__((ext_vector_type(4)));
typedef __bf16 b2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f32x4 mfma_b(b8 a, b8 b, f32x4 c)
{
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
// v = p0 + p1 + p2 exactly (see the header); hipcc emits v_cvt_pk_bf16_f32 for the casts (round to nearest even, NaN stays NaN)
__device__ __forceinline__ void split3(float v, __bf16& p0, __bf16& p1, __bf16& p2)
{
    p0 = (__bf16)v;
    const float r1 = v - (float)p0;
    p1 = (__bf16)r1;
    p2 = (__bf16)(r1 - (float)p1);
}
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// the six products of one K block for NT tiles, smallest terms first; a[t][plane], b[plane]; MFMAs on one accumulator are NT issue
// slots apart.  NB = planes of the B operand that are non-zero (3, or fewer for small integer inputs: x1 = x2 = 0 up to |x| = 256).
template <int NT, int NB = 3>
__device__ __forceinline__ void six_products(const b8 (*a)[3], const b8* b, f32x4* acc)
{
    if (NB > 2) {
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = mfma_b(a[t][0], b[2], acc[t]);
    }
    if (NB > 1) {
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = mfma_b(a[t][1], b[1], acc[t]);
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = mfma_b(a[t][2], b[0], acc[t]);
    if (NB > 1) {
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = mfma_b(a[t][0], b[1], acc[t]);
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = mfma_b(a[t][1], b[0], acc[t]);
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = mfma_b(a[t][0], b[0], acc[t]);
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Operand images in LDS and in H0.  The B fragment of lane (site n, quarter q) for K block kb of plane p is 8 bf16 = 16 bytes; the
// images keep the 16 sites of a group NEXT TO EACH OTHER:
//     frag[plane p][K block kb][quarter q][site n][8 bf16]          (256 B per (p, kb, q); 6 KB per 16 sites of a K = 64 operand)
// ds_read_b128 serves a wave in four groups of 16 lanes that mix quarters ({0-3, 12-15, 20-27}, ...: MI355X_MICROARCH.md, LDS) but
// hold every n exactly once, so with the site index in address bits 4-7 every group touches 16 distinct 16-byte bank slots whatever
// its quarters are: the reads are conflict-free (a row-per-site layout with any padding measured 46 % conflict cycles).  H0 keeps the
// same image per (16-site group, t, direction), so layer 0 flushes and layer 1 stages it with linear 16-byte copies.
// K position 32 kb + 8 q + j of a layer-0 row <-> unit 16 w + 4 u + q' with 16 w + 4 q' + u = that position (the four units u lane
// (n, q') of wave w leaves the cell with: one 8-byte write per plane); layer 1: position 8 w + 2 q' + u <-> unit 4 (2 w + u) + q'.
// ---------------------------------------------------------------------------------------------------------------------------------
constexpr int FR_H = 3 * 2 * 4 * 16 * 8;     // bf16 per 16-site image of a K = 64 operand (h rows): 3072 = 6 KB
constexpr int FR_X = 3 * 1 * 4 * 16 * 8;     // ... of the K = 32 layer-0 input operand: 1536 = 3 KB
// (+ the cell state of a 64-site workgroup: 16 B per lane, wave and site group, kept in LDS because 144 weight registers, two sets
// of accumulators and 16 cell states do not fit the 256 registers of a wave at two waves per SIMD)
constexpr int b3_l0_lds_bytes(int nsg) { return 2 * nsg * (FR_H + FR_X) * 2 + 64 + (nsg == 4 ? 4 * 4 * 64 * 16 : 0); }

// ---------------------------------------------------------------------------------------------------------------------------------
// layer 0.  Workgroup = 4 waves x NSG groups of 16 sites, one direction (blockIdx.y); wave w owns gate tiles 4w..4w+3 (hidden units
// 16w..16w+15, all four gates) and holds their W_hh (2 K blocks) and W_ih (1 K block: 18 channels + the bias column) planes in 144
// VGPRs for the whole kernel.  h_t crosses waves through a double-buffered LDS image, one LDS-only barrier per step, and is stored
// to H0 by the lanes that computed it.  Input counts are integers: up to |x| = 256 exact in ONE bf16 (every real pileup: the depth
// cap of mpileup is 144), so the input part runs 3 products; a wave that stages a larger count raises a per-step flag and the step
// runs 5 (|x| <= 65536) or all 6.
// ---------------------------------------------------------------------------------------------------------------------------------
template <int NSG>
__global__ __launch_bounds__(256, 2) void k_pileup_l0_b3(
    const int32_t* __restrict__ x, const int64_t* __restrict__ center_idx, int64_t N,
    const __bf16* __restrict__ whh0, const __bf16* __restrict__ whh1,
    const __bf16* __restrict__ wih0, const __bf16* __restrict__ wih1,
    __bf16* __restrict__ H0 /* [16-site group][t][dir][FR_H], padded to a multiple of 64 sites: stores are unconditional */)
{
    extern __shared__ b8 lds_b3[];
    __bf16* const hx = reinterpret_cast<__bf16*>(lds_b3);                 // [2][NSG][FR_H]
    __bf16* const xx = hx + 2 * NSG * FR_H;                               // [2][NSG][FR_X]
    int* const xflag = reinterpret_cast<int*>(xx + 2 * NSG * FR_X);       // [2][4]
    constexpr bool CL = NSG == 4;                                         // cell state in LDS
    f32x4* const cst = reinterpret_cast<f32x4*>(xflag + 16);              // [NSG][4 waves][64 lanes] (CL only)
    const int dir = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 15, q = lane >> 4;

    // ---- weights of this wave's four gate tiles -> registers: image [tile][kb][plane][lane] of 8 bf16 ----------------------
    b8 Whh[2][4][3], Wih[4][3];
    {
        const b8* __restrict__ ghh = reinterpret_cast<const b8*>(dir ? whh1 : whh0);
        const b8* __restrict__ gih = reinterpret_cast<const b8*>(dir ? wih1 : wih0);
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int p = 0; p < 3; ++p) {
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) Whh[kb][u][p] = ghh[(((4 * wave + u) * 2 + kb) * 3 + p) * 64 + lane];
                Wih[u][p] = gih[((4 * wave + u) * 3 + p) * 64 + lane];
            }
    }

    // ---- input staging: wave w (< NSG) converts x_t of site group w to bf16 planes -----------------------------------------
    const int64_t group0 = (int64_t)blockIdx.x * NSG;                     // first 16-site group of this workgroup
    const bool stager = wave < NSG;
    const int64_t xsite = (group0 + wave) * 16 + n;
    const int64_t xsc = (stager && xsite < N) ? xsite : N - 1;
    const int32_t* __restrict__ xs = center_idx ? x + (center_idx[xsc] - PCENTER) * PC : x + xsc * (PW * PC);
    // lane (n, q) stages K positions 8q .. 8q+7 of its site: channels 0-7, 8-15, (16, 17, the constant 1 of the bias column, 0...), zeros.
    // The counts come straight from HBM (one 72-byte row per step, no reuse), so they are requested TWO steps ahead into two register
    // sets (a load issued one step ahead was still in flight when its step came: 0.38 ms of a 2.3 ms launch, by removal timing).
    int xa[8], xb_[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { xa[j] = (q == 2 && j == 2) ? 1 : 0; xb_[j] = xa[j]; }
    auto load_x = [&](int (&xi)[8], int t) __attribute__((always_inline)) {
        if (!stager) return;
        const int32_t* p = xs + t * PC;
        if (q < 2) {
#pragma unroll
            for (int j = 0; j < 8; ++j) xi[j] = p[8 * q + j];
        } else if (q == 2) {
            xi[0] = p[16]; xi[1] = p[17];
        }
    };
    int xdirty = 0;                                        // bit b: planes 1, 2 of this wave's image in buffer b hold the values of an earlier step
    auto stage_x = [&](const int (&xi)[8], int buf) __attribute__((always_inline)) {
        if (!stager) return;
        __bf16* row = xx + (size_t)(buf * NSG + wave) * FR_X + q * 128 + n * 8;
        bool big = false;                                  // counts are exact in one bf16 up to +-256
#pragma unroll
        for (int j = 0; j < 8; ++j) big |= (unsigned)(xi[j] + 256) > 512u;
        int lvl = 0;
        if (__builtin_expect(__ballot(big) == 0ull, 1)) {
            b8 p0;
#pragma unroll
            for (int j = 0; j < 8; ++j) p0[j] = (__bf16)(float)xi[j];          // predict.py:49 int -> float
            *reinterpret_cast<b8*>(row) = p0;
            // The step's level is the workgroup's (the largest of its site groups): a group below it multiplies its planes 1 and 2 as
            // well, so they must be ZERO, not what an earlier step left there (they start as zeros: the loop in front of the first barrier)
            if (xdirty & (1 << buf)) {
                *reinterpret_cast<b8*>(row + 512) = b8{0, 0, 0, 0, 0, 0, 0, 0};
                *reinterpret_cast<b8*>(row + 1024) = b8{0, 0, 0, 0, 0, 0, 0, 0};
                xdirty &= ~(1 << buf);
            }
        } else {
            xdirty |= 1 << buf;
            b8 p0, p1, p2;
            bool nz1 = false, nz2 = false;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                __bf16 a, b, c; split3((float)xi[j], a, b, c);
                p0[j] = a; p1[j] = b; p2[j] = c;
                nz1 |= (float)b != 0.f; nz2 |= (float)c != 0.f;
            }
            lvl = __ballot(nz2) != 0ull ? 2 : (__ballot(nz1) != 0ull ? 1 : 0);
            *reinterpret_cast<b8*>(row) = p0;
            *reinterpret_cast<b8*>(row + 512) = p1;
            *reinterpret_cast<b8*>(row + 1024) = p2;
        }
        if (lane == 0) xflag[buf * 4 + wave] = lvl;
    };
    if (tid < 8) xflag[tid] = 0;
    for (int i = tid; i < 2 * NSG * FR_X / 8; i += 256) reinterpret_cast<b8*>(xx)[i] = b8{0, 0, 0, 0, 0, 0, 0, 0};
    // h_{-1} = 0: the buffer step 0 reads
    for (int i = tid; i < NSG * FR_H / 8; i += 256) reinterpret_cast<b8*>(hx + (size_t)NSG * FR_H)[i] = b8{0, 0, 0, 0, 0, 0, 0, 0};
    __syncthreads();
    auto t_of = [&](int s) { return dir ? PW - 1 - s : s; };
    load_x(xa, t_of(0));
    stage_x(xa, 0);
    load_x(xb_, t_of(1));                                   // set B: odd steps, set A: even steps
    __syncthreads();

    float c[CL ? 1 : 4 * NSG];
#pragma unroll
    for (int i = 0; i < (CL ? 1 : 4 * NSG); ++i) c[i] = 0.f;
    if (CL) {
#pragma unroll
        for (int sg = 0; sg < NSG; ++sg) cst[(sg * 4 + wave) * 64 + lane] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    // offset of this lane's four units (K positions 16 wave + 4 q + u) inside a 16-site image, plane 0
    const int wofs = (wave >> 1) * 512 + (2 * (wave & 1) + (q >> 1)) * 128 + n * 8 + 4 * (q & 1);

    auto step = [&](auto lvl_tag, int t, int xb, int hw_) __attribute__((always_inline)) {
        constexpr int LVL = decltype(lvl_tag)::value;
        const int hr = hw_ ^ 1;
#pragma unroll
        for (int sg = 0; sg < NSG; ++sg) {
            const __bf16* hrow = hx + (size_t)(hr * NSG + sg) * FR_H + q * 128 + n * 8;
            const __bf16* xrow = xx + (size_t)(xb * NSG + sg) * FR_X + q * 128 + n * 8;
            b8 bh[2][3], bx[3];
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                bh[0][p] = *reinterpret_cast<const b8*>(hrow + p * 1024);
                bh[1][p] = *reinterpret_cast<const b8*>(hrow + p * 1024 + 512);
                if (p <= LVL) bx[p] = *reinterpret_cast<const b8*>(xrow + p * 512);
            }
            f32x4 acc[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) acc[u] = f32x4{0.f, 0.f, 0.f, 0.f};
            six_products<4, LVL + 1>(Wih, bx, acc);             // input part (the bias rides on its constant-1 column)
#ifndef NSNP_B3_NOREC
            six_products<4>(Whh[0], bh[0], acc);
            six_products<4>(Whh[1], bh[1], acc);
#endif
            // cell: lane (n, q) holds units 4 (4 wave + u) + q, u = 0..3 = K positions 16 wave + 4 q + u
            f32x4 cv;
            if (CL) cv = cst[(sg * 4 + wave) * 64 + lane];
            else cv = f32x4{c[CL ? 0 : 4 * sg], c[CL ? 0 : 4 * sg + 1], c[CL ? 0 : 4 * sg + 2], c[CL ? 0 : 4 * sg + 3]};
            b4 n0, n1, n2;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                float cn;
#ifdef NSNP_B3_NOCELL
                const float h = acc[u][0] * 1e-3f + acc[u][1] * 1e-4f; cn = acc[u][2] + acc[u][3];
#else
                const float h = nsnp_cell::lstm_cell(acc[u][0], acc[u][1], acc[u][2], acc[u][3], cv[u], cn);
#endif
                cv[u] = cn;
                __bf16 a, b, d;
#ifdef NSNP_B3_NOSPLIT
                a = (__bf16)h; b = a; d = a;
#else
                split3(h, a, b, d);
#endif
                n0[u] = a; n1[u] = b; n2[u] = d;
            }
            if (CL) cst[(sg * 4 + wave) * 64 + lane] = cv;
            else {
#pragma unroll
                for (int u = 0; u < 4; ++u) c[CL ? 0 : 4 * sg + u] = cv[u];
            }
            // h_t: to the exchange image for the other waves and straight to H0 (the same image: 32 lanes fill a 256-byte run)
            __bf16* w = hx + (size_t)(hw_ * NSG + sg) * FR_H + wofs;
            *reinterpret_cast<b4*>(w) = n0;
            *reinterpret_cast<b4*>(w + 1024) = n1;
            *reinterpret_cast<b4*>(w + 2048) = n2;
#ifndef NSNP_B3_NOH0
            __bf16* g = H0 + (((group0 + sg) * PW + t) * 2 + dir) * FR_H + wofs;
            *reinterpret_cast<b4*>(g) = n0;
            *reinterpret_cast<b4*>(g + 1024) = n1;
            *reinterpret_cast<b4*>(g + 2048) = n2;
#endif
        }
    };

    // step s: the MFMAs and cells of step s, then the input image of step s + 1 (requested at step s - 1), one barrier
    auto one_step = [&](int s, int (&mine)[8], const int (&next)[8]) __attribute__((always_inline)) {
        const int t = t_of(s);
        const int xb = s & 1;
#ifndef NSNP_B3_NOX
        if (s + 2 < PW) load_x(mine, t_of(s + 2));          // this step's counts were staged at the end of step s - 1: the set is free
#endif
        const int lvl = max(max(xflag[xb * 4 + 0], xflag[xb * 4 + 1]), max(xflag[xb * 4 + 2], xflag[xb * 4 + 3]));
        if (__builtin_expect(lvl == 0, 1)) step(std::integral_constant<int, 0>{}, t, xb, s & 1);
        else if (lvl == 1)                 step(std::integral_constant<int, 1>{}, t, xb, s & 1);
        else                               step(std::integral_constant<int, 2>{}, t, xb, s & 1);
#ifndef NSNP_B3_NOX
        if (s + 1 < PW) stage_x(next, xb ^ 1);
#endif
#ifndef NSNP_B3_NOBAR
        lds_barrier();
#endif
    };
    for (int s = 0; s + 1 < PW; s += 2) {
        one_step(s, xa, xb_);
        one_step(s + 1, xb_, xa);
    }
    one_step(PW - 1, xa, xb_);                               // PW is odd
}

// ---------------------------------------------------------------------------------------------------------------------------------
// layer 0, SKEWED over two site groups (the default at batch sizes that fill the chip with 32-site workgroups).  In k_pileup_l0_b3
// the cell of a step's last site group, the exchange write and the barrier stand between the MFMAs of one step and the next: by
// removal timing the matrix work is 1.4 ms of a 2.1 ms launch (N = 131072).  Here a workgroup owns two groups A and B and runs them
// half a step apart, two barriers per step:
//     phase 1 of step s:   MFMAs of B(s)     beside   cell of A(s)  (+ the input image of step s + 1)      barrier
//     phase 2 of step s:   MFMAs of A(s + 1) beside   cell of B(s)                                          barrier
// so every cell has 60 independent MFMAs of the other group in its scheduling region, and what a barrier waits for is the other
// waves' cells, not a drained matrix pipe.  Same arithmetic in the same order per site: bit-identical to k_pileup_l0_b3.
// ---------------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void k_pileup_l0_b3p(
    const int32_t* __restrict__ x, const int64_t* __restrict__ center_idx, int64_t N,
    const __bf16* __restrict__ whh0, const __bf16* __restrict__ whh1,
    const __bf16* __restrict__ wih0, const __bf16* __restrict__ wih1,
    __bf16* __restrict__ H0)
{
    extern __shared__ b8 lds_b3[];
    constexpr int NSG = 2;
    __bf16* const hx = reinterpret_cast<__bf16*>(lds_b3);                 // [2][NSG][FR_H]
    __bf16* const xx = hx + 2 * NSG * FR_H;                               // [2][NSG][FR_X]
    int* const xflag = reinterpret_cast<int*>(xx + 2 * NSG * FR_X);       // [2][4]
    const int dir = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 15, q = lane >> 4;

    b8 Whh[2][4][3], Wih[4][3];
    {
        const b8* __restrict__ ghh = reinterpret_cast<const b8*>(dir ? whh1 : whh0);
        const b8* __restrict__ gih = reinterpret_cast<const b8*>(dir ? wih1 : wih0);
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int p = 0; p < 3; ++p) {
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) Whh[kb][u][p] = ghh[(((4 * wave + u) * 2 + kb) * 3 + p) * 64 + lane];
                Wih[u][p] = gih[((4 * wave + u) * 3 + p) * 64 + lane];
            }
    }

    const int64_t group0 = (int64_t)blockIdx.x * NSG;
    const bool stager = wave < NSG;
    const int64_t xsite = (group0 + wave) * 16 + n;
    const int64_t xsc = (stager && xsite < N) ? xsite : N - 1;
    const int32_t* __restrict__ xs = center_idx ? x + (center_idx[xsc] - PCENTER) * PC : x + xsc * (PW * PC);
    int xa[8], xb_[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { xa[j] = (q == 2 && j == 2) ? 1 : 0; xb_[j] = xa[j]; }
    auto load_x = [&](int (&xi)[8], int t) __attribute__((always_inline)) {
        if (!stager) return;
        const int32_t* p = xs + t * PC;
        if (q < 2) {
#pragma unroll
            for (int j = 0; j < 8; ++j) xi[j] = p[8 * q + j];
        } else if (q == 2) {
            xi[0] = p[16]; xi[1] = p[17];
        }
    };
    int xdirty = 0;                                        // bit b: planes 1, 2 of this wave's image in buffer b hold the values of an earlier step
    auto stage_x = [&](const int (&xi)[8], int buf) __attribute__((always_inline)) {
        if (!stager) return;
        __bf16* row = xx + (size_t)(buf * NSG + wave) * FR_X + q * 128 + n * 8;
        bool big = false;
#pragma unroll
        for (int j = 0; j < 8; ++j) big |= (unsigned)(xi[j] + 256) > 512u;
        int lvl = 0;
        if (__builtin_expect(__ballot(big) == 0ull, 1)) {
            b8 p0;
#pragma unroll
            for (int j = 0; j < 8; ++j) p0[j] = (__bf16)(float)xi[j];
            *reinterpret_cast<b8*>(row) = p0;
            if (xdirty & (1 << buf)) {                     // (planes 1, 2 of a group below the workgroup's level must read as zeros: k_pileup_l0_b3)
                *reinterpret_cast<b8*>(row + 512) = b8{0, 0, 0, 0, 0, 0, 0, 0};
                *reinterpret_cast<b8*>(row + 1024) = b8{0, 0, 0, 0, 0, 0, 0, 0};
                xdirty &= ~(1 << buf);
            }
        } else {
            xdirty |= 1 << buf;
            b8 p0, p1, p2;
            bool nz1 = false, nz2 = false;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                __bf16 a, b, c; split3((float)xi[j], a, b, c);
                p0[j] = a; p1[j] = b; p2[j] = c;
                nz1 |= (float)b != 0.f; nz2 |= (float)c != 0.f;
            }
            lvl = __ballot(nz2) != 0ull ? 2 : (__ballot(nz1) != 0ull ? 1 : 0);
            *reinterpret_cast<b8*>(row) = p0;
            *reinterpret_cast<b8*>(row + 512) = p1;
            *reinterpret_cast<b8*>(row + 1024) = p2;
        }
        if (lane == 0) xflag[buf * 4 + wave] = lvl;
    };
    if (tid < 8) xflag[tid] = 0;
    for (int i = tid; i < 2 * NSG * FR_X / 8; i += 256) reinterpret_cast<b8*>(xx)[i] = b8{0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = tid; i < NSG * FR_H / 8; i += 256) reinterpret_cast<b8*>(hx + (size_t)NSG * FR_H)[i] = b8{0, 0, 0, 0, 0, 0, 0, 0};
    __syncthreads();
    auto t_of = [&](int s) { return dir ? PW - 1 - s : s; };
    load_x(xa, t_of(0));
    stage_x(xa, 0);
    load_x(xb_, t_of(1));
    __syncthreads();

    float c[4 * NSG];
#pragma unroll
    for (int i = 0; i < 4 * NSG; ++i) c[i] = 0.f;
    const int wofs = (wave >> 1) * 512 + (2 * (wave & 1) + (q >> 1)) * 128 + n * 8 + 4 * (q & 1);

    // gate pre-activations of site group sg at step s: 60 MFMAs (+ 8 / 12 for counts beyond 256 / 65536)
    auto gemm = [&](int sg, int s, f32x4* acc) __attribute__((always_inline)) {
        const int xbuf = s & 1, hr = (s & 1) ^ 1;
        const int lvl = max(max(xflag[xbuf * 4 + 0], xflag[xbuf * 4 + 1]), max(xflag[xbuf * 4 + 2], xflag[xbuf * 4 + 3]));
        const __bf16* hrow = hx + (size_t)(hr * NSG + sg) * FR_H + q * 128 + n * 8;
        const __bf16* xrow = xx + (size_t)(xbuf * NSG + sg) * FR_X + q * 128 + n * 8;
        b8 bh[2][3], bx[3];
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            bh[0][p] = *reinterpret_cast<const b8*>(hrow + p * 1024);
            bh[1][p] = *reinterpret_cast<const b8*>(hrow + p * 1024 + 512);
        }
        bx[0] = *reinterpret_cast<const b8*>(xrow);
        // the first product of a chain takes a literal zero as its C operand (an inline constant of the MFMA: no registers to clear)
        const f32x4 zero4 = f32x4{0.f, 0.f, 0.f, 0.f};
        if (__builtin_expect(lvl != 0, 0)) {                // rare: counts beyond one bf16 term - the extra products first (smallest terms first)
            bx[1] = *reinterpret_cast<const b8*>(xrow + 512);
            bx[2] = *reinterpret_cast<const b8*>(xrow + 1024);
#pragma unroll
            for (int u = 0; u < 4; ++u) acc[u] = mfma_b(Wih[u][0], bx[2], zero4);
#pragma unroll
            for (int u = 0; u < 4; ++u) acc[u] = mfma_b(Wih[u][1], bx[1], acc[u]);
#pragma unroll
            for (int u = 0; u < 4; ++u) acc[u] = mfma_b(Wih[u][2], bx[0], acc[u]);
#pragma unroll
            for (int u = 0; u < 4; ++u) acc[u] = mfma_b(Wih[u][0], bx[1], acc[u]);
        } else {
#pragma unroll
            for (int u = 0; u < 4; ++u) acc[u] = mfma_b(Wih[u][2], bx[0], zero4);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) acc[u] = mfma_b(Wih[u][1], bx[0], acc[u]);
#pragma unroll
        for (int u = 0; u < 4; ++u) acc[u] = mfma_b(Wih[u][0], bx[0], acc[u]);
        six_products<4>(Whh[0], bh[0], acc);
        six_products<4>(Whh[1], bh[1], acc);
    };
    auto cell = [&](int sg, int s, const f32x4* acc) __attribute__((always_inline)) {
        b4 n0, n1, n2;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            float cn;
            const float h = nsnp_cell::lstm_cell(acc[u][0], acc[u][1], acc[u][2], acc[u][3], c[4 * sg + u], cn);
            c[4 * sg + u] = cn;
            __bf16 a, b, d; split3(h, a, b, d);
            n0[u] = a; n1[u] = b; n2[u] = d;
        }
        __bf16* w = hx + (size_t)((s & 1) * NSG + sg) * FR_H + wofs;
        *reinterpret_cast<b4*>(w) = n0;
        *reinterpret_cast<b4*>(w + 1024) = n1;
        *reinterpret_cast<b4*>(w + 2048) = n2;
        __bf16* g = H0 + (((group0 + sg) * PW + t_of(s)) * 2 + dir) * FR_H + wofs;
        *reinterpret_cast<b4*>(g) = n0;
        *reinterpret_cast<b4*>(g + 1024) = n1;
        *reinterpret_cast<b4*>(g + 2048) = n2;
    };

    f32x4 accA[4], accB[4];
    gemm(0, 0, accA);
    auto one_step = [&](int s, int (&mine)[8], const int (&next)[8]) __attribute__((always_inline)) {
        // phase 1: MFMAs of B(s) beside the cell of A(s); the input image of step s + 1
        if (s + 2 < PW) load_x(mine, t_of(s + 2));
        gemm(1, s, accB);
        cell(0, s, accA);
        if (s + 1 < PW) stage_x(next, (s & 1) ^ 1);
        lds_barrier();
        // phase 2: MFMAs of A(s + 1) beside the cell of B(s)
        if (s + 1 < PW) gemm(0, s + 1, accA);
        cell(1, s, accB);
        lds_barrier();
    };
    for (int s = 0; s + 1 < PW; s += 2) {
        one_step(s, xa, xb_);
        one_step(s + 1, xb_, xa);
    }
    one_step(PW - 1, xa, xb_);
}

// ---------------------------------------------------------------------------------------------------------------------------------
// layer 1: input projection (K = 128: h0_t of both directions) fused into the recurrence (K = 64), only the 17 steps per direction
// that reach position 16 (model.py:68).  Workgroup = 8 waves x NSG groups of 16 sites; wave w owns gate tiles 2w, 2w+1 and keeps
// their W_ih1 (4 K blocks) and W_hh1 (2 K blocks) planes in 144 VGPRs - one workgroup per CU (two waves per SIMD).  LDS holds
// operands only: the h0_t images of the site groups (12 KB per group: both directions, copied from H0 one step ahead,
// double-buffered) and the h1 exchange images.  A step is 6 NSG pieces per wave (site group x (4 input + 2 recurrent K blocks)):
// three 16-byte fragment reads and 12 MFMAs each, requested two pieces ahead, the cells of one site group between the MFMAs of the
// next.
// ---------------------------------------------------------------------------------------------------------------------------------
constexpr int b3_l1_lds_bytes(int nsg) { return 2 * nsg * (2 * FR_H + FR_H) * 2; }

template <int NSG>
__global__ __launch_bounds__(512, 1) void k_pileup_l1_b3(
    const __bf16* __restrict__ H0 /* [16-site group][t][dir][FR_H], padded to a multiple of 64 sites */, int64_t N,
    const __bf16* __restrict__ wih0, const __bf16* __restrict__ wih1,
    const __bf16* __restrict__ whh0, const __bf16* __restrict__ whh1,
    const float* __restrict__ bias0, const float* __restrict__ bias1,
    float* __restrict__ H1c /* padded likewise */)
{
    extern __shared__ b8 lds_b3[];
    __bf16* const h0s = reinterpret_cast<__bf16*>(lds_b3);                // [2][NSG][2 FR_H]
    __bf16* const h1x = h0s + 2 * NSG * 2 * FR_H;                         // [2][NSG][FR_H]
    const int dir = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 15, q = lane >> 4;
    const int64_t group0 = (int64_t)blockIdx.x * NSG;

    b8 Wih[4][2][3], Whh[2][2][3];                                        // [kb][tile u][plane]
    f32x4 bias[2];
    {
        const b8* __restrict__ gih = reinterpret_cast<const b8*>(dir ? wih1 : wih0);        // [tile][kb 4][plane][lane]
        const b8* __restrict__ ghh = reinterpret_cast<const b8*>(dir ? whh1 : whh0);        // [tile][kb 2][plane][lane]
        const f32x4* __restrict__ gb = reinterpret_cast<const f32x4*>(dir ? bias1 : bias0);  // [tile][lane] accumulator image (fp32 path)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
#pragma unroll
            for (int p = 0; p < 3; ++p) {
#pragma unroll
                for (int kb = 0; kb < 4; ++kb) Wih[kb][u][p] = gih[(((2 * wave + u) * 4 + kb) * 3 + p) * 64 + lane];
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) Whh[kb][u][p] = ghh[(((2 * wave + u) * 2 + kb) * 3 + p) * 64 + lane];
            }
            bias[u] = gb[(2 * wave + u) * 64 + lane];
        }
    }

    // ---- h0 staging: 768 pieces of 16 B per site group (both directions are adjacent in H0), linear ------------------------
    constexpr int NP = NSG * 768, SPT = (NP + 511) / 512;
    const __bf16* sgp[SPT];
    int sdo[SPT];
#pragma unroll
    for (int k = 0; k < SPT; ++k) {
        const int i = tid + 512 * k;
        const int sg = (i < NP ? i : 0) / 768, r = (i < NP ? i : 0) - 768 * sg;
        sgp[k] = H0 + ((group0 + sg) * PW) * (2 * FR_H) + 8 * r;            // H0 is padded: groups beyond N feed dead sites only
        sdo[k] = sg * (2 * FR_H) + 8 * r;
    }
    b8 sreg[SPT];
    auto load_h0 = [&](int t) {
#pragma unroll
        for (int k = 0; k < SPT; ++k) sreg[k] = *reinterpret_cast<const b8*>(sgp[k] + (size_t)t * (2 * FR_H));
    };
    auto store_h0 = [&](int buf) {
#pragma unroll
        for (int k = 0; k < SPT; ++k)
            if (NP % 512 == 0 || tid + 512 * k < NP) *reinterpret_cast<b8*>(h0s + (size_t)buf * NSG * 2 * FR_H + sdo[k]) = sreg[k];
    };
    // h1_{-1} = 0 in the buffer step 0 reads
    for (int i = tid; i < NSG * FR_H / 8; i += 512) reinterpret_cast<b8*>(h1x + (size_t)NSG * FR_H)[i] = b8{0, 0, 0, 0, 0, 0, 0, 0};
    load_h0(dir ? PW - 1 : 0);
    store_h0(0);
    __syncthreads();

    float c[2 * NSG];
#pragma unroll
    for (int i = 0; i < 2 * NSG; ++i) c[i] = 0.f;
    float last_h[NSG][2];

    for (int s = 0; s < PSTEPS1; ++s) {
        const int t = dir ? PW - 1 - s : s;
        const int cur = s & 1;
#ifndef NSNP_B3L1_NOSTAGE
        if (s + 1 < PSTEPS1) load_h0(dir ? t - 1 : t + 1);
#endif
        const __bf16* h0b = h0s + (size_t)cur * NSG * 2 * FR_H + q * 128 + n * 8;
        const __bf16* hrb = h1x + (size_t)(cur ^ 1) * NSG * FR_H + q * 128 + n * 8;      // h1_{s-1}
        __bf16* hwb = h1x + (size_t)cur * NSG * FR_H + ((wave >> 2) * 4 + (wave & 3)) * 128 + n * 8 + 2 * q;   // h1_s

        // piece P = sg * 6 + k: k < 4 input K block k (direction k >> 1, K block k & 1 of its image), k >= 4 recurrent K block k - 4
#ifndef NSNP_B3L1_AHEAD
#define NSNP_B3L1_AHEAD 2                  /* pieces the fragment reads run ahead of their MFMAs (A/B builds) */
#endif
        constexpr int AH = NSNP_B3L1_AHEAD;
        b8 fr[AH + 1][3];
        auto fetch = [&](int P, int slot) {
            const int sg = P / 6, k = P % 6;
            const __bf16* r = k < 4 ? h0b + sg * (2 * FR_H) + (k >> 1) * FR_H + (k & 1) * 512 : hrb + sg * FR_H + (k - 4) * 512;
#pragma unroll
            for (int p = 0; p < 3; ++p) fr[slot][p] = *reinterpret_cast<const b8*>(r + p * 1024);
        };
#pragma unroll
        for (int P = 0; P < AH; ++P) fetch(P, P);
        f32x4 acc[2][2];
#pragma unroll
        for (int P = 0; P < 6 * NSG; ++P) {
            const int sg = P / 6, k = P % 6, slot = P % (AH + 1), ab = sg & 1;
            if (P + AH < 6 * NSG) fetch(P + AH, (P + AH) % (AH + 1));
            if (k == 0) { acc[ab][0] = bias[0]; acc[ab][1] = bias[1]; }
            if (k < 4) six_products<2>(Wih[k], fr[slot], acc[ab]);
            else       six_products<2>(Whh[k - 4], fr[slot], acc[ab]);
            if (k == 5) {
                // cell of site group sg: lane (n, q) holds units 4 (2 wave + u) + q, u = 0, 1 = K positions 8 wave + 2 q + u
                b2 n0, n1, n2;
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    float cn;
#ifdef NSNP_B3L1_NOCELL
                    const float h = acc[ab][u][0] * 1e-3f + acc[ab][u][1] * 1e-4f; cn = acc[ab][u][2] + acc[ab][u][3];
#else
                    const float h = nsnp_cell::lstm_cell(acc[ab][u][0], acc[ab][u][1], acc[ab][u][2], acc[ab][u][3], c[2 * sg + u], cn);
#endif
                    c[2 * sg + u] = cn;
                    last_h[sg][u] = h;
                    __bf16 a, b, d; split3(h, a, b, d);
                    n0[u] = a; n1[u] = b; n2[u] = d;
                }
                __bf16* w = hwb + sg * FR_H;
                *reinterpret_cast<b2*>(w) = n0;
                *reinterpret_cast<b2*>(w + 1024) = n1;
                *reinterpret_cast<b2*>(w + 2048) = n2;
            }
        }
#ifndef NSNP_B3L1_NOSTAGE
        if (s + 1 < PSTEPS1) store_h0(cur ^ 1);
#endif
#ifndef NSNP_B3L1_NOBAR
        lds_barrier();
#endif
    }
    // H1c: fp32 [site][dir][unit]
#pragma unroll
    for (int sg = 0; sg < NSG; ++sg)
#pragma unroll
        for (int u = 0; u < 2; ++u) H1c[(((group0 + sg) * 16 + n) * 2 + dir) * 64 + 4 * (2 * wave + u) + q] = last_h[sg][u];
}

// ---------------------------------------------------------------------------------------------------------------------------------
// heads: output_proj at position 16 (model.py:37,68), tanh(dense) (:67), genotype / zygosity heads (:69-70), softmax (:117-118),
// argmax / max (predict.py:54-57).  One wave per 16 sites, 8 waves per workgroup sharing every weight image through one 96 KB LDS
// stage (proj, dense rows 0-127, dense rows 128-255, heads): the images leave L2 once per workgroup.
// ---------------------------------------------------------------------------------------------------------------------------------
constexpr int HB3_WAVES = 8;
constexpr int HB3_STAGE_B8 = 8 * 4 * 3 * 64;            // 96 KB: 8 tiles x 4 K blocks x 3 planes x 64 lanes of 16 B

// acc[t] += W(tile t, K block kb) . b for t < NT, kb < NKB; img: b8 elements [tile][kb][plane][lane]; tiles walked in groups of G
template <int NT, int NKB, int G>
__device__ __forceinline__ void stage_gemm_b3(const b8* img, int lane, const b8 (*b)[3], f32x4* acc)
{
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
#pragma unroll
        for (int ig = 0; ig < NT; ig += G) {
            b8 a[G][3];
#pragma unroll
            for (int u = 0; u < G; ++u)
#pragma unroll
                for (int p = 0; p < 3; ++p) a[u][p] = img[(((ig + u) * NKB + kb) * 3 + p) * 64 + lane];
            six_products<G>(a, b[kb], acc + ig);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

__global__ __launch_bounds__(64 * HB3_WAVES) void k_pileup_head_b3(
    const float* __restrict__ H1c, int64_t N,
    const __bf16* __restrict__ proj_w, const float* __restrict__ proj_b,
    const __bf16* __restrict__ dense_w, const float* __restrict__ dense_b,
    const __bf16* __restrict__ head_w, const float* __restrict__ head_b,
    float* __restrict__ gt_prob, float* __restrict__ zy_prob,
    uint8_t* __restrict__ gt_arg, uint8_t* __restrict__ zy_arg, float* __restrict__ gt_max, float* __restrict__ zy_max)
{
    extern __shared__ b8 lds_b3[];
    b8* const wst = lds_b3;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q = lane >> 4;
    const int64_t site = ((int64_t)blockIdx.x * HB3_WAVES + wave) * 16 + (lane & 15);
    const bool live = site < N;
    const int64_t sc = live ? site : N - 1;
    auto stage = [&](const __bf16* src, int n_b8) {
        __syncthreads();                                   // everyone is done with the previous image
        const b8* s8 = reinterpret_cast<const b8*>(src);
        for (int i = tid; i < n_b8; i += 64 * HB3_WAVES) wst[i] = s8[i];
        __syncthreads();
    };
    // B fragments of output_proj: K position 32 kb + 8 q + j = feature [fwd 64 | bwd 64] in natural order
    b8 bh[4][3];
    {
        const f32x4* __restrict__ hin = reinterpret_cast<const f32x4*>(H1c + sc * 128 + 8 * q);
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            const f32x4 v0 = hin[kb * 8], v1 = hin[kb * 8 + 1];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                __bf16 a, b, d; split3(j < 4 ? v0[j & 3] : v1[j & 3], a, b, d);
                bh[kb][0][j] = a; bh[kb][1][j] = b; bh[kb][2][j] = d;
            }
        }
    }
    // output_proj 128 -> 128
    f32x4 ap[8];
    {
        const f32x4* pb = reinterpret_cast<const f32x4*>(proj_b);      // [tile][q] rows 16 tile + 4 q + g
#pragma unroll
        for (int i = 0; i < 8; ++i) ap[i] = pb[i * 4 + q];
        stage(proj_w, 8 * 4 * 3 * 64);
        stage_gemm_b3<8, 4, 4>(wst, lane, bh, ap);
    }
    // dense 128 -> 256 + tanh: K position (kb, q, j) <-> proj feature 16 (2 kb + (j >> 2)) + 4 q + (j & 3): the lane's own registers
    b8 dh[4][3];
#pragma unroll
    for (int kb = 0; kb < 4; ++kb)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            __bf16 a, b, d; split3(ap[2 * kb + (j >> 2)][j & 3], a, b, d);
            dh[kb][0][j] = a; dh[kb][1][j] = b; dh[kb][2][j] = d;
        }
    f32x4 ad[16];
    {
        const f32x4* db = reinterpret_cast<const f32x4*>(dense_b);
#pragma unroll
        for (int i = 0; i < 16; ++i) ad[i] = db[i * 4 + q];
        stage(dense_w, 8 * 4 * 3 * 64);
        stage_gemm_b3<8, 4, 4>(wst, lane, dh, ad);
        stage(dense_w + (size_t)8 * 4 * 3 * 64 * 8, 8 * 4 * 3 * 64);
        stage_gemm_b3<8, 4, 4>(wst, lane, dh, ad + 8);
    }
    b8 eh[8][3];
#pragma unroll
    for (int kb = 0; kb < 8; ++kb)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float z = ad[2 * kb + (j >> 2)][j & 3];
            const float th = __builtin_fmaf(-2.0f, __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f((2.0f * nsnp_cell::LOG2E) * z)), 1.0f);
            __bf16 a, b, d; split3(th, a, b, d);
            eh[kb][0][j] = a; eh[kb][1][j] = b; eh[kb][2][j] = d;
        }
    f32x4 ah[2];
    {
        const f32x4* hb = reinterpret_cast<const f32x4*>(head_b);
        ah[0] = hb[q]; ah[1] = hb[4 + q];
        stage(head_w, 2 * 8 * 3 * 64);
        stage_gemm_b3<2, 8, 2>(wst, lane, eh, ah);
    }
    // lane (site, q) holds rows 4q..4q+3 of tile 0 and rows 16+4q.. of tile 1: genotype rows 0..20, zygosity rows 21..23
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
    const float z1 = ah[1][1], z2 = ah[1][2], z3 = ah[1][3];
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
        // predict.py:54-57 in the same launch: np.argmax / np.max over the stored probabilities (first maximum wins)
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

}  // namespace

// ---------------------------------------------------------------------------------------------------------------------------------
// host side: bf16 three-plane weight images
// ---------------------------------------------------------------------------------------------------------------------------------
namespace {

inline int gate_row(int row) { const int i = row >> 4, r = row & 15; return (r & 3) * PH + 4 * i + (r >> 2); }

inline uint16_t bf16_rne(float v)
{
    uint32_t u; memcpy(&u, &v, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);     // NaN stays NaN
    return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}
inline float bf16_f32(uint16_t h) { const uint32_t u = (uint32_t)h << 16; float v; memcpy(&v, &u, 4); return v; }

// img[tile][kb][plane][lane][j]  <-  f(row = 16 tile + (lane & 15), kb, q = lane >> 4, j)
template <typename F>
void pack_b3(uint16_t* img, int n_tiles, int n_kb, F f)
{
    for (int tile = 0; tile < n_tiles; ++tile)
        for (int kb = 0; kb < n_kb; ++kb)
            for (int lane = 0; lane < 64; ++lane)
                for (int j = 0; j < 8; ++j) {
                    const float v = f(16 * tile + (lane & 15), kb, lane >> 4, j);
                    const uint16_t p0 = bf16_rne(v);
                    const float r1 = v - bf16_f32(p0);
                    const uint16_t p1 = bf16_rne(r1);
                    const uint16_t p2 = bf16_rne(r1 - bf16_f32(p1));
                    const size_t e = (((size_t)tile * n_kb + kb) * 3) * 64 + lane;
                    img[e * 8 + j] = p0;
                    img[(e + 64) * 8 + j] = p1;
                    img[(e + 128) * 8 + j] = p2;
                }
}

}  // namespace

int nsnp_pileup_pack_weights_bf16(nsnp_ctx* ctx, const float* const* w)
{
    PileupWeightsB3& pw = ctx->pwb3;
    const size_t n_hh = (size_t)16 * 2 * 3 * 64 * 8, n_ih0 = (size_t)16 * 1 * 3 * 64 * 8, n_ih1 = (size_t)16 * 4 * 3 * 64 * 8;
    const size_t n_proj = (size_t)8 * 4 * 3 * 64 * 8, n_dense = (size_t)16 * 4 * 3 * 64 * 8, n_head = (size_t)2 * 8 * 3 * 64 * 8;
    const size_t total = 2 * (n_hh + n_ih0 + n_ih1 + n_hh) + n_proj + n_dense + n_head;
    std::vector<uint16_t> host(total);
    size_t off = 0;
    auto take = [&](size_t n) { uint16_t* p = host.data() + off; off += n; return p; };
    uint16_t *l0_hh[2], *l0_ih[2], *l1_ih[2], *l1_hh[2];
    for (int d = 0; d < 2; ++d) { l0_hh[d] = take(n_hh); l0_ih[d] = take(n_ih0); l1_ih[d] = take(n_ih1); l1_hh[d] = take(n_hh); }
    uint16_t* proj = take(n_proj); uint16_t* dense = take(n_dense); uint16_t* head = take(n_head);
    // layer-0 exchange position p = 16 w + 4 q + u <-> unit 16 w + 4 u + q; layer-1 exchange position p = 8 w + 2 q + u <-> unit 4 (2 w + u) + q
    auto l0_unit = [](int p) { return 16 * (p >> 4) + 4 * (p & 3) + ((p >> 2) & 3); };
    auto l1_unit = [](int p) { return 4 * (2 * (p >> 3) + (p & 1)) + ((p >> 1) & 3); };
    auto acc_feat = [](int kb, int q, int j) { return 16 * (2 * kb + (j >> 2)) + 4 * q + (j & 3); };
    auto gs = [](int row) { return nsnp_cell::lstm_gate_scale(row); };          // image row r: gate r & 3
    for (int d = 0; d < 2; ++d) {
        const float* const* l0 = w + d * 4;
        const float* const* l1 = w + 8 + d * 4;
        pack_b3(l0_hh[d], 16, 2, [&](int row, int kb, int q, int j) { return gs(row) * l0[1][gate_row(row) * PH + l0_unit(32 * kb + 8 * q + j)]; });
        pack_b3(l0_ih[d], 16, 1, [&](int row, int, int q, int j) {
            const int tr = gate_row(row), k = 8 * q + j;
            if (k < PC) return gs(row) * l0[0][tr * PC + k];
            if (k == PC) return gs(row) * (l0[2][tr] + l0[3][tr]);                  // the constant-1 column: b_ih + b_hh
            return 0.f;
        });
        // h0_t row of layer 1: [direction][64 layer-0 exchange positions]
        pack_b3(l1_ih[d], 16, 4, [&](int row, int kb, int q, int j) {
            return gs(row) * l1[0][gate_row(row) * 2 * PH + (kb >> 1) * PH + l0_unit(32 * (kb & 1) + 8 * q + j)]; });
        pack_b3(l1_hh[d], 16, 2, [&](int row, int kb, int q, int j) { return gs(row) * l1[1][gate_row(row) * PH + l1_unit(32 * kb + 8 * q + j)]; });
    }
    pack_b3(proj, 8, 4, [&](int row, int kb, int q, int j) { return w[16][row * 128 + 32 * kb + 8 * q + j]; });
    pack_b3(dense, 16, 4, [&](int row, int kb, int q, int j) { return w[18][row * 128 + acc_feat(kb, q, j)]; });
    pack_b3(head, 2, 8, [&](int row, int kb, int q, int j) {
        const int f = acc_feat(kb, q, j);
        if (row < 21) return w[20][row * 256 + f];
        if (row < 24) return w[22][(row - 21) * 256 + f];
        return 0.f;
    });
    // fp32 biases of the heads in accumulator order: [tile][q][g] = row 16 tile + 4 q + g
    float hb[128 + 256 + 32];
    for (int r = 0; r < 128; ++r) hb[r] = w[17][r];
    for (int r = 0; r < 256; ++r) hb[128 + r] = w[19][r];
    for (int r = 0; r < 32; ++r) hb[384 + r] = r < 21 ? w[21][r] : (r < 24 ? w[23][r - 21] : 0.f);
    const size_t wbytes = total * sizeof(uint16_t), bytes = wbytes + sizeof hb;
    if (pw.arena && pw.arena_bytes != bytes) { (void)hipFree(pw.arena); pw.arena = nullptr; }
    if (!pw.arena) { NSNP_HIP(ctx, hipMalloc((void**)&pw.arena, bytes)); pw.arena_bytes = bytes; }
    NSNP_HIP(ctx, hipMemcpy(pw.arena, host.data(), wbytes, hipMemcpyHostToDevice));
    NSNP_HIP(ctx, hipMemcpy((char*)pw.arena + wbytes, hb, sizeof hb, hipMemcpyHostToDevice));
    auto dev = [&](const uint16_t* hp) { return (void*)((char*)pw.arena + (hp - host.data()) * sizeof(uint16_t)); };
    for (int d = 0; d < 2; ++d) { pw.l0_whh[d] = dev(l0_hh[d]); pw.l0_wih[d] = dev(l0_ih[d]); pw.l1_wih[d] = dev(l1_ih[d]); pw.l1_whh[d] = dev(l1_hh[d]); }
    pw.proj_w = dev(proj); pw.dense_w = dev(dense); pw.head_w = dev(head);
    pw.proj_b = reinterpret_cast<float*>((char*)pw.arena + wbytes); pw.dense_b = pw.proj_b + 128; pw.head_b = pw.proj_b + 384;
    pw.loaded = true;
    return NSNP_OK;
}

static int set_lds_attr_b3(nsnp_ctx* ctx)
{
    if (ctx->attr_set_b3) return NSNP_OK;
#define SET(K, B) NSNP_HIP(ctx, hipFuncSetAttribute((const void*)K, hipFuncAttributeMaxDynamicSharedMemorySize, B))
    SET(k_pileup_l0_b3<1>, b3_l0_lds_bytes(1)); SET(k_pileup_l0_b3<2>, b3_l0_lds_bytes(2)); SET(k_pileup_l0_b3<4>, b3_l0_lds_bytes(4));
    SET(k_pileup_l1_b3<1>, b3_l1_lds_bytes(1)); SET(k_pileup_l1_b3<2>, b3_l1_lds_bytes(2)); SET(k_pileup_l1_b3<4>, b3_l1_lds_bytes(4));
    SET(k_pileup_l0_b3p, b3_l0_lds_bytes(2));
    SET(k_pileup_head_b3, HB3_STAGE_B8 * 16);
#undef SET
    ctx->attr_set_b3 = true;
    return NSNP_OK;
}

int nsnp_pileup_forward_bf16x3(nsnp_ctx* ctx, const int32_t* x, const int64_t* center_idx, int64_t N, float* gt, float* zy,
                               const PostOut* post, hipStream_t s)
{
    if (!ctx->pwb3.loaded || !ctx->pw.loaded) return NSNP_ENOWEIGHTS;
    if (N == 0) return NSNP_OK;
    int rc = set_lds_attr_b3(ctx);
    if (rc) return rc;
    if (!ctx->ws_h0 || ctx->ws_h0_floats < 192) { rc = nsnp_ctx_reserve(ctx, ctx->chunk_sites); if (rc) return rc; }
    const PileupWeightsB3& pw = ctx->pwb3;
    const PileupWeightsDev& p32 = ctx->pw;            // the fp32 path's layer-1 bias image (accumulator layout, gate rows scaled alike)
    __bf16* H0 = reinterpret_cast<__bf16*>(ctx->ws_h0);            // 768 B per site and step (nsnp_ctx_reserve sizes for it)
    for (int64_t base = 0; base < N; base += ctx->chunk_sites) {
        const int64_t n = (N - base < ctx->chunk_sites) ? N - base : ctx->chunk_sites;
        const int32_t* xc = center_idx ? x : x + base * (PW * PC);
        const int64_t* cc = center_idx ? center_idx + base : nullptr;
        {
            ScopedKernelTimer tm(ctx, NSNP_K_L0, s);
            // 32 sites per workgroup (measured best: 64 lose to register pressure) when that gives every CU a workgroup, else 16
            int nsg = NSNP_CDIV(n, 32) * 2 >= (int64_t)ctx->n_cu ? 2 : 1;
            if (ctx->l0_rs_groups) nsg = ctx->l0_rs_groups;
#define LAUNCH_L0(G) hipLaunchKernelGGL(k_pileup_l0_b3<G>, dim3((unsigned)NSNP_CDIV(n, 16 * G), 2), dim3(256), b3_l0_lds_bytes(G), s, xc, cc, n, \
            (const __bf16*)pw.l0_whh[0], (const __bf16*)pw.l0_whh[1], (const __bf16*)pw.l0_wih[0], (const __bf16*)pw.l0_wih[1], H0)
            // 32 sites per workgroup run the skewed two-group kernel ("l0_register_stationary" 0 selects the plain one: A/B, tests)
            if (nsg == 2 && ctx->l0_rs)
                hipLaunchKernelGGL(k_pileup_l0_b3p, dim3((unsigned)NSNP_CDIV(n, 32), 2), dim3(256), b3_l0_lds_bytes(2), s, xc, cc, n,
                                   (const __bf16*)pw.l0_whh[0], (const __bf16*)pw.l0_whh[1], (const __bf16*)pw.l0_wih[0], (const __bf16*)pw.l0_wih[1], H0);
            else if (nsg == 4) LAUNCH_L0(4); else if (nsg == 2) LAUNCH_L0(2); else LAUNCH_L0(1);
#undef LAUNCH_L0
        }
        {
            ScopedKernelTimer tm(ctx, NSNP_K_L1, s);
            // one 8-wave workgroup per CU (registers): 64 sites each when that still gives every CU one, else 32 or 16
            int g1 = 4;
            while (g1 > 1 && NSNP_CDIV(n, 16 * g1) * 2 < (int64_t)ctx->n_cu) g1 >>= 1;
            if (ctx->l1_rs_groups) g1 = ctx->l1_rs_groups;
#define LAUNCH_L1(G) hipLaunchKernelGGL(k_pileup_l1_b3<G>, dim3((unsigned)NSNP_CDIV(n, 16 * G), 2), dim3(512), b3_l1_lds_bytes(G), s, H0, n, \
            (const __bf16*)pw.l1_wih[0], (const __bf16*)pw.l1_wih[1], (const __bf16*)pw.l1_whh[0], (const __bf16*)pw.l1_whh[1], \
            p32.l1_bias[0], p32.l1_bias[1], ctx->ws_h1c)
            if (g1 == 4) LAUNCH_L1(4); else if (g1 == 2) LAUNCH_L1(2); else LAUNCH_L1(1);
#undef LAUNCH_L1
        }
        ScopedKernelTimer tm_head(ctx, NSNP_K_HEAD, s);
        const bool po = post && post->gt_arg;
#ifndef NSNP_B3_NOHEAD                                  /* (removal timing build) */
        hipLaunchKernelGGL(k_pileup_head_b3, dim3((unsigned)NSNP_CDIV(n, 16 * HB3_WAVES)), dim3(64 * HB3_WAVES), HB3_STAGE_B8 * 16, s,
                           ctx->ws_h1c, n, (const __bf16*)pw.proj_w, pw.proj_b, (const __bf16*)pw.dense_w, pw.dense_b,
                           (const __bf16*)pw.head_w, pw.head_b, gt + base * NSNP_GT_CLASSES, zy + base * NSNP_ZY_CLASSES,
                           po ? post->gt_arg + base : nullptr, po ? post->zy_arg + base : nullptr,
                           po ? post->gt_max + base : nullptr, po ? post->zy_max + base : nullptr);
#endif
    }
    NSNP_HIP(ctx, hipGetLastError());
    return NSNP_OK;
}
