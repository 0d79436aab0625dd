// hap_gemm.hpp -- the tile-image GEMM kernel shared by the HaplotypeModel forward (hap_forward.hip) and the
// legacy CatModel forward (cat_forward.hip): 128 weight rows x 128 "sites" per workgroup, K streamed through
// LDS in 16-wide chunks (every operand tile is an 8 KB image [128][16]), epilogues: LSTM cell, linear,
// linear+tanh, linear+ReLU.  With CONV the B operand of the in0 block is gathered as an implicit 3x3 / 1x1
// convolution window over a pixel image (sites = pixels, chunks = (tap, channel chunk)).
#pragma once
#include "nsnp_devclock.hpp"
#include "nsnp_common.hpp"
#include "nsnp_lstm_cell.hpp"

// main-loop pipeline shape (see k_hap_gemm); overridable for A/B builds (tools/build_variant.sh)
#ifndef NSNP_GEMM_PIPE
#define NSNP_GEMM_PIPE 2
#endif
#ifndef NSNP_GEMM_LDPOS
#define NSNP_GEMM_LDPOS 4
#endif
#ifndef NSNP_GEMM_WRPOS
#define NSNP_GEMM_WRPOS 1
#endif
#ifndef NSNP_GEMM_BIASINIT
#define NSNP_GEMM_BIASINIT 1     // accumulators start at the bias (0: the epilogue loads and adds it)
#endif
#ifndef NSNP_GEMM_CEARLY
#define NSNP_GEMM_CEARLY 0       // 1: the cell state is loaded from inside the last bursts (measured: -1 %); 0: in the epilogue
#endif
#ifndef NSNP_GEMM_CELL
#define NSNP_GEMM_CELL 1         // LSTM cell with shared reciprocals (0: sigmoid / tanh one by one)
#endif
#ifndef NSNP_GEMM_MINW
#define NSNP_GEMM_MINW 2
#endif
#ifndef NSNP_GEMM_B3X
#define NSNP_GEMM_B3X 1          // bf16x3 LSTM steps on 256 x 256 workgroup tiles (k_hap_lstm_b3x) where the launch fills the chip with them
#endif

namespace {

constexpr float LOG2E = 1.4426950408889634f;
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float sigmoid_f(float x) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-LOG2E * x)); }
__device__ __forceinline__ float tanh_f(float x) { return __builtin_fmaf(-2.0f, __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f((2.0f * LOG2E) * x)), 1.0f); }

// LSTM cell and the gate-scale contract of the packers: nsnp_lstm_cell.hpp (one definition for every recurrence kernel)
using nsnp_cell::lstm_cell;
using nsnp_cell::lstm_gate_scale;

constexpr int TS = 128;        // sites per workgroup tile
constexpr int TR = 128;        // weight rows per workgroup tile
constexpr int BK = 16;         // K chunk
constexpr int LDK = BK + 4;    // padded LDS row (80 B: conflict-free 16-byte reads across 16 lanes)
constexpr int TILE_F = TS * BK;   // floats in one [128][16] tile image (8 KB)

// position p inside a 16-unit chunk of the LSTM storage order <-> hidden unit within the chunk
// (p = h*8 + 4*rt + r4  <->  unit = 8*rt + 2*r4 + h : what a lane half h owns after the MFMAs)
__host__ __device__ inline int unit_of_pos(int p) { return 8 * ((p >> 2) & 1) + 2 * (p & 3) + (p >> 3); }

struct StepArgs {
    // per z-slice (direction / encoder) description of one fused step
    const float* w;        // weight images  [row_tiles][n_chunks][128][16]
    const float* bias;     // [rows] in image row order
    const float* in0;      // input tile images for this step: [site_tiles][nk0][128][16] (may be NULL when nk0 == 0)
    const float* in1;      // second input block (h_{t-1} in LSTM mode): [site_tiles][nk1][128][16]
    float* out;            // LSTM: h_t images, 16 chunks per site tile; linear: rows/16 chunks per site tile
    float* cstate;         // LSTM: c images, 16 chunks per site tile
    int nk0, nk1;          // chunks taken from in0 / in1 (nk1 == 0 on the first step: h = 0)
    int nk_img;            // chunks per row tile in the weight image (>= nk0 + nk1)
    int in0_tile_stride;   // floats between consecutive site tiles of in0
    int in1_tile_stride;
    int out_tile_stride;   // floats between consecutive site tiles of out
    int c_tile_stride;     // ... of cstate
    int first;             // LSTM: 1 on the first step (c = 0)
    // CONV only: in0 is a pixel image [pixel_tiles][cc0][128][16] of n_pix pixels = images of conv_h x conv_w;
    // its nk0 = 9 * cc0 chunks are (tap, channel chunk) with tap = 3*(dy+1) + (dx+1), zero padding outside the
    // image.  in1 (if any) is read at the pixel itself (the 1x1 shortcut).  Only rows < n_rows are written.
    int conv_h, conv_w, cc0_shift, n_rows;
    long long n_pix;
};
struct StepLaunch { StepArgs z[4]; };

enum { MODE_LSTM = 0, MODE_LINEAR = 1, MODE_LINEAR_TANH = 2, MODE_LINEAR_RELU = 3 };

// Dynamic-LDS ballast for a launch of `wgs` workgroups of k_hap_gemm.  The kernel's 40 KB of static LDS let four workgroups share
// a CU, and when the grid is smaller than that capacity the dispatcher fills the CUs unevenly (512 workgroups on 256 CUs: some hold
// three or four, others one or none), so the launch lasts as long as its fullest CU: 138 instead of 153 TFLOP/s for the bare
// MFMA loop (tools/probes/gemm_loop_rate.hip).  The ballast makes ceil(wgs / CUs) the most a CU can take, which spreads the grid.
inline unsigned gemm_lds_ballast(long long wgs, int n_cu)
{
    constexpr unsigned kStatic = 2u * 2u * TR * LDK * sizeof(float), kCu = 160u * 1024u;
    long long per = n_cu > 0 ? (wgs + n_cu - 1) / n_cu : 4;
    if (per < 2) per = 2;
    if (per >= 4) return 0u;
    const unsigned b = kCu / (unsigned)(per + 1) - kStatic + 512u;          // per + 1 workgroups no longer fit ...
    return (unsigned long long)per * (kStatic + b) <= kCu ? b : 0u;          // ... per still do
}

// F16 = false: operands are fp32, one element per 4 bytes, v_mfma_f32_32x32x2_f32 (exact fp32).
// F16 = true : "f16x3" - every 4-byte element is a (hi, lo) fp16 pair (hi = fp16(v), lo = fp16(v - hi)); a row
//              of a tile image holds its 16 hi halves followed by its 16 lo halves, so the two 16-byte LDS reads
//              of a lane are its hi and lo fragments of v_mfma_f32_32x32x16_f16, and a product is three MFMAs
//              (hi.hi + lo.hi + hi.lo) with fp32 accumulation.  Same image sizes, strides and launch sequence.
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void split_h(float v, _Float16& hi, _Float16& lo) { hi = (_Float16)v; lo = (_Float16)(v - (float)hi); }
// for caller-supplied inputs: both halves saturate at the fp16 maximum instead of becoming infinite, so |v| <= 131008
// is carried and anything larger acts as +-131008 (the fp32 mode has no such limit)
__device__ __forceinline__ void split_sat(float v, _Float16& hi, _Float16& lo)
{
    hi = (_Float16)__builtin_amdgcn_fmed3f(v, -65504.0f, 65504.0f);
    lo = (_Float16)__builtin_amdgcn_fmed3f(v - (float)hi, -65504.0f, 65504.0f);
}

// AR = 2 : "bf16x3" - weights are stored as three bf16 planes per element (a row of a weight tile image = 16 p0 | 16 p1 | 16 p2 bf16 =
//          96 bytes, v = p0 + p1 + p2 with p0 = bf16(v), p1 = bf16(v - p0), p2 = bf16(v - p0 - p1): the full 24-bit significand and the
//          fp32 exponent range); activations are fp32 in memory and split the same way on their way into LDS (the CatModel's launches),
//          or arrive as the same three planes, written by the epilogue that produced them (BS / OS below: the HaplotypeModel's launches);
//          a product is the six v_mfma_f32_32x32x16_bf16 whose dropped terms are below 2^-24 of it, fp32 accumulation.
//          Same launch sequence, activation images and epilogues as AR = 0; weight tile images are 12 KB instead of 8.
typedef __bf16 b8_t __attribute__((ext_vector_type(8)));
constexpr int LDB3 = 48;          // bf16 per LDS row in the bf16x3 mode: 3 planes x 16, NO padding (96 B = 6 slots of 16 B; 48 KB per workgroup, three
                                  // per CU).  Slot s of row r is stored at s ^ ((r >> 3) & 1): the 16 rows a ds_read_b128 lane group touches
                                  // (r mod 16 all different) then fall into 16 distinct bank slots (6 r + s mod 16 pairs r with r + 8; the flipped
                                  // low bit separates the two, and 6 k is never +-1 mod 16)
__device__ __forceinline__ void split3_b8(const f32x4& lo, const f32x4& hi, b8_t& p0, b8_t& p1, b8_t& p2)
{
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float v = j < 4 ? lo[j & 3] : hi[j & 3];
        const __bf16 a = (__bf16)v;
        const float r1 = v - (float)a;
        const __bf16 b = (__bf16)r1;
        p0[j] = a; p1[j] = b; p2[j] = (__bf16)(r1 - (float)b);
    }
}

typedef __bf16 b4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void split3_b4(const f32x4& v, b4_t& p0, b4_t& p1, b4_t& p2)
{
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const __bf16 a = (__bf16)v[j];
        const float r1 = v[j] - (float)a;
        const __bf16 b = (__bf16)r1;
        p0[j] = a; p1[j] = b; p2[j] = (__bf16)(r1 - (float)b);
    }
}
constexpr int TILE_F3 = TILE_F * 3 / 2;       // floats of a [128][16] tile image whose elements are three bf16 planes (rows of 96 bytes)
constexpr int ROW_F3 = BK * 3 / 2;            // floats of one of its rows

// BS (bf16x3 only): the ACTIVATION images hold three bf16 planes per element as well - a row of an activation tile image is 16 p0 | 16 p1 |
//      16 p2 like a weight row, written that way by the epilogue that produced it (OS: this launch writes its output so) - and a K chunk
//      is loads + MFMAs only.  Before, every workgroup that consumed an activation split it again on its way into LDS: the h_t of a step
//      was split by each of the 4-8 row-tile workgroups of the next step and of the next layer, 2.7 vector instructions per MFMA
//      (profiles/r05_hap_forward_bf16x3_sq_counters.json).  Same planes (the split is a function of the fp32 value), same products, same
//      order: bit-identical results.  6 instead of 4 bytes per activation element in memory.
template <int MODE, int AR, bool CONV = false, bool BS = false, bool OS = false>
__global__ __launch_bounds__(256, NSNP_GEMM_MINW) void k_hap_gemm(const StepLaunch L)
{
    constexpr bool F16 = AR == 1, B3 = AR == 2;
    static_assert(!(BS || OS) || (B3 && !CONV), "pre-split activations exist in the bf16x3 mode of the plain GEMM only");
    constexpr int LDS_ROW_F = B3 ? LDB3 / 2 : LDK;            // floats per LDS row
    constexpr int TILE_W = B3 ? TILE_F * 3 / 2 : TILE_F;      // floats per weight tile image
    constexpr int TILE_B = BS ? TILE_F3 : TILE_F;             // floats per input activation tile image
    constexpr int TILE_O = OS ? TILE_F3 : TILE_F;             // floats per output activation tile image
    __shared__ float As[2][TR][LDS_ROW_F];
    __shared__ float Bs[2][TS][LDS_ROW_F];
    NSNP_DEVCLK_START
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    // grid: x = site tile, y = row tile, z = slice (direction / encoder).  (A persistent variant - workgroups looping over the
    // tiles of the launch - was measured: it needs more registers than four waves per SIMD leave and gained nothing at equal
    // occupancy; pass sizes from 8192 to 65536 sites give the same rate, so neither dispatch turnaround nor launch ramps cost.)
    const int bx = blockIdx.x, by = blockIdx.y;
    const StepArgs& a = L.z[blockIdx.z];
    // Wave mapping: 2 x 2 waves of 64 rows x 64 sites; CONV launches with at most 64 output rows (C_out 32 / 64) use
    // 1 x 4 waves of 64 rows x 32 sites instead, so that no wave multiplies padding rows (and one row tile at C_out <= 32).
    const bool small_rows = CONV && a.n_rows <= 64;
    const int wr = small_rows ? 0 : wave >> 1, wc = wave & 1;
    const int nct = small_rows ? 1 : 2;
    const int nrt = (CONV && a.n_rows <= 32) ? 1 : 2;
    const int site0 = small_rows ? 32 * wave : 64 * wc;              // first site of the wave's tile(s)
    const int nk = a.nk0 + a.nk1;

    const float* __restrict__ wt = a.w + (size_t)by * a.nk_img * TILE_W;
    const float* __restrict__ b0 = a.in0 ? a.in0 + (size_t)bx * a.in0_tile_stride : nullptr;
    const float* __restrict__ b1 = a.in1 ? a.in1 + (size_t)bx * a.in1_tile_stride : nullptr;

    // each thread moves two 16-byte pieces of each 8 KB tile image: row = tid/2, quarter = (tid&1)*2 + {0,1}
    const int crow = tid >> 1, cq = (tid & 1) * 2;
    long long pix = 0; int py = 0, px = 0; bool pvalid = true;
    if (CONV) {
        pix = (long long)bx * TS + crow;
        pvalid = pix < a.n_pix;
        const int rem = (int)(pix % (a.conv_h * a.conv_w));
        py = rem / a.conv_w; px = rem - py * a.conv_w;
    }
    f32x4 ga2;                                     // bf16x3: the third 16-byte piece of the thread's half of a 96-byte weight row
    f32x4 gb2;                                     // BS: likewise of an activation row
    auto gload = [&](int kc, f32x4& a0, f32x4& a1, f32x4& bb0, f32x4& bb1) {
        if (B3) {
            const f32x4* pa = reinterpret_cast<const f32x4*>(wt + (size_t)kc * TILE_W) + crow * 6 + (tid & 1) * 3;
            a0 = pa[0]; a1 = pa[1]; ga2 = pa[2];
        } else {
            const f32x4* pa = reinterpret_cast<const f32x4*>(wt + (size_t)kc * TILE_F) + crow * 4 + cq;
            a0 = pa[0]; a1 = pa[1];
        }
        if (CONV && kc < a.nk0) {
            const int tap = kc >> a.cc0_shift, cc = kc & ((1 << a.cc0_shift) - 1);
            const int ty = tap / 3, dy = ty - 1, dx = tap - 3 * ty - 1;
            const int yy = py + dy, xx = px + dx;
            if (pvalid && yy >= 0 && yy < a.conv_h && xx >= 0 && xx < a.conv_w) {
                const long long ps = pix + dy * a.conv_w + dx;
                const f32x4* pb = reinterpret_cast<const f32x4*>(a.in0 + (size_t)(((ps >> 7) << a.cc0_shift) + cc) * TILE_F +
                                                                 (size_t)(ps & 127) * BK) + cq;
                bb0 = pb[0]; bb1 = pb[1];
            } else {
                bb0 = f32x4{0.f, 0.f, 0.f, 0.f}; bb1 = bb0;
            }
            return;
        }
        const float* src = kc < a.nk0 ? b0 + (size_t)kc * TILE_B : b1 + (size_t)(kc - a.nk0) * TILE_B;
        if (BS) {
            const f32x4* pb = reinterpret_cast<const f32x4*>(src) + crow * 6 + (tid & 1) * 3;
            bb0 = pb[0]; bb1 = pb[1]; gb2 = pb[2];
            return;
        }
        const f32x4* pb = reinterpret_cast<const f32x4*>(src) + crow * 4 + cq;
        bb0 = pb[0]; bb1 = pb[1];
    };
    auto lstore = [&](int buf, const f32x4& a0, const f32x4& a1, const f32x4& bb0, const f32x4& bb1) {
        if (B3) {
            const int sw = (crow >> 3) & 1, h3 = (tid & 1) * 3;
            f32x4* ra = reinterpret_cast<f32x4*>(&As[buf][crow][0]);
            ra[(h3 + 0) ^ sw] = a0; ra[(h3 + 1) ^ sw] = a1; ra[(h3 + 2) ^ sw] = ga2;
            if (BS) {                              // the activation row arrives as its three planes: a plain copy, like the weights
                f32x4* rbs = reinterpret_cast<f32x4*>(&Bs[buf][crow][0]);
                rbs[(h3 + 0) ^ sw] = bb0; rbs[(h3 + 1) ^ sw] = bb1; rbs[(h3 + 2) ^ sw] = gb2;
                return;
            }
            // the thread's 8 activations (K positions 8 (tid & 1) ..) -> its 16-byte piece of each of the three planes
            b8_t p0, p1, p2;
            split3_b8(bb0, bb1, p0, p1, p2);
            b8_t* rb = reinterpret_cast<b8_t*>(&Bs[buf][crow][0]);
            const int hb = (tid & 1) ^ sw;
            rb[hb] = p0; rb[2 + hb] = p1; rb[4 + hb] = p2;
            return;
        }
        *reinterpret_cast<f32x4*>(&As[buf][crow][cq * 4]) = a0;
        *reinterpret_cast<f32x4*>(&As[buf][crow][cq * 4 + 4]) = a1;
        *reinterpret_cast<f32x4*>(&Bs[buf][crow][cq * 4]) = bb0;
        *reinterpret_cast<f32x4*>(&Bs[buf][crow][cq * 4 + 4]) = bb1;
    };

    // accumulators start at the bias of their rows (register r = 4*r4 + g of tile rt: row 32*rt + 8*r4 + 4*lh + g of the wave's
    // 64 rows), so the epilogue neither loads it nor adds it: the sum is b + w0 x0 + w1 x1 + ... in the k order of the images
    f32x16 acc[2][2];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
            const int f = 128 * by + 64 * wr + 32 * rt + 8 * r4 + 4 * lh;
            f32x4 bz = f32x4{0.f, 0.f, 0.f, 0.f};
            if (NSNP_GEMM_BIASINIT && (!CONV || f < a.n_rows)) bz = *reinterpret_cast<const f32x4*>(a.bias + f);
#pragma unroll
            for (int g = 0; g < 4; ++g) { acc[rt][0][4 * r4 + g] = bz[g]; acc[rt][1][4 * r4 + g] = bz[g]; }
        }

    // ---- main loop: a software pipeline over the K chunks --------------------------------------------------------------
    // What a wave does between the barrier that releases chunk kc and its first MFMA is on the critical path of its SIMD (the
    // co-resident waves run the same program and tend to wait at the same time), so only the LDS reads of the first four K-steps
    // stand there.  Everything else is issued from inside the burst of 32 MFMAs (sched_barrier fences keep it there):
    //   PIPE 1  global loads of chunk kc+1 behind MFMA group LDPOS, written to LDS after the burst (one chunk ahead)
    //   PIPE 2  global loads of chunk kc+2 behind MFMA group LDPOS, written to LDS behind group WRPOS of the NEXT iteration
    //           (two chunks ahead with ONE set of staging registers: the write at WRPOS frees them before the load at LDPOS
    //           refills them; the buffer written during iteration kc+1 was last read at the start of iteration kc)
    // Measured on the HaplotypeModel forward (N = 32768): loads ahead of the LDS reads (round 2) 76.8 ms, LDPOS 2 72.9, LDPOS 6 +
    // fragment reads split 69.2.  The accumulation order is untouched: results are bit-identical.
    constexpr int PIPE = NSNP_GEMM_PIPE, LDPOS = NSNP_GEMM_LDPOS, WRPOS = NSNP_GEMM_WRPOS;
    static_assert(PIPE == 1 || (PIPE == 2 && WRPOS < LDPOS), "two-ahead needs the write in front of the load");
    const int pfd = PIPE == 1 ? 1 : 2;
    f32x4 ga0, ga1, gb0, gb1;
    gload(0, ga0, ga1, gb0, gb1);
    lstore(0, ga0, ga1, gb0, gb1);
    if (PIPE == 2 && nk > 1) gload(1, ga0, ga1, gb0, gb1);
    __syncthreads();
    for (int kc = 0; kc < nk; ++kc) {
        const int cur = kc & 1;
        // fragments: lane (li, lh) takes floats [lh*8, lh*8+8) of its row = 8 k-steps of 32x32x2; the four fragments of
        // K-steps 0..3 first (the burst starts when they are there), the other four from inside the burst
        f32x4 af[2][B3 ? 3 : 2], bf[2][B3 ? 3 : 2];
        auto read_frags3 = [&](int p) {             // bf16x3: plane p of the lane's rows / sites (8 bf16 at K positions 8 lh ..), swizzled slot
            const int sl = (2 * p + (lh ^ ((li >> 3) & 1))) * 4;
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
                af[rt][p] = *reinterpret_cast<const f32x4*>(&As[cur][64 * wr + 32 * rt + li][sl]);
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
                bf[ct][p] = *reinterpret_cast<const f32x4*>(&Bs[cur][site0 + 32 * ct + li][sl]);
        };
        auto read_frags = [&](int h) {
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
                af[rt][h] = *reinterpret_cast<const f32x4*>(&As[cur][64 * wr + 32 * rt + li][(F16 ? lh * 4 : lh * 8) + h * (F16 ? 8 : 4)]);
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
                bf[ct][h] = *reinterpret_cast<const f32x4*>(&Bs[cur][site0 + 32 * ct + li][(F16 ? lh * 4 : lh * 8) + h * (F16 ? 8 : 4)]);
        };
        if (B3) {
            read_frags3(0); read_frags3(2);          // the first product is w0 . x2
            __builtin_amdgcn_sched_barrier(0);
            read_frags3(1);
            __builtin_amdgcn_sched_barrier(0);
        } else {
        read_frags(0);
        __builtin_amdgcn_sched_barrier(0);
        read_frags(1);
        __builtin_amdgcn_sched_barrier(0);
        }
        auto side_work = [&](int pos) {
            if (PIPE == 2 && pos == WRPOS && kc + 1 < nk) {
                __builtin_amdgcn_sched_barrier(0);
                lstore(cur ^ 1, ga0, ga1, gb0, gb1);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (pos == LDPOS && kc + pfd < nk) {
                __builtin_amdgcn_sched_barrier(0);
                gload(kc + pfd, ga0, ga1, gb0, gb1);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (NSNP_GEMM_CEARLY && MODE == MODE_LSTM && pos == LDPOS && kc + pfd == nk && !a.first) {
                // the staging registers are free from here on: the cell state of the lane's 2 x 8 units rides in on them
                __builtin_amdgcn_sched_barrier(0);
                const float* c0 = a.cstate + (size_t)bx * a.c_tile_stride + (size_t)(2 * by + wr) * TILE_F + (64 * wc + li) * BK + lh * 8;
                ga0 = *reinterpret_cast<const f32x4*>(c0); ga1 = *reinterpret_cast<const f32x4*>(c0 + 4);
                gb0 = *reinterpret_cast<const f32x4*>(c0 + 32 * BK); gb1 = *reinterpret_cast<const f32x4*>(c0 + 32 * BK + 4);
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        if (B3) {
            // six products, smallest first: (w plane, x plane) = (0,2) (1,1) (2,0) (0,1) (1,0) (0,0)
            constexpr int WP[6] = {0, 1, 2, 0, 1, 0}, XP[6] = {2, 1, 0, 1, 0, 0};
#pragma unroll
            for (int g = 0; g < 6; ++g) {
                side_work(g == 0 ? -1 : (g == 1 ? WRPOS : (g == 3 ? LDPOS : -1)));
#pragma unroll
                for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                    for (int ct = 0; ct < 2; ++ct) {
                        if (CONV && (ct >= nct || rt >= nrt)) continue;
                        acc[rt][ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(b8_t, af[rt][WP[g]]), __builtin_bit_cast(b8_t, bf[ct][XP[g]]),
                                                                               acc[rt][ct], 0, 0, 0);
                    }
            }
        } else if (F16) {
            // af[rt][0] / [1] are the lane's 8 hi / 8 lo halves of row (rt), likewise bf for the site
#pragma unroll
            for (int term = 0; term < 3; ++term) {
                side_work(term == 0 ? -1 : (term == 1 ? WRPOS : LDPOS));
#pragma unroll
                for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                    for (int ct = 0; ct < 2; ++ct) {
                        if (CONV && (ct >= nct || rt >= nrt)) continue;
                        const h8 av = __builtin_bit_cast(h8, af[rt][term == 1 ? 1 : 0]);
                        const h8 bv = __builtin_bit_cast(h8, bf[ct][term == 2 ? 1 : 0]);
                        acc[rt][ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, bv, acc[rt][ct], 0, 0, 0);
                    }
            }
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                side_work(j);
#pragma unroll
                for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                    for (int ct = 0; ct < 2; ++ct) {
                        if (CONV && (ct >= nct || rt >= nrt)) continue;
                        acc[rt][ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[rt][j >> 2][j & 3], bf[ct][j >> 2][j & 3],
                                                                            acc[rt][ct], 0, 0, 0);
                    }
            }
        }
        side_work(8);
        if (PIPE == 1 && kc + 1 < nk) lstore(cur ^ 1, ga0, ga1, gb0, gb1);
        __syncthreads();
    }

    // ---- epilogue ------------------------------------------------------------------------------
    // accumulator register r = 4*r4 + g of tile (rt, ct), lane (li, lh):
    //   row within the 32-row tile = g + 8*r4 + 4*lh,  site = 64*wc + 32*ct + li
    if (MODE == MODE_LSTM) {
        // rows are [unit][gate]: unit8 = 2*r4 + lh, gate = g.  The wave's 16 units form chunk
        // kc = 2*by + wr of the output image; a lane writes positions p = lh*8 + 4*rt + r4.
        const size_t img = (size_t)(2 * by + wr) * TILE_F;
        if (a.first) { ga0 = f32x4{0.f, 0.f, 0.f, 0.f}; ga1 = ga0; gb0 = ga0; gb1 = ga0; }
        else if (!NSNP_GEMM_CEARLY) {
            const float* c0 = a.cstate + (size_t)bx * a.c_tile_stride + img + (64 * wc + li) * BK + lh * 8;
            ga0 = *reinterpret_cast<const f32x4*>(c0); ga1 = *reinterpret_cast<const f32x4*>(c0 + 4);
            gb0 = *reinterpret_cast<const f32x4*>(c0 + 32 * BK); gb1 = *reinterpret_cast<const f32x4*>(c0 + 32 * BK + 4);
        }
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
            const int site = 64 * wc + 32 * ct + li;
            float* cptr = a.cstate + (size_t)bx * a.c_tile_stride + img + site * BK + lh * 8;
            float* hptr = a.out + (size_t)bx * a.out_tile_stride + img + site * BK + lh * 8;
            f32x4 cv[2] = {ct ? gb0 : ga0, ct ? gb1 : ga1};
            f32x4 hv[2];
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    f32x4 z = f32x4{acc[rt][ct][4 * r4 + 0], acc[rt][ct][4 * r4 + 1], acc[rt][ct][4 * r4 + 2], acc[rt][ct][4 * r4 + 3]};
                    if (!NSNP_GEMM_BIASINIT) z += *reinterpret_cast<const f32x4*>(a.bias + 128 * by + 64 * wr + 32 * rt + 8 * r4 + 4 * lh);
                    float cn;
                    if (NSNP_GEMM_CELL) hv[rt][r4] = lstm_cell(z[0], z[1], z[2], z[3], cv[rt][r4], cn);
                    else {
                        // (A/B build of the plain cell: the gate rows are scaled for lstm_cell, undo it)
                        const float ui = z[0] * (-1.0f / LOG2E), uf = z[1] * (-1.0f / LOG2E), ug = z[2] * (-0.5f / LOG2E), uo = z[3] * (-1.0f / LOG2E);
                        cn = __builtin_fmaf(sigmoid_f(uf), cv[rt][r4], sigmoid_f(ui) * tanh_f(ug));
                        hv[rt][r4] = sigmoid_f(uo) * tanh_f(cn);
                    }
                    cv[rt][r4] = cn;
                }
            *reinterpret_cast<f32x4*>(cptr) = cv[0]; *reinterpret_cast<f32x4*>(cptr + 4) = cv[1];
            if (OS) {
                // the lane's 8 positions lh*8.. are 8 consecutive bf16 of each of the row's three planes
                b8_t p0, p1, p2;
                split3_b8(hv[0], hv[1], p0, p1, p2);
                b8_t* hrow = reinterpret_cast<b8_t*>(a.out + (size_t)bx * a.out_tile_stride + (size_t)(2 * by + wr) * TILE_O + site * ROW_F3);
                hrow[lh] = p0; hrow[2 + lh] = p1; hrow[4 + lh] = p2;
            } else if (F16) {
                // the lane's 8 positions lh*8.. are 8 consecutive halves of the hi block and of the lo block
                h8 hh, hl;
#pragma unroll
                for (int e = 0; e < 8; ++e) { _Float16 x, y; split_h(hv[e >> 2][e & 3], x, y); hh[e] = x; hl[e] = y; }
                _Float16* hrow = reinterpret_cast<_Float16*>(a.out + (size_t)bx * a.out_tile_stride + img + site * BK);
                *reinterpret_cast<h8*>(hrow + lh * 8) = hh;
                *reinterpret_cast<h8*>(hrow + 16 + lh * 8) = hl;
            } else {
                *reinterpret_cast<f32x4*>(hptr) = hv[0]; *reinterpret_cast<f32x4*>(hptr + 4) = hv[1];
            }
        }
    } else {
        // plain rows: feature f = 128*by + 64*wr + 32*rt + (g + 8*r4 + 4*lh); natural order in chunks of 16
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
            if (CONV && ct >= nct) continue;
            const int site = site0 + 32 * ct + li;
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    const int f = 128 * by + 64 * wr + 32 * rt + 8 * r4 + 4 * lh;
                    if (CONV && f >= a.n_rows) continue;
                    f32x4 v;
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const float t = acc[rt][ct][4 * r4 + g] + (NSNP_GEMM_BIASINIT ? 0.f : a.bias[f + g]);
                        v[g] = MODE == MODE_LINEAR_TANH ? tanh_f(t) : MODE == MODE_LINEAR_RELU ? fmaxf(t, 0.f) : t;
                    }
                    if (OS) {
                        b4_t p0, p1, p2;
                        split3_b4(v, p0, p1, p2);
                        __bf16* orow = reinterpret_cast<__bf16*>(a.out + (size_t)bx * a.out_tile_stride + (size_t)(f >> 4) * TILE_O + site * ROW_F3);
                        *reinterpret_cast<b4_t*>(orow + (f & 15)) = p0;
                        *reinterpret_cast<b4_t*>(orow + 16 + (f & 15)) = p1;
                        *reinterpret_cast<b4_t*>(orow + 32 + (f & 15)) = p2;
                        continue;
                    }
                    float* o = a.out + (size_t)bx * a.out_tile_stride + (size_t)(f >> 4) * TILE_F + site * BK;
                    if (F16) {
                        h4 vh, vl;
#pragma unroll
                        for (int g = 0; g < 4; ++g) { _Float16 x, y; split_h(v[g], x, y); vh[g] = x; vl[g] = y; }
                        _Float16* orow = reinterpret_cast<_Float16*>(o);
                        *reinterpret_cast<h4*>(orow + (f & 15)) = vh;
                        *reinterpret_cast<h4*>(orow + 16 + (f & 15)) = vl;
                    } else {
                        *reinterpret_cast<f32x4*>(o + (f & 15)) = v;
                    }
                }
        }
    }
    NSNP_DEVCLK_STOP(CONV ? 3 : 2)
}


// ---- bf16x3 LSTM step on 256 x 256 workgroup tiles -------------------------------------------------------------------------------
// In the bf16x3 mode a K chunk is 24 MFMAs per wave (768 matrix cycles) where the fp32 mode has 32 of four times the length: the
// chunk's fixed costs - 20 KB from L2, the three-way split of the activations, the barrier, the fragment reads - weigh 2.7 times
// as much, and k_hap_gemm<LSTM, 2> stands at 60 % MFMA-busy (tools/probes/gemm_b3_tile.hip reproduces it: 0.45-0.49 of the nominal
// peak for the 128 x 128 tile, 0.56-0.57 for this one).  Here a workgroup of eight waves owns two row tiles x two site tiles: wave
// (wr, wc) multiplies 64 rows (16 hidden units, all four gates) with the 128 sites of ONE site-tile image - 48 MFMAs per chunk and
// barrier, 32 instead of 40 KB from L2 per 2 x 2 tiles, one split of an activation per 256 instead of 128 rows.  Same images, same
// k order and product order per accumulator: bit-identical to k_hap_gemm<MODE_LSTM, 2>.  One workgroup per CU (96 KB of LDS).
constexpr int BT = 256;
template <bool BS>                 // BS: activations in and out as three bf16 planes (see k_hap_gemm)
__global__ __launch_bounds__(512, 1) void k_hap_lstm_b3x(const StepLaunch L)
{
    constexpr int TILE_B = BS ? TILE_F3 : TILE_F;
    constexpr int TILE_W = TILE_F * 3 / 2;
    __shared__ float As[2][BT][LDB3 / 2];
    __shared__ float Bs[2][BT][LDB3 / 2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int wr = wave >> 1, wc = wave & 1;
    const int bx = blockIdx.x, by = blockIdx.y;
    const StepArgs& a = L.z[blockIdx.z];
    const int nk = a.nk0 + a.nk1;

    // staging: the 256 x 96 B of the two weight tile images = 1536 pieces of 16 B, three per thread; the 256 x 64 B of the two
    // activation tile images = 512 halves of a site's 16 K positions, one per thread
    const float* __restrict__ wsrc[3]; int arow[3], aslot[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int p = tid + 512 * i, r = p / 6, sl = p - 6 * r;
        arow[i] = r; aslot[i] = (sl ^ ((r >> 3) & 1)) * 4;
        wsrc[i] = a.w + ((size_t)(2 * by + (r >> 7)) * a.nk_img) * TILE_W + (size_t)(r & 127) * (LDB3 / 2) + sl * 4;
    }
    const int bsite = tid >> 1, bhalf = tid & 1;
    const float* __restrict__ b0 = a.in0 ? a.in0 + (size_t)(2 * bx + (bsite >> 7)) * a.in0_tile_stride + (bsite & 127) * BK + bhalf * 8 : nullptr;
    const float* __restrict__ b1 = a.in1 ? a.in1 + (size_t)(2 * bx + (bsite >> 7)) * a.in1_tile_stride + (bsite & 127) * BK + bhalf * 8 : nullptr;
    // BS: the 256 x 96 B of the two activation tile images are 1536 pieces as well, three per thread, the same rows and slots as the weights'
    const float* __restrict__ bs0[3]; const float* __restrict__ bs1[3];
    if (BS) {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int p = tid + 512 * i, r = p / 6, sl = p - 6 * r;
            bs0[i] = a.in0 ? a.in0 + (size_t)(2 * bx + (r >> 7)) * a.in0_tile_stride + (size_t)(r & 127) * ROW_F3 + sl * 4 : nullptr;
            bs1[i] = a.in1 ? a.in1 + (size_t)(2 * bx + (r >> 7)) * a.in1_tile_stride + (size_t)(r & 127) * ROW_F3 + sl * 4 : nullptr;
        }
    }
    f32x4 ga[3], gb0, gb1, gb2;
    auto gload = [&](int kc) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 3; ++i) ga[i] = *reinterpret_cast<const f32x4*>(wsrc[i] + (size_t)kc * TILE_W);
        if (BS) {
            const bool first = kc < a.nk0;
            const size_t o = (size_t)(first ? kc : kc - a.nk0) * TILE_B;
            gb0 = *reinterpret_cast<const f32x4*>((first ? bs0[0] : bs1[0]) + o);
            gb1 = *reinterpret_cast<const f32x4*>((first ? bs0[1] : bs1[1]) + o);
            gb2 = *reinterpret_cast<const f32x4*>((first ? bs0[2] : bs1[2]) + o);
            return;
        }
        const float* src = kc < a.nk0 ? b0 + (size_t)kc * TILE_F : b1 + (size_t)(kc - a.nk0) * TILE_F;
        gb0 = *reinterpret_cast<const f32x4*>(src); gb1 = *reinterpret_cast<const f32x4*>(src + 4);
    };
    auto lstore = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 3; ++i) *reinterpret_cast<f32x4*>(&As[buf][arow[i]][aslot[i]]) = ga[i];
        if (BS) {
            *reinterpret_cast<f32x4*>(&Bs[buf][arow[0]][aslot[0]]) = gb0;
            *reinterpret_cast<f32x4*>(&Bs[buf][arow[1]][aslot[1]]) = gb1;
            *reinterpret_cast<f32x4*>(&Bs[buf][arow[2]][aslot[2]]) = gb2;
            return;
        }
        b8_t p0, p1, p2;
        split3_b8(gb0, gb1, p0, p1, p2);
        b8_t* rb = reinterpret_cast<b8_t*>(&Bs[buf][bsite][0]);
        const int hb = bhalf ^ ((bsite >> 3) & 1);
        rb[hb] = p0; rb[2 + hb] = p1; rb[4 + hb] = p2;
    };

    // accumulators start at the bias of their rows (register 4 r4 + g of tile (rt, ct): row 256 by + 64 wr + 32 rt + 8 r4 + 4 lh + g)
    f32x16 acc[2][4];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
            const f32x4 bz = *reinterpret_cast<const f32x4*>(a.bias + BT * by + 64 * wr + 32 * rt + 8 * r4 + 4 * lh);
#pragma unroll
            for (int ct = 0; ct < 4; ++ct)
#pragma unroll
                for (int g = 0; g < 4; ++g) acc[rt][ct][4 * r4 + g] = bz[g];
        }

    gload(0);
    lstore(0);
    if (nk > 1) gload(1);
    __syncthreads();
    for (int kc = 0; kc < nk; ++kc) {
        const int cur = kc & 1;
        f32x4 af[2][3], bf[4][3];
        auto read_frags3 = [&](int p) __attribute__((always_inline)) {
            const int sl = (2 * p + (lh ^ ((li >> 3) & 1))) * 4;
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) af[rt][p] = *reinterpret_cast<const f32x4*>(&As[cur][64 * wr + 32 * rt + li][sl]);
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) bf[ct][p] = *reinterpret_cast<const f32x4*>(&Bs[cur][128 * wc + 32 * ct + li][sl]);
        };
        read_frags3(0); read_frags3(2);              // the first product is w0 . x2
        __builtin_amdgcn_sched_barrier(0);
        read_frags3(1);
        __builtin_amdgcn_sched_barrier(0);
        // six products, smallest first: (w plane, x plane) = (0,2) (1,1) (2,0) (0,1) (1,0) (0,0)
        constexpr int WP[6] = {0, 1, 2, 0, 1, 0}, XP[6] = {2, 1, 0, 1, 0, 0};
#pragma unroll
        for (int g = 0; g < 6; ++g) {
            if (g == 1 && kc + 1 < nk) {             // chunk kc + 1 (in the staging registers since the last iteration) -> LDS
                __builtin_amdgcn_sched_barrier(0);
                lstore(cur ^ 1);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (g == 3 && kc + 2 < nk) {             // chunk kc + 2 from memory
                __builtin_amdgcn_sched_barrier(0);
                gload(kc + 2);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int ct = 0; ct < 4; ++ct)
                    acc[rt][ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(b8_t, af[rt][WP[g]]), __builtin_bit_cast(b8_t, bf[ct][XP[g]]),
                                                                           acc[rt][ct], 0, 0, 0);
        }
        __syncthreads();
    }

    // ---- LSTM epilogue: rows are [unit][gate]; the wave's 16 units are chunk 4 by + wr of the output image of site tile 2 bx + wc; a lane
    // writes positions p = lh * 8 + 4 * rt + r4 of its site (as k_hap_gemm) ----
    const size_t img = (size_t)(4 * by + wr) * TILE_F;
    const size_t tile = (size_t)(2 * bx + wc);
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
        const int site = 32 * ct + li;
        float* cptr = a.cstate + tile * a.c_tile_stride + img + site * BK + lh * 8;
        float* hptr = a.out + tile * a.out_tile_stride + img + site * BK + lh * 8;
        f32x4 cv[2], hv[2];
        if (a.first) { cv[0] = f32x4{0.f, 0.f, 0.f, 0.f}; cv[1] = cv[0]; }
        else { cv[0] = *reinterpret_cast<const f32x4*>(cptr); cv[1] = *reinterpret_cast<const f32x4*>(cptr + 4); }
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                float cn;
                hv[rt][r4] = lstm_cell(acc[rt][ct][4 * r4 + 0], acc[rt][ct][4 * r4 + 1], acc[rt][ct][4 * r4 + 2], acc[rt][ct][4 * r4 + 3], cv[rt][r4], cn);
                cv[rt][r4] = cn;
            }
        *reinterpret_cast<f32x4*>(cptr) = cv[0]; *reinterpret_cast<f32x4*>(cptr + 4) = cv[1];
        if (BS) {
            b8_t p0, p1, p2;
            split3_b8(hv[0], hv[1], p0, p1, p2);
            b8_t* hrow = reinterpret_cast<b8_t*>(a.out + tile * a.out_tile_stride + (size_t)(4 * by + wr) * TILE_B + site * ROW_F3);
            hrow[lh] = p0; hrow[2 + lh] = p1; hrow[4 + lh] = p2;
        } else {
            *reinterpret_cast<f32x4*>(hptr) = hv[0]; *reinterpret_cast<f32x4*>(hptr + 4) = hv[1];
        }
    }
}

// launch of the tile GEMM over n_tiles site tiles x n_rt row tiles x nz slices
// AR: 0 exact fp32, 1 f16x3, 2 bf16x3 (k_hap_gemm); `true` / `false` of the round-2 callers still mean f16x3 / fp32
template <int MODE, int AR, bool CONV = false, bool BS = false, bool OS = false>
inline void launch_hap_gemm(nsnp_ctx* ctx, hipStream_t s, StepLaunch& L, int n_tiles, int n_rt, int nz)
{
    if constexpr (MODE == MODE_LSTM && AR == 2 && !CONV && NSNP_GEMM_B3X) {
        static_assert(BS == OS, "an LSTM step reads and writes one kind of activation image");
        // two row tiles x two site tiles per workgroup where the launch still fills the chip with them (one workgroup per CU)
        if (ctx->hap_b3x && n_tiles % 2 == 0 && n_rt % 2 == 0 && (long long)(n_tiles / 2) * (n_rt / 2) * nz >= ctx->n_cu) {
            hipLaunchKernelGGL(k_hap_lstm_b3x<BS>, dim3(n_tiles / 2, n_rt / 2, nz), dim3(512), 0, s, L);
            return;
        }
    }
    // (the bf16x3 mode holds 48 KB of LDS per workgroup: three per CU, no ballast)
    hipLaunchKernelGGL((k_hap_gemm<MODE, AR, CONV, BS, OS>), dim3(n_tiles, n_rt, nz), dim3(256),
#ifdef NSNP_GEMM_FORCE_BALLAST
                       (unsigned)NSNP_GEMM_FORCE_BALLAST, s, L);       // (A/B build: a fixed dynamic-LDS ballast = fewer workgroups per CU)
#else
                       AR == 2 ? 0u : gemm_lds_ballast((long long)n_tiles * n_rt * nz, ctx->n_cu), s, L);
#endif
}

// Weight images of the three arithmetic modes live at corresponding offsets of three arenas: fp32 at float offset o, the f16x3 copy
// at the same offset of its arena (same size), the bf16x3 copy at BYTE offset 6 o of its arena (every image is 1.5 times as large).
struct WeightMap {
    const float* a32; const void* other; int ar;
    const float* operator()(const float* w) const
    {
        if (ar == 0) return w;
        if (ar == 1) return reinterpret_cast<const float*>(other) + (w - a32);
        return reinterpret_cast<const float*>(reinterpret_cast<const char*>(other) + (size_t)(w - a32) * 6);
    }
};

// host: fp32 weight images (rows of 16) -> bf16x3 images at 1.5 x the offset; n_floats a multiple of 16
inline void pack_bf16x3_rows(const float* src, uint16_t* dst_arena, size_t off_floats, size_t n_floats)
{
    auto rne = [](float v) { uint32_t u; memcpy(&u, &v, 4); if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
                             return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16); };
    auto f32 = [](uint16_t h) { const uint32_t u = (uint32_t)h << 16; float v; memcpy(&v, &u, 4); return v; };
    for (size_t r = 0; r + 16 <= n_floats; r += 16) {
        uint16_t* d = dst_arena + (off_floats + r) * 3;               // 6 bytes per element
        for (int k = 0; k < 16; ++k) {
            const float v = src[off_floats + r + k];
            const uint16_t p0 = rne(v); const float r1 = v - f32(p0);
            const uint16_t p1 = rne(r1);
            d[k] = p0; d[16 + k] = p1; d[32 + k] = rne(r1 - f32(p1));
        }
    }
}

}  // namespace
