// cat_forward.hip -- the legacy haplotype caller CatModel.predict (HaplotypeModel/model.py:332-358) on gfx950.
//
//   g0, g1 [N,40,11,5] fp32 (dataset.py:862-915: per tag 20 read rows x 11 columns x (base, baseq, mapq, mask, phase))
//     |- calculate_percentage (model.py:186-194) -> [11][N][20] -> 3-layer BiLSTM(20->256) + Linear(512->256), t = 5
//     |- ResCRNN (crnn.py:118-190): six ResBlocks (crnn.py:84-115) + four max-pools on [N,10,40,11] -> [11][N][256]
//     |    -> BiLSTM(256) + Linear(512->256) on all columns -> BiLSTM(256) + Linear(512->256), t = 5
//     `- Linear(512->10) on the two 256-vectors, softmax
//
// Design.  A feature map is a "pixel image": pixel p = (site, y, x) flattened, 128 pixels per tile, channels in
// chunks of 16 -> [pixel tile][channel chunk][128][16] floats, i.e. exactly the B-operand tile image of the
// 128x128 LDS-tiled fp32 MFMA GEMM of hap_gemm.hpp.  A 3x3 convolution is that GEMM with K = 9 x C_in gathered
// on the fly (CONV mode: the loader shifts the pixel index per tap and zero-fills outside the 40x11 image), so no
// im2col buffer ever exists in HBM.  BatchNorm (eval) is folded into the weight rows and bias when the weights are
// packed; conv2 of a block takes the 1x1 shortcut as C_in/16 extra K chunks read at the pixel itself, so a ResBlock
// is two launches: relu(bn1(conv1 x)) and relu(bn2(conv2 y) + shortcut(x)).  Max-pools run on the same images; the
// last one (height 1) writes the per-column LSTM input tiles directly.  The recurrent parts reuse the fused
// GEMM + LSTM-cell step launches of the HaplotypeModel path (one launch per time step, both directions).
// The last recurrent layer of each branch only feeds column 5, so it runs 6 steps per direction instead of 11.
#include "nsnp_common.hpp"

#include <new>

#include "hap_gemm.hpp"

namespace {

constexpr int CAT_ROWS = 40, CAT_L = 11, CAT_PLANES = 5, CAT_NH = 256, CAT_CLASSES = 10;
constexpr int CAT_CENTER = 5;                         // model.py:350,352 take row [5] of the [11,N,256] outputs
constexpr int CAT_CH[7] = {10, 32, 64, 128, 128, 256, 256};
constexpr int CAT_POOL[6] = {2, 2, 0, 3, 0, 2};       // pool height after block i (crnn.py:134-160), width 3 / pad 1

// g0,g1 -> pixel image of torch.cat((g0_s, g1_s), 1): channel g*5 + plane (model.py:333-336,352), 10 of 16 used
// F16 (f16x3 mode, as in hap_gemm.hpp): every 16-float row of an image holds 16 hi halves then 16 lo halves
template <bool F16>
__device__ __forceinline__ void put_elem(float* row16, int c, float v)
{
    if (F16) { _Float16 hi, lo; split_sat(v, hi, lo); _Float16* h = reinterpret_cast<_Float16*>(row16); h[c] = hi; h[16 + c] = lo; }
    else row16[c] = v;
}
template <bool F16>
__device__ __forceinline__ float get_elem(const float* row16, int c)
{
    if (F16) { const _Float16* h = reinterpret_cast<const _Float16*>(row16); return (float)h[c] + (float)h[16 + c]; }
    return row16[c];
}

template <bool F16>
__global__ void k_cat_pack_pixels(const float* __restrict__ g0, const float* __restrict__ g1, int64_t n_pix,
                                  int64_t n_pix_pad, float* __restrict__ img)
{
    const int64_t total = n_pix_pad * 16;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(e & 15);
        const int64_t p = e >> 4;
        float v = 0.f;
        if (p < n_pix && c < 10) v = (c < 5 ? g0 : g1)[p * CAT_PLANES + (c < 5 ? c : c - 5)];
        put_elem<F16>(img + (e & ~(int64_t)15), c, v);
    }
}

// calculate_percentage (model.py:186-194) of the four (group, tag) read stacks -> LSTM input tiles
// xT[t][site tile][2 chunks][128][16], feature (g*2 + tag)*5 + {A,C,G,T,D}, 20 of 32 used
template <bool F16>
__global__ void k_cat_percentage(const float* __restrict__ g0, const float* __restrict__ g1, int64_t N, int n_tiles,
                                 float* __restrict__ xT)
{
    const int64_t total = (int64_t)CAT_L * n_tiles * TS * 4;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int gt = (int)(e & 3);                  // g*2 + tag
        const int site = (int)((e >> 2) & 127);
        const int64_t r = e >> 9;
        const int tile = (int)(r % n_tiles);
        const int t = (int)(r / n_tiles);
        const int64_t n = (int64_t)tile * TS + site;
        float f[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
        if (n < N) {
            const float* g = (gt >> 1) ? g1 : g0;
            int cnt[5] = {0, 0, 0, 0, 0}, valid = 0;
            for (int row = 0; row < 20; ++row) {
                const float v = g[((n * CAT_ROWS + (gt & 1) * 20 + row) * CAT_L + t) * CAT_PLANES];
                valid += v != -2.0f;
                cnt[0] += v == 1.0f; cnt[1] += v == 2.0f; cnt[2] += v == 3.0f; cnt[3] += v == 4.0f; cnt[4] += v == -1.0f;
            }
            // torch evaluates int64 / (int64 + 1e-9) in float32
            const float den = (float)valid + 1e-9f;
#pragma unroll
            for (int k = 0; k < 5; ++k) f[k] = __fdiv_rn((float)cnt[k], den);
        }
        float* base = xT + ((size_t)t * n_tiles + tile) * 2 * TILE_F;
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            const int feat = gt * 5 + k;
            put_elem<F16>(base + (size_t)(feat >> 4) * TILE_F + site * BK, feat & 15, f[k]);
        }
        if (gt == 3) {
#pragma unroll
            for (int feat = 20; feat < 32; ++feat) put_elem<F16>(base + (size_t)TILE_F + site * BK, feat & 15, 0.f);
        }
    }
}

// nn.MaxPool2d(kernel (kh,3), stride (kh,1), padding (0,1)) on a pixel image with cc channel chunks.
// time_major (Ho == 1): pixel (site, x) goes to xT[x][site tile][cc][site][16], the LSTM input layout.
template <bool F16>
__global__ void k_cat_pool(const float* __restrict__ in, int cc, int64_t n_sites, int H, int W, int kh, int Ho,
                           int time_major, int n_site_tiles, float* __restrict__ out)
{
    const int64_t n_out = n_sites * Ho * W;
    const int64_t n_out_tiles = NSNP_CDIV(n_out, (int64_t)TS);
    const int64_t total = n_out_tiles * cc * TS * 4;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int q = (int)(e & 3);
        const int prow = (int)((e >> 2) & 127);
        const int64_t r = e >> 9;
        const int c = (int)(r % cc);
        const int64_t tile = r / cc;
        const int64_t po = tile * TS + prow;
        if (po >= n_out) continue;
        const int64_t n = po / (Ho * W);
        const int rem = (int)(po - n * (Ho * W));
        const int yo = rem / W, xo = rem - yo * W;
        f32x4 m = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        for (int ky = 0; ky < kh; ++ky)
            for (int kx = -1; kx <= 1; ++kx) {
                const int sx = xo + kx;
                if (sx < 0 || sx >= W) continue;
                const int64_t pi = (n * H + yo * kh + ky) * W + sx;
                const float* row = in + ((size_t)(pi >> 7) * cc + c) * TILE_F + (size_t)(pi & 127) * BK;
                f32x4 v;
                if (F16) {
                    const h4 vh = *reinterpret_cast<const h4*>(reinterpret_cast<const _Float16*>(row) + q * 4);
                    const h4 vl = *reinterpret_cast<const h4*>(reinterpret_cast<const _Float16*>(row) + 16 + q * 4);
#pragma unroll
                    for (int k = 0; k < 4; ++k) v[k] = (float)vh[k] + (float)vl[k];
                } else v = *reinterpret_cast<const f32x4*>(row + q * 4);
#pragma unroll
                for (int k = 0; k < 4; ++k) m[k] = fmaxf(m[k], v[k]);
            }
        float* orow;
        if (time_major) orow = out + (((size_t)xo * n_site_tiles + (size_t)(n >> 7)) * cc + c) * TILE_F + (size_t)(n & 127) * BK;
        else            orow = out + ((size_t)tile * cc + c) * TILE_F + (size_t)prow * BK;
        if (F16) {
            h4 vh, vl;
#pragma unroll
            for (int k = 0; k < 4; ++k) { _Float16 a, b; split_h(m[k], a, b); vh[k] = a; vl[k] = b; }
            *reinterpret_cast<h4*>(reinterpret_cast<_Float16*>(orow) + q * 4) = vh;
            *reinterpret_cast<h4*>(reinterpret_cast<_Float16*>(orow) + 16 + q * 4) = vl;
        } else *reinterpret_cast<f32x4*>(orow + q * 4) = m;
    }
}

// out_layer Linear(512 -> 10) + softmax (model.py:354-357) over the cat image [site tile][32 chunks][128][16]
template <bool F16>
__global__ __launch_bounds__(256) void k_cat_head(const float* __restrict__ cat, int64_t N, const float* __restrict__ w,
                                                   const float* __restrict__ b, float* __restrict__ gt)
{
    const int lane = threadIdx.x & 63;
    const int64_t n = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    const int64_t tile = n / TS; const int site = (int)(n % TS);
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int f = lane + 64 * j;
        v[j] = get_elem<F16>(cat + (tile * 32 + (f >> 4)) * TILE_F + site * BK, f & 15);
    }
    float logit[CAT_CLASSES];
    for (int r = 0; r < CAT_CLASSES; ++r) {
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) s += w[r * 512 + lane + 64 * j] * v[j];
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        logit[r] = s + b[r];
    }
    if (lane == 0) {
        float m = logit[0];
        for (int r = 1; r < CAT_CLASSES; ++r) m = fmaxf(m, logit[r]);
        float sum = 0.f;
        for (int r = 0; r < CAT_CLASSES; ++r) { logit[r] = __expf(logit[r] - m); sum += logit[r]; }
        for (int r = 0; r < CAT_CLASSES; ++r) gt[n * CAT_CLASSES + r] = logit[r] / sum;
    }
}

// dataset.PredictDataset.__getitem__ (dataset.py:862-915): per-tag read / base-quality / mapping-quality matrices
// [N][D][L] int32 -> one group tensor [N][40][L][5] float: rows 0..19 tag 1, 20..39 tag 2 (first 20 rows of each),
// planes (base, baseq, mapq, mask = base != -2, phase = 1 | 2)
__global__ void k_cat_groups(const int32_t* __restrict__ r1, const int32_t* __restrict__ q1, const int32_t* __restrict__ m1,
                             const int32_t* __restrict__ r2, const int32_t* __restrict__ q2, const int32_t* __restrict__ m2,
                             int64_t N, int D1, int D2, int L, float* __restrict__ g)
{
    const int64_t total = N * CAT_ROWS * L;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int x = (int)(e % L);
        const int64_t r = e / L;
        const int row = (int)(r % CAT_ROWS);
        const int64_t n = r / CAT_ROWS;
        const int tag = row >= 20;
        const int rr = row - 20 * tag;
        const int D = tag ? D2 : D1;
        const size_t src = ((size_t)n * D + rr) * L + x;
        const int base = (tag ? r2 : r1)[src];
        float* o = g + (size_t)e * CAT_PLANES;
        o[0] = (float)base; o[1] = (float)(tag ? q2 : q1)[src]; o[2] = (float)(tag ? m2 : m1)[src];
        o[3] = base != -2 ? 1.f : 0.f; o[4] = (float)(tag + 1);
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// 3x3 convolution (+ 1x1 shortcut) as a GEMM whose B operand is staged ONCE per channel chunk.  The CONV mode of k_hap_gemm gathers
// the B tile of every (tap, channel chunk) from the pixel image in global memory: the image is read nine times per tile (1.19 GB of
// fetches per launch, served by L2 / the Infinity Cache) and every chunk pays the gather's address arithmetic.  Here a workgroup
// (128 pixels x 128 output channels, the GEMM's 2 x 2 waves of 64 x 64) walks the channel chunks of C_in; for each it copies the
// 128 pixels of its tile PLUS a halo of W + 1 pixels on either side (a 3x3 neighbourhood of a row-major image is a pixel-index
// offset dy W + dx) into LDS once - double-buffered, the next chunk's block in flight during the last taps - and the nine taps are
// nine row offsets into that block: the B fragment of a lane is one ds_read_b128 at row (site + halo + offset), or at an all-zero
// row where the neighbour falls outside the image (crnn.py:92-96 pads with zeros).  Only the weight tiles (A) still stream per
// chunk.  In the bf16x3 mode the block is split into its three bf16 planes while it is staged - once per channel chunk instead of
// once per tap.  K order: channel chunk major, tap minor, the shortcut's chunks last (k_hap_gemm: tap major) - the same products,
// summed in another order.
// ---------------------------------------------------------------------------------------------------------------------------------
struct ConvArgs {
    const float* w;        // weight images [row tile][nk_img][128][16] (bf16x3: 1.5 x), chunk kc = tap * cc_in + cc, then the shortcut's
    const float* bias;
    const float* x;        // pixel image [pixel tile][cc_in][128][16]
    const float* sc;       // shortcut input image [pixel tile][cc_sc][128][16] (1x1 conv: read at the pixel itself) or null
    float* out;            // [pixel tile][out_cc][128][16]
    int cc_in, cc_sc, nk_img, conv_h, conv_w, n_rows, out_tile_stride;
    long long n_pix, n_pix_alloc;
};

constexpr int CONV_HALO = 12;                          // W + 1 for the widths of this network (W = 11)
// (a workgroup stages PIX + 2 * CONV_HALO + 1 rows: its pixels, the halo on either side and the zero row - ROWS inside k_cat_conv)

// NPT = pixel tiles (of 128) per workgroup.  2 for the blocks with <= 64 output channels (round 5): their waves had 8 / 16 MFMAs
// between two barriers at 128 pixels (0.47 / 0.65 of the peak against 0.75-0.78 for the wide blocks); with 256 pixels a wave owns 64
// pixels x all rows = 16 / 32.  Same products in the same order per accumulator: bit-identical.
template <int AR, int NPT>     // AR: 0 exact fp32 (v_mfma_f32_32x32x2_f32), 1 f16x3 (images of (hi, lo) fp16 pairs, three v_mfma_f32_32x32x16_f16 per
                               // product: hap_gemm.hpp), 2 bf16x3 (six v_mfma_f32_32x32x16_bf16 per product)
__global__ __launch_bounds__(256, 2) void k_cat_conv(const ConvArgs a)
{
    constexpr bool B3 = AR == 2, F16 = AR == 1;
    constexpr int PIX = NPT * TS;                       // pixels of the workgroup
    constexpr int ROWS = PIX + 2 * CONV_HALO + 1;       // staged pixels + halo + the zero row
    constexpr int NHT = (2 * (PIX + 2 * CONV_HALO) + 255) / 256;     // staging tasks (row, half) per thread
    constexpr int ROWF = B3 ? 28 : LDK;                 // floats per LDS row: 112 B (3 planes x 16 bf16 + pad) / 80 B (16 floats + pad)
    constexpr int TILE_W = B3 ? TILE_F * 3 / 2 : TILE_F;
    constexpr int NPL = B3 ? 3 : 2;                     // 16-byte fragments per row half: three planes / two groups of four K-steps / hi and lo halves
    auto frag_at = [](int p, int lh_) { return B3 ? (2 * p + lh_) * 4 : F16 ? lh_ * 4 + p * 8 : lh_ * 8 + p * 4; };   // float offset in an LDS row
    __shared__ float As[2][TR][ROWF];
    __shared__ float Hb[2][ROWS][ROWF];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int bx = blockIdx.x, by = blockIdx.y;
    const bool small_rows = NPT == 2 || a.n_rows <= 64;
    const int wr = small_rows ? 0 : wave >> 1, wc = wave & 1;
    const int nct = (small_rows && NPT == 1) ? 1 : 2;
    const int nrt = a.n_rows <= 32 ? 1 : 2;
    const int site0 = NPT == 2 ? 64 * wave : (small_rows ? 32 * wave : 64 * wc);
    const int W = a.conv_w, HW = a.conv_h * a.conv_w;
    const int n_chunks = 9 * a.cc_in + a.cc_sc;
    const float* __restrict__ wt = a.w + (size_t)by * a.nk_img * TILE_W;

    // ---- the lane's two sites: which of the nine taps stay inside their image -------------------------------------------------
    int srow[2]; unsigned vmask[2];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
        const int s = site0 + 32 * ct + li;
        const long long pix = (long long)bx * PIX + s;
        const int rem = (int)(pix % HW);
        const int py = rem / W, px = rem - py * W;
        unsigned m = 0u;
        if (pix < a.n_pix) {
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int yy = py + t / 3 - 1, xx = px + t % 3 - 1;
                m |= (unsigned)(yy >= 0 && yy < a.conv_h && xx >= 0 && xx < W) << t;
            }
        }
        srow[ct] = s + CONV_HALO; vmask[ct] = m;
    }

    // ---- staging roles ---------------------------------------------------------------------------------------------------------
    const int crow = tid >> 1, cq = (tid & 1) * 2;      // A: row, first of the thread's 16-byte pieces (fp32: 2 of 4, bf16x3: 3 of 6)
    f32x4 ga[3];
    auto gload_a = [&](int kc) {
        if (B3) {
            const f32x4* pa = reinterpret_cast<const f32x4*>(wt + (size_t)kc * TILE_W) + crow * 6 + (tid & 1) * 3;
            ga[0] = pa[0]; ga[1] = pa[1]; ga[2] = pa[2];
        } else {
            const f32x4* pa = reinterpret_cast<const f32x4*>(wt + (size_t)kc * TILE_F) + crow * 4 + cq;
            ga[0] = pa[0]; ga[1] = pa[1];
        }
    };
    auto lstore_a = [&](int buf) {
        f32x4* ra = reinterpret_cast<f32x4*>(&As[buf][crow][0]);
        if (B3) { ra[(tid & 1) * 3] = ga[0]; ra[(tid & 1) * 3 + 1] = ga[1]; ra[(tid & 1) * 3 + 2] = ga[2]; }
        else { ra[cq] = ga[0]; ra[cq + 1] = ga[1]; }
    };
    // block: task i = 2 row + half, i < 2 (PIX + 2 HALO) = 304 (560 with two pixel tiles): the thread's tasks are tid, tid + 256, ...
    f32x4 gh[NHT][2];
    auto gload_h = [&](int blk) {
        const bool conv = blk < a.cc_in;
        const float* __restrict__ img = conv ? a.x : a.sc;
        const int cc = conv ? blk : blk - a.cc_in, ncc = conv ? a.cc_in : a.cc_sc;
#pragma unroll
        for (int k = 0; k < NHT; ++k) {
            const int i = tid + 256 * k;
            const int row = i >> 1, half = i & 1;
            const long long pix = (long long)bx * PIX - CONV_HALO + row;
            gh[k][0] = f32x4{0.f, 0.f, 0.f, 0.f}; gh[k][1] = gh[k][0];
            if (i < 2 * (PIX + 2 * CONV_HALO) && pix >= 0 && pix < a.n_pix_alloc) {
                const f32x4* pr = reinterpret_cast<const f32x4*>(img + ((size_t)(pix >> 7) * ncc + cc) * TILE_F + (size_t)(pix & 127) * BK) + 2 * half;
                gh[k][0] = pr[0]; gh[k][1] = pr[1];
            }
        }
    };
    auto lstore_h = [&](int buf) {
#pragma unroll
        for (int k = 0; k < NHT; ++k) {
            const int i = tid + 256 * k;
            if (i >= 2 * (PIX + 2 * CONV_HALO)) continue;
            const int row = i >> 1, half = i & 1;
            f32x4* rh = reinterpret_cast<f32x4*>(&Hb[buf][row][0]);
            if (B3) {
                b8_t p0, p1, p2;
                split3_b8(gh[k][0], gh[k][1], p0, p1, p2);
                rh[half] = __builtin_bit_cast(f32x4, p0); rh[2 + half] = __builtin_bit_cast(f32x4, p1); rh[4 + half] = __builtin_bit_cast(f32x4, p2);
            } else { rh[2 * half] = gh[k][0]; rh[2 * half + 1] = gh[k][1]; }
        }
    };

    // accumulators start at the bias of their rows
    f32x16 acc[2][2];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
            const int f = 128 * by + 64 * wr + 32 * rt + 8 * r4 + 4 * lh;
            f32x4 bz = f32x4{0.f, 0.f, 0.f, 0.f};
            if (f < a.n_rows) bz = *reinterpret_cast<const f32x4*>(a.bias + f);
#pragma unroll
            for (int g = 0; g < 4; ++g) { acc[rt][0][4 * r4 + g] = bz[g]; acc[rt][1][4 * r4 + g] = bz[g]; }
        }

    // ---- prologue: zero rows, block 0, weight chunk 0 ---------------------------------------------------------------------------
    if (tid < 2 * ROWF) Hb[tid / ROWF][ROWS - 1][tid % ROWF] = 0.f;
    // chunk index (into the weight image) of the j-th chunk of the walk: blocks of nine taps over the channel chunks, then the shortcut's
    auto chunk_kc = [&](int jj) {
        const int nconv = 9 * a.cc_in;
        if (jj < nconv) { const int b = jj / 9, t = jj - 9 * b; return t * a.cc_in + b; }
        return jj;                                      // shortcut chunk s: 9 cc_in + s = its own position in the walk
    };
    gload_h(0); lstore_h(0);
    gload_a(chunk_kc(0)); lstore_a(0);                  // chunk 0 = (tap 0, cc 0), or the first shortcut chunk when there is no conv part
    if (n_chunks > 1) gload_a(chunk_kc(1));             // weights travel TWO chunks ahead with one set of staging registers (as k_hap_gemm, PIPE 2):
    __syncthreads();                                    // written to LDS early in the burst of the chunk before theirs, re-loaded later in the same burst

    int blk = 0, tap = 0, cur = 0, hb = 0;
    for (int j = 0; j < n_chunks; ++j) {
        const bool conv = blk < a.cc_in;
        const int ntap = conv ? 9 : 1;
        // successor chunk
        int nblk = blk, ntp = tap + 1;
        if (ntp == ntap) { nblk = blk + 1; ntp = 0; }
        const bool have_next = j + 1 < n_chunks;
        const bool stage_next_block = have_next && ntp == 0;          // the last tap of a block: the next block's pixels go to LDS
        // fragments of this chunk
        const int t_eff = conv ? tap : 4;                              // the shortcut reads the pixel itself
        const int off = (t_eff / 3 - 1) * W + (t_eff % 3 - 1);
        f32x4 af[2][NPL], bf[2][NPL];
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
            if (ct >= nct) continue;                                    // (row tiles of <= 64 rows: the second column tile does not exist - no read of it)
            const int row = ((vmask[ct] >> t_eff) & 1u) ? srow[ct] + off : ROWS - 1;
#pragma unroll
            for (int p = 0; p < NPL; ++p)
                bf[ct][p] = *reinterpret_cast<const f32x4*>(&Hb[hb][row][frag_at(p, lh)]);
        }
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int p = 0; p < NPL; ++p)
                af[rt][p] = *reinterpret_cast<const f32x4*>(&As[cur][64 * wr + 32 * rt + li][frag_at(p, lh)]);
        __builtin_amdgcn_sched_barrier(0);
        if (B3) {
            constexpr int WP[6] = {0, 1, 2, 0, 1, 0}, XP[6] = {2, 1, 0, 1, 0, 0};
#pragma unroll
            for (int g = 0; g < 6; ++g) {
                if (g == 1 && have_next) {                             // weights of chunk j + 1 (in the staging registers) -> LDS
                    __builtin_amdgcn_sched_barrier(0);
                    lstore_a(cur ^ 1);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (g == 3) {                                          // global loads from inside the burst: weights of chunk j + 2, pixels of the next block
                    __builtin_amdgcn_sched_barrier(0);
                    if (j + 2 < n_chunks) gload_a(chunk_kc(j + 2));
                    if (stage_next_block) gload_h(nblk);
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                    for (int ct = 0; ct < 2; ++ct) {
                        if (ct >= nct || rt >= nrt) continue;
                        acc[rt][ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(b8_t, af[rt][WP[g]]), __builtin_bit_cast(b8_t, bf[ct][XP[g]]),
                                                                               acc[rt][ct], 0, 0, 0);
                    }
            }
        } else if (F16) {
            // af[rt][0] / [1]: the lane's 8 hi / 8 lo halves of its weight row, bf likewise for its pixel: hi.hi + lo.hi + hi.lo (k_hap_gemm's order)
#pragma unroll
            for (int term = 0; term < 3; ++term) {
                if (term == 1 && have_next) {
                    __builtin_amdgcn_sched_barrier(0);
                    lstore_a(cur ^ 1);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (term == 2) {
                    __builtin_amdgcn_sched_barrier(0);
                    if (j + 2 < n_chunks) gload_a(chunk_kc(j + 2));
                    if (stage_next_block) gload_h(nblk);
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                    for (int ct = 0; ct < 2; ++ct) {
                        if (ct >= nct || rt >= nrt) continue;
                        acc[rt][ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, af[rt][term == 1 ? 1 : 0]),
                                                                              __builtin_bit_cast(h8, bf[ct][term == 2 ? 1 : 0]), acc[rt][ct], 0, 0, 0);
                    }
            }
        } else {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                if (k == 1 && have_next) {
                    __builtin_amdgcn_sched_barrier(0);
                    lstore_a(cur ^ 1);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (k == 4) {
                    __builtin_amdgcn_sched_barrier(0);
                    if (j + 2 < n_chunks) gload_a(chunk_kc(j + 2));
                    if (stage_next_block) gload_h(nblk);
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                    for (int ct = 0; ct < 2; ++ct) {
                        if (ct >= nct || rt >= nrt) continue;
                        acc[rt][ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[rt][k >> 2][k & 3], bf[ct][k >> 2][k & 3], acc[rt][ct], 0, 0, 0);
                    }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (stage_next_block) lstore_h(hb ^ 1);
        __syncthreads();
        cur ^= 1;
        if (ntp == 0) hb ^= 1;
        blk = nblk; tap = ntp;
    }

    // ---- epilogue: ReLU, rows < n_rows (the GEMM's LINEAR_RELU epilogue) ---------------------------------------------------------
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
        if (ct >= nct) continue;
        const int site = site0 + 32 * ct + li;
        if ((long long)bx * PIX + site >= a.n_pix_alloc) continue;     // (two pixel tiles per workgroup: the last one may hold one)
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const int f = 128 * by + 64 * wr + 32 * rt + 8 * r4 + 4 * lh;
                if (f >= a.n_rows) continue;
                f32x4 v;
#pragma unroll
                for (int g = 0; g < 4; ++g) v[g] = fmaxf(acc[rt][ct][4 * r4 + g], 0.f);
                float* o = a.out + (size_t)((size_t)bx * NPT + (site >> 7)) * a.out_tile_stride + (size_t)(f >> 4) * TILE_F + (site & 127) * BK;
                if (F16) {                                             // the image row: 16 hi halves, then 16 lo halves
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

struct LstmDir { float* w; float* b; };
struct CatBlock { float* w1; float* b1; float* w2; float* b2; int cin, cout, cc_in, cc_out; };

}  // namespace

struct CatWeightsDev {
    CatBlock blk[6];
    LstmDir rnn[2][2];                 // haplotype_base.rnn.{0,1}, direction
    float* emb_w[2]; float* emb_b[2];  // their Linear(512 -> 256)
    LstmDir pct[3][2];                 // haplotype_percentage.rnn layer, direction
    float* pct_w; float* pct_b;        // haplotype_percentage.out_layer
    float* out_w; float* out_b;        // out_layer [10][512], [10]
    float* arena; size_t arena_floats;
    float* arena16;                    // the same images with every weight as an fp16 (hi, lo) pair (cat_precision 1)
    void* arena_b3;                    // ... as three bf16 planes, every image at 1.5 x its fp32 offset (cat_precision 2; hap_gemm.hpp WeightMap)
};

void nsnp_cat_free(nsnp_ctx* ctx)
{
    if (ctx->cw) {
        if (ctx->cw->arena) (void)hipFree(ctx->cw->arena);
        if (ctx->cw->arena16) (void)hipFree(ctx->cw->arena16);
        if (ctx->cw->arena_b3) (void)hipFree(ctx->cw->arena_b3);
        delete ctx->cw; ctx->cw = nullptr;
    }
    if (ctx->cat_ws) { (void)hipFree(ctx->cat_ws); ctx->cat_ws = nullptr; ctx->cat_ws_bytes = 0; }
}

namespace {

int chunks_of(int channels) { int cc = 1; while (cc * BK < channels) cc <<= 1; return cc; }   // power of two
int shift_of(int cc) { int s = 0; while ((1 << s) < cc) ++s; return s; }

// LSTM weight image [8 row tiles][nki + 16][128][16]: rows [unit][gate]; input columns natural (prev_lstm = false)
// or in the storage order of a previous bidirectional layer's h (prev_lstm = true); recurrent columns in storage order
void pack_lstm(float* img, float* bias, const float* const* q, int I, int nki, bool prev_lstm)
{
    constexpr int H = CAT_NH; const int G = 4 * H, nkh = H / BK, nk = nki + nkh;
    auto lstm_row = [](int R) { return (R & 3) * H + (R >> 2); };
    for (int by = 0; by < G / TR; ++by)
        for (int kc = 0; kc < nk; ++kc)
            for (int row = 0; row < TR; ++row)
                for (int p = 0; p < BK; ++p) {
                    const int tr = lstm_row(by * TR + row);
                    float v = 0.f;
                    if (kc < nki) {
                        const int col = prev_lstm ? (kc / nkh) * H + (kc % nkh) * BK + unit_of_pos(p) : kc * BK + p;
                        if (col < I) v = q[0][(size_t)tr * I + col];
                    } else {
                        v = q[1][(size_t)tr * H + (kc - nki) * BK + unit_of_pos(p)];
                    }
                    img[(((size_t)by * nk + kc) * TR + row) * BK + p] = lstm_gate_scale(by * TR + row) * v;
                }
    for (int R = 0; R < G; ++R) bias[R] = lstm_gate_scale(R) * (q[2][lstm_row(R)] + q[3][lstm_row(R)]);
}

// Linear(512 -> 256) on [h_fwd ; h_bwd] in LSTM storage order: image [2 row tiles][32][128][16]
void pack_linear_h(float* img, const float* W)
{
    const int H = CAT_NH, nkh = H / BK;
    for (int by = 0; by < H / TR; ++by)
        for (int kc = 0; kc < 2 * nkh; ++kc)
            for (int row = 0; row < TR; ++row)
                for (int p = 0; p < BK; ++p)
                    img[(((size_t)by * 2 * nkh + kc) * TR + row) * BK + p] =
                        W[(size_t)(by * TR + row) * 2 * H + (kc / nkh) * H + (kc % nkh) * BK + unit_of_pos(p)];
}

}  // namespace

namespace {
constexpr int64_t CAT_PASS = 4096;
// workspace (floats): three rotating feature-map images, LSTM inputs / states / outputs
struct CatWsLayout { size_t map_f, xp_f, seq_f, hb_f, c_f, cat_f, bytes; };
CatWsLayout cat_ws_layout()
{
    const int64_t chunk = CAT_PASS; const size_t max_tiles = (size_t)(chunk / TS);
    CatWsLayout w;
    w.map_f = ((size_t)NSNP_CDIV(chunk * CAT_ROWS * CAT_L, TS) * 2 + 16) * TILE_F;   // 32 channels at full resolution = the largest map
    w.xp_f = (size_t)CAT_L * max_tiles * 2 * TILE_F;
    w.seq_f = (size_t)CAT_L * max_tiles * 16 * TILE_F;
    w.hb_f = (size_t)CAT_L * max_tiles * 32 * TILE_F;
    w.c_f = (size_t)2 * max_tiles * 16 * TILE_F;
    w.cat_f = max_tiles * 32 * TILE_F;
    w.bytes = (3 * w.map_f + w.xp_f + 2 * w.seq_f + 2 * w.hb_f + w.c_f + w.cat_f) * sizeof(float);
    return w;
}


// (Re)allocates the workspace of one pass; synchronous, called from nsnp_cat_load_weights only
int nsnp_cat_reserve(nsnp_ctx* ctx)
{
    const CatWsLayout w = cat_ws_layout();
    if (ctx->cat_ws && ctx->cat_ws_bytes == w.bytes) return NSNP_OK;
    NSNP_HIP(ctx, hipDeviceSynchronize());
    if (ctx->cat_ws) (void)hipFree(ctx->cat_ws);
    ctx->cat_ws = nullptr; ctx->cat_ws_bytes = 0;
    if (hipMalloc(&ctx->cat_ws, w.bytes) != hipSuccess) { ctx->cat_ws = nullptr; ctx->last_err = hipErrorOutOfMemory; return NSNP_ENOMEM; }
    ctx->cat_ws_bytes = w.bytes;
    return NSNP_OK;
}
}  // namespace

extern "C" int nsnp_cat_load_weights(nsnp_ctx* ctx, const float* const* t, int n_tensors)
{
    if (!ctx || !t) return NSNP_EINVAL;
    if (n_tensors < 132) return NSNP_EINVAL;
    for (int i = 0; i < 132; ++i) if (!t[i]) return NSNP_EINVAL;
    NSNP_HIP(ctx, hipSetDevice(ctx->device));
    const int H = CAT_NH, G = 4 * H;
    size_t total = 0;
    auto take = [&total](size_t n) { const size_t o = total; total += (n + 15) & ~(size_t)15; return o; };
    size_t o_w1[6], o_b1[6], o_w2[6], o_b2[6];
    int cc_in[6], cc_out[6], rt[6];
    for (int i = 0; i < 6; ++i) {
        cc_in[i] = chunks_of(CAT_CH[i]); cc_out[i] = chunks_of(CAT_CH[i + 1]); rt[i] = NSNP_CDIV(CAT_CH[i + 1], TR);
        o_w1[i] = take((size_t)rt[i] * 9 * cc_in[i] * TILE_F);
        o_b1[i] = take((size_t)rt[i] * TR);
        o_w2[i] = take((size_t)rt[i] * (9 * cc_out[i] + cc_in[i]) * TILE_F);
        o_b2[i] = take((size_t)rt[i] * TR);
    }
    size_t o_rw[2][2], o_rb[2][2], o_ew[2], o_eb[2], o_pw[3][2], o_pb[3][2];
    for (int r = 0; r < 2; ++r) {
        for (int d = 0; d < 2; ++d) { o_rw[r][d] = take((size_t)(G / TR) * 32 * TILE_F); o_rb[r][d] = take(G); }
        o_ew[r] = take((size_t)(H / TR) * 32 * TILE_F); o_eb[r] = take(H);
    }
    for (int l = 0; l < 3; ++l)
        for (int d = 0; d < 2; ++d) { o_pw[l][d] = take((size_t)(G / TR) * ((l == 0 ? 2 : 32) + 16) * TILE_F); o_pb[l][d] = take(G); }
    const size_t o_pcw = take((size_t)(H / TR) * 32 * TILE_F), o_pcb = take(H);
    const size_t o_ow = take((size_t)CAT_CLASSES * 2 * H), o_ob = take(16);

    std::vector<float> host(total, 0.f);
    for (int i = 0; i < 6; ++i) {
        const float* const* q = t + 14 * i;
        const int ci = CAT_CH[i], co = CAT_CH[i + 1];
        // eval-mode BatchNorm folded into the conv: s = gamma / sqrt(var + eps); w' = s * w; b' = (b - mean) * s + beta
        std::vector<float> s1(co), s2(co);
        for (int c = 0; c < co; ++c) { s1[c] = q[2][c] / sqrtf(q[5][c] + 1e-5f); s2[c] = q[8][c] / sqrtf(q[11][c] + 1e-5f); }
        {
            float* img = host.data() + o_w1[i]; const int cc = cc_in[i], nk = 9 * cc;
            for (int by = 0; by < rt[i]; ++by)
                for (int kc = 0; kc < nk; ++kc)
                    for (int row = 0; row < TR; ++row)
                        for (int p = 0; p < BK; ++p) {
                            const int o = by * TR + row, tap = kc / cc, c = (kc % cc) * BK + p;
                            float v = 0.f;
                            if (o < co && c < ci) v = q[0][((size_t)o * ci + c) * 9 + tap] * s1[o];
                            img[(((size_t)by * nk + kc) * TR + row) * BK + p] = v;
                        }
            float* bz = host.data() + o_b1[i];
            for (int o = 0; o < co; ++o) bz[o] = (q[1][o] - q[4][o]) * s1[o] + q[3][o];
        }
        {
            float* img = host.data() + o_w2[i]; const int cc = cc_out[i], nk0 = 9 * cc, nk = nk0 + cc_in[i];
            for (int by = 0; by < rt[i]; ++by)
                for (int kc = 0; kc < nk; ++kc)
                    for (int row = 0; row < TR; ++row)
                        for (int p = 0; p < BK; ++p) {
                            const int o = by * TR + row;
                            float v = 0.f;
                            if (o < co) {
                                if (kc < nk0) { const int tap = kc / cc, c = (kc % cc) * BK + p; if (c < co) v = q[6][((size_t)o * co + c) * 9 + tap] * s2[o]; }
                                else { const int c = (kc - nk0) * BK + p; if (c < ci) v = q[12][(size_t)o * ci + c]; }
                            }
                            img[(((size_t)by * nk + kc) * TR + row) * BK + p] = v;
                        }
            float* bz = host.data() + o_b2[i];
            for (int o = 0; o < co; ++o) bz[o] = (q[7][o] - q[10][o]) * s2[o] + q[9][o] + q[13][o];
        }
    }
    for (int r = 0; r < 2; ++r) {
        const float* const* q = t + 84 + 10 * r;
        for (int d = 0; d < 2; ++d) pack_lstm(host.data() + o_rw[r][d], host.data() + o_rb[r][d], q + 4 * d, H, 16, false);
        pack_linear_h(host.data() + o_ew[r], q[8]);
        memcpy(host.data() + o_eb[r], q[9], sizeof(float) * H);
    }
    {
        const float* const* q = t + 104;
        for (int l = 0; l < 3; ++l)
            for (int d = 0; d < 2; ++d)
                pack_lstm(host.data() + o_pw[l][d], host.data() + o_pb[l][d], q + (l * 2 + d) * 4, l == 0 ? 20 : 2 * H, l == 0 ? 2 : 32, l > 0);
        pack_linear_h(host.data() + o_pcw, q[24]);
        memcpy(host.data() + o_pcb, q[25], sizeof(float) * H);
    }
    memcpy(host.data() + o_ow, t[130], sizeof(float) * CAT_CLASSES * 2 * H);
    memcpy(host.data() + o_ob, t[131], sizeof(float) * CAT_CLASSES);

    if (!ctx->cw) { ctx->cw = new (std::nothrow) CatWeightsDev(); if (!ctx->cw) return NSNP_ENOMEM; memset((void*)ctx->cw, 0, sizeof(CatWeightsDev)); }
    CatWeightsDev& cw = *ctx->cw;
    if (!cw.arena) { NSNP_HIP(ctx, hipMalloc((void**)&cw.arena, total * sizeof(float))); cw.arena_floats = total; }
    NSNP_HIP(ctx, hipMemcpy(cw.arena, host.data(), total * sizeof(float), hipMemcpyHostToDevice));
    {
        // f16x3 images: rows of 16 weights become 16 hi halves + 16 lo halves; biases and out_layer stay fp32
        std::vector<float> h16(host);
        auto conv = [&](size_t off, size_t n_floats) {
            for (size_t r = 0; r + 16 <= n_floats; r += 16) {
                _Float16* dst = reinterpret_cast<_Float16*>(h16.data() + off + r);
                const float* src = host.data() + off + r;
                for (int k = 0; k < 16; ++k) { const _Float16 hi = (_Float16)src[k]; dst[k] = hi; dst[16 + k] = (_Float16)(src[k] - (float)hi); }
            }
        };
        for (int i = 0; i < 6; ++i) {
            conv(o_w1[i], (size_t)rt[i] * 9 * cc_in[i] * TILE_F);
            conv(o_w2[i], (size_t)rt[i] * (9 * cc_out[i] + cc_in[i]) * TILE_F);
        }
        for (int r = 0; r < 2; ++r) {
            for (int d = 0; d < 2; ++d) conv(o_rw[r][d], (size_t)(G / TR) * 32 * TILE_F);
            conv(o_ew[r], (size_t)(H / TR) * 32 * TILE_F);
        }
        for (int l = 0; l < 3; ++l) for (int d = 0; d < 2; ++d) conv(o_pw[l][d], (size_t)(G / TR) * ((l == 0 ? 2 : 32) + 16) * TILE_F);
        conv(o_pcw, (size_t)(H / TR) * 32 * TILE_F);
        if (!cw.arena16) NSNP_HIP(ctx, hipMalloc((void**)&cw.arena16, total * sizeof(float)));
        NSNP_HIP(ctx, hipMemcpy(cw.arena16, h16.data(), total * sizeof(float), hipMemcpyHostToDevice));
    }
    {
        // bf16x3 images: every weight as three bf16 planes, image at 1.5 x its fp32 offset; biases and out_layer stay fp32
        std::vector<uint16_t> h3(total * 3, 0);
        auto conv3 = [&](size_t off, size_t n_floats) { pack_bf16x3_rows(host.data(), h3.data(), off, n_floats); };
        for (int i = 0; i < 6; ++i) {
            conv3(o_w1[i], (size_t)rt[i] * 9 * cc_in[i] * TILE_F);
            conv3(o_w2[i], (size_t)rt[i] * (9 * cc_out[i] + cc_in[i]) * TILE_F);
        }
        for (int r = 0; r < 2; ++r) {
            for (int d = 0; d < 2; ++d) conv3(o_rw[r][d], (size_t)(G / TR) * 32 * TILE_F);
            conv3(o_ew[r], (size_t)(H / TR) * 32 * TILE_F);
        }
        for (int l = 0; l < 3; ++l) for (int d = 0; d < 2; ++d) conv3(o_pw[l][d], (size_t)(G / TR) * ((l == 0 ? 2 : 32) + 16) * TILE_F);
        conv3(o_pcw, (size_t)(H / TR) * 32 * TILE_F);
        if (!cw.arena_b3) NSNP_HIP(ctx, hipMalloc(&cw.arena_b3, total * 6));
        NSNP_HIP(ctx, hipMemcpy(cw.arena_b3, h3.data(), total * 6, hipMemcpyHostToDevice));
    }
    for (int i = 0; i < 6; ++i) {
        cw.blk[i] = CatBlock{cw.arena + o_w1[i], cw.arena + o_b1[i], cw.arena + o_w2[i], cw.arena + o_b2[i],
                             CAT_CH[i], CAT_CH[i + 1], cc_in[i], cc_out[i]};
    }
    for (int r = 0; r < 2; ++r) {
        for (int d = 0; d < 2; ++d) cw.rnn[r][d] = LstmDir{cw.arena + o_rw[r][d], cw.arena + o_rb[r][d]};
        cw.emb_w[r] = cw.arena + o_ew[r]; cw.emb_b[r] = cw.arena + o_eb[r];
    }
    for (int l = 0; l < 3; ++l) for (int d = 0; d < 2; ++d) cw.pct[l][d] = LstmDir{cw.arena + o_pw[l][d], cw.arena + o_pb[l][d]};
    cw.pct_w = cw.arena + o_pcw; cw.pct_b = cw.arena + o_pcb; cw.out_w = cw.arena + o_ow; cw.out_b = cw.arena + o_ob;
    return nsnp_cat_reserve(ctx);                        // the forward itself never allocates
}

namespace {

// one bidirectional LSTM layer: `steps` launches, both directions per launch.
//   in : [t][site tile][nk_in chunks] tile images;  hout: [t][site tile][dir][16 chunks]
template <int AR>
void run_bilstm(nsnp_ctx* ctx, hipStream_t s, const LstmDir* dirs, const WeightMap& wm, const float* in, int nk_in, float* hout, float* cst,
                int n_tiles, int T, int steps)
{
    const size_t tile_h = (size_t)16 * TILE_F, step_h = (size_t)n_tiles * 2 * tile_h;
    const size_t in_tile = (size_t)nk_in * TILE_F, in_step = (size_t)n_tiles * in_tile;
    for (int st = 0; st < steps; ++st) {
        StepLaunch L;
        for (int d = 0; d < 2; ++d) {
            const int t = d ? T - 1 - st : st, tprev = d ? t + 1 : t - 1;
            StepArgs& a = L.z[d];
            memset(&a, 0, sizeof(a));
            a.w = wm(dirs[d].w); a.bias = dirs[d].b;
            a.in0 = in + (size_t)t * in_step; a.nk0 = nk_in; a.in0_tile_stride = (int)in_tile;
            a.in1 = st ? hout + (size_t)tprev * step_h + (size_t)d * tile_h : nullptr;
            a.nk1 = st ? 16 : 0; a.in1_tile_stride = (int)(2 * tile_h);
            a.nk_img = nk_in + 16;
            a.out = hout + (size_t)t * step_h + (size_t)d * tile_h; a.out_tile_stride = (int)(2 * tile_h);
            a.cstate = cst + (size_t)d * n_tiles * tile_h; a.c_tile_stride = (int)tile_h;
            a.first = st == 0;
        }
        launch_hap_gemm<MODE_LSTM, AR>(ctx, s, L, (int)(n_tiles), 4 * CAT_NH / TR, 2);
    }
}

// Linear(512 -> 256) over `n_in_tiles` consecutive [dir][16 chunks] h tiles
template <int AR>
void run_linear_h(nsnp_ctx* ctx, hipStream_t s, const float* w, const float* b, const float* in, int n_in_tiles, float* out, int out_tile_stride)
{
    StepLaunch L; StepArgs& a = L.z[0];
    memset(&a, 0, sizeof(a));
    a.w = w; a.bias = b; a.in0 = in; a.nk0 = 32; a.in0_tile_stride = 32 * TILE_F; a.nk_img = 32;
    a.out = out; a.out_tile_stride = out_tile_stride;
    launch_hap_gemm<MODE_LINEAR, AR>(ctx, s, L, (int)(n_in_tiles), CAT_NH / TR, 1);
}

int grid_for(int64_t total) { int64_t b = NSNP_CDIV(total, (int64_t)256); return (int)(b > 16384 ? 16384 : (b < 1 ? 1 : b)); }

}  // namespace

namespace {

template <int AR>            // 0 exact fp32, 1 f16x3 (activation images hold fp16 pairs), 2 bf16x3 (activations stay fp32, weights as three bf16 planes)
int cat_forward_impl(nsnp_ctx* ctx, const float* g0, const float* g1, int64_t N, float* gt_prob, void* stream)
{
    if (!ctx || N < 0 || (N > 0 && (!g0 || !g1 || !gt_prob))) return NSNP_EINVAL;
    if (!ctx->cw) return NSNP_ENOWEIGHTS;
    if (N == 0) return NSNP_OK;
    const CatWeightsDev& cw = *ctx->cw;
    hipStream_t s = (hipStream_t)stream;
    constexpr bool F16 = AR == 1;                                 // element format of the activation images
    const WeightMap wm{cw.arena, AR == 1 ? (const void*)cw.arena16 : (const void*)cw.arena_b3, AR};
    const int64_t chunk = CAT_PASS;                               // sites per pass
    const int max_tiles = (int)(chunk / TS);
    const CatWsLayout wl = cat_ws_layout();
    if (!ctx->cat_ws || ctx->cat_ws_bytes < wl.bytes) return NSNP_ENOMEM;        // nsnp_cat_load_weights reserves it
    const size_t map_f = wl.map_f, xp_f = wl.xp_f, seq_f = wl.seq_f, hb_f = wl.hb_f, c_f = wl.c_f;
    (void)max_tiles;
    float* base = (float*)ctx->cat_ws;
    float* map[3] = {base, base + map_f, base + 2 * map_f};
    float* xp = base + 3 * map_f;
    float* seq[2] = {xp + xp_f, xp + xp_f + seq_f};
    float* hb[2] = {seq[1] + seq_f, seq[1] + seq_f + hb_f};
    float* cst = hb[1] + hb_f;
    float* cat = cst + c_f;

    for (int64_t n0 = 0; n0 < N; n0 += chunk) {
        const int64_t n = N - n0 < chunk ? N - n0 : chunk;
        const int n_tiles = (int)NSNP_CDIV(n, TS);
        const float* a0 = g0 + n0 * CAT_ROWS * CAT_L * CAT_PLANES;
        const float* a1 = g1 + n0 * CAT_ROWS * CAT_L * CAT_PLANES;
        const size_t step_h = (size_t)n_tiles * 32 * TILE_F;
        ScopedKernelTimer tm_all(ctx, NSNP_K_CAT, s);

        // ---- percentage branch: 3-layer BiLSTM on [11][n][20], Linear at column 5 -> cat chunks 0..15 ----
        hipLaunchKernelGGL(k_cat_percentage<F16>, dim3(grid_for((int64_t)CAT_L * n_tiles * TS * 4)), dim3(256), 0, s, a0, a1, n, n_tiles, xp);
        run_bilstm<AR>(ctx, s, cw.pct[0], wm, xp, 2, hb[0], cst, n_tiles, CAT_L, CAT_L);
        run_bilstm<AR>(ctx, s, cw.pct[1], wm, hb[0], 32, hb[1], cst, n_tiles, CAT_L, CAT_L);
        run_bilstm<AR>(ctx, s, cw.pct[2], wm, hb[1], 32, hb[0], cst, n_tiles, CAT_L, CAT_CENTER + 1);
        run_linear_h<AR>(ctx, s, wm(cw.pct_w), cw.pct_b, hb[0] + (size_t)CAT_CENTER * step_h, n_tiles, cat, 32 * TILE_F);

        // ---- ResCRNN branch ----
        int Hc = CAT_ROWS;
        int64_t n_pix = n * Hc * CAT_L;
        int64_t n_ptiles = NSNP_CDIV(n_pix, (int64_t)TS);
        hipLaunchKernelGGL(k_cat_pack_pixels<F16>, dim3(grid_for(n_ptiles * TS * 16)), dim3(256), 0, s, a0, a1, n_pix, n_ptiles * TS, map[0]);
        int cur = 0;                                              // map[cur] holds the block input
        ScopedKernelTimer tm_conv(ctx, NSNP_K_CATCONV, s);         // the 12 conv GEMM launches (+ 4 pools) of this pass
        for (int i = 0; i < 6; ++i) {
            const CatBlock& b = cw.blk[i];
            const int rt = NSNP_CDIV(b.cout, TR);
            float* X = map[cur]; float* Y = map[(cur + 1) % 3]; float* O = map[(cur + 2) % 3];
            if (ctx->cat_conv_lds) {
                // the LDS-staged convolution kernel (all three arithmetics; option "cat_conv_lds" 0: the gathering GEMM)
                ConvArgs c;
                c.conv_h = Hc; c.conv_w = CAT_L; c.n_pix = n_pix; c.n_pix_alloc = n_ptiles * TS; c.n_rows = b.cout;
                c.w = wm(b.w1); c.bias = b.b1; c.x = X; c.sc = nullptr; c.cc_in = b.cc_in; c.cc_sc = 0; c.nk_img = 9 * b.cc_in;
                c.out = Y; c.out_tile_stride = b.cc_out * TILE_F;
                // blocks of <= 64 output channels: two pixel tiles per workgroup (option "cat_conv_pix2", default 1)
                // (fp32 only: the bf16x3 form needs 92 KB of LDS at 256 pixels - one workgroup per CU - and measured 2 % slower)
                const bool pix2 = AR != 2 && ctx->cat_conv_pix2 && b.cout <= 64;
                const unsigned gx = pix2 ? (unsigned)((n_ptiles + 1) / 2) : (unsigned)n_ptiles;
                if (pix2) hipLaunchKernelGGL((k_cat_conv<AR == 2 ? 0 : AR, 2>), dim3(gx, rt, 1), dim3(256), 0, s, c);   // (pix2 implies AR != 2)
                else hipLaunchKernelGGL((k_cat_conv<AR, 1>), dim3(gx, rt, 1), dim3(256), 0, s, c);
                c.w = wm(b.w2); c.bias = b.b2; c.x = Y; c.sc = X; c.cc_in = b.cc_out; c.cc_sc = b.cc_in; c.nk_img = 9 * b.cc_out + b.cc_in;
                c.out = O;
                if (pix2) hipLaunchKernelGGL((k_cat_conv<AR == 2 ? 0 : AR, 2>), dim3(gx, rt, 1), dim3(256), 0, s, c);   // (pix2 implies AR != 2)
                else hipLaunchKernelGGL((k_cat_conv<AR, 1>), dim3(gx, rt, 1), dim3(256), 0, s, c);
            } else {
            StepLaunch L; StepArgs& a = L.z[0];
            memset(&a, 0, sizeof(a));
            a.conv_h = Hc; a.conv_w = CAT_L; a.n_pix = n_pix; a.n_rows = b.cout;
            // y = relu(bn1(conv1(x)))
            a.w = wm(b.w1); a.bias = b.b1; a.in0 = X; a.nk0 = 9 * b.cc_in; a.cc0_shift = shift_of(b.cc_in);
            a.nk1 = 0; a.nk_img = a.nk0; a.out = Y; a.out_tile_stride = b.cc_out * TILE_F;
            launch_hap_gemm<MODE_LINEAR_RELU, AR, true>(ctx, s, L, (int)((unsigned)n_ptiles), rt, 1);
            // out = relu(bn2(conv2(y)) + shortcut(x))
            a.w = wm(b.w2); a.bias = b.b2; a.in0 = Y; a.nk0 = 9 * b.cc_out; a.cc0_shift = shift_of(b.cc_out);
            a.in1 = X; a.nk1 = b.cc_in; a.in1_tile_stride = b.cc_in * TILE_F; a.nk_img = a.nk0 + a.nk1;
            a.out = O; a.out_tile_stride = b.cc_out * TILE_F;
            launch_hap_gemm<MODE_LINEAR_RELU, AR, true>(ctx, s, L, (int)((unsigned)n_ptiles), rt, 1);
            }
            cur = (cur + 2) % 3;
            if (CAT_POOL[i]) {
                const int kh = CAT_POOL[i], Ho = (Hc - kh) / kh + 1;
                const bool last = i == 5;
                const int64_t n_out_tiles = NSNP_CDIV(n * Ho * CAT_L, (int64_t)TS);
                float* dst = last ? seq[0] : map[(cur + 1) % 3];
                hipLaunchKernelGGL(k_cat_pool<F16>, dim3(grid_for(n_out_tiles * b.cc_out * TS * 4)), dim3(256), 0, s,
                                   map[cur], b.cc_out, n, Hc, CAT_L, kh, Ho, last ? 1 : 0, n_tiles, dst);
                Hc = Ho; n_pix = n * Hc * CAT_L; n_ptiles = n_out_tiles;
                cur = (cur + 1) % 3;
            }
        }
        tm_conv.stop();
        if (Hc != 1) return NSNP_ESHAPE;                          // crnn.py:183 asserts the same
        // BidirectionalLSTM 0: all 11 columns, embedding on every column (crnn.py:12-20)
        run_bilstm<AR>(ctx, s, cw.rnn[0], wm, seq[0], 16, hb[0], cst, n_tiles, CAT_L, CAT_L);
        run_linear_h<AR>(ctx, s, wm(cw.emb_w[0]), cw.emb_b[0], hb[0], CAT_L * n_tiles, seq[1], 16 * TILE_F);
        // BidirectionalLSTM 1: only column 5 is used downstream
        run_bilstm<AR>(ctx, s, cw.rnn[1], wm, seq[1], 16, hb[1], cst, n_tiles, CAT_L, CAT_CENTER + 1);
        run_linear_h<AR>(ctx, s, wm(cw.emb_w[1]), cw.emb_b[1], hb[1] + (size_t)CAT_CENTER * step_h, n_tiles, cat + 16 * TILE_F, 32 * TILE_F);
        hipLaunchKernelGGL(k_cat_head<F16>, dim3((unsigned)NSNP_CDIV(n, 4)), dim3(256), 0, s, cat, n, cw.out_w, cw.out_b, gt_prob + n0 * CAT_CLASSES);
    }
    NSNP_HIP(ctx, hipGetLastError());
    return NSNP_OK;
}

}  // namespace

extern "C" int nsnp_cat_forward(nsnp_ctx* ctx, const float* g0, const float* g1, int64_t N, float* gt_prob, void* stream)
{
    if (ctx && ctx->cat_precision == 1) return cat_forward_impl<1>(ctx, g0, g1, N, gt_prob, stream);
    if (ctx && ctx->cat_precision == 2) return cat_forward_impl<2>(ctx, g0, g1, N, gt_prob, stream);
    return cat_forward_impl<0>(ctx, g0, g1, N, gt_prob, stream);
}

extern "C" int nsnp_cat_groups(nsnp_ctx* ctx, const int32_t* read1, const int32_t* bq1, const int32_t* mq1, int depth1,
                               const int32_t* read2, const int32_t* bq2, const int32_t* mq2, int depth2,
                               int64_t N, int length, float* g, void* stream)
{
    if (!ctx || N < 0 || length <= 0) return NSNP_EINVAL;
    if (depth1 < 20 || depth2 < 20) return NSNP_ESHAPE;           // dataset.py:862 slices [:max_depth = 20]
    if (N == 0) return NSNP_OK;
    if (!read1 || !bq1 || !mq1 || !read2 || !bq2 || !mq2 || !g) return NSNP_EINVAL;
    hipLaunchKernelGGL(k_cat_groups, dim3(grid_for(N * CAT_ROWS * length)), dim3(256), 0, (hipStream_t)stream,
                       read1, bq1, mq1, read2, bq2, mq2, N, depth1, depth2, length, g);
    NSNP_HIP(ctx, hipGetLastError());
    return NSNP_OK;
}
