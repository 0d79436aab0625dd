// hap_forward.hip -- HaplotypeModel forward: two 3-layer BiLSTM encoders (F=105 -> H=256) over the
// 33-wide pileup window and the 11-wide haplotype window, Linear(512->256) at the centre step of
// each, tanh(Linear(512->256)), genotype / zygosity heads, softmax.
//
// Replaces model_dev.LSTMNetwork.predict (HaplotypeModel/model_dev.py:133-143; BaseEncoder :59-84,
// ForwardLayer :86-105) as called from HaplotypeModel/predict_dev.py:35-39.
//
// Design (not a translation of nn.LSTM/cuDNN): with H=256 one direction's weights are 1.5-3 MB, far
// beyond LDS, so the recurrence is run as ONE fused GEMM+cell kernel launch per time step over all
// sites of a chunk:  gates[1024 x N] = Wcat[1024 x (I+256)] . [x_t ; h_{t-1}]  on
// v_mfma_f32_32x32x2_f32 (exact fp32), 128 gate-rows x 128 sites per workgroup, K streamed through
// LDS in 16-wide chunks with register double-buffering; the epilogue applies the LSTM cell to the
// accumulators in place (a lane owns the 4 gates of a hidden unit) and writes h_t straight into the
// tile-image layout that the next step / next layer loads back with linear 8 KB copies.  Both
// directions (and both encoders while the short one is still running) share a launch through
// gridDim.z.  Only what position L/2 needs is computed in the last layer (17 of 33 / 6 of 11 steps).
#include "nsnp_common.hpp"

#include <new>

#include "hap_gemm.hpp"

#ifndef NSNP_HAP_BS
// bf16x3: 1 = activation images as three bf16 planes, written by the producing epilogue (hap_gemm.hpp BS / OS); 0 (default) = fp32 images,
// split by every consumer on their way into LDS.  Built and measured in round 6: bit-identical results (crc 24ae9da2
// both ways on 16,384 sites), and SLOWER - 21.09 against 20.41 ms per 16,384-site forward: the split's vector instructions sit in the
// shadow of the MFMAs of the other wave, what the 1.5 x larger activation images cost in L2 / HBM traffic does not.  Kept as an A/B build
// (tools/build_variant.sh bs -DNSNP_HAP_BS=1).
#define NSNP_HAP_BS 0
#endif

namespace {

// x [N][F][L] (predict_dev.py hands [N,105,L]; model_dev.py:136-137 permutes to [N,L,F]) ->
// per-step tile images xT[t][site_tile][chunk][128][16], features in natural order, zero padded.
// A transposition through LDS: workgroup (site tile, chunk, sixteenth of the tile's sites) reads the 16 x L floats that one site
// holds for the chunk's 16 features as ONE contiguous run (x is feature-major per site), coalesced, and writes for every step t
// the 16 sites x 16 features block of the tile image as one contiguous kilobyte.  (Round 2 gathered element by element, every
// lane 4 bytes out of a different 128-byte line: 3.9 GB of HBM reads per 16384-site launch for 0.23 GB of input, 0.53 ms.)
constexpr int PK_SITES = 16;                    // sites per workgroup
constexpr int PK_MAXL = 33;
template <int AR>                  // element format of the images: 0 fp32, 1 (hi, lo) fp16 pairs, 2 three bf16 planes (rows of 96 bytes)
__global__ __launch_bounds__(256) void k_hap_pack_input(const float* __restrict__ x, int64_t N, int F, int L, int n_tiles, int nkc,
                                                        float* __restrict__ xT)
{
    constexpr bool F16 = AR == 1, B3 = AR == 2;
    constexpr int TILE_X = B3 ? TILE_F3 : TILE_F, ROW_X = B3 ? ROW_F3 : BK;
    __shared__ float buf[PK_SITES][16 * PK_MAXL + 1];
    const int tid = threadIdx.x;
    const int sub = blockIdx.x % (TS / PK_SITES);
    const int kc = (blockIdx.x / (TS / PK_SITES)) % nkc;
    const int tile = blockIdx.x / ((TS / PK_SITES) * nkc);
    const int f0 = kc * 16;
    const int nf = F - f0 < 16 ? F - f0 : 16;                 // valid features of this chunk
    const int run = nf * L;                                    // contiguous floats per site
    for (int i = tid; i < PK_SITES * 16 * L; i += 256) {
        const int s = i / (16 * L), r = i - s * (16 * L);
        const int64_t n = (int64_t)tile * TS + sub * PK_SITES + s;
        buf[s][r] = (n < N && r < run) ? x[(n * F + f0) * L + r] : 0.f;
    }
    __syncthreads();
    // per step t: 16 sites x 16 features = 64 pieces of 4 features; thread -> (t, site, 4 features)
    for (int i = tid; i < L * PK_SITES * 4; i += 256) {
        const int t = i / (PK_SITES * 4), j = i - t * (PK_SITES * 4);
        const int s = j >> 2, p4 = (j & 3) * 4;
        const f32x4 v = f32x4{buf[s][(p4 + 0) * L + t], buf[s][(p4 + 1) * L + t], buf[s][(p4 + 2) * L + t], buf[s][(p4 + 3) * L + t]};
        float* row = xT + (((size_t)t * n_tiles + tile) * nkc + kc) * TILE_X + (size_t)(sub * PK_SITES + s) * ROW_X;
        if (B3) {
            b4_t p0, p1, p2;
            split3_b4(v, p0, p1, p2);
            __bf16* br = reinterpret_cast<__bf16*>(row);
            *reinterpret_cast<b4_t*>(br + p4) = p0;
            *reinterpret_cast<b4_t*>(br + 16 + p4) = p1;
            *reinterpret_cast<b4_t*>(br + 32 + p4) = p2;
        } else if (F16) {
            h4 vh, vl;
#pragma unroll
            for (int g = 0; g < 4; ++g) { _Float16 hi, lo; split_sat(v[g], hi, lo); vh[g] = hi; vl[g] = lo; }
            _Float16* hr = reinterpret_cast<_Float16*>(row);
            *reinterpret_cast<h4*>(hr + p4) = vh;
            *reinterpret_cast<h4*>(hr + 16 + p4) = vl;
        } else {
            *reinterpret_cast<f32x4*>(row + p4) = v;
        }
    }
}

// heads: logits = W[rows x 256] . inner + b over the dense output image (natural order), softmax
template <bool F16>
__global__ __launch_bounds__(256) void k_hap_heads(const float* __restrict__ inner, int64_t N,
                                                    const float* __restrict__ w, const float* __restrict__ b,
                                                    int n_gt, int n_zy, float* __restrict__ gt, float* __restrict__ zy)
{
    // one wave per site; lane k handles features k, k+64, ...
    const int lane = threadIdx.x & 63;
    const int64_t n = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    const int64_t tile = n / TS; const int site = (int)(n % TS);
    float v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int f = lane + 64 * j;
        const float* row = inner + (tile * 16 + (f >> 4)) * TILE_F + site * BK;
        if (F16) { const _Float16* hr = reinterpret_cast<const _Float16*>(row); v[j] = (float)hr[f & 15] + (float)hr[16 + (f & 15)]; }
        else v[j] = row[f & 15];
    }
    const int rows = n_gt + n_zy;
    float logit[16];
    for (int r = 0; r < rows; ++r) {
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) s += w[r * 256 + lane + 64 * j] * v[j];
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        logit[r] = s + b[r];
    }
    if (lane == 0) {
        float m = logit[0];
        for (int r = 1; r < n_gt; ++r) m = fmaxf(m, logit[r]);
        float sum = 0.f;
        for (int r = 0; r < n_gt; ++r) { logit[r] = __expf(logit[r] - m); sum += logit[r]; }
        for (int r = 0; r < n_gt; ++r) gt[n * n_gt + r] = logit[r] / sum;
        m = logit[n_gt];
        for (int r = 1; r < n_zy; ++r) m = fmaxf(m, logit[n_gt + r]);
        sum = 0.f;
        for (int r = 0; r < n_zy; ++r) { logit[n_gt + r] = __expf(logit[n_gt + r] - m); sum += logit[n_gt + r]; }
        for (int r = 0; r < n_zy; ++r) zy[n * n_zy + r] = logit[n_gt + r] / sum;
    }
}

}  // namespace

// -------------------------------------------------------------------------------------------------
// host side
// -------------------------------------------------------------------------------------------------
struct HapWeightsDev;
int nsnp_hap_reserve(nsnp_ctx* ctx);

struct HapWeightsDev {
    int F, H, n_layers, n_gt, n_zy, nk_in0;           // nk_in0 = ceil(F/16)
    // [encoder][layer][dir]: weight images + bias
    float* w[2][3][2]; float* b[2][3][2];
    float* proj_w[2]; float* proj_b[2];                // Linear(2H -> H) per encoder, inputs in LSTM storage order
    float* dense_w; float* dense_b;                    // Linear(2H -> H), inputs natural order (two proj outputs)
    float* head_w; float* head_b;                      // [(n_gt+n_zy) x H] natural order
    float* arena; size_t arena_floats;
    float* arena16;                                    // the same images with every weight as an fp16 (hi, lo) pair
    void* arena_b3;                                    // ... as three bf16 planes, every image at 1.5 x its fp32 offset (hap_gemm.hpp WeightMap)
};

// workspace of one pass of `chunk` sites (floats): packed inputs of both encoders, h of all steps / both directions for two
// layers per encoder (ping-pong), c for up to 4 z-slices, the concatenated projections, the dense output.  ~195 KB per site.
// planes3: the activation images that feed a GEMM (packed inputs, h, the concatenated projections) hold three bf16 planes per element
// (the bf16x3 mode: 6 bytes instead of 4); the cell state and the dense output stay fp32
struct HapWsLayout { size_t xT_f, hbuf_f, c_f, cat_f, inner_f, bytes; };
static HapWsLayout hap_ws_layout(int64_t chunk, int nk_in0, bool planes3)
{
    const size_t max_tiles = (size_t)(chunk / TS);
    const size_t tile_act = planes3 ? TILE_F3 : TILE_F;
    HapWsLayout w;
    w.xT_f = (size_t)33 * max_tiles * nk_in0 * tile_act;
    w.hbuf_f = (size_t)33 * max_tiles * 2 * 16 * tile_act;
    w.c_f = (size_t)4 * max_tiles * 16 * TILE_F;
    w.cat_f = max_tiles * 32 * tile_act;
    w.inner_f = max_tiles * 16 * TILE_F;
    w.bytes = (2 * w.xT_f + 4 * w.hbuf_f + w.c_f + w.cat_f + w.inner_f) * sizeof(float);
    return w;
}

// (Re)allocates the workspace for the context's pass size.  Synchronous; called where the caller already expects a
// synchronous call (nsnp_hap_load_weights, nsnp_ctx_set_option("hap_pass_sites")), never from nsnp_hap_forward.
int nsnp_hap_reserve(nsnp_ctx* ctx)
{
    if (!ctx->hw) return NSNP_OK;                       // sized when the weights arrive
    // sized for the arithmetic selected now; the larger bf16x3 images only when that mode is (or has been) chosen: the workspace never
    // shrinks for a change of arithmetic alone (callers switch back and forth), it is re-made when the pass size changes
    const HapWsLayout w = hap_ws_layout(ctx->hap_chunk, ctx->hw->nk_in0, ctx->hap_precision == 2 && NSNP_HAP_BS);
    if (ctx->hap_ws && ctx->hap_ws_chunk == ctx->hap_chunk && ctx->hap_ws_bytes >= w.bytes) return NSNP_OK;
    NSNP_HIP(ctx, hipSetDevice(ctx->device));
    NSNP_HIP(ctx, hipDeviceSynchronize());
    if (ctx->hap_ws) (void)hipFree(ctx->hap_ws);
    ctx->hap_ws = nullptr; ctx->hap_ws_bytes = 0;
    if (hipMalloc(&ctx->hap_ws, w.bytes) != hipSuccess) { ctx->hap_ws = nullptr; ctx->last_err = hipErrorOutOfMemory; return NSNP_ENOMEM; }
    ctx->hap_ws_bytes = w.bytes; ctx->hap_ws_chunk = ctx->hap_chunk;
    return NSNP_OK;
}

void nsnp_hap_free(nsnp_ctx* ctx)
{
    if (ctx->hw) {
        if (ctx->hw->arena) (void)hipFree(ctx->hw->arena);
        if (ctx->hw->arena16) (void)hipFree(ctx->hw->arena16);
        if (ctx->hw->arena_b3) (void)hipFree(ctx->hw->arena_b3);
        delete ctx->hw; ctx->hw = nullptr;
    }
    if (ctx->hap_ws) { (void)hipFree(ctx->hap_ws); ctx->hap_ws = nullptr; ctx->hap_ws_bytes = 0; }
}

namespace {

// image element [row_tile][chunk][row 0..127][p 0..15]
template <typename RowFn, typename ColFn>
void pack_rows(float* img, int n_row_tiles, int n_chunks, const float* W, int ld, RowFn rowf, ColFn colf)
{
    for (int by = 0; by < n_row_tiles; ++by)
        for (int kc = 0; kc < n_chunks; ++kc)
            for (int row = 0; row < TR; ++row)
                for (int p = 0; p < BK; ++p) {
                    const int tr = rowf(by * TR + row);
                    const int tc = colf(kc, p);
                    img[(((size_t)by * n_chunks + kc) * TR + row) * BK + p] = (tr >= 0 && tc >= 0) ? W[(size_t)tr * ld + tc] : 0.f;
                }
}

}  // namespace

extern "C" int nsnp_hap_load_weights(nsnp_ctx* ctx, const float* const* t, int n_tensors,
                                     int n_features, int hidden, int n_layers, int n_gt, int n_zy)
{
    if (!ctx || !t) return NSNP_EINVAL;
    if (hidden != 256 || n_layers != 3 || n_features <= 0 || n_features > 128 || n_gt <= 0 || n_zy <= 0 || n_gt + n_zy > 16)
        return NSNP_ESHAPE;                            // kernels are built for ont_haplotype.yaml:7-16
    const int per_enc = n_layers * 2 * 4 + 2;
    if (n_tensors < 2 * per_enc + 6) return NSNP_EINVAL;
    for (int i = 0; i < 2 * per_enc + 6; ++i) if (!t[i]) return NSNP_EINVAL;
    NSNP_HIP(ctx, hipSetDevice(ctx->device));
    const int H = hidden, F = n_features, G = 4 * H;
    const int nk0 = NSNP_CDIV(F, BK);
    const int n_rt = G / TR;                           // 8 row tiles of gates
    // LSTM image rows: image row R -> unit = R/4, gate = R%4 -> torch row gate*H + unit
    auto lstm_row = [H](int R) { return (R & 3) * H + (R >> 2); };
    auto nat_row = [](int R) { return R; };
    size_t total = 0;
    size_t off_w[2][3][2], off_b[2][3][2], off_pw[2], off_pb[2];
    for (int e = 0; e < 2; ++e)
        for (int l = 0; l < 3; ++l)
            for (int d = 0; d < 2; ++d) {
                const int nk = (l == 0 ? nk0 : 2 * H / BK) + H / BK;
                off_w[e][l][d] = total; total += (size_t)n_rt * nk * TILE_F;
                off_b[e][l][d] = total; total += G;
            }
    for (int e = 0; e < 2; ++e) { off_pw[e] = total; total += (size_t)(H / TR) * (2 * H / BK) * TILE_F; off_pb[e] = total; total += H; }
    const size_t off_dw = total; total += (size_t)(H / TR) * (2 * H / BK) * TILE_F;
    const size_t off_db = total; total += H;
    const size_t off_hw = total; total += (size_t)(n_gt + n_zy) * H;
    const size_t off_hb = total; total += 16;
    std::vector<float> host(total, 0.f);
    for (int e = 0; e < 2; ++e) {
        for (int l = 0; l < 3; ++l)
            for (int d = 0; d < 2; ++d) {
                const float* const* q = t + e * per_enc + (l * 2 + d) * 4;
                const int I = l == 0 ? F : 2 * H;
                const int nki = l == 0 ? nk0 : 2 * H / BK, nkh = H / BK, nk = nki + nkh;
                // one image with both column blocks: build [W_ih | W_hh] chunk maps
                float* img = host.data() + off_w[e][l][d];
                for (int by = 0; by < n_rt; ++by)
                    for (int kc = 0; kc < nk; ++kc)
                        for (int row = 0; row < TR; ++row)
                            for (int p = 0; p < BK; ++p) {
                                const int tr = lstm_row(by * TR + row);
                                float v = 0.f;
                                if (kc < nki) {
                                    int col;
                                    if (l == 0) col = kc * BK + p;                                          // natural features
                                    else col = (kc / (H / BK)) * H + (kc % (H / BK)) * BK + unit_of_pos(p); // prev layer h, storage order
                                    if (col < I) v = q[0][(size_t)tr * I + col];
                                } else {
                                    const int col = (kc - nki) * BK + unit_of_pos(p);
                                    v = q[1][(size_t)tr * H + col];
                                }
                                img[(((size_t)by * nk + kc) * TR + row) * BK + p] = lstm_gate_scale(by * TR + row) * v;
                            }
                float* bz = host.data() + off_b[e][l][d];
                for (int R = 0; R < G; ++R) bz[R] = lstm_gate_scale(R) * (q[2][lstm_row(R)] + q[3][lstm_row(R)]);
            }
        // output_proj: Linear(2H -> H) on [h_fwd ; h_bwd] of the last layer at the centre step
        const float* pw = t[e * per_enc + per_enc - 2]; const float* pb = t[e * per_enc + per_enc - 1];
        pack_rows(host.data() + off_pw[e], H / TR, 2 * H / BK, pw, 2 * H, nat_row,
                  [H](int kc, int p) { return (kc / (H / BK)) * H + (kc % (H / BK)) * BK + unit_of_pos(p); });
        memcpy(host.data() + off_pb[e], pb, sizeof(float) * H);
    }
    const float* const* fw = t + 2 * per_enc;
    pack_rows(host.data() + off_dw, H / TR, 2 * H / BK, fw[0], 2 * H, nat_row, [](int kc, int p) { return kc * BK + p; });
    memcpy(host.data() + off_db, fw[1], sizeof(float) * H);
    memcpy(host.data() + off_hw, fw[2], sizeof(float) * (size_t)n_gt * H);
    memcpy(host.data() + off_hw + (size_t)n_gt * H, fw[4], sizeof(float) * (size_t)n_zy * H);
    memcpy(host.data() + off_hb, fw[3], sizeof(float) * n_gt);
    memcpy(host.data() + off_hb + n_gt, fw[5], sizeof(float) * n_zy);

    if (!ctx->hw) { ctx->hw = new (std::nothrow) HapWeightsDev(); if (!ctx->hw) return NSNP_ENOMEM; memset((void*)ctx->hw, 0, sizeof(HapWeightsDev)); }
    HapWeightsDev& hw = *ctx->hw;
    if (hw.arena_floats != total) {     // both arenas are sized by `total`: a reload with other dimensions re-allocates both
        if (hw.arena) { (void)hipFree(hw.arena); hw.arena = nullptr; }
        if (hw.arena16) { (void)hipFree(hw.arena16); hw.arena16 = nullptr; }
        if (hw.arena_b3) { (void)hipFree(hw.arena_b3); hw.arena_b3 = nullptr; }
        hw.arena_floats = 0;
    }
    if (!hw.arena) NSNP_HIP(ctx, hipMalloc((void**)&hw.arena, total * sizeof(float)));
    NSNP_HIP(ctx, hipMemcpy(hw.arena, host.data(), total * sizeof(float), hipMemcpyHostToDevice));
    {
        // f16x3 images: rows of 16 weights become 16 hi halves + 16 lo halves (same 64 bytes); biases and the
        // head matrix stay fp32
        std::vector<float> h16(host);
        auto conv = [&](size_t off, size_t n_floats) {
            for (size_t r = 0; r + 16 <= n_floats; r += 16) {
                _Float16* dst = reinterpret_cast<_Float16*>(h16.data() + off + r);
                const float* src = host.data() + off + r;
                for (int k = 0; k < 16; ++k) {
                    const _Float16 hi = (_Float16)src[k];
                    dst[k] = hi; dst[16 + k] = (_Float16)(src[k] - (float)hi);
                }
            }
        };
        for (int e = 0; e < 2; ++e) {
            for (int l = 0; l < 3; ++l) for (int d = 0; d < 2; ++d)
                conv(off_w[e][l][d], (size_t)n_rt * ((l == 0 ? nk0 : 2 * H / BK) + H / BK) * TILE_F);
            conv(off_pw[e], (size_t)(H / TR) * (2 * H / BK) * TILE_F);
        }
        conv(off_dw, (size_t)(H / TR) * (2 * H / BK) * TILE_F);
        if (!hw.arena16) NSNP_HIP(ctx, hipMalloc((void**)&hw.arena16, total * sizeof(float)));
        hw.arena_floats = total;
        NSNP_HIP(ctx, hipMemcpy(hw.arena16, h16.data(), total * sizeof(float), hipMemcpyHostToDevice));
    }
    {
        // bf16x3 images: every weight as three bf16 planes (rows of 16 weights -> 16 p0 | 16 p1 | 16 p2), image at 1.5 x its fp32 offset
        std::vector<uint16_t> h3(total * 3, 0);
        for (int e = 0; e < 2; ++e) {
            for (int l = 0; l < 3; ++l) for (int d = 0; d < 2; ++d)
                pack_bf16x3_rows(host.data(), h3.data(), off_w[e][l][d], (size_t)n_rt * ((l == 0 ? nk0 : 2 * H / BK) + H / BK) * TILE_F);
            pack_bf16x3_rows(host.data(), h3.data(), off_pw[e], (size_t)(H / TR) * (2 * H / BK) * TILE_F);
        }
        pack_bf16x3_rows(host.data(), h3.data(), off_dw, (size_t)(H / TR) * (2 * H / BK) * TILE_F);
        if (!hw.arena_b3) NSNP_HIP(ctx, hipMalloc(&hw.arena_b3, total * 6));
        NSNP_HIP(ctx, hipMemcpy(hw.arena_b3, h3.data(), total * 6, hipMemcpyHostToDevice));
    }
    hw.F = F; hw.H = H; hw.n_layers = n_layers; hw.n_gt = n_gt; hw.n_zy = n_zy; hw.nk_in0 = nk0;
    for (int e = 0; e < 2; ++e) {
        for (int l = 0; l < 3; ++l) for (int d = 0; d < 2; ++d) { hw.w[e][l][d] = hw.arena + off_w[e][l][d]; hw.b[e][l][d] = hw.arena + off_b[e][l][d]; }
        hw.proj_w[e] = hw.arena + off_pw[e]; hw.proj_b[e] = hw.arena + off_pb[e];
    }
    hw.dense_w = hw.arena + off_dw; hw.dense_b = hw.arena + off_db; hw.head_w = hw.arena + off_hw; hw.head_b = hw.arena + off_hb;
    return nsnp_hap_reserve(ctx);                      // the forward itself never allocates (graph capture, no stream stalls)
}

// AR: 0 exact fp32, 1 f16x3 (activation images hold (hi, lo) fp16 pairs), 2 bf16x3 (weights AND activation images as three bf16 planes)
template <int AR>
static int hap_forward_t(nsnp_ctx* ctx, const float* xp, const float* xh, int64_t N, float* gt_prob, float* zy_prob, hipStream_t s)
{
    const HapWeightsDev& hw = *ctx->hw;
    const int H = hw.H, F = hw.F;
    constexpr bool F16 = AR == 1;                              // element format of the activation images: (hi, lo) fp16 pairs ...
    constexpr bool BS = AR == 2 && NSNP_HAP_BS;                // ... or three bf16 planes (written by the producing epilogue: hap_gemm.hpp)
    constexpr size_t TILE_A = BS ? TILE_F3 : TILE_F;           // floats of one [128][16] activation tile image
    const WeightMap wm{hw.arena, AR == 1 ? (const void*)hw.arena16 : (const void*)hw.arena_b3, AR};
    const int Lp = 33, Lh = 11;                       // ont_haplotype.yaml:10-11
    // sites per pass: every time step of a layer is one launch over all sites of the pass, so the pass size sets how long
    // each of the 83 dependent launches is (16384 sites: 2048-4096 workgroups, 2-4 rounds of the chip; ramp + tail of a launch
    // ~5 % instead of ~18 % at 4096).  Option "hap_pass_sites"; workspace ~195 KB per site, allocated at load time.
    const int64_t chunk = ctx->hap_chunk;
    const HapWsLayout wl = hap_ws_layout(chunk, hw.nk_in0, BS);
    if (!ctx->hap_ws || ctx->hap_ws_bytes < wl.bytes) return NSNP_ENOMEM;      // nsnp_hap_load_weights / nsnp_ctx_set_option reserve it
    const size_t xT_f = wl.xT_f, hbuf_f = wl.hbuf_f, c_f = wl.c_f, cat_f = wl.cat_f;
    float* base = (float*)ctx->hap_ws;
    float* xT[2] = {base, base + xT_f};
    float* hb[2][2] = {{base + 2 * xT_f, base + 2 * xT_f + hbuf_f}, {base + 2 * xT_f + 2 * hbuf_f, base + 2 * xT_f + 3 * hbuf_f}};
    float* cst = base + 2 * xT_f + 4 * hbuf_f;
    float* cat = cst + c_f;
    float* inner = cat + cat_f;

    for (int64_t n0 = 0; n0 < N; n0 += chunk) {
        const int64_t n = N - n0 < chunk ? N - n0 : chunk;
        const int n_tiles = (int)NSNP_CDIV(n, TS);
        const int Ls[2] = {Lp, Lh};
        const float* xin[2] = {xp + n0 * F * Lp, xh + n0 * F * Lh};
        for (int e = 0; e < 2; ++e) {
            const unsigned blocks = (unsigned)n_tiles * hw.nk_in0 * (TS / PK_SITES);
            hipLaunchKernelGGL(k_hap_pack_input<(AR == 2 ? (BS ? 2 : 0) : AR)>, dim3(blocks), dim3(256), 0, s, xin[e], n, F, Ls[e], n_tiles, hw.nk_in0, xT[e]);
        }
        // h of one layer: [t][site tile][dir][16 chunks][128][16] -> the 32 chunks [h_fwd ; h_bwd] of a site
        // tile at time t are contiguous (what the next layer and output_proj consume)
        const size_t tile_h = (size_t)16 * TILE_A;                 // one direction of one site tile at one step
        const size_t tile_c = (size_t)16 * TILE_F;                 // its cell state (always fp32)
        const size_t step_h = (size_t)n_tiles * 2 * tile_h;
        ScopedKernelTimer tm_lstm(ctx, NSNP_K_HAPLSTM, s);         // one event pair around the 83 fused step launches of this pass
        for (int l = 0; l < 3; ++l) {
            // encoder e reads its layer input from (l == 0 ? xT[e] : hb[e][(l-1)&1]) and writes hb[e][l&1]
            const int steps[2] = {l == 2 ? Lp / 2 + 1 : Lp, l == 2 ? Lh / 2 + 1 : Lh};
            const int max_steps = steps[0] > steps[1] ? steps[0] : steps[1];
            for (int st = 0; st < max_steps; ++st) {
                StepLaunch L; int nz = 0;
                for (int e = 0; e < 2; ++e) {
                    if (st >= steps[e]) continue;
                    for (int d = 0; d < 2; ++d) {
                        const int t = d ? Ls[e] - 1 - st : st;
                        const int tprev = d ? t + 1 : t - 1;
                        StepArgs& a = L.z[nz];
                        a.w = wm(hw.w[e][l][d]); a.bias = hw.b[e][l][d];
                        if (l == 0) {
                            a.in0 = xT[e] + (size_t)t * n_tiles * hw.nk_in0 * TILE_A;
                            a.nk0 = hw.nk_in0; a.in0_tile_stride = (int)(hw.nk_in0 * TILE_A);
                        } else {
                            a.in0 = hb[e][(l - 1) & 1] + (size_t)t * step_h;
                            a.nk0 = 2 * H / BK; a.in0_tile_stride = (int)(2 * tile_h);
                        }
                        a.in1 = st ? hb[e][l & 1] + (size_t)tprev * step_h + (size_t)d * tile_h : nullptr;
                        a.nk1 = st ? H / BK : 0; a.in1_tile_stride = (int)(2 * tile_h);
                        a.nk_img = a.nk0 + H / BK;
                        a.out = hb[e][l & 1] + (size_t)t * step_h + (size_t)d * tile_h;
                        a.out_tile_stride = (int)(2 * tile_h);
                        a.cstate = cst + (size_t)(e * 2 + d) * n_tiles * tile_c;
                        a.c_tile_stride = (int)tile_c;
                        a.first = st == 0;
                        ++nz;
                    }
                }
                launch_hap_gemm<MODE_LSTM, AR, false, BS, BS>(ctx, s, L, (int)(n_tiles), 4 * H / TR, nz);
            }
        }
        tm_lstm.stop();
        // output_proj at the centre step of the last layer (which wrote hb[e][0]), both encoders in one
        // launch -> cat image: 32 chunks per site tile = [proj_pileup(256) ; proj_haplotype(256)]
        {
            StepLaunch L;
            for (int e = 0; e < 2; ++e) {
                StepArgs& a = L.z[e];
                a.w = wm(hw.proj_w[e]); a.bias = hw.proj_b[e];
                a.in0 = hb[e][0] + (size_t)(Ls[e] / 2) * step_h;
                a.nk0 = 2 * H / BK; a.in0_tile_stride = (int)(2 * tile_h);
                a.in1 = nullptr; a.nk1 = 0; a.in1_tile_stride = 0; a.nk_img = a.nk0;
                a.out = cat + (size_t)e * 16 * TILE_A; a.out_tile_stride = (int)(32 * TILE_A);
                a.cstate = nullptr; a.c_tile_stride = 0; a.first = 0;
            }
            launch_hap_gemm<MODE_LINEAR, AR, false, BS, BS>(ctx, s, L, (int)(n_tiles), H / TR, 2);
        }
        {
            StepLaunch L; StepArgs& a = L.z[0];
            a.w = wm(hw.dense_w); a.bias = hw.dense_b; a.in0 = cat; a.nk0 = 2 * H / BK; a.in0_tile_stride = (int)(32 * TILE_A);
            a.in1 = nullptr; a.nk1 = 0; a.in1_tile_stride = 0; a.nk_img = a.nk0; a.out = inner; a.out_tile_stride = 16 * TILE_F;
            a.cstate = nullptr; a.c_tile_stride = 0; a.first = 0;
            launch_hap_gemm<MODE_LINEAR_TANH, AR, false, BS, false>(ctx, s, L, (int)(n_tiles), H / TR, 1);      // (its output feeds the heads kernel: fp32)
        }
        hipLaunchKernelGGL(k_hap_heads<F16>, dim3((unsigned)NSNP_CDIV(n, 4)), dim3(256), 0, s, inner, n, hw.head_w, hw.head_b,
                           hw.n_gt, hw.n_zy, gt_prob + n0 * hw.n_gt, zy_prob + n0 * hw.n_zy);
    }
    NSNP_HIP(ctx, hipGetLastError());
    return NSNP_OK;
}

extern "C" int nsnp_hap_forward(nsnp_ctx* ctx, const float* xp, const float* xh, int64_t N,
                                float* gt_prob, float* zy_prob, void* stream)
{
    if (!ctx || N < 0 || (N > 0 && (!xp || !xh || !gt_prob || !zy_prob))) return NSNP_EINVAL;
    if (!ctx->hw) return NSNP_ENOWEIGHTS;
    if (N == 0) return NSNP_OK;
    hipStream_t s = (hipStream_t)stream;
    if (ctx->hap_precision == 1) return hap_forward_t<1>(ctx, xp, xh, N, gt_prob, zy_prob, s);
    if (ctx->hap_precision == 2) return hap_forward_t<2>(ctx, xp, xh, N, gt_prob, zy_prob, s);
    return hap_forward_t<0>(ctx, xp, xh, N, gt_prob, zy_prob, s);
}

NSNP_DEVCLK_READER(nsnp_devclk_read_hap)
