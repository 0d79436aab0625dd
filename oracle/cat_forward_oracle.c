/*
 * oracle/cat_forward_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE (see oracle.h).
 *
 * Restates the legacy haplotype caller CatModel.predict (HaplotypeModel/model.py:332-358):
 *   calculate_percentage                 model.py:186-194  per (column, tag): share of A,C,G,T,deletion among
 *                                                           the 20 reads of the tag that are not padding (-2)
 *   haplotype_percentage = RNN           model.py:118-129  nn.LSTM(20,256,3 layers,bidirectional) + Linear(512,256),
 *                                                           row [5] of the [L,N,256] output
 *   haplotype_base = ResCRNN             crnn.py:118-190   six ResBlocks (crnn.py:84-115: conv3x3-BN-ReLU-conv3x3-BN
 *                                                           + 1x1 shortcut conv, add, ReLU), four max-pools, then two
 *                                                           BidirectionalLSTM (crnn.py:5-21: 1-layer BiLSTM + Linear)
 *                                                           over the 11 columns, row [5]
 *   out_layer Linear(512,10) + softmax   model.py:354-357
 * BatchNorm is in eval mode (predict.py:28 model.eval()): y = (x - mean) / sqrt(var + 1e-5) * w + b.
 * Trained weights are absent from the reference tree; goldens use the reference module with seeded weights.
 *
 * weights: the floating-point tensors of CatModel.state_dict() in order (num_batches_tracked skipped):
 *   6 ResBlocks x 14 (conv1 w,b; bn1 w,b,mean,var; conv2 w,b; bn2 w,b,mean,var; shortcut w,b)      = 84
 *   haplotype_base.rnn.0: w_ih,w_hh,b_ih,b_hh fwd; the same reverse; embedding w,b                  = 10
 *   haplotype_base.rnn.1: the same                                                                  = 10
 *   haplotype_percentage.rnn: 3 layers x 2 dirs x 4; out_layer w,b                                  = 26
 *   out_layer w,b                                                                                   =  2   (132)
 */
#include "oracle.h"
#include "lstm_internal.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

enum { CAT_H = 40, CAT_W = 11, CAT_P = 5, CAT_NH = 256 };

/* x [Cin][H][W] -> y [Cout][H][W], kernel k (1 or 3), stride 1, zero padding k/2 (nn.Conv2d) */
static void conv2d(const float* x, int Cin, int H, int W, const float* w, const float* b, int Cout, int k, float* y)
{
    const int pad = k / 2;
    for (int co = 0; co < Cout; ++co)
        for (int yy = 0; yy < H; ++yy)
            for (int xx = 0; xx < W; ++xx) {
                float acc = b[co];
                for (int ci = 0; ci < Cin; ++ci)
                    for (int ky = 0; ky < k; ++ky) {
                        const int sy = yy + ky - pad;
                        if (sy < 0 || sy >= H) continue;
                        for (int kx = 0; kx < k; ++kx) {
                            const int sx = xx + kx - pad;
                            if (sx < 0 || sx >= W) continue;
                            acc += w[(((size_t)co * Cin + ci) * k + ky) * k + kx] * x[((size_t)ci * H + sy) * W + sx];
                        }
                    }
                y[((size_t)co * H + yy) * W + xx] = acc;
            }
}

static void batchnorm(float* x, int C, int HW, const float* w, const float* b, const float* mean, const float* var)
{
    for (int c = 0; c < C; ++c) {
        const float inv = 1.0f / sqrtf(var[c] + 1e-5f);
        for (int i = 0; i < HW; ++i) x[(size_t)c * HW + i] = (x[(size_t)c * HW + i] - mean[c]) * inv * w[c] + b[c];
    }
}

static void relu(float* x, size_t n) { for (size_t i = 0; i < n; ++i) x[i] = x[i] > 0.f ? x[i] : 0.f; }

/* ResBlock (crnn.py:84-115); q: the block's 14 tensors */
static void resblock(const float* const* q, const float* x, int Cin, int Cout, int H, int W, float* out)
{
    const int HW = H * W;
    float* y1 = (float*)malloc(sizeof(float) * (size_t)Cout * HW);
    float* y2 = (float*)malloc(sizeof(float) * (size_t)Cout * HW);
    conv2d(x, Cin, H, W, q[0], q[1], Cout, 3, y1);
    batchnorm(y1, Cout, HW, q[2], q[3], q[4], q[5]);
    relu(y1, (size_t)Cout * HW);
    conv2d(y1, Cout, H, W, q[6], q[7], Cout, 3, y2);
    batchnorm(y2, Cout, HW, q[8], q[9], q[10], q[11]);
    conv2d(x, Cin, H, W, q[12], q[13], Cout, 1, out);
    for (size_t i = 0; i < (size_t)Cout * HW; ++i) out[i] += y2[i];
    relu(out, (size_t)Cout * HW);
    free(y1); free(y2);
}

/* nn.MaxPool2d(kernel (kh,3), stride (kh,1), padding (0,1)): crnn.py:134-160 */
static void maxpool(const float* x, int C, int H, int W, int kh, float* y, int* Ho_out)
{
    const int Ho = (H - kh) / kh + 1;
    for (int c = 0; c < C; ++c)
        for (int yo = 0; yo < Ho; ++yo)
            for (int xo = 0; xo < W; ++xo) {
                float m = -INFINITY;
                for (int ky = 0; ky < kh; ++ky)
                    for (int kx = -1; kx <= 1; ++kx) {
                        const int sx = xo + kx;
                        if (sx < 0 || sx >= W) continue;
                        const float v = x[((size_t)c * H + yo * kh + ky) * W + sx];
                        if (v > m) m = v;
                    }
                y[((size_t)c * Ho + yo) * W + xo] = m;
            }
    *Ho_out = Ho;
}

typedef struct { float* wih_t; float* whh_t; const float* b_ih; const float* b_hh; } dirw_t;

static void prep_dir(dirw_t* p, const float* const* q, int I, int H)
{
    p->wih_t = orc_transpose_(q[0], 4 * H, I);
    p->whh_t = orc_transpose_(q[1], 4 * H, H);
    p->b_ih = q[2]; p->b_hh = q[3];
}

void orc_cat_forward(const float* const* w, const float* g0 /*[N,40,11,5]*/, const float* g1, int64_t N,
                     float* gt_prob /*[N,10]*/, int nthreads)
{
    const int NH = CAT_NH, L = CAT_W;
    static const int chan[7] = {10, 32, 64, 128, 128, 256, 256};
    const float* const* wb = w;                 /* 6 x 14 */
    const float* const* wr0 = w + 84;           /* BidirectionalLSTM 0 */
    const float* const* wr1 = w + 94;           /* BidirectionalLSTM 1 */
    const float* const* wp = w + 104;           /* percentage RNN */
    const float* const* wo = w + 130;           /* out_layer */
    dirw_t r0[2], r1[2], pr[6];
    for (int d = 0; d < 2; ++d) { prep_dir(&r0[d], wr0 + 4 * d, NH, NH); prep_dir(&r1[d], wr1 + 4 * d, NH, NH); }
    for (int l = 0; l < 3; ++l) for (int d = 0; d < 2; ++d) prep_dir(&pr[l * 2 + d], wp + (l * 2 + d) * 4, l == 0 ? 20 : 2 * NH, NH);
    if (nthreads <= 0) nthreads = 1;
    #pragma omp parallel for num_threads(nthreads) schedule(dynamic, 1)
    for (int64_t n = 0; n < N; ++n) {
        const float* gs[2] = {g0 + (size_t)n * CAT_H * CAT_W * CAT_P, g1 + (size_t)n * CAT_H * CAT_W * CAT_P};
        float cat[2 * CAT_NH];
        /* ---- percentage branch (model.py:339-353) ---- */
        {
            float* cur = (float*)calloc((size_t)L * 2 * NH, sizeof(float));
            float* nxt = (float*)calloc((size_t)L * 2 * NH, sizeof(float));
            for (int t = 0; t < L; ++t)
                for (int g = 0; g < 2; ++g)
                    for (int tag = 0; tag < 2; ++tag) {
                        int cnt[5] = {0, 0, 0, 0, 0}, valid = 0;
                        for (int r = 0; r < 20; ++r) {
                            const float v = gs[g][(((size_t)(tag * 20 + r)) * CAT_W + t) * CAT_P + 0];
                            if (v != -2.0f) ++valid;
                            if (v == 1.0f) ++cnt[0]; else if (v == 2.0f) ++cnt[1]; else if (v == 3.0f) ++cnt[2];
                            else if (v == 4.0f) ++cnt[3]; else if (v == -1.0f) ++cnt[4];
                        }
                        /* int64 / (int64 + 1e-9) is evaluated in float32 by torch (type promotion with a Python
                         * scalar): the denominator is float32(valid) + 1e-9f */
                        const float den = (float)valid + 1e-9f;
                        for (int k = 0; k < 5; ++k) cur[t * 20 + (g * 2 + tag) * 5 + k] = (float)cnt[k] / den;
                    }
            int I = 20;
            for (int l = 0; l < 3; ++l) {
                for (int d = 0; d < 2; ++d) {
                    const dirw_t* p = &pr[l * 2 + d];
                    orc_lstm_dir_t_(cur, L, I, NH, p->wih_t, p->whh_t, p->b_ih, p->b_hh, d, L, nxt, 2 * NH, d * NH);
                }
                memcpy(cur, nxt, sizeof(float) * (size_t)L * 2 * NH);
                I = 2 * NH;
            }
            orc_linear(cur + (size_t)5 * 2 * NH, 2 * NH, wp[24], wp[25], NH, cat);
            free(cur); free(nxt);
        }
        /* ---- ResCRNN branch ---- */
        {
            int H = CAT_H; const int W = CAT_W;
            float* a = (float*)malloc(sizeof(float) * 256 * CAT_H * CAT_W);
            float* b = (float*)malloc(sizeof(float) * 256 * CAT_H * CAT_W);
            /* torch.cat((g0_s, g1_s), 1): channel = g*5 + plane  (model.py:333-336,352) */
            for (int g = 0; g < 2; ++g)
                for (int p = 0; p < CAT_P; ++p)
                    for (int y = 0; y < CAT_H; ++y)
                        for (int x = 0; x < CAT_W; ++x)
                            a[((size_t)(g * 5 + p) * CAT_H + y) * CAT_W + x] = gs[g][((size_t)y * CAT_W + x) * CAT_P + p];
            static const int pool_after[6] = {2, 2, 0, 3, 0, 2};   /* kernel height of the pool following block i (0: none) */
            for (int i = 0; i < 6; ++i) {
                resblock(wb + 14 * i, a, chan[i], chan[i + 1], H, W, b);
                if (pool_after[i]) { int Ho; maxpool(b, chan[i + 1], H, W, pool_after[i], a, &Ho); H = Ho; }
                else { float* t = a; a = b; b = t; }
            }
            /* a: [256][1][11] -> sequence [11][256] (crnn.py:184-185) */
            float* seq = (float*)malloc(sizeof(float) * (size_t)L * NH);
            for (int t = 0; t < L; ++t) for (int c = 0; c < NH; ++c) seq[t * NH + c] = a[(size_t)c * W + t];
            float* rec = (float*)malloc(sizeof(float) * (size_t)L * 2 * NH);
            float* emb = (float*)malloc(sizeof(float) * (size_t)L * NH);
            for (int d = 0; d < 2; ++d) orc_lstm_dir_t_(seq, L, NH, NH, r0[d].wih_t, r0[d].whh_t, r0[d].b_ih, r0[d].b_hh, d, L, rec, 2 * NH, d * NH);
            for (int t = 0; t < L; ++t) orc_linear(rec + (size_t)t * 2 * NH, 2 * NH, wr0[8], wr0[9], NH, emb + (size_t)t * NH);
            for (int d = 0; d < 2; ++d) orc_lstm_dir_t_(emb, L, NH, NH, r1[d].wih_t, r1[d].whh_t, r1[d].b_ih, r1[d].b_hh, d, L, rec, 2 * NH, d * NH);
            orc_linear(rec + (size_t)5 * 2 * NH, 2 * NH, wr1[8], wr1[9], NH, cat + NH);
            free(a); free(b); free(seq); free(rec); free(emb);
        }
        orc_linear(cat, 2 * NH, wo[0], wo[1], 10, gt_prob + n * 10);
        orc_softmax(gt_prob + n * 10, 10);
    }
    for (int d = 0; d < 2; ++d) { free(r0[d].wih_t); free(r0[d].whh_t); free(r1[d].wih_t); free(r1[d].whh_t); }
    for (int i = 0; i < 6; ++i) { free(pr[i].wih_t); free(pr[i].whh_t); }
}

/* PredictDataset.__getitem__ group assembly (HaplotypeModel/dataset.py:862-915): per tag the first 20 rows of the
 * read / base-quality / mapping-quality matrices [N][D][L], mask = (read != -2), phase = 1 (tag 1) or 2 (tag 2),
 * concatenated on the row axis then stacked on a new last axis -> [N][40][L][5]. */
void orc_cat_groups(const int32_t* r1, const int32_t* q1, const int32_t* m1, int D1,
                    const int32_t* r2, const int32_t* q2, const int32_t* m2, int D2,
                    int64_t N, int L, float* g)
{
    for (int64_t n = 0; n < N; ++n)
        for (int row = 0; row < 40; ++row)
            for (int x = 0; x < L; ++x) {
                const int tag = row >= 20, rr = row - 20 * tag, D = tag ? D2 : D1;
                const size_t src = ((size_t)n * D + rr) * L + x;
                const int32_t base = (tag ? r2 : r1)[src];
                float* o = g + (((size_t)n * 40 + row) * L + x) * 5;
                o[0] = (float)base; o[1] = (float)(tag ? q2 : q1)[src]; o[2] = (float)(tag ? m2 : m1)[src];
                o[3] = base != -2 ? 1.f : 0.f; o[4] = (float)(tag + 1);
            }
}
