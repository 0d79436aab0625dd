/*
 * oracle/hap_forward_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE (see oracle.h).
 *
 * Restates model_dev.LSTMNetwork.predict (HaplotypeModel/model_dev.py:133-143):
 *   permute(0,2,1)                       :136-137  [N,F,L] -> [N,L,F]
 *   BaseEncoder x2                       :59-84    nn.LSTM(F,H,n_layers,bidirectional) + Linear(2H,H) on all steps
 *   ForwardLayer                         :86-105   take [:, L//2, :] of each, concat, tanh(Linear(2H,H)),
 *                                                  genotype Linear(H,n_gt), zygosity Linear(H,n_zy)
 *   softmax on both heads                :140-141
 * Trained weights are absent from the reference tree (.MISSING_LARGE_BLOBS); goldens use the
 * reference module with seeded random weights.
 */
#include "oracle.h"
#include "lstm_internal.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct { float* wih_t; float* whh_t; const float* b_ih; const float* b_hh; } dirw_t;

static void encoder(const dirw_t* dw, const float* proj_w, const float* proj_b,
                    const float* x_fl /*[F][L]*/, int F, int L, int H, int n_layers, float* center_out /*[H]*/)
{
    float* cur = (float*)malloc(sizeof(float) * (size_t)L * (size_t)(F > 2 * H ? F : 2 * H));
    float* nxt = (float*)malloc(sizeof(float) * (size_t)L * 2 * (size_t)H);
    for (int l = 0; l < L; ++l) for (int f = 0; f < F; ++f) cur[l * F + f] = x_fl[f * L + l];
    int I = F;
    for (int layer = 0; layer < n_layers; ++layer) {
        for (int d = 0; d < 2; ++d) {
            const dirw_t* p = &dw[layer * 2 + d];
            orc_lstm_dir_t_(cur, L, I, H, p->wih_t, p->whh_t, p->b_ih, p->b_hh, d, L, nxt, 2 * H, d * H);
        }
        memcpy(cur, nxt, sizeof(float) * (size_t)L * 2 * (size_t)H);
        I = 2 * H;
    }
    orc_linear(cur + (size_t)(L / 2) * 2 * H, 2 * H, proj_w, proj_b, H, center_out);
    free(cur); free(nxt);
}

void orc_hap_forward(const float* const* w, const float* xp, const float* xh, int64_t N,
                     int F, int H, int n_layers, int Lp, int Lh, int n_gt, int n_zy,
                     float* gt_prob, float* zy_prob, int nthreads)
{
    const int per_enc = n_layers * 2 * 4 + 2;
    dirw_t* dw[2];
    for (int e = 0; e < 2; ++e) {
        dw[e] = (dirw_t*)malloc(sizeof(dirw_t) * (size_t)n_layers * 2);
        for (int layer = 0; layer < n_layers; ++layer)
            for (int d = 0; d < 2; ++d) {
                const float* const* q = w + e * per_enc + (layer * 2 + d) * 4;
                int I = layer == 0 ? F : 2 * H;
                dirw_t* p = &dw[e][layer * 2 + d];
                p->wih_t = orc_transpose_(q[0], 4 * H, I);
                p->whh_t = orc_transpose_(q[1], 4 * H, H);
                p->b_ih = q[2]; p->b_hh = q[3];
            }
    }
    const float* const* fw = w + 2 * per_enc;
    if (nthreads <= 0) nthreads = 1;
    #pragma omp parallel for num_threads(nthreads) schedule(dynamic, 1)
    for (int64_t n = 0; n < N; ++n) {
        float* cat = (float*)malloc(sizeof(float) * 2 * (size_t)H);
        float* inner = (float*)malloc(sizeof(float) * (size_t)H);
        encoder(dw[0], w[per_enc - 2], w[per_enc - 1], xp + (size_t)n * F * Lp, F, Lp, H, n_layers, cat);
        encoder(dw[1], w[2 * per_enc - 2], w[2 * per_enc - 1], xh + (size_t)n * F * Lh, F, Lh, H, n_layers, cat + H);
        orc_linear(cat, 2 * H, fw[0], fw[1], H, inner);
        for (int j = 0; j < H; ++j) inner[j] = tanhf(inner[j]);
        orc_linear(inner, H, fw[2], fw[3], n_gt, gt_prob + n * n_gt);
        orc_linear(inner, H, fw[4], fw[5], n_zy, zy_prob + n * n_zy);
        orc_softmax(gt_prob + n * n_gt, n_gt);
        orc_softmax(zy_prob + n * n_zy, n_zy);
        free(cat); free(inner);
    }
    for (int e = 0; e < 2; ++e) {
        for (int i = 0; i < n_layers * 2; ++i) { free(dw[e][i].wih_t); free(dw[e][i].whh_t); }
        free(dw[e]);
    }
}
