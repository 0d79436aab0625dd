/*
 * oracle/lstm_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE (see oracle.h).
 *
 * fp32 restatement of what torch.nn.LSTM / nn.Linear / softmax compute for the reference's
 * eval-mode networks.  The arithmetic of the reference lives in PyTorch (unpinned version,
 * Dockerfile:23-24); this file restates the published torch.nn.LSTM equations
 *     i = sigmoid(W_ii x + b_ii + W_hi h + b_hi)      f = sigmoid(W_if x + ...)
 *     g = tanh   (W_ig x + b_ig + W_hg h + b_hg)      o = sigmoid(W_io x + ...)
 *     c' = f*c + i*g                                  h' = o*tanh(c')
 * with the weight rows stored gate-major (i,f,g,o), h0 = c0 = 0, and the call sites
 * PileupModel/model.py:18-37 and HaplotypeModel/model_dev.py:63-81 (batch_first,
 * bidirectional, dropout inactive in eval).  Pinned by golden vectors generated from the
 * reference modules themselves (tests/golden/make_golden.py).
 */
#include "oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

static inline float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

/* transposed copy so that the inner loops run over the gate index (vectorisable without
 * re-associating the k-sum): wt[k][r] = w[r][k] */
float* orc_transpose_(const float* w, int rows, int cols)
{
    float* wt = (float*)malloc(sizeof(float) * (size_t)rows * (size_t)cols);
    for (int r = 0; r < rows; ++r)
        for (int k = 0; k < cols; ++k) wt[(size_t)k * rows + r] = w[(size_t)r * cols + k];
    return wt;
}

/* one direction of one layer with pre-transposed weights wih_t [I][4H], whh_t [H][4H];
 * out has row stride out_stride and the direction's H values go to column offset out_off.
 * steps: how many time steps to run (T for a full pass). */
void orc_lstm_dir_t_(const float* x, int T, int I, int H,
                     const float* wih_t, const float* whh_t, const float* b_ih, const float* b_hh,
                     int reverse, int steps, float* out, int out_stride, int out_off)
{
    const int G = 4 * H;
    float* gates = (float*)malloc(sizeof(float) * (size_t)G);
    float* h = (float*)calloc((size_t)H, sizeof(float));
    float* c = (float*)calloc((size_t)H, sizeof(float));
    for (int s = 0; s < steps; ++s) {
        int t = reverse ? (T - 1 - s) : s;
        const float* xt = x + (size_t)t * I;
        for (int r = 0; r < G; ++r) gates[r] = b_ih[r] + b_hh[r];
        for (int k = 0; k < I; ++k) {
            const float xv = xt[k]; const float* w = wih_t + (size_t)k * G;
            for (int r = 0; r < G; ++r) gates[r] += w[r] * xv;
        }
        for (int k = 0; k < H; ++k) {
            const float hv = h[k]; const float* w = whh_t + (size_t)k * G;
            for (int r = 0; r < G; ++r) gates[r] += w[r] * hv;
        }
        for (int j = 0; j < H; ++j) {
            float ig = sigmoidf_(gates[j]);
            float fg = sigmoidf_(gates[H + j]);
            float gg = tanhf(gates[2 * H + j]);
            float og = sigmoidf_(gates[3 * H + j]);
            c[j] = fg * c[j] + ig * gg;
            h[j] = og * tanhf(c[j]);
            out[(size_t)t * out_stride + out_off + j] = h[j];
        }
    }
    free(gates); free(h); free(c);
}

void orc_lstm_bidir_layer(const float* x, int T, int I, int H,
                          const float* w_ih_f, const float* w_hh_f,
                          const float* b_ih_f, const float* b_hh_f,
                          const float* w_ih_r, const float* w_hh_r,
                          const float* b_ih_r, const float* b_hh_r,
                          float* out)
{
    float* a = orc_transpose_(w_ih_f, 4 * H, I); float* b = orc_transpose_(w_hh_f, 4 * H, H);
    orc_lstm_dir_t_(x, T, I, H, a, b, b_ih_f, b_hh_f, 0, T, out, 2 * H, 0);
    free(a); free(b);
    a = orc_transpose_(w_ih_r, 4 * H, I); b = orc_transpose_(w_hh_r, 4 * H, H);
    orc_lstm_dir_t_(x, T, I, H, a, b, b_ih_r, b_hh_r, 1, T, out, 2 * H, H);
    free(a); free(b);
}

void orc_linear(const float* x, int n_in, const float* w, const float* b, int n_out, float* y)
{
    for (int r = 0; r < n_out; ++r) {
        float acc = b[r];
        const float* wr = w + (size_t)r * n_in;
        for (int k = 0; k < n_in; ++k) acc += wr[k] * x[k];
        y[r] = acc;
    }
}

void orc_softmax(float* v, int n)
{
    float m = v[0];
    for (int i = 1; i < n; ++i) if (v[i] > m) m = v[i];
    float s = 0.f;
    for (int i = 0; i < n; ++i) { v[i] = expf(v[i] - m); s += v[i]; }
    for (int i = 0; i < n; ++i) v[i] /= s;
}
