/*
 * oracle/pileup_forward_blocked.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE (see oracle.h).
 *
 * The CPU baseline leg of bench.py ("cpu_baseline", kind "port"): the same function as
 * orc_pileup_forward (pileup_forward_oracle.c -- LSTMNetwork.predict, PileupModel/model.py:114-119,
 * FULL reference schedule: both LSTM layers on all 33 steps, output_proj and dense on all 33
 * positions, heads at position 16), arranged the way a CPU BLAS would run it so that the number
 * printed beside the GPU rate is not a strawman: sites are processed in blocks of 32, every
 * matrix product of a step is a [32 x K] x [K x 256] GEMM whose weight panel (32 output columns
 * x K) stays in L1 across the block, inner loops vectorise to AVX2 FMA (this file alone is
 * compiled with -mavx2 -mfma -ffp-contract=fast; summation order differs from the plain
 * restatement, results agree to ~1e-6 -- tests/test_oracle_golden.py).
 * The plain restatement stays the parity checker; this one is only ever timed.
 */
#include "oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define T_ 33
#define C_ 18
#define H_ 64
#define G_ 256
#define E_ 128
#define J_ 256
#define BS 32          /* sites per block */
#define PC 32          /* output columns per weight panel */

/* panel image of W [rows][K] (row-major, y = W x): P[rows/PC][K][PC] */
static float* pack_panels(const float* w, int rows, int K)
{
    float* p = (float*)aligned_alloc(64, sizeof(float) * (size_t)rows * (size_t)K);
    for (int r0 = 0; r0 < rows; r0 += PC)
        for (int k = 0; k < K; ++k)
            for (int j = 0; j < PC; ++j) p[((size_t)(r0 / PC) * K + k) * PC + j] = w[(size_t)(r0 + j) * K + k];
    return p;
}

/* Y[b][r] = bias[r] (+ bias2[r]) + sum_k X[b][k] P(r,k)   for b < nb; X row stride ldx, Y row stride ldy */
static void gemm_panels(const float* restrict X, int ldx, int nb, int K, const float* restrict P, int rows,
                        const float* restrict bias, const float* restrict bias2, float* restrict Y, int ldy, int accumulate)
{
    for (int r0 = 0; r0 < rows; r0 += PC) {
        const float* restrict pp = P + (size_t)(r0 / PC) * K * PC;
        for (int b = 0; b < nb; ++b) {
            float acc[PC];
            if (accumulate) { for (int j = 0; j < PC; ++j) acc[j] = Y[(size_t)b * ldy + r0 + j]; }
            else { for (int j = 0; j < PC; ++j) acc[j] = bias[r0 + j] + (bias2 ? bias2[r0 + j] : 0.0f); }
            const float* restrict xb = X + (size_t)b * ldx;
            for (int k = 0; k < K; ++k) {
                const float xv = xb[k];
                const float* restrict wr = pp + (size_t)k * PC;
                for (int j = 0; j < PC; ++j) acc[j] += wr[j] * xv;
            }
            for (int j = 0; j < PC; ++j) Y[(size_t)b * ldy + r0 + j] = acc[j];
        }
    }
}

static inline float sigm(float x) { return 1.0f / (1.0f + expf(-x)); }

/* one direction of one BiLSTM layer for a block: in [nb][T][I] -> out [nb][T][2H] at column offset off */
static void lstm_dir_block(const float* in, int I, int nb, const float* Pih, const float* Phh,
                           const float* b_ih, const float* b_hh, int reverse, float* out, int off,
                           float* gates /*[BS][G]*/, float* h /*[BS][H]*/, float* c /*[BS][H]*/)
{
    memset(h, 0, sizeof(float) * BS * H_);
    memset(c, 0, sizeof(float) * BS * H_);
    for (int s = 0; s < T_; ++s) {
        const int t = reverse ? T_ - 1 - s : s;
        gemm_panels(in + (size_t)t * I, T_ * I, nb, I, Pih, G_, b_ih, b_hh, gates, G_, 0);
        if (s > 0) gemm_panels(h, H_, nb, H_, Phh, G_, NULL, NULL, gates, G_, 1);
        for (int b = 0; b < nb; ++b) {
            const float* g = gates + (size_t)b * G_;
            float* hb = h + (size_t)b * H_; float* cb = c + (size_t)b * H_;
            float* o = out + ((size_t)b * T_ + t) * 2 * H_ + off;
            for (int j = 0; j < H_; ++j) {
                const float ig = sigm(g[j]), fg = sigm(g[H_ + j]), gg = tanhf(g[2 * H_ + j]), og = sigm(g[3 * H_ + j]);
                cb[j] = fg * cb[j] + ig * gg;
                hb[j] = og * tanhf(cb[j]);
                o[j] = hb[j];
            }
        }
    }
}

void orc_pileup_forward_blocked(const float* const* w, const int32_t* x, int64_t N,
                                float* gt_prob, float* zy_prob, int nthreads)
{
    float *Pih[4], *Phh[4];
    const int in_dim[2] = { C_, 2 * H_ };
    for (int l = 0; l < 2; ++l)
        for (int d = 0; d < 2; ++d) {
            const int base = (l * 2 + d) * 4;
            Pih[l * 2 + d] = pack_panels(w[base + 0], G_, in_dim[l]);
            Phh[l * 2 + d] = pack_panels(w[base + 1], G_, H_);
        }
    float* Pproj = pack_panels(w[16], E_, 2 * H_);
    float* Pdense = pack_panels(w[18], J_, E_);
    /* the two heads as one [32 x 256] panel set: rows 0..20 genotype, 21..23 zygosity, rest zero */
    float headw[32 * J_], headb[32];
    memset(headw, 0, sizeof headw); memset(headb, 0, sizeof headb);
    memcpy(headw, w[20], sizeof(float) * 21 * J_); memcpy(headw + 21 * J_, w[22], sizeof(float) * 3 * J_);
    memcpy(headb, w[21], sizeof(float) * 21); memcpy(headb + 21, w[23], sizeof(float) * 3);
    float* Phead = pack_panels(headw, 32, J_);
    if (nthreads <= 0) nthreads = 1;
    const int64_t nblk = (N + BS - 1) / BS;
    #pragma omp parallel num_threads(nthreads)
    {
        float* xin = (float*)aligned_alloc(64, sizeof(float) * BS * T_ * C_);
        float* h0 = (float*)aligned_alloc(64, sizeof(float) * BS * T_ * 2 * H_);
        float* h1 = (float*)aligned_alloc(64, sizeof(float) * BS * T_ * 2 * H_);
        float* enc = (float*)aligned_alloc(64, sizeof(float) * BS * T_ * E_);
        float* inner = (float*)aligned_alloc(64, sizeof(float) * BS * T_ * J_);
        float* gates = (float*)aligned_alloc(64, sizeof(float) * BS * G_);
        float* h = (float*)aligned_alloc(64, sizeof(float) * BS * H_);
        float* c = (float*)aligned_alloc(64, sizeof(float) * BS * H_);
        float logits[BS * 32];
        #pragma omp for schedule(dynamic, 1)
        for (int64_t blk = 0; blk < nblk; ++blk) {
            const int64_t n0 = blk * BS;
            const int nb = (int)((N - n0 < BS) ? N - n0 : BS);
            for (int i = 0; i < nb * T_ * C_; ++i) xin[i] = (float)x[n0 * T_ * C_ + i];      /* predict.py:49 */
            for (int d = 0; d < 2; ++d)
                lstm_dir_block(xin, C_, nb, Pih[d], Phh[d], w[d * 4 + 2], w[d * 4 + 3], d, h0, d * H_, gates, h, c);
            for (int d = 0; d < 2; ++d)
                lstm_dir_block(h0, 2 * H_, nb, Pih[2 + d], Phh[2 + d], w[8 + d * 4 + 2], w[8 + d * 4 + 3], d, h1, d * H_, gates, h, c);
            /* output_proj and tanh(dense) on all 33 positions (model.py:37,67), rows = (site, t) */
            gemm_panels(h1, 2 * H_, nb * T_, 2 * H_, Pproj, E_, w[17], NULL, enc, E_, 0);
            gemm_panels(enc, E_, nb * T_, E_, Pdense, J_, w[19], NULL, inner, J_, 0);
            for (int i = 0; i < nb * T_ * J_; ++i) inner[i] = tanhf(inner[i]);
            gemm_panels(inner + 16 * J_, T_ * J_, nb, J_, Phead, 32, headb, NULL, logits, 32, 0);    /* [:,16,:] (model.py:68) */
            for (int b = 0; b < nb; ++b) {
                memcpy(gt_prob + (n0 + b) * 21, logits + b * 32, sizeof(float) * 21);
                memcpy(zy_prob + (n0 + b) * 3, logits + b * 32 + 21, sizeof(float) * 3);
                orc_softmax(gt_prob + (n0 + b) * 21, 21);
                orc_softmax(zy_prob + (n0 + b) * 3, 3);
            }
        }
        free(xin); free(h0); free(h1); free(enc); free(inner); free(gates); free(h); free(c);
    }
    for (int i = 0; i < 4; ++i) { free(Pih[i]); free(Phh[i]); }
    free(Pproj); free(Pdense); free(Phead);
}
