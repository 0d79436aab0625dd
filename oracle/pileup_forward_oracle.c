/*
 * oracle/pileup_forward_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE (see oracle.h).
 *
 * Restates LSTMNetwork.predict of the PileupModel (PileupModel/model.py:114-119):
 *   BaseEncoder.forward  model.py:31-39   nn.LSTM(18,64,2 layers,bidirectional) + Linear(128,128)
 *   ForwardLayer.forward model.py:66-73   tanh(Linear(128,256)) on all 33 steps, slice [:,16,:],
 *                                         genotype Linear(256,21), zygosity Linear(256,3)
 *   predict              model.py:117-118 softmax over each head; indel heads are dropped.
 * The int32 -> float conversion of the input is PileupModel/predict.py:49.
 * The full reference schedule is computed (all 33 positions through output_proj and dense),
 * so timing this function prices the reference's algorithm, not a reduced one.
 */
#include "oracle.h"
#include "lstm_internal.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define T_ 33
#define C_ 18
#define H_ 64
#define E_ 128
#define J_ 256

void orc_pileup_forward(const float* const* w, const int32_t* x, int64_t N,
                        float* gt_prob, float* zy_prob, int nthreads)
{
    /* transpose once per call */
    float* wt[8]; /* l0f ih,hh; l0r ih,hh; l1f ih,hh; l1r ih,hh */
    const int in_dim[2] = { C_, 2 * H_ };
    for (int l = 0; l < 2; ++l)
        for (int d = 0; d < 2; ++d) {
            const int base = (l * 2 + d) * 4;
            wt[(l * 2 + d) * 2 + 0] = orc_transpose_(w[base + 0], 4 * H_, in_dim[l]);
            wt[(l * 2 + d) * 2 + 1] = orc_transpose_(w[base + 1], 4 * H_, H_);
        }
    const float *proj_w = w[16], *proj_b = w[17], *dense_w = w[18], *dense_b = w[19];
    const float *gt_w = w[20], *gt_b = w[21], *zy_w = w[22], *zy_b = w[23];
    if (nthreads <= 0) nthreads = 1;
    #pragma omp parallel for num_threads(nthreads) schedule(dynamic, 8)
    for (int64_t n = 0; n < N; ++n) {
        float xin[T_ * C_], h0[T_ * 2 * H_], h1[T_ * 2 * H_], enc[T_ * E_], inner[T_ * J_];
        for (int i = 0; i < T_ * C_; ++i) xin[i] = (float)x[n * T_ * C_ + i];
        for (int d = 0; d < 2; ++d)
            orc_lstm_dir_t_(xin, T_, C_, H_, wt[d * 2], wt[d * 2 + 1], w[d * 4 + 2], w[d * 4 + 3],
                            d, T_, h0, 2 * H_, d * H_);
        for (int d = 0; d < 2; ++d)
            orc_lstm_dir_t_(h0, T_, 2 * H_, H_, wt[4 + d * 2], wt[4 + d * 2 + 1],
                            w[8 + d * 4 + 2], w[8 + d * 4 + 3], d, T_, h1, 2 * H_, d * H_);
        for (int t = 0; t < T_; ++t) {
            orc_linear(h1 + t * 2 * H_, 2 * H_, proj_w, proj_b, E_, enc + t * E_);
            orc_linear(enc + t * E_, E_, dense_w, dense_b, J_, inner + t * J_);
            for (int j = 0; j < J_; ++j) inner[t * J_ + j] = tanhf(inner[t * J_ + j]);
        }
        const float* mid = inner + 16 * J_;
        orc_linear(mid, J_, gt_w, gt_b, 21, gt_prob + n * 21);
        orc_linear(mid, J_, zy_w, zy_b, 3, zy_prob + n * 3);
        orc_softmax(gt_prob + n * 21, 21);
        orc_softmax(zy_prob + n * 3, 3);
    }
    for (int i = 0; i < 8; ++i) free(wt[i]);
}
