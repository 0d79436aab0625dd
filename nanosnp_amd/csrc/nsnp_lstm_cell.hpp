// nsnp_lstm_cell.hpp -- the ONE LSTM cell of every recurrence kernel (pileup_forward.hip, pileup_forward_bf16x3.hip, hap_gemm.hpp)
// and the gate-scale contract its weight packers follow.
//
// torch.nn.LSTM equations (PileupModel/model.py:34, HaplotypeModel/model_dev.py:33), gate order i f g o:
//   c' = sigmoid(f) c + sigmoid(i) tanh(g),   h' = sigmoid(o) tanh(c')
// with sigmoid(x) = 1 / (1 + e^-x), tanh(x) = (1 - e^-2x) / (1 + e^-2x): the two products share ONE reciprocal each,
//   sigmoid(i) tanh(g) = (1 - e^-2g) / ((1 + e^-i)(1 + e^-2g)),
// 5 v_exp_f32 + 3 v_rcp_f32 per unit and step instead of 5 + 5 (fp32 MFMAs and vector instructions share a SIMD's lanes: every
// instruction of the cell is matrix-pipe time).  The exponent of the tanh terms is capped at 2^64 so that 1 - e stays finite; an
// overflowing product of the denominators gives reciprocal 0, the correct limit.
//
// Contract with the packers: the four pre-activations arrive already multiplied by lstm_gate_scale(gate) - zi, zf, zo by -log2 e,
// zg by -2 log2 e (folded into the gate rows of W_ih, W_hh and the biases at pack time) - and the cell state is kept multiplied by
// -2 log2 e (it is only ever the argument of the next tanh): K c' = f (K c) + (K - K e_g) r.
#pragma once

namespace nsnp_cell {

constexpr float LOG2E = 1.4426950408889634f;

// factor folded into image row R of an LSTM weight image and its bias (image row R: unit R / 4, gate R % 4 in the order i f g o)
static inline float lstm_gate_scale(int R) { return (R & 3) == 2 ? -2.0f * LOG2E : -LOG2E; }

// min(z, 64) as ONE v_med3_f32 (fminf costs a canonicalising v_max_f32 beside its v_min_f32 where the operand is a computed value);
// the same bits for every z: below -3e38 the exponential is 0 either way
__device__ __forceinline__ float cap64(float z) { return __builtin_amdgcn_fmed3f(z, -3.0e38f, 64.0f); }

__device__ __forceinline__ float lstm_cell(float zi, float zf, float zg, float zo, float c_prev, float& c_new)
{
    const float ei = __builtin_amdgcn_exp2f(zi);
    const float ef = __builtin_amdgcn_exp2f(zf);
    const float eo = __builtin_amdgcn_exp2f(zo);
    const float eg = __builtin_amdgcn_exp2f(cap64(zg));
    constexpr float K = -2.0f * LOG2E;
    const float tg = 1.0f + eg;                                    // (1 + e_i)(1 + e_g) = e_i t + t
    const float ig = __builtin_fmaf(-K, eg, K) * __builtin_amdgcn_rcpf(__builtin_fmaf(ei, tg, tg));
    const float fg = __builtin_amdgcn_rcpf(1.0f + ef);
    const float cn = __builtin_fmaf(fg, c_prev, ig);
    const float ec = __builtin_amdgcn_exp2f(cap64(cn));
    c_new = cn;
    const float tc = 1.0f + ec;
    return (1.0f - ec) * __builtin_amdgcn_rcpf(__builtin_fmaf(eo, tc, tc));
}

}  // namespace nsnp_cell
