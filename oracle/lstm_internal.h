/* oracle/lstm_internal.h -- TEST INFRASTRUCTURE (see oracle.h): helpers shared by the
 * forward restatements so weights are transposed once per batch, not once per site. */
#ifndef NANOSNP_ORACLE_LSTM_INTERNAL_H
#define NANOSNP_ORACLE_LSTM_INTERNAL_H
float* orc_transpose_(const float* w, int rows, int cols);
void orc_lstm_dir_t_(const float* x, int T, int I, int H,
                     const float* wih_t, const float* whh_t, const float* b_ih, const float* b_hh,
                     int reverse, int steps, float* out, int out_stride, int out_off);
#endif
