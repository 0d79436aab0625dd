/*
 * oracle/oracle.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C CPU restatement of the NanoSNP candidate-site inference hot path.  Every function
 * cites the reference file:line whose behaviour it restates (paths relative to the upstream
 * repository huangnengCSU/NanoSNP).  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load this library; the product path (nanosnp_amd/) never
 * does and fails loudly when its HIP extension is missing.
 *
 * Pinning status (see DESIGN.md "Oracle"):
 *   - pileup encode (E0-E3): pinned bit-for-bit against the reference's own C++ programs
 *     compiled from the reference tree (oracle/_ref), on whole synthetic chromosomes
 *     (tests/test_oracle_vs_ref.py, tests/golden/encode_*.pd.gz).
 *   - PileupModel forward (P2-P4): pinned against golden outputs produced by importing the
 *     reference's PileupModel/model.py with the shipped ont_pileup.chkpt on CPU torch
 *     (tests/golden/make_golden.py -> tests/golden/pileup_fwd_*.npz).  The reference ships no test
 *     of its own for this boundary.
 *   - haplotype features (H4/H5): pinned against goldens from the reference's
 *     dataset_dev.get_frequency_feature run in the development container.
 *   - HaplotypeModel forward (H6): trained weights are absent from the reference tree;
 *     pinned against goldens from the reference module with seeded random weights.
 *   - legacy CatModel forward (C1/C2): the same (no trained weights in the tree): goldens from
 *     HaplotypeModel/model.py CatModel with seeded weights and BatchNorm statistics.
 */
#ifndef NANOSNP_ORACLE_H
#define NANOSNP_ORACLE_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- channel order: dna_sv_tensor/src/common/tensor.hpp:6-26 ------------------------- */
enum {
    ORC_CH_A = 0, ORC_CH_C, ORC_CH_G, ORC_CH_T, ORC_CH_I, ORC_CH_I1, ORC_CH_D, ORC_CH_D1,
    ORC_CH_STAR, ORC_CH_a, ORC_CH_c, ORC_CH_g, ORC_CH_t, ORC_CH_i, ORC_CH_i1, ORC_CH_d,
    ORC_CH_d1, ORC_CH_POUND, ORC_NCH
};

/* flag bits written per column by orc_encode_columns */
#define ORC_FLAG_PASS_AF    1u  /* pass_af as returned by make_tensor (tensor_maker.cpp:248) */
#define ORC_FLAG_PASS_SNP   2u  /* pass_snp_af   */
#define ORC_FLAG_PASS_INDEL 4u  /* pass_indel_af */
#define ORC_FLAG_CANDIDATE  8u  /* main.cpp:195: ref in ACGT && pass_af && depth >= min_cov */

/* E1: one pileup column (tensor_maker.cpp:61-249).
 * bases/len: column 5 of the mpileup line.  ref_raw: chr_seq[pos-1] as stored in the FASTA.
 * next_ref: the reference bases following the position (chr_seq[pos], ...), n_next of them
 * (only used for deletion alt keys; may be NULL when alt_info is NULL).
 * alt_info (optional, capacity alt_cap): receives the "KEY cnt KEY cnt " text of alt_dict in
 * std::map order (main.cpp:227-231 without the leading "depth-").  Returns the number of
 * bytes the full text needs (excluding the NUL). */
typedef struct {
    int32_t counts[ORC_NCH];
    int32_t depth;
    int32_t max_del_length;
    double  af;
    uint8_t pass_af, pass_snp_af, pass_indel_af;
} orc_column_t;

size_t orc_make_tensor(const char* bases, int64_t len, char ref_raw,
                       const char* next_ref, int64_t n_next,
                       double snp_min_af, double indel_min_af,
                       orc_column_t* out, char* alt_info, size_t alt_cap);

/* Batch form used as the checker for the HIP encode kernel.  bases: concatenated column
 * strings, col_off[M+1] byte offsets, ref[M] raw reference byte per column.
 * counts[M*18], depth[M], flags[M] (ORC_FLAG_*).  min_af is applied to both the SNP and
 * the indel test, as the pipeline does (make_predict_data.sh:184-194). */
void orc_encode_columns2(const uint8_t* bases, const int64_t* col_off, const uint8_t* ref,
                         int64_t M, double snp_min_af, double indel_min_af, int min_coverage,
                         int32_t* counts, int32_t* depth, uint8_t* flags);
int64_t orc_mpileup_to_pd2(const char* mpileup_path, const char* chr_seq, int64_t chr_len,
                           double snp_min_af, double indel_min_af, int min_coverage, int flank, const char* pd_path);
void orc_encode_columns(const uint8_t* bases, const int64_t* col_off, const uint8_t* ref,
                        int64_t M, double min_af, int min_coverage,
                        int32_t* counts, int32_t* depth, uint8_t* flags);

/* E2 window rule (make_candidate_snp_tensor/main.cpp:174-217): column c is emitted as a site
 * iff it is a candidate and the 33 columns c-16..c+16 are consecutive positions of one
 * uninterrupted run.  pos[M] strictly increasing within a contig (callers fold the contig
 * id into the high bits).  Writes the column indices of emitted sites to center_idx
 * (capacity cap) and returns how many there are. */
int64_t orc_select_sites(const int64_t* pos, const uint8_t* flags, int64_t M, int flank,
                         int64_t* center_idx, int64_t cap);

/* window gather: x[n][t][ch] = counts[center_idx[n]-flank+t][ch]  (main.cpp:233-244) */
void orc_gather_windows(const int32_t* counts, const int64_t* center_idx, int64_t N,
                        int flank, int32_t* x);

/* E2+E3 at file level: <chr>.mpileup text + chromosome sequence -> .pd text, byte-identical
 * to DNA_CreateCanSnpTensor | DNA_CreatePredictData (main.cpp:113-312,
 * make_predict_data/main.cpp:76-127).  Returns the number of sites written, <0 on I/O error. */
int64_t orc_mpileup_to_pd(const char* mpileup_path, const char* chr_seq, int64_t chr_len,
                          double min_af, int min_coverage, int flank, const char* pd_path);

/* mpileup text in memory -> per line the position (atoll of token 1) and the byte range of token 4 (the pileup bases), read the
 * way the reference reads the file: line_reader.cpp:95-127, cpp_aux.cpp:43-59, make_candidate_snp_tensor/main.cpp:162-172.
 * Returns the number of lines, or -(k + 1) when line k has fewer than five tokens (the reference cannot survive such a line). */
int64_t orc_mpileup_tokenise(const char* text, int64_t len, int64_t cap, int64_t* pos, int64_t* beg, int64_t* end);

/* ---- LSTM building block (torch.nn.LSTM semantics, gate order i,f,g,o) --------------- */
/* One bidirectional layer over a [T][I] sequence -> out [T][2H] (fwd in [:H], rev in [H:]).
 * Restates what nn.LSTM(batch_first, bidirectional, h0=c0=0) computes in eval mode:
 * PileupModel/model.py:18-37, HaplotypeModel/model_dev.py:63-81. */
void orc_lstm_bidir_layer(const float* x, int T, int I, int H,
                          const float* w_ih_f, const float* w_hh_f,
                          const float* b_ih_f, const float* b_hh_f,
                          const float* w_ih_r, const float* w_hh_r,
                          const float* b_ih_r, const float* b_hh_r,
                          float* out);
void orc_linear(const float* x, int n_in, const float* w, const float* b, int n_out, float* y);
void orc_softmax(float* v, int n);

/* ---- PileupModel forward (P2-P4): PileupModel/model.py:31-39,66-73,114-119 ----------- */
/* weights: the 24 tensors of ont_pileup.chkpt in state-dict order:
 *  [0..15] encoder.lstm l0 fwd(w_ih,w_hh,b_ih,b_hh), l0 rev, l1 fwd, l1 rev
 *  [16,17] encoder.output_proj weight,bias   [18,19] forward_layer.dense weight,bias
 *  [20,21] genotype_layer  [22,23] zygosity_layer  (indel heads are not used by predict) */
void orc_pileup_forward(const float* const* w, const int32_t* x /*[N,33,18]*/, int64_t N,
                        float* gt_prob /*[N,21]*/, float* zy_prob /*[N,3]*/, int nthreads);
/* the same function blocked over 32 sites with L1-resident weight panels and AVX2 FMA inner loops
 * (pileup_forward_blocked.c): only the CPU baseline leg of bench.py times it */
void orc_pileup_forward_blocked(const float* const* w, const int32_t* x /*[N,33,18]*/, int64_t N,
                                float* gt_prob /*[N,21]*/, float* zy_prob /*[N,3]*/, int nthreads);

/* ---- haplotype features (H4/H5): HaplotypeModel/dataset_dev.py:11-87,337-349 --------- */
/* four int32 planes [D][L]; ref_row[L] int32; out double [105][L] (row 104 = ref row) */
void orc_hap_features(const int32_t* seq, const int32_t* bq, const int32_t* mq,
                      const int32_t* hap, const int32_t* ref_row, int D, int L, double* out);
void orc_hap_features_batch(const int32_t* seq, const int32_t* bq, const int32_t* mq,
                            const int32_t* hap, const int32_t* ref_row, int64_t N, int D, int L,
                            float* out /*[N,105,L] cast as predict_dev.py:35 does*/, int nthreads);

/* H1/H2: keep reads covering the centre column, order by HP at the centre (stable), pad -2, cut at
 * D_out (create_pileup_haplotype.py:140-207, write_to_bins.py:15-61).  One site: inputs [rows][L]. */
void orc_hap_arrange(const int32_t* seq, const int32_t* bq, const int32_t* mq, const int32_t* hap,
                     int rows, int R, int L, int D_out,
                     int32_t* oseq, int32_t* obq, int32_t* omq, int32_t* ohap, int32_t* depth);

/* ---- HaplotypeModel forward (H6): HaplotypeModel/model_dev.py:59-143 ------------------ */
/* weights: state-dict order of model_dev.LSTMNetwork:
 *  pileup_encoder: 3 layers x 2 dirs x (w_ih,w_hh,b_ih,b_hh) = 24, output_proj w,b = 26
 *  haplotype_encoder: the same 26;  forward_layer: dense w,b, genotype w,b, zygosity w,b = 6
 *  total 58 tensors. */
void orc_hap_forward(const float* const* w, const float* xp /*[N,105,33]*/,
                     const float* xh /*[N,105,11]*/, int64_t N, int F, int H, int n_layers,
                     int Lp, int Lh, int n_gt, int n_zy,
                     float* gt_prob, float* zy_prob, int nthreads);

/* ---- legacy CatModel forward (C1/C2): HaplotypeModel/model.py:332-358, crnn.py:84-190 -- */
/* weights: the 132 floating-point tensors of CatModel.state_dict() in order (see cat_forward_oracle.c);
 * g0, g1: [N,40,11,5] float32 (predict.py:33-34); gt_prob [N,10]. */
void orc_cat_forward(const float* const* w, const float* g0, const float* g1, int64_t N,
                     float* gt_prob, int nthreads);

void orc_cat_groups(const int32_t* r1, const int32_t* q1, const int32_t* m1, int D1,
                    const int32_t* r2, const int32_t* q2, const int32_t* m2, int D2,
                    int64_t N, int L, float* g /*[N,40,L,5]*/);

/* ---- P5/H7: QUAL score (PileupModel/predict.py:31-34) --------------------------------- */
double orc_calculate_score(double p);

#ifdef __cplusplus
}
#endif
#endif
