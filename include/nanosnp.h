/*
 * nanosnp.h -- C ABI of libnanosnp_hip.so: the MI355X (gfx950) implementation of NanoSNP's
 * candidate-site inference hot path.  Plain pointers and sizes only (no torch / HIP types in
 * the signatures; `stream` is a hipStream_t passed as void*, NULL = the default stream).
 *
 * The reference (huangnengCSU/NanoSNP) has no FFI of its own; each entry point replaces the
 * narrowest existing call site of the hot path (paths relative to the upstream repository):
 *
 *   nsnp_pileup_forward          LSTMNetwork.predict            PileupModel/model.py:114-119
 *                                called from                    PileupModel/predict.py:49-51
 *   nsnp_pileup_encode_columns   TensorMaker::make_tensor       dna_sv_tensor/src/make_candidate_snp_tensor/tensor_maker.cpp:61-249
 *                                + candidate test               dna_sv_tensor/src/make_candidate_snp_tensor/main.cpp:194-201
 *   nsnp_pileup_select_sites     window / pending-queue rule    dna_sv_tensor/src/make_candidate_snp_tensor/main.cpp:174-217
 *   nsnp_mpileup_tokenise        LineReader + split_line + atoll dna_sv_tensor/src/make_candidate_snp_tensor/main.cpp:162-172
 *   nsnp_pileup_gather_windows   33-column window emission      dna_sv_tensor/src/make_candidate_snp_tensor/main.cpp:233-244
 *   nsnp_pileup_postprocess      argmax / max / depth           PileupModel/predict.py:52-65
 *   nsnp_hap_features            get_frequency_feature + ref row HaplotypeModel/dataset_dev.py:55-87,337-349
 *   nsnp_hap_forward             LSTMNetwork.predict            HaplotypeModel/model_dev.py:133-143
 *   nsnp_cat_forward             legacy CatModel.predict        HaplotypeModel/model.py:332-358 (ResCRNN: crnn.py:84-190)
 *                                called from                    HaplotypeModel/predict.py:53
 *   nsnp_cat_groups              PredictDataset.__getitem__     HaplotypeModel/dataset.py:862-915 (g0 / g1 assembly)
 *                                called from                    HaplotypeModel/predict_dev.py:35-39
 *
 * Result gather: the path's only exchange is one rooted gather of a few bytes per site at the very end.
 * PyTorch ranks do it with one torch.distributed collective (nanosnp_amd/dist.py gather_results / gather_varlen: PyTorch owns their
 * RCCL communicator and does not hand it to foreign code); callers that are not PyTorch processes use nsnp_comm_* +
 * nsnp_gather_results below, which bind a communicator of their own to the context (RCCL resolved at run time, no link dependency).
 *
 * Conventions: every pointer marked "device" is device memory owned by the caller; functions
 * are asynchronous on `stream` unless stated, return 0 on success or a negative NSNP_E* code
 * (never abort/throw; the reference's native stage abort()s, cpp_aux.cpp:10-21), and a
 * context is bound to one device and must be used by one host thread at a time.  Use one
 * context per stream when several batches are in flight.
 */
#ifndef NANOSNP_H
#define NANOSNP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NSNP_OK          0
#define NSNP_EINVAL     (-1)   /* bad argument                              */
#define NSNP_ENOMEM     (-2)   /* host or device allocation failed          */
#define NSNP_EHIP       (-3)   /* a HIP runtime call failed (see nsnp_last_hip_error) */
#define NSNP_ENOWEIGHTS (-4)   /* forward called before load_weights        */
#define NSNP_EARCH      (-5)   /* device is not gfx950                      */
#define NSNP_ESHAPE     (-6)   /* unsupported model dimensions              */
#define NSNP_ENOTSUP    (-7)   /* optional component unavailable (RCCL not found in the process or on the loader path) */
#define NSNP_ECOMM      (-8)   /* an RCCL call failed                        */

#define NSNP_PILEUP_WINDOW   33   /* PileupModel/dataset.py:11-12, 2*flanking_base+1 */
#define NSNP_PILEUP_CHANNELS 18   /* dna_sv_tensor/src/common/tensor.hpp:6-26        */
#define NSNP_GT_CLASSES      21   /* PileupModel/config/ont_pileup.yaml:17           */
#define NSNP_ZY_CLASSES       3
#define NSNP_HAP_FEATURES   105   /* HaplotypeModel/config/ont_haplotype.yaml:8-9    */

/* per-column flag bits written by nsnp_pileup_encode_columns */
#define NSNP_FLAG_PASS_AF     1u  /* pass_af as make_tensor returns it (tensor_maker.cpp:248) */
#define NSNP_FLAG_PASS_SNP    2u
#define NSNP_FLAG_PASS_INDEL  4u
#define NSNP_FLAG_CANDIDATE   8u  /* ref in ACGT && pass_af && depth >= min_coverage (main.cpp:195) */

typedef struct nsnp_ctx nsnp_ctx;

int         nsnp_version(void);
const char* nsnp_strerror(int code);
/* last hipError_t seen by this context (0 = hipSuccess) and its text; ctx == NULL reports the
 * error of the last failed nsnp_ctx_create */
int         nsnp_last_hip_error(const nsnp_ctx* ctx, const char** text);

/* Creates a context on `device` (must be gfx950).  Allocates no workspace yet. */
int nsnp_ctx_create(int device, nsnp_ctx** ctx);
int nsnp_ctx_destroy(nsnp_ctx* ctx);
/* Workspace is sized for `max_sites` sites per internal chunk (default 32768); larger calls are
 * processed in chunks.  Synchronous; call before the first forward to keep allocation out of
 * the hot loop (and out of hipGraph capture). */
int nsnp_ctx_reserve(nsnp_ctx* ctx, int64_t max_sites);

/* Options (name, value).  Unknown names / values return NSNP_EINVAL.  Precision options choose the arithmetic of the matrix
 * products (sums are fp32 in every mode):
 *   0  exact fp32 MFMA (the default: the reference computes in fp32)
 *   2  "bf16x3": every fp32 operand as three bf16 terms (8 + 8 + 8 significand bits = the full fp32 significand, fp32 exponent
 *      range: v = p0 + p1 + p2 exactly), six bf16 MFMAs per product, the three dropped cross terms below 2^-24 of the product -
 *      nothing is narrower than fp32.  Measured against a float64 evaluation of the PileupModel its error equals the fp32 path's
 *      (1-2e-6 in the probabilities); every reference golden holds the 1e-4 contract with no relaxed bound; no input range limit.
 *   1  "f16x3" (opt-in): every operand as two fp16 halves, three fp16 MFMAs per product: 21-22 significand bits per operand.
 *      DOCUMENTED BOUND: within 1e-4 of the reference on inputs whose magnitudes stay below 2048 (every pileup count, every
 *      generator-G3 feature); up to 2e-4 (measured 1.5e-4) on HaplotypeModel sites whose count-valued features reach several
 *      thousand (the saturated sites of tests/golden/hap_fwd_large.npz); caller-supplied inputs beyond +-131008 saturate.
 *      Use mode 2 where the 1e-4 contract must hold on arbitrary inputs.
 * The other options only change launch shapes.
 *   "pileup_precision"        0 fp32 (default) | 2 bf16x3 | 1 f16x3      PileupModel forward
 *   "hap_precision"           0 fp32 (default) | 2 bf16x3 | 1 f16x3      HaplotypeModel forward
 *   "cat_precision"           0 fp32 (default) | 2 bf16x3 | 1 f16x3      legacy CatModel forward
 *   "cat_conv_pix2"           1 (default) | 0                  k_cat_conv: the blocks with <= 64 output channels on 256-pixel workgroups (16 / 32 MFMAs
 *                                                              per wave between two barriers instead of 8 / 16); 0 = 128 pixels everywhere.  Same bits.
 *   "cat_conv_lds"            1 (default) | 0                  legacy CatModel 3x3 convolutions: the pixel block + halo of a channel chunk staged in
 *                                                              LDS once and read by all nine taps, or the round-3 GEMM that gathers every tap from the
 *                                                              image; the two sum the same products in a different order
 *   "hap_b3x"                 1 (default) | 0                  bf16x3 HaplotypeModel / CatModel LSTM steps: 256 x 256 workgroup tiles where the launch fills
 *                                                              the chip with them, or always the 128 x 128 tiles of the other modes (bit-identical)
 *   "hap_pass_sites"          128..131072, multiple of 128     sites per internal pass of the HaplotypeModel forward (default 16384;
 *                                                              workspace ~195 KB per site = 3.2 GB per context at the default,
 *                                                              (re)allocated synchronously by this call and by nsnp_hap_load_weights,
 *                                                              never by nsnp_hap_forward; callers with small batches or many contexts
 *                                                              set a smaller pass BEFORE loading the weights.  When the new size cannot
 *                                                              be allocated the call returns NSNP_ENOMEM and the previous pass size and
 *                                                              workspace stay in force)
 *   "recurrence_waves"        0 auto | 1/2/4/8                 waves per workgroup of the LDS-image recurrence kernels
 *   "l0_register_stationary"  1 (default) | 0                 f16x3 layer 0: weights in VGPRs + LDS exchange of h, or LDS images
 *   "l0_site_groups"          0 auto | 1/2/4                   16-site groups per workgroup of that kernel
 *   "l1_register_stationary"  1 (default) | 2 | 0             f16x3 fused layer 1: weights in VGPRs + LDS operands as four waves x four gate
 *                                                              tiles and 16 sites per workgroup (1), as eight waves x two tiles (2), or LDS images + ring (0)
 *   "l1_site_groups"          0 auto | 2/4                     16-site groups per workgroup of the eight-wave kernel (a non-zero value selects it)
 *   "fused_l1"                1 (default) | 0                 f16x3 layer 1: projection fused into the recurrence
 *   "fused_waves"             0 auto | 4/8/12                  waves per workgroup of the fused kernel
 *   "proj1_tiles"             1..64                            row tiles per wave of the unfused projection kernel
 *   fp32 path (pileup_precision 0): "l0_register_stationary" 1 (default) | 0, "l1_register_stationary" 1 four waves x four gate
 *   tiles (default) | 2 eight waves x two tiles | 0 LDS-image kernels with the Xp1 round trip, "l1_site_groups" 0 | 1 | 2 | 4,
 *   "l1_stagger" 0 (default) | 1 (eight-wave kernel: waves 4-7 issue a group's next input part ahead of its cell),
 *   "head_split" 1 (default) | 0 heads with the output tiles split over eight waves, "static_priority" 0..3 (f16x3
 *   register-stationary kernels: s_setprio 1 for waves 4-7 of layer 1 / odd workgroups of layer 0; measured without effect),
 *   "l0_input_weights_in_lds" 0 (default) | 1 (fp32 layer 0, 16-site workgroups: input-part weight fragments in LDS, 128
 *   VGPRs; measured without gain).  Combinations without effect are accepted and ignored: "l1_site_groups" under
 *   "l1_register_stationary" 1 (its workgroups are always one 16-site group), "l1_stagger" outside the eight-wave kernel,
 *   "static_priority" and "fused_*" / "proj1_tiles" on the fp32 path.
 *   bf16x3 path (pileup_precision 2): "l0_site_groups" 0 auto | 1/2/4 and "l1_site_groups" 0 auto | 1/2/4 (16-site groups per
 *   workgroup of its layer-0 / layer-1 kernel); every combination returns the same bits.
 *   Every fp32 combination returns bit-identical probabilities. */
int nsnp_ctx_set_option(nsnp_ctx* ctx, const char* name, int64_t value);

/* Optional per-kernel timing: when enabled every launch of the kernels below is bracketed by a
 * HIP event pair recorded on the launch stream (up to 8192 launches per kernel between reads).
 * nsnp_ctx_read_timing waits for the recorded events, returns their summed duration and count
 * for kernel id (0 layer-0 recurrence, 1 layer-1 projection, 2 layer-1 recurrence, 3 heads,
 * 4 column encode, 5 haplotype features) and resets the counter.  Ids 6-8 bracket a whole CHAIN of launches of one
 * internal pass with one event pair each: 6 the fused LSTM step launches of a HaplotypeModel pass (83 launches of
 * k_hap_gemm), 7 the convolution GEMM + pooling launches of a CatModel pass (12 + 4), 8 a whole CatModel pass;
 * `launches` then counts passes.  Synchronous. */
int nsnp_ctx_enable_timing(nsnp_ctx* ctx, int enable);
int nsnp_ctx_read_timing(nsnp_ctx* ctx, int kernel, double* total_ms, int64_t* launches);

/* Diagnostic: the shader clock (MHz) the device holds under a full-chip fp32 MFMA load of about 2 ms on `stream` (s_memtime
 * against the 100 MHz s_memrealtime in every workgroup of a probe kernel).  Boxes differ by 10 % and every MFMA fraction of a
 * bench line is priced at the 2.4 GHz peak, so the lines record it.  Synchronous; no product path depends on it. */
int nsnp_ctx_shader_clock(nsnp_ctx* ctx, double* mhz, void* stream);

/* ---- PileupModel ---------------------------------------------------------------------- */
/* host_tensors: the 24 fp32 tensors LSTMNetwork.predict uses, HOST pointers, in the
 * state-dict order of ont_pileup.chkpt (PileupModel/predict.py:212-214):
 *   encoder.lstm: l0 {w_ih[256,18], w_hh[256,64], b_ih[256], b_hh[256]}, l0_reverse {..},
 *                 l1 {w_ih[256,128], w_hh[256,64], b_ih, b_hh}, l1_reverse {..}      (16)
 *   encoder.output_proj {weight[128,128], bias[128]}                                 (2)
 *   forward_layer.dense {weight[256,128], bias[256]}                                 (2)
 *   forward_layer.genotype_layer {weight[21,256], bias[21]}, zygosity_layer {[3,256],[3]} (4)
 * Synchronous (packs and uploads the MFMA weight images). */
int nsnp_pileup_load_weights(nsnp_ctx* ctx, const float* const* host_tensors, int n_tensors);

/* x: device int32 [N,33,18] (the position_matrix of make_bin_predict_data.py:90-100; the
 * int->float conversion of predict.py:49 happens inside).  Outputs: device fp32 softmax
 * probabilities gt_prob [N,21], zy_prob [N,3] (model.py:117-118). */
int nsnp_pileup_forward(nsnp_ctx* ctx, const int32_t* x, int64_t N,
                        float* gt_prob, float* zy_prob, void* stream);

/* Same forward reading the windows straight out of the per-column count matrix:
 * site n uses counts[center_idx[n]-16 .. center_idx[n]+16][18] (no gathered copy). */
int nsnp_pileup_forward_windows(nsnp_ctx* ctx, const int32_t* counts, const int64_t* center_idx,
                                int64_t N, float* gt_prob, float* zy_prob, void* stream);

/* nsnp_pileup_forward_windows and predict.py:54-57 (np.argmax / np.max of both heads) in one call: the fp32 heads kernel writes
 * gt_arg / zy_arg (uint8) and gt_max / zy_max (fp32) from the registers that hold the probabilities, one launch less per batch than
 * forward + nsnp_pileup_postprocess; every other arithmetic mode / kernel generation runs the two back to back on `stream`.
 * Same values as the two-call sequence, bit for bit (first maximum wins, as np.argmax).  gt_arg, zy_arg, gt_max, zy_max may be device
 * pointers or pointers into PINNED HOST memory (hipHostMalloc / hipHostRegister: mapped on the device): the kernel then writes the 10
 * bytes per site straight into host memory and a streamed caller needs no D2H copy - the values are valid on the host once an event
 * recorded on `stream` behind the call has completed (nanosnp_amd/pipeline.py predict_pileup_bins). */
int nsnp_pileup_forward_windows_calls(nsnp_ctx* ctx, const int32_t* counts, const int64_t* center_idx, int64_t N,
                                      float* gt_prob, float* zy_prob, uint8_t* gt_arg, uint8_t* zy_arg,
                                      float* gt_max, float* zy_max, void* stream);

/* The call rows a streamed text run keeps on the device - rows: device float64 [N,13] = position, gt argmax, zy argmax, gt max, zy max,
 * the eight coverage channels x[n,16,{0,1,2,3,9,10,11,12}] of predict.py:63 (all exact in float64) - cut into the typed arrays the row
 * formatter (nsnp_vcf_format_batches, include/nsnp_host.h) takes.  The outputs may be device pointers or pointers into PINNED HOST memory
 * (as for nsnp_pileup_forward_windows_calls): the kernel then writes the 41 bytes per site straight into host memory and the run needs no
 * device-to-host copy at all (nanosnp_amd/pipeline.py call_contigs: its only copy-engine work is the text's way in). */
int nsnp_pileup_rows_unpack(nsnp_ctx* ctx, const double* rows, int64_t N, int64_t* pos, uint8_t* gt_arg, uint8_t* zy_arg,
                            float* gt_max, float* zy_max, float* cov8, void* stream);

/* predict.py:54-65: gt_arg/zy_arg = argmax, gt_max/zy_max = max probability,
 * depth = -(sum of the negative entries of x[n,16,{0,1,2,3,9,10,11,12}]).  All device. */
int nsnp_pileup_postprocess(nsnp_ctx* ctx, const float* gt_prob, const float* zy_prob,
                            const int32_t* x, int64_t N, uint8_t* gt_arg, uint8_t* zy_arg,
                            float* gt_max, float* zy_max, int32_t* depth, void* stream);

/* ---- pileup encode -------------------------------------------------------------------- */
/* bases: device bytes, the concatenated column-5 strings of M mpileup lines; col_off: device
 * int64 [M+1]; ref: device uint8 [M], chr_seq[pos-1] as stored in the FASTA.  Outputs (device):
 * counts int32 [M,18], depth int32 [M], flags uint8 [M] (NSNP_FLAG_*).  min_af is used for the
 * SNP and the indel test (make_predict_data.sh:120-124 passes .12 for both). */
int nsnp_pileup_encode_columns(nsnp_ctx* ctx, const uint8_t* bases, const int64_t* col_off,
                               const uint8_t* ref, int64_t M, double min_af, int min_coverage,
                               int32_t* counts, int32_t* depth, uint8_t* flags, void* stream);

/* The same with the reference program's two thresholds apart (DNA_CreateCanSnpTensor -snp_min_af / -indel_min_af, main.cpp:79-88;
 * tensor_maker.cpp:205-212: an indel allele - the I and D totals - is tested against indel_min_af, a base against snp_min_af). */
int nsnp_pileup_encode_columns2(nsnp_ctx* ctx, const uint8_t* bases, const int64_t* col_off,
                                const uint8_t* ref, int64_t M, double snp_min_af, double indel_min_af, int min_coverage,
                                int32_t* counts, int32_t* depth, uint8_t* flags, void* stream);

/* pos: device int64 [M], the positions in line order (fold the contig index into the high bits
 * when several contigs share a call).  A site is emitted when its 33 columns are 33 consecutive
 * positions - every step + 1, as main.cpp:174-178 resets its window at any other step: positions
 * that repeat or step back (concatenated text) are taken as they come.  center_idx: device int64 [cap] receives the
 * column indices of emitted sites in ascending order; *n_sites (device int64) their number
 * (may exceed cap: then only the first cap were written). */
int nsnp_pileup_select_sites(nsnp_ctx* ctx, const int64_t* pos, const uint8_t* flags, int64_t M,
                             int64_t* center_idx, int64_t cap, int64_t* n_sites, void* stream);

/* nsnp_pileup_select_sites for one chunk of a streamed text: the same selection, and in meta (int64 [4] in any memory the device can write:
 * device or pinned host) { sites selected, how many of them lie in front of column own_lo, in front of column own_hi, sites selected } -
 * the sites the chunk OWNS (its columns without the halo lines it re-reads: main.cpp:174-217 emits a site when its last column has been
 * read) are entries [meta[1], meta[2]) of the ascending list.  No count comes back through the host. */
int nsnp_pileup_select_sites_range(nsnp_ctx* ctx, const int64_t* pos, const uint8_t* flags, int64_t M, int64_t own_lo, int64_t own_hi,
                                   int64_t* center_idx, int64_t cap, int64_t* meta, void* stream);

/* The per-site values PileupModel/predict.py:52-65 hands to its row loop, as one [N,13] float64 array on the device: position, argmax of
 * the genotype / zygosity heads, their maxima, the coverage channels x[:, 16, [0, 1, 2, 3, 9, 10, 11, 12]] (predict.py:63) of the centre
 * column center_idx[n] of counts [M,18].  pos: device int64 [M]; gt_arg / zy_arg / gt_max / zy_max: device arrays of N (the outputs of
 * nsnp_pileup_forward_windows_calls). */
int nsnp_pileup_call_rows(nsnp_ctx* ctx, const int32_t* counts, const int64_t* center_idx, const int64_t* pos, const uint8_t* gt_arg,
                          const uint8_t* zy_arg, const float* gt_max, const float* zy_max, int64_t N, double* rows, void* stream);

/* The reader in front of the column encode, on the device: samtools-mpileup text resident in HBM (whole lines; the end of the
 * text ends its last line) -> per line the position, the reference byte and the column-5 string, in line order:
 *   LineReader::getline  dna_sv_tensor/src/common/line_reader.cpp:95-127  ('\n' or "\r\n" ends a line)
 *   split_line           dna_sv_tensor/src/common/cpp_aux.cpp:43-59       (tokens are maximal runs of non-tab bytes)
 *   create_pileup_tensor make_candidate_snp_tensor/main.cpp:162-172       (ref_off = atoll(token 1), pileup_bases = token 4)
 * text: device uint8 [text_len] (any alignment).  chr_seq: device uint8 [chr_len] as stored in the FASTA, with ref: device uint8
 * [cap_cols] receiving chr_seq[pos - 1] of every line (both may be NULL: no reference bytes).  pos: device int64 [cap_cols];
 * col_off: device int64 [cap_cols + 1]; bases: device uint8 [cap_bytes] - exactly the arrays nsnp_pileup_encode_columns takes.
 * meta: int64 [4] in any memory the device can write (device or pinned host memory): { lines, column-5 bytes, status, 0 }, valid when
 * the stream has passed this call.  status bits: NSNP_TOK_EFORMAT a line with fewer than five fields, NSNP_TOK_BLANK an empty or
 * CR-only line (the reference aborts on both: cpp_aux.cpp:10-21 via main.cpp:165), NSNP_TOK_EPOS a position outside [1, chr_len]
 * (main.cpp:170 asserts), NSNP_TOK_ERANGE lines > cap_cols or bytes > cap_bytes (nothing is written out of bounds; meta holds what is
 * needed).  With any status bit set the outputs must not be used.  One call at a time per context (scratch lives in the context). */
#define NSNP_TOK_EFORMAT 1
#define NSNP_TOK_BLANK   2
#define NSNP_TOK_EPOS    4
#define NSNP_TOK_ERANGE  8
int nsnp_mpileup_tokenise(nsnp_ctx* ctx, const uint8_t* text, int64_t text_len, const uint8_t* chr_seq, int64_t chr_len,
                          int64_t cap_cols, int64_t cap_bytes, int64_t* pos, int64_t* col_off, uint8_t* bases, uint8_t* ref,
                          int64_t* meta, void* stream);

/* x[n][t][c] = counts[center_idx[n]-16+t][c]; x: device int32 [N,33,18]. */
int nsnp_pileup_gather_windows(nsnp_ctx* ctx, const int32_t* counts, const int64_t* center_idx,
                               int64_t N, int32_t* x, void* stream);

/* ---- HaplotypeModel ------------------------------------------------------------------- */
/* seq/bq/mq/hap: device int32 [N,D,L] planes as write_to_bins.py:44-61 stores them (padding
 * rows are -2); ref_row: device int32 [N,L].  out: device fp32 [N,105,L] -- float64 math as
 * numpy does, then the fp32 cast of predict_dev.py:35-36. */
int nsnp_hap_features(nsnp_ctx* ctx, const int32_t* seq, const int32_t* bq, const int32_t* mq,
                      const int32_t* hap, const int32_t* ref_row, int64_t N, int D, int L,
                      float* out, void* stream);

/* The same reduction on int8 read planes (every value of the four planes fits: base codes -2..4, HP -2..3, base
 * quality <= 93, mapping quality <= 60 with -2 as padding): a quarter of the bytes over PCIe and from HBM.  Results are
 * bit-identical to nsnp_hap_features on the widened planes.  ref_row stays int32 [N,L]. */
int nsnp_hap_features_i8(nsnp_ctx* ctx, const int8_t* seq, const int8_t* bq, const int8_t* mq, const int8_t* hap,
                         const int32_t* ref_row, int64_t N, int D, int L, float* out, void* stream);

/* Read arrangement of the stage-4 generator (create_pileup_haplotype.py:140-207, write_to_bins.py:15-61):
 * per site keep the reads whose base at the centre column is non-zero, order them by the HP tag at
 * the centre column (ties keep input order), pad with -2 to D_out rows, cut at D_out.  Inputs are
 * device int32 [N,R,L] read matrices (n_reads[N] valid rows each, NULL = all R); outputs device int32
 * [N,D_out,L] planes in the layout nsnp_hap_features consumes; depth[N] (optional) = rows kept. */
int nsnp_hap_arrange_reads(nsnp_ctx* ctx, const int32_t* seq, const int32_t* bq, const int32_t* mq,
                           const int32_t* hap, const int32_t* n_reads, int64_t N, int R, int L, int D_out,
                           int32_t* oseq, int32_t* obq, int32_t* omq, int32_t* ohap, int32_t* depth, void* stream);

/* host_tensors: the 58 fp32 tensors of model_dev.LSTMNetwork.state_dict() in order
 * (pileup_encoder 26, haplotype_encoder 26, forward_layer 6), HOST pointers.
 * hidden must be a multiple of 64; F = 105 input features; 3 layers as ont_haplotype.yaml. */
int nsnp_hap_load_weights(nsnp_ctx* ctx, const float* const* host_tensors, int n_tensors,
                          int n_features, int hidden, int n_layers, int n_gt, int n_zy);

/* xp: device fp32 [N,105,33], xh: device fp32 [N,105,11] -> gt [N,n_gt], zy [N,n_zy]. */
int nsnp_hap_forward(nsnp_ctx* ctx, const float* xp, const float* xh, int64_t N,
                     float* gt_prob, float* zy_prob, void* stream);

/* ---- legacy haplotype caller (HaplotypeModel/predict.py -> model.CatModel; not run by run_caller.sh) ---- */
/* host_tensors: the 132 floating-point tensors of CatModel(nc0=5,nc1=5,nc2=2,nclass=10,nh=256).state_dict()
 * in order, the integer num_batches_tracked entries skipped (6 ResBlocks x 14, haplotype_base.rnn.{0,1} x 10,
 * haplotype_percentage 26, out_layer 2), HOST pointers.  BatchNorm is applied in eval mode (predict.py:28). */
int nsnp_cat_load_weights(nsnp_ctx* ctx, const float* const* host_tensors, int n_tensors);

/* g0, g1: device fp32 [N,40,11,5] exactly as predict.py:33-34,42-43 hands them to the model
 * (g2 / g3 are ignored by CatModel.predict: model.py:332-358) -> gt_prob device fp32 [N,10] (softmax). */
int nsnp_cat_forward(nsnp_ctx* ctx, const float* g0, const float* g1, int64_t N, float* gt_prob, void* stream);

/* Builds one group tensor [N,40,length,5] fp32 from the per-tag matrices the HDF5 bins hold
 * (dataset.py:862-915): read / base-quality / mapping-quality [N,depth,length] int32 per tag; the first 20
 * rows of tag 1 then of tag 2; planes (base, baseq, mapq, mask = base != -2, phase = 1 | 2).
 * depth1, depth2 >= 20 (NSNP_ESHAPE otherwise: the reference's [:20] slicing would yield a ragged tensor). */
int nsnp_cat_groups(nsnp_ctx* ctx, const int32_t* read1, const int32_t* bq1, const int32_t* mq1, int depth1,
                    const int32_t* read2, const int32_t* bq2, const int32_t* mq2, int depth2,
                    int64_t N, int length, float* g, void* stream);

/* ---- result gather over RCCL (optional; see the note at the top) ---------------------------------------- */
/* Rank 0 obtains 128 opaque bytes and shares them with the other ranks by any side channel; every rank then binds a
 * communicator to its context (collective call).  NSNP_ENOTSUP when no RCCL library can be resolved. */
int nsnp_comm_unique_id(uint8_t* id128);
int nsnp_comm_init(nsnp_ctx* ctx, const uint8_t* id128, int rank, int world);
/* A host that already owns a communicator (nsnp_gather_results(ctx, rccl_comm, ...)) binds it instead:
 * rccl_comm is its ncclComm_t, rank / world as it was created.  The context borrows it: nsnp_comm_destroy and
 * nsnp_ctx_destroy only forget it.  The communicator must come from the RCCL image this library resolves (the one already
 * loaded in the process). */
int nsnp_comm_attach(nsnp_ctx* ctx, void* rccl_comm, int rank, int world);
int nsnp_comm_destroy(nsnp_ctx* ctx);
/* Rooted gather of per-rank byte blocks (device memory) into root_buf (device, root only) at byte_off[r] .. byte_off[r+1].
 * byte_off is a HOST array of world + 1 offsets and is required on EVERY rank (byte_off[0] == 0, non-decreasing,
 * byte_off[rank + 1] - byte_off[rank] == local_bytes): all ranks validate the same table before anything is posted, so a bad
 * plan fails everywhere with NSNP_EINVAL instead of leaving peers in an unmatched send.  Grouped ncclSend / ncclRecv on
 * `stream`, asynchronous; rank order = site order, so the merge is a concatenation.  nsnp_gather_check is the device-free
 * part of that validation (usable to pre-check a plan). */
int nsnp_gather_check(int rank, int world, int64_t local_bytes, const int64_t* byte_off, int root);
int nsnp_gather_results(nsnp_ctx* ctx, const void* local, int64_t local_bytes, void* root_buf,
                        const int64_t* byte_off, int root, void* stream);

#ifdef __cplusplus
}
#endif
#endif
