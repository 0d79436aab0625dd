/*
 * nsnp_host.h -- host-side (CPU, plain C) helpers of the NanoSNP MI355X hot path:
 * synthetic workload generators and the text/binary readers that feed the device path.
 * Built into nanosnp_amd/libnanosnp_host.so with gcc; no GPU, no torch types.
 *
 * These are the native counterparts of the reference's libdnasv readers
 * (dna_sv_tensor/src/common/line_reader.cpp, ref_reader.cpp, cpp_aux.cpp:43-59) and of the
 * text->tensor converters (dna_sv_tensor/src/make_bin_data/make_bin_predict_data.py:48-77),
 * re-designed to hand flat arrays to the device instead of text/HDF5 files.
 *
 * All functions return >= 0 on success and a negative NSNP_HOST_E* code on failure; none
 * aborts the process (the reference's readers abort(): cpp_aux.cpp:10-21).
 */
#ifndef NSNP_HOST_H
#define NSNP_HOST_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NSNP_HOST_EINVAL (-1)
#define NSNP_HOST_ENOMEM (-2)
#define NSNP_HOST_EIO    (-3)
#define NSNP_HOST_EFORMAT (-4)
#define NSNP_HOST_ERANGE (-5)

/* ---- synthetic generators (G1/G2/G3) -------------------------------- */

/* G1/G2: M pileup columns in mpileup column-5 grammar.  ref[M] receives the reference base of
 * each column, col_off[M+1] the byte offsets into bases.  window = 0: plain G1 (a fraction
 * het_rate of columns heterozygous).  window = 33: stand-alone windows, the centre column
 * (c % 33 == 16) forced heterozygous w.p. 0.7, homozygous-alt 0.1, noise-only 0.2.
 * Returns the number of bytes written, or -(needed + 16) when cap is too small / bases NULL. */
int64_t nsnp_synth_columns(uint64_t seed, int64_t M, double coverage, int max_depth,
                           double het_rate, int window, uint8_t* ref, uint8_t* bases,
                           int64_t cap, int64_t* col_off);

/* G3: haplotype read planes [N][D][L] int32 (base, baseq, mapq, hap) + ref_row [N][L]. */
int nsnp_synth_hap_planes(uint64_t seed, int64_t N, double coverage, int D, int L,
                          int32_t* seq, int32_t* bq, int32_t* mq, int32_t* hap, int32_t* ref_row);

/* Columns -> samtools-mpileup text, one line per column: contig \t pos \t N \t depth \t bases \t 'I' x max(depth, 1) \n (the input
 * format of the reference's stage 1: make_predict_data.sh:117 runs samtools mpileup without -f, so the reference column is N).
 * Returns the bytes written, or -(needed + 16) when cap is too small / out NULL.  Used to put whole synthetic contigs on disk for
 * the text-to-VCF measurement. */
int64_t nsnp_columns_to_mpileup_text(const char* contig, int64_t M, const int64_t* pos, const uint8_t* bases, const int64_t* col_off,
                                     char* out, int64_t cap);

/* ---- mpileup text -> column arrays ---------------------------------------------------- */
/* Parses samtools-mpileup text (columns 0,1,4 are used, as
 * make_candidate_snp_tensor/main.cpp:162-172 does; tokens are maximal runs of non-tab bytes,
 * cpp_aux.cpp:43-59).  Two-call protocol: with bases == NULL returns the sizes through
 * n_cols / n_bytes; otherwise fills pos[M], col_off[M+1], bases[n_bytes].  All lines must
 * belong to one contig (the per-chromosome files DNA_ExtractChrPileupData writes). */
int nsnp_mpileup_parse(const char* text, int64_t text_len, int64_t* n_cols, int64_t* n_bytes,
                       int64_t* pos, int64_t* col_off, uint8_t* bases);

/* The same parse in ONE call for callers that bring their own (e.g. pinned) buffers: every line is tokenised once.  cap_cols /
 * cap_bytes are the capacities of pos, col_off (cap_cols + 1 entries) and bases; text_len / 8 columns and text_len bytes always
 * suffice.  NSNP_HOST_ERANGE when a capacity is too small (n_cols / n_bytes then hold what is needed). */
int nsnp_mpileup_parse_into(const char* text, int64_t text_len, int64_t cap_cols, int64_t cap_bytes,
                            int64_t* n_cols, int64_t* n_bytes, int64_t* pos, int64_t* col_off, uint8_t* bases);

/* nsnp_mpileup_parse_into for callers whose bookkeeping counts the LINES of the text (the streamed pipeline: chunks of whole lines
 * with 16 lines of halo, "one line = one column"): n_lines_skipped receives the number of empty / CR-only lines the parser stepped
 * over.  The reference aborts on such a line (main.cpp:162-172 -> cpp_aux.cpp:10-21); such a caller must refuse the text when the
 * count is not zero. */
int nsnp_mpileup_parse_lines(const char* text, int64_t text_len, int64_t cap_cols, int64_t cap_bytes,
                             int64_t* n_cols, int64_t* n_bytes, int64_t* pos, int64_t* col_off, uint8_t* bases,
                             int64_t* n_lines_skipped);

/* ---- staging for the streamed stage-5 pipeline (nanosnp_amd/pipeline.py stream_haplotype) ------------------------------------
 * Replaces the HDF5 reads + per-site Python of HaplotypeModel/dataset_dev.py:92-172 behind predict_dev.py:31-32's DataLoader. */

/* n values of elem_src bytes (4 = int32 as the reference's bins hold them, 1 = int8) from a file (fd >= 0, byte offset src_off: pread,
 * page cache -> destination in one copy) or from memory (fd < 0: src + src_off) into dst as elem_dst-byte values, all host threads at
 * once.  4 -> 1 and 4 -> 2 narrow; *n_out_of_range receives the number of values outside [-128, 127] / [-32768, 32767] (the caller
 * then stages as int32).  Widening is not offered. */
int nsnp_stage_values(int fd, const void* src, int64_t src_off, int elem_src, int64_t n, void* dst, int elem_dst,
                      int64_t* n_out_of_range);

/* n zero-padded fields of `width` bytes holding "ctg:pos" (candidate_positions / haplotype_positions of a bin, write_to_bins.py:49-52)
 * -> pos[n], ctg[n] = index of the contig among n_names names (blob + n_names + 1 offsets), -1 when it is not among them.
 * NSNP_HOST_EFORMAT when a field does not split into exactly two parts at ':' or the position is not a decimal integer (the
 * reference raises there: dataset_dev.py:109-110). */
int nsnp_parse_ctg_pos(const uint8_t* rows, int64_t n, int width, const char* names_blob, const int64_t* names_off, int n_names,
                       int64_t* pos, int32_t* ctg);

/* n zero-padded fields of `width` bytes holding "ctg:pos:ref33" (the `position` array of a .pd.bin, make_bin_predict_data.py:94-97) as
 * PileupModel/dataset.py:127-132 reads them (strip, split at ':' into three parts, int(pos), ord(seq[16])) -> pos[n], ctg[n] (index
 * among the names, -1 = not there), ref_base[n].  NSNP_HOST_EFORMAT where the reference raises (not three parts, a position that is
 * not an integer, a sequence shorter than 17). */
int nsnp_parse_ctg_pos_ref(const uint8_t* rows, int64_t n, int width, const char* names_blob, const int64_t* names_off, int n_names,
                           int64_t* pos, int32_t* ctg, uint8_t* ref_base);

/* out[i, c] = (float) x[i, row, channels[c]] for n staged windows [n, rows, width] of int16 (elem 2) or int32 (elem 4) values: the
 * coverage slice of PileupModel/predict.py:63 taken on the host from the staged pass (OpenMP); NSNP_HOST_EINVAL on a bad shape. */
int nsnp_window_channels(const void* x, int elem, int64_t n, int rows, int width, int row, const int32_t* channels, int n_ch, float* out);

/* threads the host routines use: the OpenMP default cut to the affinity mask and to a cgroup CPU quota (NSNP_HOST_THREADS in the
 * environment overrides the automatic count); nsnp_host_set_threads(n > 0) fixes it for the process, n <= 0 returns to automatic */
int nsnp_host_threads(void);
void nsnp_host_set_threads(int n);

/* ---- FASTA (+.fai) -------------------------------------------------------------------- */
/* Loads one contig of a FASTA file into seq (capacity cap).  Uses the .fai when present
 * (ref_reader.cpp:9-33) and a linear scan otherwise.  Returns the contig length, or a
 * negative code; with seq == NULL only the length is returned. */
int64_t nsnp_fasta_load_contig(const char* fasta_path, const char* contig, uint8_t* seq, int64_t cap);

/* ---- .pd text (make_predict_data/main.cpp:120-123) ------------------------------------ */
/* Parses n_sites lines "594 ints \t ctg:pos:REF33 \t alt_info" into x[N][33][18] int32,
 * pos[N], ref_base[N] (byte 16 of REF33, PileupModel/dataset.py:128-131) and contig ids
 * (index into a caller-visible table is left to Python; here ctg_off[N+1] are byte ranges of
 * the contig names inside text).  With x == NULL returns the number of sites. */
int64_t nsnp_pd_parse(const char* text, int64_t text_len, int32_t* x, int64_t* pos,
                      uint8_t* ref_base, int64_t* ctg_begin, int64_t* ctg_end, int64_t cap_sites);

/* ---- text writers of the predict loops -------------------------------------------------- */
/* One batch of the pileup predict loop -> pileup.vcf rows exactly as PileupModel/predict.py:66-194
 * writes them (the batch boundary matters: see nsnp_vcf.c).  names_blob/name_off: contig name
 * table; contig_id[B]; pos[B]; ref_base[B] (ASCII); gt_arg/zy_arg/gt_prob/zy_prob: argmax and max
 * of the two softmaxes; cov[B*8]: float32 x[:,16,[0,1,2,3,9,10,11,12]].  score_mode 0 = float32
 * arithmetic (NumPy >= 2), 1 = float64 (NumPy 1.x).  Returns bytes written, or -(needed+16)
 * when cap is too small; *n_rows receives the number of rows. */
int64_t nsnp_vcf_format_batch(int64_t B, const char* names_blob, const int64_t* name_off,
                              const int32_t* contig_id, const int64_t* pos, const uint8_t* ref_base,
                              const uint8_t* gt_arg, const uint8_t* zy_arg,
                              const float* gt_prob, const float* zy_prob, const float* cov,
                              int score_mode, char* out, int64_t cap, int64_t* n_rows);

/* All batches of the predict loop at once (N sites cut into consecutive batches of batch_size, the reference's DataLoader
 * batches): byte-identical to nsnp_vcf_format_batch on each slice, formatted on nthreads OpenMP threads (0: nsnp_host_threads()). */
int64_t nsnp_vcf_format_batches(int64_t N, int64_t batch_size, const char* names_blob, const int64_t* name_off,
                                const int32_t* contig_id, const int64_t* pos, const uint8_t* ref_base,
                                const uint8_t* gt_arg, const uint8_t* zy_arg,
                                const float* gt_prob, const float* zy_prob, const float* cov,
                                int score_mode, char* out, int64_t cap, int64_t* n_rows, int nthreads);

/* One rank's share of a sharded run: the rows [first, first + N) of a list of n_total sites whose batches run over the WHOLE list
 * (PileupModel/predict.py:45-47: the DataLoader cuts the whole dataset).  All a row takes from its batch is the batch's length and
 * its first ten argmax values (the gt_output[ti] quirk, predict.py:102-125): heads[10 k .. 10 k + 10) holds them for global batch k
 * (nanosnp_amd/dist.py::batch_heads exchanges them); heads NULL: first must be a multiple of batch_size and the values are read from
 * gt_arg.  The arrays hold the N local rows.  The outputs of consecutive parts, concatenated, are the bytes of
 * nsnp_vcf_format_batches over the whole list. */
int64_t nsnp_vcf_format_batches_part(int64_t N, int64_t batch_size, int64_t first, int64_t n_total, const uint8_t* heads,
                                     const char* names_blob, const int64_t* name_off,
                                     const int32_t* contig_id, const int64_t* pos, const uint8_t* ref_base,
                                     const uint8_t* gt_arg, const uint8_t* zy_arg,
                                     const float* gt_prob, const float* zy_prob, const float* cov,
                                     int score_mode, char* out, int64_t cap, int64_t* n_rows, int nthreads);

/* haplotype.csv rows (HaplotypeModel/predict_dev.py:40-47) */
int64_t nsnp_hap_csv_format(int64_t N, const char* names_blob, const int64_t* name_off,
                            const int32_t* contig_id, const int64_t* pos, const uint8_t* gt_arg,
                            const float* gt_prob, int score_mode, char* out, int64_t cap);

/* calculate_score (predict.py:31-34); *ok = 0 where the Python code raises */
double nsnp_calculate_score(float p, int score_mode, int* ok);

/* test hook: the writers' printf-free decimal output ("%f", round(x, 2) as str(float)) against glibc printf on n pseudo-random
 * doubles incl. exact ties; returns the number of differing texts (0), *first_bad the first such value */
int64_t nsnp_vcf_fmt_selftest(uint64_t seed, int64_t n, double* first_bad);

#ifdef __cplusplus
}
#endif
#endif
