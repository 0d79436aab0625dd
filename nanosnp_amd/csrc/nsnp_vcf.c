/*
 * nsnp_vcf.c -- text writers of the two predict loops (host side, plain C).
 *
 * Restates, row for row and quirk for quirk, the per-site Python loops
 *   PileupModel/predict.py:66-194      (pileup.vcf rows; header :13-27)
 *   HaplotypeModel/predict_dev.py:40-47 (haplotype.csv rows)
 * which top out around 50-100k sites/s in the reference and would otherwise bound the end-to-end
 * rate.  Quirks kept on purpose (the merge stage downstream sees them):
 *   - the "fallback" genotype search indexes gt_output[ti], the batch's ARGMAX ARRAY, with
 *     ti in [0,4,7,9] / [1,2,3,5,6,8] (predict.py:102-109,117-125): it reads the classes predicted
 *     for OTHER sites of the batch, and a batch shorter than the index raises IndexError, which the
 *     bare `except: continue` (predict.py:193-194) turns into a skipped site;
 *   - scalar arithmetic of the coverage features (support_count, depth, af) and of calculate_score
 *     (predict.py:31-34) follows the NumPy generation the loop runs under:
 *       score_mode 1 (the default of the Python wrappers) = NumPy 1.x, the reference's own environment
 *         (Dockerfile:13-29, Miniconda py38): a float32 scalar combined with a Python int/float promotes to
 *         float64, so af is a float64 quotient, the 1e-300 guards act and p == 1 scores 3010.3;
 *       score_mode 0 = NumPy >= 2 (NEP 50, what this repository's container runs): Python scalars are weak,
 *         everything stays float32, the guards vanish and p == 1 raises "math domain error" -> the site is
 *         skipped by the bare except (haplotype.csv: the loop has no except, the batch fails).
 *     Both are pinned by goldens of the reference's predict() (tests/golden/make_golden.py vcf).  The score_mode 0
 *     goldens are a plain run under this container's NumPy 2.2.  The score_mode 1 goldens are an EMULATION of NumPy 1.x
 *     under NumPy 2: make_golden.py widens the float32 arrays to float64 where Tensor.numpy() hands them to the loop, which
 *     reproduces NumPy 1.x value-based promotion for every scalar expression of predict.py (float32 scalar op Python
 *     scalar -> float64) - equivalent for those expressions, but not a run inside the reference's py38 environment, which
 *     this container cannot provide (no NumPy 1.x wheel, no network).
 */
#include "nsnp_host.h"

#include <math.h>
#ifdef _OPENMP
#include <omp.h>
#endif
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* PileupModel/options.py:9-30 */
static const char* GT_LABELS[21] = { "AA", "AC", "AG", "AT", "CC", "CG", "CT", "GG", "GT", "TT",
                                     "DD", "AD", "CD", "GD", "TD", "II", "AI", "CI", "GI", "TI", "ID" };
static const char* ZY_LABELS[3] = { "0/0", "1/1", "0/1" };

static int base_index(char c) { return c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : c == 'T' ? 3 : -1; }

/* ---- decimal output without printf --------------------------------------------------------------------------------------------
 * The row loops print two kinds of numbers: round(x, 2) shown as Python's str(float), and "%f" of the allele frequency.  glibc's
 * printf spends 0.3-0.5 us on each and the rows were formatted twice (sizing pass + writing pass): 2.7 us per row and thread, which
 * made the VCF text the slowest stage of the text-to-VCF pipeline (bench.py --workload e2e).  Both roundings are the correctly
 * rounded decimal of the EXACT binary value, ties to even (printf in the default rounding mode; Python's round(x, 2) and '%f' % x
 * go through the same correctly rounded conversion): a double is m 2^-sh with a 53-bit m, so m * 10^d fits 128 bits and the
 * rounding is one shift and one comparison with the half.  Values outside the fast range (not finite, or 2^40 and beyond) take the printf path. */

/* |v| * scale rounded to the nearest integer, ties to even, exactly.  0: done; -1: outside the fast range */
static int round_scaled(double v, uint32_t scale, uint64_t* out)
{
    uint64_t bits;
    memcpy(&bits, &v, sizeof bits);
    const int e = (int)((bits >> 52) & 0x7ff);
    if (e == 0x7ff) return -1;                                  /* inf / nan */
    if (e == 0) { *out = 0; return 0; }                         /* zero and subnormals: far below half a unit */
    const int sh = 1075 - e;                                    /* |v| = m 2^-sh */
    if (sh < 13) return -1;                                     /* |v| >= 2^40 */
    if (sh > 74) { *out = 0; return 0; }                        /* m * scale < 2^73 <= the half */
    const unsigned __int128 p = (unsigned __int128)((bits & ((1ull << 52) - 1)) | (1ull << 52)) * scale;
    uint64_t n = (uint64_t)(p >> sh);
    const unsigned __int128 rem = p & ((((unsigned __int128)1) << sh) - 1), half = ((unsigned __int128)1) << (sh - 1);
    if (rem > half || (rem == half && (n & 1))) ++n;
    *out = n;
    return 0;
}

static int put_u64(char* dst, uint64_t v)
{
    char tmp[24]; int n = 0;
    do { tmp[n++] = (char)('0' + v % 10); v /= 10; } while (v);
    for (int i = 0; i < n; ++i) dst[i] = tmp[n - 1 - i];
    return n;
}
static int put_i64(char* dst, int64_t v)
{
    if (v < 0) { dst[0] = '-'; return 1 + put_u64(dst + 1, (uint64_t)0 - (uint64_t)v); }
    return put_u64(dst, (uint64_t)v);
}

/* returns 0 and sets *q100 = round(score, 2) in hundredths, or -1 when the Python code would raise (site skipped) */
static int calc_score100(float p32, int mode, int64_t* q100)
{
    double v;
    if (mode == 0) {
        const float a = 1.0f - p32;            /* (1.0 - p) + 1e-300 : 1e-300 is 0 in float32 */
        const float r = a / p32;               /* numpy float32 division: x/0 -> inf, 0/0 -> nan */
        v = (double)r;
    } else {
        const double p = (double)p32;
        v = ((1.0 - p) + 1e-300) / (p + 1e-300);
    }
    if (isnan(v) || v <= 0.0) return -1;       /* math.log: domain error (nan propagates to int(nan): ValueError) */
    static const double log_e_10 = 0x1.bcb7b1526e50dp-2;  /* math.log(math.e, 10) = log(M_E) / log(10.0): the quotient of the two libm values,
                                                            checked against them in nsnp_vcf_fmt_selftest */
    double tmp = (-10.0 * log_e_10) * log(v) + 10.0;
    if (!(tmp > 0.0)) tmp = 0.0;               /* max(tmp, 0) */
    uint64_t n;
    if (round_scaled(tmp, 100, &n)) return -1; /* (unreachable: 0 <= tmp < 3100) */
    *q100 = (int64_t)n;                        /* round(tmp, 2): the correctly rounded 2-decimal */
    return 0;
}

/* str(float) of a 2-decimal rounding given in hundredths: shortest repr, at least one decimal ("12.0", "12.3", "12.34") */
static int fmt_pyfloat100(char* dst, int64_t q100)
{
    int n = put_u64(dst, (uint64_t)(q100 / 100));
    const int f = (int)(q100 % 100);
    dst[n++] = '.';
    dst[n++] = (char)('0' + f / 10);
    if (f % 10) dst[n++] = (char)('0' + f % 10);
    return n;
}

/* "%f" % af */
static int fmt_f6(char* dst, double af)
{
    uint64_t n6;
    if (isnan(af)) { memcpy(dst, "nan", 3); return 3; }          /* Python prints nan without a sign */
    if (round_scaled(af, 1000000u, &n6)) return sprintf(dst, "%f", af);
    int n = 0;
    if (signbit(af)) dst[n++] = '-';
    n += put_u64(dst + n, n6 / 1000000u);
    dst[n++] = '.';
    uint32_t f = (uint32_t)(n6 % 1000000u);
    for (int i = 5; i >= 0; --i) { dst[n + i] = (char)('0' + f % 10); f /= 10; }
    return n + 6;
}

typedef struct { char* p; int64_t len, cap; int overflow, nomem; } sbuf;
static void sb_put(sbuf* b, const char* s, int64_t n)
{
    if (b->len + n > b->cap) { b->overflow = 1; b->len += n; return; }
    memcpy(b->p + b->len, s, (size_t)n); b->len += n;
}

/* one row: ctg \t pos \t . \t ref \t alt \t QUAL \t filter \t . \t GT:GQ:DP:AF \t zy:int(GQ):DP:AF \n   (GQ is the same score as QUAL in
 * every branch of predict.py:66-194) */
static void emit(sbuf* b, const char* ctg, int ctg_len, int64_t pos, char sref, const char* alt,
                 int64_t qual100, const char* filter, const char* zy, float depth, double af)
{
    char stack[512];
    char* line = stack;
    if (ctg_len > 256) {                                 /* (a contig name that long: rare enough for the heap) */
        line = (char*)malloc((size_t)ctg_len + 256);
        if (!line) { b->nomem = 1; return; }
    }
    int n = 0;
    memcpy(line, ctg, (size_t)ctg_len); n += ctg_len;
    line[n++] = '\t';
    n += put_i64(line + n, pos);
    line[n++] = '\t'; line[n++] = '.'; line[n++] = '\t'; line[n++] = sref; line[n++] = '\t';
    for (const char* a = alt; *a; ++a) line[n++] = *a;
    line[n++] = '\t';
    n += fmt_pyfloat100(line + n, qual100);
    line[n++] = '\t';
    for (const char* a = filter; *a; ++a) line[n++] = *a;
    memcpy(line + n, "\t.\tGT:GQ:DP:AF\t", 15); n += 15;
    for (const char* a = zy; *a; ++a) line[n++] = *a;
    line[n++] = ':';
    n += put_u64(line + n, (uint64_t)(qual100 / 100));  /* "%d" % gt_qual: truncation of a non-negative score */
    line[n++] = ':';
    n += put_i64(line + n, (int64_t)depth);             /* "%d" % depth (finite: checked by the caller) */
    line[n++] = ':';
    n += fmt_f6(line + n, af);
    line[n++] = '\n';
    sb_put(b, line, n);
    if (line != stack) free(line);
}

/* J consecutive rows of ONE batch of B sites whose first ten argmax values are head[0 .. min(B, 10)): all a row takes from its batch
 * (the gt_output[ti] quirk and its IndexError).  The arrays point at the first of the J rows. */
static int64_t format_rows(int64_t J, int64_t B, const uint8_t* head, const char* names_blob, const int64_t* name_off,
                           const int32_t* contig_id, const int64_t* pos, const uint8_t* ref_base,
                           const uint8_t* gt_arg, const uint8_t* zy_arg,
                           const float* gt_prob, const float* zy_prob, const float* cov,
                           int score_mode, char* out, int64_t cap, int64_t* n_rows)
{
    sbuf sb = { out, 0, out ? cap : 0, 0, 0 };
    int64_t rows = 0;
    for (int64_t j = 0; j < J; ++j) {
        const int g = gt_arg[j];
        if (g >= 10) continue;                                   /* predict.py:68-69 */
        if (zy_arg[j] > 2) continue;
        const char sref = (char)ref_base[j];
        const char* lab = GT_LABELS[g];
        const char* zy = ZY_LABELS[zy_arg[j]];
        const float* c = cov + j * 8;
        float neg = 0.f;                                         /* cov[cov < 0].sum() */
        for (int k = 0; k < 8; ++k) if (c[k] < 0) neg += c[k];
        const float depth = -1.0f * neg;
        /* alt.replace(sref, '') */
        char alt[4]; int al = 0;
        for (int k = 0; k < 2; ++k) if (lab[k] != sref) alt[al++] = lab[k];
        alt[al] = 0;
        float support = 0.f; int bad = 0;
        for (int k = 0; k < al; ++k) {
            const int bi = base_index(alt[k]);
            if (bi < 0) { bad = 1; break; }                      /* KeyError -> except */
            support += c[bi]; support += c[bi + 4];
        }
        if (bad) continue;
        /* counts are integers below 2^24: the sums are exact either way, only the quotient depends on the mode */
        double af = score_mode ? (double)support / (double)depth      /* NumPy 1.x: float64 scalars; x/0 -> inf/nan */
                               : (double)(support / depth);           /* NumPy 2: float32 quotient */
        if (af > 1.0) af = 1.0;
        if (isnan(depth) || isinf(depth)) continue;              /* "%d" % nan / inf raises */
        int64_t gt_qual, zy_qual;                                /* hundredths */
        if (calc_score100(gt_prob[j], score_mode, &gt_qual)) continue;
        if (calc_score100(zy_prob[j], score_mode, &zy_qual)) continue;
        const int64_t qual = gt_qual < zy_qual ? gt_qual : zy_qual;  /* min(gt_qual, zy_qual) */
        const int32_t ci = contig_id[j];
        const char* ctg = names_blob + name_off[ci];
        const int ctg_len = (int)(name_off[ci + 1] - name_off[ci]);
        char sref_s[2] = { sref, 0 };

        if (al == 0) {                                           /* genotype is hom-ref */
            if (zy_arg[j] == 0) {
                emit(&sb, ctg, ctg_len, pos[j], sref, sref_s, qual, "RefCall", zy, depth, af); ++rows;
            } else if (zy_arg[j] == 1) {                         /* '1/1': predict.py:100-115 */
                static const int TI[4] = { 0, 4, 7, 9 };
                int max_ti = -1; int max_v = -1; int err = 0;
                for (int q = 0; q < 4; ++q) {
                    const int ti = TI[q];
                    if (GT_LABELS[ti][0] == sref) continue;
                    if (ti >= B) { err = 1; break; }             /* IndexError */
                    if ((int)head[ti] > max_v) { max_v = head[ti]; max_ti = ti; }
                }
                if (err) continue;
                /* max_ti == -1 would index labels[-1] = 'ID' in Python */
                char na[2] = { max_ti < 0 ? 'I' : GT_LABELS[max_ti][0], 0 };
                emit(&sb, ctg, ctg_len, pos[j], sref, na, zy_qual, "PASS", zy, depth, af); ++rows;
            } else {                                             /* '0/1': predict.py:116-131 */
                static const int TI[6] = { 1, 2, 3, 5, 6, 8 };
                int max_ti = -1; int max_v = -1; int err = 0;
                for (int q = 0; q < 6; ++q) {
                    const int ti = TI[q];
                    if (ti >= B) { err = 1; break; }
                    if ((int)head[ti] > max_v) { max_v = head[ti]; max_ti = ti; }
                }
                if (err) continue;
                const char* l2 = max_ti < 0 ? "ID" : GT_LABELS[max_ti];
                char na[2] = { l2[0] == sref ? l2[1] : l2[0], 0 };
                emit(&sb, ctg, ctg_len, pos[j], sref, na, zy_qual, "PASS", zy, depth, af); ++rows;
            }
            continue;
        }
        char alt_txt[8];
        if (al == 1) { alt_txt[0] = alt[0]; alt_txt[1] = 0; }
        else if (alt[0] == alt[1]) { alt_txt[0] = alt[0]; alt_txt[1] = 0; }          /* 'AA' -> 'A' */
        else { alt_txt[0] = alt[0]; alt_txt[1] = ','; alt_txt[2] = alt[1]; alt_txt[3] = 0; }
        if (strlen(alt_txt) >= 3 && zy_arg[j] != 2) zy = "1/2";
        /* `alt == sref and zy_output != 0` cannot hold: sref was removed from alt */
        if (zy_arg[j] == 0) {                                    /* predict.py:177-185 */
            emit(&sb, ctg, ctg_len, pos[j], sref, alt_txt, gt_qual, "PASS", zy, depth, af); ++rows;
            continue;
        }
        emit(&sb, ctg, ctg_len, pos[j], sref, alt_txt, qual, "PASS", zy, depth, af); ++rows;
    }
    *n_rows = rows;
    if (sb.nomem) return NSNP_HOST_ENOMEM;
    if (sb.overflow) return -(sb.len + 16);
    return sb.len;
}

int64_t nsnp_vcf_format_batch(int64_t B, const char* names_blob, const int64_t* name_off,
                              const int32_t* contig_id, const int64_t* pos, const uint8_t* ref_base,
                              const uint8_t* gt_arg, const uint8_t* zy_arg,
                              const float* gt_prob, const float* zy_prob, const float* cov,
                              int score_mode, char* out, int64_t cap, int64_t* n_rows)
{
    if (B < 0 || !name_off || !n_rows) return NSNP_HOST_EINVAL;
    return format_rows(B, B, gt_arg, names_blob, name_off, contig_id, pos, ref_base, gt_arg, zy_arg, gt_prob, zy_prob, cov, score_mode, out, cap, n_rows);
}

/* HaplotypeModel/predict_dev.py:40-47: "ctg \t pos \t GT \t qual" with GT = gt_decoded_labels[argmax] */
/* Every batch of the predict loop at once: rows of a batch depend only on that batch (the gt_output[ti] quirk indexes
 * the batch's own argmax array), so the batches are formatted independently on `nthreads` OpenMP threads - a sizing pass,
 * a prefix sum, a writing pass - and land in `out` in batch order, byte-identical to calling nsnp_vcf_format_batch on
 * consecutive slices of `batch_size` sites (PileupModel/predict.py:45-47 DataLoader batches). */
int64_t nsnp_vcf_format_batches(int64_t N, int64_t batch_size, const char* names_blob, const int64_t* name_off,
                                const int32_t* contig_id, const int64_t* pos, const uint8_t* ref_base,
                                const uint8_t* gt_arg, const uint8_t* zy_arg,
                                const float* gt_prob, const float* zy_prob, const float* cov,
                                int score_mode, char* out, int64_t cap, int64_t* n_rows, int nthreads)
{
    return nsnp_vcf_format_batches_part(N, batch_size, 0, N, NULL, names_blob, name_off, contig_id, pos, ref_base, gt_arg, zy_arg, gt_prob, zy_prob,
                                        cov, score_mode, out, cap, n_rows, nthreads);
}

/* The rows [first, first + N) of a site list of n_total sites whose batches run over the WHOLE list (one rank's share of a sharded
 * run): batch k holds the global rows [k batch_size, min((k + 1) batch_size, n_total)) and heads[10 k .. 10 k + 10) are its first ten
 * argmax values (heads NULL: every batch that has rows here must START here - first a multiple of batch_size -, and the values are
 * read from gt_arg).  The arrays hold the N local rows.  Concatenating the outputs of consecutive parts gives the bytes of
 * nsnp_vcf_format_batches over the whole list. */
int64_t nsnp_vcf_format_batches_part(int64_t N, int64_t batch_size, int64_t first, int64_t n_total, const uint8_t* heads,
                                     const char* names_blob, const int64_t* name_off,
                                     const int32_t* contig_id, const int64_t* pos, const uint8_t* ref_base,
                                     const uint8_t* gt_arg, const uint8_t* zy_arg,
                                     const float* gt_prob, const float* zy_prob, const float* cov,
                                     int score_mode, char* out, int64_t cap, int64_t* n_rows, int nthreads)
{
    if (N < 0 || batch_size <= 0 || first < 0 || n_total < first + N || !name_off || !n_rows) return NSNP_HOST_EINVAL;
    if (!heads && N > 0 && first % batch_size) return NSNP_HOST_EINVAL;
    const int64_t kb0 = first / batch_size;                                     /* the first global batch with rows here */
    const int64_t nb = N > 0 ? (first + N - 1) / batch_size - kb0 + 1 : 0;
    if (nthreads <= 0) nthreads = nsnp_host_threads();        /* 0: as many as this process may use */
    if (nthreads > nb) nthreads = nb > 0 ? (int)nb : 1;
    /* every thread formats a contiguous range of batches ONCE into a buffer of its own (grown when a batch does not fit); the
     * pieces are then copied behind one another in thread = batch order */
    typedef struct { char* p; int64_t len, cap, rows; int err; } piece;
    piece* pc = (piece*)calloc((size_t)nthreads, sizeof(piece));
    if (!pc) return NSNP_HOST_ENOMEM;
    #pragma omp parallel num_threads(nthreads)
    {
#ifdef _OPENMP
        const int t = omp_get_thread_num(), T = omp_get_num_threads();
#else
        const int t = 0, T = 1;
#endif
        for (int tt = t; tt < nthreads; tt += T) {             /* (a team smaller than asked for still covers every piece) */
            piece* me = pc + tt;
            const int64_t b0 = nb * tt / nthreads, b1 = nb * (tt + 1) / nthreads;
            me->cap = (b1 - b0) * batch_size * 72 + 4096;
            if (me->cap > N * 72 + 4096) me->cap = N * 72 + 4096;
            me->p = (char*)malloc((size_t)me->cap);
            if (!me->p) { me->err = NSNP_HOST_ENOMEM; continue; }
            for (int64_t b = b0; b < b1 && !me->err; ++b) {
                const int64_t g0 = (kb0 + b) * batch_size;                              /* the batch in global rows: [g0, g0 + B) */
                const int64_t B = (n_total - g0 < batch_size) ? n_total - g0 : batch_size;
                const int64_t lo = g0 > first ? g0 : first, hi = g0 + B < first + N ? g0 + B : first + N;
                const int64_t j0 = lo - first, J = hi - lo;                             /* its rows here: local [j0, j0 + J) */
                const uint8_t* head = heads ? heads + 10 * (kb0 + b) : gt_arg + (g0 - first);
                if (!heads && J < (B < 10 ? B : 10)) { me->err = NSNP_HOST_EINVAL; break; }     /* the ten values are not all here */
                for (;;) {
                    int64_t r = 0;
                    const int64_t need = format_rows(J, B, head, names_blob, name_off, contig_id + j0, pos + j0, ref_base + j0, gt_arg + j0,
                                                     zy_arg + j0, gt_prob + j0, zy_prob + j0, cov + j0 * 8, score_mode,
                                                     me->p + me->len, me->cap - me->len, &r);
                    if (need >= 0) { me->len += need; me->rows += r; break; }
                    if (need > -16) { me->err = (int)need; break; }             /* a real error code */
                    const int64_t want = me->len + (-need - 16), grown = want + want / 2 + 4096;
                    char* q = (char*)realloc(me->p, (size_t)grown);
                    if (!q) { me->err = NSNP_HOST_ENOMEM; break; }
                    me->p = q; me->cap = grown;
                }
            }
        }
    }
    int64_t total = 0, total_rows = 0;
    int err = 0;
    for (int t = 0; t < nthreads; ++t) { total += pc[t].len; total_rows += pc[t].rows; if (pc[t].err && !err) err = pc[t].err; }
    *n_rows = total_rows;
    int64_t rc = total;
    if (err) rc = err;
    else if (!out || cap < total) rc = -(total + 16);
    else {
        int64_t at = 0;
        for (int t = 0; t < nthreads; ++t) { pc[t].cap = at; at += pc[t].len; }       /* cap re-used: the piece's offset in out */
        #pragma omp parallel for num_threads(nthreads) schedule(static, 1)
        for (int t = 0; t < nthreads; ++t) memcpy(out + pc[t].cap, pc[t].p, (size_t)pc[t].len);
    }
    for (int t = 0; t < nthreads; ++t) free(pc[t].p);
    free(pc);
    return rc;
}

/* haplotype.csv rows (HaplotypeModel/predict_dev.py:40-47): "ctg \t pos \t GT \t qual" with GT = gt_decoded_labels[argmax] */
int64_t nsnp_hap_csv_format(int64_t N, const char* names_blob, const int64_t* name_off,
                            const int32_t* contig_id, const int64_t* pos, const uint8_t* gt_arg,
                            const float* gt_prob, int score_mode, char* out, int64_t cap)
{
    if (N < 0 || !name_off) return NSNP_HOST_EINVAL;
    sbuf sb = { out, 0, out ? cap : 0, 0, 0 };
    for (int64_t j = 0; j < N; ++j) {
        int64_t q;
        if (gt_arg[j] > 20) return NSNP_HOST_ERANGE;
        if (calc_score100(gt_prob[j], score_mode, &q)) return NSNP_HOST_ERANGE;   /* the reference loop has no try/except */
        const int32_t ci = contig_id[j];
        char tail[64];
        int n = 0;
        tail[n++] = '\t';
        n += put_i64(tail + n, pos[j]);
        tail[n++] = '\t';
        for (const char* a = GT_LABELS[gt_arg[j]]; *a; ++a) tail[n++] = *a;
        tail[n++] = '\t';
        n += fmt_pyfloat100(tail + n, q);
        tail[n++] = '\n';
        sb_put(&sb, names_blob + name_off[ci], name_off[ci + 1] - name_off[ci]);
        sb_put(&sb, tail, n);
    }
    if (sb.overflow) return -(sb.len + 16);
    return sb.len;
}

double nsnp_calculate_score(float p, int score_mode, int* ok)
{
    int64_t q100 = 0;
    const int rc = calc_score100(p, score_mode, &q100);
    if (ok) *ok = rc == 0;
    char buf[32];
    buf[fmt_pyfloat100(buf, q100)] = 0;
    return strtod(buf, NULL);                  /* the double nearest to the 2-decimal: what round(x, 2) returns */
}

/* Test hook (tests/test_vcf.py): the printf-free decimal output against glibc's printf on n pseudo-random doubles - uniform bit
 * patterns scaled into the ranges the writers see, exact binary ties (k / 2^j), values next to ties, negative values and zeros.
 * Returns the number of values whose text differs (0 expected); first_bad receives the first such value. */
int64_t nsnp_vcf_fmt_selftest(uint64_t seed, int64_t n, double* first_bad)
{
    int64_t bad = 0;
    uint64_t s = seed ? seed : 1;
    {
        volatile double e = M_E, ten = 10.0;                             /* (volatile: evaluated by libm at run time, not folded) */
        if (log(e) / log(ten) != 0x1.bcb7b1526e50dp-2) { if (first_bad) *first_bad = log(e) / log(ten); return -1; }
    }
    for (int64_t i = 0; i < n; ++i) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;                       /* xorshift64 */
        double v;
        switch (i & 7) {
        case 0: v = (double)(s >> 11) * (1.0 / 9007199254740992.0); break;                 /* [0, 1) */
        case 1: v = (double)(s >> 11) * (3100.0 / 9007199254740992.0); break;              /* [0, 3100) */
        case 2: v = (double)((s >> 20) % 400000) / (double)(1ull << (1 + (s & 15))); break; /* k / 2^j: exact ties of both roundings */
        case 3: v = nextafter((double)((s >> 20) % 400000) / 128.0, (s & 1) ? 1e9 : -1e9); break;
        case 4: v = -(double)(s >> 11) * (2.0 / 9007199254740992.0); break;
        case 5: v = (double)((s >> 24) % 4000000) / 1000.0; break;                         /* decimal-looking inputs: 0.125, 2.675, ... */
        case 6: v = ldexp((double)(s >> 11), -(int)(53 + (s & 127))); break;               /* tiny values down to 2^-127 */
        default: { const float a = (float)((s >> 40) % 300), b = (float)(1 + (s >> 20) % 300); v = (double)a / (double)b; } break;
        }
        char a[64], b[64];
        int na = fmt_f6(a, v); a[na] = 0;
        snprintf(b, sizeof b, "%f", v);
        int diff = strcmp(a, b) != 0;
        if (!diff && v >= 0.0 && v < 3100.0) {
            uint64_t q;
            snprintf(b, sizeof b, "%.2f", v);
            if (round_scaled(v, 100, &q)) diff = 1;
            else {
                const double r = strtod(b, NULL);
                char c[64];
                int nc = sprintf(c, "%.2f", r);
                while (nc > 0 && c[nc - 1] == '0' && c[nc - 2] != '.') c[--nc] = 0;       /* str(float) of a 2-decimal rounding */
                a[fmt_pyfloat100(a, (int64_t)q)] = 0;
                diff = strcmp(a, c) != 0;
            }
        }
        if (diff) { if (!bad && first_bad) *first_bad = v; ++bad; }
    }
    return bad;
}
