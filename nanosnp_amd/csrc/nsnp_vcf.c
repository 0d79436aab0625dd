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
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* PileupModel/options.py:9-30 */
static const char* GT_LABELS[21] = { "AA", "AC", "AG", "AT", "CC", "CG", "CT", "GG", "GT", "TT",
                                     "DD", "AD", "CD", "GD", "TD", "II", "AI", "CI", "GI", "TI", "ID" };
static const char* ZY_LABELS[3] = { "0/0", "1/1", "0/1" };

static int base_index(char c) { return c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : c == 'T' ? 3 : -1; }

/* returns 0 and sets *q, or -1 when the Python code would raise (site skipped) */
static int calc_score(float p32, int mode, double* q)
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
    const double log_e_10 = log(M_E) / log(10.0);
    double tmp = (-10.0 * log_e_10) * log(v) + 10.0;
    if (!(tmp > 0.0)) tmp = 0.0;               /* max(tmp, 0) */
    char buf[64];
    snprintf(buf, sizeof buf, "%.2f", tmp);    /* round(tmp, 2): correctly rounded decimal */
    *q = strtod(buf, NULL);
    return 0;
}

/* str(float) for a value that is a 2-decimal rounding: shortest repr, at least one decimal */
static int fmt_pyfloat(char* dst, double v)
{
    int n = sprintf(dst, "%.2f", v);
    while (n > 0 && dst[n - 1] == '0' && dst[n - 2] != '.') dst[--n] = 0;
    return n;
}

typedef struct { char* p; int64_t len, cap; int overflow; } sbuf;
static void sb_put(sbuf* b, const char* s, int64_t n)
{
    if (b->len + n > b->cap) { b->overflow = 1; b->len += n; return; }
    memcpy(b->p + b->len, s, (size_t)n); b->len += n;
}

static void emit(sbuf* b, const char* ctg, int ctg_len, int64_t pos, char sref, const char* alt,
                 double qual_field, const char* filter, const char* zy, double gq, float depth, double af)
{
    char line[512]; char q1[32];
    fmt_pyfloat(q1, qual_field);
    char aftxt[64];
    if (isnan(af)) strcpy(aftxt, "nan");            /* Python prints nan without a sign */
    else snprintf(aftxt, sizeof aftxt, "%f", af);
    int n = snprintf(line, sizeof line, "%.*s\t%lld\t.\t%c\t%s\t%s\t%s\t.\tGT:GQ:DP:AF\t%s:%lld:%lld:%s\n",
                     ctg_len, ctg, (long long)pos, sref, alt, q1, filter, zy, (long long)gq,
                     (long long)depth, aftxt);
    sb_put(b, line, n);
}

int64_t nsnp_vcf_format_batch(int64_t B, const char* names_blob, const int64_t* name_off,
                              const int32_t* contig_id, const int64_t* pos, const uint8_t* ref_base,
                              const uint8_t* gt_arg, const uint8_t* zy_arg,
                              const float* gt_prob, const float* zy_prob, const float* cov,
                              int score_mode, char* out, int64_t cap, int64_t* n_rows)
{
    if (B < 0 || !name_off || !n_rows) return NSNP_HOST_EINVAL;
    sbuf sb = { out, 0, out ? cap : 0, 0 };
    int64_t rows = 0;
    for (int64_t j = 0; j < B; ++j) {
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
        double gt_qual, zy_qual;
        if (calc_score(gt_prob[j], score_mode, &gt_qual)) continue;
        if (calc_score(zy_prob[j], score_mode, &zy_qual)) continue;
        const double qual = gt_qual < zy_qual ? gt_qual : zy_qual;   /* min(gt_qual, zy_qual) */
        const int32_t ci = contig_id[j];
        const char* ctg = names_blob + name_off[ci];
        const int ctg_len = (int)(name_off[ci + 1] - name_off[ci]);
        char sref_s[2] = { sref, 0 };

        if (al == 0) {                                           /* genotype is hom-ref */
            if (zy_arg[j] == 0) {
                emit(&sb, ctg, ctg_len, pos[j], sref, sref_s, qual, "RefCall", zy, qual, depth, af); ++rows;
            } else if (zy_arg[j] == 1) {                         /* '1/1': predict.py:100-115 */
                static const int TI[4] = { 0, 4, 7, 9 };
                int max_ti = -1; int max_v = -1; int err = 0;
                for (int q = 0; q < 4; ++q) {
                    const int ti = TI[q];
                    if (GT_LABELS[ti][0] == sref) continue;
                    if (ti >= B) { err = 1; break; }             /* IndexError */
                    if ((int)gt_arg[ti] > max_v) { max_v = gt_arg[ti]; max_ti = ti; }
                }
                if (err) continue;
                /* max_ti == -1 would index labels[-1] = 'ID' in Python */
                char na[2] = { max_ti < 0 ? 'I' : GT_LABELS[max_ti][0], 0 };
                emit(&sb, ctg, ctg_len, pos[j], sref, na, zy_qual, "PASS", zy, zy_qual, depth, af); ++rows;
            } else {                                             /* '0/1': predict.py:116-131 */
                static const int TI[6] = { 1, 2, 3, 5, 6, 8 };
                int max_ti = -1; int max_v = -1; int err = 0;
                for (int q = 0; q < 6; ++q) {
                    const int ti = TI[q];
                    if (ti >= B) { err = 1; break; }
                    if ((int)gt_arg[ti] > max_v) { max_v = gt_arg[ti]; max_ti = ti; }
                }
                if (err) continue;
                const char* l2 = max_ti < 0 ? "ID" : GT_LABELS[max_ti];
                char na[2] = { l2[0] == sref ? l2[1] : l2[0], 0 };
                emit(&sb, ctg, ctg_len, pos[j], sref, na, zy_qual, "PASS", zy, zy_qual, depth, af); ++rows;
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
            emit(&sb, ctg, ctg_len, pos[j], sref, alt_txt, gt_qual, "PASS", zy, gt_qual, depth, af); ++rows;
            continue;
        }
        emit(&sb, ctg, ctg_len, pos[j], sref, alt_txt, qual, "PASS", zy, qual, depth, af); ++rows;
    }
    *n_rows = rows;
    if (sb.overflow) return -(sb.len + 16);
    return sb.len;
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
    if (N < 0 || batch_size <= 0 || !name_off || !n_rows) return NSNP_HOST_EINVAL;
    const int64_t nb = (N + batch_size - 1) / batch_size;
    int64_t* size = (int64_t*)malloc(sizeof(int64_t) * (size_t)(2 * nb + 1));
    if (!size) return NSNP_HOST_ENOMEM;
    int64_t* rows = size + nb + 1;
    if (nthreads <= 0) nthreads = 1;
    int err = 0;
    #pragma omp parallel for num_threads(nthreads) schedule(dynamic, 8)
    for (int64_t b = 0; b < nb; ++b) {
        const int64_t j0 = b * batch_size, B = (N - j0 < batch_size) ? N - j0 : batch_size;
        int64_t r = 0;
        const int64_t need = nsnp_vcf_format_batch(B, names_blob, name_off, contig_id + j0, pos + j0, ref_base + j0, gt_arg + j0,
                                                   zy_arg + j0, gt_prob + j0, zy_prob + j0, cov + j0 * 8, score_mode, NULL, 0, &r);
        if (need < 0 && need > -16) { err = 1; size[b + 1] = 0; rows[b] = 0; }                  /* a real error code */
        else { size[b + 1] = need < 0 ? -need - 16 : need; rows[b] = r; }
    }
    if (err) { free(size); return NSNP_HOST_EINVAL; }
    size[0] = 0;
    int64_t total_rows = 0;
    for (int64_t b = 0; b < nb; ++b) { size[b + 1] += size[b]; total_rows += rows[b]; }
    const int64_t total = size[nb];
    *n_rows = total_rows;
    if (!out || cap < total) { free(size); return -(total + 16); }
    #pragma omp parallel for num_threads(nthreads) schedule(dynamic, 8)
    for (int64_t b = 0; b < nb; ++b) {
        const int64_t j0 = b * batch_size, B = (N - j0 < batch_size) ? N - j0 : batch_size;
        int64_t r = 0;
        (void)nsnp_vcf_format_batch(B, names_blob, name_off, contig_id + j0, pos + j0, ref_base + j0, gt_arg + j0,
                                    zy_arg + j0, gt_prob + j0, zy_prob + j0, cov + j0 * 8, score_mode,
                                    out + size[b], size[b + 1] - size[b], &r);
    }
    free(size);
    return total;
}

int64_t nsnp_hap_csv_format(int64_t N, const char* names_blob, const int64_t* name_off,
                            const int32_t* contig_id, const int64_t* pos, const uint8_t* gt_arg,
                            const float* gt_prob, int score_mode, char* out, int64_t cap)
{
    if (N < 0 || !name_off) return NSNP_HOST_EINVAL;
    sbuf sb = { out, 0, out ? cap : 0, 0 };
    for (int64_t j = 0; j < N; ++j) {
        double q;
        if (gt_arg[j] > 20) return NSNP_HOST_ERANGE;
        if (calc_score(gt_prob[j], score_mode, &q)) return NSNP_HOST_ERANGE;   /* the reference loop has no try/except */
        const int32_t ci = contig_id[j];
        char line[256], qs[32];
        fmt_pyfloat(qs, q);
        int n = snprintf(line, sizeof line, "%.*s\t%lld\t%s\t%s\n", (int)(name_off[ci + 1] - name_off[ci]),
                         names_blob + name_off[ci], (long long)pos[j], GT_LABELS[gt_arg[j]], qs);
        sb_put(&sb, line, n);
    }
    if (sb.overflow) return -(sb.len + 16);
    return sb.len;
}

double nsnp_calculate_score(float p, int score_mode, int* ok)
{
    double q = 0.0;
    const int rc = calc_score(p, score_mode, &q);
    if (ok) *ok = rc == 0;
    return q;
}
