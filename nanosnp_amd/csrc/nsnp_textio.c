/*
 * nsnp_textio.c -- host-side readers that turn the pipeline's text formats into the flat
 * arrays the device path consumes (see include/nsnp_host.h).
 *
 * Counterparts in the reference (behaviour followed, code is independent):
 *   mpileup line parsing   dna_sv_tensor/src/make_candidate_snp_tensor/main.cpp:162-172
 *   tab tokenisation       dna_sv_tensor/src/common/cpp_aux.cpp:43-59 (runs of tabs collapse)
 *   line endings           dna_sv_tensor/src/common/line_reader.cpp:95-127 (\n, \r\n)
 *   FASTA / .fai           dna_sv_tensor/src/common/ref_reader.cpp:9-61
 *   .pd parsing            dna_sv_tensor/src/make_bin_data/make_bin_predict_data.py:48-77,
 *                          PileupModel/dataset.py:124-135
 */
#include "nsnp_host.h"

#include <ctype.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* next token of a tab-separated line in [p,end); returns token start, sets *tok_end and
 * advances *pp past the token.  NULL when the line has no more tokens. */
static const char* next_tok(const char** pp, const char* end, const char** tok_end)
{
    const char* p = *pp;
    while (p < end && *p == '\t') ++p;
    if (p >= end) { *pp = p; return NULL; }
    const char* s = p;
    const char* t = memchr(p, '\t', (size_t)(end - p));       /* vectorised in libc: the bases token is ~50-100 bytes */
    p = t ? t : end;
    *tok_end = p; *pp = p;
    return s;
}

static int64_t parse_i64(const char* s, const char* e)
{
    int64_t v = 0; int neg = 0;
    while (s < e && isspace((unsigned char)*s)) ++s;
    if (s < e && (*s == '-' || *s == '+')) { neg = (*s == '-'); ++s; }
    while (s < e && *s >= '0' && *s <= '9') { v = v * 10 + (*s - '0'); ++s; }
    return neg ? -v : v;
}

/* one chunk of whole lines: counts (bases == NULL) or fills pos / col_off / bases starting at column m0, byte nb0 */
static int mpileup_chunk(const char* p, const char* end, int64_t* m_out, int64_t* nb_out,
                         int64_t m0, int64_t nb0, int64_t* pos, int64_t* col_off, uint8_t* bases)
{
    int64_t m = m0, nb = nb0;
    while (p < end) {
        const char* le = memchr(p, '\n', (size_t)(end - p));
        const char* next = le ? le + 1 : end;
        if (!le) le = end;
        if (le > p && le[-1] == '\r') --le;
        if (le > p) {
            const char* q = p; const char* te;
            const char* t0 = next_tok(&q, le, &te);                     /* contig   */
            const char* t1 = t0 ? next_tok(&q, le, &te) : NULL;         /* position */
            const char* t1e = te;
            const char* t2 = t1 ? next_tok(&q, le, &te) : NULL;         /* ref base */
            const char* t3 = t2 ? next_tok(&q, le, &te) : NULL;         /* depth    */
            const char* t4 = t3 ? next_tok(&q, le, &te) : NULL;         /* bases    */
            if (!t4) return NSNP_HOST_EFORMAT;
            const int64_t bl = te - t4;
            if (bases) {
                pos[m] = parse_i64(t1, t1e);
                col_off[m] = nb;
                memcpy(bases + nb, t4, (size_t)bl);
            }
            nb += bl; ++m;
        }
        p = next;
    }
    *m_out = m - m0; *nb_out = nb - nb0;
    return 0;
}

/* The text is cut into chunks of whole lines (one per OpenMP thread, at least 1 MB each); a counting pass gives every chunk
 * its first column and byte, the filling pass then writes all chunks at once.  samtools' text is ~100 bytes per column, the
 * encode kernel consumes 10 G columns/s: a single host thread parsing 3 M columns/s would be the whole pipeline. */
int nsnp_mpileup_parse(const char* text, int64_t text_len, int64_t* n_cols, int64_t* n_bytes,
                       int64_t* pos, int64_t* col_off, uint8_t* bases)
{
    if (!text || text_len < 0 || !n_cols || !n_bytes) return NSNP_HOST_EINVAL;
    int T = 1;
#ifdef _OPENMP
    T = omp_get_max_threads();
#endif
    if ((int64_t)T > text_len / (1 << 20)) T = (int)(text_len / (1 << 20));
    if (T < 1) T = 1;
    if (T > 1024) T = 1024;
    const char* cut[1025];
    cut[0] = text; cut[T] = text + text_len;
    for (int c = 1; c < T; ++c) {
        const char* g = text + text_len / T * c;
        if (g < cut[c - 1]) g = cut[c - 1];
        const char* nl = memchr(g, '\n', (size_t)(text + text_len - g));
        cut[c] = nl ? nl + 1 : text + text_len;
    }
    int64_t cm[1025], cb[1025];
    int err = 0;
    #pragma omp parallel for num_threads(T) schedule(static, 1)
    for (int c = 0; c < T; ++c)
        if (mpileup_chunk(cut[c], cut[c + 1], &cm[c], &cb[c], 0, 0, NULL, NULL, NULL)) {
            #pragma omp atomic write
            err = 1;
        }
    if (err) return NSNP_HOST_EFORMAT;
    int64_t m = 0, nb = 0;
    for (int c = 0; c < T; ++c) { const int64_t a = cm[c], b = cb[c]; cm[c] = m; cb[c] = nb; m += a; nb += b; }
    *n_cols = m; *n_bytes = nb;
    if (!bases) return 0;
    #pragma omp parallel for num_threads(T) schedule(static, 1)
    for (int c = 0; c < T; ++c) {
        int64_t dm, db;
        (void)mpileup_chunk(cut[c], cut[c + 1], &dm, &db, cm[c], cb[c], pos, col_off, bases);
    }
    col_off[m] = nb;
    return 0;
}

/* Single-call form for callers that bring their own (e.g. pinned) buffers: every line is tokenised ONCE - each thread records
 * (position, column-5 token) of the lines of its chunk in a growing list - then the lists are laid end to end and the tokens
 * copied.  cap_cols / cap_bytes: capacities of pos / col_off (cap_cols + 1) / bases; text_len / 8 columns and text_len bytes
 * always suffice.  NSNP_HOST_ERANGE when a capacity is too small (n_cols / n_bytes then hold what is needed). */
typedef struct { int64_t pos; const char* tok; int64_t len; } col_rec;

int nsnp_mpileup_parse_into(const char* text, int64_t text_len, int64_t cap_cols, int64_t cap_bytes,
                            int64_t* n_cols, int64_t* n_bytes, int64_t* pos, int64_t* col_off, uint8_t* bases)
{
    if (!text || text_len < 0 || !n_cols || !n_bytes || !pos || !col_off || !bases) return NSNP_HOST_EINVAL;
    int T = 1;
#ifdef _OPENMP
    T = omp_get_max_threads();
#endif
    if ((int64_t)T > text_len / (1 << 20)) T = (int)(text_len / (1 << 20));
    if (T < 1) T = 1;
    if (T > 1024) T = 1024;
    const char* cut[1025];
    cut[0] = text; cut[T] = text + text_len;
    for (int c = 1; c < T; ++c) {
        const char* g = text + text_len / T * c;
        if (g < cut[c - 1]) g = cut[c - 1];
        const char* nl = memchr(g, '\n', (size_t)(text + text_len - g));
        cut[c] = nl ? nl + 1 : text + text_len;
    }
    col_rec* recs[1024];
    int64_t cm[1025], cb[1025];
    int err = 0;
    #pragma omp parallel for num_threads(T) schedule(static, 1)
    for (int c = 0; c < T; ++c) {
        int64_t cap = (cut[c + 1] - cut[c]) / 64 + 1024, m = 0, nb = 0;
        col_rec* r = (col_rec*)malloc((size_t)cap * sizeof(col_rec));
        const char* p = cut[c]; const char* end = cut[c + 1];
        int bad = r == NULL;
        while (!bad && p < end) {
            const char* le = memchr(p, '\n', (size_t)(end - p));
            const char* next = le ? le + 1 : end;
            if (!le) le = end;
            if (le > p && le[-1] == '\r') --le;
            if (le > p) {
                const char* q = p; const char* te;
                const char* t0 = next_tok(&q, le, &te);
                const char* t1 = t0 ? next_tok(&q, le, &te) : NULL;
                const char* t1e = te;
                const char* t2 = t1 ? next_tok(&q, le, &te) : NULL;
                const char* t3 = t2 ? next_tok(&q, le, &te) : NULL;
                const char* t4 = t3 ? next_tok(&q, le, &te) : NULL;
                if (!t4) { bad = 2; break; }
                if (m == cap) {
                    cap *= 2;
                    col_rec* r2 = (col_rec*)realloc(r, (size_t)cap * sizeof(col_rec));
                    if (!r2) { bad = 1; break; }
                    r = r2;
                }
                r[m].pos = parse_i64(t1, t1e); r[m].tok = t4; r[m].len = te - t4;
                nb += te - t4; ++m;
            }
            p = next;
        }
        recs[c] = r; cm[c] = m; cb[c] = nb;
        if (bad) {
            #pragma omp atomic write
            err = bad;
        }
    }
    int64_t m = 0, nb = 0;
    for (int c = 0; c < T; ++c) { const int64_t a = cm[c], b = cb[c]; cm[c] = m; cb[c] = nb; m += a; nb += b; }
    *n_cols = m; *n_bytes = nb;
    int rc = err == 2 ? NSNP_HOST_EFORMAT : (err ? NSNP_HOST_ENOMEM : 0);
    if (!rc && (m > cap_cols || nb > cap_bytes)) rc = NSNP_HOST_ERANGE;
    if (!rc) {
        #pragma omp parallel for num_threads(T) schedule(static, 1)
        for (int c = 0; c < T; ++c) {
            const int64_t m1 = (c + 1 < T ? cm[c + 1] : m) - cm[c];
            int64_t o = cb[c];
            const col_rec* r = recs[c];
            for (int64_t i = 0; i < m1; ++i) {
                pos[cm[c] + i] = r[i].pos; col_off[cm[c] + i] = o;
                memcpy(bases + o, r[i].tok, (size_t)r[i].len);
                o += r[i].len;
            }
        }
        col_off[m] = nb;
    }
    for (int c = 0; c < T; ++c) free(recs[c]);
    return rc;
}

int64_t nsnp_fasta_load_contig(const char* fasta_path, const char* contig, uint8_t* seq, int64_t cap)
{
    if (!fasta_path || !contig) return NSNP_HOST_EINVAL;
    const size_t cl = strlen(contig);
    /* .fai: name \t length \t offset \t bases_per_line \t bytes_per_line */
    int64_t fai_len = -1, fai_off = -1;
    {
        size_t n = strlen(fasta_path) + 5; char* fp = (char*)malloc(n);
        if (!fp) return NSNP_HOST_ENOMEM;
        snprintf(fp, n, "%s.fai", fasta_path);
        FILE* f = fopen(fp, "r"); free(fp);
        if (f) {
            char* line = NULL; size_t lc = 0; ssize_t got;
            while ((got = getline(&line, &lc, f)) > 0) {
                if ((size_t)got > cl && strncmp(line, contig, cl) == 0 && line[cl] == '\t') {
                    long long a = 0, b = 0;
                    if (sscanf(line + cl + 1, "%lld\t%lld", &a, &b) == 2) { fai_len = a; fai_off = b; }
                    break;
                }
            }
            free(line); fclose(f);
        }
    }
    FILE* f = fopen(fasta_path, "r");
    if (!f) return NSNP_HOST_EIO;
    int64_t n = 0; int found = 0;
    char* line = NULL; size_t lc = 0; ssize_t got;
    if (fai_off >= 0) {
        if (fseeko(f, (off_t)fai_off, SEEK_SET) != 0) { fclose(f); return NSNP_HOST_EIO; }
        found = 1;
    }
    while ((got = getline(&line, &lc, f)) > 0) {
        while (got > 0 && (line[got - 1] == '\n' || line[got - 1] == '\r')) --got;
        if (got > 0 && line[0] == '>') {
            if (found) break;
            /* header name ends at the first blank (get_truth.py:95 splits on ' ') */
            size_t nl = 1; while (nl < (size_t)got && !isspace((unsigned char)line[nl])) ++nl;
            found = (nl - 1 == cl && strncmp(line + 1, contig, cl) == 0);
            continue;
        }
        if (!found) continue;
        if (seq) {
            if (n + got > cap) { free(line); fclose(f); return NSNP_HOST_ERANGE; }
            memcpy(seq + n, line, (size_t)got);
        }
        n += got;
        if (fai_len >= 0 && n >= fai_len) break;
    }
    free(line); fclose(f);
    if (!found) return NSNP_HOST_EFORMAT;
    return n;
}

int64_t nsnp_pd_parse(const char* text, int64_t text_len, int32_t* x, int64_t* pos,
                      uint8_t* ref_base, int64_t* ctg_begin, int64_t* ctg_end, int64_t cap_sites)
{
    if (!text || text_len < 0) return NSNP_HOST_EINVAL;
    const char* p = text; const char* end = text + text_len;
    int64_t n = 0;
    while (p < end) {
        const char* le = memchr(p, '\n', (size_t)(end - p));
        const char* next = le ? le + 1 : end;
        if (!le) le = end;
        if (le > p && le[-1] == '\r') --le;
        if (le > p) {
            const char* q = p; const char* te;
            const char* t0 = next_tok(&q, le, &te); const char* t0e = te;   /* tensor       */
            const char* t1 = t0 ? next_tok(&q, le, &te) : NULL;             /* ctg:pos:seq  */
            const char* t1e = te;
            if (!t1) return NSNP_HOST_EFORMAT;
            if (x) {
                if (n >= cap_sites) return NSNP_HOST_ERANGE;
                /* 594 whitespace-separated ints (make_bin_predict_data.py:60-62) */
                int32_t* dst = x + n * 594; int k = 0; const char* s = t0;
                while (s < t0e && k < 594) {
                    while (s < t0e && *s == ' ') ++s;
                    if (s >= t0e) break;
                    const char* e = s; while (e < t0e && *e != ' ') ++e;
                    dst[k++] = (int32_t)parse_i64(s, e);
                    s = e;
                }
                if (k != 594) return NSNP_HOST_EFORMAT;
                /* "ctg:pos:seq" split on ':' (PileupModel/dataset.py:125-131); contig names
                 * containing ':' are handled by splitting from the right */
                const char* c2 = t1e; while (c2 > t1 && c2[-1] != ':') --c2;       /* seq start */
                if (c2 <= t1) return NSNP_HOST_EFORMAT;
                const char* c1 = c2 - 1; while (c1 > t1 && c1[-1] != ':') --c1;    /* pos start */
                if (c1 <= t1) return NSNP_HOST_EFORMAT;
                if (t1e - c2 < 17) return NSNP_HOST_EFORMAT;
                pos[n] = parse_i64(c1, c2 - 1);
                ref_base[n] = (uint8_t)c2[16];
                ctg_begin[n] = t1 - text; ctg_end[n] = (c1 - 1) - text;
            }
            ++n;
        }
        p = next;
    }
    return n;
}
