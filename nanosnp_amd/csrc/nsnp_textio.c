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
#define _GNU_SOURCE
#include "nsnp_host.h"

#include <ctype.h>
#ifdef __linux__
#include <sched.h>
#endif
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* Threads a host routine should use: the OpenMP default cut to the CPUs this process may run on (affinity mask) and to a cgroup
 * CPU quota (cpu.max of cgroup v2, cpu.cfs_quota_us of v1).  A container with 256 visible CPUs and a quota of 16 otherwise runs
 * 256-thread teams whose busy-waiting workers burn the quota, and the kernel then stalls EVERY thread of the process - the one
 * issuing GPU work included - until the next accounting period (seen as a 2x slower text-to-VCF pipeline). */
static int g_host_threads_override = 0;

/* n > 0: use exactly n threads from now on (one process per GPU on a shared host: the Python loader divides the budget by
 * LOCAL_WORLD_SIZE); n <= 0: back to the automatic count */
void nsnp_host_set_threads(int n) { g_host_threads_override = n > 1024 ? 1024 : n; }

int nsnp_host_threads(void)
{
    static int cached = 0;
    if (g_host_threads_override > 0) return g_host_threads_override;
    if (cached > 0) return cached;
    {
        const char* e = getenv("NSNP_HOST_THREADS");
        if (e && atoi(e) > 0) { cached = atoi(e) > 1024 ? 1024 : atoi(e); return cached; }
    }
    int n = 1;
#ifdef _OPENMP
    n = omp_get_max_threads();
#endif
#ifdef __linux__
    {
        cpu_set_t set;
        if (sched_getaffinity(0, sizeof set, &set) == 0) { const int a = CPU_COUNT(&set); if (a > 0 && a < n) n = a; }
        long long q = -1, per = -1;
        FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r");
        if (f) {
            char qs[32] = {0};
            if (fscanf(f, "%31s %lld", qs, &per) == 2 && strcmp(qs, "max") != 0) q = atoll(qs);
            fclose(f);
        } else {
            FILE* g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r"); FILE* h = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r");
            if (g && h && fscanf(g, "%lld", &q) == 1 && fscanf(h, "%lld", &per) == 1) { /* q < 0: no quota */ } else q = -1;
            if (g) fclose(g);
            if (h) fclose(h);
        }
        if (q > 0 && per > 0) { const int c = (int)((q + per - 1) / per); if (c > 0 && c < n) n = c; }
    }
#endif
    if (n < 1) n = 1;
    if (n > 1024) n = 1024;
    cached = n;
    return n;
}

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
    int T = nsnp_host_threads();
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
typedef struct { col_rec* r; int64_t cap, m, nb; } rec_list;

static int rec_push(rec_list* L, int64_t pos, const char* tok, int64_t len)
{
    if (L->m == L->cap) {
        const int64_t cap = L->cap * 2;
        col_rec* r2 = (col_rec*)realloc(L->r, (size_t)cap * sizeof(col_rec));
        if (!r2) return 1;
        L->r = r2; L->cap = cap;
    }
    L->r[L->m].pos = pos; L->r[L->m].tok = tok; L->r[L->m].len = len;
    L->nb += len; ++L->m;
    return 0;
}

/* the lines of [p, end) -> records; 0, 1 (out of memory) or 2 (a line with fewer than five fields) */
static int tokenise_generic(const char* p, const char* end, rec_list* L)
{
    while (p < end) {
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
            if (!t4) return 2;
            if (rec_push(L, parse_i64(t1, t1e), t4, te - t4)) return 1;
        }
        p = next;
    }
    return 0;
}

#if defined(__x86_64__) && defined(__GNUC__)
#include <immintrin.h>
#define NSNP_HAVE_AVX2_PATH 1
/* first byte equal to c in [p, end), or end.  32 bytes per step where 32 readable bytes remain below `limit` (the end of the whole
 * text: reading past `end` inside the text is harmless), byte by byte behind that */
__attribute__((target("avx2"))) static inline const char* scan_byte_avx2(const char* p, const char* end, const char* limit, char c)
{
    const __m256i needle = _mm256_set1_epi8(c);
    while (p < end && p + 32 <= limit) {
        const unsigned mask = (unsigned)_mm256_movemask_epi8(_mm256_cmpeq_epi8(_mm256_loadu_si256((const __m256i*)p), needle));
        if (mask) { const char* h = p + __builtin_ctz(mask); return h < end ? h : end; }
        p += 32;
    }
    while (p < end && *p != c) ++p;
    return p < end ? p : end;                          /* (the vector loop steps past `end` when the last block holds no hit) */
}

/* The same tokenisation over 64-byte blocks: two compares per 32 bytes give one bit mask of the newlines and one of the tabs of a
 * block, and the line grammar runs over the set bits - a tab or newline that ends a non-empty token counts it (runs of tabs
 * collapse, cpp_aux.cpp:43-59), token 1 is the position, token 4 the column-5 string, and behind token 4 only the newline mask is
 * looked at.  No loop whose trip count depends on a field's length (the per-line memchr calls of the portable path cost 50 ns per
 * 90-byte line, mostly mispredicted exits and call overhead: 1.4 GB/s per thread). */
__attribute__((target("avx2"))) static int tokenise_avx2(const char* p0, const char* end, const char* limit, rec_list* L)
{
    const __m256i v_nl = _mm256_set1_epi8('\n'), v_tab = _mm256_set1_epi8('\t');
    const int64_t len = end - p0;
    const char* line = p0;            /* start of the current line */
    const char* tok = p0;             /* start of the current token (behind the last tab) */
    int ntok = 0;                     /* non-empty tokens of the line so far */
    const char* t1 = NULL; const char* t1e = NULL; const char* t4 = NULL; const char* t4e = NULL;
    for (int64_t off = 0; off <= len; off += 64) {
        uint64_t nlm, tbm;
        const int64_t left = len - off;                       /* bytes of the chunk in this block (may be 0: the virtual newline only) */
        if (p0 + off + 64 <= limit) {
            const __m256i a = _mm256_loadu_si256((const __m256i*)(p0 + off)), b = _mm256_loadu_si256((const __m256i*)(p0 + off + 32));
            nlm = (uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi8(a, v_nl)) | ((uint64_t)(uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi8(b, v_nl)) << 32);
            tbm = (uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi8(a, v_tab)) | ((uint64_t)(uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi8(b, v_tab)) << 32);
        } else {
            nlm = tbm = 0;
            for (int64_t k = 0; k < left && k < 64; ++k) { nlm |= (uint64_t)(p0[off + k] == '\n') << k; tbm |= (uint64_t)(p0[off + k] == '\t') << k; }
        }
        if (left < 64) {                                      /* the chunk ends in this block: a virtual newline behind its last byte */
            const uint64_t keep = left > 0 ? (~0ull >> (64 - left)) : 0ull;
            nlm &= keep; tbm &= keep;
            if (len == 0 || p0[len - 1] != '\n') nlm |= 1ull << left;      /* (left <= 63) */
        }
        uint64_t live = ~0ull;
        for (;;) {
            const uint64_t ev = (ntok >= 5 ? nlm : (nlm | tbm)) & live;
            if (!ev) break;
            const int bit = __builtin_ctzll(ev);
            live = bit == 63 ? 0ull : (~0ull << (bit + 1));
            const char* e = p0 + off + bit;
            if ((nlm >> bit) & 1) {
                const char* le = (e > line && e[-1] == '\r') ? e - 1 : e;
                if (ntok < 5 && le > tok) {                   /* the line's last token ends at the line end */
                    if (ntok == 1) { t1 = tok; t1e = le; } else if (ntok == 4) { t4 = tok; t4e = le; }
                    ++ntok;
                }
                if (ntok >= 5) {
                    int64_t v = 0; const char* d = t1;
                    while (d < t1e && (unsigned)(*d - '0') <= 9u) { v = v * 10 + (*d - '0'); ++d; }
                    if (d != t1e || t1e - t1 > 18) v = parse_i64(t1, t1e);
                    if (rec_push(L, v, t4, t4e - t4)) return 1;
                } else if (le > line) return 2;               /* a non-empty line with fewer than five fields */
                line = tok = e + 1; ntok = 0;
            } else {                                          /* a tab (only looked at while ntok < 5) */
                if (e > tok) {
                    if (ntok == 1) { t1 = tok; t1e = e; } else if (ntok == 4) { t4 = tok; t4e = e; }
                    ++ntok;
                }
                tok = e + 1;
            }
        }
    }
    return 0;
}

/* records -> pos / col_off / bases; tokens are copied 32 bytes at a time while 32 bytes of the thread's own output range and of the
 * text remain behind them (what the copy writes beyond a token is overwritten by the next token of the same thread) */
__attribute__((target("avx2"))) static void place_avx2(const col_rec* r, int64_t m1, int64_t m0, int64_t o, int64_t o_end, const char* limit,
                                                      int64_t* pos, int64_t* col_off, uint8_t* bases)
{
    for (int64_t i = 0; i < m1; ++i) {
        const int64_t len = r[i].len;
        pos[m0 + i] = r[i].pos; col_off[m0 + i] = o;
        const int64_t padded = (len + 31) & ~(int64_t)31;
        if (o + padded <= o_end && r[i].tok + padded <= limit) {
            for (int64_t k = 0; k < len; k += 32)
                _mm256_storeu_si256((__m256i*)(bases + o + k), _mm256_loadu_si256((const __m256i*)(r[i].tok + k)));
        } else memcpy(bases + o, r[i].tok, (size_t)len);
        o += len;
    }
}
#endif

static void place_generic(const col_rec* r, int64_t m1, int64_t m0, int64_t o, int64_t* pos, int64_t* col_off, uint8_t* bases)
{
    for (int64_t i = 0; i < m1; ++i) {
        pos[m0 + i] = r[i].pos; col_off[m0 + i] = o;
        memcpy(bases + o, r[i].tok, (size_t)r[i].len);
        o += r[i].len;
    }
}

/* NSNP_PARSE_GENERIC=1 in the environment forces the portable path (tests compare the two) */
static int use_avx2(void)
{
#ifdef NSNP_HAVE_AVX2_PATH
    const char* e = getenv("NSNP_PARSE_GENERIC");
    if (e && e[0] == '1') return 0;
    return __builtin_cpu_supports("avx2");
#else
    return 0;
#endif
}

int nsnp_mpileup_parse_into(const char* text, int64_t text_len, int64_t cap_cols, int64_t cap_bytes,
                            int64_t* n_cols, int64_t* n_bytes, int64_t* pos, int64_t* col_off, uint8_t* bases)
{
    if (!text || text_len < 0 || !n_cols || !n_bytes || !pos || !col_off || !bases) return NSNP_HOST_EINVAL;
    int T = nsnp_host_threads();
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
    const int vec = use_avx2();
    const char* const limit = text + text_len;
    (void)limit;
    /* ONE parallel region: tokenise - barrier - prefix sums by one thread - barrier - place.  The record lists live in thread-local
     * storage of the OpenMP workers and are kept between calls (the runtime keeps its workers): a fresh 0.4-6 MB allocation per
     * thread and call is an mmap, its page faults and a munmap, all of them serialised on the process's address-space lock. */
    int64_t cm[1025], cb[1025];
    int err = 0, rc = 0;
    int64_t m = 0, nb = 0;
    #pragma omp parallel num_threads(T)
    {
        static __thread rec_list tls = { NULL, 0, 0, 0 };
#ifdef _OPENMP
        const int team = omp_get_num_threads(), me = omp_get_thread_num();
#else
        const int team = 1, me = 0;
#endif
        /* (a team smaller than T: thread `me` takes the chunks me, me + team, ... - their records end to end in its one list) */
        int bad = 0;
        tls.m = 0; tls.nb = 0;
        for (int c = me; c < T && !bad; c += team) {
            const int64_t want = (cut[c + 1] - cut[c]) / 64 + 1024;
            if (tls.cap - tls.m < want) {
                const int64_t cap = tls.m + want;
                col_rec* r2 = (col_rec*)realloc(tls.r, (size_t)cap * sizeof(col_rec));
                if (!r2) { bad = 1; break; }
                tls.r = r2; tls.cap = cap;
            }
            const int64_t m_before = tls.m, nb_before = tls.nb;
#ifdef NSNP_HAVE_AVX2_PATH
            bad = vec ? tokenise_avx2(cut[c], cut[c + 1], limit, &tls) : tokenise_generic(cut[c], cut[c + 1], &tls);
#else
            bad = tokenise_generic(cut[c], cut[c + 1], &tls);
#endif
            cm[c] = tls.m - m_before; cb[c] = tls.nb - nb_before;
        }
        if (bad) {
            #pragma omp atomic write
            err = bad;
        }
        #pragma omp barrier
        #pragma omp single
        {
            for (int c = 0; c < T; ++c) { const int64_t a = cm[c], b = cb[c]; cm[c] = m; cb[c] = nb; m += a; nb += b; }
            cm[T] = m; cb[T] = nb;
            rc = err == 2 ? NSNP_HOST_EFORMAT : (err ? NSNP_HOST_ENOMEM : 0);
            if (!rc && (m > cap_cols || nb > cap_bytes)) rc = NSNP_HOST_ERANGE;
        }   /* (implicit barrier) */
        if (!rc) {
            int64_t at = 0;                             /* this thread's records of chunk c start at `at` of its list */
            for (int c = me; c < T; c += team) {
                const int64_t m1 = cm[c + 1] - cm[c];
#ifdef NSNP_HAVE_AVX2_PATH
                if (vec) place_avx2(tls.r + at, m1, cm[c], cb[c], cb[c + 1], limit, pos, col_off, bases);
                else
#endif
                place_generic(tls.r + at, m1, cm[c], cb[c], pos, col_off, bases);
                at += m1;
            }
        }
        if (tls.cap > (int64_t)(16 << 20) / (int64_t)sizeof(col_rec)) {          /* lists beyond 16 MB are given back */
            free(tls.r); tls.r = NULL; tls.cap = 0;
        }
    }
    *n_cols = m; *n_bytes = nb;
    if (!rc) col_off[m] = nb;
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
