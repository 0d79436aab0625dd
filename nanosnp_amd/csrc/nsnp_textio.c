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
typedef struct { col_rec* r; int64_t cap, m, nb, blank; } rec_list;       /* blank: empty / CR-only lines met (skipped) */

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
        } else ++L->blank;
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
#ifndef NSNP_TOK_PREFETCH
#define NSNP_TOK_PREFETCH 1024     /* bytes the tokenisers prefetch ahead of the block they look at */
#endif
/* the position token as an integer: up to 16 plain digits eight at a time (one unaligned 8-byte load per eight digits, the digits
 * combined pairwise by multiplications - no loop over the digits); anything else (a sign, blanks, more digits) through parse_i64 */
static inline uint64_t eight_digits(uint64_t v)              /* byte 0 = most significant digit, all bytes '0'..'9' */
{
    v -= 0x3030303030303030ull;
    v = v * 10 + (v >> 8);
    return (((v & 0x000000FF000000FFull) * 0x000F424000000064ull) + (((v >> 16) & 0x000000FF000000FFull) * 0x0000271000000001ull)) >> 32;
}
static inline int all_digits(uint64_t v)
{
    return ((v & 0xF0F0F0F0F0F0F0F0ull) == 0x3030303030303030ull) && ((((v + 0x0606060606060606ull) & 0xF0F0F0F0F0F0F0F0ull)) == 0x3030303030303030ull);
}
static inline int64_t parse_pos(const char* t1, const char* t1e, const char* text0)
{
    const int64_t n = t1e - t1;
    if (n >= 1 && n <= 16 && t1e - 16 >= text0) {
        uint64_t lo, hi = 0x3030303030303030ull;
        memcpy(&lo, t1e - 8, 8);                                /* the last eight bytes in front of the token's end */
        if (n < 8) lo = (lo & (~0ull << (8 * (8 - n)))) | (0x3030303030303030ull & ~(~0ull << (8 * (8 - n))));     /* bytes in front of the token: '0' */
        else if (n > 8) {
            memcpy(&hi, t1e - 16, 8);
            if (n < 16) hi = (hi & (~0ull << (8 * (16 - n)))) | (0x3030303030303030ull & ~(~0ull << (8 * (16 - n))));
        }
        if (all_digits(lo) && all_digits(hi)) return (int64_t)(eight_digits(hi) * 100000000ull + eight_digits(lo));
    }
    return parse_i64(t1, t1e);
}

/* line grammar over the event bits of one 64-byte block (shared by the AVX2 and the AVX-512 front ends) */
typedef struct { const char* line; const char* tok; const char* t1; const char* t1e; const char* t4; const char* t4e; int ntok; const char* end; } tok_state;

static inline __attribute__((always_inline)) int tok_block(tok_state* st, uint64_t nlm, uint64_t tbm, const char* base, const char* text0, rec_list* L)
{
    uint64_t live = ~0ull;
    for (;;) {
        const uint64_t ev = (st->ntok >= 5 ? nlm : (nlm | tbm)) & live;
        if (!ev) return 0;
        const int bit = __builtin_ctzll(ev);
        live = bit == 63 ? 0ull : (~0ull << (bit + 1));
        const char* e = base + bit;
        if ((nlm >> bit) & 1) {
            const char* le = (e > st->line && e[-1] == '\r') ? e - 1 : e;
            if (st->ntok < 5 && le > st->tok) {               /* the line's last token ends at the line end */
                if (st->ntok == 1) { st->t1 = st->tok; st->t1e = le; } else if (st->ntok == 4) { st->t4 = st->tok; st->t4e = le; }
                ++st->ntok;
            }
            if (st->ntok >= 5) {
                if (rec_push(L, parse_pos(st->t1, st->t1e, text0), st->t4, st->t4e - st->t4)) return 1;
            } else if (le > st->line) return 2;               /* a non-empty line with fewer than five fields */
            else if (e < st->end) ++L->blank;                  /* (the virtual newline behind a chunk's last byte is not a line) */
            st->line = st->tok = e + 1; st->ntok = 0;
        } else {                                              /* a tab (only looked at while ntok < 5) */
            if (e > st->tok) {
                if (st->ntok == 1) { st->t1 = st->tok; st->t1e = e; } else if (st->ntok == 4) { st->t4 = st->tok; st->t4e = e; }
                ++st->ntok;
            }
            st->tok = e + 1;
        }
    }
}

/* masks of a block that is not wholly inside the text, and of the chunk's last block (a virtual newline behind its last byte) */
static inline void tail_masks(const char* p0, int64_t off, int64_t len, int whole, uint64_t* nlm, uint64_t* tbm)
{
    const int64_t left = len - off;
    if (!whole) {
        *nlm = *tbm = 0;
        for (int64_t k = 0; k < left && k < 64; ++k) { *nlm |= (uint64_t)(p0[off + k] == '\n') << k; *tbm |= (uint64_t)(p0[off + k] == '\t') << k; }
    }
    if (left < 64) {
        const uint64_t keep = left > 0 ? (~0ull >> (64 - left)) : 0ull;
        *nlm &= keep; *tbm &= keep;
        if (len == 0 || p0[len - 1] != '\n') *nlm |= 1ull << left;         /* (left <= 63) */
    }
}

__attribute__((target("avx2"))) static int tokenise_avx2(const char* p0, const char* end, const char* limit, const char* text0, rec_list* L)
{
    const __m256i v_nl = _mm256_set1_epi8('\n'), v_tab = _mm256_set1_epi8('\t');
    const int64_t len = end - p0;
    tok_state st = { p0, p0, NULL, NULL, NULL, NULL, 0, end };
    for (int64_t off = 0; off <= len; off += 64) {
        uint64_t nlm = 0, tbm = 0;
        const int whole = p0 + off + 64 <= limit;
        if (whole) {
            const __m256i a = _mm256_loadu_si256((const __m256i*)(p0 + off)), b = _mm256_loadu_si256((const __m256i*)(p0 + off + 32));
            _mm_prefetch(p0 + off + NSNP_TOK_PREFETCH, _MM_HINT_T0);
            nlm = (uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi8(a, v_nl)) | ((uint64_t)(uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi8(b, v_nl)) << 32);
            tbm = (uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi8(a, v_tab)) | ((uint64_t)(uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi8(b, v_tab)) << 32);
        }
        if (!whole || len - off < 64) tail_masks(p0, off, len, whole, &nlm, &tbm);
        const int rc = tok_block(&st, nlm, tbm, p0 + off, text0, L);
        if (rc) return rc;
    }
    return 0;
}

/* AVX-512 front end: one load and two compare-into-mask instructions per 64-byte block */
__attribute__((target("avx512f,avx512bw"))) static int tokenise_avx512(const char* p0, const char* end, const char* limit, const char* text0, rec_list* L)
{
    const __m512i v_nl = _mm512_set1_epi8('\n'), v_tab = _mm512_set1_epi8('\t');
    const int64_t len = end - p0;
    tok_state st = { p0, p0, NULL, NULL, NULL, NULL, 0, end };
    for (int64_t off = 0; off <= len; off += 64) {
        uint64_t nlm = 0, tbm = 0;
        const int whole = p0 + off + 64 <= limit;
        if (whole) {
            const __m512i a = _mm512_loadu_si512((const void*)(p0 + off));
            _mm_prefetch(p0 + off + NSNP_TOK_PREFETCH, _MM_HINT_T0);
            nlm = _mm512_cmpeq_epi8_mask(a, v_nl); tbm = _mm512_cmpeq_epi8_mask(a, v_tab);
        }
        if (!whole || len - off < 64) tail_masks(p0, off, len, whole, &nlm, &tbm);
        const int rc = tok_block(&st, nlm, tbm, p0 + off, text0, L);
        if (rc) return rc;
    }
    return 0;
}

/* LINE-ORIENTED tokeniser (the default where AVX2 or AVX-512 is present).  The block-oriented one above walks every tab and newline of
 * the text through one serial chain (next event = f(previous event): 15 ns per 90-byte line whatever computes the masks - AVX-512
 * masks, an 8-digit SWAR position and software prefetch each changed nothing).  Here the masks of the 64 bytes AT THE LINE START give
 * the four tabs that end contig / position / reference base / depth by four independent lowest-set-bit steps, the end of the column-5
 * token and the newline come from one more mask step each (a second window for lines beyond 64 bytes); a line that does not look
 * like "four non-empty fields, a tab, a non-empty fifth" in its first 64 bytes (runs of tabs, a short line, a field beyond the window)
 * is handed to the portable tokeniser, one line at a time - same records in every case. */
#define NSNP_TOKENISE_LINES(NAME, TARGET, MASKS64)                                                                                          \
__attribute__((target(TARGET))) static int NAME(const char* p0, const char* end, const char* limit, const char* text0, rec_list* L)        \
{                                                                                                                                           \
    const char* p = p0;                                                                                                                     \
    while (p < end) {                                                                                                                       \
        if (p + 128 > limit) return tokenise_generic(p, end, L);            /* the last lines of the whole text: portable path */          \
        uint64_t nl, tb;                                                                                                                    \
        MASKS64(p, nl, tb);                                                                                                                 \
        uint64_t x = tb;                                                                                                                    \
        const int T0 = x ? __builtin_ctzll(x) : 64; x &= x - 1;                                                                             \
        const int T1 = x ? __builtin_ctzll(x) : 64; x &= x - 1;                                                                             \
        const int T2 = x ? __builtin_ctzll(x) : 64; x &= x - 1;                                                                             \
        const int T3 = x ? __builtin_ctzll(x) : 64;                                                                                         \
        const int N0 = nl ? __builtin_ctzll(nl) : 64;                                                                                       \
        /* four non-empty fields in front of a non-empty fifth, all of it inside the window and in front of the first newline */           \
        int fast = T3 < 62 && T0 > 0 && T1 > T0 + 1 && T2 > T1 + 1 && T3 > T2 + 1 && N0 > T3 + 1 && !((tb >> (T3 + 1)) & 1);                \
        const char* e4 = NULL; const char* nlp = NULL;                                                                                      \
        if (fast) {                                                                                                                         \
            /* end of token 4: the first tab or newline behind T3 + 1; then the newline */                                                 \
            const uint64_t m = (tb | nl) >> (T3 + 2);                                                                                       \
            const char* q = p;                                                                                                              \
            if (m) { e4 = p + T3 + 2 + __builtin_ctzll(m); }                                                                                \
            else {                                                                                                                          \
                for (q = p + 64; e4 == NULL; q += 64) {                                                                                     \
                    if (q >= end) { e4 = end; break; }                                                                                      \
                    if (q + 64 > limit) { const char* z = q; while (z < end && *z != '\t' && *z != '\n') ++z; e4 = z; break; }              \
                    uint64_t n2, t2; MASKS64(q, n2, t2);                                                                                    \
                    const uint64_t m2 = n2 | t2;                                                                                            \
                    if (m2) { e4 = q + __builtin_ctzll(m2); break; }                                                                        \
                }                                                                                                                           \
            }                                                                                                                               \
            if (e4 > end) e4 = end;                                                                                                         \
            if (e4 < end && *e4 == '\n') nlp = e4;                                                                                          \
            else if (e4 >= end) nlp = end;                                                                                                  \
            else {                                                      /* a tab: further fields; the newline behind them */                \
                const char* z = e4 + 1;                                                                                                     \
                const int64_t zo = z - p;                                                                                                   \
                if (zo < 64 && (nl >> zo)) nlp = z + __builtin_ctzll(nl >> zo);                                                             \
                else {                                                                                                                      \
                    for (q = zo < 64 ? p + 64 : z; nlp == NULL; ) {                                                                         \
                        if (q >= end) { nlp = end; break; }                                                                                 \
                        if (q + 64 > limit) { const char* y = q; while (y < end && *y != '\n') ++y; nlp = y; break; }                       \
                        uint64_t n2, t2; MASKS64(q, n2, t2); (void)t2;                                                                      \
                        if (n2) { nlp = q + __builtin_ctzll(n2); break; }                                                                   \
                        q += 64;                                                                                                            \
                    }                                                                                                                       \
                    if (nlp > end) nlp = end;                                                                                               \
                }                                                                                                                           \
            }                                                                                                                               \
            /* the line ends in front of a '\r' that precedes the newline (line_reader.cpp:95-127) */                                       \
            const char* le = (nlp > p && nlp[-1] == '\r') ? nlp - 1 : nlp;                                                                  \
            if (e4 > le) e4 = le;                                                                                                           \
            if (e4 <= p + T3 + 1) fast = 0;                                                                                                 \
        }                                                                                                                                   \
        if (fast) {                                                                                                                         \
            if (rec_push(L, parse_pos(p + T0 + 1, p + T1, text0), p + T3 + 1, e4 - (p + T3 + 1))) return 1;                                 \
            p = nlp < end ? nlp + 1 : end;                                                                                                  \
        } else {                                                                                                                            \
            const char* z = memchr(p, '\n', (size_t)(end - p));                                                                             \
            const char* next = z ? z + 1 : end;                                                                                             \
            const int rc = tokenise_generic(p, next, L);                                                                                    \
            if (rc) return rc;                                                                                                              \
            p = next;                                                                                                                       \
        }                                                                                                                                   \
    }                                                                                                                                       \
    return 0;                                                                                                                               \
}

#define NSNP_MASKS64_AVX2(ptr, nlv, tbv) do {                                                                                               \
    const __m256i a_ = _mm256_loadu_si256((const __m256i*)(ptr)), b_ = _mm256_loadu_si256((const __m256i*)((ptr) + 32));                   \
    const __m256i vn_ = _mm256_set1_epi8('\n'), vt_ = _mm256_set1_epi8('\t');                                                               \
    (nlv) = (uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi8(a_, vn_)) | ((uint64_t)(uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi8(b_, vn_)) << 32); \
    (tbv) = (uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi8(a_, vt_)) | ((uint64_t)(uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi8(b_, vt_)) << 32); \
} while (0)
#define NSNP_MASKS64_AVX512(ptr, nlv, tbv) do {                                                                                             \
    const __m512i a_ = _mm512_loadu_si512((const void*)(ptr));                                                                              \
    (nlv) = _mm512_cmpeq_epi8_mask(a_, _mm512_set1_epi8('\n')); (tbv) = _mm512_cmpeq_epi8_mask(a_, _mm512_set1_epi8('\t'));                 \
} while (0)
NSNP_TOKENISE_LINES(tokenise_lines_avx2, "avx2", NSNP_MASKS64_AVX2)
NSNP_TOKENISE_LINES(tokenise_lines_avx512, "avx512f,avx512bw", NSNP_MASKS64_AVX512)

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

/* 0 portable, 1 / 2 line-oriented AVX2 / AVX-512 (the default: the widest the CPU has), 3 / 4 block-oriented AVX2 / AVX-512.
 * NSNP_PARSE_GENERIC in the environment: 1 forces the portable path, 2 the line-oriented AVX2 path, 3 / 4 the block-oriented ones
 * (tests compare them all) */
static int use_avx2(void)
{
#ifdef NSNP_HAVE_AVX2_PATH
    const char* e = getenv("NSNP_PARSE_GENERIC");
    if (e && e[0] == '1') return 0;
    if (!__builtin_cpu_supports("avx2")) return 0;
    const int has512 = __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512bw");
    if (e && e[0] == '2') return 1;
    if (e && e[0] == '3') return 3;
    if (e && e[0] == '4') return has512 ? 4 : 3;
    return has512 ? 2 : 1;
#else
    return 0;
#endif
}

static int parse_into(const char* text, int64_t text_len, int64_t cap_cols, int64_t cap_bytes,
                      int64_t* n_cols, int64_t* n_bytes, int64_t* pos, int64_t* col_off, uint8_t* bases, int64_t* n_blank)
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
    int64_t m = 0, nb = 0, blank = 0;
    #pragma omp parallel num_threads(T)
    {
        static __thread rec_list tls = { NULL, 0, 0, 0, 0 };
#ifdef _OPENMP
        const int team = omp_get_num_threads(), me = omp_get_thread_num();
#else
        const int team = 1, me = 0;
#endif
        /* (a team smaller than T: thread `me` takes the chunks me, me + team, ... - their records end to end in its one list) */
        int bad = 0;
        tls.m = 0; tls.nb = 0; tls.blank = 0;
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
            bad = vec == 2 ? tokenise_lines_avx512(cut[c], cut[c + 1], limit, text, &tls)
                : vec == 1 ? tokenise_lines_avx2(cut[c], cut[c + 1], limit, text, &tls)
                : vec == 3 ? tokenise_avx2(cut[c], cut[c + 1], limit, text, &tls)         /* (the block-oriented form: kept for the comparison tests) */
                : vec == 4 ? tokenise_avx512(cut[c], cut[c + 1], limit, text, &tls) : tokenise_generic(cut[c], cut[c + 1], &tls);
#else
            bad = tokenise_generic(cut[c], cut[c + 1], &tls);
#endif
            cm[c] = tls.m - m_before; cb[c] = tls.nb - nb_before;
        }
        if (bad) {
            #pragma omp atomic write
            err = bad;
        }
        if (tls.blank) {
            #pragma omp atomic
            blank += tls.blank;
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
    if (n_blank) *n_blank = blank;
    if (!rc) col_off[m] = nb;
    return rc;
}

int nsnp_mpileup_parse_into(const char* text, int64_t text_len, int64_t cap_cols, int64_t cap_bytes,
                            int64_t* n_cols, int64_t* n_bytes, int64_t* pos, int64_t* col_off, uint8_t* bases)
{
    return parse_into(text, text_len, cap_cols, cap_bytes, n_cols, n_bytes, pos, col_off, bases, NULL);
}

/* the same, for callers that count LINES of the text themselves (the streamed pipeline cuts the text into chunks with 16 lines of
 * halo and takes "one line = one column" for granted): n_lines_skipped receives the number of empty / CR-only lines the parser
 * stepped over - the reference aborts on such a line (cpp_aux.cpp:10-21 via main.cpp:162-172), a caller whose bookkeeping depends
 * on the line count must refuse the text when it is not zero */
int nsnp_mpileup_parse_lines(const char* text, int64_t text_len, int64_t cap_cols, int64_t cap_bytes,
                             int64_t* n_cols, int64_t* n_bytes, int64_t* pos, int64_t* col_off, uint8_t* bases, int64_t* n_lines_skipped)
{
    if (!n_lines_skipped) return NSNP_HOST_EINVAL;
    return parse_into(text, text_len, cap_cols, cap_bytes, n_cols, n_bytes, pos, col_off, bases, n_lines_skipped);
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
