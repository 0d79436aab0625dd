/*
 * nsnp_synth.c -- deterministic synthetic workload generators (host side, plain C).
 *
 * The reference ships no data and no generator; these produce the synthetic inputs that
 * SURVEY.md section 8(d) specifies for the benchmark and the parity tests:
 *   G1  pileup columns in samtools-mpileup column-5 grammar (what
 *       dna_sv_tensor/src/make_candidate_snp_tensor/tensor_maker.cpp:83-114 parses), depth
 *       ~ Poisson(coverage) clipped to max_depth (make_predict_data.sh:117 uses 144)
 *   G2  stand-alone 33-column windows whose centre column is forced to a variant class
 *   G3  haplotype read planes as HaplotypeModel/write_to_bins.py:44-61 lays them out
 *       (base 1..4, deletion -1, not covering 0, padding -2; HP 1/2/3)
 * Everything is a pure function of (seed, index): chunks of 1024 columns / single sites get
 * their own xoshiro256** stream, so output does not depend on the thread count.
 */
#include "nsnp_host.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct { uint64_t s[4]; } rng_t;

static inline uint64_t splitmix64(uint64_t* x)
{
    uint64_t z = (*x += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static inline void rng_seed(rng_t* r, uint64_t seed, uint64_t stream)
{
    uint64_t x = seed * 0xD1342543DE82EF95ull + stream * 0x2545F4914F6CDD1Dull + 1;
    for (int i = 0; i < 4; ++i) r->s[i] = splitmix64(&x);
}
static inline uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
static inline uint64_t rng_next(rng_t* r)
{
    uint64_t* s = r->s;
    const uint64_t result = rotl(s[1] * 5, 7) * 9, t = s[1] << 17;
    s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3]; s[2] ^= t; s[3] = rotl(s[3], 45);
    return result;
}
static inline double rng_u(rng_t* r) { return (double)(rng_next(r) >> 11) * (1.0 / 9007199254740992.0); }
static inline int rng_int(rng_t* r, int n) { return (int)(rng_u(r) * n); }

static int rng_poisson(rng_t* r, double lam)
{
    /* inversion by sequential search; fine for lam <= ~700 */
    double p = exp(-lam), s = p, u = rng_u(r);
    int k = 0;
    while (u > s && k < 100000) { ++k; p *= lam / k; s += p; }
    return k;
}
static double rng_normal(rng_t* r)
{
    double u1 = rng_u(r), u2 = rng_u(r);
    if (u1 < 1e-300) u1 = 1e-300;
    return sqrt(-2.0 * log(u1)) * cos(6.283185307179586 * u2);
}

static const char UP[4] = { 'A', 'C', 'G', 'T' };
static const char LO[4] = { 'a', 'c', 'g', 't' };

/* kind: 0 noise-only, 1 heterozygous (each read carries alt w.p. 0.5), 2 homozygous alt.
 * One 64-bit draw per read, sliced into independent fields (probabilities are k/1024):
 *   strand 0.5 | "^I" prefix 10/1024 | symbol: random base 31/1024, '*'/'#' 20/1024, else ref (or
 *   alt) | indel 31/1024 with length 1..3, sign and up to 3 bases | "$" suffix 10/1024 */
static int64_t gen_column(rng_t* r, int ref_idx, int kind, double coverage, int max_depth,
                          uint8_t* out /* may be NULL: size only */)
{
    int d = rng_poisson(r, coverage);
    if (d > max_depth) d = max_depth;
    if (d < 1) d = 1; /* samtools omits zero-depth positions; keep columns contiguous */
    const int alt_idx = (ref_idx + 1) & 3;
    int64_t n = 0;
#define PUT(ch) do { const uint8_t ch_ = (uint8_t)(ch); if (out) out[n] = ch_; ++n; } while (0)
    for (int i = 0; i < d; ++i) {
        const uint64_t x = rng_next(r);
        const int rev = (int)(x & 1);
        const char* tab = rev ? LO : UP;
        if (((x >> 1) & 1023) < 10) { PUT('^'); PUT('I'); }
        const unsigned u = (unsigned)((x >> 11) & 1023);
        int sym;
        if (kind == 2 || (kind == 1 && ((x >> 21) & 1))) sym = tab[alt_idx];
        else if (u < 31) sym = tab[(x >> 51) & 3];
        else if (u < 51) sym = rev ? '#' : '*';
        else sym = tab[ref_idx];
        PUT(sym);
        if (((x >> 22) & 1023) < 31) {
            const int len = 1 + (int)(((x >> 32) & 0xff) % 3);
            PUT(((x >> 40) & 1) ? '+' : '-');
            PUT('0' + len);
            for (int k = 0; k < len; ++k) PUT(tab[(x >> (41 + 2 * k)) & 3]);
        }
        if (((x >> 53) & 1023) < 10) PUT('$');
    }
#undef PUT
    return n;
}

#define CHUNK 1024

int64_t nsnp_synth_columns(uint64_t seed, int64_t M, double coverage, int max_depth,
                           double het_rate, int window, uint8_t* ref, uint8_t* bases,
                           int64_t cap, int64_t* col_off)
{
    if (M < 0 || !ref || !col_off) return NSNP_HOST_EINVAL;
    const int64_t n_chunk = (M + CHUNK - 1) / CHUNK;
    int64_t* chunk_bytes = (int64_t*)calloc((size_t)n_chunk + 1, sizeof(int64_t));
    if (!chunk_bytes) return NSNP_HOST_ENOMEM;
    for (int pass = 0; pass < 2; ++pass) {
        #pragma omp parallel for num_threads(nsnp_host_threads()) schedule(dynamic, 16)
        for (int64_t ch = 0; ch < n_chunk; ++ch) {
            rng_t r; rng_seed(&r, seed, (uint64_t)ch);
            const int64_t c0 = ch * CHUNK, c1 = (c0 + CHUNK < M) ? c0 + CHUNK : M;
            int64_t off = pass ? chunk_bytes[ch] : 0;
            for (int64_t c = c0; c < c1; ++c) {
                const int ref_idx = rng_int(&r, 4);
                int kind;
                const double u = rng_u(&r);
                if (window > 0 && (c % window) == window / 2)
                    kind = u < 0.7 ? 1 : (u < 0.8 ? 2 : 0);          /* G2 centre column */
                else
                    kind = u < het_rate ? 1 : 0;                     /* G1 */
                if (pass) { ref[c] = (uint8_t)UP[ref_idx]; col_off[c] = off; }
                off += gen_column(&r, ref_idx, kind, coverage, max_depth, pass ? bases + off : NULL);
            }
            if (!pass) chunk_bytes[ch] = off;
        }
        if (!pass) {
            int64_t tot = 0;
            for (int64_t ch = 0; ch < n_chunk; ++ch) { int64_t b = chunk_bytes[ch]; chunk_bytes[ch] = tot; tot += b; }
            chunk_bytes[n_chunk] = tot;
            if (!bases || tot > cap) { free(chunk_bytes); return -tot - 16; /* needed size, negated */ }
        }
    }
    const int64_t total = chunk_bytes[n_chunk];
    col_off[M] = total;
    free(chunk_bytes);
    return total;
}

int nsnp_synth_hap_planes(uint64_t seed, int64_t N, double coverage, int D, int L,
                          int32_t* seq, int32_t* bq, int32_t* mq, int32_t* hap, int32_t* ref_row)
{
    if (N < 0 || D <= 0 || L <= 0 || !seq || !bq || !mq || !hap || !ref_row) return NSNP_HOST_EINVAL;
    #pragma omp parallel for num_threads(nsnp_host_threads()) schedule(dynamic, 64)
    for (int64_t n = 0; n < N; ++n) {
        rng_t r; rng_seed(&r, seed ^ 0x5851F42D4C957F2Dull, (uint64_t)n);
        int32_t* s = seq + n * D * L; int32_t* b = bq + n * D * L;
        int32_t* m = mq + n * D * L;  int32_t* h = hap + n * D * L;
        int32_t cons[64]; /* per-column consensus base */
        for (int l = 0; l < L; ++l) { ref_row[n * L + l] = 1 + rng_int(&r, 4); cons[l & 63] = 1 + rng_int(&r, 4); }
        int depth = rng_poisson(&r, coverage);
        if (depth > D) depth = D;
        /* HP per read, then rows ordered by HP as create_pileup_haplotype.py:158-165 leaves them */
        int hp_cnt[4] = { 0, 0, 0, 0 };
        for (int d = 0; d < depth; ++d) { double u = rng_u(&r); hp_cnt[u < 0.4 ? 1 : (u < 0.8 ? 2 : 3)]++; }
        int row = 0;
        for (int hp = 1; hp <= 3; ++hp)
            for (int k = 0; k < hp_cnt[hp]; ++k, ++row) {
                int lo = 0, hi = L; /* covered span [lo,hi) */
                if (rng_u(&r) >= 0.98) {
                    int cut = 1 + rng_int(&r, L - 1);
                    if (rng_u(&r) < 0.5) lo = cut; else hi = cut;
                    /* the centre column is always covered (create_pileup_haplotype.py:145-149) */
                    if (lo > L / 2) lo = L / 2;
                    if (hi <= L / 2) hi = L / 2 + 1;
                }
                const int mapq = rng_u(&r) < 0.85 ? 60 : 20 + rng_int(&r, 40);
                for (int l = 0; l < L; ++l) {
                    int32_t base = 0, q = 0, mqv = 0, hv = 0;
                    if (l >= lo && l < hi) {
                        double u = rng_u(&r);
                        if (u < 0.02) base = -1;
                        else if (u < 0.12) base = 1 + rng_int(&r, 4);
                        else base = cons[l & 63];
                        hv = hp; mqv = mapq;
                        if (base > 0) {
                            double v = floor(15.0 + 6.0 * rng_normal(&r) + 0.5);
                            q = (int32_t)(v < 0 ? 0 : (v > 60 ? 60 : v));
                        }
                    }
                    s[row * L + l] = base; b[row * L + l] = q; m[row * L + l] = mqv; h[row * L + l] = hv;
                }
            }
        for (; row < D; ++row) /* write_to_bins.py:15-30 pads with -2 */
            for (int l = 0; l < L; ++l) { s[row * L + l] = -2; b[row * L + l] = -2; m[row * L + l] = -2; h[row * L + l] = -2; }
    }
    return 0;
}

/* ---- columns -> samtools-mpileup text (the input format of the reference's stage 1, make_predict_data.sh:117) ----------------------
 * One line per column: contig \t pos \t N \t depth \t bases \t 'I' x max(depth, 1) \n, depth = the reads of the column (bytes of
 * "ACGTNacgtn*#").  Two passes over 1024-column chunks (sizes, then text), OpenMP over the chunks.  Returns the bytes written, or
 * -(needed + 16) when cap is too small / out NULL. */
static int dec_len(int64_t v) { int n = 1; while (v >= 10) { v /= 10; ++n; } return n; }
static char* put_dec(char* p, int64_t v) { char t[24]; int n = 0; do { t[n++] = (char)('0' + v % 10); v /= 10; } while (v); while (n) *p++ = t[--n]; return p; }

int64_t nsnp_columns_to_mpileup_text(const char* contig, int64_t M, const int64_t* pos, const uint8_t* bases, const int64_t* col_off,
                                     char* out, int64_t cap)
{
    if (!contig || M < 0 || !pos || !bases || !col_off) return NSNP_HOST_EINVAL;
    const int64_t n_chunk = (M + CHUNK - 1) / CHUNK;
    int64_t* cb = (int64_t*)calloc((size_t)n_chunk + 1, sizeof(int64_t));
    if (!cb) return NSNP_HOST_ENOMEM;
    const size_t ln = strlen(contig);
    uint8_t is_read[256];
    memset(is_read, 0, sizeof is_read);
    for (const char* s = "ACGTNacgtn*#"; *s; ++s) is_read[(unsigned char)*s] = 1;
    #pragma omp parallel for num_threads(nsnp_host_threads()) schedule(dynamic, 16)
    for (int64_t ch = 0; ch < n_chunk; ++ch) {
        const int64_t c0 = ch * CHUNK, c1 = (c0 + CHUNK < M) ? c0 + CHUNK : M;
        int64_t n = 0;
        for (int64_t c = c0; c < c1; ++c) {
            int64_t d = 0;
            for (int64_t i = col_off[c]; i < col_off[c + 1]; ++i) d += is_read[bases[i]];
            n += (int64_t)ln + 1 + dec_len(pos[c]) + 3 + dec_len(d) + 1 + (col_off[c + 1] - col_off[c]) + 1 + (d > 1 ? d : 1) + 1;
        }
        cb[ch] = n;
    }
    int64_t total = 0;
    for (int64_t ch = 0; ch < n_chunk; ++ch) { const int64_t n = cb[ch]; cb[ch] = total; total += n; }
    if (!out || cap < total) { free(cb); return -(total + 16); }
    #pragma omp parallel for num_threads(nsnp_host_threads()) schedule(dynamic, 16)
    for (int64_t ch = 0; ch < n_chunk; ++ch) {
        const int64_t c0 = ch * CHUNK, c1 = (c0 + CHUNK < M) ? c0 + CHUNK : M;
        char* p = out + cb[ch];
        for (int64_t c = c0; c < c1; ++c) {
            int64_t d = 0;
            for (int64_t i = col_off[c]; i < col_off[c + 1]; ++i) d += is_read[bases[i]];
            memcpy(p, contig, ln); p += ln; *p++ = '\t';
            p = put_dec(p, pos[c]); *p++ = '\t'; *p++ = 'N'; *p++ = '\t';
            p = put_dec(p, d); *p++ = '\t';
            memcpy(p, bases + col_off[c], (size_t)(col_off[c + 1] - col_off[c])); p += col_off[c + 1] - col_off[c]; *p++ = '\t';
            const int64_t nq = d > 1 ? d : 1;
            memset(p, 'I', (size_t)nq); p += nq; *p++ = '\n';
        }
    }
    free(cb);
    return total;
}
