/*
 * oracle/pileup_encode_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE (see oracle.h).
 *
 * CPU restatement of the pileup encode stage of NanoSNP:
 *   E0  channel order / nt4 table   dna_sv_tensor/src/common/tensor.hpp:6-26,
 *                                   dna_sv_tensor/src/common/cpp_aux.cpp:85-102
 *   E1  TensorMaker::make_tensor    dna_sv_tensor/src/make_candidate_snp_tensor/tensor_maker.cpp:61-249
 *   E2  create_pileup_tensor        dna_sv_tensor/src/make_candidate_snp_tensor/main.cpp:113-312
 *   E3  make_predict_array          dna_sv_tensor/src/make_predict_data/main.cpp:76-127
 *
 * Written from the behaviour of those functions, in C with flat arrays instead of
 * std::map/std::string.  Pinned byte-for-byte against the reference programs themselves
 * (oracle/_ref) by tests/test_oracle_vs_ref.py.
 */
#include "oracle.h"

#include <ctype.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define MAX_INDEL 60 /* kMaxIndelSize, tensor_maker.cpp:5 */

/* cpp_aux.cpp:85-102: A/a C/c G/g T/t -> 0..3, '-' -> 5, everything else 4 */
static int nt4(int c)
{
    switch (c) {
    case 'A': case 'a': return 0;
    case 'C': case 'c': return 1;
    case 'G': case 'g': return 2;
    case 'T': case 't': return 3;
    case '-': return 5;
    default: return 4;
    }
}

/* tensor_maker.cpp:30-38 "ACGTNacgtn*#" */
static int is_normal(int c) { return c != 0 && strchr("ACGTNacgtn*#", c) != NULL; }
/* tensor_maker.cpp:40-46 "ACGTN*"; the NUL that std::string yields for an empty indel
 * sequence is not a member, so such an indel lands on the reverse strand */
static int is_fwd(int c) { return c != 0 && strchr("ACGTN*", c) != NULL; }

/* tensor_maker.cpp:48-58 */
static int base_channel(int c)
{
    switch (c) {
    case 'A': return ORC_CH_A; case 'C': return ORC_CH_C;
    case 'G': return ORC_CH_G; case 'T': return ORC_CH_T;
    case 'a': return ORC_CH_a; case 'c': return ORC_CH_c;
    case 'g': return ORC_CH_g; case 't': return ORC_CH_t;
    case '*': return ORC_CH_STAR; case '#': return ORC_CH_POUND;
    default: return ORC_NCH;
    }
}

/* One counted indel: the `len` allele characters the column still holds behind the digits, the declared length `adv`.
 * tensor_maker.cpp:101 appends `advance` characters starting at c_str() + base_idx WHATEVER the column still holds: an allele the
 * end of the column cuts short (adv > len; only the last construct of a column can be) gets a key of adv characters = the visible
 * ones, the string's terminating NUL, and adv - len - 1 bytes of whatever follows the buffer.  The NUL alone makes that key differ
 * from every complete allele of the column (whose characters are all inside the string), so a cut allele is ALWAYS an allele of its
 * own - never merged with a complete "+2AC" that shows the same visible characters - and in std::map order it comes right behind the
 * key that equals its visible part.  (Pinned by the reference's binaries: tests/test_oracle_golden.py, the fuzzed contig.) */
typedef struct { const char* p; int len; int adv; int count; } key_t_;

static int key_cmp(const void* a, const void* b)
{
    const key_t_* x = (const key_t_*)a; const key_t_* y = (const key_t_*)b;
    int n = x->len < y->len ? x->len : y->len;
    int c = memcmp(x->p, y->p, (size_t)n);
    if (c) return c;
    if (x->len != y->len) return (x->len > y->len) - (x->len < y->len);
    const int xc = x->adv > x->len, yc = y->adv > y->len;            /* cut short: behind the complete allele with the same characters */
    return xc - yc;
}

/* owned-string variant for alt_dict keys */
typedef struct { char* s; int len; int count; } alt_t_;
static int alt_cmp(const void* a, const void* b)
{
    const alt_t_* x = (const alt_t_*)a; const alt_t_* y = (const alt_t_*)b;
    int n = x->len < y->len ? x->len : y->len;
    int c = memcmp(x->s, y->s, (size_t)n);
    if (c) return c;
    return (x->len > y->len) - (x->len < y->len);
}

size_t orc_make_tensor(const char* bases, int64_t len, char ref_raw,
                       const char* next_ref, int64_t n_next,
                       double snp_min_af, double indel_min_af,
                       orc_column_t* out, char* alt_info, size_t alt_cap)
{
    /* tensor_maker.hpp:37-44 + tensor_maker.cpp:77-78: non-ACGT reference -> 'A'/'a', then upper */
    char chr_base = (nt4((unsigned char)ref_raw) < 4) ? ref_raw
                    : (isupper((unsigned char)ref_raw) ? 'A' : 'a');
    chr_base = (char)toupper((unsigned char)chr_base);

    int32_t* t = out->counts;
    memset(t, 0, sizeof(int32_t) * ORC_NCH);

    /* --- pass 1: scan the column string (tensor_maker.cpp:83-114) ------------------- */
    int32_t single[256];
    memset(single, 0, sizeof(single));
    key_t_* indel = NULL; int n_indel = 0, cap_indel = 0;
    int64_t i = 0;
    while (i < len) {
        unsigned char b = (unsigned char)bases[i];
        if (b == '+' || b == '-') {
            ++i;
            long advance = 0;
            while (i < len && isdigit((unsigned char)bases[i])) {
                advance = advance * 10 + (bases[i] - '0');
                ++i;
            }
            if (advance <= MAX_INDEL) {
                if (n_indel == cap_indel) {
                    cap_indel = cap_indel ? cap_indel * 2 : 16;
                    indel = (key_t_*)realloc(indel, sizeof(key_t_) * (size_t)cap_indel);
                }
                /* key = sign + the advance characters that follow the digits; i-1 is the
                 * last digit (or the sign itself when there are no digits), so the key is
                 * represented as (sign char, pointer to sequence, length) */
                int64_t avail = len - i; if (avail < 0) avail = 0;
                int l = (int)(advance < avail ? advance : avail);
                indel[n_indel].p = bases + i;
                indel[n_indel].len = l;
                indel[n_indel].adv = (int)advance;
                indel[n_indel].count = (b == '+') ? 1 : -1; /* sign kept in count for now */
                ++n_indel;
            }
            i += advance - 1;
        } else if (is_normal(b)) {
            ++single[b];
        } else if (b == '^') {
            ++i; /* the next char is a mapping quality */
        }
        /* '$' and anything else: nothing */
        ++i;
    }

    /* --- distinct indel alleles, in std::map<string> order: '+' (0x2B) keys sort before
     *     '-' (0x2D) keys, then memcmp on the sequence (tensor_maker.cpp:127-169) -------- */
    key_t_* ins = (key_t_*)malloc(sizeof(key_t_) * (size_t)(n_indel + 1));
    key_t_* del = (key_t_*)malloc(sizeof(key_t_) * (size_t)(n_indel + 1));
    int n_ins = 0, n_del = 0;
    for (int k = 0; k < n_indel; ++k) {
        if (indel[k].count > 0) ins[n_ins++] = indel[k]; else del[n_del++] = indel[k];
    }
    qsort(ins, (size_t)n_ins, sizeof(key_t_), key_cmp);
    qsort(del, (size_t)n_del, sizeof(key_t_), key_cmp);
    /* collapse runs of equal keys */
    int u = 0;
    for (int k = 0; k < n_ins; ++k) {
        if (u && key_cmp(&ins[u - 1], &ins[k]) == 0) ins[u - 1].count++;
        else { ins[u] = ins[k]; ins[u].count = 1; ++u; }
    }
    n_ins = u; u = 0;
    for (int k = 0; k < n_del; ++k) {
        if (u && key_cmp(&del[u - 1], &del[k]) == 0) del[u - 1].count++;
        else { del[u] = del[k]; del[u].count = 1; ++u; }
    }
    n_del = u;

    int max_ins_0 = 0, max_ins_1 = 0, max_del_0 = 0, max_del_1 = 0;
    int depth = 0, max_del_length = 0;
    int pile_I = 0, pile_D = 0;          /* pileup_dict["I"], ["D"] */
    int pile_base[4] = {0, 0, 0, 0};     /* pileup_dict["A".."T"] (upper+lower merged) */
    int have_I = 0, have_D = 0, have_base[4] = {0, 0, 0, 0};

    alt_t_* alts = NULL; int n_alt = 0;
    if (alt_info) alts = (alt_t_*)calloc((size_t)(n_ins + n_del + 8), sizeof(alt_t_));

    for (int k = 0; k < n_ins; ++k) {
        int cnt = ins[k].count;
        have_I = 1; pile_I += cnt;
        int first = ins[k].len ? (unsigned char)ins[k].p[0] : 0;
        if (is_fwd(first)) { t[ORC_CH_I] += cnt; if (cnt > max_ins_0) max_ins_0 = cnt; }
        else               { t[ORC_CH_i] += cnt; if (cnt > max_ins_1) max_ins_1 = cnt; }
        if (alts) { /* "I" + chr_base + upper(seq)  (tensor_maker.cpp:131-136) */
            alt_t_* a = &alts[n_alt++];
            const int cut_short = ins[k].adv > ins[k].len;          /* the key then holds the column string's NUL behind the visible part */
            a->len = 2 + ins[k].len + cut_short; a->s = (char*)malloc((size_t)a->len + 1);
            a->s[0] = 'I'; a->s[1] = chr_base;
            for (int q = 0; q < ins[k].len; ++q) a->s[2 + q] = (char)toupper((unsigned char)ins[k].p[q]);
            if (cut_short) a->s[2 + ins[k].len] = '\0';
            a->count = cnt;
        }
    }
    for (int k = 0; k < n_del; ++k) {
        int cnt = del[k].count;
        have_D = 1; pile_D += cnt;
        const int dlen = del[k].adv;                                /* key.size() - 1: the DECLARED length, also for an allele cut short */
        if (dlen > max_del_length) max_del_length = dlen;
        int first = del[k].len ? (unsigned char)del[k].p[0] : 0;
        if (is_fwd(first)) { t[ORC_CH_D] += cnt; if (cnt > max_del_0) max_del_0 = cnt; }
        else               { t[ORC_CH_d] += cnt; if (cnt > max_del_1) max_del_1 = cnt; }
        if (alts) { /* "D" + the reference bases that follow the position (tensor_maker.cpp:150-155).  Past the end of the
                     * contig the reference reads the NUL that terminates its sequence buffer; the key keeps that byte
                     * (it sorts before every base) and the C-string output of alt_info stops there (see below). */
            alt_t_* a = &alts[n_alt++];
            a->len = 1 + dlen; a->s = (char*)malloc((size_t)a->len + 1);
            a->s[0] = 'D';
            for (int q = 0; q < dlen; ++q)
                a->s[1 + q] = (next_ref && q < n_next) ? next_ref[q] : '\0';
            a->count = cnt;
        }
    }
    /* single-character keys in map order; only their per-key effects matter
     * (tensor_maker.cpp:170-187).  N/n are counted nowhere. */
    static const char singles[] = "#*ACGTacgt";
    for (const char* s = singles; *s; ++s) {
        int c = (unsigned char)*s, cnt = single[c];
        if (!cnt) continue;
        if (nt4(c) < 4) {
            int bi = nt4(c);
            have_base[bi] = 1; pile_base[bi] += cnt;
            depth += cnt;
            t[base_channel(c)] += cnt;
            if (alts && toupper(c) != chr_base) { /* "X" + upper(base), merged over case */
                char up = (char)toupper(c); int found = 0;
                for (int q = 0; q < n_alt; ++q)
                    if (alts[q].len == 2 && alts[q].s[0] == 'X' && alts[q].s[1] == up) { alts[q].count += cnt; found = 1; break; }
                if (!found) {
                    alt_t_* a = &alts[n_alt++];
                    a->len = 2; a->s = (char*)malloc(3); a->s[0] = 'X'; a->s[1] = up; a->count = cnt;
                }
            }
        } else if (c == '*') { t[ORC_CH_STAR] += cnt; depth += cnt; }
        else if (c == '#')   { t[ORC_CH_POUND] += cnt; depth += cnt; }
    }

    t[ORC_CH_I1] = max_ins_0; t[ORC_CH_i1] = max_ins_1;   /* tensor_maker.cpp:190-193 */
    t[ORC_CH_D1] = max_del_0; t[ORC_CH_d1] = max_del_1;

    /* --- allele list sorted by count, ties in map order A C D G I T (tensor_maker.cpp:195-228;
     *     libstdc++ std::sort on <= 16 items is an insertion sort, which is stable) -------- */
    struct { char key; int count; } lst[6]; int nl = 0;
    if (have_base[0]) { lst[nl].key = 'A'; lst[nl++].count = pile_base[0]; }
    if (have_base[1]) { lst[nl].key = 'C'; lst[nl++].count = pile_base[1]; }
    if (have_D)       { lst[nl].key = 'D'; lst[nl++].count = pile_D; }
    if (have_base[2]) { lst[nl].key = 'G'; lst[nl++].count = pile_base[2]; }
    if (have_I)       { lst[nl].key = 'I'; lst[nl++].count = pile_I; }
    if (have_base[3]) { lst[nl].key = 'T'; lst[nl++].count = pile_base[3]; }
    for (int a = 1; a < nl; ++a) { /* stable insertion sort, descending count */
        char k = lst[a].key; int c = lst[a].count; int b = a - 1;
        while (b >= 0 && lst[b].count < c) { lst[b + 1].key = lst[b].key; lst[b + 1].count = lst[b].count; --b; }
        lst[b + 1].key = k; lst[b + 1].count = c;
    }
    int denominator = depth ? depth : 1;
    int pass_snp = 0, pass_indel = 0;
    int pass_af = nl && lst[0].key != chr_base;
    for (int a = 0; a < nl; ++a) {
        if (lst[a].key == chr_base) continue;
        if (lst[a].key == 'I' || lst[a].key == 'D') {
            pass_indel = pass_indel || (1.0 * lst[a].count / denominator >= indel_min_af);
            continue;
        }
        pass_snp = pass_snp || (1.0 * lst[a].count / denominator >= snp_min_af);
    }
    double af = (nl > 1) ? (1.0 * lst[1].count / denominator) : 0.0;
    if (nl && lst[0].key != chr_base) af = 1.0 * lst[0].count / denominator;

    /* --- reference-base channels overwritten with minus the strand's ACGT total
     *     (tensor_maker.cpp:230-246) ------------------------------------------------------- */
    int up = t[ORC_CH_A] + t[ORC_CH_C] + t[ORC_CH_G] + t[ORC_CH_T];
    t[base_channel(chr_base)] = -up;
    int lo = t[ORC_CH_a] + t[ORC_CH_c] + t[ORC_CH_g] + t[ORC_CH_t];
    t[base_channel(tolower((unsigned char)chr_base))] = -lo;

    out->depth = depth;
    out->max_del_length = max_del_length;
    out->af = af;
    out->pass_snp_af = (uint8_t)pass_snp;
    out->pass_indel_af = (uint8_t)pass_indel;
    out->pass_af = (uint8_t)(pass_af || pass_snp || pass_indel);

    /* --- alt_info text ------------------------------------------------------------------- */
    size_t need = 0;
    if (alts) {
        /* merge equal keys (e.g. "+1a" and "+1A" both become "I?A"), then map order */
        qsort(alts, (size_t)n_alt, sizeof(alt_t_), alt_cmp);
        int m = 0;
        for (int k = 0; k < n_alt; ++k) {
            if (m && alt_cmp(&alts[m - 1], &alts[k]) == 0) { alts[m - 1].count += alts[k].count; free(alts[k].s); }
            else alts[m++] = alts[k];
        }
        n_alt = m;
        size_t w = 0;
        int cut = 0;                                   /* a NUL inside a key ends the reference's "%s" output */
        for (int k = 0; k < n_alt; ++k) {
            if (!cut) {
                char num[16]; int nn = snprintf(num, sizeof num, "%d", alts[k].count);
                const void* z = memchr(alts[k].s, 0, (size_t)alts[k].len);
                const size_t klen = z ? (size_t)((const char*)z - alts[k].s) : (size_t)alts[k].len;
                const size_t piece = z ? klen : klen + 1 + (size_t)nn + 1;
                if (w + piece < alt_cap) {
                    memcpy(alt_info + w, alts[k].s, klen); w += klen;
                    if (!z) {
                        alt_info[w++] = ' ';
                        memcpy(alt_info + w, num, (size_t)nn); w += (size_t)nn;
                        alt_info[w++] = ' ';
                    }
                }
                need += piece;
                if (z) cut = 1;
            }
            free(alts[k].s);
        }
        if (alt_cap) alt_info[w < alt_cap ? w : alt_cap - 1] = 0;
        free(alts);
    }
    free(indel); free(ins); free(del);
    return need;
}

void orc_encode_columns2(const uint8_t* bases, const int64_t* col_off, const uint8_t* ref,
                         int64_t M, double snp_min_af, double indel_min_af, int min_coverage,
                         int32_t* counts, int32_t* depth, uint8_t* flags);

void orc_encode_columns(const uint8_t* bases, const int64_t* col_off, const uint8_t* ref,
                        int64_t M, double min_af, int min_coverage,
                        int32_t* counts, int32_t* depth, uint8_t* flags)
{
    orc_encode_columns2(bases, col_off, ref, M, min_af, min_af, min_coverage, counts, depth, flags);
}

/* the reference program's two thresholds apart (main.cpp:79-88: -snp_min_af, -indel_min_af) */
void orc_encode_columns2(const uint8_t* bases, const int64_t* col_off, const uint8_t* ref,
                         int64_t M, double snp_min_af, double indel_min_af, int min_coverage,
                         int32_t* counts, int32_t* depth, uint8_t* flags)
{
    #pragma omp parallel for schedule(dynamic, 1024)
    for (int64_t c = 0; c < M; ++c) {
        orc_column_t col;
        orc_make_tensor((const char*)bases + col_off[c], col_off[c + 1] - col_off[c], (char)ref[c],
                        NULL, 0, snp_min_af, indel_min_af, &col, NULL, 0);
        memcpy(counts + c * ORC_NCH, col.counts, sizeof(int32_t) * ORC_NCH);
        depth[c] = col.depth;
        uint8_t f = 0;
        if (col.pass_af) f |= ORC_FLAG_PASS_AF;
        if (col.pass_snp_af) f |= ORC_FLAG_PASS_SNP;
        if (col.pass_indel_af) f |= ORC_FLAG_PASS_INDEL;
        /* main.cpp:163,195 */
        if (nt4(toupper(ref[c])) < 4 && col.pass_af && col.depth >= min_coverage) f |= ORC_FLAG_CANDIDATE;
        flags[c] = f;
    }
}

int64_t orc_select_sites(const int64_t* pos, const uint8_t* flags, int64_t M, int flank,
                         int64_t* center_idx, int64_t cap)
{
    /* Sequential restatement of the ring-buffer logic of main.cpp:174-217: run length since
     * the last position gap, FIFO of pending candidates, emission when the stream reaches
     * centre + flank, drop when fewer than 2*flank+1 columns of the run are filled. */
    const int W = 2 * flank + 1;
    int64_t n = 0, filled = 0, pre = -1;
    int64_t* pend = (int64_t*)malloc(sizeof(int64_t) * (size_t)(W + 2));
    int head = 0, cnt = 0;
    for (int64_t c = 0; c < M; ++c) {
        if (pre + 1 != pos[c]) { filled = 0; head = 0; cnt = 0; }
        pre = pos[c];
        if (flags[c] & ORC_FLAG_CANDIDATE) { pend[(head + cnt) % (W + 2)] = c; ++cnt; }
        ++filled;
        if (cnt > 0 && pos[c] - pos[pend[head]] == flank) {
            int64_t center = pend[head];
            head = (head + 1) % (W + 2); --cnt;
            if (filled < W) continue;
            if (n < cap) center_idx[n] = center;
            ++n;
        }
    }
    free(pend);
    return n;
}

void orc_gather_windows(const int32_t* counts, const int64_t* center_idx, int64_t N,
                        int flank, int32_t* x)
{
    const int W = 2 * flank + 1;
    for (int64_t n = 0; n < N; ++n)
        memcpy(x + n * W * ORC_NCH, counts + (center_idx[n] - flank) * ORC_NCH,
               sizeof(int32_t) * (size_t)(W * ORC_NCH));
}

/* ------------------------------------------------------------------------------------------
 * File-level restatement: <chr>.mpileup -> .pd
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    int64_t pos;
    int32_t counts[ORC_NCH];
    int32_t depth;
    char*   alt;     /* alt_info text of a pending candidate */
} ring_t_;

/* split_line(line, "\t") semantics: tokens are maximal runs of non-tab characters
 * (common/cpp_aux.cpp:43-59) */
static int split_tabs(char* line, char** tok, int max_tok)
{
    int n = 0; char* p = line;
    while (*p) {
        while (*p == '\t') ++p;
        if (!*p) break;
        if (n < max_tok) tok[n] = p;
        ++n;
        while (*p && *p != '\t') ++p;
        if (*p) { *p = 0; ++p; }
    }
    return n;
}

int64_t orc_mpileup_to_pd2(const char* mpileup_path, const char* chr_seq, int64_t chr_len,
                           double snp_min_af, double indel_min_af, int min_coverage, int flank, const char* pd_path);

int64_t orc_mpileup_to_pd(const char* mpileup_path, const char* chr_seq, int64_t chr_len,
                          double min_af, int min_coverage, int flank, const char* pd_path)
{
    return orc_mpileup_to_pd2(mpileup_path, chr_seq, chr_len, min_af, min_af, min_coverage, flank, pd_path);
}

int64_t orc_mpileup_to_pd2(const char* mpileup_path, const char* chr_seq, int64_t chr_len,
                           double snp_min_af, double indel_min_af, int min_coverage, int flank, const char* pd_path)
{
    FILE* in = fopen(mpileup_path, "r");
    if (!in) return -1;
    FILE* out = fopen(pd_path, "w");
    if (!out) { fclose(in); return -2; }
    const int W = 2 * flank + 1;
    int32_t (*ring)[ORC_NCH] = calloc((size_t)W, sizeof(*ring));
    int pos_offset = 0;
    int64_t pre = -1, filled = 0, n_sites = 0;
    /* pending candidates (FIFO): position, depth, alt text */
    int64_t* ppos = malloc(sizeof(int64_t) * (size_t)(W + 2));
    int32_t* pdepth = malloc(sizeof(int32_t) * (size_t)(W + 2));
    char** palt = calloc((size_t)(W + 2), sizeof(char*));
    int head = 0, cnt = 0;

    char* line = NULL; size_t cap = 0; ssize_t got;
    while ((got = getline(&line, &cap, in)) > 0) {
        while (got > 0 && (line[got - 1] == '\n' || line[got - 1] == '\r')) line[--got] = 0;
        if (got == 0) continue;
        char* tok[8];
        int nt = split_tabs(line, tok, 8);
        if (nt < 5) continue;
        const char* chr_name = tok[0];
        int64_t ref_off = atoll(tok[1]);
        if (ref_off < 1 || ref_off > chr_len) { n_sites = -3; break; }
        const char* bases = tok[4];
        char ref_base = (char)toupper((unsigned char)chr_seq[ref_off - 1]);

        if (pre + 1 != ref_off) { /* main.cpp:174-178 */
            filled = 0; pos_offset = 0;
            for (int k = 0; k < cnt; ++k) { free(palt[(head + k) % (W + 2)]); palt[(head + k) % (W + 2)] = NULL; }
            head = 0; cnt = 0;
        }
        pre = ref_off;

        orc_column_t col;
        size_t blen = strlen(bases);
        char small[256]; char* alt = small;
        size_t need = orc_make_tensor(bases, (int64_t)blen, chr_seq[ref_off - 1],
                                      chr_seq + ref_off, chr_len - ref_off, snp_min_af, indel_min_af,
                                      &col, small, sizeof small);
        if (nt4((unsigned char)ref_base) < 4 && col.pass_af && col.depth >= min_coverage) {
            if (need + 1 > sizeof small) {
                alt = malloc(need + 1);
                orc_make_tensor(bases, (int64_t)blen, chr_seq[ref_off - 1], chr_seq + ref_off,
                                chr_len - ref_off, snp_min_af, indel_min_af, &col, alt, need + 1);
            }
            int slot = (head + cnt) % (W + 2);
            ppos[slot] = ref_off; pdepth[slot] = col.depth;
            palt[slot] = strdup(alt);
            if (alt != small) free(alt);
            ++cnt;
        }
        memcpy(ring[pos_offset], col.counts, sizeof(col.counts));
        ++filled;
        pos_offset = (pos_offset + 1) % W;

        if (cnt > 0 && ref_off - ppos[head] == flank) { /* main.cpp:208-217 */
            int64_t center = ppos[head]; int32_t depth = pdepth[head]; char* atext = palt[head];
            palt[head] = NULL; head = (head + 1) % (W + 2); --cnt;
            if (filled < W) { free(atext); continue; }
            /* centre reference base must be ACGT after upper-casing
             * (make_predict_data/main.cpp:91-92; already implied by the candidate test) */
            /* .pd line: tensor \t chr:pos:REF33(upper) \t depth-alts(right-trimmed) */
            for (int i2 = pos_offset; i2 < W; ++i2) for (int j = 0; j < ORC_NCH; ++j) fprintf(out, "%d ", ring[i2][j]);
            for (int i2 = 0; i2 < pos_offset; ++i2) for (int j = 0; j < ORC_NCH; ++j) fprintf(out, "%d ", ring[i2][j]);
            fprintf(out, "\t%s:%lld:", chr_name, (long long)center);
            for (int64_t p = center - flank; p < center + flank + 1; ++p)
                fputc(toupper((unsigned char)chr_seq[p - 1]), out);
            /* alt_info = "depth-" + pairs, trailing whitespace stripped (make_predict_data/main.cpp:88) */
            size_t al = strlen(atext);
            while (al && isspace((unsigned char)atext[al - 1])) atext[--al] = 0;
            fprintf(out, "\t%d-%s\n", depth, atext);
            free(atext);
            ++n_sites;
        }
    }
    for (int k = 0; k < W + 2; ++k) free(palt[k]);
    free(line); free(ring); free(ppos); free(pdepth); free(palt);
    fclose(in);
    if (fclose(out) != 0) return -4;
    return n_sites;
}

/* ------------------------------------------------------------------------------------------
 * mpileup text in memory -> (position, column-5 token) per line, byte at a time.
 *
 * Restates how the reference reads a <chr>.mpileup:
 *   LineReader::getline   common/line_reader.cpp:95-127   a line ends at '\n' (or at the end of the text); the '\n' and ONE
 *                                                          '\r' in front of it are dropped; a '\r' elsewhere stays in the line
 *   split_line(.., "\t")  common/cpp_aux.cpp:43-59         tokens = maximal runs of non-tab characters
 *   main.cpp:162-172                                       ref_off = atoll(token 1), pileup_bases = token 4
 * A line with fewer than five tokens (an empty line included) makes the reference index a vector out of range (it aborts or
 * worse): reported here as -(line number + 1), nothing the callers may accept.
 * Returns the number of lines; beg/end are offsets into text.  Used as the checker of the device tokeniser
 * (nanosnp_amd/csrc/mpileup_tokenise.hip) and of the host one (nsnp_textio.c).
 * ---------------------------------------------------------------------------------------- */
int64_t orc_mpileup_tokenise(const char* text, int64_t len, int64_t cap, int64_t* pos, int64_t* beg, int64_t* end)
{
    int64_t m = 0, p = 0;
    while (p < len) {
        int64_t le = p;
        while (le < len && text[le] != '\n') ++le;
        const int64_t next = le < len ? le + 1 : len;
        if (le > p && text[le - 1] == '\r') --le;
        /* tokens of text[p, le) */
        int ntok = 0; int64_t q = p, t1 = -1, t1e = -1, t4 = -1, t4e = -1;
        while (q < le) {
            while (q < le && text[q] == '\t') ++q;
            if (q >= le) break;
            const int64_t s = q;
            while (q < le && text[q] != '\t') ++q;
            if (ntok == 1) { t1 = s; t1e = q; } else if (ntok == 4) { t4 = s; t4e = q; }
            ++ntok;
        }
        if (ntok < 5) return -(m + 1);
        if (m < cap) {
            /* atoll on the token: white space, one sign, digits (the token holds no tab / newline) */
            int64_t s = t1; uint64_t v = 0; int neg = 0;
            while (s < t1e && isspace((unsigned char)text[s])) ++s;
            if (s < t1e && (text[s] == '-' || text[s] == '+')) { neg = text[s] == '-'; ++s; }
            while (s < t1e && text[s] >= '0' && text[s] <= '9') { v = v * 10u + (uint64_t)(text[s] - '0'); ++s; }
            pos[m] = neg ? (int64_t)(0u - v) : (int64_t)v;
            beg[m] = t4; end[m] = t4e;
        }
        ++m;
        p = next;
    }
    return m;
}
