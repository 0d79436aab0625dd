// pileup_encode.hip -- pileup column encode, candidate-window selection and window gather.
//
// Replaces, for in-memory columns,
//   TensorMaker::make_tensor        dna_sv_tensor/src/make_candidate_snp_tensor/tensor_maker.cpp:61-249
//   candidate test / pending queue  dna_sv_tensor/src/make_candidate_snp_tensor/main.cpp:174-217
//   33-column window emission       dna_sv_tensor/src/make_candidate_snp_tensor/main.cpp:233-244
// The reference builds a std::map<string,int> per column; here one lane walks one column's bytes
// (the +n<seq>/-n<seq>/^q grammar is inherently sequential inside a column), 64 columns per
// wave, with the byte stream fetched 8 bytes at a time.  Distinct-allele maxima (channels
// I1/D1/i1/d1) use a small per-lane table in LDS keyed by a hash and verified byte-for-byte, with
// an exact quadratic rescan when a column has more distinct indel alleles than the table holds.
// Integer outputs are bit-exact with the reference; the AF tests use the same float64 division.
#include "nsnp_common.hpp"
#include <math.h>

namespace {

enum { CH_A = 0, CH_C, CH_G, CH_T, CH_I, CH_I1, CH_D, CH_D1, CH_STAR,
       CH_a, CH_c, CH_g, CH_t, CH_i, CH_i1, CH_d, CH_d1, CH_POUND, NCH };

constexpr int ENC_BLOCK = 256;
constexpr int MAX_INDEL = 60;     // kMaxIndelSize, tensor_maker.cpp:5

// class of a pileup byte: 0..9 = counted symbol (channel via CLS_CH), 10 = ignored,
// 11 '+', 12 '-', 13 '^'
__device__ __forceinline__ int byte_class(int b)
{
    switch (b) {
    case 'A': return 0; case 'C': return 1; case 'G': return 2; case 'T': return 3;
    case 'a': return 4; case 'c': return 5; case 'g': return 6; case 't': return 7;
    case '*': return 8; case '#': return 9;
    case '+': return 11; case '-': return 12; case '^': return 13;
    default: return 10;
    }
}
__device__ __forceinline__ bool is_fwd_char(int b)   // "ACGTN*", tensor_maker.cpp:40-46
{
    return b == 'A' || b == 'C' || b == 'G' || b == 'T' || b == 'N' || b == '*';
}
__device__ __forceinline__ int nt4(int b)            // cpp_aux.cpp:85-102 (only the <4 test is used)
{
    switch (b) {
    case 'A': case 'a': return 0; case 'C': case 'c': return 1;
    case 'G': case 'g': return 2; case 'T': case 't': return 3;
    default: return 4;
    }
}

// Iterator over the counted indels of a column (those with length <= 60), following exactly
// the scan of tensor_maker.cpp:83-114: '^' swallows the next byte; '+'/'-' read decimal digits,
// then skip `advance` bytes (advance == 0 re-examines the byte after the sign/digits).
template <typename P>
struct IndelIterT {
    P base; int64_t i, end;
    __device__ __forceinline__ bool next(int64_t& off, int& len, int& sign)
    {
        while (i < end) {
            const int b = base[i];
            if (b == '+' || b == '-') {
                ++i;
                long long adv = 0;
                while (i < end && base[i] >= '0' && base[i] <= '9') { adv = adv * 10 + (base[i] - '0'); ++i; }
                const int64_t avail = end - i;
                const int64_t l = adv < avail ? adv : avail;
                const int64_t o = i;
                i += adv;           // (advance-1) + the loop's ++; past-the-end is clamped by the while
                if (adv <= MAX_INDEL) { off = o; len = (int)l; sign = b; return true; }
            } else if (b == '^') {
                i += 2;
            } else {
                ++i;
            }
        }
        return false;
    }
};
typedef IndelIterT<const uint8_t*> IndelIter;

// four small counters that are only ever indexed through compile-time selects (a runtime-indexed register array is
// placed in scratch memory: a dependent round trip to HBM-backed memory per access)
struct Quad {
    int32_t v0, v1, v2, v3;
    __device__ __forceinline__ void inc(int k) { v0 += k == 0; v1 += k == 1; v2 += k == 2; v3 += k == 3; }
    __device__ __forceinline__ void raise(int k, int32_t x)
    {
        v0 = (k == 0 && x > v0) ? x : v0; v1 = (k == 1 && x > v1) ? x : v1;
        v2 = (k == 2 && x > v2) ? x : v2; v3 = (k == 3 && x > v3) ? x : v3;
    }
};

template <typename P>
__device__ __forceinline__ Quad rescan_maxima(P base, int64_t begin, int64_t end)
{
    // exact distinct-allele maxima by comparing every counted indel with every other one: O(k^2)
    Quad mx{0, 0, 0, 0};
    IndelIterT<P> a{base, begin, end};
    int64_t ao; int al, as;
    while (a.next(ao, al, as)) {
        const int kind = (as == '-' ? 2 : 0) + (al > 0 && is_fwd_char(base[ao]) ? 0 : 1);
        IndelIterT<P> b2{base, begin, end};
        int64_t bo; int bl, bs; int same = 0;
        while (b2.next(bo, bl, bs)) {
            if (bs != as || bl != al) continue;
            bool eq = true;
            for (int k = 0; k < al; ++k) if (base[ao + k] != base[bo + k]) { eq = false; break; }
            same += eq;
        }
        mx.raise(kind, same);
    }
    return mx;
}

// symbol counters of one column in the order of byte_class(): A C G T a c g t * #
struct Counts10 { int32_t k0, k1, k2, k3, k4, k5, k6, k7, k8, k9; };

// ---- slow exact path for one column straight from global memory (no size limits) ---------------------
// Used for columns whose bytes do not fit the LDS stage of their wave.  O(k^2) in the number of indel reads k.
__device__ __forceinline__ void scan_column_global(const uint8_t* __restrict__ bases, int64_t begin, int64_t end,
                                                Counts10& cnt, Quad& tot, Quad& mx)
{
    Counts10 c{0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    Quad t{0, 0, 0, 0};
    for (int64_t i = begin; i < end;) {
        const int b = bases[i];
        const int cls = byte_class(b);
        if (cls < 10) {
            c.k0 += cls == 0; c.k1 += cls == 1; c.k2 += cls == 2; c.k3 += cls == 3; c.k4 += cls == 4;
            c.k5 += cls == 5; c.k6 += cls == 6; c.k7 += cls == 7; c.k8 += cls == 8; c.k9 += cls == 9;
            ++i;
        } else if (cls == 11 || cls == 12) {
            ++i;
            long long adv = 0;
            while (i < end && bases[i] >= '0' && bases[i] <= '9') { adv = adv * 10 + (bases[i] - '0'); ++i; }
            if (adv <= MAX_INDEL) {
                const int64_t avail = end - i;
                const int len = (int)(adv < avail ? adv : avail);
                t.inc((b == '-' ? 2 : 0) + (len > 0 && is_fwd_char(bases[i]) ? 0 : 1));
            }
            i += adv;
        } else if (cls == 13) i += 2;
        else ++i;
    }
    cnt = c; tot = t;
    mx = rescan_maxima(bases, begin, end);
}

// The candidate tests compare (double)count / (double)depth with min_af (tensor_maker.cpp:195-228: float64 division).  For
// integers count, depth < 2^31 the correctly rounded quotient is >= a double a exactly when count / depth >= the midpoint tau
// between a and its predecessor (the quotient can never BE that midpoint: tau has an odd 54- or 55-bit mantissa, which a
// denominator below 2^31 cannot produce), i.e. count * 2^k >= T * depth with T = 2 M - 1 (4 M - 1 when a is a power of two),
// a = M * 2^(e-53), k = 54 - e (55 - e).  One 128-bit comparison per allele instead of six float64 divisions per column, same
// bits.  mode 1 / 2: always / never (a <= 0 or so small that any positive count passes; NaN or a beyond every quotient).
struct AfThreshold { uint64_t t; int k; int mode; };

static AfThreshold make_af_threshold(double a)
{
    AfThreshold r{0, 0, 0};
    if (a != a) { r.mode = 2; return r; }
    if (a <= 0.0) { r.mode = 1; return r; }
    int e = 0;
    const double m = frexp(a, &e);                     // a = m 2^e, m in [0.5, 1)
    if (!(m > 0.0) || e > 40) { r.mode = 2; return r; }          // infinity, or a > 2^39 > any quotient of 31-bit integers
    const uint64_t M53 = (uint64_t)ldexp(m, 53);       // 2^52 <= M53 < 2^53 (subnormal a: fewer bits, still exact)
    const bool pow2 = M53 == (1ull << 52);
    r.t = pow2 ? 4 * M53 - 1 : 2 * M53 - 1;
    r.k = (pow2 ? 55 : 54) - e;
    if (r.k >= 88) { r.mode = 1; r.t = 0; r.k = 0; }   // count 2^k >= 2^88 > T depth for every count >= 1
    if (r.k < 0) { r.mode = 2; }                       // (e > 54: unreachable behind the e > 40 test)
    return r;
}

// ---- main kernel ----------------------------------------------------------------------------------------
// A wave stages the bytes of its 64 columns (one contiguous range) into LDS with 16-byte loads, every lane
// then walks its own column out of LDS four bytes per read.  Indels met on the way are only RECORDED
// (offset, length, sign) in a per-lane list; the distinct-allele maxima are worked out afterwards from that
// list, byte-exact, so the main scan stays short and the lanes of a wave diverge as little as the grammar
// allows.
constexpr int ENC_WAVES = ENC_BLOCK / 64;
#ifndef NSNP_ENC_STAGE
#define NSNP_ENC_STAGE 6144
#endif
#ifndef NSNP_ENC_KLIST
#define NSNP_ENC_KLIST 12
#endif
constexpr int STAGE_BYTES = NSNP_ENC_STAGE;          // per wave; 64 columns at 60x average ~4.4 KB
constexpr int KLIST = NSNP_ENC_KLIST;                  // recorded indels per column before the slow path takes over

__global__ __launch_bounds__(ENC_BLOCK) void k_encode_columns(
    const uint8_t* __restrict__ bases, const int64_t* __restrict__ col_off, const uint8_t* __restrict__ ref,
    int64_t M, AfThreshold af, int min_cov, int32_t* __restrict__ counts, int32_t* __restrict__ depth_out,
    uint8_t* __restrict__ flags)
{
    __shared__ __attribute__((aligned(16))) uint8_t stage_b[ENC_WAVES][STAGE_BYTES];
    __shared__ uint32_t ilist[ENC_WAVES][KLIST][64];      // off (16) | len (8) | sign (8)
    // byte -> 64-bit row: a one in the 6-bit field of its symbol class (A C G T a c g t * # = fields 0..9, bits 0..59), bit 63
    // set for the three bytes that open a construct (+ - ^).  Every other byte maps to zero.
    __shared__ uint2 ctab[256];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    {
        const int cls = byte_class(tid);      // ENC_BLOCK == 256: one table row per thread
        unsigned long long r = 0;
        if (cls < 10) r = 1ull << (6 * cls);
        else if (cls >= 11) r = 1ull << 63;
        ctab[tid] = uint2{(uint32_t)r, (uint32_t)(r >> 32)};
    }
    __syncthreads();
    const int64_t wave_col0 = ((int64_t)blockIdx.x * ENC_WAVES + wave) * 64;
    if (wave_col0 >= M) return;                                   // whole wave idle (no later block barrier)
    const int64_t c = wave_col0 + lane;
    const bool live = c < M;
    const int n_live = (int)(M - wave_col0 < 64 ? M - wave_col0 : 64);
    // one offset load per lane; a column's end is its right neighbour's begin (the last live lane takes the wave's end offset)
    const int64_t wave_end = col_off[wave_col0 + n_live];
    int64_t begin = live ? col_off[c] : wave_end;
    int64_t end = __shfl_down(begin, 1);
    if (lane == n_live - 1) end = wave_end;
    if (!live) { begin = 0; end = 0; }
    const int64_t total = col_off[M];

    uint8_t* st = stage_b[wave];
    const uint32_t* st32 = reinterpret_cast<const uint32_t*>(st);
    int32_t cnt[10];                     // (every index below is a compile-time constant: registers, not scratch)
#pragma unroll
    for (int k = 0; k < 10; ++k) cnt[k] = 0;
    Quad tot{0, 0, 0, 0};                // I, i, D, d   (kind = (sign=='-')*2 + reverse)
    Quad mx{0, 0, 0, 0};
    bool slow = false;

    // The wave's 64 columns are one contiguous byte range; it is staged into LDS in as few sub-batches of
    // consecutive columns as the stage buffer allows (one at 30x, one or two at 60x).  A single column
    // longer than the buffer takes the global-memory path.
    for (int first = 0; first < n_live;) {
        const int64_t b0 = __shfl(begin, first);
        const uintptr_t a_first = (uintptr_t)(bases + b0) & ~(uintptr_t)15;      // 16-byte aligned global address
        const int mis = (int)((uintptr_t)(bases + b0) - a_first);
        const bool fits = live && lane >= first && (end - b0) + mis + 16 <= STAGE_BYTES;
        const unsigned long long fm = __ballot(fits) >> first;                   // lanes first.. that fit, from bit 0
        int n_fit = (~fm) ? __builtin_ctzll(~fm) : 64;                           // leading run of fitting columns
        if (n_fit > n_live - first) n_fit = n_live - first;
        if (n_fit == 0) {                                                        // column `first` alone exceeds the stage
            if (lane == first) slow = true;
            first += 1;
            continue;
        }
        const int last = first + n_fit;                                          // sub-batch = lanes [first, last)
        const int64_t b1 = __shfl(end, last - 1);
        {
            const uintptr_t a_end = (uintptr_t)(bases + b1);
            const uintptr_t a_total = (uintptr_t)(bases + total);
            for (uintptr_t a = a_first + (uintptr_t)lane * 16; a < a_end; a += 64 * 16) {
                uint4 v;
                if (a + 16 <= a_total && a >= (uintptr_t)bases) v = *reinterpret_cast<const uint4*>(a);
                else {                                                           // buffer edge: byte loads
                    uint32_t wv[4] = {0u, 0u, 0u, 0u};
                    for (int k = 0; k < 16; ++k) {
                        const uintptr_t q = a + k;
                        if (q >= (uintptr_t)bases && q < a_total) wv[k >> 2] |= (uint32_t)(*reinterpret_cast<const uint8_t*>(q)) << (8 * (k & 3));
                    }
                    v = uint4{wv[0], wv[1], wv[2], wv[3]};
                }
                *reinterpret_cast<uint4*>(st + (a - a_first)) = v;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();

        {
            // every lane runs the passes (they hold wave-level operations); lanes outside the sub-batch have an empty range
            const bool act = lane >= first && lane < last;
            // LDS byte index of global offset g: (g - b0) + mis.
            const int lbeg = act ? (int)(begin - b0) + mis : mis, lend = act ? (int)(end - b0) + mis : mis;
            int n_list = 0;
            // ---- pass 1: EVERY byte of the column is counted through the table, no grammar state at all; the positions of
            // the construct openers (+ - ^) are collected in one bit mask per 60-byte chunk (15 words: a 6-bit field cannot
            // overflow inside a chunk).  The uniform trip counts are the longest column of the sub-batch.
            int len_here = lend - (lbeg & ~3);
            int maxlen = len_here;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) { const int v = __shfl_xor(maxlen, o); maxlen = v > maxlen ? v : maxlen; }
            const int n_chunks = (maxlen + 59) / 60;                              // same in every lane of the sub-batch
            unsigned long long sm0 = 0, sm1 = 0, sm2 = 0;                         // opener positions of chunks 0, 1, 2
            bool bad = act && n_chunks > 3 && lend - (lbeg & ~3) > 180;           // longer than 180 bytes: exact path below
            auto flush6 = [&](unsigned long long a) {
#pragma unroll
                for (int k = 0; k < 10; ++k) cnt[k] += (int)((a >> (6 * k)) & 63);
            };
            {
                const int n_ch = n_chunks > 3 ? 3 : n_chunks;                     // (lanes longer than that are `bad` already)
                for (int ch = 0; ch < n_ch; ++ch) {
                    const int cb = (lbeg & ~3) + 60 * ch;
                    int nw = (maxlen - 60 * ch + 3) >> 2; nw = nw > 15 ? 15 : nw;
                    if (bad) nw = 0;                                              // (per lane: its counts are redone below)
                    unsigned long long acc = 0, sm = 0;
                    for (int wi = 0; wi < nw; ++wi) {
                        const int p0 = cb + 4 * wi;
                        uint32_t w = st32[(p0 < STAGE_BYTES - 4 ? p0 : STAGE_BYTES - 4) >> 2];
                        // bytes outside [lbeg, lend) become 0xff (an all-zero row)
                        const int lo = lbeg - p0, hi = lend - p0;
                        const uint32_t mlo = lo <= 0 ? 0xffffffffu : (lo >= 4 ? 0u : 0xffffffffu << (8 * lo));
                        const uint32_t mhi = hi >= 4 ? 0xffffffffu : (hi <= 0 ? 0u : ~(0xffffffffu << (8 * hi)));
                        w |= ~(mlo & mhi);
                        uint32_t s4 = 0;
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const uint2 row = ctab[(w >> (8 * k)) & 0xffu];
                            acc += ((unsigned long long)(row.y & 0x7fffffffu) << 32) | row.x;
                            s4 |= (row.y >> 31) << k;
                        }
                        sm |= (unsigned long long)s4 << (4 * wi);
                    }
                    flush6(acc);
                    if (ch == 0) sm0 = sm; else if (ch == 1) sm1 = sm; else sm2 = sm;
                }
                // ---- pass 2: the openers, in order.  An opener that lies inside the bytes an earlier construct consumes is not an
                // opener at all (and what it would have skipped is): such a column is re-scanned exactly below.  For the others the
                // bytes they skip (the byte after ^, the allele behind +n / -n) are taken out of the counts again.
                unsigned long long cur = sm0, neg = 0;
                int cbase = lbeg & ~3, ci = 0, consumed = lbeg, nneg = 0;
                while (true) {
                    if (cur == 0 && ci < 2) { cur = ci == 0 ? sm1 : sm2; ++ci; cbase += 60; }
                    const bool have = (cur | (ci == 0 ? (sm1 | sm2) : (ci == 1 ? sm2 : 0ull))) != 0 && !bad;
                    if (__ballot(have) == 0ull) break;
                    if (have && cur != 0) {
                        const int p = cbase + __builtin_ctzll(cur);
                        cur &= cur - 1;
                        if (p < consumed) bad = true;
                        else {
                            // the opener and the eight bytes behind it from three aligned words (no byte loops: digits and the first
                            // four skipped bytes are taken out of this window, which covers every 1- to 3-digit indel of up to 5 bases)
                            const int b = st[p];
                            const int w0 = (p + 1) >> 2;
                            const uint32_t a0 = st32[w0], a1 = st32[w0 + 1], a2 = st32[w0 + 2];
                            const int sh = (p + 1) & 3;
                            const uint32_t wlo = __builtin_amdgcn_alignbyte(a1, a0, sh), whi = __builtin_amdgcn_alignbyte(a2, a1, sh);
                            const unsigned long long win = ((unsigned long long)whi << 32) | wlo;         // byte i = st[p + 1 + i]
                            const int avail1 = lend - (p + 1);                                              // bytes of the column behind the opener
                            const bool caret = b == '^';
                            const uint32_t d0 = (wlo & 0xffu) - '0', d1 = ((wlo >> 8) & 0xffu) - '0', d2 = ((wlo >> 16) & 0xffu) - '0', d3 = (wlo >> 24) - '0';
                            const bool k0 = !caret && avail1 > 0 && d0 < 10u, k1 = k0 && avail1 > 1 && d1 < 10u, k2 = k1 && avail1 > 2 && d2 < 10u;
                            if (k2 && avail1 > 3 && d3 < 10u) bad = true;                                   // four digits and more: exact path
                            const int L = k2 ? 3 : (k1 ? 2 : (k0 ? 1 : 0));
                            const int adv = caret ? 1 : (k2 ? (int)(d0 * 100 + d1 * 10 + d2) : (k1 ? (int)(d0 * 10 + d1) : (k0 ? (int)d0 : 0)));
                            const int q = p + 1 + L;
                            const int avail = lend - q;
                            const int nskip = adv < avail ? adv : (avail > 0 ? avail : 0);
                            if (!caret && adv <= MAX_INDEL) {
                                if (n_list < KLIST) ilist[wave][n_list][lane] = (uint32_t)q | ((uint32_t)nskip << 16) | ((uint32_t)b << 24);
                                ++n_list;
                            }
                            consumed = q + adv;
                            // skipped bytes were counted in pass 1: the first four out of the window ...
#pragma unroll
                            for (int i = 0; i < 4; ++i) {
                                if (i < nskip) {
                                    const uint2 row = ctab[(uint32_t)(win >> (8 * (L + i))) & 0xffu];
                                    neg += ((unsigned long long)(row.y & 0x7fffffffu) << 32) | row.x;
                                }
                            }
                            nneg += nskip < 4 ? nskip : 4;
                            // ... longer alleles byte by byte (rare; only the lanes that have one)
                            for (int k = q + 4; k < q + nskip; ++k) {
                                const uint2 row = ctab[st[k]];
                                neg += ((unsigned long long)(row.y & 0x7fffffffu) << 32) | row.x;
                                if (++nneg >= 59) {
#pragma unroll
                                    for (int f = 0; f < 10; ++f) cnt[f] -= (int)((neg >> (6 * f)) & 63);
                                    neg = 0; nneg = 0;
                                }
                            }
                            if (nneg >= 59) {
#pragma unroll
                                for (int f = 0; f < 10; ++f) cnt[f] -= (int)((neg >> (6 * f)) & 63);
                                neg = 0; nneg = 0;
                            }
                        }
                    }
                }
#pragma unroll
                for (int f = 0; f < 10; ++f) cnt[f] -= (int)((neg >> (6 * f)) & 63);
            }
            if (bad) {
                // exact byte-at-a-time scan of this column out of LDS (tensor_maker.cpp:83-114 verbatim in structure): openers inside
                // skipped bytes, columns beyond 180 bytes.  Divergent, but only the lanes that need it run it.
#pragma unroll
                for (int k = 0; k < 10; ++k) cnt[k] = 0;
                n_list = 0;
                for (int i = lbeg; i < lend;) {
                    const int b = st[i];
                    const int cls = byte_class(b);
                    if (cls < 10) {
#pragma unroll
                        for (int k = 0; k < 10; ++k) cnt[k] += cls == k;
                        ++i;
                    } else if (cls == 11 || cls == 12) {
                        ++i;
                        int adv = 0;
                        while (i < lend && st[i] >= '0' && st[i] <= '9') { adv = adv > 100000 ? adv : adv * 10 + (st[i] - '0'); ++i; }
                        if (adv <= MAX_INDEL) {
                            const int avail = lend - i;
                            const int len = adv < avail ? adv : avail;
                            if (n_list < KLIST) ilist[wave][n_list][lane] = (uint32_t)i | ((uint32_t)len << 16) | ((uint32_t)b << 24);
                            ++n_list;
                        }
                        i = adv < lend - i ? i + adv : lend;
                    } else if (cls == 13) i += 2;
                    else ++i;
                }
            }
            // distinct-allele maxima from the recorded list: m(e) = #{f <= e equal to e}
            const int n_rec = n_list < KLIST ? n_list : KLIST;
            for (int e = 0; e < n_rec; ++e) {
                const uint32_t ve = ilist[wave][e][lane];
                const int oe = ve & 0xffff, le = (ve >> 16) & 0xff, se = ve >> 24;
                const int kind = (se == '-' ? 2 : 0) + (le > 0 && is_fwd_char(st[oe]) ? 0 : 1);
                tot.inc(kind);
                int same = 1;
                for (int f = 0; f < e; ++f) {
                    const uint32_t vf = ilist[wave][f][lane];
                    if ((vf >> 16) != (ve >> 16)) continue;               // length and sign
                    const int of = vf & 0xffff;
                    bool eq = true;
                    for (int k = 0; k < le; ++k) if (st[oe + k] != st[of + k]) { eq = false; break; }
                    same += eq;
                }
                mx.raise(kind, same);
            }
            if (n_list > KLIST) {
                // more indel reads than the list holds: totals and maxima again, exactly, from the staged bytes
                tot = Quad{0, 0, 0, 0};
                IndelIterT<const uint8_t*> it{st, lbeg, lend};
                int64_t io; int il, is;
                while (it.next(io, il, is)) tot.inc((is == '-' ? 2 : 0) + (il > 0 && is_fwd_char(st[io]) ? 0 : 1));
                mx = rescan_maxima((const uint8_t*)st, (int64_t)lbeg, (int64_t)lend);
            }
        }
        __builtin_amdgcn_wave_barrier();            // the stage buffer is reused by the next sub-batch
        first = last;
    }
    if (slow) {          // results come back in plain structs: no array of this kernel ever has its address taken
        Counts10 c2; Quad t2, m2;
        scan_column_global(bases, begin, end, c2, t2, m2);
        cnt[0] = c2.k0; cnt[1] = c2.k1; cnt[2] = c2.k2; cnt[3] = c2.k3; cnt[4] = c2.k4;
        cnt[5] = c2.k5; cnt[6] = c2.k6; cnt[7] = c2.k7; cnt[8] = c2.k8; cnt[9] = c2.k9;
        tot = t2; mx = m2;
    }

    // ---- assemble the 18 channels, flags (tensor_maker.cpp:127-248) ------------------------------
    int32_t t[NCH];
    t[CH_A] = cnt[0]; t[CH_C] = cnt[1]; t[CH_G] = cnt[2]; t[CH_T] = cnt[3];
    t[CH_a] = cnt[4]; t[CH_c] = cnt[5]; t[CH_g] = cnt[6]; t[CH_t] = cnt[7];
    t[CH_STAR] = cnt[8]; t[CH_POUND] = cnt[9];
    t[CH_I] = tot.v0; t[CH_i] = tot.v1; t[CH_D] = tot.v2; t[CH_d] = tot.v3;
    t[CH_I1] = mx.v0; t[CH_i1] = mx.v1; t[CH_D1] = mx.v2; t[CH_d1] = mx.v3;
    const int up = cnt[0] + cnt[1] + cnt[2] + cnt[3];
    const int lo = cnt[4] + cnt[5] + cnt[6] + cnt[7];
    const int depth = up + lo + cnt[8] + cnt[9];
    const int refraw = live ? ref[c] : 'A';
    const int rb = nt4(refraw);
    const int chr_idx = rb < 4 ? rb : 0;                      // non-ACGT reference counts as 'A'
    // allele list in std::map order A C D G I T; the first maximum is what a stable sort puts first
    const int lc[6] = {cnt[0] + cnt[4], cnt[1] + cnt[5], tot.v2 + tot.v3, cnt[2] + cnt[6], tot.v0 + tot.v1, cnt[3] + cnt[7]};
    const int lk[6] = {0, 1, 5, 2, 4, 3};                     // 0..3 = base index, 4 = I, 5 = D
    int top = -1, topc = 0;
#pragma unroll
    for (int k = 0; k < 6; ++k) if (lc[k] > topc) { topc = lc[k]; top = lk[k]; }
    const uint32_t den = (uint32_t)(depth ? depth : 1);
    const unsigned __int128 rhs = (unsigned __int128)af.t * den;                 // T x depth, < 2^87
    bool pass_snp = false, pass_indel = false;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        if (lc[k] <= 0 || lk[k] == chr_idx) continue;
        // ((double)count / depth) >= min_af, decided exactly in integers (AfThreshold below)
        const bool ok = af.mode == 0 ? (((unsigned __int128)(uint32_t)lc[k] << af.k) >= rhs) : af.mode == 1;
        if (lk[k] >= 4) pass_indel = pass_indel || ok; else pass_snp = pass_snp || ok;
    }
    const bool pass_af = (top >= 0 && top != chr_idx) || pass_snp || pass_indel;
    const int up_ch[4] = {CH_A, CH_C, CH_G, CH_T}, lo_ch[4] = {CH_a, CH_c, CH_g, CH_t};
#pragma unroll
    for (int k = 0; k < 4; ++k) if (k == chr_idx) { t[up_ch[k]] = -up; t[lo_ch[k]] = -lo; }

    // ---- coalesced write-out through the (now free) stage buffer ---------------------------------
    __builtin_amdgcn_wave_barrier();
    int32_t* out_stage = reinterpret_cast<int32_t*>(st);         // 64 * 18 * 4 = 4608 B <= STAGE_BYTES
#pragma unroll
    for (int k = 0; k < NCH; k += 2) *reinterpret_cast<int2*>(out_stage + lane * NCH + k) = int2{t[k], t[k + 1]};
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const int n_valid = n_live * NCH;
    int32_t* __restrict__ dst = counts + wave_col0 * NCH;
    if ((reinterpret_cast<uintptr_t>(dst) & 15) == 0) {          // a wave's 4608 output bytes as 16-byte stores (4.5 per lane)
        const int n16 = n_valid >> 2;
        for (int k = lane; k < n16; k += 64) reinterpret_cast<int4*>(dst)[k] = reinterpret_cast<const int4*>(out_stage)[k];
        for (int k = (n16 << 2) + lane; k < n_valid; k += 64) dst[k] = out_stage[k];
    } else {
        for (int k = lane; k < n_valid; k += 64) dst[k] = out_stage[k];
    }
    if (live) {
        depth_out[c] = depth;
        uint8_t f = 0;
        if (pass_af) f |= NSNP_FLAG_PASS_AF;
        if (pass_snp) f |= NSNP_FLAG_PASS_SNP;
        if (pass_indel) f |= NSNP_FLAG_PASS_INDEL;
        if (rb < 4 && pass_af && depth >= min_cov) f |= NSNP_FLAG_CANDIDATE;
        flags[c] = f;
    }
}

// ---- site selection: candidate && 33 consecutive positions around it -------------------------------
__device__ __forceinline__ bool site_ok(const int64_t* pos, const uint8_t* flags, int64_t M, int64_t c)
{
    if (!(flags[c] & NSNP_FLAG_CANDIDATE)) return false;
    if (c < PCENTER || c + PCENTER >= M) return false;
    return pos[c + PCENTER] - pos[c] == PCENTER && pos[c] - pos[c - PCENTER] == PCENTER;
}

constexpr int SEL_BLOCK = 256, SEL_PER_THREAD = 8, SEL_TILE = SEL_BLOCK * SEL_PER_THREAD;

__global__ __launch_bounds__(SEL_BLOCK) void k_select_count(const int64_t* pos, const uint8_t* flags, int64_t M,
                                                             int64_t* block_cnt)
{
    __shared__ int wsum[SEL_BLOCK / 64];
    const int64_t base = (int64_t)blockIdx.x * SEL_TILE + threadIdx.x * SEL_PER_THREAD;
    int n = 0;
    for (int k = 0; k < SEL_PER_THREAD; ++k) { const int64_t c = base + k; if (c < M && site_ok(pos, flags, M, c)) ++n; }
    for (int o = 32; o > 0; o >>= 1) n += __shfl_down(n, o);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = n;
    __syncthreads();
    if (threadIdx.x == 0) { int s = 0; for (int w = 0; w < SEL_BLOCK / 64; ++w) s += wsum[w]; block_cnt[blockIdx.x] = s; }
}

// exclusive scan of the block counts by one block; also writes the total
__global__ __launch_bounds__(1024) void k_select_scan(int64_t* block_cnt, int64_t n_blocks, int64_t* total)
{
    __shared__ int64_t part[1024];
    const int tid = threadIdx.x;
    const int64_t per = NSNP_CDIV(n_blocks, 1024);
    const int64_t b0 = tid * per, b1 = (b0 + per < n_blocks) ? b0 + per : n_blocks;
    int64_t s = 0;
    for (int64_t b = b0; b < b1; ++b) s += block_cnt[b];
    part[tid] = s;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {     // Hillis-Steele inclusive scan
        int64_t v = tid >= o ? part[tid - o] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    int64_t run = tid ? part[tid - 1] : 0;
    for (int64_t b = b0; b < b1; ++b) { const int64_t v = block_cnt[b]; block_cnt[b] = run; run += v; }
    if (tid == 1023) *total = part[1023];
}

__global__ __launch_bounds__(SEL_BLOCK) void k_select_scatter(const int64_t* pos, const uint8_t* flags, int64_t M,
                                                               const int64_t* block_off, int64_t* center_idx, int64_t cap)
{
    __shared__ int wsum[SEL_BLOCK / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t base = (int64_t)blockIdx.x * SEL_TILE + tid * SEL_PER_THREAD;
    bool ok[SEL_PER_THREAD]; int n = 0;
#pragma unroll
    for (int k = 0; k < SEL_PER_THREAD; ++k) { const int64_t c = base + k; ok[k] = c < M && site_ok(pos, flags, M, c); n += ok[k]; }
    int incl = n;
    for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(incl, o); if (lane >= o) incl += v; }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    int woff = 0;
    for (int w = 0; w < wave; ++w) woff += wsum[w];
    int64_t o = block_off[blockIdx.x] + woff + incl - n;
#pragma unroll
    for (int k = 0; k < SEL_PER_THREAD; ++k) if (ok[k]) { if (o < cap) center_idx[o] = base + k; ++o; }
}

__global__ void k_gather_windows(const int32_t* __restrict__ counts, const int64_t* __restrict__ center_idx, int64_t N,
                                 int32_t* __restrict__ x)
{
    constexpr int W = PW * PC;   // 594 ints per site
    const int64_t total = N * W;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t n = e / W; const int r = (int)(e - n * W);
        x[e] = counts[(center_idx[n] - PCENTER) * PC + r];
    }
}

}  // namespace

extern "C" int nsnp_pileup_encode_columns(nsnp_ctx* ctx, const uint8_t* bases, const int64_t* col_off,
                                          const uint8_t* ref, int64_t M, double min_af, int min_coverage,
                                          int32_t* counts, int32_t* depth, uint8_t* flags, void* stream)
{
    if (!ctx || M < 0 || (M > 0 && (!bases || !col_off || !ref || !counts || !depth || !flags))) return NSNP_EINVAL;
    if (M == 0) return NSNP_OK;
    const unsigned grid = (unsigned)NSNP_CDIV(M, ENC_BLOCK);
    ScopedKernelTimer tm(ctx, NSNP_K_ENCODE, (hipStream_t)stream);
    hipLaunchKernelGGL(k_encode_columns, dim3(grid), dim3(ENC_BLOCK), 0, (hipStream_t)stream,
                       bases, col_off, ref, M, make_af_threshold(min_af), min_coverage, counts, depth, flags);
    NSNP_HIP(ctx, hipGetLastError());
    return NSNP_OK;
}

extern "C" int nsnp_pileup_select_sites(nsnp_ctx* ctx, const int64_t* pos, const uint8_t* flags, int64_t M,
                                        int64_t* center_idx, int64_t cap, int64_t* n_sites, void* stream)
{
    if (!ctx || M < 0 || cap < 0 || !n_sites || (M > 0 && (!pos || !flags)) || (cap > 0 && !center_idx)) return NSNP_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    if (M == 0) { NSNP_HIP(ctx, hipMemsetAsync(n_sites, 0, sizeof(int64_t), s)); return NSNP_OK; }
    const int64_t n_blocks = NSNP_CDIV(M, SEL_TILE);
    const size_t need = (size_t)n_blocks * sizeof(int64_t);
    if (ctx->sel_tmp_bytes < need) {
        // grows only when a larger M than ever before arrives (synchronous; size it with a warm-up call)
        NSNP_HIP(ctx, hipStreamSynchronize(s));
        if (ctx->sel_tmp) (void)hipFree(ctx->sel_tmp);
        ctx->sel_tmp = nullptr; ctx->sel_tmp_bytes = 0;
        NSNP_HIP(ctx, hipMalloc((void**)&ctx->sel_tmp, need));
        ctx->sel_tmp_bytes = need;
    }
    hipLaunchKernelGGL(k_select_count, dim3((unsigned)n_blocks), dim3(SEL_BLOCK), 0, s, pos, flags, M, ctx->sel_tmp);
    hipLaunchKernelGGL(k_select_scan, dim3(1), dim3(1024), 0, s, ctx->sel_tmp, n_blocks, n_sites);
    hipLaunchKernelGGL(k_select_scatter, dim3((unsigned)n_blocks), dim3(SEL_BLOCK), 0, s, pos, flags, M,
                       (const int64_t*)ctx->sel_tmp, center_idx, cap);
    NSNP_HIP(ctx, hipGetLastError());
    return NSNP_OK;
}

extern "C" int nsnp_pileup_gather_windows(nsnp_ctx* ctx, const int32_t* counts, const int64_t* center_idx,
                                          int64_t N, int32_t* x, void* stream)
{
    if (!ctx || N < 0 || (N > 0 && (!counts || !center_idx || !x))) return NSNP_EINVAL;
    if (N == 0) return NSNP_OK;
    int64_t blocks = NSNP_CDIV(N * PW * PC, 256);
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(k_gather_windows, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, counts, center_idx, N, x);
    NSNP_HIP(ctx, hipGetLastError());
    return NSNP_OK;
}
