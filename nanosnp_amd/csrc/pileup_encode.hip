// pileup_encode.hip -- pileup column encode, candidate-window selection and window gather.
//
// Replaces, for in-memory columns,
//   TensorMaker::make_tensor        dna_sv_tensor/src/make_candidate_snp_tensor/tensor_maker.cpp:61-249
//   candidate test / pending queue  dna_sv_tensor/src/make_candidate_snp_tensor/main.cpp:174-217
//   33-column window emission       dna_sv_tensor/src/make_candidate_snp_tensor/main.cpp:233-244
// The reference builds a std::map<string,int> per column; here one lane owns one column, 64 columns per wave: every byte is counted
// through a table, then the +n<seq> / -n<seq> / ^q constructs (whose grammar is sequential inside a column) are decoded one per
// lane over the whole wave and their skipped bytes taken out again; distinct-allele maxima (channels I1/D1/i1/d1) come from comparing
// every counted indel with the earlier ones of its column, byte-exact; columns the fast path cannot take are re-scanned exactly.
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
// An allele the end of the column cuts short (advance > the bytes left; only the last construct of a column can be) is an allele
// of its own in the reference: tensor_maker.cpp:101 appends `advance` characters from c_str() whatever the string still holds, so
// the key carries the string's terminating NUL and can equal no complete allele.  Such an indel comes back with bit 8 of `sign`
// set, which keeps it apart from every other in the comparisons below.
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
                if (adv <= MAX_INDEL) { off = o; len = (int)l; sign = b | (adv > avail ? 0x100 : 0); return true; }
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
        const int kind = ((as & 0xff) == '-' ? 2 : 0) + (al > 0 && is_fwd_char(base[ao]) ? 0 : 1);
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

// ---- exact path for one column, byte at a time (tensor_maker.cpp:83-114 in structure), from global memory (columns whose bytes do
// not fit the LDS stage of their wave) or from the staged bytes (columns the fast path hands back).  O(k^2) in the indel reads k.
template <typename P>
__device__ __forceinline__ void scan_column_exact(P bases, int64_t begin, int64_t end, Counts10& cnt, Quad& tot, Quad& mx)
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
            while (i < end && bases[i] >= '0' && bases[i] <= '9') { adv = adv > 100000000 ? adv : adv * 10 + (bases[i] - '0'); ++i; }
            if (adv <= MAX_INDEL) {
                const int64_t avail = end - i;
                const int len = (int)(adv < avail ? adv : avail);
                t.inc((b == '-' ? 2 : 0) + (len > 0 && is_fwd_char(bases[i]) ? 0 : 1));
            }
            i = adv < end - i ? i + adv : end;
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

// The same test as a table for the depths the fast path can produce (a column of at most 253 bytes holds at most 253 reads):
// af_min[d] = the smallest count c >= 1 with (double)c / (double)d >= min_af, 0xffff when no count <= d passes.  Built on the host
// with the exact 128-bit comparison above, once per call; a column then tests its six alleles with one LDS read and six compares
// instead of six 128-bit shifts and compares (110 of the kernel's 1,350 vector instructions per wave).
struct AfTable { uint32_t w[128]; };        // 256 x uint16, passed by value

static AfTable make_af_table(const AfThreshold& af)
{
    AfTable t;
    uint16_t m[256];
    for (int d = 0; d < 256; ++d) {
        m[d] = 0xffff;
        if (af.mode == 1) { m[d] = 1; continue; }
        if (af.mode == 2) continue;
        const unsigned __int128 rhs = (unsigned __int128)af.t * (uint32_t)(d ? d : 1);
        int lo = 1, hi = (d ? d : 1) + 1;                         // the test is monotone in c: smallest passing c by bisection, hi = none
        while (lo < hi) {
            const int c = (lo + hi) >> 1;
            if ((((unsigned __int128)(uint32_t)c) << af.k) >= rhs) hi = c; else lo = c + 1;
        }
        if (lo <= (d ? d : 1)) m[d] = (uint16_t)lo;
    }
    for (int i = 0; i < 128; ++i) t.w[i] = (uint32_t)m[2 * i] | ((uint32_t)m[2 * i + 1] << 16);
    return t;
}

// ---- main kernel ----------------------------------------------------------------------------------------
// One lane per column, 64 columns per wave, their bytes (one contiguous range) staged into LDS with 16-byte loads.  With its inputs
// and outputs served from cache the kernel still takes 82 % of its time: it is bound by vector-instruction
// issue and the LDS round trips between its phases at four waves per SIMD, so it is built around the instruction count per column:
//   pass 1   EVERY byte of the column is counted through a 16-byte table row (three words of 8-bit class counters and the flag
//            of the construct openers + - ^), whole words at a time: 5 vector instructions per byte, no grammar state, no flushes
//            (the fast path covers columns of up to 253 bytes), the opener flags gathered in one bit mask per 32 bytes;
//   openers  compacted over the wave (DPP prefix sum) and decoded ONE PER LANE whatever column they belong to: digits, the bytes
//            they skip taken out of the column's counts again (LDS atomics on the same packed counters), the record of a counted
//            indel (the counted ones compacted to the front of the list); an opener among the bytes a construct consumes (the
//            table rows of the skipped bytes tell), a four-digit length, a column beyond the fast path: that column is
//            re-scanned by the exact path;
//   indels   every counted indel finds its multiplicity among the earlier ones of its column, four records per trip (length, sign
//            and the first four allele bytes in one compare, longer alleles byte by byte); totals and maxima by kind reach the column through LDS
//            atomics.
constexpr int ENC_WAVES = ENC_BLOCK / 64;
#ifndef NSNP_ENC_STAGE
#define NSNP_ENC_STAGE 5120
#endif
#ifndef NSNP_ENC_ECAP
#define NSNP_ENC_ECAP 208
#endif
constexpr int STAGE_BYTES = NSNP_ENC_STAGE;          // per wave; 64 columns at 60x average ~4.4 KB
constexpr int ENC_ECAP = NSNP_ENC_ECAP;              // opener entries of one segment of a wave's columns (more: further segments); 64 columns at 60x
                                                     // hold 153 on average, 199 at most (generator G2): one segment; the kernel's LDS stays at four blocks per CU
#ifndef NSNP_ENC_P3STEP
#define NSNP_ENC_P3STEP 4
#endif
constexpr int P3_STEP = NSNP_ENC_P3STEP;             // records one trip of the multiplicity walk reads
constexpr int ENC_NBLK = 8;                          // 32-byte blocks of a column the fast path covers (8-bit counters: at most 253 bytes)
static_assert(STAGE_BYTES >= 64 * NCH * 4, "the stage buffer doubles as the 64 x 18 output transposition buffer");
static_assert(STAGE_BYTES <= 8192, "staged positions are kept in 13 bits");
static_assert(4 * (ENC_WAVES * (STAGE_BYTES + (ENC_ECAP + P3_STEP) * 8 + 64 * 32) + 4096 + 2 * 512) <= 160 * 1024, "four blocks per CU (LDS)");

// inclusive prefix sum over the 64 lanes of a wave with DPP row shifts and row broadcasts (no LDS)
__device__ __forceinline__ int wave_scan_incl(int v)
{
    int x = v;
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, true);      // row_shr:1
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, true);      // row_shr:2
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, true);      // row_shr:4
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, true);      // row_shr:8
    x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, true);      // row_bcast:15 into rows 1 and 3
    x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, true);      // row_bcast:31 into rows 2 and 3
    return x;
}

#ifndef NSNP_ENC_MINW
#define NSNP_ENC_MINW 4
#endif
__global__ __launch_bounds__(ENC_BLOCK, NSNP_ENC_MINW) void k_encode_columns(
    const uint8_t* __restrict__ bases, const int64_t* __restrict__ col_off, const uint8_t* __restrict__ ref,
    int64_t M, AfThreshold af, const AfTable aft, AfThreshold afi, const AfTable afti, int min_cov, int32_t* __restrict__ counts,
    int32_t* __restrict__ depth_out,
    uint8_t* __restrict__ flags)
{
    __shared__ uint32_t af_min[128];          // 256 x uint16: smallest passing count by depth (make_af_table): the SNP threshold
    __shared__ uint32_t afi_min[128];         // the same for the indel threshold (-indel_min_af; the same table when the two are equal)
    __shared__ __attribute__((aligned(16))) uint8_t stage_b[ENC_WAVES][STAGE_BYTES];
    // opener j of the segment sits behind P3_STEP end records (bit 31: no column; the walk of an indel over the earlier ones of its
    // column ends there): { position | column << 16, end of the column }, written by the column.  The lanes that decode the openers compact the
    // counted indels to the front, in order: { q | length << 13 | minus << 19 | fwd << 20 | column << 25, the first four allele
    // bytes (zero beyond the allele's length) }
    __shared__ uint2 ents[ENC_WAVES][ENC_ECAP + P3_STEP];
    // per column: [0..2] skipped bytes by class (the table's three counter words; bit 31 of [2]: hand the column to the exact path),
    // [3] indel reads by kind (8-bit fields),
    // [4..7] largest multiplicity of one allele by kind
    // Stored WORD-MAJOR ([word][column]): the lanes that decode openers reach the accumulators of OTHER columns through LDS atomics,
    // and with eight consecutive words per column (32 B) the word w of every column sat in one of four banks - an atomic instruction
    // over 50 distinct owner columns took 8 to 16 LDS cycles.  Now distinct columns are distinct banks (two columns 32 apart share one).
    __shared__ uint32_t colacc[ENC_WAVES][8][64];
    // byte -> x: A C G T, y: a c g t, z: * # (8-bit counters) | "ACGTN*" << 24, w: 1 for the construct openers + - ^
    __shared__ uint4 tab[256];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    {
        const int cls = byte_class(tid);      // ENC_BLOCK == 256: one table row per thread
        uint4 r{0u, 0u, 0u, 0u};
        if (cls < 4) r.x = 1u << (8 * cls);
        else if (cls < 8) r.y = 1u << (8 * (cls - 4));
        else if (cls < 10) r.z = 1u << (8 * (cls - 8));
        else if (cls >= 11) r.w = 1u;
        if (is_fwd_char(tid)) r.z |= 1u << 24;
        // row index = byte ^ ((byte >> 2) & 12): bits 4-5 of the byte folded into bits 2-3 of the row, so that the symbols a column
        // is made of fall into different 16-byte bank slots of a ds_read_b128 lane group - the eight letters (A a C c G g T t: slots
        // 1 9 3 11 7 15 0 8), '.' ',' of real pileups (6, 4), and the construct bytes + - ^ $ 1 2 3 mostly beside them: 1.41 LDS cycles
        // per lane group on generator G2's bytes, 1.09 on '.'/','-dominated ones, against 1.57 / 1.24 with only the case bit folded
        tab[tid ^ ((tid >> 2) & 12)] = r;
        if (tid < 128) { af_min[tid] = aft.w[tid]; afi_min[tid] = afti.w[tid]; }
        if (tid < ENC_WAVES * P3_STEP) ents[tid / P3_STEP][tid % P3_STEP] = uint2{0x80000000u, 0u};
    }
    __syncthreads();
    const int64_t wave_col0 = ((int64_t)blockIdx.x * ENC_WAVES + wave) * 64;
    if (wave_col0 >= M) return;                                   // whole wave idle (no later block barrier)
    const int64_t c = wave_col0 + lane;
    const bool live = c < M;
    const int n_live = (int)(M - wave_col0 < 64 ? M - wave_col0 : 64);
    const int64_t* __restrict__ woff = col_off + wave_col0;       // (uniform: scalar base + lane offset addressing)
    // three independent loads, issued back to back (lanes past the last column read the wave's end offset / its last reference byte)
    const int64_t wave_end = woff[n_live];
    const int64_t begin64 = woff[lane < n_live ? lane : n_live];
    const int refraw = (ref + wave_col0)[lane < n_live ? lane : n_live - 1];      // (used at the very end; in flight meanwhile)
    const int64_t total = col_off[M];
    // everything below works on 32-bit offsets relative to the wave's first byte; a wave whose columns span 1 GB or more has
    // them saturate, which sends the affected columns to the global-memory path (begin64 / end64 again)
    const int64_t wave_b0 = __builtin_amdgcn_readfirstlane((int)(begin64 & 0xffffffff)) |
                            ((int64_t)__builtin_amdgcn_readfirstlane((int)(begin64 >> 32)) << 32);
    constexpr int REL_SAT = 1 << 30;
    const int64_t rb64 = begin64 - wave_b0;
    int rbeg = (int)(rb64 < REL_SAT ? rb64 : REL_SAT);
    int rend = __shfl_down(rbeg, 1);
    {
        const int64_t re64 = wave_end - wave_b0;
        if (lane == n_live - 1) rend = (int)(re64 < REL_SAT ? re64 : REL_SAT);
    }
    if (!live) { rbeg = 0; rend = 0; }
    const uint8_t* __restrict__ wbase = bases + wave_b0;          // uniform

    uint8_t* st = stage_b[wave];
    const uint32_t* st32 = reinterpret_cast<const uint32_t*>(st);
    uint2* ent = ents[wave] + P3_STEP;                // opener j at ent[j]; ent[-1 .. -P3_STEP]: end records
    uint32_t (*const cacc)[64] = colacc[wave];           // cacc[word][column]
    int32_t cnt[10];
#pragma unroll
    for (int k = 0; k < 10; ++k) cnt[k] = 0;
    Quad tot{0, 0, 0, 0};                // I, i, D, d   (kind = (sign=='-')*2 + reverse)
    Quad mx{0, 0, 0, 0};
    bool slow = false;

    for (int first = 0; first < n_live;) {
        const int b0 = __builtin_amdgcn_readlane(rbeg, first);                    // uniform
        const int mis = (int)((uintptr_t)(wbase + b0) & 15);
        const uint8_t* __restrict__ src = wbase + (b0 - mis);                     // 16-byte aligned, uniform
        const bool fits = live && lane >= first && b0 < REL_SAT && rend <= (STAGE_BYTES - 16 - mis) + b0 && rend >= b0;
        const unsigned long long fm = __ballot(fits) >> first;
        int n_fit = (~fm) ? __builtin_ctzll(~fm) : 64;
        if (n_fit > n_live - first) n_fit = n_live - first;
        if (n_fit == 0) {
            if (lane == first) slow = true;
            first += 1;
            continue;
        }
        const int last = first + n_fit;
        const int span = __builtin_amdgcn_readlane(rend, last - 1) - b0 + mis;    // bytes to stage from src, uniform
        {
            // 16-byte pieces; the ones that stick out of the buffer (only ever the first and last pieces of the whole input) byte-wise
            const bool inside = (uintptr_t)src >= (uintptr_t)bases && (uintptr_t)src + (((uintptr_t)span + 15) & ~(uintptr_t)15) <= (uintptr_t)(bases + total);
            if (inside) {
                // all of a lane's pieces in flight together (one memory round trip per sub-batch, not one per kilobyte)
                constexpr int NP = (STAGE_BYTES + 1023) / 1024;
                const int o_last = ((span - 1) >> 4) << 4;                         // the last piece (span >= 1: a fitting column ends here)
                uint4 pc[NP];
#pragma unroll
                for (int i = 0; i < NP; ++i) { const int o = (lane + 64 * i) * 16; pc[i] = *reinterpret_cast<const uint4*>(src + (o < span ? o : o_last)); }
#pragma unroll
                for (int i = 0; i < NP; ++i) { const int o = (lane + 64 * i) * 16; if (o < span) *reinterpret_cast<uint4*>(st + o) = pc[i]; }
            } else {
                for (int o = lane * 16; o < span; o += 64 * 16) {
                    uint32_t wv[4] = {0u, 0u, 0u, 0u};
                    for (int k = 0; k < 16; ++k) {
                        const uint8_t* q = src + o + k;
                        if (q >= bases && q < bases + total) wv[k >> 2] |= (uint32_t)(*q) << (8 * (k & 3));
                    }
                    *reinterpret_cast<uint4*>(st + o) = uint4{wv[0], wv[1], wv[2], wv[3]};
                }
            }
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) cacc[k][lane] = 0u;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();

        {
            const bool act = lane >= first && lane < last;
            const int lbeg = act ? rbeg - b0 + mis : mis, lend = act ? rend - b0 + mis : mis;
            // ---- pass 1: every byte counted through the table; the opener flags land in one bit mask per 32-byte block ----
            const int wb = lbeg & ~3;                                              // the column's first word
            const int tr = act ? ((lend - 1 - wb) >> 2) : -1;                      // its last word (relative); -1: none
            bool bad = tr >= 8 * ENC_NBLK || lend - lbeg > 253;                     // beyond the fast path (word span, or a class count an 8-bit field cannot hold): exact path
            const int trp = bad ? -1 : tr;
            const uint32_t hm = (1u << (8 * (lbeg & 3))) - 1u;                     // bytes of word 0 in front of the column
            const int rem = lend - (wb + 4 * tr);                                  // bytes of the column in its last word, 1..4
            const uint32_t tm = rem >= 4 ? 0u : 0xffffffffu << (8 * (rem & 3));
            const uint32_t* cw = st32 + (wb >> 2);
            uint32_t ax = 0, ay = 0, az = 0;
            uint32_t sm[ENC_NBLK];
#pragma unroll
            for (int b = 0; b < ENC_NBLK; ++b) sm[b] = 0u;
#pragma unroll
            for (int b = 0; b < ENC_NBLK; ++b) {
                if (__ballot(trp >= 8 * b) == 0ull) break;
                uint32_t smb = 0u;
                // a rolled loop of two words per trip (unrolled, the scheduler hoists every load of the block and the kernel needs 450
                // registers); straight-line inside: words behind the column's last one become 0xffffffff (four zero rows)
#pragma nounroll
                for (int r = 0; r < 8; r += 2) {
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const int R = 8 * b + r + h;
                        uint32_t w = cw[R];
                        w |= (R == 0) ? hm : 0u;
                        w |= R < trp ? 0u : (R == trp ? tm : 0xffffffffu);
                        w ^= (w >> 2) & 0x0c0c0c0cu;                       // the table's row permutation, four bytes at once
                        const uint4 r0 = tab[w & 0xffu], r1 = tab[(w >> 8) & 0xffu], r2 = tab[(w >> 16) & 0xffu], r3 = tab[w >> 24];
                        ax += r0.x + r1.x; ax += r2.x + r3.x;
                        ay += r0.y + r1.y; ay += r2.y + r3.y;
                        az += r0.z + r1.z; az += r2.z + r3.z;
                        const uint32_t f4 = (((r3.w << 1) | r2.w) << 2) | ((r1.w << 1) | r0.w);
                        smb |= f4 << (4 * (r + h));
                    }
                    if (__ballot(trp > 8 * b + r + 1) == 0ull) break;
                }
                sm[b] = smb;
            }
            // ---- the openers of the sub-batch, compacted over the wave ----
            int n_op = 0;
#pragma unroll
            for (int b = 0; b < ENC_NBLK; ++b) n_op += __builtin_popcount(sm[b]);
            if (n_op > ENC_ECAP) { bad = true; n_op = 0; }
            const int incl = wave_scan_incl(n_op);
            const int excl = incl - n_op;
            const int total_ops = __builtin_amdgcn_readlane(incl, 63);
            for (int sfirst = first; total_ops > 0 && sfirst < last;) {
                // segment = the longest run of columns from sfirst whose openers fit the entry list
                const int sbase = __shfl(excl, sfirst);
                const bool sfit = act && lane >= sfirst && incl - sbase <= ENC_ECAP;
                const unsigned long long sfm = __ballot(sfit) >> sfirst;
                int n_seg = (~sfm) ? __builtin_ctzll(~sfm) : 64;
                if (n_seg > last - sfirst) n_seg = last - sfirst;
                const int slast = sfirst + n_seg;                                  // n_seg >= 1: every column holds <= ECAP openers
                const int T = __shfl(incl, slast - 1) - sbase;
                const bool sact = lane >= sfirst && lane < slast;
                // P2a: every column writes the positions of its openers, in order
                {
                    int j = excl - sbase;
                    const uint32_t tag = (uint32_t)wb | ((uint32_t)lane << 16);
#pragma unroll
                    for (int b = 0; b < ENC_NBLK; ++b) {
                        uint32_t cur = (sact && n_op > 0) ? sm[b] : 0u;      // (a column handed to the exact path reserved no entries)
                        while (__ballot(cur != 0u) != 0ull) {
                            if (cur != 0u) {
                                ent[j] = uint2{tag + (uint32_t)(32 * b + __builtin_ctz(cur)), (uint32_t)lend};
                                cur &= cur - 1u; ++j;
                            }
                        }
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                // P2b: one opener per lane: digits, skipped bytes out of the counts again, the record of a counted indel.  (Two or three
                // groups of 64 per lane in flight together were measured: the half-empty last group costs more than the overlap gains.)
                int n_cnt = 0;                                                     // counted indels of the segment so far (uniform)
                for (int r0i = 0; r0i < T; r0i += 64) {
                    const int j = r0i + lane;
                    const bool valid = j < T;
                    uint2 e0 = ent[valid ? j : 0];
                    // lanes beyond the last opener take a record of their own (staged position 0, no bytes: every read below stays inside
                    // the stage buffer, every result is discarded behind `valid`); from the second trip on slot 0 holds a COMPACTED record
                    // whose length bits would otherwise be read as position bits
                    if (!valid) e0 = uint2{0u, 0u};
                    const int p = e0.x & 0xffff, owner = (int)(e0.x >> 16), lend_o = (int)e0.y;
                    const int b = st[p];
                    const int w0 = (p + 1) >> 2;
                    const uint32_t a0 = st32[w0], a1 = st32[w0 + 1], a2 = st32[w0 + 2];
                    const int sh = (p + 1) & 3;
                    const uint32_t wlo = __builtin_amdgcn_alignbyte(a1, a0, sh), whi = __builtin_amdgcn_alignbyte(a2, a1, sh);   // byte i = st[p + 1 + i]
                    const int avail1 = lend_o - (p + 1);
                    const bool caret = b == '^';
                    const uint32_t d0 = (wlo & 0xffu) - '0', d1 = ((wlo >> 8) & 0xffu) - '0', d2 = ((wlo >> 16) & 0xffu) - '0', d3 = (wlo >> 24) - '0';
                    const bool k0 = !caret && avail1 > 0 && d0 < 10u, k1 = k0 && avail1 > 1 && d1 < 10u, k2 = k1 && avail1 > 2 && d2 < 10u;
                    const bool mybad = k2 && avail1 > 3 && d3 < 10u;                                  // four digits and more: exact path
                    const int L = k2 ? 3 : (k1 ? 2 : (k0 ? 1 : 0));
                    const int adv = caret ? 1 : (k2 ? (int)(d0 * 100 + d1 * 10 + d2) : (k1 ? (int)(d0 * 10 + d1) : (k0 ? (int)d0 : 0)));
                    const int q = p + 1 + L;
                    const int avail = lend_o - q;
                    const int nskip = adv < avail ? adv : (avail > 0 ? avail : 0);
                    const uint32_t al = __builtin_amdgcn_alignbyte(whi, wlo, L);                      // the first four skipped bytes
                    const uint32_t keep = nskip >= 4 ? 0xffffffffu : ((1u << (8 * (nskip & 3))) - 1u);
                    const uint32_t alm = al | ~keep;                                                  // bytes beyond the allele: 0xff (a zero row)
                    const uint32_t alp = alm ^ ((alm >> 2) & 0x0c0c0c0cu);
                    const uint4 r0 = tab[alp & 0xffu], r1 = tab[(alp >> 8) & 0xffu], r2 = tab[(alp >> 16) & 0xffu], r3 = tab[alp >> 24];
                    const bool counted = valid && !caret && adv <= MAX_INDEL;                         // (then nskip <= 60)
                    uint32_t nx = r0.x + r1.x, ny = r0.y + r1.y, nz = r0.z + r1.z;
                    nx += r2.x + r3.x; ny += r2.y + r3.y; nz += r2.z + r3.z;
                    uint32_t inner = (r0.w | r1.w) | (r2.w | r3.w);                                  // an opener among the skipped bytes
                    if (valid) {
                        for (int k = q + 4; k < q + nskip; ++k) {                                     // longer alleles (rare)
                            const uint32_t bk = st[k];
                            const uint4 rr = tab[bk ^ ((bk >> 2) & 12u)];
                            nx += rr.x; ny += rr.y; nz += rr.z; inner |= rr.w;
                        }
                        atomicAdd(&cacc[0][owner], nx); atomicAdd(&cacc[1][owner], ny);
                        if (nz & 0xffffffu) atomicAdd(&cacc[2][owner], nz & 0xffffffu);
                        // a flagged byte inside the bytes a construct consumes is not an opener (pass 1 cannot know): exact path; so is a
                        // counted allele the end of its column cuts short (an allele of its own in the reference: IndelIterT)
                        if (mybad || inner || (counted && adv > nskip)) atomicOr(&cacc[2][owner], 0x80000000u);   // (bit 31 of word 2)
                    }
                    // the counted indels move to the front of the list, order kept (slot c <= j: every slot of this trip has been read)
                    const unsigned long long cm = __ballot(counted);
                    const uint32_t fwd = (r0.z >> 24) & 1u;                                           // first allele byte in "ACGTN*" (0 without allele)
                    if (counted)
                        ent[n_cnt + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(cm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)cm, 0u))] =
                            uint2{(uint32_t)q | ((uint32_t)nskip << 13) | ((uint32_t)(b == '-') << 19) | (fwd << 20) | ((uint32_t)owner << 25), al & keep};
                    n_cnt += __builtin_popcountll(cm);
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                // P3: multiplicity of every counted indel among the earlier ones of its column, m(e) = #{f <= e equal to e}: every lane
                // steps back through the list, P3_STEP records per trip (all in flight together).  A record of the same column with the
                // same length, sign and first four bytes differs from mine in the position bits only (x ^ x' < 2^13), one of another
                // column (or an end record) in bit 25 or above.  The column's totals and maxima by kind through LDS atomics
                for (int r0i = 0; r0i < n_cnt; r0i += 64) {
                    const int j = r0i + lane;
                    bool go = j < n_cnt;
                    const uint2* f = ent + (go ? j : 0);
                    const uint2 me = *f;
                    const int le = (me.x >> 13) & 63, qe = me.x & 0x1fff;
                    int same = 1;
                    while (__ballot(go) != 0ull) {
                        if (go) {
                            f -= P3_STEP;
                            uint2 o[P3_STEP];
#pragma unroll
                            for (int u = 0; u < P3_STEP; ++u) o[u] = f[P3_STEP - 1 - u];
                            uint32_t hit = 0u;                                     // bit u: record u equals mine as far as the record tells
#pragma unroll
                            for (int u = 0; u < P3_STEP; ++u) {
                                const uint32_t t = o[u].x ^ me.x;
                                go = go && t < (1u << 25);
                                hit |= (uint32_t)(go && t < (1u << 13) && o[u].y == me.y) << u;
                            }
                            same += __builtin_popcount(hit);
                            if (le > 4) {
                                // alleles of more than four bytes: the rest byte by byte
                                for (; hit; hit &= hit - 1u) {
                                    const int qf = f[P3_STEP - 1 - __builtin_ctz(hit)].x & 0x1fff;
                                    for (int k = 4; k < le; ++k) if (st[qe + k] != st[qf + k]) { --same; break; }
                                }
                            }
                        }
                    }
                    if (j < n_cnt) {
                        const int kind = (int)((me.x >> 19) & 1u) * 2 + (int)(((me.x >> 20) & 1u) ^ 1u);
                        const int col = (int)(me.x >> 25);
                        atomicAdd(&cacc[3][col], 1u << (8 * kind));
                        atomicMax(&cacc[4 + kind][col], (uint32_t)same);
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                sfirst = slast;
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            {
                const uint4 a0 = uint4{cacc[0][lane], cacc[1][lane], cacc[2][lane], cacc[3][lane]};
                const uint4 a1 = uint4{cacc[4][lane], cacc[5][lane], cacc[6][lane], cacc[7][lane]};
                ax -= a0.x; ay -= a0.y; az -= a0.z;
                if (act) {
                    cnt[0] = ax & 0xff; cnt[1] = (ax >> 8) & 0xff; cnt[2] = (ax >> 16) & 0xff; cnt[3] = ax >> 24;
                    cnt[4] = ay & 0xff; cnt[5] = (ay >> 8) & 0xff; cnt[6] = (ay >> 16) & 0xff; cnt[7] = ay >> 24;
                    cnt[8] = az & 0xff; cnt[9] = (az >> 8) & 0xff;
                    tot = Quad{(int)(a0.w & 0xff), (int)((a0.w >> 8) & 0xff), (int)((a0.w >> 16) & 0xff), (int)(a0.w >> 24)};
                    mx = Quad{(int)a1.x, (int)a1.y, (int)a1.z, (int)a1.w};
                }
                bad = bad || (act && (a0.z >> 31));
            }
            if (bad) {
                // exact byte-at-a-time scan of this column out of LDS (tensor_maker.cpp:83-114 in structure): openers inside skipped
                // bytes, four-digit lengths, columns beyond the fast path's length or with more openers than the entry list holds
                Counts10 c2; Quad t2, m2;
                scan_column_exact((const uint8_t*)st, (int64_t)lbeg, (int64_t)lend, c2, t2, m2);
                cnt[0] = c2.k0; cnt[1] = c2.k1; cnt[2] = c2.k2; cnt[3] = c2.k3; cnt[4] = c2.k4;
                cnt[5] = c2.k5; cnt[6] = c2.k6; cnt[7] = c2.k7; cnt[8] = c2.k8; cnt[9] = c2.k9;
                tot = t2; mx = m2;
            }
        }
        __builtin_amdgcn_wave_barrier();            // the stage buffer is reused by the next sub-batch
        first = last;
    }
    if (slow) {
        Counts10 c2; Quad t2, m2;
        scan_column_exact(bases, begin64, lane == n_live - 1 ? wave_end : woff[lane + 1], c2, t2, m2);
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
    const int rb = nt4(refraw);
    const int chr_idx = rb < 4 ? rb : 0;
    const int lc[6] = {cnt[0] + cnt[4], cnt[1] + cnt[5], tot.v2 + tot.v3, cnt[2] + cnt[6], tot.v0 + tot.v1, cnt[3] + cnt[7]};
    const int lk[6] = {0, 1, 5, 2, 4, 3};
    int top = -1, topc = 0;
#pragma unroll
    for (int k = 0; k < 6; ++k) if (lc[k] > topc) { topc = lc[k]; top = lk[k]; }
    bool pass_snp = false, pass_indel = false;
    if (__ballot(depth > 255) == 0ull) {
        // every column of the wave inside the table (always, unless a column went through the exact path with more than 255 reads)
        const int thr = (int)((af_min[depth >> 1] >> (16 * (depth & 1))) & 0xffffu);
        const int thri = (int)((afi_min[depth >> 1] >> (16 * (depth & 1))) & 0xffffu);
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const bool cand = lc[k] > 0 && lk[k] != chr_idx;
            if (lk[k] >= 4) pass_indel = pass_indel || (cand && lc[k] >= thri); else pass_snp = pass_snp || (cand && lc[k] >= thr);
        }
    } else {
        const uint32_t den = (uint32_t)(depth ? depth : 1);
        const unsigned __int128 rhs = (unsigned __int128)af.t * den, rhsi = (unsigned __int128)afi.t * den;
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            if (lc[k] <= 0 || lk[k] == chr_idx) continue;
            if (lk[k] >= 4) pass_indel = pass_indel || (afi.mode == 0 ? (((unsigned __int128)(uint32_t)lc[k] << afi.k) >= rhsi) : afi.mode == 1);
            else pass_snp = pass_snp || (af.mode == 0 ? (((unsigned __int128)(uint32_t)lc[k] << af.k) >= rhs) : af.mode == 1);
        }
    }
    const bool pass_af = (top >= 0 && top != chr_idx) || pass_snp || pass_indel;
    const int up_ch[4] = {CH_A, CH_C, CH_G, CH_T}, lo_ch[4] = {CH_a, CH_c, CH_g, CH_t};
#pragma unroll
    for (int k = 0; k < 4; ++k) if (k == chr_idx) { t[up_ch[k]] = -up; t[lo_ch[k]] = -lo; }

    __builtin_amdgcn_wave_barrier();
    int32_t* out_stage = reinterpret_cast<int32_t*>(st);
#pragma unroll
    for (int k = 0; k < NCH; k += 2) *reinterpret_cast<int2*>(out_stage + lane * NCH + k) = int2{t[k], t[k + 1]};
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const int n_valid = n_live * NCH;
    int32_t* __restrict__ dst = counts + wave_col0 * NCH;
    if ((reinterpret_cast<uintptr_t>(dst) & 15) == 0) {
        const int n16 = n_valid >> 2;
        for (int k = lane; k < n16; k += 64) reinterpret_cast<int4*>(dst)[k] = reinterpret_cast<const int4*>(out_stage)[k];
        for (int k = (n16 << 2) + lane; k < n_valid; k += 64) dst[k] = out_stage[k];
    } else {
        for (int k = lane; k < n_valid; k += 64) dst[k] = out_stage[k];
    }
    if (live) {
        (depth_out + wave_col0)[lane] = depth;
        uint8_t f = 0;
        if (pass_af) f |= NSNP_FLAG_PASS_AF;
        if (pass_snp) f |= NSNP_FLAG_PASS_SNP;
        if (pass_indel) f |= NSNP_FLAG_PASS_INDEL;
        if (rb < 4 && pass_af && depth >= min_cov) f |= NSNP_FLAG_CANDIDATE;
        (flags + wave_col0)[lane] = f;
    }
}


// ---- site selection: candidate && 33 consecutive positions around it -------------------------------
__device__ __forceinline__ bool site_ok(const int64_t* pos, const uint8_t* flags, int64_t M, int64_t c)
{
    if (!(flags[c] & NSNP_FLAG_CANDIDATE)) return false;
    if (c < PCENTER || c + PCENTER >= M) return false;
    if (!(pos[c + PCENTER] - pos[c] == PCENTER && pos[c] - pos[c - PCENTER] == PCENTER)) return false;
    // main.cpp:174-178 resets its window at EVERY position that is not the previous one + 1.  For ascending positions the two end
    // differences say it all; a text whose positions repeat or step back (concatenated or damaged input) can have a gap of two and a
    // repeated position cancel inside the window, so a candidate that passed is confirmed step by step (2 % of the columns get here)
    // (all 33 positions loaded at once, no early exit: a loop that stops at the first bad step makes every load wait for the one before it,
    //  and with one or two candidates per 64 columns every wave walked it - the selection kernels took 36 + 28 us per 760 k columns)
    int64_t p[2 * PCENTER + 1];
#pragma unroll
    for (int k = 0; k <= 2 * PCENTER; ++k) p[k] = pos[c - PCENTER + k];
    bool bad = false;
#pragma unroll
    for (int k = 0; k < 2 * PCENTER; ++k) bad |= p[k + 1] - p[k] != 1;
    return !bad;
}

constexpr int SEL_BLOCK = 256, SEL_PER_THREAD = 8, SEL_TILE = SEL_BLOCK * SEL_PER_THREAD;

// A block owns SEL_TILE consecutive columns as SEL_PER_THREAD rows of SEL_BLOCK: in trip k a thread looks at column base + k * SEL_BLOCK + tid,
// so the lanes of a wave read consecutive positions and flags (round 6: a thread used to own eight consecutive columns - every load
// instruction touched 64 separate 64-byte segments - and the two kernels took 0.62 ms of a 6 M-column contig's 13 ms).
__global__ __launch_bounds__(SEL_BLOCK) void k_select_count(const int64_t* pos, const uint8_t* flags, int64_t M,
                                                             int64_t* block_cnt)
{
    __shared__ int wsum[SEL_BLOCK / 64];
    const int64_t base = (int64_t)blockIdx.x * SEL_TILE + threadIdx.x;
    int n = 0;
#pragma unroll
    for (int k = 0; k < SEL_PER_THREAD; ++k) { const int64_t c = base + (int64_t)k * SEL_BLOCK; if (c < M && site_ok(pos, flags, M, c)) ++n; }
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
    // ascending output: the sites of trip k lie behind those of the trips before it, inside a trip wave by wave, inside a wave lane by lane
    __shared__ int wcnt[SEL_PER_THREAD][SEL_BLOCK / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t base = (int64_t)blockIdx.x * SEL_TILE + tid;
    unsigned okm = 0; int before[SEL_PER_THREAD];
#pragma unroll
    for (int k = 0; k < SEL_PER_THREAD; ++k) {
        const int64_t c = base + (int64_t)k * SEL_BLOCK;
        const bool ok = c < M && site_ok(pos, flags, M, c);
        const unsigned long long bal = __ballot(ok);
        before[k] = __popcll(bal & ((1ull << lane) - 1ull));
        if (lane == 0) wcnt[k][wave] = __popcll(bal);
        okm |= (unsigned)ok << k;
    }
    __syncthreads();
    int run = 0;                                           // sites in front of (trip k, this wave)
#pragma unroll
    for (int k = 0; k < SEL_PER_THREAD; ++k) {
#pragma unroll
        for (int w = 0; w < SEL_BLOCK / 64; ++w) {
            if (w == wave && ((okm >> k) & 1u)) {
                const int64_t o = block_off[blockIdx.x] + run + before[k];
                if (o < cap) center_idx[o] = base + (int64_t)k * SEL_BLOCK;
            }
            run += wcnt[k][w];
        }
    }
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

extern "C" int nsnp_pileup_encode_columns2(nsnp_ctx* ctx, const uint8_t* bases, const int64_t* col_off,
                                           const uint8_t* ref, int64_t M, double snp_min_af, double indel_min_af, int min_coverage,
                                           int32_t* counts, int32_t* depth, uint8_t* flags, void* stream)
{
    if (!ctx || M < 0 || (M > 0 && (!bases || !col_off || !ref || !counts || !depth || !flags))) return NSNP_EINVAL;
    if (M == 0) return NSNP_OK;
    const unsigned grid = (unsigned)NSNP_CDIV(M, ENC_BLOCK);
    ScopedKernelTimer tm(ctx, NSNP_K_ENCODE, (hipStream_t)stream);
    // threshold + table of the last thresholds are kept in the context (a caller passes the same values call after call)
    static_assert(sizeof(AfTable) == sizeof(ctx->af_table_words), "AfTable is 128 words");
    uint64_t af_bits; memcpy(&af_bits, &snp_min_af, 8);
    if (!ctx->af_cached || ctx->af_bits != af_bits) {
        const AfThreshold a0 = make_af_threshold(snp_min_af);
        const AfTable t0 = make_af_table(a0);
        ctx->af_t = a0.t; ctx->af_k = a0.k; ctx->af_mode = a0.mode;
        memcpy(ctx->af_table_words, t0.w, sizeof(t0.w));
        ctx->af_bits = af_bits; ctx->af_cached = true;
    }
    uint64_t af2_bits; memcpy(&af2_bits, &indel_min_af, 8);
    if (!ctx->af2_cached || ctx->af2_bits != af2_bits) {
        const AfThreshold a0 = make_af_threshold(indel_min_af);
        const AfTable t0 = make_af_table(a0);
        ctx->af2_t = a0.t; ctx->af2_k = a0.k; ctx->af2_mode = a0.mode;
        memcpy(ctx->af2_table_words, t0.w, sizeof(t0.w));
        ctx->af2_bits = af2_bits; ctx->af2_cached = true;
    }
    AfThreshold af{ctx->af_t, ctx->af_k, ctx->af_mode}, afi{ctx->af2_t, ctx->af2_k, ctx->af2_mode};
    AfTable aft, afti; memcpy(aft.w, ctx->af_table_words, sizeof(aft.w)); memcpy(afti.w, ctx->af2_table_words, sizeof(afti.w));
    hipLaunchKernelGGL(k_encode_columns, dim3(grid), dim3(ENC_BLOCK), 0, (hipStream_t)stream,
                       bases, col_off, ref, M, af, aft, afi, afti, min_coverage, counts, depth, flags);
    NSNP_HIP(ctx, hipGetLastError());
    return NSNP_OK;
}

extern "C" int nsnp_pileup_encode_columns(nsnp_ctx* ctx, const uint8_t* bases, const int64_t* col_off,
                                          const uint8_t* ref, int64_t M, double min_af, int min_coverage,
                                          int32_t* counts, int32_t* depth, uint8_t* flags, void* stream)
{
    return nsnp_pileup_encode_columns2(ctx, bases, col_off, ref, M, min_af, min_af, min_coverage, counts, depth, flags, stream);
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

// meta = { sites selected, how many of them lie in front of column own_lo, in front of own_hi, sites selected }: the selected centres are
// ascending, so the sites a chunk OWNS (its columns without the halo it re-reads) are one run [meta[1], meta[2]) of the list.  One wave:
// two binary searches.  meta may be pinned host memory (plain stores).
namespace {
__global__ void k_select_bounds(const int64_t* __restrict__ center_idx, const int64_t* __restrict__ n_sites, int64_t cap, int64_t own_lo, int64_t own_hi,
                                int64_t* __restrict__ meta)
{
    if (threadIdx.x >= 2) return;
    const int64_t n = *n_sites < cap ? *n_sites : cap;
    const int64_t key = threadIdx.x ? own_hi : own_lo;
    int64_t a = 0, b = n;                                  // first index whose centre is >= key
    while (a < b) { const int64_t m = (a + b) >> 1; if (center_idx[m] < key) a = m + 1; else b = m; }
    meta[1 + threadIdx.x] = a;
    if (threadIdx.x == 0) { meta[0] = *n_sites; meta[3] = *n_sites; }
}

// the call rows of the text pipeline: [N][13] float64 = position, genotype / zygosity argmax and max, the eight coverage channels of
// PileupModel/predict.py:63 at the centre column (all exact in float64) - one thread per site instead of a dozen elementwise launches
__global__ void k_call_rows(const int32_t* __restrict__ counts, const int64_t* __restrict__ center_idx, const int64_t* __restrict__ pos,
                            const uint8_t* __restrict__ ga, const uint8_t* __restrict__ za, const float* __restrict__ gm, const float* __restrict__ zm,
                            int64_t n, double* __restrict__ rows)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int64_t c = center_idx[i];
    const int32_t* q = counts + c * PC;
    double* r = rows + i * 13;
    r[0] = (double)pos[c]; r[1] = (double)ga[i]; r[2] = (double)za[i]; r[3] = (double)gm[i]; r[4] = (double)zm[i];
    constexpr int CH[8] = {CH_A, CH_C, CH_G, CH_T, CH_a, CH_c, CH_g, CH_t};          // predict.py:63: x[:, 16, [0, 1, 2, 3, 9, 10, 11, 12]]
#pragma unroll
    for (int k = 0; k < 8; ++k) r[5 + k] = (double)q[CH[k]];
}
}  // namespace

extern "C" int nsnp_pileup_select_sites_range(nsnp_ctx* ctx, const int64_t* pos, const uint8_t* flags, int64_t M, int64_t own_lo, int64_t own_hi,
                                              int64_t* center_idx, int64_t cap, int64_t* meta, void* stream)
{
    if (!ctx || M < 0 || cap < 0 || !meta || (M > 0 && (!pos || !flags)) || (cap > 0 && !center_idx)) return NSNP_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    if (M == 0) { NSNP_HIP(ctx, hipMemsetAsync(meta, 0, 4 * sizeof(int64_t), s)); return NSNP_OK; }
    // the site count lives behind the block counts of the selection scratch (one word more than nsnp_pileup_select_sites needs)
    const int64_t n_blocks = NSNP_CDIV(M, SEL_TILE);
    const size_t need = (size_t)(n_blocks + 1) * sizeof(int64_t);
    if (ctx->sel_tmp_bytes < need) {
        NSNP_HIP(ctx, hipStreamSynchronize(s));
        if (ctx->sel_tmp) (void)hipFree(ctx->sel_tmp);
        ctx->sel_tmp = nullptr; ctx->sel_tmp_bytes = 0;
        NSNP_HIP(ctx, hipMalloc((void**)&ctx->sel_tmp, need + need / 4));
        ctx->sel_tmp_bytes = need + need / 4;
    }
    int64_t* n_sites = ctx->sel_tmp + n_blocks;
    hipLaunchKernelGGL(k_select_count, dim3((unsigned)n_blocks), dim3(SEL_BLOCK), 0, s, pos, flags, M, ctx->sel_tmp);
    hipLaunchKernelGGL(k_select_scan, dim3(1), dim3(1024), 0, s, ctx->sel_tmp, n_blocks, n_sites);
    hipLaunchKernelGGL(k_select_scatter, dim3((unsigned)n_blocks), dim3(SEL_BLOCK), 0, s, pos, flags, M,
                       (const int64_t*)ctx->sel_tmp, center_idx, cap);
    hipLaunchKernelGGL(k_select_bounds, dim3(1), dim3(64), 0, s, (const int64_t*)center_idx, (const int64_t*)n_sites, cap, own_lo, own_hi, meta);
    NSNP_HIP(ctx, hipGetLastError());
    return NSNP_OK;
}

extern "C" int nsnp_pileup_call_rows(nsnp_ctx* ctx, const int32_t* counts, const int64_t* center_idx, const int64_t* pos, const uint8_t* gt_arg,
                                     const uint8_t* zy_arg, const float* gt_max, const float* zy_max, int64_t N, double* rows, void* stream)
{
    if (!ctx || N < 0 || (N > 0 && (!counts || !center_idx || !pos || !gt_arg || !zy_arg || !gt_max || !zy_max || !rows))) return NSNP_EINVAL;
    if (N == 0) return NSNP_OK;
    hipLaunchKernelGGL(k_call_rows, dim3((unsigned)NSNP_CDIV(N, (int64_t)256)), dim3(256), 0, (hipStream_t)stream, counts, center_idx, pos, gt_arg, zy_arg,
                       gt_max, zy_max, N, rows);
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
