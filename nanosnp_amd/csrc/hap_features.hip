// hap_features.hip -- per-site haplotype feature reduction.
//
// Replaces get_frequency_feature + the reference row of TestDataset.__getitem__
// (HaplotypeModel/dataset_dev.py:11-87,337-349) and the fp32 cast of predict_dev.py:35-36:
// four int32 read planes [D][L] -> float [105][L] (26 statistics x {all reads, HP1, HP2,
// unphased} + the reference row).
//
// One workgroup per site streams the planes exactly once.  A wave takes R = 64 / L consecutive rows at a time (one for L = 33,
// five for L = 11: the rows are contiguous in memory, so the wave's load is one contiguous run), lane l the element l of the run:
// row l / L, column l % L.  A row's membership in the read sets (np.any(hap == g, axis=1)) is a wave ballot restricted to the
// row's lanes - no second pass over the hap plane, no row mask in LDS, no barrier before the sums; with one row per wave the set
// tests are scalar branches.  The loads of four row groups (16 per lane) are in flight together.
// The kernel is bound by vector-instruction issue, not by HBM (a quarter of the bytes - the int8 planes - take the same time), so
// the running sums are PACKED: per read set one register of four 8-bit counts and two 64-bit registers of four 16-bit quality
// sums, indexed by a shift with the base code instead of four predicated adds each; a lane flushes them into the 52 x L int64
// sums in LDS every 31 rows (255 / 8 bits, 31 x 2047 < 2^16).  A quality outside [0, 2048) - nothing a BAM holds - goes to the LDS
// sums directly, so any int32 input gives the exact integer sums; the divisions are float64 with the reference's epsilons, then
// cast to fp32 -> bit-identical to numpy + .float().  Only the 52 x L sums live in LDS, so the depth axis is tiled at any
// coverage (the 60x "spill path": D only lengthens the loop).
#include "nsnp_common.hpp"

namespace {

#ifndef NSNP_HF_U
#define NSNP_HF_U 4
#endif
constexpr int HF_BLOCK = 256;
constexpr int HF_MAX_L = 64;
constexpr int NSTAT = 13;          // per read set: cnt A C G T D, baseq sum A C G T, mapq sum A C G T

// PT: element type of the read planes, int32 (what the reference's bins hold) or int8 (every value of the four planes
// fits: base codes -2..4, HP -2..3, base quality <= 93, mapping quality <= 60; a quarter of the PCIe and HBM bytes).
// LT: the window length as a compile-time constant (33, 11) or 0 = the runtime L (any L <= 64): with LT the row stride is a
// constant and the loads of a wave's four row groups share one address register with immediate offsets.
//
// Round 5: the kernel sits AT its vector-issue bound (SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES = 0.176 per wave at six waves per SIMD =
// 1.05 of the port: profiles/r05_haplotype_sq_counters.json; ~2,000 vector instructions per wave and site), so the body was
// rewritten for instruction count, same sums bit for bit:
//   * counts: ONE 64-bit register per read set with eight 8-bit fields indexed by min(s + 2, 7) - padding, deletion, not covering,
//     A, C, G, T, junk - so that an element's count (deletions included) is one shift and one 64-bit add with no test of the code;
//   * qualities: a non-base element contributes b = m = 0, so its shift amount needs no select;
//   * the test for qualities the 16-bit fields cannot hold (never in a BAM) is one OR per element and one ballot per four rows
//     instead of two compares, a ballot and a branch per element;
//   * the row groups whose rows all exist run without per-load predicates (the last, ragged trip keeps them);
//   * the flush adds every field unconditionally (an add of zero costs less than the test for it).
template <typename PT, int LT>
// six waves per SIMD (80 registers) with four row groups in flight measured best: 272 us per 16384 sites at L = 33 against 296 at
// five waves, 321 with two groups in flight, 286-334 with six or eight groups (fewer waves)
#ifndef NSNP_HF_MINW
#define NSNP_HF_MINW 6
#endif
__global__ __launch_bounds__(HF_BLOCK, NSNP_HF_MINW) void k_hap_features(
    const PT* __restrict__ seq, const PT* __restrict__ bq, const PT* __restrict__ mq,
    const PT* __restrict__ hap, const int32_t* __restrict__ ref_row, int D, int L_rt, float* __restrict__ out)
{
    extern __shared__ unsigned long long hf_lds[];
    unsigned long long* sums = hf_lds;                                   // [4][NSTAT][L] int64
    const int L = LT ? LT : L_rt;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t n = blockIdx.x;
    const size_t plane = (size_t)n * D * L;
    const PT* __restrict__ seq_n = seq + plane; const PT* __restrict__ bq_n = bq + plane;
    const PT* __restrict__ mq_n = mq + plane; const PT* __restrict__ hap_n = hap + plane;
    constexpr int NW = HF_BLOCK / 64, U = NSNP_HF_U;      // waves per workgroup, row groups of a wave in flight together
    const int R = L > 32 ? 1 : 64 / L;            // rows per group
    const int r = lane / L, col = lane - r * L;   // this lane's row of the group and column
    const bool active = lane < R * L;
    const int rsh = r * L;                        // first lane of my row
    const uint32_t wm = L >= 32 ? 0xffffffffu : (1u << L) - 1u;          // (used for R > 1 only: L <= 32)

    for (int i = tid; i < 4 * NSTAT * L; i += HF_BLOCK) sums[i] = 0ull;
    __syncthreads();

    // per read set: counts (eight 8-bit fields by min(s + 2, 7): -2 padding, -1 deletion, 0 not covering, A C G T, junk), base / mapping
    // quality sums of A C G T (16 bits each)
    unsigned long long cnt[4] = {0ull, 0ull, 0ull, 0ull}, qb[4] = {0ull, 0ull, 0ull, 0ull}, qm[4] = {0ull, 0ull, 0ull, 0ull};
    auto flush = [&]() {
        if (active) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    atomicAdd(&sums[(g * NSTAT + k) * L + col], (cnt[g] >> (8 * (k + 3))) & 0xffull);
                    atomicAdd(&sums[(g * NSTAT + 5 + k) * L + col], (qb[g] >> (16 * k)) & 0xffffull);
                    atomicAdd(&sums[(g * NSTAT + 9 + k) * L + col], (qm[g] >> (16 * k)) & 0xffffull);
                }
                atomicAdd(&sums[(g * NSTAT + 4) * L + col], (cnt[g] >> 8) & 0xffull);
                cnt[g] = 0ull; qb[g] = 0ull; qm[g] = 0ull;
            }
        }
    };
#ifdef NSNP_HF_NOLOAD
    const int n_groups = D > 100000 ? (D + R - 1) / R : 0;      // (removal timing build: no row is loaded)
#else
    const int n_groups = (D + R - 1) / R;
#endif
    const int n_full = D / R;                     // groups whose R rows all exist
    int since_flush = 0;
    for (int g0 = wave; g0 < n_groups; g0 += NW * U) {
        int sv[U], bv[U], mv[U], hv[U];
        if (g0 + NW * (U - 1) < n_full) {
            // every row of the trip exists: ONE unsigned lane offset on the uniform plane base, constant strides (with LT: immediate
            // offsets of the load instructions - no address arithmetic per load).  Idle lanes (lane >= R x L) read element 0 of the
            // trip's rows - rows that exist - and their values are dropped below.
            const uint32_t o0 = active ? (uint32_t)((g0 * R + r) * L + col) * (uint32_t)sizeof(PT) : 0u;
            const uint32_t stride = (uint32_t)(NW * R * L) * (uint32_t)sizeof(PT);
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint32_t o = o0 + (uint32_t)u * stride;
                sv[u] = (int)*reinterpret_cast<const PT*>(reinterpret_cast<const char*>(seq_n) + o);
                hv[u] = (int)*reinterpret_cast<const PT*>(reinterpret_cast<const char*>(hap_n) + o);
                bv[u] = (int)*reinterpret_cast<const PT*>(reinterpret_cast<const char*>(bq_n) + o);
                mv[u] = (int)*reinterpret_cast<const PT*>(reinterpret_cast<const char*>(mq_n) + o);
            }
            if (!active) {
#pragma unroll
                for (int u = 0; u < U; ++u) { sv[u] = 0; hv[u] = 0; }
            }
        } else {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int d = (g0 + NW * u) * R + r;
                const bool on = active && d < D;
                const int o = on ? d * L + col : 0;                   // (uniform plane base + 32-bit lane offset)
                sv[u] = on ? (int)seq_n[o] : 0; hv[u] = on ? (int)hap_n[o] : 0;
                bv[u] = on ? (int)bq_n[o] : 0; mv[u] = on ? (int)mq_n[o] : 0;
            }
        }
        // qualities of the elements that are counted; one test per trip for values the 16-bit fields cannot hold (never in a BAM)
        uint32_t orbm = 0u;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const bool base = (unsigned)(sv[u] - 1) < 4u;             // 0 (not covering), -1, -2 (padding) and foreign codes sum no quality
            bv[u] = base ? bv[u] : 0; mv[u] = base ? mv[u] : 0;
            orbm |= (uint32_t)bv[u] | (uint32_t)mv[u];
        }
        if (__ballot(orbm >= 2048u)) {
            // (rare) such a quality goes straight into the LDS sums, for every set of its row - exact for any int32 input
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (g0 + NW * u >= n_groups) break;                   // (uniform)
                const unsigned long long h1 = __ballot(hv[u] == 1), h2 = __ballot(hv[u] == 2), h3 = __ballot(hv[u] == 3);
                const bool big = (unsigned)bv[u] >= 2048u || (unsigned)mv[u] >= 2048u;      // (then the element is a base: the others are 0)
                if (big) {
                    const bool in[4] = {true, R == 1 ? h1 != 0ull : ((uint32_t)(h1 >> rsh) & wm) != 0u,
                                        R == 1 ? h2 != 0ull : ((uint32_t)(h2 >> rsh) & wm) != 0u,
                                        R == 1 ? h3 != 0ull : ((uint32_t)(h3 >> rsh) & wm) != 0u};
#pragma unroll
                    for (int g = 0; g < 4; ++g) if (in[g]) {
                        atomicAdd(&sums[(g * NSTAT + 5 + (sv[u] - 1)) * L + col], (unsigned long long)(long long)bv[u]);
                        atomicAdd(&sums[(g * NSTAT + 9 + (sv[u] - 1)) * L + col], (unsigned long long)(long long)mv[u]);
                    }
                    bv[u] = 0; mv[u] = 0;
                }
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (g0 + NW * u >= n_groups) break;                   // (uniform)
            const int s = sv[u];
            const unsigned long long h1 = __ballot(hv[u] == 1), h2 = __ballot(hv[u] == 2), h3 = __ballot(hv[u] == 3);
            const uint32_t code = min((uint32_t)(s + 2), 7u);     // any foreign code (s < -2 wraps around) lands in the junk field
            const unsigned long long c64 = 1ull << (8 * code);
            const uint32_t sh = (uint32_t)(16 * s - 16) & 63u;    // (a non-base element shifts zeros: any amount will do)
            const unsigned long long b64 = (unsigned long long)(uint32_t)bv[u] << sh, m64 = (unsigned long long)(uint32_t)mv[u] << sh;
            cnt[0] += c64; qb[0] += b64; qm[0] += m64;
            if (R == 1) {
                // one row per wave: the sets are uniform - REAL scalar branches (the empty asm keeps hipcc from turning each of them
                // into six selects: a row of the reference's bins is in one of the three sets, so two of three bodies are skipped)
                if (h1) { asm volatile("" ::: "memory"); cnt[1] += c64; qb[1] += b64; qm[1] += m64; }
                if (h2) { asm volatile("" ::: "memory"); cnt[2] += c64; qb[2] += b64; qm[2] += m64; }
                if (h3) { asm volatile("" ::: "memory"); cnt[3] += c64; qb[3] += b64; qm[3] += m64; }
            } else {
                const bool in1 = ((uint32_t)(h1 >> rsh) & wm) != 0u, in2 = ((uint32_t)(h2 >> rsh) & wm) != 0u, in3 = ((uint32_t)(h3 >> rsh) & wm) != 0u;
                cnt[1] += in1 ? c64 : 0ull; qb[1] += in1 ? b64 : 0ull; qm[1] += in1 ? m64 : 0ull;
                cnt[2] += in2 ? c64 : 0ull; qb[2] += in2 ? b64 : 0ull; qm[2] += in2 ? m64 : 0ull;
                cnt[3] += in3 ? c64 : 0ull; qb[3] += in3 ? b64 : 0ull; qm[3] += in3 ? m64 : 0ull;
            }
        }
        since_flush += U;
#ifndef NSNP_HF_NOFLUSH
        if (since_flush > 31 - U) { flush(); since_flush = 0; }   // (uniform)
#endif
    }
#ifndef NSNP_HF_NOFLUSH
    flush();
#endif
    __syncthreads();
    // pass C: the 105 x L outputs (row order of get_seq_baseq_mapq_feat, dataset_dev.py:51).  One thread per (read set, column) reads
    // its 13 sums once, forms the column total once and writes the set's 26 rows (a wave's stores of one row are consecutive floats):
    // 13 float64 divisions per thread instead of one output element at a time with up to six LDS reads each.
    float* __restrict__ o = out + (size_t)n * 105 * L;
#ifdef NSNP_HF_NOOUT
    if (D > 100000)                                  // (removal timing build: the output pass compiled but never run)
#endif
    for (int i = tid; i < 4 * L; i += HF_BLOCK) {
        const int g = i / L, col = i - g * L;
        const long long* S = reinterpret_cast<const long long*>(sums) + (size_t)g * NSTAT * L + col;
        float* og = o + (size_t)g * 26 * L + col;
        // (rolled loops that re-read the sums from LDS: the float64 divisions would otherwise hold 13 int64 and their doubles in
        // registers, which costs the kernel two waves per SIMD - and it waits on memory, not on instruction issue)
        long long tot = 0;
#pragma unroll 1
        for (int r = 0; r < 5; ++r) tot += S[(size_t)r * L];
        const double total = (double)tot + 1e-6;                                            // dataset_dev.py:17-22
#pragma unroll 1
        for (int r = 0; r < 5; ++r) {
            const double v = (double)S[(size_t)r * L];
            og[(size_t)r * L] = (float)(v / total);                                         // frequencies
            og[(size_t)(5 + r) * L] = (float)v;                                             // counts
        }
#pragma unroll 1
        for (int k = 0; k < 4; ++k) {
            const double c = (double)S[(size_t)k * L] + 1e-9;
            const double bs = (double)S[(size_t)(5 + k) * L], ms = (double)S[(size_t)(9 + k) * L];
            og[(size_t)(10 + k) * L] = (float)bs;                                           // baseq sums
            og[(size_t)(14 + k) * L] = (float)(bs / c);                                     // baseq means
            og[(size_t)(18 + k) * L] = (float)ms;                                           // mapq sums
            og[(size_t)(22 + k) * L] = (float)(ms / c);                                     // mapq means
        }
    }
    for (int col = tid; col < L; col += HF_BLOCK) o[(size_t)104 * L + col] = (float)ref_row[n * L + col];
}

}  // namespace

namespace {
template <typename PT>
int hap_features_impl(nsnp_ctx* ctx, const PT* seq, const PT* bq, const PT* mq, const PT* hap, const int32_t* ref_row,
                      int64_t N, int D, int L, float* out, void* stream)
{
    if (!ctx || N < 0 || D <= 0 || L <= 0 || L > HF_MAX_L) return NSNP_EINVAL;
    if (N > 0 && (!seq || !bq || !mq || !hap || !ref_row || !out)) return NSNP_EINVAL;
    if (N == 0) return NSNP_OK;
    const size_t lds = (size_t)4 * NSTAT * L * 8;
    if (lds > 64 * 1024) return NSNP_ESHAPE;
    ScopedKernelTimer tm(ctx, NSNP_K_HAPFEAT, (hipStream_t)stream);
    if (L == 33)
        hipLaunchKernelGGL((k_hap_features<PT, 33>), dim3((unsigned)N), dim3(HF_BLOCK), lds, (hipStream_t)stream, seq, bq, mq, hap, ref_row, D, L, out);
    else if (L == 11)
        hipLaunchKernelGGL((k_hap_features<PT, 11>), dim3((unsigned)N), dim3(HF_BLOCK), lds, (hipStream_t)stream, seq, bq, mq, hap, ref_row, D, L, out);
    else
        hipLaunchKernelGGL((k_hap_features<PT, 0>), dim3((unsigned)N), dim3(HF_BLOCK), lds, (hipStream_t)stream, seq, bq, mq, hap, ref_row, D, L, out);
    NSNP_HIP(ctx, hipGetLastError());
    return NSNP_OK;
}
}  // namespace

extern "C" int nsnp_hap_features(nsnp_ctx* ctx, const int32_t* seq, const int32_t* bq, const int32_t* mq,
                                 const int32_t* hap, const int32_t* ref_row, int64_t N, int D, int L,
                                 float* out, void* stream)
{
    return hap_features_impl<int32_t>(ctx, seq, bq, mq, hap, ref_row, N, D, L, out, stream);
}

extern "C" int nsnp_hap_features_i8(nsnp_ctx* ctx, const int8_t* seq, const int8_t* bq, const int8_t* mq,
                                    const int8_t* hap, const int32_t* ref_row, int64_t N, int D, int L,
                                    float* out, void* stream)
{
    return hap_features_impl<int8_t>(ctx, seq, bq, mq, hap, ref_row, N, D, L, out, stream);
}
