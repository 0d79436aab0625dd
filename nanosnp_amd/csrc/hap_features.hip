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

constexpr int HF_BLOCK = 256;
constexpr int HF_MAX_L = 64;
constexpr int NSTAT = 13;          // per read set: cnt A C G T D, baseq sum A C G T, mapq sum A C G T

// PT: element type of the read planes, int32 (what the reference's bins hold) or int8 (every value of the four planes
// fits: base codes -2..4, HP -2..3, base quality <= 93, mapping quality <= 60; a quarter of the PCIe and HBM bytes)
template <typename PT>
// six waves per SIMD (80 registers) with four row groups in flight measured best: 272 us per 16384 sites at L = 33 against 296 at
// five waves, 321 with two groups in flight, 286-334 with six or eight groups (fewer waves)
#ifndef NSNP_HF_MINW
#define NSNP_HF_MINW 6
#endif
__global__ __launch_bounds__(HF_BLOCK, NSNP_HF_MINW) void k_hap_features(
    const PT* __restrict__ seq, const PT* __restrict__ bq, const PT* __restrict__ mq,
    const PT* __restrict__ hap, const int32_t* __restrict__ ref_row, int D, int L, float* __restrict__ out)
{
    extern __shared__ unsigned long long hf_lds[];
    unsigned long long* sums = hf_lds;                                   // [4][NSTAT][L] int64
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t n = blockIdx.x;
    const size_t plane = (size_t)n * D * L;
    const PT* __restrict__ seq_n = seq + plane; const PT* __restrict__ bq_n = bq + plane;
    const PT* __restrict__ mq_n = mq + plane; const PT* __restrict__ hap_n = hap + plane;
#ifndef NSNP_HF_U
#define NSNP_HF_U 4
#endif
    constexpr int NW = HF_BLOCK / 64, U = NSNP_HF_U;      // waves per workgroup, row groups of a wave in flight together
    const int R = L > 32 ? 1 : 64 / L;            // rows per group
    const int r = lane / L, col = lane - r * L;   // this lane's row of the group and column
    const bool active = lane < R * L;
    const int rsh = r * L;                        // first lane of my row
    const uint32_t wm = L >= 32 ? 0xffffffffu : (1u << L) - 1u;          // (used for R > 1 only: L <= 32)

    for (int i = tid; i < 4 * NSTAT * L; i += HF_BLOCK) sums[i] = 0ull;
    __syncthreads();

    uint32_t cntp[4] = {0u, 0u, 0u, 0u};          // per read set: counts of A C G T, 8 bits each
    uint32_t cntd = 0u;                           // deletions: 8 bits per read set
    unsigned long long qb[4] = {0ull, 0ull, 0ull, 0ull}, qm[4] = {0ull, 0ull, 0ull, 0ull};   // base / mapping quality sums of A C G T, 16 bits each
    auto flush = [&]() {
        if (active) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const unsigned long long c = (cntp[g] >> (8 * k)) & 0xffu, b = (qb[g] >> (16 * k)) & 0xffffu, m = (qm[g] >> (16 * k)) & 0xffffu;
                    if (c) atomicAdd(&sums[(g * NSTAT + k) * L + col], c);
                    if (b) atomicAdd(&sums[(g * NSTAT + 5 + k) * L + col], b);
                    if (m) atomicAdd(&sums[(g * NSTAT + 9 + k) * L + col], m);
                }
                const unsigned long long dl = (cntd >> (8 * g)) & 0xffu;
                if (dl) atomicAdd(&sums[(g * NSTAT + 4) * L + col], dl);
                cntp[g] = 0u; qb[g] = 0ull; qm[g] = 0ull;
            }
            cntd = 0u;
        }
    };
#ifdef NSNP_HF_NOLOAD
    const int n_groups = D > 100000 ? (D + R - 1) / R : 0;      // (removal timing build: no row is loaded)
#else
    const int n_groups = (D + R - 1) / R;
#endif
    int since_flush = 0;
    for (int g0 = wave; g0 < n_groups; g0 += NW * U) {
        int sv[U], bv[U], mv[U], hv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int d = (g0 + NW * u) * R + r;
            const bool on = active && d < D;
            const int o = on ? d * L + col : 0;                   // (uniform plane base + 32-bit lane offset)
            sv[u] = on ? (int)seq_n[o] : 0; hv[u] = on ? (int)hap_n[o] : 0;
            bv[u] = on ? (int)bq_n[o] : 0; mv[u] = on ? (int)mq_n[o] : 0;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (g0 + NW * u >= n_groups) break;                   // (uniform)
            const int s = sv[u];                                  // 0 (not covering) and -2 (padding) match nothing
            const bool base = (unsigned)(s - 1) < 4u;
            int b = bv[u], m = mv[u];
            // a quality that the 16-bit fields cannot take (never in a BAM): straight into the LDS sums, for every set of its row
            const bool big = base && ((unsigned)b >= 2048u || (unsigned)m >= 2048u);
            const unsigned long long h1 = __ballot(hv[u] == 1), h2 = __ballot(hv[u] == 2), h3 = __ballot(hv[u] == 3);
            const unsigned long long anybig = __ballot(big);
            const uint32_t c1 = base ? 1u << (8 * (s - 1)) : 0u;
            const int sh = base ? 16 * (s - 1) : 0;
            if (anybig) {
                // (rare) membership per lane, exact adds
                const bool in1 = R == 1 ? h1 != 0ull : ((uint32_t)(h1 >> rsh) & wm) != 0u;
                const bool in2 = R == 1 ? h2 != 0ull : ((uint32_t)(h2 >> rsh) & wm) != 0u;
                const bool in3 = R == 1 ? h3 != 0ull : ((uint32_t)(h3 >> rsh) & wm) != 0u;
                if (big) {
                    const bool in[4] = {true, in1, in2, in3};
#pragma unroll
                    for (int g = 0; g < 4; ++g) if (in[g]) {
                        atomicAdd(&sums[(g * NSTAT + 5 + (s - 1)) * L + col], (unsigned long long)(long long)b);
                        atomicAdd(&sums[(g * NSTAT + 9 + (s - 1)) * L + col], (unsigned long long)(long long)m);
                    }
                    b = 0; m = 0;
                }
            }
            const unsigned long long b64 = base ? (unsigned long long)(uint32_t)b << sh : 0ull;
            const unsigned long long m64 = base ? (unsigned long long)(uint32_t)m << sh : 0ull;
            const bool del = s == -1;
            cntp[0] += c1; qb[0] += b64; qm[0] += m64;
            if (R == 1) {
                // one row per wave: the sets are uniform
                uint32_t dbits = 1u;
                if (h1) { cntp[1] += c1; qb[1] += b64; qm[1] += m64; dbits |= 1u << 8; }
                if (h2) { cntp[2] += c1; qb[2] += b64; qm[2] += m64; dbits |= 1u << 16; }
                if (h3) { cntp[3] += c1; qb[3] += b64; qm[3] += m64; dbits |= 1u << 24; }
                cntd += del ? dbits : 0u;
            } else {
                const bool in1 = ((uint32_t)(h1 >> rsh) & wm) != 0u, in2 = ((uint32_t)(h2 >> rsh) & wm) != 0u, in3 = ((uint32_t)(h3 >> rsh) & wm) != 0u;
                cntp[1] += in1 ? c1 : 0u; qb[1] += in1 ? b64 : 0ull; qm[1] += in1 ? m64 : 0ull;
                cntp[2] += in2 ? c1 : 0u; qb[2] += in2 ? b64 : 0ull; qm[2] += in2 ? m64 : 0ull;
                cntp[3] += in3 ? c1 : 0u; qb[3] += in3 ? b64 : 0ull; qm[3] += in3 ? m64 : 0ull;
                cntd += del ? (1u | (in1 ? 1u << 8 : 0u) | (in2 ? 1u << 16 : 0u) | (in3 ? 1u << 24 : 0u)) : 0u;
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
    hipLaunchKernelGGL(k_hap_features<PT>, dim3((unsigned)N), dim3(HF_BLOCK), lds, (hipStream_t)stream,
                       seq, bq, mq, hap, ref_row, D, L, out);
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
