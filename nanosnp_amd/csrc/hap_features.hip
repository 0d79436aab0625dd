// hap_features.hip -- per-site haplotype feature reduction.
//
// Replaces get_frequency_feature + the reference row of TestDataset.__getitem__
// (HaplotypeModel/dataset_dev.py:11-87,337-349) and the fp32 cast of predict_dev.py:35-36:
// four int32 read planes [D][L] -> float [105][L] (26 statistics x {all reads, HP1, HP2,
// unphased} + the reference row).
//
// One workgroup per site streams the planes exactly once: wave w owns the rows d = w, w + 4, ..., lane l the column l, so a
// row's membership in the read sets (np.any(hap == g, axis=1)) is a wave ballot - no second pass over the hap plane, no row
// mask in LDS, no barrier before the sums - and the set tests are scalar branches.  The loads of four rows (16 per lane) are
// in flight together.  Only the 52 x L running sums live in LDS, so the depth axis is tiled at any coverage (the 60x "spill
// path": D only lengthens the loop).  Integer sums are exact (counts int32, quality sums int64); the divisions are float64 with
// the reference's epsilons, then cast to fp32 -> bit-identical to numpy + .float().
#include "nsnp_common.hpp"

namespace {

constexpr int HF_BLOCK = 256;
constexpr int HF_MAX_L = 64;
constexpr int NSTAT = 13;          // per read set: cnt A C G T D, baseq sum A C G T, mapq sum A C G T

// PT: element type of the read planes, int32 (what the reference's bins hold) or int8 (every value of the four planes
// fits: base codes -2..4, HP -2..3, base quality <= 93, mapping quality <= 60; a quarter of the PCIe and HBM bytes)
template <typename PT>
__global__ __launch_bounds__(HF_BLOCK) void k_hap_features(
    const PT* __restrict__ seq, const PT* __restrict__ bq, const PT* __restrict__ mq,
    const PT* __restrict__ hap, const int32_t* __restrict__ ref_row, int D, int L, float* __restrict__ out)
{
    extern __shared__ unsigned long long hf_lds[];
    unsigned long long* sums = hf_lds;                                   // [4][NSTAT][L] int64
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t n = blockIdx.x;
    const size_t plane = (size_t)n * D * L;
    const bool active = lane < L;
    constexpr int NW = HF_BLOCK / 64, U = 4;      // waves per workgroup, rows of a wave in flight together

    for (int i = tid; i < 4 * NSTAT * L; i += HF_BLOCK) sums[i] = 0ull;

    int cnt[4][5];                                // per read set: A C G T D
    long long qs[4][8];                           // per read set: baseq sum A C G T, mapq sum A C G T
#pragma unroll
    for (int g = 0; g < 4; ++g) {
#pragma unroll
        for (int k = 0; k < 5; ++k) cnt[g][k] = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) qs[g][k] = 0;
    }
    for (int d0 = wave; d0 < D; d0 += NW * U) {
        int sv[U], bv[U], mv[U], hv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int d = d0 + NW * u;
            const bool on = active && d < D;
            const size_t o = plane + (size_t)(on ? d : 0) * L + (on ? lane : 0);
            sv[u] = on ? (int)seq[o] : 0; hv[u] = on ? (int)hap[o] : 0;
            bv[u] = on ? (int)bq[o] : 0; mv[u] = on ? (int)mq[o] : 0;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (d0 + NW * u >= D) break;                          // (uniform)
            // read sets of this row: bit 0 "all reads", bit g any(hap == g) over the row's columns (dataset_dev.py:57-59)
            const unsigned msk = 1u | (__ballot(hv[u] == 1) ? 2u : 0u) | (__ballot(hv[u] == 2) ? 4u : 0u) | (__ballot(hv[u] == 3) ? 8u : 0u);
            const int s = sv[u];                                  // 0 (not covering) and -2 (padding) match nothing
            const long long b = bv[u], m = mv[u];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                if (!((msk >> g) & 1u)) continue;                 // (uniform)
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const bool hit = (s == k + 1);
                    cnt[g][k] += hit; qs[g][k] += hit ? b : 0; qs[g][4 + k] += hit ? m : 0;
                }
                cnt[g][4] += (s == -1);
            }
        }
    }
    __syncthreads();                                              // sums are zeroed
    if (active) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
#pragma unroll
            for (int k = 0; k < 5; ++k)
                if (cnt[g][k] != 0) atomicAdd(&sums[(g * NSTAT + k) * L + lane], (unsigned long long)cnt[g][k]);
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (qs[g][k] != 0) atomicAdd(&sums[(g * NSTAT + 5 + k) * L + lane], (unsigned long long)qs[g][k]);
        }
    }
    __syncthreads();
    // pass C: the 105 x L outputs (row order of get_seq_baseq_mapq_feat, dataset_dev.py:51).  One thread per (read set, column) reads
    // its 13 sums once, forms the column total once and writes the set's 26 rows (a wave's stores of one row are consecutive floats):
    // 13 float64 divisions per thread instead of one output element at a time with up to six LDS reads each.
    float* __restrict__ o = out + (size_t)n * 105 * L;
    for (int i = tid; i < 4 * L; i += HF_BLOCK) {
        const int g = i / L, col = i - g * L;
        const long long* S = reinterpret_cast<const long long*>(sums) + (size_t)g * NSTAT * L + col;
        long long v[NSTAT];
#pragma unroll
        for (int k = 0; k < NSTAT; ++k) v[k] = S[(size_t)k * L];
        float* og = o + (size_t)g * 26 * L + col;
        const double total = (double)(v[0] + v[1] + v[2] + v[3] + v[4]) + 1e-6;          // dataset_dev.py:17-22
#pragma unroll
        for (int r = 0; r < 5; ++r) {
            og[(size_t)r * L] = (float)((double)v[r] / total);                             // frequencies
            og[(size_t)(5 + r) * L] = (float)(double)v[r];                                 // counts
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const double c = (double)v[k] + 1e-9;
            og[(size_t)(10 + k) * L] = (float)(double)v[5 + k];                            // baseq sums
            og[(size_t)(14 + k) * L] = (float)((double)v[5 + k] / c);                      // baseq means
            og[(size_t)(18 + k) * L] = (float)(double)v[9 + k];                            // mapq sums
            og[(size_t)(22 + k) * L] = (float)((double)v[9 + k] / c);                      // mapq means
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
