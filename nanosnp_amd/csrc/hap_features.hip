// hap_features.hip -- per-site haplotype feature reduction.
//
// Replaces get_frequency_feature + the reference row of TestDataset.__getitem__
// (HaplotypeModel/dataset_dev.py:11-87,337-349) and the fp32 cast of predict_dev.py:35-36:
// four int32 read planes [D][L] -> float [105][L] (26 statistics x {all reads, HP1, HP2,
// unphased} + the reference row).
//
// One workgroup per site streams the planes exactly once with coalesced row-major loads:
// thread (rp, l) owns column l and rows d = rp, rp + RP, ...; nothing but the 52 x L running sums
// and a per-row HP mask lives in LDS, so the depth axis is tiled at any coverage (the 60x
// "spill path": D only lengthens the loop).  Integer sums are exact (int64); the divisions are
// float64 with the reference's epsilons, then cast to fp32 -> bit-identical to numpy + .float().
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
    unsigned int* rowmask = reinterpret_cast<unsigned int*>(sums + 4 * NSTAT * L);   // [D] bit g set: any(hap == g)
    const int tid = threadIdx.x;
    const int64_t n = blockIdx.x;
    const size_t plane = (size_t)n * D * L;
    const int RP = HF_BLOCK / L;                  // rows handled in parallel
    const int rp = tid / L, l = tid - rp * L;
    const bool active = rp < RP;

    for (int i = tid; i < 4 * NSTAT * L; i += HF_BLOCK) sums[i] = 0ull;
    for (int i = tid; i < D; i += HF_BLOCK) rowmask[i] = 0u;
    __syncthreads();
    // pass A: which read sets does each row belong to (np.any(hap == g, axis=1), dataset_dev.py:57-59)
    if (active)
        for (int d = rp; d < D; d += RP) {
            const int hv = hap[plane + (size_t)d * L + l];
            if (hv >= 1 && hv <= 3) atomicOr(&rowmask[d], 1u << hv);
        }
    __syncthreads();
    // pass B: column sums for the four read sets
    if (active) {
        long long acc[4][NSTAT];
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int k = 0; k < NSTAT; ++k) acc[g][k] = 0;
        for (int d = rp; d < D; d += RP) {
            const size_t o = plane + (size_t)d * L + l;
            const int s = seq[o];
            if (!((s >= 1 && s <= 4) || s == -1)) continue;      // 0 (not covering) and -2 (padding) match nothing
            const long long b = bq[o], m = mq[o];
            const unsigned int msk = rowmask[d] | 1u;            // bit 0 = "all reads"
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                if (!((msk >> g) & 1u)) continue;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const bool hit = (s == k + 1);
                    acc[g][k] += hit; acc[g][5 + k] += hit ? b : 0; acc[g][9 + k] += hit ? m : 0;
                }
                acc[g][4] += (s == -1);
            }
        }
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int k = 0; k < NSTAT; ++k)
                if (acc[g][k] != 0) atomicAdd(&sums[(g * NSTAT + k) * L + l], (unsigned long long)acc[g][k]);
    }
    __syncthreads();
    // pass C: the 105 x L outputs (row order of get_seq_baseq_mapq_feat, dataset_dev.py:51)
    float* __restrict__ o = out + (size_t)n * 105 * L;
    for (int i = tid; i < 105 * L; i += HF_BLOCK) {
        const int row = i / L, col = i - row * L;
        float v;
        if (row == 104) v = (float)ref_row[n * L + col];
        else {
            const int g = row / 26, r = row - g * 26;
            const long long* S = reinterpret_cast<const long long*>(sums) + (size_t)g * NSTAT * L + col;
#define SUM(k) S[(size_t)(k) * L]
            double x;
            if (r < 5) {            // frequency = cnt / (A+C+G+T+D + 1e-6)            dataset_dev.py:17-22
                const double total = (double)(SUM(0) + SUM(1) + SUM(2) + SUM(3) + SUM(4)) + 1e-6;
                x = (double)SUM(r) / total;
            } else if (r < 10) x = (double)SUM(r - 5);                                  // counts
            else if (r < 14) x = (double)SUM(5 + (r - 10));                             // baseq sums
            else if (r < 18) x = (double)SUM(5 + (r - 14)) / ((double)SUM(r - 14) + 1e-9);   // baseq means
            else if (r < 22) x = (double)SUM(9 + (r - 18));                             // mapq sums
            else             x = (double)SUM(9 + (r - 22)) / ((double)SUM(r - 22) + 1e-9);   // mapq means
#undef SUM
            v = (float)x;
        }
        o[i] = v;
    }
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
    const size_t lds = (size_t)4 * NSTAT * L * 8 + (size_t)D * 4;
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
