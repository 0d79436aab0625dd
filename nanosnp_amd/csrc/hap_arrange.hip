// hap_arrange.hip -- per-site read arrangement of the stage-4 feature generator.
//
// Replaces the group section of single_group_pileup_haplotype_feature
// (HaplotypeModel/create_pileup_haplotype.py:140-207) and the pad / truncate of write_to_bins
// (HaplotypeModel/write_to_bins.py:15-30,39-61) for read matrices that are already in memory:
//   keep the reads whose base at the centre column is non-zero (:145-149,:181-185), order them by
//   the HP tag at the centre column (:158-165,:193-200), pad with -2 up to D_out rows and cut at
//   D_out = min(chunk max depth, 3 x coverage).
// The reference sorts with pandas' default quicksort, whose order among equal HP values is
// implementation-defined (and irrelevant to the features, which are sums over rows, except for
// WHICH reads fall off when the site is deeper than D_out); here ties keep their input order.
#include "nsnp_common.hpp"

namespace {

constexpr int ARR_BLOCK = 256;

__global__ __launch_bounds__(ARR_BLOCK) void k_hap_arrange(
    const int32_t* __restrict__ seq, const int32_t* __restrict__ bq, const int32_t* __restrict__ mq,
    const int32_t* __restrict__ hap, const int32_t* __restrict__ n_reads, int R, int L, int D_out,
    int32_t* __restrict__ oseq, int32_t* __restrict__ obq, int32_t* __restrict__ omq, int32_t* __restrict__ ohap,
    int32_t* __restrict__ depth_out)
{
    extern __shared__ int32_t arr_lds[];
    int32_t* key = arr_lds;            // [R] HP at the centre column, INT_MAX for dropped rows
    int32_t* src = arr_lds + R;        // [D_out] source row of each output row, -1 = padding
    const int64_t n = blockIdx.x;
    const int tid = threadIdx.x;
    const int rows = n_reads ? min(n_reads[n], R) : R;
    const size_t ibase = (size_t)n * R * L;
    const int mid = L / 2;
    for (int r = tid; r < R; r += ARR_BLOCK) {
        int32_t k = 0x7fffffff;
        if (r < rows && seq[ibase + (size_t)r * L + mid] != 0) k = hap[ibase + (size_t)r * L + mid];
        key[r] = k;
    }
    for (int d = tid; d < D_out; d += ARR_BLOCK) src[d] = -1;
    __syncthreads();
    int kept = 0;
    for (int r = tid; r < R; r += ARR_BLOCK) {
        const int32_t k = key[r];
        if (k == 0x7fffffff) continue;
        int rank = 0;                   // stable rank among the kept rows
        for (int q = 0; q < R; ++q) {
            const int32_t kq = key[q];
            rank += (kq < k) || (kq == k && q < r);
        }
        if (rank < D_out) src[rank] = r;
    }
    __syncthreads();
    if (depth_out && tid == 0) {
        for (int r = 0; r < R; ++r) kept += key[r] != 0x7fffffff;
        depth_out[n] = kept < D_out ? kept : D_out;
    }
    const size_t obase = (size_t)n * D_out * L;
    for (int e = tid; e < D_out * L; e += ARR_BLOCK) {
        const int d = e / L, l = e - d * L;
        const int r = src[d];
        int32_t a = -2, b = -2, c = -2, h = -2;         // write_to_bins.py:15-30: constant_values=-2
        if (r >= 0) {
            const size_t i = ibase + (size_t)r * L + l;
            a = seq[i]; b = bq[i]; c = mq[i]; h = hap[i];
        }
        oseq[obase + e] = a; obq[obase + e] = b; omq[obase + e] = c; ohap[obase + e] = h;
    }
}

}  // namespace

extern "C" int nsnp_hap_arrange_reads(nsnp_ctx* ctx, const int32_t* seq, const int32_t* bq, const int32_t* mq,
                                      const int32_t* hap, const int32_t* n_reads, int64_t N, int R, int L, int D_out,
                                      int32_t* oseq, int32_t* obq, int32_t* omq, int32_t* ohap, int32_t* depth, void* stream)
{
    if (!ctx || N < 0 || R <= 0 || L <= 0 || D_out <= 0) return NSNP_EINVAL;
    if (N > 0 && (!seq || !bq || !mq || !hap || !oseq || !obq || !omq || !ohap)) return NSNP_EINVAL;
    if (N == 0) return NSNP_OK;
    const size_t lds = (size_t)(R + D_out) * sizeof(int32_t);
    if (lds > 64 * 1024) return NSNP_ESHAPE;
    hipLaunchKernelGGL(k_hap_arrange, dim3((unsigned)N), dim3(ARR_BLOCK), lds, (hipStream_t)stream,
                       seq, bq, mq, hap, n_reads, R, L, D_out, oseq, obq, omq, ohap, depth);
    NSNP_HIP(ctx, hipGetLastError());
    return NSNP_OK;
}
