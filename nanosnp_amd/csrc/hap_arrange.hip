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
    int32_t* key = arr_lds;            // [R] HP at the centre column (any int32: an HP tag is whatever the BAM holds)
    int32_t* src = arr_lds + R;        // [D_out] source row of each output row, -1 = padding
    const int64_t n = blockIdx.x;
    const int tid = threadIdx.x;
    const int rows = n_reads ? min(n_reads[n], R) : R;
    const size_t ibase = (size_t)n * R * L;
    const int mid = L / 2;
    // rank_acc[r]: -1 for a row that is dropped (beyond n_reads, or no base at the centre column), else its rank
    int32_t* rank_acc = src + D_out;                      // [R]
    for (int r = tid; r < R; r += ARR_BLOCK) {
        const bool keep = r < rows && seq[ibase + (size_t)r * L + mid] != 0;
        key[r] = keep ? hap[ibase + (size_t)r * L + mid] : 0;
        rank_acc[r] = keep ? 0 : -1;
    }
    for (int d = tid; d < D_out; d += ARR_BLOCK) src[d] = -1;
    __syncthreads();
    // stable rank of every kept row among the kept rows; with R <= 64 the four waves split the comparisons of a row (row r = lane,
    // wave w compares with rows q = w mod 4) and add their partial ranks in LDS.  (Dropped rows keep -1: nobody adds to them, and the
    // partial ranks of a kept row are added to its 0.)
    const int nsplit = R <= 64 ? ARR_BLOCK / 64 : 1;
    for (int r = nsplit > 1 ? (tid & 63) : tid; r < R; r += nsplit > 1 ? R : ARR_BLOCK) {
        if (rank_acc[r] < 0) continue;
        const int32_t k = key[r];
        int rank = 0;
        for (int q = nsplit > 1 ? (tid >> 6) : 0; q < R; q += nsplit) {
            if (rank_acc[q] < 0) continue;                 // (a mark is -1, or >= 0 and only ever grows while the ranks are added: never mistaken)
            const int32_t kq = key[q];
            rank += (kq < k) || (kq == k && q < r);
        }
        if (nsplit > 1) atomicAdd(&rank_acc[r], rank); else rank_acc[r] = rank;
    }
    __syncthreads();
    for (int r = tid; r < R; r += ARR_BLOCK)
        if (rank_acc[r] >= 0 && rank_acc[r] < D_out) src[rank_acc[r]] = r;
    // rows kept (after the depth cut): the sorted rows are the first `kept` output rows, everything behind them is padding
    int32_t* kept_sh = rank_acc + R;                      // [1]
    if (tid < 64) {
        int kept = 0;
        for (int r = tid; r < R; r += 64) kept += rank_acc[r] >= 0;
        for (int o = 32; o > 0; o >>= 1) kept += __shfl_xor(kept, o);
        kept = kept < D_out ? kept : D_out;
        if (tid == 0) { *kept_sh = kept; if (depth_out) depth_out[n] = kept; }
    }
    __syncthreads();
    const size_t obase = (size_t)n * D_out * L;
    const int total = D_out * L, nreal = *kept_sh * L;
    const float inv_l = 1.0f / (float)L;                  // e / L for e < 2^22: (e + 0.5) / L is at least 0.5 / L away from an integer
    // ---- the kept rows: a gather of 4-byte elements (source rows are 132 bytes, any alignment) ----
    for (int e = tid; e < nreal; e += ARR_BLOCK) {
        const int d = (int)(((float)e + 0.5f) * inv_l), l = e - d * L;
        const size_t i = ibase + (size_t)src[d] * L + l;
        oseq[obase + e] = seq[i]; obq[obase + e] = bq[i]; omq[obase + e] = mq[i]; ohap[obase + e] = hap[i];
    }
    // ---- the padding behind them (write_to_bins.py:15-30: constant_values=-2; two thirds of the output at 30x: 30 reads in 90 rows): no
    // look-up, no load - 16-byte stores between a ragged head and tail (the four planes share one alignment: equal element offsets) ----
    const size_t p0 = obase + nreal, p1 = obase + total;
    const size_t a0 = (p0 + 3) & ~(size_t)3, a1 = p1 & ~(size_t)3;          // 16-byte aligned body [a0, a1) when the plane bases are
    const bool vec = ((((uintptr_t)oseq | (uintptr_t)obq | (uintptr_t)omq | (uintptr_t)ohap) & 15) == 0) && a0 < a1;
    if (vec) {
        const int4 m2 = int4{-2, -2, -2, -2};
        for (size_t q = a0 + 4 * (size_t)tid; q < a1; q += 4 * ARR_BLOCK) {
            *reinterpret_cast<int4*>(oseq + q) = m2; *reinterpret_cast<int4*>(obq + q) = m2;
            *reinterpret_cast<int4*>(omq + q) = m2; *reinterpret_cast<int4*>(ohap + q) = m2;
        }
        if (tid < 8) {                                     // at most three elements on either side
            const size_t q = tid < 4 ? p0 + tid : a1 + (tid - 4);
            const bool in = tid < 4 ? q < a0 : q < p1;
            if (in) { oseq[q] = -2; obq[q] = -2; omq[q] = -2; ohap[q] = -2; }
        }
    } else {
        for (size_t q = p0 + tid; q < p1; q += ARR_BLOCK) { oseq[q] = -2; obq[q] = -2; omq[q] = -2; ohap[q] = -2; }
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
    const size_t lds = (size_t)(2 * R + D_out + 1) * sizeof(int32_t);
    if (lds > 64 * 1024) return NSNP_ESHAPE;
    hipLaunchKernelGGL(k_hap_arrange, dim3((unsigned)N), dim3(ARR_BLOCK), lds, (hipStream_t)stream,
                       seq, bq, mq, hap, n_reads, R, L, D_out, oseq, obq, omq, ohap, depth);
    NSNP_HIP(ctx, hipGetLastError());
    return NSNP_OK;
}
