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
    if (depth_out && tid < 64) {
        int kept = 0;
        for (int r = tid; r < R; r += 64) kept += rank_acc[r] >= 0;
        for (int o = 32; o > 0; o >>= 1) kept += __shfl_xor(kept, o);
        if (tid == 0) depth_out[n] = kept < D_out ? kept : D_out;
    }
    __syncthreads();
    const size_t obase = (size_t)n * D_out * L;
    const int total = D_out * L;
    const float inv_l = 1.0f / (float)L;                  // e / L for e < 2^22: (e + 0.5) / L is at least 0.5 / L away from an integer
    // two output elements per thread and trip where the plane's base allows 8-byte stores (D_out x L even, or an even site): the
    // output - mostly -2 padding at 30x (30 reads in 90 rows) - is 3/4 of the kernel's bytes
    if (((obase & 1) == 0) && ((total & 1) == 0)) {
        for (int e = 2 * tid; e < total; e += 2 * ARR_BLOCK) {
            int32_t va[2], vb[2], vc[2], vh[2];
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int d = (int)(((float)(e + k) + 0.5f) * inv_l), l = (e + k) - d * L;
                const int r = src[d];
                va[k] = vb[k] = vc[k] = vh[k] = -2;             // write_to_bins.py:15-30: constant_values=-2
                if (r >= 0) {
                    const size_t i = ibase + (size_t)r * L + l;
                    va[k] = seq[i]; vb[k] = bq[i]; vc[k] = mq[i]; vh[k] = hap[i];
                }
            }
            *reinterpret_cast<int2*>(oseq + obase + e) = int2{va[0], va[1]};
            *reinterpret_cast<int2*>(obq + obase + e) = int2{vb[0], vb[1]};
            *reinterpret_cast<int2*>(omq + obase + e) = int2{vc[0], vc[1]};
            *reinterpret_cast<int2*>(ohap + obase + e) = int2{vh[0], vh[1]};
        }
        return;
    }
    for (int e = tid; e < total; e += ARR_BLOCK) {
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
    const size_t lds = (size_t)(2 * R + D_out) * sizeof(int32_t);
    if (lds > 64 * 1024) return NSNP_ESHAPE;
    hipLaunchKernelGGL(k_hap_arrange, dim3((unsigned)N), dim3(ARR_BLOCK), lds, (hipStream_t)stream,
                       seq, bq, mq, hap, n_reads, R, L, D_out, oseq, obq, omq, ohap, depth);
    NSNP_HIP(ctx, hipGetLastError());
    return NSNP_OK;
}
