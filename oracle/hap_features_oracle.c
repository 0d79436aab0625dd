/*
 * oracle/hap_features_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE (see oracle.h).
 *
 * Restates HaplotypeModel/dataset_dev.py:
 *   get_base_freq         :11-22    counts of 1,2,3,4,-1 per column; freq = cnt / (sum + 1e-6)
 *   get_base_quality      :24-33    sum of baseq over rows with base X; mean = sum / (cnt + 1e-9)
 *   get_mapping_quality   :35-44    the same with mapq
 *   get_seq_baseq_mapq_feat :46-51  26 rows in the order freq[5] cnt[5] bq[4] bq_mean[4] mq[4] mq_mean[4]
 *   get_frequency_feature :55-87    four read sets: all rows; rows with any hap==1; any hap==2;
 *                                   any hap==3; an empty set contributes zeros
 *   TestDataset.__getitem__ :337-349 reference row appended -> [105][L]
 * Integer sums are exact (numpy sums int32 into int64); divisions are float64 as in numpy.
 */
#include "oracle.h"

#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static void feat26(const int32_t* seq, const int32_t* bq, const int32_t* mq,
                   const uint8_t* rowsel, int D, int L, double* out /*[26][L]*/)
{
    for (int l = 0; l < L; ++l) {
        int64_t cnt[5] = {0, 0, 0, 0, 0};   /* A C G T D */
        int64_t bqs[4] = {0, 0, 0, 0}, mqs[4] = {0, 0, 0, 0};
        for (int d = 0; d < D; ++d) {
            if (rowsel && !rowsel[d]) continue;
            int32_t s = seq[d * L + l];
            if (s >= 1 && s <= 4) { cnt[s - 1]++; bqs[s - 1] += bq[d * L + l]; mqs[s - 1] += mq[d * L + l]; }
            else if (s == -1) cnt[4]++;
        }
        /* total_cnt = A+C+G+T+D + 1e-6, summed left to right in float64 (dataset_dev.py:17) */
        double total = (double)(cnt[0] + cnt[1] + cnt[2] + cnt[3] + cnt[4]) + 1e-6;
        for (int k = 0; k < 5; ++k) out[k * L + l] = (double)cnt[k] / total;
        for (int k = 0; k < 5; ++k) out[(5 + k) * L + l] = (double)cnt[k];
        for (int k = 0; k < 4; ++k) out[(10 + k) * L + l] = (double)bqs[k];
        for (int k = 0; k < 4; ++k) out[(14 + k) * L + l] = (double)bqs[k] / ((double)cnt[k] + 1e-9);
        for (int k = 0; k < 4; ++k) out[(18 + k) * L + l] = (double)mqs[k];
        for (int k = 0; k < 4; ++k) out[(22 + k) * L + l] = (double)mqs[k] / ((double)cnt[k] + 1e-9);
    }
}

void orc_hap_features(const int32_t* seq, const int32_t* bq, const int32_t* mq,
                      const int32_t* hap, const int32_t* ref_row, int D, int L, double* out)
{
    uint8_t* sel = (uint8_t*)malloc((size_t)(D > 0 ? D : 1));
    feat26(seq, bq, mq, NULL, D, L, out);
    for (int g = 1; g <= 3; ++g) {
        int depth = 0;
        for (int d = 0; d < D; ++d) {
            int any = 0;
            for (int l = 0; l < L; ++l) if (hap[d * L + l] == g) { any = 1; break; }
            sel[d] = (uint8_t)any; depth += any;
        }
        double* o = out + (size_t)g * 26 * L;
        if (depth > 0) feat26(seq, bq, mq, sel, D, L, o);
        else memset(o, 0, sizeof(double) * 26 * (size_t)L);
    }
    for (int l = 0; l < L; ++l) out[(size_t)104 * L + l] = (double)ref_row[l];
    free(sel);
}

void orc_hap_features_batch(const int32_t* seq, const int32_t* bq, const int32_t* mq,
                            const int32_t* hap, const int32_t* ref_row, int64_t N, int D, int L,
                            float* out, int nthreads)
{
    if (nthreads <= 0) nthreads = 1;
    #pragma omp parallel for num_threads(nthreads) schedule(dynamic, 16)
    for (int64_t n = 0; n < N; ++n) {
        double* tmp = (double*)malloc(sizeof(double) * 105 * (size_t)L);
        size_t po = (size_t)n * (size_t)D * (size_t)L;
        orc_hap_features(seq + po, bq + po, mq + po, hap + po, ref_row + n * L, D, L, tmp);
        /* predict_dev.py:35-36: .type(torch.FloatTensor) */
        for (int i = 0; i < 105 * L; ++i) out[(size_t)n * 105 * L + i] = (float)tmp[i];
        free(tmp);
    }
}

/* H1/H2: read arrangement (create_pileup_haplotype.py:140-207, write_to_bins.py:15-61), ties stable */
void orc_hap_arrange(const int32_t* seq, const int32_t* bq, const int32_t* mq, const int32_t* hap,
                     int rows, int R, int L, int D_out,
                     int32_t* oseq, int32_t* obq, int32_t* omq, int32_t* ohap, int32_t* depth)
{
    (void)R;
    const int mid = L / 2;
    int* idx = (int*)malloc(sizeof(int) * (size_t)(rows > 0 ? rows : 1));
    int n = 0;
    for (int r = 0; r < rows; ++r) if (seq[r * L + mid] != 0) idx[n++] = r;      /* :145-149 */
    for (int a = 1; a < n; ++a) {                                                  /* stable insertion sort by HP at centre */
        int v = idx[a], b = a - 1;
        while (b >= 0 && hap[idx[b] * L + mid] > hap[v * L + mid]) { idx[b + 1] = idx[b]; --b; }
        idx[b + 1] = v;
    }
    for (int d = 0; d < D_out; ++d)
        for (int l = 0; l < L; ++l) {
            int32_t a = -2, b = -2, c = -2, h = -2;
            if (d < n) { const int i = idx[d] * L + l; a = seq[i]; b = bq[i]; c = mq[i]; h = hap[i]; }
            oseq[d * L + l] = a; obq[d * L + l] = b; omq[d * L + l] = c; ohap[d * L + l] = h;
        }
    if (depth) *depth = n < D_out ? n : D_out;
    free(idx);
}
