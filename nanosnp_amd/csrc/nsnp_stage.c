/* nsnp_stage.c -- host side of the streamed stage-5 pipeline (nanosnp_amd/pipeline.py: stream_haplotype):
 *
 *   nsnp_stage_values   read planes of a haplotype site file (or of arrays in memory) -> a pinned staging buffer, all host threads at
 *                       once, optionally narrowed from the reference's int32 to int8 on the way (a quarter of the bytes over PCIe and
 *                       HBM; the feature kernel has an int8 entry that gives the same bits)
 *   nsnp_parse_ctg_pos  the fixed-width "ctg:pos" strings of a bin (candidate_positions S300 [N,1], haplotype_positions S300 [N,11])
 *                       -> contig ids + integer positions
 *
 * Reference counterparts: the HDF5 reads and the per-site Python loops of HaplotypeModel/dataset_dev.py:92-172 (PileupFeature /
 * HaplotypeFeature constructors: np.array(table_file.root.*), str(...).split(":"), int(pos)) behind a torch DataLoader with four
 * worker processes (predict_dev.py:31-32).
 */
#define _GNU_SOURCE
#include "nsnp_host.h"

#include <errno.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define STAGE_BLOCK (1 << 20)          /* source bytes per work item */
#define SCRATCH     (256 << 10)        /* per-thread scratch of the file + narrow path (stays in L2) */

static int pread_all(int fd, void* dst, int64_t n, int64_t off)
{
    char* d = (char*)dst;
    while (n > 0) {
        const ssize_t got = pread(fd, d, (size_t)n, (off_t)off);
        if (got < 0) { if (errno == EINTR) continue; return NSNP_HOST_EIO; }
        if (got == 0) return NSNP_HOST_EIO;              /* the file ends inside the array */
        d += got; off += got; n -= got;
    }
    return 0;
}

/* int32 -> int8; returns the number of values that do not survive the round trip */
static int64_t narrow_block(const int32_t* restrict s, int8_t* restrict d, int64_t n)
{
    int32_t bad = 0;
    for (int64_t i = 0; i < n; ++i) {
        const int32_t v = s[i];
        d[i] = (int8_t)v;
        bad |= (v + 128) & ~255;                         /* non-zero iff v is outside [-128, 127] */
    }
    if (!bad) return 0;
    int64_t c = 0;
    for (int64_t i = 0; i < n; ++i) c += (s[i] < -128 || s[i] > 127);
    return c;
}

/* int32 -> int16 likewise */
static int64_t narrow_block16(const int32_t* restrict s, int16_t* restrict d, int64_t n)
{
    int32_t bad = 0;
    for (int64_t i = 0; i < n; ++i) {
        const int32_t v = s[i];
        d[i] = (int16_t)v;
        bad |= (v + 32768) & ~65535;
    }
    if (!bad) return 0;
    int64_t c = 0;
    for (int64_t i = 0; i < n; ++i) c += (s[i] < -32768 || s[i] > 32767);
    return c;
}

int nsnp_stage_values(int fd, const void* src, int64_t src_off, int elem_src, int64_t n, void* dst, int elem_dst,
                      int64_t* n_out_of_range)
{
    if (n < 0 || src_off < 0 || (fd < 0 && !src && n) || (!dst && n)) return NSNP_HOST_EINVAL;
    if (!((elem_src == 4 && (elem_dst == 4 || elem_dst == 2 || elem_dst == 1)) || (elem_src == elem_dst && (elem_src == 1 || elem_src == 2)))) return NSNP_HOST_EINVAL;
    if (n_out_of_range) *n_out_of_range = 0;
    if (n == 0) return 0;
    const int narrow = elem_src == 4 && elem_dst < 4;
    const int64_t per = STAGE_BLOCK / elem_src;                          /* values per work item */
    const int64_t items = (n + per - 1) / per;
    int T = nsnp_host_threads();
    if (items < 2 || T < 1) T = 1;                   /* the whole team or one thread: libgomp ends the pooled threads a smaller team leaves out, and the next
                                                        full team pays for creating them again (measured: +1 ms per pass of the streamed pipelines) */
    int err = 0;
    int64_t bad_total = 0;
    #pragma omp parallel num_threads(T)
    {
        int32_t* scratch = NULL;
        int my_err = 0;
        int64_t my_bad = 0;
        if (narrow && fd >= 0) {
            scratch = (int32_t*)malloc(SCRATCH);
            if (!scratch) my_err = NSNP_HOST_ENOMEM;
        }
        #pragma omp for schedule(dynamic, 1)
        for (int64_t it = 0; it < items; ++it) {
            if (my_err) continue;
            const int64_t v0 = it * per, vn = (v0 + per <= n ? per : n - v0);
            if (!narrow) {
                char* d = (char*)dst + v0 * elem_dst;
                if (fd >= 0) my_err = pread_all(fd, d, vn * elem_src, src_off + v0 * elem_src);
                else memcpy(d, (const char*)src + src_off + v0 * elem_src, (size_t)(vn * elem_src));
            } else if (fd < 0) {
                const int32_t* sp = (const int32_t*)((const char*)src + src_off) + v0;
                my_bad += elem_dst == 1 ? narrow_block(sp, (int8_t*)dst + v0, vn) : narrow_block16(sp, (int16_t*)dst + v0, vn);
            } else {
                const int64_t sv = SCRATCH / 4;
                for (int64_t a = 0; a < vn && !my_err; a += sv) {
                    const int64_t m = a + sv <= vn ? sv : vn - a;
                    my_err = pread_all(fd, scratch, m * 4, src_off + (v0 + a) * 4);
                    if (!my_err) my_bad += elem_dst == 1 ? narrow_block(scratch, (int8_t*)dst + v0 + a, m)
                                                        : narrow_block16(scratch, (int16_t*)dst + v0 + a, m);
                }
            }
        }
        free(scratch);
        if (my_err) {
            #pragma omp atomic write
            err = my_err;
        }
        if (my_bad) {
            #pragma omp atomic
            bad_total += my_bad;
        }
    }
    if (n_out_of_range) *n_out_of_range = bad_total;
    return err;
}

/* Python's int(text) for a decimal position field: surrounding blanks, one sign, digits with single underscores BETWEEN digits
 * ("1_000" is 1000: PEP 515), at most 18 digits.  Returns 0 where int() raises ValueError (or the value would not fit). */
static int py_int(const uint8_t* p, const uint8_t* e, int64_t* out)
{
    while (p < e && (*p == ' ' || (*p >= 9 && *p <= 13))) ++p;
    while (e > p && (e[-1] == ' ' || (e[-1] >= 9 && e[-1] <= 13))) --e;
    int neg = 0;
    if (p < e && (*p == '+' || *p == '-')) { neg = *p == '-'; ++p; }
    if (p >= e) return 0;
    int64_t v = 0; int digits = 0, last_us = 1;                     /* last_us: an underscore may not come first */
    for (; p < e; ++p) {
        if (*p == '_') { if (last_us) return 0; last_us = 1; continue; }
        if (*p < '0' || *p > '9' || ++digits > 18) return 0;
        v = v * 10 + (*p - '0'); last_us = 0;
    }
    if (last_us) return 0;                                           /* ... nor last */
    *out = neg ? -v : v;
    return 1;
}

/* "ctg:pos" in a zero-padded field of `width` bytes: exactly one ':' (str.split(":") into two names, dataset_dev.py:109,153), the
 * position an optionally signed decimal integer (int(pos); surrounding blanks are accepted as int() accepts them).  The contig is
 * looked up among `n_names` names (blob + offsets); one that is not there gets id -1 - the reference's lookup then fails inside its
 * bare `except` and the row entry is 0 (dataset_dev.py:114-118). */
int nsnp_parse_ctg_pos(const uint8_t* rows, int64_t n, int width, const char* names_blob, const int64_t* names_off, int n_names,
                       int64_t* pos, int32_t* ctg)
{
    if (n < 0 || width <= 0 || (n && (!rows || !pos || !ctg)) || n_names < 0 || (n_names && (!names_blob || !names_off))) return NSNP_HOST_EINVAL;
    int err = 0;
#define BAD() do { _Pragma("omp atomic write") err = 1; } while (0)
    int T = nsnp_host_threads();
    if (n < 4 * 4096) T = 1;                         /* the whole team or one thread (see nsnp_stage_values) */
    #pragma omp parallel for num_threads(T) schedule(static)
    for (int64_t i = 0; i < n; ++i) {
        const uint8_t* r = rows + i * (int64_t)width;
        int len = width;
        while (len > 0 && r[len - 1] == 0) --len;                        /* numpy 'S' fields are zero-padded */
        const uint8_t* colon = (const uint8_t*)memchr(r, ':', (size_t)len);
        if (!colon || memchr(colon + 1, ':', (size_t)(r + len - colon - 1))) { BAD(); continue; }
        const int cl = (int)(colon - r);
        /* position */
        if (!py_int(colon + 1, r + len, &pos[i])) { BAD(); continue; }
        /* contig (the table is short: a bin holds one contig, a run a few dozen) */
        int id = -1;
        for (int k = 0; k < n_names; ++k) {
            const int64_t a = names_off[k], b = names_off[k + 1];
            if (b - a == cl && memcmp(names_blob + a, r, (size_t)cl) == 0) { id = k; break; }
        }
        ctg[i] = id;
    }
#undef BAD
    return err ? NSNP_HOST_EFORMAT : 0;
}

/* "ctg:pos:ref33" in a zero-padded field of `width` bytes (the `position` array of a .pd.bin: make_bin_predict_data.py:94-97), read as
 * PileupModel/dataset.py:127-132 reads it: strip, split at ':' into exactly three parts, int(pos), ord(seq[16]).  The contig is looked
 * up among n_names names; -1 when it is not there (the caller learns the name: a VCF row carries it). */
int nsnp_parse_ctg_pos_ref(const uint8_t* rows, int64_t n, int width, const char* names_blob, const int64_t* names_off, int n_names,
                           int64_t* pos, int32_t* ctg, uint8_t* ref_base)
{
    if (n < 0 || width <= 0 || (n && (!rows || !pos || !ctg || !ref_base)) || n_names < 0 || (n_names && (!names_blob || !names_off))) return NSNP_HOST_EINVAL;
    int err = 0;
#define BAD() do { _Pragma("omp atomic write") err = 1; } while (0)
    int T = nsnp_host_threads();
    if (n < 4 * 4096) T = 1;                         /* the whole team or one thread (see nsnp_stage_values) */
    #pragma omp parallel for num_threads(T) schedule(static)
    for (int64_t i = 0; i < n; ++i) {
        const uint8_t* r = rows + i * (int64_t)width;
        const uint8_t* e = r + width;
        while (e > r && e[-1] == 0) --e;                                 /* numpy 'S' fields are zero-padded */
        while (r < e && (*r == ' ' || (*r >= 9 && *r <= 13))) ++r;       /* .strip() */
        while (e > r && (e[-1] == ' ' || (e[-1] >= 9 && e[-1] <= 13))) --e;
        const uint8_t* c1 = (const uint8_t*)memchr(r, ':', (size_t)(e - r));
        const uint8_t* c2 = c1 ? (const uint8_t*)memchr(c1 + 1, ':', (size_t)(e - c1 - 1)) : NULL;
        if (!c2 || memchr(c2 + 1, ':', (size_t)(e - c2 - 1)) || e - (c2 + 1) < 17) { BAD(); continue; }
        if (!py_int(c1 + 1, c2, &pos[i])) { BAD(); continue; }
        ref_base[i] = c2[1 + 16];
        const int cl = (int)(c1 - r);
        int id = -1;
        for (int k = 0; k < n_names; ++k) {
            const int64_t a = names_off[k], b = names_off[k + 1];
            if (b - a == cl && memcmp(names_blob + a, r, (size_t)cl) == 0) { id = k; break; }
        }
        ctg[i] = id;
    }
#undef BAD
    return err ? NSNP_HOST_EFORMAT : 0;
}

/* out[i, c] = (float) x[i, row, channels[c]] for n staged windows [n, rows, width] of int16 or int32 values: the coverage slice of
 * PileupModel/predict.py:63 (x[:, 16, [0,1,2,3,9,10,11,12]] of the FloatTensor) taken from the staged pass on the host, so that no
 * D2H copy carries it (every value is a count: exact in float32). */
int nsnp_window_channels(const void* x, int elem, int64_t n, int rows, int width, int row, const int32_t* channels, int n_ch, float* out)
{
    if (n < 0 || (elem != 2 && elem != 4) || rows <= 0 || width <= 0 || row < 0 || row >= rows || n_ch <= 0 || !channels || (n && (!x || !out)))
        return NSNP_HOST_EINVAL;
    for (int c = 0; c < n_ch; ++c)
        if (channels[c] < 0 || channels[c] >= width) return NSNP_HOST_EINVAL;
    const int64_t stride = (int64_t)rows * width, base = (int64_t)row * width;
    int T = nsnp_host_threads();
    if (n < 2 * 8192) T = 1;                         /* the whole team or one thread (see nsnp_stage_values) */
    #pragma omp parallel for num_threads(T) schedule(static)
    for (int64_t i = 0; i < n; ++i) {
        float* o = out + i * n_ch;
        if (elem == 2) {
            const int16_t* r = (const int16_t*)x + i * stride + base;
            for (int c = 0; c < n_ch; ++c) o[c] = (float)r[channels[c]];
        } else {
            const int32_t* r = (const int32_t*)x + i * stride + base;
            for (int c = 0; c < n_ch; ++c) o[c] = (float)r[channels[c]];
        }
    }
    return 0;
}
