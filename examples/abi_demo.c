/*
 * examples/abi_demo.c -- the C ABI of include/nanosnp.h driven from plain C: no Python, no PyTorch, only the HIP runtime
 * for device memory.  Synthetic 30x windows -> column encode -> PileupModel forward on the count matrix -> argmax / max.
 *
 *   gcc -std=c11 -O2 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude examples/abi_demo.c -o abi_demo \
 *       -Lnanosnp_amd -lnanosnp_hip -lnanosnp_host -L/opt/rocm/lib -lamdhip64 -lm \
 *       -Wl,-rpath,$PWD/nanosnp_amd -Wl,-rpath,/opt/rocm/lib
 *   ./abi_demo weights.f32 1024        # weights.f32: the 24 tensors of ont_pileup.chkpt, fp32, concatenated in order
 *
 * Prints one line per site: index, genotype class, zygosity class, the two maxima - what PileupModel/predict.py:52-60 derives
 * from model.predict (tests/test_gpu_abi_demo.py compares the lines with the Python binding on the same input).
 */
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "nanosnp.h"
#include "nsnp_host.h"

#define HIPCHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
#define NSNP(x) do { int rc_ = (x); if (rc_ != 0) { const char* t_ = 0; nsnp_last_hip_error(ctx, &t_); \
    fprintf(stderr, "%s -> %d (%s) %s\n", #x, rc_, nsnp_strerror(rc_), t_ ? t_ : ""); return 3; } } while (0)

/* element counts of the 24 tensors (include/nanosnp.h, nsnp_pileup_load_weights) */
static const int W_ELEMS[24] = { 256 * 18, 256 * 64, 256, 256, 256 * 18, 256 * 64, 256, 256,
                                 256 * 128, 256 * 64, 256, 256, 256 * 128, 256 * 64, 256, 256,
                                 128 * 128, 128, 256 * 128, 256, 21 * 256, 21, 3 * 256, 3 };

int main(int argc, char** argv)
{
    if (argc < 2) { fprintf(stderr, "usage: %s weights.f32 [n_windows]\n", argv[0]); return 1; }
    const int64_t n = argc > 2 ? atoll(argv[2]) : 1024;
    const int64_t m = n * 33;

    /* weights */
    size_t total = 0;
    for (int i = 0; i < 24; ++i) total += (size_t)W_ELEMS[i];
    float* wbuf = (float*)malloc(total * sizeof(float));
    FILE* f = fopen(argv[1], "rb");
    if (!f || fread(wbuf, sizeof(float), total, f) != total) { fprintf(stderr, "cannot read %zu floats from %s\n", total, argv[1]); return 1; }
    fclose(f);
    const float* tensors[24];
    { size_t o = 0; for (int i = 0; i < 24; ++i) { tensors[i] = wbuf + o; o += (size_t)W_ELEMS[i]; } }

    /* synthetic columns on the host (generator G2: stand-alone 33-column windows, 30x) */
    uint8_t* ref = (uint8_t*)malloc((size_t)m);
    int64_t* off = (int64_t*)malloc(sizeof(int64_t) * (size_t)(m + 1));
    int64_t need = nsnp_synth_columns(4242, m, 30.0, 144, 0.02, 33, ref, NULL, 0, off);
    if (need >= 0) { fprintf(stderr, "unexpected sizing result\n"); return 1; }
    const int64_t cap = -need;
    uint8_t* bases = (uint8_t*)malloc((size_t)cap);
    const int64_t nbytes = nsnp_synth_columns(4242, m, 30.0, 144, 0.02, 33, ref, bases, cap, off);
    if (nbytes < 0) { fprintf(stderr, "nsnp_synth_columns failed: %lld\n", (long long)nbytes); return 1; }

    nsnp_ctx* ctx = NULL;
    { int rc = nsnp_ctx_create(0, &ctx); if (rc) { fprintf(stderr, "nsnp_ctx_create -> %d (%s)\n", rc, nsnp_strerror(rc)); return 3; } }
    NSNP(nsnp_pileup_load_weights(ctx, tensors, 24));
    NSNP(nsnp_ctx_reserve(ctx, n));

    uint8_t *d_bases, *d_ref, *d_flags, *d_ga, *d_za;
    int64_t *d_off, *d_centers;
    int32_t *d_counts, *d_depth;
    float *d_gt, *d_zy, *d_gm, *d_zm;
    HIPCHECK(hipMalloc((void**)&d_bases, (size_t)nbytes + 16)); HIPCHECK(hipMalloc((void**)&d_ref, (size_t)m));
    HIPCHECK(hipMalloc((void**)&d_off, sizeof(int64_t) * (size_t)(m + 1))); HIPCHECK(hipMalloc((void**)&d_flags, (size_t)m));
    HIPCHECK(hipMalloc((void**)&d_counts, sizeof(int32_t) * 18 * (size_t)m)); HIPCHECK(hipMalloc((void**)&d_depth, sizeof(int32_t) * (size_t)m));
    HIPCHECK(hipMalloc((void**)&d_centers, sizeof(int64_t) * (size_t)n));
    HIPCHECK(hipMalloc((void**)&d_gt, sizeof(float) * 21 * (size_t)n)); HIPCHECK(hipMalloc((void**)&d_zy, sizeof(float) * 3 * (size_t)n));
    HIPCHECK(hipMalloc((void**)&d_gm, sizeof(float) * (size_t)n)); HIPCHECK(hipMalloc((void**)&d_zm, sizeof(float) * (size_t)n));
    HIPCHECK(hipMalloc((void**)&d_ga, (size_t)n)); HIPCHECK(hipMalloc((void**)&d_za, (size_t)n));
    HIPCHECK(hipMemcpy(d_bases, bases, (size_t)nbytes, hipMemcpyHostToDevice));
    HIPCHECK(hipMemcpy(d_ref, ref, (size_t)m, hipMemcpyHostToDevice));
    HIPCHECK(hipMemcpy(d_off, off, sizeof(int64_t) * (size_t)(m + 1), hipMemcpyHostToDevice));
    int64_t* centers = (int64_t*)malloc(sizeof(int64_t) * (size_t)n);
    for (int64_t i = 0; i < n; ++i) centers[i] = i * 33 + 16;             /* every window's centre column */
    HIPCHECK(hipMemcpy(d_centers, centers, sizeof(int64_t) * (size_t)n, hipMemcpyHostToDevice));

    hipStream_t s;
    HIPCHECK(hipStreamCreate(&s));
    NSNP(nsnp_pileup_encode_columns(ctx, d_bases, d_off, d_ref, m, 0.12, 6, d_counts, d_depth, d_flags, s));
    NSNP(nsnp_pileup_forward_windows(ctx, d_counts, d_centers, n, d_gt, d_zy, s));
    NSNP(nsnp_pileup_postprocess(ctx, d_gt, d_zy, NULL, n, d_ga, d_za, d_gm, d_zm, NULL, s));
    HIPCHECK(hipStreamSynchronize(s));

    uint8_t* ga = (uint8_t*)malloc((size_t)n); uint8_t* za = (uint8_t*)malloc((size_t)n);
    float* gm = (float*)malloc(sizeof(float) * (size_t)n); float* zm = (float*)malloc(sizeof(float) * (size_t)n);
    HIPCHECK(hipMemcpy(ga, d_ga, (size_t)n, hipMemcpyDeviceToHost)); HIPCHECK(hipMemcpy(za, d_za, (size_t)n, hipMemcpyDeviceToHost));
    HIPCHECK(hipMemcpy(gm, d_gm, sizeof(float) * (size_t)n, hipMemcpyDeviceToHost)); HIPCHECK(hipMemcpy(zm, d_zm, sizeof(float) * (size_t)n, hipMemcpyDeviceToHost));
    for (int64_t i = 0; i < n; ++i) printf("%lld\t%d\t%d\t%.6f\t%.6f\n", (long long)i, ga[i], za[i], gm[i], zm[i]);
    nsnp_ctx_destroy(ctx);
    return 0;
}
