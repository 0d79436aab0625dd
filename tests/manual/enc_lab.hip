// Development bench of the column-encode kernel outside Python: synthetic G2 windows (the library's own generator), the product
// kernel timed with HIP events, optionally a candidate kernel (-DNSNP_ENC_CANDIDATE=\"file\") timed beside it and compared bit for bit.
// Both are also compared with the CPU oracle (orc_encode_columns) when a 4th argument is given.
//   tests/manual/build_enc_lab.sh [candidate.hip]
//   ./build_tmp/enc_lab <windows> <coverage> <iters> [check]
#include "../../nanosnp_amd/csrc/pileup_encode.hip"
#include "../../oracle/oracle.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

ScopedKernelTimer::ScopedKernelTimer(nsnp_ctx* c, int kernel, hipStream_t stream) : ctx(c), k(kernel), s(stream), stop_ev(nullptr), on(false) {}
ScopedKernelTimer::~ScopedKernelTimer() {}
extern "C" int64_t nsnp_synth_columns(uint64_t seed, int64_t M, double coverage, int max_depth, double het_rate, int window,
                                      uint8_t* ref, uint8_t* bases, int64_t cap, int64_t* col_off);
#ifdef NSNP_ENC_CANDIDATE
#include NSNP_ENC_CANDIDATE
#endif

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

int main(int argc, char** argv)
{
    const int64_t n_win = argc > 1 ? atoll(argv[1]) : 32768;
    const double cov = argc > 2 ? atof(argv[2]) : 30.0;
    const int iters = argc > 3 ? atoi(argv[3]) : 20;
    const int64_t M = 33 * n_win;
    std::vector<uint8_t> ref(M); std::vector<int64_t> off(M + 1);
    int64_t rc = nsnp_synth_columns(20260001, M, cov, 144, 0.02, 33, ref.data(), nullptr, 0, off.data());
    const int64_t cap = -rc - 16;
    std::vector<uint8_t> bases(cap + 16);
    rc = nsnp_synth_columns(20260001, M, cov, 144, 0.02, 33, ref.data(), bases.data(), cap, off.data());
    const int64_t nb = rc;
    uint8_t *d_b, *d_r, *d_f[2]; int64_t* d_o; int32_t *d_c[2], *d_d[2];
    CK(hipMalloc(&d_b, nb + 16)); CK(hipMalloc(&d_r, M)); CK(hipMalloc(&d_o, (M + 1) * 8));
    for (int v = 0; v < 2; ++v) { CK(hipMalloc(&d_c[v], M * 18 * 4)); CK(hipMalloc(&d_d[v], M * 4)); CK(hipMalloc(&d_f[v], M)); CK(hipMemset(d_c[v], 0xee, M * 18 * 4)); }
    CK(hipMemcpy(d_b, bases.data(), nb, hipMemcpyHostToDevice)); CK(hipMemcpy(d_r, ref.data(), M, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_o, off.data(), (M + 1) * 8, hipMemcpyHostToDevice));
    const AfThreshold af = make_af_threshold(0.12);
    const AfTable aft = make_af_table(af);
    const double alg = (double)nb + 73.0 * M + 8.0 * M;      // bases + counts/depth/flag out; the 8-byte offsets listed separately
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto time_it = [&](const char* name, auto launch) {
        for (int i = 0; i < 3; ++i) launch();
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        for (int i = 0; i < iters; ++i) launch();
        CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double us = ms * 1e3 / iters;
        printf("%-28s M = %lld cov %.0f: %8.1f us per launch (back to back)  %6.0f GB/s (bases + 73 B out)  %6.0f GB/s incl. offsets\n", name, (long long)M, cov, us,
               ((double)nb + 73.0 * M) / us / 1e3, alg / us / 1e3);
    };
    time_it("k_encode_columns", [&] {
        hipLaunchKernelGGL(k_encode_columns, dim3((unsigned)NSNP_CDIV(M, ENC_BLOCK)), dim3(ENC_BLOCK), 0, 0, d_b, d_o, d_r, M, af, aft, 6, d_c[0], d_d[0], d_f[0]);
    });
    if (argc > 4) {
        std::vector<int32_t> oc(M * 18), od(M), gc(M * 18), gd(M); std::vector<uint8_t> of(M), gf(M);
        orc_encode_columns(bases.data(), off.data(), ref.data(), M, 0.12, 6, oc.data(), od.data(), of.data());
        CK(hipMemcpy(gc.data(), d_c[0], M * 72, hipMemcpyDeviceToHost)); CK(hipMemcpy(gd.data(), d_d[0], M * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(gf.data(), d_f[0], M, hipMemcpyDeviceToHost));
        int64_t nb_ = 0;
        for (int64_t c = 0; c < M; ++c) nb_ += memcmp(&oc[c * 18], &gc[c * 18], 72) || od[c] != gd[c] || of[c] != gf[c];
        printf("product kernel vs CPU oracle: %lld of %lld columns differ\n", (long long)nb_, (long long)M);
    }
#ifdef NSNP_ENC_CANDIDATE
    time_it("candidate", [&] { launch_candidate(d_b, d_o, d_r, M, af, 6, d_c[1], d_d[1], d_f[1]); });
    CK(hipGetLastError());
    std::vector<int32_t> c0(M * 18), c1(M * 18), p0(M), p1(M); std::vector<uint8_t> f0(M), f1(M);
    CK(hipMemcpy(c0.data(), d_c[0], M * 72, hipMemcpyDeviceToHost)); CK(hipMemcpy(c1.data(), d_c[1], M * 72, hipMemcpyDeviceToHost));
    CK(hipMemcpy(p0.data(), d_d[0], M * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(p1.data(), d_d[1], M * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(f0.data(), d_f[0], M, hipMemcpyDeviceToHost)); CK(hipMemcpy(f1.data(), d_f[1], M, hipMemcpyDeviceToHost));
    if (argc > 4) {
        std::vector<int32_t> oc(M * 18), od(M); std::vector<uint8_t> of(M);
        orc_encode_columns(bases.data(), off.data(), ref.data(), M, 0.12, 6, oc.data(), od.data(), of.data());
        int64_t nb_ = 0;
        for (int64_t c = 0; c < M; ++c) nb_ += memcmp(&oc[c * 18], &c1[c * 18], 72) || od[c] != p1[c] || of[c] != f1[c];
        printf("candidate vs CPU oracle: %lld of %lld columns differ\n", (long long)nb_, (long long)M);
    }
    int64_t bad = 0, first = -1;
    for (int64_t c = 0; c < M; ++c) {
        if (memcmp(&c0[c * 18], &c1[c * 18], 72) || p0[c] != p1[c] || f0[c] != f1[c]) { if (first < 0) first = c; ++bad; }
    }
    printf("candidate vs product kernel: %lld of %lld columns differ%s\n", (long long)bad, (long long)M, bad ? "" : " (bit-identical)");
    if (bad) {
        printf("first differing column %lld: bytes '", (long long)first);
        fwrite(&bases[off[first]], 1, (size_t)(off[first + 1] - off[first]), stdout);
        printf("'\n  product  :"); for (int k = 0; k < 18; ++k) printf(" %d", c0[first * 18 + k]); printf("  depth %d flags %d\n", p0[first], f0[first]);
        printf("  candidate:"); for (int k = 0; k < 18; ++k) printf(" %d", c1[first * 18 + k]); printf("  depth %d flags %d\n", p1[first], f1[first]);
    }
    return bad ? 1 : 0;
#else
    return 0;
#endif
}
