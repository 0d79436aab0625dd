// enc_classify_probe.hip -- VERDICT r5 "Next round" item 5: is a packed-byte classification (v_perm_b32 nibble look-ups, the class bits
// counted with v_dot4_u32_u8 against a ones vector) cheaper than pass 1 of k_encode_columns (every byte through a 16-byte LDS table row of
// packed 8-bit counters)?  Both forms below count the ten counted symbols (A C G T a c g t * #) of 32-byte columns and gather the bit mask
// of the construct openers (+ - ^), one lane per column, exactly the job of pass 1 (pileup_encode.hip: "pass 1").  The probe prints the
// time per 64 columns of either form on the same bytes (generator: the symbols of G2 at their frequencies) and checks that both give the
// same counts; the instruction counts per 4 input bytes are read off the ISA (hipcc -S; tools/probes/README in docs/rounds/r06.md).
//   hipcc --offload-arch=gfx950 -O3 -o enc_classify_probe tools/probes/enc_classify_probe.hip && ./enc_classify_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

__device__ __forceinline__ int byte_class(int b)
{
    switch (b) {
    case 'A': return 0; case 'C': return 1; case 'G': return 2; case 'T': return 3;
    case 'a': return 4; case 'c': return 5; case 'g': return 6; case 't': return 7;
    case '*': return 8; case '#': return 9; case '+': case '-': case '^': return 11;
    default: return 10;
    }
}

// ---- form A: the table walk of k_encode_columns' pass 1 (three words of packed 8-bit counters + opener flag per byte) ----
__global__ __launch_bounds__(256) void k_table(const uint32_t* __restrict__ text, int64_t n_cols, uint4* __restrict__ out)
{
    __shared__ uint4 tab[256];
    {
        const int t = threadIdx.x, cls = byte_class(t);
        uint4 r{0u, 0u, 0u, 0u};
        if (cls < 4) r.x = 1u << (8 * cls); else if (cls < 8) r.y = 1u << (8 * (cls - 4)); else if (cls < 10) r.z = 1u << (8 * (cls - 8)); else if (cls == 11) r.w = 1u;
        tab[t ^ ((t >> 2) & 12)] = r;
    }
    __syncthreads();
    const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (c >= n_cols) return;
    const uint4* p = reinterpret_cast<const uint4*>(text + c * 8);
    const uint4 v0 = p[0], v1 = p[1];
    const uint32_t w8[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
    uint32_t ax = 0, ay = 0, az = 0, sm = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        uint32_t w = w8[k];
        w ^= (w >> 2) & 0x0c0c0c0cu;
        const uint4 r0 = tab[w & 0xffu], r1 = tab[(w >> 8) & 0xffu], r2 = tab[(w >> 16) & 0xffu], r3 = tab[w >> 24];
        ax += r0.x + r1.x; ax += r2.x + r3.x;
        ay += r0.y + r1.y; ay += r2.y + r3.y;
        az += r0.z + r1.z; az += r2.z + r3.z;
        sm |= ((((r3.w << 1) | r2.w) << 2) | ((r1.w << 1) | r0.w)) << (4 * k);
    }
    out[c] = uint4{ax, ay, az, sm};
}

// ---- form B: nibble look-ups with v_perm_b32 (four bytes per instruction), class bits counted with v_dot4_u32_u8 ----
// class bits: byte0 = A C G T a c g t (bits 0..7), byte1 = * # + - ^ (bits 0..4).  lo-nibble tables have 16 entries (two perms + a select),
// hi-nibble tables 8 (pileup text is ASCII: bytes >= 0x80 would need one more select; left out in this form's favour)
__device__ __forceinline__ uint32_t perm8(uint32_t tab_hi, uint32_t tab_lo, uint32_t sel) { return __builtin_amdgcn_perm(tab_hi, tab_lo, sel); }
__device__ __forceinline__ uint32_t dot4(uint32_t a, uint32_t b, uint32_t acc) { return __builtin_amdgcn_udot4(a, b, acc, false); }

__global__ __launch_bounds__(256) void k_dot4(const uint32_t* __restrict__ text, int64_t n_cols, uint4* __restrict__ out)
{
    const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (c >= n_cols) return;
    const uint4* p = reinterpret_cast<const uint4*>(text + c * 8);
    const uint4 v0 = p[0], v1 = p[1];
    const uint32_t w8[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
    // lo-nibble -> letter bits: 1: A|a = 0x11, 3: C|c = 0x22, 4: T|t = 0x88, 7: G|g = 0x44     (entries 0..7 | 8..15)
    constexpr uint32_t LO0_A = 0x22001100u, LO0_B = 0x44000088u, LO0_C = 0u, LO0_D = 0u;      // bytes: n0 n1 n2 n3 | n4 n5 n6 n7 | n8.. | n12..
    // lo-nibble -> other bits: 3: # = 0x02, A: * = 0x01, B: + = 0x04, D: - = 0x08, E: ^ = 0x10
    constexpr uint32_t LO1_A = 0x02000000u, LO1_B = 0u, LO1_C = 0x04010000u, LO1_D = 0x00100800u;
    // hi-nibble -> letter bits: 4: A C G = 0x07, 5: T = 0x08, 6: a c g = 0x70, 7: t = 0x80;   other bits: 2: * # + - = 0x0f, 5: ^ = 0x10
    constexpr uint32_t HI0_A = 0u, HI0_B = 0x80700807u, HI1_A = 0x000f0000u, HI1_B = 0x00001000u;
    uint32_t cnt[10];
#pragma unroll
    for (int k = 0; k < 10; ++k) cnt[k] = 0;
    uint32_t sm = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const uint32_t w = w8[k];
        const uint32_t lo = w & 0x0f0f0f0fu, hi = (w >> 4) & 0x0f0f0f0fu;
        const uint32_t lo7 = lo & 0x07070707u;
        const uint32_t up = ((lo >> 3) & 0x01010101u) * 0xffu;                      // 0xff in the bytes whose nibble is 8..15
        const uint32_t l0 = (perm8(LO0_B, LO0_A, lo7) & ~up) | (perm8(LO0_D, LO0_C, lo7) & up);
        const uint32_t l1 = (perm8(LO1_B, LO1_A, lo7) & ~up) | (perm8(LO1_D, LO1_C, lo7) & up);
        const uint32_t c0 = l0 & perm8(HI0_B, HI0_A, hi);                            // one-hot letter class per byte
        const uint32_t c1 = l1 & perm8(HI1_B, HI1_A, hi);
#pragma unroll
        for (int b = 0; b < 8; ++b) cnt[b] = dot4((c0 >> b) & 0x01010101u, 0x01010101u, cnt[b]);
        cnt[8] = dot4(c1 & 0x01010101u, 0x01010101u, cnt[8]);
        cnt[9] = dot4((c1 >> 1) & 0x01010101u, 0x01010101u, cnt[9]);
        const uint32_t op = ((c1 >> 2) | (c1 >> 3) | (c1 >> 4)) & 0x01010101u;       // + - ^ : one bit per byte -> four mask bits
        sm |= (((op * 0x00204081u) >> 21) & 0xfu) << (4 * k);
    }
    out[c] = uint4{cnt[0] | (cnt[1] << 8) | (cnt[2] << 16) | (cnt[3] << 24), cnt[4] | (cnt[5] << 8) | (cnt[6] << 16) | (cnt[7] << 24),
                   cnt[8] | (cnt[9] << 8), sm};
}

int main(int argc, char** argv)
{
    const int64_t n_cols = argc > 1 ? atoll(argv[1]) : (int64_t)4 << 20;          // 32 bytes each
    std::vector<uint8_t> h((size_t)n_cols * 32);
    const char sym[] = "AAAAAAAACCCCCCCCGGGGGGGGTTTTTTTTaaaaaaaaccccccccggggggggtttttttt**##+-^$123ACGTNn.,";
    uint64_t s = 88172645463325252ull;
    for (auto& b : h) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; b = (uint8_t)sym[s % (sizeof(sym) - 1)]; }
    uint32_t* d_text; uint4 *d_a, *d_b;
    hipMalloc(&d_text, h.size()); hipMalloc(&d_a, n_cols * 16); hipMalloc(&d_b, n_cols * 16);
    hipMemcpy(d_text, h.data(), h.size(), hipMemcpyHostToDevice);
    const unsigned grid = (unsigned)((n_cols + 255) / 256);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms[2];
    for (int form = 0; form < 2; ++form) {
        for (int rep = 0; rep < 3; ++rep) { if (form == 0) hipLaunchKernelGGL(k_table, dim3(grid), dim3(256), 0, 0, d_text, n_cols, d_a); else hipLaunchKernelGGL(k_dot4, dim3(grid), dim3(256), 0, 0, d_text, n_cols, d_b); }
        hipEventRecord(e0);
        for (int rep = 0; rep < 20; ++rep) { if (form == 0) hipLaunchKernelGGL(k_table, dim3(grid), dim3(256), 0, 0, d_text, n_cols, d_a); else hipLaunchKernelGGL(k_dot4, dim3(grid), dim3(256), 0, 0, d_text, n_cols, d_b); }
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms[form], e0, e1); ms[form] /= 20;
    }
    std::vector<uint4> a(n_cols), b(n_cols);
    hipMemcpy(a.data(), d_a, n_cols * 16, hipMemcpyDeviceToHost); hipMemcpy(b.data(), d_b, n_cols * 16, hipMemcpyDeviceToHost);
    int64_t bad = 0;
    for (int64_t i = 0; i < n_cols; ++i) bad += a[i].x != b[i].x || a[i].y != b[i].y || a[i].z != b[i].z || a[i].w != b[i].w;
    const double bytes = (double)n_cols * 48;
    printf("%lld columns of 32 bytes: table walk %.1f us (%.0f GB/s of in + out), perm/dot4 %.1f us (%.0f GB/s): perm/dot4 takes %.2f x the table walk; %lld columns differ\n",
           (long long)n_cols, ms[0] * 1e3, bytes / ms[0] / 1e6, ms[1] * 1e3, bytes / ms[1] / 1e6, ms[1] / ms[0], (long long)bad);
    return bad != 0;
}
