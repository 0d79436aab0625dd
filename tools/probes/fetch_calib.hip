// fetch_calib.hip -- calibration of rocprofv3's FETCH_SIZE for the access shapes of this repository's HBM-bound kernels
// (MI355X_MICROARCH.md, HBM: "FETCH_SIZE reports exactly 1/2 of the bytes of a wide coalesced streaming read (16 B/lane) ...
//  other access widths are uncalibrated: calibrate on a known byte count in your own access pattern").
// Every kernel streams the SAME buffer (default 2 GiB, far beyond the 256 MiB Infinity Cache) exactly once with one access shape:
//   k_calib_b128      16 B per lane, 64 lanes: 1 KiB contiguous per wave instruction         (the documented x2 case)
//   k_calib_b32        4 B per lane, 64 lanes: 256 B contiguous per wave instruction         (k_hap_features at L = 11: 5 rows x 11 lanes + idle)
//   k_calib_b32_33     4 B per lane, 33 of 64 lanes active: 132 B runs at a 132-byte stride  (k_hap_features at L = 33, int32 planes)
//   k_calib_b8_33      1 B per lane, 33 of 64 lanes: 33 B runs                               (k_hap_features_i8)
// Run under  rocprofv3 --kernel-trace --pmc FETCH_SIZE  (tools/fetch_calib.sh); the factor of a shape = bytes streamed / (FETCH_SIZE x 1024).
//   hipcc --offload-arch=gfx950 -O3 tools/probes/fetch_calib.hip -o build_tmp/fetch_calib && ./build_tmp/fetch_calib [GiB]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void k_calib_b128(const u32x4* __restrict__ p, size_t n16, unsigned* sink)
{
    unsigned acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) { const u32x4 v = p[i]; acc += v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345678u) *sink = acc;
}
__global__ __launch_bounds__(256) void k_calib_b32(const unsigned* __restrict__ p, size_t n4, unsigned* sink)
{
    unsigned acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) acc += p[i];
    if (acc == 0x12345678u) *sink = acc;
}
// k_hap_features' own shape: a workgroup streams one contiguous "site plane" of 90 rows x 33 elements at a time (wave w its rows
// w, w + 4, ..., lanes 33..63 idle), workgroups take consecutive planes
template <typename T>
__global__ __launch_bounds__(256) void k_calib_rows33(const T* __restrict__ p, size_t n_rows, unsigned* sink)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t n_planes = n_rows / 90;
    unsigned acc = 0;
    for (size_t pl = blockIdx.x; pl < n_planes; pl += gridDim.x) {
        const T* q = p + pl * (90 * 33);
        for (int r = wave; r < 90; r += 4) if (lane < 33) acc += (unsigned)q[r * 33 + lane];
    }
    if (acc == 0x12345678u) *sink = acc;
}

// k_hap_features at L = 11: a wave takes 5 consecutive rows of 11 elements per instruction (55 lanes, 220 contiguous bytes), the four
// waves of a workgroup interleave those groups over a plane of 90 rows
__global__ __launch_bounds__(256) void k_calib_rows11(const unsigned* __restrict__ p, size_t n_rows, unsigned* sink)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t n_planes = n_rows / 90;
    unsigned acc = 0;
    for (size_t pl = blockIdx.x; pl < n_planes; pl += gridDim.x) {
        const unsigned* q = p + pl * (90 * 11);
        for (int g = wave; g < 18; g += 4) if (lane < 55) acc += q[g * 55 + lane];
    }
    if (acc == 0x12345678u) *sink = acc;
}

int main(int argc, char** argv)
{
    const double gib = argc > 1 ? atof(argv[1]) : 2.0;
    const size_t bytes = (size_t)(gib * 1024.0 * 1024.0 * 1024.0) / 47520 * 47520;     // whole site planes (90 x 33 x 4 B = 11,880 B, 90 x 11 x 4 B = 3,960 B) and 16-B pieces
    void* buf; unsigned* sink;
    if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc((void**)&sink, 4) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(buf, 1, bytes);
    hipDeviceSynchronize();
    const int grid = 256 * 16;
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(k_calib_b128, dim3(grid), dim3(256), 0, 0, (const u32x4*)buf, bytes / 16, sink);
        hipLaunchKernelGGL(k_calib_b32, dim3(grid), dim3(256), 0, 0, (const unsigned*)buf, bytes / 4, sink);
        hipLaunchKernelGGL(k_calib_rows33<unsigned>, dim3(grid), dim3(256), 0, 0, (const unsigned*)buf, bytes / 132, sink);
        hipLaunchKernelGGL(k_calib_rows33<unsigned char>, dim3(grid), dim3(256), 0, 0, (const unsigned char*)buf, bytes / 33, sink);
        hipLaunchKernelGGL(k_calib_rows11, dim3(grid), dim3(256), 0, 0, (const unsigned*)buf, bytes / 44, sink);
    }
    hipDeviceSynchronize();
    printf("{\"bytes_streamed_per_launch\": %zu}\n", bytes);
    return 0;
}
