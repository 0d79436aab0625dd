// feat_skeleton.hip -- memory-pattern ceilings for k_hap_features (L = 33, D = 90, int32 planes): the kernel's loads and stores with
// (almost) no arithmetic, in the current mapping and in two alternatives.  Built and run by tools/feat_skeleton_probe.py.
#include <hip/hip_runtime.h>
#include <stdint.h>

constexpr int L = 33, NOUT = 105;

// A: the product kernel's mapping - one workgroup (4 waves) per site, wave w takes rows w, w + 4, ...; 33 lanes load one row of each
// plane (132 B); U row groups in flight; output rows of 33 floats written by 33 lanes
template <int U>
__global__ __launch_bounds__(256) void k_rows33(const int32_t* __restrict__ seq, const int32_t* __restrict__ bq, const int32_t* __restrict__ mq,
                                                const int32_t* __restrict__ hap, int D, float* __restrict__ out)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t base = (size_t)blockIdx.x * D * L;
    int acc = 0;
    if (lane < L) {
        for (int r0 = wave; r0 < D; r0 += 4 * U) {
            int v[U][4];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int r = r0 + 4 * u;
                const bool ok = r < D;
                const size_t i = base + (size_t)(ok ? r : 0) * L + lane;
                v[u][0] = seq[i]; v[u][1] = bq[i]; v[u][2] = mq[i]; v[u][3] = hap[i];
                if (!ok) v[u][0] = v[u][1] = v[u][2] = v[u][3] = 0;
            }
#pragma unroll
            for (int u = 0; u < U; ++u) acc += v[u][0] + v[u][1] * 3 + v[u][2] * 5 + v[u][3] * 7;
        }
    }
    __shared__ int red[4][64];
    red[wave][lane] = acc;
    __syncthreads();
    const int tot = red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane];
    float* o = out + (size_t)blockIdx.x * NOUT * L;
    if (lane < L)
        for (int row = wave; row < NOUT; row += 4) o[row * L + lane] = (float)(tot + row);
}

// B: four rows per wave load - 33 lanes x 16 B = 528 B = rows 4k .. 4k + 3 of a plane as one contiguous run; wave w takes the groups
// w, w + 4, ...; a lane's four elements are fixed (row, column) slots of the group; output written as a flat run with 8-byte stores
template <int U>
__global__ __launch_bounds__(256) void k_rows4x(const int32_t* __restrict__ seq, const int32_t* __restrict__ bq, const int32_t* __restrict__ mq,
                                                const int32_t* __restrict__ hap, int D, float* __restrict__ out, int flat_out)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t base = (size_t)blockIdx.x * D * L;
    const int G = (D + 3) / 4;                       // groups of four rows (the last may be short)
    const int total = D * L;
    int acc = 0;
    if (lane < L) {
        for (int g0 = wave; g0 < G; g0 += 4 * U) {
            int v[U][4][4];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int g = g0 + 4 * u;
                const int e = g * 4 * L + lane * 4;           // first element of this lane's 16 bytes
                const bool ok = g < G && e + 3 < total;
                const size_t i = base + (ok ? e : 0);
                // (16-byte loads at 4-byte alignment: global loads take any dword alignment)
                const int4 a = *reinterpret_cast<const int4*>(seq + i), b = *reinterpret_cast<const int4*>(bq + i);
                const int4 c = *reinterpret_cast<const int4*>(mq + i), d = *reinterpret_cast<const int4*>(hap + i);
                v[u][0][0] = a.x; v[u][0][1] = a.y; v[u][0][2] = a.z; v[u][0][3] = a.w;
                v[u][1][0] = b.x; v[u][1][1] = b.y; v[u][1][2] = b.z; v[u][1][3] = b.w;
                v[u][2][0] = c.x; v[u][2][1] = c.y; v[u][2][2] = c.z; v[u][2][3] = c.w;
                v[u][3][0] = d.x; v[u][3][1] = d.y; v[u][3][2] = d.z; v[u][3][3] = d.w;
                if (!ok) for (int p = 0; p < 4; ++p) for (int q = 0; q < 4; ++q) v[u][p][q] = 0;
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int q = 0; q < 4; ++q) acc += v[u][0][q] + v[u][1][q] * 3 + v[u][2][q] * 5 + v[u][3][q] * 7;
        }
    }
    __shared__ int red[4][64];
    red[wave][lane] = acc;
    __syncthreads();
    const int tot = red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane];
    float* o = out + (size_t)blockIdx.x * NOUT * L;
    if (flat_out) {
        // 3465 floats per site as a flat run, two per thread and trip (8-byte stores; the site base is 8-byte aligned for even sites only:
        // odd sites fall back to 4-byte stores)
        if ((blockIdx.x & 1) == 0) {
            for (int e = 2 * threadIdx.x; e + 1 < NOUT * L; e += 512) *reinterpret_cast<float2*>(o + e) = float2{(float)(tot + e), (float)(tot - e)};
            if (threadIdx.x == 0) o[NOUT * L - 1] = (float)tot;
        } else {
            for (int e = threadIdx.x; e < NOUT * L; e += 256) o[e] = (float)(tot + e);
        }
    } else if (lane < L) {
        for (int row = wave; row < NOUT; row += 4) o[row * L + lane] = (float)(tot + row);
    }
}

// C: loads only / stores only (what each side costs alone)
__global__ __launch_bounds__(256) void k_store_only(float* __restrict__ out)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* o = out + (size_t)blockIdx.x * NOUT * L;
    if (lane < L)
        for (int row = wave; row < NOUT; row += 4) o[row * L + lane] = (float)row;
}

extern "C" int feat_skeleton(int which, int U, const int32_t* seq, const int32_t* bq, const int32_t* mq, const int32_t* hap, int64_t N, int D, float* out, void* stream)
{
    hipStream_t s = (hipStream_t)stream;
    if (which == 0) {
        if (U == 2) hipLaunchKernelGGL(k_rows33<2>, dim3(N), dim3(256), 0, s, seq, bq, mq, hap, D, out);
        else if (U == 4) hipLaunchKernelGGL(k_rows33<4>, dim3(N), dim3(256), 0, s, seq, bq, mq, hap, D, out);
        else hipLaunchKernelGGL(k_rows33<6>, dim3(N), dim3(256), 0, s, seq, bq, mq, hap, D, out);
    } else if (which == 1 || which == 2) {
        if (U == 1) hipLaunchKernelGGL(k_rows4x<1>, dim3(N), dim3(256), 0, s, seq, bq, mq, hap, D, out, which == 2);
        else if (U == 2) hipLaunchKernelGGL(k_rows4x<2>, dim3(N), dim3(256), 0, s, seq, bq, mq, hap, D, out, which == 2);
        else hipLaunchKernelGGL(k_rows4x<3>, dim3(N), dim3(256), 0, s, seq, bq, mq, hap, D, out, which == 2);
    } else {
        hipLaunchKernelGGL(k_store_only, dim3(N), dim3(256), 0, s, out);
    }
    return (int)hipGetLastError();
}
