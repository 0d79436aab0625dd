// nsnp_devclock.hpp -- DIAGNOSTIC BUILD ONLY (-DNSNP_DEV_CLOCK, tools/build_variant.sh): the shader clock a kernel actually runs at.
// The chip lowers its clock under load (MI355X_MICROARCH.md, "DVFS give-back"), so a fraction of the 2.4 GHz peak mixes two things:
// cycles the matrix pipe idles and cycles the chip never ran.  Thread 0 of every workgroup stamps s_memtime (shader cycles) and
// s_memrealtime (100 MHz) at its first and last instruction and adds both differences to a table no other code reads
// (give-back item 6); clock = 0.1 GHz * sum(cycles) / sum(ticks).  The product build has none of this.
#pragma once
#ifdef NSNP_DEV_CLOCK
#include <hip/hip_runtime.h>
#define NSNP_DEVCLK_SLOTS 8
__device__ unsigned long long nsnp_devclk_acc[NSNP_DEVCLK_SLOTS][3];      // per slot: shader cycles, 100 MHz ticks, workgroups
__device__ unsigned long long nsnp_devclk_ext[NSNP_DEVCLK_SLOTS][4];
#define NSNP_DEVCLK_TRACE 4096
__device__ unsigned long long nsnp_devclk_trace[NSNP_DEVCLK_SLOTS][NSNP_DEVCLK_TRACE][4];   // per workgroup (of the last launch): start tick, end tick, cycles, HW_ID | XCC_ID << 32      // per slot: ~min start tick, max end tick, max and ~min cycles of one workgroup
struct DevClock {
    unsigned long long t0, r0;
    __device__ __forceinline__ void start() { t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
    __device__ __forceinline__ void stop(int slot)
    {
        if (threadIdx.x == 0) {
            const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
            if (t1 - t0 >= (1ull << 40) || r1 - r0 >= (1ull << 40)) return;                  // (a counter that wrapped: not a sample)
            atomicAdd(&nsnp_devclk_acc[slot][0], t1 - t0); atomicAdd(&nsnp_devclk_acc[slot][1], r1 - r0); atomicAdd(&nsnp_devclk_acc[slot][2], 1ull);
            {
                const unsigned wg = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
                if (wg < NSNP_DEVCLK_TRACE) {
                    unsigned long long* tr = nsnp_devclk_trace[slot][wg];
                    tr[0] = r0; tr[1] = r1; tr[2] = t1 - t0;
                    tr[3] = (unsigned long long)__builtin_amdgcn_s_getreg(0xf804) | ((unsigned long long)__builtin_amdgcn_s_getreg(0xf814) << 32);   // HW_ID, XCC_ID
                }
            }
            atomicMax(&nsnp_devclk_ext[slot][0], ~r0); atomicMax(&nsnp_devclk_ext[slot][1], r1);
            atomicMax(&nsnp_devclk_ext[slot][2], t1 - t0); atomicMax(&nsnp_devclk_ext[slot][3], ~(t1 - t0));
        }
    }
};
#define NSNP_DEVCLK_START DevClock devclk_; devclk_.start();
#define NSNP_DEVCLK_STOP(slot) devclk_.stop(slot);
// each translation unit with stamped kernels exports its own reader (no relocatable device code): out[slot][3] then ext[slot][4] (56 words), then zeroes the tables
#define NSNP_DEVCLK_READER(name)                                                                                      \
    extern "C" int name##_trace(int slot, unsigned long long* out)                                                    \
    {                                                                                                                  \
        if (hipDeviceSynchronize() != hipSuccess) return -1;                                                           \
        return hipMemcpyFromSymbol(out, HIP_SYMBOL(nsnp_devclk_trace), sizeof(unsigned long long) * NSNP_DEVCLK_TRACE * 4, \
                                   sizeof(unsigned long long) * NSNP_DEVCLK_TRACE * 4 * slot) == hipSuccess ? 0 : -1;       \
    }                                                                                                                  \
    extern "C" int name(unsigned long long* out)                                                                      \
    {                                                                                                                  \
        unsigned long long z[NSNP_DEVCLK_SLOTS][3] = {};                                                               \
        if (hipDeviceSynchronize() != hipSuccess) return -1;                                                           \
        if (hipMemcpyFromSymbol(out, HIP_SYMBOL(nsnp_devclk_acc), sizeof(z)) != hipSuccess) return -1;                 \
        if (hipMemcpyToSymbol(HIP_SYMBOL(nsnp_devclk_acc), z, sizeof(z)) != hipSuccess) return -1;                     \
        unsigned long long e[NSNP_DEVCLK_SLOTS][4] = {};                                                               \
        if (hipMemcpyFromSymbol(out + NSNP_DEVCLK_SLOTS * 3, HIP_SYMBOL(nsnp_devclk_ext), sizeof(e)) != hipSuccess) return -1; \
        return hipMemcpyToSymbol(HIP_SYMBOL(nsnp_devclk_ext), e, sizeof(e)) == hipSuccess ? 0 : -1;                    \
    }
#else
#define NSNP_DEVCLK_START
#define NSNP_DEVCLK_STOP(slot)
#define NSNP_DEVCLK_READER(name)
#endif
