// Dev tool (tools/lds_poison_check.py): fills the LDS of every CU with NaN patterns (fp32 quiet NaN = bf16 NaN pair 0x7fc07fc0) so that a
// kernel which reads LDS it never wrote shows up as NaN / changed results.  extern "C" void lds_poison(void* stream): 4096 workgroups of
// 64 KB dynamic LDS each (two to three resident per CU: every CU's 160 KB is overwritten several times over).
#include <hip/hip_runtime.h>
#include <cstdint>
__global__ void k_poison(uint32_t pattern, uint32_t* sink)
{
    extern __shared__ uint32_t lds[];
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) lds[i] = pattern;
    __syncthreads();
    if (lds[(threadIdx.x * 37) & 16383] != pattern) sink[0] = 1;         // (keeps the stores alive)
}
extern "C" int lds_poison(void* stream, unsigned pattern)
{
    static uint32_t* sink = nullptr;
    if (!sink && hipMalloc(&sink, 4) != hipSuccess) return 1;
    static bool attr = false;
    if (!attr) { if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_poison), hipFuncAttributeMaxDynamicSharedMemorySize, 65536) != hipSuccess) return 2; attr = true; }
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(k_poison, dim3(4096), dim3(256), 65536, (hipStream_t)stream, pattern, sink);
    return hipGetLastError() == hipSuccess ? 0 : 3;
}
