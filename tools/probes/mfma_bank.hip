// Probe: does the VGPR bank of the A and B operand of v_mfma_f32_32x32x2_f32 matter?  A wave issues 32 MFMAs per iteration on four
// accumulators (the tile-image GEMM's chunk), operands in fixed physical registers:
//   SAME   A = v[64+j], B = v[80+j]   (both registers of a pair in the same bank: index mod 4 equal)
//   DIFF2  A = v[64+j], B = v[82+j]   (banks differ by 2)
//   DIFF1  A = v[64+j], B = v[81+j]   (banks differ by 1)
// One wave per SIMD x 1..4 workgroups per CU.  Reports TFLOP/s from HIP events.
#include <hip/hip_runtime.h>
#include <cstdio>

#define MFMA(acc, a, b) "v_mfma_f32_32x32x2_f32 v[" acc "], v" #a ", v" #b ", v[" acc "]\n"
#define ROW(a0, a1, b0, b1) MFMA("0:15", a0, b0) MFMA("16:31", a0, b1) MFMA("32:47", a1, b0) MFMA("48:63", a1, b1)

template <int VAR>
__global__ __launch_bounds__(256) void k(float* out, int iters)
{
    float r = 0.f;
    for (int it = 0; it < iters; ++it) {
        if (VAR == 0)
            asm volatile(ROW(64, 72, 80, 88) ROW(65, 73, 81, 89) ROW(66, 74, 82, 90) ROW(67, 75, 83, 91)
                         ROW(68, 76, 84, 92) ROW(69, 77, 85, 93) ROW(70, 78, 86, 94) ROW(71, 79, 87, 95)
                         ::: "v0","v1","v2","v3","v4","v5","v6","v7","v8","v9","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31",
                             "v32","v33","v34","v35","v36","v37","v38","v39","v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","v60","v61","v62","v63");
        else if (VAR == 1)
            asm volatile(ROW(64, 72, 82, 90) ROW(65, 73, 83, 91) ROW(66, 74, 84, 92) ROW(67, 75, 85, 93)
                         ROW(68, 76, 86, 94) ROW(69, 77, 87, 95) ROW(70, 78, 88, 96) ROW(71, 79, 89, 97)
                         ::: "v0","v1","v2","v3","v4","v5","v6","v7","v8","v9","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31",
                             "v32","v33","v34","v35","v36","v37","v38","v39","v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","v60","v61","v62","v63");
        else
            asm volatile(ROW(64, 72, 81, 89) ROW(65, 73, 82, 90) ROW(66, 74, 83, 91) ROW(67, 75, 84, 92)
                         ROW(68, 76, 85, 93) ROW(69, 77, 86, 94) ROW(70, 78, 87, 95) ROW(71, 79, 88, 96)
                         ::: "v0","v1","v2","v3","v4","v5","v6","v7","v8","v9","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31",
                             "v32","v33","v34","v35","v36","v37","v38","v39","v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","v60","v61","v62","v63");
    }
    asm volatile("v_mov_b32 %0, v0" : "=v"(r) :: "v64","v65","v66","v67","v68","v69","v70","v71","v72","v73","v74","v75","v76","v77","v78","v79","v80","v81","v82","v83","v84","v85","v86","v87","v88","v89","v90","v91","v92","v93","v94","v95","v96","v97");
    out[blockIdx.x * 256 + threadIdx.x] = r;
}

template <int VAR>
void run(const char* name, float* out, int grid)
{
    const int iters = 20000;
    hipLaunchKernelGGL((k<VAR>), dim3(grid), dim3(256), 0, 0, out, 100); (void)hipDeviceSynchronize();
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0); hipLaunchKernelGGL((k<VAR>), dim3(grid), dim3(256), 0, 0, out, iters); (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double flop = (double)grid * 4 * iters * 32 * 4096.0;
    printf("%-36s grid %4d: %8.2f ms  %6.1f TFLOP/s\n", name, grid, ms, flop / (ms * 1e-3) / 1e12);
}

int main()
{
    float* out; (void)hipMalloc(&out, 4096 * 256 * 4);
    for (int grid : {256, 512, 1024}) {
        run<0>("SAME  bank (A v64+j, B v80+j)", out, grid);
        run<1>("DIFF2 bank (A v64+j, B v82+j)", out, grid);
        run<2>("DIFF1 bank (A v64+j, B v81+j)", out, grid);
    }
    return 0;
}
