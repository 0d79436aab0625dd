// Does v_mfma_f32_16x16x32_f16 honour f16 subnormal inputs under hipcc's default float mode?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void k(float* out, float aval, float bval)
{
    h8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (_Float16)0.f; b[j] = (_Float16)0.f; }
    // A[row l&15][k=8(l>>4)+j], B[k][col l&15]: put aval at A[r][0], bval at B[0][c]
    if ((threadIdx.x >> 4) == 0) { a[0] = (_Float16)aval; b[0] = (_Float16)bval; }
    f4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    out[threadIdx.x * 4 + 0] = c[0];
}
int main()
{
    float* d; hipMalloc(&d, 64 * 4 * sizeof(float));
    float h[256];
    const float tests[][2] = {{1.0f, 1.0f}, {3.0e-6f, 1024.0f}, {5.96e-8f, 16384.0f}, {6.0e-5f, 2.0f}, {1.0e-6f, 1.0e-6f}};
    for (auto& t : tests) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, t[0], t[1]);
        hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
        printf("a=%g b=%g  f16(a)=%g f16(b)=%g  mfma=%g  expect=%g\n", t[0], t[1], (float)(_Float16)t[0], (float)(_Float16)t[1], h[0],
               (float)(_Float16)t[0] * (float)(_Float16)t[1]);
    }
    return 0;
}
