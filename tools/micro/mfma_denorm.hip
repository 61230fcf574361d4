// Does v_mfma_f32_16x16x32_f16 flush subnormal f16 INPUTS?  (The split-f16 GEMM tier multiplies lo parts a - f16(a), which are
// subnormal for small a.)  A = all x, B = all 1: every output should be 32 x.   hipcc --offload-arch=gfx950 -O2 -o mfma_denorm mfma_denorm.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(float x, float y, float* out) {
    half8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)x; b[i] = (_Float16)y; }
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    if (threadIdx.x == 0) out[0] = c[0];
}
int main() {
    float* d; hipMalloc(&d, 4);
    const float xs[] = {1.0f, 6.103515625e-05f /* 2^-14 min normal */, 3.0517578125e-05f /* 2^-15 subnormal */, 9.5367431640625e-07f /* 2^-20 */, 5.9604644775390625e-08f /* 2^-24 smallest */};
    for (float x : xs) for (float y : {1.0f, 1024.0f}) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, x, y, d);
        float h; hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
        printf("x = %.6e  y = %g: mfma = %.9e expected %.9e %s\n", x, y, h, 32.0 * x * y, h == 32.0f * x * y ? "ok" : "DIFFERENT");
    }
    // both subnormal
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, 3.0517578125e-05f, 3.0517578125e-05f, d);
    float h; hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
    printf("x = y = 2^-15: mfma = %.9e expected %.9e\n", h, 32.0 * 3.0517578125e-05 * 3.0517578125e-05);
    return 0;
}
