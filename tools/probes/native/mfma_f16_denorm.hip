// Does v_mfma_f32_32x32x16_f16 keep fp16 subnormal inputs? (A = 2^-20: subnormal; B = 2^10; 16 products of 2^-10 -> 2^-6)
//   hipcc --offload-arch=gfx950 -O2 tools/native/mfma_f16_denorm.hip -o /tmp/mfma_f16_denorm && /tmp/mfma_f16_denorm
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void k(float* out, float a, float b) {
  f16x8 A, B;
  for (int i = 0; i < 8; ++i) { A[i] = (_Float16)a; B[i] = (_Float16)b; }
  f32x16 acc;
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B, acc, 0, 0, 0);
  if (threadIdx.x == 0) out[0] = acc[0];
}
int main() {
  float* d; hipMalloc(&d, 4);
  const float as[3] = {9.5367431640625e-07f /* 2^-20 */, 5.9604644775390625e-08f /* 2^-24: smallest subnormal */, 1.f};
  for (int i = 0; i < 3; ++i) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, as[i], 1024.f);
    float h; hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
    printf("a = %.6g  b = 1024: acc = %.9g (expected %.9g)\n", as[i], h, 16.0 * as[i] * 1024.0);
  }
  return 0;
}
