// Diagnostic: do f32 MFMAs of one wave overlap with VALU / LDS work of another wave on the same SIMD?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

// mode bit0: waves 0-3 run MFMA loop; bit1: waves 4-7 run VALU loop; bit2: waves 4-7 run LDS write/read loop
__global__ __launch_bounds__(512) void k(float* out, int iters, int mode) {
  __shared__ float lds[8192];
  const int wave = threadIdx.x >> 6;
  float res = 0;
  if (wave < 4) {
    if (mode & 1) {
      f32x16 acc[4];
      for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
      float a = threadIdx.x * 0.001f, b = 1.0f;
      for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
      }
      for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) res += acc[i][r];
    }
  } else {
    if (mode & 2) {
      float x0 = threadIdx.x, x1 = 1.f, x2 = 2.f, x3 = 3.f, x4 = 4.f, x5 = 5.f, x6 = 6.f, x7 = 7.f;
      for (int it = 0; it < iters; ++it) {   // 32 independent-ish fma per iteration (4 x 8)
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          x0 = fmaf(x0, 1.0001f, 0.5f); x1 = fmaf(x1, 1.0001f, 0.5f); x2 = fmaf(x2, 1.0001f, 0.5f); x3 = fmaf(x3, 1.0001f, 0.5f);
          x4 = fmaf(x4, 1.0001f, 0.5f); x5 = fmaf(x5, 1.0001f, 0.5f); x6 = fmaf(x6, 1.0001f, 0.5f); x7 = fmaf(x7, 1.0001f, 0.5f);
        }
      }
      res = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
    }
    if (mode & 4) {
      float x = threadIdx.x;
      for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          lds[(threadIdx.x + u * 512) & 8191] = x;
          x += lds[(threadIdx.x * 3 + u * 64 + it) & 8191];
        }
      }
      res += x;
    }
  }
  out[blockIdx.x * 512 + threadIdx.x] = res;
}

int main() {
  float* out; hipMalloc(&out, 256 * 512 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 20000;
  for (int mode : {1, 2, 3, 4, 5}) {
    k<<<256, 512>>>(out, iters, mode); hipDeviceSynchronize();
    hipEventRecord(e0); k<<<256, 512>>>(out, iters, mode); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("mode %d: %.3f ms  (mfma-only ideal %.3f ms)\n", mode, ms, iters * 4 * 64 / 2.4e6);
  }
  return 0;
}
