// Diagnostic: register-only v_mfma_f32_32x32x2_f32 loop on every CU; reports TFLOP/s and the
// in-kernel shader clock (s_memtime / s_memrealtime). Build: hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(256) void mfma_loop(float* out, int iters, unsigned long long* clk) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float a = threadIdx.x * 0.001f, b = 1.0f + blockIdx.x * 0.0001f;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0;
  for (int i = 0; i < NACC; ++i)
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int NACC>
void run(int blocks, int iters) {
  float* out; unsigned long long* clk;
  hipMalloc(&out, blocks * 256 * 4); hipMalloc(&clk, blocks * 16);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  mfma_loop<NACC><<<blocks, 256>>>(out, iters, clk);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  mfma_loop<NACC><<<blocks, 256>>>(out, iters, clk);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(2 * blocks);
  hipMemcpy(h.data(), clk, blocks * 16, hipMemcpyDeviceToHost);
  double fl = (double)blocks * 4 * iters * NACC * 4096.0;
  double mhz = (double)h[0] / (double)h[1] * 100.0;
  printf("NACC=%d blocks=%d iters=%d: %.3f ms  %.1f TFLOP/s  clock %.0f MHz\n", NACC, blocks, iters, ms, fl / ms / 1e9, mhz);
  hipFree(out); hipFree(clk);
}

int main() {
  run<4>(256, 20000);
  run<4>(512, 20000);
  run<4>(768, 10000);
  run<1>(256, 40000);
  run<4>(256 * 4, 100000);
  return 0;
}
