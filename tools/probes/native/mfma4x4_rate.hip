// Diagnostic: issue rate of v_mfma_f32_4x4x1_16B_f32 (with the CBSZ/ABID broadcast the persistent LSTM
// kernel uses) against v_mfma_f32_16x16x4_f32, register-only loops, one wave per SIMD on every CU.
// Prints shader cycles per instruction (s_memtime) and TFLOP/s.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));

template <int NACC, int KIND>
__global__ __launch_bounds__(256) void loop(float* out, int iters, unsigned long long* clk) {
  f4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = f4{0.f, 0.f, 0.f, 0.f};
  float a = threadIdx.x * 0.001f, b = 1.0f + blockIdx.x * 0.0001f;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) {
      if (KIND == 0) acc[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[i], 0, 0, 0);
      if (KIND == 1) acc[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[i], 4, 5, 0);
      if (KIND == 2) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}

template <int NACC, int KIND>
void run(const char* name, double flop_per_instr) {
  const int blocks = 256, iters = 20000;
  float* out; unsigned long long* clk;
  hipMalloc(&out, blocks * 256 * 4); hipMalloc(&clk, blocks * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  loop<NACC, KIND><<<blocks, 256>>>(out, iters, clk);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  loop<NACC, KIND><<<blocks, 256>>>(out, iters, clk);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(blocks);
  hipMemcpy(h.data(), clk, blocks * 8, hipMemcpyDeviceToHost);
  printf("%-28s NACC=%d: %.2f cycles per instruction, %.1f TFLOP/s\n", name, NACC,
         (double)h[0] / ((double)iters * NACC), (double)blocks * 4 * iters * NACC * flop_per_instr / ms / 1e9);
  hipFree(out); hipFree(clk);
}

int main() {
  run<1, 0>("4x4x1_16B", 512); run<2, 0>("4x4x1_16B", 512); run<4, 0>("4x4x1_16B", 512); run<8, 0>("4x4x1_16B", 512);
  run<2, 1>("4x4x1_16B cbsz4 abid5", 512); run<4, 1>("4x4x1_16B cbsz4 abid5", 512); run<8, 1>("4x4x1_16B cbsz4 abid5", 512);
  run<1, 2>("16x16x4", 2048); run<2, 2>("16x16x4", 2048); run<4, 2>("16x16x4", 2048);
  return 0;
}
