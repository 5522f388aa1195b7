// Diagnostic: does a kernel of libcapnet_hip.so disturb the LDS or the registers of workgroups of
// ANOTHER kernel that share its CUs? A "canary" kernel (tiny LDS footprint, like the attention step
// kernels) fills its LDS and some registers with a pattern, re-checks them for a while and reports
// every word that changed, while conv1x1_fwd_bf16x6 (or the K-major kernel) loops on another stream.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../include/capnet.h"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__global__ __launch_bounds__(128) void canary(int iters, unsigned* report, int words) {
  extern __shared__ unsigned al[];
  const unsigned tag = 0xC0DE0000u + blockIdx.x;
  for (int i = threadIdx.x; i < words; i += 128) al[i] = tag ^ (unsigned)i;
  unsigned r0 = tag + 1, r1 = tag + 2, r2 = tag + 3, r3 = tag + 4;
  asm volatile("" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3));
  __syncthreads();
  unsigned bad_lds = 0, bad_reg = 0, first = 0xffffffffu, seen = 0;
  for (int it = 0; it < iters; ++it) {
    for (int i = threadIdx.x; i < words; i += 128) {
      const unsigned v = al[i];
      if (v != (tag ^ (unsigned)i)) { ++bad_lds; if (first == 0xffffffffu) { first = i; seen = v; } al[i] = tag ^ (unsigned)i; }
    }
    asm volatile("" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3));
    if (r0 != tag + 1 || r1 != tag + 2 || r2 != tag + 3 || r3 != tag + 4) ++bad_reg;
    __builtin_amdgcn_s_sleep(8);
  }
  if (bad_lds) { atomicAdd(&report[0], bad_lds); report[2] = first; report[3] = seen; }
  if (bad_reg) atomicAdd(&report[1], bad_reg);
}

int main(int argc, char** argv) {
  const int mode = argc > 1 ? atoi(argv[1]) : 1;   // 0: K-major f32 kernel, 1: bf16x6 bn 64, 2: bf16x6 bn 128
  const int B = 64, H = 28, Cin = 512, Cout = 128, M = B * H * H;
  float *x, *w, *y, *ps, *pq, *wk; unsigned *img, *rep;
  CK(hipMalloc(&x, (size_t)M * Cin * 4)); CK(hipMalloc(&w, (size_t)Cout * Cin * 4)); CK(hipMalloc(&y, (size_t)M * Cout * 4));
  CK(hipMalloc(&ps, (size_t)4096 * Cout * 4)); CK(hipMalloc(&pq, (size_t)4096 * Cout * 4)); CK(hipMalloc(&wk, (size_t)Cout * Cin * 4));
  CK(hipMalloc(&img, capnet_conv1x1_bf16x6_weight_words(Cin, Cout) * 4)); CK(hipMalloc(&rep, 16));
  std::vector<float> hx((size_t)M * Cin), hw((size_t)Cout * Cin);
  for (auto& v : hx) v = (float)rand() / RAND_MAX - 0.5f;
  for (auto& v : hw) v = ((float)rand() / RAND_MAX - 0.5f) * 0.1f;
  CK(hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
  hipStream_t lo, hi; int least, greatest;
  CK(hipDeviceGetStreamPriorityRange(&least, &greatest));
  CK(hipStreamCreateWithPriority(&lo, hipStreamNonBlocking, least)); CK(hipStreamCreateWithPriority(&hi, hipStreamNonBlocking, greatest));
  const int bn = mode == 2 ? 128 : 64;
  if (capnet_conv1x1_bf16x6_pack(w, img, Cout, Cin, bn, lo)) { printf("pack failed: %s\n", capnet_last_error()); return 1; }
  if (capnet_pack_conv_weight_kmajor(w, wk, Cout, Cin, 1, 1, Cin, lo)) { printf("pack2 failed\n"); return 1; }
  CK(hipStreamSynchronize(lo));
  for (int words : {200, 2048, 8192}) {
    CK(hipMemset(rep, 0, 16));
    for (int rnd = 0; rnd < 20; ++rnd) {
      for (int k = 0; k < 20; ++k) {
        int rc = mode == 0 ? capnet_conv2d_fwd_kmajor(x, (long)H * H * Cin, (long)H * Cin, Cin, wk, Cin, y, nullptr, nullptr, 0, ps, pq, B, H, H, Cin, Cout, 1, 1, 1, 0, 12864, nullptr, lo)
                           : capnet_conv1x1_fwd_bf16x6(x, (long)H * H * Cin, (long)H * Cin, Cin, img, bn, y, nullptr, nullptr, 0, ps, pq, B, H, H, Cin, Cout, 1, nullptr, nullptr, nullptr, 0, lo);
        if (rc) { printf("conv failed: %s\n", capnet_last_error()); return 1; }
      }
      hipLaunchKernelGGL(canary, dim3(512), dim3(128), words * 4, hi, 300, rep, words);
      CK(hipGetLastError());
    }
    CK(hipDeviceSynchronize());
    unsigned h[4];
    CK(hipMemcpy(h, rep, 16, hipMemcpyDeviceToHost));
    printf("mode %d, canary LDS %5d B: corrupted LDS words %u (first index %u, value 0x%08x), register mismatches %u\n", mode, words * 4, h[0], h[2], h[3], h[1]);
  }
  return 0;
}
