// Diagnostic: operand / result lane maps of v_mfma_f32_4x4x1_16B_f32 and its CBSZ/ABID broadcast,
// decoded from products of distinct primes-like values (a = lane + 1, b = 1000 * (lane + 1)).
// Expected (csrc/lstm_persist.hip relies on it): D[reg i][lane 4 bl + j] = A[lane 4 bl' + i] * B[lane 4 bl + j]
// with bl' = bl (cbsz 0) or ABID (cbsz 4).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
template <int CBSZ, int ABID>
__global__ void k(float* out) {
  const int l = threadIdx.x;
  f4 c = {0.f, 0.f, 0.f, 0.f};
  c = __builtin_amdgcn_mfma_f32_4x4x1f32((float)(l + 1), 1000.f * (l + 1), c, CBSZ, ABID, 0);
  for (int i = 0; i < 4; ++i) out[i * 64 + l] = c[i];
}
template <int CBSZ, int ABID>
int run(float* d) {
  float h[256];
  hipLaunchKernelGGL((k<CBSZ, ABID>), dim3(1), dim3(64), 0, 0, d);
  if (hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) return 1;
  int bad = 0;
  for (int i = 0; i < 4; ++i)
    for (int l = 0; l < 64; ++l) {
      const long v = (long)(h[i * 64 + l] + 0.5f);
      const int lb = (int)(v / 1000 / ((v % 1000) ? 1 : 1));  // decode below
      (void)lb;
      // v = (la+1) * 1000 * (lb+1): find the pair with lb = l (B is never broadcast here)
      const long q = v / (1000L * (l + 1));
      const int la = (int)q - 1;
      const int want = 4 * (CBSZ ? ABID : (l >> 2)) + i;
      if (v != (long)(la + 1) * 1000L * (l + 1) || la != want) {
        if (bad < 8) printf("cbsz %d abid %d: reg %d lane %d: value %ld -> A lane %d, expected %d\n", CBSZ, ABID, i, l, v, la, want);
        ++bad;
      }
    }
  printf("cbsz %d abid %2d: %s\n", CBSZ, ABID, bad ? "MISMATCH" : "ok (D[i][4bl+j] = A[4bl'+i] * B[4bl+j])");
  return bad;
}
int main() {
  float* d;
  if (hipMalloc(&d, 1024) != hipSuccess) return 1;
  int bad = run<0, 0>(d) + run<4, 0>(d) + run<4, 3>(d) + run<4, 15>(d);
  return bad ? 2 : 0;
}
