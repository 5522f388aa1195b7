// Diagnostic: phase timing of the fused LSTM step (copies the kernel structure with s_memtime stamps).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cmath>
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int H = 512, KSTEPS = 64, LD = 68, ROWS = 64;

__global__ __launch_bounds__(256) void probe(const float* hprev, const float* Wcat, float* G, const float* cprev,
                                             float* c_out, float* h_out, int b, unsigned long long* stamps) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
  const int u0 = blockIdx.x * 8;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  const int kw0 = wave * (H / 4);
  float wreg[KSTEPS];
  {
    const int g = li >> 3, uu = li & 7;
    const float* wrow = Wcat + ((long)g * H + u0 + uu) * H + kw0;
#pragma unroll
    for (int j = 0; j < KSTEPS; j += 2) {
      const float4 v = *reinterpret_cast<const float4*>(wrow + 2 * j);
      wreg[j] = lh ? v.y : v.x; wreg[j + 1] = lh ? v.w : v.z;
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  {
    const int row = tid & 63; const float* src = hprev + (long)row * H;
#pragma unroll
    for (int it0 = 0; it0 < 32; it0 += 16) {
      float4 v[16];
#pragma unroll
      for (int q = 0; q < 16; ++q) v[q] = *reinterpret_cast<const float4*>(src + 4 * ((tid >> 6) + 4 * (it0 + q)));
#pragma unroll
      for (int q = 0; q < 16; ++q) { float* d = lds + (4 * ((tid >> 6) + 4 * (it0 + q))) * LD + row; d[0] = v[q].x; d[LD] = v[q].y; d[2*LD] = v[q].z; d[3*LD] = v[q].w; }
    }
  }
  __syncthreads();
  unsigned long long t2 = __builtin_amdgcn_s_memtime();
  f32x16 acc[2];
  for (int mt = 0; mt < 2; ++mt) for (int r = 0; r < 16; ++r) acc[mt][r] = 0.f;
  const float* As = lds + (kw0 + lh) * LD + li;
#pragma unroll
  for (int j = 0; j < KSTEPS; ++j) {
    acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(As[(2 * j) * LD], wreg[j], acc[0], 0, 0, 0);
    acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(As[(2 * j) * LD + 32], wreg[j], acc[1], 0, 0, 0);
  }
  __syncthreads();
  unsigned long long t3 = __builtin_amdgcn_s_memtime();
  float* red = lds;
  for (int mt = 0; mt < 2; ++mt) for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
    red[((wave * 2 + mt) * 32 + row) * 33 + li] = acc[mt][r];
  }
  __syncthreads();
  for (int o = tid; o < b * 8; o += 256) {
    const int row = o >> 3, uu = o & 7, mt = row >> 5, rr = row & 31;
    float pre[4];
    for (int g = 0; g < 4; ++g) {
      float s = G[(long)row * 4 * H + (long)g * H + u0 + uu];
      for (int w = 0; w < 4; ++w) s += red[((w * 2 + mt) * 32 + rr) * 33 + g * 8 + uu];
      pre[g] = s;
    }
    const float i = 1.f / (1.f + expf(-pre[0])), f = 1.f / (1.f + expf(-pre[1])), og = 1.f / (1.f + expf(-pre[2])), gt = tanhf(pre[3]);
    const float c = f * cprev[(long)row * H + u0 + uu] + i * gt;
    G[(long)row * 4 * H + u0 + uu] = i; G[(long)row * 4 * H + H + u0 + uu] = f; G[(long)row * 4 * H + 2 * H + u0 + uu] = og; G[(long)row * 4 * H + 3 * H + u0 + uu] = gt;
    c_out[(long)row * H + u0 + uu] = c; h_out[(long)row * H + u0 + uu] = og * c;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  unsigned long long t4 = __builtin_amdgcn_s_memtime();
  if (tid == 0) { unsigned long long* s = stamps + blockIdx.x * 5; s[0] = t1 - t0; s[1] = t2 - t1; s[2] = t3 - t2; s[3] = t4 - t3; s[4] = t4 - t0; }
}

int main() {
  float *h, *W, *G, *c, *co, *ho; unsigned long long* st;
  hipMalloc(&h, 64 * H * 4); hipMalloc(&W, 4 * H * H * 4); hipMalloc(&G, 64 * 4 * H * 4); hipMalloc(&c, 64 * H * 4);
  hipMalloc(&co, 64 * H * 4); hipMalloc(&ho, 64 * H * 4); hipMalloc(&st, 64 * 5 * 8);
  hipMemset(h, 0, 64 * H * 4); hipMemset(W, 0, 4 * H * H * 4); hipMemset(G, 0, 64 * 4 * H * 4); hipMemset(c, 0, 64 * H * 4);
  size_t ldsb = (size_t)H * LD * 4;
  hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    for (int i = 0; i < 50; ++i) probe<<<64, 256, ldsb>>>(h, W, G, c, co, ho, 64, st);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> s(64 * 5); hipMemcpy(s.data(), st, 64 * 5 * 8, hipMemcpyDeviceToHost);
    double a[5] = {0};
    for (int b = 0; b < 64; ++b) for (int k = 0; k < 5; ++k) a[k] += s[b * 5 + k] / 64.0;
    printf("avg launch %.2f us | cycles(100MHz ticks?): wload %.0f hstage %.0f mfma %.0f epi %.0f total %.0f\n", ms * 1e3 / 50, a[0], a[1], a[2], a[3], a[4]);
  }
  return 0;
}
