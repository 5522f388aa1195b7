// Diagnostic: what a per-XCD persistent LSTM kernel would pay per time step for its
// synchronisation. Batch rows are sharded over the 8 XCDs, the 32 CUs of an XCD hold all of W in
// registers, and per step every workgroup publishes its 8 x 16 slice of h and reads the XCD's
// 8 x 512 block: a barrier among the 32 workgroups of ONE XCD (counter and data stay in that
// XCD's L2, no agent-scope fence) plus a 16 KB read.
//   grid = 256 workgroups x 256 threads, cooperative launch. A workgroup reads its XCC id
//   (HW_REG_XCC_ID), takes a slot on that XCD, and after one chip-wide rendezvous (so that the
//   per-XCD populations are known) runs `rounds` rounds of: plain stores of 128 floats ->
//   s_waitcnt vmcnt(0) -> relaxed atomic add on the XCD's counter -> sc1-load poll -> sc1 loads
//   of the XCD's 16 KB. Every value read is checked against what its writer stored this round.
// Every spin is bounded; a stuck barrier sets a flag and every workgroup leaves.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

struct Ctl {
  unsigned slots[8 * 32];     // per-XCD slot counters, one 128-B line each
  unsigned arrive[8 * 32];    // per-XCD arrival counters
  unsigned all;               // chip-wide rendezvous
  int stuck;
  unsigned errors;
};

__device__ __forceinline__ unsigned ld_sc1(const unsigned* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float4 ld16_sc1(const float* p) {
  float4 v;
  asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return v;
}

__global__ __launch_bounds__(256) void probe(float* buf, Ctl* ctl, int rounds, int do_read, float* sink) {
  __shared__ unsigned s_xcd, s_slot, s_pop;
  if (threadIdx.x == 0) {
    unsigned x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    x &= 7;
    s_xcd = x;
    s_slot = atomicAdd(&ctl->slots[x * 32], 1u);
    // chip-wide rendezvous: everyone has taken its slot
    __hip_atomic_fetch_add(&ctl->all, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int spins = 0;
    while (ld_sc1(&ctl->all) < gridDim.x) {
      __builtin_amdgcn_s_sleep(2);
      if (++spins > (1 << 22)) { ctl->stuck = 1; break; }
    }
    s_pop = ld_sc1(&ctl->slots[x * 32]);
  }
  __syncthreads();
  const unsigned xcd = s_xcd, slot = s_slot, pop = s_pop;
  float* mine = buf + (size_t)xcd * 2 * 64 * 128;   // [parity][slot < 64][128 floats]
  unsigned* cnt = &ctl->arrive[xcd * 32];
  float acc = 0.f;
  unsigned bad = 0;
  for (int r = 0; r < rounds; ++r) {
    if (*((volatile int*)&ctl->stuck)) break;
    float* cur = mine + (size_t)(r & 1) * 64 * 128;
    if (threadIdx.x < 128) cur[slot * 128 + threadIdx.x] = (float)(r * 1000 + (int)slot) + acc * 1e-30f;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
      __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned target = (unsigned)(r + 1) * pop;
      int spins = 0;
      while (ld_sc1(cnt) < target) {
        if (++spins > (1 << 22)) { ctl->stuck = 1; break; }
      }
    }
    __syncthreads();
    if (do_read) {
      // pop x 128 floats (16 KB at 32 workgroups): 16-B sc1 loads, checked
      const int n16 = (int)pop * 32;
      for (int i = threadIdx.x; i < n16; i += 256) {
        const float4 v = ld16_sc1(cur + 4 * i);
        const float want = (float)(r * 1000 + i / 32);
        bad += (v.x != want) + (v.w != want);
        acc += v.y * 1e-30f;
      }
    }
  }
  if (bad) atomicAdd(&ctl->errors, bad);
  if (threadIdx.x == 0) sink[blockIdx.x] = acc + (float)pop;
}

int main() {
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int nwg = prop.multiProcessorCount < 256 ? prop.multiProcessorCount : 256;
  float *buf, *sink;
  Ctl* ctl;
  CHECK(hipMalloc(&buf, (size_t)8 * 2 * 64 * 128 * sizeof(float)));
  CHECK(hipMalloc(&sink, nwg * sizeof(float)));
  CHECK(hipMalloc(&ctl, sizeof(Ctl)));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  for (int do_read = 0; do_read < 2; ++do_read)
    for (int rounds : {101, 1001, 1001}) {
      CHECK(hipMemset(ctl, 0, sizeof(Ctl)));
      CHECK(hipMemset(buf, 0, (size_t)8 * 2 * 64 * 128 * sizeof(float)));
      int r = rounds, dr = do_read;
      void* args[] = {&buf, &ctl, &r, &dr, &sink};
      CHECK(hipEventRecord(e0, 0));
      CHECK(hipLaunchCooperativeKernel((const void*)probe, dim3(nwg), dim3(256), args, 0, 0));
      CHECK(hipEventRecord(e1, 0));
      CHECK(hipEventSynchronize(e1));
      float ms = 0.f;
      CHECK(hipEventElapsedTime(&ms, e0, e1));
      Ctl h;
      CHECK(hipMemcpy(&h, ctl, sizeof(Ctl), hipMemcpyDeviceToHost));
      printf("per-XCD barrier, %d workgroups, read %d, rounds %4d: %.3f us/round, stuck %d, wrong values %u, "
             "workgroups per XCD:", nwg, do_read, rounds, 1e3 * ms / rounds, h.stuck, h.errors);
      for (int x = 0; x < 8; ++x) printf(" %u", h.slots[x * 32]);
      printf("\n");
      fflush(stdout);
      if (h.stuck) return 2;
    }
  return 0;
}
