// Diagnostic: cost of a grid-wide barrier + a dependent cross-workgroup exchange on MI355X, i.e.
// what a persistent multi-step LSTM kernel would pay per time step instead of a kernel launch.
//   grid = 256 workgroups (one per CU, cooperative launch => co-resident), 256 threads each.
//   per round: every workgroup stores 512 B to a shared buffer, release-fences, arrives on a
//   monotonic counter (agent-scope atomic), spins (bounded) until all have arrived, acquire-
//   fences and reads 64 KB of what the others wrote (the h_{t-1} exchange of the LSTM step).
// Prints us per round with and without the 64 KB read. Every spin is bounded: a stuck barrier
// sets a flag and the kernel exits.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

// mode 0: everyone polls the arrival counter with acquire loads
// mode 1: relaxed polling of the counter, one acquire fence at the end
// mode 2: the last arriver publishes the round in a separate flag; the others poll the flag (relaxed)
__device__ __forceinline__ bool grid_barrier(unsigned* counter, unsigned target, int* stuck, int mode,
                                             unsigned round, unsigned nwg) {
  __threadfence();          // release: this thread's stores are visible device-wide
  __syncthreads();
  bool ok = true;
  if (threadIdx.x == 0) {
    int spins = 0;
    if (mode == 0) {
      __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > (1 << 22)) { *stuck = 1; ok = false; break; }
      }
    } else if (mode == 1) {
      __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
        __builtin_amdgcn_s_sleep(2);
        if (++spins > (1 << 22)) { *stuck = 1; ok = false; break; }
      }
    } else {
      unsigned* flag = counter + 32;   // separate 128-B line
      const unsigned prev = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
      if (prev == target - 1) {
        __hip_atomic_store(flag, round + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      } else {
        while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < round + 1) {
          __builtin_amdgcn_s_sleep(2);
          if (++spins > (1 << 22)) { *stuck = 1; ok = false; break; }
        }
      }
    }
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
  }
  __syncthreads();
  __threadfence();          // acquire side for the other threads of the workgroup
  return ok;
}

// mode 3: no cache-wide fences at all. The exchanged data is written with write-through stores and
// read with cache-bypassing loads (sc0 sc1), so the barrier only has to order them: wait for the
// stores to be acknowledged, arrive with a relaxed agent-scope atomic, poll a flag.
typedef float f32x4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ bool grid_barrier_nofence(unsigned* counter, unsigned target, int* stuck,
                                                     unsigned round) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  bool ok = true;
  if (threadIdx.x == 0) {
    unsigned* flag = counter + 32;
    const unsigned prev = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (prev == target - 1) {
      __hip_atomic_store(flag, round + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      int spins = 0;
      while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < round + 1) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > (1 << 22)) { *stuck = 1; ok = false; break; }
      }
    }
  }
  __syncthreads();
  return ok;
}
// agent-scope relaxed atomics: the compiler emits sc1 loads / write-through stores for them
__device__ __forceinline__ unsigned long long load_coherent(const float* p) {
  return __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void store_coherent(float* p, float v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ __launch_bounds__(256) void probe(float* buf, unsigned* counter, int* stuck, int rounds,
                                             int do_read, float* sink, int mode) {
  const int nwg = gridDim.x;
  float acc = 0.f;
  for (int r = 0; r < rounds; ++r) {
    if (*((volatile int*)stuck)) break;
    // each workgroup writes its 128 floats of "h" (double-buffered by round parity)
    float* cur = buf + (size_t)(r & 1) * nwg * 128;
    if (mode == 3) {
      if (threadIdx.x < 128) store_coherent(cur + blockIdx.x * 128 + threadIdx.x, (float)(r + blockIdx.x) + acc * 1e-30f);
      if (!grid_barrier_nofence(counter, (unsigned)(r + 1) * nwg, stuck, (unsigned)r)) break;
      if (do_read) {
        unsigned long long v[32];   // 64 KB per workgroup as 8-B coherent loads
#pragma unroll
        for (int q = 0; q < 32; ++q) v[q] = load_coherent(cur + 2 * ((threadIdx.x + 256 * q) % (nwg * 64)));
#pragma unroll
        for (int q = 0; q < 32; ++q) acc += __uint_as_float((unsigned)(v[q] & 0xffffffffu));
      }
      continue;
    }
    if (threadIdx.x < 128) cur[blockIdx.x * 128 + threadIdx.x] = (float)(r + blockIdx.x) + acc * 1e-30f;
    if (!grid_barrier(counter, (unsigned)(r + 1) * nwg, stuck, mode, (unsigned)r, (unsigned)nwg)) break;
    if (do_read) {
      // read 64 KB written by the others (16 x 16-B loads per thread), like staging h_{t-1}
      const float4* src = reinterpret_cast<const float4*>(cur);
      float4 v[16];
#pragma unroll
      for (int q = 0; q < 16; ++q) v[q] = src[(threadIdx.x + 256 * q) % (nwg * 32)];
#pragma unroll
      for (int q = 0; q < 16; ++q) acc += v[q].x + v[q].w;
    }
  }
  if (acc == 12345.678f) sink[0] = acc;
  if (threadIdx.x == 0) sink[1 + blockIdx.x] = acc;
}

int main() {
  int dev = 0;
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, dev));
  const int nwg = prop.multiProcessorCount < 256 ? prop.multiProcessorCount : 256;
  float *buf, *sink;
  unsigned* counter;
  int* stuck;
  CHECK(hipMalloc(&buf, 2 * nwg * 128 * sizeof(float)));
  CHECK(hipMalloc(&sink, (nwg + 1) * sizeof(float)));
  CHECK(hipMalloc(&counter, 64 * sizeof(unsigned)));
  CHECK(hipMalloc(&stuck, sizeof(int)));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  for (int mode = 0; mode < 4; ++mode)
  for (int do_read = 0; do_read < 2; ++do_read) {
    for (int rounds : {101, 1001}) {
      CHECK(hipMemset(counter, 0, 64 * sizeof(unsigned)));
      CHECK(hipMemset(stuck, 0, sizeof(int)));
      CHECK(hipMemset(buf, 0, 2 * nwg * 128 * sizeof(float)));
      int r = rounds, dr = do_read;
      int md = mode;
      void* args[] = {&buf, &counter, &stuck, &r, &dr, &sink, &md};
      CHECK(hipEventRecord(e0, 0));
      CHECK(hipLaunchCooperativeKernel((const void*)probe, dim3(nwg), dim3(256), args, 0, 0));
      CHECK(hipEventRecord(e1, 0));
      CHECK(hipEventSynchronize(e1));
      float ms = 0.f;
      CHECK(hipEventElapsedTime(&ms, e0, e1));
      int st = 0;
      CHECK(hipMemcpy(&st, stuck, sizeof(int), hipMemcpyDeviceToHost));
      std::vector<float> hs(nwg + 1);
      CHECK(hipMemcpy(hs.data(), sink, (nwg + 1) * sizeof(float), hipMemcpyDeviceToHost));
      printf("mode %d workgroups %d read %d rounds %4d: %.3f ms total, %.3f us/round, stuck %d, check %.1f\n", mode, nwg,
             do_read, rounds, ms, 1e3 * ms / rounds, st, hs[1]);
      fflush(stdout);
      if (st) return 2;
    }
  }
  return 0;
}
