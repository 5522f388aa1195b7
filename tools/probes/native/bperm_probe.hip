// Victim probes for the co-residency hunt: wave reductions by ds_bpermute_b32 (__shfl_xor) and by DPP, on
// register-only data (no global loads before the result store). Launched beside a conv kernel from
// tools/bperm_beside.py; out[block] = number of iterations whose sum was wrong.
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/native/bperm_probe.hip -o build/libbperm_probe.so
#include <hip/hip_runtime.h>
#include <cstdlib>
__device__ __forceinline__ float sum_bperm(float v) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ float sum_dpp(float v) {
  // row_shr / row_bcast reduction (no LDS hardware): result valid in lane 63
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x111, 0xf, 0xf, false));  // row_shr:1
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x112, 0xf, 0xf, false));  // row_shr:2
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x114, 0xf, 0xe, false));  // row_shr:4 (bank mask: not bank 0)
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x118, 0xf, 0xc, false));  // row_shr:8
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x142, 0xa, 0xf, false));  // row_bcast:15
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x143, 0xc, 0xf, false));  // row_bcast:31
  return v;
}
template <bool DPP>
__global__ __launch_bounds__(256) void probe(int* out, int iters) {
  const int lane = threadIdx.x & 63;
  int bad = 0;
  for (int it = 0; it < iters; ++it) {
    // integers in float: every partial sum is exact
    const float v = (float)((lane * 7 + it * 13 + (int)blockIdx.x) & 1023);
    float want = 0.f;
    for (int l = 0; l < 64; ++l) want += (float)((l * 7 + it * 13 + (int)blockIdx.x) & 1023);
    float got;
    if (DPP) { got = sum_dpp(v); got = __shfl(got, 63); } else got = sum_bperm(v);
    asm volatile("" : "+v"(got));
    if (got != want) ++bad;
  }
  if (lane == 0 && bad) atomicAdd(out + (DPP ? 1 : 0), bad);
}
// load checker: buf[i] == (float)i; the access pattern of att_scores_fwd_kernel (a wave takes 4 rows of A floats at a time,
// lane = 4 consecutive floats, + 256 per inner step). out[2] = wrong elements, out[3..5] = first (index, bits got)
__global__ __launch_bounds__(256) void load_probe(const float* __restrict__ buf, int rows, int A, int* out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int pc = (rows + 3) / 4;
  const int p0 = blockIdx.y * pc, p1 = min(rows, p0 + pc);
  const float* a1 = buf + (long)blockIdx.x * rows * A;
  for (int p = p0 + 4 * wave; p < p1; p += 16) {
    for (int a = lane * 4; a < A; a += 256) {
      float4 x[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) x[u] = *reinterpret_cast<const float4*>(a1 + (long)min(p + u, p1 - 1) * A + a);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const long base = (long)blockIdx.x * rows * A + (long)min(p + u, p1 - 1) * A + a;
        const float e[4] = {x[u].x, x[u].y, x[u].z, x[u].w};
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (e[k] != (float)(base + k)) {
            if (atomicAdd(out + 2, 1) == 0) { out[3] = (int)(base + k); out[4] = __float_as_int(e[k]); }
          }
      }
    }
  }
}
extern "C" void launch_load_probe(const float* buf, int batch, int rows, int A, int* out, void* stream) {
  hipLaunchKernelGGL(load_probe, dim3(batch, 4), dim3(256), 0, (hipStream_t)stream, buf, rows, A, out);
}
// att_scores_fwd_kernel (csrc/att_kernels.hip) with the per-lane partial sums kept: part [rows][P][64]
template <bool NOPK>
__global__ __launch_bounds__(256) void scores_probe(const float* __restrict__ att1, const float* __restrict__ att2, long ldz,
                                                    const float* __restrict__ wf, int P, int A, float* __restrict__ escore,
                                                    float* __restrict__ part) {
  const int j = blockIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int pc = (P + 3) / 4;
  const int p0 = blockIdx.y * pc, p1 = min(P, p0 + pc);
  const float* a1 = att1 + (long)j * P * A;
  const float* a2 = att2 + (long)j * ldz;
  for (int p = p0 + 4 * wave; p < p1; p += 16) {
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    for (int a = lane * 4; a < A; a += 256) {
      const float4 y = *reinterpret_cast<const float4*>(a2 + a);
      const float4 w = *reinterpret_cast<const float4*>(wf + a);
      float4 x[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) x[u] = *reinterpret_cast<const float4*>(a1 + (long)min(p + u, p1 - 1) * A + a);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        s[u] = fmaf(fmaxf(x[u].x + y.x, 0.f), w.x, s[u]);
        s[u] = fmaf(fmaxf(x[u].y + y.y, 0.f), w.y, s[u]);
        s[u] = fmaf(fmaxf(x[u].z + y.z, 0.f), w.z, s[u]);
        s[u] = fmaf(fmaxf(x[u].w + y.w, 0.f), w.w, s[u]);
        if (NOPK) asm volatile("" : "+v"(s[u]));      // keeps the four sums out of v_pk_fma_f32
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (p + u < p1) part[((long)j * P + p + u) * 64 + lane] = s[u];
      const float t = sum_bperm(s[u]);
      if (lane == 0 && p + u < p1) escore[(long)j * P + p + u] = t;
    }
  }
}
extern "C" void launch_scores_probe(const float* att1, const float* att2, long ldz, const float* wf, int rows, int P, int A,
                                    float* escore, float* part, void* stream) {
  if (getenv("PROBE_NOPK")) hipLaunchKernelGGL(scores_probe<true>, dim3(rows, 4), dim3(256), 0, (hipStream_t)stream, att1, att2, ldz, wf, P, A, escore, part);
  else hipLaunchKernelGGL(scores_probe<false>, dim3(rows, 4), dim3(256), 0, (hipStream_t)stream, att1, att2, ldz, wf, P, A, escore, part);
}
// micro-aggressors: MFMA-only loops (registers only), optionally with VALU between the MFMAs
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef __bf16 b8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
template <int MODE>
__global__ __launch_bounds__(256, 2) void mfma_aggr(float* out, int iters) {
  h8 a, b; b8 ab, bb;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(0.5f + i); ab[i] = (__bf16)(threadIdx.x * 0.001f + i); bb[i] = (__bf16)(0.5f + i); }
  f16v acc[4];
  for (int k = 0; k < 4; ++k) for (int i = 0; i < 16; ++i) acc[k][i] = 0.f;
  float v = threadIdx.x;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (MODE == 2) acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, acc[k], 0, 0, 0);
      else acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[k], 0, 0, 0);
      if (MODE == 1) {
#pragma unroll
        for (int j = 0; j < 5; ++j) { v = v * 1.0001f + 0.5f; asm volatile("" : "+v"(v)); }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  float s = v;
  for (int k = 0; k < 4; ++k) for (int i = 0; i < 16; ++i) s += acc[k][i];
  if (s == 12345.678f) out[0] = s;
}
extern "C" void launch_mfma_aggr(int mode, float* out, int blocks, int iters, void* stream) {
  if (mode == 0) hipLaunchKernelGGL(mfma_aggr<0>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, out, iters);
  else if (mode == 1) hipLaunchKernelGGL(mfma_aggr<1>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, out, iters);
  else hipLaunchKernelGGL(mfma_aggr<2>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, out, iters);
}
// VALU victims on register-only, exactly representable data. out[6] wrong v_pk_fma_f32 (plain), out[7] wrong v_fma_f32,
// out[13] wrong v_pk_fma_f32 op_sel_hi:[1,0,1] (the broadcast form hipcc's SLP vectoriser emits), out[14] wrong v_pk_mul_f32,
// out[15] wrong v_pk_add_f32
typedef float f2 __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(256) void valu_probe(int* out, int iters) {
  const int lane = threadIdx.x & 63;
  int badp = 0, bads = 0, badb = 0, badm = 0, bada = 0;
  for (int it = 0; it < iters; ++it) {
    const float base = (float)((lane + it) & 255);
    f2 accp = {0.f, 0.f}, accb = {0.f, 0.f}, accm = {0.f, 0.f}, acca = {0.f, 0.f};
    float acc0 = 0.f, acc1 = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      f2 x = {base + k, base + 2 * k}, y = {3.f, 5.f};
      asm volatile("" : "+v"(x), "+v"(y));
      asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(accp) : "v"(x), "v"(y));
      asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(accb) : "v"(x), "v"(y));   // both halves x * y.lo
      f2 m;
      asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(m) : "v"(x), "v"(y));
      asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(accm) : "v"(m));
      asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(acca) : "v"(x));
      float xs0 = base + k, xs1 = base + 2 * k;
      asm volatile("" : "+v"(xs0), "+v"(xs1));
      acc0 = __builtin_fmaf(xs0, 3.f, acc0);
      acc1 = __builtin_fmaf(xs1, 5.f, acc1);
    }
    float w0 = 0.f, w1 = 0.f, b1 = 0.f, a0 = 0.f, a1 = 0.f;
    for (int k = 0; k < 8; ++k) { w0 += (base + k) * 3.f; w1 += (base + 2 * k) * 5.f; b1 += (base + 2 * k) * 3.f; a0 += base + k; a1 += base + 2 * k; }
    asm volatile("" : "+v"(accp), "+v"(acc0), "+v"(acc1), "+v"(accb), "+v"(accm), "+v"(acca));
    if (accp[0] != w0 || accp[1] != w1) ++badp;
    if (acc0 != w0 || acc1 != w1) ++bads;
    if (accb[0] != w0 || accb[1] != b1) ++badb;
    if (accm[0] != w0 || accm[1] != w1) ++badm;
    if (acca[0] != a0 || acca[1] != a1) ++bada;
  }
  if (badp) atomicAdd(out + 6, badp);
  if (bads) atomicAdd(out + 7, bads);
  if (badb) atomicAdd(out + 13, badb);
  if (badm) atomicAdd(out + 14, badm);
  if (bada) atomicAdd(out + 15, bada);
}
extern "C" void launch_valu_probe(int* out, int blocks, int iters, void* stream) {
  hipLaunchKernelGGL(valu_probe, dim3(blocks), dim3(256), 0, (hipStream_t)stream, out, iters);
}
// counted-wait victim: six 16-B loads, each register tuple STORED (the store reads its data registers at issue) right behind
// s_waitcnt vmcnt(4 / 4 / 3 / 2 / 1 / 0) -- vmcnt retires in issue order, so each tuple must already hold its data.
// buf[i] == (float)i. scratch: 6 x 16 B per thread. out[8 + k] = wrong elements behind the k-th wait.
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void vmcnt_probe(const float* __restrict__ buf, long n, int iters, int* out, float* scratch) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int bad[5] = {0, 0, 0, 0, 0};
  float* mine = scratch + ((long)blockIdx.x * 256 + threadIdx.x) * 24;
  for (int it = 0; it < iters; ++it) {
    unsigned off[6];
    for (int k = 0; k < 6; ++k) {
      // like att_scores: one near, L1-resident line (k = 0, 5) and four rows far apart
      const long e = (k == 0 || k == 5) ? (long)(lane * 4 + k * 256) : (((long)(blockIdx.x * 4 + wave) * 977 + it * 131 + k * 53) % (n / 512)) * 512 + lane * 4 + ((it & 1) ? 256 : 0);
      off[k] = (unsigned)(e * 4);
    }
    f4 v[6];
    for (int k = 0; k < 6; ++k) { v[k] = f4{-1.f, -1.f, -1.f, -1.f}; asm volatile("" : "+v"(v[k])); }
    const unsigned so = 0;
    asm volatile(
        "global_load_dwordx4 %0, %6, %12\n\tglobal_load_dwordx4 %1, %7, %12\n\tglobal_load_dwordx4 %2, %8, %12\n\t"
        "global_load_dwordx4 %3, %9, %12\n\tglobal_load_dwordx4 %4, %10, %12\n\tglobal_load_dwordx4 %5, %11, %12\n\t"
        "s_waitcnt vmcnt(4)\n\tglobal_store_dwordx4 %14, %0, %13\n\tglobal_store_dwordx4 %14, %1, %13 offset:16\n\t"
        "s_waitcnt vmcnt(5)\n\tglobal_store_dwordx4 %14, %2, %13 offset:32\n\t"
        "s_waitcnt vmcnt(5)\n\tglobal_store_dwordx4 %14, %3, %13 offset:48\n\t"
        "s_waitcnt vmcnt(5)\n\tglobal_store_dwordx4 %14, %4, %13 offset:64\n\t"
        "s_waitcnt vmcnt(5)\n\tglobal_store_dwordx4 %14, %5, %13 offset:80\n\t"
        "s_waitcnt vmcnt(0)"
        : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5])
        : "v"(off[0]), "v"(off[1]), "v"(off[2]), "v"(off[3]), "v"(off[4]), "v"(off[5]), "s"(buf), "s"(scratch),
          "v"((unsigned)(((long)blockIdx.x * 256 + threadIdx.x) * 96))
        : "memory");
    // (vmcnt counts the stores too: behind the first wait 4 loads are out; each later wait allows the remaining loads plus the
    //  stores issued so far minus ... -- written as "at most 5 outstanding", which with in-order retirement means the next load is in)
    const int grp[6] = {0, 0, 1, 2, 3, 4};
    for (int k = 0; k < 6; ++k) {
      const float got = mine[4 * k];
      if (got != (float)(off[k] / 4)) ++bad[grp[k]];
    }
  }
  for (int k = 0; k < 5; ++k) if (bad[k]) atomicAdd(out + 8 + k, bad[k]);
}
extern "C" void launch_vmcnt_probe(const float* buf, long n, int blocks, int iters, int* out, float* scratch, void* stream) {
  hipLaunchKernelGGL(vmcnt_probe, dim3(blocks), dim3(256), 0, (hipStream_t)stream, buf, n, iters, out, scratch);
}
// Which instructions are unsafe as FIRST consumers of freshly loaded registers beside an MFMA+VALU wave?  buf[i] == (float)i
// (ibuf[i] == i). Same access pattern as att_scores; each loaded value goes straight into ONE instruction of the kind under
// test and the result is compared with the value computed from the known index. out[16 + k]: k = 0 v_fma_f32, 1 v_pk_fma_f32,
// 2 v_mul_lo_u32, 3 v_exp_f32, 4 v_rcp_f32, 5 v_lshl_add_u64, 6 v_cvt_f32_u32, 7 v_mul_f64 (via cvt)
__global__ __launch_bounds__(256) void first_use_probe(const float* __restrict__ buf, const unsigned* __restrict__ ibuf, int rows, int A,
                                                       int* out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int pc = (rows + 3) / 4;
  const int p0 = blockIdx.y * pc, p1 = min(rows, p0 + pc);
  const long rb = (long)blockIdx.x * rows * A;
  int bad[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int p = p0 + 4 * wave; p < p1; p += 16) {
    for (int a = lane * 4; a < A; a += 256) {
      float4 x[4]; uint4 xi[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const long e = rb + (long)min(p + u, p1 - 1) * A + a;
        x[u] = *reinterpret_cast<const float4*>(buf + e);
        xi[u] = *reinterpret_cast<const uint4*>(ibuf + e);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const long e = rb + (long)min(p + u, p1 - 1) * A + a;
        const float g0 = (float)e, g1 = (float)(e + 1), g2 = (float)(e + 2), g3 = (float)(e + 3);
        float r0, r3, r4, r6; f2 r1; unsigned r2; unsigned long long r5; double r7;
        const f2 pk = {x[u].y, x[u].z}; const f2 one = {1.f, 1.f}, three = {3.f, 3.f};
        { const float c3 = 3.f, c1 = 1.f; asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(r0) : "v"(x[u].x), "v"(c3), "v"(c1)); }
        asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(r1) : "v"(pk), "v"(three), "v"(one));
        { const unsigned c7 = 7u; asm volatile("v_mul_lo_u32 %0, %1, %2" : "=v"(r2) : "v"(xi[u].x), "v"(c7)); }
        asm volatile("v_exp_f32 %0, %1" : "=v"(r3) : "v"(x[u].w));
        asm volatile("v_rcp_f32 %0, %1" : "=v"(r4) : "v"(x[u].w));
        const unsigned long long base = 0x100000000ull;
        asm volatile("v_lshl_add_u64 %0, %1, 2, %2" : "=v"(r5) : "v"((unsigned long long)xi[u].y | ((unsigned long long)xi[u].z << 32)), "v"(base));
        asm volatile("v_cvt_f32_u32 %0, %1" : "=v"(r6) : "v"(xi[u].w));
        asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(r7) : "v"(x[u].x));
        float w3, w4;
        asm volatile("v_exp_f32 %0, %1" : "=v"(w3) : "v"(g3));
        asm volatile("v_rcp_f32 %0, %1" : "=v"(w4) : "v"(g3));
        if (r0 != g0 * 3.f + 1.f) ++bad[0];
        if (r1[0] != g1 * 3.f + 1.f || r1[1] != g2 * 3.f + 1.f) ++bad[1];
        if (r2 != (unsigned)e * 7u) ++bad[2];
        if (__float_as_uint(r3) != __float_as_uint(w3)) ++bad[3];
        if (__float_as_uint(r4) != __float_as_uint(w4)) ++bad[4];
        if (r5 != ((((unsigned long long)(unsigned)(e + 1)) | ((unsigned long long)(unsigned)(e + 2) << 32)) << 2) + base) ++bad[5];
        if (r6 != g3) ++bad[6];
        if (r7 != (double)g0) ++bad[7];
      }
    }
  }
  for (int k = 0; k < 8; ++k) if (bad[k]) atomicAdd(out + 16 + k, bad[k]);
}
extern "C" void launch_first_use_probe(const float* buf, const unsigned* ibuf, int batch, int rows, int A, int* out, void* stream) {
  hipLaunchKernelGGL(first_use_probe, dim3(batch, 4), dim3(256), 0, (hipStream_t)stream, buf, ibuf, rows, A, out);
}
extern "C" void launch_probe(int dpp, int* out, int blocks, int iters, void* stream) {
  if (dpp) hipLaunchKernelGGL(probe<true>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, out, iters);
  else hipLaunchKernelGGL(probe<false>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, out, iters);
}
