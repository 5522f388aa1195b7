// f32-in / f32-accumulate MFMA block GEMM core for gfx950 (v_mfma_f32_32x32x2_f32).
//
// One 256-thread workgroup (4 waves in a 2x2 grid) computes a BM x BN tile of
//   D[m][n] = sum_k X_A[m][k] * X_B[n][k]
// Both operands are staged global -> VGPR -> LDS in a k-major LDS image
// [BK][BM+4] / [BK][BN+4], so the MFMA side is identical whatever the global layout is:
// lane l of a wave feeds A[i = l&31][k = l>>5] and B[k = l>>5][j = l&31] with one
// ds_read_b32 each per MFMA (conflict-free: 32 consecutive floats per half-wave).
// f32 MFMA is a bitwise fmaf chain (one rounding per product), which is what keeps the
// training loss within 1e-4 of the fp32 CPU reference.
//
// Pipeline: register-staged double buffer, one barrier per K tile. Loads for tile t+1 are
// issued before the 32..64 MFMAs of tile t; any operand transform (the BatchNorm+ReLU
// prologue of the conv loader) runs in store(), after the MFMAs, so HBM latency hides
// under the matrix pipe.
#pragma once
#include <hip/hip_runtime.h>

namespace capnet {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kGemmThreads = 256;

template <int BM_, int BN_, int BK_>
struct TileCfg {
  static constexpr int BM = BM_, BN = BN_, BK = BK_;
  static constexpr int LDA = BM + 4;  // floats; keeps ds_write_b128 rows 16-B aligned
  static constexpr int LDB = BN + 4;
  static constexpr int A_ELEMS = BK * LDA;
  static constexpr int B_ELEMS = BK * LDB;
  static constexpr int STAGE_ELEMS = A_ELEMS + B_ELEMS;
  static constexpr int LDS_BYTES = 2 * STAGE_ELEMS * 4;
  static constexpr int MT = BM / 64;  // 32x32 MFMA tiles per wave along M
  static constexpr int NT = BN / 64;
  static_assert(BM % 64 == 0 && BN % 64 == 0, "wave grid is 2x2 of 32-multiples");
  static_assert(BK % 8 == 0, "BK");  // the k loop is unrolled two MFMA k-steps (4 k) at a time
};

// ---- operand loaders -----------------------------------------------------------------
// Logical operand X[R][K]; tile rows [r0, r0+BR), k in [k0, k0+BK).
//
// load() only ISSUES global loads (into registers) and records what is valid; store() masks,
// transforms and writes the LDS image. Nothing in load() may depend on a loaded value: the
// mainloop calls load(t+1) BEFORE the MFMAs of tile t and store() after them, so any use of
// the data in load() would put an s_waitcnt vmcnt(0) in front of the matrix work.
// VEC = true: 16-B aligned rows (ld % 4 == 0, base % 16 == 0); loads are unconditional
// float4 from a clamped (always in-bounds) address, out-of-range elements are zeroed in
// store(). VEC = false: element-wise guarded loads (rare: odd leading dimensions).

__device__ __forceinline__ float4 mask4(float4 v, int n) {
  v.x = n > 0 ? v.x : 0.f;
  v.y = n > 1 ? v.y : 0.f;
  v.z = n > 2 ? v.z : 0.f;
  v.w = n > 3 ? v.w : 0.f;
  return v;
}

__device__ __forceinline__ float4 load4_scalar(const float* __restrict__ p, int n) {
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (n > 0) v.x = p[0];
  if (n > 1) v.y = p[1];
  if (n > 2) v.z = p[2];
  if (n > 3) v.w = p[3];
  return v;
}

// K-contiguous storage: X[r][k] = p[r*ld + k]. BK/4 threads cover one row.
template <int BR, int BK, int LD, bool VEC>
struct LoaderKContig {
  static constexpr int TPR = BK / 4;                  // threads per row
  static constexpr int RPP = kGemmThreads / TPR;      // rows per pass
  static constexpr int PASSES = BR / RPP;
  static_assert(BR % RPP == 0, "tile rows vs threads");
  const float* p;
  long ld;
  int R, K, r0;
  float4 v[PASSES];
  int nv[PASSES];
  __device__ __forceinline__ void init(const float* p_, long ld_, int R_, int K_, int r0_) {
    p = p_; ld = ld_; R = R_; K = K_; r0 = r0_;
  }
  __device__ __forceinline__ void load(int k0) {
    const int kc = threadIdx.x % TPR, rl = threadIdx.x / TPR;
    const int k = k0 + 4 * kc;
    int kn = K - k;
    kn = kn < 0 ? 0 : (kn > 4 ? 4 : kn);
    const int kk = kn > 0 ? k : 0;
#pragma unroll
    for (int ps = 0; ps < PASSES; ++ps) {
      const int r = r0 + rl + ps * RPP;
      const bool rv = r < R;
      const int rr = rv ? r : R - 1;
      nv[ps] = rv ? kn : 0;
      if (VEC) v[ps] = *reinterpret_cast<const float4*>(p + (long)rr * ld + kk);
      else v[ps] = load4_scalar(p + (long)rr * ld + kk, nv[ps]);
    }
  }
  __device__ __forceinline__ void store(float* lds) const {
    const int kc = threadIdx.x % TPR, rl = threadIdx.x / TPR;
#pragma unroll
    for (int ps = 0; ps < PASSES; ++ps) {
      const float4 t = mask4(v[ps], nv[ps]);
      float* d = lds + (4 * kc) * LD + rl + ps * RPP;
      d[0 * LD] = t.x;
      d[1 * LD] = t.y;
      d[2 * LD] = t.z;
      d[3 * LD] = t.w;
    }
  }
};

// R-contiguous storage: X[r][k] = p[k*ld + r]. BR/4 threads cover one k row.
template <int BR, int BK, int LD, bool VEC>
struct LoaderRContig {
  static constexpr int TPK = BR / 4;                  // threads per k row
  static constexpr int KPP = kGemmThreads / TPK;      // k rows per pass
  static constexpr int PASSES = (BK + KPP - 1) / KPP;
  static_assert(kGemmThreads % TPK == 0, "tile rows vs threads");
  const float* p;
  long ld;
  int R, K, r0;
  float4 v[PASSES];
  int nv[PASSES];
  __device__ __forceinline__ void init(const float* p_, long ld_, int R_, int K_, int r0_) {
    p = p_; ld = ld_; R = R_; K = K_; r0 = r0_;
  }
  __device__ __forceinline__ void load(int k0) {
    const int rc = threadIdx.x % TPK, kl = threadIdx.x / TPK;
    const int r = r0 + 4 * rc;
    int rn = R - r;
    rn = rn < 0 ? 0 : (rn > 4 ? 4 : rn);
    const int rr = rn > 0 ? r : 0;
#pragma unroll
    for (int ps = 0; ps < PASSES; ++ps) {
      const int kk = kl + ps * KPP;
      const int k = k0 + kk;
      const bool kv = k < K && kk < BK;
      const int kc = k < K ? k : K - 1;
      nv[ps] = kv ? rn : 0;
      if (VEC) v[ps] = *reinterpret_cast<const float4*>(p + (long)kc * ld + rr);
      else v[ps] = load4_scalar(p + (long)kc * ld + rr, nv[ps]);
    }
  }
  __device__ __forceinline__ void store(float* lds) const {
    const int rc = threadIdx.x % TPK, kl = threadIdx.x / TPK;
#pragma unroll
    for (int ps = 0; ps < PASSES; ++ps) {
      const int kk = kl + ps * KPP;
      if (kk < BK) *reinterpret_cast<float4*>(lds + kk * LD + 4 * rc) = mask4(v[ps], nv[ps]);
    }
  }
};

// ---- the block GEMM --------------------------------------------------------------------
// acc[mt][nt][r]: row = m_base + mt*32 + (r&3) + 8*(r>>2) + 4*(lane>>5), col = n_base + nt*32 + (lane&31)
template <class T, class AL, class BL>
__device__ __forceinline__ void gemm_block_mainloop(AL& al, BL& bl, int K, float* lds,
                                                    f32x16 (&acc)[T::MT][T::NT]) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int li = lane & 31, lh = lane >> 5;
#pragma unroll
  for (int mt = 0; mt < T::MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < T::NT; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

  const int nk = (K + T::BK - 1) / T::BK;
  al.load(0);
  bl.load(0);
  al.store(lds);
  bl.store(lds + T::A_ELEMS);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    float* cur = lds + (kt & 1) * T::STAGE_ELEMS;
    float* nxt = lds + ((kt + 1) & 1) * T::STAGE_ELEMS;
    const bool more = (kt + 1) < nk;
    if (more) {
      al.load((kt + 1) * T::BK);
      bl.load((kt + 1) * T::BK);
    }
    const float* As = cur + lh * T::LDA + wm * (T::BM / 2) + li;
    const float* Bs = cur + T::A_ELEMS + lh * T::LDB + wn * (T::BN / 2) + li;
    // fragments of k-step j+1 are read from LDS while the MFMAs of step j run (two named
    // register sets, static indexing), so a wave never idles the matrix pipe on LDS latency
    float a0[T::MT], b0[T::NT], a1[T::MT], b1[T::NT];
#pragma unroll
    for (int mt = 0; mt < T::MT; ++mt) a0[mt] = As[mt * 32];
#pragma unroll
    for (int nt = 0; nt < T::NT; ++nt) b0[nt] = Bs[nt * 32];
#pragma unroll
    for (int j = 0; j < T::BK / 2; j += 2) {
#pragma unroll
      for (int mt = 0; mt < T::MT; ++mt) a1[mt] = As[(2 * j + 2) * T::LDA + mt * 32];
#pragma unroll
      for (int nt = 0; nt < T::NT; ++nt) b1[nt] = Bs[(2 * j + 2) * T::LDB + nt * 32];
      __builtin_amdgcn_sched_barrier(0);  // keep the prefetch ahead of the MFMAs (hipcc sinks it)
#pragma unroll
      for (int mt = 0; mt < T::MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < T::NT; ++nt)
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[mt], b0[nt], acc[mt][nt], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (j + 2 < T::BK / 2) {
#pragma unroll
        for (int mt = 0; mt < T::MT; ++mt) a0[mt] = As[(2 * j + 4) * T::LDA + mt * 32];
#pragma unroll
        for (int nt = 0; nt < T::NT; ++nt) b0[nt] = Bs[(2 * j + 4) * T::LDB + nt * 32];
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int mt = 0; mt < T::MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < T::NT; ++nt)
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[mt], b1[nt], acc[mt][nt], 0, 0, 0);
    }
    if (more) {
      al.store(nxt);
      bl.store(nxt + T::A_ELEMS);
    }
    __syncthreads();
  }
}

// XCD-aware tile order: consecutive workgroup ids land on different XCDs (round-robin), so
// remap id -> (id % 8) * ceil-chunk + id / 8 (bijective form) to give every XCD a contiguous
// run of tiles; within a run the N tiles of one M panel are adjacent, so the A panel is
// fetched from HBM once per XCD and re-read from that XCD's L2. Speed only, never correctness.
__device__ __forceinline__ int xcd_remap(int id, int nwg) {
  const int q = nwg >> 3, r = nwg & 7;
  const int xcd = id & 7, within = id >> 3;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + within;
}

// Column sums of the accumulator tile (used for BatchNorm batch statistics):
// writes sum and sum of squares over this workgroup's BM rows for each of its BN columns.
// lds must hold >= 4*BN floats and be free (call after the mainloop's final barrier).
template <class T>
__device__ __forceinline__ void block_col_stats(const f32x16 (&acc)[T::MT][T::NT], float* lds,
                                                float* __restrict__ out_sum,
                                                float* __restrict__ out_sq, int n0, int N) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int li = lane & 31, lh = lane >> 5;
  float* s_sum = lds;               // [2][BN]
  float* s_sq = lds + 2 * T::BN;    // [2][BN]
#pragma unroll
  for (int nt = 0; nt < T::NT; ++nt) {
    float s = 0.f, q = 0.f;
#pragma unroll
    for (int mt = 0; mt < T::MT; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float x = acc[mt][nt][r];
        s += x;
        q = fmaf(x, x, q);
      }
    s += __shfl_xor(s, 32);
    q += __shfl_xor(q, 32);
    if (lh == 0) {
      const int c = wn * (T::BN / 2) + nt * 32 + li;
      s_sum[wm * T::BN + c] = s;
      s_sq[wm * T::BN + c] = q;
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < T::BN; c += kGemmThreads) {
    if (n0 + c < N) {
      out_sum[n0 + c] = s_sum[c] + s_sum[T::BN + c];
      out_sq[n0 + c] = s_sq[c] + s_sq[T::BN + c];
    }
  }
}

// Launches WITHOUT BatchNorm statistics (inference) have nothing that carries a non-finite output to bn_finalize's
// check, and a ReLU behind it would turn a NaN into 0: such launches look at their own outputs. t = sum of |outputs| of
// this thread; bit 3 of the error word = a non-finite value in the trunk.
__device__ __forceinline__ void flag_nonfinite(float t, int* err) {
  if (!(t < __builtin_inff())) atomicOr(err, 8);
}

// ---- helpers shared by the hand-scheduled conv kernels (conv_f32_v2.hip, conv_wino.hip) ----
typedef float f32x4 __attribute__((ext_vector_type(4)));

// mul = ceil(2^k / d), k = 24 + ceil(log2 d): exact quotient for every n < 2^24
static inline void magic_div(unsigned d, unsigned* mul, unsigned* sh) {
  unsigned l = 0;
  while ((1u << l) < d) ++l;
  const unsigned k = 24 + l;
  const unsigned long long m = ((1ull << k) + d - 1) / d;
  *mul = (unsigned)m;   // < 2^25
  *sh = k;
}
__device__ __forceinline__ unsigned fast_div(unsigned n, unsigned mul, unsigned sh) {
  return (unsigned)(((unsigned long long)n * mul) >> sh);
}

// LDS-DMA of 16 B per lane: LDS[m0_base + 16*lane] = *(sbase + voff). Inline asm on purpose:
// with the builtin, hipcc cannot tell the DMA's LDS destination from the stage being read and
// puts s_waitcnt vmcnt(0) in front of every ds_read of the k-loop (measured: the wait sat
// right before the MFMAs). An asm DMA is invisible to its bookkeeping; it is ordered by hand:
// it is issued BEFORE the tile's buffer loads, vmcnt retires in issue order, so once store()
// has consumed those loads the DMA has landed, and the __syncthreads() that follows publishes
// it to the other waves.
__device__ __forceinline__ void glds16(const float* sbase, int voff_bytes, unsigned lds_byte_addr) {
  unsigned keep;
  // s_nop 2: with the two SALU instructions in front of it, five wait states lie between anything the compiler
  // placed ahead of this statement and the DMA -- what a VMEM instruction needs after a VALU write of an SGPR it
  // reads (see gload16 below); it also covers the one state M0 needs.
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %2\n\t"
      "s_nop 2\n\t"
      "global_load_lds_dwordx4 %1, %3\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(voff_bytes), "s"(lds_byte_addr), "s"(sbase)
      : "memory");
}

// 16-B global load, address = uniform 64-bit base (SGPR pair) + per-lane 32-bit byte offset.
// Inline asm for the same reason as glds16: hipcc's own loads in this loop got 64-bit VALU
// address arithmetic and an s_waitcnt vmcnt(0) that drained the LDS-DMA before they issued.
// The result is NOT valid until the counted s_waitcnt, and the compiler does not know that: between this
// statement and the CAPNET_LANDED that follows the wait, dst must not be read, copied, spilled or re-used.
// The compiler is not obliged to honour that; tools/isa_inflight_check.py (tests/test_isa_cpu.py) verifies
// on the shipped ISA, path by path, that it did.
//
// s_nop 4 -- the cause of round 2's unexplained GPU memory faults. gfx950 needs FIVE wait states between a VALU
// instruction that writes an SGPR and a vector-memory instruction that reads it. The compiler pads that hazard for
// its own instructions but does not look inside an asm statement, and it does produce such VALU writes: the
// reload of a spilled SGPR is v_readlane_b32 sN, vSPILL, lane. Whenever register allocation put the reload of the
// base pair directly in front of the asm load (or store), the instruction went out with the OLD base: a wild
// address. Every asm statement with an "s" address operand therefore opens with its own wait states
// (tools/isa_inflight_check.py checks this, too: check_sgpr_hazard).
__device__ __forceinline__ void gload16(f32x4& dst, const float* sbase, unsigned voff_bytes) {
  asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2" : "=&v"(dst) : "v"(voff_bytes), "s"(sbase) : "memory");
}
// 4-byte store through a 64-bit VGPR address: no SGPR operand inside the statement, so the hazard above cannot
// arise (the address is one v_lshl_add_u64 of compiler code). Asm, not C++, because the hand-counted vmcnt waits
// of the persistent conv kernels rely on the exact number of stores an epilogue issues.
__device__ __forceinline__ void gstore32(float* base, unsigned off_bytes, float v) {
  float* p = reinterpret_cast<float*>(reinterpret_cast<char*>(base) + off_bytes);
  asm volatile("global_store_dword %0, %1, off" ::"v"(p), "v"(v) : "memory");
}
// "These registers of hand-issued loads have landed" -- placed behind the counted wait. Ties the registers
// (the values the program uses from here on are this statement's outputs) and leaves a marker in the
// assembly for the checker.
// Two forms. CAPNET_LANDEDn ties the registers ("+v": what the program uses afterwards are this statement's outputs, so
// no consumer can be scheduled above it). The tie is a two-address constraint, and where several branches each land
// the same registers hipcc satisfied it with a v_mov copy ABOVE the wait -- a copy of stale data (found by the checker
// in lstm_persist.hip). CAPNET_LANDED_INn only reads the registers (no new value, nothing to copy); the consumers
// are held back by the __builtin_amdgcn_sched_barrier(0) that must follow it.
#define CAPNET_LANDED_IN1(a) asm volatile("; capnet.landed %0" ::"v"(a) : "memory")
#define CAPNET_LANDED_IN4(a, b, c, d) asm volatile("; capnet.landed %0 %1 %2 %3" ::"v"(a), "v"(b), "v"(c), "v"(d) : "memory")
#define CAPNET_LANDED1(a) asm volatile("; capnet.landed %0" : "+v"(a)::"memory")
#define CAPNET_LANDED2(a, b) asm volatile("; capnet.landed %0 %1" : "+v"(a), "+v"(b)::"memory")
#define CAPNET_LANDED4(a, b, c, d) \
  asm volatile("; capnet.landed %0 %1 %2 %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d)::"memory")

}  // namespace capnet
