// Plain / batched f32 GEMM on the f32-input MFMA (decoder projections, vocabulary
// projection, encoder head, all backward GEMMs). Row-major, arbitrary leading dimensions:
//   C[M x N] (+)= op(A) . op(B) + bias[N]
//   transA = 0: A stored [M][K]      transA = 1: A stored [K][M]
//   transB = 0: B stored [K][N]      transB = 1: B stored [N][K]   (nn.Linear weight)
// Replaces the torch.nn.Linear / autograd matmuls of the reference decoders
// (stylenet/model.py:119-150,189-194; nic/model.py:77,105-113).
#include <cstdlib>

#include "common.h"
#include "mfma_core.h"
#include "kernels.h"

namespace capnet {

struct GemmArgs {
  const float* A;
  const float* B;
  float* C;
  const float* bias;
  int M, N, K;
  long lda, ldb, ldc;
  long sA, sB, sC, sBias;  // batch strides (elements)
  int accumulate;
  int tiles_m, tiles_n;
  int splitk, kchunk;  // splitk > 1: blockIdx.y = z*splitk + ks; slice ks writes slab C + ks*M*ldc
};

template <int BM, int BN, int BK, bool TA, bool TB, bool VEC>
__global__ __launch_bounds__(kGemmThreads) void gemm_f32_kernel(GemmArgs g) {
  using T = TileCfg<BM, BN, BK>;
  __shared__ __attribute__((aligned(16))) float lds[2 * T::STAGE_ELEMS];

  const int nwg = g.tiles_m * g.tiles_n;
  const int id = xcd_remap(blockIdx.x, nwg);
  const int tm = id / g.tiles_n, tn = id - tm * g.tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;
  const int z = blockIdx.y / g.splitk, ks = blockIdx.y - z * g.splitk;
  const int kb = ks * g.kchunk;
  const int Kloc = min(g.kchunk, g.K - kb);
  const float* A = g.A + (long)z * g.sA + (TA ? (long)kb * g.lda : (long)kb);
  const float* B = g.B + (long)z * g.sB + (TB ? (long)kb : (long)kb * g.ldb);
  float* C = g.C + (long)z * g.sC + (long)ks * g.M * g.ldc;
  const float* bias = g.bias ? g.bias + (long)z * g.sBias : nullptr;

  using ALoad = typename std::conditional<TA, LoaderRContig<BM, BK, T::LDA, VEC>,
                                          LoaderKContig<BM, BK, T::LDA, VEC>>::type;
  using BLoad = typename std::conditional<TB, LoaderKContig<BN, BK, T::LDB, VEC>,
                                          LoaderRContig<BN, BK, T::LDB, VEC>>::type;
  ALoad al;
  BLoad bl;
  al.init(A, g.lda, g.M, Kloc, m0);
  bl.init(B, g.ldb, g.N, Kloc, n0);

  f32x16 acc[T::MT][T::NT];
  gemm_block_mainloop<T>(al, bl, Kloc, lds, acc);

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int li = lane & 31, lh = lane >> 5;
#pragma unroll
  for (int nt = 0; nt < T::NT; ++nt) {
    const int n = n0 + wn * (BN / 2) + nt * 32 + li;
    if (n >= g.N) continue;
    const float bv = bias ? bias[n] : 0.f;
#pragma unroll
    for (int mt = 0; mt < T::MT; ++mt) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * (BM / 2) + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (m < g.M) {
          float* c = C + (long)m * g.ldc + n;
          float v = acc[mt][nt][r] + bv;
          if (g.accumulate) v += *c;
          *c = v;
        }
      }
    }
  }
}

template <int BM, int BN, int BK, bool TA, bool TB, bool VEC>
static void launch_gemm(const GemmArgs& g, int batch, hipStream_t stream) {
  dim3 grid(g.tiles_m * g.tiles_n, batch * g.splitk);
  hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, BK, TA, TB, VEC>), grid, dim3(kGemmThreads), 0,
                     stream, g);
}

template <int BM, int BN, int BK>
static void dispatch_layout(const GemmArgs& g, int batch, bool ta, bool tb, bool vec,
                            hipStream_t s) {
#define CAPNET_GEMM_CASE(TA, TB)                                     \
  if (ta == TA && tb == TB) {                                        \
    if (vec) launch_gemm<BM, BN, BK, TA, TB, true>(g, batch, s);     \
    else launch_gemm<BM, BN, BK, TA, TB, false>(g, batch, s);        \
    return;                                                          \
  }
  CAPNET_GEMM_CASE(false, true)
  CAPNET_GEMM_CASE(false, false)
  CAPNET_GEMM_CASE(true, false)
  CAPNET_GEMM_CASE(true, true)
#undef CAPNET_GEMM_CASE
}

// out[m][n] = sum_ks slab[ks][m][n] + bias[n] (+ out[m][n]); fixed summation order
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ slab,
                                                            int splitk, int M, int N,
                                                            float* __restrict__ out, long ldc,
                                                            const float* __restrict__ bias,
                                                            int accumulate) {
  const long total = (long)M * N;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long)gridDim.x * blockDim.x) {
    const int m = (int)(i / N), n = (int)(i - (long)m * N);
    float s = 0.f;
    int k = 0;
    for (; k + 8 <= splitk; k += 8) {   // 8 independent loads in flight, summed in slab order
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = slab[(long)(k + u) * total + i];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; k < splitk; ++k) s += slab[(long)k * total + i];
    if (bias) s += bias[n];
    float* o = out + (long)m * ldc + n;
    *o = accumulate ? *o + s : s;
  }
}

static bool b3_enabled() {
  static const bool on = [] { const char* e = getenv("CAPNET_NO_B3"); return !(e && e[0] == '1'); }();
  return on;
}

// Tile choice: the chip has 256 CUs and this kernel keeps ~3-4 workgroups per CU resident;
// prefer the 128x128 tile (best operand reuse) once it alone fills the chip, else 64x64.
int sgemm(bool ta, bool tb, int M, int N, int K, const float* A, long lda, const float* B,
          long ldb, float* C, long ldc, const float* bias, int accumulate, int batch, long sA,
          long sB, long sC, long sBias, int force_tile, hipStream_t stream) {
  CAPNET_REQUIRE(M >= 0 && N >= 0 && K >= 0 && batch >= 0, "sgemm: negative dimension");
  if (M == 0 || N == 0 || batch == 0) return kOk;
  CAPNET_REQUIRE(A && B && C, "sgemm: null operand");
  CAPNET_REQUIRE(lda >= (ta ? M : K) && ldb >= (tb ? K : N) && ldc >= N,
                 "sgemm: leading dimension too small (lda=%ld ldb=%ld ldc=%ld M=%d N=%d K=%d)",
                 lda, ldb, ldc, M, N, K);
  CAPNET_REQUIRE(batch <= 65535, "sgemm: batch too large");
  // Large products: the bf16 matrix cores on three exact pieces per operand (gemm_b3.hip: 1.1-1.9 x this file's and
  // gemm_dma.hip's f32-MFMA kernels on the decoders' shapes, errors against fp64 the same or smaller); CAPNET_NO_B3=1 keeps
  // everything here.
  // (without a workspace for split-K partials it needs a grid of its own: a workgroup alone on its CU takes 1.8 us per
  //  32-k step, and below ~190 tiles of 128 x 128 the f32 kernels' 64 x 64 tiles win -- tools/probes/b3_small.py)
  if (force_tile == 0 && b3_enabled() && (long)cdiv(M, 128) * cdiv(N, 128) * batch >= 192 && (double)M * N * K * batch >= 2.5e8 &&
      sgemm_b3_eligible(ta, tb, M, N, K, A, lda, B, ldb, C, ldc, bias, batch, sA, sB, sC, sBias))
    return sgemm_b3(ta, tb, M, N, K, A, lda, B, ldb, C, ldc, bias, accumulate, batch, sA, sB, sC, sBias, stream);
  // y = x . W^T + b with dense K-contiguous operands and enough rows for 128-row tiles: the LDS-DMA
  // core (gemm_dma.hip) -- the vocabulary projection, encoder_att over all pixels, ...
  if (!ta && tb && batch == 1 && !accumulate && force_tile == 0 && M > 64 &&
      (long)cdiv(M, 128) * (N / 64) >= 64 && sgemm_nt_dma_eligible(M, N, K, A, lda, B, ldb, C, ldc))
    return sgemm_nt_dma(M, N, K, A, lda, B, C, bias, stream);
  GemmArgs g;
  g.A = A; g.B = B; g.C = C; g.bias = bias;
  g.M = M; g.N = N; g.K = K;
  g.lda = lda; g.ldb = ldb; g.ldc = ldc;
  g.sA = sA; g.sB = sB; g.sC = sC; g.sBias = sBias;
  g.accumulate = accumulate;
  g.splitk = 1;
  g.kchunk = K;
  const bool vec = aligned16(A) && aligned16(B) && (lda % 4 == 0) && (ldb % 4 == 0) &&
                   (sA % 4 == 0) && (sB % 4 == 0);
  const long t128 = (long)cdiv(M, 128) * cdiv(N, 128) * batch;
  int tile = force_tile;
  if (tile == 0) tile = (t128 >= 384) ? 128 : 64;
  if (tile == 128) {
    g.tiles_m = cdiv(M, 128); g.tiles_n = cdiv(N, 128);
    dispatch_layout<128, 128, 16>(g, batch, ta, tb, vec, stream);
  } else if (tile == 64) {
    g.tiles_m = cdiv(M, 64); g.tiles_n = cdiv(N, 64);
    dispatch_layout<64, 64, 16>(g, batch, ta, tb, vec, stream);
  } else if (tile == 6432) {   // 64x64 tile, 32-deep k slices: half the k-loop round trips
    g.tiles_m = cdiv(M, 64); g.tiles_n = cdiv(N, 64);
    dispatch_layout<64, 64, 32>(g, batch, ta, tb, vec, stream);
  } else {
    CAPNET_REQUIRE(false, "sgemm: unknown tile %d", tile);
  }
  CAPNET_LAUNCH_CHECK();
  return kOk;
}

}  // namespace capnet

namespace capnet {

// ---- skinny products of the recurrence (M <= 64 rows per time step) ---------------------------
// Too few output tiles to fill 256 CUs and too little work per tile to hide a k-loop's memory
// latency: the generic kernel above spends one HBM round trip per BK = 16 slice. Here a
// workgroup owns one 64x64 output tile and ONE K chunk (KC = 64 or 128): it issues every load of
// the chunk at once (one round trip), stages both operands in LDS as 16-B cells along k
// (conflict-free half-cell reads feed two MFMA k-steps each), and writes its partial tile to a
// slab; slabs are summed in a fixed order by splitk_reduce_kernel. A is [M][K] row-major;
// TB: B is [N][K] (nn.Linear weight) else [K][N].
constexpr int kSkinnyRows = 64;
constexpr int kSkCell = kSkinnyRows + 1;  // cells per kq column of the A / B^T images
constexpr int kSkLdb = 80;                // dword row stride of the [k][n] image (TB = false)

template <int KC, bool TB>
__global__ __launch_bounds__(256) void gemm_skinny_kernel(const float* __restrict__ A, long lda,
                                                          const float* __restrict__ B, long ldb,
                                                          float* __restrict__ slab, int M, int N,
                                                          int K, int tiles_n, long sA, long sB) {
  // blockIdx.z = batch member z: A + z*sA, B + z*sB, slab columns [z*N, (z+1)*N) of a row of
  // gridDim.z*N (the gate-batched S_g / U_g products of the factored chain)
  typedef float f32x4v __attribute__((ext_vector_type(4)));
  typedef float f32x2v __attribute__((ext_vector_type(2)));
  constexpr int KQ = KC / 4;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  f32x4v* a_cells = reinterpret_cast<f32x4v*>(lds);                  // [KQ][65] cells
  float* b_img = lds + (size_t)KQ * kSkCell * 4;                     // TB: [KQ][65] cells; else [KC][80]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tn = blockIdx.x % tiles_n, ks = blockIdx.x / tiles_n, tm = blockIdx.y;
  const int m0 = tm * 64, n0 = tn * 64, k0 = ks * KC;
  A += (long)blockIdx.z * sA;
  B += (long)blockIdx.z * sB;
  const int kq_real = (min(KC, K - k0) + 3) / 4;   // K % 4 == 0 is required by the host
  constexpr int NA = (64 * KQ) / 256;              // cells per thread
  f32x4v va[NA], vb[NA];
#pragma unroll
  for (int q = 0; q < NA; ++q) {
    const int idx = tid + 256 * q;
    const int row = idx / KQ, kq = idx - row * KQ;
    const int r = min(m0 + row, M - 1), kk = min(kq, kq_real - 1);
    va[q] = *reinterpret_cast<const f32x4v*>(A + (long)r * lda + k0 + 4 * kk);
  }
  if (TB) {
#pragma unroll
    for (int q = 0; q < NA; ++q) {
      const int idx = tid + 256 * q;
      const int row = idx / KQ, kq = idx - row * KQ;
      const int r = min(n0 + row, N - 1), kk = min(kq, kq_real - 1);
      vb[q] = *reinterpret_cast<const f32x4v*>(B + (long)r * ldb + k0 + 4 * kk);
    }
  } else {
    // [K][N]: thread -> (k row, 16-B column group); N % 4 == 0 required by the host
#pragma unroll
    for (int q = 0; q < NA; ++q) {
      const int idx = tid + 256 * q;
      const int kr = idx >> 4, c4 = idx & 15;
      const int kk = min(k0 + kr, K - 1), nn = min(n0 + 4 * c4, N - 4);
      vb[q] = *reinterpret_cast<const f32x4v*>(B + (long)kk * ldb + nn);
    }
  }
#pragma unroll
  for (int q = 0; q < NA; ++q) {
    const int idx = tid + 256 * q;
    const int row = idx / KQ, kq = idx - row * KQ;
    const float m = (m0 + row < M && kq < kq_real) ? 1.f : 0.f;
    a_cells[kq * kSkCell + row] = va[q] * m;
  }
  if (TB) {
    f32x4v* b_cells = reinterpret_cast<f32x4v*>(b_img);
#pragma unroll
    for (int q = 0; q < NA; ++q) {
      const int idx = tid + 256 * q;
      const int row = idx / KQ, kq = idx - row * KQ;
      const float m = (n0 + row < N && kq < kq_real) ? 1.f : 0.f;
      b_cells[kq * kSkCell + row] = vb[q] * m;
    }
  } else {
#pragma unroll
    for (int q = 0; q < NA; ++q) {
      const int idx = tid + 256 * q;
      const int kr = idx >> 4, c4 = idx & 15;
      const bool ok = k0 + kr < K && n0 + 4 * c4 < N;   // N % 4 == 0: groups are all-or-nothing
      const float m = ok ? 1.f : 0.f;
      *reinterpret_cast<f32x4v*>(b_img + kr * kSkLdb + 4 * c4) = vb[q] * m;
    }
  }
  __syncthreads();
  const int wm = wave >> 1, wn = wave & 1, li = lane & 31, lh = lane >> 5;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const f32x2v* ap = reinterpret_cast<const f32x2v*>(lds) + 2 * (wm * 32 + li) + lh;
  if (TB) {
    const f32x2v* bp = reinterpret_cast<const f32x2v*>(b_img) + 2 * (wn * 32 + li) + lh;
#pragma unroll
    for (int kq = 0; kq < KQ; ++kq) {
      const f32x2v a = ap[2 * kq * kSkCell], b = bp[2 * kq * kSkCell];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc, 0, 0, 0);
    }
  } else {
    const float* bp = b_img + (2 * lh) * kSkLdb + wn * 32 + li;
#pragma unroll
    for (int kq = 0; kq < KQ; ++kq) {
      const f32x2v a = ap[2 * kq * kSkCell];
      const float b0 = bp[(4 * kq) * kSkLdb], b1 = bp[(4 * kq + 1) * kSkLdb];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b0, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b1, acc, 0, 0, 0);
    }
  }
  const long ldo = (long)gridDim.z * N;
  float* out = slab + (long)ks * M * ldo + (long)blockIdx.z * N;
  const int n = n0 + wn * 32 + li;
  if (n < N) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      if (m < M) out[(long)m * ldo + n] = acc[r];
    }
  }
}

template <int KC, bool TB>
static int launch_skinny(const float* A, long lda, const float* B, long ldb, float* slab, int M, int N,
                         int K, int batch, long sA, long sB, hipStream_t stream) {
  const int tiles_n = cdiv(N, 64), tiles_m = cdiv(M, 64), splits = cdiv(K, KC);
  const size_t lds_bytes = ((size_t)(KC / 4) * kSkCell * 4 +
                            (TB ? (size_t)(KC / 4) * kSkCell * 4 : (size_t)KC * kSkLdb)) * sizeof(float);
  auto kern = gemm_skinny_kernel<KC, TB>;
  static bool attr_set = false;
  if (lds_bytes > 64 * 1024 && !attr_set) {
    CAPNET_HIP_CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize,
                                         (int)lds_bytes));
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3(tiles_n * splits, tiles_m, batch), dim3(256), lds_bytes, stream, A,
                     lda, B, ldb, slab, M, N, K, tiles_n, sA, sB);
  return kOk;
}


// ---- M <= 16 rows (a 12-image batch's time step; decode; beam search): one launch, no reduce kernel --------------
// The 64 x 64-tile kernel above loads 64 rows to use 12 and leaves the K-chunk sum to a second launch, whose
// ~4 us + a kernel boundary is as long as the product itself at these sizes. Here a workgroup owns 16 rows x 64
// columns x ONE 256-k chunk (v_mfma_f32_16x16x4_f32, one 16-column strip per wave, everything loaded in one round
// trip), writes its partial tile to a slab, and the workgroup that arrives LAST at the tile's counter sums the
// slabs -- in slab order, so the result does not depend on who was last -- adds the bias and writes C.
// Cross-workgroup visibility is MI355X_MICROARCH.md's first measured hand-off row, cell by cell: every partial byte
// stored `sc1` (16-B stores), every storing wave `s_waitcnt vmcnt(0)` (asm), a workgroup barrier, ONE lane's
// agent-scope atomic add whose returned value tells the last arriver, that workgroup's other waves behind a barrier
// the adding wave joins, every load of the partials `sc1`; hipMalloc memory; one workgroup per CU (the LDS
// request is padded past half a CU's). The counters (one int per output tile, caller-provided, zero before the first
// use) are put back to zero by the workgroup that consumed them.
// The product is computed transposed (first MFMA operand = B) so that a lane holds 4 consecutive columns of one row.
constexpr int kR16KC = 256;                 // k per workgroup
constexpr int kR16ACell = 17;               // float4 cells per k-quad of the A image (16 rows + 1)
constexpr int kR16BCell = 65;               // ... of the B^T image (64 columns + 1)
constexpr int kR16Ldb = 68;                 // dword row stride of the [k][n] image (TB = false): the four k-quads of a
                                            // fragment read land 16 banks apart
constexpr size_t kR16LdsBytes = 100 * 1024; // > 80 KB: one workgroup per CU

struct R16Args {
  const float* A; const float* B; float* slab; float* C; const float* bias; int* counters;
  long lda, ldb, ldc, sA, sB;
  int M, N, K, tiles_n, splits, sub, accumulate;   // a workgroup owns `sub` consecutive 256-k chunks
  int slabs_only;                                  // leave the K-split partials in `slab` for the consumer to sum: no hand-off
};

template <bool TB>
__global__ __launch_bounds__(256) void gemm_rows16_kernel(const R16Args g) {
  typedef float f32x4v __attribute__((ext_vector_type(4)));
  constexpr int KQ = kR16KC / 4;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  __shared__ int s_last;
  f32x4v* a_cells = reinterpret_cast<f32x4v*>(lds);                    // [KQ][17] cells
  float* b_img = lds + (size_t)KQ * kR16ACell * 4;                     // TB: [KQ][65] cells; else [KC][80]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tn = blockIdx.x % g.tiles_n, ks = blockIdx.x / g.tiles_n, z = blockIdx.z;
  const int n0 = tn * 64;
  const float* A = g.A + (long)z * g.sA;
  const float* B = g.B + (long)z * g.sB;
  // this workgroup's k range: `sub` chunks of 256, one load round trip each, the next one requested before the
  // MFMAs of the current one (the host sizes splits x sub so that the grid is one wave of workgroups)
  f32x4v va[4], vb[16];
  auto request = [&](int k0) {
    const int kq_real = (min(kR16KC, g.K - k0) + 3) / 4;   // K % 4 == 0 is required by the host
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int idx = tid + 256 * q, row = idx >> 6, kq = idx & 63;
      va[q] = *reinterpret_cast<const f32x4v*>(A + (long)min(row, g.M - 1) * g.lda + k0 + 4 * min(kq, kq_real - 1));
    }
    if (TB) {
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int idx = tid + 256 * q, col = idx >> 6, kq = idx & 63;
        vb[q] = *reinterpret_cast<const f32x4v*>(B + (long)min(n0 + col, g.N - 1) * g.ldb + k0 + 4 * min(kq, kq_real - 1));
      }
    } else {
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int idx = tid + 256 * q, kr = idx >> 4, c4 = idx & 15;
        vb[q] = *reinterpret_cast<const f32x4v*>(B + (long)min(k0 + kr, g.K - 1) * g.ldb + min(n0 + 4 * c4, g.N - 4));
      }
    }
  };
  auto stage = [&](int k0) {
    const int kq_real = (min(kR16KC, g.K - k0) + 3) / 4;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int idx = tid + 256 * q, row = idx >> 6, kq = idx & 63;
      a_cells[kq * kR16ACell + row] = va[q] * ((row < g.M && kq < kq_real) ? 1.f : 0.f);
    }
    if (TB) {
      f32x4v* b_cells = reinterpret_cast<f32x4v*>(b_img);
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int idx = tid + 256 * q, col = idx >> 6, kq = idx & 63;
        b_cells[kq * kR16BCell + col] = vb[q] * ((n0 + col < g.N && kq < kq_real) ? 1.f : 0.f);
      }
    } else {
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int idx = tid + 256 * q, kr = idx >> 4, c4 = idx & 15;
        *reinterpret_cast<f32x4v*>(b_img + kr * kR16Ldb + 4 * c4) =
            vb[q] * ((k0 + kr < g.K && n0 + 4 * c4 < g.N) ? 1.f : 0.f);   // N % 4 == 0: groups are all-or-nothing
      }
    }
  };
  // lane (r, q): operand element r (a row of A / a column of the wave's strip), k = 16 i + 4 q + e
  const int r = lane & 15, q4 = lane >> 4;
  f32x4v acc = {0.f, 0.f, 0.f, 0.f};
  const int kbeg = ks * g.sub * kR16KC;
  request(kbeg);
  for (int sc = 0; sc < g.sub; ++sc) {
    const int k0 = kbeg + sc * kR16KC;
    if (k0 >= g.K) break;                                  // (uniform: the last split may own fewer chunks)
    stage(k0);
    __syncthreads();
    if (sc + 1 < g.sub && k0 + kR16KC < g.K) request(k0 + kR16KC);
    if (TB) {
      const f32x4v* b_cells = reinterpret_cast<const f32x4v*>(b_img);
#pragma unroll 4
      for (int i = 0; i < KQ / 4; ++i) {
        const f32x4v a = a_cells[(4 * i + q4) * kR16ACell + r];
        const f32x4v b = b_cells[(4 * i + q4) * kR16BCell + 16 * wave + r];
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(b[0], a[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(b[1], a[1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(b[2], a[2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(b[3], a[3], acc, 0, 0, 0);
      }
    } else {
      const float* bp = b_img + (4 * q4) * kR16Ldb + 16 * wave + r;
#pragma unroll 4
      for (int i = 0; i < KQ / 4; ++i) {
        const f32x4v a = a_cells[(4 * i + q4) * kR16ACell + r];
        const float* bk = bp + (16 * i) * kR16Ldb;
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(bk[0], a[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(bk[kR16Ldb], a[1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(bk[2 * kR16Ldb], a[2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(bk[3 * kR16Ldb], a[3], acc, 0, 0, 0);
      }
    }
    __syncthreads();
  }
  // D^T: lane (r, q) holds row m = r, columns n0 + 16 wave + 4 q + (0..3)
  const int m = r, n = n0 + 16 * wave + 4 * q4;
  const long ldo = (long)gridDim.z * g.N;               // a slab row: the batch members' columns side by side
  const long col = (long)z * g.N + n;
  if (g.slabs_only) {                                   // slab[ks][M][N]: the next launch on the stream sums them, in slab order
    if (m < g.M && n < g.N) *reinterpret_cast<f32x4v*>(g.slab + ((long)ks * g.M + m) * ldo + col) = acc;
    return;
  }
  if (g.splits == 1) {                                  // the whole K in this workgroup: no slab, no hand-off
    if (m < g.M && n < g.N) {
      float* o = g.C + (long)m * g.ldc + col;
      f32x4v v = acc;
      if (g.bias) v += *reinterpret_cast<const f32x4v*>(g.bias + col);
      if (g.accumulate) v += *reinterpret_cast<const f32x4v*>(o);
      *reinterpret_cast<f32x4v*>(o) = v;
    }
    return;
  }
  if (m < g.M && n < g.N) {
    float* o = g.slab + ((long)ks * g.M + m) * ldo + col;
    asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(o), "v"(acc) : "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  int* ctr = g.counters + z * g.tiles_n + tn;
  if (tid == 0) s_last = __hip_atomic_fetch_add(ctr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == g.splits - 1;
  __syncthreads();
  if (!s_last) return;
  {
    // 16 rows x 16 column quads = one per thread; every slab in flight at once (4-B sc1 loads)
    const int rm = tid >> 4, rn = n0 + 4 * (tid & 15);
    if (rm < g.M && rn < g.N) {
      const long rc = (long)z * g.N + rn;
      const float* p = g.slab + (long)rm * ldo + rc;
      const long slab_stride = (long)g.M * ldo;
      f32x4v sum = {0.f, 0.f, 0.f, 0.f};
      for (int k = 0; k < g.splits; k += 16) {
        f32x4v v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
          const float* pk = p + (long)min(k + u, g.splits - 1) * slab_stride;
#pragma unroll
          for (int e = 0; e < 4; ++e) v[u][e] = __hip_atomic_load(pk + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
#pragma unroll
        for (int u = 0; u < 16; ++u)
          if (k + u < g.splits) sum += v[u];
      }
      float* o = g.C + (long)rm * g.ldc + rc;
      if (g.bias) sum += *reinterpret_cast<const f32x4v*>(g.bias + rc);
      if (g.accumulate) sum += *reinterpret_cast<const f32x4v*>(o);
      *reinterpret_cast<f32x4v*>(o) = sum;
    }
    if (tid == 0) __hip_atomic_store(ctr, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// true when the shape went to gemm_rows16_kernel
static bool try_rows16(bool tb, int M, int N, int K, const float* A, long lda, const float* B, long ldb, float* C,
                       long ldc, const float* bias, int accumulate, int batch, long sA, long sB, float* ws,
                       size_t ws_floats, int* counters, size_t n_counters, hipStream_t stream) {
  if (!counters || M > 16 || N % 4 != 0 || N < 4 || ldc % 4 != 0 || !aligned16(C) || (bias && !aligned16(bias)) ||
      !aligned16(ws))
    return false;
  // One wave of workgroups (<= 256, one per CU). A workgroup takes `sub` chunks of 256 k one after the other
  // (load round trip + staging + MFMAs: ~2.2 us each); K cut into `splits` costs the hand-off once (~3.5 us + 0.25 us
  // per slab the last workgroup has to fetch). Fitted to a sweep over `sub` on the six step shapes of a 12-image
  // attention batch (12 x {4608,8192,512} x 512: no split, 7.7-7.9 us against 10.6-11.4 for skinny + reduce;
  // 12 x 2048 x 2348: 2 chunks x 5 splits, 12.1 against 15.7; tools/probes/rows16_bench.py).
  const int tiles_n = cdiv(N, 64), chunks = cdiv(K, kR16KC);
  int sub = chunks, splits = 1;
  float best = 2.2f * chunks;
  for (int ns = 1; ns < chunks; ++ns) {
    const int sp = cdiv(chunks, ns);
    if ((long)tiles_n * batch * sp > 256) continue;
    const float cost = 2.2f * ns + 3.5f + 0.25f * sp;
    if (cost < best) { best = cost; sub = ns; splits = sp; }
  }
  if ((size_t)tiles_n * batch > n_counters || (splits > 1 && (size_t)splits * M * N * batch > ws_floats)) return false;
  R16Args g;
  g.A = A; g.B = B; g.slab = ws; g.C = C; g.bias = bias; g.counters = counters;
  g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.sA = sA; g.sB = sB;
  g.M = M; g.N = N; g.K = K; g.tiles_n = tiles_n; g.splits = splits; g.sub = sub; g.accumulate = accumulate;
  g.slabs_only = 0;
  static bool attr_set[2] = {false, false};
  const void* kern = tb ? (const void*)gemm_rows16_kernel<true> : (const void*)gemm_rows16_kernel<false>;
  if (!attr_set[tb]) {
    if (hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kR16LdsBytes) != hipSuccess) {
      (void)hipGetLastError();
      return false;
    }
    attr_set[tb] = true;
  }
  if (tb) hipLaunchKernelGGL(gemm_rows16_kernel<true>, dim3(tiles_n * splits, 1, batch), dim3(256), kR16LdsBytes, stream, g);
  else hipLaunchKernelGGL(gemm_rows16_kernel<false>, dim3(tiles_n * splits, 1, batch), dim3(256), kR16LdsBytes, stream, g);
  return true;
}

// The K-split partials of a product of at most 16 rows, left for the consumer: slab[k][M][N], k < *n_slabs (0: the
// shape does not qualify). No hand-off inside the launch -- the cut only has to keep the grid within one wave of
// workgroups, a workgroup then takes as few 256-k chunks as that allows (2.2 us each).
int sgemm_rows16_slabs(bool tb, int M, int N, int K, const float* A, long lda, const float* B, long ldb, float* ws,
                       size_t ws_floats, int* n_slabs, hipStream_t stream) {
  *n_slabs = 0;
  if (M <= 0 || M > 16 || N % 4 != 0 || N < 4 || K % 4 != 0 || lda % 4 != 0 || ldb % 4 != 0 || !aligned16(A) || !aligned16(B) ||
      !ws || !aligned16(ws))
    return kOk;
  const int tiles_n = cdiv(N, 64), chunks = cdiv(K, kR16KC);
  int sub = 1;
  while ((long)tiles_n * cdiv(chunks, sub) > 256) ++sub;
  const int splits = cdiv(chunks, sub);
  if ((size_t)splits * M * N > ws_floats) return kOk;
  R16Args g;
  g.A = A; g.B = B; g.slab = ws; g.C = nullptr; g.bias = nullptr; g.counters = nullptr;
  g.lda = lda; g.ldb = ldb; g.ldc = 0; g.sA = 0; g.sB = 0;
  g.M = M; g.N = N; g.K = K; g.tiles_n = tiles_n; g.splits = splits; g.sub = sub; g.accumulate = 0;
  g.slabs_only = 1;
  const void* kern = tb ? (const void*)gemm_rows16_kernel<true> : (const void*)gemm_rows16_kernel<false>;
  static bool attr_set[2] = {false, false};
  if (!attr_set[tb]) {
    if (hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kR16LdsBytes) != hipSuccess) {
      (void)hipGetLastError();
      return kOk;
    }
    attr_set[tb] = true;
  }
  if (tb) hipLaunchKernelGGL(gemm_rows16_kernel<true>, dim3(tiles_n * splits, 1, 1), dim3(256), kR16LdsBytes, stream, g);
  else hipLaunchKernelGGL(gemm_rows16_kernel<false>, dim3(tiles_n * splits, 1, 1), dim3(256), kR16LdsBytes, stream, g);
  CAPNET_LAUNCH_CHECK();
  *n_slabs = splits;
  return kOk;
}

int reduce_slabs(const float* slab, int count, int M, int N, float* out, long ldc, const float* bias,
                 int accumulate, hipStream_t stream) {
  CAPNET_REQUIRE(slab && out && count > 0 && M > 0 && N > 0, "reduce_slabs: bad argument");
  const long total = (long)M * N;
  hipLaunchKernelGGL(splitk_reduce_kernel, dim3((int)(cdiv(total, 256) > 2048 ? 2048 : cdiv(total, 256))),
                     dim3(256), 0, stream, slab, count, M, N, out, ldc, bias, accumulate);
  CAPNET_LAUNCH_CHECK();
  return kOk;
}

// K-split product for M <= 128 rows, optionally batched over `batch` members whose outputs are
// adjacent column blocks of C (sC == N, bias concatenated); falls back to sgemm otherwise.
int sgemm_splitk_batched(bool ta, bool tb, int M, int N, int K, const float* A, long lda,
                         const float* B, long ldb, float* C, long ldc, const float* bias,
                         int accumulate, int batch, long sA, long sB, long sC, long sBias, float* ws,
                         size_t ws_floats, hipStream_t stream, int* counters, size_t n_counters) {
  if (M == 0 || N == 0 || batch == 0) return kOk;
  const bool skinny_ok = !ta && ws && M <= 128 && K % 4 == 0 && lda % 4 == 0 && aligned16(A) &&
                         aligned16(B) && ldb % 4 == 0 && (tb || N % 4 == 0) && N >= 4 && K >= 64 &&
                         (batch == 1 || (sC == N && (!bias || sBias == N) && sA % 4 == 0 &&
                                         sB % 4 == 0 && batch <= 64));
  if (!skinny_ok) {
    // many rows but few output tiles and a long K (dH = dlogits . C: 1037 x 512 x 8192): the
    // k-loop kernel with K cut into slabs, enough of them to put ~4 workgroups on every CU
    const long tiles64 = (long)cdiv(M, 64) * cdiv(N, 64);
    // (gemm_b3.hip cuts K itself when its 128 x 128 tiles are few)
    if (ws && batch == 1 && M > 128 && K >= 512 && b3_enabled() && (double)M * N * K >= 2.5e8 &&
        sgemm_b3_eligible(ta, tb, M, N, K, A, lda, B, ldb, C, ldc, bias, 1, 0, 0, 0, 0)) {
      CAPNET_REQUIRE(A && B && C, "sgemm_splitk: null operand");
      return sgemm_b3(ta, tb, M, N, K, A, lda, B, ldb, C, ldc, bias, accumulate, 1, 0, 0, 0, 0, stream, ws, ws_floats);
    }
    if (ws && batch == 1 && M > 128 && tiles64 < 256 && K >= 1024) {
      CAPNET_REQUIRE(A && B && C, "sgemm_splitk: null operand");
      int splitk = (int)(1024 / tiles64);
      if (splitk > K / 256) splitk = K / 256;
      while (splitk > 1 && (size_t)splitk * M * N > ws_floats) --splitk;
      if (splitk > 1) {
        GemmArgs g;
        g.A = A; g.B = B; g.C = ws; g.bias = nullptr;
        g.M = M; g.N = N; g.K = K;
        g.lda = lda; g.ldb = ldb; g.ldc = N;
        g.sA = g.sB = g.sC = g.sBias = 0;
        g.accumulate = 0;
        g.kchunk = cdiv(cdiv(K, splitk), 16) * 16;
        g.splitk = cdiv(K, g.kchunk);
        g.tiles_m = cdiv(M, 64);
        g.tiles_n = cdiv(N, 64);
        const bool vec = aligned16(A) && aligned16(B) && (lda % 4 == 0) && (ldb % 4 == 0);
        dispatch_layout<64, 64, 16>(g, 1, ta, tb, vec, stream);
        return reduce_slabs(ws, g.splitk, M, N, C, ldc, bias, accumulate, stream);
      }
    }
    return sgemm(ta, tb, M, N, K, A, lda, B, ldb, C, ldc, bias, accumulate, batch, sA, sB, sC, sBias,
                 0, stream);
  }
  CAPNET_REQUIRE(A && B && C, "sgemm_splitk: null operand");
  if (try_rows16(tb, M, N, K, A, lda, B, ldb, C, ldc, bias, accumulate, batch, sA, sB, ws, ws_floats, counters, n_counters,
                 stream)) {
    CAPNET_LAUNCH_CHECK();
    return kOk;
  }
  const long tiles = (long)cdiv(M, 64) * cdiv(N, 64) * batch;
  const size_t out_floats = (size_t)M * N * batch;
  // one K chunk per workgroup: 64 when that still leaves the chip under-filled, else 128
  int kc = tiles * cdiv(K, 128) < 200 ? 64 : 128;
  if ((size_t)cdiv(K, kc) * out_floats > ws_floats) kc = 128;
  if ((size_t)cdiv(K, kc) * out_floats > ws_floats)
    return sgemm(ta, tb, M, N, K, A, lda, B, ldb, C, ldc, bias, accumulate, batch, sA, sB, sC, sBias,
                 0, stream);
  const int splits = cdiv(K, kc);
  if (kc == 64) {
    if (tb) launch_skinny<64, true>(A, lda, B, ldb, ws, M, N, K, batch, sA, sB, stream);
    else launch_skinny<64, false>(A, lda, B, ldb, ws, M, N, K, batch, sA, sB, stream);
  } else {
    if (tb) launch_skinny<128, true>(A, lda, B, ldb, ws, M, N, K, batch, sA, sB, stream);
    else launch_skinny<128, false>(A, lda, B, ldb, ws, M, N, K, batch, sA, sB, stream);
  }
  const long total = (long)out_floats;
  hipLaunchKernelGGL(splitk_reduce_kernel, dim3((int)(cdiv(total, 256) > 2048 ? 2048 : cdiv(total, 256))),
                     dim3(256), 0, stream, ws, splits, M, N * batch, C, ldc, bias, accumulate);
  CAPNET_LAUNCH_CHECK();
  return kOk;
}

// The K-chunk partial products only: slab[k][M][N], k < *n_slabs, left for the consumer to sum in
// slab order (the recurrent backward step folds the sum into its gate kernel). *n_slabs = 0 when
// the shape does not qualify (the caller then uses sgemm_splitk).
int sgemm_splitk_slabs(bool tb, int M, int N, int K, const float* A, long lda, const float* B, long ldb,
                       float* ws, size_t ws_floats, int* n_slabs, hipStream_t stream) {
  *n_slabs = 0;
  if (M <= 0 || N <= 0) return kOk;
  const bool ok = ws && M <= 128 && K % 4 == 0 && lda % 4 == 0 && aligned16(A) && aligned16(B) &&
                  ldb % 4 == 0 && (tb || N % 4 == 0) && N >= 4 && K >= 64;
  if (!ok) return kOk;
  const long tiles = (long)cdiv(M, 64) * cdiv(N, 64);
  int kc = tiles * cdiv(K, 128) < 200 ? 64 : 128;
  if ((size_t)cdiv(K, kc) * M * N > ws_floats) kc = 128;
  if ((size_t)cdiv(K, kc) * M * N > ws_floats) return kOk;
  if (kc == 64) {
    if (tb) launch_skinny<64, true>(A, lda, B, ldb, ws, M, N, K, 1, 0, 0, stream);
    else launch_skinny<64, false>(A, lda, B, ldb, ws, M, N, K, 1, 0, 0, stream);
  } else {
    if (tb) launch_skinny<128, true>(A, lda, B, ldb, ws, M, N, K, 1, 0, 0, stream);
    else launch_skinny<128, false>(A, lda, B, ldb, ws, M, N, K, 1, 0, 0, stream);
  }
  CAPNET_LAUNCH_CHECK();
  *n_slabs = cdiv(K, kc);
  return kOk;
}

int sgemm_splitk(bool ta, bool tb, int M, int N, int K, const float* A, long lda, const float* B,
                 long ldb, float* C, long ldc, const float* bias, int accumulate, float* ws,
                 size_t ws_floats, hipStream_t stream, int* counters, size_t n_counters) {
  return sgemm_splitk_batched(ta, tb, M, N, K, A, lda, B, ldb, C, ldc, bias, accumulate, 1, 0, 0, 0, 0,
                              ws, ws_floats, stream, counters, n_counters);
}

}  // namespace capnet
