// Plain / batched f32 GEMM on the f32-input MFMA (decoder projections, vocabulary
// projection, encoder head, all backward GEMMs). Row-major, arbitrary leading dimensions:
//   C[M x N] (+)= op(A) . op(B) + bias[N]
//   transA = 0: A stored [M][K]      transA = 1: A stored [K][M]
//   transB = 0: B stored [K][N]      transB = 1: B stored [N][K]   (nn.Linear weight)
// Replaces the torch.nn.Linear / autograd matmuls of the reference decoders
// (stylenet/model.py:119-150,189-194; nic/model.py:77,105-113).
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
    for (int k = 0; k < splitk; ++k) s += slab[(long)k * total + i];
    if (bias) s += bias[n];
    float* o = out + (long)m * ldc + n;
    *o = accumulate ? *o + s : s;
  }
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
  } else {
    CAPNET_REQUIRE(false, "sgemm: unknown tile %d", tile);
  }
  CAPNET_LAUNCH_CHECK();
  return kOk;
}

}  // namespace capnet

namespace capnet {

// Skinny products of the recurrence (M <= 64 rows per time step): too few output tiles to fill
// 256 CUs, so K is split across workgroups into slabs in `ws` and summed in a fixed order.
// Falls back to sgemm when there are enough tiles or no workspace.
int sgemm_splitk(bool ta, bool tb, int M, int N, int K, const float* A, long lda, const float* B,
                 long ldb, float* C, long ldc, const float* bias, int accumulate, float* ws,
                 size_t ws_floats, hipStream_t stream) {
  if (M == 0 || N == 0) return kOk;
  const long tiles = (long)cdiv(M, 64) * cdiv(N, 64);
  int splitk = 1;
  if (tiles < 96 && K >= 256) {
    splitk = (int)(192 / tiles);
    if (splitk > K / 64) splitk = K / 64;
    if (splitk > 32) splitk = 32;
  }
  while (splitk > 1 && (size_t)splitk * M * N > ws_floats) --splitk;
  if (splitk <= 1 || !ws)
    return sgemm(ta, tb, M, N, K, A, lda, B, ldb, C, ldc, bias, accumulate, 1, 0, 0, 0, 0, 0, stream);
  CAPNET_REQUIRE(A && B && C, "sgemm_splitk: null operand");
  GemmArgs g;
  g.A = A; g.B = B; g.C = ws; g.bias = nullptr;
  g.M = M; g.N = N; g.K = K;
  g.lda = lda; g.ldb = ldb; g.ldc = N;
  g.sA = g.sB = g.sC = g.sBias = 0;
  g.accumulate = 0;
  g.splitk = splitk;
  g.kchunk = cdiv(cdiv(K, splitk), 16) * 16;
  g.splitk = cdiv(K, g.kchunk);
  g.tiles_m = cdiv(M, 64);
  g.tiles_n = cdiv(N, 64);
  const bool vec = aligned16(A) && aligned16(B) && (lda % 4 == 0) && (ldb % 4 == 0);
  dispatch_layout<64, 64, 16>(g, 1, ta, tb, vec, stream);
  const long total = (long)M * N;
  hipLaunchKernelGGL(splitk_reduce_kernel, dim3((int)(cdiv(total, 256) > 1024 ? 1024 : cdiv(total, 256))),
                     dim3(256), 0, stream, ws, g.splitk, M, N, C, ldc, bias, accumulate);
  CAPNET_LAUNCH_CHECK();
  return kOk;
}

}  // namespace capnet
