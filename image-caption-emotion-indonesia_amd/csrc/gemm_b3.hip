// f32 GEMM on the bf16 matrix cores with fp32-grade results: every fp32 operand is cut into THREE bf16 pieces
// x = h + m + l -- exactly: a bf16 keeps the 8 leading bits of what is left, and 3 x 8 = fp32's 24 -- and a product is the
// six piece products  l h' + h l' + m m' + m h' + h m' + h h'  accumulated in fp32; the three dropped ones (m l', l m',
// l l') are below 2^-23 of the product. bf16 has fp32's exponent range, so unlike the trunk's two-f16-piece scheme
// (conv_f16x3.hip: three products, but a power-of-two prescale per tensor derived from BatchNorm bounds) nothing has to be
// known about the operands' magnitudes: gradients of 1e-9 and activations of 1e+4 go through the same code. Six products on
// a pipe whose dense peak is 16 x the f32 MFMA's leave 2.65 x the f32 peak (157 -> 417 TFLOP/s); the decoders' products --
// vocabulary projection and its two gradients, the hoisted V / S / U chains, encoder_att -- ran at 65-108 TFLOP/s on
// gemm_f32_kernel / nt_dma_kernel.
//   C[M x N] (+)= op(A) . op(B) + bias[N], row-major, leading dimensions as capnet_sgemm's; batched over blockIdx.y.
// A workgroup (256 threads, 2 x 2 waves of 64 x 64) owns a 128 x 128 tile; per 32-k step the two operand tiles go
// global -> registers (one step ahead) -> split -> LDS as three planes in the MFMA's fragment order ([plane][16-row
// block][k quarter][row][8 k]: a fragment is 1 KB contiguous, one ds_read_b128 per lane), then 16 tiles x 6
// v_mfma_f32_16x16x32_bf16 per wave. 48 KB of LDS and ~100 VGPRs: three workgroups per CU, so that one's staging arithmetic
// runs beside the others' MFMAs.
#include "common.h"
#include "mfma_core.h"
#include "kernels.h"

namespace capnet {
namespace {

typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef unsigned short us4 __attribute__((ext_vector_type(4)));

constexpr int QM = 128, QN = 128, QK = 32;
#ifndef B3_OCC
#define B3_OCC 2               // workgroups per CU the register budget is set for (LDS allows three)
#endif
constexpr int kPlaneB = (QM / 16) * 1024;          // bytes of one plane of one operand tile (8 fragments of 1 KB)
constexpr int kOperandB = 3 * kPlaneB;             // 24 KB

struct B3Args {
  const float* A; const float* B; float* C; const float* bias;
  int M, N, K;
  long lda, ldb, ldc, sA, sB, sC, sBias;
  int accumulate, tiles_m, tiles_n;
  int splits, steps_per_split;   // K cut over blockIdx.z: partials to `slab`, summed by reduce_slabs
  float* slab;
};

// Byte offset of the 16-B cell (16-row block, k quarter, row) of a plane. A fragment (block, all four quarters) is 1 KB read by
// one ds_read_b128 per lane; inside a quarter's 256 B the rows are rotated by 4 quarter + block: a staging wave writes 8-B
// halves of cells whose (quarter, row) or (block, row) differ by multiples that would otherwise land on the same banks
// (SQ_LDS_BANK_CONFLICT was 0.6-0.8 of the LDS cycles without the rotation).
__device__ __forceinline__ int b3_cell(int blk, int kq, int row) {
  return blk * 1024 + kq * 256 + ((row + 4 * kq + blk) & 15) * 16;
}

// x = h + m + l, each a bf16 (kept as the upper half of an fp32 word): exact for every finite x whose third piece is not
// below bf16's subnormal range
__device__ __forceinline__ void b3_split(float x, unsigned short& h, unsigned short& m, unsigned short& l) {
  const unsigned hb = __float_as_uint(x) & 0xffff0000u;
  const float r1 = x - __uint_as_float(hb);
  const unsigned mb = __float_as_uint(r1) & 0xffff0000u;
  const float r2 = r1 - __uint_as_float(mb);
  h = (unsigned short)(hb >> 16);
  m = (unsigned short)(mb >> 16);
  l = (unsigned short)(__float_as_uint(r2) >> 16);
}

// One operand tile of 128 rows (m or n) x 32 k. KC: the operand is K-contiguous in memory ([rows][K]); else it is
// row-contiguous ([K][rows]). Either way a thread fetches four 16-B pieces per step and ends up with, for four
// (row, 4 consecutive k) cells, the 8 bytes of each plane. Rows past the matrix are fetched from its last rows (their
// products are never stored); only the k past K of the LAST step are zeroed (TAIL), so that the steps before it are
// address increment + load, nothing else.
template <bool KC>
struct B3Loader {
  const float* p[4];       // this thread's four pieces of the current step
  long step;               // elements from one step to the next
  int k0;                  // this thread's first k inside a step (KC) / its four k are k0 .. k0 + 3 (!KC)
  int row0;                // first row inside the tile (KC: + 32 i; !KC: rows row0 .. row0 + 3)
  int cell[4];             // byte offsets of this thread's four 8-B cells inside a plane of the tile image
  __device__ __forceinline__ void init(const float* base, long ld, int rows, int r0, int tid) {
    if (KC) {              // thread: row = tid >> 3 (+ 32 i), k = 4 (tid & 7)
      row0 = tid >> 3;
      k0 = 4 * (tid & 7);
#pragma unroll
      for (int i = 0; i < 4; ++i) p[i] = base + (long)min(r0 + row0 + 32 * i, rows - 1) * ld + k0;
      step = QK;
#pragma unroll
      for (int c = 0; c < 4; ++c) cell[c] = b3_cell((row0 + 32 * c) >> 4, k0 >> 3, (row0 + 32 * c) & 15) + (k0 & 4) * 2;
    } else {               // thread: rows 4 (tid & 31) .. + 3 (rows % 4 == 0), k = 4 (tid >> 5) + i
      row0 = 4 * (tid & 31);
      k0 = 4 * (tid >> 5);
#pragma unroll
      for (int i = 0; i < 4; ++i) p[i] = base + (long)(k0 + i) * ld + min(r0 + row0, rows - 4);
      step = (long)QK * ld;
#pragma unroll
      for (int c = 0; c < 4; ++c) cell[c] = b3_cell((row0 + c) >> 4, k0 >> 3, (row0 + c) & 15) + (k0 & 4) * 2;
    }
  }
  __device__ __forceinline__ void advance(int steps) {
#pragma unroll
    for (int i = 0; i < 4; ++i) p[i] += (long)steps * step;
  }
  // the current step (k_left = K - its first k), then on to the next one
  template <bool TAIL>
  __device__ __forceinline__ void fetch(int k_left, f32x4 (&v)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const bool ok = !TAIL || (KC ? k0 : k0 + i) < k_left;      // K % 4 == 0 (KC): a 16-B piece is inside or outside as a whole
      v[i] = ok ? *reinterpret_cast<const f32x4*>(p[i]) : f32x4{0.f, 0.f, 0.f, 0.f};
      p[i] += step;
    }
  }
  // registers -> the three planes of the tile image at `img`
  __device__ __forceinline__ void stage(unsigned char* img, const f32x4 (&v)[4]) const {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      // KC: cell c = piece c (row row0 + 32 c, k k0 .. k0 + 3); else cell c = row row0 + c, its four k are v[0..3][c]
      us4 h, m, l;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        unsigned short x0, x1, x2;
        b3_split(KC ? v[c][e] : v[e][c], x0, x1, x2);
        h[e] = x0; m[e] = x1; l[e] = x2;
      }
      unsigned char* d = img + cell[c];
      *reinterpret_cast<us4*>(d) = h;
      *reinterpret_cast<us4*>(d + kPlaneB) = m;
      *reinterpret_cast<us4*>(d + 2 * kPlaneB) = l;
    }
  }
};

template <bool A_KC, bool B_KC>
__global__ __launch_bounds__(256, B3_OCC) void gemm_b3_kernel(const B3Args g) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * kOperandB];
  unsigned char* const a_img = lds;
  unsigned char* const b_img = lds + kOperandB;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int id = xcd_remap((int)blockIdx.x, g.tiles_m * g.tiles_n);
  const int tm = id / g.tiles_n, tn = id - tm * g.tiles_n;
  const int m0 = tm * QM, n0 = tn * QN;
  const int z = blockIdx.y;
  const float* A = g.A + (long)z * g.sA;
  const float* B = g.B + (long)z * g.sB;
  float* C = g.C + (long)z * g.sC;
  const float* bias = g.bias ? g.bias + (long)z * g.sBias : nullptr;

  B3Loader<A_KC> la;
  B3Loader<B_KC> lb;
  la.init(A, g.lda, g.M, m0, tid);
  lb.init(B, g.ldb, g.N, n0, tid);

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // this wave's fragments: A row blocks 4 wm .. 4 wm + 3, B column blocks 4 wn .. 4 wn + 3 (lane = 16 quarter + row)
  const unsigned char* a_rd[4];
  const unsigned char* b_rd[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    a_rd[i] = a_img + b3_cell(4 * wm + i, lane >> 4, lane & 15);
    b_rd[i] = b_img + b3_cell(4 * wn + i, lane >> 4, lane & 15);
  }

  auto mma = [&]() __attribute__((always_inline)) {
    // D^T = B-fragment (as the MFMA's A operand) x A-fragment: a lane then holds four consecutive COLUMNS of one row of C
    bf8 bh[4], bm[4], bl[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      bh[j] = *reinterpret_cast<const bf8*>(b_rd[j]);
      bm[j] = *reinterpret_cast<const bf8*>(b_rd[j] + kPlaneB);
      bl[j] = *reinterpret_cast<const bf8*>(b_rd[j] + 2 * kPlaneB);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const bf8 ah = *reinterpret_cast<const bf8*>(a_rd[i]);
      const bf8 am = *reinterpret_cast<const bf8*>(a_rd[i] + kPlaneB);
      const bf8 al = *reinterpret_cast<const bf8*>(a_rd[i] + 2 * kPlaneB);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bl[j], ah, acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh[j], al, acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bm[j], am, acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bm[j], ah, acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh[j], am, acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh[j], ah, acc[i][j], 0, 0, 0);
      }
    }
  };
  // steps [k_begin, k_end) of this workgroup (split K: blockIdx.z); whole steps fetch unconditionally, a last partial one
  // zeroes what lies past K
  const int steps_all = (g.K + QK - 1) / QK;
  const int s_begin = (int)blockIdx.z * g.steps_per_split, s_end = min(steps_all, s_begin + g.steps_per_split);
  const int full = g.K / QK;                           // steps before `full` are whole
  if (s_begin > 0) {
    la.advance(s_begin);
    lb.advance(s_begin);
  }
  // operands run TWO steps ahead in two named register sets (a workgroup alone on its CU otherwise waits for every step's
  // loads: 1.8 us per step for 0.64 us of MFMAs)
  f32x4 va0[4], vb0[4], va1[4], vb1[4];
  auto fetch = [&](int st, f32x4 (&va)[4], f32x4 (&vb)[4]) __attribute__((always_inline)) {
    if (st < full) { la.template fetch<false>(0, va); lb.template fetch<false>(0, vb); }
    else { la.template fetch<true>(g.K - st * QK, va); lb.template fetch<true>(g.K - st * QK, vb); }
  };
  fetch(s_begin, va0, vb0);
  if (s_begin + 1 < s_end) fetch(s_begin + 1, va1, vb1);
  for (int st = s_begin; st < s_end; st += 2) {
    if (st > s_begin) __syncthreads();                 // every wave is through with the previous step's images
    la.stage(a_img, va0);
    lb.stage(b_img, vb0);
    __syncthreads();
    if (st + 2 < s_end) fetch(st + 2, va0, vb0);
    mma();
    if (st + 1 < s_end) {
      __syncthreads();
      la.stage(a_img, va1);
      lb.stage(b_img, vb1);
      __syncthreads();
      if (st + 3 < s_end) fetch(st + 3, va1, vb1);
      mma();
    }
  }
  // a split's partial goes to slab[split][M][N] as it is; bias and accumulation are reduce_slabs' then
  const bool last = g.splits == 1;
  if (!last) { C = g.slab + (long)blockIdx.z * g.M * g.N; bias = nullptr; }
  const long ldo = last ? g.ldc : (long)g.N;
  const bool accumulate = last && g.accumulate;
  // D^T[n][m]: lane (q = lane >> 4, r = lane & 15) holds row m = r of the A block, columns n = 4 q .. 4 q + 3 of the B block
  const int r = lane & 15, q = lane >> 4;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + (4 * wm + i) * 16 + r;
    if (m >= g.M) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + (4 * wn + j) * 16 + 4 * q;
      if (n >= g.N) continue;
      float* o = C + (long)m * ldo + n;
      f32x4 v = acc[i][j];
      if (n + 3 < g.N) {
        if (bias) v += *reinterpret_cast<const f32x4*>(bias + n);
        if (accumulate) v += *reinterpret_cast<const f32x4*>(o);
        *reinterpret_cast<f32x4*>(o) = v;
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (n + e < g.N) o[e] = v[e] + (bias ? bias[n + e] : 0.f) + (accumulate ? o[e] : 0.f);
      }
    }
  }
}

}  // namespace

// shapes and pointers the kernel takes: 16-B pieces along each operand's contiguous dimension, 16-B rows of C
bool sgemm_b3_eligible(bool ta, bool tb, int M, int N, int K, const float* A, long lda, const float* B, long ldb,
                       const float* C, long ldc, const float* bias, int batch, long sA, long sB, long sC, long sBias) {
  if (M < 1 || N < 1 || K < 1 || batch < 1 || batch > 65535) return false;
  const bool a_kc = !ta, b_kc = tb;
  if (!aligned16(A) || !aligned16(B) || !aligned16(C) || (bias && !aligned16(bias))) return false;
  if (lda % 4 || ldb % 4 || ldc % 4 || sA % 4 || sB % 4 || sC % 4 || sBias % 4) return false;
  if ((a_kc || b_kc) && K % 4) return false;           // a K-contiguous operand is fetched in 16-B pieces of k ...
  if ((!a_kc && M % 4) || (!b_kc && N % 4)) return false;   // ... a row-contiguous one in 16-B pieces of rows
  return true;
}

// ws (optional, batch == 1): room for split-K partials -- a product with few 128 x 128 tiles and a long K is cut over
// blockIdx.z and summed by reduce_slabs
int sgemm_b3(bool ta, bool tb, int M, int N, int K, const float* A, long lda, const float* B, long ldb, float* C, long ldc,
             const float* bias, int accumulate, int batch, long sA, long sB, long sC, long sBias, hipStream_t stream,
             float* ws, size_t ws_floats) {
  CAPNET_REQUIRE(sgemm_b3_eligible(ta, tb, M, N, K, A, lda, B, ldb, C, ldc, bias, batch, sA, sB, sC, sBias),
                 "sgemm_b3: operands not eligible");
  B3Args g;
  g.A = A; g.B = B; g.C = C; g.bias = bias;
  g.M = M; g.N = N; g.K = K;
  g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.sA = sA; g.sB = sB; g.sC = sC; g.sBias = sBias;
  g.accumulate = accumulate;
  g.tiles_m = cdiv(M, QM); g.tiles_n = cdiv(N, QN);
  const int steps = cdiv(K, QK), tiles = g.tiles_m * g.tiles_n;
  int splits = 1;
  if (ws && batch == 1 && tiles < 192 && steps >= 16) {
    splits = min(cdiv(512, tiles), steps / 8);
    while (splits > 1 && (size_t)splits * M * N > ws_floats) --splits;
  }
  g.steps_per_split = cdiv(steps, splits);
  g.splits = cdiv(steps, g.steps_per_split);
  g.slab = ws;
  const dim3 grid(tiles, batch, g.splits), block(256);
  const bool a_kc = !ta, b_kc = tb;
  if (a_kc && b_kc) hipLaunchKernelGGL((gemm_b3_kernel<true, true>), grid, block, 0, stream, g);
  else if (a_kc) hipLaunchKernelGGL((gemm_b3_kernel<true, false>), grid, block, 0, stream, g);
  else if (b_kc) hipLaunchKernelGGL((gemm_b3_kernel<false, true>), grid, block, 0, stream, g);
  else hipLaunchKernelGGL((gemm_b3_kernel<false, false>), grid, block, 0, stream, g);
  CAPNET_LAUNCH_CHECK();
  if (g.splits > 1) return reduce_slabs(ws, g.splits, M, N, C, ldc, bias, accumulate, stream);
  return kOk;
}

}  // namespace capnet
