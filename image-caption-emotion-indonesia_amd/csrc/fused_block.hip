// A bottleneck block's conv3 + BatchNorm3 + residual + ReLU and the NEXT block's conv1 as ONE kernel that never writes
// or reads y3 (torchvision Bottleneck.forward via stylenet/model.py:15-18,24).
//
// Train-mode BatchNorm needs the statistics of y3 = a2 . W3^T over all M rows before the first output can be formed,
// which is why conv3 used to materialise y3 (the block's widest tensor: written, then read back by the tail). But y3 is
// LINEAR in a2 = relu(bn2(y2)), so its per-channel sums follow from the second moments of conv3's INPUT:
//     sum_m y3[m, c]   = sum_k colsum[k] W3[c, k]            colsum[k] = sum_m a2[m, k]
//     sum_m y3[m, c]^2 = w_c^T G w_c                         G = a2^T a2     (K x K, K = Cin of conv3 = Cout / 4)
// G costs 2 M K^2 flops -- a quarter of conv3 (an eighth with its symmetry) -- and reads y2 only. With the statistics known
// up front, y3 is computed ONCE, chunk by chunk, inside the kernel that consumes it:
//     per 32-channel chunk:  P   = a2[rows, :] . W3[chunk, :]^T        (phase A, K = MID)
//                            out = relu(bn3(P) + identity)  -> HBM (the next tail's identity), split to f16 planes
//                            acc += out_chunk . W1[:, chunk]^T          (phase B, the next block's conv1)
// Per block and row: y2 read twice, identity read, out written, y1 written -- 2.75 "units" instead of 4.5 -- and conv3's
// matrix work is done once, not twice (the recompute variant sized in DESIGN r3 7 did it twice).
//
// Arithmetic: the split-f16 scheme of conv_f16x3.hip (x 2^e = h + l in f16, three products, fp32 accumulate) on
// v_mfma_f32_16x16x32_f16. A wave owns 16-row strips. Phase A is computed TRANSPOSED (A operand = W3 rows = channels,
// B operand = a2^T) so that its D tile -- lane (g = l >> 4, n = l & 15) holds channels 4 g + i of row n -- IS the A
// operand of phase B (row n, k slots of lane group g) without leaving the registers: k slot j of lane group g stands for
// channel 16 (j >> 2) + 4 g + (j & 3) of the chunk in BOTH weight images (fb_kslot). a2 of the wave's rows stays in
// registers for the whole tile (folded and split once); W3 / W1 chunks stream through a 4-slot LDS ring by LDS-DMA, three
// phases ahead, one barrier per phase; the identity rows come two chunks ahead by hand-issued loads (counted vmcnt;
// tools/isa_inflight_check.py checks the discipline on the shipped ISA).
#include <cstdlib>
#include <type_traits>

#include "common.h"
#include "mfma_core.h"
#include "kernels.h"

namespace capnet {
namespace {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef float f2 __attribute__((ext_vector_type(2)));

constexpr int kFHdr = 4;                  // image header words: [0] ew, [1] bits of max |w| (pack scratch)

__host__ __device__ constexpr int fb_kslot(int g, int j) { return 16 * (j >> 2) + 4 * g + (j & 3); }

__device__ __forceinline__ void fb_split4(const f32x4 v, h4& h, h4& l) {
  const f2 a = {v[0], v[1]}, b = {v[2], v[3]};
  const h2 ha = __builtin_convertvector(a, h2), hb = __builtin_convertvector(b, h2);
  const f2 ra = a - __builtin_convertvector(ha, f2), rb = b - __builtin_convertvector(hb, f2);
  const h2 la = __builtin_convertvector(ra, h2), lb = __builtin_convertvector(rb, h2);
  h = h4{ha[0], ha[1], hb[0], hb[1]};
  l = h4{la[0], la[1], lb[0], lb[1]};
}
__device__ __forceinline__ h8 fb_cat(const h4 a, const h4 b) { return h8{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]}; }

// ---------------------------------------------------------------------------------------------------------------------
// weight images
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void fb_absmax_kernel(const float* __restrict__ w, unsigned* __restrict__ hdr, long n) {
  float m = 0.f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) m = fmaxf(m, fabsf(w[i]));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  if ((threadIdx.x & 63) == 0 && m > 0.f) atomicMax(hdr + 1, __float_as_uint(m));
}
__device__ __forceinline__ int fb_weight_shift(unsigned absmax_bits) {      // max |w| 2^ew in [2^13, 2^14) (conv_f16x3.hip)
  if (absmax_bits == 0u) return 0;
  const int e = (int)((absmax_bits >> 23) & 0xffu) - 127;
  const int ew = 13 - e;
  return ew < -100 ? -100 : (ew > 100 ? 100 : ew);
}

// role 0: W3 [C][MID] (conv3, the A operand of phase A): cells [chunk C/32][ks MID/32][blk 2][plane 2][lane 64] of 8 halfs
//         = W3[32 chunk + 16 blk + (lane & 15)][32 ks + kslot(lane >> 4, j)], followed by an fp32 copy of W3 (the
//         statistics' quadratic forms read the weights as they are; laid out [C / 16][MID][16]);
// role 1: W1 [MID][C] (the next conv1, the B operand of phase B): cells [chunk C/32][nb MID/16][plane 2][lane 64]
//         = W1[16 nb + (lane & 15)][32 chunk + kslot(lane >> 4, j)].
__global__ __launch_bounds__(256) void fb_pack_kernel(const float* __restrict__ w, unsigned* __restrict__ img, int C, int MID,
                                                      int role, int wide) {
  const int ew = fb_weight_shift(img[1]);
  if (blockIdx.x == 0 && threadIdx.x == 0) { img[0] = (unsigned)ew; img[2] = (unsigned)wide; }
  const float ws = ldexpf(1.f, ew);
  const long cells = (long)C * MID / 8 * 2;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < cells; i += (long)gridDim.x * blockDim.x) {
    long r = i;
    const int lane = (int)(r & 63); r >>= 6;
    const int plane = (int)(r & 1); r >>= 1;
    const float* src;
    int koff[8];
    if (!wide) {
      const int g = lane >> 4, n = lane & 15;
#pragma unroll
      for (int j = 0; j < 8; ++j) koff[j] = fb_kslot(g, j);
      if (role == 0) {
        const int blk = (int)(r & 1); r >>= 1;
        const int ks = (int)(r % (MID / 32));
        const int chunk = (int)(r / (MID / 32));
        src = w + (long)(32 * chunk + 16 * blk + n) * MID + 32 * ks;
      } else {
        const int nb = (int)(r % (MID / 16));
        const int chunk = (int)(r / (MID / 16));
        src = w + (long)(16 * nb + n) * C + 32 * chunk;
      }
    } else {
      // 32-row strips on v_mfma_f32_32x32x16_f16 (fb_fused_wide_kernel): lane (n = lane & 31, h = lane >> 5)
      const int h = lane >> 5, n = lane & 31;
      if (role == 0) {
        // cells [chunk][k16 step][plane][lane]: W3[32 chunk + n][16 ks + 8 h + j]
        const int ks = (int)(r % (MID / 16));
        const int chunk = (int)(r / (MID / 16));
        src = w + (long)(32 * chunk + n) * MID + 16 * ks;
#pragma unroll
        for (int j = 0; j < 8; ++j) koff[j] = 8 * h + j;
      } else {
        // cells [chunk][nb MID / 32][step 2][plane][lane]: W1[32 nb + n][32 chunk + 16 s + 8 (j >> 2) + 4 h + (j & 3)] -- the
        // channel order in which phase A's accumulator registers 8 s .. 8 s + 7 hold a row's channels
        const int st = (int)(r & 1); r >>= 1;
        const int nb = (int)(r % (MID / 32));
        const int chunk = (int)(r / (MID / 32));
        src = w + (long)(32 * nb + n) * C + 32 * chunk + 16 * st;
#pragma unroll
        for (int j = 0; j < 8; ++j) koff[j] = 8 * (j >> 2) + 4 * h + (j & 3);
      }
    }
    unsigned out[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float x0 = src[koff[2 * q]] * ws, x1 = src[koff[2 * q + 1]] * ws;
      const _Float16 h0 = (_Float16)x0, h1 = (_Float16)x1;
      const _Float16 l0 = (_Float16)(x0 - (float)h0), l1 = (_Float16)(x1 - (float)h1);
      const h2 p = plane == 0 ? h2{h0, h1} : h2{l0, l1};
      out[q] = __builtin_bit_cast(unsigned, p);
    }
    unsigned* dst = img + kFHdr + i * 4;
    dst[0] = out[0]; dst[1] = out[1]; dst[2] = out[2]; dst[3] = out[3];
  }
  if (role == 0) {
    // fp32 copy for the statistics' quadratic forms, 16 channels interleaved: cp[c / 16][k][c % 16] = W3[c][k] (the 16
    // weights one k of a channel group needs sit in one 64-B line: scalar loads)
    float* cp = reinterpret_cast<float*>(img + kFHdr + (long)C * MID);
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < (long)C * MID; i += (long)gridDim.x * blockDim.x) {
      const int c = (int)(i / MID), k = (int)(i - (long)c * MID);
      cp[((long)(c >> 4) * MID + k) * 16 + (c & 15)] = w[i];
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// statistics of y3 from the second moments of conv3's input
// ---------------------------------------------------------------------------------------------------------------------
// rows of a slice (one workgroup): 256 where a slice's partial result is several blocks (K >= 128), 512 for K = 64
static inline int fb_gram_rows(int K) { return K == 64 ? 512 : 256; }

struct GArgs {
  const float* y2; const float* s2; const float* t2;
  float* gp;            // [slices][pairs][64][64]
  float* cs;            // [slices][K]
  int M, K, in_exp, slices, pairs, slice_rows;
};

// A workgroup takes a slice of rows and ALL channels: it stages a step of rows ONCE -- a2 = relu(y2 s2 + t2) 2^e split
// into f16 planes, image [plane][8-row group][channel][8 halfs]: a cell is 8 ROWS of one channel = the k slots of both
// operands of v_mfma_f32_32x32x16_f16 with k = row -- and forms every pair (bi <= bj) of 64-channel blocks of
// G = a2^T a2 from it (three products, fp32 accumulate; 2 x 2 waves, a 32 x 32 quadrant of each pair per wave:
// K / 64 (K / 64 + 1) / 2 accumulators). Round 4's first form gave each pair its own workgroup and re-staged the rows per
// pair: 26 VALU instructions per MFMA and 250 workgroups of 16 us where this is 49 of about the same length. A thread
// stages 8 rows x 4 channels per item (eight 16-B loads); the next step's rows are requested before this step's MFMAs.
template <int K>
__global__ __launch_bounds__(256) void fb_gram_kernel(const GArgs g) {
  constexpr int NBLK = K / 64, NPAIR = NBLK * (NBLK + 1) / 2;
  constexpr int SR = K == 64 ? 128 : 64, NRG = SR / 8;               // rows and 8-row groups of a step
  constexpr int NCQ = K / 4, NIT = NCQ * NRG / 256;                  // 4-channel groups; items per thread and step
  constexpr int kPlane = NRG * K * 16;
  static_assert(NIT >= 1 && NCQ * NRG == NIT * 256, "staging items");
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * kPlane];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, li = lane & 31, lh = lane >> 5;
  const int slice = blockIdx.x, row0 = slice * g.slice_rows;
  const int rows = min(g.slice_rows, g.M - row0);
  const int nst = (rows + SR - 1) / SR;
  const float iscale = ldexpf(1.f, g.in_exp);
  const int chq = tid % NCQ, rgb = tid / NCQ;                         // item q: row group rgb + q (256 / NCQ)
  const f32x4 sc = *reinterpret_cast<const f32x4*>(g.s2 + 4 * chq), sh = *reinterpret_cast<const f32x4*>(g.t2 + 4 * chq);
  f32x4 colsum = {0.f, 0.f, 0.f, 0.f};
  f32x16 acc[NPAIR];
#pragma unroll
  for (int p = 0; p < NPAIR; ++p)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[p][r] = 0.f;
  f32x4 v[NIT][8];
  auto fetch = [&](int st) __attribute__((always_inline)) {
#pragma unroll
    for (int q = 0; q < NIT; ++q)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int m = row0 + st * SR + (rgb + q * (256 / NCQ)) * 8 + e;
        v[q][e] = *reinterpret_cast<const f32x4*>(g.y2 + (long)min(m, g.M - 1) * K + 4 * chq);
      }
  };
  auto stage = [&](int st) __attribute__((always_inline)) {
#pragma unroll
    for (int q = 0; q < NIT; ++q) {
      const int rg = rgb + q * (256 / NCQ);
      f32x4 x[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const bool ok = row0 + st * SR + rg * 8 + e < row0 + rows;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const float t = fmaxf(fmaf(v[q][e][c], sc[c], sh[c]), 0.f) * iscale;
          x[e][c] = ok ? t : 0.f;
          colsum[c] += x[e][c];
        }
      }
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const f32x4 a = {x[0][c], x[1][c], x[2][c], x[3][c]}, b = {x[4][c], x[5][c], x[6][c], x[7][c]};
        h4 ha, la, hb, lb;
        fb_split4(a, ha, la);
        fb_split4(b, hb, lb);
        unsigned char* d = lds + (rg * K + 4 * chq + c) * 16;
        *reinterpret_cast<h8*>(d) = fb_cat(ha, hb);
        *reinterpret_cast<h8*>(d + kPlane) = fb_cat(la, lb);
      }
    }
  };
  auto mma = [&]() __attribute__((always_inline)) {
#pragma unroll 2
    for (int k = 0; k < NRG / 2; ++k) {
      const unsigned char* base = lds + ((2 * k + lh) * K + li) * 16;
      h8 ah[NBLK], al[NBLK], bh[NBLK], bl[NBLK];
#pragma unroll
      for (int b = 0; b < NBLK; ++b) {
        ah[b] = *reinterpret_cast<const h8*>(base + (b * 64 + wm * 32) * 16);
        al[b] = *reinterpret_cast<const h8*>(base + (b * 64 + wm * 32) * 16 + kPlane);
        bh[b] = *reinterpret_cast<const h8*>(base + (b * 64 + wn * 32) * 16);
        bl[b] = *reinterpret_cast<const h8*>(base + (b * 64 + wn * 32) * 16 + kPlane);
      }
      int p = 0;
#pragma unroll
      for (int bi = 0; bi < NBLK; ++bi)
#pragma unroll
        for (int bj = bi; bj < NBLK; ++bj, ++p) {
          acc[p] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[bi], bh[bj], acc[p], 0, 0, 0);
          acc[p] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[bi], bl[bj], acc[p], 0, 0, 0);
          acc[p] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[bi], bh[bj], acc[p], 0, 0, 0);
        }
    }
  };
  fetch(0);
  for (int st = 0; st < nst; ++st) {
    stage(st);
    __syncthreads();
    fetch(st + 1 < nst ? st + 1 : st);               // (past the end: the last step's rows again, never staged)
    mma();
    __syncthreads();
  }
  // D[m][n]: m = (r & 3) + 8 (r >> 2) + 4 lh = channel of block bi, n = li = channel of block bj; both operands carried 2^e
  const float osc = ldexpf(1.f, -2 * g.in_exp);
#pragma unroll
  for (int p = 0; p < NPAIR; ++p) {
    float* out = g.gp + ((long)slice * NPAIR + p) * 4096;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      out[m * 64 + wn * 32 + li] = acc[p][r] * osc;
    }
  }
  f32x4* cs_sh = reinterpret_cast<f32x4*>(lds);                // [256 / NCQ row-group lanes][NCQ channel quads]
  cs_sh[tid] = colsum;
  __syncthreads();
  if (tid < NCQ) {
    f32x4 t = cs_sh[tid];
#pragma unroll
    for (int q = 1; q < 256 / NCQ; ++q) t += cs_sh[q * NCQ + tid];
    *reinterpret_cast<f32x4*>(g.cs + (long)slice * K + 4 * tid) = t * ldexpf(1.f, -g.in_exp);
  }
}

// G (full, symmetric) and the column sums, summed over the slices in double, stored fp32: 8 threads share an entry's
// slices, each with eight independent loads in flight per round. (Measured alternatives, K = 64 / 128 / 256: a plain loop
// 7.4 / 6.5 / 12.9 us; 128 entries per workgroup with 32 loads per thread 20 / 22 / 15 us -- the partial blocks of one
// entry lie 16-160 KB apart, more loads per thread only adds pages per thread; this form 4.8 / 9.9 / 13.5 us.)
__global__ __launch_bounds__(256) void fb_gram_reduce_kernel(const float* __restrict__ gp, const float* __restrict__ cs, float* __restrict__ G,
                                                             float* __restrict__ mu, int K, int slices, int pairs) {
  __shared__ double sh[8][33];
  const int e = blockIdx.x * 32 + (threadIdx.x & 31), part = threadIdx.x >> 5;
  const int nG = K * K;
  const float* p = nullptr;
  long stride = 0;
  if (e < nG) {
    int r = e / K, c = e - r * K;
    int bi = r >> 6, bj = c >> 6, m = r & 63, n = c & 63;
    if (bi > bj) { int t = bi; bi = bj; bj = t; t = m; m = n; n = t; }
    const int nb = K / 64;
    const int pair = bi * nb - bi * (bi - 1) / 2 + (bj - bi);
    p = gp + (long)pair * 4096 + m * 64 + n;
    stride = (long)pairs * 4096;
  } else if (e < nG + K) {
    p = cs + (e - nG);
    stride = K;
  }
  double s = 0.0;
  if (p) {
    for (int s0 = part; s0 < slices; s0 += 64) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int sl = s0 + 8 * u;
        v[u] = p[(long)(sl < slices ? sl : part) * stride];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) s += (s0 + 8 * u < slices) ? (double)v[u] : 0.0;
    }
  }
  sh[part][threadIdx.x & 31] = s;
  __syncthreads();
  if (part == 0) {
#pragma unroll
    for (int q = 1; q < 8; ++q) s += sh[q][threadIdx.x & 31];
    if (e < nG) G[e] = (float)s;
    else if (e < nG + K) mu[e - nG] = (float)s;
  }
}

constexpr int kQC = 16;                  // channels per workgroup of the quadratic forms

// channel c: sum = mu . w_c, sumsq = w_c^T G w_c, then bn_finalize_kernel's arithmetic (bn_pool.hip).
// A workgroup takes 16 channels: T [16 x K] = W_g [16 x K] . G [K x K] on v_mfma_f32_16x16x4_f32 (exact fp32 fma chains --
// a sum of K^2 terms of mixed sign carries ~1e-7 sqrt-wise, the grade of the fp32 partial sums the stand-alone statistics
// carry), one 16-column block of G per wave, both operands straight from global memory with every load of a half in flight at
// once (the kernel is a chain of memory round trips otherwise: 32 us with a scalar-operand loop, 3 us like this); then
// sumsq_c = sum_t T[c][t] w_c[t] across lanes and waves. mean / variance / 1/sqrt in double.
template <int K>
__global__ __launch_bounds__(K * 4) void fb_quad_kernel(const float* __restrict__ G, const float* __restrict__ mu,
                                                        const float* __restrict__ w /* [C / 16][K][16] */, int C, double inv_count,
                                                        double unbias, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        float* __restrict__ running_mean, float* __restrict__ running_var,
                                                        float momentum, float eps, float* __restrict__ scale, float* __restrict__ shift,
                                                        float* __restrict__ batch_mean, float* __restrict__ batch_var, int* __restrict__ err) {
  constexpr int NW = K / 16, S = K / 4, SB = S < 32 ? S : 32;
  __shared__ float red[2][kQC][NW];
  const int tid = threadIdx.x, lane = tid & 63, nb = tid >> 6, g4 = lane >> 4, ln = lane & 15, c0 = blockIdx.x * kQC;
  const float* wq = w + (long)blockIdx.x * K * kQC;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  // A[m = channel ln][k = 4 s + g4] = wq[k][ln];  B[k = 4 s + g4][n = column 16 nb + ln] = G[k][16 nb + ln]
#pragma unroll
  for (int s0 = 0; s0 < S; s0 += SB) {
    float a[SB], b[SB];
#pragma unroll
    for (int i = 0; i < SB; ++i) {
      const int k = 4 * (s0 + i) + g4;
      a[i] = wq[k * kQC + ln];
      b[i] = G[(long)k * K + 16 * nb + ln];
    }
#pragma unroll
    for (int i = 0; i < SB; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[i], acc, 0, 0, 0);
  }
  // D[r]: channel 4 g4 + r, column 16 nb + ln
  const int col = 16 * nb + ln;
  const f32x4 wv = *reinterpret_cast<const f32x4*>(wq + col * kQC + 4 * g4);
  const float m = mu[col];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    float q = acc[r] * wv[r], sm = m * wv[r];
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) {
      q += __shfl_xor(q, o);
      sm += __shfl_xor(sm, o);
    }
    if (ln == 0) { red[0][4 * g4 + r][nb] = q; red[1][4 * g4 + r][nb] = sm; }
  }
  __syncthreads();
  if (tid < kQC && c0 + tid < C) {
    const int c = c0 + tid;
    double qq = 0.0, ss = 0.0;
#pragma unroll
    for (int v = 0; v < NW; ++v) { qq += (double)red[0][tid][v]; ss += (double)red[1][tid][v]; }
    if (err && !(qq < __builtin_inf())) atomicOr(err, 8);
    const double mean = ss * inv_count;
    double var = qq * inv_count - mean * mean;
    if (var < 0.0) var = 0.0;
    const float invstd = (float)(1.0 / sqrt(var + (double)eps));
    const float gm = gamma ? gamma[c] : 1.f, bt = beta ? beta[c] : 0.f;
    const float sc = gm * invstd;
    scale[c] = sc;
    shift[c] = bt - (float)mean * sc;
    if (running_mean) {      // (bn_running_blend of bn_pool.hip)
      running_mean[c] = __builtin_fmaf(momentum, (float)mean, (1.f - momentum) * running_mean[c]);
      running_var[c] = __builtin_fmaf(momentum, (float)(var * unbias), (1.f - momentum) * running_var[c]);
    }
    if (batch_mean) {
      batch_mean[c] = (float)mean;
      batch_var[c] = (float)(var * unbias);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// the fused kernel
// ---------------------------------------------------------------------------------------------------------------------
struct FArgs {
  const float* y2; const float* s2; const float* t2;
  const unsigned* w3; const unsigned* w1;
  const float* s3; const float* t3;
  const float* res; const float* sd; const float* td;
  float* out; float* y1; float* part_sum; float* part_sq;
  int M, e3, e1;
  int* err;
};

#ifndef CAPNET_FB_DBG
#define CAPNET_FB_DBG 0        // probes only: 1 = no MFMAs (fragments still read), 2 = no weight DMA (waits and barriers stay),
                               // 3 = neither MFMAs nor fragment reads (DMA, barriers, tail), 4 = as 3 without the weight DMA
#endif
template <int N>
__device__ __forceinline__ void fb_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// NW waves of RS 16-row strips each. NW = 4: one wave per SIMD (up to 512 registers: RS = 2 fits); NW = 8: two waves per
// SIMD at <= 256 registers -- a lone wave issues its own loads, waits and tail arithmetic BETWEEN its MFMAs (SQ counters,
// 4 waves x 32 rows: matrix pipe busy 40 % of the wave's time, 40 % issue stalls, 24 % waits), a second wave fills those gaps.
template <int MID, int RS, int NW>
__global__ __launch_bounds__(64 * NW, NW / 4) void fb_fused_kernel(const FArgs g) {
  constexpr int C = 4 * MID, KS = MID / 32, NB = MID / 16, NCH = C / 32;
  constexpr int SLOT = MID * 128;                     // bytes of one chunk image (either role)
  constexpr int NDMA = SLOT / 1024 / NW;              // 1-KB LDS-DMA instructions per wave and phase
  constexpr int L = 2 * RS;                           // identity loads = out stores per wave and chunk
  constexpr int TR = 16 * RS * NW;                    // rows of a tile
  constexpr int NT = 64 * NW;                         // threads
  static_assert(NDMA >= 1 && 4 * NDMA + 2 * L <= 63, "vmcnt range");
  __shared__ __attribute__((aligned(16))) unsigned char ring[4 * SLOT];
  __shared__ __attribute__((aligned(16))) float par[4][C];          // s3 2^-(ew3 + e3), t3, sd, td
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, gq = lane >> 4, ln = lane & 15;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const unsigned ring0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)ring);
  const int tile0 = (int)blockIdx.x * TR;
  const bool fold_res = g.sd != nullptr;
  const float* const w3img = reinterpret_cast<const float*>(g.w3 + kFHdr);
  const float* const w1img = reinterpret_cast<const float*>(g.w1 + kFHdr);

  // LDS-DMA of chunk `cc` of an image into ring slot `slot`: wave w moves pieces [w NDMA, (w + 1) NDMA)
  auto dma = [&](const float* img, int cc, int slot) __attribute__((always_inline)) {
    const float* src = img + (long)cc * (SLOT / 4);
#pragma unroll
    for (int q = 0; q < ((CAPNET_FB_DBG == 2 || CAPNET_FB_DBG == 4) ? 0 : NDMA); ++q)
      glds16(src, (wave_u * NDMA + q) * 1024 + lane * 16, ring0 + (unsigned)(slot * SLOT + (wave_u * NDMA + q) * 1024));
  };
  // identity rows of chunk cc: lane (gq, ln) takes channels 32 cc + 16 blk + 4 gq .. + 3 of its row in each strip
  unsigned idoff[RS];
#pragma unroll
  for (int s = 0; s < RS; ++s) {
    const int row = min(tile0 + (wave * RS + s) * 16 + ln, g.M - 1);
    idoff[s] = (unsigned)(((long)row * C + 4 * gq) * 4);
  }
  f32x4 idA[RS][2], idB[RS][2];
  auto fetch_id = [&](int cc, f32x4 (&id)[RS][2]) __attribute__((always_inline)) {
#pragma unroll
    for (int s = 0; s < RS; ++s)
#pragma unroll
      for (int blk = 0; blk < 2; ++blk) gload16(id[s][blk], g.res, idoff[s] + (unsigned)((32 * cc + 16 * blk) * 4));
  };

  // ---- prologue: the first two phases' weights, the first two chunks' identity rows, the parameters, a2
  dma(w3img, 0, 0);
  dma(w1img, 0, 1);
  fetch_id(0, idA);
  fetch_id(NCH > 1 ? 1 : 0, idB);
  {
    const float x3 = ldexpf(1.f, -((int)g.w3[0] + g.e3));
    for (int i = tid; i < C; i += NT) {
      par[0][i] = g.s3[i] * x3;
      par[1][i] = g.t3[i];
      par[2][i] = fold_res ? g.sd[i] : 1.f;
      par[3][i] = fold_res ? g.td[i] : 0.f;
    }
  }
  h8 ah[RS][KS], al[RS][KS];
  {
    const float is3 = ldexpf(1.f, g.e3);
#pragma unroll
    for (int s = 0; s < RS; ++s) {
      const int row = min(tile0 + (wave * RS + s) * 16 + ln, g.M - 1);
      const float* p = g.y2 + (long)row * MID + 4 * gq;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        f32x4 v0 = *reinterpret_cast<const f32x4*>(p + 32 * ks), v1 = *reinterpret_cast<const f32x4*>(p + 32 * ks + 16);
        const f32x4 sa = *reinterpret_cast<const f32x4*>(g.s2 + 32 * ks + 4 * gq), sb = *reinterpret_cast<const f32x4*>(g.s2 + 32 * ks + 16 + 4 * gq);
        const f32x4 ta = *reinterpret_cast<const f32x4*>(g.t2 + 32 * ks + 4 * gq), tb = *reinterpret_cast<const f32x4*>(g.t2 + 32 * ks + 16 + 4 * gq);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          v0[e] = fmaxf(fmaf(v0[e], sa[e], ta[e]), 0.f) * is3;
          v1[e] = fmaxf(fmaf(v1[e], sb[e], tb[e]), 0.f) * is3;
        }
        h4 h0, l0, h1, l1;
        fb_split4(v0, h0, l0);
        fb_split4(v1, h1, l1);
        ah[s][ks] = fb_cat(h0, h1);
        al[s][ks] = fb_cat(l0, l1);
      }
    }
  }
  fb_wait_vmcnt<0>();                 // everything of the prologue has landed: the loop's counted waits start from here
  CAPNET_LANDED4(idA[0][0], idA[0][1], idB[0][0], idB[0][1]);
  if constexpr (RS == 2) CAPNET_LANDED4(idA[1][0], idA[1][1], idB[1][0], idB[1][1]);
  __syncthreads();

  f32x4 acc[RS][NB];
#pragma unroll
  for (int s = 0; s < RS; ++s)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) acc[s][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
  const float is1 = ldexpf(1.f, g.e1);

  // The loop runs over PHASES: phase 2 c = A(c) (the y3 chunk), phase 2 c + 1 = B(c) (tail + conv1 chunk); phase p reads ring
  // slot p % 4. Iteration k: every wave has made sure its share of phase k's weights landed, barrier, every wave issues
  // its share of phase k + 2's (into the slot of phase k - 2, free since this barrier), then waves 0-3 run phase k and --
  // with eight waves -- waves 4-7 run phase k - 1: the two waves of a SIMD are ONE PHASE APART, so that one's tail
  // arithmetic and waits lie beside the other's MFMAs instead of beside its tail (both in lockstep: 90 us on the 14 x 14
  // maps, no better than a lone wave of 32 rows).
  // Vector-memory operations of a wave, in issue order, per iteration: D x NDMA, and in an iteration that runs a B phase
  // behind them ST x L, LD x L. The same stream for both groups, one iteration apart.
  //   before barrier k: D(k), issued in iteration k - 2; younger than it: one A and one B iteration      -> NDMA + 2 L
  //   the tail of B(c): LD(c), issued by B(c - 2), four iterations back; younger: D, D ST LD, D, D        -> 4 NDMA + 2 L
#ifndef CAPNET_FB_GB8
#define CAPNET_FB_GB8 2
#endif
  constexpr int GB = NW == 8 ? CAPNET_FB_GB8 : 4;      // fragment groups of a batch
#ifndef CAPNET_FB_FD
#define CAPNET_FB_FD 1
#endif
  constexpr int FD = CAPNET_FB_FD;                     // batches read ahead
  typedef const __attribute__((address_space(3))) unsigned char* lds_bytes;
  const lds_bytes ring3 = (lds_bytes)ring;
  f32x4 d[RS][2];
  auto phase_a = [&](int cc) __attribute__((always_inline)) {
#pragma unroll
    for (int s = 0; s < RS; ++s) { d[s][0] = f32x4{0.f, 0.f, 0.f, 0.f}; d[s][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    // one base per phase, opaque to the compiler: the fragments at immediate offsets (it folded the slot into a constant
    // and computed an address per read otherwise)
    unsigned wa_off = (unsigned)(((2 * cc) & 3) * SLOT + lane * 16);
    asm volatile("" : "+v"(wa_off));
    const lds_bytes wa = ring3 + wa_off;
    constexpr int GA = KS * 2, NBAT = (GA + GB - 1) / GB;          // group = (ks, blk): fragments (group * 2 + plane) KB into the slot
    // fragments are read FD batches ahead of the MFMAs that use them (a batch is GB groups = 3 GB MFMAs: 96 cycles at GB = 2,
    // an LDS read under eight waves' load comes back after 200-300)
    h8 f[FD + 1][GB][2];
    auto read_a = [&](int bb) __attribute__((always_inline)) {
#pragma unroll
      for (int q = 0; q < GB; ++q) {
        const int grp = bb * GB + q;
        if (grp < GA && CAPNET_FB_DBG < 3) {
          f[bb % (FD + 1)][q][0] = *(const __attribute__((address_space(3))) h8*)(wa + (grp * 2) * 1024);
          f[bb % (FD + 1)][q][1] = *(const __attribute__((address_space(3))) h8*)(wa + (grp * 2 + 1) * 1024);
        }
      }
    };
#pragma unroll
    for (int bb = 0; bb < FD; ++bb)
      if (bb < NBAT) read_a(bb);
#pragma unroll
    for (int b = 0; b < NBAT; ++b) {
      if (b + FD < NBAT) read_a(b + FD);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int q = 0; q < GB; ++q) {
        const int grp = b * GB + q, ks = grp >> 1, blk = grp & 1;
        if (grp < GA) {
#pragma unroll
          for (int s = 0; s < RS; ++s) {
            if (CAPNET_FB_DBG >= 3) continue;
            if (CAPNET_FB_DBG == 1) { asm volatile("" :: "v"(f[b % (FD + 1)][q][1]), "v"(f[b % (FD + 1)][q][0])); continue; }
            d[s][blk] = __builtin_amdgcn_mfma_f32_16x16x32_f16(f[b % (FD + 1)][q][1], ah[s][ks], d[s][blk], 0, 0, 0);
            d[s][blk] = __builtin_amdgcn_mfma_f32_16x16x32_f16(f[b % (FD + 1)][q][0], al[s][ks], d[s][blk], 0, 0, 0);
            d[s][blk] = __builtin_amdgcn_mfma_f32_16x16x32_f16(f[b % (FD + 1)][q][0], ah[s][ks], d[s][blk], 0, 0, 0);
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  // `id` holds chunk cc's identity rows and receives chunk cc + 2's
  auto phase_b = [&](int cc, f32x4 (&id)[RS][2]) __attribute__((always_inline)) {
    unsigned wb_off = (unsigned)(((2 * cc + 1) & 3) * SLOT + lane * 16);
    asm volatile("" : "+v"(wb_off));
    const lds_bytes wb = ring3 + wb_off;
    constexpr int NBATB = (NB + GB - 1) / GB;                       // group = nb
    h8 fb[FD + 1][GB][2];
    auto read_b = [&](int bb) __attribute__((always_inline)) {
#pragma unroll
      for (int q = 0; q < GB; ++q) {
        const int nb = bb * GB + q;
        if (nb < NB && CAPNET_FB_DBG < 3) {
          fb[bb % (FD + 1)][q][0] = *(const __attribute__((address_space(3))) h8*)(wb + (nb * 2) * 1024);
          fb[bb % (FD + 1)][q][1] = *(const __attribute__((address_space(3))) h8*)(wb + (nb * 2 + 1) * 1024);
        }
      }
    };
    // the first fragments BEFORE the tail, whose VALU work then covers their round trip
#pragma unroll
    for (int bb = 0; bb < FD; ++bb)
      if (bb < NBATB) read_b(bb);
    __builtin_amdgcn_sched_barrier(0);
    fb_wait_vmcnt<4 * NDMA + 2 * L>();
    if constexpr (RS == 2) CAPNET_LANDED4(id[0][0], id[0][1], id[1][0], id[1][1]);
    else CAPNET_LANDED2(id[0][0], id[0][1]);
    h8 oh[RS], ol[RS];
#pragma unroll
    for (int s = 0; s < RS; ++s) {
      // rows past M repeat row M - 1 bit for bit (a2 and the identity were fetched from it): their stores go to that row
      // as well -- unconditionally, the counted waits rely on every wave issuing exactly L stores per B phase
      const int row = min(tile0 + (wave * RS + s) * 16 + ln, g.M - 1);
      h4 hh[2], ll[2];
#pragma unroll
      for (int blk = 0; blk < 2; ++blk) {
        const int ch = 32 * cc + 16 * blk + 4 * gq;
        const f32x4 sv = *reinterpret_cast<const f32x4*>(&par[0][ch]), tv = *reinterpret_cast<const f32x4*>(&par[1][ch]);
        f32x4 r = id[s][blk], o;
        if (fold_res) {
          const f32x4 dv = *reinterpret_cast<const f32x4*>(&par[2][ch]), ev = *reinterpret_cast<const f32x4*>(&par[3][ch]);
#pragma unroll
          for (int e = 0; e < 4; ++e) r[e] = fmaf(r[e], dv[e], ev[e]);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = fmaxf(fmaf(d[s][blk][e], sv[e], tv[e]) + r[e], 0.f);
        *reinterpret_cast<f32x4*>(g.out + (long)row * C + ch) = o;
        fb_split4(o * is1, hh[blk], ll[blk]);
      }
      oh[s] = fb_cat(hh[0], hh[1]);
      ol[s] = fb_cat(ll[0], ll[1]);
    }
    fetch_id(cc + 2 < NCH ? cc + 2 : NCH - 1, id);
#pragma unroll
    for (int b = 0; b < NBATB; ++b) {
      if (b + FD < NBATB) read_b(b + FD);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int q = 0; q < GB; ++q) {
        const int nb = b * GB + q;
        if (nb < NB) {
#pragma unroll
          for (int s = 0; s < RS; ++s) {
            if (CAPNET_FB_DBG >= 3) continue;
            if (CAPNET_FB_DBG == 1) { asm volatile("" :: "v"(fb[b % (FD + 1)][q][1]), "v"(fb[b % (FD + 1)][q][0])); continue; }
            acc[s][nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ol[s], fb[b % (FD + 1)][q][0], acc[s][nb], 0, 0, 0);
            acc[s][nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(oh[s], fb[b % (FD + 1)][q][1], acc[s][nb], 0, 0, 0);
            acc[s][nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(oh[s], fb[b % (FD + 1)][q][0], acc[s][nb], 0, 0, 0);
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  // the head of iteration k: phase k's weights are in LDS behind the barrier; phase k + 2's are requested
  auto head = [&](int k) __attribute__((always_inline)) {
    // (the counted wait assumes the steady stream behind D(k); the first iterations of the lagging group have run no B
    //  phase yet -- fewer operations are younger than D(k) than the count allows for -- so the first four drain instead)
    if (k < 4) fb_wait_vmcnt<0>(); else fb_wait_vmcnt<NDMA + 2 * L>();
    __syncthreads();
    const int p = k + 2 < 2 * NCH ? k + 2 : 2 * NCH - 2 + (k & 1);          // past the end: a last phase of the same kind, again
    dma((p & 1) ? w1img : w3img, p >> 1, (k + 2) & 3);
  };
  static_assert(NCH % 2 == 0, "chunks come in pairs (two named register sets)");
  // One instantiation of the loop per group (LAG a constant): with the group a run-time condition inside one loop the
  // compiler merged the two paths' hand-loaded registers with copies made while the loads were in flight
  // (tools/isa_inflight_check.py: 486 findings) and spilled.
  auto run = [&](auto lag_c) __attribute__((always_inline)) {
    constexpr int LAG = decltype(lag_c)::value;
    for (int k = 0; k < 2 * NCH; k += 4) {
      const int c0 = k >> 1;                       // chunks c0 (even: idA) and c0 + 1 (odd: idB)
      head(k);
      if constexpr (LAG == 0) phase_a(c0);
      else { if (c0 > 0) phase_b(c0 - 1, idB); }
      head(k + 1);
      if constexpr (LAG == 0) phase_b(c0, idA);
      else phase_a(c0);
      head(k + 2);
      if constexpr (LAG == 0) phase_a(c0 + 1);
      else phase_b(c0, idA);
      head(k + 3);
      if constexpr (LAG == 0) phase_b(c0 + 1, idB);
      else phase_a(c0 + 1);
    }
    if constexpr (NW == 8) {
      head(2 * NCH);
      if constexpr (LAG != 0) phase_b(NCH - 1, idB);
    }
    fb_wait_vmcnt<0>();               // the clamped DMAs and loads past the end: nothing may be in flight when the LDS is re-used / handed on
    CAPNET_LANDED4(idA[0][0], idA[0][1], idB[0][0], idB[0][1]);
    if constexpr (RS == 2) CAPNET_LANDED4(idA[1][0], idA[1][1], idB[1][0], idB[1][1]);
  };
  if (NW == 4 || wave_u < 4) run(std::integral_constant<int, 0>{});
  else run(std::integral_constant<int, 1>{});
  __syncthreads();

  // ---- epilogue: y1 = acc 2^-(ew1 + e1); column statistics of the rows below M
  const float osc = ldexpf(1.f, -((int)g.w1[0] + g.e1));
  float (*const scratch)[NW][MID] = reinterpret_cast<float (*)[NW][MID]>(ring);     // [sum | sumsq][wave][col]
  float bad = 0.f;
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    float cs = 0.f, cq = 0.f;
#pragma unroll
    for (int s = 0; s < RS; ++s)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = tile0 + (wave * RS + s) * 16 + 4 * gq + i;
        const float v = acc[s][nb][i] * osc;
        if (row < g.M) {
          g.y1[(long)row * MID + 16 * nb + ln] = v;
          cs += v;
          cq = fmaf(v, v, cq);
        }
      }
    if (g.part_sum) {
      cs += __shfl_xor(cs, 16); cq += __shfl_xor(cq, 16);
      cs += __shfl_xor(cs, 32); cq += __shfl_xor(cq, 32);
      if (gq == 0) { scratch[0][wave][16 * nb + ln] = cs; scratch[1][wave][16 * nb + ln] = cq; }
    } else {
      bad += cq;
    }
  }
  if (!g.part_sum) {
    if (g.err) flag_nonfinite(bad, g.err);
    return;
  }
  __syncthreads();
  for (int c = tid; c < MID; c += NT) {
    float a = 0.f, b = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) { a += scratch[0][w][c]; b += scratch[1][w][c]; }
    g.part_sum[(long)blockIdx.x * MID + c] = a;
    g.part_sq[(long)blockIdx.x * MID + c] = b;
  }
}

// The same launch on 32-ROW strips and v_mfma_f32_32x32x16_f16: four waves (one per SIMD, up to 512 registers) of 32 rows.
// A weight fragment (1 KB) feeds 32 rows, not 16 -- half the LDS fragment traffic per row -- and a 32-cycle MFMA leaves six
// issue slots in its shadow where a 16-cycle one leaves two. Built on the expectation that a lone wave would then hide its
// fragment reads and address arithmetic behind its own MFMAs; MEASURED SLOWER than eight waves of 16 rows (104 vs 79 us
// alone at stage 3, -6 % images/s in the pipelined step): a lone wave per SIMD does not overlap its own phases, whatever
// the MFMA shape. Opt-in (CAPNET_FB_WIDE=1), kept under test as the other A/B arms are. Phase A transposed as before: its D tile (lane
// (n = l & 31, h = l >> 5): channels (r & 3) + 8 (r >> 2) + 4 h of row n in register r) is phase B's A operand in place --
// registers 8 s .. 8 s + 7 are k step s, and the W1 image carries that channel order (fb_pack_kernel, wide).
template <int MID>
__global__ __launch_bounds__(256, 1) void fb_fused_wide_kernel(const FArgs g) {
  constexpr int C = 4 * MID, KS = MID / 16, NB = MID / 32, NCH = C / 32, NW = 4;
  constexpr int SLOT = MID * 128;
  constexpr int NDMA = SLOT / 1024 / NW;
  constexpr int L = 4;                                // identity loads = out stores per wave and chunk (32 rows x 32 channels)
  constexpr int TR = 32 * NW;
  constexpr int NT = 64 * NW;
  static_assert(NDMA >= 1 && 4 * NDMA + 2 * L <= 63, "vmcnt range");
  __shared__ __attribute__((aligned(16))) unsigned char ring[4 * SLOT];
  __shared__ __attribute__((aligned(16))) float par[4][C];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lh = lane >> 5, li = lane & 31;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const unsigned ring0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)ring);
  const int tile0 = (int)blockIdx.x * TR;
  const bool fold_res = g.sd != nullptr;
  const float* const w3img = reinterpret_cast<const float*>(g.w3 + kFHdr);
  const float* const w1img = reinterpret_cast<const float*>(g.w1 + kFHdr);
  auto dma = [&](const float* img, int cc, int slot) __attribute__((always_inline)) {
    const float* src = img + (long)cc * (SLOT / 4);
#pragma unroll
    for (int q = 0; q < ((CAPNET_FB_DBG == 2 || CAPNET_FB_DBG == 4) ? 0 : NDMA); ++q)
      glds16(src, (wave_u * NDMA + q) * 1024 + lane * 16, ring0 + (unsigned)(slot * SLOT + (wave_u * NDMA + q) * 1024));
  };
  // identity rows of chunk cc: lane (li, lh) takes channels 32 cc + 8 q + 4 lh .. + 3 of its row, q = 0 .. 3
  const int myrow = min(tile0 + wave * 32 + li, g.M - 1);          // rows past M repeat row M - 1 (see the 16-row kernel)
  const unsigned idoff = (unsigned)(((long)myrow * C + 4 * lh) * 4);
  f32x4 idA[4], idB[4];
  auto fetch_id = [&](int cc, f32x4 (&id)[4]) __attribute__((always_inline)) {
#pragma unroll
    for (int q = 0; q < 4; ++q) gload16(id[q], g.res, idoff + (unsigned)((32 * cc + 8 * q) * 4));
  };
  dma(w3img, 0, 0);
  dma(w1img, 0, 1);
  fetch_id(0, idA);
  fetch_id(NCH > 1 ? 1 : 0, idB);
  {
    const float x3 = ldexpf(1.f, -((int)g.w3[0] + g.e3));
    for (int i = tid; i < C; i += NT) {
      par[0][i] = g.s3[i] * x3;
      par[1][i] = g.t3[i];
      par[2][i] = fold_res ? g.sd[i] : 1.f;
      par[3][i] = fold_res ? g.td[i] : 0.f;
    }
  }
  // a2 of the wave's 32 rows as the B operand of phase A: lane (row li, half lh) holds k = 16 ks + 8 lh + j
  h8 ah[KS], al[KS];
  {
    const float is3 = ldexpf(1.f, g.e3);
    const float* p = g.y2 + (long)myrow * MID + 8 * lh;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      f32x4 v0 = *reinterpret_cast<const f32x4*>(p + 16 * ks), v1 = *reinterpret_cast<const f32x4*>(p + 16 * ks + 4);
      const f32x4 sa = *reinterpret_cast<const f32x4*>(g.s2 + 16 * ks + 8 * lh), sb = *reinterpret_cast<const f32x4*>(g.s2 + 16 * ks + 8 * lh + 4);
      const f32x4 ta = *reinterpret_cast<const f32x4*>(g.t2 + 16 * ks + 8 * lh), tb = *reinterpret_cast<const f32x4*>(g.t2 + 16 * ks + 8 * lh + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v0[e] = fmaxf(fmaf(v0[e], sa[e], ta[e]), 0.f) * is3;
        v1[e] = fmaxf(fmaf(v1[e], sb[e], tb[e]), 0.f) * is3;
      }
      h4 h0, l0, h1, l1;
      fb_split4(v0, h0, l0);
      fb_split4(v1, h1, l1);
      ah[ks] = fb_cat(h0, h1);
      al[ks] = fb_cat(l0, l1);
    }
  }
  fb_wait_vmcnt<0>();
  CAPNET_LANDED4(idA[0], idA[1], idA[2], idA[3]);
  CAPNET_LANDED4(idB[0], idB[1], idB[2], idB[3]);
  __syncthreads();

  f32x16 acc[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[nb][r] = 0.f;
  const float is1 = ldexpf(1.f, g.e1);
  typedef const __attribute__((address_space(3))) unsigned char* lds_bytes;
  typedef const __attribute__((address_space(3))) h8* lds_h8;
  const lds_bytes ring3 = (lds_bytes)ring;
  constexpr int GB = 4;                               // groups (two fragments each) per batch of fragment reads
  f32x16 d;
  auto phase_a = [&](int cc) __attribute__((always_inline)) {
#pragma unroll
    for (int r = 0; r < 16; ++r) d[r] = 0.f;
    unsigned wa_off = (unsigned)(((2 * cc) & 3) * SLOT + lane * 16);
    asm volatile("" : "+v"(wa_off));
    const lds_bytes wa = ring3 + wa_off;
    constexpr int NBAT = (KS + GB - 1) / GB;          // group = k16 step: fragments (ks * 2 + plane) KB into the slot
    h8 f[2][GB][2];
#pragma unroll
    for (int q = 0; q < GB; ++q)
      if (q < KS) { f[0][q][0] = *(lds_h8)(wa + (q * 2) * 1024); f[0][q][1] = *(lds_h8)(wa + (q * 2 + 1) * 1024); }
#pragma unroll
    for (int b = 0; b < NBAT; ++b) {
      if (b + 1 < NBAT) {
#pragma unroll
        for (int q = 0; q < GB; ++q) {
          const int ks = (b + 1) * GB + q;
          if (ks < KS) { f[(b + 1) & 1][q][0] = *(lds_h8)(wa + (ks * 2) * 1024); f[(b + 1) & 1][q][1] = *(lds_h8)(wa + (ks * 2 + 1) * 1024); }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int q = 0; q < GB; ++q) {
        const int ks = b * GB + q;
        if (ks < KS) {
          d = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[b & 1][q][1], ah[ks], d, 0, 0, 0);
          d = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[b & 1][q][0], al[ks], d, 0, 0, 0);
          d = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[b & 1][q][0], ah[ks], d, 0, 0, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  auto phase_b = [&](int cc, f32x4 (&id)[4]) __attribute__((always_inline)) {
    unsigned wb_off = (unsigned)(((2 * cc + 1) & 3) * SLOT + lane * 16);
    asm volatile("" : "+v"(wb_off));
    const lds_bytes wb = ring3 + wb_off;
    constexpr int NG = NB * 2, NBATB = (NG + GB - 1) / GB;      // group = (nb, step): fragments (group * 2 + plane) KB
    h8 fb[2][GB][2];
#pragma unroll
    for (int q = 0; q < GB; ++q)
      if (q < NG) { fb[0][q][0] = *(lds_h8)(wb + (q * 2) * 1024); fb[0][q][1] = *(lds_h8)(wb + (q * 2 + 1) * 1024); }
    __builtin_amdgcn_sched_barrier(0);
    fb_wait_vmcnt<4 * NDMA + 2 * L>();
    CAPNET_LANDED4(id[0], id[1], id[2], id[3]);
    h4 hh[4], ll[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int ch = 32 * cc + 8 * q + 4 * lh;
      const f32x4 sv = *reinterpret_cast<const f32x4*>(&par[0][ch]), tv = *reinterpret_cast<const f32x4*>(&par[1][ch]);
      f32x4 r = id[q], o;
      if (fold_res) {
        const f32x4 dv = *reinterpret_cast<const f32x4*>(&par[2][ch]), ev = *reinterpret_cast<const f32x4*>(&par[3][ch]);
#pragma unroll
        for (int e = 0; e < 4; ++e) r[e] = fmaf(r[e], dv[e], ev[e]);
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = fmaxf(fmaf(d[4 * q + e], sv[e], tv[e]) + r[e], 0.f);
      *reinterpret_cast<f32x4*>(g.out + (long)myrow * C + ch) = o;
      fb_split4(o * is1, hh[q], ll[q]);
    }
    const h8 oh[2] = {fb_cat(hh[0], hh[1]), fb_cat(hh[2], hh[3])}, ol[2] = {fb_cat(ll[0], ll[1]), fb_cat(ll[2], ll[3])};
    fetch_id(cc + 2 < NCH ? cc + 2 : NCH - 1, id);
#pragma unroll
    for (int b = 0; b < NBATB; ++b) {
      if (b + 1 < NBATB) {
#pragma unroll
        for (int q = 0; q < GB; ++q) {
          const int grp = (b + 1) * GB + q;
          if (grp < NG) { fb[(b + 1) & 1][q][0] = *(lds_h8)(wb + (grp * 2) * 1024); fb[(b + 1) & 1][q][1] = *(lds_h8)(wb + (grp * 2 + 1) * 1024); }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int q = 0; q < GB; ++q) {
        const int grp = b * GB + q, nb = grp >> 1, st = grp & 1;
        if (grp < NG) {
          acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ol[st], fb[b & 1][q][0], acc[nb], 0, 0, 0);
          acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(oh[st], fb[b & 1][q][1], acc[nb], 0, 0, 0);
          acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(oh[st], fb[b & 1][q][0], acc[nb], 0, 0, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  auto head = [&](int k) __attribute__((always_inline)) {
    if (k < 4) fb_wait_vmcnt<0>(); else fb_wait_vmcnt<NDMA + 2 * L>();
    __syncthreads();
    const int p = k + 2 < 2 * NCH ? k + 2 : 2 * NCH - 2 + (k & 1);
    dma((p & 1) ? w1img : w3img, p >> 1, (k + 2) & 3);
  };
  static_assert(NCH % 2 == 0, "chunks come in pairs (two named register sets)");
  for (int k = 0; k < 2 * NCH; k += 4) {
    const int c0 = k >> 1;
    head(k);
    phase_a(c0);
    head(k + 1);
    phase_b(c0, idA);
    head(k + 2);
    phase_a(c0 + 1);
    head(k + 3);
    phase_b(c0 + 1, idB);
  }
  fb_wait_vmcnt<0>();
  CAPNET_LANDED4(idA[0], idA[1], idA[2], idA[3]);
  CAPNET_LANDED4(idB[0], idB[1], idB[2], idB[3]);
  __syncthreads();

  // ---- epilogue: y1 = acc 2^-(ew1 + e1); D[r]: row (r & 3) + 8 (r >> 2) + 4 lh, column 32 nb + li
  const float osc = ldexpf(1.f, -((int)g.w1[0] + g.e1));
  float (*const scratch)[NW][MID] = reinterpret_cast<float (*)[NW][MID]>(ring);
  float bad = 0.f;
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    float cs = 0.f, cq = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = tile0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      const float v = acc[nb][r] * osc;
      if (row < g.M) {
        g.y1[(long)row * MID + 32 * nb + li] = v;
        cs += v;
        cq = fmaf(v, v, cq);
      }
    }
    if (g.part_sum) {
      cs += __shfl_xor(cs, 32); cq += __shfl_xor(cq, 32);
      if (lh == 0) { scratch[0][wave][32 * nb + li] = cs; scratch[1][wave][32 * nb + li] = cq; }
    } else {
      bad += cq;
    }
  }
  if (!g.part_sum) {
    if (g.err) flag_nonfinite(bad, g.err);
    return;
  }
  __syncthreads();
  for (int c = tid; c < MID; c += NT) {
    float a = 0.f, b = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) { a += scratch[0][w][c]; b += scratch[1][w][c]; }
    g.part_sum[(long)blockIdx.x * MID + c] = a;
    g.part_sq[(long)blockIdx.x * MID + c] = b;
  }
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------------
bool fused_block_shape_ok(long M, int MID) {
  return (MID == 64 || MID == 128 || MID == 256) && M > 0 && M < (1l << 24) && M * 4 * MID * 4 < (1l << 32);
}

size_t fused_block_weight_words(int C, int MID, int role) {
  return (size_t)kFHdr + (size_t)C * MID * (role == 0 ? 2 : 1);
}

// 32-row strips on 32x32x16 MFMAs (fb_fused_wide_kernel) or 16-row strips on 16x16x32 ones: the weight images differ, so
// the choice is a function of MID alone (and of CAPNET_FB_WIDE = 0 / 1, read when the images are packed AND at launch)
static bool fb_wide(int MID) {
  const char* e = getenv("CAPNET_FB_WIDE");
  if (e && (e[0] == '0' || e[0] == '1')) return e[0] == '1';
  (void)MID;
  return false;      // measured: 104 us against 79 us alone on the 14 x 14 maps, -6 % images/s in the step: opt-in
}

// role 0: w = conv3's weights [C][MID]; role 1: w = the next conv1's weights [MID][C]
int fused_block_pack(const float* w, unsigned* img, int C, int MID, int role, hipStream_t stream) {
  CAPNET_REQUIRE(w && img && aligned16(img) && C == 4 * MID && (MID == 64 || MID == 128 || MID == 256) && (role == 0 || role == 1),
                 "fused_block_pack: bad argument (C=%d MID=%d role=%d)", C, MID, role);
  CAPNET_HIP_CHECK(hipMemsetAsync(img, 0, kFHdr * 4, stream));
  const long n = (long)C * MID;
  hipLaunchKernelGGL(fb_absmax_kernel, dim3((int)(cdiv(n, 256 * 8) > 1024 ? 1024 : cdiv(n, 256 * 8))), dim3(256), 0, stream, w, img, n);
  CAPNET_LAUNCH_CHECK();
  hipLaunchKernelGGL(fb_pack_kernel, dim3((int)(cdiv(n / 4, 256) > 2048 ? 2048 : cdiv(n / 4, 256))), dim3(256), 0, stream, w, img, C, MID, role, fb_wide(MID) ? 1 : 0);
  CAPNET_LAUNCH_CHECK();
  return kOk;
}

int fused_block_gram_slices(long M, int K) { return cdiv(M, fb_gram_rows(K)); }
int fused_block_gram_pairs(int K) { const int nb = K / 64; return nb * (nb + 1) / 2; }
// floats of the statistics workspace: partial blocks + column sums, then G and mu (fp32), 16-B aligned parts
size_t fused_block_stats_floats(long M, int K) {
  const size_t s = fused_block_gram_slices(M, K), p = fused_block_gram_pairs(K);
  return s * p * 4096 + ((s * K + 3) / 4 * 4) + ((size_t)K * K + K) + 8;
}

// (scale, shift) of the BatchNorm behind y3 = relu(y2 s2 + t2) . W3^T (conv3 of a bottleneck), y3 never formed.
// w3img: fused_block_pack(role 0) image (its fp32 copy is what the quadratic forms read).
int fused_block_stats(const float* y2, const float* s2, const float* t2, const unsigned* w3img, long M, int MID, int in_exp,
                      const float* gamma, const float* beta, float* running_mean, float* running_var, float momentum, float eps,
                      float* scale, float* shift, float* batch_mean, float* batch_var, float* work, int* err, hipStream_t stream) {
  CAPNET_REQUIRE(y2 && s2 && t2 && w3img && scale && shift && work && aligned16(work) && aligned16(y2) && fused_block_shape_ok(M, MID) &&
                     in_exp > -64 && in_exp < 64, "fused_block_stats: bad argument");
  const int K = MID, C = 4 * MID;
  GArgs a;
  a.y2 = y2; a.s2 = s2; a.t2 = t2; a.M = (int)M; a.K = K; a.in_exp = in_exp;
  a.slices = fused_block_gram_slices(M, K); a.pairs = fused_block_gram_pairs(K); a.slice_rows = fb_gram_rows(K);
  a.gp = work;
  a.cs = work + (size_t)a.slices * a.pairs * 4096;
  float* G = a.cs + ((size_t)a.slices * K + 3) / 4 * 4;
  float* mu = G + (size_t)K * K;
  if (K == 256) hipLaunchKernelGGL(fb_gram_kernel<256>, dim3(a.slices), dim3(256), 0, stream, a);
  else if (K == 128) hipLaunchKernelGGL(fb_gram_kernel<128>, dim3(a.slices), dim3(256), 0, stream, a);
  else hipLaunchKernelGGL(fb_gram_kernel<64>, dim3(a.slices), dim3(256), 0, stream, a);
  hipLaunchKernelGGL(fb_gram_reduce_kernel, dim3(cdiv((long)K * K + K, 32)), dim3(256), 0, stream, a.gp, a.cs, G, mu, K, a.slices, a.pairs);
  const double inv = 1.0 / (double)M, unbias = M > 1 ? (double)M / (double)(M - 1) : 1.0;
  const float* wcopy = reinterpret_cast<const float*>(w3img + kFHdr + (size_t)C * MID);
  if (K == 256)
    hipLaunchKernelGGL(fb_quad_kernel<256>, dim3(cdiv(C, kQC)), dim3(K * 4), 0, stream, G, mu, wcopy, C, inv, unbias, gamma, beta,
                       running_mean, running_var, momentum, eps, scale, shift, batch_mean, batch_var, err);
  else if (K == 128)
    hipLaunchKernelGGL(fb_quad_kernel<128>, dim3(cdiv(C, kQC)), dim3(K * 4), 0, stream, G, mu, wcopy, C, inv, unbias, gamma, beta,
                       running_mean, running_var, momentum, eps, scale, shift, batch_mean, batch_var, err);
  else
    hipLaunchKernelGGL(fb_quad_kernel<64>, dim3(cdiv(C, kQC)), dim3(K * 4), 0, stream, G, mu, wcopy, C, inv, unbias, gamma, beta,
                       running_mean, running_var, momentum, eps, scale, shift, batch_mean, batch_var, err);
  CAPNET_LAUNCH_CHECK();
  return kOk;
}

// Tile shape: CAPNET_FB_RS / CAPNET_FB_NW (A/B switches) or the default -- 8 waves of one strip (128 rows, two waves per SIMD).
static void fb_shape(int MID, int* rs, int* nw) {
  (void)MID;
  const char* e = getenv("CAPNET_FB_RS");
  const char* w = getenv("CAPNET_FB_NW");
  *nw = (w && w[0] == '4') ? 4 : 8;
  *rs = (e && (e[0] == '1' || e[0] == '2')) ? e[0] - '0' : (*nw == 8 ? 1 : 2);
  if (*nw == 8) *rs = 1;
}
int fused_block_tiles(long M, int MID) {
  if (fb_wide(MID)) return cdiv(M, 128);
  int rs, nw;
  fb_shape(MID, &rs, &nw);
  return cdiv(M, 16 * rs * nw);
}

// out [M][4 MID] = relu(bn3(relu(y2 s2 + t2) . W3^T) + res (sd + td)) and y1 [M][MID] = out . W1^T, statistics partials
// [fused_block_tiles][MID]; part_sum / part_sq null: none (inference), outputs checked for non-finite values instead.
int fused_block_forward(const float* y2, const float* s2, const float* t2, const unsigned* w3img, const float* s3, const float* t3,
                        const float* res, const float* sd, const float* td, float* out, const unsigned* w1img, float* y1,
                        float* part_sum, float* part_sq, long M, int MID, int e3, int e1, int* err, hipStream_t stream) {
  CAPNET_REQUIRE(y2 && s2 && t2 && w3img && s3 && t3 && res && out && w1img && y1 && fused_block_shape_ok(M, MID),
                 "fused_block_forward: bad argument");
  CAPNET_REQUIRE(aligned16(y2) && aligned16(s2) && aligned16(t2) && aligned16(w3img) && aligned16(w1img) && aligned16(res) &&
                     aligned16(out) && (sd == nullptr) == (td == nullptr) && (part_sum == nullptr) == (part_sq == nullptr) &&
                     e3 > -64 && e3 < 64 && e1 > -64 && e1 < 64, "fused_block_forward: alignment / pairs / exponents");
  FArgs a;
  a.y2 = y2; a.s2 = s2; a.t2 = t2; a.w3 = w3img; a.w1 = w1img; a.s3 = s3; a.t3 = t3; a.res = res; a.sd = sd; a.td = td;
  a.out = out; a.y1 = y1; a.part_sum = part_sum; a.part_sq = part_sq; a.M = (int)M; a.e3 = e3; a.e1 = e1; a.err = err;
  const dim3 grid(fused_block_tiles(M, MID));
  int rs, nw;
  fb_shape(MID, &rs, &nw);
  if (fb_wide(MID)) {
    const dim3 block(256);
    if (MID == 256) CAPNET_LAUNCH_TIMED((fb_fused_wide_kernel<256>), grid, block, stream, a);
    else if (MID == 128) CAPNET_LAUNCH_TIMED((fb_fused_wide_kernel<128>), grid, block, stream, a);
    else CAPNET_LAUNCH_TIMED((fb_fused_wide_kernel<64>), grid, block, stream, a);
    CAPNET_LAUNCH_CHECK();
    return kOk;
  }
  const dim3 block(64 * nw);
  if (nw == 8) {
    if (MID == 256) CAPNET_LAUNCH_TIMED((fb_fused_kernel<256, 1, 8>), grid, block, stream, a);
    else if (MID == 128) CAPNET_LAUNCH_TIMED((fb_fused_kernel<128, 1, 8>), grid, block, stream, a);
    else CAPNET_LAUNCH_TIMED((fb_fused_kernel<64, 1, 8>), grid, block, stream, a);
  } else if (rs == 2) {
    if (MID == 256) CAPNET_LAUNCH_TIMED((fb_fused_kernel<256, 2, 4>), grid, block, stream, a);
    else if (MID == 128) CAPNET_LAUNCH_TIMED((fb_fused_kernel<128, 2, 4>), grid, block, stream, a);
    else CAPNET_LAUNCH_TIMED((fb_fused_kernel<64, 2, 4>), grid, block, stream, a);
  } else {
    if (MID == 256) CAPNET_LAUNCH_TIMED((fb_fused_kernel<256, 1, 4>), grid, block, stream, a);
    else if (MID == 128) CAPNET_LAUNCH_TIMED((fb_fused_kernel<128, 1, 4>), grid, block, stream, a);
    else CAPNET_LAUNCH_TIMED((fb_fused_kernel<64, 1, 4>), grid, block, stream, a);
  }
  CAPNET_LAUNCH_CHECK();
  return kOk;
}

}  // namespace capnet
