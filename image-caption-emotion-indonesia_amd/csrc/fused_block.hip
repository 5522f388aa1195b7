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
//         = W3[32 chunk + 16 blk + (lane & 15)][32 ks + kslot(lane >> 4, j)], followed by a verbatim fp32 copy of W3 (the
//         statistics' quadratic forms read the weights as they are);
// role 1: W1 [MID][C] (the next conv1, the B operand of phase B): cells [chunk C/32][nb MID/16][plane 2][lane 64]
//         = W1[16 nb + (lane & 15)][32 chunk + kslot(lane >> 4, j)].
__global__ __launch_bounds__(256) void fb_pack_kernel(const float* __restrict__ w, unsigned* __restrict__ img, int C, int MID,
                                                      int role) {
  const int ew = fb_weight_shift(img[1]);
  if (blockIdx.x == 0 && threadIdx.x == 0) img[0] = (unsigned)ew;
  const float ws = ldexpf(1.f, ew);
  const long cells = (long)C * MID / 8 * 2;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < cells; i += (long)gridDim.x * blockDim.x) {
    long r = i;
    const int lane = (int)(r & 63); r >>= 6;
    const int plane = (int)(r & 1); r >>= 1;
    const int g = lane >> 4, n = lane & 15;
    const float* src;
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
    unsigned out[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float x0 = src[fb_kslot(g, 2 * q)] * ws, x1 = src[fb_kslot(g, 2 * q + 1)] * ws;
      const _Float16 h0 = (_Float16)x0, h1 = (_Float16)x1;
      const _Float16 l0 = (_Float16)(x0 - (float)h0), l1 = (_Float16)(x1 - (float)h1);
      const h2 p = plane == 0 ? h2{h0, h1} : h2{l0, l1};
      out[q] = __builtin_bit_cast(unsigned, p);
    }
    unsigned* dst = img + kFHdr + i * 4;
    dst[0] = out[0]; dst[1] = out[1]; dst[2] = out[2]; dst[3] = out[3];
  }
  if (role == 0) {
    float* cp = reinterpret_cast<float*>(img + kFHdr + (long)C * MID);
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < (long)C * MID; i += (long)gridDim.x * blockDim.x) cp[i] = w[i];
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// statistics of y3 from the second moments of conv3's input
// ---------------------------------------------------------------------------------------------------------------------
constexpr int kGR = 512;                  // rows of a slice
constexpr int kGStage = 64;               // rows staged per step (8 row groups of 8)

struct GArgs {
  const float* y2; const float* s2; const float* t2;
  float* gp;            // [slices][pairs][64][64]
  float* cs;            // [slices][K]
  int M, K, in_exp, slices, pairs;
};

// workgroup (slice, pair (bi <= bj) of 64-channel blocks): Gp = a2[rows, bi]^T a2[rows, bj] over the slice's rows; the
// diagonal pairs also write the column sums of their block. a2 = relu(y2 s2 + t2) 2^e, split into f16 planes while it is
// staged ([plane][8-row group][channel][8 halfs]: a cell is 8 ROWS of one channel = the k slots of both operands of
// v_mfma_f32_32x32x16_f16 with k = row); three products, fp32 accumulate. 2 x 2 waves of 32 x 32.
__global__ __launch_bounds__(256) void fb_gram_kernel(const GArgs g) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * 2 * 8 * 128 * 16];      // two stages of 32 KB
  __shared__ float cs_sh[256];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int slice = blockIdx.x / g.pairs, pair = blockIdx.x - slice * g.pairs;
  int bi = 0, bj = 0;
  {
    int p = pair, nb = g.K / 64;
    while (p >= nb - bi) { p -= nb - bi; ++bi; }
    bj = bi + p;
  }
  const bool diag = bi == bj;
  const int nch = diag ? 64 : 128;
  const int wm = wave >> 1, wn = wave & 1, li = lane & 31, lh = lane >> 5;
  const int row0 = slice * kGR;
  const int rows = min(kGR, g.M - row0);
  const int nst = (rows + kGStage - 1) / kGStage;
  const float iscale = ldexpf(1.f, g.in_exp);
  // this thread's channel (fixed) and first row group of a stage
  const int chl = tid % nch, rg0 = tid / nch, rgs = 256 / nch;       // rgs = 4 (diag) / 2
  const int chg = (chl < 64 ? bi * 64 + chl : bj * 64 + chl - 64);
  const float sc = g.s2[chg], sh = g.t2[chg];
  float colsum = 0.f;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  constexpr int kPlane = 8 * 128 * 16, kStageB = 2 * kPlane;
  float v[4][8];
  auto fetch = [&](int st) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int rg = rg0 + q * rgs;
      if (rg < 8) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int m = row0 + st * kGStage + rg * 8 + e;
          v[q][e] = g.y2[(long)min(m, g.M - 1) * g.K + chg];
        }
      }
    }
  };
  auto stage = [&](int st, int buf) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int rg = rg0 + q * rgs;
      if (rg < 8) {
        f32x4 a, b;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int m = row0 + st * kGStage + rg * 8 + e;
          float x = fmaxf(fmaf(v[q][e], sc, sh), 0.f) * iscale;
          x = (m < row0 + rows) ? x : 0.f;
          colsum += x;
          if (e < 4) a[e] = x; else b[e - 4] = x;
        }
        h4 ha, la, hb, lb;
        fb_split4(a, ha, la);
        fb_split4(b, hb, lb);
        unsigned char* d = lds + buf * kStageB + (rg * 128 + chl) * 16;
        *reinterpret_cast<h8*>(d) = fb_cat(ha, hb);
        *reinterpret_cast<h8*>(d + kPlane) = fb_cat(la, lb);
      }
    }
  };
  fetch(0);
  stage(0, 0);
  __syncthreads();
  for (int st = 0; st < nst; ++st) {
    const int buf = st & 1;
    if (st + 1 < nst) fetch(st + 1);
    const unsigned char* ia = lds + buf * kStageB + (wm * 32 + li) * 16;
    const unsigned char* ib = lds + buf * kStageB + ((diag ? 0 : 64) + wn * 32 + li) * 16;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int off = (2 * k + lh) * 128 * 16;
      const h8 ah = *reinterpret_cast<const h8*>(ia + off), al = *reinterpret_cast<const h8*>(ia + off + kPlane);
      const h8 bh = *reinterpret_cast<const h8*>(ib + off), bl = *reinterpret_cast<const h8*>(ib + off + kPlane);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc, 0, 0, 0);
    }
    if (st + 1 < nst) stage(st + 1, buf ^ 1);
    __syncthreads();
  }
  // D[m][n]: m = (r & 3) + 8 (r >> 2) + 4 lh = channel of block bi, n = li = channel of block bj; both operands carried 2^e
  const float osc = ldexpf(1.f, -2 * g.in_exp);
  float* out = g.gp + ((long)slice * g.pairs + pair) * 4096;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int m = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
    out[m * 64 + wn * 32 + li] = acc[r] * osc;
  }
  if (diag) {
    cs_sh[tid] = colsum;
    __syncthreads();
    if (tid < 64) g.cs[(long)slice * g.K + bi * 64 + tid] = ((cs_sh[tid] + cs_sh[tid + 64]) + (cs_sh[tid + 128] + cs_sh[tid + 192])) * ldexpf(1.f, -g.in_exp);
  }
}

// G (full, symmetric) and the column sums in double: 8 threads share an entry's slices
__global__ __launch_bounds__(256) void fb_gram_reduce_kernel(const float* __restrict__ gp, const float* __restrict__ cs, double* __restrict__ G,
                                                             double* __restrict__ mu, int K, int slices, int pairs) {
  __shared__ double sh[8][33];
  const int e = blockIdx.x * 32 + (threadIdx.x & 31), part = threadIdx.x >> 5;
  const int nG = K * K;
  double s = 0.0;
  if (e < nG) {
    int r = e / K, c = e - r * K;
    int bi = r >> 6, bj = c >> 6, m = r & 63, n = c & 63;
    if (bi > bj) { int t = bi; bi = bj; bj = t; t = m; m = n; n = t; }
    const int nb = K / 64;
    const int pair = bi * nb - bi * (bi - 1) / 2 + (bj - bi);
    const float* p = gp + (long)pair * 4096 + m * 64 + n;
    for (int sl = part; sl < slices; sl += 8) s += (double)p[(long)sl * pairs * 4096];
  } else if (e < nG + K) {
    const int k = e - nG;
    for (int sl = part; sl < slices; sl += 8) s += (double)cs[(long)sl * K + k];
  }
  sh[part][threadIdx.x & 31] = s;
  __syncthreads();
  if (part == 0) {
#pragma unroll
    for (int q = 1; q < 8; ++q) s += sh[q][threadIdx.x & 31];
    if (e < nG) G[e] = s;
    else if (e < nG + K) mu[e - nG] = s;
  }
}

constexpr int kQC = 8;                   // channels per workgroup of the quadratic forms

// channel c: sum = mu . w_c, sumsq = w_c^T G w_c (double), then bn_finalize_kernel's arithmetic (bn_pool.hip)
__global__ __launch_bounds__(256) void fb_quad_kernel(const double* __restrict__ G, const double* __restrict__ mu,
                                                      const float* __restrict__ w /* [C][K] */, int K, int C, double inv_count,
                                                      double unbias, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                      float* __restrict__ running_mean, float* __restrict__ running_var,
                                                      float momentum, float eps, float* __restrict__ scale, float* __restrict__ shift,
                                                      float* __restrict__ batch_mean, float* __restrict__ batch_var, int* __restrict__ err) {
  __shared__ float wsh[kQC][256];
  __shared__ double red[2][kQC][4];
  const int tid = threadIdx.x, c0 = blockIdx.x * kQC;
  for (int i = tid; i < kQC * K; i += 256) wsh[i / K][i % K] = (c0 + i / K < C) ? w[(long)c0 * K + i] : 0.f;
  __syncthreads();
  const int col = tid % K, rpart = tid / K, nparts = 256 / K;
  double t[kQC];
#pragma unroll
  for (int c = 0; c < kQC; ++c) t[c] = 0.0;
  for (int k = rpart; k < K; k += nparts) {
    const double gv = G[(long)k * K + col];
#pragma unroll
    for (int c = 0; c < kQC; ++c) t[c] = fma(gv, (double)wsh[c][k], t[c]);
  }
  double q[kQC], s[kQC];
  const double m = rpart == 0 ? mu[col] : 0.0;
#pragma unroll
  for (int c = 0; c < kQC; ++c) {
    q[c] = t[c] * (double)wsh[c][col];
    s[c] = m * (double)wsh[c][col];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      q[c] += __shfl_xor(q[c], o);
      s[c] += __shfl_xor(s[c], o);
    }
    if ((tid & 63) == 0) { red[0][c][tid >> 6] = q[c]; red[1][c][tid >> 6] = s[c]; }
  }
  __syncthreads();
  if (tid < kQC && c0 + tid < C) {
    const int c = c0 + tid;
    const double qq = (red[0][tid][0] + red[0][tid][1]) + (red[0][tid][2] + red[0][tid][3]);
    const double ss = (red[1][tid][0] + red[1][tid][1]) + (red[1][tid][2] + red[1][tid][3]);
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

template <int N>
__device__ __forceinline__ void fb_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int MID, int RS>
__global__ __launch_bounds__(256, 1) void fb_fused_kernel(const FArgs g) {
  constexpr int C = 4 * MID, KS = MID / 32, NB = MID / 16, NCH = C / 32;
  constexpr int SLOT = MID * 128;                     // bytes of one chunk image (either role)
  constexpr int NDMA = SLOT / 1024 / 4;               // 1-KB LDS-DMA instructions per wave and phase
  constexpr int L = 2 * RS;                           // identity loads = out stores per wave and chunk
  constexpr int TR = 64 * RS;                         // rows of a tile
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
  auto dma = [&](const float* img, int cc, int slot) {
    const float* src = img + (long)cc * (SLOT / 4);
#pragma unroll
    for (int q = 0; q < NDMA; ++q)
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
  auto fetch_id = [&](int cc, f32x4 (&id)[RS][2]) {
#pragma unroll
    for (int s = 0; s < RS; ++s)
#pragma unroll
      for (int blk = 0; blk < 2; ++blk) gload16(id[s][blk], g.res, idoff[s] + (unsigned)((32 * cc + 16 * blk) * 4));
  };

  // ---- prologue: the first three phases' weights, the first two chunks' identity rows, the parameters, a2
  dma(w3img, 0, 0);
  dma(w1img, 0, 1);
  dma(w3img, NCH > 1 ? 1 : 0, 2);
  fetch_id(0, idA);
  fetch_id(NCH > 1 ? 1 : 0, idB);
  {
    const float x3 = ldexpf(1.f, -((int)g.w3[0] + g.e3));
    for (int i = tid; i < C; i += 256) {
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

  // one chunk; `id` holds this chunk's identity rows and receives chunk cc + 2's.
  // vector-memory operations of a chunk in issue order: D1(cc + 1) x NDMA, ST x L, LD(cc + 2) x L, D3(cc + 2) x NDMA
  auto chunk = [&](int cc, f32x4 (&id)[RS][2]) {
    const int sA = (2 * cc) & 3, sB = (2 * cc + 1) & 3;
    // ---- phase A: needs D3(cc), the last group of chunk cc - 2; younger than it: all of chunk cc - 1
    fb_wait_vmcnt<2 * NDMA + 2 * L>();
    __syncthreads();                  // every wave's share of W3[cc] is in LDS; every wave is through with phase B(cc - 1)
    dma(w1img, cc + 1 < NCH ? cc + 1 : NCH - 1, (2 * cc + 3) & 3);
    f32x4 d[RS][2];
#pragma unroll
    for (int s = 0; s < RS; ++s) { d[s][0] = f32x4{0.f, 0.f, 0.f, 0.f}; d[s][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    const unsigned char* wa = ring + sA * SLOT + lane * 16;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int blk = 0; blk < 2; ++blk) {
        const h8 wh = *reinterpret_cast<const h8*>(wa + ((ks * 2 + blk) * 2 + 0) * 1024);
        const h8 wl = *reinterpret_cast<const h8*>(wa + ((ks * 2 + blk) * 2 + 1) * 1024);
#pragma unroll
        for (int s = 0; s < RS; ++s) {
          d[s][blk] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl, ah[s][ks], d[s][blk], 0, 0, 0);
          d[s][blk] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, al[s][ks], d[s][blk], 0, 0, 0);
          d[s][blk] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, ah[s][ks], d[s][blk], 0, 0, 0);
        }
      }
    // ---- the tail: needs LD(cc) (third group of chunk cc - 2); younger: D3(cc), chunk cc - 1, D1(cc + 1)
    fb_wait_vmcnt<4 * NDMA + 2 * L>();
    if constexpr (RS == 2) CAPNET_LANDED4(id[0][0], id[0][1], id[1][0], id[1][1]);
    else CAPNET_LANDED2(id[0][0], id[0][1]);
    h8 oh[RS], ol[RS];
#pragma unroll
    for (int s = 0; s < RS; ++s) {
      // rows past M repeat row M - 1 bit for bit (a2 and the identity were fetched from it): their stores go to that row
      // as well -- unconditionally, the counted waits rely on every wave issuing exactly L stores per chunk
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
    // ---- phase B: needs D1(cc) (first group of chunk cc - 1); younger: ST, LD, D3 of chunk cc - 1, D1, ST, LD of this chunk
    fb_wait_vmcnt<2 * NDMA + 4 * L>();
    __syncthreads();                  // W1[cc] is in LDS; every wave is through with phase A(cc)
    dma(w3img, cc + 2 < NCH ? cc + 2 : NCH - 1, (2 * cc + 4) & 3);
    const unsigned char* wb = ring + sB * SLOT + lane * 16;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      const h8 wh = *reinterpret_cast<const h8*>(wb + (nb * 2 + 0) * 1024);
      const h8 wl = *reinterpret_cast<const h8*>(wb + (nb * 2 + 1) * 1024);
#pragma unroll
      for (int s = 0; s < RS; ++s) {
        acc[s][nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ol[s], wh, acc[s][nb], 0, 0, 0);
        acc[s][nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(oh[s], wl, acc[s][nb], 0, 0, 0);
        acc[s][nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(oh[s], wh, acc[s][nb], 0, 0, 0);
      }
    }
  };
  static_assert(NCH % 2 == 0, "chunks come in pairs (two named register sets)");
  for (int cc = 0; cc < NCH; cc += 2) {
    chunk(cc, idA);
    chunk(cc + 1, idB);
  }
  fb_wait_vmcnt<0>();                 // the clamped DMAs and loads past the end: nothing may be in flight when the LDS is re-used / handed on
  CAPNET_LANDED4(idA[0][0], idA[0][1], idB[0][0], idB[0][1]);
  if constexpr (RS == 2) CAPNET_LANDED4(idA[1][0], idA[1][1], idB[1][0], idB[1][1]);
  __syncthreads();

  // ---- epilogue: y1 = acc 2^-(ew1 + e1); column statistics of the rows below M
  const float osc = ldexpf(1.f, -((int)g.w1[0] + g.e1));
  float (*const scratch)[4][MID] = reinterpret_cast<float (*)[4][MID]>(ring);       // [sum | sumsq][wave][col]
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
  for (int c = tid; c < MID; c += 256) {
    g.part_sum[(long)blockIdx.x * MID + c] = (scratch[0][0][c] + scratch[0][1][c]) + (scratch[0][2][c] + scratch[0][3][c]);
    g.part_sq[(long)blockIdx.x * MID + c] = (scratch[1][0][c] + scratch[1][1][c]) + (scratch[1][2][c] + scratch[1][3][c]);
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

// role 0: w = conv3's weights [C][MID]; role 1: w = the next conv1's weights [MID][C]
int fused_block_pack(const float* w, unsigned* img, int C, int MID, int role, hipStream_t stream) {
  CAPNET_REQUIRE(w && img && aligned16(img) && C == 4 * MID && (MID == 64 || MID == 128 || MID == 256) && (role == 0 || role == 1),
                 "fused_block_pack: bad argument (C=%d MID=%d role=%d)", C, MID, role);
  CAPNET_HIP_CHECK(hipMemsetAsync(img, 0, kFHdr * 4, stream));
  const long n = (long)C * MID;
  hipLaunchKernelGGL(fb_absmax_kernel, dim3((int)(cdiv(n, 256 * 8) > 1024 ? 1024 : cdiv(n, 256 * 8))), dim3(256), 0, stream, w, img, n);
  CAPNET_LAUNCH_CHECK();
  hipLaunchKernelGGL(fb_pack_kernel, dim3((int)(cdiv(n / 4, 256) > 2048 ? 2048 : cdiv(n / 4, 256))), dim3(256), 0, stream, w, img, C, MID, role);
  CAPNET_LAUNCH_CHECK();
  return kOk;
}

int fused_block_gram_slices(long M) { return cdiv(M, kGR); }
int fused_block_gram_pairs(int K) { const int nb = K / 64; return nb * (nb + 1) / 2; }
// floats of the statistics workspace: partial blocks + column sums (fp32), then G and mu (double), 16-B aligned parts
size_t fused_block_stats_floats(long M, int K) {
  const size_t s = fused_block_gram_slices(M), p = fused_block_gram_pairs(K);
  return s * p * 4096 + ((s * K + 3) / 4 * 4) + 2 * ((size_t)K * K + K) + 8;
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
  a.slices = fused_block_gram_slices(M); a.pairs = fused_block_gram_pairs(K);
  a.gp = work;
  a.cs = work + (size_t)a.slices * a.pairs * 4096;
  double* G = reinterpret_cast<double*>(a.cs + ((size_t)a.slices * K + 3) / 4 * 4);
  double* mu = G + (size_t)K * K;
  hipLaunchKernelGGL(fb_gram_kernel, dim3(a.slices * a.pairs), dim3(256), 0, stream, a);
  hipLaunchKernelGGL(fb_gram_reduce_kernel, dim3(cdiv((long)K * K + K, 32)), dim3(256), 0, stream, a.gp, a.cs, G, mu, K, a.slices, a.pairs);
  const double inv = 1.0 / (double)M, unbias = M > 1 ? (double)M / (double)(M - 1) : 1.0;
  const float* wcopy = reinterpret_cast<const float*>(w3img + kFHdr + (size_t)C * MID);
  hipLaunchKernelGGL(fb_quad_kernel, dim3(cdiv(C, kQC)), dim3(256), 0, stream, G, mu, wcopy, K, C, inv, unbias, gamma, beta,
                     running_mean, running_var, momentum, eps, scale, shift, batch_mean, batch_var, err);
  CAPNET_LAUNCH_CHECK();
  return kOk;
}

int fused_block_tiles(long M, int MID) {
  (void)MID;
  return cdiv(M, 64);
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
  const dim3 grid(fused_block_tiles(M, MID)), block(256);
  if (MID == 256) CAPNET_LAUNCH_TIMED((fb_fused_kernel<256, 1>), grid, block, stream, a);
  else if (MID == 128) CAPNET_LAUNCH_TIMED((fb_fused_kernel<128, 1>), grid, block, stream, a);
  else CAPNET_LAUNCH_TIMED((fb_fused_kernel<64, 1>), grid, block, stream, a);
  CAPNET_LAUNCH_CHECK();
  return kOk;
}

}  // namespace capnet
