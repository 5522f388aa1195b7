// The stem of the trunk on the f16 matrix cores: the 7x7 stride-2 pad-3 convolution of the NCHW fp32 image
// (3 channels) to 64 channels, NHWC fp32 output + the batch-statistics partials of the BatchNorm behind it
// (torchvision resnet152.conv1 under encoder.train(): stylenet/model.py:14-24 builds it, train_multitask.py:367 runs it).
//
// Arithmetic as in conv_f16x3.hip: x = h + l in f16, w 2^ew = h' + l' in f16 (per-tensor power of two, so that
// max |w| 2^ew is in [2^13, 2^14)), three products hh' + hl' + lh' accumulated in fp32 by v_mfma_f32_32x32x16_f16,
// the accumulators multiplied by 2^-ew at the end: fp32-grade results, 1/20 of the f32 pipe's matrix time.
//
// Shape of the work: Cin = 3 makes the generic implicit-GEMM loader gather single floats. Here a workgroup takes
// ONE output row segment (b, oh, 128 consecutive ow) x all 64 channels and stages the 21 input rows (channel, kh)
// it needs -- 264 contiguous floats each, 16-B loads straight from the NCHW rows -- as split f16 planes in LDS.
// The GEMM's K axis is (row r = 7 c + kh, tap e = 0..7) with e = kw + 1 and a zero weight at e = 0: then the 8
// values a lane feeds to one MFMA are 8 CONSECUTIVE input columns starting at the even column 2 ow - 4, i.e. one
// 16-B window of the staged row (read as 4 dwords: 4-B aligned, conflict-free with the row stride = 32 banks
// mod 64). 22 rows (the 22nd has zero weights) = 11 MFMA steps of K = 16. The packed weights (45 KB, B-fragment
// order) sit in LDS for the life of the workgroup; workgroups are persistent and prefetch the next segment's
// rows into registers while the MFMAs of the current one run. Column statistics are accumulated per workgroup
// over all its segments and written once: part_sum / part_sq have one row per workgroup.
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
typedef unsigned u4 __attribute__((ext_vector_type(4)));

constexpr int kRows = 22;                 // (channel, kh) rows of the K axis: 21 + one of zero weights
constexpr int kSteps = kRows / 2;         // MFMA steps of K = 16 (two rows x 8 taps)
constexpr int kSeg = 128;                 // output columns of one segment
constexpr int kCols = 2 * kSeg + 8;       // staged input columns of a row: iw0 = 2 ow0 - 4 ... iw0 + 263
constexpr int kGroups = kCols / 4;        // 16-B groups of a staged row (66)
constexpr int kRowDw = 160;               // LDS row stride in dwords: >= kCols / 2, = 32 mod 64
constexpr int kPlane = kRows * kRowDw * 4;                  // bytes of one f16 plane of the staged rows
constexpr int kImgBytes = 2 * kSteps * 2 * 64 * 16;         // planes x steps x k-groups x channels x 16 B
constexpr int kHdrWords = 4;              // image header: [0] ew, [1] bits of max |w| (pack scratch)
constexpr int kLoads = (21 * kGroups + 255) / 256;          // 16-B loads per thread and segment (6)
constexpr int kWgs = 512;
static_assert(kRowDw * 2 >= kCols && kRowDw % 64 == 32, "staged row stride");

struct StemArgs {
  const float* x;
  const unsigned* wimg;
  float* y;
  float* part_sum;
  float* part_sq;
  int B, H, W, OH, OW, chunks, tiles, in_exp;
  int* err;                                // error word (launches without statistics check their outputs)
  int sxb, sxc, sxh;                       // floats
  unsigned tile_mul, tile_sh;              // segment -> (b oh), chunk
  unsigned oh_mul, oh_sh;                  // (b oh) -> b, oh
};

__device__ __forceinline__ void split4(const f32x4 v, h4& h, h4& l) {
  const f2 a = {v[0], v[1]}, b = {v[2], v[3]};
  const h2 ha = __builtin_convertvector(a, h2), hb = __builtin_convertvector(b, h2);      // v_cvt_pk_f16_f32
  const f2 ra = a - __builtin_convertvector(ha, f2), rb = b - __builtin_convertvector(hb, f2);   // exact
  const h2 la = __builtin_convertvector(ra, h2), lb = __builtin_convertvector(rb, h2);
  h = h4{ha[0], ha[1], hb[0], hb[1]};
  l = h4{la[0], la[1], lb[0], lb[1]};
}

__global__ __launch_bounds__(256, 2) void conv_stem_f16x3_kernel(const StemArgs g) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[kImgBytes + 2 * kPlane];
  __shared__ float s_red[2][4][64];
  unsigned char* const wl = lds;                   // packed weights
  unsigned char* const xl = lds + kImgBytes;       // staged rows: plane h | plane l
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const float oscale = ldexpf(1.f, -((int)g.wimg[0] + g.in_exp));
  const float iscale = ldexpf(1.f, g.in_exp);

  // ---- once per workgroup: weights into LDS, the zero-weight row to zero (its products must be finite)
  {
    const u4* src = reinterpret_cast<const u4*>(g.wimg + kHdrWords);
    for (int i = tid; i < kImgBytes / 16; i += 256) *reinterpret_cast<u4*>(wl + i * 16) = src[i];
    for (int i = tid; i < kRowDw; i += 256) {
      *reinterpret_cast<unsigned*>(xl + (21 * kRowDw + i) * 4) = 0u;
      *reinterpret_cast<unsigned*>(xl + kPlane + (21 * kRowDw + i) * 4) = 0u;
    }
  }

  // ---- this thread's share of a segment's rows: item i = tid + 256 u -> row r = i / 66, group q = i % 66
  f32x4 pre[kLoads];
  auto fetch = [&](int tile) {
    const int boh = (int)fast_div((unsigned)tile, g.tile_mul, g.tile_sh), chunk = tile - boh * g.chunks;
    const int b = (int)fast_div((unsigned)boh, g.oh_mul, g.oh_sh), oh = boh - b * g.OH;
    const int iw0 = 2 * chunk * kSeg - 4;
#pragma unroll
    for (int u = 0; u < kLoads; ++u) {
      const int i = tid + 256 * u;
      const int r = i / kGroups, q = i - r * kGroups;
      const int c = r / 7, kh = r - 7 * c;
      const int ih = 2 * oh - 3 + kh, iw = iw0 + 4 * q;
      const bool ok = r < 21 && (unsigned)ih < (unsigned)g.H && iw >= 0 && iw < g.W;       // W % 4 == 0: a group is in or out
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (ok) v = *reinterpret_cast<const f32x4*>(g.x + (long)b * g.sxb + (long)c * g.sxc + (long)ih * g.sxh + iw);
      pre[u] = v;
    }
  };
  auto stage = [&]() {
#pragma unroll
    for (int u = 0; u < kLoads; ++u) {
      const int i = tid + 256 * u;
      const int r = i / kGroups, q = i - r * kGroups;
      if (r < 21) {
        h4 h, l;
        split4(pre[u] * iscale, h, l);            // (the image's power-of-two prescale: exact)
        *reinterpret_cast<h4*>(xl + r * kRowDw * 4 + q * 8) = h;
        *reinterpret_cast<h4*>(xl + kPlane + r * kRowDw * 4 + q * 8) = l;
      }
    }
  };

  float csum[2] = {0.f, 0.f}, csq[2] = {0.f, 0.f};       // this lane's column (li + 32 nb), its rows, all segments
  const int G = (int)gridDim.x;
  int tile = xcd_remap((int)blockIdx.x, G);              // a workgroup's segments: tile, tile + G, ...
  // (xcd_remap on the workgroup id: neighbouring output rows, which share 5 of their 7 input rows, go to one XCD's L2)
  if (tile < g.tiles) fetch(tile);
  __syncthreads();
  for (; tile < g.tiles; tile += G) {
    stage();
    __syncthreads();
    const int boh = (int)fast_div((unsigned)tile, g.tile_mul, g.tile_sh), chunk = tile - boh * g.chunks;
    if (tile + G < g.tiles) fetch(tile + G);
    f32x16 acc[2];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[nb][r] = 0.f;
    const unsigned char* arow = xl + (lh * kRowDw + 32 * wave + li) * 4;           // row 2 s + lh, column 2 ow
    const unsigned char* brow = wl + (lh * 64 + li) * 16;
#pragma unroll
    for (int s = 0; s < kSteps; ++s) {
      h8 a[2], bf[2][2];
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        const unsigned* ap = reinterpret_cast<const unsigned*>(arow + p * kPlane + s * 2 * kRowDw * 4);
        const u4 av = {ap[0], ap[1], ap[2], ap[3]};
        a[p] = __builtin_bit_cast(h8, av);
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
          bf[p][nb] = *reinterpret_cast<const h8*>(brow + ((p * kSteps + s) * 2 * 64 + nb * 32) * 16);
      }
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) {
        acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[1], bf[0][nb], acc[nb], 0, 0, 0);      // l h'
        acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], bf[1][nb], acc[nb], 0, 0, 0);      // h l'
        acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], bf[0][nb], acc[nb], 0, 0, 0);      // h h'
      }
    }
    // ---- segment epilogue: 2^-ew, NHWC store, column statistics of the valid rows
    const int ow_base = chunk * kSeg + 32 * wave + 4 * lh;
    float* yrow = g.y + ((long)boh * g.OW) * 64;
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ow = ow_base + (r & 3) + 8 * (r >> 2);
        const float v = acc[nb][r] * oscale;
        if (ow < g.OW) {
          yrow[(long)ow * 64 + nb * 32 + li] = v;
          csum[nb] += v;
          csq[nb] = fmaf(v, v, csq[nb]);
        }
      }
    }
    __syncthreads();       // every wave is through with the staged rows
  }
  if (!g.part_sum && g.err) flag_nonfinite(csq[0] + csq[1], g.err);
  // ---- the workgroup's statistics: the two half-waves, then the four waves
  if (g.part_sum) {
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
      csum[nb] += __shfl_xor(csum[nb], 32);
      csq[nb] += __shfl_xor(csq[nb], 32);
      if (lh == 0) {
        s_red[0][wave][nb * 32 + li] = csum[nb];
        s_red[1][wave][nb * 32 + li] = csq[nb];
      }
    }
    __syncthreads();
    if (tid < 64) {
      g.part_sum[(long)blockIdx.x * 64 + tid] = (s_red[0][0][tid] + s_red[0][1][tid]) + (s_red[0][2][tid] + s_red[0][3][tid]);
      g.part_sq[(long)blockIdx.x * 64 + tid] = (s_red[1][0][tid] + s_red[1][1][tid]) + (s_red[1][2][tid] + s_red[1][3][tid]);
    }
  }
}

// max |w| as float bits (non-negative floats order like unsigned integers)
__global__ __launch_bounds__(256) void stem_absmax_kernel(const float* __restrict__ w, unsigned* __restrict__ hdr, int n) {
  float m = 0.f;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) m = fmaxf(m, fabsf(w[i]));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  if ((threadIdx.x & 63) == 0 && m > 0.f) atomicMax(hdr + 1, __float_as_uint(m));
}

// one thread per 16-B cell (plane, step, k-group, channel): taps e = 0..7 of row r = 2 step + k-group; w is OIHW [64][3][7][7]
__global__ __launch_bounds__(256) void stem_pack_kernel(const float* __restrict__ w, unsigned* __restrict__ img) {
  const unsigned bits = img[1];
  int ew = 0;
  if (bits != 0u) {      // max |w| 2^ew in [2^13, 2^14): a factor 4 below the f16 range
    ew = 13 - ((int)((bits >> 23) & 0xffu) - 127);
    ew = ew < -100 ? -100 : (ew > 100 ? 100 : ew);
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) img[0] = (unsigned)ew;
  const float ws = ldexpf(1.f, ew);
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= kImgBytes / 16) return;
  const int n = i & 63, gq = (i >> 6) & 1, s = (i >> 7) % kSteps, plane = (i >> 7) / kSteps;
  const int r = 2 * s + gq, c = r / 7, kh = r - 7 * c;
  unsigned out[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    float x[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int e = 2 * q + j;
      x[j] = (r < 21 && e >= 1) ? w[((n * 3 + c) * 7 + kh) * 7 + (e - 1)] * ws : 0.f;
    }
    const _Float16 h0 = (_Float16)x[0], h1 = (_Float16)x[1];
    const _Float16 l0 = (_Float16)(x[0] - (float)h0), l1 = (_Float16)(x[1] - (float)h1);
    const h2 p = plane == 0 ? h2{h0, h1} : h2{l0, l1};
    out[q] = __builtin_bit_cast(unsigned, p);
  }
  unsigned* dst = img + kHdrWords + (long)i * 4;
  dst[0] = out[0]; dst[1] = out[1]; dst[2] = out[2]; dst[3] = out[3];
}

}  // namespace

bool conv_stem_f16x3_eligible(const float* x, long sxb, long sxc, long sxh, long sxw, int Bn, int H, int W, int Cin,
                              int Cout, int k, int stride, int pad) {
  if (!(Cin == 3 && Cout == 64 && k == 7 && stride == 2 && pad == 3)) return false;
  const long OH = (H - 1) / 2 + 1, OW = (W - 1) / 2 + 1;
  return sxw == 1 && W % 4 == 0 && H >= 7 && W >= 8 && aligned16(x) && sxb % 4 == 0 && sxc % 4 == 0 && sxh % 4 == 0 &&
         (long)Bn * sxb < (1l << 31) && (long)Bn * OH * cdiv(OW, kSeg) < (1l << 24) && (long)Bn * OH * OW * 64 < (1l << 31);
}
size_t conv_stem_f16x3_weight_words() { return (size_t)kHdrWords + kImgBytes / 4; }
// rows of the statistics partials one launch writes (one per workgroup)
int conv_stem_f16x3_part_rows(int Bn, int H, int W) {
  const long OH = (H - 1) / 2 + 1, OW = (W - 1) / 2 + 1;
  const long tiles = (long)Bn * OH * cdiv(OW, kSeg);
  return (int)(tiles < kWgs ? tiles : kWgs);
}

int conv_stem_f16x3_pack(const float* w_oihw, unsigned* img, hipStream_t stream) {
  CAPNET_REQUIRE(w_oihw && img && aligned16(img), "conv_stem_f16x3_pack: bad argument");
  CAPNET_HIP_CHECK(hipMemsetAsync(img, 0, kHdrWords * 4, stream));
  hipLaunchKernelGGL(stem_absmax_kernel, dim3(4), dim3(256), 0, stream, w_oihw, img, 64 * 3 * 7 * 7);
  CAPNET_LAUNCH_CHECK();
  hipLaunchKernelGGL(stem_pack_kernel, dim3(cdiv(kImgBytes / 16, 256)), dim3(256), 0, stream, w_oihw, img);
  CAPNET_LAUNCH_CHECK();
  return kOk;
}

// y [B][OH][OW][64] = conv7x7/2(x NCHW, strides in floats) ; part_sum / part_sq [conv_stem_f16x3_part_rows][64] or null
int conv_stem_fwd_f16x3(const float* x, long sxb, long sxc, long sxh, const unsigned* wimg, float* y, float* part_sum,
                        float* part_sq, int Bn, int H, int W, hipStream_t stream, int in_exp, int* err) {
  CAPNET_REQUIRE(x && wimg && y && aligned16(wimg) && in_exp > -64 && in_exp < 64, "conv_stem_fwd_f16x3: bad argument");
  CAPNET_REQUIRE(conv_stem_f16x3_eligible(x, sxb, sxc, sxh, 1, Bn, H, W, 3, 64, 7, 2, 3),
                 "conv_stem_fwd_f16x3: operands not eligible (B=%d %dx%d)", Bn, H, W);
  CAPNET_REQUIRE((part_sum == nullptr) == (part_sq == nullptr), "conv_stem_fwd_f16x3: stats pair");
  StemArgs a{};
  a.x = x; a.wimg = wimg; a.y = y; a.part_sum = part_sum; a.part_sq = part_sq;
  a.B = Bn; a.H = H; a.W = W; a.OH = (H - 1) / 2 + 1; a.OW = (W - 1) / 2 + 1;
  a.in_exp = in_exp; a.err = err;
  a.chunks = cdiv(a.OW, kSeg);
  a.tiles = Bn * a.OH * a.chunks;
  a.sxb = (int)sxb; a.sxc = (int)sxc; a.sxh = (int)sxh;
  magic_div((unsigned)a.chunks, &a.tile_mul, &a.tile_sh);
  magic_div((unsigned)a.OH, &a.oh_mul, &a.oh_sh);
  const int grid = a.tiles < kWgs ? a.tiles : kWgs;
  CAPNET_LAUNCH_TIMED(conv_stem_f16x3_kernel, dim3(grid), dim3(256), stream, a);
  CAPNET_LAUNCH_CHECK();
  return kOk;
}

}  // namespace capnet
