// Implicit-GEMM convolution for the ResNet-152 trunk on the f32-input MFMA.
//
//   y[m][co] = sum_{r,s,ci} pre(x[b][oh*stride+r-pad][ow*stride+s-pad][ci]) * w[co][r][s][ci]
//   m = (b*OH + oh)*OW + ow                      (NHWC output, raw conv result, no bias)
//   pre(v) = relu?(v*in_scale[ci] + in_shift[ci]) -- the PREVIOUS layer's train-mode
//            BatchNorm (+ReLU), applied while the A tile is staged, so a conv output is
//            written once and read once; zero padding is applied after pre().
// The epilogue also emits per-workgroup column sums / sums of squares of the raw output:
// the batch statistics of THIS layer's BatchNorm (reduced in double by bn_finalize).
//
// Replaces torchvision resnet152's Conv2d+BatchNorm2d(train) pairs called at
// stylenet/model.py:15-18,24 (155 convs); weights are packed [Cout][KH][KW][Cin].
#include "common.h"
#include "mfma_core.h"
#include "kernels.h"

namespace capnet {

struct ConvArgs {
  const float* x;
  const float* w;
  float* y;
  const float* in_scale;
  const float* in_shift;
  float* part_sum;
  float* part_sq;
  int Bn, H, W, Cin, OH, OW, Cout, KH, KW, stride, pad;
  long sxb, sxh, sxw, sxc;
  int M, K, Kw;
  int relu_in;
  int tiles_m, tiles_n;
};

// A-operand loader, channel-contiguous fast path: Cin % BK == 0, sxc == 1, 16-B aligned rows,
// input addressed with 32-bit element offsets. Loads are unconditional (tap coordinates are
// clamped into the image, the result is zeroed in store() when the tap is padding), so no
// wait for memory is ever placed ahead of the MFMAs.
template <int BR, int BK, int LD>
struct ConvLoaderFast {
  static constexpr int TPR = BK / 4;
  static constexpr int RPP = kGemmThreads / TPR;
  static constexpr int PASSES = BR / RPP;
  const float* x;
  const float* in_scale;
  const float* in_shift;
  int H, W, Cin, KW, sxh, sxw, relu;
  int boff[PASSES];
  int ih0[PASSES], iw0[PASSES];
  float4 v[PASSES];
  float4 sc, sh;
  unsigned ok;  // bit ps: tap in bounds for pass ps

  __device__ __forceinline__ void init(const ConvArgs& g, int m0) {
    x = g.x; in_scale = g.in_scale; in_shift = g.in_shift;
    H = g.H; W = g.W; Cin = g.Cin; KW = g.KW; sxh = (int)g.sxh; sxw = (int)g.sxw;
    relu = g.relu_in;
    const int rl = threadIdx.x / TPR;
    const int ohw = g.OH * g.OW;
#pragma unroll
    for (int ps = 0; ps < PASSES; ++ps) {
      const int m = m0 + rl + ps * RPP;
      const int mc = m < g.M ? m : g.M - 1;
      const int b = mc / ohw;
      const int rem = mc - b * ohw;
      const int oh = rem / g.OW;
      const int ow = rem - oh * g.OW;
      // rows past M get a coordinate that is out of the image for every tap
      ih0[ps] = m < g.M ? oh * g.stride - g.pad : -(1 << 20);
      iw0[ps] = ow * g.stride - g.pad;
      boff[ps] = b * (int)g.sxb;
    }
  }
  __device__ __forceinline__ void load(int k0) {
    const int kc = threadIdx.x % TPR;
    const int tap = k0 / Cin;
    const int c = k0 - tap * Cin + 4 * kc;
    const int r = tap / KW;
    const int s = tap - r * KW;
    ok = 0;
#pragma unroll
    for (int ps = 0; ps < PASSES; ++ps) {
      const int ih = ih0[ps] + r, iw = iw0[ps] + s;
      const bool inb = (unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W;
      const int ihc = min(max(ih, 0), H - 1), iwc = min(max(iw, 0), W - 1);
      v[ps] = *reinterpret_cast<const float4*>(x + (boff[ps] + ihc * sxh + iwc * sxw + c));
      ok |= (inb ? 1u : 0u) << ps;
    }
    if (in_scale) {
      sc = *reinterpret_cast<const float4*>(in_scale + c);
      sh = *reinterpret_cast<const float4*>(in_shift + c);
    }
  }
  __device__ __forceinline__ void store(float* lds) const {
    const int kc = threadIdx.x % TPR, rl = threadIdx.x / TPR;
    const bool pre = in_scale != nullptr;
#pragma unroll
    for (int ps = 0; ps < PASSES; ++ps) {
      float4 t = v[ps];
      if (pre) {
        t.x = fmaf(t.x, sc.x, sh.x);
        t.y = fmaf(t.y, sc.y, sh.y);
        t.z = fmaf(t.z, sc.z, sh.z);
        t.w = fmaf(t.w, sc.w, sh.w);
        if (relu) {
          t.x = fmaxf(t.x, 0.f);
          t.y = fmaxf(t.y, 0.f);
          t.z = fmaxf(t.z, 0.f);
          t.w = fmaxf(t.w, 0.f);
        }
      }
      t = mask4(t, ((ok >> ps) & 1u) ? 4 : 0);
      float* d = lds + (4 * kc) * LD + rl + ps * RPP;
      d[0 * LD] = t.x;
      d[1 * LD] = t.y;
      d[2 * LD] = t.z;
      d[3 * LD] = t.w;
    }
  }
};

// Generic gather path (the 7x7 stem: Cin = 3, NCHW input): per-element tap decomposition.
template <int BR, int BK, int LD>
struct ConvLoaderGeneric {
  static constexpr int TPR = BK / 4;
  static constexpr int RPP = kGemmThreads / TPR;
  static constexpr int PASSES = BR / RPP;
  ConvArgs g;  // by-value copy: uniform fields stay in SGPRs (no kernarg address taken)
  long boff[PASSES];
  int ih0[PASSES], iw0[PASSES];
  float v[PASSES][4];

  __device__ __forceinline__ void init(const ConvArgs& g_, int m0) {
    g = g_;
    const int rl = threadIdx.x / TPR;
    const int ohw = g.OH * g.OW;
#pragma unroll
    for (int ps = 0; ps < PASSES; ++ps) {
      const int m = m0 + rl + ps * RPP;
      if (m < g.M) {
        const int b = m / ohw;
        const int rem = m - b * ohw;
        const int oh = rem / g.OW;
        const int ow = rem - oh * g.OW;
        ih0[ps] = oh * g.stride - g.pad;
        iw0[ps] = ow * g.stride - g.pad;
        boff[ps] = (long)b * g.sxb;
      } else {
        ih0[ps] = -(1 << 28);
        iw0[ps] = 0;
        boff[ps] = 0;
      }
    }
  }
  __device__ __forceinline__ void load(int k0) {
    const int kc = threadIdx.x % TPR;
    const bool pre = g.in_scale != nullptr;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int k = k0 + 4 * kc + j;
      const bool kin = k < g.K;
      const int tap = kin ? k / g.Cin : 0;
      const int c = kin ? k - tap * g.Cin : 0;
      const int r = tap / g.KW;
      const int s = tap - r * g.KW;
      float scv = 1.f, shv = 0.f;
      if (pre && kin) {
        scv = g.in_scale[c];
        shv = g.in_shift[c];
      }
#pragma unroll
      for (int ps = 0; ps < PASSES; ++ps) {
        const int ih = ih0[ps] + r, iw = iw0[ps] + s;
        float t = 0.f;
        if (kin && (unsigned)ih < (unsigned)g.H && (unsigned)iw < (unsigned)g.W) {
          t = g.x[boff[ps] + (long)ih * g.sxh + (long)iw * g.sxw + (long)c * g.sxc];
          if (pre) {
            t = fmaf(t, scv, shv);
            if (g.relu_in) t = fmaxf(t, 0.f);
          }
        }
        v[ps][j] = t;
      }
    }
  }
  __device__ __forceinline__ void store(float* lds) const {
    const int kc = threadIdx.x % TPR, rl = threadIdx.x / TPR;
#pragma unroll
    for (int ps = 0; ps < PASSES; ++ps) {
      float* d = lds + (4 * kc) * LD + rl + ps * RPP;
#pragma unroll
      for (int j = 0; j < 4; ++j) d[j * LD] = v[ps][j];
    }
  }
};

template <int BM, int BN, int BK, bool FAST>
__global__ __launch_bounds__(kGemmThreads) void conv_f32_kernel(ConvArgs g) {
  using T = TileCfg<BM, BN, BK>;
  __shared__ __attribute__((aligned(16))) float lds[2 * T::STAGE_ELEMS];

  const int nwg = g.tiles_m * g.tiles_n;
  const int id = xcd_remap(blockIdx.x, nwg);
  const int tm = id / g.tiles_n, tn = id - tm * g.tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;

  using ALoad = typename std::conditional<FAST, ConvLoaderFast<BM, BK, T::LDA>,
                                          ConvLoaderGeneric<BM, BK, T::LDA>>::type;
  ALoad al;
  al.init(g, m0);
  LoaderKContig<BN, BK, T::LDB, true> bl;
  bl.init(g.w, g.Kw, g.Cout, g.Kw, n0);

  f32x16 acc[T::MT][T::NT];
  gemm_block_mainloop<T>(al, bl, g.Kw, lds, acc);

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int li = lane & 31, lh = lane >> 5;
#pragma unroll
  for (int nt = 0; nt < T::NT; ++nt) {
    const int n = n0 + wn * (BN / 2) + nt * 32 + li;
    if (n >= g.Cout) continue;
#pragma unroll
    for (int mt = 0; mt < T::MT; ++mt) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * (BM / 2) + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (m < g.M) g.y[(long)m * g.Cout + n] = acc[mt][nt][r];
      }
    }
  }
  if (g.part_sum) {
    // rows >= M contributed exact zeros (zero-filled A), so the sums are over valid rows only
    block_col_stats<T>(acc, lds, g.part_sum + (long)tm * g.Cout, g.part_sq + (long)tm * g.Cout,
                       n0, g.Cout);
  }
}

template <int BM, int BN>
static void launch_conv(ConvArgs& g, bool fast, hipStream_t stream) {
  g.tiles_m = cdiv(g.M, BM);
  g.tiles_n = cdiv(g.Cout, BN);
  dim3 grid(g.tiles_m * g.tiles_n);
  if (fast)
    hipLaunchKernelGGL((conv_f32_kernel<BM, BN, 16, true>), grid, dim3(kGemmThreads), 0, stream, g);
  else
    hipLaunchKernelGGL((conv_f32_kernel<BM, BN, 16, false>), grid, dim3(kGemmThreads), 0, stream, g);
}

int conv_tile_rows(int tile) { return tile == 64 ? 64 : 128; }

// tile: 0 = auto, 128 = 128x128, 64 = 64x64, 12864 = 128x64
int conv2d_fwd(const float* x, long sxb, long sxh, long sxw, long sxc, const float* w_packed,
               int Kw, float* y, const float* in_scale, const float* in_shift, int relu_in,
               float* part_sum, float* part_sq, int Bn, int H, int W, int Cin, int Cout, int KH,
               int KW, int stride, int pad, int tile, hipStream_t stream) {
  CAPNET_REQUIRE(x && w_packed && y, "conv2d_fwd: null pointer");
  CAPNET_REQUIRE(Bn > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0 && KH > 0 && KW > 0 &&
                     stride > 0 && pad >= 0,
                 "conv2d_fwd: bad shape");
  ConvArgs g;
  g.x = x; g.w = w_packed; g.y = y;
  g.in_scale = in_scale; g.in_shift = in_shift;
  g.part_sum = part_sum; g.part_sq = part_sq;
  g.Bn = Bn; g.H = H; g.W = W; g.Cin = Cin; g.Cout = Cout; g.KH = KH; g.KW = KW;
  g.stride = stride; g.pad = pad;
  g.OH = (H + 2 * pad - KH) / stride + 1;
  g.OW = (W + 2 * pad - KW) / stride + 1;
  g.sxb = sxb; g.sxh = sxh; g.sxw = sxw; g.sxc = sxc;
  const long M = (long)Bn * g.OH * g.OW;
  CAPNET_REQUIRE(M < (1L << 31), "conv2d_fwd: too many output pixels");
  g.M = (int)M;
  g.K = KH * KW * Cin;
  g.Kw = Kw;
  CAPNET_REQUIRE(Kw >= g.K && Kw % 16 == 0, "conv2d_fwd: packed weight stride %d (K=%d)", Kw, g.K);
  CAPNET_REQUIRE(aligned16(w_packed), "conv2d_fwd: weights must be 16-B aligned");
  CAPNET_REQUIRE((in_scale == nullptr) == (in_shift == nullptr), "conv2d_fwd: scale/shift pair");
  CAPNET_REQUIRE((part_sum == nullptr) == (part_sq == nullptr), "conv2d_fwd: stats pair");
  g.relu_in = relu_in;
  const bool fast = (Cin % 16 == 0) && sxc == 1 && (sxw % 4 == 0) && (sxh % 4 == 0) &&
                    (sxb % 4 == 0) && aligned16(x) && ((long)Bn * sxb < (1L << 31)) &&
                    (!in_scale || (aligned16(in_scale) && aligned16(in_shift)));
  if (tile == 0) tile = conv_auto_tile(g.M, Cout);
  if (tile == 128) launch_conv<128, 128>(g, fast, stream);
  else if (tile == 64) launch_conv<64, 64>(g, fast, stream);
  else if (tile == 12864) launch_conv<128, 64>(g, fast, stream);
  else CAPNET_REQUIRE(false, "conv2d_fwd: unknown tile %d", tile);
  CAPNET_LAUNCH_CHECK();
  return kOk;
}

int conv_auto_tile(int M, int Cout) {
  if (Cout <= 64) return 12864;
  const long t128 = (long)cdiv(M, 128) * cdiv(Cout, 128);
  return t128 >= 384 ? 128 : 64;
}

int conv_tiles_m(int M, int tile) { return cdiv(M, tile == 64 ? 64 : 128); }

}  // namespace capnet
