// Input pipeline on the GPU: the reference's torchvision transform chain
//   Resize((336, 336)) -> RandomCrop(224) -> RandomHorizontalFlip() -> ToTensor() -> Normalize(mean, std)
// (stylenet/train_multitask.py:62-69) on uint8 HWC images, producing the fp32 NCHW batch the
// trunk's stem reads. All three kernels are byte/integer work bound by HBM.
//   * resize: Pillow's two-pass antialiased resample for 8-bit images (torchvision 0.2.2 Resize
//     calls PIL.Image.resize(BILINEAR)). The filter coefficients are built on the host exactly as
//     Pillow does (double precision, normalised, rounded to 22-bit fixed point); the kernels do
//     the integer multiply-accumulate, rounding (1 << 21) and clamp of Resample.c, so the output
//     is bit-identical to Pillow's.
//   * crop + flip + ToTensor + Normalize: out = (u8 / 255 - mean[c]) / std[c] in fp32, the same
//     two roundings torch performs.
#include "common.h"
#include "kernels.h"

namespace capnet {

constexpr int kResizePrecisionBits = 32 - 8 - 2;

// out[y][xo][c] = clip8((2^21 + sum_k in[y][xmin[xo] + k][c] * coef[xo][k]) >> 22)
// horizontal: lines = rows, in_stride = Ws*3; vertical: the same kernel on the transposed roles
// (`along` = pixels along the filtered axis, `pitch` = bytes between consecutive taps).
__global__ __launch_bounds__(256) void resample_u8_kernel(
    const unsigned char* __restrict__ src, unsigned char* __restrict__ dst, int out_len, int lines,
    long src_line_stride, long src_tap_pitch, long dst_line_stride, long dst_pix_pitch,
    const int* __restrict__ bounds, const int* __restrict__ coef, int kmax) {
  const long total = (long)lines * out_len * 3;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % 3);
    const long r = i / 3;
    const int xo = (int)(r % out_len);
    const long line = r / out_len;
    const int x0 = bounds[2 * xo], n = bounds[2 * xo + 1];
    const int* k = coef + (long)xo * kmax;
    const unsigned char* p = src + line * src_line_stride + (long)x0 * src_tap_pitch + c;
    int ss = 1 << (kResizePrecisionBits - 1);
    for (int t = 0; t < n; ++t) ss += (int)p[(long)t * src_tap_pitch] * k[t];
    ss >>= kResizePrecisionBits;
    dst[line * dst_line_stride + (long)xo * dst_pix_pitch + c] =
        (unsigned char)(ss < 0 ? 0 : (ss > 255 ? 255 : ss));
  }
}

int resize_u8(const unsigned char* src, int Hs, int Ws, unsigned char* tmp, unsigned char* dst, int Ho,
              int Wo, const int* bounds_h, const int* coef_h, int kmax_h, const int* bounds_v,
              const int* coef_v, int kmax_v, hipStream_t stream) {
  CAPNET_REQUIRE(src && tmp && dst && bounds_h && coef_h && bounds_v && coef_v, "resize_u8: null argument");
  CAPNET_REQUIRE(Hs > 0 && Ws > 0 && Ho > 0 && Wo > 0 && kmax_h > 0 && kmax_v > 0, "resize_u8: bad size");
  // horizontal pass: [Hs][Ws][3] -> tmp [Hs][Wo][3]
  {
    const long total = (long)Hs * Wo * 3;
    const int blocks = (int)(cdiv(total, 256) > 8192 ? 8192 : cdiv(total, 256));
    hipLaunchKernelGGL(resample_u8_kernel, dim3(blocks), dim3(256), 0, stream, src, tmp, Wo, Hs,
                       (long)Ws * 3, 3l, (long)Wo * 3, 3l, bounds_h, coef_h, kmax_h);
  }
  // vertical pass: tmp [Hs][Wo][3] -> dst [Ho][Wo][3]; a "line" is a column
  {
    const long total = (long)Ho * Wo * 3;
    const int blocks = (int)(cdiv(total, 256) > 8192 ? 8192 : cdiv(total, 256));
    hipLaunchKernelGGL(resample_u8_kernel, dim3(blocks), dim3(256), 0, stream, tmp, dst, Ho, Wo, 3l,
                       (long)Wo * 3, 3l, (long)Wo * 3, bounds_v, coef_v, kmax_v);
  }
  CAPNET_LAUNCH_CHECK();
  return kOk;
}

// src [B][Hs][Ws][3] uint8 -> dst [B][3][Hc][Wc] fp32; params[b] = {top, left, flip}
__global__ __launch_bounds__(256) void crop_flip_normalize_kernel(
    const unsigned char* __restrict__ src, int Hs, int Ws, const int* __restrict__ params,
    float* __restrict__ dst, int B, int Hc, int Wc, float m0, float m1, float m2, float s0, float s1,
    float s2) {
  const long total = (long)B * 3 * Hc * Wc;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long)gridDim.x * blockDim.x) {
    const int x = (int)(i % Wc);
    long r = i / Wc;
    const int y = (int)(r % Hc);
    r /= Hc;
    const int c = (int)(r % 3);
    const int b = (int)(r / 3);
    const int top = params[3 * b], left = params[3 * b + 1], flip = params[3 * b + 2];
    const int sx = left + (flip ? Wc - 1 - x : x);
    const float v = (float)src[(((long)b * Hs + top + y) * Ws + sx) * 3 + c] / 255.f;
    const float mean = c == 0 ? m0 : (c == 1 ? m1 : m2);
    const float sd = c == 0 ? s0 : (c == 1 ? s1 : s2);
    dst[i] = (v - mean) / sd;
  }
}

int crop_flip_normalize(const unsigned char* src, int B, int Hs, int Ws, const int* params, float* dst,
                        int Hc, int Wc, const float* mean, const float* stdv, hipStream_t stream) {
  CAPNET_REQUIRE(src && params && dst && mean && stdv, "crop_flip_normalize: null argument");
  CAPNET_REQUIRE(B >= 0 && Hc > 0 && Wc > 0 && Hc <= Hs && Wc <= Ws, "crop_flip_normalize: crop %dx%d of %dx%d",
                 Hc, Wc, Hs, Ws);
  if (B == 0) return kOk;
  const long total = (long)B * 3 * Hc * Wc;
  const int blocks = (int)(cdiv(total, 256) > 16384 ? 16384 : cdiv(total, 256));
  hipLaunchKernelGGL(crop_flip_normalize_kernel, dim3(blocks), dim3(256), 0, stream, src, Hs, Ws, params,
                     dst, B, Hc, Wc, mean[0], mean[1], mean[2], stdv[0], stdv[1], stdv[2]);
  CAPNET_LAUNCH_CHECK();
  return kOk;
}

}  // namespace capnet
