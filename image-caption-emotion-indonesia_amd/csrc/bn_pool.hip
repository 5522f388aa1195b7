// BatchNorm2d bookkeeping and the memory-bound elementwise/pooling kernels of the trunk.
// All activations are NHWC fp32; every kernel here is HBM-bound and moves 16 B per lane.
// Semantics follow torch.nn.BatchNorm2d as used by torchvision's resnet152 under
// encoder.train() (stylenet/train_multitask.py:367, stylenet/model.py:23-24): batch mean and
// BIASED variance normalise, running_var is updated with the UNBIASED variance, momentum 0.1.
#include <cstdlib>

#include "common.h"
#include "kernels.h"

namespace capnet {

// ---- batch statistics -> (scale, shift), running-stat update --------------------------
// part_sum/part_sq: [tiles][C] per-workgroup partials from the conv epilogue. Reduced in
// double so that E[x^2]-E[x]^2 does not cancel in fp32.
// 16 channels (one 64-B segment) x 64 partial rows per workgroup: C/16 workgroups, and each
// thread keeps 16 independent loads in flight -- the kernel is pure latency otherwise.
constexpr int kFinCh = 16, kFinRows = 64;
// ... and for at most 128 partial rows (the 14 x 14 and 7 x 7 maps: 119 of the trunk's 155 BatchNorms) 16 rows = 256
// threads: beside the convolutions' eight-wave workgroups a 1 024-thread block waits for a CU with sixteen free wave
// slots, a 256-thread one for four (+1.2 % images/s in the pipelined step; 512 threads up to 512 partial rows: +0.5 %;
// CAPNET_FIN_SMALL=0 / CAPNET_FIN_MID=0 for A/B)
constexpr int kFinRowsSmall = 16, kFinSmallMaxTiles = 128;

// running statistic <- (1 - momentum) running + momentum batch, with ONE rounding pattern wherever it is applied: the
// update fused into bn_finalize and the deferred bn_running_update_kernel (TrunkPipeline) must agree bit for bit, and
// left to the compiler the two kernels contracted different halves of the expression into an fma
__device__ __forceinline__ float bn_running_blend(float running, float batch, float momentum) {
  return __builtin_fmaf(momentum, batch, (1.f - momentum) * running);
}

template <int ROWS>
__global__ __launch_bounds__(kFinCh* ROWS) void bn_finalize_kernel(
    const float* __restrict__ part_sum, const float* __restrict__ part_sq, int tiles, int C,
    double inv_count, double unbias, const float* __restrict__ gamma,
    const float* __restrict__ beta, float* __restrict__ running_mean,
    float* __restrict__ running_var, float momentum, float eps, float* __restrict__ scale,
    float* __restrict__ shift, float* __restrict__ batch_mean, float* __restrict__ batch_var, int* __restrict__ err) {
  __shared__ double s_sum[ROWS][kFinCh + 1];
  __shared__ double s_sq[ROWS][kFinCh + 1];
  const int cx = threadIdx.x % kFinCh, ry = threadIdx.x / kFinCh;
  const int c = blockIdx.x * kFinCh + cx;
  double s = 0.0, q = 0.0;
  if (c < C) {
    // batches of 8 independent (clamped, masked) loads per thread: the reduction is pure latency
    for (int t = ry; t < tiles; t += 8 * ROWS) {
      float a[8], b[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int tu = min(t + u * ROWS, tiles - 1);
        a[u] = part_sum[(long)tu * C + c];
        b[u] = part_sq[(long)tu * C + c];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const bool ok = t + u * ROWS < tiles;
        s += ok ? (double)a[u] : 0.0;
        q += ok ? (double)b[u] : 0.0;
      }
    }
  }
  s_sum[ry][cx] = s;
  s_sq[ry][cx] = q;
  __syncthreads();
  if (ry == 0 && c < C) {
#pragma unroll 8
    for (int r = 1; r < ROWS; ++r) {
      s += s_sum[r][cx];
      q += s_sq[r][cx];
    }
    // a non-finite sum of squares = a non-finite output of the convolution: an activation beyond the f16 range of the
    // split operands (65 504 after the layer's power-of-two prescale), or a genuine fp32 overflow
    if (err && !(q < __builtin_inf())) atomicOr(err, 8);
    const double mean = s * inv_count;
    double var = q * inv_count - mean * mean;
    if (var < 0.0) var = 0.0;
    const float invstd = (float)(1.0 / sqrt(var + (double)eps));
    const float g = gamma ? gamma[c] : 1.f;
    const float b = beta ? beta[c] : 0.f;
    const float sc = g * invstd;
    scale[c] = sc;
    shift[c] = b - (float)mean * sc;
    if (running_mean) {
      running_mean[c] = bn_running_blend(running_mean[c], (float)mean, momentum);
      running_var[c] = bn_running_blend(running_var[c], (float)(var * unbias), momentum);
    }
    if (batch_mean) {   // deferred running-statistics update (bn_running_update_multi)
      batch_mean[c] = (float)mean;
      batch_var[c] = (float)(var * unbias);
    }
  }
}

int bn_finalize(const float* part_sum, const float* part_sq, int tiles, int C, long count,
                const float* gamma, const float* beta, float* running_mean, float* running_var,
                float momentum, float eps, float* scale, float* shift, hipStream_t stream,
                float* batch_mean, float* batch_var, int* err) {
  CAPNET_REQUIRE(part_sum && part_sq && scale && shift && tiles > 0 && C > 0 && count > 0,
                 "bn_finalize: bad argument");
  CAPNET_REQUIRE((batch_mean == nullptr) == (batch_var == nullptr), "bn_finalize: batch stat pair");
  const double inv = 1.0 / (double)count;
  const double unbias = count > 1 ? (double)count / (double)(count - 1) : 1.0;
  if (tiles <= kFinSmallMaxTiles)
    hipLaunchKernelGGL(bn_finalize_kernel<kFinRowsSmall>, dim3(cdiv(C, kFinCh)), dim3(kFinCh * kFinRowsSmall), 0,
                       stream, part_sum, part_sq, tiles, C, inv, unbias, gamma, beta, running_mean,
                       running_var, momentum, eps, scale, shift, batch_mean, batch_var, err);
  else if (tiles <= 512)
    hipLaunchKernelGGL(bn_finalize_kernel<32>, dim3(cdiv(C, kFinCh)), dim3(kFinCh * 32), 0,
                       stream, part_sum, part_sq, tiles, C, inv, unbias, gamma, beta, running_mean,
                       running_var, momentum, eps, scale, shift, batch_mean, batch_var, err);
  else
    hipLaunchKernelGGL(bn_finalize_kernel<kFinRows>, dim3(cdiv(C, kFinCh)), dim3(kFinCh * kFinRows), 0,
                       stream, part_sum, part_sq, tiles, C, inv, unbias, gamma, beta, running_mean,
                       running_var, momentum, eps, scale, shift, batch_mean, batch_var, err);
  CAPNET_LAUNCH_CHECK();
  return kOk;
}

// eval mode: (scale, shift) from the running statistics
__global__ void bn_eval_kernel(const float* __restrict__ gamma, const float* __restrict__ beta,
                               const float* __restrict__ rm, const float* __restrict__ rv,
                               float eps, int C, float* __restrict__ scale,
                               float* __restrict__ shift) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float invstd = 1.f / sqrtf(rv[c] + eps);
  const float sc = (gamma ? gamma[c] : 1.f) * invstd;
  scale[c] = sc;
  shift[c] = (beta ? beta[c] : 0.f) - rm[c] * sc;
}

int bn_eval_scale_shift(const float* gamma, const float* beta, const float* rm, const float* rv,
                        float eps, int C, float* scale, float* shift, hipStream_t stream) {
  CAPNET_REQUIRE(rm && rv && scale && shift && C > 0, "bn_eval_scale_shift: bad argument");
  hipLaunchKernelGGL(bn_eval_kernel, dim3(cdiv(C, 256)), dim3(256), 0, stream, gamma, beta, rm,
                     rv, eps, C, scale, shift);
  CAPNET_LAUNCH_CHECK();
  return kOk;
}

// eval mode, every BatchNorm of the trunk in a few launches: up to kBnEvalMax BNs per launch, their
// pointers travel in the kernel arguments
constexpr int kBnEvalMax = 32;
struct BnEvalTable {
  const float* gamma[kBnEvalMax];
  const float* beta[kBnEvalMax];
  const float* rm[kBnEvalMax];
  const float* rv[kBnEvalMax];
  float* scale[kBnEvalMax];
  float* shift[kBnEvalMax];
  int C[kBnEvalMax];
};

__global__ __launch_bounds__(256) void bn_eval_multi_kernel(BnEvalTable t, float eps) {
  const int k = blockIdx.y;
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= t.C[k]) return;
  const float invstd = 1.f / sqrtf(t.rv[k][c] + eps);
  const float sc = (t.gamma[k] ? t.gamma[k][c] : 1.f) * invstd;
  t.scale[k][c] = sc;
  t.shift[k][c] = (t.beta[k] ? t.beta[k][c] : 0.f) - t.rm[k][c] * sc;
}

int bn_eval_multi(int n, const float* const* gamma, const float* const* beta, const float* const* rm,
                  const float* const* rv, const int* C, float* const* scale, float* const* shift,
                  float eps, hipStream_t stream) {
  CAPNET_REQUIRE(n >= 0 && rm && rv && C && scale && shift, "bn_eval_multi: bad argument");
  for (int i0 = 0; i0 < n; i0 += kBnEvalMax) {
    BnEvalTable t;
    const int cnt = n - i0 < kBnEvalMax ? n - i0 : kBnEvalMax;
    int maxc = 0;
    for (int k = 0; k < cnt; ++k) {
      CAPNET_REQUIRE(rm[i0 + k] && rv[i0 + k] && scale[i0 + k] && shift[i0 + k] && C[i0 + k] > 0,
                     "bn_eval_multi: BN %d", i0 + k);
      t.gamma[k] = gamma ? gamma[i0 + k] : nullptr;
      t.beta[k] = beta ? beta[i0 + k] : nullptr;
      t.rm[k] = rm[i0 + k]; t.rv[k] = rv[i0 + k];
      t.scale[k] = scale[i0 + k]; t.shift[k] = shift[i0 + k];
      t.C[k] = C[i0 + k];
      maxc = C[i0 + k] > maxc ? C[i0 + k] : maxc;
    }
    hipLaunchKernelGGL(bn_eval_multi_kernel, dim3(cdiv(maxc, 256), cnt), dim3(256), 0, stream, t, eps);
  }
  CAPNET_LAUNCH_CHECK();
  return kOk;
}

// running statistics of many BatchNorms from the batch statistics bn_finalize left behind
// (same arithmetic as the inline update): the trunk defers them to the end of a pass so that two
// passes in flight on different streams update them in pass order (capnet.train.TrunkPipeline)
struct BnRunTable {
  const float* mean[kBnEvalMax];
  const float* var[kBnEvalMax];
  float* rm[kBnEvalMax];
  float* rv[kBnEvalMax];
  int C[kBnEvalMax];
};

__global__ __launch_bounds__(256) void bn_running_update_kernel(BnRunTable t, float momentum) {
  const int k = blockIdx.y;
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= t.C[k]) return;
  t.rm[k][c] = bn_running_blend(t.rm[k][c], t.mean[k][c], momentum);
  t.rv[k][c] = bn_running_blend(t.rv[k][c], t.var[k][c], momentum);
}

int bn_running_update_multi(int n, const float* const* mean, const float* const* var, float* const* rm,
                            float* const* rv, const int* C, float momentum, hipStream_t stream) {
  CAPNET_REQUIRE(n >= 0 && mean && var && rm && rv && C, "bn_running_update_multi: bad argument");
  for (int i0 = 0; i0 < n; i0 += kBnEvalMax) {
    BnRunTable t;
    const int cnt = n - i0 < kBnEvalMax ? n - i0 : kBnEvalMax;
    int maxc = 0;
    for (int k = 0; k < cnt; ++k) {
      CAPNET_REQUIRE(mean[i0 + k] && var[i0 + k] && rm[i0 + k] && rv[i0 + k] && C[i0 + k] > 0,
                     "bn_running_update_multi: BN %d", i0 + k);
      t.mean[k] = mean[i0 + k]; t.var[k] = var[i0 + k];
      t.rm[k] = rm[i0 + k]; t.rv[k] = rv[i0 + k];
      t.C[k] = C[i0 + k];
      maxc = C[i0 + k] > maxc ? C[i0 + k] : maxc;
    }
    hipLaunchKernelGGL(bn_running_update_kernel, dim3(cdiv(maxc, 256), cnt), dim3(256), 0, stream, t,
                       momentum);
  }
  CAPNET_LAUNCH_CHECK();
  return kOk;
}

// ---- bottleneck tail: out = relu(bn3(y) + identity) or relu(bn3(y) + bn_ds(r)) ---------
// Two independent 16-B elements per thread and iteration. (Non-temporal loads of the two dead
// inputs were measured: 0.05 ms slower over the trunk, so the loads are plain.)
typedef float bn_f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void bn_add_relu_kernel(
    const bn_f32x4* __restrict__ y, const float* __restrict__ s1, const float* __restrict__ t1,
    const bn_f32x4* __restrict__ res, const float* __restrict__ s2, const float* __restrict__ t2,
    bn_f32x4* __restrict__ out, long n4, int C4) {
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i0 = (long)blockIdx.x * blockDim.x + threadIdx.x; i0 < n4; i0 += 2 * stride) {
    const long idx[2] = {i0, i0 + stride};
    bn_f32x4 a[2], r[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const long i = idx[u] < n4 ? idx[u] : i0;
      a[u] = y[i];
      if (res) r[u] = res[i];
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      if (idx[u] >= n4) continue;
      const int c = (int)(idx[u] % C4) * 4;
      const bn_f32x4 sc = *reinterpret_cast<const bn_f32x4*>(s1 + c);
      const bn_f32x4 sh = *reinterpret_cast<const bn_f32x4*>(t1 + c);
      bn_f32x4 v;
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] = fmaf(a[u][k], sc[k], sh[k]);
      if (res) {
        bn_f32x4 rr = r[u];
        if (s2) {
          const bn_f32x4 sc2 = *reinterpret_cast<const bn_f32x4*>(s2 + c);
          const bn_f32x4 sh2 = *reinterpret_cast<const bn_f32x4*>(t2 + c);
#pragma unroll
          for (int k = 0; k < 4; ++k) rr[k] = fmaf(rr[k], sc2[k], sh2[k]);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] += rr[k];
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] = fmaxf(v[k], 0.f);
      out[idx[u]] = v;
    }
  }
}

int bn_add_relu(const float* y, const float* s1, const float* t1, const float* res,
                const float* s2, const float* t2, float* out, long rows, int C,
                hipStream_t stream) {
  CAPNET_REQUIRE(y && s1 && t1 && out && rows > 0 && C > 0 && C % 4 == 0,
                 "bn_add_relu: bad argument");
  CAPNET_REQUIRE((s2 == nullptr) == (t2 == nullptr), "bn_add_relu: s2/t2 pair");
  const long n4 = rows * (C / 4);
  const int blocks = (int)(n4 / 256 < 1 ? 1 : (n4 / 256 > 8192 ? 8192 : n4 / 256));
  hipLaunchKernelGGL(bn_add_relu_kernel, dim3(blocks), dim3(256), 0, stream, (const bn_f32x4*)y, s1,
                     t1, (const bn_f32x4*)res, s2, t2, (bn_f32x4*)out, n4, C / 4);
  CAPNET_LAUNCH_CHECK();
  return kOk;
}

// ---- stem tail: relu(bn1(y)) then MaxPool 3x3 / stride 2 / pad 1 -------------------------
__global__ __launch_bounds__(256) void bn_relu_maxpool_kernel(
    const float* __restrict__ y, const float* __restrict__ sc, const float* __restrict__ sh,
    float* __restrict__ out, int Bn, int H, int W, int C, int OH, int OW) {
  const int C4 = C / 4;
  const long total = (long)Bn * OH * OW * C4;
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int c = (int)(i % C4) * 4;
    long p = i / C4;
    const int ow = (int)(p % OW); p /= OW;
    const int oh = (int)(p % OH);
    const int b = (int)(p / OH);
    const float4 s = *reinterpret_cast<const float4*>(sc + c);
    const float4 t = *reinterpret_cast<const float4*>(sh + c);
    float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const int ih = oh * 2 - 1 + r;
      if ((unsigned)ih >= (unsigned)H) continue;
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        const int iw = ow * 2 - 1 + q;
        if ((unsigned)iw >= (unsigned)W) continue;
        const float4 a = *reinterpret_cast<const float4*>(y + (((long)b * H + ih) * W + iw) * C + c);
        m.x = fmaxf(m.x, fmaxf(fmaf(a.x, s.x, t.x), 0.f));
        m.y = fmaxf(m.y, fmaxf(fmaf(a.y, s.y, t.y), 0.f));
        m.z = fmaxf(m.z, fmaxf(fmaf(a.z, s.z, t.z), 0.f));
        m.w = fmaxf(m.w, fmaxf(fmaf(a.w, s.w, t.w), 0.f));
      }
    }
    *reinterpret_cast<float4*>(out + (((long)b * OH + oh) * OW + ow) * C + c) = m;
  }
}

int bn_relu_maxpool(const float* y, const float* scale, const float* shift, float* out, int Bn,
                    int H, int W, int C, hipStream_t stream) {
  CAPNET_REQUIRE(y && scale && shift && out && C % 4 == 0, "bn_relu_maxpool: bad argument");
  const int OH = (H + 2 - 3) / 2 + 1, OW = (W + 2 - 3) / 2 + 1;
  const long total = (long)Bn * OH * OW * (C / 4);
  const int blocks = (int)(total / 256 < 1 ? 1 : (total / 256 > 8192 ? 8192 : total / 256));
  hipLaunchKernelGGL(bn_relu_maxpool_kernel, dim3(blocks), dim3(256), 0, stream, y, scale, shift,
                     out, Bn, H, W, C, OH, OW);
  CAPNET_LAUNCH_CHECK();
  return kOk;
}

// ---- global average pool: [B][HW][C] -> [B][C] --------------------------------------------
// A workgroup takes 64 channel quads of a sample; its four waves take the pixels p = wave (mod 4), eight loads in flight
// each, and the four sums meet in LDS in a fixed order. (One thread per channel quad walking all pixels was a chain of HW
// dependent round trips on B C / 1024 workgroups: 86 us for the attention decoder's 12 x 196 x 2048 map, 19 MB.)
__global__ __launch_bounds__(256) void avgpool_kernel(const float* __restrict__ x,
                                                      float* __restrict__ out, int Bn, int HW,
                                                      int C, float inv, int* __restrict__ err) {
  __shared__ float4 part[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int quads = C / 4, qb = (quads + 63) / 64;
  const int b = blockIdx.x / qb, q = (blockIdx.x - b * qb) * 64 + lane;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  if (q < quads) {
    const float* px = x + (long)b * HW * C + 4 * q;
    for (int p = wave; p < HW; p += 32) {
      float4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const float4*>(px + (long)min(p + 4 * u, HW - 1) * C);
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (p + 4 * u < HW) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
    }
  }
  part[wave][lane] = s;
  __syncthreads();
  if (wave == 0 && q < quads) {
    const float4 a0 = part[0][lane], a1 = part[1][lane], a2 = part[2][lane], a3 = part[3][lane];
    s.x = ((a0.x + a1.x) + (a2.x + a3.x)) * inv;
    s.y = ((a0.y + a1.y) + (a2.y + a3.y)) * inv;
    s.z = ((a0.z + a1.z) + (a2.z + a3.z)) * inv;
    s.w = ((a0.w + a1.w) + (a2.w + a3.w)) * inv;
    if (err && !(fabsf(s.x) + fabsf(s.y) + fabsf(s.z) + fabsf(s.w) < __builtin_inff())) atomicOr(err, 8);
    *reinterpret_cast<float4*>(out + (long)b * C + 4 * q) = s;
  }
}

int global_avgpool(const float* x, float* out, int Bn, int HW, int C, hipStream_t stream, int* err) {
  CAPNET_REQUIRE(x && out && C % 4 == 0 && HW > 0 && Bn > 0, "global_avgpool: bad argument");
  const int qb = (C / 4 + 63) / 64;
  hipLaunchKernelGGL(avgpool_kernel, dim3(Bn * qb), dim3(256), 0, stream, x, out, Bn, HW,
                     C, 1.f / (float)HW, err);
  CAPNET_LAUNCH_CHECK();
  return kOk;
}

// ---- AdaptiveAvgPool2d(OUT) on an NHWC map whose side divides OUT (7 -> 14 is a 2x
// replication, stylenet/model_att.py:19-20,26) -------------------------------------------
__global__ __launch_bounds__(256) void upsample_nhwc_kernel(const float4* __restrict__ x,
                                                            float4* __restrict__ out, int Bn,
                                                            int S, int OUT, int C4) {
  const long total = (long)Bn * OUT * OUT * C4;
  const int f = OUT / S;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C4);
    long p = i / C4;
    const int ow = (int)(p % OUT); p /= OUT;
    const int oh = (int)(p % OUT);
    const int b = (int)(p / OUT);
    out[i] = x[(((long)b * S + oh / f) * S + ow / f) * C4 + c];
  }
}

int adaptive_pool_replicate(const float* x, float* out, int Bn, int S, int OUT, int C,
                            hipStream_t stream) {
  CAPNET_REQUIRE(x && out && C % 4 == 0 && S > 0 && OUT % S == 0,
                 "adaptive_pool_replicate: output side must be a multiple of the input side");
  const long total = (long)Bn * OUT * OUT * (C / 4);
  const int blocks = (int)(total / 256 < 1 ? 1 : (total / 256 > 8192 ? 8192 : total / 256));
  hipLaunchKernelGGL(upsample_nhwc_kernel, dim3(blocks), dim3(256), 0, stream, (const float4*)x,
                     (float4*)out, Bn, S, OUT, C / 4);
  CAPNET_LAUNCH_CHECK();
  return kOk;
}

// ---- weight packing: OIHW (torch) -> [Cout][KH][KW][Cin] padded to Kw per row ------------
__global__ void pack_weight_kernel(const float* __restrict__ w, float* __restrict__ out, int Cout,
                                   int Cin, int KH, int KW, int Kw) {
  const long total = (long)Cout * Kw;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long)gridDim.x * blockDim.x) {
    const int k = (int)(i % Kw);
    const int co = (int)(i / Kw);
    float v = 0.f;
    if (k < KH * KW * Cin) {
      const int ci = k % Cin;
      const int tap = k / Cin;
      const int s = tap % KW, r = tap / KW;
      v = w[(((long)co * Cin + ci) * KH + r) * KW + s];
    }
    out[i] = v;
  }
}

int pack_conv_weight(const float* w_oihw, float* out, int Cout, int Cin, int KH, int KW, int Kw,
                     hipStream_t stream) {
  CAPNET_REQUIRE(w_oihw && out && Kw >= KH * KW * Cin, "pack_conv_weight: bad argument");
  const long total = (long)Cout * Kw;
  const int blocks = (int)(total / 256 < 1 ? 1 : (total / 256 > 4096 ? 4096 : total / 256));
  hipLaunchKernelGGL(pack_weight_kernel, dim3(blocks), dim3(256), 0, stream, w_oihw, out, Cout,
                     Cin, KH, KW, Kw);
  CAPNET_LAUNCH_CHECK();
  return kOk;
}

// OIHW -> K-major [Kw][Cout] (k = (r*KW + s)*Cin + ci), the LDS-DMA friendly image of conv v2
__global__ void pack_weight_kmajor_kernel(const float* __restrict__ w, float* __restrict__ out,
                                          int Cout, int Cin, int KH, int KW, int Kw) {
  const long total = (long)Kw * Cout;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long)gridDim.x * blockDim.x) {
    const int co = (int)(i % Cout);
    const int k = (int)(i / Cout);
    float v = 0.f;
    if (k < KH * KW * Cin) {
      const int ci = k % Cin;
      const int tap = k / Cin;
      const int s = tap % KW, r = tap / KW;
      v = w[(((long)co * Cin + ci) * KH + r) * KW + s];
    }
    out[i] = v;
  }
}

int pack_conv_weight_kmajor(const float* w_oihw, float* out, int Cout, int Cin, int KH, int KW,
                            int Kw, hipStream_t stream) {
  CAPNET_REQUIRE(w_oihw && out && Kw >= KH * KW * Cin, "pack_conv_weight_kmajor: bad argument");
  const long total = (long)Cout * Kw;
  const int blocks = (int)(total / 256 < 1 ? 1 : (total / 256 > 4096 ? 4096 : total / 256));
  hipLaunchKernelGGL(pack_weight_kmajor_kernel, dim3(blocks), dim3(256), 0, stream, w_oihw, out,
                     Cout, Cin, KH, KW, Kw);
  CAPNET_LAUNCH_CHECK();
  return kOk;
}

}  // namespace capnet
