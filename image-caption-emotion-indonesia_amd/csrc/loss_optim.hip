// Softmax + NLL loss (fused log-softmax / gather / mean), element-wise gradient clamp + Adam,
// and the encoder head's BatchNorm1d. All HBM-bound.
//   loss        : nn.CrossEntropyLoss()(outputs, targets)       stylenet/train_multitask.py:134,383
//   clamp + Adam: utils.clip_gradient + torch.optim.Adam.step   stylenet/utils.py:51-60,
//                                                               stylenet/train_multitask.py:388-389
//   BatchNorm1d : EncoderCNN.bn (momentum 0.01)                 stylenet/model.py:20,26
#include "common.h"
#include "kernels.h"

namespace capnet {

// ---- cross entropy ---------------------------------------------------------------------
__device__ __forceinline__ float block_reduce_max(float v, float* s) {
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = v;
  __syncthreads();
  float r = s[0];
  for (int w = 1; w < (int)(blockDim.x >> 6); ++w) r = fmaxf(r, s[w]);
  __syncthreads();
  return r;
}
__device__ __forceinline__ float block_reduce_sum(float v, float* s) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = v;
  __syncthreads();
  float r = s[0];
  for (int w = 1; w < (int)(blockDim.x >> 6); ++w) r += s[w];
  __syncthreads();
  return r;
}

// one workgroup per row: lse[row] = log sum exp, row_loss[row] = lse - logit[target]
__global__ __launch_bounds__(256) void xent_fwd_kernel(const float* __restrict__ logits, long ld,
                                                       int V, const long long* __restrict__ target,
                                                       float* __restrict__ lse,
                                                       float* __restrict__ row_loss,
                                                       int* __restrict__ err_flag) {
  __shared__ float s[4];
  const int row = blockIdx.x;
  const float* p = logits + (long)row * ld;
  float m = -INFINITY;
  for (int j = threadIdx.x; j < V; j += blockDim.x) m = fmaxf(m, p[j]);
  m = block_reduce_max(m, s);
  float z = 0.f;
  for (int j = threadIdx.x; j < V; j += blockDim.x) z += expf(p[j] - m);
  z = block_reduce_sum(z, s);
  if (threadIdx.x == 0) {
    const float l = m + logf(z);
    lse[row] = l;
    const long long t = target[row];
    if (t < 0 || t >= V) {
      atomicExch(err_flag, 2);
      row_loss[row] = 0.f;
    } else {
      row_loss[row] = l - p[t];
    }
  }
}

// deterministic mean of n values by a single workgroup
__global__ __launch_bounds__(256) void mean_kernel(const float* __restrict__ x, int n,
                                                   float* __restrict__ out) {
  __shared__ float s[4];
  float a = 0.f;
  for (int i = threadIdx.x; i < n; i += blockDim.x) a += x[i];
  a = block_reduce_sum(a, s);
  if (threadIdx.x == 0) out[0] = a / (float)n;
}

int xent_fwd(const float* logits, long ld, int N, int V, const long long* targets, float* lse,
             float* row_loss, float* loss, int* err_flag, hipStream_t stream) {
  CAPNET_REQUIRE(logits && targets && lse && row_loss && loss && err_flag && N > 0 && V > 0,
                 "xent_fwd: bad argument");
  hipLaunchKernelGGL(xent_fwd_kernel, dim3(N), dim3(256), 0, stream, logits, ld, V, targets, lse,
                     row_loss, err_flag);
  hipLaunchKernelGGL(mean_kernel, dim3(1), dim3(256), 0, stream, row_loss, N, loss);
  CAPNET_LAUNCH_CHECK();
  return kOk;
}

// dlogits[row][j] = (softmax - onehot) * gout[0] / N
__global__ __launch_bounds__(256) void xent_bwd_kernel(const float* __restrict__ logits, long ld,
                                                       int V, const long long* __restrict__ target,
                                                       const float* __restrict__ lse,
                                                       const float* __restrict__ gout, float inv_n,
                                                       float* __restrict__ dlogits, long ldd) {
  const int row = blockIdx.x;
  const float* p = logits + (long)row * ld;
  float* d = dlogits + (long)row * ldd;
  const float l = lse[row];
  const float g = gout[0] * inv_n;
  const long long t = target[row];
  for (int j = threadIdx.x; j < V; j += blockDim.x) {
    float v = expf(p[j] - l);
    if (j == t) v -= 1.f;
    d[j] = v * g;
  }
}

int xent_bwd(const float* logits, long ld, int N, int V, const long long* targets,
             const float* lse, const float* gout, float* dlogits, long ldd, hipStream_t stream) {
  CAPNET_REQUIRE(logits && targets && lse && gout && dlogits && N > 0 && V > 0,
                 "xent_bwd: bad argument");
  hipLaunchKernelGGL(xent_bwd_kernel, dim3(N), dim3(256), 0, stream, logits, ld, V, targets, lse,
                     gout, 1.f / (float)N, dlogits, ldd);
  CAPNET_LAUNCH_CHECK();
  return kOk;
}

// ---- clamp + Adam, many tensors per launch ------------------------------------------------
constexpr int kAdamMaxTensors = 40;
constexpr int kAdamChunk = 2048;  // elements per workgroup (256 threads x 2 float4)

struct AdamTable {
  float* p[kAdamMaxTensors];
  float* g[kAdamMaxTensors];
  float* m[kAdamMaxTensors];
  float* v[kAdamMaxTensors];
  long n[kAdamMaxTensors];
  int chunk_start[kAdamMaxTensors + 1];
  float step_size[kAdamMaxTensors];   // lr / (1 - b1^t)
  float sqrt_bc2[kAdamMaxTensors];    // sqrt(1 - b2^t)
  int count;
};

// torch.optim.Adam (no amsgrad, no weight decay), single-tensor formulation:
//   m = lerp(m, g, 1-b1) ; v = b2*v + (1-b2)*g*g
//   p -= step_size * m / (sqrt(v) / sqrt_bc2 + eps)
// skip_flag (may be null): the device-side error word of the step (token id out of range, target out of range, a
// persistent-LSTM wait that expired). While it is non-zero the gradients are garbage: nothing is updated.
__global__ __launch_bounds__(256) void clamp_adam_kernel(AdamTable t, float b1, float b2, float eps,
                                                         float clip, int write_grad, const int* __restrict__ skip_flag) {
  if (skip_flag && *skip_flag != 0) return;
  int ti = 0;
  const int blk = blockIdx.x;
  while (ti + 1 < t.count && blk >= t.chunk_start[ti + 1]) ++ti;
  const long base = (long)(blk - t.chunk_start[ti]) * kAdamChunk;
  float* __restrict__ p = t.p[ti];
  float* __restrict__ g = t.g[ti];
  float* __restrict__ m = t.m[ti];
  float* __restrict__ v = t.v[ti];
  const long n = t.n[ti];
  const float ss = t.step_size[ti], sb = t.sqrt_bc2[ti];
  const float w1 = 1.f - b1, w2 = 1.f - b2;
  for (int k = 0; k < kAdamChunk / 256; ++k) {
    const long i = base + k * 256 + threadIdx.x;
    if (i >= n) break;
    float gi = g[i];
    if (clip > 0.f) {
      gi = fminf(fmaxf(gi, -clip), clip);
      if (write_grad) g[i] = gi;
    }
    const float m0 = m[i];
    const float mi = m0 + w1 * (gi - m0);
    const float vi = v[i] * b2 + w2 * gi * gi;
    m[i] = mi;
    v[i] = vi;
    const float denom = sqrtf(vi) / sb + eps;
    p[i] = p[i] - ss * (mi / denom);
  }
}

int clamp_adam(int n_tensors, float* const* params, float* const* grads, float* const* exp_avg,
               float* const* exp_avg_sq, const long* numel, const int* step, float lr, float b1,
               float b2, float eps, float clip, int write_grad, const int* skip_flag, hipStream_t stream) {
  CAPNET_REQUIRE(n_tensors >= 0 && (n_tensors == 0 || (params && grads && exp_avg && exp_avg_sq &&
                                                       numel && step)),
                 "clamp_adam: bad argument");
  int i = 0;
  while (i < n_tensors) {
    AdamTable t;
    t.count = 0;
    int chunks = 0;
    while (i < n_tensors && t.count < kAdamMaxTensors) {
      CAPNET_REQUIRE(step[i] >= 1 && numel[i] >= 0, "clamp_adam: tensor %d step %d numel %ld", i,
                     step[i], numel[i]);
      if (numel[i] == 0) { ++i; continue; }
      const int k = t.count++;
      t.p[k] = params[i]; t.g[k] = grads[i]; t.m[k] = exp_avg[i]; t.v[k] = exp_avg_sq[i];
      t.n[k] = numel[i];
      t.chunk_start[k] = chunks;
      chunks += cdiv(numel[i], kAdamChunk);
      const double bc1 = 1.0 - pow((double)b1, (double)step[i]);
      const double bc2 = 1.0 - pow((double)b2, (double)step[i]);
      t.step_size[k] = (float)((double)lr / bc1);
      t.sqrt_bc2[k] = (float)sqrt(bc2);
      ++i;
    }
    t.chunk_start[t.count] = chunks;
    if (t.count == 0) break;
    hipLaunchKernelGGL(clamp_adam_kernel, dim3(chunks), dim3(256), 0, stream, t, b1, b2, eps, clip,
                       write_grad, skip_flag);
  }
  CAPNET_LAUNCH_CHECK();
  return kOk;
}

// ---- flat gradient buffer: gather many tensors into one (for ONE RCCL all-reduce) and back --
struct PackTable {
  float* t[kAdamMaxTensors];
  long n[kAdamMaxTensors];
  long flat_off[kAdamMaxTensors];
  int chunk_start[kAdamMaxTensors + 1];
  int count;
};

// dir 0: flat[off + i] = t[i] ; dir 1: t[i] = flat[off + i] * scale
__global__ __launch_bounds__(256) void pack_kernel(PackTable t, float* __restrict__ flat, int dir,
                                                   float scale) {
  int ti = 0;
  const int blk = blockIdx.x;
  while (ti + 1 < t.count && blk >= t.chunk_start[ti + 1]) ++ti;
  const long base = (long)(blk - t.chunk_start[ti]) * kAdamChunk;
  float* __restrict__ x = t.t[ti];
  float* __restrict__ f = flat + t.flat_off[ti];
  const long n = t.n[ti];
  for (int k = 0; k < kAdamChunk / 256; ++k) {
    const long i = base + k * 256 + threadIdx.x;
    if (i >= n) break;
    if (dir == 0) f[i] = x[i];
    else x[i] = f[i] * scale;
  }
}

int pack_tensors(int n_tensors, float* const* tensors, const long* numel, float* flat, int dir,
                 float scale, hipStream_t stream) {
  CAPNET_REQUIRE(n_tensors >= 0 && (n_tensors == 0 || (tensors && numel && flat)),
                 "pack_tensors: bad argument");
  int i = 0;
  long off = 0;
  while (i < n_tensors) {
    PackTable t;
    t.count = 0;
    int chunks = 0;
    while (i < n_tensors && t.count < kAdamMaxTensors) {
      CAPNET_REQUIRE(numel[i] >= 0 && (numel[i] == 0 || tensors[i]), "pack_tensors: tensor %d", i);
      if (numel[i] == 0) { ++i; continue; }
      const int k = t.count++;
      t.t[k] = tensors[i]; t.n[k] = numel[i]; t.flat_off[k] = off;
      t.chunk_start[k] = chunks;
      chunks += cdiv(numel[i], kAdamChunk);
      off += numel[i];
      ++i;
    }
    t.chunk_start[t.count] = chunks;
    if (t.count == 0) break;
    hipLaunchKernelGGL(pack_kernel, dim3(chunks), dim3(256), 0, stream, t, flat, dir, scale);
  }
  CAPNET_LAUNCH_CHECK();
  return kOk;
}

// ---- the device error word across ranks and across dropped steps ------------------------------------
// clamp_adam skips its update while the LOCAL error word is set; with several ranks that decision has to be the same
// everywhere or the replicas' parameters drift apart. The word rides in one extra float behind the flat gradient buffer
// that is all-reduced anyway: dir 0 writes (word != 0) into the slot, dir 1 (after the SUM) raises bit 4 (16, "another
// rank dropped this step") locally when any rank had a fault.
__global__ void err_word_exchange_kernel(int* __restrict__ err_flag, float* __restrict__ slot, int dir) {
  if (dir == 0) *slot = (*err_flag != 0) ? 1.f : 0.f;
  else if (*slot > 0.f && *err_flag == 0) *err_flag = 16;
}

int err_word_exchange(int* err_flag, float* slot, int dir, hipStream_t stream) {
  CAPNET_REQUIRE(err_flag && slot && (dir == 0 || dir == 1), "err_word_exchange: bad argument");
  hipLaunchKernelGGL(err_word_exchange_kernel, dim3(1), dim3(1), 0, stream, err_flag, slot, dir);
  CAPNET_LAUNCH_CHECK();
  return kOk;
}

// counter += 1 while the error word is set: launched once per optimizer step in front of its clamp_adam launches, so that
// the host can take the dropped steps out of its per-parameter step counts (Adam's bias correction) when it learns of them.
__global__ void count_skipped_kernel(const int* __restrict__ err_flag, int* __restrict__ counter) {
  if (*err_flag != 0) *counter += 1;
}

int count_skipped(const int* err_flag, int* counter, hipStream_t stream) {
  CAPNET_REQUIRE(err_flag && counter, "count_skipped: bad argument");
  hipLaunchKernelGGL(count_skipped_kernel, dim3(1), dim3(1), 0, stream, err_flag, counter);
  CAPNET_LAUNCH_CHECK();
  return kOk;
}

// ---- top-k accuracy: utils.accuracy (stylenet/utils.py:127-140) --------------------------------
// count += #rows whose target is among the k largest logits. Rank of the target = number of
// entries that beat it (strictly larger, or equal with a lower index: the order torch.topk
// returns on ties is unspecified, lower index first is the CPU behaviour). One workgroup per row.
__global__ __launch_bounds__(256) void topk_correct_kernel(const float* __restrict__ logits, long ld,
                                                           int V, const long long* __restrict__ targets,
                                                           int k, int* __restrict__ count,
                                                           int* __restrict__ err_flag) {
  const int row = blockIdx.x;
  const long long tg = targets[row];
  if (tg < 0 || tg >= V) {
    if (threadIdx.x == 0) atomicMax(err_flag, 2);
    return;
  }
  const float* p = logits + (long)row * ld;
  const float tv = p[tg];
  int beat = 0;
  for (int v = threadIdx.x; v < V; v += 256) {
    const float x = p[v];
    beat += (x > tv || (x == tv && v < tg)) ? 1 : 0;
  }
  for (int o = 32; o > 0; o >>= 1) beat += __shfl_xor(beat, o);
  __shared__ int s_b[4];
  if ((threadIdx.x & 63) == 0) s_b[threadIdx.x >> 6] = beat;
  __syncthreads();
  if (threadIdx.x == 0 && s_b[0] + s_b[1] + s_b[2] + s_b[3] < k) atomicAdd(count, 1);
}

int topk_correct(const float* logits, long ld, int N, int V, const long long* targets, int k,
                 int* count, int* err_flag, hipStream_t stream) {
  CAPNET_REQUIRE(logits && targets && count && err_flag && V > 0 && ld >= V && k > 0 && N >= 0,
                 "topk_correct: bad argument");
  CAPNET_HIP_CHECK(hipMemsetAsync(count, 0, sizeof(int), stream));
  if (N == 0) return kOk;
  hipLaunchKernelGGL(topk_correct_kernel, dim3(N), dim3(256), 0, stream, logits, ld, V, targets, k,
                     count, err_flag);
  CAPNET_LAUNCH_CHECK();
  return kOk;
}

// ---- stand-alone element-wise clamp (utils.clip_gradient with a foreign optimiser) -----------
__global__ __launch_bounds__(256) void clamp_kernel(float* __restrict__ x, long n, float lo,
                                                    float hi) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (long)gridDim.x * blockDim.x)
    x[i] = fminf(fmaxf(x[i], lo), hi);
}

int clamp_inplace(float* x, long n, float lo, float hi, hipStream_t stream) {
  CAPNET_REQUIRE(x && n >= 0 && lo <= hi, "clamp_inplace: bad argument");
  if (n == 0) return kOk;
  const long blocks = cdiv(n, 256);
  hipLaunchKernelGGL(clamp_kernel, dim3((int)(blocks > 4096 ? 4096 : blocks)), dim3(256), 0, stream,
                     x, n, lo, hi);
  CAPNET_LAUNCH_CHECK();
  return kOk;
}

// ---- BatchNorm1d over [B][C] (encoder head) -----------------------------------------------
__global__ __launch_bounds__(64) void bn1d_fwd_kernel(const float* __restrict__ x, int B, int C,
                                                      const float* __restrict__ gamma,
                                                      const float* __restrict__ beta,
                                                      float* __restrict__ rmean,
                                                      float* __restrict__ rvar, int train,
                                                      float momentum, float eps,
                                                      float* __restrict__ y,
                                                      float* __restrict__ save_mean,
                                                      float* __restrict__ save_invstd) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  float mean, invstd;
  if (train) {
    double s = 0.0;
    for (int r = 0; r < B; ++r) s += (double)x[(long)r * C + c];
    const double mu = s / B;
    double q = 0.0;
    for (int r = 0; r < B; ++r) {
      const double d = (double)x[(long)r * C + c] - mu;
      q += d * d;
    }
    const double var = q / B;
    mean = (float)mu;
    invstd = (float)(1.0 / sqrt(var + (double)eps));
    if (rmean) {
      const double unb = B > 1 ? q / (B - 1) : var;
      rmean[c] = (1.f - momentum) * rmean[c] + momentum * mean;
      rvar[c] = (1.f - momentum) * rvar[c] + momentum * (float)unb;
    }
  } else {
    mean = rmean[c];
    invstd = 1.f / sqrtf(rvar[c] + eps);
  }
  if (save_mean) { save_mean[c] = mean; save_invstd[c] = invstd; }
  const float g = gamma[c], b = beta[c];
  for (int r = 0; r < B; ++r) y[(long)r * C + c] = (x[(long)r * C + c] - mean) * invstd * g + b;
}

int bn1d_fwd(const float* x, int B, int C, const float* gamma, const float* beta, float* rmean,
             float* rvar, int train, float momentum, float eps, float* y, float* save_mean,
             float* save_invstd, hipStream_t stream) {
  CAPNET_REQUIRE(x && y && gamma && beta && B > 0 && C > 0, "bn1d_fwd: bad argument");
  CAPNET_REQUIRE(train || (rmean && rvar), "bn1d_fwd: eval mode needs running stats");
  hipLaunchKernelGGL(bn1d_fwd_kernel, dim3(cdiv(C, 64)), dim3(64), 0, stream, x, B, C, gamma, beta,
                     rmean, rvar, train, momentum, eps, y, save_mean, save_invstd);
  CAPNET_LAUNCH_CHECK();
  return kOk;
}

// train-mode backward: dx = g*invstd/B * (B*dy - sum(dy) - xhat*sum(dy*xhat))
__global__ __launch_bounds__(64) void bn1d_bwd_kernel(const float* __restrict__ dy,
                                                      const float* __restrict__ x, int B, int C,
                                                      const float* __restrict__ gamma,
                                                      const float* __restrict__ save_mean,
                                                      const float* __restrict__ save_invstd,
                                                      float* __restrict__ dx,
                                                      float* __restrict__ dgamma,
                                                      float* __restrict__ dbeta) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float mean = save_mean[c], invstd = save_invstd[c];
  float s1 = 0.f, s2 = 0.f;
  for (int r = 0; r < B; ++r) {
    const float d = dy[(long)r * C + c];
    const float xh = (x[(long)r * C + c] - mean) * invstd;
    s1 += d;
    s2 += d * xh;
  }
  dgamma[c] = s2;
  dbeta[c] = s1;
  const float k = gamma[c] * invstd / (float)B;
  for (int r = 0; r < B; ++r) {
    const float d = dy[(long)r * C + c];
    const float xh = (x[(long)r * C + c] - mean) * invstd;
    dx[(long)r * C + c] = k * ((float)B * d - s1 - xh * s2);
  }
}

int bn1d_bwd(const float* dy, const float* x, int B, int C, const float* gamma,
             const float* save_mean, const float* save_invstd, float* dx, float* dgamma,
             float* dbeta, hipStream_t stream) {
  CAPNET_REQUIRE(dy && x && gamma && save_mean && save_invstd && dx && dgamma && dbeta && B > 0 &&
                     C > 0,
                 "bn1d_bwd: bad argument");
  hipLaunchKernelGGL(bn1d_bwd_kernel, dim3(cdiv(C, 64)), dim3(64), 0, stream, dy, x, B, C, gamma,
                     save_mean, save_invstd, dx, dgamma, dbeta);
  CAPNET_LAUNCH_CHECK();
  return kOk;
}

// ---- attention loss: nll + alpha_c * ((1 - sum_t alphas[b][t][p])^2).mean() -------------------------
// (stylenet/train_multitask_att.py:409-411). Two launches: B*P sums over the steps (fixed order, one thread each,
// any number of workgroups; `colsum` [B*P] keeps sum_t alpha for the backward kernel), then one workgroup
// reduces (1 - colsum)^2 in a fixed order and writes the scalar total. (One workgroup doing both took 390 us
// at B = 64 on the critical path of every attention step.)
__global__ __launch_bounds__(256) void att_loss_colsum_kernel(const float* __restrict__ alphas, int B, int steps,
                                                              int P, float* __restrict__ colsum) {
  const int n = B * P;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const int b = i / P, p = i - b * P;
    const float* a = alphas + (long)b * steps * P + p;
    float s = 0.f;
    for (int t = 0; t < steps; ++t) s += a[(long)t * P];
    colsum[i] = s;
  }
}
__global__ __launch_bounds__(256) void att_loss_fwd_kernel(const float* __restrict__ nll,
                                                           const float* __restrict__ colsum, int n,
                                                           float alpha_c, float* __restrict__ out) {
  __shared__ float red[256];
  float part = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) {
    const float d = 1.f - colsum[i];
    part = fmaf(d, d, part);
  }
  red[threadIdx.x] = part;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = nll[0] + alpha_c * (red[0] / (float)n);
}

// dalphas[b][t][p] = gout * alpha_c * 2 (sum_t alpha - 1) / (B P), the same for every step t
__global__ __launch_bounds__(256) void att_loss_bwd_kernel(const float* __restrict__ gout,
                                                           const float* __restrict__ colsum, int B,
                                                           int steps, int P, float alpha_c,
                                                           float* __restrict__ dalphas) {
  const long total = (long)B * steps * P;
  const float k = gout[0] * alpha_c * 2.f / (float)(B * P);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long)gridDim.x * blockDim.x) {
    const int p = (int)(i % P);
    const int b = (int)(i / ((long)steps * P));
    dalphas[i] = k * (colsum[b * P + p] - 1.f);
  }
}

int att_loss_fwd(const float* nll, const float* alphas, int B, int steps, int P, float alpha_c,
                 float* colsum, float* out, hipStream_t stream) {
  CAPNET_REQUIRE(nll && alphas && colsum && out && B > 0 && steps > 0 && P > 0, "att_loss_fwd: bad argument");
  const int n = B * P;
  hipLaunchKernelGGL(att_loss_colsum_kernel, dim3(cdiv(n, 256) > 512 ? 512 : cdiv(n, 256)), dim3(256), 0, stream, alphas, B,
                     steps, P, colsum);
  hipLaunchKernelGGL(att_loss_fwd_kernel, dim3(1), dim3(256), 0, stream, nll, colsum, n, alpha_c, out);
  CAPNET_LAUNCH_CHECK();
  return kOk;
}

int att_loss_bwd(const float* gout, const float* colsum, int B, int steps, int P, float alpha_c,
                 float* dalphas, hipStream_t stream) {
  CAPNET_REQUIRE(gout && colsum && dalphas && B > 0 && steps > 0 && P > 0, "att_loss_bwd: bad argument");
  const long total = (long)B * steps * P;
  hipLaunchKernelGGL(att_loss_bwd_kernel, dim3(cdiv(total, 256) > 1024 ? 1024 : cdiv(total, 256)), dim3(256), 0,
                     stream, gout, colsum, B, steps, P, alpha_c, dalphas);
  CAPNET_LAUNCH_CHECK();
  return kOk;
}

}  // namespace capnet
