// Memory-bound kernels of the caption decoders (FactoredLSTM / LSTMCell recurrences):
// embedding gather + dropout, gate pointwise forward/backward, row argmax, row gather,
// column sums (bias gradients) and the embedding-gradient scatter.
// Packed ("time-major") row order is torch's pack_padded_sequence order used by
// stylenet/model.py:173-194: rows of step t are contiguous, sample order preserved.
#include "common.h"
#include "kernels.h"

namespace capnet {

// counter-based dropout mask: keep iff u(seed, sample, col, e) >= p. Recomputed in backward.
__device__ __forceinline__ float dropout_scale(unsigned long long seed, int sample, int col, int e,
                                               float p, float inv_keep) {
  unsigned long long z = seed + 0x9E3779B97F4A7C15ull *
                                    ((((unsigned long long)(unsigned)sample << 20) ^
                                      ((unsigned long long)(unsigned)col << 10)) *
                                         1000003ull +
                                     (unsigned long long)(unsigned)e + 1ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z = z ^ (z >> 31);
  const float u = (float)(z >> 40) * (1.0f / 16777216.0f);
  return u >= p ? inv_keep : 0.f;
}

// ---- row bookkeeping, built on device from kernel arguments (no H2D copy, no sync) -------
// row r of step t (rows of a step are contiguous, sample j at offset j):
//   row_sample = j, prev_row = row of the same sample at step t-1 (or -1), row_token = -1,
//   row_col: >= 0  caption column, dropout applies        (teacher forced, model.py:182)
//            -1    image feature                           (step 0 with features, model.py:171)
//            -2    token predicted from h_{t-1}, no dropout (model.py:184,190-191)
//            <= -3 caption column (-3 - col), no dropout   (t = 0 free-running: captions[:,0])
__global__ __launch_bounds__(256) void build_rows_kernel(SeqMeta m, int* __restrict__ row_sample,
                                                         int* __restrict__ row_col,
                                                         int* __restrict__ row_token,
                                                         int* __restrict__ prev_row) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= m.N) return;
  int t = 0;
  while (t + 1 < m.steps && r >= m.off[t + 1]) ++t;
  const int j = r - m.off[t];
  row_sample[r] = j;
  row_token[r] = -1;
  prev_row[r] = t > 0 ? m.off[t - 1] + j : -1;
  int col;
  if (m.tf[t]) col = m.has_features ? (t == 0 ? -1 : t - 1) : t;
  else col = (t == 0) ? -3 : -2;
  row_col[r] = col;
}

int build_rows(const SeqMeta& m, int* row_sample, int* row_col, int* row_token, int* prev_row,
               hipStream_t stream) {
  hipLaunchKernelGGL(build_rows_kernel, dim3(cdiv(m.N, 256)), dim3(256), 0, stream, m, row_sample,
                     row_col, row_token, prev_row);
  CAPNET_LAUNCH_CHECK();
  return kOk;
}

// ---- build decoder inputs -------------------------------------------------------------
// X[r] = feature[sample] or emb[token] (* dropout mask). dynamic = 0: every row whose input is
// known before the recurrence starts; dynamic = 1: rows whose token was just predicted.
__global__ __launch_bounds__(128) void gather_inputs_kernel(
    const long long* __restrict__ captions, int T, const float* __restrict__ features,
    const float* __restrict__ emb, int E, int V, const int* __restrict__ row_sample,
    const int* __restrict__ row_col, int* __restrict__ row_token, float* __restrict__ X, long ldx,
    int r0, int r1, float p, unsigned long long seed, int use_dropout, int dynamic,
    int* __restrict__ err_flag) {
  const int r = r0 + blockIdx.x;
  if (r >= r1) return;
  const int sample = row_sample[r];
  const int col = row_col[r];
  if ((col == -2) != (dynamic != 0)) return;
  const float* src;
  bool drop = false;
  if (col == -1) {
    src = features + (long)sample * E;
  } else {
    int tok;
    if (col == -2) {
      tok = row_token[r];
    } else {
      const int cc = col >= 0 ? col : -3 - col;
      tok = (int)captions[(long)sample * T + cc];
      if (threadIdx.x == 0) row_token[r] = tok;
      drop = (col >= 0) && use_dropout != 0;
    }
    if (tok < 0 || tok >= V) {  // out-of-range token id: flag it, read row 0 (never fault)
      if (threadIdx.x == 0) atomicExch(err_flag, 1);
      tok = 0;
    }
    src = emb + (long)tok * E;
  }
  const float inv_keep = p < 1.f ? 1.f / (1.f - p) : 0.f;
  for (int e = threadIdx.x; e < E; e += blockDim.x) {
    float v = src[e];
    if (drop) v *= dropout_scale(seed, sample, col, e, p, inv_keep);
    X[(long)r * ldx + e] = v;
  }
}

int gather_inputs(const long long* captions, int T, const float* features, const float* emb, int E,
                  int V, const int* row_sample, const int* row_col, int* row_token, float* X,
                  long ldx, int r0, int r1, float p, unsigned long long seed, int use_dropout,
                  int dynamic, int* err_flag, hipStream_t stream) {
  if (r1 <= r0) return kOk;
  hipLaunchKernelGGL(gather_inputs_kernel, dim3(r1 - r0), dim3(128), 0, stream, captions, T,
                     features, emb, E, V, row_sample, row_col, row_token, X, ldx, r0, r1, p, seed,
                     use_dropout, dynamic, err_flag);
  CAPNET_LAUNCH_CHECK();
  return kOk;
}

// ---- plain embedding lookup (sample()/forward_step path): out[r] = emb[idx[r]] -------------
__global__ __launch_bounds__(128) void embedding_fwd_kernel(const long long* __restrict__ idx,
                                                            const float* __restrict__ emb, int E,
                                                            int V, float* __restrict__ out,
                                                            int* __restrict__ err_flag) {
  const int r = blockIdx.x;
  long long t = idx[r];
  if (t < 0 || t >= V) {
    if (threadIdx.x == 0) atomicExch(err_flag, 1);
    t = 0;
  }
  for (int e = threadIdx.x; e < E; e += blockDim.x) out[(long)r * E + e] = emb[t * E + e];
}

int embedding_fwd(const long long* idx, int n, const float* emb, int E, int V, float* out,
                  int* err_flag, hipStream_t stream) {
  CAPNET_REQUIRE(idx && emb && out && err_flag && E > 0 && V > 0, "embedding_fwd: bad argument");
  if (n <= 0) return kOk;
  hipLaunchKernelGGL(embedding_fwd_kernel, dim3(n), dim3(128), 0, stream, idx, emb, E, V, out,
                     err_flag);
  CAPNET_LAUNCH_CHECK();
  return kOk;
}

// ---- pack_padded_sequence(captions, lengths, batch_first=True)[0] for int64 captions ---------
__global__ __launch_bounds__(256) void packed_targets_kernel(SeqMeta m,
                                                             const long long* __restrict__ captions,
                                                             int T, long long* __restrict__ out) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= m.N) return;
  int t = 0;
  while (t + 1 < m.steps && r >= m.off[t + 1]) ++t;
  out[r] = captions[(long)(r - m.off[t]) * T + t];
}

int packed_targets(const SeqMeta& m, const long long* captions, int T, long long* out,
                   hipStream_t stream) {
  CAPNET_REQUIRE(captions && out && m.steps <= T, "packed_targets: bad argument");
  hipLaunchKernelGGL(packed_targets_kernel, dim3(cdiv(m.N, 256)), dim3(256), 0, stream, m, captions,
                     T, out);
  CAPNET_LAUNCH_CHECK();
  return kOk;
}

// out = a + b
__global__ void vec_add_kernel(const float* __restrict__ a, const float* __restrict__ b,
                               float* __restrict__ out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = a[i] + b[i];
}

int vec_add(const float* a, const float* b, float* out, int n, hipStream_t stream) {
  CAPNET_REQUIRE(a && b && out && n > 0, "vec_add: bad argument");
  hipLaunchKernelGGL(vec_add_kernel, dim3(cdiv(n, 256)), dim3(256), 0, stream, a, b, out, n);
  CAPNET_LAUNCH_CHECK();
  return kOk;
}

// ---- many small device-to-device copies / sums in ONE launch --------------------------------
// dst[i] = src[i] (+ src2[i]): the per-forward weight packing of the attention decoder was 28 hipMemcpyAsync + 4
// vec_add on the decoder's serial chain.
__global__ __launch_bounds__(256) void multi_copy_kernel(const CopyTable t) {
  const int it = blockIdx.y;
  const float* __restrict__ a = t.src[it];
  const float* __restrict__ b = t.src2[it];
  float* __restrict__ d = t.dst[it];
  const size_t n = t.n[it];
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  const bool v4 = (n % 4 == 0) && ((((size_t)a | (size_t)d | (size_t)b) & 15) == 0);
  if (v4) {
    const float4* a4 = reinterpret_cast<const float4*>(a);
    const float4* b4 = reinterpret_cast<const float4*>(b);
    float4* d4 = reinterpret_cast<float4*>(d);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n / 4; i += stride) {
      float4 v = a4[i];
      if (b) { const float4 w = b4[i]; v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w; }
      d4[i] = v;
    }
  } else {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) d[i] = b ? a[i] + b[i] : a[i];
  }
}

int multi_copy(const CopyTable& t, hipStream_t stream) {
  CAPNET_REQUIRE(t.count > 0 && t.count <= CopyTable::kMax, "multi_copy: %d items", t.count);
  size_t nmax = 0;
  for (int i = 0; i < t.count; ++i) {
    CAPNET_REQUIRE(t.src[i] && t.dst[i] && t.n[i] > 0, "multi_copy: item %d", i);
    nmax = t.n[i] > nmax ? t.n[i] : nmax;
  }
  const size_t want = (nmax / 4 + 255) / 256;
  const int bx = (int)(want < 1 ? 1 : (want > 256 ? 256 : want));
  hipLaunchKernelGGL(multi_copy_kernel, dim3(bx, t.count), dim3(256), 0, stream, t);
  CAPNET_LAUNCH_CHECK();
  return kOk;
}

// ---- LSTM gate pointwise ---------------------------------------------------------------
// pre: [b][4H] pre-activations, column block gi/gf/go/gg selects the gate. In place:
// pre is overwritten with the ACTIVATED gates (saved for backward).
// tanh_out = 0: h = o*c            (FactoredLSTM, stylenet/model.py:152-153)
// tanh_out = 1: h = o*tanh(c)      (nn.LSTMCell, nic/model.py:77)
__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

__global__ __launch_bounds__(256) void lstm_pointwise_fwd_kernel(
    float* __restrict__ pre, long ldp, const float* __restrict__ c_prev, float* __restrict__ c_out,
    float* __restrict__ h_out, int b, int H, int gi, int gf, int go, int gg, int tanh_out,
    const float* __restrict__ slabs, int n_slabs) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= b * H) return;
  const int row = idx / H, j = idx - row * H;
  float* p = pre + (long)row * ldp;
  float x[4] = {p[gi * H + j], p[gf * H + j], p[go * H + j], p[gg * H + j]};
  if (n_slabs > 0) {
    // + the K-chunk partials of the input product (sgemm_rows16_slabs: [k][b][4H]), in slab order; all in flight at once
    const int col[4] = {gi * H + j, gf * H + j, go * H + j, gg * H + j};
    const float* sp = slabs + (long)row * 4 * H;
    const long stride = (long)b * 4 * H;
    for (int k = 0; k < n_slabs; k += 4) {
      float v[4][4];
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int q = 0; q < 4; ++q) v[u][q] = sp[(long)min(k + u, n_slabs - 1) * stride + col[q]];
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (k + u < n_slabs) {
#pragma unroll
          for (int q = 0; q < 4; ++q) x[q] += v[u][q];
        }
    }
  }
  const float i = sigmoidf_(x[0]);
  const float f = sigmoidf_(x[1]);
  const float o = sigmoidf_(x[2]);
  const float g = tanhf(x[3]);
  const float cp = c_prev ? c_prev[(long)row * H + j] : 0.f;
  const float c = f * cp + i * g;
  p[gi * H + j] = i;
  p[gf * H + j] = f;
  p[go * H + j] = o;
  p[gg * H + j] = g;
  c_out[(long)row * H + j] = c;
  h_out[(long)row * H + j] = tanh_out ? o * tanhf(c) : o * c;
}

int lstm_pointwise_fwd(float* pre, long ldp, const float* c_prev, float* c_out, float* h_out, int b,
                       int H, int gi, int gf, int go, int gg, int tanh_out, hipStream_t stream,
                       const float* slabs, int n_slabs) {
  if (b <= 0) return kOk;
  hipLaunchKernelGGL(lstm_pointwise_fwd_kernel, dim3(cdiv((long)b * H, 256)), dim3(256), 0, stream,
                     pre, ldp, c_prev, c_out, h_out, b, H, gi, gf, go, gg, tanh_out, slabs, slabs ? n_slabs : 0);
  CAPNET_LAUNCH_CHECK();
  return kOk;
}

// gates: activated [b][4H]; dh = dH_out[row] + dh_rec[row] (dh_rec rows < b_next only);
// dc_io: in = dL/dc_t from step t+1 (rows < b_next, else 0), out = dL/dc_{t-1}.
// dpre: [b][4H] gradient wrt pre-activations.
__global__ __launch_bounds__(256) void lstm_pointwise_bwd_kernel(
    const float* __restrict__ gates, long ldg, const float* __restrict__ c,
    const float* __restrict__ c_prev, const float* __restrict__ dH,
    const float* __restrict__ dh_rec, float* __restrict__ dc_io, float* __restrict__ dpre, long ldq,
    int b, int b_next, int H, int gi, int gf, int go, int gg, int tanh_out, int dh_slabs,
    long dh_slab_stride) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= b * H) return;
  const int row = idx / H, j = idx - row * H;
  const float* p = gates + (long)row * ldg;
  const float i = p[gi * H + j], f = p[gf * H + j], o = p[go * H + j], g = p[gg * H + j];
  const float ct = c[(long)row * H + j];
  const float cp = c_prev ? c_prev[(long)row * H + j] : 0.f;
  float dh = dH[(long)row * H + j];
  float dc = 0.f;
  if (row < b_next) {
    if (dh_slabs > 0) {
      // dh_rec = sum of the K-chunk slabs of dG_{t+1} . W (sgemm_splitk_slabs), in slab order
      const float* sp = dh_rec + (long)row * H + j;
      float s = 0.f;
      int k = 0;
      for (; k + 8 <= dh_slabs; k += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = sp[(long)(k + u) * dh_slab_stride];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += v[u];
      }
      for (; k < dh_slabs; ++k) s += sp[(long)k * dh_slab_stride];
      dh += s;
    } else if (dh_rec) {
      dh += dh_rec[(long)row * H + j];
    }
    dc = dc_io[(long)row * H + j];
  }
  float d_o;
  if (tanh_out) {
    const float tc = tanhf(ct);
    d_o = dh * tc;
    dc += dh * o * (1.f - tc * tc);
  } else {
    d_o = dh * ct;
    dc += dh * o;
  }
  const float di = dc * g, dg = dc * i, df = dc * cp;
  float* q = dpre + (long)row * ldq;
  q[gi * H + j] = di * i * (1.f - i);
  q[gf * H + j] = df * f * (1.f - f);
  q[go * H + j] = d_o * o * (1.f - o);
  q[gg * H + j] = dg * (1.f - g * g);
  dc_io[(long)row * H + j] = dc * f;
}

int lstm_pointwise_bwd(const float* gates, long ldg, const float* c, const float* c_prev,
                       const float* dH, const float* dh_rec, float* dc_io, float* dpre, long ldq,
                       int b, int b_next, int H, int gi, int gf, int go, int gg, int tanh_out,
                       hipStream_t stream, int dh_slabs, long dh_slab_stride) {
  if (b <= 0) return kOk;
  hipLaunchKernelGGL(lstm_pointwise_bwd_kernel, dim3(cdiv((long)b * H, 256)), dim3(256), 0, stream,
                     gates, ldg, c, c_prev, dH, dh_rec, dc_io, dpre, ldq, b, b_next, H, gi, gf, go,
                     gg, tanh_out, dh_slabs, dh_slab_stride);
  CAPNET_LAUNCH_CHECK();
  return kOk;
}

// ---- row argmax (first maximum, like torch.max(1) on CPU) --------------------------------
// (16-B loads, eight in flight per thread, when the row allows it; a thread meets its indices in increasing order, so a
//  strict comparison keeps the first maximum, and ties between threads go to the lower index)
__global__ __launch_bounds__(256) void argmax_rows_kernel(const float* __restrict__ x, int ld,
                                                          int V, int* __restrict__ out) {
  const int row = blockIdx.x;
  const float* p = x + (long)row * ld;
  float best = -INFINITY;
  int bi = 0x7fffffff;
  int done = 0;
  if (ld % 4 == 0 && (reinterpret_cast<size_t>(x) & 15) == 0) {
    const int V4 = V / 4;
    for (int j0 = threadIdx.x; j0 < V4; j0 += 8 * 256) {
      float4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const float4*>(p + 4 * (long)min(j0 + 256 * u, V4 - 1));
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int j = 4 * (j0 + 256 * u);
        if (j0 + 256 * u < V4) {
          if (v[u].x > best) { best = v[u].x; bi = j; }
          if (v[u].y > best) { best = v[u].y; bi = j + 1; }
          if (v[u].z > best) { best = v[u].z; bi = j + 2; }
          if (v[u].w > best) { best = v[u].w; bi = j + 3; }
        }
      }
    }
    done = 4 * V4;
  }
  for (int j = done + threadIdx.x; j < V; j += blockDim.x) {
    const float v = p[j];
    if (v > best || (v == best && j < bi)) { best = v; bi = j; }
  }
  // wave: six exchange rounds; then the four waves' winners through LDS
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float v = __shfl_xor(best, o);
    const int i2 = __shfl_xor(bi, o);
    if (v > best || (v == best && i2 < bi)) { best = v; bi = i2; }
  }
  __shared__ float s_v[4];
  __shared__ int s_i[4];
  if ((threadIdx.x & 63) == 0) { s_v[threadIdx.x >> 6] = best; s_i[threadIdx.x >> 6] = bi; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 4; ++w)
      if (s_v[w] > best || (s_v[w] == best && s_i[w] < bi)) { best = s_v[w]; bi = s_i[w]; }
    out[row] = bi == 0x7fffffff ? 0 : bi;
  }
}

int argmax_rows(const float* x, int rows, int ld, int V, int* out, hipStream_t stream) {
  if (rows <= 0) return kOk;
  hipLaunchKernelGGL(argmax_rows_kernel, dim3(rows), dim3(256), 0, stream, x, ld, V, out);
  CAPNET_LAUNCH_CHECK();
  return kOk;
}

// ---- beam-search expansion: log-softmax + running score + top-k over rows*V ------------------
// One launch per decode step of sample() (stylenet/model.py:233-249): scores[r][v] =
// prev[r] + log_softmax(logits[r])[v]; the k best of the flattened [rows*V] array, best first,
// ties to the lower flat index. One workgroup: rows <= 16 beams x V is a few hundred KB.
constexpr int kBeamThreads = 1024, kBeamMax = 16;

__device__ __forceinline__ void beam_better(float v, long i, float& bv, long& bi) {
  if (v > bv || (v == bv && i < bi)) { bv = v; bi = i; }
}

// the expansion of ONE image's beams (a whole workgroup)
__device__ __forceinline__ void beam_topk_body(const float* __restrict__ logits, long ld, int rows, int V,
                                               const float* __restrict__ prev, int k, float* __restrict__ out_scores,
                                               long long* __restrict__ out_index) {
  __shared__ float s_red[kBeamThreads / 64];
  __shared__ long s_idx[kBeamThreads / 64];
  __shared__ float s_lse[kBeamMax];
  __shared__ long s_sel[kBeamMax];
  __shared__ float s_bcast;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int r = 0; r < rows; ++r) {
    const float* p = logits + (long)r * ld;
    float m = -INFINITY;
    for (int v = tid; v < V; v += kBeamThreads) m = fmaxf(m, p[v]);
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if (lane == 0) s_red[wave] = m;
    __syncthreads();
    if (tid == 0) {
      float mm = s_red[0];
      for (int w = 1; w < kBeamThreads / 64; ++w) mm = fmaxf(mm, s_red[w]);
      s_bcast = mm;
    }
    __syncthreads();
    m = s_bcast;
    float sum = 0.f;
    for (int v = tid; v < V; v += kBeamThreads) sum += expf(p[v] - m);
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    __syncthreads();
    if (lane == 0) s_red[wave] = sum;
    __syncthreads();
    if (tid == 0) {
      float t = 0.f;
      for (int w = 0; w < kBeamThreads / 64; ++w) t += s_red[w];
      s_lse[r] = m + logf(t);
    }
    __syncthreads();
  }
  for (int j = 0; j < k; ++j) {
    float bv = -INFINITY;
    long bi = 0x7fffffffffffffffL;
    for (int r = 0; r < rows; ++r) {
      const float* p = logits + (long)r * ld;
      const float base = prev[r] - s_lse[r];
      for (int v = tid; v < V; v += kBeamThreads) {
        const long flat = (long)r * V + v;
        bool taken = false;
        for (int q = 0; q < j; ++q) taken |= s_sel[q] == flat;
        if (!taken) beam_better(base + p[v], flat, bv, bi);
      }
    }
    for (int o = 32; o > 0; o >>= 1) {
      const float ov = __shfl_xor(bv, o);
      const long oi = __shfl_xor(bi, o);
      beam_better(ov, oi, bv, bi);
    }
    __syncthreads();
    if (lane == 0) { s_red[wave] = bv; s_idx[wave] = bi; }
    __syncthreads();
    if (tid == 0) {
      float v = s_red[0];
      long i = s_idx[0];
      for (int w = 1; w < kBeamThreads / 64; ++w) beam_better(s_red[w], s_idx[w], v, i);
      s_sel[j] = i;
      out_scores[j] = v;
      out_index[j] = i;
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(kBeamThreads) void beam_topk_kernel(
    const float* __restrict__ logits, long ld, int rows, int V, const float* __restrict__ prev,
    int k, float* __restrict__ out_scores, long long* __restrict__ out_index) {
  beam_topk_body(logits, ld, rows, V, prev, k, out_scores, out_index);
}

// many images at once (the test-set evaluator, stylenet/evaluator.py:63-120, decodes a whole batch per step): workgroup
// i expands image i's beams -- rows meta[3 i] .. + meta[3 i + 1] of the logits, its meta[3 i + 2] best; flat indices are
// local to the image (row within the image * V + word); images with k = 0 (finished) are skipped
__global__ __launch_bounds__(kBeamThreads) void beam_topk_batched_kernel(
    const float* __restrict__ logits, long ld, int V, const float* __restrict__ prev, const int* __restrict__ meta,
    float* __restrict__ out_scores, long long* __restrict__ out_index) {
  const int row0 = meta[3 * blockIdx.x], rows = meta[3 * blockIdx.x + 1], k = meta[3 * blockIdx.x + 2];
  if (k <= 0 || rows <= 0) return;
  beam_topk_body(logits + (long)row0 * ld, ld, rows, V, prev + row0, k, out_scores + (long)blockIdx.x * kBeamMax,
                 out_index + (long)blockIdx.x * kBeamMax);
}

int beam_topk(const float* logits, long ld, int rows, int V, const float* prev, int k,
              float* out_scores, long long* out_index, hipStream_t stream) {
  CAPNET_REQUIRE(logits && prev && out_scores && out_index, "beam_topk: null argument");
  CAPNET_REQUIRE(rows >= 1 && rows <= kBeamMax && k >= 1 && k <= kBeamMax && V >= 1 && ld >= V &&
                     (long)rows * V >= k,
                 "beam_topk: rows=%d k=%d V=%d (rows, k <= %d; k <= rows*V)", rows, k, V, kBeamMax);
  hipLaunchKernelGGL(beam_topk_kernel, dim3(1), dim3(kBeamThreads), 0, stream, logits, ld, rows, V,
                     prev, k, out_scores, out_index);
  CAPNET_LAUNCH_CHECK();
  return kOk;
}

// meta: DEVICE array [n][3] = (first row, rows that compete, k) per image, rows and k <= 16 (the caller checks: the values
// live on the device); outputs [n][16]
int beam_topk_batched(const float* logits, long ld, int V, const float* prev, const int* meta, int n, float* out_scores,
                      long long* out_index, hipStream_t stream) {
  CAPNET_REQUIRE(logits && prev && meta && out_scores && out_index && n >= 0 && V >= 1 && ld >= V, "beam_topk_batched: bad argument");
  if (n == 0) return kOk;
  hipLaunchKernelGGL(beam_topk_batched_kernel, dim3(n), dim3(kBeamThreads), 0, stream, logits, ld, V, prev, meta, out_scores,
                     out_index);
  CAPNET_LAUNCH_CHECK();
  return kOk;
}

// ---- dropout between stacked layers: dst = src * mask / keep over rows [r0, r1) ------------------------------------
__global__ __launch_bounds__(256) void rows_dropout_kernel(const float* __restrict__ src, float* __restrict__ dst, int r0,
                                                           int rows, int C, float p, unsigned long long seed, int layer,
                                                           int use_dropout) {
  const long total = (long)rows * C;
  const float inv_keep = p < 1.f ? 1.f / (1.f - p) : 0.f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int r = r0 + (int)(i / C), e = (int)(i % C);
    float v = src[(long)r * C + e];
    // (the mask's stream: the layer number stands where the input dropout has the caption column, offset out of its range)
    if (use_dropout) v *= dropout_scale(seed, r, 0x40000000 + layer, e, p, inv_keep);
    dst[(long)r * C + e] = v;
  }
}

int rows_dropout(const float* src, float* dst, int r0, int r1, int C, float p, unsigned long long seed, int layer,
                 int use_dropout, hipStream_t stream) {
  CAPNET_REQUIRE(src && dst && r0 >= 0 && r1 >= r0 && C > 0, "rows_dropout: bad argument");
  if (r1 == r0) return kOk;
  const long total = (long)(r1 - r0) * C;
  hipLaunchKernelGGL(rows_dropout_kernel, dim3((int)(cdiv(total, 256) > 2048 ? 2048 : cdiv(total, 256))), dim3(256), 0, stream,
                     src, dst, r0, r1 - r0, C, p, seed, layer, use_dropout);
  CAPNET_LAUNCH_CHECK();
  return kOk;
}

// ---- out[r] = idx[r] >= 0 ? src[idx[r]] : 0  (rows of width C) ----------------------------
__global__ __launch_bounds__(256) void gather_rows_kernel(const float* __restrict__ src,
                                                          const int* __restrict__ idx,
                                                          float* __restrict__ out, int rows,
                                                          int C) {
  const long total = (long)rows * C;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long)gridDim.x * blockDim.x) {
    const int r = (int)(i / C), c = (int)(i - (long)r * C);
    const int s = idx[r];
    out[i] = s >= 0 ? src[(long)s * C + c] : 0.f;
  }
}

int gather_rows(const float* src, const int* idx, float* out, int rows, int C,
                hipStream_t stream) {
  if (rows <= 0) return kOk;
  const long total = (long)rows * C;
  hipLaunchKernelGGL(gather_rows_kernel, dim3((int)(cdiv(total, 256) > 4096 ? 4096 : cdiv(total, 256))),
                     dim3(256), 0, stream, src, idx, out, rows, C);
  CAPNET_LAUNCH_CHECK();
  return kOk;
}

// out[r] = idx[r] >= 0 ? src[idx[r]] : first[sample[r]]  (h_{t-1} of every packed row when the
// initial state is not zero: attention decoders, model_att.py:260)
__global__ __launch_bounds__(256) void gather_prev_rows_kernel(const float* __restrict__ src,
                                                               const int* __restrict__ idx,
                                                               const float* __restrict__ first,
                                                               const int* __restrict__ sample,
                                                               float* __restrict__ out, int rows,
                                                               int C) {
  const long total = (long)rows * C;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long)gridDim.x * blockDim.x) {
    const int r = (int)(i / C), c = (int)(i - (long)r * C);
    const int s = idx[r];
    out[i] = s >= 0 ? src[(long)s * C + c] : first[(long)sample[r] * C + c];
  }
}

int gather_prev_rows(const float* src, const int* idx, const float* first, const int* sample,
                     float* out, int rows, int C, hipStream_t stream) {
  if (rows <= 0) return kOk;
  const long total = (long)rows * C;
  hipLaunchKernelGGL(gather_prev_rows_kernel,
                     dim3((int)(cdiv(total, 256) > 4096 ? 4096 : cdiv(total, 256))), dim3(256), 0,
                     stream, src, idx, first, sample, out, rows, C);
  CAPNET_LAUNCH_CHECK();
  return kOk;
}

// ---- column sums: out[c] (+)= sum_r x[r][c] ------------------------------------------------
// Bias gradients. A workgroup owns 32 columns (one 128-B segment per row) and a chunk of rows;
// its 32 row lanes each keep 8 loads in flight and are summed through LDS in a fixed order
// (deterministic). With a workspace, long inputs are cut into row chunks whose partial sums are
// added by the slab reducer of gemm_f32.hip.
constexpr int kColsumCols = 32, kColsumLanes = 32;

__global__ __launch_bounds__(kColsumCols* kColsumLanes) void colsum_kernel(
    const float* __restrict__ x, long ld, int rows, int C, int rows_per_chunk,
    float* __restrict__ out, long out_chunk_stride, int accumulate) {
  __shared__ float s[kColsumLanes][kColsumCols + 1];
  const int cx = threadIdx.x % kColsumCols, ry = threadIdx.x / kColsumCols;
  const int c = blockIdx.x * kColsumCols + cx;
  const int r0 = blockIdx.y * rows_per_chunk, r1 = min(rows, r0 + rows_per_chunk);
  float acc = 0.f;
  if (c < C) {
    for (int r = r0 + ry; r < r1; r += 8 * kColsumLanes) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = x[(long)min(r + u * kColsumLanes, r1 - 1) * ld + c];
#pragma unroll
      for (int u = 0; u < 8; ++u) acc += (r + u * kColsumLanes < r1) ? v[u] : 0.f;
    }
  }
  s[ry][cx] = acc;
  __syncthreads();
  if (ry == 0 && c < C) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < kColsumLanes; ++k) t += s[k][cx];
    float* o = out + (long)blockIdx.y * out_chunk_stride + c;
    *o = accumulate ? *o + t : t;
  }
}

int colsum(const float* x, long ld, int rows, int C, float* out, int accumulate, hipStream_t stream,
           float* ws, size_t ws_floats) {
  CAPNET_REQUIRE(x && out && C > 0 && rows >= 0, "colsum: bad argument");
  int chunks = 1;
  if (ws && rows > 4096) {
    chunks = min(32, rows / 1024);
    if ((size_t)chunks * C > ws_floats) chunks = 1;
  }
  const int rpc = chunks > 1 ? cdiv(rows, chunks) : (rows > 0 ? rows : 1);
  if (chunks > 1) chunks = cdiv(rows, rpc);
  if (chunks <= 1) {
    hipLaunchKernelGGL(colsum_kernel, dim3(cdiv(C, kColsumCols), 1), dim3(kColsumCols * kColsumLanes), 0,
                       stream, x, ld, rows, C, rpc, out, 0l, accumulate);
    CAPNET_LAUNCH_CHECK();
    return kOk;
  }
  hipLaunchKernelGGL(colsum_kernel, dim3(cdiv(C, kColsumCols), chunks), dim3(kColsumCols * kColsumLanes),
                     0, stream, x, ld, rows, C, rpc, ws, (long)C, 0);
  CAPNET_LAUNCH_CHECK();
  return reduce_slabs(ws, chunks, 1, C, out, C, nullptr, accumulate, stream);
}

// ---- embedding / feature gradient scatter --------------------------------------------------
// dX [N][E] -> dEmb[token] += the rows of that token, dFeat[sample] = the sample's feature row. Tokens repeat: the FIRST row
// of a token sums every row of it in row order and adds the sum -- one writer per table row, a fixed order (an atomic add per
// row was the one launch of the step whose result depended on arrival order: the checkpoint round trip's bitwise comparison
// failed one run in three on the last bit of the third loss).
// Who is first and how many rows a token has comes from two integer tables over the vocabulary (scatter_count_kernel:
// atomicMax / atomicAdd on ints -- order-free): most tokens occur once and their row is copied without looking at any
// other; an owner of several rows scans the tokens from its own row on, 128 at a time through wave ballots, until it has met
// them all. (Every workgroup scanning all N tokens for its turn, the first form: 255 us at 2 000 rows.)
constexpr int kScatterE = 4;                 // columns per thread and sweep: E <= 512 in one
__global__ __launch_bounds__(256) void scatter_count_kernel(const int* __restrict__ row_col, const int* __restrict__ row_token,
                                                            int N, int V, int* __restrict__ first_enc, int* __restrict__ cnt) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= N || row_col[r] == -1) return;
  const int tok = row_token[r];
  if (tok < 0 || tok >= V) return;
  atomicMax(first_enc + tok, N - r);         // (zero-initialised: the first row has the largest N - r)
  atomicAdd(cnt + tok, 1);
}

__global__ __launch_bounds__(128) void scatter_input_grad_kernel(
    const float* __restrict__ dX, long ldx, int N, int E, const int* __restrict__ row_sample,
    const int* __restrict__ row_col, const int* __restrict__ row_token, float* __restrict__ dEmb,
    float* __restrict__ dFeat, int V, float p, unsigned long long seed, int use_dropout,
    const int* __restrict__ first_enc, const int* __restrict__ cnt) {
  __shared__ unsigned long long s_mask[2][2];
  const int r = blockIdx.x, tid = threadIdx.x;
  const float inv_keep = p < 1.f ? 1.f / (1.f - p) : 0.f;
  if (row_col[r] == -1) {
    const int sample = row_sample[r];
    if (dFeat)
      for (int e = tid; e < E; e += blockDim.x)
        dFeat[(long)sample * E + e] = dX[(long)r * ldx + e];
    return;
  }
  const int tok = row_token[r];
  if (tok < 0 || tok >= V) return;
  if (first_enc[tok] != N - r) return;
  const int want = cnt[tok];                 // rows of this token, this one included
  for (int e0 = 0; e0 < E; e0 += 128 * kScatterE) {
    float acc[kScatterE];
#pragma unroll
    for (int u = 0; u < kScatterE; ++u) acc[u] = 0.f;
    // rows j[0 .. n) in order, their loads all in flight before the first add (a token of many rows -- <start>: one per
    // caption -- was a chain of n dependent round trips: 97 us at 64 captions)
    auto add_rows = [&](const int (&j)[8], int n) {
      float g[8][kScatterE];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int jk = j[k < n ? k : 0];
        const int sample = row_sample[jk], col = row_col[jk];
        const bool drop = (col >= 0) && use_dropout;
#pragma unroll
        for (int u = 0; u < kScatterE; ++u) {
          const int e = e0 + u * 128 + tid;
          float v = (e < E) ? dX[(long)jk * ldx + e] : 0.f;
          if (drop && e < E) v *= dropout_scale(seed, sample, col, e, p, inv_keep);
          g[k][u] = v;
        }
      }
#pragma unroll
      for (int k = 0; k < 8; ++k)
        if (k < n) {
#pragma unroll
          for (int u = 0; u < kScatterE; ++u) acc[u] += g[k][u];
        }
    };
    int jb[8] = {r, r, r, r, r, r, r, r};
    if (want == 1) {
      add_rows(jb, 1);
    } else {
      int met = 0, nb = 0;
      for (int base = r & ~127, it = 0; base < N && met < want; base += 128, ++it) {
        const int rr = base + tid;
        const bool match = rr < N && rr >= r && row_col[rr] != -1 && row_token[rr] == tok;
        const unsigned long long m = __ballot(match);
        if ((tid & 63) == 0) s_mask[it & 1][tid >> 6] = m;
        __syncthreads();                       // (two buffers: the next round's writes cannot pass this round's reads)
        const unsigned long long m0 = s_mask[it & 1][0], m1 = s_mask[it & 1][1];
        for (int half = 0; half < 2; ++half) {
          unsigned long long mm = half ? m1 : m0;
          while (mm) {
            jb[nb++] = base + 64 * half + __builtin_ctzll(mm);
            mm &= mm - 1;
            ++met;
            if (nb == 8) { add_rows(jb, 8); nb = 0; }
          }
        }
      }
      if (nb) add_rows(jb, nb);
    }
#pragma unroll
    for (int u = 0; u < kScatterE; ++u) {
      const int e = e0 + u * 128 + tid;
      if (e < E) dEmb[(long)tok * E + e] += acc[u];
    }
  }
}

// tables: workspace of at least 2 V ints
int scatter_input_grad(const float* dX, long ldx, int N, int E, const int* row_sample, const int* row_col,
                       const int* row_token, float* dEmb, float* dFeat, int V, float p,
                       unsigned long long seed, int use_dropout, hipStream_t stream, int* tables, size_t table_ints) {
  if (N <= 0) return kOk;
  CAPNET_REQUIRE(tables && table_ints >= 2 * (size_t)V, "scatter_input_grad: workspace for two tables of V = %d ints", V);
  int* first_enc = tables;
  int* cnt = tables + V;
  CAPNET_HIP_CHECK(hipMemsetAsync(tables, 0, 2 * (size_t)V * sizeof(int), stream));
  hipLaunchKernelGGL(scatter_count_kernel, dim3(cdiv(N, 256)), dim3(256), 0, stream, row_col, row_token, N, V, first_enc, cnt);
  hipLaunchKernelGGL(scatter_input_grad_kernel, dim3(N), dim3(128), 0, stream, dX, ldx, N, E, row_sample,
                     row_col, row_token, dEmb, dFeat, V, p, seed, use_dropout, first_enc, cnt);
  CAPNET_LAUNCH_CHECK();
  return kOk;
}

}  // namespace capnet
