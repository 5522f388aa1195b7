// Additive ("soft") attention step of the attention decoders, forward and backward.
// Follows Attention.forward (stylenet/model_att.py:51-70) and the gate of
// DecoderFactoredLSTMAtt.forward (:279-290):
//   att1 = encoder_att(features)            -- HOISTED: it does not depend on the time step, the
//                                              reference recomputes it every step (:59,279)
//   att2 = decoder_att(h)                   e = full_att(relu(att1 + att2))      alpha = softmax_P(e)
//   awe  = sum_p alpha[p] * features[p]     gate = sigmoid(f_beta(h))            out = gate * awe
// All four kernels are HBM-bound: per decoded row they stream att1 (P*A*4 B) and the feature
// map (P*C*4 B). Rows of a time step are independent; one step's rows map to distinct samples.
#include "common.h"
#include "kernels.h"

namespace capnet {

constexpr int kAttThreads = 256;

// Sum over the 64 lanes, the same value in every lane. Inside a row of 16 lanes by DPP (four VALU adds: no trip
// through the LDS crossbar, which is what __shfl_xor compiles to -- six dependent ds_bpermute per sum, and the
// per-pixel sums of the score / context kernels are chains of them), the four rows through v_readlane.
template <int CTRL> __device__ __forceinline__ float dpp_get(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float wave_sum(float v) {
  v += dpp_get<0xB1>(v);      // quad_perm [1,0,3,2]
  v += dpp_get<0x4E>(v);      // quad_perm [2,3,0,1]
  v += dpp_get<0x141>(v);     // row_half_mirror
  v += dpp_get<0x140>(v);     // row_mirror: every lane of a row holds the row's sum
  const int b = __builtin_bit_cast(int, v);
  const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 0));
  const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 16));
  const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 32));
  const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 48));
  return ((r0 + r1) + r2) + r3;
}

// ---- forward 1: raw scores ----------------------------------------------------------------
// grid = (rows of the step, pixel chunks): 4 chunks at b = 64 (256 workgroups), more for fewer rows so that a
// 12-row step still makes ~150 workgroups (score_chunks). A pixel's score is one wave's sum whatever the chunking.
// att1 [B][P][A] (indexed by sample), att2 rows at ld `ldz`. escore [rows][P]: e before softmax.
static int score_chunks(int rows, int P) {
  int c = (512 + rows - 1) / rows;
  const int most = (P + 15) / 16;          // >= 16 pixels per workgroup: one sweep of its four waves
  if (c > most) c = most;
  return c < 4 ? 4 : c;
}

__global__ __launch_bounds__(kAttThreads) void att_scores_fwd_kernel(
    const float* __restrict__ att1, const float* __restrict__ att2, long ldz,
    const float* __restrict__ wf, const float* __restrict__ bf, int P, int A,
    float* __restrict__ escore) {
  const int j = blockIdx.x;      // row inside the step == sample index
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int pc = (P + (int)gridDim.y - 1) / (int)gridDim.y;
  const int p0 = blockIdx.y * pc, p1 = min(P, p0 + pc);
  if (p0 >= p1) return;
  const float* a1 = att1 + (long)j * P * A;
  const float* a2 = att2 + (long)j * ldz;
  const float b0 = bf[0];
  // a wave takes 4 pixels at a time (4 x A/256 independent 16-B loads in flight per lane)
  for (int p = p0 + 4 * wave; p < p1; p += 4 * (kAttThreads / 64)) {
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    for (int a = lane * 4; a < A; a += 256) {
      const float4 y = *reinterpret_cast<const float4*>(a2 + a);
      const float4 w = *reinterpret_cast<const float4*>(wf + a);
      float4 x[4];
#pragma unroll
      for (int u = 0; u < 4; ++u)
        x[u] = *reinterpret_cast<const float4*>(a1 + (long)min(p + u, p1 - 1) * A + a);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        s[u] = fmaf(fmaxf(x[u].x + y.x, 0.f), w.x, s[u]);
        s[u] = fmaf(fmaxf(x[u].y + y.y, 0.f), w.y, s[u]);
        s[u] = fmaf(fmaxf(x[u].z + y.z, 0.f), w.z, s[u]);
        s[u] = fmaf(fmaxf(x[u].w + y.w, 0.f), w.w, s[u]);
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float t = wave_sum(s[u]);
      if (lane == 0 && p + u < p1) escore[(long)j * P + p + u] = t + b0;
    }
  }
}

// ---- forward 2: softmax + context vector + gate ----------------------------------------------
// grid = (rows, C/256); 256 threads: lane = 4 channels (float4), the four waves split the pixels (a quarter each, 14
// independent 16-B loads in flight per lane: 4 dependent round trips for P = 196 where one wave per 512 channels needed
// 14) and their partial sums meet in LDS, added in wave order. Every workgroup of a row recomputes the row's
// softmax over P from the raw scores (P is a few hundred values); workgroup y = 0 stores alpha
// (packed [N][P] row) and the user-visible alphas_bt [B][steps][P] row (j, t).
// gate_io: in = f_beta(h) pre-activation (ld ldz), out = sigmoid of it (saved for backward).
// awe_out [rows][C] (pre-gate, saved), xa_out: the decoder input slice (ld ldx) = gate * awe.
constexpr int kCtxCh = 256;

__global__ __launch_bounds__(kAttThreads) void att_context_fwd_kernel(
    const float* __restrict__ feat, const float* __restrict__ escore, int P, int C,
    float* __restrict__ gate_io, long ldz, float* __restrict__ alpha_out,
    float* __restrict__ alphas_bt, int steps, int t, float* __restrict__ awe_out,
    float* __restrict__ xa_out, long ldx) {
  extern __shared__ float al[];  // alpha[P], then the waves' partial sums [4][64] float4
  __shared__ float red[8];
  const int j = blockIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float m = -INFINITY;
  for (int p = threadIdx.x; p < P; p += kAttThreads) {
    const float e = escore[(long)j * P + p];
    al[p] = e;
    m = fmaxf(m, e);
  }
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  if (lane == 0) red[wave] = m;
  __syncthreads();
  m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  float z = 0.f;
  for (int p = threadIdx.x; p < P; p += kAttThreads) {
    const float e = expf(al[p] - m);
    al[p] = e;
    z += e;
  }
  z = wave_sum(z);
  if (lane == 0) red[4 + wave] = z;
  __syncthreads();
  const float inv = 1.f / (red[4] + red[5] + red[6] + red[7]);
  for (int p = threadIdx.x; p < P; p += kAttThreads) {
    const float a = al[p] * inv;
    al[p] = a;
    if (blockIdx.y == 0) {
      alpha_out[(long)j * P + p] = a;
      alphas_bt[((long)j * steps + t) * P + p] = a;
    }
  }
  __syncthreads();
  const int c = blockIdx.y * kCtxCh + lane * 4;
  const float* f = feat + (long)j * P * C + c;
  const int pq = (P + 3) / 4, pa = wave * pq, pb = min(P, pa + pq);
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  int p = pa;
  for (; p + 14 <= pb; p += 14) {   // 14 independent 16-B loads in flight per lane
    float4 v[14];
#pragma unroll
    for (int u = 0; u < 14; ++u) v[u] = *reinterpret_cast<const float4*>(f + (long)(p + u) * C);
#pragma unroll
    for (int u = 0; u < 14; ++u) {
      const float a = al[p + u];
      s.x = fmaf(a, v[u].x, s.x); s.y = fmaf(a, v[u].y, s.y);
      s.z = fmaf(a, v[u].z, s.z); s.w = fmaf(a, v[u].w, s.w);
    }
  }
  if (p < pb) {                      // the rest of the quarter, still all at once (clamped, weight 0 past the end)
    float4 v[14];
#pragma unroll
    for (int u = 0; u < 14; ++u) v[u] = *reinterpret_cast<const float4*>(f + (long)min(p + u, pb - 1) * C);
#pragma unroll
    for (int u = 0; u < 14; ++u) {
      const float a = p + u < pb ? al[p + u] : 0.f;
      s.x = fmaf(a, v[u].x, s.x); s.y = fmaf(a, v[u].y, s.y);
      s.z = fmaf(a, v[u].z, s.z); s.w = fmaf(a, v[u].w, s.w);
    }
  }
  float4* part = reinterpret_cast<float4*>(al + ((P + 3) & ~3));
  part[wave * 64 + lane] = s;
  __syncthreads();
  if (wave != 0) return;
  {
    const float4 s1 = part[64 + lane], s2 = part[128 + lane], s3 = part[192 + lane];
    s.x = ((s.x + s1.x) + s2.x) + s3.x; s.y = ((s.y + s1.y) + s2.y) + s3.y;
    s.z = ((s.z + s1.z) + s2.z) + s3.z; s.w = ((s.w + s1.w) + s2.w) + s3.w;
  }
  float4 g = *reinterpret_cast<const float4*>(gate_io + (long)j * ldz + c);
  g.x = 1.f / (1.f + expf(-g.x)); g.y = 1.f / (1.f + expf(-g.y));
  g.z = 1.f / (1.f + expf(-g.z)); g.w = 1.f / (1.f + expf(-g.w));
  *reinterpret_cast<float4*>(gate_io + (long)j * ldz + c) = g;
  *reinterpret_cast<float4*>(awe_out + (long)j * C + c) = s;
  float4 o;
  o.x = g.x * s.x; o.y = g.y * s.y; o.z = g.z * s.z; o.w = g.w * s.w;
  *reinterpret_cast<float4*>(xa_out + (long)j * ldx + c) = o;
}

int att_step_fwd(const float* att1, const float* feat, const float* att2, float* gate_io, long ldz,
                 const float* wf, const float* bf, int rows, int P, int A, int C,
                 float* alpha_out, float* alphas_bt, int steps, int t, float* awe_out,
                 float* xa_out, long ldx, float* escore, hipStream_t stream) {
  if (rows <= 0) return kOk;
  CAPNET_REQUIRE(att1 && feat && att2 && gate_io && wf && bf && alpha_out && alphas_bt && awe_out &&
                     xa_out && escore, "att_step_fwd: null argument");
  CAPNET_REQUIRE(A % 4 == 0 && C % 512 == 0 && P > 0 && P <= 4096 && ldz % 4 == 0 && ldx % 4 == 0,
                 "att_step_fwd: A=%d C=%d P=%d (need A%%4==0, C%%512==0)", A, C, P);
  CAPNET_REQUIRE(aligned16(feat) && aligned16(gate_io) && aligned16(awe_out) && aligned16(xa_out) && aligned16(att1) &&
                     aligned16(att2) && aligned16(wf), "att_step_fwd: 16-byte alignment of the rows");
  hipLaunchKernelGGL(att_scores_fwd_kernel, dim3(rows, score_chunks(rows, P)), dim3(kAttThreads), 0, stream,
                     att1, att2, ldz, wf, bf, P, A, escore);
  hipLaunchKernelGGL(att_context_fwd_kernel, dim3(rows, C / kCtxCh), dim3(kAttThreads),
                     (((P + 3) & ~3) + 4 * 64 * 4) * sizeof(float), stream, feat, escore, P, C, gate_io, ldz, alpha_out, alphas_bt, steps, t, awe_out,
                     xa_out, ldx);
  CAPNET_LAUNCH_CHECK();
  return kOk;
}

// ---- backward 1: gate and d(alpha) partials -----------------------------------------------
// grid = (rows, C/256), 256 threads = 4 waves; a wave owns 14 pixels at a time (14 independent
// 16-B loads in flight: 4 sweeps for P = 196), its lanes 256 channels (float4). dxa: gradient of the gated context
// (ld ldx). Outputs: dgate_out (ld ldz) = d f_beta pre-activation; dalpha_part [rows][C/256][P].
// dxa_slabs (optional): d[x | ctx] still lies in n_slabs K-chunk partials [k][rows][emb_cols + C] (sgemm_rows16_slabs); every
// workgroup sums its own channels, in slab order, and the row's first one also sums the embedding columns into
// dxa[-emb_cols .. 0) (what scatter_input_grad reads after the time loop).
__global__ __launch_bounds__(kAttThreads) void att_context_bwd_kernel(
    const float* __restrict__ feat, float* __restrict__ dxa, long ldx,
    const float* __restrict__ gate, long ldzg, const float* __restrict__ awe, int P, int C,
    float* __restrict__ dgate_out, long ldz, float* __restrict__ dalpha_part,
    const float* __restrict__ dxa_slabs, int n_slabs, int emb_cols) {
  const int j = blockIdx.x, cb = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = cb * kCtxCh + lane * 4;
  float d[4];
  {
    // (three independent 16-B loads, then the store: element-wise dword loads around a conditional store came out
    //  as four dependent round trips)
    float4 dg4;
    if (n_slabs > 0) {
      const long xw = emb_cols + C, stride = (long)gridDim.x * xw;
      auto slab_sum = [&](const float* q) {
        float4 v[8], a = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int k = 0; k < n_slabs; k += 8) {
#pragma unroll
          for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const float4*>(q + (long)min(k + u, n_slabs - 1) * stride);
#pragma unroll
          for (int u = 0; u < 8; ++u)
            if (k + u < n_slabs) { a.x += v[u].x; a.y += v[u].y; a.z += v[u].z; a.w += v[u].w; }
        }
        return a;
      };
      dg4 = slab_sum(dxa_slabs + (long)j * xw + emb_cols + c);
      if (cb == 0 && (int)threadIdx.x * 4 < emb_cols)
        *reinterpret_cast<float4*>(dxa + (long)j * ldx - emb_cols + 4 * threadIdx.x) = slab_sum(dxa_slabs + (long)j * xw + 4 * threadIdx.x);
    } else {
      dg4 = *reinterpret_cast<const float4*>(dxa + (long)j * ldx + c);
    }
    const float4 g4 = *reinterpret_cast<const float4*>(gate + (long)j * ldzg + c);
    const float4 aw4 = *reinterpret_cast<const float4*>(awe + (long)j * C + c);
    const float dg[4] = {dg4.x, dg4.y, dg4.z, dg4.w}, g[4] = {g4.x, g4.y, g4.z, g4.w}, aw[4] = {aw4.x, aw4.y, aw4.z, aw4.w};
    float o[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      d[k] = dg[k] * g[k];  // d(awe)
      o[k] = dg[k] * aw[k] * g[k] * (1.f - g[k]);
    }
    if (wave == 0) *reinterpret_cast<float4*>(dgate_out + (long)j * ldz + c) = make_float4(o[0], o[1], o[2], o[3]);
  }
  const float* f = feat + (long)j * P * C + c;
  constexpr int U = 14;
  for (int p = U * wave; p < P; p += U * (kAttThreads / 64)) {
    float4 v0[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v0[u] = *reinterpret_cast<const float4*>(f + (long)min(p + u, P - 1) * C);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      float s = d[0] * v0[u].x + d[1] * v0[u].y + d[2] * v0[u].z + d[3] * v0[u].w;
      s = wave_sum(s);
      if (lane == 0 && p + u < P) dalpha_part[((long)j * gridDim.y + cb) * P + p + u] = s;
    }
  }
}

// ---- backward 2: softmax, relu, full_att / decoder_att gradients ---------------------------
// grid = (rows, A/128): a workgroup owns 128 attention channels of a row (lane = 2 channels),
// its 4 waves sweep the pixels 14 at a time. Every workgroup of a row recomputes d e = softmax
// backward over P (a few hundred values); workgroup y = 0 stores it (de_out [rows][P], read after
// the time loop by att_datt1_kernel) and the full_att bias gradient.
// dalphas_bt: gradient of the user-visible alphas tensor [B][steps][P] (may be null).
// Outputs: datt2 (ld ldz), dwf_rows [rows][A], dbf_rows [rows], de_out.
constexpr int kScoreBwdCh = 128;

__global__ __launch_bounds__(kAttThreads) void att_scores_bwd_kernel(
    const float* __restrict__ att1, const float* __restrict__ att2, long ldz2,
    const float* __restrict__ wf, const float* __restrict__ alpha,
    const float* __restrict__ dalpha_part, int nparts, const float* __restrict__ dalphas_bt,
    int steps, int t, int P, int A, float* __restrict__ datt2, long ldz,
    float* __restrict__ de_out, float* __restrict__ dwf_rows, float* __restrict__ dbf_rows) {
  extern __shared__ float sh[];  // de[P], then cross-wave scratch [4][128] x 2
  __shared__ float red[4];
  const int j = blockIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float* de = sh;
  float part = 0.f;
  for (int p = threadIdx.x; p < P; p += kAttThreads) {
    float da = dalphas_bt ? dalphas_bt[((long)j * steps + t) * P + p] : 0.f;
    for (int k = 0; k < nparts; ++k) da += dalpha_part[((long)j * nparts + k) * P + p];
    de[p] = da;
    part = fmaf(alpha[(long)j * P + p], da, part);
  }
  part = wave_sum(part);
  if (lane == 0) red[wave] = part;
  __syncthreads();
  const float dot = red[0] + red[1] + red[2] + red[3];
  float dbf = 0.f;
  for (int p = threadIdx.x; p < P; p += kAttThreads) {
    const float v = alpha[(long)j * P + p] * (de[p] - dot);
    de[p] = v;
    dbf += v;
    if (blockIdx.y == 0) de_out[(long)j * P + p] = v;
  }
  __syncthreads();
  dbf = wave_sum(dbf);
  if (lane == 0) red[wave] = dbf;
  __syncthreads();
  if (threadIdx.x == 0 && blockIdx.y == 0) dbf_rows[j] = red[0] + red[1] + red[2] + red[3];

  const int a0 = blockIdx.y * kScoreBwdCh + lane * 2;
  float* s_d2 = sh + P;                    // [4][128]
  float* s_dw = sh + P + 4 * kScoreBwdCh;  // [4][128]
  float2 g2 = make_float2(0.f, 0.f), gw = make_float2(0.f, 0.f);
  if (a0 < A) {
    const float* a1 = att1 + (long)j * P * A + a0;
    const float2 y = *reinterpret_cast<const float2*>(att2 + (long)j * ldz2 + a0);
    const float2 w = *reinterpret_cast<const float2*>(wf + a0);
    constexpr int U = 14;
    for (int p = U * wave; p < P; p += U * (kAttThreads / 64)) {
      float2 x[U];
#pragma unroll
      for (int u = 0; u < U; ++u)
        x[u] = *reinterpret_cast<const float2*>(a1 + (long)min(p + u, P - 1) * A);
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const float e = p + u < P ? de[p + u] : 0.f;
        const float z0 = x[u].x + y.x, z1 = x[u].y + y.y;
        g2.x += z0 > 0.f ? e * w.x : 0.f;
        g2.y += z1 > 0.f ? e * w.y : 0.f;
        gw.x = fmaf(e, fmaxf(z0, 0.f), gw.x);
        gw.y = fmaf(e, fmaxf(z1, 0.f), gw.y);
      }
    }
  }
  *reinterpret_cast<float2*>(s_d2 + wave * kScoreBwdCh + lane * 2) = g2;
  *reinterpret_cast<float2*>(s_dw + wave * kScoreBwdCh + lane * 2) = gw;
  __syncthreads();
  for (int k = threadIdx.x; k < kScoreBwdCh; k += kAttThreads) {
    const int a = blockIdx.y * kScoreBwdCh + k;
    if (a < A) {
      datt2[(long)j * ldz + a] = s_d2[k] + s_d2[kScoreBwdCh + k] + s_d2[2 * kScoreBwdCh + k] +
                                 s_d2[3 * kScoreBwdCh + k];
      dwf_rows[(long)j * A + a] = s_dw[k] + s_dw[kScoreBwdCh + k] + s_dw[2 * kScoreBwdCh + k] +
                                  s_dw[3 * kScoreBwdCh + k];
    }
  }
}

int att_step_bwd(const float* att1, const float* feat, const float* att2, long ldz2,
                 const float* gate, long ldzg, const float* awe, const float* alpha,
                 const float* wf, float* dxa, long ldx, const float* dalphas_bt, int steps,
                 int t, int rows, int P, int A, int C, float* dalpha_part, float* dgate_out,
                 float* datt2, long ldz, float* de_out, float* dwf_rows, float* dbf_rows,
                 hipStream_t stream, const float* dxa_slabs, int n_slabs, int emb_cols) {
  if (rows <= 0) return kOk;
  CAPNET_REQUIRE(att1 && feat && att2 && gate && awe && alpha && wf && dxa && dalpha_part &&
                     dgate_out && datt2 && de_out && dwf_rows && dbf_rows,
                 "att_step_bwd: null argument");
  CAPNET_REQUIRE(ldzg % 4 == 0 && aligned16(dxa) && aligned16(gate) && aligned16(awe) && aligned16(dgate_out) && aligned16(feat),
                 "att_step_bwd: 16-byte alignment of the context rows");
  CAPNET_REQUIRE(!dxa_slabs || (emb_cols % 4 == 0 && emb_cols <= 4 * kAttThreads && aligned16(dxa_slabs)),
                 "att_step_bwd: embedding columns of the slab form");
  CAPNET_REQUIRE(A % 4 == 0 && C % kCtxCh == 0 && P > 0 && ldz % 4 == 0 && ldx % 4 == 0 && ldz2 % 2 == 0 &&
                     (size_t)(P + 8 * kScoreBwdCh) * 4 <= 64 * 1024,
                 "att_step_bwd: A=%d C=%d P=%d", A, C, P);
  // gate rows use their own leading dimension (saved forward Z buffer)
  hipLaunchKernelGGL(att_context_bwd_kernel, dim3(rows, C / kCtxCh), dim3(kAttThreads), 0, stream, feat,
                     dxa, ldx, gate, ldzg, awe, P, C, dgate_out, ldz, dalpha_part, dxa_slabs, dxa_slabs ? n_slabs : 0, emb_cols);
  hipLaunchKernelGGL(att_scores_bwd_kernel, dim3(rows, cdiv(A, kScoreBwdCh)), dim3(kAttThreads),
                     (P + 8 * kScoreBwdCh) * sizeof(float), stream, att1, att2, ldz2, wf, alpha,
                     dalpha_part, C / kCtxCh, dalphas_bt, steps, t, P, A, datt2, ldz, de_out, dwf_rows,
                     dbf_rows);
  CAPNET_LAUNCH_CHECK();
  return kOk;
}

// ---- after the time loop: gradient of att1 = encoder_att(features) ---------------------------
//   d att1[b][p][a] = w_full[a] * sum_t de[row(b,t)][p] * [att1[b][p][a] + att2[row(b,t)][a] > 0]
// One pass over att1 (read once, d att1 written once) instead of a read-modify-write of the
// whole [B][P][A] tensor in every step of the BPTT loop. grid = (B, pixel chunks), 128 threads,
// lane = 4 channels at a time; the sample's att2 rows and de rows sit in LDS.
struct AttRowTable { int off[kMaxSteps + 1]; int steps; };
constexpr int kDatt1Pix = 14;

__global__ __launch_bounds__(128) void att_datt1_kernel(
    const float* __restrict__ att1, const float* __restrict__ att2_rows, long ldz2,
    const float* __restrict__ de_rows, const float* __restrict__ wf, AttRowTable rt, int P, int A,
    int t_chunk, float* __restrict__ datt1) {
  extern __shared__ float sh[];  // att2 [t_chunk][A] then de [t_chunk][kDatt1Pix]
  const int b = blockIdx.x;
  const int p0 = blockIdx.y * kDatt1Pix, np = min(kDatt1Pix, P - p0);
  int T = 0;  // steps in which sample b is alive: rows off[t] + b while b < batch size of step t
  while (T < rt.steps && b < rt.off[T + 1] - rt.off[T]) ++T;
  float* s_a2 = sh;
  float* s_de = sh + (size_t)t_chunk * A;
  // t_chunk covers every step unless the sequence is unusually long (then later chunks add)
  for (int t0 = 0; t0 == 0 || t0 < T; t0 += t_chunk) {
    const int tc = max(0, min(t_chunk, T - t0));
    __syncthreads();
    for (int i = threadIdx.x; i < tc * (A / 4); i += 128) {
      const int t = i / (A / 4), a4 = i - t * (A / 4);
      *reinterpret_cast<float4*>(s_a2 + (size_t)t * A + 4 * a4) =
          *reinterpret_cast<const float4*>(att2_rows + (long)(rt.off[t0 + t] + b) * ldz2 + 4 * a4);
    }
    for (int i = threadIdx.x; i < tc * kDatt1Pix; i += 128) {
      const int t = i / kDatt1Pix, q = i - t * kDatt1Pix;
      s_de[i] = q < np ? de_rows[(long)(rt.off[t0 + t] + b) * P + p0 + q] : 0.f;
    }
    __syncthreads();
    for (int a = threadIdx.x * 4; a < A; a += 512) {
      const float4 w = *reinterpret_cast<const float4*>(wf + a);
      for (int q = 0; q < np; ++q) {
        const long idx = ((long)b * P + p0 + q) * A + a;
        const float4 x = *reinterpret_cast<const float4*>(att1 + idx);
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int t = 0; t < tc; ++t) {
          const float e = s_de[t * kDatt1Pix + q];
          const float4 y = *reinterpret_cast<const float4*>(s_a2 + (size_t)t * A + a);
          acc.x += x.x + y.x > 0.f ? e : 0.f;
          acc.y += x.y + y.y > 0.f ? e : 0.f;
          acc.z += x.z + y.z > 0.f ? e : 0.f;
          acc.w += x.w + y.w > 0.f ? e : 0.f;
        }
        acc.x *= w.x; acc.y *= w.y; acc.z *= w.z; acc.w *= w.w;
        if (t0 > 0) {
          const float4 prev = *reinterpret_cast<const float4*>(datt1 + idx);
          acc.x += prev.x; acc.y += prev.y; acc.z += prev.z; acc.w += prev.w;
        }
        *reinterpret_cast<float4*>(datt1 + idx) = acc;
      }
    }
  }
}

int att_datt1(const float* att1, const float* att2_rows, long ldz2, const float* de_rows,
              const float* wf, const int* off, int steps, int B, int P, int A, float* datt1,
              hipStream_t stream) {
  CAPNET_REQUIRE(att1 && att2_rows && de_rows && wf && off && datt1, "att_datt1: null argument");
  CAPNET_REQUIRE(steps > 0 && steps <= kMaxSteps && A % 4 == 0 && ldz2 % 4 == 0, "att_datt1: bad argument");
  AttRowTable rt;
  rt.steps = steps;
  for (int t = 0; t <= steps; ++t) rt.off[t] = off[t];
  int t_chunk = (int)((60 * 1024) / ((size_t)(A + kDatt1Pix) * sizeof(float)));
  CAPNET_REQUIRE(t_chunk >= 1, "att_datt1: A=%d too large", A);
  if (t_chunk > steps) t_chunk = steps;
  const size_t lds = (size_t)t_chunk * (A + kDatt1Pix) * sizeof(float);
  hipLaunchKernelGGL(att_datt1_kernel, dim3(B, cdiv(P, kDatt1Pix)), dim3(128), lds, stream, att1,
                     att2_rows, ldz2, de_rows, wf, rt, P, A, t_chunk, datt1);
  CAPNET_LAUNCH_CHECK();
  return kOk;
}

}  // namespace capnet
