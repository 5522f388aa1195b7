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

__device__ __forceinline__ float wave_sum(float v) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// ---- forward 1: scores + softmax ------------------------------------------------------
// grid = rows of the step. att1 [B][P][A] (indexed by sample), att2 rows at ld `ldz`.
// alpha_out: packed [N][P] row r; alphas_bt: [B][steps][P] user-visible tensor (row (j, t)).
__global__ __launch_bounds__(kAttThreads) void att_scores_fwd_kernel(
    const float* __restrict__ att1, const float* __restrict__ att2, long ldz,
    const float* __restrict__ wf, const float* __restrict__ bf, int P, int A,
    float* __restrict__ alpha_out, float* __restrict__ alphas_bt, int steps, int t) {
  extern __shared__ float sh[];  // e[P]
  __shared__ float red[8];
  const int j = blockIdx.x;      // row inside the step == sample index
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float* a1 = att1 + (long)j * P * A;
  const float* a2 = att2 + (long)j * ldz;
  for (int p = wave; p < P; p += kAttThreads / 64) {
    float s = 0.f;
    for (int a = lane * 4; a < A; a += 256) {
      const float4 x = *reinterpret_cast<const float4*>(a1 + (long)p * A + a);
      const float4 y = *reinterpret_cast<const float4*>(a2 + a);
      const float4 w = *reinterpret_cast<const float4*>(wf + a);
      s = fmaf(fmaxf(x.x + y.x, 0.f), w.x, s);
      s = fmaf(fmaxf(x.y + y.y, 0.f), w.y, s);
      s = fmaf(fmaxf(x.z + y.z, 0.f), w.z, s);
      s = fmaf(fmaxf(x.w + y.w, 0.f), w.w, s);
    }
    s = wave_sum(s);
    if (lane == 0) sh[p] = s + bf[0];
  }
  __syncthreads();
  float m = -INFINITY;
  for (int p = threadIdx.x; p < P; p += kAttThreads) m = fmaxf(m, sh[p]);
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  if (lane == 0) red[wave] = m;
  __syncthreads();
  m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  float z = 0.f;
  for (int p = threadIdx.x; p < P; p += kAttThreads) {
    const float e = expf(sh[p] - m);
    sh[p] = e;
    z += e;
  }
  z = wave_sum(z);
  if (lane == 0) red[4 + wave] = z;
  __syncthreads();
  z = red[4] + red[5] + red[6] + red[7];
  const float inv = 1.f / z;
  for (int p = threadIdx.x; p < P; p += kAttThreads) {
    const float al = sh[p] * inv;
    alpha_out[(long)j * P + p] = al;
    alphas_bt[((long)j * steps + t) * P + p] = al;
  }
}

// ---- forward 2: context vector + gate ------------------------------------------------------
// grid = (rows, C/512); 128 threads x float4. gate_io: in = f_beta(h) pre-activation (ld ldz),
// out = sigmoid of it (saved for backward). awe_out [rows][C] (pre-gate, saved), xa_out: the
// decoder input slice (ld ldx) = gate * awe.
__global__ __launch_bounds__(128) void att_context_fwd_kernel(
    const float* __restrict__ feat, const float* __restrict__ alpha, int P, int C,
    float* __restrict__ gate_io, long ldz, float* __restrict__ awe_out,
    float* __restrict__ xa_out, long ldx) {
  extern __shared__ float al[];  // alpha[P]
  const int j = blockIdx.x;
  const int c = blockIdx.y * 512 + threadIdx.x * 4;
  for (int p = threadIdx.x; p < P; p += 128) al[p] = alpha[(long)j * P + p];
  __syncthreads();
  const float* f = feat + (long)j * P * C + c;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 4
  for (int p = 0; p < P; ++p) {
    const float4 v = *reinterpret_cast<const float4*>(f + (long)p * C);
    const float a = al[p];
    s.x = fmaf(a, v.x, s.x); s.y = fmaf(a, v.y, s.y); s.z = fmaf(a, v.z, s.z); s.w = fmaf(a, v.w, s.w);
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
                 float* xa_out, long ldx, hipStream_t stream) {
  if (rows <= 0) return kOk;
  CAPNET_REQUIRE(att1 && feat && att2 && gate_io && wf && bf && alpha_out && alphas_bt && awe_out &&
                     xa_out, "att_step_fwd: null argument");
  CAPNET_REQUIRE(A % 4 == 0 && C % 512 == 0 && P > 0 && P <= 4096 && ldz % 4 == 0 && ldx % 4 == 0,
                 "att_step_fwd: A=%d C=%d P=%d (need A%%4==0, C%%512==0)", A, C, P);
  hipLaunchKernelGGL(att_scores_fwd_kernel, dim3(rows), dim3(kAttThreads), P * sizeof(float), stream,
                     att1, att2, ldz, wf, bf, P, A, alpha_out, alphas_bt, steps, t);
  hipLaunchKernelGGL(att_context_fwd_kernel, dim3(rows, C / 512), dim3(128), P * sizeof(float),
                     stream, feat, alpha_out, P, C, gate_io, ldz, awe_out, xa_out, ldx);
  CAPNET_LAUNCH_CHECK();
  return kOk;
}

// ---- backward 1: gate and d(alpha) partials -----------------------------------------------
// grid = (rows, C/512), 256 threads = 4 waves; a wave owns a pixel at a time, its lanes 512
// channels (2 x float4). dxa: gradient of the gated context (ld ldx). Outputs:
//   dgate_out (ld ldz) = d f_beta pre-activation; dalpha_part [rows][C/512][P].
__global__ __launch_bounds__(kAttThreads) void att_context_bwd_kernel(
    const float* __restrict__ feat, const float* __restrict__ dxa, long ldx,
    const float* __restrict__ gate, long ldzg, const float* __restrict__ awe, int P, int C,
    float* __restrict__ dgate_out, long ldz, float* __restrict__ dalpha_part) {
  const int j = blockIdx.x, cb = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = cb * 512 + lane * 8;
  float d[8];
  {
    const float* gx = dxa + (long)j * ldx + c;
    const float* gg = gate + (long)j * ldzg + c;
    const float* aw = awe + (long)j * C + c;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float g = gg[k], dg = gx[k];
      d[k] = dg * g;  // d(awe)
      if (wave == 0) dgate_out[(long)j * ldz + c + k] = dg * aw[k] * g * (1.f - g);
    }
  }
  const float* f = feat + (long)j * P * C + c;
  for (int p = wave; p < P; p += kAttThreads / 64) {
    const float4 v0 = *reinterpret_cast<const float4*>(f + (long)p * C);
    const float4 v1 = *reinterpret_cast<const float4*>(f + (long)p * C + 4);
    float s = d[0] * v0.x + d[1] * v0.y + d[2] * v0.z + d[3] * v0.w + d[4] * v1.x + d[5] * v1.y +
              d[6] * v1.z + d[7] * v1.w;
    s = wave_sum(s);
    if (lane == 0) dalpha_part[((long)j * gridDim.y + cb) * P + p] = s;
  }
}

// ---- backward 2: softmax, relu, full_att / decoder_att / encoder_att gradients --------------
// grid = rows. dalphas_bt: gradient of the user-visible alphas tensor [B][steps][P] (may be null).
// Outputs: datt2 (ld ldz), datt1_acc [B][P][A] += , dwf_rows [rows][A], dbf_rows [rows].
__global__ __launch_bounds__(kAttThreads) void att_scores_bwd_kernel(
    const float* __restrict__ att1, const float* __restrict__ att2, long ldz2,
    const float* __restrict__ wf, const float* __restrict__ alpha,
    const float* __restrict__ dalpha_part, int nparts, const float* __restrict__ dalphas_bt,
    int steps, int t, int P, int A, float* __restrict__ datt2, long ldz,
    float* __restrict__ datt1_acc, float* __restrict__ dwf_rows, float* __restrict__ dbf_rows) {
  extern __shared__ float sh[];  // de[P], then cross-wave scratch [4][A] x 2
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
  }
  __syncthreads();
  dbf = wave_sum(dbf);
  if (lane == 0) red[wave] = dbf;
  __syncthreads();
  if (threadIdx.x == 0) dbf_rows[j] = red[0] + red[1] + red[2] + red[3];

  // each wave sweeps pixels wave, wave+4, ...; lanes own channels a = lane*4 + 256*i
  const float* a1 = att1 + (long)j * P * A;
  const float* a2 = att2 + (long)j * ldz2;
  float* d1 = datt1_acc + (long)j * P * A;
  float* s_d2 = sh + P;           // [4][A]
  float* s_dw = sh + P + 4 * A;   // [4][A]
  for (int a0 = lane * 4; a0 < A; a0 += 256) {
    const float4 y = *reinterpret_cast<const float4*>(a2 + a0);
    const float4 w = *reinterpret_cast<const float4*>(wf + a0);
    float4 g2 = make_float4(0.f, 0.f, 0.f, 0.f), gw = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int p = wave; p < P; p += kAttThreads / 64) {
      const float e = de[p];
      const float4 x = *reinterpret_cast<const float4*>(a1 + (long)p * A + a0);
      float4 acc = *reinterpret_cast<const float4*>(d1 + (long)p * A + a0);
      const float z0 = x.x + y.x, z1 = x.y + y.y, z2 = x.z + y.z, z3 = x.w + y.w;
      const float q0 = z0 > 0.f ? e * w.x : 0.f, q1 = z1 > 0.f ? e * w.y : 0.f;
      const float q2 = z2 > 0.f ? e * w.z : 0.f, q3 = z3 > 0.f ? e * w.w : 0.f;
      acc.x += q0; acc.y += q1; acc.z += q2; acc.w += q3;
      *reinterpret_cast<float4*>(d1 + (long)p * A + a0) = acc;
      g2.x += q0; g2.y += q1; g2.z += q2; g2.w += q3;
      gw.x = fmaf(e, fmaxf(z0, 0.f), gw.x); gw.y = fmaf(e, fmaxf(z1, 0.f), gw.y);
      gw.z = fmaf(e, fmaxf(z2, 0.f), gw.z); gw.w = fmaf(e, fmaxf(z3, 0.f), gw.w);
    }
    *reinterpret_cast<float4*>(s_d2 + wave * A + a0) = g2;
    *reinterpret_cast<float4*>(s_dw + wave * A + a0) = gw;
  }
  __syncthreads();
  for (int a = threadIdx.x; a < A; a += kAttThreads) {
    datt2[(long)j * ldz + a] = s_d2[a] + s_d2[A + a] + s_d2[2 * A + a] + s_d2[3 * A + a];
    dwf_rows[(long)j * A + a] = s_dw[a] + s_dw[A + a] + s_dw[2 * A + a] + s_dw[3 * A + a];
  }
}

int att_step_bwd(const float* att1, const float* feat, const float* att2, long ldz2,
                 const float* gate, long ldzg, const float* awe, const float* alpha,
                 const float* wf, const float* dxa, long ldx, const float* dalphas_bt, int steps,
                 int t, int rows, int P, int A, int C, float* dalpha_part, float* dgate_out,
                 float* datt2, long ldz, float* datt1_acc, float* dwf_rows, float* dbf_rows,
                 hipStream_t stream) {
  if (rows <= 0) return kOk;
  CAPNET_REQUIRE(att1 && feat && att2 && gate && awe && alpha && wf && dxa && dalpha_part &&
                     dgate_out && datt2 && datt1_acc && dwf_rows && dbf_rows,
                 "att_step_bwd: null argument");
  CAPNET_REQUIRE(A % 4 == 0 && C % 512 == 0 && P > 0 && ldz % 4 == 0 && ldx % 4 == 0 &&
                     (size_t)(P + 8 * A) * 4 <= 64 * 1024,
                 "att_step_bwd: A=%d C=%d P=%d", A, C, P);
  // gate rows use their own leading dimension (saved forward Z buffer)
  hipLaunchKernelGGL(att_context_bwd_kernel, dim3(rows, C / 512), dim3(kAttThreads), 0, stream, feat,
                     dxa, ldx, gate, ldzg, awe, P, C, dgate_out, ldz, dalpha_part);
  hipLaunchKernelGGL(att_scores_bwd_kernel, dim3(rows), dim3(kAttThreads),
                     (P + 8 * A) * sizeof(float), stream, att1, att2, ldz2, wf, alpha, dalpha_part,
                     C / 512, dalphas_bt, steps, t, P, A, datt2, ldz, datt1_acc, dwf_rows, dbf_rows);
  CAPNET_LAUNCH_CHECK();
  return kOk;
}

}  // namespace capnet
