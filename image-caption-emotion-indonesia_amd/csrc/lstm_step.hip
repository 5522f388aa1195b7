// Fused recurrent LSTM step: gates += h_{t-1} . W^T, activations, c/h update -- ONE launch per
// time step (it replaces a split-K GEMM + slab reduce + pointwise kernel, 3 launches).
//   DecoderFactoredLSTM.forward_step: i,f,o,c~ = act(U(S(V x)) + W h);  c = f c + i c~;  h = o c
//                                     (stylenet/model.py:147-153)
//   nn.LSTMCell:                      gates i,f,g,o;  h = o tanh(c)          (nic/model.py:77)
// Mapping. A workgroup owns 8 hidden units = 32 gate columns (one 32-wide MFMA N tile), all b <= 64
// rows (two 32-row M tiles) and the whole K = H. Its 4 waves split K; each wave keeps ITS slice of
// the recurrent weights in registers for the whole launch (H/8 VGPRs: "wavefront-resident"
// weights, read from HBM/L2 once per step as 16-B loads of whole 512-B row segments), h_{t-1} is
// staged once, transposed, in LDS (k-major image, conflict-free fragment reads), the four
// K-partial accumulators are summed through LDS and the gate non-linearities run in the epilogue.
// HBM traffic per step = W (4*H*H*4 B) + h,c in/out + pre-activations in + gates out: the
// algorithmic 5.77 MB of SURVEY.md 8(d) at b = 64, H = 512.
#include "common.h"
#include "mfma_core.h"
#include "kernels.h"

namespace capnet {

size_t lstm_wfrag_floats(int H);

constexpr int kStepRows = 64;  // max rows (batch) per step
// LDS image of h_{t-1}: hq[kq][row][4 k] as 16-B cells, kq stride (64+1) cells = 260 dwords.
//   writes: ds_write_b128, lanes along kq (coalesced global reads of a row): 8-lane groups land
//           on banks 0,4,..,28 (+4 dwords per lane) -> conflict-free;
//   reads:  ds_read_b64 of one half of a cell, lanes along rows (MFMA A operand): cells are
//           stored (k0, k0+2 | k0+1, k0+3) so each half-wave's two k-steps are one 8-B word.
constexpr int kCellsPerKq = kStepRows + 1;

__device__ __forceinline__ float sigm(float x) { return 1.f / (1.f + expf(-x)); }

// Fragment-major copy of the recurrent weights, built once per forward by
// lstm_pack_wfrag_kernel: [unit group ug][wave w][q][lane][4] with
//   value(ug, w, q, lane, e) = W[row(n)][k],  n = lane & 31 (gate = n >> 3, unit = ug*8 + (n & 7)),
//   k = w*H/4 + 8q + 2*(e >> 1)*2 ... (see the kernel) -- exactly the B operand of MFMA k-step
//   j = 4q + e of that wave, so a wave loads its slice with H/32 fully coalesced 1-KB reads.
// TWO: b > 32 rows (two 32-row M tiles). A template parameter on purpose: as a runtime branch
// around every second MFMA it made hipcc copy the accumulators through VGPRs each time
// (180 cycles per MFMA instead of 64).
template <int KSTEPS, bool TWO>  // KSTEPS = H / 8 : MFMA k-steps (2 k each) per wave
__global__ __launch_bounds__(256) void lstm_step_fused_kernel(
    const float* __restrict__ hprev, const float* __restrict__ Wfrag, float* __restrict__ G,
    long ldg, const float* __restrict__ cprev, float* __restrict__ c_out,
    float* __restrict__ h_out, int b, int gi, int gf, int go, int gg, int tanh_out,
    unsigned long long* __restrict__ stamps) {
  // stamps != nullptr only in the diagnostic build path (tools/step_phases.py): five s_memtime
  // readings per workgroup, written to a buffer nothing else reads
  unsigned long long ts[5];
  if (stamps) ts[0] = __builtin_amdgcn_s_memtime();
  constexpr int H = KSTEPS * 8;
  constexpr int NQ = KSTEPS / 4 > 0 ? KSTEPS / 4 : 1;  // float4 of weights per lane
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int u0 = blockIdx.x * 8;
  const int gsel[4] = {gi, gf, go, gg};

  // ---- this wave's weight fragments (stay in registers for the whole launch) ----
  float wreg[KSTEPS];
  {
    const float4* wf = reinterpret_cast<const float4*>(Wfrag) +
                       ((long)(blockIdx.x * 4 + wave) * NQ) * 64 + lane;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const float4 v = wf[(long)q * 64];
      wreg[4 * q + 0] = v.x;
      if (KSTEPS > 1) wreg[4 * q + 1] = v.y;
      if (KSTEPS > 2) { wreg[4 * q + 2] = v.z; wreg[4 * q + 3] = v.w; }
    }
  }
  // ---- stage h_{t-1} into the cell image (coalesced 16-B reads along k) ----
  {
    constexpr int KQ = H / 4;                       // 16-B cells per row
    constexpr int NIT = (kStepRows * KQ) / 256;     // cells per thread
    constexpr int BATCH = NIT < 16 ? NIT : 16;
    float4* cells = reinterpret_cast<float4*>(lds);
#pragma unroll
    for (int it0 = 0; it0 < NIT; it0 += BATCH) {
      float4 v[BATCH];
#pragma unroll
      for (int q = 0; q < BATCH; ++q) {
        const int idx = tid + 256 * (it0 + q);
        const int row = idx / KQ, kq = idx - row * KQ;
        // unconditional load from a clamped row (a guarded load makes hipcc wait per load)
        v[q] = *reinterpret_cast<const float4*>(hprev + (long)(row < b ? row : b - 1) * H + 4 * kq);
      }
#pragma unroll
      for (int q = 0; q < BATCH; ++q) {
        const int idx = tid + 256 * (it0 + q);
        const int row = idx / KQ, kq = idx - row * KQ;
        // cell order (k0, k0+2, k0+1, k0+3): the lh = 0 / 1 half-waves then read their two MFMA
        // k-steps as one 8-B word each (no per-MFMA select on the shared VALU/MFMA pipe)
        const float m = row < b ? 1.f : 0.f;
        cells[kq * kCellsPerKq + row] = make_float4(m * v[q].x, m * v[q].z, m * v[q].y, m * v[q].w);
      }
    }
  }
  __syncthreads();
  if (stamps) ts[1] = __builtin_amdgcn_s_memtime();
  // ---- partial products over this wave's K range ----
  f32x16 acc[2];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[mt][r] = 0.f;
  constexpr bool two = TWO;
  {
    const float2* cells = reinterpret_cast<const float2*>(lds) +
                          2 * ((wave * (H / 16)) * kCellsPerKq + li) + lh;
    constexpr int NC = H / 16;  // cells along k in this wave's range (2 MFMA k-steps each)
    constexpr int CB = NC < 4 ? NC : 4;
    float2 c0[CB], c1[CB], n0[CB], n1[CB];
#pragma unroll
    for (int q = 0; q < CB; ++q) {
      c0[q] = cells[2 * (q * kCellsPerKq)];
      if (two) c1[q] = cells[2 * (q * kCellsPerKq + 32)];
    }
#pragma unroll
    for (int cq = 0; cq < NC; cq += CB) {
      if (cq + CB < NC) {
#pragma unroll
        for (int q = 0; q < CB; ++q) {
          n0[q] = cells[2 * ((cq + CB + q) * kCellsPerKq)];
          if (two) n1[q] = cells[2 * ((cq + CB + q) * kCellsPerKq + 32)];
        }
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int q = 0; q < CB; ++q) {
        const int j = 2 * (cq + q);
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(c0[q].x, wreg[j], acc[0], 0, 0, 0);
        if (two) acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(c1[q].x, wreg[j], acc[1], 0, 0, 0);
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(c0[q].y, wreg[j + 1], acc[0], 0, 0, 0);
        if (two) acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(c1[q].y, wreg[j + 1], acc[1], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int q = 0; q < CB; ++q) {
        c0[q] = n0[q];
        if (two) c1[q] = n1[q];
      }
    }
  }
  __syncthreads();  // everyone is done reading the h image: reuse LDS for the K reduction
  if (stamps) ts[2] = __builtin_amdgcn_s_memtime();
  float* red = lds;  // [wave][mt][32 rows][33]
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
      red[((wave * 2 + mt) * 32 + row) * 33 + li] = acc[mt][r];
    }
  __syncthreads();
  if (stamps) ts[3] = __builtin_amdgcn_s_memtime();
  // ---- epilogue: thread -> (row, unit) ----
  for (int o = tid; o < b * 8; o += 256) {
    const int row = o >> 3, uu = o & 7;
    const int mt = row >> 5, rr = row & 31;
    float pre[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      float s = G[(long)row * ldg + (long)gsel[g] * H + u0 + uu];
#pragma unroll
      for (int w = 0; w < 4; ++w) s += red[((w * 2 + mt) * 32 + rr) * 33 + g * 8 + uu];
      pre[g] = s;
    }
    const float i = sigm(pre[0]), f = sigm(pre[1]), og = sigm(pre[2]), gt = tanhf(pre[3]);
    const float cp = cprev[(long)row * H + u0 + uu];
    const float c = f * cp + i * gt;
    G[(long)row * ldg + (long)gi * H + u0 + uu] = i;
    G[(long)row * ldg + (long)gf * H + u0 + uu] = f;
    G[(long)row * ldg + (long)go * H + u0 + uu] = og;
    G[(long)row * ldg + (long)gg * H + u0 + uu] = gt;
    c_out[(long)row * H + u0 + uu] = c;
    h_out[(long)row * H + u0 + uu] = tanh_out ? og * tanhf(c) : og * c;
  }
  if (stamps) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    ts[4] = __builtin_amdgcn_s_memtime();
    if (tid == 0)
      for (int k = 0; k < 5; ++k) stamps[blockIdx.x * 5 + k] = ts[k];
  }
}

// Wfrag[((ug*4 + w)*NQ + q)*64 + lane][e] = W[gsel-independent row][k]:
//   n = lane & 31, gate block = n >> 3 (ROLE order i,f,o,g is applied by the caller through
//   `grow`: grow[role] = row block of that role in Wcat), unit = ug*8 + (n & 7),
//   k = w*H/4 + 2*(4q + e) + (lane >> 5)
__global__ __launch_bounds__(256) void lstm_pack_wfrag_kernel(const float* __restrict__ Wcat,
                                                              float* __restrict__ Wfrag, int H,
                                                              int g0, int g1, int g2, int g3) {
  const int NQ = H / 32 > 0 ? H / 32 : 1;
  const long total = (long)(H / 8) * 4 * NQ * 64 * 4;
  const int ksteps = H / 8;
  const int grow[4] = {g0, g1, g2, g3};
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long)gridDim.x * blockDim.x) {
    const int e = (int)(i & 3);
    const int lane = (int)((i >> 2) & 63);
    long r = i >> 8;
    const int q = (int)(r % NQ); r /= NQ;
    const int w = (int)(r & 3);
    const int ug = (int)(r >> 2);
    const int j = 4 * q + e;
    float v = 0.f;
    if (j < ksteps) {
      const int n = lane & 31, lh = lane >> 5;
      const int k = w * (H / 4) + 2 * j + lh;
      v = Wcat[((long)grow[n >> 3] * H + ug * 8 + (n & 7)) * H + k];
    }
    Wfrag[i] = v;
  }
}

int lstm_pack_wfrag(const float* Wcat, float* Wfrag, int H, int gi, int gf, int go, int gg,
                    hipStream_t stream) {
  CAPNET_REQUIRE(Wcat && Wfrag && H % 16 == 0, "lstm_pack_wfrag: bad argument");
  const long total = (long)lstm_wfrag_floats(H);
  hipLaunchKernelGGL(lstm_pack_wfrag_kernel, dim3((int)(cdiv(total, 256) > 2048 ? 2048 : cdiv(total, 256))),
                     dim3(256), 0, stream, Wcat, Wfrag, H, gi, gf, go, gg);
  CAPNET_LAUNCH_CHECK();
  return kOk;
}

size_t lstm_wfrag_floats(int H) {
  const int NQ = H / 32 > 0 ? H / 32 : 1;
  return (size_t)(H / 8) * 4 * NQ * 64 * 4;
}

bool lstm_step_fused_supported(int b, int H) {
  if (b < 1 || b > kStepRows) return false;
  switch (H) {
    case 16: case 32: case 64: case 96: case 128: case 256: case 512: return true;
    default: return false;
  }
}

int lstm_step_fused(const float* hprev, const float* Wfrag, float* G, long ldg, const float* cprev,
                    float* c_out, float* h_out, int b, int H, int gi, int gf, int go, int gg,
                    int tanh_out, hipStream_t stream, unsigned long long* stamps) {
  CAPNET_REQUIRE(hprev && Wfrag && G && cprev && c_out && h_out, "lstm_step_fused: null argument");
  CAPNET_REQUIRE(lstm_step_fused_supported(b, H), "lstm_step_fused: unsupported b=%d H=%d", b, H);
  CAPNET_REQUIRE(aligned16(hprev) && aligned16(Wfrag), "lstm_step_fused: alignment");
  const size_t lds_bytes =
      std::max((size_t)(H / 4) * kCellsPerKq * 4, (size_t)4 * 2 * 32 * 33) * sizeof(float);
  const dim3 grid(H / 8), block(256);
#define CAPNET_STEP_CASE(HH)                                                                    \
  case HH: {                                                                                    \
    auto kern = b > 32 ? lstm_step_fused_kernel<HH / 8, true> : lstm_step_fused_kernel<HH / 8, false>; \
    static bool attr_set[2] = {false, false}; /* one-time opt-in to > 64 KB of dynamic LDS */   \
    if (lds_bytes > 64 * 1024 && !attr_set[b > 32]) {                                           \
      CAPNET_HIP_CHECK(hipFuncSetAttribute((const void*)kern,                                   \
                                           hipFuncAttributeMaxDynamicSharedMemorySize,          \
                                           (int)lds_bytes));                                    \
      attr_set[b > 32] = true;                                                                  \
    }                                                                                           \
    hipLaunchKernelGGL(kern, grid, block, lds_bytes, stream, hprev, Wfrag, G, ldg, cprev, c_out, \
                       h_out, b, gi, gf, go, gg, tanh_out, stamps);                             \
  } break;
  switch (H) {
    CAPNET_STEP_CASE(16)
    CAPNET_STEP_CASE(32)
    CAPNET_STEP_CASE(64)
    CAPNET_STEP_CASE(96)
    CAPNET_STEP_CASE(128)
    CAPNET_STEP_CASE(256)
    CAPNET_STEP_CASE(512)
    default: break;
  }
#undef CAPNET_STEP_CASE
  CAPNET_LAUNCH_CHECK();
  return kOk;
}

}  // namespace capnet
