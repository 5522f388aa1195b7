// Fused recurrent LSTM step: gates += h_{t-1} . W^T, activations, c/h update -- ONE launch per
// time step (it replaces a split-K GEMM + slab reduce + pointwise kernel, 3 launches).
//   DecoderFactoredLSTM.forward_step: i,f,o,c~ = act(U(S(V x)) + W h);  c = f c + i c~;  h = o c
//                                     (stylenet/model.py:147-153)
//   nn.LSTMCell:                      gates i,f,g,o;  h = o tanh(c)          (nic/model.py:77)
// Mapping (b = 64, H = 512: 256 workgroups = one per CU, 1024 waves = one per SIMD).
//   A workgroup owns 4 hidden units = 16 gate columns (one N tile of v_mfma_f32_16x16x4_f32), 32
//   of the b rows (two 16-row M tiles) and the whole K = H. Its 4 waves take the 16-wide k groups
//   round-robin; each wave keeps ITS slice of the recurrent weights in registers for the whole
//   launch (H/16 VGPRs: "wavefront-resident" weights, read once per step as fully coalesced 1-KB
//   loads from a fragment-major image), its 32 rows of h_{t-1} are staged once in LDS as 16-B
//   cells (coalesced 16-B global reads along k; one ds_read_b128 feeds four MFMAs), the four
//   K-partial accumulators are summed through LDS and the gate non-linearities, c and h run in
//   the epilogue, whose operands (pre-activations, c_{t-1}) were requested at kernel entry.
// The step is latency-bound, not throughput-bound: its critical path is one HBM round trip for
// h_{t-1} (written by the previous launch on other XCDs), H/16 * 2 dependent MFMAs per wave, one
// LDS exchange and one store. HBM traffic per step = W (4*H*H*4 B) + h,c in/out + pre-activations
// in + gates out: the algorithmic 5.77 MB of SURVEY.md 8(d) at b = 64, H = 512.
#include "common.h"
#include "mfma_core.h"
#include "kernels.h"

namespace capnet {

size_t lstm_wfrag_floats(int H);

typedef float f32x4v __attribute__((ext_vector_type(4)));

constexpr int kStepRows = 64;   // max rows (batch) per step
constexpr int kWgRows = 32;     // rows per workgroup
// LDS image of h_{t-1}: cell[kq][row] = h[row][4kq .. 4kq+3], kq stride (32+1) cells = 132 dwords.
//   writes: ds_write_b128, lanes along kq (coalesced global reads of a row);
//   reads:  ds_read_b128, lanes along rows: lane (i = l & 15, e = l >> 4) of k group g reads
//           cell[4g + e][16 mt + i] and uses its four floats as the A operand of MFMA steps
//           4g .. 4g+3 (k = 16g + 4e + step): any k <-> (step, lane quarter) bijection is a valid
//           reduction order as long as the weight fragments use the same one.
constexpr int kCellStride = kWgRows + 1;

__device__ __forceinline__ float sigm(float x) { return 1.f / (1.f + expf(-x)); }

__host__ __device__ inline int step_ngw(int H) {  // 16-wide k groups per wave, padded to 1/2/4/8
  const int need = (H + 63) / 64;
  return need <= 1 ? 1 : need <= 2 ? 2 : need <= 4 ? 4 : 8;
}

// Workgroup id -> (unit group, row half). Consecutive unit groups share 128-B lines of G, c and
// h, and the two row halves of a unit group share its weights: keep both inside one XCD (the
// dispatcher deals workgroups round-robin over the 8 XCDs, each with its own L2).
__device__ __forceinline__ void step_map(int lid, int nug, int mh_shift, int& ug, int& half) {
  if (nug % 8 == 0) {
    const int xcd = lid & 7, slot = lid >> 3;
    ug = xcd * (nug >> 3) + (slot >> mh_shift);
    half = slot & mh_shift;
  } else {
    ug = lid >> mh_shift;
    half = lid & mh_shift;
  }
}

// Fragment-major copy of the recurrent weights, built once per forward by
// lstm_pack_wfrag_kernel: [unit group ug][wave w][gg][lane][4]; see that kernel.
template <int NGW, int MT>  // NGW: k groups per wave; MT: 16-row M tiles of this launch (1 or 2)
__global__ __launch_bounds__(256) void lstm_step_fused_kernel(
    const float* __restrict__ hprev, const float* __restrict__ Wfrag, float* __restrict__ G,
    const float* __restrict__ cprev, float* __restrict__ c_out, float* __restrict__ h_out, int ldg,
    int b, int H, int cfg, unsigned long long* __restrict__ stamps) {
  // The first 16 dwords of the kernel arguments are preloaded into SGPRs at dispatch
  // (-amdgpu-kernarg-preload-count=16 in the Makefile): everything the loads below need, so the
  // step does not start with a dependent scalar fetch of its own arguments.
  //   cfg: bits 0-1 gi, 2-3 gf, 4-5 go, 6-7 gg (column block of each gate role in G),
  //        bit 8 tanh_out, bit 9 two row halves
  const int gi = cfg & 3, gf = (cfg >> 2) & 3, go = (cfg >> 4) & 3, gg = (cfg >> 6) & 3;
  const int tanh_out = (cfg >> 8) & 1, mh_shift = (cfg >> 9) & 1;
  // stamps != nullptr only in the diagnostic build path (tools/probes/step_phases.py): five s_memtime
  // readings per workgroup, written to a buffer nothing else reads
  unsigned long long ts[5];
  if (stamps) ts[0] = __builtin_amdgcn_s_memtime();
  constexpr int KQP = NGW * 16;  // 16-B cells per row of the (zero padded) image
  extern __shared__ __attribute__((aligned(16))) float lds[];
  f32x4v* cells = reinterpret_cast<f32x4v*>(lds);
  float* red = lds + (size_t)KQP * kCellStride * 4;  // [wave][mt][16 rows][17]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, le = lane >> 4;
  int ug, half;
  step_map(blockIdx.x, H >> 2, mh_shift, ug, half);
  const int u0 = ug * 4, row0 = half * kWgRows;
  const int nrows = min(kWgRows, b - row0);  // >= 1 by construction of the grid
  const int gsel[4] = {gi, gf, go, gg};

  // Loads are issued in the order their consumers run: h (k half 0), W (half 0), h (half 1),
  // W (half 1), epilogue operands. vmcnt retires in order, so the first half's MFMAs start
  // while the second half is still in flight.
  constexpr int NS = NGW >= 2 ? 2 : 1;        // k halves
  constexpr int KQH = KQP / NS;               // cells per row per half
  constexpr int NH = (kWgRows * KQH) / 256;   // cells per thread per half
  constexpr int QH = NGW / NS;                // k groups per wave per half
  const int kq_real = H / 4;
  f32x4v v[NS][NH];
  f32x4v wreg[NGW];
  const f32x4v* wf = reinterpret_cast<const f32x4v*>(Wfrag) + ((long)(ug * 4 + wave) * NGW) * 64 + lane;
#pragma unroll
  for (int hh = 0; hh < NS; ++hh) {
#pragma unroll
    for (int q = 0; q < NH; ++q) {
      const int idx = tid + 256 * q;
      const int row = idx / KQH, kq = hh * KQH + (idx - row * KQH);
      // unconditional load from a clamped cell (a guarded load makes hipcc wait per load)
      v[hh][q] = *reinterpret_cast<const f32x4v*>(
          hprev + (long)(row0 + (row < nrows ? row : nrows - 1)) * H +
          4 * (kq < kq_real ? kq : kq_real - 1));
    }
#pragma unroll
    for (int q = 0; q < QH; ++q) wreg[hh * QH + q] = wf[(long)(hh * QH + q) * 64];
    __builtin_amdgcn_sched_barrier(0);
  }
  // ---- epilogue operands: thread -> (row er, unit eu) ----
  const int er = tid >> 2, eu = tid & 3;
  const bool evalid = tid < 4 * kWgRows && er < nrows;
  const long erow = row0 + (er < nrows ? er : nrows - 1);
  float pre[4], cp;
#pragma unroll
  for (int g = 0; g < 4; ++g) pre[g] = G[erow * ldg + (long)gsel[g] * H + u0 + eu];
  cp = cprev[erow * H + u0 + eu];
  __builtin_amdgcn_sched_barrier(0);

  f32x4v acc[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) acc[mt] = f32x4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int hh = 0; hh < NS; ++hh) {
    // ---- this k half of the workgroup's rows of h_{t-1} -> cell image ----
#pragma unroll
    for (int q = 0; q < NH; ++q) {
      const int idx = tid + 256 * q;
      const int row = idx / KQH, kq = hh * KQH + (idx - row * KQH);
      const float m = (row < nrows && kq < kq_real) ? 1.f : 0.f;
      cells[kq * kCellStride + row] = v[hh][q] * m;
    }
    __syncthreads();
    if (stamps && hh == 0) ts[1] = __builtin_amdgcn_s_memtime();
    // ---- partial products over this wave's k groups (w, w+4, w+8, ...) of the half ----
    f32x4v a[QH][MT];
#pragma unroll
    for (int q = 0; q < QH; ++q)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
        a[q][mt] = cells[(4 * (wave + 4 * (hh * QH + q)) + le) * kCellStride + 16 * mt + li];
#pragma unroll
    for (int q = 0; q < QH; ++q)
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
          acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q][mt][e], wreg[hh * QH + q][e], acc[mt], 0, 0, 0);
  }
  if (stamps) ts[2] = __builtin_amdgcn_s_memtime();
  // D layout of the 16x16 tile: column = lane & 15, rows 4 * (lane >> 4) + r
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int r = 0; r < 4; ++r)
      red[((wave * 2 + mt) * 16 + 4 * le + r) * 17 + li] = acc[mt][r];
  __syncthreads();
  if (stamps) ts[3] = __builtin_amdgcn_s_memtime();
  // ---- epilogue ----
  if (evalid) {
    const int mt = er >> 4, rr = er & 15;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      float s = pre[g];
#pragma unroll
      for (int w = 0; w < 4; ++w) s += red[((w * 2 + mt) * 16 + rr) * 17 + g * 4 + eu];
      pre[g] = s;
    }
    const float i = sigm(pre[0]), f = sigm(pre[1]), og = sigm(pre[2]), gt = tanhf(pre[3]);
    const float c = f * cp + i * gt;
    G[erow * ldg + (long)gi * H + u0 + eu] = i;
    G[erow * ldg + (long)gf * H + u0 + eu] = f;
    G[erow * ldg + (long)go * H + u0 + eu] = og;
    G[erow * ldg + (long)gg * H + u0 + eu] = gt;
    c_out[erow * H + u0 + eu] = c;
    h_out[erow * H + u0 + eu] = tanh_out ? og * tanhf(c) : og * c;
  }
  if (stamps) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    ts[4] = __builtin_amdgcn_s_memtime();
    if (tid == 0)
      for (int k = 0; k < 5; ++k) stamps[blockIdx.x * 5 + k] = ts[k];
  }
}

// Wfrag[(((ug*4 + w)*NGW + q)*64 + lane)*4 + e] = W[row][k] with
//   n = lane & 15: gate role = n >> 2 (role order i,f,o,g; grow[role] = row block of that role in
//   Wcat), unit = ug*4 + (n & 3);  k = 16*(w + 4q) + 4*(lane >> 4) + e  (0 where k >= H)
__global__ __launch_bounds__(256) void lstm_pack_wfrag_kernel(const float* __restrict__ Wcat,
                                                              float* __restrict__ Wfrag, int H,
                                                              int g0, int g1, int g2, int g3) {
  const int ngw = step_ngw(H);
  const long total = (long)(H / 4) * 4 * ngw * 64 * 4;
  const int grow[4] = {g0, g1, g2, g3};
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long)gridDim.x * blockDim.x) {
    const int e = (int)(i & 3);
    const int lane = (int)((i >> 2) & 63);
    long r = i >> 8;
    const int q = (int)(r % ngw); r /= ngw;
    const int w = (int)(r & 3);
    const int ug = (int)(r >> 2);
    const int n = lane & 15;
    const int k = 16 * (w + 4 * q) + 4 * (lane >> 4) + e;
    Wfrag[i] = k < H ? Wcat[((long)grow[n >> 2] * H + ug * 4 + (n & 3)) * H + k] : 0.f;
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

size_t lstm_wfrag_floats(int H) { return (size_t)(H / 4) * 4 * step_ngw(H) * 64 * 4; }

bool lstm_step_fused_supported(int b, int H) {
  return b >= 1 && b <= kStepRows && H >= 16 && H <= 512 && H % 16 == 0;
}

template <int NGW>
static int launch_step(const float* hprev, const float* Wfrag, float* G, long ldg,
                       const float* cprev, float* c_out, float* h_out, int b, int H, int gi, int gf,
                       int go, int gg, int tanh_out, hipStream_t stream,
                       unsigned long long* stamps) {
  const size_t lds_bytes = ((size_t)NGW * 16 * kCellStride * 4 + 4 * 2 * 16 * 17) * sizeof(float);
  const int mh = cdiv(b, kWgRows);
  const bool one_tile = b <= 16;
  auto kern = one_tile ? lstm_step_fused_kernel<NGW, 1> : lstm_step_fused_kernel<NGW, 2>;
  static bool attr_set[2] = {false, false};  // one-time opt-in to > 64 KB of dynamic LDS
  if (lds_bytes > 64 * 1024 && !attr_set[one_tile]) {
    CAPNET_HIP_CHECK(hipFuncSetAttribute((const void*)kern,
                                         hipFuncAttributeMaxDynamicSharedMemorySize,
                                         (int)lds_bytes));
    attr_set[one_tile] = true;
  }
  CAPNET_REQUIRE(ldg < (1l << 31), "lstm_step_fused: ldg too large");
  const int cfg = gi | (gf << 2) | (go << 4) | (gg << 6) | ((tanh_out ? 1 : 0) << 8) | ((mh - 1) << 9);
  hipLaunchKernelGGL(kern, dim3((H / 4) * mh), dim3(256), lds_bytes, stream, hprev, Wfrag, G, cprev,
                     c_out, h_out, (int)ldg, b, H, cfg, stamps);
  CAPNET_LAUNCH_CHECK();
  return kOk;
}

int lstm_step_fused(const float* hprev, const float* Wfrag, float* G, long ldg, const float* cprev,
                    float* c_out, float* h_out, int b, int H, int gi, int gf, int go, int gg,
                    int tanh_out, hipStream_t stream, unsigned long long* stamps) {
  CAPNET_REQUIRE(hprev && Wfrag && G && cprev && c_out && h_out, "lstm_step_fused: null argument");
  CAPNET_REQUIRE(lstm_step_fused_supported(b, H), "lstm_step_fused: unsupported b=%d H=%d", b, H);
  CAPNET_REQUIRE(aligned16(hprev) && aligned16(Wfrag), "lstm_step_fused: alignment");
  switch (step_ngw(H)) {
    case 1: return launch_step<1>(hprev, Wfrag, G, ldg, cprev, c_out, h_out, b, H, gi, gf, go, gg, tanh_out, stream, stamps);
    case 2: return launch_step<2>(hprev, Wfrag, G, ldg, cprev, c_out, h_out, b, H, gi, gf, go, gg, tanh_out, stream, stamps);
    case 4: return launch_step<4>(hprev, Wfrag, G, ldg, cprev, c_out, h_out, b, H, gi, gf, go, gg, tanh_out, stream, stamps);
    default: return launch_step<8>(hprev, Wfrag, G, ldg, cprev, c_out, h_out, b, H, gi, gf, go, gg, tanh_out, stream, stamps);
  }
}

}  // namespace capnet
