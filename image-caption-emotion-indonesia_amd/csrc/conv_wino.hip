// 3x3 / stride 1 / pad 1 convolution as Winograd F(2x2, 3x3) in fp32 on the matrix cores: the
// conv2 of every ResNet-152 bottleneck outside the stride-2 blocks (45 of the trunk's 50 3x3
// convolutions, 44 % of its conv time with the direct kernel).
//
//   Y = A^T [ sum_c (G g_c G^T) .* (B^T d_c B) ] A          per 2x2 output tile, 4x4 input patch
//
// 16 multiplies per 4 outputs and input channel instead of 36: the implicit-GEMM kernel
// (conv_f32_v2.hip) is bound by the f32 MFMA pipe at 64-67 % of peak on these layers, so doing
// 2.25x fewer matrix flops is the one lever left. Error against fp64: 4.3e-7 rms relative on a
// stage-3 layer, 1.8e-7 for the direct fp32 sum (tests/test_kernels_gpu.py holds both to 3e-6).
//
// One 512-thread workgroup = 64 output tiles (256 pixels) x 64 output channels x all 16
// frequencies; the K loop walks the input channels 8 at a time.
//   * staging (A side): thread (tile, patch row r, channel quad q) loads the 4 pixels of its
//     patch row as float4 (4 channels), applies the previous BatchNorm + ReLU and the zero
//     padding (fma + v_med3, as conv_f32_v2), transforms along the row in registers, and gets
//     the one other row it needs for the column transform from a neighbour lane by DPP
//     (quad_perm): V[r][j] = R_r[j] + s_r * R_o(r)[j]. Row 3 comes out negated; the packed
//     weights carry the same sign flip, so the products are unchanged. Patch offsets and masks
//     are loop invariant: the k-loop's address arithmetic is one scalar add.
//   * weights (B side): transformed once per weight version into 32 KB blocks per (k-tile, 64
//     channels), rows in MFMA lane order, streamed by LDS-DMA (no VGPRs).
//   * MFMA: v_mfma_f32_16x16x4_f32; wave (mh, nw) owns 32 tiles x output channels 16nw..16nw+15
//     for all 16 frequencies (128 accumulator registers; 2 waves per SIMD), so the output
//     transform A^T M A runs in registers with no exchange, and the BatchNorm partial sums need
//     two shuffles. Both LDS images are stored per frequency and 16-row block in MFMA lane order
//     ([k >> 1][row][k & 1]): lane (row i, kq) reads its two k with one ds_read_b64 and a wave
//     reads 512 contiguous bytes (any k <-> (step, kq) bijection is a valid reduction order as
//     long as A and B use the same one).
//   * the k-loop is a software pipeline: one piece of side work (weight DMA of tile k+1, fold and
//     transform of tile k+1, activation loads of tile k+2) between the MFMA groups of each
//     frequency; see the comment at the loop.
// Same contract as conv2d_fwd_v2: raw NHWC output + per-workgroup column sums / sums of squares
// (train-mode BatchNorm), or the folded-BN (+ReLU) inference epilogue.
#include <cstdlib>
#include <type_traits>
#include "common.h"
#include "mfma_core.h"
#include "kernels.h"

namespace capnet {

typedef float w_f32x2 __attribute__((ext_vector_type(2)));

constexpr int WBT = 64;              // output tiles (2x2 pixels each) per workgroup
constexpr int WBN = 64;              // output channels per workgroup
constexpr int WBK = 8;               // input channels per k-tile
constexpr int W_THREADS = 8 * WBT;   // one thread per (tile, patch row, channel quad)
constexpr int W_WAVES = W_THREADS / 64;
constexpr int W_FS_A = WBT * WBK;     // floats per frequency plane of the A image
constexpr int W_FS_B = WBN * WBK;
constexpr int W_A_ST = 16 * W_FS_A;  // floats per A stage
constexpr int W_B_ST = 16 * W_FS_B;
constexpr int W_LDS_FLOATS = 2 * (W_A_ST + W_B_ST);
static_assert(W_WAVES % 4 == 0 && 32 % W_WAVES == 0, "wave grid: (tile halves) x 4 channel groups");

struct WinoArgs {
  const float* x;
  const float* wp;
  float* y;
  const float* in_scale;
  const float* in_shift;
  float* part_sum;
  float* part_sq;
  const float* out_scale;
  const float* out_shift;
  int H, W, C, N;
  int sxb, sxh, sxw;
  int T, TH, TW;  // tiles in all, per column, per row
  int tiles_m, tiles_n;
  int relu_in, relu_out;
  unsigned thw_mul, thw_sh, tw_mul, tw_sh, tn_mul, tn_sh;
};

// ABL: compile-time ablation mask for tools/conv_bench.py (CAPNET_WINO_ABLATE), 0 in the product:
// 1 no ds_read + MFMA, 2 no transform / LDS store, 4 no weight DMA, 8 no activation loads,
// 16 MFMAs without their LDS reads, 32 no barrier in the k-loop, 128 no wait for the weight DMA,
// 256 no wait for the activation loads
template <bool PRE, bool EPI, int ABL = 0>
__global__ __launch_bounds__(W_THREADS) void conv_wino_kernel(WinoArgs g) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // consecutive ids share an A panel (same tm): keep them on one XCD's L2
  const int id = xcd_remap(blockIdx.x, gridDim.x);
  const int tm = (int)fast_div((unsigned)id, g.tn_mul, g.tn_sh), tn = id - tm * g.tiles_n;
  const int m0 = tm * WBT, n0 = tn * WBN;
  const int nk = g.C / WBK;
  const float inf = __builtin_inff();
  const float lo = g.relu_in ? 0.f : -inf;

  // ---- staging geometry: loop invariant ----
  const int r = tid & 3, q = (tid >> 2) & 1, tl = tid >> 3;
  unsigned voff[4];
  float hi[4], lw[4];
  {
    const int t = m0 + tl;
    const bool valid = t < g.T;
    const int tc = valid ? t : g.T - 1;
    const int b = (int)fast_div((unsigned)tc, g.thw_mul, g.thw_sh);
    const int rem = tc - b * (g.TH * g.TW);
    const int th = (int)fast_div((unsigned)rem, g.tw_mul, g.tw_sh);
    const int tw = rem - th * g.TW;
    const int ih = 2 * th - 1 + r;
    const bool row_in = valid && (unsigned)ih < (unsigned)g.H;
    const int ihc = min(max(ih, 0), g.H - 1);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int iw = 2 * tw - 1 + s;
      const bool inb = row_in && (unsigned)iw < (unsigned)g.W;
      const int iwc = min(max(iw, 0), g.W - 1);
      voff[s] = (unsigned)(b * g.sxb + ihc * g.sxh + iwc * g.sxw + 4 * q) * 4u;
      hi[s] = inb ? inf : 0.f;
      lw[s] = inb ? lo : 0.f;
    }
  }
  // column transform: V[r] = R_r + sgn * R_other (row 3 negated, see the header)
  const float sgn = (r == 1) ? 1.f : -1.f;
  // A image: [frequency][16-tile block][kq = k >> 1][tile in block][k & 1], the order in which
  // the MFMA lanes (lane = 16 kq + tile) read it: one contiguous 512 B per ds_read_b64 (a
  // [tile][k] image costs a 4-way bank conflict per read: measured 83 vs 55 us of pure MFMA time)
  // Inside a 16-tile row the tiles of frequency row i = f >> 2 are rotated by 4i: the four patch-
  // row lanes of a quad (which write the four frequency rows) then hit different banks, and the
  // planes can stay 2 KB apart, so every fragment read is one base register + an immediate.
  const int awr = (4 * r) * W_FS_A + (tl >> 4) * 128 + (2 * q * 16 + ((tl + 4 * r) & 15)) * 2;  // + j*FS_A; k pair 2q+1: + 32

  const unsigned lds_b0 = __builtin_amdgcn_readfirstlane(
      (unsigned)(size_t)(__attribute__((address_space(3))) float*)(lds));   // B stages first: their
  // ds_read_b64 immediates (16 bit) then reach every plane; the A reads use the 512-B-unit offsets of ds_read2st64
  const float* wblk = g.wp + (size_t)tn * W_B_ST;          // + kt * tiles_n * W_B_ST
  const size_t wstep = (size_t)g.tiles_n * W_B_ST;

  // ---- pipeline pieces. Everything that is not an MFMA is cut into small pieces and placed
  // BETWEEN the MFMA groups of the k-loop (one piece per frequency step). All eight waves of a
  // workgroup run in lockstep behind the per-k-tile barrier; with the loads issued as one burst
  // after it and the transform as one block in front of it, the matrix pipe idled while every
  // wave sat in the same VMEM / VALU / LDS-write phase (measured: 79 us of MFMA + LDS reads grew
  // to 104 us, and removing the *waits* changed nothing -- it is the issue phases themselves).
  constexpr int DPW = 32 / W_WAVES;  // weight DMA instructions (1 KB each) per wave and k-tile
  f32x4 av[4], scv, shv;
  float d[4][4];  // [pixel s][channel] of the tile being transformed
  auto clampk = [&](int kt) { return kt < nk ? kt : nk - 1; };  // past the end: reload the last tile
  auto dma_b = [&](int stage, int kt, int j) {
    if (ABL & 4) return;
    const float* bs = wblk + (size_t)clampk(kt) * wstep + wave * (DPW * 256);
    const unsigned bdst = lds_b0 + (unsigned)(stage * W_B_ST + wave * (DPW * 256)) * 4u;
    glds16(bs + j * 256, lane * 16, bdst + (unsigned)(j * 256) * 4u);
  };
  auto load_a = [&](int kt, int i) {   // piece i of 0..5
    if (ABL & 8) return;
    const int kc = clampk(kt) * WBK;
    if (i < 4) gload16(av[i], g.x + kc, voff[i]);
    else if (PRE && i == 4) gload16(scv, g.in_scale + kc, (unsigned)(16 * q));
    else if (PRE && i == 5) gload16(shv, g.in_shift + kc, (unsigned)(16 * q));
  };
  constexpr int NLA = PRE ? 6 : 4;     // VMEM loads of one tile's activations per thread
  // the activations of the next tile have landed once at most `keep` younger VMEM ops are in flight
  auto fold = [&](auto keep) {
    if (ABL & 2) return;
    constexpr int KEEP = (ABL & 256) ? 63 : decltype(keep)::value;
    if (PRE)
      asm volatile("s_waitcnt vmcnt(%6)"
                   : "+v"(av[0]), "+v"(av[1]), "+v"(av[2]), "+v"(av[3]), "+v"(scv), "+v"(shv)
                   : "n"(KEEP) : "memory");
    else
      asm volatile("s_waitcnt vmcnt(%4)" : "+v"(av[0]), "+v"(av[1]), "+v"(av[2]), "+v"(av[3]) : "n"(KEEP) : "memory");
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        float v = av[s][c];
        if (PRE) v = fmaf(v, scv[c], shv[c]);
        d[s][c] = __builtin_amdgcn_fmed3f(v, lw[s], hi[s]);
      }
  };
  auto transform = [&](int stage, int j) {   // frequency column j of this thread's patch row
    if (ABL & 2) return;
    float* dst = lds + 2 * W_B_ST + stage * W_A_ST + awr + j * W_FS_A;
    f32x4 o;
#pragma unroll
    for (int c = 0; c < 4; ++c)   // row transform B^T along the patch row
      o[c] = j == 0 ? d[0][c] - d[2][c]
           : j == 1 ? d[1][c] + d[2][c]
           : j == 2 ? d[2][c] - d[1][c]
                    : d[1][c] - d[3][c];
    // column transform through the quad: o += sgn * o[lane of the other row], ONE VALU each
    // (v_fmac with a DPP source, quad_perm [2,2,1,1]: lanes 0,1 of a quad read lane 2, lanes 2,3
    // read lane 1; hipcc emits v_mov_dpp + v_fma for the builtin). s_nop 1: the two wait states a
    // DPP read needs after a VALU write of the same register.
    asm volatile(
        "s_nop 1\n\t"
        "v_fmac_f32_dpp %0, %0, %4 quad_perm:[2,2,1,1] row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %1, %1, %4 quad_perm:[2,2,1,1] row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %2, %2, %4 quad_perm:[2,2,1,1] row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %3, %3, %4 quad_perm:[2,2,1,1] row_mask:0xf bank_mask:0xf"
        : "+v"(o[0]), "+v"(o[1]), "+v"(o[2]), "+v"(o[3])
        : "v"(sgn));
    *reinterpret_cast<w_f32x2*>(dst) = w_f32x2{o[0], o[1]};
    *reinterpret_cast<w_f32x2*>(dst + 32) = w_f32x2{o[2], o[3]};
  };

  f32x4 acc[16][2];
#pragma unroll
  for (int f = 0; f < 16; ++f)
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) acc[f][mb] = f32x4{0.f, 0.f, 0.f, 0.f};

  // wave -> (tile half mh: tiles 32 mh .. 32 mh + 31, channel group nw: channels 16 nw .. + 15)
  const int li = lane & 15, kq = lane >> 4;
  const int mh = wave >> 2, nw = wave & 3;
  int ard[4];                                                          // + f*FS_A + mb*128
#pragma unroll
  for (int i = 0; i < 4; ++i) ard[i] = 2 * mh * 128 + kq * 32 + ((li + 4 * i) & 15) * 2;
  const int brd = nw * 128 + 2 * lane;                                 // + f*FS_B
  using K0 = std::integral_constant<int, 0>;
  using KD = std::integral_constant<int, DPW>;

  // prologue: tile 0 staged, the activations of tile 1 in flight
#pragma unroll
  for (int j = 0; j < DPW; ++j) dma_b(0, 0, j);
#pragma unroll
  for (int i = 0; i < NLA; ++i) load_a(0, i);
  fold(K0{});
#pragma unroll
  for (int j = 0; j < 4; ++j) transform(0, j);
#pragma unroll
  for (int i = 0; i < NLA; ++i) load_a(1, i);
  __syncthreads();

  // iteration kt: MFMAs of tile kt (stage cur); between them the weight DMA of tile kt+1, the
  // transform of tile kt+1 (activations requested one iteration ago) and the activation loads of
  // tile kt+2. ONE instance of the body (run-time stage offsets): unrolled over the stages hipcc
  // gave the two copies different accumulator homes and moved 72 accumulators per iteration.
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1, nxt = cur ^ 1;
    const float* As4[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) As4[i] = lds + 2 * W_B_ST + cur * W_A_ST + ard[i];
    const float* Bs = lds + cur * W_B_ST + brd;
    // fragments of frequency f+2 are requested while the MFMAs of f run (three register sets)
    w_f32x2 a0[3], a1[3], b[3];
    if (!(ABL & 1)) {
#pragma unroll
      for (int f = 0; f < 2; ++f) {
        a0[f] = *reinterpret_cast<const w_f32x2*>(As4[f >> 2] + f * W_FS_A);
        a1[f] = *reinterpret_cast<const w_f32x2*>(As4[f >> 2] + f * W_FS_A + 128);
        b[f] = *reinterpret_cast<const w_f32x2*>(Bs + f * W_FS_B);
      }
    }
#pragma unroll
    for (int f = 0; f < 16; ++f) {
      const int c3 = f % 3, n3 = (f + 2) % 3;
      if (!(ABL & 1) && f + 2 < 16) {
        a0[n3] = *reinterpret_cast<const w_f32x2*>(As4[(f + 2) >> 2] + (f + 2) * W_FS_A);
        a1[n3] = *reinterpret_cast<const w_f32x2*>(As4[(f + 2) >> 2] + (f + 2) * W_FS_A + 128);
        b[n3] = *reinterpret_cast<const w_f32x2*>(Bs + (f + 2) * W_FS_B);
      }
      // one piece of side work per frequency step (VMEM ops in this order: DMA, then loads)
      if (f < DPW) dma_b(nxt, kt + 1, f);
      if (f == DPW) fold(KD{});                       // younger than tile kt+1's loads: the DMAs
      if (f > DPW && f <= DPW + 4) transform(nxt, f - DPW - 1);
      // the loads of tile kt+2 reuse the registers fold() has just read: issued right behind it,
      // a whole iteration before their own fold (10 steps ahead left ~10 us of exposed latency)
      if (f >= DPW && f < DPW + NLA) load_a(kt + 2, f - DPW);
      __builtin_amdgcn_sched_barrier(0);
      if (!(ABL & 1)) {
        acc[f][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[c3].x, b[c3].x, acc[f][0], 0, 0, 0);
        acc[f][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[c3].x, b[c3].x, acc[f][1], 0, 0, 0);
        acc[f][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[c3].y, b[c3].y, acc[f][0], 0, 0, 0);
        acc[f][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[c3].y, b[c3].y, acc[f][1], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    // the weight DMA of tile kt+1 has landed (only the NLA loads of tile kt+2 are younger)
    if (!(ABL & 128)) {
      if (NLA == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    }
    if (!(ABL & 32)) __syncthreads();
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the clamped re-loads of the last iterations

  // ---- output transform + store; D layout of the 16x16 tile: column = lane & 15, rows
  // 4 * (lane >> 4) + e ----
  const int n = n0 + 16 * nw + li;
  float osc = 1.f, osh = 0.f;
  if (EPI) {
    osc = g.out_scale[n];
    osh = g.out_shift[n];
  }
  float cs = 0.f, cq = 0.f;
#pragma unroll
  for (int mb = 0; mb < 2; ++mb)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int t = m0 + 32 * mh + 16 * mb + 4 * kq + e;
      float ta[2][4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float m0j = acc[j][mb][e], m1j = acc[4 + j][mb][e], m2j = acc[8 + j][mb][e],
                    m3j = acc[12 + j][mb][e];
        ta[0][j] = m0j + m1j + m2j;
        ta[1][j] = m1j - m2j - m3j;
      }
      if (t < g.T) {
        const int b = (int)fast_div((unsigned)t, g.thw_mul, g.thw_sh);
        const int rem = t - b * (g.TH * g.TW);
        const int th = (int)fast_div((unsigned)rem, g.tw_mul, g.tw_sh);
        const int tw = rem - th * g.TW;
#pragma unroll
        for (int a = 0; a < 2; ++a) {
          float y0 = ta[a][0] + ta[a][1] + ta[a][2];
          float y1 = ta[a][1] - ta[a][2] - ta[a][3];
          if (EPI) {
            y0 = fmaf(y0, osc, osh);
            y1 = fmaf(y1, osc, osh);
            if (g.relu_out) {
              y0 = fmaxf(y0, 0.f);
              y1 = fmaxf(y1, 0.f);
            }
          } else {
            cs += y0 + y1;
            cq = fmaf(y0, y0, cq);
            cq = fmaf(y1, y1, cq);
          }
          // 32-bit byte offsets from one scalar base (the launcher checks 4*M*N < 4 GB)
          const unsigned off = ((unsigned)((b * g.H + 2 * th + a) * g.W + 2 * tw) * (unsigned)g.N + (unsigned)n) * 4u;
          asm volatile("global_store_dword %0, %1, %2" ::"v"(off), "v"(y0), "s"(g.y) : "memory");
          asm volatile("global_store_dword %0, %1, %2" ::"v"(off + (unsigned)g.N * 4u), "v"(y1), "s"(g.y) : "memory");
        }
      }
    }
  if (!EPI && g.part_sum) {
    cs += __shfl_xor(cs, 16);
    cq += __shfl_xor(cq, 16);
    cs += __shfl_xor(cs, 32);
    cq += __shfl_xor(cq, 32);
    if (kq == 0) {   // one partial row per 32 tiles (tile half of the workgroup)
      const long prow = (long)tm * (WBT / 32) + mh;
      g.part_sum[prow * g.N + n] = cs;
      g.part_sq[prow * g.N + n] = cq;
    }
  }
}

// U = G g G^T per (cout, cin), row 3 negated (see the kernel), written as
// [k-tile = c/8][n-tile = n/64][frequency 4i+j][16-channel block][(c % 8) >> 1][n % 16][c & 1]
// (per frequency and block the order in which the MFMA lanes read it, see the A image)
__global__ __launch_bounds__(256) void wino_pack_kernel(const float* __restrict__ w, float* __restrict__ out,
                                                        int Cout, int Cin) {
  const long total = (long)Cout * Cin;
  const int tiles_n = Cout / WBN;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % Cin), n = (int)(i / Cin);
    const float* g = w + ((long)n * Cin + c) * 9;
    double gg[3][3];
    for (int a = 0; a < 3; ++a)
      for (int b = 0; b < 3; ++b) gg[a][b] = (double)g[a * 3 + b];
    double t[4][3];  // G g
    for (int b = 0; b < 3; ++b) {
      t[0][b] = gg[0][b];
      t[1][b] = 0.5 * (gg[0][b] + gg[1][b] + gg[2][b]);
      t[2][b] = 0.5 * (gg[0][b] - gg[1][b] + gg[2][b]);
      t[3][b] = gg[2][b];
    }
    float* dst = out + ((size_t)(c / WBK) * tiles_n + n / WBN) * W_B_ST + (size_t)((n % WBN) / 16) * 128 +
                 (((c % WBK) >> 1) * 16 + n % 16) * 2 + (c & 1);
    for (int a = 0; a < 4; ++a) {
      const double u0 = t[a][0];
      const double u1 = 0.5 * (t[a][0] + t[a][1] + t[a][2]);
      const double u2 = 0.5 * (t[a][0] - t[a][1] + t[a][2]);
      const double u3 = t[a][2];
      const double s = a == 3 ? -1.0 : 1.0;
      dst[(size_t)(4 * a + 0) * W_FS_B] = (float)(s * u0);
      dst[(size_t)(4 * a + 1) * W_FS_B] = (float)(s * u1);
      dst[(size_t)(4 * a + 2) * W_FS_B] = (float)(s * u2);
      dst[(size_t)(4 * a + 3) * W_FS_B] = (float)(s * u3);
    }
  }
}

size_t conv_wino_weight_floats(int Cin, int Cout) { return (size_t)16 * Cin * Cout; }

// rows of part_sum / part_sq: one per 32 tiles, rounded up to whole workgroups
int conv_wino_tiles_m(int Bn, int H, int W) { return cdiv((long)Bn * (H / 2) * (W / 2), WBT) * (WBT / 32); }

bool conv_wino_shape_ok(int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad) {
  return KH == 3 && KW == 3 && stride == 1 && pad == 1 && H % 2 == 0 && W % 2 == 0 && H >= 2 && W >= 2 &&
         Cin % WBK == 0 && Cout % WBN == 0;
}

int pack_conv_weight_wino(const float* w_oihw, float* out, int Cout, int Cin, hipStream_t stream) {
  CAPNET_REQUIRE(w_oihw && out, "pack_conv_weight_wino: null pointer");
  CAPNET_REQUIRE(Cin % WBK == 0 && Cout % WBN == 0, "pack_conv_weight_wino: Cin %% 8, Cout %% 64 (got %d, %d)",
                 Cin, Cout);
  const long total = (long)Cout * Cin;
  hipLaunchKernelGGL(wino_pack_kernel, dim3((int)(cdiv(total, 256) > 4096 ? 4096 : cdiv(total, 256))),
                     dim3(256), 0, stream, w_oihw, out, Cout, Cin);
  CAPNET_LAUNCH_CHECK();
  return kOk;
}

int conv2d_fwd_wino(const float* x, long sxb, long sxh, long sxw, const float* wp, float* y,
                    const float* in_scale, const float* in_shift, int relu_in, float* part_sum,
                    float* part_sq, int Bn, int H, int W, int Cin, int Cout, hipStream_t stream,
                    const float* out_scale, const float* out_shift, int relu_out) {
  CAPNET_REQUIRE(x && wp && y, "conv2d_fwd_wino: null pointer");
  CAPNET_REQUIRE(conv_wino_shape_ok(H, W, Cin, Cout, 3, 3, 1, 1),
                 "conv2d_fwd_wino: needs even H, W, Cin %% 8 == 0, Cout %% 64 == 0 (got %dx%d, %d -> %d)", H, W,
                 Cin, Cout);
  CAPNET_REQUIRE(Bn > 0 && aligned16(x) && aligned16(wp) && sxw % 4 == 0 && sxh % 4 == 0 && sxb % 4 == 0 &&
                     (long)Bn * sxb * 4 < (1L << 31),
                 "conv2d_fwd_wino: input alignment / size");
  CAPNET_REQUIRE((in_scale == nullptr) == (in_shift == nullptr), "conv2d_fwd_wino: scale/shift pair");
  CAPNET_REQUIRE(!in_scale || (aligned16(in_scale) && aligned16(in_shift)), "conv2d_fwd_wino: scale alignment");
  CAPNET_REQUIRE(in_scale || !relu_in, "conv2d_fwd_wino: relu_in needs a scale/shift prologue");
  CAPNET_REQUIRE((part_sum == nullptr) == (part_sq == nullptr), "conv2d_fwd_wino: stats pair");
  CAPNET_REQUIRE((out_scale == nullptr) == (out_shift == nullptr), "conv2d_fwd_wino: epilogue scale/shift pair");
  CAPNET_REQUIRE(!out_scale || !part_sum, "conv2d_fwd_wino: the folded-BN epilogue produces no statistics");
  CAPNET_REQUIRE(out_scale || !relu_out, "conv2d_fwd_wino: relu_out needs the epilogue");
  WinoArgs g;
  g.x = x; g.wp = wp; g.y = y;
  g.in_scale = in_scale; g.in_shift = in_shift;
  g.part_sum = part_sum; g.part_sq = part_sq;
  g.out_scale = out_scale; g.out_shift = out_shift;
  g.H = H; g.W = W; g.C = Cin; g.N = Cout;
  g.sxb = (int)sxb; g.sxh = (int)sxh; g.sxw = (int)sxw;
  g.TH = H / 2; g.TW = W / 2;
  const long T = (long)Bn * g.TH * g.TW;
  CAPNET_REQUIRE(T < (1L << 24), "conv2d_fwd_wino: too many tiles (%ld)", T);
  CAPNET_REQUIRE((long)Bn * H * W * Cout * 4 < (1L << 32), "conv2d_fwd_wino: output larger than 4 GB");
  g.T = (int)T;
  g.tiles_m = cdiv(T, WBT);
  g.tiles_n = Cout / WBN;
  g.relu_in = relu_in; g.relu_out = relu_out;
  magic_div((unsigned)(g.TH * g.TW), &g.thw_mul, &g.thw_sh);
  magic_div((unsigned)g.TW, &g.tw_mul, &g.tw_sh);
  magic_div((unsigned)g.tiles_n, &g.tn_mul, &g.tn_sh);
  const long wgs = (long)g.tiles_m * g.tiles_n;
  CAPNET_REQUIRE(wgs < (1L << 24), "conv2d_fwd_wino: grid too large");
  const size_t lds_bytes = (size_t)W_LDS_FLOATS * sizeof(float);
  const int variant = (in_scale ? 1 : 0) | (out_scale ? 2 : 0);
  static bool attr_set[4] = {false, false, false, false};
  const void* fn = variant == 0   ? (const void*)conv_wino_kernel<false, false>
                   : variant == 1 ? (const void*)conv_wino_kernel<true, false>
                   : variant == 2 ? (const void*)conv_wino_kernel<false, true>
                                  : (const void*)conv_wino_kernel<true, true>;
  if (!attr_set[variant]) {  // one-time opt-in to > 64 KB of dynamic LDS
    CAPNET_HIP_CHECK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    attr_set[variant] = true;
  }
  const dim3 grid((unsigned)wgs), block(W_THREADS);
  if (const char* e = getenv("CAPNET_WINO_ABLATE")) {   // diagnostics (tools/conv_bench.py): wrong results
    const int abl = atoi(e);
    CAPNET_REQUIRE(variant == 1, "CAPNET_WINO_ABLATE: prologue variant only");
#define CAPNET_WINO_ABL(A)                                                                              \
  case A: {                                                                                             \
    CAPNET_HIP_CHECK(hipFuncSetAttribute((const void*)conv_wino_kernel<true, false, A>,                 \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));  \
    hipLaunchKernelGGL((conv_wino_kernel<true, false, A>), grid, block, lds_bytes, stream, g);          \
    CAPNET_LAUNCH_CHECK();                                                                              \
    return kOk;                                                                                         \
  }
    switch (abl) {
      // MFMA + LDS reads; MFMAs alone (with / without the barrier); no transform; no loads; no MFMA;
      // no waits at all (results are wrong in every one of them)
      CAPNET_WINO_ABL(14) CAPNET_WINO_ABL(30) CAPNET_WINO_ABL(62) CAPNET_WINO_ABL(2) CAPNET_WINO_ABL(12)
      CAPNET_WINO_ABL(1) CAPNET_WINO_ABL(384)
      default: break;
    }
#undef CAPNET_WINO_ABL
  }
  switch (variant) {
    case 0: hipLaunchKernelGGL((conv_wino_kernel<false, false>), grid, block, lds_bytes, stream, g); break;
    case 1: hipLaunchKernelGGL((conv_wino_kernel<true, false>), grid, block, lds_bytes, stream, g); break;
    case 2: hipLaunchKernelGGL((conv_wino_kernel<false, true>), grid, block, lds_bytes, stream, g); break;
    default: hipLaunchKernelGGL((conv_wino_kernel<true, true>), grid, block, lds_bytes, stream, g); break;
  }
  CAPNET_LAUNCH_CHECK();
  return kOk;
}

}  // namespace capnet
