// 3x3 / stride 1 / pad 1 convolutions of the trunk on the split-f16 arithmetic of conv_f16x3.hip, with the INPUT PATCH
// of a tile resident in LDS: the implicit GEMM over (tap, channel) of conv_f16x3.hip fetches, folds (BatchNorm + ReLU of
// the producing layer) and splits every input element once per tap -- nine times. Here a workgroup stages, per 32-channel
// chunk, the 128 + 2 W + 2 input pixels its 128 output pixels touch (stride 1: output pixel m sits on input pixel m, tap
// (kh, kw) on pixel m + (kh - 1) W + (kw - 1)) ONCE as split f16 planes, and the nine taps read their A fragments from
// that patch at nine offsets; positions in the zero padding are masked per lane and tap. Per k-step only the packed
// weights (the same image as conv_f16x3.hip's, 16 KB, copied verbatim) still go through registers.
// Eight waves per workgroup, in two arrangements:
//  * 4 x 2 waves of 32 rows x BN / 2 columns at <= 128 VGPRs -- the one used whenever other kernels share the chip
//    (several trunk passes in flight, shared_chip): two such workgroups, or one and a 250-VGPR workgroup of
//    conv_f16x3.hip, fit a CU. It is the slower one ALONE and the faster one in the pipelined step.
//  * KSPLIT (a pass that has the chip to itself, launches of at most one tile per CU: the 14 x 14 and 7 x 7 maps): a lone
//    workgroup per CU whose eight waves are four PAIRS on the 2 x 2 grid of 64 x (BN / 2) blocks; the two waves of a pair
//    split K -- each takes one of the step's two k16 groups -- so the workgroup reads no more LDS than four waves would
//    (the LDS port is as busy as the matrix pipe here). The halves meet once per tile, through LDS, each wave finishing
//    one of the pair's 32-row blocks. 185 VGPRs: beside other kernels it keeps their workgroups off the CU.
// conv1x1_tail_kernel (below): a bottleneck block's tail relu(bn3(y3) + identity) as the staging step of the next
// block's stride-1 1x1 conv1, same wave arrangement and barrier discipline.
#include <cstdlib>

#include "common.h"
#include "mfma_core.h"
#include "kernels.h"

namespace capnet {
namespace {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));

constexpr int PBM = 128;                 // output pixels of a tile
constexpr int kPmax = 248;               // patch pixels: >= 128 + 2 * 56 + 2, = 8 mod 16 (the four channel groups 32 banks apart)
constexpr int kPatchSub = kPmax * 16;    // bytes of one (plane, 8-channel group) array: [pixel][8 halfs]
constexpr int kPatch = 8 * kPatchSub;    // 2 planes x 4 groups
constexpr int kThreads = 512;
constexpr int kPL = (kPmax * 8 + kThreads - 1) / kThreads;   // 16-B loads per thread and chunk (4)
constexpr int kHdrWords = 4;             // as conv_f16x3.hip: [0] ew
static_assert(kPmax % 16 == 8 && kPmax >= PBM + 2 * 56 + 2, "patch size");

struct PArgs {
  const float* x;
  const unsigned* wimg;
  float* y;
  const float* in_scale;
  const float* in_shift;
  float* part_sum;
  float* part_sq;
  int M, Cin, Cout, relu_in, H, W;
  int in_exp;                // the input enters the f16 planes multiplied by 2^in_exp (undone in the epilogue)
  int* err;                  // error word (launches without statistics check their outputs for non-finite values)
  int tiles_m, tiles_n;
  unsigned tn_mul, tn_sh, hw_mul, hw_sh, w_mul, w_sh;
  // conv1x1_tail_kernel: the block tail it absorbs -- input = relu(x * in_scale + in_shift + res (* res_scale + res_shift)),
  // also written to tail_out
  const float* res;
  const float* res_scale;
  const float* res_shift;
  float* tail_out;
};

__device__ __forceinline__ void p_split4(const f32x4 v, h4& h, h4& l) {
  const f2 a = {v[0], v[1]}, b = {v[2], v[3]};
  const h2 ha = __builtin_convertvector(a, h2), hb = __builtin_convertvector(b, h2);      // v_cvt_pk_f16_f32
  const f2 ra = a - __builtin_convertvector(ha, f2), rb = b - __builtin_convertvector(hb, f2);   // exact
  const h2 la = __builtin_convertvector(ra, h2), lb = __builtin_convertvector(rb, h2);
  h = h4{ha[0], ha[1], hb[0], hb[1]};
  l = h4{la[0], la[1], lb[0], lb[1]};
}
// byte offset of weight cell (row n, 8-channel half c) inside one (plane, k16 group) sub-image -- conv_f16x3.hip's h_cell
__device__ __forceinline__ unsigned p_cell(int row, int c) {
  const int r = row & 15;
  return (unsigned)((row * 2 + (c ^ ((r >> 3) & 1))) * 16);
}

// MODE 0: 4 x 2 waves of 32 rows x BN / 2 columns (<= 128 VGPRs); 1 (KSPLIT): four pairs of waves on 64 x BN / 2 blocks, the
// two waves of a pair split K; 2 (BN = 256 only): 2 x 4 waves of 64 x 64 -- one 128 x 256 tile per row tile (the patch is
// staged once instead of once per column tile, 0.67 LDS fragments per MFMA instead of 1; as conv1x1_tail_kernel<256>)
template <int BN, int MODE>
__global__ __launch_bounds__(kThreads, MODE == 0 ? 4 : 2) void conv3x3_patch_kernel(const PArgs g) {
  constexpr bool KSPLIT = MODE == 1, WIDE = MODE == 2;
  static_assert(!WIDE || BN == 256, "the wide arrangement is the 256-column tile's");
  constexpr int WN = WIDE ? 4 : 2;                    // waves (pairs of waves) along the columns
  constexpr int NT = BN / (32 * WN);
  constexpr int MT = MODE == 0 ? 1 : 2;               // 32-row blocks of a wave
  constexpr int kSubB = BN * 2 * 16, kImgB = 4 * kSubB;
  // The packed weights of a k-step (kImgB bytes, copied verbatim) go global -> LDS by LDS-DMA into a ring of kRing
  // buffers, kRing - 1 steps ahead of the MFMAs that read them: a step lasts ~400-500 cycles, a weight fetch from L2
  // ~2 000 under load, and with the weights passing through registers one step ahead (round 2) every step waited for
  // its fetch (1 970 cycles per step measured on the 14 x 14 maps). The DMA needs no registers for data in flight.
  constexpr int kRing = 3;                            // (divides the 9 taps of a chunk: buffer = tap % 3, a constant)
  constexpr int NDMA = kImgB / 1024 / 8;              // 1-KB DMA instructions per wave and step (2 / 1)
  static_assert(NDMA * 8 * 1024 == kImgB, "weight image per wave");
  constexpr int kXch = KSPLIT ? 8 * NT * 16 * 64 * 4 : 0;       // the pairs' exchange at the end of a tile reuses the buffers
  constexpr int kLds = kPatch + kRing * kImgB > kXch ? kPatch + kRing * kImgB : kXch;
  __shared__ __attribute__((aligned(16))) unsigned char lds[kLds];
  // (the column statistics' scratch lies over the patch, which is free by then: with it apart two workgroups would
  //  no longer fit a CU's 160 KB)
  float (*const scratch)[4][BN] = reinterpret_cast<float (*)[4][BN]>(lds);
  static_assert(2 * 4 * BN * 4 <= kPatch, "statistics scratch inside the patch");
  unsigned char* const patch = lds;
  unsigned char* const bbuf = lds + kPatch;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // KSPLIT: wave = (pair on the 2 x 2 grid of 64-row blocks, k16 group kq2); else 4 x 2 waves of 32 rows, both groups
  const int kq2 = KSPLIT ? (wave & 1) : 0;
  const int wm = MODE == 0 ? (wave >> 1) : (wave >> 2), wn = KSPLIT ? ((wave >> 1) & 1) : WIDE ? (wave & 3) : (wave & 1);
  const int rows0 = wm * (32 * MT);                   // this wave's first row of the tile
  const int li = lane & 31, lh = lane >> 5;
  const int W = g.W, Cin = g.Cin;
  const int nkc = Cin / 32, nk = 9 * nkc;
  const int total = g.tiles_m * g.tiles_n, G = (int)gridDim.x;
  const float oscale = ldexpf(1.f, -((int)g.wimg[0] + g.in_exp));
  const float iscale = ldexpf(1.f, g.in_exp);
  const int pq = tid & 7;                             // this thread's 4 channels of a chunk: 4 pq .. 4 pq + 3
  const unsigned char* const a_rd = patch + (kq2 * 2 + lh) * kPatchSub + (rows0 + li) * 16;      // + plane, k16, mt, tap offsets
  const unsigned char* const b_rd = bbuf + p_cell(wn * (BN / WN) + li, lh);
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const unsigned lds_b0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)bbuf);

  for (int wk = (int)blockIdx.x; wk < total; wk += G) {
    const int id = xcd_remap(wk, total);
    const int tm = (int)fast_div((unsigned)id, g.tn_mul, g.tn_sh), tn = id - tm * g.tiles_n;
    const int m0 = tm * PBM, n0 = tn * BN;
    const int pb = m0 - W - 1;                         // input pixel of patch position 0
    const int P = PBM + 2 * W + 2;                     // patch pixels
    // which taps of this lane's two rows fall inside the map
    unsigned vmask[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int m = m0 + rows0 + mt * 32 + li;
      const int mc = m < g.M ? m : g.M - 1;
      const int b = (int)fast_div((unsigned)mc, g.hw_mul, g.hw_sh), rem = mc - b * g.H * W;
      const int oh = (int)fast_div((unsigned)rem, g.w_mul, g.w_sh), ow = rem - oh * W;
      unsigned vm = 0;
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int ih = oh + t / 3 - 1, iw = ow + t % 3 - 1;
        if ((unsigned)ih < (unsigned)g.H && (unsigned)iw < (unsigned)W) vm |= 1u << t;
      }
      vmask[mt] = vm;
    }
    const float* wsrc = reinterpret_cast<const float*>(g.wimg + kHdrWords) + (long)tn * nk * (kImgB / 4);

    f32x4 pre[kPL], fs, ft;
    auto fetch_patch = [&](int c) {
#pragma unroll
      for (int u = 0; u < kPL; ++u) {
        const int px = (tid >> 3) + (kThreads / 8) * u;
        const long p = (long)pb + px;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (px < P && p >= 0 && p < g.M) v = *reinterpret_cast<const f32x4*>(g.x + p * Cin + c * 32 + 4 * pq);
        pre[u] = v;
      }
      if (g.in_scale) {
        fs = *reinterpret_cast<const f32x4*>(g.in_scale + c * 32 + 4 * pq);
        ft = *reinterpret_cast<const f32x4*>(g.in_shift + c * 32 + 4 * pq);
      }
    };
    auto stage_patch = [&]() {
      unsigned char* d = patch + (pq >> 1) * kPatchSub + (pq & 1) * 8;
      // the prescale rides in the fold's scale / shift (8 multiplies per thread and chunk; a power of two: exact)
      const f32x4 fs2 = fs * iscale, ft2 = ft * iscale;
#pragma unroll
      for (int u = 0; u < kPL; ++u) {
        const int px = (tid >> 3) + (kThreads / 8) * u;
        if (px < P) {
          f32x4 v = pre[u];
          if (g.in_scale) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaf(v[e], fs2[e], ft2[e]);
          } else {
            v *= iscale;
          }
          if (g.relu_in) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
          }
          h4 h, l;
          p_split4(v, h, l);
          *reinterpret_cast<h4*>(d + px * 16) = h;
          *reinterpret_cast<h4*>(d + 4 * kPatchSub + px * 16) = l;
        }
      }
    };
    // weights of k-step kt -> ring buffer `buf`: wave w moves bytes [w, w + 1) * NDMA KB of the image
    auto dma_b = [&](int kt, int buf) {
      const float* src = wsrc + (long)kt * (kImgB / 4);
#pragma unroll
      for (int q = 0; q < NDMA; ++q)
        glds16(src, (wave_u * NDMA + q) * 1024 + lane * 16, lds_b0 + (unsigned)(buf * kImgB + (wave_u * NDMA + q) * 1024));
    };

    f32x16 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

    // k-step s = 9 c + t reads weight buffer s % kRing; behind the barrier that opens step s (every wave is then through
    // with step s - 1) the DMA of step s + kRing - 1 is issued into the buffer step s - 1 read.
    auto kt_of = [&](int s2) { const int c2 = s2 / 9; return (s2 - 9 * c2) * nkc + c2; };       // kt = tap * nkc + chunk
    fetch_patch(0);
    __syncthreads();                                   // every wave is through with the previous tile's buffers
#pragma unroll
    for (int r = 0; r < kRing - 1; ++r)
      if (r < nk) dma_b(kt_of(r), r);
    int step = 0;
    for (int c = 0; c < nkc; ++c) {
      if (c > 0) __syncthreads();                      // every wave is through with the previous chunk's patch
      stage_patch();
      if (c + 1 < nkc) fetch_patch(c + 1);
#pragma unroll
      for (int t = 0; t < 9; ++t, ++step) {
        constexpr int kR = kRing;
        const int buf = t % kR;                                       // = step % kRing: 9 % kRing == 0
        // this wave's share of step `step` has landed once at most the DMAs of the kRing - 2 steps behind it are in
        // flight (vmcnt retires in issue order; anything else the wave has in flight only makes the wait longer)
        if (step + kRing - 2 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((kRing - 2) * NDMA) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        // the buffer of step - 1 is free now: the weights of step + kRing - 1 go there
        if (step + kRing - 1 < nk) {
          const int s2 = step + kRing - 1, c2 = s2 / 9;
          dma_b((s2 - 9 * c2) * nkc + c2, (t + kR - 1) % kR);
        }
        const int toff = ((t / 3) * W + t % 3) * 16;                 // patch position of row 0 under this tap
#pragma unroll
        for (int gi = 0; gi < (KSPLIT ? 1 : 2); ++gi) {              // k16 groups of this wave: its own one / both
          const int gsel = KSPLIT ? kq2 : gi;                        // (a_rd already points at group kq2)
          h8 af[MT][2], bf[NT][2];
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) {
            const bool ok = (vmask[mt] >> t) & 1u;
#pragma unroll
            for (int p = 0; p < 2; ++p) {
              const h8 v = *reinterpret_cast<const h8*>(a_rd + (p * 4 + gi * 2) * kPatchSub + mt * 512 + toff);
              const h8 z = {0, 0, 0, 0, 0, 0, 0, 0};
              af[mt][p] = ok ? v : z;
            }
          }
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int p = 0; p < 2; ++p)
              bf[nt][p] = *reinterpret_cast<const h8*>(b_rd + buf * kImgB + (p * 2 + gsel) * kSubB + nt * 1024);
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
              acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[mt][1], bf[nt][0], acc[mt][nt], 0, 0, 0);    // l h'
              acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[mt][0], bf[nt][1], acc[mt][nt], 0, 0, 0);    // h l'
              acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[mt][0], bf[nt][0], acc[mt][nt], 0, 0, 0);    // h h'
            }
        }
      }
    }
    // ---- KSPLIT: the pair's two K halves meet -- wave kq2 finishes row block mt = kq2 and hands the other one over through LDS
    if (KSPLIT) {
      __syncthreads();                                 // patch and weight buffers are free
      float* xch = reinterpret_cast<float*>(lds) + (long)wave * (NT * 16 * 64);       // this wave's outgoing block
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) xch[(nt * 16 + r) * 64 + lane] = kq2 ? acc[0][nt][r] : acc[MT - 1][nt][r];
      __syncthreads();
    }
    // the 32-row blocks this wave finishes: its only one (MODE 0), the pair's block kq2 (KSPLIT), both of its own (WIDE)
    constexpr int NFIN = WIDE ? 2 : 1;
    float cs[NFIN][NT], cq[NFIN][NT];
    {
      const float* xin = reinterpret_cast<const float*>(lds) + (long)(wave ^ 1) * (NT * 16 * 64);
#pragma unroll
      for (int f = 0; f < NFIN; ++f) {
        const int mb = WIDE ? f : kq2;                 // the wave's row block (0 for MODE 0: kq2 = 0)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          cs[f][nt] = 0.f;
          cq[f][nt] = 0.f;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int row = m0 + rows0 + mb * 32 + 4 * lh + (r & 3) + 8 * (r >> 2);
            float v = WIDE ? acc[f][nt][r] : (kq2 ? acc[MT - 1][nt][r] : acc[0][nt][r]);
            if (KSPLIT) v += xin[(nt * 16 + r) * 64 + lane];
            v *= oscale;
            if (row < g.M) {
              g.y[(long)row * g.Cout + n0 + wn * (BN / WN) + nt * 32 + li] = v;
              cs[f][nt] += v;
              cq[f][nt] = fmaf(v, v, cq[f][nt]);
            }
          }
        }
      }
    }
    if (!g.part_sum && g.err) {
      float t = 0.f;
#pragma unroll
      for (int f = 0; f < NFIN; ++f)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) t += cq[f][nt];     // sums of squares of what this thread stored
      flag_nonfinite(t, g.err);
    }
    if (g.part_sum) {
      __syncthreads();                                 // the patch / the pairs' exchange (under the scratch) have been read for the last time
#pragma unroll
      for (int f = 0; f < NFIN; ++f) {
        const int rb = MODE == 0 ? wm : wm * 2 + (WIDE ? f : kq2);       // the tile's four 32-row blocks
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          cs[f][nt] += __shfl_xor(cs[f][nt], 32);
          cq[f][nt] += __shfl_xor(cq[f][nt], 32);
          if (lh == 0) {
            scratch[0][rb][wn * (BN / WN) + nt * 32 + li] = cs[f][nt];
            scratch[1][rb][wn * (BN / WN) + nt * 32 + li] = cq[f][nt];
          }
        }
      }
      __syncthreads();
      if (tid < BN) {
        g.part_sum[(long)tm * g.Cout + n0 + tid] = (scratch[0][0][tid] + scratch[0][1][tid]) + (scratch[0][2][tid] + scratch[0][3][tid]);
        g.part_sq[(long)tm * g.Cout + n0 + tid] = (scratch[1][0][tid] + scratch[1][1][tid]) + (scratch[1][2][tid] + scratch[1][3][tid]);
      }
    }
  }
}

// A bottleneck block's tail fused into the next block's first convolution. The tail -- out = relu(bn3(y3) + identity),
// three passes over the block's widest tensor (bn_add_relu, 40 % of the trunk's memory traffic) -- has one consumer that
// needs every element anyway: the stride-1 1x1 conv1 that follows. This kernel is that conv1 with the tail as its staging
// step: per 32-channel chunk a thread reads its 2 x 4 values of y3 and of the identity (and the two BatchNorms' scale /
// shift), forms out, WRITES it (workgroups of column tile 0 only: the next tail needs it as its identity) and splits it
// into the f16 planes of the A operand. Eight waves at <= 128 VGPRs, both operands double-buffered in LDS, the (tile,
// chunk) steps of a workgroup one stream: operands of step s + 1 are written behind the barrier that opens step s,
// those of step s + 2 fetched then -- also across the end of a tile.
// Two wave arrangements: BN = 64 / 128: 4 x 2 waves of 32 rows x BN / 2 columns (<= 128 VGPRs, two workgroups per CU);
// BN = 256 (conv1 of stage 3, Cout = 256): 2 x 4 waves of 64 x 64 -- ONE column tile, so y3 and the identity are read and
// the tail is formed once per row tile instead of once per column tile (two workgroups read the same 1 MB before), and a
// wave reads 0.67 LDS fragments per MFMA instead of 1 (the eight-wave kernels are LDS-bandwidth-bound); 98
// workgroups of ~150 VGPRs on the 14 x 14 maps: they take 98 CUs for themselves and leave the rest to the other passes.
template <int BN>
__global__ __launch_bounds__(kThreads, BN == 256 ? 2 : 4) void conv1x1_tail_kernel(const PArgs g) {
  constexpr int MT = BN == 256 ? 2 : 1;               // 32-row blocks of a wave
  constexpr int WN = BN == 256 ? 4 : 2;               // waves along the columns
  constexpr int NT = BN / (32 * WN);                  // 32-column blocks of a wave (1 / 2 / 2)
  constexpr int kSubB = BN * 2 * 16, kImgB = 4 * kSubB;
  constexpr int kRing = 3;                            // weight buffers: LDS-DMA, two steps ahead (as conv3x3_patch_kernel)
  constexpr int NDMA = kImgB / 1024 / 8;              // 1-KB DMA instructions per wave and step
  constexpr int kSubA = PBM * 16, kImgA = 8 * kSubA;            // [plane 2][group 4][pixel 128][8 halfs]
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * kImgA + kRing * kImgB];
  static_assert(2 * (8 / WN) * BN * 4 <= kImgA, "the statistics' scratch lies over an A buffer (see the end of a tile)");
  unsigned char* const abuf = lds;
  unsigned char* const bbuf = lds + 2 * kImgA;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = BN == 256 ? wave >> 2 : wave >> 1, wn = BN == 256 ? (wave & 3) : (wave & 1);
  const int li = lane & 31, lh = lane >> 5;
  const int Cin = g.Cin, nkc = Cin / 32, kc_sh = g.H;           // nkc = 1 << kc_sh
  const int total = g.tiles_m * g.tiles_n, G = (int)gridDim.x;
  const int my_tiles = (total - 1 - (int)blockIdx.x) / G + 1;
  const int n_steps = my_tiles * nkc;
  const float oscale = ldexpf(1.f, -((int)g.wimg[0] + g.in_exp));
  const float iscale = ldexpf(1.f, g.in_exp);
  const int pq = tid & 7, ppx = tid >> 3;              // this thread's 4 channels of a chunk and its pixel (and pixel + 64)
  const unsigned char* const a_rd = abuf + lh * kSubA + (wm * 32 * MT + li) * 16;
  const unsigned char* const b_rd = bbuf + p_cell(wn * (BN / WN) + li, lh);
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const unsigned lds_b0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)bbuf);

  // Activations run TWO steps ahead in two register sets (a step of this kernel is ~450 cycles of MFMAs; one step ahead,
  // every step waited ~2 us for its rows of y3 and of the identity: 63 us per launch on the 14 x 14 maps, 32 steps of
  // 2 us); the BatchNorms' scale / shift of the chunk one step ahead (they come from L2).
  // (two NAMED register sets and a step body instantiated per set: a run-time index into a register array makes hipcc
  //  load into a temporary, wait vmcnt(0) and select -- every load completed where it was issued)
  f32x4 preA[2], rsdA[2], preB[2], rsdB[2], fs, ft, gs, gt;
  auto tile_of = [&](int s, int& tm, int& tn) {
    const int id = xcd_remap((int)blockIdx.x + (s >> kc_sh) * G, total);
    tm = (int)fast_div((unsigned)id, g.tn_mul, g.tn_sh);
    tn = id - tm * g.tiles_n;
  };
  auto fetch = [&](int s, f32x4 (&pre)[2], f32x4 (&rsd)[2]) {
    int tm, tn;
    tile_of(s, tm, tn);
    const int c = s & (nkc - 1);
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int m = tm * PBM + ppx + 64 * u;
      const long e = (long)(m < g.M ? m : g.M - 1) * Cin + c * 32 + 4 * pq;        // rows past M: a valid row, never stored
      pre[u] = *reinterpret_cast<const f32x4*>(g.x + e);
      rsd[u] = *reinterpret_cast<const f32x4*>(g.res + e);
    }
  };
  auto fetch_par = [&](int s) {
    const int c = s & (nkc - 1);
    fs = *reinterpret_cast<const f32x4*>(g.in_scale + c * 32 + 4 * pq);
    ft = *reinterpret_cast<const f32x4*>(g.in_shift + c * 32 + 4 * pq);
    if (g.res_scale) {
      gs = *reinterpret_cast<const f32x4*>(g.res_scale + c * 32 + 4 * pq);
      gt = *reinterpret_cast<const f32x4*>(g.res_shift + c * 32 + 4 * pq);
    }
  };
  // weights of step s -> ring buffer s % 3: wave w moves bytes [w, w + 1) * NDMA KB of the (tn, chunk) image
  auto dma_b = [&](int s, int buf) {
    int tm, tn;
    tile_of(s, tm, tn);
    const int c = s & (nkc - 1);
    const float* src = reinterpret_cast<const float*>(g.wimg + kHdrWords) + ((long)tn * nkc + c) * (kImgB / 4);
#pragma unroll
    for (int q = 0; q < NDMA; ++q)
      glds16(src, (wave_u * NDMA + q) * 1024 + lane * 16, lds_b0 + (unsigned)(buf * kImgB + (wave_u * NDMA + q) * 1024));
  };
  // (the register set and the parameter registers hold step s)
  auto stage = [&](int s, const f32x4 (&pre)[2], const f32x4 (&rsd)[2], bool real, int buf) {
    int tm, tn;
    tile_of(s, tm, tn);
    const int c = s & (nkc - 1);
    unsigned char* d = abuf + buf * kImgA + (pq >> 1) * kSubA + (pq & 1) * 8;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      // exactly bn_add_relu_kernel's arithmetic (bn_pool.hip): fma, (fma,) add, max
      f32x4 v, rr = rsd[u];
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = fmaf(pre[u][e], fs[e], ft[e]);
      if (g.res_scale) {
#pragma unroll
        for (int e = 0; e < 4; ++e) rr[e] = fmaf(rr[e], gs[e], gt[e]);
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e] + rr[e], 0.f);
      const int m = tm * PBM + ppx + 64 * u;
      if (real && tn == 0 && m < g.M) *reinterpret_cast<f32x4*>(g.tail_out + (long)m * Cin + c * 32 + 4 * pq) = v;
      h4 h, l;
      p_split4(v * iscale, h, l);                      // (the written tail is unscaled; the A operand carries the prescale)
      *reinterpret_cast<h4*>(d + (ppx + 64 * u) * 16) = h;
      *reinterpret_cast<h4*>(d + 4 * kSubA + (ppx + 64 * u) * 16) = l;
    }
  };

  f32x16 acc[MT][NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;
  dma_b(0, 0);
  if (n_steps > 1) dma_b(1, 1);
  fetch(0, preA, rsdA);
  fetch_par(0);
  fetch(n_steps > 1 ? 1 : 0, preB, rsdB);
  stage(0, preA, rsdA, true, 0);
  fetch_par(n_steps > 1 ? 1 : 0);
  asm volatile("" ::: "memory");
  fetch(n_steps > 2 ? 2 : n_steps - 1, preA, rsdA);
  int rbuf = 0;                                        // s % kRing
  // one step; (pre, rsd) = the register set that holds step s + 1 and then receives step s + 3
  auto body = [&](int s, f32x4 (&pre)[2], f32x4 (&rsd)[2]) {
    const int buf = s & 1;
    // This wave's share of the weights of step s has landed once nothing older than what it issued AFTER that DMA is in
    // flight (vmcnt retires in issue order). Issued after the DMA of step s, at the least: the parameters of step s and
    // the rows of step s + 1 (2 + 4 loads, behind the DMA in step s - 2), the DMA of step s + 1, the parameters of step
    // s + 1 and the rows of step s + 2 (step s - 1). More than that (a second BatchNorm's parameters, the tail's stores,
    // a tile's epilogue) only makes the wait longer, never shorter. (The loads are issued unconditionally, to the very end.)
    if (s + 1 < n_steps) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(12 + NDMA) : "memory");
    else asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    __syncthreads();                                   // step s is in LDS; every wave is through with step s - 1
    // staging first: its waits (the compiler's, for the rows requested two steps ago) then see only the compiler's own
    // younger loads; a DMA issued ahead of it would be counted as one of them and complete rows too early
    // (no branches around the loads: past the end they repeat the last step's -- with loads under conditions hipcc's
    //  wait-count pass merges the paths and falls back to vmcnt(0) in front of every use)
    const int last = n_steps - 1;
    stage(s + 1 < last ? s + 1 : last, pre, rsd, s + 1 <= last, buf ^ 1);      // (past the end: into the buffer nobody reads)
    if (s + 2 <= last) dma_b(s + 2, rbuf == 0 ? kRing - 1 : rbuf - 1);
    fetch_par(s + 2 < last ? s + 2 : last);
    asm volatile("" ::: "memory");                   // parameters BEFORE rows in issue order: waiting for them (next step) must not drain the rows
    fetch(s + 3 < last ? s + 3 : last, pre, rsd);
    const unsigned char* const b_cur = b_rd + rbuf * kImgB;
#pragma unroll
    for (int gq = 0; gq < 2; ++gq) {
      h8 af[MT][2], bf[NT][2];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int p = 0; p < 2; ++p) af[mt][p] = *reinterpret_cast<const h8*>(a_rd + buf * kImgA + (p * 4 + gq * 2) * kSubA + mt * 512);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int p = 0; p < 2; ++p)
          bf[nt][p] = *reinterpret_cast<const h8*>(b_cur + (p * 2 + gq) * kSubB + nt * 1024);
      // term by term over the wave's MT x NT blocks: consecutive MFMAs go to different accumulators
#pragma unroll
      for (int term = 0; term < 3; ++term)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[mt][term == 0 ? 1 : 0], bf[nt][term == 1 ? 1 : 0], acc[mt][nt], 0, 0, 0);   // l h', h l', h h'
    }
    rbuf = rbuf + 1 == kRing ? 0 : rbuf + 1;
    if ((s & (nkc - 1)) == nkc - 1) {
      // ---- end of a tile: 2^-ew, store, column statistics of the rows below M
      int tm, tn;
      tile_of(s, tm, tn);
      const int m0 = tm * PBM, n0 = tn * BN;
      float cs[NT], cq[NT];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        cs[nt] = 0.f;
        cq[nt] = 0.f;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int row = m0 + (wm * MT + mt) * 32 + 4 * lh + (r & 3) + 8 * (r >> 2);
            const float v = acc[mt][nt][r] * oscale;
            acc[mt][nt][r] = 0.f;
            if (row < g.M) {
              g.y[(long)row * g.Cout + n0 + wn * (BN / WN) + nt * 32 + li] = v;
              cs[nt] += v;
              cq[nt] = fmaf(v, v, cq[nt]);
            }
          }
      }
      if (!g.part_sum && g.err) {
        float t = 0.f;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) t += cq[nt];
        flag_nonfinite(t, g.err);
      }
      if (g.part_sum) {
        // scratch = the A buffer this step read: free once every wave is through with the step's MFMAs (the barrier), and
        // staged into again only behind the next step's barrier
        __syncthreads();
        constexpr int WM = 8 / WN;                     // waves along the rows (4 / 2): partial sums per column
        float (*const scratch)[WM][BN] = reinterpret_cast<float (*)[WM][BN]>(abuf + buf * kImgA);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          cs[nt] += __shfl_xor(cs[nt], 32);
          cq[nt] += __shfl_xor(cq[nt], 32);
          if (lh == 0) {
            scratch[0][wm][wn * (BN / WN) + nt * 32 + li] = cs[nt];
            scratch[1][wm][wn * (BN / WN) + nt * 32 + li] = cq[nt];
          }
        }
        __syncthreads();
        if (tid < BN) {
          if (WM == 4) {
            g.part_sum[(long)tm * g.Cout + n0 + tid] = (scratch[0][0][tid] + scratch[0][1][tid]) + (scratch[0][2][tid] + scratch[0][3][tid]);
            g.part_sq[(long)tm * g.Cout + n0 + tid] = (scratch[1][0][tid] + scratch[1][1][tid]) + (scratch[1][2][tid] + scratch[1][3][tid]);
          } else {
            g.part_sum[(long)tm * g.Cout + n0 + tid] = scratch[0][0][tid] + scratch[0][1][tid];
            g.part_sq[(long)tm * g.Cout + n0 + tid] = scratch[1][0][tid] + scratch[1][1][tid];
          }
        }
      }
    }
  };
  for (int s = 0; s < n_steps; s += 2) {
    body(s, preB, rsdB);                               // step s + 1 (odd) lives in set B
    if (s + 1 < n_steps) body(s + 1, preA, rsdA);
  }
}

}  // namespace

bool conv3x3_patch_eligible(const float* x, long sxb, long sxh, long sxw, long sxc, int Bn, int H, int W, int Cin,
                            int Cout, int k, int stride, int pad, const float* in_scale, const float* in_shift) {
  if (!(k == 3 && stride == 1 && pad == 1)) return false;
  return sxc == 1 && sxw == Cin && sxh == (long)W * Cin && sxb == (long)H * W * Cin && Cin % 32 == 0 && Cout % 64 == 0 &&
         W >= 3 && H >= 3 && PBM + 2 * W + 2 <= kPmax && aligned16(x) && (long)Bn * H * W < (1l << 24) &&
         (long)Bn * H * W * (Cin > Cout ? Cin : Cout) < (1l << 31) &&
         (!in_scale || (aligned16(in_scale) && aligned16(in_shift)));
}

// same weight image and tile width as conv_fwd_f16x3 (conv_f16x3_pack, conv1x1_f16x3_bn); statistics rows conv1x1_tiles_m(M).
// shared_chip: other kernels run beside this one (several trunk passes in flight)
int conv3x3_fwd_patch(const float* x, const unsigned* wimg, int bn, float* y, const float* in_scale, const float* in_shift,
                      int relu_in, float* part_sum, float* part_sq, int Bn, int H, int W, int Cin, int Cout,
                      hipStream_t stream, bool shared_chip, int in_exp, int* err) {
  CAPNET_REQUIRE(x && wimg && y && aligned16(wimg) && (bn == 64 || bn == 128 || bn == 256) && Cout % bn == 0 && in_exp > -64 && in_exp < 64,
                 "conv3x3_fwd_patch: bad argument");
  CAPNET_REQUIRE(conv3x3_patch_eligible(x, (long)H * W * Cin, (long)W * Cin, Cin, 1, Bn, H, W, Cin, Cout, 3, 1, 1, in_scale, in_shift),
                 "conv3x3_fwd_patch: operands not eligible");
  CAPNET_REQUIRE((in_scale == nullptr) == (in_shift == nullptr) && (part_sum == nullptr) == (part_sq == nullptr),
                 "conv3x3_fwd_patch: scale / shift and statistics come in pairs");
  PArgs a{};
  a.x = x; a.wimg = wimg; a.y = y; a.in_scale = in_scale; a.in_shift = in_shift; a.part_sum = part_sum; a.part_sq = part_sq;
  a.M = Bn * H * W; a.Cin = Cin; a.Cout = Cout; a.relu_in = relu_in; a.H = H; a.W = W; a.in_exp = in_exp; a.err = err;
  a.tiles_m = cdiv(a.M, PBM); a.tiles_n = Cout / bn;
  magic_div((unsigned)a.tiles_n, &a.tn_mul, &a.tn_sh);
  magic_div((unsigned)(H * W), &a.hw_mul, &a.hw_sh);
  magic_div((unsigned)W, &a.w_mul, &a.w_sh);
  const int cap = 512;
  const int total = a.tiles_m * a.tiles_n;
  const dim3 grid(total <= cap ? total : cap), block(kThreads);
  // at most one tile per CU and nothing else on the chip: a workgroup is alone on its CU, the pairs split K (see the
  // top of the file). Beside other passes' kernels (shared_chip) the <= 128-VGPR arrangement wins although it is the
  // slower one alone: a 185-VGPR workgroup of 8 waves keeps every other conv workgroup off its CU. Measured in the
  // pipelined step: 8 319 images/s without K split, 8 227 with it on the 7 x 7 maps only, 7 981 on 14 x 14 and 7 x 7,
  // 8 097 on the implicit-GEMM kernel.
  const int ks_max = shared_chip ? 0 : 256;
  const bool ksplit = total <= ks_max;
  if (bn == 256) {
    CAPNET_LAUNCH_TIMED((conv3x3_patch_kernel<256, 2>), grid, block, stream, a);
  } else if (bn == 128) {
    if (ksplit) CAPNET_LAUNCH_TIMED((conv3x3_patch_kernel<128, 1>), grid, block, stream, a);
    else CAPNET_LAUNCH_TIMED((conv3x3_patch_kernel<128, 0>), grid, block, stream, a);
  } else {
    if (ksplit) CAPNET_LAUNCH_TIMED((conv3x3_patch_kernel<64, 1>), grid, block, stream, a);
    else CAPNET_LAUNCH_TIMED((conv3x3_patch_kernel<64, 0>), grid, block, stream, a);
  }
  CAPNET_LAUNCH_CHECK();
  return kOk;
}

bool conv1x1_tail_eligible(const float* y3, const float* res, long M, int Cin, int Cout) {
  const int nkc = Cin / 32;
  return Cin % 32 == 0 && nkc > 0 && (nkc & (nkc - 1)) == 0 && Cout % 64 == 0 && aligned16(y3) && aligned16(res) && M > 0 &&
         M < (1l << 24) && M * (Cin > Cout ? Cin : Cout) < (1l << 31);
}

// tail_out [M][Cin] = relu(y3 * s1 + t1 + res (* s2 + t2))   (a bottleneck block's tail, bn_add_relu's arithmetic) and
// y [M][Cout] = tail_out . w^T (the next block's stride-1 1x1 conv1) in one launch; weight image, tile width and
// statistics rows as conv_fwd_f16x3 with k = 1. s2 / t2 null: the identity is used as it is.
int conv1x1_fwd_tail(const float* y3, const float* s1, const float* t1, const float* res, const float* s2, const float* t2,
                     float* tail_out, const unsigned* wimg, int bn, float* y, float* part_sum, float* part_sq, long M,
                     int Cin, int Cout, hipStream_t stream, int in_exp, int* err) {
  CAPNET_REQUIRE(in_exp > -64 && in_exp < 64, "conv1x1_fwd_tail: input exponent %d", in_exp);
  CAPNET_REQUIRE(y3 && s1 && t1 && res && tail_out && wimg && y && aligned16(wimg) && aligned16(tail_out) && aligned16(s1) &&
                     aligned16(t1) && (bn == 64 || bn == 128 || bn == 256) && Cout % bn == 0 && conv1x1_tail_eligible(y3, res, M, Cin, Cout),
                 "conv1x1_fwd_tail: bad argument");
  CAPNET_REQUIRE((s2 == nullptr) == (t2 == nullptr) && (!s2 || (aligned16(s2) && aligned16(t2))) &&
                     (part_sum == nullptr) == (part_sq == nullptr), "conv1x1_fwd_tail: scale / shift and statistics come in pairs");
  PArgs a{};
  a.x = y3; a.in_scale = s1; a.in_shift = t1; a.res = res; a.res_scale = s2; a.res_shift = t2; a.tail_out = tail_out;
  a.wimg = wimg; a.y = y; a.part_sum = part_sum; a.part_sq = part_sq;
  a.M = (int)M; a.Cin = Cin; a.Cout = Cout; a.relu_in = 1; a.W = 1; a.in_exp = in_exp; a.err = err;
  const int nkc = Cin / 32;
  a.H = 0;
  while ((1 << a.H) < nkc) ++a.H;                     // (H carries log2 of the k-steps per tile)
  a.tiles_m = cdiv(a.M, PBM); a.tiles_n = Cout / bn;
  magic_div((unsigned)a.tiles_n, &a.tn_mul, &a.tn_sh);
  const int cap = 512;
  const int total = a.tiles_m * a.tiles_n;
  const dim3 grid(total <= cap ? total : cap), block(kThreads);
  if (bn == 256) CAPNET_LAUNCH_TIMED((conv1x1_tail_kernel<256>), grid, block, stream, a);
  else if (bn == 128) CAPNET_LAUNCH_TIMED((conv1x1_tail_kernel<128>), grid, block, stream, a);
  else CAPNET_LAUNCH_TIMED((conv1x1_tail_kernel<64>), grid, block, stream, a);
  CAPNET_LAUNCH_CHECK();
  return kOk;
}

}  // namespace capnet
