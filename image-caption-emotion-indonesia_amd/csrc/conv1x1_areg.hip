// 1x1 convolutions with a short K (Cin = 64 / 128 / 256: conv3 of the bottlenecks of stages 1-3, torchvision
// Bottleneck.conv3 via stylenet/model.py:15-18,24) with the A OPERAND RESIDENT IN REGISTERS.
//
// The tiled kernels (conv_f16x3.hip) fetch, fold (BatchNorm + ReLU of the producing layer) and split a 128-row A tile
// once per 128-column output tile: for conv3 of stage 3 (256 -> 1024 channels) eight times, which is exactly its 7.8
// VALU instructions per MFMA (profiles/round2_sq_conv_f16x3.csv). Here a workgroup owns 128 rows for ALL output
// columns. Each of its four waves keeps its 32 rows, folded and split ONCE, as MFMA A fragments in registers for the
// whole K (K = 256: 16 k16 groups x 2 planes x 4 VGPRs = 128 VGPRs) and sweeps the output columns 32 at a time; only
// the packed weights stream through LDS (the image of conv_f16x3_pack, copied verbatim in 1-KB pieces). Consequences:
//   * staging VALU per MFMA: ~0.3 instead of 7.8 -- nothing but MFMAs, two ds_read_b128 per three MFMAs and the
//     epilogue in the column loop; A never touches LDS;
//   * one barrier per 32 columns (48 MFMAs per wave at K = 256), none inside;
//   * 98 workgroups of four waves for the 14 x 14 maps at batch 64: the launch occupies 98 CUs' worth of one wave per
//     SIMD and leaves the rest of the chip to the other passes' kernels (in the pipelined step a kernel is
//     worth what it leaves room for).
// Arithmetic: conv_f16x3.hip's -- x 2^ea = h + l in f16, weights 2^ew-scaled and split at pack time, the three products
// l h', h l', h h' accumulated in fp32 by v_mfma_f32_32x32x16_f16, accumulators x 2^-(ew + ea) in the epilogue.
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

constexpr int RBM = 128;                 // rows of a workgroup: 4 waves x 32
constexpr int RNB = 32;                  // output columns of one sweep step (a "granule" of weights: 32 columns x K)
constexpr int kHdrWords = 4;             // conv_f16x3.hip's image header: [0] ew

struct RArgs {
  const float* x;            // [M][Cin] dense
  const unsigned* wimg;      // conv_f16x3_pack image for tile width bn
  float* y;                  // [M][Cout] raw
  const float* in_scale;
  const float* in_shift;
  float* part_sum;           // [ceil(M / 128)][Cout]
  float* part_sq;
  int M, Cout, bn, relu_in, in_exp;
  int* err;                  // error word (launches without statistics check their outputs for non-finite values)
};

__device__ __forceinline__ void r_split4(const f32x4 v, h4& h, h4& l) {
  const f2 a = {v[0], v[1]}, b = {v[2], v[3]};
  const h2 ha = __builtin_convertvector(a, h2), hb = __builtin_convertvector(b, h2);      // v_cvt_pk_f16_f32
  const f2 ra = a - __builtin_convertvector(ha, f2), rb = b - __builtin_convertvector(hb, f2);   // exact
  const h2 la = __builtin_convertvector(ra, h2), lb = __builtin_convertvector(rb, h2);
  h = h4{ha[0], ha[1], hb[0], hb[1]};
  l = h4{la[0], la[1], lb[0], lb[1]};
}
// byte offset of weight cell (column n of the granule, 8-channel half c) inside one 1-KB piece -- conv_f16x3.hip's h_cell
__device__ __forceinline__ unsigned r_cell(int row, int c) {
  const int r = row & 15;
  return (unsigned)((row * 2 + (c ^ ((r >> 3) & 1))) * 16);
}

template <int KG, bool PRE>
__global__ __launch_bounds__(256) void conv1x1_areg_kernel(const RArgs g) {
  constexpr int K = 32 * KG, NK16 = 2 * KG;
  constexpr int kGran = KG * 4096;                  // bytes of a granule: KG k-steps x (2 planes x 2 k16 groups) x 1 KB
  constexpr int kRing = 3;                          // weight granules in LDS: one being multiplied, two on their way
  constexpr int NDMA = KG;                          // 1-KB LDS-DMA instructions per wave and granule
  __shared__ __attribute__((aligned(16))) unsigned char bbuf[kRing * kGran];
  __shared__ __attribute__((aligned(16))) float fold[PRE ? 2 * K : 4];
  __shared__ float scratch[2][2][4][RNB];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int M = g.M, Cout = g.Cout;
  const int m0 = (int)blockIdx.x * RBM;
  const int n_gran = Cout / RNB, gran_per_tn = g.bn / RNB, nk = KG;
  const int ew = (int)g.wimg[0];
  const float oscale = ldexpf(1.f, -(ew + g.in_exp));
  const unsigned char* const wbase = reinterpret_cast<const unsigned char*>(g.wimg + kHdrWords);
  const long sub_bytes = (long)g.bn * 32;            // one (plane, k16 group) sub-image of a (tn, kt) step

  // ---- weights: a granule = KG k-steps x 4 pieces (plane, k16 group) of 1 KB, LDS image [k-step][piece][1 KB]. Wave w
  // moves pieces w KG .. w KG + KG - 1 by LDS-DMA (no registers for data in flight), two granules ahead of the MFMAs:
  // a granule lasts ~1 500 cycles, a fetch from L2 ~2 000 under load -- through registers, half a granule ahead, every
  // granule waited twice for its weights (61 us on the 14 x 14 maps; tools/probes/areg_bench.py).
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const unsigned lds_b0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)bbuf);
  auto dma_b = [&](int nb, int buf) {
    const int tn = nb / gran_per_tn, sub = nb - tn * gran_per_tn;
#pragma unroll
    for (int i = 0; i < NDMA; ++i) {
      const int pc = wave_u * NDMA + i;             // = k-step * 4 + piece
      const unsigned char* src = wbase + ((long)tn * nk * 4 + pc) * sub_bytes + sub * 1024;
      glds16(reinterpret_cast<const float*>(src), lane * 16, lds_b0 + (unsigned)(buf * kGran + pc * 1024));
    }
  };
  dma_b(0, 0);
  if (n_gran > 1) dma_b(1, 1);

  // ---- A: this lane's row, k = 16 kk + 8 lh .. + 7 of every k16 group kk, folded and split once
  if (PRE) {
    const float es = ldexpf(1.f, g.in_exp);
    for (int i = tid; i < K; i += 256) {
      fold[i] = g.in_scale[i] * es;                 // (power of two: exact)
      fold[K + i] = g.in_shift[i] * es;
    }
  }
  const bool full = m0 + RBM <= M;                  // (uniform) every row of this workgroup exists
  const int row = m0 + wave * 32 + li;
  const float* xr = g.x + (long)(row < M ? row : M - 1) * K + 8 * lh;     // rows past M: a valid row, never stored
  h8 ah[NK16], al[NK16];
  __syncthreads();                                  // fold[] is in LDS
  const float esx = ldexpf(1.f, g.in_exp);
  const float lo = g.relu_in ? 0.f : -__builtin_inff();
  constexpr int KB = 2;                             // k16 groups per batch of loads (registers: 8 + 16 of scale / shift per group)
#pragma unroll
  for (int b0 = 0; b0 < NK16; b0 += KB) {
    f32x4 raw[KB][2];
#pragma unroll
    for (int j = 0; j < KB; ++j) {
      raw[j][0] = *reinterpret_cast<const f32x4*>(xr + 16 * (b0 + j));
      raw[j][1] = *reinterpret_cast<const f32x4*>(xr + 16 * (b0 + j) + 4);
    }
#pragma unroll
    for (int j = 0; j < KB; ++j) {
      const int kk = b0 + j;
      h4 hh[2], ll[2];
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        f32x4 v = raw[j][q];
        if (PRE) {
          const f32x4 s = *reinterpret_cast<const f32x4*>(fold + 16 * kk + 8 * lh + 4 * q);
          const f32x4 t = *reinterpret_cast<const f32x4*>(fold + K + 16 * kk + 8 * lh + 4 * q);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = fmaf(v[e], s[e], t[e]);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] *= esx;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], lo);
        r_split4(v, hh[q], ll[q]);
      }
      ah[kk] = h8{hh[0][0], hh[0][1], hh[0][2], hh[0][3], hh[1][0], hh[1][1], hh[1][2], hh[1][3]};
      al[kk] = h8{ll[0][0], ll[0][1], ll[0][2], ll[0][3], ll[1][0], ll[1][1], ll[1][2], ll[1][3]};
      // finished HERE: left to itself hipcc sinks the fold + split of the later groups into the first sweep step,
      // behind its first MFMAs, and spills the raw loads on the way
      asm volatile("" : "+v"(ah[kk]), "+v"(al[kk]));
    }
    __builtin_amdgcn_sched_barrier(0);              // one batch of loads in flight at a time
  }

  const unsigned char* const b_rd = bbuf + r_cell(li, lh);
  // ---- the column sweep: granule nb is in ring buffer nb % 3. Behind the barrier that opens it (every wave is then
  // through with granule nb - 1) the DMA of granule nb + 2 goes into the buffer granule nb - 1 occupied.
  // Waits are counted: what this wave issued after the DMA of granule nb is, at the top of iteration nb, the 16 stores of
  // granule nb - 2's epilogue, the NDMA instructions of granule nb + 1 and the 16 stores of granule nb - 1 (full tiles:
  // every store unconditional; otherwise only the DMAs are counted on -- a smaller count only waits longer).
  int buf = 0;
  for (int nb = 0; nb < n_gran; ++nb) {
    if (nb + 1 < n_gran) {
      if (full && nb >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA + 32) : "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA) : "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();                                // granule nb is in LDS; every wave is through with granule nb - 1
    if (nb + 2 < n_gran) dma_b(nb + 2, buf == 0 ? 2 : buf - 1);
    if (nb > 0 && g.part_sum && tid < RNB) {
      // column statistics of the previous granule: the four waves' partial sums
      const float (*sc)[4][RNB] = scratch[(nb - 1) & 1];
      const long o = (long)blockIdx.x * Cout + (nb - 1) * RNB + tid;
      g.part_sum[o] = (sc[0][0][tid] + sc[0][1][tid]) + (sc[0][2][tid] + sc[0][3][tid]);
      g.part_sq[o] = (sc[1][0][tid] + sc[1][1][tid]) + (sc[1][2][tid] + sc[1][3][tid]);
    }
    // Three accumulators, one per product term: an MFMA that accumulates into the result of the one just issued waits
    // for it (~3 issue slots of this shape, measured: one chain of 48 ran at a third of the rate), three chains do not.
    f32x16 acc, acc1, acc2;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = acc1[r] = acc2[r] = 0.f;
    // B fragments run kAhead k16 groups ahead of the MFMAs that use them: with one wave per SIMD nothing else hides the
    // LDS latency (~250 cycles beside three other waves' reads against 96 cycles of MFMAs per group: one group ahead the
    // sweep ran at 4 000 cycles per granule, LDS-latency-bound, whatever else was switched off)
    constexpr int kAhead = NK16 < 4 ? NK16 - 1 : 3, kSets = kAhead + 1;
    h8 bh[kSets], bl[kSets];
    auto read_b = [&](int kk, int set) {
      const int pc = (kk >> 1) * 4 + (kk & 1);      // piece of plane h; plane l is two pieces on
      bh[set] = *reinterpret_cast<const h8*>(b_rd + buf * kGran + pc * 1024);
      bl[set] = *reinterpret_cast<const h8*>(b_rd + buf * kGran + (pc + 2) * 1024);
    };
#pragma unroll
    for (int kk = 0; kk < kAhead; ++kk) read_b(kk, kk % kSets);
#pragma unroll
    for (int kk = 0; kk < NK16; ++kk) {
      const int set = kk % kSets;
      if (kk + kAhead < NK16) read_b(kk + kAhead, (kk + kAhead) % kSets);
      __builtin_amdgcn_sched_barrier(0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[kk], bh[set], acc1, 0, 0, 0);    // l h'
      acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[kk], bl[set], acc2, 0, 0, 0);    // h l'
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[kk], bh[set], acc, 0, 0, 0);      // h h'
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] += acc1[r] + acc2[r];      // the two small terms first
    // ---- epilogue of the granule: 2^-(ew + ea), store, column statistics of the rows below M
    float cs = 0.f, cq = 0.f;
    {
      float* const yb = g.y + (long)m0 * Cout;                       // uniform base + 32-bit lane offsets
      const unsigned o0 = (unsigned)(wave * 32 + 4 * lh) * (unsigned)Cout + (unsigned)(nb * RNB + li);
      if (full) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float v = acc[r] * oscale;
          yb[o0 + (unsigned)((r & 3) + 8 * (r >> 2)) * (unsigned)Cout] = v;
          cs += v;
          cq = fmaf(v, v, cq);
        }
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int rl = wave * 32 + 4 * lh + (r & 3) + 8 * (r >> 2);
          const float v = m0 + rl < M ? acc[r] * oscale : 0.f;
          if (m0 + rl < M) yb[o0 + (unsigned)((r & 3) + 8 * (r >> 2)) * (unsigned)Cout] = v;
          cs += v;
          cq = fmaf(v, v, cq);
        }
      }
    }
    if (!g.part_sum && g.err) flag_nonfinite(cq, g.err);
    if (g.part_sum) {
      cs += __shfl_xor(cs, 32);
      cq += __shfl_xor(cq, 32);
      if (lh == 0) {
        scratch[nb & 1][0][wave][li] = cs;
        scratch[nb & 1][1][wave][li] = cq;
      }
    }
    buf = buf == 2 ? 0 : buf + 1;
  }
  if (g.part_sum) {
    __syncthreads();
    if (tid < RNB) {
      const float (*sc)[4][RNB] = scratch[(n_gran - 1) & 1];
      const long o = (long)blockIdx.x * Cout + (n_gran - 1) * RNB + tid;
      g.part_sum[o] = (sc[0][0][tid] + sc[0][1][tid]) + (sc[0][2][tid] + sc[0][3][tid]);
      g.part_sq[o] = (sc[1][0][tid] + sc[1][1][tid]) + (sc[1][2][tid] + sc[1][3][tid]);
    }
  }
}

}  // namespace

// dense [M][Cin] input (a stride-1 1x1 convolution on an NHWC tensor), Cin = 64 / 128 / 256
bool conv1x1_areg_eligible(const float* x, long M, int Cin, int Cout, int bn, const float* in_scale, const float* in_shift) {
  return (Cin == 64 || Cin == 128 || Cin == 256) && (bn == 64 || bn == 128) && Cout % bn == 0 && M > 0 && M < (1l << 24) &&
         M * (long)(Cin > Cout ? Cin : Cout) < (1l << 31) && aligned16(x) &&
         (!in_scale || (aligned16(in_scale) && aligned16(in_shift)));
}

// weight image, tile width and statistics rows (conv1x1_tiles_m(M) = ceil(M / 128)) as conv_fwd_f16x3 with k = 1.
// in_exp: the input is multiplied by 2^in_exp on its way into the f16 planes (exact; undone in the epilogue).
int conv1x1_fwd_areg(const float* x, const unsigned* wimg, int bn, float* y, const float* in_scale, const float* in_shift,
                     int relu_in, float* part_sum, float* part_sq, long M, int Cin, int Cout, int in_exp, hipStream_t stream,
                     int* err) {
  CAPNET_REQUIRE(x && wimg && y && aligned16(wimg) && conv1x1_areg_eligible(x, M, Cin, Cout, bn, in_scale, in_shift),
                 "conv1x1_fwd_areg: operands not eligible (Cin=%d Cout=%d bn=%d)", Cin, Cout, bn);
  CAPNET_REQUIRE((in_scale == nullptr) == (in_shift == nullptr) && (part_sum == nullptr) == (part_sq == nullptr),
                 "conv1x1_fwd_areg: scale / shift and statistics come in pairs");
  CAPNET_REQUIRE(in_exp > -64 && in_exp < 64, "conv1x1_fwd_areg: input exponent %d", in_exp);
  RArgs a{};
  a.x = x; a.wimg = wimg; a.y = y; a.in_scale = in_scale; a.in_shift = in_shift; a.part_sum = part_sum; a.part_sq = part_sq;
  a.M = (int)M; a.Cout = Cout; a.bn = bn; a.relu_in = relu_in; a.in_exp = in_exp; a.err = err;
  const dim3 grid(cdiv((int)M, RBM)), block(256);
#define CAPNET_AREG_LAUNCH(KG_)                                                                        \
  do {                                                                                                \
    if (in_scale) CAPNET_LAUNCH_TIMED((conv1x1_areg_kernel<KG_, true>), grid, block, stream, a);      \
    else CAPNET_LAUNCH_TIMED((conv1x1_areg_kernel<KG_, false>), grid, block, stream, a);              \
  } while (0)
  if (Cin == 64) CAPNET_AREG_LAUNCH(2);
  else if (Cin == 128) CAPNET_AREG_LAUNCH(4);
  else CAPNET_AREG_LAUNCH(8);
#undef CAPNET_AREG_LAUNCH
  CAPNET_LAUNCH_CHECK();
  return kOk;
}

}  // namespace capnet
