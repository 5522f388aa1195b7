// 1x1 convolution with fp32-grade results from THREE f16 MFMA products per multiply: every fp32 operand
// is scaled by a power of two and split into two f16 pieces x 2^s = h + l (h = f16(x 2^s), l = f16(x 2^s - h):
// 22..23 significant bits), and
//     x y 2^(s+s')  =  h h' + (h l' + l h')          [ + l l' below 2^-22 |x y| ],
// each an exact f16 x f16 product accumulated in fp32 by v_mfma_f32_32x32x16_f16 (which keeps f16
// subnormals: tools/probes/native/mfma_f16_denorm.hip). Against the split into three bf16 pieces
// (conv_bf16x6.hip: six products) this halves the matrix work, moves 4 instead of 6 bytes per element through
// LDS and needs 5 instead of 11 VALU per pair for the split. Measured rms error against fp64 on the trunk's
// shapes (tests/test_kernels_gpu.py): at or below the f32-MFMA kernels' -- the fp32 accumulation of K
// products, not the 2^-23 of the operands, sets it (numpy model of all three: tools/probes/split_error.py).
//
// Scales. Weights: 2^ew chosen per tensor at pack time so that max |w| 2^ew lies in [2^13, 2^14) (stored in
// the image's header): their residuals are normal f16 numbers. Activations are split as they are: the
// residual of an activation below 0.25 is a subnormal f16, exact to 2^-25 absolute, which the MFMA keeps
// (a 2^4 prescale changed no digit of the rms error in the model or on the GPU and cost two VALU per pair);
// the range ends at |x| = 65504. The epilogue multiplies the accumulators by 2^-ew: exact.
//
// The same kernel runs the 3x3 convolutions (TAPS = 9) as an implicit GEMM over k = (tap, channel): a step is
// 32 channels of one tap, the A loader moves to the next tap's pixel every Cin / 32 steps and zeroes what
// falls into the padding (med3 bounds of the fold, or a 0 / 1 factor) -- three f16 products on the 16-bit pipe
// (27 Cin f16 MACs per output) need 2.4x less matrix time than Winograd's 4 Cin f32 MACs on the f32 pipe.
//
// Replaces conv_bf16x6.hip on the 1x1 convolutions of the ResNet-152 bottlenecks in train mode
// (torchvision Bottleneck conv1 / conv3 / downsample, call sites stylenet/model.py:15-18,24) whenever
// Cin is a multiple of 64: raw output + per-tile column sums / sums of squares for the BatchNorm that
// follows; PRE applies the previous BatchNorm + ReLU while the A tile is staged.
//
// Kernel structure (persistent; tile 128 x BN, BN = 64 / 128; 256 threads = 2 x 2 waves of 64 x BN/2;
// k-tile 32 = two k16 groups): as conv_bf16x6.hip's persistent kernel --
//   * a workgroup walks tiles blockIdx.x, + gridDim.x, ...; all (tile, k-tile) steps form one software
//     pipeline: loads run two steps ahead in two register sets (4 + NBR 16-B loads per thread and step),
//     waits are counted (vmcnt retires in issue order, stores included);
//   * LDS image of a step: [plane h,l][k16 group][row][2 cells], a cell = 8 consecutive k of one row as f16
//     (16 B); cell position XOR-ed with bit 3 of the row: conflict-free ds_read_b128 fragments. Weights are
//     laid out once per weight version as that exact image and go global -> registers -> LDS;
//   * per step 12 NT MFMAs: [deferred h h' of the previous step's second group, while this step's fragments
//     arrive] then 8 phases of 2..3 MFMAs with one pair-of-pairs of the next step's fold + split between them
//     (sched_barrier pins the phases, a volatile use pins each phase's VALU), the last two MFMAs cover the
//     LDS writes and the issue of the loads two steps ahead;
//   * epilogue: accumulators x 2^-ew, stored straight from the MFMA layout (2 rows x 128 B per instruction);
//     column statistics as everywhere (block_col_stats).
#include <cstdlib>

#include "common.h"
#include "mfma_core.h"
#include "kernels.h"

namespace capnet {

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int HBM = 128, HBK = 32;
// bytes from one (plane, k16 group) of the A image (128 rows x 2 cells) to the next; + 16: lanes 2i / 2i + 1 stage the same row
// of the two groups, 4 KB apart they hit the same banks (SQ_LDS_BANK_CONFLICT: 3 % of the kernel's cycles)
constexpr int kSubA = HBM * 2 * 16 + 16;
constexpr int kHdrWords = 4;                     // image header: [0] ew, [1] bits of max |w| (pack scratch)
constexpr int kFoldMaxH = 512;                   // input channels whose BatchNorm scale / shift live in LDS

struct HArgs {
  const float* x;
  const unsigned* wimg;      // header + packed weight image
  float* y;
  const float* in_scale;
  const float* in_shift;
  float* part_sum;
  float* part_sq;
  // inference epilogue (out_scale != null): y = act(acc * out_scale[n] + out_shift[n] + res)
  const float* out_scale;
  const float* out_shift;
  const float* res;
  int relu_out;
  int M, Cin, Cout, relu_in;
  int in_exp;                // the input enters the f16 planes multiplied by 2^in_exp (undone in the epilogue)
  int* err;                  // error word (launches without statistics check their outputs for non-finite values)
  int H, W, pad;             // input map and zero padding (3x3 convolutions)
  int tiles_m, tiles_n;
  unsigned tn_mul, tn_sh;
  int OW, OHW, stride, sxb, sxh, sxw;
  unsigned ohw_mul, ohw_sh, ow_mul, ow_sh;
};

// output row m -> image offset (floats) and the input coordinates of tap (0, 0)
__device__ __forceinline__ void h_row_origin(const HArgs& g, int m, int& boff, int& ih0, int& iw0) {
  const int b = (int)fast_div((unsigned)m, g.ohw_mul, g.ohw_sh);
  const int rem = m - b * g.OHW;
  const int oh = (int)fast_div((unsigned)rem, g.ow_mul, g.ow_sh);
  const int ow = rem - oh * g.OW;
  boff = b * g.sxb;
  ih0 = oh * g.stride - g.pad;
  iw0 = ow * g.stride - g.pad;
}

// byte offset of cell (row, c) inside one (plane, group) sub-image
__host__ __device__ inline unsigned h_cell(int row, int c) {
  const int r = row & 15;
  return (unsigned)((row * 2 + (c ^ ((r >> 3) & 1))) * 16);
}

// (x0, x1) -> packed f16 pairs of the two pieces; the subtraction is exact in fp32
__device__ __forceinline__ void split2(float x0, float x1, unsigned& h, unsigned& l) {
  const f32x2 v = {x0, x1};
  const f16x2 hh = __builtin_convertvector(v, f16x2);          // v_cvt_pk_f16_f32 (round to nearest even)
  const f32x2 r = v - __builtin_convertvector(hh, f32x2);
  h = __builtin_bit_cast(unsigned, hh);
  l = __builtin_bit_cast(unsigned, __builtin_convertvector(r, f16x2));
}

template <int NBR>
struct HRegs {
  f32x4 a[4], b[NBR];
  float lo, hi;      // TAPS > 1: the fold's bounds for this step's tap (0, 0 in the padding; PRE) or the factor 0 / 1 in hi
};

template <int N> __device__ __forceinline__ void h_wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int BN, bool PRE, int TAPS>
__global__ __launch_bounds__(256, 2) void conv_f16x3_kernel(const HArgs g) {
  constexpr int NT = BN / 64;
  constexpr int kSubB = BN * 2 * 16;
  constexpr int kImgA = 4 * kSubA, kImgB = 4 * kSubB;
  constexpr int kStage = kImgA + kImgB;
  constexpr int NBR = kImgB / 16 / 256;             // 16-B cells of the B image per thread (4 / 2)
  static_assert(NBR * 256 * 16 == kImgB, "weight image cells per thread");
  constexpr int NLD = NBR + 4;                      // loads of one step
  constexpr int kWaitEpi = NLD + 32 * NT > 63 ? 63 : NLD + 32 * NT;            // leaves one step's loads and a plain epilogue's 32 NT stores in flight (6-bit counter)
  constexpr int kFold = PRE ? 2 * kFoldMaxH * 4 : 0;
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * kStage + 4 * BN * 4 + kFold];
  float* scratch = reinterpret_cast<float*>(lds + 2 * kStage);
  const float* fold = reinterpret_cast<const float*>(lds + 2 * kStage + 4 * BN * 4);    // scale 2^4 | shift 2^4
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int li = lane & 31, lh = lane >> 5;
  const int total = g.tiles_m * g.tiles_n, G = (int)gridDim.x;
  const int nkc = g.Cin / HBK;               // steps per tap
  const int nk = TAPS * nkc;
  const int my_tiles = (total - 1 - (int)blockIdx.x) / G + 1;
  const int n_it = my_tiles * nk;
  const float* wimg = reinterpret_cast<const float*>(g.wimg + kHdrWords);
  const float oscale = ldexpf(1.f, -((int)g.wimg[0] + g.in_exp));
  const float iscale = ldexpf(1.f, g.in_exp);

  const int arow = tid >> 1, ag = tid & 1;          // A staging: this thread's row and k16 group

  // ---- issue cursor: two steps ahead of the MFMAs
  int iw = (int)blockIdx.x, ikc = 0, itap = 0;
  unsigned i_avoff = 0;
  int i_boff = 0, i_ih0 = 0, i_iw0 = 0;
  bool i_ok = true;
  const float* i_sA = g.x;
  const float* i_sB = nullptr;
  // this thread's pixel of the current tap: clamped into the map (the load is always legal), i_ok says whether it counts
  auto i_tap = [&]() {
    const int kh = TAPS == 1 ? 0 : itap / 3, kw = TAPS == 1 ? 0 : itap - 3 * (itap / 3);
    const int ih = i_ih0 + kh, iwp = i_iw0 + kw;
    if (TAPS > 1) i_ok = (unsigned)ih < (unsigned)g.H && (unsigned)iwp < (unsigned)g.W;
    const int ihc = TAPS == 1 ? ih : min(max(ih, 0), g.H - 1), iwc = TAPS == 1 ? iwp : min(max(iwp, 0), g.W - 1);
    i_avoff = (unsigned)(i_boff + ihc * g.sxh + iwc * g.sxw + 16 * ag) * 4u;
    i_sA = g.x;
  };
  auto i_tile = [&]() {
    const int id = xcd_remap(iw, total);
    const int tm = (int)fast_div((unsigned)id, g.tn_mul, g.tn_sh), tn = id - tm * g.tiles_n;
    const int am = tm * HBM + arow;
    h_row_origin(g, am < g.M ? am : g.M - 1, i_boff, i_ih0, i_iw0);   // rows past M: a valid row, zeroed in the epilogue
    itap = 0;
    i_tap();
    i_sB = wimg + ((long)tn * nk) * (kImgB / 4);
  };
  i_tile();
  if (PRE) {
    float* f = reinterpret_cast<float*>(lds + 2 * kStage + 4 * BN * 4);
    for (int i = tid; i < g.Cin; i += 256) {
      f[i] = g.in_scale[i] * iscale;               // (a power of two: exact) -- the prescale costs nothing in the loop
      f[kFoldMaxH + i] = g.in_shift[i] * iscale;
    }
    __syncthreads();
  }
  const float lo = g.relu_in ? 0.f : -__builtin_inff();
  // Loads are only ever issued for steps that exist and every issued set is consumed (landed): a register
  // written by a load nobody consumes is free for the compiler to reuse at once, and the load would land in
  // whatever lives there by then.
  // ONE asm statement per step: s_nop 4 (the SGPR bases may come straight from a v_readlane reload: mfma_core.h,
  // gload16) + the NBR weight-image loads + the four activation loads (immediate offsets off one base). Outputs
  // are early-clobber: no destination may share a register with an offset a later load of the statement reads.
  const unsigned voffB = (unsigned)tid * 16u + 4096u;      // cells tid, tid + 256 at immediate offsets -4096 / 0 (13-bit signed field)
  auto issue = [&](HRegs<NBR>& R) {
    if constexpr (NBR == 4) {
      asm volatile(
          "s_nop 4\n\t"
          "global_load_dwordx4 %0, %8, %10 offset:-4096\n\t"
          "global_load_dwordx4 %1, %8, %10\n\t"
          "global_load_dwordx4 %2, %11, %10 offset:-4096\n\t"
          "global_load_dwordx4 %3, %11, %10\n\t"
          "global_load_dwordx4 %4, %9, %12\n\t"
          "global_load_dwordx4 %5, %9, %12 offset:16\n\t"
          "global_load_dwordx4 %6, %9, %12 offset:32\n\t"
          "global_load_dwordx4 %7, %9, %12 offset:48"
          : "=&v"(R.b[0]), "=&v"(R.b[1]), "=&v"(R.b[2]), "=&v"(R.b[3]), "=&v"(R.a[0]), "=&v"(R.a[1]), "=&v"(R.a[2]),
            "=&v"(R.a[3])
          : "v"(voffB), "v"(i_avoff), "s"(i_sB), "v"(voffB + 8192u), "s"(i_sA)
          : "memory");
    } else {
      asm volatile(
          "s_nop 4\n\t"
          "global_load_dwordx4 %0, %6, %8 offset:-4096\n\t"
          "global_load_dwordx4 %1, %6, %8\n\t"
          "global_load_dwordx4 %2, %7, %9\n\t"
          "global_load_dwordx4 %3, %7, %9 offset:16\n\t"
          "global_load_dwordx4 %4, %7, %9 offset:32\n\t"
          "global_load_dwordx4 %5, %7, %9 offset:48"
          : "=&v"(R.b[0]), "=&v"(R.b[1]), "=&v"(R.a[0]), "=&v"(R.a[1]), "=&v"(R.a[2]), "=&v"(R.a[3])
          : "v"(voffB), "v"(i_avoff), "s"(i_sB), "s"(i_sA)
          : "memory");
    }
    if (TAPS > 1) {
      R.lo = i_ok ? lo : 0.f;
      R.hi = i_ok ? (PRE ? __builtin_inff() : iscale) : 0.f;
    }
    i_sB += kImgB / 4;
    i_sA += HBK;
    if (++ikc == nkc) {
      ikc = 0;
      if (++itap == TAPS) {
        if (iw + G < total) iw += G;
        i_tile();
      } else {
        i_tap();
      }
    }
  };
  auto landed = [&](HRegs<NBR>& R) {
    CAPNET_LANDED4(R.a[0], R.a[1], R.a[2], R.a[3]);
#pragma unroll
    for (int q = 0; q < NBR; ++q) CAPNET_LANDED1(R.b[q]);
  };

  f32x16 acc[2][NT];
  auto zero_acc = [&]() {
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;
  };
  zero_acc();

  int st_k = 16 * ag;            // this thread's first channel of the step being staged (PRE)
  const unsigned awr0 = h_cell(arow, 0), awr1 = h_cell(arow, 1);
  // fold + split of one pair of the staged step (pair p = floats 2p, 2p + 1 of the thread's 16)
  // (blo, bhi: the fold's bounds -- [lo, inf) inside the map, [0, 0] in the padding of a 3x3 convolution; without a
  //  fold bhi is the factor 1 / 0)
  auto stage_pair = [&](const float* x, const float* fs, const float* ft, int p, float blo, float bhi, unsigned& h, unsigned& l) {
    float x0 = x[2 * p], x1 = x[2 * p + 1];
    if (PRE) {
      x0 = fmaf(x0, fs[2 * p], ft[2 * p]);
      x1 = fmaf(x1, fs[2 * p + 1], ft[2 * p + 1]);
      if (TAPS > 1) {
        x0 = __builtin_amdgcn_fmed3f(x0, blo, bhi);
        x1 = __builtin_amdgcn_fmed3f(x1, blo, bhi);
      } else {
        x0 = fmaxf(x0, lo);
        x1 = fmaxf(x1, lo);
      }
    } else {
      x0 *= bhi;             // without a fold: the prescale (and the 0 of a padding tap)
      x1 *= bhi;
    }
    split2(x0, x1, h, l);
  };

  const unsigned char* a_rd = lds + h_cell(wm * 64 + li, lh);
  const unsigned char* b_rd = lds + kImgA + h_cell(wn * (BN / 2) + li, lh);
  struct HTail { f16x8 ah[2], bh[NT]; };            // planes of the deferred term h h' of the second k16 group
  auto tail = [&](const HTail& T) {
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(T.ah[mt], T.bh[nt], acc[mt][nt], 0, 0, 0);
  };
  // the prologue's plain staging (no MFMAs to hide behind)
  auto store = [&](HRegs<NBR>& R, int stage) {
#pragma unroll
    for (int q = 0; q < NBR; ++q) *reinterpret_cast<f32x4*>(lds + stage * kStage + kImgA + (tid + 256 * q) * 16) = R.b[q];
    float x[16], fs[16], ft[16];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        x[4 * q + j] = R.a[q][j];
        fs[4 * q + j] = PRE ? fold[st_k + 4 * q + j] : 0.f;
        ft[4 * q + j] = PRE ? fold[kFoldMaxH + st_k + 4 * q + j] : 0.f;
      }
    if (PRE) { st_k += HBK; if (st_k >= g.Cin) st_k -= g.Cin; }
    u32x4 ph[2], pl[2];
#pragma unroll
    for (int p = 0; p < 8; ++p) {
      unsigned h, l;
      stage_pair(x, fs, ft, p, R.lo, TAPS > 1 ? R.hi : iscale, h, l);
      ph[p >> 2][p & 3] = h; pl[p >> 2][p & 3] = l;
    }
    unsigned char* d = lds + stage * kStage + ag * kSubA;
    *reinterpret_cast<u32x4*>(d + awr0) = ph[0];
    *reinterpret_cast<u32x4*>(d + awr1) = ph[1];
    *reinterpret_cast<u32x4*>(d + 2 * kSubA + awr0) = pl[0];
    *reinterpret_cast<u32x4*>(d + 2 * kSubA + awr1) = pl[1];
  };

  // One step: the MFMAs of LDS[stage] with fold + split of the next step's A cells (R -> LDS[1 - stage]) between them.
  auto body = [&](HRegs<NBR>& R, int stage, HTail& Tc, const HTail& Tp, bool pending, bool do_issue) {
    // fragments [group][plane]: plane 0 = h, 1 = l; read in the order the terms need them (l h', h l', h h');
    // the second group's are requested in phase 1, well ahead of their first MFMA
    f16x8 af[2][2][2], bf[2][NT][2];
    auto read_frags = [&](int gq) {
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) af[gq][mt][1] = *reinterpret_cast<const f16x8*>(a_rd + stage * kStage + (2 + gq) * kSubA + mt * 1024);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) bf[gq][nt][0] = *reinterpret_cast<const f16x8*>(b_rd + stage * kStage + gq * kSubB + nt * 1024);
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) af[gq][mt][0] = *reinterpret_cast<const f16x8*>(a_rd + stage * kStage + gq * kSubA + mt * 1024);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) bf[gq][nt][1] = *reinterpret_cast<const f16x8*>(b_rd + stage * kStage + (2 + gq) * kSubB + nt * 1024);
    };
    read_frags(0);
    // the previous BatchNorm's scale and shift of this thread's 16 channels, 8 at a time
    f32x4 fsv[2], ftv[2];
    const int fk = st_k;
    auto read_fold = [&](int half) {
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        fsv[q] = *reinterpret_cast<const f32x4*>(fold + fk + 8 * half + 4 * q);
        ftv[q] = *reinterpret_cast<const f32x4*>(fold + kFoldMaxH + fk + 8 * half + 4 * q);
      }
    };
    if (PRE) {
      read_fold(0);
      st_k += HBK;
      if (st_k >= g.Cin) st_k -= g.Cin;
    }
    __builtin_amdgcn_sched_barrier(0);
    if (pending) tail(Tp);
    __builtin_amdgcn_sched_barrier(0);
    float x[16];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int j = 0; j < 4; ++j) x[4 * q + j] = R.a[q][j];
    u32x4 ph[2], pl[2];
    unsigned char* const dB = lds + (1 - stage) * kStage + kImgA;
    unsigned char* const dA = lds + (1 - stage) * kStage + ag * kSubA;
    // head MFMAs of this step: group 0 all three terms, group 1 the two small terms (its h h' is the next tail)
    constexpr int NH = 5 * 2 * NT;            // 20 / 10
    constexpr int HP = NH - 2;                // spread over the 8 phases; the last two cover writes + issue
    auto head = [&](int idx) {
      const int t5 = idx / (2 * NT), mt = (idx % (2 * NT)) / NT, nt = idx % NT;
      const int gq = t5 < 3 ? 0 : 1, term = t5 < 3 ? t5 : t5 - 3;          // term 0: l h', 1: h l', 2: h h'
      const int pa = term == 0 ? 1 : 0, pb = term == 1 ? 1 : 0;
      acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[gq][mt][pa], bf[gq][nt][pb], acc[mt][nt], 0, 0, 0);
    };
#pragma unroll
    for (int phase = 0; phase < 8; ++phase) {
      if (phase == 1) read_frags(1);
      if (PRE && phase == 4) read_fold(1);
      unsigned h, l;
      {
        const int p = phase & 3;
        const float fs[8] = {fsv[0][0], fsv[0][1], fsv[0][2], fsv[0][3], fsv[1][0], fsv[1][1], fsv[1][2], fsv[1][3]};
        const float ft[8] = {ftv[0][0], ftv[0][1], ftv[0][2], ftv[0][3], ftv[1][0], ftv[1][1], ftv[1][2], ftv[1][3]};
        stage_pair(x + 8 * (phase >> 2), fs, ft, p, R.lo, TAPS > 1 ? R.hi : iscale, h, l);
      }
      asm volatile("" : "+v"(h), "+v"(l));        // (a volatile use keeps the phase's VALU here: pure IR sinks to its ds_write)
      ph[phase >> 2][phase & 3] = h;
      pl[phase >> 2][phase & 3] = l;
#pragma unroll
      for (int idx = (phase * HP) / 8; idx < ((phase + 1) * HP) / 8; ++idx) head(idx);
      if (phase < NBR) *reinterpret_cast<f32x4*>(dB + (tid + 256 * phase) * 16) = R.b[phase];
      if (phase == 4) {
        *reinterpret_cast<u32x4*>(dA + awr0) = ph[0];
        *reinterpret_cast<u32x4*>(dA + 2 * kSubA + awr0) = pl[0];
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    head(HP);
    *reinterpret_cast<u32x4*>(dA + awr1) = ph[1];
    *reinterpret_cast<u32x4*>(dA + 2 * kSubA + awr1) = pl[1];
    __builtin_amdgcn_sched_barrier(0);
    head(HP + 1);
    __builtin_amdgcn_sched_barrier(0);
    if (do_issue) issue(R);
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) Tc.ah[mt] = af[1][mt][0];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) Tc.bh[nt] = bf[1][nt][0];
  };

  // ---- compute cursor
  int cw = (int)blockIdx.x, ckt = 0;
  int after_epi = 0;         // waits still to be taken with the epilogue's stores counted in
  auto epilogue = [&]() {
    const int id = xcd_remap(cw, total);
    const int tm = (int)fast_div((unsigned)id, g.tn_mul, g.tn_sh), tn = id - tm * g.tiles_n;
    const int m0 = tm * HBM, n0 = tn * BN;
    const bool ragged = m0 + HBM > g.M;
    const bool plain = !ragged && !g.out_scale;
    const unsigned rstep = (unsigned)g.Cout * 4u;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mt][nt][r] *= oscale;          // 2^-ew: exact
    if (g.err && !g.part_sum) {
      float t = 0.f;
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
          for (int r = 0; r < 16; ++r) t += fabsf(acc[mt][nt][r]);
      flag_nonfinite(t, g.err);
    }
    if (ragged) {
      // rows past M were computed from a clamped row: keep them out of the statistics
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        const int row0 = m0 + wm * 64 + mt * 32 + 4 * lh;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = row0 + (r & 3) + 8 * (r >> 2);
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) acc[mt][nt][r] = row < g.M ? acc[mt][nt][r] : 0.f;
        }
      }
    }
    if (plain) {
      // 32 NT unconditional stores (the count the waits after this epilogue rely on). A variant that turned the 32 x 32
      // blocks through LDS to store 16 B per lane took the same 5 400 cycles per tile (the burst of every workgroup's
      // 64 KB, not the instruction count, sets it) and is gone.
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int n = n0 + wn * (BN / 2) + nt * 32 + li;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
          unsigned off = ((unsigned)(m0 + wm * 64 + mt * 32 + 4 * lh) * (unsigned)g.Cout + (unsigned)n) * 4u;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            gstore32(g.y, off, acc[mt][nt][r]);
            off += ((r & 3) == 3 ? 5u : 1u) * rstep;
          }
        }
      }
    } else {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int n = n0 + wn * (BN / 2) + nt * 32 + li;
        const float osc = g.out_scale ? g.out_scale[n] : 1.f, osh = g.out_scale ? g.out_shift[n] : 0.f;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
          int row = m0 + wm * 64 + mt * 32 + 4 * lh;
          unsigned off = ((unsigned)row * (unsigned)g.Cout + (unsigned)n) * 4u;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            if (row < g.M) {
              float v = acc[mt][nt][r];
              if (g.out_scale) {
                v = fmaf(v, osc, osh);
                if (g.res) v += *reinterpret_cast<const float*>(reinterpret_cast<const char*>(g.res) + off);
                if (g.relu_out) v = fmaxf(v, 0.f);
              }
              gstore32(g.y, off, v);
            }
            if ((r & 3) == 3) { row += 5; off += 5u * rstep; } else { row += 1; off += rstep; }
          }
        }
      }
    }
    if (g.part_sum) {
      using T = TileCfg<HBM, BN, 16>;
      block_col_stats<T>(acc, scratch, g.part_sum + (long)tm * g.Cout, g.part_sq + (long)tm * g.Cout, n0, g.Cout);
      __syncthreads();       // scratch is reused by the next tile's statistics
    }
    zero_acc();
    // a plain tile put exactly 32 NT stores behind the loads in flight; anything else (ragged rows, the folded
    // epilogue's own loads) is not counted on: the next wait drains the queue
    after_epi = plain ? 2 : -1;
  };

  HRegs<NBR> R0, R1;
  issue(R0);
  if (n_it > 1) {
    issue(R1);
    h_wait_vmcnt<NLD>();
  } else {
    h_wait_vmcnt<0>();
  }
  landed(R0);
  store(R0, 0);
  if (n_it > 2) issue(R0);
  __syncthreads();

  // nk is even (the launcher requires Cin % 64 == 0), so a tile ends only behind the second step of a pair: one
  // copy of the epilogue. landed + the staging are unconditional -- the last step of a workgroup re-stages stale
  // registers into the stage nobody reads again -- because a branch there splits the scheduling region.
  HTail T0, T1;
  auto step = [&](HRegs<NBR>& R, int stage, int it, HTail& Tc, const HTail& Tp, bool second) {
    if (it + 2 >= n_it) { h_wait_vmcnt<0>(); after_epi = 0; }     // nothing younger in flight
    else if (after_epi > 0) { h_wait_vmcnt<kWaitEpi>(); --after_epi; }
    else if (after_epi < 0) { h_wait_vmcnt<0>(); after_epi = 0; }
    else h_wait_vmcnt<NLD>();
    landed(R);
    body(R, stage, Tc, Tp, second || ckt > 0, it + 3 < n_it);
    __syncthreads();
    ++ckt;
    if (second && ckt == nk) {
      tail(Tc);
      epilogue();
      ckt = 0;
      cw += G;
    }
  };
  for (int it = 0; it < n_it; it += 2) {
    step(R1, 0, it, T0, T1, false);
    step(R0, 1, it + 1, T1, T0, true);
  }
  h_wait_vmcnt<0>();       // nothing of this workgroup may still be in flight when its LDS is handed on
}

// max |w| as float bits (non-negative floats order like unsigned integers)
__global__ __launch_bounds__(256) void conv1x1_f16x3_absmax_kernel(const float* __restrict__ w, unsigned* __restrict__ hdr, long n) {
  float m = 0.f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) m = fmaxf(m, fabsf(w[i]));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  if ((threadIdx.x & 63) == 0 && m > 0.f) atomicMax(hdr + 1, __float_as_uint(m));
}

__device__ __forceinline__ int h_weight_shift(unsigned absmax_bits) {
  // max |w| 2^ew in [2^13, 2^14): a factor 4 below the f16 range, residuals of all but the smallest weights normal
  if (absmax_bits == 0u) return 0;
  const int e = (int)((absmax_bits >> 23) & 0xffu) - 127;       // floor(log2 max|w|)
  const int ew = 13 - e;
  return ew < -100 ? -100 : (ew > 100 ? 100 : ew);
}

// One thread per (tn, kt, plane, group, row, pos): 8 consecutive channels of output channel n and tap kt / (Cin / 32)
// -> one 16-B cell. w is OIHW: [Cout][Cin][taps].
template <int BN>
__global__ __launch_bounds__(256) void conv_f16x3_pack_kernel(const float* __restrict__ w, unsigned* __restrict__ img,
                                                              int Cout, int Cin, int taps) {
  const int ew = h_weight_shift(img[1]);
  if (blockIdx.x == 0 && threadIdx.x == 0) img[0] = (unsigned)ew;
  const float ws = ldexpf(1.f, ew);
  const int nkc = Cin / HBK, nk = taps * nkc, tiles_n = Cout / BN;
  const long cells = (long)tiles_n * nk * 4 * BN * 2;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < cells; i += (long)gridDim.x * blockDim.x) {
    long r = i;
    const int pos = (int)(r & 1); r >>= 1;
    const int row = (int)(r % BN); r /= BN;
    const int sub = (int)(r & 3); r >>= 2;          // plane * 2 + group
    const int kt = (int)(r % nk);
    const int tn = (int)(r / nk);
    const int plane = sub >> 1, gq = sub & 1;
    const int c = pos ^ (((row & 15) >> 3) & 1);
    const int tap = kt / nkc, c0 = (kt - tap * nkc) * HBK + gq * 16 + 8 * c;
    const float* src = w + ((long)(tn * BN + row) * Cin + c0) * taps + tap;
    unsigned out[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float x0 = src[(2 * q) * taps] * ws, x1 = src[(2 * q + 1) * taps] * ws;
      const _Float16 h0 = (_Float16)x0, h1 = (_Float16)x1;
      const _Float16 l0 = (_Float16)(x0 - (float)h0), l1 = (_Float16)(x1 - (float)h1);
      const f16x2 p = plane == 0 ? f16x2{h0, h1} : f16x2{l0, l1};
      out[q] = __builtin_bit_cast(unsigned, p);
    }
    unsigned* dst = img + kHdrWords + (((long)(tn * nk + kt) * 4 + sub) * BN * 2 + (long)row * 2 + pos) * 4;
    dst[0] = out[0]; dst[1] = out[1]; dst[2] = out[2]; dst[3] = out[3];
  }
}

}  // namespace

bool conv_f16x3_eligible(const float* x, long sxb, long sxh, long sxw, long sxc, int Bn, int H, int W, int Cin,
                         int Cout, int k, int stride, int pad, const float* in_scale, const float* in_shift) {
  if (!((k == 1 && pad == 0) || (k == 3 && pad == 1))) return false;
  const long OH = (H + 2 * pad - k) / stride + 1, OW = (W + 2 * pad - k) / stride + 1;
  return sxc == 1 && Cin % (2 * HBK) == 0 && Cout % 64 == 0 && aligned16(x) && sxb % 4 == 0 && sxh % 4 == 0 &&
         sxw % 4 == 0 && (long)Bn * sxb * 4 < (1l << 31) && (long)Bn * OH * OW < (1l << 24) &&
         (long)Bn * OH * OW * Cout * 4 < (1l << 32) &&
         (!in_scale || (aligned16(in_scale) && aligned16(in_shift) && Cin <= kFoldMaxH));
}
bool conv1x1_f16x3_eligible(const float* x, long sxb, long sxh, long sxw, long sxc, int Bn, int H, int W,
                            int Cin, int Cout, int stride, const float* in_scale, const float* in_shift) {
  return conv_f16x3_eligible(x, sxb, sxh, sxw, sxc, Bn, H, W, Cin, Cout, 1, stride, 0, in_scale, in_shift);
}

int conv1x1_f16x3_bn(long M, int Cout) {
  (void)M;
  return Cout % 128 == 0 ? 128 : 64;
}
size_t conv_f16x3_weight_words(int Cin, int Cout, int k) { return (size_t)kHdrWords + (size_t)Cout * Cin * k * k; }
size_t conv1x1_f16x3_weight_words(int Cin, int Cout) { return conv_f16x3_weight_words(Cin, Cout, 1); }

// w OIHW fp32 ([Cout][Cin][k][k]) -> header + the split f16 image for tile width bn
int conv_f16x3_pack(const float* w, unsigned* img, int Cout, int Cin, int k, int bn, hipStream_t stream) {
  CAPNET_REQUIRE(w && img && Cin % HBK == 0 && (k == 1 || k == 3) && (bn == 64 || bn == 128 || bn == 256) && Cout % bn == 0 && aligned16(img),
                 "conv_f16x3_pack: bad argument (Cin=%d Cout=%d k=%d bn=%d)", Cin, Cout, k, bn);
  CAPNET_HIP_CHECK(hipMemsetAsync(img, 0, kHdrWords * 4, stream));
  const long n = (long)Cout * Cin * k * k;
  hipLaunchKernelGGL(conv1x1_f16x3_absmax_kernel, dim3((int)(cdiv(n, 256 * 8) > 1024 ? 1024 : cdiv(n, 256 * 8))), dim3(256), 0,
                     stream, w, img, n);
  CAPNET_LAUNCH_CHECK();
  const long cells = n / 2;
  const int grid = (int)(cdiv(cells, 256) > 4096 ? 4096 : cdiv(cells, 256));
  if (bn == 256) hipLaunchKernelGGL(conv_f16x3_pack_kernel<256>, dim3(grid), dim3(256), 0, stream, w, img, Cout, Cin, k * k);   // (conv1x1_tail_kernel<256> only)
  else if (bn == 128) hipLaunchKernelGGL(conv_f16x3_pack_kernel<128>, dim3(grid), dim3(256), 0, stream, w, img, Cout, Cin, k * k);
  else hipLaunchKernelGGL(conv_f16x3_pack_kernel<64>, dim3(grid), dim3(256), 0, stream, w, img, Cout, Cin, k * k);
  CAPNET_LAUNCH_CHECK();
  return kOk;
}
int conv1x1_f16x3_pack(const float* w, unsigned* img, int Cout, int Cin, int bn, hipStream_t stream) {
  return conv_f16x3_pack(w, img, Cout, Cin, 1, bn, stream);
}

int conv_fwd_f16x3(const float* x, long sxb, long sxh, long sxw, const unsigned* wimg, int bn, float* y,
                   const float* in_scale, const float* in_shift, int relu_in, float* part_sum, float* part_sq, int Bn,
                   int H, int W, int Cin, int Cout, int k, int stride, int pad, hipStream_t stream,
                   const float* out_scale, const float* out_shift, const float* res, int relu_out, int in_exp, int* err) {
  CAPNET_REQUIRE(x && wimg && y && stride >= 1 && in_exp > -64 && in_exp < 64, "conv_fwd_f16x3: bad argument");
  CAPNET_REQUIRE(conv_f16x3_eligible(x, sxb, sxh, sxw, 1, Bn, H, W, Cin, Cout, k, stride, pad, in_scale, in_shift) &&
                     aligned16(wimg) && (bn == 64 || bn == 128) && Cout % bn == 0,
                 "conv_fwd_f16x3: operands not eligible");
  CAPNET_REQUIRE((in_scale == nullptr) == (in_shift == nullptr), "conv_fwd_f16x3: scale/shift pair");
  CAPNET_REQUIRE((part_sum == nullptr) == (part_sq == nullptr), "conv_fwd_f16x3: stats pair");
  CAPNET_REQUIRE(!out_scale || (out_shift && !part_sum), "conv_fwd_f16x3: folded epilogue takes no statistics");
  HArgs a{};
  const int OH = (H + 2 * pad - k) / stride + 1, OW = (W + 2 * pad - k) / stride + 1;
  a.x = x; a.wimg = wimg; a.y = y; a.in_scale = in_scale; a.in_shift = in_shift;
  a.part_sum = part_sum; a.part_sq = part_sq;
  a.out_scale = out_scale; a.out_shift = out_shift; a.res = res; a.relu_out = relu_out;
  a.M = Bn * OH * OW; a.Cin = Cin; a.Cout = Cout; a.relu_in = relu_in; a.in_exp = in_exp; a.err = err;
  a.H = H; a.W = W; a.pad = pad;
  a.tiles_m = cdiv(a.M, HBM); a.tiles_n = Cout / bn;
  magic_div((unsigned)a.tiles_n, &a.tn_mul, &a.tn_sh);
  a.OW = OW; a.OHW = OH * OW; a.stride = stride;
  a.sxb = (int)sxb; a.sxh = (int)sxh; a.sxw = (int)sxw;
  magic_div((unsigned)(OH * OW), &a.ohw_mul, &a.ohw_sh);
  magic_div((unsigned)OW, &a.ow_mul, &a.ow_sh);
  // persistent: two workgroups per CU walk the tiles (a multiple of 8, so that a workgroup stays on its XCD's share
  // of the tile order)
  const int cap = 512;
  const int total = a.tiles_m * a.tiles_n;
  const dim3 grid(total <= cap ? total : cap), block(256);
#define CAPNET_H3_LAUNCH(BN_, PRE_, TAPS_) CAPNET_LAUNCH_TIMED((conv_f16x3_kernel<BN_, PRE_, TAPS_>), grid, block, stream, a)
  if (k == 1) {
    if (bn == 128) { if (in_scale) CAPNET_H3_LAUNCH(128, true, 1); else CAPNET_H3_LAUNCH(128, false, 1); }
    else { if (in_scale) CAPNET_H3_LAUNCH(64, true, 1); else CAPNET_H3_LAUNCH(64, false, 1); }
  } else {
    if (bn == 128) { if (in_scale) CAPNET_H3_LAUNCH(128, true, 9); else CAPNET_H3_LAUNCH(128, false, 9); }
    else { if (in_scale) CAPNET_H3_LAUNCH(64, true, 9); else CAPNET_H3_LAUNCH(64, false, 9); }
  }
#undef CAPNET_H3_LAUNCH
  CAPNET_LAUNCH_CHECK();
  return kOk;
}
int conv1x1_fwd_f16x3(const float* x, long sxb, long sxh, long sxw, const unsigned* wimg, int bn, float* y,
                      const float* in_scale, const float* in_shift, int relu_in, float* part_sum,
                      float* part_sq, int Bn, int H, int W, int Cin, int Cout, int stride,
                      hipStream_t stream, const float* out_scale, const float* out_shift, const float* res,
                      int relu_out) {
  return conv_fwd_f16x3(x, sxb, sxh, sxw, wimg, bn, y, in_scale, in_shift, relu_in, part_sum, part_sq, Bn, H, W, Cin,
                        Cout, 1, stride, 0, stream, out_scale, out_shift, res, relu_out);
}

}  // namespace capnet
