// 1x1 convolution with fp32-grade results on the bf16 matrix cores: every fp32 operand is split
// into three bf16 pieces x = h + m + l (h = bf16(x), m = bf16(x - h), l = bf16(x - h - m): 24
// mantissa bits in all) and the product is formed from the six partial products that matter,
//     x y  =  h h' + (h m' + m h') + (m m' + h l' + l h')   [ + terms below 2^-24 |x y| ],
// each an exact bf16 x bf16 product accumulated in fp32 by v_mfma_f32_32x32x16_bf16. Six bf16 MFMAs
// per K = 16 cost 6 x 32 cycles where eight K = 2 f32 MFMAs cost 512: 2.67x the f32 matrix rate, and
// the bf16 MFMA does not share its pipe with the VALU (the f32 one does).
// Accuracy (tests/test_kernels_gpu.py, tools/split_error.py): rms error against fp64 5e-8 .. 1e-7 on
// the trunk's shapes -- not above the f32 MFMA kernels' (1e-7 .. 2e-7): the results are fp32 results.
//
// Replaces the K-major f32 kernel on the 1x1 convolutions of the ResNet-152 bottlenecks in train mode
// (torchvision Bottleneck conv1 / conv3 / downsample, call sites stylenet/model.py:15-18,24): raw
// output + per-tile column sums / sums of squares for the BatchNorm that follows; PRE applies the
// previous BatchNorm + ReLU while the A tile is staged, exactly as conv_f32_v2.hip does.
//
// Layout (tile 128 x BN, BN = 64 / 128; 256 threads = 2 x 2 waves of 64 x BN/2; k-tile = 16):
//   * B (weights): split and laid out ONCE per weight version as the exact LDS image of every
//     (n-tile, k-tile): [plane h,m,l][16-row group][row][2 cells], a cell = 8 consecutive k of one
//     output channel as bf16 (16 B): 1-KB contiguous pieces per wave-instruction, global -> registers
//     -> LDS (see the note on LDS-DMA below).
//   * A (activations): thread (row = tid >> 1, cell = tid & 1) loads 8 consecutive fp32 of its pixel
//     (two 16-B loads off one scalar base), folds the BatchNorm, splits them (11 VALU per pair:
//     3 v_cvt_pk_bf16_f32, 4 bit extractions, 4 exact subtractions) and writes one 16-B cell per
//     plane. Cell position inside a row is XOR-ed with bit 3 of the row, so that the fragment reads
//     (ds_read_b128, 16 lanes per LDS pass, row stride 32 B) touch 16 distinct bank slots.
//   * fragments: lane (row = lane & 31, half = lane >> 5) needs k = 8 half .. 8 half + 7 of its row
//     = ONE cell per plane: 12 ds_read_b128 feed the 24 MFMAs of a wave's k-tile.
//   * 24 KB of LDS per stage, two stages: three workgroups per CU.
//
// Why the weights do not use LDS-DMA here. The first version streamed the B image with
// global_load_lds_dwordx4 (two or three per wave and k-tile). Its own results were right, but workgroups
// of OTHER kernels sharing its CUs computed wrong values now and then: att_scores_fwd_kernel (no LDS of
// its own; a wave reduction through __shfl_xor) returned one wrong score in ~25 % of its launches
// beside it (tools/victim2.py), which showed up as TrunkPipeline(attention=True) steps that were no longer
// reproducible. No out-of-bounds access (guard bands, tools/x6_guard.py), no LDS or register damage
// in a canary kernel (tools/native/canary_probe.hip), wait states around the M0 writes changed nothing,
// and neither the K-major conv kernel nor gemm_dma.hip (both LDS-DMA users) trigger it. With the
// weights going through registers the interference is gone (0 / 100) at the same speed (-4 % in
// isolation). tests/test_step_gpu.py keeps a decoder-beside-trunk reproducibility check for this.
#include <cstdlib>

#include "common.h"
#include "mfma_core.h"
#include "kernels.h"

namespace capnet {

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int XBM = 128, XBK = 16;
constexpr int kPlaneA = XBM * 2 * 16;            // bytes of one plane of the A image (128 rows x 2 cells)

struct XArgs {
  const float* x;
  const unsigned* wimg;      // packed weight image
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
  int abl;   // diagnostics (CAPNET_X6_ABLATE, persistent kernel): 4 no statistics, 8 no output stores
  int tiles_m, tiles_n;
  long long* stamps;   // diagnostics (CAPNET_X6_STAMPS = device address): workgroup 0 records 4 clock values per step
  unsigned tn_mul, tn_sh;
  int OW, OHW, stride, sxb, sxh, sxw;
  unsigned ohw_mul, ohw_sh, ow_mul, ow_sh;
};

__device__ __forceinline__ unsigned x_row_offset(const XArgs& g, int m) {
  const int b = (int)fast_div((unsigned)m, g.ohw_mul, g.ohw_sh);
  const int rem = m - b * g.OHW;
  const int oh = (int)fast_div((unsigned)rem, g.ow_mul, g.ow_sh);
  const int ow = rem - oh * g.OW;
  return (unsigned)(b * g.sxb + oh * g.stride * g.sxh + ow * g.stride * g.sxw);
}

// byte offset of cell (row, c) inside one plane
__host__ __device__ inline unsigned x_cell(int row, int c) {
  const int rg = row >> 4, r = row & 15;
  return (unsigned)(((rg * 16 + r) * 2 + (c ^ ((r >> 3) & 1))) * 16);
}

typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned cvt_pk_bf16(float a, float b) {      // v_cvt_pk_bf16_f32 (round to nearest even)
  const f32x2 v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}

// (x0, x1) -> packed bf16 pairs of the three pieces; the subtractions are exact in fp32
__device__ __forceinline__ void split_pair(float x0, float x1, unsigned& h, unsigned& m, unsigned& l) {
  h = cvt_pk_bf16(x0, x1);
  const float r0 = x0 - __uint_as_float(h << 16), r1 = x1 - __uint_as_float(h & 0xffff0000u);
  m = cvt_pk_bf16(r0, r1);
  const float s0 = r0 - __uint_as_float(m << 16), s1 = r1 - __uint_as_float(m & 0xffff0000u);
  l = cvt_pk_bf16(s0, s1);
}

template <int BN, bool PRE>
__global__ __launch_bounds__(256, 3) void conv1x1_bf16x6_kernel(const XArgs g) {
  constexpr int NT = BN / 64;
  constexpr int kPlaneB = BN * 2 * 16;
  constexpr int kImgA = 3 * kPlaneA, kImgB = 3 * kPlaneB;
  constexpr int kStage = kImgA + kImgB;
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * kStage];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int li = lane & 31, lh = lane >> 5;
  const int id = xcd_remap(blockIdx.x, g.tiles_m * g.tiles_n);
  const int tm = (int)fast_div((unsigned)id, g.tn_mul, g.tn_sh), tn = id - tm * g.tiles_n;
  const int m0 = tm * XBM, n0 = tn * BN;
  const int nk = g.Cin / XBK;

  // ---- A staging: thread (row, cell)
  const int arow = tid >> 1, ac = tid & 1;
  const int am = m0 + arow;
  const bool a_ok = am < g.M;
  const unsigned avoff = (x_row_offset(g, a_ok ? am : g.M - 1) + 8u * ac) * 4u;
  const unsigned awr = x_cell(arow, ac);
  const float lo = g.relu_in ? 0.f : -__builtin_inff();
  // ---- B: the image of (tn, kt) is kImgB contiguous bytes = the LDS image; thread t moves the
  // 16-B cells t, t + 256, ... through registers (NOT by LDS-DMA: see the note at the end of the header)
  const float* sA = g.x;
  const float* sS = g.in_scale;
  const float* sT = g.in_shift;

  f32x4 a0, a1, sc0, sc1, sh0, sh1;
  constexpr int NBR = (kImgB / 16 + 255) / 256;     // 16-B cells of the B image per thread
  f32x4 bq[NBR];
  const float* sBr = reinterpret_cast<const float*>(g.wimg) + ((long)tn * nk) * (kImgB / 4);
  auto issue = [&](int stage) {
#pragma unroll
    for (int q = 0; q < NBR; ++q) {
      const int cell = tid + 256 * q;
      gload16(bq[q], sBr, (unsigned)((cell < kImgB / 16 ? cell : 0) * 16));
    }
    sBr += kImgB / 4;
    gload16(a0, sA, avoff);
    gload16(a1, sA + 4, avoff);
    if (PRE) {
      gload16(sc0, sS, (unsigned)(32 * ac));
      gload16(sc1, sS + 4, (unsigned)(32 * ac));
      gload16(sh0, sT, (unsigned)(32 * ac));
      gload16(sh1, sT + 4, (unsigned)(32 * ac));
      sS += XBK;
      sT += XBK;
    }
    sA += XBK;
  };
  auto store = [&](int stage) {
    // every load and DMA of this tile has landed after this wait (vmcnt retires in order); the "+v"
    // operands make the loaded registers defined HERE for the compiler
    if (PRE)
      asm volatile("s_waitcnt vmcnt(0)" : "+v"(a0), "+v"(a1), "+v"(sc0), "+v"(sc1), "+v"(sh0), "+v"(sh1)::"memory");
    else
      asm volatile("s_waitcnt vmcnt(0)" : "+v"(a0), "+v"(a1)::"memory");
#pragma unroll
    for (int q = 0; q < NBR; ++q) {
      asm volatile("" : "+v"(bq[q]));       // (loaded by asm: defined for the compiler from here on)
      const int cell = tid + 256 * q;
      if (cell < kImgB / 16) *reinterpret_cast<f32x4*>(lds + stage * kStage + kImgA + cell * 16) = bq[q];
    }
    float x[8] = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
    if (PRE) {
      const float s[8] = {sc0[0], sc0[1], sc0[2], sc0[3], sc1[0], sc1[1], sc1[2], sc1[3]};
      const float t[8] = {sh0[0], sh0[1], sh0[2], sh0[3], sh1[0], sh1[1], sh1[2], sh1[3]};
#pragma unroll
      for (int i = 0; i < 8; ++i) x[i] = fmaxf(fmaf(x[i], s[i], t[i]), lo);
    }
    if (!a_ok) {
#pragma unroll
      for (int i = 0; i < 8; ++i) x[i] = 0.f;      // rows past M: zero products, zero statistics
    }
    u32x4 ph, pm, pl;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      unsigned h, m, l;
      split_pair(x[2 * i], x[2 * i + 1], h, m, l);
      ph[i] = h; pm[i] = m; pl[i] = l;
    }
    unsigned char* d = lds + stage * kStage + awr;
    *reinterpret_cast<u32x4*>(d) = ph;
    *reinterpret_cast<u32x4*>(d + kPlaneA) = pm;
    *reinterpret_cast<u32x4*>(d + 2 * kPlaneA) = pl;
  };

  f32x16 acc[2][NT];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

  // fragment addresses: cell `lh` of row (wave's 64 rows + li); + 32 rows = 2 row groups = 1024 B
  const unsigned char* a_rd = lds + x_cell(wm * 64 + li, lh);
  const unsigned char* b_rd = lds + kImgA + x_cell(wn * (BN / 2) + li, lh);
  auto compute = [&](int stage) {
    bf16x8 af[2][3], bf[NT][3];
#pragma unroll
    for (int p = 0; p < 3; ++p) {
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
        af[mt][p] = *reinterpret_cast<const bf16x8*>(a_rd + stage * kStage + p * kPlaneA + mt * 1024);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
        bf[nt][p] = *reinterpret_cast<const bf16x8*>(b_rd + stage * kStage + p * kPlaneB + nt * 1024);
    }
    // small terms first: (l h', h l', m m'), (m h', h m'), h h'
    constexpr int PA[6] = {2, 0, 1, 1, 0, 0};
    constexpr int PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
    for (int term = 0; term < 6; ++term)
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[mt][PA[term]], bf[nt][PB[term]], acc[mt][nt], 0, 0, 0);
  };

  issue(0);
  store(0);
  __syncthreads();
  int kt = 0;
  for (; kt + 2 <= nk; kt += 2) {
    issue(1);
    __builtin_amdgcn_sched_barrier(0);
    compute(0);
    __builtin_amdgcn_sched_barrier(0);    // (without it hipcc pulls store()'s vmcnt(0) up behind the second MFMA)
    store(1);
    __syncthreads();
    const bool more = kt + 2 < nk;
    if (more) issue(0);
    __builtin_amdgcn_sched_barrier(0);
    compute(1);
    __builtin_amdgcn_sched_barrier(0);
    if (more) store(0);
    __syncthreads();
  }
  if (kt < nk) compute(0);

  // ---- epilogue: raw output (D layout: column = lane & 31, rows (r & 3) + 8 (r >> 2) + 4 lh)
  const bool ragged = m0 + XBM > g.M;
  const unsigned rstep = (unsigned)g.Cout * 4u;
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
        if (!ragged || row < g.M) {
          float v = acc[mt][nt][r];
          if (g.out_scale) {
            v = fmaf(v, osc, osh);
            if (g.res) v += *reinterpret_cast<const float*>(reinterpret_cast<const char*>(g.res) + off);
            if (g.relu_out) v = fmaxf(v, 0.f);
          }
          asm volatile("global_store_dword %0, %1, %2" ::"v"(off), "v"(v), "s"(g.y) : "memory");
        }
        if ((r & 3) == 3) { row += 5; off += 5u * rstep; } else { row += 1; off += rstep; }
      }
    }
  }
  if (g.part_sum) {
    using T = TileCfg<XBM, BN, 16>;
    __syncthreads();
    block_col_stats<T>(acc, reinterpret_cast<float*>(lds), g.part_sum + (long)tm * g.Cout,
                       g.part_sq + (long)tm * g.Cout, n0, g.Cout);
  }
}

// ---- persistent version -------------------------------------------------------------------------
// Same tile, LDS image and MFMA order; what changes is the pipeline around them (SQ counters of the
// kernel above on 14x14x256 -> 1024: matrix pipe busy 30 % of the time, waves waiting on vmcnt half
// of theirs -- one k-tile of MFMAs, 0.3 us, does not cover a global load issued one tile ahead):
//   * a workgroup walks tiles w = blockIdx.x, + gridDim.x, ...; the (tile, k-tile) steps form ONE
//     software pipeline, so the first loads of the next tile are in flight during the last k-tiles and
//     the epilogue of this one;
//   * loads run TWO steps ahead in two register sets; waits are counted (vmcnt retires in issue
//     order, stores included): vmcnt(NLD) leaves exactly the younger set in flight. After an epilogue
//     of 64 stores the next two waits use vmcnt(63) (the wanted loads are older than the stores), the
//     third finds the stores retired;
//   * fold + split of step s+1 (VALU) is interleaved with the MFMAs of step s by the scheduler hints.
template <int NBR>
struct XRegs {
  f32x4 a0, a1, b[NBR];
};
constexpr int kFoldMax = 512;      // input channels whose BatchNorm scale / shift the persistent kernel keeps in LDS

template <int N> __device__ __forceinline__ void x_wait_vmcnt_plain() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int BN, bool PRE>
__global__ __launch_bounds__(256, 2) void conv1x1_bf16x6_p_kernel(const XArgs g) {
  constexpr int NT = BN / 64;
  constexpr int kPlaneB = BN * 2 * 16;
  constexpr int kImgA = 3 * kPlaneA, kImgB = 3 * kPlaneB;
  constexpr int kStage = kImgA + kImgB;
  constexpr int NBR = (kImgB / 16 + 255) / 256;     // 16-B cells of the B image per thread
  constexpr int NLD = NBR + 2;                      // loads of one step
  // the wait that must leave one step's loads AND a plain epilogue's 32 NT stores in flight (6-bit counter)
  constexpr int kWaitEpi = NLD + 32 * NT > 63 ? 63 : NLD + 32 * NT;
  constexpr int kFold = PRE ? 2 * kFoldMax * 4 : 0;
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * kStage + 4 * BN * 4 + kFold];
  float* scratch = reinterpret_cast<float*>(lds + 2 * kStage);
  // the previous BatchNorm's scale | shift, once per workgroup (every step re-read them through the
  // vector memory path before: 4 of 9 loads, 16 KB of the 36 KB a step moved into registers)
  const float* fold = reinterpret_cast<const float*>(lds + 2 * kStage + 4 * BN * 4);
  __shared__ long long st[4 * 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int li = lane & 31, lh = lane >> 5;
  const bool diag = g.stamps != nullptr && blockIdx.x == 0;
  auto stamp = [&](int it, int j) {
    if (diag && it < 64) {
      const long long t = __builtin_readcyclecounter();
      if (tid == 0) st[4 * it + j] = t;
    }
  };
  const int total = g.tiles_m * g.tiles_n, G = (int)gridDim.x;
  const int nk = g.Cin / XBK;
  const int my_tiles = (total - 1 - (int)blockIdx.x) / G + 1;
  const int n_it = my_tiles * nk;

  const int arow = tid >> 1, ac = tid & 1;
  const unsigned awr = x_cell(arow, ac);
  const float lo = g.relu_in ? 0.f : -__builtin_inff();

  // ---- issue cursor: two steps ahead of the MFMAs
  int iw = (int)blockIdx.x, ikt = 0;
  unsigned i_avoff = 0;
  const float* i_sA = g.x;
  const float* i_sB = nullptr;
  auto i_tile = [&]() {
    const int id = xcd_remap(iw, total);
    const int tm = (int)fast_div((unsigned)id, g.tn_mul, g.tn_sh), tn = id - tm * g.tiles_n;
    const int am = tm * XBM + arow;
    i_avoff = (x_row_offset(g, am < g.M ? am : g.M - 1) + 8u * ac) * 4u;   // rows past M: a valid row, zeroed in the epilogue
    i_sA = g.x;
    i_sB = reinterpret_cast<const float*>(g.wimg) + ((long)tn * nk) * (kImgB / 4);
  };
  i_tile();
  if (PRE) {
    float* f = reinterpret_cast<float*>(lds + 2 * kStage + 4 * BN * 4);
    for (int i = tid; i < g.Cin; i += 256) {
      f[i] = g.in_scale[i];
      f[kFoldMax + i] = g.in_shift[i];
    }
    __syncthreads();
  }
  auto issue = [&](XRegs<NBR>& R) {
#pragma unroll
    for (int q = 0; q < NBR; ++q) {
      const int cell = tid + 256 * q;
      gload16(R.b[q], i_sB, (unsigned)((cell < kImgB / 16 ? cell : 0) * 16));
    }
    gload16(R.a0, i_sA, i_avoff);
    gload16(R.a1, i_sA + 4, i_avoff);
    i_sB += kImgB / 4;
    i_sA += XBK;
    if (++ikt == nk) {          // next tile (past the last one: the same tile again, loads nobody uses)
      ikt = 0;
      if (iw + G < total) iw += G;
      i_tile();
    }
  };
  // the registers of R become defined for the compiler HERE (they were written by asm loads)
  auto landed = [&](XRegs<NBR>& R) {
    asm volatile("" : "+v"(R.a0), "+v"(R.a1)::"memory");
#pragma unroll
    for (int q = 0; q < NBR; ++q) asm volatile("" : "+v"(R.b[q]));
  };
  int st_k = 8 * ac;           // this thread's first channel of the step being staged
  auto store = [&](XRegs<NBR>& R, int stage) {
#pragma unroll
    for (int q = 0; q < NBR; ++q) {
      const int cell = tid + 256 * q;
      if ((kImgB / 16) % 256 == 0 || cell < kImgB / 16)
        *reinterpret_cast<f32x4*>(lds + stage * kStage + kImgA + cell * 16) = R.b[q];
    }
    float x[8] = {R.a0[0], R.a0[1], R.a0[2], R.a0[3], R.a1[0], R.a1[1], R.a1[2], R.a1[3]};
    if (PRE) {
      const f32x4 sc0 = *reinterpret_cast<const f32x4*>(fold + st_k), sc1 = *reinterpret_cast<const f32x4*>(fold + st_k + 4);
      const f32x4 sh0 = *reinterpret_cast<const f32x4*>(fold + kFoldMax + st_k);
      const f32x4 sh1 = *reinterpret_cast<const f32x4*>(fold + kFoldMax + st_k + 4);
      st_k += XBK;
      if (st_k >= g.Cin) st_k -= g.Cin;
      const float s[8] = {sc0[0], sc0[1], sc0[2], sc0[3], sc1[0], sc1[1], sc1[2], sc1[3]};
      const float t[8] = {sh0[0], sh0[1], sh0[2], sh0[3], sh1[0], sh1[1], sh1[2], sh1[3]};
#pragma unroll
      for (int i = 0; i < 8; ++i) x[i] = fmaxf(fmaf(x[i], s[i], t[i]), lo);
    }
    u32x4 ph, pm, pl;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      unsigned h, m, l;
      split_pair(x[2 * i], x[2 * i + 1], h, m, l);
      ph[i] = h; pm[i] = m; pl[i] = l;
    }
    unsigned char* d = lds + stage * kStage + awr;
    *reinterpret_cast<u32x4*>(d) = ph;
    *reinterpret_cast<u32x4*>(d + kPlaneA) = pm;
    *reinterpret_cast<u32x4*>(d + 2 * kPlaneA) = pl;
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

  const unsigned char* a_rd = lds + x_cell(wm * 64 + li, lh);
  const unsigned char* b_rd = lds + kImgA + x_cell(wn * (BN / 2) + li, lh);
  // One step = the 12 NT MFMAs of LDS[stage] + fold and split of the next step's A cells (R -> LDS[1 - stage]),
  // laid out by hand:
  //   * the fragment reads go first, and the LAST two terms of the PREVIOUS step (h m', h h': 4 NT MFMAs whose
  //     three planes stayed in registers, XTail) run while they are on their way -- a lone workgroup on a CU
  //     otherwise idles its matrix pipe through every LDS round trip behind the barrier (clock stamps: 1 150
  //     cycles for 768 of MFMA);
  //   * then 12 phases of one or two MFMAs and the 5 VALU of one third of a pair's split (an MFMA holds the
  //     vector issue for 8 of its 32 cycles; 5 more instructions of 4 fit in the rest). sched_barrier(0) pins
  //     the phases and a volatile use pins each phase's VALU (pure IR otherwise sinks to the final ds_write;
  //     sched_group_barrier hints lost the pattern as soon as the fold read its scale from LDS);
  //   * the last two MFMAs of the head cover the A-cell writes and the issue of the loads two steps ahead.
  struct XTail { bf16x8 ah[2], bm[NT], bh[NT]; };
  auto tail = [&](const XTail& T) {
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(T.ah[mt], T.bm[nt], acc[mt][nt], 0, 0, 0);
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(T.ah[mt], T.bh[nt], acc[mt][nt], 0, 0, 0);
  };
  auto body = [&](XRegs<NBR>& R, int stage, XTail& Tc, const XTail& Tp, bool pending, bool do_issue) {
    constexpr int PA[4] = {2, 0, 1, 1};       // small terms first: (l h', h l', m m'), (m h', [tail: h m']), [h h']
    constexpr int PB[4] = {0, 2, 1, 0};
    f32x4 sc0, sc1, sh0, sh1;
    if (PRE) {
      sc0 = *reinterpret_cast<const f32x4*>(fold + st_k); sc1 = *reinterpret_cast<const f32x4*>(fold + st_k + 4);
      sh0 = *reinterpret_cast<const f32x4*>(fold + kFoldMax + st_k);
      sh1 = *reinterpret_cast<const f32x4*>(fold + kFoldMax + st_k + 4);
      st_k += XBK;
      if (st_k >= g.Cin) st_k -= g.Cin;
    }
    bf16x8 af[2][3], bf[NT][3];
#pragma unroll
    for (int o = 0; o < 3; ++o) {
      const int pa = o == 0 ? 2 : o == 1 ? 0 : 1, pb = o == 0 ? 0 : o == 1 ? 2 : 1;   // in the order the terms need them
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
        af[mt][pa] = *reinterpret_cast<const bf16x8*>(a_rd + stage * kStage + pa * kPlaneA + mt * 1024);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
        bf[nt][pb] = *reinterpret_cast<const bf16x8*>(b_rd + stage * kStage + pb * kPlaneB + nt * 1024);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (pending) tail(Tp);
    __builtin_amdgcn_sched_barrier(0);
    float x[8] = {R.a0[0], R.a0[1], R.a0[2], R.a0[3], R.a1[0], R.a1[1], R.a1[2], R.a1[3]};
    const float fs[8] = {sc0[0], sc0[1], sc0[2], sc0[3], sc1[0], sc1[1], sc1[2], sc1[3]};
    const float ft[8] = {sh0[0], sh0[1], sh0[2], sh0[3], sh1[0], sh1[1], sh1[2], sh1[3]};
    u32x4 ph, pm, pl;
    float r0 = 0.f, r1 = 0.f;
    unsigned char* const dB = lds + (1 - stage) * kStage + kImgA;
    constexpr int HP = 8 * NT - 2;          // head MFMAs spread over the phases (the last two cover writes + issue)
    auto head = [&](int idx) {
      const int term = idx / (2 * NT), mt = (idx % (2 * NT)) / NT, nt = idx % NT;
      acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[mt][PA[term]], bf[nt][PB[term]], acc[mt][nt], 0, 0, 0);
    };
#pragma unroll
    for (int phase = 0; phase < 12; ++phase) {
      const int p = phase / 3, sub = phase % 3;
      if (sub == 0) {
        if (PRE) {
          x[2 * p] = fmaxf(fmaf(x[2 * p], fs[2 * p], ft[2 * p]), lo);
          x[2 * p + 1] = fmaxf(fmaf(x[2 * p + 1], fs[2 * p + 1], ft[2 * p + 1]), lo);
        }
        unsigned h = cvt_pk_bf16(x[2 * p], x[2 * p + 1]);
        asm volatile("" : "+v"(h));
        ph[p] = h;
      } else if (sub == 1) {
        r0 = x[2 * p] - __uint_as_float(ph[p] << 16);
        r1 = x[2 * p + 1] - __uint_as_float(ph[p] & 0xffff0000u);
        unsigned m = cvt_pk_bf16(r0, r1);
        asm volatile("" : "+v"(m), "+v"(r0), "+v"(r1));
        pm[p] = m;
      } else {
        r0 -= __uint_as_float(pm[p] << 16);
        r1 -= __uint_as_float(pm[p] & 0xffff0000u);
        unsigned l = cvt_pk_bf16(r0, r1);
        asm volatile("" : "+v"(l));
        pl[p] = l;
      }
#pragma unroll
      for (int idx = (phase * HP) / 12; idx < ((phase + 1) * HP) / 12; ++idx) head(idx);
      if (phase >= 1 && phase <= NBR) {
        const int q = phase - 1;
        const int cell = tid + 256 * q;
        if ((kImgB / 16) % 256 == 0 || cell < kImgB / 16) *reinterpret_cast<f32x4*>(dB + cell * 16) = R.b[q];
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    head(HP);
    unsigned char* d = lds + (1 - stage) * kStage + awr;
    *reinterpret_cast<u32x4*>(d) = ph;
    *reinterpret_cast<u32x4*>(d + kPlaneA) = pm;
    *reinterpret_cast<u32x4*>(d + 2 * kPlaneA) = pl;
    __builtin_amdgcn_sched_barrier(0);
    head(HP + 1);
    __builtin_amdgcn_sched_barrier(0);
    if (do_issue) issue(R);
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) Tc.ah[mt] = af[mt][0];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) { Tc.bm[nt] = bf[nt][1]; Tc.bh[nt] = bf[nt][0]; }
  };

  // ---- compute cursor
  int cw = (int)blockIdx.x, ckt = 0;
  int after_epi = 0;         // waits still to be taken with the epilogue's 64 stores counted in
  auto epilogue = [&]() {
    const int id = xcd_remap(cw, total);
    const int tm = (int)fast_div((unsigned)id, g.tn_mul, g.tn_sh), tn = id - tm * g.tiles_n;
    const int m0 = tm * XBM, n0 = tn * BN;
    const bool ragged = m0 + XBM > g.M;
    const bool plain = !ragged && !g.out_scale;
    const unsigned rstep = (unsigned)g.Cout * 4u;
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
    if (plain && (g.abl & 8)) {
    } else if (plain) {
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
            asm volatile("global_store_dword %0, %1, %2" ::"v"(off), "v"(acc[mt][nt][r]), "s"(g.y) : "memory");
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
              asm volatile("global_store_dword %0, %1, %2" ::"v"(off), "v"(v), "s"(g.y) : "memory");
            }
            if ((r & 3) == 3) { row += 5; off += 5u * rstep; } else { row += 1; off += rstep; }
          }
        }
      }
    }
    if (g.part_sum && !(g.abl & 4)) {
      using T = TileCfg<XBM, BN, 16>;
      block_col_stats<T>(acc, scratch, g.part_sum + (long)tm * g.Cout, g.part_sq + (long)tm * g.Cout, n0, g.Cout);
      __syncthreads();       // scratch is reused by the next tile's statistics
    }
    zero_acc();
    // a plain tile put exactly 32 NT stores behind the loads in flight; anything else (ragged rows,
    // the folded epilogue's own loads) is not counted on: the next wait drains the queue
    after_epi = (plain && !(g.abl & 8)) ? 2 : -1;
  };

  // Loads are only ever issued for steps that exist: a register written by a load nobody consumes is free
  // for the compiler to reuse at once, and the load would land in whatever lives there by then.
  XRegs<NBR> R0, R1;
  issue(R0);
  if (n_it > 1) {
    issue(R1);
    x_wait_vmcnt_plain<NLD>();
  } else {
    x_wait_vmcnt_plain<0>();
  }
  landed(R0);
  store(R0, 0);
  if (n_it > 2) issue(R0);
  __syncthreads();

  // nk is even (the launcher sends odd k-tile counts to the one-tile kernel), so a tile ends only behind
  // the second step of a pair: one copy of the epilogue. landed + store are unconditional -- the last
  // step of a workgroup re-stores stale registers into the stage nobody reads again -- because a branch
  // there splits the block and the scheduler can no longer put the split's VALU between the MFMAs.
  XTail T0, T1;
  auto step = [&](XRegs<NBR>& R, int stage, int it, XTail& Tc, const XTail& Tp, bool second) {
    // LDS[stage] holds step `it`; R holds step it + 1 (issued two steps ago) and goes to LDS[1 - stage]
    stamp(it, 0);
    if (it + 2 >= n_it) { x_wait_vmcnt_plain<0>(); after_epi = 0; }     // nothing younger in flight
    else if (after_epi > 0) { x_wait_vmcnt_plain<kWaitEpi>(); --after_epi; }
    else if (after_epi < 0) { x_wait_vmcnt_plain<0>(); after_epi = 0; }
    else x_wait_vmcnt_plain<NLD>();
    stamp(it, 1);
    landed(R);
    // a tile starts with the first step of a pair only (nk is even): the second always has a tail to run
    body(R, stage, Tc, Tp, second || ckt > 0, it + 3 < n_it);
    stamp(it, 2);
    __syncthreads();
    stamp(it, 3);
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
  if (diag) {
    __syncthreads();
    if (tid < 64) {
      for (int j = 0; j < 4; ++j) g.stamps[4 * tid + j] = st[4 * tid + j];
    }
  }
  x_wait_vmcnt_plain<0>();       // nothing of this workgroup may still be in flight when its LDS is handed on
}

// One thread per (tn, kt, plane, row, pos): 8 consecutive k of output channel n -> one 16-B cell.
template <int BN>
__global__ __launch_bounds__(256) void conv1x1_bf16x6_pack_kernel(const float* __restrict__ w, unsigned* __restrict__ img,
                                                                  int Cout, int Cin) {
  const int nk = Cin / XBK, tiles_n = Cout / BN;
  const long cells = (long)tiles_n * nk * 3 * BN * 2;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < cells; i += (long)gridDim.x * blockDim.x) {
    long r = i;
    const int pos = (int)(r & 1); r >>= 1;
    const int row = (int)(r % BN); r /= BN;
    const int plane = (int)(r % 3); r /= 3;
    const int kt = (int)(r % nk);
    const int tn = (int)(r / nk);
    const int c = pos ^ (((row & 15) >> 3) & 1);
    const float* src = w + (long)(tn * BN + row) * Cin + kt * XBK + 8 * c;
    unsigned out[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float x0 = src[2 * q], x1 = src[2 * q + 1];
      // same split as the device staging code, written with plain conversions
      auto bf = [](float v) { return __uint_as_float(((__float_as_uint(v) + 0x7fffu + ((__float_as_uint(v) >> 16) & 1u)) & 0xffff0000u)); };
      const float h0 = bf(x0), h1 = bf(x1);
      const float m0 = bf(x0 - h0), m1 = bf(x1 - h1);
      const float l0 = bf(x0 - h0 - m0), l1 = bf(x1 - h1 - m1);
      const float p0 = plane == 0 ? h0 : plane == 1 ? m0 : l0;
      const float p1 = plane == 0 ? h1 : plane == 1 ? m1 : l1;
      out[q] = (__float_as_uint(p0) >> 16) | (__float_as_uint(p1) & 0xffff0000u);
    }
    // cell index inside the image of (tn, kt): [plane][row][pos]
    unsigned* dst = img + (((long)(tn * nk + kt) * 3 + plane) * BN * 2 + (long)row * 2 + pos) * 4;
    dst[0] = out[0]; dst[1] = out[1]; dst[2] = out[2]; dst[3] = out[3];
  }
}

// 128-wide tiles wherever Cout allows: the A tile is folded and split once per output tile, so its
// VALU cost per MFMA halves (pipelined step: 5928 vs 5631 images/s with 64-wide tiles everywhere)
int x_pick_bn(int M, int Cout) {
  const char* f = getenv("CAPNET_X6_BN");     // diagnostics: force 64
  if (f && f[0] == '6') return 64;
  (void)M;
  return Cout % 128 == 0 ? 128 : 64;
}

}  // namespace

bool conv1x1_bf16x6_eligible(const float* x, long sxb, long sxh, long sxw, long sxc, int Bn, int H, int W,
                             int Cin, int Cout, int stride, const float* in_scale, const float* in_shift) {
  const long OH = (H - 1) / stride + 1, OW = (W - 1) / stride + 1;
  return sxc == 1 && Cin % XBK == 0 && Cout % 64 == 0 && aligned16(x) && sxb % 4 == 0 && sxh % 4 == 0 &&
         sxw % 4 == 0 && (long)Bn * sxb * 4 < (1l << 32) && (long)Bn * OH * OW < (1l << 24) &&
         (long)Bn * OH * OW * Cout * 4 < (1l << 32) &&
         (!in_scale || (aligned16(in_scale) && aligned16(in_shift)));
}

int conv1x1_bf16x6_tiles_m(long M) { return cdiv(M, XBM); }
int conv1x1_bf16x6_bn(long M, int Cout) { return x_pick_bn((int)M, Cout); }
size_t conv1x1_bf16x6_weight_words(int Cin, int Cout) { return (size_t)Cout * Cin * 3 / 2; }

// w [Cout][Cin] fp32 -> the split bf16 image for tile width bn
int conv1x1_bf16x6_pack(const float* w, unsigned* img, int Cout, int Cin, int bn, hipStream_t stream) {
  CAPNET_REQUIRE(w && img && Cin % XBK == 0 && (bn == 64 || bn == 128) && Cout % bn == 0 && aligned16(img),
                 "conv1x1_bf16x6_pack: bad argument (Cin=%d Cout=%d bn=%d)", Cin, Cout, bn);
  const long cells = (long)Cout * Cin * 3 / 8;
  const int grid = (int)(cdiv(cells, 256) > 4096 ? 4096 : cdiv(cells, 256));
  if (bn == 128) hipLaunchKernelGGL(conv1x1_bf16x6_pack_kernel<128>, dim3(grid), dim3(256), 0, stream, w, img, Cout, Cin);
  else hipLaunchKernelGGL(conv1x1_bf16x6_pack_kernel<64>, dim3(grid), dim3(256), 0, stream, w, img, Cout, Cin);
  CAPNET_LAUNCH_CHECK();
  return kOk;
}

int conv1x1_fwd_bf16x6(const float* x, long sxb, long sxh, long sxw, const unsigned* wimg, int bn, float* y,
                       const float* in_scale, const float* in_shift, int relu_in, float* part_sum,
                       float* part_sq, int Bn, int H, int W, int Cin, int Cout, int stride,
                       hipStream_t stream, const float* out_scale, const float* out_shift, const float* res,
                       int relu_out) {
  CAPNET_REQUIRE(x && wimg && y && stride >= 1, "conv1x1_fwd_bf16x6: bad argument");
  CAPNET_REQUIRE(conv1x1_bf16x6_eligible(x, sxb, sxh, sxw, 1, Bn, H, W, Cin, Cout, stride, in_scale, in_shift) &&
                     aligned16(wimg) && (bn == 64 || bn == 128) && Cout % bn == 0,
                 "conv1x1_fwd_bf16x6: operands not eligible");
  CAPNET_REQUIRE((in_scale == nullptr) == (in_shift == nullptr), "conv1x1_fwd_bf16x6: scale/shift pair");
  CAPNET_REQUIRE((part_sum == nullptr) == (part_sq == nullptr), "conv1x1_fwd_bf16x6: stats pair");
  XArgs a{};
  const int OH = (H - 1) / stride + 1, OW = (W - 1) / stride + 1;
  a.x = x; a.wimg = wimg; a.y = y; a.in_scale = in_scale; a.in_shift = in_shift;
  a.part_sum = part_sum; a.part_sq = part_sq;
  a.out_scale = out_scale; a.out_shift = out_shift; a.res = res; a.relu_out = relu_out;
  CAPNET_REQUIRE(!out_scale || (out_shift && !part_sum), "conv1x1_fwd_bf16x6: folded epilogue takes no statistics");
  { const char* e = getenv("CAPNET_X6_ABLATE"); a.abl = e ? atoi(e) : 0; }
  { const char* e = getenv("CAPNET_X6_STAMPS"); a.stamps = e ? reinterpret_cast<long long*>(strtoull(e, nullptr, 0)) : nullptr; }
  a.M = Bn * OH * OW; a.Cin = Cin; a.Cout = Cout; a.relu_in = relu_in;
  a.tiles_m = cdiv(a.M, XBM); a.tiles_n = Cout / bn;
  magic_div((unsigned)a.tiles_n, &a.tn_mul, &a.tn_sh);
  a.OW = OW; a.OHW = OH * OW; a.stride = stride;
  a.sxb = (int)sxb; a.sxh = (int)sxh; a.sxw = (int)sxw;
  magic_div((unsigned)(OH * OW), &a.ohw_mul, &a.ohw_sh);
  magic_div((unsigned)OW, &a.ow_mul, &a.ow_sh);
  const dim3 block(256);
  const char* v1 = getenv("CAPNET_X6_V1");         // A/B: the one-tile-per-workgroup kernel
  if (!(v1 && v1[0] == '1') && (Cin / XBK) % 2 == 0 && (!in_scale || Cin <= kFoldMax)) {
    // persistent: two workgroups per CU walk the tiles (a multiple of 8, so that a workgroup stays on its XCD's
    // share of the tile order)
    const char* ge = getenv("CAPNET_X6_WGS");
    const int cap = ge ? atoi(ge) : 512;
    const int total = a.tiles_m * a.tiles_n;
    const dim3 pgrid(total <= cap ? total : cap);
    if (bn == 128) {
      if (in_scale) hipLaunchKernelGGL((conv1x1_bf16x6_p_kernel<128, true>), pgrid, block, 0, stream, a);
      else hipLaunchKernelGGL((conv1x1_bf16x6_p_kernel<128, false>), pgrid, block, 0, stream, a);
    } else {
      if (in_scale) hipLaunchKernelGGL((conv1x1_bf16x6_p_kernel<64, true>), pgrid, block, 0, stream, a);
      else hipLaunchKernelGGL((conv1x1_bf16x6_p_kernel<64, false>), pgrid, block, 0, stream, a);
    }
    CAPNET_LAUNCH_CHECK();
    return kOk;
  }
  const dim3 grid(a.tiles_m * a.tiles_n);
  if (bn == 128) {
    if (in_scale) hipLaunchKernelGGL((conv1x1_bf16x6_kernel<128, true>), grid, block, 0, stream, a);
    else hipLaunchKernelGGL((conv1x1_bf16x6_kernel<128, false>), grid, block, 0, stream, a);
  } else {
    if (in_scale) hipLaunchKernelGGL((conv1x1_bf16x6_kernel<64, true>), grid, block, 0, stream, a);
    else hipLaunchKernelGGL((conv1x1_bf16x6_kernel<64, false>), grid, block, 0, stream, a);
  }
  CAPNET_LAUNCH_CHECK();
  return kOk;
}

}  // namespace capnet
