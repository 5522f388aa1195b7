// Implicit-GEMM convolution, second generation: built around one measured fact about gfx950 --
// v_mfma_f32_32x32x2_f32 runs on the SAME pipe as the f32 VALU (a co-resident wave's VALU work
// adds to MFMA time instead of hiding under it; tools/probes/native/coexec.hip measures 2.31 ms MFMA
// + 0.88 ms VALU = 3.01 ms together). So the k-loop is written to issue almost no VALU:
//   * weights are pre-packed K-MAJOR ([Kw][Cout]) and streamed straight into LDS by LDS-DMA
//     (global_load_lds_dwordx4): no VGPRs, no ds_write, no address arithmetic in the loop;
//   * activations are fetched with buffer_load_dwordx4 whose per-lane byte offset is constant
//     for a whole filter tap and whose k advance is a scalar soffset;
//   * the previous layer's BatchNorm+ReLU and the zero padding are 2 v_pk_fma + 4 v_med3 per
//     float4 (v_med3(x, 0, hi) with hi = +inf / 0 is ReLU and the padding mask in one op);
//   * every LDS address is one per-thread base + an immediate (k-loop unrolled over the two
//     LDS stages), fragments for k-step j+1 are read while the MFMAs of step j run.
// Same contract as conv_f32.hip (raw NHWC output + per-workgroup column sums for the batch
// statistics); requires Cin % 16 == 0, channel-contiguous input and Cout % BN == 0.
#include <algorithm>
#include "common.h"
#include "mfma_core.h"
#include "kernels.h"

namespace capnet {

struct ConvArgs2 {
  const float* x;
  const float* wk;  // [Kw][Cout]
  float* y;
  const float* in_scale;
  const float* in_shift;
  float* part_sum;
  float* part_sq;
  int H, W, Cin, OH, OW, Cout, KW, stride, pad;
  int sxb, sxh, sxw;
  int M, Kw;
  int relu_in;
  int tiles_m, tiles_n;
  unsigned x_bytes, ss_bytes;
  // tail balancing: workgroups [0, full_tiles) compute whole tiles; the remaining `tiles -
  // full_tiles` tiles are each cut into `split` K-slices of `kps` k-tiles that write fp32 partial
  // slabs ([rem_tile][slice][BM*BN]) summed by conv_tail_fixup_kernel
  int full_tiles, split, kps;
  float* slabs;
  // inference epilogue (EPI instantiation only): y = act(acc * out_scale[n] + out_shift[n] + res)
  // -- the BatchNorm that follows the conv folded in from its running statistics, the residual
  // add and the ReLU of a bottleneck's tail -- instead of raw output + batch statistics
  const float* out_scale;
  const float* out_shift;
  const float* res;
  int relu_out;
  // exact n / d for n < 2^24 as (n * mul) >> sh (host: magic_div): the per-thread pixel
  // decomposition m -> (image, oh, ow) costs 3 VALU per division instead of ~35
  unsigned ohw_mul, ohw_sh, ow_mul, ow_sh;
  unsigned tn_mul, tn_sh, cin_mul, cin_sh, kw_mul, kw_sh, split_mul, split_sh;  // tiles_n, Cin, KW, split
};

// MODE (compile time, chosen by the launcher): bit 0 PRE = the previous BatchNorm (+ReLU) is
// applied while the A tile is staged; bit 1 MASK = some staged element must be forced to 0
// (padding taps of a 3x3, rows past M in a ragged last M tile). As runtime flags hipcc
// if-converted both paths: every k-tile paid fma + med3 + 8 v_cndmask whether it needed them or
// not (measured 2.5-4 VALU instructions per MFMA, and f32 MFMA shares its pipe with the VALU).
// Specialised: a 1x1 conv on an activated input stages with no VALU at all, conv3 with fma + max,
// conv2 with fma + med3.
template <int BM, int BN, bool EPI, int MODE>
__global__ __launch_bounds__(kGemmThreads) void conv_f32_v2_kernel(ConvArgs2 g) {
  constexpr int BK = 16;
  constexpr int LDA = BM + 4;
  constexpr int A_ST = BK * LDA;  // floats per A stage (k-major, padded: transposing writes)
  constexpr int B_ST = BK * BN;   // floats per B stage (k-major, unpadded: filled by LDS-DMA)
  constexpr int MT = BM / 64, NT = BN / 64;
  constexpr int PASSES = BM / 64;  // A rows per thread (4 threads per row)
  constexpr int LPR = BN / 4;      // lanes per k-row of the B tile
  constexpr int RPI = 64 / LPR;    // k-rows per DMA instruction
  constexpr int NI = BK / RPI / 4; // DMA instructions per wave and tile
  static_assert(NI >= 1, "B tile too small for 4 waves");
  __shared__ __attribute__((aligned(16))) float lds[2 * A_ST + 2 * B_ST];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int li = lane & 31, lh = lane >> 5;
  const int nk = g.Kw / BK;
  int id, kt0 = 0, kt1 = nk, slice = 0;
  const bool partial = (int)blockIdx.x >= g.full_tiles;
  if (!partial) {
    id = xcd_remap(blockIdx.x, g.full_tiles);
  } else {
    const int u = blockIdx.x - g.full_tiles;
    const int uq = (int)fast_div((unsigned)u, g.split_mul, g.split_sh);
    id = g.full_tiles + uq;
    slice = u - uq * g.split;
    kt0 = slice * g.kps;
    kt1 = min(nk, kt0 + g.kps);
  }
  const int tm = (int)fast_div((unsigned)id, g.tn_mul, g.tn_sh), tn = id - tm * g.tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;

  // ---- A staging geometry (fixed per thread) ----
  const int kc = tid & 3, rl = tid >> 2;
  int boff[PASSES], ih0[PASSES], iw0[PASSES];
  {
    const int ohw = g.OH * g.OW;
#pragma unroll
    for (int ps = 0; ps < PASSES; ++ps) {
      const int m = m0 + rl + ps * 64;
      const int mc = m < g.M ? m : g.M - 1;
      const int b = (int)fast_div((unsigned)mc, g.ohw_mul, g.ohw_sh);
      const int rem = mc - b * ohw;
      const int oh = (int)fast_div((unsigned)rem, g.ow_mul, g.ow_sh);
      const int ow = rem - oh * g.OW;
      ih0[ps] = m < g.M ? oh * g.stride - g.pad : -(1 << 20);  // rows past M: never in the image
      iw0[ps] = ow * g.stride - g.pad;
      boff[ps] = b * g.sxb + 4 * kc;
    }
  }
  constexpr bool pre = (MODE & 1) != 0;
  const float inf = __builtin_inff();
  const float lo = g.relu_in ? 0.f : -inf;  // v_med3(x, lo, hi): relu iff lo == 0

  // need_mask: some staged element must become 0 (padding taps; rows past M in a ragged last
  // M tile). Otherwise store() issues no VALU at all besides the prologue's fma.
  constexpr bool need_mask = (MODE & 2) != 0;
  unsigned voff[PASSES];         // byte offset of this thread's float4 for the current tap
  float hi[PASSES], lw[PASSES];  // v_med3(x, lw, hi): (lo, +inf) inside the image, (0, 0) outside
  int tap = (int)fast_div((unsigned)(kt0 * BK), g.cin_mul, g.cin_sh), c0 = kt0 * BK - tap * g.Cin;
  auto set_tap = [&](int t) {
    const int r = (int)fast_div((unsigned)t, g.kw_mul, g.kw_sh), s = t - r * g.KW;
#pragma unroll
    for (int ps = 0; ps < PASSES; ++ps) {
      const int ih = ih0[ps] + r, iw = iw0[ps] + s;
      const bool inb = (unsigned)ih < (unsigned)g.H && (unsigned)iw < (unsigned)g.W;
      const int ihc = min(max(ih, 0), g.H - 1), iwc = min(max(iw, 0), g.W - 1);
      voff[ps] = (unsigned)(boff[ps] + ihc * g.sxh + iwc * g.sxw) * 4u;
      if (need_mask) {
        hi[ps] = inb ? inf : 0.f;
        lw[ps] = inb ? lo : 0.f;
      }
    }
  };
  set_tap(tap);

  // ---- B DMA geometry ----
  int bsrc[NI];  // byte offset of this lane's 16 B inside the [Kw][Cout] matrix, tile k0 = 0
#pragma unroll
  for (int q = 0; q < NI; ++q) {
    const int krow = (wave * NI + q) * RPI + lane / LPR;
    bsrc[q] = (krow * g.Cout + n0 + 4 * (lane % LPR)) * 4;   // bytes
  }
  const float* wk = g.wk;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const unsigned lds_b0 = __builtin_amdgcn_readfirstlane(
      (unsigned)(size_t)(__attribute__((address_space(3))) float*)(lds + 2 * A_ST));

  f32x4 av[PASSES];
  f32x4 scv, shv;
  auto issue = [&](int stage, int kt) {
    // B first: an LDS-DMA issued while VGPR-destination loads are outstanding makes hipcc drain them
    const unsigned bdst = lds_b0 + (unsigned)(stage * B_ST + wave_u * NI * RPI * BN) * 4u;
    const float* bs = wk + (long)kt * BK * g.Cout;   // uniform: stays in SGPRs
#pragma unroll
    for (int q = 0; q < NI; ++q) glds16(bs, bsrc[q], bdst + (unsigned)(q * RPI * BN) * 4u);
    // uniform base (SGPR pair, advanced by scalar adds) + per-lane 32-bit offset that only
    // changes with the filter tap: global_load_dwordx4 v, v_off, s[base] -- no VALU per tile
    const float* xb = g.x + c0;
#pragma unroll
    for (int ps = 0; ps < PASSES; ++ps) gload16(av[ps], xb, voff[ps]);
    if (pre) {
      gload16(scv, g.in_scale + c0, (unsigned)(16 * kc));
      gload16(shv, g.in_shift + c0, (unsigned)(16 * kc));
    }
    // advance (tap, c0) for the next call
    c0 += BK;
    if (c0 >= g.Cin) {
      c0 = 0;
      ++tap;
      set_tap(tap);
    }
  };
  // hi/ok of the tile that was LOADED (set_tap above may already have moved on): keep a copy
  float hi_ld[PASSES], lw_ld[PASSES];
  auto snapshot = [&]() {
    if (!need_mask) return;
#pragma unroll
    for (int ps = 0; ps < PASSES; ++ps) {
      hi_ld[ps] = hi[ps];
      lw_ld[ps] = lw[ps];
    }
  };
  const int awr = (4 * kc) * LDA + rl;  // A store index inside a stage
  // Every load and DMA of the tile has landed after this wait (vmcnt retires in order); CAPNET_LANDED makes the
  // loaded registers defined HERE for the compiler. Called unconditionally, also where no tile was requested
  // (nothing in flight then): a wait under the same condition as the issue is correct, but the ISA checker
  // (tools/isa_inflight_check.py) follows every path and cannot know that two branches take the same side.
  auto landed = [&]() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (PASSES == 2) CAPNET_LANDED2(av[0], av[PASSES - 1]); else CAPNET_LANDED1(av[0]);
    if (pre) CAPNET_LANDED2(scv, shv);
  };
  auto store = [&](int stage) {
    float* d0 = lds + stage * A_ST + awr;
#pragma unroll
    for (int ps = 0; ps < PASSES; ++ps) {
      float x0 = av[ps].x, x1 = av[ps].y, x2 = av[ps].z, x3 = av[ps].w;
      if (pre) {
        x0 = fmaf(x0, scv.x, shv.x);
        x1 = fmaf(x1, scv.y, shv.y);
        x2 = fmaf(x2, scv.z, shv.z);
        x3 = fmaf(x3, scv.w, shv.w);
      }
      // relu (lw = 0, hi = inf), identity (lw = -inf, hi = inf) or forced zero for padding and
      // rows past M (lw = hi = 0): one v_med3 per element
      if (need_mask) {
        const float h = hi_ld[ps], l = lw_ld[ps];
        x0 = __builtin_amdgcn_fmed3f(x0, l, h);
        x1 = __builtin_amdgcn_fmed3f(x1, l, h);
        x2 = __builtin_amdgcn_fmed3f(x2, l, h);
        x3 = __builtin_amdgcn_fmed3f(x3, l, h);
      } else if (pre) {
        // nothing to zero: only the ReLU of the folded BatchNorm (lo = 0) or nothing (lo = -inf)
        x0 = fmaxf(x0, lo);
        x1 = fmaxf(x1, lo);
        x2 = fmaxf(x2, lo);
        x3 = fmaxf(x3, lo);
      }
      float* d = d0 + ps * 64;
      d[0 * LDA] = x0;
      d[1 * LDA] = x1;
      d[2 * LDA] = x2;
      d[3 * LDA] = x3;
    }
  };

  f32x16 acc[MT][NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

  const int ard = lh * LDA + wm * (BM / 2) + li;               // A fragment index inside a stage
  const int brd = 2 * A_ST + lh * BN + wn * (BN / 2) + li;     // B fragment index (stage 0)
  auto compute = [&](int stage) {
    const float* As = lds + stage * A_ST + ard;
    const float* Bs = lds + stage * B_ST + brd;
    float a0[MT], b0[NT], a1[MT], b1[NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) a0[mt] = As[mt * 32];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) b0[nt] = Bs[nt * 32];
#pragma unroll
    for (int j = 0; j < BK / 2; j += 2) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) a1[mt] = As[(2 * j + 2) * LDA + mt * 32];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) b1[nt] = Bs[(2 * j + 2) * BN + nt * 32];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[mt], b0[nt], acc[mt][nt], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (j + 2 < BK / 2) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) a0[mt] = As[(2 * j + 4) * LDA + mt * 32];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) b0[nt] = Bs[(2 * j + 4) * BN + nt * 32];
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[mt], b1[nt], acc[mt][nt], 0, 0, 0);
    }
  };

  snapshot();
  issue(0, kt0);
  landed();
  store(0);
  __syncthreads();
  int kt = kt0;
  for (; kt + 2 <= kt1; kt += 2) {
    // tile kt is in stage 0; tile kt+1 is fetched into stage 1 while stage 0 is consumed
    snapshot();
    issue(1, kt + 1);
    compute(0);
    landed();
    store(1);
    __syncthreads();
    const bool more = kt + 2 < kt1;
    if (more) {
      snapshot();
      issue(0, kt + 2);
    }
    compute(1);
    landed();
    if (more) store(0);
    __syncthreads();
  }
  if (kt < kt1) compute(0);  // odd number of k tiles

  if (partial) {
    // K-slice of a tail tile: dense fp32 slab, summed (with the statistics) by the fixup kernel
    float* slab = g.slabs + ((size_t)(id - g.full_tiles) * g.split + slice) * (BM * BN);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int ml = wm * (BM / 2) + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          slab[ml * BN + wn * (BN / 2) + nt * 32 + li] = acc[mt][nt][r];
        }
    return;
  }

  if (EPI) {
    // ---- inference epilogue: folded BatchNorm (+ residual) (+ ReLU), no statistics ----
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int n = n0 + wn * (BN / 2) + nt * 32 + li;
      const float sc = g.out_scale[n], sh = g.out_shift[n];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = m0 + wm * (BM / 2) + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          if (m < g.M) {
            float v = fmaf(acc[mt][nt][r], sc, sh);
            if (g.res) v += g.res[(long)m * g.Cout + n];
            if (g.relu_out) v = fmaxf(v, 0.f);
            g.y[(long)m * g.Cout + n] = v;
          }
        }
      }
    }
    return;
  }
  // ---- epilogue: raw output + batch-statistics partials ----
  if (!need_mask) {
    // every row of the tile is inside M (MASK is set otherwise): no guards, one scalar base and
    // 32-bit byte offsets (the launcher checks M*Cout*4 < 4 GB) -- one v_add per store instead of
    // a compare, an exec mask and a 64-bit multiply-add
    const unsigned step = (unsigned)g.Cout * 4u;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const unsigned n = (unsigned)(n0 + wn * (BN / 2) + nt * 32 + li);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const unsigned row0 = (unsigned)(m0 + wm * (BM / 2) + mt * 32 + 4 * lh);
        const unsigned base = (row0 * (unsigned)g.Cout + n) * 4u;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const unsigned off = base + (unsigned)((r & 3) + 8 * (r >> 2)) * step;
          gstore32(g.y, off, acc[mt][nt][r]);
        }
      }
    }
  } else {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int n = n0 + wn * (BN / 2) + nt * 32 + li;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = m0 + wm * (BM / 2) + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          if (m < g.M) g.y[(long)m * g.Cout + n] = acc[mt][nt][r];
        }
      }
    }
  }
  if (g.part_sum) {
    using T = TileCfg<BM, BN, BK>;
    block_col_stats<T>(acc, lds, g.part_sum + (long)tm * g.Cout, g.part_sq + (long)tm * g.Cout,
                       n0, g.Cout);
  }
}

// One 1024-thread workgroup per tail tile: y = sum of its K-slice slabs (fixed order), plus that
// tile's column sums / sums of squares for the batch statistics. The slab loads of one output
// element are independent and issued together (the kernel is pure latency otherwise).
template <int BM, int BN, bool EPI>
__global__ __launch_bounds__(1024) void conv_tail_fixup_kernel(ConvArgs2 g) {
  constexpr int RL = 1024 / BN;  // row lanes
  __shared__ float s_sum[RL][BN], s_sq[RL][BN];
  const int id = g.full_tiles + blockIdx.x;
  const int tm = id / g.tiles_n, tn = id - tm * g.tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;
  const int c = threadIdx.x % BN, rl = threadIdx.x / BN;
  const float* slab = g.slabs + (size_t)blockIdx.x * g.split * (BM * BN);
  const int sp = g.split;
  float cs = 0.f, cq = 0.f;
  for (int ml = rl; ml < BM; ml += RL) {
    float p[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) p[k] = slab[(size_t)(k < sp ? k : sp - 1) * (BM * BN) + ml * BN + c];
    float v = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) v += k < sp ? p[k] : 0.f;
    const int m = m0 + ml;
    if (m < g.M) {
      if (EPI) {
        v = fmaf(v, g.out_scale[n0 + c], g.out_shift[n0 + c]);
        if (g.res) v += g.res[(long)m * g.Cout + n0 + c];
        if (g.relu_out) v = fmaxf(v, 0.f);
      }
      g.y[(long)m * g.Cout + n0 + c] = v;
      cs += v;
      cq = fmaf(v, v, cq);
    }
  }
  if (!EPI && g.part_sum) {
    s_sum[rl][c] = cs;
    s_sq[rl][c] = cq;
    __syncthreads();
    if (rl == 0) {
      for (int r = 1; r < RL; ++r) { cs += s_sum[r][c]; cq += s_sq[r][c]; }
      g.part_sum[(long)tm * g.Cout + n0 + c] = cs;
      g.part_sq[(long)tm * g.Cout + n0 + c] = cq;
    }
  }
}

// Tail plan for T tiles (BM x BN, nk k-tiles each) on 256 CUs, in units of one tile's time.
// Unsplit, the tiles past the last full round of 256 cost a whole extra tile time; cut into
// `split` K-slices they cost ceil(R*split/256)/split of it, plus the slab round trip through
// HBM (priced at 4 TB/s against ~0.45 TFLOP/s per CU). Returns the estimated time; *split = 1
// means "leave it alone".
static double plan_tail(int T, int nk, int BM, int BN, int* full_tiles, int* split, int* kps) {
  const int full = (T / 256) * 256;
  const int R = T - full;
  *full_tiles = T;
  *split = 1;
  *kps = nk;
  const double unsplit = (double)(full / 256) + (R > 0 ? 1.0 : 0.0);
  // Below two full rounds the CUs are latency-bound, not MFMA-bound (one 64x64 tile alone on a
  // CU takes 66 us, two take 91 us): an extra tile is cheap there and slicing it does not pay.
  if (R == 0 || full < 512) return unsplit;
  const double tile_s = 2.0 * BM * BN * nk * 16 / 0.45e12;
  double best = unsplit;
  int best_slices = 1, best_per = nk;
  for (int sp = 2; sp <= 16; ++sp) {
    const int per = cdiv(nk, sp);
    if (per < 4) break;  // keep >= 4 k-tiles per slice
    const int slices = cdiv(nk, per);
    const double slab_s = 2.0 * R * slices * BM * BN * 4 / 4e12 + 4e-6;  // + the fix-up launch
    const double t = (double)(full / 256) + (double)cdiv((long)R * slices, 256) * per / nk +
                     slab_s / tile_s;
    if (t < best) {
      best = t;
      best_slices = slices;
      best_per = per;
    }
  }
  if (best_slices > 1 && best < 0.96 * unsplit) {
    *full_tiles = full;
    *split = best_slices;
    *kps = best_per;
    return best;
  }
  return unsplit;
}

static void tile_dims(int tile, int* BM, int* BN) {
  *BM = tile == 64 ? 64 : 128;
  *BN = tile == 128 ? 128 : 64;
}

// Tile choice for the K-major kernel. Measured model (tools/probes/conv_bench.py, B sweep): with n
// tiles per CU a k-tile step costs max(n * c / 0.8, c / 0.47) cycles, c = MFMA cycles of one
// tile's k-tile per SIMD (512 / 1024 / 2048 for 64x64 / 128x64 / 128x128): a lone workgroup per
// CU is latency-bound at ~47 % of the matrix pipe, several share it at ~80 %. n comes from the
// tail plan (fractional once the tail is K-sliced).
int conv_v2_auto_tile(int M, int Cout, int Kw) {
  const int cands[3] = {128, 12864, 64};
  int best = 64;
  double best_t = 1e30;
  for (int tile : cands) {
    int BM, BN;
    tile_dims(tile, &BM, &BN);
    if (Cout % BN != 0) continue;
    const int T = cdiv(M, BM) * (Cout / BN);
    int full, sp, kps;
    const double n = plan_tail(T, Kw / 16, BM, BN, &full, &sp, &kps);
    const double c = (double)BM * BN / 8.0;
    double t = std::max(n * c / 0.8, c / 0.47);
    if (tile != 128) t *= 1.02;  // smaller tiles re-read more operand bytes through L2
    if (t < best_t) { best_t = t; best = tile; }
  }
  return best;
}

// A requested tile whose N extent does not divide Cout falls back to the 64-wide one. Every
// planning helper and the launcher go through this, so workspaces are sized for the tile that runs.
static int norm_tile(int M, int Cout, int Kw, int tile) {
  if (tile == 0) return conv_v2_auto_tile(M, Cout, Kw);
  if (tile == 128 && Cout % 128 != 0) return 12864;
  return tile;
}

// out = {tile, tiles, full_tiles, split, k-tiles per slice}
void conv_v2_plan(int M, int Cout, int Kw, int tile, int* out) {
  tile = norm_tile(M, Cout, Kw, tile);
  int BM, BN;
  tile_dims(tile, &BM, &BN);
  const int T = cdiv(M, BM) * (Cout / BN);
  int full, sp, kps;
  plan_tail(T, Kw / 16, BM, BN, &full, &sp, &kps);
  out[0] = tile; out[1] = T; out[2] = full; out[3] = sp; out[4] = kps;
}

size_t conv_v2_slab_floats(int M, int Cout, int Kw, int tile) {
  tile = norm_tile(M, Cout, Kw, tile);
  int BM, BN;
  tile_dims(tile, &BM, &BN);
  if (Cout % BN != 0) return 0;
  const int T = cdiv(M, BM) * (Cout / BN);
  int full, sp, kps;
  plan_tail(T, Kw / 16, BM, BN, &full, &sp, &kps);
  return sp > 1 ? (size_t)(T - full) * sp * BM * BN : 0;
}

template <int BM, int BN>
static void launch_v2(ConvArgs2& g, hipStream_t stream) {
  g.tiles_m = cdiv(g.M, BM);
  g.tiles_n = g.Cout / BN;
  const int T = g.tiles_m * g.tiles_n;
  magic_div((unsigned)g.tiles_n, &g.tn_mul, &g.tn_sh);
  magic_div((unsigned)g.Cin, &g.cin_mul, &g.cin_sh);
  magic_div((unsigned)g.KW, &g.kw_mul, &g.kw_sh);
  g.full_tiles = T;
  g.split = 1;
  g.kps = g.Kw / 16;
  if (g.slabs) {
    int full, sp, kps;
    plan_tail(T, g.Kw / 16, BM, BN, &full, &sp, &kps);
    if (sp > 1) { g.full_tiles = full; g.split = sp; g.kps = kps; }
  }
  magic_div((unsigned)g.split, &g.split_mul, &g.split_sh);
  const int rem = T - g.full_tiles;
  // staging mode of the A tile (see the kernel's MODE): PRE iff a BatchNorm is folded into the
  // load, MASK iff something must be zeroed (padding, ragged last M tile)
  // (MASK also when the output does not fit 32-bit byte offsets: the unguarded store path uses them)
  const int mode = (g.in_scale ? 1 : 0) |
                   ((g.pad > 0 || (g.M % BM) != 0 || (long)g.M * g.Cout * 4 >= (1L << 32)) ? 2 : 0);
  const dim3 grid(g.full_tiles + rem * g.split), block(kGemmThreads);
#define CAPNET_CONV_LAUNCH(E, MD) \
  hipLaunchKernelGGL((conv_f32_v2_kernel<BM, BN, E, MD>), grid, block, 0, stream, g)
  if (g.out_scale) {
    switch (mode) {
      case 0: CAPNET_CONV_LAUNCH(true, 0); break;
      case 1: CAPNET_CONV_LAUNCH(true, 1); break;
      case 2: CAPNET_CONV_LAUNCH(true, 2); break;
      default: CAPNET_CONV_LAUNCH(true, 3); break;
    }
    if (rem > 0)
      hipLaunchKernelGGL((conv_tail_fixup_kernel<BM, BN, true>), dim3(rem), dim3(1024), 0, stream, g);
    return;
  }
  switch (mode) {
    case 0: CAPNET_CONV_LAUNCH(false, 0); break;
    case 1: CAPNET_CONV_LAUNCH(false, 1); break;
    case 2: CAPNET_CONV_LAUNCH(false, 2); break;
    default: CAPNET_CONV_LAUNCH(false, 3); break;
  }
#undef CAPNET_CONV_LAUNCH
  if (rem > 0)
    hipLaunchKernelGGL((conv_tail_fixup_kernel<BM, BN, false>), dim3(rem), dim3(1024), 0, stream, g);
}

bool conv_v2_eligible(const float* x, long sxb, long sxh, long sxw, long sxc, int Bn, int Cin,
                      int Cout, const float* in_scale, const float* in_shift) {
  return (Cin % 16 == 0) && sxc == 1 && (sxw % 4 == 0) && (sxh % 4 == 0) && (sxb % 4 == 0) &&
         aligned16(x) && ((long)Bn * sxb * 4 < (1L << 31)) && (Cout % 64 == 0) &&
         (!in_scale || (aligned16(in_scale) && aligned16(in_shift)));
}

// wk: weights packed [Kw][Cout] (pack_conv_weight_kmajor). tile: 0 auto, 128, 64, 12864.
int conv2d_fwd_v2(const float* x, long sxb, long sxh, long sxw, const float* wk, int Kw, float* y,
                  const float* in_scale, const float* in_shift, int relu_in, float* part_sum,
                  float* part_sq, int Bn, int H, int W, int Cin, int Cout, int KH, int KW,
                  int stride, int pad, int tile, float* slabs, hipStream_t stream,
                  const float* out_scale, const float* out_shift, const float* res, int relu_out) {
  CAPNET_REQUIRE(x && wk && y, "conv2d_fwd_v2: null pointer");
  CAPNET_REQUIRE((out_scale == nullptr) == (out_shift == nullptr), "conv2d_fwd_v2: epilogue scale/shift pair");
  CAPNET_REQUIRE(!out_scale || !part_sum, "conv2d_fwd_v2: the folded-BN epilogue produces no statistics");
  CAPNET_REQUIRE(out_scale || (!res && !relu_out), "conv2d_fwd_v2: residual / ReLU need the epilogue");
  CAPNET_REQUIRE(conv_v2_eligible(x, sxb, sxh, sxw, 1, Bn, Cin, Cout, in_scale, in_shift),
                 "conv2d_fwd_v2: shape/alignment not supported (Cin=%d Cout=%d)", Cin, Cout);
  CAPNET_REQUIRE(Kw == KH * KW * Cin && aligned16(wk), "conv2d_fwd_v2: packed weight stride");
  CAPNET_REQUIRE((in_scale == nullptr) == (in_shift == nullptr), "conv2d_fwd_v2: scale/shift pair");
  CAPNET_REQUIRE((part_sum == nullptr) == (part_sq == nullptr), "conv2d_fwd_v2: stats pair");
  CAPNET_REQUIRE(in_scale || !relu_in, "conv2d_fwd_v2: relu_in needs a scale/shift prologue");
  ConvArgs2 g;
  g.x = x; g.wk = wk; g.y = y;
  g.in_scale = in_scale; g.in_shift = in_shift;
  g.part_sum = part_sum; g.part_sq = part_sq;
  g.out_scale = out_scale; g.out_shift = out_shift; g.res = res; g.relu_out = relu_out;
  g.H = H; g.W = W; g.Cin = Cin; g.Cout = Cout; g.KW = KW; g.stride = stride; g.pad = pad;
  g.OH = (H + 2 * pad - KH) / stride + 1;
  g.OW = (W + 2 * pad - KW) / stride + 1;
  g.sxb = (int)sxb; g.sxh = (int)sxh; g.sxw = (int)sxw;
  const long M = (long)Bn * g.OH * g.OW;
  CAPNET_REQUIRE(M < (1L << 24), "conv2d_fwd_v2: too many output pixels (%ld >= 2^24)", M);
  g.M = (int)M;
  magic_div((unsigned)(g.OH * g.OW), &g.ohw_mul, &g.ohw_sh);
  magic_div((unsigned)g.OW, &g.ow_mul, &g.ow_sh);
  g.Kw = Kw;
  g.relu_in = relu_in;
  g.slabs = slabs;
  g.x_bytes = (unsigned)((long)Bn * sxb * 4);
  g.ss_bytes = (unsigned)(Cin * 4);
  tile = norm_tile(g.M, Cout, Kw, tile);
  if (tile == 128) launch_v2<128, 128>(g, stream);
  else if (tile == 64) launch_v2<64, 64>(g, stream);
  else if (tile == 12864) launch_v2<128, 64>(g, stream);
  else CAPNET_REQUIRE(false, "conv2d_fwd_v2: unknown tile %d", tile);
  CAPNET_LAUNCH_CHECK();
  return kOk;
}

}  // namespace capnet
