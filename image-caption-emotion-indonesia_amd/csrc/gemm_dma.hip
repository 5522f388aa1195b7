// C[M][N] = A'[M][K] . B[N][K]^T on v_mfma_f32_32x32x2_f32, both operands K-contiguous, staged by
// LDS-DMA with NO vector-ALU instruction in the k-loop.
//
// Used for (i) the 1x1 convolutions of the ResNet-152 trunk whose input is already activated --
// conv1 and the downsample branch of every bottleneck (torchvision Bottleneck, call sites
// stylenet/model.py:15-18,24): an NHWC activation tensor is A[M = B*OH*OW][Cin] (row m = one output
// pixel, also for stride 2) and the OIHW weight of a 1x1 conv IS B[Cout][Cin], no packing; the
// epilogue writes the raw output and the per-tile column sums / sums of squares of the train-mode
// BatchNorm that follows -- and (ii) the vocabulary projection logits = hiddens . C^T + bias
// (stylenet/model.py:193-194), whose operands have the same shape.
//
// Why a second GEMM core. On gfx950 the f32-input MFMA executes on the vector datapath: every VALU
// instruction a wave issues is time the matrix pipe does not get (tools/probes/native/coexec.hip).
// The K-major conv kernel (conv_f32_v2.hip) stages A through registers (8 ds_write_b32 + address
// VALU per k-tile) and reads fragments with one ds_read_b32 per MFMA operand. Here:
//   * A and B tiles (128 x 32 and BN x 32 floats per k-tile) go global -> LDS by
//     global_load_lds_dwordx4, 16 B per lane, 8 instructions per wave and k-tile; the per-lane
//     source offset is a loop constant (row stride x row + swizzled k quad), the k advance is a
//     scalar add on the base: zero VALU.
//   * LDS image per operand: 16-B cells holding 4 consecutive k of one row, ordered
//     [row >> 4][k quad >> 2][row & 15][pos], pos = (k quad & 3) ^ ((row >> 2) & 3). One DMA
//     instruction fills 16 rows x 64 B (16 cache lines touched instead of 64), lane-linear in LDS as
//     the DMA requires; the XOR lives in the SOURCE address, so that a wave's ds_read_b128 of one k
//     quad for 32 consecutive rows hits 16 distinct 16-B bank slots per 16-lane group.
//   * fragments: ds_read_b128 = 4 MFMA k-steps of one operand; 16 reads feed 64 MFMAs per wave and
//     k-tile. The k <-> (MFMA step, lane half) map is k = 4 * (2 jj + half) + e for step 4 jj + e,
//     the same for both operands, so the sum over k is complete whatever the order.
//   * 128 x 128 (or 128 x 64) tile, 4 waves of 64 x 64 (64 x 32), BK = 32, two LDS stages, one
//     barrier per k-tile, two workgroups per CU.
#include <cstdlib>

#include "common.h"
#include "mfma_core.h"
#include "kernels.h"

namespace capnet {

namespace {

constexpr int DBM = 128;

struct NtArgs {
  const float* A;
  const float* B;
  float* C;
  const float* bias;
  float* part_sum;
  float* part_sq;
  // inference epilogue (out_scale != null): y = act(acc * out_scale[n] + out_shift[n] + res)
  const float* out_scale;
  const float* out_shift;
  const float* res;
  int relu_out;
  int M, N, K;
  int tiles_m, tiles_n;
  unsigned tn_mul, tn_sh;
  // row m of A starts at float offset: conv ? b*sxb + oh*stride*sxh + ow*stride*sxw : m*lda
  int conv, lda, OW, OHW, stride, sxb, sxh, sxw;
  unsigned ohw_mul, ohw_sh, ow_mul, ow_sh;
};

__device__ __forceinline__ unsigned row_offset_floats(const NtArgs& g, int m) {
  if (!g.conv) return (unsigned)m * (unsigned)g.lda;
  const int b = (int)fast_div((unsigned)m, g.ohw_mul, g.ohw_sh);
  const int rem = m - b * g.OHW;
  const int oh = (int)fast_div((unsigned)rem, g.ow_mul, g.ow_sh);
  const int ow = rem - oh * g.OW;
  return (unsigned)(b * g.sxb + oh * g.stride * g.sxh + ow * g.stride * g.sxw);
}

// LDS-DMA of 16 B per lane: LDS[lds_base + IMM + GOFF + 16 * lane] = *(sbase + voff + GOFF bytes): the
// instruction's offset field moves BOTH addresses (callers subtract GOFF from IMM). The LDS
// address and the global offset are formed from ONE scalar base each plus immediates: as separate
// precomputed SGPR values (16 destinations, 4 source pairs) the kernel ran out of scalar registers
// and hipcc spilled them to VGPR lanes -- v_readlane / v_writelane in the k-loop, i.e. VALU
// instructions in front of the f32 MFMAs (measured: 1.8 VALU per MFMA instead of ~0.5).
template <int IMM, int GOFF>
__device__ __forceinline__ void glds16_imm(const float* sbase, unsigned voff_bytes, unsigned lds_base) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_add_u32 m0, %2, %4\n\t"
      "s_nop 2\n\t"               // 5 wait states in all: M0 (1) and an SGPR base fresh from a VALU write (5), mfma_core.h
      "global_load_lds_dwordx4 %1, %3 offset:%5\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(voff_bytes), "s"(lds_base), "s"(sbase), "i"(IMM), "i"(GOFF)
      : "memory", "scc");
}

// byte address of cell (row, kq) inside one operand image; HALVES = BK / 16 groups of 4 k quads
template <int HALVES>
__device__ __forceinline__ unsigned cell_addr(int row, int kq) {
  const int rg = row >> 4, r = row & 15;
  return (unsigned)((((rg * HALVES + (kq >> 2)) * 16 + r) * 4 + ((kq & 3) ^ ((r >> 2) & 3))) * 16);
}

// Persistent: the grid is min(tiles, 2 per CU) workgroups; a workgroup walks tiles id, id + grid, ...
// and requests the first k-tile of its NEXT tile before it stores the current one, so that a tile's
// DMA start-up latency and its 64-KB store / statistics overlap (a trunk 1x1 conv has 2-32 k-tiles
// per output tile: without this the start-up and the epilogue are 40-50 % of a tile's time).
// BK = 32: 64 KB of LDS, two workgroups per CU (long K, many tiles: the vocabulary projection).
// BK = 16: 32 KB, four per CU -- with 2-32 k-tiles per output tile and about as many tiles as CUs a
// trunk 1x1 conv needs the extra waves per SIMD to keep the matrix pipe fed across the barriers.
template <int BN, int DBK>
__global__ __launch_bounds__(256, DBK == 32 ? 2 : 4) void nt_dma_kernel(const NtArgs g) {
  constexpr int NT = BN / 64;                          // 32-column MFMA tiles per wave
  constexpr int HALVES = DBK / 16;
  constexpr int kCellsPerRow = DBK / 4;
  constexpr int kImgA = DBM * kCellsPerRow * 16;       // bytes of the A image of one stage
  constexpr int kImgB = BN * kCellsPerRow * 16;
  constexpr int kRG = HALVES * 1024;                   // bytes of one 16-row group
  constexpr int kStage = kImgA + kImgB;
  constexpr int RGA = DBM / 16 / 4;                    // A row groups per wave (2)
  constexpr int RGB = BN / 16 / 4;                     // B row groups per wave (2 or 1)
  __shared__ __attribute__((aligned(16))) float lds[2 * kStage / 4];
  __shared__ float s_stats[4 * BN];                    // column sums: apart from the DMA stages
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int li = lane & 31, lh = lane >> 5;
  const int ntiles = g.tiles_m * g.tiles_n;
  const int nk = g.K / DBK;

  // ---- DMA geometry: lane (r = lane >> 2, pos = lane & 3) of instruction (row group, k half)
  const int dr = lane >> 2, dpos = lane & 3;
  const int dq = dpos ^ ((dr >> 2) & 3);               // k quad (within the half) this lane fetches
  unsigned voffA[RGA], voffB[RGB];
  const unsigned lds0 = __builtin_amdgcn_readfirstlane(
      (unsigned)(size_t)(__attribute__((address_space(3))) float*)lds);
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const float* sA = g.A;
  const float* sB = g.B;
  int m0 = 0, n0 = 0, tm = 0;
  // tile -> logical id: the tiles one XCD works on (tile % 8 = blockIdx % 8: the dispatcher deals
  // workgroups round-robin over the XCDs) are consecutive ids, and consecutive ids sweep the N
  // tiles of one row block, so the rows of A a XCD reads stay in ITS L2
  auto setup = [&](int tile) {
    const int id = xcd_remap(tile, ntiles);
    tm = (int)fast_div((unsigned)id, g.tn_mul, g.tn_sh);
    const int tn = id - tm * g.tiles_n;
    m0 = tm * DBM;
    n0 = tn * BN;
#pragma unroll
    for (int q = 0; q < RGA; ++q) {
      const int m = m0 + (wave * RGA + q) * 16 + dr;
      voffA[q] = (row_offset_floats(g, m < g.M ? m : g.M - 1) + 4u * dq) * 4u;
    }
#pragma unroll
    for (int q = 0; q < RGB; ++q) {
      const int n = n0 + (wave * RGB + q) * 16 + dr;
      voffB[q] = ((unsigned)n * (unsigned)g.K + 4u * dq) * 4u;
    }
    sA = g.A;
    sB = g.B;
  };
  // (stage is a literal at every call site: all LDS offsets below fold into immediates)
  const unsigned ldsA = lds0 + (unsigned)(wave_u * RGA * kRG);
  const unsigned ldsB = lds0 + (unsigned)kImgA + (unsigned)(wave_u * RGB * kRG);
#define CAPNET_ISSUE(STAGE)                                                                  \
  do {                                                                                       \
    glds16_imm<(STAGE) * kStage, 0>(sA, voffA[0], ldsA);                                     \
    if (HALVES == 2) glds16_imm<(STAGE) * kStage + 1024 - 64, 64>(sA, voffA[0], ldsA);       \
    glds16_imm<(STAGE) * kStage + kRG, 0>(sA, voffA[RGA - 1], ldsA);                         \
    if (HALVES == 2) glds16_imm<(STAGE) * kStage + kRG + 1024 - 64, 64>(sA, voffA[RGA - 1], ldsA); \
    glds16_imm<(STAGE) * kStage, 0>(sB, voffB[0], ldsB);                                     \
    if (HALVES == 2) glds16_imm<(STAGE) * kStage + 1024 - 64, 64>(sB, voffB[0], ldsB);       \
    if (RGB == 2) {                                                                          \
      glds16_imm<(STAGE) * kStage + kRG, 0>(sB, voffB[RGB - 1], ldsB);                       \
      if (HALVES == 2) glds16_imm<(STAGE) * kStage + kRG + 1024 - 64, 64>(sB, voffB[RGB - 1], ldsB); \
    }                                                                                        \
    sA += DBK;                                                                               \
    sB += DBK;                                                                               \
  } while (0)

  // ---- fragment geometry: lane (li, lh); quad 2 jj + lh of its row; the two (jj & 1) variants
  const char* ldsc = reinterpret_cast<const char*>(lds);
  const char* a_rd[2];
  const char* b_rd[2];
#pragma unroll
  for (int v = 0; v < 2; ++v) {
    a_rd[v] = ldsc + cell_addr<HALVES>(wm * 64 + li, 2 * v + lh);
    b_rd[v] = ldsc + kImgA + cell_addr<HALVES>(wn * (BN / 2) + li, 2 * v + lh);
  }

  f32x16 acc[2][NT];
  auto zero_acc = [&]() {
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;
  };

  auto frag = [&](const char* const (&rd)[2], int stage, int t32, int jj) -> f32x4 {
    // rows + 32 t32: two row groups further; quads 4..7: the second k half (1024 B)
    return *reinterpret_cast<const f32x4*>(rd[jj & 1] + stage * kStage + t32 * 2 * kRG + (jj >> 1) * 1024);
  };
  auto compute = [&](int stage) {
    f32x4 af[2][2], bf[2][NT];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) af[0][mt] = frag(a_rd, stage, mt, 0);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) bf[0][nt] = frag(b_rd, stage, nt, 0);
#pragma unroll
    for (int jj = 0; jj < 2 * HALVES; ++jj) {
      const int cur = jj & 1, nxt = cur ^ 1;
      if (jj + 1 < 2 * HALVES) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) af[nxt][mt] = frag(a_rd, stage, mt, jj + 1);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) bf[nxt][nt] = frag(b_rd, stage, nt, jj + 1);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cur][mt][e], bf[cur][nt][e], acc[mt][nt], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  auto landed = [&]() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  };

  int tile = blockIdx.x;
  setup(tile);
  CAPNET_ISSUE(0);
  while (true) {
    zero_acc();
    landed();
    int kt = 0;
    for (; kt + 2 <= nk; kt += 2) {
      CAPNET_ISSUE(1);
      compute(0);
      landed();
      if (kt + 2 < nk) CAPNET_ISSUE(0);
      compute(1);
      landed();
    }
    if (kt < nk) {   // odd number of k-tiles: the last one sits in stage 0
      compute(0);
      __syncthreads();
    }
    // ---- the next tile's first k-tile goes out before this tile is stored
    const int em0 = m0, en0 = n0, etm = tm;
    const int next = tile + (int)gridDim.x;
    const bool more = next < ntiles;
    if (more) {
      setup(next);
      CAPNET_ISSUE(0);
    }

    // ---- epilogue: D layout of the 32x32 tile: column = lane & 31, rows (r & 3) + 8 (r >> 2) + 4 lh
    // (offsets advance by one row step in a VGPR: as 16 precomputed multiples of the row step the
    // scalar registers of the k-loop overflowed into VGPR lanes)
    const bool ragged = em0 + DBM > g.M;
    const unsigned rstep = (unsigned)g.N * 4u;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int n = en0 + wn * (BN / 2) + nt * 32 + li;
      const float bv = g.bias ? g.bias[n] : 0.f;
      const float osc = g.out_scale ? g.out_scale[n] : 1.f, osh = g.out_scale ? g.out_shift[n] : bv;
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        int row = em0 + wm * 64 + mt * 32 + 4 * lh;
        unsigned off = ((unsigned)row * (unsigned)g.N + (unsigned)n) * 4u;   // (M*N*4 < 2^32: eligibility)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          if (!ragged || row < g.M) {
            float v = acc[mt][nt][r];
            if (g.out_scale) {
              v = fmaf(v, osc, osh);
              if (g.res) v += *reinterpret_cast<const float*>(reinterpret_cast<const char*>(g.res) + off);
              if (g.relu_out) v = fmaxf(v, 0.f);
            } else {
              v += bv;
            }
            gstore32(g.C, off, v);
          } else {
            acc[mt][nt][r] = 0.f;      // rows past M stay out of the statistics below
          }
          // next row of the D layout: +1, +1, +1, +5 (rows (r & 3) + 8 (r >> 2))
          if ((r & 3) == 3) { row += 5; off += 5u * rstep; } else { row += 1; off += rstep; }
        }
      }
    }
    if (g.part_sum) {
      using T = TileCfg<DBM, BN, 16>;
      block_col_stats<T>(acc, s_stats, g.part_sum + (long)etm * g.N, g.part_sq + (long)etm * g.N, en0, g.N);
    }
    if (!more) break;
    tile = next;
  }
}

int launch_nt(const NtArgs& a, int BN, int bk, hipStream_t stream) {
  static int cus = 0;
  if (cus == 0) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) {
      (void)hipGetLastError();
      cus = 256;
    }
  }
  const int tiles = a.tiles_m * a.tiles_n;
  const int per_cu = bk == 32 ? 2 : 4;                     // resident workgroups per CU
  const int grid = tiles < per_cu * cus ? tiles : per_cu * cus;
  if (bk == 32) {
    if (BN == 128) hipLaunchKernelGGL((nt_dma_kernel<128, 32>), dim3(grid), dim3(256), 0, stream, a);
    else hipLaunchKernelGGL((nt_dma_kernel<64, 32>), dim3(grid), dim3(256), 0, stream, a);
  } else {
    if (BN == 128) hipLaunchKernelGGL((nt_dma_kernel<128, 16>), dim3(grid), dim3(256), 0, stream, a);
    else hipLaunchKernelGGL((nt_dma_kernel<64, 16>), dim3(grid), dim3(256), 0, stream, a);
  }
  CAPNET_LAUNCH_CHECK();
  return kOk;
}

int pick_bn(int N) {
  return N % 128 == 0 ? 128 : 64;
}

int pick_bk(int K, int tiles) {
  return (K % 32 == 0 && K >= 512 && tiles >= 512) ? 32 : 16;
}

}  // namespace

bool sgemm_nt_dma_eligible(int M, int N, int K, const float* A, long lda, const float* B, long ldb,
                           const float* C, long ldc) {
  return M >= 1 && N % 64 == 0 && K % 16 == 0 && ldb == K && ldc == N && lda % 4 == 0 &&
         aligned16(A) && aligned16(B) && (long)M * lda * 4 < (1l << 32) && (long)N * K * 4 < (1l << 32) &&
         (long)M * N * 4 < (1l << 32);
}

// C[M][N] = A[M][K] . B[N][K]^T + bias[N]
int sgemm_nt_dma(int M, int N, int K, const float* A, long lda, const float* B, float* C,
                 const float* bias, hipStream_t stream) {
  CAPNET_REQUIRE(sgemm_nt_dma_eligible(M, N, K, A, lda, B, K, C, N), "sgemm_nt_dma: operands not eligible");
  NtArgs a{};
  a.A = A; a.B = B; a.C = C; a.bias = bias;
  a.M = M; a.N = N; a.K = K;
  // (128-column tiles once they alone fill the chip: encoder_att over 12 x 196 pixels, 2352 x 512, was 76 workgroups)
  const int BN = (pick_bn(N) == 128 && (long)cdiv(M, DBM) * (N / 128) >= 200) ? 128 : 64;
  a.tiles_m = cdiv(M, DBM); a.tiles_n = N / BN;
  magic_div((unsigned)a.tiles_n, &a.tn_mul, &a.tn_sh);
  a.conv = 0; a.lda = (int)lda;
  return launch_nt(a, BN, pick_bk(K, a.tiles_m * a.tiles_n), stream);
}

bool conv1x1_dma_eligible(const float* x, long sxb, long sxh, long sxw, long sxc, int Bn, int H, int W,
                          int Cin, int Cout, int stride) {
  const long OH = (H - 1) / stride + 1, OW = (W - 1) / stride + 1;
  return sxc == 1 && Cin % 16 == 0 && Cout % 64 == 0 && aligned16(x) && sxb % 4 == 0 && sxh % 4 == 0 &&
         sxw % 4 == 0 && (long)Bn * sxb * 4 < (1l << 32) && (long)Bn * OH * OW < (1l << 24) &&
         (long)Bn * OH * OW * Cout * 4 < (1l << 32);
}

int conv1x1_tiles_m(long M) { return cdiv(M, DBM); }

// y[B*OH*OW][Cout] = x(pixels at stride)[..][Cin] . w[Cout][Cin]^T; part_sum / part_sq
// [conv1x1_tiles_m(M)][Cout]: column sums and sums of squares of each 128-row tile (or null)
int conv1x1_fwd_dma(const float* x, long sxb, long sxh, long sxw, const float* w_oi, float* y,
                    float* part_sum, float* part_sq, int Bn, int H, int W, int Cin, int Cout, int stride,
                    hipStream_t stream, const float* out_scale, const float* out_shift, const float* res,
                    int relu_out) {
  CAPNET_REQUIRE(x && w_oi && y && stride >= 1, "conv1x1_fwd_dma: bad argument");
  CAPNET_REQUIRE(conv1x1_dma_eligible(x, sxb, sxh, sxw, 1, Bn, H, W, Cin, Cout, stride) && aligned16(w_oi),
                 "conv1x1_fwd_dma: operands not eligible");
  NtArgs a{};
  const int OH = (H - 1) / stride + 1, OW = (W - 1) / stride + 1;
  a.A = x; a.B = w_oi; a.C = y; a.bias = nullptr; a.part_sum = part_sum; a.part_sq = part_sq;
  a.out_scale = out_scale; a.out_shift = out_shift; a.res = res; a.relu_out = relu_out;
  CAPNET_REQUIRE(!out_scale || (out_shift && !part_sum), "conv1x1_fwd_dma: folded epilogue takes no statistics");
  a.M = Bn * OH * OW; a.N = Cout; a.K = Cin;
  const int BN = pick_bn(Cout);
  a.tiles_m = cdiv(a.M, DBM); a.tiles_n = Cout / BN;
  magic_div((unsigned)a.tiles_n, &a.tn_mul, &a.tn_sh);
  a.conv = 1; a.OW = OW; a.OHW = OH * OW; a.stride = stride;
  a.sxb = (int)sxb; a.sxh = (int)sxh; a.sxw = (int)sxw;
  magic_div((unsigned)(OH * OW), &a.ohw_mul, &a.ohw_sh);
  magic_div((unsigned)OW, &a.ow_mul, &a.ow_sh);
  return launch_nt(a, BN, pick_bk(Cin, a.tiles_m * a.tiles_n), stream);
}

}  // namespace capnet
