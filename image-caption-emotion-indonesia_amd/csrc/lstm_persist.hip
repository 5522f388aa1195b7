// Persistent LSTM sequence kernel: ONE launch runs a whole run of consecutive teacher-forced time
// steps of the recurrence (stylenet/model.py:180-191 around forward_step :147-153; nn.LSTMCell of
// nic/model.py:77), with the recurrent weights resident in registers for the launch.
//
//   per step t:  G_t += h_{t-1} . Wcat^T ;  i,f,o = sigmoid, g = tanh ;  c = f c + i g ;
//                h = o c  (FactoredLSTM, model.py:153)  |  h = o tanh(c)  (LSTMCell)
//
// Decomposition (H = 512, b <= 128; 256 workgroups x 256 threads = one per CU):
//   * workgroup id -> (shard = id % 8, slot = id / 8). A shard owns the batch rows g = 8 m + shard
//     (m < 16) and ALL of Wcat; the dispatcher deals workgroups round-robin over the 8 XCDs, so a
//     shard normally sits on ONE XCD and its 32 workgroups exchange h through that XCD's L2.
//   * slot s owns hidden units 16 s .. 16 s + 15 = 64 gate columns, over the whole K = 512: 128 KB
//     of weights (two f16 pieces per weight, as many bytes as fp32) = 128 VGPRs per lane, loaded once
//     per launch. Wave w holds k in [128 w, 128 w + 128): split-K over the 4 waves, summed through LDS.
//   * the product is split-f16 on v_mfma_f32_16x16x32_f16 (three products per multiply, fp32
//     accumulation: the trunk's arithmetic, see above persist_mfma): 16 rows of the shard x 64 columns
//     x 128 k per wave = 48 MFMAs of 16 cycles per step. (Round 2 used v_mfma_f32_4x4x1_16B_f32, the
//     f32 pipe without padding rows: 256 instructions of 8..10 cycles, 45 % of the step.) h_{t-1}
//     goes from global memory into registers, is split there (5 VALU per pair) and fed to the MFMAs:
//     no LDS staging.
//   * exchange of h between the steps: the rows of h_t are the rows of the `hiddens` output
//     itself. A wave stores its part, waits for the stores (vmcnt), then stores tick t + 1 into its
//     flag; a consuming wave polls the 32 flags of the 8 slots that produce its k range, then loads
//     h with L1-bypassing (sc1) loads. Nothing is ever overwritten (every step has its own rows,
//     flags only grow), so there is no write-after-read hazard and no reset.
//   * c never leaves registers; pre-activations of step t are requested before the poll.
//
// Cross-workgroup visibility (MI355X_MICROARCH.md, "Workgroup dispatch, XCD placement &
// inter-workgroup visibility"): correctness never assumes a placement. At start every workgroup
// publishes its XCC id (HW_REG_XCC_ID) in a table, with a write-through store, and reads its
// shard's 32 entries. If they are all equal the shard is in LOCAL mode: plain stores (they stay in
// that XCD's L2, which all 32 workgroups share) + sc1 loads. Otherwise it is in SAFE mode: every
// handed-off byte and flag is stored write-through (agent scope, sc1) and loaded with sc1 loads,
// the form the guide measures as valid across XCDs. Both modes run the same code; only the cache
// policy of the h / flag stores differs.
// Every spin is bounded: a workgroup that waits too long (co-residency lost, a peer died) raises the
// abort word, sets bit 2 of err_flag and every workgroup leaves. The hiddens of that segment are then
// garbage and so is everything computed from them; what keeps the TRAINING STATE intact is that the
// optimizer kernel reads the same flag and leaves parameters and moments alone while it is set
// (loss_optim.hip, clamp_adam). The host sees the flag at its next capnet.ops.check_device_errors(),
// which switches the process to the launch-per-step path (lstm_persist_set_mode) and reports the
// dropped steps. Both rare paths -- SAFE mode and the abort -- can be forced for tests
// (lstm_persist_set_mode bits 1 and 2; tests/test_lstm_persist_gpu.py).
#include <cstdlib>

#include "common.h"
#include "kernels.h"
#include "mfma_core.h"

namespace capnet {

typedef float pf32x4 __attribute__((ext_vector_type(4)));

constexpr int kPH = 512;            // hidden size
constexpr int kPShards = 8;
constexpr int kPSlots = 32;
constexpr int kPUnits = 16;         // hidden units per slot
constexpr int kPGrid = kPShards * kPSlots;
constexpr int kPMaxRows = 16;       // rows per shard (b <= 128)
// control block (ints): flags [shard][slot][wave] | xcc table [shard][slot] | abort
constexpr int kCtlFlags = 0;
constexpr int kCtlXcc = kPGrid * 4;
constexpr int kCtlAbort = kCtlXcc + kPGrid;
constexpr int kCtlInts = 2048;
constexpr int kSpinBound = 1 << 21;

struct PersistArgs {
  const float* Wp;
  float* G;
  float* Cst;
  float* hiddens;
  int* ctl;
  int* err_flag;
  unsigned long long* stamps;   // diagnostics only (tools/probes/persist_phases.py), else nullptr
  int t0, t1, cfg, seg;
  int off[kMaxSteps + 1];
  short b[kMaxSteps];
};

__device__ __forceinline__ float p_sigm(float x) { return 1.f / (1.f + expf(-x)); }

__device__ __forceinline__ int ld_sc1_i32(const int* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_sc1_i32(int* p, int v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_sc1_f32(float* p, float v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// 16-B load that bypasses the vector L1 (served by L2 or beyond); not valid before wait_vm0()
template <int OFF> __device__ __forceinline__ void ld16_sc1(pf32x4& dst, const float* p) {
  asm volatile("global_load_dwordx4 %0, %1, off offset:%2 sc1" : "=v"(dst) : "v"(p), "n"(OFF) : "memory");
}
__device__ __forceinline__ void wait_vm0() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// ---- the product of one step: out[16 rows of the shard][64 gate columns] over this wave's 128 k ----
// Split-f16 arithmetic on v_mfma_f32_16x16x32_f16 (the trunk's scheme, conv_f16x3.hip): x = hi + lo with
// hi = f16(x), lo = f16(x - hi) (the subtraction is exact in fp32), and x w = lo hi' + hi lo' + hi hi' accumulated in
// fp32; the dropped lo lo' term is 2^-22 of a product. W is split once per pack, scaled by 2^kWExp so that the
// residuals of weights down to 2^-22 are normal f16 numbers (|w| < 2^(16 - kWExp) = 64 is the domain; beyond it the
// f16 piece is inf, the loss NaN, and the pack says so: error word bit 5, lstm_persist_pack_kernel); h is split as it is (|h| < 65 504; its residual is a
// subnormal f16 below |h| = 0.12, i.e. an absolute error of 2^-25 per element, fp32's own rounding at |h| = 0.5 --
// subnormal operands run at full rate, tools/probes/native/mfma_f16_denorm.hip).
// 48 MFMAs of 16 cycles per wave and step where the f32 4x4x1 form issued 256 of 8..10 (round 2).
//
// Operand map: the MFMA computes out^T -- A = W (lane l holds gate column n = l & 15 of a 16-column block), B = h
// (lane l holds batch row l & 15), both with the 8 k of k-quarter kq = l >> 4 of a 32-k group -- so that D puts the
// four gates of ONE (batch row, unit) into the four registers of one lane (rows 4 kq + i of D = columns n = 4 kq + i
// = gates i of unit kq of the block): one 16-B LDS write per column block instead of four scattered words. The k of element j of group g is NOT 32 g + 8 kq + j but
//     k(g, kq, j) = 128 wave + 32 g + 16 (j >> 2) + 4 kq + (j & 3):
// any one-to-one map works as long as the weight image uses the same one, and with this one every 16-B load
// instruction of h reads 64 contiguous bytes per row (4 kq x 16 B) instead of every other 16 B of 128.
// Column n of column block nb is gate n & 3 of unit 4 nb + (n >> 2) of the slot: D lands in the reduction buffer as
// [batch row][4 unit + gate], the layout the epilogue reads (rows padded to 68 floats: the 16 lanes of a k-quarter
// write 16 different rows at the same column, 4 banks apart).
constexpr int kWExp = 10;
typedef _Float16 ph16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 ph16x2 __attribute__((ext_vector_type(2)));
typedef float pf32x2 __attribute__((ext_vector_type(2)));
typedef unsigned pu32x4 __attribute__((ext_vector_type(4)));

// (x0, x1) -> packed f16 pairs of the two pieces
__device__ __forceinline__ void p_split2(float x0, float x1, unsigned& h, unsigned& l) {
  const pf32x2 v = {x0, x1};
  const ph16x2 hh = __builtin_convertvector(v, ph16x2);          // v_cvt_pk_f16_f32, round to nearest even
  const pf32x2 r = v - __builtin_convertvector(hh, pf32x2);
  h = __builtin_bit_cast(unsigned, hh);
  l = __builtin_bit_cast(unsigned, __builtin_convertvector(r, ph16x2));
}
__device__ __forceinline__ void p_split8(const pf32x4& x0, const pf32x4& x1, ph16x8& hi, ph16x8& lo) {
  unsigned h0, h1, h2, h3, l0, l1, l2, l3;
  p_split2(x0[0], x0[1], h0, l0);
  p_split2(x0[2], x0[3], h1, l1);
  p_split2(x1[0], x1[1], h2, l2);
  p_split2(x1[2], x1[3], h3, l3);
  const pu32x4 h = {h0, h1, h2, h3}, l = {l0, l1, l2, l3};
  hi = __builtin_bit_cast(ph16x8, h);
  lo = __builtin_bit_cast(ph16x8, l);
}

template <int N> __device__ __forceinline__ void wait_vmcnt() {
  static_assert(N >= 0 && N <= 63, "vmcnt immediate");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// wq[(4 nb + g) * 2 + plane]: the 8 f16 of (column block nb, k group g), plane 0 = hi, 1 = lo
__device__ __forceinline__ void persist_mfma_group(const ph16x8& ahi, const ph16x8& alo, const pf32x4 (&wq)[32], int g,
                                                   pf32x4 (&acc)[4]) {
  // product-major: an accumulator is touched again three MFMAs later, past the instruction's own latency
#pragma unroll
  for (int nb = 0; nb < 4; ++nb)
    acc[nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(ph16x8, wq[(4 * nb + g) * 2]), alo, acc[nb], 0, 0, 0);
#pragma unroll
  for (int nb = 0; nb < 4; ++nb)
    acc[nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(ph16x8, wq[(4 * nb + g) * 2 + 1]), ahi, acc[nb], 0, 0, 0);
#pragma unroll
  for (int nb = 0; nb < 4; ++nb)
    acc[nb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(ph16x8, wq[(4 * nb + g) * 2]), ahi, acc[nb], 0, 0, 0);
}

// The 8 h loads of a step were issued in k order (group g = loads 2 g, 2 g + 1) and vmcnt retires in issue order:
// group g has landed when at most 6 - 2 g + YOUNG operations are outstanding, YOUNG = what this wave issued after
// the h loads (4 pre-activation loads and, if it executed them -- `stores` -- the 5 deferred stores). The MFMAs of
// group g are issued together with the split of group g + 1, whose loads had the time of the previous stage to arrive.
// Only the wait itself branches on `stores`: with the whole product duplicated under the two counts, hipcc hoisted
// the arithmetic the two arms had in common -- the first split -- above both waits (tools/isa_inflight_check.py).
template <int N> __device__ __forceinline__ void wait_h_loads(bool stores) {
  if (stores) wait_vmcnt<N + 9>();
  else wait_vmcnt<N + 4>();
}
__device__ __forceinline__ void persist_mfma(const pf32x4 (&hv)[8], const pf32x4 (&wq)[32], pf32x4 (&acc)[4], bool stores) {
#pragma unroll
  for (int nb = 0; nb < 4; ++nb) acc[nb] = pf32x4{0.f, 0.f, 0.f, 0.f};
  ph16x8 ahi[2], alo[2];
  // (input-only markers for tools/isa_inflight_check.py: a "+v" tie here made hipcc copy the still-stale registers
  //  above the wait -- mfma_core.h, CAPNET_LANDED_IN)
  wait_h_loads<6>(stores);
  CAPNET_LANDED_IN1(hv[0]);
  CAPNET_LANDED_IN1(hv[1]);
  __builtin_amdgcn_sched_barrier(0);
  p_split8(hv[0], hv[1], ahi[0], alo[0]);
  // (the empty statements pin a split to the stage it is written in: hipcc otherwise sinks it past the next wait, to
  //  just in front of its MFMAs, and nothing overlaps)
  asm volatile("" : "+v"(ahi[0]), "+v"(alo[0]));
#define CAPNET_PSTAGE(G)                                                                      \
  __builtin_amdgcn_sched_barrier(0);                                                            \
  wait_h_loads<4 - 2 * (G)>(stores);                                                           \
  CAPNET_LANDED_IN1(hv[2 * (G) + 2]);                                                           \
  CAPNET_LANDED_IN1(hv[2 * (G) + 3]);                                                           \
  __builtin_amdgcn_sched_barrier(0);                                                            \
  p_split8(hv[2 * (G) + 2], hv[2 * (G) + 3], ahi[((G) + 1) & 1], alo[((G) + 1) & 1]);           \
  persist_mfma_group(ahi[(G) & 1], alo[(G) & 1], wq, G, acc);                                  \
  asm volatile("" : "+v"(ahi[((G) + 1) & 1]), "+v"(alo[((G) + 1) & 1]));
  CAPNET_PSTAGE(0) CAPNET_PSTAGE(1) CAPNET_PSTAGE(2)
#undef CAPNET_PSTAGE
  persist_mfma_group(ahi[1], alo[1], wq, 3, acc);
}

// sigmoid / tanh on v_exp_f32 + v_rcp_f32 (about 1 ulp each): the pointwise part sits on the
// step's critical path between the last MFMA and the store of h
__device__ __forceinline__ float fast_sigm(float x) {
  return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
}
__device__ __forceinline__ float fast_tanh(float x) {
  return 1.f - 2.f * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(2.8853900817779268f * x));
}

// DIAG: s_memtime stamps (tools/probes/persist_phases.py); the product instantiation has none
template <bool DIAG>
__global__ __launch_bounds__(256) void lstm_persist_kernel(const PersistArgs a) {
  constexpr int H = kPH;
  __shared__ __attribute__((aligned(16))) float red[2][4][kPMaxRows][68];
  __shared__ int s_local, s_abort;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int shard = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int lm = lane & 15, kq = lane >> 4;      // MFMA operand role: row (A) / column (B, D) and k quarter
  int* ctl = a.ctl;
  const int gi = a.cfg & 3, gf = (a.cfg >> 2) & 3, go = (a.cfg >> 4) & 3, gg = (a.cfg >> 6) & 3;
  const int tanh_out = (a.cfg >> 8) & 1;
  const int u0 = slot * kPUnits;

  // ---- weights of this wave: 128 VGPRs (16 (column block, k group) pairs x {hi, lo} x 8 f16), requested first so
  // that the handshake hides behind them
  pf32x4 wq[32];
  {
    const pf32x4* wp = reinterpret_cast<const pf32x4*>(a.Wp) + ((long)(slot * 4 + wave) * 32) * 64 + lane;
#pragma unroll
    for (int q = 0; q < 32; ++q) wq[q] = wp[(long)q * 64];
  }
  // ---- placement handshake: publish my XCC id, read the shard's 32 entries
  if (tid == 0) {
    s_abort = 0;
    unsigned x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    st_sc1_i32(&ctl[kCtlXcc + shard * kPSlots + slot], (a.seg << 4) | (int)(x & 15));
  }
  if (wave == 0) {
    int spins = 0, ok = 0, same = 0;
    while (true) {
      const int v = ld_sc1_i32(&ctl[kCtlXcc + shard * kPSlots + (lane & 31)]);
      const int mine = __shfl(v, slot, 64);
      ok = __all((v >> 4) == a.seg);
      same = __all(v == mine);
      if (ok) break;
      if (++spins > kSpinBound || ((spins & 255) == 0 && ld_sc1_i32(&ctl[kCtlAbort]) != 0)) break;
      __builtin_amdgcn_s_sleep(4);
    }
    if (a.cfg & 1024) ok = 0;              // test hook: pretend the handshake timed out
    if (lane == 0) {
      s_local = ok && same && !(a.cfg & 512);   // (bit 9: SAFE mode forced)
      if (!ok) {
        s_abort = 1;
        st_sc1_i32(&ctl[kCtlAbort], 1);
        atomicOr(a.err_flag, 4);
      }
    }
  }
  __syncthreads();
  if (s_abort) return;
  const bool local = s_local != 0;
  if (tid == 0 && slot == 0) ctl[kCtlAbort + 1 + shard] = local ? 1 : 2;   // diagnostics: mode taken
  if (tid == 0 && blockIdx.x == 0 && reinterpret_cast<const unsigned*>(a.Wp)[4l * kPH * kPH] != 0u) atomicOr(a.err_flag, 32);

  // ---- epilogue role: thread e -> (row m of the shard, unit u of the slot)
  const int em = tid >> 4, eu = tid & 15;
  const int grow = 8 * em + shard;          // global batch row
  float c_reg = 0.f;
  if (a.t0 > 0 && grow < a.b[a.t0 - 1])
    c_reg = a.Cst[(long)(a.off[a.t0 - 1] + grow) * H + u0 + eu];
  // flags this wave polls: the 8 slots that produce k in [128 wave, 128 wave + 128), 4 waves each
  const int* my_flags = ctl + kCtlFlags + (shard * kPSlots + 8 * wave) * 4 + (lane & 31);
  int* out_flag = ctl + kCtlFlags + (shard * kPSlots + slot) * 4 + wave;

  // step metadata in two VGPRs (lane l: steps l and l + 64), read per step with v_readlane: a
  // scalar load from the argument segment per step would be a dependent fetch on the critical path
  unsigned long long rt0 = 0, mt0 = 0;
  if (DIAG) { rt0 = __builtin_amdgcn_s_memrealtime(); mt0 = __builtin_amdgcn_s_memtime(); }
  int v_off0 = a.off[lane], v_off1 = a.off[64 + lane < kMaxSteps ? 64 + lane : kMaxSteps];
  int v_b0 = a.b[lane], v_b1 = a.b[64 + lane < kMaxSteps ? 64 + lane : kMaxSteps - 1];
  // complete the four loads HERE and hide their origin: otherwise hipcc's waitcnt pass treats them
  // as possibly pending around the loop's back edge and drains vmcnt(0) in front of every readlane
  wait_vm0();
  asm volatile("" : "+v"(v_off0), "+v"(v_off1), "+v"(v_b0), "+v"(v_b1));
  auto off_of = [&](int t) { return t < 64 ? __builtin_amdgcn_readlane(v_off0, t) : __builtin_amdgcn_readlane(v_off1, t - 64); };
  auto b_of = [&](int t) { return t < 64 ? __builtin_amdgcn_readlane(v_b0, t) : __builtin_amdgcn_readlane(v_b1, t - 64); };
  // Every vector-memory instruction of the loop is inline asm with hand-counted vmcnt (in issue
  // order per step: 8 h loads, 4 pre-activation loads, [5 deferred stores of the previous step],
  // the h store). hipcc's own loads in a loop make its waitcnt pass drain vmcnt(0) at points it
  // cannot see our in-flight operations from; only the poll is a compiler-visible (atomic) load,
  // and at that point nothing but the previous flag store is in flight.
  pf32x4 dgate = {0.f, 0.f, 0.f, 0.f};    // activated gates + c of the previous step, stored one
  float dcell = 0.f;                       // step late, behind the next step's h loads
  long drow = -1;

  auto st32 = [](float* p, float v) { asm volatile("global_store_dword %0, %1, off" :: "v"(p), "v"(v) : "memory"); };
  auto flush_deferred = [&]() {
    if (drow >= 0) {
      float* gp = a.G + drow * (4 * H) + u0 + eu;
      st32(gp + (long)gi * H, dgate[0]);
      st32(gp + (long)gf * H, dgate[1]);
      st32(gp + (long)go * H, dgate[2]);
      st32(gp + (long)gg * H, dgate[3]);
      st32(a.Cst + drow * H + u0 + eu, dcell);
    }
  };

  for (int t = a.t0; t < a.t1; ++t) {
    unsigned long long ts[6] = {0, 0, 0, 0, 0, 0};
    if (DIAG) ts[0] = __builtin_amdgcn_s_memtime();
    const int bt = b_of(t);
    const int rows_t = (bt - shard + 7) >> 3;        // rows of this shard still alive
    const int ro = off_of(t);
    const bool evalid = em < rows_t;                  // (rows_t <= 16 by construction)
    const long erow = ro + (evalid ? grow : 0);       // row `ro` always exists (b_t >= 1)
    const bool product = t > 0;
    pf32x4 hv[8];
    if (product) {
      // ---- wait for h_{t-1}: produced inside this launch for t > t0, by earlier launches at t0
      if (t > a.t0) {
        int spins = 0;
        while (true) {
          const int v = ld_sc1_i32(my_flags);
          if (__all(v - t >= 0)) break;
          if (++spins > kSpinBound || ((spins & 255) == 0 && ld_sc1_i32(&ctl[kCtlAbort]) != 0)) {
            if (lane == 0) {
              st_sc1_i32(&ctl[kCtlAbort], 1);
              atomicOr(a.err_flag, 4);
            }
            break;
          }
        }
      }
      if (DIAG) ts[1] = __builtin_amdgcn_s_memtime();
      // ---- A operands: load q of lane (lm, kq) = h[row lm][128 wave + 16 q + 4 kq + e], e < 4 (see the k map above)
      const int rp = off_of(t - 1);
      // Only the lanes of live rows load (a dead row costs no L2 traffic; at b = 64 that is half of it). What the
      // registers of the other lanes hold feeds output columns nobody reads.
      const float* hrow = a.hiddens + (long)(rp + 8 * lm + shard) * H + 128 * wave + 4 * kq;
      if (lm < rows_t) {
        ld16_sc1<0>(hv[0], hrow);
        ld16_sc1<64>(hv[1], hrow);
        ld16_sc1<128>(hv[2], hrow);
        ld16_sc1<192>(hv[3], hrow);
        ld16_sc1<256>(hv[4], hrow);
        ld16_sc1<320>(hv[5], hrow);
        ld16_sc1<384>(hv[6], hrow);
        ld16_sc1<448>(hv[7], hrow);
      } else {
#pragma unroll
        for (int q = 0; q < 8; ++q) asm volatile("" : "=v"(hv[q]));
      }
    }
    // ---- pre-activations of this step (written by the input-chain GEMMs before this launch):
    // behind the h loads, they have the whole product to arrive
    pf32x4 pre;
    {
      const float* gp = a.G + erow * (4 * H) + u0 + eu;
      asm volatile("global_load_dword %0, %1, off" : "=v"(pre[0]) : "v"(gp + (long)gi * H) : "memory");
      asm volatile("global_load_dword %0, %1, off" : "=v"(pre[1]) : "v"(gp + (long)gf * H) : "memory");
      asm volatile("global_load_dword %0, %1, off" : "=v"(pre[2]) : "v"(gp + (long)go * H) : "memory");
      asm volatile("global_load_dword %0, %1, off" : "=v"(pre[3]) : "v"(gp + (long)gg * H) : "memory");
    }
    // ---- gates / cell state of the previous step: off the critical path, behind this step's loads
    const bool had_deferred = drow >= 0;
    flush_deferred();
    drow = -1;
    __builtin_amdgcn_sched_barrier(0);

    if (product) {
      // Younger than the h loads: 4 loads and, if this wave executed them, the 5 deferred stores (the wave executes
      // them iff one of its lanes had a live row; if the count assumed here were too low the wait would only be
      // longer, never shorter).
      pf32x4 acc[4];
      const bool stores_in_flight = __any(had_deferred);
      persist_mfma(hv, wq, acc, stores_in_flight);
      if (DIAG) ts[2] = __builtin_amdgcn_s_memtime();
      // D: register i of lane (lm, kq) of column block nb = out[batch row lm][unit 4 nb + kq][gate i]
      if (lm < rows_t) {
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) *reinterpret_cast<pf32x4*>(&red[t & 1][wave][lm][4 * (4 * nb + kq)]) = acc[nb];
      }
    }
    wait_vm0();     // pre-activations (and the deferred stores, long gone)
    __builtin_amdgcn_sched_barrier(0);
    if (product) {
      __syncthreads();
      if (evalid) {
        pf32x4 sum = *reinterpret_cast<const pf32x4*>(&red[t & 1][0][em][4 * eu]);
#pragma unroll
        for (int w = 1; w < 4; ++w) sum += *reinterpret_cast<const pf32x4*>(&red[t & 1][w][em][4 * eu]);
        pre += sum * (1.f / (float)(1 << kWExp));
      }
    }
    if (DIAG) ts[3] = __builtin_amdgcn_s_memtime();
    if (evalid) {
      const float i = fast_sigm(pre[0]), f = fast_sigm(pre[1]), og = fast_sigm(pre[2]), gt = fast_tanh(pre[3]);
      c_reg = f * c_reg + i * gt;
      const float h = tanh_out ? og * fast_tanh(c_reg) : og * c_reg;
      float* hp = a.hiddens + erow * H + u0 + eu;
      if (local) st32(hp, h);
      else asm volatile("global_store_dword %0, %1, off sc1" :: "v"(hp), "v"(h) : "memory");
      dgate = pf32x4{i, f, og, gt};
      dcell = c_reg;
      drow = erow;
    }
    if (t + 1 < a.t1) {
      // every wave signals for its own stores: the h store is the only operation in flight
      wait_vm0();
      if (DIAG) ts[4] = __builtin_amdgcn_s_memtime();
      if (lane == 0) {
        // (a C++ volatile store becomes flat_store sc0 sc1 + vmcnt(0): write-through and a drain)
        if (local) asm volatile("global_store_dword %0, %1, off" :: "v"(out_flag), "v"(t + 1) : "memory");
        else asm volatile("global_store_dword %0, %1, off sc1" :: "v"(out_flag), "v"(t + 1) : "memory");
      }
    }
    if (DIAG) {
      ts[5] = __builtin_amdgcn_s_memtime();
      if (tid == 0) {
        unsigned long long* st = a.stamps + ((long)(t - a.t0) * kPGrid + blockIdx.x) * 8;
#pragma unroll
        for (int k = 0; k < 6; ++k) st[k] = ts[k];
      }
      wait_vm0();
    }
  }
  flush_deferred();
  if (DIAG && tid == 0) {     // shader clock over the launch: d(s_memtime) / d(s_memrealtime) x 100 MHz
    unsigned long long* st = a.stamps + (long)(a.t1 - a.t0) * kPGrid * 8 + blockIdx.x * 2;
    st[0] = __builtin_amdgcn_s_memtime() - mt0;
    st[1] = __builtin_amdgcn_s_memrealtime() - rt0;
  }
}

// The image, in 32-bit words (two f16 each): word [((slot*4 + w)*32 + (4 nb + g)*2 + plane)*256 + lane*4 + jp] holds
// elements j = 2 jp, 2 jp + 1 of plane hi / lo of  2^kWExp Wcat[grow[n & 3]*H + 16 slot + 4 nb + (n >> 2)][k(g, kq, j)],
// lane = 16 kq + n (the operand map above the kernel). As many bytes as the fp32 matrix.
__global__ __launch_bounds__(256) void lstm_persist_pack_kernel(const float* __restrict__ Wcat,
                                                                unsigned* __restrict__ Wp, int g0,
                                                                int g1, int g2, int g3) {
  const int grow[4] = {g0, g1, g2, g3};
  const long total = 2L * kPH * kPH;                 // pairs of weights
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long)gridDim.x * blockDim.x) {
    const int jp = (int)(i & 3), lane = (int)((i >> 2) & 63);
    long r = i >> 8;
    const int g = (int)(r & 3); r >>= 2;
    const int nb = (int)(r & 3); r >>= 2;
    const int w = (int)(r & 3);
    const int slot = (int)(r >> 2);
    const int n = lane & 15, kq = lane >> 4;
    const int k = 128 * w + 32 * g + 16 * (jp >> 1) + 4 * kq + 2 * (jp & 1);
    const float* src = Wcat + ((long)grow[n & 3] * kPH + 16 * slot + 4 * nb + (n >> 2)) * kPH + k;
    unsigned h, l;
    p_split2(src[0] * (float)(1 << kWExp), src[1] * (float)(1 << kWExp), h, l);
    const long q = ((long)(slot * 4 + w) * 32 + (4 * nb + g) * 2) * 256 + lane * 4 + jp;
    Wp[q] = h;
    Wp[q + 256] = l;
    // a weight whose scaled value leaves f16's range (|w| >= 2^(16 - kWExp) = 64; NaN included) becomes inf in the image: the
    // word behind the image says so and the sequence kernel raises bit 5 of the error word (the caller's remedy: the launch-per-
    // step path, CAPNET_NO_PERSISTENT_LSTM=1)
    const float lim = 65504.f / (float)(1 << kWExp);
    if (!(fabsf(src[0]) <= lim && fabsf(src[1]) <= lim)) Wp[4l * kPH * kPH] = 1u;
  }
}

size_t lstm_persist_w_floats() { return 4ul * kPH * kPH + 4; }   // the image + its out-of-domain word
size_t lstm_persist_ctl_ints() { return kCtlInts; }

// Process-wide mode of the persistent path: bit 0 off (launch per step everywhere), bit 1 SAFE mode forced
// (write-through hand-off although the shard shares an XCD), bit 2 inject a handshake timeout (tests).
// Starts from CAPNET_NO_PERSISTENT_LSTM=1 in the environment.
static int& persist_mode() {
  static int mode = [] { const char* off = getenv("CAPNET_NO_PERSISTENT_LSTM"); return (off && off[0] == '1') ? 1 : 0; }();
  return mode;
}
int lstm_persist_set_mode(int mode) {
  const int old = persist_mode();
  if (mode >= 0) persist_mode() = mode & 7;
  return old;
}

// One-time residency check: what a cooperative launch would verify (all 256 workgroups can be
// resident at once on an idle device), without paying its per-launch cost.
static int persist_device_ok() {
  static int cached = -1;
  if (cached >= 0) return cached;
  int dev = 0, cus = 0, per_cu = 0;
  if (hipGetDevice(&dev) != hipSuccess ||
      hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) {
    (void)hipGetLastError();
    return cached = 0;
  }
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)lstm_persist_kernel<false>, 256,
                                                   0) != hipSuccess) {
    (void)hipGetLastError();
    return cached = 0;
  }
  cached = (cus >= kPGrid && per_cu >= 1) ? 1 : 0;
  return cached;
}

bool lstm_persist_supported(int b, int H) {
  return H == kPH && b >= 1 && b <= 8 * kPMaxRows && !(persist_mode() & 1) && persist_device_ok();
}

int lstm_persist_pack(const float* Wcat, float* Wp, int gi, int gf, int go, int gg, hipStream_t stream) {
  CAPNET_REQUIRE(Wcat && Wp && aligned16(Wp), "lstm_persist_pack: bad argument");
  CAPNET_HIP_CHECK(hipMemsetAsync(Wp + 4l * kPH * kPH, 0, 4 * sizeof(float), stream));
  hipLaunchKernelGGL(lstm_persist_pack_kernel, dim3(1024), dim3(256), 0, stream, Wcat, reinterpret_cast<unsigned*>(Wp), gi, gf,
                     go, gg);
  CAPNET_LAUNCH_CHECK();
  return kOk;
}

int lstm_persist_run(const float* Wp, float* G, float* Cst, float* hiddens, const int* off,
                     const int* batch_sizes, int t0, int t1, int H, int gi, int gf, int go, int gg,
                     int tanh_out, int seg, int* ctl, int* err_flag, hipStream_t stream,
                     unsigned long long* stamps) {
  CAPNET_REQUIRE(Wp && G && Cst && hiddens && off && batch_sizes && ctl && err_flag,
                 "lstm_persist_run: null argument");
  CAPNET_REQUIRE(0 <= t0 && t0 < t1 && t1 <= kMaxSteps, "lstm_persist_run: steps [%d, %d)", t0, t1);
  CAPNET_REQUIRE(seg > 0 && seg < (1 << 26), "lstm_persist_run: segment tag %d", seg);
  const int b0 = batch_sizes[t0 > 0 ? t0 - 1 : t0];
  CAPNET_REQUIRE(lstm_persist_supported(b0, H), "lstm_persist_run: unsupported b=%d H=%d", b0, H);
  CAPNET_REQUIRE(aligned16(Wp) && aligned16(hiddens), "lstm_persist_run: alignment");
  PersistArgs a;
  a.Wp = Wp; a.G = G; a.Cst = Cst; a.hiddens = hiddens; a.ctl = ctl; a.err_flag = err_flag;
  a.stamps = stamps;
  a.t0 = t0; a.t1 = t1; a.seg = seg;
  a.cfg = gi | (gf << 2) | (go << 4) | (gg << 6) | ((tanh_out ? 1 : 0) << 8) | ((persist_mode() & 2) ? 512 : 0) |
          ((persist_mode() & 4) ? 1024 : 0);
  for (int t = 0; t <= t1; ++t) a.off[t] = off[t];
  int prev = 1 << 30;
  for (int t = (t0 > 0 ? t0 - 1 : 0); t < t1; ++t) {
    CAPNET_REQUIRE(batch_sizes[t] >= 1 && batch_sizes[t] <= prev, "lstm_persist_run: batch sizes must not grow");
    prev = batch_sizes[t];
  }
  for (int t = 0; t < t1; ++t) a.b[t] = (short)batch_sizes[t];
  if (stamps) hipLaunchKernelGGL((lstm_persist_kernel<true>), dim3(kPGrid), dim3(256), 0, stream, a);
  else hipLaunchKernelGGL((lstm_persist_kernel<false>), dim3(kPGrid), dim3(256), 0, stream, a);
  CAPNET_LAUNCH_CHECK();
  return kOk;
}

}  // namespace capnet
