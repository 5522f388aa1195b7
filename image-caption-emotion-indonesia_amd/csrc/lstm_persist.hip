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
//     of fp32 weights = 128 VGPRs per lane, loaded once per launch. Wave w holds k in
//     [128 w, 128 w + 128): split-K over the 4 waves, summed through LDS.
//   * the product uses v_mfma_f32_4x4x1_16B_f32: 16 independent 4x4 outer products per instruction
//     at the full f32 matrix rate (64 FLOP/clk/SIMD), so 8 rows per shard waste nothing (a 16x16x4
//     tile would be half empty at b = 64). Block bl of an instruction = gate columns of unit bl of
//     the slot; its A operand (4 rows x one k) comes from ONE VGPR that holds 16 different k (lane
//     4 bl + i <-> row i, k0 + bl): the instruction's CBSZ = 4 / ABID = bl' broadcast selects which
//     k. h_{t-1} therefore goes from global memory straight into MFMA operands: no LDS staging.
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
  unsigned long long* stamps;   // diagnostics only (tools/persist_phases.py), else nullptr
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
__device__ __forceinline__ void ld16_sc1(pf32x4& dst, const float* p) {
  asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(dst) : "v"(p) : "memory");
}
__device__ __forceinline__ void wait_vm0() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// out[r] = rows 4r..4r+3 of h . W^T over this wave's 128 k, for r < NRB.
// v_mfma_f32_4x4x1 needs ~50 cycles before its result can be accumulated into again
// (tools/native/mfma4x4_rate.hip: 48 / 28 / 16 / 12 cycles per instruction with 1 / 2 / 4 / 8
// independent accumulators), so every row block accumulates into NS independent chains that are
// summed at the end: 8 chains for one live row block, 4 for two or three, 2 for four.
// (ABID must be an immediate: one macro instance per broadcast block.)
constexpr int persist_chains(int nrb) { return nrb == 1 ? 8 : nrb == 4 ? 2 : 4; }
constexpr int persist_max_chains(int rb) { return rb == 1 ? 8 : rb == 2 ? 8 : 12; }

template <int RB, int NRB, int Q>
__device__ __forceinline__ void persist_mfma_half(const pf32x4 (&av)[RB][2], const pf32x4 (&wq)[32],
                                                  pf32x4 (&acc)[persist_max_chains(RB)]) {
  constexpr int NS = persist_chains(NRB);
#define CAPNET_PBLK(B2)                                                                             \
  _Pragma("unroll") for (int e = 0; e < 4; ++e)                                                     \
  _Pragma("unroll") for (int r = 0; r < NRB; ++r)                                                   \
      acc[r * NS + ((e + 4 * (B2 & 1)) % NS)] = __builtin_amdgcn_mfma_f32_4x4x1f32(                 \
          av[r][Q][e], wq[16 * Q + B2][e], acc[r * NS + ((e + 4 * (B2 & 1)) % NS)], 4, B2, 0);
  CAPNET_PBLK(0) CAPNET_PBLK(1) CAPNET_PBLK(2) CAPNET_PBLK(3)
  CAPNET_PBLK(4) CAPNET_PBLK(5) CAPNET_PBLK(6) CAPNET_PBLK(7)
  CAPNET_PBLK(8) CAPNET_PBLK(9) CAPNET_PBLK(10) CAPNET_PBLK(11)
  CAPNET_PBLK(12) CAPNET_PBLK(13) CAPNET_PBLK(14) CAPNET_PBLK(15)
#undef CAPNET_PBLK
}

// the k half Q = 0 runs while the loads of half Q = 1 are still in flight (vmcnt retires in issue
// order: all but the RB youngest loads done = half 0 has landed)
template <int N> __device__ __forceinline__ void wait_vmcnt() {
  static_assert(N >= 0 && N <= 13, "vmcnt immediate");
  if (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (N == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
  if (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  if (N == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
  if (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  if (N == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
  if (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  if (N == 7) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
  if (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  if (N == 9) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
  if (N == 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
  if (N == 11) asm volatile("s_waitcnt vmcnt(11)" ::: "memory");
  if (N == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
  if (N == 13) asm volatile("s_waitcnt vmcnt(13)" ::: "memory");
}

// The k half Q = 0 runs while the loads of half Q = 1 are still in flight. vmcnt retires in issue
// order; YOUNG = operations this wave issued after the h loads (pre-activation loads, deferred stores).
template <int RB, int NRB, int YOUNG>
__device__ __forceinline__ void persist_mfma(const pf32x4 (&av)[RB][2], const pf32x4 (&wq)[32],
                                             pf32x4 (&out)[RB]) {
  constexpr int NS = persist_chains(NRB);
  pf32x4 acc[persist_max_chains(RB)];
#pragma unroll
  for (int c = 0; c < NRB * NS; ++c) acc[c] = pf32x4{0.f, 0.f, 0.f, 0.f};
  // (input-only markers for tools/isa_inflight_check.py: a "+v" tie here made hipcc copy the still-stale registers
  //  above the wait in some of the eight instantiations a step branches to -- mfma_core.h, CAPNET_LANDED_IN)
  wait_vmcnt<RB + YOUNG>();
#pragma unroll
  for (int r = 0; r < RB; ++r) CAPNET_LANDED_IN1(av[r][0]);
  __builtin_amdgcn_sched_barrier(0);
  persist_mfma_half<RB, NRB, 0>(av, wq, acc);
  __builtin_amdgcn_sched_barrier(0);
  wait_vmcnt<YOUNG>();
#pragma unroll
  for (int r = 0; r < RB; ++r) CAPNET_LANDED_IN1(av[r][1]);
  __builtin_amdgcn_sched_barrier(0);
  persist_mfma_half<RB, NRB, 1>(av, wq, acc);
#pragma unroll
  for (int r = 0; r < RB; ++r) {
    if (r < NRB) {
      pf32x4 s = acc[r * NS];
#pragma unroll
      for (int c = 1; c < NS; ++c) s += acc[r * NS + c];
      out[r] = s;
    } else {
      out[r] = pf32x4{0.f, 0.f, 0.f, 0.f};
    }
  }
}

// sigmoid / tanh on v_exp_f32 + v_rcp_f32 (about 1 ulp each): the pointwise part sits on the
// step's critical path between the last MFMA and the store of h
__device__ __forceinline__ float fast_sigm(float x) {
  return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
}
__device__ __forceinline__ float fast_tanh(float x) {
  return 1.f - 2.f * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(2.8853900817779268f * x));
}

// RB: 4-row blocks per shard (rows per shard <= 4 RB)
// DIAG: s_memtime stamps (tools/persist_phases.py); the product instantiation has none
template <int RB, bool DIAG>
__global__ __launch_bounds__(256) void lstm_persist_kernel(const PersistArgs a) {
  constexpr int H = kPH;
  __shared__ __attribute__((aligned(16))) float red[2][4][4 * RB][64];
  __shared__ int s_local, s_abort;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int shard = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int bl = lane >> 2, li = lane & 3;
  int* ctl = a.ctl;
  const int gi = a.cfg & 3, gf = (a.cfg >> 2) & 3, go = (a.cfg >> 4) & 3, gg = (a.cfg >> 6) & 3;
  const int tanh_out = (a.cfg >> 8) & 1;
  const int u0 = slot * kPUnits;

  // ---- weights of this wave: 128 VGPRs, requested first so that the handshake hides behind them
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

  // ---- epilogue role: thread e -> (row m of the shard, unit u of the slot)
  const int em = tid >> 4, eu = tid & 15;
  const int grow = 8 * em + shard;          // global batch row
  float c_reg = 0.f;
  if (a.t0 > 0 && em < 4 * RB && grow < a.b[a.t0 - 1])
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
  // order per step: [5 deferred stores of the previous step] 2 RB h loads, 4 pre-activation loads,
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
    const bool evalid = em < rows_t;                  // (rows_t <= 4 RB by construction)
    const long erow = ro + (evalid ? grow : 0);       // row `ro` always exists (b_t >= 1)
    const bool product = t > 0;
    pf32x4 av[RB][2];
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
      // ---- A operands: lane (bl, li) of row block r holds h[row 4 r + li][128 wave + 64 Q + 4 bl + e]
      const int rp = off_of(t - 1);
      const float* hrow[RB];
#pragma unroll
      for (int r = 0; r < RB; ++r) {
        const int m = 4 * r + li;
        const int g = m < rows_t ? 8 * m + shard : 0;   // clamped rows only feed ignored outputs
        hrow[r] = a.hiddens + (long)(rp + g) * H + 128 * wave + 4 * bl;
      }
#pragma unroll
      for (int r = 0; r < RB; ++r) ld16_sc1(av[r][0], hrow[r]);
#pragma unroll
      for (int r = 0; r < RB; ++r) ld16_sc1(av[r][1], hrow[r] + 64);
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
      // Younger than the first k half of the h loads: RB loads, 4 loads and, if this wave executed
      // them, the 5 deferred stores (the wave executes them iff one of its lanes had a live row; if
      // the count assumed here were too low the wait would only be longer, never shorter).
      pf32x4 sum[RB];
      const int nrb = (rows_t + 3) >> 2;
      const bool stores_in_flight = __any(had_deferred);
      if (stores_in_flight) {
        if (RB >= 4 && nrb >= 4) persist_mfma<RB, (RB >= 4 ? 4 : RB), 9>(av, wq, sum);
        else if (RB >= 3 && nrb == 3) persist_mfma<RB, (RB >= 3 ? 3 : RB), 9>(av, wq, sum);
        else if (RB >= 2 && nrb == 2) persist_mfma<RB, (RB >= 2 ? 2 : RB), 9>(av, wq, sum);
        else persist_mfma<RB, 1, 9>(av, wq, sum);
      } else {
        if (RB >= 4 && nrb >= 4) persist_mfma<RB, (RB >= 4 ? 4 : RB), 4>(av, wq, sum);
        else if (RB >= 3 && nrb == 3) persist_mfma<RB, (RB >= 3 ? 3 : RB), 4>(av, wq, sum);
        else if (RB >= 2 && nrb == 2) persist_mfma<RB, (RB >= 2 ? 2 : RB), 4>(av, wq, sum);
        else persist_mfma<RB, 1, 4>(av, wq, sum);
      }
      if (DIAG) ts[2] = __builtin_amdgcn_s_memtime();
      // D: register i, lane 4 bl + j = out[row 4 r + i][unit bl, gate j]
#pragma unroll
      for (int r = 0; r < RB; ++r)
#pragma unroll
        for (int i = 0; i < 4; ++i) red[t & 1][wave][4 * r + i][lane] = sum[r][i];
    }
    wait_vm0();     // pre-activations (and the deferred stores, long gone)
    __builtin_amdgcn_sched_barrier(0);
    if (product) {
      __syncthreads();
      if (evalid) {
#pragma unroll
        for (int w = 0; w < 4; ++w) pre += *reinterpret_cast<const pf32x4*>(&red[t & 1][w][em][4 * eu]);
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

// Wp[((slot*4 + w)*32 + q)*256 + lane*4 + e] = Wcat[grow[j]*H + 16 slot + bl][128 w + 4 q + e],
// lane = 4 bl + j (bl: unit of the slot, j: gate role i,f,o,g)
__global__ __launch_bounds__(256) void lstm_persist_pack_kernel(const float* __restrict__ Wcat,
                                                                float* __restrict__ Wp, int g0,
                                                                int g1, int g2, int g3) {
  const int grow[4] = {g0, g1, g2, g3};
  const long total = 4L * kPH * kPH;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long)gridDim.x * blockDim.x) {
    const int e = (int)(i & 3), lane = (int)((i >> 2) & 63);
    long r = i >> 8;
    const int q = (int)(r & 31); r >>= 5;
    const int w = (int)(r & 3);
    const int slot = (int)(r >> 2);
    const int bl = lane >> 2, j = lane & 3;
    Wp[i] = Wcat[((long)grow[j] * kPH + 16 * slot + bl) * kPH + 128 * w + 4 * q + e];
  }
}

size_t lstm_persist_w_floats() { return 4ul * kPH * kPH; }
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
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)lstm_persist_kernel<4, false>, 256,
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
  hipLaunchKernelGGL(lstm_persist_pack_kernel, dim3(1024), dim3(256), 0, stream, Wcat, Wp, gi, gf, go, gg);
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
  const int rows0 = (batch_sizes[t0] + 7) / 8;        // rows of shard 0 at the first step
  const int rb = (rows0 + 3) / 4;
#define CAPNET_PLAUNCH(RBV)                                                                          \
  do {                                                                                              \
    if (stamps) hipLaunchKernelGGL((lstm_persist_kernel<RBV, true>), dim3(kPGrid), dim3(256), 0, stream, a); \
    else hipLaunchKernelGGL((lstm_persist_kernel<RBV, false>), dim3(kPGrid), dim3(256), 0, stream, a);       \
  } while (0)
  switch (rb) {
    case 1: CAPNET_PLAUNCH(1); break;
    case 2: CAPNET_PLAUNCH(2); break;
    case 3: CAPNET_PLAUNCH(3); break;
    default: CAPNET_PLAUNCH(4); break;
  }
#undef CAPNET_PLAUNCH
  CAPNET_LAUNCH_CHECK();
  return kOk;
}

}  // namespace capnet
