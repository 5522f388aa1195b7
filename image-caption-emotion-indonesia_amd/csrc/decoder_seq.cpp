// Sequence drivers for the caption decoders: one C call runs the whole scheduled-sampling
// recurrence (forward) or the whole BPTT (backward) as a chain of HIP launches on one stream.
//
// Follows DecoderFactoredLSTM.forward / forward_step (stylenet/model.py:115-196) and
// DecoderRNN.forward (nic/model.py:74-115):
//   * inputs are [image feature, dropout(B(w_0)), ..., dropout(B(w_{L-2}))], packed time-major
//     (pack_padded_sequence order, batch shrinking with `batch_sizes`);
//   * step t is teacher forced iff tf_mask[t] (the caller draws random.random() per step,
//     model.py:181); otherwise its input is B(argmax(C h_{t-1})) WITHOUT dropout (model.py:184),
//     or B(captions[:,0]) at t = 0;
//   * FactoredLSTM gate pre-activation = U_g(S_g(V_g(x))) + W_g(h); c = f*c + i*c~; h = o*c.
// MI355X mapping: the input chain of all teacher-forced rows is three batched MFMA GEMMs over
// N = sum(lengths) rows (gate-concatenated / gate-batched weights); only the recurrent
// 4H x H product and the pointwise gate update run per time step.
#include <vector>

#include "common.h"
#include "kernels.h"

namespace capnet {

namespace {

struct Layout {
  // saved float buffer
  size_t X, A1, A2, G, Cst, Vcat, Scat, Ucat, Wcat, Wfrag, Wp, bV, bS, bUW, total;
  // int buffer
  size_t row_sample, row_col, row_token, prev_row, ctl, itotal;
};

Layout make_layout(const SeqDims& d) {
  Layout L;
  size_t o = 0;
  auto take = [&](size_t n) { size_t r = o; o += (n + 3) / 4 * 4; return r; };
  const size_t N = d.N, E = d.E, F = d.F, H = d.H;
  L.X = take(N * E);
  L.G = take(N * 4 * H);
  L.Cst = take(N * H);
  L.Wcat = take(4 * H * H);
  L.Wfrag = take(H % 16 == 0 ? lstm_wfrag_floats((int)H) : 4);
  L.Wp = take(H == 512 ? lstm_persist_w_floats() : 4);     // image of the persistent sequence kernel
  L.bUW = take(4 * H);
  if (d.cell == kCellFactored) {
    L.A1 = take(N * 4 * F);
    L.A2 = take(N * 4 * F);
    L.Vcat = take(4 * F * E);
    L.Scat = take(4 * F * F);
    L.Ucat = take(4 * H * F);
    L.bV = take(4 * F);
    L.bS = take(4 * F);
  } else {
    L.A1 = L.A2 = L.Scat = L.Ucat = L.bV = L.bS = 0;
    L.Vcat = take(4 * H * E);  // weight_ih copy (kept so backward sees the forward's weights)
  }
  L.total = o;
  size_t io = 0;
  auto itake = [&](size_t n) { size_t r = io; io += (n + 3) / 4 * 4; return r; };
  L.row_sample = itake(N);
  L.row_col = itake(N);
  L.row_token = itake(N);
  L.prev_row = itake(N);
  L.ctl = itake(lstm_persist_ctl_ints());
  L.itotal = io;
  return L;
}

int check_dims(const SeqDims& d, const int* batch_sizes) {
  CAPNET_REQUIRE(d.B > 0 && d.T > 0 && d.steps > 0 && d.N > 0 && d.E > 0 && d.H > 0 && d.V > 0,
                 "decoder: bad dims B=%d T=%d steps=%d N=%d E=%d H=%d V=%d", d.B, d.T, d.steps,
                 d.N, d.E, d.H, d.V);
  CAPNET_REQUIRE(d.cell == kCellLSTM || d.F > 0, "decoder: factored size");
  CAPNET_REQUIRE(batch_sizes != nullptr, "decoder: null batch_sizes");
  long n = 0;
  int prev = d.B;
  for (int t = 0; t < d.steps; ++t) {
    CAPNET_REQUIRE(batch_sizes[t] > 0 && batch_sizes[t] <= prev,
                   "decoder: batch_sizes must be positive and non-increasing (step %d: %d after %d)",
                   t, batch_sizes[t], prev);
    prev = batch_sizes[t];
    n += batch_sizes[t];
  }
  CAPNET_REQUIRE(n == d.N, "decoder: sum(batch_sizes)=%ld != N=%d", n, d.N);
  CAPNET_REQUIRE(d.steps <= d.T + (d.has_features ? 1 : 0),
                 "decoder: %d steps need more caption columns than T=%d", d.steps, d.T);
  return kOk;
}

struct GateOrder { int gi, gf, go, gg, tanh_out; };
GateOrder gate_order(int cell) {
  // FactoredLSTM packs i,f,o,c~ ; nn.LSTMCell stores i,f,g,o
  return cell == kCellFactored ? GateOrder{0, 1, 2, 3, 0} : GateOrder{0, 1, 3, 2, 1};
}

#define RC(x) do { int _rc = (x); if (_rc) return _rc; } while (0)

int copy_d2d(float* dst, const float* src, size_t n, hipStream_t s) {
  CAPNET_HIP_CHECK(hipMemcpyAsync(dst, src, n * sizeof(float), hipMemcpyDeviceToDevice, s));
  return kOk;
}

// gate pre-activations of rows [r0, r1) from their inputs X. ws: slab workspace of the per-step
// (few rows) products; the all-rows call up front has enough tiles for the plain kernel.
int input_chain(const SeqDims& d, const Layout& L, float* sv, int r0, int r1, float* ws,
                size_t ws_floats, hipStream_t s, int* ctr = nullptr) {
  const int n = r1 - r0;
  if (n <= 0) return kOk;
  const int E = d.E, F = d.F, H = d.H;
  if (d.cell == kCellFactored) {
    // A1 = X . Vcat^T + bV                          [n x 4F]
    RC(sgemm_splitk(false, true, n, 4 * F, E, sv + L.X + (size_t)r0 * E, E, sv + L.Vcat, E,
                    sv + L.A1 + (size_t)r0 * 4 * F, 4 * F, sv + L.bV, 0, n <= 128 ? ws : nullptr,
                    ws_floats, s, ctr, kSplitKCounters));
    // A2[:, g] = A1[:, g] . S_g^T + bS_g             4 gate groups
    RC(sgemm_splitk_batched(false, true, n, F, F, sv + L.A1 + (size_t)r0 * 4 * F, 4 * F, sv + L.Scat, F,
                            sv + L.A2 + (size_t)r0 * 4 * F, 4 * F, sv + L.bS, 0, 4, F, (long)F * F, F,
                            F, n <= 128 ? ws : nullptr, ws_floats, s, ctr, kSplitKCounters));
    // G[:, g] = A2[:, g] . U_g^T + (bU_g + bW_g)
    RC(sgemm_splitk_batched(false, true, n, H, F, sv + L.A2 + (size_t)r0 * 4 * F, 4 * F, sv + L.Ucat, F,
                            sv + L.G + (size_t)r0 * 4 * H, 4 * H, sv + L.bUW, 0, 4, F, (long)H * F, H,
                            H, n <= 128 ? ws : nullptr, ws_floats, s, ctr, kSplitKCounters));
  } else {
    // G = X . W_ih^T + (b_ih + b_hh)
    RC(sgemm_splitk(false, true, n, 4 * H, E, sv + L.X + (size_t)r0 * E, E, sv + L.Vcat, E,
                    sv + L.G + (size_t)r0 * 4 * H, 4 * H, sv + L.bUW, 0, n <= 128 ? ws : nullptr,
                    ws_floats, s, ctr, kSplitKCounters));
  }
  return kOk;
}

}  // namespace

size_t seq_saved_floats(const SeqDims& d) { return make_layout(d).total; }
size_t seq_saved_ints(const SeqDims& d) { return make_layout(d).itotal; }

constexpr size_t kSplitKFloats = 32ull * 64 * 2048;  // slabs for the per-step skinny GEMMs (16 MB)
constexpr size_t kSplitKWs = kSplitKFloats - kSplitKCounters;   // slabs | tile counters of the one-launch products (<= 16 rows)

size_t seq_fwd_scratch_floats(const SeqDims& d) { return (size_t)d.B * d.V + 64 + kSplitKFloats; }

size_t seq_bwd_scratch_floats(const SeqDims& d) {
  const size_t N = d.N;
  size_t n = N * 4 * d.H + N * d.H + 2 * (size_t)d.B * d.H + N * d.E + 256 + kSplitKFloats;
  if (d.cell == kCellFactored) n += 2 * N * 4 * d.F;
  return n;
}

namespace {
// one layer of the (possibly stacked) recurrence: its dims (E = its input width), saved buffers and output rows
struct LayerCtx {
  SeqDims d;
  Layout L;
  float* sv;
  int* svi;
  float* hid;
  bool fused_step = false, persist = false;
  int segment = 0;
};

// gate-concatenated weight copies, the fused-step fragment image and the persistent kernel's image
int pack_layer(LayerCtx& c, const SeqWeights& w, const int* batch_sizes, hipStream_t s) {
  const SeqDims& d = c.d;
  const Layout& L = c.L;
  float* sv = c.sv;
  const int E = d.E, F = d.F, H = d.H;
  const GateOrder go = gate_order(d.cell);
  if (d.cell == kCellFactored) {
    // gate-concatenated copies of the 4x6 per-gate tensors: one multi-tensor launch per group
    auto concat4 = [&](const float* const* src, size_t n, float* dst) -> int {
      float* ptrs[4];
      long numel[4];
      for (int g = 0; g < 4; ++g) { ptrs[g] = const_cast<float*>(src[g]); numel[g] = (long)n; }
      return pack_tensors(4, ptrs, numel, dst, 0, 1.f, s);
    };
    RC(concat4(w.Vw, (size_t)F * E, sv + L.Vcat));
    RC(concat4(w.Sw, (size_t)F * F, sv + L.Scat));
    RC(concat4(w.Uw, (size_t)H * F, sv + L.Ucat));
    RC(concat4(w.Ww, (size_t)H * H, sv + L.Wcat));
    RC(concat4(w.Vb, F, sv + L.bV));
    RC(concat4(w.Sb, F, sv + L.bS));
    for (int g = 0; g < 4; ++g) RC(vec_add(w.Ub[g], w.Wb[g], sv + L.bUW + (size_t)g * H, H, s));
  } else {
    RC(copy_d2d(sv + L.Vcat, w.Vw[0], (size_t)4 * H * E, s));
    RC(copy_d2d(sv + L.Wcat, w.Ww[0], (size_t)4 * H * H, s));
    RC(vec_add(w.Vb[0], w.Wb[0], sv + L.bUW, 4 * H, s));
  }
  c.fused_step = H % 16 == 0 && lstm_step_fused_supported(batch_sizes[0], H);
  if (c.fused_step) RC(lstm_pack_wfrag(sv + L.Wcat, sv + L.Wfrag, H, go.gi, go.gf, go.go, go.gg, s));
  // runs of teacher-forced steps go to ONE launch of the persistent kernel (csrc/lstm_persist.hip)
  c.persist = lstm_persist_supported(batch_sizes[0], H);
  if (c.persist) {
    RC(lstm_persist_pack(sv + L.Wcat, sv + L.Wp, go.gi, go.gf, go.go, go.gg, s));
    CAPNET_HIP_CHECK(hipMemsetAsync(c.svi + L.ctl, 0, lstm_persist_ctl_ints() * sizeof(int), s));
  }
  c.segment = 0;
  return kOk;
}

// steps [t, t1) of one layer: step t's gate pre-activations (without the recurrent product) are in G, the following
// steps are teacher forced. One persistent launch where the kernel takes the size, else step by step.
int recur(LayerCtx& c, const std::vector<int>& off, const int* batch_sizes, int t, int t1, float* skws, int* skctr,
          int* err_flag, hipStream_t s) {
  const SeqDims& d = c.d;
  const Layout& L = c.L;
  float* sv = c.sv;
  const int H = d.H;
  const GateOrder go = gate_order(d.cell);
  const bool single_fused = t1 == t + 1 && t > 0 && c.fused_step;   // one free-running step: no weights to keep
  if (c.persist && !single_fused) {
    return lstm_persist_run(sv + L.Wp, sv + L.G, sv + L.Cst, c.hid, off.data(), batch_sizes, t, t1, H, go.gi, go.gf, go.go,
                            go.gg, go.tanh_out, ++c.segment, c.svi + L.ctl, err_flag, s, nullptr);
  }
  for (int u = t; u < t1; ++u) {
    const int b = batch_sizes[u], r0 = off[u];
    if (u > 0) {
      const float* h_prev = c.hid + (size_t)off[u - 1] * H;
      if (c.fused_step) {
        // gates += h_{t-1} . Wcat^T, activations and the c/h update in one launch
        RC(lstm_step_fused(h_prev, sv + L.Wfrag, sv + L.G + (size_t)r0 * 4 * H, 4 * H,
                           sv + L.Cst + (size_t)off[u - 1] * H, sv + L.Cst + (size_t)r0 * H,
                           c.hid + (size_t)r0 * H, b, H, go.gi, go.gf, go.go, go.gg, go.tanh_out, s));
        continue;
      }
      // G[rows] += h_{t-1} . Wcat^T
      RC(sgemm_splitk(false, true, b, 4 * H, H, h_prev, H, sv + L.Wcat, H,
                      sv + L.G + (size_t)r0 * 4 * H, 4 * H, nullptr, 1, skws, kSplitKWs, s, skctr, kSplitKCounters));
    }
    RC(lstm_pointwise_fwd(sv + L.G + (size_t)r0 * 4 * H, 4 * H,
                          u > 0 ? sv + L.Cst + (size_t)off[u - 1] * H : nullptr,
                          sv + L.Cst + (size_t)r0 * H, c.hid + (size_t)r0 * H, b, H, go.gi, go.gf,
                          go.go, go.gg, go.tanh_out, s));
  }
  return kOk;
}

SeqDims upper_dims(const SeqDims& d0) {
  SeqDims d = d0;
  d.E = d0.H;               // a layer above the first reads the hidden state of the layer below
  d.has_features = 0;
  return d;
}
}  // namespace

SeqDims seq_upper_dims(const SeqDims& d0) { return upper_dims(d0); }

// nlayers stacked cells (capnet.stacked: SURVEY App. A-1's semantics, PERF-ONLY / PARITY UNPINNED -- the reference ignores
// num_layers, stylenet/model.py:37): layer 0 is seq_forward's cell on [feature, dropout(B(w))...]; layer l > 0 is the same
// cell on dropout(hidden of layer l - 1) at the same step; the top layer's hidden feeds C on free-running steps.
// Runs of teacher-forced steps outside, layers inside: a run's rows go up the stack before the next run starts (a
// free-running step's input needs the TOP layer's previous hidden state).
int seq_forward_stacked(const SeqDims& d0, int nlayers, const int* batch_sizes, const unsigned char* tf_mask,
                        const long long* captions, const float* features, const float* emb, const SeqWeights* w,
                        const float* Cw, const float* Cb, float dropout_p, unsigned long long seed, int training,
                        float* const* saved, int* const* saved_i, float* scratch, float* const* hiddens, int* err_flag,
                        hipStream_t s) {
  RC(check_dims(d0, batch_sizes));
  CAPNET_REQUIRE(nlayers >= 1 && nlayers <= 8 && w && saved && saved_i && hiddens, "seq_forward_stacked: bad argument");
  CAPNET_REQUIRE(tf_mask && captions && emb && scratch && err_flag, "seq_forward: null argument");
  CAPNET_REQUIRE(!d0.has_features || features, "seq_forward: features missing");
  CAPNET_REQUIRE(dropout_p >= 0.f && dropout_p < 1.f, "seq_forward: dropout p=%f", dropout_p);
  CAPNET_REQUIRE(nlayers == 1 || d0.cell == kCellFactored, "seq_forward_stacked: stacked layers are the factored cell's");
  const int E = d0.E, N = d0.N, H = d0.H;
  bool any_free = false;
  for (int t = 1; t < d0.steps; ++t) any_free |= !tf_mask[t];
  CAPNET_REQUIRE(!any_free || (Cw && Cb), "seq_forward: output projection needed for free-running steps");
  CAPNET_REQUIRE(d0.steps <= kMaxSteps, "seq_forward: %d steps > %d", d0.steps, kMaxSteps);
  std::vector<int> off(d0.steps + 1, 0);
  for (int t = 0; t < d0.steps; ++t) off[t + 1] = off[t] + batch_sizes[t];

  std::vector<LayerCtx> lay(nlayers);
  for (int l = 0; l < nlayers; ++l) {
    CAPNET_REQUIRE(saved[l] && saved_i[l] && hiddens[l], "seq_forward: null buffer of layer %d", l);
    lay[l].d = l == 0 ? d0 : upper_dims(d0);
    lay[l].L = make_layout(lay[l].d);
    lay[l].sv = saved[l]; lay[l].svi = saved_i[l]; lay[l].hid = hiddens[l];
    // ---- row bookkeeping, built on device from kernel arguments (no copy, no sync)
    SeqMeta m;
    m.N = N; m.steps = d0.steps; m.has_features = d0.has_features;
    for (int t = 0; t <= d0.steps; ++t) m.off[t] = off[t];
    for (int t = 0; t < d0.steps; ++t) m.tf[t] = tf_mask[t] ? 1 : 0;
    RC(build_rows(m, lay[l].svi + lay[l].L.row_sample, lay[l].svi + lay[l].L.row_col, lay[l].svi + lay[l].L.row_token,
                  lay[l].svi + lay[l].L.prev_row, s));
    RC(pack_layer(lay[l], w[l], batch_sizes, s));
  }
  LayerCtx& c0 = lay[0];
  LayerCtx& top = lay[nlayers - 1];

  // ---- layer 0: inputs + input chain for every row whose input is known up front
  CAPNET_HIP_CHECK(hipMemsetAsync(c0.sv + c0.L.X, 0, (size_t)N * E * sizeof(float), s));
  RC(gather_inputs(captions, d0.T, features, emb, E, d0.V, c0.svi + c0.L.row_sample, c0.svi + c0.L.row_col,
                   c0.svi + c0.L.row_token, c0.sv + c0.L.X, E, 0, N, dropout_p, seed, training && dropout_p > 0.f, 0,
                   err_flag, s));
  float* skws = scratch + (size_t)d0.B * d0.V + 64;
  int* skctr = reinterpret_cast<int*>(skws + kSplitKWs);
  CAPNET_HIP_CHECK(hipMemsetAsync(skctr, 0, kSplitKCounters * sizeof(int), s));
  RC(input_chain(c0.d, c0.L, c0.sv, 0, N, skws, kSplitKWs, s, skctr));

  // ---- recurrence, run by run
  for (int t = 0; t < d0.steps;) {
    int t1 = t + 1;
    while (t1 < d0.steps && tf_mask[t1]) ++t1;
    const int b = batch_sizes[t], r0 = off[t], r1 = off[t1];
    if (t > 0 && !tf_mask[t]) {
      // predicted = argmax(C h_{t-1}) of the TOP layer for the b surviving rows; then this step's input chain
      const float* h_prev = top.hid + (size_t)off[t - 1] * H;
      RC(sgemm_splitk(false, true, b, d0.V, H, h_prev, H, Cw, H, scratch, d0.V, Cb, 0, skws, kSplitKWs, s, skctr,
                      kSplitKCounters));
      RC(argmax_rows(scratch, b, d0.V, d0.V, c0.svi + c0.L.row_token + r0, s));
      RC(gather_inputs(captions, d0.T, features, emb, E, d0.V, c0.svi + c0.L.row_sample, c0.svi + c0.L.row_col,
                       c0.svi + c0.L.row_token, c0.sv + c0.L.X, E, r0, r0 + b, dropout_p, seed, 0, 1, err_flag, s));
      RC(input_chain(c0.d, c0.L, c0.sv, r0, r0 + b, skws, kSplitKWs, s, skctr));
    }
    for (int l = 0; l < nlayers; ++l) {
      LayerCtx& c = lay[l];
      if (l > 0) {
        // X_l = dropout(hidden of the layer below) for the run's rows, then its input chain
        RC(rows_dropout(lay[l - 1].hid, c.sv + c.L.X, r0, r1, H, dropout_p, seed, l, training && dropout_p > 0.f, s));
        RC(input_chain(c.d, c.L, c.sv, r0, r1, skws, kSplitKWs, s, skctr));
      }
      RC(recur(c, off, batch_sizes, t, t1, skws, skctr, err_flag, s));
    }
    t = t1;
  }
  return kOk;
}

int seq_forward(const SeqDims& d, const int* batch_sizes, const unsigned char* tf_mask,
                const long long* captions, const float* features, const float* emb,
                const SeqWeights& w, const float* Cw, const float* Cb, float dropout_p,
                unsigned long long seed, int training, float* saved, int* saved_i, float* scratch,
                float* hiddens, int* err_flag, hipStream_t s) {
  CAPNET_REQUIRE(saved && saved_i && hiddens, "seq_forward: null argument");
  float* sv[1] = {saved};
  int* svi[1] = {saved_i};
  float* hid[1] = {hiddens};
  return seq_forward_stacked(d, 1, batch_sizes, tf_mask, captions, features, emb, &w, Cw, Cb, dropout_p, seed, training, sv,
                             svi, scratch, hid, err_flag, s);
}

// layer > 0 (a stacked layer above the first): the input gradient goes to dH_below = d hidden of the layer below (through
// the dropout between the layers) instead of the embedding / feature scatter
static int seq_backward_layer(const SeqDims& d, const int* batch_sizes, const float* dH, const float* hiddens,
                              const float* saved, const int* saved_i, float* scratch, const SeqGrads& g,
                              float dropout_p, unsigned long long seed, int training, int layer, float* dH_below,
                              hipStream_t s) {
  RC(check_dims(d, batch_sizes));
  CAPNET_REQUIRE(dH && hiddens && saved && saved_i && scratch, "seq_backward: null argument");
  CAPNET_REQUIRE(g.dWcat && g.dbUW && g.dVcat && (layer > 0 ? dH_below != nullptr : g.dEmb != nullptr), "seq_backward: null gradient buffer");
  const Layout L = make_layout(d);
  const int E = d.E, F = d.F, H = d.H, N = d.N;
  const GateOrder go = gate_order(d.cell);
  std::vector<int> off(d.steps + 1, 0);
  for (int t = 0; t < d.steps; ++t) off[t + 1] = off[t] + batch_sizes[t];

  size_t o = 0;
  auto take = [&](size_t n) { size_t r = o; o += (n + 3) / 4 * 4; return r; };
  float* dPre = scratch + take((size_t)N * 4 * H);
  float* Hprev = scratch + take((size_t)N * H);
  float* dh_rec = scratch + take((size_t)d.B * H);
  float* dc = scratch + take((size_t)d.B * H);
  float* dX = scratch + take((size_t)N * E);
  float* dA2 = nullptr;
  float* dA1 = nullptr;
  if (d.cell == kCellFactored) {
    dA2 = scratch + take((size_t)N * 4 * F);
    dA1 = scratch + take((size_t)N * 4 * F);
  }
  float* skws = scratch + take(kSplitKFloats);
  const float* sv = saved;
  CAPNET_HIP_CHECK(hipMemsetAsync(dh_rec, 0, (size_t)d.B * H * sizeof(float), s));
  CAPNET_HIP_CHECK(hipMemsetAsync(dc, 0, (size_t)d.B * H * sizeof(float), s));

  int slabs = 0;   // > 0: dh of the following step is still in `skws` as K-chunk slabs
  for (int t = d.steps - 1; t >= 0; --t) {
    const int b = batch_sizes[t], r0 = off[t];
    const int b_next = (t + 1 < d.steps) ? batch_sizes[t + 1] : 0;
    RC(lstm_pointwise_bwd(sv + L.G + (size_t)r0 * 4 * H, 4 * H, sv + L.Cst + (size_t)r0 * H,
                          t > 0 ? sv + L.Cst + (size_t)off[t - 1] * H : nullptr,
                          dH + (size_t)r0 * H, slabs > 0 ? skws : dh_rec, dc, dPre + (size_t)r0 * 4 * H,
                          4 * H, b, b_next, H, go.gi, go.gf, go.go, go.gg, go.tanh_out, s, slabs,
                          (long)b_next * H));
    slabs = 0;
    if (t > 0) {
      // dh_{t-1}[0:b] = dPre_t . Wcat     (rows b..b_{t-1} of step t-1 have no successor);
      // the K-chunk partials stay in slabs and are summed by the next gate kernel
      RC(sgemm_splitk_slabs(false, b, H, 4 * H, dPre + (size_t)r0 * 4 * H, 4 * H, sv + L.Wcat, H, skws,
                            kSplitKFloats, &slabs, s));
      if (slabs == 0)
        RC(sgemm_splitk(false, false, b, H, 4 * H, dPre + (size_t)r0 * 4 * H, 4 * H, sv + L.Wcat, H,
                        dh_rec, H, nullptr, 0, skws, kSplitKFloats, s));
    }
  }
  // recurrent weight gradient over all steps at once: dWcat = dPre^T . h_{t-1}
  RC(gather_rows(hiddens, saved_i + L.prev_row, Hprev, N, H, s));
  RC(sgemm_splitk(true, false, 4 * H, H, N, dPre, 4 * H, Hprev, H, g.dWcat, H, nullptr, 0, skws, kSplitKFloats, s));
  RC(colsum(dPre, 4 * H, N, 4 * H, g.dbUW, 0, s));

  if (d.cell == kCellFactored) {
    CAPNET_REQUIRE(g.dUcat && g.dScat && g.dbS && g.dbV, "seq_backward: null factored gradient buffer");
    // U: dU_g = dPre_g^T . A2_g ; dA2_g = dPre_g . U_g
    RC(sgemm(true, false, H, F, N, dPre, 4 * H, sv + L.A2, 4 * F, g.dUcat, F, nullptr, 0, 4, H, F,
             (long)H * F, 0, 0, s));
    RC(sgemm(false, false, N, F, H, dPre, 4 * H, sv + L.Ucat, F, dA2, 4 * F, nullptr, 0, 4, H,
             (long)H * F, F, 0, 0, s));
    RC(colsum(dA2, 4 * F, N, 4 * F, g.dbS, 0, s));
    // S: dS_g = dA2_g^T . A1_g ; dA1_g = dA2_g . S_g
    RC(sgemm(true, false, F, F, N, dA2, 4 * F, sv + L.A1, 4 * F, g.dScat, F, nullptr, 0, 4, F, F,
             (long)F * F, 0, 0, s));
    RC(sgemm(false, false, N, F, F, dA2, 4 * F, sv + L.Scat, F, dA1, 4 * F, nullptr, 0, 4, F,
             (long)F * F, F, 0, 0, s));
    RC(colsum(dA1, 4 * F, N, 4 * F, g.dbV, 0, s));
    // V: dVcat = dA1^T . X ; dX = dA1 . Vcat
    RC(sgemm(true, false, 4 * F, E, N, dA1, 4 * F, sv + L.X, E, g.dVcat, E, nullptr, 0, 1, 0, 0, 0, 0, 0, s));
    RC(sgemm_splitk(false, false, N, E, 4 * F, dA1, 4 * F, sv + L.Vcat, E, dX, E, nullptr, 0, skws,
                    kSplitKFloats, s));
  } else {
    RC(sgemm(true, false, 4 * H, E, N, dPre, 4 * H, sv + L.X, E, g.dVcat, E, nullptr, 0, 1, 0, 0, 0, 0, 0, s));
    RC(sgemm_splitk(false, false, N, E, 4 * H, dPre, 4 * H, sv + L.Vcat, E, dX, E, nullptr, 0, skws,
                    kSplitKFloats, s));
  }
  if (layer > 0) return rows_dropout(dX, dH_below, 0, N, E, dropout_p, seed, layer, training && dropout_p > 0.f, s);
  CAPNET_HIP_CHECK(hipMemsetAsync(g.dEmb, 0, (size_t)d.V * E * sizeof(float), s));
  if (g.dFeat) CAPNET_HIP_CHECK(hipMemsetAsync(g.dFeat, 0, (size_t)d.B * E * sizeof(float), s));
  // (the split-K slab area is free by now: the scatter's two integer tables over the vocabulary go there)
  RC(scatter_input_grad(dX, E, N, E, saved_i + L.row_sample, saved_i + L.row_col,
                        saved_i + L.row_token, g.dEmb, g.dFeat, d.V, dropout_p, seed,
                        training && dropout_p > 0.f, s, reinterpret_cast<int*>(skws), kSplitKFloats));
  return kOk;
}

int seq_backward(const SeqDims& d, const int* batch_sizes, const float* dH, const float* hiddens,
                 const float* saved, const int* saved_i, float* scratch, const SeqGrads& g,
                 float dropout_p, unsigned long long seed, int training, hipStream_t s) {
  return seq_backward_layer(d, batch_sizes, dH, hiddens, saved, saved_i, scratch, g, dropout_p, seed, training, 0, nullptr, s);
}

// BPTT of seq_forward_stacked, top layer first: a layer's whole backward through time, then its input gradient becomes the
// hidden-state gradient of the layer below (only the top layer's hiddens have consumers outside the stack).
// dH_work: nlayers - 1 buffers [N][H]; grads: one SeqGrads per layer (dEmb / dFeat of layer 0 only).
int seq_backward_stacked(const SeqDims& d0, int nlayers, const int* batch_sizes, const float* dH_top,
                         const float* const* hiddens, const float* const* saved, const int* const* saved_i, float* scratch,
                         float* const* dH_work, const SeqGrads* g, float dropout_p, unsigned long long seed, int training,
                         hipStream_t s) {
  CAPNET_REQUIRE(nlayers >= 1 && nlayers <= 8 && hiddens && saved && saved_i && g && (nlayers == 1 || dH_work),
                 "seq_backward_stacked: bad argument");
  const float* dH = dH_top;
  for (int l = nlayers - 1; l >= 0; --l) {
    const SeqDims d = l == 0 ? d0 : upper_dims(d0);
    float* below = l > 0 ? dH_work[l - 1] : nullptr;
    RC(seq_backward_layer(d, batch_sizes, dH, hiddens[l], saved[l], saved_i[l], scratch, g[l], dropout_p, seed, training, l,
                          below, s));
    dH = below;
  }
  return kOk;
}

}  // namespace capnet
