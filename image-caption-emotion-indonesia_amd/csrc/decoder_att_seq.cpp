// Sequence driver of the attention decoders: DecoderFactoredLSTMAtt.forward
// (stylenet/model_att.py:238-305) and nic DecoderRNNAtt.forward (nic/model_att.py:152-202, the
// same loop around nn.LSTMCell(E + C, H)) and their BPTT, one C call each.
//
// Per step t (rows = batch_sizes[t], previous state h, c -- the initial state is
// init_h/init_c(mean over pixels of the feature map), :185-194,260):
//   alpha, awe = attention(features, h)          (:279-280, Attention.forward :51-70)
//   gate = sigmoid(f_beta(h)); awe = gate * awe  (:283-284)
//   x = teacher forced ? dropout(B(w_t)) : B(argmax(C h))   (:285-288)
//   h, c = factored_lstm_step(cat[x, awe], h, c)             (:290-293, forward_step :196-236)
//   alphas[:rows, t] = alpha                                 (:296)
// MI355X mapping. Every product with h_{t-1} is ONE skinny GEMM per step against the stacked
// weight Wz = [W_i; W_f; W_o; W_c; decoder_att; f_beta] ([4H+A+C] x H): the forward writes
// Z = [gate pre-acts | att2 | f_beta(h)] rows, the backward reads dZ rows and gets dh in one
// product; the weight gradients of all six matrices are one TN GEMM over all packed rows after
// the loop. With cell = kCellLSTM the V/S/U chain is the single product [x | ctx] . weight_ih^T
// (gate order i,f,g,o; h = o tanh(c)) and Wz = [weight_hh; decoder_att; f_beta].
// encoder_att(features) is hoisted out of the time loop (the reference recomputes it
// every step) and its weight gradient is accumulated per sample and reduced by one GEMM.
#include <cstdlib>
#include <vector>

#include "common.h"
#include "kernels.h"

namespace capnet {

namespace {

struct ALayout {
  size_t XA, Zf, A1, A2, Cst, alpha, awe, att1, mean, h0, c0, Vcat, Scat, Ucat, Wz, bV, bS, bz, US, Weff, c1, total;
  size_t row_sample, row_col, row_token, prev_row, itotal;
  int ZW, XW;
};

ALayout make_alayout(const AttDims& d) {
  ALayout L;
  L.ZW = 4 * d.H + d.A + d.C;
  L.XW = d.E + d.C;
  size_t o = 0;
  auto take = [&](size_t n) { size_t r = o; o += (n + 3) / 4 * 4; return r; };
  const size_t N = d.N, F = d.F, H = d.H;
  const bool fac = d.cell == kCellFactored;
  L.XA = take(N * L.XW);
  L.Zf = take(N * L.ZW);
  L.A1 = take(fac ? N * 4 * F : 4);
  L.A2 = take(fac ? N * 4 * F : 4);
  L.Cst = take(N * H);
  L.alpha = take(N * d.P);
  L.awe = take(N * d.C);
  L.att1 = take((size_t)d.B * d.P * d.A);
  L.mean = take((size_t)d.B * d.C);
  L.h0 = take((size_t)d.B * H);
  L.c0 = take((size_t)d.B * H);
  L.Vcat = take(fac ? 4 * F * L.XW : 4 * H * L.XW);   // LSTM: weight_ih [4H][E+C]
  L.Scat = take(fac ? 4 * F * F : 4);
  L.Ucat = take(fac ? 4 * H * F : 4);
  L.Wz = take((size_t)L.ZW * H);
  L.bV = take(4 * F);
  L.bS = take(4 * F);
  L.bz = take(L.ZW);
  // the factored chain as ONE matrix (steps of few rows, see chain_collapsed): U_g S_g [H][F], U_g S_g V_g [4H][E+C], S_g bV_g + bS_g
  L.US = take(fac ? 4 * H * F : 4);
  L.Weff = take(fac ? 4 * H * L.XW : 4);
  L.c1 = take(4 * F);
  L.total = o;
  size_t io = 0;
  auto itake = [&](size_t n) { size_t r = io; io += (n + 3) / 4 * 4; return r; };
  L.row_sample = itake(N);
  L.row_col = itake(N);
  L.row_token = itake(N);
  L.prev_row = itake(N);
  L.itotal = io;
  return L;
}

// Steps of at most 16 rows are a chain of dependent launches of 8-12 us each, whatever they compute (NOTEBOOK 4l): the
// factored input product U_g (S_g (V_g x + bV_g) + bS_g) is three of them per step and three more on the way back. With
// few rows the chain is run as ONE product per step against W_g = U_g S_g V_g (formed once per call: a 512 x 512 x 512
// and a 512 x 512 x 2348 product per gate, ~5 GFLOP, where the steps' own products are 0.05 GFLOP each) and bias
// U_g (S_g bV_g + bS_g); the intermediate rows A1 = V x + bV, A2 = S A1 + bS the weight gradients need, and their
// gradients dA2 = dgates U, dA1 = dA2 S, are formed for ALL rows at once after the backward loop -- the same sums in another
// association (fp32 rounding apart: the fixtures of the reference's own class hold either way). It pays up to the 128 rows
// the K-split step products take (64 / 96 rows: +5.4 % / +4.1 % images per second). 0 = by shape (B <= 128), 1 = always,
// -1 = never.
int g_att_chain_mode = [] {
  const char* e = getenv("CAPNET_ATT_CHAIN");        // "3": three products per step everywhere, "1": one everywhere (A/B)
  return e && e[0] == '3' ? -1 : (e && e[0] == '1' ? 1 : 0);
}();
bool chain_collapsed(const AttDims& d) {
  if (d.cell != kCellFactored || g_att_chain_mode < 0) return false;
  return g_att_chain_mode > 0 || d.B <= 128;
}

// The K-chunk partials of a step product, left for its consumer to sum: slab[k][M][N], k < *n (0: the shape qualifies
// for neither kernel and the caller takes the summed form).
int product_slabs(bool tb, int M, int N, int K, const float* A, long lda, const float* B, long ldb, float* ws, size_t ws_floats,
                  int* n, hipStream_t s) {
  const int rc = sgemm_rows16_slabs(tb, M, N, K, A, lda, B, ldb, ws, ws_floats, n, s);
  if (rc || *n) return rc;
  return sgemm_splitk_slabs(tb, M, N, K, A, lda, B, ldb, ws, ws_floats, n, s);
}

constexpr size_t kAttSplitKFloats = 32ull * 64 * 4608;
constexpr size_t kAttSplitKWs = kAttSplitKFloats - kSplitKCounters;   // slabs | tile counters

int check(const AttDims& d, const int* bs) {
  CAPNET_REQUIRE(d.cell == kCellFactored || d.cell == kCellLSTM, "att decoder: unknown cell %d", d.cell);
  CAPNET_REQUIRE(d.B > 0 && d.T > 0 && d.steps > 0 && d.N > 0 && d.E > 0 && d.F > 0 && d.H > 0 &&
                     d.V > 0 && d.A > 0 && d.P > 0 && d.C > 0,
                 "att decoder: bad dims");
  CAPNET_REQUIRE(d.E % 4 == 0 && d.H % 4 == 0 && d.A % 4 == 0 && d.C % 512 == 0 && d.F % 4 == 0,
                 "att decoder: E, H, A, F must be multiples of 4 and the feature size of 512 "
                 "(E=%d H=%d A=%d F=%d C=%d)", d.E, d.H, d.A, d.F, d.C);
  CAPNET_REQUIRE(bs != nullptr && d.steps <= kMaxSteps && d.steps <= d.T, "att decoder: steps");
  long n = 0;
  int prev = d.B;
  CAPNET_REQUIRE(bs[0] == d.B, "att decoder: batch_sizes[0] must equal the batch");
  for (int t = 0; t < d.steps; ++t) {
    CAPNET_REQUIRE(bs[t] > 0 && bs[t] <= prev, "att decoder: batch_sizes must be non-increasing");
    prev = bs[t];
    n += bs[t];
  }
  CAPNET_REQUIRE(n == d.N, "att decoder: sum(batch_sizes) != N");
  return kOk;
}

#define RC(x) do { int _rc = (x); if (_rc) return _rc; } while (0)

struct GateOrderA { int gi, gf, go, gg, tanh_out; };  // column block of each gate role

}  // namespace

int att_set_chain_mode(int mode) {
  const int old = g_att_chain_mode;
  g_att_chain_mode = mode < 0 ? -1 : (mode > 0 ? 1 : 0);
  return old;
}

size_t att_saved_floats(const AttDims& d) { return make_alayout(d).total; }
size_t att_saved_ints(const AttDims& d) { return make_alayout(d).itotal; }
size_t att_fwd_scratch_floats(const AttDims& d) {
  return (size_t)d.B * d.V + 64 + kAttSplitKFloats + (size_t)d.B * d.P + 64;
}
size_t att_bwd_scratch_floats(const AttDims& d) {
  const ALayout L = make_alayout(d);
  const size_t N = d.N;
  return N * L.ZW + 4 * (d.cell == kCellFactored ? N * 4 * d.F : 8) + N * L.XW + N * d.H + 2 * (size_t)d.B * d.H +
         (size_t)d.B * (d.C / 256) * d.P + (size_t)d.B * d.P * d.A + N * d.A + N + N * d.P + 4096 +
         kAttSplitKFloats + (d.B <= 128 ? (size_t)cdiv(L.ZW, 64) * d.B * d.H : 4) + 4;
}

int att_seq_forward(const AttDims& d, const int* bs, const unsigned char* tf,
                    const long long* captions, const float* feat, const float* emb,
                    const AttWeights& w, const float* Cw, const float* Cb, float dropout_p,
                    unsigned long long seed, int training, float* saved, int* saved_i,
                    float* scratch, float* hiddens, float* alphas_bt, int* err_flag,
                    hipStream_t s) {
  RC(check(d, bs));
  CAPNET_REQUIRE(tf && captions && feat && emb && Cw && Cb && saved && saved_i && scratch &&
                     hiddens && alphas_bt && err_flag,
                 "att_seq_forward: null argument");
  CAPNET_REQUIRE(dropout_p >= 0.f && dropout_p < 1.f, "att_seq_forward: dropout p");
  const ALayout L = make_alayout(d);
  const int E = d.E, F = d.F, H = d.H, N = d.N, A = d.A, P = d.P, C = d.C, ZW = L.ZW, XW = L.XW;
  float* sv = saved;
  std::vector<int> off(d.steps + 1, 0);
  for (int t = 0; t < d.steps; ++t) off[t + 1] = off[t] + bs[t];
  {
    SeqMeta m;
    m.N = N; m.steps = d.steps; m.has_features = 0;
    for (int t = 0; t <= d.steps; ++t) m.off[t] = off[t];
    for (int t = 0; t < d.steps; ++t) m.tf[t] = tf[t] ? 1 : 0;
    RC(build_rows(m, saved_i + L.row_sample, saved_i + L.row_col, saved_i + L.row_token,
                  saved_i + L.prev_row, s));
  }
  float* skws = scratch + (size_t)d.B * d.V + 64;
  float* escore = skws + kAttSplitKFloats;   // raw attention scores of the current step [b][P]
  // tile counters of the one-launch products for steps of <= 16 rows (gemm_rows16_kernel): the tail of the slab area
  int* skctr = reinterpret_cast<int*>(skws + kAttSplitKWs);
  CAPNET_HIP_CHECK(hipMemsetAsync(skctr, 0, kSplitKCounters * sizeof(int), s));
  // ---- pack weights
  const bool fac = d.cell == kCellFactored;
  const GateOrderA go = fac ? GateOrderA{0, 1, 2, 3, 0} : GateOrderA{0, 1, 3, 2, 1};
  {
    CopyTable ct;      // one launch for the whole packing
    if (fac) {
      for (int g = 0; g < 4; ++g) {
        ct.add(sv + L.Vcat + (size_t)g * F * XW, w.Vw[g], (size_t)F * XW);
        ct.add(sv + L.Scat + (size_t)g * F * F, w.Sw[g], (size_t)F * F);
        ct.add(sv + L.Ucat + (size_t)g * H * F, w.Uw[g], (size_t)H * F);
        ct.add(sv + L.Wz + (size_t)g * H * H, w.Ww[g], (size_t)H * H);
        ct.add(sv + L.bV + (size_t)g * F, w.Vb[g], F);
        ct.add(sv + L.bS + (size_t)g * F, w.Sb[g], F);
        ct.add(sv + L.bz + (size_t)g * H, w.Ub[g], H, w.Wb[g]);
      }
    } else {
      ct.add(sv + L.Vcat, w.Vw[0], (size_t)4 * H * XW);          // weight_ih
      ct.add(sv + L.Wz, w.Ww[0], (size_t)4 * H * H);             // weight_hh
      ct.add(sv + L.bz, w.Vb[0], 4 * H, w.Wb[0]);                // bias_ih + bias_hh
    }
    ct.add(sv + L.Wz + (size_t)4 * H * H, w.dec_att_w, (size_t)A * H);
    ct.add(sv + L.Wz + (size_t)(4 * H + A) * H, w.f_beta_w, (size_t)C * H);
    ct.add(sv + L.bz + 4 * H, w.dec_att_b, A);
    ct.add(sv + L.bz + 4 * H + A, w.f_beta_b, C);
    RC(multi_copy(ct, s));
  }
  const bool one_product = chain_collapsed(d);
  if (one_product) {
    // US_g = U_g S_g;  Weff_g = US_g V_g;  bz[gate g] += U_g (S_g bV_g + bS_g)
    RC(sgemm(false, false, H, F, F, sv + L.Ucat, F, sv + L.Scat, F, sv + L.US, F, nullptr, 0, 4, (long)H * F, (long)F * F,
             (long)H * F, 0, 0, s));
    RC(sgemm(false, false, H, XW, F, sv + L.US, F, sv + L.Vcat, XW, sv + L.Weff, XW, nullptr, 0, 4, (long)H * F,
             (long)F * XW, (long)H * XW, 0, 0, s));
    RC(sgemm_splitk_batched(false, true, 1, F, F, sv + L.bV, F, sv + L.Scat, F, sv + L.c1, F, sv + L.bS, 0, 4, F, (long)F * F, F, F,
                            skws, kAttSplitKWs, s, skctr, kSplitKCounters));
    RC(sgemm_splitk_batched(false, true, 1, H, F, sv + L.c1, F, sv + L.Ucat, F, sv + L.bz, H, nullptr, 1, 4, F, (long)H * F, H, 0,
                            skws, kAttSplitKWs, s, skctr, kSplitKCounters));
  }

  // ---- time-invariant parts
  RC(global_avgpool(feat, sv + L.mean, d.B, P, C, s));
  // (B x 512 x 2048: a handful of 64 x 64 tiles walking the whole K took 80 us each at 12 rows; K-split: 9)
  RC(sgemm_splitk(false, true, d.B, H, C, sv + L.mean, C, w.init_h_w, C, sv + L.h0, H, w.init_h_b, 0, skws, kAttSplitKWs, s,
                  skctr, kSplitKCounters));
  RC(sgemm_splitk(false, true, d.B, H, C, sv + L.mean, C, w.init_c_w, C, sv + L.c0, H, w.init_c_b, 0, skws, kAttSplitKWs, s,
                  skctr, kSplitKCounters));
  // (through the K-split entry: with few images its 128 x 128 tiles are few and the product is cut over the chip)
  RC(sgemm_splitk(false, true, d.B * P, A, C, feat, C, w.enc_att_w, C, sv + L.att1, A, w.enc_att_b, 0, skws, kAttSplitKWs, s,
                  skctr, kSplitKCounters));
  CAPNET_HIP_CHECK(hipMemsetAsync(sv + L.XA, 0, (size_t)N * XW * sizeof(float), s));
  CAPNET_HIP_CHECK(hipMemsetAsync(alphas_bt, 0, (size_t)d.B * d.steps * P * sizeof(float), s));
  RC(gather_inputs(captions, d.T, nullptr, emb, E, d.V, saved_i + L.row_sample, saved_i + L.row_col,
                   saved_i + L.row_token, sv + L.XA, XW, 0, N, dropout_p, seed,
                   training && dropout_p > 0.f, 0, err_flag, s));

  for (int t = 0; t < d.steps; ++t) {
    const int b = bs[t], r0 = off[t];
    const float* hprev = t > 0 ? hiddens + (size_t)off[t - 1] * H : sv + L.h0;
    const float* cprev = t > 0 ? sv + L.Cst + (size_t)off[t - 1] * H : sv + L.c0;
    float* Z = sv + L.Zf + (size_t)r0 * ZW;
    // Z = h . Wz^T + bz  ->  [recurrent gate pre-acts | att2 | f_beta(h)]
    RC(sgemm_splitk(false, true, b, ZW, H, hprev, H, sv + L.Wz, H, Z, ZW, sv + L.bz, 0, skws, kAttSplitKWs, s, skctr, kSplitKCounters));
    RC(att_step_fwd(sv + L.att1, feat, Z + 4 * H, Z + 4 * H + A, ZW, w.full_att_w, w.full_att_b, b, P,
                    A, C, sv + L.alpha + (size_t)r0 * P, alphas_bt, d.steps, t,
                    sv + L.awe + (size_t)r0 * C, sv + L.XA + (size_t)r0 * XW + E, XW, escore, s));
    if (t > 0 && !tf[t]) {
      RC(sgemm_splitk(false, true, b, d.V, H, hprev, H, Cw, H, scratch, d.V, Cb, 0, skws, kAttSplitKWs, s, skctr, kSplitKCounters));
      RC(argmax_rows(scratch, b, d.V, d.V, saved_i + L.row_token + r0, s));
      RC(gather_inputs(captions, d.T, nullptr, emb, E, d.V, saved_i + L.row_sample,
                       saved_i + L.row_col, saved_i + L.row_token, sv + L.XA, XW, r0, r0 + b,
                       dropout_p, seed, 0, 1, err_flag, s));
    }
    int x_slabs = 0;      // the input product's K-chunk partials, summed by the gate kernel (no hand-off inside the product's launch)
    // one product per step: [x | gated context] . Wx^T with Wx = U S V (the collapsed chain) or nn.LSTMCell's weight_ih
    const float* Wx = one_product ? sv + L.Weff : (fac ? nullptr : sv + L.Vcat);
    if (Wx) {
      RC(product_slabs(true, b, 4 * H, XW, sv + L.XA + (size_t)r0 * XW, XW, Wx, XW, skws, kAttSplitKWs, &x_slabs, s));
      if (!x_slabs)
        RC(sgemm_splitk(false, true, b, 4 * H, XW, sv + L.XA + (size_t)r0 * XW, XW, Wx, XW, Z, ZW, nullptr, 1, skws,
                        kAttSplitKWs, s, skctr, kSplitKCounters));
    } else {
      // factored chain on [x | gated context]
      RC(sgemm_splitk(false, true, b, 4 * F, XW, sv + L.XA + (size_t)r0 * XW, XW, sv + L.Vcat, XW,
                      sv + L.A1 + (size_t)r0 * 4 * F, 4 * F, sv + L.bV, 0, skws, kAttSplitKWs, s, skctr, kSplitKCounters));
      RC(sgemm_splitk_batched(false, true, b, F, F, sv + L.A1 + (size_t)r0 * 4 * F, 4 * F, sv + L.Scat,
                              F, sv + L.A2 + (size_t)r0 * 4 * F, 4 * F, sv + L.bS, 0, 4, F,
                              (long)F * F, F, F, skws, kAttSplitKWs, s, skctr, kSplitKCounters));
      RC(sgemm_splitk_batched(false, true, b, H, F, sv + L.A2 + (size_t)r0 * 4 * F, 4 * F, sv + L.Ucat,
                              F, Z, ZW, nullptr, 1, 4, F, (long)H * F, H, 0, skws, kAttSplitKWs, s, skctr, kSplitKCounters));
    }
    RC(lstm_pointwise_fwd(Z, ZW, cprev, sv + L.Cst + (size_t)r0 * H, hiddens + (size_t)r0 * H, b, H,
                          go.gi, go.gf, go.go, go.gg, go.tanh_out, s, x_slabs ? skws : nullptr, x_slabs));
  }
  return kOk;
}

int att_seq_backward(const AttDims& d, const int* bs, const float* dH, const float* dalphas_bt,
                     const float* hiddens, const float* feat, const AttWeights& w,
                     const float* saved, const int* saved_i, float* scratch, const AttGrads& g,
                     float dropout_p, unsigned long long seed, int training, hipStream_t s) {
  RC(check(d, bs));
  CAPNET_REQUIRE(dH && hiddens && feat && saved && saved_i && scratch, "att_seq_backward: null argument");
  const bool fac = d.cell == kCellFactored;
  const GateOrderA go = fac ? GateOrderA{0, 1, 2, 3, 0} : GateOrderA{0, 1, 3, 2, 1};
  CAPNET_REQUIRE(g.dVcat && g.dWz && g.dbz && g.dWe && g.dbe && g.dwf && g.dbf && g.dWih && g.dbih &&
                     g.dWic && g.dbic && g.dEmb && (!fac || (g.dbV && g.dScat && g.dbS && g.dUcat)),
                 "att_seq_backward: null gradient buffer");
  const ALayout L = make_alayout(d);
  const int E = d.E, F = d.F, H = d.H, N = d.N, A = d.A, P = d.P, C = d.C, ZW = L.ZW, XW = L.XW;
  const float* sv = saved;
  std::vector<int> off(d.steps + 1, 0);
  for (int t = 0; t < d.steps; ++t) off[t + 1] = off[t] + bs[t];
  size_t o = 0;
  auto take = [&](size_t n) { size_t r = o; o += (n + 3) / 4 * 4; return r; };
  float* Zb = scratch + take((size_t)N * ZW);
  float* dA2 = scratch + take(fac ? (size_t)N * 4 * F : 4);
  float* dA1 = scratch + take(fac ? (size_t)N * 4 * F : 4);
  const bool one_product = chain_collapsed(d);
  float* A1c = scratch + take(one_product ? (size_t)N * 4 * F : 4);     // the chain's intermediate rows, formed here
  float* A2c = scratch + take(one_product ? (size_t)N * 4 * F : 4);
  const float* A1 = one_product ? A1c : sv + L.A1;
  const float* A2 = one_product ? A2c : sv + L.A2;
  float* dXA = scratch + take((size_t)N * XW);
  float* Hprev = scratch + take((size_t)N * H);
  float* dh_rec = scratch + take((size_t)d.B * H);
  float* dc = scratch + take((size_t)d.B * H);
  float* dalpha_part = scratch + take((size_t)d.B * (C / 256) * P);
  float* datt1 = scratch + take((size_t)d.B * P * A);
  float* dwf_rows = scratch + take((size_t)N * A);
  float* dbf_rows = scratch + take((size_t)N);
  float* de_all = scratch + take((size_t)N * P);   // softmax-backward scores of every row
  float* skws = scratch + take(kAttSplitKFloats);
  int* skctr = reinterpret_cast<int*>(skws + kAttSplitKWs);
  CAPNET_HIP_CHECK(hipMemsetAsync(skctr, 0, kSplitKCounters * sizeof(int), s));
  CAPNET_HIP_CHECK(hipMemsetAsync(dh_rec, 0, (size_t)d.B * H * sizeof(float), s));
  CAPNET_HIP_CHECK(hipMemsetAsync(dc, 0, (size_t)d.B * H * sizeof(float), s));

  // dh_{t-1} = dZ_t . Wz has K = 4H + A + C: eighteen 256-k chunks. Its partials stay in the slab area and the gate kernel of
  // step t - 1 -- the next launch -- sums them (<= 16 rows: the hand-off inside the launch was 8 of the product's 12.6 us;
  // up to 128 rows: a splitk_reduce launch less per step)
  const size_t dh_ws_floats = d.B <= 128 ? (size_t)cdiv(ZW, 64) * d.B * H : 4;
  float* dh_slabs_ws = scratch + take(dh_ws_floats);
  int dh_slabs = 0;
  for (int t = d.steps - 1; t >= 0; --t) {
    const int b = bs[t], r0 = off[t];
    const int b_next = (t + 1 < d.steps) ? bs[t + 1] : 0;
    const float* cprev = t > 0 ? sv + L.Cst + (size_t)off[t - 1] * H : sv + L.c0;
    const float* Zf = sv + L.Zf + (size_t)r0 * ZW;
    float* Z = Zb + (size_t)r0 * ZW;
    RC(lstm_pointwise_bwd(Zf, ZW, sv + L.Cst + (size_t)r0 * H, cprev, dH + (size_t)r0 * H, dh_slabs ? dh_slabs_ws : dh_rec, dc,
                          Z, ZW, b, b_next, H, go.gi, go.gf, go.go, go.gg, go.tanh_out, s, dh_slabs, (long)b_next * H));
    int dx_slabs = 0;     // d[x | ctx] as K-chunk partials: the context kernel -- the next launch -- sums them
    const float* Wx = one_product ? sv + L.Weff : (fac ? nullptr : sv + L.Vcat);
    if (Wx) {
      // d[x | ctx] = d gates . Wx
      RC(product_slabs(false, b, XW, 4 * H, Z, ZW, Wx, XW, skws, kAttSplitKWs, &dx_slabs, s));
      if (!dx_slabs)
        RC(sgemm_splitk(false, false, b, XW, 4 * H, Z, ZW, Wx, XW, dXA + (size_t)r0 * XW, XW,
                        nullptr, 0, skws, kAttSplitKWs, s, skctr, kSplitKCounters));
    } else {
      RC(sgemm_splitk_batched(false, false, b, F, H, Z, ZW, sv + L.Ucat, F, dA2 + (size_t)r0 * 4 * F,
                              4 * F, nullptr, 0, 4, H, (long)H * F, F, 0, skws, kAttSplitKWs, s, skctr, kSplitKCounters));
      RC(sgemm_splitk_batched(false, false, b, F, F, dA2 + (size_t)r0 * 4 * F, 4 * F, sv + L.Scat, F,
                              dA1 + (size_t)r0 * 4 * F, 4 * F, nullptr, 0, 4, F, (long)F * F, F, 0,
                              skws, kAttSplitKWs, s, skctr, kSplitKCounters));
      RC(sgemm_splitk(false, false, b, XW, 4 * F, dA1 + (size_t)r0 * 4 * F, 4 * F, sv + L.Vcat, XW,
                      dXA + (size_t)r0 * XW, XW, nullptr, 0, skws, kAttSplitKWs, s, skctr, kSplitKCounters));
    }
    RC(att_step_bwd(sv + L.att1, feat, Zf + 4 * H, ZW, Zf + 4 * H + A, ZW, sv + L.awe + (size_t)r0 * C,
                    sv + L.alpha + (size_t)r0 * P, w.full_att_w, dXA + (size_t)r0 * XW + E, XW,
                    dalphas_bt, d.steps, t, b, P, A, C, dalpha_part, Z + 4 * H + A, Z + 4 * H, ZW,
                    de_all + (size_t)r0 * P, dwf_rows + (size_t)r0 * A, dbf_rows + r0, s, dx_slabs ? skws : nullptr, dx_slabs, E));
    // dh_{t-1} (or dh0) = dZ . Wz
    dh_slabs = 0;
    if (d.B <= 128) RC(product_slabs(false, b, H, ZW, Z, ZW, sv + L.Wz, H, dh_slabs_ws, dh_ws_floats, &dh_slabs, s));
    if (!dh_slabs)
      RC(sgemm_splitk(false, false, b, H, ZW, Z, ZW, sv + L.Wz, H, dh_rec, H, nullptr, 0, skws, kAttSplitKWs, s, skctr, kSplitKCounters));
  }
  if (dh_slabs) RC(reduce_slabs(dh_slabs_ws, dh_slabs, d.B, H, dh_rec, H, nullptr, 0, s));      // dh0: all B rows are alive at t = 0
  // ---- weight gradients over all rows at once
  RC(gather_prev_rows(hiddens, saved_i + L.prev_row, sv + L.h0, saved_i + L.row_sample, Hprev, N, H, s));
  RC(sgemm_splitk(true, false, ZW, H, N, Zb, ZW, Hprev, H, g.dWz, H, nullptr, 0, skws, kAttSplitKWs, s));
  RC(colsum(Zb, ZW, N, ZW, g.dbz, 0, s));
  if (one_product) {
    // all rows at once: A1 = XA V^T + bV, A2_g = A1_g S_g^T + bS_g;  dA2_g = dgates_g U_g, dA1_g = dA2_g S_g
    RC(sgemm_splitk(false, true, N, 4 * F, XW, sv + L.XA, XW, sv + L.Vcat, XW, A1c, 4 * F, sv + L.bV, 0, skws, kAttSplitKWs, s,
                    skctr, kSplitKCounters));
    RC(sgemm(false, true, N, F, F, A1c, 4 * F, sv + L.Scat, F, A2c, 4 * F, sv + L.bS, 0, 4, F, (long)F * F, F, F, 0, s));
    RC(sgemm(false, false, N, F, H, Zb, ZW, sv + L.Ucat, F, dA2, 4 * F, nullptr, 0, 4, H, (long)H * F, F, 0, 0, s));
    RC(sgemm(false, false, N, F, F, dA2, 4 * F, sv + L.Scat, F, dA1, 4 * F, nullptr, 0, 4, F, (long)F * F, F, 0, 0, s));
  }
  if (fac) {
    RC(sgemm(true, false, H, F, N, Zb, ZW, A2, 4 * F, g.dUcat, F, nullptr, 0, 4, H, F,
             (long)H * F, 0, 0, s));
    RC(colsum(dA2, 4 * F, N, 4 * F, g.dbS, 0, s));
    RC(sgemm(true, false, F, F, N, dA2, 4 * F, A1, 4 * F, g.dScat, F, nullptr, 0, 4, F, F,
             (long)F * F, 0, 0, s));
    RC(colsum(dA1, 4 * F, N, 4 * F, g.dbV, 0, s));
    RC(sgemm(true, false, 4 * F, XW, N, dA1, 4 * F, sv + L.XA, XW, g.dVcat, XW, nullptr, 0, 1, 0, 0, 0,
             0, 0, s));
  } else {
    // d weight_ih [4H][E+C] = d gates^T . [x | ctx]   (d bias_ih = d bias_hh = dbz[0:4H])
    RC(sgemm(true, false, 4 * H, XW, N, Zb, ZW, sv + L.XA, XW, g.dVcat, XW, nullptr, 0, 1, 0, 0, 0,
             0, 0, s));
  }
  RC(colsum(dwf_rows, A, N, A, g.dwf, 0, s));
  RC(colsum(dbf_rows, 1, N, 1, g.dbf, 0, s));
  // encoder_att: d att1 summed per sample over its steps in one pass over att1
  RC(att_datt1(sv + L.att1, sv + L.Zf + 4 * H, ZW, de_all, w.full_att_w, off.data(), d.steps, d.B, P, A,
               datt1, s));
  RC(sgemm_splitk(true, false, A, C, d.B * P, datt1, A, feat, C, g.dWe, C, nullptr, 0, skws, kAttSplitKWs, s));
  RC(colsum(datt1, A, d.B * P, A, g.dbe, 0, s, skws, kAttSplitKWs));
  // init_h / init_c: dh0 = dh_rec, dc0 = dc (all B rows are alive at t = 0)
  RC(sgemm(true, false, H, C, d.B, dh_rec, H, sv + L.mean, C, g.dWih, C, nullptr, 0, 1, 0, 0, 0, 0, 0, s));
  RC(colsum(dh_rec, H, d.B, H, g.dbih, 0, s));
  RC(sgemm(true, false, H, C, d.B, dc, H, sv + L.mean, C, g.dWic, C, nullptr, 0, 1, 0, 0, 0, 0, 0, s));
  RC(colsum(dc, H, d.B, H, g.dbic, 0, s));
  CAPNET_HIP_CHECK(hipMemsetAsync(g.dEmb, 0, (size_t)d.V * E * sizeof(float), s));
  RC(scatter_input_grad(dXA, XW, N, E, saved_i + L.row_sample, saved_i + L.row_col,
                        saved_i + L.row_token, g.dEmb, nullptr, d.V, dropout_p, seed,
                        training && dropout_p > 0.f, s, reinterpret_cast<int*>(skws), kAttSplitKWs));
  return kOk;
}

}  // namespace capnet
