// ResNet-152 trunk plan + forward driver (the frozen image encoder of the captioning step).
//
// Restates torchvision 0.2.2 `resnet152` children[:-1] / [:-2] as called from
// stylenet/model.py:15-18,24 and stylenet/model_att.py:15-18,24:
//   conv1 7x7/2 p3 -> BN -> ReLU -> MaxPool 3x3/2 p1 -> layer1..4 = Bottleneck x [3,8,36,3]
//   (planes 64/128/256/512, expansion 4, 1x1 -> 3x3 (stride here) -> 1x1, 1x1(stride)+BN
//   downsample on the first block of each layer) -> AvgPool 7.
// The reference runs it under torch.no_grad() but in train mode, so every BN uses batch
// statistics and updates its running stats; nothing is saved for backward.
//
// Data flow per bottleneck (NHWC fp32, each conv output written once / read once):
//   X --conv1--> Y1 raw (+stats) --[bn1+relu on load] conv2--> Y2 raw (+stats)
//     --[bn2+relu on load] conv3--> Y3 raw (+stats);  [X --convD--> D raw (+stats)]
//   OUT = relu(bn3(Y3) + (bnD(D) | X))         (bn_add_relu)
#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <cstdio>
#include <utility>
#include <vector>

#include "common.h"
#include "kernels.h"

namespace capnet {

struct TrunkConv {
  int Cin, Cout, k, stride, pad;
  int H, W;    // input spatial size
  int OH, OW;  // output spatial size
  int Kw;      // packed K extent (rows of the K-major image / row stride of the row-major one)
  bool kmajor; // weights packed [Kw][Cout] for conv_f32_v2 (else [Cout][Kw] for conv_f32)
  bool h3;     // Cin % 64 == 0, 1x1 or 3x3: three f16 MFMA products of 2-way split operands (conv_f16x3.hip and its siblings)
  int tile_n;  // tile width the split-f16 weight image was laid out for
  bool stem_h3 = false;   // the 7x7 / 2 stem on the same arithmetic, NCHW image in (conv_stem.hip)
  // fused_block.hip: conv3 of a block whose statistics come from the Gram matrix of its input and whose product is formed
  // inside the next block's conv1 launch (fused3), and that conv1 (fused1). Their weight images are fused_block_pack's.
  bool fused3 = false, fused1 = false;
};

struct Trunk {
  int B, H, W;
  std::vector<TrunkConv> convs;  // torchvision parameter order
  // workspace layout (float offsets)
  size_t off_x[2], off_y1, off_y2, off_y3, off_d, off_part, off_slab, off_gram, off_ss, total_floats;
  std::vector<size_t> ss_off;  // per conv: offset of [scale | shift] (2*Cout floats)
  std::vector<size_t> bs_off;  // per conv: offset of [batch mean | unbiased batch var] (2*Cout floats)
  int final_side;
  // optional per-convolution hipEvent timing (bench.py roofline): pairs recorded on the launch
  // stream around every conv kernel while enabled, summed by trunk_collect_timing
  // K-sliced tail balancing of the convs (conv_f32_v2.hip): pays when one pass has the chip to
  // itself; with several passes in flight (TrunkPipeline) the other passes fill the idle CUs of a
  // partially filled last round and the slabs + fix-up launches only cost (measured +2 % images/s
  // without them), so the pipeline turns it off
  bool tail_balance = true;
  // A/B and fallback switches, read from the environment when the plan is made (tests/test_encoder_gpu.py runs a
  // trunk under each): CAPNET_NO_P3=1 the stride-1 3x3 convolutions on the implicit-GEMM kernel instead of the patch
  // kernel; CAPNET_NO_TAIL_FUSION=1 every block tail as its own bn_add_relu launch
  // CAPNET_AREG=1 conv3 of stages 1-3 on the A-in-registers kernel (conv1x1_areg.hip) instead of the tiled one: built for
  // VERDICT r2 #4, measured slower in the pipelined step, kept as an option
  bool use_patch = true, fuse_tails = true, use_areg = false;
  // CAPNET_NO_FUSED_BLOCK=1: conv3 materialises y3 everywhere (round 3's data flow); CAPNET_FUSED_STAGES=<mask> (bit L =
  // stage L + 1) picks the stages whose inner block boundaries run on fused_block.hip (default: see trunk_create)
  int fused_stages = 0;
  bool timing = false;
  int timing_every = 1;   // ... on every N-th pass (an event pair is a bubble in the stream: 310 per pass cost 2.5 % images/s)
  long pass_no = 0;
  bool timing_now = false;
  bool timing_once = false;   // trunk_time_next_pass: the next pass is bracketed whatever `timing` says
  std::vector<hipEvent_t> ev_pool;    // events are created once and handed out again after every collect
  size_t ev_next = 0;
  std::vector<hipEvent_t> ev;
  double timed_flops = 0;
};

static int round_up(int v, int m) { return (v + m - 1) / m * m; }

int trunk_create(int B, int H, int W, Trunk** out) {
  CAPNET_REQUIRE(out != nullptr, "trunk_create: null out");
  CAPNET_REQUIRE(B > 0 && H >= 32 && W >= 32 && H % 32 == 0 && W % 32 == 0,
                 "trunk_create: batch %d image %dx%d (sides must be multiples of 32)", B, H, W);
  Trunk* t = new Trunk();
  t->B = B; t->H = H; t->W = W;
  auto env_on = [](const char* name) { const char* e = getenv(name); return e && e[0] == '1'; };
  // CAPNET_NO_H3=1: the whole trunk on the f32-MFMA kernels (conv_f32_v2.hip / conv_f32.hip), the family that also
  // serves shapes the split-f16 kernels do not take
  const bool use_h3 = !env_on("CAPNET_NO_H3");
  t->use_patch = !env_on("CAPNET_NO_P3");
  t->fuse_tails = !env_on("CAPNET_NO_TAIL_FUSION");
  t->use_areg = env_on("CAPNET_AREG");
  // conv1 of every block but the first rides with the previous block's tail (conv1x1_tail_kernel), the stride-1 3x3
  // convolutions run on the patch kernel; where Cout is a multiple of 256 (stages 3 and 4) both take ONE 128 x 256 tile
  // per row tile -- their own weight image, which conv_f16x3_kernel does not read: planned only while such a convolution
  // is certain to run on those kernels (not with CAPNET_NO_TAIL_FUSION=1 / CAPNET_NO_P3=1 or the folded inference trunk,
  // CAPNET_EVAL_FOLDED=1). The tail kernel's wide tile is the default (+1.5 % images/s although 30 % slower alone:
  // 98 workgroups leave 158 CUs to the other passes; CAPNET_NO_WIDE_TAIL=1 for the 128-column tiles), the patch
  // kernel's is not (86 vs 61 us alone, -0.9 % in the step; CAPNET_WIDE_P3=1 turns it on) -- DESIGN 4j
  const bool no_folded = !env_on("CAPNET_EVAL_FOLDED");
  const bool wide_tails = t->fuse_tails && no_folded && !env_on("CAPNET_NO_WIDE_TAIL");
  const bool wide_p3 = t->use_patch && no_folded && env_on("CAPNET_WIDE_P3");
  // the wide tail tile (one workgroup per 128 rows) only where that still makes a quarter of the chip's workgroups: at
  // batch 64 stage 3 (98 row tiles) takes it, stage 4 (25) keeps 4 x 25 narrow ones (+0.5 % images/s); a 12-image batch none
  const long wide_min_tiles = 64;
  auto add = [&](int cin, int cout, int k, int stride, int pad, int h, int w, bool activated_input = false,
                 bool tail_conv1 = false) {
    TrunkConv c;
    c.Cin = cin; c.Cout = cout; c.k = k; c.stride = stride; c.pad = pad; c.H = h; c.W = w;
    c.OH = (h + 2 * pad - k) / stride + 1;
    c.OW = (w + 2 * pad - k) / stride + 1;
    c.Kw = round_up(k * k * cin, 16);
    c.kmajor = (cin % 16 == 0) && (cout % 64 == 0);
    // (a folded input needs its BatchNorm's scale / shift in LDS: Cin <= 512)
    c.h3 = use_h3 && cin % 64 == 0 && cout % 64 == 0 && ((k == 1 && pad == 0 && (activated_input || cin <= 512)) || (k == 3 && pad == 1 && cin <= 512));
    // ... and the kernels' 32-bit offset arithmetic must cover the whole tensor at this batch size: checked here,
    // on the dense NHWC strides every trunk tensor has, so that a shape they do not take is PLANNED for the f32 kernels
    // (with its K-major weight image) instead of failing at launch time
    if (c.h3) {
      const float* aligned = reinterpret_cast<const float*>(uintptr_t(256));
      c.h3 = conv_f16x3_eligible(aligned, (long)h * w * cin, (long)w * cin, cin, 1, B, h, w, cin, cout, k, stride, pad,
                                 activated_input ? nullptr : aligned, activated_input ? nullptr : aligned);
    }
    c.tile_n = c.h3 ? conv1x1_f16x3_bn((long)B * c.OH * c.OW, cout) : 0;
    if (c.h3 && tail_conv1 && wide_tails && cout % 256 == 0 && ((long)B * h * w + 127) / 128 >= wide_min_tiles &&
        conv1x1_tail_eligible(reinterpret_cast<const float*>(uintptr_t(256)), reinterpret_cast<const float*>(uintptr_t(256)),
                              (long)B * h * w, cin, cout))
      c.tile_n = 256;
    if (c.h3 && wide_p3 && k == 3 && stride == 1 && cout % 256 == 0 &&
        conv3x3_patch_eligible(reinterpret_cast<const float*>(uintptr_t(256)), (long)h * w * cin, (long)w * cin, cin, 1, B, h, w, cin,
                               cout, k, stride, pad, reinterpret_cast<const float*>(uintptr_t(256)),
                               reinterpret_cast<const float*>(uintptr_t(256))))
      c.tile_n = 256;
    t->convs.push_back(c);
    return c;
  };
  TrunkConv stem = add(3, 64, 7, 2, 3, H, W);
  {
    // CAPNET_NO_STEM_H3=1 keeps the stem on the generic f32 gather kernel (A/B runs)
    t->convs[0].stem_h3 = use_h3 && !env_on("CAPNET_NO_STEM_H3") && W % 4 == 0;
  }
  int h = (stem.OH + 2 - 3) / 2 + 1, w = (stem.OW + 2 - 3) / 2 + 1;  // maxpool
  int inplanes = 64;
  const int blocks[4] = {3, 8, 36, 3};
  const int planes[4] = {64, 128, 256, 512};
  {
    const char* fs = getenv("CAPNET_FUSED_STAGES");
    t->fused_stages = (!use_h3 || !t->fuse_tails || !no_folded || env_on("CAPNET_NO_FUSED_BLOCK")) ? 0 : (fs ? atoi(fs) & 7 : 7);
  }
  size_t max_x = (size_t)h * w * 64, max_y1 = 0, max_y2 = 0, max_y3 = 0, max_d = 0, max_gram = 0, max_fpart = 0;
  for (int L = 0; L < 4; ++L) {
    for (int b = 0; b < blocks[L]; ++b) {
      const int stride = (b == 0 && L > 0) ? 2 : 1;
      const int p = planes[L];
      TrunkConv c1 = add(inplanes, p, 1, 1, 0, h, w, true, !(L == 0 && b == 0));
      TrunkConv c2 = add(p, p, 3, stride, 1, h, w);
      TrunkConv c3 = add(p, p * 4, 1, 1, 0, c2.OH, c2.OW);
      // the boundary (b - 1 -> b) inside a stage: the previous block's conv3 (three or four entries back) and this conv1
      if (b > 0 && L < 3 && ((t->fused_stages >> L) & 1) && c1.h3 && fused_block_shape_ok((long)B * h * w, p)) {
        const size_t i1 = t->convs.size() - 3, i3p = i1 - (b == 1 ? 2 : 1);
        if (t->convs[i3p].h3 && t->convs[i3p].k == 1 && t->convs[i3p].Cout == inplanes && t->convs[i3p].Cin == p) {
          t->convs[i1].fused1 = true;
          t->convs[i3p].fused3 = true;
          t->convs[i1].tile_n = 0;
          max_gram = std::max(max_gram, fused_block_stats_floats((long)B * h * w, p));
          max_fpart = std::max(max_fpart, (size_t)fused_block_tiles((long)B * h * w, p) * p);
        }
      }
      max_y1 = std::max(max_y1, (size_t)c1.OH * c1.OW * c1.Cout);
      max_y2 = std::max(max_y2, (size_t)c2.OH * c2.OW * c2.Cout);
      max_y3 = std::max(max_y3, (size_t)c3.OH * c3.OW * c3.Cout);
      if (b == 0) {
        TrunkConv d = add(inplanes, p * 4, 1, stride, 0, h, w, true);
        max_d = std::max(max_d, (size_t)d.OH * d.OW * d.Cout);
      }
      h = c2.OH; w = c2.OW;
      inplanes = p * 4;
      max_x = std::max(max_x, (size_t)h * w * inplanes);
    }
  }
  t->final_side = h;
  CAPNET_REQUIRE(h == w, "trunk_create: only square feature maps are supported");
  // stem raw output shares the Y3 buffer
  max_y3 = std::max(max_y3, (size_t)stem.OH * stem.OW * 64);
  size_t off = 0;
  auto take = [&](size_t n) { size_t o = off; off += (n + 63) / 64 * 64; return o; };
  t->off_x[0] = take(max_x * B);
  t->off_x[1] = take(max_x * B);
  t->off_y1 = take(max_y1 * B);
  t->off_y2 = take(max_y2 * B);
  t->off_y3 = take(max_y3 * B);
  t->off_d = take(max_d * B);
  size_t max_part = 0;
  for (auto& c : t->convs) {
    const long M = (long)B * c.OH * c.OW;
    const int tile = c.kmajor ? conv_v2_auto_tile((int)M, c.Cout, c.Kw) : conv_auto_tile((int)M, c.Cout);
    max_part = std::max(max_part, (size_t)conv_tiles_m((int)M, tile) * c.Cout);
    if (c.h3) max_part = std::max(max_part, (size_t)conv1x1_tiles_m(M) * c.Cout);
    if (c.stem_h3) max_part = std::max(max_part, (size_t)conv_stem_f16x3_part_rows(B, c.H, c.W) * c.Cout);
  }
  max_part = std::max(max_part, max_fpart);
  t->off_part = take(2 * max_part);
  size_t max_slab = 0;
  for (auto& c : t->convs)
    if (c.kmajor) max_slab = std::max(max_slab, conv_v2_slab_floats(B * c.OH * c.OW, c.Cout, c.Kw, 0));
  t->off_slab = take(max_slab + 64);
  t->off_gram = take(max_gram + 64);
  t->off_ss = off;
  for (auto& c : t->convs) t->ss_off.push_back(take(2 * (size_t)c.Cout));
  for (auto& c : t->convs) t->bs_off.push_back(take(2 * (size_t)c.Cout));
  t->total_floats = off;
  *out = t;
  return kOk;
}

void trunk_destroy(Trunk* t) {
  if (!t) return;
  for (hipEvent_t e : t->ev_pool) (void)hipEventDestroy(e);
  delete t;
}

int trunk_set_timing(Trunk* t, int enable) {
  CAPNET_REQUIRE(t != nullptr, "trunk_set_timing: null");
  t->timing = enable != 0;
  t->timing_every = enable > 1 ? enable : 1;
  t->pass_no = 0;
  t->timing_now = false;
  return kOk;
}

// The next pass only (a caller that replays the other passes from hipGraphs, where events cannot ride, launches every
// N-th pass directly and brackets that one).
int trunk_time_next_pass(Trunk* t) {
  CAPNET_REQUIRE(t != nullptr, "trunk_time_next_pass: null");
  t->timing_once = true;
  return kOk;
}

// Synchronises on the recorded events; returns total conv-kernel ms, launches and flops since
// the last collect.
int trunk_set_tail_balance(Trunk* t, int on) {
  CAPNET_REQUIRE(t != nullptr, "trunk_set_tail_balance: null");
  t->tail_balance = on != 0;
  return kOk;
}

int trunk_collect_timing(Trunk* t, double* conv_ms, long* conv_launches, double* conv_flops) {
  CAPNET_REQUIRE(t && conv_ms && conv_launches && conv_flops, "trunk_collect_timing: null");
  // Time during which at least one timed conv launch was running: the union of the [start, end]
  // intervals. With one pass at a time this is the sum of the launch durations; with two passes in
  // flight on different streams (capnet.train.TrunkPipeline) the launches of the two passes
  // overlap and a plain sum would count that time twice.
  std::vector<std::pair<float, float>> iv;
  for (size_t i = 0; i + 1 < t->ev.size(); i += 2) {
    CAPNET_HIP_CHECK(hipEventSynchronize(t->ev[i + 1]));
    float a = 0, b = 0;
    if (i > 0) CAPNET_HIP_CHECK(hipEventElapsedTime(&a, t->ev[0], t->ev[i]));
    CAPNET_HIP_CHECK(hipEventElapsedTime(&b, t->ev[i], t->ev[i + 1]));
    iv.emplace_back(a, a + b);
  }
  std::sort(iv.begin(), iv.end());
  double ms = 0;
  float cur_a = 0, cur_b = 0;
  bool open = false;
  for (auto& p : iv) {
    if (!open) { cur_a = p.first; cur_b = p.second; open = true; continue; }
    if (p.first <= cur_b) { cur_b = std::max(cur_b, p.second); continue; }
    ms += cur_b - cur_a;
    cur_a = p.first; cur_b = p.second;
  }
  if (open) ms += cur_b - cur_a;
  *conv_ms = ms;
  *conv_launches = (long)(t->ev.size() / 2);
  *conv_flops = t->timed_flops;
  t->ev.clear();          // (the events stay in the pool)
  t->ev_next = 0;
  t->timed_flops = 0;
  return kOk;
}
size_t trunk_workspace_bytes(const Trunk* t) { return t->total_floats * sizeof(float); }
int trunk_num_convs(const Trunk* t) { return (int)t->convs.size(); }
int trunk_final_side(const Trunk* t) { return t->final_side; }

int trunk_conv_shape(const Trunk* t, int i, int* cout, int* cin, int* k, int* stride, int* kw) {
  CAPNET_REQUIRE(i >= 0 && i < (int)t->convs.size(), "trunk_conv_shape: index %d", i);
  const TrunkConv& c = t->convs[i];
  *cout = c.Cout; *cin = c.Cin; *k = c.k; *stride = c.stride; *kw = c.Kw;
  return kOk;
}

int trunk_conv_kmajor(const Trunk* t, int i) {
  if (i < 0 || i >= (int)t->convs.size()) return 0;
  return t->convs[i].fused3 ? 7 : t->convs[i].fused1 ? 8 : t->convs[i].stem_h3 ? 6 : t->convs[i].h3 ? 5 : (t->convs[i].kmajor ? 1 : 0);
}

int trunk_conv_tile_n(const Trunk* t, int i) {
  if (i < 0 || i >= (int)t->convs.size()) return 0;
  return t->convs[i].tile_n;
}

double trunk_conv_flops(const Trunk* t, int i) {
  if (i < 0 || i >= (int)t->convs.size()) return 0.0;
  const TrunkConv& c = t->convs[i];
  return 2.0 * t->B * c.OH * c.OW * (double)c.Cout * c.k * c.k * c.Cin;
}

double trunk_flops(const Trunk* t) {
  double f = 0;
  for (auto& c : t->convs)
    f += 2.0 * t->B * c.OH * c.OW * (double)c.Cout * c.k * c.k * c.Cin;
  return f;
}

namespace {
// A pair of timing events for one conv launch. Kernels launched through CAPNET_LAUNCH_TIMED (the split-f16 family) get
// them attached to their dispatch (park = true: no hipEventRecord at all); the others are bracketed by two records.
static int take_events(Trunk* t, hipEvent_t* e0, hipEvent_t* e1) {
  while (t->ev_pool.size() < t->ev_next + 2) {
    hipEvent_t e;
    CAPNET_HIP_CHECK(hipEventCreate(&e));
    t->ev_pool.push_back(e);
  }
  *e0 = t->ev_pool[t->ev_next++];
  *e1 = t->ev_pool[t->ev_next++];
  return kOk;
}
static int timing_begin(Trunk* t, bool attach, hipStream_t s, hipEvent_t* e0, hipEvent_t* e1) {
  const int rc = take_events(t, e0, e1);
  if (rc) return rc;
  if (attach) {
    launch_events().start = *e0;
    launch_events().stop = *e1;
  } else {
    CAPNET_HIP_CHECK(hipEventRecord(*e0, s));
  }
  return kOk;
}
static int timing_end(Trunk* t, bool attach, hipStream_t s, hipEvent_t e0, hipEvent_t e1, double flops) {
  if (attach) {
    CAPNET_REQUIRE(launch_events().start == nullptr, "trunk: a conv launcher did not take its timing events");
  } else {
    CAPNET_HIP_CHECK(hipEventRecord(e1, s));
  }
  t->ev.push_back(e0);
  t->ev.push_back(e1);
  t->timed_flops += flops;
  return kOk;
}

struct Ctx {
  Trunk* t;
  const float* const* w;
  const float* const* gamma;
  const float* const* beta;
  float* const* rmean;
  float* const* rvar;
  int train;
  float momentum, eps;
  float* ws;
  hipStream_t s;
  bool eval_ready = false;   // inference: every BatchNorm's (scale, shift) already sits in the workspace (bn_eval_multi)
  const int* in_exps = nullptr;   // per convolution: power-of-two prescale of its INPUT on the split-f16 kernels (null: none)
  int* err = nullptr;             // device error word: bit 3 = a non-finite value in the trunk
  int in_exp(int i) const { return in_exps ? in_exps[i] : 0; }
  float* scale(int i) const { return ws + t->ss_off[i]; }
  float* shift(int i) const { return ws + t->ss_off[i] + t->convs[i].Cout; }
  float* bmean(int i) const { return ws + t->bs_off[i]; }
  float* bvar(int i) const { return ws + t->bs_off[i] + t->convs[i].Cout; }
};

// A bottleneck block's tail that has not run yet: out = relu(y3 * s1 + t1 + res (* s2 + t2)), [rows][C]
struct BlockTail {
  const float* y3; const float* s1; const float* t1;
  const float* res; const float* s2; const float* t2;
  float* out;
  long rows;
  int C;
};
static int run_tail(const Ctx& c, const BlockTail& t) {
  return bn_add_relu(t.y3, t.s1, t.t1, t.res, t.s2, t.t2, t.out, t.rows, t.C, c.s);
}

// conv i + the (scale, shift) of the BatchNorm that follows it. tail: the previous block's tail is still pending and
// x is its output -- it rides in this launch where the kernel exists (a stride-1 1x1 conv1 on the split-f16 path),
// otherwise it is run first.
int conv_bn(const Ctx& c, int i, const float* x, long sxb, long sxh, long sxw, long sxc,
            const float* in_scale, const float* in_shift, int relu_in, float* y, const BlockTail* tail = nullptr) {
  const TrunkConv& d = c.t->convs[i];
  const long M = (long)c.t->B * d.OH * d.OW;
  CAPNET_REQUIRE(d.tile_n != 256 || d.k == 3 || (tail && (c.train || c.eval_ready)), "trunk: conv %d was planned for the wide tail kernel", i);
  const bool fuse_tail = tail && (c.train || c.eval_ready) && c.t->fuse_tails && d.h3 && d.k == 1 && d.stride == 1 && !in_scale &&
                         x == tail->out && tail->C == d.Cin && tail->rows == M && sxc == 1 && sxw == d.Cin &&
                         sxh == (long)d.W * d.Cin && sxb == (long)d.H * d.W * d.Cin &&
                         conv1x1_tail_eligible(tail->y3, tail->res, M, d.Cin, d.Cout);
  if (tail && !fuse_tail) {
    const int rt = run_tail(c, *tail);
    if (rt) return rt;
  }
  // single pass: the occupancy model picks the tile (64x64 almost everywhere: least quantisation
  // loss on 256 CUs). Several passes sharing the chip (tail_balance off, see above): quantisation
  // is filled by the other passes, so the 128x64 tile (half the B-tile traffic and staging VALU per
  // MFMA) wins on the layers with many output rows -- measured +2 % images/s for M >= 5000
  int tile = d.kmajor ? conv_v2_auto_tile((int)M, d.Cout, d.Kw) : conv_auto_tile((int)M, d.Cout);
  if (d.kmajor && !c.t->tail_balance && M >= 5000) tile = 12864;
  // rows of the statistics partials this conv writes
  const int prows = d.stem_h3 ? conv_stem_f16x3_part_rows(c.t->B, d.H, d.W) : d.h3 ? conv1x1_tiles_m(M) : conv_tiles_m((int)M, tile);
  float* psum = c.ws + c.t->off_part;
  float* psq = psum + (size_t)prows * d.Cout;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  const bool attach = fuse_tail || d.stem_h3 || d.h3;      // one kernel, launched through CAPNET_LAUNCH_TIMED
  if (c.t->timing_now) {
    const int rt = timing_begin(c.t, attach, c.s, &e0, &e1);
    if (rt) return rt;
  }
  int rc;
  if (fuse_tail) {
    rc = conv1x1_fwd_tail(tail->y3, tail->s1, tail->t1, tail->res, tail->s2, tail->t2, tail->out,
                          reinterpret_cast<const unsigned*>(c.w[i]), d.tile_n, y, c.train ? psum : nullptr,
                          c.train ? psq : nullptr, M, d.Cin, d.Cout, c.s, c.in_exp(i), c.err);
  } else if (d.stem_h3) {
    CAPNET_REQUIRE(!in_scale && conv_stem_f16x3_eligible(x, sxb, sxc, sxh, sxw, c.t->B, d.H, d.W, d.Cin, d.Cout, d.k, d.stride, d.pad),
                   "trunk: the stem is planned for the split-f16 kernel but its operands are not eligible");
    rc = conv_stem_fwd_f16x3(x, sxb, sxc, sxh, reinterpret_cast<const unsigned*>(c.w[i]), y, c.train ? psum : nullptr,
                             c.train ? psq : nullptr, c.t->B, d.H, d.W, c.s, c.in_exp(i), c.err);
  } else if (d.h3 && d.k == 3 && c.t->use_patch &&
             conv3x3_patch_eligible(x, sxb, sxh, sxw, sxc, c.t->B, d.H, d.W, d.Cin, d.Cout, d.k, d.stride, d.pad, in_scale, in_shift)) {
    // stride-1 3x3: the tile's input patch staged once instead of once per tap (conv3x3_patch.hip), same weight image
    rc = conv3x3_fwd_patch(x, reinterpret_cast<const unsigned*>(c.w[i]), d.tile_n, y, in_scale, in_shift, relu_in,
                           c.train ? psum : nullptr, c.train ? psq : nullptr, c.t->B, d.H, d.W, d.Cin, d.Cout, c.s,
                           !c.t->tail_balance, c.in_exp(i), c.err);
  } else if (d.h3 && d.k == 1 && d.stride == 1 && c.t->use_areg && sxc == 1 && sxw == d.Cin && sxh == (long)d.W * d.Cin &&
             sxb == (long)d.H * d.W * d.Cin && conv1x1_areg_eligible(x, M, d.Cin, d.Cout, d.tile_n, in_scale, in_shift)) {
    // short K (conv3 of stages 1-3): the A operand folded and split once per 128 rows, resident in registers
    rc = conv1x1_fwd_areg(x, reinterpret_cast<const unsigned*>(c.w[i]), d.tile_n, y, in_scale, in_shift, relu_in,
                          c.train ? psum : nullptr, c.train ? psq : nullptr, M, d.Cin, d.Cout, c.in_exp(i), c.s, c.err);
  } else if (d.h3) {
    CAPNET_REQUIRE(conv_f16x3_eligible(x, sxb, sxh, sxw, sxc, c.t->B, d.H, d.W, d.Cin, d.Cout, d.k, d.stride, d.pad, in_scale, in_shift),
                   "trunk: conv %d planned for the split-f16 kernel but its operands are not eligible", i);
    rc = conv_fwd_f16x3(x, sxb, sxh, sxw, reinterpret_cast<const unsigned*>(c.w[i]), d.tile_n, y, in_scale,
                        in_shift, relu_in, c.train ? psum : nullptr, c.train ? psq : nullptr, c.t->B, d.H,
                        d.W, d.Cin, d.Cout, d.k, d.stride, d.pad, c.s, nullptr, nullptr, nullptr, 0, c.in_exp(i), c.err);
  } else if (d.kmajor) {
    CAPNET_REQUIRE(conv_v2_eligible(x, sxb, sxh, sxw, sxc, c.t->B, d.Cin, d.Cout, in_scale, in_shift),
                   "trunk: conv %d planned for the K-major kernel but its operands are not eligible", i);
    rc = conv2d_fwd_v2(x, sxb, sxh, sxw, c.w[i], d.Kw, y, in_scale, in_shift, relu_in,
                       c.train ? psum : nullptr, c.train ? psq : nullptr, c.t->B, d.H, d.W, d.Cin,
                       d.Cout, d.k, d.k, d.stride, d.pad, tile,
                       c.t->tail_balance ? c.ws + c.t->off_slab : nullptr, c.s);
  } else {
    rc = conv2d_fwd(x, sxb, sxh, sxw, sxc, c.w[i], d.Kw, y, in_scale, in_shift, relu_in,
                    c.train ? psum : nullptr, c.train ? psq : nullptr, c.t->B, d.H, d.W, d.Cin,
                    d.Cout, d.k, d.k, d.stride, d.pad, tile, c.s);
  }
  if (c.t->timing_now) {
    if (rc) launch_events() = LaunchEvents{};
    else {
      const int rt = timing_end(c.t, attach, c.s, e0, e1, 2.0 * (double)M * d.Cout * d.k * d.k * d.Cin);
      if (rt) return rt;
    }
  }
  if (rc) return rc;
  if (c.train == 2)   // running statistics deferred to trunk_update_running
    return bn_finalize(psum, psq, prows, d.Cout, M, c.gamma[i], c.beta[i],
                       nullptr, nullptr, c.momentum, c.eps, c.scale(i), c.shift(i), c.s, c.bmean(i),
                       c.bvar(i), c.err);
  if (c.train)
    return bn_finalize(psum, psq, prows, d.Cout, M, c.gamma[i], c.beta[i],
                       c.rmean[i], c.rvar[i], c.momentum, c.eps, c.scale(i), c.shift(i), c.s, nullptr, nullptr, c.err);
  if (c.eval_ready) return kOk;
  return bn_eval_scale_shift(c.gamma[i], c.beta[i], c.rmean[i], c.rvar[i], c.eps, d.Cout,
                             c.scale(i), c.shift(i), c.s);
}
// ---- a block boundary on fused_block.hip --------------------------------------------------------------------------
// conv3 (i3) of block b is never launched: its BatchNorm's (scale, shift) come from the Gram matrix of its input y2 ...
int fused_stats_bn(const Ctx& c, int i3, const float* y2, const float* s2, const float* t2) {
  const TrunkConv& d = c.t->convs[i3];
  const long M = (long)c.t->B * d.OH * d.OW;
  if (!c.train) return kOk;          // inference: every (scale, shift) is already there (eval_ready)
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (c.t->timing_now) {
    const int rt = timing_begin(c.t, false, c.s, &e0, &e1);
    if (rt) return rt;
  }
  const bool defer = c.train == 2;
  const int rc = fused_block_stats(y2, s2, t2, reinterpret_cast<const unsigned*>(c.w[i3]), M, d.Cin, c.in_exp(i3), c.gamma[i3],
                                   c.beta[i3], defer ? nullptr : c.rmean[i3], defer ? nullptr : c.rvar[i3], c.momentum, c.eps,
                                   c.scale(i3), c.shift(i3), defer ? c.bmean(i3) : nullptr, defer ? c.bvar(i3) : nullptr,
                                   c.ws + c.t->off_gram, c.err, c.s);
  if (rc) return rc;
  // (the statistics' launches are conv time without algorithmic flops: the product itself is timed with the fused launch)
  if (c.t->timing_now) return timing_end(c.t, false, c.s, e0, e1, 0.0);
  return kOk;
}
// ... and its product, the block's tail and conv1 (i1) of block b + 1 are ONE launch
struct FusedTail {
  int i3;
  const float* y2; const float* s2; const float* t2;
  const float* res; const float* sd; const float* td;
  float* out;
};
int fused_conv1_bn(const Ctx& c, int i1, const FusedTail& f, float* y1) {
  const TrunkConv& d = c.t->convs[i1];
  const TrunkConv& d3 = c.t->convs[f.i3];
  const long M = (long)c.t->B * d.OH * d.OW;
  const int prows = fused_block_tiles(M, d.Cout);
  float* psum = c.ws + c.t->off_part;
  float* psq = psum + (size_t)prows * d.Cout;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (c.t->timing_now) {
    const int rt = timing_begin(c.t, true, c.s, &e0, &e1);
    if (rt) return rt;
  }
  const int rc = fused_block_forward(f.y2, f.s2, f.t2, reinterpret_cast<const unsigned*>(c.w[f.i3]), c.scale(f.i3), c.shift(f.i3),
                                     f.res, f.sd, f.td, f.out, reinterpret_cast<const unsigned*>(c.w[i1]), y1,
                                     c.train ? psum : nullptr, c.train ? psq : nullptr, M, d.Cout, c.in_exp(f.i3), c.in_exp(i1),
                                     c.err, c.s);
  if (c.t->timing_now) {
    if (rc) launch_events() = LaunchEvents{};
    else {
      const int rt = timing_end(c.t, true, c.s, e0, e1, 2.0 * (double)M * d.Cout * d.Cin + 2.0 * (double)M * d3.Cout * d3.Cin);
      if (rt) return rt;
    }
  }
  if (rc) return rc;
  if (c.train == 2)
    return bn_finalize(psum, psq, prows, d.Cout, M, c.gamma[i1], c.beta[i1], nullptr, nullptr, c.momentum, c.eps, c.scale(i1),
                       c.shift(i1), c.s, c.bmean(i1), c.bvar(i1), c.err);
  if (c.train)
    return bn_finalize(psum, psq, prows, d.Cout, M, c.gamma[i1], c.beta[i1], c.rmean[i1], c.rvar[i1], c.momentum, c.eps,
                       c.scale(i1), c.shift(i1), c.s, nullptr, nullptr, c.err);
  return kOk;
}

// conv i with the BatchNorm that follows it folded into the epilogue (inference)
int conv_folded(const Ctx& c, int i, const float* x, const float* res, int relu, float* y) {
  const TrunkConv& d = c.t->convs[i];
  const long M = (long)c.t->B * d.OH * d.OW;
  const long sw = d.Cin, sh = (long)d.W * d.Cin, sb = (long)d.H * d.W * d.Cin;
  CAPNET_REQUIRE(d.tile_n != 256 && !d.fused3 && !d.fused1, "trunk: conv %d was planned for the wide tail kernel or for fused_block.hip; CAPNET_EVAL_FOLDED=1 must be set before the plan is made", i);
  CAPNET_REQUIRE(d.h3 || (d.kmajor && conv_v2_eligible(x, sb, sh, sw, 1, c.t->B, d.Cin, d.Cout, nullptr, nullptr)),
                 "trunk: conv %d is not eligible for the folded-BN kernel", i);
  hipEvent_t e0 = nullptr, e1 = nullptr;
  const bool attach = d.h3;
  if (c.t->timing_now) {
    const int rt = timing_begin(c.t, attach, c.s, &e0, &e1);
    if (rt) return rt;
  }
  int rc;
  if (d.h3) {
    rc = conv_fwd_f16x3(x, sb, sh, sw, reinterpret_cast<const unsigned*>(c.w[i]), d.tile_n, y, nullptr, nullptr,
                        0, nullptr, nullptr, c.t->B, d.H, d.W, d.Cin, d.Cout, d.k, d.stride, d.pad, c.s, c.scale(i),
                        c.shift(i), res, relu, c.in_exp(i), c.err);
  } else {
    rc = conv2d_fwd_v2(x, sb, sh, sw, c.w[i], d.Kw, y, nullptr, nullptr, 0, nullptr, nullptr,
                       c.t->B, d.H, d.W, d.Cin, d.Cout, d.k, d.k, d.stride, d.pad, 0,
                       c.t->tail_balance ? c.ws + c.t->off_slab : nullptr, c.s, c.scale(i),
                       c.shift(i), res, relu);
  }
  if (c.t->timing_now) {
    if (rc) launch_events() = LaunchEvents{};
    else {
      const int rt = timing_end(c.t, attach, c.s, e0, e1, 2.0 * (double)M * d.Cout * d.k * d.k * d.Cin);
      if (rt) return rt;
    }
  }
  return rc;
}

// Inference trunk (encoder.eval(): validation, sample()): every BatchNorm uses its running
// statistics, so it is an affine map known before the conv runs. Each conv applies it in its
// epilogue together with the ReLU, and conv3 of a bottleneck also adds the identity / downsample
// branch: 155 conv launches and nothing else (no raw tensors, no statistics, no bn_add_relu).
int trunk_forward_eval(const Ctx& c, const float* images_nchw, float* out_pooled, float* out_map) {
  Trunk* t = c.t;
  const int B = t->B;
  float* ws = c.ws;
  float* X[2] = {ws + t->off_x[0], ws + t->off_x[1]};
  float* Y1 = ws + t->off_y1;
  float* Y2 = ws + t->off_y2;
  float* Y3 = ws + t->off_y3;
  float* D = ws + t->off_d;
  const int n = (int)t->convs.size();
  {
    std::vector<const float*> g(n), b(n), rm(n), rv(n);
    std::vector<float*> sc(n), sh(n);
    std::vector<int> C(n);
    for (int i = 0; i < n; ++i) {
      g[i] = c.gamma[i]; b[i] = c.beta[i]; rm[i] = c.rmean[i]; rv[i] = c.rvar[i];
      sc[i] = c.scale(i); sh[i] = c.shift(i); C[i] = t->convs[i].Cout;
    }
    int rc = bn_eval_multi(n, g.data(), b.data(), rm.data(), rv.data(), C.data(), sc.data(), sh.data(),
                           c.eps, c.s);
    if (rc) return rc;
  }
  int rc;
  {
    // stem: the gather-loader kernel writes the raw conv, BN + ReLU ride on the max-pool
    const TrunkConv& d = t->convs[0];
    const long M = (long)B * d.OH * d.OW;
    if (d.stem_h3)
      rc = conv_stem_fwd_f16x3(images_nchw, (long)3 * d.H * d.W, (long)d.H * d.W, d.W, reinterpret_cast<const unsigned*>(c.w[0]),
                               Y3, nullptr, nullptr, B, d.H, d.W, c.s, c.in_exp(0), c.err);
    else
      rc = conv2d_fwd(images_nchw, (long)3 * d.H * d.W, d.W, 1, (long)d.H * d.W, c.w[0], d.Kw, Y3, nullptr,
                      nullptr, 0, nullptr, nullptr, B, d.H, d.W, d.Cin, d.Cout, d.k, d.k, d.stride, d.pad,
                      conv_auto_tile((int)M, d.Cout), c.s);
    if (rc) return rc;
    rc = bn_relu_maxpool(Y3, c.scale(0), c.shift(0), X[0], B, d.OH, d.OW, 64, c.s);
    if (rc) return rc;
  }
  int ci = 1, cur = 0;
  const int blocks[4] = {3, 8, 36, 3};
  for (int L = 0; L < 4; ++L) {
    for (int bk = 0; bk < blocks[L]; ++bk) {
      const int i1 = ci, i2 = ci + 1, i3 = ci + 2;
      const int id = (bk == 0) ? ci + 3 : -1;
      ci += (bk == 0) ? 4 : 3;
      const float* x = X[cur];
      float* out = X[cur ^ 1];
      rc = conv_folded(c, i1, x, nullptr, 1, Y1);
      if (rc) return rc;
      rc = conv_folded(c, i2, Y1, nullptr, 1, Y2);
      if (rc) return rc;
      const float* res = x;
      if (id >= 0) {
        rc = conv_folded(c, id, x, nullptr, 0, D);
        if (rc) return rc;
        res = D;
      }
      rc = conv_folded(c, i3, Y2, res, 1, out);
      if (rc) return rc;
      cur ^= 1;
    }
  }
  const int side = t->final_side;
  if (out_pooled) {
    rc = global_avgpool(X[cur], out_pooled, B, side * side, 2048, c.s, c.err);
    if (rc) return rc;
  }
  if (out_map)
    CAPNET_HIP_CHECK(hipMemcpyAsync(out_map, X[cur], (size_t)B * side * side * 2048 * sizeof(float),
                                    hipMemcpyDeviceToDevice, c.s));
  return kOk;
}
}  // namespace

int trunk_update_running(Trunk* t, const float* workspace, float* const* bn_rmean, float* const* bn_rvar,
                         float momentum, hipStream_t stream) {
  CAPNET_REQUIRE(t && workspace && bn_rmean && bn_rvar, "trunk_update_running: null argument");
  const int n = (int)t->convs.size();
  std::vector<const float*> mean(n), var(n);
  std::vector<int> C(n);
  for (int i = 0; i < n; ++i) {
    mean[i] = workspace + t->bs_off[i];
    var[i] = workspace + t->bs_off[i] + t->convs[i].Cout;
    C[i] = t->convs[i].Cout;
  }
  return bn_running_update_multi(n, mean.data(), var.data(), bn_rmean, bn_rvar, C.data(), momentum, stream);
}

int trunk_forward(Trunk* t, const float* images_nchw, const float* const* w_packed,
                  const float* const* bn_gamma, const float* const* bn_beta,
                  float* const* bn_rmean, float* const* bn_rvar, int train, float momentum,
                  float eps, float* workspace, float* out_pooled, float* out_map,
                  const int* in_exps, int* err_flag, hipStream_t stream) {
  CAPNET_REQUIRE(t && images_nchw && w_packed && bn_gamma && bn_beta && bn_rmean && bn_rvar &&
                     workspace,
                 "trunk_forward: null argument");
  CAPNET_REQUIRE(out_pooled || out_map, "trunk_forward: no output requested");
  CAPNET_REQUIRE(aligned16(workspace), "trunk_forward: workspace must be 16-B aligned");
  Ctx c{t, w_packed, bn_gamma, bn_beta, bn_rmean, bn_rvar, train, momentum, eps, workspace, stream};
  c.in_exps = in_exps;
  c.err = err_flag;
  t->timing_now = t->timing_once || (t->timing && (t->pass_no++ % t->timing_every == 0));
  t->timing_once = false;
  const int B = t->B;
  float* X[2] = {workspace + t->off_x[0], workspace + t->off_x[1]};
  float* Y1 = workspace + t->off_y1;
  float* Y2 = workspace + t->off_y2;
  float* Y3 = workspace + t->off_y3;
  float* D = workspace + t->off_d;
  int rc;
  int ci = 0;
  // Inference. CAPNET_EVAL_FOLDED=1: the convolutions' epilogues apply the BatchNorms (+ residual + ReLU): the least
  // traffic, but that epilogue of the split-f16 kernel is slow (12.3 ms per pass at B = 64). Default: the training
  // pass's kernels with every (scale, shift) computed up front from the running statistics -- no statistics, no
  // finalize launches, BatchNorm + ReLU folded into the consumers' staging and the tails into the next conv1.
  if (!train) {
    const char* fe = getenv("CAPNET_EVAL_FOLDED");      // (read per call: the tests run both paths in one process)
    if (fe && fe[0] == '1') return trunk_forward_eval(c, images_nchw, out_pooled, out_map);
    const int n = (int)t->convs.size();
    std::vector<const float*> g(n), b(n), rm(n), rv(n);
    std::vector<float*> sc(n), sh(n);
    std::vector<int> Cs(n);
    for (int i = 0; i < n; ++i) {
      g[i] = c.gamma[i]; b[i] = c.beta[i]; rm[i] = c.rmean[i]; rv[i] = c.rvar[i];
      sc[i] = c.scale(i); sh[i] = c.shift(i); Cs[i] = t->convs[i].Cout;
    }
    rc = bn_eval_multi(n, g.data(), b.data(), rm.data(), rv.data(), Cs.data(), sc.data(), sh.data(), c.eps, c.s);
    if (rc) return rc;
    c.eval_ready = true;
  }
  // stem: NCHW image read through the generic gather loader
  {
    const TrunkConv& d = t->convs[0];
    rc = conv_bn(c, 0, images_nchw, (long)3 * d.H * d.W, d.W, 1, (long)d.H * d.W, nullptr,
                 nullptr, 0, Y3);
    if (rc) return rc;
    rc = bn_relu_maxpool(Y3, c.scale(0), c.shift(0), X[0], B, d.OH, d.OW, 64, stream);
    if (rc) return rc;
    ci = 1;
  }
  int cur = 0;
  BlockTail tail{};
  FusedTail ftail{};
  bool have_tail = false, have_ftail = false;
  const int blocks[4] = {3, 8, 36, 3};
  for (int L = 0; L < 4; ++L) {
    for (int b = 0; b < blocks[L]; ++b) {
      const int i1 = ci, i2 = ci + 1, i3 = ci + 2;
      const int id = (b == 0) ? ci + 3 : -1;
      ci += (b == 0) ? 4 : 3;
      const TrunkConv& c1 = t->convs[i1];
      const TrunkConv& c2 = t->convs[i2];
      const TrunkConv& c3 = t->convs[i3];
      const float* x = X[cur];
      float* out = X[cur ^ 1];
      auto nhwc = [](const TrunkConv& d, long* sb, long* sh, long* sw) {
        *sw = d.Cin; *sh = (long)d.W * d.Cin; *sb = (long)d.H * d.W * d.Cin;
      };
      long sb, sh, sw;
      nhwc(c1, &sb, &sh, &sw);
      if (have_ftail) {
        // the previous block's conv3 product, its tail (-> x = X[cur]) and this conv1: one launch (fused_block.hip)
        CAPNET_REQUIRE(c1.fused1 && (c.train || c.eval_ready), "trunk: conv %d is not the fused boundary the plan made", i1);
        rc = fused_conv1_bn(c, i1, ftail, Y1);
        have_ftail = false;
      } else {
        CAPNET_REQUIRE(!c1.fused1, "trunk: conv %d was planned for fused_block.hip", i1);
        rc = conv_bn(c, i1, x, sb, sh, sw, 1, nullptr, nullptr, 0, Y1, have_tail ? &tail : nullptr);
      }
      have_tail = false;
      if (rc) return rc;
      nhwc(c2, &sb, &sh, &sw);
      rc = conv_bn(c, i2, Y1, sb, sh, sw, 1, c.scale(i1), c.shift(i1), 1, Y2);
      if (rc) return rc;
      const long rows = (long)B * c3.OH * c3.OW;
      if (c3.fused3 && (c.train || c.eval_ready)) {
        // y3 is never formed: statistics from y2's second moments now, the product inside the next block's conv1 launch
        rc = fused_stats_bn(c, i3, Y2, c.scale(i2), c.shift(i2));
        if (rc) return rc;
        if (id >= 0) {
          const TrunkConv& cd = t->convs[id];
          nhwc(cd, &sb, &sh, &sw);
          rc = conv_bn(c, id, x, sb, sh, sw, 1, nullptr, nullptr, 0, D);
          if (rc) return rc;
          ftail = FusedTail{i3, Y2, c.scale(i2), c.shift(i2), D, c.scale(id), c.shift(id), out};
        } else {
          ftail = FusedTail{i3, Y2, c.scale(i2), c.shift(i2), x, nullptr, nullptr, out};
        }
        have_ftail = true;
        cur ^= 1;
        continue;
      }
      CAPNET_REQUIRE(!c3.fused3, "trunk: conv %d was planned for fused_block.hip (the folded inference trunk must be chosen before the plan is made)", i3);
      nhwc(c3, &sb, &sh, &sw);
      rc = conv_bn(c, i3, Y2, sb, sh, sw, 1, c.scale(i2), c.shift(i2), 1, Y3);
      if (rc) return rc;
      if (id >= 0) {
        const TrunkConv& cd = t->convs[id];
        nhwc(cd, &sb, &sh, &sw);
        rc = conv_bn(c, id, x, sb, sh, sw, 1, nullptr, nullptr, 0, D);
        if (rc) return rc;
        tail = BlockTail{Y3, c.scale(i3), c.shift(i3), D, c.scale(id), c.shift(id), out, rows, c3.Cout};
      } else {
        tail = BlockTail{Y3, c.scale(i3), c.shift(i3), x, nullptr, nullptr, out, rows, c3.Cout};
      }
      have_tail = true;      // runs inside the next block's conv1 (conv_bn), or below after the last block
      cur ^= 1;
    }
  }
  if (have_tail) {
    rc = run_tail(c, tail);
    if (rc) return rc;
  }
  const int side = t->final_side;
  if (out_pooled) {
    rc = global_avgpool(X[cur], out_pooled, B, side * side, 2048, stream, c.err);
    if (rc) return rc;
  }
  if (out_map) {
    CAPNET_HIP_CHECK(hipMemcpyAsync(out_map, X[cur], (size_t)B * side * side * 2048 * sizeof(float),
                                    hipMemcpyDeviceToDevice, stream));
  }
  return kOk;
}

}  // namespace capnet
