// Internal C++ entry points of libcapnet_hip (one per kernel family). The extern "C"
// boundary in capi.cpp forwards to these; nothing here allocates or synchronises.
#pragma once
#include <hip/hip_runtime.h>

namespace capnet {

// gemm_f32.hip
int sgemm(bool ta, bool tb, int M, int N, int K, const float* A, long lda, const float* B,
          long ldb, float* C, long ldc, const float* bias, int accumulate, int batch, long sA,
          long sB, long sC, long sBias, int force_tile, hipStream_t stream);

// gemm_b3.hip: the same product on the bf16 matrix cores, three bf16 pieces per fp32 operand and six products per multiply
// (fp32-grade, fp32's range); sgemm takes it by itself for large products
bool sgemm_b3_eligible(bool ta, bool tb, int M, int N, int K, const float* A, long lda, const float* B, long ldb,
                       const float* C, long ldc, const float* bias, int batch, long sA, long sB, long sC, long sBias);
int sgemm_b3(bool ta, bool tb, int M, int N, int K, const float* A, long lda, const float* B, long ldb, float* C, long ldc,
             const float* bias, int accumulate, int batch, long sA, long sB, long sC, long sBias, hipStream_t stream,
             float* ws = nullptr, size_t ws_floats = 0);

// counters (optional): n_counters ints, zero before the first use and left zero by every call -- one per 64-column
// output tile and batch member; with them, products of M <= 16 rows are ONE launch (gemm_rows16_kernel: the
// workgroup that arrives last at a tile sums the K-chunk partials)
constexpr size_t kSplitKCounters = 1024;
int sgemm_splitk(bool ta, bool tb, int M, int N, int K, const float* A, long lda, const float* B,
                 long ldb, float* C, long ldc, const float* bias, int accumulate, float* ws,
                 size_t ws_floats, hipStream_t stream, int* counters = nullptr, size_t n_counters = 0);
int sgemm_splitk_slabs(bool tb, int M, int N, int K, const float* A, long lda, const float* B, long ldb,
                       float* ws, size_t ws_floats, int* n_slabs, hipStream_t stream);
int sgemm_rows16_slabs(bool tb, int M, int N, int K, const float* A, long lda, const float* B, long ldb, float* ws,
                       size_t ws_floats, int* n_slabs, hipStream_t stream);
int sgemm_splitk_batched(bool ta, bool tb, int M, int N, int K, const float* A, long lda,
                         const float* B, long ldb, float* C, long ldc, const float* bias,
                         int accumulate, int batch, long sA, long sB, long sC, long sBias, float* ws,
                         size_t ws_floats, hipStream_t stream, int* counters = nullptr, size_t n_counters = 0);

// conv_f32.hip
int conv2d_fwd(const float* x, long sxb, long sxh, long sxw, long sxc, const float* w_packed,
               int Kw, float* y, const float* in_scale, const float* in_shift, int relu_in,
               float* part_sum, float* part_sq, int Bn, int H, int W, int Cin, int Cout, int KH,
               int KW, int stride, int pad, int tile, hipStream_t stream);
int conv_auto_tile(int M, int Cout);
// conv_f32_v2.hip (weights packed K-major [Kw][Cout])
bool conv_v2_eligible(const float* x, long sxb, long sxh, long sxw, long sxc, int Bn, int Cin,
                      int Cout, const float* in_scale, const float* in_shift);
int conv2d_fwd_v2(const float* x, long sxb, long sxh, long sxw, const float* wk, int Kw, float* y,
                  const float* in_scale, const float* in_shift, int relu_in, float* part_sum,
                  float* part_sq, int Bn, int H, int W, int Cin, int Cout, int KH, int KW,
                  int stride, int pad, int tile, float* slabs, hipStream_t stream,
                  const float* out_scale = nullptr, const float* out_shift = nullptr,
                  const float* res = nullptr, int relu_out = 0);
size_t conv_v2_slab_floats(int M, int Cout, int Kw, int tile);
int conv_v2_auto_tile(int M, int Cout, int Kw);
void conv_v2_plan(int M, int Cout, int Kw, int tile, int* out);
int conv_tiles_m(int M, int tile);
// gemm_dma.hip: C = A . B^T with both operands K-contiguous, LDS-DMA staging (1x1 convs, vocabulary projection)
bool sgemm_nt_dma_eligible(int M, int N, int K, const float* A, long lda, const float* B, long ldb,
                           const float* C, long ldc);
int sgemm_nt_dma(int M, int N, int K, const float* A, long lda, const float* B, float* C,
                 const float* bias, hipStream_t stream);
bool conv1x1_dma_eligible(const float* x, long sxb, long sxh, long sxw, long sxc, int Bn, int H, int W,
                          int Cin, int Cout, int stride);
int conv1x1_tiles_m(long M);
int conv1x1_fwd_dma(const float* x, long sxb, long sxh, long sxw, const float* w_oi, float* y,
                    float* part_sum, float* part_sq, int Bn, int H, int W, int Cin, int Cout, int stride,
                    hipStream_t stream, const float* out_scale = nullptr, const float* out_shift = nullptr,
                    const float* res = nullptr, int relu_out = 0);

// conv_f16x3.hip: 1x1 convolution as three f16 MFMA products of 2-way split, power-of-two scaled fp32 operands
// (fp32-grade results; the default for Cin % 64 == 0)
bool conv1x1_f16x3_eligible(const float* x, long sxb, long sxh, long sxw, long sxc, int Bn, int H, int W,
                            int Cin, int Cout, int stride, const float* in_scale, const float* in_shift);
int conv1x1_f16x3_bn(long M, int Cout);
// ... and the 3x3 (pad 1) convolutions as an implicit GEMM over (tap, channel) in the same kernel: k = 1 or 3
bool conv_f16x3_eligible(const float* x, long sxb, long sxh, long sxw, long sxc, int Bn, int H, int W, int Cin,
                         int Cout, int k, int stride, int pad, const float* in_scale, const float* in_shift);
size_t conv_f16x3_weight_words(int Cin, int Cout, int k);
int conv_f16x3_pack(const float* w_oihw, unsigned* img, int Cout, int Cin, int k, int bn, hipStream_t stream);
int conv_fwd_f16x3(const float* x, long sxb, long sxh, long sxw, const unsigned* wimg, int bn, float* y,
                   const float* in_scale, const float* in_shift, int relu_in, float* part_sum, float* part_sq, int Bn,
                   int H, int W, int Cin, int Cout, int k, int stride, int pad, hipStream_t stream,
                   const float* out_scale = nullptr, const float* out_shift = nullptr, const float* res = nullptr,
                   int relu_out = 0, int in_exp = 0, int* err = nullptr);
// (in_exp, here and below: the input is multiplied by 2^in_exp on its way into the f16 planes -- folded into the
//  BatchNorm's scale / shift where there is one -- and the accumulators by 2^-in_exp: exact, and what keeps the split
//  operands inside f16's range whatever the scale of the tensor; chosen per tensor by the trunk)
// The stem (7x7 stride 2 pad 3, 3 -> 64 channels, NCHW image in, NHWC out) on the same split-f16 arithmetic
// (conv_stem.hip); statistics partials: one row per workgroup
bool conv_stem_f16x3_eligible(const float* x, long sxb, long sxc, long sxh, long sxw, int Bn, int H, int W, int Cin,
                              int Cout, int k, int stride, int pad);
size_t conv_stem_f16x3_weight_words();
int conv_stem_f16x3_part_rows(int Bn, int H, int W);
int conv_stem_f16x3_pack(const float* w_oihw, unsigned* img, hipStream_t stream);
int conv_stem_fwd_f16x3(const float* x, long sxb, long sxc, long sxh, const unsigned* wimg, float* y, float* part_sum,
                        float* part_sq, int Bn, int H, int W, hipStream_t stream, int in_exp = 0, int* err = nullptr);
// ... and the stride-1 3x3 ones with the tile's input patch resident in LDS (conv3x3_patch.hip): same weight image,
// tile width and statistics rows as conv_fwd_f16x3; dense NHWC input
bool conv3x3_patch_eligible(const float* x, long sxb, long sxh, long sxw, long sxc, int Bn, int H, int W, int Cin,
                            int Cout, int k, int stride, int pad, const float* in_scale, const float* in_shift);
int conv3x3_fwd_patch(const float* x, const unsigned* wimg, int bn, float* y, const float* in_scale, const float* in_shift,
                      int relu_in, float* part_sum, float* part_sq, int Bn, int H, int W, int Cin, int Cout,
                      hipStream_t stream, bool shared_chip = false, int in_exp = 0, int* err = nullptr);
// Stride-1 1x1 convolutions with Cin = 64 / 128 / 256 (conv3 of stages 1-3) with the A operand resident in registers
// (conv1x1_areg.hip): dense [M][Cin] input, same weight image, tile width and statistics rows as conv_fwd_f16x3
bool conv1x1_areg_eligible(const float* x, long M, int Cin, int Cout, int bn, const float* in_scale, const float* in_shift);
int conv1x1_fwd_areg(const float* x, const unsigned* wimg, int bn, float* y, const float* in_scale, const float* in_shift,
                     int relu_in, float* part_sum, float* part_sq, long M, int Cin, int Cout, int in_exp, hipStream_t stream,
                     int* err = nullptr);
// A bottleneck block's tail (bn_add_relu) fused into the next block's stride-1 1x1 conv1 (conv3x3_patch.hip)
bool conv1x1_tail_eligible(const float* y3, const float* res, long M, int Cin, int Cout);
int conv1x1_fwd_tail(const float* y3, const float* s1, const float* t1, const float* res, const float* s2, const float* t2,
                     float* tail_out, const unsigned* wimg, int bn, float* y, float* part_sum, float* part_sq, long M,
                     int Cin, int Cout, hipStream_t stream, int in_exp = 0, int* err = nullptr);
size_t conv1x1_f16x3_weight_words(int Cin, int Cout);
int conv1x1_f16x3_pack(const float* w, unsigned* img, int Cout, int Cin, int bn, hipStream_t stream);
int conv1x1_fwd_f16x3(const float* x, long sxb, long sxh, long sxw, const unsigned* wimg, int bn, float* y,
                      const float* in_scale, const float* in_shift, int relu_in, float* part_sum,
                      float* part_sq, int Bn, int H, int W, int Cin, int Cout, int stride,
                      hipStream_t stream, const float* out_scale = nullptr, const float* out_shift = nullptr,
                      const float* res = nullptr, int relu_out = 0);

// bn_pool.hip
int bn_finalize(const float* part_sum, const float* part_sq, int tiles, int C, long count,
                const float* gamma, const float* beta, float* running_mean, float* running_var,
                float momentum, float eps, float* scale, float* shift, hipStream_t stream,
                float* batch_mean = nullptr, float* batch_var = nullptr, int* err = nullptr);
int bn_eval_scale_shift(const float* gamma, const float* beta, const float* rm, const float* rv,
                        float eps, int C, float* scale, float* shift, hipStream_t stream);
int bn_running_update_multi(int n, const float* const* mean, const float* const* var, float* const* rm,
                            float* const* rv, const int* C, float momentum, hipStream_t stream);
int bn_eval_multi(int n, const float* const* gamma, const float* const* beta, const float* const* rm,
                  const float* const* rv, const int* C, float* const* scale, float* const* shift,
                  float eps, hipStream_t stream);
int bn_add_relu(const float* y, const float* s1, const float* t1, const float* res,
                const float* s2, const float* t2, float* out, long rows, int C,
                hipStream_t stream);
int bn_relu_maxpool(const float* y, const float* scale, const float* shift, float* out, int Bn,
                    int H, int W, int C, hipStream_t stream);
int global_avgpool(const float* x, float* out, int Bn, int HW, int C, hipStream_t stream, int* err = nullptr);
int adaptive_pool_replicate(const float* x, float* out, int Bn, int S, int OUT, int C,
                            hipStream_t stream);
int pack_conv_weight_kmajor(const float* w_oihw, float* out, int Cout, int Cin, int KH, int KW,
                            int Kw, hipStream_t stream);
int pack_conv_weight(const float* w_oihw, float* out, int Cout, int Cin, int KH, int KW, int Kw,
                     hipStream_t stream);

// trunk.cpp
struct Trunk;
int trunk_create(int B, int H, int W, Trunk** out);
void trunk_destroy(Trunk* t);
size_t trunk_workspace_bytes(const Trunk* t);
int trunk_num_convs(const Trunk* t);
int trunk_final_side(const Trunk* t);
int trunk_conv_shape(const Trunk* t, int i, int* cout, int* cin, int* k, int* stride, int* kw);
int trunk_conv_kmajor(const Trunk* t, int i);
int trunk_conv_tile_n(const Trunk* t, int i);
double trunk_flops(const Trunk* t);
double trunk_conv_flops(const Trunk* t, int i);
int trunk_set_timing(Trunk* t, int enable);
int trunk_time_next_pass(Trunk* t);
int trunk_collect_timing(Trunk* t, double* conv_ms, long* conv_launches, double* conv_flops);
int trunk_set_tail_balance(Trunk* t, int on);
int trunk_update_running(Trunk* t, const float* workspace, float* const* bn_rmean, float* const* bn_rvar,
                         float momentum, hipStream_t stream);
int trunk_forward(Trunk* t, const float* images_nchw, const float* const* w_packed,
                  const float* const* bn_gamma, const float* const* bn_beta,
                  float* const* bn_rmean, float* const* bn_rvar, int train, float momentum,
                  float eps, float* workspace, float* out_pooled, float* out_map,
                  const int* in_exps, int* err_flag, hipStream_t stream);

// seq_kernels.hip
constexpr int kMaxSteps = 128;
struct SeqMeta {
  int N, steps, has_features;
  int off[kMaxSteps + 1];
  unsigned char tf[kMaxSteps];
};
int build_rows(const SeqMeta& m, int* row_sample, int* row_col, int* row_token, int* prev_row,
               hipStream_t stream);
int gather_inputs(const long long* captions, int T, const float* features, const float* emb, int E,
                  int V, const int* row_sample, const int* row_col, int* row_token, float* X,
                  long ldx, int r0, int r1, float p, unsigned long long seed, int use_dropout,
                  int dynamic, int* err_flag, hipStream_t stream);
int embedding_fwd(const long long* idx, int n, const float* emb, int E, int V, float* out,
                  int* err_flag, hipStream_t stream);
int packed_targets(const SeqMeta& m, const long long* captions, int T, long long* out,
                   hipStream_t stream);
int vec_add(const float* a, const float* b, float* out, int n, hipStream_t stream);
// dst[i] = src[i] (+ src2[i] where given), up to 40 items in one launch
struct CopyTable {
  static constexpr int kMax = 40;
  const float* src[kMax];
  const float* src2[kMax];
  float* dst[kMax];
  size_t n[kMax];
  int count = 0;
  void add(float* d, const float* a, size_t len, const float* b = nullptr) {
    if (count < kMax) { src[count] = a; src2[count] = b; dst[count] = d; n[count] = len; }
    ++count;      // (an overflow is caught by multi_copy)
  }
};
int multi_copy(const CopyTable& t, hipStream_t stream);
// (slabs: n_slabs partial products [k][b][4H] still to be added to `pre`, in slab order -- sgemm_rows16_slabs)
int lstm_pointwise_fwd(float* pre, long ldp, const float* c_prev, float* c_out, float* h_out, int b,
                       int H, int gi, int gf, int go, int gg, int tanh_out, hipStream_t stream,
                       const float* slabs = nullptr, int n_slabs = 0);
int lstm_pointwise_bwd(const float* gates, long ldg, const float* c, const float* c_prev,
                       const float* dH, const float* dh_rec, float* dc_io, float* dpre, long ldq,
                       int b, int b_next, int H, int gi, int gf, int go, int gg, int tanh_out,
                       hipStream_t stream, int dh_slabs = 0, long dh_slab_stride = 0);
int gather_prev_rows(const float* src, const int* idx, const float* first, const int* sample,
                     float* out, int rows, int C, hipStream_t stream);
int argmax_rows(const float* x, int rows, int ld, int V, int* out, hipStream_t stream);
int beam_topk_batched(const float* logits, long ld, int V, const float* prev, const int* meta, int n, float* out_scores,
                      long long* out_index, hipStream_t stream);
int beam_topk(const float* logits, long ld, int rows, int V, const float* prev, int k,
              float* out_scores, long long* out_index, hipStream_t stream);
int gather_rows(const float* src, const int* idx, float* out, int rows, int C, hipStream_t stream);
int colsum(const float* x, long ld, int rows, int C, float* out, int accumulate, hipStream_t stream,
           float* ws = nullptr, size_t ws_floats = 0);
// out[m][n] = sum_k slab[k][m][n] (+ bias[n]) (+ out[m][n]), fixed order (gemm_f32.hip)
int reduce_slabs(const float* slab, int count, int M, int N, float* out, long ldc, const float* bias,
                 int accumulate, hipStream_t stream);
// dst[r][e] = src[r][e] * mask(seed, layer, r, e) / keep for rows [r0, r1) of width C (the dropout between stacked layers;
// use_dropout 0: a copy). The same call maps a gradient back through the mask.
int rows_dropout(const float* src, float* dst, int r0, int r1, int C, float p, unsigned long long seed, int layer,
                 int use_dropout, hipStream_t stream);
int scatter_input_grad(const float* dX, long ldx, int N, int E, const int* row_sample, const int* row_col,
                       const int* row_token, float* dEmb, float* dFeat, int V, float p,
                       unsigned long long seed, int use_dropout, hipStream_t stream, int* tables, size_t table_ints);

// lstm_step.hip
bool lstm_step_fused_supported(int b, int H);
size_t lstm_wfrag_floats(int H);
int lstm_pack_wfrag(const float* Wcat, float* Wfrag, int H, int gi, int gf, int go, int gg,
                    hipStream_t stream);
int lstm_step_fused(const float* hprev, const float* Wfrag, float* G, long ldg, const float* cprev,
                    float* c_out, float* h_out, int b, int H, int gi, int gf, int go, int gg,
                    int tanh_out, hipStream_t stream, unsigned long long* stamps = nullptr);

// lstm_persist.hip: a run of teacher-forced steps [t0, t1) in one launch (H = 512, b <= 128)
bool lstm_persist_supported(int b, int H);
size_t lstm_persist_w_floats();
size_t lstm_persist_ctl_ints();
int lstm_persist_pack(const float* Wcat, float* Wp, int gi, int gf, int go, int gg, hipStream_t stream);
int lstm_persist_run(const float* Wp, float* G, float* Cst, float* hiddens, const int* off,
                     const int* batch_sizes, int t0, int t1, int H, int gi, int gf, int go, int gg,
                     int tanh_out, int seg, int* ctl, int* err_flag, hipStream_t stream,
                     unsigned long long* stamps);

// att_kernels.hip
int att_step_fwd(const float* att1, const float* feat, const float* att2, float* gate_io, long ldz,
                 const float* wf, const float* bf, int rows, int P, int A, int C,
                 float* alpha_out, float* alphas_bt, int steps, int t, float* awe_out,
                 float* xa_out, long ldx, float* escore, hipStream_t stream);
int att_step_bwd(const float* att1, const float* feat, const float* att2, long ldz2,
                 const float* gate, long ldzg, const float* awe, const float* alpha,
                 const float* wf, float* dxa, long ldx, const float* dalphas_bt, int steps,
                 int t, int rows, int P, int A, int C, float* dalpha_part, float* dgate_out,
                 float* datt2, long ldz, float* de_out, float* dwf_rows, float* dbf_rows,
                 hipStream_t stream, const float* dxa_slabs = nullptr, int n_slabs = 0, int emb_cols = 0);
int att_datt1(const float* att1, const float* att2_rows, long ldz2, const float* de_rows,
              const float* wf, const int* off, int steps, int B, int P, int A, float* datt1,
              hipStream_t stream);

// image_ops.hip
int resize_u8(const unsigned char* src, int Hs, int Ws, unsigned char* tmp, unsigned char* dst, int Ho,
              int Wo, const int* bounds_h, const int* coef_h, int kmax_h, const int* bounds_v,
              const int* coef_v, int kmax_v, hipStream_t stream);
int crop_flip_normalize(const unsigned char* src, int B, int Hs, int Ws, const int* params, float* dst,
                        int Hc, int Wc, const float* mean, const float* stdv, hipStream_t stream);

// loss_optim.hip
int xent_fwd(const float* logits, long ld, int N, int V, const long long* targets, float* lse,
             float* row_loss, float* loss, int* err_flag, hipStream_t stream);
int topk_correct(const float* logits, long ld, int N, int V, const long long* targets, int k,
                 int* count, int* err_flag, hipStream_t stream);
int xent_bwd(const float* logits, long ld, int N, int V, const long long* targets,
             const float* lse, const float* gout, float* dlogits, long ldd, hipStream_t stream);
int att_loss_fwd(const float* nll, const float* alphas, int B, int steps, int P, float alpha_c,
                 float* colsum, float* out, hipStream_t stream);
int att_loss_bwd(const float* gout, const float* colsum, int B, int steps, int P, float alpha_c,
                 float* dalphas, hipStream_t stream);
int lstm_persist_set_mode(int mode);
int clamp_adam(int n_tensors, float* const* params, float* const* grads, float* const* exp_avg,
               float* const* exp_avg_sq, const long* numel, const int* step, float lr, float b1,
               float b2, float eps, float clip, int write_grad, const int* skip_flag, hipStream_t stream);
// fused_block.hip: conv3 + BatchNorm3 + residual + ReLU + the next conv1 without y3 (statistics from the Gram matrix of conv3's input)
bool fused_block_shape_ok(long M, int MID);
size_t fused_block_weight_words(int C, int MID, int role);
int fused_block_pack(const float* w, unsigned* img, int C, int MID, int role, hipStream_t stream);
size_t fused_block_stats_floats(long M, int K);
int fused_block_stats(const float* y2, const float* s2, const float* t2, const unsigned* w3img, long M, int MID, int in_exp,
                      const float* gamma, const float* beta, float* running_mean, float* running_var, float momentum, float eps,
                      float* scale, float* shift, float* batch_mean, float* batch_var, float* work, int* err, hipStream_t stream);
int fused_block_tiles(long M, int MID);
int fused_block_forward(const float* y2, const float* s2, const float* t2, const unsigned* w3img, const float* s3, const float* t3,
                        const float* res, const float* sd, const float* td, float* out, const unsigned* w1img, float* y1,
                        float* part_sum, float* part_sq, long M, int MID, int e3, int e1, int* err, hipStream_t stream);
struct Comm;
int comm_unique_id(void* id128);
int comm_create(const void* id128, int rank, int world, Comm** out);
int comm_destroy(Comm* c);
int allreduce_grads(Comm* c, float* flat, long count, hipStream_t stream);
int err_word_exchange(int* err_flag, float* slot, int dir, hipStream_t stream);
int count_skipped(const int* err_flag, int* counter, hipStream_t stream);
int pack_tensors(int n_tensors, float* const* tensors, const long* numel, float* flat, int dir,
                 float scale, hipStream_t stream);
int clamp_inplace(float* x, long n, float lo, float hi, hipStream_t stream);
int bn1d_fwd(const float* x, int B, int C, const float* gamma, const float* beta, float* rmean,
             float* rvar, int train, float momentum, float eps, float* y, float* save_mean,
             float* save_invstd, hipStream_t stream);
int bn1d_bwd(const float* dy, const float* x, int B, int C, const float* gamma,
             const float* save_mean, const float* save_invstd, float* dx, float* dgamma,
             float* dbeta, hipStream_t stream);

// decoder_seq.cpp
constexpr int kCellFactored = 0;  // DecoderFactoredLSTM (stylenet/model.py)
constexpr int kCellLSTM = 1;      // nn.LSTMCell (nic/model.py)
struct SeqDims {
  int B, T, steps, N, E, F, H, V, has_features, cell;
};
// Factored: Vw/Vb/Sw/Sb/Uw/Ub/Ww/Wb per gate (i, f, o, c). LSTM cell: Vw[0] = weight_ih,
// Vb[0] = bias_ih, Ww[0] = weight_hh, Wb[0] = bias_hh (gate order i, f, g, o).
struct SeqWeights {
  const float* Vw[4]; const float* Vb[4];
  const float* Sw[4]; const float* Sb[4];
  const float* Uw[4]; const float* Ub[4];
  const float* Ww[4]; const float* Wb[4];
};
// packed gradients: dVcat [4F][E] (or weight_ih [4H][E]), dbV [4F], dScat [4][F][F], dbS [4F],
// dUcat [4][H][F], dbUW [4H] (= grad of U bias = grad of W bias; LSTM: of both biases),
// dWcat [4H][H], dEmb [V][E], dFeat [B][E] or null
struct SeqGrads {
  float* dVcat; float* dbV; float* dScat; float* dbS; float* dUcat; float* dbUW; float* dWcat;
  float* dEmb; float* dFeat;
};
size_t seq_saved_floats(const SeqDims& d);
size_t seq_saved_ints(const SeqDims& d);
size_t seq_fwd_scratch_floats(const SeqDims& d);
size_t seq_bwd_scratch_floats(const SeqDims& d);
int seq_forward(const SeqDims& d, const int* batch_sizes, const unsigned char* tf_mask,
                const long long* captions, const float* features, const float* emb,
                const SeqWeights& w, const float* Cw, const float* Cb, float dropout_p,
                unsigned long long seed, int training, float* saved, int* saved_i, float* scratch,
                float* hiddens, int* err_flag, hipStream_t s);
SeqDims seq_upper_dims(const SeqDims& d0);
int seq_forward_stacked(const SeqDims& d0, int nlayers, const int* batch_sizes, const unsigned char* tf_mask,
                        const long long* captions, const float* features, const float* emb, const SeqWeights* w,
                        const float* Cw, const float* Cb, float dropout_p, unsigned long long seed, int training,
                        float* const* saved, int* const* saved_i, float* scratch, float* const* hiddens, int* err_flag,
                        hipStream_t s);
int seq_backward_stacked(const SeqDims& d0, int nlayers, const int* batch_sizes, const float* dH_top,
                         const float* const* hiddens, const float* const* saved, const int* const* saved_i, float* scratch,
                         float* const* dH_work, const SeqGrads* g, float dropout_p, unsigned long long seed, int training,
                         hipStream_t s);
int seq_backward(const SeqDims& d, const int* batch_sizes, const float* dH, const float* hiddens,
                 const float* saved, const int* saved_i, float* scratch, const SeqGrads& g,
                 float dropout_p, unsigned long long seed, int training, hipStream_t s);

// decoder_att_seq.cpp -- DecoderFactoredLSTMAtt (stylenet/model_att.py:73-305)
struct AttDims {
  int B, T, steps, N, E, F, H, V, A, P, C;
  int cell;  // kCellFactored: DecoderFactoredLSTMAtt; kCellLSTM: nic DecoderRNNAtt (nn.LSTMCell)
};
struct AttWeights {
  const float* Vw[4]; const float* Vb[4];   // V_g: [F][E+C]
  const float* Sw[4]; const float* Sb[4];   // mode-selected S_g
  const float* Uw[4]; const float* Ub[4];
  const float* Ww[4]; const float* Wb[4];
  const float* init_h_w; const float* init_h_b; const float* init_c_w; const float* init_c_b;
  const float* enc_att_w; const float* enc_att_b;   // [A][C]   (mode-selected attention module)
  const float* dec_att_w; const float* dec_att_b;   // [A][H]
  const float* full_att_w; const float* full_att_b; // [1][A], [1]
  const float* f_beta_w; const float* f_beta_b;     // [C][H]
};
// dWz [4H+A+C][H] = [dW_i; dW_f; dW_o; dW_c; d decoder_att; d f_beta]; dbz likewise
// (dbz[0:4H] is the gradient of both the U and the W biases)
struct AttGrads {
  float* dVcat; float* dbV; float* dScat; float* dbS; float* dUcat; float* dWz; float* dbz;
  float* dWe; float* dbe; float* dwf; float* dbf; float* dWih; float* dbih; float* dWic;
  float* dbic; float* dEmb;
};
int att_set_chain_mode(int mode);      // the factored input product as one matrix per call: 0 by shape, 1 always, -1 never
size_t att_saved_floats(const AttDims& d);
size_t att_saved_ints(const AttDims& d);
size_t att_fwd_scratch_floats(const AttDims& d);
size_t att_bwd_scratch_floats(const AttDims& d);
int att_seq_forward(const AttDims& d, const int* bs, const unsigned char* tf,
                    const long long* captions, const float* feat, const float* emb,
                    const AttWeights& w, const float* Cw, const float* Cb, float dropout_p,
                    unsigned long long seed, int training, float* saved, int* saved_i,
                    float* scratch, float* hiddens, float* alphas_bt, int* err_flag,
                    hipStream_t s);
int att_seq_backward(const AttDims& d, const int* bs, const float* dH, const float* dalphas_bt,
                     const float* hiddens, const float* feat, const AttWeights& w,
                     const float* saved, const int* saved_i, float* scratch, const AttGrads& g,
                     float dropout_p, unsigned long long seed, int training, hipStream_t s);

}  // namespace capnet
