// extern "C" boundary of libcapnet_hip.so (declared in include/capnet.h).
#include "../../include/capnet.h"

#include "common.h"
#include "kernels.h"

using namespace capnet;

static inline hipStream_t S(capnet_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

static SeqDims to_dims(const int* d) {
  SeqDims r;
  r.B = d[0]; r.T = d[1]; r.steps = d[2]; r.N = d[3]; r.E = d[4]; r.F = d[5]; r.H = d[6];
  r.V = d[7]; r.has_features = d[8]; r.cell = d[9];
  return r;
}

extern "C" {

const char* capnet_last_error(void) { return last_error(); }
int capnet_abi_version(void) { return 1; }

int capnet_sgemm(int transA, int transB, int M, int N, int K, const float* A, long lda,
                 const float* B, long ldb, float* C, long ldc, const float* bias, int accumulate,
                 int batch, long strideA, long strideB, long strideC, long strideBias,
                 int force_tile, capnet_stream_t stream) {
  return sgemm(transA != 0, transB != 0, M, N, K, A, lda, B, ldb, C, ldc, bias, accumulate, batch,
               strideA, strideB, strideC, strideBias, force_tile, S(stream));
}

int capnet_sgemm_b3(int transA, int transB, int M, int N, int K, const float* A, long lda, const float* B, long ldb,
                    float* C, long ldc, const float* bias, int accumulate, int batch, long strideA, long strideB,
                    long strideC, long strideBias, float* ws, size_t ws_floats, capnet_stream_t stream) {
  return sgemm_b3(transA != 0, transB != 0, M, N, K, A, lda, B, ldb, C, ldc, bias, accumulate, batch, strideA, strideB,
                  strideC, strideBias, S(stream), ws, ws_floats);
}
int capnet_sgemm_b3_eligible(int transA, int transB, int M, int N, int K, const float* A, long lda, const float* B, long ldb,
                             const float* C, long ldc, const float* bias, int batch, long strideA, long strideB,
                             long strideC, long strideBias) {
  return sgemm_b3_eligible(transA != 0, transB != 0, M, N, K, A, lda, B, ldb, C, ldc, bias, batch, strideA, strideB, strideC,
                           strideBias) ? 1 : 0;
}

int capnet_sgemm_nt_dma_eligible(int M, int N, int K, const float* A, long lda, const float* B,
                                 long ldb, const float* C, long ldc) {
  return sgemm_nt_dma_eligible(M, N, K, A, lda, B, ldb, C, ldc) ? 1 : 0;
}
int capnet_sgemm_nt_dma(int M, int N, int K, const float* A, long lda, const float* B, float* C,
                        const float* bias, capnet_stream_t stream) {
  CAPNET_REQUIRE(A && B && C, "sgemm_nt_dma: null operand");
  return sgemm_nt_dma(M, N, K, A, lda, B, C, bias, S(stream));
}
int capnet_conv1x1_tiles_m(long M) { return conv1x1_tiles_m(M); }
int capnet_conv1x1_fwd_dma(const float* x, long sxb, long sxh, long sxw, const float* w_oi, float* y,
                           float* part_sum, float* part_sq, int B, int H, int W, int Cin, int Cout,
                           int stride, const float* out_scale, const float* out_shift, const float* res,
                           int relu_out, capnet_stream_t stream) {
  return conv1x1_fwd_dma(x, sxb, sxh, sxw, w_oi, y, part_sum, part_sq, B, H, W, Cin, Cout, stride,
                         S(stream), out_scale, out_shift, res, relu_out);
}

int capnet_conv1x1_fwd_areg(const float* x, const unsigned* image, int bn, float* y, const float* in_scale,
                            const float* in_shift, int relu_in, float* part_sum, float* part_sq, long M, int Cin, int Cout,
                            int in_exp, capnet_stream_t stream) {
  return conv1x1_fwd_areg(x, image, bn, y, in_scale, in_shift, relu_in, part_sum, part_sq, M, Cin, Cout, in_exp, S(stream));
}
size_t capnet_conv1x1_f16x3_weight_words(int Cin, int Cout) { return conv1x1_f16x3_weight_words(Cin, Cout); }
int capnet_conv1x1_f16x3_bn(long M, int Cout) { return conv1x1_f16x3_bn(M, Cout); }
int capnet_conv1x1_f16x3_pack(const float* w_oi, unsigned* image, int Cout, int Cin, int bn,
                              capnet_stream_t stream) {
  return conv1x1_f16x3_pack(w_oi, image, Cout, Cin, bn, S(stream));
}
int capnet_conv1x1_fwd_f16x3(const float* x, long sxb, long sxh, long sxw, const unsigned* image, int bn,
                             float* y, const float* in_scale, const float* in_shift, int relu_in,
                             float* part_sum, float* part_sq, int B, int H, int W, int Cin, int Cout,
                             int stride, const float* out_scale, const float* out_shift,
                             const float* res, int relu_out, capnet_stream_t stream) {
  CAPNET_REQUIRE(conv1x1_f16x3_eligible(x, sxb, sxh, sxw, 1, B, H, W, Cin, Cout, stride, in_scale, in_shift),
                 "capnet_conv1x1_fwd_f16x3: operands not eligible (Cin %% 64, Cout %% 64, 16-B aligned rows, Cin <= 512 with a folded input)");
  return conv1x1_fwd_f16x3(x, sxb, sxh, sxw, image, bn, y, in_scale, in_shift, relu_in, part_sum, part_sq, B,
                           H, W, Cin, Cout, stride, S(stream), out_scale, out_shift, res, relu_out);
}

size_t capnet_conv_f16x3_weight_words(int Cin, int Cout, int k) { return conv_f16x3_weight_words(Cin, Cout, k); }
int capnet_conv_f16x3_pack(const float* w_oihw, unsigned* image, int Cout, int Cin, int k, int bn,
                           capnet_stream_t stream) {
  return conv_f16x3_pack(w_oihw, image, Cout, Cin, k, bn, S(stream));
}
int capnet_conv2d_fwd_f16x3(const float* x, long sxb, long sxh, long sxw, const unsigned* image, int bn, float* y,
                            const float* in_scale, const float* in_shift, int relu_in, float* part_sum,
                            float* part_sq, int B, int H, int W, int Cin, int Cout, int k, int stride, int pad,
                            const float* out_scale, const float* out_shift, const float* res, int relu_out,
                            capnet_stream_t stream) {
  CAPNET_REQUIRE(conv_f16x3_eligible(x, sxb, sxh, sxw, 1, B, H, W, Cin, Cout, k, stride, pad, in_scale, in_shift),
                 "capnet_conv2d_fwd_f16x3: operands not eligible (k = 1 / pad 0 or k = 3 / pad 1, Cin %% 64, Cout %% 64, 16-B aligned rows, Cin <= 512 with a folded input)");
  return conv_fwd_f16x3(x, sxb, sxh, sxw, image, bn, y, in_scale, in_shift, relu_in, part_sum, part_sq, B, H, W,
                        Cin, Cout, k, stride, pad, S(stream), out_scale, out_shift, res, relu_out);
}

int capnet_conv3x3_fwd_patch(const float* x, const unsigned* image, int bn, float* y, const float* in_scale,
                             const float* in_shift, int relu_in, float* part_sum, float* part_sq, int B, int H, int W,
                             int Cin, int Cout, int shared_chip, capnet_stream_t stream) {
  return conv3x3_fwd_patch(x, image, bn, y, in_scale, in_shift, relu_in, part_sum, part_sq, B, H, W, Cin, Cout, S(stream),
                           shared_chip != 0);
}
int capnet_conv1x1_fwd_tail(const float* y3, const float* s1, const float* t1, const float* res, const float* s2,
                            const float* t2, float* tail_out, const unsigned* image, int bn, float* y, float* part_sum,
                            float* part_sq, long M, int Cin, int Cout, capnet_stream_t stream) {
  return conv1x1_fwd_tail(y3, s1, t1, res, s2, t2, tail_out, image, bn, y, part_sum, part_sq, M, Cin, Cout, S(stream));
}
// the same three with the input's power-of-two prescale (capnet.h)
int capnet_conv2d_fwd_f16x3_scaled(const float* x, long sxb, long sxh, long sxw, const unsigned* image, int bn, float* y,
                                   const float* in_scale, const float* in_shift, int relu_in, float* part_sum,
                                   float* part_sq, int B, int H, int W, int Cin, int Cout, int k, int stride, int pad,
                                   int in_exp, capnet_stream_t stream) {
  CAPNET_REQUIRE(conv_f16x3_eligible(x, sxb, sxh, sxw, 1, B, H, W, Cin, Cout, k, stride, pad, in_scale, in_shift),
                 "capnet_conv2d_fwd_f16x3_scaled: operands not eligible");
  return conv_fwd_f16x3(x, sxb, sxh, sxw, image, bn, y, in_scale, in_shift, relu_in, part_sum, part_sq, B, H, W,
                        Cin, Cout, k, stride, pad, S(stream), nullptr, nullptr, nullptr, 0, in_exp);
}
int capnet_conv3x3_fwd_patch_scaled(const float* x, const unsigned* image, int bn, float* y, const float* in_scale,
                                    const float* in_shift, int relu_in, float* part_sum, float* part_sq, int B, int H,
                                    int W, int Cin, int Cout, int shared_chip, int in_exp, capnet_stream_t stream) {
  return conv3x3_fwd_patch(x, image, bn, y, in_scale, in_shift, relu_in, part_sum, part_sq, B, H, W, Cin, Cout, S(stream),
                           shared_chip != 0, in_exp);
}
int capnet_conv1x1_fwd_tail_scaled(const float* y3, const float* s1, const float* t1, const float* res, const float* s2,
                                   const float* t2, float* tail_out, const unsigned* image, int bn, float* y,
                                   float* part_sum, float* part_sq, long M, int Cin, int Cout, int in_exp,
                                   capnet_stream_t stream) {
  return conv1x1_fwd_tail(y3, s1, t1, res, s2, t2, tail_out, image, bn, y, part_sum, part_sq, M, Cin, Cout, S(stream), in_exp);
}
size_t capnet_conv_stem_f16x3_weight_words(void) { return conv_stem_f16x3_weight_words(); }
int capnet_conv_stem_f16x3_part_rows(int B, int H, int W) { return conv_stem_f16x3_part_rows(B, H, W); }
int capnet_conv_stem_f16x3_pack(const float* w_oihw, unsigned* image, capnet_stream_t stream) {
  return conv_stem_f16x3_pack(w_oihw, image, S(stream));
}
int capnet_conv_stem_fwd_f16x3(const float* x, long sxb, long sxc, long sxh, const unsigned* image, float* y,
                               float* part_sum, float* part_sq, int B, int H, int W, capnet_stream_t stream) {
  return conv_stem_fwd_f16x3(x, sxb, sxc, sxh, image, y, part_sum, part_sq, B, H, W, S(stream));
}

int capnet_sgemm_splitk(int transA, int transB, int M, int N, int K, const float* A, long lda,
                        const float* B, long ldb, float* C, long ldc, const float* bias,
                        int accumulate, float* workspace, size_t workspace_floats,
                        capnet_stream_t stream) {
  CAPNET_REQUIRE(M >= 0 && N >= 0 && K >= 0, "sgemm_splitk: negative dimension");
  return sgemm_splitk(transA != 0, transB != 0, M, N, K, A, lda, B, ldb, C, ldc, bias, accumulate,
                      workspace, workspace_floats, S(stream));
}
int capnet_sgemm_splitk_fused(int transA, int transB, int M, int N, int K, const float* A, long lda,
                              const float* B, long ldb, float* C, long ldc, const float* bias,
                              int accumulate, float* workspace, size_t workspace_floats, int* counters,
                              size_t n_counters, capnet_stream_t stream) {
  CAPNET_REQUIRE(M >= 0 && N >= 0 && K >= 0, "sgemm_splitk_fused: negative dimension");
  return sgemm_splitk(transA != 0, transB != 0, M, N, K, A, lda, B, ldb, C, ldc, bias, accumulate,
                      workspace, workspace_floats, S(stream), counters, n_counters);
}
int capnet_colsum(const float* x, long ld, int rows, int C, float* out, int accumulate,
                  capnet_stream_t stream) {
  return colsum(x, ld, rows, C, out, accumulate, S(stream));
}

int capnet_argmax_rows(const float* x, int rows, int ld, int V, int* out, capnet_stream_t stream) {
  CAPNET_REQUIRE(x && out && ld >= V && V > 0, "argmax_rows: bad argument");
  return argmax_rows(x, rows, ld, V, out, S(stream));
}

int capnet_trunk_create(int batch, int height, int width, capnet_trunk_t** out) {
  return trunk_create(batch, height, width, reinterpret_cast<Trunk**>(out));
}
void capnet_trunk_destroy(capnet_trunk_t* t) { trunk_destroy(reinterpret_cast<Trunk*>(t)); }
size_t capnet_trunk_workspace_bytes(const capnet_trunk_t* t) {
  return trunk_workspace_bytes(reinterpret_cast<const Trunk*>(t));
}
int capnet_trunk_num_convs(const capnet_trunk_t* t) {
  return trunk_num_convs(reinterpret_cast<const Trunk*>(t));
}
int capnet_trunk_final_side(const capnet_trunk_t* t) {
  return trunk_final_side(reinterpret_cast<const Trunk*>(t));
}
double capnet_trunk_flops(const capnet_trunk_t* t) {
  return trunk_flops(reinterpret_cast<const Trunk*>(t));
}
int capnet_trunk_conv_shape(const capnet_trunk_t* t, int i, int* cout, int* cin, int* ksize,
                            int* stride, int* row_stride) {
  CAPNET_REQUIRE(t && cout && cin && ksize && stride && row_stride, "trunk_conv_shape: null");
  return trunk_conv_shape(reinterpret_cast<const Trunk*>(t), i, cout, cin, ksize, stride,
                          row_stride);
}
int capnet_trunk_forward(const capnet_trunk_t* t, const float* images_nchw,
                         const float* const* w_packed, const float* const* bn_weight,
                         const float* const* bn_bias, float* const* bn_running_mean,
                         float* const* bn_running_var, int train, float momentum, float eps,
                         void* workspace, float* out_pooled, float* out_map,
                         const int* input_exponents, int* err_flag, capnet_stream_t stream) {
  return trunk_forward(reinterpret_cast<Trunk*>(const_cast<capnet_trunk_t*>(t)), images_nchw, w_packed, bn_weight,
                       bn_bias, bn_running_mean, bn_running_var, train, momentum, eps,
                       reinterpret_cast<float*>(workspace), out_pooled, out_map, input_exponents, err_flag, S(stream));
}
int capnet_trunk_set_tail_balance(const capnet_trunk_t* t, int on) {
  return trunk_set_tail_balance(reinterpret_cast<Trunk*>(const_cast<capnet_trunk_t*>(t)), on);
}
int capnet_trunk_update_running(const capnet_trunk_t* t, const void* workspace,
                                float* const* bn_running_mean, float* const* bn_running_var,
                                float momentum, capnet_stream_t stream) {
  return trunk_update_running(reinterpret_cast<Trunk*>(const_cast<capnet_trunk_t*>(t)),
                              reinterpret_cast<const float*>(workspace), bn_running_mean, bn_running_var,
                              momentum, S(stream));
}
double capnet_trunk_conv_flops(const capnet_trunk_t* t, int i) {
  return trunk_conv_flops(reinterpret_cast<const Trunk*>(t), i);
}
int capnet_trunk_conv_kmajor(const capnet_trunk_t* t, int i) {
  return trunk_conv_kmajor(reinterpret_cast<const Trunk*>(t), i);
}
int capnet_trunk_conv_tile_n(const capnet_trunk_t* t, int i) { return trunk_conv_tile_n(reinterpret_cast<const Trunk*>(t), i); }
int capnet_pack_conv_weight_kmajor(const float* w_oihw, float* out, int Cout, int Cin, int KH,
                                   int KW, int k_rows, capnet_stream_t stream) {
  return pack_conv_weight_kmajor(w_oihw, out, Cout, Cin, KH, KW, k_rows, S(stream));
}
int capnet_conv2d_fwd_kmajor(const float* x, long sxb, long sxh, long sxw, const float* w_kmajor,
                             int k_rows, float* y, const float* in_scale, const float* in_shift,
                             int relu_in, float* part_sum, float* part_sq, int B, int H, int W,
                             int Cin, int Cout, int KH, int KW, int stride, int pad, int tile,
                             float* slabs, capnet_stream_t stream) {
  return conv2d_fwd_v2(x, sxb, sxh, sxw, w_kmajor, k_rows, y, in_scale, in_shift, relu_in, part_sum,
                       part_sq, B, H, W, Cin, Cout, KH, KW, stride, pad, tile, slabs, S(stream));
}
size_t capnet_conv_kmajor_slab_floats(int M, int Cout, int k_rows, int tile) {
  return conv_v2_slab_floats(M, Cout, k_rows, tile);
}
int capnet_pack_conv_weight(const float* w_oihw, float* out, int Cout, int Cin, int KH, int KW,
                            int row_stride, capnet_stream_t stream) {
  return pack_conv_weight(w_oihw, out, Cout, Cin, KH, KW, row_stride, S(stream));
}
int capnet_adaptive_pool_replicate(const float* x, float* out, int B, int side, int out_side,
                                   int C, capnet_stream_t stream) {
  return adaptive_pool_replicate(x, out, B, side, out_side, C, S(stream));
}

int capnet_conv2d_fwd(const float* x, long sxb, long sxh, long sxw, long sxc,
                      const float* w_packed, int row_stride, float* y, const float* in_scale,
                      const float* in_shift, int relu_in, float* part_sum, float* part_sq, int B,
                      int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad,
                      int tile, capnet_stream_t stream) {
  return conv2d_fwd(x, sxb, sxh, sxw, sxc, w_packed, row_stride, y, in_scale, in_shift, relu_in,
                    part_sum, part_sq, B, H, W, Cin, Cout, KH, KW, stride, pad, tile, S(stream));
}
int capnet_conv_tiles_m(int M, int Cout, int tile) {
  if (tile == 0) tile = conv_auto_tile(M, Cout);
  return conv_tiles_m(M, tile);
}
void capnet_conv_kmajor_plan(int M, int Cout, int k_rows, int tile, int* out5) {
  conv_v2_plan(M, Cout, k_rows, tile, out5);
}
int capnet_conv_kmajor_tiles_m(int M, int Cout, int k_rows, int tile) {
  int plan[5];
  conv_v2_plan(M, Cout, k_rows, tile, plan);   // normalises the tile exactly as the launcher does
  return conv_tiles_m(M, plan[0]);
}
int capnet_bn_finalize(const float* part_sum, const float* part_sq, int tiles, int C, long count,
                       const float* gamma, const float* beta, float* running_mean,
                       float* running_var, float momentum, float eps, float* scale, float* shift,
                       capnet_stream_t stream) {
  return bn_finalize(part_sum, part_sq, tiles, C, count, gamma, beta, running_mean, running_var,
                     momentum, eps, scale, shift, S(stream));
}
int capnet_bn_add_relu(const float* y, const float* s1, const float* t1, const float* res,
                       const float* s2, const float* t2, float* out, long rows, int C,
                       capnet_stream_t stream) {
  return bn_add_relu(y, s1, t1, res, s2, t2, out, rows, C, S(stream));
}
int capnet_bn_relu_maxpool(const float* y, const float* scale, const float* shift, float* out,
                           int B, int H, int W, int C, capnet_stream_t stream) {
  return bn_relu_maxpool(y, scale, shift, out, B, H, W, C, S(stream));
}
int capnet_global_avgpool(const float* x, float* out, int B, int HW, int C,
                          capnet_stream_t stream) {
  return global_avgpool(x, out, B, HW, C, S(stream));
}

int capnet_bn1d_fwd(const float* x, int B, int C, const float* gamma, const float* beta,
                    float* running_mean, float* running_var, int train, float momentum, float eps,
                    float* y, float* save_mean, float* save_invstd, capnet_stream_t stream) {
  return bn1d_fwd(x, B, C, gamma, beta, running_mean, running_var, train, momentum, eps, y,
                  save_mean, save_invstd, S(stream));
}
int capnet_bn1d_bwd(const float* dy, const float* x, int B, int C, const float* gamma,
                    const float* save_mean, const float* save_invstd, float* dx, float* dgamma,
                    float* dbeta, capnet_stream_t stream) {
  return bn1d_bwd(dy, x, B, C, gamma, save_mean, save_invstd, dx, dgamma, dbeta, S(stream));
}

int capnet_embedding_fwd(const long long* idx, int n, const float* emb, int E, int V, float* out,
                         int* err_flag, capnet_stream_t stream) {
  return embedding_fwd(idx, n, emb, E, V, out, err_flag, S(stream));
}
int capnet_lstm_pointwise_fwd(float* pre, const float* c_prev, float* c_out, float* h_out, int b,
                              int H, int cell, capnet_stream_t stream) {
  CAPNET_REQUIRE(pre && c_out && h_out && b >= 0 && H > 0, "lstm_pointwise_fwd: bad argument");
  if (cell == kCellFactored)
    return lstm_pointwise_fwd(pre, 4L * H, c_prev, c_out, h_out, b, H, 0, 1, 2, 3, 0, S(stream));
  CAPNET_REQUIRE(cell == kCellLSTM, "lstm_pointwise_fwd: unknown cell %d", cell);
  return lstm_pointwise_fwd(pre, 4L * H, c_prev, c_out, h_out, b, H, 0, 1, 3, 2, 1, S(stream));
}

int capnet_lstm_pointwise_bwd(const float* gates, const float* c, const float* c_prev, const float* dh, float* dc_io,
                              float* dpre, int b, int H, int cell, capnet_stream_t stream) {
  CAPNET_REQUIRE(gates && c && dh && dc_io && dpre && b >= 0 && H > 0, "lstm_pointwise_bwd: bad argument");
  CAPNET_REQUIRE(cell == kCellFactored || cell == kCellLSTM, "lstm_pointwise_bwd: unknown cell %d", cell);
  const bool f = cell == kCellFactored;
  // (no separate recurrent dh: the caller's dh is the whole gradient of h; dc_io: in dL/dc, out dL/dc_prev)
  return lstm_pointwise_bwd(gates, 4L * H, c, c_prev, dh, nullptr, dc_io, dpre, 4L * H, b, b, H, 0, 1, f ? 2 : 3, f ? 3 : 2,
                            f ? 0 : 1, S(stream), 0, 0);
}
size_t capnet_lstm_wfrag_floats(int H) { return lstm_wfrag_floats(H); }
int capnet_lstm_pack_wfrag(const float* w_cat, float* w_frag, int H, int cell, capnet_stream_t stream) {
  CAPNET_REQUIRE(cell == kCellFactored || cell == kCellLSTM, "lstm_pack_wfrag: unknown cell %d", cell);
  return cell == kCellFactored ? lstm_pack_wfrag(w_cat, w_frag, H, 0, 1, 2, 3, S(stream))
                               : lstm_pack_wfrag(w_cat, w_frag, H, 0, 1, 3, 2, S(stream));
}
int capnet_lstm_step_fused(const float* h_prev, const float* w_cat, float* gates, long ldg,
                           const float* c_prev, float* c_out, float* h_out, int b, int H, int cell,
                           capnet_stream_t stream) {
  CAPNET_REQUIRE(cell == kCellFactored || cell == kCellLSTM, "lstm_step_fused: unknown cell %d", cell);
  if (cell == kCellFactored)
    return lstm_step_fused(h_prev, w_cat, gates, ldg, c_prev, c_out, h_out, b, H, 0, 1, 2, 3, 0, S(stream));
  return lstm_step_fused(h_prev, w_cat, gates, ldg, c_prev, c_out, h_out, b, H, 0, 1, 3, 2, 1, S(stream));
}
int capnet_lstm_step_fused_stamped(const float* h_prev, const float* w_frag, float* gates, long ldg,
                                   const float* c_prev, float* c_out, float* h_out, int b, int H,
                                   unsigned long long* stamps, capnet_stream_t stream) {
  return lstm_step_fused(h_prev, w_frag, gates, ldg, c_prev, c_out, h_out, b, H, 0, 1, 2, 3, 0,
                         S(stream), stamps);
}
int capnet_lstm_step_fused_supported(int b, int H) { return lstm_step_fused_supported(b, H) ? 1 : 0; }

int capnet_lstm_persist_supported(int b, int H) { return lstm_persist_supported(b, H) ? 1 : 0; }
int capnet_lstm_persist_set_mode(int mode) { return lstm_persist_set_mode(mode); }
size_t capnet_lstm_persist_w_floats(void) { return lstm_persist_w_floats(); }
size_t capnet_lstm_persist_ctl_ints(void) { return lstm_persist_ctl_ints(); }
int capnet_lstm_persist_pack(const float* w_cat, float* w_img, int cell, capnet_stream_t stream) {
  CAPNET_REQUIRE(cell == kCellFactored || cell == kCellLSTM, "lstm_persist_pack: unknown cell %d", cell);
  return cell == kCellFactored ? lstm_persist_pack(w_cat, w_img, 0, 1, 2, 3, S(stream))
                               : lstm_persist_pack(w_cat, w_img, 0, 1, 3, 2, S(stream));
}
int capnet_lstm_persist_run(const float* w_img, float* gates, float* cell_states, float* hiddens,
                            const int* batch_sizes, int t0, int t1, int H, int cell, int segment,
                            int* ctl, int* err_flag, unsigned long long* stamps,
                            capnet_stream_t stream) {
  CAPNET_REQUIRE(cell == kCellFactored || cell == kCellLSTM, "lstm_persist_run: unknown cell %d", cell);
  CAPNET_REQUIRE(batch_sizes && t1 > 0 && t1 <= kMaxSteps, "lstm_persist_run: bad steps");
  int off[kMaxSteps + 1];
  off[0] = 0;
  for (int t = 0; t < t1; ++t) off[t + 1] = off[t] + batch_sizes[t];
  const bool f = cell == kCellFactored;
  return lstm_persist_run(w_img, gates, cell_states, hiddens, off, batch_sizes, t0, t1, H, 0, 1,
                          f ? 2 : 3, f ? 3 : 2, f ? 0 : 1, segment, ctl, err_flag, S(stream), stamps);
}

size_t capnet_seq_saved_floats(const int* dims) { return seq_saved_floats(to_dims(dims)); }
size_t capnet_seq_saved_ints(const int* dims) { return seq_saved_ints(to_dims(dims)); }
size_t capnet_seq_fwd_scratch_floats(const int* dims) { return seq_fwd_scratch_floats(to_dims(dims)); }
size_t capnet_seq_bwd_scratch_floats(const int* dims) { return seq_bwd_scratch_floats(to_dims(dims)); }

int capnet_seq_forward(const int* dims, const int* batch_sizes, const unsigned char* tf_mask,
                       const long long* captions, const float* features, const float* emb,
                       const float* const* weights, const float* Cw, const float* Cb,
                       float dropout_p, unsigned long long seed, int training, float* saved,
                       int* saved_i, float* scratch, float* hiddens, int* err_flag,
                       capnet_stream_t stream) {
  CAPNET_REQUIRE(dims && weights, "seq_forward: null dims/weights");
  SeqWeights w;
  for (int g = 0; g < 4; ++g) {
    w.Vw[g] = weights[0 + g];  w.Vb[g] = weights[4 + g];
    w.Sw[g] = weights[8 + g];  w.Sb[g] = weights[12 + g];
    w.Uw[g] = weights[16 + g]; w.Ub[g] = weights[20 + g];
    w.Ww[g] = weights[24 + g]; w.Wb[g] = weights[28 + g];
  }
  const SeqDims d = to_dims(dims);
  if (d.cell == kCellFactored) {
    for (int i = 0; i < 32; ++i) CAPNET_REQUIRE(weights[i], "seq_forward: weight %d is null", i);
  } else {
    CAPNET_REQUIRE(d.cell == kCellLSTM, "seq_forward: unknown cell %d", d.cell);
    CAPNET_REQUIRE(w.Vw[0] && w.Vb[0] && w.Ww[0] && w.Wb[0], "seq_forward: LSTM weight is null");
  }
  return seq_forward(d, batch_sizes, tf_mask, captions, features, emb, w, Cw, Cb, dropout_p, seed,
                     training, saved, saved_i, scratch, hiddens, err_flag, S(stream));
}

int capnet_seq_backward(const int* dims, const int* batch_sizes, const float* d_hiddens,
                        const float* hiddens, const float* saved, const int* saved_i,
                        float* scratch, float* const* grads, float dropout_p,
                        unsigned long long seed, int training, capnet_stream_t stream) {
  CAPNET_REQUIRE(dims && grads, "seq_backward: null dims/grads");
  SeqGrads g;
  g.dVcat = grads[0]; g.dbV = grads[1]; g.dScat = grads[2]; g.dbS = grads[3]; g.dUcat = grads[4];
  g.dbUW = grads[5]; g.dWcat = grads[6]; g.dEmb = grads[7]; g.dFeat = grads[8];
  return seq_backward(to_dims(dims), batch_sizes, d_hiddens, hiddens, saved, saved_i, scratch, g,
                      dropout_p, seed, training, S(stream));
}

static void to_weights(const float* const* weights, SeqWeights& w) {
  for (int g = 0; g < 4; ++g) {
    w.Vw[g] = weights[0 + g];  w.Vb[g] = weights[4 + g];
    w.Sw[g] = weights[8 + g];  w.Sb[g] = weights[12 + g];
    w.Uw[g] = weights[16 + g]; w.Ub[g] = weights[20 + g];
    w.Ww[g] = weights[24 + g]; w.Wb[g] = weights[28 + g];
  }
}

int capnet_seq_forward_stacked(const int* dims, int nlayers, const int* batch_sizes, const unsigned char* tf_mask,
                               const long long* captions, const float* features, const float* emb,
                               const float* const* weights, const float* Cw, const float* Cb, float dropout_p,
                               unsigned long long seed, int training, float* const* saved, int* const* saved_i,
                               float* scratch, float* const* hiddens, int* err_flag, capnet_stream_t stream) {
  CAPNET_REQUIRE(dims && weights && nlayers >= 1 && nlayers <= 8, "seq_forward_stacked: null dims / weights or layers %d", nlayers);
  const SeqDims d = to_dims(dims);
  CAPNET_REQUIRE(d.cell == kCellFactored, "seq_forward_stacked: the factored cell only");
  SeqWeights w[8];
  for (int l = 0; l < nlayers; ++l) {
    for (int i = 0; i < 32; ++i) CAPNET_REQUIRE(weights[32 * l + i], "seq_forward_stacked: weight %d of layer %d is null", i, l);
    to_weights(weights + 32 * l, w[l]);
  }
  return seq_forward_stacked(d, nlayers, batch_sizes, tf_mask, captions, features, emb, w, Cw, Cb, dropout_p, seed, training,
                             saved, saved_i, scratch, hiddens, err_flag, S(stream));
}

int capnet_seq_backward_stacked(const int* dims, int nlayers, const int* batch_sizes, const float* d_hiddens,
                                const float* const* hiddens, const float* const* saved, const int* const* saved_i,
                                float* scratch, float* const* dh_work, float* const* grads, float dropout_p,
                                unsigned long long seed, int training, capnet_stream_t stream) {
  CAPNET_REQUIRE(dims && grads && nlayers >= 1 && nlayers <= 8, "seq_backward_stacked: null dims / grads or layers %d", nlayers);
  SeqGrads g[8];
  for (int l = 0; l < nlayers; ++l) {
    float* const* q = grads + 9 * l;
    g[l].dVcat = q[0]; g[l].dbV = q[1]; g[l].dScat = q[2]; g[l].dbS = q[3]; g[l].dUcat = q[4];
    g[l].dbUW = q[5]; g[l].dWcat = q[6]; g[l].dEmb = q[7]; g[l].dFeat = q[8];
  }
  return seq_backward_stacked(to_dims(dims), nlayers, batch_sizes, d_hiddens, hiddens, saved, saved_i, scratch, dh_work, g,
                              dropout_p, seed, training, S(stream));
}

static AttDims to_adims(const int* d) {
  AttDims r;
  r.B = d[0]; r.T = d[1]; r.steps = d[2]; r.N = d[3]; r.E = d[4]; r.F = d[5]; r.H = d[6];
  r.V = d[7]; r.A = d[8]; r.P = d[9]; r.C = d[10]; r.cell = d[11];
  return r;
}
static int to_aweights(const float* const* p, AttWeights* w, int cell) {
  CAPNET_REQUIRE(p != nullptr, "att decoder: null weight table");
  CAPNET_REQUIRE(cell == kCellFactored || cell == kCellLSTM, "att decoder: unknown cell %d", cell);
  for (int i = 0; i < 44; ++i) {
    // nn.LSTMCell: only slot 0 of the V (weight_ih, bias_ih) and W (weight_hh, bias_hh) groups
    const bool used = cell == kCellFactored || i >= 32 || i == 0 || i == 4 || i == 24 || i == 28;
    CAPNET_REQUIRE(!used || p[i] != nullptr, "att decoder: weight %d is null", i);
  }
  for (int g = 0; g < 4; ++g) {
    w->Vw[g] = p[0 + g];  w->Vb[g] = p[4 + g];
    w->Sw[g] = p[8 + g];  w->Sb[g] = p[12 + g];
    w->Uw[g] = p[16 + g]; w->Ub[g] = p[20 + g];
    w->Ww[g] = p[24 + g]; w->Wb[g] = p[28 + g];
  }
  w->init_h_w = p[32]; w->init_h_b = p[33]; w->init_c_w = p[34]; w->init_c_b = p[35];
  w->enc_att_w = p[36]; w->enc_att_b = p[37]; w->dec_att_w = p[38]; w->dec_att_b = p[39];
  w->full_att_w = p[40]; w->full_att_b = p[41]; w->f_beta_w = p[42]; w->f_beta_b = p[43];
  return kOk;
}
int capnet_att_set_chain_mode(int mode) { return att_set_chain_mode(mode); }
size_t capnet_att_saved_floats(const int* dims) { return att_saved_floats(to_adims(dims)); }
size_t capnet_att_saved_ints(const int* dims) { return att_saved_ints(to_adims(dims)); }
size_t capnet_att_fwd_scratch_floats(const int* dims) { return att_fwd_scratch_floats(to_adims(dims)); }
size_t capnet_att_bwd_scratch_floats(const int* dims) { return att_bwd_scratch_floats(to_adims(dims)); }

int capnet_beam_topk_batched(const float* logits, long ld, int V, const float* prev_scores, const int* meta, int n,
                             float* top_scores, long long* top_index, capnet_stream_t stream) {
  return beam_topk_batched(logits, ld, V, prev_scores, meta, n, top_scores, top_index, S(stream));
}
int capnet_beam_topk(const float* logits, long ld, int rows, int V, const float* prev_scores, int k,
                     float* top_scores, long long* top_index, capnet_stream_t stream) {
  return beam_topk(logits, ld, rows, V, prev_scores, k, top_scores, top_index, S(stream));
}
int capnet_att_step_fwd(const float* att1, const float* feat, const float* att2, float* gate_io,
                        long ldz, const float* w_full, const float* b_full, int rows, int P, int A,
                        int C, float* alpha_out, float* alphas_bt, int steps, int t, float* awe_out,
                        float* xa_out, long ldx, float* scores_ws, capnet_stream_t stream) {
  CAPNET_REQUIRE(steps > 0 && t >= 0 && t < steps, "att_step_fwd: t=%d steps=%d", t, steps);
  return att_step_fwd(att1, feat, att2, gate_io, ldz, w_full, b_full, rows, P, A, C, alpha_out,
                      alphas_bt, steps, t, awe_out, xa_out, ldx, scores_ws, S(stream));
}
int capnet_att_seq_forward(const int* dims, const int* batch_sizes, const unsigned char* tf_mask,
                           const long long* captions, const float* features, const float* emb,
                           const float* const* weights, const float* Cw, const float* Cb,
                           float dropout_p, unsigned long long seed, int training, float* saved,
                           int* saved_i, float* scratch, float* hiddens, float* alphas,
                           int* err_flag, capnet_stream_t stream) {
  CAPNET_REQUIRE(dims != nullptr, "att_seq_forward: null dims");
  AttWeights w;
  int rc = to_aweights(weights, &w, dims ? dims[11] : 0);
  if (rc) return rc;
  return att_seq_forward(to_adims(dims), batch_sizes, tf_mask, captions, features, emb, w, Cw, Cb,
                         dropout_p, seed, training, saved, saved_i, scratch, hiddens, alphas,
                         err_flag, S(stream));
}

int capnet_att_seq_backward(const int* dims, const int* batch_sizes, const float* d_hiddens,
                            const float* d_alphas, const float* hiddens, const float* features,
                            const float* const* weights, const float* saved, const int* saved_i,
                            float* scratch, float* const* grads, float dropout_p,
                            unsigned long long seed, int training, capnet_stream_t stream) {
  CAPNET_REQUIRE(dims && grads, "att_seq_backward: null dims/grads");
  AttWeights w;
  int rc = to_aweights(weights, &w, dims ? dims[11] : 0);
  if (rc) return rc;
  AttGrads g;
  g.dVcat = grads[0]; g.dbV = grads[1]; g.dScat = grads[2]; g.dbS = grads[3]; g.dUcat = grads[4];
  g.dWz = grads[5]; g.dbz = grads[6]; g.dWe = grads[7]; g.dbe = grads[8]; g.dwf = grads[9];
  g.dbf = grads[10]; g.dWih = grads[11]; g.dbih = grads[12]; g.dWic = grads[13]; g.dbic = grads[14];
  g.dEmb = grads[15];
  return att_seq_backward(to_adims(dims), batch_sizes, d_hiddens, d_alphas, hiddens, features, w,
                          saved, saved_i, scratch, g, dropout_p, seed, training, S(stream));
}

int capnet_xent_fwd(const float* logits, long ld, int N, int V, const long long* targets,
                    float* lse, float* row_loss, float* loss, int* err_flag,
                    capnet_stream_t stream) {
  return xent_fwd(logits, ld, N, V, targets, lse, row_loss, loss, err_flag, S(stream));
}
int capnet_resize_u8(const unsigned char* src, int Hs, int Ws, unsigned char* tmp, unsigned char* dst,
                     int Ho, int Wo, const int* bounds_h, const int* coef_h, int kmax_h,
                     const int* bounds_v, const int* coef_v, int kmax_v, capnet_stream_t stream) {
  return resize_u8(src, Hs, Ws, tmp, dst, Ho, Wo, bounds_h, coef_h, kmax_h, bounds_v, coef_v, kmax_v,
                   S(stream));
}
int capnet_crop_flip_normalize(const unsigned char* src, int B, int Hs, int Ws, const int* params,
                               float* dst, int Hc, int Wc, const float* mean, const float* stdv,
                               capnet_stream_t stream) {
  return crop_flip_normalize(src, B, Hs, Ws, params, dst, Hc, Wc, mean, stdv, S(stream));
}
int capnet_topk_correct(const float* logits, long ld, int N, int V, const long long* targets, int k,
                        int* count, int* err_flag, capnet_stream_t stream) {
  return topk_correct(logits, ld, N, V, targets, k, count, err_flag, S(stream));
}
int capnet_xent_bwd(const float* logits, long ld, int N, int V, const long long* targets,
                    const float* lse, const float* grad_out, float* dlogits, long ldd,
                    capnet_stream_t stream) {
  return xent_bwd(logits, ld, N, V, targets, lse, grad_out, dlogits, ldd, S(stream));
}
int capnet_att_loss_fwd(const float* nll, const float* alphas, int B, int steps, int P, float alpha_c,
                        float* colsum, float* out, capnet_stream_t stream) {
  return att_loss_fwd(nll, alphas, B, steps, P, alpha_c, colsum, out, S(stream));
}
int capnet_att_loss_bwd(const float* gout, const float* colsum, int B, int steps, int P, float alpha_c,
                        float* dalphas, capnet_stream_t stream) {
  return att_loss_bwd(gout, colsum, B, steps, P, alpha_c, dalphas, S(stream));
}

int capnet_clamp_adam(int n, float* const* params, float* const* grads, float* const* exp_avg,
                      float* const* exp_avg_sq, const long* numel, const int* step, float lr,
                      float beta1, float beta2, float eps, float clip, int write_grad,
                      const int* skip_flag, capnet_stream_t stream) {
  return clamp_adam(n, params, grads, exp_avg, exp_avg_sq, numel, step, lr, beta1, beta2, eps, clip,
                    write_grad, skip_flag, S(stream));
}

int capnet_trunk_set_timing(capnet_trunk_t* t, int enable) {
  return trunk_set_timing(reinterpret_cast<Trunk*>(t), enable);
}
int capnet_trunk_time_next_pass(capnet_trunk_t* t) { return trunk_time_next_pass(reinterpret_cast<Trunk*>(t)); }
int capnet_trunk_collect_timing(capnet_trunk_t* t, double* conv_ms, long* conv_launches,
                                double* conv_flops) {
  return trunk_collect_timing(reinterpret_cast<Trunk*>(t), conv_ms, conv_launches, conv_flops);
}

int capnet_packed_targets(const long long* captions, int T, int steps, const int* batch_sizes,
                          long long* out, capnet_stream_t stream) {
  CAPNET_REQUIRE(batch_sizes && steps > 0 && steps <= kMaxSteps, "packed_targets: steps %d", steps);
  SeqMeta m;
  m.steps = steps; m.has_features = 0; m.off[0] = 0;
  for (int t = 0; t < steps; ++t) {
    CAPNET_REQUIRE(batch_sizes[t] > 0, "packed_targets: batch_sizes[%d]", t);
    m.off[t + 1] = m.off[t] + batch_sizes[t];
    m.tf[t] = 1;
  }
  m.N = m.off[steps];
  return packed_targets(m, captions, T, out, S(stream));
}

int capnet_pack_tensors(int n, float* const* tensors, const long* numel, float* flat,
                        int direction, float scale, capnet_stream_t stream) {
  return pack_tensors(n, tensors, numel, flat, direction, scale, S(stream));
}

size_t capnet_fused_block_weight_words(int C, int MID, int role) { return fused_block_weight_words(C, MID, role); }
int capnet_fused_block_pack(const float* w, unsigned* img, int C, int MID, int role, capnet_stream_t stream) {
  return fused_block_pack(w, img, C, MID, role, S(stream));
}
size_t capnet_fused_block_stats_floats(long M, int MID) { return fused_block_stats_floats(M, MID); }
int capnet_fused_block_stats(const float* y2, const float* s2, const float* t2, const unsigned* w3img, long M, int MID,
                             int in_exp, const float* gamma, const float* beta, float* running_mean, float* running_var,
                             float momentum, float eps, float* scale, float* shift, float* work, int* err_flag,
                             capnet_stream_t stream) {
  return fused_block_stats(y2, s2, t2, w3img, M, MID, in_exp, gamma, beta, running_mean, running_var, momentum, eps, scale,
                           shift, nullptr, nullptr, work, err_flag, S(stream));
}
int capnet_fused_block_tiles(long M, int MID) { return fused_block_tiles(M, MID); }
int capnet_fused_block_forward(const float* y2, const float* s2, const float* t2, const unsigned* w3img, const float* s3,
                               const float* t3, const float* res, const float* sd, const float* td, float* out,
                               const unsigned* w1img, float* y1, float* part_sum, float* part_sq, long M, int MID, int e3,
                               int e1, int* err_flag, capnet_stream_t stream) {
  return fused_block_forward(y2, s2, t2, w3img, s3, t3, res, sd, td, out, w1img, y1, part_sum, part_sq, M, MID, e3, e1,
                             err_flag, S(stream));
}

int capnet_comm_unique_id(void* id128) { return comm_unique_id(id128); }
int capnet_comm_create(const void* id128, int rank, int world, capnet_comm_t** out) {
  return comm_create(id128, rank, world, reinterpret_cast<Comm**>(out));
}
int capnet_comm_destroy(capnet_comm_t* comm) { return comm_destroy(reinterpret_cast<Comm*>(comm)); }
int capnet_allreduce_grads(capnet_comm_t* comm, float* flat, long count, capnet_stream_t stream) {
  return allreduce_grads(reinterpret_cast<Comm*>(comm), flat, count, S(stream));
}

int capnet_err_word_exchange(int* err_flag, float* slot, int direction, capnet_stream_t stream) {
  return err_word_exchange(err_flag, slot, direction, S(stream));
}

int capnet_count_skipped(const int* err_flag, int* counter, capnet_stream_t stream) {
  return count_skipped(err_flag, counter, S(stream));
}

int capnet_clamp(float* x, long n, float lo, float hi, capnet_stream_t stream) {
  return clamp_inplace(x, n, lo, hi, S(stream));
}

}  // extern "C"
