/* capnet.h -- C ABI of libcapnet_hip.so: the MI355X (gfx950) training-step kernels of the
 * StyleNet / NIC captioning path.
 *
 * The reference (deryrahman/image-caption-emotion-indonesia) has no native/FFI layer: its
 * boundary is the Python class surface of stylenet/model.py, nic/model.py and the loops in
 * stylenet/train_multitask.py. Each entry point below names the torch call(s) of the
 * reference it replaces (file:line under /root/reference). The Python mirror of those classes
 * binds these symbols through ctypes.
 *
 * Conventions
 *   - every function returns 0 on success, a negative status otherwise;
 *     capnet_last_error() returns the message of the calling thread's last failure;
 *   - all `float*` / `long long*` / `int*` data pointers are CALLER-OWNED DEVICE pointers
 *     (tensor.data_ptr()); arrays documented as "host" are ordinary host memory;
 *   - `stream` is a hipStream_t; nothing here allocates device memory, synchronises the
 *     stream or touches the default stream, so every call can be captured in a hipGraph;
 *   - arithmetic is fp32 with fp32 accumulation (v_mfma_f32_32x32x2_f32 == fmaf chain).
 */
#ifndef CAPNET_H_
#define CAPNET_H_

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* capnet_stream_t; /* hipStream_t */
typedef struct capnet_trunk capnet_trunk_t;

#define CAPNET_OK 0
#define CAPNET_ERR_INVALID_ARG (-1)
#define CAPNET_ERR_HIP (-2)

const char* capnet_last_error(void);
int capnet_abi_version(void);

/* ---- dense projections --------------------------------------------------------------
 * C[M x N] (+)= op(A) . op(B) + bias[N], row-major, `batch` independent problems at the given
 * element strides. transA=0: A is [M][K]; 1: [K][M]. transB=0: B is [K][N]; 1: [N][K].
 * Replaces every nn.Linear forward and autograd matmul of the decoders and the encoder head:
 * stylenet/model.py:119-150 (V/S/U/W gates), :189-194 (C), :19,26 (encoder.linear);
 * nic/model.py:52,77,105-113. force_tile: 0 = auto, 64 or 128. */
int capnet_sgemm(int transA, int transB, int M, int N, int K, const float* A, long lda,
                 const float* B, long ldb, float* C, long ldc, const float* bias, int accumulate,
                 int batch, long strideA, long strideB, long strideC, long strideBias,
                 int force_tile, capnet_stream_t stream);
/* The same product on the bf16 matrix cores: each fp32 operand cut into three bf16 pieces (exactly), six piece products
 * per multiply accumulated in fp32 -- fp32-grade results over fp32's whole exponent range, no prescale (csrc/gemm_b3.hip).
 * capnet_sgemm takes this path by itself for large products (CAPNET_NO_B3=1 in the environment keeps everything on the f32
 * MFMA). Eligible: 16-B aligned operands, leading dimensions and batch strides multiples of 4, K % 4 == 0 when an operand
 * is K-contiguous, M % 4 == 0 / N % 4 == 0 when A / B is stored with that dimension contiguous. ws (optional, batch 1):
 * ws_floats floats for split-K partials -- products of few 128 x 128 tiles and a long K are cut over the chip. */
int capnet_sgemm_b3(int transA, int transB, int M, int N, int K, const float* A, long lda, const float* B, long ldb,
                    float* C, long ldc, const float* bias, int accumulate, int batch, long strideA, long strideB,
                    long strideC, long strideBias, float* ws, size_t ws_floats, capnet_stream_t stream);
int capnet_sgemm_b3_eligible(int transA, int transB, int M, int N, int K, const float* A, long lda, const float* B, long ldb,
                             const float* C, long ldc, const float* bias, int batch, long strideA, long strideB,
                             long strideC, long strideBias);

/* The same product for the per-time-step shapes of the decoders (M <= 128 rows, A not transposed):
 * K is cut into chunks, one workgroup per (64x64 tile, chunk) loads its chunk in one round trip and
 * writes a partial tile to `workspace` (workspace_floats >= ceil(K/64)*M*N for the finest split);
 * the partials are summed in a fixed order. Falls back to capnet_sgemm's kernel when the shape
 * does not qualify. Replaces the per-step nn.Linear calls of stylenet/model.py:147-150,189 and
 * model_att.py:59-60,283. */
int capnet_sgemm_splitk(int transA, int transB, int M, int N, int K, const float* A, long lda,
                        const float* B, long ldb, float* C, long ldc, const float* bias,
                        int accumulate, float* workspace, size_t workspace_floats,
                        capnet_stream_t stream);

/* The same, with tile counters: `counters` = n_counters ints (>= ceil(N/64)), zero before the first call and left
 * zero by every call. With them a product of M <= 16 rows (the time step of a 12-image batch, decoding) is ONE
 * launch: a workgroup per (16 rows x 64 columns, 256-k chunk), and the workgroup that arrives last at a tile's
 * counter sums the chunk partials in chunk order (csrc/gemm_f32.hip, gemm_rows16_kernel). Other shapes: as
 * capnet_sgemm_splitk. Calls that share `counters` or `workspace` must be ordered on one stream. */
int capnet_sgemm_splitk_fused(int transA, int transB, int M, int N, int K, const float* A, long lda,
                              const float* B, long ldb, float* C, long ldc, const float* bias,
                              int accumulate, float* workspace, size_t workspace_floats, int* counters,
                              size_t n_counters, capnet_stream_t stream);

/* out[c] (+)= sum_r x[r][c]  -- bias gradients (autograd of nn.Linear bias). */
int capnet_colsum(const float* x, long ld, int rows, int C, float* out, int accumulate,
                  capnet_stream_t stream);

/* out[row] = index of the first maximum of x[row][0:V]  -- `output.max(1)` at
 * stylenet/model.py:190, nic/model.py:109. */
int capnet_argmax_rows(const float* x, int rows, int ld, int V, int* out, capnet_stream_t stream);

/* One beam-search expansion of `sample()` (stylenet/model.py:229-249, model_att.py:362-384,
 * nic/model.py sample): scores[r][v] = prev_scores[r] + log_softmax(logits[r])[v]; returns the k
 * best entries of the flattened [rows*V] scores, best first (ties: lower flat index):
 * top_scores[k], top_index[k] (flat index = r*V + v; the caller splits it with // and %).
 * rows, k <= 16. The reference passes rows = 1 on the first step (all beams identical). */
int capnet_beam_topk(const float* logits, long ld, int rows, int V, const float* prev_scores, int k,
                     float* top_scores, long long* top_index, capnet_stream_t stream);
/* The same expansion for n images in one launch (the test-set evaluator, stylenet/evaluator.py:63-120, decodes every image
 * of a batch with sample(); here all of them advance together): meta is a DEVICE array [n][3] = (first row of the image's
 * beams in logits / prev_scores, rows that compete, k); outputs [n][16], flat indices local to the image. Images whose
 * k is 0 (every beam finished) are skipped. rows, k <= 16. */
int capnet_beam_topk_batched(const float* logits, long ld, int V, const float* prev_scores, const int* meta, int n,
                             float* top_scores, long long* top_index, capnet_stream_t stream);

/* ---- ResNet-152 trunk -------------------------------------------------------------------
 * torchvision resnet152 children[:-1] (pooled [B][2048]) or [:-2] (NHWC map [B][S][S][2048]),
 * as run under torch.no_grad() by EncoderCNN.forward: stylenet/model.py:15-18,23-25,
 * stylenet/model_att.py:15-18,23-24, nic/model.py:14-17,22-24. train != 0 reproduces
 * encoder.train() (stylenet/train_multitask.py:367): batch statistics + running-stat update
 * (momentum 0.1, eps 1e-5); train == 0 uses the running statistics.
 * Convolution i (torchvision parameter order, 155 of them) takes its weights PACKED as
 * [Cout][KH][KW][Cin] rows padded to `row_stride` floats (capnet_pack_conv_weight). */
int capnet_trunk_create(int batch, int height, int width, capnet_trunk_t** out);
void capnet_trunk_destroy(capnet_trunk_t* t);
size_t capnet_trunk_workspace_bytes(const capnet_trunk_t* t);
int capnet_trunk_num_convs(const capnet_trunk_t* t);
int capnet_trunk_final_side(const capnet_trunk_t* t);
double capnet_trunk_flops(const capnet_trunk_t* t); /* 2*MACs of all convolutions, this batch */
double capnet_trunk_conv_flops(const capnet_trunk_t* t, int i); /* 2*MACs of convolution i (direct sum) */
int capnet_trunk_conv_shape(const capnet_trunk_t* t, int i, int* cout, int* cin, int* ksize,
                            int* stride, int* row_stride);
/* w_packed / bn_* : HOST arrays of num_convs DEVICE pointers. out_pooled and/or out_map may
 * be NULL (at least one must be given).
 * input_exponents: NULL, or a HOST array of num_convs ints e[i]: convolution i's input tensor is multiplied by 2^e[i] on
 * its way into the f16 planes of the split-f16 kernels (folded into the preceding BatchNorm's scale / shift where there
 * is one; exact; undone in the epilogue). What gives the split operands the domain of fp32: the caller derives e[i] from
 * the BatchNorm parameters so that the tensor's largest possible value sits just below f16's 65 504
 * (capnet.model._TrunkRunner._input_exponents).
 * err_flag: NULL, or the device error word: bit 3 (value 8) is set when a convolution's output statistics (train mode)
 * or the pooled features (inference) are not finite -- an activation beyond the range the exponents allow for, or a
 * genuine fp32 overflow; never silently inf. */
int capnet_trunk_forward(const capnet_trunk_t* t, const float* images_nchw,
                         const float* const* w_packed, const float* const* bn_weight,
                         const float* const* bn_bias, float* const* bn_running_mean,
                         float* const* bn_running_var, int train, float momentum, float eps,
                         void* workspace, float* out_pooled, float* out_map,
                         const int* input_exponents, int* err_flag, capnet_stream_t stream);

/* train == 2 in capnet_trunk_forward: batch statistics as for train == 1, but the running
 * statistics are NOT touched; the pass leaves every BatchNorm's batch mean / unbiased variance in
 * the workspace and this call applies them (same arithmetic, momentum as given). Lets two passes
 * be in flight on different streams (different workspaces) while the running statistics are still
 * updated in pass order: the caller orders these calls with stream events
 * (capnet.train.TrunkPipeline). */
/* Tail balancing of the convolutions (tiles past the last full round of 256 CUs are cut into K
 * slices summed by a fix-up launch): on by default; pays for a single pass, costs when several
 * passes share the chip (capnet.train.TrunkPipeline turns it off). Applies to later forwards. */
int capnet_trunk_set_tail_balance(const capnet_trunk_t* t, int on);

int capnet_trunk_update_running(const capnet_trunk_t* t, const void* workspace,
                                float* const* bn_running_mean, float* const* bn_running_var,
                                float momentum, capnet_stream_t stream);
/* Per-convolution hipEvent timing for bench.py's roofline: while enabled, every conv kernel
 * launched by capnet_trunk_forward is bracketed by two events on the launch stream;
 * collect synchronises on them and returns the totals since the previous collect. enable = N > 1:
 * only every N-th pass is bracketed (an event pair is a bubble in its stream). */
int capnet_trunk_set_timing(capnet_trunk_t* t, int enable);
/* Brackets the NEXT capnet_trunk_forward only, whatever capnet_trunk_set_timing says: for a caller that replays most passes
 * from hipGraphs (events cannot ride in a replayed graph) and launches every N-th one directly. */
int capnet_trunk_time_next_pass(capnet_trunk_t* t);
int capnet_trunk_collect_timing(capnet_trunk_t* t, double* conv_ms, long* conv_launches,
                                double* conv_flops);
/* Weight image convolution i expects: 0 = rows [Cout][row_stride] (capnet_pack_conv_weight),
 * 1 = K-major [row_stride][Cout] (capnet_pack_conv_weight_kmajor; streamed to LDS by LDS-DMA) -- the f32-MFMA kernels,
 * which take every convolution when the trunk is created with CAPNET_NO_H3=1 in the environment and otherwise those the
 * split-f16 kernels do not (none in ResNet-152),
 * 5 = the split-f16 image of a 1x1 or 3x3 convolution (capnet_conv_f16x3_pack with tile width
 * capnet_trunk_conv_tile_n(t, i)),
 * 6 = the split-f16 image of the stem (capnet_conv_stem_f16x3_pack; unless CAPNET_NO_STEM_H3=1). */
int capnet_trunk_conv_kmajor(const capnet_trunk_t* t, int i);
int capnet_trunk_conv_tile_n(const capnet_trunk_t* t, int i);
int capnet_pack_conv_weight_kmajor(const float* w_oihw, float* out, int Cout, int Cin, int KH,
                                   int KW, int k_rows, capnet_stream_t stream);
/* OIHW (torch Conv2d.weight) -> packed rows */
int capnet_pack_conv_weight(const float* w_oihw, float* out, int Cout, int Cin, int KH, int KW,
                            int row_stride, capnet_stream_t stream);
/* nn.AdaptiveAvgPool2d(out_side) on an NHWC map whose side divides out_side (7 -> 14 is a
 * 2x replication): stylenet/model_att.py:19-20,26-28. */
int capnet_adaptive_pool_replicate(const float* x, float* out, int B, int side, int out_side,
                                   int C, capnet_stream_t stream);

/* single building blocks of the trunk, exported for the parity tests */
int capnet_conv2d_fwd(const float* x, long sxb, long sxh, long sxw, long sxc,
                      const float* w_packed, int row_stride, float* y, const float* in_scale,
                      const float* in_shift, int relu_in, float* part_sum, float* part_sq, int B,
                      int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad,
                      int tile, capnet_stream_t stream);
/* C[M][N] = A[M][K] . B[N][K]^T + bias[N] with both operands staged by LDS-DMA (csrc/gemm_dma.hip):
 * the vocabulary projection logits = hiddens . C^T + b (stylenet/model.py:193-194) and every
 * nn.Linear forward whose shape is eligible (N % 64 == 0, K % 32 == 0, B and C dense, 16-B aligned). */
int capnet_sgemm_nt_dma_eligible(int M, int N, int K, const float* A, long lda, const float* B,
                                 long ldb, const float* C, long ldc);
int capnet_sgemm_nt_dma(int M, int N, int K, const float* A, long lda, const float* B, float* C,
                        const float* bias, capnet_stream_t stream);
/* 1x1 convolution (any stride) on an NHWC input, weights in their OIHW = [Cout][Cin] layout, same
 * kernel: the conv1 / conv3 / downsample convolutions of the bottlenecks (torchvision resnet152 via
 * stylenet/model.py:15-18,24). Cin % 32 == 0, Cout % 64 == 0. Either raw output +
 * capnet_conv1x1_tiles_m(M) rows of part_sum / part_sq (train-mode BatchNorm statistics of each
 * 128-row tile), or, with out_scale / out_shift, y = act(conv * out_scale[n] + out_shift[n] + res). */
int capnet_conv1x1_tiles_m(long M);
int capnet_conv1x1_fwd_dma(const float* x, long sxb, long sxh, long sxw, const float* w_oi, float* y,
                           float* part_sum, float* part_sq, int B, int H, int W, int Cin, int Cout,
                           int stride, const float* out_scale, const float* out_shift, const float* res,
                           int relu_out, capnet_stream_t stream);

/* 1x1 convolution with fp32-grade results from three f16 MFMA products per multiply (csrc/conv_f16x3.hip):
 * both operands scaled by a power of two and split into two f16 pieces (activations by 2^4, weights by a
 * per-tensor 2^ew kept in the image's header), fp32 accumulation, the accumulators scaled back exactly;
 * rms error against fp64 at or below the f32-MFMA kernels' for |x| < 4094. Needs Cin % 64 == 0,
 * Cout % 64 == 0 and, with a folded input (in_scale), Cin <= 512. in_scale / in_shift / relu_in: BatchNorm + ReLU of the
 * producer applied on load; out_scale / out_shift / res / relu_out: folded inference epilogue (then no statistics);
 * statistics rows: capnet_conv1x1_tiles_m(M). Weights: capnet_conv1x1_f16x3_pack for tile width bn = capnet_conv1x1_f16x3_bn(M, Cout). */
/* The same convolution for a dense [M][Cin] input with Cin = 64 / 128 / 256 (conv3 of the bottlenecks of stages 1-3) with
 * the A operand resident in registers (csrc/conv1x1_areg.hip): every 128 rows are folded and split once for all output
 * columns. Same weight image (capnet_conv1x1_f16x3_pack, tile width bn), statistics rows capnet_conv1x1_tiles_m(M).
 * in_exp: the input is scaled by 2^in_exp on its way into the f16 planes and the result scaled back (exact). */
int capnet_conv1x1_fwd_areg(const float* x, const unsigned* image, int bn, float* y, const float* in_scale,
                            const float* in_shift, int relu_in, float* part_sum, float* part_sq, long M, int Cin, int Cout,
                            int in_exp, capnet_stream_t stream);
size_t capnet_conv1x1_f16x3_weight_words(int Cin, int Cout);
int capnet_conv1x1_f16x3_bn(long M, int Cout);
int capnet_conv1x1_f16x3_pack(const float* w_oi, unsigned* image, int Cout, int Cin, int bn,
                              capnet_stream_t stream);
int capnet_conv1x1_fwd_f16x3(const float* x, long sxb, long sxh, long sxw, const unsigned* image, int bn,
                             float* y, const float* in_scale, const float* in_shift, int relu_in,
                             float* part_sum, float* part_sq, int B, int H, int W, int Cin, int Cout,
                             int stride, const float* out_scale, const float* out_shift,
                             const float* res, int relu_out, capnet_stream_t stream);

/* The same kernel as an implicit GEMM over (tap, channel): k = 1 (pad 0) or k = 3 (pad 1), any stride. Weights:
 * capnet_conv_f16x3_pack of the OIHW tensor (capnet_conv_f16x3_weight_words(Cin, Cout, k) words) for tile width
 * bn = capnet_conv1x1_f16x3_bn(M, Cout); statistics rows capnet_conv1x1_tiles_m(M), M = B * OH * OW. */
size_t capnet_conv_f16x3_weight_words(int Cin, int Cout, int k);
int capnet_conv_f16x3_pack(const float* w_oihw, unsigned* image, int Cout, int Cin, int k, int bn,
                           capnet_stream_t stream);
int capnet_conv2d_fwd_f16x3(const float* x, long sxb, long sxh, long sxw, const unsigned* image, int bn, float* y,
                            const float* in_scale, const float* in_shift, int relu_in, float* part_sum,
                            float* part_sq, int B, int H, int W, int Cin, int Cout, int k, int stride, int pad,
                            const float* out_scale, const float* out_shift, const float* res, int relu_out,
                            capnet_stream_t stream);

/* The stride-1 3x3 convolutions with the tile's input patch staged once in LDS instead of once per tap
 * (csrc/conv3x3_patch.hip; torchvision Bottleneck.conv2, /root/reference/stylenet/model.py:14-33): dense NHWC x
 * [B][H][W][Cin], Cin % 32 == 0, Cout % bn == 0, W <= 56; weight image, bn and statistics rows exactly as
 * capnet_conv2d_fwd_f16x3 with k = 3. shared_chip != 0: other kernels run beside this one (several trunk passes in
 * flight) -- the <= 128-VGPR wave arrangement is used whatever the tile count; 0: launches of at most one tile per CU
 * split K between wave pairs (faster alone). Results differ between the two at rounding level only. */
int capnet_conv3x3_fwd_patch(const float* x, const unsigned* image, int bn, float* y, const float* in_scale,
                             const float* in_shift, int relu_in, float* part_sum, float* part_sq, int B, int H, int W,
                             int Cin, int Cout, int shared_chip, capnet_stream_t stream);

/* A bottleneck block's tail fused into the next block's first convolution (csrc/conv3x3_patch.hip). One launch computes
 *   tail_out [M][Cin] = relu(y3 * s1 + t1 + res (* s2 + t2))     -- torchvision Bottleneck.forward's `out += identity;
 *                                                                    relu(out)` behind bn3 (and the downsample's BatchNorm)
 *   y [M][Cout]       = tail_out . w^T                            -- the next block's conv1 (1x1, stride 1)
 * as /root/reference/stylenet/model.py:33 runs them through resnet152. s2 / t2 null: res is added as it is. Cin / 32 a
 * power of two; weight image (capnet_conv_f16x3_pack, k = 1), bn and statistics rows as capnet_conv2d_fwd_f16x3.
 * tail_out has exactly the values capnet_bn_add_relu would have written. */
int capnet_conv1x1_fwd_tail(const float* y3, const float* s1, const float* t1, const float* res, const float* s2,
                            const float* t2, float* tail_out, const unsigned* image, int bn, float* y, float* part_sum,
                            float* part_sq, long M, int Cin, int Cout, capnet_stream_t stream);
/* The three entry points above with the input's power-of-two prescale: the input tensor is multiplied by 2^in_exp on its
 * way into the f16 planes (folded into in_scale / in_shift where given; the tail written by capnet_conv1x1_fwd_tail
 * stays unscaled) and the accumulators by 2^-in_exp -- exact. With in_exp chosen so that max |input| 2^in_exp lies in
 * [2^14, 2^15) the result has fp32-grade RELATIVE accuracy whatever the scale of the tensor; without it (in_exp = 0)
 * inputs beyond 65 504 overflow the f16 pieces and inputs far below 1 lose their residuals to f16's subnormals.
 * capnet_trunk_forward applies it per tensor (input_exponents). */
int capnet_conv2d_fwd_f16x3_scaled(const float* x, long sxb, long sxh, long sxw, const unsigned* image, int bn, float* y,
                                   const float* in_scale, const float* in_shift, int relu_in, float* part_sum,
                                   float* part_sq, int B, int H, int W, int Cin, int Cout, int k, int stride, int pad,
                                   int in_exp, capnet_stream_t stream);
int capnet_conv3x3_fwd_patch_scaled(const float* x, const unsigned* image, int bn, float* y, const float* in_scale,
                                    const float* in_shift, int relu_in, float* part_sum, float* part_sq, int B, int H,
                                    int W, int Cin, int Cout, int shared_chip, int in_exp, capnet_stream_t stream);
int capnet_conv1x1_fwd_tail_scaled(const float* y3, const float* s1, const float* t1, const float* res, const float* s2,
                                   const float* t2, float* tail_out, const unsigned* image, int bn, float* y,
                                   float* part_sum, float* part_sq, long M, int Cin, int Cout, int in_exp,
                                   capnet_stream_t stream);

/* The stem on the same arithmetic (csrc/conv_stem.hip): 7x7, stride 2, pad 3, 3 -> 64 channels; x is the NCHW image
 * (strides in floats, unit stride along W, W % 4 == 0, 16-B aligned rows), y is NHWC [B][OH][OW][64]. Replaces
 * torchvision resnet152.conv1 as run by /root/reference/stylenet/model.py:14-24,33. Weights: capnet_conv_stem_f16x3_pack
 * of the OIHW tensor [64][3][7][7] into capnet_conv_stem_f16x3_weight_words() words. part_sum / part_sq (or both
 * null): [capnet_conv_stem_f16x3_part_rows(B, H, W)][64], one row per workgroup, for capnet_bn_finalize. */
size_t capnet_conv_stem_f16x3_weight_words(void);
int capnet_conv_stem_f16x3_part_rows(int B, int H, int W);
int capnet_conv_stem_f16x3_pack(const float* w_oihw, unsigned* image, capnet_stream_t stream);
int capnet_conv_stem_fwd_f16x3(const float* x, long sxb, long sxc, long sxh, const unsigned* image, float* y,
                               float* part_sum, float* part_sq, int B, int H, int W, capnet_stream_t stream);

/* the low-VALU kernel used for every trunk convolution with Cin % 16 == 0 and Cout % 64 == 0:
 * NHWC channel-contiguous input, K-major weights, k_rows == KH*KW*Cin */
int capnet_conv2d_fwd_kmajor(const float* x, long sxb, long sxh, long sxw, const float* w_kmajor,
                             int k_rows, float* y, const float* in_scale, const float* in_shift,
                             int relu_in, float* part_sum, float* part_sq, int B, int H, int W,
                             int Cin, int Cout, int KH, int KW, int stride, int pad, int tile,
                             float* slabs, capnet_stream_t stream);
/* `slabs` (may be NULL = no tail balancing): scratch of capnet_conv_kmajor_slab_floats floats.
 * When the tile count is not a multiple of the 256 CUs, the tiles past the last full round are
 * cut into K-slices (fp32 partial slabs, summed in a fixed order by a fix-up launch) so that
 * every CU ends at the same time. */
size_t capnet_conv_kmajor_slab_floats(int M, int Cout, int k_rows, int tile);
/* the launch plan of that call: out5 = {tile, tiles, whole tiles, K-slices per tail tile,
 * k-tiles per slice} (diagnostics / tests) */
void capnet_conv_kmajor_plan(int M, int Cout, int k_rows, int tile, int* out5);
/* rows of the part_sum / part_sq arrays that call writes (tile = 0: its own choice) */
int capnet_conv_kmajor_tiles_m(int M, int Cout, int k_rows, int tile);
int capnet_conv_tiles_m(int M, int Cout, int tile);
int capnet_bn_finalize(const float* part_sum, const float* part_sq, int tiles, int C, long count,
                       const float* gamma, const float* beta, float* running_mean,
                       float* running_var, float momentum, float eps, float* scale, float* shift,
                       capnet_stream_t stream);
int capnet_bn_add_relu(const float* y, const float* s1, const float* t1, const float* res,
                       const float* s2, const float* t2, float* out, long rows, int C,
                       capnet_stream_t stream);
int capnet_bn_relu_maxpool(const float* y, const float* scale, const float* shift, float* out,
                           int B, int H, int W, int C, capnet_stream_t stream);
int capnet_global_avgpool(const float* x, float* out, int B, int HW, int C,
                          capnet_stream_t stream);

/* ---- encoder head: nn.BatchNorm1d(embed, momentum=0.01) -- stylenet/model.py:20,26 ------- */
int capnet_bn1d_fwd(const float* x, int B, int C, const float* gamma, const float* beta,
                    float* running_mean, float* running_var, int train, float momentum, float eps,
                    float* y, float* save_mean, float* save_invstd, capnet_stream_t stream);
int capnet_bn1d_bwd(const float* dy, const float* x, int B, int C, const float* gamma,
                    const float* save_mean, const float* save_invstd, float* dx, float* dgamma,
                    float* dbeta, capnet_stream_t stream);

/* ---- caption decoders: whole-sequence forward / BPTT ------------------------------------
 * dims (host int[10]) = {B, T, steps, N, E, F, H, V, has_features, cell}
 *   B batch, T caption columns, steps = max(lengths), N = sum(lengths) packed rows,
 *   E embed, F factored (ignored for cell 1), H hidden, V vocab,
 *   cell 0 = DecoderFactoredLSTM.forward  (stylenet/model.py:157-196, forward_step :115-155)
 *   cell 1 = DecoderRNN.forward           (nic/model.py:81-115, nn.LSTMCell :52,77)
 * batch_sizes (host int[steps]) = pack_padded_sequence(...).batch_sizes; tf_mask (host
 * uint8[steps]) = the per-step `random.random() < teacher_forcing_ratio` draws (model.py:181).
 * weights (host array of 32 device pointers):
 *   cell 0: [0..3]=V_{i,f,o,c}.weight [F][E]  [4..7]=V bias  [8..11]=S_g.weight (mode-selected)
 *           [12..15]=S bias  [16..19]=U_g.weight [H][F]  [20..23]=U bias
 *           [24..27]=W_g.weight [H][H]  [28..31]=W bias
 *   cell 1: [0]=lstm.weight_ih [4H][E]  [4]=lstm.bias_ih  [24]=lstm.weight_hh  [28]=lstm.bias_hh
 * emb = B.weight / embed.weight [V][E]; Cw/Cb = output projection (only read on free-running
 * steps, for the argmax feedback). hiddens [N][H] is the packed `torch.cat(hiddens, 0)`.
 * saved / saved_i / scratch: caller-provided device buffers of the sizes queried below; saved
 * buffers must reach the matching backward call untouched. err_flag: device int, set non-zero
 * when a token id is out of range (the caller checks it when it next synchronises). */
size_t capnet_seq_saved_floats(const int* dims);
size_t capnet_seq_saved_ints(const int* dims);
size_t capnet_seq_fwd_scratch_floats(const int* dims);
size_t capnet_seq_bwd_scratch_floats(const int* dims);
int capnet_seq_forward(const int* dims, const int* batch_sizes, const unsigned char* tf_mask,
                       const long long* captions, const float* features, const float* emb,
                       const float* const* weights, const float* Cw, const float* Cb,
                       float dropout_p, unsigned long long seed, int training, float* saved,
                       int* saved_i, float* scratch, float* hiddens, int* err_flag,
                       capnet_stream_t stream);
/* grads (host array of 9 device pointers, all written, not accumulated):
 *   [0] dV  [4F][E] (cell 1: d weight_ih [4H][E])   [1] dV bias [4F]      (cell 1: unused)
 *   [2] dS  [4][F][F] (mode-selected S)              [3] dS bias [4F]
 *   [4] dU  [4][H][F]                                [5] dbias [4H] (U bias = W bias; cell 1:
 *                                                        bias_ih = bias_hh)
 *   [6] dW  [4H][H] (cell 1: d weight_hh)            [7] d emb [V][E]   [8] d features [B][E]|NULL
 * Gate order inside the packed blocks: cell 0 i,f,o,c ; cell 1 i,f,g,o (nn.LSTMCell). */
int capnet_seq_backward(const int* dims, const int* batch_sizes, const float* d_hiddens,
                        const float* hiddens, const float* saved, const int* saved_i,
                        float* scratch, float* const* grads, float dropout_p,
                        unsigned long long seed, int training, capnet_stream_t stream);

/* Stacked FactoredLSTM layers (BASELINE configs[3] / [4]: "2-layer", "3-layer"). PERF-ONLY, PARITY UNPINNED: the reference
 * advertises lstm_layers 1, 2, 3 (README.md:24) but its decoders ignore num_layers (stylenet/model.py:37); the semantics
 * are SURVEY App. A-1's, modelled on the only stacking in the tree (seq2seq/model.py:45-49): layer l > 0 is the same
 * factored cell on dropout(hidden of layer l - 1) at the same step (V: Linear(H -> F)), the top layer's hidden feeds C.
 * dims = capnet_seq_forward's of layer 0; the layers above have E = H and no features (sizes: capnet_seq_saved_floats
 * etc. with those dims). weights: nlayers x 32 pointers in capnet_seq_forward's order; saved / saved_i / hiddens: one
 * buffer per layer (the top layer's hiddens [N][H] are the output); scratch as capnet_seq_forward's.
 * Backward: d_hiddens of the top layer in; grads: nlayers x 9 pointers in capnet_seq_backward's order (dEmb / dFeat of
 * layer 0 only); dh_work: nlayers - 1 buffers [N][H]; scratch: the largest capnet_seq_bwd_scratch_floats of the layers. */
int capnet_seq_forward_stacked(const int* dims, int nlayers, const int* batch_sizes, const unsigned char* tf_mask,
                               const long long* captions, const float* features, const float* emb,
                               const float* const* weights, const float* Cw, const float* Cb, float dropout_p,
                               unsigned long long seed, int training, float* const* saved, int* const* saved_i,
                               float* scratch, float* const* hiddens, int* err_flag, capnet_stream_t stream);
int capnet_seq_backward_stacked(const int* dims, int nlayers, const int* batch_sizes, const float* d_hiddens,
                                const float* const* hiddens, const float* const* saved, const int* const* saved_i,
                                float* scratch, float* const* dh_work, float* const* grads, float dropout_p,
                                unsigned long long seed, int training, capnet_stream_t stream);

/* ---- attention decoders: DecoderFactoredLSTMAtt.forward (stylenet/model_att.py:238-305) and
 * DecoderRNNAtt.forward (nic/model_att.py:152-202) ----
 * dims (host int[12]) = {B, T, steps, N, E, F, H, V, A, P, C, cell}: A attention size, P pixels
 * (14*14), C feature size (2048; must be a multiple of 512); cell 0 = factored LSTM, cell 1 =
 * nn.LSTMCell(E+C, H) (F ignored, pass 4; weights[0]/[4] = weight_ih [4H][E+C] / bias_ih,
 * weights[24]/[28] = weight_hh / bias_hh, the other V/S/U/W slots NULL; grads[0] = d weight_ih,
 * grads[5]/[6] first 4H rows = d weight_hh / d bias_ih = d bias_hh, grads[1..4] may be NULL). captions are the reference's
 * captions[:, :-1] (T columns), batch_sizes / tf_mask as for capnet_seq_forward, features the
 * NHWC map [B][P][C] of the attention encoder (no gradient: the trunk is frozen).
 * weights (host array of 44 device pointers): [0..31] V/S/U/W weights and biases per gate as in
 * capnet_seq_forward (V_g is [F][E+C]); [32,33] init_h w,b  [34,35] init_c  [36,37] the selected
 * attention module's encoder_att  [38,39] decoder_att  [40,41] full_att  [42,43] f_beta.
 * Outputs: hiddens [N][H]; alphas [B][steps][P] (zero where a sequence has ended, :261,296).
 * encoder_att(features) is computed once per call, not once per step (:59,279). */
/* Process-wide choice for the factored cell's input product U_g (S_g (V_g x + bV_g) + bS_g) (stylenet/model_att.py:196-236)
 * inside capnet_att_seq_forward/backward: 0 = by shape (batches of at most 16 rows run it as ONE product per step against
 * U_g S_g V_g, formed once per call, and form the intermediate rows for all steps at once in the backward: the steps of
 * such a batch are chains of dependent launches, six of them this product's), 1 = always, -1 = never (three products per
 * step each way). Returns the previous choice. Set it between calls, not between a forward and its backward. */
int capnet_att_set_chain_mode(int mode);
size_t capnet_att_saved_floats(const int* dims);
size_t capnet_att_saved_ints(const int* dims);
size_t capnet_att_fwd_scratch_floats(const int* dims);
size_t capnet_att_bwd_scratch_floats(const int* dims);
/* One attention step for `rows` rows (Attention.forward, stylenet/model_att.py:51-70, plus the
 * f_beta gate of :283-284), as used per time step inside capnet_att_seq_forward and per decode
 * step by sample() (model_att.py:352-357). att1 [rows][P][A] = encoder_att(features) (hoisted),
 * feat [rows][P][C], att2 (ld ldz) = decoder_att(h), gate_io (ld ldz): f_beta(h) in, sigmoid out;
 * w_full [A], b_full [1]. Outputs: alpha_out [rows][P]; alphas_bt[(row*steps + t)*P + p] (the
 * user-visible alphas tensor); awe_out [rows][C] (ungated context); xa_out (ld ldx) = gate*awe.
 * scores_ws: scratch [rows][P] (the scores before the softmax). */
int capnet_att_step_fwd(const float* att1, const float* feat, const float* att2, float* gate_io,
                        long ldz, const float* w_full, const float* b_full, int rows, int P, int A,
                        int C, float* alpha_out, float* alphas_bt, int steps, int t, float* awe_out,
                        float* xa_out, long ldx, float* scores_ws, capnet_stream_t stream);

int capnet_att_seq_forward(const int* dims, const int* batch_sizes, const unsigned char* tf_mask,
                           const long long* captions, const float* features, const float* emb,
                           const float* const* weights, const float* Cw, const float* Cb,
                           float dropout_p, unsigned long long seed, int training, float* saved,
                           int* saved_i, float* scratch, float* hiddens, float* alphas,
                           int* err_flag, capnet_stream_t stream);
/* grads (host array of 16 device pointers, all written): [0] dV [4F][E+C]  [1] dV bias [4F]
 * [2] dS [4][F][F]  [3] dS bias  [4] dU [4][H][F]
 * [5] dWz [4H+A+C][H] = [dW_i; dW_f; dW_o; dW_c; d decoder_att.weight; d f_beta.weight]
 * [6] dbz [4H+A+C] (first 4H: U and W biases)  [7] d encoder_att.weight [A][C]  [8] its bias
 * [9] d full_att.weight [A]  [10] d full_att.bias [1]  [11] d init_h.weight [H][C]  [12] bias
 * [13] d init_c.weight  [14] bias  [15] d emb [V][E].  d_alphas may be NULL. */
int capnet_att_seq_backward(const int* dims, const int* batch_sizes, const float* d_hiddens,
                            const float* d_alphas, const float* hiddens, const float* features,
                            const float* const* weights, const float* saved, const int* saved_i,
                            float* scratch, float* const* grads, float dropout_p,
                            unsigned long long seed, int training, capnet_stream_t stream);

/* Single-step pieces used by forward_step() / sample() (no autograd):
 *   out[r] = emb[idx[r]]                      -- self.B(k_prev_words), stylenet/model.py:221
 *   gate pointwise on pre-activations [b][4H] (in place: overwritten by the activated gates),
 *   cell 0: blocks i,f,o,c~ and h = o*c (model.py:147-153); cell 1: i,f,g,o and h = o*tanh(c). */
int capnet_embedding_fwd(const long long* idx, int n, const float* emb, int E, int V, float* out,
                         int* err_flag, capnet_stream_t stream);
int capnet_lstm_pointwise_fwd(float* pre, const float* c_prev, float* c_out, float* h_out, int b,
                              int H, int cell, capnet_stream_t stream);
/* ... and its backward (the cell of capnet.stacked's layers): gates = the ACTIVATED gates the forward left in `pre`,
 * c / c_prev the new / previous cell state (c_prev NULL = zeros), dh = dL/dh; dc_io: in dL/dc, out dL/dc_prev;
 * dpre [b][4H] = dL/d pre-activations. */
int capnet_lstm_pointwise_bwd(const float* gates, const float* c, const float* c_prev, const float* dh, float* dc_io,
                              float* dpre, int b, int H, int cell, capnet_stream_t stream);

/* One recurrent step in one launch (used inside capnet_seq_forward for t > 0):
 *   gates[b][4H] (in: U(S(V x)) + biases, ld ldg) += h_prev[b][H] . W[4H][H]^T (W given as the
 *   capnet_lstm_pack_wfrag image), then the gate
 *   non-linearities, c = f*c_prev + i*g, h = o*c (cell 0) / o*tanh(c) (cell 1); gates are
 *   overwritten by the activated values. Each wave keeps its slice of w_cat in registers for
 *   the launch. b <= 64, H in {16,32,64,96,128,256,512} (capnet_lstm_step_fused_supported). */
size_t capnet_lstm_wfrag_floats(int H);
/* w_cat [4H][H] (gate blocks in the cell's storage order) -> the fragment-major image the step
 * kernel streams with fully coalesced 1-KB reads (one per 4 MFMA k-steps and wave) */
int capnet_lstm_pack_wfrag(const float* w_cat, float* w_frag, int H, int cell,
                           capnet_stream_t stream);
int capnet_lstm_step_fused(const float* h_prev, const float* w_frag, float* gates, long ldg,
                           const float* c_prev, float* c_out, float* h_out, int b, int H, int cell,
                           capnet_stream_t stream);
/* diagnostic variant (cell 0): also writes 5 s_memtime readings per workgroup to `stamps`
 * ([H/8][5] uint64: start, operands staged, MFMAs done, K-reduced, end) */
int capnet_lstm_step_fused_stamped(const float* h_prev, const float* w_frag, float* gates, long ldg,
                                   const float* c_prev, float* c_out, float* h_out, int b, int H,
                                   unsigned long long* stamps, capnet_stream_t stream);
int capnet_lstm_step_fused_supported(int b, int H);

/* Persistent sequence kernel: steps [t0, t1) of the recurrence in ONE launch, the recurrent
 * weights register-resident for the launch (north star: "the LSTM step as a persistent
 * wavefront-resident kernel"; the loop it replaces: stylenet/model.py:180-191 for teacher-forced
 * steps, cell :147-153 / nic/model.py:77). H = 512, b <= 128, a device with >= 256 CUs
 * (capnet_lstm_persist_supported; CAPNET_NO_PERSISTENT_LSTM=1 disables it).
 *   w_img        capnet_lstm_persist_pack image of w_cat [4H][H]     (capnet_lstm_persist_w_floats; the weights as two f16
 *                pieces each, scaled by 2^10, in MFMA operand order: fp32-grade products for |w| < 32)
 *   gates        [N][4H] packed time-major: in = pre-activations without the recurrent term,
 *                out = activated gates (blocks in the cell's order)
 *   cell_states  [N][H] out (row block of step t0-1 is read when t0 > 0)
 *   hiddens      [N][H] out (row block of step t0-1 is read when t0 > 0); also the medium through
 *                which the workgroups hand h_t to each other
 *   batch_sizes  [t1] non-increasing rows per step (host array)
 *   ctl          capnet_lstm_persist_ctl_ints() ints, zeroed once before the first segment of a
 *                forward pass; segment = 1, 2, ... numbers the launches that share it
 *   err_flag     bit 5 (value 32) is set if the weight image holds a value beyond f16's range (a recurrent weight with
 *                |w| >= 64: capnet_lstm_persist_pack marks the image, the results are not finite);
 *                bit 2 (value 4) is set if a bounded wait expired (the results are then invalid; capnet_clamp_adam
 *                given the same word skips its update)
 *   stamps       NULL, or ([t1-t0][256][8] + [256][2]) uint64 s_memtime readings (diagnostic instantiation) */
int capnet_lstm_persist_supported(int b, int H);
/* Process-wide mode of the persistent path; returns the previous mode, mode < 0 only queries. Bit 0: off (one launch
 * per step everywhere; the initial value when CAPNET_NO_PERSISTENT_LSTM=1 is in the environment). Bits 1 and 2 force
 * the two rare paths for tests: the cross-XCD (write-through) hand-off, and a handshake time-out (every workgroup
 * leaves, err_flag |= 4). */
int capnet_lstm_persist_set_mode(int mode);
size_t capnet_lstm_persist_w_floats(void);
size_t capnet_lstm_persist_ctl_ints(void);
int capnet_lstm_persist_pack(const float* w_cat, float* w_img, int cell, capnet_stream_t stream);
int capnet_lstm_persist_run(const float* w_img, float* gates, float* cell_states, float* hiddens,
                            const int* batch_sizes, int t0, int t1, int H, int cell, int segment,
                            int* ctl, int* err_flag, unsigned long long* stamps,
                            capnet_stream_t stream);

/* ---- loss: nn.CrossEntropyLoss() (mean) -- stylenet/train_multitask.py:134,383 ---------- */
int capnet_xent_fwd(const float* logits, long ld, int N, int V, const long long* targets,
                    float* lse, float* row_loss, float* loss, int* err_flag,
                    capnet_stream_t stream);
int capnet_xent_bwd(const float* logits, long ld, int N, int V, const long long* targets,
                    const float* lse, const float* grad_out, float* dlogits, long ldd,
                    capnet_stream_t stream);
/* Loss of the attention loops (stylenet/train_multitask_att.py:409-411):
 *   out = nll + alpha_c * ((1 - sum_t alphas[b][t][p])^2).mean()     alphas [B][steps][P]
 * colsum [B*P] receives sum_t alphas for the backward call, which writes
 *   dalphas[b][t][p] = gout * alpha_c * 2 (colsum[b][p] - 1) / (B P). */
int capnet_att_loss_fwd(const float* nll, const float* alphas, int B, int steps, int P, float alpha_c,
                        float* colsum, float* out, capnet_stream_t stream);
int capnet_att_loss_bwd(const float* gout, const float* colsum, int B, int steps, int P, float alpha_c,
                        float* dalphas, capnet_stream_t stream);

/* ---- input pipeline (stylenet/train_multitask.py:62-69: Resize((336,336)) -> RandomCrop(224) ->
 * RandomHorizontalFlip -> ToTensor -> Normalize) on uint8 HWC device images ----------------------
 * capnet_resize_u8: Pillow's two-pass antialiased resample (what torchvision 0.2.2's Resize calls).
 * src [Hs][Ws][3], tmp [Hs][Wo][3] scratch, dst [Ho][Wo][3]. bounds_* (device int [out][2] =
 * {first tap, tap count}) and coef_* (device int [out][kmax], 22-bit fixed point) are the filter
 * tables of the horizontal / vertical pass, built on the host as Pillow's precompute_coeffs does
 * (capnet.data.resample_tables); the result is bit-identical to PIL.Image.resize(BILINEAR).
 * capnet_crop_flip_normalize: src [B][Hs][Ws][3] uint8, params (device int [B][3] = {top, left,
 * flip}), mean/std host float[3]; dst [B][3][Hc][Wc] fp32 = (u8/255 - mean)/std. */
int capnet_resize_u8(const unsigned char* src, int Hs, int Ws, unsigned char* tmp, unsigned char* dst,
                     int Ho, int Wo, const int* bounds_h, const int* coef_h, int kmax_h,
                     const int* bounds_v, const int* coef_v, int kmax_v, capnet_stream_t stream);
int capnet_crop_flip_normalize(const unsigned char* src, int B, int Hs, int Ws, const int* params,
                               float* dst, int Hc, int Wc, const float* mean, const float* stdv,
                               capnet_stream_t stream);

/* count[0] = number of rows whose target is among the k largest logits of its row -- the numerator
 * of utils.accuracy(scores, targets, k) (stylenet/utils.py:127-140; top-5 in val_factual,
 * train_multitask.py:306). Ties are ranked lower index first. count is overwritten. */
int capnet_topk_correct(const float* logits, long ld, int N, int V, const long long* targets, int k,
                        int* count, int* err_flag, capnet_stream_t stream);

/* ---- optimiser: utils.clip_gradient (element-wise clamp, stylenet/utils.py:51-60) fused with
 * torch.optim.Adam.step (stylenet/train_multitask.py:166-167,389; no weight decay, no amsgrad).
 * Host arrays of n device pointers / sizes / per-tensor step counts (>= 1, already
 * incremented). clip <= 0 disables the clamp; write_grad != 0 also stores the clamped grad.
 * skip_flag: NULL, or the device error word the step's kernels were given (err_flag): while it is non-zero
 * the launch leaves parameters, moments and gradients untouched -- a step whose inputs were invalid (token id or
 * target out of range, an expired wait of the persistent LSTM kernel) is dropped, not applied. */
int capnet_clamp_adam(int n, float* const* params, float* const* grads, float* const* exp_avg,
                      float* const* exp_avg_sq, const long* numel, const int* step, float lr,
                      float beta1, float beta2, float eps, float clip, int write_grad,
                      const int* skip_flag, capnet_stream_t stream);

/* targets = pack_padded_sequence(captions, lengths, batch_first=True)[0]
 * (stylenet/train_multitask.py:377-379); batch_sizes is a host array. out: [sum(batch_sizes)]. */
int capnet_packed_targets(const long long* captions, int T, int steps, const int* batch_sizes,
                          long long* out, capnet_stream_t stream);

/* Data-parallel plumbing (no reference counterpart: the reference is single-device): gather
 * n gradient tensors into one flat buffer (direction 0) for ONE RCCL all-reduce, and scatter
 * it back multiplied by `scale` (direction 1). Host arrays of device pointers / sizes. */
int capnet_pack_tensors(int n, float* const* tensors, const long* numel, float* flat,
                        int direction, float scale, capnet_stream_t stream);

/* ---- a bottleneck's conv3 + bn3 + residual + ReLU + the next block's conv1 without ever forming y3 (csrc/fused_block.hip).
 * Replaces, between two blocks of one stage of torchvision's resnet152 (Bottleneck.forward, called from
 * stylenet/model.py:15-18,24): `out = relu(bn3(conv3(relu(bn2(y2)))) + identity)` of block b and `y1 = conv1(out)` of
 * block b + 1. MID = conv3's input channels (64 / 128 / 256), C = 4 MID its output channels; all tensors NHWC fp32
 * flattened to [M = B H W][channels].
 *   capnet_fused_block_pack        role 0: conv3's weights [C][MID] -> its image (+ an fp32 copy, read by the statistics);
 *                                  role 1: the next conv1's weights [MID][C] -> its image. Sizes: .._weight_words.
 *   capnet_fused_block_stats       train mode: (scale, shift) of bn3 -- and its running statistics, momentum as torch --
 *                                  from the second moments of conv3's INPUT: sum_m y3[m,c]^2 = w_c^T (a2^T a2) w_c with
 *                                  a2 = relu(y2 s2 + t2). `work`: .._stats_floats(M, MID) floats, 16-B aligned.
 *   capnet_fused_block_forward     writes out [M][C] and y1 [M][MID]; part_sum / part_sq [.._tiles(M, MID)][MID]: per-tile
 *                                  column sums of y1 for capnet_bn_finalize (null: inference). res = the block's input
 *                                  (sd = td = null) or its downsample branch's raw output with that BatchNorm's (sd, td).
 *   e3 / e1: power-of-two prescales of conv3's and conv1's inputs in the f16 planes (capnet_trunk_forward's exponents). */
size_t capnet_fused_block_weight_words(int C, int MID, int role);
int capnet_fused_block_pack(const float* w, unsigned* img, int C, int MID, int role, capnet_stream_t stream);
size_t capnet_fused_block_stats_floats(long M, int MID);
int capnet_fused_block_stats(const float* y2, const float* s2, const float* t2, const unsigned* w3img, long M, int MID,
                             int in_exp, const float* gamma, const float* beta, float* running_mean, float* running_var,
                             float momentum, float eps, float* scale, float* shift, float* work, int* err_flag,
                             capnet_stream_t stream);
int capnet_fused_block_tiles(long M, int MID);
int capnet_fused_block_forward(const float* y2, const float* s2, const float* t2, const unsigned* w3img, const float* s3,
                               const float* t3, const float* res, const float* sd, const float* td, float* out,
                               const unsigned* w1img, float* y1, float* part_sum, float* part_sq, long M, int MID, int e3,
                               int e1, int* err_flag, capnet_stream_t stream);

/* Data-parallel plumbing, the collective (no reference counterpart -- the reference is single-device): an
 * RCCL communicator behind an opaque handle and the step's ONE collective, an in-place SUM all-reduce of the flat fp32
 * gradient buffer, enqueued on the caller's stream (the side stream of capnet.parallel: it overlaps the next trunk
 * passes). One process per GPU. capnet_comm_unique_id fills 128 bytes on ONE rank; the caller hands them to every rank
 * by its own means (capnet.parallel: torch.distributed's store); capnet_comm_create is collective over the `world`
 * ranks and binds the communicator to the calling process's current HIP device. librccl is resolved at run time (an
 * RCCL already mapped into the process first, then /opt/rocm/lib): a single-GPU host never loads it. */
typedef struct capnet_comm capnet_comm_t;
int capnet_comm_unique_id(void* id128);
int capnet_comm_create(const void* id128, int rank, int world, capnet_comm_t** out);
int capnet_comm_destroy(capnet_comm_t* comm);
int capnet_allreduce_grads(capnet_comm_t* comm, float* flat, long count, capnet_stream_t stream);

/* Data-parallel plumbing, the error word: capnet_clamp_adam drops a step while the LOCAL device error word is set, and
 * every rank has to take that decision alike. direction 0: *slot = (word != 0) -- the slot is one extra float behind the
 * flat gradient buffer, so it is summed by the same all-reduce; direction 1: if the sum is positive and the local word is
 * clear, set it to 16 ("another rank dropped this step"). No reference counterpart. */
int capnet_err_word_exchange(int* err_flag, float* slot, int direction, capnet_stream_t stream);

/* *counter += 1 while *err_flag != 0. capnet.optim.Adam launches it once per step() in front of capnet_clamp_adam and
 * takes the dropped steps out of its host-side step counts (torch.optim.Adam's bias correction, train_multitask.py:389)
 * when check_device_errors() reports them. */
int capnet_count_skipped(const int* err_flag, int* counter, capnet_stream_t stream);

/* x[i] = min(max(x[i], lo), hi): utils.clip_gradient alone (stylenet/utils.py:57-60). */
int capnet_clamp(float* x, long n, float lo, float hi, capnet_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* CAPNET_H_ */
