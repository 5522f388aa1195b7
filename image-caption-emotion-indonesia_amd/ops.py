"""Host-side operators over the C ABI: plain wrappers and torch.autograd.Functions.

PyTorch is plumbing here (device memory, streams, autograd bookkeeping). Every arithmetic
step of the hot path is a HIP kernel in libcapnet_hip.so reached through ctypes; nothing in
this file computes on the CPU or through a torch operator. Inputs must be CUDA fp32 tensors.
"""
import ctypes as C

import sys
import weakref

import torch
from torch.autograd.function import once_differentiable

from . import _lib
from ._lib import CapnetError, check, current_stream, int_array, ptr, ptr_array

CELL_FACTORED = 0
CELL_LSTM = 1

_err_flags = {}


def _need_cuda(*tensors):
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise CapnetError("capnet operators run on the GPU only (got a %s tensor); there is "
                              "no CPU fallback" % t.device)
        if t.dtype not in (torch.float32, torch.int64, torch.int32):
            raise CapnetError("unsupported dtype %s" % t.dtype)


def _c(t):
    return t if t.is_contiguous() else t.contiguous()


def err_flag(device):
    """Per-device int32 flag set by kernels on out-of-range token ids / targets."""
    key = torch.device(device).index or 0
    f = _err_flags.get(key)
    if f is None:
        f = torch.zeros(1, dtype=torch.int32, device=device)
        _err_flags[key] = f
    return f


_ERR_BITS = {1: "token id out of range", 2: "target out of range",
             8: "a non-finite value in the ResNet-152 trunk (an activation beyond the range its layer's power-of-two prescale "
                "allows for the split-f16 operands -- possible only in inference, with running statistics far from the data "
                "-- or a genuine fp32 overflow)",
             4: "a bounded wait of the persistent LSTM kernel expired (its 256 workgroups were not co-resident in time); "
                "the process now runs one launch per LSTM step, as CAPNET_NO_PERSISTENT_LSTM=1 does from the start",
             32: "a recurrent LSTM weight beyond |w| < 64, the range the persistent LSTM kernel's f16 weight image covers (the "
                 "step's results are not finite); CAPNET_NO_PERSISTENT_LSTM=1 runs one launch per step on f32 operands",
             16: "another rank of the data-parallel job raised its error word: this rank dropped the same steps so that "
                 "the replicas stay identical (capnet.parallel)"}

_optimizers = weakref.WeakSet()      # capnet.optim.Adam instances: they hold the per-optimizer dropped-step counters


def register_optimizer(opt):
    _optimizers.add(opt)


def skip_counter(device):
    return torch.zeros(1, dtype=torch.int32, device=device)


def count_skipped(counter):
    """counter += 1 on the device while the error word is set (one launch per optimizer step)."""
    check(_lib.lib().capnet_count_skipped(ptr(err_flag(counter.device)), ptr(counter), current_stream()),
          "capnet_count_skipped")


def err_word_exchange(slot, direction):
    check(_lib.lib().capnet_err_word_exchange(ptr(err_flag(slot.device)), ptr(slot), int(direction), current_stream()),
          "capnet_err_word_exchange")


def check_device_errors(recover_lstm_timeout=False):
    """Synchronising check of the device-side error flags (call where the loop already syncs).

    While a flag is set capnet.optim.Adam's update kernel leaves parameters and moments alone (capnet_clamp_adam's
    skip_flag), so the steps between the fault and this check were DROPPED, not applied with garbage gradients; every
    registered optimizer takes the dropped steps out of its host-side step counts here (bias correction stays in step
    with the moments on the device), whether or not the error is then raised.
    recover_lstm_timeout: an expired wait of the persistent LSTM kernel ALONE (bit 4: a transient loss of co-residency
    beside the trunk passes) is logged instead of raised -- the process continues on the launch-per-step kernel, as the
    training loops do (capnet.train.train_factual / train_emotion)."""
    for f in _err_flags.values():
        v = int(f.item())
        if not v:
            continue
        f.zero_()
        dropped = [o.forget_dropped_steps() for o in list(_optimizers)]
        if v & 4:
            _lib.lib().capnet_lstm_persist_set_mode(1)
        msg = ("device-side error flag %d: %s. %s optimizer steps since the previous check were skipped."
               % (v, "; ".join(m for b, m in _ERR_BITS.items() if v & b) or "?", sum(dropped) if dropped else "The"))
        if recover_lstm_timeout and v == 4:
            sys.stderr.write("[capnet] " + msg + " Continuing.\n")
            continue
        raise CapnetError(msg)
    for o in list(_optimizers):
        o.forget_dropped_steps(none_dropped=True)


# ---------------------------------------------------------------------------------------
# plain wrappers
# ---------------------------------------------------------------------------------------
def sgemm(A, B, transA=False, transB=False, bias=None, out=None, accumulate=False, force_tile=0):
    """out[M,N] (+)= op(A) @ op(B) + bias, 2-D contiguous operands."""
    _need_cuda(A, B, bias, out)
    A, B = _c(A), _c(B)
    M, K = (A.shape[1], A.shape[0]) if transA else (A.shape[0], A.shape[1])
    if transB:
        N, Kb = B.shape
    else:
        Kb, N = B.shape
    if K != Kb:
        raise CapnetError("sgemm: inner dimensions differ (%d vs %d)" % (K, Kb))
    if out is None:
        if accumulate:
            raise CapnetError("sgemm: accumulate needs an output tensor")
        out = torch.empty((M, N), dtype=torch.float32, device=A.device)
    # `out` may be a column block of a wider row-major matrix (unit column stride)
    if out.dim() != 2 or tuple(out.shape) != (M, N) or out.stride(1) != 1 or out.stride(0) < N:
        raise CapnetError("sgemm: bad output tensor")
    check(_lib.lib().capnet_sgemm(int(transA), int(transB), M, N, K, ptr(A), A.shape[1], ptr(B),
                                  B.shape[1], ptr(out), out.stride(0), ptr(bias), int(accumulate), 1, 0, 0, 0,
                                  0, force_tile, current_stream()), "capnet_sgemm")
    return out


_splitk_ws = {}


def sgemm_splitk(A, B, transB=False, bias=None):
    """A @ op(B) + bias through capnet_sgemm_splitk: the library cuts K into slabs when the output
    has few tiles (per-step products; dH = dlogits @ C with K = vocab). One cached 32 MB slab
    workspace per device."""
    _need_cuda(A, B, bias)
    A, B = _c(A), _c(B)
    M, K = A.shape
    N = B.shape[0] if transB else B.shape[1]
    if (B.shape[1] if transB else B.shape[0]) != K:
        raise CapnetError("sgemm_splitk: inner dimensions differ")
    key = A.device.index or 0
    ws = _splitk_ws.get(key)
    if ws is None:
        ws = torch.empty(8 * 1024 * 1024, dtype=torch.float32, device=A.device)
        _splitk_ws[key] = ws
    out = torch.empty((M, N), dtype=torch.float32, device=A.device)
    check(_lib.lib().capnet_sgemm_splitk(0, int(transB), M, N, K, ptr(A), A.shape[1], ptr(B), B.shape[1],
                                         ptr(out), N, ptr(bias), 0, ptr(ws), ws.numel(),
                                         current_stream()), "capnet_sgemm_splitk")
    return out


def colsum(x, out=None):
    _need_cuda(x)
    x = _c(x)
    if out is None:
        out = torch.empty(x.shape[1], dtype=torch.float32, device=x.device)
    check(_lib.lib().capnet_colsum(ptr(x), x.shape[1], x.shape[0], x.shape[1], ptr(out), 0,
                                   current_stream()), "capnet_colsum")
    return out


def argmax_rows(x):
    _need_cuda(x)
    x = _c(x)
    out = torch.empty(x.shape[0], dtype=torch.int32, device=x.device)
    check(_lib.lib().capnet_argmax_rows(ptr(x), x.shape[0], x.shape[1], x.shape[1], ptr(out),
                                        current_stream()), "capnet_argmax_rows")
    return out


def beam_topk(logits, prev_scores, rows, k):
    """k best of prev_scores[r] + log_softmax(logits[r]) over the first `rows` rows, flattened.
    Returns (scores [k] float32, flat_index [k] int64) on the device."""
    _need_cuda(logits)
    logits = _c(logits)
    prev_scores = _c(prev_scores)
    scores = torch.empty(k, dtype=torch.float32, device=logits.device)
    index = torch.empty(k, dtype=torch.int64, device=logits.device)
    check(_lib.lib().capnet_beam_topk(ptr(logits), logits.shape[1], rows, logits.shape[1],
                                      ptr(prev_scores), k, ptr(scores), ptr(index),
                                      current_stream()), "capnet_beam_topk")
    return scores, index


def beam_topk_batched(logits, prev_scores, meta):
    """capnet_beam_topk_batched: meta int32 [n, 3] on the device = (first row, competing rows, k) per image.
    Returns (scores [n, 16] float32, flat_index [n, 16] int64) on the device; only the first k entries of a row are set."""
    _need_cuda(logits, prev_scores, meta)
    logits, prev_scores = _c(logits), _c(prev_scores)
    if meta.dtype != torch.int32 or meta.dim() != 2 or meta.shape[1] != 3 or not meta.is_contiguous():
        raise CapnetError("beam_topk_batched: meta must be a contiguous int32 [n, 3] tensor")
    n = meta.shape[0]
    scores = torch.empty((n, 16), dtype=torch.float32, device=logits.device)
    index = torch.empty((n, 16), dtype=torch.int64, device=logits.device)
    check(_lib.lib().capnet_beam_topk_batched(ptr(logits), logits.shape[1], logits.shape[1], ptr(prev_scores), ptr(meta), n,
                                              ptr(scores), ptr(index), current_stream()), "capnet_beam_topk_batched")
    return scores, index


def attention_step(att1, feat, z, A, w_full, b_full, xa=None, xa_col=0):
    """One attention step for s rows (no autograd; Attention.forward / sample()).
    att1 [s, P, A] = encoder_att(features); feat [s, P, C]; z [s, A + C] = [decoder_att(h) |
    f_beta(h)] pre-activations (the gate half is overwritten with its sigmoid).
    Returns (awe [s, C] ungated context, alpha [s, P]); if `xa` [s, W] is given, the GATED context
    is written into xa[:, xa_col : xa_col + C]."""
    _need_cuda(att1, feat, z)
    s_rows, P, Adim = att1.shape
    Cdim = feat.shape[2]
    if not (att1.is_contiguous() and feat.is_contiguous() and z.is_contiguous()):
        raise CapnetError("attention_step: inputs must be contiguous")
    if Adim != A or z.shape[1] != A + Cdim or feat.shape[0] != s_rows or feat.shape[1] != P:
        raise CapnetError("attention_step: shape mismatch")
    dev = att1.device
    alpha = torch.empty((s_rows, P), dtype=torch.float32, device=dev)
    alphas_bt = torch.empty((s_rows, 1, P), dtype=torch.float32, device=dev)
    awe = torch.empty((s_rows, Cdim), dtype=torch.float32, device=dev)
    scores_ws = torch.empty((s_rows, P), dtype=torch.float32, device=dev)
    if xa is None:
        xa, xa_col = torch.empty((s_rows, Cdim), dtype=torch.float32, device=dev), 0
    if not xa.is_contiguous() or xa.shape[0] != s_rows or xa_col + Cdim > xa.shape[1] or xa_col % 4:
        raise CapnetError("attention_step: bad xa")
    wf, bf = _c(w_full.detach()).reshape(-1), _c(b_full.detach()).reshape(-1)
    check(_lib.lib().capnet_att_step_fwd(ptr(att1), ptr(feat), ptr(z), C.c_void_p(z.data_ptr() + 4 * A), z.shape[1],
                                         ptr(wf), ptr(bf), s_rows, P, A, Cdim, ptr(alpha),
                                         ptr(alphas_bt), 1, 0, ptr(awe), C.c_void_p(xa.data_ptr() + 4 * xa_col),
                                         xa.shape[1], ptr(scores_ws), current_stream()), "capnet_att_step_fwd")
    return awe, alpha


def embedding(idx, weight):
    """weight[idx] (no autograd: used by forward_step()/sample())."""
    _need_cuda(idx, weight)
    idx, w = _c(idx), _c(weight.detach())
    if idx.dtype != torch.int64:
        raise CapnetError("embedding indices must be int64")
    out = torch.empty(tuple(idx.shape) + (w.shape[1],), dtype=torch.float32, device=w.device)
    check(_lib.lib().capnet_embedding_fwd(ptr(idx), idx.numel(), ptr(w), w.shape[1], w.shape[0],
                                          ptr(out), ptr(err_flag(w.device)), current_stream()),
          "capnet_embedding_fwd")
    return out


def lstm_pointwise(pre, c_prev, cell):
    """(h, c) from gate pre-activations [b, 4H] and the previous cell state (no autograd)."""
    _need_cuda(pre, c_prev)
    pre = pre.detach().clone().contiguous()
    c_prev = _c(c_prev.detach())
    b, h4 = pre.shape
    H = h4 // 4
    c = torch.empty((b, H), dtype=torch.float32, device=pre.device)
    h = torch.empty((b, H), dtype=torch.float32, device=pre.device)
    check(_lib.lib().capnet_lstm_pointwise_fwd(ptr(pre), ptr(c_prev), ptr(c), ptr(h), b, H, cell,
                                               current_stream()), "capnet_lstm_pointwise_fwd")
    return h, c


class LstmCellFn(torch.autograd.Function):
    """(h, c) = cell(pre-activations [b, 4H], c_prev [b, H]) with autograd: capnet_lstm_pointwise_fwd / _bwd.
    cell 0: gates i,f,o,c~ and h = o*c (stylenet/model.py:147-153); cell 1: i,f,g,o and h = o*tanh(c)."""

    @staticmethod
    def forward(ctx, pre, c_prev, cell):
        _need_cuda(pre, c_prev)
        gates = pre.detach().clone().contiguous()          # the kernel leaves the activated gates here
        cp = _c(c_prev.detach())
        b, h4 = gates.shape
        H = h4 // 4
        c = torch.empty((b, H), dtype=torch.float32, device=gates.device)
        h = torch.empty((b, H), dtype=torch.float32, device=gates.device)
        check(_lib.lib().capnet_lstm_pointwise_fwd(ptr(gates), ptr(cp), ptr(c), ptr(h), b, H, cell, current_stream()),
              "capnet_lstm_pointwise_fwd")
        ctx.save_for_backward(gates, c, cp)
        ctx.cell = cell
        ctx.mark_non_differentiable()
        return h, c

    @staticmethod
    def backward(ctx, dh, dc):
        gates, c, cp = ctx.saved_tensors
        b, h4 = gates.shape
        H = h4 // 4
        dh = _c(dh) if dh is not None else torch.zeros_like(c)
        dc_io = dc.clone().contiguous() if dc is not None else torch.zeros_like(c)
        dpre = torch.empty_like(gates)
        check(_lib.lib().capnet_lstm_pointwise_bwd(ptr(gates), ptr(c), ptr(cp), ptr(dh), ptr(dc_io), ptr(dpre), b, H,
                                                   ctx.cell, current_stream()), "capnet_lstm_pointwise_bwd")
        return dpre, dc_io, None


def lstm_cell(pre, c_prev, cell=0):
    return LstmCellFn.apply(pre, c_prev, cell)


def packed_targets(captions, lengths):
    """pack_padded_sequence(captions, lengths, batch_first=True)[0] for int64 captions."""
    _need_cuda(captions)
    captions = _c(captions)
    bs = batch_sizes_from_lengths(lengths)
    if bs[0] != captions.shape[0] or len(bs) > captions.shape[1]:
        raise CapnetError("packed_targets: lengths do not match captions %s" % (tuple(captions.shape),))
    out = torch.empty(sum(bs), dtype=torch.int64, device=captions.device)
    check(_lib.lib().capnet_packed_targets(ptr(captions), captions.shape[1], len(bs), int_array(bs),
                                           ptr(out), current_stream()), "capnet_packed_targets")
    return out


def pack_tensors(tensors, flat, unpack=False, scale=1.0):
    """Gather `tensors` into `flat` (or scatter back, scaled)."""
    n = len(tensors)
    if n == 0:
        return
    _need_cuda(flat, *tensors)
    numel = (C.c_long * n)(*[t.numel() for t in tensors])
    if sum(t.numel() for t in tensors) > flat.numel():
        raise CapnetError("pack_tensors: flat buffer too small")
    for t in tensors:
        if not t.is_contiguous():
            raise CapnetError("pack_tensors: tensors must be contiguous")
    check(_lib.lib().capnet_pack_tensors(n, ptr_array(tensors), numel, ptr(flat), int(unpack),
                                         float(scale), current_stream()), "capnet_pack_tensors")


def pack_conv_weight(w_oihw, row_stride, kmajor=False):
    """OIHW -> [Cout][row_stride] rows, or (kmajor) the K-major [row_stride][Cout] image."""
    _need_cuda(w_oihw)
    w = _c(w_oihw)
    co, ci, kh, kw = w.shape
    if kmajor:
        out = torch.empty((row_stride, co), dtype=torch.float32, device=w.device)
        check(_lib.lib().capnet_pack_conv_weight_kmajor(ptr(w), ptr(out), co, ci, kh, kw, row_stride,
                                                        current_stream()), "capnet_pack_conv_weight_kmajor")
        return out
    out = torch.empty((co, row_stride), dtype=torch.float32, device=w.device)
    check(_lib.lib().capnet_pack_conv_weight(ptr(w), ptr(out), co, ci, kh, kw, row_stride,
                                             current_stream()), "capnet_pack_conv_weight")
    return out


def pack_conv_weight_f16x3(w_oihw, bn):
    """1x1 or 3x3 OIHW weights -> header + the split-f16 image of capnet_conv2d_fwd_f16x3 for tile width bn
    (two f16 pieces per weight scaled by the per-tensor power of two in the header, laid out as the
    kernel's LDS image, k = (tap, channel))."""
    _need_cuda(w_oihw)
    w = _c(w_oihw)
    co, ci, kh, kw = w.shape
    if (kh, kw) not in ((1, 1), (3, 3)):
        raise CapnetError("pack_conv_weight_f16x3: 1x1 or 3x3 weights only")
    out = torch.empty(_lib.lib().capnet_conv_f16x3_weight_words(ci, co, kh), dtype=torch.int32, device=w.device)
    check(_lib.lib().capnet_conv_f16x3_pack(ptr(w), ptr(out), co, ci, kh, int(bn), current_stream()),
          "capnet_conv_f16x3_pack")
    return out


def pack_fused_block_weight(w_oihw, role):
    """1x1 OIHW weights -> the image of capnet_fused_block_forward: role 0 = conv3 of a bottleneck ([C][MID][1][1]),
    role 1 = the next block's conv1 ([MID][C][1][1])."""
    _need_cuda(w_oihw)
    w = _c(w_oihw)
    co, ci = w.shape[0], w.shape[1]
    cc, mid = (co, ci) if role == 0 else (ci, co)
    if tuple(w.shape[2:]) != (1, 1) or cc != 4 * mid:
        raise CapnetError("pack_fused_block_weight: role %d expects a 1x1 convolution between MID and 4 MID channels" % role)
    out = torch.empty(_lib.lib().capnet_fused_block_weight_words(cc, mid, role), dtype=torch.int32, device=w.device)
    check(_lib.lib().capnet_fused_block_pack(ptr(w), ptr(out), cc, mid, role, current_stream()), "capnet_fused_block_pack")
    return out


def fused_block_stats(y2, s2, t2, w3img, gamma, beta, running_mean=None, running_var=None, momentum=0.1, eps=1e-5, in_exp=0):
    """(scale, shift) of bn3 behind conv3(relu(y2 s2 + t2)) without forming conv3's output (capnet_fused_block_stats)."""
    _need_cuda(y2, s2, t2, w3img, gamma, beta, running_mean, running_var)
    M, mid = y2.shape
    work = torch.empty(_lib.lib().capnet_fused_block_stats_floats(M, mid), dtype=torch.float32, device=y2.device)
    scale = torch.empty(4 * mid, dtype=torch.float32, device=y2.device)
    shift = torch.empty_like(scale)
    check(_lib.lib().capnet_fused_block_stats(ptr(y2), ptr(s2), ptr(t2), ptr(w3img), M, mid, in_exp, ptr(gamma), ptr(beta),
                                              ptr(running_mean), ptr(running_var), momentum, eps, ptr(scale), ptr(shift),
                                              ptr(work), ptr(err_flag(y2.device)), current_stream()), "capnet_fused_block_stats")
    return scale, shift


def fused_block_forward(y2, s2, t2, w3img, s3, t3, res, w1img, sd=None, td=None, stats=True, e3=0, e1=0):
    """-> (out [M, 4 MID], y1 [M, MID], part_sum, part_sq) (capnet_fused_block_forward)."""
    _need_cuda(y2, s2, t2, w3img, s3, t3, res, w1img, sd, td)
    M, mid = y2.shape
    dev = y2.device
    out = torch.empty(M, 4 * mid, dtype=torch.float32, device=dev)
    y1 = torch.empty(M, mid, dtype=torch.float32, device=dev)
    tiles = _lib.lib().capnet_fused_block_tiles(M, mid)
    ps = torch.empty(tiles, mid, dtype=torch.float32, device=dev) if stats else None
    pq = torch.empty(tiles, mid, dtype=torch.float32, device=dev) if stats else None
    check(_lib.lib().capnet_fused_block_forward(ptr(y2), ptr(s2), ptr(t2), ptr(w3img), ptr(s3), ptr(t3), ptr(res), ptr(sd),
                                                ptr(td), ptr(out), ptr(w1img), ptr(y1), ptr(ps), ptr(pq), M, mid, e3, e1,
                                                ptr(err_flag(dev)), current_stream()), "capnet_fused_block_forward")
    return out, y1, ps, pq


def pack_conv_weight_stem_f16x3(w_oihw):
    """The stem's OIHW weights [64][3][7][7] -> header + the split-f16 image of capnet_conv_stem_fwd_f16x3."""
    _need_cuda(w_oihw)
    w = _c(w_oihw)
    if tuple(w.shape) != (64, 3, 7, 7):
        raise CapnetError("pack_conv_weight_stem_f16x3: weights must be [64][3][7][7]")
    out = torch.empty(_lib.lib().capnet_conv_stem_f16x3_weight_words(), dtype=torch.int32, device=w.device)
    check(_lib.lib().capnet_conv_stem_f16x3_pack(ptr(w), ptr(out), current_stream()), "capnet_conv_stem_f16x3_pack")
    return out


def clamp_adam(params, grads, exp_avg, exp_avg_sq, steps, lr, beta1, beta2, eps, clip,
               write_grad=True):
    """Fused element-wise clamp + Adam over a list of tensors (in place); a no-op on the device while the
    device's error word is set (check_device_errors)."""
    n = len(params)
    if n == 0:
        return
    _need_cuda(*params, *grads, *exp_avg, *exp_avg_sq)
    for p, g in zip(params, grads):
        if not (p.is_contiguous() and g.is_contiguous()) or p.numel() != g.numel():
            raise CapnetError("clamp_adam: parameters and gradients must be contiguous and equal-sized")
    numel = (C.c_long * n)(*[p.numel() for p in params])
    check(_lib.lib().capnet_clamp_adam(n, ptr_array(params), ptr_array(grads), ptr_array(exp_avg),
                                       ptr_array(exp_avg_sq), numel, int_array(steps), lr, beta1,
                                       beta2, eps, clip if clip else 0.0, int(write_grad),
                                       ptr(err_flag(params[0].device)), current_stream()), "capnet_clamp_adam")


# ---------------------------------------------------------------------------------------
# autograd: nn.Linear
# ---------------------------------------------------------------------------------------
class LinearFn(torch.autograd.Function):
    """y = x @ w.T + b   (x [M,K], w [N,K])."""

    @staticmethod
    def forward(ctx, x, w, b):
        _need_cuda(x, w, b)
        x, w = _c(x), _c(w)
        ctx.save_for_backward(x, w)
        ctx.has_bias = b is not None
        # (through the K-split entry: the encoder head's 64 x 300 x 2048 is five 64 x 64 tiles walking the whole K on the
        #  plain one -- 94 us; large products go to the same kernels either way)
        return sgemm_splitk(x, w, transB=True, bias=b)

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dy = _c(dy)
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = sgemm_splitk(dy, w)               # [M,N] @ [N,K]; K-split when N (vocab) is long
        if ctx.needs_input_grad[1]:
            dw = sgemm(dy, x, transA=True)         # dy^T [N,M] @ x [M,K]
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = colsum(dy)
        return dx, dw, db


def linear(x, w, b=None):
    return LinearFn.apply(x, w, b)


# ---------------------------------------------------------------------------------------
# autograd: nn.CrossEntropyLoss (mean)
# ---------------------------------------------------------------------------------------
class CrossEntropyFn(torch.autograd.Function):

    @staticmethod
    def forward(ctx, logits, targets):
        _need_cuda(logits, targets)
        logits, targets = _c(logits), _c(targets)
        if targets.dtype != torch.int64:
            raise CapnetError("targets must be int64")
        n, v = logits.shape
        if targets.numel() != n:
            raise CapnetError("cross entropy: %d logits rows vs %d targets" % (n, targets.numel()))
        lse = torch.empty(n, dtype=torch.float32, device=logits.device)
        row_loss = torch.empty(n, dtype=torch.float32, device=logits.device)
        loss = torch.empty((), dtype=torch.float32, device=logits.device)
        check(_lib.lib().capnet_xent_fwd(ptr(logits), v, n, v, ptr(targets), ptr(lse),
                                         ptr(row_loss), ptr(loss), ptr(err_flag(logits.device)),
                                         current_stream()), "capnet_xent_fwd")
        ctx.save_for_backward(logits, targets, lse)
        return loss

    @staticmethod
    @once_differentiable
    def backward(ctx, gout):
        logits, targets, lse = ctx.saved_tensors
        n, v = logits.shape
        gout = _c(gout.to(torch.float32)).reshape(1)
        dlogits = torch.empty_like(logits)
        check(_lib.lib().capnet_xent_bwd(ptr(logits), v, n, v, ptr(targets), ptr(lse), ptr(gout),
                                         ptr(dlogits), v, current_stream()), "capnet_xent_bwd")
        return dlogits, None


def cross_entropy(logits, targets):
    return CrossEntropyFn.apply(logits, targets)


class AttentionLossFn(torch.autograd.Function):
    """nll + alpha_c * ((1 - alphas.sum(dim=1)) ** 2).mean()  (train_multitask_att.py:409-411)."""

    @staticmethod
    def forward(ctx, nll, alphas, alpha_c):
        _need_cuda(nll, alphas)
        alphas = _c(alphas)
        B, steps, P = alphas.shape
        colsum = torch.empty(B * P, dtype=torch.float32, device=alphas.device)
        out = torch.empty((), dtype=torch.float32, device=alphas.device)
        check(_lib.lib().capnet_att_loss_fwd(ptr(_c(nll.reshape(1))), ptr(alphas), B, steps, P, float(alpha_c),
                                             ptr(colsum), ptr(out), current_stream()), "capnet_att_loss_fwd")
        ctx.dims, ctx.alpha_c = (B, steps, P), float(alpha_c)
        ctx.save_for_backward(colsum)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, gout):
        (colsum,) = ctx.saved_tensors
        B, steps, P = ctx.dims
        gout = _c(gout.to(torch.float32)).reshape(1)
        dalphas = torch.empty((B, steps, P), dtype=torch.float32, device=colsum.device)
        check(_lib.lib().capnet_att_loss_bwd(ptr(gout), ptr(colsum), B, steps, P, ctx.alpha_c, ptr(dalphas),
                                             current_stream()), "capnet_att_loss_bwd")
        return gout.reshape(()), dalphas, None


def attention_loss(nll, alphas, alpha_c=1.0):
    return AttentionLossFn.apply(nll, alphas, alpha_c)


# ---------------------------------------------------------------------------------------
# autograd: nn.BatchNorm1d (encoder head)
# ---------------------------------------------------------------------------------------
class BatchNorm1dFn(torch.autograd.Function):

    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, train, momentum, eps):
        _need_cuda(x, gamma, beta, running_mean, running_var)
        x = _c(x)
        b, c = x.shape
        y = torch.empty_like(x)
        mean = torch.empty(c, dtype=torch.float32, device=x.device)
        invstd = torch.empty(c, dtype=torch.float32, device=x.device)
        check(_lib.lib().capnet_bn1d_fwd(ptr(x), b, c, ptr(gamma), ptr(beta), ptr(running_mean),
                                         ptr(running_var), int(train), momentum, eps, ptr(y),
                                         ptr(mean), ptr(invstd), current_stream()),
              "capnet_bn1d_fwd")
        ctx.train = train
        ctx.save_for_backward(x, gamma, mean, invstd)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        if not ctx.train:
            raise CapnetError("BatchNorm1d backward is implemented for train mode only")
        x, gamma, mean, invstd = ctx.saved_tensors
        dy = _c(dy)
        b, c = x.shape
        dx = torch.empty_like(x)
        dgamma = torch.empty_like(gamma)
        dbeta = torch.empty_like(gamma)
        check(_lib.lib().capnet_bn1d_bwd(ptr(dy), ptr(x), b, c, ptr(gamma), ptr(mean), ptr(invstd),
                                         ptr(dx), ptr(dgamma), ptr(dbeta), current_stream()),
              "capnet_bn1d_bwd")
        return dx, dgamma, dbeta, None, None, None, None, None


def batch_norm1d(x, gamma, beta, running_mean, running_var, train, momentum, eps):
    return BatchNorm1dFn.apply(x, gamma, beta, running_mean, running_var, train, momentum, eps)


# ---------------------------------------------------------------------------------------
# autograd: whole-sequence decoder recurrence
# ---------------------------------------------------------------------------------------
def batch_sizes_from_lengths(lengths):
    """pack_padded_sequence(..., lengths).batch_sizes for lengths sorted in decreasing order."""
    lengths = [int(l) for l in lengths]
    if not lengths or any(l <= 0 for l in lengths):
        raise CapnetError("lengths must be positive")
    if any(lengths[i] < lengths[i + 1] for i in range(len(lengths) - 1)):
        raise CapnetError("lengths must be sorted in decreasing order (pack_padded_sequence contract)")
    return [sum(1 for l in lengths if l > t) for t in range(lengths[0])]


class DecoderSeqFn(torch.autograd.Function):
    """hiddens[N,H] of the scheduled-sampling recurrence; one C call forward, one backward.

    weights: cell 0 -> 32 tensors (V w x4, V b x4, S w x4, S b x4, U w x4, U b x4, W w x4, W b x4)
             cell 1 -> (weight_ih, bias_ih, weight_hh, bias_hh)
    """

    @staticmethod
    def forward(ctx, cfg, captions, features, emb, Cw, Cb, *weights):
        cell = cfg["cell"]
        _need_cuda(captions, features, emb, Cw, Cb, *weights)
        captions = _c(captions)
        if captions.dtype != torch.int64:
            raise CapnetError("captions must be int64")
        dev = emb.device
        bs = cfg["batch_sizes"]
        tf = cfg["tf_mask"]
        B, T = captions.shape
        V, E = emb.shape
        H = cfg["hidden_size"]
        F = cfg.get("factored_size", 0)
        N = sum(bs)
        if len(tf) != len(bs):
            raise CapnetError("tf_mask has %d entries for %d steps" % (len(tf), len(bs)))
        if bs[0] != B:
            raise CapnetError("batch_sizes[0]=%d but captions has %d rows" % (bs[0], B))
        dims = [B, T, len(bs), N, E, F, H, V, int(features is not None), cell]
        ws = [_c(w) for w in weights]
        if cell == CELL_FACTORED:
            if len(ws) != 32:
                raise CapnetError("factored cell takes 32 weight tensors")
            wptrs = ws
        else:
            if len(ws) != 4:
                raise CapnetError("LSTM cell takes 4 weight tensors")
            wptrs = [None] * 32
            wptrs[0], wptrs[4], wptrs[24], wptrs[28] = ws
        if features is not None:
            features = _c(features)
            if tuple(features.shape) != (B, E):
                raise CapnetError("features must be [batch, embed_size]")
        emb_c, Cw_c, Cb_c = _c(emb), _c(Cw), _c(Cb)
        cdims = int_array(dims)
        L = _lib.lib()
        saved = torch.empty(L.capnet_seq_saved_floats(cdims), dtype=torch.float32, device=dev)
        saved_i = torch.empty(L.capnet_seq_saved_ints(cdims), dtype=torch.int32, device=dev)
        scratch = torch.empty(L.capnet_seq_fwd_scratch_floats(cdims), dtype=torch.float32, device=dev)
        hiddens = torch.empty((N, H), dtype=torch.float32, device=dev)
        tfm = (C.c_ubyte * len(tf))(*[1 if x else 0 for x in tf])
        check(L.capnet_seq_forward(cdims, int_array(bs), tfm, ptr(captions), ptr(features),
                                   ptr(emb_c), ptr_array(wptrs), ptr(Cw_c), ptr(Cb_c),
                                   float(cfg["dropout"]), int(cfg["seed"]), int(cfg["training"]),
                                   ptr(saved), ptr(saved_i), ptr(scratch), ptr(hiddens),
                                   ptr(err_flag(dev)), current_stream()), "capnet_seq_forward")
        ctx.cfg = cfg
        ctx.dims = dims
        ctx.n_weights = len(ws)
        ctx.has_features = features is not None
        ctx.save_for_backward(saved, saved_i, hiddens)
        return hiddens

    @staticmethod
    @once_differentiable
    def backward(ctx, d_hiddens):
        saved, saved_i, hiddens = ctx.saved_tensors
        cfg, dims = ctx.cfg, ctx.dims
        B, T, steps, N, E, F, H, V, _, cell = dims
        dev = saved.device
        d_hiddens = _c(d_hiddens)
        L = _lib.lib()
        cdims = int_array(dims)
        scratch = torch.empty(L.capnet_seq_bwd_scratch_floats(cdims), dtype=torch.float32, device=dev)

        def new(*shape):
            return torch.empty(shape, dtype=torch.float32, device=dev)

        dEmb = new(V, E)
        dFeat = new(B, E) if ctx.has_features else None
        dW = new(4 * H, H)
        dbUW = new(4 * H)
        if cell == CELL_FACTORED:
            dV, dbV, dS, dbS, dU = new(4 * F, E), new(4 * F), new(4, F, F), new(4 * F), new(4, H, F)
        else:
            dV, dbV, dS, dbS, dU = new(4 * H, E), None, None, None, None
        grads = [dV, dbV, dS, dbS, dU, dbUW, dW, dEmb, dFeat]
        check(L.capnet_seq_backward(cdims, int_array(cfg["batch_sizes"]), ptr(d_hiddens),
                                    ptr(hiddens), ptr(saved), ptr(saved_i), ptr(scratch),
                                    ptr_array(grads), float(cfg["dropout"]), int(cfg["seed"]),
                                    int(cfg["training"]), current_stream()), "capnet_seq_backward")
        if cell == CELL_FACTORED:
            wg = ([dV[g * F:(g + 1) * F] for g in range(4)] +
                  [dbV[g * F:(g + 1) * F] for g in range(4)] +
                  [dS[g] for g in range(4)] +
                  [dbS[g * F:(g + 1) * F] for g in range(4)] +
                  [dU[g] for g in range(4)] +
                  [dbUW[g * H:(g + 1) * H] for g in range(4)] +
                  [dW[g * H:(g + 1) * H] for g in range(4)] +
                  [dbUW[g * H:(g + 1) * H].clone() for g in range(4)])
        else:
            wg = [dV, dbUW, dW, dbUW.clone()]
        # cfg, captions, features, emb, Cw, Cb, *weights
        return (None, None, dFeat, dEmb, None, None) + tuple(wg)


def decoder_sequence(cfg, captions, features, emb, Cw, Cb, weights):
    return DecoderSeqFn.apply(cfg, captions, features, emb, Cw, Cb, *weights)


# ---------------------------------------------------------------------------------------
# autograd: whole-sequence attention decoder (DecoderFactoredLSTMAtt)
# ---------------------------------------------------------------------------------------
def _att_slots(ws, cell):
    """The 44-slot weight table of capnet_att_seq_forward/backward."""
    if cell == CELL_FACTORED:
        return ws
    slots = [None] * 32
    slots[0], slots[4], slots[24], slots[28] = ws[0], ws[1], ws[2], ws[3]
    return slots + list(ws[4:])


class DecoderAttSeqFn(torch.autograd.Function):
    """(hiddens [N,H], alphas [B,steps,P]) of the attention recurrence.

    weights, cfg["cell"] == CELL_FACTORED (default): 44 tensors in the order of
    capnet_att_seq_forward (V w x4, V b x4, S w x4, S b x4, U w x4, U b x4, W w x4, W b x4,
    init_h w,b, init_c w,b, encoder_att w,b, decoder_att w,b, full_att w,b, f_beta w,b).
    cfg["cell"] == CELL_LSTM (nic DecoderRNNAtt): 16 tensors weight_ih, bias_ih, weight_hh,
    bias_hh followed by the same 12 attention / init tensors.
    `features` gets no gradient (frozen trunk)."""

    @staticmethod
    def forward(ctx, cfg, captions, features, emb, Cw, Cb, *weights):
        _need_cuda(captions, features, emb, Cw, Cb, *weights)
        cell = cfg.get("cell", CELL_FACTORED)
        if len(weights) != (44 if cell == CELL_FACTORED else 16):
            raise CapnetError("attention decoder takes 44 (factored) / 16 (LSTMCell) weight tensors")
        captions = _c(captions)
        if captions.dtype != torch.int64:
            raise CapnetError("captions must be int64")
        dev = emb.device
        bs, tf = cfg["batch_sizes"], cfg["tf_mask"]
        B, T = captions.shape
        V, E = emb.shape
        H, A = cfg["hidden_size"], cfg["attention_size"]
        F = cfg["factored_size"] if cell == CELL_FACTORED else 4
        features = _c(features)
        if features.dim() != 3 or features.shape[0] != B:
            raise CapnetError("features must be [batch, pixels, feature_size]")
        P, Cf = features.shape[1], features.shape[2]
        N = sum(bs)
        if len(tf) != len(bs) or bs[0] != B:
            raise CapnetError("attention decoder: batch_sizes / tf_mask do not match the batch")
        dims = [B, T, len(bs), N, E, F, H, V, A, P, Cf, cell]
        ws = [_c(w) for w in weights]
        emb_c, Cw_c, Cb_c = _c(emb), _c(Cw), _c(Cb)
        cdims = int_array(dims)
        L = _lib.lib()
        saved = torch.empty(L.capnet_att_saved_floats(cdims), dtype=torch.float32, device=dev)
        saved_i = torch.empty(L.capnet_att_saved_ints(cdims), dtype=torch.int32, device=dev)
        scratch = torch.empty(L.capnet_att_fwd_scratch_floats(cdims), dtype=torch.float32, device=dev)
        hiddens = torch.empty((N, H), dtype=torch.float32, device=dev)
        alphas = torch.empty((B, len(bs), P), dtype=torch.float32, device=dev)
        tfm = (C.c_ubyte * len(tf))(*[1 if x else 0 for x in tf])
        check(L.capnet_att_seq_forward(cdims, int_array(bs), tfm, ptr(captions), ptr(features),
                                       ptr(emb_c), ptr_array(_att_slots(ws, cell)), ptr(Cw_c), ptr(Cb_c),
                                       float(cfg["dropout"]), int(cfg["seed"]), int(cfg["training"]),
                                       ptr(saved), ptr(saved_i), ptr(scratch), ptr(hiddens),
                                       ptr(alphas), ptr(err_flag(dev)), current_stream()),
              "capnet_att_seq_forward")
        ctx.cfg, ctx.dims = cfg, dims
        ctx.save_for_backward(saved, saved_i, hiddens, features, *ws)
        return hiddens, alphas

    @staticmethod
    @once_differentiable
    def backward(ctx, d_hiddens, d_alphas):
        saved, saved_i, hiddens, features = ctx.saved_tensors[:4]
        ws = list(ctx.saved_tensors[4:])
        cfg, dims = ctx.cfg, ctx.dims
        B, T, steps, N, E, F, H, V, A, P, Cf, cell = dims
        dev = saved.device
        L = _lib.lib()
        cdims = int_array(dims)
        scratch = torch.empty(L.capnet_att_bwd_scratch_floats(cdims), dtype=torch.float32, device=dev)
        d_hiddens = _c(d_hiddens)
        d_alphas = _c(d_alphas) if d_alphas is not None else None

        def new(*shape):
            return torch.empty(shape, dtype=torch.float32, device=dev)

        ZW, XW = 4 * H + A + Cf, E + Cf
        if cell == CELL_FACTORED:
            dV, dbV, dS, dbS, dU = new(4 * F, XW), new(4 * F), new(4, F, F), new(4 * F), new(4, H, F)
        else:
            dV, dbV, dS, dbS, dU = new(4 * H, XW), None, None, None, None
        dWz, dbz = new(ZW, H), new(ZW)
        dWe, dbe, dwf, dbf = new(A, Cf), new(A), new(1, A), new(1)
        dWih, dbih, dWic, dbic = new(H, Cf), new(H), new(H, Cf), new(H)
        dEmb = new(V, E)
        grads = [dV, dbV, dS, dbS, dU, dWz, dbz, dWe, dbe, dwf, dbf, dWih, dbih, dWic, dbic, dEmb]
        check(L.capnet_att_seq_backward(cdims, int_array(cfg["batch_sizes"]), ptr(d_hiddens),
                                        ptr(d_alphas), ptr(hiddens), ptr(features),
                                        ptr_array(_att_slots(ws, cell)),
                                        ptr(saved), ptr(saved_i), ptr(scratch), ptr_array(grads),
                                        float(cfg["dropout"]), int(cfg["seed"]), int(cfg["training"]),
                                        current_stream()), "capnet_att_seq_backward")
        tail = [dWih, dbih, dWic, dbic, dWe, dbe,
                dWz[4 * H:4 * H + A], dbz[4 * H:4 * H + A], dwf, dbf,
                dWz[4 * H + A:], dbz[4 * H + A:]]
        if cell != CELL_FACTORED:
            wg = [dV, dbz[:4 * H], dWz[:4 * H], dbz[:4 * H].clone()] + tail
            return (None, None, None, dEmb, None, None) + tuple(wg)
        wg = ([dV[g * F:(g + 1) * F] for g in range(4)] +
              [dbV[g * F:(g + 1) * F] for g in range(4)] +
              [dS[g] for g in range(4)] +
              [dbS[g * F:(g + 1) * F] for g in range(4)] +
              [dU[g] for g in range(4)] +
              [dbz[g * H:(g + 1) * H] for g in range(4)] +
              [dWz[g * H:(g + 1) * H] for g in range(4)] +
              [dbz[g * H:(g + 1) * H].clone() for g in range(4)] +
              [dWih, dbih, dWic, dbic, dWe, dbe,
               dWz[4 * H:4 * H + A], dbz[4 * H:4 * H + A], dwf, dbf,
               dWz[4 * H + A:], dbz[4 * H + A:]])
        # cfg, captions, features, emb, Cw, Cb, *weights
        return (None, None, None, dEmb, None, None) + tuple(wg)


def decoder_att_sequence(cfg, captions, features, emb, Cw, Cb, weights):
    return DecoderAttSeqFn.apply(cfg, captions, features, emb, Cw, Cb, *weights)
