"""Stacked FactoredLSTM decoder for BASELINE configs[3] / [4] ("2-layer", "3-layer").

PERF-ONLY, PARITY UNPINNED. The reference advertises `lstm_layers: 1, 2, 3` (README.md:24, BLEU rows :63,:66) but its
decoders accept `num_layers` and ignore it (stylenet/model.py:37, model_att.py:81): there is no stacking code in the
tree to be equal to. The semantics built here are SURVEY.md App. A-1's, modelled on the only stacking the tree has
(seq2seq/model.py:45-49, nn.LSTM(num_layers) = each layer reads the layer below at the same time step):

  * layer 0 is the reference's cell with the reference's parameter names (B, U_g, S_{mode}g, V_g: Linear(E -> F), W_g);
  * layer l > 0 is the same factored cell on the hidden state of the layer below at the same step, with its own
    V{l}_g: Linear(H -> F), S{l}_{mode}g, U{l}_g, W{l}_g, and dropout between the layers;
  * only the top layer feeds C (the packed logits and, on free-running steps, the argmax that is fed back);
  * everything else -- feature prepended as step 0, one teacher-forcing draw per step, shrinking batches, the feedback
    embedding without dropout -- is stylenet/model.py:157-196.

With num_layers = 1 it is DecoderFactoredLSTM's function (tests/test_stacked_gpu.py); the numbers for more layers are
checked against the same definition restated on the CPU (oracle/decoders_ref.py: stacked_factored_lstm_forward).

Engine (default, `engine = "c"`): the whole stacked recurrence is ONE C call each way (StackedSeqFn ->
capnet_seq_forward_stacked / capnet_seq_backward_stacked, csrc/decoder_seq.cpp: the single-layer driver with a layer loop
inside; embedding gather, dropout masks, chains, persistent runs, BPTT are all kernels of the library).
The op-by-op engines of round 3 stay for comparison (`engine = "python"`): runs of teacher-forced steps are processed layer by layer -- a layer's input chain U(S(V x)) over all rows of
the run as three MFMA GEMMs (the four gates' layers stacked: GateLinearFn), then the run's recurrence in ONE launch of the persistent kernel with its
own backward through time (LstmRunFn below; H = 512, <= 128 rows) -- or, for a lone step and for sizes the persistent
kernel does not take, step by step (W GEMM + the fused cell kernel, ops.lstm_cell, torch autograd composing the
backward). torch is glue here (embedding gather, dropout masks, bias adds, concatenations). The single-layer path's
fused sequence driver (csrc/decoder_seq.cpp) is still the faster structure; it does not apply to stacked cells yet
(DESIGN 7)."""
import random
import sys

import torch
import torch.nn as nn
import torch.nn.functional as Fn

from torch.autograd.function import once_differentiable

from . import _lib, ops
from ._lib import CapnetError, check, current_stream, int_array, ptr
from .model import Embedding as _Embedding, Linear as _Linear

MODES = ("factual", "happy", "sad", "angry")
_S_PREFIX = {"factual": "f", "happy": "happy_", "sad": "sad_", "angry": "angry_"}


class LstmRunFn(torch.autograd.Function):
    """A run of recurrent steps of one LSTM layer in ONE launch of the persistent kernel (csrc/lstm_persist.hip), with
    its backward through time: the recurrence of stylenet/model.py:147-153,180-191 (cell 0) / nn.LSTMCell (cell 1) over
    steps whose inputs are all known up front.

        pre   [sum(bs), 4H]  pre-activations of the steps without the recurrent product (all biases included), packed
                             step-major (rows of step t: bs[t], non-increasing)
        w     [4H, H]        the recurrent weights (gate blocks in the cell's order)
        h0, c0 [bs[0], H]     state in front of the first step
    -> hiddens [sum(bs), H] (packed like pre), c_last [bs[-1], H]

    Backward: per step (last to first) one gate-backward kernel and one product dPre . w for the state gradient,
    then the weight gradient over all rows at once."""

    @staticmethod
    def forward(ctx, pre, w, h0, c0, bs, cell):
        ops._need_cuda(pre, w, h0, c0)
        L = _lib.lib()
        n, rows, H = len(bs), sum(bs), w.shape[1]
        b0 = bs[0]
        if pre.shape != (rows, 4 * H) or h0.shape != (b0, H) or c0.shape != (b0, H) or w.shape[0] != 4 * H:
            raise CapnetError("lstm_run: shapes do not match the batch sizes")
        if not L.capnet_lstm_persist_supported(b0, H) or n + 1 > 128:
            raise CapnetError("lstm_run: the persistent kernel does not take b=%d H=%d steps=%d" % (b0, H, n))
        dev = pre.device
        # step 0 of the buffers is the incoming state: the kernel picks h, c of step t0 - 1 up from the output buffers
        G = torch.empty((b0 + rows, 4 * H), dtype=torch.float32, device=dev)
        G[b0:].copy_(pre.detach())
        hid = torch.empty((b0 + rows, H), dtype=torch.float32, device=dev)
        cst = torch.empty((b0 + rows, H), dtype=torch.float32, device=dev)
        hid[:b0].copy_(h0.detach())
        cst[:b0].copy_(c0.detach())
        img = torch.empty(L.capnet_lstm_persist_w_floats(), dtype=torch.float32, device=dev)
        wc = w.detach().contiguous()
        check(L.capnet_lstm_persist_pack(ptr(wc), ptr(img), cell, current_stream()), "capnet_lstm_persist_pack")
        ctl = torch.zeros(L.capnet_lstm_persist_ctl_ints(), dtype=torch.int32, device=dev)
        full = [b0] + list(bs)
        check(L.capnet_lstm_persist_run(ptr(img), ptr(G), ptr(cst), ptr(hid), int_array(full), 1, n + 1, H, cell, 1,
                                        ptr(ctl), ptr(ops.err_flag(dev)), None, current_stream()), "capnet_lstm_persist_run")
        ctx.save_for_backward(G, cst, hid, wc)
        ctx.full, ctx.cell = full, cell
        return hid[b0:], cst[b0 + rows - bs[-1]:]

    @staticmethod
    @once_differentiable
    def backward(ctx, d_hid, d_clast):
        G, cst, hid, w = ctx.saved_tensors
        full, cell = ctx.full, ctx.cell
        L = _lib.lib()
        n, b0, H = len(full) - 1, full[0], w.shape[1]
        off = [0]
        for v in full:
            off.append(off[-1] + v)
        rows = off[-1] - b0
        dev = G.device
        d_hid = d_hid.contiguous() if d_hid is not None else torch.zeros((rows, H), dtype=torch.float32, device=dev)
        dpre = torch.empty((rows, 4 * H), dtype=torch.float32, device=dev)
        dh_rec = torch.zeros((b0, H), dtype=torch.float32, device=dev)      # rows beyond a step's batch: samples that
        dc = torch.zeros((b0, H), dtype=torch.float32, device=dev)          # have ended -- their later steps give nothing
        if d_clast is not None:
            dc[:full[-1]] += d_clast
        for t in range(n, 0, -1):
            bt, r0 = full[t], off[t]
            dh = d_hid[r0 - b0:r0 - b0 + bt] + dh_rec[:bt]
            cp = cst[off[t - 1]:off[t - 1] + bt]
            dp = dpre[r0 - b0:r0 - b0 + bt]
            check(L.capnet_lstm_pointwise_bwd(ptr(G[r0:r0 + bt]), ptr(cst[r0:r0 + bt]), ptr(cp), ptr(dh), ptr(dc[:bt]),
                                              ptr(dp), bt, H, cell, current_stream()), "capnet_lstm_pointwise_bwd")
            dh_rec[:bt] = ops.sgemm_splitk(dp, w)                                 # [bt, 4H] @ [4H, H]
        hprev = torch.cat([hid[off[t - 1]:off[t - 1] + full[t]] for t in range(1, n + 1)], 0)
        dw = ops.sgemm(dpre, hprev, transA=True)                                 # [4H, rows] @ [rows, H]
        return dpre, dw, dh_rec, dc, None, None


def lstm_run(pre, w, h0, c0, batch_sizes, cell=0):
    return LstmRunFn.apply(pre, w, h0, c0, [int(v) for v in batch_sizes], cell)


def lstm_run_supported(b, H, steps):
    return bool(_lib.lib().capnet_lstm_persist_supported(int(b), int(H))) and 2 <= steps < 128


class GateLinearFn(torch.autograd.Function):
    """The four gates' nn.Linear layers of one stage of the factored chain (S_g or U_g, stylenet/model.py:119-150) as
    ONE batched product: y[:, g-th block of `out`] = x[:, g-th block of `inp`] @ w[g]^T + b[g].
        x [rows, 4*inp], w [4, out, inp] (the four weights stacked), b [4*out]  ->  y [rows, 4*out]
    Forward, d x and d w are one capnet_sgemm call each (batch = 4 over column blocks), d b a column sum."""

    @staticmethod
    def forward(ctx, x, w, b):
        ops._need_cuda(x, w, b)
        x, w = x.contiguous(), w.contiguous()
        rows, out, inp = x.shape[0], w.shape[1], w.shape[2]
        if w.shape[0] != 4 or x.shape[1] != 4 * inp or b.shape[0] != 4 * out:
            raise CapnetError("gate_linear: shapes")
        y = torch.empty((rows, 4 * out), dtype=torch.float32, device=x.device)
        check(_lib.lib().capnet_sgemm(0, 1, rows, out, inp, ptr(x), 4 * inp, ptr(w), inp, ptr(y), 4 * out, ptr(b), 0,
                                      4, inp, out * inp, out, out, 0, current_stream()), "capnet_sgemm")
        ctx.save_for_backward(x, w)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dy = dy.contiguous()
        rows, out, inp = x.shape[0], w.shape[1], w.shape[2]
        L = _lib.lib()
        dx = torch.empty_like(x)
        check(L.capnet_sgemm(0, 0, rows, inp, out, ptr(dy), 4 * out, ptr(w), inp, ptr(dx), 4 * inp, None, 0,
                             4, out, out * inp, inp, 0, 0, current_stream()), "capnet_sgemm")
        dw = torch.empty_like(w)
        check(L.capnet_sgemm(1, 0, out, inp, rows, ptr(dy), 4 * out, ptr(x), 4 * inp, ptr(dw), inp, None, 0,
                             4, out, inp, out * inp, 0, 0, current_stream()), "capnet_sgemm")
        return dx, dw, ops.colsum(dy)


class StackedSeqFn(torch.autograd.Function):
    """The whole stacked recurrence as ONE C call each way (capnet_seq_forward_stacked / capnet_seq_backward_stacked,
    csrc/decoder_seq.cpp): embedding gather, dropout masks (input and between layers), the gate chains, the persistent
    runs, free-running steps and the full BPTT are kernels of the library; no torch operator on the path.
        weights: num_layers x 32 tensors in DecoderSeqFn's order (V w x4, V b x4, S w x4, S b x4, U w x4, U b x4,
        W w x4, W b x4), layer 0 first.  -> the top layer's hiddens [N, H] in pack_padded_sequence order."""

    @staticmethod
    def forward(ctx, cfg, captions, features, emb, Cw, Cb, *weights):
        ops._need_cuda(captions, features, emb, Cw, Cb, *weights)
        nl = cfg["num_layers"]
        if len(weights) != 32 * nl:
            raise CapnetError("stacked decoder takes 32 weight tensors per layer")
        captions = captions.contiguous()
        if captions.dtype != torch.int64:
            raise CapnetError("captions must be int64")
        dev = emb.device
        bs, tf = cfg["batch_sizes"], cfg["tf_mask"]
        B, T = captions.shape
        V, E = emb.shape
        H, F = cfg["hidden_size"], cfg["factored_size"]
        N = sum(bs)
        if len(tf) != len(bs) or bs[0] != B:
            raise CapnetError("stacked decoder: batch_sizes / tf_mask do not match the batch")
        if features is not None:
            features = features.contiguous()
            if tuple(features.shape) != (B, E):
                raise CapnetError("features must be [batch, embed_size]")
        dims = [[B, T, len(bs), N, E if l == 0 else H, F, H, V, int(features is not None) if l == 0 else 0, ops.CELL_FACTORED]
                for l in range(nl)]
        ws = [w.contiguous() for w in weights]
        emb_c, Cw_c, Cb_c = emb.contiguous(), Cw.contiguous(), Cb.contiguous()
        L = _lib.lib()
        cd = [int_array(d) for d in dims]
        saved = [torch.empty(L.capnet_seq_saved_floats(c), dtype=torch.float32, device=dev) for c in cd]
        saved_i = [torch.empty(L.capnet_seq_saved_ints(c), dtype=torch.int32, device=dev) for c in cd]
        scratch = torch.empty(L.capnet_seq_fwd_scratch_floats(cd[0]), dtype=torch.float32, device=dev)
        hid = [torch.empty((N, H), dtype=torch.float32, device=dev) for _ in range(nl)]
        tfm = (_lib.C.c_ubyte * len(tf))(*[1 if x else 0 for x in tf])
        check(L.capnet_seq_forward_stacked(cd[0], nl, int_array(bs), tfm, ptr(captions), ptr(features), ptr(emb_c),
                                           _lib.ptr_array(ws), ptr(Cw_c), ptr(Cb_c), float(cfg["dropout"]), int(cfg["seed"]),
                                           int(cfg["training"]), _lib.ptr_array(saved), _lib.ptr_array(saved_i), ptr(scratch),
                                           _lib.ptr_array(hid), ptr(ops.err_flag(dev)), current_stream()),
              "capnet_seq_forward_stacked")
        ctx.cfg, ctx.dims, ctx.has_features = cfg, dims, features is not None
        ctx.save_for_backward(*(saved + saved_i + hid))
        return hid[-1]

    @staticmethod
    @once_differentiable
    def backward(ctx, d_hiddens):
        cfg, dims = ctx.cfg, ctx.dims
        nl = cfg["num_layers"]
        t = ctx.saved_tensors
        saved, saved_i, hid = list(t[:nl]), list(t[nl:2 * nl]), list(t[2 * nl:])
        B, T, steps, N, E, F, H, V, _, _ = dims[0]
        dev = saved[0].device
        L = _lib.lib()
        cd = [int_array(d) for d in dims]
        scratch = torch.empty(max(L.capnet_seq_bwd_scratch_floats(c) for c in cd), dtype=torch.float32, device=dev)

        def new(*shape):
            return torch.empty(shape, dtype=torch.float32, device=dev)

        dEmb = new(V, E)
        dFeat = new(B, E) if ctx.has_features else None
        grads, per_layer = [], []
        for l in range(nl):
            El = E if l == 0 else H
            g = [new(4 * F, El), new(4 * F), new(4, F, F), new(4 * F), new(4, H, F), new(4 * H), new(4 * H, H),
                 dEmb if l == 0 else None, dFeat if l == 0 else None]
            grads += g
            per_layer.append(g)
        dh_work = [new(N, H) for _ in range(nl - 1)]
        check(L.capnet_seq_backward_stacked(cd[0], nl, int_array(cfg["batch_sizes"]), ptr(d_hiddens.contiguous()),
                                            _lib.ptr_array(hid), _lib.ptr_array(saved), _lib.ptr_array(saved_i), ptr(scratch),
                                            _lib.ptr_array(dh_work) if dh_work else None, _lib.ptr_array(grads),
                                            float(cfg["dropout"]), int(cfg["seed"]), int(cfg["training"]), current_stream()),
              "capnet_seq_backward_stacked")
        wg = []
        for dV, dbV, dS, dbS, dU, dbUW, dW, _, _ in per_layer:
            wg += ([dV[g * F:(g + 1) * F] for g in range(4)] + [dbV[g * F:(g + 1) * F] for g in range(4)] +
                   [dS[g] for g in range(4)] + [dbS[g * F:(g + 1) * F] for g in range(4)] + [dU[g] for g in range(4)] +
                   [dbUW[g * H:(g + 1) * H] for g in range(4)] + [dW[g * H:(g + 1) * H] for g in range(4)] +
                   [dbUW[g * H:(g + 1) * H].clone() for g in range(4)])
        # cfg, captions, features, emb, Cw, Cb, *weights
        return (None, None, dFeat, dEmb, None, None) + tuple(wg)


class StackedFactoredLSTM(nn.Module):
    """DecoderFactoredLSTM(embed_size, hidden_size, factored_size, vocab_size, num_layers, ...) whose num_layers is
    honoured. Layer 0 carries the reference's parameter names, layer l > 0 the same names with the layer number in
    front of the gate (`U1_i`, `S1_fi`, `S1_happy_i`, `V1_i`, `W1_i`, ...)."""

    def __init__(self, embed_size, hidden_size, factored_size, vocab_size, num_layers, feature_size=2048, bias=True,
                 dropout=0.22, max_seq_length=40):
        super().__init__()
        if not bias:
            raise CapnetError("bias=False is not supported")
        if num_layers < 1:
            raise CapnetError("num_layers must be >= 1")
        self.embed_size, self.hidden_size, self.factored_size = embed_size, hidden_size, factored_size
        self.vocab_size, self.num_layers = vocab_size, num_layers
        self.feature_size, self.max_seq_length = feature_size, max_seq_length
        self.dropout_p = dropout
        self.fast_runs = True        # Python engines only: runs of teacher-forced steps through the persistent kernel (False: step by step)
        self.engine = "c"            # "c": the whole recurrence as one C call each way (StackedSeqFn); "python": the op-by-op engines below
        self.B = _Embedding(vocab_size, embed_size)
        for l in range(num_layers):
            tag = "" if l == 0 else str(l)
            for g in "ifoc":
                setattr(self, "U%s_%s" % (tag, g), _Linear(factored_size, hidden_size))
                setattr(self, "S%s_f%s" % (tag, g), _Linear(factored_size, factored_size))
                setattr(self, "V%s_%s" % (tag, g), _Linear(embed_size if l == 0 else hidden_size, factored_size))
                setattr(self, "W%s_%s" % (tag, g), _Linear(hidden_size, hidden_size))
        for l in range(num_layers):
            tag = "" if l == 0 else str(l)
            for emo in ("happy", "sad", "angry"):
                for g in "ifoc":
                    setattr(self, "S%s_%s_%s" % (tag, emo, g), _Linear(factored_size, factored_size))
        self.C = _Linear(hidden_size, vocab_size)
        self.reset_parameters()

    def reset_parameters(self):
        """stylenet/model.py:99-113: xavier on matrices, zeros on vectors, then B and C.weight U(-0.1, 0.1)."""
        for p in self.parameters():
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)
            else:
                nn.init.zeros_(p)
        self.B.weight.data.uniform_(-0.1, 0.1)
        self.C.weight.data.uniform_(-0.1, 0.1)

    # ---- one layer's pieces -------------------------------------------------------------------
    def _mods(self, l, mode):
        tag = "" if l == 0 else str(l)
        s = _S_PREFIX[mode]
        return ([getattr(self, "V%s_%s" % (tag, g)) for g in "ifoc"],
                [getattr(self, "S%s_%s%s" % (tag, s, g)) for g in "ifoc"],
                [getattr(self, "U%s_%s" % (tag, g)) for g in "ifoc"],
                [getattr(self, "W%s_%s" % (tag, g)) for g in "ifoc"])

    def _chain(self, l, mode, x, cat=None):
        """U_g(S_g(V_g(x))) for the four gates over all rows of x -> [rows, 4H] (gate order i,f,o,c). `cat`: the layer's
        gate-stacked weights (_chain_cat, built once per forward): three products instead of twelve."""
        if cat is not None:
            Vw, Vb, Sw, Sb, Uw, Ub = cat
            return GateLinearFn.apply(GateLinearFn.apply(ops.linear(x, Vw, Vb), Sw, Sb), Uw, Ub)
        V, S, U, _ = self._mods(l, mode)
        return torch.cat([ops.linear(ops.linear(ops.linear(x, V[k].weight, V[k].bias), S[k].weight, S[k].bias),
                                     U[k].weight, U[k].bias) for k in range(4)], 1)

    def _chain_cat(self, l, mode):
        V, S, U, _ = self._mods(l, mode)
        return (torch.cat([m.weight for m in V], 0), torch.cat([m.bias for m in V], 0),
                torch.stack([m.weight for m in S], 0), torch.cat([m.bias for m in S], 0),
                torch.stack([m.weight for m in U], 0), torch.cat([m.bias for m in U], 0))

    def _wcat(self, l, mode):
        _, _, _, W = self._mods(l, mode)
        return torch.cat([w.weight for w in W], 0), torch.cat([w.bias for w in W], 0)

    # ---- forward ------------------------------------------------------------------------------
    def forward(self, captions, lengths, features=None, teacher_forcing_ratio=0.8, mode="factual", tf_mask=None):
        """-> packed logits [sum(lengths), V] in pack_padded_sequence order (stylenet/model.py:157-196 with stacked
        cells). tf_mask: the per-step teacher-forcing decisions (else one random.random() draw per step)."""
        if mode not in MODES:
            sys.stderr.write("mode name wrong!")
            raise ValueError("unknown mode %r" % (mode,))
        if not captions.is_cuda:
            raise CapnetError("StackedFactoredLSTM runs on the GPU only")
        L, H = self.num_layers, self.hidden_size
        bs = ops.batch_sizes_from_lengths(lengths)
        steps = len(bs)
        if tf_mask is None:
            tf_mask = [random.random() < teacher_forcing_ratio for _ in range(steps)]
        if len(tf_mask) < steps:
            raise CapnetError("tf_mask has %d entries, %d steps needed" % (len(tf_mask), steps))
        tf_mask = [bool(v) for v in tf_mask[:steps]]         # (longer masks are cut, as the other decoders do)
        if self.engine == "c":
            from .model import _dropout_seed
            cfg = {"batch_sizes": bs, "tf_mask": tf_mask, "hidden_size": H, "factored_size": self.factored_size,
                   "num_layers": L, "dropout": self.dropout_p if self.training else 0.0,
                   "seed": _dropout_seed(self.training, self.dropout_p), "training": self.training}
            weights = []
            for l in range(L):
                V, S, U, W = self._mods(l, mode)
                weights += ([m.weight for m in V] + [m.bias for m in V] + [m.weight for m in S] + [m.bias for m in S] +
                            [m.weight for m in U] + [m.bias for m in U] + [m.weight for m in W] + [m.bias for m in W])
            hiddens = StackedSeqFn.apply(cfg, captions, features, self.B.weight, self.C.weight, self.C.bias, *weights)
            return ops.linear(hiddens, self.C.weight, self.C.bias)
        Bn = captions.size(0)
        emb = Fn.embedding(captions, self.B.weight)                       # B(captions)
        if self.training and self.dropout_p > 0:
            emb = Fn.dropout(emb, self.dropout_p, True)
        if features is not None:
            emb = torch.cat((features.unsqueeze(1), emb), 1)
        wcat = [self._wcat(l, mode) for l in range(L)]
        ccat = [self._chain_cat(l, mode) if self.fast_runs else None for l in range(L)]
        h = [torch.zeros(Bn, H, device=captions.device) for _ in range(L)]
        c = [torch.zeros(Bn, H, device=captions.device) for _ in range(L)]
        top = []                                                          # top-layer hiddens, step by step
        predicted = None
        t = 0
        while t < steps:
            # a run: consecutive teacher-forced steps, or ONE free-running step (its input needs the previous step's top)
            t1 = t + 1
            if tf_mask[t]:
                while t1 < steps and tf_mask[t1]:
                    t1 += 1
                x = torch.cat([emb[:bs[u], u, :] for u in range(t, t1)], 0)
            else:
                if predicted is None:                                     # step 0 free-running: B(<start>) (model.py:179)
                    predicted = captions[:, 0]
                x = Fn.embedding(predicted[:bs[t]], self.B.weight)        # no dropout on the feedback (model.py:184)
            for l in range(L):
                pre = self._chain(l, mode, x, ccat[l])
                Wc, bc = wcat[l]
                if self.fast_runs and lstm_run_supported(bs[t], H, t1 - t):
                    # the run's recurrence in ONE launch of the persistent kernel (and one backward function)
                    x, c[l] = lstm_run(pre + bc, Wc, h[l][:bs[t]], c[l][:bs[t]], bs[t:t1], 0)
                    h[l] = x[x.shape[0] - bs[t1 - 1]:]
                else:
                    outs, off = [], 0
                    for u in range(t, t1):
                        b = bs[u]
                        g = pre[off:off + b] + ops.linear(h[l][:b], Wc, bc)
                        h[l], c[l] = ops.lstm_cell(g, c[l][:b], 0)
                        outs.append(h[l])
                        off += b
                    x = torch.cat(outs, 0) if len(outs) > 1 else outs[0]
                if l + 1 < L and self.training and self.dropout_p > 0:
                    x = Fn.dropout(x, self.dropout_p, True)
            top.append(x)
            if t1 < steps and not tf_mask[t1]:
                with torch.no_grad():
                    predicted = ops.argmax_rows(ops.linear(h[L - 1].detach(), self.C.weight.detach(), self.C.bias.detach())).long()
            t = t1
        hiddens = torch.cat(top, 0)
        return ops.linear(hiddens, self.C.weight, self.C.bias)
