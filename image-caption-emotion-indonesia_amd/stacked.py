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

Engine: runs of teacher-forced steps are processed layer by layer -- a layer's input chain U(S(V x)) over all rows of
the run as batched MFMA GEMMs (ops.linear), the recurrence step by step (W GEMM + the fused cell kernel,
ops.lstm_cell) -- with torch autograd composing the backward from those kernels' own backward functions. It is the
simple engine, not the fast one: the single-layer path's fused sequence driver and persistent kernel do not apply
to it yet."""
import random
import sys

import torch
import torch.nn as nn
import torch.nn.functional as Fn

from . import ops
from ._lib import CapnetError
from .model import Embedding as _Embedding, Linear as _Linear

MODES = ("factual", "happy", "sad", "angry")
_S_PREFIX = {"factual": "f", "happy": "happy_", "sad": "sad_", "angry": "angry_"}


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

    def _chain(self, l, mode, x):
        """U_g(S_g(V_g(x))) for the four gates over all rows of x -> [rows, 4H] (gate order i,f,o,c)."""
        V, S, U, _ = self._mods(l, mode)
        return torch.cat([ops.linear(ops.linear(ops.linear(x, V[k].weight, V[k].bias), S[k].weight, S[k].bias),
                                     U[k].weight, U[k].bias) for k in range(4)], 1)

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
        if len(tf_mask) != steps:
            raise CapnetError("tf_mask must have one entry per time step")
        Bn = captions.size(0)
        emb = Fn.embedding(captions, self.B.weight)                       # B(captions)
        if self.training and self.dropout_p > 0:
            emb = Fn.dropout(emb, self.dropout_p, True)
        if features is not None:
            emb = torch.cat((features.unsqueeze(1), emb), 1)
        wcat = [self._wcat(l, mode) for l in range(L)]
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
                pre = self._chain(l, mode, x)
                Wc, bc = wcat[l]
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
