"""Attention path on the MI355X kernels: EncoderCNN, Attention, DecoderFactoredLSTMAtt
(stylenet/model_att.py).

EncoderCNN(encoded_image_size=14): ResNet-152 children[:-2] under no_grad, AdaptiveAvgPool2d
to 14x14 (an exact 2x replication of the 7x7 map), permuted to NHWC. The trunk already produces
NHWC, so the permute costs nothing here.
"""
import sys

import torch
import torch.nn as nn

from . import _lib, ops
from ._lib import CapnetError, check, current_stream, ptr
from .model import (Dropout, Embedding, Linear, _Marker, _MODES, _TrunkRunner, _dropout_seed,
                    _resnet152_children, _resolve_tf_mask)


class EncoderCNN(nn.Module):

    def __init__(self, encoded_image_size=14):
        super(EncoderCNN, self).__init__()
        self.resnet = _resnet152_children(with_avgpool=False)
        self.adaptive_pool = _Marker()   # parameter-less, as nn.AdaptiveAvgPool2d
        self.encoded_image_size = encoded_image_size
        self._runner = [None]

    def _trunk(self):
        if self._runner[0] is None:
            self._runner[0] = _TrunkRunner(self.resnet)
        return self._runner[0]

    def forward(self, images, slot=0, defer_stats=False, balance_tails=True, graph=False):
        """slot / defer_stats: see _TrunkRunner.forward (capnet.train.TrunkPipeline); with
        defer_stats the result is (features, apply_running_stats or None)."""
        with torch.no_grad():
            res = self._trunk().forward(images, self.training, False, True, slot=slot,
                                        defer_stats=defer_stats and self.training,
                                        balance_tails=balance_tails, graph=graph)
            fmap = res[1]
            apply_fn = res[2] if len(res) > 2 else None
            out = self._pool(fmap)
        return (out, apply_fn) if defer_stats else out

    def zero_grad(self, set_to_none=True):
        """No trainable parameter and no gradient anywhere (the trunk runs under no_grad)."""
        return None

    def _pool(self, fmap):
        """AdaptiveAvgPool2d(encoded_image_size) of the 7x7 map: an exact replication."""
        b, side = fmap.shape[0], fmap.shape[1]
        out_side = self.encoded_image_size
        if out_side == side:
            return fmap
        if out_side % side != 0:
            raise CapnetError("encoded_image_size %d must be a multiple of the trunk's %d"
                              % (out_side, side))
        out = torch.empty((b, out_side, out_side, 2048), dtype=torch.float32, device=fmap.device)
        check(_lib.lib().capnet_adaptive_pool_replicate(ptr(fmap), ptr(out), b, side, out_side,
                                                        2048, current_stream()),
              "capnet_adaptive_pool_replicate")
        return out


class Attention(nn.Module):
    """stylenet/model_att.py:32-70 (parameter container + single-step forward)."""

    def __init__(self, encoder_dim, decoder_dim, attention_dim):
        super(Attention, self).__init__()
        self.encoder_att = Linear(encoder_dim, attention_dim)
        self.decoder_att = Linear(decoder_dim, attention_dim)
        self.full_att = Linear(attention_dim, 1)
        self.relu = _Marker()
        self.softmax = _Marker()

    def forward(self, encoder_out, decoder_hidden):
        """(attention-weighted encoding [s, C], alpha [s, P]) -- stylenet/model_att.py:51-70.
        Inference-only entry (no autograd); training goes through the fused sequence kernels."""
        with torch.no_grad():
            s_rows, P, Cdim = encoder_out.shape
            A = self.encoder_att.weight.shape[0]
            att1 = self.encoder_att(encoder_out.reshape(s_rows * P, Cdim)).reshape(s_rows, P, A)
            z = torch.zeros((s_rows, A + Cdim), dtype=torch.float32, device=encoder_out.device)
            z[:, :A] = self.decoder_att(decoder_hidden)
            return ops.attention_step(att1.contiguous(), encoder_out.contiguous(), z, A,
                                      self.full_att.weight, self.full_att.bias)


class DecoderFactoredLSTMAtt(nn.Module):
    """stylenet/model_att.py:73-426. `num_layers` is accepted and ignored, as in the reference.
    The image features carry no gradient (the attention encoder has no trainable parameter)."""

    def __init__(self,
                 attention_size,
                 embed_size,
                 hidden_size,
                 factored_size,
                 vocab_size,
                 num_layers,
                 feature_size=2048,
                 bias=True,
                 dropout=0.22,
                 max_seq_length=40):
        super(DecoderFactoredLSTMAtt, self).__init__()
        if not bias:
            raise CapnetError("DecoderFactoredLSTMAtt: bias=False is not supported by the HIP path")
        self.attention_size = attention_size
        self.feature_size = feature_size
        self.hidden_size = hidden_size
        self.factored_size = factored_size
        self.embed_size = embed_size
        self.vocab_size = vocab_size
        self.max_seq_length = max_seq_length
        # registration order follows stylenet/model_att.py:92-164 (state_dict order)
        self.init_h = Linear(feature_size, hidden_size)
        self.init_c = Linear(feature_size, hidden_size)
        self.dropout = Dropout(dropout)
        self.attention = Attention(feature_size, hidden_size, attention_size)
        self.B = Embedding(vocab_size, embed_size)
        self.f_beta = Linear(hidden_size, feature_size)
        self.sigmoid = _Marker()
        for g in "ifoc":
            setattr(self, "U_" + g, Linear(factored_size, hidden_size, bias=bias))
            setattr(self, "S_f" + g, Linear(factored_size, factored_size, bias=bias))
            setattr(self, "V_" + g, Linear(embed_size + feature_size, factored_size, bias=bias))
            setattr(self, "W_" + g, Linear(hidden_size, hidden_size, bias=bias))
        for emo in ("happy", "sad", "angry"):
            setattr(self, "attention_" + emo, Attention(feature_size, hidden_size, attention_size))
            for g in "ifoc":
                setattr(self, "S_%s_%s" % (emo, g), Linear(factored_size, factored_size, bias=bias))
        self.C = Linear(hidden_size, vocab_size, bias=bias)
        self.reset_parameters()
        self.init_weights()

    def reset_parameters(self):
        for p in self.parameters():
            if p.data.ndimension() >= 2:
                nn.init.xavier_uniform_(p.data)
            else:
                nn.init.zeros_(p.data)

    def init_weights(self):
        self.B.weight.data.uniform_(-0.1, 0.1)
        self.C.bias.data.fill_(0)
        self.C.weight.data.uniform_(-0.1, 0.1)

    def _mode_modules(self, mode):
        if mode == "factual":
            return self.attention, [getattr(self, "S_f" + g) for g in "ifoc"]
        if mode in _MODES:
            return (getattr(self, "attention_" + mode),
                    [getattr(self, "S_%s_%s" % (mode, g)) for g in "ifoc"])
        sys.stderr.write("mode name wrong!")      # the reference then fails with UnboundLocalError
        raise ValueError("unknown mode %r (expected one of %s)" % (mode, ", ".join(_MODES)))

    def _weights(self, mode):
        att, S = self._mode_modules(mode)
        V = [getattr(self, "V_" + g) for g in "ifoc"]
        U = [getattr(self, "U_" + g) for g in "ifoc"]
        W = [getattr(self, "W_" + g) for g in "ifoc"]
        out = []
        for grp in (V, S, U, W):
            out += [m.weight for m in grp]
            out += [m.bias for m in grp]
        for m in (self.init_h, self.init_c, att.encoder_att, att.decoder_att, att.full_att, self.f_beta):
            out += [m.weight, m.bias]
        return out

    def init_hidden_state(self, features):
        """h0, c0 = init_h / init_c (mean over pixels) -- stylenet/model_att.py:185-194."""
        mean_features = features.mean(dim=1)
        return self.init_h(mean_features), self.init_c(mean_features)

    def forward_step(self, embedded, states, mode):
        """One factored-LSTM step on [embedding | gated context] -- model_att.py:196-236."""
        h_t, c_t = states
        _, S = self._mode_modules(mode)
        V = [getattr(self, "V_" + g) for g in "ifoc"]
        U = [getattr(self, "U_" + g) for g in "ifoc"]
        W = [getattr(self, "W_" + g) for g in "ifoc"]
        pre = torch.cat([U[k](S[k](V[k](embedded))) + W[k](h_t) for k in range(4)], 1)
        h_t, c_t = ops.lstm_pointwise(pre, c_t, ops.CELL_FACTORED)
        return h_t, (h_t, c_t)

    def sample(self, features, start_token, end_token, k=5, factual_limit=-1, mode='factual'):
        """Beam search with attention, stylenet/model_att.py:307-426. `features`: the encoder map
        of ONE image ([1, S, S, C] or [1, P, C]). encoder_att(features) is computed once (the
        reference recomputes it for every beam and step). Returns LongTensor [1, L]."""
        from .beam import beam_search
        dev = self.B.weight.device
        attention, _ = self._mode_modules(mode)
        E, A, Cdim = self.embed_size, self.attention_size, features.size(-1)
        with torch.no_grad():
            feat1 = features.reshape(1, -1, Cdim).to(dev).contiguous()
            P = feat1.size(1)
            feat_k = feat1.expand(k, P, Cdim).contiguous()
            att1_k = attention.encoder_att(feat1[0]).reshape(1, P, A).expand(k, P, A).contiguous()
            h0, c0 = self.init_hidden_state(feat_k)
            # [decoder_att ; f_beta] stacked: one product with h per step
            wz = torch.cat([attention.decoder_att.weight, self.f_beta.weight], 0).contiguous()
            bz = torch.cat([attention.decoder_att.bias, self.f_beta.bias], 0).contiguous()

            def step_fn(prev_words, state):
                h, c = state
                s_rows = h.shape[0]
                z = ops.linear(h, wz, bz).contiguous()
                xa = torch.empty((s_rows, E + Cdim), dtype=torch.float32, device=dev)
                xa[:, :E] = self.B(prev_words)
                # all beams look at the same image: rows of feat_k / att1_k are identical, so
                # re-indexing them (model_att.py:413) is a slice
                ops.attention_step(att1_k[:s_rows], feat_k[:s_rows], z, A, attention.full_att.weight,
                                   attention.full_att.bias, xa=xa, xa_col=E)
                hidden, (h, c) = self.forward_step(xa, (h, c), mode=mode)
                return self.C(hidden), (h, c)

            return beam_search(step_fn, (h0, c0), self.vocab_size, start_token, end_token, k,
                               self.max_seq_length, dev)

    def sample_batch(self, features, start_token, end_token, k=5, factual_limit=-1, mode='factual'):
        """sample() for every image of `features` ([n, S, S, C] or [n, P, C]) at once: the reference's evaluator
        (stylenet/evaluator.py:63-120) decodes its test images one sample() call at a time; here all live beams of all
        images take their decoder step together (capnet.beam.beam_search_batched). encoder_att(features) once per image;
        a beam's rows of the map are gathered by its image index. Returns a list of token lists."""
        from .beam import beam_search_batched
        dev = self.B.weight.device
        attention, _ = self._mode_modules(mode)
        E, A, Cdim = self.embed_size, self.attention_size, features.size(-1)
        n = features.size(0)
        with torch.no_grad():
            feat = features.reshape(n, -1, Cdim).to(dev).contiguous()
            P = feat.size(1)
            att1 = attention.encoder_att(feat.reshape(n * P, Cdim)).reshape(n, P, A).contiguous()
            h0, c0 = self.init_hidden_state(feat)
            img = torch.arange(n, device=dev).repeat_interleave(k)
            h0, c0 = h0.index_select(0, img).contiguous(), c0.index_select(0, img).contiguous()
            wz = torch.cat([attention.decoder_att.weight, self.f_beta.weight], 0).contiguous()
            bz = torch.cat([attention.decoder_att.bias, self.f_beta.bias], 0).contiguous()

            def step_fn(prev_words, state):
                h, c, im = state
                s_rows = h.shape[0]
                z = ops.linear(h, wz, bz).contiguous()
                xa = torch.empty((s_rows, E + Cdim), dtype=torch.float32, device=dev)
                xa[:, :E] = self.B(prev_words)
                ops.attention_step(att1.index_select(0, im), feat.index_select(0, im), z, A, attention.full_att.weight,
                                   attention.full_att.bias, xa=xa, xa_col=E)
                hidden, (h, c) = self.forward_step(xa, (h, c), mode=mode)
                return self.C(hidden), (h, c, im)

            return beam_search_batched(step_fn, (h0, c0, img), n, self.vocab_size, start_token, end_token, k,
                                       self.max_seq_length, dev)

    def forward(self,
                captions,
                lengths,
                features,
                teacher_forcing_ratio=0.8,
                mode='factual',
                tf_mask=None):
        """Returns (outputs [N, V], alphas [B, max(lengths), P])."""
        batch_size = captions.size(0)
        features = features.reshape(batch_size, -1, features.size(-1))
        batch_sizes = ops.batch_sizes_from_lengths(lengths)
        cfg = {
            "batch_sizes": batch_sizes,
            "tf_mask": _resolve_tf_mask(tf_mask, len(batch_sizes), teacher_forcing_ratio),
            "hidden_size": self.hidden_size,
            "factored_size": self.factored_size,
            "attention_size": self.attention_size,
            "dropout": self.dropout.p if self.training else 0.0,
            "seed": _dropout_seed(self.training, self.dropout.p),
            "training": self.training,
        }
        hiddens, alphas = ops.decoder_att_sequence(cfg, captions, features.detach(), self.B.weight,
                                                   self.C.weight, self.C.bias, self._weights(mode))
        return self.C(hiddens), alphas
