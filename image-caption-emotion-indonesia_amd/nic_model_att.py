"""NIC attention path on the MI355X kernels: EncoderCNN, Attention, DecoderRNNAtt
(nic/model_att.py of the reference).

DecoderRNNAtt is the attention loop of stylenet/model_att.py around an nn.LSTMCell(E + C, H)
instead of the factored cell: no modes, gate order i,f,g,o, h = o*tanh(c). The sequence runs in
libcapnet_hip.so (capnet_att_seq_forward/backward with cell = 1).
"""
import torch
import torch.nn as nn

from . import ops
from .model import Dropout, Embedding, Linear, _Marker, _dropout_seed, _resolve_tf_mask
from .model_att import Attention, EncoderCNN  # noqa: F401  (same classes as the StyleNet path)
from .nic_model import LSTMCell


class DecoderRNNAtt(nn.Module):
    """nic/model_att.py:72-297. `num_layers` is accepted and ignored, as in the reference."""

    def __init__(self,
                 attention_size,
                 embed_size,
                 hidden_size,
                 vocab_size,
                 num_layers,
                 feature_size=2048,
                 dropout=0.22,
                 max_seq_length=40):
        super(DecoderRNNAtt, self).__init__()
        self.attention_size = attention_size
        self.feature_size = feature_size
        self.hidden_size = hidden_size
        self.embed_size = embed_size
        self.vocab_size = vocab_size
        self.max_seq_length = max_seq_length
        # registration order follows nic/model_att.py:89-113 (state_dict order)
        self.init_h = Linear(feature_size, hidden_size)
        self.init_c = Linear(feature_size, hidden_size)
        self.dropout = Dropout(dropout)
        self.attention = Attention(feature_size, hidden_size, attention_size)
        self.embed = Embedding(vocab_size, embed_size)
        self.f_beta = Linear(hidden_size, feature_size)
        self.sigmoid = _Marker()
        self.lstm = LSTMCell(embed_size + feature_size, hidden_size, bias=True)
        self.linear = Linear(hidden_size, vocab_size)
        self.reset_parameters()
        self.init_weights()

    def reset_parameters(self):
        for p in self.parameters():
            if p.data.ndimension() >= 2:
                nn.init.xavier_uniform_(p.data)
            else:
                nn.init.zeros_(p.data)

    def init_weights(self):
        self.embed.weight.data.uniform_(-0.1, 0.1)
        self.linear.bias.data.fill_(0)
        self.linear.weight.data.uniform_(-0.1, 0.1)

    def init_hidden_state(self, feature):
        mean_feature = feature.mean(dim=1)
        return self.init_h(mean_feature), self.init_c(mean_feature)

    def forward_step(self, embedded, states):
        h_t, c_t = self.lstm(embedded, states)
        return h_t, (h_t, c_t)

    def _weights(self):
        att = self.attention
        out = [self.lstm.weight_ih, self.lstm.bias_ih, self.lstm.weight_hh, self.lstm.bias_hh]
        for m in (self.init_h, self.init_c, att.encoder_att, att.decoder_att, att.full_att, self.f_beta):
            out += [m.weight, m.bias]
        return out

    def forward(self, captions, lengths, features, teacher_forcing_ratio=0.8, tf_mask=None):
        """Returns (outputs [N, V], alphas [B, max(lengths), P]) -- nic/model_att.py:152-202."""
        batch_size = captions.size(0)
        features = features.reshape(batch_size, -1, features.size(-1))
        batch_sizes = ops.batch_sizes_from_lengths(lengths)
        cfg = {
            "cell": ops.CELL_LSTM,
            "batch_sizes": batch_sizes,
            "tf_mask": _resolve_tf_mask(tf_mask, len(batch_sizes), teacher_forcing_ratio),
            "hidden_size": self.hidden_size,
            "attention_size": self.attention_size,
            "dropout": self.dropout.p if self.training else 0.0,
            "seed": _dropout_seed(self.training, self.dropout.p),
            "training": self.training,
        }
        hiddens, alphas = ops.decoder_att_sequence(cfg, captions, features.detach(), self.embed.weight,
                                                   self.linear.weight, self.linear.bias,
                                                   self._weights())
        return self.linear(hiddens), alphas

    def sample(self, features, start_token, end_token, k=5):
        """Beam search with attention, nic/model_att.py:204-297. Returns LongTensor [1, L]."""
        from .beam import beam_search
        dev = self.embed.weight.device
        attention = self.attention
        E, A, Cdim = self.embed_size, self.attention_size, features.size(-1)
        with torch.no_grad():
            feat1 = features.reshape(1, -1, Cdim).to(dev).contiguous()
            P = feat1.size(1)
            feat_k = feat1.expand(k, P, Cdim).contiguous()
            att1_k = attention.encoder_att(feat1[0]).reshape(1, P, A).expand(k, P, A).contiguous()
            h0, c0 = self.init_hidden_state(feat_k)
            wz = torch.cat([attention.decoder_att.weight, self.f_beta.weight], 0).contiguous()
            bz = torch.cat([attention.decoder_att.bias, self.f_beta.bias], 0).contiguous()

            def step_fn(prev_words, state):
                h, c = state
                s_rows = h.shape[0]
                z = ops.linear(h, wz, bz).contiguous()
                xa = torch.empty((s_rows, E + Cdim), dtype=torch.float32, device=dev)
                xa[:, :E] = self.embed(prev_words)
                ops.attention_step(att1_k[:s_rows], feat_k[:s_rows], z, A, attention.full_att.weight,
                                   attention.full_att.bias, xa=xa, xa_col=E)
                hidden, (h, c) = self.forward_step(xa, (h, c))
                return self.linear(hidden), (h, c)

            return beam_search(step_fn, (h0, c0), self.vocab_size, start_token, end_token, k,
                               self.max_seq_length, dev)
