"""NIC models on the MI355X kernels: EncoderCNN (shared) and DecoderRNN.

Mirrors nic/model.py of the reference: DecoderRNN = Embedding + nn.LSTMCell(embed, hidden) +
Linear(hidden, vocab), same scheduled-sampling loop as the StyleNet decoder. The recurrence
runs in libcapnet_hip.so (cell 1 of capnet_seq_forward/backward).
"""
import torch
import torch.nn as nn

from . import ops
from .model import (Dropout, Embedding, EncoderCNN, Linear, _dropout_seed,  # noqa: F401
                    _resolve_tf_mask)


class LSTMCell(nn.Module):
    """nn.LSTMCell parameter container (weight_ih [4H,E], weight_hh [4H,H], two biases;
    gate order i, f, g, o)."""

    def __init__(self, input_size, hidden_size, bias=True):
        super().__init__()
        self.input_size, self.hidden_size = input_size, hidden_size
        k = 1.0 / hidden_size ** 0.5
        self.weight_ih = nn.Parameter(torch.empty(4 * hidden_size, input_size).uniform_(-k, k))
        self.weight_hh = nn.Parameter(torch.empty(4 * hidden_size, hidden_size).uniform_(-k, k))
        self.bias_ih = nn.Parameter(torch.empty(4 * hidden_size).uniform_(-k, k))
        self.bias_hh = nn.Parameter(torch.empty(4 * hidden_size).uniform_(-k, k))

    def forward(self, x, states):
        h, c = states
        pre = ops.linear(x, self.weight_ih, self.bias_ih) + ops.linear(h, self.weight_hh, self.bias_hh)
        return ops.lstm_pointwise(pre, c, ops.CELL_LSTM)


class DecoderRNN(nn.Module):
    """nic/model.py:29-207. `num_layers` is accepted and ignored, as in the reference."""

    def __init__(self,
                 embed_size,
                 hidden_size,
                 vocab_size,
                 num_layers,
                 feature_size=2048,
                 dropout=0.22,
                 max_seq_length=40):
        super(DecoderRNN, self).__init__()
        self.feature_size = feature_size
        self.hidden_size = hidden_size
        self.embed_size = embed_size
        self.vocab_size = vocab_size
        self.max_seq_length = max_seq_length
        self.dropout = Dropout(dropout)
        self.embed = Embedding(vocab_size, embed_size)
        self.lstm = LSTMCell(embed_size, hidden_size, bias=True)
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

    def forward_step(self, embedded, states):
        h_t, c_t = self.lstm(embedded, states)
        return h_t, (h_t, c_t)

    def forward(self, captions, lengths, features, teacher_forcing_ratio=0.8, tf_mask=None):
        batch_sizes = ops.batch_sizes_from_lengths(lengths)
        cfg = {
            "cell": ops.CELL_LSTM,
            "batch_sizes": batch_sizes,
            "tf_mask": _resolve_tf_mask(tf_mask, len(batch_sizes), teacher_forcing_ratio),
            "hidden_size": self.hidden_size,
            "dropout": self.dropout.p if self.training else 0.0,
            "seed": _dropout_seed(self.training, self.dropout.p),
            "training": self.training,
        }
        weights = [self.lstm.weight_ih, self.lstm.bias_ih, self.lstm.weight_hh, self.lstm.bias_hh]
        hiddens = ops.decoder_sequence(cfg, captions, features, self.embed.weight,
                                       self.linear.weight, self.linear.bias, weights)
        return self.linear(hiddens)

    def sample(self, features, start_token, end_token, k=5):
        """Beam search, nic/model.py:117-207 (the image features are not an input of the decode
        steps there either). Returns LongTensor [1, L]."""
        from .beam import beam_search
        dev = self.embed.weight.device

        def step_fn(prev_words, state):
            hidden, (h, c) = self.forward_step(self.embed(prev_words), state)
            return self.linear(hidden), (h, c)

        with torch.no_grad():
            zeros = torch.zeros(k, self.hidden_size, dtype=torch.float32, device=dev)
            return beam_search(step_fn, (zeros, zeros.clone()), self.vocab_size, start_token,
                               end_token, k, self.max_seq_length, dev)

    def sample_batch(self, features, start_token, end_token, k=5):
        """sample() for every row of `features` at once (capnet.beam.beam_search_batched)."""
        from .beam import beam_search_batched
        dev = self.embed.weight.device
        n = features.size(0)

        def step_fn(prev_words, state):
            hidden, (h, c) = self.forward_step(self.embed(prev_words), state)
            return self.linear(hidden), (h, c)

        with torch.no_grad():
            zeros = torch.zeros(n * k, self.hidden_size, dtype=torch.float32, device=dev)
            return beam_search_batched(step_fn, (zeros, zeros.clone()), n, self.vocab_size, start_token, end_token, k,
                                       self.max_seq_length, dev)
