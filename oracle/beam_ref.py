"""CPU restatement of the reference's beam search `sample()`. TEST INFRASTRUCTURE.

  sample_factored      <- DecoderFactoredLSTM.sample,     stylenet/model.py:198-294
  sample_lstm          <- DecoderRNN.sample,              nic/model.py:117-207
  sample_factored_att  <- DecoderFactoredLSTMAtt.sample,  stylenet/model_att.py:307-426
  sample_lstm_att      <- DecoderRNNAtt.sample,           nic/model_att.py:204-297
The loop follows the reference statement by statement (tensor bookkeeping included) with ONE
documented deviation: `top_k_words / vocab_size` (model.py:249) is written `//`. Under the
torch 1.1 the reference pins, `/` on a LongTensor was an integer division; under current torch the
reference line produces a float tensor and the next indexing statement raises. Pinned by
tests/golden/sample_tiny.npz (the reference's own methods run with the legacy integer division,
tools/gen_golden.py sample_tiny).
"""
import torch
import torch.nn.functional as Fn

from . import decoders_ref as D


def _beam(step_fn, state, vocab_size, start_token, end_token, k, max_seq_length):
    """step_fn(prev_words [s, 1], state) -> (hidden-projected logits [s, V], state')."""
    k_prev_words = torch.LongTensor([[start_token]] * k)
    seqs = k_prev_words
    top_k_scores = torch.zeros(k, 1)
    complete_seqs, complete_seqs_scores = [], []
    step = 1
    while True:
        output, state = step_fn(k_prev_words, state)
        scores = Fn.log_softmax(output, dim=1)
        scores = top_k_scores.expand_as(scores) + scores
        if step == 1:
            top_k_scores, top_k_words = scores[0].topk(k, 0, True, True)
        else:
            top_k_scores, top_k_words = scores.view(-1).topk(k, 0, True, True)
        prev_word_inds = top_k_words // vocab_size
        next_word_inds = top_k_words % vocab_size
        seqs = torch.cat([seqs[prev_word_inds], next_word_inds.unsqueeze(1)], dim=1)
        incomplete_inds = [ind for ind, w in enumerate(next_word_inds) if w != end_token]
        complete_inds = list(set(range(len(next_word_inds))) - set(incomplete_inds))
        if len(complete_inds) > 0:
            complete_seqs.extend(seqs[complete_inds].tolist())
            complete_seqs_scores.extend(top_k_scores[complete_inds])
        k -= len(complete_inds)
        if k == 0:
            break
        seqs = seqs[incomplete_inds]
        state = tuple(s[prev_word_inds[incomplete_inds]] for s in state)
        top_k_scores = top_k_scores[incomplete_inds].unsqueeze(1)
        k_prev_words = next_word_inds[incomplete_inds].unsqueeze(1)
        if step > max_seq_length:
            break
        step += 1
    if len(complete_seqs_scores) == 0:
        return torch.Tensor([[end_token]]).long()
    i = complete_seqs_scores.index(max(complete_seqs_scores))
    return torch.Tensor([complete_seqs[i]]).long()


def sample_factored(p, hidden_size, start_token, end_token, k=5, mode="factual", max_seq_length=40):
    V = p["C.weight"].shape[0]

    def step_fn(prev_words, state):
        h, c = D.factored_step(p, p["B.weight"][prev_words].squeeze(1), state[0], state[1], mode)
        return Fn.linear(h, p["C.weight"], p["C.bias"]), (h, c)

    z = torch.zeros(k, hidden_size)
    return _beam(step_fn, (z, z.clone()), V, start_token, end_token, k, max_seq_length)


def sample_lstm(p, hidden_size, start_token, end_token, k=5, max_seq_length=40):
    V = p["linear.weight"].shape[0]

    def step_fn(prev_words, state):
        h, c = D.lstmcell_step(p, p["embed.weight"][prev_words].squeeze(1), state[0], state[1])
        return Fn.linear(h, p["linear.weight"], p["linear.bias"]), (h, c)

    z = torch.zeros(k, hidden_size)
    return _beam(step_fn, (z, z.clone()), V, start_token, end_token, k, max_seq_length)


def sample_factored_att(p, features, start_token, end_token, k=5, mode="factual", max_seq_length=40):
    V = p["C.weight"].shape[0]
    feat = features.reshape(1, -1, features.size(-1))
    feat = feat.expand(k, feat.size(1), feat.size(2))
    mean = feat.mean(dim=1)
    h0, c0 = D._lin(p, "init_h", mean), D._lin(p, "init_c", mean)
    att = D.MODE_ATT[mode]

    def step_fn(prev_words, state):
        h, c, f = state
        awe, _ = D.attention_step(p, att, f, h)
        awe = torch.sigmoid(D._lin(p, "f_beta", h)) * awe
        x = torch.cat([p["B.weight"][prev_words].squeeze(1), awe], dim=1)
        h, c = D.factored_step(p, x, h, c, mode)
        return Fn.linear(h, p["C.weight"], p["C.bias"]), (h, c, f)

    return _beam(step_fn, (h0, c0, feat), V, start_token, end_token, k, max_seq_length)


def sample_lstm_att(p, features, start_token, end_token, k=5, max_seq_length=40):
    V = p["linear.weight"].shape[0]
    feat = features.reshape(1, -1, features.size(-1))
    feat = feat.expand(k, feat.size(1), feat.size(2))
    mean = feat.mean(dim=1)
    h0, c0 = D._lin(p, "init_h", mean), D._lin(p, "init_c", mean)

    def step_fn(prev_words, state):
        h, c, f = state
        awe, _ = D.attention_step(p, "attention", f, h)
        awe = torch.sigmoid(D._lin(p, "f_beta", h)) * awe
        x = torch.cat([p["embed.weight"][prev_words].squeeze(1), awe], dim=1)
        h, c = D.lstmcell_step(p, x, h, c)
        return Fn.linear(h, p["linear.weight"], p["linear.bias"]), (h, c, f)

    return _beam(step_fn, (h0, c0, feat), V, start_token, end_token, k, max_seq_length)
