"""Beam search shared by the decoders' `sample()` methods.

Mirrors the loop of stylenet/model.py:198-294 (identical in nic/model.py and, with the attention
inputs, stylenet/model_att.py:307-426): k live beams, every step expands them by
log-softmax + running score, keeps the k best of the flattened [s*V] scores, retires beams that
produced <end>, and returns the completed sequence with the highest score.

Device work per step: the decoder's own step (C-ABI kernels) and ONE `capnet_beam_topk` launch
(log-softmax + add + top-k fused); the k scores / indices come back in one transfer and the
bookkeeping (sequence lists, which beam survives) is host Python, as in the reference.

Deviation from the reference text: `top_k_words / vocab_size` (model.py:249) is an integer
division here. Under the torch 1.1 the reference pins it was one; under current torch the
reference line raises.
"""
import torch

from . import ops


def beam_search(step_fn, state, vocab_size, start_token, end_token, k, max_seq_length, device):
    """step_fn(prev_words LongTensor[s], state) -> (logits [s, V], state'); `state` is a tuple of
    tensors whose leading dimension is the beam and which are re-indexed here when beams are
    re-ordered or retired. Returns LongTensor [1, L] (starts with start_token)."""
    k_prev_words = torch.full((k,), start_token, dtype=torch.long, device=device)
    seqs = [[int(start_token)] for _ in range(k)]
    top_k_scores = torch.zeros(k, dtype=torch.float32, device=device)
    complete_seqs, complete_seqs_scores = [], []
    step = 1
    while True:
        logits, state = step_fn(k_prev_words, state)
        # first step: all k beams are identical, only row 0 competes (model.py:240-242)
        rows = 1 if step == 1 else logits.shape[0]
        scores_d, flat_d = ops.beam_topk(logits, top_k_scores, rows, k)
        scores = scores_d.tolist()
        flat = flat_d.tolist()
        prev_word_inds = [f // vocab_size for f in flat]
        next_word_inds = [f % vocab_size for f in flat]
        seqs = [seqs[p] + [w] for p, w in zip(prev_word_inds, next_word_inds)]
        incomplete_inds = [i for i, w in enumerate(next_word_inds) if w != end_token]
        complete_inds = [i for i, w in enumerate(next_word_inds) if w == end_token]
        for i in complete_inds:
            complete_seqs.append(seqs[i])
            complete_seqs_scores.append(scores[i])
        k -= len(complete_inds)
        if k == 0:
            break
        seqs = [seqs[i] for i in incomplete_inds]
        keep = torch.tensor([prev_word_inds[i] for i in incomplete_inds], dtype=torch.long,
                            device=device)
        state = tuple(s.index_select(0, keep) for s in state)
        top_k_scores = scores_d[torch.tensor(incomplete_inds, dtype=torch.long, device=device)]
        k_prev_words = torch.tensor([next_word_inds[i] for i in incomplete_inds], dtype=torch.long,
                                    device=device)
        if step > max_seq_length:
            break
        step += 1
    if not complete_seqs_scores:   # "prevent empty sequence", model.py:287-289
        return torch.tensor([[int(end_token)]], dtype=torch.long, device=device)
    best = complete_seqs_scores.index(max(complete_seqs_scores))
    return torch.tensor([complete_seqs[best]], dtype=torch.long, device=device)


def beam_search_batched(step_fn, state, n, vocab_size, start_token, end_token, k, max_seq_length, device):
    """beam_search for n images at once: the evaluator of the reference (stylenet/evaluator.py:63-120) calls sample() per
    test image, k rows per decode step; here the beams of ALL images advance together -- one decoder step over sum(live
    beams) rows and one capnet_beam_topk_batched launch per step -- with per image exactly beam_search's bookkeeping
    (first step: row 0 only; <end> retires a beam; the best completed sequence wins; empty -> [<end>]).
    `state`: tuple of tensors with n * k leading rows (image i's k beams at rows i k .. i k + k - 1).
    Returns a list of n token lists (each starts with start_token)."""
    if k > 16:
        raise ops.CapnetError("beam_search_batched: k <= 16")
    live = [k] * n                                           # beams still open per image
    seqs = [[[int(start_token)] for _ in range(k)] for _ in range(n)]
    done = [[] for _ in range(n)]                            # (score, sequence) of completed beams
    prev_words = torch.full((n * k,), start_token, dtype=torch.long, device=device)
    top_scores = torch.zeros(n * k, dtype=torch.float32, device=device)
    step = 1
    while True:
        logits, state = step_fn(prev_words, state)
        row0, meta = 0, []
        for i in range(n):
            meta.append((row0, (1 if step == 1 else live[i]) if live[i] else 0, live[i]))
            row0 += live[i]
        scores_d, flat_d = ops.beam_topk_batched(logits, top_scores, torch.tensor(meta, dtype=torch.int32, device=device))
        scores, flat = scores_d.tolist(), flat_d.tolist()
        keep_rows, next_words, next_scores = [], [], []
        for i in range(n):
            if not live[i]:
                continue
            r0 = meta[i][0]
            new_seqs = []
            for j in range(live[i]):
                p, w = flat[i][j] // vocab_size, flat[i][j] % vocab_size
                sq = seqs[i][p] + [w]
                if w == end_token:
                    done[i].append((scores[i][j], sq))
                else:
                    new_seqs.append(sq)
                    keep_rows.append(r0 + p)
                    next_words.append(w)
                    next_scores.append(scores[i][j])
            seqs[i] = new_seqs
            live[i] = len(new_seqs)
        if not keep_rows or step > max_seq_length:
            break
        keep = torch.tensor(keep_rows, dtype=torch.long, device=device)
        state = tuple(s.index_select(0, keep) for s in state)
        prev_words = torch.tensor(next_words, dtype=torch.long, device=device)
        top_scores = torch.tensor(next_scores, dtype=torch.float32, device=device)
        step += 1
    out = []
    for i in range(n):
        if not done[i]:
            out.append([int(end_token)])
        else:
            best = max(range(len(done[i])), key=lambda q: (done[i][q][0], -q))       # first maximum, as list.index(max(...))
            out.append(done[i][best][1])
    return out
