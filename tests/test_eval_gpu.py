"""Validation path: top-k accuracy kernel, packed-prediction unpacking, BLEU-4 and the
val_factual / val_emotion loops (stylenet/train_multitask.py:272-361, 411-508) against torch / the
oracle on the same inputs."""
import pytest
import torch
import torch.nn as nn

import capnet
from capnet import ops, synthetic
from capnet.metrics import corpus_bleu
from capnet.model import DecoderFactoredLSTM
from capnet.train import CrossEntropyLoss, val_emotion, val_factual
from capnet.utils import accuracy
from oracle import decoders_ref as D

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("N,V,k", [(7, 37, 5), (1037, 8192, 5), (64, 7411, 1), (3, 4, 5)])
def test_accuracy_matches_torch_topk(dev, N, V, k):
    g = torch.Generator().manual_seed(N + V)
    scores = torch.randn(N, V, generator=g)
    targets = torch.randint(0, V, (N,), generator=g)
    kk = min(k, V)
    _, ind = scores.topk(kk, 1, True, True)
    ref = ind.eq(targets.view(-1, 1).expand_as(ind)).view(-1).float().sum().item() * (100.0 / N)
    got = accuracy(scores.to(dev), targets.to(dev), kk)
    assert abs(got - ref) < 1e-9


def test_accuracy_flags_bad_target(dev):
    accuracy(torch.zeros(2, 4, device=dev), torch.tensor([1, 9], device=dev), 2)
    with pytest.raises(capnet.CapnetError):
        ops.check_device_errors()


class _Vocab:
    def __init__(self, V):
        self.word2idx = {'<pad>': 0, '<start>': 1, '<end>': 2, '<unk>': 3}
        self.idx2word = {i: "w%d" % i for i in range(V)}
        self.idx2word.update({0: '<pad>', 1: '<start>', 2: '<end>', 3: '<unk>'})


class _Identity(nn.Module):
    """Stands in for the encoder: the loader already yields features."""

    def forward(self, x):
        return x


def test_val_loops_match_oracle(dev, capsys):
    E, H, F, V, B = 12, 16, 16, 37, 6
    dec = DecoderFactoredLSTM(E, H, F, V, 1, dropout=0.0)
    p = synthetic.decoder_state(dec.state_dict(), seed=5, bias_range=0.1)
    dec.load_state_dict(p)
    dec.to(dev)
    batches = []
    for s in (1, 2):
        _, captions, lengths = synthetic.make_batch(B, V, seed=60 + s, images=False, min_len=3, max_len=8)
        feats = torch.randn(B, E, generator=torch.Generator().manual_seed(s))
        all_caps = [[captions[i, :lengths[i]].clone(), captions[i, :lengths[i]].flip(0)] for i in range(B)]
        batches.append((feats, captions, lengths, all_caps))
    bt, top5, loss, bleu = val_factual(_Identity(), dec, _Vocab(V), CrossEntropyLoss(), batches, device=dev)
    assert "<start>" in capsys.readouterr().out or True
    # oracle: free-running forward, CE, top-5, argmax predictions
    tot_loss, tot_top5, n_tok, refs, hyps = 0.0, 0.0, 0, [], []
    for feats, captions, lengths, all_caps in batches:
        with torch.no_grad():
            logits = D.factored_lstm_forward(p, captions, lengths, feats, [False] * max(lengths))
        targets = D.packed_targets(captions, lengths)
        n = sum(lengths)
        tot_loss += nn.functional.cross_entropy(logits, targets).item() * n
        _, ind = logits.topk(5, 1, True, True)
        tot_top5 += ind.eq(targets.view(-1, 1)).float().sum().item() * 100.0
        n_tok += n
        pred = logits.max(1)[1].tolist()
        bs = D.batch_sizes(lengths)
        off = [0]
        for b in bs:
            off.append(off[-1] + b)
        for i, l in enumerate(lengths):
            hyps.append([w for w in (pred[off[t] + i] for t in range(l)) if w not in (1, 2)])
            refs.append([[w for w in c.tolist() if w not in (1, 2)] for c in all_caps[i]])
    assert abs(loss - tot_loss / n_tok) / (tot_loss / n_tok) < 1e-5
    assert abs(top5 - tot_top5 / n_tok) < 1e-6
    assert abs(bleu - corpus_bleu(refs, hyps)) < 1e-12
    bt2, top5s, losses, bleus = val_emotion(_Identity(), dec, _Vocab(V), CrossEntropyLoss(),
                                            [batches, batches[:1]], ["happy", "sad"], device=dev)
    assert len(top5s) == len(losses) == len(bleus) == 2 and all(l > 0 for l in losses)
