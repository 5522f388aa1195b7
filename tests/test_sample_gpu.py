"""GPU parity of beam search `sample()` (stylenet/model.py:198-294, nic/model.py:117-207,
stylenet/model_att.py:307-426) against sequences produced by the reference's OWN methods
(tests/golden/sample_tiny.npz, tools/gen_golden.py sample_tiny), plus the fused
log-softmax/top-k kernel against torch."""
import pytest
import torch

import capnet
from capnet import ops
from capnet.model import DecoderFactoredLSTM
from capnet.model_att import DecoderFactoredLSTMAtt
from capnet.nic_model import DecoderRNN
from helpers import load_golden, t

pytestmark = pytest.mark.gpu

Z = load_golden("sample_tiny.npz")
CASES = [str(c) for c in Z["cases"]]


def _case(name):
    pre = "case.%s." % name
    c = {k[len(pre):]: Z[k] for k in Z.files if k.startswith(pre)}
    params = {k[len("param."):]: t(v) for k, v in c.items() if k.startswith("param.")}
    return c, params


@pytest.mark.parametrize("name", CASES)
def test_sample_matches_reference_sequences(dev, name):
    c, params = _case(name)
    kind = str(c["kind"])
    start, end = [int(v) for v in Z["start_end"]]
    dims = [int(v) for v in c["dims"]]
    E, H, F, V, k, maxlen = dims[:6]
    if kind == "factored":
        dec = DecoderFactoredLSTM(E, H, F, V, 1, dropout=0.0, max_seq_length=maxlen)
    elif kind == "nic":
        dec = DecoderRNN(E, H, V, 1, dropout=0.0, max_seq_length=maxlen)
    else:
        dec = DecoderFactoredLSTMAtt(dims[6], E, H, F, V, 1, feature_size=dims[7], dropout=0.0,
                                     max_seq_length=maxlen)
    dec.load_state_dict(params)
    dec.to(dev).eval()
    if kind == "nic":
        seq = dec.sample(torch.zeros(1, E, device=dev), start, end, k=k)
    elif kind == "factored":
        seq = dec.sample(torch.zeros(1, E, device=dev), start, end, k=k, mode=str(c["mode"]))
    else:
        seq = dec.sample(t(c["features"]).to(dev), start, end, k=k, mode=str(c["mode"]))
    ops.check_device_errors()
    assert seq.dtype == torch.int64 and seq.dim() == 2 and seq.shape[0] == 1
    assert seq.cpu().tolist() == c["seq"].tolist()


@pytest.mark.parametrize("rows,V,k", [(1, 37, 5), (5, 8192, 5), (3, 7411, 3), (16, 1000, 16), (2, 5, 7)])
def test_beam_topk_matches_torch(dev, rows, V, k):
    g = torch.Generator().manual_seed(rows * 1000 + V + k)
    logits = (torch.randn(rows, V, generator=g) * 3.0)
    prev = torch.randn(rows, generator=g)
    scores, flat = ops.beam_topk(logits.to(dev), prev.to(dev), rows, k)
    ref = (prev.double().unsqueeze(1) + torch.log_softmax(logits.double(), dim=1)).view(-1)
    ref_scores, ref_flat = ref.topk(k, 0, True, True)
    assert flat.cpu().tolist() == ref_flat.tolist()
    assert (scores.cpu().double() - ref_scores).abs().max().item() < 5e-6


def test_beam_topk_first_step_uses_row_zero_only(dev):
    logits = torch.randn(4, 50, generator=torch.Generator().manual_seed(3))
    logits[2, 7] = 100.0     # would win if all rows competed
    scores, flat = ops.beam_topk(logits.to(dev), torch.zeros(4, device=dev), 1, 4)
    assert max(flat.cpu().tolist()) < 50


def test_beam_topk_rejects_bad_arguments(dev):
    x = torch.zeros(2, 4, device=dev)
    with pytest.raises(capnet.CapnetError):
        ops.beam_topk(x, torch.zeros(2, device=dev), 2, 9)      # k > rows * V
    with pytest.raises(capnet.CapnetError):
        ops.beam_topk(torch.zeros(17, 4, device=dev), torch.zeros(17, device=dev), 17, 2)


def test_attention_forward_single_step_matches_oracle(dev):
    from oracle import decoders_ref as D
    c, params = _case("att_factual_k5")
    dims = [int(v) for v in c["dims"]]
    E, H, F, V, k, maxlen, A, Cf = dims
    dec = DecoderFactoredLSTMAtt(A, E, H, F, V, 1, feature_size=Cf, dropout=0.0)
    dec.load_state_dict(params)
    dec.to(dev).eval()
    g = torch.Generator().manual_seed(5)
    feat = torch.randn(3, 4, Cf, generator=g)
    h = torch.randn(3, H, generator=g)
    awe, alpha = dec.attention(feat.to(dev), h.to(dev))
    awe_ref, alpha_ref = D.attention_step(params, "attention", feat, h)
    assert (alpha.cpu() - alpha_ref).abs().max().item() < 2e-6
    assert (awe.cpu() - awe_ref).abs().max().item() < 2e-5 * awe_ref.abs().max().item() + 1e-6
    h0, c0 = dec.init_hidden_state(feat.to(dev))
    mean = feat.mean(dim=1)
    assert (h0.cpu() - D._lin(params, "init_h", mean)).abs().max().item() < 1e-5
    assert (c0.cpu() - D._lin(params, "init_c", mean)).abs().max().item() < 1e-5


# ---- the batched beam search of the test-set evaluator (stylenet/evaluator.py:63-120) ---------------------------------
def test_beam_topk_batched_matches_the_single_image_kernel(dev):
    g = torch.Generator().manual_seed(11)
    V = 777
    rows = [5, 1, 3, 0, 4]            # live beams per image (image 3 has finished)
    ks = [5, 1, 2, 0, 4]
    first = [0, 0, 0, 0, 1]           # image 4 is at its first step: only its row 0 competes
    logits = (torch.randn(sum(rows), V, generator=g) * 3.0).to(dev)
    prev = torch.randn(sum(rows), generator=g).to(dev)
    meta, r0 = [], 0
    for r, k, f in zip(rows, ks, first):
        meta.append((r0, (1 if f else r) if r else 0, k))
        r0 += r
    sc, ix = ops.beam_topk_batched(logits, prev, torch.tensor(meta, dtype=torch.int32, device=dev))
    for i, (r0, r, k) in enumerate(meta):
        if k == 0:
            continue
        s1, i1 = ops.beam_topk(logits[r0:r0 + max(r, 1)], prev[r0:r0 + max(r, 1)], r, k)
        assert ix[i, :k].tolist() == i1.tolist() and torch.equal(sc[i, :k], s1)


@pytest.mark.parametrize("name", CASES)
def test_sample_batch_equals_sample_per_image(dev, name):
    """decoder.sample_batch (all images' beams advance together) returns, image by image, what decoder.sample returns --
    which tests above pin to the reference's own sequences."""
    c, params = _case(name)
    kind = str(c["kind"])
    start, end = [int(v) for v in Z["start_end"]]
    dims = [int(v) for v in c["dims"]]
    E, H, F, V, k, maxlen = dims[:6]
    if kind == "factored":
        dec = DecoderFactoredLSTM(E, H, F, V, 1, dropout=0.0, max_seq_length=maxlen)
    elif kind == "nic":
        dec = DecoderRNN(E, H, V, 1, dropout=0.0, max_seq_length=maxlen)
    else:
        dec = DecoderFactoredLSTMAtt(dims[6], E, H, F, V, 1, feature_size=dims[7], dropout=0.0, max_seq_length=maxlen)
    dec.load_state_dict(params)
    dec.to(dev).eval()
    kw = {} if kind == "nic" else {"mode": str(c["mode"])}
    if kind == "att":
        base = t(c["features"]).to(dev)
        g = torch.Generator().manual_seed(3)
        feats = torch.cat([base] + [base * (0.5 + torch.rand(1, generator=g).item()) + 0.3 * torch.randn(base.shape, generator=g).to(dev)
                                    for _ in range(4)], 0)
    else:
        feats = torch.zeros(3, E, device=dev)
    got = dec.sample_batch(feats, start, end, k=k, **kw)
    want = [dec.sample(feats[i:i + 1], start, end, k=k, **kw)[0].tolist() for i in range(feats.shape[0])]
    assert got == want
    assert got[0] == c["seq"].tolist()[0]
    ops.check_device_errors()


def test_evaluate_reports_the_four_bleu_scores(dev):
    """capnet.train.evaluate == decoder.sample per image + corpus BLEU with the reference's four weight tuples."""
    from capnet.metrics import corpus_bleu
    from capnet.train import evaluate
    c, params = _case("att_factual_k5")
    dims = [int(v) for v in c["dims"]]
    E, H, F, V, k, maxlen, A, Cf = dims
    dec = DecoderFactoredLSTMAtt(A, E, H, F, V, 1, feature_size=Cf, dropout=0.0, max_seq_length=maxlen)
    dec.load_state_dict(params)
    dec.to(dev).eval()
    start, end = [int(v) for v in Z["start_end"]]

    class Vocab:
        word2idx = {"<start>": start, "<end>": end}
        idx2word = {i: ("<end>" if i == end else "<start>" if i == start else "w%d" % i) for i in range(V)}

    class Enc(torch.nn.Module):
        def forward(self, images):
            return images                      # the "images" of this loader ARE feature maps
    g = torch.Generator().manual_seed(8)
    base = t(c["features"])
    batches = []
    for b in range(2):
        feats = torch.cat([base * (0.6 + 0.2 * i + b) + 0.2 * torch.randn(base.shape, generator=g) for i in range(3)], 0)
        caps = [[torch.tensor([start] + torch.randint(3, V, (4,), generator=g).tolist() + [end]) for _ in range(2)] for _ in range(3)]
        batches.append((feats, None, None, caps))
    got = evaluate(Enc(), dec, Vocab(), batches, mode="factual", k=k)
    refs, hyps = [], []
    for feats, _, _, caps in batches:
        for i in range(3):
            hyps.append(dec.sample(feats[i:i + 1].to(dev), start, end, k=k, mode="factual")[0].tolist())
            refs.append([cc.tolist() for cc in caps[i]])
    want = tuple(corpus_bleu(refs, hyps, weights=w) for w in ((1, 0, 0, 0), (0.5, 0.5, 0, 0), (0.33, 0.33, 0.33, 0), (0.25,) * 4))
    assert got == want and len(got) == 4 and all(0.0 <= b <= 1.0 for b in got)
