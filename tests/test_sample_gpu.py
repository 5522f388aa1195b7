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
