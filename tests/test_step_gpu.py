"""Whole train step on the GPU against the CPU oracle (north star: training loss within 1e-4
relative for fixed seeds): images -> ResNet-152 trunk (train-mode BN) -> Linear + BatchNorm1d ->
decoder (scheduled sampling) -> NLL -> backward -> clamp -> Adam, three consecutive steps, so
the second and third losses also check the update (including the running-statistics path).
Batch 8 = BASELINE configs[0] (the reference's own CPU-runnable case); weights are the seeded
synthetic ones on both sides; dropout 0 (the GPU's dropout stream is its own)."""
import random

import pytest
import torch
import torch.nn.functional as Fn

import capnet
from capnet import synthetic
from capnet.model import DecoderFactoredLSTM, EncoderCNN
from capnet.nic_model import DecoderRNN
from capnet.optim import Adam
from capnet.train import CrossEntropyLoss, train_step
from oracle import decoders_ref as D
from oracle import step_ref as S
from oracle.resnet152_ref import EncoderCNNRef

pytestmark = pytest.mark.gpu


def _encoder_state(enc):
    sd = enc.state_dict()
    new = synthetic.trunk_state({k: v for k, v in sd.items() if k.startswith("resnet.")}, seed=1234)
    new["linear.weight"] = synthetic.param_tensor("linear.weight", sd["linear.weight"].shape, 1234, "xavier")
    new["linear.bias"] = synthetic.param_tensor("linear.bias", sd["linear.bias"].shape, 1234, "bias", 0.05)
    new["bn.weight"] = synthetic.param_tensor("bn.weight", sd["bn.weight"].shape, 1234, "bias", 0.5) + 1.0
    new["bn.bias"] = synthetic.param_tensor("bn.bias", sd["bn.bias"].shape, 1234, "bias", 0.2)
    for k in ("bn.running_mean", "bn.running_var", "bn.num_batches_tracked"):
        new[k] = sd[k]
    return new


@pytest.mark.parametrize("kind", ["nic", "factored"])
def test_three_train_steps_match_the_cpu_oracle(dev, kind):
    B, V, lr, clip, steps = 8, 8192, 2e-3, 0.5, 3
    torch.set_num_threads(16)
    enc = EncoderCNN(300)
    est = _encoder_state(enc)
    enc.load_state_dict(est)
    if kind == "nic":
        dec = DecoderRNN(300, 512, V, 1, dropout=0.0)
    else:
        dec = DecoderFactoredLSTM(300, 512, 512, V, 1, dropout=0.0)
    p = synthetic.decoder_state(dec.state_dict(), seed=1234)
    dec.load_state_dict(p)
    imgs, captions, lengths = synthetic.make_batch(B, V, seed=0)
    random.seed(3)
    tfs = [[random.random() < 0.8 for _ in range(max(lengths))] for _ in range(steps)]

    # ---- CPU oracle
    ref_enc = EncoderCNNRef(300)
    ref_enc.load_state_dict({k: v.clone() for k, v in est.items()})
    ref_enc.train()
    p_ref = {k: v.clone() for k, v in p.items()}
    opt_ref = S.AdamRef(lr=lr)
    ref_losses = []
    for it in range(steps):
        leaves = {k: v.detach().clone().requires_grad_(True) for k, v in p_ref.items()}
        feats = ref_enc(imgs)
        if kind == "nic":
            logits = D.lstm_forward(leaves, captions, lengths, feats, tfs[it])
        else:
            logits = D.factored_lstm_forward(leaves, captions, lengths, feats, tfs[it], "factual")
        loss = Fn.cross_entropy(logits, D.packed_targets(captions, lengths))
        ref_enc.zero_grad()
        loss.backward()
        grads = {k: v.grad for k, v in leaves.items()}
        S.clip_gradient_(grads.values(), clip)
        with torch.no_grad():
            hp = {("enc." + k): v for k, v in ref_enc.named_parameters() if not k.startswith("resnet.")}
            hg = {k: v.grad for k, v in hp.items()}
            S.clip_gradient_([g for g in hg.values() if g is not None], clip)
            both, both_g = dict(p_ref), dict(grads)
            both.update(hp)
            both_g.update(hg)
            opt_ref.step(both, both_g)
        ref_losses.append(float(loss.detach()))

    # ---- GPU product
    enc.to(dev).train()
    dec.to(dev).train()
    params = list(dec.parameters()) + list(enc.linear.parameters()) + list(enc.bn.parameters())
    opt = Adam(params, lr=lr)
    crit = CrossEntropyLoss()
    imgs_d, caps_d = imgs.to(dev), captions.to(dev)
    got = []
    for it in range(steps):
        got.append(float(train_step(enc, dec, opt, crit, imgs_d, caps_d, lengths, clip, tf_mask=tfs[it]).item()))
    capnet.ops.check_device_errors()
    print(kind, "oracle", ref_losses, "gpu", got)
    for a, b in zip(got, ref_losses):
        assert abs(a - b) / abs(b) < 1e-4, (got, ref_losses)
    assert ref_losses[2] < ref_losses[0]      # the updates did something
