"""The drop-in boundary in its strongest form (VERDICT r2 missing #2 / #3):

* the reference's loop body run UNMODIFIED over the capnet modules -- `torch.optim.Adam`, `nn.CrossEntropyLoss()`,
  `pack_padded_sequence(...)[0]`, the reference's torch `clip_gradient` (stylenet/train_multitask.py:134,163-167,373-389,
  stylenet/utils.py:51-60) -- gives the losses of the capnet.optim.Adam / capnet.train.CrossEntropyLoss path;
* `save_checkpoint` from device-resident modules + capnet.optim.Adam after two steps, `load_checkpoint` into fresh
  objects, and step three is the uninterrupted run's step three (stylenet/utils.py:63-90, resume at
  stylenet/train_multitask.py:168-177)."""
import random

import pytest
import torch
import torch.nn as nn
from torch.nn.utils.rnn import pack_padded_sequence

import capnet
from capnet import synthetic
from capnet.model import DecoderFactoredLSTM, EncoderCNN
from capnet.optim import Adam
from capnet.train import CrossEntropyLoss, train_step
from capnet.utils import load_checkpoint, save_checkpoint
from test_step_gpu import _encoder_state

pytestmark = pytest.mark.gpu
V, B, LR, CLIP = 1000, 8, 2e-3, 0.5


def _build(dev, seed=1234):
    enc = EncoderCNN(300)
    enc.load_state_dict(_encoder_state(enc))
    dec = DecoderFactoredLSTM(300, 512, 512, V, 1, dropout=0.0)
    dec.load_state_dict(synthetic.decoder_state(dec.state_dict(), seed=seed))
    enc.to(dev).train()
    dec.to(dev).train()
    return enc, dec


def _params(enc, dec):
    # stylenet/train_multitask.py:163-164
    return list(dec.parameters()) + list(enc.linear.parameters()) + list(enc.bn.parameters())


def test_reference_loop_body_runs_unmodified_over_the_capnet_modules(dev):
    batches = [synthetic.make_batch(B, V, seed=s) for s in range(3)]

    # ---- the reference's objects and loop body
    encoder, decoder = _build(dev)
    criterion = nn.CrossEntropyLoss().to(dev)
    optimizer = torch.optim.Adam(params=_params(encoder, decoder), lr=LR)

    def clip_gradient(optimizer, grad_clip):                     # stylenet/utils.py:51-60, torch arithmetic
        for group in optimizer.param_groups:
            for param in group['params']:
                if param.grad is not None:
                    param.grad.data.clamp_(-grad_clip, grad_clip)

    random.seed(17)
    ref_style = []
    for images, captions, lengths in batches:
        images = images.to(dev)
        captions = captions.to(dev)
        targets = pack_padded_sequence(input=captions, lengths=lengths, batch_first=True)[0]
        features = encoder(images)
        outputs = decoder(captions, lengths, features)          # scheduled sampling draws from `random`
        loss = criterion(outputs, targets)
        decoder.zero_grad()
        encoder.zero_grad()
        loss.backward()
        clip_gradient(optimizer, CLIP)
        optimizer.step()
        ref_style.append(loss.item())

    # ---- the same steps on the fused path
    enc, dec = _build(dev)
    opt = Adam(_params(enc, dec), lr=LR)
    crit = CrossEntropyLoss()
    random.seed(17)
    fused = []
    for images, captions, lengths in batches:
        fused.append(train_step(enc, dec, opt, crit, images.to(dev), captions.to(dev), lengths, CLIP).item())
    capnet.ops.check_device_errors()
    print("reference-style loop", ref_style, "fused loop", fused)
    for a, b in zip(ref_style, fused):
        assert abs(a - b) / abs(b) < 1e-6
    assert fused[-1] < fused[0]
    # and the parameters the two optimisers left behind agree (on average: Adam's first steps are sign-like, so single
    # elements whose gradient is of the order of eps = 1e-8 move by up to lr either way on a last-bit difference)
    for (k, p), q in zip(decoder.state_dict().items(), dec.state_dict().values()):
        assert float((p - q).abs().mean()) < 1e-6 and float((p - q).abs().max()) <= 3 * 2 * LR, k


def test_checkpoint_round_trip_resumes_to_the_same_loss(dev, tmp_path):
    batches = [synthetic.make_batch(B, V, seed=10 + s) for s in range(3)]
    random.seed(2)
    tfs = [[random.random() < 0.8 for _ in range(24)] for _ in range(3)]

    def step(enc, dec, opt, i):
        images, captions, lengths = batches[i]
        return train_step(enc, dec, opt, CrossEntropyLoss(), images.to(dev), captions.to(dev), lengths, CLIP,
                          tf_mask=tfs[i][:max(lengths)]).item()

    enc, dec = _build(dev)
    opt = Adam(_params(enc, dec), lr=LR)
    straight = [step(enc, dec, opt, i) for i in range(3)]

    enc, dec = _build(dev)
    opt = Adam(_params(enc, dec), lr=LR)
    first = [step(enc, dec, opt, i) for i in range(2)]
    save_checkpoint(str(tmp_path), "synthetic", "factual", 3, 0, enc, dec, opt, None, 0.25, True)
    del enc, dec, opt

    enc2, dec2 = _build(dev, seed=99)                                # other weights: everything must come from the file
    opt2 = Adam(_params(enc2, dec2), lr=LR * 10)
    meta = load_checkpoint(str(tmp_path / "factual_BEST_checkpoint_synthetic.pth.tar"), enc2, dec2, opt2,
                           map_location=dev)
    assert meta == {"epoch": 3, "epochs_since_improvement": 0, "bleu-4": 0.25}
    assert opt2.param_groups[0]["lr"] == LR
    enc2.train()
    dec2.train()
    resumed = step(enc2, dec2, opt2, 2)
    capnet.ops.check_device_errors()
    print("uninterrupted", straight, "resumed", first + [resumed])
    assert first == straight[:2]
    assert resumed == straight[2]                                    # same kernels, same state: the same number
    # the file holds plain state_dicts with the reference's keys (loadable with weights_only=True, as load_checkpoint does)
    state = torch.load(str(tmp_path / "factual_checkpoint_synthetic.pth.tar"), map_location="cpu", weights_only=True)
    assert set(state) == {"epoch", "epochs_since_improvement", "bleu-4", "encoder", "decoder", "optimizer", "lang_optimizer"}
    assert "C.weight" in state["decoder"] and "linear.weight" in state["encoder"]
