"""capnet.stacked.StackedFactoredLSTM (BASELINE configs[3] / [4]: "2-layer", "3-layer"). PARITY UNPINNED: the reference
ignores num_layers (stylenet/model.py:37); the semantics are SURVEY App. A-1's. Checked: against the CPU restatement of
that definition (oracle/decoders_ref.py: logits, loss and every gradient, scheduled sampling included), and with one
layer against DecoderFactoredLSTM itself (which IS pinned to the reference)."""
import random

import pytest
import torch
import torch.nn.functional as Fn

import capnet
from capnet import ops, synthetic
from capnet.model import DecoderFactoredLSTM
from capnet.stacked import StackedFactoredLSTM
from oracle import decoders_ref as D
from helpers import rel_err

pytestmark = pytest.mark.gpu


def _state(dec, seed):
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for k, v in dec.state_dict().items():
        lim = 0.3 if v.dim() > 1 else 0.05
        sd[k] = (torch.rand(v.shape, generator=g) * 2 - 1) * lim
    return sd


@pytest.mark.parametrize("layers,mode", [(2, "factual"), (3, "happy"), (2, "sad")])
def test_stacked_decoder_matches_its_cpu_restatement(dev, layers, mode):
    E, H, F, V, B = 20, 24, 16, 61, 5
    dec = StackedFactoredLSTM(E, H, F, V, layers, dropout=0.0)
    p = _state(dec, 7 + layers)
    dec.load_state_dict(p)
    dec.to(dev).train()
    _, caps, lens = synthetic.make_batch(B, V, seed=3, min_len=4, max_len=8)
    feats = torch.randn(B, E, generator=torch.Generator().manual_seed(1))
    random.seed(11)
    tf = [random.random() < 0.6 for _ in range(max(lens))]
    tf[0] = True
    assert not all(tf)
    out = dec(caps.to(dev), lens, feats.to(dev), mode=mode, tf_mask=tf)
    loss = ops.cross_entropy(out, ops.packed_targets(caps.to(dev), lens))
    loss.backward()
    leaves = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    ref = D.stacked_factored_lstm_forward(leaves, caps, lens, feats, tf, mode, layers)
    ref_loss = Fn.cross_entropy(ref, D.packed_targets(caps, lens))
    ref_loss.backward()
    assert rel_err(out, ref.detach()) < 2e-5
    assert abs(loss.item() - ref_loss.item()) / ref_loss.item() < 1e-5
    got = dict(dec.named_parameters())
    n_checked = 0
    for k, leaf in leaves.items():
        if leaf.grad is None or float(leaf.grad.abs().max()) == 0.0:
            assert got[k].grad is None or float(got[k].grad.abs().max()) == 0.0, k     # the other modes' S
            continue
        assert rel_err(got[k].grad, leaf.grad) < 2e-4, k
        n_checked += 1
    assert n_checked >= 1 + layers * 16 + 2


def test_one_layer_is_the_reference_decoder(dev):
    E, H, F, V, B = 300, 512, 512, 1000, 6
    ref = DecoderFactoredLSTM(E, H, F, V, 1, dropout=0.0)
    p = synthetic.decoder_state(ref.state_dict(), seed=5)
    ref.load_state_dict(p)
    dec = StackedFactoredLSTM(E, H, F, V, 1, dropout=0.0)
    assert list(dec.state_dict().keys()) == list(ref.state_dict().keys())
    dec.load_state_dict(p)
    ref.to(dev).train()
    dec.to(dev).train()
    _, caps, lens = synthetic.make_batch(B, V, seed=9)
    feats = torch.randn(B, E, generator=torch.Generator().manual_seed(2)).to(dev)
    random.seed(4)
    tf = [random.random() < 0.8 for _ in range(max(lens))]
    a = ref(caps.to(dev), lens, feats, tf_mask=tf)
    b = dec(caps.to(dev), lens, feats, tf_mask=tf)
    assert rel_err(b, a) < 1e-5
    la = ops.cross_entropy(a, ops.packed_targets(caps.to(dev), lens)); la.backward()
    lb = ops.cross_entropy(b, ops.packed_targets(caps.to(dev), lens)); lb.backward()
    assert rel_err(dec.W_i.weight.grad, ref.W_i.weight.grad) < 1e-4
    assert rel_err(dec.B.weight.grad, ref.B.weight.grad) < 1e-4


@pytest.mark.parametrize("engine_a", ["c", "python"])
def test_runs_through_the_persistent_kernel_equal_the_step_by_step_engine(dev, engine_a):
    """Three layers at the size the persistent kernel takes (H = 512): logits, loss and every gradient of (c) the whole
    recurrence as one C call each way (capnet_seq_forward_stacked / _backward_stacked) and of (python) the engine that runs
    each teacher-forced run of a layer in one launch (capnet.stacked.LstmRunFn) against the step-by-step engine (W GEMM +
    cell kernel per step, torch autograd composing the backward), same weights, same scheduled-sampling decisions,
    shrinking batches."""
    E, H, F, V, B, layers = 300, 512, 256, 500, 9, 3
    a = StackedFactoredLSTM(E, H, F, V, layers, dropout=0.0)
    p = synthetic.decoder_state(a.state_dict(), seed=21)
    a.load_state_dict(p)
    b = StackedFactoredLSTM(E, H, F, V, layers, dropout=0.0)
    b.load_state_dict(p)
    b.engine, b.fast_runs = "python", False
    a.engine = engine_a
    a.to(dev).train()
    b.to(dev).train()
    _, caps, lens = synthetic.make_batch(B, V, seed=4)
    feats = torch.randn(B, E, generator=torch.Generator().manual_seed(3)).to(dev)
    random.seed(8)
    tf = [random.random() < 0.75 for _ in range(max(lens))]
    assert not all(tf) and sum(tf) >= 4
    oa = a(caps.to(dev), lens, feats, mode="happy", tf_mask=tf)
    ob = b(caps.to(dev), lens, feats, mode="happy", tf_mask=tf)
    assert rel_err(oa, ob) < 1e-5
    la = ops.cross_entropy(oa, ops.packed_targets(caps.to(dev), lens)); la.backward()
    lb = ops.cross_entropy(ob, ops.packed_targets(caps.to(dev), lens)); lb.backward()
    ops.check_device_errors()
    assert abs(la.item() - lb.item()) / lb.item() < 1e-6
    n = 0
    for (k, pa), pb in zip(a.named_parameters(), b.parameters()):
        if pb.grad is None:
            assert pa.grad is None, k
            continue
        assert rel_err(pa.grad, pb.grad) < 1e-4, k
        n += 1
    assert n >= 1 + layers * 16 + 2


def test_stacked_train_step_runs_and_learns(dev):
    """configs[4]'s decoder shape (3 layers, factored 1024) through capnet.train.train_step with capnet.optim.Adam."""
    from capnet.optim import Adam
    from capnet.train import CrossEntropyLoss, train_step

    class Enc(torch.nn.Module):                       # a stand-in encoder head: features are given
        def __init__(self, f):
            super().__init__()
            self.f = f

        def forward(self, images):
            return self.f
    import random
    B, V = 8, 500
    torch.manual_seed(3)            # initial weights, features, dropout masks
    random.seed(3)                  # the scheduled-sampling draws (steps with free-running inputs have higher losses)
    dec = StackedFactoredLSTM(300, 512, 1024, V, 3, dropout=0.3).to(dev).train()
    feats = torch.randn(B, 300, device=dev)
    enc = Enc(feats)
    _, caps, lens = synthetic.make_batch(B, V, seed=1)
    opt = Adam(dec.parameters(), lr=1e-3)
    losses = [train_step(enc, dec, opt, CrossEntropyLoss(), None, caps.to(dev), lens, 0.5).item() for _ in range(8)]
    ops.check_device_errors()
    print("stacked train steps:", [round(x, 3) for x in losses])
    assert all(torch.isfinite(torch.tensor(losses))) and min(losses[2:]) < losses[0]


def test_eight_clamp_adam_steps_match_the_cpu_restatement(dev):
    """round 3's log of this shape (3 layers, factored 1024, lr 1e-3) showed losses
    6.21, 6.04, 5.00, 5.70, 19.54, 11.42, 8.37, 6.88 and the test was relaxed. Model or engine? Here the SAME eight steps
    -- same initial weights (seed 3 state), the scheduled-sampling masks of random.seed(3), dropout 0, element-wise clamp
    0.5 + Adam 1e-3 -- run through the engine on the GPU and through oracle.decoders_ref.stacked_factored_lstm_forward
    + oracle.step_ref on the CPU, and every loss must agree to 1e-4 relative (the north star's bound) for as long as the
    two trajectories can be compared at all: an excursion of the un-tanh'd h = o c stack at this learning rate amplifies
    rounding differences, so steps behind a loss above 2 ln V are held to 5 %.
    PERF-ONLY / PARITY UNPINNED semantics; what this pins is the engine to its restatement."""
    import math
    from capnet.optim import Adam
    from capnet.utils import clip_gradient
    from oracle import step_ref as S
    B, V, layers, steps = 8, 500, 3, 8
    dec = StackedFactoredLSTM(300, 512, 1024, V, layers, dropout=0.0)
    p = synthetic.decoder_state(dec.state_dict(), seed=3)
    dec.load_state_dict(p)
    dec.to(dev).train()
    _, caps, lens = synthetic.make_batch(B, V, seed=1)
    feats = torch.randn(B, 300, generator=torch.Generator().manual_seed(3))
    random.seed(3)
    tfs = [[random.random() < 0.8 for _ in range(max(lens))] for _ in range(steps)]
    opt = Adam(dec.parameters(), lr=1e-3)
    targets = ops.packed_targets(caps.to(dev), lens)
    gpu = []
    for it in range(steps):
        out = dec(caps.to(dev), lens, feats.to(dev), tf_mask=tfs[it])
        loss = ops.cross_entropy(out, targets)
        dec.zero_grad()
        loss.backward()
        clip_gradient(opt, 0.5)
        opt.step()
        gpu.append(loss.item())
    ops.check_device_errors()
    torch.set_num_threads(16)
    pr = {k: v.clone() for k, v in p.items()}
    oref = S.AdamRef(lr=1e-3)
    cpu = []
    for it in range(steps):
        loss, grads, _, _ = S.decoder_loss_and_grads(D.stacked_factored_lstm_forward, pr, caps, lens, feats, tfs[it],
                                                     mode="factual", num_layers=layers)
        S.clip_gradient_(grads.values(), 0.5)
        oref.step(pr, grads)
        cpu.append(loss.item())
    print("stacked 3 x 1024, eight steps: gpu", [round(x, 4) for x in gpu], "cpu", [round(x, 4) for x in cpu])
    wild = False
    for a, b in zip(gpu, cpu):
        tol = 5e-2 if wild else 1e-4
        assert abs(a - b) <= tol * abs(b), (gpu, cpu)
        wild = wild or b > 2 * math.log(V)
    assert min(cpu[1:]) < cpu[0]
