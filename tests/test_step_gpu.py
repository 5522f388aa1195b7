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


@pytest.mark.parametrize("kind,B,steps", [("nic", 8, 3), ("factored", 8, 3), ("factored", 64, 2)])
def test_three_train_steps_match_the_cpu_oracle(dev, kind, B, steps):
    """(factored, 64, 2) is BASELINE configs[1] at its full size: batch 64, V = 8192, two steps (the
    second loss also checks the update); ~10 s of CPU oracle on 16 threads."""
    V, lr, clip = 8192, 2e-3, 0.5
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

    # ---- CPU oracle (fp32 = the reference's arithmetic; fp64 = the same step without rounding noise)
    def oracle_losses(dtype):
        ref_enc = EncoderCNNRef(300)
        ref_enc.load_state_dict({k: v.clone() for k, v in est.items()})
        ref_enc.to(dtype).train()
        p_ref = {k: v.clone().to(dtype) for k, v in p.items()}
        opt_ref = S.AdamRef(lr=lr)
        out = []
        for it in range(steps):
            leaves = {k: v.detach().clone().requires_grad_(True) for k, v in p_ref.items()}
            feats = ref_enc(imgs.to(dtype))
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
            out.append(float(loss.detach()))
        return out
    ref_losses = oracle_losses(torch.float32)
    ref64 = oracle_losses(torch.float64) if B < 64 else None

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
    print(kind, "oracle", ref_losses, "gpu", got, "oracle in fp64", ref64)
    # North star: 1e-4 relative on the training loss. The first loss is the forward parity and is held to it, and at
    # BASELINE's batch of 64 so is every step. With 8 images, train-mode BatchNorm through 152 layers leaves any two
    # fp32-grade trunks ~7e-4 apart in their features and Adam's first, sign-like updates carry that into the later
    # losses -- for the reference's own fp32 arithmetic as much as for this one. That is SHOWN here, not assumed: the
    # same three steps in float64 say how far the fp32 CPU oracle itself is from the exact step, and the GPU may be no
    # further from the exact step than 1e-4 or 1.25 x that, whichever is larger (measured: oracle 0.6-1.3e-4, GPU
    # 0.4-1.2e-4 on steps 2-3).
    for i, (a, b) in enumerate(zip(got, ref_losses)):
        if i == 0 or B >= 64:
            assert abs(a - b) / abs(b) < 1e-4, (got, ref_losses)
        else:
            e_cpu = abs(b - ref64[i]) / abs(ref64[i])
            e_gpu = abs(a - ref64[i]) / abs(ref64[i])
            print("step %d: fp32 oracle vs fp64 %.2e, GPU vs fp64 %.2e, GPU vs fp32 oracle %.2e" % (i, e_cpu, e_gpu, abs(a - b) / abs(b)))
            assert e_gpu < max(1e-4, 1.25 * e_cpu), (got, ref_losses, ref64)
    assert ref_losses[-1] < ref_losses[0]     # the updates did something


def _pipeline_vs_sequential(dev, tuning, graph=False, timing_every=0, keep=None):
    from capnet.train import TrunkPipeline
    B, V, steps = 8, 1000, 5
    batches = [synthetic.make_batch(B, V, seed=s) for s in range(steps)]
    random.seed(5)
    tfs = [[random.random() < 0.8 for _ in range(24)] for _ in range(steps)]

    def build():
        enc = EncoderCNN(300)
        enc.load_state_dict(_encoder_state(enc))
        dec = DecoderFactoredLSTM(300, 512, 512, V, 1, dropout=0.0)
        dec.load_state_dict(synthetic.decoder_state(dec.state_dict(), seed=1234))
        enc.to(dev).train()
        dec.to(dev).train()
        params = list(dec.parameters()) + list(enc.linear.parameters()) + list(enc.bn.parameters())
        return enc, dec, Adam(params, lr=2e-3)

    enc, dec, opt = build()
    seq = []
    for (imgs, caps, lens), tf in zip(batches, tfs):
        seq.append(train_step(enc, dec, opt, CrossEntropyLoss(), imgs.to(dev), caps.to(dev), lens, 0.5,
                              tf_mask=tf[:max(lens)]))
    seq = [float(l.item()) for l in seq]
    ref_params = {k: v.detach().clone() for k, v in dec.state_dict().items()}
    ref_rm = enc.resnet[7][2].bn3.running_mean.clone()

    enc, dec, opt = build()
    if keep is not None:
        keep["enc"] = enc
    if timing_every:
        # bench.py's conv events beside graph replay: every N-th pass is launched directly and bracketed
        runner = enc._trunk()
        runner.set_timing(runner._plan(B, 224, 224, dev), timing_every)
    pipe = TrunkPipeline(enc, dec, opt, CrossEntropyLoss(), 0.5, shared_chip_tuning=tuning,
                         graph_trunk=graph)
    dev_batches = [(i.to(dev), c.to(dev), l) for i, c, l in batches]
    d = pipe.depth
    assert d == 3 and steps > d
    for k in range(d):
        pipe.prefetch(dev_batches[k][0])      # `depth` trunk passes in flight
    with pytest.raises(RuntimeError):
        pipe.prefetch(dev_batches[d][0])      # one more is refused
    got = []
    for k, ((imgs, caps, lens), tf) in enumerate(zip(dev_batches, tfs)):
        nxt = dev_batches[k + d][0] if k + d < steps else None
        got.append(pipe.step(caps, lens, next_images=nxt, tf_mask=tf[:max(lens)]))
    assert pipe.in_flight() == 0
    pipe.finish()
    torch.cuda.synchronize()
    got = [float(l.item()) for l in got]
    print("tuning", tuning, "sequential", seq, "pipelined", got)
    assert int(enc.resnet[1].num_batches_tracked) == steps
    with pytest.raises(RuntimeError):
        pipe.step(dev_batches[0][1], dev_batches[0][2])                # nothing prefetched
    return seq, got, ref_params, dec.state_dict(), ref_rm, enc.resnet[7][2].bn3.running_mean


def test_pipelined_steps_equal_sequential_steps(dev):
    """capnet.train.TrunkPipeline with the sequential schedule's kernels (shared_chip_tuning=False):
    three trunk passes in flight, the decoder half on a side stream -- and the numbers of the
    sequential loop: same losses step by step, same parameters, same running statistics."""
    seq, got, ref_params, params, ref_rm, rm = _pipeline_vs_sequential(dev, False)
    for a, b in zip(got, seq):
        assert abs(a - b) / abs(b) < 2e-6
    for k, v in params.items():
        dlt = (v - ref_params[k]).abs()
        if k == "B.weight":
            # (the embedding gradient was a scatter with float atomics until round 4 -- order varied run to run; it is
            #  summed in row order now, the allowance stays for Adam's sign-like first steps on +-1e-9 gradients)
            assert (dlt > 1e-6).float().mean().item() < 1e-3
        else:
            # (the differing embedding rows feed the later steps, so the rest agrees closely, not bitwise)
            assert dlt.max().item() <= 2e-3 * ref_params[k].abs().max().item() + 1e-7, k
    # the trunk's running statistics saw the passes in order (deferred, event-ordered updates)
    assert torch.equal(rm, ref_rm)


def test_pipelined_steps_with_graphed_trunk_equal_sequential_steps(dev):
    """graph_trunk=True: every trunk pass replayed from a hipGraph captured per pipeline slot."""
    seq, got, _, _, ref_rm, rm = _pipeline_vs_sequential(dev, False, graph=True)
    for a, b in zip(got, seq):
        assert abs(a - b) / abs(b) < 2e-6
    assert torch.equal(rm, ref_rm)


def test_graphed_trunk_with_every_second_pass_timed(dev):
    """Events cannot ride in a replayed graph: with conv timing on, every N-th pass is launched directly and bracketed
    (capnet_trunk_time_next_pass), the others replay -- same numbers, and the timed passes' launches are collected."""
    import ctypes as C
    from capnet._lib import check, lib
    keep = {}
    seq, got, _, _, ref_rm, rm = _pipeline_vs_sequential(dev, False, graph=True, timing_every=2, keep=keep)
    for a, b in zip(got, seq):
        assert abs(a - b) / abs(b) < 2e-6
    assert torch.equal(rm, ref_rm)
    runner = keep["enc"]._trunk()
    plan = runner._plan(8, 224, 224, dev)
    ms, n, fl = C.c_double(), C.c_long(), C.c_double()
    check(lib().capnet_trunk_collect_timing(plan["handle"], C.byref(ms), C.byref(n), C.byref(fl)))
    # five passes, the 1st, 3rd and 5th launched directly: 3 x (155 convolutions + the fused boundaries' statistics launches)
    assert 3 * 155 <= n.value <= 3 * 230 and ms.value > 0
    assert abs(fl.value / (3 * 8 * 23.02e9) - 1) < 0.02            # 23.02 GFLOP per image
    runner.set_timing(plan, False)


def test_pipelined_steps_with_shared_chip_tuning_stay_close(dev):
    """Default pipeline: no tail balancing and the 128x64 tile on the large layers while passes
    share the chip. Only the order of fp32 additions changes (BatchNorm partial sums), which an
    8-image batch statistic turns into a ~3e-5 difference of the first loss; Adam's sign-like first
    updates then let the two runs drift apart slowly."""
    seq, got, ref_params, params, ref_rm, rm = _pipeline_vs_sequential(dev, True)
    assert abs(got[0] - seq[0]) / abs(seq[0]) < 1e-4
    for a, b in zip(got, seq):
        assert abs(a - b) / abs(b) < 2e-3
    assert (rm - ref_rm).abs().max().item() <= 5e-3 * ref_rm.abs().max().item()


def test_training_loops_run_pipelined_and_sequential(dev, capsys):
    """train_factual / train_emotion (stylenet/train_multitask.py:364-408, 511-557) on a tiny
    in-memory loader: the pipelined loop reports the losses of the sequential one."""
    from capnet.train import train_emotion, train_factual
    B, V = 2, 200
    loader = []
    for s in range(3):
        imgs, caps, lens = synthetic.make_batch(B, V, seed=10 + s, min_len=3, max_len=6)
        loader.append((imgs, caps, lens, [[c] for c in caps]))

    def run(pipeline):
        enc = EncoderCNN(300)
        enc.load_state_dict(_encoder_state(enc))
        dec = DecoderFactoredLSTM(300, 64, 64, V, 1, dropout=0.0)
        dec.load_state_dict(synthetic.decoder_state(dec.state_dict(), seed=4))
        enc.to(dev)
        dec.to(dev)
        opt = Adam(list(dec.parameters()) + list(enc.linear.parameters()) + list(enc.bn.parameters()), lr=1e-3)
        random.seed(77)
        fac = train_factual(enc, dec, opt, CrossEntropyLoss(), loader, 2, 0.5, device=dev, pipeline=pipeline)
        emo = train_emotion(enc, dec, opt, CrossEntropyLoss(), [loader[:2], loader[1:]], ["happy", "sad"], 2,
                            0.5, device=dev, pipeline=pipeline)
        return fac, emo

    fac_p, emo_p = run(True)
    fac_s, emo_s = run(False)
    out = capsys.readouterr().out
    assert "[FAC]" in out and "[HAP]" in out and "[SAD]" in out
    assert abs(fac_p - fac_s) / fac_s < 2e-3
    for a, b in zip(emo_p, emo_s):
        assert abs(a - b) / b < 2e-3


# ---------------------------------------------------------------------------------------------
# BASELINE configs[3]: attention whole step (stylenet/train_multitask_att.py:398-417)
# ---------------------------------------------------------------------------------------------
def _att_modules(V, dev):
    from capnet import model_att
    enc = model_att.EncoderCNN(14)
    est = synthetic.trunk_state(enc.state_dict(), seed=1234)
    enc.load_state_dict(est)
    dec = model_att.DecoderFactoredLSTMAtt(512, 300, 512, 512, V, 1, dropout=0.0)
    p = synthetic.decoder_state(dec.state_dict(), seed=1234)
    dec.load_state_dict(p)
    return enc, est, dec, p


@pytest.mark.parametrize("trunk,steps,tol", [("oracle", 2, 1e-4), ("shared", 4, 2e-5)])
def test_attention_train_steps_match_the_cpu_oracle(dev, trunk, steps, tol):
    """configs[3] per GPU: 12 images, spatial ResNet-152 features [12, 14, 14, 2048], additive
    attention decoder, V = 8192; inputs captions[:, :-1], targets captions[:, 1:], lengths - 1, loss
    = NLL + ((1 - sum_t alpha)^2).mean() (train_multitask_att.py:402-411), Adam lr 2e-4 (:655).
    trunk="oracle": the CPU side runs its own ResNet-152 (EncoderCNNAttRef): the whole step, two
        losses within 1e-4 (the second one checks backward + clamp + Adam of every attention / init /
        f_beta / factored parameter). Not more steps: at 12 images the train-mode BatchNorm chain
        leaves the two fp32 trunks ~1e-4 apart, and the loss, blind to the features at step 0
        (6e-6), follows them more closely with every update (2e-5 at step 1, 2e-4 at step 2; the
        decoder itself is exact: tools/probes/att_grad_check.py, same features -> loss 1e-8, grads 2e-6).
    trunk="shared": the CPU decoder is fed the GPU trunk's features, which isolates
        train_step_att + decoder + optimiser over four steps, held to 2e-5."""
    from oracle.resnet152_ref import EncoderCNNAttRef
    from capnet.train import train_step_att
    V, B, lr, clip = 8192, 12, 2e-4, 0.5
    torch.set_num_threads(16)
    enc, est, dec, p = _att_modules(V, dev)
    imgs, captions, lengths = synthetic.make_batch(B, V, seed=0)
    random.seed(11)
    tfs = [[random.random() < 0.8 for _ in range(max(lengths))] for _ in range(steps)]
    enc.to(dev).train()
    dec.to(dev).train()
    imgs_d, caps_d = imgs.to(dev), captions.to(dev)

    if trunk == "oracle":
        ref_enc = EncoderCNNAttRef(14)
        ref_enc.load_state_dict({k: v.clone() for k, v in est.items()})
        ref_enc.train()
        feats_of = lambda: ref_enc(imgs)
    else:
        shared = enc(imgs_d).cpu()       # train-mode batch statistics: the same features at every step
        feats_of = lambda: shared
    p_ref = {k: v.clone() for k, v in p.items()}
    opt_ref = S.AdamRef(lr=lr)
    lens1 = [l - 1 for l in lengths]
    targets = D.packed_targets(captions[:, 1:], lens1)
    ref_losses = []
    for it in range(steps):
        leaves = {k: v.detach().clone().requires_grad_(True) for k, v in p_ref.items()}
        logits, alphas = D.factored_att_forward(leaves, captions[:, :-1], lens1, feats_of(), tfs[it], "factual")
        loss = D.att_loss(logits, alphas, targets, 1.0)
        loss.backward()
        grads = {k: v.grad for k, v in leaves.items()}
        S.clip_gradient_(grads.values(), clip)
        with torch.no_grad():
            opt_ref.step(p_ref, grads)
        ref_losses.append(float(loss.detach()))

    opt = Adam(list(dec.parameters()), lr=lr)
    got = [float(train_step_att(enc, dec, opt, CrossEntropyLoss(), imgs_d, caps_d, lengths, clip,
                                tf_mask=tfs[it]).item()) for it in range(steps)]
    capnet.ops.check_device_errors()
    print("att", trunk, "oracle", ref_losses, "gpu", got)
    for a, b in zip(got, ref_losses):
        assert abs(a - b) / abs(b) < tol, (got, ref_losses)
    assert ref_losses[-1] < ref_losses[0] - 0.5
    # one updated parameter of each family: it moved, and all but the rounding-level elements moved
    # the same way (an element whose gradient sign is noise ends up to 2 * lr * steps away)
    sd = dec.state_dict()
    for k in ("attention.encoder_att.weight", "attention.full_att.weight", "f_beta.weight", "init_h.weight",
              "init_c.bias", "V_i.weight", "S_fo.weight", "U_c.weight", "W_f.weight", "C.weight"):
        got_k = sd[k].cpu()
        assert (got_k - p[k]).abs().max().item() > 0.5 * lr, k
        off = ((got_k - p_ref[k]).abs() > 0.1 * lr).float().mean().item()
        assert off < 2e-2, (k, off)
        assert (got_k - p_ref[k]).abs().max().item() <= 2.01 * lr * steps, k


def test_pipelined_attention_steps_equal_sequential_steps(dev):
    """TrunkPipeline(attention=True) with the sequential schedule's kernels reproduces
    train_step_att step by step (same losses, same running statistics of the trunk)."""
    from capnet.train import TrunkPipeline, train_step_att
    V, B, steps = 1000, 4, 5
    batches = [synthetic.make_batch(B, V, seed=40 + s) for s in range(steps)]
    random.seed(6)
    tfs = [[random.random() < 0.8 for _ in range(24)] for _ in range(steps)]

    def build():
        enc, _, dec, _ = _att_modules(V, dev)
        enc.to(dev).train()
        dec.to(dev).train()
        return enc, dec, Adam(list(dec.parameters()), lr=2e-4)

    enc, dec, opt = build()
    seq = [float(train_step_att(enc, dec, opt, CrossEntropyLoss(), i.to(dev), c.to(dev), l, 0.5,
                                tf_mask=tf).item()) for (i, c, l), tf in zip(batches, tfs)]
    ref_rm = enc.resnet[7][2].bn3.running_mean.clone()
    ref_w = dec.attention.decoder_att.weight.detach().clone()

    enc, dec, opt = build()
    pipe = TrunkPipeline(enc, dec, opt, CrossEntropyLoss(), 0.5, attention=True, shared_chip_tuning=False)
    dev_batches = [(i.to(dev), c.to(dev), l) for i, c, l in batches]
    for k in range(pipe.depth):
        pipe.prefetch(dev_batches[k][0])
    got = []
    for k, ((imgs, caps, lens), tf) in enumerate(zip(dev_batches, tfs)):
        nxt = dev_batches[k + pipe.depth][0] if k + pipe.depth < steps else None
        got.append(pipe.step(caps, lens, next_images=nxt, tf_mask=tf))
    pipe.finish()
    torch.cuda.synchronize()
    got = [float(l.item()) for l in got]
    print("att sequential", seq, "pipelined", got)
    # the embedding gradient is scattered with float atomics (order varies run to run) and Adam turns
    # a rounding-level gradient into a +-lr update, so later steps agree closely, not bitwise; a step
    # on the wrong features / tokens / parameters would be off by 1e-2
    for a, b in zip(got[:2], seq[:2]):
        assert abs(a - b) / abs(b) < 2e-6
    for a, b in zip(got, seq):
        assert abs(a - b) / abs(b) < 2e-5
    assert torch.equal(enc.resnet[7][2].bn3.running_mean, ref_rm)
    assert (dec.attention.decoder_att.weight - ref_w).abs().max().item() <= 2e-3 * ref_w.abs().max().item()


# ---------------------------------------------------------------------------------------------
# BASELINE configs[4] (1-layer cell, the only one with reference semantics): F = 1024, batch 96,
# alternating factual / emotion steps (stylenet/train_multitask.py:163-167, 373-389, 527-537)
# ---------------------------------------------------------------------------------------------
def test_multitask_alternating_steps_match_the_cpu_oracle(dev):
    """factual step: Adam(lr 2e-4) over decoder + encoder head, both modules zero_grad'ed;
    emotion step: mode='happy', lang Adam(lr 5e-4) over the decoder only, encoder.zero_grad() NOT
    called (train_multitask.py:534 is commented out) so the head's gradients pile up unused; then a
    factual step again, which sees both updates, the two optimisers' separate moments, and S_happy
    skipped for lack of a gradient. Losses of all three steps within 1e-4."""
    V, B, F, clip = 8192, 96, 1024, 0.5
    torch.set_num_threads(16)
    enc = EncoderCNN(300)
    est = _encoder_state(enc)
    enc.load_state_dict(est)
    dec = DecoderFactoredLSTM(300, 512, F, V, 1, dropout=0.0)
    p = synthetic.decoder_state(dec.state_dict(), seed=1234)
    dec.load_state_dict(p)
    schedule = [(None, 0), ("happy", 1), (None, 2)]          # (mode, batch seed)
    batches = [synthetic.make_batch(B, V, seed=s) for _, s in schedule]
    random.seed(13)
    tfs = [[random.random() < 0.8 for _ in range(max(b[2]))] for b in batches]

    ref_enc = EncoderCNNRef(300)
    ref_enc.load_state_dict({k: v.clone() for k, v in est.items()})
    ref_enc.train()
    p_ref = {k: v.clone() for k, v in p.items()}
    opt_ref, lang_ref = S.AdamRef(lr=2e-4), S.AdamRef(lr=5e-4)
    ref_losses = []
    for (mode, _), (imgs, captions, lengths), tf in zip(schedule, batches, tfs):
        leaves = {k: v.detach().clone().requires_grad_(True) for k, v in p_ref.items()}
        feats = ref_enc(imgs)
        logits = D.factored_lstm_forward(leaves, captions, lengths, feats, tf, mode or "factual")
        loss = Fn.cross_entropy(logits, D.packed_targets(captions, lengths))
        if mode is None:
            ref_enc.zero_grad()
        loss.backward()
        grads = {k: v.grad for k, v in leaves.items()}
        S.clip_gradient_(grads.values(), clip)
        with torch.no_grad():
            if mode is None:
                hp = {("enc." + k): v for k, v in ref_enc.named_parameters() if not k.startswith("resnet.")}
                hg = {k: v.grad for k, v in hp.items()}
                S.clip_gradient_([g for g in hg.values() if g is not None], clip)
                both, both_g = dict(p_ref), dict(grads)
                both.update(hp)
                both_g.update(hg)
                opt_ref.step(both, both_g)
            else:
                lang_ref.step(p_ref, grads)
        ref_losses.append(float(loss.detach()))

    enc.to(dev).train()
    dec.to(dev).train()
    opt = Adam(list(dec.parameters()) + list(enc.linear.parameters()) + list(enc.bn.parameters()), lr=2e-4)
    lang = Adam(list(dec.parameters()), lr=5e-4)
    got = []
    for (mode, _), (imgs, captions, lengths), tf in zip(schedule, batches, tfs):
        l = train_step(enc, dec, opt if mode is None else lang, CrossEntropyLoss(), imgs.to(dev),
                       captions.to(dev), lengths, clip, mode=mode, zero_encoder_grad=mode is None, tf_mask=tf)
        got.append(float(l.item()))
        if mode is not None:
            assert enc.linear.weight.grad is not None      # left in place, as in the reference
    capnet.ops.check_device_errors()
    print("multitask oracle", ref_losses, "gpu", got)
    for a, b in zip(got, ref_losses):
        assert abs(a - b) / abs(b) < 1e-4, (got, ref_losses)
    sd = dec.state_dict()
    for k in ("S_happy_i.weight", "S_fi.weight", "W_c.weight", "C.weight"):
        # all but the elements whose gradient is at rounding level moved as on the CPU (those end up
        # to 2 * lr per step away: Adam's first steps are sign-like)
        d = (sd[k].cpu() - p_ref[k]).abs()
        assert (d > 2e-5).float().mean().item() < 2e-2, k
        assert d.max().item() <= 2.01 * (2e-4 * 2 + 5e-4), (k, d.max().item())
    assert (sd["S_happy_i.weight"].cpu() - p["S_happy_i.weight"]).abs().max().item() > 1e-4   # it moved
    assert torch.equal(sd["S_sad_i.weight"].cpu(), p["S_sad_i.weight"])                        # it did not


def test_pipelined_loop_keeps_each_batch_with_its_captions(dev):
    """in the loops the captions are copied to the device on the caller's stream
    and freed on the host side while the pipeline's side stream is still `depth` steps behind. Ten
    distinct batches through _pipelined_loop without an intermediate finish(): every step's loss
    equals the sequential loop's (a step that read another batch's tokens would be off by ~1e-2)."""
    from capnet.train import TrunkPipeline, _pipelined_loop
    B, V, n = 2, 300, 10
    loader = []
    for s in range(n):
        imgs, caps, lens = synthetic.make_batch(B, V, seed=100 + s, min_len=5, max_len=9)
        # batch s draws its words from its own 25-id band; with the output bias ramp below its loss
        # sits ~0.5 away from its neighbours' -- a step fed another batch's tokens cannot hide
        band = (caps >= 4)
        caps = torch.where(band, 4 + 25 * s + (caps - 4) % 25, caps)
        loader.append((imgs, caps, lens, None))
    random.seed(21)
    tfs = [[random.random() < 0.8 for _ in range(9)] for _ in range(n)]

    def build():
        enc = EncoderCNN(300)
        enc.load_state_dict(_encoder_state(enc))
        dec = DecoderFactoredLSTM(300, 64, 64, V, 1, dropout=0.0)
        sd = synthetic.decoder_state(dec.state_dict(), seed=4)
        sd["C.bias"] = -0.02 * torch.arange(V, dtype=torch.float32)
        dec.load_state_dict(sd)
        enc.to(dev).train()
        dec.to(dev).train()
        return enc, dec, Adam(list(dec.parameters()) + list(enc.linear.parameters()) + list(enc.bn.parameters()), lr=1e-4)

    enc, dec, opt = build()
    seq = [float(train_step(enc, dec, opt, CrossEntropyLoss(), i.to(dev), c.to(dev), l, 0.5,
                            tf_mask=tf[:max(l)]).item()) for (i, c, l, _), tf in zip(loader, tfs)]
    enc, dec, opt = build()
    pipe = TrunkPipeline(enc, dec, opt, CrossEntropyLoss(), 0.5, shared_chip_tuning=False)
    losses = [loss for _, loss, _ in _pipelined_loop(
        pipe, loader, dev, lambda i: {"tf_mask": tfs[i][:max(loader[i][2])]})]
    pipe.finish()
    torch.cuda.synchronize()
    got = [float(l.item()) for l in losses]
    print("loop sequential", seq, "pipelined", got)
    assert min(abs(seq[i] - seq[j]) for i in range(n) for j in range(i)) > 0.1     # distinguishable
    for a, b in zip(got, seq):
        assert abs(a - b) / abs(b) < 2e-5


def test_decoder_beside_trunk_passes_is_reproducible(dev):
    """The trainable half runs on its own stream beside the trunk passes of the following batches
    (TrunkPipeline), i.e. its workgroups share CUs with the conv kernels. A decoder forward beside three
    trunk passes must return exactly what it returns alone -- a first split-bf16 conv kernel that
    streamed its weights by LDS-DMA changed single attention scores of co-resident workgroups
    (csrc/conv_bf16x6.hip, tools/victim2.py); this keeps watch over every conv kernel of the trunk."""
    from capnet import model_att
    V, B = 1000, 4
    enc, _, dec, _ = _att_modules(V, dev)
    enc.to(dev).train()
    dec.to(dev).train()
    imgs, caps, lens = synthetic.make_batch(B, V, seed=40)
    feats = enc(imgs.to(dev))
    lens1 = [l - 1 for l in lens]
    random.seed(6)
    tf = [random.random() < 0.8 for _ in range(24)]
    cin = caps[:, :-1].contiguous().to(dev)

    def fwd():
        with torch.no_grad():
            return dec(cin, lens1, feats, tf_mask=tf)
    ref_out, ref_al = fwd()
    torch.cuda.synchronize()
    side = torch.cuda.Stream(priority=-1)
    streams = [torch.cuda.Stream() for _ in range(3)]
    other = [synthetic.make_batch(B, V, seed=50 + k)[0].to(dev) for k in range(3)]
    for rep in range(6):
        for k, st in enumerate(streams):
            st.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(st):
                enc(other[k], slot=k, defer_stats=True)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            out, al = fwd()
        torch.cuda.synchronize()
        assert torch.equal(out, ref_out) and torch.equal(al, ref_al), rep


def _trunk_passes_in_flight(enc, dev, B, n=3, seed=70):
    """-> a function that queues one train-mode trunk pass of batch B on each of n streams (as TrunkPipeline does)."""
    streams = [torch.cuda.Stream() for _ in range(n)]
    imgs = [synthetic.make_batch(B, 100, seed=seed + k)[0].to(dev) for k in range(n)]

    def launch():
        for k, st in enumerate(streams):
            st.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(st):
                enc.trunk_features(imgs[k], slot=k, defer_stats=True, balance_tails=False)   # as TrunkPipeline queues them
    return launch


def test_kernels_we_do_not_build_are_exact_beside_the_trunk(dev):
    """VERDICT r2 / ADVICE r2: this library is built without packed-fp32 VALU because a v_pk_fma_f32 on freshly
    loaded registers returned stale lanes beside the split-f16 conv kernels (records in
    profiles/round3_pk_hazard_probe.txt). torch's element-wise kernels and RCCL's reduction are NOT built with that
    flag, and capnet.parallel overlaps the gradient all-reduce + update with the next trunk passes. Guard: an fp32 sum
    of two freshly written buffers of the flat gradient's size (14.2 M floats = 57 MB), through torch.add, through an
    in-place add_ with a scale (the fused multiply-add form) and through dist.all_reduce in a single-rank RCCL group
    (all RCCL can be asked for on a one-GPU box: it refuses two ranks per device), on a side stream beside three
    full-size trunk passes -- bitwise equal to the same operations alone, 4 rounds."""
    import os
    import torch.distributed as dist
    n = 14_195_588
    g = torch.Generator().manual_seed(9)
    src_a = torch.randn(n, generator=g).to(dev)
    src_b = torch.randn(n, generator=g).to(dev)
    a, b = torch.empty_like(src_a), torch.empty_like(src_b)
    own_group = False
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29653")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
        own_group = True

    def ops_under_test():
        a.copy_(src_a)                       # freshly written operands, as the packed gradient is
        b.copy_(src_b)
        c = torch.add(a, b)
        d = a.clone().add_(b, alpha=0.37)
        r = a.clone()
        dist.all_reduce(r, op=dist.ReduceOp.SUM)
        return c, d, r

    try:
        ref = ops_under_test()
        torch.cuda.synchronize()
        assert torch.equal(ref[2], src_a)
        enc = EncoderCNN(300)
        enc.load_state_dict(_encoder_state(enc))
        enc.to(dev).train()
        launch = _trunk_passes_in_flight(enc, dev, 64)
        launch()
        torch.cuda.synchronize()            # plans, workspaces, packed weights
        side = torch.cuda.Stream(priority=-1)
        for rep in range(4):
            launch()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                got = ops_under_test()
            torch.cuda.synchronize()
            for name, x, y in zip(("torch.add", "add_(alpha)", "all_reduce"), got, ref):
                assert torch.equal(x, y), (rep, name, int((x != y).sum()))
    finally:
        if own_group:
            dist.destroy_process_group()


@pytest.mark.parametrize("B", [64, 96])
def test_persistent_lstm_beside_full_size_trunk_passes(dev, B):
    """the persistent LSTM kernel needs its 256 workgroups co-resident while up to three trunk
    passes' persistent conv workgroups fill the chip. DecoderFactoredLSTM forward at the bench's batch on a side
    stream beside three full-size trunk passes: bit-equal to alone, no device error flag."""
    from capnet import ops as cops
    V = 8192
    dec = DecoderFactoredLSTM(300, 512, 512, V, 1, dropout=0.0)
    dec.load_state_dict(synthetic.decoder_state(dec.state_dict(), seed=1234))
    dec.to(dev).train()
    enc = EncoderCNN(300)
    enc.load_state_dict(_encoder_state(enc))
    enc.to(dev).train()
    _, caps, lens = synthetic.make_batch(B, V, seed=3)
    feats = torch.randn(B, 300, generator=torch.Generator().manual_seed(1)).to(dev)
    cd = caps.to(dev)
    tf = [True] * max(lens)                 # one persistent launch over all steps

    def fwd():
        with torch.no_grad():
            return dec(cd, lens, feats, tf_mask=tf)
    ref = fwd()
    torch.cuda.synchronize()
    cops.check_device_errors()
    launch = _trunk_passes_in_flight(enc, dev, B)
    launch()
    torch.cuda.synchronize()
    side = torch.cuda.Stream(priority=-1)
    for rep in range(4):
        launch()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            out = fwd()
        torch.cuda.synchronize()
        cops.check_device_errors()
        assert torch.equal(out, ref), rep
