"""The training loop of the reference on the MI355X kernels.

  train_step     one iteration of the bodies of train_factual / train_emotion
                 (stylenet/train_multitask.py:373-389, 527-537; nic/train_transfer_fac.py:252-296)
  train_factual  stylenet/train_multitask.py:364-408
  train_emotion  stylenet/train_multitask.py:511-557
  val_factual / val_emotion   stylenet/train_multitask.py:272-361, 411-508 (eval-mode trunk on the
                 running BN statistics, free-running decode, top-5 accuracy, BLEU-4, one beam sample)
Same order of operations: targets -> encoder -> decoder -> CrossEntropyLoss -> zero_grad ->
backward -> clip_gradient -> optimizer.step. The loss stays on the device; `.item()` is taken
once per log interval instead of every step (the reference syncs every step, :393,396).
"""
import random

import torch
import torch.nn as nn

from . import ops
from .metrics import corpus_bleu
from .utils import AverageMeter, accuracy, clip_gradient


_scale_cache = {}


def _backward(loss, loss_scale):
    """loss.backward(), or d(loss_scale * loss): the weight goes in as the incoming gradient of the
    loss kernels' backward instead of a torch multiply on the scalar."""
    if loss_scale is None:
        loss.backward()
        return
    key = (loss.device, float(loss_scale))
    g = _scale_cache.get(key)
    if g is None:
        if len(_scale_cache) > 64:
            _scale_cache.clear()
        g = torch.full((), float(loss_scale), dtype=torch.float32, device=loss.device)
        _scale_cache[key] = g
    loss.backward(g)


class CrossEntropyLoss(nn.Module):
    """nn.CrossEntropyLoss() (mean reduction) on the fused HIP softmax+NLL kernels."""

    def forward(self, outputs, targets):
        return ops.cross_entropy(outputs, targets)


def train_step(encoder, decoder, optimizer, criterion, images, captions, lengths, grad_clip,
               mode=None, zero_encoder_grad=True, teacher_forcing_ratio=0.8, tf_mask=None,
               loss_scale=None):
    """One optimisation step; returns the (device) loss tensor of this batch.

    mode=None: factual step (decoder called without `mode`, both modules zero_grad'ed).
    mode='happy'|...: emotion step; the reference leaves encoder.zero_grad() commented out
    there (train_multitask.py:534) -> pass zero_encoder_grad=False to reproduce that.
    loss_scale: data-parallel weight N_rank/N_global applied to the back-propagated loss.
    """
    targets = ops.packed_targets(captions, lengths)
    features = encoder(images)
    kw = {}
    if mode is not None:
        kw["mode"] = mode
    if tf_mask is not None:
        kw["tf_mask"] = tf_mask
    outputs = decoder(captions, lengths, features, teacher_forcing_ratio=teacher_forcing_ratio, **kw)
    loss = criterion(outputs, targets)
    decoder.zero_grad()
    if zero_encoder_grad:
        encoder.zero_grad()
    _backward(loss, loss_scale)
    clip_gradient(optimizer, grad_clip)
    optimizer.step()
    return loss.detach()


def train_step_att(encoder, decoder, optimizer, criterion, images, captions, lengths, grad_clip,
                   mode=None, zero_encoder_grad=True, teacher_forcing_ratio=0.8, tf_mask=None,
                   alpha_c=1.0, loss_scale=None):
    """One step of the attention loops (stylenet/train_multitask_att.py:398-417): inputs
    captions[:, :-1], targets captions[:, 1:], lengths-1, loss += alpha_c*((1-sum_t alpha)^2).mean().
    `lengths` are the loader's lengths (the -1 is applied here, as the reference does at :402)."""
    lengths = [l - 1 for l in lengths]
    targets = ops.packed_targets(captions[:, 1:].contiguous(), lengths)
    features = encoder(images)
    # the attention encoder has no trainable head to hang pre_head_hook on: a data-parallel
    # optimiser's overlapped update (capnet.parallel) must have landed before the decoder reads
    # the parameters
    if hasattr(optimizer, "wait_for_update"):
        optimizer.wait_for_update()
    kw = {}
    if mode is not None:
        kw["mode"] = mode
    if tf_mask is not None:
        kw["tf_mask"] = tf_mask
    outputs, alphas = decoder(captions[:, :-1].contiguous(), lengths, features,
                              teacher_forcing_ratio=teacher_forcing_ratio, **kw)
    loss = criterion(outputs, targets)
    loss = ops.attention_loss(loss, alphas, alpha_c)
    decoder.zero_grad()
    if zero_encoder_grad:
        encoder.zero_grad()
    _backward(loss, loss_scale)
    clip_gradient(optimizer, grad_clip)
    optimizer.step()
    return loss.detach()


class TrunkPipeline(object):
    """Software pipeline over training steps: the frozen trunks of the next `depth` batches run on
    their own streams while head + decoder + loss + backward + clamp + Adam of the current batch
    run on a side stream.

    The ResNet-152 trunk runs under no_grad and none of its inputs depend on the parameter update
    (stylenet/model.py:23-25; train_multitask.py:163-167 only optimises the head and the decoder),
    so this reorders nothing that is ordered in the reference: every step sees exactly the
    parameters and random draws of the sequential loop and produces the same numbers; the BatchNorm
    running statistics of the trunk are updated once per pass, in pass order (the passes defer the
    update and an event chain orders it, capnet_trunk_update_running). What it changes:
      * the decoder's many small, latency-bound launches share the chip with the convolutions of
        the following batches instead of leaving it idle;
      * `depth` (default 3) trunk passes are in flight: while one sits in a launch-bound bubble
        (bn_finalize, tail fix-ups: tiny kernels between the convolutions) the convolutions of
        another one run (tools/probes/trunk_overlap_probe.py: 17.0 -> 15.2-16.0 ms per pass with two;
        measured step time 18.9 / 16.1 / 15.7 / 16.8 ms at depth 1 / 2 / 3 / 4).

        pipe = TrunkPipeline(encoder, decoder, optimizer, criterion, grad_clip)
        for k in range(pipe.depth): pipe.prefetch(images_k)      # up to `depth` batches ahead
        for i in range(n):
            loss = pipe.step(captions_i, lengths_i, next_images=images_{i+depth} or None)
        pipe.finish()            # before reading a loss or the parameters on the caller's stream
    """

    def __init__(self, encoder, decoder, optimizer, criterion, grad_clip, attention=False,
                 alpha_c=1.0, depth=3, shared_chip_tuning=True, graph_trunk=False):
        self.encoder, self.decoder, self.optimizer = encoder, decoder, optimizer
        self.criterion, self.grad_clip = criterion, grad_clip
        self.attention, self.alpha_c = attention, alpha_c
        self.depth = max(1, int(depth))
        # shared_chip_tuning: with several passes in flight the trunk drops the K-sliced tail
        # balancing and prefers the 128x64 conv tile on the large layers (csrc/trunk.cpp; +4 %
        # images/s). Both only change the ORDER of fp32 additions (BatchNorm partial sums, K slices),
        # so results differ from the sequential loop at rounding level; False keeps the sequential
        # schedule's kernels and reproduces its numbers exactly.
        self.balance_tails = not (shared_chip_tuning and self.depth >= 2)
        # graph_trunk: replay each trunk pass (~330 launches, fixed arguments) from a hipGraph
        # captured per slot. It removes ~3.5 ms of launch calls per pass from the host, but the step
        # is bound by the GPU (the host runs ahead until the queues are full), so it measured no
        # gain on one GPU: 4370 vs 4355 images/s; off by default.
        self.graph_trunk = bool(graph_trunk)
        # The trainable half gets the HIGH-priority stream: its launches are small (a few
        # workgroups, microseconds) and form a long dependent chain, so they must be dispatched as
        # soon as they are ready; a convolution of the trunk has thousands of workgroups queued and
        # loses nothing by yielding a few slots.
        self.side = torch.cuda.Stream(priority=-1)
        self.trunk_streams = [torch.cuda.Stream() for _ in range(self.depth)]
        self._queue = []          # prefetched batches, oldest first: (features, ready event)
        self._issued = 0
        self._stats_done = None   # event after the previous pass's running-statistics update

    def prefetch(self, images):
        """Enqueue the trunk of `images` (ordered after the caller's current stream)."""
        if len(self._queue) >= self.depth:
            raise RuntimeError("TrunkPipeline: %d batches already in flight" % len(self._queue))
        slot = self._issued % self.depth
        self._issued += 1
        stream = self.trunk_streams[slot]
        stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(stream):
            if self.attention:
                feats, apply_stats = self.encoder(images, slot=slot, defer_stats=True,
                                                  balance_tails=self.balance_tails, graph=self.graph_trunk)
            else:
                feats, apply_stats = self.encoder.trunk_features(images, slot=slot, defer_stats=True,
                                                                 balance_tails=self.balance_tails, graph=self.graph_trunk)
            ready = torch.cuda.Event()
            ready.record()
            if apply_stats is not None:
                # running statistics: after the previous pass's update, whatever stream it ran on
                if self._stats_done is not None:
                    stream.wait_event(self._stats_done)
                apply_stats()
                self._stats_done = torch.cuda.Event()
                self._stats_done.record()
        images.record_stream(stream)
        feats.record_stream(self.side)
        self._queue.append((feats, ready))

    def step(self, captions, lengths, next_images=None, mode=None, tf_mask=None, loss_scale=None,
             teacher_forcing_ratio=0.8, zero_encoder_grad=True):
        if not self._queue:
            raise RuntimeError("TrunkPipeline.step() without a prefetched batch")
        feats, ready = self._queue.pop(0)
        if next_images is not None:
            self.prefetch(next_images)      # queued before the decoder work of this batch
        enc, dec = self.encoder, self.decoder
        kw = {}
        if mode is not None:
            kw["mode"] = mode
        if tf_mask is not None:
            kw["tf_mask"] = tf_mask
        # `captions` was produced on the caller's stream (an H2D copy in the loops): order the side
        # stream after it, and keep its memory from being handed out again while the side stream
        # (up to `depth` steps behind the host) still reads it
        self.side.wait_stream(torch.cuda.current_stream())
        captions.record_stream(self.side)
        with torch.cuda.stream(self.side):
            self.side.wait_event(ready)
            if self.attention:
                if hasattr(self.optimizer, "wait_for_update"):
                    self.optimizer.wait_for_update()
                lens = [l - 1 for l in lengths]
                targets = ops.packed_targets(captions[:, 1:].contiguous(), lens)
                outputs, alphas = dec(captions[:, :-1].contiguous(), lens, feats,
                                      teacher_forcing_ratio=teacher_forcing_ratio, **kw)
                loss = self.criterion(outputs, targets)
                loss = ops.attention_loss(loss, alphas, self.alpha_c)
            else:
                targets = ops.packed_targets(captions, lengths)
                if enc.pre_head_hook is not None:
                    enc.pre_head_hook()
                features = enc.bn(enc.linear(feats))
                outputs = dec(captions, lengths, features,
                              teacher_forcing_ratio=teacher_forcing_ratio, **kw)
                loss = self.criterion(outputs, targets)
            dec.zero_grad()
            if zero_encoder_grad:
                enc.zero_grad()
            _backward(loss, loss_scale)
            clip_gradient(self.optimizer, self.grad_clip)
            self.optimizer.step()
            loss = loss.detach()
        return loss

    def in_flight(self):
        return len(self._queue)

    def finish(self):
        """Make the caller's stream wait for everything queued by the pipeline."""
        cur = torch.cuda.current_stream()
        cur.wait_stream(self.side)
        for s in self.trunk_streams:
            cur.wait_stream(s)


def _drain(pending, meter):
    for loss, n in pending:
        meter.update(loss.item(), n)
    del pending[:]


def _pipelined_loop(pipe, loader, device, step_kwargs):
    """Yields (i, loss, lengths): keeps `pipe.depth` batches prefetched ahead of the step.
    step_kwargs: dict, or callable(step index) -> dict (e.g. an explicit tf_mask per step)."""
    it = iter(loader)
    meta = []

    def feed():
        try:
            images, captions, lengths, _ = next(it)
        except StopIteration:
            return False
        captions = captions.to(device, non_blocking=True)      # before prefetch's wait_stream
        pipe.prefetch(images.to(device, non_blocking=True))
        meta.append((captions, lengths))
        return True

    for _ in range(pipe.depth):
        if not feed():
            break
    i = 0
    while meta:
        captions, lengths = meta.pop(0)
        loss = pipe.step(captions, lengths, **(step_kwargs(i) if callable(step_kwargs) else step_kwargs))
        feed()
        yield i, loss, lengths
        i += 1


def train_factual(encoder, decoder, optimizer, criterion, data_loader, log_step, grad_clip,
                  device=None, pipeline=True):
    """stylenet/train_multitask.py:364-408. pipeline=True overlaps each batch's trainable half with
    the trunks of the following batches (TrunkPipeline); the numbers are those of the sequential
    loop."""
    decoder.train()
    encoder.train()
    losses = AverageMeter()
    pending = []
    device = device or next(decoder.parameters()).device
    if pipeline:
        pipe = TrunkPipeline(encoder, decoder, optimizer, criterion, grad_clip)
        steps = _pipelined_loop(pipe, data_loader, device, {})
    else:
        pipe = None
        steps = ((i, train_step(encoder, decoder, optimizer, criterion,
                                images.to(device, non_blocking=True),
                                captions.to(device, non_blocking=True), lengths, grad_clip), lengths)
                 for i, (images, captions, lengths, _) in enumerate(data_loader))
    for i, loss, lengths in steps:
        pending.append((loss, sum(lengths)))
        if i % log_step == 0:
            if pipe is not None:
                pipe.finish()
            _drain(pending, losses)
            ops.check_device_errors(recover_lstm_timeout=True)
            print("""Step [{}/{}], [FAC], Loss: {:.4f}""".format(i, len(data_loader), losses.val))
    if pipe is not None:
        pipe.finish()
    _drain(pending, losses)
    ops.check_device_errors(recover_lstm_timeout=True)
    return losses.avg


def train_emotion(encoder, decoder, optimizer, criterion, data_loaders, tags, log_step,
                  grad_clip, device=None, pipeline=True):
    """stylenet/train_multitask.py:511-557 (emotion order drawn with random.sample, as there)."""
    decoder.train()
    encoder.train()
    losses = [AverageMeter() for _ in range(len(tags))]
    device = device or next(decoder.parameters()).device
    pipe = TrunkPipeline(encoder, decoder, optimizer, criterion, grad_clip) if pipeline else None
    for j in random.sample([i for i in range(len(tags))], len(tags)):
        pending = []
        if pipe is not None:
            steps = _pipelined_loop(pipe, data_loaders[j], device,
                                    {"mode": tags[j], "zero_encoder_grad": False})
        else:
            steps = ((i, train_step(encoder, decoder, optimizer, criterion,
                                    images.to(device, non_blocking=True),
                                    captions.to(device, non_blocking=True), lengths, grad_clip,
                                    mode=tags[j], zero_encoder_grad=False), lengths)
                     for i, (images, captions, lengths, _) in enumerate(data_loaders[j]))
        for i, loss, lengths in steps:
            pending.append((loss, sum(lengths)))
            if i % log_step == 0:
                if pipe is not None:
                    pipe.finish()
                _drain(pending, losses[j])
                ops.check_device_errors(recover_lstm_timeout=True)
                print("""Step [{}/{}], [{}], Loss: {:.4f}""".format(
                    i, len(data_loaders[j]), tags[j][:3].upper(), losses[j].val))
        if pipe is not None:
            pipe.finish()
        _drain(pending, losses[j])
    ops.check_device_errors(recover_lstm_timeout=True)
    return [l.avg for l in losses]


def _unpack_predictions(outputs, batch_sizes, lengths):
    """Per-sample argmax word ids from packed logits: the `pad_packed_sequence` + `torch.max(s, 1)`
    + `[:l]` of stylenet/train_multitask.py:311-329 (one argmax kernel over all N rows, then host
    index arithmetic)."""
    pred = ops.argmax_rows(outputs).tolist()
    off = [0]
    for b in batch_sizes:
        off.append(off[-1] + b)
    out = []
    for i, l in enumerate(lengths):
        out.append([pred[off[t] + i] for t in range(l)])
    return out


def _validate(encoder, decoder, vocab, criterion, data_loader, mode, device):
    decoder.eval()
    encoder.eval()
    import time
    from .utils import AverageMeter as _AM
    batch_time, losses, top5accs = _AM(), _AM(), _AM()
    t0 = time.time()
    references, hypotheses = [], []
    device = device or next(decoder.parameters()).device
    start, end = vocab.word2idx['<start>'], vocab.word2idx['<end>']
    features = None
    kw = {} if mode is None else {"mode": mode}
    for i, (images, captions, lengths, all_captions) in enumerate(data_loader):
        images = images.to(device)
        captions = captions.to(device)
        targets = ops.packed_targets(captions, lengths)
        with torch.no_grad():
            features = encoder(images)
            outputs = decoder(captions, lengths, features, teacher_forcing_ratio=0, **kw)
            loss = criterion(outputs, targets)
        losses.update(loss.item(), sum(lengths))
        top5accs.update(accuracy(outputs, targets, 5), sum(lengths))
        batch_time.update(time.time() - t0)
        for caps in all_captions:
            caps = [[int(w) for w in (c.tolist() if hasattr(c, "tolist") else c)] for c in caps]
            references.append([[w for w in c if w != start and w != end] for c in caps])
        for pred in _unpack_predictions(outputs, ops.batch_sizes_from_lengths(lengths), lengths):
            hypotheses.append([w for w in pred if w != start and w != end])
        assert len(references) == len(hypotheses)
    ops.check_device_errors()
    bleu4 = corpus_bleu(references, hypotheses)
    feature = features[0].unsqueeze(0)
    sampled_ids = decoder.sample(feature, start_token=start, end_token=end, **kw)[0].tolist()
    sampled_caption = []
    for word_id in sampled_ids:
        word = vocab.idx2word[word_id]
        sampled_caption.append(word)
        if word == '<end>':
            break
    print(sampled_caption)
    return batch_time.val, top5accs.avg, losses.avg, bleu4


def val_factual(encoder, decoder, vocab, criterion, data_loader, device=None):
    """stylenet/train_multitask.py:272-361. Returns (batch_time, top-5 accuracy %, loss, BLEU-4)."""
    from .nic_model import DecoderRNN
    return _validate(encoder, decoder, vocab, criterion, data_loader,
                     None if isinstance(decoder, DecoderRNN) else "factual", device)


def val_emotion(encoder, decoder, vocab, criterion, data_loaders, tags, device=None):
    """stylenet/train_multitask.py:411-508: one validation pass per emotion loader, decoded in that
    emotion's mode. Returns (batch_time, [top5 per tag], [loss per tag], [bleu4 per tag])."""
    res = [_validate(encoder, decoder, vocab, criterion, data_loaders[j], tags[j], device)
           for j in range(len(tags))]
    return (res[-1][0] if res else 0.0, [r[1] for r in res], [r[2] for r in res],
            [r[3] for r in res])


def evaluate(encoder, decoder, vocab, data_loader, mode='factual', k=5, device=None, verbose=False):
    """The test-set evaluator, stylenet/evaluator.py:55-120: encoder + decoder in eval mode, every test image decoded by
    beam search, corpus BLEU-1..4 with the reference's four weight tuples (references and hypotheses as the reference
    builds them: the captions' and the sampled ids as they are, <start> / <end> included). The reference calls
    decoder.sample() per image; here a loader batch is decoded at once (decoder.sample_batch: B x k rows per step) --
    the same sequences (tests/test_sample_gpu.py). Returns (bleu_1, bleu_2, bleu_3, bleu_4)."""
    decoder.eval()
    encoder.eval()
    device = device or next(decoder.parameters()).device
    start, end = vocab.word2idx['<start>'], vocab.word2idx['<end>']
    kw = {} if mode is None else {"mode": mode}
    references, hypotheses = [], []
    for images, captions, lengths, all_captions in data_loader:
        with torch.no_grad():
            features = encoder(images.to(device))
        if hasattr(decoder, "sample_batch"):
            seqs = decoder.sample_batch(features, start_token=start, end_token=end, k=k, **kw)
        else:       # (nic DecoderRNNAtt: image by image, as the reference does)
            seqs = [decoder.sample(features[i:i + 1], start_token=start, end_token=end, k=k, **kw)[0].tolist()
                    for i in range(features.size(0))]
        for sampled_ids, caps in zip(seqs, all_captions):
            caps = [[int(w) for w in (c.tolist() if hasattr(c, "tolist") else c)] for c in caps]
            references.append(caps)
            hypotheses.append([int(w) for w in sampled_ids])
            if verbose:
                for name, ids in (("ref", caps[0]), ("pred", sampled_ids)):
                    words = []
                    for word_id in ids:
                        words.append(vocab.idx2word[word_id])
                        if words[-1] == '<end>':
                            break
                    print(name, ' '.join(words))
    assert len(references) == len(hypotheses)
    ops.check_device_errors()
    out = tuple(corpus_bleu(references, hypotheses, weights=w)
                for w in ((1, 0, 0, 0), (0.5, 0.5, 0, 0), (0.33, 0.33, 0.33, 0), (0.25, 0.25, 0.25, 0.25)))
    if verbose:
        for i, b in enumerate(out):
            print('BLEU-%d' % (i + 1), b)
    return out
