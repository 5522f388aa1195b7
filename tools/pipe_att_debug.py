"""Diagnostic: TrunkPipeline(attention=True): are the features the decoder gets the sequential ones?"""
import random, sys, torch
sys.path.insert(0, '/root/repo')
import capnet
from capnet import synthetic, model_att
from capnet.optim import Adam
from capnet.train import CrossEntropyLoss, TrunkPipeline, train_step_att
dev = torch.device('cuda:0')
V, B, steps = 1000, 4, 5
batches = [synthetic.make_batch(B, V, seed=40 + s) for s in range(steps)]
random.seed(6)
tfs = [[random.random() < 0.8 for _ in range(24)] for _ in range(steps)]
def build():
    enc = model_att.EncoderCNN(14)
    enc.load_state_dict(synthetic.trunk_state(enc.state_dict(), seed=1234))
    dec = model_att.DecoderFactoredLSTMAtt(512, 300, 512, 512, V, 1, dropout=0.0)
    dec.load_state_dict(synthetic.decoder_state(dec.state_dict(), seed=1234))
    enc.to(dev).train(); dec.to(dev).train()
    return enc, dec, Adam(list(dec.parameters()), lr=2e-4)
enc, dec, opt = build()
ref_feats = [enc(i.to(dev)).clone() for i, _, _ in batches]
enc, dec, opt = build()
seq = [float(train_step_att(enc, dec, opt, CrossEntropyLoss(), i.to(dev), c.to(dev), l, 0.5, tf_mask=tf).item())
       for (i, c, l), tf in zip(batches, tfs)]
for trial in range(3):
    enc, dec, opt = build()
    pipe = TrunkPipeline(enc, dec, opt, CrossEntropyLoss(), 0.5, attention=True, shared_chip_tuning=False)
    devb = [(i.to(dev), c.to(dev), l) for i, c, l in batches]
    for k in range(pipe.depth):
        pipe.prefetch(devb[k][0])
    got, snaps = [], []
    for k, ((imgs, caps, lens), tf) in enumerate(zip(devb, tfs)):
        f, ev = pipe._queue[0]
        torch.cuda.current_stream().wait_event(ev)
        snaps.append(f.clone())
        nxt = devb[k + pipe.depth][0] if k + pipe.depth < steps else None
        got.append(pipe.step(caps, lens, next_images=nxt, tf_mask=tf))
    pipe.finish(); torch.cuda.synchronize()
    got = [float(l.item()) for l in got]
    print("trial", trial, "feats equal:", [bool(torch.equal(s, r)) for s, r in zip(snaps, ref_feats)],
          "max diff", [float((s - r).abs().max()) for s, r in zip(snaps, ref_feats)])
    print("   losses pipelined", got)
print("   losses sequential", seq)
