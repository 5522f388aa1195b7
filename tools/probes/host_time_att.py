"""Diagnostic: host-side time of one attention train step, by phase (no GPU sync inside)."""
import os, sys, time, random, cProfile, pstats
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import capnet
from capnet import synthetic, model_att
from capnet.optim import Adam
from capnet.train import CrossEntropyLoss, train_step_att
dev = torch.device("cuda:0")
B, V = 64, 8192
enc = model_att.EncoderCNN(14).to(dev).train()
dec = model_att.DecoderFactoredLSTMAtt(512, 300, 512, 512, V, 1, dropout=0.5)
dec.load_state_dict(synthetic.decoder_state(dec.state_dict(), seed=1234))
dec.to(dev).train()
opt = Adam(list(dec.parameters()), lr=2e-4)
crit = CrossEntropyLoss()
imgs, caps, lens = synthetic.make_batch(B, V, seed=0)
imgs, caps = imgs.to(dev), caps.to(dev)
random.seed(0)
for _ in range(3):
    train_step_att(enc, dec, opt, crit, imgs, caps, lens, 0.5)
torch.cuda.synchronize()
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
for _ in range(5):
    train_step_att(enc, dec, opt, crit, imgs, caps, lens, 0.5)
pr.disable()
t1 = time.perf_counter()
torch.cuda.synchronize()
print("host time per step %.2f ms (gpu finished %.2f ms later)" % ((t1 - t0) / 5 * 1e3, (time.perf_counter() - t1) * 1e3))
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
