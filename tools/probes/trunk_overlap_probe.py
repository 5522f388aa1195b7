"""Diagnostic: two independent ResNet-152 trunk passes (train-mode BN, B=64) on two streams vs back to back."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import capnet
from capnet import synthetic
from capnet.model import EncoderCNN
dev = torch.device("cuda:0")
encs = [EncoderCNN(300).to(dev).train() for _ in range(2)]
imgs = synthetic.make_batch(64, 100, seed=0)[0].to(dev)
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
for e in encs:
    e.trunk_features(imgs)
torch.cuda.synchronize()

def seq(n):
    for _ in range(n):
        for e in encs:
            e.trunk_features(imgs)

def par(n):
    for _ in range(n):
        for e, s in zip(encs, streams):
            with torch.cuda.stream(s):
                e.trunk_features(imgs)

for name, fn in (("sequential", seq), ("two streams", par), ("sequential", seq), ("two streams", par)):
    torch.cuda.synchronize(); t0 = time.perf_counter(); fn(5); torch.cuda.synchronize()
    print("%s: %.3f ms per trunk pass" % (name, (time.perf_counter() - t0) / 10 * 1e3))
