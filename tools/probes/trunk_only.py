"""Time of the ResNet-152 trunk alone (train-mode BN, B=64), for comparison with the pipelined step."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import capnet
from capnet import synthetic
from capnet.model import EncoderCNN
dev = torch.device("cuda:0")
enc = EncoderCNN(300).to(dev)
enc.train() if "--eval" not in sys.argv else enc.eval()
imgs = synthetic.make_batch(64, 100, seed=0)[0].to(dev)
for _ in range(3):
    enc.trunk_features(imgs)
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 20
for _ in range(n):
    enc.trunk_features(imgs)
torch.cuda.synchronize()
print(("eval " if "--eval" in sys.argv else "train ") + "trunk only: %.3f ms per batch of 64" % ((time.perf_counter() - t0) / n * 1e3))
