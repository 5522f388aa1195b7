"""Inference trunk pass (encoder.eval()) at B = 64: ms per pass.   python tools/probes/eval_trunk_bench.py
CAPNET_EVAL_FOLDED=1: BatchNorms applied in the convolutions' epilogues (the previous inference path)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import capnet
from capnet import synthetic
from capnet.model import EncoderCNN
dev = torch.device("cuda:0")
enc = EncoderCNN(300).to(dev).eval()
imgs = synthetic.make_batch(64, 100, seed=0)[0].to(dev)
with torch.no_grad():
    for _ in range(3): f = enc.trunk_features(imgs)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): f = enc.trunk_features(imgs)
    e1.record(); torch.cuda.synchronize()
print("eval trunk pass: %.2f ms  (folded=%s)  checksum %.6f" % (e0.elapsed_time(e1) / 10, os.environ.get("CAPNET_EVAL_FOLDED", "0"), f[0].float().abs().mean().item() if isinstance(f, tuple) else f.abs().mean().item()))
