"""Trunk-only throughput with D passes in flight on D streams (what TrunkPipeline does, without any decoder): ms per pass
at B = 64. The full train step cannot be faster than this.   python tools/probes/trunk_depth_probe.py"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import capnet
from capnet import synthetic
from capnet.model import EncoderCNN
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
enc = EncoderCNN(300).to(dev).train()
imgs = synthetic.make_batch(B, 100, seed=0)[0].to(dev)
for D in tuple(int(x) for x in os.environ.get('DEPTHS', '1,2,3,4,3,2').split(',')):
    streams = [torch.cuda.Stream() for _ in range(D)]
    extra = torch.cuda.Stream(priority=-1) if os.environ.get('SIDE') else None       # (the step's high-priority side stream, idle)
    def run(n):
        for i in range(n):
            k = i % D
            with torch.cuda.stream(streams[k]):
                enc.trunk_features(imgs, slot=k, defer_stats=True, balance_tails=(D == 1))
    run(2 * D); torch.cuda.synchronize()
    n = 60
    t0 = time.perf_counter(); run(n); host = time.perf_counter() - t0; torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("depth %d: %.3f ms per pass (%.0f images/s); host enqueue %.3f ms per pass" % (D, dt / n * 1e3, B * n / dt, host / n * 1e3))
