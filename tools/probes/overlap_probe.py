"""Diagnostic: do kernels of two HIP streams overlap on MI355X? Stream A: the ResNet-152 trunk;
stream B: a dependent chain of fused LSTM steps (tiny, latency-bound). Prints A alone, B alone,
A and B together."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import capnet
from capnet import synthetic
from capnet._lib import lib, check, current_stream
from capnet.model import EncoderCNN
dev = torch.device("cuda:0")
L = lib()
enc = EncoderCNN(300).to(dev).train()
imgs = synthetic.make_batch(64, 100, seed=0)[0].to(dev)
H, b, n_chain = 512, 64, 2000
hp = torch.randn(b, H, device=dev) * 0.1
wf = torch.empty(L.capnet_lstm_wfrag_floats(H), device=dev)
W = torch.randn(4 * H, H, device=dev) * 0.05
check(L.capnet_lstm_pack_wfrag(W.data_ptr(), wf.data_ptr(), H, 0, current_stream()))
G = torch.randn(b, 4 * H, device=dev); cp = torch.randn(b, H, device=dev)
co, ho = torch.empty_like(cp), torch.empty_like(cp)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream(priority=-1)

def run_a(n=3):
    with torch.cuda.stream(sa):
        for _ in range(n):
            enc.trunk_features(imgs)

graph = torch.cuda.CUDAGraph()
with torch.cuda.stream(sb):
    st = current_stream()
    for _ in range(3):
        check(L.capnet_lstm_step_fused(hp.data_ptr(), wf.data_ptr(), G.data_ptr(), 4 * H, cp.data_ptr(), co.data_ptr(), ho.data_ptr(), b, H, 0, st))
    torch.cuda.synchronize()
    with torch.cuda.graph(graph, stream=sb):
        st = current_stream()
        for _ in range(n_chain):
            check(L.capnet_lstm_step_fused(hp.data_ptr(), wf.data_ptr(), G.data_ptr(), 4 * H, cp.data_ptr(), co.data_ptr(), ho.data_ptr(), b, H, 0, st))

def run_b():
    with torch.cuda.stream(sb):
        graph.replay()

def timed(fn):
    torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3

run_a(1); run_b(); torch.cuda.synchronize()
ta = timed(lambda: run_a(3)); tb = timed(run_b)
tab = timed(lambda: (run_b(), run_a(3)))
print("A alone %.2f ms, B alone %.2f ms (%d steps, %.2f us each), A||B %.2f ms (sum %.2f)" % (ta, tb, n_chain, tb * 1e3 / n_chain, tab, ta + tb))
