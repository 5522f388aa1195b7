"""conv3 of the bottlenecks (stride-1 1x1, folded input) at B = 64: A-in-registers kernel (csrc/conv1x1_areg.hip) against the
tiled split-f16 kernel (csrc/conv_f16x3.hip), alone on the chip and beside a stream of stage-3 3x3 convolutions.
    python tools/probes/areg_bench.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import capnet
from capnet import ops
from capnet._lib import check, lib, ptr, current_stream
import ctypes as C
dev = torch.device("cuda:0"); L = lib()
B = 64
# the competing work: stage-3 conv2 on its own stream
cside, cch = 14, 256
cM = B * cside * cside
cx = torch.randn(cM, cch, device=dev); cw = torch.randn(cch, cch, 3, 3, device=dev) * 0.05
cimg = ops.pack_conv_weight_f16x3(cw, 128); cy = torch.empty(cM, cch, device=dev)
csc = torch.rand(cch, device=dev) + 0.5; csh = torch.randn(cch, device=dev)
ct = L.capnet_conv1x1_tiles_m(cM); cps, cpq = torch.empty(ct, cch, device=dev), torch.empty(ct, cch, device=dev)
other = torch.cuda.Stream()
def competitor(n):
    for _ in range(n):
        check(L.capnet_conv3x3_fwd_patch(ptr(cx), ptr(cimg), 128, ptr(cy), ptr(csc), ptr(csh), 1, ptr(cps), ptr(cpq), B, cside, cside, cch, cch, 1,
                                         C.c_void_p(other.cuda_stream)))
for side, cin, cout in [tuple(int(v) for v in t.split("x")) for t in os.environ.get("AREG_SHAPES", "56x64x256,28x128x512,14x256x1024").split(",")]:
    M = B * side * side
    bn = 128
    x = torch.randn(M, cin, device=dev); w = torch.randn(cout, cin, device=dev) * 0.05
    sc = torch.rand(cin, device=dev) + 0.5; sh = torch.randn(cin, device=dev)
    img = torch.empty(L.capnet_conv1x1_f16x3_weight_words(cin, cout), dtype=torch.int32, device=dev)
    check(L.capnet_conv1x1_f16x3_pack(ptr(w), ptr(img), cout, cin, bn, current_stream()))
    tiles = L.capnet_conv1x1_tiles_m(M)
    ps, pq = torch.empty(tiles, cout, device=dev), torch.empty(tiles, cout, device=dev)
    y1, y2 = torch.empty(M, cout, device=dev), torch.empty(M, cout, device=dev)
    def areg():
        check(L.capnet_conv1x1_fwd_areg(ptr(x), ptr(img), bn, ptr(y1), ptr(sc), ptr(sh), 1, ptr(ps), ptr(pq), M, cin, cout, 0, current_stream()))
    def tiled():
        check(L.capnet_conv1x1_fwd_f16x3(ptr(x), side * side * cin, side * cin, cin, ptr(img), bn, ptr(y2), ptr(sc), ptr(sh), 1, ptr(ps), ptr(pq),
                                         B, side, side, cin, cout, 1, None, None, None, 0, current_stream()))
    res = []
    for beside in (False, True):
        for f in (areg, tiled):
            for _ in range(3): f()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            if beside:
                competitor(60)
            e0.record()
            for _ in range(20): f()
            e1.record(); torch.cuda.synchronize()
            res.append(e0.elapsed_time(e1) * 1e3 / 20)
    byt = (M * cin + M * cout) * 4
    print("%2d x %2d  %4d -> %4d: A in registers %.1f us (%.2f TB/s)  tiled %.1f us | beside 3x3 convs: %.1f us  tiled %.1f us | max |diff| %.2e of %.2e" %
          (side, side, cin, cout, res[0], byt / res[0] * 1e-6, res[1], res[2], res[3], (y1 - y2).abs().max().item(), y2.abs().max().item()))
