"""The stem convolution alone at B = 64, 224 x 224: split-f16 kernel against the generic f32 gather kernel.
    python tools/probes/stem_bench.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import capnet
from capnet import ops
from capnet._lib import check, lib, ptr, current_stream
dev = torch.device("cuda:0"); L = lib()
B, H, W = 64, 224, 224
x = torch.randn(B, 3, H, W, device=dev); w = torch.randn(64, 3, 7, 7, device=dev) * 0.05
OH = OW = 112; M = B * OH * OW
y = torch.empty(M, 64, device=dev)
img = ops.pack_conv_weight_stem_f16x3(w)
rows = L.capnet_conv_stem_f16x3_part_rows(B, H, W)
ps, pq = torch.empty(rows, 64, device=dev), torch.empty(rows, 64, device=dev)
def stem():
    check(L.capnet_conv_stem_fwd_f16x3(ptr(x), 3 * H * W, H * W, W, ptr(img), ptr(y), ptr(ps), ptr(pq), B, H, W, current_stream()))
kw = 160
wp = ops.pack_conv_weight(w, kw, kmajor=False)
tiles = (M + 63) // 64
ps2, pq2 = torch.empty(tiles, 64, device=dev), torch.empty(tiles, 64, device=dev)
y2 = torch.empty(M, 64, device=dev)
def generic():
    check(L.capnet_conv2d_fwd(ptr(x), 3 * H * W, W, 1, H * W, ptr(wp), kw, ptr(y2), None, None, 0, ptr(ps2), ptr(pq2),
                              B, H, W, 3, 64, 7, 7, 2, 3, 0, current_stream()))
for name, f in (("split-f16 stem", stem), ("generic f32", generic)):
    try:
        for _ in range(3): f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): f()
        e1.record(); torch.cuda.synchronize()
        print("%-16s %.1f us" % (name, e0.elapsed_time(e1) * 1e3 / 20))
    except Exception as ex:
        print(name, "failed:", ex)
print("max |diff| between the two:", (y - y2).abs().max().item(), "of", y2.abs().max().item())
