"""bf16x6 1x1 conv kernel against the K-major f32 kernel on the trunk's shapes (B = 64)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import capnet
from capnet import ops
from capnet._lib import check, current_stream, lib, ptr
dev = torch.device("cuda:0"); L = lib()
KIND = os.environ.get("X6_KIND", "f16x3")      # bf16x6 | f16x3
def timed(fn, iters=30):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters
B = 64
SH = {"s1c1": (56, 256, 64, 1, 0), "s1c3": (56, 64, 256, 1, 1), "s2c1": (28, 512, 128, 1, 0), "s2c3": (28, 128, 512, 1, 1),
      "s3c1": (14, 1024, 256, 1, 0), "s3c3": (14, 256, 1024, 1, 1), "s4c1": (7, 2048, 512, 1, 0), "s4c3": (7, 512, 2048, 1, 1),
      "s2ds": (56, 256, 512, 2, 0), "s3ds": (28, 512, 1024, 2, 0), "s4ds": (14, 1024, 2048, 2, 0)}
ONLY = os.environ.get("X6_ONLY")
bns = [int(v) for v in sys.argv[1].split(",")] if len(sys.argv) > 1 else [64, 128]
tot = {}
for name, (H, Cin, Cout, stride, pre) in SH.items():
    if ONLY and name not in ONLY.split(","): continue
    OH = (H - 1) // stride + 1
    M = B * OH * OH
    x = torch.randn(B, H, H, Cin, device=dev); w = torch.randn(Cout, Cin, device=dev) * 0.05
    y = torch.empty(M, Cout, device=dev)
    sc, sh = (torch.rand(Cin, device=dev) + 0.5, torch.randn(Cin, device=dev)) if pre else (None, None)
    t = L.capnet_conv1x1_tiles_m(M)
    ps, pq = torch.empty(2 * t, Cout, device=dev), torch.empty(2 * t, Cout, device=dev)
    fl = 2.0 * M * Cout * Cin
    line = "%s M=%6d N=%4d K=%4d s%d pre%d:" % (name, M, Cout, Cin, stride, pre)
    for bn in bns:
        if Cout % bn: continue
        img = torch.empty(getattr(L, 'capnet_conv1x1_%s_weight_words' % KIND)(Cin, Cout), dtype=torch.int32, device=dev)
        check(getattr(L, 'capnet_conv1x1_%s_pack' % KIND)(ptr(w), ptr(img), Cout, Cin, bn, current_stream()))
        us = timed(lambda: check(getattr(L, 'capnet_conv1x1_fwd_%s' % KIND)(ptr(x), H * H * Cin, H * Cin, Cin, ptr(img), bn, ptr(y), ptr(sc), ptr(sh), pre,
                                                              ptr(ps), ptr(pq), B, H, H, Cin, Cout, stride, None, None, None, 0, current_stream())))
        tot[bn] = tot.get(bn, 0) + us
        line += " x6/%d %6.1f us %6.1f TF/s |" % (bn, us, fl / us / 1e6)
    wk = ops.pack_conv_weight(w.reshape(Cout, Cin, 1, 1), Cin, kmajor=True)
    tiles = L.capnet_conv_kmajor_tiles_m(M, Cout, Cin, 12864)
    ps2, pq2 = torch.empty(tiles, Cout, device=dev), torch.empty(tiles, Cout, device=dev)
    us_old = timed(lambda: check(L.capnet_conv2d_fwd_kmajor(ptr(x), H * H * Cin, H * Cin, Cin, ptr(wk), Cin, ptr(y), ptr(sc), ptr(sh), pre,
                                                            ptr(ps2), ptr(pq2), B, H, H, Cin, Cout, 1, 1, stride, 0, 12864, None, current_stream())))
    tot["f32"] = tot.get("f32", 0) + us_old
    tot["fl"] = tot.get("fl", 0) + fl
    print(line + " f32 k-major %6.1f us %6.1f TF/s" % (us_old, fl / us_old / 1e6))
print("sum:", {k: (round(v, 1), round(tot["fl"] / v / 1e6, 1)) for k, v in tot.items() if k != "fl"})
