"""One trunk layer launched N times, for rocprofv3 --pmc passes (tools/pmc_sq2.sh).
    python tools/layer_one.py x6 s3c3 128 [iters]      bf16x6 1x1 conv
    python tools/layer_one.py wino s3c2 0 [iters]      Winograd 3x3 conv
    python tools/layer_one.py h33 s3c2 0 [iters]       3x3 conv on the split-f16 kernel
    python tools/layer_one.py p33 s3c2 0 [iters]       ... with the input patch in LDS (P3_ALONE=1: the K-split arrangement)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import capnet
from capnet import ops
from capnet._lib import check, current_stream, lib, ptr
dev = torch.device("cuda:0"); L = lib(); B = 64
KIND = os.environ.get("X6_KIND", "f16x3")      # bf16x6 | f16x3; B = 64
X6 = {"s1c1": (56, 256, 64, 1, 0), "s1c3": (56, 64, 256, 1, 1), "s2c1": (28, 512, 128, 1, 0), "s2c3": (28, 128, 512, 1, 1),
      "s3c1": (14, 1024, 256, 1, 0), "s3c3": (14, 256, 1024, 1, 1), "s4c1": (7, 2048, 512, 1, 0), "s4c3": (7, 512, 2048, 1, 1)}
WI = {"s1c2": (56, 64), "s2c2": (28, 128), "s3c2": (14, 256), "s4c2": (7, 512)}
kind, name, bn = sys.argv[1], sys.argv[2], int(sys.argv[3]); iters = int(sys.argv[4]) if len(sys.argv) > 4 else 5
if kind == "x6":
    H, Cin, Cout, stride, pre = X6[name]
    M = B * H * H
    x = torch.randn(B, H, H, Cin, device=dev); w = torch.randn(Cout, Cin, device=dev) * 0.05
    y = torch.empty(M, Cout, device=dev)
    sc, sh = (torch.rand(Cin, device=dev) + 0.5, torch.randn(Cin, device=dev)) if pre else (None, None)
    t = L.capnet_conv1x1_tiles_m(M)
    ps, pq = torch.empty(2 * t, Cout, device=dev), torch.empty(2 * t, Cout, device=dev)
    img = torch.empty(getattr(L, 'capnet_conv1x1_%s_weight_words' % KIND)(Cin, Cout), dtype=torch.int32, device=dev)
    check(getattr(L, 'capnet_conv1x1_%s_pack' % KIND)(ptr(w), ptr(img), Cout, Cin, bn, current_stream()))
    run = lambda: check(getattr(L, 'capnet_conv1x1_fwd_%s' % KIND)(ptr(x), H * H * Cin, H * Cin, Cin, ptr(img), bn, ptr(y), ptr(sc), ptr(sh), pre,
                                                     ptr(ps), ptr(pq), B, H, H, Cin, Cout, stride, None, None, None, 0, current_stream()))
elif kind == "h33":
    H, C = WI[name]
    M = B * H * H
    x = torch.randn(B, H, H, C, device=dev); w = torch.randn(C, C, 3, 3, device=dev) * 0.05
    y = torch.empty(M, C, device=dev)
    sc, sh = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev)
    bnw = 128 if C % 128 == 0 else 64
    img = ops.pack_conv_weight_f16x3(w, bnw)
    t = L.capnet_conv1x1_tiles_m(M)
    ps, pq = torch.empty(t, C, device=dev), torch.empty(t, C, device=dev)
    run = lambda: check(L.capnet_conv2d_fwd_f16x3(ptr(x), H * H * C, H * C, C, ptr(img), bnw, ptr(y), ptr(sc), ptr(sh), 1, ptr(ps), ptr(pq),
                                                   B, H, H, C, C, 3, 1, 1, None, None, None, 0, current_stream()))
elif kind == "p33":
    H, C = WI[name]
    M = B * H * H
    x = torch.randn(B, H, H, C, device=dev); w = torch.randn(C, C, 3, 3, device=dev) * 0.05
    y = torch.empty(M, C, device=dev)
    sc, sh = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev)
    bnw = 128 if C % 128 == 0 else 64
    img = ops.pack_conv_weight_f16x3(w, bnw)
    t = L.capnet_conv1x1_tiles_m(M)
    ps, pq = torch.empty(t, C, device=dev), torch.empty(t, C, device=dev)
    run = lambda: check(L.capnet_conv3x3_fwd_patch(ptr(x), ptr(img), bnw, ptr(y), ptr(sc), ptr(sh), 1, ptr(ps), ptr(pq), B, H, H, C, C, int(os.environ.get("P3_SHARED", "0")), current_stream()))
else:
    H, C = WI[name]
    x = torch.randn(B, H, H, C, device=dev); w = torch.randn(C, C, 3, 3, device=dev) * 0.05
    y = torch.empty(B * H * H, C, device=dev)
    sc, sh = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev)
    ww = ops.pack_conv_weight_wino(w); wt = L.capnet_conv_wino_tiles_m(B, H, H)
    ps, pq = torch.empty(wt, C, device=dev), torch.empty(wt, C, device=dev)
    run = lambda: check(L.capnet_conv2d_fwd_wino(ptr(x), H * H * C, H * C, C, ptr(ww), ptr(y), ptr(sc), ptr(sh), 1, ptr(ps), ptr(pq),
                                                  B, H, H, C, C, None, None, 0, current_stream()))
for _ in range(iters): run()
torch.cuda.synchronize()
print("done", kind, name)
