import os, sys, torch
sys.path.insert(0, '/root/repo')
import capnet
from capnet import ops
from capnet._lib import check, lib, ptr
import ctypes as C
dev = torch.device("cuda:0"); L = lib()
B = 4
SH = [(56, 64, 64, 1, 0), (56, 64, 256, 1, 1), (56, 256, 64, 1, 0), (56, 256, 512, 2, 0), (28, 512, 128, 1, 0), (28, 128, 512, 1, 1),
      (14, 1024, 256, 1, 0), (14, 256, 1024, 1, 1), (7, 2048, 512, 1, 0), (7, 512, 2048, 1, 1), (14, 1024, 2048, 2, 0)]
streams = [torch.cuda.Stream() for _ in range(3)]
for (H, Cin, Cout, stride, pre) in SH:
    OH = (H - 1) // stride + 1
    M = B * OH * OH
    x = torch.randn(B, H, H, Cin, device=dev); w = torch.randn(Cout, Cin, device=dev) * 0.05
    sc, sh = (torch.rand(Cin, device=dev) + 0.5, torch.randn(Cin, device=dev)) if pre else (None, None)
    bn = L.capnet_conv1x1_bf16x6_bn(M, Cout)
    img = torch.empty(L.capnet_conv1x1_bf16x6_weight_words(Cin, Cout), dtype=torch.int32, device=dev)
    check(L.capnet_conv1x1_bf16x6_pack(ptr(w), ptr(img), Cout, Cin, bn, C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    t = L.capnet_conv1x1_tiles_m(M)
    torch.cuda.synchronize()
    outs = []
    for rep in range(6):
        for s in streams:
            y = torch.full((M, Cout), float("nan"), device=dev)
            ps, pq = torch.full((t, Cout), float("nan"), device=dev), torch.full((t, Cout), float("nan"), device=dev)
            with torch.cuda.stream(s):
                check(L.capnet_conv1x1_fwd_bf16x6(ptr(x), H * H * Cin, H * Cin, Cin, ptr(img), bn, ptr(y), ptr(sc), ptr(sh), pre,
                                                  ptr(ps), ptr(pq), B, H, H, Cin, Cout, stride, None, None, None, 0, C.c_void_p(s.cuda_stream)))
            outs.append((y, ps, pq))
    torch.cuda.synchronize()
    bad = sum(int(not (torch.equal(o[0], outs[0][0]) and torch.equal(o[1], outs[0][1]) and torch.equal(o[2], outs[0][2]))) for o in outs)
    nan = int(torch.isnan(outs[0][0]).any()) + int(torch.isnan(outs[0][1]).any())
    print((H, Cin, Cout, stride, pre), "bn", bn, "M", M, "mismatching runs:", bad, "nan:", nan)
