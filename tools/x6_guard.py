"""Diagnostic: does conv1x1_fwd_bf16x6 write outside its outputs? (guard bands around y, part_sum, part_sq)"""
import sys, torch
sys.path.insert(0, '/root/repo')
import capnet
from capnet._lib import check, lib, ptr, current_stream
dev = torch.device('cuda:0'); L = lib()
for (Bc, H, Cin, Cout, bn, pre) in ((64, 28, 512, 128, 64, 0), (64, 28, 512, 128, 128, 0), (4, 7, 2048, 512, 64, 0), (64, 56, 64, 256, 64, 1)):
    M = Bc * H * H
    x = torch.randn(Bc, H, H, Cin, device=dev); w = torch.randn(Cout, Cin, device=dev) * 0.05
    sc, sh = torch.rand(Cin, device=dev) + 0.5, torch.randn(Cin, device=dev)
    img = torch.empty(L.capnet_conv1x1_bf16x6_weight_words(Cin, Cout), dtype=torch.int32, device=dev)
    check(L.capnet_conv1x1_bf16x6_pack(ptr(w), ptr(img), Cout, Cin, bn, current_stream()))
    t = L.capnet_conv1x1_tiles_m(M)
    G = 1 << 20
    big = torch.full((3 * G + M * Cout + 2 * t * Cout + 3 * G,), 7.0, device=dev)
    o = G
    y = big[o:o + M * Cout]; o += M * Cout + G
    ps = big[o:o + t * Cout]; o += t * Cout + G
    pq = big[o:o + t * Cout]; o += t * Cout
    y.fill_(float("nan")); ps.fill_(float("nan")); pq.fill_(float("nan"))
    torch.cuda.synchronize()
    check(L.capnet_conv1x1_fwd_bf16x6(ptr(x), H * H * Cin, H * Cin, Cin, ptr(img), bn, ptr(y), ptr(sc) if pre else None,
                                      ptr(sh) if pre else None, pre, ptr(ps), ptr(pq), Bc, H, H, Cin, Cout, 1, None, None, None, 0,
                                      current_stream()))
    torch.cuda.synchronize()
    inside = torch.zeros_like(big, dtype=torch.bool)
    inside[G:G + M * Cout] = True
    a = G + M * Cout + G
    inside[a:a + t * Cout] = True
    a += t * Cout + G
    inside[a:a + t * Cout] = True
    guard_bad = int(((big != 7.0) & ~inside).sum())
    unwritten = int(torch.isnan(big[inside]).sum())
    print((Bc, H, Cin, Cout, bn, pre), "guard elements modified:", guard_bad, "| output elements left unwritten:", unwritten)
