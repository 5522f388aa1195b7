"""Diagnostic: per-column-tile error of capnet_sgemm_splitk."""
import sys, torch
sys.path.insert(0, '/root/repo')
import capnet
from capnet._lib import lib, check, ptr, current_stream
dev = torch.device('cuda:0')
for (M, N, K, tb) in [(64, 2348, 2048, False), (64, 2348, 2048, True), (64, 2304, 2048, False), (64, 576, 128, False)]:
    g = torch.Generator().manual_seed(1)
    A = torch.randn(M, K, generator=g)
    B = torch.randn(N, K, generator=g) if tb else torch.randn(K, N, generator=g)
    ref = A.double() @ (B.double().t() if tb else B.double())
    Ad, Bd = A.to(dev), B.to(dev)
    Cd = torch.zeros(M, N, device=dev)
    ws = torch.empty(32 * 64 * 4608, device=dev)
    check(lib().capnet_sgemm_splitk(0, int(tb), M, N, K, ptr(Ad), K, ptr(Bd), Bd.shape[1], ptr(Cd), N, None, 0, ptr(ws), ws.numel(), current_stream()))
    e = (Cd.cpu().double() - ref).abs()
    cols = e.max(0).values
    bad = (cols > 1e-3).nonzero().flatten().tolist()
    print(M, N, K, tb, "max err", e.max().item(), "bad cols", bad[:10], len(bad))
print("generic kernel (no workspace):")
for (M, N, K, tb, acc) in [(64, 2348, 2048, False, 0), (64, 2348, 2048, False, 1), (64, 2304, 2048, False, 1), (64, 2348, 512, False, 1), (64, 2348, 2048, True, 1)]:
    g = torch.Generator().manual_seed(1)
    A = torch.randn(M, K, generator=g)
    B = torch.randn(N, K, generator=g) if tb else torch.randn(K, N, generator=g)
    bias = torch.randn(N, generator=g); C0 = torch.randn(M, N, generator=g)
    ref = A.double() @ (B.double().t() if tb else B.double()) + bias.double() + (C0.double() if acc else 0)
    Ad, Bd, bd = A.to(dev), B.to(dev), bias.to(dev)
    Cd = C0.to(dev)
    check(lib().capnet_sgemm_splitk(0, int(tb), M, N, K, ptr(Ad), K, ptr(Bd), Bd.shape[1], ptr(Cd), N, ptr(bd), acc, None, 0, current_stream()))
    e = (Cd.cpu().double() - ref).abs()
    cols = e.max(0).values
    bad = (cols > 1e-3).nonzero().flatten().tolist()
    print(M, N, K, tb, acc, "max err", e.max().item(), "rel", e.max().item()/ref.abs().max().item(), "bad cols", bad[:10], len(bad))
