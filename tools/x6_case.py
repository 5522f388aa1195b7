"""One bf16x6 1x1 conv case against fp64 (debugging aid): python tools/x6_case.py B H W Cin Cout stride pre bn"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import capnet
from capnet._lib import check, current_stream, lib, ptr
B, H, W, Cin, Cout, stride, pre, bn = [int(v) for v in sys.argv[1:9]]
dev = torch.device("cuda:0"); L = lib()
KIND = os.environ.get("X6_KIND", "f16x3")      # bf16x6 | f16x3
g = torch.Generator().manual_seed(1)
x = torch.randn(B, H, W, Cin, generator=g); w = torch.randn(Cout, Cin, generator=g) * 0.1
sc = torch.rand(Cin, generator=g) - 0.3; sh = torch.randn(Cin, generator=g)
xin = torch.relu(x * sc + sh).double() if pre else x.double()
xs = xin[:, ::stride, ::stride, :]
M = xs.shape[0] * xs.shape[1] * xs.shape[2]
ref = xs.reshape(M, Cin) @ w.double().t()
xd, wd = x.to(dev), w.to(dev)
img = torch.empty(getattr(L, 'capnet_conv1x1_%s_weight_words' % KIND)(Cin, Cout), dtype=torch.int32, device=dev)
check(getattr(L, 'capnet_conv1x1_%s_pack' % KIND)(ptr(wd), ptr(img), Cout, Cin, bn, current_stream()))
torch.cuda.synchronize(); print("packed", flush=True)
y = torch.full((M, Cout), float("nan"), device=dev)
t = L.capnet_conv1x1_tiles_m(M)
ps = torch.full((t, Cout), float("nan"), device=dev); pq = torch.full((t, Cout), float("nan"), device=dev)
sd, hd = (sc.to(dev), sh.to(dev)) if pre else (None, None)
check(getattr(L, 'capnet_conv1x1_fwd_%s' % KIND)(ptr(xd), H * W * Cin, W * Cin, Cin, ptr(img), bn, ptr(y), ptr(sd), ptr(hd), pre, ptr(ps), ptr(pq),
                                  B, H, W, Cin, Cout, stride, None, None, None, 0, current_stream()))
torch.cuda.synchronize(); print("ran", flush=True)
err = ((y.double().cpu() - ref).abs().max() / ref.abs().max()).item()
print("max rel err %.2e  sum err %.2e" % (err, ((ps.sum(0).double().cpu() - ref.sum(0)).abs().max() / ref.sum(0).abs().max()).item()))
if err > 1e-4:
    d = (y.double().cpu() - ref).abs() / ref.abs().max()
    bad = d > 1e-4
    rows = bad.any(dim=1).nonzero().flatten().tolist(); cols = bad.any(dim=0).nonzero().flatten().tolist()
    print("bad elements %d of %d; rows %s... (%d rows); cols %s... (%d cols)" % (bad.sum(), bad.numel(), rows[:16], len(rows), cols[:16], len(cols)))
    r0, c0 = rows[0], cols[0]
    print("y[%d,%d] got %.6f want %.6f ratio %.4f" % (r0, c0, y[r0, c0].item(), ref[r0, c0].item(), y[r0, c0].item() / ref[r0, c0].item()))
