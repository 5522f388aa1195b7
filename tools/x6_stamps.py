"""Clock stamps of the persistent bf16x6 1x1 kernel (workgroup 0, first 64 steps): per step the cycles spent
waiting for the loads, in MFMA + fold/split + LDS writes, issuing the next loads, and at the barrier.
    python tools/x6_stamps.py s3c3 128"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import capnet
from capnet._lib import check, current_stream, lib, ptr
dev = torch.device("cuda:0"); L = lib(); B = 64
KIND = os.environ.get("X6_KIND", "f16x3")      # bf16x6 | f16x3; B = 64
X6 = {"s1c1": (56, 256, 64, 1, 0), "s1c3": (56, 64, 256, 1, 1), "s2c1": (28, 512, 128, 1, 0), "s2c3": (28, 128, 512, 1, 1),
      "s3c1": (14, 1024, 256, 1, 0), "s3c3": (14, 256, 1024, 1, 1), "s4c1": (7, 2048, 512, 1, 0), "s4c3": (7, 512, 2048, 1, 1)}
name, bn = sys.argv[1], int(sys.argv[2])
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
for _ in range(3): run()
st = torch.zeros(64, 4, dtype=torch.int64, device=dev)
os.environ["CAPNET_H3_STAMPS" if KIND == "f16x3" else "CAPNET_X6_STAMPS"] = hex(st.data_ptr())
run(); torch.cuda.synchronize()
del os.environ["CAPNET_H3_STAMPS" if KIND == "f16x3" else "CAPNET_X6_STAMPS"]
s = st.cpu()
n = min(int((s[:, 0] != 0).sum()), int(sys.argv[4]) if len(sys.argv) > 4 else 64)
s = s[:n]
d = lambda a, b: (s[:, b] - s[:, a]).float()
nxt = (s[1:, 0] - s[:-1, 3]).float()
print("%s bn %d  abl %s: %d steps; cycles per step: wait %.0f  mfma+split+lds %.0f  issue %.0f  barrier(+epilogue) %.0f   total %.0f" % (
    name, bn, os.environ.get("CAPNET_X6_ABLATE", "0"), n, d(0, 1).mean(), d(1, 2).mean(), d(2, 3).mean(), nxt.mean(),
    (s[-1, 0] - s[0, 0]).item() / max(n - 1, 1)))
print("first 20 steps [wait, work, issue, barrier]:")
for i in range(min(int(sys.argv[3]) if len(sys.argv) > 3 else 20, n - 1)):
    print("  %2d: %5d %5d %5d %5d" % (i, d(0, 1)[i], d(1, 2)[i], d(2, 3)[i], nxt[i]))
