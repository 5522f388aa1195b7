"""Diagnostic: which decoder kernel is disturbed by conv1x1_fwd_bf16x6 running on another stream?"""
import sys, torch, ctypes as C
sys.path.insert(0, '/root/repo')
import capnet
from capnet import ops
from capnet._lib import check, lib, ptr
dev = torch.device('cuda:0'); L = lib()
side = torch.cuda.Stream(priority=-1); other = torch.cuda.Stream()
Bc, H, Cin, Cout = 64, 28, 512, 128
M = Bc * H * H
x = torch.randn(Bc, H, H, Cin, device=dev); w = torch.randn(Cout, Cin, device=dev) * 0.05
y = torch.empty(M, Cout, device=dev)
t = L.capnet_conv1x1_tiles_m(M)
ps, pq = torch.empty(t, Cout, device=dev), torch.empty(t, Cout, device=dev)
img = torch.empty(L.capnet_conv1x1_bf16x6_weight_words(Cin, Cout), dtype=torch.int32, device=dev)
check(L.capnet_conv1x1_bf16x6_pack(ptr(w), ptr(img), Cout, Cin, 64, C.c_void_p(torch.cuda.current_stream().cuda_stream)))
def noise():
    s = C.c_void_p(other.cuda_stream)
    for _ in range(60):
        check(L.capnet_conv1x1_fwd_bf16x6(ptr(x), H * H * Cin, H * Cin, Cin, ptr(img), 64, ptr(y), None, None, 0, ptr(ps), ptr(pq),
                                          Bc, H, H, Cin, Cout, 1, None, None, None, 0, s))
g = torch.Generator().manual_seed(1)
A784 = torch.randn(784, 2048, generator=g).to(dev); We = (torch.randn(512, 2048, generator=g) * 0.02).to(dev); be = torch.zeros(512, device=dev)
h = torch.randn(4, 512, generator=g).to(dev); Wz = (torch.randn(4608, 512, generator=g) * 0.05).to(dev)
xa = torch.randn(4, 2348, generator=g).to(dev); Vc = (torch.randn(2048, 2348, generator=g) * 0.02).to(dev)
att1 = torch.randn(4, 196, 512, generator=g).to(dev); feat = torch.rand(4, 196, 2048, generator=g).to(dev)
z = torch.randn(4, 512 + 2048, generator=g).to(dev); wf = (torch.randn(1, 512, generator=g) * 0.1).to(dev); bfv = torch.zeros(1, device=dev)
cases = {
    "attention_step": lambda: torch.cat([t_.reshape(-1) for t_ in ops.attention_step(att1, feat, z.clone(), 512, wf, bfv)]),
}
for name, fn in cases.items():
    ref = fn().clone(); torch.cuda.synchronize()
    bad = 0; worst = 0.0
    for rep in range(5):
        other.wait_stream(torch.cuda.current_stream()); side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(other):
            noise()
        with torch.cuda.stream(side):
            outs = [fn() for _ in range(20)]
        torch.cuda.synchronize()
        for o in outs:
            if not torch.equal(o, ref):
                bad += 1; worst = max(worst, float((o - ref).abs().max()))
    print("%-40s mismatches %3d / 100, worst %.2e" % (name, bad, worst))
