"""Diagnostic: attention decoder forward beside ONE kind of conv kernel looping on another stream."""
import random, sys, torch
sys.path.insert(0, '/root/repo')
import capnet
from capnet import synthetic, model_att, ops
from capnet._lib import check, lib, ptr
import ctypes as C
dev = torch.device('cuda:0'); L = lib()
V, B = 1000, 4
dec = model_att.DecoderFactoredLSTMAtt(512, 300, 512, 512, V, 1, dropout=0.0)
dec.load_state_dict(synthetic.decoder_state(dec.state_dict(), seed=1234))
dec.to(dev).train()
_, caps, lens = synthetic.make_batch(B, V, seed=40, images=False)
caps = caps.to(dev)
feats = torch.rand(B, 14, 14, 2048, device=dev)
lens1 = [l - 1 for l in lens]
random.seed(6)
tf = [random.random() < 0.8 for _ in range(24)]
cin = caps[:, :-1].contiguous()
def fwd():
    with torch.no_grad():
        return dec(cin, lens1, feats, tf_mask=tf)
ref_out, ref_al = fwd()
torch.cuda.synchronize()
side = torch.cuda.Stream(priority=-1)
other = torch.cuda.Stream()
# conv operands in their own big allocations
Bc, H, Cin, Cout = 64, 28, 512, 128
M = Bc * H * H
x = torch.randn(Bc, H, H, Cin, device=dev); w = torch.randn(Cout, Cin, device=dev) * 0.05
y = torch.empty(M + 4096, Cout, device=dev)
sc, sh = torch.rand(Cin, device=dev) + 0.5, torch.randn(Cin, device=dev)
t = L.capnet_conv1x1_tiles_m(M)
ps, pq = torch.empty(t + 64, Cout, device=dev), torch.empty(t + 64, Cout, device=dev)
wk = ops.pack_conv_weight(w.reshape(Cout, Cin, 1, 1), Cin, kmajor=True)
def conv(kind, st):
    s = C.c_void_p(st.cuda_stream)
    if kind.startswith("x6"):
        bn = 128 if "128" in kind else 64
        pre = 1 if "pre" in kind else 0
        check(L.capnet_conv1x1_fwd_bf16x6(ptr(x), H * H * Cin, H * Cin, Cin, ptr(imgs[bn]), bn, ptr(y), ptr(sc) if pre else None,
                                          ptr(sh) if pre else None, pre, ptr(ps), ptr(pq), Bc, H, H, Cin, Cout, 1, None, None, None, 0, s))
    else:
        check(L.capnet_conv2d_fwd_kmajor(ptr(x), H * H * Cin, H * Cin, Cin, ptr(wk), Cin, ptr(y), None, None, 0, ptr(ps), ptr(pq),
                                         Bc, H, H, Cin, Cout, 1, 1, 1, 0, 12864, None, s))
imgs = {}
for bn in (64, 128):
    imgs[bn] = torch.empty(L.capnet_conv1x1_bf16x6_weight_words(Cin, Cout), dtype=torch.int32, device=dev)
    check(L.capnet_conv1x1_bf16x6_pack(ptr(w), ptr(imgs[bn]), Cout, Cin, bn, C.c_void_p(torch.cuda.current_stream().cuda_stream)))
torch.cuda.synchronize()
big = torch.randn(64 * 1024 * 1024, device=dev)
def torch_noise(st):
    for _ in range(30):
        z = torch.sin(big) * 1.0001
for kind in ("torch-sin", "kmajor", "x6-64", "x6-128", "x6-64-pre", "x6-128-pre"):
    bad = 0; worst = 0.0
    for rep in range(5):
        other.wait_stream(torch.cuda.current_stream()); side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(other):
            if kind == "torch-sin":
                torch_noise(other)
            else:
                for _ in range(60):
                    conv(kind, other)
        with torch.cuda.stream(side):
            out, al = fwd()
        torch.cuda.synchronize()
        if not torch.equal(out, ref_out):
            bad += 1; worst = max(worst, float((out - ref_out).abs().max()))
    print(kind, "decoder mismatches: %d / 5, worst logits diff %.2e" % (bad, worst))
