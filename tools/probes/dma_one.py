"""One trunk 1x1 shape through both 1x1 kernels (for counter passes)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import capnet
from capnet import ops
from capnet._lib import check, current_stream, lib, ptr
dev = torch.device("cuda:0"); L = lib()
B, H, Cin, Cout, stride = 64, int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), 1
M = B * H * H
x = torch.randn(B, H, H, Cin, device=dev); w = torch.randn(Cout, Cin, device=dev) * 0.05
y = torch.empty(M, Cout, device=dev)
t = L.capnet_conv1x1_tiles_m(M)
ps, pq = torch.empty(2 * t, Cout, device=dev), torch.empty(2 * t, Cout, device=dev)
wk = ops.pack_conv_weight(w.reshape(Cout, Cin, 1, 1), Cin, kmajor=True)
slabs = torch.empty(max(1, L.capnet_conv_kmajor_slab_floats(M, Cout, Cin, 12864)), device=dev)
for _ in range(5):
    check(L.capnet_conv1x1_fwd_dma(ptr(x), H * H * Cin, H * Cin, Cin, ptr(w), ptr(y), ptr(ps), ptr(pq), B, H, H, Cin, Cout, stride, None, None, None, 0, current_stream()))
    check(L.capnet_conv2d_fwd_kmajor(ptr(x), H * H * Cin, H * Cin, Cin, ptr(wk), Cin, ptr(y), None, None, 0, ptr(ps), ptr(pq), B, H, H, Cin, Cout, 1, 1, stride, 0, 12864, None, current_stream()))
torch.cuda.synchronize()
