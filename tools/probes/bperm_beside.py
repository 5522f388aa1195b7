"""Register-only wave reductions (ds_bpermute vs DPP) beside a conv kernel: does the conv kernel disturb them?
    python tools/probes/bperm_beside.py [h3|none]"""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import capnet
from capnet._lib import check, lib, ptr
dev = torch.device("cuda:0"); L = lib()
P = C.CDLL(os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "build", "libbperm_probe.so"))
P.launch_probe.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
KIND = sys.argv[1] if len(sys.argv) > 1 else "h3"
Bc, H, Cin, Cout = 64, 28, 512, 128
M = Bc * H * H
x = torch.randn(Bc, H, H, Cin, device=dev); w = torch.randn(Cout, Cin, device=dev) * 0.05
y = torch.empty(M, Cout, device=dev)
t = L.capnet_conv1x1_tiles_m(M)
ps, pq = torch.empty(t, Cout, device=dev), torch.empty(t, Cout, device=dev)
name = {"h3": "f16x3"}.get(KIND)
if name:
    img = torch.empty(getattr(L, "capnet_conv1x1_%s_weight_words" % name)(Cin, Cout), dtype=torch.int32, device=dev)
    check(getattr(L, "capnet_conv1x1_%s_pack" % name)(ptr(w), ptr(img), Cout, Cin, 128, C.c_void_p(torch.cuda.current_stream().cuda_stream)))
P.launch_vmcnt_probe.argtypes = [C.c_void_p, C.c_long, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
vscr = torch.zeros(16 * 256 * 24, device=dev)
vbuf = torch.arange(4 * 1024 * 1024, device=dev, dtype=torch.float32)
P.launch_first_use_probe.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
fbuf = torch.arange(4 * 196 * 512, device=dev, dtype=torch.float32); ibuf = torch.arange(4 * 196 * 512, device=dev, dtype=torch.int32)
P.launch_valu_probe.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
P.launch_mfma_aggr.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
dummy = torch.zeros(4, device=dev)
P.launch_load_probe.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
lbuf = torch.arange(4 * 196 * 512, device=dev, dtype=torch.float32)
other = torch.cuda.Stream(); side = torch.cuda.Stream(priority=-1)
out = torch.zeros(32, dtype=torch.int32, device=dev)
torch.cuda.synchronize()
for rep in range(5):
    other.wait_stream(torch.cuda.current_stream()); side.wait_stream(torch.cuda.current_stream())
    if name:
        for _ in range(60):
            check(getattr(L, "capnet_conv1x1_fwd_%s" % name)(ptr(x), H * H * Cin, H * Cin, Cin, ptr(img), 128, ptr(y), None, None, 0, ptr(ps), ptr(pq),
                                                               Bc, H, H, Cin, Cout, 1, None, None, None, 0, C.c_void_p(other.cuda_stream)))
    if KIND.startswith("mfma"):
        for _ in range(6):
            P.launch_mfma_aggr(int(KIND[4:]), dummy.data_ptr(), 512, 20000, other.cuda_stream)
    for _ in range(40):
        P.launch_valu_probe(out.data_ptr(), 16, 2000, side.cuda_stream)
        for _ in range(10):
            P.launch_first_use_probe(fbuf.data_ptr(), ibuf.data_ptr(), 4, 196, 512, out.data_ptr(), side.cuda_stream)
        P.launch_vmcnt_probe(vbuf.data_ptr(), vbuf.numel(), 16, 200, out.data_ptr(), vscr.data_ptr(), side.cuda_stream)
        P.launch_probe(0, out.data_ptr(), 16, 2000, side.cuda_stream)
        P.launch_probe(1, out.data_ptr(), 16, 2000, side.cuda_stream)
        P.launch_load_probe(lbuf.data_ptr(), 4, 196, 512, out.data_ptr(), side.cuda_stream)
    torch.cuda.synchronize()
o = out.cpu().tolist()
print("first consumers of loaded registers (wrong of %d each): v_fma_f32 %d, v_pk_fma_f32 %d, v_mul_lo_u32 %d, v_exp_f32 %d, v_rcp_f32 %d, v_lshl_add_u64 %d, v_cvt_f32_u32 %d, v_cvt_f64_f32 %d"
      % ((5 * 40 * 10 * 4 * 196 * 512 // 4,) + tuple(o[16:24])))
print("counted-wait probe: wrong copies behind vmcnt(4) %d, (3) %d, (2) %d, (1) %d, (0) %d  (of %d loads each)" % (o[8], o[9], o[10], o[11], o[12], 5 * 40 * 16 * 256 * 200))
print("valu probe (of %d each): wrong v_pk_fma_f32 %d, v_pk_fma_f32 op_sel_hi %d, v_pk_mul_f32 %d, v_pk_add_f32 %d, v_fma_f32 %d" % (5 * 40 * 16 * 256 * 2000, o[6], o[13], o[14], o[15], o[7]))
print("load probe: wrong elements %d (first index %d got bits 0x%08x = %s)" % (o[2], o[3], o[4] & 0xffffffff, torch.tensor([o[4]], dtype=torch.int32).view(torch.float32).item()))
print("aggressor %s: wrong reductions  ds_bpermute %d   dpp %d   (of %d each)" % (KIND, out[0].item(), out[1].item(), 5 * 40 * 16 * 4 * 2000))
