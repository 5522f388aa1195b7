"""att_scores_fwd_kernel with its per-lane partial sums kept, beside a conv kernel: which lanes, which values go wrong?
    python tools/probes/scores_beside.py [h3|none]"""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import capnet
from capnet._lib import check, lib, ptr
dev = torch.device("cuda:0"); L = lib()
Pl = C.CDLL(os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "build", "libbperm_probe.so"))
Pl.launch_scores_probe.argtypes = [C.c_void_p, C.c_void_p, C.c_long, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
KIND = sys.argv[1] if len(sys.argv) > 1 else "h3"
Bc, H, Cin, Cout = 64, 28, 512, 128
M = Bc * H * H
x = torch.full((Bc, H, H, Cin), float(os.environ.get("AGG_X", "0")) or 1.0, device=dev) if os.environ.get("AGG_X") else torch.randn(Bc, H, H, Cin, device=dev); w = torch.randn(Cout, Cin, device=dev) * 0.05
y = torch.empty(M, Cout, device=dev)
t = L.capnet_conv1x1_tiles_m(M)
ps, pq = torch.empty(t, Cout, device=dev), torch.empty(t, Cout, device=dev)
name = {"h3": "f16x3"}.get(KIND)
if name:
    img = torch.empty(getattr(L, "capnet_conv1x1_%s_weight_words" % name)(Cin, Cout), dtype=torch.int32, device=dev)
    check(getattr(L, "capnet_conv1x1_%s_pack" % name)(ptr(w), ptr(img), Cout, Cin, 128, C.c_void_p(torch.cuda.current_stream().cuda_stream)))
Pl.launch_mfma_aggr.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
dummy = torch.zeros(4, device=dev)
g = torch.Generator().manual_seed(1)
rows, P, A = 4, 196, 512
att1 = torch.randn(rows, P, A, generator=g).to(dev); z = torch.randn(rows, 512 + 2048, generator=g).to(dev); wf = (torch.randn(A, generator=g) * 0.1).to(dev)
def run(stream):
    esc = torch.empty(rows, P, device=dev); part = torch.empty(rows, P, 64, device=dev)
    Pl.launch_scores_probe(att1.data_ptr(), z.data_ptr(), z.shape[1], wf.data_ptr(), rows, P, A, esc.data_ptr(), part.data_ptr(), stream.cuda_stream)
    return esc, part
ref_e, ref_p = run(torch.cuda.current_stream()); torch.cuda.synchronize()
other = torch.cuda.Stream(); side = torch.cuda.Stream(priority=-1)
bad = 0; shown = 0
for rep in range(5):
    other.wait_stream(torch.cuda.current_stream()); side.wait_stream(torch.cuda.current_stream())
    if name:
        for _ in range(60):
            check(getattr(L, "capnet_conv1x1_fwd_%s" % name)(ptr(x), H * H * Cin, H * Cin, Cin, ptr(img), 128, ptr(y), None, None, 0, ptr(ps), ptr(pq),
                                                               Bc, H, H, Cin, Cout, 1, None, None, None, 0, C.c_void_p(other.cuda_stream)))
    if KIND.startswith("mfma"):
        for _ in range(6):
            Pl.launch_mfma_aggr(int(KIND[4:]), dummy.data_ptr(), int(os.environ.get("AGG_BLOCKS", "512")), 20000, other.cuda_stream)
    with torch.cuda.stream(side):
        outs = [run(side) for _ in range(20)]
    torch.cuda.synchronize()
    for e, p in outs:
        if not (torch.equal(e, ref_e) and torch.equal(p, ref_p)):
            bad += 1
            if shown < 4:
                shown += 1
                de = torch.nonzero(e != ref_e); dp = torch.nonzero(p != ref_p)
                print("escore wrong at %d places, partials wrong at %d places" % (de.shape[0], dp.shape[0]))
                for i in dp[:6].tolist():
                    jj, pp, ll = i
                    print("   partial (row %d, pixel %d, lane %d): got %.7g want %.7g | same-lane partial of pixel-1: %.7g, pixel+1: %.7g" % (
                        jj, pp, ll, p[jj, pp, ll].item(), ref_p[jj, pp, ll].item(), ref_p[jj, max(pp - 1, 0), ll].item(), ref_p[jj, min(pp + 1, P - 1), ll].item()))
                for i in de[:4].tolist():
                    print("   escore (row %d, pixel %d): got %.7g want %.7g; sum of got partials %.7g" % (i[0], i[1], e[i[0], i[1]].item(), ref_e[i[0], i[1]].item(), p[i[0], i[1]].sum().item()))
print("aggressor %s: %d of 100 launches differ" % (KIND, bad))
