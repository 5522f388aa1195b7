"""Run-to-run determinism of the fused block path: the launches alone on the trunk's stage shapes (bit-for-bit against their
first run, outputs pre-filled with noise), then the whole trunk pass.   python tools/probes/fused_determinism.py [B] [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import capnet  # noqa: E402,F401
from capnet import ops, synthetic  # noqa: E402
from capnet._lib import check, current_stream, lib, ptr  # noqa: E402
from capnet.model import EncoderCNN  # noqa: E402

dev = torch.device("cuda:0")
L = lib()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
REPS = int(sys.argv[2]) if len(sys.argv) > 2 else 100

for side, MID in ((56, 64), (28, 128), (14, 256)):
    M, C = B * side * side, 4 * MID
    g = torch.Generator().manual_seed(MID)
    y2 = torch.randn(M, MID, generator=g).to(dev)
    s2, t2 = (torch.rand(MID, generator=g) + 0.5).to(dev), (torch.randn(MID, generator=g) * 0.5).to(dev)
    w3 = (torch.randn(C, MID, 1, 1, generator=g) * (2.0 / MID) ** 0.5).to(dev)
    w1 = (torch.randn(MID, C, 1, 1, generator=g) * (2.0 / C) ** 0.5).to(dev)
    gamma, beta = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    res = torch.randn(M, C, generator=g).to(dev)
    sd, td = (torch.rand(C, generator=g) + 0.5).to(dev), (torch.randn(C, generator=g) * 0.5).to(dev)
    img3, img1 = ops.pack_fused_block_weight(w3, 0), ops.pack_fused_block_weight(w1, 1)
    nwork = L.capnet_fused_block_stats_floats(M, MID)
    tiles = L.capnet_fused_block_tiles(M, MID)
    err = ops.err_flag(dev)
    st = current_stream()
    first = None
    bad = {"sc": 0, "sh": 0, "work": 0, "out": 0, "y1": 0, "ps": 0, "pq": 0, "out_fold": 0, "y1_fold": 0}
    for it in range(REPS):
        work = torch.randn(nwork, device=dev)
        sc, sh = torch.randn(C, device=dev), torch.randn(C, device=dev)
        check(L.capnet_fused_block_stats(ptr(y2), ptr(s2), ptr(t2), ptr(img3), M, MID, 0, ptr(gamma), ptr(beta), None, None, 0.1,
                                         1e-5, ptr(sc), ptr(sh), ptr(work), ptr(err), st))
        outs = []
        for fold in (False, True):
            out, y1 = torch.randn(M, C, device=dev), torch.randn(M, MID, device=dev)
            ps, pq = torch.randn(tiles, MID, device=dev), torch.randn(tiles, MID, device=dev)
            check(L.capnet_fused_block_forward(ptr(y2), ptr(s2), ptr(t2), ptr(img3), ptr(sc), ptr(sh), ptr(res),
                                               ptr(sd) if fold else None, ptr(td) if fold else None, ptr(out),
                                               ptr(img1), ptr(y1), ptr(ps), ptr(pq), M, MID, 0, 0, ptr(err), st))
            outs += [out, y1, ps, pq]
        cur = {"sc": sc, "sh": sh, "out": outs[0], "y1": outs[1], "ps": outs[2], "pq": outs[3], "out_fold": outs[4], "y1_fold": outs[5]}
        if first is None:
            first = cur
        else:
            for k, v in cur.items():
                if not torch.equal(v, first[k]):
                    bad[k] += 1
                    if bad[k] == 1:
                        d = (v - first[k]).abs()
                        print("   first difference in %s: %d elements, max %.3e (of max %.3e), rows %s" %
                              (k, int((d > 0).sum()), float(d.max()), float(first[k].abs().max()),
                               (d.reshape(d.shape[0], -1).amax(1) > 0).nonzero().flatten()[:8].tolist() if d.dim() > 1 else ""))
    print("stage %dx%d MID %3d M %6d: %d runs, differing runs %s" % (side, side, MID, M, REPS, {k: v for k, v in bad.items() if v}))

enc = EncoderCNN(300)
enc.load_state_dict(synthetic.encoder_state(enc.state_dict(), seed=5))
enc.to(dev).train()
images = synthetic.make_batch(B, 1000, seed=12)[0].to(dev)
first, n_bad = None, 0
for it in range(max(REPS // 4, 10)):
    junk = torch.randn(1 << 20, device=dev)
    with torch.no_grad():
        f = enc(images).clone()
    if first is None:
        first = f
    elif not torch.equal(f, first):
        n_bad += 1
        if n_bad == 1:
            print("   trunk features differ: max %.3e of %.3e" % (float((f - first).abs().max()), float(first.abs().max())))
print("whole encoder pass, B %d: %d runs, %d differ from the first (fused block %s)" %
      (B, max(REPS // 4, 10), n_bad, "off" if os.environ.get("CAPNET_NO_FUSED_BLOCK") == "1" else "on"))
capnet.ops.check_device_errors()
