"""Train-mode trunk features of one batch under the 1x1 kernel variants, compared with each other (GPU vs GPU).
    python tools/trunk_variants.py [B]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import capnet
from capnet import synthetic
from capnet.model import EncoderCNN
dev = torch.device("cuda:0"); B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
imgs = synthetic.make_batch(B, 100, seed=0)[0].to(dev)
enc0 = EncoderCNN(300)
sd = enc0.state_dict()
st = synthetic.trunk_state({k: v for k, v in sd.items() if k.startswith("resnet.")}, seed=1234)
for k, v in sd.items():
    if k not in st: st[k] = v
def run(env):
    for k in ("CAPNET_NO_X6", "CAPNET_NO_H3"): os.environ.pop(k, None)
    for k in env: os.environ[k] = "1"
    enc = EncoderCNN(300); enc.load_state_dict({k: v.clone() for k, v in st.items()}); enc.to(dev).train()
    pooled, fmap = enc._trunk().forward(imgs, True, True, True)
    return pooled.double().cpu(), fmap.double().cpu()
rel = lambda a, b: ((a - b).abs().max() / b.abs().max()).item()
p32, m32 = run(["CAPNET_NO_X6"]); pb, mb = run(["CAPNET_NO_H3"]); ph, mh = run([])
p32b, _ = run(["CAPNET_NO_X6"])
print("pooled  f32 rerun %.2e | bf16x6 vs f32 %.2e | f16x3 vs f32 %.2e | f16x3 vs bf16x6 %.2e" % (rel(p32b, p32), rel(pb, p32), rel(ph, p32), rel(ph, pb)))
print("map     bf16x6 vs f32 %.2e | f16x3 vs f32 %.2e" % (rel(mb, m32), rel(mh, m32)))
d = (ph - p32).abs(); print("f16x3 - f32: mean abs %.3e  max %.3e at %s; value range %.3f" % (d.mean(), d.max(), tuple(int(i) for i in (d == d.max()).nonzero()[0]), p32.abs().max()))
d = (pb - p32).abs(); print("bf16x6 - f32: mean abs %.3e  max %.3e" % (d.mean(), d.max()))
print("finite:", torch.isfinite(ph).all().item(), " max |map| %.1f" % mh.abs().max())
