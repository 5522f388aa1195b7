"""Diagnostic: attention decoder forward, alone vs beside three trunk passes on other streams."""
import random, sys, torch
sys.path.insert(0, '/root/repo')
import capnet
from capnet import synthetic, model_att, ops
dev = torch.device('cuda:0')
V, B = 1000, 4
enc = model_att.EncoderCNN(14)
enc.load_state_dict(synthetic.trunk_state(enc.state_dict(), seed=1234))
dec = model_att.DecoderFactoredLSTMAtt(512, 300, 512, 512, V, 1, dropout=0.0)
dec.load_state_dict(synthetic.decoder_state(dec.state_dict(), seed=1234))
enc.to(dev).train(); dec.to(dev).train()
imgs, caps, lens = synthetic.make_batch(B, V, seed=40)
imgs, caps = imgs.to(dev), caps.to(dev)
feats = enc(imgs)
lens1 = [l - 1 for l in lens]
random.seed(6)
tf = [random.random() < 0.8 for _ in range(24)]
cin = caps[:, :-1].contiguous()
def fwd():
    with torch.no_grad():
        out, alphas = dec(cin, lens1, feats, tf_mask=tf)
    return out, alphas
ref_out, ref_al = fwd()
torch.cuda.synchronize()
side = torch.cuda.Stream(priority=-1)
streams = [torch.cuda.Stream() for _ in range(3)]
other = [synthetic.make_batch(B, V, seed=50 + k)[0].to(dev) for k in range(3)]
for rep in range(6):
    for k, st in enumerate(streams):
        st.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(st):
            enc(other[k], slot=k, defer_stats=True)
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        out, al = fwd()
    torch.cuda.synchronize()
    print("rep", rep, "logits equal", bool(torch.equal(out, ref_out)), float((out - ref_out).abs().max()),
          "alphas equal", bool(torch.equal(al, ref_al)), float((al - ref_al).abs().max()))
