"""Diagnostic: three trunk passes in flight on three streams (as TrunkPipeline runs them) must produce
exactly what the same passes produce one after the other."""
import sys, torch
sys.path.insert(0, '/root/repo')
import capnet
from capnet import synthetic, model_att
dev = torch.device('cuda:0')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
enc = model_att.EncoderCNN(14)
enc.load_state_dict(synthetic.trunk_state(enc.state_dict(), seed=1234))
enc.to(dev).train()
imgs = [synthetic.make_batch(B, 100, seed=3 + k)[0].to(dev) for k in range(3)]
ref = [enc(im, slot=k, defer_stats=True)[0].clone() for k, im in enumerate(imgs)]
torch.cuda.synchronize()
streams = [torch.cuda.Stream() for _ in range(3)]
for rep in range(4):
    outs = []
    for k, (im, st) in enumerate(zip(imgs, streams)):
        st.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(st):
            outs.append(enc(im, slot=k, defer_stats=True)[0])
    torch.cuda.synchronize()
    print("rep", rep, [(torch.equal(o, r), float((o - r).abs().max())) for o, r in zip(outs, ref)])
# the same with a chain of small launches on a high-priority stream beside the passes (the decoder's role)
from capnet import ops
hi = torch.cuda.Stream(priority=-1)
xa = torch.randn(12, 512, device=dev); wa = torch.randn(2048, 512, device=dev) * 0.05
for rep in range(4):
    outs = []
    for k, (im, st) in enumerate(zip(imgs, streams)):
        st.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(st):
            outs.append(enc(im, slot=k, defer_stats=True)[0])
    with torch.cuda.stream(hi):
        for _ in range(1500):
            y = ops.sgemm_splitk(xa, wa, transB=True)
    torch.cuda.synchronize()
    print("with high-priority chain, rep", rep, [(torch.equal(o, r), float((o - r).abs().max())) for o, r in zip(outs, ref)])
