"""numpy model of the split-operand dot products (fp32 accumulation over k16 groups, as the MFMA does): rms error against
fp64 of the fp32 fma chain, the 3-way bf16 split (six products, csrc/conv_bf16x6.hip), the 2-way f16 split (three products,
csrc/conv_f16x3.hip) with and without power-of-two prescaling and with f16 subnormals flushed.
    python tools/probes/split_error.py"""
import numpy as np
rng = np.random.default_rng(0)
def bf16(x):
    u = x.astype(np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7fff + ((u >> 16) & 1)) & 0xffff0000
    return u.astype(np.uint32).view(np.float32)
def f16(x): return x.astype(np.float16).astype(np.float32)
def acc32(prods):   # sequential fp32 accumulation along last axis
    s = np.zeros(prods.shape[:-1], np.float32)
    for k in range(prods.shape[-1]): s = (s + prods[..., k]).astype(np.float32)
    return s
for K in (256, 1024):
    M, N = 64, 64
    x = (np.maximum(rng.standard_normal((M, K)), 0) * np.exp(rng.standard_normal((M, K)))).astype(np.float32)
    w = (rng.standard_normal((N, K)) * 0.05 * np.exp(0.5 * rng.standard_normal((N, K)))).astype(np.float32)
    ref = x.astype(np.float64) @ w.astype(np.float64).T
    rms = lambda y: np.sqrt(((y - ref) ** 2).mean() / (ref ** 2).mean())
    # fp32 fma chain (emulated: exact product in f64 then round to f32 each accumulate)
    s = np.zeros((M, N), np.float32)
    for k in range(K): s = (s.astype(np.float64) + x[:, k:k+1].astype(np.float64) * w[:, k].astype(np.float64)[None, :]).astype(np.float32)
    print("K", K, "fp32 fma chain rms", rms(s))
    # bf16x3 six terms, MFMA-like: groups of 16 k summed exactly (f64) then added to f32 acc
    xh = bf16(x); xm = bf16(x - xh); xl = bf16(x - xh - xm)
    wh = bf16(w); wm = bf16(w - wh); wl = bf16(w - wh - wm)
    def mfma_acc(terms, kb=16):
        acc = np.zeros((M, N), np.float32)
        for k0 in range(0, K, kb):
            for a, b in terms:
                p = a[:, k0:k0+kb].astype(np.float64) @ b[:, k0:k0+kb].astype(np.float64).T
                acc = (acc.astype(np.float64) + p).astype(np.float32)
        return acc
    print("   bf16x3 6-term rms", rms(mfma_acc([(xl, wh), (xh, wl), (xm, wm), (xm, wh), (xh, wm), (xh, wh)])))
    # fp16x2 with scaled residual, separate accumulators for cross terms
    S = 2048.0
    xh6 = f16(x); xl6 = f16((x - xh6) * S); wh6 = f16(w); wl6 = f16((w - wh6) * S)
    print("   f16 overflow/underflow check: max |x|", np.abs(x).max(), " zero-flushed residuals:", (xl6 == 0).mean(), (wl6 == 0).mean())
    a_hh = mfma_acc([(xh6, wh6)])
    a_cr = mfma_acc([(xl6, wh6), (xh6, wl6)])
    y = (a_hh.astype(np.float64) + a_cr.astype(np.float64) / S).astype(np.float32)
    print("   fp16x2 3-term (scaled residual, 2 accumulators) rms", rms(y))
    # single accumulator with unscaled residual
    xl6u = f16(x - xh6); wl6u = f16(w - wh6)
    print("   fp16x2 3-term unscaled, 1 accumulator rms", rms(mfma_acc([(xl6u, wh6), (xh6, wl6u), (xh6, wh6)])))
    # bf16x2 3-term for comparison
    print("   bf16x2 3-term rms", rms(mfma_acc([(xm, wh), (xh, wm), (xh, wh)])))
print("---- prescaled single-accumulator variants")
for K in (64, 256, 1024, 2048):
    M, N = 64, 64
    x = (np.maximum(rng.standard_normal((M, K)), 0) * np.exp(rng.standard_normal((M, K)))).astype(np.float32)
    w = (rng.standard_normal((N, K)) * 0.05 * np.exp(0.5 * rng.standard_normal((N, K)))).astype(np.float32)
    ref = x.astype(np.float64) @ w.astype(np.float64).T
    rms = lambda y: np.sqrt(((y - ref) ** 2).mean() / (ref ** 2).mean())
    s = np.zeros((M, N), np.float32)
    for k in range(K): s = (s.astype(np.float64) + x[:, k:k+1].astype(np.float64) * w[:, k].astype(np.float64)[None, :]).astype(np.float32)
    def mfma_acc(terms, kb=16):
        acc = np.zeros((M, N), np.float32)
        for k0 in range(0, K, kb):
            for a, b in terms:
                p = a[:, k0:k0+kb].astype(np.float64) @ b[:, k0:k0+kb].astype(np.float64).T
                acc = (acc.astype(np.float64) + p).astype(np.float32)
        return acc
    out = ["K %4d fp32 %.2e" % (K, rms(s))]
    for sx, sw in ((1, 1), (16, 256), (256, 256), (256, 4096)):
        xs, ws = x * sx, w * sw
        xh = f16(xs); xl = f16(xs - xh); wh = f16(ws); wl = f16(ws - wh)
        y = mfma_acc([(xl, wh), (xh, wl), (xh, wh)]) / np.float32(sx * sw)
        out.append("s(%d,%d) %.2e" % (sx, sw, rms(y)))
        # flush-to-zero of fp16 subnormals
        ftz = lambda a: np.where(np.abs(a) < 6.1035e-5, 0, a).astype(np.float32)
        y2 = mfma_acc([(ftz(xl), ftz(wh)), (ftz(xh), ftz(wl)), (ftz(xh), ftz(wh))]) / np.float32(sx * sw)
        out.append("ftz %.2e" % rms(y2))
    print(" | ".join(out))
