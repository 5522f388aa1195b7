"""GPU parity of the individual HIP kernels against plain torch CPU references (fp64 where the
op is a contraction). Everything goes through the C ABI (capnet._lib / capnet.ops)."""
import ctypes as C
import random

import pytest
import torch

import capnet
from capnet import ops
from capnet._lib import check, current_stream, lib, ptr, ptr_array, int_array

pytestmark = pytest.mark.gpu


def rel_err(a, b):
    a = a.double().cpu()
    b = b.double().cpu()
    return ((a - b).abs().max() / (b.abs().max() + 1e-30)).item()


@pytest.mark.parametrize("tile", [64, 128])
@pytest.mark.parametrize("ta,tb", [(False, True), (False, False), (True, False), (True, True)])
@pytest.mark.parametrize("M,N,K", [(64, 64, 32), (130, 70, 50), (257, 300, 301), (33, 2048, 512),
                                   (1037, 512, 300)])
def test_sgemm_layouts(dev, tile, ta, tb, M, N, K):
    g = torch.Generator().manual_seed(M * 7 + N * 3 + K)
    A = torch.randn((K, M) if ta else (M, K), generator=g)
    B = torch.randn((N, K) if tb else (K, N), generator=g)
    bias = torch.randn(N, generator=g)
    ref = (A.double().t() if ta else A.double()) @ (B.double().t() if tb else B.double()) + bias.double()
    out = ops.sgemm(A.to(dev), B.to(dev), transA=ta, transB=tb, bias=bias.to(dev), force_tile=tile)
    assert rel_err(out, ref) < 2e-6
    # asymmetric check of orientation: A = I picks rows of op(B)
    acc = torch.ones((M, N), device=dev)
    out2 = ops.sgemm(A.to(dev), B.to(dev), transA=ta, transB=tb, out=acc.clone(), accumulate=True,
                     force_tile=tile)
    assert rel_err(out2, ref - bias.double() + 1.0) < 2e-6


def test_sgemm_batched_strided(dev):
    # the gate-batched S/U products of the FactoredLSTM chain: 4 groups inside [n, 4F] buffers
    n, F, H = 77, 48, 40
    g = torch.Generator().manual_seed(5)
    A1 = torch.randn(n, 4 * F, generator=g)
    S = torch.randn(4, H, F, generator=g)
    b = torch.randn(4 * H, generator=g)
    out = torch.zeros(n, 4 * H, device=dev)
    A1d, Sd, bd = A1.to(dev), S.to(dev), b.to(dev)
    check(lib().capnet_sgemm(0, 1, n, H, F, ptr(A1d), 4 * F, ptr(Sd), F, ptr(out), 4 * H, ptr(bd),
                             0, 4, F, H * F, H, H, 0, current_stream()))
    ref = torch.cat([A1[:, k * F:(k + 1) * F].double() @ S[k].double().t() for k in range(4)], 1) + b.double()
    assert rel_err(out, ref) < 2e-6


def _conv_ref(x_nchw, w, stride, pad, scale=None, shift=None, relu=False):
    x = x_nchw.double()
    if scale is not None:
        x = x * scale.double().view(1, -1, 1, 1) + shift.double().view(1, -1, 1, 1)
        if relu:
            x = x.clamp_min(0)
    return torch.nn.functional.conv2d(x, w.double(), stride=stride, padding=pad)


@pytest.mark.parametrize("tile", [0, 64, 128, 12864])
@pytest.mark.parametrize("B,H,W,Cin,Cout,k,stride,pad,pre", [
    (2, 14, 14, 64, 64, 3, 1, 1, True),
    (3, 9, 11, 32, 96, 3, 2, 1, True),
    (2, 8, 8, 128, 256, 1, 1, 0, False),
    (2, 8, 8, 128, 80, 1, 2, 0, True),
])
def test_conv_fast_path(dev, tile, B, H, W, Cin, Cout, k, stride, pad, pre):
    g = torch.Generator().manual_seed(B + H + Cin + Cout + k)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, k, k, generator=g) * 0.1
    scale = torch.rand(Cin, generator=g) + 0.5 if pre else None
    shift = torch.randn(Cin, generator=g) if pre else None
    ref = _conv_ref(x, w, stride, pad, scale, shift, relu=pre)           # NCHW
    OH, OW = ref.shape[2], ref.shape[3]
    xd = x.permute(0, 2, 3, 1).contiguous().to(dev)                     # NHWC
    Kw = (k * k * Cin + 15) // 16 * 16
    wp = ops.pack_conv_weight(w.to(dev), Kw)
    y = torch.empty(B * OH * OW, Cout, device=dev)
    M = B * OH * OW
    tiles = lib().capnet_conv_tiles_m(M, Cout, tile)
    psum = torch.zeros(tiles, Cout, device=dev)
    psq = torch.zeros(tiles, Cout, device=dev)
    sd = scale.to(dev) if pre else None
    hd = shift.to(dev) if pre else None
    check(lib().capnet_conv2d_fwd(ptr(xd), H * W * Cin, W * Cin, Cin, 1, ptr(wp), Kw, ptr(y),
                                  ptr(sd), ptr(hd), int(pre), ptr(psum), ptr(psq), B, H, W, Cin,
                                  Cout, k, k, stride, pad, tile, current_stream()))
    ref_nhwc = ref.permute(0, 2, 3, 1).reshape(M, Cout)
    assert rel_err(y, ref_nhwc) < 3e-6
    assert rel_err(psum.sum(0), ref_nhwc.sum(0)) < 1e-5
    assert rel_err(psq.sum(0), (ref_nhwc ** 2).sum(0)) < 1e-5


@pytest.mark.parametrize("tile", [0, 64, 128, 12864])
@pytest.mark.parametrize("B,H,W,Cin,Cout,k,stride,pad,pre", [
    (2, 14, 14, 64, 64, 3, 1, 1, True),
    (3, 9, 11, 32, 128, 3, 2, 1, True),
    (2, 8, 8, 128, 256, 1, 1, 0, False),      # no prologue, M = 128
    (3, 7, 7, 128, 128, 1, 1, 0, False),      # no prologue, ragged last M tile (147 rows)
    (2, 8, 8, 128, 64, 1, 2, 0, True),
    (2, 6, 6, 48, 192, 3, 1, 1, True),        # 3 k-tiles per tap, odd tile count (27)
    (23, 14, 14, 64, 256, 3, 1, 1, True),     # 4508 rows: tiles past a multiple of 256 -> K-sliced tail
    (5, 28, 28, 64, 128, 1, 1, 0, False),     # 3920 rows, tail tiles without prologue
])
def test_conv_kmajor_lds_dma_path(dev, tile, B, H, W, Cin, Cout, k, stride, pad, pre):
    """conv_f32_v2: K-major weights by LDS-DMA, saddr loads, med3 mask -- the kernel every trunk
    convolution except the stem runs on."""
    g = torch.Generator().manual_seed(B + H + Cin + Cout + k + 1)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, k, k, generator=g) * 0.1
    scale = torch.rand(Cin, generator=g) - 0.3 if pre else None      # some negative scales
    shift = torch.randn(Cin, generator=g) if pre else None
    ref = _conv_ref(x, w, stride, pad, scale, shift, relu=pre)
    OH, OW = ref.shape[2], ref.shape[3]
    xd = x.permute(0, 2, 3, 1).contiguous().to(dev)
    Kw = k * k * Cin
    wk = ops.pack_conv_weight(w.to(dev), Kw, kmajor=True)
    M = B * OH * OW
    y = torch.full((M, Cout), float("nan"), device=dev)
    tiles = lib().capnet_conv_kmajor_tiles_m(M, Cout, Kw, tile)
    psum = torch.zeros(tiles, Cout, device=dev)
    psq = torch.zeros(tiles, Cout, device=dev)
    sd = scale.to(dev) if pre else None
    hd = shift.to(dev) if pre else None
    slabs = torch.empty(max(1, lib().capnet_conv_kmajor_slab_floats(M, Cout, Kw, tile)), device=dev)
    check(lib().capnet_conv2d_fwd_kmajor(ptr(xd), H * W * Cin, W * Cin, Cin, ptr(wk), Kw, ptr(y),
                                         ptr(sd), ptr(hd), int(pre), ptr(psum), ptr(psq), B, H, W,
                                         Cin, Cout, k, k, stride, pad, tile, ptr(slabs),
                                         current_stream()))
    ref_nhwc = ref.permute(0, 2, 3, 1).reshape(M, Cout)
    assert rel_err(y, ref_nhwc) < 3e-6
    assert rel_err(psum.sum(0), ref_nhwc.sum(0)) < 1e-5
    assert rel_err(psq.sum(0), (ref_nhwc ** 2).sum(0)) < 1e-5


def test_conv_stem_generic_nchw(dev):
    B, H, W = 2, 64, 64
    g = torch.Generator().manual_seed(11)
    x = torch.randn(B, 3, H, W, generator=g)
    w = torch.randn(64, 3, 7, 7, generator=g) * 0.1
    ref = _conv_ref(x, w, 2, 3)
    OH, OW = ref.shape[2], ref.shape[3]
    Kw = 160
    wp = ops.pack_conv_weight(w.to(dev), Kw)
    xd = x.to(dev)
    y = torch.empty(B * OH * OW, 64, device=dev)
    check(lib().capnet_conv2d_fwd(ptr(xd), 3 * H * W, W, 1, H * W, ptr(wp), Kw, ptr(y), None, None,
                                  0, None, None, B, H, W, 3, 64, 7, 7, 2, 3, 0, current_stream()))
    assert rel_err(y, ref.permute(0, 2, 3, 1).reshape(-1, 64)) < 3e-6


def test_bn_finalize_and_tails(dev):
    g = torch.Generator().manual_seed(3)
    rows, Cc = 1000, 96
    y = torch.randn(rows, Cc, generator=g) * 2 + 0.7
    tiles = 8
    part = y.view(tiles, rows // tiles, Cc)
    psum, psq = part.sum(1).to(dev), (part ** 2).sum(1).to(dev)
    gamma, beta = torch.rand(Cc, generator=g) + 0.5, torch.randn(Cc, generator=g)
    rm, rv = torch.zeros(Cc), torch.ones(Cc)
    bn = torch.nn.BatchNorm1d(Cc)
    with torch.no_grad():
        bn.weight.copy_(gamma); bn.bias.copy_(beta)
    bn.train()
    ref = bn(y)
    scale, shift = torch.empty(Cc, device=dev), torch.empty(Cc, device=dev)
    rmd, rvd = rm.to(dev), rv.to(dev)
    gd, bd, yd = gamma.to(dev), beta.to(dev), y.to(dev)   # keep device tensors alive across calls
    check(lib().capnet_bn_finalize(ptr(psum), ptr(psq), tiles, Cc, rows, ptr(gd), ptr(bd),
                                   ptr(rmd), ptr(rvd), 0.1, 1e-5, ptr(scale), ptr(shift),
                                   current_stream()))
    out = yd * scale + shift
    assert rel_err(out, ref) < 2e-5
    assert rel_err(rmd, bn.running_mean) < 1e-5
    assert rel_err(rvd, bn.running_var) < 1e-5
    # bottleneck tail
    res = torch.randn(rows, Cc, generator=g)
    o = torch.empty(rows, Cc, device=dev)
    resd = res.to(dev)
    check(lib().capnet_bn_add_relu(ptr(yd), ptr(scale), ptr(shift), ptr(resd), None,
                                   None, ptr(o), rows, Cc, current_stream()))
    assert rel_err(o, (ref + res).clamp_min(0)) < 2e-5
    check(lib().capnet_bn_add_relu(ptr(yd), ptr(scale), ptr(shift), ptr(resd),
                                   ptr(scale), ptr(shift), ptr(o), rows, Cc, current_stream()))
    ref2 = (ref + res * scale.cpu() + shift.cpu()).clamp_min(0)
    assert rel_err(o, ref2) < 2e-5


def test_maxpool_avgpool_upsample(dev):
    g = torch.Generator().manual_seed(4)
    B, H, W, Cc = 2, 12, 12, 64
    y = torch.randn(B, H, W, Cc, generator=g)
    sc, sh = torch.rand(Cc, generator=g) - 0.3, torch.randn(Cc, generator=g)   # some negative scales
    ref = torch.nn.functional.max_pool2d(((y * sc + sh).clamp_min(0)).permute(0, 3, 1, 2), 3, 2, 1)
    out = torch.empty(B, 6, 6, Cc, device=dev)
    yd, scd, shd = y.to(dev), sc.to(dev), sh.to(dev)
    check(lib().capnet_bn_relu_maxpool(ptr(yd), ptr(scd), ptr(shd), ptr(out),
                                       B, H, W, Cc, current_stream()))
    assert rel_err(out, ref.permute(0, 2, 3, 1)) < 1e-6
    x = torch.randn(B, 49, 128, generator=g)
    o = torch.empty(B, 128, device=dev)
    xd = x.to(dev)
    check(lib().capnet_global_avgpool(ptr(xd), ptr(o), B, 49, 128, current_stream()))
    assert rel_err(o, x.mean(1)) < 1e-6
    for b2, hw, c2 in ((12, 196, 2048), (3, 1, 260), (5, 33, 4)):          # the attention map; one pixel; one channel quad
        x = torch.randn(b2, hw, c2, generator=g)
        o = torch.full((b2, c2), 7.0, device=dev)
        xd = x.to(dev)
        check(lib().capnet_global_avgpool(ptr(xd), ptr(o), b2, hw, c2, current_stream()))
        assert rel_err(o, x.double().mean(1)) < 1e-6
    m = torch.randn(B, 7, 7, 32, generator=g)
    up = torch.empty(B, 14, 14, 32, device=dev)
    md = m.to(dev)
    check(lib().capnet_adaptive_pool_replicate(ptr(md), ptr(up), B, 7, 14, 32, current_stream()))
    ref = torch.nn.functional.adaptive_avg_pool2d(m.permute(0, 3, 1, 2), 14).permute(0, 2, 3, 1)
    assert rel_err(up, ref) < 1e-6


def test_linear_xent_autograd(dev):
    g = torch.Generator().manual_seed(9)
    N, H, V = 37, 48, 211
    x = torch.randn(N, H, generator=g)
    w = torch.randn(V, H, generator=g) * 0.2
    b = torch.randn(V, generator=g) * 0.1
    t = torch.randint(0, V, (N,), generator=g)
    xr, wr, br = [v.clone().double().requires_grad_() for v in (x, w, b)]
    loss_ref = torch.nn.functional.cross_entropy(xr @ wr.t() + br, t)
    loss_ref.backward()
    xd, wd, bd = [v.to(dev).requires_grad_() for v in (x, w, b)]
    loss = ops.cross_entropy(ops.linear(xd, wd, bd), t.to(dev))
    loss.backward()
    assert abs(loss.item() - loss_ref.item()) / loss_ref.item() < 1e-6
    assert rel_err(xd.grad, xr.grad) < 1e-5
    assert rel_err(wd.grad, wr.grad) < 1e-5
    assert rel_err(bd.grad, br.grad) < 1e-5
    ops.check_device_errors()


def test_bn1d_autograd(dev):
    g = torch.Generator().manual_seed(10)
    B, Cc = 16, 50
    x = torch.randn(B, Cc, generator=g) * 3 + 1
    bn = torch.nn.BatchNorm1d(Cc, momentum=0.01).double()
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5); bn.bias.normal_()
    xr = x.clone().double().requires_grad_()
    dy = torch.randn(B, Cc, generator=g)
    yr = bn(xr)
    yr.backward(dy.double())
    gam, bet = bn.weight.detach().float().to(dev).requires_grad_(), bn.bias.detach().float().to(dev).requires_grad_()
    rm, rv = torch.zeros(Cc, device=dev), torch.ones(Cc, device=dev)
    xd = x.to(dev).requires_grad_()
    y = ops.batch_norm1d(xd, gam, bet, rm, rv, True, 0.01, 1e-5)
    y.backward(dy.to(dev))
    assert rel_err(y, yr) < 1e-5
    assert rel_err(xd.grad, xr.grad) < 1e-4
    assert rel_err(gam.grad, bn.weight.grad) < 1e-5
    assert rel_err(bet.grad, bn.bias.grad) < 1e-5
    assert rel_err(rm, bn.running_mean) < 1e-5 and rel_err(rv, bn.running_var) < 1e-5


def test_clamp_adam_matches_torch(dev):
    g = torch.Generator().manual_seed(12)
    shapes = [(300, 17), (5,), (2049,), (64, 64)]
    ps = [torch.randn(s, generator=g) for s in shapes]
    ref_p = [p.clone().requires_grad_() for p in ps]
    opt = torch.optim.Adam(ref_p, lr=2e-4, betas=(0.9, 0.999), eps=1e-8)
    dp = [p.to(dev) for p in ps]
    m = [torch.zeros_like(p) for p in dp]
    v = [torch.zeros_like(p) for p in dp]
    for step in range(1, 4):
        grads = [torch.randn(s, generator=g) * 2 for s in shapes]
        for p, gr in zip(ref_p, grads):
            p.grad = gr.clone().clamp_(-0.5, 0.5)
        opt.step()
        dg = [gr.to(dev) for gr in grads]
        ops.clamp_adam(dp, dg, m, v, [step] * len(dp), 2e-4, 0.9, 0.999, 1e-8, 0.5)
        for a, b_, gg, gr in zip(dp, ref_p, dg, grads):
            assert (a.cpu() - b_.detach()).abs().max().item() < 5e-7   # 1-2 ulp of O(1) values
            assert torch.equal(gg.cpu(), gr.clamp(-0.5, 0.5))


def test_argmax_first_max(dev):
    x = torch.zeros(5, 1000)
    x[0, 7] = 1; x[0, 900] = 1
    x[1, 999] = 3
    x[2] = -1; x[2, 0] = -0.5
    x[3, 255] = 2; x[3, 256] = 2
    out = ops.argmax_rows(x.to(dev)).cpu().tolist()
    assert out == [7, 999, 0, 255, 0]
    g = torch.Generator().manual_seed(3)
    for rows, V in ((12, 8192), (7, 7411), (3, 5), (2, 2051)):             # 16-B rows; ragged vocabularies (scalar tail / scalar path)
        y = torch.randint(0, 50, (rows, V), generator=g).float()          # many ties: the first maximum counts
        assert ops.argmax_rows(y.to(dev)).cpu().tolist() == y.argmax(1).tolist() or \
            ops.argmax_rows(y.to(dev)).cpu().tolist() == [int((r == r.max()).nonzero()[0]) for r in y]
        z = torch.randn(rows, V, generator=g)
        assert ops.argmax_rows(z.to(dev)).cpu().tolist() == z.argmax(1).tolist()


# ---- per-step (skinny) product: one K chunk per workgroup + fixed-order slab sum -----------------
@pytest.mark.parametrize("tb", [True, False])
@pytest.mark.parametrize("M,N,K", [(64, 2048, 512), (64, 512, 2048), (9, 4608, 512), (64, 8192, 512),
                                   (33, 300, 2048), (64, 2348, 2048), (96, 512, 512), (1, 64, 64),
                                   (64, 512, 100), (5, 36, 72), (1037, 512, 8192), (1037, 300, 2048)])
def test_sgemm_splitk_matches_fp64(dev, M, N, K, tb):
    from capnet._lib import lib, check, ptr, current_stream
    g = torch.Generator().manual_seed(M * 7 + N + K)
    A = torch.randn(M, K, generator=g)
    B = torch.randn(N, K, generator=g) if tb else torch.randn(K, N, generator=g)
    bias = torch.randn(N, generator=g)
    C0 = torch.randn(M, N, generator=g)
    ref = A.double() @ (B.double().t() if tb else B.double()) + bias.double() + C0.double()
    Ad, Bd, bd, Cd = A.to(dev), B.to(dev), bias.to(dev), C0.to(dev)
    ws = torch.empty(32 * 64 * 4608, device=dev)
    check(lib().capnet_sgemm_splitk(0, int(tb), M, N, K, ptr(Ad), K, ptr(Bd), Bd.shape[1], ptr(Cd), N,
                                    ptr(bd), 1, ptr(ws), ws.numel(), current_stream()))
    err = (Cd.cpu().double() - ref).abs().max().item() / ref.abs().max().item()
    assert err < 2e-6, err
    # no workspace: falls back to the generic kernel, same answer
    Cd2 = C0.to(dev)
    check(lib().capnet_sgemm_splitk(0, int(tb), M, N, K, ptr(Ad), K, ptr(Bd), Bd.shape[1], ptr(Cd2), N,
                                    ptr(bd), 1, None, 0, current_stream()))
    # (one fp32 accumulator over all of K: looser than the chunked sum above)
    assert (Cd2.cpu().double() - ref).abs().max().item() / ref.abs().max().item() < 6e-6


# ---- the same for steps of <= 16 rows: one launch, the last workgroup of a tile sums the K chunks ---------
@pytest.mark.parametrize("tb", [True, False])
@pytest.mark.parametrize("M,N,K", [(12, 4608, 512), (12, 2048, 2348), (12, 512, 4608), (12, 2348, 2048), (16, 8192, 512),
                                   (1, 64, 64), (5, 36, 72), (3, 300, 1000), (12, 512, 256), (9, 2048, 9216)])
def test_sgemm_rows16_one_launch_matches_fp64_and_is_reproducible(dev, M, N, K, tb):
    """gemm_rows16_kernel (csrc/gemm_f32.hip): every step product of a 12-image attention batch
    (model_att.py:59-60,196-236,283 at b = 12), against float64; the result does not depend on which workgroup
    arrives last (20 repeats are bit-identical, also with other work on the device), the counters are left zero,
    bias and accumulation as in the two-launch path."""
    from capnet._lib import lib, check, ptr, current_stream
    g = torch.Generator().manual_seed(M * 7 + N + K)
    A = torch.randn(M, K, generator=g)
    B = torch.randn(N, K, generator=g) if tb else torch.randn(K, N, generator=g)
    bias = torch.randn(N, generator=g)
    C0 = torch.randn(M, N, generator=g)
    ref = A.double() @ (B.double().t() if tb else B.double()) + bias.double() + C0.double()
    Ad, Bd, bd = A.to(dev), B.to(dev), bias.to(dev)
    ws = torch.empty(40 * 16 * 8192, device=dev)
    ctr = torch.zeros(1024, dtype=torch.int32, device=dev)
    noise = torch.randn(4096, 4096, device=dev)
    side = torch.cuda.Stream()
    outs = []
    for rep in range(20):
        Cd = C0.to(dev)
        if rep >= 10:                                  # uneven load beside it
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                noise @ noise
        check(lib().capnet_sgemm_splitk_fused(0, int(tb), M, N, K, ptr(Ad), K, ptr(Bd), Bd.shape[1], ptr(Cd), N,
                                              ptr(bd), 1, ptr(ws), ws.numel(), ptr(ctr), ctr.numel(), current_stream()))
        outs.append(Cd)
    torch.cuda.synchronize()
    assert int(ctr.abs().sum().item()) == 0
    err = (outs[0].cpu().double() - ref).abs().max().item() / ref.abs().max().item()
    assert err < 2e-6, err
    for o in outs[1:]:
        assert torch.equal(o, outs[0])
    # without bias, without accumulation
    Cd = torch.full((M, N), float("nan"), device=dev)
    check(lib().capnet_sgemm_splitk_fused(0, int(tb), M, N, K, ptr(Ad), K, ptr(Bd), Bd.shape[1], ptr(Cd), N,
                                          None, 0, ptr(ws), ws.numel(), ptr(ctr), ctr.numel(), current_stream()))
    ref0 = A.double() @ (B.double().t() if tb else B.double())
    assert (Cd.cpu().double() - ref0).abs().max().item() / ref0.abs().max().item() < 2e-6


# ---- LDS-DMA NT GEMM / 1x1 convolution core (csrc/gemm_dma.hip) --------------------------------
@pytest.mark.parametrize("M,N,K,bias", [(999, 8192, 512, True),     # the vocabulary projection of configs[1]
                                        (128, 128, 32, False),       # one tile, one k-tile
                                        (130, 64, 64, True),         # ragged second M tile, 128x64 tiles
                                        (257, 192, 96, False),       # 64-wide N tiles, odd k-tile count
                                        (64, 256, 2048, True), (1, 64, 32, True)])
def test_sgemm_nt_dma_matches_fp64(dev, M, N, K, bias):
    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randn(M, K, generator=g)
    Bw = torch.randn(N, K, generator=g) * 0.1
    b = torch.randn(N, generator=g) if bias else None
    ref = A.double() @ Bw.double().t() + (b.double() if bias else 0)
    Ad, Bd = A.to(dev), Bw.to(dev)
    bd = b.to(dev) if bias else None
    out = torch.full((M, N), float("nan"), device=dev)
    assert lib().capnet_sgemm_nt_dma_eligible(M, N, K, ptr(Ad), K, ptr(Bd), K, ptr(out), N) == 1
    check(lib().capnet_sgemm_nt_dma(M, N, K, ptr(Ad), K, ptr(Bd), ptr(out), ptr(bd), current_stream()))
    assert rel_err(out, ref) < 3e-6
    # capnet_sgemm (ops.linear) takes this path by itself once there are enough 128-row tiles (and gemm_b3.hip beyond 2.5e8
    # multiply-adds)
    out2 = ops.linear(Ad, Bd, bd)
    if M > 64 and ((M + 127) // 128) * (N // 64) >= 64 and (M * N * K < 2.5e8 or ((M + 127) // 128) * ((N + 127) // 128) < 192):
        assert torch.equal(out2, out)
    else:
        assert rel_err(out2, ref) < 3e-6


@pytest.mark.parametrize("ta,tb", [(0, 1), (0, 0), (1, 0), (1, 1)])
@pytest.mark.parametrize("M,N,K,batch,split", [
    (1344, 8192, 512, 1, False),       # the vocabulary projection's family at configs[1]'s row count
    (1036, 512, 8192, 1, True),        # few tiles, long K: cut over the chip, partials summed by reduce_slabs
    (260, 132, 300, 1, False),         # ragged tiles, K = 9 whole steps + 12 (the embedding width)
    (128, 128, 32, 1, False),          # one tile, one step
    (4, 4, 4, 1, False),               # the smallest eligible product
    (516, 2348, 2048, 1, True),        # the attention cell's input width as N
    (200, 256, 96, 4, False),          # batched (the four gates)
])
def test_sgemm_b3_matches_fp64(dev, ta, tb, M, N, K, batch, split):
    """csrc/gemm_b3.hip: three bf16 pieces per operand, six products -- fp32-grade against float64 on every layout, with
    bias and accumulation, over leading dimensions wider than the matrices, and with operands of 1e-9 and 1e+5 (bf16 pieces
    have fp32's range: no prescale)."""
    g = torch.Generator().manual_seed(M + N + K + 7 * ta + 3 * tb)
    pad = 4
    Ash, Bsh = ((K, M) if ta else (M, K)), ((N, K) if tb else (K, N))
    A = torch.randn(batch, Ash[0], Ash[1] + pad, generator=g)
    B = torch.randn(batch, Bsh[0], Bsh[1] + pad, generator=g)
    A[..., :, : Ash[1] // 2] *= 1e-9                      # (half of each operand's columns tiny, a few entries huge)
    B.view(-1)[::97] *= 3e4
    bias = torch.randn(batch, N, generator=g)
    C0 = torch.randn(batch, M, N + pad, generator=g)
    Av, Bv = A[..., : Ash[1]].double(), B[..., : Bsh[1]].double()
    ref = (Av.transpose(1, 2) if ta else Av) @ (Bv.transpose(1, 2) if tb else Bv) + bias.double()[:, None, :] + C0[..., :N].double()
    Ad, Bd, bd, Cd = A.to(dev), B.to(dev), bias.to(dev), C0.to(dev)
    ws = torch.empty(8 << 20, device=dev) if split else None
    L = lib()
    assert L.capnet_sgemm_b3_eligible(ta, tb, M, N, K, ptr(Ad), Ash[1] + pad, ptr(Bd), Bsh[1] + pad, ptr(Cd), N + pad, ptr(bd),
                                      batch, Ad.stride(0), Bd.stride(0), Cd.stride(0), N) == 1
    check(L.capnet_sgemm_b3(ta, tb, M, N, K, ptr(Ad), Ash[1] + pad, ptr(Bd), Bsh[1] + pad, ptr(Cd), N + pad, ptr(bd), 1, batch,
                            Ad.stride(0), Bd.stride(0), Cd.stride(0), N, ptr(ws), ws.numel() if split else 0, current_stream()))
    torch.cuda.synchronize()
    got = Cd[..., :N].double().cpu()
    assert torch.equal(Cd[..., N:].cpu(), C0[..., N:])                     # nothing written past a row
    # error against the sum of absolute products (what fp32 accumulation is held to), per element
    mag = ((Av.abs().transpose(1, 2) if ta else Av.abs()) @ (Bv.abs().transpose(1, 2) if tb else Bv.abs()) +
           bias.double().abs()[:, None, :] + C0[..., :N].double().abs())
    assert float(((got - ref).abs() / mag).max()) < 2e-6


def test_sgemm_b3_rejects_ineligible(dev):
    L = lib()
    A, B, Cc = torch.zeros(8, 42, device=dev), torch.zeros(16, 42, device=dev), torch.zeros(8, 16, device=dev)
    # K = 42 is no multiple of 4 and both operands are K-contiguous
    assert L.capnet_sgemm_b3_eligible(0, 1, 8, 16, 42, ptr(A), 42, ptr(B), 42, ptr(Cc), 16, None, 1, 0, 0, 0, 0) == 0
    assert L.capnet_sgemm_b3(0, 1, 8, 16, 42, ptr(A), 42, ptr(B), 42, ptr(Cc), 16, None, 0, 1, 0, 0, 0, 0, None, 0,
                             current_stream()) != 0


def test_sgemm_nt_dma_rejects_ineligible(dev):
    A, Bw, out = torch.zeros(8, 40, device=dev), torch.zeros(70, 40, device=dev), torch.zeros(8, 70, device=dev)
    assert lib().capnet_sgemm_nt_dma_eligible(8, 70, 40, ptr(A), 40, ptr(Bw), 40, ptr(out), 70) == 0
    assert lib().capnet_sgemm_nt_dma(8, 70, 40, ptr(A), 40, ptr(Bw), ptr(out), None, current_stream()) != 0


@pytest.mark.parametrize("B,H,W,Cin,Cout,stride,epi", [
    (2, 8, 8, 128, 256, 1, 0),        # M = 128: one row of tiles
    (3, 7, 7, 128, 128, 1, 0),        # ragged last M tile (147 rows): statistics ignore the padding rows
    (2, 8, 8, 64, 64, 2, 0),          # stride 2 (downsample branch), 128x64 tiles
    (5, 14, 14, 256, 64, 1, 0),
    (4, 14, 14, 1024, 256, 1, 0),     # stage-3 conv1 shape at batch 4
    (3, 6, 10, 96, 192, 2, 0),        # non-square map, odd k-tile count
    (2, 7, 7, 64, 128, 1, 1),         # folded BatchNorm + residual + ReLU epilogue (inference trunk)
    (2, 8, 8, 32, 64, 2, 2),          # folded epilogue without residual / ReLU
])
def test_conv1x1_dma_path(dev, B, H, W, Cin, Cout, stride, epi):
    g = torch.Generator().manual_seed(B + H + Cin + Cout + stride)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 1, 1, generator=g) * 0.1
    ref = torch.nn.functional.conv2d(x.double(), w.double(), stride=stride)
    OH, OW = ref.shape[2], ref.shape[3]
    M = B * OH * OW
    ref = ref.permute(0, 2, 3, 1).reshape(M, Cout)
    xd = x.permute(0, 2, 3, 1).contiguous().to(dev)
    wd = w.reshape(Cout, Cin).contiguous().to(dev)
    y = torch.full((M, Cout), float("nan"), device=dev)
    tiles = lib().capnet_conv1x1_tiles_m(M)
    assert tiles == (M + 127) // 128
    if epi == 0:
        psum = torch.full((tiles, Cout), float("nan"), device=dev)
        psq = torch.full((tiles, Cout), float("nan"), device=dev)
        check(lib().capnet_conv1x1_fwd_dma(ptr(xd), H * W * Cin, W * Cin, Cin, ptr(wd), ptr(y), ptr(psum), ptr(psq),
                                           B, H, W, Cin, Cout, stride, None, None, None, 0, current_stream()))
        assert rel_err(y, ref) < 3e-6
        assert rel_err(psum.sum(0), ref.sum(0)) < 1e-5
        assert rel_err(psq.sum(0), (ref ** 2).sum(0)) < 1e-5
        return
    sc = (torch.rand(Cout, generator=g) + 0.5)
    sh = torch.randn(Cout, generator=g)
    res = torch.randn(M, Cout, generator=g) if epi == 1 else None
    want = ref * sc.double() + sh.double()
    if res is not None:
        want = torch.relu(want + res.double())
    rd = res.to(dev) if res is not None else None
    scd, shd = sc.to(dev), sh.to(dev)
    check(lib().capnet_conv1x1_fwd_dma(ptr(xd), H * W * Cin, W * Cin, Cin, ptr(wd), ptr(y), None, None, B, H, W,
                                       Cin, Cout, stride, ptr(scd), ptr(shd), ptr(rd),
                                       1 if epi == 1 else 0, current_stream()))
    assert rel_err(y, want) < 3e-6


def test_attention_loss_matches_torch(dev):
    """ops.attention_loss = nll + alpha_c * ((1 - alphas.sum(dim=1)) ** 2).mean()
    (stylenet/train_multitask_att.py:409-411), value and both gradients."""
    g = torch.Generator().manual_seed(3)
    B, steps, P = 5, 7, 196
    alphas = torch.rand(B, steps, P, generator=g) * 0.3
    alphas[3:, 5:] = 0.0                       # rows of finished sequences stay zero and still count
    nll = torch.tensor(3.25)
    a_ref = alphas.double().requires_grad_(True)
    n_ref = nll.double().requires_grad_(True)
    want = n_ref + 0.7 * ((1.0 - a_ref.sum(dim=1)) ** 2).mean()
    (want * 1.5).backward()
    a_d = alphas.to(dev).requires_grad_(True)
    n_d = nll.to(dev).requires_grad_(True)
    got = ops.attention_loss(n_d, a_d, 0.7)
    got.backward(torch.tensor(1.5, device=dev))
    assert abs(got.item() - want.item()) < 1e-6 * abs(want.item())
    assert rel_err(a_d.grad, a_ref.grad) < 1e-6
    assert abs(n_d.grad.item() - 1.5) < 1e-7


# ---- 1x1 convolutions on split operands: one case against fp64 and the f32-MFMA kernel ----
def _split_conv1x1_case(dev, kind, bn, B, H, W, Cin, Cout, stride, pre, epi):
    g = torch.Generator().manual_seed(B + H + Cin + Cout + stride + bn)
    # values spread over many binades, as activations and weights are
    x = torch.randn(B, Cin, H, W, generator=g) * torch.exp(torch.randn(B, Cin, H, W, generator=g))
    w = torch.randn(Cout, Cin, 1, 1, generator=g) * 0.1 * torch.exp(torch.randn(Cout, Cin, 1, 1, generator=g))
    scale = torch.rand(Cin, generator=g) - 0.3 if pre else None
    shift = torch.randn(Cin, generator=g) if pre else None
    xin = x.double()
    if pre:
        xin = torch.relu(x * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)).double()   # the fold is an fp32 fma + max
    ref = torch.nn.functional.conv2d(xin, w.double(), stride=stride)
    OH, OW = ref.shape[2], ref.shape[3]
    M = B * OH * OW
    ref = ref.permute(0, 2, 3, 1).reshape(M, Cout)
    xd = x.permute(0, 2, 3, 1).contiguous().to(dev)
    wd = w.reshape(Cout, Cin).contiguous().to(dev)
    L = lib()
    img = torch.empty(getattr(L, 'capnet_conv1x1_%s_weight_words' % kind)(Cin, Cout), dtype=torch.int32, device=dev)
    check(getattr(L, 'capnet_conv1x1_%s_pack' % kind)(ptr(wd), ptr(img), Cout, Cin, bn, current_stream()))
    y = torch.full((M, Cout), float("nan"), device=dev)
    tiles = L.capnet_conv1x1_tiles_m(M)
    sd, hd = (scale.to(dev), shift.to(dev)) if pre else (None, None)
    if epi:
        sc, sh = torch.rand(Cout, generator=g) + 0.5, torch.randn(Cout, generator=g)
        res = torch.randn(M, Cout, generator=g)
        scd, shd, rd = sc.to(dev), sh.to(dev), res.to(dev)
        check(getattr(L, 'capnet_conv1x1_fwd_%s' % kind)(ptr(xd), H * W * Cin, W * Cin, Cin, ptr(img), bn, ptr(y), ptr(sd), ptr(hd),
                                          int(pre), None, None, B, H, W, Cin, Cout, stride, ptr(scd), ptr(shd),
                                          ptr(rd), 1, current_stream()))
        want = torch.relu(ref * sc.double() + sh.double() + res.double())
        assert rel_err(y, want) < 3e-6
        return
    psum = torch.full((tiles, Cout), float("nan"), device=dev)
    psq = torch.full((tiles, Cout), float("nan"), device=dev)
    check(getattr(L, 'capnet_conv1x1_fwd_%s' % kind)(ptr(xd), H * W * Cin, W * Cin, Cin, ptr(img), bn, ptr(y), ptr(sd), ptr(hd),
                                      int(pre), ptr(psum), ptr(psq), B, H, W, Cin, Cout, stride, None, None, None, 0,
                                      current_stream()))
    assert rel_err(y, ref) < 3e-6                      # the bound the f32-MFMA conv kernels are held to
    # fp32-grade: rms error against fp64 not above that of the f32-MFMA kernel on the same operands
    # (an fp32 dot product of these lengths and this dynamic range sits at 1e-7 .. 6e-7)
    rms = lambda t: (((t.double().cpu() - ref) ** 2).mean().sqrt() / (ref ** 2).mean().sqrt()).item()
    wk = ops.pack_conv_weight(w.to(dev), Cin, kmajor=True)
    y32 = torch.empty((M, Cout), device=dev)
    t32 = lib().capnet_conv_kmajor_tiles_m(M, Cout, Cin, 0)
    p1, p2 = torch.empty(t32, Cout, device=dev), torch.empty(t32, Cout, device=dev)
    check(lib().capnet_conv2d_fwd_kmajor(ptr(xd), H * W * Cin, W * Cin, Cin, ptr(wk), Cin, ptr(y32), ptr(sd), ptr(hd),
                                         int(pre), ptr(p1), ptr(p2), B, H, W, Cin, Cout, 1, 1, stride, 0, 0, None,
                                         current_stream()))
    print("rms vs fp64: %s %.2e, f32 MFMA %.2e" % (kind, rms(y), rms(y32)))
    assert rms(y) < 1.25 * rms(y32) + 2e-8
    assert rel_err(psum.sum(0), ref.sum(0)) < 1e-5
    assert rel_err(psq.sum(0), (ref ** 2).sum(0)) < 1e-5



# ---- 1x1 convolution as three f16 products of 2-way split, power-of-two scaled operands (csrc/conv_f16x3.hip) ----
@pytest.mark.parametrize("bn", [64, 128])
@pytest.mark.parametrize("B,H,W,Cin,Cout,stride,pre,epi", [
    (2, 8, 8, 128, 256, 1, False, 0),       # M = 128, 4 k-tiles
    (3, 7, 7, 128, 128, 1, True, 0),        # ragged last M tile (147 rows), folded BatchNorm + ReLU on load
    (2, 8, 8, 64, 128, 2, False, 0),        # stride 2 (downsample branch), one pair of k-tiles
    (4, 14, 14, 1024, 256, 1, False, 0),    # stage-3 conv1 at batch 4: 32 k-tiles
    (4, 14, 14, 256, 1024, 1, True, 0),     # stage-3 conv3
    (9, 14, 14, 512, 128, 1, True, 0),      # 14 M tiles x 1 / 2 N tiles: workgroups walking several tiles need more
    (2, 7, 7, 64, 128, 1, False, 1),        # folded inference epilogue with residual + ReLU
])
def test_conv1x1_f16x3_has_fp32_accuracy(dev, bn, B, H, W, Cin, Cout, stride, pre, epi):
    _split_conv1x1_case(dev, "f16x3", bn, B, H, W, Cin, Cout, stride, pre, epi)


def test_conv1x1_f16x3_walks_several_tiles_per_workgroup(dev):
    """The persistent loop with more tiles than its 512 workgroups (392 row tiles x 2 column tiles at batch 16 of the
    56 x 56 maps): cross-tile prefetch, the counted waits behind an epilogue and the accumulator reset. Against fp64,
    statistics included, and bit-identical from launch to launch."""
    B, H, W, Cin, Cout, bn = 16, 56, 56, 64, 256, 128
    g = torch.Generator().manual_seed(5)
    x = torch.randn(B, H, W, Cin, generator=g); w = torch.randn(Cout, Cin, generator=g) * 0.05
    sc, sh = torch.rand(Cin, generator=g) + 0.5, torch.randn(Cin, generator=g)
    M = B * H * W
    ref = torch.relu(x.reshape(M, Cin) * sc + sh).double() @ w.double().t()
    xd, wd, scd, shd = x.to(dev), w.to(dev), sc.to(dev), sh.to(dev)
    L = lib()
    img = torch.empty(L.capnet_conv1x1_f16x3_weight_words(Cin, Cout), dtype=torch.int32, device=dev)
    check(L.capnet_conv1x1_f16x3_pack(ptr(wd), ptr(img), Cout, Cin, bn, current_stream()))
    t = L.capnet_conv1x1_tiles_m(M)
    assert t * (Cout // bn) > 512
    outs = []
    for _ in range(2):
        y = torch.full((M, Cout), float("nan"), device=dev)
        ps = torch.full((t, Cout), float("nan"), device=dev); pq = torch.full((t, Cout), float("nan"), device=dev)
        check(L.capnet_conv1x1_fwd_f16x3(ptr(xd), H * W * Cin, W * Cin, Cin, ptr(img), bn, ptr(y), ptr(scd), ptr(shd), 1,
                                         ptr(ps), ptr(pq), B, H, W, Cin, Cout, 1, None, None, None, 0, current_stream()))
        outs.append((y, ps, pq))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1]) and torch.equal(outs[0][2], outs[1][2])
    assert rel_err(outs[0][0], ref) < 3e-6
    assert rel_err(outs[0][1].sum(0), ref.sum(0)) < 1e-5 and rel_err(outs[0][2].sum(0), (ref ** 2).sum(0)) < 1e-5

# ---- conv3 of stages 1-3: short K, the A operand resident in registers (csrc/conv1x1_areg.hip) ----
@pytest.mark.parametrize("M,Cin,Cout,bn,pre,relu,in_exp", [
    (4 * 14 * 14, 256, 1024, 128, True, 1, 0),      # stage-3 conv3 at batch 4: 7 workgroups, the last one ragged (784 = 6 x 128 + 16)
    (2 * 28 * 28, 128, 512, 128, True, 1, 0),       # stage 2
    (3136, 64, 256, 128, True, 1, 0),               # stage 1, one image
    (300, 256, 128, 64, True, 1, 0),                # tile width 64, three workgroups, ragged
    (256, 128, 256, 128, False, 0, 0),              # activated input, no fold, no ReLU
    (384, 256, 256, 128, True, 1, 5),               # input scaled by 2^5 on its way into the f16 planes: the same result
    (384, 64, 64, 64, False, 1, -3),
])
def test_conv1x1_areg_has_fp32_accuracy(dev, M, Cin, Cout, bn, pre, relu, in_exp):
    g = torch.Generator().manual_seed(M + Cin + Cout + bn)
    x = torch.randn(M, Cin, generator=g) * torch.exp(torch.randn(M, Cin, generator=g))
    w = torch.randn(Cout, Cin, generator=g) * 0.1 * torch.exp(torch.randn(Cout, Cin, generator=g))
    scale = torch.rand(Cin, generator=g) - 0.3 if pre else None
    shift = torch.randn(Cin, generator=g) if pre else None
    xin = x * scale + shift if pre else x            # the fold is an fp32 fma
    if relu:
        xin = torch.relu(xin)
    ref = xin.double() @ w.double().t()
    L = lib()
    xd, wd = x.to(dev), w.to(dev)
    img = torch.empty(L.capnet_conv1x1_f16x3_weight_words(Cin, Cout), dtype=torch.int32, device=dev)
    check(L.capnet_conv1x1_f16x3_pack(ptr(wd), ptr(img), Cout, Cin, bn, current_stream()))
    tiles = L.capnet_conv1x1_tiles_m(M)
    sd, hd = (scale.to(dev), shift.to(dev)) if pre else (None, None)

    def run(stats, e):
        y = torch.full((M, Cout), float("nan"), device=dev)
        ps = torch.full((tiles, Cout), float("nan"), device=dev) if stats else None
        pq = torch.full((tiles, Cout), float("nan"), device=dev) if stats else None
        check(L.capnet_conv1x1_fwd_areg(ptr(xd), ptr(img), bn, ptr(y), ptr(sd), ptr(hd), relu, ptr(ps), ptr(pq), M, Cin, Cout,
                                        e, current_stream()), "capnet_conv1x1_fwd_areg")
        return y, ps, pq
    y, ps, pq = run(True, in_exp)
    assert rel_err(y, ref) < 3e-6
    assert rel_err(ps.sum(0), ref.sum(0)) < 1e-5 and rel_err(pq.sum(0), (ref ** 2).sum(0)) < 1e-5
    # fp32-grade: rms error against fp64 not above the tiled split-f16 kernel's on the same operands
    rms = lambda t: (((t.double().cpu() - ref) ** 2).mean().sqrt() / (ref ** 2).mean().sqrt()).item()
    y2 = torch.empty((M, Cout), device=dev)
    p1, p2 = torch.empty(tiles, Cout, device=dev), torch.empty(tiles, Cout, device=dev)
    if Cin % 64 == 0 and relu == (1 if pre else relu):
        check(L.capnet_conv1x1_fwd_f16x3(ptr(xd), M * Cin, M * Cin, Cin, ptr(img), bn, ptr(y2), ptr(sd), ptr(hd), relu if pre else 0,
                                         ptr(p1), ptr(p2), 1, 1, M, Cin, Cout, 1, None, None, None, 0, current_stream()))
        if pre or not relu:
            print("rms vs fp64: A in registers %.2e, tiled kernel %.2e" % (rms(y), rms(y2)))
            assert rms(y) < 1.25 * rms(y2) + 2e-8
    y0, _, _ = run(False, in_exp)                     # without statistics: the same output
    assert torch.equal(y0, y)
    if in_exp:
        y1, _, _ = run(False, 0)                      # a power-of-two prescale changes nothing but the f16 pieces' range
        assert rel_err(y, y1) < 3e-7


# ---- 3x3 convolutions through the split-f16 kernel (implicit GEMM over (tap, channel), csrc/conv_f16x3.hip) ----
@pytest.mark.parametrize("bn", [64, 128])
@pytest.mark.parametrize("B,H,W,Cin,Cout,stride,pre,epi", [
    (2, 8, 8, 64, 128, 1, True, 0),         # 128 rows, 18 steps; padding on every side, folded BatchNorm + ReLU
    (3, 7, 7, 128, 128, 1, True, 0),        # odd map, ragged last M tile
    (2, 14, 14, 64, 128, 2, True, 0),       # stride 2 (first block of a stage): 7x7 outputs
    (2, 9, 11, 64, 256, 2, False, 0),       # odd rectangular map, stride 2, activated input (the 0 / 1 factor)
    (4, 14, 14, 256, 256, 1, True, 0),      # stage-3 conv2 at batch 4: 72 steps, several tiles per workgroup at bn = 64
    (2, 7, 7, 64, 128, 1, False, 1),        # folded inference epilogue (+ ReLU), activated input
])
def test_conv3x3_f16x3_matches_fp64(dev, bn, B, H, W, Cin, Cout, stride, pre, epi):
    g = torch.Generator().manual_seed(7 * B + H + Cin + Cout + stride + bn)
    x = torch.randn(B, Cin, H, W, generator=g) * torch.exp(0.5 * torch.randn(B, Cin, H, W, generator=g))
    w = torch.randn(Cout, Cin, 3, 3, generator=g) * 0.05 * torch.exp(0.5 * torch.randn(Cout, Cin, 3, 3, generator=g))
    scale = torch.rand(Cin, generator=g) - 0.3 if pre else None
    shift = torch.randn(Cin, generator=g) if pre else None
    xin = torch.relu(x * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)).double() if pre else x.double()
    ref = torch.nn.functional.conv2d(xin, w.double(), stride=stride, padding=1)
    OH, OW = ref.shape[2], ref.shape[3]
    M = B * OH * OW
    ref = ref.permute(0, 2, 3, 1).reshape(M, Cout)
    xd = x.permute(0, 2, 3, 1).contiguous().to(dev)
    L = lib()
    img = ops.pack_conv_weight_f16x3(w.to(dev), bn)
    y = torch.full((M, Cout), float("nan"), device=dev)
    tiles = L.capnet_conv1x1_tiles_m(M)
    sd, hd = (scale.to(dev), shift.to(dev)) if pre else (None, None)
    if epi:
        sc, sh = torch.rand(Cout, generator=g) + 0.5, torch.randn(Cout, generator=g)
        scd, shd = sc.to(dev), sh.to(dev)
        check(L.capnet_conv2d_fwd_f16x3(ptr(xd), H * W * Cin, W * Cin, Cin, ptr(img), bn, ptr(y), ptr(sd), ptr(hd), int(pre),
                                        None, None, B, H, W, Cin, Cout, 3, stride, 1, ptr(scd), ptr(shd), None, 1,
                                        current_stream()))
        assert rel_err(y, torch.relu(ref * sc.double() + sh.double())) < 3e-6
        return
    psum = torch.full((tiles, Cout), float("nan"), device=dev)
    psq = torch.full((tiles, Cout), float("nan"), device=dev)
    check(L.capnet_conv2d_fwd_f16x3(ptr(xd), H * W * Cin, W * Cin, Cin, ptr(img), bn, ptr(y), ptr(sd), ptr(hd), int(pre),
                                    ptr(psum), ptr(psq), B, H, W, Cin, Cout, 3, stride, 1, None, None, None, 0,
                                    current_stream()))
    assert rel_err(y, ref) < 3e-6
    rms = (((y.double().cpu() - ref) ** 2).mean().sqrt() / (ref ** 2).mean().sqrt()).item()
    print("3x3 split-f16 rms vs fp64: %.2e" % rms)
    assert rms < 6e-7
    assert rel_err(psum.sum(0), ref.sum(0)) < 1e-5
    assert rel_err(psq.sum(0), (ref ** 2).sum(0)) < 1e-5


@pytest.mark.parametrize("B,H,W", [(2, 224, 224), (3, 64, 96), (1, 40, 264), (2, 32, 520), (5, 7, 8)])
def test_conv_stem_f16x3_matches_fp64(dev, B, H, W):
    """The 7x7 / 2 stem on the split-f16 kernel (NCHW image in, NHWC out): fp32-grade against fp64, odd and
    rectangular maps, output rows of one, two and three 128-wide segments, a strided (non-contiguous) batch."""
    g = torch.Generator().manual_seed(11 * B + H + W)
    big = torch.randn(B, 2, 3, H, W, generator=g) * torch.exp(0.5 * torch.randn(B, 2, 3, H, W, generator=g))
    x = big[:, 1]                                   # batch stride 2 * 3 * H * W
    w = torch.randn(64, 3, 7, 7, generator=g) * 0.05 * torch.exp(0.5 * torch.randn(64, 3, 7, 7, generator=g))
    ref = torch.nn.functional.conv2d(x.double(), w.double(), stride=2, padding=3)
    OH, OW = ref.shape[2], ref.shape[3]
    ref = ref.permute(0, 2, 3, 1).reshape(B * OH * OW, 64)
    L = lib()
    bigd = big.to(dev)
    xd = bigd[:, 1]
    img = ops.pack_conv_weight_stem_f16x3(w.to(dev))
    y = torch.full((B * OH * OW, 64), float("nan"), device=dev)
    rows = L.capnet_conv_stem_f16x3_part_rows(B, H, W)
    psum = torch.full((rows, 64), float("nan"), device=dev)
    psq = torch.full((rows, 64), float("nan"), device=dev)
    check(L.capnet_conv_stem_fwd_f16x3(ptr(xd), xd.stride(0), xd.stride(1), xd.stride(2), ptr(img), ptr(y), ptr(psum),
                                       ptr(psq), B, H, W, current_stream()))
    assert rel_err(y, ref) < 3e-6
    rms = (((y.double().cpu() - ref) ** 2).mean().sqrt() / (ref ** 2).mean().sqrt()).item()
    print("stem split-f16 rms vs fp64: %.2e" % rms)
    assert rms < 6e-7
    assert rel_err(psum.sum(0), ref.sum(0)) < 1e-5
    assert rel_err(psq.sum(0), (ref ** 2).sum(0)) < 1e-5
    # no statistics asked for: same output
    y2 = torch.full_like(y, float("nan"))
    check(L.capnet_conv_stem_fwd_f16x3(ptr(xd), xd.stride(0), xd.stride(1), xd.stride(2), ptr(img), ptr(y2), None,
                                       None, B, H, W, current_stream()))
    assert torch.equal(y, y2)


@pytest.mark.parametrize("shared", [0, 1])
@pytest.mark.parametrize("B,H,W,Cin,Cout,bn,pre", [
    (2, 14, 14, 256, 256, 128, True), (3, 7, 7, 512, 512, 128, True), (1, 56, 56, 64, 64, 64, True),
    (2, 28, 28, 128, 128, 128, False), (2, 9, 11, 64, 128, 64, True), (5, 3, 3, 32, 64, 64, False),
    (2, 14, 14, 256, 256, 256, True), (3, 7, 7, 512, 512, 256, True), (2, 9, 11, 64, 256, 256, False)])   # the 128 x 256 tile
def test_conv3x3_patch_matches_fp64(dev, B, H, W, Cin, Cout, bn, pre, shared):
    """Stride-1 3x3 convolution with the tile's input patch resident in LDS: fp32-grade against fp64 on the trunk's
    four map sizes, odd maps, tiles that span several images and ragged last tiles; same weight image as the
    implicit-GEMM kernel, whose output it must reproduce to rounding. shared = 0: wave pairs split K (these launches
    have at most one tile per CU); 1: the <= 128-VGPR arrangement used beside other kernels."""
    g = torch.Generator().manual_seed(13 * B + H + W + Cin + Cout)
    x = torch.randn(B, Cin, H, W, generator=g) * torch.exp(0.5 * torch.randn(B, Cin, H, W, generator=g))
    w = torch.randn(Cout, Cin, 3, 3, generator=g) * 0.05 * torch.exp(0.5 * torch.randn(Cout, Cin, 3, 3, generator=g))
    scale = torch.rand(Cin, generator=g) - 0.3 if pre else None
    shift = torch.randn(Cin, generator=g) if pre else None
    xin = torch.relu(x * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)).double() if pre else x.double()
    ref = torch.nn.functional.conv2d(xin, w.double(), stride=1, padding=1)
    M = B * H * W
    ref = ref.permute(0, 2, 3, 1).reshape(M, Cout)
    xd = x.permute(0, 2, 3, 1).contiguous().to(dev)
    L = lib()
    img = ops.pack_conv_weight_f16x3(w.to(dev), bn)
    y = torch.full((M, Cout), float("nan"), device=dev)
    tiles = L.capnet_conv1x1_tiles_m(M)
    psum = torch.full((tiles, Cout), float("nan"), device=dev)
    psq = torch.full((tiles, Cout), float("nan"), device=dev)
    sd, hd = (scale.to(dev), shift.to(dev)) if pre else (None, None)
    check(L.capnet_conv3x3_fwd_patch(ptr(xd), ptr(img), bn, ptr(y), ptr(sd), ptr(hd), int(pre), ptr(psum), ptr(psq),
                                     B, H, W, Cin, Cout, shared, current_stream()))
    assert rel_err(y, ref) < 3e-6
    rms = (((y.double().cpu() - ref) ** 2).mean().sqrt() / (ref ** 2).mean().sqrt()).item()
    print("3x3 patch kernel rms vs fp64: %.2e" % rms)
    assert rms < 8e-7          # K = 9 * 512 = 4 608 fp32 accumulations at the deep end
    assert rel_err(psum.sum(0), ref.sum(0)) < 1e-5
    assert rel_err(psq.sum(0), (ref ** 2).sum(0)) < 1e-5
    if Cin % 64 == 0 and bn <= 128:               # (the 256-wide image is the patch / tail kernels' own)
        y2 = torch.full_like(y, float("nan"))
        check(L.capnet_conv2d_fwd_f16x3(ptr(xd), H * W * Cin, W * Cin, Cin, ptr(img), bn, ptr(y2), ptr(sd), ptr(hd), int(pre),
                                        None, None, B, H, W, Cin, Cout, 3, 1, 1, None, None, None, 0, current_stream()))
        assert rel_err(y, y2) < 4e-6          # another order of the same fp32 accumulation


@pytest.mark.parametrize("M,Cin,Cout,bn,ds", [(12544, 1024, 256, 128, False), (300, 256, 64, 64, True),
                                              (3136, 2048, 512, 128, False), (1000, 64, 128, 64, True),
                                              (12544, 1024, 256, 256, False),     # stage 3 on the 128 x 256 tile (64 x 64 wave tiles)
                                              (3136, 2048, 512, 256, True), (300, 512, 256, 256, True)])
def test_conv1x1_tail_fusion_matches_its_two_kernels(dev, M, Cin, Cout, bn, ds):
    """A block's tail + the next block's conv1 in one launch: the tail it writes is bit-for-bit bn_add_relu's, the
    convolution is fp32-grade against fp64 (ragged tiles, 2 to 64 k-steps per tile, with and without a BatchNorm on
    the identity branch), and equals the split-f16 1x1 kernel run on that tail to rounding."""
    g = torch.Generator().manual_seed(M + Cin + Cout)
    y3 = torch.randn(M, Cin, generator=g) * torch.exp(0.5 * torch.randn(M, Cin, generator=g))
    res = torch.randn(M, Cin, generator=g)
    s1, t1 = torch.rand(Cin, generator=g) + 0.5, torch.randn(Cin, generator=g)
    s2, t2 = (torch.rand(Cin, generator=g) + 0.5, torch.randn(Cin, generator=g)) if ds else (None, None)
    w = torch.randn(Cout, Cin, 1, 1, generator=g) * 0.05
    L = lib()
    d = lambda t: None if t is None else t.to(dev)
    y3d, resd, s1d, t1d, s2d, t2d = d(y3), d(res), d(s1), d(t1), d(s2), d(t2)
    want_tail = torch.empty(M, Cin, device=dev)
    check(L.capnet_bn_add_relu(ptr(y3d), ptr(s1d), ptr(t1d), ptr(resd), ptr(s2d), ptr(t2d), ptr(want_tail), M, Cin,
                               current_stream()))
    img = ops.pack_conv_weight_f16x3(w.to(dev), bn)
    tiles = L.capnet_conv1x1_tiles_m(M)
    tail = torch.full((M, Cin), float("nan"), device=dev)
    y = torch.full((M, Cout), float("nan"), device=dev)
    psum = torch.full((tiles, Cout), float("nan"), device=dev)
    psq = torch.full((tiles, Cout), float("nan"), device=dev)
    check(L.capnet_conv1x1_fwd_tail(ptr(y3d), ptr(s1d), ptr(t1d), ptr(resd), ptr(s2d), ptr(t2d), ptr(tail), ptr(img), bn,
                                    ptr(y), ptr(psum), ptr(psq), M, Cin, Cout, current_stream()))
    assert torch.equal(tail, want_tail)
    ref = want_tail.double().cpu() @ w.reshape(Cout, Cin).double().t()
    assert rel_err(y, ref) < 3e-6
    assert rel_err(psum.sum(0), ref.sum(0)) < 1e-5
    assert rel_err(psq.sum(0), (ref ** 2).sum(0)) < 1e-5
    y2 = torch.full_like(y, float("nan"))
    bn2 = min(bn, 128)                              # (the 256-wide image is the tail kernel's own)
    img2 = img if bn2 == bn else ops.pack_conv_weight_f16x3(w.to(dev), bn2)
    check(L.capnet_conv2d_fwd_f16x3(ptr(want_tail), Cin, Cin, Cin, ptr(img2), bn2, ptr(y2), None, None, 0, None, None,
                                    1, M, 1, Cin, Cout, 1, 1, 0, None, None, None, 0, current_stream()))
    assert rel_err(y, y2) < 4e-6


def test_conv3x3_patch_random_shape_sweep(dev):
    """24 seeded random shapes (maps 3..56 wide, 1..4 images, channel counts off the trunk's grid, both wave arrangements)
    against fp64: the patch kernel's masks, image boundaries inside a tile and ragged tiles."""
    rng = random.Random(20261004)
    L = lib()
    worst = 0.0
    for case in range(24):
        B, H, W = rng.randint(1, 4), rng.randint(3, 56), rng.randint(3, 56)
        Cin, Cout = rng.choice([32, 64, 96, 128]), rng.choice([64, 128, 192, 256])
        pre, shared = rng.random() < 0.7, rng.randint(0, 1)
        bn = 128 if Cout % 128 == 0 else 64
        g = torch.Generator().manual_seed(1000 + case)
        x = torch.randn(B, Cin, H, W, generator=g)
        w = torch.randn(Cout, Cin, 3, 3, generator=g) * 0.05
        scale = torch.rand(Cin, generator=g) - 0.3 if pre else None
        shift = torch.randn(Cin, generator=g) if pre else None
        xin = torch.relu(x * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)).double() if pre else x.double()
        M = B * H * W
        ref = torch.nn.functional.conv2d(xin, w.double(), padding=1).permute(0, 2, 3, 1).reshape(M, Cout)
        xd = x.permute(0, 2, 3, 1).contiguous().to(dev)
        img = ops.pack_conv_weight_f16x3(w.to(dev), bn)
        y = torch.full((M, Cout), float("nan"), device=dev)
        tiles = L.capnet_conv1x1_tiles_m(M)
        psum = torch.full((tiles, Cout), float("nan"), device=dev)
        psq = torch.full((tiles, Cout), float("nan"), device=dev)
        sd, hd = (scale.to(dev), shift.to(dev)) if pre else (None, None)
        check(L.capnet_conv3x3_fwd_patch(ptr(xd), ptr(img), bn, ptr(y), ptr(sd), ptr(hd), int(pre), ptr(psum), ptr(psq),
                                         B, H, W, Cin, Cout, shared, current_stream()))
        e = rel_err(y, ref)
        worst = max(worst, e)
        assert e < 3e-6, (case, B, H, W, Cin, Cout, pre, shared, e)
        assert rel_err(psum.sum(0), ref.sum(0)) < 1e-5, (case, "sum")
        assert rel_err(psq.sum(0), (ref ** 2).sum(0)) < 1e-5, (case, "sq")
    print("patch kernel sweep: worst max-norm error %.2e" % worst)


def test_conv1x1_tail_random_shape_sweep(dev):
    """20 seeded random shapes of the fused tail + conv1 (1..5000 rows, 1 to 16 k-steps per tile, with and without a
    BatchNorm on the identity): the tail bit-for-bit bn_add_relu's, the product against fp64."""
    rng = random.Random(4102026)
    L = lib()
    for case in range(20):
        M = rng.choice([1, 7, 127, 128, 129, 300, 1000, 2049, 5000])
        Cin, Cout = rng.choice([32, 64, 128, 256, 512]), rng.choice([64, 128, 192, 256])
        ds = rng.random() < 0.4
        bn = 128 if Cout % 128 == 0 else 64
        g = torch.Generator().manual_seed(2000 + case)
        y3, res = torch.randn(M, Cin, generator=g), torch.randn(M, Cin, generator=g)
        s1, t1 = torch.rand(Cin, generator=g) + 0.5, torch.randn(Cin, generator=g)
        s2, t2 = (torch.rand(Cin, generator=g) + 0.5, torch.randn(Cin, generator=g)) if ds else (None, None)
        w = torch.randn(Cout, Cin, 1, 1, generator=g) * 0.05
        d = lambda t: None if t is None else t.to(dev)
        y3d, resd, s1d, t1d, s2d, t2d = d(y3), d(res), d(s1), d(t1), d(s2), d(t2)
        want = torch.empty(M, Cin, device=dev)
        check(L.capnet_bn_add_relu(ptr(y3d), ptr(s1d), ptr(t1d), ptr(resd), ptr(s2d), ptr(t2d), ptr(want), M, Cin,
                                   current_stream()))
        img = ops.pack_conv_weight_f16x3(w.to(dev), bn)
        tiles = L.capnet_conv1x1_tiles_m(M)
        tail = torch.full((M, Cin), float("nan"), device=dev)
        y = torch.full((M, Cout), float("nan"), device=dev)
        psum = torch.full((tiles, Cout), float("nan"), device=dev)
        psq = torch.full((tiles, Cout), float("nan"), device=dev)
        check(L.capnet_conv1x1_fwd_tail(ptr(y3d), ptr(s1d), ptr(t1d), ptr(resd), ptr(s2d), ptr(t2d), ptr(tail), ptr(img),
                                        bn, ptr(y), ptr(psum), ptr(psq), M, Cin, Cout, current_stream()))
        assert torch.equal(tail, want), (case, M, Cin, Cout, ds)
        ref = want.double().cpu() @ w.reshape(Cout, Cin).double().t()
        assert rel_err(y, ref) < 3e-6, (case, M, Cin, Cout, ds)
        assert rel_err(psum.sum(0), ref.sum(0)) < 1e-5, (case, "sum")
        assert rel_err(psq.sum(0), (ref ** 2).sum(0)) < 1e-5, (case, "sq")
