"""Per-kernel parity tests: every libmgld_hip entry point (called through the C ABI) vs a plain fp32 torch-CPU
statement of the same op.  Tolerances are relative-L2; fp16-operand kernels are compared against the fp32 result of
the SAME fp16-rounded operands, so the tolerance only has to cover fp32-accumulation order + the fp16 output rounding.
"""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DEV = "cuda"


def rel_l2(a, b):
    a = a.double().flatten()
    b = b.double().flatten()
    return float((a - b).norm() / (b.norm() + 1e-30))


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def h16(t):
    return t.half()


# ------------------------------------------------------------------------------------------------------------------
# igemm
# ------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (256, 320, 320), (100, 72, 40), (4096, 1280, 320), (512, 1280, 1280),
                                   (77, 640, 1024), (33, 4, 320), (300, 32, 544), (64, 64, 8), (8192, 640, 2560),
                                   (16384, 320, 640), (16384, 1280, 320)])   # the last: >= 384 tiles -> 128x128, 8 waves
def test_igemm_linear(hip, M, N, K):
    a, w, b = h16(rnd(M, K, seed=1)), h16(rnd(N, K, seed=2, scale=K ** -0.5)), rnd(N, seed=3)
    ref = a.float() @ w.float().t() + b
    out = torch.empty(M, N, dtype=torch.half, device=DEV)
    hip.igemm(a.to(DEV), w.to(DEV), out, bias=b.to(DEV))
    torch.cuda.synchronize()
    assert rel_l2(out.cpu().float(), ref) < 1e-3


@pytest.mark.parametrize("depth", [2, 3, 4, 10])
@pytest.mark.parametrize("M,N,K,act", [(32768, 320, 320, 0), (8192, 640, 640, 0), (2048, 1280, 1280, 3), (4096, 512, 128, 0),
                                       (300, 200, 64, 0), (8192, 640, 320, 4), (1000, 2560, 704, 4)])
def test_igemm_linear_ring_depth(hip, depth, M, N, K, act):
    """LINEAR fast path with a 2 / 3 / 4-deep LDS ring (tune = depth - 1): counted vmcnt keeps depth - 2 stages in flight
    across the stage barrier; K shorter than the ring, ragged M / N and the GEGLU epilogue included.  depth 10 (tune 9) is the
    register-staged variant (global_load -> VGPR -> ds_write, two LDS buffers).  All must give the same bits as depth 2 (same
    products, same summation order)."""
    from mgld_vsr_amd.engine import pack_geglu
    a, w, b = h16(rnd(M, K, seed=31)), h16(rnd(N, K, seed=32, scale=K ** -0.5)), rnd(N, seed=33)
    pre = a.float() @ w.float().t() + b
    if act == hip.ACT_GEGLU:
        ref = pre[:, :N // 2] * F.gelu(pre[:, N // 2:])
        w, b = pack_geglu(w, b)
    else:
        ref = F.silu(pre) if act == hip.ACT_SILU else pre
    n_out = N // 2 if act == hip.ACT_GEGLU else N
    out = torch.full((M, n_out), float("nan"), dtype=torch.half, device=DEV)
    base = torch.empty_like(out)
    hip.igemm(a.to(DEV), w.to(DEV), base, bias=b.to(DEV), act=act, tune=1)
    hip.igemm(a.to(DEV), w.to(DEV), out, bias=b.to(DEV), act=act, tune=depth - 1)
    torch.cuda.synchronize()
    assert rel_l2(out.cpu().float(), ref) < 1e-3
    assert torch.equal(out, base)


@pytest.mark.parametrize("M,K,act,resid", [(32768, 320, 0, True), (16384, 1280, 3, False), (1000, 640, 0, True), (128, 960, 0, False)])
def test_igemm_linear_full_n_tile(hip, M, K, act, resid):
    """128 x 320 tiles (eight waves of 32 x 160, the whole N = 320 in one tile: tune = 10) against the 128 x 64 tiles (tune = 11):
    same products per output element and the same k order -> identical bits; bias, SiLU, residual and ragged M included"""
    N = 320
    a, w, b = h16(rnd(M, K, seed=51)), h16(rnd(N, K, seed=52, scale=K ** -0.5)), rnd(N, seed=53)
    r = h16(rnd(M, N, seed=54)) if resid else None
    pre = a.float() @ w.float().t() + b
    ref = (F.silu(pre) if act == hip.ACT_SILU else pre) + (r.float() if resid else 0)
    outs = []
    for tune in (10, 11):
        out = torch.full((M, N), float("nan"), dtype=torch.half, device=DEV)
        hip.igemm(a.to(DEV), w.to(DEV), out, bias=b.to(DEV), act=act, resid=None if r is None else r.to(DEV), tune=tune)
        outs.append(out)
    torch.cuda.synchronize()
    assert rel_l2(outs[0].cpu().float(), ref) < 1e-3
    assert torch.equal(outs[0], outs[1])


@pytest.mark.parametrize("M,N,K", [(512, 1280, 11520), (2048, 640, 5760), (300, 200, 2048), (64, 1280, 23040),
                                   (2048, 1280, 5120), (1000, 384, 4096)])
def test_igemm_splitk(hip, M, N, K):
    """few output tiles + deep K -> split along K into fp32 slabs + reduce/epilogue kernel"""
    ws = hip._test_ws   # session-lifetime scratch registered by the `hip` fixture
    hip.set_workspace(ws)
    a, w, b = h16(rnd(M, K, seed=1)), h16(rnd(N, K, seed=2, scale=K ** -0.5)), rnd(N, seed=3)
    r = h16(rnd(M, N, seed=4))
    ref = 0.5 * F.silu(a.float() @ w.float().t() + b) + 2.0 * r.float()
    out = torch.empty(M, N, dtype=torch.half, device=DEV)
    ad, wd = a.to(DEV), w.to(DEV)
    hip.igemm(ad, wd, out, bias=b.to(DEV), resid=r.to(DEV), act=hip.ACT_SILU, alpha=0.5, beta=2.0)
    p = hip.MgldIGemm()
    p.M, p.N, p.K, p.batch, p.act = M, N, K, 1, hip.ACT_SILU
    assert hip.igemm_config(p) >= 2000000, "expected the split-K path"
    assert rel_l2(out.cpu().float(), ref) < 1e-3
    hip.set_workspace(None)
    out2 = torch.empty(M, N, dtype=torch.half, device=DEV)
    hip.igemm(ad, wd, out2, bias=b.to(DEV), resid=r.to(DEV), act=hip.ACT_SILU, alpha=0.5, beta=2.0)
    assert hip.igemm_config(p) < 1000000
    assert rel_l2(out2.cpu().float(), ref) < 1e-3
    hip.set_workspace(ws)


def test_igemm_splitk_repeatable(hip):
    """the K slices are summed in a fixed order: back-to-back launches of the same problem (no host sync in between)
    give bit-identical results"""
    M, N, K = 2048, 1280, 5120
    a, w = h16(rnd(M, K, seed=5)).to(DEV), h16(rnd(N, K, seed=6, scale=K ** -0.5)).to(DEV)
    p = hip.MgldIGemm()
    p.M, p.N, p.K, p.batch = M, N, K, 1
    cfg = hip.igemm_config(p)
    assert cfg >= 2000000, cfg               # split along K
    outs = [torch.empty(M, N, dtype=torch.float32, device=DEV) for _ in range(12)]
    for o in outs:
        hip.igemm(a, w, o)
    torch.cuda.synchronize()
    ref = a.float() @ w.float().t()
    assert rel_l2(outs[0], ref) < 1e-3
    for o in outs[1:]:
        assert torch.equal(o, outs[0])


def test_igemm_transpose_detect(hip):
    # A = I-like with asymmetric W catches row/col swaps of the MFMA output mapping
    M = N = K = 64
    a = torch.eye(M, K).half()
    w = (torch.arange(N * K).reshape(N, K).float() / (N * K)).half()
    out = torch.empty(M, N, dtype=torch.float32, device=DEV)
    hip.igemm(a.to(DEV), w.to(DEV), out)
    assert torch.allclose(out.cpu(), w.float().t(), atol=1e-6)


def test_igemm_strided_epilogues(hip):
    # lda/ldc/ldr strides (concat-by-slices), residual with alpha/beta, SiLU, fp32 out, bias_m
    M, N, K = 200, 96, 128
    abig = h16(rnd(M, K + 64, seed=4)).to(DEV)
    a = abig[:, 32:32 + K]
    w = h16(rnd(N, K, seed=5, scale=K ** -0.5)).to(DEV)
    rbig = h16(rnd(M, N + 16, seed=6)).to(DEV)
    r = rbig[:, 8:8 + N]
    bm = rnd(M, seed=7).to(DEV)
    obig = torch.zeros(M, N + 40, dtype=torch.half, device=DEV)
    out = obig[:, 16:16 + N]
    hip.igemm(a, w, out, bias_m=bm, resid=r, act=hip.ACT_SILU, alpha=0.3, beta=0.7)
    ref = 0.3 * F.silu(a.cpu().float() @ w.cpu().float().t() + bm.cpu()[:, None]) + 0.7 * r.cpu().float()
    assert rel_l2(out.cpu().float(), ref) < 1e-3
    assert float(obig[:, :16].abs().max()) == 0 and float(obig[:, 16 + N:].abs().max()) == 0
    o32 = torch.empty(M, N, dtype=torch.float32, device=DEV)
    hip.igemm(a, w, o32, act=hip.ACT_RELU)
    assert rel_l2(o32.cpu(), F.relu(a.cpu().float() @ w.cpu().float().t())) < 1e-5


def test_igemm_epilogues_large_tiles(hip):
    """the fused epilogue (per-row bias, residual with alpha/beta, SiLU, per-frame rowvec) on the 128x128 tile configuration
    (>= 384 tiles), which the small cases above never select"""
    M, N, K, rpf = 32768, 256, 128, 4096
    a = h16(rnd(M, K, seed=40)).to(DEV)
    w = h16(rnd(N, K, seed=41, scale=K ** -0.5)).to(DEV)
    r = h16(rnd(M, N, seed=42)).to(DEV)
    b, rv = rnd(N, seed=43).to(DEV), rnd(M // rpf, N, seed=44).to(DEV)
    out = torch.empty(M, N, dtype=torch.half, device=DEV)
    hip.igemm(a, w, out, bias=b, rowvec=rv, rows_per_frame=rpf, resid=r, act=hip.ACT_SILU, alpha=0.6, beta=1.4, tune=12)
    p = hip.MgldIGemm()
    p.M, p.N, p.K, p.batch, p.tune = M, N, K, 1, 12       # (tune 12: the 128-class kernels; the planner gives this shape to ppgemm)
    assert hip.igemm_config(p) == 128128
    ref = 0.6 * F.silu(a.cpu().float() @ w.cpu().float().t() + b.cpu() + rv.cpu().repeat_interleave(rpf, 0)) + 1.4 * r.cpu().float()
    assert rel_l2(out.cpu().float(), ref) < 1e-3


def test_igemm_batched_nt(hip):
    # V^T projection: per-frame out[C, tokens] = Wv[C, Cin] @ x_f[tokens, Cin]^T  (A shared, W strided)
    Fr, tokens, Cin, Cc = 3, 200, 64, 128
    x = h16(rnd(Fr * tokens, Cin, seed=8)).to(DEV)
    wv = h16(rnd(Cc, Cin, seed=9, scale=Cin ** -0.5)).to(DEV)
    tp = 208
    out = torch.zeros(Fr * Cc, tp, dtype=torch.half, device=DEV)
    hip.igemm(wv, x, out, M=Cc, N=tokens, K=Cin, batch=Fr, strideA=0, strideW=tokens * Cin, strideC=Cc * tp)
    ref = torch.einsum("ck,ftk->fct", wv.cpu().float(), x.cpu().float().reshape(Fr, tokens, Cin))
    assert rel_l2(out.cpu().float().reshape(Fr, Cc, tp)[:, :, :tokens], ref) < 1e-3


@pytest.mark.parametrize("M,dim,inner", [(300, 64, 256), (4096, 128, 512), (32768, 320, 1280)])   # 64x128 and 128x128 tile paths
def test_igemm_geglu(hip, M, dim, inner):
    a = h16(rnd(M, dim, seed=10)).to(DEV)
    w = h16(rnd(2 * inner, dim, seed=11, scale=dim ** -0.5))
    b = rnd(2 * inner, seed=12)
    from mgld_vsr_amd.engine import pack_geglu
    wp, bp = pack_geglu(w, b)
    out = torch.empty(M, inner, dtype=torch.half, device=DEV)
    hip.igemm(a, wp.to(DEV), out, bias=bp.to(DEV), act=hip.ACT_GEGLU)
    y = a.cpu().float() @ w.float().t() + b
    ref = y[:, :inner] * F.gelu(y[:, inner:])
    assert rel_l2(out.cpu().float(), ref) < 1e-3


# ping-pong LINEAR kernels (csrc/ppgemm.hip): tune = 21 + configuration id forces a tile configuration, tune = 12 keeps the 128-class kernel
PP_TILES = {0: (256, 256), 1: (256, 160), 2: (128, 160), 3: (128, 256), 4: (256, 128), 5: (128, 128), 6: (256, 320)}


@pytest.mark.parametrize("cfg", sorted(PP_TILES))
@pytest.mark.parametrize("mt,nt,K,epi", [(1, 1, 64, 0), (2, 3, 320, 1), (8, 2, 192, 2), (3, 1, 1280, 3), (16, 8, 128, 1)])
def test_igemm_pingpong_linear(hip, cfg, mt, nt, K, epi):
    """every tile configuration of the ping-pong LINEAR kernel: one stage (K = 64), K shorter / longer than the ring, all three tile orders
    (8 | tiles_m, 8 | tiles_n, neither); epilogues: bias, bias + per-frame row vector + SiLU, residual with alpha / beta into strided
    buffers, nothing at all.  Reference: fp32 products of the same fp16 operands."""
    bm, bn = PP_TILES[cfg]
    M, N = mt * bm, nt * bn
    a, w = h16(rnd(M, K, seed=51)).to(DEV), h16(rnd(N, K, seed=52, scale=K ** -0.5)).to(DEV)
    b = rnd(N, seed=53).to(DEV)
    pre = a.cpu().float() @ w.cpu().float().t()
    p = hip.MgldIGemm()
    p.M, p.N, p.K, p.batch, p.tune, p.ldc = M, N, K, 1, 21 + cfg, N
    assert hip.igemm_config(p) == 500000 + cfg
    if epi == 0:
        out = torch.empty(M, N, dtype=torch.half, device=DEV)
        hip.igemm(a, w, out, bias=b, tune=21 + cfg)
        ref = pre + b.cpu()
    elif epi == 1:
        rpf = M // 4 if M % 4 == 0 else M
        rv = rnd(M // rpf, N, seed=54).to(DEV)
        out = torch.empty(M, N, dtype=torch.half, device=DEV)
        hip.igemm(a, w, out, bias=b, rowvec=rv, rows_per_frame=rpf, act=hip.ACT_SILU, tune=21 + cfg)
        ref = F.silu(pre + b.cpu() + rv.cpu().repeat_interleave(rpf, 0))
    elif epi == 2:
        rbig = h16(rnd(M, N + 16, seed=55)).to(DEV)
        r = rbig[:, 8:8 + N]
        obig = torch.zeros(M, N + 40, dtype=torch.half, device=DEV)
        out = obig[:, 16:16 + N]
        hip.igemm(a, w, out, bias=b, resid=r, alpha=0.6, beta=1.4, tune=21 + cfg)
        ref = 0.6 * (pre + b.cpu()) + 1.4 * r.cpu().float()
        assert float(obig[:, :16].abs().max()) == 0 and float(obig[:, 16 + N:].abs().max()) == 0
    else:
        out = torch.empty(M, N, dtype=torch.half, device=DEV)
        hip.igemm(a, w, out, tune=21 + cfg)
        ref = pre
    torch.cuda.synchronize()
    assert rel_l2(out.cpu().float(), ref) < 1e-3
    # same products, fp32 accumulation in another order: the 128-class kernel agrees to the fp16 output rounding
    if epi == 3:
        old = torch.empty(M, N, dtype=torch.half, device=DEV)
        hip.igemm(a, w, old, tune=12)
        assert rel_l2(out.cpu().float(), old.cpu().float()) < 5e-4


@pytest.mark.parametrize("cfg", [0, 3, 4, 5])
@pytest.mark.parametrize("mt,nt,dim", [(1, 1, 64), (4, 5, 320), (8, 8, 640)])
def test_igemm_pingpong_geglu(hip, cfg, mt, nt, dim):
    bm, bn = PP_TILES[cfg]
    M, inner = mt * bm, nt * bn // 2
    a = h16(rnd(M, dim, seed=56)).to(DEV)
    w = h16(rnd(2 * inner, dim, seed=57, scale=dim ** -0.5))
    b = rnd(2 * inner, seed=58)
    from mgld_vsr_amd.engine import pack_geglu
    wp, bp = pack_geglu(w, b)
    out = torch.empty(M, inner, dtype=torch.half, device=DEV)
    hip.igemm(a, wp.to(DEV), out, bias=bp.to(DEV), act=hip.ACT_GEGLU, tune=21 + cfg)
    y = a.cpu().float() @ w.float().t() + b
    ref = y[:, :inner] * F.gelu(y[:, inner:])
    assert rel_l2(out.cpu().float(), ref) < 1e-3


def test_igemm_pingpong_race_screen(hip):
    """the counted waits / barrier placement of the ring: many launches of a long-K problem on every configuration must give the same
    bits every time (a stage read before its DMA landed shows up as launch-to-launch differences)"""
    K = 2560
    for cfg, (bm, bn) in PP_TILES.items():
        M, N = 16 * bm, 2 * bn
        a, w = h16(rnd(M, K, seed=59)).to(DEV), h16(rnd(N, K, seed=60, scale=K ** -0.5)).to(DEV)
        outs = [torch.empty(M, N, dtype=torch.half, device=DEV) for _ in range(6)]
        for o in outs:
            hip.igemm(a, w, o, tune=21 + cfg)
        torch.cuda.synchronize()
        for o in outs[1:]:
            assert torch.equal(o, outs[0]), f"configuration {cfg}: launches differ"
        ref = a.cpu().float() @ w.cpu().float().t()
        assert rel_l2(outs[0].cpu().float(), ref) < 1e-3


def _conv_ref(x_nchw, w, b, stride, pad):
    return F.conv2d(F.pad(x_nchw, pad), w, b, stride=stride)


def _to_tok(x_nchw):
    n, c, h, w = x_nchw.shape
    return x_nchw.permute(0, 2, 3, 1).reshape(n * h * w, c).contiguous()


def _from_tok(t, n, h, w):
    return t.reshape(n, h, w, -1).permute(0, 3, 1, 2)


@pytest.mark.parametrize("n,cin,cout,h,w,stride,pads,up2", [
    (2, 32, 64, 16, 16, 1, (1, 1, 1, 1), 0),
    (1, 8, 320, 12, 20, 1, (1, 1, 1, 1), 0),
    (2, 64, 64, 16, 16, 2, (1, 1, 1, 1), 0),     # UNet Downsample: stride 2, pad 1
    (2, 64, 32, 16, 16, 2, (0, 1, 0, 1), 0),     # VAE Downsample: pad (0,1,0,1), stride 2
    (2, 64, 48, 8, 8, 1, (1, 1, 1, 1), 1),       # nearest-2x upsample folded into the gather
    (1, 544, 32, 10, 10, 1, (1, 1, 1, 1), 0),    # RDB growth conv
    (3, 320, 4, 8, 8, 1, (1, 1, 1, 1), 0),       # out conv
    (3, 128, 200, 12, 12, 1, (1, 1, 1, 1), 0),   # (tap, Cin) K order, FAST path, K = 1152
    (2, 192, 96, 16, 16, 2, (0, 1, 0, 1), 0),    # stride 2 + asymmetric pad, FAST path
])
def test_igemm_conv3x3(hip, n, cin, cout, h, w, stride, pads, up2):
    x = h16(rnd(n, cin, h, w, seed=20))
    wt = h16(rnd(cout, cin, 3, 3, seed=21, scale=(9 * cin) ** -0.5))
    b = rnd(cout, seed=22)
    xin = F.interpolate(x.float(), scale_factor=2, mode="nearest") if up2 else x.float()
    ref = _conv_ref(xin, wt.float(), b, stride, pads)
    ho, wo = ref.shape[2], ref.shape[3]
    wk = wt.permute(0, 2, 3, 1).reshape(cout, 9 * cin).contiguous().to(DEV)
    out = torch.empty(n * ho * wo, cout, dtype=torch.half, device=DEV)
    hip.igemm(_to_tok(x).to(DEV), wk, out, mode=hip.MODE_CONV3X3, bias=b.to(DEV),
              conv=(cin, h, w, ho, wo, stride, pads[2], pads[0], up2))
    assert rel_l2(_from_tok(out.cpu().float(), n, ho, wo), ref) < 1e-3


@pytest.mark.parametrize("n,cin,cout,h,w,stride", [(2, 128, 96, 16, 16, 1), (1, 320, 320, 24, 16, 1), (2, 64, 64, 16, 16, 2),
                                                   (8, 1280, 256, 8, 8, 1),
                                                   (4, 320, 640, 64, 64, 1)])   # 640 tiles: the 128x128 DMA-path config
def test_igemm_conv3x3_tap_inner(hip, n, cin, cout, h, w, stride):
    """(64-channel block, tap, channel) K order — the layout the engine packs whenever Cin % 64 == 0"""
    from mgld_vsr_amd.engine import pack_conv3x3
    x = h16(rnd(n, cin, h, w, seed=60))
    wt = h16(rnd(cout, cin, 3, 3, seed=61, scale=(9 * cin) ** -0.5))
    b = rnd(cout, seed=62)
    ref = F.conv2d(x.float(), wt.float(), b, stride=stride, padding=1)
    ho, wo = ref.shape[2], ref.shape[3]
    wk = pack_conv3x3(wt, tap_inner=True).to(DEV)
    out = torch.empty(n * ho * wo, cout, dtype=torch.half, device=DEV)
    hip.igemm(_to_tok(x).to(DEV), wk, out, mode=hip.MODE_CONV3X3, bias=b.to(DEV), conv=(cin, h, w, ho, wo, stride, 1, 1, 0),
              tap_inner=1)
    assert rel_l2(_from_tok(out.cpu().float(), n, ho, wo), ref) < 1e-3


@pytest.mark.parametrize("n,cin,cout,h,w,ti,epi", [
    (2, 128, 96, 16, 16, False, 0),      # 128 weight rows per block, (tap, Cin) K order, N not a multiple of the tile
    (4, 320, 640, 64, 64, True, 0),      # W = 64: 64 weight rows per block, 2 blocks per CU
    (8, 640, 640, 32, 32, True, 1),      # W = 32, residual + SiLU epilogue
    (8, 1280, 1280, 16, 16, True, 2),    # W = 16: split along the channel slices (fp32 slabs + reduce), per-frame row vector
    (3, 96, 64, 16, 16, False, 2),       # Cin a multiple of 32 only, strided (concat-slice) input
    (1, 320, 320, 24, 16, True, 0),      # non-square frame
    (2, 2560, 1280, 16, 16, True, 0),    # deepest K of the UNet
    (1, 32, 40, 16, 16, False, 0)])      # single slice, ragged N
def test_igemm_conv3x3_patch(hip, n, cin, cout, h, w, ti, epi):
    """the patch-staged 3x3 conv (stride 1, pad 1, W in {16, 32, 64}): both [N, K] weight orders and the tiled layout
    (tap_inner = 2) the engine feeds it; the tiled launch must be bit-identical to the [N, K] launch"""
    from mgld_vsr_amd.engine import pack_conv3x3, tile_conv3p
    hip.set_workspace(hip._test_ws)
    x = h16(rnd(n, cin, h, w, seed=70))
    wt = h16(rnd(cout, cin, 3, 3, seed=71, scale=(9 * cin) ** -0.5))
    b = rnd(cout, seed=72)
    ref = F.conv2d(x.float(), wt.float(), b, padding=1)
    wk = (pack_conv3x3(wt, tap_inner=True) if ti else wt.permute(0, 2, 3, 1).reshape(cout, 9 * cin).contiguous()).to(DEV)
    xt = _to_tok(x).to(DEV)
    kw = {}
    if epi == 1:
        r = h16(rnd(n * h * w, cout, seed=73))
        kw = dict(resid=r.to(DEV), act=hip.ACT_SILU, alpha=0.5, beta=2.0)
        ref = 0.5 * F.silu(ref) + 2.0 * _from_tok(r.float(), n, h, w)
    elif epi == 2:
        emb = rnd(n, cout, seed=74)
        kw = dict(rowvec=emb.to(DEV), rows_per_frame=h * w)
        ref = ref + emb[:, :, None, None]
        big = torch.zeros(n * h * w, cin + 64, dtype=torch.half, device=DEV)
        big[:, 32:32 + cin] = xt
        xt = big[:, 32:32 + cin]
    assert hip.conv3p_applies(n, cin, cout, h, w)
    conv = (cin, h, w, h, w, 1, 1, 1, 0)
    out = torch.empty(n * h * w, cout, dtype=torch.half, device=DEV)
    hip.igemm(xt, wk, out, mode=hip.MODE_CONV3X3, bias=b.to(DEV), conv=conv, tap_inner=1 if ti else 0, **kw)
    out2 = torch.empty_like(out)
    hip.igemm(xt, tile_conv3p(wk, cin, ti), out2, mode=hip.MODE_CONV3X3, bias=b.to(DEV), conv=conv, tap_inner=2, N=cout,
              K=9 * cin, **kw)
    torch.cuda.synchronize()
    assert rel_l2(_from_tok(out.cpu().float(), n, h, w), ref) < 1e-3
    # the tiled launch runs the 2-D-tile kernel: same products, same per-output summation order unless the K split differs
    assert rel_l2(_from_tok(out2.cpu().float(), n, h, w), ref) < 1e-3 and rel_l2(out2.float(), out.float()) < 3e-4


@pytest.mark.parametrize("variant", [0, 1, 2, 3, 4, 5, 7])
@pytest.mark.parametrize("n,cin,cout,h,w,epi", [
    (2, 64, 96, 24, 40, 0),        # ragged in both directions for every tile shape, ragged N
    (1, 128, 128, 136, 144, 1),    # W > 64 (the VAE's large levels), residual + SiLU epilogue
    (3, 320, 320, 64, 64, 2),      # the UNet's dominant shape, per-frame row vector, strided (concat-slice) input
    (1, 32, 64, 16, 16, 0)])       # single slice, one tile
def test_igemm_conv3x3_tile2d(hip, variant, n, cin, cout, h, w, epi):
    """conv3q: the patch-staged 3x3 conv on 2-D pixel tiles, every kernel variant (tune = id + 1) on square, ragged and
    W > 64 frames vs torch's conv2d on the same fp16 operands"""
    from mgld_vsr_amd.engine import tile_conv3p
    hip.set_workspace(hip._test_ws)
    x = h16(rnd(n, cin, h, w, seed=170))
    wt = h16(rnd(cout, cin, 3, 3, seed=171, scale=(9 * cin) ** -0.5))
    b = rnd(cout, seed=172)
    ref = F.conv2d(x.float(), wt.float(), b, padding=1)
    wk = wt.permute(0, 2, 3, 1).reshape(cout, 9 * cin).contiguous().to(DEV)
    xt = _to_tok(x).to(DEV)
    kw = {}
    if epi == 1:
        r = h16(rnd(n * h * w, cout, seed=173))
        kw = dict(resid=r.to(DEV), act=hip.ACT_SILU, alpha=0.5, beta=2.0)
        ref = 0.5 * F.silu(ref) + 2.0 * _from_tok(r.float(), n, h, w)
    elif epi == 2:
        emb = rnd(n, cout, seed=174)
        kw = dict(rowvec=emb.to(DEV), rows_per_frame=h * w)
        ref = ref + emb[:, :, None, None]
        big = torch.zeros(n * h * w, cin + 64, dtype=torch.half, device=DEV)
        big[:, 32:32 + cin] = xt
        xt = big[:, 32:32 + cin]
    assert hip.conv3p_applies(n, cin, cout, h, w)
    out = torch.full((n * h * w, cout), float("nan"), dtype=torch.half, device=DEV)
    hip.igemm(xt, tile_conv3p(wk, cin, False), out, mode=hip.MODE_CONV3X3, bias=b.to(DEV), conv=(cin, h, w, h, w, 1, 1, 1, 0),
              tap_inner=2, N=cout, K=9 * cin, tune=variant + 1, **kw)
    torch.cuda.synchronize()
    assert rel_l2(_from_tok(out.cpu().float(), n, h, w), ref) < 1e-3


R3_BN = {0: 160, 1: 320, 2: 128, 3: 256, 4: 160, 5: 160, 6: 80, 7: 80, 8: 128}      # weight rows per block of the ping-pong patch conv configurations (csrc/conv3r.hip)


@pytest.mark.parametrize("cfg", sorted(R3_BN))
@pytest.mark.parametrize("n,cin,nt,h,w,epi", [
    (2, 64, 1, 24, 40, 0),         # ragged in both directions for every tile shape (zero-filled halo AND ragged tiles)
    (1, 128, 2, 72, 80, 1),        # W > 64, residual + SiLU epilogue, two column tiles
    (3, 320, 2, 64, 64, 2),        # the UNet's dominant shape, per-frame row vector, strided (concat-slice) input
    (1, 32, 1, 16, 16, 0),         # single slice (the ring is deeper than the problem), one tile
    (2, 96, 1, 8, 16, 3),          # three slices, nothing in the epilogue
    (2, 64, 1, 16, 32, 4)])        # ReLU (SPADE's shared convolution)
def test_igemm_conv3x3_pingpong(hip, cfg, n, cin, nt, h, w, epi):
    """conv3r: the ping-pong patch conv, every configuration (tune = 31 + id) vs torch's conv2d on the same fp16 operands"""
    from mgld_vsr_amd.engine import tile_conv3p
    cout = nt * R3_BN[cfg]
    x = h16(rnd(n, cin, h, w, seed=270))
    wt = h16(rnd(cout, cin, 3, 3, seed=271, scale=(9 * cin) ** -0.5))
    b = rnd(cout, seed=272)
    ref = F.conv2d(x.float(), wt.float(), b if epi != 3 else None, padding=1)
    wk = wt.permute(0, 2, 3, 1).reshape(cout, 9 * cin).contiguous().to(DEV)
    xt = _to_tok(x).to(DEV)
    kw = dict(bias=b.to(DEV)) if epi != 3 else {}
    if epi == 1:
        r = h16(rnd(n * h * w, cout, seed=273))
        kw.update(resid=r.to(DEV), act=hip.ACT_SILU, alpha=0.5, beta=2.0)
        ref = 0.5 * F.silu(ref) + 2.0 * _from_tok(r.float(), n, h, w)
    elif epi == 2:
        emb = rnd(n, cout, seed=274)
        kw.update(rowvec=emb.to(DEV), rows_per_frame=h * w)
        ref = ref + emb[:, :, None, None]
        big = torch.zeros(n * h * w, cin + 64, dtype=torch.half, device=DEV)
        big[:, 32:32 + cin] = xt
        xt = big[:, 32:32 + cin]
    elif epi == 4:
        kw.update(act=hip.ACT_RELU)
        ref = F.relu(ref)
    p = hip.MgldIGemm()
    p.mode, p.M, p.N, p.K, p.batch, p.tap_inner, p.tune = hip.MODE_CONV3X3, n * h * w, cout, 9 * cin, 1, 2, 31 + cfg
    p.Cin, p.Hin, p.Win, p.Hout, p.Wout, p.stride, p.pad_t, p.pad_l, p.up2 = cin, h, w, h, w, 1, 1, 1, 0
    p.lda, p.ldc = cin, cout
    assert hip.igemm_config(p) == 600000 + cfg
    out = torch.full((n * h * w, cout), float("nan"), dtype=torch.half, device=DEV)
    hip.igemm(xt, tile_conv3p(wk, cin, False), out, mode=hip.MODE_CONV3X3, conv=(cin, h, w, h, w, 1, 1, 1, 0),
              tap_inner=2, N=cout, K=9 * cin, tune=31 + cfg, **kw)
    torch.cuda.synchronize()
    assert rel_l2(_from_tok(out.cpu().float(), n, h, w), ref) < 1e-3


@pytest.mark.parametrize("cfg,n,cin,nt,h,w", [(0, 2, 64, 1, 16, 32), (8, 1, 128, 2, 32, 64), (2, 3, 32, 1, 24, 40), (6, 1, 256, 2, 16, 32)])
def test_igemm_conv3x3_pingpong_w2(hip, cfg, n, cin, nt, h, w):
    """conv3r with the weight-residual pass (MgldIGemm.W2: every slice visited twice, the accumulators scaled in between): against
    fp64 on the fp32 weights it must be far closer than the one-pass product of the fp16-rounded weights, and agree with conv3q's
    residual pass on the same operands"""
    from mgld_vsr_amd.engine import split_residual, tile_conv3p
    hip.set_workspace(hip._test_ws)
    cout = nt * R3_BN[cfg]
    x = h16(rnd(n, cin, h, w, seed=290))
    w32 = rnd(cout, cin, 3, 3, seed=291, scale=(9 * cin) ** -0.5)
    ref = F.conv2d(x.double(), w32.double(), None, padding=1)
    hi_, lo_ = split_residual(w32.permute(0, 2, 3, 1).reshape(cout, 9 * cin).contiguous())
    whi, wlo = tile_conv3p(hi_.to(DEV), cin, False), tile_conv3p(lo_.to(DEV), cin, False)
    xt = _to_tok(x).to(DEV)
    errs, outs = [], []
    for tune, w2 in ((31 + cfg, None), (31 + cfg, wlo), (5, wlo)):
        out = torch.empty(n * h * w, cout, dtype=torch.half, device=DEV)
        hip.igemm(xt, whi, out, mode=hip.MODE_CONV3X3, conv=(cin, h, w, h, w, 1, 1, 1, 0), tap_inner=2, N=cout, K=9 * cin, tune=tune, w2=w2)
        torch.cuda.synchronize()
        outs.append(out.cpu().float())
        errs.append(rel_l2(_from_tok(outs[-1], n, h, w), ref))
    assert errs[1] < errs[0] and errs[1] < 4.5e-4, errs        # (fp16 output rounding alone is ~2.9e-4)
    assert rel_l2(outs[1], outs[2]) < 2e-4


@pytest.mark.parametrize("cfg,n,cin,nt,h,w,epi", [(0, 2, 64, 2, 16, 32, 2), (8, 2, 128, 5, 32, 64, 1), (2, 3, 32, 1, 24, 40, 0), (6, 2, 64, 4, 20, 36, 2),
                                                  (3, 1, 64, 1, 16, 64, 0), (5, 2, 32, 2, 8, 16, 1), (7, 2, 32, 4, 16, 40, 2), (1, 2, 32, 1, 16, 24, 0)])
def test_igemm_conv3x3_pingpong_gn_stats(hip, cfg, n, cin, nt, h, w, epi):
    """conv3r with MgldIGemm.gn_part: same output bits as without; the per-tile channel sums add up to the moments of the stored
    output (ragged tiles included; sums are taken before the fp16 rounding); GroupNorm fed by them (mgld_gn_apply2, MGLD_GN_CHANNEL_SUMS) agrees with the mgld_gn_stats path"""
    from mgld_vsr_amd.engine import tile_conv3p
    cout = nt * R3_BN[cfg]
    x = h16(rnd(n, cin, h, w, seed=300))
    wt = h16(rnd(cout, cin, 3, 3, seed=301, scale=(9 * cin) ** -0.5))
    wk = tile_conv3p(wt.permute(0, 2, 3, 1).reshape(cout, 9 * cin).contiguous().to(DEV), cin, False)
    xt = _to_tok(x).to(DEV)
    kw = dict(bias=(rnd(cout, seed=302) + 0.5).to(DEV))
    if epi == 1:
        kw.update(resid=h16(rnd(n * h * w, cout, seed=303)).to(DEV), act=hip.ACT_SILU, alpha=0.5, beta=2.0)
    elif epi == 2:
        kw.update(rowvec=rnd(n, cout, seed=304).to(DEV), rows_per_frame=h * w)
    common = dict(mode=hip.MODE_CONV3X3, conv=(cin, h, w, h, w, 1, 1, 1, 0), tap_inner=2, N=cout, K=9 * cin, tune=31 + cfg, **kw)
    plain = torch.empty(n * h * w, cout, dtype=torch.half, device=DEV)
    hip.igemm(xt, wk, plain, **common)
    got = []

    def part(chunks):
        got.append((torch.full((n * chunks, 2, cout), float("nan"), dtype=torch.float32, device=DEV), chunks))
        return got[0][0]
    out = torch.empty_like(plain)
    hip.igemm(xt, wk, out, gn_part=part, **common)
    torch.cuda.synchronize()
    assert got, "the ping-pong patch convolution must offer the statistics output"
    pt, chunks = got[0]
    assert torch.equal(out, plain)
    o = out.double().reshape(n, h * w, cout)
    tot = pt.double().reshape(n, chunks, 2, cout).sum(1)
    assert torch.isfinite(pt).all()
    # (sums are taken before the fp16 rounding: the first moment is compared on the scale of the rms, not of the — possibly ~0 — mean)
    rms = ((o * o).sum(1) / (h * w)).sqrt()
    assert (((tot[:, 0] - o.sum(1)) / (h * w)).abs() / rms).max() < 1e-4 and rel_l2(tot[:, 1], (o * o).sum(1)) < 1e-4
    if cout % 32 == 0:
        rows = h * w
        gamma, beta = (1 + 0.1 * rnd(cout, seed=305)).to(DEV), (0.1 * rnd(cout, seed=306)).to(DEV)
        gs = torch.empty(n, hip.gn_chunks(rows), 32, 2, dtype=torch.float64, device=DEV)
        hip.gn_stats(out, n, rows, 32, gs)
        y0, y1 = torch.empty_like(out), torch.empty_like(out)
        hip.gn_apply(out, gs, 1e-5, gamma, beta, y0, n, rows, 32, True)
        hip.gn_apply(out, pt, 1e-5, gamma, beta, y1, n, rows, 32, True, kind=hip.GN_CHANNEL_SUMS, chunks=chunks)
        torch.cuda.synchronize()
        assert rel_l2(y1.float(), y0.float()) < 1e-4


@pytest.mark.parametrize("cfg,n,cin,nt,h,w,epi,two", [(5, 8, 1280, 8, 16, 16, 2, False), (8, 8, 640, 10, 16, 16, 1, False), (4, 3, 320, 2, 16, 16, 0, False),
                                                      (5, 2, 256, 1, 16, 24, 1, True), (8, 5, 2560, 10, 16, 16, 0, False)])
def test_igemm_conv3x3_pingpong_ksplit(hip, cfg, n, cin, nt, h, w, epi, two):
    """conv3r with the channel slices split over grid.z (tune = 50 + id: the 16^2 UNet level, too few tiles for the chip): raw fp32 slabs +
    splitk_reduce == the unsplit kernel to the summation order, against conv2d on the same fp16 operands; with the weight-residual pass too"""
    from mgld_vsr_amd.engine import split_residual, tile_conv3p
    hip.set_workspace(hip._test_ws)
    cout = nt * R3_BN[cfg]
    x = h16(rnd(n, cin, h, w, seed=320))
    w32 = rnd(cout, cin, 3, 3, seed=321, scale=(9 * cin) ** -0.5)
    hi_, lo_ = split_residual(w32.permute(0, 2, 3, 1).reshape(cout, 9 * cin).contiguous())
    wk = tile_conv3p(hi_.to(DEV), cin, False)
    w2 = tile_conv3p(lo_.to(DEV), cin, False) if two else None
    b = rnd(cout, seed=322)
    wref = (w32 if two else hi_.reshape(cout, 3, 3, cin).permute(0, 3, 1, 2)).double()
    ref = F.conv2d(x.double(), wref, b.double(), padding=1)
    xt = _to_tok(x).to(DEV)
    kw = dict(bias=b.to(DEV), w2=w2)
    if epi == 1:
        r = h16(rnd(n * h * w, cout, seed=323))
        kw.update(resid=r.to(DEV), act=hip.ACT_SILU, alpha=0.5, beta=2.0)
        ref = 0.5 * F.silu(ref) + 2.0 * _from_tok(r.double(), n, h, w)
    elif epi == 2:
        emb = rnd(n, cout, seed=324)
        kw.update(rowvec=emb.to(DEV), rows_per_frame=h * w)
        ref = ref + emb.double()[:, :, None, None]
    p = hip.MgldIGemm()
    p.mode, p.M, p.N, p.K, p.batch, p.tap_inner, p.tune = hip.MODE_CONV3X3, n * h * w, cout, 9 * cin, 1, 2, 50 + cfg
    p.Cin, p.Hin, p.Win, p.Hout, p.Wout, p.stride, p.pad_t, p.pad_l, p.up2 = cin, h, w, h, w, 1, 1, 1, 0
    p.lda, p.ldc = cin, cout
    code = hip.igemm_config(p)
    assert code % 1000000 == 600000 + cfg and code // 1000000 >= 2, code       # split taken
    name, splits = hip.igemm_kernel_name(p)
    assert name.endswith("true>") and splits == code // 1000000, (name, splits)
    common = dict(mode=hip.MODE_CONV3X3, conv=(cin, h, w, h, w, 1, 1, 1, 0), tap_inner=2, N=cout, K=9 * cin, **kw)
    outs = [torch.full((n * h * w, cout), float("nan"), dtype=torch.half, device=DEV) for _ in range(3)]
    for o in outs:
        hip.igemm(xt, wk, o, tune=50 + cfg, **common)
    one = torch.empty_like(outs[0])
    hip.igemm(xt, wk, one, tune=31 + cfg, **common)
    torch.cuda.synchronize()
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    assert rel_l2(_from_tok(outs[0].cpu().double(), n, h, w), ref) < (4.5e-4 if two else 1e-3)
    assert rel_l2(outs[0].float(), one.float()) < 3e-4


@pytest.mark.parametrize("cfg,n,cin,nt,epi,split", [(9, 8, 1280, 8, 2, True), (10, 8, 1280, 8, 1, True), (9, 5, 320, 2, 0, True), (10, 3, 640, 1, 2, True),
                                                    (9, 8, 64, 2, 1, False), (10, 2, 32, 1, 0, False), (10, 1, 2560, 8, 0, True)])
def test_igemm_conv3x3_pingpong_frame_stacked(hip, cfg, n, cin, nt, epi, split):
    """conv3r on the 8 x 8 level: tiles of 4 (configuration 9) or 2 (10) whole frames stacked (two image rows per fragment, frames that do
    not exist masked), with and without the K split, against conv2d on the same fp16 operands and against conv3q; bit-repeatable"""
    from mgld_vsr_amd.engine import tile_conv3p
    hip.set_workspace(hip._test_ws)
    h = w = 8
    cout = nt * 160
    x = h16(rnd(n, cin, h, w, seed=330))
    wt = h16(rnd(cout, cin, 3, 3, seed=331, scale=(9 * cin) ** -0.5))
    b = rnd(cout, seed=332)
    ref = F.conv2d(x.double(), wt.double(), b.double(), padding=1)
    wk = tile_conv3p(wt.permute(0, 2, 3, 1).reshape(cout, 9 * cin).contiguous().to(DEV), cin, False)
    xt = _to_tok(x).to(DEV)
    kw = dict(bias=b.to(DEV))
    if epi == 1:
        r = h16(rnd(n * h * w, cout, seed=333))
        kw.update(resid=r.to(DEV), act=hip.ACT_SILU, alpha=0.5, beta=2.0)
        ref = 0.5 * F.silu(ref) + 2.0 * _from_tok(r.double(), n, h, w)
    elif epi == 2:
        emb = rnd(n, cout, seed=334)
        kw.update(rowvec=emb.to(DEV), rows_per_frame=h * w)
        ref = ref + emb.double()[:, :, None, None]
    tune = (50 if split else 31) + cfg
    p = hip.MgldIGemm()
    p.mode, p.M, p.N, p.K, p.batch, p.tap_inner, p.tune = hip.MODE_CONV3X3, n * h * w, cout, 9 * cin, 1, 2, tune
    p.Cin, p.Hin, p.Win, p.Hout, p.Wout, p.stride, p.pad_t, p.pad_l, p.up2 = cin, h, w, h, w, 1, 1, 1, 0
    p.lda, p.ldc = cin, cout
    code = hip.igemm_config(p)
    assert code % 1000000 == 600000 + cfg and (code // 1000000 >= 2) == (split and cin >= 128), code
    common = dict(mode=hip.MODE_CONV3X3, conv=(cin, h, w, h, w, 1, 1, 1, 0), tap_inner=2, N=cout, K=9 * cin, **kw)
    outs = [torch.full((n * h * w, cout), float("nan"), dtype=torch.half, device=DEV) for _ in range(3)]
    for o in outs:
        hip.igemm(xt, wk, o, tune=tune, **common)
    old = torch.empty_like(outs[0])
    hip.igemm(xt, wk, old, tune=5, **common)           # conv3q
    torch.cuda.synchronize()
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    assert rel_l2(_from_tok(outs[0].cpu().double(), n, h, w), ref) < 1e-3
    assert rel_l2(outs[0].float(), old.float()) < 5e-4


def test_igemm_conv3x3_pingpong_race_screen(hip):
    """counted waits of the weight ring / patch double buffer: repeated launches of a deep-K problem give the same bits on every
    configuration, and those bits agree with conv3q's (same products, another summation order) to the fp16 output rounding"""
    from mgld_vsr_amd.engine import tile_conv3p
    hip.set_workspace(hip._test_ws)
    n, cin, h, w = 4, 640, 32, 64
    x = h16(rnd(n, cin, h, w, seed=280))
    xt = _to_tok(x).to(DEV)
    for cfg, bn in R3_BN.items():
        cout = 2 * bn
        wt = h16(rnd(cout, cin, 3, 3, seed=281, scale=(9 * cin) ** -0.5))
        wk = tile_conv3p(wt.permute(0, 2, 3, 1).reshape(cout, 9 * cin).contiguous().to(DEV), cin, False)
        outs = [torch.empty(n * h * w, cout, dtype=torch.half, device=DEV) for _ in range(5)]
        for o in outs:
            hip.igemm(xt, wk, o, mode=hip.MODE_CONV3X3, conv=(cin, h, w, h, w, 1, 1, 1, 0), tap_inner=2, N=cout, K=9 * cin, tune=31 + cfg)
        old = torch.empty(n * h * w, cout, dtype=torch.half, device=DEV)
        hip.igemm(xt, wk, old, mode=hip.MODE_CONV3X3, conv=(cin, h, w, h, w, 1, 1, 1, 0), tap_inner=2, N=cout, K=9 * cin, tune=5)
        torch.cuda.synchronize()
        for o in outs[1:]:
            assert torch.equal(o, outs[0]), f"configuration {cfg}: launches differ"
        assert rel_l2(outs[0].cpu().float(), old.cpu().float()) < 5e-4


@pytest.mark.parametrize("variant", [0, 1, 7])
@pytest.mark.parametrize("n,cin,cout,h,w", [(2, 64, 96, 12, 20), (1, 256, 256, 64, 64), (2, 1280, 1280, 8, 8)])
def test_igemm_conv3x3_tile2d_upsample(hip, variant, n, cin, cout, h, w):
    """conv3q with the nearest-2x upsample folded into the tap offsets (Upsample blocks: openaimodel.py:185, model.py:96) vs
    conv2d(interpolate(x, 2, nearest)); (h, w) = the low-resolution input"""
    from mgld_vsr_amd.engine import tile_conv3p
    hip.set_workspace(hip._test_ws)
    x = h16(rnd(n, cin, h, w, seed=180))
    wt = h16(rnd(cout, cin, 3, 3, seed=181, scale=(9 * cin) ** -0.5))
    b = rnd(cout, seed=182)
    ref = F.conv2d(F.interpolate(x.float(), scale_factor=2, mode="nearest"), wt.float(), b, padding=1)
    wk = wt.permute(0, 2, 3, 1).reshape(cout, 9 * cin).contiguous().to(DEV)
    assert hip.conv3p_applies(n, cin, cout, h, w, True)
    out = torch.full((n * 4 * h * w, cout), float("nan"), dtype=torch.half, device=DEV)
    hip.igemm(_to_tok(x).to(DEV), tile_conv3p(wk, cin, False), out, mode=hip.MODE_CONV3X3, bias=b.to(DEV),
              conv=(cin, h, w, 2 * h, 2 * w, 1, 1, 1, 1), tap_inner=2, N=cout, K=9 * cin, tune=variant + 1)
    torch.cuda.synchronize()
    assert rel_l2(_from_tok(out.cpu().float(), n, 2 * h, 2 * w), ref) < 1e-3


@pytest.mark.parametrize("n,cin,cout,epi", [(8, 1280, 1280, 2), (3, 512, 512, 1), (2, 256, 128, 0), (1, 128, 2560, 0), (5, 64, 96, 0)])
def test_igemm_conv3x3_tile2d_8x8(hip, n, cin, cout, epi):
    """conv3q on the 8x8 UNet level (one 8x8 tile per frame, 128 weight rows, K split over the channel slices) vs conv2d"""
    from mgld_vsr_amd.engine import tile_conv3p
    hip.set_workspace(hip._test_ws)
    h = w = 8
    x = h16(rnd(n, cin, h, w, seed=190))
    wt = h16(rnd(cout, cin, 3, 3, seed=191, scale=(9 * cin) ** -0.5))
    b = rnd(cout, seed=192)
    ref = F.conv2d(x.float(), wt.float(), b, padding=1)
    wk = wt.permute(0, 2, 3, 1).reshape(cout, 9 * cin).contiguous().to(DEV)
    kw = {}
    if epi == 1:
        r = h16(rnd(n * h * w, cout, seed=193))
        kw = dict(resid=r.to(DEV), act=hip.ACT_SILU, alpha=0.5, beta=2.0)
        ref = 0.5 * F.silu(ref) + 2.0 * _from_tok(r.float(), n, h, w)
    elif epi == 2:
        emb = rnd(n, cout, seed=194)
        kw = dict(rowvec=emb.to(DEV), rows_per_frame=h * w)
        ref = ref + emb[:, :, None, None]
    assert hip.conv3p_applies(n, cin, cout, h, w)
    out = torch.full((n * h * w, cout), float("nan"), dtype=torch.half, device=DEV)
    hip.igemm(_to_tok(x).to(DEV), tile_conv3p(wk, cin, False), out, mode=hip.MODE_CONV3X3, bias=b.to(DEV), conv=(cin, h, w, h, w, 1, 1, 1, 0),
              tap_inner=2, N=cout, K=9 * cin, **kw)
    torch.cuda.synchronize()
    assert rel_l2(_from_tok(out.cpu().float(), n, h, w), ref) < 1e-3


def test_igemm_tiled_weights_rejected_off_the_patch_path(hip):
    """tap_inner = 2 is only defined for problems the patch kernel takes: anything else must fail loudly"""
    from mgld_vsr_amd.engine import tile_conv3p
    n, cin, cout, h, w = 1, 64, 64, 4, 4            # 4x4 frames: im2col path
    assert not hip.conv3p_applies(n, cin, cout, h, w)
    x = _to_tok(h16(rnd(n, cin, h, w, seed=75))).to(DEV)
    wk = h16(rnd(cout, 9 * cin, seed=76)).to(DEV)
    out = torch.empty(n * h * w, cout, dtype=torch.half, device=DEV)
    with pytest.raises(RuntimeError):
        hip.igemm(x, tile_conv3p(wk, cin, False), out, mode=hip.MODE_CONV3X3, conv=(cin, h, w, h, w, 1, 1, 1, 0), tap_inner=2,
                  N=cout, K=9 * cin)


def test_igemm_conv_rowvec(hip):
    n, cin, cout, h, w = 3, 32, 64, 8, 8
    x = h16(rnd(n, cin, h, w, seed=23))
    wt = h16(rnd(cout, cin, 3, 3, seed=24, scale=(9 * cin) ** -0.5))
    b, emb = rnd(cout, seed=25), rnd(n, cout, seed=26)
    ref = F.conv2d(x.float(), wt.float(), b, padding=1) + emb[:, :, None, None]
    out = torch.empty(n * h * w, cout, dtype=torch.half, device=DEV)
    wk = wt.permute(0, 2, 3, 1).reshape(cout, 9 * cin).contiguous().to(DEV)
    hip.igemm(_to_tok(x).to(DEV), wk, out, mode=hip.MODE_CONV3X3, bias=b.to(DEV), rowvec=emb.to(DEV), rows_per_frame=h * w,
              conv=(cin, h, w, h, w, 1, 1, 1, 0))
    assert rel_l2(_from_tok(out.cpu().float(), n, h, w), ref) < 1e-3


@pytest.mark.parametrize("clips,T,c", [(1, 5, 64), (2, 3, 64), (1, 1, 64), (2, 4, 384)])
def test_igemm_tconv(hip, clips, T, c):
    h, w = 6, 6
    x = h16(rnd(clips * T, c, h, w, seed=27))
    wt = h16(rnd(c, c, 3, 1, 1, seed=28, scale=(3 * c) ** -0.5))
    b = rnd(c, seed=29)
    alpha = 0.37
    x5 = x.float().reshape(clips, T, c, h, w).permute(0, 2, 1, 3, 4)
    res = F.conv3d(x5, wt.float(), b, padding=(1, 0, 0)).permute(0, 2, 1, 3, 4).reshape(clips * T, c, h, w)
    ref = alpha * res + (1 - alpha) * x.float()
    xt = _to_tok(x).to(DEV)
    wk = wt[:, :, :, 0, 0].permute(0, 2, 1).reshape(c, 3 * c).contiguous().to(DEV)
    out = torch.empty_like(xt)
    hip.igemm(xt, wk, out, mode=hip.MODE_TCONV3, bias=b.to(DEV), resid=xt, alpha=alpha, beta=1 - alpha, tconv=(c, T, h * w))
    assert rel_l2(_from_tok(out.cpu().float(), clips * T, h, w), ref) < 1e-3


@pytest.mark.parametrize("clips,T,c,h,w", [(2, 8, 128, 8, 8), (1, 4, 64, 16, 8), (3, 8, 96, 4, 4), (1, 2, 256, 8, 8), (1, 8, 1280, 8, 8)])
def test_igemm_tconv_frame_interleaved_rows(hip, clips, T, c, h, w):
    """T a power of two and whole tiles per frame: the launcher hands a tile the same pixels of EVERY frame (the three temporal taps
    re-read one set of rows; csrc/igemm.hip `tconv_rows_lg`, default for frames >= 24 MB, forced here by tune = 14).  Same arithmetic
    per output element: bit-identical to tiles of consecutive rows (tune = 15), FAST and per-lane paths, split-K included; and both match Conv3d."""
    hip.set_workspace(hip._test_ws)
    x = h16(rnd(clips * T, c, h, w, seed=331))
    wt = h16(rnd(c, c, 3, 1, 1, seed=332, scale=(3 * c) ** -0.5))
    b = rnd(c, seed=333)
    alpha = 0.6
    x5 = x.float().reshape(clips, T, c, h, w).permute(0, 2, 1, 3, 4)
    res = F.conv3d(x5, wt.float(), b, padding=(1, 0, 0)).permute(0, 2, 1, 3, 4).reshape(clips * T, c, h, w)
    ref = alpha * res + (1 - alpha) * x.float()
    xt = _to_tok(x).to(DEV)
    wk = wt[:, :, :, 0, 0].permute(0, 2, 1).reshape(c, 3 * c).contiguous().to(DEV)
    outs = []
    for tune in (14, 15):             # 14: interleaved whatever the frame size (the launcher's own threshold is 24 MB per frame)
        out = torch.full_like(xt, float("nan"))
        hip.igemm(xt, wk, out, mode=hip.MODE_TCONV3, bias=b.to(DEV), resid=xt, alpha=alpha, beta=1 - alpha, tconv=(c, T, h * w), tune=tune)
        torch.cuda.synchronize()
        outs.append(out.cpu())
    assert torch.equal(outs[0], outs[1])
    assert rel_l2(_from_tok(outs[0].float(), clips * T, h, w), ref) < 1e-3


@pytest.mark.parametrize("clips,T,c,h,w,w2", [(1, 8, 128, 8, 8, False), (2, 8, 256, 16, 8, False), (3, 4, 128, 8, 8, False), (1, 4, 512, 16, 16, True),
                                              (2, 8, 128, 32, 32, True), (1, 8, 640, 8, 4, False)])
def test_igemm_tconv_pingpong(hip, clips, T, c, h, w, w2):
    """pptconv (tune = 40): the temporal Conv3d on ping-pong tiles of 256 / T pixels x all T frames — first / last frame of every clip
    (the zero padding comes from lanes pushed out of the DMA descriptor's range), several clips, both tile widths, K order (tap, Cin),
    the blend epilogue and the weight-residual pass; against Conv3d on the same fp16 operands (W2: on the fp32 weights)."""
    from mgld_vsr_amd.engine import split_residual
    x = h16(rnd(clips * T, c, h, w, seed=341))
    w32 = rnd(c, c, 3, 1, 1, seed=342, scale=(3 * c) ** -0.5)
    b = rnd(c, seed=343)
    alpha = 0.6
    wk32 = w32[:, :, :, 0, 0].permute(0, 2, 1).reshape(c, 3 * c).contiguous()
    hi_, lo_ = split_residual(wk32)
    wref = w32.double() if w2 else hi_.double().reshape(c, 3, c).permute(0, 2, 1)[..., None, None]
    x5 = x.double().reshape(clips, T, c, h, w).permute(0, 2, 1, 3, 4)
    res = F.conv3d(x5, wref, b.double(), padding=(1, 0, 0)).permute(0, 2, 1, 3, 4).reshape(clips * T, c, h, w)
    ref = alpha * res + (1 - alpha) * x.double()
    xt = _to_tok(x).to(DEV)
    p = hip.MgldIGemm()
    p.mode, p.M, p.N, p.K, p.batch, p.tune, p.Cin, p.T, p.HW = hip.MODE_TCONV3, clips * T * h * w, c, 3 * c, 1, 40, c, T, h * w
    p.lda = p.ldw = p.ldc = 8
    assert hip.igemm_config(p) // 100000 == 7
    outs = []
    for _ in range(3):
        out = torch.full_like(xt, float("nan"))
        hip.igemm(xt, hi_.to(DEV), out, mode=hip.MODE_TCONV3, bias=b.to(DEV), resid=xt, alpha=alpha, beta=1 - alpha, tconv=(c, T, h * w), tune=40,
                  w2=lo_.to(DEV) if w2 else None)
        torch.cuda.synchronize()
        outs.append(out.cpu())
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    assert rel_l2(_from_tok(outs[0].float(), clips * T, h, w), ref) < (4.5e-4 if w2 else 1e-3)


def test_igemm_skinny_output_splits_k(hip):
    """N <= 32 over a deep K with fewer row tiles than 2 x CUs: the launcher splits K (csrc/igemm.hip choose()); result vs conv2d"""
    hip.set_workspace(hip._test_ws)
    n, cin, cout, h, w = 2, 320, 4, 32, 32
    x = h16(rnd(n, cin, h, w, seed=341))
    wt = h16(rnd(cout, cin, 3, 3, seed=342, scale=(9 * cin) ** -0.5))
    b = rnd(cout, seed=343)
    ref = F.conv2d(x.float(), wt.float(), b, padding=1)
    wk = wt.permute(0, 2, 3, 1).reshape(cout, 9 * cin).contiguous().to(DEV)
    out = torch.full((n * h * w, cout), float("nan"), dtype=torch.half, device=DEV)
    hip.IGEMM_LOG = []
    try:
        hip.igemm(_to_tok(x).to(DEV), wk, out, mode=hip.MODE_CONV3X3, bias=b.to(DEV), conv=(cin, h, w, h, w, 1, 1, 1, 0))
        cfg = hip.igemm_config(hip.IGEMM_LOG[0])
    finally:
        hip.IGEMM_LOG = None
    assert cfg % 1000000 == 128032 and cfg // 1000000 >= 2, cfg
    assert rel_l2(_from_tok(out.cpu().float(), n, h, w), ref) < 1e-3


# ---- W2: second MFMA pass on the fp16 rounding residual of the weights (MgldIGemm.W2) -------------------------------------
def _w2_check(err_hi, err_w2):
    """fp32 weights are the truth; activations are exact fp16.  With fp16 weights the product carries their rounding (~2.3e-4 of the
    output for Gaussian operands); the residual pass removes it down to the 2^-22 level."""
    assert err_hi > 1.2e-4, err_hi                     # the plain launch does show the weight rounding (the test can see the effect)
    assert err_w2 < 2e-5 and err_w2 < 0.15 * err_hi, (err_hi, err_w2)


@pytest.mark.parametrize("M,N,K", [(256, 320, 320), (4096, 1280, 320), (100, 72, 40), (8192, 640, 2560), (512, 1280, 5120), (33, 4, 64)])
def test_igemm_w2_linear(hip, M, N, K):
    """LINEAR (fast and general path, 128x128 / 64x128 / 64x64 / 128x32 tiles, split-K): acc = w2_scale * A W2^T + A W^T"""
    from mgld_vsr_amd.engine import split_residual
    hip.set_workspace(hip._test_ws)
    a, w32, b = h16(rnd(M, K, seed=301)), rnd(N, K, seed=302, scale=K ** -0.5), rnd(N, seed=303)
    ref = a.double() @ w32.double().t() + b.double()
    hi, lo = split_residual(w32)
    out_hi = torch.empty(M, N, dtype=torch.float32, device=DEV)
    out_w2 = torch.empty(M, N, dtype=torch.float32, device=DEV)
    hip.igemm(a.to(DEV), hi.to(DEV), out_hi, bias=b.to(DEV))
    hip.igemm(a.to(DEV), hi.to(DEV), out_w2, bias=b.to(DEV), w2=lo.to(DEV))
    torch.cuda.synchronize()
    _w2_check(rel_l2(out_hi.cpu(), ref), rel_l2(out_w2.cpu(), ref))


@pytest.mark.parametrize("n,cin,cout,h,w,up2", [
    (2, 128, 128, 32, 32, False),        # 8x32 tiles
    (1, 64, 96, 24, 40, False),          # ragged tiles, ragged N
    (8, 1280, 1280, 16, 16, False),      # 16x16 x 128 tiles, split along the channel slices (each split runs both passes)
    (2, 1280, 1280, 8, 8, False),        # the 8x8 level
    (2, 128, 128, 16, 16, True),         # nearest-2x upsample folded in
    (1, 8, 128, 16, 16, False),          # Cin = 8: the per-lane gather path of igemm_kernel (conv_in of the decoders)
    (1, 128, 3, 16, 16, False),          # N = 3: conv_out
    (2, 320, 4, 32, 32, False)])         # N = 4 over K = 2880: the UNet's last conv, K split over grid.z (skinny-output split)
def test_igemm_w2_conv3x3(hip, n, cin, cout, h, w, up2):
    """3x3 convolutions with the residual pass through the engine's own route (tiled weights where the patch kernel applies)"""
    from mgld_vsr_amd.engine import Act, Engine, pack_conv3x3, split_residual
    hip.set_workspace(hip._test_ws)
    eng = Engine()
    x = h16(rnd(n, cin, h, w, seed=311))
    w32 = rnd(cout, cin, 3, 3, seed=312, scale=(9 * cin) ** -0.5)
    b = rnd(cout, seed=313)
    xin = F.interpolate(x.double(), scale_factor=2, mode="nearest") if up2 else x.double()
    ref = F.conv2d(xin, w32.double(), b.double(), padding=1)
    wp = pack_conv3x3(w32, cin, tap_inner=False if up2 else None)
    hi, lo = split_residual(wp)
    xa = Act(_to_tok(x).to(DEV), n, h, w)
    ho, wo = (2 * h, 2 * w) if up2 else (h, w)
    errs = []
    for w2 in (None, lo.to(DEV)):
        out = eng.conv3x3(xa, hi.to(DEV), b.to(DEV), cout, up2=up2, out_dtype=torch.float32, w2=w2)
        torch.cuda.synchronize()
        errs.append(rel_l2(_from_tok(out.v.cpu(), n, ho, wo), ref))
    _w2_check(*errs)


def test_igemm_w2_tconv(hip):
    from mgld_vsr_amd.engine import split_residual
    clips, T, c, h, w = 2, 4, 128, 6, 6
    x = h16(rnd(clips * T, c, h, w, seed=321))
    w32 = rnd(c, c, 3, 1, 1, seed=322, scale=(3 * c) ** -0.5)
    x5 = x.double().reshape(clips, T, c, h, w).permute(0, 2, 1, 3, 4)
    ref = F.conv3d(x5, w32.double(), None, padding=(1, 0, 0)).permute(0, 2, 1, 3, 4).reshape(clips * T, c, h, w)
    xt = _to_tok(x).to(DEV)
    hi, lo = split_residual(w32[:, :, :, 0, 0].permute(0, 2, 1).reshape(c, 3 * c).contiguous())
    errs = []
    for w2 in (None, lo.to(DEV)):
        out = torch.empty(xt.shape, dtype=torch.float32, device=DEV)
        hip.igemm(xt, hi.to(DEV), out, mode=hip.MODE_TCONV3, tconv=(c, T, h * w), w2=w2)
        torch.cuda.synchronize()
        errs.append(rel_l2(_from_tok(out.cpu(), clips * T, h, w), ref))
    _w2_check(*errs)


# ------------------------------------------------------------------------------------------------------------------
# norms
# ------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("frames,C,h,w,eps", [(2, 320, 16, 16, 1e-5), (1, 1920, 8, 8, 1e-5), (3, 128, 24, 24, 1e-6),
                                              (1, 2560, 8, 8, 1e-5), (2, 64, 4, 4, 1e-6), (2, 320, 64, 64, 1e-5)])
def test_groupnorm(hip, frames, C, h, w, eps):
    x = h16(rnd(frames, C, h, w, seed=30) * 1.5 + 0.3)
    gamma, beta = 1 + 0.1 * rnd(C, seed=31), 0.1 * rnd(C, seed=32)
    ld = C + 24
    xt = torch.zeros(frames * h * w, ld, dtype=torch.half, device=DEV)
    xt[:, 8:8 + C] = _to_tok(x).to(DEV)
    xv = xt[:, 8:8 + C]
    rows = h * w
    gsums = torch.empty(frames, hip.gn_chunks(rows), 32, 2, dtype=torch.float64, device=DEV)
    hip.gn_stats(xv, frames, rows, 32, gsums)
    y = torch.empty(frames * rows, C, dtype=torch.half, device=DEV)
    hip.gn_apply(xv, gsums, eps, gamma.to(DEV), beta.to(DEV), y, frames, rows, 32, True)
    ref = F.silu(F.group_norm(x.float(), 32, gamma, beta, eps))
    xg = x.double().reshape(frames, 32, -1)
    tot = gsums.sum(1).cpu()                       # [frames, groups, (sum, sumsq)]: the chunk sums add up to the moments
    n = xg.shape[-1]
    assert torch.allclose(tot[:, :, 0] / n, xg.mean(-1), atol=2e-5)
    assert rel_l2(tot[:, :, 1] / n, (xg * xg).mean(-1)) < 1e-5
    assert rel_l2(_from_tok(y.cpu().float(), frames, h, w), ref) < 1e-3


def test_spade_apply(hip):
    frames, C, h, w = 2, 320, 8, 8
    rows = h * w
    hh, skip = h16(rnd(frames * rows, C, seed=33)), h16(rnd(frames * rows, C, seed=34))
    gb = h16(rnd(frames * rows, 2 * C, seed=35, scale=0.5))
    gamma, beta = 1 + 0.1 * rnd(C, seed=36), 0.1 * rnd(C, seed=37)
    gsums = torch.empty(frames, hip.gn_chunks(rows), 32, 2, dtype=torch.float64, device=DEV)
    hd = hh.to(DEV)
    hip.gn_stats(hd, frames, rows, 32, gsums)
    y = torch.empty(frames * rows, C, dtype=torch.half, device=DEV)
    hip.spade_apply(hd, gsums, 1e-5, gamma.to(DEV), beta.to(DEV), gb.to(DEV), skip.to(DEV), y, frames, rows, 32)
    hn = F.group_norm(_from_tok(hh.float(), frames, h, w), 32, gamma, beta, 1e-5)
    ref = _from_tok(skip.float(), frames, h, w) + hn * (1 + _from_tok(gb.float()[:, :C], frames, h, w)) + _from_tok(
        gb.float()[:, C:], frames, h, w)
    assert rel_l2(_from_tok(y.cpu().float(), frames, h, w), ref) < 1e-3


@pytest.mark.parametrize("frames,C,h,w", [(2, 320, 64, 64), (3, 640, 32, 32), (1, 2560, 20, 24), (2, 128, 48, 40), (2, 960, 17, 23),
                                          # ragged channel split: the LAST window is narrower and runs MORE row slots than
                                          # the full ones (C = 1280 -> windows of 440 + a 400 tail), the LDS must be sized for it
                                          (5, 1280, 24, 24), (3, 1280, 32, 32), (7, 1280, 20, 20)])
def test_groupnorm_stats_of_output(hip, frames, C, h, w):
    """mgld_spade_apply2 / mgld_gn_apply2 with stats_out: the output bits do not change, the per-group chunk sums add up to the moments of
    the stored output, and a GroupNorm fed by them agrees with one fed by mgld_gn_stats on that output"""
    rows = h * w
    hh, skip = h16(rnd(frames * rows, C, seed=310) + 0.2).to(DEV), h16(rnd(frames * rows, C, seed=311)).to(DEV)
    gb = h16(rnd(frames * rows, 2 * C, seed=312, scale=0.5)).to(DEV)
    gamma, beta = (1 + 0.1 * rnd(C, seed=313)).to(DEV), (0.1 * rnd(C, seed=314)).to(DEV)
    gs = torch.empty(frames, hip.gn_chunks(rows), 32, 2, dtype=torch.float64, device=DEV)
    hip.gn_stats(hh, frames, rows, 32, gs)
    oc = hip.gn_apply_chunks(frames, rows, C, 32)
    assert oc > 0
    for spade in (True, False):
        y0, y1 = torch.empty_like(hh), torch.empty_like(hh)
        so = torch.full((frames, oc, 32, 2), float("nan"), dtype=torch.float64, device=DEV)
        if spade:
            hip.spade_apply(hh, gs, 1e-5, gamma, beta, gb, skip, y0, frames, rows, 32)
            hip.spade_apply(hh, gs, 1e-5, gamma, beta, gb, skip, y1, frames, rows, 32, stats_out=so)
        else:
            hip.gn_apply(hh, gs, 1e-5, gamma, beta, y0, frames, rows, 32, True)
            hip.gn_apply(hh, gs, 1e-5, gamma, beta, y1, frames, rows, 32, True, stats_out=so)
        torch.cuda.synchronize()
        assert torch.equal(y0, y1)
        yg = y1.double().reshape(frames, rows, 32, C // 32)
        tot = so.sum(1)
        assert torch.isfinite(so).all()
        # (sums are taken before the fp16 rounding: the first moment is compared on the scale of the rms, not of the — possibly ~0 — mean)
        nel = rows * (C // 32)
        rms = ((yg * yg).sum((1, 3)) / nel).sqrt()
        assert (((tot[:, :, 0] - yg.sum((1, 3))) / nel).abs() / rms).max() < 2e-5 and rel_l2(tot[:, :, 1], (yg * yg).sum((1, 3))) < 1e-4
        gs2 = torch.empty_like(gs)
        hip.gn_stats(y1, frames, rows, 32, gs2)
        z0, z1 = torch.empty_like(hh), torch.empty_like(hh)
        hip.gn_apply(y1, gs2, 1e-5, gamma, beta, z0, frames, rows, 32, False)
        hip.gn_apply(y1, so, 1e-5, gamma, beta, z1, frames, rows, 32, False, kind=hip.GN_GROUP_SUMS, chunks=oc)
        torch.cuda.synchronize()
        assert rel_l2(z1.float(), z0.float()) < 1e-4


@pytest.mark.parametrize("frames,C,h,w,silu", [(2, 1280, 16, 16, 1), (3, 1920, 8, 8, 1), (2, 2560, 16, 16, 1), (1, 320, 16, 16, 0),
                                               (2, 960, 8, 8, 1), (2, 640, 16, 12, 2), (5, 64, 4, 4, 1), (1, 512, 16, 16, 0)])
def test_groupnorm_single_launch(hip, frames, C, h, w, silu):
    """mgld_gn_fused (frames of <= 256 rows: statistics + apply in one launch) against torch and against the two-launch path it
    replaces (same per-channel fp32 / per-group fp64 arithmetic: agreement at the fp16 rounding level, strided input view)"""
    rows = h * w
    assert hip.gn_fused_applies(rows, C, 32)
    assert not hip.gn_fused_applies(1024, C, 32)
    x = h16(rnd(frames, C, h, w, seed=40) * 1.5 + 0.3)
    gamma, beta = 1 + 0.1 * rnd(C, seed=41), 0.1 * rnd(C, seed=42)
    xt = torch.zeros(frames * rows, C + 24, dtype=torch.half, device=DEV)
    xt[:, 8:8 + C] = _to_tok(x).to(DEV)
    xv = xt[:, 8:8 + C]
    y = torch.full((frames * rows, C), float("nan"), dtype=torch.half, device=DEV)
    hip.gn_fused(xv, 1e-5, gamma.to(DEV), beta.to(DEV), y, frames, rows, 32, silu)
    gsums = torch.empty(frames, hip.gn_chunks(rows), 32, 2, dtype=torch.float64, device=DEV)
    hip.gn_stats(xv, frames, rows, 32, gsums)
    y2 = torch.empty_like(y)
    hip.gn_apply(xv, gsums, 1e-5, gamma.to(DEV), beta.to(DEV), y2, frames, rows, 32, silu)
    ref = F.group_norm(x.float(), 32, gamma, beta, 1e-5)
    ref = F.silu(ref) if silu == 1 else F.relu(ref) if silu == 2 else ref
    assert rel_l2(_from_tok(y.cpu().float(), frames, h, w), ref) < 1e-3
    assert rel_l2(y.float(), y2.float()) < 3e-4


def test_spade_single_launch(hip):
    """SPADE form of mgld_gn_fused, with the per-step modulation table indexed on the device (step_idx)"""
    frames, C, h, w, S = 2, 1280, 16, 16, 3
    rows = h * w
    hh, skip = h16(rnd(frames * rows, C, seed=43)), h16(rnd(frames * rows, C, seed=44))
    table = h16(rnd(S, frames * rows, 2 * C, seed=45, scale=0.5)).to(DEV)
    gamma, beta = 1 + 0.1 * rnd(C, seed=46), 0.1 * rnd(C, seed=47)
    step = torch.tensor([2], dtype=torch.int32, device=DEV)
    y = torch.full((frames * rows, C), float("nan"), dtype=torch.half, device=DEV)
    hip.gn_fused(hh.to(DEV), 1e-5, gamma.to(DEV), beta.to(DEV), y, frames, rows, 32, 0, gb=table[0], skip=skip.to(DEV), step_idx=step,
                 step_stride=table.stride(0))
    gb = table[2].cpu().float()
    hn = F.group_norm(_from_tok(hh.float(), frames, h, w), 32, gamma, beta, 1e-5)
    ref = _from_tok(skip.float(), frames, h, w) + hn * (1 + _from_tok(gb[:, :C], frames, h, w)) + _from_tok(gb[:, C:], frames, h, w)
    assert rel_l2(_from_tok(y.cpu().float(), frames, h, w), ref) < 1e-3


@pytest.mark.parametrize("rows,C", [(130, 320), (64, 640), (37, 1280), (9, 64)])
def test_layernorm(hip, rows, C):
    x = h16(rnd(rows, C, seed=38) * 2 + 0.5)
    g, b = 1 + 0.1 * rnd(C, seed=39), 0.1 * rnd(C, seed=40)
    y = torch.empty(rows, C, dtype=torch.half, device=DEV)
    hip.layernorm(x.to(DEV), g.to(DEV), b.to(DEV), y)
    assert rel_l2(y.cpu().float(), F.layer_norm(x.float(), (C,), g, b, 1e-5)) < 1e-3


# ------------------------------------------------------------------------------------------------------------------
# attention
# ------------------------------------------------------------------------------------------------------------------
def _attn_ref(q, k, v, scale):
    s = torch.einsum("bhid,bhjd->bhij", q, k) * scale
    return torch.einsum("bhij,bhjd->bhid", s.softmax(-1), v)


@pytest.mark.parametrize("M,C,N,ptune,ctune,act", [(4096, 320, 960, 27, 27, 0), (2048, 320, 320, 27, 20, 3), (1024, 640, 1920, 22, 20, 0), (512, 1280, 1280, 23, 20, 0),
                                                   (4096, 320, 2560, 27, 21, 4), (1024, 640, 5120, 22, 24, 4), (512, 1280, 10240, 26, 20, 4), (256, 128, 256, 26, 26, 4)])
def test_layernorm_folded_into_the_projection(hip, M, C, N, ptune, ctune, act):
    """MgldIGemm.row_part / ln_part: a projection writes the row sums of its output t (producer, every ping-pong tile width: one to eight
    column tiles per row), the next projection runs on the RAW rows of t against W diag(gamma) and applies the LayerNorm in its epilogue
    (plain, SiLU and GEGLU forms) — against fp64 LayerNorm(t) W^T + b of the stored fp16 t"""
    from mgld_vsr_amd.engine import pack_geglu
    a, wp_, bp_ = h16(rnd(M, C, seed=400)), h16(rnd(C, C, seed=401, scale=C ** -0.5)), rnd(C, seed=402)
    r = h16(rnd(M, C, seed=403) * 2.0 + 0.5)
    t = torch.full((M, C), float("nan"), dtype=torch.half, device=DEV)
    got = []

    def part(chunks):
        got.append(torch.full((chunks, M, 2), float("nan"), dtype=torch.float32, device=DEV))
        return got[0]
    hip.igemm(a.to(DEV), wp_.to(DEV), t, bias=bp_.to(DEV), resid=r.to(DEV), tune=ptune, row_part=part)
    torch.cuda.synchronize()
    assert got, "the ping-pong kernel writes the row sums of its output"
    t64 = a.double() @ wp_.double().t() + bp_.double() + r.double()
    sums = got[0].cpu().double().sum(0)
    assert rel_l2(sums[:, 0], t64.sum(1)) < 2e-5 and rel_l2(sums[:, 1], (t64 * t64).sum(1)) < 1e-5
    # consumer
    gam, bet = 1 + 0.2 * rnd(C, seed=404), 0.2 * rnd(C, seed=405)
    W, b = rnd(N, C, seed=406, scale=C ** -0.5), rnd(N, seed=407)
    if act == 4:
        W, b = pack_geglu(W, b)
    Wp = (W * gam[None, :]).half()
    sv = Wp.float().sum(1)
    b2 = W @ bet + b
    ts = t.cpu()
    ln = F.layer_norm(ts.double(), (C,), gam.double(), bet.double(), 1e-5)
    pre = ln @ W.double().t() + b.double()
    if act == 4:
        pv = pre.reshape(M, N // 64, 2, 32)
        ref = (pv[:, :, 0] * F.gelu(pv[:, :, 1])).reshape(M, N // 2)
    else:
        ref = F.silu(pre) if act == 3 else pre
    out = torch.full((M, N // 2 if act == 4 else N), float("nan"), dtype=torch.half, device=DEV)
    hip.igemm(t, Wp.to(DEV), out, bias=b2.to(DEV), act=act, tune=ctune, N=N, ln=(got[0], got[0].shape[0], sv.to(DEV), 1e-5))
    torch.cuda.synchronize()
    # against the exact weights: the rounding of fp16(W gamma) is part of the figure (2.8e-4 per weight) beside the rounding of the stored output
    assert rel_l2(out.cpu(), ref) < 6e-4
    # against the same algebra on the ROUNDED weights in fp64: only the output's own rounding is left — no normalised fp16 copy of t in between
    mu, var = ts.double().mean(1, keepdim=True), ts.double().var(1, unbiased=False, keepdim=True)
    pre_r = ((ts.double() - mu) / (var + 1e-5).sqrt()) @ Wp.double().t() + b2.double()
    if act == 4:
        pv = pre_r.reshape(M, N // 64, 2, 32)
        ref_r = (pv[:, :, 0] * F.gelu(pv[:, :, 1])).reshape(M, N // 2)
    else:
        ref_r = F.silu(pre_r) if act == 3 else pre_r
    assert rel_l2(out.cpu(), ref_r) < 3.2e-4


def test_layernorm_fold_refused_off_the_pingpong_path(hip):
    a, w = h16(rnd(100, 72, seed=410)).to(DEV), h16(rnd(72, 72, seed=411)).to(DEV)
    out = torch.empty(100, 72, dtype=torch.half, device=DEV)
    assert hip.igemm(a, w, out, query_row_chunks=True) == 0
    called = []
    hip.igemm(a, w, out, row_part=lambda c: called.append(c))            # a producer the family does not take: no statistics, no error
    assert not called
    with pytest.raises(RuntimeError):
        hip.igemm(a, w, out, ln=(torch.zeros(1, 100, 2, device=DEV), 1, torch.zeros(72, device=DEV), 1e-5))


@pytest.mark.parametrize("B,H,Nq,Nkv,D", [(2, 5, 256, 256, 64), (1, 2, 200, 77, 64), (2, 4, 130, 130, 128), (1, 1, 64, 64, 64),
                                          (1, 3, 1024, 1024, 64), (3, 2, 64, 5, 64),
                                          (8, 5, 300, 300, 64), (4, 6, 256, 77, 64)])   # batch * heads % 8 == 0: XCD-grouped block order
def test_flash_attention(hip, B, H, Nq, Nkv, D):
    q, k, v = h16(rnd(B, H, Nq, D, seed=41)), h16(rnd(B, H, Nkv, D, seed=42)), h16(rnd(B, H, Nkv, D, seed=43))
    scale = D ** -0.5
    ref = _attn_ref(q.float(), k.float(), v.float(), scale)
    C_ = H * D
    qt = q.permute(0, 2, 1, 3).reshape(B * Nq, C_).contiguous().to(DEV)      # token-major [B*Nq, H*D]
    kt = k.permute(0, 2, 1, 3).reshape(B * Nkv, C_).contiguous().to(DEV)
    nkp = (Nkv + 7) // 8 * 8
    vt = torch.zeros(B, H, D, nkp, dtype=torch.half)
    vt[..., :Nkv] = v.permute(0, 1, 3, 2)
    vt = vt.to(DEV)
    o = torch.empty(B * Nq, C_, dtype=torch.half, device=DEV)
    hip.attention(qt, kt, vt, o, batch=B, heads=H, Nq=Nq, Nkv=Nkv, head_dim=D,
                  q_strides=(Nq * C_, C_, D), k_strides=(Nkv * C_, C_, D), vt_strides=(H * D * nkp, D * nkp, nkp),
                  o_strides=(Nq * C_, C_, D), scale=scale)
    got = o.cpu().float().reshape(B, Nq, H, D).permute(0, 2, 1, 3)
    assert rel_l2(got, ref) < 2e-3


@pytest.mark.parametrize("B,H,Nq,Nkv,D", [(2, 5, 256, 256, 64), (1, 2, 200, 77, 64), (2, 4, 130, 130, 128), (1, 1, 64, 64, 64),
                                          (1, 3, 1024, 1024, 64), (3, 2, 64, 5, 64), (1, 4, 300, 300, 128),
                                          (8, 5, 300, 300, 64), (4, 6, 256, 77, 64)])
def test_flash_attention_rowmajor_v(hip, B, H, Nq, Nkv, D):
    """v_rowmajor: q, k, v are the three column blocks of ONE fused projection [tokens, 3*H*D]; V is transposed by the LDS read
    (ds_read_b64_tr_b16).  Must give the same bits as the V^T form on the same operands (same products, same summation order), ragged
    last tiles included."""
    q, k, v = h16(rnd(B, H, Nq, D, seed=41)), h16(rnd(B, H, Nkv, D, seed=42)), h16(rnd(B, H, Nkv, D, seed=43))
    scale = D ** -0.5
    ref = _attn_ref(q.float(), k.float(), v.float(), scale)
    C_ = H * D
    N = max(Nq, Nkv)
    qkv = torch.zeros(B * N, 3 * C_, dtype=torch.half)          # one token-major buffer, q | k | v side by side (rows past Nq / Nkv unused)
    qkv.view(B, N, 3 * C_)[:, :Nq, :C_] = q.permute(0, 2, 1, 3).reshape(B, Nq, C_)
    qkv.view(B, N, 3 * C_)[:, :Nkv, C_:2 * C_] = k.permute(0, 2, 1, 3).reshape(B, Nkv, C_)
    qkv.view(B, N, 3 * C_)[:, :Nkv, 2 * C_:] = v.permute(0, 2, 1, 3).reshape(B, Nkv, C_)
    qkv = qkv.to(DEV)
    o = torch.empty(B * Nq, C_, dtype=torch.half, device=DEV)
    st = (N * 3 * C_, 3 * C_, D)
    hip.attention(qkv, qkv[:, C_:], qkv[:, 2 * C_:], o, batch=B, heads=H, Nq=Nq, Nkv=Nkv, head_dim=D, q_strides=st, k_strides=st,
                  vt_strides=st, o_strides=(Nq * C_, C_, D), scale=scale, v_rowmajor=True)
    got = o.cpu().float().reshape(B, Nq, H, D).permute(0, 2, 1, 3)
    assert rel_l2(got, ref) < 2e-3
    # against the V^T form
    nkp = (Nkv + 7) // 8 * 8
    vt = torch.zeros(B, H, D, nkp, dtype=torch.half)
    vt[..., :Nkv] = v.permute(0, 1, 3, 2)
    o2 = torch.empty_like(o)
    hip.attention(qkv, qkv[:, C_:], vt.to(DEV), o2, batch=B, heads=H, Nq=Nq, Nkv=Nkv, head_dim=D, q_strides=st, k_strides=st,
                  vt_strides=(H * D * nkp, D * nkp, nkp), o_strides=(Nq * C_, C_, D), scale=scale)
    assert torch.equal(o, o2)


@pytest.mark.parametrize("B,H,Nq,Nkv,D,rowmajor", [(8, 5, 4096, 4096, 64, True), (8, 5, 4096, 77, 64, False), (8, 10, 1024, 1024, 64, True)])
def test_flash_attention_production_shapes(hip, B, H, Nq, Nkv, D, rowmajor):
    """the attention launches of the 8 x 512^2 segment as they are launched there: 64^2 self-attention (40 (frame, head) pairs x 4096
    tokens, q | k | v of one fused projection, row-major V: the LDS-DMA kernel), its cross-attention against the 77 context tokens
    (V^T form), the 32^2 self-attention — against fp32 torch on the device, frame by frame"""
    C_ = H * D
    scale = D ** -0.5
    g = torch.Generator().manual_seed(77)
    if rowmajor:
        qkv = (torch.randn(B * Nq, 3 * C_, generator=g) * 0.8).half()
        # spikes: keys that match one query far better than the rest, late in the key axis — the online-softmax rescale path of a row in
        # the FIRST and of a row in the SECOND 32-row half of a wave's 64 query rows (flash_attn_kernel<64, true, true, 2>), frame 0 head 0
        for qrow, krow in ((7, Nkv - 100), (40, Nkv - 37)):
            qkv[krow, C_:C_ + D] = qkv[qrow, :D] * 4
        qkv = qkv.to(DEV)
        q2, k2, v2 = qkv[:, :C_], qkv[:, C_:2 * C_], qkv[:, 2 * C_:]
        st = (Nq * 3 * C_, 3 * C_, D)
        o = torch.empty(B * Nq, C_, dtype=torch.half, device=DEV)
        hip.attention(qkv, k2, v2, o, batch=B, heads=H, Nq=Nq, Nkv=Nkv, head_dim=D, q_strides=st, k_strides=st, vt_strides=st,
                      o_strides=(Nq * C_, C_, D), scale=scale, v_rowmajor=True)
        kq = lambda t, n: t.float().reshape(B, n, H, D).permute(0, 2, 1, 3)
        qf, kf, vf = kq(q2, Nq), kq(k2, Nkv), kq(v2, Nkv)
    else:
        qt = (torch.randn(B * Nq, C_, generator=g) * 0.8).half().to(DEV)
        kt = (torch.randn(B * Nkv, C_, generator=g) * 0.8).half().to(DEV)
        v = (torch.randn(B, H, Nkv, D, generator=g) * 0.8).half()
        nkp = (Nkv + 7) // 8 * 8
        vt = torch.zeros(B, H, D, nkp, dtype=torch.half)
        vt[..., :Nkv] = v.permute(0, 1, 3, 2)
        o = torch.empty(B * Nq, C_, dtype=torch.half, device=DEV)
        hip.attention(qt, kt, vt.to(DEV), o, batch=B, heads=H, Nq=Nq, Nkv=Nkv, head_dim=D, q_strides=(Nq * C_, C_, D),
                      k_strides=(Nkv * C_, C_, D), vt_strides=(H * D * nkp, D * nkp, nkp), o_strides=(Nq * C_, C_, D), scale=scale)
        qf = qt.float().reshape(B, Nq, H, D).permute(0, 2, 1, 3)
        kf = kt.float().reshape(B, Nkv, H, D).permute(0, 2, 1, 3)
        vf = v.float().to(DEV)
    got = o.float().reshape(B, Nq, H, D).permute(0, 2, 1, 3)
    num = den = 0.0
    for b in range(B):                      # fp32 reference on the device, one frame at a time (the 64^2 logits are 335 MB per frame)
        ref = torch.softmax(qf[b] @ kf[b].transpose(-1, -2) * scale, dim=-1) @ vf[b]
        num += float(((got[b] - ref).double() ** 2).sum())
        den += float((ref.double() ** 2).sum())
    assert torch.isfinite(o).all() and (num / den) ** 0.5 < 1e-3


def test_flash_attention_spike(hip):
    # force a large running-max jump in a late key tile (online-softmax rescale path)
    B, H, N, D = 1, 1, 192, 64
    q, k, v = h16(rnd(B, H, N, D, seed=44)), h16(rnd(B, H, N, D, seed=45)), h16(rnd(B, H, N, D, seed=46))
    k[0, 0, 150] = q[0, 0, 7] * 4
    scale = D ** -0.5
    ref = _attn_ref(q.float(), k.float(), v.float(), scale)
    vt = v.permute(0, 1, 3, 2).contiguous().to(DEV)
    qt, kt = q.reshape(N, D).to(DEV), k.reshape(N, D).to(DEV)
    o = torch.empty(N, D, dtype=torch.half, device=DEV)
    hip.attention(qt, kt, vt, o, batch=1, heads=1, Nq=N, Nkv=N, head_dim=D, q_strides=(N * D, D, D), k_strides=(N * D, D, D),
                  vt_strides=(D * N, D * N, N), o_strides=(N * D, D, D), scale=scale)
    assert rel_l2(o.cpu().float().reshape(1, 1, N, D), ref) < 2e-3


@pytest.mark.parametrize("T,HW,heads,D", [(5, 64, 20, 64), (8, 16, 2, 64), (3, 9, 1, 128)])
def test_temporal_attention(hip, T, HW, heads, D):
    C_ = heads * D
    q, k, v = h16(rnd(T * HW, C_, seed=47)), h16(rnd(T * HW, C_, seed=48)), h16(rnd(T * HW, C_, seed=49))
    o = torch.empty(T * HW, C_, dtype=torch.half, device=DEV)
    hip.temporal_attention(q.to(DEV), k.to(DEV), v.to(DEV), o, T, HW, heads, D, D ** -0.5)

    def sh(t):  # [T*HW, C] -> [HW, heads, T, D]
        return t.float().reshape(T, HW, heads, D).permute(1, 2, 0, 3)
    ref = _attn_ref(sh(q), sh(k), sh(v), D ** -0.5).permute(2, 0, 1, 3).reshape(T * HW, C_)
    assert rel_l2(o.cpu().float(), ref) < 1e-3


def test_softmax_rows(hip):
    rows, cols = 300, 1000
    s = rnd(rows, cols, seed=50) * 3
    p = torch.empty(rows, cols + 8, dtype=torch.half, device=DEV)
    hip.softmax_rows(s.to(DEV), p, rows, cols)
    assert rel_l2(p[:, :cols].cpu().float(), s.softmax(-1)) < 1e-3


# ------------------------------------------------------------------------------------------------------------------
# small dense + embedding + layout
# ------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("M,N,K", [(5, 1280, 320), (8, 20000, 1280), (1, 64, 8), (16, 100, 256)])
def test_linear_small(hip, M, N, K):
    a, w, b = rnd(M, K, seed=51), h16(rnd(N, K, seed=52, scale=K ** -0.5)), rnd(N, seed=53)
    y = torch.empty(M, N, dtype=torch.float32, device=DEV)
    hip.linear_small(a.to(DEV), w.to(DEV), b.to(DEV), y, silu_in=True, silu_out=True)
    ref = F.silu(F.silu(a) @ w.float().t() + b)
    assert rel_l2(y.cpu(), ref) < 1e-5


def test_timestep_embedding(hip):
    t = torch.tensor([0., 20., 41., 999., 500.])
    dim = 320
    out = torch.empty(5, dim, dtype=torch.float32, device=DEV)
    hip.timestep_embedding(t.to(DEV), out)
    half = dim // 2
    freqs = torch.exp(-math.log(10000) * torch.arange(half, dtype=torch.float32) / half)
    args = t[:, None] * freqs[None]
    ref = torch.cat([torch.cos(args), torch.sin(args)], -1)
    assert float((out.cpu() - ref).abs().max()) < 2e-4   # fp32 sin/cos of arguments up to 999


def test_layout_roundtrip(hip):
    x = rnd(3, 4, 10, 12, seed=54)
    y = torch.empty(3 * 10 * 12, 16, dtype=torch.half, device=DEV)
    hip.nchw_to_nhwc(x.to(DEV), y, 8)
    assert float(y[:, 4:8].abs().max()) == 0
    back = torch.empty(3, 4, 10, 12, dtype=torch.float32, device=DEV)
    hip.nhwc_to_nchw(y, back)
    assert torch.equal(back.cpu(), x.half().float())
    z = torch.zeros(3 * 10 * 12, 24, dtype=torch.half, device=DEV)
    hip.copy2d(y[:, :8], z[:, 16:24])
    assert torch.equal(z[:, 16:24], y[:, :8])
    hip.axpby(y[:, :8], z[:, 16:24], 2.0, 0.5)
    assert torch.allclose(z[:, 16:24].float(), y[:, :8].float() * 2.5, atol=2e-3, rtol=2e-3)


# ------------------------------------------------------------------------------------------------------------------
# high-precision first-stage encoder pieces (csrc/hpenc.hip, MgldIGemm.r_f32)
# ------------------------------------------------------------------------------------------------------------------
def _unsplit(a3, C):
    """[ah | 16 al | ah/256] -> ah + al"""
    return a3[:, :C].double() + a3[:, C:2 * C].double() / 16.0


@pytest.mark.parametrize("frames,rows,C,groups", [(2, 4096, 128, 32), (3, 1000, 512, 32), (1, 300, 32, 32), (2, 64, 64, 32)])
def test_hp_groupnorm_split(hip, frames, rows, C, groups):
    """mgld_hp_gn_stats + mgld_hp_gn_split vs F.group_norm (+ SiLU) in fp64: fp32 output to 1e-6, the split operand reassembles to 2^-21,
    its third block is the first / 256; identity mode (no statistics) splits the input itself"""
    x = rnd(frames * rows, C, seed=500) * 3 + 0.7
    x[:, 5] += 40.0                                                   # a channel with a large mean
    gamma, beta = 1 + 0.1 * rnd(C, seed=501), 0.1 * rnd(C, seed=502)
    xd = x.to(DEV)
    gs = torch.empty(frames * hip.hp_chunks(rows) * groups * 2, dtype=torch.float64, device=DEV)
    hip.hp_gn_stats(xd, frames, rows, groups, gs)
    xr = x.double().view(frames, rows, C).permute(0, 2, 1)
    for silu in (False, True):
        ref = F.group_norm(xr, groups, gamma.double(), beta.double(), 1e-6)
        if silu:
            ref = ref * torch.sigmoid(ref)
        ref = ref.permute(0, 2, 1).reshape(frames * rows, C)
        o32 = torch.empty(frames * rows, C, device=DEV)
        hip.hp_gn_split(xd, gs, 1e-6, gamma.to(DEV), beta.to(DEV), silu, o32, frames, rows, groups)
        assert rel_l2(o32.cpu().double(), ref) < 2e-6
        o3 = torch.empty(frames * rows, 3 * C, dtype=torch.half, device=DEV)
        hip.hp_gn_split(xd, gs, 1e-6, gamma.to(DEV), beta.to(DEV), silu, o3, frames, rows, groups)
        assert rel_l2(_unsplit(o3.cpu(), C), ref) < 3e-6
        assert torch.equal(o3[:, 2 * C:].cpu().float(), (o3[:, :C].cpu().float() / 256.0).half().float())
    o3 = torch.empty(frames * rows, 3 * C, dtype=torch.half, device=DEV)
    hip.hp_gn_split(xd, None, 0.0, None, None, False, o3, frames, rows, 1)
    assert rel_l2(_unsplit(o3.cpu(), C), x.double()) < 1e-6
    # deterministic: a second pass gives the same bits
    gs2 = torch.empty_like(gs)
    hip.hp_gn_stats(xd, frames, rows, groups, gs2)
    assert torch.equal(gs, gs2)


def test_hp_softmax_rows(hip):
    S = (rnd(300, 1000, seed=510) * 4).to(DEV)
    ref = torch.softmax(S.cpu().double(), -1)
    hip.hp_softmax_rows(S)
    assert rel_l2(S.cpu().double(), ref) < 1e-6 and float((S.sum(-1) - 1).abs().max()) < 1e-5


@pytest.mark.parametrize("n,h,w,cin,cout,stride", [(2, 32, 32, 128, 128, 1), (1, 64, 48, 128, 256, 1), (2, 16, 16, 512, 512, 1), (2, 32, 32, 128, 128, 2),
                                                    (1, 8, 8, 64, 32, 1), (4, 256, 256, 128, 128, 1)])    # the last: enough tiles for the ping-pong kernel's fp32 epilogue
def test_hp_split_convolution(hip, n, h, w, cin, cout, stride):
    """the split-fp16 contraction end to end: fp32 input -> mgld_hp_gn_split (identity) -> mgld_igemm over 3 Cin channels against
    engine.pack_hp weights, fp32 output + fp32 residual (MgldIGemm.r_f32) vs F.conv2d in fp64: ~1e-6 where the plain fp16 kernels
    give 3e-4.  stride 2 = the encoder's Downsample (F.pad (0,1,0,1) + valid conv, model.py:114-118)."""
    from mgld_vsr_amd.engine import Act, Engine, pack_conv3x3, pack_hp
    eng = Engine()
    x = rnd(n, cin, h, w, seed=520) * 2
    wt = rnd(cout, cin, 3, 3, seed=521) / (cin * 9) ** 0.5
    b = 0.1 * rnd(cout, seed=522)
    if stride == 1:
        ref = F.conv2d(x.double(), wt.double(), b.double(), padding=1)
    else:
        ref = F.conv2d(F.pad(x.double(), (0, 1, 0, 1)), wt.double(), b.double(), stride=2)
    ho, wo = ref.shape[-2:]
    skip = rnd(n * ho * wo, cout, seed=523)
    ref = ref.permute(0, 2, 3, 1).reshape(-1, cout) + skip.double()
    xa = x.permute(0, 2, 3, 1).reshape(-1, cin).contiguous().to(DEV)
    a3 = torch.empty(xa.shape[0], 3 * cin, dtype=torch.half, device=DEV)
    hip.hp_gn_split(xa, None, 0.0, None, None, False, a3, n, h * w, 1)
    wp = pack_conv3x3(pack_hp(wt), 3 * cin).half().to(DEV)
    kw = dict(stride=2, pad=(0, 0), hw_out=(ho, wo)) if stride == 2 else {}
    hip.IGEMM_LOG = []
    try:
        out = eng.conv3x3(Act(a3, n, h, w), wp, b.to(DEV), cout, resid=Act(skip.to(DEV), n, ho, wo), out_dtype=torch.float32, **kw)
        log = hip.IGEMM_LOG
    finally:
        hip.IGEMM_LOG = None
    assert out.v.dtype == torch.float32
    if n * h * w >= 1 << 18:
        assert "conv3r" in hip.igemm_kernel_name(log[-1])[0]
    assert rel_l2(out.v.cpu().double(), ref) < 3e-6


@pytest.mark.parametrize("scale,bound", [(1e-2, 4e-6), (1e-3, 2e-5), (1e-4, 2e-4)])
def test_hp_split_convolution_small_activations(hip, scale, bound):
    """in the split operand [ah | 16 al | ah / 256] the two correction terms leave fp16's normal range for |a| below ~1e-2
    (ah / 256 < 6.1e-5) and ~1e-3 (16 al): what the split contraction still delivers for activations that small (every element: the
    worst case; a post-GroupNorm tensor is O(1) with a minority of small elements).  Measured: 1e-2 -> ~1e-6, 1e-3 -> ~6e-6, 1e-4 -> ~6e-5
    relative — the corrections degrade gracefully through the subnormals (the MFMA does not flush them), never worse than the plain fp16
    contraction's 3e-4."""
    from mgld_vsr_amd.engine import Act, Engine, pack_conv3x3, pack_hp
    eng = Engine()
    n, h, w, cin, cout = 2, 32, 32, 128, 128
    x = rnd(n, cin, h, w, seed=530) * scale
    wt = rnd(cout, cin, 3, 3, seed=531) / (cin * 9) ** 0.5
    ref = F.conv2d(x.double(), wt.double(), None, padding=1).permute(0, 2, 3, 1).reshape(-1, cout)
    xa = x.permute(0, 2, 3, 1).reshape(-1, cin).contiguous().to(DEV)
    a3 = torch.empty(xa.shape[0], 3 * cin, dtype=torch.half, device=DEV)
    hip.hp_gn_split(xa, None, 0.0, None, None, False, a3, n, h * w, 1)
    wp = pack_conv3x3(pack_hp(wt), 3 * cin).half().to(DEV)
    out = eng.conv3x3(Act(a3, n, h, w), wp, torch.zeros(cout, device=DEV), cout, out_dtype=torch.float32)
    err = rel_l2(out.v.cpu().double(), ref)
    print(f"hp split conv, activations ~{scale:g}: rel. L2 {err:.2e}")
    assert err < bound, err


@pytest.mark.parametrize("n,cin,ti", [(320, 320, False), (640, 128, True), (100, 64, True), (160, 96, False)])
def test_tile_conv3p_kernel_matches_the_host_layout(hip, n, cin, ti):
    """mgld_tile_conv3p == engine.tile_conv3p (the torch statement of MgldIGemm.tap_inner = 2's layout, pinned by tests/test_host_cpu.py) bit for bit,
    for both K orders of the packed weights and an N that is not a multiple of 64 (zero rows)"""
    from mgld_vsr_amd.engine import pack_conv3x3, tile_conv3p
    w = rnd(n, cin, 3, 3, seed=600)
    wp = pack_conv3x3(w, cin, tap_inner=ti).half().to(DEV)
    assert torch.equal(hip.tile_conv3p(wp, cin, ti), tile_conv3p(wp, cin, ti))


@pytest.mark.parametrize("B,H,N", [(8, 5, 4096), (8, 10, 1024), (3, 5, 256), (1, 2, 64), (2, 3, 192), (2, 2, 320), (1, 1, 128), (16, 20, 256)])
def test_flash_attention_prescaled_queries(hip, B, H, N):
    """the software-pipelined d = 64 self-attention (flash_attn_sp2_kernel / flash_attn_sp_kernel, round 6): queries pre-scaled by
    d^-1/2 log2(e) (scale = ln 2 selects it), the running max carried into the scores as the C operand of the score chain, probabilities
    computed speculatively with the row-sum partials as the overflow trigger.  Rows with spikes late in the key axis (the rescale path), a
    spike far past exp2's range (inf in the speculative pass), a whole tile moderately above the running max (the row-sum trigger without
    one large probability), rows whose FIRST key tile is far / hundreds below the rest (the anchor at t = 0 and a large later jump) and
    rows whose scores are all very negative."""
    D = 64
    C_ = H * D
    f = D ** -0.5 * 1.4426950408889634
    g = torch.Generator().manual_seed(78)
    q = torch.randn(B * N, C_, generator=g) * 0.8
    k = torch.randn(B * N, C_, generator=g) * 0.8
    v = torch.randn(B * N, C_, generator=g) * 0.8
    if N >= 256:
        k[N - 100, :D] = q[7, :D] * 4                   # late spikes (frame 0, head 0)
        k[N - 37, :D] = q[40, :D] * 4
        k[:64, :D] = -q[9, :D] * 3                      # query 9: its first key tile scores ~ -3 |q|^2, far below the later ones
        q[11, :D] = 0.0                                 # flat row
        k[:, D:2 * D] -= 2.5 * torch.sign(q[13, D:2 * D])     # head 1, query 13: every score strongly negative
        k[N - 70, :D] = q[50, :D] * 60                  # far past exp2's range relative to the anchor
        k[130:190, :D] += q[60, :D] * 0.9               # a whole tile moderately above query 60's running max
        if H > 2:
            k[:64, 2 * D:3 * D] = -q[70, 2 * D:3 * D] * 40   # head 2, query 70: anchor tile hundreds below the rest
    qkv = torch.cat([(q * f), k, v], 1).half().to(DEV)
    o = torch.empty(B * N, C_, dtype=torch.half, device=DEV)
    st = (N * 3 * C_, 3 * C_, D)
    hip.TIMED = []
    try:
        hip.attention(qkv, qkv[:, C_:], qkv[:, 2 * C_:], o, batch=B, heads=H, Nq=N, Nkv=N, head_dim=D, q_strides=st, k_strides=st, vt_strides=st,
                      o_strides=(N * C_, C_, D), scale=1.0 / 1.4426950408889634, v_rowmajor=True)
        want = "flash_attn_sp2_kernel" if (N % 128 == 0 and N >= 256) else "flash_attn_sp_kernel"
        assert hip.TIMED[0][1]["kernel"] == want, hip.TIMED[0][1]["kernel"]
    finally:
        hip.TIMED = None
    kq = lambda t: t.float().reshape(B, N, H, D).permute(0, 2, 1, 3)
    qf, kf, vf = kq(qkv[:, :C_]), kq(qkv[:, C_:2 * C_]), kq(qkv[:, 2 * C_:])
    got = o.float().reshape(B, N, H, D).permute(0, 2, 1, 3)
    num = den = 0.0
    worst = 0.0
    for b in range(B):
        ref = torch.softmax(qf[b] @ kf[b].transpose(-1, -2) * 0.6931471805599453, dim=-1) @ vf[b]      # 2^(q' k) = e^(ln 2 q' k)
        num += float(((got[b] - ref).double() ** 2).sum())
        den += float((ref.double() ** 2).sum())
        worst = max(worst, float((got[b] - ref).abs().max()))
    assert torch.isfinite(o).all() and (num / den) ** 0.5 < 1e-3 and worst < 2e-2, ((num / den) ** 0.5, worst)
