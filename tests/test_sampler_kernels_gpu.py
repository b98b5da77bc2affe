"""GPU parity: DDPM step, flow warp, motion guidance (closed-form adjoint vs autograd oracle), fb-consistency,
flow resize, AdaIN / wavelet colour fix, aggregation-sampling tile ops — libmgld_hip (C ABI) vs oracle/ (torch CPU)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
if HERE not in sys.path:
    sys.path.insert(0, HERE)

from oracle import colorfix as ocf
from oracle import flow as oflow
from oracle import schedule as osched

pytestmark = pytest.mark.gpu
DEV = "cuda"


def rel_l2(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a - b).norm() / (b.norm() + 1e-30))


def rnd(*shape, seed=0, scale=1.0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale


def smooth_flow(n, h, w, seed, amp=1.5):
    g = rnd(n, 2, 8, 8, seed=seed, scale=amp)
    return F.interpolate(g, size=(h, w), mode="bilinear", align_corners=True).contiguous()


def coef_table(buf, ori):
    S = len(ori)
    t = torch.zeros(S, 8)
    t[:, 0] = buf["sqrt_recip_alphas_cumprod"]
    t[:, 1] = buf["sqrt_recipm1_alphas_cumprod"]
    t[:, 2] = buf["posterior_mean_coef1"]
    t[:, 3] = buf["posterior_mean_coef2"]
    t[:, 4] = buf["posterior_log_variance_clipped"]
    t[:, 5] = 1.0
    t[0, 5] = 0.0
    t[:, 6] = torch.tensor(ori, dtype=torch.float32)
    return t


@pytest.mark.parametrize("step", [0, 1, 25, 49])
def test_ddpm_step(hip, step):
    _, buf, ori = osched.respaced_schedule(50)
    coef = coef_table(buf, ori).to(DEV)
    n, c, h, w = 5, 4, 16, 16
    x, eps, noise = rnd(n, c, h, w, seed=1), rnd(n, c, h, w, seed=2), rnd(n, c, h, w, seed=3)
    eps_tok = eps.permute(0, 2, 3, 1).reshape(n * h * w, c).contiguous()
    z = torch.empty(n, c, h, w, device=DEV)
    sidx = torch.tensor([step], dtype=torch.int32, device=DEV)
    hip.ddpm_step(x.to(DEV), eps_tok.to(DEV), noise.to(DEV), coef, sidx, z)
    ref, _ = osched.p_step(buf, step, x, eps, noise)
    assert rel_l2(z.cpu(), ref) < 1e-6
    tv = torch.empty(n, device=DEV)
    hip.step_timestep(coef, sidx, tv)
    assert float(tv[0]) == float(ori[step])
    hip.step_advance(sidx, -1)
    assert int(sidx[0]) == step - 1


def test_flow_warp(hip):
    n, c, h, w = 4, 4, 24, 20
    x = rnd(n, c, h, w, seed=4)
    flow = smooth_flow(n, h, w, 5, amp=4.0)
    flow[0, :, :4] += 30.0      # far out of bounds
    flow[1, :, :, :3] -= 0.5    # straddling the border
    flow[2] = 0.0               # identity
    out = torch.empty_like(x, device=DEV)
    hip.flow_warp(x.to(DEV), flow.to(DEV), out)
    ref = oflow.flow_warp(x, flow.permute(0, 2, 3, 1))
    assert float((out.cpu() - ref).abs().max()) < 2e-5
    assert torch.allclose(out[2].cpu(), x[2], atol=1e-6)


@pytest.mark.parametrize("T,h,w", [(5, 16, 16), (3, 12, 20), (2, 8, 8), (8, 64, 64)])
def test_guidance(hip, T, h, w):
    c = 4
    _, buf, ori = osched.respaced_schedule(50)
    coef = coef_table(buf, ori).to(DEV)
    z = rnd(T, c, h, w, seed=6, scale=0.8)
    ff, fb = smooth_flow(T - 1, h, w, 7), smooth_flow(T - 1, h, w, 8)
    focc, bocc = oflow.forward_backward_consistency_check(fb, ff)
    # make sure both mask values occur
    focc[:, :2] = 1.0
    step = 30
    gscale = -10.0
    ref, loss_ref = oflow.guidance_update(z, (ff[None], fb[None]), (focc[None, :, None], bocc[None, :, None]), T, gscale,
                                          buf["posterior_log_variance_clipped"][step])
    work = torch.empty(hip.guidance_work_bytes(T, c, h, w), dtype=torch.uint8, device=DEV)
    sidx = torch.tensor([step], dtype=torch.int32, device=DEV)
    out = torch.empty(T, c, h, w, device=DEV)
    zd = z.to(DEV)
    args = (zd, ff.to(DEV), fb.to(DEV), focc.to(DEV), bocc.to(DEV))
    hip.guidance(*args, coef, sidx, gscale, out, work)
    loss = torch.empty(1, device=DEV)
    hip.guidance_loss(*args, loss, work)
    assert abs(float(loss) - loss_ref) < 1e-5 * max(1.0, abs(loss_ref))
    d = (out.cpu() - ref)
    upd = (ref - z)
    # sign() is discontinuous: allow a vanishing fraction of elements whose |a-b| sits at fp32 rounding level
    bad = (d.abs() > 1e-6 + 1e-4 * upd.abs().max()).float().mean()
    assert float(bad) < 1e-4
    assert rel_l2(out.cpu(), ref) < 1e-4


def test_fb_consistency_and_resize(hip):
    n, h, w = 4, 32, 40
    fwd, bwd = smooth_flow(n, h, w, 9, amp=2.0), smooth_flow(n, h, w, 10, amp=2.0)
    bwd = -fwd + 0.3 * bwd   # partially consistent so both classes occur
    focc = torch.empty(n, h, w, device=DEV)
    bocc = torch.empty(n, h, w, device=DEV)
    hip.fb_consistency(fwd.to(DEV), bwd.to(DEV), 0.01, 0.5, focc, bocc)
    rf, rb = oflow.forward_backward_consistency_check(fwd, bwd)
    assert 0.02 < float(rf.mean()) < 0.98
    assert float((focc.cpu() != rf).float().mean()) < 2e-3 and float((bocc.cpu() != rb).float().mean()) < 2e-3
    out = torch.empty(n, 2, h // 2, w // 2, device=DEV)
    hip.resize_flow(fwd.to(DEV), out)
    assert rel_l2(out.cpu(), oflow.resize_flow(fwd, h // 2, w // 2)) < 1e-6
    out2 = torch.empty(n, 2, 48, 30, device=DEV)
    hip.resize_flow(fwd.to(DEV), out2)
    assert rel_l2(out2.cpu(), oflow.resize_flow(fwd, 48, 30)) < 1e-6


def test_colorfix(hip):
    n, c, h, w = 3, 3, 64, 48
    content = rnd(n, c, h, w, seed=11) * 0.4 + 0.1
    style = rnd(n, c, h, w, seed=12) * 0.2 - 0.3
    out = torch.empty(n, c, h, w, device=DEV)
    work = torch.empty(4 * n * c * h * w, dtype=torch.float32, device=DEV)
    hip.adain(content.to(DEV), style.to(DEV), out, work)
    assert rel_l2(out.cpu(), ocf.adaptive_instance_normalization(content, style)) < 1e-5
    hip.wavelet_reconstruction(content.to(DEV), style.to(DEV), out, work)
    assert rel_l2(out.cpu(), ocf.wavelet_reconstruction(content, style)) < 1e-5


@pytest.mark.parametrize("n,c,h,w", [(2, 3, 512, 512), (1, 3, 33, 31), (1, 2, 1100, 1100)])   # 16 chunks / unaligned planes / the 64-chunk cap
def test_adain_chunked_planes(hip, n, c, h, w):
    """AdaIN statistics over many blocks per plane (fp64 partial sums per 64 KiB chunk, combined in the apply kernel's prologue)"""
    content = rnd(n, c, h, w, seed=14) * 0.4 + 0.1
    style = rnd(n, c, h, w, seed=15) * 0.2 - 0.3
    out = torch.empty(n, c, h, w, device=DEV)
    work = torch.empty(512 * n * c + 8, dtype=torch.float32, device=DEV)
    hip.adain(content.to(DEV), style.to(DEV), out, work)
    assert rel_l2(out.cpu(), ocf.adaptive_instance_normalization(content, style)) < 1e-5


def test_tile_ops(hip):
    n, c, H, W = 2, 4, 24, 32
    src = rnd(n, c, H, W, seed=13)
    dst = torch.empty(n, c, 16, 16, device=DEV)
    hip.crop(src.to(DEV), dst, 8, 16)
    assert torch.equal(dst.cpu(), src[:, :, 8:24, 16:32])
    acc = torch.zeros(n, c, H, W, device=DEV)
    cnt = torch.zeros(n, c, H, W, device=DEV)
    wgt = torch.rand(16, 16, generator=torch.Generator().manual_seed(14)) + 0.1
    racc, rcnt = torch.zeros(n, c, H, W), torch.zeros(n, c, H, W)
    for (y0, x0) in [(0, 0), (0, 16), (8, 0), (8, 16), (4, 8)]:
        tile = src[:, :, y0:y0 + 16, x0:x0 + 16].contiguous() * 1.5
        hip.tile_accumulate(tile.to(DEV), wgt.to(DEV), acc, cnt, y0, x0)
        racc[:, :, y0:y0 + 16, x0:x0 + 16] += tile * wgt
        rcnt[:, :, y0:y0 + 16, x0:x0 + 16] += wgt
    out = torch.empty_like(acc)
    hip.tile_normalize(acc, cnt, out)
    assert rel_l2(out.cpu(), racc / rcnt) < 1e-6


# ------------------------------------------------------------------------------------------------------------------
# K11: pre/post-processing kernels vs the oracle restatement of the script's torch calls
# ------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("T,h,w,upscale", [(2, 128, 128, 4.0), (1, 90, 160, 4.0), (2, 45, 64, 4.0), (1, 270, 480, 4.0)])
def test_preproc_upsample_pad_flowinput(hip, T, h, w, upscale):
    from mgld_vsr_amd import preproc
    from oracle import preproc as opre
    x = (torch.rand(T, 3, h, w, generator=torch.Generator().manual_seed(5)) * 2 - 1)
    up = preproc.upsample_lr(x, upscale)
    ref = opre.upsample_lr(x, upscale)
    assert up.shape == ref.shape
    assert float((up.cpu() - ref).abs().max()) < 2e-6          # same taps, same weights; only fma contraction differs
    pad, oh, ow = preproc.pad_to_32(up)
    rpad, roh, row = opre.pad_to_32(ref)
    assert (oh, ow) == (roh, row) and pad.shape == rpad.shape
    assert torch.equal(pad.cpu(), opre.pad_to_32(up.cpu())[0])  # pure data movement: bit-exact on the same input
    fi = preproc.flow_input(pad)
    assert float((fi.cpu() - opre.flow_input(pad.cpu())).abs().max()) < 2e-6


def test_preproc_png_payload(hip):
    from mgld_vsr_amd import preproc
    from oracle import preproc as opre
    g = torch.Generator().manual_seed(6)
    out = torch.rand(2, 3, 96, 128, generator=g)
    out[0, 0, 0, :4] = torch.tensor([0.0, 1.0, 0.5, 254.9999 / 255.0])
    got = preproc.to_png_payload(out.cuda(), 90, 120)
    ref = opre.to_png_payload(out, 90, 120)
    assert got.dtype == ref.dtype and got.shape == ref.shape == (2, 90, 120, 3)
    assert (got == ref).all()                                   # integer payload: bit-exact


@pytest.mark.parametrize("h,w,size", [(128, 192, 64), (96, 64, 64), (160, 160, 96), (64, 96, 64)])
def test_resize_center_crop(hip, h, w, size):
    """the fixed-size scripts' Resize(input_size) + CenterCrop(input_size) (old.py:253-256) as one device kernel vs torch"""
    from oracle import preproc as opre
    x = rnd(2, 3, h, w, seed=91)
    got = hip.resize_center_crop(x.to(DEV), size).cpu()
    ref = opre.resize_center_crop(x, size)
    assert got.shape == ref.shape == (2, 3, size, size)
    assert float((got - ref).abs().max()) < 2e-6


def test_image_spliter_device_vs_reference_fixture(hip):
    """scripts.util_image.ImageSpliterTh on device tensors (crop / accumulate / normalise kernels) against outputs of the
    REFERENCE class captured in tests/golden/g_spliter.npz (iteration order, index tuples, uniform-count gather; sf 1 and 2),
    and — at the script's patch settings (960 / 750) — patch contents against plain slicing of the input."""
    import numpy as np
    from scripts.util_image import ImageSpliterTh
    from test_host_cpu import _spliter_case
    g = np.load(os.path.join(HERE, "golden", "g_spliter.npz"))
    for sf in (1, 2):
        idx, out = _spliter_case(ImageSpliterTh, g, sf, to_dev=lambda t: t.cuda())
        assert (idx == g[f"it_index_sf{sf}"]).all()
        assert torch.allclose(out, torch.from_numpy(g[f"it_gather_sf{sf}"]), atol=1e-6, rtol=0)
    im = torch.randn(2, 3, 1024, 1100, generator=torch.Generator().manual_seed(11))
    dev = ImageSpliterTh(im.cuda(), 960, 750, sf=1)
    assert dev.height_starts_list == g["h_1024_960_750"].tolist() and len(dev) == 4
    for pd, (h0, h1, w0, w1) in dev:
        assert torch.equal(pd.cpu(), im[:, :, h0:h1, w0:w1])


# ------------------------------------------------------------------------------------------------------------------
# direct checks of the smaller C-ABI entry points added for RAFT / the text tower / the hoisted tables
# ------------------------------------------------------------------------------------------------------------------
def test_copy_step_selects_slice(hip):
    tab = torch.randn(7, 33, 16, generator=torch.Generator().manual_seed(1)).half().cuda()
    dst = torch.empty(33, 16, dtype=torch.half, device="cuda")
    for i in (0, 3, 6):
        hip.copy_step(tab, dst, torch.tensor([i], dtype=torch.int32, device="cuda"))
        assert torch.equal(dst, tab[i])


def test_replicate_pad_and_avgpool(hip):
    x = torch.randn(2, 3, 13, 17, generator=torch.Generator().manual_seed(2))
    pad = (2, 1, 0, 3)
    assert torch.equal(hip.replicate_pad(x.cuda(), pad).cpu(), F.pad(x, pad, mode="replicate"))
    p = torch.randn(10, 9, 14, generator=torch.Generator().manual_seed(3))
    assert torch.allclose(hip.avgpool2(p.cuda()).cpu(), F.avg_pool2d(p[:, None], 2, stride=2)[:, 0], atol=1e-6)


def test_softmax_rows_masked(hip):
    L, Lp, H = 77, 80, 3
    S = torch.randn(H * L, L, generator=torch.Generator().manual_seed(4)) * 3
    P = torch.full((H * L, Lp), 7.0, dtype=torch.half, device="cuda")
    hip.softmax_rows_masked(S.cuda(), P, H * L, L, Lp, L)
    mask = torch.full((L, L), float("-inf")).triu_(1)
    ref = torch.softmax(S.view(H, L, L) + mask, dim=-1).view(H * L, L)
    assert float((P[:, :L].cpu().float() - ref).abs().max()) < 1e-3
    assert float(P[:, L:].abs().max()) == 0.0                      # K padding columns are zero-filled


def test_corr_lookup_and_convex_upsample_vs_oracle(hip):
    from oracle import raft as oraft
    g = torch.Generator().manual_seed(5)
    B, D, H, W = 2, 32, 16, 20
    f1, f2 = torch.randn(B, D, H, W, generator=g), torch.randn(B, D, H, W, generator=g)
    cb = oraft.CorrBlock(f1, f2, num_levels=4, radius=4)
    coords = oraft.coords_grid(B, H, W) + torch.randn(B, 2, H, W, generator=g) * 2.5
    ref = cb(coords)                                                                  # [B, 324, H, W]
    levels = [p[:, 0].contiguous().cuda() for p in cb.pyramid]                        # [B*H*W, h_l, w_l]
    out = torch.zeros(B * H * W, 324, dtype=torch.float32, device="cuda")
    hip.corr_lookup(levels, coords.cuda().contiguous(), 4, out)
    got = out.cpu().view(B, H, W, 324).permute(0, 3, 1, 2)
    assert rel_l2(got, ref) < 1e-6                                                    # fp32 end to end (round 5)
    flow = torch.randn(B, 2, H, W, generator=g)
    mask = torch.randn(B, 576, H, W, generator=g)
    up = hip.convex_upsample(flow.cuda().contiguous(), mask.permute(0, 2, 3, 1).reshape(B * H * W, 576).cuda().contiguous())
    assert rel_l2(up.cpu(), oraft.upsample_flow(flow, mask)) < 1e-6


def test_gru_and_flow_update_kernels(hip):
    g = torch.Generator().manual_seed(6)
    M, Ch, Cx = 300, 128, 256
    hx = torch.randn(M, Ch + Cx, generator=g).cuda()
    r, z, q = (torch.rand(M, Ch, generator=g).cuda() for _ in range(3))
    rhx = torch.empty_like(hx)
    hip.gru_rh(r, hx, rhx, Ch)
    assert torch.allclose(rhx[:, :Ch], r * hx[:, :Ch], atol=1e-6) and torch.equal(rhx[:, Ch:], hx[:, Ch:])
    h0 = hx[:, :Ch].clone()
    hip.gru_gate(z, q, hx[:, :Ch])
    assert torch.allclose(hx[:, :Ch], (1 - z) * h0 + z * q, atol=1e-6)
    with pytest.raises(RuntimeError):
        hip.gru_gate(z.half(), q.half(), hx[:, :Ch].half())                      # the RAFT kernels are fp32 only
    B, H, W = 2, 5, 6
    c0 = torch.randn(B, 2, H, W, generator=g).cuda()
    c1 = c0 + 1.0
    d = torch.randn(B * H * W, 8, generator=g).cuda()
    flow = torch.empty_like(c0)
    mot = torch.zeros(B * H * W, 8, dtype=torch.float32, device="cuda")
    c1_ref = c1 + d[:, :2].reshape(B, H, W, 2).permute(0, 3, 1, 2)
    hip.flow_update(c1, c0, d[:, :2], flow, mot=mot[:, 6:8])
    assert torch.allclose(c1, c1_ref) and torch.allclose(flow, c1_ref - c0)
    assert torch.equal(mot[:, 6:8], flow.permute(0, 2, 3, 1).reshape(-1, 2)) and float(mot[:, :6].abs().max()) == 0.0


@pytest.mark.parametrize("geo", [
    # n, h, w, cin, cout, ksize, stride, pad, act            (every convolution shape of RAFT_SR, raft_arch.py)
    (2, 31, 29, 3, 64, (7, 7), 2, (3, 3), "none"),            # encoder stem: RGB padded to 4 columns, odd sizes
    (2, 16, 20, 64, 96, (3, 3), 2, (1, 1), "relu"),           # strided ResidualBlock conv
    (2, 16, 20, 64, 96, (1, 1), 2, (0, 0), "none"),           # downsample branch
    (3, 9, 11, 384, 256, (1, 5), 1, (0, 2), "sigmoid"),       # SepConvGRU horizontal z|r
    (3, 9, 11, 384, 128, (5, 1), 1, (2, 0), "tanh"),          # SepConvGRU vertical q
    (3, 9, 11, 2, 128, (7, 7), 1, (3, 3), "relu"),            # flow conv: 2 used channels of a 4-column buffer
    (1, 9, 11, 324, 256, (1, 1), 1, (0, 0), "relu"),          # correlation features (324 = 20 slices + 4)
    (1, 9, 11, 256, 126, (3, 3), 1, (1, 1), "relu"),          # 126 output channels into a column slice
    (1, 9, 11, 256, 2, (3, 3), 1, (1, 1), "none"),            # flow head
])
def test_conv_f32_vs_torch(hip, geo):
    """mgld_conv_f32 (f32-input MFMA implicit GEMM) vs F.conv2d in fp64: fp32 round-off only"""
    from mgld_vsr_amd.raft import pack_conv_f32
    n, h, w, cin, cout, ks, st, pad, act = geo
    g = torch.Generator().manual_seed(11)
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, *ks, generator=g) / (cin * ks[0] * ks[1]) ** 0.5
    b = torch.randn(cout, generator=g) * 0.1
    ref = F.conv2d(x.double(), wt.double(), b.double(), stride=st, padding=pad)
    ref = {"none": lambda t: t, "relu": torch.relu, "sigmoid": torch.sigmoid, "tanh": torch.tanh}[act](1.5 * ref)
    ho, wo = ref.shape[-2:]
    ld_in = (cin + 3) // 4 * 4
    xin = torch.zeros(n * h * w, ld_in + 4, device="cuda")
    xin[:, :cin] = x.permute(0, 2, 3, 1).reshape(-1, cin).cuda()
    out = torch.full((n * ho * wo, cout + 6), 7.0, device="cuda")
    code = {"none": hip.ACT_NONE, "relu": hip.ACT_RELU, "sigmoid": hip.ACT_SIGMOID, "tanh": hip.ACT_TANH}[act]
    hip.conv_f32(xin[:, :ld_in], pack_conv_f32(wt).cuda(), out[:, 2:2 + cout], n, h, w, cin, ks, st, pad, bias=b.cuda(), act=code, alpha=1.5)
    got = out[:, 2:2 + cout].cpu().view(n, ho, wo, cout).permute(0, 3, 1, 2)
    assert rel_l2(got, ref.float()) < 2e-6
    assert float((out[:, :2] - 7.0).abs().max()) == 0.0 and float((out[:, 2 + cout:] - 7.0).abs().max()) == 0.0     # slice only
    # ResidualBlock tail in the epilogue: relu(skip + act(.))
    skip = torch.randn(n * ho * wo, cout, generator=g).cuda()
    out2 = torch.empty(n * ho * wo, cout, device="cuda")
    hip.conv_f32(xin[:, :ld_in], pack_conv_f32(wt).cuda(), out2, n, h, w, cin, ks, st, pad, bias=b.cuda(), act=code, alpha=1.5, resid=skip,
                 post_relu=True)
    ref2 = torch.relu(ref.permute(0, 2, 3, 1).reshape(-1, cout).float() + skip.cpu())
    assert rel_l2(out2.cpu(), ref2) < 2e-6


def test_conv_f32_batched_correlation(hip):
    """the all-pairs correlation (raft_arch.py:82-85) as the batched LINEAR form: per pair fmap1 [hw,256] . fmap2^T / 16"""
    g = torch.Generator().manual_seed(12)
    B, hw, D = 3, 15 * 17, 256
    f = torch.randn(2 * B * hw, D, generator=g).cuda()
    corr = torch.empty(B * hw, hw, device="cuda")
    hip.conv_f32(f[:B * hw], f[B * hw:], corr, 1, 15, 17, D, alpha=1.0 / 16.0, batch=B, strideA=hw * D, strideW=hw * D, strideC=hw * hw,
                 n_out=hw)
    ref = torch.matmul(f[:B * hw].cpu().double().view(B, hw, D), f[B * hw:].cpu().double().view(B, hw, D).transpose(1, 2)) / 16.0
    assert rel_l2(corr.cpu().view(B, hw, hw), ref.float()) < 2e-6


@pytest.mark.parametrize("n,h,w,C", [(3, 64, 64, 64), (2, 17, 13, 96), (1, 8, 8, 128)])
def test_instnorm_f32_vs_torch(hip, n, h, w, C):
    """mgld_instnorm_f32 vs F.instance_norm (no affine, eps 1e-5) incl. ReLU and the relu(skip + y) tail; a channel with a large
    mean (the fp64 sums are there for it)"""
    g = torch.Generator().manual_seed(13)
    x = torch.randn(n, C, h, w, generator=g)
    x[:, 5] += 300.0
    skip = torch.randn(n, C, h, w, generator=g)
    xin = x.permute(0, 2, 3, 1).reshape(-1, C).cuda().contiguous()
    sk = skip.permute(0, 2, 3, 1).reshape(-1, C).cuda().contiguous()
    part = torch.empty(n * hip.instnorm_chunks(h * w) * C * 2, dtype=torch.float64, device="cuda")
    ref = F.instance_norm(x.double(), eps=1e-5)
    for relu, use_skip in ((False, False), (True, False), (True, True)):
        out = torch.empty_like(xin)
        hip.instnorm_f32(xin, part, out, n, h * w, 1e-5, relu, skip=sk if use_skip else None)
        r = torch.relu(ref) if relu else ref
        if use_skip:
            r = torch.relu(r + skip.double())
        got = out.cpu().view(n, h, w, C).permute(0, 3, 1, 2)
        assert float((got - r.float()).abs().max()) < 2e-4 and rel_l2(got, r.float()) < 2e-6
