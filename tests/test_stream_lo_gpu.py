"""The residual stream as two fp16 planes (value = hi + 2^-11 lo; MgldIGemm.Rlo / Clo, mgld_*_lo): every kernel family that adds a residual
or writes one, and the normalisations that read it, through the C ABI against fp64 torch-CPU math of the same operands.  What is asserted:
the two-plane result carries the fp32 value the epilogue computed (error ~1e-6, three orders under the 2^-11 of the hi plane alone), the
hi plane IS fp16 of it, and the residual's low plane was added."""
import pytest
import torch
import torch.nn.functional as F

from test_kernels_gpu import DEV, R3_BN, _from_tok, _to_tok, h16, rel_l2, rnd

pytestmark = pytest.mark.gpu

LO = 2.0 ** -11


def planes(x32):
    """fp32 -> (hi, lo) fp16 planes, and the value they carry"""
    hi = x32.half()
    lo = ((x32 - hi.float()) * 2048.0).half()
    return hi, lo, hi.double() + LO * lo.double()


def check_planes(out_hi, out_lo, ref64, tol=4e-6):
    val = out_hi.double() + LO * out_lo.double()
    e2, e1 = rel_l2(val, ref64), rel_l2(out_hi, ref64)
    assert e2 < tol, (e2, e1)
    assert e1 > 20 * e2, (e2, e1)                                   # (the low plane is doing the work)
    # the hi plane is the value rounded ONCE; the low plane is what the rounding dropped, to fp16's own precision
    assert (out_hi.double() - ref64).abs().max() <= (ref64.abs() * 2.0 ** -11 + 1e-7).max()
    return e2


@pytest.mark.parametrize("M,N,K,tune,act", [(4096, 320, 320, 20, 0), (512, 1280, 1280, 20, 3), (1024, 640, 640, 21 + 1, 0), (2048, 320, 320, 21 + 6, 0),
                                            (100, 72, 40, 0, 0), (300, 200, 2048, 0, 0), (512, 1280, 11520, 0, 3), (64, 1280, 23040, 0, 0)])
def test_linear_lo_planes(hip, M, N, K, tune, act):
    """LINEAR: ping-pong tiles (tune 20 / 21 + id), the 128-class kernels, the K split (splitk_reduce does the epilogue)"""
    a, w, b = h16(rnd(M, K, seed=1)), h16(rnd(N, K, seed=2, scale=K ** -0.5)), rnd(N, seed=3)
    rh, rl, rv = planes(rnd(M, N, seed=4) * 3.0)
    pre = a.double() @ w.double().t() + b.double()
    ref = 0.5 * (F.silu(pre) if act == 3 else pre) + 2.0 * rv
    out = torch.full((M, N), float("nan"), dtype=torch.half, device=DEV)
    olo = torch.full_like(out, float("nan"))
    hip.igemm(a.to(DEV), w.to(DEV), out, bias=b.to(DEV), resid=rh.to(DEV), resid_lo=rl.to(DEV), out_lo=olo, act=act, alpha=0.5, beta=2.0, tune=tune)
    torch.cuda.synchronize()
    check_planes(out.cpu(), olo.cpu(), ref, tol=4e-6 if K < 4096 else 1.5e-5)
    # without the residual's low plane the result moves by what that plane carries
    out2, olo2 = torch.empty_like(out), torch.empty_like(out)
    hip.igemm(a.to(DEV), w.to(DEV), out2, bias=b.to(DEV), resid=rh.to(DEV), out_lo=olo2, act=act, alpha=0.5, beta=2.0, tune=tune)
    torch.cuda.synchronize()
    d = (out.cpu().double() + LO * olo.cpu().double()) - (out2.cpu().double() + LO * olo2.cpu().double())
    assert rel_l2(d, 2.0 * LO * rl.double()) < 2e-2


def test_linear_lo_planes_strided_slices(hip):
    """residual and output as column slices of wider buffers (the UNet's concat slots): planes share the leading dimension"""
    M, N, K = 1024, 320, 640
    a, w = h16(rnd(M, K, seed=11)), h16(rnd(N, K, seed=12, scale=K ** -0.5))
    rh, rl, rv = planes(rnd(M, N, seed=13))
    Rh, Rl = torch.zeros(M, N + 64, dtype=torch.half, device=DEV), torch.zeros(M, N + 64, dtype=torch.half, device=DEV)
    Rh[:, 32:32 + N], Rl[:, 32:32 + N] = rh.to(DEV), rl.to(DEV)
    Oh, Ol = torch.zeros(M, 2 * N, dtype=torch.half, device=DEV), torch.zeros(M, 2 * N, dtype=torch.half, device=DEV)
    hip.igemm(a.to(DEV), w.to(DEV), Oh[:, N:], resid=Rh[:, 32:32 + N], resid_lo=Rl[:, 32:32 + N], out_lo=Ol[:, N:])
    torch.cuda.synchronize()
    check_planes(Oh[:, N:].cpu(), Ol[:, N:].cpu(), a.double() @ w.double().t() + rv)
    assert not Oh[:, :N].any() and not Ol[:, :N].any()


def _conv_case(hip, n, cin, cout, h, w, tune, stats=False, seed=50):
    from mgld_vsr_amd.engine import tile_conv3p
    x = h16(rnd(n, cin, h, w, seed=seed))
    wt = h16(rnd(cout, cin, 3, 3, seed=seed + 1, scale=(9 * cin) ** -0.5))
    b = rnd(cout, seed=seed + 2)
    rh, rl, rv = planes(rnd(n * h * w, cout, seed=seed + 3) * 2.0)
    ref = _to_tok(F.conv2d(x.double(), wt.double(), b.double(), padding=1)) + rv
    wk = wt.permute(0, 2, 3, 1).reshape(cout, 9 * cin).contiguous().to(DEV)
    out = torch.full((n * h * w, cout), float("nan"), dtype=torch.half, device=DEV)
    olo = torch.full_like(out, float("nan"))
    kw = {}
    got = []
    if stats:
        def part(chunks):
            got.append((torch.empty(n * chunks, 2, cout, dtype=torch.float32, device=DEV), chunks))
            return got[0][0]
        kw["gn_part"] = part
    hip.igemm(_to_tok(x).to(DEV), tile_conv3p(wk, cin, False), out, mode=hip.MODE_CONV3X3, conv=(cin, h, w, h, w, 1, 1, 1, 0), tap_inner=2, N=cout,
              K=9 * cin, tune=tune, bias=b.to(DEV), resid=rh.to(DEV), resid_lo=rl.to(DEV), out_lo=olo, **kw)
    torch.cuda.synchronize()
    check_planes(out.cpu(), olo.cpu(), ref)
    if stats:
        assert got, "this configuration writes the statistics of its output"
        part_t, chunks = got[0]
        tot = part_t.cpu().double().reshape(n, chunks, 2, cout).sum(1)
        rf = ref.reshape(n, h * w, cout)
        assert rel_l2(tot[:, 0], rf.sum(1)) < 1e-4 and rel_l2(tot[:, 1], (rf * rf).sum(1)) < 1e-5


@pytest.mark.parametrize("cfg,n,cin,nt,h,w,stats", [(0, 2, 64, 1, 24, 40, False), (6, 3, 320, 4, 64, 64, True), (8, 1, 128, 2, 72, 80, True), (5, 2, 96, 1, 16, 16, False),
                                                    (2, 2, 32, 3, 32, 32, True), (3, 1, 64, 1, 16, 32, False), (7, 2, 128, 3, 32, 32, True)])
def test_conv3x3_pingpong_lo_planes(hip, cfg, n, cin, nt, h, w, stats):
    """conv3r (tune 31 + id): the plain register epilogue and the statistics-writing one (ResnetBlock's `x + h` with the next norm's sums)"""
    _conv_case(hip, n, cin, nt * R3_BN[cfg], h, w, 31 + cfg, stats)


@pytest.mark.parametrize("cfg,n,cin,nt", [(5, 8, 1280, 8), (8, 8, 640, 10), (4, 3, 320, 2)])
def test_conv3x3_pingpong_ksplit_lo_planes(hip, cfg, n, cin, nt):
    """conv3r with the channel slices over grid.z (tune 50 + id): the reduce kernel writes both planes"""
    _conv_case(hip, n, cin, nt * R3_BN[cfg], 16, 16, 50 + cfg)


@pytest.mark.parametrize("variant,n,cin,cout,h,w", [(0, 2, 64, 96, 24, 40), (1, 1, 128, 128, 32, 32), (4, 8, 1280, 1280, 8, 8), (7, 2, 64, 64, 16, 16)])
def test_conv3x3_tile2d_lo_planes(hip, variant, n, cin, cout, h, w):
    """conv3q (the 128-class 2-D-tile patch convolution, tune = variant + 1; the 8^2 level runs it with a K split)"""
    _conv_case(hip, n, cin, cout, h, w, variant + 1)


def test_conv3x3_gather_lo_planes(hip):
    """the implicit-GEMM kernel itself ([N, K] weights, stride 2: the Downsample convolution writes the stream's next value)"""
    n, cin, cout, h, w = 2, 64, 96, 16, 16
    x = h16(rnd(n, cin, h, w, seed=70))
    wt = h16(rnd(cout, cin, 3, 3, seed=71, scale=(9 * cin) ** -0.5))
    ref = _to_tok(F.conv2d(x.double(), wt.double(), None, stride=2, padding=1))
    wk = wt.permute(0, 2, 3, 1).reshape(cout, 9 * cin).contiguous().to(DEV)
    out = torch.full((n * 64, cout), float("nan"), dtype=torch.half, device=DEV)
    olo = torch.full_like(out, float("nan"))
    hip.igemm(_to_tok(x).to(DEV), wk, out, mode=hip.MODE_CONV3X3, conv=(cin, h, w, 8, 8, 2, 1, 1, 0), out_lo=olo)
    torch.cuda.synchronize()
    check_planes(out.cpu(), olo.cpu(), ref)


@pytest.mark.parametrize("clips,T,c,h,w,tune", [(1, 8, 128, 8, 8, 40), (2, 8, 256, 16, 8, 40), (1, 4, 512, 16, 16, 40), (2, 5, 64, 8, 8, 0), (1, 8, 1280, 8, 8, 0)])
def test_tconv_lo_planes(hip, clips, T, c, h, w, tune):
    """SpatialTemporalConv `a conv3d(x) + (1 - a) x`: x is operand (hi plane) AND residual (both planes); ping-pong tiles and the implicit GEMM"""
    xh, xl, xv = planes(_to_tok(rnd(clips * T, c, h, w, seed=80)))
    w32 = h16(rnd(c, c, 3, 1, 1, seed=81, scale=(3 * c) ** -0.5))
    b = rnd(c, seed=82)
    alpha = 0.6
    x5 = _from_tok(xh.double(), clips * T, h, w).reshape(clips, T, c, h, w).permute(0, 2, 1, 3, 4)
    res = F.conv3d(x5, w32.double(), b.double(), padding=(1, 0, 0)).permute(0, 2, 1, 3, 4).reshape(clips * T, c, h, w)
    ref = alpha * _to_tok(res) + (1 - alpha) * xv
    wk = w32[:, :, :, 0, 0].permute(0, 2, 1).reshape(c, 3 * c).contiguous()
    out = torch.full((clips * T * h * w, c), float("nan"), dtype=torch.half, device=DEV)
    olo = torch.full_like(out, float("nan"))
    xd = xh.to(DEV)
    hip.igemm(xd, wk.to(DEV), out, mode=hip.MODE_TCONV3, bias=b.to(DEV), resid=xd, resid_lo=xl.to(DEV), out_lo=olo, alpha=alpha, beta=1 - alpha,
              tconv=(c, T, h * w), tune=tune)
    torch.cuda.synchronize()
    check_planes(out.cpu(), olo.cpu(), ref)


def test_lo_planes_refused_where_they_cannot_go(hip):
    a, w = h16(rnd(128, 64, seed=90)).to(DEV), h16(rnd(128, 64, seed=91)).to(DEV)
    o16, o32 = torch.empty(128, 128, dtype=torch.half, device=DEV), torch.empty(128, 128, dtype=torch.float32, device=DEV)
    with pytest.raises(RuntimeError):
        hip.igemm(a, w, torch.empty(128, 64, dtype=torch.half, device=DEV), act=hip.ACT_GEGLU, out_lo=torch.empty(128, 64, dtype=torch.half, device=DEV), N=128)
    p = hip.MgldIGemm()
    p.A, p.W, p.C, p.Clo = a.data_ptr(), w.data_ptr(), o32.data_ptr(), o16.data_ptr()
    p.M, p.N, p.K, p.lda, p.ldw, p.ldc, p.batch, p.out_f32, p.alpha, p.beta = 128, 128, 64, 64, 64, 128, 1, 1, 1.0, 1.0
    with pytest.raises(RuntimeError):
        hip.igemm_relaunch(p)
    p.out_f32, p.C, p.Clo, p.Rlo = 0, o16.data_ptr(), None, o16.data_ptr()      # a low plane of a residual that is not there
    with pytest.raises(RuntimeError):
        hip.igemm_relaunch(p)


# ---- normalisations that read the stream --------------------------------------------------------------------------------------------
@pytest.mark.parametrize("frames,C,h,w,silu", [(2, 320, 64, 64, 1), (3, 640, 32, 32, 1), (1, 2560, 20, 24, 0), (2, 128, 48, 40, 1), (2, 960, 17, 23, 1),
                                               (2, 1280, 16, 16, 1), (3, 1920, 8, 8, 0), (1, 320, 16, 16, 1)])
def test_groupnorm_lo_plane(hip, frames, C, h, w, silu):
    """gn_apply / the single-launch form on (x, xlo): what the normalised operand gains is the rounding of x (2.8e-4 of it), so the check is
    against fp64 GroupNorm of the two-plane VALUE with a bound the hi plane alone cannot meet"""
    rows = h * w
    # a stream with a large common offset: the rounding of x (relative to |x|) is large against the spread the norm rescales to one
    x32 = rnd(frames * rows, C, seed=100) * 0.25 + 6.0
    xh, xl, xv = planes(x32)
    gamma, beta = 1 + 0.1 * rnd(C, seed=101), 0.1 * rnd(C, seed=102)
    ld = C + 24
    Xh, Xl = torch.zeros(frames * rows, ld, dtype=torch.half, device=DEV), torch.zeros(frames * rows, ld, dtype=torch.half, device=DEV)
    Xh[:, 8:8 + C], Xl[:, 8:8 + C] = xh.to(DEV), xl.to(DEV)
    xv_h, xv_l = Xh[:, 8:8 + C], Xl[:, 8:8 + C]
    ref = F.group_norm(_from_tok(xv, frames, h, w), 32, gamma.double(), beta.double(), 1e-5)
    ref = F.silu(ref) if silu else ref
    y = torch.full((frames * rows, C), float("nan"), dtype=torch.half, device=DEV)
    y0 = torch.empty_like(y)
    if hip.gn_fused_applies(rows, C, 32):
        hip.gn_fused(xv_h, 1e-5, gamma.to(DEV), beta.to(DEV), y, frames, rows, 32, silu, lo_in=xv_l)
        hip.gn_fused(xv_h, 1e-5, gamma.to(DEV), beta.to(DEV), y0, frames, rows, 32, silu)
    else:
        gsums = torch.empty(frames, hip.gn_chunks(rows), 32, 2, dtype=torch.float64, device=DEV)
        hip.gn_stats(xv_h, frames, rows, 32, gsums)
        hip.gn_apply(xv_h, gsums, 1e-5, gamma.to(DEV), beta.to(DEV), y, frames, rows, 32, silu, x_lo=xv_l)
        hip.gn_apply(xv_h, gsums, 1e-5, gamma.to(DEV), beta.to(DEV), y0, frames, rows, 32, silu)
    torch.cuda.synchronize()
    e_lo, e_hi = rel_l2(_from_tok(y.cpu().float(), frames, h, w), ref), rel_l2(_from_tok(y0.cpu().float(), frames, h, w), ref)
    assert e_lo < 4e-4, (e_lo, e_hi)                 # = the fp16 rounding of y itself
    assert e_hi > 3 * e_lo, (e_lo, e_hi)             # the hi plane alone: x's rounding, amplified by 6.0 / 0.25


@pytest.mark.parametrize("frames,C,h,w,want_stats", [(2, 320, 64, 64, True), (2, 640, 32, 32, False), (3, 1280, 16, 16, False), (2, 1280, 8, 8, False), (1, 960, 24, 20, True)])
def test_spade_lo_planes(hip, frames, C, h, w, want_stats):
    """ResBlockDual's output `skip + spade(h)`: skip read as two planes, the result written as two planes (+ the next norm's statistics)"""
    rows = h * w
    hh = h16(rnd(frames * rows, C, seed=110))
    sh, sl, sv = planes(rnd(frames * rows, C, seed=111) * 4.0)
    gb = h16(rnd(frames * rows, 2 * C, seed=112, scale=0.5))
    gamma, beta = 1 + 0.1 * rnd(C, seed=113), 0.1 * rnd(C, seed=114)
    hn = F.group_norm(_from_tok(hh.double(), frames, h, w), 32, gamma.double(), beta.double(), 1e-5)
    ref = _to_tok(hn) * (1 + gb.double()[:, :C]) + gb.double()[:, C:] + sv
    y = torch.full((frames * rows, C), float("nan"), dtype=torch.half, device=DEV)
    yl = torch.full_like(y, float("nan"))
    hd = hh.to(DEV)
    if hip.gn_fused_applies(rows, C, 32):
        hip.gn_fused(hd, 1e-5, gamma.to(DEV), beta.to(DEV), y, frames, rows, 32, 0, gb=gb.to(DEV), skip=sh.to(DEV), lo_in=sl.to(DEV), lo_out=yl)
    else:
        gsums = torch.empty(frames, hip.gn_chunks(rows), 32, 2, dtype=torch.float64, device=DEV)
        hip.gn_stats(hd, frames, rows, 32, gsums)
        so = None
        if want_stats:
            so = torch.empty(frames, hip.gn_apply_chunks(frames, rows, C, 32), 32, 2, dtype=torch.float64, device=DEV)
        hip.spade_apply(hd, gsums, 1e-5, gamma.to(DEV), beta.to(DEV), gb.to(DEV), sh.to(DEV), y, frames, rows, 32, stats_out=so, skip_lo=sl.to(DEV), y_lo=yl)
        if want_stats:
            torch.cuda.synchronize()
            tot = so.sum(1).cpu()
            rg = _from_tok(ref, frames, h, w).reshape(frames, 32, -1)
            assert rel_l2(tot[:, :, 0], rg.sum(-1)) < 1e-5 and rel_l2(tot[:, :, 1], (rg * rg).sum(-1)) < 1e-5
    torch.cuda.synchronize()
    check_planes(y.cpu(), yl.cpu(), ref, tol=5e-6)


@pytest.mark.parametrize("rows,C", [(130, 320), (64, 640), (37, 1280), (4096, 320)])
def test_layernorm_lo_plane(hip, rows, C):
    xh, xl, xv = planes(rnd(rows, C, seed=120) * 0.25 + 6.0)
    gamma, beta = 1 + 0.1 * rnd(C, seed=121), 0.1 * rnd(C, seed=122)
    ref = F.layer_norm(xv, (C,), gamma.double(), beta.double(), 1e-5)
    y, y0 = torch.empty(rows, C, dtype=torch.half, device=DEV), torch.empty(rows, C, dtype=torch.half, device=DEV)
    hip.layernorm(xh.to(DEV), gamma.to(DEV), beta.to(DEV), y, x_lo=xl.to(DEV))
    hip.layernorm(xh.to(DEV), gamma.to(DEV), beta.to(DEV), y0)
    torch.cuda.synchronize()
    e_lo, e_hi = rel_l2(y.cpu(), ref), rel_l2(y0.cpu(), ref)
    assert e_lo < 4e-4 and e_hi > 3 * e_lo, (e_lo, e_hi)


def test_axpby_lo_planes(hip):
    rows, C = 1000, 256
    xh, xl, xv = planes(rnd(rows, C, seed=130))
    yh, yl, yv = planes(rnd(rows, C, seed=131) * 2.0)
    Y, Yl = torch.zeros(rows, C + 64, dtype=torch.half, device=DEV), torch.zeros(rows, C + 64, dtype=torch.half, device=DEV)
    Y[:, :C], Yl[:, :C] = yh.to(DEV), yl.to(DEV)
    hip.axpby_lo(xh.to(DEV), xl.to(DEV), Y[:, :C], Yl[:, :C], 0.7, 1.0)
    torch.cuda.synchronize()
    check_planes(Y[:, :C].cpu(), Yl[:, :C].cpu(), 0.7 * xv + yv, tol=3e-6)
    hip.axpby_lo(xh.to(DEV), None, Y[:, :C], Yl[:, :C], -0.7, 1.0)                 # a source without a low plane
    torch.cuda.synchronize()
    check_planes(Y[:, :C].cpu(), Yl[:, :C].cpu(), 0.7 * xv + yv - 0.7 * xh.double(), tol=3e-6)
