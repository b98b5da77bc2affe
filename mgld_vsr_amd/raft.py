"""RAFT_SR flow estimator on the MI355X kernels (reference basicsr/archs/raft_arch.py:668-807).

`compute_flow` (ddpm.py:3404-3429) runs this network on the quarter-resolution LR frames right before the hot path; with it
the pipeline no longer needs precomputed flows.  The nn.Modules only own parameters under the reference's state_dict keys
('normal' model: fnet = BasicEncoder(256, instance), cnet = BasicEncoder(256, batch), BasicUpdateBlock with SepConvGRU);
`forward` emits C-ABI launches.  Round 5: the whole network runs in **fp32** — the flows feed a thresholded occlusion check
and sub-pixel warps of the latents, and with fp16 activations through the ten recurrent updates they sat 1.6-2.1e-3 from the
reference's (one flipped mask pixel moved the sampled latents by 4.8e-3), while the network is < 0.1 % of a segment's
arithmetic.  Every convolution and the all-pairs correlation is an `mgld_conv_f32` (implicit GEMM on the f32-input MFMA),
InstanceNorm is `mgld_instnorm_f32` (fp64 sums; the ResidualBlock tail relu(x + y) rides in its apply pass), BatchNorm (eval)
is folded into the preceding convolution when the weights are packed (ReLU, skip add and the tail ReLU in that convolution's
epilogue), the pyramid lookup / GRU gates / convex upsampling are the kernels of raft.hip.  Activations are fp32 NHWC matrices.
No CPU fallback.
"""
import torch
import torch.nn as nn

from . import hip
from .engine import Act

_EPS = 1e-5


def _fold_bn(w, b, g, beta, mean, var):
    """conv followed by eval-mode BatchNorm2d -> one conv"""
    s = g / torch.sqrt(var + _EPS)
    return w * s.view(-1, 1, 1, 1), (b - mean) * s + beta


def _make_norm(kind, planes):
    if kind == "instance":
        return nn.InstanceNorm2d(planes)
    if kind == "batch":
        return nn.BatchNorm2d(planes)
    raise ValueError(kind)


class ResidualBlock(nn.Module):
    """raft_arch.py:89-138"""

    def __init__(self, in_planes, planes, norm_fn="instance", stride=1):
        super().__init__()
        self.conv1 = nn.Conv2d(in_planes, planes, kernel_size=3, padding=1, stride=stride)
        self.conv2 = nn.Conv2d(planes, planes, kernel_size=3, padding=1)
        self.norm_fn, self.stride, self.planes = norm_fn, stride, planes
        self.norm1, self.norm2 = _make_norm(norm_fn, planes), _make_norm(norm_fn, planes)
        if stride != 1:
            self.norm3 = _make_norm(norm_fn, planes)
            self.downsample = nn.Sequential(nn.Conv2d(in_planes, planes, kernel_size=1, stride=stride), self.norm3)
        else:
            self.downsample = None


class BasicEncoder(nn.Module):
    """raft_arch.py:199-272"""

    def __init__(self, output_dim=128, norm_fn="batch", dropout=0.0):
        super().__init__()
        self.norm_fn, self.output_dim = norm_fn, output_dim
        self.norm1 = _make_norm(norm_fn, 64)
        self.conv1 = nn.Conv2d(3, 64, kernel_size=7, stride=2, padding=3)
        self.layer1 = nn.Sequential(ResidualBlock(64, 64, norm_fn, 1), ResidualBlock(64, 64, norm_fn, 1))
        self.layer2 = nn.Sequential(ResidualBlock(64, 96, norm_fn, 2), ResidualBlock(96, 96, norm_fn, 1))
        self.layer3 = nn.Sequential(ResidualBlock(96, 128, norm_fn, 2), ResidualBlock(128, 128, norm_fn, 1))
        self.conv2 = nn.Conv2d(128, output_dim, kernel_size=1)

    # ---- launches ----
    def _conv_norm(self, eng, x, conv, norm, ksize, stride, pad, relu, skip=None):
        """conv -> norm -> (relu) -> (relu(skip + .)).  batch: BatchNorm folded into the weights, everything in the convolution's
        epilogue (one launch); instance: convolution, then statistics + apply (the apply pass carries ReLU and the skip tail)."""
        if self.norm_fn == "batch":
            wp, bp = eng.weight("cbn32", (conv.weight, conv.bias, norm.weight, norm.bias, norm.running_mean, norm.running_var),
                                lambda *a: (lambda wf, bf: (pack_conv_f32(wf), bf))(*_fold_bn(*a)), torch.float32)
            return _conv(eng, x, wp, bp, conv.out_channels, ksize, stride, pad, act=hip.ACT_RELU if relu else hip.ACT_NONE, resid=skip,
                         post_relu=skip is not None)
        y = _conv(eng, x, eng.weight("c32", (conv.weight,), pack_conv_f32, torch.float32), eng.f32("b", conv.bias), conv.out_channels,
                  ksize, stride, pad)
        part = eng.arena.alloc((y.n * hip.instnorm_chunks(y.hw) * y.C * 2,), torch.float64)
        out = eng.act(y.n, y.h, y.w, y.C, torch.float32)
        hip.instnorm_f32(y.v, part, out.v, y.n, y.hw, _EPS, relu, skip=None if skip is None else skip.v)
        eng.launches += 2
        return out

    def _block(self, eng, blk, x):
        """ResidualBlock.forward (raft_arch.py:130-138)"""
        y = self._conv_norm(eng, x, blk.conv1, blk.norm1, (3, 3), blk.stride, (1, 1), True)
        if blk.downsample is not None:
            x = self._conv_norm(eng, x, blk.downsample[0], blk.downsample[1], (1, 1), blk.stride, (0, 0), False)
        return self._conv_norm(eng, y, blk.conv2, blk.norm2, (3, 3), 1, (1, 1), True, skip=x)

    def run(self, eng, x):
        """x: fp32 Act [n,h,w,4] (RGB + one zero column) -> Act [n,h/8,w/8,128] (the caller applies conv2 and the tanh / relu split)"""
        h = self._conv_norm(eng, x, self.conv1, self.norm1, (7, 7), 2, (3, 3), True)
        for layer in (self.layer1, self.layer2, self.layer3):
            for blk in layer:
                h = self._block(eng, blk, h)
        return h


def pack_conv_f32(w):
    """[Cout, Cin, kh, kw] (or [Cout, Cin]) fp32 -> [Cout, kh*kw*Cin4], K index = (ky*kw+kx)*Cin4 + c, Cin4 = Cin rounded up to 4
    (mgld_conv_f32's layout: every float4 of a row lies inside one tap)"""
    if w.dim() == 2:
        w = w[:, :, None, None]
    cout, cin, kh, kw = w.shape
    c4 = (cin + 3) // 4 * 4
    out = torch.zeros(cout, kh * kw, c4, dtype=torch.float32)
    out[:, :, :cin] = w.permute(0, 2, 3, 1).reshape(cout, kh * kw, cin)
    return out.reshape(cout, kh * kw * c4).contiguous()


def _conv(eng, x, wp, bias, cout, ksize=(1, 1), stride=1, pad=(0, 0), act=hip.ACT_NONE, alpha=1.0, out=None, resid=None, post_relu=False,
          cin=None):
    """one mgld_conv_f32 launch on an fp32 Act; out: optional Act view (column slice of a concat buffer)"""
    kh, kw = ksize
    ho = (x.h + 2 * pad[0] - kh) // stride + 1
    wo = (x.w + 2 * pad[1] - kw) // stride + 1
    if out is None:
        out = eng.act(x.n, ho, wo, cout, torch.float32)
    hip.conv_f32(x.v, wp, out.v, x.n, x.h, x.w, cin or x.C, ksize, stride, pad, bias=bias, act=act, alpha=alpha,
                 resid=None if resid is None else resid.v, post_relu=post_relu)
    eng.launches += 1
    return out


class BasicMotionEncoder(nn.Module):
    """raft_arch.py:426-444"""

    def __init__(self, corr_levels, corr_radius):
        super().__init__()
        cor_planes = corr_levels * (2 * corr_radius + 1) ** 2
        self.convc1 = nn.Conv2d(cor_planes, 256, 1, padding=0)
        self.convc2 = nn.Conv2d(256, 192, 3, padding=1)
        self.convf1 = nn.Conv2d(2, 128, 7, padding=3)
        self.convf2 = nn.Conv2d(128, 64, 3, padding=1)
        self.conv = nn.Conv2d(64 + 192, 128 - 2, 3, padding=1)


class SepConvGRU(nn.Module):
    """raft_arch.py:379-405"""

    def __init__(self, hidden_dim=128, input_dim=192 + 128):
        super().__init__()
        for n, k, p in (("1", (1, 5), (0, 2)), ("2", (5, 1), (2, 0))):
            for g in "zrq":
                setattr(self, f"conv{g}{n}", nn.Conv2d(hidden_dim + input_dim, hidden_dim, k, padding=p))


class FlowHead(nn.Module):
    """raft_arch.py:350-358"""

    def __init__(self, input_dim=128, hidden_dim=256):
        super().__init__()
        self.conv1 = nn.Conv2d(input_dim, hidden_dim, 3, padding=1)
        self.conv2 = nn.Conv2d(hidden_dim, 2, 3, padding=1)


class BasicUpdateBlock(nn.Module):
    """raft_arch.py:463-485"""

    def __init__(self, corr_levels, corr_radius, hidden_dim=128, input_dim=128):
        super().__init__()
        self.encoder = BasicMotionEncoder(corr_levels, corr_radius)
        self.gru = SepConvGRU(hidden_dim=hidden_dim, input_dim=128 + hidden_dim)
        self.flow_head = FlowHead(hidden_dim, hidden_dim=256)
        self.mask = nn.Sequential(nn.Conv2d(128, 256, 3, padding=1), nn.ReLU(inplace=True), nn.Conv2d(256, 64 * 9, 1, padding=0))


class RAFT_SR(nn.Module):
    """raft_arch.py:668-807, 'normal' model."""

    def __init__(self, model="normal", load_path=None, **kw):
        super().__init__()
        if model == "small":
            raise NotImplementedError("the shipped config uses the 'normal' RAFT (mgldvsr_512_realbasicvsr_deg.yaml:110-114)")
        self.hidden_dim = self.context_dim = 128
        self.corr_levels, self.corr_radius = 4, 4
        self.fnet = BasicEncoder(output_dim=256, norm_fn="instance")
        self.cnet = BasicEncoder(output_dim=256, norm_fn="batch")
        self.update_block = BasicUpdateBlock(self.corr_levels, self.corr_radius, hidden_dim=128)
        self._engine = None
        if load_path:
            from .util import load_trusted_checkpoint
            sd = load_trusted_checkpoint(load_path)
            self.load_state_dict({(k[7:] if k.startswith("module.") else k): v for k, v in sd.items()})

    def set_engine(self, eng):
        self._engine = eng

    def engine(self):
        if self._engine is None:
            from .engine import Engine
            self._engine = Engine()
        return self._engine

    # ---- one batch of (ref, sup) pairs ----
    @torch.no_grad()
    def forward(self, ref, sup, iters=10, flow_init=None, upsample=True, keep_arena=False):
        """ref, sup: [N,3,H,W] in [0,1] (device or host) -> flow ref->sup [N,2,H,W] fp32 on the device.
        keep_arena: called INSIDE a sampler call that shares its engine with this network (the `lr_images` guidance term): allocate past
        what the caller holds and release on return instead of rewinding the arena over the caller's buffers (the results are torch-owned)."""
        assert ref.shape == sup.shape and flow_init is None and upsample
        eng = self.engine()
        mark = eng.arena.mark() if keep_arena else None
        if mark is None:
            eng.reset()
        dev = eng.device
        ref, sup = ref.to(dev, torch.float32), sup.to(dev, torch.float32)
        N, _, ht, wd = ref.shape
        pad_ht = (((ht // 8) + 1) * 8 - ht) % 8                      # InputPadder ('sintel'), raft_arch.py:18-34
        pad_wd = (((wd // 8) + 1) * 8 - wd) % 8
        pad = (pad_wd // 2, pad_wd - pad_wd // 2, pad_ht // 2, pad_ht - pad_ht // 2)
        if pad_ht or pad_wd:
            ref, sup = hip.replicate_pad(ref, pad), hip.replicate_pad(sup, pad)
        H, W = ref.shape[-2:]
        H8, W8 = H // 8, W // 8
        if min(H8, W8) < 2 ** (self.corr_levels - 1):
            # the reference fails here too (F.avg_pool2d: "Output size is too small" while building the 4-level pyramid)
            raise ValueError(f"RAFT needs frames of at least {8 * 2 ** (self.corr_levels - 1)} px per side, got {ht}x{wd}")
        hw, M = H8 * W8, N * H8 * W8

        # feature / context encoders (fp32 NHWC, RGB padded to 4 columns)
        def nhwc(t):
            a = eng.act(t.shape[0], H, W, 4, torch.float32)
            hip.nchw_to_nhwc_f32(t, a.v)
            eng.launches += 1
            return a
        pk = lambda tag, conv: eng.weight(tag, (conv.weight,), pack_conv_f32, torch.float32)
        fm = self.fnet.run(eng, nhwc(torch.cat([ref, sup], 0)))
        fmap = _conv(eng, fm, pk("c32", self.fnet.conv2), eng.f32("b", self.fnet.conv2.bias), 256)                 # [2M, 256]
        cm = self.cnet.run(eng, nhwc(ref))
        k2 = self.cnet.conv2
        f32 = torch.float32
        hx = eng.arena.alloc((M, 384), f32)          # cat([net(128), inp(128), motion(126), flow(2)])
        rhx = eng.arena.alloc((M, 384), f32)
        act_of = lambda v: Act(v, N, H8, W8)
        wk, bk = pk("c32", k2), eng.f32("b", k2.bias)
        _conv(eng, cm, wk[:128], bk[:128], 128, out=act_of(hx[:, 0:128]), act=hip.ACT_TANH)       # net = tanh(.)   (:757-759)
        _conv(eng, cm, wk[128:], bk[128:], 128, out=act_of(hx[:, 128:256]), act=hip.ACT_RELU)     # inp = relu(.)

        # all-pairs correlation and its pyramid (:78-86, :47-51): per pair fmap1 [hw,256] . fmap2^T / sqrt(256)
        corr = torch.empty(N, hw, hw, dtype=f32, device=dev)
        hip.conv_f32(fmap.v[:M], fmap.v[M:], corr.view(N * hw, hw), 1, H8, W8, 256, alpha=1.0 / 16.0, batch=N, strideA=hw * 256,
                     strideW=hw * 256, strideC=hw * hw, n_out=hw)
        eng.launches += 1
        levels = [corr.view(N * hw, H8, W8)]
        for _ in range(self.corr_levels - 1):
            levels.append(hip.avgpool2(levels[-1]))

        ys, xs = torch.meshgrid(torch.arange(H8), torch.arange(W8), indexing="ij")
        coords0 = torch.stack([xs, ys], 0).float()[None].repeat(N, 1, 1, 1).to(dev).contiguous()      # (:711-718)
        coords1 = coords0.clone()
        flow = torch.empty_like(coords0)
        ub = self.update_block
        enc, gru = ub.encoder, ub.gru
        D = self.corr_levels * (2 * self.corr_radius + 1) ** 2                                          # 324
        cfeat = eng.arena.alloc((M, D), f32)
        fin = eng.arena.alloc((M, 4), f32)                                                               # flow in columns 0, 1
        fin.zero_()
        corflo = eng.arena.alloc((M, 256), f32)
        delta = eng.arena.alloc((M, 4), f32)
        zr = eng.arena.alloc((M, 256), f32)
        q = eng.arena.alloc((M, 128), f32)

        w_c1, w_c2, w_f1, w_f2, w_cv = (pk("c32", c) for c in (enc.convc1, enc.convc2, enc.convf1, enc.convf2, enc.conv))
        gw = {}
        for n_, ks in (("1", (1, 5)), ("2", (5, 1))):
            cz, cr, cq = getattr(gru, "convz" + n_), getattr(gru, "convr" + n_), getattr(gru, "convq" + n_)
            gw[n_] = (eng.weight("zr32", (cz.weight, cr.weight), lambda a, b: pack_conv_f32(torch.cat([a, b], 0)), f32),
                      eng.weight("zrb", (cz.bias, cr.bias), lambda a, b: torch.cat([a, b], 0), f32),
                      pk("c32", cq), eng.f32("b", cq.bias), ks)
        fh, mk = ub.flow_head, ub.mask
        w_h1, w_h2, w_m0, w_m2 = (pk("c32", c) for c in (fh.conv1, fh.conv2, mk[0], mk[2]))
        b_ = lambda conv: eng.f32("b", conv.bias)
        P1 = (1, 1)

        mask = None
        for it in range(iters):
            # flow = coords1 - coords0 (after the previous iteration's update)                        (:768-776)
            hip.flow_update(coords1, coords0, delta[:, :2] if it > 0 else None, flow, mot=hx[:, 382:384], fin=fin[:, :2])
            hip.corr_lookup(levels, coords1, self.corr_radius, cfeat)
            eng.launches += 2
            # motion encoder                                                                          (:436-444)
            cor = _conv(eng, act_of(cfeat), w_c1, b_(enc.convc1), 256, act=hip.ACT_RELU)
            _conv(eng, cor, w_c2, b_(enc.convc2), 192, (3, 3), 1, P1, act=hip.ACT_RELU, out=act_of(corflo[:, 0:192]))
            flo = _conv(eng, act_of(fin), w_f1, b_(enc.convf1), 128, (7, 7), 1, (3, 3), act=hip.ACT_RELU, cin=2)
            _conv(eng, flo, w_f2, b_(enc.convf2), 64, (3, 3), 1, P1, act=hip.ACT_RELU, out=act_of(corflo[:, 192:256]))
            _conv(eng, act_of(corflo), w_cv, b_(enc.conv), 126, (3, 3), 1, P1, act=hip.ACT_RELU, out=act_of(hx[:, 256:382]))
            # SepConvGRU: horizontal then vertical pass                                               (:390-405)
            for n_ in ("1", "2"):
                wzr, bzr, wq, bq, ks = gw[n_]
                pd = (0, 2) if ks == (1, 5) else (2, 0)
                _conv(eng, act_of(hx), wzr, bzr, 256, ks, 1, pd, act=hip.ACT_SIGMOID, out=act_of(zr))
                hip.gru_rh(zr[:, 128:256], hx, rhx, 128)
                _conv(eng, act_of(rhx), wq, bq, 128, ks, 1, pd, act=hip.ACT_TANH, out=act_of(q))
                hip.gru_gate(zr[:, 0:128], q, hx[:, 0:128])
                eng.launches += 2
            net = act_of(hx[:, 0:128])
            # flow head and, on the last iteration only, the upsampling mask                           (:481-485)
            d1 = _conv(eng, net, w_h1, b_(fh.conv1), 256, (3, 3), 1, P1, act=hip.ACT_RELU)
            _conv(eng, d1, w_h2, b_(fh.conv2), 2, (3, 3), 1, P1, out=act_of(delta[:, :2]))
            if it == iters - 1:
                m1 = _conv(eng, net, w_m0, b_(mk[0]), 256, (3, 3), 1, P1, act=hip.ACT_RELU)
                mask = _conv(eng, m1, w_m2, b_(mk[2]), 576, alpha=0.25)      # .25 * mask(net): alpha * (acc + bias)
        hip.flow_update(coords1, coords0, delta[:, :2], flow)
        up = hip.convex_upsample(flow, mask.v)
        eng.launches += 2
        if mark is not None:
            eng.arena.release(mark)
        if pad_ht or pad_wd:                                                                             # unpad (:30-33)
            out = torch.empty(N, 2, ht, wd, dtype=torch.float32, device=dev)
            hip.crop(up, out, pad[2], pad[0])
            return out
        return up


def compute_flow(flownet, lrs, iters=10, keep_arena=False):
    """ddpm.py:3404-3429 with both directions in ONE batch: lrs [n,t,3,h,w] in [0,1] -> (flows_forward, flows_backward)."""
    n, t, c, h, w = lrs.shape
    a, b = lrs[:, :-1].reshape(-1, c, h, w), lrs[:, 1:].reshape(-1, c, h, w)
    kw = {"keep_arena": True} if keep_arena else {}       # (a stand-in flow network of a test takes the reference's argument list)
    out = flownet(torch.cat([a, b], 0), torch.cat([b, a], 0), iters=iters, **kw)
    k = a.shape[0]
    return out[k:].view(n, t - 1, 2, h, w), out[:k].view(n, t - 1, 2, h, w)
