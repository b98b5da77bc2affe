"""RAFT_SR flow estimator on the MI355X kernels (reference basicsr/archs/raft_arch.py:668-807).

`compute_flow` (ddpm.py:3404-3429) runs this network on the quarter-resolution LR frames right before the hot path; with it
the pipeline no longer needs precomputed flows.  The nn.Modules only own parameters under the reference's state_dict keys
('normal' model: fnet = BasicEncoder(256, instance), cnet = BasicEncoder(256, batch), BasicUpdateBlock with SepConvGRU);
`forward` emits C-ABI launches: every convolution is an `mgld_igemm` (3x3 -> the DMA fast path where Cin % 64 == 0, the
7x7 / 1x5 / 5x1 / strided 1x1 kernels -> the general-tap path), InstanceNorm = the GroupNorm kernels with groups == C,
BatchNorm (eval) is folded into the preceding convolution when the weights are packed, the all-pairs correlation is a
batched NT GEMM with fp32 output, and the pyramid lookup / GRU gates / convex upsampling are the kernels of raft.hip.
Activations are fp16 NHWC, flow / coordinates / correlation volume fp32.  No CPU fallback.
"""
import torch
import torch.nn as nn

from . import hip
from .engine import Act, pack_conv, pack_conv1x1, pack_conv3x3

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
    def _conv_norm(self, eng, x, conv, norm, ksize, stride, pad, relu):
        """conv -> norm -> (relu).  batch: folded, one launch; instance: conv, stats, apply."""
        kh, kw = ksize
        is3 = (kh, kw) == (3, 3)
        if self.norm_fn == "batch":
            def pk(w, b, g, beta, mean, var):
                wf, bf = _fold_bn(w, b, g, beta, mean, var)
                return (pack_conv3x3(wf, x.C) if is3 else pack_conv(wf, x.C)), bf
            wp, bp = eng.weight("cbn", (conv.weight, conv.bias, norm.weight, norm.bias, norm.running_mean, norm.running_var), pk)
            act = hip.ACT_RELU if relu else hip.ACT_NONE
            if is3:
                return eng.conv3x3(x, wp, bp, conv.out_channels, stride=stride, act=act)
            return eng.conv2d(x, wp, bp, conv.out_channels, ksize, stride, pad, act=act)
        wp = eng.weight("c", (conv.weight,), lambda w: pack_conv3x3(w, x.C) if is3 else pack_conv(w, x.C))
        b = eng.f32("b", conv.bias)
        y = eng.conv3x3(x, wp, b, conv.out_channels, stride=stride) if is3 else eng.conv2d(x, wp, b, conv.out_channels, ksize,
                                                                                       stride, pad)
        C = conv.out_channels
        ones, zeros = _affine_identity(eng, C)
        return eng.gn_apply(y, eng.gn_stats(y, _EPS, groups=C), ones, zeros, 2 if relu else 0, groups=C)

    def _block(self, eng, blk, x):
        y = self._conv_norm(eng, x, blk.conv1, blk.norm1, (3, 3), blk.stride, (1, 1), True)
        y = self._conv_norm(eng, y, blk.conv2, blk.norm2, (3, 3), 1, (1, 1), True)
        if blk.downsample is not None:
            x = self._conv_norm(eng, x, blk.downsample[0], blk.downsample[1], (1, 1), blk.stride, (0, 0), False)
        out = eng.act(y.n, y.h, y.w, y.C)
        hip.add_relu(x.v, y.v, out.v)
        eng.launches += 1
        return out

    def run(self, eng, x):
        """x: Act [n,h,w,8] (RGB zero-padded to 8 channels) -> Act [n,h/8,w/8,output_dim] (caller applies tanh/relu splits)"""
        h = self._conv_norm(eng, x, self.conv1, self.norm1, (7, 7), 2, (3, 3), True)
        for layer in (self.layer1, self.layer2, self.layer3):
            for blk in layer:
                h = self._block(eng, blk, h)
        return h


def _affine_identity(eng, C):
    return eng.const(("in_affine", C), lambda: (torch.ones(C, device=eng.device), torch.zeros(C, device=eng.device)))


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
    def forward(self, ref, sup, iters=10, flow_init=None, upsample=True):
        """ref, sup: [N,3,H,W] in [0,1] (device or host) -> flow ref->sup [N,2,H,W] fp32 on the device."""
        assert ref.shape == sup.shape and flow_init is None and upsample
        eng = self.engine()
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

        # feature / context encoders
        fm = self.fnet.run(eng, eng.from_nchw(torch.cat([ref, sup], 0), 8))
        c2 = self.fnet.conv2
        fmap = eng.linear(fm, eng.weight("c1", (c2.weight,), pack_conv1x1), eng.f32("b", c2.bias))          # [2M, 256]
        cm = self.cnet.run(eng, eng.from_nchw(ref, 8))
        k2 = self.cnet.conv2
        hx = eng.arena.alloc((M, 384), torch.float16)          # cat([net(128), inp(128), motion(128)])
        rhx = eng.arena.alloc((M, 384), torch.float16)
        wk, bk = eng.weight("c1", (k2.weight,), pack_conv1x1), eng.f32("b", k2.bias)
        eng.linear(cm.v, wk[:128], bk[:128], out=hx[:, 0:128], act=hip.ACT_TANH)      # net = tanh(.)   (:757-759)
        eng.linear(cm.v, wk[128:], bk[128:], out=hx[:, 128:256], act=hip.ACT_RELU)    # inp = relu(.)

        # all-pairs correlation (fp32) and its pyramid (:78-86, :47-51)
        corr = torch.empty(N, hw, hw, dtype=torch.float32, device=dev)
        hip.igemm(fmap.v[:M], fmap.v[M:], corr.view(N * hw, hw), M=hw, N=hw, K=256, batch=N, strideA=hw * 256,
                  strideW=hw * 256, strideC=hw * hw, alpha=1.0 / 16.0)
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
        cfeat = eng.arena.alloc((M, (D + 7) // 8 * 8), torch.float16)
        cfeat.zero_()                                                                                    # pad columns stay 0
        fin = eng.arena.alloc((M, 8), torch.float16)
        fin.zero_()
        corflo = eng.arena.alloc((M, 256), torch.float16)
        delta = torch.zeros(M, 8, dtype=torch.float32, device=dev)
        zr = eng.arena.alloc((M, 256), torch.float16)
        q = eng.arena.alloc((M, 128), torch.float16)
        act_of = lambda v: Act(v, N, H8, W8)

        w_c1 = eng.weight("c1", (enc.convc1.weight,), lambda w: pack_conv1x1(w, cfeat.shape[1]))
        w_c2 = eng.weight("c3", (enc.convc2.weight,), pack_conv3x3)
        w_f1 = eng.weight("c", (enc.convf1.weight,), lambda w: pack_conv(w, 8))
        w_f2 = eng.weight("c3", (enc.convf2.weight,), pack_conv3x3)
        w_cv = eng.weight("c3", (enc.conv.weight,), pack_conv3x3)
        gw = {}
        for n_, ks in (("1", (1, 5)), ("2", (5, 1))):
            cz, cr, cq = getattr(gru, "convz" + n_), getattr(gru, "convr" + n_), getattr(gru, "convq" + n_)
            gw[n_] = (eng.weight("zr", (cz.weight, cr.weight), lambda a, b: pack_conv(torch.cat([a, b], 0))),
                      eng.weight("zrb", (cz.bias, cr.bias), lambda a, b: torch.cat([a, b], 0), torch.float32),
                      eng.weight("c", (cq.weight,), pack_conv), eng.f32("b", cq.bias), ks)
        fh, mk = ub.flow_head, ub.mask
        w_h1, w_h2 = eng.weight("c3", (fh.conv1.weight,), pack_conv3x3), eng.weight("c3", (fh.conv2.weight,), pack_conv3x3)
        w_m0, w_m2 = eng.weight("c3", (mk[0].weight,), pack_conv3x3), eng.weight("c1", (mk[2].weight,), pack_conv1x1)

        mask = None
        for it in range(iters):
            # flow = coords1 - coords0 (after the previous iteration's update)                        (:768-776)
            hip.flow_update(coords1, coords0, delta[:, :2] if it > 0 else None, flow, mot=hx[:, 382:384], fin=fin[:, :2])
            hip.corr_lookup(levels, coords1, self.corr_radius, cfeat)
            eng.launches += 2
            # motion encoder                                                                          (:436-444)
            cor = eng.linear(cfeat, w_c1, eng.f32("b", enc.convc1.bias), act=hip.ACT_RELU)
            eng.conv3x3(act_of(cor), w_c2, eng.f32("b", enc.convc2.bias), 192, out=act_of(corflo[:, 0:192]), act=hip.ACT_RELU)
            flo = eng.conv2d(act_of(fin), w_f1, eng.f32("b", enc.convf1.bias), 128, (7, 7), 1, (3, 3), act=hip.ACT_RELU)
            eng.conv3x3(flo, w_f2, eng.f32("b", enc.convf2.bias), 64, out=act_of(corflo[:, 192:256]), act=hip.ACT_RELU)
            eng.conv3x3(act_of(corflo), w_cv, eng.f32("b", enc.conv.bias), 126, out=act_of(hx[:, 256:382]), act=hip.ACT_RELU)
            # SepConvGRU: horizontal then vertical pass                                               (:390-405)
            for n_ in ("1", "2"):
                wzr, bzr, wq, bq, ks = gw[n_]
                pd = (0, 2) if ks == (1, 5) else (2, 0)
                eng.conv2d(act_of(hx), wzr, bzr, 256, ks, 1, pd, out=act_of(zr), act=hip.ACT_SIGMOID)
                hip.gru_rh(zr[:, 128:256], hx, rhx, 128)
                eng.conv2d(act_of(rhx), wq, bq, 128, ks, 1, pd, out=act_of(q), act=hip.ACT_TANH)
                hip.gru_gate(zr[:, 0:128], q, hx[:, 0:128])
                eng.launches += 2
            net = act_of(hx[:, 0:128])
            # flow head (fp32 delta) and, on the last iteration only, the upsampling mask              (:481-485)
            d1 = eng.conv3x3(net, w_h1, eng.f32("b", fh.conv1.bias), 256, act=hip.ACT_RELU)
            eng.conv3x3(d1, w_h2, eng.f32("b", fh.conv2.bias), 2, out=Act(delta[:, :2], N, H8, W8), out_dtype=torch.float32)
            if it == iters - 1:
                m1 = eng.conv3x3(net, w_m0, eng.f32("b", mk[0].bias), 256, act=hip.ACT_RELU)
                mask = eng.linear(m1, w_m2, eng.f32("b", mk[2].bias), alpha=0.25)      # .25 * mask(net): alpha * (acc + bias)
        hip.flow_update(coords1, coords0, delta[:, :2], flow)
        up = hip.convex_upsample(flow, mask.v)
        eng.launches += 2
        if pad_ht or pad_wd:                                                                             # unpad (:30-33)
            out = torch.empty(N, 2, ht, wd, dtype=torch.float32, device=dev)
            hip.crop(up, out, pad[2], pad[0])
            return out
        return up


def compute_flow(flownet, lrs, iters=10):
    """ddpm.py:3404-3429 with both directions in ONE batch: lrs [n,t,3,h,w] in [0,1] -> (flows_forward, flows_backward)."""
    n, t, c, h, w = lrs.shape
    a, b = lrs[:, :-1].reshape(-1, c, h, w), lrs[:, 1:].reshape(-1, c, h, w)
    out = flownet(torch.cat([a, b], 0), torch.cat([b, a], 0), iters=iters)
    k = a.shape[0]
    return out[k:].view(n, t - 1, 2, h, w), out[:k].view(n, t - 1, 2, h, w)
