"""TEST INFRASTRUCTURE — CPU fp32 restatement of the reference's RAFT_SR flow estimator ('normal' model) as pure functions
over its state_dict (basicsr/archs/raft_arch.py; every function cites the lines it follows).  Pinned to the reference's own
module by tests/golden/g_raft.npz (tests/test_oracle_golden.py).  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this package; the product path never does."""
import torch
import torch.nn.functional as F


def _conv(sd, name, x, stride=1, padding=0):
    return F.conv2d(x, sd[name + ".weight"], sd.get(name + ".bias"), stride=stride, padding=padding)


def _norm(sd, name, x, kind):
    """raft_arch.py:100-124 — 'instance': nn.InstanceNorm2d (no affine, eps 1e-5); 'batch': nn.BatchNorm2d in eval mode"""
    if kind == "instance":
        return F.instance_norm(x, eps=1e-5)
    if kind == "batch":
        return F.batch_norm(x, sd[name + ".running_mean"], sd[name + ".running_var"], sd[name + ".weight"], sd[name + ".bias"],
                            training=False, eps=1e-5)
    raise ValueError(kind)


def _resblock(sd, pre, x, kind, stride):
    """ResidualBlock.forward, raft_arch.py:130-138"""
    y = F.relu(_norm(sd, pre + ".norm1", _conv(sd, pre + ".conv1", x, stride, 1), kind))
    y = F.relu(_norm(sd, pre + ".norm2", _conv(sd, pre + ".conv2", y, 1, 1), kind))
    if stride != 1:
        # downsample = Sequential(conv1x1 stride, norm3): ONE module registered twice (`norm3.*` and `downsample.1.*` hold
        # the same tensors in a real checkpoint; load_state_dict applies `downsample.1.*` last, so that name is read here)
        x = _norm(sd, pre + ".downsample.1", _conv(sd, pre + ".downsample.0", x, stride, 0), kind)
    return F.relu(x + y)


def encoder(sd, pre, x, kind):
    """BasicEncoder.forward, raft_arch.py:248-272 (eval mode: no dropout)"""
    x = F.relu(_norm(sd, pre + ".norm1", _conv(sd, pre + ".conv1", x, 2, 3), kind))
    for layer, stride in (("layer1", 1), ("layer2", 2), ("layer3", 2)):
        x = _resblock(sd, f"{pre}.{layer}.0", x, kind, stride)
        x = _resblock(sd, f"{pre}.{layer}.1", x, kind, 1)
    return _conv(sd, pre + ".conv2", x)


def coords_grid(batch, ht, wd):
    """raft_arch.py:536-539 — channel 0 = x, channel 1 = y"""
    ys, xs = torch.meshgrid(torch.arange(ht), torch.arange(wd), indexing="ij")
    return torch.stack([xs, ys], 0).float()[None].repeat(batch, 1, 1, 1)


def bilinear_sampler(img, coords):
    """raft_arch.py:519-533"""
    H, W = img.shape[-2:]
    xgrid, ygrid = coords.split([1, 1], dim=-1)
    xgrid = 2 * xgrid / (W - 1) - 1
    ygrid = 2 * ygrid / (H - 1) - 1
    return F.grid_sample(img, torch.cat([xgrid, ygrid], dim=-1), align_corners=True)


class CorrBlock:
    """raft_arch.py:37-86"""

    def __init__(self, fmap1, fmap2, num_levels=4, radius=4):
        self.num_levels, self.radius = num_levels, radius
        batch, dim, ht, wd = fmap1.shape
        corr = torch.matmul(fmap1.view(batch, dim, ht * wd).transpose(1, 2), fmap2.view(batch, dim, ht * wd))
        corr = corr.view(batch, ht, wd, 1, ht, wd) / torch.sqrt(torch.tensor(dim).float())
        corr = corr.reshape(batch * ht * wd, 1, ht, wd)
        self.pyramid = [corr]
        for _ in range(num_levels - 1):
            corr = F.avg_pool2d(corr, 2, stride=2)
            self.pyramid.append(corr)

    def __call__(self, coords):
        r = self.radius
        coords = coords.permute(0, 2, 3, 1)
        batch, h1, w1, _ = coords.shape
        out = []
        for i in range(self.num_levels):
            d = torch.linspace(-r, r, 2 * r + 1)
            delta = torch.stack(torch.meshgrid(d, d, indexing="ij"), dim=-1)          # meshgrid(dy, dx) added to (x, y)
            centroid = coords.reshape(batch * h1 * w1, 1, 1, 2) / 2 ** i
            c = bilinear_sampler(self.pyramid[i], centroid + delta.view(1, 2 * r + 1, 2 * r + 1, 2))
            out.append(c.view(batch, h1, w1, -1))
        return torch.cat(out, dim=-1).permute(0, 3, 1, 2).contiguous().float()


def motion_encoder(sd, pre, flow, corr):
    """BasicMotionEncoder.forward, raft_arch.py:436-444"""
    cor = F.relu(_conv(sd, pre + ".convc1", corr))
    cor = F.relu(_conv(sd, pre + ".convc2", cor, 1, 1))
    flo = F.relu(_conv(sd, pre + ".convf1", flow, 1, 3))
    flo = F.relu(_conv(sd, pre + ".convf2", flo, 1, 1))
    out = F.relu(_conv(sd, pre + ".conv", torch.cat([cor, flo], 1), 1, 1))
    return torch.cat([out, flow], 1)


def sep_conv_gru(sd, pre, h, x):
    """SepConvGRU.forward, raft_arch.py:390-405"""
    for sfx, pad in (("1", (0, 2)), ("2", (2, 0))):
        hx = torch.cat([h, x], 1)
        z = torch.sigmoid(_conv(sd, f"{pre}.convz{sfx}", hx, 1, pad))
        r = torch.sigmoid(_conv(sd, f"{pre}.convr{sfx}", hx, 1, pad))
        q = torch.tanh(_conv(sd, f"{pre}.convq{sfx}", torch.cat([r * h, x], 1), 1, pad))
        h = (1 - z) * h + z * q
    return h


def update_block(sd, pre, net, inp, corr, flow):
    """BasicUpdateBlock.forward, raft_arch.py:475-485"""
    mot = motion_encoder(sd, pre + ".encoder", flow, corr)
    net = sep_conv_gru(sd, pre + ".gru", net, torch.cat([inp, mot], 1))
    delta = _conv(sd, pre + ".flow_head.conv2", F.relu(_conv(sd, pre + ".flow_head.conv1", net, 1, 1)), 1, 1)
    mask = 0.25 * _conv(sd, pre + ".mask.2", F.relu(_conv(sd, pre + ".mask.0", net, 1, 1)))
    return net, mask, delta


def upsample_flow(flow, mask):
    """RAFT_SR.upsample_flow, raft_arch.py:720-731"""
    N, _, H, W = flow.shape
    mask = torch.softmax(mask.view(N, 1, 9, 8, 8, H, W), dim=2)
    up = F.unfold(8 * flow, [3, 3], padding=1).view(N, 2, 9, 1, 1, H, W)
    up = torch.sum(mask * up, dim=2).permute(0, 1, 4, 2, 5, 3)
    return up.reshape(N, 2, 8 * H, 8 * W)


def raft_sr(sd, ref, sup, iters=10):
    """RAFT_SR.forward -> process, raft_arch.py:733-806 ('normal' model: hidden 128, context 128, 4 levels, radius 4).
    ref / sup: [N,3,H,W]; returns flow [N,2,H,W]."""
    ht, wd = ref.shape[-2:]
    pad_ht = (((ht // 8) + 1) * 8 - ht) % 8                                   # InputPadder, :18-34 ('sintel' mode)
    pad_wd = (((wd // 8) + 1) * 8 - wd) % 8
    pad = [pad_wd // 2, pad_wd - pad_wd // 2, pad_ht // 2, pad_ht - pad_ht // 2]
    ref, sup = F.pad(ref, pad, mode="replicate"), F.pad(sup, pad, mode="replicate")
    f = encoder(sd, "fnet", torch.cat([ref, sup], 0), "instance").float()
    fmap1, fmap2 = f[:ref.shape[0]], f[ref.shape[0]:]
    corr_fn = CorrBlock(fmap1, fmap2, radius=4)
    cnet = encoder(sd, "cnet", ref, "batch")
    net, inp = torch.split(cnet, [128, 128], dim=1)
    net, inp = torch.tanh(net), torch.relu(inp)
    N, _, H, W = ref.shape
    coords0, coords1 = coords_grid(N, H // 8, W // 8), coords_grid(N, H // 8, W // 8)
    flow_up = None
    for _ in range(iters):
        corr = corr_fn(coords1)
        net, mask, delta = update_block(sd, "update_block", net, inp, corr, coords1 - coords0)
        coords1 = coords1 + delta
        flow_up = upsample_flow(coords1 - coords0, mask)
    hh, ww = flow_up.shape[-2:]
    return flow_up[..., pad[2]:hh - pad[3], pad[0]:ww - pad[1]]


def compute_flow(sd, lrs, iters=10):
    """LatentDiffusionVSRTextWT.compute_flow, ddpm.py:3404-3429: lrs [n,t,3,h,w] in [0,1] ->
    (flows_forward, flows_backward), each [n,t-1,2,h,w]."""
    n, t, c, h, w = lrs.shape
    lrs_1 = lrs[:, :-1].reshape(-1, c, h, w)
    lrs_2 = lrs[:, 1:].reshape(-1, c, h, w)
    flows_backward = raft_sr(sd, lrs_1, lrs_2, iters).view(n, t - 1, 2, h, w)
    flows_forward = raft_sr(sd, lrs_2, lrs_1, iters).view(n, t - 1, 2, h, w)
    return flows_forward, flows_backward
