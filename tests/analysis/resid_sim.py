#!/usr/bin/env python
"""What-if study for the fp32 residual stream (CPU, no GPU): the oracle UNet / video decoder re-run with the roundings of the HIP path placed
exactly where the kernels round — operands entering a contraction (I), weights (W), tensors stored between kernels inside a block (H: the
first convolution's output, q / k / v / attention output, GEGLU output, SPADE gamma / beta) and the RESIDUAL STREAM itself (S: every
`x + f(x)` a block hands to the next one, the down / upsample convolutions, the temporal mixes).  Policies: a policy is a set of letters that
ARE rounded to fp16.  "IWHS" is the shipped fp16 path; "IWH" is the fp32 residual stream; "IW" keeps every stored tensor fp32.
Analysis script behind DESIGN.md section 5 (test infrastructure: it drives the oracle; not collected by pytest)."""
import json
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from configs import UNET_SMALL, VAE_DD_SMALL  # noqa: E402
from mgld_vsr_amd import synth  # noqa: E402
from oracle import nets  # noqa: E402

POL = set("IWHS")
R = lambda t: t.half().float()
rI = lambda t: R(t) if "I" in POL else t
rW = lambda t: R(t) if "W" in POL else t
rH = lambda t: R(t) if "H" in POL else t
rS = lambda t: R(t) if "S" in POL else t
_conv, _lin = F.conv2d, F.linear


def conv(x, p, stride=1, padding=0):
    return _conv(rI(x), rW(p["weight"]), p["bias"] if p.has("bias") else None, stride=stride, padding=padding)


def linear(x, p):
    return _lin(rI(x), rW(p["weight"]), p["bias"] if p.has("bias") else None)


gn = nets.gn


def attention_core(q, k, v, heads, scale=None):
    b, nq, c = q.shape
    d = c // heads
    sh = lambda t: rI(t).reshape(b, t.shape[1], heads, d).permute(0, 2, 1, 3)
    q, k, v = sh(q), sh(k), sh(v)
    s = torch.einsum("bhid,bhjd->bhij", q, k) * (scale if scale is not None else d ** -0.5)
    o = torch.einsum("bhij,bhjd->bhid", rI(s.softmax(-1)), v)
    return o.permute(0, 2, 1, 3).reshape(b, nq, c)


def resblock(x, emb, p, struct_cond=None):
    h = rH(conv(F.silu(gn(x, p.sub("in_layers.0"), 1e-5)), p.sub("in_layers.2"), padding=1) +
           linear(F.silu(emb), p.sub("emb_layers.1"))[:, :, None, None])
    h = conv(F.silu(gn(h, p.sub("out_layers.0"), 1e-5)), p.sub("out_layers.3"), padding=1)
    if struct_cond is not None:
        h = rH(h)
        q = p.sub("spade")
        actv = F.relu(conv(struct_cond[str(h.size(-1))], q.sub("mlp_shared.0"), padding=1))
        gamma, beta = rH(conv(actv, q.sub("mlp_gamma"), padding=1)), rH(conv(actv, q.sub("mlp_beta"), padding=1))
        h = gn(h, q.sub("param_free_norm"), 1e-5) * (1 + gamma) + beta
    if p.has("skip_connection.weight"):
        w = p["skip_connection.weight"]
        x = rS(_conv(rI(x), rW(w), p["skip_connection.bias"], padding=w.shape[-1] // 2))
    return rS(x + h)


def cross_attention(x, context, p, heads):
    q = rH(linear(x, p.sub("to_q")))
    ctx = x if context is None else context
    if x.shape[0] != ctx.shape[0]:
        ctx = torch.repeat_interleave(ctx, x.shape[0] // ctx.shape[0], dim=0)
    k, v = rH(linear(ctx, p.sub("to_k"))), rH(linear(ctx, p.sub("to_v")))
    return linear(rH(attention_core(q, k, v, heads)), p.sub("to_out.0"))


def transformer_block(x, context, p, heads):
    ln = lambda t, q: F.layer_norm(t, (t.shape[-1],), q["weight"], q["bias"], 1e-5)
    x = rS(cross_attention(ln(x, p.sub("norm1")), None, p.sub("attn1"), heads) + x)
    x = rS(cross_attention(ln(x, p.sub("norm2")), context, p.sub("attn2"), heads) + x)
    y = linear(ln(x, p.sub("norm3")), p.sub("ff.net.0.proj"))
    a, gate = y.chunk(2, dim=-1)
    return rS(linear(rH(a * F.gelu(gate)), p.sub("ff.net.2")) + x)


def spatial_transformer(x, context, p, heads):
    b, c, h, w = x.shape
    x_in = x
    x = gn(x, p.sub("norm"), 1e-6).permute(0, 2, 3, 1).reshape(b, h * w, c)
    x = rS(linear(x, p.sub("proj_in")))
    x = transformer_block(x, context, p.sub("transformer_blocks.0"), heads)
    x = linear(x, p.sub("proj_out"))
    return rS(x.reshape(b, h, w, c).permute(0, 3, 1, 2) + x_in)


def spatial_temporal_conv(x, p, num_frames):
    bt, c, h, w = x.shape
    b = bt // num_frames
    x5 = rI(x).reshape(b, num_frames, c, h, w).permute(0, 2, 1, 3, 4)
    res = F.conv3d(x5, rW(p["temporal_conv.weight"]), p["temporal_conv.bias"], padding=(1, 0, 0))
    res = res.permute(0, 2, 1, 3, 4).reshape(bt, c, h, w)
    a = p["temporal_alpha"]
    return rS(a * res + (1 - a) * x)


def temporal_attention(x, p, heads, num_frames):
    bt, c, h, w = x.shape
    b = bt // num_frames
    t3 = x.reshape(b, num_frames, c, h, w).permute(0, 3, 4, 1, 2).reshape(b * h * w, num_frames, c)
    n = F.layer_norm(t3, (c,), p["norm.weight"], p["norm.bias"], 1e-5)
    q, k, v = (rH(linear(n, p.sub(f"temporal_attn.to_{s}"))) for s in "qkv")
    res = linear(rH(attention_core(q, k, v, heads)), p.sub("temporal_attn.to_out.0"))
    res = res.reshape(b, h, w, num_frames, c).permute(0, 3, 4, 1, 2).reshape(bt, c, h, w)
    a = p["temporal_alpha"]
    return rS(a * res + (1 - a) * x)


def stream_conv(x, p, stride=1, padding=0):
    return rS(conv(x, p, stride, padding))


def vae_resnet(x, p):
    h = rH(conv(nets.swish(gn(x, p.sub("norm1"), 1e-6)), p.sub("conv1"), padding=1))
    h = conv(nets.swish(gn(h, p.sub("norm2"), 1e-6)), p.sub("conv2"), padding=1)
    if p.has("nin_shortcut.weight"):
        x = rS(conv(x, p.sub("nin_shortcut")))
    return rS(x + h)


def vae_attn(x, p):
    b, c, h, w = x.shape
    hn = gn(x, p.sub("norm"), 1e-6)
    tok = lambda t: t.reshape(b, c, h * w).permute(0, 2, 1)
    q, k, v = (tok(rH(conv(hn, p.sub(s)))) for s in "qkv")
    o = rH(attention_core(q, k, v, 1, scale=int(c) ** (-0.5)))
    return rS(x + conv(o.permute(0, 2, 1).reshape(b, c, h, w), p.sub("proj_out")))


def fuse_resblock(x, p):
    h = rH(conv(nets.swish(gn(x, p.sub("norm1"), 1e-6)), p.sub("conv1"), padding=1))
    h = conv(nets.swish(gn(h, p.sub("norm2"), 1e-6)), p.sub("conv2"), padding=1)
    if p.has("conv_out.weight"):
        x = rS(conv(x, p.sub("conv_out")))
    return rS(h + x)


def rdb(x, p):
    lr = lambda t: F.leaky_relu(t, 0.2)
    x1 = rH(lr(conv(x, p.sub("conv1"), padding=1)))
    x2 = rH(lr(conv(torch.cat((x, x1), 1), p.sub("conv2"), padding=1)))
    x3 = rH(lr(conv(torch.cat((x, x1, x2), 1), p.sub("conv3"), padding=1)))
    x4 = rH(lr(conv(torch.cat((x, x1, x2, x3), 1), p.sub("conv4"), padding=1)))
    x5 = conv(torch.cat((x, x1, x2, x3, x4), 1), p.sub("conv5"), padding=1)
    return rS(x5 * 0.2 + x)


def fuse_block(enc_feat, dec_feat, p, w, num_block):
    e = fuse_resblock(torch.cat([enc_feat, dec_feat], dim=1), p.sub("encode_enc_1"))
    for i in range(num_block):
        e = rdb(e, p.sub(f"encode_enc_2.{i}"))
    e = fuse_resblock(e, p.sub("encode_enc_3"))
    return rS(dec_feat + w * e)


def install():
    for k in ("resblock", "cross_attention", "transformer_block", "spatial_transformer", "spatial_temporal_conv", "temporal_attention",
              "attention_core", "vae_resnet", "vae_attn", "fuse_resblock", "rdb", "fuse_block", "linear"):
        setattr(nets, k, globals()[k])
    nets.conv = stream_conv          # what unet_forward / vae_decode call directly: stem, down / upsample, output convolutions


def rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm())


def G(name):
    d = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
    return {k: (torch.from_numpy(d[k]) if d[k].dtype.kind in "fiu" else d[k]) for k in d.files}


def main():
    install()
    torch.set_num_threads(8)
    g = G("g_unet")
    usd = synth.synth_state_dict(json.loads(str(g["unet_params"])), "unet")
    sc = {k[3:]: v for k, v in g.items() if k.startswith("sc_")}
    gv = G("g_vae")
    vsd = synth.synth_state_dict(json.loads(str(gv["vae_params"])), "vae")
    for pol in sys.argv[1:] or ["", "IWHS", "IWH", "IWS", "IW", "WHS", "IHS", "I", "W", "H", "S"]:
        POL.clear()
        POL.update(pol)
        with torch.no_grad():
            eps = nets.unet_forward(usd, UNET_SMALL, g["x"], g["t"], g["ctx"], sc)
            dec = nets.vae_decode(vsd, VAE_DD_SMALL, gv["z"], [gv["fea0"], gv["fea1"]], fusion_w=1.0)
            dec05 = nets.vae_decode(vsd, VAE_DD_SMALL, gv["z"], [gv["fea0"], gv["fea1"]], fusion_w=0.5)
        print(f"{pol or 'fp32':6s} unet {rel(eps, g['eps']):.2e}   vae dec {rel(dec, gv['dec']):.2e}   dec_w05 {rel(dec05, gv['dec_w05']):.2e}", flush=True)


if __name__ == "__main__":
    main()
