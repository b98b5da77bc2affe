"""Oracle (test infrastructure): functional torch-CPU fp32 restatement of the hot-path networks.

Every function takes the reference's own state_dict (same key names) and follows the cited reference code:
  UNet            ldm/modules/diffusionmodules/openaimodel.py:1903-2313 (InflatedUNetModelDualcondV2)
  struct-cond     openaimodel.py:2316-2525 (InflatedEncoderUNetModelWT)
  blocks          openaimodel.py:160-188,204-230,233-359,362-482,485-531,554-594
  transformer     ldm/modules/attention.py:48-75,124-143,262-381,406-435,484-546 (xformers branch semantics:
                  softmax(q k^T d^-1/2) v, context broadcast over frames)
  SPADE           ldm/modules/spade.py:68-111
  temporal conv   ldm/modules/diffusionmodules/util.py:291-310
  VAE             ldm/modules/diffusionmodules/model.py:84-183,192-244,473-572,926-1056,1312-1367;
                  basicsr/archs/rrdbnet_arch.py:9-38; ldm/models/autoencoder.py:1674-1690
No weights are owned here: `sd` maps reference key -> tensor.
"""
import math

import torch
import torch.nn.functional as F


class SD:
    """state-dict view with a key prefix."""

    def __init__(self, sd, prefix=""):
        self.sd, self.prefix = sd, prefix

    def sub(self, name):
        return SD(self.sd, f"{self.prefix}{name}.")

    def __getitem__(self, k):
        return self.sd[self.prefix + k]

    def has(self, k):
        return (self.prefix + k) in self.sd


def timestep_embedding(timesteps, dim, max_period=10000):
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(0, half, dtype=torch.float32) / half)
    args = timesteps[:, None].float() * freqs[None]
    emb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
    if dim % 2:
        emb = torch.cat([emb, torch.zeros_like(emb[:, :1])], dim=-1)
    return emb


def gn(x, p, eps, groups=32):
    return F.group_norm(x.float(), groups, p["weight"], p["bias"], eps)


def conv(x, p, stride=1, padding=0):
    return F.conv2d(x, p["weight"], p["bias"] if p.has("bias") else None, stride=stride, padding=padding)


def linear(x, p):
    return F.linear(x, p["weight"], p["bias"] if p.has("bias") else None)


def attention_core(q, k, v, heads, scale=None):
    """xformers.ops.memory_efficient_attention semantics on [b, n, heads*d] tensors."""
    b, nq, c = q.shape
    d = c // heads
    sh = lambda t: t.reshape(b, t.shape[1], heads, d).permute(0, 2, 1, 3)
    q, k, v = sh(q), sh(k), sh(v)
    s = torch.einsum("bhid,bhjd->bhij", q, k) * (scale if scale is not None else d ** -0.5)
    o = torch.einsum("bhij,bhjd->bhid", s.softmax(-1), v)
    return o.permute(0, 2, 1, 3).reshape(b, nq, c)


# ---- UNet blocks ---------------------------------------------------------------------------------------------------
def spade(x, segmap, p):
    """spade.py:93-111"""
    normalized = gn(x, p.sub("param_free_norm"), 1e-5)
    actv = F.relu(conv(segmap, p.sub("mlp_shared.0"), padding=1))
    gamma = conv(actv, p.sub("mlp_gamma"), padding=1)
    beta = conv(actv, p.sub("mlp_beta"), padding=1)
    return normalized * (1 + gamma) + beta


def resblock(x, emb, p, struct_cond=None):
    """ResBlock (openaimodel.py:329-359) / ResBlockDual (:454-482, SPADE when struct_cond is given)."""
    h = conv(F.silu(gn(x, p.sub("in_layers.0"), 1e-5)), p.sub("in_layers.2"), padding=1)
    emb_out = linear(F.silu(emb), p.sub("emb_layers.1"))
    h = h + emb_out[:, :, None, None]
    h = conv(F.silu(gn(h, p.sub("out_layers.0"), 1e-5)), p.sub("out_layers.3"), padding=1)
    if struct_cond is not None:
        h = spade(h, struct_cond[str(h.size(-1))], p.sub("spade"))
    if p.has("skip_connection.weight"):
        w = p["skip_connection.weight"]
        x = F.conv2d(x, w, p["skip_connection.bias"], padding=w.shape[-1] // 2)
    return x + h


def attention_block(x, p, heads):
    """AttentionBlock + QKVAttentionLegacy (openaimodel.py:525-531, 564-594)."""
    b, c = x.shape[:2]
    xf = x.reshape(b, c, -1)
    qkv = F.conv1d(gn(xf, p.sub("norm"), 1e-5), p["qkv.weight"], p["qkv.bias"])
    length = qkv.shape[-1]
    ch = c // heads
    q, k, v = qkv.reshape(b * heads, ch * 3, length).split(ch, dim=1)
    w = torch.einsum("bct,bcs->bts", q, k) * (ch ** -0.5)
    a = torch.einsum("bts,bcs->bct", w.softmax(-1), v).reshape(b, -1, length)
    h = F.conv1d(a, p["proj_out.weight"], p["proj_out.bias"])
    return (xf + h).reshape(x.shape)


def cross_attention(x, context, p, heads):
    """MemoryEfficientCrossAttention (attention.py:333-381)."""
    q = linear(x, p.sub("to_q"))
    ctx = x if context is None else context
    if x.shape[0] != ctx.shape[0]:
        ctx = torch.repeat_interleave(ctx, x.shape[0] // ctx.shape[0], dim=0)
    k, v = linear(ctx, p.sub("to_k")), linear(ctx, p.sub("to_v"))
    return linear(attention_core(q, k, v, heads), p.sub("to_out.0"))


def transformer_block(x, context, p, heads):
    """BasicTransformerBlockV2._forward (attention.py:431-435), GEGLU feed-forward (:48-75)."""
    ln = lambda t, q: F.layer_norm(t, (t.shape[-1],), q["weight"], q["bias"], 1e-5)
    x = cross_attention(ln(x, p.sub("norm1")), None, p.sub("attn1"), heads) + x
    x = cross_attention(ln(x, p.sub("norm2")), context, p.sub("attn2"), heads) + x
    y = linear(ln(x, p.sub("norm3")), p.sub("ff.net.0.proj"))
    a, gate = y.chunk(2, dim=-1)
    return linear(a * F.gelu(gate), p.sub("ff.net.2")) + x


def spatial_transformer(x, context, p, heads):
    """SpatialTransformerV2.forward with use_linear=True, depth 1 (attention.py:527-546)."""
    b, c, h, w = x.shape
    x_in = x
    x = gn(x, p.sub("norm"), 1e-6)
    x = x.permute(0, 2, 3, 1).reshape(b, h * w, c)
    x = linear(x, p.sub("proj_in"))
    x = transformer_block(x, context, p.sub("transformer_blocks.0"), heads)
    x = linear(x, p.sub("proj_out"))
    return x.reshape(b, h, w, c).permute(0, 3, 1, 2) + x_in


def spatial_temporal_conv(x, p, num_frames):
    """SpatialTemporalConv.forward (util.py:301-310)."""
    bt, c, h, w = x.shape
    b = bt // num_frames
    x5 = x.reshape(b, num_frames, c, h, w).permute(0, 2, 1, 3, 4)
    res = F.conv3d(x5, p["temporal_conv.weight"], p["temporal_conv.bias"], padding=(1, 0, 0))
    res = res.permute(0, 2, 1, 3, 4).reshape(bt, c, h, w)
    a = p["temporal_alpha"]
    return a * res + (1 - a) * x


def temporal_attention(x, p, heads, num_frames):
    """TemporalAttention.forward (attention.py:135-143)."""
    bt, c, h, w = x.shape
    b = bt // num_frames
    t3 = x.reshape(b, num_frames, c, h, w).permute(0, 3, 4, 1, 2).reshape(b * h * w, num_frames, c)
    n = F.layer_norm(t3, (c,), p["norm.weight"], p["norm.bias"], 1e-5)
    q, k, v = (linear(n, p.sub(f"temporal_attn.to_{s}")) for s in "qkv")
    res = linear(attention_core(q, k, v, heads), p.sub("temporal_attn.to_out.0"))
    res = res.reshape(b, h, w, num_frames, c).permute(0, 3, 4, 1, 2).reshape(bt, c, h, w)
    a = p["temporal_alpha"]
    return a * res + (1 - a) * x


def unet_layout(cfg):
    """Block structure of InflatedUNetModelDualcondV2.__init__ (openaimodel.py:2033-2260): returns lists of
    per-block layer kinds for input_blocks / output_blocks and per-level head counts."""
    mc, mult, nrb = cfg["model_channels"], cfg["channel_mult"], cfg["num_res_blocks"]
    attn_res, nhc = cfg["attention_resolutions"], cfg["num_head_channels"]
    inp = [["conv"]]
    chans = [mc]
    ch, ds = mc, 1
    for level, m in enumerate(mult):
        for _ in range(nrb):
            layers = [("res", ch, m * mc)]
            ch = m * mc
            if ds in attn_res:
                layers.append(("st", ch, ch // nhc))
            inp.append(layers)
            chans.append(ch)
        if level != len(mult) - 1:
            inp.append([("down", ch)])
            chans.append(ch)
            ds *= 2
    mid_ch, mid_heads = ch, ch // nhc
    out = []
    for level, m in list(enumerate(mult))[::-1]:
        for i in range(nrb + 1):
            ich = chans.pop()
            layers = [("res", ch + ich, mc * m)]
            ch = mc * m
            if ds in attn_res:
                layers.append(("st", ch, ch // nhc))
            if level and i == nrb:
                layers.append(("up", ch))
                ds //= 2
            out.append(layers)
    return inp, (mid_ch, mid_heads), out


def unet_forward(sd, cfg, x, timesteps, context, struct_cond):
    """InflatedUNetModelDualcondV2.forward (openaimodel.py:2281-2313)."""
    p = SD(sd)
    T = cfg["num_frames"]
    inp, (mid_ch, mid_heads), outb = unet_layout(cfg)
    emb = timestep_embedding(timesteps, cfg["model_channels"])
    emb = linear(F.silu(linear(emb, p.sub("time_embed.0"))), p.sub("time_embed.2"))

    def run(h, layers, q):
        for j, l in enumerate(layers):
            lp = q.sub(str(j))
            if l == "conv":
                h = conv(h, lp, padding=1)
            elif l[0] == "res":
                h = resblock(h, emb, lp, struct_cond)
            elif l[0] == "st":
                h = spatial_transformer(h, context, lp, l[2])
            elif l[0] == "down":
                h = conv(h, lp.sub("op"), stride=2, padding=1)
            elif l[0] == "up":
                h = conv(F.interpolate(h, scale_factor=2, mode="nearest"), lp.sub("conv"), padding=1)
        return h

    hs = []
    h = x.float()
    for i, layers in enumerate(inp):
        h = run(h, layers, p.sub(f"input_blocks.{i}"))
        hs.append(h)
    mb = p.sub("middle_block")
    h = resblock(h, emb, mb.sub("0"), struct_cond)
    h = spatial_temporal_conv(h, mb.sub("1"), T)
    h = spatial_transformer(h, context, mb.sub("2"), mid_heads)
    h = temporal_attention(h, mb.sub("3"), mid_heads, T)
    h = resblock(h, emb, mb.sub("4"), struct_cond)
    h = spatial_temporal_conv(h, mb.sub("5"), T)
    for i, layers in enumerate(outb):
        h = torch.cat([h, hs.pop()], dim=1)
        h = run(h, layers, p.sub(f"output_blocks.{i}"))
    return conv(F.silu(gn(h, p.sub("out.0"), 1e-5)), p.sub("out.2"), padding=1)


def structcond_layout(cfg):
    mc, mult, nrb = cfg["model_channels"], cfg["channel_mult"], cfg["num_res_blocks"]
    attn_res = cfg["attention_resolutions"]
    inp = [["conv"]]
    ch, ds = mc, 1
    for level, m in enumerate(mult):
        for _ in range(nrb):
            layers = [("res", ch, m * mc)]
            ch = m * mc
            if ds in attn_res:
                layers.append(("attn", ch))
            inp.append(layers)
        if level != len(mult) - 1:
            inp.append([("down", ch)])
            ds *= 2
    return inp, ch


def structcond_heads(cfg, ch):
    return cfg["num_heads"] if cfg.get("num_head_channels", -1) == -1 else ch // cfg["num_head_channels"]


def structcond_forward(sd, cfg, x, timesteps):
    """InflatedEncoderUNetModelWT.forward (openaimodel.py:2500-2525) -> dict str(width) -> [T, out_channels, r, r]."""
    p = SD(sd)
    inp, mid_ch = structcond_layout(cfg)
    emb = timestep_embedding(timesteps, cfg["model_channels"])
    emb = linear(F.silu(linear(emb, p.sub("time_embed.0"))), p.sub("time_embed.2"))
    results = []
    h = x.float()
    for i, layers in enumerate(inp):
        last = h
        q = p.sub(f"input_blocks.{i}")
        for j, l in enumerate(layers):
            lp = q.sub(str(j))
            if l == "conv":
                h = conv(h, lp, padding=1)
            elif l[0] == "res":
                h = resblock(h, emb, lp)
            elif l[0] == "attn":
                h = attention_block(h, lp, structcond_heads(cfg, l[1]))
            elif l[0] == "down":
                h = conv(h, lp.sub("op"), stride=2, padding=1)
        if h.size(-1) != last.size(-1):
            results.append(last)
    mb = p.sub("middle_block")
    h = resblock(h, emb, mb.sub("0"))
    h = attention_block(h, mb.sub("1"), structcond_heads(cfg, mid_ch))
    h = resblock(h, emb, mb.sub("2"))
    results.append(h)
    return {str(r.size(-1)): resblock(r, emb, p.sub(f"fea_tran.{i}")) for i, r in enumerate(results)}


# ---- VAE -------------------------------------------------------------------------------------------------------------
def swish(x):
    return x * torch.sigmoid(x)


def vae_resnet(x, p):
    """ResnetBlock.forward with temb=None (model.py:160-183)."""
    h = conv(swish(gn(x, p.sub("norm1"), 1e-6)), p.sub("conv1"), padding=1)
    h = conv(swish(gn(h, p.sub("norm2"), 1e-6)), p.sub("conv2"), padding=1)
    if p.has("nin_shortcut.weight"):
        x = conv(x, p.sub("nin_shortcut"))
    return x + h


def vae_attn(x, p):
    """AttnBlock / MemoryEfficientAttnBlock (model.py:220-244,276-304): 1 head, scale c^-1/2."""
    b, c, h, w = x.shape
    hn = gn(x, p.sub("norm"), 1e-6)
    tok = lambda t: t.reshape(b, c, h * w).permute(0, 2, 1)
    q, k, v = tok(conv(hn, p.sub("q"))), tok(conv(hn, p.sub("k"))), tok(conv(hn, p.sub("v")))
    o = attention_core(q, k, v, 1, scale=int(c) ** (-0.5))
    return x + conv(o.permute(0, 2, 1).reshape(b, c, h, w), p.sub("proj_out"))


def vae_encode(sd, ddconfig, x, return_fea=True, prefix="encoder."):
    """Encoder.forward(return_fea=True) (model.py:539-572) -> (h, [fea level1, fea level2])."""
    p = SD(sd, prefix)
    n_res, nrb = len(ddconfig["ch_mult"]), ddconfig["num_res_blocks"]
    h = conv(x, p.sub("conv_in"), padding=1)
    fea = []
    for lvl in range(n_res):
        for b in range(nrb):
            h = vae_resnet(h, p.sub(f"down.{lvl}.block.{b}"))
        if lvl in (1, 2):
            fea.append(h)
        if lvl != n_res - 1:
            h = conv(F.pad(h, (0, 1, 0, 1)), p.sub(f"down.{lvl}.downsample.conv"), stride=2)
    h = vae_resnet(h, p.sub("mid.block_1"))
    h = vae_attn(h, p.sub("mid.attn_1"))
    h = vae_resnet(h, p.sub("mid.block_2"))
    h = conv(swish(gn(h, p.sub("norm_out"), 1e-6)), p.sub("conv_out"), padding=1)
    return (h, fea) if return_fea else h


def vae_moments(sd, ddconfig, x):
    """VideoAutoencoderKLResi.encode (autoencoder.py:1674-1679) up to the posterior parameters."""
    h, fea = vae_encode(sd, ddconfig, x)
    moments = conv(h, SD(sd, "quant_conv."))
    mean, logvar = torch.chunk(moments, 2, dim=1)
    return mean, torch.clamp(logvar, -30.0, 20.0), fea


def fuse_resblock(x, p):
    """model.py:1324-1335"""
    h = conv(swish(gn(x, p.sub("norm1"), 1e-6)), p.sub("conv1"), padding=1)
    h = conv(swish(gn(h, p.sub("norm2"), 1e-6)), p.sub("conv2"), padding=1)
    if p.has("conv_out.weight"):
        x = conv(x, p.sub("conv_out"))
    return h + x


def rdb(x, p):
    """ResidualDenseBlock.forward (rrdbnet_arch.py:31-38)."""
    lr = lambda t: F.leaky_relu(t, 0.2)
    x1 = lr(conv(x, p.sub("conv1"), padding=1))
    x2 = lr(conv(torch.cat((x, x1), 1), p.sub("conv2"), padding=1))
    x3 = lr(conv(torch.cat((x, x1, x2), 1), p.sub("conv3"), padding=1))
    x4 = lr(conv(torch.cat((x, x1, x2, x3), 1), p.sub("conv4"), padding=1))
    x5 = conv(torch.cat((x, x1, x2, x3, x4), 1), p.sub("conv5"), padding=1)
    return x5 * 0.2 + x


def fuse_block(enc_feat, dec_feat, p, w, num_block):
    """Fuse_sft_block_ResidualDenseBlock.forward (model.py:1361-1367)."""
    e = fuse_resblock(torch.cat([enc_feat, dec_feat], dim=1), p.sub("encode_enc_1"))
    for i in range(num_block):
        e = rdb(e, p.sub(f"encode_enc_2.{i}"))
    e = fuse_resblock(e, p.sub("encode_enc_3"))
    return dec_feat + w * e


def vae_decode(sd, ddconfig, z, enc_fea, fusion_w=1.0, num_fuse_block=2):
    """VideoAutoencoderKLResi.decode (autoencoder.py:1687-1690) -> VideoDecoder_Mix.forward (model.py:1017-1056)."""
    T = ddconfig["num_frames"]
    n_res, nrb = len(ddconfig["ch_mult"]), ddconfig["num_res_blocks"]
    z = conv(z, SD(sd, "post_quant_conv."))
    p = SD(sd, "decoder.")
    h = conv(z, p.sub("conv_in"), padding=1)
    h = vae_resnet(h, p.sub("mid.block_1"))
    h = spatial_temporal_conv(h, p.sub("temporal_mixing"), T)
    h = vae_attn(h, p.sub("mid.attn_1"))
    h = vae_resnet(h, p.sub("mid.block_2"))
    for lvl in reversed(range(n_res)):
        for b in range(nrb + 1):
            h = vae_resnet(h, p.sub(f"up.{lvl}.block.{b}"))
            h = spatial_temporal_conv(h, p.sub(f"up.{lvl}.temporal_mixing.{b}"), T)
        if lvl != n_res - 1 and lvl != 0:
            h = fuse_block(enc_fea[lvl - 1], h, p.sub(f"fusion_layer_{lvl}"), fusion_w, num_fuse_block)
        if lvl != 0:
            h = conv(F.interpolate(h, scale_factor=2.0, mode="nearest"), p.sub(f"up.{lvl}.upsample.conv"), padding=1)
    return conv(swish(gn(h, p.sub("norm_out"), 1e-6)), p.sub("conv_out"), padding=1)


def vae_image_decode(sd, ddconfig, z, prefix=""):
    """AutoencoderKL.decode (autoencoder.py:361-364) -> Decoder.forward (model.py:648-690): the image decoder behind
    decode_first_stage (ddpm.py:3786)."""
    n_res, nrb = len(ddconfig["ch_mult"]), ddconfig["num_res_blocks"]
    z = conv(z, SD(sd, prefix + "post_quant_conv."))
    p = SD(sd, prefix + "decoder.")
    h = conv(z, p.sub("conv_in"), padding=1)
    h = vae_resnet(h, p.sub("mid.block_1"))
    h = vae_attn(h, p.sub("mid.attn_1"))
    h = vae_resnet(h, p.sub("mid.block_2"))
    for lvl in reversed(range(n_res)):
        for b in range(nrb + 1):
            h = vae_resnet(h, p.sub(f"up.{lvl}.block.{b}"))
        if lvl != 0:
            h = conv(F.interpolate(h, scale_factor=2.0, mode="nearest"), p.sub(f"up.{lvl}.upsample.conv"), padding=1)
    return conv(swish(gn(h, p.sub("norm_out"), 1e-6)), p.sub("conv_out"), padding=1)
