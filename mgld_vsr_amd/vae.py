"""KL-VAE encoder and temporal-aware video decoder on the HIP engine.

Interface mirror of ldm/modules/diffusionmodules/model.py (Encoder :473-572, VideoDecoder_Mix :926-1056, ResnetBlock
:124-183, AttnBlock :192-244, Upsample/Downsample :84-121, ResBlock :1312-1335, Fuse_sft_block_ResidualDenseBlock
:1354-1367), basicsr/archs/rrdbnet_arch.py:9-38 (ResidualDenseBlock), ldm/modules/distributions/distributions.py:24-40
and ldm/models/autoencoder.py (AutoencoderKL :299, VideoAutoencoderKLResi :1564-1690): same constructor kwargs, same
state_dict keys, same encode/decode signatures.

MI355X notes: every 3x3 conv is the MFMA implicit GEMM on NHWC; the dense block's concatenations are slices of one
growing buffer; the asymmetric (0,1,0,1) pad of the encoder downsample and the nearest-2x upsample are folded into
the conv gather; the single-head d=512 mid attention runs as two batched GEMMs around an fp32 row softmax.
"""
import torch
import torch.nn as nn

import os

from . import hip
from .engine import Act, Engine, pack_conv1x1, pack_conv3x3, pack_hp
from .unet import SpatialTemporalConv, _meta_module

# High-precision first-stage encoder (round 5, csrc/hpenc.hip): the latent it produces conditions every sampling step, so its fp16
# error is a bias on x_0 (2.4e-3 on smooth frames against 9.0e-4 with an exact latent).  fp32 activations between the kernels,
# split-fp16 operands inside the contractions (3x the encoder's MFMA work = +6 % of a segment's).  MGLD_HP_ENCODER=0: the fp16 encoder.
HP_ENCODER = os.environ.get("MGLD_HP_ENCODER", "1") != "0"


def Normalize(in_channels, num_groups=32):
    return nn.GroupNorm(num_groups=num_groups, num_channels=in_channels, eps=1e-6, affine=True)


def _conv3(eng, conv, x, out=None, resid=None, act=hip.ACT_NONE, alpha=1.0, beta=1.0, **kw):
    up2 = bool(kw.get("up2", False))
    w, w2 = eng.weight2("c3up" if up2 else "c3", (conv.weight,), lambda t: pack_conv3x3(t, x.C, tap_inner=False if up2 else None))
    return eng.conv3x3(x, w, eng.f32("b", conv.bias), conv.out_channels, out=out, resid=resid, act=act, alpha=alpha, beta=beta,
                       w2=w2, **kw)


def _conv1(eng, conv, x, out=None, resid=None, alpha=1.0, beta=1.0, lo=False):
    w, w2 = eng.weight2("c1", (conv.weight,), lambda t: pack_conv1x1(t, x.C))
    return eng.linear(x, w, eng.f32("b", conv.bias), out=out, resid=resid, alpha=alpha, beta=beta, w2=w2, lo=lo)


def _gn(eng, norm, x, silu):
    return eng.groupnorm(x, eng.f32("g", norm.weight), eng.f32("b", norm.bias), norm.eps, silu)


# ---- high-precision (hp) building blocks: fp32 Acts in and out ---------------------------------------------------------------
def _hp_split(eng, x, norm=None, silu=False, out_f32=False):
    """fp32 Act -> [GroupNorm (+ SiLU) in fp32 ->] the split-fp16 contraction operand [yh | 16 yl | yh/256] (Act fp16 [rows, 3C]),
    or the normalised tensor itself as fp32 (out_f32)"""
    gs = g = b = None
    if norm is not None:
        gs = eng.arena.alloc((x.n * hip.hp_chunks(x.hw) * norm.num_groups * 2,), torch.float64)
        hip.hp_gn_stats(x.v, x.n, x.hw, norm.num_groups, gs)
        g, b = eng.f32("g", norm.weight), eng.f32("b", norm.bias)
        eng.launches += 1
    out = eng.act(x.n, x.h, x.w, x.C if out_f32 else 3 * x.C, torch.float32 if out_f32 else torch.float16)
    hip.hp_gn_split(x.v, gs, norm.eps if norm is not None else 0.0, g, b, silu, out.v, x.n, x.hw, norm.num_groups if norm is not None else 1)
    eng.launches += 1
    return out


def _hp_conv3(eng, conv, a3, resid=None, stride=1, pad=(1, 1), hw_out=None):
    """3x3 convolution of a split operand against split weights -> fp32 Act (+ fp32 residual)"""
    w = eng.weight("c3hp", (conv.weight,), lambda t: pack_conv3x3(pack_hp(t), a3.C))
    return eng.conv3x3(a3, w, eng.f32("b", conv.bias), conv.out_channels, resid=resid, stride=stride, pad=pad, hw_out=hw_out,
                       out_dtype=torch.float32)


def _hp_conv1(eng, conv, a3, resid=None):
    w = eng.weight("c1hp", (conv.weight,), lambda t: pack_conv1x1(pack_hp(t.reshape(t.shape[0], t.shape[1])), a3.C))
    return eng.linear(a3, w, eng.f32("b", conv.bias), resid=resid, out_dtype=torch.float32)


def _f32_conv(eng, conv, x, ksize=(1, 1), pad=(0, 0), resid=None, out=None, cin=None):
    """a convolution on the f32-input MFMA (mgld_conv_f32): the 3- / 8-channel ends of the encoder and its attention block"""
    from .raft import _conv, pack_conv_f32
    w = eng.weight("c32", (conv.weight,), pack_conv_f32, torch.float32)
    return _conv(eng, x, w, eng.f32("b", conv.bias), conv.out_channels, ksize, 1, pad, resid=resid, out=out, cin=cin)


class Upsample(nn.Module):
    def __init__(self, in_channels, with_conv):
        super().__init__()
        assert with_conv
        self.conv = nn.Conv2d(in_channels, in_channels, 3, 1, 1)

    def run(self, eng, x):
        return _conv3(eng, self.conv, x, up2=True, lo=True)


class Downsample(nn.Module):
    def __init__(self, in_channels, with_conv):
        super().__init__()
        assert with_conv
        self.conv = nn.Conv2d(in_channels, in_channels, 3, 2, 0)

    def run(self, eng, x):
        # F.pad (0,1,0,1) then stride-2 valid conv (model.py:114-118): pad_t = pad_l = 0, bottom/right implied
        return _conv3(eng, self.conv, x, stride=2, pad=(0, 0), hw_out=(x.h // 2, x.w // 2), lo=True)

    def run_hp(self, eng, x):
        return _hp_conv3(eng, self.conv, _hp_split(eng, x), stride=2, pad=(0, 0), hw_out=(x.h // 2, x.w // 2))


class ResnetBlock(nn.Module):
    def __init__(self, *, in_channels, out_channels=None, conv_shortcut=False, dropout, temb_channels=512):
        super().__init__()
        assert not conv_shortcut and temb_channels == 0
        out_channels = in_channels if out_channels is None else out_channels
        self.in_channels, self.out_channels = in_channels, out_channels
        self.norm1 = Normalize(in_channels)
        self.conv1 = nn.Conv2d(in_channels, out_channels, 3, 1, 1)
        self.norm2 = Normalize(out_channels)
        self.dropout = nn.Dropout(dropout)
        self.conv2 = nn.Conv2d(out_channels, out_channels, 3, 1, 1)
        if in_channels != out_channels:
            self.nin_shortcut = nn.Conv2d(in_channels, out_channels, 1, 1, 0)

    def run(self, eng, x, out=None):
        h = _conv3(eng, self.conv1, _gn(eng, self.norm1, x, True), stats=True)          # norm2 reads the epilogue's statistics
        skip = x if self.in_channels == self.out_channels else _conv1(eng, self.nin_shortcut, x, lo=True)
        return _conv3(eng, self.conv2, _gn(eng, self.norm2, h, True), out=out, resid=skip, stats=True, lo=True)   # the next block's norm likewise

    def run_hp(self, eng, x):
        """model.py:162-183 with fp32 activations (x, the result and the skip are fp32 Acts)"""
        h = _hp_conv3(eng, self.conv1, _hp_split(eng, x, self.norm1, True))
        skip = x if self.in_channels == self.out_channels else _hp_conv1(eng, self.nin_shortcut, _hp_split(eng, x))
        return _hp_conv3(eng, self.conv2, _hp_split(eng, h, self.norm2, True), resid=skip)


class AttnBlock(nn.Module):
    """single head, d = C (512 at full size): S = q k^T * C^-1/2 (fp32) -> row softmax -> P v."""

    def __init__(self, in_channels):
        super().__init__()
        self.in_channels = in_channels
        self.norm = Normalize(in_channels)
        self.q = nn.Conv2d(in_channels, in_channels, 1)
        self.k = nn.Conv2d(in_channels, in_channels, 1)
        self.v = nn.Conv2d(in_channels, in_channels, 1)
        self.proj_out = nn.Conv2d(in_channels, in_channels, 1)

    def run(self, eng, x):
        C, N, F = self.in_channels, x.hw, x.n
        hn = _gn(eng, self.norm, x, False)
        wqk = eng.weight("qk", (self.q.weight, self.k.weight), lambda q, k: torch.cat([q.reshape(C, C), k.reshape(C, C)], 0))
        bqk = eng.weight("bqk", (self.q.bias, self.k.bias), lambda q, k: torch.cat([q, k], 0), torch.float32)
        qk = eng.linear(hn, wqk, bqk).v                                             # [F*N, 2C]
        wv = eng.weight("c1", (self.v.weight,), pack_conv1x1)
        vt = eng.arena.alloc((F * C, N), torch.float16)                             # per frame V^T [C, N]
        hip.igemm(wv, hn.v, vt, bias_m=eng.f32("b", self.v.bias), M=C, N=N, K=C, batch=F, strideA=0,
                  strideW=N * hn.v.stride(0), strideC=C * N)
        S = eng.arena.alloc((F * N, N), torch.float32)
        hip.igemm(qk, qk[:, C:], S, M=N, N=N, K=C, alpha=float(int(C) ** (-0.5)), batch=F, strideA=N * 2 * C,
                  strideW=N * 2 * C, strideC=N * N)
        P = eng.arena.alloc((F * N, N), torch.float16)
        hip.softmax_rows(S, P, F * N, N)
        o = eng.act(x.n, x.h, x.w, C)
        hip.igemm(P, vt, o.v, M=N, N=C, K=N, batch=F, strideA=N * N, strideW=C * N, strideC=N * C)
        eng.launches += 4
        return _conv1(eng, self.proj_out, o, resid=x, lo=True)

    def run_hp(self, eng, x):
        """the same block entirely in fp32 on the f32-input MFMA (model.py:209-244): q, k, v^T, S = q k^T / sqrt(C), row softmax,
        P v (+ the v bias: the rows of P sum to one), proj_out + x"""
        from .raft import _conv, pack_conv_f32
        C, N, F = self.in_channels, x.hw, x.n
        f32 = torch.float32
        hn = _hp_split(eng, x, self.norm, False, out_f32=True)
        q, k = _f32_conv(eng, self.q, hn), _f32_conv(eng, self.k, hn)
        wv = eng.weight("c32", (self.v.weight,), pack_conv_f32, f32)                    # [C, C]
        vt = eng.arena.alloc((F * C, N), f32)                                           # per frame V^T [C, N] = Wv hn^T
        hip.conv_f32(wv, hn.v, vt, 1, 1, C, C, batch=F, strideA=0, strideW=N * C, strideC=C * N, n_out=N)
        S = eng.arena.alloc((F * N, N), f32)
        hip.conv_f32(q.v, k.v, S, 1, x.h, x.w, C, alpha=float(int(C) ** (-0.5)), batch=F, strideA=N * C, strideW=N * C, strideC=N * N, n_out=N)
        hip.hp_softmax_rows(S)
        o = eng.act(x.n, x.h, x.w, C, f32)
        hip.conv_f32(S, vt, o.v, 1, x.h, x.w, N, bias=eng.f32("b", self.v.bias), batch=F, strideA=N * N, strideW=C * N, strideC=N * C, n_out=C)
        eng.launches += 4
        return _f32_conv(eng, self.proj_out, o, resid=x)


def make_attn(in_channels, attn_type="vanilla"):
    assert attn_type == "vanilla"
    return AttnBlock(in_channels)


class Encoder(nn.Module):
    def __init__(self, *, ch, out_ch, ch_mult=(1, 2, 4, 8), num_res_blocks, attn_resolutions, dropout=0.0,
                 resamp_with_conv=True, in_channels, resolution, z_channels, double_z=True, use_linear_attn=False,
                 attn_type="vanilla", **ignore_kwargs):
        super().__init__()
        assert not attn_resolutions and not use_linear_attn
        self.ch, self.num_resolutions, self.num_res_blocks = ch, len(ch_mult), num_res_blocks
        self.resolution, self.in_channels = resolution, in_channels
        self.conv_in = nn.Conv2d(in_channels, ch, 3, 1, 1)
        in_ch_mult = (1,) + tuple(ch_mult)
        self.down = nn.ModuleList()
        block_in = ch
        for i_level in range(self.num_resolutions):
            block = nn.ModuleList()
            block_in, block_out = ch * in_ch_mult[i_level], ch * ch_mult[i_level]
            for _ in range(num_res_blocks):
                block.append(ResnetBlock(in_channels=block_in, out_channels=block_out, temb_channels=0, dropout=dropout))
                block_in = block_out
            down = nn.Module()
            down.block = block
            down.attn = nn.ModuleList()
            if i_level != self.num_resolutions - 1:
                down.downsample = Downsample(block_in, resamp_with_conv)
            self.down.append(down)
        self.mid = nn.Module()
        self.mid.block_1 = ResnetBlock(in_channels=block_in, out_channels=block_in, temb_channels=0, dropout=dropout)
        self.mid.attn_1 = make_attn(block_in, attn_type)
        self.mid.block_2 = ResnetBlock(in_channels=block_in, out_channels=block_in, temb_channels=0, dropout=dropout)
        self.norm_out = Normalize(block_in)
        self.conv_out = nn.Conv2d(block_in, 2 * z_channels if double_z else z_channels, 3, 1, 1)

    def run(self, eng, x, fea_out=None):
        """x: Act (channels padded to 8). Returns (h Act [n, h/8, w/8, 2z], [fea level1, fea level2]).
        fea_out: optional list of 2 Acts to write the encoder features into (decoder concat buffers)."""
        h = _conv3(eng, self.conv_in, x, lo=True)
        fea = []
        for lvl in range(self.num_resolutions):
            nb = len(self.down[lvl].block)
            for bi, blk in enumerate(self.down[lvl].block):
                o = None
                if fea_out is not None and lvl in (1, 2) and bi == nb - 1:
                    o = fea_out[lvl - 1]
                h = blk.run(eng, h, out=o)
            if lvl in (1, 2):
                fea.append(h)
            if lvl != self.num_resolutions - 1:
                h = self.down[lvl].downsample.run(eng, h)
        h = self.mid.block_1.run(eng, h)
        h = self.mid.attn_1.run(eng, h)
        h = self.mid.block_2.run(eng, h)
        h = _conv3(eng, self.conv_out, _gn(eng, self.norm_out, h, True))
        return h, fea

    def run_hp(self, eng, x):
        """the encoder in high precision (csrc/hpenc.hip): x NCHW fp32 device tensor -> fp32 Act [n, h/8, w/8, 2z]"""
        n, c, H, W = x.shape
        xa = eng.act(n, H, W, 4, torch.float32)
        hip.nchw_to_nhwc_f32(x, xa.v)
        eng.launches += 1
        h = _f32_conv(eng, self.conv_in, xa, (3, 3), (1, 1))
        fea = []
        for lvl in range(self.num_resolutions):
            for blk in self.down[lvl].block:
                h = blk.run_hp(eng, h)
            if lvl in (1, 2):
                fea.append(h)                                   # (fp32; the video VAE's decoder features when it encodes this way)
            if lvl != self.num_resolutions - 1:
                h = self.down[lvl].downsample.run_hp(eng, h)
        h = self.mid.block_1.run_hp(eng, h)
        h = self.mid.attn_1.run_hp(eng, h)
        h = self.mid.block_2.run_hp(eng, h)
        return _f32_conv(eng, self.conv_out, _hp_split(eng, h, self.norm_out, True, out_f32=True), (3, 3), (1, 1)), fea


class Decoder(nn.Module):
    """model.py:575-690, the image decoder of AutoencoderKL (`decode_first_stage`; the VSR scripts decode with the video VAE, this
    one serves `LatentDiffusionVSRTextWT.decode_first_stage` / `AutoencoderKL.decode`)."""

    def __init__(self, *, ch, out_ch, ch_mult=(1, 2, 4, 8), num_res_blocks, attn_resolutions, dropout=0.0, resamp_with_conv=True,
                 in_channels, resolution, z_channels, give_pre_end=False, tanh_out=False, use_linear_attn=False, attn_type="vanilla",
                 **ignorekwargs):
        super().__init__()
        assert not attn_resolutions and not give_pre_end and not tanh_out and not use_linear_attn
        self.ch, self.num_resolutions, self.num_res_blocks, self.out_ch = ch, len(ch_mult), num_res_blocks, out_ch
        block_in = ch * ch_mult[self.num_resolutions - 1]
        self.conv_in = nn.Conv2d(z_channels, block_in, 3, 1, 1)
        self.mid = nn.Module()
        self.mid.block_1 = ResnetBlock(in_channels=block_in, out_channels=block_in, temb_channels=0, dropout=dropout)
        self.mid.attn_1 = make_attn(block_in, attn_type)
        self.mid.block_2 = ResnetBlock(in_channels=block_in, out_channels=block_in, temb_channels=0, dropout=dropout)
        self.up = nn.ModuleList()
        for i_level in reversed(range(self.num_resolutions)):
            block = nn.ModuleList()
            block_out = ch * ch_mult[i_level]
            for _ in range(num_res_blocks + 1):
                block.append(ResnetBlock(in_channels=block_in, out_channels=block_out, temb_channels=0, dropout=dropout))
                block_in = block_out
            up = nn.Module()
            up.block, up.attn = block, nn.ModuleList()
            if i_level != 0:
                up.upsample = Upsample(block_in, resamp_with_conv)
            self.up.insert(0, up)
        self.norm_out = Normalize(block_in)
        self.conv_out = nn.Conv2d(block_in, out_ch, 3, 1, 1)

    def run(self, eng, z):
        h = _conv3(eng, self.conv_in, z, lo=True)
        h = self.mid.block_1.run(eng, h)
        h = self.mid.attn_1.run(eng, h)
        h = self.mid.block_2.run(eng, h)
        for lvl in reversed(range(self.num_resolutions)):
            for blk in self.up[lvl].block:
                h = blk.run(eng, h)
            if lvl != 0:
                h = self.up[lvl].upsample.run(eng, h)
        t = _gn(eng, self.norm_out, h, True)
        out = Act(eng.arena.alloc((h.rows, self.out_ch), torch.float32), h.n, h.h, h.w)
        return _conv3(eng, self.conv_out, t, out=out)


class ResBlock(nn.Module):
    """model.py:1312-1335 (fusion-layer residual block: GN/swish/conv x2, 1x1 conv_out on the skip)."""

    def __init__(self, in_channels, out_channels=None):
        super().__init__()
        self.in_channels = in_channels
        self.out_channels = in_channels if out_channels is None else out_channels
        self.norm1 = Normalize(in_channels)
        self.conv1 = nn.Conv2d(in_channels, out_channels, 3, 1, 1)
        self.norm2 = Normalize(out_channels)
        self.conv2 = nn.Conv2d(out_channels, out_channels, 3, 1, 1)
        if self.in_channels != self.out_channels:
            self.conv_out = nn.Conv2d(in_channels, out_channels, 1, 1, 0)

    def run(self, eng, x, out=None, resid_extra=None):
        h = _conv3(eng, self.conv1, _gn(eng, self.norm1, x, True), stats=True)          # norm2 reads the epilogue's statistics
        skip = x if self.in_channels == self.out_channels else _conv1(eng, self.conv_out, x, lo=True)
        return _conv3(eng, self.conv2, _gn(eng, self.norm2, h, True), out=out, resid=skip, stats=True, lo=True)   # the next block's norm likewise


class ResidualDenseBlock(nn.Module):
    """rrdbnet_arch.py:9-38: x1..x4 grow inside one [rows, C+4*32] buffer (no concatenation copies)."""

    def __init__(self, num_feat=64, num_grow_ch=32):
        super().__init__()
        self.nf, self.gc = num_feat, num_grow_ch
        self.conv1 = nn.Conv2d(num_feat, num_grow_ch, 3, 1, 1)
        self.conv2 = nn.Conv2d(num_feat + num_grow_ch, num_grow_ch, 3, 1, 1)
        self.conv3 = nn.Conv2d(num_feat + 2 * num_grow_ch, num_grow_ch, 3, 1, 1)
        self.conv4 = nn.Conv2d(num_feat + 3 * num_grow_ch, num_grow_ch, 3, 1, 1)
        self.conv5 = nn.Conv2d(num_feat + 4 * num_grow_ch, num_feat, 3, 1, 1)

    def run(self, eng, buf, out_buf):
        """buf: Act [rows, nf+4gc] whose first nf columns hold x; writes x5*0.2+x into out_buf[:, :nf]."""
        nf, gc = self.nf, self.gc
        for i, conv in enumerate([self.conv1, self.conv2, self.conv3, self.conv4]):
            _conv3(eng, conv, buf.cols(0, nf + i * gc), out=buf.cols(nf + i * gc, nf + (i + 1) * gc, lo=False), act=hip.ACT_LRELU02)
        return _conv3(eng, self.conv5, buf, out=out_buf.cols(0, nf), resid=buf.cols(0, nf), alpha=0.2, beta=1.0)   # x5 * 0.2 + x: the stream


class Fuse_sft_block_ResidualDenseBlock(nn.Module):
    def __init__(self, in_ch, out_ch, num_block=1, num_grow_ch=32):
        super().__init__()
        self.in_ch, self.gc = in_ch, num_grow_ch
        self.encode_enc_1 = ResBlock(2 * in_ch, in_ch)
        self.encode_enc_2 = nn.Sequential(*[ResidualDenseBlock(num_feat=in_ch, num_grow_ch=num_grow_ch) for _ in range(num_block)])
        self.encode_enc_3 = ResBlock(in_ch, out_ch)

    def run(self, eng, cat, w):
        """cat: Act [rows, 2*in_ch] = [enc_feat | dec_feat]; returns dec_feat + w * f(cat)."""
        dec = cat.cols(self.in_ch, 2 * self.in_ch)
        width = self.in_ch + 4 * self.gc
        bufs = [eng.act(cat.n, cat.h, cat.w, width, lo=True) for _ in range(len(self.encode_enc_2) + 1)]   # (low plane: the first in_ch columns use it)
        self.encode_enc_1.run(eng, cat, out=bufs[0].cols(0, self.in_ch))
        for i, blk in enumerate(self.encode_enc_2):
            blk.run(eng, bufs[i], bufs[i + 1])
        e = self.encode_enc_3.run(eng, bufs[-1].cols(0, self.in_ch))
        out = eng.act(cat.n, cat.h, cat.w, self.in_ch, lo=True)
        hip.copy2d(dec.v, out.v)
        if out.lo is not None:                    # dec_feat + w * f(cat) on the two-plane stream
            if dec.lo is not None:
                hip.copy2d(dec.lo, out.lo)
            else:
                out.lo.zero_()
            hip.axpby_lo(e.v, e.lo, out.v, out.lo, float(w), 1.0)
            eng.launches += 1
        else:
            hip.axpby(e.v, out.v, float(w), 1.0)
        eng.launches += 2
        return out


class VideoDecoder_Mix(nn.Module):
    def __init__(self, *, ch, out_ch, ch_mult=(1, 2, 4, 8), num_res_blocks, attn_resolutions, num_frames=1, dropout=0.0,
                 resamp_with_conv=True, in_channels, resolution, z_channels, give_pre_end=False, tanh_out=False,
                 use_linear_attn=False, attn_type="vanilla", num_fuse_block=2, fusion_w=1.0, **ignorekwargs):
        super().__init__()
        assert not attn_resolutions and not give_pre_end and not tanh_out
        self.ch, self.num_resolutions, self.num_res_blocks = ch, len(ch_mult), num_res_blocks
        self.resolution, self.in_channels, self.num_frames, self.fusion_w = resolution, in_channels, num_frames, fusion_w
        self.out_ch = out_ch
        block_in = ch * ch_mult[self.num_resolutions - 1]
        curr_res = resolution // 2 ** (self.num_resolutions - 1)
        self.z_shape = (1, z_channels, curr_res, curr_res)
        self.conv_in = nn.Conv2d(z_channels, block_in, 3, 1, 1)
        self.mid = nn.Module()
        self.mid.block_1 = ResnetBlock(in_channels=block_in, out_channels=block_in, temb_channels=0, dropout=dropout)
        self.temporal_mixing = SpatialTemporalConv(num_feat=block_in, num_frames=num_frames)
        self.mid.attn_1 = make_attn(block_in, attn_type)
        self.mid.block_2 = ResnetBlock(in_channels=block_in, out_channels=block_in, temb_channels=0, dropout=dropout)
        self.up = nn.ModuleList()
        for i_level in reversed(range(self.num_resolutions)):
            block, temporal_mixing = nn.ModuleList(), nn.ModuleList()
            block_out = ch * ch_mult[i_level]
            if i_level != self.num_resolutions - 1 and i_level != 0:
                setattr(self, f"fusion_layer_{i_level}",
                        Fuse_sft_block_ResidualDenseBlock(in_ch=block_out, out_ch=block_out, num_block=num_fuse_block))
            for _ in range(num_res_blocks + 1):
                block.append(ResnetBlock(in_channels=block_in, out_channels=block_out, temb_channels=0, dropout=dropout))
                block_in = block_out
                temporal_mixing.append(SpatialTemporalConv(num_feat=block_in, num_frames=num_frames))
            up = nn.Module()
            up.block, up.temporal_mixing, up.attn = block, temporal_mixing, nn.ModuleList()
            if i_level != 0:
                up.upsample = Upsample(block_in, resamp_with_conv)
            self.up.insert(0, up)
        self.norm_out = Normalize(block_in)
        self.conv_out = nn.Conv2d(block_in, out_ch, 3, 1, 1)

    def run(self, eng, z, enc_fea):
        """z: Act [n,h,w,8] (post_quant_conv output, padded); enc_fea: [Act level1 (H/2), Act level2 (H/4)].
        Returns fp32 token-major output Act [n*H*W, out_ch]."""
        # (precision scopes, engine.w2_scopes: vae_dec_mid / vae_dec_up<level> / vae_dec_fuse / vae_dec_out inside the caller's vae_dec)
        with eng.scope("vae_dec_mid"):
            h = _conv3(eng, self.conv_in, z, lo=True)
            h = self.mid.block_1.run(eng, h)
            h = self.temporal_mixing.run(eng, h)
            h = self.mid.attn_1.run(eng, h)
            h = self.mid.block_2.run(eng, h)
        for lvl in reversed(range(self.num_resolutions)):
            fuse = lvl != self.num_resolutions - 1 and lvl != 0
            nb = self.num_res_blocks + 1
            cat = None
            with eng.scope(f"vae_dec_up{lvl}"):
                for b in range(nb):
                    h = self.up[lvl].block[b].run(eng, h)
                    o = None
                    if fuse and b == nb - 1:
                        ef = enc_fea[lvl - 1]
                        cat = eng.act(h.n, h.h, h.w, ef.C + h.C, lo=True)
                        hip.copy2d(ef.v, cat.v[:, :ef.C])
                        if cat.lo is not None:
                            if ef.lo is not None:
                                hip.copy2d(ef.lo, cat.lo[:, :ef.C])
                            else:
                                cat.lo[:, :ef.C].zero_()
                        eng.launches += 1
                        o = cat.cols(ef.C, ef.C + h.C)
                    h = self.up[lvl].temporal_mixing[b].run(eng, h, out=o)
            if fuse:
                with eng.scope("vae_dec_fuse"):
                    h = getattr(self, f"fusion_layer_{lvl}").run(eng, cat, self.fusion_w)
            if lvl != 0:
                with eng.scope(f"vae_dec_up{lvl}"):
                    h = self.up[lvl].upsample.run(eng, h)
        t = _gn(eng, self.norm_out, h, True)
        out = Act(eng.arena.alloc((h.rows, self.out_ch), torch.float32), h.n, h.h, h.w)
        with eng.scope("vae_dec_out"):
            return _conv3(eng, self.conv_out, t, out=out)


class DiagonalGaussianDistribution(object):
    """ldm/modules/distributions/distributions.py:24-40 (device tensors, fp32)."""

    def __init__(self, parameters, deterministic=False):
        self.parameters = parameters
        self.mean, self._raw_logvar = torch.chunk(parameters, 2, dim=1)          # views: no launch
        self.deterministic = deterministic
        self._lv = self._std = self._var = None

    # clamp / exp run only if somebody asks for these tensors: the segment path samples through mgld_init_latent (one launch for
    # clamp + exp + sample + scale + q_sample) instead of five vendor elementwise kernels
    @property
    def logvar(self):
        if self._lv is None:
            self._lv = torch.clamp(self._raw_logvar, -30.0, 20.0)
        return self._lv

    @property
    def std(self):
        if self._std is None:
            self._std = torch.zeros_like(self.mean) if self.deterministic else torch.exp(0.5 * self.logvar)
        return self._std

    @property
    def var(self):
        if self._var is None:
            self._var = torch.zeros_like(self.mean) if self.deterministic else torch.exp(self.logvar)
        return self._var

    def sample(self, noise=None):
        if noise is None:  # the reference draws on the CPU then moves (distributions.py:35-37)
            noise = torch.randn(self.mean.shape)
        return self.mean + self.std * noise.to(self.parameters.device)

    def mode(self):
        return self.mean


class _AutoencoderBase(nn.Module):
    ENC_SCOPE, DEC_SCOPE = "first", "first_dec"      # precision scopes (engine.w2_scopes) of the first-stage (image) autoencoder

    def engine(self):
        if self._engine is None:
            self._engine = Engine()
        return self._engine

    def set_engine(self, eng):
        self._engine = eng

    def _finish_init(self):
        self.to_empty(device="cpu")
        for p in self.parameters():
            p.data.zero_()
            p.requires_grad_(False)
        self._engine = None

    def _moments(self, eng, x, fea_out=None):
        """x: NCHW fp32 device tensor -> (moments NCHW fp32 [n, 2*embed, h/8, w/8], [fea Acts])."""
        xa = eng.from_nchw(x)
        with eng.scope(self.ENC_SCOPE):
            h, fea = self.encoder.run(eng, xa, fea_out=fea_out)
            m = Act(eng.arena.alloc((h.rows, 2 * self.embed_dim), torch.float32), h.n, h.h, h.w)
            wq, wq2 = eng.weight2("c1", (self.quant_conv.weight,), lambda t: pack_conv1x1(t, h.C))
            eng.linear(h, wq, eng.f32("b", self.quant_conv.bias), out=m, w2=wq2)
        return eng.to_nchw(m, 2 * self.embed_dim), fea

    HP_FRAMES = 4      # frames per pass of the high-precision encoder (every op of it is per frame): bounds its fp32 arena to ~17 GiB at 512^2

    def _moments_hp(self, eng, x, want_fea=False):
        """the moments through the high-precision encoder + quant_conv on the f32-input MFMA -> (NCHW fp32, [fp16 feature Acts in
        owned storage] when want_fea).  Runs HP_FRAMES frames at a time and rewinds the arena between the passes (stream-ordered
        reuse): the fp32 / split tensors of the 512^2 level are 1-1.6 GB per 8 frames each."""
        x = x.contiguous()
        n, _, H, W = x.shape
        out = torch.empty(n, 2 * self.embed_dim, H // 8, W // 8, dtype=torch.float32, device=eng.device)
        keep = None
        mark = (eng.arena.ci, eng.arena.off)
        for c0 in range(0, n, self.HP_FRAMES):
            c1 = min(n, c0 + self.HP_FRAMES)
            eng.arena.ci, eng.arena.off = mark
            h, fea = self.encoder.run_hp(eng, x[c0:c1])
            m = _f32_conv(eng, self.quant_conv, h)
            hip.nhwc_to_nchw(m.v, out[c0:c1])
            eng.launches += 1
            if want_fea:
                if keep is None:
                    keep = [Act(torch.empty(n * f.hw, f.C, dtype=torch.float16, device=eng.device), n, f.h, f.w) for f in fea]
                for k, f in zip(keep, fea):
                    hip.copy2d(_hp_split(eng, f).v[:, :f.C], k.v[c0 * f.hw:c1 * f.hw])      # first block of the split = fp16(f)
                    eng.launches += 1
        return out, keep

    def init_from_ckpt(self, path, ignore_keys=list(), only_model=False):
        """autoencoder.py:1652-1672: accepts full-model checkpoints by stripping the `first_stage_model.` prefix."""
        from .util import load_trusted_checkpoint
        sd = load_trusted_checkpoint(path)
        if "state_dict" in sd:
            sd = sd["state_dict"]
        for k in list(sd.keys()):
            if "first_stage_model" in k:
                sd[k[18:]] = sd.pop(k)
        for k in list(sd.keys()):
            if any(k.startswith(ik) for ik in ignore_keys):
                del sd[k]
        missing, unexpected = self.load_state_dict(sd, strict=False)
        self.last_load = (list(missing), list(unexpected))       # (the reference prints both lists and returns `missing`)
        return missing


class AutoencoderKL(_AutoencoderBase):
    """ldm/models/autoencoder.py:299 — encode is on the VSR hot path (init latent, ddpm.py:3906-3943); decode (:361) backs
    `decode_first_stage`."""

    def __init__(self, ddconfig, lossconfig=None, embed_dim=4, ckpt_path=None, ignore_keys=[], image_key="image",
                 colorize_nlabels=None, monitor=None, **kw):
        super().__init__()
        self.embed_dim = embed_dim
        with _meta_module():
            self.encoder = Encoder(**ddconfig)
            self.decoder = Decoder(**ddconfig)
            self.quant_conv = nn.Conv2d(2 * ddconfig["z_channels"], 2 * embed_dim, 1)
            self.post_quant_conv = nn.Conv2d(embed_dim, ddconfig["z_channels"], 1)
        self._finish_init()
        if ckpt_path is not None:
            self.init_from_ckpt(ckpt_path, ignore_keys)

    @torch.no_grad()
    def decode(self, z):
        """autoencoder.py:361-364: post_quant_conv -> Decoder.  z [n, embed_dim, h, w] -> [n, 3, 8h, 8w] fp32 on the device."""
        eng = self.engine()
        eng.reset()
        za = eng.from_nchw(z.to(eng.device, torch.float32))
        zq = eng.act(za.n, za.h, za.w, 8)
        zq.v.zero_()
        with eng.scope(self.DEC_SCOPE):
            wq, wq2 = eng.weight2("c1", (self.post_quant_conv.weight,), lambda t: pack_conv1x1(t, za.C))
            eng.linear(za, wq, eng.f32("b", self.post_quant_conv.bias), out=zq.cols(0, self.post_quant_conv.out_channels), w2=wq2)
            out = self.decoder.run(eng, zq)
        return eng.to_nchw(out, self.decoder.out_ch)

    @torch.no_grad()
    def encode(self, x, return_encfea=False):
        """autoencoder.py:347-353: the posterior, plus the moments tensor it was built from when `return_encfea`"""
        eng = self.engine()
        eng.reset()
        if HP_ENCODER:
            m, _ = self._moments_hp(eng, x.to(eng.device, torch.float32))
        else:
            m, _ = self._moments(eng, x.to(eng.device, torch.float32))
        posterior = DiagonalGaussianDistribution(m)
        return (posterior, m) if return_encfea else posterior


class VideoAutoencoderKLResi(_AutoencoderBase):
    """ldm/models/autoencoder.py:1564-1690: encode(x) -> (posterior, enc_fea); decode(z, enc_fea) -> frames."""
    ENC_SCOPE, DEC_SCOPE = "vae_enc", "vae_dec"

    def __init__(self, ddconfig, lossconfig=None, embed_dim=4, ckpt_path=None, ignore_keys=[], image_key="image",
                 colorize_nlabels=None, monitor=None, fusion_w=1.0, freeze_dec=True, synthesis_data=False, use_usm=False,
                 test_gt=False, version=1):
        super().__init__()
        assert version == 1, "VideoDecoder_MixV2 is not on the MGLD-VSR hot path"
        self.embed_dim = embed_dim
        with _meta_module():
            self.encoder = Encoder(**ddconfig)
            self.decoder = VideoDecoder_Mix(**ddconfig)
            self.quant_conv = nn.Conv2d(2 * ddconfig["z_channels"], 2 * embed_dim, 1)
            self.post_quant_conv = nn.Conv2d(embed_dim, ddconfig["z_channels"], 1)
        self._finish_init()
        self.decoder.fusion_w = fusion_w
        if ckpt_path is not None:
            self.init_from_ckpt(ckpt_path, ignore_keys)

    @torch.no_grad()
    def encode(self, x, hp=False):
        """-> (DiagonalGaussianDistribution, [fea1, fea2]); features stay on the device as NHWC fp16 Acts.
        hp: the posterior of this call becomes the sampler's struct-cond latent (the `_old` / `_w_latent` scripts take it from the video
        VAE, old.py:328-329): run the encoder in high precision like the first-stage one (the features are its fp32 ones, rounded once)."""
        eng = self.engine()
        eng.reset()
        if hp and HP_ENCODER:
            m, keep = self._moments_hp(eng, x.to(eng.device, torch.float32), want_fea=True)
            return DiagonalGaussianDistribution(m), keep
        m, fea = self._moments(eng, x.to(eng.device, torch.float32))
        # features outlive the arena pass: move them to owned storage
        keep = []
        for f in fea:
            t = torch.empty(f.rows, f.C, dtype=torch.float16, device=eng.device)
            hip.copy2d(f.v, t)
            tl = None
            if f.lo is not None:                  # the features are residual-stream tensors of the encoder: their low plane goes along
                tl = torch.empty(f.rows, f.C, dtype=torch.float16, device=eng.device)
                hip.copy2d(f.lo, tl)
            keep.append(Act(t, f.n, f.h, f.w, tl))
        return DiagonalGaussianDistribution(m), keep

    @torch.no_grad()
    def decode(self, z, enc_fea):
        eng = self.engine()
        eng.reset()
        z = z.to(eng.device, torch.float32)
        fea = [f if isinstance(f, Act) else eng.from_nchw(f.to(eng.device, torch.float32)) for f in enc_fea]
        za = eng.from_nchw(z)
        # post_quant_conv output padded to 8 channels (zero weight rows) so conv_in sees Cin % 8 == 0
        zq = eng.act(za.n, za.h, za.w, 8)
        zq.v.zero_()
        with eng.scope(self.DEC_SCOPE):
            wq, wq2 = eng.weight2("c1", (self.post_quant_conv.weight,), lambda t: pack_conv1x1(t, za.C))
            eng.linear(za, wq, eng.f32("b", self.post_quant_conv.bias), out=zq.cols(0, self.post_quant_conv.out_channels), w2=wq2)
            out = self.decoder.run(eng, zq, fea)
        return eng.to_nchw(out, self.decoder.out_ch)

    def forward(self, input, latent, sample_posterior=True):
        posterior, enc_fea = self.encode(input)
        return self.decode(latent, enc_fea), posterior
