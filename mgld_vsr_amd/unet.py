"""SD-2.1 conditional UNet (InflatedUNetModelDualcondV2) and time-aware structure-condition encoder
(InflatedEncoderUNetModelWT) on the HIP engine.

Interface mirror of ldm/modules/diffusionmodules/openaimodel.py:1903-2313 and :2316-2525 (same constructor kwargs,
same state_dict key layout, same forward signature), so `instantiate_from_config` targets and public checkpoints
resolve unchanged.  The nn.Modules below only OWN parameters; forward() emits launches of libmgld_hip kernels on
token-major (NHWC) fp16 activations:
  * channel concatenation of the UNet skip connections never copies: producers write into slices of the consumer's
    concat buffer (leading-dimension addressing);
  * the time-embedding add is the conv epilogue; SPADE's gamma/beta convs are one GEMM and the modulation + residual
    is one elementwise kernel; nearest-2x upsampling is folded into the following conv's gather;
  * all 22+ emb_layers projections are ONE weight-streaming GEMV per step (they share SiLU(emb));
  * cross-attention K / V^T of the (constant) text context are computed once and cached.
"""

import os

import torch
import torch.nn as nn

from . import hip
from .engine import Act, Engine, pack_conv1x1, pack_conv3x3, pack_geglu, pack_tconv3


def _meta_module():
    return torch.device("meta")


class GroupNorm32(nn.GroupNorm):
    pass


def normalization(channels, norm_channel=32):
    return GroupNorm32(norm_channel, channels)


# ------------------------------------------------------------------------------------------------------------------
# parameter-owning blocks (names follow the reference so checkpoints load)
# ------------------------------------------------------------------------------------------------------------------
class SPADE(nn.Module):
    """ldm/modules/spade.py:68-111"""

    def __init__(self, norm_nc, label_nc):
        super().__init__()
        self.param_free_norm = normalization(norm_nc)
        nhidden = 128
        self.mlp_shared = nn.Sequential(nn.Conv2d(label_nc, nhidden, 3, padding=1), nn.ReLU())
        self.mlp_gamma = nn.Conv2d(nhidden, norm_nc, 3, padding=1)
        self.mlp_beta = nn.Conv2d(nhidden, norm_nc, 3, padding=1)


class Upsample(nn.Module):
    """openaimodel.py:160-188"""

    def __init__(self, channels, use_conv, dims=2, out_channels=None, padding=1):
        super().__init__()
        self.channels, self.out_channels = channels, out_channels or channels
        assert use_conv and dims == 2
        self.conv = nn.Conv2d(self.channels, self.out_channels, 3, padding=padding)

    def run(self, eng, x, out=None):
        # nearest-2x upsample folded into the conv gather: per-lane taps -> (tap, Cin) weight order
        w = eng.weight("c3up", (self.conv.weight,), lambda t: pack_conv3x3(t, tap_inner=False))
        b = eng.f32("b", self.conv.bias)
        return eng.conv3x3(x, w, b, self.out_channels, out=out, up2=True, lo=True)


class Downsample(nn.Module):
    """openaimodel.py:204-230"""

    def __init__(self, channels, use_conv, dims=2, out_channels=None, padding=1):
        super().__init__()
        self.channels, self.out_channels = channels, out_channels or channels
        assert use_conv and dims == 2
        self.op = nn.Conv2d(self.channels, self.out_channels, 3, stride=2, padding=padding)

    def run(self, eng, x, out=None):
        w = eng.weight("c3", (self.op.weight,), pack_conv3x3)
        b = eng.f32("b", self.op.bias)
        return eng.conv3x3(x, w, b, self.out_channels, out=out, stride=2, lo=True)


class ResBlock(nn.Module):
    """openaimodel.py:233-359 (no up/down, no scale-shift norm — the shipped configs use neither)."""

    def __init__(self, channels, emb_channels, dropout, out_channels=None, use_conv=False, use_scale_shift_norm=False,
                 dims=2, use_checkpoint=False, up=False, down=False):
        super().__init__()
        assert not (up or down or use_scale_shift_norm or use_conv) and dims == 2, "variant not on the MGLD-VSR hot path"
        self.channels, self.emb_channels = channels, emb_channels
        self.out_channels = out_channels or channels
        self.in_layers = nn.Sequential(normalization(channels), nn.SiLU(), nn.Conv2d(channels, self.out_channels, 3, padding=1))
        self.emb_layers = nn.Sequential(nn.SiLU(), nn.Linear(emb_channels, self.out_channels))
        self.out_layers = nn.Sequential(normalization(self.out_channels), nn.SiLU(), nn.Dropout(p=dropout),
                                        nn.Conv2d(self.out_channels, self.out_channels, 3, padding=1))
        if self.out_channels == channels:
            self.skip_connection = nn.Identity()
        else:
            self.skip_connection = nn.Conv2d(channels, self.out_channels, 1)
        self.emb_slot = None  # (offset into the fused emb projection), set by the owning network

    def _skip(self, eng, x):
        if isinstance(self.skip_connection, nn.Identity):
            return x
        w = eng.weight("c1", (self.skip_connection.weight,), pack_conv1x1)
        return eng.linear(x, w, eng.f32("b", self.skip_connection.bias), lo=True)

    def _trunk(self, eng, x, emb_all, emb_rpf):
        gn1, conv1 = self.in_layers[0], self.in_layers[2]
        t = eng.groupnorm(x, eng.f32("g", gn1.weight), eng.f32("b", gn1.bias), gn1.eps, True)
        rv = emb_all[:, self.emb_slot:self.emb_slot + self.out_channels]
        h = eng.conv3x3(t, eng.weight("c3", (conv1.weight,), pack_conv3x3), eng.f32("b", conv1.bias), self.out_channels,
                        rowvec=rv, rows_per_frame=emb_rpf, stats=True)      # feeds gn2: statistics from the convolution's epilogue
        gn2 = self.out_layers[0]
        return eng.groupnorm(h, eng.f32("g", gn2.weight), eng.f32("b", gn2.bias), gn2.eps, True)

    def run(self, eng, x, emb_all, emb_rpf, out=None):
        t = self._trunk(eng, x, emb_all, emb_rpf)
        conv2 = self.out_layers[3]
        skip = self._skip(eng, x)
        return eng.conv3x3(t, eng.weight("c3", (conv2.weight,), pack_conv3x3), eng.f32("b", conv2.bias), self.out_channels,
                           out=out, resid=skip, lo=True)


class ResBlockDual(ResBlock):
    """openaimodel.py:362-482: ResBlock whose output is SPADE-modulated by the structure condition."""

    def __init__(self, channels, emb_channels, dropout, semb_channels, out_channels=None, **kw):
        super().__init__(channels, emb_channels, dropout, out_channels=out_channels, **kw)
        self.spade = SPADE(self.out_channels, semb_channels)

    def run(self, eng, x, emb_all, emb_rpf, struct_cond, out=None):
        t = self._trunk(eng, x, emb_all, emb_rpf)
        conv2 = self.out_layers[3]
        h = eng.conv3x3(t, eng.weight("c3", (conv2.weight,), pack_conv3x3), eng.f32("b", conv2.bias), self.out_channels, stats=True)
        sp = self.spade
        self._sc_key = str(h.w)                      # which struct-cond scale this block reads (recorded for the hoisting pass)
        stats = eng.gn_stats(h, sp.param_free_norm.eps)
        skip = self._skip(eng, x)
        g, b = eng.f32("g", sp.param_free_norm.weight), eng.f32("b", sp.param_free_norm.bias)
        hoisted = struct_cond.get("__spade__", {}).get(id(self)) if isinstance(struct_cond, dict) else None
        if hoisted is not None:                       # gamma/beta of every step precomputed (ddpm._precompute_spade)
            table, stride, step_idx = hoisted
            return eng.spade_apply(h, stats, g, b, table[0], skip, out=out, step_idx=step_idx, step_stride=stride, want_stats=True)
        gb = self.spade_modulation(eng, struct_cond[self._sc_key])
        return eng.spade_apply(h, stats, g, b, gb, skip, out=out, want_stats=True)   # the transformer's norm reads this next

    def spade_modulation(self, eng, seg, out=None):
        """[gamma | beta] = conv(relu(conv(seg))) (spade.py:93-104): a function of the struct-cond features only; out: Act to write into
        (the hoisted tables: no copy)"""
        sp = self.spade
        actv = eng.conv3x3(seg, eng.weight("c3", (sp.mlp_shared[0].weight,), pack_conv3x3), eng.f32("b", sp.mlp_shared[0].bias),
                           128, act=hip.ACT_RELU)
        wgb = eng.weight("c3gb", (sp.mlp_gamma.weight, sp.mlp_beta.weight), lambda g, b: pack_conv3x3(torch.cat([g, b], 0)))
        bgb = eng.weight("bgb", (sp.mlp_gamma.bias, sp.mlp_beta.bias), lambda g, b: torch.cat([g, b], 0), torch.float32)
        return eng.conv3x3(actv, wgb, bgb, 2 * self.out_channels, out=out)


class QKVAttentionLegacy(nn.Module):
    def __init__(self, n_heads):
        super().__init__()
        self.n_heads = n_heads


class AttentionBlock(nn.Module):
    """openaimodel.py:485-531 + QKVAttentionLegacy 554-594 (channel order per head: [q | k | v])."""

    def __init__(self, channels, num_heads=1, num_head_channels=-1, use_checkpoint=False, use_new_attention_order=False):
        super().__init__()
        self.channels = channels
        self.num_heads = num_heads if num_head_channels == -1 else channels // num_head_channels
        self.norm = normalization(channels)
        self.qkv = nn.Conv1d(channels, channels * 3, 1)
        self.attention = QKVAttentionLegacy(self.num_heads)
        self.proj_out = nn.Conv1d(channels, channels, 1)

    def run(self, eng, x, out=None):
        C, H = self.channels, self.num_heads
        ch = C // H
        xn = eng.groupnorm(x, eng.f32("g", self.norm.weight), eng.f32("b", self.norm.bias), self.norm.eps, False)

        def split(w, b):
            w = w.reshape(H, 3, ch, C)
            b = b.reshape(H, 3, ch)
            wqk = torch.cat([w[:, 0].reshape(C, C), w[:, 1].reshape(C, C)], 0)
            bqk = torch.cat([b[:, 0].reshape(C), b[:, 1].reshape(C)], 0)
            return wqk, bqk, w[:, 2].reshape(C, C), b[:, 2].reshape(C)
        N = x.hw
        if QKV_FUSED:      # the conv1d qkv (per head [q | k | v] rows, openaimodel.py:515-519, 582-594) re-ordered to [q | k | v] x heads: one GEMM
            # round 6: the softmax scale (QKVAttentionLegacy scales q and k by ch^-1/4 each, :588-590) and log2(e) folded into the q rows of
            # weight AND bias in fp32, as the transformer's self-attention does: the software-pipelined kernel takes the launch
            ps = ch == 64
            f = ch ** -0.5 * LOG2E if ps else 1.0

            def fused(w, b):
                wqk_, bqk_, wv_, bv_ = split(w, b)
                wqk_, bqk_ = wqk_.clone(), bqk_.clone()
                wqk_[:C] *= f
                bqk_[:C] *= f
                return torch.cat([wqk_, wv_], 0), torch.cat([bqk_, bv_], 0)
            wqkv, bqkv = eng.weight("qkvps" if ps else "qkv", (self.qkv.weight, self.qkv.bias), fused)
            qkv = eng.linear(x=xn, w=wqkv, bias=bqkv)                 # [rows, 3C]: q | k | v, head-major inside each
            o = eng.act(x.n, x.h, x.w, C)
            hip.attention(qkv.v, qkv.v[:, C:], qkv.v[:, 2 * C:], o.v, batch=x.n, heads=H, Nq=N, Nkv=N, head_dim=ch,
                          q_strides=(N * 3 * C, 3 * C, ch), k_strides=(N * 3 * C, 3 * C, ch), vt_strides=(N * 3 * C, 3 * C, ch),
                          o_strides=(N * C, C, ch), scale=(1.0 / LOG2E) if ps else ch ** -0.5, v_rowmajor=True)
            eng.launches += 1
            wp = eng.weight("c1", (self.proj_out.weight,), pack_conv1x1)
            return eng.linear(o, wp, eng.f32("b", self.proj_out.bias), out=out, resid=x, lo=True)
        wqk, bqk = eng.weight("qk", (self.qkv.weight, self.qkv.bias), lambda w, b: split(w, b)[:2])
        wv, bv = eng.weight("v", (self.qkv.weight, self.qkv.bias), lambda w, b: split(w, b)[2:])
        qk = eng.linear(x=xn, w=wqk, bias=bqk)                      # [rows, 2C]: q | k, head-major
        Np = (N + 7) // 8 * 8                                       # key axis padded to 16-byte rows
        vt = eng.arena.alloc((x.n * C, Np), torch.float16)          # per frame [C, Np] = V^T
        hip.igemm(wv, xn.v, vt, bias_m=bv, M=C, N=N, K=C, batch=x.n, strideA=0, strideW=N * xn.v.stride(0), strideC=C * Np)
        o = eng.act(x.n, x.h, x.w, C)
        hip.attention(qk.v, qk.v[:, C:], vt, o.v, batch=x.n, heads=H, Nq=N, Nkv=N, head_dim=ch,
                      q_strides=(N * 2 * C, 2 * C, ch), k_strides=(N * 2 * C, 2 * C, ch), vt_strides=(C * Np, ch * Np, Np),
                      o_strides=(N * C, C, ch), scale=ch ** -0.5)
        eng.launches += 2
        wp = eng.weight("c1", (self.proj_out.weight,), pack_conv1x1)
        return eng.linear(o, wp, eng.f32("b", self.proj_out.bias), out=out, resid=x, lo=True)


class GEGLU(nn.Module):
    def __init__(self, dim_in, dim_out):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out * 2)


class FeedForward(nn.Module):
    def __init__(self, dim, mult=4, glu=True, dropout=0.):
        super().__init__()
        inner = int(dim * mult)
        self.net = nn.Sequential(GEGLU(dim, inner), nn.Dropout(dropout), nn.Linear(inner, dim))


class MemoryEfficientCrossAttention(nn.Module):
    """ldm/modules/attention.py:311-381 (parameter layout); computed by the flash kernel."""

    def __init__(self, query_dim, context_dim=None, heads=8, dim_head=64, dropout=0.0):
        super().__init__()
        inner = dim_head * heads
        context_dim = context_dim or query_dim
        self.heads, self.dim_head = heads, dim_head
        self.to_q = nn.Linear(query_dim, inner, bias=False)
        self.to_k = nn.Linear(context_dim, inner, bias=False)
        self.to_v = nn.Linear(context_dim, inner, bias=False)
        self.to_out = nn.Sequential(nn.Linear(inner, query_dim), nn.Dropout(dropout))


MemoryEfficientSelfAttention = MemoryEfficientCrossAttention


QKV_FUSED = os.environ.get("MGLD_QKV_FUSED", "1") != "0"    # q|k|v as ONE projection, V consumed row-major by the attention kernel
LOG2E = 1.4426950408889634


def _self_attention(eng, attn, norm, frames, N, resid, out=None):
    """resid + to_out(softmax(q k^T / sqrt(d)) v) over the N tokens of each frame, q / k / v = projections of norm(resid); resid: the
    residual-stream Act (-> the stream's next Act).  The LayerNorm is folded into the fused q|k|v projection where Engine.linear_ln can."""
    C, H, d = attn.heads * attn.dim_head, attn.heads, attn.dim_head
    if QKV_FUSED:
        # to_q | to_k | to_v (attention.py:323-330) as one GEMM with N = 3C: the normalised tokens are read once, and the attention
        # kernel takes V as it lies (row-major, transposing LDS reads) — no transposed V^T projection, one launch less per block
        # round 5: the softmax scale and log2(e) are folded into the q rows of the fused weight (fp32, before the one rounding to fp16), so
        # the scores leave the matrix pipe in log2 units and the attention kernel's probability is a bare exp2 (`scale` = ln 2 tells the
        # launcher: flash_attn_kernel<64, true, true, 1, true>, csrc/attention.hip); every other kernel variant computes the same softmax
        f = d ** -0.5 * LOG2E
        qkv = eng.linear_ln(resid, norm, "qkvps", (attn.to_q.weight, attn.to_k.weight, attn.to_v.weight),
                            lambda q, k, v: (torch.cat([q * f, k, v], 0), None))
        o = eng.empty(frames * N, C)
        hip.attention(qkv, qkv[:, C:], qkv[:, 2 * C:], o, batch=frames, heads=H, Nq=N, Nkv=N, head_dim=d,
                      q_strides=(N * 3 * C, 3 * C, d), k_strides=(N * 3 * C, 3 * C, d), vt_strides=(N * 3 * C, 3 * C, d),
                      o_strides=(N * C, C, d), scale=1.0 / LOG2E, v_rowmajor=True)
        eng.launches += 1
        wo = eng.weight("w", (attn.to_out[0].weight,), lambda w: w)
        return eng.linear(Act(o, resid.n, resid.h, resid.w), wo, eng.f32("b", attn.to_out[0].bias), out=out, resid=resid, lo=resid.lo is not None,
                          rowstats=True)       # feeds norm2
    xn = eng.layernorm(resid, eng.f32("g", norm.weight), eng.f32("b", norm.bias), norm.eps).v
    wqk = eng.weight("qk", (attn.to_q.weight, attn.to_k.weight), lambda q, k: torch.cat([q, k], 0))
    wv = eng.weight("w", (attn.to_v.weight,), lambda v: v)
    qk = eng.linear(xn, wqk, None)
    Np = (N + 7) // 8 * 8
    vt = eng.arena.alloc((frames * C, Np), torch.float16)
    hip.igemm(wv, xn, vt, M=C, N=N, K=xn.shape[1], batch=frames, strideA=0, strideW=N * xn.stride(0), strideC=C * Np)
    o = eng.empty(frames * N, C)
    hip.attention(qk, qk[:, C:], vt, o, batch=frames, heads=H, Nq=N, Nkv=N, head_dim=d,
                  q_strides=(N * 2 * C, 2 * C, d), k_strides=(N * 2 * C, 2 * C, d), vt_strides=(C * Np, d * Np, Np),
                  o_strides=(N * C, C, d), scale=d ** -0.5)
    eng.launches += 2
    wo = eng.weight("w", (attn.to_out[0].weight,), lambda w: w)
    return eng.linear(Act(o, resid.n, resid.h, resid.w), wo, eng.f32("b", attn.to_out[0].bias), out=out, resid=resid, lo=resid.lo is not None)


class ContextCache:
    """K and V^T projections of the (per-segment constant) text context, keyed by layer."""

    def __init__(self, eng, context):
        # context: fp32 device tensor [1, L, Dc] (the reference broadcasts a batch-1 context over frames,
        # attention.py:336-337)
        assert context.dim() == 3 and context.shape[0] == 1, "only a batch-1 context (broadcast over frames) is supported"
        self.eng = eng
        self.L = context.shape[1]
        self.Lp = (self.L + 7) // 8 * 8
        self.ctx16 = context[0].to(torch.float16).contiguous()
        self.kv = {}

    def get(self, attn):
        key = id(attn)
        if key not in self.kv:
            eng = self.eng
            C = attn.heads * attn.dim_head
            wk = eng.weight("w", (attn.to_k.weight,), lambda w: w)
            wv = eng.weight("w", (attn.to_v.weight,), lambda w: w)
            k = torch.empty(self.L, C, dtype=torch.float16, device=eng.device)
            vt = torch.zeros(C, self.Lp, dtype=torch.float16, device=eng.device)
            hip.igemm(self.ctx16, wk, k)
            hip.igemm(wv, self.ctx16, vt, M=C, N=self.L, K=self.ctx16.shape[1])
            self.kv[key] = (k, vt)
        return self.kv[key]


def _cross_attention(eng, attn, norm, frames, N, ctx_cache, resid):
    C, H, d = attn.heads * attn.dim_head, attn.heads, attn.dim_head
    q = eng.linear_ln(resid, norm, "w", (attn.to_q.weight,), lambda w: (w, None))
    k, vt = ctx_cache.get(attn)
    o = eng.empty(frames * N, C)
    hip.attention(q, k, vt, o, batch=frames, heads=H, Nq=N, Nkv=ctx_cache.L, head_dim=d, q_strides=(N * C, C, d),
                  k_strides=(0, C, d), vt_strides=(0, d * ctx_cache.Lp, ctx_cache.Lp), o_strides=(N * C, C, d),
                  scale=d ** -0.5)
    eng.launches += 1
    wo = eng.weight("w", (attn.to_out[0].weight,), lambda w: w)
    return eng.linear(Act(o, resid.n, resid.h, resid.w), wo, eng.f32("b", attn.to_out[0].bias), resid=resid, lo=resid.lo is not None,
                      rowstats=True)           # feeds norm3


class BasicTransformerBlockV2(nn.Module):
    """ldm/modules/attention.py:406-435 (xformers attention mode, disable_self_attn=False)."""

    def __init__(self, dim, n_heads, d_head, dropout=0., context_dim=None, gated_ff=True, checkpoint=True,
                 disable_self_attn=False):
        super().__init__()
        assert not disable_self_attn and gated_ff
        self.attn1 = MemoryEfficientCrossAttention(dim, None, n_heads, d_head, dropout)
        self.ff = FeedForward(dim, dropout=dropout, glu=True)
        self.attn2 = MemoryEfficientCrossAttention(dim, context_dim, n_heads, d_head, dropout)
        self.norm1, self.norm2, self.norm3 = nn.LayerNorm(dim), nn.LayerNorm(dim), nn.LayerNorm(dim)

    def run(self, eng, t, frames, N, ctx_cache):
        """t: the token stream as an Act (its low plane, when it has one, feeds the LayerNorms and the residual adds)"""
        t = _self_attention(eng, self.attn1, self.norm1, frames, N, resid=t)
        t = _cross_attention(eng, self.attn2, self.norm2, frames, N, ctx_cache, resid=t)
        proj = self.ff.net[0].proj
        g = eng.linear_ln(t, self.norm3, "geglu", (proj.weight, proj.bias), pack_geglu, act=hip.ACT_GEGLU)
        w2 = eng.weight("w", (self.ff.net[2].weight,), lambda w: w)
        return eng.linear(Act(g, t.n, t.h, t.w), w2, eng.f32("b", self.ff.net[2].bias), resid=t, lo=t.lo is not None)


class SpatialTransformerV2(nn.Module):
    """ldm/modules/attention.py:484-546 with use_linear=True, depth=1."""

    def __init__(self, in_channels, n_heads, d_head, depth=1, dropout=0., context_dim=None, disable_self_attn=False,
                 use_linear=False, use_checkpoint=False):
        super().__init__()
        assert use_linear and depth == 1, "the shipped config uses linear projections and depth 1"
        if context_dim is not None and not isinstance(context_dim, list):
            context_dim = [context_dim]
        inner = n_heads * d_head
        self.in_channels = in_channels
        self.norm = nn.GroupNorm(32, in_channels, eps=1e-6, affine=True)
        self.proj_in = nn.Linear(in_channels, inner)
        self.transformer_blocks = nn.ModuleList(
            [BasicTransformerBlockV2(inner, n_heads, d_head, dropout=dropout, context_dim=context_dim[d],
                                     disable_self_attn=disable_self_attn, checkpoint=use_checkpoint) for d in range(depth)])
        self.proj_out = nn.Linear(in_channels, inner)

    def run(self, eng, x, ctx_cache, out=None):
        xn = eng.groupnorm(x, eng.f32("g", self.norm.weight), eng.f32("b", self.norm.bias), self.norm.eps, False)
        t = eng.linear(xn, eng.weight("w", (self.proj_in.weight,), lambda w: w), eng.f32("b", self.proj_in.bias), lo=eng.lo_inner,
                       rowstats=True)          # feeds the block's norm1
        t = self.transformer_blocks[0].run(eng, t, x.n, x.hw, ctx_cache)
        return eng.linear(t, eng.weight("w", (self.proj_out.weight,), lambda w: w), eng.f32("b", self.proj_out.bias), out=out, resid=x, lo=True)


class SpatialTemporalConv(nn.Module):
    """ldm/modules/diffusionmodules/util.py:291-310"""

    def __init__(self, num_feat, num_frames=1):
        super().__init__()
        self.num_frames = num_frames
        self.temporal_conv = nn.Conv3d(num_feat, num_feat, (3, 1, 1), padding=(1, 0, 0))
        self.temporal_alpha = nn.Parameter(torch.zeros(1))

    def run(self, eng, x, out=None):
        w, w2 = eng.weight2("t3", (self.temporal_conv.weight,), pack_tconv3)
        return eng.tconv3(x, w, eng.f32("b", self.temporal_conv.bias), self.num_frames, float(self.temporal_alpha.detach()),
                          out=out, w2=w2)


class TemporalAttention(nn.Module):
    """ldm/modules/attention.py:124-143"""

    def __init__(self, num_feat, num_heads=8, dim_head=64, num_frames=1):
        super().__init__()
        self.num_frames = num_frames
        self.temporal_attn = MemoryEfficientSelfAttention(num_feat, heads=num_heads, dim_head=dim_head, dropout=0.0)
        self.norm = nn.LayerNorm(num_feat)
        self.temporal_alpha = nn.Parameter(torch.zeros(1))

    def run(self, eng, x, out=None):
        a, T = self.temporal_attn, self.num_frames
        C, H, d = a.heads * a.dim_head, a.heads, a.dim_head
        n = eng.layernorm(x, eng.f32("g", self.norm.weight), eng.f32("b", self.norm.bias), self.norm.eps)
        wqkv = eng.weight("qkv", (a.to_q.weight, a.to_k.weight, a.to_v.weight), lambda q, k, v: torch.cat([q, k, v], 0))
        qkv = eng.linear(n.v, wqkv, None)
        sh = eng.shard
        if sh is not None:
            # frame-sharded clip: every frame attends to all T frames -> all-gather the q|k|v rows (T*hw x 3C, a few MB at
            # the 8x8 level), run the (tiny) kernel on the whole clip, keep this rank's rows
            if x.n != sh.F:
                raise RuntimeError("sharded temporal attention expects one clip per segment")
            full = eng.arena.alloc((sh.T * x.hw, 3 * C), torch.float16)      # fixed address: the exchange of every replayed step lands here
            eng.collective(lambda: sh.all_gather(qkv, out=full))
            of = eng.empty(sh.T * x.hw, C)
            hip.temporal_attention(full[:, 0:C], full[:, C:2 * C], full[:, 2 * C:3 * C], of, sh.T, x.hw, H, d, d ** -0.5)
            eng.launches += 1
            o = of[sh.f0 * x.hw:sh.f1 * x.hw]
        else:
            o = eng.empty(x.rows, C)
            clips = x.n // T
            per = T * x.hw
            for c in range(clips):
                s = slice(c * per, (c + 1) * per)
                hip.temporal_attention(qkv[s, 0:C], qkv[s, C:2 * C], qkv[s, 2 * C:3 * C], o[s], T, x.hw, H, d, d ** -0.5)
                eng.launches += 1
        alpha = float(self.temporal_alpha.detach())
        wo = eng.weight("w", (a.to_out[0].weight,), lambda w: w)
        return eng.linear(Act(o, x.n, x.h, x.w), wo, eng.f32("b", a.to_out[0].bias), out=out, resid=x, alpha=alpha, beta=1.0 - alpha, lo=True)


class TimestepEmbedSequential(nn.Sequential):
    pass


# ------------------------------------------------------------------------------------------------------------------
# time embedding shared machinery
# ------------------------------------------------------------------------------------------------------------------
class _TimeEmbedMixin:
    def _register_emb_slots(self):
        off = 0
        self._emb_blocks = []
        for m in self.modules():
            if isinstance(m, ResBlock):
                m.emb_slot = off
                off += m.out_channels
                self._emb_blocks.append(m)
        self._emb_total = off

    def _time_embedding(self, eng, tvals):
        """tvals: fp32 device tensor [M] (M = 1 when all frames share the step). Returns fp32 [M, sum(Cout)] holding
        emb_layers[1](SiLU(time_embed(timestep_embedding(t)))) for every ResBlock, and the rows-per-row divisor."""
        M = tvals.shape[0]
        mc = self.model_channels
        te = eng.arena.alloc((M, mc), torch.float32)
        hip.timestep_embedding(tvals, te)
        l0, l2 = self.time_embed[0], self.time_embed[2]
        h1 = eng.arena.alloc((M, l0.out_features), torch.float32)
        hip.linear_small(te, eng.weight("w", (l0.weight,), lambda w: w), eng.f32("b", l0.bias), h1, silu_out=True)
        es = eng.arena.alloc((M, l2.out_features), torch.float32)
        hip.linear_small(h1, eng.weight("w", (l2.weight,), lambda w: w), eng.f32("b", l2.bias), es, silu_out=True)
        ws = tuple(b.emb_layers[1].weight for b in self._emb_blocks)
        bs = tuple(b.emb_layers[1].bias for b in self._emb_blocks)
        wcat = eng.weight("embcat", ws, lambda *w: torch.cat(w, 0))
        bcat = eng.weight("embcatb", bs, lambda *b: torch.cat(b, 0), torch.float32)
        out = eng.arena.alloc((M, self._emb_total), torch.float32)
        hip.linear_small(es, wcat, bcat, out)
        eng.launches += 4
        return out

    @staticmethod
    def _tvals(eng, timesteps, n_frames):
        """-> (fp32 device tvals [M], rows-per-embedding-row multiplier): M=1 if all timesteps are equal."""
        if isinstance(timesteps, torch.Tensor) and timesteps.is_cuda and timesteps.dtype == torch.float32 and timesteps.numel() == 1:
            return timesteps.reshape(1), None  # already a device scalar (graph-replayable path)
        t = torch.as_tensor(timesteps).reshape(-1)
        if t.numel() == 1 or bool((t == t[0]).all()):
            return t[:1].to(eng.device, torch.float32), None
        if t.numel() != n_frames or t.numel() > 16:
            raise RuntimeError("per-frame timesteps must have one entry per frame (<=16) or be uniform")
        return t.to(eng.device, torch.float32), 1


# ------------------------------------------------------------------------------------------------------------------
# networks
# ------------------------------------------------------------------------------------------------------------------
class InflatedUNetModelDualcondV2(nn.Module, _TimeEmbedMixin):
    """openaimodel.py:1903-2313.  forward(x, timesteps, context, struct_cond) -> eps, NCHW fp32 in/out."""

    def __init__(self, image_size, in_channels, model_channels, out_channels, num_res_blocks, attention_resolutions,
                 dropout=0, channel_mult=(1, 2, 4, 8), conv_resample=True, dims=2, num_frames=1, num_classes=None,
                 use_checkpoint=False, use_fp16=False, num_heads=-1, num_head_channels=-1, num_heads_upsample=-1,
                 use_scale_shift_norm=False, resblock_updown=False, use_new_attention_order=False,
                 use_spatial_transformer=False, transformer_depth=1, context_dim=None, n_embed=None, legacy=True,
                 disable_self_attentions=None, num_attention_blocks=None, disable_middle_self_attn=False,
                 use_linear_in_transformer=False, semb_channels=None):
        super().__init__()
        assert use_spatial_transformer and context_dim is not None and num_classes is None and dims == 2
        assert num_head_channels != -1 and not resblock_updown and not use_scale_shift_norm and n_embed is None
        assert isinstance(num_res_blocks, int)
        self.image_size, self.num_frames = image_size, num_frames
        self.in_channels, self.model_channels, self.out_channels = in_channels, model_channels, out_channels
        self.num_res_blocks, self.attention_resolutions = num_res_blocks, list(attention_resolutions)
        self.channel_mult, self.num_head_channels = list(channel_mult), num_head_channels
        self.dtype = torch.float16 if use_fp16 else torch.float32
        ted = model_channels * 4
        with _meta_module():
            self.time_embed = nn.Sequential(nn.Linear(model_channels, ted), nn.SiLU(), nn.Linear(ted, ted))
            self.input_blocks = nn.ModuleList([TimestepEmbedSequential(nn.Conv2d(in_channels, model_channels, 3, padding=1))])
            chans = [model_channels]
            ch, ds = model_channels, 1

            def st(c):
                return SpatialTransformerV2(c, c // num_head_channels, num_head_channels, depth=transformer_depth,
                                            context_dim=context_dim, use_linear=use_linear_in_transformer)
            for level, mult in enumerate(channel_mult):
                for _ in range(num_res_blocks):
                    layers = [ResBlockDual(ch, ted, dropout, semb_channels=semb_channels, out_channels=mult * model_channels)]
                    ch = mult * model_channels
                    if ds in attention_resolutions:
                        layers.append(st(ch))
                    self.input_blocks.append(TimestepEmbedSequential(*layers))
                    chans.append(ch)
                if level != len(channel_mult) - 1:
                    self.input_blocks.append(TimestepEmbedSequential(Downsample(ch, conv_resample, out_channels=ch)))
                    chans.append(ch)
                    ds *= 2
            heads = ch // num_head_channels
            self.middle_block = TimestepEmbedSequential(
                ResBlockDual(ch, ted, dropout, semb_channels=semb_channels),
                SpatialTemporalConv(ch, num_frames=num_frames),
                st(ch),
                TemporalAttention(ch, num_heads=heads, dim_head=num_head_channels, num_frames=num_frames),
                ResBlockDual(ch, ted, dropout, semb_channels=semb_channels),
                SpatialTemporalConv(ch, num_frames=num_frames))
            self.output_blocks = nn.ModuleList([])
            for level, mult in list(enumerate(channel_mult))[::-1]:
                for i in range(num_res_blocks + 1):
                    ich = chans.pop()
                    layers = [ResBlockDual(ch + ich, ted, dropout, semb_channels=semb_channels, out_channels=model_channels * mult)]
                    ch = model_channels * mult
                    if ds in attention_resolutions:
                        layers.append(st(ch))
                    if level and i == num_res_blocks:
                        layers.append(Upsample(ch, conv_resample, out_channels=ch))
                        ds //= 2
                    self.output_blocks.append(TimestepEmbedSequential(*layers))
            self.out = nn.Sequential(normalization(ch), nn.SiLU(), nn.Conv2d(model_channels, out_channels, 3, padding=1))
        self.to_empty(device="cpu")
        for p in self.parameters():
            p.data.zero_()
            p.requires_grad_(False)
        self._register_emb_slots()
        self._engine = None
        self._ctx_cache = None
        self._ctx_key = None

    # ---- engine plumbing ----
    def engine(self):
        if self._engine is None:
            self._engine = Engine()
        return self._engine

    def set_engine(self, eng):
        self._engine = eng

    def context_cache(self, eng, context):
        key = (context.data_ptr(), context._version, tuple(context.shape))
        if self._ctx_key != key:
            self._ctx_cache = ContextCache(eng, context.to(eng.device, torch.float32))
            self._ctx_key = key
        return self._ctx_cache

    def resblock_geometry(self, h, w):
        """[(ResBlockDual, height, width)] of every residual block for an h x w latent (host-side planning: table sizes)"""
        out, hh, ww = [], h, w
        for blk in self.input_blocks:
            for layer in blk:
                if isinstance(layer, Downsample):
                    hh, ww = hh // 2, ww // 2
                elif isinstance(layer, ResBlockDual):
                    out.append((layer, hh, ww))
        out += [(layer, hh, ww) for layer in self.middle_block if isinstance(layer, ResBlockDual)]
        for blk in self.output_blocks:
            for layer in blk:
                if isinstance(layer, ResBlockDual):
                    out.append((layer, hh, ww))
                elif isinstance(layer, Upsample):
                    hh, ww = 2 * hh, 2 * ww
        return out

    # ---- core: Acts in, Act out ----
    def run(self, eng, x, tvals, emb_rows, ctx_cache, struct_cond, out_eps=None):
        """x: Act [n,h,w,8] (latent channels zero-padded to 8); struct_cond: dict str(width)->Act; returns the fp32
        token-major eps buffer [n*h*w, 4] (ld = out_channels)."""
        with eng.scope("unet"):       # (engine.STREAM_LO_DEFAULT: the residual stream as two fp16 planes)
            return self._run(eng, x, tvals, emb_rows, ctx_cache, struct_cond, out_eps)

    def _run(self, eng, x, tvals, emb_rows, ctx_cache, struct_cond, out_eps):
        emb = self._time_embedding(eng, tvals)
        rpf = emb_rows if emb_rows is not None else None

        def erpf(a):
            return a.hw if rpf is not None else a.rows  # uniform timestep: every row maps to embedding row 0

        # geometry of the concat buffers: output block j consumes hs[-(j+1)]
        n_in = len(self.input_blocks)
        in_ch, in_hw = [], []
        h_, w_ = x.h, x.w
        for blk in self.input_blocks:
            first = blk[0]
            if isinstance(first, nn.Conv2d):
                c = first.out_channels
            elif isinstance(first, Downsample):
                c = first.out_channels
                h_, w_ = h_ // 2, w_ // 2
            else:
                c = first.out_channels
            in_ch.append(c)
            in_hw.append((h_, w_))
        # channels of h entering each output block
        ch_in = []
        cprev = self.middle_block[0].out_channels
        for blk in self.output_blocks:
            ch_in.append(cprev)
            cprev = blk[0].out_channels
        cats = []
        for j in range(len(self.output_blocks)):
            i = n_in - 1 - j
            hh, ww = in_hw[i]
            cats.append(eng.act(x.n, hh, ww, ch_in[j] + in_ch[i], lo=True))

        def skip_slot(i):  # where input block i writes its output
            j = n_in - 1 - i
            return cats[j].cols(ch_in[j], ch_in[j] + in_ch[i])

        def run_layers(h, layers, final_out):
            last = len(layers) - 1
            for li, layer in enumerate(layers):
                o = final_out if li == last else None
                if isinstance(layer, ResBlockDual):
                    h = layer.run(eng, h, emb, erpf(h), struct_cond, out=o)
                elif isinstance(layer, SpatialTransformerV2):
                    h = layer.run(eng, h, ctx_cache, out=o)
                elif isinstance(layer, (Downsample, Upsample, SpatialTemporalConv, TemporalAttention)):
                    h = layer.run(eng, h, out=o)
                elif isinstance(layer, nn.Conv2d):                     # input_blocks.0: the 4 -> 320 convolution on the noisy latent
                    with eng.scope("unet_io"):
                        wi, wi2 = eng.weight2("c3", (layer.weight,), lambda w: pack_conv3x3(w, h.C))
                        h = eng.conv3x3(h, wi, eng.f32("b", layer.bias), layer.out_channels, out=o, w2=wi2)
                else:
                    raise RuntimeError(f"unexpected layer {type(layer)}")
            return h

        h = x
        for i, blk in enumerate(self.input_blocks):
            h = run_layers(h, list(blk), skip_slot(i))
        # middle block writes the h half of cats[0]
        h = run_layers(h, list(self.middle_block), cats[0].cols(0, ch_in[0]))
        for j, blk in enumerate(self.output_blocks):
            nxt = cats[j + 1].cols(0, ch_in[j + 1]) if j + 1 < len(self.output_blocks) else None
            h = run_layers(cats[j], list(blk), nxt)
        gn, conv = self.out[0], self.out[2]
        t = eng.groupnorm(h, eng.f32("g", gn.weight), eng.f32("b", gn.bias), gn.eps, True)
        if out_eps is None:
            out_eps = Act(eng.arena.alloc((x.rows, self.out_channels), torch.float32), x.n, x.h, x.w)
        with eng.scope("unet_io"):                                              # out: the 320 -> 4 convolution that produces eps
            wo, wo2 = eng.weight2("c3", (conv.weight,), pack_conv3x3)
            eng.conv3x3(t, wo, eng.f32("b", conv.bias), self.out_channels, out=out_eps, w2=wo2)
        return out_eps

    # ---- reference-compatible entry ----
    @torch.no_grad()
    def forward(self, x, timesteps=None, context=None, struct_cond=None, y=None, **kwargs):
        assert y is None
        eng = self.engine()
        eng.reset()
        x = x.to(eng.device, torch.float32)
        xa = eng.from_nchw(x)
        tv, rows = self._tvals(eng, timesteps, x.shape[0])
        sc = {}
        for k, v in struct_cond.items():
            sc[k] = v if isinstance(v, Act) else eng.from_nchw(v.to(eng.device, torch.float32))
        eps = self.run(eng, xa, tv, rows, self.context_cache(eng, context), sc)
        return eng.to_nchw(eps, self.out_channels)


class InflatedEncoderUNetModelWT(nn.Module, _TimeEmbedMixin):
    """openaimodel.py:2316-2525.  forward(x, timesteps) -> {str(width): [n, out_channels, r, r]} NCHW fp32."""

    def __init__(self, image_size, in_channels, model_channels, out_channels, num_res_blocks, attention_resolutions,
                 dropout=0, channel_mult=(1, 2, 4, 8), conv_resample=True, dims=2, num_frames=1, use_checkpoint=False,
                 use_fp16=False, num_heads=1, num_head_channels=-1, num_heads_upsample=-1, use_scale_shift_norm=False,
                 resblock_updown=False, use_new_attention_order=False, *args, **kwargs):
        super().__init__()
        assert dims == 2 and not resblock_updown and not use_scale_shift_norm and not use_new_attention_order
        self.in_channels, self.model_channels, self.out_channels = in_channels, model_channels, out_channels
        self.num_res_blocks, self.attention_resolutions = num_res_blocks, list(attention_resolutions)
        self.channel_mult = list(channel_mult)
        self.dtype = torch.float16 if use_fp16 else torch.float32
        self.num_heads, self.num_head_channels = num_heads, num_head_channels
        ted = model_channels * 4
        with _meta_module():
            self.time_embed = nn.Sequential(nn.Linear(model_channels, ted), nn.SiLU(), nn.Linear(ted, ted))
            self.input_blocks = nn.ModuleList([TimestepEmbedSequential(nn.Conv2d(in_channels, model_channels, 3, padding=1))])
            chans = []
            ch, ds = model_channels, 1
            for level, mult in enumerate(channel_mult):
                for _ in range(num_res_blocks):
                    layers = [ResBlock(ch, ted, dropout, out_channels=mult * model_channels)]
                    ch = mult * model_channels
                    if ds in attention_resolutions:
                        layers.append(AttentionBlock(ch, num_heads=num_heads, num_head_channels=num_head_channels))
                    self.input_blocks.append(TimestepEmbedSequential(*layers))
                if level != len(channel_mult) - 1:
                    self.input_blocks.append(TimestepEmbedSequential(Downsample(ch, conv_resample, out_channels=ch)))
                    chans.append(ch)
                    ds *= 2
            self.middle_block = TimestepEmbedSequential(
                ResBlock(ch, ted, dropout),
                AttentionBlock(ch, num_heads=num_heads, num_head_channels=num_head_channels),
                ResBlock(ch, ted, dropout))
            chans.append(ch)
            self.input_block_chans = chans
            self.fea_tran = nn.ModuleList([ResBlock(c, ted, dropout, out_channels=out_channels) for c in chans])
        self.to_empty(device="cpu")
        for p in self.parameters():
            p.data.zero_()
            p.requires_grad_(False)
        self._register_emb_slots()
        self._engine = None

    def engine(self):
        if self._engine is None:
            self._engine = Engine()
        return self._engine

    def set_engine(self, eng):
        self._engine = eng

    def run(self, eng, x, tvals, emb_rows):
        """x: Act [n,h,w,8] -> dict str(width) -> Act [n,r,r,out_channels]."""
        with eng.scope("struct"):
            return self._run(eng, x, tvals, emb_rows)

    def _run(self, eng, x, tvals, emb_rows):
        emb = self._time_embedding(eng, tvals)

        def erpf(a):   # rows sharing one embedding row: all (None), one frame (1), or emb_rows consecutive frames
            return a.hw * int(emb_rows) if emb_rows is not None else a.rows
        results = []
        h = x
        for blk in self.input_blocks:
            last = h
            for layer in blk:
                if isinstance(layer, nn.Conv2d):
                    h = eng.conv3x3(h, eng.weight("c3", (layer.weight,), lambda w: pack_conv3x3(w, h.C)), eng.f32("b", layer.bias),
                                    layer.out_channels, lo=True)
                elif isinstance(layer, ResBlock):
                    h = layer.run(eng, h, emb, erpf(h))
                elif isinstance(layer, (AttentionBlock, Downsample)):
                    h = layer.run(eng, h)
            if h.w != last.w:
                results.append(last)
        h = self.middle_block[0].run(eng, h, emb, erpf(h))
        h = self.middle_block[1].run(eng, h)
        h = self.middle_block[2].run(eng, h, emb, erpf(h))
        results.append(h)
        assert len(results) == len(self.fea_tran)
        return {str(r.w): self.fea_tran[i].run(eng, r, emb, erpf(r)) for i, r in enumerate(results)}

    @torch.no_grad()
    def forward(self, x, timesteps):
        eng = self.engine()
        eng.reset()
        x = x.to(eng.device, torch.float32)
        tv, rows = self._tvals(eng, timesteps, x.shape[0])
        res = self.run(eng, eng.from_nchw(x), tv, rows)
        return {k: eng.to_nchw(v) for k, v in res.items()}
