"""Host-side execution engine: weight packing for the kernel layouts, a bump-allocated activation arena,
and launch recording (eager first run -> replayable launch list -> hipGraph).

The reference drives one CUDA stream from a Python loop through nn.Module.forward.  Here the
nn.Modules only own parameters (checkpoint-compatible names); their forward() emits C-ABI launches on token-major
(NHWC) fp16 buffers.
"""
import os

import torch

from . import hip


# ------------------------------------------------------------------------------------------------------------------
# weight packing (done once, on the host, in fp32 -> fp16)
# ------------------------------------------------------------------------------------------------------------------
def conv_tap_inner(cin_pad, up2=False):
    """K-axis order the igemm uses for a 3x3 conv: True = (64-channel block, tap, channel) — all 9 taps of one channel
    block back to back so the shifted re-reads of the input hit L1/L2; False = (tap, Cin) (any Cin % 8, upsample fold)."""
    return (cin_pad % 64 == 0) and not up2


def pack_conv3x3(w, cin_pad=None, tap_inner=None):
    """[Cout, Cin, 3, 3] -> [Cout, 9*Cin_pad].  K index = (ky*3+kx)*Cin_pad + c, or, when tap_inner (default: whenever
    Cin_pad % 64 == 0), K index = ((c // 64) * 9 + (ky*3+kx)) * 64 + c % 64."""
    cout, cin, kh, kw = w.shape
    assert kh == 3 and kw == 3
    cp = cin_pad or ((cin + 7) // 8 * 8)
    if tap_inner is None:
        tap_inner = conv_tap_inner(cp)
    out = torch.zeros(cout, 9, cp, dtype=w.dtype)
    out[:, :, :cin] = w.permute(0, 2, 3, 1).reshape(cout, 9, cin)
    if tap_inner:
        assert cp % 64 == 0
        out = out.reshape(cout, 9, cp // 64, 64).permute(0, 2, 1, 3)
    return out.reshape(cout, 9 * cp).contiguous()


def tile_conv3p(wp, cin, tap_inner):
    """(host / test form; the engine re-lays weights on the device with the kernel of the same layout, hip.tile_conv3p)
    [N, 9*Cin] packed conv weights (either K order of pack_conv3x3) -> the patch kernel's tiled layout
    [N64/64][Cin/32][3 dy][4 row groups][3 dx][16 rows][32 ch] as a [., 32] tensor (include/mgld_hip.h, tap_inner = 2); rows
    padded with zeros to a multiple of 64."""
    n = wp.shape[0]
    if tap_inner:
        w3 = wp.reshape(n, cin // 64, 9, 64).permute(0, 2, 1, 3).reshape(n, 9, cin)
    else:
        w3 = wp.reshape(n, 9, cin)
    n64 = (n + 63) // 64 * 64
    if n64 != n:
        w3 = torch.cat([w3, w3.new_zeros(n64 - n, 9, cin)], 0)
    # [g64, rb 4, row 16, dy 3, dx 3, slice, chunk 4, 8] -> [g64, slice, dy, rb, dx, row, chunk, 8]
    t = w3.reshape(n64 // 64, 4, 16, 3, 3, cin // 32, 4, 8).permute(0, 5, 3, 1, 4, 2, 6, 7).contiguous()
    # LDS-image order: 16-B slot c of tile row r holds logical chunk c ^ ((r >> 2) & 3) (the kernel's bank swizzle), so a DMA
    # piece is read lane-linearly
    rows = torch.arange(16, device=t.device)
    src = torch.arange(4, device=t.device)[None, :] ^ ((rows >> 2) & 3)[:, None]                 # [16, 4]
    t = torch.gather(t, 6, src[None, None, None, None, None, :, :, None].expand(*t.shape))
    return t.reshape(-1, 32)


def pack_conv(w, cin_pad=None):
    """[Cout, Cin, kh, kw] -> [Cout, kh*kw*Cin_pad], K index = (ky*kw+kx)*Cin_pad + c (general-tap igemm layout)."""
    cout, cin, kh, kw = w.shape
    cp = cin_pad or ((cin + 7) // 8 * 8)
    out = torch.zeros(cout, kh * kw, cp, dtype=w.dtype)
    out[:, :, :cin] = w.permute(0, 2, 3, 1).reshape(cout, kh * kw, cin)
    return out.reshape(cout, kh * kw * cp).contiguous()


def pack_conv1x1(w, cin_pad=None):
    """[Cout, Cin, 1, 1] or [Cout, Cin, 1] or [Cout, Cin] -> [Cout, Cin_pad]."""
    cout, cin = w.shape[0], w.shape[1]
    cp = cin_pad or ((cin + 7) // 8 * 8)
    out = torch.zeros(cout, cp, dtype=w.dtype)
    out[:, :cin] = w.reshape(cout, cin)
    return out.contiguous()


def pack_tconv3(w):
    """Conv3d weight [C, C, 3, 1, 1] -> [C, 3*C] with K index = dt*C + c."""
    c_out, c_in = w.shape[0], w.shape[1]
    return w[:, :, :, 0, 0].permute(0, 2, 1).reshape(c_out, 3 * c_in).contiguous()


def pack_geglu(w, b):
    """GEGLU projection [2*inner, dim]: interleave value/gate rows in blocks of 32 so that the igemm epilogue finds
    value j and gate j in the same lane (rows [64g, 64g+32) = values 32g.., rows [64g+32, 64g+64) = gates 32g..)."""
    two_inner, dim = w.shape
    inner = two_inner // 2
    assert inner % 32 == 0
    g = inner // 32
    wv, wg = w[:inner].reshape(g, 32, dim), w[inner:].reshape(g, 32, dim)
    wp = torch.stack([wv, wg], 1).reshape(two_inner, dim).contiguous()
    bp = None
    if b is not None:
        bp = torch.stack([b[:inner].reshape(g, 32), b[inner:].reshape(g, 32)], 1).reshape(two_inner).contiguous()
    return wp, bp


def pack_ln_fold(W, bias, gamma, beta):
    """LayerNorm(x; gamma, beta) W^T + bias as a contraction on the RAW rows (MgldIGemm.ln_part): -> (W' = W diag(gamma), s, b') with
    y[m, n] = rstd[m] * (x[m] . W'[n] - mean[m] * s[n]) + b'[n];  s = row sums of the fp16-ROUNDED W' (what the MFMA actually sums, so the
    mean correction cancels it exactly), b' = W beta + bias.  attention.py:427-435."""
    Wp = W * gamma[None, :]
    s = Wp.to(torch.float16).to(torch.float32).sum(1)
    b2 = W @ beta + (bias if bias is not None else 0.0)
    return Wp, s, b2


def pack_hp(w32):
    """split-fp16 weights of the high-precision encoder (csrc/hpenc.hip): [Cout, Cin, ...] fp32 -> [Cout, 3*Cin, ...] =
    [wh | wh / 16 | 256 wl] along the input-channel axis (wh = fp16(w), wl = w - wh), every entry exactly representable in fp16 (wl to
    2^-11 of itself); meets activation rows [ah | 16 al | ah / 256], so one ordinary contraction over 3 Cin channels yields
    ah wh + al wh + ah wl."""
    wh = w32.to(torch.float16).to(torch.float32)
    wl = w32 - wh
    return torch.cat([wh, wh / 16.0, wl * 256.0], 1)


def split_residual(w32, scale=None):
    """fp32 packed weights -> (hi, lo): hi = fp16(w), lo = fp16((w - hi) / scale) with scale = 2^-11, i.e. the rounding residual of
    the weights lifted into fp16's normal range (MgldIGemm.W2: the kernel computes scale * (A lo^T) + A hi^T)."""
    scale = hip.W2_SCALE if scale is None else scale
    hi = w32.to(torch.float16)
    lo = ((w32 - hi.to(torch.float32)) / scale).to(torch.float16)
    return hi, lo


# where the second MFMA pass on the weight residual is switched on (env MGLD_W2 = comma-separated scope names, "all", or "0"):
#   vae_dec   every contraction of the video decoder (VideoDecoder_Mix incl. fusion layers, temporal convs, post_quant_conv)
#   vae_enc   the video VAE's encoder (its features feed the decoder's fusion layers)
#   first     the first-stage (image) encoder that produces the init latent / struct-cond input
#   unet_io   the UNet's input_blocks.0 and out convolutions
#   unet      every contraction of the UNet and the struct-cond encoder
#   vae_dec_mid / vae_dec_up<0..3> / vae_dec_fuse / vae_dec_out   parts of the video decoder (level 0 = 512^2)
# Default = the video decoder's mid block, 64^2 / 128^2 / 512^2 levels and conv_out + the UNet's first / last convolution.  Measured per
# scope on the full-width workloads (profiles/r03_w2_scopes_*): 8 x 512^2 frames -6 % / latent -1.7 % for ~+2 % segment time; BASELINE
# configs[0] (one frame, the worst case: decoder error 2.0e-3) frames 1.031e-3 -> 0.986e-3.  Adding the 256^2 level (up1) gives 0.968e-3
# for another +1.5 % time, the fusion layers cost +1.9 % for -0.3 % / -1.2 %: not taken.
W2_DEFAULT = "vae_dec_mid,vae_dec_up0,vae_dec_up2,vae_dec_up3,vae_dec_out,unet_io"


def w2_scopes():
    v = os.environ.get("MGLD_W2", W2_DEFAULT).strip()
    if v in ("", "0", "none", "off"):
        return frozenset()
    return frozenset(t.strip() for t in v.split(",") if t.strip())


# where the RESIDUAL STREAM is kept as two fp16 planes (env MGLD_STREAM_LO = comma-separated scope names, "all", or "0"):
#   unet      the UNet: every `x + f(x)` a block hands to the next one (ResBlockDual, the transformer's three residual adds, proj_in / proj_out,
#             the temporal mixes, down / upsample convolutions, the skip concatenations)
#   struct    the struct-cond encoder (hoisted out of the loop: five batched passes per segment)
#   vae_dec   the video decoder (ResnetBlocks, mid attention, temporal mixes, fusion layers, upsample convolutions)
#   vae_enc   the video VAE's fp16 encoder (features for the decoder's fusion layers)
# value = hi + 2^-11 lo: hi is the fp16 tensor every contraction keeps reading as its operand, lo = fp16((x - hi) 2^11) is read by the
# normalisations and the residual adds only (Act.lo, MgldIGemm.Rlo / Clo, mgld_*_lo).  What it removes is the rounding of the stream itself,
# the largest single term of a network evaluation's error (tests/analysis/resid_sim.py: UNet 1.91e-3 -> 1.57e-3, video decoder 2.26 -> 1.70).
STREAM_LO_DEFAULT = "unet,struct,vae_dec,vae_enc"


def stream_lo_scopes():
    v = os.environ.get("MGLD_STREAM_LO", STREAM_LO_DEFAULT).strip()
    if v in ("", "0", "none", "off"):
        return frozenset()
    return frozenset(t.strip() for t in v.split(",") if t.strip())


# ------------------------------------------------------------------------------------------------------------------
# activations + arena
# ------------------------------------------------------------------------------------------------------------------
class Act:
    """Token-major (NHWC) activation: `v` is a 2-D fp16 view [n*h*w, C] with row stride ld >= C."""
    __slots__ = ("v", "n", "h", "w", "stats", "lo", "rstats")

    def __init__(self, v, n, h, w, lo=None):
        assert v.dim() == 2 and v.shape[0] == n * h * w, (v.shape, n, h, w)
        self.v, self.n, self.h, self.w = v, n, h, w
        # low plane of a residual-stream tensor (same shape and row stride as v), or None: value = v + 2^-11 lo (stream_lo_scopes).  Kernels
        # that take the tensor as a contraction operand read v alone; normalisations and residual adds read both; whoever writes v writes lo.
        assert lo is None or (lo.shape == v.shape and lo.stride() == v.stride() and lo.dtype == v.dtype)
        self.lo = lo
        # per-ROW (sum, sumsq) over all channels of this tensor, written by the projection that produced it (Engine.linear(rowstats=True)):
        # (float32 [chunks, rows, 2], chunks) — what a LayerNorm folded into the consuming projection needs (Engine.linear_ln); None otherwise
        self.rstats = None
        # GroupNorm statistics of THIS tensor written by the kernel that produced it (Engine.conv3x3(stats=True), spade_apply(want_stats=True)):
        # (sums tensor, hip.GN_* kind, chunks per frame, groups or None); Engine.gn_stats() then launches nothing.  Any op that writes into
        # an existing Act (`out=`) clears it first.
        self.stats = None

    @property
    def C(self):
        return self.v.shape[1]

    @property
    def rows(self):
        return self.v.shape[0]

    @property
    def hw(self):
        return self.h * self.w

    def cols(self, c0, c1, lo=True):
        """a channel window; lo=False: without the low plane (columns that do not belong to the residual stream)"""
        return Act(self.v[:, c0:c1], self.n, self.h, self.w, None if (self.lo is None or not lo) else self.lo[:, c0:c1])


class Arena:
    """Bump allocator over large device chunks.  reset() rewinds; identical call sequences therefore return
    identical pointers, which is what makes a captured hipGraph of the launch sequence replayable."""

    def __init__(self, device, chunk_bytes=1 << 30):
        self.device = device
        self.chunk_bytes = chunk_bytes
        self.chunks = []
        self.ci = 0
        self.off = 0
        self.frozen = False

    def reset(self):
        self.ci, self.off = 0, 0

    def mark(self):
        """the bump position, for release(): a callee that shares this arena (the flow network inside a sampler call) allocates past what its
        caller holds and hands the space back on return, instead of rewinding to 0 over the caller's buffers"""
        return (self.ci, self.off)

    def release(self, mark):
        self.ci, self.off = mark

    def alloc(self, shape, dtype):
        nbytes = int(torch.tensor([], dtype=dtype).element_size())
        for s in shape:
            nbytes *= int(s)
        nbytes = (nbytes + 255) & ~255
        while True:
            if self.ci >= len(self.chunks):
                if self.frozen:
                    raise RuntimeError("arena grew after a graph was captured")
                self.chunks.append(torch.empty(max(self.chunk_bytes, nbytes), dtype=torch.uint8, device=self.device))
                self.off = 0
            ch = self.chunks[self.ci]
            if self.off + nbytes <= ch.numel():
                t = ch[self.off:self.off + nbytes].view(dtype)
                self.off += nbytes
                n = 1
                for s in shape:
                    n *= int(s)
                return t[:n].view(*shape)
            self.ci += 1
            self.off = 0

    def bytes_reserved(self):
        return sum(c.numel() for c in self.chunks)


class GraphPieces:
    """A sharded step as hipGraph PIECES with the collectives between them (frame / tile sharding: a collective cannot sit
    inside a captured graph, and ~550 eager launches per step make a step host-bound once a rank holds one frame).  Recording
    pass = the first step after the eager one: every `Engine.collective(fn)` ends the running capture, launches that piece (so
    the exchange sees real data), runs `fn`, and starts the next capture; later steps replay the list [graph, fn, graph, ...].
    Legal for the same reason the single graph is: the arena hands out identical pointers for identical call sequences, and the
    exchanges write into caller-owned buffers of fixed address."""

    def __init__(self):
        self.items, self.cur = [], None

    def begin(self):
        self.cur = hip.Graph()
        self.cur.begin()

    def _close(self):
        g, self.cur = self.cur, None
        g.end()
        g.launch()
        self.items.append(g)

    def collective(self, fn):
        self._close()
        fn()
        self.items.append(fn)
        self.begin()

    def finish(self):
        self._close()

    def abort(self):
        if self.cur is not None:
            try:
                self.cur.end()
            finally:
                self.cur = None

    def replay(self):
        for it in self.items:
            if isinstance(it, hip.Graph):
                it.launch()
            else:
                it()

    @property
    def n_graphs(self):
        return sum(1 for it in self.items if isinstance(it, hip.Graph))


class Engine:
    """Owns the arena, the packed-weight cache and the op helpers every network forward is written in."""

    GROUPS = 32
    GN_FUSED = os.environ.get("MGLD_GN_FUSED", "1") != "0"    # single-launch GroupNorm for frames of <= 256 rows
    GN_PRODUCER = os.environ.get("MGLD_GN_PRODUCER", "1") != "0"   # GroupNorm statistics written by the kernel that produces the tensor
    LN_FOLD = os.environ.get("MGLD_LN_FOLD", "1") != "0"           # LayerNorm folded into the projection that consumes it (linear_ln)
    # ... also into the GEGLU projection: measured SLOWER (13.67 against 13.93 frames/s; unfolded 13.84): that epilogue is VALU-bound already and
    # the kernel is 8 % of a segment — off by default, the LayerNorm in front of the feed-forward stays a launch (profiles/r06_ln_fold.md)
    LN_FOLD_GEGLU = os.environ.get("MGLD_LN_FOLD_GEGLU", "0") != "0"

    def __init__(self, device="cuda", chunk_bytes=1 << 30, workspace_bytes=256 << 20):
        hip.lib()  # fail loudly if the HIP library is missing
        if not torch.cuda.is_available():
            raise RuntimeError("mgld_vsr_amd needs a HIP device: the hot path has no CPU fallback")
        self.device = torch.device(device)
        self.arena = Arena(self.device, chunk_bytes)
        self._wcache = {}
        self._c3p_geo, self._c3p_w = {}, {}     # patch-conv applicability per geometry / tiled weights per packed tensor
        self.launches = 0
        self.gn_stats_saved = 0   # GroupNorm statistics launches that a producer's epilogue replaced
        self.ln_folded = 0        # LayerNorm launches folded into their consumer (linear_ln)
        self.shard = None   # parallel.FrameShard when the frames of one segment are split over ranks
        self.tile_shard = None   # parallel.TileShard when the latent tiles of aggregation sampling are split over ranks
        self.pieces = None       # GraphPieces while a sharded step is recorded / replayed as hipGraph pieces
        self.w2_scopes = w2_scopes()   # precision scopes that run the second MFMA pass on the weight residual (split_residual)
        self.lo_scopes = stream_lo_scopes()   # scopes that keep the residual stream as two fp16 planes (Act.lo)
        # measured choices (profiles/r06_stream_lo.md): the normalisations read both planes (hi alone: +6 % latent error for 1.3 % of time);
        # the token stream INSIDE a transformer block (proj_in .. ff) stays one plane — its five projections per block are HBM-bound at the
        # 64^2 level and the low plane there cost 3.2 % of a segment for 3 % of the latent's error
        self.lo_norms = os.environ.get("MGLD_STREAM_LO_NORMS", "1") != "0"
        self.lo_inner = os.environ.get("MGLD_STREAM_LO_INNER", "0") != "0"   # (A/B: inside the UNet, low planes only for frames of at most this many pixels)
        self._scope = []               # stack of active scope names (Engine.scope)
        # split-K scratch of the igemm launcher (fp32 partials of the low-resolution, deep-K convolutions)
        self._splitk_ws = hip.ensure_workspace(workspace_bytes) if self.device.type == "cuda" else None

    # ---- memory ----
    def reset(self):
        self.arena.reset()

    def collective(self, fn):
        """run an inter-rank exchange: directly, or as a break between two graph pieces while a sharded step is being recorded"""
        if self.pieces is not None and self.pieces.cur is not None:
            self.pieces.collective(fn)
        else:
            fn()

    def empty(self, rows, C, dtype=torch.float16):
        return self.arena.alloc((rows, C), dtype)

    def act(self, n, h, w, C, dtype=torch.float16, lo=False):
        """lo: a residual-stream tensor — it gets a low plane when the active scope keeps the stream as two planes (lo_on)"""
        a = Act(self.arena.alloc((n * h * w, C), dtype), n, h, w)
        if lo and self.lo_on:
            a.lo = self.arena.alloc((n * h * w, C), dtype)
        return a

    # ---- packed weights (cached per (tag, parameter identity, version)) ----
    def weight(self, tag, params, fn, dtype=torch.float16):
        """params: tuple of parameters/tensors; fn(*cpu fp32 tensors) -> tensor (or tuple of tensors)."""
        key = (tag,) + tuple((id(p), p._version) for p in params if p is not None)
        hit = self._wcache.get(key)
        if hit is not None:
            return hit
        with torch.no_grad():
            out = fn(*[None if p is None else p.detach().float().cpu() for p in params])
        if isinstance(out, (tuple, list)):
            dev = tuple(None if o is None else o.to(self.device, dtype if o.dtype.is_floating_point and i == 0 else torch.float32).contiguous()
                        for i, o in enumerate(out))
        else:
            dev = out.to(self.device, dtype).contiguous()
        self._wcache[key] = dev
        return dev

    # ---- precision scopes ----
    class _Scope:
        def __init__(self, eng, name):
            self.eng, self.name = eng, name

        def __enter__(self):
            self.eng._scope.append(self.name)

        def __exit__(self, *exc):
            self.eng._scope.pop()
            return False

    def scope(self, name):
        """`with eng.scope("vae_dec"):` — contractions launched inside use split weights when MGLD_W2 names the scope"""
        return Engine._Scope(self, name)

    @property
    def w2_on(self):
        s = self.w2_scopes
        return bool(s) and ("all" in s or any(n in s for n in self._scope))

    @property
    def lo_on(self):
        s = self.lo_scopes
        return bool(s) and ("all" in s or any(n in s for n in self._scope))

    def weight2(self, tag, params, fn):
        """like weight(), for a contraction that may run the residual pass: -> (hi, lo-or-None) fp16 device tensors; fn must return
        ONE fp32 matrix.  lo is built (once) only while a scope that MGLD_W2 names is active."""
        hi = self.weight(tag, params, fn)
        if not self.w2_on:
            return hi, None
        key = (tag + "#w2",) + tuple((id(p), p._version) for p in params if p is not None)
        lo = self._wcache.get(key)
        if lo is None:
            with torch.no_grad():
                w32 = fn(*[None if p is None else p.detach().float().cpu() for p in params])
            lo = self._wcache[key] = split_residual(w32)[1].to(self.device).contiguous()
        return hi, lo

    def const(self, key, fn):
        """device constant built once per engine (e.g. identity affine vectors): fn() -> tensor or tuple of tensors"""
        hit = self._wcache.get(("const", key))
        if hit is None:
            hit = self._wcache[("const", key)] = fn()
        return hit

    def f32(self, tag, p):
        return self.weight(tag, (p,), lambda t: t, torch.float32)

    # ---- ops ----
    def conv3x3(self, x, wp, bias, cout, out=None, stride=1, pad=(1, 1), up2=False, hw_out=None, rowvec=None,
                rows_per_frame=0, resid=None, act=hip.ACT_NONE, alpha=1.0, beta=1.0, out_dtype=torch.float16, w2=None, stats=False, lo=False):
        """stats: the output feeds a GroupNorm over all of its channels — where the kernel the launcher picks can, it also writes the
        per-tile channel sums of what it stores (MgldIGemm.gn_part) and the returned Act carries them (Act.stats).
        lo: the output is a residual-stream tensor (Engine.act(lo=True)); an `out` / `resid` that carries a low plane has it written / added."""
        hin, win = x.h, x.w
        if hw_out is None:
            hv, wv = (2 * hin, 2 * win) if up2 else (hin, win)
            hw_out = (hv, wv) if stride == 1 else (hv // 2, wv // 2)
        ho, wo = hw_out
        if out is None:
            out = self.act(x.n, ho, wo, cout, out_dtype, lo=lo and out_dtype == torch.float16)
        cin = x.C
        assert wp.shape[1] == 9 * cin, (wp.shape, cin)
        tap_inner = 1 if conv_tap_inner(cin, up2) else 0          # must match pack_conv3x3(..., tap_inner) of wp
        kw = {}
        # (MGLD_CONV3Q=0, the documented A/B switch, leaves only the raster patch kernel for tiled weights, and that one does not take
        # the weight-residual pass: such layers keep the [N, K] layout and run on the implicit-GEMM kernel instead of failing)
        raster_only = w2 is not None and os.environ.get("MGLD_CONV3Q", "1") == "0"
        if not raster_only and stride == 1 and tuple(pad) == (1, 1) and (ho, wo) == ((2 * hin, 2 * win) if up2 else (hin, win)):
            wt = self._conv3p_tiled(wp, x.n, cin, cout, hin, win, tap_inner, up2)
            if wt is not None:
                w2t = None if w2 is None else self._conv3p_tiled(w2, x.n, cin, cout, hin, win, tap_inner, up2)
                if w2 is None or w2t is not None:
                    wp, w2, tap_inner, kw = wt, w2t, 2, dict(N=cout, K=9 * cin)
        out.stats = out.rstats = None
        got = []
        if stats and self.GN_PRODUCER and out.v.dtype == torch.float16 and not (self.GN_FUSED and hip.gn_fused_applies(ho * wo, cout, self.GROUPS)):
            def part(chunks):
                got.append((self.arena.alloc((x.n * chunks, 2, cout), torch.float32), hip.GN_CHANNEL_SUMS, chunks, None))
                return got[0][0]
            kw["gn_part"] = part
        hip.igemm(x.v, wp, out.v, mode=hip.MODE_CONV3X3, bias=bias, rowvec=rowvec, w2=w2,
                  rows_per_frame=rows_per_frame or ho * wo, resid=None if resid is None else resid.v, act=act, alpha=alpha,
                  beta=beta, conv=(cin, hin, win, ho, wo, stride, pad[0], pad[1], 1 if up2 else 0), tap_inner=tap_inner,
                  resid_lo=None if resid is None else resid.lo, out_lo=out.lo, **kw)
        if got:
            out.stats = got[0]
        self.launches += 1
        return out

    def _conv3p_tiled(self, wp, frames, cin, cout, h, w, tap_inner, up2=False):
        """weights of a convolution the library's patch kernel takes, re-laid as [N/16][Cin/32][9][16][32] so that every
        1-KiB DMA piece of a weight stage is contiguous (include/mgld_hip.h, tap_inner = 2).  Built once per weight tensor on
        first use (eager pass); None = keep the [N, K] layout."""
        geo = (cin, cout, h, w, bool(up2))
        ok = self._c3p_geo.get(geo)
        if ok is None:
            ok = self._c3p_geo[geo] = bool(hip.conv3p_applies(frames, cin, cout, h, w, up2))
        if not ok:
            return None
        key = (wp.data_ptr(), tap_inner)
        hit = self._c3p_w.get(key)
        if hit is None:
            if torch.cuda.is_current_stream_capturing():
                return None
            hit = self._c3p_w[key] = (hip.tile_conv3p(wp, cin, tap_inner), wp)     # mgld_tile_conv3p (keeps wp alive: the key is its address)
        return hit[0]

    def conv2d(self, x, wp, bias, cout, ksize, stride=1, pad=(0, 0), out=None, act=hip.ACT_NONE, alpha=1.0,
               out_dtype=torch.float16):
        """general kh x kw convolution (per-lane gather path of the igemm): wp from pack_conv, pad = (top, left)."""
        kh, kw = ksize
        ho = (x.h + 2 * pad[0] - kh) // stride + 1
        wo = (x.w + 2 * pad[1] - kw) // stride + 1
        if out is None:
            out = Act(self.arena.alloc((x.n * ho * wo, cout), out_dtype), x.n, ho, wo)
        cin = x.C
        out.stats = out.rstats = None
        assert wp.shape[1] == kh * kw * cin, (wp.shape, kh, kw, cin)
        hip.igemm(x.v, wp, out.v, mode=hip.MODE_CONV3X3, bias=bias, act=act, alpha=alpha, N=cout,
                  conv=(cin, x.h, x.w, ho, wo, stride, pad[0], pad[1], 0), ksize=(kh, kw))
        self.launches += 1
        return out

    def linear(self, x, w, bias, out=None, resid=None, act=hip.ACT_NONE, alpha=1.0, beta=1.0, out_dtype=torch.float16, n_out=None,
               w2=None, lo=False, rowstats=False, ln=None):
        """x: Act or 2-D view; w [N, K] fp16 packed.  lo (x an Act, no `out`): the output is a residual-stream tensor; an Act `out` / `resid`
        that carries a low plane has it written / added.  rowstats: the output feeds a LayerNorm — where the kernel the launcher picks can, it
        also writes the per-row sums of what it stores and the returned Act carries them (Act.rstats).  ln: see linear_ln."""
        xv = x.v if isinstance(x, Act) else x
        N = n_out if n_out is not None else (w.shape[0] // 2 if act == hip.ACT_GEGLU else w.shape[0])
        if out is None:
            if isinstance(x, Act):
                out = self.act(x.n, x.h, x.w, N, out_dtype, lo=lo and out_dtype == torch.float16)
            else:
                assert not lo, "a residual-stream output needs an Act input (frame geometry)"
                out = self.arena.alloc((xv.shape[0], N), out_dtype)
        ov = out.v if isinstance(out, Act) else out
        if isinstance(out, Act):
            out.stats = out.rstats = None
        rv = resid.v if isinstance(resid, Act) else resid
        kw, got = {}, []
        if rowstats and self.LN_FOLD and isinstance(out, Act) and ov.dtype == torch.float16:
            def part(chunks):
                got.append((self.arena.alloc((chunks, xv.shape[0], 2), torch.float32), chunks))
                return got[0][0]
            kw["row_part"] = part
        hip.igemm(xv, w, ov, bias=bias, resid=rv, act=act, alpha=alpha, beta=beta, M=xv.shape[0], N=w.shape[0], K=w.shape[1], w2=w2,
                  resid_lo=resid.lo if isinstance(resid, Act) else None, out_lo=out.lo if isinstance(out, Act) else None, ln=ln, **kw)
        if got:
            out.rstats = got[0]
        self.launches += 1
        return out

    def linear_ln(self, x, norm, tag, params, fn, act=hip.ACT_NONE):
        """act(LayerNorm(x) W^T + b) -> 2-D fp16 tensor; x: Act of token rows [rows, C]; fn(*fp32 params) -> (W [N, C], bias [N] or None) (GEGLU:
        both in the packed row order of pack_geglu).  When x carries the row sums its producer wrote (Act.rstats) and the ping-pong kernel takes
        the problem, the LayerNorm is FOLDED into the projection (MgldIGemm.ln_part): the contraction runs on the raw rows against W diag(gamma),
        the epilogue applies rstd_m (acc - mean_m s_n) + b'_n — no normalised copy, no LayerNorm launch, one rounding of the operand less.
        Otherwise: the LayerNorm kernel + the plain projection."""
        xv = x.v
        rows, C = xv.shape
        if self.LN_FOLD and x.rstats is not None and (act != hip.ACT_GEGLU or self.LN_FOLD_GEGLU):
            def pack(*ts):
                W, bz = fn(*ts[:-2])
                return pack_ln_fold(W, bz, ts[-2], ts[-1])
            wp, sv, b2 = self.weight(tag + "#ln", tuple(params) + (norm.weight, norm.bias), pack)
            N = wp.shape[0] // 2 if act == hip.ACT_GEGLU else wp.shape[0]
            key = ("lnq", rows, wp.shape[0], C, act)
            ok = self._wcache.get(key)
            out = self.arena.alloc((rows, N), torch.float16)
            if ok is None:
                ok = self._wcache[key] = hip.igemm(xv, wp, out, bias=b2, act=act, M=rows, N=wp.shape[0], K=C, query_row_chunks=True) > 0
            if ok:
                part, chunks = x.rstats
                hip.igemm(xv, wp, out, bias=b2, act=act, M=rows, N=wp.shape[0], K=C, ln=(part, chunks, sv, float(norm.eps)))
                self.launches += 1
                self.ln_folded += 1
                return out
        xn = self.layernorm(x, self.f32("g", norm.weight), self.f32("b", norm.bias), norm.eps).v

        def plain(*ts):
            W, bz = fn(*ts)
            return (W, bz) if bz is not None else (W, torch.zeros(0))
        w, bz = self.weight(tag, tuple(params), plain)
        return self.linear(xn, w, bz if bz.numel() else None, act=act)

    def tconv3(self, x, wp, bias, T, alpha_blend, out=None, w2=None):
        """SpatialTemporalConv: out = a*(conv3d_t(x)+b) + (1-a)*x."""
        if out is None:
            out = self.act(x.n, x.h, x.w, x.C, lo=x.lo is not None)
        out.stats = out.rstats = None
        sh = self.shard
        if sh is None:
            hip.igemm(x.v, wp, out.v, mode=hip.MODE_TCONV3, bias=bias, resid=x.v, alpha=alpha_blend, beta=1.0 - alpha_blend,
                      tconv=(x.C, T, x.hw), w2=w2, resid_lo=x.lo, out_lo=out.lo)
            self.launches += 1
            return out
        # frame-sharded clip (parallel.FrameShard): this rank holds F consecutive frames.  The conv runs on a halo-extended
        # copy [left neighbour's last frame | F frames | right neighbour's first frame]; zero frames at the clip's ends
        # reproduce the Conv3d zero padding.
        if x.n != sh.F:
            raise RuntimeError(f"sharded temporal conv: {x.n} local frames, shard expects {sh.F} (one clip per segment)")
        F, hw = x.n, x.hw
        ext = self.arena.alloc(((F + 2) * hw, x.C), torch.float16)
        mid = ext[hw:(F + 1) * hw]
        hip.copy2d(x.v, mid)
        self.collective(lambda: sh.halo(x.v, hw, ext[:hw], ext[(F + 1) * hw:]))
        hip.igemm(mid, wp, out.v, mode=hip.MODE_TCONV3, bias=bias, resid=x.v, alpha=alpha_blend, beta=1.0 - alpha_blend,
                  tconv=(x.C, F + 2, hw), t_off=1, w2=w2, resid_lo=x.lo, out_lo=out.lo)
        self.launches += 2
        return out

    def gn_stats(self, x, eps, groups=None):
        """One launch: per-chunk group sums (fp64).  Returns (gsums, eps); the apply kernels finish the reduction.
        Small frames (hip.gn_fused_applies: the 16x16 / 8x8 levels) return (None, eps): their consumer runs statistics + apply
        as ONE launch (mgld_gn_fused), so nothing is computed here."""
        g = groups or self.GROUPS
        if self.GN_FUSED and hip.gn_fused_applies(x.hw, x.C, g):
            return None, float(eps), 0, 0
        st = x.stats
        if st is not None and (st[3] is None or st[3] == g):       # written by the producer of x: nothing to launch
            self.gn_stats_saved += 1
            return st[0], float(eps), st[1], st[2]
        gsums = self.arena.alloc((x.n, hip.gn_chunks(x.hw), g, 2), torch.float64)
        hip.gn_stats(x.v, x.n, x.hw, g, gsums)
        self.launches += 1
        return gsums, float(eps), hip.GN_GROUP_SUMS, 0

    def gn_apply(self, x, stats, gamma, beta, silu, out=None, groups=None):
        if out is None:
            out = self.act(x.n, x.h, x.w, x.C)
        out.stats = out.rstats = None
        gsums, eps, kind, chunks = stats
        if gsums is None:
            hip.gn_fused(x.v, eps, gamma, beta, out.v, x.n, x.hw, groups or self.GROUPS, silu, lo_in=x.lo if self.lo_norms else None)
        else:
            hip.gn_apply(x.v, gsums, eps, gamma, beta, out.v, x.n, x.hw, groups or self.GROUPS, silu, kind=kind, chunks=chunks, x_lo=x.lo if self.lo_norms else None)
        self.launches += 1
        return out

    def groupnorm(self, x, gamma, beta, eps, silu, out=None):
        return self.gn_apply(x, self.gn_stats(x, eps), gamma, beta, silu, out)

    def spade_apply(self, h, stats, gamma, beta, gb, skip, out=None, step_idx=None, step_stride=0, want_stats=False):
        """want_stats: the output feeds a GroupNorm (the transformer's norm after a ResBlockDual): the apply kernel also reduces what it
        stores, and the returned Act carries the sums (Act.stats)"""
        if out is None:
            out = self.act(h.n, h.h, h.w, h.C, lo=True)       # skip + spade(h): the block's output, the next value of the residual stream
        out.stats = out.rstats = None
        gsums, eps, kind, chunks = stats
        gbv = gb.v if isinstance(gb, Act) else gb
        skip_lo, y_lo = skip.lo, out.lo
        if y_lo is not None and skip_lo is None:      # (a skip tensor without a low plane: an all-zero one)
            skip_lo = self.const(("lozeros", tuple(skip.v.shape), skip.v.stride(0)),
                                 lambda: torch.zeros(skip.v.shape[0] * skip.v.stride(0), dtype=torch.float16, device=self.device)
                                 ).as_strided(skip.v.shape, skip.v.stride())
        if y_lo is None:
            skip_lo = None
        if gsums is None:
            hip.gn_fused(h.v, eps, gamma, beta, out.v, h.n, h.hw, self.GROUPS, 0, gb=gbv, skip=skip.v, step_idx=step_idx,
                         step_stride=step_stride, lo_in=skip_lo, lo_out=y_lo)
        else:
            so = None
            if want_stats and self.GN_PRODUCER:
                oc = hip.gn_apply_chunks(h.n, h.hw, h.C, self.GROUPS)
                so = self.arena.alloc((h.n, oc, self.GROUPS, 2), torch.float64)
            hip.spade_apply(h.v, gsums, eps, gamma, beta, gbv, skip.v, out.v, h.n, h.hw, self.GROUPS, step_idx, step_stride, kind=kind,
                            chunks=chunks, stats_out=so, skip_lo=skip_lo, y_lo=y_lo)
            if so is not None:
                out.stats = (so, hip.GN_GROUP_SUMS, oc, self.GROUPS)
        self.launches += 1
        return out

    def layernorm(self, x, gamma, beta, eps=1e-5):
        xv = x.v if isinstance(x, Act) else x
        out = self.arena.alloc((xv.shape[0], xv.shape[1]), torch.float16)
        hip.layernorm(xv, gamma, beta, out, eps, x_lo=x.lo if (isinstance(x, Act) and self.lo_norms) else None)
        self.launches += 1
        return Act(out, x.n, x.h, x.w) if isinstance(x, Act) else out

    def from_nchw(self, x, cpad=None):
        """fp32 NCHW device tensor -> fp16 Act with channels zero-padded to a multiple of 8."""
        n, c, h, w = x.shape
        cp = cpad or ((c + 7) // 8 * 8)
        out = self.act(n, h, w, cp)
        hip.nchw_to_nhwc(x.contiguous().float(), out.v, cp)
        self.launches += 1
        return out

    def to_nchw(self, a, c=None, out=None):
        c = c or a.C
        if out is None:
            out = torch.empty(a.n, c, a.h, a.w, dtype=torch.float32, device=self.device)
        hip.nhwc_to_nchw(a.v, out)
        self.launches += 1
        return out
