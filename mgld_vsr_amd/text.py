"""Text conditioning (reference ldm/modules/encoders/modules.py:140-199, FrozenOpenCLIPEmbedder).

The VSR scripts call the text tower once per segment with the empty prompt, so its output is a CONSTANT [n,77,1024] tensor
outside the per-step hot path.  Two modes:

* `build_tower=False` (default; every BASELINE config): no OpenCLIP parameters exist, `forward` returns the deterministic
  synthetic context (or whatever `set_context` installed).
* `build_tower=True`: the OpenCLIP text transformer (ViT-H-14 text: width 1024, 16 heads, 24 layers, context 77) is
  instantiated under open_clip's own parameter names (`model.token_embedding`, `model.positional_embedding`,
  `model.transformer.resblocks.N.{ln_1, attn.in_proj_*, attn.out_proj, ln_2, mlp.c_fc, mlp.c_proj}`, `model.ln_final`,
  `model.text_projection`, `model.logit_scale`), so the `cond_stage_model.*` entries of the public checkpoint load as they
  are, and `forward` runs it on the C-ABI kernels (LayerNorm, igemm with fused bias / GELU / residual, fp32 logits +
  causal row softmax).  Tokenisation: mgld_vsr_amd/tokenizer.py restates the CLIP BPE algorithm; its merge table is open_clip's
  data file (found in an installed open_clip / clip package or via $MGLD_BPE_VOCAB).  The empty prompt — the only one the
  inference scripts use — tokenises to [SOT, EOT, 0, ...] without the table.
"""
import warnings

import torch
import torch.nn as nn

from . import hip, synth


class _ResidualAttentionBlock(nn.Module):
    """open_clip ResidualAttentionBlock parameter layout"""

    def __init__(self, width, heads):
        super().__init__()
        self.ln_1 = nn.LayerNorm(width)
        self.attn = nn.MultiheadAttention(width, heads)      # in_proj_weight / in_proj_bias / out_proj
        self.ln_2 = nn.LayerNorm(width)
        self.mlp = nn.Sequential()
        self.mlp.add_module("c_fc", nn.Linear(width, width * 4))
        self.mlp.add_module("gelu", nn.GELU())
        self.mlp.add_module("c_proj", nn.Linear(width * 4, width))


class _Transformer(nn.Module):
    def __init__(self, width, layers, heads):
        super().__init__()
        self.resblocks = nn.ModuleList([_ResidualAttentionBlock(width, heads) for _ in range(layers)])


class _TextTower(nn.Module):
    def __init__(self, vocab_size, context_length, width, layers, heads):
        super().__init__()
        self.token_embedding = nn.Embedding(vocab_size, width)
        self.positional_embedding = nn.Parameter(torch.empty(context_length, width))
        self.transformer = _Transformer(width, layers, heads)
        self.ln_final = nn.LayerNorm(width)
        self.text_projection = nn.Parameter(torch.empty(width, width))
        self.logit_scale = nn.Parameter(torch.ones([]))


class FrozenOpenCLIPEmbedder(nn.Module):
    LAYERS = ["last", "penultimate"]

    def __init__(self, arch="ViT-H-14", version="laion2b_s32b_b79k", device="cuda", max_length=77, freeze=True,
                 layer="last", context_dim=1024, build_tower=False, heads=None, layers=24, vocab_size=49408):
        super().__init__()
        assert layer in self.LAYERS
        self.device, self.max_length, self.layer, self.context_dim = device, max_length, layer, context_dim
        self.layer_idx = 0 if layer == "last" else 1
        self.heads = heads or max(1, context_dim // 64)
        self.vocab_size = vocab_size
        self._context = None
        self._engine = None
        self.model = _TextTower(vocab_size, max_length, context_dim, layers, self.heads) if build_tower else None
        if freeze:
            self.freeze()

    def freeze(self):
        """modules.py:167-170"""
        if self.model is not None:
            self.model = self.model.eval()
        for p in self.parameters():
            p.requires_grad = False

    def build_tower(self, layers=24, vocab_size=None, heads=None):
        """instantiate the text transformer after construction (checkpoint loading: the `cond_stage_model.model.*` entries need
        parameters to land in)"""
        if vocab_size is not None:
            self.vocab_size = vocab_size
        if heads is not None:
            self.heads = heads
        self.model = _TextTower(self.vocab_size, self.max_length, self.context_dim, layers, self.heads).eval()
        for p in self.parameters():
            p.requires_grad = False
        return self

    def set_context(self, ctx):
        """install a precomputed empty-prompt embedding [1,77,context_dim]"""
        self._context = ctx

    def set_engine(self, eng):
        self._engine = eng

    # ---- tokenisation ------------------------------------------------------------------------------------------
    def tokenize(self, text):
        """open_clip.tokenize: [SOT, bpe..., EOT, 0 padding]; SOT / EOT are the last two vocabulary entries."""
        toks = torch.zeros(len(text), self.max_length, dtype=torch.long)
        for i, t in enumerate(text):
            if t == "":
                toks[i, 0], toks[i, 1] = self.vocab_size - 2, self.vocab_size - 1
                continue
            toks[i] = self._bpe().tokenize([t], self.max_length)[0]
        return toks

    def _bpe(self):
        """the BPE tokenizer over open_clip's merge table (mgld_vsr_amd/tokenizer.py: the algorithm is restated there, the table is
        looked up in $MGLD_BPE_VOCAB / an installed open_clip or clip package); built on first use by a non-empty prompt"""
        tk = getattr(self, "_tokenizer", None)
        if tk is None:
            from .tokenizer import SimpleTokenizer, find_vocab
            path = find_vocab()
            if path is None:
                raise NotImplementedError("non-empty prompts need open_clip's BPE merge table (bpe_simple_vocab_16e6.txt.gz): set "
                                          "MGLD_BPE_VOCAB or install open_clip; the empty prompt — the only one the inference "
                                          "scripts use — works without it")
            tk = self._tokenizer = SimpleTokenizer(path, vocab_size=self.vocab_size)
            assert tk.eot == self.vocab_size - 1, "BPE table and text tower disagree on the vocabulary size"
        return tk

    # ---- forward ------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def forward(self, text):
        text = list(text) if isinstance(text, (list, tuple)) else [text]
        if self.model is None:
            if self._context is None:
                if getattr(self, "require_real_context", False):
                    raise RuntimeError("FrozenOpenCLIPEmbedder: real weights were loaded but neither a text tower (cond_stage_model.* "
                                       "in the checkpoint) nor a precomputed context (set_context) is available")
                warnings.warn("FrozenOpenCLIPEmbedder: no OpenCLIP weights on this path; using the synthetic constant context")
                self._context = synth.synth_tensor("ctx", (1, self.max_length, self.context_dim))
            return self._context.repeat(len(text), 1, 1)
        return self.encode_with_transformer(self.tokenize(text))

    def encode(self, text):
        return self(text)

    @torch.no_grad()
    def encode_with_transformer(self, tokens):
        """modules.py:181-199 on the HIP kernels.  tokens [n,77] (host) -> [n,77,width] fp32 on the device."""
        from .engine import Engine
        if self._engine is None:
            self._engine = Engine()
        eng = self._engine
        eng.reset()
        m = self.model
        n, L = tokens.shape
        W, H = self.context_dim, self.heads
        d = W // H
        Lp = (L + 7) // 8 * 8                                       # key axis padded for the P @ V GEMM
        x0 = (m.token_embedding.weight.detach()[tokens] + m.positional_embedding.detach()).float()      # host gather
        x = x0.reshape(n * L, W).to(eng.device, torch.float16).contiguous()
        blocks = m.transformer.resblocks
        for blk in list(blocks)[:len(blocks) - self.layer_idx]:
            a = blk.attn
            h = eng.layernorm(x, eng.f32("g", blk.ln_1.weight), eng.f32("b", blk.ln_1.bias), blk.ln_1.eps)
            qkv = eng.linear(h, eng.weight("w", (a.in_proj_weight,), lambda w: w), eng.f32("b", a.in_proj_bias))   # [nL, 3W]
            # per (sequence, head): S = q k^T / sqrt(d) (fp32), causal softmax, O = P V with V^T from a batched NT GEMM
            o = eng.arena.alloc((n * L, W), torch.float16)
            for b in range(n):
                r = slice(b * L, (b + 1) * L)
                S = eng.arena.alloc((H * L, L), torch.float32)
                hip.igemm(qkv[r, 0:W], qkv[r, W:2 * W], S, M=L, N=L, K=d, alpha=float(d) ** -0.5, batch=H, strideA=d, strideW=d,
                          strideC=L * L)
                P = eng.arena.alloc((H * L, Lp), torch.float16)
                hip.softmax_rows_masked(S, P, H * L, L, Lp, L)
                vt = eng.arena.alloc((W, Lp), torch.float16)          # V^T [W, Lp]; pad columns meet P's zeros but must be finite
                vt.zero_()
                wv = eng.weight("wv", (a.in_proj_weight,), lambda w: w[2 * W:])
                bv = eng.weight("bv", (a.in_proj_bias,), lambda t: t[2 * W:], torch.float32)
                hip.igemm(wv, h[r], vt, bias_m=bv, M=W, N=L, K=W)
                hip.igemm(P, vt, o[r], M=L, N=d, K=Lp, batch=H, strideA=L * Lp, strideW=d * Lp, strideC=d)
                eng.launches += 4
            x = eng.linear(o, eng.weight("w", (a.out_proj.weight,), lambda w: w), eng.f32("b", a.out_proj.bias), resid=x)
            h2 = eng.layernorm(x, eng.f32("g", blk.ln_2.weight), eng.f32("b", blk.ln_2.bias), blk.ln_2.eps)
            f = eng.linear(h2, eng.weight("w", (blk.mlp.c_fc.weight,), lambda w: w), eng.f32("b", blk.mlp.c_fc.bias),
                           act=hip.ACT_GELU)
            x = eng.linear(f, eng.weight("w", (blk.mlp.c_proj.weight,), lambda w: w), eng.f32("b", blk.mlp.c_proj.bias), resid=x)
        y = eng.layernorm(x, eng.f32("g", m.ln_final.weight), eng.f32("b", m.ln_final.bias), m.ln_final.eps)
        return y.float().reshape(n, L, W)
