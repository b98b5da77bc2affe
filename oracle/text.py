"""TEST INFRASTRUCTURE — CPU fp32 restatement of FrozenOpenCLIPEmbedder.encode_with_transformer
(ldm/modules/encoders/modules.py:181-199) over open_clip's text-transformer state_dict.

PINNED (round 5) to the reference's own embedder class: tests/golden/g_text_openclip.npz holds outputs of
`FrozenOpenCLIPEmbedder.encode_with_transformer` (the reference's code: embedding sum, permutes, `attn_mask`, the `penultimate` layer
slice, `ln_final`) run in the build container on a stand-in for the model object `open_clip.create_model_and_transforms` returns —
open_clip_torch itself (un-vendored dependency, modules.py:12) is not installed, so the ResidualAttentionBlock it would supply
(x += MHA(ln_1(x), causal mask); x += c_proj(GELU(c_fc(ln_2(x))))) is restated there from its published definition, on
nn.MultiheadAttention as open_clip has it (tests/test_oracle_golden.py::test_text_tower_openclip_golden, 1e-5).  Second, independent
pin: transformers' CLIPTextModel, the class the reference's FrozenCLIPEmbedder binds (modules.py:7, :207), on synthetic weights,
`last` and `penultimate` layer (tests/golden/g_text_hf.npz, ::test_text_tower_golden).  What stays unpinned: open_clip's own block
implementation bit for bit (its attention goes through the same torch function this file calls).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package."""
import torch
import torch.nn.functional as F


def build_attention_mask(n_ctx):
    """open_clip: additive causal mask, -inf above the diagonal"""
    mask = torch.empty(n_ctx, n_ctx)
    mask.fill_(float("-inf"))
    mask.triu_(1)
    return mask


def encode_with_transformer(sd, tokens, heads, layer_idx=1, pre="model."):
    """sd: state_dict of the embedder (`model.*` keys); tokens [n, n_ctx] long -> [n, n_ctx, width]"""
    x = sd[pre + "token_embedding.weight"][tokens] + sd[pre + "positional_embedding"]          # :182-183
    x = x.permute(1, 0, 2)                                                                       # NLD -> LND (:184)
    n_layers = 1 + max(int(k.split(".")[3]) for k in sd if k.startswith(pre + "transformer.resblocks."))
    mask = build_attention_mask(tokens.shape[1])
    W = x.shape[-1]
    for i in range(n_layers - layer_idx):                                                        # :190-193 (penultimate: skip the last)
        b = f"{pre}transformer.resblocks.{i}."
        h = F.layer_norm(x, (W,), sd[b + "ln_1.weight"], sd[b + "ln_1.bias"], 1e-5)
        a, _ = F.multi_head_attention_forward(h, h, h, W, heads, sd[b + "attn.in_proj_weight"], sd[b + "attn.in_proj_bias"], None, None,
                                              False, 0.0, sd[b + "attn.out_proj.weight"], sd[b + "attn.out_proj.bias"], training=False,
                                              need_weights=False, attn_mask=mask)
        x = x + a
        h = F.layer_norm(x, (W,), sd[b + "ln_2.weight"], sd[b + "ln_2.bias"], 1e-5)
        h = F.linear(F.gelu(F.linear(h, sd[b + "mlp.c_fc.weight"], sd[b + "mlp.c_fc.bias"])), sd[b + "mlp.c_proj.weight"],
                     sd[b + "mlp.c_proj.bias"])
        x = x + h
    x = x.permute(1, 0, 2)                                                                       # :186
    return F.layer_norm(x, (W,), sd[pre + "ln_final.weight"], sd[pre + "ln_final.bias"], 1e-5)  # :187
