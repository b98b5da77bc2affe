"""CPU-only checks of the host side: C-ABI exports, checkpoint key layout (state_dict contract),
schedule buffers, tiling geometry, weight packing.  No compute call is made without a GPU."""
import ctypes
import json
import os
import re
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(HERE, "golden"))

from configs import STRUCT_SMALL, T, UNET_SMALL, VAE_DD_SMALL  # noqa: E402


def G(name):
    d = np.load(os.path.join(HERE, "golden", name + ".npz"))
    return {k: (torch.from_numpy(d[k]) if d[k].dtype.kind in "fiu" else d[k]) for k in d.files}


def test_library_exports_every_declared_symbol():
    from mgld_vsr_amd import build, hip
    path = build.build(verbose=False)
    lib = ctypes.CDLL(path)
    header = open(os.path.join(ROOT, "include", "mgld_hip.h")).read()
    declared = set(re.findall(r"\b(mgld_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations parsed"
    for sym in sorted(declared):
        assert hasattr(lib, sym), f"libmgld_hip.so does not export {sym}"
    assert declared == set(hip.EXPORTS), declared ^ set(hip.EXPORTS)
    assert lib.mgld_version() >= 100


def _keys(module):
    return {k: tuple(v.shape) for k, v in module.state_dict().items() if v.is_floating_point()}


def test_ctypes_structs_match_the_c_header(tmp_path):
    """include/mgld_hip.h is a plain-C header: compile a probe with gcc that prints sizeof / offsetof of every field of the
    argument structs and compare with the ctypes mirror in mgld_vsr_amd/hip.py (the binding a maintainer would write)."""
    import ctypes
    import shutil
    import subprocess
    from mgld_vsr_amd import hip
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("no C compiler on this box")
    lines = ["#include <stdio.h>", "#include <stddef.h>", '#include "mgld_hip.h"', "int main(void) {"]
    structs = (("MgldIGemm", hip.MgldIGemm), ("MgldAttn", hip.MgldAttn), ("MgldGnStats", hip.MgldGnStats))
    for name, cls in structs:
        lines.append(f'  printf("{name} %zu\\n", sizeof({name}));')
        for f, _ in cls._fields_:
            lines.append(f'  printf("{name}.{f} %zu\\n", offsetof({name}, {f}));')
    lines += ["  return 0;", "}"]
    src = tmp_path / "probe.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "probe"
    subprocess.check_call([gcc, "-std=c99", "-I", os.path.join(ROOT, "include"), "-o", str(exe), str(src)])
    got = dict(ln.split() for ln in subprocess.check_output([str(exe)], text=True).splitlines())
    for name, cls in structs:
        assert int(got[name]) == ctypes.sizeof(cls)
        for f, _ in cls._fields_:
            assert int(got[f"{name}.{f}"]) == getattr(cls, f).offset, (name, f)


def test_cli_option_surface_matches_the_reference_script(capsys):
    """every option of scripts/vsr_val_ddpm_text_T_vqganfin_oldcanvas_tile.py (reference :134-:262) is accepted with the
    reference's default; the defaults that name files the reference ships fall back to built-ins when absent"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("cli", os.path.join(ROOT, "scripts", "vsr_val_ddpm_text_T_vqganfin_oldcanvas_tile.py"))
    cli = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(cli)
    o = cli.parse([])
    ref_defaults = dict(seqs_path="inputs/user_upload", outdir="outputs/user_upload", device="cuda", ddpm_steps=1000, n_iter=1, C=4,
                        f=8, n_frames=5, n_samples=1, seed=42, precision="autocast", select_idx=0, n_gpus=1, dec_w=0.5,
                        tile_overlap=32, upscale=4.0, colorfix_type="nofix", vqgantile_stride=750, vqgantile_size=960)
    for k, v in ref_defaults.items():
        assert getattr(o, k) == v, k
    assert (cli.REF_CONFIG, cli.REF_CKPT, cli.REF_VQGAN_CKPT) == ("configs/stable-diffusion/v1-inference.yaml",
                                                                 "checkpoints/stablevsr_025.ckpt", "checkpoints/vqgan_cfw_00011.ckpt")
    assert o.config is None and o.ckpt is None and o.vqgan_ckpt is None      # default files absent here -> built-ins
    o = cli.parse(["--seqs-path", "a", "--outdir", "b", "--ddpm_steps", "50", "--n_iter", "1", "--C", "4", "--f", "8", "--n_frames", "5",
                   "--n_samples", "1", "--seed", "1", "--precision", "full", "--select_idx", "1", "--n_gpus", "2", "--dec_w", "1.0",
                   "--tile_overlap", "32", "--upscale", "4", "--colorfix_type", "adain", "--vqgantile_stride", "750",
                   "--vqgantile_size", "960", "--device", "cuda"])
    assert o.select_idx == 1 and o.n_gpus == 2 and o.colorfix_type == "adain"
    for bad in (["--device", "cpu"], ["--C", "8"], ["--ckpt", "/nonexistent/x.ckpt"]):
        with pytest.raises(SystemExit):
            cli.parse(bad)
    capsys.readouterr()


def test_unet_state_dict_contract():
    from ldm.modules.diffusionmodules.openaimodel import InflatedEncoderUNetModelWT, InflatedUNetModelDualcondV2
    g = G("g_unet")
    ref = {k: tuple(s) for k, s in json.loads(str(g["unet_params"]))}
    assert _keys(InflatedUNetModelDualcondV2(**UNET_SMALL)) == ref
    ref = {k: tuple(s) for k, s in json.loads(str(g["struct_params"]))}
    assert _keys(InflatedEncoderUNetModelWT(**STRUCT_SMALL)) == ref


def test_vae_state_dict_contract():
    from ldm.models.autoencoder import AutoencoderKL, VideoAutoencoderKLResi
    g = G("g_vae")
    ref = {k: tuple(s) for k, s in json.loads(str(g["vae_params"]))}
    mine = _keys(VideoAutoencoderKLResi(ddconfig=dict(VAE_DD_SMALL), lossconfig={"target": "torch.nn.Identity"}, embed_dim=4))
    assert mine == ref
    g = G("g_first_stage")
    ref = {k: tuple(s) for k, s in json.loads(str(g["params"]))}
    dd = dict(VAE_DD_SMALL)
    dd.pop("num_frames")
    mine = _keys(AutoencoderKL(ddconfig=dd, lossconfig={"target": "torch.nn.Identity"}, embed_dim=4))
    assert set(mine) <= set(ref) and all(mine[k] == ref[k] for k in mine)
    assert all(k.startswith("decoder.") for k in set(ref) - set(mine))  # image decoder: not on the VSR path


def _small_model():
    from ldm.models.diffusion.ddpm import LatentDiffusionVSRTextWT
    dd = dict(VAE_DD_SMALL)
    dd.pop("num_frames")
    return LatentDiffusionVSRTextWT(
        first_stage_config={"target": "ldm.models.autoencoder.AutoencoderKL",
                            "params": {"embed_dim": 4, "ddconfig": dd, "lossconfig": {"target": "torch.nn.Identity"}}},
        cond_stage_config={"target": "ldm.modules.encoders.modules.FrozenOpenCLIPEmbedder",
                           "params": {"freeze": True, "layer": "penultimate", "device": "cuda", "context_dim": 64}},
        structcond_stage_config={"target": "ldm.modules.diffusionmodules.openaimodel.InflatedEncoderUNetModelWT",
                                 "params": dict(STRUCT_SMALL)},
        flownet_config={"target": "basicsr.archs.raft_arch.RAFT_SR", "params": {"model": "normal", "load_path": None}},
        unet_config={"target": "ldm.modules.diffusionmodules.openaimodel.InflatedUNetModelDualcondV2",
                     "params": dict(UNET_SMALL)},
        num_frames=T, linear_start=0.00085, linear_end=0.0120, num_timesteps_cond=1, log_every_t=200, timesteps=1000,
        first_stage_key="image", cond_stage_key="caption", image_size=128, channels=4, cond_stage_trainable=False,
        conditioning_key="crossattn", scale_factor=0.18215, use_ema=False, time_replace=1000, use_usm=True)


@pytest.mark.parametrize("S", [4, 50])
def test_model_schedule_matches_reference(S):
    import copy
    from ldm.models.diffusion.ddpm import space_timesteps
    g = G("g_schedule")
    model = _small_model()
    sd_keys = set(model.state_dict().keys())
    assert any(k.startswith("model.diffusion_model.input_blocks.") for k in sd_keys)
    assert any(k.startswith("structcond_stage_model.fea_tran.") for k in sd_keys)
    assert any(k.startswith("first_stage_model.encoder.") for k in sd_keys) and "betas" in sd_keys
    # the script's respacing procedure (oldcanvas_tile.py:308-329) against the drop-in model
    model.register_schedule(given_betas=None, beta_schedule="linear", timesteps=1000, linear_start=0.00085,
                            linear_end=0.0120, cosine_s=8e-3)
    model.num_timesteps = 1000
    sac = copy.deepcopy(model.sqrt_alphas_cumprod)
    use = set(space_timesteps(1000, [S]))
    last, new_betas = 1.0, []
    for i, ac in enumerate(model.alphas_cumprod):
        if i in use:
            new_betas.append(1 - ac / last)
            last = ac
    new_betas = [b.data.cpu().numpy() for b in new_betas]
    model.register_schedule(given_betas=np.array(new_betas), timesteps=len(new_betas))
    model.ori_timesteps = sorted(list(use))
    assert model.ori_timesteps == g[f"S{S}_ori_timesteps"].tolist()
    for k in ["betas", "alphas_cumprod", "alphas_cumprod_prev", "sqrt_alphas_cumprod", "sqrt_one_minus_alphas_cumprod",
              "sqrt_recip_alphas_cumprod", "sqrt_recipm1_alphas_cumprod", "posterior_variance",
              "posterior_log_variance_clipped", "posterior_mean_coef1", "posterior_mean_coef2"]:
        assert torch.equal(getattr(model, k), g[f"S{S}_{k}"]), k
    assert torch.equal(sac, g[f"S{S}_full_sqrt_alphas_cumprod"])
    t = torch.tensor([999] * g["qs_x0"].shape[0]).long()
    out = model.q_sample_respace(g["qs_x0"], t, sac, g[f"S{S}_full_sqrt_one_minus_alphas_cumprod"], g["qs_noise"])
    assert torch.equal(out, g["qs_out"])


def test_ddpm_helper_surface():
    """the schedule helpers the reference class inherits from DDPM (q_sample, predict_start_from_noise, q_posterior:
    ddpm.py:340-353, 398-401) and the two loop entry points behind sample / sample_canvas exist with the reference's
    argument lists; the helpers are buffer gathers and invert each other"""
    import inspect
    model = _small_model()
    model.register_schedule(given_betas=None, beta_schedule="linear", timesteps=1000, linear_start=0.00085, linear_end=0.0120,
                            cosine_s=8e-3)
    x0, eps = torch.randn(3, 4, 8, 8), torch.randn(3, 4, 8, 8)
    t = torch.tensor([10, 500, 999])
    xt = model.q_sample(x0, t, eps)
    a = model.sqrt_alphas_cumprod[t].view(3, 1, 1, 1)
    b = model.sqrt_one_minus_alphas_cumprod[t].view(3, 1, 1, 1)
    assert torch.equal(xt, a * x0 + b * eps)
    assert torch.allclose(model.predict_start_from_noise(xt, t, eps), x0, atol=2e-3)   # sqrt(1/ac - 1) reaches ~240 at t = 999
    mean, var, logvar = model.q_posterior(x0, xt, t)
    c1, c2 = model.posterior_mean_coef1[t].view(3, 1, 1, 1), model.posterior_mean_coef2[t].view(3, 1, 1, 1)
    assert torch.equal(mean, c1 * x0 + c2 * xt) and var.shape == (3, 1, 1, 1)
    assert torch.equal(logvar.flatten(), model.posterior_log_variance_clipped[t])
    ref_loop = ["cond", "struct_cond", "shape", "guidance_scale", "lr_images", "flows", "masks", "return_intermediates", "x_T",
                "verbose", "callback", "timesteps", "quantize_denoised", "mask", "x0", "img_callback", "start_T", "log_every_t",
                "time_replace", "adain_fea", "interfea_path"]
    sig = lambda f: [k for k in inspect.signature(f).parameters if k != "self"]
    assert sig(model.p_sample_loop) == ref_loop                                          # ddpm.py:4501-4505
    assert sig(model.p_sample_loop_canvas) == ref_loop + ["tile_size", "tile_overlap", "batch_size"]   # ddpm.py:4619-4623
    with pytest.raises(NotImplementedError):      # still refused, never ignored: the PCA feature dump (lr_images is implemented since round 5:
        model.p_sample_loop(None, None, (3, 4, 8, 8), interfea_path="/tmp/fea")      # tests/test_nets_gpu.py::test_sample_lr_images_guidance_vs_reference)
    with pytest.raises(NotImplementedError):
        model.p_sample_loop(None, None, (3, 4, 8, 8), quantize_denoised=True)


def test_drop_in_signatures():
    """every public entry point of the reference on this path exists here under the same dotted name and takes the
    reference's arguments in the reference's order (tests/golden/g_signatures.json, read from the reference sources with
    `ast` by make_golden.py); this side may only append optional extras (noise=, use_graph=, **kw)"""
    import importlib
    import inspect
    import json
    with open(os.path.join(HERE, "golden", "g_signatures.json")) as fh:
        ref = json.load(fh)
    problems = []
    for key, want in ref.items():
        rel, cname, fname = key.split(":")
        mod = importlib.import_module(rel[:-3].replace("/", "."))
        owner = getattr(mod, cname) if cname and cname != "DDPM" else (
            importlib.import_module("ldm.models.diffusion.ddpm").LatentDiffusionVSRTextWT if cname == "DDPM" else mod)
        fn = getattr(owner, fname, None)
        if fn is None:
            problems.append(f"missing {key}")
            continue
        sig = inspect.signature(fn)
        have = [k for k, p in sig.parameters.items() if k != "self" and p.kind not in (p.VAR_POSITIONAL, p.VAR_KEYWORD)]
        renamed_ok = {("encode_with_transformer", "text"): "tokens"}      # the reference passes token ids under the name `text`
        have_cmp = have[:len(want)]
        want_cmp = [renamed_ok.get((fname, w), w) for w in want]
        if have_cmp != want_cmp:
            problems.append(f"{key}: reference {want} / here {have}")
            continue
        for extra in have[len(want):]:
            if sig.parameters[extra].default is inspect.Parameter.empty:
                problems.append(f"{key}: extra required argument {extra}")
    assert not problems, "\n".join(problems)


def test_tiling_geometry_and_weights():
    g = G("g_sample")
    model = _small_model()
    assert torch.equal(model._gaussian_weights(16, 16, 1)[0, 0], g["gauss16"])
    assert torch.equal(model._gaussian_weights(64, 64, 1)[0, 0], g["gauss64"])
    from oracle import sampler as osamp
    for (h, w, ts, ov) in [(128, 128, 64, 32), (24, 24, 16, 8), (64, 64, 64, 32), (96, 160, 64, 32), (70, 70, 64, 32)]:
        assert model._tile_origins(h, w, ts, ov) == osamp.tile_origins(h, w, ts, ov)


def test_weight_packing():
    from mgld_vsr_amd.engine import pack_conv1x1, pack_conv3x3, pack_geglu, pack_tconv3
    w = torch.randn(5, 3, 3, 3)
    p = pack_conv3x3(w)
    assert p.shape == (5, 72)
    assert torch.equal(p.reshape(5, 3, 3, 8)[:, 1, 2, :3], w[:, :, 1, 2]) and float(p.reshape(5, 9, 8)[:, :, 3:].abs().max()) == 0
    assert pack_conv1x1(torch.randn(6, 4, 1, 1)).shape == (6, 8)
    w3 = torch.randn(4, 4, 3, 1, 1)
    assert torch.equal(pack_tconv3(w3).reshape(4, 3, 4)[:, 2], w3[:, :, 2, 0, 0])
    wg, bg = torch.randn(128, 16), torch.randn(128)
    wp, bp = pack_geglu(wg, bg)
    assert torch.equal(wp[0:32], wg[0:32]) and torch.equal(wp[32:64], wg[64:96]) and torch.equal(wp[64:96], wg[32:64])
    assert torch.equal(bp[32:64], bg[64:96])


def test_conv3p_tiled_weight_layout():
    """tile_conv3p: element (n, tap, c) of either [N, K] order lands where include/mgld_hip.h (tap_inner = 2) says:
    [N/64][Cin/32][dy][16-row group][dx][row][16-byte slot ^ ((row >> 2) & 3)][8]; rows past N are zero"""
    from mgld_vsr_amd.engine import pack_conv3x3, tile_conv3p
    n_out, cin = 96, 128
    w = torch.randn(n_out, cin, 3, 3)
    nh = cin // 32
    for ti in (False, True):
        wp = pack_conv3x3(w, tap_inner=ti)
        t = tile_conv3p(wp, cin, ti).reshape(-1)
        assert t.numel() == 128 * 9 * cin
        g = torch.Generator().manual_seed(0)
        for _ in range(500):
            n, tap, c = (int(torch.randint(0, hi, (1,), generator=g)) for hi in (n_out, 9, cin))
            g64, rb, row = n // 64, (n % 64) // 16, n % 16
            h, cc = c // 32, c % 32
            dy, dx = tap // 3, tap % 3
            slot = (cc // 8) ^ ((row >> 2) & 3)
            piece = (((g64 * nh + h) * 3 + dy) * 4 + rb) * 3 + dx
            assert t[piece * 512 + row * 32 + slot * 8 + cc % 8] == w[n, c, dy, dx]
        # rows 96..127 of the second 64-row group are padding
        tz = t.reshape(2, nh, 3, 4, 3, 16, 32)
        assert float(tz[1, :, :, 2:].abs().max()) == 0


def test_no_kernel_spills_to_scratch():
    """Every kernel of the shipped library, from the code objects' own metadata: no private segment (scratch), no VGPR spill — the
    256 x 320 LINEAR tile sits at 256 VGPRs, and one more runtime check in its epilogue (round 6: a ReLU branch) put it into scratch and cost
    30 % of its time before this test existed."""
    import codeobj
    from mgld_vsr_amd import hip
    hip.lib()
    ks = codeobj.kernels(hip.lib_path())
    assert len(ks) > 200, len(ks)
    bad = [(k[".name"], k.get(".private_segment_fixed_size"), k.get(".vgpr_spill_count")) for k in ks
           if k.get(".private_segment_fixed_size", 0) or k.get(".vgpr_spill_count", 0)]
    assert not bad, bad
    assert max(k[".vgpr_count"] for k in ks) <= 256 and all(k[".wavefront_size"] == 64 for k in ks)


def test_conv3p_planner_routes_the_unet_convolutions():
    """mgld_igemm_config is host logic (no launch): which 3x3 convolutions take the patch-staged kernel, and with how many
    weight rows per block.  Code = 300000 + rows (+ splits * 1e6 once a split-K workspace is registered, which needs a GPU)."""
    from mgld_vsr_amd import hip
    hip.lib()

    def code(frames, cin, cout, h, w, stride=1, up2=0, pad=1, batch=1, tap_inner=0):
        p = hip.MgldIGemm()
        p.mode, p.M, p.N, p.K, p.batch, p.tap_inner = hip.MODE_CONV3X3, frames * h * w, cout, 9 * cin, batch, tap_inner
        p.Cin, p.Hin, p.Win, p.Hout, p.Wout, p.stride, p.pad_t, p.pad_l, p.up2 = cin, h, w, h, w, stride, pad, pad, up2
        return hip.igemm_config(p) % 1000000

    assert code(8, 320, 320, 64, 64) == 300064          # W = 64: 64 weight rows (two blocks per CU)
    assert code(8, 640, 640, 32, 32, tap_inner=1) == 300064
    assert code(8, 1280, 1280, 16, 16) == 300128        # W = 16: 128 rows
    assert code(8, 1280, 320, 16, 16) == 300064         # N = 64 (mod 128)
    assert hip.conv3p_applies(8, 512, 512, 64, 64) and hip.conv3p_applies(1, 32, 40, 16, 16)
    for args, kw in [((8, 1280, 1280, 8, 8), {}),                  # W = 8: H*W < 128 pixels per frame
                     ((8, 256, 256, 128, 128), {}),                # W > 64
                     ((8, 320, 320, 64, 64), dict(stride=2)),
                     ((8, 320, 320, 64, 64), dict(up2=1)),
                     ((8, 320, 320, 64, 64), dict(pad=0)),
                     ((8, 320, 4, 64, 64), {}),                    # N <= 32: the 128x32 im2col tile
                     ((8, 328, 320, 64, 64), {}),                  # Cin % 32 != 0
                     ((8, 96, 64, 16, 16), dict(tap_inner=1))]:    # (64-block, tap, c) order needs Cin % 64 == 0
        assert code(*args, **kw) < 300000, (args, kw)
    assert hip.conv3p_applies(8, 1280, 1280, 8, 8) and not hip.conv3p_applies(8, 1280, 1280, 4, 4)

    # tiled weights (tap_inner = 2, what the engine feeds): the 2-D-tile patch kernel, code 400000 + variant id
    def qcode(frames, cin, cout, h, w, up2=0, tune=0):
        p = hip.MgldIGemm()
        sc = 2 if up2 else 1
        p.mode, p.M, p.N, p.K, p.batch, p.tap_inner, p.tune = hip.MODE_CONV3X3, frames * h * w * sc * sc, cout, 9 * cin, 1, 2, tune
        p.Cin, p.Hin, p.Win, p.Hout, p.Wout, p.stride, p.pad_t, p.pad_l, p.up2 = cin, h, w, sc * h, sc * w, 1, 1, 1, up2
        return hip.igemm_config(p) % 1000000

    # round 4: the ping-pong patch conv (conv3r, code 600000 + configuration) takes the large levels; tune = conv3q variant + 1 keeps conv3q
    assert qcode(8, 320, 320, 64, 64) == 600006         # 16x32-pixel tiles x 80 channels: 256 blocks, one per CU
    assert qcode(8, 320, 320, 64, 64, tune=31) == 600000 and qcode(8, 320, 320, 64, 64, tune=33) < 600000   # forced: 160 | 320, 128 does not
    assert qcode(8, 320, 320, 64, 64, tune=8) == 400007 # (round 3) 8x32 tiles run by four waves of 64 pixels x 64 channels: opt-in
    assert qcode(8, 640, 640, 32, 32) == 600002         # 8x32 tiles x 128 channels (160 blocks beat 256 smaller ones, profiles/r04_pp_conv.txt)
    assert qcode(8, 640, 640, 32, 32, tune=6) == 400005 # conv3q: too few 256-pixel tiles for 256 CUs: 8x16 tiles, four waves
    assert qcode(8, 1280, 1280, 16, 16) == 400003       # 16x16 level: few tiles -> conv3q, one tile per frame, 128 weight rows, K split
    assert qcode(8, 128, 128, 512, 512) == 600008       # the VAE's large levels: 16x32 tiles x 128 channels
    assert qcode(8, 512, 512, 64, 64, up2=1) == 400007  # nearest-2x upsample folded into the tap offsets (16x16 tiles, four 64x64 waves: round 3)
    assert qcode(1, 64, 64, 8, 8, up2=1) == 400000
    assert qcode(8, 320, 320, 64, 64, tune=5) == 400004
    assert hip.conv3p_applies(8, 256, 256, 128, 128) and hip.conv3p_applies(5, 512, 512, 90, 120) and hip.conv3p_applies(8, 512, 512, 64, 64, True)
    assert qcode(8, 1280, 1280, 8, 8) == 400006         # 8x8 level: one 8x8 tile per frame (split along the channel slices on a GPU)
    assert qcode(8, 320, 4, 64, 64) < 300000 and qcode(8, 64, 64, 8, 12) < 300000
    # frame-stacked conv3r tiles of the 8 x 8 level: configurations 9 / 10 by tune only (the planner leaves the level to conv3q by default)
    assert qcode(8, 1280, 1280, 8, 8, tune=40) == 600009 and qcode(8, 1280, 1280, 8, 8, tune=41) == 600010
    assert qcode(8, 1280, 1280, 16, 16, tune=40) < 600000      # they take 8 x 8 frames only
    # round 6 (two segments batched as clips: 16 frames; profiles/r06_conv_2clip.txt): a choice whose last round of tiles is mostly empty gives
    # way to a large tile that fills whole rounds — 32^2 at 16 frames: 8x32x128 = 320 blocks (1.25 rounds) -> 16x32x80 = 256 blocks; the
    # SPADE convolution 128 -> 640 at 8 frames x 64^2: 16x32x128 = 320 blocks -> 16x32x80 = 512; choices with >= 0.8 of the last round stay
    assert qcode(16, 640, 640, 32, 32) == 600006 and qcode(8, 640, 640, 32, 32) == 600002
    assert qcode(8, 128, 640, 64, 64) == 600006 and qcode(16, 128, 640, 64, 64) == 600008        # 640 of 768 slots: stays
    assert qcode(16, 320, 320, 64, 64) == 600006 and qcode(16, 1280, 1280, 16, 16) == 600005     # 16^2 at 16 frames: 8x16x160 tiles = 256 blocks, no K split
    assert qcode(5, 320, 320, 64, 64) == 600007 and qcode(10, 320, 320, 64, 64) == 600006        # no full-round alternative: the table's choice

    def rcode(frames, cin, cout, h, w, act):
        p = hip.MgldIGemm()
        p.mode, p.M, p.N, p.K, p.batch, p.tap_inner, p.act = hip.MODE_CONV3X3, frames * h * w, cout, 9 * cin, 1, 2, act
        p.Cin, p.Hin, p.Win, p.Hout, p.Wout, p.stride, p.pad_t, p.pad_l, p.up2 = cin, h, w, h, w, 1, 1, 1, 0
        return hip.igemm_config(p) % 1000000
    # SPADE's shared convolution + ReLU rides the ping-pong kernel since round 6; activations its epilogue does not hold stay on conv3q
    assert rcode(16, 256, 128, 64, 64, hip.ACT_RELU) == 600002 and rcode(16, 256, 128, 64, 64, hip.ACT_SILU) == 600002
    assert 400000 <= rcode(16, 256, 128, 64, 64, hip.ACT_LRELU02) < 500000

    # statistics output of the producer (MgldIGemm.gn_part): tiles per frame where the picked kernel writes it, 0 where it does not
    def chunks(frames, cin, cout, h, w, tune=0, mode=hip.MODE_CONV3X3):
        p = hip.MgldIGemm()
        p.mode, p.M, p.N, p.K, p.batch, p.tap_inner, p.tune = mode, frames * h * w, cout, (9 if mode == hip.MODE_CONV3X3 else 1) * cin, 1, 2 if mode == hip.MODE_CONV3X3 else 0, tune
        if mode == hip.MODE_CONV3X3:
            p.Cin, p.Hin, p.Win, p.Hout, p.Wout, p.stride, p.pad_t, p.pad_l, p.up2 = cin, h, w, h, w, 1, 1, 1, 0
        p.lda, p.ldc, p.ldw = cin, cout, p.K
        return hip.lib().mgld_igemm_gn_chunks(C.byref(p))
    import ctypes as C
    assert chunks(8, 320, 320, 64, 64) == 8                    # 16 x 32-pixel tiles: 2 x 4 per 64 x 64 frame
    assert chunks(8, 640, 640, 32, 32) == 4                    # 8 x 32-pixel tiles
    assert chunks(8, 1280, 1280, 16, 16) == 0                  # conv3q level
    assert chunks(8, 1280, 1280, 8, 8, tune=40) == 0           # frame-stacked tiles span frames
    assert chunks(8, 320, 320, 64, 64, mode=hip.MODE_LINEAR) == 0
    assert hip.gn_apply_chunks(8, 4096, 320, 32) > 0 and hip.gn_apply_chunks(0, 4096, 320, 32) == 0


def _spliter_case(ImageSpliterTh, g, sf, to_dev=lambda t: t):
    """replay the generator's per-patch stand-in model (tests/golden/make_golden.py::gen_spliter) through a spliter class"""
    import torch.nn.functional as F
    im = torch.from_numpy(g[f"it_im_sf{sf}"])
    sp = ImageSpliterTh(to_dev(im), 24, 16, sf=sf)
    idx, k = [], 0
    for pch, index_infos in sp:
        res = F.interpolate(pch.cpu() * (1.0 + 0.1 * k) + 0.01 * k, scale_factor=sf, mode="nearest")
        sp.update(to_dev(res), index_infos)
        idx.append(list(index_infos))
        k += 1
    return np.array(idx), sp.gather().cpu()


def test_spliter_vs_reference_fixture():
    """H3: ImageSpliterTh start lists, iteration order, update() and gather() against outputs of the reference class
    (scripts/util_image.py:686-769) captured in tests/golden/g_spliter.npz"""
    from scripts.util_image import ImageSpliterTh
    g = np.load(os.path.join(HERE, "golden", "g_spliter.npz"))
    for key in g.files:
        if key.startswith("h_"):
            L, size, stride = (int(v) for v in key.split("_")[1:])
            sp = ImageSpliterTh(torch.zeros(1, 1, L, L + 8), size, stride, sf=1)
            assert sp.height_starts_list == g[key].tolist(), key
            assert sp.width_starts_list == g["w" + key[1:]].tolist(), key
            assert len(sp) == len(g[key]) * len(g["w" + key[1:]])
    for sf in (1, 2):
        idx, out = _spliter_case(ImageSpliterTh, g, sf)
        assert (idx == g[f"it_index_sf{sf}"]).all()
        assert torch.equal(out, torch.from_numpy(g[f"it_gather_sf{sf}"]))       # same fp32 arithmetic: bit-equal


def test_product_path_never_imports_oracle():
    """the oracle is test infrastructure: no product module may reference it"""
    bad = []
    for base in ("mgld_vsr_amd", "ldm", "basicsr", "scripts"):
        for dp, _, fs in os.walk(os.path.join(ROOT, base)):
            for f in fs:
                if f.endswith(".py"):
                    src = open(os.path.join(dp, f)).read()
                    if re.search(r"^\s*(from|import)\s+oracle\b", src, re.M):
                        bad.append(os.path.join(dp, f))
    assert not bad, bad


def test_text_oracle_matches_module_forward():
    """oracle/text.py vs the same tower evaluated through its nn modules the way open_clip's ResidualAttentionBlock does
    (x += attn(ln_1(x), attn_mask); x += mlp(ln_2(x))).  open_clip itself is not installed (oracle/text.py): this pins the
    functional restatement to torch's own MultiheadAttention, tests/test_oracle_golden.py::test_text_tower_golden pins it to
    transformers' CLIPTextModel."""
    import torch
    from mgld_vsr_amd import synth
    from mgld_vsr_amd.text import FrozenOpenCLIPEmbedder
    from oracle import text as otext
    emb = FrozenOpenCLIPEmbedder(layer="penultimate", context_dim=64, build_tower=True, heads=2, layers=3, vocab_size=100)
    synth.fill_module_(emb, "clip")
    toks = emb.tokenize(["", ""])
    toks[1, 2:6] = torch.tensor([5, 17, 3, 99])
    m = emb.model
    with torch.no_grad():
        x = (m.token_embedding(toks) + m.positional_embedding).permute(1, 0, 2)
        mask = otext.build_attention_mask(77)
        for blk in list(m.transformer.resblocks)[:-1]:
            h = blk.ln_1(x)
            x = x + blk.attn(h, h, h, need_weights=False, attn_mask=mask)[0]
            x = x + blk.mlp(blk.ln_2(x))
        ref = m.ln_final(x.permute(1, 0, 2))
        got = otext.encode_with_transformer(emb.state_dict(), toks, heads=2, layer_idx=1)
    assert torch.allclose(got, ref, atol=1e-5)
    keys = set(emb.state_dict())
    assert {"model.token_embedding.weight", "model.positional_embedding", "model.transformer.resblocks.0.attn.in_proj_weight",
            "model.transformer.resblocks.2.mlp.c_proj.bias", "model.ln_final.weight", "model.text_projection",
            "model.logit_scale"} <= keys


def _reduced_cfgs(n_frames=T):
    from mgld_vsr_amd.pipeline import model_configs
    return model_configs(n_frames, unet_overrides=dict(model_channels=64, context_dim=64, semb_channels=64),
                         struct_overrides=dict(model_channels=64, out_channels=64, num_heads=1),
                         vae_overrides=dict(ch=32, resolution=64), context_dim=64)


def test_checkpoint_loads_before_respacing(tmp_path):
    """the reference checkpoint stores the schedule buffers at length 1000 and the script loads it into the
    1000-step model, respacing afterwards (oldcanvas_tile.py:91-108, 308-329).  VSRPipeline.load_checkpoint must take a
    checkpoint with 1000-long buffers at any --ddpm_steps, must USE the `cond_stage_model.*` text-tower weights the checkpoint
    carries (not the synthetic context), and must refuse to invent a context for real weights that come without a tower."""
    from mgld_vsr_amd import synth
    from mgld_vsr_amd.pipeline import VSRPipeline
    from mgld_vsr_amd.util import instantiate_from_config
    cfgs = _reduced_cfgs()
    src = instantiate_from_config(cfgs[0])                       # constructed with the full 1000-step schedule
    src.cond_stage_model.build_tower(layers=2, vocab_size=100, heads=2)
    for name, mod in (("unet", src.model.diffusion_model), ("structcond", src.structcond_stage_model),
                      ("first_stage", src.first_stage_model), ("clip", src.cond_stage_model)):
        synth.fill_module_(mod, name)
    sd = src.state_dict()
    assert sd["betas"].shape == (1000,) and "cond_stage_model.model.token_embedding.weight" in sd
    ck = tmp_path / "model.ckpt"
    torch.save({"state_dict": sd, "global_step": 1}, ck)
    pipe = VSRPipeline(num_frames=T, ddpm_steps=50, synthetic_weights=False, configs=cfgs)
    assert pipe.model.betas.shape == (50,)
    missing, unexpected = pipe.load_checkpoint(str(ck), verbose=False)
    assert not unexpected and not [k for k in missing if not k.startswith("flownet_model.")]
    m = pipe.model
    assert m.betas.shape == (50,) and len(m.ori_timesteps) == 50 and m.ori_timesteps[-1] == 999      # respaced AFTER loading
    assert pipe.sqrt_alphas_cumprod.shape == (1000,)
    k = "model.diffusion_model.input_blocks.1.0.in_layers.2.weight"
    assert torch.equal(m.state_dict()[k], sd[k])
    tower = m.cond_stage_model.model
    assert tower is not None and len(tower.transformer.resblocks) == 2
    assert torch.equal(tower.token_embedding.weight, sd["cond_stage_model.model.token_embedding.weight"])
    # real weights without a text tower and without a precomputed context: fail hard, never the synthetic context
    sd2 = {k: v for k, v in sd.items() if not k.startswith("cond_stage_model.")}
    pipe2 = VSRPipeline(num_frames=T, ddpm_steps=4, synthetic_weights=False, configs=cfgs)
    pipe2.load_checkpoint({"state_dict": sd2}, verbose=False)
    with pytest.raises(RuntimeError):
        pipe2.model.cond_stage_model([""])
    pipe3 = VSRPipeline(num_frames=T, ddpm_steps=4, synthetic_weights=False, configs=cfgs)
    ctx = torch.randn(1, 77, 64)
    pipe3.load_checkpoint({"state_dict": sd2}, context=ctx, verbose=False)
    assert torch.equal(pipe3.model.cond_stage_model([""]), ctx)


def test_frame_writer_writes_everything_and_reraises(tmp_path):
    """preproc.FrameWriter: PNG / .npy output on a thread pool; close() waits for every file and surfaces the first failure"""
    from PIL import Image
    from mgld_vsr_amd.preproc import FrameWriter
    rng = np.random.default_rng(0)
    imgs = [(rng.random((24, 40, 3)) * 255).astype(np.uint8) for _ in range(6)]
    with FrameWriter(workers=3) as w:
        for k, im in enumerate(imgs):
            w.png(str(tmp_path / f"{k:02d}.png"), im)
            w.npy(str(tmp_path / f"{k:02d}.npy"), im[:2, :2].astype(np.float32))
    for k, im in enumerate(imgs):
        assert np.array_equal(np.asarray(Image.open(tmp_path / f"{k:02d}.png")), im)
        assert np.array_equal(np.load(tmp_path / f"{k:02d}.npy"), im[:2, :2].astype(np.float32))
    with FrameWriter(workers=4) as w:          # the same path twice (repeat-last padding of a short segment): the later write wins
        for im in imgs:
            w.png(str(tmp_path / "dup.png"), im)
    assert np.array_equal(np.asarray(Image.open(tmp_path / "dup.png")), imgs[-1])
    w = FrameWriter(workers=1)
    w.png(str(tmp_path / "missing_dir" / "x.png"), imgs[0])
    w.png(str(tmp_path / "ok.png"), imgs[1])
    with pytest.raises(FileNotFoundError):
        w.close()
    assert (tmp_path / "ok.png").exists()


# ---- BPE tokenizer (modules.py:174 `open_clip.tokenize`) ----------------------------------------------------------
_TOY_MERGES = [("h", "e"), ("l", "l"), ("he", "ll"), ("hell", "o</w>"), ("w", "o"), ("r", "l"), ("wo", "rl"), ("worl", "d</w>"),
               ("a", "n"), ("an", "d</w>"), ("t", "h"), ("th", "e</w>"), ("i", "n"), ("in", "g</w>"), ("e", "r"), ("o", "n</w>")]


def _bpe_bruteforce(word, merges):
    """independent statement of greedy BPE: repeatedly merge the adjacent pair of lowest rank (first occurrence class: all occurrences,
    left to right), until no adjacent pair is in the table"""
    sym = list(word[:-1]) + [word[-1] + "</w>"]
    rank = {m: i for i, m in enumerate(merges)}
    while len(sym) > 1:
        pairs = [(rank.get((a, b), 1 << 30), i) for i, (a, b) in enumerate(zip(sym[:-1], sym[1:]))]
        best = min(pairs)[0]
        if best == 1 << 30:
            break
        a, b = merges[best]
        out, i = [], 0
        while i < len(sym):
            if i < len(sym) - 1 and sym[i] == a and sym[i + 1] == b:
                out.append(a + b)
                i += 2
            else:
                out.append(sym[i])
                i += 1
        sym = out
    return sym


def test_bpe_tokenizer_algorithm():
    from mgld_vsr_amd.tokenizer import SimpleTokenizer, bytes_to_unicode
    tk = SimpleTokenizer(merges=_TOY_MERGES, vocab_size=256 + 256 + len(_TOY_MERGES) + 2)
    b2u = bytes_to_unicode()
    assert len(b2u) == 256 and len(set(b2u.values())) == 256 and b2u[ord("a")] == "a" and b2u[ord(" ")] != " "
    assert len(tk.encoder) == 512 + len(_TOY_MERGES) + 2 and tk.sot == len(tk.encoder) - 2 and tk.eot == len(tk.encoder) - 1
    # whole words that are in the table become ONE token; case and whitespace are normalised; punctuation splits off
    ids = tk.encode("  Hello   WORLD!! ")
    assert ids[0] == tk.encoder["hello</w>"] and ids[1] == tk.encoder["world</w>"] and tk.decode(ids) == "hello world !! "
    # greedy lowest-rank merging == the brute-force statement, on random lower-case words
    import random
    rnd = random.Random(7)
    for _ in range(300):
        wrd = "".join(rnd.choice("helowrdantig") for _ in range(rnd.randint(1, 9)))
        assert tk.bpe(wrd).split(" ") == _bpe_bruteforce(wrd, _TOY_MERGES), wrd
    # contractions and digits follow the CLIP pattern: 's is its own pre-token (two symbols here: the toy table has no merge for
    # it), every digit is its own token
    assert [tk.decoder[i] for i in tk.encode("the ring's 42")] == ["the</w>", "r", "ing</w>", "'", "s</w>", "4</w>", "2</w>"]
    # utf-8 bytes outside ASCII go through the byte table and round-trip
    assert tk.decode(tk.encode("café 中")) == "café 中 "
    # tokenize: [SOT, ..., EOT, 0 ...]; the empty prompt is [SOT, EOT]; over-long prompts are cut and still end in EOT
    t = tk.tokenize(["", "hello and the world", "hello " * 100], context_length=12)
    assert t.dtype == torch.long and t.shape == (3, 12)
    assert t[0].tolist() == [tk.sot, tk.eot] + [0] * 10
    assert t[1, 0] == tk.sot and t[1, 5] == tk.eot and int((t[1] != 0).sum()) == 6
    assert t[2, 0] == tk.sot and t[2, -1] == tk.eot and bool((t[2, 1:-1] == tk.encoder["hello</w>"]).all())


def test_text_embedder_tokenizes_with_a_merge_table_file(tmp_path, monkeypatch):
    """the embedder's tokenize(): empty prompt without any table; other prompts through the gzip'ed merge table open_clip ships (same file
    format: a header line, then one merge per line), located through $MGLD_BPE_VOCAB"""
    import gzip
    from ldm.modules.encoders.modules import FrozenOpenCLIPEmbedder
    from mgld_vsr_amd import tokenizer
    vs = 512 + len(_TOY_MERGES) + 2
    emb = FrozenOpenCLIPEmbedder(context_dim=64, vocab_size=vs)
    assert emb.tokenize([""])[0].tolist()[:3] == [vs - 2, vs - 1, 0]
    monkeypatch.delenv("MGLD_BPE_VOCAB", raising=False)
    if tokenizer.find_vocab() is None:
        with pytest.raises(NotImplementedError):
            emb.tokenize(["hello"])
    path = tmp_path / tokenizer.VOCAB_FILE
    with gzip.open(path, "wt", encoding="utf-8") as fh:
        fh.write('"bpe_simple_vocab_16e6.txt#version: 0.2\n' + "\n".join(" ".join(m) for m in _TOY_MERGES) + "\nz z\n")   # one line past the cut
    monkeypatch.setenv("MGLD_BPE_VOCAB", str(path))
    emb2 = FrozenOpenCLIPEmbedder(context_dim=64, vocab_size=vs)
    t = emb2.tokenize(["Hello world", ""])
    assert t.shape == (2, 77) and t[0, :4].tolist() == [vs - 2, 512 + 3, 512 + 7, vs - 1] and t[1, :2].tolist() == [vs - 2, vs - 1]


def test_fullwidth_checkpoint_every_reference_key_loads(tmp_path, monkeypatch):
    """SURVEY 8(f) row 3 / VERDICT r2 "missing" #2: a FULL-width `{"state_dict": ...}` holding EVERY key of the reference's own model
    (names, shapes, dtypes read from the reference classes: tests/golden/g_ckpt_keys.json — UNet, struct-cond encoder, first-stage VAE,
    RAFT_SR `flownet_model.*`, the 1000-long schedule buffers) plus open_clip's ViT-H-14 text tower under `cond_stage_model.model.*`
    goes through VSRPipeline.load_checkpoint the way the script loads it (oldcanvas_tile.py:91-108: strict=False) with NOTHING missing
    and NOTHING unexpected, the schedule is respaced afterwards, and the video VAE's keys load strictly through init_from_ckpt from a
    Lightning-style file (`state_dict` next to pickled non-tensor objects, autoencoder.py:1652-1672)."""
    import json as _json
    from mgld_vsr_amd.pipeline import VSRPipeline, model_configs
    with open(os.path.join(HERE, "golden", "g_ckpt_keys.json")) as fh:
        keys = _json.load(fh)

    def fake(entries):      # stride-0 views: the 1.4 G parameters of the state dict cost no memory, load_state_dict copies from them
        return {k: torch.zeros((), dtype=getattr(torch, dt)).expand(*shape) if shape else torch.zeros((), dtype=getattr(torch, dt))
                for k, shape, dt in entries}
    sd = fake(keys["ldm"] + keys["cond_stage_openclip_vit_h_14_text"])
    assert sd["betas"].shape == (1000,) and sd["cond_stage_model.model.token_embedding.weight"].shape == (49408, 1024)
    assert any(k.startswith("flownet_model.") for k in sd) and any(k.startswith("first_stage_model.decoder.") for k in sd)
    pipe = VSRPipeline(num_frames=5, ddpm_steps=50, synthetic_weights=False, configs=model_configs(5))
    missing, unexpected = pipe.load_checkpoint({"state_dict": sd, "global_step": 7}, verbose=False)
    assert list(missing) == [] and list(unexpected) == [], (list(missing)[:5], list(unexpected)[:5])
    m = pipe.model
    assert m.betas.shape == (50,) and len(m.ori_timesteps) == 50 and pipe.sqrt_alphas_cumprod.shape == (1000,)
    tower = m.cond_stage_model.model
    assert tower is not None and len(tower.transformer.resblocks) == 24 and tower.token_embedding.weight.shape == (49408, 1024)
    # and nothing of the model was left without a checkpoint entry (every parameter / persistent buffer has a key in the reference list)
    assert set(m.state_dict().keys()) - set(sd.keys()) == set()

    # the video VAE: a Lightning-style checkpoint file with a pickled callback object of a module that is not importable here
    import sys
    import types
    mod = types.ModuleType("pytorch_lightning_fake_callbacks")

    def _init(self):
        self.best_k_models, self.monitor = {"a": 1.0}, "val/rec_loss"
    ModelCheckpoint = type("ModelCheckpoint", (), {"__init__": _init, "__module__": mod.__name__, "__qualname__": "ModelCheckpoint"})
    mod.ModelCheckpoint = ModelCheckpoint
    sys.modules[mod.__name__] = mod
    vsd = {k: (torch.zeros(shape, dtype=getattr(torch, dt)) + 0.25) for k, shape, dt in keys["vq"]}
    ck = tmp_path / "vqgan.ckpt"
    try:
        torch.save({"state_dict": vsd, "callbacks": {"ckpt": ModelCheckpoint()}, "epoch": 11, "hyper_parameters": {"lr": 1e-4}}, ck)
    finally:
        del sys.modules[mod.__name__]
    from mgld_vsr_amd.util import load_trusted_checkpoint
    # torch >= 2.6's weights_only loader refuses the callback object, and the loader does NOT fall back to the full unpickler silently:
    # without the opt-in it raises and names the file and the switch
    monkeypatch.delenv("MGLD_TRUST_CKPT", raising=False)
    with pytest.raises(RuntimeError, match="MGLD_TRUST_CKPT=1"):
        load_trusted_checkpoint(str(ck))
    monkeypatch.setenv("MGLD_TRUST_CKPT", "1")
    with pytest.warns(UserWarning, match="FULL unpickler"):
        raw = load_trusted_checkpoint(str(ck))
    assert raw["epoch"] == 11 and set(raw["state_dict"].keys()) == set(vsd.keys())
    # a plain state-dict file needs no opt-in
    plain = tmp_path / "plain.ckpt"
    torch.save({"state_dict": {"w": torch.ones(3)}, "epoch": 2}, plain)
    monkeypatch.delenv("MGLD_TRUST_CKPT", raising=False)
    assert load_trusted_checkpoint(str(plain))["epoch"] == 2
    monkeypatch.setenv("MGLD_TRUST_CKPT", "1")
    assert list(pipe.vq_model.init_from_ckpt(str(ck))) == []
    assert pipe.vq_model.last_load == ([], [])
    assert set(pipe.vq_model.state_dict().keys()) == set(vsd.keys())
    assert float(pipe.vq_model.decoder.conv_out.weight.mean()) == 0.25


def test_pmc_traffic_table_matches_kernel_sources():
    """bench.py reports `roofline.traffic` only from a PMC table taken on the CURRENT GEMM-family / attention sources (sha256 key).  A stale
    table is not an error (the field is then null) — but it should be noticed here, not in the driver's bench line."""
    import hashlib
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path = os.path.join(root, "profiles", "r03_pmc_traffic.json")
    if not os.path.exists(path):
        pytest.skip("no PMC table")
    src = b"".join(open(os.path.join(root, "mgld_vsr_amd", "csrc", f), "rb").read()
                   for f in ("igemm_common.h", "igemm.hip", "conv3q.hip", "attention.hip"))
    with open(path) as fh:
        key = json.load(fh)["gemm_src_sha16"]
    if key != hashlib.sha256(src).hexdigest()[:16]:
        pytest.skip("profiles/r03_pmc_traffic.json was taken on other kernel sources: re-run tools/pmc_traffic.sh on the GPU")


def test_pipeline_clone_shares_weights_not_modules():
    """VSRPipeline.clone_shared(): the instances SegmentPool keeps in flight share parameter / buffer storage (one model in host memory) but
    own their module objects; refused once the source has an engine attached."""
    from mgld_vsr_amd.pipeline import VSRPipeline, model_configs
    cfgs = model_configs(T, unet_overrides=dict(model_channels=32, context_dim=32, semb_channels=32, num_head_channels=32),
                         struct_overrides=dict(model_channels=32, out_channels=32, num_heads=1), vae_overrides=dict(ch=32), context_dim=32)
    a = VSRPipeline(num_frames=T, ddpm_steps=4, configs=cfgs)
    b = a.clone_shared()
    pa, pb = dict(a.model.named_parameters()), dict(b.model.named_parameters())
    assert pa.keys() == pb.keys() and len(pa) > 100
    assert all(pa[k].data_ptr() == pb[k].data_ptr() for k in pa)
    va, vb = dict(a.vq_model.named_parameters()), dict(b.vq_model.named_parameters())
    assert all(va[k].data_ptr() == vb[k].data_ptr() for k in va)
    ba, bb = dict(a.model.named_buffers()), dict(b.model.named_buffers())
    assert all(ba[k].data_ptr() == bb[k].data_ptr() for k in ba)
    mods_a = {id(m) for m in a.model.modules()} | {id(m) for m in a.vq_model.modules()}
    assert not any(id(m) in mods_a for m in b.model.modules()) and not any(id(m) in mods_a for m in b.vq_model.modules())
    assert b.model._engine is None and b.ddpm_steps == a.ddpm_steps and b.model.num_timesteps == a.model.num_timesteps
    b.vq_model.decoder.fusion_w = 0.25                       # per-instance attribute
    assert a.vq_model.decoder.fusion_w != 0.25
    a.model._engine = object()
    with pytest.raises(RuntimeError):
        a.clone_shared()
    a.model._engine = None


def test_layernorm_fold_packing_is_the_same_linear_map():
    """engine.pack_ln_fold (MgldIGemm.ln_part): LayerNorm(x) W^T + b == rstd (x W'^T - mean s) + b' on the raw rows, in fp64 — plain and in the
    packed GEGLU row order (the fold commutes with any row permutation of W)"""
    import torch.nn.functional as F
    from mgld_vsr_amd.engine import pack_geglu, pack_ln_fold
    g = torch.Generator().manual_seed(5)
    M, C, N = 37, 96, 128
    x = (torch.randn(M, C, generator=g) * 3 + 1.5).double()
    W, b = torch.randn(N, C, generator=g).double() * C ** -0.5, torch.randn(N, generator=g).double()
    gam, bet = 1 + 0.3 * torch.randn(C, generator=g).double(), 0.3 * torch.randn(C, generator=g).double()
    ref = F.layer_norm(x, (C,), gam, bet, 1e-5) @ W.t() + b
    mean, rstd = x.mean(1, keepdim=True), (x.var(1, unbiased=False, keepdim=True) + 1e-5).rsqrt()
    for Wk, bk, refk in ((W, b, ref), pack_geglu(W, b) + (None,)):
        if refk is None:
            refk = F.layer_norm(x, (C,), gam, bet, 1e-5) @ Wk.t() + bk
        Wp, s, b2 = pack_ln_fold(Wk, bk, gam, bet)
        # s is taken from the fp16-rounded rows on purpose: evaluate the identity with the matrix the MFMA sees
        W16 = Wp.to(torch.float16).double()
        got = rstd * (x @ W16.t() - mean * s.double()) + b2
        want = rstd * ((x - mean) @ W16.t()) + b2
        assert float((got - want).abs().max()) < 1e-4                  # the mean correction cancels what was summed (fp32 row sums)
        assert float((rstd * (x @ Wp.t() - mean * Wp.sum(1)) + b2 - refk).abs().max()) < 1e-9
    W0, s0, b0 = pack_ln_fold(W, None, gam, bet)
    assert torch.equal(b0, W @ bet)
