"""LatentDiffusionVSRTextWT: the respaced DDPM sampling loop with motion guidance, on the HIP engine.

Interface mirror of ldm/models/diffusion/ddpm.py: DDPM.register_schedule (:237-292), q_sample_respace (:403-406),
LatentDiffusionVSRTextWT (:3166-4909: encode_first_stage / get_first_stage_encoding :3382-3389,3906-3943,
compute_temporal_condition_v4 :3538-3574, apply_model :3984-4085, p_sample_loop / sample :4501-4599,4696-4719,
sample_canvas :4722-4746 with _gaussian_weights :4601-4616), DiffusionWrapper (:4911-4940).

MI355X design: the whole reverse step (struct-cond encoder -> UNet -> posterior -> guidance) is a fixed launch
sequence over a bump arena; after the first eager step it is captured into ONE hipGraph and replayed for the
remaining steps with no host synchronisation — per-step scalars (schedule coefficients, the network timestep, the
noise slice) are read on the device through a step index that the graph itself decrements.
"""
from contextlib import contextmanager

import os

import numpy as np
import torch
import torch.nn as nn

from . import hip
from .engine import Act, Engine, GraphPieces
from .util import default, instantiate_from_config
from .vae import DiagonalGaussianDistribution


def make_beta_schedule(schedule, n_timestep, linear_start=1e-4, linear_end=2e-2, cosine_s=8e-3):
    """ldm/modules/diffusionmodules/util.py:21-43 ('linear' is the only schedule the VSR path uses)."""
    if schedule != "linear":
        raise NotImplementedError(f"beta schedule '{schedule}' is not on the MGLD-VSR hot path")
    return (torch.linspace(linear_start ** 0.5, linear_end ** 0.5, n_timestep, dtype=torch.float64) ** 2).numpy()


def space_timesteps(num_timesteps, section_counts):
    """ddpm.py:101-154."""
    if isinstance(section_counts, str):
        section_counts = [int(x) for x in section_counts.split(",")]
    size_per, extra = num_timesteps // len(section_counts), num_timesteps % len(section_counts)
    start_idx, all_steps = 0, []
    for i, cnt in enumerate(section_counts):
        size = size_per + (1 if i < extra else 0)
        if size < cnt:
            raise ValueError(f"cannot divide section of {size} steps into {cnt}")
        stride = 1 if cnt <= 1 else (size - 1) / (cnt - 1)
        cur = 0.0
        for _ in range(cnt):
            all_steps.append(start_idx + round(cur))
            cur += stride
        start_idx += size
    return set(all_steps)


def extract_into_tensor(a, t, x_shape):
    b = t.shape[0]
    return a.gather(-1, t).reshape(b, *((1,) * (len(x_shape) - 1)))


class DiffusionWrapper(nn.Module):
    """ddpm.py:4911-4940 (crossattn conditioning)."""

    def __init__(self, diff_model_config, conditioning_key):
        super().__init__()
        self.diffusion_model = instantiate_from_config(diff_model_config)
        self.conditioning_key = conditioning_key
        assert conditioning_key == "crossattn", "MGLD-VSR uses cross-attention conditioning"

    def forward(self, x, t, c_concat=None, c_crossattn=None, struct_cond=None, seg_cond=None):
        cc = torch.cat(c_crossattn, 1)
        return self.diffusion_model(x, t, context=cc, struct_cond=struct_cond)


class LatentDiffusionVSRTextWT(nn.Module):
    """Drop-in for `ldm.models.diffusion.ddpm.LatentDiffusionVSRTextWT` on the inference path."""

    def __init__(self, first_stage_config, cond_stage_config, structcond_stage_config, flownet_config, unet_config=None,
                 num_frames=1, num_timesteps_cond=None, cond_stage_key="image", cond_stage_trainable=False, concat_mode=True,
                 cond_stage_forward=None, conditioning_key=None, scale_factor=1.0, scale_by_std=False,
                 train_temporal_module=True, unfrozen_diff=False, random_size=False, test_gt=False, p2_gamma=None, p2_k=None,
                 time_replace=None, use_usm=False, mix_ratio=0.0, timesteps=1000, beta_schedule="linear",
                 linear_start=1e-4, linear_end=2e-2, cosine_s=8e-3, image_size=256, channels=3, log_every_t=100,
                 clip_denoised=True, parameterization="eps", use_ema=True, first_stage_key="image", v_posterior=0.,
                 ckpt_path=None, ignore_keys=(), **kwargs):
        super().__init__()
        assert parameterization == "eps" and v_posterior == 0.
        self.num_frames, self.channels, self.image_size = num_frames, channels, image_size
        self.scale_factor = scale_factor
        self.log_every_t, self.time_replace = log_every_t, time_replace
        self.clip_denoised = False  # ddpm.py:3228
        self.use_ema = False
        self.parameterization, self.v_posterior = parameterization, v_posterior
        self.cond_stage_key, self.first_stage_key = cond_stage_key, first_stage_key
        self.model = DiffusionWrapper(unet_config, conditioning_key or ("concat" if concat_mode else "crossattn"))
        self.first_stage_model = instantiate_from_config(first_stage_config)
        self.cond_stage_model = instantiate_from_config(cond_stage_config)
        self.structcond_stage_model = instantiate_from_config(structcond_stage_config)
        self.flownet_model = instantiate_from_config(flownet_config)
        self.register_schedule(beta_schedule=beta_schedule, timesteps=timesteps, linear_start=linear_start,
                               linear_end=linear_end, cosine_s=cosine_s)
        # time respacing as in the reference constructor (ddpm.py:3280-3295)
        if self.time_replace is None:
            self.time_replace = timesteps
        use = set(space_timesteps(timesteps, [self.time_replace]))
        self._respace(use, linear_start, linear_end)
        self.configs = None
        self._engine = None
        self._stream = None
        # struct-cond features of all steps in batched passes before the loop (see _precompute_structcond); MGLD_SC_PRECOMPUTE=0
        # keeps the encoder inside every step (rocprofv3 --pmc passes, A/B timing)
        self.precompute_structcond = os.environ.get("MGLD_SC_PRECOMPUTE", "1") != "0"
        self._graph = None
        self._graph_key = None
        if ckpt_path is not None:
            self.init_from_ckpt(ckpt_path, ignore_keys)

    # ---- schedule -------------------------------------------------------------------------------------------------
    def register_schedule(self, given_betas=None, beta_schedule="linear", timesteps=1000, linear_start=1e-4,
                          linear_end=2e-2, cosine_s=8e-3):
        """ddpm.py:237-292: float64 (or the dtype of given_betas) math, float32 buffers."""
        betas = given_betas if given_betas is not None else make_beta_schedule(beta_schedule, timesteps, linear_start,
                                                                             linear_end, cosine_s)
        betas = np.asarray(betas)
        alphas = 1. - betas
        ac = np.cumprod(alphas, axis=0)
        ac_prev = np.append(1., ac[:-1])
        self.num_timesteps = int(betas.shape[0])
        self.linear_start, self.linear_end = linear_start, linear_end
        t32 = lambda a: torch.tensor(a, dtype=torch.float32)
        post_var = (1 - self.v_posterior) * betas * (1. - ac_prev) / (1. - ac) + self.v_posterior * betas
        for name, val in [
            ("betas", betas), ("alphas_cumprod", ac), ("alphas_cumprod_prev", ac_prev),
            ("sqrt_alphas_cumprod", np.sqrt(ac)), ("sqrt_one_minus_alphas_cumprod", np.sqrt(1. - ac)),
            ("log_one_minus_alphas_cumprod", np.log(1. - ac)), ("sqrt_recip_alphas_cumprod", np.sqrt(1. / ac)),
            ("sqrt_recipm1_alphas_cumprod", np.sqrt(1. / ac - 1)), ("posterior_variance", post_var),
            ("posterior_log_variance_clipped", np.log(np.maximum(post_var, 1e-20))),
            ("posterior_mean_coef1", betas * np.sqrt(ac_prev) / (1. - ac)),
            ("posterior_mean_coef2", (1. - ac_prev) * np.sqrt(alphas) / (1. - ac)),
        ]:
            if name in self._buffers:
                self._buffers[name] = t32(val)
            else:
                self.register_buffer(name, t32(val))
        self._graph_key = None

    def _respace(self, use_timesteps, linear_start, linear_end):
        last, new_betas = 1.0, []
        for i, ac in enumerate(self.alphas_cumprod):
            if i in use_timesteps:
                new_betas.append(1 - ac / last)
                last = ac
        new_betas = [b.data.cpu().numpy() for b in new_betas]
        self.register_schedule(given_betas=np.array(new_betas), timesteps=len(new_betas), linear_start=linear_start,
                               linear_end=linear_end)
        self.ori_timesteps = sorted(list(use_timesteps))

    @contextmanager
    def ema_scope(self, context=None):
        yield None  # use_ema=False in the shipped config (mgldvsr_512_realbasicvsr_deg.yaml:21)

    def init_from_ckpt(self, path, ignore_keys=(), only_model=False):
        from .util import load_trusted_checkpoint
        sd = load_trusted_checkpoint(path)
        if "state_dict" in sd:
            sd = sd["state_dict"]
        for k in list(sd.keys()):
            if any(k.startswith(ik) for ik in ignore_keys):
                del sd[k]
        return self.load_state_dict(sd, strict=False)

    # ---- engine ----------------------------------------------------------------------------------------------------
    def engine(self):
        if self._engine is None:
            self._engine = Engine()
            for m in (self.model.diffusion_model, self.structcond_stage_model, self.first_stage_model):
                if hasattr(m, "set_engine"):
                    m.set_engine(self._engine)
        return self._engine

    @property
    def device(self):
        return self.engine().device

    def cuda(self, device=None):  # parameters stay on the host; packed fp16 copies live in the engine cache
        return self

    def to(self, *a, **k):
        return self

    # ---- first stage / conditioning ---------------------------------------------------------------------------------
    @torch.no_grad()
    def encode_first_stage(self, x):
        self.engine()
        return self.first_stage_model.encode(x)

    def get_first_stage_encoding(self, encoder_posterior, noise=None):
        if isinstance(encoder_posterior, DiagonalGaussianDistribution):
            z = encoder_posterior.sample(noise) if noise is not None else encoder_posterior.sample()
        elif isinstance(encoder_posterior, torch.Tensor):
            z = encoder_posterior
        else:
            raise NotImplementedError(type(encoder_posterior))
        return self.scale_factor * z

    @torch.no_grad()
    def decode_first_stage(self, z, predict_cids=False, force_not_quantize=False):
        """ddpm.py:3786-3850 (KL first stage, no split_input_params): first_stage_model.decode(z / scale_factor)"""
        if predict_cids:
            raise NotImplementedError("predict_cids belongs to VQ first stages; the VSR model's first stage is a KL autoencoder")
        self.engine()
        return self.first_stage_model.decode(z * (1.0 / self.scale_factor))

    def get_learned_conditioning(self, c):
        return self.cond_stage_model(c)

    def q_sample_respace(self, x_start, t, sqrt_alphas_cumprod, sqrt_one_minus_alphas_cumprod, noise=None):
        noise = default(noise, lambda: torch.randn_like(x_start))
        dev = x_start.device
        return (extract_into_tensor(sqrt_alphas_cumprod.to(dev), t.to(dev), x_start.shape) * x_start +
                extract_into_tensor(sqrt_one_minus_alphas_cumprod.to(dev), t.to(dev), x_start.shape) * noise)

    # ---- schedule helpers inherited from DDPM in the reference (ddpm.py:340-353, 398-401): gathers over the fp32 buffers --
    def _gather(self, name, t, like):
        return extract_into_tensor(getattr(self, name).to(like.device), t.to(like.device), like.shape)

    def q_sample(self, x_start, t, noise=None):
        return self.q_sample_respace(x_start, t, self.sqrt_alphas_cumprod, self.sqrt_one_minus_alphas_cumprod, noise)

    def predict_start_from_noise(self, x_t, t, noise):
        return self._gather("sqrt_recip_alphas_cumprod", t, x_t) * x_t - self._gather("sqrt_recipm1_alphas_cumprod", t, x_t) * noise

    def q_posterior(self, x_start, x_t, t):
        mean = self._gather("posterior_mean_coef1", t, x_t) * x_start + self._gather("posterior_mean_coef2", t, x_t) * x_t
        return mean, self._gather("posterior_variance", t, x_t), self._gather("posterior_log_variance_clipped", t, x_t)

    def compute_flow(self, lrs, _inside_sampler=False):
        """ddpm.py:3404-3429: lrs [n,t,3,h,w] in [0,1] -> (flows_forward, flows_backward), each [n,t-1,2,h,w].  Both
        directions go through the flow network (RAFT_SR, mgld_vsr_amd/raft.py) as one batch of frame pairs.
        _inside_sampler: the call sits inside sample() / p_sample() (the `lr_images` term), where the flow network runs on the sampler's
        own engine: it must not rewind the shared arena over buffers the sampler already holds (RAFT_SR.forward keep_arena)."""
        from .raft import compute_flow
        if getattr(self.flownet_model, "_engine", None) is None and hasattr(self.flownet_model, "set_engine"):
            self.flownet_model.set_engine(self.engine())
        return compute_flow(self.flownet_model, lrs, keep_arena=_inside_sampler)

    @torch.no_grad()
    def compute_temporal_condition_v4(self, flows, latents, masks):
        """Scalar guidance loss (ddpm.py:3538-3574) evaluated by the HIP kernels."""
        eng = self.engine()
        ff, fb, fo, bo = self._flows_to_device(eng, flows, masks)
        z = latents.to(eng.device, torch.float32).contiguous()
        T, c, h, w = z.shape
        work = torch.empty(hip.guidance_work_bytes(T, c, h, w), dtype=torch.uint8, device=eng.device)
        loss = torch.empty(1, device=eng.device)
        hip.guidance_loss(z, ff, fb, fo, bo, loss, work)
        return loss[0]

    @torch.no_grad()
    def apply_model(self, x_noisy, t, cond, struct_cond, return_ids=False):
        if not isinstance(cond, dict):
            cond = {"c_crossattn": cond if isinstance(cond, list) else [cond]}
        return self.model(x_noisy, t, **cond, struct_cond=struct_cond)

    # ---- sampling ---------------------------------------------------------------------------------------------------
    @staticmethod
    def _flows_to_device(eng, flows, masks):
        ff, fb = flows
        fo, bo = masks
        assert ff.shape[0] == 1, "motion guidance operates on one clip (b = 1), as the reference scripts do"
        f32 = lambda t: t.to(eng.device, torch.float32).contiguous()
        ff, fb = f32(ff[0]), f32(fb[0])                       # [T-1, 2, h, w]
        fo, bo = f32(fo[0].reshape(fo.shape[1], *fo.shape[-2:])), f32(bo[0].reshape(bo.shape[1], *bo.shape[-2:]))
        return ff, fb, fo, bo

    def _coef_table(self, steps_total, use_t_replace):
        S = self.posterior_mean_coef1.shape[0]
        t = torch.zeros(S, 8)
        t[:, 0] = self.sqrt_recip_alphas_cumprod
        t[:, 1] = self.sqrt_recipm1_alphas_cumprod
        t[:, 2] = self.posterior_mean_coef1
        t[:, 3] = self.posterior_mean_coef2
        t[:, 4] = self.posterior_log_variance_clipped
        t[:, 5] = 1.0
        t[0, 5] = 0.0
        t[:, 6] = torch.tensor(self.ori_timesteps, dtype=torch.float32) if use_t_replace else torch.arange(S).float()
        return t

    def _gaussian_weights(self, tile_width, tile_height, nbatches=1):
        """ddpm.py:4601-4616 (float64, asymmetric midpoints)."""
        var = 0.01
        mid = (tile_width - 1) / 2
        xs = [np.exp(-(x - mid) * (x - mid) / (tile_width * tile_width) / (2 * var)) / np.sqrt(2 * np.pi * var)
              for x in range(tile_width)]
        mid = tile_height / 2
        ys = [np.exp(-(y - mid) * (y - mid) / (tile_height * tile_height) / (2 * var)) / np.sqrt(2 * np.pi * var)
              for y in range(tile_height)]
        w = torch.tensor(np.outer(ys, xs))
        return torch.tile(w, (nbatches, self.channels, 1, 1))

    @staticmethod
    def _tile_origins(h, w, tile_size, tile_overlap):
        """tile enumeration of p_mean_variance_canvas (ddpm.py:4205-4236), reference order."""
        def count(L):
            n, cur = 0, 0
            while cur < L:
                cur = max(n * tile_size - tile_overlap * n, 0) + tile_size
                n += 1
            return n
        rows, cols = count(w), count(h)
        out, ofs_x, ofs_y = [], 0, 0
        for row in range(rows):
            for col in range(cols):
                if col < cols - 1 or row < rows - 1:
                    ofs_x = max(row * tile_size - tile_overlap * row, 0)
                    ofs_y = max(col * tile_size - tile_overlap * col, 0)
                if row == rows - 1:
                    ofs_x = w - tile_size
                if col == cols - 1:
                    ofs_y = h - tile_size
                out.append((ofs_y, ofs_x))
        return out

    # ---- struct-cond features: a function of (LR latent, timestep) only, NOT of the sample ---------------------------
    STRUCTCOND_CHUNK = 10   # schedule steps evaluated per batched encoder pass (one time-embedding row each; <= 16)

    def _sc_chunk(self, n_frames):
        """schedule steps per batched pass: at most STRUCTCOND_CHUNK and about 96 frames (bounds the arena of a pass)"""
        return max(1, min(self.STRUCTCOND_CHUNK, 96 // max(1, n_frames)))

    HOIST_TABLE_BUDGET = 40 << 30   # bytes of HBM the hoisted struct-cond + SPADE tables of ONE window may take (288 GB per GPU)

    def _hoist_window(self, eng, lat_act, S):
        """schedule steps per hoisting window: the per-step tables (struct-cond features of every scale + the SPADE gamma|beta of
        the low-resolution blocks) are sized from the latent geometry and capped by HOIST_TABLE_BUDGET, so long schedules (the
        CLI's default --ddpm_steps 1000), large patches or many frames are hoisted window by window instead of allocating
        [S, ...] tables (132 GB of fp16 at 1000 steps x 45 latent tiles)."""
        unet, sc_net = self.model.diffusion_model, self.structcond_stage_model
        n, h, w = lat_act.n, lat_act.h, lat_act.w
        per_step, hh, ww = 0, h, w
        for _ in range(len(sc_net.input_block_chans)):
            per_step += n * hh * ww * sc_net.out_channels * 2
            hh, ww = max(1, hh // 2), max(1, ww // 2)
        for blk, bh, bw in unet.resblock_geometry(h, w):
            if bh * bw <= 1024:                                  # the blocks _precompute_spade hoists
                per_step += n * bh * bw * 2 * blk.out_channels * 2
        return int(max(1, min(S, self.HOIST_TABLE_BUDGET // max(1, per_step))))

    def _precompute_structcond(self, eng, st, lat_act, i_high, i_low):
        """The reference evaluates structcond_stage_model(lat, t) inside every step (ddpm.py:4344-4350); its inputs are
        the constant LR latent and the step's timestep, so the evaluations are known before sampling starts.  They are
        run here as a few large batched passes (STRUCTCOND_CHUNK steps x frames per pass: big, efficient GEMMs instead of
        ~350 latency-bound launches inside each step) into per-scale tables [window, frames*r*r, C] for the schedule indices
        i_low..i_high of one hoisting window; a step then only copies its slice (mgld_copy_step, indexed by a device-side
        window position that the step graph itself decrements, so the captured graph stays replayable across windows)."""
        sc_net = self.structcond_stage_model
        n, rows = lat_act.n, lat_act.rows
        Wn = st["hoist_window"]
        tables, dims = st.get("sc_tables", {}), st.get("sc_dims", {})
        chunk = self._sc_chunk(n)
        cnt = i_high - i_low + 1
        for c0 in range(0, cnt, chunk):
            k = min(chunk, cnt - c0)
            eng.reset()
            tv = st["coef"][i_low + c0:i_low + c0 + k, 6].contiguous()      # network timesteps of schedule indices i_low+c0 ..
            x = eng.arena.alloc((k * rows, lat_act.C), torch.float16)
            for j in range(k):
                hip.copy2d(lat_act.v, x[j * rows:(j + 1) * rows])
            res = sc_net.run(eng, Act(x, k * n, lat_act.h, lat_act.w), tv, n)   # one embedding row per step: n frames each
            for key, a in res.items():
                if key not in tables:
                    tables[key] = torch.empty((Wn, n * a.hw, a.C), dtype=torch.float16, device=eng.device)
                    dims[key] = (a.h, a.w)
                hip.copy2d(a.v, tables[key][c0:c0 + k].view(a.rows, a.C))
        st["sc_tables"], st["sc_dims"] = tables, dims
        if "sc_step" not in st:
            st["sc_step"] = {key: (torch.empty_like(t[0]), dims[key]) for key, t in tables.items()}
        st["sc_frames"] = n
        st["win_idx"].fill_(i_high - i_low)                                   # window position of the step about to run
        st["win"] = (i_high, i_low)

    def _precompute_spade(self, eng, st):
        """The SPADE modulation of a ResBlockDual, [gamma|beta] = conv(relu(conv(struct_cond))) (spade.py:93-104,
        openaimodel.py:481-482), depends on the struct-cond features only — like them it is known for every step before
        sampling.  For the low-resolution blocks (<= 32x32 latents, where the in-step convolutions are small and
        latency-bound) it is evaluated here in batched passes into per-block tables [window, frames*h*w, 2C] for the current
        hoisting window; the step's spade_apply then indexes the table by the device-side window position.  First called after
        the first (eager) step, which records the struct-cond scale each block reads."""
        from .unet import ResBlockDual
        if "sc_tables" not in st:
            return
        n, Wn = st["sc_frames"], st["hoist_window"]
        i_high, i_low = st["win"]
        if "spade_tabs" not in st:
            blocks = [m for m in self.model.diffusion_model.modules() if isinstance(m, ResBlockDual) and getattr(m, "_sc_key", None)]
            tabs = {}
            for blk in blocks:
                fh, fw = st["sc_step"][blk._sc_key][1]
                if fh * fw <= 1024:
                    tabs[id(blk)] = (blk, torch.empty((Wn, n * fh * fw, 2 * blk.out_channels), dtype=torch.float16, device=eng.device))
            st["spade_tabs"] = tabs
        tables = st["spade_tabs"]
        if not tables:
            return
        chunk = self._sc_chunk(n)
        cnt = i_high - i_low + 1
        for c0 in range(0, cnt, chunk):
            k = min(chunk, cnt - c0)
            eng.reset()
            for blk, tab in tables.values():
                src = st["sc_tables"][blk._sc_key]
                fh, fw = st["sc_step"][blk._sc_key][1]
                seg = Act(src[c0:c0 + k].view(k * n * fh * fw, src.shape[2]), k * n, fh, fw)
                blk.spade_modulation(eng, seg, out=Act(tab[c0:c0 + k].view(k * n * fh * fw, tab.shape[2]), k * n, fh, fw))   # straight into the table
        st["spade"] = {key: (tab, tab.shape[1] * tab.shape[2], st["win_idx"]) for key, (blk, tab) in tables.items()}

    def _structcond_of_step(self, eng, st, lat_act):
        if "sc_tables" not in st:
            return self.structcond_stage_model.run(eng, lat_act, st["tvals"], None)
        out = {"__spade__": st["spade"]} if "spade" in st else {}
        for key, tab in st["sc_tables"].items():
            buf, (fh, fw) = st["sc_step"][key]
            hip.copy_step(tab, buf, st["win_idx"])
            eng.launches += 1
            out[key] = Act(buf, st["sc_frames"], fh, fw)
        return out

    def _step_body(self, eng, st):
        """One reverse step as a pure launch sequence (graph-capturable)."""
        unet, sc_net = self.model.diffusion_model, self.structcond_stage_model
        eng.reset()
        hip.step_timestep(st["coef"], st["step_idx"], st["tvals"])
        x = st["x"]
        if st["tiles"] is None:
            xa = eng.from_nchw(x)
            sc = self._structcond_of_step(eng, st, st["lat_act"])
            eps = unet.run(eng, xa, st["tvals"], None, st["ctx"], sc).v
        else:
            ts = st["tile_size"]
            nt = len(st["tiles"])
            T, c = x.shape[0], x.shape[1]
            tsh = eng.tile_shard
            mine = list(enumerate(st["tiles"])) if tsh is None else list(enumerate(st["tiles"]))[tsh.k0:tsh.k1]
            nl = len(mine) if tsh is None else tsh.per          # local tile slots (zero-padded to `per` when sharded)
            xt = eng.arena.alloc((max(1, len(mine)) * T, c, ts, ts), torch.float32)
            for j, (k, (y0, x0)) in enumerate(mine):
                hip.crop(x, xt[j * T:(j + 1) * T], y0, x0)
            et = eng.arena.alloc((nl * T, c, ts, ts), torch.float32)
            if mine:
                xa = eng.from_nchw(xt[:len(mine) * T])
                sc = self._structcond_of_step(eng, st, st["lat_tiles"])
                e = unet.run(eng, xa, st["tvals"], None, st["ctx"], sc)
                hip.nhwc_to_nchw(e.v, et[:len(mine) * T])
            if tsh is not None:
                if len(mine) < nl:
                    et[len(mine) * T:].zero_()
                etl = et
                et = eng.arena.alloc((tsh.world * nl * T, c, ts, ts), torch.float32)
                eng.collective(lambda: tsh.gather(etl, out=et))   # every tile's eps on every rank (one exchange per step)
            acc, cnt = st["acc"], st["cnt"]
            acc.zero_()
            cnt.zero_()
            for k, (y0, x0) in enumerate(st["tiles"]):
                j = k if tsh is None else tsh.slot(k)
                hip.tile_accumulate(et[j * T:(j + 1) * T], st["wgt"], acc, cnt, y0, x0)
            eps = st["eps_canvas"]
            hip.tile_normalize(acc, cnt, eps)
        if st["guided"]:
            hip.ddpm_step(x, eps, st["noise"], st["coef"], st["step_idx"], st["z"], st["noise_stride"])
            sh = eng.shard
            if sh is None:
                # one guidance chain per clip (independent segments batched as clips of this pass each carry their own flows); the
                # `lr_images` term first, then the flows / masks term, as p_sample applies them (ddpm.py:4359-4373)
                Tn = self.num_frames
                terms = [st[k] for k in ("guid_lr", "guid") if k in st]
                src = st["z"]
                for ti, chain in enumerate(terms):
                    dst = x if ti == len(terms) - 1 else st["z2"]
                    for ci, (ff, fb, fo, bo) in enumerate(chain):
                        hip.guidance(src[ci * Tn:(ci + 1) * Tn], ff, fb, fo, bo, st["coef"], st["step_idx"], st["gscale"],
                                     dst[ci * Tn:(ci + 1) * Tn], st["work"][ci])
                    src = dst
            else:
                # frame-sharded clip: the guidance chain couples neighbouring frames -> all-gather the (64 KiB/frame)
                # latents, evaluate the tiny gradient on the whole clip on every rank, keep this rank's frames
                zf = st["z_full"]
                eng.collective(lambda: sh.all_gather(st["z"], out=zf))
                hip.guidance(zf, *st["guid"][0], st["coef"], st["step_idx"], st["gscale"], st["x_full"], st["work"][0])
                x.copy_(sh.local(st["x_full"]))
        else:
            hip.ddpm_step(x, eps, st["noise"], st["coef"], st["step_idx"], x, st["noise_stride"])
        hip.step_advance(st["step_idx"], -1)
        if "sc_tables" in st:
            hip.step_advance(st["win_idx"], -1)

    @torch.no_grad()
    def _sample_loop(self, cond, struct_cond, shape, guidance_scale, flows, masks, x_T, timesteps, time_replace,
                     return_intermediates, log_every_t, noise, tile, use_graph=True, hooks=None, lr_images=None):
        """hooks: the option branches of the reference loop (ddpm.py:4501-4599) that sit BETWEEN steps — start_T (skip the schedule
        indices above a timestep), mask + x0 (inpainting blend with a re-noised x0 after every step; optional mask_noise
        [steps, ...] instead of fresh draws), adain_fea (latent-space AdaIN after the last step), callback(i) /
        img_callback(img, i).  They run as stream-ordered work between the replays of the captured step graph."""
        hooks = {k: v for k, v in (hooks or {}).items() if v is not None}
        eng = self.engine()
        if self._stream is None:
            self._stream = torch.cuda.Stream(device=eng.device)
        S = timesteps if timesteps is not None else self.num_timesteps
        if S > self.posterior_mean_coef1.shape[0]:
            raise ValueError("timesteps exceeds the registered (respaced) schedule length")
        use_t_replace = not (time_replace is None or time_replace == 1000)
        if not log_every_t:
            log_every_t = self.log_every_t
        dev = eng.device
        T_total, c, h, w = shape
        cur = torch.cuda.current_stream(dev)
        self._stream.wait_stream(cur)
        with torch.cuda.stream(self._stream):
            x = (torch.randn(shape, device=dev) if x_T is None else x_T.to(dev, torch.float32)).contiguous().clone()
            if noise is None:
                noise = torch.randn((S,) + tuple(shape), device=dev)
            else:
                noise = noise.to(dev, torch.float32).contiguous()
                assert noise.shape == (S,) + tuple(shape), "noise must be [steps, T, C, h, w], indexed by schedule index i"
            lat = struct_cond.to(dev, torch.float32).contiguous()
            ctx = cond["c_crossattn"][0] if isinstance(cond, dict) else (cond[0] if isinstance(cond, list) else cond)
            ctx = ctx.to(dev, torch.float32)[:1].contiguous()
            idxs = list(reversed(range(S)))
            if "start_T" in hooks and tile is not None:
                # p_sample_loop_canvas has ANOTHER rule (ddpm.py:4639-4640): timesteps = min(timesteps, start_T), i.e. it walks the
                # schedule indices start_T-1 .. 0 whatever their original timesteps are
                idxs = list(reversed(range(min(S, int(hooks["start_T"])))))
            elif "start_T" in hooks:     # ddpm.py:4541-4550: `continue` while the step's (original) timestep is above start_T
                idxs = [i for i in idxs if (self.ori_timesteps[i] if use_t_replace else i) <= hooks["start_T"]]
            if "mask" in hooks:
                assert "x0" in hooks, "mask needs x0 (ddpm.py:4522-4524)"
                m_mask, m_x0 = hooks["mask"].to(dev, torch.float32), hooks["x0"].to(dev, torch.float32)
                assert m_x0.shape[2:3] == m_mask.shape[2:3]
            st = {
                "x": x, "coef": self._coef_table(S, use_t_replace).to(dev), "noise": noise, "noise_stride": x.numel(),
                "step_idx": torch.tensor([idxs[0] if idxs else 0], dtype=torch.int32, device=dev), "tvals": torch.zeros(1, device=dev),
                "ctx": self.model.diffusion_model.context_cache(eng, ctx), "guided": flows is not None or lr_images is not None,
                "gscale": float(guidance_scale), "tiles": None,
            }
            pieces_mode = False
            sh = eng.shard
            if sh is not None:
                assert tile is None and T_total == sh.F, "frame sharding: one clip per segment, local frames only"
                # collectives sit between the launches of a step: the step replays as graph PIECES around them (engine.GraphPieces)
                pieces_mode = use_graph and sh.world > 1 and os.environ.get("MGLD_SHARD_GRAPH", "1") != "0"
                use_graph = use_graph and sh.world == 1
            if flows is not None:
                T_clip = T_total if sh is None else sh.T
                Tn = self.num_frames
                # the reference guides ONE clip (b = 1: its p_sample cannot broadcast the t == 0 mask for b > 1, ddpm.py:4348); several
                # independent segments batched as clips of one pass (pipeline.run_segment with k * num_frames frames) are k such
                # problems: flows / masks carry a leading clip dimension, every clip runs its own chain
                nclips = T_clip // Tn
                assert T_clip % Tn == 0 and (sh is None or nclips == 1), "guidance expects whole clips of num_frames frames"
                assert flows[0].shape[0] == nclips, f"flows carry {flows[0].shape[0]} clips, the sample has {nclips}"
                st["guid"] = [self._flows_to_device(eng, tuple(f[i:i + 1] for f in flows), tuple(mk[i:i + 1] for mk in masks))
                              for i in range(nclips)]
                st["z"] = torch.empty_like(x)
                st["work"] = [torch.empty(hip.guidance_work_bytes(Tn, c, h, w), dtype=torch.uint8, device=dev) for _ in range(nclips)]
                if sh is not None:
                    st["x_full"] = torch.empty((T_clip, c, h, w), device=dev)
                    st["z_full"] = torch.empty((T_clip, c, h, w), device=dev)
            if lr_images is not None:
                # the `lr_images` term (ddpm.py:4359-4366 -> compute_temporal_condition_v2 :3469-3500): the reference resizes the LR frames
                # to the latent grid (bicubic), runs its flow network on them INSIDE every step and pulls the latents along those flows
                # with nothing occluded.  The flows depend on lr_images only: estimated once here (fp32 RAFT), then the same guidance
                # kernel with all-valid masks
                if sh is not None:
                    raise NotImplementedError("lr_images guidance with a frame-sharded clip")
                Tn = self.num_frames
                assert T_total % Tn == 0 and lr_images.shape[0] == T_total, "lr_images: one LR frame per latent frame"
                nclips = T_total // Tn
                res = hip.resize_bicubic(lr_images.to(dev, torch.float32), (h, w))
                f_f, f_b = self.compute_flow(res.view(nclips, Tn, res.shape[1], h, w), _inside_sampler=True)
                zero = torch.zeros(Tn - 1, h, w, device=dev)
                st["guid_lr"] = [(f_f[i].contiguous(), f_b[i].contiguous(), zero, zero) for i in range(nclips)]
                if "z" not in st:
                    st["z"] = torch.empty_like(x)
                    st["work"] = [torch.empty(hip.guidance_work_bytes(Tn, c, h, w), dtype=torch.uint8, device=dev) for _ in range(nclips)]
                if flows is not None:
                    st["z2"] = torch.empty_like(x)
            # persistent (non-arena) conditioning buffers
            if tile is None:
                la = torch.empty(T_total * h * w, 8, dtype=torch.float16, device=dev)
                hip.nchw_to_nhwc(lat, la, 8)
                st["lat_act"] = Act(la, T_total, h, w)
            else:
                ts, ov = tile
                tiles = self._tile_origins(h, w, ts, ov)
                st["tiles"], st["tile_size"] = tiles, ts
                tsh = eng.tile_shard
                if tsh is not None:
                    if tsh.n_tiles != len(tiles):
                        raise ValueError(f"tile shard built for {tsh.n_tiles} tiles, this canvas has {len(tiles)}")
                    pieces_mode = use_graph and tsh.world > 1 and os.environ.get("MGLD_SHARD_GRAPH", "1") != "0"
                    use_graph = use_graph and tsh.world == 1     # the tile exchange sits between the launches of a step
                own = tiles if tsh is None else tiles[tsh.k0:tsh.k1]
                n_own = max(1, len(own))
                lt = torch.zeros(n_own * T_total, c, ts, ts, device=dev)
                for k, (y0, x0) in enumerate(own):
                    hip.crop(lat, lt[k * T_total:(k + 1) * T_total], y0, x0)
                la = torch.empty(n_own * T_total * ts * ts, 8, dtype=torch.float16, device=dev)
                hip.nchw_to_nhwc(lt, la, 8)
                st["lat_tiles"] = Act(la, n_own * T_total, ts, ts)
                st["wgt"] = self._gaussian_weights(ts, ts, 1)[0, 0].to(dev, torch.float32).contiguous()
                st["acc"], st["cnt"], st["eps_canvas"] = torch.empty_like(x), torch.empty_like(x), torch.empty_like(x)
            lat_in = st["lat_act"] if tile is None else st["lat_tiles"]
            hoist_spade = self.precompute_structcond and S > 1 and os.environ.get("MGLD_SPADE_PRECOMPUTE", "1") != "0"
            if self.precompute_structcond:
                st["win_idx"] = torch.zeros(1, dtype=torch.int32, device=dev)
                st["hoist_window"] = Wn = min(self._hoist_window(eng, lat_in, S), int(os.environ.get("MGLD_HOIST_WINDOW", S)))
            intermediates = [x.clone()]
            graph = None
            eng.pieces = None
            for k, i in enumerate(idxs):
                if self.precompute_structcond and k % Wn == 0:  # a new hoisting window: schedule indices i .. i-Wn+1
                    self._precompute_structcond(eng, st, lat_in, i, max(0, i - Wn + 1))
                    if hoist_spade and k > 0:
                        self._precompute_spade(eng, st)
                if k == 1 and hoist_spade:
                    self._precompute_spade(eng, st)             # after the eager first step (it records the block scales)
                if pieces_mode and k > 0:
                    if eng.pieces is None:                      # record the step as graph pieces around its collectives
                        eng.arena.frozen = True
                        eng.pieces = GraphPieces()
                        eng.pieces.begin()
                        try:
                            self._step_body(eng, st)
                            eng.pieces.finish()
                        except BaseException:
                            eng.pieces.abort()
                            eng.pieces = None
                            raise
                        finally:
                            eng.arena.frozen = False
                    else:
                        eng.pieces.replay()
                elif k == 0 or not use_graph:
                    self._step_body(eng, st)
                else:
                    if graph is None:
                        eng.arena.frozen = True
                        graph = hip.Graph()
                        graph.begin()
                        try:
                            self._step_body(eng, st)
                        finally:
                            graph.end()
                            eng.arena.frozen = False
                    graph.launch()
                if hooks:
                    if "adain_fea" in hooks and i < 1:          # ddpm.py:4565-4567
                        from .flowops import adaptive_instance_normalization
                        x.copy_(adaptive_instance_normalization(x, hooks["adain_fea"].to(dev, torch.float32)))
                    if "mask" in hooks:                         # ddpm.py:4568-4570: img = q_sample(x0, ts) * mask + (1 - mask) * img
                        mn = hooks["mask_noise"][i].to(dev, torch.float32) if "mask_noise" in hooks else None
                        img_orig = self.q_sample(m_x0, torch.full((max(1, T_total // self.num_frames),), i, dtype=torch.long), noise=mn)
                        x.copy_(img_orig * m_mask + (1.0 - m_mask) * x)
                if return_intermediates and (i % log_every_t == 0 or i == S - 1):
                    intermediates.append(x.clone())
                if "callback" in hooks:
                    hooks["callback"](i)
                if "img_callback" in hooks:
                    hooks["img_callback"](x.clone(), i)
            self.last_launches_per_step = eng.launches
            self.last_graph_pieces = eng.pieces.n_graphs if eng.pieces is not None else (1 if graph is not None else 0)
            eng.pieces = None
            out = x
        cur.wait_stream(self._stream)
        if return_intermediates:
            return out, intermediates
        return out

    # ---- single-step API (ddpm.py:4157-4189, 4191-4322, 4325-4380, 4383-4442) ---------------------------------------------
    # Inside `sample` / `sample_canvas` a step exists only as a captured launch sequence; these are the same launches issued
    # eagerly, one network evaluation per call, for callers that drive their own loop.
    def _eps_canvas(self, x, c, struct_cond, t_in, tile_size, tile_overlap, tile_weights):
        """aggregation of the tile predictions (ddpm.py:4205-4300): every tile through the struct-cond encoder + UNet in ONE
        batched pass, Gaussian-weighted accumulation, normalisation"""
        eng = self.engine()
        dev = eng.device
        x = x.to(dev, torch.float32).contiguous()
        lat = struct_cond.to(dev, torch.float32).contiguous()
        Tn, ch, h, w = x.shape
        tiles = self._tile_origins(h, w, tile_size, tile_overlap)
        xt = torch.empty(len(tiles) * Tn, ch, tile_size, tile_size, device=dev)
        lt = torch.empty_like(xt)
        for k, (y0, x0) in enumerate(tiles):
            hip.crop(x, xt[k * Tn:(k + 1) * Tn], y0, x0)
            hip.crop(lat, lt[k * Tn:(k + 1) * Tn], y0, x0)
        tv = t_in.reshape(-1)[:1].to(dev)
        sc = self.structcond_stage_model(lt, tv)
        et = self.apply_model(xt, tv, c, sc)
        wgt = tile_weights[0, 0].to(dev, torch.float32).contiguous() if tile_weights.dim() == 4 else tile_weights.to(dev, torch.float32).contiguous()
        acc, cnt = torch.zeros_like(x), torch.zeros_like(x)
        for k, (y0, x0) in enumerate(tiles):
            hip.tile_accumulate(et[k * Tn:(k + 1) * Tn].contiguous(), wgt, acc, cnt, y0, x0)
        return hip.tile_normalize(acc, cnt, torch.empty_like(x))

    def _posterior_from_eps(self, x, t, eps, clip_denoised, return_x0):
        x = x.to(eps.device, torch.float32)
        x_recon = self.predict_start_from_noise(x, t=t, noise=eps)
        if clip_denoised:
            x_recon.clamp_(-1., 1.)
        mean, var, logvar = self.q_posterior(x_start=x_recon, x_t=x, t=t)
        return (mean, var, logvar, x_recon) if return_x0 else (mean, var, logvar)

    @torch.no_grad()
    def p_mean_variance(self, x, c, struct_cond, t, clip_denoised, return_codebook_ids=False, quantize_denoised=False, return_x0=False,
                        score_corrector=None, corrector_kwargs=None, t_replace=None):
        """ddpm.py:4157-4189.  struct_cond: the dict the struct-cond encoder returned for this step."""
        self._check_unsupported(score_corrector=score_corrector, return_codebook_ids=return_codebook_ids or None,
                                quantize_denoised=quantize_denoised or None)
        eps = self.apply_model(x, t if t_replace is None else t_replace, c, struct_cond)
        return self._posterior_from_eps(x, t, eps, clip_denoised, return_x0)

    @torch.no_grad()
    def p_mean_variance_canvas(self, x, c, struct_cond, t, clip_denoised, return_codebook_ids=False, quantize_denoised=False,
                               return_x0=False, score_corrector=None, corrector_kwargs=None, t_replace=None, tile_size=64,
                               tile_overlap=32, batch_size=4, tile_weights=None):
        """ddpm.py:4191-4322.  struct_cond: the LR latent (the struct-cond encoder runs per tile inside)."""
        assert tile_weights is not None
        self._check_unsupported(score_corrector=score_corrector, return_codebook_ids=return_codebook_ids or None,
                                quantize_denoised=quantize_denoised or None)
        eps = self._eps_canvas(x, c, struct_cond, t if t_replace is None else t_replace, tile_size, tile_overlap, tile_weights)
        return self._posterior_from_eps(x, t, eps, clip_denoised, return_x0)

    def _finish_p_sample(self, x, outputs, t, guidance_scale, flows, masks, return_x0, temperature, noise, lr_images=None):
        mean, logvar = outputs[0], outputs[2]
        b = x.shape[0] // self.num_frames
        if noise is None:
            noise = torch.randn(x.shape, device=mean.device)                   # noise_like (util.py:265-268)
        noise = noise.to(mean.device, torch.float32) * temperature
        nonzero = (1 - (t == 0).float()).reshape(b, *((1,) * (x.dim() - 1))).to(mean.device)
        latents = mean + nonzero * (0.5 * logvar).exp() * noise
        if lr_images is not None:                                              # (:4359-4366) compute_temporal_condition_v2: flows of the
            eng = self.engine()                                                #  LR frames at the latent grid, nothing occluded
            Tn, ch, h, w = latents.shape
            assert Tn == self.num_frames and lr_images.shape[0] == Tn, "lr_images guidance operates on one clip"
            res = hip.resize_bicubic(lr_images.to(eng.device, torch.float32), (h, w))
            f_f, f_b = self.compute_flow(res[None], _inside_sampler=True)
            zero = torch.zeros(Tn - 1, h, w, device=eng.device)
            coef = torch.zeros(1, 8, device=eng.device)
            coef[0, 4] = logvar.reshape(-1)[0]
            out = torch.empty_like(latents)
            work = torch.empty(hip.guidance_work_bytes(Tn, ch, h, w), dtype=torch.uint8, device=eng.device)
            hip.guidance(latents.contiguous(), f_f[0].contiguous(), f_b[0].contiguous(), zero, zero, coef,
                         torch.zeros(1, dtype=torch.int32, device=eng.device), float(guidance_scale), out, work)
            latents = out
        if flows is not None:                                                  # (:4367-4373) latents -= s * logvar * dL/dlatents
            eng = self.engine()
            ff, fb, fo, bo = self._flows_to_device(eng, flows, masks)
            Tn, ch, h, w = latents.shape
            coef = torch.zeros(1, 8, device=eng.device)
            coef[0, 4] = logvar.reshape(-1)[0]
            out = torch.empty_like(latents)
            work = torch.empty(hip.guidance_work_bytes(Tn, ch, h, w), dtype=torch.uint8, device=eng.device)
            hip.guidance(latents.contiguous(), ff, fb, fo, bo, coef, torch.zeros(1, dtype=torch.int32, device=eng.device),
                         float(guidance_scale), out, work)
            latents = out
        return (latents, outputs[3]) if return_x0 else latents

    @torch.no_grad()
    def p_sample(self, x, c, struct_cond, t, guidance_scale=-1.0, lr_images=None, flows=None, masks=None, clip_denoised=False,
                 repeat_noise=False, return_codebook_ids=False, quantize_denoised=False, return_x0=False, temperature=1.,
                 noise_dropout=0., score_corrector=None, corrector_kwargs=None, t_replace=None, noise=None):
        """ddpm.py:4325-4380: one reverse step.  Extra kwarg `noise`: the step's Gaussian draw (the reference draws it inside)."""
        self._check_unsupported(repeat_noise=repeat_noise or None, noise_dropout=noise_dropout or None)
        outputs = self.p_mean_variance(x=x, c=c, struct_cond=struct_cond, t=t, clip_denoised=clip_denoised,
                                       return_codebook_ids=return_codebook_ids, quantize_denoised=quantize_denoised, return_x0=return_x0,
                                       score_corrector=score_corrector, corrector_kwargs=corrector_kwargs, t_replace=t_replace)
        return self._finish_p_sample(x, outputs, t, guidance_scale, flows, masks, return_x0, temperature, noise, lr_images=lr_images)

    @torch.no_grad()
    def p_sample_canvas(self, x, c, struct_cond, t, guidance_scale=-1.0, lr_images=None, flows=None, masks=None, clip_denoised=False,
                        repeat_noise=False, return_codebook_ids=False, quantize_denoised=False, return_x0=False, temperature=1.,
                        noise_dropout=0., score_corrector=None, corrector_kwargs=None, t_replace=None, tile_size=64, tile_overlap=32,
                        batch_size=4, tile_weights=None, noise=None):
        """ddpm.py:4383-4442"""
        self._check_unsupported(repeat_noise=repeat_noise or None, noise_dropout=noise_dropout or None)
        outputs = self.p_mean_variance_canvas(x=x, c=c, struct_cond=struct_cond, t=t, clip_denoised=clip_denoised,
                                              return_codebook_ids=return_codebook_ids, quantize_denoised=quantize_denoised,
                                              return_x0=return_x0, score_corrector=score_corrector, corrector_kwargs=corrector_kwargs,
                                              t_replace=t_replace, tile_size=tile_size, tile_overlap=tile_overlap, batch_size=batch_size,
                                              tile_weights=tile_weights)
        return self._finish_p_sample(x, outputs, t, guidance_scale, flows, masks, return_x0, temperature, noise, lr_images=lr_images)

    def _check_unsupported(self, **kw):
        for k, v in kw.items():
            if v is not None:
                raise NotImplementedError(f"option '{k}' is not used by the MGLD-VSR inference scripts and is not implemented")

    @torch.no_grad()
    def sample(self, cond, struct_cond, guidance_scale=-1.0, lr_images=None, flows=None, masks=None, batch_size=16,
               return_intermediates=False, x_T=None, verbose=True, timesteps=None, quantize_denoised=False, mask=None,
               x0=None, shape=None, time_replace=None, adain_fea=None, interfea_path=None, start_T=None, noise=None,
               use_graph=True, **kwargs):
        """ddpm.py:4696-4719 -> p_sample_loop.  Extra kwargs: `noise` [steps,T,C,h,w] (injected noise indexed by
        the schedule index; the reference draws randn per step) and `use_graph`."""
        self._check_unsupported(interfea_path=interfea_path, quantize_denoised=quantize_denoised or None)
        if shape is None:
            shape = tuple(struct_cond.shape) if x_T is None else tuple(x_T.shape)
        if cond is not None and not isinstance(cond, (dict, list)):
            cond = cond[:batch_size]
        hooks = dict(mask=mask, x0=x0, adain_fea=adain_fea, start_T=start_T, mask_noise=kwargs.get("mask_noise"),
                     callback=kwargs.get("callback"), img_callback=kwargs.get("img_callback"))
        return self._sample_loop(cond, struct_cond, shape, guidance_scale, flows, masks, x_T, timesteps, time_replace,
                                 return_intermediates, None, noise, None, use_graph, hooks=hooks, lr_images=lr_images)

    @torch.no_grad()
    def p_sample_loop(self, cond, struct_cond, shape, guidance_scale=-1.0, lr_images=None, flows=None, masks=None,
                      return_intermediates=False, x_T=None, verbose=True, callback=None, timesteps=None, quantize_denoised=False,
                      mask=None, x0=None, img_callback=None, start_T=None, log_every_t=None, time_replace=None, adain_fea=None,
                      interfea_path=None):
        """ddpm.py:4501-4616, the loop `sample` enters: same arguments.  A step runs as the captured step graph; the options that act
        BETWEEN steps (start_T, mask / x0 inpainting, adain_fea, callback / img_callback) run as stream-ordered work between the
        replays.  lr_images: the other guidance term (compute_temporal_condition_v2; round 5).  interfea_path (PCA pictures of the
        struct-cond features, a debugging aid: ddpm.py:4574-4597) is refused rather than ignored."""
        self._check_unsupported(interfea_path=interfea_path, quantize_denoised=quantize_denoised or None)
        hooks = dict(mask=mask, x0=x0, adain_fea=adain_fea, start_T=start_T, callback=callback, img_callback=img_callback)
        return self._sample_loop(cond, struct_cond, tuple(shape), guidance_scale, flows, masks, x_T, timesteps, time_replace,
                                 return_intermediates, log_every_t, None, None, True, hooks=hooks, lr_images=lr_images)

    @torch.no_grad()
    def p_sample_loop_canvas(self, cond, struct_cond, shape, guidance_scale=-1.0, lr_images=None, flows=None, masks=None,
                             return_intermediates=False, x_T=None, verbose=True, callback=None, timesteps=None,
                             quantize_denoised=False, mask=None, x0=None, img_callback=None, start_T=None, log_every_t=None,
                             time_replace=None, adain_fea=None, interfea_path=None, tile_size=64, tile_overlap=32, batch_size=4):
        """ddpm.py:4619-4693, the loop `sample_canvas` enters (`batch_size` = tiles per UNet pass in the reference; here every
        tile of a step goes through one pass)."""
        assert tile_size is not None
        self._check_unsupported(interfea_path=interfea_path, quantize_denoised=quantize_denoised or None)
        hooks = dict(mask=mask, x0=x0, adain_fea=adain_fea, start_T=start_T, callback=callback, img_callback=img_callback)
        return self._sample_loop(cond, struct_cond, tuple(shape), guidance_scale, flows, masks, x_T, timesteps, time_replace,
                                 return_intermediates, log_every_t, None, (tile_size, tile_overlap), True, hooks=hooks, lr_images=lr_images)

    @torch.no_grad()
    def sample_canvas(self, cond, struct_cond, guidance_scale=-1.0, lr_images=None, flows=None, masks=None, batch_size=16,
                      return_intermediates=False, x_T=None, verbose=True, timesteps=None, quantize_denoised=False,
                      mask=None, x0=None, shape=None, time_replace=None, adain_fea=None, interfea_path=None, tile_size=64,
                      tile_overlap=32, batch_size_sample=4, log_every_t=None, noise=None, use_graph=True, **kwargs):
        """ddpm.py:4722-4746 -> p_sample_loop_canvas: aggregation sampling over overlapping latent tiles.  All tiles
        of a step are batched into one struct-cond + UNet pass (each tile is an independent clip)."""
        self._check_unsupported(interfea_path=interfea_path, quantize_denoised=quantize_denoised or None)
        if shape is None:
            shape = tuple(struct_cond.shape) if x_T is None else tuple(x_T.shape)
        hooks = dict(mask=mask, x0=x0, adain_fea=adain_fea, mask_noise=kwargs.get("mask_noise"), callback=kwargs.get("callback"),
                     img_callback=kwargs.get("img_callback"))
        return self._sample_loop(cond, struct_cond, shape, guidance_scale, flows, masks, x_T, timesteps, time_replace,
                                 return_intermediates, log_every_t, noise, (tile_size, tile_overlap), use_graph, hooks=hooks, lr_images=lr_images)
