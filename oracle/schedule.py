"""Oracle (test infrastructure): DDPM schedule, respacing, posterior coefficients, one reverse step.

Follows ldm/modules/diffusionmodules/util.py:21-43 (make_beta_schedule 'linear'), ldm/models/diffusion/ddpm.py:101-154
(space_timesteps), :237-292 (register_schedule), :340-353 (predict_start_from_noise / q_posterior), :403-406
(q_sample_respace), scripts/vsr_val_ddpm_text_T_vqganfin_oldcanvas_tile.py:308-329 (respacing loop).
"""
import numpy as np
import torch


def make_beta_schedule_linear(n_timestep=1000, linear_start=0.00085, linear_end=0.012):
    betas = torch.linspace(linear_start ** 0.5, linear_end ** 0.5, n_timestep, dtype=torch.float64) ** 2
    return betas.numpy()


def space_timesteps(num_timesteps, section_counts):
    """ddpm.py:101-154 for a list of section counts."""
    if isinstance(section_counts, str):
        section_counts = [int(x) for x in section_counts.split(",")]
    size_per = num_timesteps // len(section_counts)
    extra = num_timesteps % len(section_counts)
    start_idx = 0
    all_steps = []
    for i, section_count in enumerate(section_counts):
        size = size_per + (1 if i < extra else 0)
        if size < section_count:
            raise ValueError(f"cannot divide section of {size} steps into {section_count}")
        frac_stride = 1 if section_count <= 1 else (size - 1) / (section_count - 1)
        cur_idx = 0.0
        taken = []
        for _ in range(section_count):
            taken.append(start_idx + round(cur_idx))
            cur_idx += frac_stride
        all_steps += taken
        start_idx += size
    return set(all_steps)


def schedule_buffers(betas):
    """register_schedule (ddpm.py:237-292) with v_posterior=0: float64 math, float32 buffers."""
    betas = np.asarray(betas)   # dtype preserved: float64 for the full schedule, float32 for respaced betas
    alphas = 1.0 - betas
    ac = np.cumprod(alphas, axis=0)
    ac_prev = np.append(1.0, ac[:-1])
    post_var = betas * (1.0 - ac_prev) / (1.0 - ac)
    f32 = lambda a: torch.tensor(a, dtype=torch.float32)
    return {
        "betas": f32(betas), "alphas_cumprod": f32(ac), "alphas_cumprod_prev": f32(ac_prev),
        "sqrt_alphas_cumprod": f32(np.sqrt(ac)), "sqrt_one_minus_alphas_cumprod": f32(np.sqrt(1.0 - ac)),
        "log_one_minus_alphas_cumprod": f32(np.log(1.0 - ac)),
        "sqrt_recip_alphas_cumprod": f32(np.sqrt(1.0 / ac)), "sqrt_recipm1_alphas_cumprod": f32(np.sqrt(1.0 / ac - 1)),
        "posterior_variance": f32(post_var),
        "posterior_log_variance_clipped": f32(np.log(np.maximum(post_var, 1e-20))),
        "posterior_mean_coef1": f32(betas * np.sqrt(ac_prev) / (1.0 - ac)),
        "posterior_mean_coef2": f32((1.0 - ac_prev) * np.sqrt(alphas) / (1.0 - ac)),
    }


def respaced_schedule(ddpm_steps, n_timestep=1000, linear_start=0.00085, linear_end=0.012):
    """oldcanvas_tile.py:308-329: full schedule -> use_timesteps -> new betas -> buffers; returns
    (full buffers, respaced buffers, ori_timesteps)."""
    full = schedule_buffers(make_beta_schedule_linear(n_timestep, linear_start, linear_end))
    use = space_timesteps(n_timestep, [ddpm_steps])
    last = 1.0
    new_betas = []
    for i, ac in enumerate(full["alphas_cumprod"]):   # float32 tensor iteration, as the script does
        if i in use:
            new_betas.append(1 - ac / last)
            last = ac
    new_betas = [b.data.cpu().numpy() for b in new_betas]     # float32 0-d arrays, exactly as the script
    resp = schedule_buffers(np.array(new_betas))
    ori = sorted(list(use))
    return full, resp, ori


def q_sample_respace(x_start, t, sqrt_ac, sqrt_1mac, noise):
    shp = (t.shape[0],) + (1,) * (x_start.dim() - 1)
    return sqrt_ac.gather(-1, t).reshape(shp) * x_start + sqrt_1mac.gather(-1, t).reshape(shp) * noise


def p_step(buf, i, x, eps, noise):
    """One reverse step without guidance (ddpm.py:340-353, 4344-4357) for schedule index i. Returns (z, logvar)."""
    x0 = buf["sqrt_recip_alphas_cumprod"][i] * x - buf["sqrt_recipm1_alphas_cumprod"][i] * eps
    mean = buf["posterior_mean_coef1"][i] * x0 + buf["posterior_mean_coef2"][i] * x
    logvar = buf["posterior_log_variance_clipped"][i]
    nonzero = 0.0 if i == 0 else 1.0
    return mean + nonzero * (0.5 * logvar).exp() * noise, logvar
