"""Deterministic synthetic weights and inputs: no checkpoint or dataset is reachable here, so
every BASELINE config runs on random-init weights of the reference architecture.  The recipe depends only on
(parameter name, shape, salt), so the build container (golden fixtures from the reference import), the GPU box
(parity tests, bench) and the oracle all regenerate bit-identical fp32 tensors without shipping them.

Zero-initialised reference modules (zero_module convs / proj_out, openaimodel.py:301-304,433-435,519,2256;
attention.py:524) and the uninitialised `temporal_alpha` (util.py:299, attention.py:133) get the SAME non-zero
recipe — otherwise a random-init network is identically zero and parity is vacuous.
"""
import zlib

import numpy as np
import torch


def _rng(name, salt):
    return np.random.Generator(np.random.PCG64(zlib.crc32(f"{salt}/{name}".encode()) & 0xFFFFFFFF))


def synth_param(name, shape, salt="w"):
    """fp32 tensor for parameter `name` of the given shape."""
    shape = tuple(int(s) for s in shape)
    if name.endswith("temporal_alpha"):
        return torch.full(shape, 0.5, dtype=torch.float32)
    g = _rng(name, salt)
    n = int(np.prod(shape)) if len(shape) else 1
    v = g.standard_normal(n, dtype=np.float32).reshape(shape)
    leaf = name.rsplit(".", 1)[-1]
    if leaf == "running_var":       # BatchNorm statistics (RAFT context encoder): strictly positive
        v = 1.0 + 0.3 * np.abs(v)
    elif leaf == "running_mean":
        v *= 0.1
    elif leaf == "bias":
        v *= 0.02
    elif len(shape) >= 2:
        fan_in = int(np.prod(shape[1:]))
        v *= 1.0 / np.sqrt(fan_in)
    elif leaf == "weight":  # 1-D: norm scale
        v = 1.0 + 0.1 * v
    else:
        v *= 0.02
    return torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32))


def _parallel(items, salt):
    """generate many tensors concurrently (numpy releases the GIL while drawing); per-name seeding keeps it deterministic"""
    import os
    from concurrent.futures import ThreadPoolExecutor
    items = list(items)
    with ThreadPoolExecutor(max_workers=max(1, min(32, os.cpu_count() or 1))) as ex:
        vals = list(ex.map(lambda ns: synth_param(ns[0], ns[1], salt), items))
    return {n: v for (n, _), v in zip(items, vals)}


def synth_state_dict(names_shapes, salt="w"):
    """names_shapes: iterable of (name, shape) -> {name: tensor}"""
    return _parallel(names_shapes, salt)


def fill_module_(module, salt="w"):
    """In-place: overwrite every floating-point entry of module.state_dict() with the recipe."""
    sd = module.state_dict()
    gen = _parallel([(k, tuple(v.shape)) for k, v in sd.items() if v.is_floating_point()], salt)
    new = {k: (gen[k].to(v.dtype) if k in gen else v) for k, v in sd.items()}
    module.load_state_dict(new, strict=True)
    return module


def synth_tensor(tag, shape, scale=1.0):
    """seeded N(0, scale^2) input tensor."""
    g = _rng(tag, "input")
    return torch.from_numpy(g.standard_normal(int(np.prod(shape)), dtype=np.float32).reshape(shape) * np.float32(scale))


def smooth_flow(tag, n, h, w, amp=1.5):
    """smooth synthetic optical flow [n,2,h,w] in pixels: bilinear upsampling of an 8x8 N(0, amp^2) grid."""
    g = synth_tensor(tag, (n, 2, 8, 8), amp)
    return torch.nn.functional.interpolate(g, size=(h, w), mode="bilinear", align_corners=True).contiguous()
