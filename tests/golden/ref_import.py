"""Import shim for running the reference's own Python (read-only at /root/reference) in the BUILD container.

Used ONLY by tests/golden/make_golden.py to produce golden vectors; never on the GPU box, never by the product.
Follows the recipe of SURVEY.md §8(c): stub the missing third-party modules, bypass basicsr/__init__.py, restore
einops-0.3 `t=None` semantics, and give `xformers.ops.memory_efficient_attention` its exact-softmax meaning.
"""
import importlib
import os
import sys
import types

import torch
import torch.nn as nn

REF = os.environ.get("MGLD_REFERENCE", "/root/reference")


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def install():
    if getattr(install, "_done", False):
        return
    if not os.path.isdir(REF):
        raise RuntimeError(f"reference tree not found at {REF}")
    sys.path.insert(0, REF)

    # ---- pytorch_lightning ----
    class LightningModule(nn.Module):
        @property
        def device(self):
            try:
                return next(self.parameters()).device
            except StopIteration:
                return torch.device("cpu")

        def log(self, *a, **k):
            pass

        def log_dict(self, *a, **k):
            pass

    def seed_everything(seed):
        torch.manual_seed(seed)

    pl = _mod("pytorch_lightning", LightningModule=LightningModule, seed_everything=seed_everything)
    _mod("pytorch_lightning.utilities")
    _mod("pytorch_lightning.utilities.distributed", rank_zero_only=lambda f: f)
    pl.utilities = sys.modules["pytorch_lightning.utilities"]

    # ---- torchvision ----
    tv = _mod("torchvision", __version__="0.0")
    tv.utils = _mod("torchvision.utils", make_grid=lambda *a, **k: None)
    tv.transforms = _mod("torchvision.transforms", ToTensor=object, ToPILImage=object)
    tv.ops = _mod("torchvision.ops")
    tv.models = _mod("torchvision.models")

    _mod("cv2")
    _mod("skimage", img_as_ubyte=None, img_as_float32=None)
    _mod("taming")
    _mod("taming.modules")
    _mod("taming.modules.vqvae")
    _mod("taming.modules.vqvae.quantize", VectorQuantizer2=type("VectorQuantizer2", (nn.Module,), {}))
    _mod("omegaconf")
    _mod("omegaconf.listconfig", ListConfig=type("ListConfig", (list,), {}))
    _mod("mmcv")
    _mod("mmcv.ops", Correlation=type("Correlation", (nn.Module,), {}))

    # ---- xformers: exact softmax attention ----
    def memory_efficient_attention(q, k, v, attn_bias=None, op=None, scale=None):
        s = scale if scale is not None else q.shape[-1] ** -0.5
        return torch.softmax(q @ k.transpose(-2, -1) * s, dim=-1) @ v

    xf = _mod("xformers")
    xf.ops = _mod("xformers.ops", memory_efficient_attention=memory_efficient_attention)

    # ---- einops 0.3 semantics: None-valued axis kwargs are ignored ----
    import einops
    _orig = einops.rearrange

    def rearrange(tensor, pattern, **axes):
        return _orig(tensor, pattern, **{k: v for k, v in axes.items() if v is not None})

    einops.rearrange = rearrange

    # ---- basicsr without its __init__ ----
    def pkg(name, path):
        m = types.ModuleType(name)
        m.__path__ = [path]
        sys.modules[name] = m
        return m

    # `ldm` and `scripts` are namespace packages in the reference (no __init__.py) while the repo root holds regular packages
    # of the same names (the product's drop-in surface).  Pin both names to the reference tree explicitly, so that whatever
    # sys.path order the caller has, `ldm.*` / `scripts.*` imported through this shim are the reference's files.
    for top in ("ldm", "scripts"):
        stale = [k for k in sys.modules if k == top or k.startswith(top + ".")]
        for k in stale:
            f = getattr(sys.modules[k], "__file__", None) or ""
            if not f.startswith(REF):
                del sys.modules[k]
        pkg(top, os.path.join(REF, top))
    b = pkg("basicsr", os.path.join(REF, "basicsr"))
    b.archs = pkg("basicsr.archs", os.path.join(REF, "basicsr", "archs"))
    b.ops = pkg("basicsr.ops", os.path.join(REF, "basicsr", "ops"))
    b.data = pkg("basicsr.data", os.path.join(REF, "basicsr", "data"))
    bu = pkg("basicsr.utils", os.path.join(REF, "basicsr", "utils"))
    b.utils = bu
    for n in ("DiffJPEG", "USMSharp"):
        setattr(bu, n, type(n, (nn.Module,), {}))
    bu.get_root_logger = lambda *a, **k: __import__("logging").getLogger("basicsr")
    bu.flow_to_image = lambda *a, **k: None
    _mod("basicsr.utils.img_process_util", filter2D=None)

    class _Reg:
        def register(self, *a, **k):
            return (lambda f: f) if not a or not callable(a[0]) else a[0]

    _mod("basicsr.utils.registry", ARCH_REGISTRY=_Reg(), DATASET_REGISTRY=_Reg(), LOSS_REGISTRY=_Reg(),
         METRIC_REGISTRY=_Reg(), MODEL_REGISTRY=_Reg())
    _mod("basicsr.data.transforms", paired_random_crop=None, triplet_random_crop=None)
    _mod("basicsr.data.degradations", random_add_gaussian_noise_pt=None, random_add_poisson_noise_pt=None,
         random_add_speckle_noise_pt=None, random_add_saltpepper_noise_pt=None, bivariate_Gaussian=None)
    install._done = True


def ref(modname):
    """import a module of the reference tree; refuses anything that does not resolve to a file under REF"""
    install()
    m = importlib.import_module(modname)
    f = os.path.realpath(getattr(m, "__file__", None) or "")
    if not f.startswith(os.path.realpath(REF) + os.sep):
        raise RuntimeError(f"ref_import.ref('{modname}') resolved to {f or '<no file>'}, not to the reference tree {REF}")
    return m
