"""Config-string plugin glue, mirror of ldm/util.py:78-102 (instantiate_from_config / get_obj_from_str)."""
import importlib


def get_obj_from_str(string, reload=False):
    module, cls = string.rsplit(".", 1)
    if reload:
        importlib.reload(importlib.import_module(module))
    return getattr(importlib.import_module(module, package=None), cls)


def _get(cfg, key, default=None):
    if isinstance(cfg, dict):
        return cfg.get(key, default)
    return getattr(cfg, key, default) if hasattr(cfg, key) else (cfg.get(key, default) if hasattr(cfg, "get") else default)


def instantiate_from_config(config):
    if "target" not in config:
        if config == "__is_first_stage__" or config == "__is_unconditional__":
            return None
        raise KeyError("Expected key `target` to instantiate.")
    params = _get(config, "params", None) or {}
    return get_obj_from_str(_get(config, "target"))(**dict(params))


def exists(x):
    return x is not None


def default(val, d):
    return val if val is not None else (d() if callable(d) else d)


def load_trusted_checkpoint(path, map_location="cpu"):
    """torch.load for the reference's checkpoint files (scripts/vsr_val_ddpm_text_T_vqganfin_oldcanvas_tile.py:91-108 calls
    `torch.load(ckpt, map_location="cpu")` with the pre-2.6 default, i.e. the full unpickler).  stablevsr_025.ckpt / vqgan_cfw_00011.ckpt
    are Lightning checkpoints: next to `state_dict` they pickle callback / hyper-parameter objects, which torch >= 2.6's default
    `weights_only=True` refuses ("Unsupported global").  Order: the safe loader first (plain state dicts need nothing else; the
    handful of harmless globals Lightning checkpoints carry are allow-listed).  When it still refuses, the file is NOT unpickled
    unless the caller opted in — env MGLD_TRUST_CKPT=1 (or trust=True): the full unpickler executes code from the file.  With the
    opt-in: the full unpickler on this local, user-supplied file, with a loud warning naming it; classes of modules that are not
    importable here (pytorch_lightning is not a dependency of this path) become inert placeholders, since only the tensors are read."""
    import collections
    import os
    import pickle
    import warnings

    import torch
    safe = [collections.OrderedDict, collections.defaultdict, dict, list, tuple, set, int, float, str, bool, bytes, slice, complex]
    try:
        import argparse
        safe.append(argparse.Namespace)              # Lightning's hyper_parameters
    except ImportError:
        pass
    try:
        with torch.serialization.safe_globals(safe):
            return torch.load(path, map_location=map_location, weights_only=True)
    except pickle.UnpicklingError as e:
        refused = str(e).splitlines()[0] if str(e) else "unsupported global"
    if os.environ.get("MGLD_TRUST_CKPT", "0") != "1" and not getattr(load_trusted_checkpoint, "trust", False):
        raise RuntimeError(
            f"{path}: the safe loader (torch.load(weights_only=True)) refuses this file ({refused}).  It is a pickle that can run code "
            "when loaded.  If you trust where it came from (the reference's own stablevsr_025.ckpt / vqgan_cfw_00011.ckpt are Lightning "
            "pickles of this kind), set MGLD_TRUST_CKPT=1 to load it with the full unpickler, or convert it once to a plain state dict.")
    warnings.warn(f"[mgld] loading {path} with the FULL unpickler (MGLD_TRUST_CKPT=1): code inside the file can run. "
                  f"The safe loader refused it: {refused}", stacklevel=2)

    class _Placeholder:
        def __init__(self, *a, **k):
            pass

        def __setstate__(self, state):
            self.__dict__["_state"] = state

        def __call__(self, *a, **k):
            return _Placeholder()

    class _Unpickler(pickle.Unpickler):
        def find_class(self, module, name):
            try:
                return super().find_class(module, name)
            except (ImportError, AttributeError):
                return type(name, (_Placeholder,), {"__module__": module})

    class _PickleModule:
        __name__ = "mgld_vsr_amd.util.tolerant_pickle"
        Unpickler = _Unpickler
        load = staticmethod(lambda f, **kw: _Unpickler(f, **kw).load())
        loads = staticmethod(pickle.loads)
        dump, dumps, Pickler = staticmethod(pickle.dump), staticmethod(pickle.dumps), pickle.Pickler
        PickleError, UnpicklingError, PicklingError = pickle.PickleError, pickle.UnpicklingError, pickle.PicklingError

    return torch.load(path, map_location=map_location, weights_only=False, pickle_module=_PickleModule)
