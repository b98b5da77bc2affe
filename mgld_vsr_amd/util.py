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
