"""Config / plugin seam, same contract as the reference's ldm/util.py:78-93:
`instantiate_from_config({"target": "dotted.path.Class", "params": {...}}, **kwargs)`."""
import importlib


def get_obj_from_str(string, reload=False):
    module, cls = string.rsplit(".", 1)
    if reload:
        importlib.reload(importlib.import_module(module))
    return getattr(importlib.import_module(module, package=None), cls)


def instantiate_from_config(config, **kwargs):
    if "target" not in config:
        if config in ("__is_first_stage__", "__is_unconditional__"):
            return None
        raise KeyError("Expected key `target` to instantiate.")
    return get_obj_from_str(config["target"])(**dict(config.get("params", dict()) or {}), **kwargs)
