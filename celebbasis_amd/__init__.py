"""celebbasis_amd — MI355X-native (gfx950) CelebBasis personalization training step.

    celebbasis_amd.lib            ctypes binding of libcelebbasis_hip.so (include/celebbasis_hip.h)
    celebbasis_amd.ldm.*          host-side mirror of the reference's plugin classes (same dotted names under `ldm.`)
    celebbasis_amd.install_ldm_alias()   makes `ldm.*` / `main.*` target strings of the reference YAMLs resolve here
"""
import sys


def install_ldm_alias():
    """Let the reference's YAML `target:` strings (`ldm.models.diffusion.ddpm.LatentDiffusion`, `main.DataModuleFromConfig`, …)
    resolve to this package: every `celebbasis_amd.ldm.*` module is registered under the matching `ldm.*` name (same module
    objects). Refuses to shadow a real `ldm` package that is already imported."""
    import importlib
    import pkgutil
    from . import ldm as _ldm
    from . import main as _main
    cur = sys.modules.get("ldm")
    if cur is not None and cur is not _ldm:
        raise RuntimeError("a different `ldm` package is already imported; cannot alias celebbasis_amd.ldm")
    sys.modules["ldm"] = _ldm
    for info in pkgutil.walk_packages(_ldm.__path__, prefix="celebbasis_amd.ldm."):
        mod = importlib.import_module(info.name)
        sys.modules["ldm." + info.name[len("celebbasis_amd.ldm."):]] = mod
    sys.modules.setdefault("main", _main)
    return _ldm
