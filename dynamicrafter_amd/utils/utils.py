"""The operator registry of the reference (utils/utils.py:27-42): `target:` dotted path -> class -> instance.

`lvdm.*` targets of the released YAMLs resolve to this package's HIP-backed classes (same dotted layout under
`dynamicrafter_amd.`), so the reference's configs work unmodified. Unlike the reference module this one does
not import cv2.
"""
import importlib

ALIASES = ("lvdm.", "utils.")


def get_obj_from_str(string, reload=False):
    module, cls = string.rsplit(".", 1)
    if module.startswith(ALIASES):
        module = "dynamicrafter_amd." + module
    mod = importlib.import_module(module)
    if reload:
        mod = importlib.reload(mod)
    return getattr(mod, cls)


def instantiate_from_config(config):
    if "target" not in config:
        if config in ("__is_first_stage__", "__is_unconditional__"):
            return None
        raise KeyError("Expected key `target` to instantiate.")
    params = config.get("params", dict())
    return get_obj_from_str(config["target"])(**(params or {}))


def count_params(model, verbose=False):
    total = sum(p.numel() for p in model.parameters())
    if verbose:
        print(f"{model.__class__.__name__} has {total * 1.e-6:.2f} M params.")
    return total
