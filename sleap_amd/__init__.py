

# `sleap.load_model / load_file / Video / Labels` spelled the same here (resolved lazily: importing the package must not
# import torch or load the HIP library).
_LAZY = {"load_model": ("sleap_amd.nn.inference", "load_model"), "load_file": ("sleap_amd.io.labels", "Labels"),
         "Video": ("sleap_amd.io.video", "Video"), "Labels": ("sleap_amd.io.labels", "Labels")}


def __getattr__(name):
    if name in _LAZY:
        import importlib

        mod, attr = _LAZY[name]
        obj = getattr(importlib.import_module(mod), attr)
        return obj.load_file if name == "load_file" else obj
    raise AttributeError(f"module 'sleap_amd' has no attribute {name!r}")
