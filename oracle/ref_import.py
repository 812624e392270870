"""ORACLE tooling (build container only): import the reference's arch modules from
/root/reference WITHOUT executing basicsr/__init__.py (which needs cv2, torchvision, timm,
torchinfo -- none installed).  Used only by oracle/make_golden.py to generate fixtures and by
tests that are skipped when /root/reference is absent.  Nothing here travels to the GPU box
in a useful form: without /root/reference ``available()`` is False.
"""
from __future__ import annotations

import importlib.util
import os
import sys
import types

REF = "/root/reference"


def available() -> bool:
    return os.path.isdir(os.path.join(REF, "basicsr", "archs"))


def _load(name: str, rel: str):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, rel))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def load_reference_archs():
    """Returns a namespace with .nafnet, .restormer, .dc, .arch_util, .registry modules of the
    reference, loaded under private names so they never shadow this repo's own ``basicsr``."""
    if not available():
        raise RuntimeError("reference tree not present")
    saved = {k: v for k, v in sys.modules.items() if k == "basicsr" or k.startswith("basicsr.")}
    for k in saved:
        del sys.modules[k]
    try:
        for name, path in [("basicsr", "basicsr"), ("basicsr.utils", "basicsr/utils"), ("basicsr.archs", "basicsr/archs")]:
            m = types.ModuleType(name)
            m.__path__ = [os.path.join(REF, path)]
            sys.modules[name] = m
        ns = types.SimpleNamespace()
        ns.registry = _load("basicsr.utils.registry", "basicsr/utils/registry.py")
        ns.arch_util = _load("basicsr.archs.arch_util", "basicsr/archs/arch_util.py")
        ns.nafnet = _load("basicsr.archs.nafnet_arch", "basicsr/archs/nafnet_arch.py")
        try:
            ns.restormer = _load("basicsr.archs.restormer_arch", "basicsr/archs/restormer_arch.py")
        except Exception as e:  # einops missing etc.
            ns.restormer = None
            ns.restormer_error = e
        try:
            ns.promptir = _load("basicsr.archs.promptir_arch", "basicsr/archs/promptir_arch.py")
        except Exception as e:
            ns.promptir = None
            ns.promptir_error = e
        # fvcore is not installed; c2_msra_fill only initialises weights (degrad_classify_arch.py:213)
        if "fvcore" not in sys.modules:
            import torch.nn as nn

            fv = types.ModuleType("fvcore")
            fvnn = types.ModuleType("fvcore.nn")
            fvwi = types.ModuleType("fvcore.nn.weight_init")

            def c2_msra_fill(module):
                nn.init.kaiming_normal_(module.weight, mode="fan_out", nonlinearity="relu")
                if module.bias is not None:
                    nn.init.constant_(module.bias, 0)

            fvwi.c2_msra_fill = c2_msra_fill
            fv.nn = fvnn
            fvnn.weight_init = fvwi
            sys.modules["fvcore"] = fv
            sys.modules["fvcore.nn"] = fvnn
            sys.modules["fvcore.nn.weight_init"] = fvwi
        try:
            ns.dc = _load("basicsr.archs.degrad_classify_arch", "basicsr/archs/degrad_classify_arch.py")
        except Exception as e:
            ns.dc = None
            ns.dc_error = e
        return ns
    finally:
        for k in [k for k in sys.modules if k == "basicsr" or k.startswith("basicsr.")]:
            del sys.modules[k]
        sys.modules.update(saved)
