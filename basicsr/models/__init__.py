"""build_model + auto-import of ``*_model.py`` (reference basicsr/models/__init__.py:13-38)."""
import importlib
from copy import deepcopy
from os import path as osp

from basicsr.utils import get_root_logger, scandir
from basicsr.utils.registry import MODEL_REGISTRY

__all__ = ["build_model"]

_here = osp.dirname(osp.abspath(__file__))
_model_modules = [
    importlib.import_module(f"basicsr.models.{osp.splitext(osp.basename(f))[0]}")
    for f in scandir(_here)
    if f.endswith("_model.py")
]


def build_model(opt):
    opt = deepcopy(opt)
    model = MODEL_REGISTRY.get(opt["model_type"])(opt)
    get_root_logger().info(f"Model [{model.__class__.__name__}] is created.")
    return model
