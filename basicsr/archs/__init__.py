"""build_network + auto-import of ``*_arch.py`` (reference basicsr/archs/__init__.py:12-31)."""
import importlib
from copy import deepcopy
from os import path as osp

from basicsr.utils import get_root_logger, scandir
from basicsr.utils.registry import ARCH_REGISTRY

__all__ = ["build_network"]

_here = osp.dirname(osp.abspath(__file__))
_arch_modules = [
    importlib.import_module(f"basicsr.archs.{osp.splitext(osp.basename(f))[0]}")
    for f in scandir(_here)
    if f.endswith("_arch.py")
]


def build_network(opt):
    opt = deepcopy(opt)
    net = ARCH_REGISTRY.get(opt.pop("type"))(**opt)
    get_root_logger().info(f"Network [{net.__class__.__name__}] is created.")
    return net
