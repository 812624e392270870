"""Small host utilities used by the path's callers (reference basicsr/utils/misc.py)."""
import os
import random
import time
from os import path as osp

import numpy as np
import torch

from .dist_util import master_only


def set_random_seed(seed):
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)


def get_time_str():
    return time.strftime("%Y%m%d_%H%M%S", time.localtime())


def mkdir_and_rename(path):
    if osp.exists(path):
        new_name = path + "_archived_" + get_time_str()
        print(f"Path already exists. Rename it to {new_name}", flush=True)
        os.rename(path, new_name)
    os.makedirs(path, exist_ok=True)


@master_only
def make_exp_dirs(opt):
    path_opt = dict(opt["path"])
    root_key = "experiments_root" if opt["is_train"] else "results_root"
    mkdir_and_rename(path_opt.pop(root_key))
    for key, path in path_opt.items():
        if path is None or any(s in key for s in ("strict_load", "pretrain_network", "resume", "param_key")):
            continue
        os.makedirs(path, exist_ok=True)


def scandir(dir_path, suffix=None, recursive=False, full_path=False):
    if suffix is not None and not isinstance(suffix, (str, tuple)):
        raise TypeError('"suffix" must be a string or tuple of strings')
    root = dir_path

    def _walk(d):
        for entry in sorted(os.scandir(d), key=lambda e: e.name):
            if not entry.name.startswith(".") and entry.is_file():
                ret = entry.path if full_path else osp.relpath(entry.path, root)
                if suffix is None or ret.endswith(suffix):
                    yield ret
            elif recursive and entry.is_dir():
                yield from _walk(entry.path)

    return _walk(dir_path)
