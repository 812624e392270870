"""YAML options + CLI of the mirror (reference basicsr/utils/options.py:14-205): same flags
(-opt, --launcher, --auto_resume, --debug, --local_rank, --force_yml k:k=v), same derived keys."""
from __future__ import annotations

import argparse
import os
import random
from collections import OrderedDict
from os import path as osp

import torch
import yaml

from .dist_util import get_dist_info, init_dist
from .misc import set_random_seed


class _OrderedLoader(yaml.SafeLoader):
    pass


_OrderedLoader.add_constructor(yaml.resolver.BaseResolver.DEFAULT_MAPPING_TAG,
                               lambda loader, node: OrderedDict(loader.construct_pairs(node)))


def yaml_load(f):
    """Load a YAML file path or YAML text into an OrderedDict."""
    if os.path.isfile(f):
        with open(f, "r") as fh:
            return yaml.load(fh, Loader=_OrderedLoader)
    return yaml.load(f, Loader=_OrderedLoader)


def dict2str(opt, indent_level=1):
    pad = " " * (indent_level * 2)
    out = "\n"
    for k, v in opt.items():
        if isinstance(v, dict):
            out += f"{pad}{k}:[{dict2str(v, indent_level + 1)}{pad}]\n"
        else:
            out += f"{pad}{k}: {v}\n"
    return out


def _parse_value(text: str):
    low = text.lower()
    if text == "~" or low == "none":
        return None
    if low in ("true", "false"):
        return low == "true"
    if text.startswith("!!float"):
        return float(text.replace("!!float", ""))
    if text.isdigit():
        return int(text)
    if text.replace(".", "", 1).isdigit() and text.count(".") < 2:
        return float(text)
    if text.startswith("["):
        return yaml.safe_load(text)
    return text


def apply_force_yml(opt, entries):
    """``a:b:c=value`` overrides of existing keys (the reference does this with exec, options.py:145-156)."""
    for entry in entries or []:
        keys, value = entry.split("=", 1)
        node = opt
        path = [k.strip() for k in keys.strip().split(":")]
        for k in path[:-1]:
            node = node[k]
        if path[-1] not in node:
            raise KeyError(f"--force_yml cannot create new key {keys}")
        node[path[-1]] = _parse_value(value.strip())
    return opt


def parse_options(root_path, is_train=True, argv=None):
    parser = argparse.ArgumentParser()
    parser.add_argument("-opt", type=str, required=True, help="Path to option YAML file.")
    parser.add_argument("--launcher", choices=["none", "pytorch", "slurm"], default="none", help="job launcher")
    parser.add_argument("--auto_resume", action="store_true")
    parser.add_argument("--debug", action="store_true")
    parser.add_argument("--local_rank", type=int, default=0)
    parser.add_argument("--force_yml", nargs="+", default=None, help="e.g. train:ema_decay=0.999")
    args = parser.parse_args(argv)

    opt = yaml_load(args.opt)
    if args.launcher == "none":
        opt["dist"] = False
        print("Disable distributed.", flush=True)
    else:
        opt["dist"] = True
        if args.launcher == "slurm" and "dist_params" in opt:
            init_dist(args.launcher, **opt["dist_params"])
        else:
            init_dist(args.launcher)
    opt["rank"], opt["world_size"] = get_dist_info()

    seed = opt.get("manual_seed")
    if seed is None:
        seed = random.randint(1, 10000)
        opt["manual_seed"] = seed
    set_random_seed(seed + opt["rank"])

    apply_force_yml(opt, args.force_yml)
    opt["auto_resume"] = args.auto_resume
    opt["is_train"] = is_train
    if args.debug and not opt["name"].startswith("debug"):
        opt["name"] = "debug_" + opt["name"]
    if opt["num_gpu"] == "auto":
        opt["num_gpu"] = torch.cuda.device_count()

    for phase, dataset in opt["datasets"].items():
        dataset["phase"] = phase.split("_")[0]
        if "scale" in opt:
            dataset["scale"] = opt["scale"]
        for key in ("dataroot_gt", "dataroot_lq"):
            if dataset.get(key) is not None:
                dataset[key] = osp.expanduser(dataset[key])
    for key, val in opt["path"].items():
        if val is not None and ("resume_state" in key or "pretrain_network" in key):
            opt["path"][key] = osp.expanduser(val)

    if is_train:
        root = osp.join(root_path, "experiments", opt["name"])
        opt["path"].update(experiments_root=root, models=osp.join(root, "models"),
                           training_states=osp.join(root, "training_states"), log=root,
                           visualization=osp.join(root, "visualization"))
        if "debug" in opt["name"]:
            if "val" in opt:
                opt["val"]["val_freq"] = 8
            opt["logger"]["print_freq"] = 1
            opt["logger"]["save_checkpoint_freq"] = 8
    else:
        root = osp.join(root_path, "results", opt["name"])
        opt["path"].update(results_root=root, log=root, visualization=osp.join(root, "visualization"))
    return opt, args
