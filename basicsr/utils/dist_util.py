"""Distributed bootstrap (reference basicsr/utils/dist_util.py:11-82).  One process per GPU;
backend "nccl" is RCCL on ROCm.  Rendezvous via the env:// variables torchrun sets."""
import functools
import os
import subprocess

import torch
import torch.distributed as dist


def init_dist(launcher, backend="nccl", **kwargs):
    if launcher == "pytorch":
        rank = int(os.environ["RANK"])
        if torch.cuda.is_available():
            torch.cuda.set_device(rank % torch.cuda.device_count())
        else:
            backend = "gloo"
        dist.init_process_group(backend=backend, **kwargs)
    elif launcher == "slurm":
        proc_id = int(os.environ["SLURM_PROCID"])
        ntasks = int(os.environ["SLURM_NTASKS"])
        node_list = os.environ["SLURM_NODELIST"]
        addr = subprocess.getoutput(f"scontrol show hostname {node_list} | head -n1")
        os.environ.setdefault("MASTER_PORT", str(kwargs.pop("port", 29500)))
        os.environ["MASTER_ADDR"] = addr
        os.environ["WORLD_SIZE"] = str(ntasks)
        os.environ["LOCAL_RANK"] = str(proc_id % max(1, torch.cuda.device_count()))
        os.environ["RANK"] = str(proc_id)
        if torch.cuda.is_available():
            torch.cuda.set_device(proc_id % torch.cuda.device_count())
        dist.init_process_group(backend=backend)
    else:
        raise ValueError(f"Invalid launcher type: {launcher}")


def get_dist_info():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def master_only(func):
    @functools.wraps(func)
    def wrapper(*args, **kwargs):
        if get_dist_info()[0] == 0:
            return func(*args, **kwargs)

    return wrapper
