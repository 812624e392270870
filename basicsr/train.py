"""``python basicsr/train.py -opt <yml> [--launcher pytorch] [--auto_resume]``: iteration-based training of the path's models
(SRModel, DCPTModel / DCTModel / DCModel, DCDistModel).  The reference ships the models but no training entry point
(SURVEY 8f rank 3); this is the standard BasicSR loop over the reference's option schema: ``datasets.train*`` (several
``train_*`` entries are concatenated and labelled with ``dataset_idx`` -- the degradation classes), ``datasets.val*``,
``train.total_iter / warmup_iter / scheduler``, ``logger.print_freq / save_checkpoint_freq``, ``val.val_freq``."""
import datetime
import logging
import math
import sys
import time
from os import path as osp

import torch

sys.path.insert(0, osp.abspath(osp.join(osp.dirname(__file__), osp.pardir)))

from basicsr.data import build_dataloader, build_dataset  # noqa: E402
from basicsr.data.concat_dataset import ConcatDataset  # noqa: E402
from basicsr.data.data_sampler import EnlargedSampler  # noqa: E402
from basicsr.models import build_model  # noqa: E402
from basicsr.utils import get_env_info, get_root_logger, get_time_str, make_exp_dirs, scandir  # noqa: E402
from basicsr.utils.options import dict2str, parse_options  # noqa: E402


def create_train_val_dataloader(opt, logger):
    train_sets, train_opts, val_loaders = [], [], []
    for phase, dataset_opt in opt["datasets"].items():   # file order (the YAML loader builds OrderedDicts): train_2 before train_10,
        # and ``dataset_idx`` -- the degradation label ConcatDataset injects -- counts the train sets in the order they are written
        kind = phase.split("_")[0]
        if kind == "train":
            train_sets.append(build_dataset(dataset_opt))
            train_opts.append(dataset_opt)
        elif kind in ("val", "test"):
            val_set = build_dataset(dataset_opt)
            val_loaders.append(build_dataloader(val_set, dataset_opt, num_gpu=opt["num_gpu"], dist=opt["dist"], sampler=None,
                                                seed=opt["manual_seed"]))
            logger.info(f"Number of val images/folders in {dataset_opt['name']}: {len(val_set)}")
        else:
            raise ValueError(f"Dataset phase {phase} is not recognized.")
    if not train_sets:
        raise ValueError("no datasets.train* entry in the option file")
    first = train_opts[0]
    if len(train_sets) > 1 or any("enlarge_ratio" in o for o in train_opts):
        train_set = ConcatDataset(train_sets, [int(o.get("enlarge_ratio", 1)) for o in train_opts])
    else:
        train_set = train_sets[0]
    ratio = first.get("dataset_enlarge_ratio", 1)
    sampler = EnlargedSampler(train_set, opt["world_size"], opt["rank"], ratio)
    loader = build_dataloader(train_set, first, num_gpu=opt["num_gpu"], dist=opt["dist"], sampler=sampler, seed=opt["manual_seed"])
    per_epoch = math.ceil(len(train_set) * ratio / (first["batch_size_per_gpu"] * opt["world_size"]))
    total_iters = int(opt["train"]["total_iter"])
    total_epochs = math.ceil(total_iters / per_epoch)
    logger.info(f"Training statistics:\n\tNumber of train images: {len(train_set)}\n\tDataset enlarge ratio: {ratio}"
                f"\n\tBatch size per gpu: {first['batch_size_per_gpu']}\n\tWorld size (gpu number): {opt['world_size']}"
                f"\n\tRequire iter number per epoch: {per_epoch}\n\tTotal epochs: {total_epochs}; iters: {total_iters}.")
    return loader, sampler, val_loaders, total_epochs, total_iters


def load_resume_state(opt):
    """``path.resume_state`` or, with --auto_resume, the newest ``<iter>.state`` of this experiment; on resume the networks
    are re-loaded from the checkpoints of that iteration (reference basicsr/utils/options.py check_resume semantics)."""
    path = opt["path"].get("resume_state")
    if opt.get("auto_resume"):
        state_dir = opt["path"]["training_states"]
        if osp.isdir(state_dir):
            states = [int(f.split(".state")[0]) for f in scandir(state_dir, suffix="state")]
            if states:
                path = osp.join(state_dir, f"{max(states)}.state")
                opt["path"]["resume_state"] = path
    if not path:
        return None
    state = torch.load(path, map_location="cpu", weights_only=False)
    for key in [k for k in opt if k.startswith("network_")]:
        tag = key[len("network_"):]
        ckpt = osp.join(opt["path"]["models"], f"net_{tag}_{state['iter']}.pth")
        if osp.exists(ckpt):
            opt["path"][f"pretrain_network_{tag}"] = ckpt
            opt["path"][f"param_key_{tag}"] = "params"
    return state


class MessageLogger:
    """iteration log line (reference basicsr/utils/logger.py:33-105): epoch, iter, lr, eta, data / iter time, losses"""

    def __init__(self, opt, start_iter=1):
        self.exp_name, self.interval = opt["name"], opt["logger"]["print_freq"]
        self.start_iter, self.max_iters = start_iter, opt["train"]["total_iter"]
        self.start_time = time.time()
        self.logger = get_root_logger()

    def __call__(self, log_vars):
        epoch, it, lrs = log_vars.pop("epoch"), log_vars.pop("iter"), log_vars.pop("lrs")
        msg = f"[{self.exp_name[:5]}..][epoch:{epoch:3d}, iter:{it:8,d}, lr:(" + ",".join(f"{v:.3e}" for v in lrs) + ")] "
        if "time" in log_vars:
            iter_time, data_time = log_vars.pop("time"), log_vars.pop("data_time")
            per_iter = (time.time() - self.start_time) / (it - self.start_iter + 1)
            eta = str(datetime.timedelta(seconds=int(per_iter * (self.max_iters - it - 1))))
            msg += f"[eta: {eta}, time (data): {iter_time:.3f} ({data_time:.3f})] "
        msg += " ".join(f"{k}: {v:.4e}" for k, v in log_vars.items())
        self.logger.info(msg)


def train_pipeline(root_path, argv=None):
    opt, args = parse_options(root_path, is_train=True, argv=argv)
    opt["root_path"] = root_path
    torch.backends.cudnn.benchmark = True
    resume_state = load_resume_state(opt)
    if resume_state is None:
        make_exp_dirs(opt)
    log_file = osp.join(opt["path"]["log"], f"train_{opt['name']}_{get_time_str()}.log")
    logger = get_root_logger(logger_name="basicsr", log_level=logging.INFO, log_file=log_file)
    logger.info(get_env_info())
    logger.info(dict2str(opt))

    train_loader, train_sampler, val_loaders, total_epochs, total_iters = create_train_val_dataloader(opt, logger)
    model = build_model(opt)
    if resume_state:
        model.resume_training(resume_state)
        logger.info(f"Resuming training from epoch: {resume_state['epoch']}, iter: {resume_state['iter']}.")
        start_epoch, current_iter = resume_state["epoch"], resume_state["iter"]
    else:
        start_epoch, current_iter = 0, 0
    msg_logger = MessageLogger(opt, current_iter)
    warmup = opt["train"].get("warmup_iter", -1)
    val_freq = opt.get("val", {}).get("val_freq")
    logger.info(f"Start training from epoch: {start_epoch}, iter: {current_iter}")
    data_timer = iter_timer = time.time()
    for epoch in range(start_epoch, total_epochs + 1):
        train_sampler.set_epoch(epoch)
        for train_data in train_loader:
            data_time = time.time() - data_timer
            current_iter += 1
            if current_iter > total_iters:
                break
            model.update_learning_rate(current_iter, warmup_iter=warmup)
            model.feed_data(train_data)
            model.optimize_parameters(current_iter)
            iter_time = time.time() - iter_timer
            if current_iter % opt["logger"]["print_freq"] == 0:
                log_vars = {"epoch": epoch, "iter": current_iter, "lrs": model.get_current_learning_rate(), "time": iter_time,
                            "data_time": data_time}
                log_vars.update(model.get_current_log())
                msg_logger(log_vars)
            if current_iter % opt["logger"]["save_checkpoint_freq"] == 0:
                logger.info("Saving models and training states.")
                model.save(epoch, current_iter)
            if val_freq and current_iter % val_freq == 0:
                for loader in val_loaders:
                    model.validation(loader, current_iter, None, opt["val"].get("save_img", False))
            data_timer = iter_timer = time.time()
        if current_iter > total_iters:
            break
    logger.info("End of training. Save the latest model.")
    model.save(epoch=-1, current_iter=-1)
    results = {}
    if opt.get("val") is not None:
        for loader in val_loaders:
            results[loader.dataset.opt["name"]] = model.validation(loader, current_iter, None, opt["val"].get("save_img", False))
    return model, results


if __name__ == "__main__":
    train_pipeline(osp.abspath(osp.join(__file__, osp.pardir, osp.pardir)))
