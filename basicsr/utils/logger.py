"""Root logger of the mirror (reference basicsr/utils/logger.py:156-195: logger "basicsr",
rank-0 at the requested level, other ranks at ERROR, optional file handler)."""
import logging

_initialized = set()


def get_root_logger(logger_name="basicsr", log_level=logging.INFO, log_file=None):
    logger = logging.getLogger(logger_name)
    if logger_name in _initialized:
        return logger
    fmt = logging.Formatter("%(asctime)s %(levelname)s: %(message)s")
    sh = logging.StreamHandler()
    sh.setFormatter(fmt)
    logger.addHandler(sh)
    logger.propagate = False
    from .dist_util import get_dist_info

    rank, _ = get_dist_info()
    if rank != 0:
        logger.setLevel(logging.ERROR)
    else:
        logger.setLevel(log_level)
        if log_file is not None:
            fh = logging.FileHandler(log_file, "w")
            fh.setFormatter(fmt)
            fh.setLevel(log_level)
            logger.addHandler(fh)
    _initialized.add(logger_name)
    return logger


def get_env_info():
    import torch

    return f"\nbasicsr (dcpt_amd mirror)\n\tPyTorch: {torch.__version__}\n\tHIP: {getattr(torch.version, 'hip', None)}"
