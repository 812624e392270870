from .logger import get_env_info, get_root_logger
from .misc import get_time_str, make_exp_dirs, mkdir_and_rename, scandir, set_random_seed

__all__ = ["get_root_logger", "get_env_info", "set_random_seed", "get_time_str", "mkdir_and_rename", "make_exp_dirs",
           "scandir"]
