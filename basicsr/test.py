"""``python basicsr/test.py -opt <yml>``: the reference's only entry point (basicsr/test.py:21-70)."""
import logging
import sys
from os import path as osp

sys.path.insert(0, osp.abspath(osp.join(osp.dirname(__file__), osp.pardir)))

from basicsr.data import build_dataloader, build_dataset  # noqa: E402
from basicsr.models import build_model  # noqa: E402
from basicsr.utils import get_env_info, get_root_logger, get_time_str, make_exp_dirs  # noqa: E402
from basicsr.utils.options import dict2str, parse_options  # noqa: E402


def test_pipeline(root_path, argv=None):
    opt, _ = parse_options(root_path, is_train=False, argv=argv)
    make_exp_dirs(opt)
    log_file = osp.join(opt["path"]["log"], f"test_{opt['name']}_{get_time_str()}.log")
    logger = get_root_logger(logger_name="basicsr", log_level=logging.INFO, log_file=log_file)
    logger.info(get_env_info())
    logger.info(dict2str(opt))

    loaders = []
    for _, dataset_opt in sorted(opt["datasets"].items()):
        test_set = build_dataset(dataset_opt)
        loaders.append(build_dataloader(test_set, dataset_opt, num_gpu=opt["num_gpu"], dist=opt["dist"], sampler=None,
                                        seed=opt["manual_seed"]))
        logger.info(f"Number of test images in {dataset_opt['name']}: {len(test_set)}")

    model = build_model(opt)
    results = {}
    for loader in loaders:
        name = loader.dataset.opt["name"]
        logger.info(f"Testing {name}...")
        results[name] = model.validation(loader, current_iter=opt["name"], tb_logger=None, save_img=opt["val"]["save_img"])
    return results


if __name__ == "__main__":
    test_pipeline(osp.abspath(osp.join(__file__, osp.pardir, osp.pardir)))
