"""GPU: the reference's only entry point, ``python basicsr/test.py -opt <yml>`` (reference basicsr/test.py:21-70), run as a
separate process on the shipped option files -- registry lookup, SRModel pre_test padding / validation loop / metrics and the
HIP kernels underneath, end to end.  The datasets fall back to seeded synthetic pairs (no data offline); the weights are the
constructor's initialisation (no checkpoint offline), so the numbers only have to be sane and reproducible."""
import os
import re
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(yml, extra=()):
    env = dict(os.environ, PYTHONPATH=ROOT)
    cmd = [sys.executable, os.path.join(ROOT, "basicsr", "test.py"), "-opt", os.path.join(ROOT, "options", "all_in_one", "test", yml),
           *extra]
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    out = p.stdout + p.stderr
    vals = {}
    for chunk in out.split("Validation ")[1:]:   # "Validation <name>\n\t # psnr: 12.3456\n\t # ssim: 0.1234\n"
        name = chunk.split()[0]
        for metric, v in re.findall(r"#\s*(psnr|ssim):\s*([-0-9.eE+]+)", chunk[:200]):
            vals[(name, metric)] = float(v)
    return out, vals


@pytest.mark.parametrize("yml", ["test_NAFNet_5d.yml", "test_Restormer_5d.yml", "test_PromptIR_5d.yml"])
def test_cli_on_shipped_options(yml):
    out, vals = _run(yml)
    assert {("Rain100L", "psnr"), ("Rain100L", "ssim"), ("CBSD68", "psnr"), ("CBSD68", "ssim")} <= set(vals), out[-1500:]
    for (name, metric), v in vals.items():
        assert (0.0 < v <= 1.0) if metric == "ssim" else (5.0 < v < 80.0), (name, metric, v)
    out2, vals2 = _run(yml)
    assert vals2 == vals, "two runs of the same option file must report identical metrics"


def test_train_cli_dcpt(tmp_path):
    """``python basicsr/train.py -opt options/all_in_one/train/train_DCPT_NAFNet_5d.yml`` scaled down through --force_yml:
    three concatenated degradation sets, DCPT step on the HIP path, periodic checkpoints + validation, then --auto_resume"""
    import shutil

    import torch

    over = ["network_g:width=8", "network_g:enc_blk_nums=[1,1,1,1]", "network_dc:feature_dims=[8,16,32,64]",
            "datasets:train_1:batch_size_per_gpu=4", "datasets:train_1:gt_size=32", "datasets:train_2:gt_size=32",
            "datasets:train_3:gt_size=32", "datasets:train_1:num=8", "datasets:train_2:num=8", "datasets:train_3:num=8",
            "datasets:val_1:size=32", "train:scheduler:periods=[8]", "logger:print_freq=1", "logger:save_checkpoint_freq=3",
            "val:val_freq=3", "train:optim_g:lr=0.001", "train:optim_dc:lr=0.001"]
    env = dict(os.environ, PYTHONPATH=ROOT)
    yml = os.path.join(ROOT, "options", "all_in_one", "train", "train_DCPT_NAFNet_5d.yml")
    exp = os.path.join(ROOT, "experiments", "DCPT_NAFNet_5d")
    shutil.rmtree(exp, ignore_errors=True)
    try:
        p = subprocess.run([sys.executable, os.path.join(ROOT, "basicsr", "train.py"), "-opt", yml, "--force_yml", *over, "train:total_iter=4"],
                           cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
        out = p.stdout + p.stderr
        assert p.returncode == 0, out[-3000:]
        losses = [float(v) for v in re.findall(r"l_classify: ([-0-9.eE+]+)", out)]
        assert len(losses) == 4 and all(0.0 < v < 20.0 for v in losses), out[-2000:]
        assert "# top1:" in out and os.path.exists(os.path.join(exp, "models", "net_dc_3.pth"))
        assert os.path.exists(os.path.join(exp, "training_states", "3.state"))
        p = subprocess.run([sys.executable, os.path.join(ROOT, "basicsr", "train.py"), "-opt", yml, "--auto_resume", "--force_yml", *over,
                            "train:total_iter=7"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
        out = p.stdout + p.stderr
        assert p.returncode == 0, out[-3000:]
        assert "Resuming training from epoch" in out and "iter: 3" in out
        st = torch.load(os.path.join(exp, "training_states", "6.state"), weights_only=False)
        assert st["iter"] == 6 and len(st["optimizers"]) == 2 and len(st["schedulers"]) == 2
    finally:
        shutil.rmtree(exp, ignore_errors=True)
