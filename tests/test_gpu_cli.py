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
