import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
if os.path.dirname(os.path.abspath(__file__)) not in sys.path:   # (helpers next to the tests: kernel_trace.py)
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

GOLDEN = os.path.join(ROOT, "tests", "golden")

if os.environ.get("DCPT_TOOL_LIB"):   # development only: run the suite against a variant build (tools/build_variant.sh) instead of the product library
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import _variant  # noqa: E402,F401


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session", autouse=True)
def _built_library():
    """(re)build libdcpt_hip.so if the sources changed -- a digest check, seconds when up to date"""
    from dcpt_amd import build

    build.build()


# ---- the fp32 parity suite a second time with the wide GEMMs in the split-operand mode ("bf16x3", dcpt_amd/csrc/gemm_x3.hip) ----------
# Every GPU test of these modules runs once with the exact fp32 MFMA kernels (the product default and the headline arithmetic) and once
# with the mode FORCED onto every eligible launch (min_tiles = 1), at the SAME fp32 tolerances: the evidence behind reporting that
# mode as an fp32-class second line in bench.py.  Module-scoped fixtures (the trained PSNR-gate network) are built in fp32 mode.
X3_MODULES = ("test_gpu_parity", "test_gpu_configs", "test_gpu_dcpt_step", "test_gpu_streams")


def pytest_generate_tests(metafunc):
    mod = metafunc.module.__name__.rsplit(".", 1)[-1]
    if mod in X3_MODULES and "gemm_mode" in metafunc.fixturenames:
        gpu = metafunc.definition.get_closest_marker("gpu") is not None
        metafunc.parametrize("gemm_mode", ["fp32", "bf16x3"] if gpu else ["fp32"], indirect=True)


@pytest.fixture(autouse=True)
def gemm_mode(request):
    mode = getattr(request, "param", "fp32")
    if mode == "fp32":
        yield mode
        return
    from dcpt_amd import functional as DF

    misses = DF.gemm_x3_scratch_misses()
    DF.set_gemm_precision("bf16x3", min_tiles=1, scratch_mb=768)   # (per-image conv3 weights of a stacked B = 64 batch at C = 1024: 403 MB)
    try:
        yield mode
    finally:
        DF.set_gemm_precision("fp32")
    assert DF.gemm_x3_scratch_misses() == misses, "a launch fell back to the fp32 kernels: the split-image scratch is too small"
