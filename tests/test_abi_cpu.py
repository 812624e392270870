"""CPU: the C-ABI library builds, loads and exports every symbol include/dcpt_hip.h declares;
the Python binding table mirrors the header; the product path refuses CPU tensors (no fallback)."""
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    txt = open(os.path.join(ROOT, "include", "dcpt_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(dcpt_[a-z0-9_]+)\s*\(", txt)))


def test_library_builds_and_exports_header_symbols():
    from dcpt_amd import _lib, build

    build.build()
    lib = _lib.load()
    syms = _header_symbols()
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in dcpt_hip.h but not exported"
    assert sorted(_lib.SIGNATURES.keys()) == syms, "ctypes table and header disagree"
    assert lib.dcpt_abi_version() == _lib.ABI_VERSION == 15


def test_workspace_queries_need_no_gpu():
    from dcpt_amd import _lib

    lib = _lib.load()
    assert lib.dcpt_nafblock_fwd_ws_bytes(32, 256, 256, 64) > 0
    assert lib.dcpt_nafblock_bwd_ws_bytes(32, 32, 32, 512) > lib.dcpt_nafblock_fwd_ws_bytes(32, 32, 32, 512)
    assert lib.dcpt_down2x2_ws_bytes(2, 8, 8, 16, 1) > lib.dcpt_down2x2_ws_bytes(2, 8, 8, 16, 0)


def test_bad_arguments_are_reported_not_crashed():
    from dcpt_amd import _lib

    lib = _lib.load()
    rc = lib.dcpt_ln2d_fwd(None, None, None, None, None, None, 4, 8, 1e-6, None)
    assert rc != 0 and b"null" in lib.dcpt_last_error()


def test_bottleneck_entry_points_and_launch_trace_without_a_gpu():
    """ABI 15 on the CPU: the workspace query needs no device, bad arguments are reported (not crashed) before any launch, the launch trace
    starts empty and returns NUL-terminated text"""
    import ctypes as C

    from dcpt_amd import _lib

    lib = _lib.load()
    fwd, bwd = lib.dcpt_bottleneck_bf16_ws_bytes(2, 16, 16, 64, 0), lib.dcpt_bottleneck_bf16_ws_bytes(2, 16, 16, 64, 1)
    assert 0 < fwd < bwd
    assert lib.dcpt_bottleneck_bf16_ws_bytes(32, 256, 256, 64, 1) > 2 * 32 * 256 * 256 * 64   # (three gradient maps of the widths 128 / 128 / 64 in bf16)
    g = (_lib.BneckGroup * 3)()
    assert lib.dcpt_bottleneck_fwd_bf16(None, g, None, 0, 2, 16, 16, 64, None) != 0 and b"null" in lib.dcpt_last_error()
    assert lib.dcpt_bottleneck_fwd_bf16(1, g, None, 0, 2, 16, 16, 60, None) != 0 and b"multiple of 8" in lib.dcpt_last_error()
    assert lib.dcpt_bottleneck_bwd_bf16(1, 1, g, None, None, 0, 2, 16, 16, 64, None) != 0 and b"null" in lib.dcpt_last_error()
    assert lib.dcpt_trace_enable(1) == 0
    need = lib.dcpt_trace_read(None, 0)
    buf = C.create_string_buffer(64)
    assert need == 1 and lib.dcpt_trace_read(buf, 64) == 1 and buf.value == b""
    assert lib.dcpt_trace_enable(0) == 0


def test_no_cpu_fallback():
    from dcpt_amd import _lib
    from dcpt_amd import functional as DF

    x = torch.zeros(1, 8, 4, 4)
    with pytest.raises(_lib.DcptHipError):
        DF.layernorm2d(x, torch.ones(8), torch.zeros(8))


def test_product_never_imports_oracle():
    for base in ("dcpt_amd", "basicsr"):
        for d, _, files in os.walk(os.path.join(ROOT, base)):
            for f in files:
                if f.endswith(".py"):
                    src = open(os.path.join(d, f)).read()
                    assert "oracle" not in re.sub(r'""".*?"""', "", src, flags=re.S), f"{d}/{f} mentions the oracle"


def test_graft_entry_build_runs():
    """the driver's build check (__graft_entry__.build): compiles / loads the library, checks the ABI version against the binding table and
    imports the package -- it once pinned a literal ABI number and broke when the ABI grew"""
    import __graft_entry__ as g

    g.build()


def test_device_code_has_no_packed_fp32_instructions(tmp_path):
    """dcpt_amd/build.py builds without packed-fp32 VALU instructions (their operand-select forms are not safe next to another stream's
    bf16 MFMA GEMMs on this part: LABNOTES.md 4h).  Compile three sources that used them most with the product's flags and look at the ISA."""
    import re
    import subprocess
    from concurrent.futures import ThreadPoolExecutor

    from dcpt_amd import build as B

    def isa(src):
        out = tmp_path / (src + ".s")
        r = subprocess.run([B._hipcc(), *B.FLAGS, "--cuda-device-only", "-S", os.path.join(B.CSRC, src + ".hip"), "-o", str(out)],
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        return out.read_text()

    with ThreadPoolExecutor(3) as ex:
        texts = list(ex.map(isa, ["conv3x3", "bf16_ops", "misc"]))
    for t in texts:
        assert "s_endpgm" in t
        assert not re.search(r"\bv_pk_(mul|fma|add)_f32\b", t)
