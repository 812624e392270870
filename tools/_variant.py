"""Diagnostic tools only: run against a variant build of the library (tools/build_variant.sh) by pointing DCPT_TOOL_LIB at it.
The product loader (dcpt_amd/_lib.py) has no such override; this module patches its path before the first load."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dcpt_amd import _lib  # noqa: E402

if os.environ.get("DCPT_TOOL_LIB"):
    assert _lib._lib is None, "select the variant before the library is loaded"
    _lib.LIB_PATH = os.path.abspath(os.environ["DCPT_TOOL_LIB"])
