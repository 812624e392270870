"""bench_extra.py against a variant build of the library (DCPT_TOOL_LIB=experiments/lib/...so): same arguments as bench_extra.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); import _variant  # noqa: E401,F401
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench_extra
bench_extra.main()
