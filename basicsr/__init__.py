"""Drop-in mirror of the reference's ``basicsr`` plugin surface for the DCPT hot path
(ARCH_REGISTRY / MODEL_REGISTRY / ``python basicsr/test.py -opt <yaml>``), backed by the
MI355X kernels in ``dcpt_amd`` (libdcpt_hip.so).  Only the path named in DESIGN.md is here."""
__version__ = "0.1.0"
