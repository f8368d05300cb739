"""fnssl — MI355X-native runtime for the FN-SSL DP-IPD forward path.

Host side only (Python); every numeric op on the path is a hand-written HIP
kernel reached through the C-ABI library ``csrc/libfnssl_hip.so``
(declarations: ``include/fnssl.h``).  There is deliberately no CPU fallback:
calling an op without the library or without a ROCm device raises.
"""
from . import weights  # noqa: F401

__all__ = ["weights"]
