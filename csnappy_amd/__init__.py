"""csnappy_amd -- MI355X-native Snappy block codec behind the csnappy.h C API.

The product is csnappy_amd/lib/libcsnappy.so (HIP kernels + C-ABI, sources in csnappy_amd/csrc,
headers in include/).  `csnappy_amd.api` is the ctypes binding used by tests/ and bench.py.
"""
from . import api  # noqa: F401

__all__ = ["api"]
