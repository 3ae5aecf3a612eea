"""sp1_amd — MI355X (gfx950) backend for SP1's core-shard commit/open hot path.

The product is the C-ABI library `sp1_amd/lib/libsp1hip.so` (declared in include/sp1hip.h, sources in
sp1_amd/csrc). `sp1_amd.api` is a thin host-side mirror of the reference's operator interfaces used
by the tests and by bench.py; importing it requires torch only as the owner of device memory.
"""
from . import _lib  # noqa: F401

__all__ = ["_lib"]
