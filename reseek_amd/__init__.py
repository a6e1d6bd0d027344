"""reseek_amd -- MI355X (gfx950) implementation of reseek's -search hot path.

The product is the C-ABI shared library reseek_amd/librsk.so (include/reseek_amd.h), built from
reseek_amd/csrc/*.hip by __graft_entry__.build().  This package is only the thin ctypes binding
used by tests and bench.py; it fails loudly if the HIP library is missing (there is no CPU path).
"""
from .capi import Ctx, Db, RskError, lib, LIB_PATH  # noqa: F401
