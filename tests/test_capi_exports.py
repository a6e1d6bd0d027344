"""CPU-side checks of the C-ABI boundary: the library loads, exports every symbol the header
declares, and refuses to run without a GPU (no CPU fallback)."""
import ctypes as C
import os
import re

import pytest

import reseek_amd
from reseek_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    txt = open(os.path.join(ROOT, "include", "reseek_amd.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(rsk_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    L = reseek_amd.lib()
    syms = header_symbols()
    assert len(syms) >= 12
    for s in syms:
        assert hasattr(L, s), f"librsk.so does not export {s}"
        assert s in capi.SIGNATURES, f"ctypes binding lacks {s}"
    for s in capi.SIGNATURES:
        assert s in syms, f"binding declares {s} which is not in the header"


def test_version_string():
    assert b"gfx950" in reseek_amd.lib().rsk_version()


def test_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(reseek_amd.RskError) as e:
        reseek_amd.Ctx(0)
    assert "no CPU fallback" in str(e.value) or "HIP" in str(e.value)


def test_null_arguments_are_errors_not_crashes():
    L = reseek_amd.lib()
    assert L.rsk_ctx_create(0, None) == -1
    assert L.rsk_ctx_sync(None) == -1
    assert L.rsk_mu_gapless_matrix_dev(None, None, None, 0, None, 0) == -1
    assert b"NULL" in L.rsk_last_error()
