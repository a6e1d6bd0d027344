"""Build-container-only boundary check: the reference's own caller of the -search path,
/root/reference/src/search.cpp, must compile UNMODIFIED against reseek_host.h (through the name shim
tests/ref_shaped/shim/) and link against librsk.so.  Nothing of that file is stored in the repo; the product of the build
(oracle/_ref/search_refsrc) travels to the GPU box, where tests/test_gpu_ref_shaped.py runs it against the goldens.
Skipped where /root/reference does not exist (the GPU box)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFSRC = "/root/reference/src/search.cpp"


@pytest.mark.skipif(not os.path.exists(REFSRC), reason="reference sources are only present in the build container")
def test_reference_search_cpp_compiles_against_the_host_layer():
    if not os.path.exists(os.path.join(ROOT, "reseek_amd", "librsk.so")):
        pytest.skip("librsk.so not built yet (__graft_entry__.build())")
    exe = os.path.join(ROOT, "oracle", "_ref", "search_refsrc")
    r = subprocess.run(["make", "-B", "-f", "oracle/Makefile.ref", "oracle/_ref/search_refsrc"], cwd=ROOT, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    syms = subprocess.run(["nm", "-C", exe], capture_output=True, text=True, check=True).stdout
    # defined by the reference's file ...
    assert " T reseek_amd::cmd_search()" in syms
    # ... and resolved by the host layer with the reference's argument lists (search.cpp:9-18)
    assert "U reseek_amd::MuPreFilter(reseek_amd::DSSParams const&, reseek_amd::SeqDB&, reseek_amd::MuSeqSource&, std::" in syms
    assert "U reseek_amd::PostMuFilter(reseek_amd::DSSParams const&, std::" in syms
    for cls in ("DBSearcher::LoadDB", "DBSearcher::Setup", "DBSearcher::RunSelf", "DBSearcher::RunQuery(reseek_amd::ChainReader2&)",
                "ChainReader2::Open", "MuSeqSource::OpenChains", "MuSeqSource::OpenFasta", "SeqDB::FromSS", "DSSParams::SetDSSParams(reseek_amd::DECIDE_MODE)"):
        assert "U reseek_amd::" + cls in syms, cls
    # no text of the reference's file anywhere in the repo's tracked tree: the shim maps names only
    shim = open(os.path.join(ROOT, "tests", "ref_shaped", "shim", "myutils.h")).read()
    assert "SelfSearch" not in shim and "Search_NoMuFilter" not in shim
