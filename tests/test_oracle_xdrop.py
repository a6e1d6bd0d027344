"""CPU: the oracle's X-drop restatement (oracle/rsk_oracle.c rsko_xdrop_fwd / _bwd / rsko_merge_fwd_bwd) against the
reference's own functions on REAL long-chain pairs: tests/golden/xdrophsp_palms_sensitive.bin.gz holds, for every
long-chain pair of palms.bca, the start XDropHSP derived and the XDropFwd / XDropBwd score bits and paths the reference
computed from it (oracle/ref_harness xdrophsp; chains of 600+ residues, bands of hundreds of columns -- the rows that grow
and re-open columns, which the peptide vectors of test_xdrop_kat.py barely reach).  With this the oracle is a pinned
checker for the device kernel on data no fixture holds (tests/test_gpu_xdrop.py)."""
import ctypes as C
import struct

import numpy as np

import fixtures as fx
import oracle_lib as ol


def bits(x):
    return struct.unpack("<I", struct.pack("<f", x))[0]


def smx(pa, pb):
    pa, pb = np.ascontiguousarray(pa), np.ascontiguousarray(pb)
    S = np.zeros((pa.shape[1], pb.shape[1]), np.float32)
    ol.lib().rsko_set_smx(pa.ctypes.data_as(C.POINTER(C.c_uint8)), pa.shape[1], pb.ctypes.data_as(C.POINTER(C.c_uint8)), pb.shape[1],
                          S.ctypes.data_as(C.POINTER(C.c_float)))
    return S


import pytest


@pytest.mark.parametrize("name,min_pairs", [("palms", 500), ("taildb", 900)])
def test_oracle_xdrop_matches_the_reference_on_long_chain_pairs(name, min_pairs):
    """palms: the reference's own test chains (418 .. 2,099 residues); taildb: 48 synthetic chains of 17 .. 5,000 residues."""
    chains = fx.read_rskdb(name + "_sensitive.rskdb.gz")
    n, recs = fx.read_xdrophsp("xdrophsp_" + name + "_sensitive.bin.gz")
    assert n == len(chains)
    gated = [r for r in recs if r["gate"]]
    assert len(gated) > min_pairs
    X, go, ge = 8.0, -0.685533, -0.051881
    nlong = nmerged = 0
    for r in gated:
        S = smx(chains[r["i"]].prof, chains[r["j"]].prof)
        sf, pf = ol.xdrop_fwd(S, X, go, ge, r["lo_a"], r["lo_b"])
        sb, pb = ol.xdrop_bwd(S, X, go, ge, r["lo_a"] - 1, r["lo_b"] - 1)
        assert (bits(sf), pf) == r["fwd"], (r["i"], r["j"], "fwd")
        assert (bits(sb), pb) == r["bwd"], (r["i"], r["j"], "bwd")
        nlong += len(pf) > 100 or len(pb) > 100
        if r["path"]:
            lo_a, lo_b, _, _, path = ol.merge_fwd_bwd(r["lo_a"], r["lo_b"], pf, r["lo_a"] - 1, r["lo_b"] - 1, pb)
            assert path == r["path"] and (lo_a, lo_b) == (r["mlo_a"], r["mlo_b"]), (r["i"], r["j"], "merge")
            total = np.float32(sf) + np.float32(sb)
            assert bits(float(total)) == r["total"], (r["i"], r["j"], "total")
            nmerged += 1
    assert nlong > 20 and nmerged > 100
