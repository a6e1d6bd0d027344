"""rsk_dss_densities (k_dss.hip): the two density features of DSS (dss.cpp:217-244, 339-372) for chains and their reversed
copies on the device.  The device exp() is not libm's, so the values are compared with a numpy restatement to a relative
1e-12 here; the LETTERS the host bins from them are the host's by construction (DSS::UseDeviceDensities keeps a chain's
device values only if nothing binned is within 1e-9 of a bin boundary) -- the search tests against the goldens and the
reference binary run with the device densities, and the last test compares a search with and without them."""
import os

import numpy as np
import pytest

import fixtures as fx

pytestmark = pytest.mark.gpu
DBL_MAX = np.finfo(np.float64).max


@pytest.fixture(scope="module")
def ctx():
    import torch
    import reseek_amd
    assert torch.cuda.is_available()
    c = reseek_amd.Ctx(0, stream=torch.cuda.current_stream().cuda_stream)
    yield c
    c.close()


def ref_densities(x, y, z, ss, W=50, w1=3, w2=8, radius=20.0, eps=1.0):
    """SetDensities of one chain: float distances (pdbchain.cpp:310), double exp, sums in ascending position."""
    L = len(x)
    dens = np.full(L, DBL_MAX)
    sd = np.full(L, DBL_MAX)
    x, y, z = (np.asarray(v, np.float32) for v in (x, y, z))
    for pos in range(1, L - 1):
        q = np.arange(max(0, pos - W), min(L - 1, pos + W) + 1)
        dx, dy, dz = x[pos] - x[q], y[pos] - y[q], z[pos] - z[q]
        d2 = (dx * dx + dy * dy).astype(np.float32) + (dz * dz).astype(np.float32)
        f = np.exp(-np.sqrt(d2.astype(np.float32)).astype(np.float64) / radius)
        k = np.abs(q - pos)
        dens[pos] = f[k > w1].sum()
        far = k > w2
        d2sum = f[far].sum()
        dc = f[far & (ss[q] == ord("s"))].sum()
        sd[pos] = dc / (d2sum + eps)
    return dens, sd


def ref_neighbours(x, y, z, NW=100, Nw=12):
    """CalcNEN / CalcREN (dss.cpp:374-440): first nearest in ascending position, running minimum from 999."""
    L = len(x)
    x, y, z = (np.asarray(v, np.float32) for v in (x, y, z))
    NONE = 0xFFFFFFFF
    nen = np.full(L, NONE, np.uint32)
    ren = np.full(L, NONE, np.uint32)

    def nearest(pos, lo, hi):
        q = np.arange(lo, hi + 1)
        q = q[np.abs(q - pos) > Nw]
        if len(q) == 0:
            return NONE
        dx, dy, dz = x[pos] - x[q], y[pos] - y[q], z[pos] - z[q]
        d2 = (dx * dx + dy * dy).astype(np.float32) + (dz * dz).astype(np.float32)
        d = np.sqrt(d2.astype(np.float32)).astype(np.float64)
        k = int(np.argmin(d))                      # first minimum
        return int(q[k]) if d[k] < 999 else NONE

    for pos in range(L):
        lo, hi = max(0, pos - NW), min(L - 1, pos + NW)
        n = nearest(pos, lo, hi)
        nen[pos] = n
        if n != NONE:
            ren[pos] = nearest(pos, lo, pos - 1) if n > pos else nearest(pos, pos + 1, hi)
    return nen, ren


def dist64(x, y, z, a, b):
    """(double) PDBChain::GetDist: float expression, float sqrt"""
    dx, dy, dz = np.float32(x[a] - x[b]), np.float32(y[a] - y[b]), np.float32(z[a] - z[b])
    d2 = np.float32(np.float32(np.float32(dx * dx) + np.float32(dy * dy)) + np.float32(dz * dz))
    return np.float64(np.sqrt(d2, dtype=np.float32))


CONF_MEANS = None


def conf_means():
    """the 16 x 9 cluster centres of reseek_amd/csrc/host/dss_data.h"""
    global CONF_MEANS
    if CONF_MEANS is None:
        import re
        txt = open(os.path.join(os.path.dirname(fx.GOLDEN), "..", "reseek_amd", "csrc", "host", "dss_data.h")).read()
        rows = re.findall(r"\{\s*([0-9.,\s]+)\}", txt[txt.index("rsk_conf_means"):])
        CONF_MEANS = np.array([[float(v) for v in r.split(",") if v.strip()] for r in rows[:16]], np.float64)
        assert CONF_MEANS.shape == (16, 9)
    return CONF_MEANS


def ref_local(x, y, z):
    """PDBChain::GetSS (getss.cpp:6-60) and DSS::ConfLetter (myss.cpp:125-160)"""
    L = len(x)
    x, y, z = (np.asarray(v, np.float32) for v in (x, y, z))
    ss = np.full(L, ord("~"), np.uint8)
    conf = np.full(L, 255, np.uint8)
    M = conf_means()
    iv, jv = (-2, -2, -2, -1, -1, 0, -3, 0, -3), (0, 1, 2, 1, 2, 2, 3, 3, 0)
    for p in range(L):
        if 2 <= p < L - 2:
            d13, d14, d15 = dist64(x, y, z, p - 2, p), dist64(x, y, z, p - 2, p + 1), dist64(x, y, z, p - 2, p + 2)
            d24, d25, d35 = dist64(x, y, z, p - 1, p + 1), dist64(x, y, z, p - 1, p + 2), dist64(x, y, z, p, p + 2)
            if all(abs(d - c) < 2.1 for d, c in ((d15, 6.37), (d14, 5.18), (d25, 5.18), (d13, 5.45), (d24, 5.45), (d35, 5.45))):
                ss[p] = ord("h")
            elif all(abs(d - c) < 1.42 for d, c in ((d15, 13), (d14, 10.4), (d25, 10.4), (d13, 6.1), (d24, 6.1), (d35, 6.1))):
                ss[p] = ord("s")
            elif d15 < 8.2:
                ss[p] = ord("t")
        if 3 <= p < L - 3:
            v = np.array([dist64(x, y, z, p + i, p + j) for i, j in zip(iv, jv)], np.float64)
            best, mind = 0, None
            for k in range(16):
                s2 = np.float64(0)
                for m in range(9):
                    d = v[m] - M[k][m]
                    s2 = s2 + d * d
                dk = np.sqrt(s2)
                if k == 0 or dk < mind:
                    best, mind = k, dk
            conf[p] = best
    return ss, conf


def test_device_densities_match_a_numpy_restatement(ctx):
    rng = np.random.default_rng(11)
    lens = np.array([1, 2, 3, 4, 5, 6, 7, 9, 17, 60, 61, 130, 411, 1203], np.uint32)
    tot = int(lens.sum())
    # persistent random walks with 3.8 A steps: helix- and strand-like stretches occur (SS letters h / s / t / ~ all appear)
    from scipy.signal import lfilter
    d = lfilter([0.6], [1.0, -0.8], rng.normal(0, 1, (tot, 3)) / 0.6, axis=0)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    walk = np.cumsum(3.8 * d, axis=0)
    xyz = [np.ascontiguousarray(walk[:, k], np.float32) for k in range(3)]
    r = ctx.dss_densities(lens, *xyz)
    df, sf, dr, sr = r["dens_fwd"], r["sdens_fwd"], r["dens_rev"], r["sdens_rev"]
    nf, rf, nr, rr = r["nen_fwd"], r["ren_fwd"], r["nen_rev"], r["ren_rev"]
    ssf, ssr = r["ss_fwd"], r["ss_rev"]
    assert len(set(ssf.tolist())) >= 3
    o = 0
    for L in lens:
        L = int(L)
        sl = slice(o, o + L)
        x, y, z = (v[sl] for v in xyz)
        for got_d, got_s, want in ((df[sl], sf[sl], ref_densities(x, y, z, ssf[sl])),
                                   (dr[sl], sr[sl], ref_densities(x[::-1], y[::-1], z[::-1], ssr[sl]))):
            wd, ws = want
            assert np.array_equal(got_d == DBL_MAX, wd == DBL_MAX) and np.array_equal(got_s == DBL_MAX, ws == DBL_MAX)
            m = wd != DBL_MAX
            assert np.allclose(got_d[m], wd[m], rtol=1e-12, atol=1e-13) and np.allclose(got_s[m], ws[m], rtol=1e-12, atol=1e-13)
        # SS characters and Conf letters: comparison chains on float / double values, exact
        for got_s, got_c, want in ((ssf[sl], r["conf_fwd"][sl], ref_local(x, y, z)), (ssr[sl], r["conf_rev"][sl], ref_local(x[::-1], y[::-1], z[::-1]))):
            assert np.array_equal(got_s, want[0]) and np.array_equal(got_c, want[1])
        # nearest neighbours: float arithmetic only, so the positions are the host's exactly
        for got_n, got_r, want in ((nf[sl], rf[sl], ref_neighbours(x, y, z)), (nr[sl], rr[sl], ref_neighbours(x[::-1], y[::-1], z[::-1]))):
            assert np.array_equal(got_n, want[0]) and np.array_equal(got_r, want[1])
        o += L


def test_acceptance_margin_is_orders_above_the_observed_device_vs_libm_difference(ctx):
    """DSS::UseDeviceDensities keeps a chain's device densities only if every binned quantity is further than 1e-9 from all
    bin boundaries.  The bound behind that margin, measured: over 4,000 SCOP40-length chains and their reversed copies
    (> 1e8 device exp() evaluations) the device values differ from the host's (glibc exp, same summation order) by less
    than 1e-12 absolute -- a density is a sum of <= 100 terms in (0, 1], each within an ulp or two -- so the margin is more
    than a thousand times the largest difference ever seen and a letter cannot flip inside it."""
    from scipy.signal import lfilter
    from reseek_amd import capi
    rng = np.random.default_rng(2024)
    lens = fx.scop40_lengths()[rng.choice(11211, 4000, replace=False)].astype(np.uint32)
    tot = int(lens.sum())
    d = lfilter([0.6], [1.0, -0.8], rng.normal(0, 1, (tot, 3)) / 0.6, axis=0)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    walk = np.cumsum(3.8 * d, axis=0)
    start = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    for k in range(len(lens)):                                          # every chain around the origin, .bca-like magnitudes
        walk[start[k]:start[k + 1]] -= walk[start[k]:start[k + 1]].mean(axis=0)
    xyz = [np.ascontiguousarray(walk[:, k], np.float32) for k in range(3)]
    r = ctx.dss_densities(lens, *xyz)
    nexp = 0
    worst = 0.0
    for k in range(len(lens)):
        sl = slice(int(start[k]), int(start[k + 1]))
        L = int(lens[k])
        hd, hs = capi.dss_densities_host(xyz[0][sl], xyz[1][sl], xyz[2][sl])
        hdr, hsr = capi.dss_densities_host(xyz[0][sl][::-1], xyz[1][sl][::-1], xyz[2][sl][::-1])
        for got, want in ((r["dens_fwd"][sl], hd), (r["sdens_fwd"][sl], hs), (r["dens_rev"][sl], hdr), (r["sdens_rev"][sl], hsr)):
            assert np.array_equal(got == DBL_MAX, want == DBL_MAX), k
            m = want != DBL_MAX
            if m.any():
                worst = max(worst, float(np.abs(got[m] - want[m]).max()))
        nexp += 2 * sum(min(L - 1, p + 50) - max(0, p - 50) for p in range(1, L - 1))      # window terms, chain + reversed copy
    assert nexp > 1e8, nexp
    assert worst < 1e-12, worst                                          # margin 1e-9 >= 1000 x the worst difference
    print("device-vs-libm densities: %.3g exps, max |diff| = %.3g" % (nexp, worst))


def test_rejects_bad_windows(ctx):
    import reseek_amd
    one = np.zeros(4, np.float32)
    with pytest.raises(reseek_amd.RskError):
        ctx.dss_densities([4], one, one, one, W=50, w1=9, w2=8)


@pytest.mark.parametrize("mode", ["sensitive", "verysensitive"])
def test_search_is_the_same_with_and_without_device_densities(ctx, tmpdir, monkeypatch, mode):
    import gzip
    q = os.path.join(str(tmpdir), "q100.bca")
    with gzip.open(os.path.join(fx.GOLDEN, "q100.bca.gz"), "rb") as f, open(q, "wb") as g:
        g.write(f.read())
    outs = []
    for dev, chunk in (("1", None), ("0", None), ("1", "700")):      # device, host only, device in calls of <= 700 residues
        monkeypatch.setenv("RSK_GPU_DENSITY", dev)
        if chunk:
            monkeypatch.setenv("RSK_DSS_CHUNK_RESIDUES", chunk)
        out = os.path.join(str(tmpdir), "hits_%s_%s.tsv" % (dev, chunk))
        ctx.search(q, out, mode=mode, db=q)
        outs.append(sorted(open(out).read().splitlines()))
    assert outs[0] == outs[1] == outs[2] and len(outs[0]) > 100
