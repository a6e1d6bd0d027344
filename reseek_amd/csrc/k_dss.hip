// k_dss.hip -- the exp()-heavy part of the DSS featurisation on the device (the per-chain
// inputs of the path): the two density features of dss.cpp,
//   GetDensity      dss.cpp:217-244   D1(Pos)  = sum over |q - Pos| in (w1, W] of exp(-dist(Pos, q) / R)
//   GetSSDensity    dss.cpp:339-372   Dc / (D2 + eps), D2 the same sum over |q - Pos| in (w2, W], Dc its strand ('s') part
// for every position of a batch of chains AND of their reversed copies (GetSelfRevScore alignpair.cpp:7-24 featurises
// the reversed chain).  On the host they are two thirds of the featurisation time (libm exp), which is what a 16-CPU
// container spends most of its cycles on while a .bca database streams past a query batch.
//
// Exactness: distances are the reference's float expression (pdbchain.cpp:310) and the sums run in its order
// (ascending q), in double; the only operation that differs from the host is exp() (device libm, <= 1 ulp, against
// glibc's), so a density differs from the host's by a few 1e-16 relative.  The host (DSS::UseDeviceDensities,
// host/dss.cpp) bins the values and recomputes, with libm, every chain in which a binned quantity lies within 1e-9 of a
// bin boundary -- the letters it emits are therefore the host's, bit for bit.
//
// One thread per residue: the forward-chain position and the reversed-chain position of the same index (independent
// loops over <= 2 W neighbours).  Inputs are L2 resident; bound: double-precision exp throughput (1.1e9 per 32 k chains).
#include <algorithm>
#include <vector>

#include "rsk_internal.h"
#include "host/dss_data.h"

static __device__ __constant__ double c_conf_means[16][9];

struct dss_args {
    const float *x, *y, *z;            // concatenated chains
    uint8_t *ss_fwd, *ss_rev;          // SS characters of the chains / of the reversed chains (index = reversed position): k_dss_local
    uint8_t *conf_fwd, *conf_rev;      // Conf letters (0xFF = none)
    const uint32_t *res_chain;         // chain of each residue
    const uint64_t *off;               // first residue of each chain
    const uint32_t *len;
    uint64_t total;
    int W, w1, w2;
    double radius, eps;
    double *dens_fwd, *sdens_fwd, *dens_rev, *sdens_rev;
    int NW, Nw;                        // nearest-neighbour window / exclusion (dss.h: m_NEN_W, m_NEN_w); NW = 0: not wanted
    uint32_t *nen_fwd, *ren_fwd, *nen_rev, *ren_rev;
};

__device__ __forceinline__ double dss_factor(const float *X, const float *Y, const float *Z, uint32_t a, uint32_t b, double radius)
{
    // (double) PDBChain::GetDist: float dx*dx + dy*dy + dz*dz, float sqrt; then exp(-d / radius) in double
    const float dx = X[a] - X[b], dy = Y[a] - Y[b], dz = Z[a] - Z[b];
    float d2 = dx * dx;
    d2 += dy * dy;
    d2 += dz * dz;
    const float d = sqrtf(d2);
    return exp(-(double) d / radius);
}

// DSS::SetDensities (host/dss.cpp) for position Pos of a chain whose residue i is residue `map(i)` of the stored chain:
// REV = false: map(i) = i; REV = true: map(i) = L - 1 - i (the reversed chain; dist is bitwise symmetric)
template <bool REV>
__device__ __forceinline__ void dss_position(const float *X, const float *Y, const float *Z, const uint8_t *SS, int L, int Pos, int W, int w1, int w2,
                                             double radius, double eps, double &dens, double &sdens)
{
    if (Pos < 1 || Pos + 1 >= L) { dens = 1.7976931348623157e308; sdens = 1.7976931348623157e308; return; }      // DBL_MAX: no value
    auto fac = [&](int q) { return REV ? dss_factor(X, Y, Z, (uint32_t) (L - 1 - Pos), (uint32_t) (L - 1 - q), radius) : dss_factor(X, Y, Z, (uint32_t) Pos, (uint32_t) q, radius); };
    const int lo = max(0, Pos - W), hi = min(L - 1, Pos + W);
    double D1 = 0, D2 = 0, Dc = 0;
    int q = lo;
    for (; q < Pos - w2; ++q) {
        const double F = fac(q);
        D1 += F; D2 += F;
        if (SS[q] == 's') Dc += F;
    }
    for (; q < Pos - w1; ++q) D1 += fac(q);
    for (q = Pos + w1 + 1; q <= min(hi, Pos + w2); ++q) D1 += fac(q);
    for (; q <= hi; ++q) {
        const double F = fac(q);
        D1 += F; D2 += F;
        if (SS[q] == 's') Dc += F;
    }
    dens = D1;
    sdens = Dc / (D2 + eps);
}

// PDBChain::GetSS (getss.cpp:6-60, after TM-align's sec_str) and DSS::ConfLetter (myss.cpp:125-160: nine CA-CA distances
// around Pos, nearest of 16 cluster centres, first wins ties) of position Pos of the chain / the reversed chain.  Float
// distances, double differences, products, sums and square root rounded one by one as on the host: identical letters.
template <bool REV>
__device__ __forceinline__ void dss_local(const float *X, const float *Y, const float *Z, int L, int Pos, uint8_t &ss, uint8_t &conf)
{
    auto dist = [&](int a, int b) {
        const uint32_t s0 = (uint32_t) (REV ? L - 1 - a : a), s1 = (uint32_t) (REV ? L - 1 - b : b);
        const float dx = X[s0] - X[s1], dy = Y[s0] - Y[s1], dz = Z[s0] - Z[s1];
        float d2 = dx * dx;
        d2 += dy * dy;
        d2 += dz * dz;
        return (double) sqrtf(d2);
    };
    if (Pos < 2 || Pos + 2 >= L) ss = '~';
    else {
        const double d13 = dist(Pos - 2, Pos), d14 = dist(Pos - 2, Pos + 1), d15 = dist(Pos - 2, Pos + 2);
        const double d24 = dist(Pos - 1, Pos + 1), d25 = dist(Pos - 1, Pos + 2), d35 = dist(Pos, Pos + 2);
        const double DH = 2.1, DS = 1.42;
        if (fabs(d15 - 6.37) < DH && fabs(d14 - 5.18) < DH && fabs(d25 - 5.18) < DH && fabs(d13 - 5.45) < DH && fabs(d24 - 5.45) < DH &&
            fabs(d35 - 5.45) < DH)
            ss = 'h';
        else if (fabs(d15 - 13) < DS && fabs(d14 - 10.4) < DS && fabs(d25 - 10.4) < DS && fabs(d13 - 6.1) < DS && fabs(d24 - 6.1) < DS &&
                 fabs(d35 - 6.1) < DS)
            ss = 's';
        else if (d15 < 8.2) ss = 't';
        else ss = '~';
    }
    if (Pos < 3 || Pos + 3 >= L) { conf = 0xFF; return; }
    const int iv[9] = { -2, -2, -2, -1, -1, 0, -3, 0, -3 }, jv[9] = { 0, 1, 2, 1, 2, 2, 3, 3, 0 };
    double v[9];
#pragma unroll
    for (int m = 0; m < 9; ++m) v[m] = dist(Pos + iv[m], Pos + jv[m]);
    double MinDist = 1.7976931348623157e308;
    uint32_t Best = 0;
    for (uint32_t k = 0; k < 16; ++k) {
        double Sum2 = 0;
#pragma unroll
        for (int m = 0; m < 9; ++m) { const double diff = v[m] - c_conf_means[k][m]; Sum2 += diff * diff; }
        const double d = sqrt(Sum2);
        if (k == 0 || d < MinDist) { Best = k; MinDist = d; }
    }
    conf = (uint8_t) Best;
}

__global__ __launch_bounds__(256) void k_dss_local(dss_args a)
{
    const uint64_t r = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= a.total) return;
    const uint32_t c = a.res_chain[r];
    const uint64_t o = a.off[c];
    const int L = (int) a.len[c], Pos = (int) (r - o);
    const float *X = a.x + o, *Y = a.y + o, *Z = a.z + o;
    uint8_t ss, cf;
    dss_local<false>(X, Y, Z, L, Pos, ss, cf);
    a.ss_fwd[r] = ss; a.conf_fwd[r] = cf;
    dss_local<true>(X, Y, Z, L, Pos, ss, cf);
    a.ss_rev[r] = ss; a.conf_rev[r] = cf;
}

// DSS::CalcNEN / CalcREN (dss.cpp:374-440): the nearest residue within +-NW positions outside +-Nw (first one in
// ascending position among equals: the reference's running "Dist < MinDist" from 999), and the nearest one on the other
// side of Pos.  Float distances and comparisons only: the device result IS the host's.
template <bool REV>
__device__ __forceinline__ void dss_neighbours(const float *X, const float *Y, const float *Z, int L, int Pos, int NW, int Nw, uint32_t &nen, uint32_t &ren)
{
    auto dist = [&](int q) {
        const uint32_t s0 = (uint32_t) (REV ? L - 1 - Pos : Pos), s1 = (uint32_t) (REV ? L - 1 - q : q);
        const float dx = X[s0] - X[s1], dy = Y[s0] - Y[s1], dz = Z[s0] - Z[s1];
        float d2 = dx * dx;
        d2 += dy * dy;
        d2 += dz * dz;
        return (double) sqrtf(d2);
    };
    auto nearest = [&](int lo, int hi) {
        double MinDist = 999;
        uint32_t MinPos = 0xFFFFFFFFu;
        for (int q = lo; q <= hi; ++q) {
            if (q + Nw >= Pos && q <= Pos + Nw) continue;
            const double d = dist(q);
            if (d < MinDist) { MinDist = d; MinPos = (uint32_t) q; }
        }
        return MinPos;
    };
    const int lo = max(0, Pos - NW), hi = min(L - 1, Pos + NW);
    nen = nearest(lo, hi);
    if (nen == 0xFFFFFFFFu) { ren = 0xFFFFFFFFu; return; }
    ren = (int) nen > Pos ? nearest(lo, Pos - 1) : nearest(Pos + 1, hi);
}

__global__ __launch_bounds__(256) void k_dss_density(dss_args a)
{
    const uint64_t r = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= a.total) return;
    const uint32_t c = a.res_chain[r];
    const uint64_t o = a.off[c];
    const int L = (int) a.len[c], Pos = (int) (r - o);
    const float *X = a.x + o, *Y = a.y + o, *Z = a.z + o;
    double d, s;
    dss_position<false>(X, Y, Z, a.ss_fwd + o, L, Pos, a.W, a.w1, a.w2, a.radius, a.eps, d, s);
    a.dens_fwd[r] = d; a.sdens_fwd[r] = s;
    dss_position<true>(X, Y, Z, a.ss_rev + o, L, Pos, a.W, a.w1, a.w2, a.radius, a.eps, d, s);
    a.dens_rev[r] = d; a.sdens_rev[r] = s;
    if (a.NW > 0) {
        uint32_t ne, re;
        dss_neighbours<false>(X, Y, Z, L, Pos, a.NW, a.Nw, ne, re);
        a.nen_fwd[r] = ne; a.ren_fwd[r] = re;
        dss_neighbours<true>(X, Y, Z, L, Pos, a.NW, a.Nw, ne, re);
        a.nen_rev[r] = ne; a.ren_rev[r] = re;
    }
}

// Per-residue DSS quantities of n chains (concatenated coordinates) and of their reversed copies (index = position in the
// reversed chain): SS characters and Conf letters (0xFF = none), the two densities (DBL_MAX where the reference leaves no
// value) and, with nen_W > 0, the nearest-neighbour positions NEN / REN (UINT_MAX = none).  Host arrays in and out.
extern "C" int rsk_dss_densities(rsk_ctx *ctx, uint32_t n, const uint32_t *len, const float *x, const float *y, const float *z, char *ss_fwd,
                                 char *ss_rev, uint8_t *conf_fwd, uint8_t *conf_rev, int W, int w1, int w2, double radius, double eps,
                                 double *dens_fwd, double *sdens_fwd, double *dens_rev, double *sdens_rev, int nen_W, int nen_w,
                                 uint32_t *nen_fwd, uint32_t *ren_fwd, uint32_t *nen_rev, uint32_t *ren_rev)
{
    if (nen_W > 0 && (nen_w < 0 || !nen_fwd || !ren_fwd || !nen_rev || !ren_rev)) { rsk_set_error("rsk_dss_densities: neighbour outputs missing"); return RSK_E_INVALID; }
    if (!ctx || (n && (!len || !x || !y || !z || !ss_fwd || !ss_rev || !conf_fwd || !conf_rev || !dens_fwd || !sdens_fwd || !dens_rev || !sdens_rev))) {
        rsk_set_error("rsk_dss_densities: NULL argument");
        return RSK_E_INVALID;
    }
    if (W < 1 || w1 < 0 || w2 < w1 || w2 > W || !(radius > 0)) { rsk_set_error("rsk_dss_densities: windows must satisfy 0 <= w1 <= w2 <= W, radius > 0"); return RSK_E_INVALID; }
    if (n == 0) return RSK_OK;
    std::vector<uint64_t> off((size_t) n + 1, 0);
    for (uint32_t i = 0; i < n; ++i) off[i + 1] = off[i] + len[i];
    const uint64_t total = off[n];
    if (total == 0) return RSK_OK;
    std::vector<uint32_t> res_chain(total);
    rsk_parallel_for(n, 1024, [&](size_t lo, size_t hi) {
        for (size_t i = lo; i < hi; ++i) std::fill(res_chain.begin() + off[i], res_chain.begin() + off[i + 1], (uint32_t) i);
    });
    RSK_HIP(hipSetDevice(ctx->device));
    {
        static std::atomic<int> done[64];
        const int trc = rsk_once_per_device(done, ctx->device, [&]() -> int {
            RSK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(c_conf_means), rsk_conf_means, sizeof(rsk_conf_means)));
            return RSK_OK;
        });
        if (trc != RSK_OK) return trc;
    }
    rsk_scratch ws(ctx);
    float *d_x, *d_y, *d_z;
    uint8_t *d_b;                      // ss_fwd, ss_rev, conf_fwd, conf_rev
    uint32_t *d_rc, *d_len;
    uint64_t *d_off;
    double *d_out;
    uint32_t *d_nn = nullptr;
    int rc;
    if (nen_W > 0 && (rc = ws.alloc(&d_nn, 4 * total)) != RSK_OK) return rc;
    if ((rc = ws.alloc(&d_x, total)) || (rc = ws.alloc(&d_y, total)) || (rc = ws.alloc(&d_z, total)) || (rc = ws.alloc(&d_b, 4 * total)) ||
        (rc = ws.alloc(&d_rc, total)) || (rc = ws.alloc(&d_len, (size_t) n)) || (rc = ws.alloc(&d_off, (size_t) n + 1)) ||
        (rc = ws.alloc(&d_out, 4 * total)))
        return rc;
    RSK_HIP(hipMemcpyAsync(d_x, x, total * 4, hipMemcpyHostToDevice, ctx->stream));
    RSK_HIP(hipMemcpyAsync(d_y, y, total * 4, hipMemcpyHostToDevice, ctx->stream));
    RSK_HIP(hipMemcpyAsync(d_z, z, total * 4, hipMemcpyHostToDevice, ctx->stream));
    RSK_HIP(hipMemcpyAsync(d_rc, res_chain.data(), total * 4, hipMemcpyHostToDevice, ctx->stream));
    RSK_HIP(hipMemcpyAsync(d_len, len, (size_t) n * 4, hipMemcpyHostToDevice, ctx->stream));
    RSK_HIP(hipMemcpyAsync(d_off, off.data(), ((size_t) n + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
    dss_args a = {};
    a.x = d_x; a.y = d_y; a.z = d_z; a.res_chain = d_rc; a.off = d_off; a.len = d_len; a.total = total;
    a.ss_fwd = d_b; a.ss_rev = d_b + total; a.conf_fwd = d_b + 2 * total; a.conf_rev = d_b + 3 * total;
    a.W = W; a.w1 = w1; a.w2 = w2; a.radius = radius; a.eps = eps;
    a.dens_fwd = d_out; a.sdens_fwd = d_out + total; a.dens_rev = d_out + 2 * total; a.sdens_rev = d_out + 3 * total;
    a.NW = nen_W > 0 ? nen_W : 0; a.Nw = nen_w;
    if (d_nn) { a.nen_fwd = d_nn; a.ren_fwd = d_nn + total; a.nen_rev = d_nn + 2 * total; a.ren_rev = d_nn + 3 * total; }
    const dim3 grid((unsigned) ((total + 255) / 256));
    hipLaunchKernelGGL(k_dss_local, grid, dim3(256), 0, ctx->stream, a);            // SS of every residue before the densities read their neighbours'
    hipLaunchKernelGGL(k_dss_density, grid, dim3(256), 0, ctx->stream, a);
    RSK_HIP(hipGetLastError());
    RSK_HIP(hipMemcpyAsync(ss_fwd, a.ss_fwd, total, hipMemcpyDeviceToHost, ctx->stream));
    RSK_HIP(hipMemcpyAsync(ss_rev, a.ss_rev, total, hipMemcpyDeviceToHost, ctx->stream));
    RSK_HIP(hipMemcpyAsync(conf_fwd, a.conf_fwd, total, hipMemcpyDeviceToHost, ctx->stream));
    RSK_HIP(hipMemcpyAsync(conf_rev, a.conf_rev, total, hipMemcpyDeviceToHost, ctx->stream));
    RSK_HIP(hipMemcpyAsync(dens_fwd, a.dens_fwd, total * 8, hipMemcpyDeviceToHost, ctx->stream));
    RSK_HIP(hipMemcpyAsync(sdens_fwd, a.sdens_fwd, total * 8, hipMemcpyDeviceToHost, ctx->stream));
    RSK_HIP(hipMemcpyAsync(dens_rev, a.dens_rev, total * 8, hipMemcpyDeviceToHost, ctx->stream));
    RSK_HIP(hipMemcpyAsync(sdens_rev, a.sdens_rev, total * 8, hipMemcpyDeviceToHost, ctx->stream));
    if (d_nn) {
        RSK_HIP(hipMemcpyAsync(nen_fwd, a.nen_fwd, total * 4, hipMemcpyDeviceToHost, ctx->stream));
        RSK_HIP(hipMemcpyAsync(ren_fwd, a.ren_fwd, total * 4, hipMemcpyDeviceToHost, ctx->stream));
        RSK_HIP(hipMemcpyAsync(nen_rev, a.nen_rev, total * 4, hipMemcpyDeviceToHost, ctx->stream));
        RSK_HIP(hipMemcpyAsync(ren_rev, a.ren_rev, total * 4, hipMemcpyDeviceToHost, ctx->stream));
    }
    RSK_HIP(hipStreamSynchronize(ctx->stream));
    return RSK_OK;
}
