// rsk_api.hip -- host side of the C-ABI (include/reseek_amd.h): context, HBM chain sets.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <atomic>
#include <chrono>
#include <memory>
#include <cfloat>

#include "rsk_internal.h"

static thread_local char g_err[512] = "";

void rsk_set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int rsk_hip_fail(hipError_t e, const char *what, const char *file, int line)
{
    rsk_set_error("HIP error %d (%s) at %s:%d in %s", (int) e, hipGetErrorString(e), file, line, what);
    return RSK_E_DEVICE;
}

extern "C" const char *rsk_version(void) { return "reseek_amd 0.1 (gfx950)"; }
extern "C" const char *rsk_last_error(void) { return g_err; }

extern "C" int rsk_ctx_create(int device, rsk_ctx **out)
{
    if (!out) { rsk_set_error("rsk_ctx_create: out is NULL"); return RSK_E_INVALID; }
    *out = nullptr;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) {
        rsk_set_error("rsk_ctx_create: no HIP device available (librsk has no CPU fallback)");
        return RSK_E_DEVICE;
    }
    if (device < 0 || device >= ndev) { rsk_set_error("rsk_ctx_create: device %d out of range", device); return RSK_E_INVALID; }
    RSK_HIP(hipSetDevice(device));
    hipDeviceProp_t prop;
    RSK_HIP(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        rsk_set_error("rsk_ctx_create: device %d is %s; librsk is built for gfx950 only", device, prop.gcnArchName);
        return RSK_E_DEVICE;
    }
    std::unique_ptr<rsk_ctx> c(new rsk_ctx);
    c->device = device;
    c->num_cus = prop.multiProcessorCount;
    RSK_HIP(hipEventCreate(&c->ev0));
    if (hipError_t ee = hipEventCreate(&c->ev1)) { (void) hipEventDestroy(c->ev0); return rsk_hip_fail(ee, "hipEventCreate", __FILE__, __LINE__); }
    *out = c.release();
    return RSK_OK;
}

rsk_scratch::~rsk_scratch() { for (void *p : all) rsk_pool_free(ctx, p); }
int rsk_scratch::alloc(void **p, size_t bytes)
{
    const int r = rsk_pool_alloc(ctx, p, std::max<size_t>(bytes, 16));
    if (r == RSK_OK) all.push_back(*p);
    return r;
}

static std::atomic<void (*)(int)> g_oom_hook{nullptr};
void rsk_set_oom_hook(void (*release_idle)(int device)) { g_oom_hook.store(release_idle); }

int rsk_dev_malloc(rsk_ctx *ctx, void **p, size_t bytes)
{
    hipError_t e = hipMalloc(p, bytes);
    if (e == hipSuccess) return RSK_OK;
    (void) hipGetLastError();
    if (ctx) {
        rsk_pool_release(ctx);
        if ((e = hipMalloc(p, bytes)) == hipSuccess) return RSK_OK;
        (void) hipGetLastError();
    }
    if (auto hook = g_oom_hook.load()) {
        int dev = 0;
        (void) hipGetDevice(&dev);
        hook(dev);
        if ((e = hipMalloc(p, bytes)) == hipSuccess) return RSK_OK;
        (void) hipGetLastError();
    }
    *p = nullptr;
    rsk_set_error("out of device memory allocating %zu bytes (%s)", bytes, hipGetErrorString(e));
    return RSK_E_NOMEM;
}

int rsk_db_malloc(const rsk_db *db, rsk_ctx *caller, void **p, size_t bytes)
{
    const int dev = db && db->ctx ? db->ctx->device : 0;
    rsk_device_guard on(dev);
    return rsk_dev_malloc(caller && caller->device == dev ? caller : nullptr, p, bytes);
}

int rsk_pool_alloc(rsk_ctx *ctx, void **p, size_t bytes)
{
    size_t cls = 256;
    while (cls < bytes) cls <<= 1;
    if (cls > (1ull << 30)) {
        // > 1 GiB: steps of 1/8 of the power of two below (256 MiB at least), so that the slightly different scratch sizes
        // of consecutive batches land in one class and reuse one block (a fresh hipMalloc of tens of GB can cost 0.7 s)
        const size_t step = std::max<size_t>(1ull << 28, (cls >> 1) >> 3);
        cls = (bytes + step - 1) / step * step;
    }
    // smallest cached block that holds the request without wasting more than its size again
    auto it = ctx->pool_free.lower_bound(cls);
    if (it != ctx->pool_free.end() && it->first <= 2 * cls) {
        *p = it->second;
        const size_t actual = it->first;             // the block keeps its real size
        ctx->pool_free.erase(it);
        ctx->pool_live[*p] = actual;
        return RSK_OK;
    }
    static const bool trace = getenv("RSK_TRACE") != nullptr;
    const auto t0 = std::chrono::steady_clock::now();
    const int rc = rsk_dev_malloc(ctx, p, cls);      // on failure: drops this context's cached blocks, then the idle helper contexts' pools
    if (trace && cls >= (64u << 20))
        fprintf(stderr, "[pool] hipMalloc %.2f GB (pool of this context %.2f GB) took %.2f ms\n", cls / 1073741824.0, ctx->pool_bytes / 1073741824.0,
                std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    if (rc != RSK_OK) return rc;
    ctx->pool_live[*p] = cls;
    ctx->pool_bytes += cls;
    return RSK_OK;
}

int rsk_stream_wait(rsk_ctx *ctx)
{
    if (!ctx->ev_wait) RSK_HIP(hipEventCreateWithFlags(&ctx->ev_wait, hipEventBlockingSync | hipEventDisableTiming));
    RSK_HIP(hipEventRecord(ctx->ev_wait, ctx->stream));
    RSK_HIP(hipEventSynchronize(ctx->ev_wait));
    return RSK_OK;
}

void rsk_pool_free(rsk_ctx *ctx, void *p)
{
    if (!p) return;
    auto it = ctx->pool_live.find(p);
    if (it == ctx->pool_live.end()) { (void) hipFree(p); return; }
    ctx->pool_free.insert({ it->second, p });
    ctx->pool_live.erase(it);
}

void rsk_pool_release(rsk_ctx *ctx)
{
    for (auto &kv : ctx->pool_free) { (void) hipFree(kv.second); ctx->pool_bytes -= kv.first; }
    ctx->pool_free.clear();
}

int rsk_pinned(rsk_ctx *ctx, int slot, size_t bytes, void **p)
{
    if (ctx->pin_bytes[slot] < bytes) {
        if (ctx->pin[slot]) (void) hipHostFree(ctx->pin[slot]);
        ctx->pin[slot] = nullptr;
        ctx->pin_bytes[slot] = 0;
        size_t cap = 1 << 20;
        while (cap < bytes) cap <<= 1;
        if (hipHostMalloc(&ctx->pin[slot], cap, hipHostMallocPortable) != hipSuccess) {
            rsk_set_error("out of pinned host memory allocating %zu bytes", cap);
            return RSK_E_NOMEM;
        }
        ctx->pin_bytes[slot] = cap;
    }
    *p = ctx->pin[slot];
    return RSK_OK;
}

extern "C" void rsk_ctx_destroy(rsk_ctx *ctx)
{
    if (!ctx) return;
    for (int s = 0; s < 4; ++s) if (ctx->pin[s]) (void) hipHostFree(ctx->pin[s]);
    rsk_pool_release(ctx);
    for (auto &kv : ctx->pool_live) (void) hipFree(kv.first);
    if (ctx->ev0) (void) hipEventDestroy(ctx->ev0);
    if (ctx->ev1) (void) hipEventDestroy(ctx->ev1);
    if (ctx->ev_wait) (void) hipEventDestroy(ctx->ev_wait);
    if (ctx->ev_fork) (void) hipEventDestroy(ctx->ev_fork);
    if (ctx->ev_join) (void) hipEventDestroy(ctx->ev_join);
    if (ctx->aux) (void) hipStreamDestroy(ctx->aux);
    if (ctx->ev_al0) (void) hipEventDestroy(ctx->ev_al0);
    if (ctx->ev_al1) (void) hipEventDestroy(ctx->ev_al1);
    if (ctx->ev_tb) (void) hipEventDestroy(ctx->ev_tb);
    if (ctx->ev_st) (void) hipEventDestroy(ctx->ev_st);
    delete ctx;
}

extern "C" int rsk_ctx_set_stream(rsk_ctx *ctx, void *hip_stream)
{
    if (!ctx) { rsk_set_error("rsk_ctx_set_stream: ctx is NULL"); return RSK_E_INVALID; }
    ctx->stream = (hipStream_t) hip_stream;
    return RSK_OK;
}

extern "C" int rsk_ctx_sync(rsk_ctx *ctx)
{
    if (!ctx) { rsk_set_error("rsk_ctx_sync: ctx is NULL"); return RSK_E_INVALID; }
    RSK_HIP(hipStreamSynchronize(ctx->stream));
    return RSK_OK;
}

extern "C" float rsk_ctx_last_kernel_ms(rsk_ctx *ctx)
{
    if (!ctx || !ctx->ev0) return -1.0f;
    if (hipEventSynchronize(ctx->ev1) != hipSuccess) return -1.0f;
    float ms = -1.0f;
    if (hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1) != hipSuccess) return -1.0f;
    return ms;
}

// cb[r][f] = letter * 4, ra[r][f] = letter * alphabet(f) * 4 from the feature-major profile bytes (padding letters are 0)
__global__ void k_db_derive(const uint8_t *prof, size_t npad, uint16_t *cb, uint16_t *ra)
{
    const size_t r = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= npad) return;
    uint16_t c[8], a[8];
#pragma unroll
    for (int f = 0; f < 8; ++f) {
        const uint32_t l = prof[(size_t) f * npad + r];
        c[f] = (uint16_t) (l * 4);
        a[f] = (uint16_t) (l * (f == 0 ? 20 : 16) * 4);
    }
    uint4 vc, va;
    vc.x = c[0] | ((uint32_t) c[1] << 16); vc.y = c[2] | ((uint32_t) c[3] << 16); vc.z = c[4] | ((uint32_t) c[5] << 16); vc.w = c[6] | ((uint32_t) c[7] << 16);
    va.x = a[0] | ((uint32_t) a[1] << 16); va.y = a[2] | ((uint32_t) a[3] << 16); va.z = a[4] | ((uint32_t) a[5] << 16); va.w = a[6] | ((uint32_t) a[7] << 16);
    ((uint4 *) cb)[r] = vc;
    ((uint4 *) ra)[r] = va;
}

// Host -> device staging of a chain set: the padded SoA arrays are packed by the host threads straight into ONE
// page-locked buffer of the context (slot 2, grow-only, capped) and leave with ONE hipMemcpyAsync per array -- r01-r05 packed into
// pageable vectors and called hipMemcpy on each, which the runtime cut into a staging copy per few MB: ~1,066 copyBuffer
// dispatches per configs[3] run, 8 % of its traced GPU time.  While array k is on the wire the threads pack array k + 1; the
// stream is waited for once at the end (or when the buffer is full: a set larger than the cap goes through it in pieces).
#define RSK_UPLOAD_STAGE_MAX ((size_t) 512 << 20)
struct rsk_uploader {
    rsk_ctx *ctx;
    char *base = nullptr;
    size_t cap = 0, cur = 0;
    uint64_t &hbm;
    rsk_uploader(rsk_ctx *c, uint64_t &hbm_bytes) : ctx(c), hbm(hbm_bytes) {}
    int begin(size_t total)
    {
        void *p = nullptr;
        const int rc = rsk_pinned(ctx, 2, std::min(std::max<size_t>(total, 1), RSK_UPLOAD_STAGE_MAX), &p);
        if (rc != RSK_OK) return rc;
        base = (char *) p; cap = ctx->pin_bytes[2]; cur = 0;
        return RSK_OK;
    }
    int flush() { RSK_HIP(hipStreamSynchronize(ctx->stream)); cur = 0; return RSK_OK; }
    // device array of `bytes` (+ 64 bytes of slack: kernels read whole dwords / strips past the last padded chain) and the piece of
    // the staging buffer that will be copied to it by send()
    int reserve(void **d, size_t bytes, char **h)
    {
        *d = nullptr; *h = nullptr;
        if (bytes == 0) return RSK_OK;
        int rc = rsk_dev_malloc(ctx, d, bytes + 64);
        if (rc != RSK_OK) return rc;
        const size_t need = (bytes + 255) & ~(size_t) 255;
        if (need > cap) {                                   // one array beyond the cap: a buffer of its own size after all
            if ((rc = flush()) != RSK_OK) return rc;
            void *p = nullptr;
            if ((rc = rsk_pinned(ctx, 2, need, &p)) != RSK_OK) return rc;
            base = (char *) p; cap = ctx->pin_bytes[2];
        }
        if (cur + need > cap && (rc = flush()) != RSK_OK) return rc;
        *h = base + cur;
        cur += need;
        return RSK_OK;
    }
    int send(void *d, const char *h, size_t bytes)
    {
        if (bytes == 0) return RSK_OK;
        RSK_HIP(hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, ctx->stream));
        hbm += bytes;
        g_rsk_counters.upload_copies += 1;
        g_rsk_counters.upload_bytes += bytes;
        return RSK_OK;
    }
};

// The chains come through three copy functions (one chain's Mu letters / one feature row / one coordinate axis -> dst[L]): the C-ABI
// entry point copies out of the caller's flat arrays, DBSearcher::UploadToGpu straight out of its per-chain vectors -- either way
// every byte is written once, into the staging buffer (r01-r05: gathered into flat arrays first, then re-packed).
int rsk_db_create_from(rsk_ctx *ctx, uint32_t n, const uint32_t *lengths, const rsk_chain_source &from, const float *selfrev, rsk_db **out)
{
    if (!ctx || !out || (n && !lengths)) { rsk_set_error("rsk_db_create: NULL argument"); return RSK_E_INVALID; }
    *out = nullptr;
    const bool mu = (bool) from.mu, prof = (bool) from.prof, x = (bool) from.xyz;
    RSK_HIP(hipSetDevice(ctx->device));
    static std::atomic<uint64_t> next_uid{1};
    rsk_db *db = new rsk_db;
    std::unique_ptr<rsk_db, void (*)(rsk_db *)> owner(db, rsk_db_destroy);   // every error return below releases the device arrays too
    db->ctx = ctx;
    db->uid = next_uid.fetch_add(1);
    db->n = n;
    db->len.assign(lengths, lengths + n);
    db->off.resize((size_t) n + 1);
    uint64_t o = 0, nres = 0;
    for (uint32_t i = 0; i < n; ++i) {
        if (lengths[i] == 0 || lengths[i] >= 65535) {   // uint16 positions, mukmerfilter.cpp:211
            rsk_set_error("rsk_db_create: chain %u has length %u (must be 1..65534)", i, lengths[i]);
            return RSK_E_RANGE;
        }
        db->off[i] = (uint32_t) o;
        o += (lengths[i] + RSK_CHAIN_PAD - 1) / RSK_CHAIN_PAD * RSK_CHAIN_PAD;
        nres += lengths[i];
        if (o >= 0xFFFF0000ull) { rsk_set_error("rsk_db_create: chain set too large for 32-bit offsets"); return RSK_E_RANGE; }
    }
    db->off[n] = (uint32_t) o;
    db->nres = nres;
    db->npad = o;
    int rc;
    rsk_uploader up(ctx, db->hbm_bytes);
    {
        const size_t total = (size_t) n * 8 + 4 + 256 * 9 + (mu ? (size_t) o + 64 : 0) + (prof ? (size_t) RSK_NFEAT * o : 0) + (x ? (size_t) 12 * o : 0) + (size_t) n * 4;
        if ((rc = up.begin(total)) != RSK_OK) return rc;
    }
    // any error return below must not leave a copy in flight from the staging buffer the next call will overwrite
    struct drain_on_exit { rsk_ctx *c; ~drain_on_exit() { (void) hipStreamSynchronize(c->stream); } } drain{ ctx };
    char *h = nullptr;
    if ((rc = up.reserve((void **) &db->d_len, (size_t) n * 4, &h)) != RSK_OK) return rc;
    if (n) memcpy(h, db->len.data(), (size_t) n * 4);
    if ((rc = up.send(db->d_len, h, (size_t) n * 4)) != RSK_OK) return rc;
    if ((rc = up.reserve((void **) &db->d_off, ((size_t) n + 1) * 4, &h)) != RSK_OK) return rc;
    memcpy(h, db->off.data(), ((size_t) n + 1) * 4);
    if ((rc = up.send(db->d_off, h, ((size_t) n + 1) * 4)) != RSK_OK) return rc;
    // packing into the padded device layout runs on the host threads, chains are independent
    if (mu) {
        db->h_mu.assign((size_t) o + 64, (uint8_t) RSK_MU_NULL);
        std::atomic<uint32_t> bad{UINT32_MAX};
        rsk_parallel_for(n, 512, [&](size_t lo, size_t hi) {
            for (size_t i = lo; i < hi; ++i) {
                uint8_t *row = &db->h_mu[db->off[i]];
                from.mu((uint32_t) i, row);
                uint8_t mx = 0;
                for (uint32_t k = 0; k < lengths[i]; ++k) mx = row[k] > mx ? row[k] : mx;
                if (mx >= RSK_MU_ALPHA) {
                    uint32_t cur = bad.load();
                    while ((uint32_t) i < cur && !bad.compare_exchange_weak(cur, (uint32_t) i)) {}
                }
            }
        });
        if (bad.load() != UINT32_MAX) {
            const uint32_t i = bad.load();
            uint8_t c = 0;
            for (uint32_t k = 0; k < lengths[i] && c < RSK_MU_ALPHA; ++k) c = db->h_mu[db->off[i] + k];
            rsk_set_error("rsk_db_create: Mu letter %u out of range in chain %u", c, i);
            return RSK_E_INVALID;
        }
        // (the host keeps its copy for the ring construction of the gapless kernel; one byte per residue)
        if ((rc = up.reserve((void **) &db->d_mu, db->h_mu.size(), &h)) != RSK_OK) return rc;
        memcpy(h, db->h_mu.data(), db->h_mu.size());
        if ((rc = up.send(db->d_mu, h, db->h_mu.size())) != RSK_OK) return rc;
    }
    if (prof) {
        if ((rc = up.reserve((void **) &db->d_prof, (size_t) RSK_NFEAT * o, &h)) != RSK_OK) return rc;
        uint8_t *const hp = (uint8_t *) h;                            // chains are copied, pads zeroed below
        std::atomic<uint64_t> bad{UINT64_MAX};                        // (chain << 8) | feature of the first offender
        rsk_parallel_for(n, 512, [&](size_t lo, size_t hi) {
            for (size_t i = lo; i < hi; ++i)
                for (int f = 0; f < RSK_NFEAT; ++f) {
                    uint8_t *row = &hp[(size_t) f * o + db->off[i]];
                    from.prof((uint32_t) i, f, row);
                    uint8_t mx = 0;
                    for (uint32_t k = 0; k < lengths[i]; ++k) mx = row[k] > mx ? row[k] : mx;
                    if (mx >= (f == 0 ? 20 : 16)) {
                        const uint64_t key = ((uint64_t) i << 8) | (uint64_t) f;
                        uint64_t cur = bad.load();
                        while (key < cur && !bad.compare_exchange_weak(cur, key)) {}
                    }
                    memset(row + lengths[i], 0, db->off[i + 1] - db->off[i] - lengths[i]);
                }
        });
        if (bad.load() != UINT64_MAX) {
            const uint32_t i = (uint32_t) (bad.load() >> 8);
            const int f = (int) (bad.load() & 255);
            const uint8_t *row = &hp[(size_t) f * o + db->off[i]];
            uint8_t mx = 0;
            for (uint32_t k = 0; k < lengths[i]; ++k) mx = row[k] > mx ? row[k] : mx;
            rsk_set_error("rsk_db_create: profile letter %u out of range (chain %u feature %d)", mx, i, f);
            return RSK_E_INVALID;
        }
        if ((rc = up.send(db->d_prof, h, (size_t) RSK_NFEAT * o)) != RSK_OK) return rc;
        // the float-SW kernels read letter * 4 (column offsets) and letter * alphabet * 4 (row offsets) per feature,
        // residue-major: derived on the device from the bytes just uploaded (same stream: behind the copy)
        const size_t nrec = ((size_t) o + 64) * 8;
        if ((rc = rsk_dev_malloc(ctx, (void **) &db->d_prof_cb, nrec * 2)) != RSK_OK) return rc;
        if ((rc = rsk_dev_malloc(ctx, (void **) &db->d_prof_ra, nrec * 2)) != RSK_OK) return rc;
        db->hbm_bytes += nrec * 4;
        RSK_HIP(hipMemsetAsync(db->d_prof_cb + (size_t) o * 8, 0, 64 * 8 * 2, ctx->stream));
        RSK_HIP(hipMemsetAsync(db->d_prof_ra + (size_t) o * 8, 0, 64 * 8 * 2, ctx->stream));
        if (o) hipLaunchKernelGGL(k_db_derive, dim3((unsigned) ((o + 255) / 256)), dim3(256), 0, ctx->stream, db->d_prof, (size_t) o,
                                  db->d_prof_cb, db->d_prof_ra);
        RSK_HIP(hipGetLastError());
    }
    if (x) {
        float **dsts[3] = { &db->d_x, &db->d_y, &db->d_z };
        for (int ax = 0; ax < 3; ++ax) {
            if ((rc = up.reserve((void **) dsts[ax], (size_t) o * 4, &h)) != RSK_OK) return rc;
            float *const hx = (float *) h;
            rsk_parallel_for(n, 512, [&](size_t lo, size_t hi) {
                for (size_t i = lo; i < hi; ++i) {
                    from.xyz((uint32_t) i, ax, &hx[db->off[i]]);
                    for (uint32_t k = db->off[i] + lengths[i]; k < db->off[i + 1]; ++k) hx[k] = 0.f;
                }
            });
            if ((rc = up.send(*dsts[ax], h, (size_t) o * 4)) != RSK_OK) return rc;
        }
    }
    {
        db->h_selfrev.assign(n, FLT_MAX);
        if (selfrev) db->h_selfrev.assign(selfrev, selfrev + n);
        if ((rc = up.reserve((void **) &db->d_selfrev, (size_t) n * 4, &h)) != RSK_OK) return rc;
        if (n) memcpy(h, db->h_selfrev.data(), (size_t) n * 4);
        if ((rc = up.send(db->d_selfrev, h, (size_t) n * 4)) != RSK_OK) return rc;
    }
    RSK_HIP(hipStreamSynchronize(ctx->stream));             // the staging buffer belongs to the next call from here on
    *out = owner.release();
    return RSK_OK;
}


extern "C" int rsk_db_create(rsk_ctx *ctx, uint32_t n, const uint32_t *lengths, const uint8_t *mu,
                             const uint8_t *prof, const float *x, const float *y, const float *z,
                             const float *selfrev, rsk_db **out)
{
    if (!ctx || !out || (n && !lengths)) { rsk_set_error("rsk_db_create: NULL argument"); return RSK_E_INVALID; }
    *out = nullptr;
    if ((x || y || z) && !(x && y && z)) { rsk_set_error("rsk_db_create: x,y,z must be given together"); return RSK_E_INVALID; }
    std::vector<uint64_t> src((size_t) n + 1, 0);                     // residues before chain i in the caller's flat arrays
    for (uint32_t i = 0; i < n; ++i) src[i + 1] = src[i] + lengths[i];
    rsk_chain_source from;
    if (mu) from.mu = [&](uint32_t i, uint8_t *dst) { memcpy(dst, mu + src[i], lengths[i]); };
    if (prof) from.prof = [&](uint32_t i, int f, uint8_t *dst) { memcpy(dst, prof + (uint64_t) RSK_NFEAT * src[i] + (size_t) f * lengths[i], lengths[i]); };
    if (x) from.xyz = [&](uint32_t i, int ax, float *dst) { memcpy(dst, (ax == 0 ? x : ax == 1 ? y : z) + src[i], 4 * (size_t) lengths[i]); };
    return rsk_db_create_from(ctx, n, lengths, from, selfrev, out);
}

// The self-rev scores of a chain set that was uploaded before they were known (DBSearcher::ComputeSelfRevScores aligns
// the uploaded set against its reversed copy and then completes it): synchronous copy.
int rsk_db_update_selfrev(rsk_db *db, const float *selfrev)
{
    if (!db || !selfrev || !db->d_selfrev) { rsk_set_error("rsk_db_update_selfrev: NULL argument"); return RSK_E_INVALID; }
    db->h_selfrev.assign(selfrev, selfrev + db->n);
    RSK_HIP(hipMemcpy(db->d_selfrev, selfrev, (size_t) db->n * 4, hipMemcpyHostToDevice));
    return RSK_OK;
}

extern "C" int rsk_db_set_seq(rsk_db *db, const char *seq)
{
    if (!db || (db->nres && !seq)) { rsk_set_error("rsk_db_set_seq: NULL argument"); return RSK_E_INVALID; }
    if (db->n == 0) return RSK_OK;
    std::vector<uint8_t> h((size_t) db->npad + 64, 0);
    uint64_t src = 0;
    for (uint32_t i = 0; i < db->n; ++i) { memcpy(&h[db->off[i]], seq + src, db->len[i]); src += db->len[i]; }
    if (!db->d_seq) {
        const int rc = rsk_db_malloc(db, nullptr, (void **) &db->d_seq, h.size());
        if (rc != RSK_OK) return rc;
        db->hbm_bytes += h.size();
    }
    RSK_HIP(hipMemcpy(db->d_seq, h.data(), h.size(), hipMemcpyHostToDevice));
    return RSK_OK;
}

extern "C" void rsk_db_destroy(rsk_db *db)
{
    if (!db) return;
    void *ptrs[] = { db->d_seq, db->d_len, db->d_off, db->d_mu, db->d_prof, db->d_x, db->d_y, db->d_z, db->d_selfrev,
                     db->d_ring_tab, db->d_ring_letters, db->d_ring_laneq, db->d_ring_qid, db->d_ring_perm, db->d_prof_cb, db->d_prof_ra, db->d_pf_table, db->d_pf_postings, db->d_len_perm, db->d_len_rank };
    for (void *p : ptrs)
        if (p) (void) hipFree(p);
    for (auto &kv : db->tri_claims) (void) hipFree(kv.second);
    for (auto &kv : db->nat_claims) (void) hipFree(kv.second);
    for (auto &w : db->work_cache) {
        if (w.d_work) (void) hipFree(w.d_work);
        if (w.d_long_iq) (void) hipFree(w.d_long_iq);
        if (w.d_long_it) (void) hipFree(w.d_long_it);
    }
    delete db;
}

extern "C" uint32_t rsk_db_nchains(const rsk_db *db) { return db ? db->n : 0; }
extern "C" uint64_t rsk_db_nresidues(const rsk_db *db) { return db ? db->nres : 0; }
extern "C" uint64_t rsk_db_hbm_bytes(const rsk_db *db) { return db ? db->hbm_bytes : 0; }

// ---- D1 gapless -----------------------------------------------------------------------------

static int gapless_dense_checks(const char *who, rsk_ctx *ctx, const rsk_db *q, const rsk_db *t, int self_triangle, const uint16_t *d_scores, size_t ldo)
{
    if (!ctx || !q || !t) { rsk_set_error("%s: NULL argument", who); return RSK_E_INVALID; }
    if (!q->d_mu || !t->d_mu) { rsk_set_error("%s: chain set has no Mu letters", who); return RSK_E_INVALID; }
    if (d_scores && ldo < t->n) { rsk_set_error("%s: ldo < number of targets", who); return RSK_E_INVALID; }
    if (self_triangle && q != t) { rsk_set_error("%s: self_triangle needs q == t", who); return RSK_E_INVALID; }
    // uint16 scores: the largest score is 4 * min(LA, LB), so a pair of chains > 16383 could saturate; the pair-list form
    // (int32 scores) has no such limit
    uint32_t mq = 0, mt = 0;
    for (uint32_t L : q->len) mq = std::max(mq, L);
    for (uint32_t L : t->len) mt = std::max(mt, L);
    if (std::min(mq, mt) > 16383) {
        rsk_set_error("%s: chains of %u and %u residues could exceed the uint16 score range (use rsk_mu_gapless_pairs)", who, mq, mt);
        return RSK_E_RANGE;
    }
    return RSK_OK;
}

extern "C" int rsk_mu_gapless_matrix_dev(rsk_ctx *ctx, const rsk_db *q, const rsk_db *t, int self_triangle,
                                         uint16_t *d_scores, size_t ldo)
{
    if (!d_scores) { rsk_set_error("rsk_mu_gapless_matrix_dev: NULL argument"); return RSK_E_INVALID; }
    int rc = gapless_dense_checks("rsk_mu_gapless_matrix_dev", ctx, q, t, self_triangle, d_scores, ldo);
    if (rc != RSK_OK) return rc;
    RSK_HIP(hipSetDevice(ctx->device));
    if (!q->rings_built && (rc = rsk_build_rings(const_cast<rsk_db *>(q))) != RSK_OK) return rc;
    return rsk_launch_gapless_rings(ctx, q, t, self_triangle, d_scores, ldo, 0, 0, 0, nullptr, 0, nullptr);
}

extern "C" int rsk_mu_gapless_hits_dev(rsk_ctx *ctx, const rsk_db *q, const rsk_db *t, int self_triangle, uint16_t *d_scores, size_t ldo,
                                       uint32_t min_score, uint32_t q_base, uint32_t t_base, uint32_t *d_records, uint32_t capacity,
                                       uint32_t *d_count)
{
    if (!d_records || !d_count) { rsk_set_error("rsk_mu_gapless_hits_dev: NULL record buffer / counter"); return RSK_E_INVALID; }
    int rc = gapless_dense_checks("rsk_mu_gapless_hits_dev", ctx, q, t, self_triangle, d_scores, ldo);
    if (rc != RSK_OK) return rc;
    RSK_HIP(hipSetDevice(ctx->device));
    if (!q->rings_built && (rc = rsk_build_rings(const_cast<rsk_db *>(q))) != RSK_OK) return rc;
    RSK_HIP(hipMemsetAsync(d_count, 0, 4, ctx->stream));
    return rsk_launch_gapless_rings(ctx, q, t, self_triangle, d_scores, ldo, min_score, q_base, t_base, d_records, capacity, d_count);
}

extern "C" int rsk_mu_gapless_shard_window(const rsk_db *db, uint32_t shard_index, uint32_t shard_count, uint32_t *pos_lo, uint32_t *pos_hi)
{
    if (!db || !pos_lo || !pos_hi || shard_count == 0 || shard_index >= shard_count) { rsk_set_error("rsk_mu_gapless_shard_window: invalid argument"); return RSK_E_INVALID; }
    int rc;
    if (!db->rings_built) {
        rsk_device_guard g(db->ctx->device);
        if ((rc = rsk_build_rings(const_cast<rsk_db *>(db))) != RSK_OK) return rc;
    }
    // cost of target position p = its letters (in pairs) x the ring slots that walk it (rings whose first member stands at
    // or before p) + its rows against the long chains at or before p: the slots the kernel issues for it
    // Two refinements of the model were measured (r05, per-rank kernel times of the bench set on one GPU) and are off by default:
    // RSK_WINDOW_RAGGED = x charges every ring's ragged first work item (it begins at the ring's first member, in the middle of a
    // target block) as x full 256-target items at that position -- the first of 8 windows holds a third of all ring starts and
    // runs 9 % over the mean; x = 1 brings N = 8 from 0.874 to 0.893 predicted efficiency but N = 2 / 4 from 0.977 / 0.949 to
    // 0.960 / 0.931 (the work moves to the last window, which is the slowest there).  RSK_WINDOW_LONGW = w weighs a row of the
    // per-pair kernel (chains beyond a ring, all in the last window) w times a ring slot: no effect up to w = 8.
    const uint32_t n = db->n;
    std::vector<double> cum((size_t) n + 1, 0.0);
    {
        static const double ragged = getenv("RSK_WINDOW_RAGGED") ? atof(getenv("RSK_WINDOW_RAGGED")) : 0.0;
        static const double longw = getenv("RSK_WINDOW_LONGW") ? atof(getenv("RSK_WINDOW_LONGW")) : 1.0;      // a row of the per-pair kernel (chains beyond a ring) against a ring slot
        std::vector<uint64_t> ring_slots((size_t) n + 1, 0);
        for (const rsk_ring &r : db->rings) ring_slots[r.min_q] += 128ull * r.D;
        const double mean_len = n ? (double) db->nres / (double) n : 0.0;
        uint64_t slots = 0, long_rows = 0;
        const uint32_t first_long = n - (uint32_t) db->long_q.size();
        for (uint32_t p = 0; p < n; ++p) {
            slots += ring_slots[p];
            const uint32_t L = db->len[db->h_ring_perm[p]];
            if (p >= first_long) long_rows += L;
            cum[p + 1] = cum[p] + (double) ((L + 1) / 2 * 2) * ((double) slots + longw * (double) long_rows) + ragged * 256.0 * mean_len * (double) ring_slots[p];
        }
    }
    auto bound = [&](uint32_t r) -> uint32_t {
        if (r == 0) return 0;
        if (r >= shard_count) return n;
        const double want = cum[n] * (double) r / (double) shard_count;
        return (uint32_t) (std::lower_bound(cum.begin(), cum.end(), want) - cum.begin());
    };
    *pos_lo = std::min(bound(shard_index), n);
    *pos_hi = std::max(*pos_lo, std::min(bound(shard_index + 1), n));
    return RSK_OK;
}

extern "C" int rsk_mu_gapless_hits_window_dev(rsk_ctx *ctx, const rsk_db *db, uint32_t pos_lo, uint32_t pos_hi, uint16_t *d_scores, size_t ldo,
                                              uint32_t min_score, uint32_t base, uint32_t *d_records, uint32_t capacity, uint32_t *d_count)
{
    if (!d_records || !d_count) { rsk_set_error("rsk_mu_gapless_hits_window_dev: NULL record buffer / counter"); return RSK_E_INVALID; }
    int rc = gapless_dense_checks("rsk_mu_gapless_hits_window_dev", ctx, db, db, 1, d_scores, ldo);
    if (rc != RSK_OK) return rc;
    if (pos_lo > pos_hi || pos_hi > db->n) { rsk_set_error("rsk_mu_gapless_hits_window_dev: window [%u, %u) outside the set's %u positions", pos_lo, pos_hi, db->n); return RSK_E_INVALID; }
    RSK_HIP(hipSetDevice(ctx->device));
    if (!db->rings_built && (rc = rsk_build_rings(const_cast<rsk_db *>(db))) != RSK_OK) return rc;
    RSK_HIP(hipMemsetAsync(d_count, 0, 4, ctx->stream));
    if (pos_lo == pos_hi) {
        // an empty window launches nothing: record the two timing events back to back so that rsk_ctx_last_kernel_ms reports this
        // call (0 ms), not the launch before it
        ctx->gl_pairs = ctx->gl_cells = ctx->gl_slots = 0; ctx->last_ms = 0.0f;
        RSK_HIP(hipEventRecord(ctx->ev0, ctx->stream));
        RSK_HIP(hipEventRecord(ctx->ev1, ctx->stream));
        return RSK_OK;
    }
    return rsk_launch_gapless_rings(ctx, db, db, 1, d_scores, ldo, min_score, base, base, d_records, capacity, d_count, pos_lo, pos_hi);
}

extern "C" int rsk_mu_gapless_pairs(rsk_ctx *ctx, const rsk_db *q, const rsk_db *t, const uint32_t *iq,
                                    const uint32_t *it, size_t npairs, int32_t *scores, uint32_t *besti,
                                    uint32_t *bestj)
{
    if (!ctx || !q || !t || (npairs && (!iq || !it || !scores))) { rsk_set_error("rsk_mu_gapless_pairs: NULL argument"); return RSK_E_INVALID; }
    if (!q->d_mu || !t->d_mu) { rsk_set_error("rsk_mu_gapless_pairs: chain set has no Mu letters"); return RSK_E_INVALID; }
    for (size_t p = 0; p < npairs; ++p)
        if (iq[p] >= q->n || it[p] >= t->n) { rsk_set_error("rsk_mu_gapless_pairs: pair %zu out of range", p); return RSK_E_INVALID; }
    if (npairs == 0) return RSK_OK;
    RSK_HIP(hipSetDevice(ctx->device));
    rsk_scratch ws(ctx);
    uint32_t *d_iq, *d_it, *d_bi, *d_bj;
    int32_t *d_sc;
    int rc;
    if ((rc = ws.alloc(&d_iq, npairs)) || (rc = ws.alloc(&d_it, npairs)) || (rc = ws.alloc(&d_sc, npairs)) || (rc = ws.alloc(&d_bi, npairs)) ||
        (rc = ws.alloc(&d_bj, npairs)))
        return rc;
    RSK_HIP(hipMemcpyAsync(d_iq, iq, npairs * 4, hipMemcpyHostToDevice, ctx->stream));
    RSK_HIP(hipMemcpyAsync(d_it, it, npairs * 4, hipMemcpyHostToDevice, ctx->stream));
    if ((rc = rsk_launch_gapless_pairs(ctx, q, t, d_iq, d_it, npairs, d_sc, d_bi, d_bj)) != RSK_OK) return rc;
    RSK_HIP(hipMemcpyAsync(scores, d_sc, npairs * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (besti) RSK_HIP(hipMemcpyAsync(besti, d_bi, npairs * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (bestj) RSK_HIP(hipMemcpyAsync(bestj, d_bj, npairs * 4, hipMemcpyDeviceToHost, ctx->stream));
    RSK_HIP(hipStreamSynchronize(ctx->stream));
    return RSK_OK;
}

extern "C" int rsk_mu_gapless_last_work(rsk_ctx *ctx, uint64_t *pairs, uint64_t *cells, uint64_t *cell_slots)
{
    if (!ctx) { rsk_set_error("rsk_mu_gapless_last_work: ctx is NULL"); return RSK_E_INVALID; }
    if (pairs) *pairs = ctx->gl_pairs;
    if (cells) *cells = ctx->gl_cells;
    if (cell_slots) *cell_slots = ctx->gl_slots;
    return RSK_OK;
}
