// search_api.cpp -- the C-ABI entry points that stand for `reseek -search` (search.cpp:20-111): rsk_search /
// rsk_search_rskdb (SelfSearch, Search_NoMuFilter, the two-stage -fast -db path), the options struct, the hit-line digest,
// rsk_shard_range, rsk_ctx_trim / rsk_shutdown.
#include "host_internal.h"

extern "C" int rsk_shard_range(int kind, const uint32_t *lengths, uint64_t n, uint32_t index, uint32_t count, uint64_t *lo, uint64_t *hi)
{
    if ((n && !lengths) || !lo || !hi || count == 0 || index >= count || kind < 0 || kind > 2) { rsk_set_error("rsk_shard_range: invalid argument"); return RSK_E_INVALID; }
    if (kind == 0) reseek_amd::DBSearcher::SelfShardRange(lengths, n, index, count, *lo, *hi);
    else if (kind == 2) reseek_amd::DBSearcher::SelfWindowRange(lengths, n, index, count, *lo, *hi);
    else reseek_amd::DBSearcher::ResidueShardRange(lengths, n, index, count, *lo, *hi);
    return RSK_OK;
}

extern "C" void rsk_ctx_trim(rsk_ctx *ctx)
{
    if (!ctx) return;
    reseek_amd::SecondaryCtx::Trim(ctx->device);
    reseek_amd::PinnedPool::Shared().Trim();
    rsk_pool_release(ctx);
}

// ---------------------------------------------------------------------------------------------
// C-ABI: `reseek -search Q [-db DB] -fast|-sensitive|-verysensitive -output F [-columns C] [-evalue E]`
// ---------------------------------------------------------------------------------------------
using namespace reseek_amd;

static bool parse_mode(const char *mode, SearchOptions &o)
{
    const std::string m = mode ? mode : "";
    if (m == "fast") o.mode = AM_Fast;
    else if (m == "sensitive") o.mode = AM_Sensitive;
    else if (m == "verysensitive") o.mode = AM_VerySensitive;
    else return false;
    return true;
}

void rsk_set_error(const char *fmt, ...);
namespace reseek_amd {
// rsk_search_opts -> SearchOptions.  Reads no member beyond opts->struct_size (members appended to the struct by later
// headers read as "not given" for a caller built with an older one).
int ParseSearchOpts(const rsk_search_opts *opts, SearchOptions &o, const char *who)
{
    const size_t have = opts->struct_size;
#define RSK_OPT_HAS(f) (have >= offsetof(rsk_search_opts, f) + sizeof(opts->f))
    if (!RSK_OPT_HAS(mode) || have > 4096) {
        rsk_set_error("%s: opts.struct_size = %zu; set it to sizeof(rsk_search_opts) (first member since ABI 4)", who, have);
        return RSK_E_INVALID;
    }
    if (!parse_mode(opts->mode, o)) { rsk_set_error("%s: mode must be fast, sensitive or verysensitive", who); return RSK_E_INVALID; }
    if (RSK_OPT_HAS(columns) && opts->columns) o.columns = opts->columns;
    if (RSK_OPT_HAS(evalue_set) && opts->evalue_set) { o.evalue_set = true; o.evalue = opts->evalue; }
    if (RSK_OPT_HAS(mints_set) && opts->mints_set) { o.mints_set = true; o.mints = opts->mints; }
    if (RSK_OPT_HAS(pvalue_set) && opts->pvalue_set) { o.pvalue_set = true; o.pvalue = opts->pvalue; }
    if (RSK_OPT_HAS(noself)) o.noself = opts->noself != 0;
    if (RSK_OPT_HAS(selfrev0)) o.selfrev0 = opts->selfrev0 != 0;
    if (RSK_OPT_HAS(idx_mode)) {
        if (opts->idx_mode < 0 || opts->idx_mode > 2) { rsk_set_error("%s: idx_mode must be 0, 1 or 2", who); return RSK_E_INVALID; }
        o.idx_mode = opts->idx_mode == 0 ? -1 : opts->idx_mode;
    }
    if (RSK_OPT_HAS(rsb_size) && opts->rsb_size) o.rsb_size = opts->rsb_size;
    if (RSK_OPT_HAS(dbmu) && opts->dbmu) o.dbmu = opts->dbmu;
    if (RSK_OPT_HAS(keeptmp)) o.keeptmp = opts->keeptmp != 0;
    if (RSK_OPT_HAS(shard_index)) o.shard_index = opts->shard_index;
    if (RSK_OPT_HAS(shard_count)) o.shard_count = opts->shard_count;
    if (RSK_OPT_HAS(devices) && opts->devices) o.devices = opts->devices;
    if (RSK_OPT_HAS(hits_digest)) o.hits_digest = opts->hits_digest != 0;
#undef RSK_OPT_HAS
    return RSK_OK;
}
void FastDbOnContexts(const std::vector<rsk_ctx *> &Ctx, const char *query_path, const char *db_path, const SearchOptions &o, const char *out_tsv,
                      const char *tmp_tsv, uint64_t *nhits, uint64_t *stats8);
}

// rsk_search_opts.hits_digest: the hit lines go to a digest instead of a file.  A `-verysensitive` search of 1k queries
// against a PDB-sized DB writes 7e8 lines (30 GB); to compare the union of 8 shards with the unsharded table only an
// order-independent summary is needed: number of lines, their bytes, and the sum and xor of a 64-bit hash of every line.
// The FILE the searchers write to is a glibc cookie stream that cuts the byte stream at newlines (stdio's buffer
// boundaries are arbitrary) and hashes each line; out_tsv then receives ONE line "digest\t<lines>\t<bytes>\t<sum>\t<xor>".
namespace {
struct HitsDigest {
    uint64_t lines = 0, bytes = 0, sum = 0, x = 0;
    std::string carry;
    static uint64_t hash_line(const char *p, size_t n)
    {
        uint64_t h = 0x9E3779B97F4A7C15ull ^ (n * 0xD6E8FEB86659FD93ull);
        auto mix = [&](uint64_t v) { h = (h ^ v) * 0xFF51AFD7ED558CCDull; h ^= h >> 32; };
        for (; n >= 8; p += 8, n -= 8) { uint64_t v; memcpy(&v, p, 8); mix(v); }
        if (n) { uint64_t v = 0; memcpy(&v, p, n); mix(v); }
        h *= 0xC4CEB9FE1A85EC53ull;
        return h ^ (h >> 29);
    }
    void line(const char *p, size_t n) { const uint64_t h = hash_line(p, n); ++lines; bytes += n + 1; sum += h; x ^= h; }
    void feed(const char *p, size_t n)
    {
        const char *end = p + n;
        if (!carry.empty()) {
            const char *nl = (const char *) memchr(p, '\n', n);
            if (!nl) { carry.append(p, n); return; }
            carry.append(p, (size_t) (nl - p));
            line(carry.data(), carry.size());
            carry.clear();
            p = nl + 1;
        }
        while (p < end) {
            const char *nl = (const char *) memchr(p, '\n', (size_t) (end - p));
            if (!nl) { carry.assign(p, (size_t) (end - p)); return; }
            line(p, (size_t) (nl - p));
            p = nl + 1;
        }
    }
    static ssize_t cookie_write(void *c, const char *buf, size_t n) { ((HitsDigest *) c)->feed(buf, n); return (ssize_t) n; }
    FILE *open()
    {
        cookie_io_functions_t io = {};
        io.write = &HitsDigest::cookie_write;
        FILE *f = fopencookie(this, "w", io);
        if (f) setvbuf(f, nullptr, _IOFBF, 8u << 20);
        return f;
    }
};
struct FileCloser { FILE *f; ~FileCloser() { if (f) fclose(f); } };
}   // namespace

static bool keep_tmp_env() { const char *e = getenv("RSK_KEEPTMP"); return e && *e && *e != '0'; }    // -keeptmp

static int search_impl(rsk_ctx *ctx, const char *query_rskdb, const char *db_rskdb, const SearchOptions &o, const char *out_tsv,
                       uint64_t *nhits, uint64_t *stats8)
{
    try {
        DSSParams Params;
        Params.SetDSSParams(o);
        const bool have_db = db_rskdb != nullptr && *db_rskdb;
        const bool prefilter_path = have_db && o.mode == AM_Fast;        // search.cpp:76-111
        SearchOptions o2 = o;
        o2.mode = AM_Sensitive;                  // DM_AlwaysSensitive dssparams.cpp:27-42
        DSSParams Params2;
        Params2.SetDSSParams(o2);
        // option errors are errors of the call on every route below (one device, a device list, a named device)
        {
            DSSAligner Cols;
            Cols.SetColumns(o.columns);
            for (USERFIELD u : Cols.m_UFs)
                if (u == UF_Undefined) { rsk_set_error("rsk_search_rskdb: invalid -columns field"); return RSK_E_INVALID; }
        }
        if (prefilter_path && o.hits_digest) { rsk_set_error("rsk_search: hits_digest is not available on the -fast -db path"); return RSK_E_INVALID; }
        const std::vector<int> devs = DBSearcher::ParseDeviceList(o.devices.empty() ? getenv("RSK_DEVICES") : o.devices.c_str());
        // (the several-device form streams its target shards from a .bca file; any other -db container keeps the one-device
        // two-stage path below, which takes both -- a device list must not make a call fail that works without it)
        const bool db_is_bca = have_db && std::string(db_rskdb).size() >= 4 && std::string(db_rskdb).compare(std::string(db_rskdb).size() - 4, 4, ".bca") == 0;
        if (prefilter_path && devs.size() > 1 && o.shard_count <= 1 && db_is_bca) {
            // the two-stage path on several devices: one target shard per context, the top-B exchange in host memory
            DeviceTeam Team(devs);
            std::vector<rsk_ctx *> cs;
            for (size_t k = 0; k < devs.size(); ++k) cs.push_back(Team.ctx(k));
            const std::string tmp = std::string(out_tsv) + ".prefilter.tmp";
            const bool keep = o.keeptmp || keep_tmp_env();
            FastDbOnContexts(cs, query_rskdb, db_rskdb, o, out_tsv, keep ? tmp.c_str() : nullptr, nhits, stats8);
            return RSK_OK;
        }
        if (devs.size() == 1 && devs[0] != ctx->device) {
            // a one-entry list names THE device of the call: the search runs on a helper context there
            DeviceTeam Team(devs);
            SearchOptions o1 = o;
            o1.devices = std::to_string(devs[0]);
            return search_impl(Team.ctx(0), query_rskdb, db_rskdb, o1, out_tsv, nhits, stats8);      // (o1.devices set: the environment is not consulted again)
        }
        DBSearcher DBS;                       // SelfSearch search.cpp:20-37 / Search_NoMuFilter :39-60
        DBS.m_Params = prefilter_path ? &Params2 : &Params;
        DBS.m_SelfRevQueryFlavour = prefilter_path;      // PostMuFilter computes query self-rev scores itself (postmufilter.cpp:79)
        DBS.m_Opts = o;
        DBS.m_Ctx = ctx;
        if (!o.devices.empty()) DBS.m_Devices = DBSearcher::ParseDeviceList(o.devices.c_str());
        DBS.LoadDB(query_rskdb);
        DBS.Setup();
        if (prefilter_path && o.shard_count > 1) {
            rsk_set_error("rsk_search: shards are not supported on the -fast -db path (the per-query top-B of the prefilter is a reduction over all targets)");
            return RSK_E_INVALID;
        }
        if (prefilter_path) {
            // cmd_search search.cpp:76-111: k-mer prefilter, then the candidates under the "sensitive" preset
            DBSearcher Src;
            Src.m_Params = &Params2;
            Src.m_SelfRevQueryFlavour = true;            // postmufilter.cpp:171
            Src.m_Opts = o;
            Src.m_Ctx = ctx;
            // `-search X -db X`: the two sides are the same file read under the same parameters and the same self-rev
            // flavour (both stages of cmd_search load it with DM_AlwaysSensitive) -- one load, the DB side is a view of it
            if (std::string(db_rskdb) == std::string(query_rskdb)) Src.MakeView(DBS, 0, DBS.GetDBChainCount());
            else Src.LoadDB(db_rskdb);
            // the candidates go from stage to stage in memory, in the hand-off file's order; the file itself
            // (rankedscoresbag.cpp:185-231) is written for -keeptmp only
            const std::string tmp = std::string(out_tsv) + ".prefilter.tmp";
            std::vector<uint32_t> pq, pt;
            MuPreFilterToPairs(DBS, Src, pq, pt, o.keeptmp || keep_tmp_env() ? tmp : std::string());
            if (pq.empty()) fprintf(stderr, "Warning: No hits found by mufilter pass\n");      // postmufilter.cpp:219-223 (no hits file)
            else PostMuFilterPairs(Params2, DBS, Src, pq, pt, out_tsv);
            if (nhits) *nhits = DBS.m_HitCount;
            if (stats8) {
                stats8[0] = DBS.m_ProcessedPairCount; stats8[1] = DBS.m_ProcessedPairCount - DBS.m_MKFPairCount; stats8[2] = DBS.m_MuFilterInputCount;
                stats8[3] = DBS.m_MuFilterDiscardCount; stats8[4] = DBS.m_MKFPairCount; stats8[5] = DBS.m_SWCount;
                stats8[6] = DBS.m_HitCount; stats8[7] = 1;
            }
            return RSK_OK;
        }
        HitsDigest Digest;
        FILE *f = o.hits_digest ? Digest.open() : fopen(out_tsv, "w");
        if (!f) { rsk_set_error("rsk_search_rskdb: cannot create %s", out_tsv); return RSK_E_INVALID; }
        FileCloser closer{ f };                          // closed on every exit path
        DBS.m_fTsv = f;
        if (o.shard_count > 1 && o.shard_index >= o.shard_count) { rsk_set_error("rsk_search: shard_index >= shard_count"); return RSK_E_INVALID; }
        if (!have_db) {
            if (o.shard_count > 1) DBS.RunSelfShard(o.shard_index, o.shard_count);
            else DBS.RunSelf();
        } else {
            const std::string dbfn = db_rskdb;
            if (dbfn.size() >= 4 && dbfn.compare(dbfn.size() - 4, 4, ".bca") == 0) {
                // Search_NoMuFilter search.cpp:39-60: the -db file streams through a ChainReader2
                ChainReader2 CR;
                if (o.shard_count > 1) {
                    // -db mode: contiguous target shards balanced by residues, the query set is replicated
                    BCAData B;
                    B.Open(dbfn);
                    uint64_t Lo, Hi;
                    DBSearcher::ResidueShardRange(B.m_SeqLengths.data(), B.GetChainCount(), o.shard_index, o.shard_count, Lo, Hi);
                    CR.OpenRange(dbfn, Lo, Hi);
                } else
                    CR.Open(dbfn);
                if (const char *e = getenv("RSK_STREAM_CHAINS")) { const long v = atol(e); if (v > 0) DBS.m_StreamBatchChains = (uint) v; }
                DBS.RunQuery(CR);
            } else {
            DBSearcher Src;
            Src.m_Params = &Params;
            Src.m_SelfRevQueryFlavour = true;            // runquery.cpp:43-44
            Src.m_Opts = o;
            Src.m_Ctx = ctx;
            Src.LoadDB(db_rskdb);
            if (o.shard_count > 1) {
                // -db mode: contiguous target shards balanced by residues, the query set is replicated
                const uint NS = Src.GetDBChainCount();
                std::vector<uint32_t> Lens(NS);
                for (uint i = 0; i < NS; ++i) Lens[i] = Src.m_DBChains[i]->GetSeqLength();
                uint64_t Lo, Hi;
                DBSearcher::ResidueShardRange(Lens.data(), NS, o.shard_index, o.shard_count, Lo, Hi);
                DBSearcher View;
                View.MakeView(Src, (uint) Lo, (uint) Hi);
                if (Hi > Lo) DBS.RunQuery(View);
            } else
                DBS.RunQuery(Src);
            }
        }
        closer.f = nullptr;
        if (fclose(f) != 0) { rsk_set_error("rsk_search: writing %s failed", out_tsv); return RSK_E_INVALID; }
        if (o.hits_digest) {
            if (!Digest.carry.empty()) Digest.line(Digest.carry.data(), Digest.carry.size());
            FILE *g = fopen(out_tsv, "w");
            if (!g) { rsk_set_error("rsk_search: cannot create %s", out_tsv); return RSK_E_INVALID; }
            fprintf(g, "digest\t%llu\t%llu\t%016llx\t%016llx\n", (unsigned long long) Digest.lines, (unsigned long long) Digest.bytes,
                    (unsigned long long) Digest.sum, (unsigned long long) Digest.x);
            fclose(g);
        }
        if (nhits) *nhits = DBS.m_HitCount;
        if (stats8) {
            stats8[0] = DBS.m_ProcessedPairCount; stats8[1] = DBS.m_AlnCount; stats8[2] = DBS.m_MuFilterInputCount;
            stats8[3] = DBS.m_MuFilterDiscardCount; stats8[4] = DBS.m_MKFPairCount; stats8[5] = DBS.m_SWCount;
            stats8[6] = DBS.m_HitCount; stats8[7] = 0;
        }
    } catch (const std::exception &e) {
        rsk_set_error("rsk_search_rskdb: %s", e.what());
        return RSK_E_INVALID;
    }
    return RSK_OK;
}

extern "C" int rsk_search_rskdb(rsk_ctx *ctx, const char *query_rskdb, const char *db_rskdb, const char *mode, const char *columns,
                                double evalue, int noself, const char *out_tsv, uint64_t *nhits, uint64_t *stats8)
{
    if (!ctx || !query_rskdb || !out_tsv) { rsk_set_error("rsk_search_rskdb: NULL argument"); return RSK_E_INVALID; }
    SearchOptions o;
    if (!parse_mode(mode, o)) { rsk_set_error("rsk_search_rskdb: mode must be fast, sensitive or verysensitive"); return RSK_E_INVALID; }
    if (columns) o.columns = columns;
    if (evalue >= 0) { o.evalue_set = true; o.evalue = evalue; }
    o.noself = noself != 0;
    return search_impl(ctx, query_rskdb, db_rskdb, o, out_tsv, nhits, stats8);
}

extern "C" int rsk_search(rsk_ctx *ctx, const char *query_path, const char *db_path, const rsk_search_opts *opts, const char *out_tsv,
                          uint64_t *nhits, uint64_t *stats8)
{
    if (!ctx || !query_path || !out_tsv || !opts) { rsk_set_error("rsk_search: NULL argument"); return RSK_E_INVALID; }
    SearchOptions o;
    const int rc = reseek_amd::ParseSearchOpts(opts, o, "rsk_search");
    if (rc != RSK_OK) return rc;
    return search_impl(ctx, query_path, db_path, o, out_tsv, nhits, stats8);
}

// Chains [lo, hi) of a .bca file -- shard `shard_index` of `shard_count`, contiguous, balanced by residues -- featurised (DSS
// profiles, Mu letters, self-rev scores: what LoadDB computes per chain, profileloader.cpp:59, under the call's mode and the
// self-rev flavour the search will use) and written as an RSKDB1 container.  The ranks of a multi-GPU self search from a .bca
// file featurise one slice each and exchange the containers (reseek_amd/dist.py: one all-gather of ~26 bytes per residue)
// instead of every rank featurising every chain.
extern "C" int rsk_bca_to_rskdb(rsk_ctx *ctx, const char *in_bca, uint32_t shard_index, uint32_t shard_count, const rsk_search_opts *opts,
                                int query_flavour, const char *out_rskdb, uint64_t *nchains)
{
    if (!ctx || !in_bca || !opts || !out_rskdb) { rsk_set_error("rsk_bca_to_rskdb: NULL argument"); return RSK_E_INVALID; }
    if (shard_count == 0) shard_count = 1;
    if (shard_index >= shard_count) { rsk_set_error("rsk_bca_to_rskdb: shard_index >= shard_count"); return RSK_E_INVALID; }
    SearchOptions o;
    const int rc = reseek_amd::ParseSearchOpts(opts, o, "rsk_bca_to_rskdb");
    if (rc != RSK_OK) return rc;
    try {
        DSSParams Params;
        Params.SetDSSParams(o);
        BCAData B;
        B.Open(in_bca);
        uint64_t Lo, Hi;
        DBSearcher::ResidueShardRange(B.m_SeqLengths.data(), B.GetChainCount(), shard_index, shard_count, Lo, Hi);
        ChainReader2 CR;
        CR.OpenRange(in_bca, Lo, Hi);
        std::vector<PDBChain *> Chains;
        while (PDBChain *C = CR.GetNext()) Chains.push_back(C);
        DBSearcher S;
        S.m_Params = &Params;
        S.m_Opts = o;
        S.m_Ctx = ctx;
        S.m_SelfRevQueryFlavour = query_flavour != 0;
        try {
            S.LoadChains(Chains);
        } catch (...) {
            for (PDBChain *C : Chains) delete C;
            throw;
        }
        S.WriteRskdb(out_rskdb);
        if (nchains) *nchains = S.GetDBChainCount();
    } catch (const std::exception &e) {
        rsk_set_error("rsk_bca_to_rskdb: %s", e.what());
        return RSK_E_INVALID;
    }
    return RSK_OK;
}

extern "C" int rsk_abi_version(void) { return RSK_ABI_VERSION; }

// ---- rsk_path_counters (the reference's static DSSAligner statistics, dssaligner.h:90-96, for a GPU run) ----------------------
rsk_host_counters g_rsk_counters;

extern "C" int rsk_path_counters_read(rsk_ctx *ctx, rsk_path_counters *out)
{
    if (!ctx || !out || out->struct_size < sizeof(uint32_t)) { rsk_set_error("rsk_path_counters_read: NULL argument / struct_size not set"); return RSK_E_INVALID; }
    rsk_path_counters c;
    memset(&c, 0, sizeof c);
    c.sw_pairs = g_rsk_counters.sw_pairs.load();
    c.sw_pairs_scored = g_rsk_counters.sw_pairs_scored.load();
    c.sw_pairs_rescored = g_rsk_counters.sw_pairs_rescored.load();
    c.upload_copies = g_rsk_counters.upload_copies.load();
    c.upload_bytes = g_rsk_counters.upload_bytes.load();
    c.db_batches = g_rsk_counters.db_batches.load();
    c.loader_seconds = (double) g_rsk_counters.loader_ns.load() * 1e-9;
    c.featurise_seconds = (double) g_rsk_counters.featurise_ns.load() * 1e-9;
    c.upload_seconds = (double) g_rsk_counters.upload_ns.load() * 1e-9;
    if (unsigned long long *w = rsk_swqp_clock_words(ctx->device)) {
        rsk_device_guard g(ctx->device);
        unsigned long long h[2] = { 0, 0 };
        if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(h, w, 16, hipMemcpyDeviceToHost) != hipSuccess) return rsk_hip_fail(hipGetLastError(), "hipMemcpy (k_sw_qp clock words)", __FILE__, __LINE__);
        c.swqp_cycles = h[0]; c.swqp_ref_ticks = h[1];
    }
    const uint32_t want = out->struct_size;
    c.struct_size = (uint32_t) std::min<size_t>(want, sizeof c);
    memcpy(out, &c, c.struct_size);                      // never beyond what the caller's header declares
    return RSK_OK;
}

extern "C" int rsk_path_counters_reset(rsk_ctx *ctx)
{
    if (!ctx) { rsk_set_error("rsk_path_counters_reset: ctx is NULL"); return RSK_E_INVALID; }
    g_rsk_counters.sw_pairs = 0; g_rsk_counters.sw_pairs_scored = 0; g_rsk_counters.sw_pairs_rescored = 0;
    g_rsk_counters.upload_copies = 0; g_rsk_counters.upload_bytes = 0; g_rsk_counters.db_batches = 0;
    g_rsk_counters.loader_ns = 0; g_rsk_counters.featurise_ns = 0; g_rsk_counters.upload_ns = 0;
    if (unsigned long long *w = rsk_swqp_clock_words(ctx->device)) {
        rsk_device_guard g(ctx->device);
        if (hipDeviceSynchronize() != hipSuccess || hipMemset(w, 0, 16) != hipSuccess) return rsk_hip_fail(hipGetLastError(), "hipMemset (k_sw_qp clock words)", __FILE__, __LINE__);
    }
    return RSK_OK;
}

extern "C" void rsk_shutdown(void) { reseek_amd::SecondaryCtx::Trim(-1); reseek_amd::PinnedPool::Shared().Trim(); }

