/* reseek_amd.h -- C-ABI of librsk.so: the MI355X (gfx950) implementation of reseek's -search hot path.
 *
 * The reference (rcedgar/reseek v2.8) has no plugin/FFI layer; its boundary for this path is the
 * public surface of three C++ classes (DBSearcher dbsearcher.h:14, DSSAligner dssaligner.h:18,
 * MuKmerFilter mukmerfilter.h:10) that work one pair at a time on caller-owned std::vectors.
 * A GPU cannot be fed one pair at a time, so each entry point below is the *batch* form of one
 * reference method: plain pointers + sizes in, plain arrays out, no C++/torch types.  The C++
 * classes in reseek_amd/csrc/host/ keep the reference's names/signatures and forward to these
 * calls (INTEGRATION.md shows the binding a reseek maintainer would add).
 *
 * Conventions
 *   - every function returns RSK_OK (0) or a negative RSK_E_* code; rsk_last_error() gives text.
 *     (The reference Die()s on error, myutils.cpp:785; a library must not exit the host process.)
 *   - "d_" pointers are device (HBM) pointers owned by the caller (e.g. a torch tensor's
 *     data_ptr); all other pointers are host memory.
 *   - work is enqueued on the context's HIP stream (rsk_ctx_set_stream); functions that return
 *     host data synchronise that stream, the *_dev forms do not.
 *   - there is NO CPU fallback: without a usable gfx950 device rsk_ctx_create fails.
 *   - file:line citations are relative to /root/reference/src.
 */
#ifndef RESEEK_AMD_H
#define RESEEK_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RSK_OK 0
#define RSK_E_INVALID (-1)   /* bad argument */
#define RSK_E_DEVICE (-2)    /* HIP error / no gfx950 device */
#define RSK_E_NOMEM (-3)
#define RSK_E_RANGE (-4)     /* chain too long for the requested kernel */

#define RSK_NFEAT 8          /* profile rows per chain: AA,NENDist,Conf,NENConf,RENDist,DstNxtHlx,StrandDens,NormDens
                                (namedparams.cpp:36-43; summation order of SetSMx_NoRev dssaligner.cpp:529) */
#define RSK_MU_ALPHA 36      /* Mu alphabet (dss.cpp:700) */
#define RSK_NO_POS 0xFFFFFFFFu

typedef struct rsk_ctx rsk_ctx;   /* one per process/GPU: device, stream, scratch */
typedef struct rsk_db rsk_db;     /* a chain set resident in HBM as SoA */

/* Bumped whenever a struct layout or a signature of this header changes (INTEGRATION.md lists the breaks):
 * 4 = rsk_search_opts leads with struct_size; rsk_shutdown added.  5 / 6: additive. */
#define RSK_ABI_VERSION 6
int rsk_abi_version(void);        /* the RSK_ABI_VERSION the library was built with */
const char *rsk_version(void);
const char *rsk_last_error(void);

/* ---- context ------------------------------------------------------------------------------- */
int rsk_ctx_create(int device, rsk_ctx **out);
void rsk_ctx_destroy(rsk_ctx *ctx);
/* hipStream_t to enqueue on (NULL = default stream).  Pass torch.cuda.current_stream().cuda_stream
 * so torch events/timers see the kernels. */
int rsk_ctx_set_stream(rsk_ctx *ctx, void *hip_stream);
int rsk_ctx_sync(rsk_ctx *ctx);
/* Device scratch is cached: per context in an allocator pool, and the search drivers keep their helper contexts (own
 * streams, own pools) idle per device between calls.  rsk_ctx_trim returns all of that to the device (hipFree). */
void rsk_ctx_trim(rsk_ctx *ctx);
/* Destroys every parked helper context of every device (their streams and pools).  Optional: call it before the process
 * unloads the HIP runtime if the device memory must be returned earlier than process exit; the library itself makes no
 * HIP call from a static destructor. */
void rsk_shutdown(void);
/* Average duration (ms) of the device work enqueued by the last compute call, measured with HIP
 * events on the context stream; < 0 if none. */
float rsk_ctx_last_kernel_ms(rsk_ctx *ctx);

/* ---- chain set (replaces DBSearcher's in-RAM vectors m_DBChains/m_DBProfiles/m_DBMuLettersVec/
 *      m_DBSelfRevScores, dbsearcher.h:26-33, filled by ProfileLoader::Load profileloader.cpp:73) --
 * lengths[n]; mu = all chains' Mu letters back to back (sum L bytes, values 0..35);
 * prof = per chain 8 rows x L bytes back to back (chain-major, feature-major inside a chain) or NULL;
 * x,y,z = CA coordinates back to back (sum L floats each) or NULL; selfrev[n] or NULL (=> FLT_MAX).
 * The data is copied to HBM (SoA, chains padded); host buffers may be freed afterwards. */
int rsk_db_create(rsk_ctx *ctx, uint32_t nchains, const uint32_t *lengths, const uint8_t *mu,
                  const uint8_t *prof, const float *x, const float *y, const float *z,
                  const float *selfrev, rsk_db **out);
void rsk_db_destroy(rsk_db *db);
uint32_t rsk_db_nchains(const rsk_db *db);
uint64_t rsk_db_nresidues(const rsk_db *db);
uint64_t rsk_db_hbm_bytes(const rsk_db *db);
/* Residue characters of the chains (PDBChain::m_Seq, concatenated in chain order, sum of the lengths bytes; host memory):
 * optional, only the pctid column needs them (rsk_aln.nident). */
int rsk_db_set_seq(rsk_db *db, const char *seq);

/* ---- D1: gapless integer Mu score ------------------------------------------------------------
 * Batch form of SWFastGapless_Int (swgaplessint.cpp:7) == SWFastPinopGapless
 * (swfastpinopgapless.cpp:6): H(i,j) = max(0,H(i-1,j-1)) + IntScoreMx_Mu[a_i][b_j], best cell.
 *
 * Dense block: every query of q against every target of t -> d_scores[iq*ldo + it] (uint16,
 * device memory, ldo >= nt).  With self_triangle != 0 (q == t, all-vs-all) only pairs the
 * reference's RunSelf enumerates (it >= iq, runself.cpp:72-99) are guaranteed to be written;
 * other cells may or may not be.  Asynchronous on the context stream.
 * Scores are exact up to 65535 = 4 * 16383: if BOTH sets hold a chain longer than 16383 residues the call returns RSK_E_RANGE
 * (a pair of such chains could saturate); queries longer than 1022 take a per-pair kernel inside the same call. */
int rsk_mu_gapless_matrix_dev(rsk_ctx *ctx, const rsk_db *q, const rsk_db *t, int self_triangle,
                              uint16_t *d_scores, size_t ldo);
/* The same pass with the search's view of the result: every scored pair whose score is >= min_score is appended to
 * d_records as three uint32 {q_base + query, t_base + target, score} (self_triangle: {min, max} of the two chain ids, every
 * unordered pair at most once); *d_count = hits found (device, reset by the call; records beyond `capacity` are dropped:
 * enlarge and repeat).  d_scores may be NULL: no dense matrix is written at all -- the hit list is then the only HBM
 * write of the pass.  q_base / t_base turn the indices of a shard into indices of the whole set (multi-GPU: the ranks'
 * record buffers are gathered as they are).  Asynchronous on the context stream. */
int rsk_mu_gapless_hits_dev(rsk_ctx *ctx, const rsk_db *q, const rsk_db *t, int self_triangle, uint16_t *d_scores, size_t ldo,
                            uint32_t min_score, uint32_t q_base, uint32_t t_base, uint32_t *d_records, uint32_t capacity,
                            uint32_t *d_count);
/* One rank's share of the self-search triangle (multi-GPU; SURVEY 8e -- the reference deals the pairs of RunSelf to threads
 * through one locked counter, runself.cpp:72-99; here a rank's share is fixed up front).  Every rank keeps the WHOLE set.
 * The set is processed in a fixed order (ring members first, chains too long for a ring last); a pair belongs to the rank
 * whose window of positions [pos_lo, pos_hi) holds the pair's later member.  rsk_mu_gapless_shard_window cuts the positions
 * into shard_count windows of equal modelled cost (the DP slots the kernel issues for them) -- host arithmetic, the same
 * result on every rank; rsk_mu_gapless_hits_window_dev is rsk_mu_gapless_hits_dev(db, db, self_triangle = 1) restricted
 * to a window: ONE launch of the same shape as the whole triangle (the same rings against fewer targets).  The windows of
 * all shards score every unordered pair exactly once; records are {base + min, base + max, score}, the dense matrix (optional)
 * is the n x n matrix of the whole set, of which a rank writes its pairs' cells [min][max]. */
int rsk_mu_gapless_shard_window(const rsk_db *db, uint32_t shard_index, uint32_t shard_count, uint32_t *pos_lo, uint32_t *pos_hi);
int rsk_mu_gapless_hits_window_dev(rsk_ctx *ctx, const rsk_db *db, uint32_t pos_lo, uint32_t pos_hi, uint16_t *d_scores, size_t ldo,
                                   uint32_t min_score, uint32_t base, uint32_t *d_records, uint32_t capacity, uint32_t *d_count);

/* The path's one collective at the C-ABI: all-gather of the ranks' hit records over RCCL / xGMI.  The reference has
 * none: it deals pairs to threads through one locked counter (runself.cpp:72-99) and its threads write hits under a lock
 * (dbsearcher.cpp:98-106); one process per GPU exchanges the hit buffers instead.  Rank 0 calls rsk_comm_unique_id and hands
 * the RSK_COMM_ID_BYTES bytes to the other ranks (a file, MPI, an environment variable: out of band, as with ncclUniqueId);
 * every rank then calls rsk_comm_create(its context, id, rank, world).  rsk_gather_hits: every rank contributes n_local
 * records of rec_bytes bytes from a device buffer of its context's GPU; on return *d_all (device memory owned by the
 * communicator, valid until its next gather or rsk_comm_destroy) holds all ranks' records in rank order on EVERY rank, *n_all
 * their number, counts[world] (optional) the per-rank numbers.  Collective: every rank must call it.  librccl.so is opened at
 * run time; without it these calls return RSK_E_INVALID and everything else works. */
#define RSK_COMM_ID_BYTES 128
typedef struct rsk_comm rsk_comm;
int rsk_comm_unique_id(unsigned char *id /* [RSK_COMM_ID_BYTES] */);
int rsk_comm_create(rsk_ctx *ctx, const unsigned char *id, int rank, int world, rsk_comm **out);
void rsk_comm_destroy(rsk_comm *c);
int rsk_comm_rank(const rsk_comm *c);
int rsk_comm_world(const rsk_comm *c);
int rsk_gather_hits(rsk_comm *c, const void *d_local, uint64_t n_local, uint32_t rec_bytes, void **d_all, uint64_t *n_all, uint64_t *counts);
/* Pair-list form with the position of the first strict maximum in row-major order (Besti/Bestj of
 * SWFastGapless_Int; RSK_NO_POS when the score is 0).  besti/bestj may be NULL.  Synchronous. */
int rsk_mu_gapless_pairs(rsk_ctx *ctx, const rsk_db *q, const rsk_db *t, const uint32_t *iq,
                         const uint32_t *it, size_t npairs, int32_t *scores, uint32_t *besti,
                         uint32_t *bestj);
/* Work accounting for the last rsk_mu_gapless_matrix_dev call: real DP cells (sum LA*LB over the
 * pairs the call is responsible for) and cell slots actually issued by the kernel (>= cells:
 * ring padding, triangle overshoot). */
int rsk_mu_gapless_last_work(rsk_ctx *ctx, uint64_t *pairs, uint64_t *cells, uint64_t *cell_slots);

/* ---- P3/P4: Mu-letter affine SW filter -------------------------------------------------------
 * Batch form of DSSAligner::SetMuQP_Para (parasail_mu.cpp:163, query profiles are built inside the
 * kernel) + AlignMuQP_Para (parasail_mu.cpp:120) + MuFilter (dssaligner.cpp:619).
 *
 * rsk_mu_sw_matrix_dev: the raw score parasail_sw_striped_profile_avx2_256_8 (parasail.cpp:515)
 * returns for query iq (reversed if reverse_query) vs target it: d_scores[iq*ldo + it] in 0..250, or
 * 255 = saturated.  gap_open = cost of the first gap residue, gap_ext = each further one
 * (m_ParaMuGapOpen/Ext = 2/1, dssparams.h:45-46).  Same triangle convention as the gapless call. */
int rsk_mu_sw_matrix_dev(rsk_ctx *ctx, const rsk_db *q, const rsk_db *t, int self_triangle,
                         int reverse_query, int gap_open, int gap_ext, uint8_t *d_scores, size_t ldo);
/* The whole filter on the device: forward scores of every enumerated pair -> d_fwd (as above);
 * pairs with fwd >= omega_fwd (fwd = 777 if saturated) are re-scored against the reversed query;
 * pairs with fwd - rev >= omega (rev = 255 if saturated) are appended, in no particular order, to
 * d_pairs_q/d_pairs_t (+ d_pairs_fwd/d_pairs_rev if not NULL).  *d_npairs (device) receives the
 * number of survivors; entries beyond `capacity` are dropped (check *d_npairs <= capacity).
 * Presets: Fast omega 22 / omega_fwd 50, Sensitive 12 / 20 (dssparams.cpp:52-71). */
int rsk_mu_filter_dev(rsk_ctx *ctx, const rsk_db *q, const rsk_db *t, int self_triangle, int gap_open,
                      int gap_ext, float omega, float omega_fwd, uint8_t *d_fwd, size_t ldo,
                      uint32_t *d_pairs_q, uint32_t *d_pairs_t, int32_t *d_pairs_fwd,
                      int32_t *d_pairs_rev, size_t capacity, uint32_t *d_npairs);
/* One shard of the self-search triangle (r06; the reference deals the pairs of RunSelf to threads through one locked counter,
 * runself.cpp:72-99): the filter over the pairs {a, b} of ONE set whose LONGER member (the later one of equal lengths) stands at a
 * position in [rank_lo, rank_hi) of the set's length order (rsk_len_rank) -- the order the triangle mode walks its targets in, so a
 * shard is one launch of the whole triangle's shape against fewer targets.  Every pair once, survivors as (lower index, higher
 * index); d_fwd is the db->n x ldo matrix of rsk_mu_filter_dev's triangle mode (cells outside the window untouched).  The
 * windows rsk_shard_range(kind 2) returns for shards 0 .. count-1 tile the triangle. */
int rsk_mu_filter_window_dev(rsk_ctx *ctx, const rsk_db *db, uint32_t rank_lo, uint32_t rank_hi, int gap_open, int gap_ext,
                             float omega, float omega_fwd, uint8_t *d_fwd, size_t ldo, uint32_t *d_pairs_q, uint32_t *d_pairs_t,
                             int32_t *d_pairs_fwd, int32_t *d_pairs_rev, size_t capacity, uint32_t *d_npairs);
/* rank[i] = position of chain i in the set's length order: a stable sort by chain length (equal lengths keep the chain order). */
int rsk_len_rank(const rsk_db *db, uint32_t *rank);
/* The survivor list of rsk_mu_filter_dev in a deterministic order: both device columns sorted in place by (d_major[k],
 * d_minor[k]) ascending -- the order the reference walks its pairs in (GetNextPairSelf runself.cpp:72-99,
 * runquery.cpp:82).  major_bound = an exclusive upper bound of the major index (0 = unknown): only its bits are sorted. */
int rsk_pairs_sort_dev(rsk_ctx *ctx, uint32_t *d_major, uint32_t *d_minor, size_t n, uint32_t major_bound);
/* Counters of the last rsk_mu_filter_dev call (m_MuFilterInputCount and the number of pairs that
 * needed the reverse pass, cf. dssaligner.h:90-96). */
int rsk_mu_filter_last_work(rsk_ctx *ctx, uint64_t *pairs, uint64_t *candidates);

/* ---- P5/P6/P7: the main alignment --------------------------------------------------------------
 * Batch form of DSSAligner::Align_NoAccel (dssaligner.cpp:929): SetSMx_NoRev (:529, fused) + SWFast
 * (sw.cpp:79) + TraceBackBitSW (sw.cpp:8) + CalcEvalue (:852; LDDT lddt.cpp:63, StatSig
 * statsig.cpp:27-50).  One record per pair; fields are the reference's result members
 * (dssaligner.h:46-69).  "No alignment" is path_len == 0 / evalue == FLT_MAX, as in the reference
 * (m_Path.empty(), m_EvalueA == FLT_MAX). */
typedef struct rsk_aln {
    float score;            /* m_AlnFwdScore (bit-exact) */
    uint32_t lo_a, lo_b;    /* m_LoA, m_LoB (0-based; RSK_NO_POS if score == 0) */
    uint32_t hi_a, hi_b;    /* m_HiA, m_HiB (RSK_NO_POS when CalcEvalue was skipped) */
    uint32_t ids, gaps;     /* m_Ids (M columns), m_Gaps (D + I) */
    uint32_t path_len;      /* strlen(m_Path) */
    uint64_t path_off;      /* offset of the NUL-terminated path (chars M/D/I) in `paths` */
    float lddt, ts;         /* GetLDDT(), m_NewTestStatisticA (-FLT_MAX if skipped) */
    float pvalue, evalue, qual; /* m_PvalueA, m_EvalueA, m_QualityA (FLT_MAX if skipped) */
    uint32_t nident;        /* M columns with equal residue characters (numerator of GetPctId dssaligner.cpp:1325; the denominator is */
                            /* ids); RSK_NO_POS when the statistics were skipped or a chain set has no sequence (rsk_db_set_seq) */
} rsk_aln;
/* Bytes the `paths` buffer of rsk_align_pairs must hold for this pair list. */
size_t rsk_align_paths_bytes(const rsk_db *a, const rsk_db *b, const uint32_t *ia, const uint32_t *ib,
                             size_t npairs);
/* Aligns chain ia[p] of set a (rows, "A"/query of the reference) with chain ib[p] of set b.
 * gap_open/gap_ext are the reference's m_GapOpen/m_GapExt (<= 0; defaults -0.685533/-0.051881,
 * namedparams.cpp:45-46).  min_fwd_score = m_MinFwdScore (7.0; 0 for -verysensitive): pairs scoring
 * below it keep evalue = FLT_MAX (dssaligner.cpp:861).  Statistics need coordinates and self-rev
 * scores in both chain sets and a non-NULL `paths`.  Synchronous. */
int rsk_align_pairs(rsk_ctx *ctx, const rsk_db *a, const rsk_db *b, const uint32_t *ia, const uint32_t *ib,
                    size_t npairs, float gap_open, float gap_ext, float min_fwd_score, rsk_aln *out,
                    char *paths, size_t paths_bytes);
/* Pairs, DP cells (sum LA*LB) and trace bytes written to HBM by the last rsk_align_pairs call. */
int rsk_align_last_work(rsk_ctx *ctx, uint64_t *pairs, uint64_t *cells, uint64_t *tb_bytes);
/* Kernel times (ms, HIP events on the context's stream) of the stages of the last rsk_align_pairs call: the Smith-Waterman
 * kernels (k_sw_qp / k_sw_float: SWFast sw.cpp:79), the traceback kernel (TraceBackBitSW sw.cpp:8) and the statistics kernels
 * (GetLDDT_mu_fast lddt.cpp:63); -1 where a stage did not run.  Measurement only (bench.py roofline_live). */
int rsk_align_last_times(rsk_ctx *ctx, float *sw_ms, float *traceback_ms, float *stats_ms);

/* ---- D1: the reference's remaining dead-but-named kernels, pair-list form (host arrays) -------------
 * rsk_mu_pinop_pairs:         SWFastPinop swfastpinop.cpp:6 (int32 3-state local DP on IntScoreMx_Mu rows;
 *                             Open/Ext are the reference's negative int8 values, e.g. -2/-1).
 * rsk_mu_gapless_profb_pairs: SWFastGaplessProfb swgaplessprofb.cpp:6 (float gapless score on ScoreMx_Mu rows
 *                             minus the same for the reversed query, DSSAligner::AlignMuQP dssaligner.cpp:1064).
 * rsk_gapless_float_pairs:    SWFastGapless swgapless.cpp:46 on the SetSMx_NoRev matrix of (a[ia], b[ib]);
 *                             besti/bestj (optional) = first best cell in row-major order, RSK_NO_POS if score 0. */
int rsk_mu_pinop_pairs(rsk_ctx *ctx, const rsk_db *q, const rsk_db *t, const uint32_t *iq, const uint32_t *it,
                       size_t npairs, int open, int ext, int32_t *scores);
int rsk_mu_gapless_profb_pairs(rsk_ctx *ctx, const rsk_db *q, const rsk_db *t, const uint32_t *iq,
                               const uint32_t *it, size_t npairs, float *scores);
int rsk_gapless_float_pairs(rsk_ctx *ctx, const rsk_db *a, const rsk_db *b, const uint32_t *ia, const uint32_t *ib,
                            size_t npairs, float *scores, uint32_t *besti, uint32_t *bestj);

/* Pair-list form of the Mu filter: DSSAligner::AlignMuParaBags (parasail_mu.cpp:183) as
 * DSSAligner::AlignBags applies it to each prefilter candidate (chainbag.cpp:68-74).  Host arrays;
 * pass[p] = 1 iff MuScore >= omega, with MuScore = 0 when fwd' < omega_fwd.  fwd/rev (optional) get
 * fwd' (777 if saturated) and the raw reverse score (255 if saturated; 0 when not computed). */
int rsk_mu_filter_pairs(rsk_ctx *ctx, const rsk_db *q, const rsk_db *t, const uint32_t *iq, const uint32_t *it,
                        size_t npairs, int gap_open, int gap_ext, float omega, float omega_fwd, uint8_t *pass,
                        int32_t *fwd, int32_t *rev);

/* ---- P9 (first half): seeding stage of the long-chain MKF path -----------------------------------------
 * MuKmerFilter::SetHashTable (mukmerfilter.cpp:208) + the seed loop of MuKmerFilter::Align (:316-389) with
 * MuXDrop (:105).  found[p] = 1 iff some seed HSP of pair p scores >= min_hsp_score (m_MKF_MinHSPScore, 50);
 * for those pairs a record is appended (any order): rec_pair = p, rec_nkept = number of HSPs the reference
 * keeps (strictly improving score, new Loi, in (PosT, slot) order), rec_kept[r*cap*4 ...] = their
 * (Loi, Loj, Len, Score), at most cap stored (cap <= 1024; the search runs cap = 32 and redoes a pair whose count
 * exceeds it with cap = 1024 -- the keep rule's "new Loi" test only sees the stored entries, so a count > cap is an upper
 * bound, not a result).  *nrecords may exceed max_records (then enlarge and call again).  x1 = m_MKF_X1 (8).
 * Chaining of the kept HSPs: k_mkf_chain inside rsk_mkf_chain_align_pairs (host/dssaligner.cpp keeps the libc-qsort form
 * for pairs whose chain depends on qsort's order of tied end points); the gapped extensions of the found pairs are
 * rsk_mkf_align_pairs / rsk_xdrop_pairs below. */
int rsk_mkf_seed_pairs(rsk_ctx *ctx, const rsk_db *q, const rsk_db *t, const uint32_t *iq, const uint32_t *it,
                       size_t npairs, int x1, int min_hsp_score, uint32_t cap, uint8_t *found, size_t max_records,
                       size_t *nrecords, uint32_t *rec_pair, uint32_t *rec_nkept, int32_t *rec_kept);

/* ---- P9 (second half): the gapped float X-drop building blocks on an explicit LA x LB score matrix (row-major
 * S[a * LB + b], host memory) -- XDropFwd (xdropfwd.cpp:71), XDropBwd (xdropbwd.cpp:28), MergeFwdBwd (mergefwdback.cpp:6),
 * in the form the reference's own self-test drives them (test_xdrop.cpp:81-175).  rsk_xdrop_fwd / rsk_xdrop_bwd run ONE
 * extension on the device, through the same kernel as the search's long-chain batch (k_xdrop_wave with the matrix in
 * place of the profile tables): there is no host implementation of the X-drop DP in this library.
 * XDropFwd extends from (lo_a, lo_b) towards the chain ends (lo < L on both sides), XDropBwd from (hi_a, hi_b) towards the
 * starts; open / ext are added (pass them negative).  path: caller buffer, NUL-terminated, *path_len = its length.
 * rsk_merge_fwd_bwd is plain host code (path concatenation and the Lo / Hi arithmetic). */
int rsk_xdrop_fwd(rsk_ctx *ctx, const float *S, uint32_t LA, uint32_t LB, float X, float open, float ext, uint32_t lo_a, uint32_t lo_b,
                  float *score, char *path, size_t path_cap, uint32_t *path_len);
int rsk_xdrop_bwd(rsk_ctx *ctx, const float *S, uint32_t LA, uint32_t LB, float X, float open, float ext, uint32_t hi_a, uint32_t hi_b,
                  float *score, char *path, size_t path_cap, uint32_t *path_len);
int rsk_merge_fwd_bwd(uint32_t LA, uint32_t LB, uint32_t fwd_lo_a, uint32_t fwd_lo_b, const char *fwd_path, uint32_t bwd_hi_a,
                      uint32_t bwd_hi_b, const char *bwd_path, uint32_t *lo_a, uint32_t *lo_b, uint32_t *hi_a, uint32_t *hi_b,
                      char *path, size_t path_cap, uint32_t *path_len);

/* P9 (second half, device): the two gapped X-drop extensions of XDropHSP (xdrophsp.cpp:97-108) for a list of seeded
 * pairs -- XDropFwd from (lo_a, lo_b) to the chain ends and XDropBwd from (lo_a - 1, lo_b - 1) to the chain starts,
 * with DSSAligner::SubstScore (xdrophsp.cpp:8) as the substitution function; X = m_MKF_X2, gap_open / gap_ext =
 * m_GapOpen / m_GapExt (negative).  1 <= lo < L on both chains.  One wave per extension (k_xdrop_wave).  Host arrays:
 * score_fwd / score_bwd [n]; the path of extension k is paths[*_off[k] .. + *_len[k]) (not NUL-terminated);
 * paths_bytes >= sum(LA + LB + 4). */
int rsk_xdrop_pairs(rsk_ctx *ctx, const rsk_db *a, const rsk_db *b, const uint32_t *ia, const uint32_t *ib,
                    const uint32_t *lo_a, const uint32_t *lo_b, size_t n, float X, float gap_open, float gap_ext,
                    float *score_fwd, float *score_bwd, char *paths, size_t paths_bytes, uint64_t *fwd_off,
                    uint32_t *fwd_len, uint64_t *bwd_off, uint32_t *bwd_len);

/* P9 after the chaining, in ONE device batch (what DBSearcher's long-chain stage uses): per pair the chained seed HSPs
 * (MuKmerFilter::ChainHSPs mukmerfilter.cpp:391; rsk_mkf_chain_align_pairs below chains on the device as well) as a CSR list hsp_first[npairs + 1] / hsp_lo_a / hsp_lo_b / hsp_len ->
 *   PostAlignMKF (dssaligner.cpp:1395: GetMegaHSPScore :488 of every HSP, sum < min_mega_score => no alignment, best HSP),
 *   XDropHSP (xdrophsp.cpp:42: best 8-mer of that HSP = start, XDropFwd + XDropBwd with X = x2, total < 10 => no alignment),
 *   MergeFwdBwd (mergefwdback.cpp:6) and CalcEvalue (dssaligner.cpp:852).
 * out / paths as rsk_align_pairs (path_len == 0 = no alignment).  status[p]: 0 = gated out, 1 = extended on the device,
 * 2 = the start fell outside 1..L-1 of a chain (only possible for chains shorter than 8, where the reference's own extents
 * wrap around): no extension is run, path_len == 0. */
int rsk_mkf_align_pairs(rsk_ctx *ctx, const rsk_db *a, const rsk_db *b, const uint32_t *ia, const uint32_t *ib, size_t npairs,
                        const uint32_t *hsp_first, const int32_t *hsp_lo_a, const int32_t *hsp_lo_b, const int32_t *hsp_len, float x2,
                        float gap_open, float gap_ext, float min_mega_score, float min_fwd_score, rsk_aln *out, uint8_t *status, char *paths,
                        size_t paths_bytes);

/* The same batch starting one step earlier, from the UNCHAINED seed HSPs of every pair (what rsk_mkf_seed_pairs keeps: lo_a, lo_b,
 * len, integer score per HSP; at most 64 per pair): MuKmerFilter::ChainHSPs (mukmerfilter.cpp:391) / Chainer::Chain (chainer.cpp:31)
 * run on the device too, then everything above.  status[p] as above, plus 3 = the reference's chaining of this pair depends
 * on how libc qsort orders two equal interval end points (two HSPs ending at one query position with equal chain scores,
 * chainer.cpp:11-29,121-124): nothing is aligned for it here, the caller chains that pair with qsort itself and passes it to
 * rsk_mkf_align_pairs.  A chain whose total score is <= 0 counts as "no alignment" (status 0, dssaligner.cpp:1397). */
int rsk_mkf_chain_align_pairs(rsk_ctx *ctx, const rsk_db *a, const rsk_db *b, const uint32_t *ia, const uint32_t *ib, size_t npairs,
                              const uint32_t *hsp_first, const int32_t *hsp_lo_a, const int32_t *hsp_lo_b, const int32_t *hsp_len,
                              const int32_t *hsp_score, float x2, float gap_open, float gap_ext, float min_mega_score, float min_fwd_score,
                              rsk_aln *out, uint8_t *status, char *paths, size_t paths_bytes);

/* ---- P10/P11/P12: Mu k-mer prefilter ----------------------------------------------------------------
 * Batch form of MuDex::FromSeqDB (mudex.cpp:386; index of the QUERY set, built once and cached in q)
 * + PrefilterMu::Search over every target (prefiltermu.cpp:382): spaced 5-of-7 k-mers, self-score
 * mask 36, two-hit diagonals, FindHSP diagonal score.  Appends (query, target, score) triples, in
 * no particular order, one per (query, target) with a two-hit diagonal scoring > 0; *d_n = count
 * (entries beyond `capacity` are dropped).
 * neighbourhood: 0 = exact k-mers (`reseek -prefilter_mu`, cmd_prefiltermu.cpp:52);
 *                1 = "idxq" (query k-mers indexed with their >= 36 neighbourhood, exact matches listed
 *                    twice as in mudex.cpp:201-219; `-search -fast -db` with <= 100 queries or -idxq);
 *                2 = "idxt" (neighbourhood of each target k-mer, prefiltermu.cpp:174-199; > 100 queries
 *                    or -idxt);  -1 = the reference's choice by query count (muprefilter.cpp:78-87).
 * At most 65535 queries (uint16 query index in the reference as well). */
int rsk_mu_prefilter_dev(rsk_ctx *ctx, const rsk_db *q, const rsk_db *t, int neighbourhood, uint32_t *d_out_q,
                         uint32_t *d_out_t, uint32_t *d_out_score, size_t capacity, uint32_t *d_n);
/* The same scan for the targets t_lo <= index < t_hi of t only (target indices in the triples stay those of the whole set):
 * a search scans the DB in a few contiguous target ranges so that the host replays the bags of range k (targets ascending:
 * the reference's arrival order) while the device scans range k + 1. */
int rsk_mu_prefilter_range_dev(rsk_ctx *ctx, const rsk_db *q, const rsk_db *t, int neighbourhood, uint32_t t_lo, uint32_t t_hi,
                               uint32_t *d_out_q, uint32_t *d_out_t, uint32_t *d_out_score, size_t capacity, uint32_t *d_n);
/* Work counters of the last rsk_mu_prefilter_dev call (any pointer may be NULL): seed items = (target position, posting)
 * pairs read (prefiltermu.cpp:213-260), postings of the query index, two-hit diagonals found (twohitdiag.cpp:368-398) and
 * the cells their FindHSP scans visited (prefiltermu.cpp:12-48). */
int rsk_mu_prefilter_last_work(rsk_ctx *ctx, uint64_t *seed_items, uint64_t *index_postings, uint64_t *twohit_diagonals, uint64_t *diagonal_cells);
/* RankedScoresBag (rankedscoresbag.cpp:34-51,185-231) on host arrays: per query keep the top
 * rsb_size targets exactly as the reference does with -threads 1 (truncation at 2B, quicksort tie
 * order).  out_* (capacity n) may be NULL; *nout = survivors.  tmp_tsv_path (optional) receives the
 * reference's prefilter hand-off file ("prefilter\t<#targets>", then "TIdx\tK\tQIdx..." lines). */
int rsk_rsb_select(const uint32_t *q, const uint32_t *t, const uint32_t *score, size_t n, uint32_t nqueries,
                   uint32_t rsb_size, uint32_t *out_q, uint32_t *out_t, uint32_t *out_score, size_t *nout,
                   const char *tmp_tsv_path);
/* The same selection from the device-sorted form of the triples: rsk_triples_sort_dev packs the three device columns of
 * rsk_mu_prefilter_dev into keys query << 48 | target << 16 | score and sorts them ascending on the device (d_keys: n
 * uint64) = grouped by query, targets ascending -- the arrival order of RankedScoresBag::AddScore with -threads 1
 * (muprefilter.cpp:21-60); rsk_rsb_select_keys replays the bags from the host copy of those keys (one linear pass per query on
 * the host threads; no grouping or sorting on the host).  Needs query < 65536 (as the prefilter does) and score < 65536. */
int rsk_triples_sort_dev(rsk_ctx *ctx, const uint32_t *d_q, const uint32_t *d_t, const uint32_t *d_score, size_t n, uint64_t *d_keys);
int rsk_rsb_select_keys(const uint64_t *keys, size_t n, uint32_t nqueries, uint32_t rsb_size, uint32_t *out_q, uint32_t *out_t,
                        uint32_t *out_score, size_t *nout, const char *tmp_tsv_path);

/* ---- (f) rows 1-2: per-chain featurisation and the .bca container (host code, no GPU needed) --------
 * rsk_dss_featurize: DSS::GetProfile (dss.cpp:716: AA, NENDist, Conf, NENConf, RENDist, DstNxtHlx, StrandDens,
 *   NormDens -> prof[8][L] feature-major) and DSS::GetMuLetters (dss.cpp:700 -> mu[L]) of one chain given its
 *   amino-acid characters and CA coordinates.  prof or mu may be NULL.
 * rsk_dss_featurize_reversed: the profile of the REVERSED chain (PDBChain::GetReverse pdbchain.cpp:470 + DSS::GetProfile),
 *   i.e. the target side of GetSelfRevScore (alignpair.cpp:7-24), given the un-reversed chain; position p of the result is
 *   residue L-1-p.  Same bytes as rsk_dss_featurize on the reversed arrays, without a second round of exp() calls.
 * rsk_bca_info / rsk_bca_read_chain: BCAData::Open / ReadChain (bcadata.cpp:60,191): coordinates are the
 *   quantised floats IC/10.0f - 1000 (pdbchain.h:90).  *L receives the chain length (also on RSK_E_RANGE). */
int rsk_dss_featurize(const char *seq, const float *x, const float *y, const float *z, uint32_t L, uint8_t *prof,
                      uint8_t *mu);
/* Self-test of the hit-line number formatting ("%.1f", "%.3g" fast paths of DSSAligner::WriteUserField userfields.cpp:45)
 * against snprintf on n pseudo-random values: returns the number of differing strings (0 expected). Host code. */
uint64_t rsk_selftest_format(uint64_t seed, uint64_t n);
/* The two density features of a batch of chains and of their reversed copies on the device: DSS::GetDensity dss.cpp:217-244
 * and DSS::GetSSDensity(Pos, 's') dss.cpp:339-372 (window W, exclusion zones w1 <= w2, exp(-dist / radius), strand density
 * Dc / (D2 + eps)).  Also returned, computed first on the device: the SS strings (PDBChain::GetSS getss.cpp:6-60) and the
 * Conf letters (DSS::ConfLetter myss.cpp:125-160, 0xFF = none) of the chains / of the reversed chains, concatenated like
 * the coordinates -- float / double comparison chains, identical to the host's.  Densities: sum(len) doubles each,
 * DBL_MAX where the reference has no value (chain ends).  The device
 * exp() differs from libm's in the last bit: the host mirror bins these values only where they are further than 1e-9
 * from every bin boundary (DSS::UseDeviceDensities) and recomputes the other chains itself.  nen_W > 0: also
 * DSS::CalcNEN / CalcREN dss.cpp:374-440 (nearest residue within +-nen_W outside +-nen_w, and the nearest on the other
 * side) of every residue, UINT32_MAX = none -- float distances and comparisons only, identical to the host's.
 * Host arrays in and out. */
int rsk_dss_densities(rsk_ctx *ctx, uint32_t n, const uint32_t *len, const float *x, const float *y, const float *z,
                      char *ss_fwd, char *ss_rev, uint8_t *conf_fwd, uint8_t *conf_rev,
                      int W, int w1, int w2, double radius, double eps,
                      double *dens_fwd, double *sdens_fwd, double *dens_rev, double *sdens_rev,
                      int nen_W, int nen_w, uint32_t *nen_fwd, uint32_t *ren_fwd, uint32_t *nen_rev, uint32_t *ren_rev);
/* The same two features of ONE chain as the host computes them (glibc exp): the yardstick of the acceptance margin above. */
int rsk_dss_densities_host(const float *x, const float *y, const float *z, uint32_t L, double *dens, double *sdens);
int rsk_dss_featurize_reversed(const char *seq, const float *x, const float *y, const float *z, uint32_t L, uint8_t *prof);
int rsk_bca_info(const char *path, uint64_t *nchains, uint64_t *nresidues, uint32_t *max_len, uint32_t *max_label);
int rsk_bca_read_chain(const char *path, uint64_t idx, char *label, size_t label_cap, char *seq, float *x, float *y,
                       float *z, uint32_t cap, uint32_t *L);

/* ---- P1/P2/P13: the search drivers ---------------------------------------------------------------
 * `reseek -search Q [-db DB] -fast|-sensitive|-verysensitive -output F [-columns C] [-evalue E] [-noself]`
 * (cmd_search search.cpp:62 -> SelfSearch :20 / Search_NoMuFilter :39), driven by the C++ mirror
 * classes in reseek_amd/csrc/host/ (DBSearcher::LoadDB/Setup/RunSelf/RunQuery/BaseOnAln, DSSAligner).
 * Chain sets are read from .bca files (BCAData + DSS featurisation on the host cores + self-rev scores
 * in one GPU batch, P8) or from ".rskdb" containers (precomputed per-chain DSS profile, Mu letters, CA
 * coordinates, self-rev score), chosen by the file extension.
 * db_rskdb == NULL/"" => self search (all-vs-all, both orientations of every hit are written, as
 * runself.cpp:59-68).  evalue < 0 => the mode's default (10; none for -verysensitive).
 * mode "fast" WITH a db follows search.cpp:76-111: MuPreFilter (k-mer neighbourhood prefilter over the
 * Mu letters, top-1500 per query, hand-off TSV) then PostMuFilter (AlignBags of every candidate under
 * the "sensitive" preset, DM_AlwaysSensitive); the containers must then carry the self-rev scores of
 * the sensitive preset, as PostMuFilter computes them (postmufilter.cpp:79,171).  The hand-off file
 * <out_tsv>.prefilter.tmp is kept when RSK_KEEPTMP=1 (-keeptmp).
 * stats8 (optional, 8 values): pairs, m_AlnCount, m_MuFilterInputCount, m_MuFilterDiscardCount,
 * MKF pairs, full alignments, hits, 0 (1 for the prefilter path, where "pairs" = prefilter candidates). */
int rsk_search_rskdb(rsk_ctx *ctx, const char *query_rskdb, const char *db_rskdb, const char *mode,
                     const char *columns, double evalue, int noself, const char *out_tsv, uint64_t *nhits,
                     uint64_t *stats8);

/* The same driver with the remaining command-line options of `reseek -search` that touch the path
 * (myutils options of search.cpp / dssparams.cpp / dbsearcher.cpp / muprefilter.cpp / postmufilter.cpp).
 * Zero-initialise, set `mode`; fields left 0/NULL mean "option not given". */
typedef struct rsk_search_opts {
    uint32_t struct_size;      /* = sizeof(rsk_search_opts) of the header the CALLER was built with (required, first member  */
                               /* since ABI 4): the library reads no member beyond it, so a caller built against an older    */
                               /* header keeps working when members are appended; 0 or a size that cuts `mode` is rejected.  */
    const char *mode;          /* "fast" | "sensitive" | "verysensitive" */
    const char *columns;       /* -columns */
    double evalue;             /* -evalue; used when evalue_set != 0 */
    int evalue_set;
    double mints;              /* -mints; used when mints_set != 0 */
    int mints_set;
    double pvalue;             /* -pvalue (PostMuFilter accept rule); used when pvalue_set != 0 */
    int pvalue_set;
    int noself;                /* -noself */
    int selfrev0;              /* -selfrev0 */
    int idx_mode;              /* 0 = by query count, 1 = -idxq, 2 = -idxt */
    uint32_t rsb_size;         /* -rsb_size (0 = 1500) */
    const char *dbmu;          /* -dbmu: Mu FASTA of the DB chains for the prefilter stage (search.cpp:93-96) */
    int keeptmp;               /* -keeptmp: keep <out_tsv>.prefilter.tmp */
    uint32_t shard_index;      /* multi-GPU (one process per GPU): this process handles shard shard_index of       */
    uint32_t shard_count;      /* shard_count: -db mode = a contiguous range of DB chains balanced by residues;     */
                               /* self search = the pairs whose longer chain stands in a window of the set's length  */
                               /* order, windows of equal DP cells, + one shard_count-th of the long-chain pair list (no */
                               /* filter: the pairs (i <= j) whose j lies in a range balanced by DP cells).         */
                               /* The union of the shards' hit tables is the unsharded table.  0 or 1 = no shards. */
    const char *devices;       /* multi-GPU (ONE process): "0,1,2,3" = the call drives these devices, one context + host thread */
                               /* each, shards as above, one hits file (an id may repeat: several contexts on one device).  */
                               /* NULL = the environment's RSK_DEVICES if set, else the device of ctx only.                  */
    int hits_digest;           /* != 0: out_tsv receives ONE line "digest\t<lines>\t<bytes>\t<sum>\t<xor>" instead of the hit     */
                               /* table: number of hit lines, their bytes, and the sum / xor (hex) of a 64-bit hash of every line */
                               /* -- order-independent, so the digests of the shards of a search add up (lines, bytes, sum) /    */
                               /* xor together to the digest of the unsharded table.  For tables too large to keep (a            */
                               /* -verysensitive search against a PDB-sized DB writes ~30 GB).  Not on the -fast -db path.       */
} rsk_search_opts;
int rsk_search(rsk_ctx *ctx, const char *query_path, const char *db_path, const rsk_search_opts *opts,
               const char *out_tsv, uint64_t *nhits, uint64_t *stats8);

/* Chains of shard `shard_index` of `shard_count` of a .bca file (contiguous, balanced by residues: rsk_shard_range kind 1),
 * featurised as LoadDB does per chain (DSS profiles + Mu letters dss.cpp:716, self-rev scores alignpair.cpp:7 under opts->mode;
 * query_flavour != 0: the scores a -db search computes for its streamed chains, runquery.cpp:43) and written as an RSKDB1
 * container, the prepared form rsk_search reads without featurising.  The ranks of a multi-GPU self search from a .bca file
 * featurise one slice each and exchange the containers (reseek_amd/dist.py) instead of every rank featurising every chain. */
int rsk_bca_to_rskdb(rsk_ctx *ctx, const char *in_bca, uint32_t shard_index, uint32_t shard_count, const rsk_search_opts *opts,
                     int query_flavour, const char *out_rskdb, uint64_t *nchains);

/* Counters of the search path since the last reset -- process-wide, the counterpart of the static statistics the reference keeps
 * in DSSAligner (dssaligner.h:90-96: m_AlnCount, m_SWCount, m_MuFilterInputCount, ...), extended by what a GPU run needs to be
 * read: how many Smith-Waterman pairs reached the E-value stage, what the streamed -db loader cost, how many host->device copies
 * the chain-set uploads issued, and the clock k_sw_qp actually held (a power-limited kernel runs below the nominal 2.4 GHz).
 * `struct_size` as in rsk_search_opts: the library writes no member beyond it. */
typedef struct rsk_path_counters {
    uint32_t struct_size;
    uint64_t sw_pairs;            /* pairs through SWFast (sw.cpp:79) in rsk_align_pairs calls of the search drivers             */
    uint64_t sw_pairs_scored;     /* ... whose score reached m_MinFwdScore: CalcEvalue ran (dssaligner.cpp:852-861)              */
    uint64_t sw_pairs_rescored;   /* pairs the score-first route sent through the traced kernel (second pass)                   */
    uint64_t upload_copies;       /* host->device copies issued by chain-set uploads (rsk_db_create)                              */
    uint64_t upload_bytes;
    uint64_t db_batches;          /* batches of a streamed -db file (runquery.cpp:82-125 in batches of chains)                   */
    double loader_seconds;        /* loader thread, summed over the batches: read + DSS featurisation + self-rev + upload         */
    double featurise_seconds;     /* ... of which LoadChains (DSS featurisation on the host threads + device densities)           */
    double upload_seconds;        /* ... of which the upload of the batch                                                          */
    uint64_t swqp_cycles;         /* k_sw_qp: shader cycles (s_memtime) summed over its workgroups                                */
    uint64_t swqp_ref_ticks;      /* ... and 100 MHz ticks (s_memrealtime) over the same intervals: clock = cycles / ticks x 0.1 GHz */
} rsk_path_counters;
int rsk_path_counters_read(rsk_ctx *ctx, rsk_path_counters *out);      /* ctx names the device whose k_sw_qp clock words are read */
int rsk_path_counters_reset(rsk_ctx *ctx);

/* The shard bounds the searches use (pure host arithmetic, no device): kind 0 = self search without a Mu filter (or a set beyond
 * one filter pass), targets [lo, hi) of the triangle of pairs (i <= j) such that every shard covers the same number of DP cells;
 * kind 1 = -db search, a contiguous range of chains with the same number of residues per shard; kind 2 (r06) = self search with a
 * Mu filter: a WINDOW [lo, hi) of positions of the set's length order (stable sort by length, rsk_len_rank), equal DP cells of
 * the pairs each position closes with the positions before it.  The shards 0 .. count-1 tile [0, n) in order. */
int rsk_shard_range(int kind, const uint32_t *lengths, uint64_t n, uint32_t index, uint32_t count, uint64_t *lo, uint64_t *hi);

/* ---- `-search -fast -db` on several GPUs: the per-query top-B of the prefilter (RankedScoresBag,
 * rankedscoresbag.cpp:34-51) is a reduction over all targets, so the target-sharded form has one exchange between the
 * two stages of cmd_search (search.cpp:76-111).  One process per GPU:
 *   rsk_fast_shard_open        loads the queries (self-rev under the sensitive preset, postmufilter.cpp:79) and runs MuPreFilter
 *                              (muprefilter.cpp:70) over target range shard_index of shard_count (contiguous, balanced by
 *                              residues);
 *   rsk_fast_shard_triples     -> EVERY (query, GLOBAL target index, score) triple of the shard, arrays owned by the handle;
 *                              the caller all-gathers them over the ranks (reseek_amd/dist.py: RCCL all_gather);
 *   rsk_fast_shard_finish_exact  replays the reference's bags over the union (rsk_rsb_select: truncation at 2B, quicksort tie
 *                              order of rankedscoresbag.cpp:34-51), runs PostMuFilter (postmufilter.cpp:190: AlignBags, Accept,
 *                              ToTsv) on the selected candidates whose target lies in this rank's range and writes this rank's
 *                              hit table; tmp_tsv (optional) receives the hand-off file.  Candidates, hand-off file and the
 *                              union of the ranks' hit tables are the single-GPU ones -- the reference's -- for any shard count.
 * A lighter exchange for very large candidate sets: rsk_fast_shard_candidates (the local top rsb_size per query) +
 * rsk_fast_shard_finish (merge, top rsb_size per query).  Which equal-scoring candidates survive the reference's cut depends
 * on every element its quicksort saw, so this form uses a tie rule of its own -- higher score, then lower target
 * index (rsk_rsb_merge); the result equals the single-GPU table whenever no query has more than rsb_size candidates or the
 * cut is not tied. */
typedef struct rsk_fast_shard rsk_fast_shard;
int rsk_fast_shard_open(rsk_ctx *ctx, const char *query_path, const char *db_path, const rsk_search_opts *opts, rsk_fast_shard **out);
int rsk_fast_shard_triples(rsk_fast_shard *s, const uint32_t **q, const uint32_t **t, const uint32_t **score, size_t *n);
int rsk_fast_shard_finish_exact(rsk_fast_shard *s, const uint32_t *q, const uint32_t *t, const uint32_t *score, size_t n, const char *out_tsv,
                                const char *tmp_tsv, uint64_t *nhits, uint64_t *stats8);
int rsk_fast_shard_candidates(rsk_fast_shard *s, const uint32_t **q, const uint32_t **t, const uint32_t **score, size_t *n);
int rsk_fast_shard_finish(rsk_fast_shard *s, const uint32_t *q, const uint32_t *t, const uint32_t *score, size_t n, const char *out_tsv,
                          const char *tmp_tsv, uint64_t *nhits, uint64_t *stats8);
void rsk_fast_shard_close(rsk_fast_shard *s);
/* Per query the rsb_size best of the given triples (score descending, then target ascending), grouped by query; host code. */
int rsk_rsb_merge(const uint32_t *q, const uint32_t *t, const uint32_t *score, size_t n, uint32_t nqueries, uint32_t rsb_size,
                  uint32_t *out_q, uint32_t *out_t, uint32_t *out_score, size_t *nout);

/* `reseek -convert in.bca -bca out.bca` / `-feature_fasta out.fa` (convert.cpp:262, Mu alphabet): BCAData
 * writer (bcadata.cpp:15-58,140-168: a file it wrote is reproduced byte for byte) and the Mu FASTA of a
 * .bca file ('A' + letter, 80 columns, chains in file order). */
int rsk_bca_copy(const char *in_bca, const char *out_bca);
int rsk_bca_to_mu_fasta(const char *in_bca, const char *out_fasta);

#ifdef __cplusplus
}
#endif
#endif
