// reseek_host.h -- host side above the C-ABI: C++ classes that keep the reference's names, call
// signatures, result fields and -output/-columns semantics for the -search path, but run the
// pair scoring in batches on the GPU through librsk's C-ABI (include/reseek_amd.h).
//
// Mirrors (file:line relative to /root/reference/src):
//   DSSParams   dssparams.h:27, presets dssparams.cpp:44-104, defaults namedparams.cpp:32-53
//   DSSAligner  dssaligner.h:18   (SetQuery/SetTarget/AlignQueryTarget, result fields, ToTsv)
//   MuKmerFilter mukmerfilter.h:10 (seeding + chaining of one long-chain pair; the search seeds in batches on the device)
//   DBSearcher  dbsearcher.h:14   (LoadDB/Setup/RunSelf/RunQuery/BaseOnAln/OnAln)
// The reference reads command-line options through global opt(x) macros (myutils.h:365-372); here
// they are one explicit SearchOptions object.
#pragma once

#include <cfloat>
#include <climits>
#include <cstdint>
#include <cstdio>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <mutex>
#include <string>
#include <vector>

#include "../../../include/reseek_amd.h"

namespace reseek_amd {

// Worker threads for per-chain / per-pair host work: min(hardware threads, this process's cgroup CPU quota, cap);
// RSK_HOST_THREADS overrides.  (Threads beyond the quota only get the whole process throttled.)
unsigned HostThreads(unsigned cap);

struct PhaseTimer {                              // RSK_TRACE=1: wall time of the driver's phases on stderr
    const char *who;
    bool on = getenv("RSK_TRACE") != nullptr;
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    explicit PhaseTimer(const char *w = "RunPairs") : who(w) {}
    void lap(const char *what)
    {
        if (!on) return;
        const auto t1 = std::chrono::steady_clock::now();
        // @: wall clock in ms modulo 100 s, to line up the stages of concurrent threads
        fprintf(stderr, "[%s] %-22s %9.3f ms   @%.1f\n", who, what, std::chrono::duration<double, std::milli>(t1 - t0).count(),
                std::chrono::duration<double, std::milli>(t1.time_since_epoch()).count() - 1e5 * std::floor(std::chrono::duration<double>(t1.time_since_epoch()).count() / 100));
        t0 = t1;
    }
};


typedef unsigned char byte;
typedef unsigned int uint;

enum ALGO_MODE { AM_Invalid, AM_Fast, AM_Sensitive, AM_VerySensitive };   // dssparams.h:8-14

// user output fields, userfieldnames.h
enum USERFIELD {
    UF_Undefined, UF_query, UF_target, UF_pvalue, UF_evalue, UF_qlo, UF_qhi, UF_tlo, UF_thi, UF_ql, UF_tl, UF_pctid,
    UF_cigar, UF_qrow, UF_trow, UF_qrowg, UF_trowg, UF_ts, UF_newts, UF_dpscore, UF_lddt, UF_ids, UF_gaps, UF_aq,
    UF_muhsp, UF_muchain, UF_gscore, UF_raw, UF_muscore, UF_qcovpct, UF_tcovpct
};
USERFIELD StrToUF(const std::string &Str);       // userfields.cpp:11 (returns UF_Undefined instead of Die)

// The options of the reference's -search command that influence the hot path (myopts.h).
struct SearchOptions {
    ALGO_MODE mode = AM_Invalid;        // -fast / -sensitive / -verysensitive
    bool evalue_set = false;  double evalue = 10;      // -evalue
    bool mints_set = false;   double mints = 0;        // -mints
    bool noself = false;                               // -noself
    bool scores_are_not_evalues = false;
    std::string columns;                               // -columns a+b+c ("" = default 10 columns, "std" allowed)
    bool omega_set = false;   float omega = 0;         // -omega
    bool omegafwd_set = false; float omegafwd = 0;     // -omegafwd
    bool minfwdscore_set = false; float minfwdscore = 0;
    bool gapopen_set = false; float gapopen = 0;       // -gapopen (positive)
    float gapext = 0;                                  // applied only when -gapopen is set (dssparams.cpp:96-97, kept)
    bool mkfl_set = false;    uint mkfl = 0;           // -mkfl
    bool selfrev0 = false;                             // -selfrev0
    bool pvalue_set = false;  double pvalue = -1;      // -pvalue (PostMuFilter Accept, postmufilter.cpp:106-115)
    int idx_mode = -1;                                 // -idxq (1) / -idxt (2); -1 = by query count (muprefilter.cpp:78-87)
    uint rsb_size = 1500;                              // -rsb_size (prefiltermuparams.h:15)
    std::string dbmu;                                  // -dbmu: Mu FASTA of the DB for the prefilter stage (search.cpp:93-96)
    std::string db, output;                            // -db, -output (read by the reference-shaped callers: g_Opts below)
    bool fast_set() const { return mode == AM_Fast; }  // optset_fast
    bool keeptmp = false;                              // -keeptmp
    uint shard_index = 0, shard_count = 0;             // multi-GPU: this rank's shard of the targets (0/0 or x/1 = everything)
    bool hits_digest = false;                          // rsk_search_opts.hits_digest: digest line instead of the hit table
    std::string devices;                               // one process, several devices: "0,1,2,3" (DBSearcher::m_Devices); "" = RSK_DEVICES
    size_t batch_pairs = 1u << 20;                     // upper bound of pairs per GPU alignment batch
    // ... and of DP cells per batch (~0.8 trace byte per cell in HBM).  10 G: the trace block stays below the size from which
    // a fresh hipMalloc costs ~30 ms per GB (a 32 GB block: 1 s per context of a cold call), and two stages in flight overlap
    // better than with 24 G batches (1000 x 30,000 -verysensitive: 2.45 s cold and warm against 4.4 / 2.7 s)
    uint64_t batch_cells = 10ull << 30;
};

enum DECIDE_MODE {                                      // dssparams.h:16-25
    DM_Invalid, DM_AlwaysFast, DM_AlwaysSensitive, DM_AlwaysVerysensitive, DM_DefaultFast, DM_DefaultSensitive, DM_UseCommandLineOption
};

// The reference's process-wide state, for callers written like search.cpp: the parsed command line behind its opt(x) /
// optset_x macros (myutils.h:365-372), the first positional argument, and the hits file of output.cpp:8-20.
extern SearchOptions g_Opts;
extern std::string g_Arg1;
extern FILE *g_fTsv;
void OpenOutputFiles();                                 // output.cpp:8  (creates g_Opts.output; "" = no file)
void CloseOutputFiles();                                // output.cpp:15
// A library needs a device context where the reference has none: DBSearcher / DSSAligner objects whose m_Ctx was never
// set use this one (created on first use on device RSK_DEVICE, default 0; destroyed at exit).
rsk_ctx *DefaultCtx();
// An rsk_ctx (stream, events, allocator pool) is not thread-safe.  The reference's callers keep one DSSAligner per thread
// (AlignBags / AlignMKF behind chainbag.cpp:44); here such aligners may share one context, so every batch-of-one call the
// mirror classes make (MuKmerFilter::Align, DSSAligner::XDropHSP / PostAlignMKF) holds this per-context mutex.
std::mutex &CtxMutex(rsk_ctx *ctx);

// A device allocation made by the host layer itself (survivor lists, counters): on the context's device, through the
// library's out-of-memory ladder (cached pool blocks and idle helper contexts are released before it fails), freed on
// every exit path.  Make() also makes the context's device the calling thread's current one, so the caller's plain
// hipMemcpy calls on the buffer are safe in a process that drives several devices.
class DeviceBuffer {
    void *m_Ptr = nullptr;
public:
    DeviceBuffer() = default;
    DeviceBuffer(rsk_ctx *Ctx, size_t Bytes, const char *What) { Make(Ctx, Bytes, What); }
    DeviceBuffer(const DeviceBuffer &) = delete;
    DeviceBuffer &operator=(const DeviceBuffer &) = delete;
    ~DeviceBuffer() { Free(); }
    void Make(rsk_ctx *Ctx, size_t Bytes, const char *What);     // throws std::runtime_error
    void Free();
    template <class T> T *As() const { return (T *) m_Ptr; }
};

class DSSParams {
public:
    float m_GapOpen = FLT_MAX, m_GapExt = FLT_MAX;
    float m_MinFwdScore = FLT_MAX;
    float m_Omega = FLT_MAX, m_OmegaFwd = FLT_MAX;
    bool m_UsePara = true;
    int m_ParaMuGapOpen = 2, m_ParaMuGapExt = 1;       // dssparams.h:45-46
    uint m_MKFL = UINT_MAX;
    int m_MKF_X1 = INT_MAX, m_MKF_X2 = INT_MAX, m_MKF_MinHSPScore = INT_MAX;
    float m_MKF_MinMegaHSPScore = FLT_MAX;
    std::string m_MKFPatternStr = "111";
    void SetDefaults();                                 // namedparams.cpp:32
    void SetDSSParams(const SearchOptions &Opts);       // dssparams.cpp:44 with an explicit command line
    void SetDSSParams(DECIDE_MODE DM);                  // dssparams.cpp:44: the mode from DM / g_Opts, the other options from g_Opts
    std::string m_MuPrefPatternStr = "1110011";         // prefiltermuparams.h (asserted by cmd_search search.cpp:82-83)
};

class PDBChain {                                        // pdbchain.h:10 (the members the path uses)
public:
    std::string m_Label, m_Seq;
    std::vector<float> m_Xs, m_Ys, m_Zs;
    uint m_Idx = UINT_MAX;
    uint GetSeqLength() const { return (uint) m_Seq.size(); }
    float GetDist(uint Pos1, uint Pos2) const;          // pdbchain.cpp:310
    void GetReverse(PDBChain &Rev) const;               // pdbchain.cpp:478
    void GetSS(std::string &SS) const;                  // getss.cpp:33
};

class DSS {                                             // dss.h:15 (Discrete Structure States: per-chain featurisation)
public:
    const PDBChain *m_Chain = nullptr;
    std::vector<double> m_Density_ScaledValues;
    std::vector<double> m_DistFactors;                 // [L][W]: exp(-dist(i, i + k + 1) / radius)
    int m_DistFactorW = 0;
    std::vector<uint> m_NENs, m_RENs, m_ConfLetters;
    std::vector<double> m_DensityValues, m_StrandDensValues;
    std::string m_SS;
    int m_Density_W = 50, m_Density_w = 3, m_SSDensity_W = 50, m_SSDensity_w = 8;     // dss.h:24-37
    double m_Density_Radius = 20.0;
    int m_NEN_W = 100, m_NEN_w = 12;
    double m_DefaultNENDist = 10.0, m_SSDensity_epsilon = 1;
    uint m_SSE_MinLength = 8, m_SSE_Margin = 8;
    std::vector<uint> m_SSE_Mids;
    std::vector<char> m_SSE_cs;
    bool m_SSEsDone = false;
    const DSSParams *m_Params = nullptr;

    void Init(const PDBChain &Chain);
    void SetParams(const DSSParams &Params) { m_Params = &Params; }
    uint GetSeqLength() const { return m_Chain->GetSeqLength(); }
    uint GetFeature(uint FeatureIndex, uint Pos);       // index into the 8 profile features (rsk_feature_name)
    void GetProfile(std::vector<std::vector<byte> > &Profile);                          // dss.cpp:716
    void GetMuLetters(std::vector<byte> &Letters);                                      // dss.cpp:700
    static void GetMuKmers(const std::vector<byte> &MuLetters, std::vector<uint> &Kmers, const std::string &PatternStr);   // dss.cpp:659
    void SetSS();
    void SetNENs();
    void SetSSEs();
    void SetDensity_ScaledValues();
    double GetDensity(uint Pos);
    void SetDistFactors();
    void SetConfLetters();
    void SetDensities();
    void InitReversed(const PDBChain &Rev, DSS &Fwd);   // Init(Rev) + the exp() table mirrored from Fwd (Fwd is on the un-reversed chain)
    // Densities of this chain as rsk_dss_densities (k_dss.hip) computed them: device exp() differs from libm's in the last
    // bit, so they are accepted only if every quantity that gets binned lies further than 1e-9 from all bin boundaries
    // (then the letters are the host's, bit for bit); false = nothing kept, featurise on the host as usual.
    bool UseDeviceDensities(const double *Dens, const double *StrandDens);
    // SS characters and Conf letters (0xFF = none) of every position from the same call: exact on the device
    void UseDeviceLocal(const char *SS, const uint8_t *Conf)
    {
        const uint L = GetSeqLength();
        m_SS.assign(SS, SS + L);
        m_ConfLetters.resize(L);
        for (uint Pos = 0; Pos < L; ++Pos) m_ConfLetters[Pos] = Conf[Pos] == 0xFF ? UINT_MAX : Conf[Pos];
    }
    // NEN / REN of every position as rsk_dss_densities computed them (float comparisons only: identical to SetNENs)
    void UseDeviceNENs(const uint32_t *NEN, const uint32_t *REN) { const uint L = GetSeqLength(); m_NENs.assign(NEN, NEN + L); m_RENs.assign(REN, REN + L); }
    double DistFactor(uint Pos, uint Pos2) const      // Pos != Pos2, |Pos - Pos2| <= window
    {
        return Pos2 > Pos ? m_DistFactors[(size_t) Pos * m_DistFactorW + (Pos2 - Pos - 1)] : m_DistFactors[(size_t) Pos2 * m_DistFactorW + (Pos - Pos2 - 1)];
    }
    double GetSSDensity(uint Pos, char c);
    double GetFloat_DstNxtHlx(uint Pos);
    uint ConfLetter(uint Pos) const;
};

const uint32_t BCA_MAGIC = 0xBCABCA;                    // bcadata.h:36
class BCAData {                                         // bcadata.h:8 (reader side)
public:
    FILE *m_f = nullptr;
    std::vector<std::string> m_Labels;
    std::vector<uint64_t> m_Offsets;
    std::vector<uint32_t> m_SeqLengths;
    std::mutex m_ReadLock;
    ~BCAData() { Close(); }
    bool m_Writing = false;
    void Open(const std::string &FN);
    void Close();
    uint64_t GetChainCount() const { return m_SeqLengths.size(); }
    void ReadChain(uint64_t ChainIdx, PDBChain &Chain);
    void Create(const std::string &FN);                 // bcadata.cpp:15
    void WriteChain(const PDBChain &Chain);             // bcadata.cpp:34
};
// Mu FASTA: letters are written as 'A' + letter (pdbchain.cpp:70) and read back through g_CharToLetterMu
// (alpha.cpp:3291: 'L' -> 10, 'K' -> 11), exactly as the reference does.
void ReadMuFasta(const std::string &FN, std::vector<std::string> &Labels, std::vector<std::vector<byte> > &Seqs);

// chainreader2.h:10 -- streams the chains of a structure file.  Only the .bca container is on this path (PDB / mmCIF /
// .cal parsing is out of scope: SURVEY section 2); Open() throws for anything else.  GetNext() returns a new PDBChain
// the caller deletes, 0 at the end (runquery.cpp:33-35); thread-safe like the reference's (m_CRGlobalLock).
class ChainReader2 {
public:
    BCAData m_BCA;
    uint64_t m_ChainIdx_BCA = 0, m_EndIdx_BCA = 0;
    std::string m_CurrentFN;
    std::mutex m_CRGlobalLock;
    void Open(const std::string &FileName);
    void OpenRange(const std::string &FileName, uint64_t Lo, uint64_t Hi);   // chains [Lo, Hi) only (multi-GPU target shards)
    PDBChain *GetNext();
    uint64_t GetChainCount() const { return m_EndIdx_BCA; }
};

class ChainBag {                                        // chainbag.h:5 -- borrowed pointers to one chain's search inputs
public:
    const PDBChain *m_ptrChain = nullptr;
    const std::vector<std::vector<byte> > *m_ptrProfile = nullptr;
    const std::vector<byte> *m_ptrMuLetters = nullptr;
    const std::vector<uint> *m_ptrMuKmers = nullptr;
    const void *m_ptrProfPara = nullptr, *m_ptrProfParaRev = nullptr;   // parasail profiles: unused here (the Mu kernel builds its own in LDS)
    const uint16_t *m_ptrKmerHashTableQ = nullptr;
    float m_SelfRevScore = FLT_MAX;
};

class DSSAligner;

class MuKmerFilter {                                    // mukmerfilter.h:10
public:
    const DSSParams *m_Params = nullptr;
    rsk_ctx *m_Ctx = nullptr;                           // the owning DSSAligner's context (null: DefaultCtx())
    const std::vector<byte> *m_ptrMuLettersQ = nullptr;
    const std::vector<uint> *m_ptrMuKmersQ = nullptr;
    const std::vector<byte> *m_ptrMuLettersT = nullptr;
    std::vector<int> m_MuKmerHSPLois, m_MuKmerHSPLojs, m_MuKmerHSPLens, m_MuKmerHSPScores;
    int m_BestChainScore = 0, m_BestHSPScore = 0;
    std::vector<int> m_ChainHSPLois, m_ChainHSPLojs, m_ChainHSPLens;

    void SetParams(const DSSParams &Params) { m_Params = &Params; }
    void ResetQ();
    void SetQ(const std::string &LabelQ, const std::vector<byte> *ptrMuLettersQ, const std::vector<uint> *ptrMuKmersQ);
    void Align(const std::vector<byte> &MuLettersT, const std::vector<uint> &MuKmersT);
    void SetBagQ(const ChainBag &BagQ);                 // mukmerfilter.h:82 (mukmerfilter2.cpp): SetQ from a bag
    void AlignBag(const ChainBag &BagT);                // mukmerfilter.h:87: Align against a bag's letters / k-mers
    // the state Align() leaves when its seed loop kept these HSPs (computed by rsk_mkf_seed_pairs), then ChainHSPs()
    void SetSeedHSPs(const int32_t *Kept4, uint Count);
    void ChainHSPs();
};

class DSSAligner {                                      // dssaligner.h:18
    const DSSParams *m_Params = nullptr;

public:
    const PDBChain *m_ChainA = nullptr, *m_ChainB = nullptr;
    const std::vector<std::vector<byte> > *m_ProfileA = nullptr, *m_ProfileB = nullptr;
    const std::vector<byte> *m_MuLettersA = nullptr, *m_MuLettersB = nullptr;
    const std::vector<uint> *m_MuKmersA = nullptr, *m_MuKmersB = nullptr;
    MuKmerFilter m_MKF;
    float m_XDropScore = 0;
    std::string m_XDropPath;

    std::string m_Path;
    uint m_LoA = UINT_MAX, m_LoB = UINT_MAX, m_HiA = UINT_MAX, m_HiB = UINT_MAX;
    float m_PvalueA = FLT_MAX, m_PvalueB = FLT_MAX, m_EvalueA = FLT_MAX, m_EvalueB = FLT_MAX;
    float m_QualityA = FLT_MAX, m_QualityB = FLT_MAX;
    float m_TestStatisticA = -FLT_MAX, m_TestStatisticB = -FLT_MAX;
    float m_NewTestStatisticA = -FLT_MAX, m_NewTestStatisticB = -FLT_MAX;
    uint m_Ids = UINT_MAX, m_Gaps = UINT_MAX;
    uint m_IdentCount = UINT_MAX;                       // M columns with equal residues, counted on the device (UINT_MAX: GetPctId walks the path)
    float m_SelfRevScoreA = FLT_MAX, m_SelfRevScoreB = FLT_MAX;
    float m_AlnFwdScore = 0;
    float m_LDDT = FLT_MAX;
    std::vector<USERFIELD> m_UFs;
    rsk_ctx *m_Ctx = nullptr;                           // GPU context for the single-pair form of AlignQueryTarget

    static std::mutex m_OutputLock;

    DSSAligner();
    void SetColumns(const std::string &Columns);        // -columns (dssaligner.cpp:114-136); "" = default_columns :100-112
    void SetParams(const DSSParams &Params);
    const DSSParams &GetParams() const { return *m_Params; }
    void UnsetQuery();
    void SetQuery(const PDBChain &Chain, const std::vector<std::vector<byte> > *ptrProfile, const std::vector<byte> *ptrMuLetters,
                  const std::vector<uint> *ptrMuKmers, float SelfRevScore);
    void SetTarget(const PDBChain &Chain, const std::vector<std::vector<byte> > *ptrProfile, const std::vector<byte> *ptrMuLetters,
                   const std::vector<uint> *ptrMuKmers, float SelfRevScore);
    bool DoMKF() const;                                 // dssaligner.cpp:715
    void ClearAlign();                                  // dssaligner.cpp:906
    void AlignQueryTarget();                            // dssaligner.cpp:793 (one pair; batch of 1 on the GPU)
    bool DoMKF_Bags(const ChainBag &BagA, const ChainBag &BagB) const;   // chainbag.cpp:6
    void AlignBags(const ChainBag &BagA, const ChainBag &BagB);          // chainbag.cpp:44 (one pair; batch of 1 on the GPU)
    void AlignBagsMKF(const ChainBag &BagA, const ChainBag &BagB);       // chainbag.cpp:24
    void AlignPairOnGpu();                              // MuFilter (:619) + Align_NoAccel (:929) + CalcEvalue of the current pair, batch of one
    void AlignMKF();                                    // dssaligner.cpp:1387 (seeds + chain here, the rest a device batch of one)
    void AlignMKF_FromSeeds(const int32_t *Kept4, uint Count);   // same, seeding stage already done on the GPU
    void PostAlignMKF();                                // dssaligner.cpp:1395 (rsk_mkf_align_pairs, one pair)
    float GetMegaHSPScore(uint Lo_i, uint Lo_j, uint Len);   // dssaligner.cpp:488
    float SubstScore(uint PosA, uint PosB);             // xdrophsp.cpp:8
    float XDropHSP(uint Loi_in, uint Loj_in, uint Len, uint &Loi_out, uint &Loj_out, uint &Hii_out, uint &Hij_out);
    void CalcEvalue();                                  // dssaligner.cpp:852 (host form of the statistics the batches compute on the device)
    float GetLDDT() const;                              // dssaligner.cpp:1313
    void AppendTsv(std::string &out, bool Up);               // the hit line of ToTsv, appended to a caller buffer
    void AppendUserField(std::string &out, USERFIELD UF, bool Up);
    void SetFromAln(const rsk_aln &Aln, const char *Path);   // fill the result fields from a GPU batch record

    void ToTsv(FILE *f, bool Up, bool NoSelf = false);  // dssaligner.cpp:1016
    void WriteUserField(FILE *f, USERFIELD UF, bool Up);
    const char *GetLabel(bool Top) const { return Top ? m_ChainA->m_Label.c_str() : m_ChainB->m_Label.c_str(); }
    uint GetQL(bool Top) const { return Top ? m_ChainA->GetSeqLength() : m_ChainB->GetSeqLength(); }
    uint GetTL(bool Top) const { return Top ? m_ChainB->GetSeqLength() : m_ChainA->GetSeqLength(); }
    uint GetLo(bool Top) const { return Top ? m_LoA : m_LoB; }
    uint GetHi(bool Top) const { return Top ? m_HiA : m_HiB; }
    float GetTestStatistic(bool Top) const { return Top ? m_TestStatisticA : m_TestStatisticB; }
    float GetNewTestStatistic(bool Top) const { return Top ? m_NewTestStatisticA : m_NewTestStatisticB; }
    float GetEvalue(bool Top) const { return Top ? m_EvalueA : m_EvalueB; }
    float GetPvalue(bool Top) const { return Top ? m_PvalueA : m_PvalueB; }
    float GetAQ(bool Top) const { return Top ? m_QualityA : m_QualityB; }
    double GetQCovPct(bool Top) const;
    double GetTCovPct(bool Top) const;
    float GetPctId() const;                             // dssaligner.cpp:1325
    void GetRow(bool Up, bool Top, bool Global, std::string &Row) const;
    void GetRow_A(std::string &Row, bool Global) const;
    void GetRow_B(std::string &Row, bool Global) const;
};

void InvertPath(const std::string &Path, std::string &InvPath);              // dssaligner.cpp:58
void GetPathCounts(const std::string &Path, uint &M, uint &D, uint &I);      // dssaligner.cpp:75
void PathToCIGAR(const char *Path, std::string &CIGAR, bool FlipDI);         // cigar.cpp:95

class RankedScoresBag {                                 // rankedscoresbag.h:16
public:
    uint m_B = 1500;                                    // RSB_SIZE prefiltermuparams.h:15 (-rsb_size)
    std::vector<std::vector<uint16_t> > m_QueryIdxToScoreVec;
    std::vector<std::vector<uint> > m_QueryIdxToTargetIdxVec;
    std::vector<uint16_t> m_QueryIdxToLoScore;
    uint m_QueryCount = UINT_MAX;
    void Init(uint QueryCount);
    void TruncateVecs(uint QIdx);
    void AddScore(uint QueryIdx, uint TargetIdx, uint16_t Score);
    void Finish();                                      // the final TruncateVecs pass of ToTsv
    void ToTsv(FILE *fTsv);                             // "prefilter\t<#targets>" + "TIdx\tK\tQIdx..." lines
    // not in the reference: the selection target-major (the hand-off order) without the file
    void GroupByTarget(std::vector<size_t> &First, std::vector<uint> &Queries, uint &TargetCount);
    void ToPairs(std::vector<uint32_t> &pq, std::vector<uint32_t> &pt);
};

class DBSearcher {                                      // dbsearcher.h:14
public:
    virtual ~DBSearcher();
    std::mutex m_Lock;
    const DSSParams *m_Params = nullptr;
    SearchOptions m_Opts;
    std::vector<PDBChain *> m_DBChains;
    std::vector<std::vector<std::vector<byte> > *> m_DBProfiles;
    std::vector<std::vector<byte> *> m_DBMuLettersVec;
    std::vector<std::vector<uint> *> m_DBMuKmersVec;
    std::vector<float> m_DBSelfRevScores;
    std::vector<std::vector<std::vector<byte> > > m_RevProfiles;   // LoadBCA -> ComputeSelfRevScores: profiles of the reversed chains
    double m_MaxEvalue = 10;
    uint64_t m_HitCount = 0;
    uint64_t m_ProcessedPairCount = 0;
    // run statistics (cf. DSSAligner::Stats dssaligner.cpp:1088)
    uint64_t m_AlnCount = 0, m_MuFilterInputCount = 0, m_MuFilterDiscardCount = 0, m_MKFPairCount = 0, m_SWCount = 0;
    FILE *m_fTsv = nullptr;                             // g_fTsv of output.cpp
    rsk_ctx *m_Ctx = nullptr;
    rsk_db *m_Db = nullptr;                             // the loaded chains in HBM
    DSSAligner m_DA;                                    // the aligner the hits are replayed through

    uint GetDBChainCount() const { return (uint) m_DBChains.size(); }
    // Loads chains + per-chain features.  Stage 1 reads the ".rskdb" container (precomputed
    // DSS profile / Mu letters / 3-mers / CA coordinates / self-rev score per chain, format in
    // tests/fixtures.py and DESIGN.md); .bca + on-the-fly DSS featurisation is row (f) "next".
    void LoadDB(const std::string &DBFN);
    // .bca input: chains are featurised on the host cores (DSS) and the self-rev scores (P8, GetSelfRevScore
    // alignpair.cpp:7) are computed in one GPU batch.  m_SelfRevQueryFlavour selects the aligner settings the
    // reference uses for them: false = ProfileLoader (profileloader.cpp:23-26: Omega 0, no Mu filter),
    // true = the search/PostMuFilter params themselves (runquery.cpp:43, postmufilter.cpp:79,171).
    bool m_SelfRevQueryFlavour = false;
    void LoadBCA(const std::string &FN);
    void ComputeSelfRevScores();
    void AddChain(PDBChain *ptrChain, std::vector<std::vector<byte> > *ptrProfile, std::vector<byte> *ptrMuLetters);
    void Setup();                                       // dbsearcher.cpp:73
    void RunSelf();                                     // runself.cpp:101
    void RunQuery(ChainReader2 &QCR);                   // runquery.cpp:82: the reader's chains are streamed past our chains in batches
    void RunQuery(DBSearcher &DBChainsSource);          // the same for a chain set that is already loaded (A = its chains, B = ours)
    // chains per streamed batch of RunQuery(ChainReader2 &): bounds the host RAM / HBM a -db file needs, whatever its size
    // (measured, 256 queries x 125,000 chains -sensitive: 131072 per batch 4.9 s, 32768 3.5 s, 16384 3.4 s -- the loader
    // thread featurises batch k + 1 under batch k's kernels)
    uint m_StreamBatchChains = 1u << 15;
    uint64_t m_StreamBatchResidues = 8ull << 20;
    void WriteRskdb(const std::string &FN) const;       // the loaded (featurised) chains as an RSKDB1 container: what LoadDB reads back
    void LoadChains(std::vector<PDBChain *> &Chains);   // take ownership, featurise on the host threads, self-rev scores on the GPU
    bool m_OwnsChains = true;
    void MakeView(const DBSearcher &Src, uint Lo, uint Hi);
    // Devices this searcher drives.  Empty or one entry = the device of m_Ctx.  Setup() fills it from
    // RSK_DEVICES ("0,1,2,3"; an id may repeat = several contexts on one device) unless the caller set it.  With several
    // entries RunSelf / RunQuery run one shard per entry, each on a host thread and a context of its own.
    std::vector<int> m_Devices;
    bool OnSeveralDevices() const;
    static std::vector<int> ParseDeviceList(const char *Str);      // "0,1,2" -> {0, 1, 2}; throws on anything else
    // shard bounds (pure arithmetic): targets [Lo, Hi) of the self-search triangle with equal DP cells per shard; contiguous
    // chain ranges with equal residues per shard
    static void SelfShardRange(const uint32_t *Lens, uint64_t N, uint Index, uint Count, uint64_t &Lo, uint64_t &Hi);
    static void SelfWindowRange(const uint32_t *Lens, uint64_t N, uint Index, uint Count, uint64_t &Lo, uint64_t &Hi);   // positions of the length order
    static void ResidueShardRange(const uint32_t *Lens, uint64_t N, uint Index, uint Count, uint64_t &Lo, uint64_t &Hi);
    void RunSelfShard(uint Index, uint Count);          // one rank's part of the self-search triangle
    bool Reject(DSSAligner &DA, bool Up) const;         // dbsearcher.cpp:258
    void BaseOnAln(DSSAligner &DA, bool Up);            // dbsearcher.cpp:267
    virtual void OnSetup() {}
    virtual void OnAln(DSSAligner &DA, bool Up) {}
    bool m_HasOnAlnOverride = false;                    // subclasses that override Reject/BaseOnAln semantics set this: every aligned pair is then replayed

    void UploadToGpu();

private:
    void AlignPairBatch(const std::vector<uint32_t> &ia, const std::vector<uint32_t> &ib, bool Self);
};

// `reseek -search Q -db DB -fast` (search.cpp:62-111): k-mer prefilter over the Mu letters, then the
// candidates of its hand-off file are aligned under the "sensitive" preset.
//   MuPreFilter  muprefilter.cpp:70   (query index + neighbourhoods, per-target scan, RankedScoresBag, ToTsv)
//   PostMuFilter postmufilter.cpp:190 (per target line: AlignBags against each listed query, Accept, ToTsv)
// The MKF path for a list of (A, B) pairs: seeding of every pair on the GPU (rsk_mkf_seed_pairs), chaining +
// gapped X-drop + statistics of the pairs with a seed HSP on host threads; OnHit is called under a lock
// for every pair that ends with an alignment (DA.m_Path non-empty is NOT required: the caller decides).
void ForEachAlignedBatch(const DSSParams &P, rsk_ctx *ctx, const SearchOptions &O, DBSearcher &SrcA, DBSearcher &SrcB,
                         const std::vector<uint32_t> &ia, const std::vector<uint32_t> &ib,
                         const std::function<void(const std::vector<uint32_t> &, const std::vector<uint32_t> &, const std::vector<rsk_aln> &,
                                                  const char *)> &OnBatch);
// OnHit is called under a lock (one hit at a time, the reference's m_Lock semantics).  OnHitOfWorker, if given, replaces it:
// called without the lock with the index of the calling worker thread (< HostThreads(128)); calls of one worker are serial.
void RunMKFPairs(rsk_ctx *Ctx, const DSSParams &Params, const std::string &Columns, DBSearcher &SrcA, DBSearcher &SrcB,
                 const std::vector<std::pair<uint32_t, uint32_t> > &Pairs, const std::function<void(DSSAligner &, uint, uint)> &OnHit,
                 const std::function<void(DSSAligner &, uint, uint, unsigned)> *OnHitOfWorker = nullptr);

// The long-chain job beside another job of the same search: RunMKFPairs on a helper context of `Ctx`'s device (own stream,
// own host thread) while AlignJob() runs on the calling thread; a pair that ends with an alignment is offered to Keep
// (called concurrently from the workers: no shared state), kept hits are formatted into per-worker buffers (ToTsv, Up) and
// appended to fTsv after both jobs -- the order of the hits file stays "alignment job, then long-chain job".  Returns the
// number of long-chain hits.
uint64_t RunMKFPairsBeside(rsk_ctx *Ctx, const DSSParams &Params, const std::string &Columns, DBSearcher &SrcA, DBSearcher &SrcB,
                           const std::vector<std::pair<uint32_t, uint32_t> > &Pairs, const std::function<void()> &AlignJob,
                           const std::function<bool(const DSSAligner &)> &Keep, bool Up, FILE *fTsv);

// [b, e) ranges of a pair list such that each batch has <= batch_pairs pairs and <= batch_cells DP cells
std::vector<std::pair<size_t, size_t> > AlignBatches(const SearchOptions &O, const DBSearcher &A, const DBSearcher &B,
                                                     const std::vector<uint32_t> &ia, const std::vector<uint32_t> &ib);

// museqsource.h:8 -- the Mu-letter view of a chain file (OpenChains: chains are featurised on the fly) or of a Mu FASTA
// (OpenFasta, -dbmu).  m_ASCII = true yields 'A' + letter characters (what SeqDB::FromSS stores), false raw letters.
class MuSeqSource {
public:
    bool m_IsFasta = false, m_ASCII = true;
    ChainReader2 m_CR;
    const DSSParams *m_Params = nullptr;
    std::vector<std::string> m_FaLabels;                // OpenFasta: read up front
    std::vector<std::vector<byte> > m_FaSeqs;
    size_t m_FaNext = 0;
    void OpenFasta(const std::string &FileName);
    void OpenChains(const std::string &FileName, const DSSParams &Params);
    bool GetNext(std::string &Label, std::vector<byte> &Seq);   // SeqSource::GetNext (seqsource.h), value form
    // everything that is left, featurised on the host threads (what MuPreFilter / SeqDB::FromSS consume)
    void GetAll(std::vector<std::string> &Labels, std::vector<std::vector<byte> > &Seqs);
    void Close() {}
};

class SeqDB {                                           // seqdb.h:9 (the members the path uses)
public:
    std::vector<std::string> m_Labels, m_Seqs;
    uint GetSeqCount() const { return (uint) m_Seqs.size(); }
    unsigned AddSeq(const std::string &Label, const std::string &Seq) { m_Labels.push_back(Label); m_Seqs.push_back(Seq); return GetSeqCount() - 1; }
    const std::string &GetSeq(unsigned i) const { return m_Seqs[i]; }
    const std::string &GetLabel(unsigned i) const { return m_Labels[i]; }
    unsigned GetSeqLength(unsigned i) const { return (unsigned) m_Seqs[i].size(); }
    void FromSS(MuSeqSource &SS);                       // seqdb.cpp FromSS
    void ToLetters(const byte *CharToLetter);           // seqdb.cpp ToLetters (in place)
};
extern const byte *const g_CharToLetterMu;              // alpha.cpp:3291, 256 entries (sic: 'K' -> 10, 'L' -> 11; 0xFF = invalid)

// search.cpp:9-18: the two stages of `-search -fast -db` with the reference's argument lists ...
void MuPreFilter(const DSSParams &Params, SeqDB &QueryDB, MuSeqSource &FSS, const std::string &OutputFN);
void PostMuFilter(const DSSParams &Params, const std::string &MuFilterTsvFN, const std::string &QueryCAFN, const std::string &DBBCAFN,
                  const std::string &HitsFN);
// ... and on chain sets that are already loaded (what rsk_search uses)
void MuPreFilter(const DSSParams &Params, DBSearcher &QDB, DBSearcher &TDB, const std::string &OutputFN);
void PostMuFilter(const DSSParams &Params, const std::string &MuFilterTsvFN, DBSearcher &Q, DBSearcher &DB,
                  const std::string &HitsFN);
// ... and with the candidates handed over in memory, in the hand-off file's order (rsk_search; KeepTmpFN != "" also writes the file)
void MuPreFilterToPairs(DBSearcher &QDB, DBSearcher &TDB, std::vector<uint32_t> &pq, std::vector<uint32_t> &pt, const std::string &KeepTmpFN);
void PostMuFilterPairs(const DSSParams &Params, DBSearcher &Q, DBSearcher &DB, const std::vector<uint32_t> &pq, const std::vector<uint32_t> &pt,
                       const std::string &HitsFN);

}   // namespace reseek_amd
