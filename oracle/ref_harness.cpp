// ref_harness -- TEST INFRASTRUCTURE ONLY (never linked into or called by the product).
//
// Our own main() linked against the *unmodified* reference objects (everything in
// /root/reference/src except reseek_main.cpp; recipe: oracle/Makefile.ref).  It calls the
// reference's own classes/functions and dumps (a) per-chain inputs of the -search hot
// path, (b) per-pair intermediates, (c) known-answer vectors for the scalar kernels that
// have no live caller, (d) the in-memory constant tables.  Outputs are *data* fixtures
// (tests/golden/) -- no reference source text is emitted.
//
// Reference entry points exercised (file:line are relative to /root/reference/src):
//   DSS::GetProfile/GetMuLetters/GetMuKmers   dss.cpp:716,700,659
//   GetSelfRevScore                           alignpair.cpp:7
//   DSSAligner::SetQuery/SetTarget/...        dssaligner.cpp:674,701,929
//   parasail_sw_striped_profile_avx2_256_8    parasail.cpp:515
//   SWFastPinopGapless / SWFastPinop          swfastpinopgapless.cpp:6 / swfastpinop.cpp:6
//   SWFastGapless_Int                         swgaplessint.cpp:7
//   SWFast                                    sw.cpp:79
//   DSSParams::ApplyWeights                   dssparams.cpp:344
//
// usage: ref_harness <subcmd> <args...> [-- <reseek options, e.g. -sensitive>]
//   tables   <out.h>
//   db       <in.bca> <out.rskdb>            [-- -sensitive|-fast|-verysensitive]   (self-rev as DBSearcher::LoadDB)
//   dbq      <in.bca> <out.rskdb>            [-- mode]   (self-rev as ThreadBodyQuery computes it for streamed -db chains)
//   pairs    <in.bca> <out.bin> <maxchains>  [-- mode]
//   mukat    <in.mu.fa> <first> <count> <out.bin>
//   randkat  <seed> <npairs> <out.bin>
//   xdropkat <seed> <nrandom> <out.bin>      (the reference's -test_xdrop / testsw peptide pairs + random ones)
//   xdrophsp <in.bca> <out.bin> <maxchains>  [-- mode]   (long-chain pairs: chained HSPs, mega scores, XDropHSP start, XDropFwd / XDropBwd, merge)
//   benchmu  <in.mu.fa> <npairs> <threads>    (times the reference kernels; bench.py cpu_baseline)

#include "myutils.h"
#include "dss.h"
#include "dssaligner.h"
#include "museqsource.h"
#include "seqinfo.h"
#include "prefiltermu.h"
#include "seqdb.h"
#include "chainreader2.h"
#include "parasail.h"
#include "mumx.h"
#include "xdpmem.h"
#include "mx.h"
#include "seqdb.h"
#include "alpha.h"
#include <algorithm>
#include <cstdio>
#include <cstring>

int g_Frame = 0;
string g_Arg1;

float GetSelfRevScore(DSSAligner &DA, DSS &D, const PDBChain &Chain,
  const vector<vector<byte> > &Profile, const vector<byte> *ptrMuLetters,
  const vector<uint> *ptrMuKmers);
uint SWFastPinopGapless(const int8_t * const *AP, uint LA, const int8_t *B, uint LB);
uint SWFastPinop(XDPMem &Mem, const int8_t * const *AP, uint LA, const int8_t *B, uint LB,
  int8_t Open, int8_t Ext);
int SWFastGapless_Int(XDPMem &Mem, const Mx<int8_t> &SMx, uint LA, uint LB,
  uint &Besti, uint &Bestj);
float SWFast(XDPMem &Mem, const float * const *SMxData, uint LA, uint LB,
  float Open, float Ext, uint &Loi, uint &Loj, uint &Leni, uint &Lenj, string &Path);
extern parasail_matrix_t parasail_mu_matrix;

static void w32(FILE *f, uint32_t v) { fwrite(&v, 4, 1, f); }
static void wi32(FILE *f, int32_t v) { fwrite(&v, 4, 1, f); }
static void wf32(FILE *f, float v) { fwrite(&v, 4, 1, f); }
static void wbytes(FILE *f, const void *p, size_t n) { if (n) fwrite(p, 1, n, f); }
static void wstr(FILE *f, const string &s);

static void InitOpts(int argc, char **argv, int first_opt)
	{
	// Build a fake reseek command line so the reference option globals are initialised.
	vector<string> Args;
	Args.push_back("reseek");
	Args.push_back("-search");
	Args.push_back("dummy");
	Args.push_back("-threads");
	Args.push_back("1");
	Args.push_back("-quiet");
	for (int i = first_opt; i < argc; ++i)
		Args.push_back(argv[i]);
	vector<char *> av;
	for (size_t i = 0; i < Args.size(); ++i)
		av.push_back(strdup(Args[i].c_str()));
	MyCmdLine((int) av.size(), av.data());
	}

struct ChainData
	{
	PDBChain *Chain = 0;
	vector<vector<byte> > Profile;
	vector<byte> Mu;
	vector<uint> Kmers;
	float SelfRev = 0;
	};

// Same per-chain precompute as ProfileLoader::ThreadBody (profileloader.cpp:18-70).
static bool s_QueryModeSelfRev = false;

static void LoadChains(const string &FN, const DSSParams &Params,
  vector<ChainData *> &CDs, uint MaxChains)
	{
	ChainReader2 CR;
	CR.Open(FN);
	DSS D;
	D.SetParams(Params);
	DSSAligner DA;
	DSSParams DA_Params = Params;
	if (!s_QueryModeSelfRev)
		{
	// LoadDB flavour (profileloader.cpp:23-26)
		DA_Params.m_UsePara = false;
		DA_Params.m_Omega = 0;
		}
	// else: RunQuery flavour: self-rev of a streamed -db chain uses the search params (runquery.cpp:43-44)
	DA_Params.m_OwnScoreMxs = false;
	DA.SetParams(DA_Params);
	for (;;)
		{
		if (SIZE(CDs) >= MaxChains)
			break;
		PDBChain *Chain = CR.GetNext();
		if (Chain == 0)
			break;
		ChainData *CD = new ChainData;
		CD->Chain = Chain;
		D.Init(*Chain);
		D.GetProfile(CD->Profile);
		D.GetMuLetters(CD->Mu);
		D.GetMuKmers(CD->Mu, CD->Kmers, Params.m_MKFPatternStr);
		CD->SelfRev = GetSelfRevScore(DA, D, *Chain, CD->Profile, &CD->Mu, &CD->Kmers);
		Chain->m_Idx = SIZE(CDs);
		CDs.push_back(CD);
		}
	}

static void cmd_tables(const string &OutFN)
	{
	DSSParams Params;
	Params.SetDSSParams(DM_AlwaysSensitive);
	FILE *f = fopen(OutFN.c_str(), "w");
	asserta(f != 0);
	fprintf(f, "// GENERATED DATA -- do not edit.  Produced by oracle/ref_harness `tables` from the\n");
	fprintf(f, "// reference's in-memory constant tables (values, not source):\n");
	fprintf(f, "//   rsk_feature_mx[f] = w_f * g_ScoreMxs2[F_f]   (dssparams.cpp:344-362, weights namedparams.cpp:36-43)\n");
	fprintf(f, "//   rsk_mu_int        = IntScoreMx_Mu == parasail_mu_ (mumx_data.cpp:42, parasail_mu.cpp:23-60)\n");
	fprintf(f, "//   rsk_mu_s8         = Mu_S_ij_i8 (mumx_data.cpp:81)\n");
	fprintf(f, "//   rsk_mu_f32        = ScoreMx_Mu (mumx_data.cpp:3)\n");
	fprintf(f, "//   rsk_aa_letter     = g_CharToLetterAmino (alpha.cpp:271);  rsk_bins_* = thresholds of DSS::ValueToInt_* (valuetoint.cpp)\n");
	fprintf(f, "// Floats are written as exact hex-float literals.\n");
	const uint FC = Params.GetFeatureCount();
	fprintf(f, "#define RSK_NFEATURES %u\n", FC);
	fprintf(f, "static const unsigned rsk_feature_alpha[RSK_NFEATURES] = {");
	for (uint k = 0; k < FC; ++k)
		fprintf(f, "%s%u", k ? ", " : "", g_AlphaSizes2[Params.m_Features[k]]);
	fprintf(f, "};\n");
	fprintf(f, "static const char *const rsk_feature_name[RSK_NFEATURES] = {");
	for (uint k = 0; k < FC; ++k)
		fprintf(f, "%s\"%s\"", k ? ", " : "", FeatureToStr(Params.m_Features[k]));
	fprintf(f, "};\n");
	fprintf(f, "static const float rsk_feature_weight[RSK_NFEATURES] = {");
	for (uint k = 0; k < FC; ++k)
		fprintf(f, "%s%af", k ? ", " : "", Params.m_Weights[k]);
	fprintf(f, "};\n");
	// all feature matrices are stored padded to 20x20 (row-major), unused = 0
	fprintf(f, "#define RSK_FEATURE_DIM 20\n");
	fprintf(f, "static const float rsk_feature_mx[RSK_NFEATURES][RSK_FEATURE_DIM*RSK_FEATURE_DIM] = {\n");
	for (uint k = 0; k < FC; ++k)
		{
		FEATURE F = Params.m_Features[k];
		uint AS = g_AlphaSizes2[F];
		fprintf(f, " { // %s alpha=%u\n", FeatureToStr(F), AS);
		for (uint a = 0; a < 20; ++a)
			{
			fprintf(f, "  ");
			for (uint b = 0; b < 20; ++b)
				{
				float v = (a < AS && b < AS) ? Params.m_ScoreMxs[F][a][b] : 0.0f;
				fprintf(f, "%af,", v);
				}
			fprintf(f, "\n");
			}
		fprintf(f, " },\n");
		}
	fprintf(f, "};\n");
	fprintf(f, "static const float rsk_gap_open = %af; // m_GapOpen namedparams.cpp:45\n", Params.m_GapOpen);
	fprintf(f, "static const float rsk_gap_ext = %af;  // m_GapExt namedparams.cpp:46\n", Params.m_GapExt);
	fprintf(f, "static const signed char rsk_mu_int[36*36] = {\n");
	for (uint a = 0; a < 36; ++a)
		{
		fprintf(f, " ");
		for (uint b = 0; b < 36; ++b)
			{
			asserta(IntScoreMx_Mu[a][b] == parasail_mu_matrix.matrix[a*36+b]);
			fprintf(f, "%d,", (int) IntScoreMx_Mu[a][b]);
			}
		fprintf(f, "\n");
		}
	fprintf(f, "};\n");
	fprintf(f, "static const signed char rsk_mu_s8[36*36] = {\n");
	for (uint a = 0; a < 36; ++a)
		{
		fprintf(f, " ");
		for (uint b = 0; b < 36; ++b)
			fprintf(f, "%d,", (int) Mu_S_ij_i8[a][b]);
		fprintf(f, "\n");
		}
	fprintf(f, "};\n");
	fprintf(f, "static const float rsk_mu_f32[36*36] = {\n");
	for (uint a = 0; a < 36; ++a)
		{
		fprintf(f, " ");
		for (uint b = 0; b < 36; ++b)
			fprintf(f, "%af,", ScoreMx_Mu[a][b]);
		fprintf(f, "\n");
		}
	fprintf(f, "};\n");
	// amino-acid character -> letter (alpha.cpp:271 g_CharToLetterAmino; 255 = not a letter)
	fprintf(f, "static const unsigned char rsk_aa_letter[256] = {\n");
	for (uint a = 0; a < 256; ++a)
		fprintf(f, "%u,%s", (unsigned) g_CharToLetterAmino[a], a%32 == 31 ? "\n" : "");
	fprintf(f, "};\n");
	// bin thresholds of the float features of the profile (valuetoint.cpp; probed through
	// DSS::ValueToInt_*: t_k = smallest double with ValueToInt(t_k) == k+1, found by bisection)
	{
	DSS D;
	const char *Names[5] = { "NENDist", "RENDist", "DstNxtHlx", "StrandDens", "NormDens" };
	for (int w = 0; w < 5; ++w)
		{
		fprintf(f, "static const double rsk_bins_%s[15] = {", Names[w]);
		for (uint k = 0; k < 15; ++k)
			{
			auto V2I = [&](double v) -> uint
				{
				switch (w)
					{
				case 0: return D.ValueToInt_NENDist(v);
				case 1: return D.ValueToInt_RENDist(v);
				case 2: return D.ValueToInt_DstNxtHlx(v);
				case 3: return D.ValueToInt_StrandDens(v);
				default: return D.ValueToInt_NormDens(v);
					}
				};
			double lo = 0, hi = 1000;      // V2I(lo) <= k < V2I(hi)
			asserta(V2I(lo) <= k && V2I(hi) > k);
			for (int it = 0; it < 200; ++it)
				{
				double mid = lo + (hi - lo)/2;
				if (mid == lo || mid == hi)
					break;
				if (V2I(mid) > k) hi = mid; else lo = mid;
				}
			fprintf(f, "%s%a", k ? ", " : "", hi);
			}
		fprintf(f, "};\n");
		}
	}
	fclose(f);
	}

// rskdb fixture: "RSKDB1\0\0", u32 n, u32 nfeat, then per chain:
// u32 L, u32 labellen, label, seq[L], mu[L], prof[nfeat][L], x[L] y[L] z[L] (f32), selfrev f32,
// u32 nkmers, kmers u32[nkmers]
static void cmd_db(const string &InFN, const string &OutFN)
	{
	DSSParams Params;
	Params.SetDSSParams(DM_UseCommandLineOption);
	vector<ChainData *> CDs;
	LoadChains(InFN, Params, CDs, UINT_MAX);
	FILE *f = fopen(OutFN.c_str(), "wb");
	asserta(f != 0);
	wbytes(f, "RSKDB1\0\0", 8);
	w32(f, SIZE(CDs));
	w32(f, Params.GetFeatureCount());
	for (uint i = 0; i < SIZE(CDs); ++i)
		{
		const ChainData &CD = *CDs[i];
		const PDBChain &C = *CD.Chain;
		uint L = C.GetSeqLength();
		w32(f, L);
		w32(f, SIZE(C.m_Label));
		wbytes(f, C.m_Label.data(), C.m_Label.size());
		wbytes(f, C.m_Seq.data(), L);
		asserta(SIZE(CD.Mu) == L);
		wbytes(f, CD.Mu.data(), L);
		for (uint k = 0; k < SIZE(CD.Profile); ++k)
			{
			asserta(SIZE(CD.Profile[k]) == L);
			wbytes(f, CD.Profile[k].data(), L);
			}
		wbytes(f, C.m_Xs.data(), 4*L);
		wbytes(f, C.m_Ys.data(), 4*L);
		wbytes(f, C.m_Zs.data(), 4*L);
		wf32(f, CD.SelfRev);
		w32(f, SIZE(CD.Kmers));
		wbytes(f, CD.Kmers.data(), 4*CD.Kmers.size());
		}
	fclose(f);
	fprintf(stderr, "db: %u chains -> %s\n", SIZE(CDs), OutFN.c_str());
	}

static int ParaRaw(const parasail_profile_t *prof, const vector<byte> &B, int Open, int Ext, int &Sat)
	{
	parasail_result_t *r = parasail_sw_striped_profile_avx2_256_8(prof,
	  (const char *) B.data(), (int) B.size(), Open, Ext);
	int s = r->score;
	Sat = (r->flag & PARASAIL_FLAG_SATURATED) ? 1 : 0;
	parasail_result_free(r);
	return s;
	}

// Per-pair record (all little-endian 32-bit unless noted):
// u32 i, j, LA, LB
// i32 para_fwd_raw, para_fwd_sat, para_rev_raw, para_rev_sat   (parasail.cpp:515 on A / reversed A vs B)
// f32 mufilter  (AlignMuQP_Para under the mode's Omega/OmegaFwd; parasail_mu.cpp:120)
// i32 gapless_fwd, gapless_rev                                 (SWFastPinopGapless)
// i32 gli_score; u32 gli_besti, gli_bestj                       (SWFastGapless_Int)
// i32 pinop                                                    (SWFastPinop Open=-2 Ext=-1)
// f32 sw_score; u32 loA, loB, pathlen; char path[pathlen]      (SetSMx_NoRev+SWFast)
// u32 hiA, hiB, ids, gaps; f32 lddt, ts, pvalue, evalue, qual    (CalcEvalue; FLT_MAX when skipped)
static void cmd_pairs(const string &InFN, const string &OutFN, uint MaxChains)
	{
	DSSParams Params;
	Params.SetDSSParams(DM_UseCommandLineOption);
	vector<ChainData *> CDs;
	LoadChains(InFN, Params, CDs, MaxChains);
	const uint N = SIZE(CDs);
	FILE *f = fopen(OutFN.c_str(), "wb");
	asserta(f != 0);
	wbytes(f, "RSKPR1\0\0", 8);
	w32(f, N);
	w32(f, N*(N+1)/2);

	// Params with the Mu filter as configured by the mode, para on
	DSSAligner DA;
	DA.SetParams(Params);
	// An aligner with MKF/Omega off so Align_NoAccel is reachable for every pair
	DSSParams PNo = Params;
	PNo.m_OwnScoreMxs = false;
	PNo.m_Omega = 0;
	PNo.m_MKFL = 999999;
	DSSAligner DN;
	DN.SetParams(PNo);
	XDPMem Mem;
	const int Open = Params.m_ParaMuGapOpen;
	const int Ext = Params.m_ParaMuGapExt;

	for (uint i = 0; i < N; ++i)
		{
		const ChainData &A = *CDs[i];
		const uint LA = A.Chain->GetSeqLength();
		parasail_profile_t *ProfF = parasail_profile_create_avx_256_8(
		  (const char *) A.Mu.data(), LA, &parasail_mu_matrix);
		vector<byte> RevA(A.Mu.rbegin(), A.Mu.rend());
		parasail_profile_t *ProfR = parasail_profile_create_avx_256_8(
		  (const char *) RevA.data(), LA, &parasail_mu_matrix);
		vector<const int8_t *> APf(LA), APr(LA);
		for (uint p = 0; p < LA; ++p)
			{
			APf[p] = IntScoreMx_Mu[A.Mu[p]];
			APr[LA-p-1] = IntScoreMx_Mu[A.Mu[p]];
			}
		DA.SetQuery(*A.Chain, &A.Profile, &A.Mu, &A.Kmers, A.SelfRev);
		DN.SetQuery(*A.Chain, &A.Profile, &A.Mu, 0, A.SelfRev);
		for (uint j = i; j < N; ++j)
			{
			const ChainData &B = *CDs[j];
			const uint LB = B.Chain->GetSeqLength();
			w32(f, i); w32(f, j); w32(f, LA); w32(f, LB);
			int SatF, SatR;
			int RawF = ParaRaw(ProfF, B.Mu, Open, Ext, SatF);
			int RawR = ParaRaw(ProfR, B.Mu, Open, Ext, SatR);
			wi32(f, RawF); wi32(f, SatF); wi32(f, RawR); wi32(f, SatR);

			DA.SetTarget(*B.Chain, &B.Profile, &B.Mu, &B.Kmers, B.SelfRev);
			float MuF = 0;
			if (Params.m_Omega > 0)
				MuF = DA.AlignMuQP_Para();
			wf32(f, MuF);

			int GF = (int) SWFastPinopGapless(APf.data(), LA, (const int8_t *) B.Mu.data(), LB);
			int GR = (int) SWFastPinopGapless(APr.data(), LA, (const int8_t *) B.Mu.data(), LB);
			wi32(f, GF); wi32(f, GR);

			Mx<int8_t> SMx;
			SMx.Alloc(LA, LB, __FILE__, __LINE__);
			for (uint p = 0; p < LA; ++p)
				for (uint q = 0; q < LB; ++q)
					SMx.m_Data[p][q] = IntScoreMx_Mu[A.Mu[p]][B.Mu[q]];
			uint Besti, Bestj;
			int GLI = SWFastGapless_Int(Mem, SMx, LA, LB, Besti, Bestj);
			wi32(f, GLI); w32(f, Besti); w32(f, Bestj);

			int Pin = (int) SWFastPinop(Mem, APf.data(), LA, (const int8_t *) B.Mu.data(), LB, -2, -1);
			wi32(f, Pin);

			DN.SetTarget(*B.Chain, &B.Profile, &B.Mu, 0, B.SelfRev);
			DN.Align_NoAccel();
			wf32(f, DN.m_AlnFwdScore);
			w32(f, DN.m_LoA); w32(f, DN.m_LoB);
			w32(f, SIZE(DN.m_Path));
			wbytes(f, DN.m_Path.data(), DN.m_Path.size());
			w32(f, DN.m_HiA); w32(f, DN.m_HiB); w32(f, DN.m_Ids); w32(f, DN.m_Gaps);
			float LDDT = FLT_MAX;
			if (DN.m_EvalueA != FLT_MAX)
				LDDT = DN.GetLDDT();
			wf32(f, LDDT);
			wf32(f, DN.m_NewTestStatisticA);
			wf32(f, DN.m_PvalueA);
			wf32(f, DN.m_EvalueA);
			wf32(f, DN.m_QualityA);
			}
		parasail_profile_free(ProfF);
		parasail_profile_free(ProfR);
		}
	fclose(f);
	fprintf(stderr, "pairs: %u chains -> %s\n", N, OutFN.c_str());
	}

// Mu-only KATs on sequences [first, first+count) of a Mu FASTA, all ordered pairs incl. self.
// Header "RSKMK1\0\0", u32 count; per seq u32 L, bytes; then count*count records of
// i32 para_raw, para_sat, gapless, pinop   (query = row, target = column)
static void cmd_mukat(const string &FaFN, uint First, uint Count, const string &OutFN)
	{
	SeqDB DB;
	DB.FromFasta(FaFN);
	DB.ToLetters(g_CharToLetterMu);
	asserta(First + Count <= DB.GetSeqCount());
	FILE *f = fopen(OutFN.c_str(), "wb");
	asserta(f != 0);
	wbytes(f, "RSKMK1\0\0", 8);
	w32(f, Count);
	vector<vector<byte> > Seqs(Count);
	for (uint k = 0; k < Count; ++k)
		{
		const string &S = DB.GetSeq(First + k);
		Seqs[k].assign(S.begin(), S.end());
		w32(f, SIZE(Seqs[k]));
		wbytes(f, Seqs[k].data(), Seqs[k].size());
		}
	XDPMem Mem;
	for (uint i = 0; i < Count; ++i)
		{
		const vector<byte> &A = Seqs[i];
		const uint LA = SIZE(A);
		parasail_profile_t *Prof = parasail_profile_create_avx_256_8(
		  (const char *) A.data(), LA, &parasail_mu_matrix);
		vector<const int8_t *> AP(LA);
		for (uint p = 0; p < LA; ++p)
			AP[p] = IntScoreMx_Mu[A[p]];
		for (uint j = 0; j < Count; ++j)
			{
			const vector<byte> &B = Seqs[j];
			int Sat;
			int Raw = ParaRaw(Prof, B, 2, 1, Sat);
			int G = (int) SWFastPinopGapless(AP.data(), LA, (const int8_t *) B.data(), SIZE(B));
			int P = (int) SWFastPinop(Mem, AP.data(), LA, (const int8_t *) B.data(), SIZE(B), -2, -1);
			wi32(f, Raw); wi32(f, Sat); wi32(f, G); wi32(f, P);
			}
		parasail_profile_free(Prof);
		}
	fclose(f);
	}

static uint64_t s_rng;
static uint64_t splitmix64()
	{
	uint64_t z = (s_rng += 0x9E3779B97F4A7C15ull);
	z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
	z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
	return z ^ (z >> 31);
	}

// Random + adversarial Mu pairs (low-complexity repeats force saturation and long gaps).
// "RSKRK1\0\0", u32 npairs; per pair: u32 LA, A, u32 LB, B, i32 para_raw, para_sat, gapless, pinop
static void cmd_randkat(uint64_t Seed, uint NPairs, const string &OutFN)
	{
	s_rng = Seed;
	FILE *f = fopen(OutFN.c_str(), "wb");
	asserta(f != 0);
	wbytes(f, "RSKRK1\0\0", 8);
	w32(f, NPairs);
	XDPMem Mem;
	for (uint n = 0; n < NPairs; ++n)
		{
		uint kind = n % 4;
		uint LA = 1 + splitmix64() % (kind == 3 ? 700 : 300);
		vector<byte> A(LA);
		uint alpha = (kind == 1) ? 3 : 36;   // kind 1: tiny alphabet -> high scores / saturation
		for (uint p = 0; p < LA; ++p)
			A[p] = byte(splitmix64() % alpha);
		vector<byte> B;
		if (kind == 2 || kind == 3)
			{
			// mutated copy: sub 0.3 / ins 0.1 / del 0.1 (cf. test_para.cpp:150-174)
			for (uint p = 0; p < LA; ++p)
				{
				uint r = splitmix64() % 100;
				if (r < 10) continue;
				if (r < 20) B.push_back(byte(splitmix64() % 36));
				if (r < 50) B.push_back(byte(splitmix64() % 36));
				else B.push_back(A[p]);
				}
			if (B.empty()) B.push_back(0);
			}
		else
			{
			uint LB = 1 + splitmix64() % 300;
			B.resize(LB);
			for (uint p = 0; p < LB; ++p)
				B[p] = byte(splitmix64() % alpha);
			}
		parasail_profile_t *Prof = parasail_profile_create_avx_256_8(
		  (const char *) A.data(), LA, &parasail_mu_matrix);
		vector<const int8_t *> AP(LA);
		for (uint p = 0; p < LA; ++p)
			AP[p] = IntScoreMx_Mu[A[p]];
		int Sat;
		int Raw = ParaRaw(Prof, B, 2, 1, Sat);
		int G = (int) SWFastPinopGapless(AP.data(), LA, (const int8_t *) B.data(), SIZE(B));
		int P = (int) SWFastPinop(Mem, AP.data(), LA, (const int8_t *) B.data(), SIZE(B), -2, -1);
		parasail_profile_free(Prof);
		w32(f, LA); wbytes(f, A.data(), LA);
		w32(f, SIZE(B)); wbytes(f, B.data(), B.size());
		wi32(f, Raw); wi32(f, Sat); wi32(f, G); wi32(f, P);
		}
	fclose(f);
	}

// CPU baseline for bench.py: time the reference's own scalar/AVX2 kernels on the first
// npairs pairs (row-major upper triangle order of RunSelf, runself.cpp:72-99) of a Mu FASTA.
// Prints one JSON line: cells, seconds per kernel.
#include <thread>
#include <chrono>
static void cmd_benchmu(const string &FaFN, uint64_t NPairs, uint Threads)
	{
	SeqDB DB;
	DB.FromFasta(FaFN);
	DB.ToLetters(g_CharToLetterMu);
	const uint N = DB.GetSeqCount();
	vector<vector<byte> > Seqs(N);
	for (uint k = 0; k < N; ++k)
		{
		const string &S = DB.GetSeq(k);
		Seqs[k].assign(S.begin(), S.end());
		}
	vector<pair<uint,uint> > Pairs;
	for (uint i = 0; i < N && Pairs.size() < NPairs; ++i)
		for (uint j = i; j < N && Pairs.size() < NPairs; ++j)
			Pairs.push_back(make_pair(i, j));
	// deterministic spread over the whole set: stride through the triangle instead of its first rows
	uint64_t Total = uint64_t(N)*(N+1)/2;
	if (Total > NPairs)
		{
		Pairs.clear();
		uint64_t Step = Total/NPairs;
		uint64_t Next = 0, Idx = 0;
		for (uint i = 0; i < N && Pairs.size() < NPairs; ++i)
			{
			uint64_t RowLen = N - i;
			while (Next < Idx + RowLen && Pairs.size() < NPairs)
				{
				Pairs.push_back(make_pair(i, uint(i + (Next - Idx))));
				Next += Step;
				}
			Idx += RowLen;
			}
		}
	double Cells = 0;
	for (size_t p = 0; p < Pairs.size(); ++p)
		Cells += double(Seqs[Pairs[p].first].size())*double(Seqs[Pairs[p].second].size());
	double Secs[2] = {0, 0};
	uint64_t Check[2] = {0, 0};
	for (int Kernel = 0; Kernel < 2; ++Kernel)
		{
		vector<uint64_t> Sums(Threads, 0);
		auto t0 = std::chrono::steady_clock::now();
		vector<std::thread> ts;
		for (uint T = 0; T < Threads; ++T)
			ts.emplace_back([&, T]()
				{
				uint64_t Sum = 0;
				uint PrevQ = UINT_MAX;
				vector<const int8_t *> AP;
				parasail_profile_t *Prof = 0;
				for (size_t p = T; p < Pairs.size(); p += Threads)
					{
					const vector<byte> &A = Seqs[Pairs[p].first];
					const vector<byte> &B = Seqs[Pairs[p].second];
					if (Pairs[p].first != PrevQ)
						{
						PrevQ = Pairs[p].first;
						AP.resize(A.size());
						for (size_t k = 0; k < A.size(); ++k)
							AP[k] = IntScoreMx_Mu[A[k]];
						if (Prof) parasail_profile_free(Prof);
						Prof = parasail_profile_create_avx_256_8((const char *) A.data(), (int) A.size(), &parasail_mu_matrix);
						}
					if (Kernel == 0)
						Sum += SWFastPinopGapless(AP.data(), SIZE(A), (const int8_t *) B.data(), SIZE(B));
					else
						{
						int Sat;
						Sum += (uint64_t) ParaRaw(Prof, B, 2, 1, Sat);
						}
					}
				if (Prof) parasail_profile_free(Prof);
				Sums[T] = Sum;
				});
		for (auto &t : ts) t.join();
		auto t1 = std::chrono::steady_clock::now();
		Secs[Kernel] = std::chrono::duration<double>(t1 - t0).count();
		for (uint T = 0; T < Threads; ++T) Check[Kernel] += Sums[T];
		}
	printf("{\"pairs\": %zu, \"cells\": %.0f, \"threads\": %u, \"gapless_secs\": %.4f, \"parasail_fwd_secs\": %.4f, "
	  "\"gapless_checksum\": %llu, \"parasail_checksum\": %llu}\n",
	  Pairs.size(), Cells, Threads, Secs[0], Secs[1], (unsigned long long) Check[0], (unsigned long long) Check[1]);
	}

// D1 leftovers on real chains: for all ordered pairs (i, j) of the first N chains:
//   f32 SWFastGaplessProfb(ProfMu(A), B)                     (swgaplessprofb.cpp:6; ProfMu = rows of ScoreMx_Mu, dssaligner.cpp:429)
//   f32 SWFastGapless(SMx(A,B)); u32 Besti, Bestj           (swgapless.cpp:46 on the 8-feature SetSMx_NoRev matrix)
// Header "RSKD11\0\0", u32 N.
float SWFastGaplessProfb(float *DProw_, const float * const *ProfA, uint LA, const byte *B, uint LB);
float SWFastGapless(XDPMem &Mem, const Mx<float> &SMx, uint LA, uint LB, uint &Besti, uint &Bestj);
static void cmd_d1pairs(const string &InFN, const string &OutFN, uint MaxChains)
	{
	DSSParams Params;
	Params.SetDSSParams(DM_UseCommandLineOption);
	vector<ChainData *> CDs;
	LoadChains(InFN, Params, CDs, MaxChains);
	const uint N = SIZE(CDs);
	FILE *f = fopen(OutFN.c_str(), "wb");
	asserta(f != 0);
	wbytes(f, "RSKD11\0\0", 8);
	w32(f, N);
	DSSParams PNo = Params;
	PNo.m_OwnScoreMxs = false;
	PNo.m_Omega = 0;
	PNo.m_MKFL = 999999;
	DSSAligner DN;
	DN.SetParams(PNo);
	XDPMem Mem;
	for (uint i = 0; i < N; ++i)
		{
		const ChainData &A = *CDs[i];
		const uint LA = A.Chain->GetSeqLength();
		vector<const float *> ProfMu(LA);
		for (uint p = 0; p < LA; ++p)
			ProfMu[p] = ScoreMx_Mu[A.Mu[p]];
		for (uint j = 0; j < N; ++j)
			{
			const ChainData &B = *CDs[j];
			const uint LB = B.Chain->GetSeqLength();
			vector<float> DProw(2*LB + 8);
			float Pb = SWFastGaplessProfb(DProw.data(), ProfMu.data(), LA, B.Mu.data(), LB);
			wf32(f, Pb);
			DN.SetQuery(*A.Chain, &A.Profile, &A.Mu, 0, A.SelfRev);
			DN.SetTarget(*B.Chain, &B.Profile, &B.Mu, 0, B.SelfRev);
			DN.Align_NoAccel();          // fills the SetSMx_NoRev matrix for this pair
			uint Besti, Bestj;
			Mx<float> SMx;
			SMx.Alloc(LA, LB, __FILE__, __LINE__);
			const float * const *SD = DN.GetSMxData();
			for (uint p = 0; p < LA; ++p)
				for (uint q = 0; q < LB; ++q)
					SMx.m_Data[p][q] = SD[p][q];
			float G = SWFastGapless(Mem, SMx, LA, LB, Besti, Bestj);
			wf32(f, G); w32(f, Besti); w32(f, Bestj);
			}
		}
	fclose(f);
	fprintf(stderr, "d1pairs: %u chains -> %s\n", N, OutFN.c_str());
	}

// MKF seeding known answers: MuKmerFilter::SetQ(A) + Align(B) (mukmerfilter.cpp:234,316) for all ordered pairs of
// the first N chains: kept seed HSPs and the chain.  "RSKMF1\0\0", u32 N; per pair: u32 nkept, nkept x (i32 Loi,
// Loj, Len, Score), i32 BestChainScore, u32 nchain, nchain x (i32 Loi, Loj, Len).
#include "mukmerfilter.h"
static void cmd_mkfkat(const string &InFN, const string &OutFN, uint MaxChains)
	{
	DSSParams Params;
	Params.SetDSSParams(DM_UseCommandLineOption);
	vector<ChainData *> CDs;
	LoadChains(InFN, Params, CDs, MaxChains);
	const uint N = SIZE(CDs);
	FILE *f = fopen(OutFN.c_str(), "wb");
	asserta(f != 0);
	wbytes(f, "RSKMF1\0\0", 8);
	w32(f, N);
	MuKmerFilter MKF;
	MKF.SetParams(Params);
	for (uint i = 0; i < N; ++i)
		{
		const ChainData &A = *CDs[i];
		MKF.SetQ(A.Chain->m_Label, &A.Mu, &A.Kmers);
		for (uint j = 0; j < N; ++j)
			{
			const ChainData &B = *CDs[j];
			MKF.Align(B.Mu, B.Kmers);
			const uint nk = SIZE(MKF.m_MuKmerHSPLois);
			w32(f, nk);
			for (uint k = 0; k < nk; ++k)
				{
				wi32(f, MKF.m_MuKmerHSPLois[k]); wi32(f, MKF.m_MuKmerHSPLojs[k]);
				wi32(f, MKF.m_MuKmerHSPLens[k]); wi32(f, MKF.m_MuKmerHSPScores[k]);
				}
			wi32(f, MKF.m_BestChainScore);
			const uint nc = SIZE(MKF.m_ChainHSPLois);
			w32(f, nc);
			for (uint k = 0; k < nc; ++k)
				{
				wi32(f, MKF.m_ChainHSPLois[k]); wi32(f, MKF.m_ChainHSPLojs[k]); wi32(f, MKF.m_ChainHSPLens[k]);
				}
			}
		MKF.ResetQ();
		}
	fclose(f);
	fprintf(stderr, "mkfkat: %u chains -> %s\n", N, OutFN.c_str());
	}

// xdrophsp: the long-chain path of every ordered pair of the first N chains, step by step through the reference's own
// functions (PostAlignMKF dssaligner.cpp:1395 / XDropHSP xdrophsp.cpp:42 restated as calls, not copied): chained HSPs of
// MuKmerFilter::Align, DSSAligner::GetMegaHSPScore of each, the start XDropHSP derives from StaticSubstScore, XDropFwd and
// XDropBwd from that start, MergeFwdBwd.  The result is checked against DSSAligner::AlignMKF of the same pair before it is
// written, so the fixture is the reference's own answer split into its stages.
static void cmd_xdrophsp(const string &InFN, const string &OutFN, uint MaxChains)
	{
	DSSParams Params;
	Params.SetDSSParams(DM_UseCommandLineOption);
	vector<ChainData *> CDs;
	LoadChains(InFN, Params, CDs, MaxChains);
	const uint N = SIZE(CDs);
	FILE *f = fopen(OutFN.c_str(), "wb");
	asserta(f != 0);
	wbytes(f, "RSKXH1\0\0", 8);
	w32(f, N);
	DSSAligner DA;
	DA.SetParams(Params);
	XDPMem Mem;
	uint NRec = 0, NAln = 0;
	for (uint i = 0; i < N; ++i)
		{
		const ChainData &A = *CDs[i];
		for (uint j = 0; j < N; ++j)
			{
			const ChainData &B = *CDs[j];
			DA.SetQuery(*A.Chain, &A.Profile, &A.Mu, &A.Kmers, A.SelfRev);
			DA.SetTarget(*B.Chain, &B.Profile, &B.Mu, &B.Kmers, B.SelfRev);
			if (!DA.DoMKF())
				continue;
			DA.AlignMKF();                          // the reference's answer for the pair
			const string RefPath = DA.m_Path;
			const float RefScore = DA.m_AlnFwdScore;
			const uint RefLoA = DA.m_LoA, RefLoB = DA.m_LoB;
			// ... and its stages
			const MuKmerFilter &MKF = DA.m_MKF;
			const uint M = SIZE(MKF.m_ChainHSPLois);
			w32(f, i); w32(f, j);
			wi32(f, MKF.m_BestChainScore);
			w32(f, M);
			float MegaTotal = 0, BestMega = 0;
			uint BestIdx = 0;
			for (uint k = 0; k < M; ++k)
				{
				const float Mega = DA.GetMegaHSPScore((uint) MKF.m_ChainHSPLois[k], (uint) MKF.m_ChainHSPLojs[k], (uint) MKF.m_ChainHSPLens[k]);
				wi32(f, MKF.m_ChainHSPLois[k]); wi32(f, MKF.m_ChainHSPLojs[k]); wi32(f, MKF.m_ChainHSPLens[k]); wf32(f, Mega);
				if (Mega > BestMega) { BestMega = Mega; BestIdx = k; }
				MegaTotal += Mega;
				}
			wf32(f, MegaTotal);
			const bool Gate = MKF.m_BestChainScore > 0 && M > 0 && !(MegaTotal < Params.m_MKF_MinMegaHSPScore);
			w32(f, Gate ? 1 : 0);
			++NRec;
			if (!Gate)
				{
				asserta(RefPath.empty());
				continue;
				}
			// start of the gapped extensions: best 8-mer of the best HSP under StaticSubstScore
			const uint Li = (uint) MKF.m_ChainHSPLois[BestIdx], Lj = (uint) MKF.m_ChainHSPLojs[BestIdx], Len = (uint) MKF.m_ChainHSPLens[BestIdx];
			uint LoA = Li + Len/2, LoB = Lj + Len/2;
			float BestMer = 0;
			for (uint s0 = 0; s0 + 8 <= Len; ++s0)
				{
				float Mer = 0;
				for (uint k = 0; k < 8; ++k)
					Mer += DSSAligner::StaticSubstScore((void *) &DA, Li + s0 + k, Lj + s0 + k);
				if (Mer > BestMer) { BestMer = Mer; LoA = Li + s0; LoB = Lj + s0; }
				}
			if (min(LoA, LoB) < 4) { LoA += 4; LoB += 4; }
			const uint LA = A.Chain->GetSeqLength(), LB = B.Chain->GetSeqLength();
			string FwdPath, BwdPath, Path;
			uint s1, s2;
			const float ScoreFwd = XDropFwd(Mem, float(Params.m_MKF_X2), Params.m_GapOpen, Params.m_GapExt, DSSAligner::StaticSubstScore,
			  (void *) &DA, LoA, LA, LoB, LB, &s1, &s2, FwdPath);
			const float ScoreBwd = XDropBwd(Mem, float(Params.m_MKF_X2), Params.m_GapOpen, Params.m_GapExt, DSSAligner::StaticSubstScore,
			  (void *) &DA, LoA - 1, LA, LoB - 1, LB, &s1, &s2, BwdPath);
			const float Total = ScoreFwd + ScoreBwd;
			uint MLoA = UINT_MAX, MLoB = UINT_MAX, MHiA = UINT_MAX, MHiB = UINT_MAX;
			if (!(Total < 10))
				MergeFwdBwd(LA, LB, LoA, LoB, FwdPath, LoA - 1, LoB - 1, BwdPath, MLoA, MLoB, MHiA, MHiB, Path);
			// the stages reproduce the reference's own result
			if (Total < 10)
				asserta(RefPath.empty() && RefScore == 0);
			else
				asserta(RefPath == Path && RefScore == Total && RefLoA == MLoA && RefLoB == MLoB);
			w32(f, BestIdx); w32(f, LoA); w32(f, LoB);
			wf32(f, ScoreFwd); wstr(f, FwdPath);
			wf32(f, ScoreBwd); wstr(f, BwdPath);
			wf32(f, Total < 10 ? 0.0f : Total);
			w32(f, MLoA); w32(f, MLoB);
			wstr(f, Path);
			wf32(f, DA.m_EvalueA); wf32(f, DA.GetLDDT());
			if (!Path.empty()) ++NAln;
			}
		}
	w32(f, UINT_MAX);
	fclose(f);
	fprintf(stderr, "xdrophsp: %u chains, %u long-chain pairs, %u aligned -> %s\n", N, NRec, NAln, OutFN.c_str());
	}

// The k-mer neighbourhood prefilter as cmd_search runs it (search.cpp:78-100 -> MuPreFilter
// muprefilter.cpp:70), on Mu FASTA inputs: writes the (query, target, score) list of the
// RankedScoresBag and the target-major hand-off TSV.  Mode: idxq | idxt | auto (via -idxq/-idxt after --).
void MuPreFilter(const DSSParams &Params, SeqDB &QDB, MuSeqSource &FSS, const string &OutputFN);
static void cmd_prefhood(const string &QFa, const string &TFa, const string &ScoresFN, const string &TmpFN)
	{
	DSSParams Params;
	Params.SetDSSParams(DM_UseCommandLineOption);
	SeqDB QDB;
	QDB.FromFasta(QFa);
	SeqDB TDB;
	TDB.FromFasta(TFa);
	MuSeqSource FSS;
	FSS.OpenFasta(TFa);
	MuPreFilter(Params, QDB, FSS, TmpFN);
	vector<string> QLabels, TLabels;
	for (uint i = 0; i < QDB.GetSeqCount(); ++i) QLabels.push_back(QDB.GetLabel(i));
	for (uint i = 0; i < TDB.GetSeqCount(); ++i) TLabels.push_back(TDB.GetLabel(i));
	FILE *f = CreateStdioFile(ScoresFN);
	PrefilterMu::m_RSB.ToScoreTsv(f, QLabels, TLabels);
	CloseStdioFile(f);
	}


// prefrange / rsbreplay: stage 1 of `-search Q.bca -db DB.bca -fast` (search.cpp:76-111) cut into target ranges so that
// several one-thread processes share it (the literal command takes hours on one thread for an 11,211-chain set and its
// bags are only reproducible with one thread).  The reference's own objects do all the work:
//   prefrange Q.bca DB.bca lo hi out.bin : MuSeqSource::OpenChains + SeqDB::FromSS / ToLetters for the queries exactly as
//     cmd_search hands them to MuPreFilter (search.cpp:91-98), the MuDex / MerMx set-up of MuPreFilter (muprefilter.cpp:70-110),
//     then PrefilterMu::Search (prefiltermu.cpp:382) for the targets lo <= index < hi of the DB in file order -- with a bag that
//     never truncates (m_B huge), so PrefilterMu::m_RSB ends up holding EVERY (query, target, score) the scan produced for
//     these targets.  They are written as uint32 triples.
//   rsbreplay NQ B out.tsv in1.bin in2.bin ... : the triples of all ranges, fed to the reference's RankedScoresBag::AddScore in
//     target order (what one thread walking the DB does; the order among the queries of ONE target does not matter, they are
//     different bags), then RankedScoresBag::ToTsv = the hand-off file.
// make_full_golden.py checks this route against the literal one-thread command on a sample before using it.
static void cmd_prefrange(const string &QBca, const string &DBBca, uint Lo, uint Hi, const string &OutFN)
	{
	DSSParams Params;
	Params.SetDSSParams(DM_UseCommandLineOption);
	MuSeqSource QSS;
	QSS.OpenChains(QBca, Params);
	SeqDB QDB;
	QDB.FromSS(QSS);
	const uint QSeqCount = QDB.GetSeqCount();
	if (opt(idxq)) g_QueryNeighborhood = true;
	else if (opt(idxt)) g_QueryNeighborhood = false;
	else g_QueryNeighborhood = (QSeqCount <= MAX_QUERY_CHAINS_FOR_QUERY_NEIGHBORHOOD);
	const uint k = MuDex::m_k;
	QDB.ToLetters(g_CharToLetterMu);
	PrefilterMu::m_RSB.m_B = 0x3FFFFFFF;                 // no truncation: the bag keeps every triple of this range
	PrefilterMu::m_RSB.Init(QSeqCount);
	const MerMx &ScoreMx = GetMuMerMx(k);
	MuDex QKmerIndex;
	QKmerIndex.m_AddNeighborhood = g_QueryNeighborhood;
	QKmerIndex.m_KmerSelfScores = ScoreMx.BuildSelfScores_Kmers();
	QKmerIndex.m_MinKmerSelfScore = MIN_KMER_PAIR_SCORE;
	QKmerIndex.FromSeqDB(QDB);
	PrefilterMu Pref;
	Pref.m_OneHitDiag = false;
	Pref.m_ScoreMx = &ScoreMx;
	Pref.m_QKmerIndex = &QKmerIndex;
	Pref.m_KmerSelfScores = QKmerIndex.m_KmerSelfScores;
	Pref.SetQDB(QDB);
	MuSeqSource DBSS;
	DBSS.OpenChains(DBBca, Params);
	DBSS.m_ASCII = false;
	ObjMgr OM;
	for (uint TIdx = 0; TIdx < Hi; ++TIdx)
		{
		SeqInfo *SI = OM.GetSeqInfo();
		if (!DBSS.GetNext(SI)) { OM.Down(SI); break; }
		if (TIdx >= Lo && SI->m_L != 0)
			Pref.Search(TIdx, string(SI->m_Label), SI->m_Seq, SI->m_L);
		OM.Down(SI);
		}
	FILE *f = CreateStdioFile(OutFN);
	uint64_t n = 0;
	const RankedScoresBag &B = PrefilterMu::m_RSB;
	for (uint q = 0; q < QSeqCount; ++q)
		for (size_t i = 0; i < B.m_QueryIdxToScoreVec[q].size(); ++i)
			{
			w32(f, q); w32(f, B.m_QueryIdxToTargetIdxVec[q][i]); w32(f, B.m_QueryIdxToScoreVec[q][i]);
			++n;
			}
	CloseStdioFile(f);
	fprintf(stderr, "prefrange [%u, %u): %llu triples\n", Lo, Hi, (unsigned long long) n);
	}

static void cmd_rsbreplay(uint NQ, uint B, const string &OutFN, const vector<string> &Ins)
	{
	struct Tr { uint32_t q, t, s; };
	vector<Tr> All;
	for (const string &fn : Ins)
		{
		FILE *f = fopen(fn.c_str(), "rb");
		if (!f) { fprintf(stderr, "cannot open %s\n", fn.c_str()); exit(2); }
		Tr x;
		while (fread(&x, sizeof x, 1, f) == 1) All.push_back(x);
		fclose(f);
		}
	std::stable_sort(All.begin(), All.end(), [](const Tr &a, const Tr &b) { return a.t < b.t; });
	RankedScoresBag RSB;
	RSB.m_B = B;
	RSB.Init(NQ);
	for (const Tr &x : All) RSB.AddScore(x.q, x.t, (uint16_t) x.s);
	FILE *f = CreateStdioFile(OutFN);
	RSB.ToTsv(f);
	CloseStdioFile(f);
	fprintf(stderr, "rsbreplay: %zu triples\n", All.size());
	}

// xdropkat: the reference's own X-drop / SW self-test vectors (test_xdrop.cpp:177-187: three peptide pairs through
// SWFast, XDropFwd, XDropBwd, MergeFwdBwd with BLOSUM62, Open -3, Ext -1, X 8; swgaplessprof.cpp:158-166: six
// peptide pairs through SWGapless) plus <nrandom> random peptide pairs run the same way.  The fixture holds the
// inputs as data (the explicit score matrix of each pair) and every output of the reference functions.
void SetBLOSUM62();
float GetBlosum62Score(char a, char b);
float SWGapless(Mx<float> &DPMx, const Mx<float> &SMx, uint LA, uint LB, uint &Loi, uint &Loj, uint &ColCount);
void MergeFwdBwd(uint LA, uint LB, uint FwdLoA, uint FwdLoB, const string &FwdPath, uint BwdHiA, uint BwdHiB, const string &BwdPath,
  uint &LoA, uint &LoB, uint &HiA, uint &HiB, string &Path);
static const float * const *s_KatS;
static float KatSubFn(void *, uint PosA, uint PosB) { return s_KatS[PosA][PosB]; }
static void wstr(FILE *f, const string &s) { w32(f, (uint32_t) s.size()); wbytes(f, s.data(), s.size()); }

static void xdropkat_case(FILE *f, const string &A, const string &B, float Open, float Ext, float X)
	{
	const uint LA = SIZE(A), LB = SIZE(B);
	Mx<float> SMx;
	SMx.Alloc(LA, LB, __FILE__, __LINE__);
	float **S = SMx.GetData();
	for (uint i = 0; i < LA; ++i)
		for (uint j = 0; j < LB; ++j)
			S[i][j] = GetBlosum62Score(A[i], B[j]);
	s_KatS = S;
	wstr(f, A); wstr(f, B);
	wf32(f, Open); wf32(f, Ext); wf32(f, X);
	for (uint i = 0; i < LA; ++i) wbytes(f, S[i], 4 * (size_t) LB);
	XDPMem Mem;
	string SWPath;
	uint Loi, Loj, Leni, Lenj;
	float SWScore = SWFast(Mem, S, LA, LB, Open, Ext, Loi, Loj, Leni, Lenj, SWPath);
	wf32(f, SWScore); w32(f, Loi); w32(f, Loj); w32(f, Leni); w32(f, Lenj); wstr(f, SWPath);
	Mx<float> DPMx;
	uint gLoi = 0, gLoj = 0, gCols = 0;
	float GScore = SWGapless(DPMx, SMx, LA, LB, gLoi, gLoj, gCols);
	wf32(f, GScore); w32(f, gLoi); w32(f, gLoj); w32(f, gCols);
	const uint ColCount = SIZE(SWPath);
	if (ColCount < 8) { w32(f, 0); return; }           // test_xdrop.cpp:113-114
	w32(f, 1);
	uint MidPosA = Loi, MidPosB = Loj;
	for (uint Col = 0; Col < ColCount / 2; ++Col)
		{
		char c = SWPath[Col];
		if (c == 'M' || c == 'D') ++MidPosA;
		if (c == 'M' || c == 'I') ++MidPosB;
		}
	w32(f, MidPosA); w32(f, MidPosB);
	string FwdPath, BwdPath, MergedPath;
	uint FwdSegLoA = 0, FwdSegLoB = 0, BwdSegLoA = 0, BwdSegLoB = 0;
	float FwdScore = XDropFwd(Mem, X, Open, Ext, KatSubFn, 0, MidPosA + 1, LA, MidPosB + 1, LB, &FwdSegLoA, &FwdSegLoB, FwdPath);
	wf32(f, FwdScore); w32(f, FwdSegLoA); w32(f, FwdSegLoB); wstr(f, FwdPath);
	float BwdScore = XDropBwd(Mem, X, Open, Ext, KatSubFn, 0, MidPosA, LA, MidPosB, LB, &BwdSegLoA, &BwdSegLoB, BwdPath);
	wf32(f, BwdScore); w32(f, BwdSegLoA); w32(f, BwdSegLoB); wstr(f, BwdPath);
	if (FwdPath.empty() && BwdPath.empty()) { w32(f, 0); return; }     // MergeFwdBwd asserts on this input (mergefwdback.cpp:11)
	w32(f, 1);
	uint MLoA, MLoB, MHiA, MHiB;
	MergeFwdBwd(LA, LB, MidPosA + 1, MidPosB + 1, FwdPath, MidPosA, MidPosB, BwdPath, MLoA, MLoB, MHiA, MHiB, MergedPath);
	w32(f, MLoA); w32(f, MLoB); w32(f, MHiA); w32(f, MHiB); wstr(f, MergedPath);
	}

static void cmd_xdropkat(uint64_t Seed, uint NRandom, const string &OutFN)
	{
	SetBLOSUM62();
	FILE *f = fopen(OutFN.c_str(), "wb");
	if (!f) Die("cannot create %s", OutFN.c_str());
	static const char *Fixed[][2] = {
		{ "DVLGYLRFLTKGERQANLNF", "WVLGLRFLTKGERQANLNF" },           // test_xdrop.cpp:179-186
		{ "DVLGYLRFLTERQANLNF", "WVLGLRFLTKGERQANLNF" },
		{ "DVLGYLRFLTKGERQANLNF", "WVLGLINSRFLTKGERQANLNF" },
		{ "LQNGSEQVENCE", "LQNGSEQVENCE" },                          // swgaplessprof.cpp:160-165
		{ "QNGSEQVENCE", "LQNGSEQVENCE" },
		{ "LQNGSEQVENCE", "QNGSEQVENCE" },
		{ "LQNGSEQVENC", "QNGSEQVENCE" },
		{ "SEQVENCE", "QVE" },
		{ "QVE", "SEQVENCE" },
	};
	const uint NFixed = sizeof(Fixed) / sizeof(Fixed[0]);
	wbytes(f, "XDKAT1\0\0", 8);
	w32(f, NFixed + NRandom);
	for (uint k = 0; k < NFixed; ++k)
		xdropkat_case(f, Fixed[k][0], Fixed[k][1], -3, -1, 8);
	s_rng = Seed;
	static const char AA[] = "ACDEFGHIKLMNPQRSTVWY";
	for (uint k = 0; k < NRandom; ++k)
		{
		const uint LA = 8 + splitmix64() % 150;
		string A, B;
		for (uint p = 0; p < LA; ++p) A += AA[splitmix64() % 20];
		// mutated copy (sub 0.2 / ins 0.08 / del 0.08), random flanks, sometimes a second copy of a segment
		for (uint p = splitmix64() % 6; p > 0; --p) B += AA[splitmix64() % 20];
		for (uint p = 0; p < LA; ++p)
			{
			uint r = splitmix64() % 100;
			if (r < 8) continue;
			if (r < 16) B += AA[splitmix64() % 20];
			B += r < 36 ? AA[splitmix64() % 20] : A[p];
			}
		if (k % 5 == 0 && LA > 30) B += A.substr(LA / 3, LA / 3);
		for (uint p = splitmix64() % 6; p > 0; --p) B += AA[splitmix64() % 20];
		const float Open = k % 3 == 0 ? -3.0f : (k % 3 == 1 ? -1.5f : -0.685533f * 8);
		const float Ext = k % 3 == 0 ? -1.0f : (k % 3 == 1 ? -0.25f : -0.051881f * 8);
		const float X = k % 4 == 0 ? 8.0f : (k % 4 == 1 ? 4.0f : (k % 4 == 2 ? 16.0f : 2.5f));
		xdropkat_case(f, A, B, Open, Ext, X);
		}
	fclose(f);
	fprintf(stderr, "xdropkat: %u cases -> %s\n", NFixed + NRandom, OutFN.c_str());
	}

int main(int argc, char **argv)
	{
	if (argc < 2)
		{
		fprintf(stderr, "usage: ref_harness tables|db|pairs|mukat|randkat ...\n");
		return 2;
		}
	int dd = argc;
	for (int i = 1; i < argc; ++i)
		if (!strcmp(argv[i], "--"))
			{
			dd = i;
			break;
			}
	InitOpts(argc, argv, dd + 1);
	string Cmd = argv[1];
	vector<string> A;
	for (int i = 2; i < dd; ++i)
		A.push_back(argv[i]);
	if (Cmd == "dbq" && A.size() == 2)
		{
		s_QueryModeSelfRev = true;
		cmd_db(A[0], A[1]);
		}
	else if (Cmd == "tables" && A.size() == 1)
		cmd_tables(A[0]);
	else if (Cmd == "db" && A.size() == 2)
		cmd_db(A[0], A[1]);
	else if (Cmd == "pairs" && A.size() == 3)
		cmd_pairs(A[0], A[1], (uint) atoi(A[2].c_str()));
	else if (Cmd == "mukat" && A.size() == 4)
		cmd_mukat(A[0], (uint) atoi(A[1].c_str()), (uint) atoi(A[2].c_str()), A[3]);
	else if (Cmd == "randkat" && A.size() == 3)
		cmd_randkat(strtoull(A[0].c_str(), 0, 0), (uint) atoi(A[1].c_str()), A[2]);
	else if (Cmd == "mkfkat" && A.size() == 3)
		cmd_mkfkat(A[0], A[1], (uint) atoi(A[2].c_str()));
	else if (Cmd == "d1pairs" && A.size() == 3)
		cmd_d1pairs(A[0], A[1], (uint) atoi(A[2].c_str()));
	else if (Cmd == "prefhood" && A.size() == 4)
		cmd_prefhood(A[0], A[1], A[2], A[3]);
	else if (Cmd == "prefrange" && A.size() == 5)
		cmd_prefrange(A[0], A[1], (uint) atoi(A[2].c_str()), (uint) atoi(A[3].c_str()), A[4]);
	else if (Cmd == "rsbreplay" && A.size() >= 4)
		cmd_rsbreplay((uint) atoi(A[0].c_str()), (uint) atoi(A[1].c_str()), A[2], vector<string>(A.begin() + 3, A.end()));
	else if (Cmd == "xdrophsp" && A.size() == 3)
		cmd_xdrophsp(A[0], A[1], (uint) atoi(A[2].c_str()));
	else if (Cmd == "xdropkat" && A.size() == 3)
		cmd_xdropkat(strtoull(A[0].c_str(), 0, 0), (uint) atoi(A[1].c_str()), A[2]);
	else if (Cmd == "benchmu" && A.size() == 3)
		cmd_benchmu(A[0], strtoull(A[1].c_str(), 0, 0), (uint) atoi(A[2].c_str()));
	else
		{
		fprintf(stderr, "bad subcommand/args\n");
		return 2;
		}
	return 0;
	}
