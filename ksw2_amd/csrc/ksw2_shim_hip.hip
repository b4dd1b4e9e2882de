/*
 * ksw2_shim_hip.hip -- gfx950 kernels + HIP runtime side of the shim (ksw2_shim.h).
 *
 * K1-K4 (fill): k2a_fill_kernel<G, C, DUAL, MODE>   one alignment per group of G lanes, see ksw2_lane.h
 * K5 (trace):   k2a_trace_kernel<G, C, DUAL>        one thread per alignment walks its traceback block
 *
 * The lane-to-lane hand-off of the bottom row's (H, E[, E~]) is a single DPP rotate per value per step:
 * wave_ror:1 for G = 64, row_ror:1 for G = 16 (four independent alignments per wavefront).
 */
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include "ksw2_shim.h"
#include "ksw2_lane.h"
#include "ksw2_lane_pk.h"
#include "ksw2_lane_pkmp.h"
#include "ksw2_lane_solo.h"
#include "ksw2_lane_dm.h"
#include "ksw2_lane_extf.h"
#include "ksw2_lane_ssec.h"
#include "ksw2_lane_ssecb.h"
#include "ksw2_lane_extfb.h"

#define K2A_WPB 4          /* wavefronts per workgroup; waves never synchronise with each other */
/* The traceback walk is a chain of dependent loads and a few dozen instructions per step on ONE lane; what it needs is many
 * wavefronts in flight, not many lanes per wavefront.  Walks per wavefront (1..8) so that a launch has about four wavefronts
 * per SIMD: measured on 10 k x 10 k CIGARs, 4096 walks: 8 per wavefront 6.7 ms, 1 per wavefront 5.7 ms; 16384 walks of
 * config 3: 8 -> 1.6 ms, 1 -> 3.3 ms. */
#define K2A_TRACE_PPW_MAX 8
static int k2a_trace_ppw(int nwalks)
{
	const int target_waves = 4096;
	int ppw = (nwalks + target_waves - 1) / target_waves;
	return ppw < 1 ? 1 : ppw > K2A_TRACE_PPW_MAX ? K2A_TRACE_PPW_MAX : ppw;
}

static thread_local char g_err[512] = "";

static int set_err(hipError_t e, const char *what)
{
	if (e == hipSuccess) return 0;
	snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
	return -1;
}
#define CHECK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return set_err(e_, #call); } while (0)

/* The wavefront index, optionally as a scalar.  With one alignment per wavefront the compiler can then keep everything
 * derived from the task (lengths, band, pointers, strip schedule) in SGPRs: 25-40 fewer VGPRs, often one more resident
 * wavefront -- but the work moves to the scalar unit, which the four SIMDs of a CU share, and its latency sits in front
 * of every step.  Measured per kernel on one box, back to back (tools/scripts/ab_variants.sh): splice-aware kernels +23 %,
 * int32 strips of config 3's shape +12 % (2 -> 3 wavefronts), packed (64, 8) +4 %, generation-serial +2 %, solo neutral;
 * packed (64, 16) -1 % (score only) to -4 % (two-piece with traceback), X-drop register window -14 %: those stay vector. */
template<bool UNIFORM>
__device__ __forceinline__ int k2a_wave_id()
{
	return UNIFORM ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) : (int)(threadIdx.x >> 6);
}

template<int G>
__device__ __forceinline__ int k2a_rot1(int v)
{
	/* lane l <- lane l-1 inside its group (lane 0 <- lane G-1) */
	if (G == 64) return __builtin_amdgcn_update_dpp(0, v, 0x13C /* wave_ror:1 */, 0xf, 0xf, false);
	else if (G == 16) return __builtin_amdgcn_update_dpp(0, v, 0x121 /* row_ror:1  */, 0xf, 0xf, false);
	else {
		/* G == 8: two groups per 16-lane row.  row_shr:1 serves lanes 1..7 and 9..15; lanes 0 and 8 take lanes 7 and 15,
		 * which row_ror:9 delivers (lane l <- lane (l-9) mod 16). */
		const int shifted = __builtin_amdgcn_update_dpp(0, v, 0x111 /* row_shr:1 */, 0xf, 0xf, false);
		const int wrapped = __builtin_amdgcn_update_dpp(0, v, 0x129 /* row_ror:9 */, 0xf, 0xf, false);
		return (threadIdx.x & 7) == 0 ? wrapped : shifted;
	}
}

template<int G, int C, bool DUAL, int MODE>
__global__ void __launch_bounds__(64 * K2A_WPB)
k2a_fill_kernel(const K2aScoring sc, const K2aPair *__restrict__ pairs, const uint32_t *__restrict__ order, int ntasks,
                const uint8_t *__restrict__ seq, uint8_t *__restrict__ tb, K2aResult *__restrict__ res)
{
	constexpr int NG = 64 / G;
	typedef K2aLane<G, C, DUAL, MODE> Lane;
	__shared__ K2aBook book[K2A_WPB][NG];
	__shared__ uint32_t tabs[16];                     /* [0..4] row profiles, [8..12] scores against the query wildcard */
	__shared__ int rowbuf[K2A_WPB][NG][3 * C];        /* strip epilogue staging */
	__shared__ int8_t mtab[K2A_MAXM * K2A_MAXM];      /* m > 5: the whole matrix, one byte per residue pair */
	if (threadIdx.x < 5) { tabs[threadIdx.x] = sc.prof[threadIdx.x]; tabs[8 + threadIdx.x] = (uint32_t)sc.colw[threadIdx.x]; }
	if (sc.m > 5) for (int x = threadIdx.x; x < sc.m * sc.m; x += blockDim.x) mtab[x] = sc.mat[x];
	__syncthreads();

	const int lane = threadIdx.x & 63, wave = k2a_wave_id<true>();
	const int grp = lane / G, gl = lane % G;
	const int task = (blockIdx.x * K2A_WPB + wave) * NG + grp;
	const bool valid = task < ntasks;
	const uint32_t pi = order[valid ? task : 0];
	const K2aPair pr = pairs[pi];

	Lane L;
	L.setup(pr, seq, gl, valid);
	K2aBook *bk = &book[wave][grp];
	if (gl == 0) k2a_book_reset(bk);
	__builtin_amdgcn_wave_barrier();

	const int klast = L.last_step();                  /* group-uniform; -1 for an idle group */
	int kmax = __builtin_amdgcn_readfirstlane(klast);
#pragma unroll
	for (int g = 1; g < NG; ++g) kmax = max(kmax, __builtin_amdgcn_readlane(klast, g * G));

	const size_t tbsteps = (size_t)(klast + 1);
	uint8_t *tbp = tb + pr.tb_off;
	bool gdone = !valid;
	L.qb = L.next_query_code(-1);
	/* the first strip of an alignment reads the virtual row -1 instead of a neighbour lane: steps 0..ktop only */
	const int ktop1 = valid ? min(pr.qlen - 1, min(C - 1, pr.tlen - 1) + pr.w) : -1;
	int ktop = __builtin_amdgcn_readfirstlane(ktop1);
#pragma unroll
	for (int g = 1; g < NG; ++g) ktop = max(ktop, __builtin_amdgcn_readlane(ktop1, g * G));
	/* only a Z-drop can end a group before its last step */
	const bool zany = __builtin_amdgcn_ballot_w64(valid && pr.zdrop >= 0) != 0;

	for (int k = 0; k <= kmax; ++k) {
		int hin = k2a_rot1<G>(L.hout);
		int ein = k2a_rot1<G>(L.eout);
		int e2in = DUAL ? k2a_rot1<G>(L.e2out) : 0;

		const bool ninit = L.need_init(k);
		if (__builtin_amdgcn_ballot_w64(ninit) != 0) {
			if (ninit) L.template do_init<true>(sc, tabs);      /* uses hu_prev = what arrived one step ago */
		}
		L.hu_prev = hin;
		const int qnext = L.next_query_code(k);
		if (k <= ktop) L.top_inputs(sc, k, hin, ein, e2in);

		uint32_t tw[Lane::TBWORDS];
		const bool wild = sc.m > 5 || __builtin_amdgcn_ballot_w64(L.qb >= 4) != 0;
		const bool live = L.step(sc, tabs + 8, mtab, wild, k, hin, ein, e2in, tw);
		if (MODE != K2A_MODE_SCORE) {
			if (live) {
				uint32_t *dst = (uint32_t*)(tbp + k2a_tb_word((size_t)k, gl, tbsteps, G, Lane::TBWORDS * 4));
				if (Lane::TBWORDS == 1) dst[0] = tw[0];
				else if (Lane::TBWORDS == 2) *(uint2*)dst = make_uint2(tw[0], tw[1]);
				else {
#pragma unroll
					for (int x = 0; x < Lane::TBWORDS; x += 4)
						*(uint4*)(dst + x) = make_uint4(tw[x], tw[x + 1 < Lane::TBWORDS ? x + 1 : x], tw[x + 2 < Lane::TBWORDS ? x + 2 : x],
						                               tw[x + 3 < Lane::TBWORDS ? x + 3 : x]);
				}
			}
		}

		const bool nfin = L.need_fin(k);
		if (__builtin_amdgcn_ballot_w64(nfin) != 0) {
			if (nfin) L.do_fin(sc, bk, pr.zdrop, rowbuf[wave][grp]);
			__builtin_amdgcn_wave_barrier();
			if (bk->dropped) gdone = true;
		}
		L.qb = qnext;
		if (zany && __builtin_amdgcn_ballot_w64(!(gdone || k >= klast)) == 0) break;
	}
	__builtin_amdgcn_wave_barrier();
	if (valid && gl == 0) {
		const K2aBook b = *bk;
		k2a_finish(pr, b, &res[pi]);
	}
}

/* Traceback words leave the wavefront as whole cache lines.  A lane's words of consecutive steps are contiguous in the
 * lane-major block (k2a_tb_word), but written one 16 / 32-byte store per lane and step they reach HBM as partial-line
 * writes: config 3 moved 1.9 x its algorithmic bytes that way (profiles/r1g_cfg3_pmc.json).  Instead every lane parks its
 * WB-byte word of a step in LDS ([slot][lane], lane index swizzled by the slot so that the transposed read spreads over the
 * banks), and every NS steps the wavefront writes out each lane's NS * WB contiguous bytes with 16-byte pieces of 64 lanes
 * side by side: every store instruction covers whole 64- or 128-byte segments.  Lane runs are padded to K2A_TB_PAD steps
 * (ksw2_types.h), so a block never crosses into the next lane's run; slots of steps that were not executed hold stale
 * words that no walk ever visits.  Lanes that were idle for the whole block are skipped (`livemask`). */
template<int WB, int NS>
struct K2aTbStage {
	enum { PIECES = WB / 16, BLOCK = NS * WB, CPB = BLOCK / 16, NI = NS * PIECES, WORDS = NS * 64 * PIECES };   /* uint4 per wavefront */
	static_assert(WB == 16 || WB == 32, "16- or 32-byte lane-step words");
	static_assert(K2A_TB_PAD % NS == 0 && 64 % CPB == 0, "blocks tile the padded runs");
	uint4 *buf;                       /* this wavefront's [NS][64][PIECES] */
	unsigned long long *gbase;        /* per lane: global address of its run (byte address of word (lane, 0)) */
	uint64_t livemask;
	int lane;

	__device__ __forceinline__ void init(uint4 *buf_, unsigned long long *gb, int lane_, uint8_t *run)
	{
		buf = buf_; gbase = gb; lane = lane_; livemask = 0;
		gbase[lane] = (unsigned long long)run;
		__builtin_amdgcn_wave_barrier();
	}
	__device__ __forceinline__ void put(int k, const uint32_t *tw, bool live)
	{
		const int s = k & (NS - 1);
		uint4 *d = buf + (s * 64 + (lane ^ s)) * PIECES;
#pragma unroll
		for (int x = 0; x < PIECES; ++x) d[x] = make_uint4(tw[4 * x], tw[4 * x + 1], tw[4 * x + 2], tw[4 * x + 3]);
		livemask |= __builtin_amdgcn_ballot_w64(live);
	}
	/* write out the block of steps [k0, k0 + NS) */
	__device__ __forceinline__ void flush(int k0)
	{
		__builtin_amdgcn_wave_barrier();
		if (livemask != 0) {
#pragma unroll
			for (int x = 0; x < NI; ++x) {
				const int id = x * 64 + lane, L = id / CPB, c = id % CPB;      /* source lane, 16-byte piece of its block */
				const int s = c / PIECES, piece = c % PIECES;
				const uint4 v = buf[(s * 64 + (L ^ s)) * PIECES + piece];
				if ((livemask >> L) & 1) *(uint4*)(gbase[L] + (size_t)k0 * WB + (size_t)c * 16) = v;
			}
		}
		livemask = 0;
		__builtin_amdgcn_wave_barrier();
	}
	__device__ __forceinline__ void step_done(int k) { if ((k & (NS - 1)) == NS - 1) flush(k - (NS - 1)); }
	__device__ __forceinline__ void finish(int kdone) { if (kdone >= 0 && (kdone & (NS - 1)) != NS - 1) flush(kdone & ~(NS - 1)); }   /* kdone = last executed step */
};
#define K2A_PK_TB_NS(WB, LDSROW) ((WB) == 32 && (LDSROW) == 1 ? 2 : 8)      /* row state in LDS: two wavefronts per SIMD need the room */

/* Streamed launches (K2aQueueDesc, ksw2_types.h): before a wavefront of a QUEUE build starts its wavefront-task it waits until the
 * upload pieces the task's sequences lie in have landed: word 0 of the plan's watermark block, written by DMA copies that the upload
 * stream orders behind each piece, polled with system-scope loads between s_sleep's.  Wave-uniform; false = do not run (the launch
 * was aborted).  The wait is bounded: past qd->timeout_ticks (100 MHz) the wavefront raises qd->abort and every wavefront that
 * starts later leaves at once -- a kernel of this library never spins on data that may not come.  Tasks are dispatched in grid
 * order, longest first, which for a batch in its arena's order is the order of the pieces; a wavefront that waits holds its slot,
 * but the pieces arrive whatever the CUs do (the DMA engines move them), so the launch always makes progress.
 * (Round 4 first ran these launches as persistent loops over a task counter: the loop cost the kernels 6-40 registers -- the
 * headline's a resident wavefront, 4 830 -> 4 610 GCUPS -- for nothing a wait in front of the body does not give.) */
/* 2-bit wire format (ksw2_lane.h): arena bytes [b0, b1) -- whole pairs of `stride` bytes -- out of the upload, by `nl` lanes of which
 * this is lane `l`: sixteen codes per lane and round, then the pairs' escape entries (codes above 3: rare) on top of what was expanded */
__device__ __forceinline__ void k2a_wire2_task(const uint8_t *__restrict__ src8, uint8_t *__restrict__ dst8, uint32_t b0, uint32_t b1, uint32_t stride, int l, int nl)
{
	const uint32_t *src = (const uint32_t*)(src8 + (b0 >> 2));
	uint4 *dst = (uint4*)(dst8 + b0);
	for (uint32_t x = (uint32_t)l; x < (b1 - b0) >> 4; x += (uint32_t)nl) {
		uint32_t o[4];
		k2a_wire2_expand(src[x], o);
		dst[x] = make_uint4(o[0], o[1], o[2], o[3]);
	}
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");      /* the expanded bytes before the escapes' */
	const uint32_t np = (b1 - b0) / stride;
	for (uint32_t idx = (uint32_t)l; idx < np * 8u; idx += (uint32_t)nl) {
		const uint32_t pair = idx >> 3, e = idx & 7u;
		if (e >= K2A_WIRE2_ESC) continue;
		const uint32_t base = b0 + pair * stride;
		const uint32_t ent = *(const uint32_t*)(src8 + ((base + stride) >> 2) - K2A_WIRE2_SLOT + 4 * e);
		if (ent == 0) continue;
		uint8_t *at = dst8 + base + (ent & 0xfffffu);
		const uint32_t len = (ent >> 20) & 0xffu, code = ent >> 28;
		for (uint32_t y = 0; y < len; ++y) at[y] = (uint8_t)code;
	}
}

__device__ __forceinline__ bool k2a_queue_wait(K2aQueueDesc *qd, int wt)
{
	if ((uint32_t)wt >= qd->nwt) return true;                   /* (an idle wavefront of the last workgroup) */
	if (__hip_atomic_load(&qd->abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return false;
	const uint32_t need = qd->need[wt];
	if (need) {
		const uint32_t *wm = qd->wm;
		const uint64_t t0 = wall_clock64(), limit = qd->timeout_ticks;
		for (int it = 0; __hip_atomic_load(wm, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < need; ++it) {
			if (__hip_atomic_load(&qd->abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return false;
			if (wall_clock64() - t0 > limit) { __hip_atomic_store(&qd->abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return false; }
			if (it < 16) __builtin_amdgcn_s_sleep(16); else __builtin_amdgcn_s_sleep(127);      /* ~0.4 us, then ~3.4 us */
		}
	}
	/* ONE acquire behind the poll, system scope: the pieces were written by the DMA engines while this kernel was already running,
	 * and without it the sequence loads that follow are served stale lines (MI355X_MICROARCH.md, "Consumer, always: ONE relaxed
	 * poll -> ONE acquire -> plain loads"; found by the fuzz script: batches whose data lands after the launch read the PREVIOUS
	 * plan's bytes from the recycled arena -- which a test that runs the same batch twice never sees).  Once per wavefront-task. */
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
	/* 4-bit wire format (uniform plans): this wavefront-task's pairs from the upload into the arena, eight codes per lane and round;
	 * then the arena bytes this wavefront wrote are what it reads (release / acquire at agent scope: the L1 may hold lines of the
	 * recycled arena that a neighbour's look past its own sequences brought in). */
	if (qd->unp_bytes) {
		const uint32_t b0 = (uint32_t)wt * qd->unp_bytes, b1 = min(b0 + qd->unp_bytes, qd->unp_total);
		if ((qd->unp_fmt >> 30) == 2u) k2a_wire2_task(qd->unp_src, qd->unp_dst, b0, b1, qd->unp_fmt & 0x3fffffffu, (int)(threadIdx.x & 63), 64);
		else {
			const uint32_t *src = (const uint32_t*)(qd->unp_src + (b0 >> 1));
			uint2 *dst = (uint2*)(qd->unp_dst + b0);
			for (uint32_t x = threadIdx.x & 63; x < (b1 - b0) >> 3; x += 64) {
				uint32_t lo, hi;
				k2a_wire4_expand(src[x], lo, hi);
				dst[x] = make_uint2(lo, hi);
			}
		}
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
	}
	if ((threadIdx.x & 63) == 0) __hip_atomic_fetch_add(&qd->next, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      /* tasks started: what the host checks a run by */
	return true;
}

/* The column profiles of the packed kernels (K2aScoring.cp, ksw2_lane_pk.h) in LDS, one copy per wavefront: wavefronts of these
 * kernels never synchronise with each other (and wavefronts of a streamed launch may leave before others start). */
__device__ __forceinline__ void k2a_cptab_fill(const K2aScoring &sc, uint32_t *tab, int lane)
{
	uint32_t v = sc.cp[0];
#pragma unroll
	for (int x = 1; x < 8; ++x) v = lane == x ? sc.cp[x] : v;
	if (lane < 8) tab[lane] = v;
	__builtin_amdgcn_wave_barrier();
}

/* Target wildcards in the packed kernels (K2aScoring.pk_tn1, round 6).  A wavefront-task first LOOKS at its targets (k2a_scan_codes:
 * a few microseconds against a task of milliseconds; unscanned arenas -- flat, streamed, uniform plans -- need no host pass for it) and
 * then runs ONE of two builds of the kernel body: the plain one, or the TN one, whose lanes turn a wildcard row's selector into
 * "penalty 0" and take the row's constant off its candidate in a branch that only steps with such a row in the wavefront enter
 * (K2aLanePk<.., TN>).  That branch inside the plain build cost every batch 1-3 % (profiles/r6_ab_tn.txt); here the plain build is
 * the code of round 5, and only wavefronts that hold a wildcard pay.  K2A_SYNC_WN: "some lane of this wavefront holds a wildcard row",
 * refreshed where strips start and end.
 * k2a_scan_codes: the codes >= 4 among target bytes [0, n) as seen by lane gl of a group of G: bit 0 = the wildcard (4), bit 1 = a code
 * above 4 (reported like before: K2aResult.pad[0], the host re-runs the pair). */
#define K2A_SYNC_WN(L) do { (L).wn = __builtin_amdgcn_ballot_w64((L).hasn != 0) != 0; } while (0)
template<int G>
__device__ __forceinline__ uint32_t k2a_scan_codes(const uint8_t *__restrict__ t, int n, int gl)
{
	uint32_t acc = 0, hi = 0;
	/* sixteen bytes per lane and round, the four loads in flight together (unaligned dword loads, as everywhere; the arena is readable
	 * past a sequence's end -- what lies there is masked off) */
	for (int x = gl * 16; x < n; x += G * 16) {
		uint32_t d[4];
#pragma unroll
		for (int y = 0; y < 4; ++y) __builtin_memcpy(&d[y], t + x + 4 * y, 4);
#pragma unroll
		for (int y = 0; y < 4; ++y) {
			const int left = n - (x + 4 * y);                  /* bytes of this dword inside the target */
			const uint32_t v = left >= 4 ? d[y] : left > 0 ? d[y] & ((1u << (8 * left)) - 1u) : 0u;
			acc |= v;
			if (v & 0x04040404u) hi |= k2a_codes_above4(v);
		}
	}
	return ((acc & 0x04040404u) ? 1u : 0u) | ((hi | (acc & 0xf8f8f8f8u)) ? 2u : 0u);
}

/* Packed-int16 resident fill: two same-shape alignments per lane group (ksw2_lane_pk.h).  k2a_fill_pk_body = one wavefront-task, in the
 * plain build or the TN one (target wildcard rows, see k2a_scan_codes); scan = what the task's look at its targets found (TN: bit 1 per
 * lane = a code above 4 in its group's targets). */
template<int G, int C, bool DUAL, int MODE, bool RB, bool NOMAX, int LDSROW, bool DEFER, bool TN>
__device__ __forceinline__ void
k2a_fill_pk_body(const K2aScoring &sc, const K2aPair *__restrict__ pairs, const uint32_t *__restrict__ order2, int ntasks,
                 const uint8_t *__restrict__ seq, uint8_t *__restrict__ tb, K2aResult *__restrict__ res,
                 int wt, int wave, int lane, int grp, int gl, bool scan_on, uint32_t scan,
                 K2aBook (*book)[64 / G][2], uint32_t *stage_w, uint32_t *lrows, uint32_t (*cptab)[8], uint4 *tbstage, unsigned long long *tbruns)
{
	constexpr int NG = 64 / G;
	typedef K2aLanePk<G, C, DUAL, MODE, RB, NOMAX, LDSROW, DEFER, TN> Lane;
	constexpr int LROW_WORDS = LDSROW == 1 ? K2A_PK_LDSROW_WORDS(C) : LDSROW == 2 ? K2A_PK_LDSCODE_WORDS(C) : 0;
	constexpr int WB = Lane::TBWORDS * 4;
	constexpr bool STAGED = MODE != K2A_MODE_SCORE && (WB == 16 || WB == 32);
	typedef K2aTbStage<STAGED ? WB : 16, K2A_PK_TB_NS(WB, LDSROW)> Stage;

	const int task = wt * NG + grp;
	const bool valid = task < ntasks;
	const uint32_t piA = order2[valid ? 2 * task : 0], piB = order2[valid ? 2 * task + 1 : 0];
	const K2aPair prA = pairs[piA], prB = pairs[piB];
	const int zdropA = prA.zdrop, zdropB = prB.zdrop;
	/* a Z-drop test anywhere in the wavefront selects the sequential strip epilogue for all of it */
	const bool zseq = NOMAX || RB || __builtin_amdgcn_ballot_w64(valid && (zdropA >= 0 || zdropB >= 0)) != 0;   /* NOMAX: books only */

	k2a_cptab_fill(sc, cptab[wave], lane);
	Lane L;
	L.lrow = &lrows[LDSROW ? wave * LROW_WORDS + lane : 0];
	L.setup(prA, prB, seq, gl, valid, cptab[wave]);
	K2aBook *bkA = &book[wave][grp][0], *bkB = &book[wave][grp][1];
	if (gl == 0) { k2a_book_reset(bkA); k2a_book_reset(bkB); }
	__builtin_amdgcn_wave_barrier();

	const int klast = L.last_step();
	int kmax = __builtin_amdgcn_readfirstlane(klast);
#pragma unroll
	for (int g = 1; g < NG; ++g) kmax = max(kmax, __builtin_amdgcn_readlane(klast, g * G));

	/* the first strip of an alignment reads the virtual row -1 instead of a neighbour lane: steps 0..ktop only */
	const int ktop1 = valid ? min(prA.qlen - 1, min(C - 1, prA.tlen - 1) + prA.w) : -1;
	int ktop = __builtin_amdgcn_readfirstlane(ktop1);
#pragma unroll
	for (int g = 1; g < NG; ++g) ktop = max(ktop, __builtin_amdgcn_readlane(ktop1, g * G));

	bool gdone = !valid;
	/* query codes: one unaligned dword per alignment and four steps (K2aLanePk::load_query_group) */
	L.load_query_group(0, L.knext == 0 ? L.koff_next : L.koff, L.qwA, L.qwB);
	const size_t tbsteps = (size_t)(klast + 1);
	uint8_t *tbp = tb + prA.tb_off;
	Stage ST;
	if (STAGED) ST.init(&tbstage[wave * Stage::WORDS], &tbruns[wave * 64], lane, tbp + k2a_tb_word(0, gl, tbsteps, G, WB));
	int kdone = -1;
	/* DEFER: this wavefront's checkpoint block in `tb` (the host gave every pair of the wavefront the same tb_off): the stream
	 * [step][lane] of what each lane receives from above, then one header per group and strip (K2aCkHead).  Lane 0's task is
	 * valid whenever the wavefront has one. */
	const bool ckon = DEFER && __builtin_amdgcn_readfirstlane((int)valid) != 0;
	const size_t ckoff = DEFER ? ((size_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(prA.tb_off >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)prA.tb_off) : 0;
	uint2 *ckst = (uint2*)(tb + ckoff) + lane;
	K2aCkHead *ckhd = (K2aCkHead*)(tb + ckoff + (size_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)prA.bnd_off) * K2A_CK_STEP_BYTES) +
	                  (size_t)grp * (uint32_t)__builtin_amdgcn_readfirstlane((int)prA.cig_off);

	/* Steps in groups of four: the query dwords of the NEXT group are asked for at the top of a group and taken over below its
	 * last step, four steps in flight (K2aLanePk::load_query_group).  The inner loop stays rolled. */
	bool stop = false;
	for (int kg = 0; kg <= kmax && !stop; kg += 4) {
	uint32_t qpa, qpb;
	L.load_query_group(kg + 4, L.knext <= kg + 4 ? L.koff_next : L.koff, qpa, qpb);
	const int kge = min(kg + 3, kmax);
#pragma nounroll
	for (int k = kg; k <= kge; ++k) {
		k2a_pk hin = (k2a_pk)k2a_rot1<G>((int)L.hout);
		k2a_pk ein = (k2a_pk)k2a_rot1<G>((int)L.eout);
		k2a_pk e2in = DUAL ? (k2a_pk)k2a_rot1<G>((int)L.e2out) : 0u;

		const bool ninit = L.need_init(k);
		if (__builtin_amdgcn_ballot_w64(ninit) != 0) {
			const int bsA = RB ? k2a_rot1<G>(L.baseA) : 0, bsB = RB ? k2a_rot1<G>(L.baseB) : 0;
			if (ninit) {
				L.do_init(sc, bsA, bsB);                          /* uses hu_prev = what arrived one step ago */
				if (k & 3) L.reload_query_group(k);                /* the group was fetched under the previous strip's offset */
				if (ckon) { K2aCkHead h; h.baseA = L.baseA; h.baseB = L.baseB; h.hd0 = L.hd0; h.pad = 0; ckhd[L.S] = h; }
			}
			if (TN) K2A_SYNC_WN(L);
		}
		L.hu_prev = hin;
		if (RB) { hin = k2a_pk_add(hin, L.delta); ein = k2a_pk_add(ein, L.delta); if (DUAL) e2in = k2a_pk_add(e2in, L.delta); }
		L.set_qb(Lane::query_pick(L.qwA, L.qwB, k & 3));      /* the codes and their column profiles (one LDS look-up per alignment) */
		if (k <= ktop) L.top_inputs(sc, k, hin, ein, e2in);

		if (ckon) ckst[(size_t)k * 64] = make_uint2(hin, ein);      /* 512 contiguous bytes per wavefront and step */
		uint32_t tw[Lane::TBWORDS];
		const bool live = L.step(sc, k, hin, ein, e2in, tw);
		if (STAGED) { ST.put(k, tw, live); ST.step_done(k); kdone = k; }
		else if (MODE != K2A_MODE_SCORE) {
			if (live) {
				uint32_t *dst = (uint32_t*)(tbp + k2a_tb_word((size_t)k, gl, tbsteps, G, Lane::TBWORDS * 4));
#pragma unroll
				for (int x = 0; x + 3 < Lane::TBWORDS; x += 4) *(uint4*)(dst + x) = make_uint4(tw[x], tw[x + 1], tw[x + 2], tw[x + 3]);
				if (Lane::TBWORDS & 2) *(uint2*)(dst + (Lane::TBWORDS & ~3)) = make_uint2(tw[Lane::TBWORDS & ~3], tw[(Lane::TBWORDS & ~3) + 1]);
			}
		}

		const bool nfin = L.need_fin(k);
		const uint64_t finmask = __builtin_amdgcn_ballot_w64(nfin);
		if (finmask != 0) {
			uint32_t *rowbuf = &stage_w[grp * K2A_PK_STAGE(C)];
			if (NOMAX) {
				if (nfin) L.fin_score_only(sc, bkA, bkB);
			} else if (zseq) {
				/* most strips fold into the books from registers; the rest take the row-by-row scan */
				const bool slow = nfin && !L.fin_fast(sc, bkA, bkB, zdropA, zdropB);
				if (__builtin_amdgcn_ballot_w64(slow) != 0) {
					if (slow) { L.stage_rows(rowbuf); L.do_fin_seq(sc, bkA, bkB, zdropA, zdropB, rowbuf); }
					__builtin_amdgcn_wave_barrier();
					if (!DEFER && bkA->dropped && bkB->dropped) gdone = true;      /* (DEFER: "dropped" there = book frozen; the third pass needs the checkpoints of the strips that follow) */
				}
			} else {
				/* at most one strip per group ends at a step; all lanes of that group share its rows */
				const bool gfin = ((finmask >> (grp * G)) & (G == 64 ? ~0ull : (1ull << (G & 63)) - 1)) != 0;
				if (nfin) L.stage_rows(rowbuf);
				__builtin_amdgcn_wave_barrier();
				if (gfin) L.fin_local_rows(sc, rowbuf);
				if (nfin) L.end_strip();
				__builtin_amdgcn_wave_barrier();
			}
			if (TN && L.wn) K2A_SYNC_WN(L);
		}
		if (zseq && __builtin_amdgcn_ballot_w64(!(gdone || k >= klast)) == 0) { stop = true; break; }   /* only a Z-drop ends a group early */
	}
	L.qwA = qpa; L.qwB = qpb;
	}
	if (STAGED) ST.finish(kdone);
	__builtin_amdgcn_wave_barrier();
	if (!zseq) {
		/* merge the lane-local bests of each group; the lane that finished the last target row adds mte / score */
		uint32_t *loc = &stage_w[0];
		loc[lane * 5 + 0] = L.lmax; loc[lane * 5 + 1] = L.lmax_t; loc[lane * 5 + 2] = L.lmax_q;
		loc[lane * 5 + 3] = L.lmqe; loc[lane * 5 + 4] = L.lmqe_t;
		__builtin_amdgcn_wave_barrier();
		if (valid && gl == 0) {
			k2a_merge_local(loc + grp * G * 5, G, 0, bkA);
			k2a_merge_local(loc + grp * G * 5, G, 1, bkB);
			bkA->rows = bkB->rows = prA.tlen;
		}
		__builtin_amdgcn_wave_barrier();
		if (valid && prA.tlen == prA.tlen_full && gl == ((prA.tlen_full - 1) % C) % G) {   /* the lane that took the last row */
			const bool reach = prA.tlen_full - 1 + prA.w >= prA.qlen - 1;
			bkA->mte = k2a_pk_lo(L.last_m); bkA->mte_q = k2a_pk_lo(L.last_j);
			bkB->mte = k2a_pk_hi(L.last_m); bkB->mte_q = k2a_pk_hi(L.last_j);
			if (reach) { bkA->score = k2a_pk_lo(L.last_h); bkB->score = k2a_pk_hi(L.last_h); }
		}
		__builtin_amdgcn_wave_barrier();
	}
	/* a code >= 4 among the bytes this group read: only an unscanned (flat) plan can get here with one; the host re-runs the pair */
	const uint64_t sawmask = __builtin_amdgcn_ballot_w64(valid && (scan_on ? (scan & 2u) != 0 : L.saw_wildcard()));
	const bool gsaw = ((sawmask >> (grp * G)) & (G == 64 ? ~0ull : (1ull << (G & 63)) - 1)) != 0;
	if (valid && gl == 0) {
		const K2aBook a = *bkA, b = *bkB;
		k2a_finish(prA, a, &res[piA]);
		if (piB != piA) k2a_finish(prB, b, &res[piB]);
		if (gsaw) { res[piA].pad[0] = 1; res[piB].pad[0] = 1; }
	}
}

template<int G, int C, bool DUAL, int MODE, bool RB, bool NOMAX, int LDSROW = 0, bool DEFER = false, bool QUEUE = false>      /* LDSROW: 1 = row state in LDS, 2 = only the target-code planes; DEFER: K2aLanePk; QUEUE: streamed launches */
__global__ void __launch_bounds__(64 * K2A_WPB, LDSROW == 2 ? (G == 16 ? 4 : 3) : LDSROW ? 2 : 1)      /* no floor elsewhere: capping the register form of the score-only kernels at 168 VGPRs spills and is 18 % slower */
k2a_fill_pk_kernel(const K2aScoring sc, const K2aPair *__restrict__ pairs, const uint32_t *__restrict__ order2, int ntasks,
                   const uint8_t *__restrict__ seq, uint8_t *__restrict__ tb, K2aResult *__restrict__ res, K2aQueueDesc *qd)
{
	constexpr int NG = 64 / G;
	typedef K2aLanePk<G, C, DUAL, MODE, RB, NOMAX, LDSROW, DEFER, false> Lane;      /* (sizes only: both builds of the body have the same) */
	__shared__ K2aBook book[K2A_WPB][NG][2];
	__shared__ uint32_t stage[K2A_WPB][(NG * K2A_PK_STAGE(C) > 64 * 5) ? NG * K2A_PK_STAGE(C) : 64 * 5];   /* row buffers / final lane records */

	const int lane = threadIdx.x & 63, wave = k2a_wave_id<(C <= 8 || LDSROW || (NOMAX && DUAL && MODE != K2A_MODE_SCORE))>();   /* 16 rows, two-piece, traceback: part of what gets those kernels to two wavefronts */
	const int grp = lane / G, gl = lane % G;
	/* one wavefront-task (NG tasks) per wavefront, by position in the grid; the QUEUE builds (streamed launches) first wait for the
	 * task's inputs (k2a_queue_wait) */
	const int wt = blockIdx.x * K2A_WPB + wave;
	if (DEFER && blockIdx.x == 0 && threadIdx.x == 0) {      /* the list of frozen books (k2a_argmax_kernel fills it, k2a_zscan_kernel works it off): empty */
		uint32_t *zlist = (uint32_t*)(tb + pairs[order2[0]].tb_off) - K2A_ZLIST_WORDS(ntasks);
		__hip_atomic_store(&zlist[0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	}
	if (QUEUE && !k2a_queue_wait(qd, wt)) return;
	constexpr int LROW_WORDS = LDSROW == 1 ? K2A_PK_LDSROW_WORDS(C) : LDSROW == 2 ? K2A_PK_LDSCODE_WORDS(C) : 0;
	__shared__ uint32_t lrows[LDSROW ? K2A_WPB * LROW_WORDS : 1];   /* per-row maxima / arg-max / target codes of the LDSROW classes */
	__shared__ uint32_t cptab[K2A_WPB][8];
	constexpr int WB = Lane::TBWORDS * 4;
	constexpr bool STAGED = MODE != K2A_MODE_SCORE && (WB == 16 || WB == 32);
	typedef K2aTbStage<STAGED ? WB : 16, K2A_PK_TB_NS(WB, LDSROW)> Stage;
	__shared__ uint4 tbstage[STAGED ? K2A_WPB * Stage::WORDS : 1];
	__shared__ unsigned long long tbruns[STAGED ? K2A_WPB * 64 : 1];
	/* the task's look at its targets (k2a_scan_codes), then the plain or the TN build of the body */
	const int task = wt * NG + grp;
	uint32_t scan = 0;
	const bool scan_on = sc.pk_tn1 != 0;
	if (scan_on && task < ntasks) {
		const K2aPair pa = pairs[order2[2 * task]], pb = pairs[order2[2 * task + 1]];
		scan = k2a_scan_codes<G>(seq + pa.toff, pa.tlen_full, gl) | k2a_scan_codes<G>(seq + pb.toff, pb.tlen_full, gl);
	}
	{	/* per group: what any of its lanes saw */
		const uint64_t m1 = __builtin_amdgcn_ballot_w64((scan & 1u) != 0), m2 = __builtin_amdgcn_ballot_w64((scan & 2u) != 0);
		const uint64_t gm = (G == 64 ? ~0ull : (1ull << (G & 63)) - 1) << (grp * G);
		scan = ((m1 & gm) ? 1u : 0u) | ((m2 & gm) ? 2u : 0u);
		if (m1 != 0)
			k2a_fill_pk_body<G, C, DUAL, MODE, RB, NOMAX, LDSROW, DEFER, true>(sc, pairs, order2, ntasks, seq, tb, res, wt, wave, lane, grp, gl, scan_on, scan, book, stage[wave], lrows, cptab, tbstage, tbruns);
		else
			k2a_fill_pk_body<G, C, DUAL, MODE, RB, NOMAX, LDSROW, DEFER, false>(sc, pairs, order2, ntasks, seq, tb, res, wt, wave, lane, grp, gl, scan_on, scan, book, stage[wave], lrows, cptab, tbstage, tbruns);
	}
}

/* Second pass of the deferred arg-max (K2aLanePk, DEFER): three jobs per task -- the strip that holds alignment A's maximum,
 * the one that holds B's, and the strip of the last target row (mte_q of both) -- one job per LANE: every lane re-runs its strip
 * with the ordinary exact lane code (arg-max on), its top inputs read back from the checkpoint stream, its bases from the strip's
 * header, and writes the one or two columns that were asked for.  No lane talks to another.  An alignment whose book the fill froze
 * (K2aResult.pad[1]: a Z-drop could not be ruled out without columns) is entered in the class's list for the third pass
 * (k2a_zscan_kernel); its max_q is found here like everybody's. */
template<int G, int C, bool RB>
__global__ void __launch_bounds__(64 * K2A_WPB)
k2a_argmax_kernel(const K2aScoring sc, const K2aPair *__restrict__ pairs, const uint32_t *__restrict__ order2, int ntasks,
                  const uint8_t *__restrict__ seq, uint8_t *__restrict__ ck, K2aResult *__restrict__ res, const K2aQueueDesc *qd)
{
	constexpr int NG = 64 / G;
	/* behind a streamed fill that was abandoned (k2a_queue_wait: abort raised, or wavefront-tasks that never started): the result
	 * records and checkpoint blocks of the tasks the fill never ran hold whatever the recycled buffers held before -- a stale max_t
	 * would index strips megabytes past the task's block.  Nothing to do here then: the host repeats the whole plan unstreamed. */
	if (qd && (__hip_atomic_load(&qd->abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0 ||
	           __hip_atomic_load(&qd->next, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != qd->nwt)) return;
	typedef K2aLanePk<G, C, false, K2A_MODE_SCORE, RB, false, 0, false, true> Lane;      /* (the TN build: a re-run strip may hold a target wildcard row) */
	const int job = blockIdx.x * (64 * K2A_WPB) + threadIdx.x;
	const int task = job / 3, which = job - 3 * task;
	bool go = task < ntasks;
	const bool live = go;
	const uint32_t piA = order2[go ? 2 * task : 0], piB = order2[go ? 2 * task + 1 : 0];
	const K2aPair prA = pairs[piA], prB = pairs[piB];
	const bool inexA = res[piA].pad[1] != 0, inexB = res[piB].pad[1] != 0;
	int row = -1;
	if (go) {
		if (which == 0) row = res[piA].max_t;
		else if (which == 1) row = piB == piA ? -1 : res[piB].max_t;
		else row = (prA.tlen == prA.tlen_full && !(inexA && inexB)) ? prA.tlen_full - 1 : -1;
	}
	go = go && row >= 0;
	const int S = go ? row / C : 0, grp = task % NG;
	__shared__ uint32_t cptab[K2A_WPB][8];
	k2a_cptab_fill(sc, cptab[threadIdx.x >> 6], (int)(threadIdx.x & 63));
	Lane L;
	L.lrow = 0;
	L.setup(prA, prB, seq, S % G, go, cptab[threadIdx.x >> 6]);
	L.Snext = S;
	L.schedule_next();
	const int kbeg = L.knext;
	const uint8_t *blk = ck + prA.tb_off;
	if (go) L.do_init(sc, 0, 0, (const K2aCkHead*)(blk + (size_t)prA.bnd_off * K2A_CK_STEP_BYTES) + (size_t)grp * prA.cig_off + S);
	K2A_SYNC_WN(L);
	const int n = go ? L.kfin - kbeg + 1 : 0;
	const uint2 *st = (const uint2*)blk + (grp * G + S % G);
	for (int t = 0; __builtin_amdgcn_ballot_w64(t < n) != 0; ++t) {
		if (t < n) {
			const int k = kbeg + t;
			const uint2 in = st[(size_t)k * 64];
			const int jc = min(max(k - L.koff, 0), L.qlen - 1);
			L.set_qb(k2a_pair16(L.qa[jc], L.qbp[jc]));
			uint32_t tw[Lane::TBWORDS];
			L.step(sc, k, in.x, in.y, 0u, tw);
		}
	}
	if (go) {
		const int c = row - S * C;
		k2a_pk v = 0;
#pragma unroll
		for (int cc = 0; cc < C; ++cc) if (cc == c) v = L.rmj(cc);
		if (which == 0) res[piA].max_q = (int)(v & 0xffffu);
		else if (which == 1) res[piB].max_q = (int)(v >> 16);
		else {
			if (!inexA) res[piA].mte_q = (int)(v & 0xffffu);
			if (piB != piA && !inexB) res[piB].mte_q = (int)(v >> 16);
		}
	}
	/* frozen books: into the list of the third pass, { task, half } (the list lies in front of the class's first checkpoint block;
	 * the fill kernel zeroed its counter) */
	if (live && ((which == 0 && inexA) || (which == 1 && inexB && piB != piA))) {
		uint32_t *zlist = (uint32_t*)(ck + pairs[order2[0]].tb_off) - K2A_ZLIST_WORDS(ntasks);
		const uint32_t at = __hip_atomic_fetch_add(&zlist[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		zlist[1 + at] = (uint32_t)task * 2u + (uint32_t)which;
	}
}

/* Third pass of the deferred arg-max: the alignments whose book the fill froze at a row where a Z-drop could not be ruled out
 * without the arg-max columns (max - H > zdrop there; no earlier row came that far).  The reference's test (ksw2.h:191-207) also
 * wants j >= max_q and adds |dt - dq| * e to the threshold, and on reads whose tail diverges the best cell of a row wanders off the
 * diagonal of the maximum: the drop comes tens to hundreds of rows AFTER the first row that is zdrop below the maximum (round 5,
 * measured on the simulator: of 44 frozen alignments none dropped at that first row).  So the fill no longer stops such an
 * alignment's wavefront, its checkpoints run on to the last strip, and here a group of 16 lanes takes one frozen alignment: the
 * lanes re-run 16 consecutive strips at once from the frozen row's strip on -- every strip is independent given its checkpoints --
 * with the exact lane code, stage their rows in LDS, and the group's first lane folds them into the book in row order with the
 * reference's own per-row logic (k2a_fin_rows_half = K2aLanePk::do_fin_seq's exact branch): 256 rows per round, until the drop or
 * the last row.  Then the record is rewritten from the book like any fill's (k2a_finish).  Nothing is handed back to the host any
 * more (rounds 3-4: every frozen alignment was run again from the start by the ordinary kernels behind the batch -- 10 k reads of
 * which a fifth drops ran at 0.75 of the rate of reads that do not). */
template<int G, int C, bool RB>
__global__ void __launch_bounds__(64 * K2A_WPB)
k2a_zscan_kernel(const K2aScoring sc, const K2aPair *__restrict__ pairs, const uint32_t *__restrict__ order2, int ntasks,
                 const uint8_t *__restrict__ seq, const uint8_t *__restrict__ ck, K2aResult *__restrict__ res, const K2aQueueDesc *qd)
{
	constexpr int NG = 64 / G, ZG = 16, ZNG = 64 / ZG;
	if (qd && (__hip_atomic_load(&qd->abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0 ||
	           __hip_atomic_load(&qd->next, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != qd->nwt)) return;
	typedef K2aLanePk<G, C, false, K2A_MODE_SCORE, RB, false, 0, false, true> Lane;      /* (the TN build: a re-run strip may hold a target wildcard row) */
	__shared__ uint32_t stage[K2A_WPB][64][K2A_PK_STAGE(C)];
	__shared__ K2aBook book[K2A_WPB][ZNG];
	__shared__ int done[K2A_WPB][ZNG];
	__shared__ uint32_t cptab[K2A_WPB][8];
	const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
	const int zg = lane / ZG, zl = lane % ZG;
	const uint32_t *zlist = (const uint32_t*)(ck + pairs[order2[0]].tb_off) - K2A_ZLIST_WORDS(ntasks);
	const int count = (int)zlist[0];
	const int gi = (blockIdx.x * K2A_WPB + wave) * ZNG + zg;
	const bool active = gi < count;
	if (__builtin_amdgcn_ballot_w64(active) == 0) return;
	k2a_cptab_fill(sc, cptab[wave], lane);
	const uint32_t ent = zlist[1 + (active ? gi : 0)];
	const int task = (int)(ent >> 1), half = (int)(ent & 1u);
	const uint32_t piA = order2[2 * task], piB = order2[2 * task + 1], pi = half ? piB : piA;
	const K2aPair prA = pairs[piA], prB = pairs[piB], pr = half ? prB : prA;
	const int grp = task % NG, nstrips = (prA.tlen + C - 1) / C;
	K2aBook *bk = &book[wave][zg];
	const K2aResult r0 = res[pi];
	if (zl == 0) {
		K2aBook b;
		b.max = r0.max; b.max_t = r0.max_t; b.max_q = r0.max_q; b.mqe = r0.mqe; b.mqe_t = r0.mqe_t; b.mte = r0.mte; b.mte_q = r0.mte_q;
		b.score = r0.score; b.dropped = 0; b.rows = r0.rows_done; b.inexact = 0;
		*bk = b;
		done[wave][zg] = active ? 0 : 1;
	}
	__builtin_amdgcn_wave_barrier();
	const int S1 = active ? max(r0.rows_done - 1, 0) / C : 0;            /* the strip of the row the fill froze the book at */
	const uint8_t *blk = ck + prA.tb_off;
	for (int round = 0; ; ++round) {
		const bool open = done[wave][zg] == 0;
		const int S = S1 + round * ZG + zl;
		const bool go = open && S < nstrips;
		Lane L;
		L.lrow = 0;
		L.setup(prA, prB, seq, S % G, go, cptab[wave]);
		L.Snext = go ? S : 0;
		L.schedule_next();
		const int kbeg = L.knext;
		if (go) L.do_init(sc, 0, 0, (const K2aCkHead*)(blk + (size_t)prA.bnd_off * K2A_CK_STEP_BYTES) + (size_t)grp * prA.cig_off + S);
		K2A_SYNC_WN(L);
		const int n = go ? L.kfin - kbeg + 1 : 0;
		const uint2 *st = (const uint2*)blk + (grp * G + S % G);
		/* the group's lanes walk the fill's steps TOGETHER: at one iteration all of them read the same step of the checkpoint stream,
		 * sixteen neighbouring 8-byte entries (strip S + 1 belonged to the fill's next lane and started C + 1 steps later) -- one
		 * 128-byte line per group instead of one line per lane; lanes whose strip has not begun or is over sit the iteration out */
		const int kfirst = __builtin_amdgcn_readlane(kbeg, 0) * (zg == 0) + __builtin_amdgcn_readlane(kbeg, 16) * (zg == 1) +
		                   __builtin_amdgcn_readlane(kbeg, 32) * (zg == 2) + __builtin_amdgcn_readlane(kbeg, 48) * (zg == 3);      /* first step of the group's first strip */
		const int off = go ? kbeg - kfirst : 0;
		/* what step t needs from memory -- the checkpoint entry and the two query codes -- is asked for one iteration ahead and by every
		 * lane, unconditionally (clamped into the lane's own range; a lane without a strip reads step 0 of its block): two wavefronts
		 * per SIMD do not hide an L2 round trip per step, and hipcc waits right behind a load that sits under a condition */
		/* ... and FOUR steps ahead (round 6): the checkpoint stream is 137 GB of write-once data, i.e. HBM, and one step of look-ahead
		 * left the pass at ~1.5 us per step (9.8 ms of a 105 ms launch when a fifth of the pairs freeze).  Steps in groups of four: the next
		 * group's three loads per step are issued at the top of a group into registers of their own and taken over below its last step
		 * (a value that is still in flight must not be copied: k2a_load_early) */
		uint2 in_c[4]; uint32_t qa_c[4], qb_c[4];
		auto ask = [&](int t, uint2 &in, uint32_t &qa, uint32_t &qb) {
			const int k = n > 0 ? kfirst + min(max(t, off), off + n - 1) : 0, jc = min(max(k - L.koff, 0), L.qlen - 1);
			in = st[(size_t)k * 64]; qa = L.qa[jc]; qb = L.qbp[jc];
		};
#pragma unroll
		for (int y = 0; y < 4; ++y) ask(y, in_c[y], qa_c[y], qb_c[y]);
		for (int tg = 0; __builtin_amdgcn_ballot_w64(tg < off + n) != 0; tg += 4) {
			uint2 in_n[4]; uint32_t qa_n[4], qb_n[4];
#pragma unroll
			for (int y = 0; y < 4; ++y) ask(tg + 4 + y, in_n[y], qa_n[y], qb_n[y]);
#pragma unroll
			for (int y = 0; y < 4; ++y) {
				const int t = tg + y;
				if (t >= off && t < off + n) {
					L.set_qb(k2a_pair16(qa_c[y], qb_c[y]));
					uint32_t tw[Lane::TBWORDS];
					L.step(sc, kfirst + t, in_c[y].x, in_c[y].y, 0u, tw);
				}
			}
#pragma unroll
			for (int y = 0; y < 4; ++y) { in_c[y] = in_n[y]; qa_c[y] = qa_n[y]; qb_c[y] = qb_n[y]; }
		}
		if (go) L.stage_rows(stage[wave][lane]);
		__builtin_amdgcn_wave_barrier();
		if (open && zl == 0) {
			int l = 0;
			for (; l < ZG && S1 + round * ZG + l < nstrips && !bk->dropped; ++l)
				k2a_fin_rows_half<C>(sc, bk, pr.zdrop, stage[wave][zg * ZG + l], half, RB, pr.qlen, pr.tlen, pr.tlen_full, pr.w);
			if (bk->dropped || S1 + (round + 1) * ZG >= nstrips) {
				const K2aBook b = *bk;
				k2a_finish(pr, b, &res[pi]);                                  /* (pad[1] = 0: settled) */
				res[pi].pad[0] = r0.pad[0];                                   /* ... and what the fill reported stays: a wildcard code in the target (unscanned plans: the host re-runs the pair) */
				done[wave][zg] = 1;
			}
		}
		__builtin_amdgcn_wave_barrier();
		if (__builtin_amdgcn_ballot_w64(done[wave][zg] == 0) == 0) break;
	}
}

/* Packed-int16 fill of ONE alignment per wavefront on both register halves (ksw2_lane_solo.h): reads without a partner of
 * identical shape.  The high half's bottom row goes to the next lane's low half (wave_ror:1 + v_alignbit), the low half's
 * bottom row to the lane's own high half one step later. */
template<int C, bool DUAL, int MODE, bool TN>
__device__ __forceinline__ void
k2a_fill_solo_body(const K2aScoring &sc, const K2aPair *__restrict__ pairs, const uint32_t *__restrict__ order, int ntasks,
                   const uint8_t *__restrict__ seq, uint8_t *__restrict__ tb, K2aResult *__restrict__ res, int task, int wave, int lane, bool scan_on, uint32_t scan,
                   K2aBook *book, uint32_t (*stage)[K2A_SOLO_STAGE(C)], uint32_t (*cptab)[8], uint4 *tbstage, unsigned long long *tbruns)
{
	typedef K2aLaneSolo<C, DUAL, MODE, TN> Lane;

	const bool valid = task < ntasks;
	const uint32_t pi = order[valid ? task : 0];
	const K2aPair pr = pairs[pi];

	k2a_cptab_fill(sc, cptab[wave], lane);
	Lane L;
	L.setup(pr, seq, lane, valid, cptab[wave]);
	K2aBook *bk = &book[wave];
	if (lane == 0) k2a_book_reset(bk);
	__builtin_amdgcn_wave_barrier();

	const int klast = __builtin_amdgcn_readfirstlane(L.last_step());
	const int ktop = valid ? min(pr.qlen - 1, min(C - 1, pr.tlen - 1) + pr.w) : -1;
	const size_t tbsteps = k2a_solo_steps<C>(pr.qlen, pr.tlen, pr.w);
	uint8_t *tbp = tb + pr.tb_off;
	/* query codes: one unaligned dword per four steps (K2aLaneSolo::load_query_group) */
	L.load_query_group(0, L.knext == 0 ? L.koff_next : L.koff, L.qw);
	/* traceback words as whole cache lines (K2aTbStage, see the packed kernels): config 5's unique read shapes run here */
	constexpr int WB = Lane::TBWORDS * 4;
	constexpr bool STAGED = MODE != K2A_MODE_SCORE && (WB == 16 || WB == 32);
	typedef K2aTbStage<STAGED ? WB : 16, 8> Stage;
	Stage ST;
	if (STAGED) ST.init(&tbstage[wave * Stage::WORDS], &tbruns[wave * 64], lane, tbp + k2a_tb_word(0, lane, tbsteps, 64, WB));
	int kdone = -1;

	bool stop = false;
	for (int kg = 0; kg <= klast && !stop; kg += 4) {        /* groups of four steps as in k2a_fill_pk_kernel */
	uint32_t qp;
	L.load_query_group(kg + 4, L.knext <= kg + 4 ? L.koff_next : L.koff, qp);
	const int kge = min(kg + 3, klast);
#pragma nounroll
	for (int k = kg; k <= kge; ++k) {
		const k2a_pk rh = (k2a_pk)k2a_rot1<64>((int)L.hout);
		k2a_pk hin = __builtin_amdgcn_alignbit(L.hout, rh, 16);                                   /* { lane above's high half, own low half } */
		k2a_pk ein = __builtin_amdgcn_alignbit(L.eout, (k2a_pk)k2a_rot1<64>((int)L.eout), 16);
		k2a_pk e2in = DUAL ? __builtin_amdgcn_alignbit(L.e2out, (k2a_pk)k2a_rot1<64>((int)L.e2out), 16) : 0u;

		const bool ninit = L.need_init(k);
		if (__builtin_amdgcn_ballot_w64(ninit) != 0) {
			const int bs = k2a_rot1<64>(L.baseB);
			if (ninit) L.do_init(sc, bs);                      /* uses hu_prev = what arrived one step ago; brings its first query group along */
			if (TN) K2A_SYNC_WN(L);
		}
		const bool nhigh = L.need_init_high(k);
		if (__builtin_amdgcn_ballot_w64(nhigh) != 0) {
			if (nhigh) L.start_high(sc);                       /* the high half's base: uses hd0 = what its low half handed over */
		}
		L.hu_prev = rh;
		hin = k2a_pk_add(hin, L.delta); ein = k2a_pk_add(ein, L.delta); if (DUAL) e2in = k2a_pk_add(e2in, L.delta);
		L.advance_query(k & 3);
		if (k <= ktop) L.top_inputs(sc, k, hin, ein, e2in);

		uint32_t tw[Lane::TBWORDS];
		const bool live = L.step(sc, k, hin, ein, e2in, tw);
		if (STAGED) { ST.put(k, tw, live); ST.step_done(k); kdone = k; }
		else if (MODE != K2A_MODE_SCORE) {
			if (live) {
				uint32_t *dst = (uint32_t*)(tbp + k2a_tb_word((size_t)k, lane, tbsteps, 64, Lane::TBWORDS * 4));
#pragma unroll
				for (int x = 0; x + 3 < Lane::TBWORDS; x += 4) *(uint4*)(dst + x) = make_uint4(tw[x], tw[x + 1], tw[x + 2], tw[x + 3]);
				if (Lane::TBWORDS & 2) *(uint2*)(dst + (Lane::TBWORDS & ~3)) = make_uint2(tw[Lane::TBWORDS & ~3], tw[(Lane::TBWORDS & ~3) + 1]);
			}
		}
		const bool nsave = L.need_save(k);
		if (__builtin_amdgcn_ballot_w64(nsave) != 0) {
			if (nsave) L.save_low();
		}
		const bool nfin = L.need_fin(k);
		if (__builtin_amdgcn_ballot_w64(nfin) != 0) {
			/* one double strip ends per step at most; most fold into the book from registers, the rest take the row scan */
			const bool slow = nfin && !L.fin_fast(sc, bk, pr.zdrop);
			if (__builtin_amdgcn_ballot_w64(slow) != 0) {
				if (slow) { L.stage_rows(stage[wave]); L.do_fin_seq(sc, bk, pr.zdrop, stage[wave]); }
				__builtin_amdgcn_wave_barrier();
				if (bk->dropped) { stop = true; break; }
			}
			if (TN && L.wn) K2A_SYNC_WN(L);
		}
	}
	L.qw = qp;
	}
	if (STAGED) ST.finish(kdone);
	__builtin_amdgcn_wave_barrier();
	/* a code >= 4 among the bytes this wavefront read: only an unscanned (flat) plan can get here with one; the host re-runs the pair */
	const bool saw = scan_on ? (scan & 2u) != 0 : __builtin_amdgcn_ballot_w64(valid && L.saw_wildcard()) != 0;
	if (valid && lane == 0) {
		const K2aBook b = *bk;
		k2a_finish(pr, b, &res[pi]);
		if (saw) res[pi].pad[0] = 1;
	}
}

/* the wavefront-task looks at its target first (k2a_scan_codes) and takes the plain or the TN build of the body */
template<int C, bool DUAL, int MODE>
__global__ void __launch_bounds__(64 * K2A_WPB)
k2a_fill_solo_kernel(const K2aScoring sc, const K2aPair *__restrict__ pairs, const uint32_t *__restrict__ order, int ntasks,
                     const uint8_t *__restrict__ seq, uint8_t *__restrict__ tb, K2aResult *__restrict__ res)
{
	typedef K2aLaneSolo<C, DUAL, MODE, false> Lane;      /* (sizes only) */
	constexpr int WB = Lane::TBWORDS * 4;
	constexpr bool STAGED = MODE != K2A_MODE_SCORE && (WB == 16 || WB == 32);
	typedef K2aTbStage<STAGED ? WB : 16, 8> Stage;
	__shared__ K2aBook book[K2A_WPB];
	__shared__ uint32_t stage[K2A_WPB][K2A_SOLO_STAGE(C)];
	__shared__ uint32_t cptab[K2A_WPB][8];
	__shared__ uint4 tbstage[STAGED ? K2A_WPB * Stage::WORDS : 1];
	__shared__ unsigned long long tbruns[STAGED ? K2A_WPB * 64 : 1];
	const int lane = threadIdx.x & 63, wave = k2a_wave_id<true>();
	const int task = blockIdx.x * K2A_WPB + wave;
	const bool scan_on = sc.pk_tn1 != 0;
	uint32_t scan = 0;
	if (scan_on && task < ntasks) {
		const K2aPair pr = pairs[order[task]];
		scan = k2a_scan_codes<64>(seq + pr.toff, pr.tlen_full, lane);
	}
	const bool any4 = __builtin_amdgcn_ballot_w64((scan & 1u) != 0) != 0;
	scan = (any4 ? 1u : 0u) | (__builtin_amdgcn_ballot_w64((scan & 2u) != 0) != 0 ? 2u : 0u);
	if (any4) k2a_fill_solo_body<C, DUAL, MODE, true>(sc, pairs, order, ntasks, seq, tb, res, task, wave, lane, scan_on, scan, book, stage, cptab, tbstage, tbruns);
	else k2a_fill_solo_body<C, DUAL, MODE, false>(sc, pairs, order, ntasks, seq, tb, res, task, wave, lane, scan_on, scan, book, stage, cptab, tbstage, tbruns);
}

template<int C>
__global__ void __launch_bounds__(64)
k2a_trace_solo_kernel(const K2aPair *__restrict__ pairs, const uint32_t *__restrict__ order, int ntasks,
                      const uint8_t *__restrict__ tb, K2aResult *__restrict__ res, uint32_t *__restrict__ cig, int ppw)
{
	if ((int)threadIdx.x >= ppw) return;
	const int t = blockIdx.x * ppw + threadIdx.x;
	if (t >= ntasks) return;
	const uint32_t pi = order[t];
	const K2aPair pr = pairs[pi];
	const int ti = res[pi].ti, tj = res[pi].tj;
	int n = 0;
	if (ti >= 0 && tj >= 0) n = k2a_trace_solo<C>(tb + pr.tb_off, ti, tj, cig + pr.cig_off, pr.qlen, pr.tlen, pr.w);
	res[pi].n_cigar = n;
}

/* Generation-serial fill (class K2A_CFG_MP): bands too wide to keep resident.  One alignment per wavefront;
 * generation g = rows g*G*C .. (g+1)*G*C-1 over all their in-band columns.  Lane G-1 streams its bottom row
 * (H, E[, E~]) to the boundary buffer, lane 0 of the next generation streams it back in (L1-bypassing loads,
 * the producer wrote them thousands of steps earlier from this same wavefront). */
template<int G, int C, bool DUAL, int MODE, bool LROW = false>
__global__ void __launch_bounds__(64 * K2A_WPB, LROW ? 2 : 1)
k2a_fill_mp_kernel(const K2aScoring sc, const K2aPair *__restrict__ pairs, const uint32_t *__restrict__ order, int ntasks,
                   const uint8_t *__restrict__ seq, uint8_t *__restrict__ tb, int32_t *bnd, K2aResult *__restrict__ res)
{
	static_assert(G == 64, "one alignment per wavefront");
	/* LROW (single-gap traceback): 264 registers with the row maxima in VGPRs (one wavefront per SIMD), 8 KiB of LDS per
	 * wavefront for them instead makes it two; chosen by the launcher like the packed kernels' LDS form */
	typedef K2aLane<G, C, DUAL, MODE, LROW> Lane;
	__shared__ K2aBook book[K2A_WPB];
	__shared__ int rowbuf[K2A_WPB][3 * C];
	__shared__ int lrows[LROW ? K2A_WPB * K2A_LROW_WORDS(C) : 1];
	__shared__ uint32_t tabs[16];
	__shared__ int8_t mtab[K2A_MAXM * K2A_MAXM];      /* m > 5: the whole matrix, one byte per residue pair */
	if (threadIdx.x < 5) { tabs[threadIdx.x] = sc.prof[threadIdx.x]; tabs[8 + threadIdx.x] = (uint32_t)sc.colw[threadIdx.x]; }
	if (sc.m > 5) for (int x = threadIdx.x; x < sc.m * sc.m; x += blockDim.x) mtab[x] = sc.mat[x];
	__syncthreads();

	const int gl = threadIdx.x & 63, wave = k2a_wave_id<true>();
	const int task = blockIdx.x * K2A_WPB + wave;
	const bool valid = task < ntasks;
	const uint32_t pi = order[valid ? task : 0];
	const K2aPair pr = pairs[pi];

	Lane L;
	L.lrow = &lrows[LROW ? wave * K2A_LROW_WORDS(C) + gl : 0];
	L.setup(pr, seq, gl, valid);
	K2aBook *bk = &book[wave];
	if (gl == 0) k2a_book_reset(bk);
	__builtin_amdgcn_wave_barrier();

	int32_t *Bh = bnd + pr.bnd_off, *Be = Bh + pr.qlen, *Be2 = Be + pr.qlen;
	uint8_t *tbp = tb + pr.tb_off;
	const size_t tbsteps = (MODE != K2A_MODE_SCORE && valid) ? k2a_tb_steps<G, C, true>(pr.qlen, pr.tlen, pr.w) : 0;
	const int R = G * C;
	const int ngen = valid ? (pr.tlen + R - 1) / R : 0;
	size_t kbase = 0;
	bool dropped = false;
	const int ktop = __builtin_amdgcn_readfirstlane(valid ? min(pr.qlen - 1, min(C - 1, pr.tlen - 1) + pr.w) : -1);

	for (int g = 0; g < ngen && !dropped; ++g) {
		int jlo, nsteps;
		k2a_gen_cols<G, C>(g, pr.qlen, pr.tlen, pr.w, &jlo, &nsteps);
		L.begin_generation(g, jlo);
		const bool feeder = (gl == 0 && g > 0);             /* takes its top row from the boundary buffer */
		const bool drain = (gl == G - 1);                   /* bottom row of the generation */
		/* two-deep prefetch of the boundary row for lane 0 */
		int ph0 = K2A_NEG, pe0 = K2A_NEG, pe20 = K2A_NEG, ph1 = K2A_NEG, pe1 = K2A_NEG, pe21 = K2A_NEG;
		if (feeder) {
			if (jlo > 0) L.hu_prev = __hip_atomic_load(&Bh[jlo - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			ph0 = __hip_atomic_load(&Bh[jlo], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			pe0 = __hip_atomic_load(&Be[jlo], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			if (DUAL) pe20 = __hip_atomic_load(&Be2[jlo], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			if (jlo + 1 < pr.qlen) {
				ph1 = __hip_atomic_load(&Bh[jlo + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				pe1 = __hip_atomic_load(&Be[jlo + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				if (DUAL) pe21 = __hip_atomic_load(&Be2[jlo + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			}
		}
		L.qb = L.next_query_code(-1);
		for (int k = 0; k < nsteps; ++k) {
			int hin = k2a_rot1<G>(L.hout);
			int ein = k2a_rot1<G>(L.eout);
			int e2in = DUAL ? k2a_rot1<G>(L.e2out) : 0;
			if (feeder) {
				hin = ph0; ein = pe0; e2in = pe20;
				ph0 = ph1; pe0 = pe1; pe20 = pe21;
				const int jn = jlo + k + 2;
				if (jn < pr.qlen) {
					ph1 = __hip_atomic_load(&Bh[jn], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
					pe1 = __hip_atomic_load(&Be[jn], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
					if (DUAL) pe21 = __hip_atomic_load(&Be2[jn], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				}
			}
			const bool ninit = L.need_init(k);
			if (__builtin_amdgcn_ballot_w64(ninit) != 0) {
				if (ninit) L.template do_init<false>(sc, tabs);
			}
			L.hu_prev = hin;
			const int qnext = L.next_query_code(k);
			if (g == 0 && k <= ktop) L.top_inputs(sc, k, hin, ein, e2in);
			uint32_t tw[Lane::TBWORDS];
			const int jj = L.column(k);
			const bool mine = L.S >= 0 && jj >= 0 && jj <= L.je;
			const bool wild = sc.m > 5 || __builtin_amdgcn_ballot_w64(L.qb >= 4) != 0;
			const bool live = L.step(sc, tabs + 8, mtab, wild, k, hin, ein, e2in, tw);
			if (MODE != K2A_MODE_SCORE) {
				if (live) {
					uint32_t *dst = (uint32_t*)(tbp + k2a_tb_word(kbase + (size_t)k, gl, tbsteps, G, Lane::TBWORDS * 4));
					if (Lane::TBWORDS == 1) dst[0] = tw[0];
					else if (Lane::TBWORDS == 2) *(uint2*)dst = make_uint2(tw[0], tw[1]);
					else {
#pragma unroll
						for (int x = 0; x < Lane::TBWORDS; x += 4)
							*(uint4*)(dst + x) = make_uint4(tw[x], tw[x + 1 < Lane::TBWORDS ? x + 1 : x], tw[x + 2 < Lane::TBWORDS ? x + 2 : x],
							                               tw[x + 3 < Lane::TBWORDS ? x + 3 : x]);
					}
				}
			}
			if (drain && mine) {
				Bh[jj] = L.hout; Be[jj] = L.eout;
				if (DUAL) Be2[jj] = L.e2out;
			}
			const bool nfin = L.need_fin(k);
			if (__builtin_amdgcn_ballot_w64(nfin) != 0) {
				if (nfin) L.do_fin(sc, bk, pr.zdrop, rowbuf[wave]);
				__builtin_amdgcn_wave_barrier();
				if (bk->dropped) dropped = true;
			}
			L.qb = qnext;
			if (dropped) break;
		}
		kbase += k2a_gen_pad(nsteps);
	}
	__builtin_amdgcn_wave_barrier();
	if (valid && gl == 0) {
		const K2aBook b = *bk;
		k2a_finish(pr, b, &res[pi]);
	}
}

template<int G, int C, bool DUAL, bool MP>
__global__ void __launch_bounds__(64)
k2a_trace_kernel(const K2aPair *__restrict__ pairs, const uint32_t *__restrict__ order, int ntasks,
                 const uint8_t *__restrict__ tb, K2aResult *__restrict__ res, uint32_t *__restrict__ cig, int ppw)
{
	/* the walk is a chain of dependent, scattered loads: a few walks per wavefront on many wavefronts beats
	 * 64 walks whose loads serialise in one texture-address unit */
	__shared__ K2aUnit16 win[K2A_TRACE_PPW_MAX][K2A_WALK_SLOT / 16];      /* every walk's window of lane-step words (k2a_trace_walk) */
	if ((int)threadIdx.x >= ppw) return;
	const int t = blockIdx.x * ppw + threadIdx.x;
	if (t >= ntasks) return;
	const uint32_t pi = order[t];
	const K2aPair pr = pairs[pi];
	const int ti = res[pi].ti, tj = res[pi].tj;
	int n = 0;
	if (ti >= 0 && tj >= 0) n = k2a_trace_pair<G, C, DUAL, MP>(tb + pr.tb_off, ti, tj, cig + pr.cig_off, pr.qlen, pr.tlen, pr.w, (uint8_t*)win[threadIdx.x]);
	res[pi].n_cigar = n;
}

__global__ void __launch_bounds__(64)
k2a_compact_kernel(const K2aPair *__restrict__ pairs, const K2aResult *__restrict__ res, const uint32_t *__restrict__ pos, int n,
                   const uint32_t *__restrict__ cig, uint32_t *__restrict__ pool)
{
	const int i = blockIdx.x;
	if (i >= n) return;
	const int nc = res[i].n_cigar;
	const uint32_t *src = cig + pairs[i].cig_off;
	uint32_t *dst = pool + pos[i];
	/* the walk emits end -> start; callers get start -> end unless KSW_EZ_REV_CIGAR (ksw2.h:157-159): turned round here, so that
	 * the host's assembly is one memcpy per alignment */
	const bool rev = (pairs[i].flag & K2A_F_REV_CIGAR) != 0;
	for (int k = threadIdx.x; k < nc; k += 64) dst[k] = src[rev ? k : nc - 1 - k];
}

/* uniform plans (K2aUniform): the batch's records, task list and piece counts by rule, one thread per pair */
__global__ void __launch_bounds__(256)
k2a_uniform_layout_kernel(const K2aUniform u, K2aPair *__restrict__ pairs, uint32_t *__restrict__ order2, uint32_t *__restrict__ need)
{
	const uint32_t i = blockIdx.x * 256u + threadIdx.x;
	if (i < u.n) {
		pairs[i] = k2a_uniform_pair(u, i);
		order2[i] = i;                                                    /* task t = pairs 2t, 2t + 1 */
	}
	if (need && i < (u.ntasks + u.ng - 1) / u.ng) need[i] = k2a_uniform_need(u, i);
}

/* ---------------------------------------------------------------- dispatch tables */

typedef void (*fill_fn)(const K2aScoring, const K2aPair*, const uint32_t*, int, const uint8_t*, uint8_t*, K2aResult*);
typedef void (*trace_fn)(const K2aPair*, const uint32_t*, int, const uint8_t*, K2aResult*, uint32_t*, int);

#define FILL_ROW(G, C) { { k2a_fill_kernel<G, C, false, 0>, k2a_fill_kernel<G, C, false, 1>, k2a_fill_kernel<G, C, false, 2> }, \
                         { k2a_fill_kernel<G, C, true, 0>,  k2a_fill_kernel<G, C, true, 1>,  k2a_fill_kernel<G, C, true, 2> } }
static const fill_fn g_fill[4][2][3] = { FILL_ROW(16, 8), FILL_ROW(64, 8), FILL_ROW(64, 16), FILL_ROW(64, 32) };
typedef void (*fill_mp_fn)(const K2aScoring, const K2aPair*, const uint32_t*, int, const uint8_t*, uint8_t*, int32_t*, K2aResult*);
static const fill_mp_fn g_fill_mp[2][3] = {
	{ k2a_fill_mp_kernel<64, 16, false, 0>, k2a_fill_mp_kernel<64, 16, false, 1>, k2a_fill_mp_kernel<64, 16, false, 2> },
	{ k2a_fill_mp_kernel<64, 16, true, 0>,  k2a_fill_mp_kernel<64, 16, true, 1>,  k2a_fill_mp_kernel<64, 16, true, 2> } };
static const fill_mp_fn g_fill_mp_lds[2] = { k2a_fill_mp_kernel<64, 16, false, 1, true>, k2a_fill_mp_kernel<64, 16, false, 2, true> };   /* [mode - 1] */
#define TRACE_ROW(G, C, MP) { k2a_trace_kernel<G, C, false, MP>, k2a_trace_kernel<G, C, true, MP> }
static const trace_fn g_trace[K2A_NCFG][2] = { TRACE_ROW(16, 8, false), TRACE_ROW(64, 8, false), TRACE_ROW(64, 16, false),
                                               TRACE_ROW(64, 32, false), TRACE_ROW(64, 16, true) };

static const fill_fn g_fill_solo[2][3] = {
	{ k2a_fill_solo_kernel<K2A_SOLO_CS, false, 0>, k2a_fill_solo_kernel<K2A_SOLO_C, false, 1>, k2a_fill_solo_kernel<K2A_SOLO_C, false, 2> },
	{ k2a_fill_solo_kernel<K2A_SOLO_CS, true, 0>,  k2a_fill_solo_kernel<K2A_SOLO_C, true, 1>,  k2a_fill_solo_kernel<K2A_SOLO_C, true, 2> } };

/* packed walk: thread t = alignment (t & 1) of task (t >> 1) */
template<int G, int C, bool DUAL, bool MP = false>
__global__ void __launch_bounds__(64)
k2a_trace_pk_kernel(const K2aPair *__restrict__ pairs, const uint32_t *__restrict__ order2, int ntasks,
                    const uint8_t *__restrict__ tb, K2aResult *__restrict__ res, uint32_t *__restrict__ cig, int ppw)
{
	__shared__ K2aUnit16 win[K2A_TRACE_PPW_MAX][K2A_WALK_SLOT / 16];
	if ((int)threadIdx.x >= ppw) return;
	const int t = blockIdx.x * ppw + threadIdx.x;
	if (t >= 2 * ntasks) return;
	const int half = t & 1;
	const uint32_t piA = order2[t & ~1], pi = order2[t];
	if (half && pi == piA) return;                    /* unpaired leftover: the task holds one alignment */
	const K2aPair pr = pairs[pi];
	const int ti = res[pi].ti, tj = res[pi].tj;
	int n = 0;
	if (ti >= 0 && tj >= 0) n = k2a_trace_pair_pk<G, C, DUAL, MP>(tb + pr.tb_off, half, ti, tj, cig + pr.cig_off, pr.qlen, pr.tlen, pr.w, (uint8_t*)win[threadIdx.x]);
	res[pi].n_cigar = n;
}

/* Packed-int16 generation-serial fill (ksw2_lane_pkmp.h): ONE pair of same-shape alignments per workgroup of K2A_PKMP_WAVES
 * wavefronts; wavefront v runs generations v, v + W, v + 2W, ... (1024 target rows each), so up to W generations of a pair are
 * in flight, each reading the boundary row its predecessor streams into HBM ({H, E, baseA, baseB} per column, L1-bypassing
 * loads two columns ahead).  No spinning: the wavefronts advance in phases of K2A_PKMP_T steps with one workgroup barrier per
 * phase, and generation g + 1 starts k2a_pkmp_lag phases after generation g -- late enough that every boundary column it reads
 * was written in an earlier phase and that its strip epilogues (row order!) come after all of generation g's.  The schedule
 * (start phase of every generation) is a function of the shape alone and is tabulated once per workgroup. */
template<bool DUAL, int MODE, bool TN>
__device__ __forceinline__ void
k2a_fill_pkmp_body(const K2aScoring &sc, const K2aPair *__restrict__ pairs, const uint32_t *__restrict__ order2, int ntasks,
                   const uint8_t *__restrict__ seq, uint8_t *__restrict__ tb, uint32_t *bnd, K2aResult *__restrict__ res, int task, int wave, int lane,
                   K2aBook *book, uint32_t (*rowbuf)[16], int *pstart, int *pcount, uint4 *tbstage, unsigned long long *tbruns, uint32_t (*cptab)[8])
{
	constexpr int C = 16, G = 64, W = K2A_PKMP_WAVES, T = K2A_PKMP_T, R = G * C;
	typedef K2aLanePkMp<C, DUAL, MODE, TN> Lane;
	constexpr int WB = Lane::TBWORDS * 4;                 /* 16 bytes (single gap: 4-bit codes) or 32 per lane-step */
	constexpr bool STAGED = MODE != K2A_MODE_SCORE;
	typedef K2aTbStage<WB, 8> Stage;

	const uint32_t piA = order2[2 * task], piB = order2[2 * task + 1];
	const K2aPair prA = pairs[piA], prB = pairs[piB];
	const int qlen = prA.qlen, tlen = prA.tlen, w = prA.w;
	const int ngen = (tlen + R - 1) / R;

	k2a_cptab_fill(sc, cptab[wave], lane);
	Lane L;
	/* the task's block of `bnd`: boundary entries, then one block of row-maximum keys per wavefront */
	L.setup(prA, prB, seq, lane, true, (unsigned long long*)(bnd + prA.bnd_off + K2A_PKMP_BND_WORDS(qlen, DUAL)) + (size_t)wave * (K2A_PKMP_SPILL_WORDS(C) / 2), cptab[wave]);
	K2aBook *bkA = &book[0], *bkB = &book[1];
	if (threadIdx.x == 0) {
		k2a_book_reset(bkA); k2a_book_reset(bkB);
		int pend[W], prev_start = 0, prev_jlo = 0, total = 0;
		for (int x = 0; x < W; ++x) pend[x] = 0;
		for (int x = 0; x < ngen; ++x) {
			int jlo, ns;
			k2a_gen_cols<G, C>(x, qlen, tlen, w, &jlo, &ns);
			int ps = x >= W ? pend[x % W] : 0;
			if (x >= 1) ps = max(ps, prev_start + k2a_pkmp_lag(prev_jlo, jlo));
			pstart[x] = ps; pcount[x] = (ns + T - 1) / T;
			pend[x % W] = ps + pcount[x];
			total = max(total, pend[x % W]);
			prev_start = ps; prev_jlo = jlo;
		}
		pstart[64] = total;
	}
	__syncthreads();
	const int total_phases = pstart[64];

	uint4 *B1 = (uint4*)(bnd + prA.bnd_off);
	uint32_t *B2 = (uint32_t*)(B1 + qlen);
	uint8_t *tbp = tb + prA.tb_off;
	const size_t tbsteps = STAGED ? k2a_tb_steps<G, C, true>(qlen, tlen, w) : 0;
	Stage ST;
	if (STAGED) ST.init(&tbstage[wave * Stage::WORDS], &tbruns[wave * 64], lane, tbp + k2a_tb_word(0, lane, tbsteps, G, WB));
	const int zdropA = prA.zdrop, zdropB = prB.zdrop;
	const int ktop = __builtin_amdgcn_readfirstlane(min(qlen - 1, min(C - 1, tlen - 1) + w));
	const k2a_pk neg = k2a_pku(K2A_NEG16);

	int g = wave;                                     /* the generation this wavefront runs next */
	int jlo = 0, nsteps = 0, je_prev = -1, kdone = -1;
	int kbase = 0, gk = 0;                            /* padded steps of the generations before g; generations counted into it */
	bool feeder = false, drain = false;
	/* lane 0 of generations > 0: boundary entries {H, E, baseA, baseB} (+ E~) of this step's column and of the next one */
	uint4 cur = make_uint4(0, 0, 0, 0), nxt = cur;
	uint32_t cur2 = 0, nxt2 = 0;
	int bs0A = 0, bs0B = 0;                           /* bases of column jlo - 1 (the first cell's diagonal input) */
	/* The boundary entry of column j.  EVERY lane loads it, from the same (wave-uniform, clamped) address -- one request -- and in
	 * every step, although only lane 0 of the generations > 0 uses it: a load under `if (feeder)` into a variable that lives around
	 * the step loop is waited for right behind the load (hipcc's phi copy, ksw2_lane_pk.h: k2a_load_early), and that wait, an L2
	 * round trip, stood in every step of every wavefront of config 4.  Unconditional, the entry is in flight for a whole step; whether
	 * it is one the generation above wrote is decided when it is used (`take`). */
	auto fetch_raw = [&](int j, uint4 &v, uint32_t &v2) {
		const int jc = min(max(j, 0), qlen - 1);
		const uint32_t *e = (const uint32_t*)&B1[jc];
		v.x = __hip_atomic_load(&e[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); v.y = __hip_atomic_load(&e[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		v.z = __hip_atomic_load(&e[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); v.w = __hip_atomic_load(&e[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		v2 = DUAL ? __hip_atomic_load(&B2[jc], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
	};
	auto take = [&](int j, uint4 &v, uint32_t &v2) {        /* past what the generation above wrote: outside the band */
		if (j > je_prev || j < 0) { v = make_uint4(neg, neg, 0, 0); v2 = neg; }
	};

	for (int phase = 0; phase < total_phases; ++phase) {
		if (bkA->dropped && bkB->dropped) break;                       /* uniform: the books were last written before the barrier */
		if (g < ngen && phase >= pstart[g]) {
			if (phase == pstart[g]) {                                  /* begin the generation */
				k2a_gen_cols<G, C>(g, qlen, tlen, w, &jlo, &nsteps);
				for (; gk < g; ++gk) { int a, b; k2a_gen_cols<G, C>(gk, qlen, tlen, w, &a, &b); kbase += (int)k2a_gen_pad(b); }
				L.begin_generation(g, jlo);
				L.clear_spill();
				feeder = lane == 0 && g > 0;
				drain = lane == G - 1 && g + 1 < ngen;
				je_prev = g > 0 ? min(qlen - 1, g * R - 1 + w) : -1;          /* last column the generation above wrote */
				if (g > 0) {
					if (jlo > 0) {                                            /* H(i0 - 1, jlo - 1): the diagonal input of the first cell */
						uint4 pv; uint32_t pv2;
						fetch_raw(jlo - 1, pv, pv2);
						take(jlo - 1, pv, pv2);
						if (feeder) { L.P.hu_prev = pv.x; bs0A = (int)pv.z; bs0B = (int)pv.w; }
					}
					fetch_raw(jlo, cur, cur2);
					fetch_raw(jlo + 1, nxt, nxt2);
				}
				L.P.set_qb(L.P.next_query_codes(-1));
				kdone = -1;
				__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");   /* the cleared keys are in place before any atomic on them */
			}
			const int k0 = (phase - pstart[g]) * T, k1 = min(nsteps, k0 + T);
			for (int k = k0; k < k1; ++k) {
				k2a_pk hin = (k2a_pk)k2a_rot1<G>((int)L.P.hout);
				k2a_pk ein = (k2a_pk)k2a_rot1<G>((int)L.P.eout);
				k2a_pk e2in = DUAL ? (k2a_pk)k2a_rot1<G>((int)L.P.e2out) : 0u;
				const bool ninit = L.need_init(k);
				if (__builtin_amdgcn_ballot_w64(ninit) != 0) {
					int bsA = k2a_rot1<G>(L.P.baseA), bsB = k2a_rot1<G>(L.P.baseB);
					if (feeder) { bsA = bs0A; bsB = bs0B; }
					if (ninit) L.do_init(sc, bsA, bsB);                  /* uses hu_prev = what arrived one step ago */
					bsA = k2a_rot1<G>(L.P.baseA); bsB = k2a_rot1<G>(L.P.baseB);
					L.refresh_delta(bsA, bsB);                           /* a base changed: every lane re-reads its neighbour's */
					if (TN) K2A_SYNC_WN(L.P);
				}
				if (feeder) {
					take(jlo + k, cur, cur2);
					hin = cur.x; ein = cur.y; e2in = cur2;
					L.refresh_delta((int)cur.z, (int)cur.w);             /* every entry carries the bases it is relative to */
				}
				cur = nxt; cur2 = nxt2;
				fetch_raw(jlo + k + 2, nxt, nxt2);                       /* every lane: see fetch_raw */
				L.P.hu_prev = hin;
				hin = L.adopt(hin); ein = L.adopt(ein);
				if (DUAL) e2in = L.adopt(e2in);
				const uint32_t qnext = L.P.next_query_codes(k);
				if (g == 0 && k <= ktop) L.P.top_inputs(sc, k, hin, ein, e2in);
				uint32_t tw[Lane::TBWORDS];
				const int jj = L.column(k);
				const bool mine = L.P.S >= 0 && jj >= 0 && jj <= L.P.je;
				const bool live = L.P.step(sc, k, hin, ein, e2in, tw);
				if (STAGED) { ST.put(kbase + k, tw, live); ST.step_done(kbase + k); kdone = k; }
				if (drain && mine) {
					B1[jj] = make_uint4(L.P.hout, L.P.eout, (uint32_t)L.P.baseA, (uint32_t)L.P.baseB);
					if (DUAL) B2[jj] = L.P.e2out;
				}
				const bool nfin = L.need_fin(k);
				if (__builtin_amdgcn_ballot_w64(nfin) != 0) {
					if (nfin) L.flush_rowmax();
					__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");   /* the keys' atomics (performed at L2) are done; do_fin reads them past L1 */
					if (nfin) L.do_fin(sc, bkA, bkB, zdropA, zdropB, rowbuf[wave]);
					__builtin_amdgcn_wave_barrier();
					if (TN && L.P.wn) K2A_SYNC_WN(L.P);
				}
				L.P.set_qb(qnext);                                       /* next step's codes and column profiles */
				if ((k & (T - 1)) == T - 1) {
					const k2a_pk d = L.rebase();
					const k2a_pk da = (k2a_pk)k2a_rot1<G>((int)d);
					const int bsA = k2a_rot1<G>(L.P.baseA), bsB = k2a_rot1<G>(L.P.baseB);
					L.after_rebase(da, bsA, bsB);
				}
			}
			if (k1 == nsteps) {                                         /* generation done: on to this wavefront's next one */
				if (STAGED && kdone >= 0) ST.finish(kbase + kdone);
				g += W;
			}
		}
		/* producer and consumer of a boundary column are wavefronts of this workgroup: the stores only have to have left the CU's
		 * write-through L1 (workgroup-scope release = wait for them), the consumer's loads bypass L1.  An agent-scope release here
		 * would write back the whole L2 -- with the traceback stream in it -- every 64 steps. */
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
		__syncthreads();
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
	}
	__syncthreads();
	if (threadIdx.x == 0) {
		const K2aBook a = *bkA, b = *bkB;
		k2a_finish(prA, a, &res[piA]);
		if (piB != piA) k2a_finish(prB, b, &res[piB]);
	}
}

/* every wavefront of the workgroup looks at the task's two targets (k2a_scan_codes; the same answer in all of them) and takes the plain
 * or the TN build of the body */
template<bool DUAL, int MODE>
__global__ void __launch_bounds__(64 * K2A_PKMP_WAVES, K2A_PKMP_WAVES > 4 ? 3 : 2)
k2a_fill_pkmp_kernel(const K2aScoring sc, const K2aPair *__restrict__ pairs, const uint32_t *__restrict__ order2, int ntasks,
                     const uint8_t *__restrict__ seq, uint8_t *__restrict__ tb, uint32_t *bnd, K2aResult *__restrict__ res)
{
	constexpr int C = 16, W = K2A_PKMP_WAVES;
	typedef K2aLanePkMp<C, DUAL, MODE, false> Lane;       /* (sizes only) */
	constexpr int WB = Lane::TBWORDS * 4;
	constexpr bool STAGED = MODE != K2A_MODE_SCORE;
	typedef K2aTbStage<WB, 8> Stage;
	__shared__ K2aBook book[2];
	__shared__ uint32_t rowbuf[W][C];
	__shared__ int pstart[64 + 1], pcount[64];            /* start phase / phases of every generation (at most 64: reads up to 65 000) */
	__shared__ uint4 tbstage[STAGED ? W * Stage::WORDS : 1];
	__shared__ unsigned long long tbruns[STAGED ? W * 64 : 1];
	__shared__ uint32_t cptab[W][8];
	const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
	const int task = blockIdx.x;
	if (task >= ntasks) return;
	uint32_t scan = 0;
	if (sc.pk_tn1) {
		const K2aPair pa = pairs[order2[2 * task]], pb = pairs[order2[2 * task + 1]];
		scan = k2a_scan_codes<64>(seq + pa.toff, pa.tlen_full, lane) | k2a_scan_codes<64>(seq + pb.toff, pb.tlen_full, lane);
	}
	if (__builtin_amdgcn_ballot_w64((scan & 1u) != 0) != 0)
		k2a_fill_pkmp_body<DUAL, MODE, true>(sc, pairs, order2, ntasks, seq, tb, bnd, res, task, wave, lane, book, rowbuf, pstart, pcount, tbstage, tbruns, cptab);
	else
		k2a_fill_pkmp_body<DUAL, MODE, false>(sc, pairs, order2, ntasks, seq, tb, bnd, res, task, wave, lane, book, rowbuf, pstart, pcount, tbstage, tbruns, cptab);
}

/* ---------------------------------------------------------------- splice-aware extension, diagonal-major (ksw2_lane_dm.h) */

/* lane l <- lane l-1 across the whole wavefront, lane 0 <- `carry` (the previous slot's lane 63) */
__device__ __forceinline__ int k2a_shr1_carry(int v, int carry)
{
	return __builtin_amdgcn_update_dpp(carry, v, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
}

/* maximum of a 64-bit key over the wavefront, uniform result: four row_ror steps leave every 16-lane row's maximum in all
 * of its lanes, the four row results meet on the scalar side */
template<int CTRL>
__device__ __forceinline__ uint64_t k2a_dpp_max_u64(uint64_t k)
{
	const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)k, CTRL, 0xf, 0xf, false);
	const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(k >> 32), CTRL, 0xf, 0xf, false);
	const uint64_t o = ((uint64_t)hi << 32) | lo;
	return o > k ? o : k;
}
template<int CTRL>
__device__ __forceinline__ uint32_t k2a_dpp_max_u32(uint32_t v)
{
	const uint32_t o = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, false);
	return o > v ? o : v;
}
__device__ __forceinline__ uint32_t k2a_wave_max_u32(uint32_t v)
{
	v = k2a_dpp_max_u32<0x121>(v);                    /* row_ror:1, 2, 4, 8: every lane of a row holds the row's maximum */
	v = k2a_dpp_max_u32<0x122>(v);
	v = k2a_dpp_max_u32<0x124>(v);
	v = k2a_dpp_max_u32<0x128>(v);
	const uint32_t a = (uint32_t)__builtin_amdgcn_readlane((int)v, 0), b = (uint32_t)__builtin_amdgcn_readlane((int)v, 16);
	const uint32_t c = (uint32_t)__builtin_amdgcn_readlane((int)v, 32), d = (uint32_t)__builtin_amdgcn_readlane((int)v, 48);
	const uint32_t ab = a > b ? a : b, cd = c > d ? c : d;
	return ab > cd ? ab : cd;
}
/* the wavefront's largest 64-bit key, in two 32-bit rounds (round 6): the largest high word, then the largest low word among the
 * lanes that hold it -- 18 vector instructions where the 64-bit rotate-compare-select (k2a_dpp_max_u64) took 36: the SSE-compatible
 * register form +4 % at four wavefronts per SIMD, `exts` +3.7 % (profiles/r6_ab_wave_max32.txt) */
__device__ __forceinline__ uint64_t k2a_wave_max_u64(uint64_t k)
{
	const uint32_t hi = (uint32_t)(k >> 32), lo = (uint32_t)k;
	const uint32_t M = k2a_wave_max_u32(hi);
	const uint32_t L = k2a_wave_max_u32(hi == M ? lo : 0u);
	return ((uint64_t)M << 32) | L;
}

template<int MODE, int K>
__global__ void __launch_bounds__(64 * K2A_WPB)
k2a_exts_kernel(const K2aSplice sp, const K2aPair *__restrict__ pairs, const uint32_t *__restrict__ order, int ntasks,
                const uint8_t *__restrict__ seq, uint8_t *__restrict__ tb, K2aResult *__restrict__ res)
{
	__shared__ int8_t mtab[K2A_MAXM * K2A_MAXM];
	/* this diagonal's H by window position: the bookkeeping reads five cells by position (last, first, <= 3 tail cells).  Finding
	 * them with v_readlane inside the slot loop cost ~35 scalar instructions per slot and diagonal for the range tests -- the
	 * kernel ran on the CU's one scalar unit (PMC: 967 k SALU against 835 k VALU instructions per wavefront) */
	__shared__ int hrow[K2A_WPB][K * 64];
	for (int x = threadIdx.x; x < sp.m * sp.m; x += blockDim.x) mtab[x] = sp.mat[x];
	__syncthreads();

	const int lane = threadIdx.x & 63, wave = k2a_wave_id<true>();
	const int task = blockIdx.x * K2A_WPB + wave;
	if (task >= ntasks) return;                       /* whole wavefronts leave; nobody synchronises below */
	int *const hw = hrow[wave];                       /* (plain LDS accesses: one wavefront's DS operations execute in order) */
	const uint32_t pi = order[task];
	const K2aPair pr = pairs[pi];
	const int qlen = pr.qlen, tlen = pr.tlen_full, ncol = min(qlen, tlen);
	const uint8_t *qry = seq + pr.qoff;
	const uint32_t *cst = (const uint32_t*)seq + pr.bnd_off;
	uint8_t *tbp = tb + pr.tb_off;

	int H1[K], H2[K], En[K], E2n[K], Fn[K];
	uint32_t Q[K], Cst[K];
#pragma unroll
	for (int s = 0; s < K; ++s) {
		H1[s] = H2[s] = En[s] = E2n[s] = Fn[s] = K2A_NEG; Q[s] = 0;
		Cst[s] = cst[min(s * 64 + lane, tlen - 1)];
	}
	int base = 0;
	K2aBook book;
	k2a_book_reset(&book);
	uint32_t qnext = qry[0];

	for (int r = 0; r < qlen + tlen - 1; ++r) {
		const int st0 = max(0, r - qlen + 1), en0 = min(tlen - 1, r), en1 = st0 + (en0 - st0) / 4 * 4;
		if (st0 >= 1 && (st0 - 1) / 64 > base) {                           /* slide the window (st0 grows by at most 1 per diagonal): slot s <- slot s+1 */
#pragma unroll
			for (int s = 0; s + 1 < K; ++s) {
				H1[s] = H1[s + 1]; H2[s] = H2[s + 1]; En[s] = En[s + 1]; E2n[s] = E2n[s + 1]; Fn[s] = Fn[s + 1]; Q[s] = Q[s + 1]; Cst[s] = Cst[s + 1];
			}
			++base;
			H1[K - 1] = H2[K - 1] = En[K - 1] = E2n[K - 1] = Fn[K - 1] = K2A_NEG; Q[K - 1] = 0;
			Cst[K - 1] = cst[min((base + K - 1) * 64 + lane, tlen - 1)];
		}
		const uint32_t qcur = qnext;
		qnext = qry[min(r + 1, qlen - 1)];                               /* used one diagonal later */
		const int br = k2a_dm_border(sp, r), br1 = k2a_dm_border(sp, r + 1);
		int cH2 = K2A_NEG, cEn = K2A_NEG, cE2n = K2A_NEG, cQ = 0;
		int A = K2A_NEG, S = K2A_NEG, T0 = K2A_NEG, T1 = K2A_NEG, T2 = K2A_NEG;
		int bH = K2A_NEG, bT = -1;
#pragma unroll
		for (int s = 0; s < K; ++s) {
			const int t0 = (base + s) * 64, t = t0 + lane;
			/* wave-uniform: slots below the diagonal's first cell are finished, slots above its last cell (+1: the query code
			 * that shifts into next diagonal's new cell) hold nothing yet */
			if (t0 > en0 + 1 || t0 + 63 < st0 - 1) continue;
			const int h2s = k2a_shr1_carry(H2[s], cH2), ens = k2a_shr1_carry(En[s], cEn), e2ns = k2a_shr1_carry(E2n[s], cE2n);
			const uint32_t qs = (uint32_t)k2a_shr1_carry((int)Q[s], cQ);
			cH2 = __builtin_amdgcn_readlane(H2[s], 63); cEn = __builtin_amdgcn_readlane(En[s], 63);
			cE2n = __builtin_amdgcn_readlane(E2n[s], 63); cQ = __builtin_amdgcn_readlane((int)Q[s], 63);
			Q[s] = t == 0 ? qcur : qs;
			if (t0 <= en0 && t0 + 63 >= st0) {                              /* the slot holds cells of this diagonal */
				/* first row: t == 0, first column: t == r -- the border values both need are those of r and r + 1 (uniform, formed
				 * once per diagonal: as border(t) inside the slot loop they were divergent scalar branches in every slot) */
				const bool active = t >= st0 && t <= en0, first_row = t == 0, first_col = t == r;
				const int diag = (first_row || first_col) ? br : h2s;
				const int ein = first_row ? br1 - sp.q - sp.e : ens;
				const int e2in = first_row ? br1 - sp.q2 : e2ns;
				const int fin = first_col ? br1 - sp.q - sp.e : Fn[s];
				const uint32_t c = Cst[s];
				const int sc = (int)mtab[(c & 0xffu) * (uint32_t)sp.m + (Q[s] & 0xffu)];
				int z, en, e2n, fn;
				uint32_t dir;
				k2a_dm_cell<MODE>(sp, diag, ein, e2in, fin, sc, c, z, en, e2n, fn, dir);
				if (active) {
					H2[s] = H1[s]; H1[s] = z; En[s] = en; E2n[s] = e2n; Fn[s] = fn;
					if (MODE != K2A_MODE_SCORE) tbp[(size_t)r * ncol + (t - st0)] = (uint8_t)dir;
					if (t < en1 && z > bH) { bH = z; bT = t; }
				}
				hw[s * 64 + lane] = H1[s];                                    /* read back by position below */
			}
		}
		{
			/* the cells the bookkeeping reads by position: the diagonal's last and first cell, the (<= 3) tail cells; lane x
			 * fetches the x-th of them (every one lies in [st0, en0], inside the window) */
			const int wb = base * 64;
			const int pos = lane == 0 ? en0 : lane == 1 ? st0 : min(en1 + (lane - 2), en0);
			__builtin_amdgcn_wave_barrier();
			const int v = lane < 5 ? hw[pos - wb] : K2A_NEG;
			__builtin_amdgcn_wave_barrier();
			A = __builtin_amdgcn_readlane(v, 0); S = __builtin_amdgcn_readlane(v, 1);
			if (en1 < en0) T0 = __builtin_amdgcn_readlane(v, 2);
			if (en1 + 1 < en0) T1 = __builtin_amdgcn_readlane(v, 3);
			if (en1 + 2 < en0) T2 = __builtin_amdgcn_readlane(v, 4);
		}
		const uint64_t Bkey = k2a_wave_max_u64(bT >= 0 ? k2a_dm_key(bH, bT, st0) : 0ull);
		if (k2a_dm_book(&book, r, st0, en0, qlen, tlen, pr.zdrop, A, Bkey, T0, T1, T2, S)) break;
	}
	if (lane == 0) k2a_finish(pr, book, &res[pi]);
}

/* The same function for diagonals of any length: the per-position state lives in a scratch array in HBM (L2-resident:
 * 9 ints per target position) instead of registers, double-buffered by diagonal parity (H: three diagonals), so the value
 * of position t-1 is simply read at t-1.  Writer and reader are lanes of the same wavefront, so workgroup scope is enough:
 * one release / acquire pair per diagonal orders a diagonal's stores before the next diagonal's loads (the CU's L1 is
 * write-through and shared by the wavefront).  Slower than the register
 * kernel (about 44 bytes of L2 traffic per cell), it exists so that no input size is refused. */
template<int MODE>
__global__ void __launch_bounds__(64 * K2A_WPB)
k2a_exts_big_kernel(const K2aSplice sp, const K2aPair *__restrict__ pairs, const uint32_t *__restrict__ order, int ntasks,
                    const uint8_t *__restrict__ seq, uint8_t *__restrict__ tb, int32_t *scratch, K2aResult *__restrict__ res)
{
	__shared__ int8_t mtab[K2A_MAXM * K2A_MAXM];
	for (int x = threadIdx.x; x < sp.m * sp.m; x += blockDim.x) mtab[x] = sp.mat[x];
	__syncthreads();

	const int lane = threadIdx.x & 63, wave = k2a_wave_id<true>();
	const int task = blockIdx.x * K2A_WPB + wave;
	if (task >= ntasks) return;
	const uint32_t pi = order[task];
	const K2aPair pr = pairs[pi];
	const int qlen = pr.qlen, tlen = pr.tlen_full, ncol = min(qlen, tlen);
	const uint8_t *qry = seq + pr.qoff;
	const uint32_t *cst = (const uint32_t*)seq + pr.bnd_off;
	uint8_t *tbp = tb + pr.tb_off;
	int32_t *W = scratch + (size_t)pr.pad * 4;         /* pad = scratch offset in units of 4 ints */
	K2aBook book;
	k2a_book_reset(&book);

	for (int r = 0; r < qlen + tlen - 1; ++r) {
		const int st0 = max(0, r - qlen + 1), en0 = min(tlen - 1, r), en1 = st0 + (en0 - st0) / 4 * 4;
		int32_t *Hc = W + (size_t)(r % 3) * tlen;
		const int32_t *H2 = W + (size_t)((r + 1) % 3) * tlen;                         /* diagonal r - 2 */
		int32_t *Ec = W + (size_t)(3 + (r & 1)) * tlen, *E2c = W + (size_t)(5 + (r & 1)) * tlen, *Fc = W + (size_t)(7 + (r & 1)) * tlen;
		const int32_t *Ep = W + (size_t)(3 + ((r + 1) & 1)) * tlen, *E2p = W + (size_t)(5 + ((r + 1) & 1)) * tlen, *Fp = W + (size_t)(7 + ((r + 1) & 1)) * tlen;
		int A = K2A_NEG, S = K2A_NEG, T0 = K2A_NEG, T1 = K2A_NEG, T2 = K2A_NEG;
		int bH = K2A_NEG, bT = -1;
		for (int t0 = st0 & ~63; t0 <= en0; t0 += 64) {
			const int t = t0 + lane;
			const bool active = t >= st0 && t <= en0;
			int z = K2A_NEG;
			if (active) {
				const bool first_row = t == 0, first_col = t == r;
				const int diag = first_row ? k2a_dm_border(sp, r) : first_col ? k2a_dm_border(sp, t)
				                 : __hip_atomic_load(&H2[t - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
				const int ein = first_row ? k2a_dm_border(sp, r + 1) - sp.q - sp.e : __hip_atomic_load(&Ep[t - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
				const int e2in = first_row ? k2a_dm_border(sp, r + 1) - sp.q2 : __hip_atomic_load(&E2p[t - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
				const int fin = first_col ? k2a_dm_border(sp, t + 1) - sp.q - sp.e : __hip_atomic_load(&Fp[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
				const uint32_t c = cst[t];
				const int sc = (int)mtab[(c & 0xffu) * (uint32_t)sp.m + qry[r - t]];
				int en, e2n, fn;
				uint32_t dir;
				k2a_dm_cell<MODE>(sp, diag, ein, e2in, fin, sc, c, z, en, e2n, fn, dir);
				Hc[t] = z; Ec[t] = en; E2c[t] = e2n; Fc[t] = fn;
				if (MODE != K2A_MODE_SCORE) tbp[(size_t)r * ncol + (t - st0)] = (uint8_t)dir;
				if (t < en1 && z > bH) { bH = z; bT = t; }
			}
			if (en0 >= t0 && en0 < t0 + 64) A = __builtin_amdgcn_readlane(z, en0 & 63);
			if (st0 >= t0 && st0 < t0 + 64) S = __builtin_amdgcn_readlane(z, st0 & 63);
			if (en1 < en0 && en1 >= t0 && en1 < t0 + 64) T0 = __builtin_amdgcn_readlane(z, en1 & 63);
			if (en1 + 1 < en0 && en1 + 1 >= t0 && en1 + 1 < t0 + 64) T1 = __builtin_amdgcn_readlane(z, (en1 + 1) & 63);
			if (en1 + 2 < en0 && en1 + 2 >= t0 && en1 + 2 < t0 + 64) T2 = __builtin_amdgcn_readlane(z, (en1 + 2) & 63);
		}
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");                /* this diagonal's state is written ... */
		const uint64_t Bkey = k2a_wave_max_u64(bT >= 0 ? k2a_dm_key(bH, bT, st0) : 0ull);
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");                /* ... before anybody reads it on the next one */
		if (k2a_dm_book(&book, r, st0, en0, qlen, tlen, pr.zdrop, A, Bkey, T0, T1, T2, S)) break;
	}
	if (lane == 0) k2a_finish(pr, book, &res[pi]);
}

__global__ void __launch_bounds__(64)
k2a_exts_trace_kernel(const K2aSplice sp, const K2aPair *__restrict__ pairs, const uint32_t *__restrict__ order, int ntasks,
                      const uint8_t *__restrict__ tb, K2aResult *__restrict__ res, uint32_t *__restrict__ cig, int ppw)
{
	if ((int)threadIdx.x >= ppw) return;
	const int t = blockIdx.x * ppw + threadIdx.x;
	if (t >= ntasks) return;
	const uint32_t pi = order[t];
	const K2aPair pr = pairs[pi];
	const int ti = res[pi].ti, tj = res[pi].tj;
	int n = 0;
	if (ti >= 0 && tj >= 0) n = k2a_dm_trace(tb + pr.tb_off, min(pr.qlen, pr.tlen_full), ti, tj, cig + pr.cig_off, pr.qlen, sp.long_thres);
	res[pi].n_cigar = n;
}

/* per-position constants of the splice-aware plans (k2a_splice_const): one workgroup per pair, from the target (and annotation)
 * bytes at pr.toff into the dwords at pr.bnd_off that the fill kernels read */
__global__ void __launch_bounds__(256)
k2a_splice_const_kernel(const K2aPair *__restrict__ pairs, int n, uint8_t *seq, int noncan, int junc_bonus)
{
	const int i = blockIdx.x;
	if (i >= n) return;
	const K2aPair pr = pairs[i];
	if (pr.qlen <= 0 || pr.tlen_full <= 0) return;
	const uint8_t *T = seq + pr.toff, *J = (pr.flag & K2A_F_HAS_JUNC) ? T + ((pr.tlen_full + 3) & ~3) : 0;
	uint32_t *out = (uint32_t*)seq + pr.bnd_off;
	for (int t = threadIdx.x; t < pr.tlen_full; t += 256) out[t] = k2a_splice_const(T, J, t, pr.tlen_full, pr.flag, noncan, junc_bonus);
}

/* ---------------------------------------------------------------- SSE-compatible mode (ksw2_lane_ssec.h) */

/* One alignment per wavefront, lane <-> target position, one step per anti-diagonal; the reference's byte arrays u v x y
 * [x~ y~] s and its int32 H live in `scratch` (16 * pairs[i].bnd_off bytes in; (5 or 7) + 4 bytes per padded target position,
 * L2-resident).  Positions move between lanes from one anti-diagonal to the next and the phases of an anti-diagonal read what
 * other lanes wrote in the phase before, so workgroup-scope release / acquire pairs separate them (writer and reader are
 * lanes of one wavefront; the CU's L1 is write-through). */
/* LDS = true: the same arrays in the workgroup's dynamic LDS (one wavefront per workgroup, (5 or 7) + 4 bytes per padded target
 * position): a wavefront's DS operations execute in order, so the phases need no memory fence, only the compiler kept from
 * reordering -- and no L2 round trip per phase, which is what an anti-diagonal of the HBM form consists of. */
#define K2A_SSEC_SYNC() do { if (LDS) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } \
                             else { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); } } while (0)
extern __shared__ uint8_t k2a_ssec_lds[];
template<bool DUAL, int MODE, bool LDS>
__global__ void __launch_bounds__(64 * K2A_WPB)
k2a_ssec_kernel(const K2aSsec P, const K2aPair *__restrict__ pairs, const uint32_t *__restrict__ order, int ntasks,
                const uint8_t *__restrict__ seq, uint8_t *__restrict__ tb, uint8_t *scratch, K2aResult *__restrict__ res)
{
	const int lane = threadIdx.x & 63, wave = k2a_wave_id<true>();
	const int task = LDS ? (int)blockIdx.x : blockIdx.x * K2A_WPB + wave;
	if (task >= ntasks) return;                       /* whole wavefronts leave; nobody synchronises below */
	const uint32_t pi = order[task];
	const K2aPair pr = pairs[pi];
	const int qlen = pr.qlen, tlen = pr.tlen_full, w = pr.w, T16 = (tlen + 15) / 16 * 16, ncol = k2a_ssec_ncol(qlen, tlen, w);
	const bool approx = (pr.pad & K2A_SSEC_APPROX) != 0, adrop = (pr.pad & K2A_SSEC_APPROX_DROP) != 0, generic = (pr.pad & K2A_SSEC_GENERIC) != 0;
	const uint8_t *qry = seq + pr.qoff, *tgt = seq + pr.toff;
	uint8_t *tbp = tb + pr.tb_off;
	uint8_t *U = LDS ? k2a_ssec_lds : scratch + (size_t)pr.bnd_off * 16, *V = U + T16, *X = V + T16, *Y = X + T16;
	uint8_t *X2 = DUAL ? Y + T16 : Y, *Y2 = DUAL ? X2 + T16 : Y, *S = (DUAL ? Y2 : Y) + T16;
	int32_t *H = (int32_t*)(S + T16);
	const int slope = DUAL ? P.e2 : P.e;

	/* the reference's allocation: zeros (ksw2_extz2_sse.c:84) or the gap-open differences (ksw2_extd2_sse.c:111-116), H = -inf */
	for (int x = lane; x < T16; x += 64) {
		const uint8_t g1 = DUAL ? (uint8_t)(-P.q - P.e) : 0, g2 = (uint8_t)(-P.q2 - P.e2);
		U[x] = g1; V[x] = g1; X[x] = g1; Y[x] = g1; S[x] = 0;
		if (DUAL) { X2[x] = g2; Y2[x] = g2; }
		H[x] = K2A_NEG;
	}
	K2A_SSEC_SYNC();

	K2aBook book;
	k2a_book_reset(&book);
	K2aSsecFollow fol = { 0, 0 };
	int last_st = -1, last_en = -1;
	for (int r = 0; r < qlen + tlen - 1; ++r) {
		int st0, en0, st, en;
		if (!k2a_ssec_bounds(r, qlen, tlen, w, st0, en0, st, en)) { book.dropped = 1; break; }      /* ksw2_extz2_sse.c:111-114 */
		/* what the first position of the blocks reads to its left (uniform loads), the first-column cell */
		int cx, cv, cx2 = 0;
		const bool prev_ok = st > 0 && st - 1 >= last_st && st - 1 <= last_en;
		if (!DUAL) {
			cx = prev_ok ? k2a_s8(__builtin_amdgcn_readfirstlane((int)X[st - 1])) : 0;
			cv = st > 0 ? (prev_ok ? k2a_s8(__builtin_amdgcn_readfirstlane((int)V[st - 1])) : 0) : (r ? P.q : 0);
			if (en >= r && lane == 0) { Y[r] = 0; U[r] = (uint8_t)(r ? P.q : 0); }
		} else {
			const int edge = k2a_ssec_edge(P, r);
			cx = prev_ok ? k2a_s8(__builtin_amdgcn_readfirstlane((int)X[st - 1])) : -P.q - P.e;
			cx2 = prev_ok ? k2a_s8(__builtin_amdgcn_readfirstlane((int)X2[st - 1])) : -P.q2 - P.e2;
			cv = st > 0 ? (prev_ok ? k2a_s8(__builtin_amdgcn_readfirstlane((int)V[st - 1])) : -P.q - P.e) : edge;
			if (en >= r && lane == 0) { Y[r] = (uint8_t)(-P.q - P.e); Y2[r] = (uint8_t)(-P.q2 - P.e2); U[r] = (uint8_t)edge; }
		}
		/* scores in runs of 16 from st0 (:125-140); bytes past the padded target length would land in the reference's target copy
		 * at positions no later anti-diagonal reads (below st0), so they are dropped */
		{
			const int pend = generic ? en0 + 1 : st0 + ((en0 - st0) / 16 + 1) * 16;
			for (int p = st0 + lane; p < pend; p += 64)
				if (p < T16) S[p] = (uint8_t)k2a_ssec_score(P, generic, k2a_ssec_tcode(tgt, qry, tlen, qlen, T16, p), k2a_ssec_qcode(qry, r, p));
		}
		K2A_SSEC_SYNC();
		for (int base = st; base <= en; base += 64) {
			const int p = base + lane;
			const bool act = p <= en;
			int xo = 0, vo = 0, x2o = 0, uo = 0, yo = 0, y2o = 0, so = 0;
			if (act) { xo = k2a_s8(X[p]); vo = k2a_s8(V[p]); uo = k2a_s8(U[p]); yo = k2a_s8(Y[p]); so = k2a_s8(S[p]); if (DUAL) { x2o = k2a_s8(X2[p]); y2o = k2a_s8(Y2[p]); } }
			const int xt1 = k2a_shr1_carry(xo, cx), vt1 = k2a_shr1_carry(vo, cv), x2t1 = DUAL ? k2a_shr1_carry(x2o, cx2) : 0;
			cx = __builtin_amdgcn_readlane(xo, 63); cv = __builtin_amdgcn_readlane(vo, 63);
			if (DUAL) cx2 = __builtin_amdgcn_readlane(x2o, 63);
			int un, vn, xn, yn, x2n, y2n;
			uint32_t dir;
			k2a_ssec_cell<DUAL, MODE>(P, so, xt1, vt1, x2t1, uo, yo, y2o, un, vn, xn, yn, x2n, y2n, dir);
			if (act) {
				U[p] = (uint8_t)un; V[p] = (uint8_t)vn; X[p] = (uint8_t)xn; Y[p] = (uint8_t)yn;
				if (DUAL) { X2[p] = (uint8_t)x2n; Y2[p] = (uint8_t)y2n; }
				if (MODE != K2A_MODE_SCORE) tbp[(size_t)r * ncol + (p - st)] = (uint8_t)dir;
			}
		}
		K2A_SSEC_SYNC();
		int stop;
		if (!approx) {
			int A, Sv, T0 = K2A_NEG, T1 = K2A_NEG, T2 = K2A_NEG, bH = K2A_NEG, bT = -1;
			const int en1 = st0 + (en0 - st0) / 4 * 4;
			if (r > 0) {
				/* H of the last in-band position first, from its neighbour's value of the previous anti-diagonal (:229) */
				const int hprev = __builtin_amdgcn_readfirstlane(en0 > 0 ? H[en0 - 1] : H[en0]);
				const int dl = __builtin_amdgcn_readfirstlane((int)(en0 > 0 ? U[en0] : V[en0]));
				A = hprev + k2a_ssec_dh<DUAL>(P, dl);
				Sv = A;
				for (int t0 = st0; t0 < en0; t0 += 64) {
					const int t = t0 + lane;
					int h = K2A_NEG;
					if (t < en0) {
						h = H[t] + k2a_ssec_dh<DUAL>(P, (int)V[t]);
						H[t] = h;
						if (t < en1 && h > bH) { bH = h; bT = t; }
					}
					if (!LDS) {
						if (t0 == st0) Sv = __builtin_amdgcn_readlane(h, 0);
						if (en1 < en0 && en1 >= t0 && en1 < t0 + 64) T0 = __builtin_amdgcn_readlane(h, (en1 - t0) & 63);
						if (en1 + 1 < en0 && en1 + 1 >= t0 && en1 + 1 < t0 + 64) T1 = __builtin_amdgcn_readlane(h, (en1 + 1 - t0) & 63);
						if (en1 + 2 < en0 && en1 + 2 >= t0 && en1 + 2 < t0 + 64) T2 = __builtin_amdgcn_readlane(h, (en1 + 2 - t0) & 63);
					}
				}
				if (LDS) {
					/* the first cell and the (<= 3) tail cells by position, straight out of the row just written (four lanes, one
					 * read) instead of range-tested v_readlane in every block: scalar instructions are half of this kernel */
					__builtin_amdgcn_wave_barrier();
					const int pos = lane == 0 ? st0 : en1 + lane - 1;
					const int hv = (lane < 4 && pos < en0) ? H[pos] : K2A_NEG;
					__builtin_amdgcn_wave_barrier();
					if (st0 < en0) Sv = __builtin_amdgcn_readlane(hv, 0);
					T0 = __builtin_amdgcn_readlane(hv, 1); T1 = __builtin_amdgcn_readlane(hv, 2); T2 = __builtin_amdgcn_readlane(hv, 3);
				}
				if (lane == 0) H[en0] = A;
			} else {
				A = Sv = k2a_ssec_dh<DUAL>(P, __builtin_amdgcn_readfirstlane((int)V[0])) - (DUAL ? P.qe_first : P.q + P.e);
				if (lane == 0) H[0] = A;
			}
			const uint64_t Bkey = k2a_wave_max_u64(bT >= 0 ? k2a_dm_key(bH, bT, st0) : 0ull);
			stop = k2a_ssec_book(&book, r, st0, en0, en, qlen, tlen, pr.zdrop, slope, A, Bkey, T0, T1, T2, Sv);
		} else {
			const int l0 = min(max(fol.last, 0), T16 - 1), l1 = min(max(fol.last + 1, 0), T16 - 1);
			const int vl = __builtin_amdgcn_readfirstlane((int)V[l0]), un = __builtin_amdgcn_readfirstlane((int)U[l1]);
			const int v0 = __builtin_amdgcn_readfirstlane((int)V[0]);
			stop = k2a_ssec_follow<DUAL>(P, fol, &book, r, st0, en0, qlen, tlen, pr.zdrop, adrop, vl, un, v0);
		}
		K2A_SSEC_SYNC();
		if (stop) break;
		last_st = st; last_en = en;
	}
	if (lane == 0) k2a_finish(pr, book, &res[pi]);
}

/* The same mode with the state in registers (ksw2_lane_ssecb.h): lane <-> one 16-position block of the reference's arrays (a ring
 * of 64 blocks), one step per anti-diagonal, H in a 1 024-entry LDS ring per wavefront.  Score-only tasks, simple scoring, bands up
 * to K2A_SSECB_SPAN positions.  What is uniform per anti-diagonal (the band's bounds, the first block's left edge, the cell at
 * position r, the book) is computed by every lane on scalar registers; the values it needs by position come from the owning lane
 * through v_readlane (u, v bytes) or from the LDS ring (H). */
#define K2A_SSECB_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)
/* the anti-diagonals of one task; APPROX: the approximate modes' one followed cell instead of H (a template parameter and not the
 * task's flag tested per anti-diagonal: with both forms in one loop hipcc merges their stores into the book through a selected
 * address and the book moves to scratch memory) */
template<bool DUAL, bool APPROX, int MODE, bool QWILD>
__device__ __forceinline__ void k2a_ssec_blk_task(const K2aSsec &P, const K2aPair &pr, const uint8_t *__restrict__ seq, uint8_t *__restrict__ tb, int *hl, int lane, K2aBook &book)
{
	const int qlen = pr.qlen, tlen = pr.tlen_full, w = pr.w, T16 = (tlen + 15) / 16 * 16, ncol = k2a_ssec_ncol(qlen, tlen, w);
	const bool adrop = (pr.pad & K2A_SSEC_APPROX_DROP) != 0;
	const uint8_t *qry = seq + pr.qoff, *tgt = seq + pr.toff;
	const int slope = DUAL ? P.e2 : P.e;

	K2aSsecBlk<DUAL> B;
	B.blk = -1; B.qn = 0;
	B.U = B.V = B.X = B.Y = B.X2 = B.Y2 = B.S = B.P0 = B.P1 = B.QW = k2a_blk{ 0, 0, 0, 0, 0, 0, 0, 0 };
	k2a_book_reset(&book);
	K2aSsecFollow fol = { 0, 0 };
	int last_st = -1, last_en = -1, last_st0 = 0, last_en0 = 0, hprev = 0;
	for (int r = 0; r < qlen + tlen - 1; ++r) {
		int st0, en0, st, en;
		if (!k2a_ssec_bounds(r, qlen, tlen, w, st0, en0, st, en)) { book.dropped = 1; break; }      /* ksw2_extz2_sse.c:111-114 */
		const int pend = min(st0 + (int)(((uint32_t)(en0 - st0) >> 4) + 1u) * 16, T16);       /* the score refresh's end (:125-140), see k2a_ssec_kernel */
		B.shift_query(P, r);
		/* blocks that enter: the lane's next turn in the ring */
		{
			const int need = max(en, pend - 1) >> 4, nb = B.blk < 0 ? lane : B.blk + 64;
			if (nb <= need) {
				B.init_block(P, nb, tgt, tlen, qry, qlen, r);
				int *hp = hl + k2a_ssecb_slot(nb << 4);
#pragma unroll
				for (int s = 0; s < 16; ++s) hp[s] = K2A_NEG;
			}
		}
		B.ask_query(qry, qlen, r);
		const bool act = B.blk >= (st >> 4) && B.blk <= (en >> 4);
		/* exact mode: the block's H and the H the last in-band cell starts from (:229), asked for now and used after the update.  The
		 * ring's entry is that H unless the position was below the band already (the band is one position wide and did not move): then
		 * it is the value carried from the previous anti-diagonal (see advance_H) */
		int hv[16], hnew = 0;
		if (!APPROX && r > 0) {
			K2A_SSECB_SYNC();
			if (act) {
				const int *hp = hl + k2a_ssecb_slot(B.p0());
#pragma unroll
				for (int s = 0; s < 16; ++s) hv[s] = hp[s];
			}
			hnew = hl[k2a_ssecb_slot(en0 > 0 ? en0 - 1 : en0)];
		}
		/* the cell at position r (first column) and what the first block reads to its left */
		const bool prev_ok = st > 0 && st - 1 >= last_st && st - 1 <= last_en;
		int cv, cx, cx2 = 0;
		if (!DUAL) {
			cx = 0; cv = st > 0 ? 0 : (r ? P.q : 0);
			if (en >= r && B.blk == (r >> 4)) { k2a_sb_set(B.Y, r & 15, 0); k2a_sb_set(B.U, r & 15, r ? P.q : 0); }
		} else {
			const int edge = k2a_ssec_edge(P, r);
			cx = -P.q - P.e; cx2 = -P.q2 - P.e2; cv = st > 0 ? -P.q - P.e : edge;
			if (en >= r && B.blk == (r >> 4)) { k2a_sb_set(B.Y, r & 15, -P.q - P.e); k2a_sb_set(B.Y2, r & 15, -P.q2 - P.e2); k2a_sb_set(B.U, r & 15, edge); }
		}
		uint32_t pv = (uint32_t)k2a_rot1<64>((int)B.V[7]), px = (uint32_t)k2a_rot1<64>((int)B.X[7]), px2 = DUAL ? (uint32_t)k2a_rot1<64>((int)B.X2[7]) : 0u;
		if (B.blk == (st >> 4) && !prev_ok) { pv = k2a_sb_c(cv); px = k2a_sb_c(cx); px2 = k2a_sb_c(cx2); }
		B.template refresh_scores<QWILD>(P, st0, pend);
		if (act) {
			uint32_t dirw[4];
			B.template update<MODE>(P, pv, px, px2, dirw);
			if (MODE != K2A_MODE_SCORE)                          /* the block's 16 direction bytes, where the reference's row r has them (k2a_ssec_trace) */
				*(uint4*)(tb + pr.tb_off + (size_t)r * ncol + (size_t)(B.p0() - st)) = make_uint4(dirw[0], dirw[1], dirw[2], dirw[3]);
		}
		int stop;
		if (!APPROX) {
			int A, Sv, T0 = K2A_NEG, T1 = K2A_NEG, T2 = K2A_NEG;
			uint64_t bk = 0;
			const int en1 = st0 + (int)((uint32_t)(en0 - st0) & ~3u);
			if (r > 0) {
				if (!(en0 == last_en0 && en0 - 1 < last_st0 && en0 > 0)) hprev = __builtin_amdgcn_readfirstlane(hnew);
				int dl;
				if (en0 > 0) dl = __builtin_amdgcn_readlane((int)k2a_sb_get(B.U, en0 & 15), (en0 >> 4) & 63);
				else dl = __builtin_amdgcn_readlane((int)k2a_sb_get(B.V, 0), 0);
				A = hprev + k2a_ssec_dh<DUAL>(P, dl);
				if (act) bk = B.advance_H(P, hl, hv, st0, en1);
				K2A_SSECB_SYNC();
				const int pos = lane == 0 ? st0 : en1 + lane - 1;
				const int hvv = (lane < 4 && pos < en0) ? hl[k2a_ssecb_slot(pos)] : K2A_NEG;
				Sv = st0 < en0 ? __builtin_amdgcn_readlane(hvv, 0) : A;
				T0 = __builtin_amdgcn_readlane(hvv, 1); T1 = __builtin_amdgcn_readlane(hvv, 2); T2 = __builtin_amdgcn_readlane(hvv, 3);
				K2A_SSECB_SYNC();
				if (lane == 0) hl[k2a_ssecb_slot(en0)] = A;
			} else {
				A = Sv = k2a_ssec_dh<DUAL>(P, __builtin_amdgcn_readlane((int)k2a_sb_get(B.V, 0), 0)) - (DUAL ? P.qe_first : P.q + P.e);
				if (lane == 0) hl[0] = A;
			}
			K2A_SSECB_SYNC();
			const uint64_t Bkey = k2a_wave_max_u64(bk);
			stop = k2a_ssec_book(&book, r, st0, en0, en, qlen, tlen, pr.zdrop, slope, A, Bkey, T0, T1, T2, Sv);
		} else {
			const int l0 = min(max(fol.last, 0), T16 - 1), l1 = min(max(fol.last + 1, 0), T16 - 1);
			const int vl = __builtin_amdgcn_readlane((int)k2a_sb_get(B.V, l0 & 15), (l0 >> 4) & 63);
			const int un = __builtin_amdgcn_readlane((int)k2a_sb_get(B.U, l1 & 15), (l1 >> 4) & 63);
			const int v0 = r == 0 ? __builtin_amdgcn_readlane((int)k2a_sb_get(B.V, 0), 0) : 0;      /* (the first anti-diagonal's only; a select tree per look-up otherwise) */
			stop = k2a_ssec_follow<DUAL>(P, fol, &book, r, st0, en0, qlen, tlen, pr.zdrop, adrop, vl, un, v0);
		}
		if (stop) break;
		last_st = st; last_en = en; last_st0 = st0; last_en0 = en0;
	}
}

template<bool DUAL, int MODE>
__global__ void __launch_bounds__(64 * K2A_WPB)
k2a_ssec_blk_kernel(const K2aSsec P, const K2aPair *__restrict__ pairs, const uint32_t *__restrict__ order, int ntasks,
                    const uint8_t *__restrict__ seq, uint8_t *__restrict__ tb, K2aResult *__restrict__ res)
{
	__shared__ int hl_all[K2A_WPB][K2A_SSECB_RING_WORDS];
	const int lane = threadIdx.x & 63, wave = k2a_wave_id<true>();
	const int task = blockIdx.x * K2A_WPB + wave;
	if (task >= ntasks) return;                       /* whole wavefronts leave; nobody synchronises below */
	int *hl = hl_all[wave];
	const uint32_t pi = order[task];
	const K2aPair pr = pairs[pi];
	K2aBook book;
	/* does the query hold the wildcard code?  (sixteen bytes per lane and round; the score refresh of a task without one skips the
	 * wildcard select: round 6) */
	bool qw = false;
	{
		const uint8_t *q = seq + pr.qoff;
		const uint32_t wc = (uint32_t)(P.m - 1) * 0x01010101u;
		for (int x = lane * 4; x < pr.qlen; x += 256) {
			uint32_t d;
			__builtin_memcpy(&d, q + x, 4);
			if (x + 4 > pr.qlen) d |= 0xffffff00u << (8 * (pr.qlen - x - 1));      /* bytes past the end: never the wildcard (codes are below 128) */
			const uint32_t e = d ^ wc;                                            /* a zero byte = the wildcard */
			qw |= ((e - 0x01010101u) & ~e & 0x80808080u) != 0;
		}
		qw = __builtin_amdgcn_ballot_w64(qw) != 0;
	}
	if (pr.pad & K2A_SSEC_APPROX) { if (qw) k2a_ssec_blk_task<DUAL, true, MODE, true>(P, pr, seq, tb, hl, lane, book); else k2a_ssec_blk_task<DUAL, true, MODE, false>(P, pr, seq, tb, hl, lane, book); }
	else if (qw) k2a_ssec_blk_task<DUAL, false, MODE, true>(P, pr, seq, tb, hl, lane, book);
	else k2a_ssec_blk_task<DUAL, false, MODE, false>(P, pr, seq, tb, hl, lane, book);
	if (lane == 0) k2a_finish(pr, book, &res[pi]);
}

__global__ void __launch_bounds__(64)
k2a_ssec_trace_kernel(const K2aPair *__restrict__ pairs, const uint32_t *__restrict__ order, int ntasks,
                      const uint8_t *__restrict__ tb, K2aResult *__restrict__ res, uint32_t *__restrict__ cig, int ppw)
{
	if ((int)threadIdx.x >= ppw) return;
	const int t = blockIdx.x * ppw + threadIdx.x;
	if (t >= ntasks) return;
	const uint32_t pi = order[t];
	const K2aPair pr = pairs[pi];
	const int ti = res[pi].ti, tj = res[pi].tj;
	int n = 0;
	if (ti >= 0 && tj >= 0) n = k2a_ssec_trace(tb + pr.tb_off, k2a_ssec_ncol(pr.qlen, pr.tlen_full, pr.w), ti, tj, cig + pr.cig_off, pr.qlen, pr.tlen_full, pr.w);
	res[pi].n_cigar = n;
}

/* ---------------------------------------------------------------- gap-linear X-drop extension (ksw2_lane_extf.h) */

/* One alignment per wavefront (one wavefront per workgroup: the LDS a workgroup asks for decides how many share a CU).
 * STATE_HBM = false: U, V, S (3 x padded target length bytes) in dynamic LDS; true: in `scratch` at pairs[i].tb_off. */
template<bool STATE_HBM>
__global__ void __launch_bounds__(64)
k2a_extf_kernel(const K2aExtf par, const K2aPair *__restrict__ pairs, const uint32_t *__restrict__ order, int ntasks,
                const uint8_t *__restrict__ seq, uint8_t *__restrict__ scratch, K2aResult *__restrict__ res)
{
	extern __shared__ uint8_t k2a_extf_lds[];
	const int lane = threadIdx.x;
	const uint32_t pi = order[blockIdx.x];
	const K2aPair pr = pairs[pi];
	const int qlen = pr.qlen, tlen = pr.tlen, w = pr.w, xdrop = pr.zdrop;
	const int tpad = (tlen + 15) & ~15;
	const uint8_t *qa = seq + pr.qoff, *ta = seq + pr.toff;
	uint8_t *U = STATE_HBM ? scratch + pr.tb_off : k2a_extf_lds, *V = U + tpad, *S = V + tpad;
	const uint32_t two_e = (uint32_t)(par.e * 2) & 0xffu;

	for (int x = lane * 4; x < 3 * tpad; x += 256) *(uint32_t*)(U + x) = 0u;      /* the reference's kcalloc (ksw2_extf2_sse.c:25) */
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");

	K2aExtfBook bk;
	k2a_extf_book_reset(bk);
	int prev_lo = -1, prev_hi = -1, r;
	const int nr = qlen + tlen - 1;
	for (r = 0; r < nr; ++r) {
		K2aExtfDiag d;
		if (!k2a_extf_diag(r, qlen, tlen, w, tpad, d)) break;
		int carry = (d.blo > 0 && d.blo - 1 >= prev_lo && d.blo - 1 <= prev_hi) ? __builtin_amdgcn_readfirstlane((int)V[d.blo - 1]) : 0;
		const int last = k2a_max(d.bhi, d.fresh_end - 1);
		const bool top0 = d.bhi >= r;                       /* ksw2_extf2_sse.c:46: U of the anti-diagonal's first-row cell reads 0 */
		for (int base = d.blo; base <= last; base += 64) {
			const int x = base + lane;
			const bool act = x <= d.bhi, fresh = x >= d.lo && x < d.fresh_end;
			uint32_t vold = 0, b = 0, sv = 0;
			if (act) { vold = V[x]; b = U[x]; }
			if (fresh) sv = k2a_extf_score(par, qa, ta, qlen, tlen, r, x);
			else if (act) sv = S[x];
			if (top0 && x == r) b = 0;
			const uint32_t a = (uint32_t)k2a_shr1_carry((int)vold, carry);
			carry = __builtin_amdgcn_readlane((int)vold, 63);
			uint32_t u, v;
			k2a_extf_cell(sv, a, b, two_e, u, v);
			if (act) { U[x] = (uint8_t)u; V[x] = (uint8_t)v; }
			if (fresh) S[x] = (uint8_t)sv;
		}
		/* positions move between lanes from one anti-diagonal to the next (blo moves in steps of 16) */
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
		const uint32_t vf = (uint32_t)__builtin_amdgcn_readfirstlane((int)V[bk.follow]);
		const uint32_t un = (uint32_t)__builtin_amdgcn_readfirstlane((int)U[bk.follow + 1]);
		if (!k2a_extf_follow(bk, d, r, par.e, xdrop, vf, un)) break;
		prev_lo = d.blo; prev_hi = d.bhi;
	}
	if (lane == 0) k2a_extf_finish(bk, r, nr, &res[pi]);
}

/* One extension per lane (ksw2_lane_extf.h, K2aExtfLaneMem): task t of the launch = lane t & 63 of wavefront t >> 6; the pairs of a
 * wavefront share one group block -- pairs[i].toff / qoff = byte offsets of the group's interleaved target / reversed-query codes
 * in `seq`, tb_off = its zeroed state rows in `scratch`, pad = rows (dwords per lane) of each state array.  Lanes run their own
 * anti-diagonal loops; the wavefront iterates until its last lane is done. */
__global__ void __launch_bounds__(64 * K2A_WPB)
k2a_extf_lane_kernel(const K2aExtf par, const K2aPair *__restrict__ pairs, const uint32_t *__restrict__ order, int ntasks,
                     const uint8_t *__restrict__ seq, uint8_t *__restrict__ scratch, K2aResult *__restrict__ res)
{
	const int lane = threadIdx.x & 63, wave = k2a_wave_id<false>();
	const int task = (blockIdx.x * K2A_WPB + wave) * 64 + lane;
	const bool valid = task < ntasks;
	const uint32_t pi = order[valid ? task : 0];
	const K2aPair pr = pairs[pi];
	const int qlen = pr.qlen, tlen = pr.tlen, w = pr.w, xdrop = pr.zdrop, tpad = (tlen + 15) & ~15;
	const size_t rows = (size_t)pr.pad * 64;                    /* dwords per state array of the group */
	K2aExtfLaneMem m;
	m.U4 = (uint32_t*)(scratch + pr.tb_off) + lane; m.V4 = m.U4 + rows; m.S4 = m.V4 + rows;
	m.TT = (const uint32_t*)(seq + pr.toff) + lane; m.QR = (const uint32_t*)(seq + pr.qoff) + lane;
	m.ring = 0; m.ztop = 0;
	K2aExtfBook bk;
	k2a_extf_book_reset(bk);
	int prev_lo = -1, prev_hi = -1, r = 0;
	const int nr = qlen + tlen - 1;
	bool go = valid;
	while (__builtin_amdgcn_ballot_w64(go && r < nr) != 0) {
		if (go && r < nr) {
			if (k2a_extf_lane_diag(par, qlen, tlen, w, tpad, xdrop, r, m, prev_lo, prev_hi, bk)) ++r;
			else go = false;
		}
	}
	if (valid) k2a_extf_finish(bk, r, nr, &res[pi]);
}

/* The same with the state arrays in LDS: a lane only ever touches the few dozen rows around its band (K2A_EXTF_RING_ROWS), so
 * they live in a ring of par.ring rows per array -- 3 x ring x 256 bytes per wavefront, one wavefront per workgroup -- instead of
 * whole arrays in HBM scratch, whose re-reading on every anti-diagonal was ~9 bytes of HBM traffic per cell (round 2:
 * profiles/r2z_extf-lane_pmc.json, 488 GB per launch of 262 144 extensions).  Target codes and the reversed query still come from
 * the interleaved blocks in global memory (read only). */
__global__ void __launch_bounds__(64)
k2a_extf_lane_ring_kernel(const K2aExtf par, const K2aPair *__restrict__ pairs, const uint32_t *__restrict__ order, int ntasks,
                          const uint8_t *__restrict__ seq, K2aResult *__restrict__ res)
{
	extern __shared__ uint32_t k2a_ring[];
	const int lane = threadIdx.x & 63;
	const int task = blockIdx.x * 64 + lane;
	const bool valid = task < ntasks;
	const uint32_t pi = order[valid ? task : 0];
	const K2aPair pr = pairs[pi];
	const int qlen = pr.qlen, tlen = pr.tlen, w = pr.w, xdrop = pr.zdrop, tpad = (tlen + 15) & ~15;
	K2aExtfLaneMem m;
	m.ring = par.ring; m.ztop = 0;
	m.U4 = k2a_ring + lane; m.V4 = m.U4 + (size_t)par.ring * 64; m.S4 = m.V4 + (size_t)par.ring * 64;
	m.TT = (const uint32_t*)(seq + pr.toff) + lane; m.QR = (const uint32_t*)(seq + pr.qoff) + lane;
	K2aExtfBook bk;
	k2a_extf_book_reset(bk);
	int prev_lo = -1, prev_hi = -1, r = 0;
	const int nr = qlen + tlen - 1;
	bool go = valid;
	while (__builtin_amdgcn_ballot_w64(go && r < nr) != 0) {
		if (go && r < nr) {
			if (k2a_extf_lane_diag(par, qlen, tlen, w, tpad, xdrop, r, m, prev_lo, prev_hi, bk)) ++r;
			else go = false;
		}
	}
	if (valid) k2a_extf_finish(bk, r, nr, &res[pi]);
}

/* Register-window form: bands up to K2A_EXTF_WIN_SPAN(K) positions wide.  U, V, S and the target code of K x 64 positions
 * stay in registers; the window slides up a 64-block at a time (a block entering it is all zero, like the reference's
 * fresh allocation); no LDS, no fences.  Slots are visited in a fixed order: every slot's old V of lane 63 is taken first,
 * so slot s finds its left neighbour in slot s - 1 whatever block that holds. */
template<int K>
__global__ void __launch_bounds__(64 * K2A_WPB)
k2a_extf_win_kernel(const K2aExtf par, const K2aPair *__restrict__ pairs, const uint32_t *__restrict__ order, int ntasks,
                    const uint8_t *__restrict__ seq, K2aResult *__restrict__ res)
{
	const int lane = threadIdx.x & 63, wave = k2a_wave_id<false>();
	const int task = blockIdx.x * K2A_WPB + wave;
	if (task >= ntasks) return;
	const uint32_t pi = order[task];
	const K2aPair pr = pairs[pi];
	const int qlen = pr.qlen, tlen = pr.tlen, w = pr.w, xdrop = pr.zdrop;
	const int tpad = (tlen + 15) & ~15;
	const uint8_t *qa = seq + pr.qoff, *ta = seq + pr.toff;
	const uint32_t two_e = (uint32_t)(par.e * 2) & 0xffu;

	uint32_t U[K], V[K], S[K], T[K];
	int wb_cur = 0;
#pragma unroll
	for (int s = 0; s < K; ++s) { U[s] = V[s] = S[s] = 0u; T[s] = 64 * s + lane < tlen ? ta[64 * s + lane] : 0u; }

	K2aExtfBook bk;
	k2a_extf_book_reset(bk);
	int prev_lo = -1, prev_hi = -1, r;
	const int nr = qlen + tlen - 1;
	for (r = 0; r < nr; ++r) {
		K2aExtfDiag d;
		if (!k2a_extf_diag(r, qlen, tlen, w, tpad, d)) break;
		const int wb = __builtin_amdgcn_readfirstlane(k2a_extf_win_base(d));
		const bool carry_ok = d.blo > 0 && d.blo - 1 >= prev_lo && d.blo - 1 <= prev_hi;
		const int last = k2a_max(d.bhi, d.fresh_end - 1);
		if (wb != wb_cur) {                                   /* a block left the window: its slot takes the next one above */
#pragma unroll
			for (int s = 0; s < K; ++s) {
				const int nb = k2a_extf_win_block<K>(wb, s);
				if (nb != k2a_extf_win_block<K>(wb_cur, s)) {
					U[s] = V[s] = S[s] = 0u;
					T[s] = 64 * nb + lane < tlen ? ta[64 * nb + lane] : 0u;
				}
			}
			wb_cur = wb;
		}
		int c63[K];
#pragma unroll
		for (int s = 0; s < K; ++s) c63[s] = __builtin_amdgcn_readlane((int)V[s], 63);
		const int jl = r - lane;                               /* query index of the lane's position in block 0 */
#pragma unroll
		for (int s = 0; s < K; ++s) {
			const int base = 64 * k2a_extf_win_block<K>(wb, s);
			if (base > last || base + 63 < d.blo) continue;
			const int x = base + lane, j = jl - base;
			const uint32_t jc = (uint32_t)j < (uint32_t)qlen ? (uint32_t)j : 0u;
			const uint32_t qc = (uint32_t)j < (uint32_t)qlen ? (uint32_t)qa[jc] : 0u;
			const uint32_t vshift = (uint32_t)k2a_shr1_carry((int)V[s], c63[(s + K - 1) & (K - 1)]);
			k2a_extf_win_cell(par, d, r, x, two_e, carry_ok, T[s], qc, vshift, U[s], V[s], S[s]);
		}
		/* the followed cell: V[follow], U[follow + 1] out of their slots (follow >= lo - 1, inside the window) */
		const int fs = (bk.follow >> 6) & (K - 1), gs = ((bk.follow + 1) >> 6) & (K - 1);
		uint32_t vsel = V[0], usel = U[0];
#pragma unroll
		for (int s = 1; s < K; ++s) { if (fs == s) vsel = V[s]; if (gs == s) usel = U[s]; }
		const uint32_t vf = (uint32_t)__builtin_amdgcn_readlane((int)vsel, bk.follow & 63);
		const uint32_t un = (uint32_t)__builtin_amdgcn_readlane((int)usel, (bk.follow + 1) & 63);
		if (!k2a_extf_follow(bk, d, r, par.e, xdrop, vf, un)) break;
		prev_lo = d.blo; prev_hi = d.bhi;
	}
	if (lane == 0) k2a_extf_finish(bk, r, nr, &res[pi]);
}

typedef void (*fill_pk_fn)(const K2aScoring, const K2aPair*, const uint32_t*, int, const uint8_t*, uint8_t*, K2aResult*, K2aQueueDesc*);
#define PK_ROW(G, C, RB, NM) { { k2a_fill_pk_kernel<G, C, false, 0, RB, NM>, k2a_fill_pk_kernel<G, C, false, 1, RB, NM>, k2a_fill_pk_kernel<G, C, false, 2, RB, NM> }, \
                               { k2a_fill_pk_kernel<G, C, true, 0, RB, NM>,  k2a_fill_pk_kernel<G, C, true, 1, RB, NM>,  k2a_fill_pk_kernel<G, C, true, 2, RB, NM> } }
#define PK_SET(NM) { { PK_ROW(8, 18, false, NM), PK_ROW(16, 8, false, NM), PK_ROW(64, 8, false, NM), PK_ROW(64, 16, false, NM) }, \
                     { PK_ROW(8, 18, true, NM),  PK_ROW(16, 8, true, NM),  PK_ROW(64, 8, true, NM),  PK_ROW(64, 16, true, NM) } }
static const fill_pk_fn g_fill_pk[2][2][K2A_NPKCFG][2][3] = { PK_SET(false), PK_SET(true) };     /* [nomax][rebased][cfg][dual][mode] */
/* the QUEUE builds (streamed launches): score-only kernels only -- what the batch entry points stream; [nomax][rebased][cfg][dual] */
#define PKQ_ROW(G, C, RB, NM) { k2a_fill_pk_kernel<G, C, false, 0, RB, NM, 0, false, true>, k2a_fill_pk_kernel<G, C, true, 0, RB, NM, 0, false, true> }
#define PKQ_SET(NM) { { PKQ_ROW(8, 18, false, NM), PKQ_ROW(16, 8, false, NM), PKQ_ROW(64, 8, false, NM), PKQ_ROW(64, 16, false, NM) }, \
                      { PKQ_ROW(8, 18, true, NM),  PKQ_ROW(16, 8, true, NM),  PKQ_ROW(64, 8, true, NM),  PKQ_ROW(64, 16, true, NM) } }
static const fill_pk_fn g_fill_pkq[2][2][K2A_NPKCFG][2] = { PKQ_SET(false), PKQ_SET(true) };
/* the K2A_PK_LDSROWS classes with their row state in LDS: [rebased][mode - 1] */
static const fill_pk_fn g_fill_pk_lds[2][2] = {
	{ k2a_fill_pk_kernel<64, 16, true, 1, false, false, 1>, k2a_fill_pk_kernel<64, 16, true, 2, false, false, 1> },
	{ k2a_fill_pk_kernel<64, 16, true, 1, true, false, 1>,  k2a_fill_pk_kernel<64, 16, true, 2, true, false, 1> } };

/* exact / no-maximum score-only kernels with the code planes in LDS: [geometry: (64, 16), (8, 18)][nomax][rebased] */
#define LDSCODE_SET(G, C) { { k2a_fill_pk_kernel<G, C, false, 0, false, false, 2>, k2a_fill_pk_kernel<G, C, false, 0, true, false, 2> }, \
                            { k2a_fill_pk_kernel<G, C, false, 0, false, true, 2>,  k2a_fill_pk_kernel<G, C, false, 0, true, true, 2> } }
static const fill_pk_fn g_fill_pk_ldscodes[3][2][2] = { LDSCODE_SET(64, 16), LDSCODE_SET(8, 18), LDSCODE_SET(16, 8) };
#define LDSCODEQ_SET(G, C) { { k2a_fill_pk_kernel<G, C, false, 0, false, false, 2, false, true>, k2a_fill_pk_kernel<G, C, false, 0, true, false, 2, false, true> }, \
                             { k2a_fill_pk_kernel<G, C, false, 0, false, true, 2, false, true>,  k2a_fill_pk_kernel<G, C, false, 0, true, true, 2, false, true> } }
static const fill_pk_fn g_fill_pkq_ldscodes[3][2][2] = { LDSCODEQ_SET(64, 16), LDSCODEQ_SET(8, 18), LDSCODEQ_SET(16, 8) };
/* exact score-only single-gap kernels with the arg-max deferred (K2aLanePk, DEFER) and their second pass: [cfg][rebased]; the
 * fill in the form the launcher prefers for big launches of that geometry (code planes in LDS for (16, 8) and (64, 16)) */
#define DEFER_ROW(G, C, LR) { k2a_fill_pk_kernel<G, C, false, 0, false, false, LR, true>, k2a_fill_pk_kernel<G, C, false, 0, true, false, LR, true> }
static const fill_pk_fn g_fill_pk_defer[4][2] = { DEFER_ROW(8, 18, 0), DEFER_ROW(16, 8, 2), DEFER_ROW(64, 8, 0), DEFER_ROW(64, 16, 2) };
#define DEFERQ_ROW(G, C, LR) { k2a_fill_pk_kernel<G, C, false, 0, false, false, LR, true, true>, k2a_fill_pk_kernel<G, C, false, 0, true, false, LR, true, true> }
static const fill_pk_fn g_fill_pkq_defer[4][2] = { DEFERQ_ROW(8, 18, 0), DEFERQ_ROW(16, 8, 2), DEFERQ_ROW(64, 8, 0), DEFERQ_ROW(64, 16, 2) };
typedef void (*argmax_fn)(const K2aScoring, const K2aPair*, const uint32_t*, int, const uint8_t*, uint8_t*, K2aResult*, const K2aQueueDesc*);
#define ARGMAX_ROW(G, C) { k2a_argmax_kernel<G, C, false>, k2a_argmax_kernel<G, C, true> }
static const argmax_fn g_argmax[4][2] = { ARGMAX_ROW(8, 18), ARGMAX_ROW(16, 8), ARGMAX_ROW(64, 8), ARGMAX_ROW(64, 16) };
typedef void (*zscan_fn)(const K2aScoring, const K2aPair*, const uint32_t*, int, const uint8_t*, const uint8_t*, K2aResult*, const K2aQueueDesc*);
#define ZSCAN_ROW(G, C) { k2a_zscan_kernel<G, C, false>, k2a_zscan_kernel<G, C, true> }
static const zscan_fn g_zscan[4][2] = { ZSCAN_ROW(8, 18), ZSCAN_ROW(16, 8), ZSCAN_ROW(64, 8), ZSCAN_ROW(64, 16) };

/* Launch-time kernel forms.  Every choice the launcher makes has a forcing switch (k2a_shim_set_option: -1 automatic, 0 / 1
 * forced; the host maps KSW2AMD_LDSCODES / KSW2AMD_LDSROWS onto it) and is reported by k2a_shim_pk_form / k2a_shim_mp_form, so
 * that tests can pin each form against the oracle and check which one an unforced launch took. */
static int g_opt[K2A_NOPT] = { -1, -1 };

/* code planes in LDS: worth it once SIMDs would hold a third wavefront (a pooled batch launches chunks of two wavefronts
 * per SIMD side by side) */
static bool k2a_use_ldscodes(int waves, int G)
{
	if (g_opt[K2A_OPT_LDSCODES] >= 0) return g_opt[K2A_OPT_LDSCODES] != 0;
	/* (8, 18) -- config 2's geometry -- goes from two to three wavefronts per SIMD this way (224 -> 168 registers) and LOSES:
	 * 65 536 pairs of 512 x 512 are 4 096 wavefronts, two full rounds of two per SIMD but 1.33 rounds of three, and the LDS
	 * reads are not free at that occupancy: 2 751 against 2 859 GCUPS (round 3, same box).  Forced form only (tests, A/B). */
	if (G == 8) return false;
	return 2 * (long)waves >= 3 * (long)k2a_shim_simd_count();
}

/* Row state in LDS (two wavefronts per SIMD) or in registers (one): the LDS form wins as soon as SIMDs hold two
 * wavefronts, the register form when they hold one (measured: config 5, 8 per SIMD, 836 -> 1035 GCUPS; config 4, one per
 * SIMD, 757 -> 640). */
static bool k2a_use_ldsrows(int waves)
{
	if (g_opt[K2A_OPT_LDSROWS] >= 0) return g_opt[K2A_OPT_LDSROWS] != 0;
	return 2 * (long)waves >= 3 * (long)k2a_shim_simd_count();
}
#define TRACE_PK_ROW(D) { k2a_trace_pk_kernel<8, 18, D>, k2a_trace_pk_kernel<16, 8, D>, k2a_trace_pk_kernel<64, 8, D>, k2a_trace_pk_kernel<64, 16, D>, \
                          k2a_trace_pk_kernel<64, 16, D, true> }      /* last: generation-serial layout */
static const trace_fn g_trace_pk[2][K2A_NPKCFG] = { TRACE_PK_ROW(false), TRACE_PK_ROW(true) };      /* [dual][cfg] */


/* 64 / G extensions per wavefront, G = 16 / 32 / 64 lanes each, one 16-position block of U / V / S per lane (ksw2_lane_extfb.h).  What
 * is uniform per extension is computed by its G lanes; a lane's left neighbour is one DPP rotate inside its group (inside a row of
 * 16; of the whole wavefront for G = 64; for G = 32 the whole-wavefront rotate with the two wrap-around lanes patched), the followed
 * cell's two bytes come from their owner lanes through ds_bpermute_b32. */
template<int G>
__device__ __forceinline__ uint32_t k2a_extf_grp_rot(uint32_t v, int lane)
{
	if (G == 16) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x121 /* row_ror:1 */, 0xf, 0xf, false);
	const uint32_t r = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x13C /* wave_ror:1 */, 0xf, 0xf, false);
	if (G == 64) return r;
	const uint32_t v31 = (uint32_t)__builtin_amdgcn_readlane((int)v, 31), v63 = (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
	return lane == 0 ? v31 : lane == 32 ? v63 : r;
}
template<int G>
__global__ void __launch_bounds__(64 * K2A_WPB)
k2a_extf_grp_kernel(const K2aExtf par, const K2aPair *__restrict__ pairs, const uint32_t *__restrict__ order, int ntasks,
                    const uint8_t *__restrict__ seq, K2aResult *__restrict__ res)
{
	constexpr int NG = 64 / G;
	const int lane = threadIdx.x & 63, wave = k2a_wave_id<false>(), gl = lane & (G - 1);
	const int task0 = (blockIdx.x * K2A_WPB + wave) * NG, task = task0 + lane / G;
	if (task0 >= ntasks) return;
	const bool live = task < ntasks;
	const uint32_t pi = order[live ? task : ntasks - 1];
	const K2aPair pr = pairs[pi];
	K2aExtfBlk<G> B;
	B.start(par, pr, seq, gl, live);
	for (int r = 0; ; ++r) {
		const bool on = B.begin(par, r);
		if (__builtin_amdgcn_ballot_w64(on) == 0) break;
		B.ask(r);
		const uint32_t pv = k2a_extf_grp_rot<G>(B.V[7], lane);
		uint32_t vsel, usel;
		B.update(par, pv, vsel, usel);
		const uint32_t vf = (uint32_t)__builtin_amdgcn_ds_bpermute(((lane & ~(G - 1)) | B.vlane()) << 2, (int)vsel);
		const uint32_t un = (uint32_t)__builtin_amdgcn_ds_bpermute(((lane & ~(G - 1)) | B.ulane()) << 2, (int)usel);
		B.finish_diag(par, r, vf, un);
	}
	if (live && gl == 0) k2a_extf_finish(B.bk, B.rdone, B.nr, &res[pi]);
}

extern "C" {

const char *k2a_shim_backend(void) { return "hip:gfx950"; }

void k2a_shim_set_option(int opt, int value) { if (opt >= 0 && opt < K2A_NOPT) g_opt[opt] = value < 0 ? -1 : value != 0; }

int k2a_shim_pk_form(int cfg, int dual, int mode, int nomax, int ntasks)
{
	if (cfg < 0 || cfg >= K2A_NPKCFG) return 0;
	const int per_wave = 64 / k2a_pkcfg_G[cfg], waves = (ntasks + per_wave - 1) / per_wave;
	if (K2A_PK_LDSROWS(k2a_pkcfg_G[cfg], k2a_pkcfg_C[cfg], dual, mode, nomax) && k2a_use_ldsrows(waves)) return 1;
	if (K2A_PK_LDSCODES(k2a_pkcfg_G[cfg], k2a_pkcfg_C[cfg], dual, mode, nomax) && k2a_use_ldscodes(waves, k2a_pkcfg_G[cfg])) return 2;
	return 0;
}

int k2a_shim_mp_form(int dual, int mode, int ntasks) { return !dual && mode != K2A_MODE_SCORE && k2a_use_ldsrows(ntasks); }
const char *k2a_shim_last_error(void) { return g_err; }

int k2a_shim_device_count(void)
{
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess) return 0;
	return n;
}

int k2a_shim_simd_count(void)
{
	static int simds[64];
	int dev = 0, cus = 0;
	if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
	if (simds[dev]) return simds[dev];
	if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
	return simds[dev] = 4 * cus;
}

/* "dddd:bb:dd.f" of the current device (its directory name under /sys/bus/pci/devices), or -1 */
int k2a_shim_pci_bus_id(char *buf, int cap)
{
	int dev = 0;
	if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetPCIBusId(buf, cap, dev) != hipSuccess) return -1;
	for (char *c = buf; *c; ++c) if (*c >= 'A' && *c <= 'F') *c = (char)(*c - 'A' + 'a');
	return 0;
}

int k2a_shim_set_device(int dev) { CHECK(hipSetDevice(dev)); return 0; }
int k2a_shim_get_device(void) { int dev = -1; return hipGetDevice(&dev) == hipSuccess ? dev : -1; }

int k2a_shim_mem_info(size_t *free_b, size_t *total_b) { CHECK(hipMemGetInfo(free_b, total_b)); return 0; }

void *k2a_shim_malloc(size_t bytes)
{
	void *p = 0;
	if (set_err(hipMalloc(&p, bytes ? bytes : 16), "hipMalloc")) return 0;
	return p;
}
void k2a_shim_free(void *p) { if (p) (void)hipFree(p); }

void *k2a_shim_host_malloc(size_t bytes)
{
	void *p = 0;
	if (set_err(hipHostMalloc(&p, bytes ? bytes : 16, hipHostMallocDefault), "hipHostMalloc")) return 0;
	return p;
}
void k2a_shim_host_free(void *p) { if (p) (void)hipHostFree(p); }

int k2a_shim_async_launches(void) { return 1; }
int k2a_shim_h2d(void *dst, const void *src, size_t bytes, void *stream)
{
	if (bytes) CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, (hipStream_t)stream));
	return 0;
}
int k2a_shim_d2h(void *dst, const void *src, size_t bytes, void *stream)
{
	if (bytes) CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, (hipStream_t)stream));
	return 0;
}
int k2a_shim_d2d(void *dst, const void *src, size_t bytes, void *stream)
{
	if (bytes) CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream));
	return 0;
}
int k2a_shim_host_register(void *p, size_t bytes) { CHECK(hipHostRegister(p, bytes, hipHostRegisterDefault)); return 0; }
int k2a_shim_host_unregister(void *p) { CHECK(hipHostUnregister(p)); return 0; }
int k2a_shim_memset(void *dst, int v, size_t bytes, void *stream)
{
	if (bytes) CHECK(hipMemsetAsync(dst, v, bytes, (hipStream_t)stream));
	return 0;
}

void *k2a_shim_stream_create(void)
{
	hipStream_t s = 0;
	if (set_err(hipStreamCreateWithFlags(&s, hipStreamNonBlocking), "hipStreamCreate")) return 0;
	return (void*)s;
}
/* a stream of the highest priority the device offers: the runtime keeps a pool of hardware queues PER PRIORITY and deals a new stream the
 * least used queue of its pool, so a handful of these get hardware queues of their own, whatever else the process has created */
void *k2a_shim_stream_create_high(void)
{
	hipStream_t s = 0;
	int lo = 0, hi = 0;
	if (hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess) { lo = hi = 0; }
	if (set_err(hipStreamCreateWithPriority(&s, hipStreamNonBlocking, hi), "hipStreamCreateWithPriority")) return 0;
	return (void*)s;
}
/* ... and of the lowest: the per-device upload streams.  They carry DMA copies and events only, and a copy must never sit in a hardware
 * queue behind a kernel of a streamed launch that waits for that very copy (two ordinary streams may share a queue, whichever the
 * runtime deals them: the launch then only ends by its timeout) */
void *k2a_shim_stream_create_low(void)
{
	hipStream_t s = 0;
	int lo = 0, hi = 0;
	if (hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess) { lo = hi = 0; }
	if (set_err(hipStreamCreateWithPriority(&s, hipStreamNonBlocking, lo), "hipStreamCreateWithPriority")) return 0;
	return (void*)s;
}
void k2a_shim_stream_destroy(void *stream) { if (stream) (void)hipStreamDestroy((hipStream_t)stream); }
int k2a_shim_stream_sync(void *stream) { CHECK(hipStreamSynchronize((hipStream_t)stream)); return 0; }

void *k2a_shim_event_create(void)
{
	hipEvent_t e = 0;
	if (set_err(hipEventCreate(&e), "hipEventCreate")) return 0;
	return (void*)e;
}
void k2a_shim_event_destroy(void *ev) { if (ev) (void)hipEventDestroy((hipEvent_t)ev); }
int k2a_shim_event_record(void *ev, void *stream) { CHECK(hipEventRecord((hipEvent_t)ev, (hipStream_t)stream)); return 0; }
int k2a_shim_stream_wait_event(void *stream, void *ev) { CHECK(hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)ev, 0)); return 0; }
int k2a_shim_event_sync(void *ev) { CHECK(hipEventSynchronize((hipEvent_t)ev)); return 0; }
float k2a_shim_event_ms(void *start, void *stop)
{
	float ms = -1.0f;
	if (hipEventSynchronize((hipEvent_t)stop) != hipSuccess) return -1.0f;
	if (hipEventElapsedTime(&ms, (hipEvent_t)start, (hipEvent_t)stop) != hipSuccess) return -1.0f;
	return ms;
}

int k2a_shim_launch_fill(int cfg, int dual, int mode, const K2aScoring *sc, const K2aPair *pairs, const uint32_t *order,
                         int ntasks, const uint8_t *seq, uint8_t *tb, int32_t *bnd, K2aResult *res, void *stream)
{
	if (ntasks <= 0) return 0;
	if (cfg < 0 || cfg >= K2A_NCFG || mode < 0 || mode > 2) { snprintf(g_err, sizeof(g_err), "bad kernel class"); return -1; }
	const int per_block = K2A_WPB * (64 / k2a_cfg_G[cfg]);
	const int blocks = (ntasks + per_block - 1) / per_block;
	if (cfg == K2A_CFG_MP)
		hipLaunchKernelGGL(k2a_shim_mp_form(dual, mode, ntasks) ? g_fill_mp_lds[mode - 1] : g_fill_mp[dual ? 1 : 0][mode],
		                   dim3(blocks), dim3(64 * K2A_WPB), 0, (hipStream_t)stream, *sc, pairs, order, ntasks, seq, tb, bnd, res);
	else
		hipLaunchKernelGGL(g_fill[cfg][dual ? 1 : 0][mode], dim3(blocks), dim3(64 * K2A_WPB), 0, (hipStream_t)stream,
		                   *sc, pairs, order, ntasks, seq, tb, res);
	CHECK(hipGetLastError());
	return 0;
}

int k2a_shim_launch_trace(int cfg, int dual, const K2aPair *pairs, const uint32_t *order, int ntasks, const uint8_t *tb,
                          K2aResult *res, uint32_t *cig, void *stream)
{
	if (ntasks <= 0) return 0;
	if (cfg < 0 || cfg >= K2A_NCFG) { snprintf(g_err, sizeof(g_err), "bad kernel class"); return -1; }
	const int ppw = k2a_trace_ppw(ntasks);
	hipLaunchKernelGGL(g_trace[cfg][dual ? 1 : 0], dim3((ntasks + ppw - 1) / ppw), dim3(64), 0, (hipStream_t)stream,
	                   pairs, order, ntasks, tb, res, cig, ppw);
	CHECK(hipGetLastError());
	return 0;
}

int k2a_shim_launch_fill_pk(int cfg, int dual, int mode, int rebased, int nomax, int defer, const K2aScoring *sc, const K2aPair *pairs, const uint32_t *order2,
                            int ntasks, const uint8_t *seq, uint8_t *tb, K2aResult *res, K2aQueueDesc *qd, void *stream)
{
	if (ntasks <= 0) return 0;
	if (cfg < 0 || cfg >= K2A_NPKCFG || mode < 0 || mode > 2) { snprintf(g_err, sizeof(g_err), "bad packed kernel class"); return -1; }
	const int per_block = K2A_WPB * (64 / k2a_pkcfg_G[cfg]);
	const int blocks = (ntasks + per_block - 1) / per_block;
	if (defer) {
		/* deferred arg-max: the fill streams its checkpoints into `tb` (block offsets in K2aPair.tb_off / bnd_off / cig_off),
		 * the second pass fills in max_q / mte_q */
		if (cfg >= K2A_PKCFG_MP || dual || mode != K2A_MODE_SCORE || nomax || !tb) { snprintf(g_err, sizeof(g_err), "bad deferred arg-max class"); return -1; }
		const fill_pk_fn fn = (qd ? g_fill_pkq_defer : g_fill_pk_defer)[cfg][rebased ? 1 : 0];
		hipLaunchKernelGGL(fn, dim3(blocks), dim3(64 * K2A_WPB), 0, (hipStream_t)stream, *sc, pairs, order2, ntasks, seq, tb, res, qd);
		CHECK(hipGetLastError());
		hipLaunchKernelGGL(g_argmax[cfg][rebased ? 1 : 0], dim3((3 * ntasks + 64 * K2A_WPB - 1) / (64 * K2A_WPB)), dim3(64 * K2A_WPB), 0, (hipStream_t)stream,
		                   *sc, pairs, order2, ntasks, seq, tb, res, (const K2aQueueDesc*)qd);
		CHECK(hipGetLastError());
		/* third pass: one group of 16 lanes per frozen book; the grid covers every alignment of the class, wavefronts beyond the
		 * list's length leave at once (the list is filled by the kernel in front: its length is not known here) */
		hipLaunchKernelGGL(g_zscan[cfg][rebased ? 1 : 0], dim3((2 * ntasks + 4 * K2A_WPB - 1) / (4 * K2A_WPB)), dim3(64 * K2A_WPB), 0, (hipStream_t)stream,
		                   *sc, pairs, order2, ntasks, seq, (const uint8_t*)tb, res, (const K2aQueueDesc*)qd);
		CHECK(hipGetLastError());
		return 0;
	}
	const int form = k2a_shim_pk_form(cfg, dual, mode, nomax, ntasks);
	const bool lds = form == 1, ldc = form == 2;
	if (qd && (mode != K2A_MODE_SCORE || lds)) { snprintf(g_err, sizeof(g_err), "streamed launches exist for the score-only packed kernels only"); return -1; }
	const fill_pk_fn fn = lds ? g_fill_pk_lds[rebased ? 1 : 0][mode - 1]
	                    : ldc ? (qd ? g_fill_pkq_ldscodes : g_fill_pk_ldscodes)[k2a_pkcfg_G[cfg] == 8 ? 1 : k2a_pkcfg_G[cfg] == 16 ? 2 : 0][nomax ? 1 : 0][rebased ? 1 : 0]
	                    : qd ? g_fill_pkq[nomax ? 1 : 0][rebased ? 1 : 0][cfg][dual ? 1 : 0] : g_fill_pk[nomax ? 1 : 0][rebased ? 1 : 0][cfg][dual ? 1 : 0][mode];
	hipLaunchKernelGGL(fn, dim3(blocks), dim3(64 * K2A_WPB), 0, (hipStream_t)stream,
	                   *sc, pairs, order2, ntasks, seq, tb, res, qd);
	CHECK(hipGetLastError());
	return 0;
}

__global__ void __launch_bounds__(256)
k2a_gather_kernel(const K2aGather *__restrict__ tab, uint8_t *__restrict__ dst)
{
	const K2aGather g = tab[blockIdx.x];
	const uint8_t *src = (const uint8_t*)(uintptr_t)g.src;
	uint8_t *d = dst + g.dst;
	for (uint32_t x = threadIdx.x; x < g.len; x += 256) d[x] = src[x];
}
int k2a_shim_launch_gather(const K2aGather *tab, int n, uint8_t *dst, void *stream)
{
	if (n <= 0) return 0;
	hipLaunchKernelGGL(k2a_gather_kernel, dim3((unsigned)n), dim3(256), 0, (hipStream_t)stream, tab, dst);
	CHECK(hipGetLastError());
	return 0;
}

__global__ void __launch_bounds__(256)
k2a_wire4_expand_kernel(const uint32_t *__restrict__ src, uint2 *__restrict__ dst, size_t n8)
{
	for (size_t x = (size_t)blockIdx.x * 256 + threadIdx.x; x < n8; x += (size_t)gridDim.x * 256) {
		uint32_t lo, hi;
		k2a_wire4_expand(src[x], lo, hi);
		dst[x] = make_uint2(lo, hi);
	}
}
/* the 2-bit format: one workgroup per group of `ppb` pairs (k2a_wire2_task) */
__global__ void __launch_bounds__(256)
k2a_wire2_expand_kernel(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst, uint32_t total, uint32_t stride, uint32_t ppb)
{
	const uint32_t b0 = blockIdx.x * ppb * stride, b1 = min(b0 + ppb * stride, total);
	if (b0 < b1) k2a_wire2_task(src, dst, b0, b1, stride, (int)threadIdx.x, 256);
}
/* the whole arena out of its upload (a streamed launch that was abandoned: the wavefront-tasks that never started have not expanded
 * their pairs); fmt / stride: K2aQueueDesc.unp_fmt */
int k2a_shim_launch_wire_expand(const uint8_t *src, uint8_t *dst, size_t bytes, int fmt, uint32_t stride, void *stream)
{
	if (fmt == 2) {
		const uint32_t ppb = 16, npairs = (uint32_t)(bytes / stride);
		if (npairs == 0) return 0;
		/* (the escapes of a pair are written by the workgroup that expanded it, behind its own stores: k2a_wire2_task's fence) */
		hipLaunchKernelGGL(k2a_wire2_expand_kernel, dim3((npairs + ppb - 1) / ppb), dim3(256), 0, (hipStream_t)stream, src, dst, (uint32_t)bytes, stride, ppb);
		CHECK(hipGetLastError());
		return 0;
	}
	const size_t n8 = bytes >> 3;
	if (n8 == 0) return 0;
	hipLaunchKernelGGL(k2a_wire4_expand_kernel, dim3((unsigned)((n8 + 255) / 256 < 65536 ? (n8 + 255) / 256 : 65536)), dim3(256), 0, (hipStream_t)stream,
	                   (const uint32_t*)src, (uint2*)dst, n8);
	CHECK(hipGetLastError());
	return 0;
}

int k2a_shim_launch_uniform_layout(const K2aUniform *u, K2aPair *pairs, uint32_t *order2, uint32_t *need, void *stream)
{
	if (!u || u->n == 0) return 0;
	hipLaunchKernelGGL(k2a_uniform_layout_kernel, dim3((u->n + 255) / 256), dim3(256), 0, (hipStream_t)stream, *u, pairs, order2, need);
	CHECK(hipGetLastError());
	return 0;
}

int k2a_shim_launch_trace_pk(int cfg, int dual, const K2aPair *pairs, const uint32_t *order2, int ntasks, const uint8_t *tb,
                             K2aResult *res, uint32_t *cig, void *stream)
{
	if (ntasks <= 0) return 0;
	if (cfg < 0 || cfg >= K2A_NPKCFG) { snprintf(g_err, sizeof(g_err), "bad packed kernel class"); return -1; }
	const int ppw = k2a_trace_ppw(2 * ntasks);
	hipLaunchKernelGGL(g_trace_pk[dual ? 1 : 0][cfg], dim3((2 * ntasks + ppw - 1) / ppw), dim3(64), 0, (hipStream_t)stream,
	                   pairs, order2, ntasks, tb, res, cig, ppw);
	CHECK(hipGetLastError());
	return 0;
}

typedef void (*fill_pkmp_fn)(const K2aScoring, const K2aPair*, const uint32_t*, int, const uint8_t*, uint8_t*, uint32_t*, K2aResult*);
static const fill_pkmp_fn g_fill_pkmp[2][3] = { { k2a_fill_pkmp_kernel<false, 0>, k2a_fill_pkmp_kernel<false, 1>, k2a_fill_pkmp_kernel<false, 2> },
                                                { k2a_fill_pkmp_kernel<true, 0>,  k2a_fill_pkmp_kernel<true, 1>,  k2a_fill_pkmp_kernel<true, 2> } };

int k2a_shim_launch_fill_pkmp(int dual, int mode, const K2aScoring *sc, const K2aPair *pairs, const uint32_t *order2, int ntasks,
                              const uint8_t *seq, uint8_t *tb, uint32_t *bnd, K2aResult *res, void *stream)
{
	if (ntasks <= 0) return 0;
	hipLaunchKernelGGL(g_fill_pkmp[dual ? 1 : 0][mode], dim3(ntasks), dim3(64 * K2A_PKMP_WAVES), 0, (hipStream_t)stream,
	                   *sc, pairs, order2, ntasks, seq, tb, bnd, res);
	CHECK(hipGetLastError());
	return 0;
}

int k2a_shim_launch_fill_solo(int dual, int mode, const K2aScoring *sc, const K2aPair *pairs, const uint32_t *order, int ntasks,
                              const uint8_t *seq, uint8_t *tb, K2aResult *res, void *stream)
{
	if (ntasks <= 0) return 0;
	if (mode < 0 || mode > 2) { snprintf(g_err, sizeof(g_err), "bad kernel class"); return -1; }
	hipLaunchKernelGGL(g_fill_solo[dual ? 1 : 0][mode], dim3((ntasks + K2A_WPB - 1) / K2A_WPB), dim3(64 * K2A_WPB), 0, (hipStream_t)stream,
	                   *sc, pairs, order, ntasks, seq, tb, res);
	CHECK(hipGetLastError());
	return 0;
}

int k2a_shim_launch_trace_solo(const K2aPair *pairs, const uint32_t *order, int ntasks, const uint8_t *tb, K2aResult *res, uint32_t *cig, void *stream)
{
	if (ntasks <= 0) return 0;
	const int ppw = k2a_trace_ppw(ntasks);
	hipLaunchKernelGGL(k2a_trace_solo_kernel<K2A_SOLO_C>, dim3((ntasks + ppw - 1) / ppw), dim3(64), 0, (hipStream_t)stream,
	                   pairs, order, ntasks, tb, res, cig, ppw);
	CHECK(hipGetLastError());
	return 0;
}

int k2a_shim_launch_exts(int mode, int win, const K2aSplice *sp, const K2aPair *pairs, const uint32_t *order, int ntasks, const uint8_t *seq,
                         uint8_t *tb, int32_t *scratch, K2aResult *res, void *stream)
{
	if (ntasks <= 0) return 0;
	const dim3 grid((ntasks + K2A_WPB - 1) / K2A_WPB), block(64 * K2A_WPB);
	typedef void (*exts_fn)(const K2aSplice, const K2aPair*, const uint32_t*, int, const uint8_t*, uint8_t*, K2aResult*);
	static const exts_fn fn[3][2] = {
		{ k2a_exts_kernel<0, K2A_DM_SLOTS_S>, k2a_exts_kernel<0, K2A_DM_SLOTS> },
		{ k2a_exts_kernel<1, K2A_DM_SLOTS_S>, k2a_exts_kernel<1, K2A_DM_SLOTS> },
		{ k2a_exts_kernel<2, K2A_DM_SLOTS_S>, k2a_exts_kernel<2, K2A_DM_SLOTS> } };
	if (mode < 0 || mode > 2 || win < 0 || win > 2) { snprintf(g_err, sizeof(g_err), "bad splice kernel class"); return -1; }
	if (win == 2) {
		if (mode == 0) hipLaunchKernelGGL(k2a_exts_big_kernel<0>, grid, block, 0, (hipStream_t)stream, *sp, pairs, order, ntasks, seq, tb, scratch, res);
		else if (mode == 1) hipLaunchKernelGGL(k2a_exts_big_kernel<1>, grid, block, 0, (hipStream_t)stream, *sp, pairs, order, ntasks, seq, tb, scratch, res);
		else hipLaunchKernelGGL(k2a_exts_big_kernel<2>, grid, block, 0, (hipStream_t)stream, *sp, pairs, order, ntasks, seq, tb, scratch, res);
	} else hipLaunchKernelGGL(fn[mode][win], grid, block, 0, (hipStream_t)stream, *sp, pairs, order, ntasks, seq, tb, res);
	CHECK(hipGetLastError());
	return 0;
}

int k2a_shim_launch_splice_const(const K2aPair *pairs, int n, uint8_t *seq, int noncan, int junc_bonus, void *stream)
{
	if (n <= 0) return 0;
	hipLaunchKernelGGL(k2a_splice_const_kernel, dim3(n), dim3(256), 0, (hipStream_t)stream, pairs, n, seq, noncan, junc_bonus);
	CHECK(hipGetLastError());
	return 0;
}

int k2a_shim_launch_exts_trace(const K2aSplice *sp, const K2aPair *pairs, const uint32_t *order, int ntasks, const uint8_t *tb,
                               K2aResult *res, uint32_t *cig, void *stream)
{
	if (ntasks <= 0) return 0;
	const int ppw = k2a_trace_ppw(ntasks);
	hipLaunchKernelGGL(k2a_exts_trace_kernel, dim3((ntasks + ppw - 1) / ppw), dim3(64), 0, (hipStream_t)stream,
	                   *sp, pairs, order, ntasks, tb, res, cig, ppw);
	CHECK(hipGetLastError());
	return 0;
}

typedef void (*ssec_fn)(const K2aSsec, const K2aPair*, const uint32_t*, int, const uint8_t*, uint8_t*, uint8_t*, K2aResult*);
static const ssec_fn g_ssec[2][2][3] = {      /* [state in LDS][dual][mode] */
	{ { k2a_ssec_kernel<false, 0, false>, k2a_ssec_kernel<false, 1, false>, k2a_ssec_kernel<false, 2, false> },
	  { k2a_ssec_kernel<true, 0, false>,  k2a_ssec_kernel<true, 1, false>,  k2a_ssec_kernel<true, 2, false> } },
	{ { k2a_ssec_kernel<false, 0, true>,  k2a_ssec_kernel<false, 1, true>,  k2a_ssec_kernel<false, 2, true> },
	  { k2a_ssec_kernel<true, 0, true>,   k2a_ssec_kernel<true, 1, true>,   k2a_ssec_kernel<true, 2, true> } } };

int k2a_shim_launch_ssec(int dual, int mode, size_t lds_bytes, const K2aSsec *par, const K2aPair *pairs, const uint32_t *order, int ntasks, const uint8_t *seq,
                         uint8_t *tb, uint8_t *scratch, K2aResult *res, void *stream)
{
	if (ntasks <= 0) return 0;
	if (lds_bytes > 0) {
		const ssec_fn f = g_ssec[1][dual ? 1 : 0][mode];
		if (lds_bytes > 48 * 1024) CHECK(hipFuncSetAttribute((const void*)f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
		hipLaunchKernelGGL(f, dim3(ntasks), dim3(64), lds_bytes, (hipStream_t)stream, *par, pairs, order, ntasks, seq, tb, scratch, res);
	} else
	hipLaunchKernelGGL(g_ssec[0][dual ? 1 : 0][mode], dim3((ntasks + K2A_WPB - 1) / K2A_WPB), dim3(64 * K2A_WPB), 0, (hipStream_t)stream,
	                   *par, pairs, order, ntasks, seq, tb, scratch, res);
	CHECK(hipGetLastError());
	return 0;
}

/* tasks with the state in registers (k2a_ssec_blk_kernel); mode != SCORE: direction bytes into tb at pairs[i].tb_off, as k2a_ssec_kernel lays them out */
int k2a_shim_launch_ssec_blk(int dual, int mode, const K2aSsec *par, const K2aPair *pairs, const uint32_t *order, int ntasks, const uint8_t *seq, uint8_t *tb, K2aResult *res, void *stream)
{
	typedef void (*blk_fn)(const K2aSsec, const K2aPair*, const uint32_t*, int, const uint8_t*, uint8_t*, K2aResult*);
	static const blk_fn fn[2][3] = { { k2a_ssec_blk_kernel<false, 0>, k2a_ssec_blk_kernel<false, 1>, k2a_ssec_blk_kernel<false, 2> },
	                                 { k2a_ssec_blk_kernel<true, 0>, k2a_ssec_blk_kernel<true, 1>, k2a_ssec_blk_kernel<true, 2> } };
	if (ntasks <= 0) return 0;
	if (mode < 0 || mode > 2) { snprintf(g_err, sizeof(g_err), "bad kernel class"); return -1; }
	hipLaunchKernelGGL(fn[dual ? 1 : 0][mode], dim3((ntasks + K2A_WPB - 1) / K2A_WPB), dim3(64 * K2A_WPB), 0, (hipStream_t)stream, *par, pairs, order, ntasks, seq, tb, res);
	CHECK(hipGetLastError());
	return 0;
}

int k2a_shim_launch_ssec_trace(const K2aPair *pairs, const uint32_t *order, int ntasks, const uint8_t *tb, K2aResult *res, uint32_t *cig, void *stream)
{
	if (ntasks <= 0) return 0;
	const int ppw = k2a_trace_ppw(ntasks);
	hipLaunchKernelGGL(k2a_ssec_trace_kernel, dim3((ntasks + ppw - 1) / ppw), dim3(64), 0, (hipStream_t)stream, pairs, order, ntasks, tb, res, cig, ppw);
	CHECK(hipGetLastError());
	return 0;
}

/* cls 0..2: state in LDS (targets up to 1024 / 4096 / 21504 residues), 3: state in `scratch` (3 x padded length bytes at
 * pairs[i].tb_off), 4 / 5: state in registers (bands up to K2A_EXTF_WIN_SPAN(4 / 8) positions), 6: one extension per lane,
 * 7 / 8 / 9: state in registers, four / two / one extensions per wavefront (bands up to K2A_EXTFB_SPAN(16 / 32 / 64) positions) */
int k2a_shim_launch_extf(int cls, const K2aExtf *par, const K2aPair *pairs, const uint32_t *order, int ntasks, const uint8_t *seq,
                         uint8_t *scratch, K2aResult *res, void *stream)
{
	static const int lds_bytes[4] = { 3 * 1024, 3 * 4096, 3 * 21504, 0 };
	if (ntasks <= 0) return 0;
	if (cls < 0 || cls > 9) { snprintf(g_err, sizeof(g_err), "bad kernel class"); return -1; }
	if (cls >= 7) {                                   /* state in registers: four / two / one extensions per wavefront */
		const int ng = cls == 7 ? 4 : cls == 8 ? 2 : 1, waves = (ntasks + ng - 1) / ng;
		const dim3 grid((waves + K2A_WPB - 1) / K2A_WPB), block(64 * K2A_WPB);
		if (cls == 7) hipLaunchKernelGGL(k2a_extf_grp_kernel<16>, grid, block, 0, (hipStream_t)stream, *par, pairs, order, ntasks, seq, res);
		else if (cls == 8) hipLaunchKernelGGL(k2a_extf_grp_kernel<32>, grid, block, 0, (hipStream_t)stream, *par, pairs, order, ntasks, seq, res);
		else hipLaunchKernelGGL(k2a_extf_grp_kernel<64>, grid, block, 0, (hipStream_t)stream, *par, pairs, order, ntasks, seq, res);
	} else if (cls == 6) {                                   /* one extension per lane: 64 tasks per wavefront */
		const int waves = (ntasks + 63) / 64;
		if (par->ring > 0) {
			static int raised;
			const size_t lds = (size_t)3 * par->ring * 256;
			if (!raised && lds > 48 * 1024) { CHECK(hipFuncSetAttribute((const void*)k2a_extf_lane_ring_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024)); raised = 1; }
			hipLaunchKernelGGL(k2a_extf_lane_ring_kernel, dim3(waves), dim3(64), lds, (hipStream_t)stream, *par, pairs, order, ntasks, seq, res);
		} else
		hipLaunchKernelGGL(k2a_extf_lane_kernel, dim3((waves + K2A_WPB - 1) / K2A_WPB), dim3(64 * K2A_WPB), 0, (hipStream_t)stream, *par, pairs, order, ntasks, seq, scratch, res);
	} else if (cls >= 4) {
		const int blocks = (ntasks + K2A_WPB - 1) / K2A_WPB;
		if (cls == 4) hipLaunchKernelGGL(k2a_extf_win_kernel<4>, dim3(blocks), dim3(64 * K2A_WPB), 0, (hipStream_t)stream, *par, pairs, order, ntasks, seq, res);
		else hipLaunchKernelGGL(k2a_extf_win_kernel<8>, dim3(blocks), dim3(64 * K2A_WPB), 0, (hipStream_t)stream, *par, pairs, order, ntasks, seq, res);
	} else if (cls == 3)
		hipLaunchKernelGGL(k2a_extf_kernel<true>, dim3(ntasks), dim3(64), 0, (hipStream_t)stream, *par, pairs, order, ntasks, seq, scratch, res);
	else
		hipLaunchKernelGGL(k2a_extf_kernel<false>, dim3(ntasks), dim3(64), lds_bytes[cls], (hipStream_t)stream, *par, pairs, order, ntasks, seq, scratch, res);
	CHECK(hipGetLastError());
	return 0;
}

int k2a_shim_launch_compact(const K2aPair *pairs, const K2aResult *res, const uint32_t *pos, int n, const uint32_t *cig,
                            uint32_t *pool, void *stream)
{
	if (n <= 0) return 0;
	hipLaunchKernelGGL(k2a_compact_kernel, dim3(n), dim3(64), 0, (hipStream_t)stream, pairs, res, pos, n, cig, pool);
	CHECK(hipGetLastError());
	return 0;
}

} /* extern "C" */
