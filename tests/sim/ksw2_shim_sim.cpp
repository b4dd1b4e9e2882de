/*
 * ksw2_shim_sim.cpp -- TEST INFRASTRUCTURE: a host-memory implementation of ksw2_shim.h that runs the
 * very same per-lane code (ksw2_amd/csrc/ksw2_lane.h) for 64 "lanes" in lock step, mirroring the control
 * flow of k2a_fill_kernel in ksw2_shim_hip.hip line by line.  Linked with ksw2_host_*.c into
 * tests/sim/libksw2_amd_sim.so so the packing / geometry / scheduling / bookkeeping logic can be checked
 * against the oracle in the CPU test tier.  It is never part of the product library libksw2_amd.so.
 */
#include <assert.h>
#include <chrono>
#include <vector>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include "../../ksw2_amd/csrc/ksw2_shim.h"
#include "../../ksw2_amd/csrc/ksw2_lane.h"
#include "../../ksw2_amd/csrc/ksw2_lane_dm.h"
#include "../../ksw2_amd/csrc/ksw2_lane_solo.h"
#include "../../ksw2_amd/csrc/ksw2_lane_extf.h"
#include "../../ksw2_amd/csrc/ksw2_lane_pk.h"
#include "../../ksw2_amd/csrc/ksw2_lane_pkmp.h"
#include "../../ksw2_amd/csrc/ksw2_lane_ssec.h"
#include "../../ksw2_amd/csrc/ksw2_lane_ssecb.h"
#include "../../ksw2_amd/csrc/ksw2_lane_extfb.h"

static thread_local char g_err[256] = "";

static void make_tabs(const K2aScoring &sc, uint32_t *tabs)
{
	for (int x = 0; x < 5; ++x) { tabs[x] = sc.prof[x]; tabs[8 + x] = (uint32_t)sc.colw[x]; }
}

/* k2a_wire2_task (ksw2_shim_hip.hip): arena bytes [b0, b1) out of the 2-bit upload, then the pairs' escape entries */
static void sim_wire2_task(const uint8_t *src8, uint8_t *dst8, uint32_t b0, uint32_t b1, uint32_t stride)
{
	for (uint32_t x = 0; x < (b1 - b0) >> 4; ++x) {
		uint32_t w, o[4];
		memcpy(&w, src8 + (b0 >> 2) + 4 * (size_t)x, 4);
		k2a_wire2_expand(w, o);
		memcpy(dst8 + b0 + 16 * (size_t)x, o, 16);
	}
	for (uint32_t pair = 0; pair < (b1 - b0) / stride; ++pair)
		for (uint32_t e = 0; e < K2A_WIRE2_ESC; ++e) {
			const uint32_t base = b0 + pair * stride;
			uint32_t ent;
			memcpy(&ent, src8 + ((base + stride) >> 2) - K2A_WIRE2_SLOT + 4 * e, 4);
			if (ent) memset(dst8 + base + (ent & 0xfffffu), (int)(ent >> 28), (ent >> 20) & 0xffu);
		}
}

/* what k2a_scan_codes reports as "a code above 4" (the wavefront-task's look at its targets on the device) */
static bool sim_codes_above4(const uint8_t *t, int n)
{
	for (int i = 0; i < n; ++i) if (t[i] > 4) return true;
	return false;
}

template<int G, int C, bool DUAL, int MODE>
static void sim_fill(const K2aScoring sc, const K2aPair *pairs, const uint32_t *order, int ntasks, const uint8_t *seq, uint8_t *tb,
                     K2aResult *res)
{
	constexpr int NG = 64 / G;
	typedef K2aLane<G, C, DUAL, MODE> Lane;
	const int nwaves = (ntasks + NG - 1) / NG;
	uint32_t tabs[16] = {0};
	make_tabs(sc, tabs);
	for (int wv = 0; wv < nwaves; ++wv) {
		static thread_local Lane L[64];
		K2aBook book[NG];
		int rowbuf[NG][3 * C];
		K2aPair pr[64];
		uint32_t pi[64];
		bool valid[64], gdone[64];
		int klast[64], kmax = -1, ktop = -1;
		bool zany = false;
		uint8_t *tbp[64];
		for (int lane = 0; lane < 64; ++lane) {
			const int grp = lane / G, gl = lane % G, task = wv * NG + grp;
			valid[lane] = task < ntasks;
			pi[lane] = order[valid[lane] ? task : 0];
			pr[lane] = pairs[pi[lane]];
			L[lane].setup(pr[lane], seq, gl, valid[lane]);
			if (gl == 0) k2a_book_reset(&book[grp]);
			klast[lane] = L[lane].last_step();
			if (klast[lane] > kmax) kmax = klast[lane];
			tbp[lane] = tb + pr[lane].tb_off;
			gdone[lane] = !valid[lane];
			L[lane].qb = L[lane].next_query_code(-1);
			if (valid[lane]) {
				ktop = k2a_max(ktop, k2a_min(pr[lane].qlen - 1, k2a_min(C - 1, pr[lane].tlen - 1) + pr[lane].w));
				zany |= pr[lane].zdrop >= 0;
			}
		}
		for (int k = 0; k <= kmax; ++k) {
			int hin[64], ein[64], e2in[64], qnext[64];
			for (int lane = 0; lane < 64; ++lane) {          /* DPP rotate inside each group */
				const int grp = lane / G, gl = lane % G, src = grp * G + (gl + G - 1) % G;
				hin[lane] = L[src].hout; ein[lane] = L[src].eout; e2in[lane] = DUAL ? L[src].e2out : 0;
			}
			bool anyfin = false, wild = sc.m > 5;
			bool nfin[64];
			for (int lane = 0; lane < 64; ++lane) {      /* init events first: the wave-uniform wildcard test follows them */
				if (L[lane].need_init(k)) L[lane].template do_init<true>(sc, tabs);
				L[lane].hu_prev = hin[lane];
				wild |= L[lane].qb >= 4;
			}
			for (int lane = 0; lane < 64; ++lane) {
				qnext[lane] = L[lane].next_query_code(k);
				if (k <= ktop) L[lane].top_inputs(sc, k, hin[lane], ein[lane], e2in[lane]);
				uint32_t tw[Lane::TBWORDS];
				const bool live = L[lane].step(sc, tabs + 8, sc.mat, wild, k, hin[lane], ein[lane], e2in[lane], tw);
				if (MODE != K2A_MODE_SCORE && live)
					memcpy(tbp[lane] + k2a_tb_word((size_t)k, lane % G, (size_t)(klast[lane] + 1), G, Lane::TBWORDS * 4), tw, sizeof(tw));
				nfin[lane] = L[lane].need_fin(k);
				anyfin |= nfin[lane];
			}
			if (anyfin) {
				for (int lane = 0; lane < 64; ++lane)
					if (nfin[lane]) L[lane].do_fin(sc, &book[lane / G], pr[lane].zdrop, rowbuf[lane / G]);
				for (int lane = 0; lane < 64; ++lane)
					if (book[lane / G].dropped) gdone[lane] = true;
			}
			bool all_done = true;
			for (int lane = 0; lane < 64; ++lane) {
				L[lane].qb = qnext[lane];
				if (!(gdone[lane] || k >= klast[lane])) all_done = false;
			}
			if (zany && all_done) break;
		}
		for (int lane = 0; lane < 64; ++lane)
			if (valid[lane] && lane % G == 0) k2a_finish(pr[lane], book[lane / G], &res[pi[lane]]);
	}
}

/* mirrors k2a_fill_pk_kernel */
template<int G, int C, bool DUAL, int MODE, bool RB, bool NOMAX, int LDSROW = 0, bool DEFER = false>
static void sim_fill_pk(const K2aScoring sc, const K2aPair *pairs, const uint32_t *order2, int ntasks, const uint8_t *seq, uint8_t *tb,
                        K2aResult *res, K2aQueueDesc *qd)
{
	constexpr int NG = 64 / G;
	typedef K2aLanePk<G, C, DUAL, MODE, RB, NOMAX, LDSROW, DEFER, true> Lane;      /* the TN build everywhere: it is the plain one plus the wildcard rows (the device picks per wavefront-task, k2a_scan_codes) */
	const int nwaves = (ntasks + NG - 1) / NG;
	for (int wv = 0; wv < nwaves; ++wv) {
		/* streamed launches (k2a_queue_wait): wavefront-tasks in grid order; the simulator's uploads are synchronous, so a piece
		 * that has not landed by now never will -- what the kernel finds out by its timeout */
		if (qd) {
			if (qd->nwt != (uint32_t)nwaves) { qd->abort = 2; break; }
			qd->next = (uint32_t)wv + 1;
			if (qd->abort) break;
			if (qd->need[wv] > qd->wm[0]) { qd->abort = 1; break; }
			if (qd->unp_bytes) {                                 /* 4-bit wire format: the wavefront-task expands its own pairs (k2a_queue_wait) */
				const uint32_t b0 = (uint32_t)wv * qd->unp_bytes, b1 = std::min(b0 + qd->unp_bytes, qd->unp_total);
				if ((qd->unp_fmt >> 30) == 2u) sim_wire2_task(qd->unp_src, qd->unp_dst, b0, b1, qd->unp_fmt & 0x3fffffffu);
				else for (uint32_t x = 0; x < (b1 - b0) >> 3; ++x) {
					uint32_t w4, lo, hi;
					memcpy(&w4, qd->unp_src + (b0 >> 1) + 4 * (size_t)x, 4);
					k2a_wire4_expand(w4, lo, hi);
					memcpy(qd->unp_dst + b0 + 8 * (size_t)x, &lo, 4); memcpy(qd->unp_dst + b0 + 8 * (size_t)x + 4, &hi, 4);
				}
			}
		}
		static thread_local Lane L[64];
		static thread_local uint32_t lrows[K2A_PK_LDSROW_WORDS(C)];
		K2aBook book[NG][2];
		K2aPair prA[64];
		uint32_t piA[64], piB[64], stage[(NG * K2A_PK_STAGE(C) > 64 * 5) ? NG * K2A_PK_STAGE(C) : 64 * 5];
		int zdA[64], zdB[64], klast[64], kmax = -1, ktop = -1;
		bool valid[64], gdone[64], zseq = RB || NOMAX;
		for (int lane = 0; lane < 64; ++lane) {
			const int grp = lane / G, gl = lane % G, task = wv * NG + grp;
			valid[lane] = task < ntasks;
			piA[lane] = order2[valid[lane] ? 2 * task : 0]; piB[lane] = order2[valid[lane] ? 2 * task + 1 : 0];
			prA[lane] = pairs[piA[lane]];
			zdA[lane] = prA[lane].zdrop; zdB[lane] = pairs[piB[lane]].zdrop;
			L[lane].lrow = lrows + lane;
			L[lane].setup(prA[lane], pairs[piB[lane]], seq, gl, valid[lane], sc.cp);
			if (gl == 0) { k2a_book_reset(&book[grp][0]); k2a_book_reset(&book[grp][1]); }
			klast[lane] = L[lane].last_step();
			if (klast[lane] > kmax) kmax = klast[lane];
			gdone[lane] = !valid[lane];
			L[lane].load_query_group(0, L[lane].knext == 0 ? L[lane].koff_next : L[lane].koff, L[lane].qwA, L[lane].qwB);
			zseq |= valid[lane] && (zdA[lane] >= 0 || zdB[lane] >= 0);
			if (valid[lane]) {
				const int kt = k2a_min(prA[lane].qlen - 1, k2a_min(C - 1, prA[lane].tlen - 1) + prA[lane].w);
				if (kt > ktop) ktop = kt;
			}
		}
		for (int k = 0; k <= kmax; ++k) {
			k2a_pk hin[64], ein[64], e2in[64];
			static thread_local uint32_t qpa[64], qpb[64];
			for (int lane = 0; lane < 64; ++lane) {
				const int grp = lane / G, gl = lane % G, src = grp * G + (gl + G - 1) % G;
				hin[lane] = L[src].hout; ein[lane] = L[src].eout; e2in[lane] = DUAL ? L[src].e2out : 0;
			}
			bool anyfin = false, nfin[64];
			int bsA[64], bsB[64];
			for (int lane = 0; lane < 64; ++lane) {
				const int src = (lane / G) * G + (lane % G + G - 1) % G;
				bsA[lane] = L[src].baseA; bsB[lane] = L[src].baseB;
			}
			for (int lane = 0; lane < 64; ++lane) {
				if (L[lane].need_init(k)) {
					L[lane].do_init(sc, bsA[lane], bsB[lane]);
					if (k & 3) L[lane].reload_query_group(k);
					if (DEFER && valid[0]) {                    /* the strip's checkpoint header (mirrors k2a_fill_pk_kernel) */
						K2aCkHead h; h.baseA = L[lane].baseA; h.baseB = L[lane].baseB; h.hd0 = L[lane].hd0; h.pad = 0;
						((K2aCkHead*)(tb + prA[0].tb_off + (size_t)prA[0].bnd_off * K2A_CK_STEP_BYTES))[(size_t)(lane / G) * prA[0].cig_off + L[lane].S] = h;
					}
				}
				L[lane].hu_prev = hin[lane];
				if (RB) {
					hin[lane] = k2a_pk_add(hin[lane], L[lane].delta); ein[lane] = k2a_pk_add(ein[lane], L[lane].delta);
					if (DUAL) e2in[lane] = k2a_pk_add(e2in[lane], L[lane].delta);
				}
				if ((k & 3) == 0) L[lane].load_query_group(k + 4, L[lane].knext <= k + 4 ? L[lane].koff_next : L[lane].koff, qpa[lane], qpb[lane]);
				L[lane].set_qb(Lane::query_pick(L[lane].qwA, L[lane].qwB, k & 3));
				if (k <= ktop) L[lane].top_inputs(sc, k, hin[lane], ein[lane], e2in[lane]);
			}
			{	/* K2A_SYNC_WN: "some lane of the wavefront holds a target wildcard row" (the kernel refreshes it where strips start and end) */
				bool wn = false;
				for (int lane = 0; lane < 64; ++lane) wn |= L[lane].hasn != 0;
				for (int lane = 0; lane < 64; ++lane) L[lane].wn = wn;
			}
			for (int lane = 0; lane < 64; ++lane) {
				uint32_t tw[Lane::TBWORDS];
				if (DEFER && valid[0]) { uint32_t *ck = (uint32_t*)(tb + prA[0].tb_off) + 2 * ((size_t)k * 64 + lane); ck[0] = hin[lane]; ck[1] = ein[lane]; }
				const bool live = L[lane].step(sc, k, hin[lane], ein[lane], e2in[lane], tw);
				if (MODE != K2A_MODE_SCORE && live)
					memcpy(tb + prA[lane].tb_off + k2a_tb_word((size_t)k, lane % G, (size_t)(klast[lane] + 1), G, Lane::TBWORDS * 4), tw, sizeof(tw));
				nfin[lane] = L[lane].need_fin(k);
				anyfin |= nfin[lane];
			}
			if (anyfin) {
				bool gfin[NG];
				for (int g = 0; g < NG; ++g) gfin[g] = false;
				for (int lane = 0; lane < 64; ++lane) {
					if (!nfin[lane]) continue;
					uint32_t *rowbuf = &stage[(lane / G) * K2A_PK_STAGE(C)];
					assert(!gfin[lane / G]);                          /* one strip per group and step */
					gfin[lane / G] = true;
					if (NOMAX) L[lane].fin_score_only(sc, &book[lane / G][0], &book[lane / G][1]);
					else if (zseq) {
						if (!L[lane].fin_fast(sc, &book[lane / G][0], &book[lane / G][1], zdA[lane], zdB[lane])) {
							L[lane].stage_rows(rowbuf);
							L[lane].do_fin_seq(sc, &book[lane / G][0], &book[lane / G][1], zdA[lane], zdB[lane], rowbuf);
						}
					} else L[lane].stage_rows(rowbuf);
				}
				if (!zseq)
					for (int lane = 0; lane < 64; ++lane) {
						if (gfin[lane / G]) L[lane].fin_local_rows(sc, &stage[(lane / G) * K2A_PK_STAGE(C)]);
						if (nfin[lane]) L[lane].end_strip();
					}
				if (zseq)
					for (int lane = 0; lane < 64; ++lane)
						if (!DEFER && book[lane / G][0].dropped && book[lane / G][1].dropped) gdone[lane] = true;
			}
			bool all_done = true;
			for (int lane = 0; lane < 64; ++lane) {
				if ((k & 3) == 3) { L[lane].qwA = qpa[lane]; L[lane].qwB = qpb[lane]; }
				if (!(gdone[lane] || k >= klast[lane])) all_done = false;
			}
			if (zseq && all_done) break;
		}
		if (!zseq) {
			for (int lane = 0; lane < 64; ++lane) {
				stage[lane * 5 + 0] = L[lane].lmax; stage[lane * 5 + 1] = L[lane].lmax_t; stage[lane * 5 + 2] = L[lane].lmax_q;
				stage[lane * 5 + 3] = L[lane].lmqe; stage[lane * 5 + 4] = L[lane].lmqe_t;
			}
			for (int lane = 0; lane < 64; ++lane)
				if (valid[lane] && lane % G == 0) {
					k2a_merge_local(stage + (lane / G) * G * 5, G, 0, &book[lane / G][0]);
					k2a_merge_local(stage + (lane / G) * G * 5, G, 1, &book[lane / G][1]);
					book[lane / G][0].rows = book[lane / G][1].rows = prA[lane].tlen;
				}
			for (int lane = 0; lane < 64; ++lane) {
				const K2aPair &pr = prA[lane];
				if (valid[lane] && pr.tlen == pr.tlen_full && lane % G == ((pr.tlen_full - 1) % C) % G) {
					K2aBook *a = &book[lane / G][0], *b = &book[lane / G][1];
					a->mte = k2a_pk_lo(L[lane].last_m); a->mte_q = k2a_pk_lo(L[lane].last_j);
					b->mte = k2a_pk_hi(L[lane].last_m); b->mte_q = k2a_pk_hi(L[lane].last_j);
					if (pr.tlen_full - 1 + pr.w >= pr.qlen - 1) { a->score = k2a_pk_lo(L[lane].last_h); b->score = k2a_pk_hi(L[lane].last_h); }
				}
			}
		}
		for (int lane = 0; lane < 64; ++lane)
			if (valid[lane] && lane % G == 0) {
				bool gsaw = false;
				if (sc.pk_tn1) gsaw = sim_codes_above4(seq + prA[lane].toff, prA[lane].tlen_full) || sim_codes_above4(seq + pairs[piB[lane]].toff, pairs[piB[lane]].tlen_full);      /* k2a_scan_codes */
				else for (int l = lane; l < lane + G; ++l) gsaw |= L[l].saw_wildcard();
				k2a_finish(prA[lane], book[lane / G][0], &res[piA[lane]]);
				if (piB[lane] != piA[lane]) k2a_finish(pairs[piB[lane]], book[lane / G][1], &res[piB[lane]]);
				if (gsaw) { res[piA[lane]].pad[0] = 1; res[piB[lane]].pad[0] = 1; }
			}
	}
}

/* mirrors k2a_argmax_kernel: one job per (task, which), every job an independent lane; frozen books go into the class's list */
template<int G, int C, bool RB>
static void sim_argmax(const K2aScoring sc, const K2aPair *pairs, const uint32_t *order2, int ntasks, const uint8_t *seq, uint8_t *ck, K2aResult *res)
{
	constexpr int NG = 64 / G;
	typedef K2aLanePk<G, C, false, K2A_MODE_SCORE, RB, false, 0, false, true> Lane;
	uint32_t *zlist = (uint32_t*)(ck + pairs[order2[0]].tb_off) - K2A_ZLIST_WORDS(ntasks);
	zlist[0] = 0;                                         /* (the device's fill kernel does this) */
	for (int job = 0; job < 3 * ntasks; ++job) {
		const int task = job / 3, which = job % 3;
		const uint32_t piA = order2[2 * task], piB = order2[2 * task + 1];
		const K2aPair prA = pairs[piA], prB = pairs[piB];
		const bool inexA = res[piA].pad[1] != 0, inexB = res[piB].pad[1] != 0;
		if ((which == 0 && inexA) || (which == 1 && inexB && piB != piA)) zlist[1 + zlist[0]++] = (uint32_t)task * 2u + (uint32_t)which;
		int row;
		if (which == 0) row = res[piA].max_t;
		else if (which == 1) row = piB == piA ? -1 : res[piB].max_t;
		else row = (prA.tlen == prA.tlen_full && !(inexA && inexB)) ? prA.tlen_full - 1 : -1;
		if (row < 0) continue;
		const int S = row / C, grp = task % NG;
		static thread_local Lane L;
		L.lrow = 0;
		L.setup(prA, prB, seq, S % G, true, sc.cp);
		L.Snext = S;
		L.schedule_next();
		const int kbeg = L.knext;
		const uint8_t *blk = ck + prA.tb_off;
		L.do_init(sc, 0, 0, (const K2aCkHead*)(blk + (size_t)prA.bnd_off * K2A_CK_STEP_BYTES) + (size_t)grp * prA.cig_off + S);
		L.wn = L.hasn != 0;                                  /* K2A_SYNC_WN: a wavefront-uniform "maybe" on the device, exact here -- the fix is a no-op on rows without a wildcard */
		const uint32_t *st = (const uint32_t*)blk;
		for (int k = kbeg; k <= L.kfin; ++k) {
			const size_t at = 2 * ((size_t)k * 64 + grp * G + S % G);
			const int jc = k2a_min(k2a_max(k - L.koff, 0), L.qlen - 1);
			L.set_qb(k2a_pair16(L.qa[jc], L.qbp[jc]));
			uint32_t tw[Lane::TBWORDS];
			L.step(sc, k, st[at], st[at + 1], 0u, tw);
		}
		const k2a_pk v = L.rmj(row - S * C);
		if (which == 0) res[piA].max_q = (int)(v & 0xffffu);
		else if (which == 1) res[piB].max_q = (int)(v >> 16);
		else {
			if (!inexA) res[piA].mte_q = (int)(v & 0xffffu);
			if (piB != piA && !inexB) res[piB].mte_q = (int)(v >> 16);
		}
	}
}

/* mirrors k2a_zscan_kernel: per frozen book, rounds of 16 strips re-run with the exact lane code and folded into the book in row order */
template<int G, int C, bool RB>
static void sim_zscan(const K2aScoring sc, const K2aPair *pairs, const uint32_t *order2, int ntasks, const uint8_t *seq, const uint8_t *ck, K2aResult *res)
{
	constexpr int NG = 64 / G, ZG = 16;
	typedef K2aLanePk<G, C, false, K2A_MODE_SCORE, RB, false, 0, false, true> Lane;
	const uint32_t *zlist = (const uint32_t*)(ck + pairs[order2[0]].tb_off) - K2A_ZLIST_WORDS(ntasks);
	for (uint32_t gi = 0; gi < zlist[0]; ++gi) {
		const uint32_t ent = zlist[1 + gi];
		const int task = (int)(ent >> 1), half = (int)(ent & 1u);
		const uint32_t piA = order2[2 * task], piB = order2[2 * task + 1], pi = half ? piB : piA;
		const K2aPair prA = pairs[piA], prB = pairs[piB], pr = half ? prB : prA;
		const int grp = task % NG, nstrips = (prA.tlen + C - 1) / C;
		const K2aResult r0 = res[pi];
		K2aBook b;
		b.max = r0.max; b.max_t = r0.max_t; b.max_q = r0.max_q; b.mqe = r0.mqe; b.mqe_t = r0.mqe_t; b.mte = r0.mte; b.mte_q = r0.mte_q;
		b.score = r0.score; b.dropped = 0; b.rows = r0.rows_done; b.inexact = 0;
		const int S1 = k2a_max(r0.rows_done - 1, 0) / C;
		const uint8_t *blk = ck + prA.tb_off;
		for (int round = 0; ; ++round) {
			static thread_local uint32_t stage[ZG][K2A_PK_STAGE(C)];
			for (int zl = 0; zl < ZG; ++zl) {
				const int S = S1 + round * ZG + zl;
				if (S >= nstrips) continue;
				static thread_local Lane L;
				L.lrow = 0;
				L.setup(prA, prB, seq, S % G, true, sc.cp);
				L.Snext = S;
				L.schedule_next();
				const int kbeg = L.knext;
				L.do_init(sc, 0, 0, (const K2aCkHead*)(blk + (size_t)prA.bnd_off * K2A_CK_STEP_BYTES) + (size_t)grp * prA.cig_off + S);
				L.wn = L.hasn != 0;
				const uint32_t *st = (const uint32_t*)blk;
				for (int k = kbeg; k <= L.kfin; ++k) {
					const size_t at = 2 * ((size_t)k * 64 + grp * G + S % G);
					const int jc = k2a_min(k2a_max(k - L.koff, 0), L.qlen - 1);
					L.set_qb(k2a_pair16(L.qa[jc], L.qbp[jc]));
					uint32_t tw[Lane::TBWORDS];
					L.step(sc, k, st[at], st[at + 1], 0u, tw);
				}
				L.stage_rows(stage[zl]);
			}
			for (int l = 0; l < ZG && S1 + round * ZG + l < nstrips && !b.dropped; ++l)
				k2a_fin_rows_half<C>(sc, &b, pr.zdrop, stage[l], half, RB, pr.qlen, pr.tlen, pr.tlen_full, pr.w);
			if (b.dropped || S1 + (round + 1) * ZG >= nstrips) {
				if (getenv("K2A_DBG")) fprintf(stderr, "zscan: frozen at row %d, settled at row %d (%s), %d rows later, %d round(s)\n", r0.rows_done - 1, b.rows - 1, b.dropped ? "drop" : "end", b.rows - r0.rows_done, round + 1);
				k2a_finish(pr, b, &res[pi]); res[pi].pad[0] = r0.pad[0]; break;
			}
		}
	}
}

/* mirrors k2a_fill_mp_kernel */
template<int G, int C, bool DUAL, int MODE, bool LROW = false>
static void sim_fill_mp(const K2aScoring sc, const K2aPair *pairs, const uint32_t *order, int ntasks, const uint8_t *seq, uint8_t *tb,
                        int32_t *bnd, K2aResult *res)
{
	typedef K2aLane<G, C, DUAL, MODE, LROW> Lane;
	for (int task = 0; task < ntasks; ++task) {
		static thread_local Lane L[64];
		static thread_local int lrows[K2A_LROW_WORDS(C)];
		K2aBook book;
		int rowbuf[3 * C];
		const uint32_t pi = order[task];
		const K2aPair pr = pairs[pi];
		uint32_t tabs[16] = {0};
		make_tabs(sc, tabs);
		for (int gl = 0; gl < 64; ++gl) { L[gl].lrow = lrows + gl; L[gl].setup(pr, seq, gl, true); }
		k2a_book_reset(&book);
		int32_t *Bh = bnd + pr.bnd_off, *Be = Bh + pr.qlen, *Be2 = Be + pr.qlen;
		const int R = G * C, ngen = (pr.tlen + R - 1) / R;
		size_t kbase = 0;
		bool dropped = false;
		const int ktop = k2a_min(pr.qlen - 1, k2a_min(C - 1, pr.tlen - 1) + pr.w);
		const size_t tbsteps = k2a_tb_steps<G, C, true>(pr.qlen, pr.tlen, pr.w);
		for (int g = 0; g < ngen && !dropped; ++g) {
			int jlo, nsteps;
			k2a_gen_cols<G, C>(g, pr.qlen, pr.tlen, pr.w, &jlo, &nsteps);
			for (int gl = 0; gl < 64; ++gl) { L[gl].begin_generation(g, jlo); }
			if (g > 0 && jlo > 0) L[0].hu_prev = Bh[jlo - 1];
			for (int gl = 0; gl < 64; ++gl) L[gl].qb = L[gl].next_query_code(-1);
			for (int k = 0; k < nsteps; ++k) {
				int hin[64], ein[64], e2in[64], qnext[64];
				for (int gl = 0; gl < 64; ++gl) {
					const int src = (gl + G - 1) % G;
					hin[gl] = L[src].hout; ein[gl] = L[src].eout; e2in[gl] = DUAL ? L[src].e2out : 0;
				}
				if (g > 0) {
					const int j = jlo + k;
					hin[0] = j < pr.qlen ? Bh[j] : K2A_NEG; ein[0] = j < pr.qlen ? Be[j] : K2A_NEG;
					e2in[0] = (DUAL && j < pr.qlen) ? Be2[j] : K2A_NEG;
				}
				bool nfin[64], anyfin = false, wild = sc.m > 5;
				for (int gl = 0; gl < 64; ++gl) {
					if (L[gl].need_init(k)) L[gl].template do_init<false>(sc, tabs);
					L[gl].hu_prev = hin[gl];
					wild |= L[gl].qb >= 4;
				}
				for (int gl = 0; gl < 64; ++gl) {
					qnext[gl] = L[gl].next_query_code(k);
					if (g == 0 && k <= ktop) L[gl].top_inputs(sc, k, hin[gl], ein[gl], e2in[gl]);
					uint32_t tw[Lane::TBWORDS];
					const int jj = L[gl].column(k);
					const bool mine = L[gl].S >= 0 && jj >= 0 && jj <= L[gl].je;
					const bool live = L[gl].step(sc, tabs + 8, sc.mat, wild, k, hin[gl], ein[gl], e2in[gl], tw);
					if (MODE != K2A_MODE_SCORE && live)
						memcpy(tb + pr.tb_off + k2a_tb_word(kbase + (size_t)k, gl, tbsteps, G, Lane::TBWORDS * 4), tw, sizeof(tw));
					if (gl == G - 1 && mine) { Bh[jj] = L[gl].hout; Be[jj] = L[gl].eout; if (DUAL) Be2[jj] = L[gl].e2out; }
					nfin[gl] = L[gl].need_fin(k);
					anyfin |= nfin[gl];
				}
				if (anyfin) {
					for (int gl = 0; gl < 64; ++gl) if (nfin[gl]) L[gl].do_fin(sc, &book, pr.zdrop, rowbuf);
					if (book.dropped) dropped = true;
				}
				for (int gl = 0; gl < 64; ++gl) L[gl].qb = qnext[gl];
				if (dropped) break;
			}
			kbase += k2a_gen_pad(nsteps);
		}
		k2a_finish(pr, book, &res[pi]);
	}
}

template<int G, int C, bool DUAL, bool MP>
static void sim_trace(const K2aPair *pairs, const uint32_t *order, int ntasks, const uint8_t *tb, K2aResult *res, uint32_t *cig)
{
	K2aUnit16 win[K2A_WALK_SLOT / 16];                 /* the walk's window (the kernels: LDS) */
	for (int t = 0; t < ntasks; ++t) {
		const uint32_t pi = order[t];
		const K2aPair pr = pairs[pi];
		int n = 0;
		if (res[pi].ti >= 0 && res[pi].tj >= 0)
			n = k2a_trace_pair<G, C, DUAL, MP>(tb + pr.tb_off, res[pi].ti, res[pi].tj, cig + pr.cig_off, pr.qlen, pr.tlen, pr.w, (uint8_t*)win);
		res[pi].n_cigar = n;
	}
}

typedef void (*fill_fn)(const K2aScoring, const K2aPair*, const uint32_t*, int, const uint8_t*, uint8_t*, K2aResult*);
typedef void (*trace_fn)(const K2aPair*, const uint32_t*, int, const uint8_t*, K2aResult*, uint32_t*);
#define FILL_ROW(G, C) { { sim_fill<G, C, false, 0>, sim_fill<G, C, false, 1>, sim_fill<G, C, false, 2> }, \
                         { sim_fill<G, C, true, 0>,  sim_fill<G, C, true, 1>,  sim_fill<G, C, true, 2> } }
static const fill_fn g_fill[4][2][3] = { FILL_ROW(16, 8), FILL_ROW(64, 8), FILL_ROW(64, 16), FILL_ROW(64, 32) };
typedef void (*fill_mp_fn)(const K2aScoring, const K2aPair*, const uint32_t*, int, const uint8_t*, uint8_t*, int32_t*, K2aResult*);
static const fill_mp_fn g_fill_mp[2][3] = {
	{ sim_fill_mp<64, 16, false, 0>, sim_fill_mp<64, 16, false, 1>, sim_fill_mp<64, 16, false, 2> },
	{ sim_fill_mp<64, 16, true, 0>,  sim_fill_mp<64, 16, true, 1>,  sim_fill_mp<64, 16, true, 2> } };
static const fill_mp_fn g_fill_mp_lds[2] = { sim_fill_mp<64, 16, false, 1, true>, sim_fill_mp<64, 16, false, 2, true> };
#define TRACE_ROW(G, C, MP) { sim_trace<G, C, false, MP>, sim_trace<G, C, true, MP> }
static const trace_fn g_trace[K2A_NCFG][2] = { TRACE_ROW(16, 8, false), TRACE_ROW(64, 8, false), TRACE_ROW(64, 16, false),
                                               TRACE_ROW(64, 32, false), TRACE_ROW(64, 16, true) };

template<int G, int C, bool DUAL, bool MP = false>
static void sim_trace_pk(const K2aPair *pairs, const uint32_t *order2, int ntasks, const uint8_t *tb, K2aResult *res, uint32_t *cig)
{
	K2aUnit16 win[K2A_WALK_SLOT / 16];
	for (int t = 0; t < 2 * ntasks; ++t) {
		const int half = t & 1;
		const uint32_t piA = order2[t & ~1], pi = order2[t];
		if (half && pi == piA) continue;
		const K2aPair pr = pairs[pi];
		int n = 0;
		if (res[pi].ti >= 0 && res[pi].tj >= 0)
			n = k2a_trace_pair_pk<G, C, DUAL, MP>(tb + pr.tb_off, half, res[pi].ti, res[pi].tj, cig + pr.cig_off, pr.qlen, pr.tlen, pr.w, (uint8_t*)win);
		res[pi].n_cigar = n;
	}
}

typedef void (*fill_pk_fn)(const K2aScoring, const K2aPair*, const uint32_t*, int, const uint8_t*, uint8_t*, K2aResult*, K2aQueueDesc*);
#define PK_ROW(G, C, RB, NM) { { sim_fill_pk<G, C, false, 0, RB, NM>, sim_fill_pk<G, C, false, 1, RB, NM>, sim_fill_pk<G, C, false, 2, RB, NM> }, \
                               { sim_fill_pk<G, C, true, 0, RB, NM>,  sim_fill_pk<G, C, true, 1, RB, NM>,  sim_fill_pk<G, C, true, 2, RB, NM> } }
#define PK_SET(NM) { { PK_ROW(8, 18, false, NM), PK_ROW(16, 8, false, NM), PK_ROW(64, 8, false, NM), PK_ROW(64, 16, false, NM) }, \
                     { PK_ROW(8, 18, true, NM),  PK_ROW(16, 8, true, NM),  PK_ROW(64, 8, true, NM),  PK_ROW(64, 16, true, NM) } }
static const fill_pk_fn g_fill_pk[2][2][K2A_NPKCFG][2][3] = { PK_SET(false), PK_SET(true) };
static const fill_pk_fn g_fill_pk_lds[2][2] = {
	{ sim_fill_pk<64, 16, true, 1, false, false, 1>, sim_fill_pk<64, 16, true, 2, false, false, 1> },
	{ sim_fill_pk<64, 16, true, 1, true, false, 1>,  sim_fill_pk<64, 16, true, 2, true, false, 1> } };
#define LDSCODE_SET(G, C) { { sim_fill_pk<G, C, false, 0, false, false, 2>, sim_fill_pk<G, C, false, 0, true, false, 2> }, \
                            { sim_fill_pk<G, C, false, 0, false, true, 2>,  sim_fill_pk<G, C, false, 0, true, true, 2> } }
static const fill_pk_fn g_fill_pk_ldscodes[3][2][2] = { LDSCODE_SET(64, 16), LDSCODE_SET(8, 18), LDSCODE_SET(16, 8) };      /* [geometry][nomax][rebased] */
/* launch-time forms (ksw2_shim.h): the simulator takes the LDS forms unless the option says 0 (the GPU launcher decides by the
 * number of tasks when the option is -1) */
static int g_opt[K2A_NOPT] = { -1, -1 };
static bool sim_use_ldscodes(void) { return g_opt[K2A_OPT_LDSCODES] != 0; }
static bool sim_use_ldsrows(void) { return g_opt[K2A_OPT_LDSROWS] != 0; }
void k2a_shim_set_option(int opt, int value) { if (opt >= 0 && opt < K2A_NOPT) g_opt[opt] = value < 0 ? -1 : value != 0; }
int k2a_shim_pk_form(int cfg, int dual, int mode, int nomax, int)
{
	if (cfg < 0 || cfg >= K2A_NPKCFG) return 0;
	if (K2A_PK_LDSROWS(k2a_pkcfg_G[cfg], k2a_pkcfg_C[cfg], dual, mode, nomax) && sim_use_ldsrows()) return 1;
	if (K2A_PK_LDSCODES(k2a_pkcfg_G[cfg], k2a_pkcfg_C[cfg], dual, mode, nomax) && sim_use_ldscodes()) return 2;
	return 0;
}
int k2a_shim_mp_form(int dual, int mode, int) { return !dual && mode != K2A_MODE_SCORE && sim_use_ldsrows(); }
#define TRACE_PK_ROW(D) { sim_trace_pk<8, 18, D>, sim_trace_pk<16, 8, D>, sim_trace_pk<64, 8, D>, sim_trace_pk<64, 16, D>, sim_trace_pk<64, 16, D, true> }
static const trace_fn g_trace_pk[2][K2A_NPKCFG] = { TRACE_PK_ROW(false), TRACE_PK_ROW(true) };


/* mirrors k2a_exts_kernel: one alignment per wavefront, diagonal-major, K2A_DM_SLOTS slots of 64 target positions */
template<int MODE, int K>
static void sim_exts(const K2aSplice sp, const K2aPair *pairs, const uint32_t *order, int ntasks, const uint8_t *seq, uint8_t *tb, K2aResult *res)
{
	for (int task = 0; task < ntasks; ++task) {
		const uint32_t pi = order[task];
		const K2aPair pr = pairs[pi];
		const int qlen = pr.qlen, tlen = pr.tlen_full;
		const uint8_t *qry = seq + pr.qoff;
		const uint32_t *cst = (const uint32_t*)seq + pr.bnd_off;
		uint8_t *tbp = tb + pr.tb_off;
		const int ncol = k2a_min(qlen, tlen);
		static thread_local int H1[K][64], H2[K][64], En[K][64], E2n[K][64], Fn[K][64];
		static thread_local uint32_t Q[K][64], Cst[K][64];
		int base = 0;
		K2aBook book;
		k2a_book_reset(&book);
		for (int s = 0; s < K; ++s)
			for (int l = 0; l < 64; ++l) {
				const int t = s * 64 + l;
				H1[s][l] = H2[s][l] = En[s][l] = E2n[s][l] = Fn[s][l] = K2A_NEG; Q[s][l] = 0;
				Cst[s][l] = cst[k2a_min(t, tlen - 1)];
			}
		for (int r = 0; r < qlen + tlen - 1; ++r) {
			const int st0 = k2a_max(0, r - qlen + 1), en0 = k2a_min(tlen - 1, r), en1 = st0 + (en0 - st0) / 4 * 4;
			while (st0 >= 1 && (st0 - 1) / 64 > base) {                    /* slide the window: slot s <- slot s+1 */
				for (int s = 0; s + 1 < K; ++s)
					for (int l = 0; l < 64; ++l) {
						H1[s][l] = H1[s + 1][l]; H2[s][l] = H2[s + 1][l]; En[s][l] = En[s + 1][l]; E2n[s][l] = E2n[s + 1][l];
						Fn[s][l] = Fn[s + 1][l]; Q[s][l] = Q[s + 1][l]; Cst[s][l] = Cst[s + 1][l];
					}
				++base;
				for (int l = 0; l < 64; ++l) {
					const int t = (base + K - 1) * 64 + l;
					H1[K - 1][l] = H2[K - 1][l] = En[K - 1][l] = E2n[K - 1][l] = Fn[K - 1][l] = K2A_NEG; Q[K - 1][l] = 0;
					Cst[K - 1][l] = cst[k2a_min(t, tlen - 1)];
				}
			}
			const uint32_t qcur = qry[k2a_min(r, qlen - 1)];
			int cH2 = K2A_NEG, cEn = K2A_NEG, cE2n = K2A_NEG;
			uint32_t cQ = 0;
			int A = K2A_NEG, S = K2A_NEG, T[3] = { K2A_NEG, K2A_NEG, K2A_NEG };
			int bH[64], bT[64];
			for (int l = 0; l < 64; ++l) { bH[l] = K2A_NEG; bT[l] = -1; }
			for (int s = 0; s < K; ++s) {
				int h2s[64], ens[64], e2ns[64];
				uint32_t qs[64];
				if ((base + s) * 64 > en0 + 1 || (base + s) * 64 + 63 < st0 - 1) continue;   /* finished / not yet reached */
				for (int l = 0; l < 64; ++l) {                             /* one-lane shift with the previous slot's lane 63 carried in */
					h2s[l] = l ? H2[s][l - 1] : cH2; ens[l] = l ? En[s][l - 1] : cEn; e2ns[l] = l ? E2n[s][l - 1] : cE2n;
					qs[l] = l ? Q[s][l - 1] : cQ;
				}
				cH2 = H2[s][63]; cEn = En[s][63]; cE2n = E2n[s][63]; cQ = Q[s][63];
				const int t0 = (base + s) * 64;
				const bool slot_live = t0 <= en0 && t0 + 63 >= st0;
				for (int l = 0; l < 64; ++l) {
					const int t = t0 + l;
					Q[s][l] = t == 0 ? qcur : qs[l];
					if (!slot_live) continue;
					const bool active = t >= st0 && t <= en0, first_row = t == 0, first_col = t == r;
					const int diag = first_row ? k2a_dm_border(sp, r) : first_col ? k2a_dm_border(sp, t) : h2s[l];
					const int ein = first_row ? k2a_dm_border(sp, r + 1) - sp.q - sp.e : ens[l];
					const int e2in = first_row ? k2a_dm_border(sp, r + 1) - sp.q2 : e2ns[l];
					const int fin = first_col ? k2a_dm_border(sp, t + 1) - sp.q - sp.e : Fn[s][l];
					const uint32_t c = Cst[s][l];
					const int sc = (int)sp.mat[(c & 0xffu) * (uint32_t)sp.m + (Q[s][l] & 0xffu)];
					int z, en, e2n, fn;
					uint32_t dir;
					k2a_dm_cell<MODE>(sp, diag, ein, e2in, fin, sc, c, z, en, e2n, fn, dir);
					if (!active) continue;
					H2[s][l] = H1[s][l]; H1[s][l] = z; En[s][l] = en; E2n[s][l] = e2n; Fn[s][l] = fn;
					if (MODE != K2A_MODE_SCORE) tbp[(size_t)r * ncol + (t - st0)] = (uint8_t)dir;
					if (t == en0) A = z;
					if (t == st0) S = z;
					if (t >= en1 && t < en0) T[t - en1] = z;
					if (t < en1 && z > bH[l]) { bH[l] = z; bT[l] = t; }
				}
			}
			uint64_t Bkey = 0;
			for (int l = 0; l < 64; ++l)
				if (bT[l] >= 0) { const uint64_t k = k2a_dm_key(bH[l], bT[l], st0); if (k > Bkey) Bkey = k; }
			if (k2a_dm_book(&book, r, st0, en0, qlen, tlen, pr.zdrop, A, Bkey, T[0], T[1], T[2], S)) break;
		}
		k2a_finish(pr, book, &res[pi]);
	}
}

/* mirrors k2a_exts_big_kernel: state in a scratch array, double-buffered by diagonal parity */
template<int MODE>
static void sim_exts_big(const K2aSplice sp, const K2aPair *pairs, const uint32_t *order, int ntasks, const uint8_t *seq, uint8_t *tb,
                         int32_t *scratch, K2aResult *res)
{
	for (int task = 0; task < ntasks; ++task) {
		const uint32_t pi = order[task];
		const K2aPair pr = pairs[pi];
		const int qlen = pr.qlen, tlen = pr.tlen_full, ncol = k2a_min(qlen, tlen);
		const uint8_t *qry = seq + pr.qoff;
		const uint32_t *cst = (const uint32_t*)seq + pr.bnd_off;
		uint8_t *tbp = tb + pr.tb_off;
		int32_t *W = scratch + (size_t)pr.pad * 4;
		K2aBook book;
		k2a_book_reset(&book);
		for (int r = 0; r < qlen + tlen - 1; ++r) {
			const int st0 = k2a_max(0, r - qlen + 1), en0 = k2a_min(tlen - 1, r), en1 = st0 + (en0 - st0) / 4 * 4;
			int32_t *Hc = W + (size_t)(r % 3) * tlen;
			const int32_t *H2 = W + (size_t)((r + 1) % 3) * tlen;
			int32_t *Ec = W + (size_t)(3 + (r & 1)) * tlen, *E2c = W + (size_t)(5 + (r & 1)) * tlen, *Fc = W + (size_t)(7 + (r & 1)) * tlen;
			const int32_t *Ep = W + (size_t)(3 + ((r + 1) & 1)) * tlen, *E2p = W + (size_t)(5 + ((r + 1) & 1)) * tlen, *Fp = W + (size_t)(7 + ((r + 1) & 1)) * tlen;
			int A = K2A_NEG, S = K2A_NEG, T[3] = { K2A_NEG, K2A_NEG, K2A_NEG };
			int bH[64], bT[64];
			for (int l = 0; l < 64; ++l) { bH[l] = K2A_NEG; bT[l] = -1; }
			for (int t = st0; t <= en0; ++t) {
				const bool first_row = t == 0, first_col = t == r;
				const int diag = first_row ? k2a_dm_border(sp, r) : first_col ? k2a_dm_border(sp, t) : H2[t - 1];
				const int ein = first_row ? k2a_dm_border(sp, r + 1) - sp.q - sp.e : Ep[t - 1];
				const int e2in = first_row ? k2a_dm_border(sp, r + 1) - sp.q2 : E2p[t - 1];
				const int fin = first_col ? k2a_dm_border(sp, t + 1) - sp.q - sp.e : Fp[t];
				const uint32_t c = cst[t];
				const int sc = (int)sp.mat[(c & 0xffu) * (uint32_t)sp.m + qry[r - t]];
				int z, en, e2n, fn;
				uint32_t dir;
				k2a_dm_cell<MODE>(sp, diag, ein, e2in, fin, sc, c, z, en, e2n, fn, dir);
				Hc[t] = z; Ec[t] = en; E2c[t] = e2n; Fc[t] = fn;
				if (MODE != K2A_MODE_SCORE) tbp[(size_t)r * ncol + (t - st0)] = (uint8_t)dir;
				if (t == en0) A = z;
				if (t == st0) S = z;
				if (t >= en1 && t < en0) T[t - en1] = z;
				if (t < en1 && z > bH[t & 63]) { bH[t & 63] = z; bT[t & 63] = t; }
			}
			uint64_t Bkey = 0;
			for (int l = 0; l < 64; ++l)
				if (bT[l] >= 0) { const uint64_t k = k2a_dm_key(bH[l], bT[l], st0); if (k > Bkey) Bkey = k; }
			if (k2a_dm_book(&book, r, st0, en0, qlen, tlen, pr.zdrop, A, Bkey, T[0], T[1], T[2], S)) break;
		}
		k2a_finish(pr, book, &res[pi]);
	}
}

static void sim_exts_trace(const K2aSplice sp, const K2aPair *pairs, const uint32_t *order, int ntasks, const uint8_t *tb, K2aResult *res, uint32_t *cig)
{
	for (int task = 0; task < ntasks; ++task) {
		const K2aPair pr = pairs[order[task]];
		K2aResult &r = res[order[task]];
		r.n_cigar = r.ti >= 0 ? k2a_dm_trace(tb + pr.tb_off, k2a_min(pr.qlen, pr.tlen_full), r.ti, r.tj, cig + pr.cig_off, pr.qlen, sp.long_thres) : 0;
	}
}



/* mirrors k2a_ssec_kernel: one alignment per "wavefront", 64 positions per pass, the reference's byte arrays in `scratch` */
template<bool DUAL, int MODE>
static void sim_ssec(const K2aSsec P, const K2aPair *pairs, const uint32_t *order, int ntasks, const uint8_t *seq, uint8_t *tb, uint8_t *scratch,
                     K2aResult *res)
{
	for (int task = 0; task < ntasks; ++task) {
		const uint32_t pi = order[task];
		const K2aPair pr = pairs[pi];
		const int qlen = pr.qlen, tlen = pr.tlen_full, w = pr.w, T16 = (tlen + 15) / 16 * 16, ncol = k2a_ssec_ncol(qlen, tlen, w);
		const bool approx = (pr.pad & K2A_SSEC_APPROX) != 0, adrop = (pr.pad & K2A_SSEC_APPROX_DROP) != 0, generic = (pr.pad & K2A_SSEC_GENERIC) != 0;
		const uint8_t *qry = seq + pr.qoff, *tgt = seq + pr.toff;
		uint8_t *tbp = tb + pr.tb_off;
		uint8_t *U = scratch + (size_t)pr.bnd_off * 16, *V = U + T16, *X = V + T16, *Y = X + T16;
		uint8_t *X2 = DUAL ? Y + T16 : Y, *Y2 = DUAL ? X2 + T16 : Y, *S = (DUAL ? Y2 : Y) + T16;
		int32_t *H = (int32_t*)(S + T16);
		const int slope = DUAL ? P.e2 : P.e;
		for (int x = 0; x < T16; ++x) {
			const uint8_t g1 = DUAL ? (uint8_t)(-P.q - P.e) : 0, g2 = (uint8_t)(-P.q2 - P.e2);
			U[x] = g1; V[x] = g1; X[x] = g1; Y[x] = g1; S[x] = 0;
			if (DUAL) { X2[x] = g2; Y2[x] = g2; }
			H[x] = K2A_NEG;
		}
		K2aBook book;
		k2a_book_reset(&book);
		K2aSsecFollow fol = { 0, 0 };
		int last_st = -1, last_en = -1;
		for (int r = 0; r < qlen + tlen - 1; ++r) {
			int st0, en0, st, en;
			if (!k2a_ssec_bounds(r, qlen, tlen, w, st0, en0, st, en)) { book.dropped = 1; break; }
			int cx, cv, cx2 = 0;
			const bool prev_ok = st > 0 && st - 1 >= last_st && st - 1 <= last_en;
			if (!DUAL) {
				cx = prev_ok ? k2a_s8(X[st - 1]) : 0;
				cv = st > 0 ? (prev_ok ? k2a_s8(V[st - 1]) : 0) : (r ? P.q : 0);
				if (en >= r) { Y[r] = 0; U[r] = (uint8_t)(r ? P.q : 0); }
			} else {
				const int edge = k2a_ssec_edge(P, r);
				cx = prev_ok ? k2a_s8(X[st - 1]) : -P.q - P.e;
				cx2 = prev_ok ? k2a_s8(X2[st - 1]) : -P.q2 - P.e2;
				cv = st > 0 ? (prev_ok ? k2a_s8(V[st - 1]) : -P.q - P.e) : edge;
				if (en >= r) { Y[r] = (uint8_t)(-P.q - P.e); Y2[r] = (uint8_t)(-P.q2 - P.e2); U[r] = (uint8_t)edge; }
			}
			const int pend = generic ? en0 + 1 : st0 + ((en0 - st0) / 16 + 1) * 16;
			for (int p = st0; p < pend; ++p)
				if (p < T16) S[p] = (uint8_t)k2a_ssec_score(P, generic, k2a_ssec_tcode(tgt, qry, tlen, qlen, T16, p), k2a_ssec_qcode(qry, r, p));
			for (int base = st; base <= en; base += 64) {
				int xo[64], vo[64], x2o[64], uo[64], yo[64], y2o[64], so[64];
				for (int l = 0; l < 64; ++l) {
					const int p = base + l;
					xo[l] = vo[l] = x2o[l] = uo[l] = yo[l] = y2o[l] = so[l] = 0;
					if (p <= en) { xo[l] = k2a_s8(X[p]); vo[l] = k2a_s8(V[p]); uo[l] = k2a_s8(U[p]); yo[l] = k2a_s8(Y[p]); so[l] = k2a_s8(S[p]); if (DUAL) { x2o[l] = k2a_s8(X2[p]); y2o[l] = k2a_s8(Y2[p]); } }
				}
				for (int l = 0; l < 64; ++l) {
					const int p = base + l;
					const int xt1 = l ? xo[l - 1] : cx, vt1 = l ? vo[l - 1] : cv, x2t1 = DUAL ? (l ? x2o[l - 1] : cx2) : 0;
					int un, vn, xn, yn, x2n, y2n;
					uint32_t dir;
					k2a_ssec_cell<DUAL, MODE>(P, so[l], xt1, vt1, x2t1, uo[l], yo[l], y2o[l], un, vn, xn, yn, x2n, y2n, dir);
					if (p <= en) {
						U[p] = (uint8_t)un; V[p] = (uint8_t)vn; X[p] = (uint8_t)xn; Y[p] = (uint8_t)yn;
						if (DUAL) { X2[p] = (uint8_t)x2n; Y2[p] = (uint8_t)y2n; }
						if (MODE != K2A_MODE_SCORE) tbp[(size_t)r * ncol + (p - st)] = (uint8_t)dir;
					}
				}
				cx = xo[63]; cv = vo[63]; cx2 = x2o[63];
			}
			int stop;
			if (!approx) {
				int A, Sv, T[3] = { K2A_NEG, K2A_NEG, K2A_NEG };
				const int en1 = st0 + (en0 - st0) / 4 * 4;
				uint64_t Bkey = 0;
				if (r > 0) {
					const int hprev = en0 > 0 ? H[en0 - 1] : H[en0];
					A = hprev + k2a_ssec_dh<DUAL>(P, en0 > 0 ? U[en0] : V[en0]);
					Sv = A;
					int bH[64], bT[64];
					for (int l = 0; l < 64; ++l) { bH[l] = K2A_NEG; bT[l] = -1; }
					for (int t = st0; t < en0; ++t) {
						const int h = H[t] + k2a_ssec_dh<DUAL>(P, V[t]), l = (t - st0) & 63;
						H[t] = h;
						if (t < en1 && h > bH[l]) { bH[l] = h; bT[l] = t; }
						if (t == st0) Sv = h;
						if (t >= en1) T[t - en1] = h;
					}
					H[en0] = A;
					for (int l = 0; l < 64; ++l)
						if (bT[l] >= 0) { const uint64_t k = k2a_dm_key(bH[l], bT[l], st0); if (k > Bkey) Bkey = k; }
				} else { A = Sv = k2a_ssec_dh<DUAL>(P, V[0]) - (DUAL ? P.qe_first : P.q + P.e); H[0] = A; }
				stop = k2a_ssec_book(&book, r, st0, en0, en, qlen, tlen, pr.zdrop, slope, A, Bkey, T[0], T[1], T[2], Sv);
			} else {
				const int l0 = k2a_min(k2a_max(fol.last, 0), T16 - 1), l1 = k2a_min(k2a_max(fol.last + 1, 0), T16 - 1);
				stop = k2a_ssec_follow<DUAL>(P, fol, &book, r, st0, en0, qlen, tlen, pr.zdrop, adrop, V[l0], U[l1], V[0]);
			}
			if (stop) break;
			last_st = st; last_en = en;
		}
		k2a_finish(pr, book, &res[pi]);
	}
}

/* mirrors k2a_ssec_blk_kernel: 64 lanes, each one 16-position block of the ring; the phases of an anti-diagonal in the kernel's order */
template<bool DUAL, int MODE>
static void sim_ssec_blk(const K2aSsec P, const K2aPair *pairs, const uint32_t *order, int ntasks, const uint8_t *seq, uint8_t *tb, K2aResult *res)
{
	for (int task = 0; task < ntasks; ++task) {
		const uint32_t pi = order[task];
		const K2aPair pr = pairs[pi];
		const int qlen = pr.qlen, tlen = pr.tlen_full, w = pr.w, T16 = (tlen + 15) / 16 * 16;
		const bool approx = (pr.pad & K2A_SSEC_APPROX) != 0, adrop = (pr.pad & K2A_SSEC_APPROX_DROP) != 0;
		const uint8_t *qry = seq + pr.qoff, *tgt = seq + pr.toff;
		const int slope = DUAL ? P.e2 : P.e;
		std::vector<K2aSsecBlk<DUAL>> Bv(64);
		std::vector<int> hl(K2A_SSECB_RING_WORDS, 0);
		for (int l = 0; l < 64; ++l) {
			K2aSsecBlk<DUAL> &B = Bv[l];
			B.blk = -1; B.qn = 0;
			B.U = B.V = B.X = B.Y = B.X2 = B.Y2 = B.S = B.P0 = B.P1 = B.QW = k2a_blk{ 0, 0, 0, 0, 0, 0, 0, 0 };
		}
		K2aBook book;
		k2a_book_reset(&book);
		K2aSsecFollow fol = { 0, 0 };
		int last_st = -1, last_en = -1, last_st0 = 0, last_en0 = 0, hprev = 0;
		for (int r = 0; r < qlen + tlen - 1; ++r) {
			int st0, en0, st, en;
			if (!k2a_ssec_bounds(r, qlen, tlen, w, st0, en0, st, en)) { book.dropped = 1; break; }
			const int pend = k2a_min(st0 + ((en0 - st0) / 16 + 1) * 16, T16);
			const int need = k2a_max(en, pend - 1) >> 4;
			const bool prev_ok = st > 0 && st - 1 >= last_st && st - 1 <= last_en;
			int cv, cx, cx2 = 0;
			if (!DUAL) { cx = 0; cv = st > 0 ? 0 : (r ? P.q : 0); }
			else { cx = -P.q - P.e; cx2 = -P.q2 - P.e2; cv = st > 0 ? -P.q - P.e : k2a_ssec_edge(P, r); }
			for (int l = 0; l < 64; ++l) {
				K2aSsecBlk<DUAL> &B = Bv[l];
				B.shift_query(P, r);
				const int nb = B.blk < 0 ? l : B.blk + 64;
				if (nb <= need) {
					B.init_block(P, nb, tgt, tlen, qry, qlen, r);
					for (int s2 = 0; s2 < 16; ++s2) hl[k2a_ssecb_slot(nb << 4) + s2] = K2A_NEG;
				}
				B.ask_query(qry, qlen, r);
				if (en >= r && B.blk == (r >> 4)) {
					if (!DUAL) { k2a_sb_set(B.Y, r & 15, 0); k2a_sb_set(B.U, r & 15, r ? P.q : 0); }
					else { k2a_sb_set(B.Y, r & 15, -P.q - P.e); k2a_sb_set(B.Y2, r & 15, -P.q2 - P.e2); k2a_sb_set(B.U, r & 15, k2a_ssec_edge(P, r)); }
				}
			}
			bool act[64];
			int hv[64][16], hnew = 0;
			for (int l = 0; l < 64; ++l) {
				act[l] = Bv[l].blk >= (st >> 4) && Bv[l].blk <= (en >> 4);
				if (!approx && r > 0 && act[l]) for (int s2 = 0; s2 < 16; ++s2) hv[l][s2] = hl[k2a_ssecb_slot(Bv[l].p0()) + s2];
			}
			if (!approx && r > 0) hnew = hl[k2a_ssecb_slot(en0 > 0 ? en0 - 1 : en0)];
			uint32_t pv[64], px[64], px2[64];
			for (int l = 0; l < 64; ++l) { const K2aSsecBlk<DUAL> &Bp = Bv[(l + 63) & 63]; pv[l] = Bp.V[7]; px[l] = Bp.X[7]; px2[l] = DUAL ? Bp.X2[7] : 0u; }
			for (int l = 0; l < 64; ++l) {
				K2aSsecBlk<DUAL> &B = Bv[l];
				if (B.blk == (st >> 4) && !prev_ok) { pv[l] = k2a_sb_c(cv); px[l] = k2a_sb_c(cx); px2[l] = k2a_sb_c(cx2); }
				B.refresh_scores(P, st0, pend);
				if (act[l]) {
					uint32_t dirw[4];
					B.template update<MODE>(P, pv[l], px[l], px2[l], dirw);
					if (MODE != K2A_MODE_SCORE) memcpy(tb + pr.tb_off + (size_t)r * k2a_ssec_ncol(qlen, tlen, w) + (size_t)(B.p0() - st), dirw, 16);
				}
			}
			int stop;
			if (!approx) {
				int A, Sv, T[3] = { K2A_NEG, K2A_NEG, K2A_NEG };
				uint64_t Bkey = 0;
				const int en1 = st0 + (en0 - st0) / 4 * 4;
				if (r > 0) {
					if (!(en0 == last_en0 && en0 - 1 < last_st0 && en0 > 0)) hprev = hnew;
					const K2aSsecBlk<DUAL> &Bo = Bv[(en0 >> 4) & 63];
					const int dl = (int)(en0 > 0 ? k2a_sb_get(Bo.U, en0 & 15) : k2a_sb_get(Bo.V, en0 & 15));
					A = hprev + k2a_ssec_dh<DUAL>(P, dl);
					for (int l = 0; l < 64; ++l) if (act[l]) { const uint64_t k = Bv[l].advance_H(P, hl.data(), hv[l], st0, en1); if (k > Bkey) Bkey = k; }
					Sv = st0 < en0 ? hl[k2a_ssecb_slot(st0)] : A;
					for (int k = 0; k < 3; ++k) if (en1 + k < en0) T[k] = hl[k2a_ssecb_slot(en1 + k)];
					hl[k2a_ssecb_slot(en0)] = A;
				} else {
					A = Sv = k2a_ssec_dh<DUAL>(P, (int)k2a_sb_get(Bv[0].V, 0)) - (DUAL ? P.qe_first : P.q + P.e);
					hl[0] = A;
				}
				stop = k2a_ssec_book(&book, r, st0, en0, en, qlen, tlen, pr.zdrop, slope, A, Bkey, T[0], T[1], T[2], Sv);
			} else {
				const int l0 = k2a_min(k2a_max(fol.last, 0), T16 - 1), l1 = k2a_min(k2a_max(fol.last + 1, 0), T16 - 1);
				const int vl = (int)k2a_sb_get(Bv[(l0 >> 4) & 63].V, l0 & 15), un = (int)k2a_sb_get(Bv[(l1 >> 4) & 63].U, l1 & 15);
				const int v0 = (int)k2a_sb_get(Bv[0].V, 0);
				stop = k2a_ssec_follow<DUAL>(P, fol, &book, r, st0, en0, qlen, tlen, pr.zdrop, adrop, vl, un, v0);
			}
			if (stop) break;
			last_st = st; last_en = en; last_st0 = st0; last_en0 = en0;
		}
		k2a_finish(pr, book, &res[pi]);
	}
}

/* mirrors k2a_fill_pkmp_kernel.  The generations of a task run one after the other here (on the device four wavefronts pipeline
 * them; the data flow -- boundary entries through `bnd`, row-maximum keys in the task's spill blocks, a re-base every
 * K2A_PKMP_T steps -- is the same, and so is every value). */
template<bool DUAL, int MODE>
static void sim_fill_pkmp(const K2aScoring sc, const K2aPair *pairs, const uint32_t *order2, int ntasks, const uint8_t *seq, uint8_t *tb,
                          uint32_t *bnd, K2aResult *res)
{
	constexpr int C = 16, G = 64, W = K2A_PKMP_WAVES, T = K2A_PKMP_T, R = G * C;
	typedef K2aLanePkMp<C, DUAL, MODE, true> Lane;
	constexpr int WB = Lane::TBWORDS * 4;
	for (int task = 0; task < ntasks; ++task) {
		static thread_local Lane L[64];
		const uint32_t piA = order2[2 * task], piB = order2[2 * task + 1];
		const K2aPair prA = pairs[piA], prB = pairs[piB];
		const int qlen = prA.qlen, tlen = prA.tlen, w = prA.w, ngen = (tlen + R - 1) / R;
		K2aBook bkA, bkB;
		uint32_t rowbuf[C];
		k2a_book_reset(&bkA); k2a_book_reset(&bkB);
		uint32_t *B1 = bnd + prA.bnd_off, *B2 = B1 + 4 * (size_t)qlen;
		unsigned long long *spill = (unsigned long long*)(bnd + prA.bnd_off + K2A_PKMP_BND_WORDS(qlen, DUAL));
		const size_t tbsteps = MODE != K2A_MODE_SCORE ? k2a_tb_steps<G, C, true>(qlen, tlen, w) : 0;
		const int ktop = k2a_min(qlen - 1, k2a_min(C - 1, tlen - 1) + w);
		const k2a_pk neg = k2a_pku(K2A_NEG16);
		size_t kbase = 0;
		bool stop = false;
		for (int g = 0; g < ngen && !stop; ++g) {
			int jlo, nsteps;
			k2a_gen_cols<G, C>(g, qlen, tlen, w, &jlo, &nsteps);
			const int wave = g % W, je_prev = g > 0 ? k2a_min(qlen - 1, g * R - 1 + w) : -1;
			/* (a wavefront's lane state does not survive begin_generation / do_init -- bases are re-derived at every init -- so
			 * setting the lanes up afresh for every generation is equivalent to the device's wavefront g mod 4 carrying on) */
			for (int l = 0; l < 64; ++l) {
				L[l].setup(prA, prB, seq, l, true, spill + (size_t)wave * (K2A_PKMP_SPILL_WORDS(C) / 2), sc.cp);
				L[l].begin_generation(g, jlo); L[l].clear_spill();
			}
			int bs0A = 0, bs0B = 0;
			auto fetch = [&](int j, uint32_t v[4], uint32_t &v2) {
				v[0] = v[1] = neg; v[2] = v[3] = 0; v2 = neg;
				if (j <= je_prev) { for (int x = 0; x < 4; ++x) v[x] = B1[4 * (size_t)j + x]; if (DUAL) v2 = B2[j]; }
			};
			if (g > 0 && jlo > 0) { uint32_t pv[4], pv2; fetch(jlo - 1, pv, pv2); L[0].P.hu_prev = pv[0]; bs0A = (int)pv[2]; bs0B = (int)pv[3]; }
			for (int l = 0; l < 64; ++l) L[l].P.set_qb(L[l].P.next_query_codes(-1));
			for (int k = 0; k < nsteps && !stop; ++k) {
				k2a_pk hin[64], ein[64], e2in[64];
				uint32_t qnext[64];
				for (int l = 0; l < 64; ++l) { const int src = (l + 63) % 64; hin[l] = L[src].P.hout; ein[l] = L[src].P.eout; e2in[l] = DUAL ? L[src].P.e2out : 0u; }
				bool anyinit = false;
				for (int l = 0; l < 64; ++l) anyinit |= L[l].need_init(k);
				if (anyinit) {
					int bsA[64], bsB[64];
					for (int l = 0; l < 64; ++l) { bsA[l] = L[(l + 63) % 64].P.baseA; bsB[l] = L[(l + 63) % 64].P.baseB; }
					if (g > 0) { bsA[0] = bs0A; bsB[0] = bs0B; }
					for (int l = 0; l < 64; ++l) if (L[l].need_init(k)) L[l].do_init(sc, bsA[l], bsB[l]);
					for (int l = 0; l < 64; ++l) { bsA[l] = L[(l + 63) % 64].P.baseA; bsB[l] = L[(l + 63) % 64].P.baseB; }
					for (int l = 0; l < 64; ++l) L[l].refresh_delta(bsA[l], bsB[l]);
				}
				if (g > 0) {
					uint32_t v[4], v2;
					fetch(jlo + k, v, v2);
					hin[0] = v[0]; ein[0] = v[1]; e2in[0] = v2;
					L[0].refresh_delta((int)v[2], (int)v[3]);
				}
				bool nfin[64], anyfin = false;
				{	/* K2A_SYNC_WN */
					bool wn = false;
					for (int l = 0; l < 64; ++l) wn |= L[l].P.hasn != 0;
					for (int l = 0; l < 64; ++l) L[l].P.wn = wn;
				}
				for (int l = 0; l < 64; ++l) {
					L[l].P.hu_prev = hin[l];
					hin[l] = L[l].adopt(hin[l]); ein[l] = L[l].adopt(ein[l]);
					if (DUAL) e2in[l] = L[l].adopt(e2in[l]);
					qnext[l] = L[l].P.next_query_codes(k);
					if (g == 0 && k <= ktop) L[l].P.top_inputs(sc, k, hin[l], ein[l], e2in[l]);
					uint32_t tw[Lane::TBWORDS];
					const int jj = L[l].column(k);
					const bool mine = L[l].P.S >= 0 && jj >= 0 && jj <= L[l].P.je;
					const bool live = L[l].P.step(sc, k, hin[l], ein[l], e2in[l], tw);
					if (MODE != K2A_MODE_SCORE && live) memcpy(tb + prA.tb_off + k2a_tb_word(kbase + (size_t)k, l, tbsteps, G, WB), tw, sizeof(tw));
					if (l == G - 1 && g + 1 < ngen && mine) {
						B1[4 * (size_t)jj] = L[l].P.hout; B1[4 * (size_t)jj + 1] = L[l].P.eout; B1[4 * (size_t)jj + 2] = (uint32_t)L[l].P.baseA; B1[4 * (size_t)jj + 3] = (uint32_t)L[l].P.baseB;
						if (DUAL) B2[jj] = L[l].P.e2out;
					}
					nfin[l] = L[l].need_fin(k);
					anyfin |= nfin[l];
				}
				if (anyfin) {
					for (int l = 0; l < 64; ++l) if (nfin[l]) { L[l].flush_rowmax(); L[l].do_fin(sc, &bkA, &bkB, prA.zdrop, prB.zdrop, rowbuf); }
					if (bkA.dropped && bkB.dropped) stop = true;
				}
				for (int l = 0; l < 64; ++l) L[l].P.set_qb(qnext[l]);
				if ((k & (T - 1)) == T - 1) {
					k2a_pk d[64];
					for (int l = 0; l < 64; ++l) d[l] = L[l].rebase();
					int bsA[64], bsB[64];
					for (int l = 0; l < 64; ++l) { bsA[l] = L[(l + 63) % 64].P.baseA; bsB[l] = L[(l + 63) % 64].P.baseB; }
					for (int l = 0; l < 64; ++l) L[l].after_rebase(d[(l + 63) % 64], bsA[l], bsB[l]);
				}
			}
			kbase += k2a_gen_pad(nsteps);
		}
		k2a_finish(prA, bkA, &res[piA]);
		if (piB != piA) k2a_finish(prB, bkB, &res[piB]);
	}
}

/* mirrors k2a_fill_solo_kernel: one alignment per wavefront, both halves of every lane */
template<int C, bool DUAL, int MODE>
static void sim_fill_solo(const K2aScoring sc, const K2aPair *pairs, const uint32_t *order, int ntasks, const uint8_t *seq, uint8_t *tb,
                          K2aResult *res)
{
	typedef K2aLaneSolo<C, DUAL, MODE, true> Lane;
	for (int task = 0; task < ntasks; ++task) {
		const uint32_t pi = order[task];
		const K2aPair pr = pairs[pi];
		static thread_local Lane L[64];
		K2aBook book;
		uint32_t rowbuf[K2A_SOLO_STAGE(C)];
		k2a_book_reset(&book);
		for (int l = 0; l < 64; ++l) { L[l].setup(pr, seq, l, true, sc.cp); }
		const int klast = L[0].last_step();
		const int ktop = k2a_min(pr.qlen - 1, k2a_min(C - 1, pr.tlen - 1) + pr.w);
		const size_t tbsteps = k2a_solo_steps<C>(pr.qlen, pr.tlen, pr.w);
		uint8_t *tbp = tb + pr.tb_off;
		uint32_t qp[64];
		for (int l = 0; l < 64; ++l) { qp[l] = 0; L[l].load_query_group(0, L[l].knext == 0 ? L[l].koff_next : L[l].koff, L[l].qw); }
		bool done = false;
		for (int k = 0; k <= klast && !done; ++k) {
			k2a_pk rh[64], re[64], re2[64];
			int rb[64];
			for (int l = 0; l < 64; ++l) {
				const int src = (l + 63) & 63;
				rh[l] = L[src].hout; re[l] = L[src].eout; re2[l] = L[src].e2out; rb[l] = L[src].baseB;
			}
			k2a_pk oh[64], oe[64], oe2[64];
			for (int l = 0; l < 64; ++l) { oh[l] = L[l].hout; oe[l] = L[l].eout; oe2[l] = L[l].e2out; }
			for (int l = 0; l < 64; ++l) {                        /* (every lane's do_init comes before any lane's step, as on the device) */
				if (L[l].need_init(k)) L[l].do_init(sc, rb[l]);
				if (L[l].need_init_high(k)) L[l].start_high(sc);
			}
			{	/* K2A_SYNC_WN */
				bool wn = false;
				for (int l = 0; l < 64; ++l) wn |= L[l].hasn != 0;
				for (int l = 0; l < 64; ++l) L[l].wn = wn;
			}
			for (int l = 0; l < 64; ++l) {
				L[l].hu_prev = rh[l];
				k2a_pk hin = (rh[l] >> 16) | (oh[l] << 16), ein = (re[l] >> 16) | (oe[l] << 16), e2in = DUAL ? (re2[l] >> 16) | (oe2[l] << 16) : 0u;
				hin = k2a_pk_add(hin, L[l].delta); ein = k2a_pk_add(ein, L[l].delta); if (DUAL) e2in = k2a_pk_add(e2in, L[l].delta);
				if ((k & 3) == 0) L[l].load_query_group(k + 4, L[l].knext <= k + 4 ? L[l].koff_next : L[l].koff, qp[l]);
				L[l].advance_query(k & 3);
				if (k <= ktop) L[l].top_inputs(sc, k, hin, ein, e2in);
				uint32_t tw[Lane::TBWORDS];
				const bool live = L[l].step(sc, k, hin, ein, e2in, tw);
				if (MODE != K2A_MODE_SCORE && live)
					memcpy(tbp + k2a_tb_word((size_t)k, l, tbsteps, 64, Lane::TBWORDS * 4), tw, sizeof(tw));
				if (L[l].need_save(k)) L[l].save_low();
			}
			for (int l = 0; l < 64; ++l) {
				if (!L[l].need_fin(k)) continue;
				if (!L[l].fin_fast(sc, &book, pr.zdrop)) { L[l].stage_rows(rowbuf); L[l].do_fin_seq(sc, &book, pr.zdrop, rowbuf); }
				if (book.dropped) done = true;
			}
			if ((k & 3) == 3) for (int l = 0; l < 64; ++l) { L[l].qw = qp[l]; }
		}
		k2a_finish(pr, book, &res[pi]);
		bool saw = false;
		if (sc.pk_tn1) saw = sim_codes_above4(seq + pr.toff, pr.tlen_full);      /* k2a_scan_codes */
		else for (int l = 0; l < 64; ++l) saw |= L[l].saw_wildcard();
		if (saw) res[pi].pad[0] = 1;
	}
}

template<int C>
static void sim_trace_solo(const K2aPair *pairs, const uint32_t *order, int ntasks, const uint8_t *tb, K2aResult *res, uint32_t *cig)
{
	for (int task = 0; task < ntasks; ++task) {
		const K2aPair pr = pairs[order[task]];
		K2aResult &r = res[order[task]];
		r.n_cigar = r.ti >= 0 ? k2a_trace_solo<C>(tb + pr.tb_off, r.ti, r.tj, cig + pr.cig_off, pr.qlen, pr.tlen, pr.w) : 0;
	}
}

/* mirrors k2a_extf_kernel: one alignment per wavefront, passes of 64 target positions, V's neighbour through the lane shift */
static void sim_extf(const K2aExtf par, const K2aPair *pairs, const uint32_t *order, int ntasks, const uint8_t *seq, uint8_t *scratch,
                     K2aResult *res, bool state_hbm)
{
	for (int task = 0; task < ntasks; ++task) {
		const uint32_t pi = order[task];
		const K2aPair pr = pairs[pi];
		const int qlen = pr.qlen, tlen = pr.tlen, w = pr.w, xdrop = pr.zdrop, tpad = (tlen + 15) & ~15;
		const uint8_t *qa = seq + pr.qoff, *ta = seq + pr.toff;
		std::vector<uint8_t> lds((size_t)3 * tpad + 16);
		uint8_t *U = state_hbm ? scratch + pr.tb_off : lds.data(), *V = U + tpad, *S = V + tpad;
		const uint32_t two_e = (uint32_t)(par.e * 2) & 0xffu;
		memset(U, 0, (size_t)3 * tpad);
		K2aExtfBook bk;
		k2a_extf_book_reset(bk);
		int prev_lo = -1, prev_hi = -1, r;
		const int nr = qlen + tlen - 1;
		for (r = 0; r < nr; ++r) {
			K2aExtfDiag d;
			if (!k2a_extf_diag(r, qlen, tlen, w, tpad, d)) break;
			uint32_t carry = (d.blo > 0 && d.blo - 1 >= prev_lo && d.blo - 1 <= prev_hi) ? V[d.blo - 1] : 0u;
			const int last = k2a_max(d.bhi, d.fresh_end - 1);
			const bool top0 = d.bhi >= r;
			for (int base = d.blo; base <= last; base += 64) {
				uint32_t vold[64], b[64], sv[64], u[64], v[64];
				for (int lane = 0; lane < 64; ++lane) {         /* loads of the whole wavefront first, as on the GPU */
					const int x = base + lane;
					const bool act = x <= d.bhi, fresh = x >= d.lo && x < d.fresh_end;
					vold[lane] = act ? V[x] : 0u; b[lane] = act ? U[x] : 0u;
					sv[lane] = fresh ? k2a_extf_score(par, qa, ta, qlen, tlen, r, x) : act ? S[x] : 0u;
					if (top0 && x == r) b[lane] = 0;
				}
				for (int lane = 0; lane < 64; ++lane) k2a_extf_cell(sv[lane], lane ? vold[lane - 1] : carry, b[lane], two_e, u[lane], v[lane]);
				carry = vold[63];
				for (int lane = 0; lane < 64; ++lane) {
					const int x = base + lane;
					if (x <= d.bhi) { U[x] = (uint8_t)u[lane]; V[x] = (uint8_t)v[lane]; }
					if (x >= d.lo && x < d.fresh_end) S[x] = (uint8_t)sv[lane];
				}
			}
			if (!k2a_extf_follow(bk, d, r, par.e, xdrop, V[bk.follow], U[bk.follow + 1])) break;
			prev_lo = d.blo; prev_hi = d.bhi;
		}
		k2a_extf_finish(bk, r, nr, &res[pi]);
	}
}

/* mirrors k2a_extf_win_kernel<K>: the register window, lanes in lock step */
template<int K>
static void sim_extf_win(const K2aExtf par, const K2aPair *pairs, const uint32_t *order, int ntasks, const uint8_t *seq, K2aResult *res)
{
	for (int task = 0; task < ntasks; ++task) {
		const uint32_t pi = order[task];
		const K2aPair pr = pairs[pi];
		const int qlen = pr.qlen, tlen = pr.tlen, w = pr.w, xdrop = pr.zdrop, tpad = (tlen + 15) & ~15;
		const uint8_t *qa = seq + pr.qoff, *ta = seq + pr.toff;
		const uint32_t two_e = (uint32_t)(par.e * 2) & 0xffu;
		static thread_local uint32_t U[K][64], V[K][64], S[K][64], T[K][64];
		int blk[K];
		for (int s = 0; s < K; ++s) {
			blk[s] = s;
			for (int lane = 0; lane < 64; ++lane) { U[s][lane] = V[s][lane] = S[s][lane] = 0u; T[s][lane] = 64 * s + lane < tlen ? ta[64 * s + lane] : 0u; }
		}
		K2aExtfBook bk;
		k2a_extf_book_reset(bk);
		int prev_lo = -1, prev_hi = -1, r;
		const int nr = qlen + tlen - 1;
		for (r = 0; r < nr; ++r) {
			K2aExtfDiag d;
			if (!k2a_extf_diag(r, qlen, tlen, w, tpad, d)) break;
			const int wb = k2a_extf_win_base(d);
			const bool carry_ok = d.blo > 0 && d.blo - 1 >= prev_lo && d.blo - 1 <= prev_hi;
			const int last = k2a_max(d.bhi, d.fresh_end - 1);
			assert(64 * (wb + K) - 1 >= last);
			uint32_t c63[K];
			for (int s = 0; s < K; ++s) {
				const int nb = k2a_extf_win_block<K>(wb, s);
				if (nb != blk[s]) {
					assert(nb > blk[s]);
					blk[s] = nb;
					for (int lane = 0; lane < 64; ++lane) { U[s][lane] = V[s][lane] = S[s][lane] = 0u; T[s][lane] = 64 * nb + lane < tlen ? ta[64 * nb + lane] : 0u; }
				}
				c63[s] = V[s][63];
			}
			for (int s = 0; s < K; ++s) {
				const int base = 64 * blk[s];
				if (base > last || base + 63 < d.blo) continue;
				uint32_t vold[64];
				for (int lane = 0; lane < 64; ++lane) vold[lane] = V[s][lane];
				for (int lane = 0; lane < 64; ++lane) {
					const int x = base + lane, j = r - x;
					const uint32_t qc = (uint32_t)j < (uint32_t)qlen ? (uint32_t)qa[j] : 0u;
					k2a_extf_win_cell(par, d, r, x, two_e, carry_ok, T[s][lane], qc, lane ? vold[lane - 1] : c63[(s + K - 1) & (K - 1)],
					                  U[s][lane], V[s][lane], S[s][lane]);
				}
			}
			assert(bk.follow >= 64 * wb && bk.follow + 1 <= 64 * (wb + K) - 1);
			const uint32_t vf = V[(bk.follow >> 6) & (K - 1)][bk.follow & 63], un = U[((bk.follow + 1) >> 6) & (K - 1)][(bk.follow + 1) & 63];
			if (!k2a_extf_follow(bk, d, r, par.e, xdrop, vf, un)) break;
			prev_lo = d.blo; prev_hi = d.bhi;
		}
		k2a_extf_finish(bk, r, nr, &res[pi]);
	}
}

extern "C" {

const char *k2a_shim_backend(void) { return "sim"; }
const char *k2a_shim_last_error(void) { return g_err; }
/* KSW2AMD_SIM_DEVICES=N pretends to have N devices (all the same host memory) so that the host's multi-device worker
 * pool can be exercised without hardware */
static thread_local int g_sim_dev = 0;
int k2a_shim_device_count(void) { const char *e = getenv("KSW2AMD_SIM_DEVICES"); const int n = e ? atoi(e) : 1; return n > 0 ? n : 1; }
int k2a_shim_simd_count(void) { const char *e = getenv("KSW2AMD_SIM_SIMDS"); return e ? atoi(e) : 0; }      /* tests of the host's chunking rules: a small simulated device */
int k2a_shim_pci_bus_id(char *, int) { return -1; }
int k2a_shim_set_device(int dev) { if (dev < 0 || dev >= k2a_shim_device_count()) return -1; g_sim_dev = dev; return 0; }
int k2a_shim_get_device(void) { return g_sim_dev; }
int k2a_shim_mem_info(size_t *free_b, size_t *total_b) { *free_b = (size_t)8 << 30; *total_b = (size_t)8 << 30; return 0; }
void *k2a_shim_malloc(size_t bytes) { return calloc(bytes ? bytes : 16, 1); }
void k2a_shim_free(void *p) { free(p); }
void *k2a_shim_host_malloc(size_t bytes) { return malloc(bytes ? bytes : 16); }
void k2a_shim_host_free(void *p) { free(p); }
int k2a_shim_async_launches(void) { return 0; }
int k2a_shim_h2d(void *dst, const void *src, size_t bytes, void *) { memcpy(dst, src, bytes); return 0; }
int k2a_shim_d2h(void *dst, const void *src, size_t bytes, void *) { memcpy(dst, src, bytes); return 0; }
int k2a_shim_d2d(void *dst, const void *src, size_t bytes, void *) { memcpy(dst, src, bytes); return 0; }
int k2a_shim_host_register(void *, size_t) { return 0; }
int k2a_shim_host_unregister(void *) { return 0; }
int k2a_shim_memset(void *dst, int v, size_t bytes, void *) { memset(dst, v, bytes); return 0; }
void *k2a_shim_stream_create(void) { return (void*)1; }
void *k2a_shim_stream_create_high(void) { return (void*)1; }
void *k2a_shim_stream_create_low(void) { return (void*)1; }
void k2a_shim_stream_destroy(void *) {}
int k2a_shim_stream_sync(void *) { return 0; }
void *k2a_shim_event_create(void) { return calloc(1, sizeof(double)); }
void k2a_shim_event_destroy(void *ev) { free(ev); }
int k2a_shim_event_sync(void *) { return 0; }
int k2a_shim_stream_wait_event(void *, void *) { return 0; }
int k2a_shim_event_record(void *ev, void *)
{
	*(double*)ev = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
	return 0;
}
float k2a_shim_event_ms(void *a, void *b) { return (float)(*(double*)b - *(double*)a); }

int k2a_shim_launch_fill(int cfg, int dual, int mode, const K2aScoring *sc, const K2aPair *pairs, const uint32_t *order,
                         int ntasks, const uint8_t *seq, uint8_t *tb, int32_t *bnd, K2aResult *res, void *)
{
	if (ntasks <= 0) return 0;
	if (cfg == K2A_CFG_MP) (k2a_shim_mp_form(dual, mode, ntasks) ? g_fill_mp_lds[mode - 1] : g_fill_mp[dual ? 1 : 0][mode])(*sc, pairs, order, ntasks, seq, tb, bnd, res);
	else g_fill[cfg][dual ? 1 : 0][mode](*sc, pairs, order, ntasks, seq, tb, res);
	return 0;
}
#define DEFER_ROW(G, C, LR) { sim_fill_pk<G, C, false, 0, false, false, LR, true>, sim_fill_pk<G, C, false, 0, true, false, LR, true> }
static const fill_pk_fn g_fill_pk_defer[4][2] = { DEFER_ROW(8, 18, 0), DEFER_ROW(16, 8, 2), DEFER_ROW(64, 8, 0), DEFER_ROW(64, 16, 2) };
typedef void (*argmax_fn)(const K2aScoring, const K2aPair*, const uint32_t*, int, const uint8_t*, uint8_t*, K2aResult*);
#define ARGMAX_ROW(G, C) { sim_argmax<G, C, false>, sim_argmax<G, C, true> }
static const argmax_fn g_argmax[4][2] = { ARGMAX_ROW(8, 18), ARGMAX_ROW(16, 8), ARGMAX_ROW(64, 8), ARGMAX_ROW(64, 16) };
typedef void (*zscan_fn)(const K2aScoring, const K2aPair*, const uint32_t*, int, const uint8_t*, const uint8_t*, K2aResult*);
#define ZSCAN_ROW(G, C) { sim_zscan<G, C, false>, sim_zscan<G, C, true> }
static const zscan_fn g_zscan[4][2] = { ZSCAN_ROW(8, 18), ZSCAN_ROW(16, 8), ZSCAN_ROW(64, 8), ZSCAN_ROW(64, 16) };

int k2a_shim_launch_fill_pk(int cfg, int dual, int mode, int rebased, int nomax, int defer, const K2aScoring *sc, const K2aPair *pairs, const uint32_t *order2,
                            int ntasks, const uint8_t *seq, uint8_t *tb, K2aResult *res, K2aQueueDesc *qd, void *)
{
	if (defer && ntasks > 0) {
		g_fill_pk_defer[cfg][rebased ? 1 : 0](*sc, pairs, order2, ntasks, seq, tb, res, qd);
		if (qd && (qd->abort || qd->next != qd->nwt)) return 0;      /* the device's second and third pass leave at once behind an abandoned streamed fill */
		g_argmax[cfg][rebased ? 1 : 0](*sc, pairs, order2, ntasks, seq, tb, res);
		g_zscan[cfg][rebased ? 1 : 0](*sc, pairs, order2, ntasks, seq, tb, res);
		return 0;
	}
	const int form = k2a_shim_pk_form(cfg, dual, mode, nomax, ntasks);
	const bool lds = form == 1, ldc = form == 2;
	if (ntasks > 0) (lds ? g_fill_pk_lds[rebased ? 1 : 0][mode - 1] : ldc ? g_fill_pk_ldscodes[k2a_pkcfg_G[cfg] == 8 ? 1 : k2a_pkcfg_G[cfg] == 16 ? 2 : 0][nomax ? 1 : 0][rebased ? 1 : 0] : g_fill_pk[nomax ? 1 : 0][rebased ? 1 : 0][cfg][dual ? 1 : 0][mode])(*sc, pairs, order2, ntasks, seq, tb, res, qd);
	return 0;
}
int k2a_shim_launch_gather(const K2aGather *tab, int n, uint8_t *dst, void *)
{
	for (int k = 0; k < n; ++k) memcpy(dst + tab[k].dst, (const void*)(uintptr_t)tab[k].src, tab[k].len);
	return 0;
}
int k2a_shim_launch_wire_expand(const uint8_t *src, uint8_t *dst, size_t bytes, int fmt, uint32_t stride, void *)
{
	if (fmt == 2) { sim_wire2_task(src, dst, 0, (uint32_t)(bytes / stride * stride), stride); return 0; }
	for (size_t x = 0; x < bytes >> 3; ++x) {
		uint32_t w4, lo, hi;
		memcpy(&w4, src + 4 * x, 4);
		k2a_wire4_expand(w4, lo, hi);
		memcpy(dst + 8 * x, &lo, 4); memcpy(dst + 8 * x + 4, &hi, 4);
	}
	return 0;
}
int k2a_shim_launch_uniform_layout(const K2aUniform *u, K2aPair *pairs, uint32_t *order2, uint32_t *need, void *)
{
	if (!u) return 0;
	for (uint32_t i = 0; i < u->n; ++i) { pairs[i] = k2a_uniform_pair(*u, i); order2[i] = i; }
	if (need) for (uint32_t wt = 0; wt < (u->ntasks + u->ng - 1) / u->ng; ++wt) need[wt] = k2a_uniform_need(*u, wt);
	return 0;
}

int k2a_shim_launch_trace_pk(int cfg, int dual, const K2aPair *pairs, const uint32_t *order2, int ntasks, const uint8_t *tb,
                             K2aResult *res, uint32_t *cig, void *)
{
	if (ntasks > 0) g_trace_pk[dual ? 1 : 0][cfg](pairs, order2, ntasks, tb, res, cig);
	return 0;
}
int k2a_shim_launch_trace(int cfg, int dual, const K2aPair *pairs, const uint32_t *order, int ntasks, const uint8_t *tb,
                          K2aResult *res, uint32_t *cig, void *)
{
	if (ntasks > 0) g_trace[cfg][dual ? 1 : 0](pairs, order, ntasks, tb, res, cig);
	return 0;
}
/* mirrors k2a_extf_lane_kernel: one extension per lane, the lanes of a wavefront one after the other (they do not interact) */
static void sim_extf_lane(const K2aExtf par, const K2aPair *pairs, const uint32_t *order, int ntasks, const uint8_t *seq, uint8_t *scratch, K2aResult *res)
{
	for (int task = 0; task < ntasks; ++task) {
		const int lane = task & 63;
		const uint32_t pi = order[task];
		const K2aPair pr = pairs[pi];
		const int qlen = pr.qlen, tlen = pr.tlen, w = pr.w, xdrop = pr.zdrop, tpad = (tlen + 15) & ~15;
		const size_t rows = (size_t)pr.pad * 64;
		K2aExtfLaneMem m;
		m.U4 = (uint32_t*)(scratch + pr.tb_off) + lane; m.V4 = m.U4 + rows; m.S4 = m.V4 + rows;
		m.TT = (const uint32_t*)(seq + pr.toff) + lane; m.QR = (const uint32_t*)(seq + pr.qoff) + lane;
		m.ring = par.ring; m.ztop = 0;
		std::vector<uint32_t> ringbuf;
		if (par.ring) {                                     /* mirrors k2a_extf_lane_ring_kernel: the lane's three rings, filled with garbage first */
			ringbuf.assign((size_t)3 * par.ring * 64, 0xdeadbeefu);
			m.U4 = ringbuf.data(); m.V4 = m.U4 + (size_t)par.ring * 64; m.S4 = m.V4 + (size_t)par.ring * 64;
		}
		K2aExtfBook bk;
		k2a_extf_book_reset(bk);
		int prev_lo = -1, prev_hi = -1, r = 0;
		const int nr = qlen + tlen - 1;
		while (r < nr && k2a_extf_lane_diag(par, qlen, tlen, w, tpad, xdrop, r, m, prev_lo, prev_hi, bk)) ++r;
		k2a_extf_finish(bk, r, nr, &res[pi]);
	}
}

extern "C++" {
/* mirrors k2a_extf_grp_kernel: wavefronts of four extensions, 16 lanes each; the phases of an anti-diagonal in the kernel's order */
template<int G>
static void sim_extf_grp(const K2aExtf par, const K2aPair *pairs, const uint32_t *order, int ntasks, const uint8_t *seq, K2aResult *res)
{
	for (int task0 = 0; task0 < ntasks; task0 += 64 / G) {
		std::vector<K2aExtfBlk<G>> Bv(64);
		uint32_t pis[64];
		bool live[64];
		for (int l = 0; l < 64; ++l) {
			const int task = task0 + l / G;
			live[l] = task < ntasks;
			pis[l] = order[live[l] ? task : ntasks - 1];
			Bv[l].start(par, pairs[pis[l]], seq, l & (G - 1), live[l]);
		}
		for (int r = 0; ; ++r) {
			bool any = false;
			for (int l = 0; l < 64; ++l) any |= Bv[l].begin(par, r);
			if (!any) break;
			uint32_t pv[64], vsel[64], usel[64];
			for (int l = 0; l < 64; ++l) { Bv[l].ask(r); pv[l] = Bv[(l & ~(G - 1)) | ((l + G - 1) & (G - 1))].V[7]; }
			for (int l = 0; l < 64; ++l) Bv[l].update(par, pv[l], vsel[l], usel[l]);
			for (int l = 0; l < 64; ++l) Bv[l].finish_diag(par, r, vsel[(l & ~(G - 1)) | Bv[l].vlane()], usel[(l & ~(G - 1)) | Bv[l].ulane()]);
		}
		for (int l = 0; l < 64; l += G) if (live[l]) k2a_extf_finish(Bv[l].bk, Bv[l].rdone, Bv[l].nr, &res[pis[l]]);
	}
}

}

int k2a_shim_launch_extf(int cls, const K2aExtf *par, const K2aPair *pairs, const uint32_t *order, int ntasks, const uint8_t *seq,
                         uint8_t *scratch, K2aResult *res, void *)
{
	if (ntasks > 0 && cls == 7) sim_extf_grp<16>(*par, pairs, order, ntasks, seq, res);
	else if (ntasks > 0 && cls == 8) sim_extf_grp<32>(*par, pairs, order, ntasks, seq, res);
	else if (ntasks > 0 && cls == 9) sim_extf_grp<64>(*par, pairs, order, ntasks, seq, res);
	else if (ntasks > 0 && cls == 6) sim_extf_lane(*par, pairs, order, ntasks, seq, scratch, res);
	else if (ntasks > 0 && cls == 4) sim_extf_win<4>(*par, pairs, order, ntasks, seq, res);
	else if (ntasks > 0 && cls == 5) sim_extf_win<8>(*par, pairs, order, ntasks, seq, res);
	else if (ntasks > 0) sim_extf(*par, pairs, order, ntasks, seq, scratch, res, cls == 3);
	return 0;
}
int k2a_shim_launch_compact(const K2aPair *pairs, const K2aResult *res, const uint32_t *pos, int n, const uint32_t *cig,
                            uint32_t *pool, void *)
{
	for (int i = 0; i < n; ++i)
		for (int k = 0; k < res[i].n_cigar; ++k) pool[pos[i] + k] = cig[pairs[i].cig_off + ((pairs[i].flag & K2A_F_REV_CIGAR) ? k : res[i].n_cigar - 1 - k)];
	return 0;
}

int k2a_shim_launch_splice_const(const K2aPair *pairs, int n, uint8_t *seq, int noncan, int junc_bonus, void *)
{
	for (int i = 0; i < n; ++i) {
		const K2aPair pr = pairs[i];
		if (pr.qlen <= 0 || pr.tlen_full <= 0) continue;
		const uint8_t *T = seq + pr.toff, *J = (pr.flag & K2A_F_HAS_JUNC) ? T + ((pr.tlen_full + 3) & ~3) : 0;
		uint32_t *out = (uint32_t*)seq + pr.bnd_off;
		for (int t = 0; t < pr.tlen_full; ++t) out[t] = k2a_splice_const(T, J, t, pr.tlen_full, pr.flag, noncan, junc_bonus);
	}
	return 0;
}
int k2a_shim_launch_exts(int mode, int win, const K2aSplice *sp, const K2aPair *pairs, const uint32_t *order, int ntasks, const uint8_t *seq,
                         uint8_t *tb, int32_t *scratch, K2aResult *res, void *)
{
	if (win == 2 && ntasks > 0) {
		if (mode == 0) sim_exts_big<0>(*sp, pairs, order, ntasks, seq, tb, scratch, res);
		else if (mode == 1) sim_exts_big<1>(*sp, pairs, order, ntasks, seq, tb, scratch, res);
		else sim_exts_big<2>(*sp, pairs, order, ntasks, seq, tb, scratch, res);
		return 0;
	}
	typedef void (*exts_fn)(const K2aSplice, const K2aPair*, const uint32_t*, int, const uint8_t*, uint8_t*, K2aResult*);
	static const exts_fn fn[3][2] = {
		{ sim_exts<0, K2A_DM_SLOTS_S>, sim_exts<0, K2A_DM_SLOTS> },
		{ sim_exts<1, K2A_DM_SLOTS_S>, sim_exts<1, K2A_DM_SLOTS> },
		{ sim_exts<2, K2A_DM_SLOTS_S>, sim_exts<2, K2A_DM_SLOTS> } };
	if (ntasks > 0) fn[mode][win](*sp, pairs, order, ntasks, seq, tb, res);
	return 0;
}
int k2a_shim_launch_exts_trace(const K2aSplice *sp, const K2aPair *pairs, const uint32_t *order, int ntasks, const uint8_t *tb, K2aResult *res,
                               uint32_t *cig, void *)
{
	if (ntasks > 0) sim_exts_trace(*sp, pairs, order, ntasks, tb, res, cig);
	return 0;
}


int k2a_shim_launch_fill_pkmp(int dual, int mode, const K2aScoring *sc, const K2aPair *pairs, const uint32_t *order2, int ntasks,
                              const uint8_t *seq, uint8_t *tb, uint32_t *bnd, K2aResult *res, void *)
{
	typedef void (*fn)(const K2aScoring, const K2aPair*, const uint32_t*, int, const uint8_t*, uint8_t*, uint32_t*, K2aResult*);
	static const fn f[2][3] = { { sim_fill_pkmp<false, 0>, sim_fill_pkmp<false, 1>, sim_fill_pkmp<false, 2> },
	                            { sim_fill_pkmp<true, 0>, sim_fill_pkmp<true, 1>, sim_fill_pkmp<true, 2> } };
	if (ntasks > 0) f[dual ? 1 : 0][mode](*sc, pairs, order2, ntasks, seq, tb, bnd, res);
	return 0;
}
int k2a_shim_launch_ssec(int dual, int mode, size_t, const K2aSsec *par, const K2aPair *pairs, const uint32_t *order, int ntasks, const uint8_t *seq,
                         uint8_t *tb, uint8_t *scratch, K2aResult *res, void *)
{
	typedef void (*fn)(const K2aSsec, const K2aPair*, const uint32_t*, int, const uint8_t*, uint8_t*, uint8_t*, K2aResult*);
	static const fn f[2][3] = { { sim_ssec<false, 0>, sim_ssec<false, 1>, sim_ssec<false, 2> }, { sim_ssec<true, 0>, sim_ssec<true, 1>, sim_ssec<true, 2> } };
	f[dual ? 1 : 0][mode](*par, pairs, order, ntasks, seq, tb, scratch, res);
	return 0;
}
int k2a_shim_launch_ssec_blk(int dual, int mode, const K2aSsec *par, const K2aPair *pairs, const uint32_t *order, int ntasks, const uint8_t *seq, uint8_t *tb, K2aResult *res, void *)
{
	typedef void (*fn)(const K2aSsec, const K2aPair*, const uint32_t*, int, const uint8_t*, uint8_t*, K2aResult*);
	static const fn f[2][3] = { { sim_ssec_blk<false, 0>, sim_ssec_blk<false, 1>, sim_ssec_blk<false, 2> }, { sim_ssec_blk<true, 0>, sim_ssec_blk<true, 1>, sim_ssec_blk<true, 2> } };
	f[dual ? 1 : 0][mode](*par, pairs, order, ntasks, seq, tb, res);
	return 0;
}
int k2a_shim_launch_ssec_trace(const K2aPair *pairs, const uint32_t *order, int ntasks, const uint8_t *tb, K2aResult *res, uint32_t *cig, void *)
{
	for (int t = 0; t < ntasks; ++t) {
		const K2aPair pr = pairs[order[t]];
		K2aResult &r = res[order[t]];
		r.n_cigar = r.ti >= 0 && r.tj >= 0 ? k2a_ssec_trace(tb + pr.tb_off, k2a_ssec_ncol(pr.qlen, pr.tlen_full, pr.w), r.ti, r.tj, cig + pr.cig_off, pr.qlen, pr.tlen_full, pr.w) : 0;
	}
	return 0;
}
int k2a_shim_launch_fill_solo(int dual, int mode, const K2aScoring *sc, const K2aPair *pairs, const uint32_t *order, int ntasks,
                              const uint8_t *seq, uint8_t *tb, K2aResult *res, void *)
{
	typedef void (*fn_t)(const K2aScoring, const K2aPair*, const uint32_t*, int, const uint8_t*, uint8_t*, K2aResult*);
	static const fn_t fn[2][3] = { { sim_fill_solo<K2A_SOLO_CS, false, 0>, sim_fill_solo<K2A_SOLO_C, false, 1>, sim_fill_solo<K2A_SOLO_C, false, 2> },
	                               { sim_fill_solo<K2A_SOLO_CS, true, 0>, sim_fill_solo<K2A_SOLO_C, true, 1>, sim_fill_solo<K2A_SOLO_C, true, 2> } };
	if (ntasks > 0) fn[dual ? 1 : 0][mode](*sc, pairs, order, ntasks, seq, tb, res);
	return 0;
}
int k2a_shim_launch_trace_solo(const K2aPair *pairs, const uint32_t *order, int ntasks, const uint8_t *tb, K2aResult *res, uint32_t *cig, void *)
{
	if (ntasks > 0) sim_trace_solo<K2A_SOLO_C>(pairs, order, ntasks, tb, res, cig);
	return 0;
}

}
