/*
 * ksw2_lane_ssec.h -- per-position code of the SSE-COMPATIBLE mode: ksw_extz2_sse / ksw_extd2_sse exactly as the reference's
 * SSE kernels return them (ksw2_extz2_sse.c:101-301, ksw2_extd2_sse.c:131-398), for callers that must reproduce the output of
 * an SSE build bit for bit.  Opt-in (KSW2AMD_EZ_SSE_COMPAT / ksw2amd_set_sse_compat, include/ksw2_amd.h), and the path every
 * KSW_EZ_APPROX_MAX | KSW_EZ_APPROX_DROP request takes, because that heuristic is defined by this data flow.
 *
 * What differs from the exact contract of the other kernels is a property of the SSE data flow, so
 * the kernel keeps that data flow: one alignment per wavefront, lane <-> target position t, one step per anti-diagonal r;
 * the state is the reference's: wrapping 8-bit differences u, v, x, y (and x~, y~) and the score byte s per target position,
 * H of the in-band positions as int32 (exact mode).  Per anti-diagonal the in-band range [st0, en0] is widened to whole
 * 16-position blocks [st, en] and the padding positions are updated from whatever their bytes hold ("leaky band"); scores are
 * refreshed in runs of 16 from st0, reading past the sequences' ends like the reference does (a target code past the end is
 * 0, past the padded length it is the reversed query; a query code outside the query is 0: ksw2_extz2_sse.c:84-99,125-136).
 * The state arrays live in HBM scratch (L2-resident, a few bytes per target position); a position's left neighbour of the
 * previous anti-diagonal comes through a one-lane shift of the values the lane read for itself.
 *
 * Bookkeeping per anti-diagonal (k2a_ssec_book): maximum with the four-lane scan's tie order (k2a_dm_key, ksw2_lane_dm.h),
 * mte / mte_q = r - (padded en), mqe, ksw_apply_zdrop(.., is_rot = 1, ..) with the gap extension as slope, score; or the
 * one followed cell of the approximate modes (k2a_ssec_follow).
 */
#ifndef KSW2_LANE_SSEC_H_
#define KSW2_LANE_SSEC_H_

#include "ksw2_lane_dm.h"

/* K2aSsec (batch-uniform parameters) and the K2A_SSEC_* pair bits: ksw2_types.h */

K2A_FN int k2a_ssec_ncol(int qlen, int tlen, int w)     /* bytes per anti-diagonal of the direction matrix: n_col_ * 16 (ksw2_extz2_sse.c:74-76) */
{
	const int n = k2a_min(k2a_min(qlen, tlen), w + 1);
	return ((n + 15) / 16 + 1) * 16;
}

/* the in-band range [st0, en0] of anti-diagonal r and its 16-position blocks [st, en] (ksw2_extz2_sse.c:107-116); false: empty */
K2A_FN bool k2a_ssec_bounds(int r, int qlen, int tlen, int w, int &st0, int &en0, int &st, int &en)
{
	int a = 0, b = tlen - 1;
	if (a < r - qlen + 1) a = r - qlen + 1;
	if (b > r) b = r;
	if (a < (r - w + 1) >> 1) a = (r - w + 1) >> 1;
	if (b > (r + w) >> 1) b = (r + w) >> 1;
	st0 = a; en0 = b;
	st = a & ~15; en = ((b + 16) & ~15) - 1;               /* a, b >= 0 where the range is not empty */
	return a <= b;
}

/* the byte the reference's score loop reads as target / query code of position p on anti-diagonal r */
K2A_FN uint32_t k2a_ssec_tcode(const uint8_t *tgt, const uint8_t *qry, int tlen, int qlen, int T16, int p)
{
	if (p < tlen) return tgt[p];
	if (p < T16) return 0;
	return p - T16 < qlen ? qry[qlen - 1 - (p - T16)] : 0;            /* the reversed query follows the padded target (:86,:99) */
}
K2A_FN uint32_t k2a_ssec_qcode(const uint8_t *qry, int r, int p) { return r - p >= 0 ? qry[r - p] : 0; }
K2A_FN int k2a_ssec_score(const K2aSsec &P, bool generic, uint32_t a, uint32_t b)
{
	if (generic) return (int)P.mat[a * (uint32_t)P.m + b];
	return (a == (uint32_t)(P.m - 1) || b == (uint32_t)(P.m - 1)) ? P.sc_N : a == b ? P.sc_mch : P.sc_mis;
}

K2A_FN int k2a_s8(int v) { return (int)(int8_t)v; }
K2A_FN int k2a_u8(int v) { return (int)(uint8_t)v; }

/* extd2: what the first-column cell and the block's left edge read (ksw2_extd2_sse.c:158-163) */
K2A_FN int k2a_ssec_edge(const K2aSsec &P, int r)
{
	return r == 0 ? -P.q - P.e : r < P.long_thres ? -P.e : r == P.long_thres ? P.long_diff : -P.e2;
}

/* One position.  All arguments are the bytes of the previous anti-diagonal as signed values: s (score), xt1 / vt1 / x2t1 of
 * position t - 1, ut / yt / y2t of position t.  Out: the new bytes (low 8 bits significant) and the direction byte.
 * Single gap: ksw2_extz2_sse.c:27-47,146-222 (differences shifted by q + e, unsigned byte max / min in block 2);
 * two-piece: ksw2_extd2_sse.c:38-66,189-321 (signed throughout).  SSE4.1 path. */
template<bool DUAL, int MODE>
K2A_FN void k2a_ssec_cell(const K2aSsec &P, int s, int xt1, int vt1, int x2t1, int ut, int yt, int y2t,
                          int &un, int &vn, int &xn, int &yn, int &x2n, int &y2n, uint32_t &dir)
{
	uint32_t d = 0;
	if (!DUAL) {
		const int qe = P.q + P.e;
		int z = k2a_s8(s + 2 * qe), a = k2a_s8(xt1 + vt1), b = k2a_s8(yt + ut);
		if (MODE == K2A_MODE_LEFT) d = a > z ? 1u : 0u;
		if (MODE == K2A_MODE_RIGHT) d = z > a ? 0u : 1u;
		z = z > a ? z : a;
		if (MODE == K2A_MODE_LEFT) d = b > z ? 2u : d;
		if (MODE == K2A_MODE_RIGHT) d = z > b ? d : 2u;
		int zu = k2a_u8(z) > k2a_u8(b) ? k2a_u8(z) : k2a_u8(b);
		const int cap = k2a_u8(P.sc_mch + 2 * qe);
		if (zu > cap) zu = cap;
		un = zu - vt1; vn = zu - ut;
		z = k2a_s8(zu - P.q); a = k2a_s8(a - z); b = k2a_s8(b - z);
		if (MODE != K2A_MODE_RIGHT) {
			xn = a > 0 ? a : 0; yn = b > 0 ? b : 0;
			if (MODE == K2A_MODE_LEFT) d |= (a > 0 ? 0x08u : 0u) | (b > 0 ? 0x10u : 0u);
		} else {
			xn = 0 > a ? 0 : a; yn = 0 > b ? 0 : b;
			d |= (0 > a ? 0u : 0x08u) | (0 > b ? 0u : 0x10u);
		}
		x2n = y2n = 0;
	} else {
		const int qe = P.q + P.e, qe2 = P.q2 + P.e2;
		int z = k2a_s8(s), a = k2a_s8(xt1 + vt1), b = k2a_s8(yt + ut), a2 = k2a_s8(x2t1 + vt1), b2 = k2a_s8(y2t + ut);
		if (MODE == K2A_MODE_SCORE) {
			z = k2a_max(k2a_max(z, a), k2a_max(b, k2a_max(a2, b2)));
		} else if (MODE == K2A_MODE_LEFT) {
			d = a > z ? 1u : 0u;  z = k2a_max(z, a);
			d = b > z ? 2u : d;   z = k2a_max(z, b);
			d = a2 > z ? 3u : d;  z = k2a_max(z, a2);
			d = b2 > z ? 4u : d;  z = k2a_max(z, b2);
		} else {
			d = z > a ? 0u : 1u;  z = k2a_max(z, a);
			d = z > b ? d : 2u;   z = k2a_max(z, b);
			d = z > a2 ? d : 3u;  z = k2a_max(z, a2);
			d = z > b2 ? d : 4u;  z = k2a_max(z, b2);
		}
		z = k2a_min(z, P.sc_mch);
		un = z - vt1; vn = z - ut;
		int tmp = k2a_s8(z - P.q);
		a = k2a_s8(a - tmp); b = k2a_s8(b - tmp);
		tmp = k2a_s8(z - P.q2);
		a2 = k2a_s8(a2 - tmp); b2 = k2a_s8(b2 - tmp);
		if (MODE != K2A_MODE_RIGHT) {
			xn = (a > 0 ? a : 0) - qe; yn = (b > 0 ? b : 0) - qe; x2n = (a2 > 0 ? a2 : 0) - qe2; y2n = (b2 > 0 ? b2 : 0) - qe2;
			if (MODE == K2A_MODE_LEFT) d |= (a > 0 ? 0x08u : 0u) | (b > 0 ? 0x10u : 0u) | (a2 > 0 ? 0x20u : 0u) | (b2 > 0 ? 0x40u : 0u);
		} else {
			xn = (0 > a ? 0 : a) - qe; yn = (0 > b ? 0 : b) - qe; x2n = (0 > a2 ? 0 : a2) - qe2; y2n = (0 > b2 ? 0 : b2) - qe2;
			d |= (0 > a ? 0u : 0x08u) | (0 > b ? 0u : 0x10u) | (0 > a2 ? 0u : 0x20u) | (0 > b2 ? 0u : 0x40u);
		}
	}
	dir = d;
}

/* what a difference byte adds to H: the single-gap kernel reads its bytes unsigned and takes q + e off (ksw2_extz2_sse.c:229,
 * 238-239), the two-piece kernel reads them signed (ksw2_extd2_sse.c:328,335) */
template<bool DUAL>
K2A_FN int k2a_ssec_dh(const K2aSsec &P, int byte) { return DUAL ? k2a_s8(byte) : k2a_u8(byte) - (P.q + P.e); }

/* ksw_apply_zdrop with is_rot = 1 (ksw2.h:191-207) on the book; returns 1 on a drop.  Written as selects: from an if / else that
 * stores to different fields hipcc makes ONE store through a selected address, and the book leaves its registers for scratch memory. */
K2A_FN int k2a_ssec_zdrop(K2aBook *b, int H, int r, int t, int zdrop, int slope)
{
	const bool up = H > b->max;
	const int tl = t - b->max_t, ql = (r - t) - b->max_q, l = tl > ql ? tl - ql : ql - tl;
	const int drop = (!up && tl >= 0 && ql >= 0 && zdrop >= 0 && b->max - H > zdrop + l * slope) ? 1 : 0;
	b->max_t = up ? t : b->max_t; b->max_q = up ? r - t : b->max_q; b->max = up ? H : b->max;
	b->dropped |= drop;
	return drop;
}

/* exact mode, uniform values of one anti-diagonal (ksw2_extz2_sse.c:229-269): A = H at en0, Bkey = winner of the four-lane
 * region [st0, en1) (0 = empty), T0..T2 = H at en1 + 0..2, S = H at st0.  Returns 1 on a Z-drop. */
K2A_FN int k2a_ssec_book(K2aBook *b, int r, int st0, int en0, int en, int qlen, int tlen, int zdrop, int slope,
                         int A, uint64_t Bkey, int T0, int T1, int T2, int S)
{
	const int en1 = st0 + (int)((uint32_t)(en0 - st0) & ~3u);
	int max_H = A, max_t = en0;
	if (Bkey != 0 && k2a_dm_key_H(Bkey) > max_H) { max_H = k2a_dm_key_H(Bkey); max_t = k2a_dm_key_t(Bkey); }
	if (en1 < en0 && T0 > max_H) { max_H = T0; max_t = en1; }
	if (en1 + 1 < en0 && T1 > max_H) { max_H = T1; max_t = en1 + 1; }
	if (en1 + 2 < en0 && T2 > max_H) { max_H = T2; max_t = en1 + 2; }
	if (en0 == tlen - 1 && A > b->mte) { b->mte = A; b->mte_q = r - en; }
	if (r - st0 == qlen - 1 && S > b->mqe) { b->mqe = S; b->mqe_t = st0; }
	if (k2a_ssec_zdrop(b, max_H, r, max_t, zdrop, slope)) return 1;
	if (r == qlen + tlen - 2 && en0 == tlen - 1) b->score = A;
	b->rows = r + 1;
	return 0;
}

/* approximate modes: the one followed cell (ksw2_extz2_sse.c:270-286, ksw2_extd2_sse.c:366-382).  vl = v byte at `last`,
 * un = u byte at last + 1, v0 = v byte at 0 (r == 0).  The two kernels differ in where the drop test sits at r = 0. */
typedef struct K2aSsecFollow { int H0, last; } K2aSsecFollow;
template<bool DUAL>
K2A_FN int k2a_ssec_follow(const K2aSsec &P, K2aSsecFollow &f, K2aBook *b, int r, int st0, int en0, int qlen, int tlen, int zdrop, bool drop,
                           int vl, int un, int v0)
{
	const int slope = DUAL ? P.e2 : P.e;
	if (r > 0) {
		const bool in0 = f.last >= st0 && f.last <= en0, in1 = f.last + 1 >= st0 && f.last + 1 <= en0;
		if (in0 && in1) {
			const int d0 = k2a_ssec_dh<DUAL>(P, vl), d1 = k2a_ssec_dh<DUAL>(P, un);
			if (d0 > d1) f.H0 += d0; else { f.H0 += d1; ++f.last; }
		} else if (in0) f.H0 += k2a_ssec_dh<DUAL>(P, vl);
		else { ++f.last; f.H0 += k2a_ssec_dh<DUAL>(P, un); }
		if (!DUAL && drop && k2a_ssec_zdrop(b, f.H0, r, f.last, zdrop, slope)) return 1;
	} else { f.H0 = k2a_ssec_dh<DUAL>(P, v0) - (DUAL ? P.qe_first : P.q + P.e); f.last = 0; }
	if (DUAL && drop && k2a_ssec_zdrop(b, f.H0, r, f.last, zdrop, slope)) return 1;
	if (r == qlen + tlen - 2 && en0 == tlen - 1) b->score = f.H0;
	b->rows = r + 1;
	return 0;
}

/* ksw_backtrack with is_rot = 1 (ksw2.h:129-161) on tb[r * ncol + t - st(r)]: a position outside the stored blocks of its
 * anti-diagonal forces an insertion (below) or a deletion (above).  CIGAR in walk order (end -> start); returns the op count. */
K2A_FN int k2a_ssec_trace(const uint8_t *tb, int ncol, int i, int j, uint32_t *out, int qlen, int tlen, int w)
{
	int n = 0, state = 0;
	uint32_t last_op = 0xffffffffu, run = 0;
	while (i >= 0 && j >= 0) {
		const int r = i + j;
		int st0, en0, st, en, force = -1;
		k2a_ssec_bounds(r, qlen, tlen, w, st0, en0, st, en);
		if (i < st) force = 2;
		if (i > en) force = 1;
		const uint32_t d = force < 0 ? tb[(size_t)r * ncol + (i - st)] : 0;
		if (state == 0) state = d & 7;
		else if (!((d >> (state + 2)) & 1)) state = 0;
		if (state == 0) state = d & 7;
		if (force >= 0) state = force;
		uint32_t op;
		if (state == 0) { op = 0; --i; --j; }
		else if (state == 1 || state == 3) { op = 2; --i; }
		else { op = 1; --j; }
		if (op == last_op) ++run;
		else { if (run) out[n++] = run << 4 | last_op; last_op = op; run = 1; }
	}
	if (i >= 0) {
		if (last_op == 2) run += i + 1;
		else { if (run) out[n++] = run << 4 | last_op; last_op = 2; run = i + 1; }
	}
	if (j >= 0) {
		if (last_op == 1) run += j + 1;
		else { if (run) out[n++] = run << 4 | last_op; last_op = 1; run = j + 1; }
	}
	if (run) out[n++] = run << 4 | last_op;
	return n;
}

#endif
