/*
 * ksw2_oracle_sse.c -- CPU restatement of what ksw_extz2_sse / ksw_extd2_sse THEMSELVES return: the "leaky band" of
 * their 16-lane blocks, the anti-diagonal Z-drop, the padded mte_q, and the KSW_EZ_APPROX_MAX / APPROX_DROP heuristics.
 *
 * TEST INFRASTRUCTURE ONLY (see ksw2_oracle.h).  Parity status: PINNED -- tests/test_oracle_sse.py checks every field and
 * CIGAR against tests/golden/sse_cases.npz, produced by the unmodified reference (oracle/gen_golden_sse.py), and live
 * against oracle/_ref where that artefact exists.
 *
 * Unlike kso_extz2 / kso_extd2 (the exact-band, row-wise contract of the scalar functions), these follow the SSE kernels'
 * memory image position by position, because that image IS their definition:
 *   - per anti-diagonal r the in-band range [st0, en0] is widened to whole 16-position blocks [st, en]
 *     (ksw2_extz2_sse.c:116, ksw2_extd2_sse.c:147); the extra positions are updated from whatever their bytes hold and
 *     are read by in-band cells on later anti-diagonals;
 *   - u, v, x, y (and x~, y~) are wrapping 8-bit differences; the single-gap kernel keeps them shifted by q + e and
 *     combines them with UNSIGNED byte maxima / minima (ksw2_extz2_sse.c:40-47), the two-piece kernel keeps them signed
 *     (ksw2_extd2_sse.c:38-66);
 *   - scores are refreshed in runs of 16 starting at st0 (:123-136 / :166-179), so positions past en0 get the score of
 *     codes read beyond the sequences' ends: the arrays are laid out in ONE zero-initialised allocation in the
 *     reference's order (u v x y [x~ y~] s target query-reversed, :84-86 / :107-110) and indexed exactly like it, which
 *     reproduces every such read and the few score bytes that spill into the target copy;
 *   - H of the exact mode lives in an int32 array over target positions and is advanced by v (:224-261 / :323-360), the
 *     maximum of an anti-diagonal is found by four interleaved scans whose tie order is reproduced (`diag_max`);
 *   - bookkeeping per anti-diagonal: mte / mte_q = r - en with the PADDED en (:263-264), mqe, ksw_apply_zdrop(.., is_rot = 1)
 *     (ksw2.h:191-207), score; approximate modes follow one cell (:270-286 / :366-382).
 * SSE4.1 code path (the build of oracle/_ref); the SSE2 path gives identical results.
 */
#include <stdlib.h>
#include <string.h>
#include "ksw2_oracle.h"

#define NEG KSO_NEG_INF

typedef struct { uint32_t *a; int n, cap; } scig_t;

static void scig_add(scig_t *c, uint32_t op, int len)                     /* ksw2.h:113-123 */
{
	if (c->n > 0 && (c->a[c->n - 1] & 0xfu) == op) { c->a[c->n - 1] += (uint32_t)len << 4; return; }
	if (c->n == c->cap) { c->cap = c->cap ? c->cap * 2 : 4; c->a = (uint32_t*)realloc(c->a, sizeof(uint32_t) * (size_t)c->cap); }
	c->a[c->n++] = (uint32_t)len << 4 | op;
}

static void sse_reset(kso_extz_t *ez)                                       /* ksw2.h:184-189 */
{
	ez->max = 0; ez->zdropped = 0;
	ez->max_q = ez->max_t = ez->mqe_t = ez->mte_q = -1;
	ez->mqe = ez->mte = ez->score = NEG;
	ez->n_cigar = 0; ez->reach_end = 0;
}

/* ksw_apply_zdrop with is_rot = 1 (ksw2.h:191-207): (r, t) = anti-diagonal and target position of the cell */
static int zdrop_diag(kso_extz_t *ez, int32_t H, int r, int t, int zdrop, int e)
{
	if (H > (int32_t)ez->max) { ez->max = (uint32_t)H; ez->max_t = t; ez->max_q = r - t; }
	else if (t >= ez->max_t && r - t >= ez->max_q) {
		const int tl = t - ez->max_t, ql = (r - t) - ez->max_q, l = tl > ql ? tl - ql : ql - tl;
		if (zdrop >= 0 && (int32_t)ez->max - H > zdrop + l * e) { ez->zdropped = 1; return 1; }
	}
	return 0;
}

/* ksw_backtrack with is_rot = 1 (ksw2.h:129-161): direction bytes at p[r * ncol + t - off[r]]; a target position outside
 * the stored block of its anti-diagonal forces an insertion (below it) or a deletion (above it) */
static void backtrack_rot(int rev, const uint8_t *p, const int *off, const int *off_end, size_t ncol, int i0, int j0, kso_extz_t *ez)
{
	scig_t c = { ez->cigar, 0, ez->m_cigar };
	int i = i0, j = j0, state = 0, k;
	while (i >= 0 && j >= 0) {
		const int r = i + j;
		int force = -1;
		uint32_t d;
		if (i < off[r]) force = 2;
		if (i > off_end[r]) force = 1;
		d = force < 0 ? p[(size_t)r * ncol + (size_t)(i - off[r])] : 0;
		if (state == 0) state = d & 7;
		else if (!(d >> (state + 2) & 1)) state = 0;
		if (state == 0) state = d & 7;
		if (force >= 0) state = force;
		if (state == 0) { scig_add(&c, 0, 1); --i; --j; }
		else if (state == 1 || state == 3) { scig_add(&c, 2, 1); --i; }
		else { scig_add(&c, 1, 1); --j; }
	}
	if (i >= 0) scig_add(&c, 2, i + 1);
	if (j >= 0) scig_add(&c, 1, j + 1);
	if (!rev) for (k = 0; k < c.n >> 1; ++k) { const uint32_t t = c.a[k]; c.a[k] = c.a[c.n - 1 - k]; c.a[c.n - 1 - k] = t; }
	ez->cigar = c.a; ez->n_cigar = c.n; ez->m_cigar = c.cap;
}

/* M -> = / X (ksw2.h:163-182, done on a private copy: the reference drops a realloc result there) */
static void sse_eqx(const uint8_t *query, const uint8_t *target, kso_extz_t *ez)
{
	scig_t c = { 0, 0, 0 };
	int k, i, x = 0, y = 0;
	for (k = 0; k < ez->n_cigar; ++k) {
		const int op = ez->cigar[k] & 0xf, len = (int)(ez->cigar[k] >> 4);
		if (op == 0) { for (i = 0; i < len; ++i) scig_add(&c, target[x + i] == query[y + i] ? 7 : 8, 1); x += len; y += len; }
		else { scig_add(&c, (uint32_t)op, len); if (op == 2 || op == 3) x += len; else if (op == 1) y += len; }
	}
	free(ez->cigar);
	ez->cigar = c.a; ez->n_cigar = c.n; ez->m_cigar = c.cap;
}

/* maximum of H over [st0, en0] of one anti-diagonal with the SSE code's tie order (ksw2_extz2_sse.c:229-259): H[en0] seeds the
 * maximum, four scans over t = st0 + 4k + lane take strictly larger values and remember the run's start, the lanes are
 * merged in order with strict comparisons, then the up-to-three positions before en0.  `dv` = what is added to H[t]. */
static void diag_max(int32_t *H, int st0, int en0, const int32_t *dv, int32_t *max_H, int *max_t)
{
	const int en1 = st0 + (en0 - st0) / 4 * 4;
	int32_t HH[4], mh = H[en0];
	int tt[4], mt = en0, t, i;
	for (i = 0; i < 4; ++i) { HH[i] = mh; tt[i] = mt; }
	for (t = st0; t < en1; t += 4)
		for (i = 0; i < 4; ++i) { H[t + i] += dv[t + i - st0]; if (H[t + i] > HH[i]) { HH[i] = H[t + i]; tt[i] = t; } }
	for (i = 0; i < 4; ++i) if (mh < HH[i]) { mh = HH[i]; mt = tt[i] + i; }
	for (; t < en0; ++t) { H[t] += dv[t - st0]; if (H[t] > mh) { mh = H[t]; mt = t; } }
	*max_H = mh; *max_t = mt;
}

static void sse_core(int dual, int qlen, const uint8_t *query, int tlen, const uint8_t *target, int8_t m, const int8_t *mat,
                     int q, int e, int q2, int e2, int w, int zdrop, int end_bonus, int flag, kso_extz_t *ez)
{
	const int with_cigar = !(flag & KSO_SCORE_ONLY), approx = !!(flag & KSO_APPROX_MAX), right = !!(flag & KSO_RIGHT);
	int r, t, tlen_, qlen_, n_col_, T16, last_st = -1, last_en = -1, min_sc, long_thres = 0, long_diff = 0, qe, qe2, narr, zslope;
	const int qe_first = q + e;       /* extd2 :67 sets its scalar `qe` BEFORE the pieces are swapped (:78) and uses it for H of the first cell only (:353,:377) */
	int32_t *H = 0, *dv = 0, H0 = 0;
	int last_H0_t = 0, *off = 0, *off_end = 0;
	uint8_t *mem, *u, *v, *x, *y, *x2 = 0, *y2 = 0, *s, *sf, *qr, *p = 0;
	int8_t sc_mch, sc_mis, sc_N;

	sse_reset(ez);
	if (m <= (dual ? 1 : 0) || qlen <= 0 || tlen <= 0) return;                                   /* extz2 :57, extd2 :76 */
	if (dual && q2 + e2 < q + e) { t = q; q = q2; q2 = t; t = e; e = e2; e2 = t; }                /* extd2 :78 */
	qe = q + e; qe2 = q2 + e2; zslope = dual ? e2 : e;
	sc_mch = mat[0]; sc_mis = mat[1];
	sc_N = mat[m * m - 1] == 0 ? (int8_t)-(dual ? e2 : e) : mat[m * m - 1];                        /* extz2 :68, extd2 :87 */
	if (w < 0) w = tlen > qlen ? tlen : qlen;
	tlen_ = (tlen + 15) / 16; qlen_ = (qlen + 15) / 16; T16 = tlen_ * 16;
	n_col_ = qlen < tlen ? qlen : tlen;
	n_col_ = ((n_col_ < w + 1 ? n_col_ : w + 1) + 15) / 16 + 1;
	for (t = 1, min_sc = mat[1]; t < m * m; ++t) min_sc = min_sc < mat[t] ? min_sc : mat[t];
	if (-min_sc > 2 * (q + e)) return;                                                           /* :82 / :100 */
	if (dual) {                                                                                  /* extd2 :102-105 */
		long_thres = e != e2 ? (q2 - q) / (e - e2) - 1 : 0;
		if (q2 + e2 + long_thres * e2 > q + e + long_thres * e) ++long_thres;
		long_diff = long_thres * (e - e2) - (q2 - q) - e2;
	}
	/* one allocation in the reference's order (extz2 :84-86: u v x y s target query; extd2 :107-110: u v x y x~ y~ s target query) */
	narr = dual ? 8 : 6;
	mem = (uint8_t*)calloc((size_t)tlen_ * narr + qlen_ + 1, 16);
	u = mem; v = u + T16; x = v + T16; y = x + T16;
	if (dual) { x2 = y + T16; y2 = x2 + T16; s = y2 + T16; } else s = y + T16;
	sf = s + T16; qr = sf + T16;
	if (dual) {                                                                                  /* extd2 :111-116 */
		memset(u, -q - e, (size_t)T16); memset(v, -q - e, (size_t)T16); memset(x, -q - e, (size_t)T16); memset(y, -q - e, (size_t)T16);
		memset(x2, -q2 - e2, (size_t)T16); memset(y2, -q2 - e2, (size_t)T16);
	}
	if (!approx) {
		H = (int32_t*)malloc(sizeof(int32_t) * (size_t)T16);
		dv = (int32_t*)malloc(sizeof(int32_t) * (size_t)T16);
		for (t = 0; t < T16; ++t) H[t] = NEG;
	}
	if (with_cigar) {
		p = (uint8_t*)malloc(((size_t)(qlen + tlen - 1) * n_col_ + 1) * 16);
		off = (int*)malloc(sizeof(int) * 2 * (size_t)(qlen + tlen - 1));
		off_end = off + qlen + tlen - 1;
	}
	for (t = 0; t < qlen; ++t) qr[t] = query[qlen - 1 - t];
	memcpy(sf, target, (size_t)tlen);

	for (r = 0; r < qlen + tlen - 1; ++r) {
		int st = 0, en = tlen - 1, st0, en0;
		int8_t x1, x21 = 0, v1;
		const uint8_t *qrr = qr + (qlen - 1 - r);
		uint8_t *pr = 0;
		if (st < r - qlen + 1) st = r - qlen + 1;
		if (en > r) en = r;
		if (st < (r - w + 1) >> 1) st = (r - w + 1) >> 1;
		if (en > (r + w) >> 1) en = (r + w) >> 1;
		if (st > en) { ez->zdropped = 1; break; }                                                /* :111-114 / :142-145 */
		st0 = st; en0 = en;
		st = st / 16 * 16; en = (en + 16) / 16 * 16 - 1;
		/* what the block's first position sees to its left, and the first-column cell (extz2 :118-124, extd2 :149-164) */
		if (!dual) {
			if (st > 0) { if (st - 1 >= last_st && st - 1 <= last_en) { x1 = (int8_t)x[st - 1]; v1 = (int8_t)v[st - 1]; } else x1 = v1 = 0; }
			else { x1 = 0; v1 = (int8_t)(r ? q : 0); }
			if (en >= r) { y[r] = 0; u[r] = (uint8_t)(r ? q : 0); }
		} else {
			const int8_t edge = (int8_t)(r == 0 ? -q - e : r < long_thres ? -e : r == long_thres ? long_diff : -e2);
			if (st > 0) {
				if (st - 1 >= last_st && st - 1 <= last_en) { x1 = (int8_t)x[st - 1]; x21 = (int8_t)x2[st - 1]; v1 = (int8_t)v[st - 1]; }
				else { x1 = (int8_t)(-q - e); x21 = (int8_t)(-q2 - e2); v1 = (int8_t)(-q - e); }
			} else { x1 = (int8_t)(-q - e); x21 = (int8_t)(-q2 - e2); v1 = edge; }
			if (en >= r) { y[r] = (uint8_t)(-q - e); y2[r] = (uint8_t)(-q2 - e2); u[r] = (uint8_t)edge; }
		}
		/* scores, in runs of 16 from st0 (:125-140 / :166-182) */
		if (!(flag & KSO_GENERIC_SC)) {
			for (t = st0; t <= en0; t += 16) {
				int k;
				uint8_t run[16];
				for (k = 0; k < 16; ++k) {
					const uint8_t a = sf[t + k], b = qrr[t + k];
					run[k] = (uint8_t)((a == (uint8_t)(m - 1) || b == (uint8_t)(m - 1)) ? sc_N : a == b ? sc_mch : sc_mis);
				}
				memcpy(s + t, run, 16);                           /* all 16 loaded before any is stored */
			}
		} else for (t = st0; t <= en0; ++t) s[t] = (uint8_t)mat[sf[t] * m + qrr[t]];
		if (with_cigar) { pr = p + (size_t)r * n_col_ * 16; off[r] = st; off_end[r] = en; }
		/* the cell update over the whole blocks; position t reads x, v of position t - 1 as they were before this anti-diagonal */
		for (t = st; t <= en; ++t) {
			const int8_t xt1 = x1, vt1 = v1, ut = (int8_t)u[t];
			uint8_t d = 0;
			x1 = (int8_t)x[t]; v1 = (int8_t)v[t];
			if (!dual) {                                                                         /* ksw2_extz2_sse.c:27-47,146-222 */
				int8_t z = (int8_t)(s[t] + 2 * qe), a = (int8_t)(xt1 + vt1), b = (int8_t)(y[t] + ut);
				uint8_t zu;
				if (with_cigar && !right) d = a > z ? 1 : 0;
				if (with_cigar && right) d = z > a ? 0 : 1;
				z = z > a ? z : a;                                                               /* signed */
				if (with_cigar && !right) d = b > z ? 2 : d;
				if (with_cigar && right) d = z > b ? d : 2;
				zu = (uint8_t)z > (uint8_t)b ? (uint8_t)z : (uint8_t)b;                          /* unsigned */
				if (zu > (uint8_t)(int8_t)(sc_mch + 2 * qe)) zu = (uint8_t)(int8_t)(sc_mch + 2 * qe);
				u[t] = (uint8_t)(zu - (uint8_t)vt1); v[t] = (uint8_t)(zu - (uint8_t)ut);
				z = (int8_t)(zu - q); a = (int8_t)(a - z); b = (int8_t)(b - z);
				if (!with_cigar || !right) {
					x[t] = (uint8_t)(a > 0 ? a : 0); y[t] = (uint8_t)(b > 0 ? b : 0);
					if (with_cigar) d |= (a > 0 ? 0x08 : 0) | (b > 0 ? 0x10 : 0);
				} else {
					x[t] = (uint8_t)(0 > a ? 0 : a); y[t] = (uint8_t)(0 > b ? 0 : b);
					d |= (0 > a ? 0 : 0x08) | (0 > b ? 0 : 0x10);
				}
			} else {                                                                             /* ksw2_extd2_sse.c:38-66,189-321 */
				const int8_t x2t1 = x21;
				int8_t z = (int8_t)s[t], a = (int8_t)(xt1 + vt1), b = (int8_t)(y[t] + ut), a2 = (int8_t)(x2t1 + vt1), b2 = (int8_t)(y2[t] + ut), tmp;
				x21 = (int8_t)x2[t];
				if (!with_cigar) {
					z = z > a ? z : a; z = z > b ? z : b; z = z > a2 ? z : a2; z = z > b2 ? z : b2;
				} else if (!right) {
					d = a > z ? 1 : 0;   z = z > a ? z : a;
					d = b > z ? 2 : d;   z = z > b ? z : b;
					d = a2 > z ? 3 : d;  z = z > a2 ? z : a2;
					d = b2 > z ? 4 : d;  z = z > b2 ? z : b2;
				} else {
					d = z > a ? 0 : 1;   z = z > a ? z : a;
					d = z > b ? d : 2;   z = z > b ? z : b;
					d = z > a2 ? d : 3;  z = z > a2 ? z : a2;
					d = z > b2 ? d : 4;  z = z > b2 ? z : b2;
				}
				z = z < sc_mch ? z : sc_mch;
				u[t] = (uint8_t)(z - vt1); v[t] = (uint8_t)(z - ut);
				tmp = (int8_t)(z - q);  a = (int8_t)(a - tmp);  b = (int8_t)(b - tmp);
				tmp = (int8_t)(z - q2); a2 = (int8_t)(a2 - tmp); b2 = (int8_t)(b2 - tmp);
				if (!with_cigar || !right) {
					x[t] = (uint8_t)((a > 0 ? a : 0) - qe);    y[t] = (uint8_t)((b > 0 ? b : 0) - qe);
					x2[t] = (uint8_t)((a2 > 0 ? a2 : 0) - qe2); y2[t] = (uint8_t)((b2 > 0 ? b2 : 0) - qe2);
					if (with_cigar) d |= (a > 0 ? 0x08 : 0) | (b > 0 ? 0x10 : 0) | (a2 > 0 ? 0x20 : 0) | (b2 > 0 ? 0x40 : 0);
				} else {
					x[t] = (uint8_t)((0 > a ? 0 : a) - qe);    y[t] = (uint8_t)((0 > b ? 0 : b) - qe);
					x2[t] = (uint8_t)((0 > a2 ? 0 : a2) - qe2); y2[t] = (uint8_t)((0 > b2 ? 0 : b2) - qe2);
					d |= (0 > a ? 0 : 0x08) | (0 > b ? 0 : 0x10) | (0 > a2 ? 0 : 0x20) | (0 > b2 ? 0 : 0x40);
				}
			}
			if (with_cigar) pr[t - st] = d;
		}
		/* H of the in-band positions, the anti-diagonal's best cell, bookkeeping (extz2 :224-269 unsigned bytes minus q + e;
		 * extd2 :323-365 signed bytes) */
#define UD(arr, i) (dual ? (int32_t)(int8_t)(arr)[i] : (int32_t)(arr)[i] - qe)
		if (!approx) {
			int32_t max_H;
			int max_t;
			if (r > 0) {
				H[en0] = en0 > 0 ? H[en0 - 1] + UD(u, en0) : H[en0] + UD(v, en0);
				for (t = st0; t < en0; ++t) dv[t - st0] = UD(v, t);
				diag_max(H, st0, en0, dv, &max_H, &max_t);
			} else { H[0] = UD(v, 0) - (dual ? qe_first : qe); max_H = H[0]; max_t = 0; }
			if (en0 == tlen - 1 && H[en0] > ez->mte) { ez->mte = H[en0]; ez->mte_q = r - en; }
			if (r - st0 == qlen - 1 && H[st0] > ez->mqe) { ez->mqe = H[st0]; ez->mqe_t = st0; }
			if (zdrop_diag(ez, max_H, r, max_t, zdrop, zslope)) break;
			if (r == qlen + tlen - 2 && en0 == tlen - 1) ez->score = H[tlen - 1];
		} else {
			if (r > 0) {
				if (last_H0_t >= st0 && last_H0_t <= en0 && last_H0_t + 1 >= st0 && last_H0_t + 1 <= en0) {
					const int32_t d0 = UD(v, last_H0_t), d1 = UD(u, last_H0_t + 1);
					if (d0 > d1) H0 += d0; else { H0 += d1; ++last_H0_t; }
				} else if (last_H0_t >= st0 && last_H0_t <= en0) H0 += UD(v, last_H0_t);
				else { ++last_H0_t; H0 += UD(u, last_H0_t); }
				if (!dual && (flag & KSO_APPROX_DROP) && zdrop_diag(ez, H0, r, last_H0_t, zdrop, zslope)) break;      /* extz2 :281: inside r > 0 */
			} else { H0 = UD(v, 0) - (dual ? qe_first : qe); last_H0_t = 0; }
			if (dual && (flag & KSO_APPROX_DROP) && zdrop_diag(ez, H0, r, last_H0_t, zdrop, zslope)) break;          /* extd2 :379: every r */
			if (r == qlen + tlen - 2 && en0 == tlen - 1) ez->score = H0;
		}
#undef UD
		last_st = st; last_en = en;
	}
	free(mem); free(H); free(dv);
	if (with_cigar) {
		const int rev = !!(flag & KSO_REV_CIGAR);
		const size_t ncol = (size_t)n_col_ * 16;
		if (!ez->zdropped && !(flag & KSO_EXTZ_ONLY)) backtrack_rot(rev, p, off, off_end, ncol, tlen - 1, qlen - 1, ez);
		else if (!ez->zdropped && (flag & KSO_EXTZ_ONLY) && ez->mqe + end_bonus > (int)ez->max) {
			ez->reach_end = 1;
			backtrack_rot(rev, p, off, off_end, ncol, ez->mqe_t, qlen - 1, ez);
		} else if (ez->max_t >= 0 && ez->max_q >= 0) backtrack_rot(rev, p, off, off_end, ncol, ez->max_t, ez->max_q, ez);
		if (dual && (flag & KSO_EQX)) sse_eqx(query, target, ez);                               /* extd2 :399-406 */
		free(p); free(off);
	}
}

void kso_extz2_sse(int qlen, const uint8_t *query, int tlen, const uint8_t *target, int8_t m, const int8_t *mat,
                   int8_t q, int8_t e, int w, int zdrop, int end_bonus, int flag, kso_extz_t *ez)
{
	sse_core(0, qlen, query, tlen, target, m, mat, q, e, 0, 0, w, zdrop, end_bonus, flag, ez);
}

void kso_extd2_sse(int qlen, const uint8_t *query, int tlen, const uint8_t *target, int8_t m, const int8_t *mat,
                   int8_t q, int8_t e, int8_t q2, int8_t e2, int w, int zdrop, int end_bonus, int flag, kso_extz_t *ez)
{
	sse_core(1, qlen, query, tlen, target, m, mat, q, e, q2, e2, w, zdrop, end_bonus, flag, ez);
}
