/*
 * ksw2_oracle.c -- CPU restatement of the ksw2 banded extension / global alignment results.
 *
 * TEST INFRASTRUCTURE ONLY (see ksw2_oracle.h for the rules and for the parity-pinning status).
 *
 * This is a plain int32 row-by-row evaluation of the Gotoh/Green recurrences written from the
 * mathematical statement in SURVEY.md Appendix A.1 and from the *behaviour* of the reference's scalar
 * functions (cited per block below).  One core routine serves the single- and the two-piece gap
 * models; thin wrappers reproduce the two calling contracts (scalar `ksw_extz/extd/gg`, and the
 * `...2_sse` signatures with end_bonus / reach_end / implicit wildcard scoring).
 *
 *      H(i,j)   = max{ H(i-1,j-1) + S(i,j), E(i,j), F(i,j) [, E~(i,j), F~(i,j)] }
 *      E(i+1,j) = max{ H(i,j) - q, E(i,j) } - e          (gap consuming target: "D")
 *      F(i,j+1) = max{ H(i,j) - q, F(i,j) } - e          (gap consuming query:  "I")
 *      band: cell (i,j) exists iff |i-j| <= w; everything outside is -infinity
 */
#include <stdlib.h>
#include <string.h>
#include "ksw2_oracle.h"

#define NEG KSO_NEG_INF

/* ---------------------------------------------------------------- small helpers */

static inline int imin(int a, int b) { return a < b ? a : b; }
static inline int imax(int a, int b) { return a > b ? a : b; }

/* growable CIGAR; same run-length encoding and doubling-from-4 growth as ksw2.h:113-123 */
typedef struct { uint32_t *a; int n, cap; } cig_t;

static void cig_add(cig_t *c, uint32_t op, int len)
{
	if (c->n > 0 && (c->a[c->n - 1] & 0xfu) == op) { c->a[c->n - 1] += (uint32_t)len << 4; return; }
	if (c->n == c->cap) {
		c->cap = c->cap ? c->cap * 2 : 4;
		c->a = (uint32_t*)realloc(c->a, sizeof(uint32_t) * (size_t)c->cap);
	}
	c->a[c->n++] = (uint32_t)len << 4 | op;
}

/* result reset: ksw2.h:184-189 (cigar buffer and capacity survive) */
static void ez_reset(kso_extz_t *ez)
{
	ez->max = 0; ez->zdropped = 0;
	ez->max_q = ez->max_t = ez->mqe_t = ez->mte_q = -1;
	ez->mqe = ez->mte = ez->score = NEG;
	ez->n_cigar = 0; ez->reach_end = 0;
}

/* Z-drop test for the best cell (score H, row i, column j) of a finished row: ksw2.h:191-207, is_rot=0 */
static int zdrop_row(kso_extz_t *ez, int32_t H, int i, int j, int zdrop, int slope)
{
	if (H > (int32_t)ez->max) {
		ez->max = (uint32_t)H; ez->max_t = i; ez->max_q = j;
	} else if (i >= ez->max_t && j >= ez->max_q) {
		int dt = i - ez->max_t, dq = j - ez->max_q;
		int skew = dt > dq ? dt - dq : dq - dt;
		if (zdrop >= 0 && (int32_t)ez->max - H > zdrop + skew * slope) { ez->zdropped = 1; return 1; }
	}
	return 0;
}

/* value of the virtual row -1 / column -1 at distance k from the origin (SURVEY Appendix A.1;
 * ksw2_extz.c:32-35,43 and ksw2_extd.c:33-40,49-50) */
static inline int32_t border(int k, int q, int e, int q2, int e2, int dual)
{
	int32_t a, b;
	if (k <= 0) return 0;
	a = -(q + k * e);
	if (!dual) return a;
	b = -(q2 + k * e2);
	return a > b ? a : b;
}

/* ---------------------------------------------------------------- the DP core */

typedef struct {
	int qlen, tlen, m;
	const uint8_t *query, *target;
	const int8_t *mat;
	int q, e, q2, e2, dual;
	int w;                 /* already >= 0 */
	int zdrop, zslope;
	int right;             /* KSO_RIGHT tie rules for direction / continuation bits */
	int tie_first;         /* row maximum: 1 = first column wins ties, 0 = last column wins */
	int want_tb;
} job_t;

typedef struct {
	uint8_t *tb;           /* tlen x ncol direction bytes, row i starts at column max(0,i-w) */
	int ncol;
	int band_empty;        /* some row had no in-band cell (corner not reachable through the band) */
	int rows_done;
} fill_t;

/*
 * Fill the band row by row.  Follows ksw2_extz.c:38-125 / ksw2_extd.c:44-165 cell for cell, but with
 * explicit arrays:  up[j] = H(i-1,j),  ev[j] = E(i,j),  ev2[j] = E~(i,j)  on entry to row i.
 * Direction byte (ksw2.h:125-128): bits 0-2 winner (0 diag,1 E,2 F,3 E~,4 F~), 0x08/0x10/0x20/0x40 = the
 * E/F/E~/F~ gap that leaves this cell is an extension rather than a fresh opening.
 */
static void fill_band(const job_t *J, kso_extz_t *ez, fill_t *out)
{
	const int qlen = J->qlen, tlen = J->tlen, w = J->w, dual = J->dual;
	const int qe = J->q + J->e, qe2 = J->q2 + J->e2;
	int32_t *up, *ev, *ev2;
	int i, j;

	out->tb = 0; out->band_empty = 0; out->rows_done = 0;
	out->ncol = imin(qlen, 2 * w + 1);
	if (J->want_tb) out->tb = (uint8_t*)malloc((size_t)out->ncol * (size_t)tlen);
	up  = (int32_t*)malloc(sizeof(int32_t) * (size_t)qlen * 3);
	ev  = up + qlen; ev2 = ev + qlen;

	/* virtual row -1 (ksw2_extz.c:32-35; ksw2_extd.c:33-41) */
	for (j = 0; j < qlen; ++j) {
		int32_t hb = border(j + 1, J->q, J->e, J->q2, J->e2, dual);
		up[j] = hb;
		ev[j]  = j <= w ? hb - qe  : NEG;
		ev2[j] = j <= w ? hb - qe2 : NEG;
	}

	for (i = 0; i < tlen; ++i) {
		const int st = imax(0, i - w), en = imin(qlen - 1, i + w);
		const int8_t *srow = J->mat + (int)J->target[i] * J->m;
		uint8_t *trow = out->tb ? out->tb + (size_t)i * out->ncol : 0;
		int32_t diag, left, f, f2, best = NEG;
		int best_j = 0, reach;
		if (st > en) { out->band_empty = 1; break; }
		if (st == 0) {                         /* virtual column -1 (ksw2_extz.c:43-44; ksw2_extd.c:49-52) */
			left = border(i + 1, J->q, J->e, J->q2, J->e2, dual);
			f = left - qe; f2 = left - qe2;
			diag = border(i, J->q, J->e, J->q2, J->e2, dual);
		} else {
			left = NEG; f = f2 = NEG;
			diag = up[st - 1];
		}
		(void)left;
		for (j = st; j <= en; ++j) {
			/* the cell above is outside the band exactly at the right edge j == i+w (i>0) */
			const int has_up = (i == 0) || (j < i + w);
			int32_t ee = has_up ? ev[j] : NEG, ee2 = has_up ? ev2[j] : NEG;
			int32_t h = diag + srow[J->query[j]], open, open2;
			uint8_t d = 0;
			diag = up[j];
			if (!J->right) {                   /* ksw2_extz.c:72-75; ksw2_extd.c:88-95 */
				if (!(h >= ee)) { d = 1; h = ee; }
				if (!(h >= f))  { d = 2; h = f; }
				if (dual) {
					if (!(h >= ee2)) { d = 3; h = ee2; }
					if (!(h >= f2))  { d = 4; h = f2; }
				}
			} else {                           /* ksw2_extz.c:98-101; ksw2_extd.c:126-133 */
				if (!(h > ee)) { d = 1; h = ee; }
				if (!(h > f))  { d = 2; h = f; }
				if (dual) {
					if (!(h > ee2)) { d = 3; h = ee2; }
					if (!(h > f2))  { d = 4; h = f2; }
				}
			}
			up[j] = h;
			/* row maximum (ksw2_extz.c:54-55,77-78,103-104; ksw2_extd.c:64-65,97-98,135-136) */
			if (J->tie_first) { if (h > best)  { best = h; best_j = j; } }
			else              { if (h >= best) { best = h; best_j = j; } }
			/* gaps leaving this cell (ksw2_extz.c:79-86,105-112; ksw2_extd.c:99-114,137-152) */
			open = h - qe;
			ee -= J->e; f -= J->e;
			if (!J->right) {
				if (ee > open) d |= 0x08; else ee = open;
				if (f  > open) d |= 0x10; else f  = open;
			} else {
				if (ee >= open) d |= 0x08; else ee = open;
				if (f  >= open) d |= 0x10; else f  = open;
			}
			ev[j] = ee;
			if (dual) {
				open2 = h - qe2;
				ee2 -= J->e2; f2 -= J->e2;
				if (!J->right) {
					if (ee2 > open2) d |= 0x20; else ee2 = open2;
					if (f2  > open2) d |= 0x40; else f2  = open2;
				} else {
					if (ee2 >= open2) d |= 0x20; else ee2 = open2;
					if (f2  >= open2) d |= 0x40; else f2  = open2;
				}
				ev2[j] = ee2;
			}
			if (trow) trow[j - st] = d;
		}
		out->rows_done = i + 1;
		/* per-row bookkeeping, in this order (ksw2_extz.c:116-124; ksw2_extd.c:156-164) */
		reach = (en == qlen - 1);
		if (reach && up[qlen - 1] > ez->mqe) { ez->mqe = up[qlen - 1]; ez->mqe_t = i; }
		if (i == tlen - 1) { ez->mte = best; ez->mte_q = best_j; }
		if (zdrop_row(ez, best, i, best_j, J->zdrop, J->zslope)) break;
		if (i == tlen - 1 && reach) ez->score = up[qlen - 1];
	}
	free(up);
}

/* Walk the direction bytes back from (i,j); state machine of ksw2.h:129-161 for the row-major layout
 * (is_rot = 0, off_end = NULL, min_intron_len = 0). */
static void walk_back(const fill_t *F, int w, int i, int j, int keep_reversed, cig_t *c)
{
	int state = 0, k;
	c->n = 0;
	while (i >= 0 && j >= 0) {
		const int st = imax(0, i - w);
		int forced = -1;
		uint32_t d;
		if (j < st) forced = 2;                 /* left of the band: can only be an insertion */
		d = forced < 0 ? F->tb[(size_t)i * F->ncol + (j - st)] : 0;
		if (state == 0) state = d & 7;
		else if (!((d >> (state + 2)) & 1)) state = 0;
		if (state == 0) state = d & 7;
		if (forced >= 0) state = forced;
		if (state == 0)                      { cig_add(c, 0, 1); --i; --j; }   /* M */
		else if (state == 1 || state == 3)   { cig_add(c, 2, 1); --i; }        /* D */
		else                                 { cig_add(c, 1, 1); --j; }        /* I */
	}
	if (i >= 0) cig_add(c, 2, i + 1);
	if (j >= 0) cig_add(c, 1, j + 1);
	if (!keep_reversed)
		for (k = 0; k < c->n >> 1; ++k) {
			uint32_t t = c->a[k]; c->a[k] = c->a[c->n - 1 - k]; c->a[c->n - 1 - k] = t;
		}
}

/* M -> =/X rewrite (what ksw2.h:163-182 is meant to do; the reference drops krealloc's return there) */
static void to_eqx(const uint8_t *query, const uint8_t *target, cig_t *c)
{
	cig_t o = {0, 0, 0};
	int k, x = 0, y = 0, i;
	for (k = 0; k < c->n; ++k) {
		int op = c->a[k] & 0xf, len = (int)(c->a[k] >> 4);
		if (op == 0) {
			for (i = 0; i < len; ++i) cig_add(&o, target[x + i] == query[y + i] ? 7 : 8, 1);
			x += len; y += len;
		} else {
			cig_add(&o, (uint32_t)op, len);
			if (op == 2 || op == 3) x += len;
			else if (op == 1) y += len;
			else if (op == 7 || op == 8) { x += len; y += len; }
		}
	}
	free(c->a);
	*c = o;
}

/* ---------------------------------------------------------------- contracts */

typedef enum { CONTRACT_SCALAR = 0, CONTRACT_SSE_SIG = 1 } contract_t;

static void run_ext(contract_t ct, int qlen, const uint8_t *query, int tlen, const uint8_t *target, int m,
                    const int8_t *mat, int q, int e, int q2, int e2, int dual, int w, int zdrop, int end_bonus,
                    int flag, kso_extz_t *ez)
{
	job_t J;
	fill_t F;
	cig_t c;
	int8_t *eff = 0;
	/* KSW_EZ_APPROX_MAX without KSW_EZ_APPROX_DROP (ksw2_extz2_sse.c:270-286, ksw2_extd2_sse.c:366-382): the SSE kernels then
	 * track one cell per diagonal only to deliver the final score -- no max / mqe / mte, no Z-drop at all -- so the result is
	 * { score, CIGAR from the corner unless EXTZ_ONLY } with every other field left reset.  (With APPROX_DROP the reference's
	 * heuristic drop test depends on its padded band; that mode is computed exactly here.) */
	const int approx = ct == CONTRACT_SSE_SIG && (flag & KSO_APPROX_MAX) && !(flag & KSO_APPROX_DROP);

	ez_reset(ez);
	if (approx) zdrop = -1;
	if (ct == CONTRACT_SSE_SIG) {
		int k, lo;
		/* ksw2_extz2_sse.c:57, ksw2_extd2_sse.c:76 */
		if ((dual ? m <= 1 : m <= 0) || qlen <= 0 || tlen <= 0) return;
		/* ksw2_extd2_sse.c:78: the cheaper-to-open piece goes first */
		if (dual && q2 + e2 < q + e) { int t = q; q = q2; q2 = t; t = e; e = e2; e2 = t; }
		/* ksw2_extz2_sse.c:78-82, ksw2_extd2_sse.c:96-100 */
		for (k = 1, lo = mat[1]; k < m * m; ++k) lo = imin(lo, mat[k]);
		if (-lo > 2 * (q + e)) return;
		if (!(flag & KSO_GENERIC_SC)) {
			/* implicit match/mismatch/wildcard scoring: ksw2_extz2_sse.c:66-69,125-140; ksw2_extd2_sse.c:85-88,166-180 */
			int a, b, scN = mat[m * m - 1] == 0 ? -(dual ? e2 : e) : mat[m * m - 1];
			eff = (int8_t*)malloc((size_t)m * m);
			for (a = 0; a < m; ++a)
				for (b = 0; b < m; ++b)
					eff[a * m + b] = (int8_t)((a == m - 1 || b == m - 1) ? scN : a == b ? mat[0] : mat[1]);
			mat = eff;
		}
	}
	if (w < 0 || w > imax(qlen, tlen)) w = imax(qlen, tlen);

	memset(&J, 0, sizeof(J));
	J.qlen = qlen; J.tlen = tlen; J.m = m; J.query = query; J.target = target; J.mat = mat;
	J.q = q; J.e = e; J.q2 = q2; J.e2 = e2; J.dual = dual; J.w = w;
	J.zdrop = zdrop; J.zslope = dual ? e2 : e;         /* ksw2_extz.c:122, ksw2_extd.c:162 */
	J.want_tb = !(flag & KSO_SCORE_ONLY);
	J.right = J.want_tb && (flag & KSO_RIGHT);         /* score-only ignores RIGHT: ksw2_extz.c:45 */
	/* SURVEY 8a rule 3: only extz + RIGHT + CIGAR breaks row-max ties towards the first column */
	J.tie_first = (!dual && J.right);

	fill_band(&J, ez, &F);
	/* rows (or the corner column) that the band cannot reach: behave like the SSE kernels, which stop
	 * with zdropped=1 when a diagonal has no in-band cell (ksw2_extz2_sse.c:111-114).  The scalar
	 * reference is undefined there, so this is outside the scalar parity contract. */
	if (!ez->zdropped && (F.band_empty || (tlen - 1) + w < qlen - 1)) ez->zdropped = 1;
	if (approx) {
		ez->max = 0; ez->max_t = ez->max_q = ez->mqe_t = ez->mte_q = -1; ez->mqe = ez->mte = NEG;
	}

	if (J.want_tb) {
		int rev = !!(flag & KSO_REV_CIGAR), si = -1, sj = -1;
		if (!ez->zdropped && !(flag & KSO_EXTZ_ONLY)) { si = tlen - 1; sj = qlen - 1; }
		else if (ct == CONTRACT_SSE_SIG && !ez->zdropped && (flag & KSO_EXTZ_ONLY) && ez->mqe + end_bonus > (int)ez->max) {
			ez->reach_end = 1; si = ez->mqe_t; sj = qlen - 1;   /* ksw2_extz2_sse.c:296-298 */
		} else if (ez->max_t >= 0 && ez->max_q >= 0) { si = ez->max_t; sj = ez->max_q; }
		if (si >= 0) {
			c.a = ez->cigar; c.n = 0; c.cap = ez->m_cigar;
			walk_back(&F, w, si, sj, rev, &c);
			if (ct == CONTRACT_SSE_SIG && dual && (flag & KSO_EQX)) to_eqx(query, target, &c);
			ez->cigar = c.a; ez->n_cigar = c.n; ez->m_cigar = c.cap;
		}
		free(F.tb);
	}
	free(eff);
}

void kso_extz(int qlen, const uint8_t *query, int tlen, const uint8_t *target, int8_t m, const int8_t *mat,
              int8_t q, int8_t e, int w, int zdrop, int flag, kso_extz_t *ez)
{
	run_ext(CONTRACT_SCALAR, qlen, query, tlen, target, m, mat, q, e, 0, 0, 0, w, zdrop, 0, flag, ez);
}

void kso_extd(int qlen, const uint8_t *query, int tlen, const uint8_t *target, int8_t m, const int8_t *mat,
              int8_t q, int8_t e, int8_t q2, int8_t e2, int w, int zdrop, int flag, kso_extz_t *ez)
{
	run_ext(CONTRACT_SCALAR, qlen, query, tlen, target, m, mat, q, e, q2, e2, 1, w, zdrop, 0, flag, ez);
}

void kso_extz2(int qlen, const uint8_t *query, int tlen, const uint8_t *target, int8_t m, const int8_t *mat,
               int8_t q, int8_t e, int w, int zdrop, int end_bonus, int flag, kso_extz_t *ez)
{
	run_ext(CONTRACT_SSE_SIG, qlen, query, tlen, target, m, mat, q, e, 0, 0, 0, w, zdrop, end_bonus, flag, ez);
}

void kso_extd2(int qlen, const uint8_t *query, int tlen, const uint8_t *target, int8_t m, const int8_t *mat,
               int8_t q, int8_t e, int8_t q2, int8_t e2, int w, int zdrop, int end_bonus, int flag, kso_extz_t *ez)
{
	run_ext(CONTRACT_SSE_SIG, qlen, query, tlen, target, m, mat, q, e, q2, e2, 1, w, zdrop, end_bonus, flag, ez);
}

/* Global alignment: ksw2_gg.c:6-102.  Same cells as kso_extz with Z-drop off, generic scoring,
 * left-aligned gaps; returns H(tlen-1,qlen-1).  CIGAR only when all three out-pointers are given.
 * If the band cannot reach the corner the reference's result is undefined: we
 * define it as KSO_NEG_INF with an empty CIGAR. */
int kso_gg(int qlen, const uint8_t *query, int tlen, const uint8_t *target, int8_t m, const int8_t *mat,
           int8_t q, int8_t e, int w, int *m_cigar, int *n_cigar, uint32_t **cigar)
{
	kso_extz_t ez;
	int with_cigar = m_cigar && n_cigar && cigar;
	memset(&ez, 0, sizeof(ez));
	if (with_cigar) { ez.cigar = *cigar; ez.m_cigar = *m_cigar; *n_cigar = 0; }
	if (qlen <= 0 || tlen <= 0 || m <= 0) return NEG;
	run_ext(CONTRACT_SCALAR, qlen, query, tlen, target, m, mat, q, e, 0, 0, 0, w, -1, 0,
	        with_cigar ? 0 : KSO_SCORE_ONLY, &ez);
	if (with_cigar) {
		if (ez.zdropped) ez.n_cigar = 0;
		*cigar = ez.cigar; *m_cigar = ez.m_cigar; *n_cigar = ez.n_cigar;
	}
	return ez.zdropped ? NEG : ez.score;
}

int kso_gg2(int qlen, const uint8_t *query, int tlen, const uint8_t *target, int8_t m, const int8_t *mat,
            int8_t q, int8_t e, int w, int *m_cigar, int *n_cigar, uint32_t **cigar)
{
	/* ksw_gg2 / ksw_gg2_sse return the same score and CIGAR as ksw_gg whenever w >= |tlen-qlen|
	 * (gg == gg2 == extz always; ksw2_gg2.c:102-107 sums the differences to H(tlen-1,qlen-1)). */
	return kso_gg(qlen, query, tlen, target, m, mat, q, e, w, m_cigar, n_cigar, cigar);
}

int64_t kso_band_cells(int qlen, int tlen, int w)
{
	int64_t n = 0;
	int i;
	if (qlen <= 0 || tlen <= 0) return 0;
	if (w < 0 || w > imax(qlen, tlen)) w = imax(qlen, tlen);
	for (i = 0; i < tlen; ++i) {
		int st = imax(0, i - w), en = imin(qlen - 1, i + w);
		if (st <= en) n += en - st + 1;
	}
	return n;
}

int kso_cigar_score(int n_cigar, const uint32_t *cigar, int qlen, const uint8_t *query, int tlen, const uint8_t *target, int m, const int8_t *mat,
                    int q, int e, int q2, int e2, int *qused, int *tused)
{
	int x = 0, y = 0, k, i;        /* x: target bases consumed, y: query bases */
	int64_t sc = 0;
	for (k = 0; k < n_cigar; ++k) {
		const int op = (int)(cigar[k] & 0xf), len = (int)(cigar[k] >> 4);
		if (op == 0) {
			if (x + len > tlen || y + len > qlen) return KSO_NEG_INF;
			for (i = 0; i < len; ++i) sc += mat[(int)target[x + i] * m + query[y + i]];
			x += len; y += len;
		} else if (op == 1 || op == 2) {
			int64_t c = (int64_t)q + (int64_t)len * e;
			if (q2 >= 0) { const int64_t c2 = (int64_t)q2 + (int64_t)len * e2; if (c2 < c) c = c2; }
			sc -= c;
			if (op == 2) { x += len; if (x > tlen) return KSO_NEG_INF; } else { y += len; if (y > qlen) return KSO_NEG_INF; }
		} else return KSO_NEG_INF;
	}
	if (qused) *qused = y;
	if (tused) *tused = x;
	return (int)sc;
}
