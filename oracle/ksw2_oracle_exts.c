/*
 * ksw2_oracle_exts.c -- CPU restatement of ksw_exts2_sse (splice-aware extension, ksw2_exts2_sse.c:33-415).
 *
 * TEST INFRASTRUCTURE ONLY (see ksw2_oracle.h).  Parity status: PINNED against the compiled reference
 * (oracle/_ref/libksw2ref.so, tests/test_oracle_exts.py and the golden vectors made by oracle/gen_golden_exts.py).
 *
 * The reference has no scalar version of this function; its SSE code is the definition.  It has no band, so none of
 * the 16-lane padding effects of the banded kernels arise: every cell of the qlen x tlen matrix is well defined by the
 * recurrence below, restated here row by row in int32 absolute scores (the reference carries int8 differences,
 * tex/ksw2.tex:78-176; equal as long as those differences fit int8, which the reference silently assumes as well).
 *
 *   gap types: E / F with cost q + k*e (target / query consuming); a long target-consuming gap E~ ("intron") with open
 *   q2 - donor[i], no extension cost, and acceptor[i] added when it closes (ksw2_exts2_sse.c:36-66,246-260);
 *   H(i,j) = max{ H(i-1,j-1) + s, E, F, E~ + acceptor[i] };  E~(i+1,j) = max{ E~(i,j), H(i,j) + donor[i] - q2 }.
 *   Row -1 and column -1: -(q + k*e) for k <= long_thres, then -q2 (ksw2_exts2_sse.c:102-105,196-213).
 *
 * What is NOT row-wise is the bookkeeping (ksw2_exts2_sse.c:345-384): max / Z-drop / mqe / mte are evaluated once per
 * anti-diagonal r = i + j, on the diagonal's best cell, and ties inside a diagonal follow the reference's 4-lane scan
 * (its last cell first, then four interleaved lanes, then the scalar tail).  mte_q is taken from the 16-padded end of
 * the diagonal (`r - en`, :373), as the reference does.
 */
#include <stdlib.h>
#include <string.h>
#include "ksw2_oracle.h"

#define NEG KSO_NEG_INF

typedef struct { uint32_t *a; int n, cap; } cigx_t;

static void cigx_add(cigx_t *c, uint32_t op, int len)          /* ksw2.h:113-123 */
{
	if (c->n > 0 && (c->a[c->n - 1] & 0xfu) == op) { c->a[c->n - 1] += (uint32_t)len << 4; return; }
	if (c->n == c->cap) {
		c->cap = c->cap ? c->cap * 2 : 4;
		c->a = (uint32_t*)realloc(c->a, sizeof(uint32_t) * (size_t)c->cap);
	}
	c->a[c->n++] = (uint32_t)len << 4 | op;
}

/* donor[i] / acceptor[i]: 0 at a canonical site with the preferred flanking base, -noncan/2 (with KSO_SPLICE_FLANK)
 * or 0 at a plain GT / AG, -noncan elsewhere, plus junc_bonus at annotated junctions (ksw2_exts2_sse.c:121-173).
 * Assumes the 0/1/2/3 = A/C/G/T encoding like the reference. */
void kso_splice_signals(int tlen, const uint8_t *target, int noncan, int junc_bonus, int flag, const uint8_t *junc,
                        int8_t *donor, int8_t *acceptor)
{
	int t;
	memset(donor, 0, (size_t)tlen);
	memset(acceptor, 0, (size_t)tlen);
	if (!(flag & (KSO_SPLICE_FOR | KSO_SPLICE_REV))) return;
	{
		const int semi = (flag & KSO_SPLICE_FLANK) ? -noncan / 2 : 0;
		const int fwd = !!(flag & KSO_SPLICE_FOR), rev = !!(flag & KSO_SPLICE_REV);
		memset(donor, -noncan, (size_t)tlen);
		memset(acceptor, -noncan, (size_t)tlen);
		if (!(flag & KSO_REV_CIGAR)) {
			for (t = 0; t < tlen - 4; ++t) {
				int can = 0;
				if (fwd && target[t + 1] == 2 && target[t + 2] == 3) can = 1;   /* GT */
				if (rev && target[t + 1] == 1 && target[t + 2] == 3) can = 1;   /* CT */
				if (can && (target[t + 3] == 0 || target[t + 3] == 2)) can = 2;
				if (can) donor[t] = (int8_t)(can == 2 ? 0 : semi);
			}
			if (junc)
				for (t = 0; t < tlen - 1; ++t)
					if ((fwd && (junc[t + 1] & 1)) || (rev && (junc[t + 1] & 8))) donor[t] = (int8_t)(donor[t] + junc_bonus);
			for (t = 2; t < tlen; ++t) {
				int can = 0;
				if (fwd && target[t - 1] == 0 && target[t] == 2) can = 1;       /* AG */
				if (rev && target[t - 1] == 0 && target[t] == 1) can = 1;       /* AC */
				if (can && (target[t - 2] == 1 || target[t - 2] == 3)) can = 2;
				if (can) acceptor[t] = (int8_t)(can == 2 ? 0 : semi);
			}
			if (junc)
				for (t = 0; t < tlen; ++t)
					if ((fwd && (junc[t] & 2)) || (rev && (junc[t] & 4))) acceptor[t] = (int8_t)(acceptor[t] + junc_bonus);
		} else {
			for (t = 0; t < tlen - 4; ++t) {
				int can = 0;
				if (fwd && target[t + 1] == 2 && target[t + 2] == 0) can = 1;   /* GA */
				if (rev && target[t + 1] == 1 && target[t + 2] == 0) can = 1;   /* CA */
				if (can && (target[t + 3] == 1 || target[t + 3] == 3)) can = 2;
				if (can) donor[t] = (int8_t)(can == 2 ? 0 : semi);
			}
			if (junc)
				for (t = 0; t < tlen - 1; ++t)
					if ((fwd && (junc[t + 1] & 2)) || (rev && (junc[t + 1] & 4))) donor[t] = (int8_t)(donor[t] + junc_bonus);
			for (t = 2; t < tlen; ++t) {
				int can = 0;
				if (fwd && target[t - 1] == 3 && target[t] == 2) can = 1;       /* TG */
				if (rev && target[t - 1] == 3 && target[t] == 1) can = 1;       /* TC */
				if (can && (target[t - 2] == 0 || target[t - 2] == 2)) can = 2;
				if (can) acceptor[t] = (int8_t)(can == 2 ? 0 : semi);
			}
			if (junc)
				for (t = 0; t < tlen; ++t)
					if ((fwd && (junc[t] & 1)) || (rev && (junc[t] & 8))) acceptor[t] = (int8_t)(acceptor[t] + junc_bonus);
		}
	}
}

/* intron threshold of the gap model: the gap length from which the long piece is the cheaper one (ksw2_exts2_sse.c:102-105) */
int kso_long_thres(int q, int e, int q2)
{
	int lt = (q2 - q) / e - 1;
	if (q2 > q + e + lt * e) ++lt;
	return lt;
}

/* best cell of anti-diagonal r in the reference's scan order (ksw2_exts2_sse.c:347-377); Hd[t] = H(t, r - t) */
static void diag_best(const int32_t *Hd, int r, int st0, int en0, int32_t *max_H, int *max_t)
{
	int32_t mH, lane_H[4];
	int mt, lane_t[4], t, i, en1;
	if (r == 0) { *max_H = Hd[0]; *max_t = 0; return; }
	mH = Hd[en0]; mt = en0;
	en1 = st0 + (en0 - st0) / 4 * 4;
	for (i = 0; i < 4; ++i) { lane_H[i] = mH; lane_t[i] = mt; }
	for (t = st0; t < en1; t += 4)
		for (i = 0; i < 4; ++i)
			if (Hd[t + i] > lane_H[i]) { lane_H[i] = Hd[t + i]; lane_t[i] = t; }      /* group base; + i below */
	for (i = 0; i < 4; ++i)
		if (mH < lane_H[i]) { mH = lane_H[i]; mt = lane_t[i] + i; }
	for (t = en1; t < en0; ++t)
		if (Hd[t] > mH) { mH = Hd[t]; mt = t; }
	*max_H = mH; *max_t = mt;
}

void kso_exts2(int qlen, const uint8_t *query, int tlen, const uint8_t *target, int8_t m, const int8_t *mat,
               int8_t q, int8_t e, int8_t q2, int8_t noncan, int zdrop, int8_t junc_bonus, int flag, const uint8_t *junc,
               kso_extz_t *ez)
{
	const int with_cigar = !(flag & KSO_SCORE_ONLY), right = !!(flag & KSO_RIGHT);
	/* approximate-max mode without APPROX_DROP (ksw2_exts2_sse.c:386-404): only the final score is tracked */
	const int approx = (flag & KSO_APPROX_MAX) && !(flag & KSO_APPROX_DROP);
	int i, j, r, k, min_sc, long_thres, scN;
	int32_t *H, *En, *E2n, *Hd;
	uint8_t *dir = 0;
	int8_t *donor, *acceptor;

	ez->max = 0; ez->zdropped = 0; ez->max_q = ez->max_t = ez->mqe_t = ez->mte_q = -1;
	ez->mqe = ez->mte = ez->score = NEG; ez->n_cigar = 0; ez->reach_end = 0;
	if (m <= 1 || qlen <= 0 || tlen <= 0 || q2 <= q + e) return;                     /* ksw2_exts2_sse.c:74 */
	for (k = 1, min_sc = mat[1]; k < m * m; ++k) min_sc = min_sc < mat[k] ? min_sc : mat[k];
	if (-min_sc > 2 * (q + e)) return;                                                /* :91 */
	long_thres = kso_long_thres(q, e, q2);
	scN = mat[m * m - 1] == 0 ? -e : mat[m * m - 1];

	H = (int32_t*)malloc(sizeof(int32_t) * (size_t)tlen * qlen);
	En = (int32_t*)malloc(sizeof(int32_t) * (size_t)qlen);
	E2n = (int32_t*)malloc(sizeof(int32_t) * (size_t)qlen);
	Hd = (int32_t*)malloc(sizeof(int32_t) * (size_t)tlen);
	donor = (int8_t*)malloc((size_t)tlen);
	acceptor = (int8_t*)malloc((size_t)tlen);
	if (with_cigar) dir = (uint8_t*)malloc((size_t)tlen * qlen);
	kso_splice_signals(tlen, target, noncan, junc_bonus, flag, junc, donor, acceptor);

#define HB(k_) ((k_) <= 0 ? 0 : (k_) <= long_thres ? -(q + (k_) * e) : -q2)      /* H(-1, k-1) = H(k-1, -1) */
	for (i = 0; i < tlen; ++i) {
		int32_t F = HB(i + 1) - q - e;                                                /* F entering (i, 0) */
		for (j = 0; j < qlen; ++j) {
			int32_t s, diag, a, a2, a2a, z, t1, t2;
			int d;
			if (flag & KSO_GENERIC_SC) s = mat[target[i] * m + query[j]];
			else s = (target[i] == m - 1 || query[j] == m - 1) ? scN : target[i] == query[j] ? mat[0] : mat[1];
			diag = (i > 0 && j > 0) ? H[(size_t)(i - 1) * qlen + j - 1] : i == 0 ? HB(j) : HB(i);
			a = i == 0 ? HB(j + 1) - q - e : En[j];
			a2 = i == 0 ? HB(j + 1) - q2 : E2n[j];
			a2a = a2 + acceptor[i];
			z = diag + s;
			if (!right) {
				d = a > z ? 1 : 0;   z = z > a ? z : a;
				d = F > z ? 2 : d;   z = z > F ? z : F;
				d = a2a > z ? 3 : d; z = z > a2a ? z : a2a;
			} else {
				d = z > a ? 0 : 1;   z = z > a ? z : a;
				d = z > F ? d : 2;   z = z > F ? z : F;
				d = z > a2a ? d : 3; z = z > a2a ? z : a2a;
			}
			H[(size_t)i * qlen + j] = z;
			t1 = z - q; t2 = z - q2 + donor[i];
			if (!right) {
				if (a > t1) d |= 0x08;
				if (F > t1) d |= 0x10;
				if (a2 > t2) d |= 0x20;
			} else {
				if (a >= t1) d |= 0x08;
				if (F >= t1) d |= 0x10;
				if (a2 >= t2) d |= 0x20;
			}
			En[j] = (a > t1 ? a : t1) - e;
			F = (F > t1 ? F : t1) - e;
			E2n[j] = a2 > t2 ? a2 : t2;
			if (with_cigar) dir[(size_t)i * qlen + j] = (uint8_t)d;
		}
	}
#undef HB

	/* bookkeeping per anti-diagonal (exact-max mode; APPROX_MAX | APPROX_DROP is computed exactly as well) */
	if (approx) ez->score = H[(size_t)(tlen - 1) * qlen + qlen - 1];
	for (r = 0; !approx && r < qlen + tlen - 1; ++r) {
		int st0 = 0, en0 = tlen - 1, t, max_t;
		int32_t max_H;
		if (st0 < r - qlen + 1) st0 = r - qlen + 1;
		if (en0 > r) en0 = r;
		for (t = st0; t <= en0; ++t) Hd[t] = H[(size_t)t * qlen + (r - t)];
		diag_best(Hd, r, st0, en0, &max_H, &max_t);
		if (en0 == tlen - 1 && Hd[en0] > ez->mte) { ez->mte = Hd[en0]; ez->mte_q = r - ((en0 + 16) / 16 * 16 - 1); }   /* :373, padded `en` */
		if (r - st0 == qlen - 1 && Hd[st0] > ez->mqe) { ez->mqe = Hd[st0]; ez->mqe_t = st0; }
		if (max_H > (int32_t)ez->max) { ez->max = (uint32_t)max_H; ez->max_t = max_t; ez->max_q = r - max_t; }      /* ksw2.h:191-207, is_rot, e = 0 */
		else if (max_t >= ez->max_t && r - max_t >= ez->max_q) {
			if (zdrop >= 0 && (int32_t)ez->max - max_H > zdrop) { ez->zdropped = 1; break; }
		}
		if (r == qlen + tlen - 2 && en0 == tlen - 1) ez->score = Hd[tlen - 1];
	}

	if (with_cigar) {                                                                 /* ksw2.h:129-161 with is_rot = 1 */
		int si = -1, sj = -1, rev = !!(flag & KSO_REV_CIGAR), state = 0;
		cigx_t c;
		if (!ez->zdropped && !(flag & KSO_EXTZ_ONLY)) { si = tlen - 1; sj = qlen - 1; }
		else if (ez->max_t >= 0 && ez->max_q >= 0) { si = ez->max_t; sj = ez->max_q; }
		if (si >= 0) {
			c.a = ez->cigar; c.n = 0; c.cap = ez->m_cigar;
			i = si; j = sj;
			while (i >= 0 && j >= 0) {
				const uint32_t tmp = dir[(size_t)i * qlen + j];
				if (state == 0) state = tmp & 7;
				else if (!((tmp >> (state + 2)) & 1)) state = 0;
				if (state == 0) state = tmp & 7;
				if (state == 0) { cigx_add(&c, 0, 1); --i; --j; }
				else if (state == 1 || (state == 3 && long_thres <= 0)) { cigx_add(&c, 2, 1); --i; }
				else if (state == 3 && long_thres > 0) { cigx_add(&c, 3, 1); --i; }
				else { cigx_add(&c, 1, 1); --j; }
			}
			if (i >= 0) cigx_add(&c, long_thres > 0 && i >= long_thres ? 3 : 2, i + 1);
			if (j >= 0) cigx_add(&c, 1, j + 1);
			if (!rev)
				for (k = 0; k < c.n >> 1; ++k) { uint32_t x = c.a[k]; c.a[k] = c.a[c.n - 1 - k]; c.a[c.n - 1 - k] = x; }
			ez->cigar = c.a; ez->n_cigar = c.n; ez->m_cigar = c.cap;
		}
	}
	free(H); free(En); free(E2n); free(Hd); free(donor); free(acceptor); free(dir);
}
