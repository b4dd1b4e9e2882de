/*
 * ksw2_lane_dm.h -- per-lane code of the splice-aware extension (replaces ksw_exts2_sse, ksw2_exts2_sse.c:33-415).
 *
 * Unlike the banded kernels (ksw2_lane.h) this function is unbanded and does its bookkeeping -- running maximum,
 * Z-drop, mqe / mte -- once per ANTI-DIAGONAL r = i + j on the diagonal's best cell (ksw2_exts2_sse.c:345-384), with
 * ties inside a diagonal resolved by the reference's own scan order.  So the kernel is diagonal-major like the
 * reference: one alignment per wavefront, lane <-> target position t (64 per register "slot", 8 or 16 slots in a
 * window that slides with the diagonal; longer diagonals keep the same state in a scratch array), one step per diagonal.  A lane keeps H of the last two diagonals and the gap
 * states leaving its cell; the values a cell needs from row t-1 arrive by a one-lane shift (DPP wave_shr:1 with the
 * previous slot's last lane carried in).  Everything is int32 absolute scores (the reference's int8 differences, summed).
 *
 *   H(i,j)    = max{ H(i-1,j-1) + s(i,j), E(i,j), F(i,j), E~(i,j) + acceptor[i] }
 *   E(i+1,j)  = max{ E(i,j), H(i,j) - q } - e         F(i,j+1) likewise
 *   E~(i+1,j) = max{ E~(i,j), H(i,j) - q2 + donor[i] }                  (long gap: no extension cost)
 *   row -1 / column -1: Hb(k) = -(q + k e) for k <= long_thres, else -q2      (ksw2_exts2_sse.c:102-105,196-213)
 *
 * Direction byte (ksw2.h:125-128): bits 0-2 winner {0 diag, 1 E, 2 F, 3 E~}, 0x08 / 0x10 / 0x20 = the E / F / E~ leaving
 * the cell continues a gap; strictness of every comparison as in ksw2_exts2_sse.c:262-343 (left / right alignment).
 */
#ifndef KSW2_LANE_DM_H_
#define KSW2_LANE_DM_H_

#include "ksw2_lane.h"


/* virtual row -1 / column -1 at distance k from the origin */
K2A_FN int k2a_dm_border(const K2aSplice &sp, int k)
{
	return k <= 0 ? 0 : k <= sp.long_thres ? -(sp.q + k * sp.e) : -sp.q2;
}

/* per-target-position constants, packed by the host: bits 0-7 residue code, 8-15 donor[t], 16-23 acceptor[t] (int8) */
K2A_FN uint32_t k2a_dm_pack(uint32_t code, int donor, int acceptor)
{
	return (code & 0xffu) | ((uint32_t)(uint8_t)(int8_t)donor << 8) | ((uint32_t)(uint8_t)(int8_t)acceptor << 16);
}

/* one cell.  diag = H(i-1,j-1), ein / e2in / fin = E, E~, F entering the cell, s = score, cst = packed constants of row i.
 * MODE as in the banded kernels (0 score only, 1 left-aligned, 2 right-aligned gaps). */
template<int MODE>
K2A_FN void k2a_dm_cell(const K2aSplice &sp, int diag, int ein, int e2in, int fin, int s, uint32_t cst,
                        int &z, int &en, int &e2n, int &fn, uint32_t &dir)
{
	const int donor = (int)(int8_t)(cst >> 8), acceptor = (int)(int8_t)(cst >> 16);
	const int a2a = e2in + acceptor;
	uint32_t d = 0;
	z = diag + s;
	if (MODE == K2A_MODE_RIGHT) {
		d = z > ein ? 0u : 1u;  z = k2a_max(z, ein);
		d = z > fin ? d : 2u;   z = k2a_max(z, fin);
		d = z > a2a ? d : 3u;   z = k2a_max(z, a2a);
	} else {
		d = ein > z ? 1u : 0u;  z = k2a_max(z, ein);
		d = fin > z ? 2u : d;   z = k2a_max(z, fin);
		d = a2a > z ? 3u : d;   z = k2a_max(z, a2a);
	}
	const int t1 = z - sp.q, t2 = z - sp.q2 + donor;
	if (MODE == K2A_MODE_LEFT) {
		d |= ein > t1 ? 0x08u : 0u; d |= fin > t1 ? 0x10u : 0u; d |= e2in > t2 ? 0x20u : 0u;
	} else if (MODE == K2A_MODE_RIGHT) {
		d |= ein >= t1 ? 0x08u : 0u; d |= fin >= t1 ? 0x10u : 0u; d |= e2in >= t2 ? 0x20u : 0u;
	}
	en = k2a_max(ein, t1) - sp.e;
	fn = k2a_max(fin, t1) - sp.e;
	e2n = k2a_max(e2in, t2);
	dir = d;
}

/* The best cell of a diagonal in the reference's scan order (ksw2_exts2_sse.c:347-377): the diagonal's last cell en0 first;
 * it is replaced only by a strictly larger cell of the "4-lane region" [st0, en1), among which the larger H, then the
 * smaller (t - st0) mod 4, then the smaller t wins; then by a strictly larger cell of the tail [en1, en0) in ascending t.
 * Lanes turn their best region cell into a key whose maximum over the wavefront is that winner. */
K2A_FN uint64_t k2a_dm_key(int H, int t, int st0)
{
	const uint32_t i = (uint32_t)(t - st0) & 3u;
	return ((uint64_t)((uint32_t)H ^ 0x80000000u) << 32) | (uint64_t)(0xffffffffu - ((i << 28) | (uint32_t)t));
}
K2A_FN int k2a_dm_key_H(uint64_t key) { return (int)((uint32_t)(key >> 32) ^ 0x80000000u); }
K2A_FN int k2a_dm_key_t(uint64_t key) { return (int)((0xffffffffu - (uint32_t)key) & 0x0fffffffu); }

/* per-diagonal bookkeeping on uniform values (ksw2_exts2_sse.c:371-377, ksw_apply_zdrop with is_rot = 1 and e = 0):
 * A = H at en0, Bkey = region winner (0 = empty region), T0..T2 = H at en1 + 0..2, S = H at st0.  Returns 1 on a Z-drop. */
K2A_FN int k2a_dm_book(K2aBook *b, int r, int st0, int en0, int qlen, int tlen, int zdrop, int A, uint64_t Bkey, int T0, int T1, int T2, int S)
{
	const int en1 = st0 + (en0 - st0) / 4 * 4;
	int max_H = A, max_t = en0;
	if (Bkey != 0 && k2a_dm_key_H(Bkey) > max_H) { max_H = k2a_dm_key_H(Bkey); max_t = k2a_dm_key_t(Bkey); }
	if (en1 < en0 && T0 > max_H) { max_H = T0; max_t = en1; }
	if (en1 + 1 < en0 && T1 > max_H) { max_H = T1; max_t = en1 + 1; }
	if (en1 + 2 < en0 && T2 > max_H) { max_H = T2; max_t = en1 + 2; }
	if (en0 == tlen - 1 && A > b->mte) { b->mte = A; b->mte_q = r - ((en0 + 16) / 16 * 16 - 1); }   /* the reference's padded `en` */
	if (r - st0 == qlen - 1 && S > b->mqe) { b->mqe = S; b->mqe_t = st0; }
	if (max_H > b->max) { b->max = max_H; b->max_t = max_t; b->max_q = r - max_t; }
	else if (max_t >= b->max_t && r - max_t >= b->max_q) {
		if (zdrop >= 0 && b->max - max_H > zdrop) { b->dropped = 1; return 1; }
	}
	if (r == qlen + tlen - 2 && en0 == tlen - 1) b->score = A;
	b->rows = r + 1;
	return 0;
}

/* ksw_backtrack with is_rot = 1 (ksw2.h:129-161) on the diagonal-major direction bytes tb[r*ncol + i - st0(r)]:
 * writes the CIGAR in walk order (end -> start), returns the op count.  State 3 is an intron (N) when long_thres > 0. */
K2A_FN int k2a_dm_trace(const uint8_t *tb, int ncol, int i, int j, uint32_t *out, int qlen, int long_thres)
{
	int n = 0, state = 0;
	uint32_t last_op = 0xffffffffu, run = 0;
	while (i >= 0 && j >= 0) {
		const int r = i + j, st0 = k2a_max(0, r - qlen + 1);
		const uint32_t d = tb[(size_t)r * ncol + (i - st0)];
		if (state == 0) state = d & 7;
		else if (!((d >> (state + 2)) & 1)) state = 0;
		if (state == 0) state = d & 7;
		uint32_t op;
		if (state == 0) { op = 0; --i; --j; }
		else if (state == 1 || (state == 3 && long_thres <= 0)) { op = 2; --i; }
		else if (state == 3) { op = 3; --i; }
		else { op = 1; --j; }
		if (op == last_op) ++run;
		else { if (run) out[n++] = run << 4 | last_op; last_op = op; run = 1; }
	}
	if (i >= 0) {                                     /* leading deletion / intron */
		const uint32_t op = long_thres > 0 && i >= long_thres ? 3u : 2u;
		if (last_op == op) run += i + 1;
		else { if (run) out[n++] = run << 4 | last_op; last_op = op; run = i + 1; }
	}
	if (j >= 0) {                                     /* leading insertion */
		if (last_op == 1) run += j + 1;
		else { if (run) out[n++] = run << 4 | last_op; last_op = 1; run = j + 1; }
	}
	if (run) out[n++] = run << 4 | last_op;
	return n;
}


/* donor[t] / acceptor[t] of the reference's splice model (ksw2_exts2_sse.c:121-173; residues 0/1/2/3 = A/C/G/T), packed with
 * the residue code into the per-position dword the kernels read: -noncan everywhere, 0 where an intron may start after t
 * (G T | C T on the reverse strand, read through a reversed CIGAR as G A | C A) resp. end at t (A G | A C, reversed T G | T C) with
 * the preferred flanking base, half the penalty (KSW_EZ_SPLICE_FLANK) or 0 without it; junc_bonus on annotated junction positions.
 * `fl`: the KSW_EZ_SPLICE_* / REV_CIGAR bits; J = 0: no annotation.  Runs on the device (k2a_splice_const_kernel): the host
 * uploads target and annotation bytes, not four bytes per position. */
K2A_FN uint32_t k2a_splice_const(const uint8_t *T, const uint8_t *J, int t, int tlen, int fl, int noncan, int junc_bonus)
{
	const bool fwd = (fl & K2A_F_SPLICE_FOR) != 0, rev = (fl & K2A_F_SPLICE_REV) != 0, rc = (fl & K2A_F_REV_CIGAR) != 0;
	const bool on = fwd || rev;
	const int base = on ? -noncan : 0, semi = (fl & K2A_F_SPLICE_FLANK) ? -noncan / 2 : 0;
	/* motif bases seen from position t: donor looks at t+1, t+2 (flank t+3), acceptor at t-1, t (flank t-2) */
	const int d1f = 2, d1r = 1, d2 = rc ? 0 : 3, a1 = rc ? 3 : 0, a2f = 2, a2r = 1;
	const int jd_f = rc ? 2 : 1, jd_r = rc ? 4 : 8, ja_f = rc ? 1 : 2, ja_r = rc ? 8 : 4;
	int don = base, acc = base;
	if (on) {
		if (t < tlen - 4 && T[t + 2] == d2 && ((fwd && T[t + 1] == d1f) || (rev && T[t + 1] == d1r))) {
			const bool flank = rc ? (T[t + 3] == 1 || T[t + 3] == 3) : (T[t + 3] == 0 || T[t + 3] == 2);
			don = flank ? 0 : semi;
		}
		if (J && t < tlen - 1 && ((fwd && (J[t + 1] & jd_f)) || (rev && (J[t + 1] & jd_r)))) don = (int)(int8_t)(don + junc_bonus);
		if (t >= 2 && T[t - 1] == a1 && ((fwd && T[t] == a2f) || (rev && T[t] == a2r))) {
			const bool flank = rc ? (T[t - 2] == 0 || T[t - 2] == 2) : (T[t - 2] == 1 || T[t - 2] == 3);
			acc = flank ? 0 : semi;
		}
		if (J && ((fwd && (J[t] & ja_f)) || (rev && (J[t] & ja_r)))) acc = (int)(int8_t)(acc + junc_bonus);
	}
	return (uint32_t)T[t] | (uint32_t)(uint8_t)(int8_t)don << 8 | (uint32_t)(uint8_t)(int8_t)acc << 16;
}

#endif
