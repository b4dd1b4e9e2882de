/*
 * ksw2_lane_extf.h -- per-position code of the gap-linear X-drop extension (replaces ksw_extf2_sse, ksw2_extf2_sse.c:11-98;
 * SURVEY section 8f row N3), shared by the gfx950 kernel (ksw2_shim_hip.hip) and the lock-step simulator (tests/sim).
 *
 * The reference keeps three byte arrays over target positions -- U, V (difference encoding of the anti-diagonal DP, 8-bit
 * wrapping arithmetic) and S (match / mismatch score on the current anti-diagonal) -- and updates them in 16-byte blocks
 * around the in-band range [lo, hi] of anti-diagonal r.  Positions of those blocks outside [lo, hi] are updated as well,
 * from whatever S holds there, and the band's edge cells read them on later anti-diagonals: the result depends on the
 * blocking, so the kernel reproduces it position by position.  One alignment per wavefront, lane <-> target position,
 * 64 positions per pass over [blo, bhi]; U, V, S live in LDS (HBM scratch for targets over 21 k); V's left neighbour comes
 * through a one-lane shift with the previous pass's last V carried in.  The score follows one cell per anti-diagonal
 * (ksw2_extf2_sse.c:80-92): wavefront-uniform work on two bytes.
 */
#ifndef KSW2_LANE_EXTF_H_
#define KSW2_LANE_EXTF_H_

#include "ksw2_lane.h"

/* in-band target positions of anti-diagonal r (ksw2_extf2_sse.c:35-41), the 16-aligned block range around them (:42) and
 * the end of the positions whose S is recomputed (:48-61: chunks of 16 from lo, clipped to the padded array) */
struct K2aExtfDiag {
	int lo, hi, blo, bhi, fresh_end;
};
K2A_FN bool k2a_extf_diag(int r, int qlen, int tlen, int w, int tpad, K2aExtfDiag &d)
{
	d.lo = k2a_max3(0, r - qlen + 1, (r - w + 1) >> 1);
	d.hi = k2a_min(k2a_min(tlen - 1, r), (r + w) >> 1);
	if (d.lo > d.hi) return false;
	d.blo = d.lo & ~15; d.bhi = d.hi | 15;
	d.fresh_end = k2a_min(tpad, d.lo + (int)(((uint32_t)(d.hi - d.lo) >> 4) + 1u) * 16);      /* (hi >= lo: a shift, not a signed division) */
	return true;
}

/* S of position x on anti-diagonal r: codes read 0 past the target's end and outside the query (the reference's zeroed padding) */
K2A_FN uint32_t k2a_extf_score(const K2aExtf &par, const uint8_t *qa, const uint8_t *ta, int qlen, int tlen, int r, int x)
{
	const int j = r - x;
	const uint32_t tc = x < tlen ? ta[x] : 0u, qc = (j >= 0 && j < qlen) ? qa[j] : 0u;
	return (uint32_t)(tc == qc ? par.mch : par.mis) & 0xffu;
}

/* one position (ksw2_extf2_sse.c:64-78, SSE4.1 build): a = V of the position below on the previous anti-diagonal, b = U */
K2A_FN void k2a_extf_cell(uint32_t s, uint32_t a, uint32_t b, uint32_t two_e, uint32_t &u, uint32_t &v)
{
	uint32_t z = (s + two_e) & 0xffu;
	if ((int8_t)z < (int8_t)a) z = a;
	if (z < b) z = b;
	u = (z - a) & 0xffu; v = (z - b) & 0xffu;
}

/* the followed cell (ksw2_extf2_sse.c:80-92); vf = V[follow], un = U[follow + 1] after the update.  Returns false on X-drop. */
struct K2aExtfBook {
	int32_t H0, follow, max, max_t, max_q;
};
K2A_FN void k2a_extf_book_reset(K2aExtfBook &b) { b.H0 = 0; b.follow = 0; b.max = 0; b.max_t = b.max_q = -1; }
K2A_FN bool k2a_extf_follow(K2aExtfBook &b, const K2aExtfDiag &d, int r, int e, int xdrop, uint32_t vf, uint32_t un)
{
	if (r == 0) { b.H0 = (int)vf - e - e; b.follow = 0; return true; }          /* vf = V[0] */
	const bool in0 = b.follow >= d.lo && b.follow <= d.hi, in1 = b.follow + 1 >= d.lo && b.follow + 1 <= d.hi;
	if (in0 && in1) {
		const int d0 = (int)vf - e, d1 = (int)un - e;
		if (d0 > d1) b.H0 += d0;
		else { b.H0 += d1; ++b.follow; }
	} else if (in0) b.H0 += (int)vf - e;
	else { ++b.follow; b.H0 += (int)un - e; }
	if (b.H0 > b.max) { b.max = b.H0; b.max_t = b.follow; b.max_q = r - b.follow; }
	else if (xdrop >= 0 && b.max - b.H0 > xdrop) return false;
	return true;
}


/* ---- register window (k2a_extf_win_kernel<K>): K slots of 64 positions per wavefront, position x in lane x & 63 of slot
 * (x >> 6) & (K - 1).  The window starts at the 64-block of blo - 16 (the followed cell may sit one position below lo, the
 * carry into blo one below blo) and must reach hi + 15 (the padded block and the S refresh): 64 K >= in-band span + 109 (K2A_EXTF_WIN_SPAN, ksw2_types.h). */
K2A_FN int k2a_extf_win_base(const K2aExtfDiag &d) { return k2a_max(0, d.blo - 16) >> 6; }
/* block held by slot s when the window starts at block wb */
template<int K> K2A_FN int k2a_extf_win_block(int wb, int s) { return wb + ((s - wb) & (K - 1)); }

/* one position of one slot: vshift = old V of position x - 1 (what the lane shift delivers), qc = query code of (r, x).
 * tcode reads 0 past the target's end, qc 0 outside the query (the caller's job).  Updates u / v / sreg in place where the
 * reference would write them. */
K2A_FN void k2a_extf_win_cell(const K2aExtf &par, const K2aExtfDiag &d, int r, int x, uint32_t two_e, bool carry_ok,
                              uint32_t tcode, uint32_t qc, uint32_t vshift, uint32_t &u, uint32_t &v, uint32_t &sreg)
{
	/* unsigned range tests: x - lo < n  <=>  lo <= x < lo + n */
	const bool act = (uint32_t)(x - d.blo) <= (uint32_t)(d.bhi - d.blo), fresh = (uint32_t)(x - d.lo) < (uint32_t)(d.fresh_end - d.lo);
	const uint32_t sv = fresh ? ((uint32_t)(tcode == qc ? par.mch : par.mis) & 0xffu) : sreg;
	const uint32_t a = (x == d.blo && !carry_ok) ? 0u : vshift;
	const uint32_t b = (d.bhi >= r && x == r) ? 0u : u;
	uint32_t nu, nv;
	k2a_extf_cell(sv, a, b, two_e, nu, nv);
	if (act) { u = nu; v = nv; }
	sreg = sv;
}

/* ---- one extension per LANE (k2a_extf_lane_kernel): 64 extensions per wavefront, every lane runs the reference's loop over the
 * padded blocks of its own pair.  The three byte arrays and the sequences of a group of 64 pairs are interleaved by lane, four
 * positions per dword -- dword (x >> 2) * 64 + lane holds positions x .. x+3 of that lane's pair -- so a wavefront whose pairs
 * have the same shape touches one 256-byte row per access.  The query is stored reversed and zero-padded like the reference's
 * own copy (ksw2_extf2_sse.c:31), QR[k] = query[qlen - 1 - k]: the codes of positions x .. x+3 on anti-diagonal r are QR[k0 + x ..]
 * with k0 = qlen - 1 - r, one funnel shift of two neighbouring dwords.  Per cell this costs about 25 instructions of ONE lane
 * (2.7 wavefront instructions per cell in the position-per-lane kernels above, 0.4 here); it needs >= 64 extensions per
 * wavefront, so the host takes it for large batches (ksw2_host_*.c). */
struct K2aExtfLaneMem {
	uint32_t *U4, *V4, *S4;            /* state, this lane's column: index (x >> 2) * 64 -- or, as a ring, ((x >> 2) % ring) * 64 */
	const uint32_t *TT, *QR;           /* target codes by position, reversed query by k: same layout, zero past the ends */
	int ring, ztop;                    /* ring: rows of the LDS ring (0: whole arrays in HBM scratch, zeroed by the host side);
	                                    * ztop: first row the ring has not handed out yet -- a row entering the window is zeroed
	                                    * first, like the reference's freshly allocated arrays (its slot held row - ring) */
	K2A_FN size_t row(int x4) const { return (size_t)(ring ? x4 % ring : x4) * 64; }
};
K2A_FN uint32_t k2a_funnel(uint32_t lo, uint32_t hi, int bytes) { return bytes == 0 ? lo : (lo >> (8 * bytes)) | (hi << (32 - 8 * bytes)); }

/* anti-diagonal r of one lane's pair; false = the lane stops (band left the matrix or X-drop) */
K2A_FN bool k2a_extf_lane_diag(const K2aExtf &par, int qlen, int tlen, int w, int tpad, int xdrop, int r, K2aExtfLaneMem &m,
                               int &prev_lo, int &prev_hi, K2aExtfBook &bk)
{
	K2aExtfDiag d;
	if (!k2a_extf_diag(r, qlen, tlen, w, tpad, d)) return false;
	if (m.ring) {                                                  /* rows entering the window: up to the followed cell's neighbour */
		const int top = k2a_max(k2a_max(d.bhi, d.fresh_end - 1), bk.follow + 1) >> 2;
		for (; m.ztop <= top; ++m.ztop) { const size_t z = m.row(m.ztop); m.U4[z] = 0u; m.V4[z] = 0u; m.S4[z] = 0u; }
	}
	const uint32_t two_e = (uint32_t)(par.e * 2) & 0xffu, mch = (uint32_t)par.mch & 0xffu, mis = (uint32_t)par.mis & 0xffu;
	const bool top0 = d.bhi >= r;                                  /* ksw2_extf2_sse.c:46 */
	const int last = k2a_max(d.bhi, d.fresh_end - 1), k0 = qlen - 1 - r;
	uint32_t carry = 0;                                            /* V of position blo - 1 on the previous anti-diagonal (:45) */
	if (d.blo > 0 && d.blo - 1 >= prev_lo && d.blo - 1 <= prev_hi) carry = (m.V4[m.row((d.blo - 1) >> 2)] >> (8 * ((d.blo - 1) & 3))) & 0xffu;
	int slot = m.ring ? (d.blo >> 2) % m.ring : d.blo >> 2;        /* the ring index by increments: one modulo per anti-diagonal */
	for (int x4 = d.blo >> 2; x4 <= last >> 2; ++x4, slot = (m.ring && slot + 1 == m.ring) ? 0 : slot + 1) {
		const int x0 = x4 << 2;
		const size_t row = (size_t)slot * 64;
		uint32_t u4 = m.U4[row], v4 = m.V4[row], s4 = m.S4[row];
		const uint32_t t4 = m.TT[(size_t)x4 * 64];
		/* QR[k0 + x0 .. +3]; k0 + x0 < 0 only where the dword starts below lo: positions of it at or above lo read QR[0..] */
		const int k = k0 + x0, kc = k2a_max(k, 0);
		uint32_t q4 = k2a_funnel(m.QR[(size_t)(kc >> 2) * 64], m.QR[(size_t)((kc >> 2) + 1) * 64], kc & 3);
		if (k < 0) q4 = k > -4 ? q4 << (8 * -k) : 0u;
		uint32_t nu4 = u4, nv4 = v4, ns4 = s4;
#pragma unroll
		for (int i = 0; i < 4; ++i) {
			const int x = x0 + i, sh = 8 * i;
			const bool act = x <= d.bhi, fresh = x >= d.lo && x < d.fresh_end;
			const uint32_t tc = (t4 >> sh) & 0xffu, qc = (q4 >> sh) & 0xffu;
			const uint32_t sv = fresh ? (tc == qc ? mch : mis) : (s4 >> sh) & 0xffu;
			const uint32_t a = carry, vold = (v4 >> sh) & 0xffu;
			const uint32_t b = (top0 && x == r) ? 0u : (u4 >> sh) & 0xffu;
			uint32_t u, v;
			k2a_extf_cell(sv, a, b, two_e, u, v);
			carry = vold;
			if (act) { nu4 = (nu4 & ~(0xffu << sh)) | (u << sh); nv4 = (nv4 & ~(0xffu << sh)) | (v << sh); }
			if (fresh) ns4 = (ns4 & ~(0xffu << sh)) | (sv << sh);
		}
		m.U4[row] = nu4; m.V4[row] = nv4; m.S4[row] = ns4;
	}
	/* the followed cell reads the updated bytes (it may sit one position below lo: not touched on this anti-diagonal) */
	const int f0 = bk.follow, f1 = bk.follow + 1;
	const uint32_t vf = r == 0 ? (m.V4[0] & 0xffu) : (m.V4[m.row(f0 >> 2)] >> (8 * (f0 & 3))) & 0xffu;
	const uint32_t un = r == 0 ? 0u : (m.U4[m.row(f1 >> 2)] >> (8 * (f1 & 3))) & 0xffu;
	if (!k2a_extf_follow(bk, d, r, par.e, xdrop, vf, un)) return false;
	prev_lo = d.blo; prev_hi = d.bhi;
	return true;
}

/* rdone = anti-diagonals completed, nr = anti-diagonals of the pair; K2aResult.rows_done carries rdone here (diagnostics: the cells an
 * X-drop saved) */
K2A_FN void k2a_extf_finish(const K2aExtfBook &b, int rdone, int nr, K2aResult *r)
{
	const bool complete = rdone == nr;
	r->max = b.max; r->max_t = b.max_t; r->max_q = b.max_q;
	r->zdropped = complete ? 0 : 1;
	r->score = complete ? b.H0 : K2A_NEG;
	r->mqe = r->mte = K2A_NEG; r->mqe_t = r->mte_q = -1;
	r->reach_end = 0; r->n_cigar = 0; r->rows_done = rdone; r->ti = r->tj = -1;
}

#endif
