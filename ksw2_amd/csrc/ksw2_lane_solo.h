/*
 * ksw2_lane_solo.h -- packed-int16 per-lane code for ONE alignment per 64-lane wavefront that uses BOTH 16-bit halves of
 * every register: the low half carries rows [i0, i0+C) of a "double strip" at query column jj, the high half rows
 * [i0+C, i0+2C) at column jj-1.  The two halves are two consecutive strips of the schedule of ksw2_lane.h (strip 2D and
 * 2D+1, one step apart), so the low half's bottom row reaches the high half inside the lane one step later, and the high
 * half's bottom row reaches the next lane's low half through the usual one-lane rotate.  Same cell update and number
 * format as K2aLanePk (offset form, row bias, per-strip score base = re-based variant; each half has its own base, like the
 * two alignments of a K2aLanePk lane, so the band window a half must hold is that of a C-row strip), but no partner
 * alignment of identical shape is needed: this is the kernel for reads whose shape is unique in their batch, and for
 * batches of so few long reads that one read per wavefront fills more SIMDs than two (1 024 reads of 10 k on 1 024 SIMDs).
 * Rows per half: K2A_SOLO_C with a traceback, K2A_SOLO_CS score only (ksw2_shim.h).
 *
 * Query codes: the high half works on the column the low half had one step earlier, so the lane keeps the previous code in
 * the high half of `qb`; the codes themselves come four steps per unaligned dword (load_query_group, as in K2aLanePk).
 *
 * Differences to K2aLanePk worth knowing when reading step(): the live-row mask, the "cell above is outside the band" test
 * and the column index differ between the halves; the strip epilogue runs once per double strip, over 2C rows, after the
 * high half has finished (the low half's last-column H values are saved when the low half finishes, C+1 steps earlier).
 * Replaces the same reference loops as ksw2_lane_pk.h with the scalar ksw_extz / ksw_extd semantics.
 */
#ifndef KSW2_LANE_SOLO_H_
#define KSW2_LANE_SOLO_H_

#include "ksw2_lane_pk.h"

#define K2A_SOLO_STAGE(C) (6 * (C) + 4)   /* LDS words of a staged double strip: H, row max, arg-max of 2C rows; first row, base */

/* steps of a solo task (also the step dimension of its traceback block): the high half of the last double strip ends at
 * 2*(nds-1) + 1 + last column */
template<int C>
K2A_FN size_t k2a_solo_steps(int qlen, int tlen, int w)
{
	const int nds = (tlen + 2 * C - 1) / (2 * C);
	return nds > 0 ? (size_t)(2 * (nds - 1) + 2 + k2a_min(qlen - 1, tlen - 1 + w)) : 0;
}

template<int C, bool DUAL, int MODE = K2A_MODE_SCORE, bool TN = false>      /* TN: K2aLanePk */
struct K2aLaneSolo {
	enum { G = 64, TBWORDS = C / 2 };
	int qlen, tlen, tlen_full, w, nds;      /* nds = double strips */
	const uint8_t *qa, *ta;
	int gl, D, i0, koff, Dnext, knext, koff_next;
	int kfin, kfinA, kd, kB0, rowsA_m1, rowsB_m1, wupA;   /* kB0: first step of the high half (its first in-band column) */
	k2a_pk hout, eout, e2out, hd0, hu_prev;
	int baseA, baseB;                      /* absolute (row-biased) scores the low / high half's values are relative to */
	k2a_pk delta;                           /* low half: base of the lane above's high half minus baseA; high half: baseA - baseB */
	uint32_t qb, qw;                        /* { query code at column jj, at column jj-1 }; the dword of the current four steps */
	uint32_t tnA[(C + 3) / 4], tnB[(C + 3) / 4], qn0;   /* prefetched for the NEXT double strip: its target codes, its first query dword */
	int qn_sh;
	const uint32_t *cptab;                  /* column profiles (K2aLanePk): cpA = cp[code at column jj] for the low half, cpB = what the low half had one step ago */
	uint32_t cpA, cpB;
	uint32_t seen;                          /* OR of every TARGET code dword this lane used (K2aLanePk::seen: a flat plan's unscanned bytes; code >= 4 = wildcard -> the host re-runs the pair) */
	uint32_t hasn;                          /* target wildcard rows, as in K2aLanePk (K2aScoring.pk_tn1) */
	bool wn;
	K2A_FN void note_codes(uint32_t a) { seen |= a; }
	K2A_FN bool saw_wildcard() const { return (seen & 0xfcfcfcfcu) != 0; }
	k2a_pk hl[C], f[C], f2[DUAL ? C : 1], rmax[C], rmj[C], tc[C], hsave[C];      /* tc: per row the v_perm_b32 selector { t(row c), 0x0c, 4 + t(row C + c), 0x0c } */

	K2A_FN static int first_col(int D_, int w_) { return k2a_max(0, D_ * 2 * C - w_); }
	K2A_FN void schedule_next()
	{
		koff_next = 2 * Dnext;
		knext = Dnext < nds ? koff_next + first_col(Dnext, w) : K2A_KNONE;
	}

	/* What the next double strip's do_init needs from memory, asked for one strip early (about a thousand steps): with one
	 * wavefront per SIMD nobody else covers a load's latency, and a strip starts somewhere in the wavefront every 16 steps
	 * (profiles/r3_solo_experiments.txt: 26 % of the kernel's time was s_waitcnt).  hipcc still waits for these right behind the
	 * loads (they are set under a condition and live around the step loop: ksw2_lane_pk.h, k2a_load_early), but the wait then
	 * overlaps the rest of do_init instead of standing in front of it.  Target codes of its 2C rows and the query
	 * dword of the group of four steps its first step falls in (shifted when column 0 comes later than the group's first step). */
	K2A_FN void prefetch_next()
	{
		qn_sh = 0;
		if (Dnext < nds) {
			const uint8_t *tp = ta + (size_t)Dnext * 2 * C;
#pragma unroll
			for (int x = 0; x < (C + 3) / 4; ++x) { tnA[x] = k2a_load_early(tp + 4 * x); tnB[x] = k2a_load_early(tp + C + 4 * x); }
			const int j = (knext & ~3) - koff_next;                    /* < 0: the strip's column 0 comes -j steps into the group */
			qn_sh = 8 * k2a_min(k2a_max(-j, 0), 3);
			qn0 = k2a_load_early(qa + k2a_min(k2a_max(j, 0), qlen - 1));
		}
	}

	K2A_FN void setup(const K2aPair &pr, const uint8_t *seq, int lane, bool valid, const uint32_t *cptab_)
	{
		cptab = cptab_; cpA = cpB = 0;
		qlen = pr.qlen; tlen = pr.tlen; tlen_full = pr.tlen_full; w = pr.w;
		qa = seq + pr.qoff; ta = seq + pr.toff;
		nds = valid ? (tlen + 2 * C - 1) / (2 * C) : 0;
		gl = lane;
		D = -1; i0 = 0; koff = 0; kfin = kfinA = K2A_KNONE; kd = 0; kB0 = 0; rowsA_m1 = rowsB_m1 = -1; wupA = w;
		Dnext = gl;
		schedule_next();
#pragma unroll
		for (int x = 0; x < (C + 3) / 4; ++x) tnA[x] = tnB[x] = 0;
		qn0 = 0; seen = 0; hasn = 0; wn = false;
		prefetch_next();
		const k2a_pk neg = k2a_pku(K2A_NEG16);
		hout = eout = e2out = hd0 = hu_prev = neg; qb = 0; qw = 0; baseA = baseB = 0; delta = 0;
#pragma unroll
		for (int c = 0; c < C; ++c) { hl[c] = f[c] = rmax[c] = hsave[c] = neg; rmj[c] = 0; tc[c] = K2A_TSEL_BASE; if (DUAL) f2[c] = neg; }
		if (!DUAL) f2[0] = 0;
	}

	/* last step of the alignment: the high half of the last double strip (or its low half when the high half is empty) */
	K2A_FN int last_step() const
	{
		if (nds <= 0) return -1;
		const int Dl = nds - 1, rB = tlen - 1 - (Dl * 2 * C + C);
		return rB >= 0 ? 2 * Dl + 1 + k2a_min(qlen - 1, tlen - 1 + w) : 2 * Dl + k2a_min(qlen - 1, tlen - 1 + w);
	}
	K2A_FN bool need_init(int k) const { return k == knext; }
	K2A_FN bool need_fin(int k) const { return k == kfin; }
	K2A_FN bool need_save(int k) const { return k == kfinA; }

	/* bs = base of the high half of the double strip above (rotated in by the kernel) */
	K2A_FN void do_init(const K2aScoring &sc, int bs)
	{
		D = Dnext; i0 = D * 2 * C; koff = koff_next;
		const int i0b = i0 + C;
		rowsA_m1 = k2a_min(C - 1, tlen - 1 - i0);
		rowsB_m1 = k2a_max(-1, k2a_min(C - 1, tlen - 1 - i0b));
		const int jeA = k2a_min(qlen - 1, k2a_min(i0 + C - 1, tlen - 1) + w);
		const int jeB = k2a_min(qlen - 1, k2a_min(i0b + C - 1, tlen - 1) + w);
		kfinA = koff + jeA;
		kfin = rowsB_m1 >= 0 ? koff + 1 + jeB : kfinA;
		kd = koff + i0;
		kB0 = rowsB_m1 >= 0 ? koff + 1 + k2a_max(0, i0b - w) : K2A_KNONE;
		wupA = w + (D == 0 ? 1 : 0);
		const int js = k2a_max(0, i0 - w);
		const k2a_pk neg = k2a_pku(K2A_NEG16);
		/* target codes of the 2C rows as selectors into the step's column profiles (ksw2_lane_pk.h), low half = rows i0.., high
		 * half = rows i0+C..: four rows of both halves per pair of dwords */
		hasn = 0;
#pragma unroll
		for (int c4 = 0; c4 < C; c4 += 4) {
			const uint32_t da = tnA[c4 / 4], db = tnB[c4 / 4];             /* prefetch_next(), one strip ago */
			note_codes(da | db);                                           /* (rows past the target's end read the arena's next bytes: at worst a needless re-run) */
			if (TN && sc.pk_tn1 && ((da | db) & 0x04040404u)) {                 /* a wildcard among these rows' target codes (K2aLanePk::do_init) */
				hasn = 1;
#pragma unroll
				for (int r = 0; r < 4; ++r) tc[c4 + r] = k2a_tsel_wild(k2a_byte_pair(da, db, r));
			} else {
#pragma unroll
				for (int r = 0; r < 4; ++r) tc[c4 + r] = k2a_byte_pair(da, db, r) + K2A_TSEL_BASE;
			}
		}
#pragma unroll
		for (int c = 0; c < C; ++c) { hl[c] = neg; f[c] = neg; if (DUAL) f2[c] = neg; rmax[c] = neg; rmj[c] = 0; }
		/* base = the diagonal input of the low half's first cell */
		const int hcorner = k2a_border<DUAL>(sc, i0) + sc.e * (i0 - 1);
		const int nb = js == 0 ? hcorner : bs + k2a_pk_hi(k2a_ofs_off(hu_prev));          /* what arrived: the high half of the lane above */
		delta = D == 0 ? 0u : ((uint32_t)(bs - nb) & 0xffffu);                            /* high half: 0 until start_high() */
		baseA = nb;
		if (i0 <= w) {                                          /* rows starting at column 0: virtual column -1 (ksw2_extz.c:43-44) */
#pragma unroll
			for (int c = 0; c < C; ++c) {
				const int ha = k2a_border<DUAL>(sc, i0 + c + 1) + sc.e * (i0 + c);
				const uint32_t va = i0 + c <= w ? k2a_h16(ha - baseA) : k2a_h16(K2A_NEG16);
				hl[c] = k2a_pair16(va, neg >> 16);
				const k2a_pk fl = k2a_pk_sub(hl[c], k2a_pk2(sc.q + sc.e));
				f[c] = k2a_pair16(i0 + c <= w ? fl & 0xffffu : neg & 0xffffu, neg >> 16);
				if (DUAL) {
					const k2a_pk fl2 = k2a_pk_sub(hl[c], k2a_pk2(sc.q2 + sc.e2));
					f2[c] = k2a_pair16(i0 + c <= w ? fl2 & 0xffffu : neg & 0xffffu, neg >> 16);
				}
			}
		}
		/* diagonal input of the low half's first cell: 0 by construction of the base.  The high half takes the low half's bottom
		 * row one step later: start_high() gives it its own base just before its first step. */
		hd0 = k2a_pair16(k2a_h16(0), neg >> 16);
		qw = qn0 << qn_sh;                                      /* this strip's first group of query codes */
		Dnext += G;
		schedule_next();
		prefetch_next();
	}

	K2A_FN bool need_init_high(int k) const { return k == kB0; }

	/* The high half's first step.  Its base is the diagonal input of its first cell: H(i0b - 1, first column - 1), which is what
	 * the low half's bottom row handed over one step ago (hd0's high half, still relative to baseA: delta's high half is 0
	 * since do_init) -- or the virtual column -1 when the rows start at column 0, whose H / F / F~ are loaded here too (any
	 * earlier and the steps in between, which run the high half dead, would wipe them). */
	K2A_FN void start_high(const K2aScoring &sc)
	{
		const int i0b = i0 + C;
		if (i0b <= w) {
			const uint32_t negh = (uint32_t)k2a_h16(K2A_NEG16);
			baseB = k2a_border<DUAL>(sc, i0b) + sc.e * (i0b - 1);
#pragma unroll
			for (int c = 0; c < C; ++c) {
				const bool in = i0b + c <= w;
				const int hb = k2a_border<DUAL>(sc, i0b + c + 1) + sc.e * (i0b + c) - baseB;
				const uint32_t vb = in ? k2a_h16(hb) : negh;
				const uint32_t fb = in ? k2a_h16(hb - (sc.q + sc.e)) : negh;
				hl[c] = (hl[c] & 0xffffu) | (vb << 16);
				f[c] = (f[c] & 0xffffu) | (fb << 16);
				if (DUAL) {
					const uint32_t fb2 = in ? k2a_h16(hb - (sc.q2 + sc.e2)) : negh;
					f2[c] = (f2[c] & 0xffffu) | (fb2 << 16);
				}
			}
		} else baseB = baseA + k2a_pk_hi(k2a_ofs_off(hd0));
		hd0 = (hd0 & 0xffffu) | ((uint32_t)k2a_h16(0) << 16);
		delta = (delta & 0xffffu) | (((uint32_t)(baseA - baseB) & 0xffffu) << 16);
	}

	/* double strip 0, low half only: the cells above row 0 are the virtual row -1 (ksw2_extz.c:32-35) */
	K2A_FN void top_inputs(const K2aScoring &sc, int k, k2a_pk &hin, k2a_pk &ein, k2a_pk &e2in) const
	{
		if (D == 0) {
			const int hb = k2a_border<DUAL>(sc, k - koff + 1) - baseA;
			const uint32_t h0 = k2a_h16(hb - sc.e), e0 = k2a_h16(hb - (sc.q + sc.e));
			const uint32_t e20 = k2a_h16(hb - (sc.q2 + sc.e2));
			hin = (hin & 0xffff0000u) | h0; ein = (ein & 0xffff0000u) | e0; e2in = (e2in & 0xffff0000u) | e20;
		}
	}

	/* One step: column jj = k - koff for the low half's rows, jj - 1 for the high half's.  hin / ein / e2in: low half = bottom
	 * row of the lane above's high half, high half = this lane's own low-half bottom row of the previous step (both already
	 * shifted by delta).  Rows in chunks of CH as in K2aLanePk::step. */
	K2A_FN bool step(const K2aScoring &sc, int k, k2a_pk hin, k2a_pk ein, k2a_pk e2in, uint32_t *tbw)
	{
		const int ddA = k - kd, ddB = ddA - C - 1;
		const k2a_pk neg = k2a_pku(K2A_NEG16);
		const k2a_pk gq = k2a_pk2(sc.q), ge = k2a_pk2(sc.e), gq2 = k2a_pk2(sc.q2), ge2 = k2a_pk2(sc.e2), de2 = k2a_pk2(sc.e2 - sc.e);
		const k2a_pk bias = k2a_pk2(sc.pk_smax + sc.e);        /* largest score + row-bias step; the rows subtract their penalties from it */
		/* the cell above is outside the band: per half */
		const k2a_pk cut = k2a_pair16(ddA >= wupA ? 0xffffu : 0u, ddB >= w ? 0xffffu : 0u);
		k2a_pk e = k2a_pk_sel(cut, neg, ein), e2 = DUAL ? k2a_pk_sel(cut, neg, e2in) : 0u;
		/* the low half keeps stepping after its last column while the high half finishes, the high half is carried along before
		 * its first column: no live rows there */
		const int loA = k2a_max(0, ddA - w), hiA = k2a_min(rowsA_m1, ddA + w), cntA = k <= kfinA ? k2a_max(hiA - loA + 1, 0) : 0;
		const int loB = k2a_max(0, ddB - w), hiB = k2a_min(rowsB_m1, ddB + w), cntB = k >= kB0 ? k2a_max(hiB - loB + 1, 0) : 0;
		const uint32_t liveA = ((1u << cntA) - 1u) << (loA & 31), liveB = ((1u << cntB) - 1u) << (loB & 31);   /* lo >= 32 only with cnt = 0 */
		/* row c of the low half at bit c, of the high half at bit 16 + c: ONE 32-bit shift by 15 - c puts both at their halves'
		 * sign bits (what the low half's higher rows spill into the high half stays below bit 31) */
		const uint32_t lv = liveA | (liveB << 16);
		const k2a_pk jjpk = k2a_pair16((uint32_t)(k - koff) & 0xffffu, (uint32_t)(k - koff - 1) & 0xffffu);
		/* score only: all rows' candidates at once -- these tasks run at one or two wavefronts per SIMD, where the longer
		 * independent stretch is worth more than the registers (1 024 x 10 k x 10 k: 6.74 ms against 7.07 ms) */
		constexpr int CH = (MODE != K2A_MODE_SCORE && C % 4 == 0) ? 4 : C;
		k2a_pk dprev = 0, above_old = hd0;
#pragma unroll
		for (int c0 = 0; c0 < C; c0 += CH) {
			k2a_pk cand[CH];
			const k2a_pk last_old = hl[c0 + CH - 1];
#pragma unroll
			for (int r = 0; r < CH; ++r) {
				const int c = c0 + r;
				cand[r] = k2a_sub32(k2a_add32(r == 0 ? above_old : hl[c - 1], bias), k2a_perm(cpB, cpA, tc[c]));
			}
			if (TN && wn) {                                          /* a target wildcard row somewhere in the wavefront (K2aLanePk::step) */
#pragma unroll
				for (int r = 0; r < CH; ++r) cand[r] = k2a_tn_fix(cand[r], tc[c0 + r], sc.pk_tn1);
			}
			above_old = last_old;
			if (CH < C) K2A_SCHED_FENCE();
#pragma unroll
			for (int r = 0; r < CH; ++r) {
				const int c = c0 + r;
				const k2a_pk fc = f[c];
				k2a_pk h = cand[r], s1 = 0, s2 = 0, s3 = 0, s4 = 0, x1 = 0, x2 = 0, x3 = 0, x4 = 0;
				if (MODE == K2A_MODE_SCORE) {
					h = k2a_pk_max3u(h, e, fc);                    /* v_pk_maximum3_f16 on offset-form patterns (ksw2_lane_pk.h) */
					if (DUAL) h = k2a_pk_max3u(h, e2, f2[c]);
				} else if (MODE == K2A_MODE_LEFT) {            /* negative = the gap state wins (K2aLanePk::step, k2a_dir_flags) */
					s1 = k2a_pk_sub(h, e);  h = k2a_pk_maxu(h, e);
					s2 = k2a_pk_sub(h, fc); h = k2a_pk_maxu(h, fc);
					if (DUAL) { s3 = k2a_pk_sub(h, e2); h = k2a_pk_maxu(h, e2); s4 = k2a_pk_sub(h, f2[c]); h = k2a_pk_maxu(h, f2[c]); }
				} else {                                       /* right-aligned: negative = the gap state does NOT win */
					s1 = k2a_pk_sub(e, h);  h = k2a_pk_maxu(h, e);
					s2 = k2a_pk_sub(fc, h); h = k2a_pk_maxu(h, fc);
					if (DUAL) { s3 = k2a_pk_sub(e2, h); h = k2a_pk_maxu(h, e2); s4 = k2a_pk_sub(f2[c], h); h = k2a_pk_maxu(h, f2[c]); }
				}
				h = k2a_pk_sel(k2a_pk_sign(lv << (15 - c)), h, neg);                 /* live mask of row c, per half */
				if (!DUAL && MODE == K2A_MODE_RIGHT) rmj[c] = k2a_pk_selv(k2a_pk_sign(k2a_pk_sub(rmax[c], h)), jjpk, rmj[c]);
				else rmj[c] = k2a_pk_selv(k2a_pk_sign(k2a_pk_sub(h, rmax[c])), rmj[c], jjpk);
				rmax[c] = k2a_pk_maxu(rmax[c], h);
				const k2a_pk t = k2a_sub32(h, gq);
				if (MODE == K2A_MODE_LEFT) { x1 = k2a_pk_sub(t, e); x2 = k2a_pk_sub(t, fc); }
				else if (MODE == K2A_MODE_RIGHT) { x1 = k2a_pk_sub(e, t); x2 = k2a_pk_sub(fc, t); }
				e = k2a_pk_maxu(e, t);
				f[c] = k2a_sub32(k2a_pk_maxu(fc, t), ge);
				if (DUAL) {
					const k2a_pk t2 = k2a_sub32(h, gq2);
					if (MODE == K2A_MODE_LEFT) { x3 = k2a_pk_sub(t2, e2); x4 = k2a_pk_sub(t2, f2[c]); }
					else if (MODE == K2A_MODE_RIGHT) { x3 = k2a_pk_sub(e2, t2); x4 = k2a_pk_sub(f2[c], t2); }
					e2 = k2a_pk_sub(k2a_pk_maxu(e2, t2), de2);
					f2[c] = k2a_sub32(k2a_pk_maxu(f2[c], t2), ge2);
				}
				if (MODE != K2A_MODE_SCORE) {                  /* one flag byte per cell: byte 0 = the low half's row, byte 1 = the high half's */
					const uint32_t fl = k2a_dir_flags<DUAL, false, MODE == K2A_MODE_RIGHT>(s1, s2, s3, s4, x1, x2, x3, x4);
					if (c & 1) tbw[c >> 1] = k2a_perm(fl, dprev, 0x05040100u);
					else dprev = fl;
				}
				hl[c] = h;
			}
			if (CH < C) K2A_SCHED_FENCE();
		}
		hd0 = hin;
		hout = hl[C - 1]; eout = e; e2out = e2;
		return (liveA | liveB) != 0;
	}

	/* Query codes, four steps per unaligned dword (K2aLanePk::load_query_group): the group of steps kg .. kg+3 under the column
	 * offset `koff_use`; nothing here waits for the load.  The kernel asks for a group four steps before its first step; a double
	 * strip that starts in between brings its own first group (prefetch_next). */
	K2A_FN void load_query_group(int kg, int koff_use, uint32_t &a) const
	{
		const int jc = k2a_min(k2a_max(kg - koff_use, 0), qlen - 1);
		a = k2a_load_early(qa + jc);
	}
	/* the codes of step kg + kk: the low half's from the group, the high half's = what the low half had one step ago -- and the
	 * same for their column profiles: ONE table look-up per step */
	K2A_FN void advance_query(int kk)
	{
#if defined(__HIP_DEVICE_COMPILE__)
		qb = __builtin_amdgcn_perm(qb, qw, 0x0c040c00u + (uint32_t)kk);
#else
		qb = ((qw >> (8 * kk)) & 0xffu) | ((qb & 0xffu) << 16);
#endif
		cpB = cpA; cpA = cptab[qb & 7u];
	}

	/* the low half is done C+1 steps before the high half and keeps stepping over dead cells: keep its last-column H */
	K2A_FN void save_low() {
#pragma unroll
		for (int c = 0; c < C; ++c) hsave[c] = hl[c];
	}

	/* rows of the finished double strip in row order into LDS: [0,2C) H(i, last column), [2C,4C) row max, [4C,6C) arg-max */
	K2A_FN void stage_rows(uint32_t *rowbuf) const
	{
#pragma unroll
		for (int c = 0; c < C; ++c) {
			const k2a_pk hA = k2a_ofs_off(rowsB_m1 >= 0 ? hsave[c] : hl[c]), hB = k2a_ofs_off(hl[c]), m = k2a_ofs_off(rmax[c]);
			rowbuf[c] = hA & 0xffffu; rowbuf[C + c] = hB >> 16;
			rowbuf[2 * C + c] = m & 0xffffu; rowbuf[3 * C + c] = m >> 16;
			rowbuf[4 * C + c] = rmj[c] & 0xffffu; rowbuf[5 * C + c] = rmj[c] >> 16;
		}
	}

	/* the scalar reference's per-row epilogue over the 2C rows (K2aLane::do_fin) */
	K2A_FN void do_fin_seq(const K2aScoring &sc, K2aBook *b, int zdrop, const uint32_t *rowbuf)
	{
		const int zslope = DUAL ? sc.e2 : sc.e;
		int bmax = b->max, bmax_t = b->max_t, bmax_q = b->max_q, bmqe = b->mqe, bmqe_t = b->mqe_t;
		int bmte = b->mte, bmte_q = b->mte_q, bscore = b->score, bdrop = b->dropped, brows = b->rows;
#pragma nounroll
		for (int c = 0; c < 2 * C; ++c) {
			const int i = i0 + c;
			if (i < tlen && !bdrop) {
				const bool reach = i + w >= qlen - 1;
				const int unb = (c < C ? baseA : baseB) - sc.e * i;
				const int hend = (int)(int16_t)rowbuf[c] + unb, H = (int)(int16_t)rowbuf[2 * C + c] + unb;
				const int j = (int)(uint16_t)rowbuf[4 * C + c];
				if (reach && hend > bmqe) { bmqe = hend; bmqe_t = i; }
				if (i == tlen_full - 1) { bmte = H; bmte_q = j; }
				if (H > bmax) { bmax = H; bmax_t = i; bmax_q = j; }
				else if (i >= bmax_t && j >= bmax_q) {
					const int dt = i - bmax_t, dq = j - bmax_q;
					const int skew = dt > dq ? dt - dq : dq - dt;
					if (zdrop >= 0 && bmax - H > zdrop + skew * zslope) bdrop = 1;
				}
				if (!bdrop && i == tlen_full - 1 && reach) bscore = hend;
				brows = i + 1;
			}
		}
		b->max = bmax; b->max_t = bmax_t; b->max_q = bmax_q; b->mqe = bmqe; b->mqe_t = bmqe_t;
		b->mte = bmte; b->mte_q = bmte_q; b->score = bscore; b->dropped = bdrop; b->rows = brows;
		end_strip();
	}

	/* shortcut of K2aLanePk::fin_fast over the 2C rows of the double strip (low-half rows come first) */
	K2A_FN bool fin_fast(const K2aScoring &sc, K2aBook *b, int zdrop)
	{
		if (i0 + 2 * C >= tlen || i0 + 2 * C - 1 + w >= qlen - 1) return false;
		/* rows compare without their bias: low half v_c = rmax[c] - e*c, high half rmax[c] - e*(C+c) */
		const k2a_pk hb = k2a_pair16(0u, (uint32_t)(sc.e * C) & 0xffffu);
		k2a_pk m = k2a_pk_sub(k2a_ofs_off(rmax[0]), hb), mn = m, arg = 0, argj = rmj[0];
#pragma unroll
		for (int c = 1; c < C; ++c) {
			const k2a_pk v = k2a_pk_sub(k2a_pk_sub(k2a_ofs_off(rmax[c]), hb), k2a_pk2(sc.e * c));
			const k2a_pk gt = k2a_pk_sign(k2a_pk_sub(m, v));
			arg = k2a_pk_sel(gt, k2a_pk2(c), arg);
			argj = k2a_pk_sel(gt, rmj[c], argj);
			m = k2a_pk_max(m, v);
			mn = k2a_pk_min(mn, v);
		}
		const int offA = baseA - sc.e * i0, offB = baseB - sc.e * i0;
		const int MA = k2a_pk_lo(m) + offA, MB = k2a_pk_hi(m) + offB;
		const int mnn = k2a_min(k2a_pk_lo(mn) + offA, k2a_pk_hi(mn) + offB);
		const int M = k2a_max(MA, MB), bm = b->max;
		if (b->dropped) { end_strip(); return true; }
		if (zdrop >= 0 && k2a_max(bm, M) - mnn > zdrop) return false;
		if (M > bm) {
			if (MB > MA) { b->max = MB; b->max_t = i0 + C + k2a_pk_hi(arg); b->max_q = (int)(argj >> 16); }
			else { b->max = MA; b->max_t = i0 + k2a_pk_lo(arg); b->max_q = (int)(argj & 0xffffu); }
		}
		b->rows = i0 + 2 * C;
		end_strip();
		return true;
	}

	K2A_FN void end_strip() { D = -1; kfin = kfinA = K2A_KNONE; rowsA_m1 = rowsB_m1 = -1; hasn = 0; }
};

/* Traceback walk over a solo task's direction bytes: cell (i, j) is byte 2c + half of the word written at step
 * j + 2D + half by lane D mod 64, with D = i / 2C, half = (i mod 2C) / C, c = i mod C (lane-major block as everywhere). */
template<int C>
K2A_FN int k2a_trace_solo(const uint8_t *tb, int i, int j, uint32_t *out, int qlen, int tlen, int w)
{
	enum { WB = 2 * C, AHEAD = 8 };
	int n = 0, state = 0;
	uint32_t last_op = 0xffffffffu, run = 0;
	const size_t nsteps = k2a_solo_steps<C>(qlen, tlen, w);
	/* The walk is a chain of dependent loads and most moves are diagonal (k2a_trace_walk, ksw2_lane.h): the bytes of the next AHEAD
	 * cells on the diagonal -- one step and one row back each, i.e. WB + 2 bytes lower in the lane's run, as long as the row stays
	 * in the same half of the double strip -- are requested together and consumed while the path really is diagonal. */
	while (i >= 0 && j >= 0) {
		const int D = i / (2 * C), r = i - D * 2 * C, half = r >= C ? 1 : 0, c = r - half * C;
		const uint8_t *p = tb + k2a_tb_word((size_t)(j + 2 * D + half), D % 64, nsteps, 64, WB) + 2 * c + half;
		const int nq = k2a_min(k2a_min(AHEAD, c + 1), k2a_min(i, j) + 1);
		uint32_t bq[AHEAD];
#pragma unroll
		for (int k = 0; k < AHEAD; ++k) bq[k] = p[-(ptrdiff_t)k2a_min(k, nq - 1) * (WB + 2)];
		bool diagonal = true;
#pragma unroll
		for (int k = 0; k < AHEAD; ++k) {
			if (k >= nq || !diagonal) break;
			const uint32_t d = k2a_flags_decode(bq[k]);             /* flag byte -> the reference's direction byte */
			if (state == 0) state = d & 7;
			else if (!((d >> (state + 2)) & 1)) state = 0;
			if (state == 0) state = d & 7;
			uint32_t op;
			if (state == 0) { op = 0; --i; --j; }
			else if (state == 1 || state == 3) { op = 2; --i; diagonal = false; }
			else { op = 1; --j; diagonal = false; }
			if (op == last_op) ++run;
			else { if (run) out[n++] = run << 4 | last_op; last_op = op; run = 1; }
		}
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
