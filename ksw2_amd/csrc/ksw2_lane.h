/*
 * ksw2_lane.h -- per-lane code of the MI355X banded extension kernels (extz2 / extd2 / gg2 hot path).
 *
 * Replaces the per-diagonal inner loops of ksw_extz2_sse (ksw2_extz2_sse.c:101-289) and
 * ksw_extd2_sse (ksw2_extd2_sse.c:131-387); results follow the exact-band / row-wise Z-drop contract
 * of the scalar ksw_extz / ksw_extd (ksw2_extz.c:38-125, ksw2_extd.c:44-165).
 *
 * Mapping: one alignment per lane group of G lanes (G = 64: one per wavefront,
 * G = 16: four per wavefront).  Target rows are cut into strips of C consecutive rows; strip S is owned
 * by lane S mod G and walks the query one column per step, skewed by one step per strip:
 *
 *        step k, strip S  ->  column jj = k - S, rows S*C .. S*C+C-1
 *
 * so all active lanes sit on one anti-diagonal of the (strip, column) grid.  Inside a lane the C rows
 * are a register-resident chain (E and the diagonal H flow down the rows without any lane traffic);
 * between lanes only the bottom row's (H, E[, E~]) moves, one lane up per step (DPP rotate).  Each row
 * keeps int32 H(i, jj-1), F, running row maximum and arg-max in registers, so the scalar reference's
 * per-row bookkeeping (mqe, mte, Z-drop, score) falls out in row order when a strip finishes.
 *
 * The file is plain C++: the HIP kernel (ksw2_kernels.hip) instantiates it per thread; tests/sim
 * instantiates the same code on the host, 64 "lanes" in lock step, to check the schedule without a GPU.
 */
#ifndef KSW2_LANE_H_
#define KSW2_LANE_H_

#include <stdint.h>

#if defined(__HIP_DEVICE_COMPILE__) || defined(__HIPCC__)
#define K2A_FN __device__ __forceinline__
#else
#define K2A_FN inline
#endif

#include "ksw2_types.h"

K2A_FN int k2a_min(int a, int b) { return a < b ? a : b; }
K2A_FN int k2a_max(int a, int b) { return a > b ? a : b; }
K2A_FN int k2a_max3(int a, int b, int c) { return k2a_max(k2a_max(a, b), c); }

/* H on the virtual row -1 / column -1 at distance k >= 0 from the origin */
template<bool DUAL>
K2A_FN int k2a_border(const K2aScoring &sc, int k)
{
	int a = -(sc.q + k * sc.e);
	if (DUAL) { int b = -(sc.q2 + k * sc.e2); a = a > b ? a : b; }
	return k <= 0 ? 0 : a;
}

K2A_FN void k2a_book_reset(K2aBook *b)
{
	b->max = 0; b->max_t = b->max_q = b->mqe_t = b->mte_q = -1;
	b->mqe = b->mte = b->score = K2A_NEG; b->dropped = 0; b->rows = 0; b->inexact = 0;
}

/* traceback cell encodings (our own layout; the walk in k2a_trace_pair() is the only reader)
 *   single gap (4 bits): bits 0-1 winner {0 diag, 1 E, 2 F}, bit 2 E continues, bit 3 F continues
 *   dual gap   (8 bits): bits 0-2 winner {0..4}, bits 3-6 E / F / E~ / F~ continue   (as ksw2.h:125-128) */
template<bool DUAL> struct K2aTb { enum { BITS = DUAL ? 8 : 4 }; };

/* LROW: the per-row maxima and arg-max columns live in LDS ([2][C][64] dwords per wavefront, `lrow` = this lane's column)
 * instead of registers -- for kernels that are a few registers over an occupancy step (ksw2_shim_hip.hip) */
#define K2A_LROW_WORDS(C) (2 * (C) * 64)

template<int G, int C, bool DUAL, int MODE, bool LROW = false>
struct K2aLane {
	enum { TBBITS = K2aTb<DUAL>::BITS, TBWORDS = (C * TBBITS + 31) / 32 };

	/* group-uniform */
	int qlen, tlen, tlen_full, w, nstrips;
	const uint8_t *qry, *tgt;
	/* schedule: a lane sees column jj = k - koff at step k while it owns strip S */
	int gl, S, i0, je, koff, Snext, knext, koff_next;
	int kfin, kd, rows_m1, wup;   /* last step of the strip, koff + i0, live-row clamp (-1 = no strip), band reach upwards */
	/* systolic ports: what this lane produced for the strip below at the previous step */
	int hout, eout, e2out;
	int hd0;                      /* H(i0-1, jj-1) */
	int hu_prev;                  /* H(i0-1, jj-1) candidate received last step */
	int qb;                       /* query code for this step's column */
	/* rows */
	int hl[C], f[C], f2[DUAL ? C : 1], rmax_[LROW ? 1 : C], rmj_[LROW ? 1 : C];
	int *lrow;
	K2A_FN int &rmaxr(int c) { return LROW ? lrow[(0 * C + c) * 64] : rmax_[LROW ? 0 : c]; }
	K2A_FN int &rmjr(int c) { return LROW ? lrow[(1 * C + c) * 64] : rmj_[LROW ? 0 : c]; }
	uint32_t P[C];
	uint32_t tbp[(C + 3) / 4];    /* packed target codes (only needed against the query wildcard) */
	uint32_t tnext[(C + 3) / 4];  /* prefetched target codes of strip Snext */

	K2A_FN static int first_col(int S_, int w_) { return k2a_max(0, S_ * C - w_); }

	/* resident-band schedule: strip S is skewed by S steps, a lane moves on to strip S+G when done */
	K2A_FN void schedule_next_resident()
	{
		koff_next = Snext;
		knext = Snext < nstrips ? koff_next + first_col(Snext, w) : K2A_KNONE;
	}

	K2A_FN void load_tnext()
	{
		/* target arena is padded, any strip start is readable for C bytes; rows >= tlen are never live */
		if (Snext < nstrips) {
			const uint32_t *p = (const uint32_t*)(tgt + (size_t)Snext * C);
#pragma unroll
			for (int x = 0; x < (C + 3) / 4; ++x) tnext[x] = p[x];
		}
	}

	K2A_FN void setup(const K2aPair &pr, const uint8_t *seq, int lane_in_group, bool valid)
	{
		qlen = pr.qlen; tlen = pr.tlen; tlen_full = pr.tlen_full; w = pr.w;
		qry = seq + pr.qoff; tgt = seq + pr.toff;
		nstrips = valid ? (tlen + C - 1) / C : 0;
		gl = lane_in_group;
		S = -1; i0 = 0; je = -1; koff = 0; kfin = K2A_KNONE; kd = 0; rows_m1 = -1; wup = w;
		Snext = gl;
		schedule_next_resident();
		hout = eout = e2out = K2A_NEG; hd0 = K2A_NEG; hu_prev = K2A_NEG; qb = 0;
#pragma unroll
		for (int c = 0; c < C; ++c) { hl[c] = f[c] = K2A_NEG; rmaxr(c) = K2A_NEG; rmjr(c) = 0; P[c] = 0; if (DUAL) f2[c] = K2A_NEG; }
		if (!DUAL) f2[0] = 0;
#pragma unroll
		for (int x = 0; x < (C + 3) / 4; ++x) tbp[x] = tnext[x] = 0;
		load_tnext();
	}

	/* last step at which any lane of this alignment computes a live cell */
	K2A_FN int last_step() const
	{
		int Sl = nstrips - 1;
		return nstrips > 0 ? Sl + k2a_min(qlen - 1, tlen - 1 + w) : -1;
	}

	K2A_FN bool need_init(int k) const { return k == knext; }

	/* Generation-serial schedule (bands too wide to stay resident): generation g =
	 * strips g*G .. g*G+G-1, all columns of their band [jlo, jhi], then the next generation; lane l is skewed
	 * by l steps.  The bottom row of a generation reaches the next one through a boundary buffer in HBM. */
	K2A_FN void begin_generation(int g, int jlo)
	{
		S = -1; je = -1; kfin = K2A_KNONE; rows_m1 = -1;
		Snext = g * G + gl;
		koff_next = gl - jlo;
		knext = Snext < nstrips ? koff_next + first_col(Snext, w) : K2A_KNONE;
		hout = eout = e2out = K2A_NEG; hu_prev = K2A_NEG;
		load_tnext();
	}

	/* start strip Snext at step k (its first column): row state from the virtual column -1 or -inf */
	template<bool RESIDENT>
	K2A_FN void do_init(const K2aScoring &sc, const uint32_t *ptab)
	{
		S = Snext; i0 = S * C; koff = koff_next;
		je = k2a_min(qlen - 1, k2a_min(i0 + C - 1, tlen - 1) + w);
		kfin = koff + je;
		kd = koff + i0;
		rows_m1 = k2a_min(C - 1, tlen - 1 - i0);
		wup = w + (S == 0 ? 1 : 0);                        /* the virtual row -1 reaches one column further (E(0,w) exists) */
		const int js = k2a_max(0, i0 - w);
#pragma unroll
		for (int x = 0; x < (C + 3) / 4; ++x) tbp[x] = tnext[x];
#pragma unroll
		for (int c = 0; c < C; ++c) {
			const uint32_t tb = (tbp[c >> 2] >> (8 * (c & 3))) & 0xff;
			P[c] = ptab[tb < 4 ? tb : 4];                     /* score profile of this row's target code (LDS table) */
			hl[c] = K2A_NEG; f[c] = K2A_NEG; if (DUAL) f2[c] = K2A_NEG;
			rmaxr(c) = K2A_NEG; rmjr(c) = 0;
		}
		if (i0 <= w) {                                      /* some rows start at column 0: virtual column -1 */
#pragma unroll
			for (int c = 0; c < C; ++c) {                    /* ksw2_extz.c:43-44, ksw2_extd.c:49-52 */
				const int hb = k2a_border<DUAL>(sc, i0 + c + 1);
				if (i0 + c <= w) {
					hl[c] = hb; f[c] = hb - (sc.q + sc.e);
					if (DUAL) f2[c] = hb - (sc.q2 + sc.e2);
				}
			}
		}
		if (js == 0) hd0 = k2a_border<DUAL>(sc, i0);       /* H(i0-1,-1); 0 at the origin */
		else hd0 = hu_prev;                               /* H(i0-1, js-1), received one step ago */
		if (RESIDENT) {
			Snext += G;
			schedule_next_resident();
			load_tnext();
		} else knext = K2A_KNONE;
	}

	/* first strip only: the cells above row 0 are the virtual row -1 (ksw2_extz.c:32-35, ksw2_extd.c:33-41) */
	K2A_FN void top_inputs(const K2aScoring &sc, int k, int &hin, int &ein, int &e2in) const
	{
		if (S == 0) {
			const int hb = k2a_border<DUAL>(sc, k - koff + 1);
			hin = hb; ein = hb - (sc.q + sc.e); e2in = hb - (sc.q2 + sc.e2);
		}
	}

	/* One step: column jj = k - koff for the C rows of the current strip.
	 *   hin/ein/e2in: bottom row of the strip above at this column (rotated in from the previous lane, or the
	 *                 virtual row -1 / the boundary buffer, see the kernels)
	 *   tbw: traceback word(s) out (MODE != SCORE)
	 *   wild: some lane of the wavefront sees the query wildcard in this column (wave-uniform, rare), or m > 5: scores
	 *         come from the per-target-code column table `ctab` instead of the 4-entry register profile; the branch
	 *         (m > 5: from the LDS copy `mtab` of the whole matrix); the branch wraps phase 1 only, so just the C candidates merge (a whole-step if/else costs ~60 registers)
	 * Phase 1 forms every row's diagonal candidate H(i-1,j-1) + s(i,j) while the old H row is intact, phase 2 runs the
	 * E chain down the rows and rewrites the H row in place.  Returns true when the lane computed live cells. */
	K2A_FN bool step(const K2aScoring &sc, const uint32_t *ctab, const int8_t *mtab, bool wild, int k, int hin, int ein, int e2in, uint32_t *tbw)
	{
		const int jj = k - koff;
		const int dd = k - kd;                               /* jj - i0 */
		const int qe = sc.q + sc.e, qe2 = sc.q2 + sc.e2;
		int e = ein, e2 = e2in;
		if (dd >= wup) { e = K2A_NEG; e2 = K2A_NEG; }        /* the cell above is outside the band */
		/* live rows lo..hi of this strip at this column (none while the lane owns no strip: rows_m1 = -1) */
		const int lo = k2a_max(0, dd - w);
		const int hi = k2a_min(rows_m1, dd + w);
		const int cnt = k2a_max(hi - lo + 1, 0);
		const uint32_t live = cnt >= 32 ? 0xffffffffu : ((1u << cnt) - 1u) << (lo & 31);      /* lo >= 32 only with cnt = 0 */
		/* query code of this column; wildcard columns take the slow score path */
		const int qcode = qb;
		const int qsh = (qcode & 3) * 8;
		const bool qwild = qcode >= 4;
		int cand[C];
		if (!wild) {
#pragma unroll
			for (int c = 0; c < C; ++c) cand[c] = (c == 0 ? hd0 : hl[c - 1]) + (int)(int8_t)(P[c] >> qsh);
		} else if (sc.m <= 5) {
#pragma unroll
			for (int c = 0; c < C; ++c) {
				const uint32_t tb = (tbp[c >> 2] >> (8 * (c & 3))) & 0xff;
				const int sw = (int)ctab[tb < 4 ? tb : 4];
				cand[c] = (c == 0 ? hd0 : hl[c - 1]) + (qwild ? sw : (int)(int8_t)(P[c] >> qsh));
			}
		} else {                                             /* wide alphabets: one LDS byte per cell, mat[target*m + query] */
#pragma unroll
			for (int c = 0; c < C; ++c) {
				const uint32_t tb = (tbp[c >> 2] >> (8 * (c & 3))) & 0xff;
				cand[c] = (c == 0 ? hd0 : hl[c - 1]) + (int)mtab[tb * (uint32_t)sc.m + (uint32_t)qcode];
			}
		}
		uint32_t tw[TBWORDS];
#pragma unroll
		for (int x = 0; x < TBWORDS; ++x) tw[x] = 0;
#pragma unroll
		for (int c = 0; c < C; ++c) {
			int h = cand[c];
			const int fc = f[c];
			uint32_t d = 0;
			if (MODE == K2A_MODE_SCORE) {
				h = k2a_max3(h, e, fc);
				if (DUAL) h = k2a_max3(h, e2, f2[c]);
			} else if (MODE == K2A_MODE_LEFT) {           /* ksw2_extz.c:72-75, ksw2_extd.c:88-95 */
				d = h >= e ? 0u : 1u;  h = k2a_max(h, e);
				d = h >= fc ? d : 2u;  h = k2a_max(h, fc);
				if (DUAL) {
					d = h >= e2 ? d : 3u;    h = k2a_max(h, e2);
					d = h >= f2[c] ? d : 4u; h = k2a_max(h, f2[c]);
				}
			} else {                                      /* ksw2_extz.c:98-101, ksw2_extd.c:126-133 */
				d = h > e ? 0u : 1u;   h = k2a_max(h, e);
				d = h > fc ? d : 2u;   h = k2a_max(h, fc);
				if (DUAL) {
					d = h > e2 ? d : 3u;     h = k2a_max(h, e2);
					d = h > f2[c] ? d : 4u;  h = k2a_max(h, f2[c]);
				}
			}
			const bool lv = (live >> c) & 1u;
			h = lv ? h : K2A_NEG;
			/* running row maximum; ties to the last column unless extz + RIGHT + CIGAR (SURVEY 8a rule 3) */
			const int rm = rmaxr(c), rj = rmjr(c);            /* read once: with LROW these are LDS loads the scheduler can issue early */
			const bool upd = (!DUAL && MODE == K2A_MODE_RIGHT) ? (h > rm) : (h >= rm);
			rmjr(c) = upd ? jj : rj;
			rmaxr(c) = k2a_max(rm, h);
			/* gaps leaving this cell */
			const int t = h - qe;
			const int ex = e - sc.e, fx = fc - sc.e;
			if (MODE == K2A_MODE_LEFT) {
				d |= (ex > t ? 1u : 0u) << (DUAL ? 3 : 2);
				d |= (fx > t ? 1u : 0u) << (DUAL ? 4 : 3);
			} else if (MODE == K2A_MODE_RIGHT) {
				d |= (ex >= t ? 1u : 0u) << (DUAL ? 3 : 2);
				d |= (fx >= t ? 1u : 0u) << (DUAL ? 4 : 3);
			}
			e = k2a_max(ex, t);
			f[c] = k2a_max(fx, t);
			if (DUAL) {
				const int t2 = h - qe2;
				const int ex2 = e2 - sc.e2, fx2 = f2[c] - sc.e2;
				if (MODE == K2A_MODE_LEFT)  { d |= (ex2 > t2 ? 1u : 0u) << 5;  d |= (fx2 > t2 ? 1u : 0u) << 6; }
				if (MODE == K2A_MODE_RIGHT) { d |= (ex2 >= t2 ? 1u : 0u) << 5; d |= (fx2 >= t2 ? 1u : 0u) << 6; }
				e2 = k2a_max(ex2, t2);
				f2[c] = k2a_max(fx2, t2);
			}
			if (MODE != K2A_MODE_SCORE) tw[(c * TBBITS) >> 5] |= d << ((c * TBBITS) & 31);
			hl[c] = h;
		}
		hd0 = hin;
		hout = hl[C - 1]; eout = e; e2out = e2;
		if (MODE != K2A_MODE_SCORE) {
#pragma unroll
			for (int x = 0; x < TBWORDS; ++x) tbw[x] = tw[x];
		}
		return live != 0;
	}

	/* query code of the column this lane sees at step k+1; idle lanes read a clamped (valid, unused) column */
	K2A_FN int next_query_code(int k) const
	{
		const int j = k + 1 - ((k + 1 == knext) ? koff_next : koff);
		return (int)qry[k2a_min(k2a_max(j, 0), qlen - 1)];
	}

	K2A_FN bool need_fin(int k) const { return k == kfin; }
	K2A_FN int column(int k) const { return k - koff; }

	/* The strip's last column is done: replay the scalar reference's per-row epilogue for its rows, in row order
	 * (ksw2_extz.c:116-124, ksw2_extd.c:156-164; Z-drop test ksw2.h:191-207 with is_rot = 0).  The rows are staged in an
	 * LDS row buffer (3*C words per lane group) and walked in a ROLLED loop: unrolled, hipcc keeps every row's values
	 * live at once and the kernels lose a wave of occupancy for code that runs once per strip. */
	K2A_FN void do_fin(const K2aScoring &sc, K2aBook *b, int zdrop, int *rowbuf)
	{
		const int zslope = DUAL ? sc.e2 : sc.e;
#pragma unroll
		for (int c = 0; c < C; ++c) { rowbuf[c] = hl[c]; rowbuf[C + c] = rmaxr(c); rowbuf[2 * C + c] = rmjr(c); }
		int bmax = b->max, bmax_t = b->max_t, bmax_q = b->max_q, bmqe = b->mqe, bmqe_t = b->mqe_t;
		int bmte = b->mte, bmte_q = b->mte_q, bscore = b->score, bdrop = b->dropped, brows = b->rows;
#pragma nounroll
		for (int c = 0; c < C; ++c) {
			const int i = i0 + c;
			if (i < tlen && !bdrop) {
				const bool reach = i + w >= qlen - 1;           /* the row's last cell is column qlen-1 */
				const int hend = rowbuf[c], H = rowbuf[C + c], j = rowbuf[2 * C + c];
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
		S = -1; je = -1; kfin = K2A_KNONE; rows_m1 = -1;
	}
};

/* ------------------------------------------------------------------------------------------------
 * Traceback walk for one alignment (K5).  Same state machine as ksw_backtrack (ksw2.h:129-161) on
 * our own bit layout: cell (i,j) lives in the word written at step k = i/C + j by lane (i/C) % G.
 * Writes the CIGAR in walk order (end -> start) to `out`, returns the number of operations.
 * ------------------------------------------------------------------------------------------------ */
/* generation-serial layout: steps of generation g start at kbase(g); lane l of it sees column j at local
 * step j - jlo(g) + l.  These two helpers are the single definition used by the fill and by the walk. */
/* a generation's steps are padded to a multiple of 8 in the traceback block, so that the 128-byte blocks the packed kernels
 * write (K2aTbStage, ksw2_shim_hip.hip) never span two generations (which different wavefronts may be writing) */
K2A_FN size_t k2a_gen_pad(int nsteps) { return (size_t)((nsteps + 7) & ~7); }

template<int G, int C>
K2A_FN void k2a_gen_cols(int g, int qlen, int tlen, int w, int *jlo, int *nsteps)
{
	const int R = G * C;
	const int lo = k2a_max(0, g * R - w);
	const int hi = k2a_min(qlen - 1, k2a_min(g * R + R - 1, tlen - 1) + w);
	const int nl = k2a_min(G, (tlen - g * R + C - 1) / C);      /* strips in this generation */
	*jlo = lo;
	*nsteps = hi >= lo ? (hi - lo + 1) + (nl - 1) : 0;
}

/* Traceback block layout: one WB-byte word per (step, lane).  Lane-major -- word (lane, step) at (lane*nsteps + step) --
 * keeps the words a walk visits along a diagonal (same lane, consecutive steps) in the same cache lines, which is what
 * the latency-bound walk needs; the fill pays with 64 narrow stores per wave-step instead of one wide one, but it is
 * VALU-bound and has the memory pipeline to spare. */
K2A_FN size_t k2a_tb_word(size_t step, int lane, size_t nsteps, int G, int WB)
{
	(void)G;
	return ((size_t)lane * K2A_TB_PADDED(nsteps) + step) * (size_t)WB;      /* padded runs: ksw2_types.h */
}

template<int G, int C, bool MP>
K2A_FN size_t k2a_tb_steps(int qlen, int tlen, int w)
{
	if (!MP) return (size_t)((tlen + C - 1) / C - 1) + (size_t)k2a_min(qlen - 1, tlen - 1 + w) + 1;
	const int R = G * C, ngen = (tlen + R - 1) / R;
	size_t tot = 0;
	for (int g = 0; g < ngen; ++g) { int jlo, ns; k2a_gen_cols<G, C>(g, qlen, tlen, w, &jlo, &ns); tot += k2a_gen_pad(ns); }
	return tot;
}

/* Walker over one alignment's direction codes.  Inside a strip every move is an increment: a diagonal or insertion step
 * goes one lane-step back (-WB bytes in the lane-major block), a deletion stays in the same word; only crossing into the
 * strip above (c < 0) re-derives the address.  LAYOUT 0/1: the int32 kernels' 4-bit / 8-bit cells; LAYOUT 2: packed
 * tasks, byte 2c + half of a 2C-byte word. */
template<int G, int C, int LAYOUT, bool MP>
struct K2aWalk {
	enum { WB = LAYOUT == 2 ? 2 * C : (LAYOUT == 1 || LAYOUT == 3) ? C : C / 2 };
	const uint8_t *tb, *p;          /* block, current word */
	int qlen, tlen, w, half;
	size_t nsteps, gbase;
	int gcur, gjlo;
	int c;                          /* row inside the strip */

	K2A_FN void init(const uint8_t *tb_, int qlen_, int tlen_, int w_, int half_)
	{
		tb = tb_; qlen = qlen_; tlen = tlen_; w = w_; half = half_;
		nsteps = k2a_tb_steps<G, C, MP>(qlen, tlen, w);
		gcur = -1; gjlo = 0; gbase = 0; p = tb; c = 0;
	}

	K2A_FN void locate(int i, int j)
	{
		const int S = i / C;
		c = i - S * C;
		if (!MP) p = tb + k2a_tb_word((size_t)(S + j), S % G, nsteps, G, WB);
		else {
			const int g = S / G;
			if (g != gcur) {                          /* (re)locate the generation: rare, the walk only moves up */
				int ns;
				if (gcur < 0) {
					gbase = 0;
					for (int x = 0; x < g; ++x) { k2a_gen_cols<G, C>(x, qlen, tlen, w, &gjlo, &ns); gbase += k2a_gen_pad(ns); }
				} else {
					for (int x = gcur - 1; x >= g; --x) { k2a_gen_cols<G, C>(x, qlen, tlen, w, &gjlo, &ns); gbase -= k2a_gen_pad(ns); }
				}
				k2a_gen_cols<G, C>(g, qlen, tlen, w, &gjlo, &ns);
				gcur = g;
			}
			const int l = S - g * G;
			p = tb + k2a_tb_word(gbase + (size_t)(j - gjlo + l), l, nsteps, G, WB);
		}
	}

	/* direction code in the reference's byte layout (ksw2.h:125-128) from that byte; ck = row of the cell in the strip.  The packed
	 * kernels (LAYOUT 2 / 3) store direction FLAGS -- bit 0 E wins, 1 F wins [, 2 E~ wins, 3 F~ wins], then the extension flags; the
	 * winner is the highest win flag (ksw2_lane_pk.h: k2a_dir_flags) */
	K2A_FN uint32_t decode(uint32_t b, int ck) const
	{
		if (LAYOUT == 1) return b;
		if (LAYOUT == 2) { const uint32_t win = (b & 8u) ? 4u : (b & 4u) ? 3u : (b & 2u) ? 2u : (b & 1u); return win | ((b >> 4) << 3); }
		const uint32_t r4 = (b >> ((ck & 1) * 4)) & 0xfu;
		if (LAYOUT == 3) return ((r4 & 2u) ? 2u : (r4 & 1u)) | ((r4 & 4u) << 1) | ((r4 & 8u) << 1);
		return (r4 & 3u) | ((r4 & 4u) << 1) | ((r4 & 8u) << 1);
	}
	/* byte of row ck inside a lane-step word (LAYOUT 3 -- packed single gap, 4-bit flags: word ck / 4 of the lane-step, 16-bit half
	 * `half`, rows 4g .. 4g + 3 from the lowest nibble up; LAYOUT 2: byte 2 ck + half; LAYOUT 1: byte ck; LAYOUT 0: nibble ck);
	 * "the winner is not the diagonal" from that byte (decode(b, ck) & 7 != 0) */
	K2A_FN int off(int ck) const { return LAYOUT == 3 ? 4 * (ck >> 2) + 2 * half + ((ck & 3) >> 1) : LAYOUT == 2 ? 2 * ck + half : LAYOUT == 1 ? ck : ck >> 1; }
	K2A_FN uint32_t gap(uint32_t b, int ck) const
	{
		return LAYOUT == 1 ? (b & 7u) : LAYOUT == 2 ? (b & 15u) : (b >> ((ck & 1) * 4)) & 3u;
	}
};

/* the walk's window: NU units from `src` (the last valid one repeated past `nu`: every load valid, none conditional), all requested
 * before the first is stored */
struct alignas(16) K2aUnit16 { uint32_t a, b, c, d; };
struct alignas(8) K2aUnit8 { uint32_t a, b; };
template<int UNIT> struct K2aWalkUnit { typedef uint32_t T; };
template<> struct K2aWalkUnit<16> { typedef K2aUnit16 T; };
template<> struct K2aWalkUnit<8> { typedef K2aUnit8 T; };
template<class T, int NU>
K2A_FN void k2a_walk_fetch(const uint8_t *src, uint8_t *dst, int nu)
{
	/* (named values, sixteen loads in flight: an array indexed by the loop counter stays in scratch memory when the unrolling comes late) */
	static_assert(NU % 16 == 0, "window units");
	const T *s = (const T*)src;
	T *d = (T*)dst;
	const int last = nu - 1;
#pragma unroll
	for (int x = 0; x < NU; x += 16) {
		const T a0 = s[k2a_min(x, last)], a1 = s[k2a_min(x + 1, last)], a2 = s[k2a_min(x + 2, last)], a3 = s[k2a_min(x + 3, last)];
		const T a4 = s[k2a_min(x + 4, last)], a5 = s[k2a_min(x + 5, last)], a6 = s[k2a_min(x + 6, last)], a7 = s[k2a_min(x + 7, last)];
		const T a8 = s[k2a_min(x + 8, last)], a9 = s[k2a_min(x + 9, last)], a10 = s[k2a_min(x + 10, last)], a11 = s[k2a_min(x + 11, last)];
		const T a12 = s[k2a_min(x + 12, last)], a13 = s[k2a_min(x + 13, last)], a14 = s[k2a_min(x + 14, last)], a15 = s[k2a_min(x + 15, last)];
		d[x] = a0; d[x + 1] = a1; d[x + 2] = a2; d[x + 3] = a3; d[x + 4] = a4; d[x + 5] = a5; d[x + 6] = a6; d[x + 7] = a7;
		d[x + 8] = a8; d[x + 9] = a9; d[x + 10] = a10; d[x + 11] = a11; d[x + 12] = a12; d[x + 13] = a13; d[x + 14] = a14; d[x + 15] = a15;
	}
}

/* ksw_backtrack (ksw2.h:129-161) on a K2aWalk: writes the CIGAR in walk order (end -> start), returns the op count.
 *
 * The walk is a chain of dependent loads.  One lane-step word holds ALL rows of the strip at that step, and the lane's steps lie next
 * to each other in memory (k2a_tb_word), so whatever the next moves are -- M: one row and one step back, D: one row back, I: one step
 * back -- their cells lie in the K2A_WALK_NW words that end at the current one: those are requested together (one round trip to
 * memory), parked in `win` (the walk's K2A_WALK_SLOT bytes of LDS) and walked from there until the window's steps are used up or the
 * path leaves the strip.  (Rounds 1-4: the next eight cells of the DIAGONAL per round trip, thrown away at the first gap move.) */
#define K2A_WALK_NW 16
#define K2A_WALK_SLOT (K2A_WALK_NW * 32 + 16)       /* the widest lane-step word has 32 bytes; + 16: slots of neighbouring walks on different LDS banks */
template<int G, int C, int LAYOUT, bool MP>
K2A_FN int k2a_trace_walk(const uint8_t *tb, int half, int i, int j, uint32_t *out, int qlen, int tlen, int w, uint8_t *win)
{
	typedef K2aWalk<G, C, LAYOUT, MP> Walk;
	enum { WB = Walk::WB, NW = K2A_WALK_NW, UNIT = WB % 16 == 0 ? 16 : WB % 8 == 0 ? 8 : 4, NU = NW * WB / UNIT };
	Walk W;
	int n = 0, state = 0;
	uint32_t last_op = 0xffffffffu, run = 0;
	W.init(tb, qlen, tlen, w, half);
	if (i >= 0 && j >= 0) W.locate(i, j);
	while (i >= 0 && j >= 0) {
		/* steps back that stay inside the matrix (column j - k >= 0) and inside the block */
		const int back = k2a_min(k2a_min(j, (int)k2a_min((size_t)(W.p - tb) / WB, (size_t)NW)), NW - 1), nw = back + 1;
		const uint8_t *lo = W.p - (size_t)back * WB;          /* the window: nw words, [lo, lo + nw * WB) */
		const int nu = nw * WB / UNIT;
		k2a_walk_fetch<typename K2aWalkUnit<UNIT>::T, NU>(lo, win, nu);
		int k = 0;                                            /* steps of the window used up */
		while (k < nw && W.c >= 0 && j >= 0) {
			const uint8_t *pk = win + (size_t)(back - k) * WB;
			const int ck = W.c;
			/* most moves are diagonal: four cells of the diagonal at once while none of them names another winner (and no gap is open) */
			if (C >= 16 && state == 0 && k + 3 < nw && ck >= 3 && j >= 3) {      /* (strips of 8 rows: the test costs more than it saves, 1.32 against 1.19 ms on config 3) */
				const uint32_t g4 = W.gap(pk[W.off(ck)], ck) | W.gap(pk[W.off(ck - 1) - WB], ck - 1) | W.gap(pk[W.off(ck - 2) - 2 * WB], ck - 2) |
				                    W.gap(pk[W.off(ck - 3) - 3 * WB], ck - 3);
				if (g4 == 0) {
					i -= 4; j -= 4; W.c -= 4; k += 4;
					if (last_op == 0) run += 4;
					else { if (run) out[n++] = run << 4 | last_op; last_op = 0; run = 4; }
					continue;
				}
			}
			const uint32_t b = pk[W.off(ck)];
			const uint32_t d = W.decode(b, ck);
			if (state == 0) state = d & 7;
			else if (!((d >> (state + 2)) & 1)) state = 0;
			if (state == 0) state = d & 7;
			uint32_t op;
			if (state == 0) { op = 0; --i; --j; --W.c; ++k; }                      /* M: previous row, previous lane-step */
			else if (state == 1 || state == 3) { op = 2; --i; --W.c; }             /* D: previous row, same lane-step */
			else { op = 1; --j; ++k; }                                             /* I: same row, previous lane-step */
			if (op == last_op) ++run;
			else { if (run) out[n++] = run << 4 | last_op; last_op = op; run = 1; }
		}
		W.p -= (size_t)k * WB;
		if (W.c < 0 && i >= 0 && j >= 0) W.locate(i, j);                           /* crossed into the strip above */
	}
	if (i >= 0) {                                     /* leading deletion */
		if (last_op == 2) run += i + 1;
		else { if (run) out[n++] = run << 4 | last_op; last_op = 2; run = i + 1; }
	}
	if (j >= 0) {                                     /* leading insertion */
		if (last_op == 1) run += j + 1;
		else { if (run) out[n++] = run << 4 | last_op; last_op = 1; run = j + 1; }
	}
	if (run) out[n++] = run << 4 | last_op;
	return n;
}

template<int G, int C, bool DUAL, bool MP>
K2A_FN int k2a_trace_pair(const uint8_t *tb, int i, int j, uint32_t *out, int qlen, int tlen, int w, uint8_t *win)
{
	return k2a_trace_walk<G, C, DUAL ? 1 : 0, MP>(tb, 0, i, j, out, qlen, tlen, w, win);
}

/* the 4-bit wire format of uniform plans (K2aQueueDesc.unp_*): byte k of the upload holds residue codes 2k (low nibble) and 2k + 1;
 * four upload bytes -> eight arena bytes */
K2A_FN void k2a_wire4_expand(uint32_t w4, uint32_t &lo, uint32_t &hi)
{
	const uint32_t ev = w4 & 0x0f0f0f0fu, od = (w4 >> 4) & 0x0f0f0f0fu;
	lo = (ev & 0xffu) | ((od & 0xffu) << 8) | ((ev & 0xff00u) << 8) | ((od & 0xff00u) << 16);
	hi = ((ev >> 16) & 0xffu) | (((od >> 16) & 0xffu) << 8) | ((ev >> 24) << 16) | ((od >> 24) << 24);
}

/* The 2-bit wire format (round 6; K2aQueueDesc.unp_fmt = 2): byte k of the upload holds residue codes 4k .. 4k + 3, two bits each; a
 * code above 3 travels as 0 plus an ESCAPE entry -- { offset inside the pair's region : 20, run length : 8, code : 4 } -- in the last
 * K2A_WIRE2_SLOT upload bytes of the pair's region (arena padding nobody reads); a run of wildcards is one entry.  More runs in a
 * pair than entries, or a code above 15: the host repeats the batch on the general path (stream_up_t.wire_bad).
 * Four upload bytes -> sixteen arena bytes. */
/* (K2A_WIRE2_ESC / _SLOT / _PAD: ksw2_types.h -- the host's packing loop uses them) */
K2A_FN void k2a_wire2_expand(uint32_t w2, uint32_t out[4])
{
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
	for (int j = 0; j < 4; ++j) {
		const uint32_t t = (w2 >> (8 * j)) & 0xffu;
		out[j] = (t | (t << 6) | (t << 12) | (t << 18)) & 0x03030303u;
	}
}

/* Uniform plans (K2aUniform): record of pair i, and the pieces wavefront-task wt of a streamed launch waits for -- the rules the
 * host's gather follows when it copies the sequences (ksw2_host_plan.c: uni_fill_range) */
K2A_FN K2aPair k2a_uniform_pair(const K2aUniform &u, uint32_t i)
{
	K2aPair p = u.tmpl;
	p.qoff = i * u.stride; p.toff = i * u.stride + u.qpad;
	if (u.defer) p.tb_off = u.blk_base + (uint64_t)((i >> 1) / u.ng) * u.blk_bytes;
	return p;
}
K2A_FN uint32_t k2a_uniform_need(const K2aUniform &u, uint32_t wt)
{
	const uint32_t last = k2a_min((int)u.n, (int)((wt + 1) * u.ng * 2)) - 1;                    /* the task's pairs lie in the arena in index order */
	uint64_t lim = (uint64_t)last * u.stride + u.qpad + (uint32_t)u.tmpl.tlen_full + u.margin;
	if (lim > u.seq_bytes) lim = u.seq_bytes;
	uint32_t k = 0;
	while (k + 1 < u.npieces && u.pb[k + 1] < lim) ++k;
	return k + 1;
}

/* Finish one alignment after the fill: turn the bookkeeping state into the ksw_extz_t fields and pick
 * the traceback start (ksw2_extz2_sse.c:292-301 / ksw2_extz.c:127-133; SURVEY 8a rules 6-7). */
K2A_FN void k2a_finish(const K2aPair &pr, const K2aBook &b, K2aResult *r)
{
	int dropped = b.dropped;
	/* rows, or the corner column, that the band cannot reach: stop like the SSE kernels do */
	if (!dropped && (pr.tlen < pr.tlen_full || (pr.tlen_full - 1) + pr.w < pr.qlen - 1)) dropped = 1;
	r->max = b.max; r->zdropped = dropped; r->max_q = b.max_q; r->max_t = b.max_t;
	r->mqe = b.mqe; r->mqe_t = b.mqe_t; r->mte = b.mte; r->mte_q = b.mte_q; r->score = b.score;
	r->reach_end = 0; r->n_cigar = 0; r->rows_done = b.rows;
	int ti = -1, tj = -1;
	if (pr.flag & K2A_F_SCORE_ONLY) { /* no traceback, reach_end stays 0 (ksw2_extz2_sse.c:292) */ }
	else if (!dropped && !(pr.flag & K2A_F_EXTZ_ONLY)) { ti = pr.tlen_full - 1; tj = pr.qlen - 1; }
	else if (!dropped && (pr.flag & K2A_F_EXTZ_ONLY) && b.mqe + pr.end_bonus > b.max) {
		r->reach_end = 1; ti = b.mqe_t; tj = pr.qlen - 1;
	} else if (b.max_t >= 0 && b.max_q >= 0) { ti = b.max_t; tj = b.max_q; }
	r->ti = ti; r->tj = tj; r->pad[0] = 0; r->pad[1] = b.inexact;
}

#endif
