/*
 * ksw2_lane_extfb.h -- the gap-linear X-drop extension (ksw_extf2_sse, ksw2_extf2_sse.c:11-98; ksw2_lane_extf.h) in registers:
 * four, two or one extensions per wavefront, G = 16 / 32 / 64 lanes each (bands up to 160 / 416 / 928 positions), every lane one 16-position block of the reference's U / V / S byte arrays in
 * registers (the number format and the block helpers of ksw2_lane_ssecb.h: a byte b is the 16-bit number b << 8, two positions per
 * register, so the packed 16-bit instructions wrap and compare exactly like the reference's byte instructions).
 *
 * Why: an anti-diagonal of a band of 100 positions is two passes of the position-per-lane kernels, and what those spend per
 * anti-diagonal is the wavefront-uniform part -- band bounds, the followed cell, the loop: 79 vector + 120 scalar instructions for 101
 * cells (profiles/r5p_extf_pmc.json: the LDS form), 0.019 of the VALU roofline.  Here that part is computed by every lane for its own extension
 * (16 lanes redundantly) and serves four extensions per instruction; the one-extension-per-lane form (k2a_extf_lane_diag) goes further
 * but needs 64 extensions per wavefront, i.e. batches of 10^5 extensions to fill the device, this one 4.
 *
 * An extension's G lanes form a ring of G blocks (16 G positions): the blocks from the one below the first updated block (the carry
 * into it, the followed cell) to the one the score refresh reaches must be different lanes, and a lane that takes its next block
 * must have been fed that block's query codes for 16 anti-diagonals before the block is first refreshed: bands of at most
 * K2A_EXTFB_SPAN(G) positions.  A lane takes its next block (G blocks up) as soon as its block has left that range; the block's target
 * codes were asked for on every anti-diagonal before (an unconditional load: ksw2_lane_pk.h, k2a_load_early).
 */
#ifndef KSW2_LANE_EXTFB_H_
#define KSW2_LANE_EXTFB_H_

#include "ksw2_lane_extf.h"
#include "ksw2_lane_ssecb.h"

#define K2A_EXTFB_SPAN(G) (((G) - 6) * 16)      /* G lanes per extension: blocks in use <= span / 16 + 4 <= G - 2 of the ring's G (160, 416, 928) */

/* one lane of a group of 16: its extension's parameters (the same in the 16 lanes), its block, the followed cell */
template<int G>
struct K2aExtfBlk {
	int qlen, tlen, w, xdrop, tpad, nr;
	const uint8_t *qa, *ta;
	int blk;                                   /* positions 16 * blk .. 16 * blk + 15 */
	k2a_blk U, V, S, TC, QW;                   /* S with the 2 e the cell adds first (ksw2_extf2_sse.c:66) already in; codes one per half */
	k2a_quad tn;                               /* target bytes of block blk + 16, as loaded (no plain array in here: one subscript by a loop
	                                            * counter and the whole lane state stays in scratch memory) */
	uint32_t qn;                               /* query byte slot 0 pairs with on the next anti-diagonal */
	K2aExtfBook bk;
	int prev_lo, prev_hi, rdone;
	bool done;
	K2aExtfDiag d;                             /* the current anti-diagonal */

	K2A_FN int p0() const { return blk << 4; }
	K2A_FN static k2a_blk zero() { return k2a_blk{ 0, 0, 0, 0, 0, 0, 0, 0 }; }

	/* target bytes of block b; past the padded target the address only has to stay inside the arena */
	K2A_FN void ask_target(int b)
	{
		const uint32_t *p = (const uint32_t*)(ta + k2a_min(b << 4, tpad));      /* the arena aligns sequences to 4 bytes */
		tn = k2a_quad{ p[0], p[1], p[2], p[3] };
	}
	K2A_FN void take_target(int b)               /* tn -> TC: one code per half, 0 past the target's end */
	{
#pragma unroll
		for (int i = 0; i < 8; ++i) {
			const int p = (b << 4) + 2 * i;
			const uint32_t w4 = tn[i >> 1], c0 = (w4 >> (16 * (i & 1))) & 0xffu, c1 = (w4 >> (16 * (i & 1) + 8)) & 0xffu;
			TC[i] = (p < tlen ? c0 : 0u) | ((p + 1 < tlen ? c1 : 0u) << 16);
		}
	}
	K2A_FN void start(const K2aExtf &par, const K2aPair &pr, const uint8_t *seq, int gl, bool live)
	{
		qlen = pr.qlen; tlen = pr.tlen; w = pr.w; xdrop = pr.zdrop; tpad = (tlen + 15) & ~15; nr = qlen + tlen - 1;
		qa = seq + pr.qoff; ta = seq + pr.toff;
		blk = gl;
		U = V = QW = zero();
		const uint32_t s0 = k2a_sb_c(2 * par.e);
		S = k2a_blk{ s0, s0, s0, s0, s0, s0, s0, s0 };
		ask_target(gl); take_target(gl); ask_target(gl + G);
		qn = qbyte(0);
		k2a_extf_book_reset(bk);
		prev_lo = prev_hi = -1; rdone = 0; done = !live;
		if (!live) { qlen = tlen = 1; nr = 1; tpad = 16; w = 0; }
	}

	K2A_FN uint32_t qbyte(int r) const { return qa[k2a_min(k2a_max(r - p0(), 0), qlen - 1)]; }

	/* phase A of anti-diagonal r: bounds, query codes one slot up, the lane's next block if its own has left, the loads for r + 1,
	 * the cell at position r (ksw2_extf2_sse.c:46).  Returns false once the extension has ended. */
	K2A_FN bool begin(const K2aExtf &par, int r)
	{
		if (!done && (r >= nr || !k2a_extf_diag(r, qlen, tlen, w, tpad, d))) { done = true; rdone = k2a_min(r, nr); }
		if (done) return false;
		const int j = r - p0();
#pragma unroll
		for (int i = 7; i > 0; --i) QW[i] = k2a_sb_shift(QW[i], QW[i - 1]);
		QW[0] = (QW[0] << 16) | ((j >= 0 && j < qlen) ? qn : 0u);
		if (blk < (d.blo >> 4) - 1) {
			blk += G;
			take_target(blk);
			U = V = zero();
			const uint32_t s0 = k2a_sb_c(2 * par.e);
			S = k2a_blk{ s0, s0, s0, s0, s0, s0, s0, s0 };
		}
		if (d.bhi >= r && blk == (r >> 4)) k2a_sb_set(U, r & 15, 0);
		return true;
	}
	K2A_FN void ask(int r) { qn = qbyte(r + 1); ask_target(blk + G); }      /* every lane, every anti-diagonal: unconditional loads */

	/* phase B: pv = V[7] of the lane below in the ring (previous anti-diagonal; position 16 * blk - 1 in the high half).  S refresh
	 * (:48-61), the block's cells (:64-78) if it lies in [blo, bhi]; returns the followed cell's two bytes as this lane holds them */
	K2A_FN void update(const K2aExtf &par, uint32_t pv, uint32_t &vsel, uint32_t &usel)
	{
		if (!done) {
			const bool carry_ok = d.blo > 0 && d.blo - 1 >= prev_lo && d.blo - 1 <= prev_hi;
			if (blk == (d.blo >> 4) && !carry_ok) pv = 0;
			const int a = k2a_min(k2a_max(d.lo - p0(), 0), 16), b = k2a_min(k2a_max(d.fresh_end - p0(), 0), 16);
			const uint32_t em = b > a ? ((1u << b) - 1u) & ~((1u << a) - 1u) : 0u;
			if (em) {
				const uint32_t cm = k2a_sb_c(par.mch + 2 * par.e), cd = k2a_sb_c(par.mis - par.mch);
				const uint32_t em2 = em | (em << 15);                  /* slot 2i at bit 2i, slot 2i + 1 at bit 2i + 16: one shift puts both at their halves' sign bits (K2aSsecBlk::refresh_scores) */
#pragma unroll
				for (int i = 0; i < 8; ++i) {
					const uint32_t ne = k2a_sb_minu(TC[i] ^ QW[i], 0x00010001u);
					const uint32_t sc = k2a_pk_mad(ne, cd, cm);
					S[i] = k2a_pk_selv(k2a_pk_sign(em2 << (15 - 2 * i)), sc, S[i]);
				}
			}
			if (blk >= (d.blo >> 4) && blk <= (d.bhi >> 4)) {
#pragma unroll
				for (int i = 7; i >= 0; --i) {                      /* top register down: its left neighbour is the register below, still untouched (no copies of the old values) */
					const uint32_t av = k2a_sb_shift(V[i], i ? V[i - 1] : pv);
					const uint32_t z = k2a_pk_maxu(k2a_pk_max(S[i], av), U[i]);
					V[i] = k2a_pk_sub(z, U[i]); U[i] = k2a_pk_sub(z, av);
				}
			}
		}
		vsel = k2a_sb_get(V, bk.follow & 15); usel = k2a_sb_get(U, (bk.follow + 1) & 15);
	}
	K2A_FN int vlane() const { return (bk.follow >> 4) & (G - 1); }        /* the lanes of the group that hold V[follow] / U[follow + 1] */
	K2A_FN int ulane() const { return ((bk.follow + 1) >> 4) & (G - 1); }

	/* phase C: the followed cell (:80-92) */
	K2A_FN void finish_diag(const K2aExtf &par, int r, uint32_t vf, uint32_t un)
	{
		if (done) return;
		if (!k2a_extf_follow(bk, d, r, par.e, xdrop, vf, un)) { done = true; rdone = r; }
		prev_lo = d.blo; prev_hi = d.bhi;
	}
};

#endif
