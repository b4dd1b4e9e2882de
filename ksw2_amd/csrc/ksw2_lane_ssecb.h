/*
 * ksw2_lane_ssecb.h -- the SSE-COMPATIBLE mode (ksw2_lane_ssec.h: ksw_extz2_sse / ksw_extd2_sse exactly as the reference's SSE
 * kernels return them, ksw2_extz2_sse.c:101-301, ksw2_extd2_sse.c:131-398) with the state in REGISTERS: one 16-position block
 * of the reference's byte arrays per lane -- a lane is one __m128i of the reference's inner loop.
 *
 * The position-per-lane kernel (k2a_ssec_kernel) keeps u, v, x, y [, x~, y~], s as byte arrays in LDS or HBM and walks an
 * anti-diagonal in passes of 64 positions with a fence between its phases: 1 100 cycles per pass, 65 GCUPS on 10 k reads (0.006 of
 * the VALU roofline, round 4).  Here target position p lives in lane (p >> 4) & 63, slot p & 15, for as long as it is in the band:
 * the band of anti-diagonal r covers at most w + 1 positions, so 64 lanes hold bands up to 960 positions, and a lane whose block
 * falls out of the band below takes the block that enters 64 blocks above (a ring: no data ever moves).  One step = one
 * anti-diagonal = every lane updates its 16 positions.
 *
 * Number format: the reference's arithmetic is WRAPPING int8 with signed and unsigned byte comparisons, and the padded positions
 * of the edge blocks compute on whatever their bytes hold ("leaky band") -- nothing about value ranges may be assumed.
 * A byte b is held as the 16-bit number b << 8, two positions per 32-bit register: v_pk_add_u16 / v_pk_sub_u16 then wrap exactly
 * like paddb / psubb, v_pk_max_i16 / v_pk_min_i16 are pmaxsb / pminsb, v_pk_max_u16 / v_pk_min_u16 are pmaxub / pminub (the low
 * bytes are zero and stay zero).  13 packed instructions update two positions of the single-gap recurrence, 25 of the two-piece
 * one, with no range checks anywhere.  The value of position p - 1 (the reference's shifted loads) is the neighbouring half:
 * one v_alignbit_b32 per register, the previous lane's last position through one DPP rotate.
 * H (exact-max mode; int32 per position) lives in a 1 024-entry LDS ring per wavefront: it is read and written by position
 * (last in-band cell, the three tail cells, the first cell), which registers cannot do without a select chain per access.
 *
 * Score-only tasks (KSW_EZ_SCORE_ONLY; the approximate modes with KSW_EZ_EXTZ_ONLY included), simple scoring (no
 * KSW_EZ_GENERIC_SC), bands of at most K2A_SSECB_SPAN positions; everything else keeps the position-per-lane kernel.  Same
 * bookkeeping code as there (k2a_ssec_book / k2a_ssec_follow): the results are the same bits.
 */
#ifndef KSW2_LANE_SSECB_H_
#define KSW2_LANE_SSECB_H_

#include "ksw2_lane_ssec.h"
#include "ksw2_lane_pk.h"

#define K2A_SSECB_RING 1024                       /* positions the 64 lanes hold */
/* where position p's H lives in the LDS ring: a lane's 16 entries are contiguous (four 16-byte accesses), and every fourth lane's run
 * starts four banks further on, so that the 16 lanes an access is served for at a time hit 64 different banks (unpadded, lanes L and
 * L + 4 meet in the same banks: 126 of 1 053 quad-cycles per anti-diagonal went into bank conflicts; 21 of 1 025 now,
 * profiles/r5z_10k-ssec_pmc.json) */
#define K2A_SSECB_RING_WORDS (K2A_SSECB_RING + K2A_SSECB_RING / 16)
K2A_FN int k2a_ssecb_slot(int p) { const int x = p & (K2A_SSECB_RING - 1); return x + ((x >> 6) << 2); }
#define K2A_SSECB_SPAN (K2A_SSECB_RING - 64)      /* widest band (positions of one anti-diagonal): the blocks from the one that holds
                                                   * position st - 1 to the one the score refresh reaches must be 64 different lanes */

K2A_FN uint32_t k2a_sb_c(int v) { const uint32_t h = ((uint32_t)v << 8) & 0xffffu; return h | (h << 16); }      /* the byte v in both halves */
#if defined(__HIP_DEVICE_COMPILE__)
K2A_FN uint32_t k2a_sb_minu(uint32_t a, uint32_t b)      /* asm: against the constant 1 hipcc expands the builtin into compares, selects and a v_perm */
{
	uint32_t d;
	asm("v_pk_min_u16 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
	return d;
}
K2A_FN uint32_t k2a_sb_shift(uint32_t cur, uint32_t below) { return __builtin_amdgcn_alignbit(cur, below, 16); }   /* { high half of `below`, low half of `cur` } */
#else
K2A_FN uint32_t k2a_sb_minu(uint32_t a, uint32_t b)
{
	const uint32_t al = a & 0xffffu, ah = a >> 16, bl = b & 0xffffu, bh = b >> 16;
	return (al < bl ? al : bl) | ((ah < bh ? ah : bh) << 16);
}
K2A_FN uint32_t k2a_sb_shift(uint32_t cur, uint32_t below) { return (below >> 16) | (cur << 16); }
#endif
/* signed comparison of the bytes two registers hold: 0xffff per half where a > b.  The difference of two values of the full 16-bit range
 * overflows: it saturates (v_pk_sub_i16 clamp), which keeps its sign */
#if defined(__HIP_DEVICE_COMPILE__)
K2A_FN uint32_t k2a_sb_subs(uint32_t a, uint32_t b) { uint32_t d; asm("v_pk_sub_i16 %0, %1, %2 clamp" : "=v"(d) : "v"(a), "v"(b)); return d; }
#else
K2A_FN uint32_t k2a_sb_subs(uint32_t a, uint32_t b)
{
	int lo = k2a_pk_lo(a) - k2a_pk_lo(b), hi = k2a_pk_hi(a) - k2a_pk_hi(b);
	lo = lo > 32767 ? 32767 : lo < -32768 ? -32768 : lo; hi = hi > 32767 ? 32767 : hi < -32768 ? -32768 : hi;
	return k2a_pk_mk(lo, hi);
}
#endif
K2A_FN uint32_t k2a_sb_gt(uint32_t a, uint32_t b) { return k2a_pk_sign(k2a_sb_subs(b, a)); }

/* a block: 16 positions, two per register.  A vector type, so that a read or write by slot number is an element extract / insert on a
 * value (register indexing or a select chain, the compiler's choice) and never an address into a stack copy of the lane's state */
#if defined(__clang__)
typedef uint32_t k2a_blk __attribute__((ext_vector_type(8)));      /* subscripts are element extracts / inserts (a vector_size type's are addresses) */
typedef uint32_t k2a_quad __attribute__((ext_vector_type(4)));
#else
typedef uint32_t k2a_blk __attribute__((vector_size(32)));
typedef uint32_t k2a_quad __attribute__((vector_size(16)));
#endif
/* one position's byte out of a block (slot 0..15), as the low 8 bits.  Selects over the eight registers and no subscript by the slot
 * number: hipcc turns such a subscript on a member into an address, and the lane's whole state then lives in scratch memory */
K2A_FN uint32_t k2a_sb_get(const k2a_blk &a, int slot)
{
	/* a tree of seven selects on the three bits of the register number (three masks): 10 instructions where one compare and select
	 * per register takes 24 */
	const uint32_t m0 = k2a_bit_mask((uint32_t)slot, 1), m1 = k2a_bit_mask((uint32_t)slot, 2), m2 = k2a_bit_mask((uint32_t)slot, 3);
	const uint32_t t01 = k2a_pk_selv(m0, a[1], a[0]), t23 = k2a_pk_selv(m0, a[3], a[2]), t45 = k2a_pk_selv(m0, a[5], a[4]), t67 = k2a_pk_selv(m0, a[7], a[6]);
	const uint32_t d = k2a_pk_selv(m2, k2a_pk_selv(m1, t67, t45), k2a_pk_selv(m1, t23, t01));
	return ((slot & 1) ? d >> 24 : d >> 8) & 0xffu;
}
K2A_FN void k2a_sb_set(k2a_blk &a, int slot, int byte)
{
	const uint32_t h = ((uint32_t)byte << 8) & 0xffffu, keep = (slot & 1) ? 0x0000ffffu : 0xffff0000u, put = (slot & 1) ? h << 16 : h;
#pragma unroll
	for (int i = 0; i < 8; ++i) {
		const uint32_t m = (uint32_t)-(int)((slot >> 1) == i);
		a[i] = (a[i] & (keep | ~m)) | (put & m);
	}
}

template<bool DUAL>
struct K2aSsecBlk {
	int blk;                                   /* the block this lane holds: positions 16 * blk .. 16 * blk + 15; -1: none yet */
	k2a_blk U, V, X, Y, X2, Y2, S;             /* the reference's bytes, b << 8 per half (X2, Y2: two-piece only; S of the single-gap kernel: with the 2 (q + e)
	                                            * its cell adds first (ksw2_extz2_sse.c:163) already in) */
	/* Scores (round 6): one PROFILE dword per position -- byte c = the score byte of the position's target code against query code c
	 * (0..3; with the 2 (q + e) of the single-gap kernel in) -- P0 = the even slots', P1 = the odd slots'; the query codes of a register's
	 * two positions are v_perm_b32 selector bytes, so a register's two scores are ONE v_perm_b32 (rounds 4-5: xor, min, mad, two
	 * selects on target / query code registers).  m <= 5: the host sends wider alphabets to the position-per-lane kernels. */
	k2a_blk P0, P1;
	k2a_blk QW;                                /* per half: { 0x0c, query code r - p of the position on the current anti-diagonal } (0 outside the query), bit 15: the wildcard */
	uint32_t qn;                               /* the query byte slot 0 pairs with on the next anti-diagonal (ask_query) */

	K2A_FN int p0() const { return blk << 4; }

	/* the query code position p pairs with on anti-diagonal r (k2a_ssec_qcode): the byte at the clamped index (an index past the
	 * query belongs to a position the refresh never reaches, p < st0; the clamp only keeps the load inside the arena), then what
	 * the code is */
	K2A_FN static uint32_t qbyte(const uint8_t *qry, int qlen, int r, int p) { return qry[k2a_min(k2a_max(r - p, 0), qlen - 1)]; }
	K2A_FN static uint32_t qcode_of(const K2aSsec &P, uint32_t byte, int r, int p)
	{
		const uint32_t c = r - p >= 0 ? byte : 0u;
		return ((c & 0x7fu) << 8) | (c == (uint32_t)(P.m - 1) ? 0x8000u : 0u) | 0x0cu;      /* selector form: the code picks the profile byte, 0x0c the zero low byte; bit 15: the wildcard */
	}
	/* the profile of a position whose target code is t: sc_mch at byte t, sc_mis elsewhere, sc_N everywhere for the wildcard and at
	 * the wildcard's byte (alphabets below five codes) -- k2a_ssec_score for query codes 0..3 */
	K2A_FN static uint32_t profile_of(const K2aSsec &P, uint32_t t, int ofs)
	{
		const uint32_t cm = (uint32_t)(P.sc_mch + ofs) & 0xffu, cd = (uint32_t)(P.sc_mis + ofs) & 0xffu, cn = (uint32_t)(P.sc_N + ofs) & 0xffu, wc = (uint32_t)(P.m - 1);
		uint32_t v = cd * 0x01010101u;
		if (t < 4u) v = (v & ~(0xffu << (8 * t))) | (cm << (8 * t));
		if (wc < 4u) v = (v & ~(0xffu << (8 * wc))) | (cn << (8 * wc));
		return t == wc ? cn * 0x01010101u : v;
	}
	K2A_FN static uint32_t qcode(const K2aSsec &P, const uint8_t *qry, int qlen, int r, int p) { return qcode_of(P, qbyte(qry, qlen, r, p), r, p); }

	/* take block b: the reference's freshly allocated arrays (zeros / the gap-open differences, ksw2_extz2_sse.c:84,
	 * ksw2_extd2_sse.c:111-116), the target codes (0 past the target's end), the query codes of anti-diagonal r */
	K2A_FN void init_block(const K2aSsec &P, int b, const uint8_t *tgt, int tlen, const uint8_t *qry, int qlen, int r)
	{
		const uint32_t g1 = DUAL ? k2a_sb_c(-P.q - P.e) : 0u, g2 = k2a_sb_c(-P.q2 - P.e2);
		blk = b;
#pragma unroll
		for (int i = 0; i < 8; ++i) {
			const int p = (b << 4) + 2 * i;
			const uint32_t t0 = p < tlen ? tgt[p] : 0u, t1 = p + 1 < tlen ? tgt[p + 1] : 0u;
			U[i] = V[i] = X[i] = Y[i] = g1; S[i] = DUAL ? 0u : k2a_sb_c(2 * (P.q + P.e));
			if (DUAL) { X2[i] = g2; Y2[i] = g2; }
			P0[i] = profile_of(P, t0, DUAL ? 0 : 2 * (P.q + P.e)); P1[i] = profile_of(P, t1, DUAL ? 0 : 2 * (P.q + P.e));
			QW[i] = qcode(P, qry, qlen, r, p) | (qcode(P, qry, qlen, r, p + 1) << 16);
		}
	}

	/* anti-diagonal r - 1 -> r: every position's query index grows by one, i.e. the codes move one slot up; `qn` is the byte asked
	 * for on the previous anti-diagonal (ask_query) */
	K2A_FN void shift_query(const K2aSsec &P, int r)
	{
#pragma unroll
		for (int i = 7; i > 0; --i) QW[i] = k2a_sb_shift(QW[i], QW[i - 1]);
		QW[0] = (QW[0] << 16) | qcode_of(P, qn, r, p0());
	}
	/* the byte slot 0 pairs with on anti-diagonal r + 1.  Called once per anti-diagonal, unconditionally and after the lane may have
	 * taken a new block: a load under a condition into a value that lives around the loop is waited for on the spot
	 * (ksw2_lane_pk.h, k2a_load_early) */
	K2A_FN void ask_query(const uint8_t *qry, int qlen, int r) { qn = qbyte(qry, qlen, r + 1, p0()); }

	/* 16-bit slot mask of the block's positions inside [lo, hi) */
	K2A_FN uint32_t slot_mask(int lo, int hi) const
	{
		const int a = k2a_min(k2a_max(lo - p0(), 0), 16), b = k2a_min(k2a_max(hi - p0(), 0), 16);
		return b > a ? ((1u << b) - 1u) & ~((1u << a) - 1u) : 0u;
	}
	K2A_FN static uint32_t half_mask(uint32_t m16, int i)   /* slots 2i, 2i + 1 of a slot mask as a halves mask */
	{
		return k2a_pk_sel(0x0000ffffu, k2a_bit_mask(m16, 2 * i), k2a_bit_mask(m16, 2 * i + 1));
	}

	/* the scores of positions [st0, pend) below the padded target length (ksw2_extz2_sse.c:125-140, simple scoring) */
	/* QWILD = false: the build for tasks whose QUERY holds no wildcard code (the kernel looks first, k2a_ssec_blk_kernel): two
	 * instructions per register less */
	template<bool QWILD = true>
	K2A_FN void refresh_scores(const K2aSsec &P, int st0, int pend)
	{
		const uint32_t em = slot_mask(st0, pend);
		if (em == 0) return;
		const uint32_t cn = k2a_sb_c(P.sc_N + (DUAL ? 0 : 2 * (P.q + P.e)));
		const uint32_t em2 = em | (em << 15);                                                      /* slot 2i at bit 2i, slot 2i + 1 at bit 2i + 16: one shift puts both at their halves' sign bits */
#pragma unroll
		for (int i = 0; i < 8; ++i) {
			/* { 0, P0[code lo], 0, P1[code hi] }: the selector's high half reads the second operand's bytes (4 + code) */
			uint32_t sc = k2a_perm(P1[i], P0[i], QW[i] | 0x04000000u);
			if (QWILD) sc = k2a_pk_selv(k2a_pk_sign(QW[i]), cn, sc);                               /* the query's wildcard */
			S[i] = k2a_pk_selv(k2a_pk_sign(em2 << (15 - 2 * i)), sc, S[i]);
		}
	}

	/* One anti-diagonal for the block (ksw2_extz2_sse.c:146-222 / ksw2_extd2_sse.c:189-321): pv / px / px2 = the previous
	 * anti-diagonal's v / x / x~ of position 16 * blk - 1 in the HIGH half (the lane below's last register, or the band edge's
	 * constants).  Only for a block inside [st, en].  MODE != SCORE: dirw = the block's 16 direction bytes (k2a_ssec_cell: the
	 * winner by the kernels' own strict / non-strict comparisons per alignment mode, the continuation flags from the signs of the
	 * gap states before they are clipped at 0) in position order, as the reference stores them. */
	template<int MODE>
	K2A_FN void update(const K2aSsec &P, uint32_t pv, uint32_t px, uint32_t px2, uint32_t *dirw)
	{
		uint32_t dd[8];
		if (!DUAL) {
			const uint32_t ccap = k2a_sb_c(P.sc_mch + 2 * (P.q + P.e)), cq = k2a_sb_c(P.q);
			/* registers from the top one down: a register's left neighbours are the OLD values of the register below it, which is
			 * still untouched then -- walking upwards, every old V[i] / X[i] had to be kept in a copy for the next register (round 6) */
#pragma unroll
			for (int i = 7; i >= 0; --i) {
				const uint32_t vt1 = k2a_sb_shift(V[i], i ? V[i - 1] : pv), xt1 = k2a_sb_shift(X[i], i ? X[i - 1] : px);
				const uint32_t a = k2a_pk_add(xt1, vt1), b = k2a_pk_add(Y[i], U[i]);
				const uint32_t z = k2a_pk_max(S[i], a);
				uint32_t zu = k2a_pk_maxu(z, b);
				zu = k2a_sb_minu(zu, ccap);
				const uint32_t z2 = k2a_pk_sub(zu, cq);
				const uint32_t a1 = k2a_pk_sub(a, z2), b1 = k2a_pk_sub(b, z2);
				if (MODE == K2A_MODE_LEFT) {
					uint32_t d = k2a_sb_gt(a, S[i]) & 0x00010001u;
					d = k2a_pk_selv(k2a_sb_gt(b, z), 0x00020002u, d);
					dd[i] = d | (k2a_sb_gt(a1, 0u) & 0x00080008u) | (k2a_sb_gt(b1, 0u) & 0x00100010u);
				} else if (MODE == K2A_MODE_RIGHT) {
					uint32_t d = ~k2a_sb_gt(S[i], a) & 0x00010001u;
					d = k2a_pk_selv(k2a_sb_gt(z, b), d, 0x00020002u);
					dd[i] = d | (~k2a_sb_gt(0u, a1) & 0x00080008u) | (~k2a_sb_gt(0u, b1) & 0x00100010u);
				}
				V[i] = k2a_pk_sub(zu, U[i]); U[i] = k2a_pk_sub(zu, vt1);
				X[i] = k2a_pk_max(a1, 0u); Y[i] = k2a_pk_max(b1, 0u);
			}
		} else {
			const uint32_t cm = k2a_sb_c(P.sc_mch), cq = k2a_sb_c(P.q), cq2 = k2a_sb_c(P.q2), cqe = k2a_sb_c(P.q + P.e), cqe2 = k2a_sb_c(P.q2 + P.e2);
#pragma unroll
			for (int i = 7; i >= 0; --i) {                        /* top register down: see the single-gap loop */
				const uint32_t vt1 = k2a_sb_shift(V[i], i ? V[i - 1] : pv), xt1 = k2a_sb_shift(X[i], i ? X[i - 1] : px), x2t1 = k2a_sb_shift(X2[i], i ? X2[i - 1] : px2);
				uint32_t a = k2a_pk_add(xt1, vt1), b = k2a_pk_add(Y[i], U[i]), a2 = k2a_pk_add(x2t1, vt1), b2 = k2a_pk_add(Y2[i], U[i]);
				uint32_t z = S[i], d = 0;
				if (MODE == K2A_MODE_SCORE) z = k2a_pk_max(k2a_pk_max(z, a), k2a_pk_max(b, k2a_pk_max(a2, b2)));
				else if (MODE == K2A_MODE_LEFT) {
					d = k2a_sb_gt(a, z) & 0x00010001u;                  z = k2a_pk_max(z, a);
					d = k2a_pk_selv(k2a_sb_gt(b, z), 0x00020002u, d);    z = k2a_pk_max(z, b);
					d = k2a_pk_selv(k2a_sb_gt(a2, z), 0x00030003u, d);   z = k2a_pk_max(z, a2);
					d = k2a_pk_selv(k2a_sb_gt(b2, z), 0x00040004u, d);   z = k2a_pk_max(z, b2);
				} else {
					d = ~k2a_sb_gt(z, a) & 0x00010001u;                 z = k2a_pk_max(z, a);
					d = k2a_pk_selv(k2a_sb_gt(z, b), d, 0x00020002u);    z = k2a_pk_max(z, b);
					d = k2a_pk_selv(k2a_sb_gt(z, a2), d, 0x00030003u);   z = k2a_pk_max(z, a2);
					d = k2a_pk_selv(k2a_sb_gt(z, b2), d, 0x00040004u);   z = k2a_pk_max(z, b2);
				}
				z = k2a_pk_min(z, cm);
				V[i] = k2a_pk_sub(z, U[i]); U[i] = k2a_pk_sub(z, vt1);
				uint32_t tmp = k2a_pk_sub(z, cq);
				a = k2a_pk_sub(a, tmp); b = k2a_pk_sub(b, tmp);
				tmp = k2a_pk_sub(z, cq2);
				a2 = k2a_pk_sub(a2, tmp); b2 = k2a_pk_sub(b2, tmp);
				if (MODE == K2A_MODE_LEFT)
					dd[i] = d | (k2a_sb_gt(a, 0u) & 0x00080008u) | (k2a_sb_gt(b, 0u) & 0x00100010u) | (k2a_sb_gt(a2, 0u) & 0x00200020u) | (k2a_sb_gt(b2, 0u) & 0x00400040u);
				else if (MODE == K2A_MODE_RIGHT)
					dd[i] = d | (~k2a_sb_gt(0u, a) & 0x00080008u) | (~k2a_sb_gt(0u, b) & 0x00100010u) | (~k2a_sb_gt(0u, a2) & 0x00200020u) | (~k2a_sb_gt(0u, b2) & 0x00400040u);
				X[i] = k2a_pk_sub(k2a_pk_max(a, 0u), cqe); Y[i] = k2a_pk_sub(k2a_pk_max(b, 0u), cqe);
				X2[i] = k2a_pk_sub(k2a_pk_max(a2, 0u), cqe2); Y2[i] = k2a_pk_sub(k2a_pk_max(b2, 0u), cqe2);
			}
		}
		if (MODE != K2A_MODE_SCORE) {
#pragma unroll
			for (int j = 0; j < 4; ++j) dirw[j] = k2a_perm(dd[2 * j + 1], dd[2 * j], 0x06040200u);      /* the low byte of each half, four positions per word */
		}
	}

	/* what the v byte of slot s adds to H (k2a_ssec_dh) */
	K2A_FN int dh(const K2aSsec &P, int s) const
	{
		const uint32_t d = V[s >> 1];
		if (DUAL) return (s & 1) ? (int)d >> 24 : (int)(d << 16) >> 24;
		return (int)((s & 1) ? d >> 24 : (d >> 8) & 0xffu) - (P.q + P.e);
	}

	/* Exact mode (ksw2_extz2_sse.c:238-247): H += v for all 16 positions of the block, from `hv` (their H before this anti-diagonal,
	 * read out of the ring `hl` ahead of time) back into the ring; returns the block's best cell of the four-lane region [st0, en1)
	 * as k2a_dm_key (0: none).  Only the positions in [st0, en0) are H values; what the others receive is never used: the one at en0
	 * is assigned by the caller, those above are assigned before they enter, those below st0 have left -- except the one the next
	 * anti-diagonal's last cell starts from while the band is a single position, which the caller carries itself.
	 * Inside the block a cell is better if its H is higher, then if its lane class (t - st0) & 3 is lower, then if t is lower --
	 * the order of k2a_dm_key -- held as H * 64 + rank with |H| < 2^25. */
	K2A_FN uint64_t advance_H(const K2aSsec &P, int *hl, const int *hv, int st0, int en1)
	{
		const uint32_t em = slot_mask(st0, en1);
		int *hp = hl + k2a_ssecb_slot(p0());
		/* the keys without their lane class first -- (h << 6) + 63 - s, a constant per slot -- and the best of every fourth slot; the class
		 * term ((s - st0) & 3) << 4 is the same for the slots s = k mod 4 and comes off once per k (round 6: it was sixteen scalar
		 * shift-and-or chains per anti-diagonal) */
		const int none = INT32_MIN + 64;                       /* stays below every key when a class term comes off */
		int b4[4] = { none, none, none, none };
#pragma unroll
		for (int s = 0; s < 16; ++s) {
			const int h = hv[s] + dh(P, s);
			hp[s] = h;
			const int key = (int)(((uint32_t)h << 6) + (uint32_t)(63 - s));
			b4[s & 3] = k2a_max(b4[s & 3], (int)k2a_pk_selv(k2a_bit_mask(em, s), (uint32_t)key, (uint32_t)none));
		}
		int best = INT32_MIN;
#pragma unroll
		for (int k = 0; k < 4; ++k) best = k2a_max(best, b4[k] - (int)((((uint32_t)(k - st0)) & 3u) << 4));
		if (best < INT32_MIN + 128) best = INT32_MIN;
		if (best == INT32_MIN) return 0;
		return k2a_dm_key(best >> 6, p0() + (15 - (best & 15)), st0);
	}
};

#endif
