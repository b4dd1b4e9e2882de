/*
 * ksw2_lane_pkmp.h -- packed-int16 per-lane code for the GENERATION-SERIAL schedule: bands too wide to stay resident in a
 * systolic array (config 4: 16.5 k x 16.5 k, unbanded), two same-shape alignments per wavefront-lane like ksw2_lane_pk.h.
 * Replaces the same reference loops (ksw2_extz2_sse.c:101-289, ksw2_extd2_sse.c:131-387) with the scalar ksw_extz / ksw_extd
 * semantics.
 *
 * The re-based packed kernels keep a strip's values relative to ONE base, which works while the strip's window is a few
 * hundred columns.  Here a strip walks over every column of a 16 k-long row, so the base SLIDES: every K2A_PKMP_T steps each
 * lane re-centres its values on the largest H it currently holds (`rebase`): the difference bounds of the affine recurrence
 * keep everything a lane holds at one column within a few hundred units of that, whatever the read length.  What does not
 * stay close is a row's running maximum (it may lie thousands of columns back): it is tracked in registers only inside the
 * current window and merged, as a 64-bit key (absolute value, column), into a per-row slot in HBM scratch at every re-base
 * (`flush_rowmax`: one no-return atomic max per row and alignment; ties resolve exactly like the scan order of the
 * reference because the key's low word orders the columns).  -inf (-16384) is re-clamped at every re-base, and a value
 * arriving from the lane above is taken as -inf when it is below K2A_PKMP_DEAD before the difference of the two lanes'
 * bases is added (a finished or not yet started lane's base is arbitrary).
 *
 * Schedule: generation g = strips g*64 .. g*64+63 over all their in-band columns (k2a_gen_cols, ksw2_lane.h), lane l skewed
 * by l steps.  The bottom row of a generation goes to the next one through a boundary array in HBM as packed values plus
 * the producing lane's bases (16 bytes per column); different generations of one pair run on different wavefronts of a
 * workgroup, pipelined (ksw2_shim_hip.hip: k2a_fill_pkmp_kernel).
 */
#ifndef KSW2_LANE_PKMP_H_
#define KSW2_LANE_PKMP_H_

#include "ksw2_lane_pk.h"
#include "ksw2_shim.h"

/* K2A_PKMP_T, K2A_PKMP_DEAD, K2A_PKMP_RMAX_LIMIT: ksw2_types.h (the host's range check pk_slide_ok uses them) */
/* K2A_PKMP_WAVES, the wavefronts (generations in flight) per pair of alignments: ksw2_shim.h (the host sizes the key blocks by it) */
#define K2A_PKMP_SPILL_WORDS(C) (64 * (C) * 2 * 2)     /* uint32 per wavefront: one 64-bit key per row, lane and alignment */
#define K2A_PKMP_BND_WORDS(qlen, dual) ((((size_t)(qlen) * ((dual) ? 5 : 4) + 16) + 3) & ~(size_t)3)   /* uint32 per task: {H, E, baseA, baseB}[qlen] (+ E~[qlen]), rounded so that the 64-bit keys behind it stay aligned */

#if defined(__HIP_DEVICE_COMPILE__)
K2A_FN void k2a_key_max(unsigned long long *slot, unsigned long long key)
{
	/* result unused: a no-return atomic performed in the XCD's L2.  Workgroup scope: every slot belongs to one lane of one
	 * wavefront, and agent-scope atomics are carried out beyond the L2 on this part -- 5.4e9 of them per launch of config 4
	 * showed up as 140 GB of extra HBM writes (profiles/r2_cfg4_pmc.json) */
	__hip_atomic_fetch_max(slot, key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
K2A_FN unsigned long long k2a_key_load(const unsigned long long *slot)       /* past the L1, where an older copy of the slot may sit */
{
	return __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
#else
K2A_FN void k2a_key_max(unsigned long long *slot, unsigned long long key) { if (key > *slot) *slot = key; }
K2A_FN unsigned long long k2a_key_load(const unsigned long long *slot) { return *slot; }
#endif

template<int C, bool DUAL, int MODE, bool TN = false>
struct K2aLanePkMp {
	typedef K2aLanePk<64, C, DUAL, MODE, true, false, 0, false, TN> Pk;
	enum { G = 64, TBWORDS = Pk::TBWORDS, FIRSTJ = (!DUAL && MODE == K2A_MODE_RIGHT) };    /* extz + RIGHT + CIGAR: ties to the first column */
	Pk P;
	unsigned long long *spill;          /* this lane's C x 2 keys: spill[c * 2 + half] (per-wavefront block, lane-major) */

	K2A_FN void setup(const K2aPair &prA, const K2aPair &prB, const uint8_t *seq, int lane, bool valid, unsigned long long *spill_wave, const uint32_t *cptab)
	{
		P.setup(prA, prB, seq, lane, valid, cptab);
		P.Snext = 0; P.knext = K2A_KNONE;
		spill = spill_wave + (size_t)lane * (C * 2);
	}
	K2A_FN void clear_spill()
	{
#pragma unroll
		for (int x = 0; x < 2 * C; ++x) spill[x] = 0ull;
	}

	K2A_FN void begin_generation(int g, int jlo)
	{
		const k2a_pk neg = k2a_pku(K2A_NEG16);
		P.S = -1; P.je = -1; P.kfin = K2A_KNONE; P.rows_m1 = -1; P.hasn = 0; P.wn = false;
		P.Snext = g * G + P.gl;
		P.koff_next = P.gl - jlo;
		P.knext = P.Snext < P.nstrips ? P.koff_next + Pk::first_col(P.Snext, P.w) : K2A_KNONE;
		P.hout = P.eout = P.e2out = neg; P.hu_prev = neg; P.hd0 = neg; P.delta = 0;
	}
	K2A_FN bool need_init(int k) const { return k == P.knext; }
	K2A_FN bool need_fin(int k) const { return k == P.kfin; }
	K2A_FN int column(int k) const { return k - P.koff; }
	/* start the lane's strip of this generation; bsA / bsB and P.hu_prev describe H(i0 - 1, first column - 1) */
	K2A_FN void do_init(const K2aScoring &sc, int bsA, int bsB)
	{
		P.do_init(sc, bsA, bsB);
		P.knext = K2A_KNONE;            /* one strip per lane and generation */
	}
	/* after any base changed in the wavefront: bsA / bsB = the current bases of the lane above */
	K2A_FN void refresh_delta(int bsA, int bsB)
	{
		P.delta = k2a_pair16((uint32_t)(bsA - P.baseA) & 0xffffu, (uint32_t)(bsB - P.baseB) & 0xffffu);
	}
	/* a value arriving from the lane above (relative to ITS base): -inf stays -inf, anything else moves to this lane's base */
	K2A_FN k2a_pk adopt(k2a_pk raw) const
	{
		const k2a_pk dead = k2a_pk_sign(k2a_pk_sub(raw, k2a_pku(K2A_PKMP_DEAD)));       /* per half: raw < DEAD */
		return k2a_pk_sel(dead, k2a_pku(K2A_NEG16), k2a_pk_add(raw, P.delta));
	}

	/* v - d per half with -inf kept at -16384 (offset form in and out) */
	K2A_FN static k2a_pk shift(k2a_pk v, k2a_pk d)
	{
		const k2a_pk s = k2a_pk_sub(v, d);                                           /* still in offset form */
		const k2a_pk dead = k2a_pk_sign(k2a_pk_sub(s, k2a_pku(K2A_PKMP_DEAD)));
		return k2a_pk_sel(dead, k2a_pku(K2A_NEG16), s);
	}

	/* merge the window's row maxima into the per-row keys and start a new window (needs the bases the values are relative to) */
	K2A_FN void flush_rowmax()
	{
#pragma unroll
		for (int c = 0; c < C; ++c) {
			const k2a_pk m = k2a_ofs_off(P.rmax(c)), j = P.rmj(c);
			const int mA = k2a_pk_lo(m), mB = k2a_pk_hi(m);
			const uint32_t jA = j & 0xffffu, jB = j >> 16;
			if (mA > K2A_PKMP_DEAD) k2a_key_max(&spill[2 * c], ((unsigned long long)((uint32_t)(mA + P.baseA) ^ 0x80000000u) << 32) | (FIRSTJ ? 0xffffu - jA : jA));
			if (mB > K2A_PKMP_DEAD) k2a_key_max(&spill[2 * c + 1], ((unsigned long long)((uint32_t)(mB + P.baseB) ^ 0x80000000u) << 32) | (FIRSTJ ? 0xffffu - jB : jB));
			P.set_rmax(c, k2a_pku(K2A_NEG16)); P.set_rmj(c, 0);
		}
	}

	/* Re-centre on the largest H of the lane's current column.  Returns the shift {dA, dB} (packed, plain halves); the caller
	 * rotates it one lane down for `after_rebase`.  The window's row maxima move with the base as long as they fit: a row's
	 * maximum drifts away from the current values by a few units per column once the lane has passed the row's best cell, so
	 * it is merged into its key (flush_rowmax: atomics that show up as HBM writes, r2 profiles) only when a half would leave
	 * K2A_PKMP_RMAX_LIMIT -- once or twice per row of 16 k columns instead of every 64 steps. */
	K2A_FN k2a_pk rebase()
	{
		k2a_pk m = P.hl[0];
#pragma unroll
		for (int c = 1; c < C; ++c) m = k2a_pk_maxu(m, P.hl[c]);
		m = k2a_ofs_off(m);
		const k2a_pk dead = k2a_pk_sign(k2a_pk_sub(m, k2a_pk2(K2A_PKMP_DEAD)));
		const k2a_pk d = k2a_pk_sel(dead, 0u, m);                                   /* no live row: stay */
		/* would any row maximum overflow after the shift?  (offset form: plain value = stored ^ OFS) */
		k2a_pk over = 0;
#pragma unroll
		for (int c = 0; c < C; ++c) over |= k2a_pk_sign(k2a_pk_sub(k2a_pk2(K2A_PKMP_RMAX_LIMIT), k2a_pk_sub(k2a_ofs_off(P.rmax(c)), d)));
		if (over != 0) flush_rowmax();                                              /* with the bases the maxima are relative to */
		P.baseA += k2a_pk_lo(d); P.baseB += k2a_pk_hi(d);
#pragma unroll
		for (int c = 0; c < C; ++c) {
			P.hl[c] = shift(P.hl[c], d); P.f[c] = shift(P.f[c], d); if (DUAL) P.f2[c] = shift(P.f2[c], d);
			P.set_rmax(c, shift(P.rmax(c), d));
		}
		P.hd0 = shift(P.hd0, d); P.hout = shift(P.hout, d); P.eout = shift(P.eout, d);
		if (DUAL) P.e2out = shift(P.e2out, d);
		return d;
	}
	/* d_above = the shift the lane above just made (its outputs, which this lane holds raw in hu_prev, moved by it) */
	K2A_FN void after_rebase(k2a_pk d_above, int bsA, int bsB)
	{
		P.hu_prev = shift(P.hu_prev, d_above);
		refresh_delta(bsA, bsB);
	}

	/* The strip's last column is done: the scalar reference's per-row epilogue (K2aLane::do_fin; ksw2_extz.c:116-124,
	 * ksw2_extd.c:156-164) for both alignments on int32 values: H at the last column from the registers, the row maximum
	 * and its column from the keys.  The caller has ordered the key atomics before this (fence). */
	K2A_FN void do_fin(const K2aScoring &sc, K2aBook *bA, K2aBook *bB, int zdropA, int zdropB, uint32_t *rowbuf)
	{
		const int zslope = DUAL ? sc.e2 : sc.e;
#pragma unroll
		for (int c = 0; c < C; ++c) rowbuf[c] = k2a_ofs_off(P.hl[c]);
#pragma nounroll
		for (int half = 0; half < 2; ++half) {
			K2aBook *b = half ? bB : bA;
			const int zdrop = half ? zdropB : zdropA, base = half ? P.baseB : P.baseA, sh = half ? 16 : 0;
			int bmax = b->max, bmax_t = b->max_t, bmax_q = b->max_q, bmqe = b->mqe, bmqe_t = b->mqe_t;
			int bmte = b->mte, bmte_q = b->mte_q, bscore = b->score, bdrop = b->dropped, brows = b->rows;
#pragma nounroll
			for (int c = 0; c < C; ++c) {
				const int i = P.i0 + c;
				if (i < P.tlen && !bdrop) {
					const bool reach = i + P.w >= P.qlen - 1;
					const unsigned long long key = k2a_key_load(&spill[2 * c + half]);
					const int hend = (int)(int16_t)(rowbuf[c] >> sh) + base - sc.e * i;
					const int H = (int)((uint32_t)(key >> 32) ^ 0x80000000u) - sc.e * i;
					const int j = FIRSTJ ? 0xffff - (int)(key & 0xffffu) : (int)(key & 0xffffu);
					if (reach && hend > bmqe) { bmqe = hend; bmqe_t = i; }
					if (i == P.tlen_full - 1) { bmte = H; bmte_q = j; }
					if (H > bmax) { bmax = H; bmax_t = i; bmax_q = j; }
					else if (i >= bmax_t && j >= bmax_q) {
						const int dt = i - bmax_t, dq = j - bmax_q;
						const int skew = dt > dq ? dt - dq : dq - dt;
						if (zdrop >= 0 && bmax - H > zdrop + skew * zslope) bdrop = 1;
					}
					if (!bdrop && i == P.tlen_full - 1 && reach) bscore = hend;
					brows = i + 1;
				}
			}
			b->max = bmax; b->max_t = bmax_t; b->max_q = bmax_q; b->mqe = bmqe; b->mqe_t = bmqe_t;
			b->mte = bmte; b->mte_q = bmte_q; b->score = bscore; b->dropped = bdrop; b->rows = brows;
		}
		P.end_strip();
	}
};

/* lag, in phases of K2A_PKMP_T steps, by which generation g + 1 must start after generation g so that (a) every boundary
 * column it reads was written in an earlier phase and (b) its first strip epilogue comes in a later phase than the last one of
 * generation g (ksw2_shim_hip.hip).  jlo_p / jlo_c = first columns of the two generations. */
K2A_FN int k2a_pkmp_lag(int jlo_p, int jlo_c) { return (jlo_c - jlo_p + K2A_PKMP_T + K2A_PKMP_T - 1) / K2A_PKMP_T + 1; }

#endif
