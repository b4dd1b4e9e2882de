/*
 * ksw2_oracle_extf.c -- CPU restatement of ksw_extf2_sse (gap-linear X-drop extension, score only).
 *
 * TEST INFRASTRUCTURE ONLY (see ksw2_oracle.h): never linked into the product.
 * Parity status: PINNED against the unmodified reference compiled with gcc -O2 -msse4.1 (oracle/_ref), golden vectors
 * tests/golden/extf_cases.npz (generator oracle/gen_golden_extf.py), checked by tests/test_oracle_extf.py.
 *
 * The reference (ksw2_extf2_sse.c:11-98) has no scalar definition, and what it returns depends on how its SSE loops are
 * blocked, so this restatement follows the vector code's *memory image* position by position instead of a recurrence:
 *
 *   - three byte arrays over target positions, padded to a multiple of 16 (ksw2_extf2_sse.c:21-28): U, V (the difference
 *     encoding of the anti-diagonal DP, 8-bit wrapping arithmetic) and S (match / mismatch score of the position on the
 *     current anti-diagonal);
 *   - on anti-diagonal r the in-band positions are [lo, hi] (ksw2_extf2_sse.c:35-41), but the update loop runs over the
 *     16-aligned blocks around them (:42, :62-79), so positions outside [lo, hi] are updated too, from whatever S holds
 *     there: S is refreshed in chunks of 16 starting at lo (:48-61), i.e. on [lo, lo + 16 * ceil((hi - lo + 1) / 16)),
 *     from target codes that read 0 past the target's end and query codes that read 0 before the query's start; anything
 *     older stays.  Those out-of-band values are what the band's edge cells read on later anti-diagonals;
 *   - refresh writes past the padded S array land in the reference's copy of the target, at positions below lo that are
 *     never read again (lo never decreases): dropped here;
 *   - the score follows ONE cell per anti-diagonal, greedily (ksw2_extf2_sse.c:80-92), and the X-drop test is on that
 *     cell; leaving the band before the last anti-diagonal is reported as a drop (:37, :95-96).
 *
 * Written for the SSE4.1 build of the reference (signed byte maximum at :70); the SSE2 fallback at :72-73 clamps negative
 * sums to 0 first, which is the same thing whenever the shifted V value is below 128.
 */
#include <stdlib.h>
#include <string.h>
#include "ksw2_oracle.h"

static void extf_reset(kso_extz_t *ez)      /* ksw2.h:184-189 */
{
	ez->max_q = ez->max_t = ez->mqe_t = ez->mte_q = -1;
	ez->max = 0; ez->score = ez->mqe = ez->mte = KSO_NEG_INF;
	ez->n_cigar = 0; ez->zdropped = 0; ez->reach_end = 0;
}

void kso_extf2(int qlen, const uint8_t *query, int tlen, const uint8_t *target, int8_t mch, int8_t mis, int8_t e, int w, int xdrop,
               kso_extz_t *ez)
{
	const int tpad = (tlen + 15) / 16 * 16;
	const uint8_t sc_match = (uint8_t)mch, sc_mism = (uint8_t)(mis < 0 ? mis : -mis), two_e = (uint8_t)(e * 2);
	uint8_t *U, *V, *S;
	int r, prev_lo = -1, prev_hi = -1, follow = 0, done = 0;
	int32_t H0 = 0;

	extf_reset(ez);
	if (w < 0) w = tlen > qlen ? tlen : qlen;
	U = (uint8_t*)calloc((size_t)3 * tpad + 16, 1);
	V = U + tpad; S = V + tpad;

	for (r = 0; r < qlen + tlen - 1; ++r) {
		int lo = 0, hi = tlen - 1, blo, bhi, x, fresh_end;
		uint8_t carry;
		if (lo < r - qlen + 1) lo = r - qlen + 1;
		if (hi > r) hi = r;
		if (lo < ((r - w + 1) >> 1)) lo = (r - w + 1) >> 1;      /* arithmetic shift: floor, also for negative values */
		if (hi > ((r + w) >> 1)) hi = (r + w) >> 1;
		if (lo > hi) break;
		blo = lo & ~15; bhi = hi | 15;
		carry = (blo > 0 && blo - 1 >= prev_lo && blo - 1 <= prev_hi) ? V[blo - 1] : 0;
		if (bhi >= r) U[r] = 0;
		fresh_end = lo + ((hi - lo) / 16 + 1) * 16;
		if (fresh_end > tpad) fresh_end = tpad;
		for (x = lo; x < fresh_end; ++x) {
			const uint8_t tc = x < tlen ? target[x] : 0;
			const int j = r - x;
			const uint8_t qc = (j >= 0 && j < qlen) ? query[j] : 0;
			S[x] = tc == qc ? sc_match : sc_mism;
		}
		for (x = blo; x <= bhi; ++x) {
			const uint8_t a = carry, b = U[x];
			uint8_t z = (uint8_t)(S[x] + two_e);
			carry = V[x];
			if ((int8_t)z < (int8_t)a) z = a;
			if (z < b) z = b;
			U[x] = (uint8_t)(z - a);
			V[x] = (uint8_t)(z - b);
		}
		if (r > 0) {
			const int in0 = follow >= lo && follow <= hi, in1 = follow + 1 >= lo && follow + 1 <= hi;
			if (in0 && in1) {
				const int32_t d0 = V[follow] - e, d1 = U[follow + 1] - e;
				if (d0 > d1) H0 += d0;
				else { H0 += d1; ++follow; }
			} else if (in0) H0 += V[follow] - e;
			else { ++follow; H0 += U[follow] - e; }
			if (H0 > (int32_t)ez->max) { ez->max = (uint32_t)H0; ez->max_t = follow; ez->max_q = r - follow; }
			else if (xdrop >= 0 && (int32_t)ez->max - H0 > xdrop) break;
		} else { H0 = V[0] - e - e; follow = 0; }
		prev_lo = blo; prev_hi = bhi;
	}
	done = r == qlen + tlen - 1;
	if (done) ez->score = H0;
	else ez->zdropped = 1;
	free(U);
}
