/*
 * ksw2_lane_pk.h -- packed-int16 variant of the per-lane kernel code: every lane carries TWO alignments of
 * identical shape (qlen, tlen, w) in the low / high halves of its 32-bit registers.
 *
 * Same schedule, band masks and row bookkeeping as K2aLane (ksw2_lane.h; replaces the same reference loops,
 * ksw2_extz2_sse.c:101-289 / ksw2_extd2_sse.c:131-387, with the scalar ksw_extz / ksw_extd semantics), but the
 * Gotoh cell update runs on v_pk_add/sub/max_i16: one VALU instruction advances two cells, and every per-step
 * control instruction (live mask, schedule, DPP rotate, loop) is shared by the two alignments.
 *
 * Row bias: every value of target row i (H, E, F and the second-piece E~, F~) is stored with + e*i added.  Along a
 * column the first-piece E then needs no extension subtract at all -- E'(i+1,j) = max(E'(i,j), H'(i,j) - q) -- the
 * diagonal absorbs the bias into the score constants (s + e), and comparisons inside a row are unaffected; strip
 * epilogues subtract e*i again.  One instruction less per cell.
 *
 * Re-based variant (RB = true) for reads of any length: every strip stores its values relative to a per-alignment base
 * = the (biased) H of the cell diagonally above its first cell, so only the score spread inside a strip's window --
 * at most (2w + 2C)(smax + q + e), a property of the band, not of the read length -- has to fit 16 bits.  The bottom
 * row handed to the next lane is shifted by the difference of the two bases (`delta`, one v_pk_add per value per step),
 * and the strip epilogue adds the base back.  Re-based launches always use the sequential epilogue.
 *
 * Number format inside the fill loop: every DP value v is held as v + 0x4C00 per half ("offset" form, K2A_OFS), so that the
 * whole working range -- -inf = -16384 with 3072 units of slack below it, real values up to +12287 -- maps onto the bit
 * patterns 0x0000 .. 0x7BFF.  Order is preserved under unsigned compare (v_pk_max_u16), differences of two values are
 * unchanged, adding or subtracting a small non-negative constant can be ONE 32-bit v_add_u32 / v_sub_u32 for both halves
 * (the low half can neither carry nor borrow while values stay inside the window), and -- new in round 3 -- those patterns
 * are exactly the non-negative finite IEEE halves, whose float order is their integer order: gfx950's
 * v_pk_maximum3_f16 returns the selected operand bit for bit (denormals included; tools/probe/max3_probe.hip,
 * profiles/r3_max3_probe.txt), so H = max(cand, E, F) is ONE instruction (5.1 cycles at four wavefronts per SIMD) instead of
 * two v_pk_max_u16 (2 x 4.4).  A pattern outside 0x0000 .. 0x7BFF (only garbage that a band mask discards anyway) yields
 * garbage, which the same mask discards.  On gfx950 v_add_u32, v_sub_u32,
 * v_xor_b32, v_bitop3_b32 issue a wave64 in 2 cycles, every v_pk_*_i16 (and v_bfi, v_bfe, v_max_i32) in 4
 * (tools/probe/valu_rate.hip, profiles/r1d_valu_rate.txt).  Cold code (init, epilogues) converts at its edges.
 *
 * Scores (round 5: "column profiles", any matrix -- ksw2_extz2_sse.c:142-143 / ksw2_extd2_sse.c:182-183 take any m x m `mat` at
 * full rate, and so do these kernels for m <= 5): a score is held as its penalty below the matrix's largest entry, smax - s(t, q),
 * one byte.  Per step a lane looks up the COLUMN profile of its query code -- cp[q] = the four penalties against target codes
 * 0..3, one dword from an 8-entry table in LDS (K2aScoring.cp), fetched one step ahead -- for each of its two alignments; per row
 * it keeps ONE register, the byte selector { tA, 0x0c, 4 + tB, 0x0c } formed when the strip starts, and the diagonal candidate is
 *     H(i-1,j-1) + (smax + e) - v_perm_b32(cpB, cpA, selector[row])
 * : three instructions, 9 cycles, for any matrix (rounds 1-4: two bit planes of the target codes x (match - mismatch), xor + v_bitop3,
 * four instructions, two registers per row, match / mismatch scoring only; the traceback kernels compare + multiply-add, 15 cycles).
 * The query's wildcard (code 4) is table entry 4.  A TARGET code of 4 has no byte in a four-byte profile: such pairs take the
 * int32 kernels (the host's scan finds them; kernels of unscanned plans report them, K2aLanePk::seen).
 *
 * Preconditions, checked by the host (ksw2_host_plan.c::pk_eligible / pk_window_ok): m <= 5, no wildcard code in the TARGET, gap
 * costs and smax + e non-negative, and every in-band H, E, F provably inside (-16384 + max(q+e, q2+e2), 12287 - max(q+e, q2+e2))
 * so that -16384 can stand for -infinity (K2A_PK_VMAX = 12287 is the largest value the offset form holds, ksw2_types.h).
 */
#ifndef KSW2_LANE_PK_H_
#define KSW2_LANE_PK_H_

#include "ksw2_lane.h"

#define K2A_PK_STAGE(C) (3 * (C) + 4)    /* LDS words per lane group for a strip's staged rows: H, row max, arg-max; first row, bases */
typedef uint32_t k2a_pk;                 /* { int16 lo = alignment A, int16 hi = alignment B } */
#define K2A_OFS   (K2A_OFS16 * 0x10001u) /* K2A_OFS16, K2A_NEG16, K2A_PK_VMAX, K2A_PK_SLACK: ksw2_types.h (the host's range checks use them) */

K2A_FN k2a_pk k2a_pk2(int v) { return ((uint32_t)v & 0xffffu) | ((uint32_t)v << 16); }
K2A_FN int k2a_pk_lo(k2a_pk v) { return (int)(int16_t)(v & 0xffffu); }
K2A_FN int k2a_pk_hi(k2a_pk v) { return (int)(int16_t)(v >> 16); }
K2A_FN k2a_pk k2a_pk_sel(k2a_pk m, k2a_pk a, k2a_pk b) { return (m & a) | (~m & b); }   /* v_bfi / v_bitop3 */
K2A_FN k2a_pk k2a_pair16(uint32_t lo, uint32_t hi) { return lo | (hi << 16); }                /* two small codes -> halves */
K2A_FN k2a_pk k2a_byte_pair(uint32_t a, uint32_t b, int r)                                      /* { byte r of a, byte r of b } -> halves; r constant */
{
#if defined(__HIP_DEVICE_COMPILE__)
	return __builtin_amdgcn_perm(b, a, 0x0c040c00u + (uint32_t)r * 0x00010001u);
#else
	return ((a >> (8 * r)) & 0xffu) | (((b >> (8 * r)) & 0xffu) << 16);
#endif
}
/* v_perm_b32 with a register selector: result byte k = byte sel[k] of the eight bytes { s1 (0..3), s0 (4..7) }; selector values
 * 8..11 = 0x00 / 0xff by the sign bit of byte 1 / 3 / 5 / 7, 12 = 0x00, >= 13 = 0xff (ISA: V_PERM_B32).  The simulator build runs
 * the C twin; tools/probe/perm_probe.hip compares the two on the device for every selector value. */
K2A_FN uint32_t k2a_perm(uint32_t s0, uint32_t s1, uint32_t sel)
{
#if defined(__HIP_DEVICE_COMPILE__)
	return __builtin_amdgcn_perm(s0, s1, sel);
#else
	const uint64_t src = ((uint64_t)s0 << 32) | s1;
	uint32_t r = 0;
	for (int k = 0; k < 4; ++k) {
		const uint32_t x = (sel >> (8 * k)) & 0xffu;
		uint32_t b;
		if (x >= 13) b = 0xffu;
		else if (x == 12) b = 0u;
		else if (x >= 8) b = ((src >> (16 * (x - 8) + 15)) & 1u) ? 0xffu : 0u;
		else b = (uint32_t)(src >> (8 * x)) & 0xffu;
		r |= b << (8 * k);
	}
	return r;
#endif
}
#define K2A_TSEL_BASE 0x0c040c00u           /* row selector = { tA, 0x0c, 4 + tB, 0x0c }: byte tA of cpA -> low half, byte tB of cpB -> high half */
K2A_FN uint32_t k2a_h16(int v) { return (uint32_t)(v + K2A_OFS16) & 0xffffu; }                 /* one half in offset form */
K2A_FN k2a_pk k2a_pku(int v) { return k2a_pk2(v + K2A_OFS16); }                                 /* constant in offset form */
/* both halves at once with one 32-bit op: exact as long as the low half neither carries nor borrows (offset form,
 * non-negative addend); the simulator build runs the very same 32-bit arithmetic, so a violated range shows up there */
K2A_FN k2a_pk k2a_add32(k2a_pk a, k2a_pk b) { return a + b; }
K2A_FN k2a_pk k2a_sub32(k2a_pk a, k2a_pk b) { return a - b; }

#if defined(__HIP_DEVICE_COMPILE__)
typedef short k2a_s2 __attribute__((ext_vector_type(2)));
typedef unsigned short k2a_u2 __attribute__((ext_vector_type(2)));
K2A_FN k2a_pk k2a_pk_add(k2a_pk a, k2a_pk b) { return __builtin_bit_cast(k2a_pk, (k2a_s2)(__builtin_bit_cast(k2a_s2, a) + __builtin_bit_cast(k2a_s2, b))); }
K2A_FN k2a_pk k2a_pk_sub(k2a_pk a, k2a_pk b) { return __builtin_bit_cast(k2a_pk, (k2a_s2)(__builtin_bit_cast(k2a_s2, a) - __builtin_bit_cast(k2a_s2, b))); }
K2A_FN k2a_pk k2a_pk_max(k2a_pk a, k2a_pk b) { return __builtin_bit_cast(k2a_pk, __builtin_elementwise_max(__builtin_bit_cast(k2a_s2, a), __builtin_bit_cast(k2a_s2, b))); }
K2A_FN k2a_pk k2a_pk_min(k2a_pk a, k2a_pk b) { return __builtin_bit_cast(k2a_pk, __builtin_elementwise_min(__builtin_bit_cast(k2a_s2, a), __builtin_bit_cast(k2a_s2, b))); }
K2A_FN k2a_pk k2a_pk_maxu(k2a_pk a, k2a_pk b) { return __builtin_bit_cast(k2a_pk, __builtin_elementwise_max(__builtin_bit_cast(k2a_u2, a), __builtin_bit_cast(k2a_u2, b))); }
K2A_FN k2a_pk k2a_pk_max3u(k2a_pk a, k2a_pk b, k2a_pk c)   /* per half max of three offset-form values (patterns 0 .. 0x7BFF) */
{
	k2a_pk d;
	asm("v_pk_maximum3_f16 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
	return d;
}
K2A_FN k2a_pk k2a_pk_mad(k2a_pk a, k2a_pk b, k2a_pk c)
{
	k2a_pk d;
	asm("v_pk_mad_i16 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
	return d;
}
K2A_FN k2a_pk k2a_pk_sign(k2a_pk a)     /* per half: 0xffff if negative else 0 */
{
	k2a_pk d;
	asm("v_pk_ashrrev_i16 %0, 15, %1 op_sel_hi:[0,1]" : "=v"(d) : "v"(a));
	return d;
}
K2A_FN k2a_pk k2a_pk_selv(k2a_pk m, k2a_pk a, k2a_pk b)   /* k2a_pk_sel on three registers */
{
	k2a_pk d;      /* asm: with the mask coming out of the asm above, hipcc expands the select into and / not / and / or.
	                * v_bitop3_b32 issues in 2 cycles per wave64, v_bfi_b32 in 4 (profiles/r1d_valu_rate.txt); 0xe4 = src2 ? src0 : src1 */
	asm("v_bitop3_b32 %0, %1, %2, %3 bitop3:0xe4" : "=v"(d) : "v"(a), "v"(b), "v"(m));
	return d;
}
K2A_FN k2a_pk k2a_pk_shl(k2a_pk a, int n)   /* per half, n a compile-time constant */
{
	return n == 0 ? a : __builtin_bit_cast(k2a_pk, (k2a_u2)(__builtin_bit_cast(k2a_u2, a) << (k2a_u2)(unsigned short)n));
}
K2A_FN k2a_pk k2a_bit_mask(uint32_t bits, int c) { return (k2a_pk)__builtin_amdgcn_sbfe((int)bits, c, 1); }
#define K2A_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
/* Loads whose result is needed many steps later.
 * k2a_load_early: a plain dword load (a volatile one becomes a system-scope flat load with its wait right behind it); its
 * s_waitcnt lands in front of the first use as for any load.  For values DEFINED UNCONDITIONALLY in the block that issues
 * them (the query dword of the next group of four steps: issued at the top of a group, used below the group's steps).  A load into
 * a variable that is set under a condition and lives around the step loop must not be written this way: hipcc copies the loaded
 * register right behind the load (the phi of "loaded now / kept from before") and therefore waits there -- which is how every
 * kernel's query "prefetch" used to be a blocking load (s_waitcnt vmcnt(0) two instructions after global_load_dword: 23-26 % of a
 * lone wavefront's time, profiles/r3_solo_experiments.txt).
 * (An inline-asm load that the compiler does not know as one, with a hand-placed s_waitcnt, does not work for those either:
 * the register allocator copies the in-flight register at the loop's back edge -- tools/scripts/async_load_audit.py found that
 * in every kernel it was tried in.) */
K2A_FN uint32_t k2a_load_early(const uint8_t *p) { uint32_t v; __builtin_memcpy(&v, p, 4); return v; }
#else
#define K2A_SCHED_FENCE() do {} while (0)
K2A_FN uint32_t k2a_load_early(const uint8_t *p) { uint32_t v; __builtin_memcpy(&v, p, 4); return v; }
K2A_FN k2a_pk k2a_pk_mk(int lo, int hi) { return ((uint32_t)lo & 0xffffu) | ((uint32_t)hi << 16); }
K2A_FN k2a_pk k2a_pk_add(k2a_pk a, k2a_pk b) { return k2a_pk_mk(k2a_pk_lo(a) + k2a_pk_lo(b), k2a_pk_hi(a) + k2a_pk_hi(b)); }
K2A_FN k2a_pk k2a_pk_sub(k2a_pk a, k2a_pk b) { return k2a_pk_mk(k2a_pk_lo(a) - k2a_pk_lo(b), k2a_pk_hi(a) - k2a_pk_hi(b)); }
K2A_FN k2a_pk k2a_pk_max(k2a_pk a, k2a_pk b) { return k2a_pk_mk(k2a_max(k2a_pk_lo(a), k2a_pk_lo(b)), k2a_max(k2a_pk_hi(a), k2a_pk_hi(b))); }
K2A_FN k2a_pk k2a_pk_min(k2a_pk a, k2a_pk b) { return k2a_pk_mk(k2a_min(k2a_pk_lo(a), k2a_pk_lo(b)), k2a_min(k2a_pk_hi(a), k2a_pk_hi(b))); }
K2A_FN k2a_pk k2a_pk_maxu(k2a_pk a, k2a_pk b)
{
	const uint32_t al = a & 0xffffu, ah = a >> 16, bl = b & 0xffffu, bh = b >> 16;
	return (al > bl ? al : bl) | ((ah > bh ? ah : bh) << 16);
}
/* what v_pk_maximum3_f16 computes on the patterns the kernels feed it; outside them the hardware compares them as floats
 * (negative numbers, NaNs): the simulator returns a poison pattern instead, so that a range violation which reaches a live
 * cell cannot go unnoticed on the CPU test tier */
K2A_FN k2a_pk k2a_pk_max3u(k2a_pk a, k2a_pk b, k2a_pk c)
{
	uint32_t r = 0;
	for (int h = 0; h < 2; ++h) {
		const uint32_t x = (a >> (16 * h)) & 0xffffu, y = (b >> (16 * h)) & 0xffffu, z = (c >> (16 * h)) & 0xffffu;
		uint32_t m = x > y ? x : y;
		m = m > z ? m : z;
		if (m > 0x7BFFu) m = 0x7BFFu - (m & 0xffu);            /* poison: a large, wrong, in-range value */
		r |= m << (16 * h);
	}
	return r;
}
K2A_FN k2a_pk k2a_pk_mad(k2a_pk a, k2a_pk b, k2a_pk c) { return k2a_pk_mk(k2a_pk_lo(a) * k2a_pk_lo(b) + k2a_pk_lo(c), k2a_pk_hi(a) * k2a_pk_hi(b) + k2a_pk_hi(c)); }
K2A_FN k2a_pk k2a_pk_sign(k2a_pk a) { return ((a & 0x8000u) ? 0xffffu : 0u) | ((a & 0x80000000u) ? 0xffff0000u : 0u); }
K2A_FN k2a_pk k2a_pk_selv(k2a_pk m, k2a_pk a, k2a_pk b) { return k2a_pk_sel(m, a, b); }
K2A_FN k2a_pk k2a_pk_shl(k2a_pk a, int n) { return (((a & 0xffffu) << n) & 0xffffu) | ((((a >> 16) << n) & 0xffffu) << 16); }
K2A_FN k2a_pk k2a_bit_mask(uint32_t bits, int c) { return ((bits >> c) & 1u) ? 0xffffffffu : 0u; }
#endif

/* Target wildcard rows (K2aScoring.pk_tn1, round 6).  x = { tA, 0, tB, 0 }, the raw codes of a row:
 * k2a_tsel_wild: the row's selector with 0x0c ("byte 0x00": penalty 0) where the code has bit 2 set, the ordinary one elsewhere;
 * k2a_tn_fix: what such a row's candidate loses instead -- bit 3 of a selector byte marks the half (ordinary bytes are 0..7);
 * k2a_codes_above4: nonzero if a byte of a code dword is above 4 (an OR over dwords cannot tell 4 | 1 from 5). */
K2A_FN uint32_t k2a_tsel_wild(uint32_t x)
{
	const uint32_t n = x & 0x00040004u;
	return ((x ^ n) + K2A_TSEL_BASE) | n | (n << 1);
}
K2A_FN k2a_pk k2a_tn_fix(k2a_pk cand, uint32_t sel, int tn1)
{
	const uint32_t m = (sel >> 3) & 0x00010001u;                  /* 1 per half that holds a wildcard row */
	return cand - m * ((uint32_t)(tn1 - 1) & 0xffu);              /* both halves with one 32-bit subtract (k2a_sub32: the low half does not borrow) */
}
K2A_FN uint32_t k2a_codes_above4(uint32_t d) { return (d & 0xf8f8f8f8u) | ((d >> 2) & (d | (d >> 1)) & 0x01010101u); }

/* Direction flags of one row for both halves (traceback kernels, round 5).  Every decision of the reference's direction logic
 * (ksw2_extz.c:72-86 / 98-112, ksw2_extd.c:88-114 / 126-152) is the SIGN of a packed difference: s1..s4 = candidate - gap state in
 * the order E, F, E~, F~ (each against the running maximum before it: "the gap state wins"), x1..x4 = opening - gap state ("the gap
 * leaving the cell is an extension").  Rounds 1-4 turned every sign into a mask (v_pk_ashrrev_i16), positioned it (v_and) and merged
 * it (v_or / v_bitop3): three instructions per decision.  v_perm_b32's selectors 8..11 deliver the sign of a 16-bit half as a whole
 * byte: ONE v_perm_b32 gathers the four sign bytes of two differences { s.A, s.B, x.A, x.B }, bit-interleaving v_bitop3_b32's
 * (2.5 cycles each) merge them, and one shifted merge folds the x-bytes onto the s-bytes:
 *   result byte 0 = alignment A, byte 1 = alignment B: bit 0 E wins, 1 F wins, [2 E~ wins, 3 F~ wins,] then the extension flags
 *   E, F [, E~, F~] -- at bits 2, 3 (single gap, four-bit form; the byte's upper nibble is garbage) or 4..7 (one byte per cell;
 *   single gap: bits 2, 3, 6, 7 cleared).  Bytes 2 and 3 are garbage.
 * The winner is the HIGHEST set win flag (a later gap state was compared against the maximum that already included the earlier
 * ones); k2a_flags_decode turns a flag byte into the reference's code (ksw2.h:125-128) for the walks.  Right-aligned mode compares
 * the other way round (gap state - candidate: set = the gap state does NOT win) and inverts the word before it is stored. */
#define K2A_SGN_SEL 0x09080b0au             /* v_perm_b32(s, x, .) -> { sign(s.lo), sign(s.hi), sign(x.lo), sign(x.hi) } as 0x00 / 0xff bytes */
template<bool DUAL, bool NIBBLE, bool INV>
K2A_FN uint32_t k2a_dir_flags(uint32_t s1, uint32_t s2, uint32_t s3, uint32_t s4, uint32_t x1, uint32_t x2, uint32_t x3, uint32_t x4)
{
	const uint32_t q1 = k2a_perm(s1, x1, K2A_SGN_SEL), q2 = k2a_perm(s2, x2, K2A_SGN_SEL);
	uint32_t r = k2a_pk_selv(0x55555555u, q1, q2);                          /* per byte: even bits E, odd bits F */
	if (DUAL) {
		const uint32_t q3 = k2a_perm(s3, x3, K2A_SGN_SEL), q4 = k2a_perm(s4, x4, K2A_SGN_SEL);
		r = k2a_pk_selv(0x33333333u, r, k2a_pk_selv(0x55555555u, q3, q4));  /* bits 0-1 first piece, 2-3 second piece, repeated up the byte */
	}
	if (INV) r = ~r;
	if (NIBBLE) return k2a_pk_selv(0x03030303u, r, r >> 16);                /* win flags at bits 0-1, extension flags at 2-3 */
	const uint32_t t = k2a_pk_selv(0x0f0f0f0fu, r, r >> 16);                /* win flags in the low nibble, extension flags in the high one */
	return DUAL ? t : t & 0x33333333u;
}
K2A_FN uint32_t k2a_flags_decode(uint32_t b)                               /* flag byte -> the reference's direction byte (ksw2.h:125-128) */
{
	const uint32_t win = (b & 8u) ? 4u : (b & 4u) ? 3u : (b & 2u) ? 2u : (b & 1u);
	return win | ((b >> 4) << 3);
}
/* plain int16 halves <-> offset form (cold code: strip prologues and epilogues) */
K2A_FN k2a_pk k2a_ofs_on(k2a_pk plain) { return k2a_pk_add(plain, K2A_OFS); }
K2A_FN k2a_pk k2a_ofs_off(k2a_pk ofs) { return k2a_pk_sub(ofs, K2A_OFS); }

/* packed traceback: one byte per cell and alignment in the reference's own layout (ksw2.h:125-128): bits 0-2 winner
 * {0 diag, 1 E, 2 F, 3 E~, 4 F~}, 0x08/0x10/0x20/0x40 = the E/F/E~/F~ gap leaving the cell is an extension.
 * A lane-step word holds C cells x 2 alignments: byte 2c = alignment A, byte 2c+1 = alignment B. */
/* NOMAX: KSW_EZ_APPROX_MAX launches (ksw2_host_int.h::is_approx) need the final score and the direction bytes only: no row
 * maximum, no arg-max, no Z-drop -- four instructions per row and two registers per row less. */
/* Classes whose per-row maxima, arg-max columns and target codes live in LDS instead of registers (3 x C dwords per
 * lane, [array][row][lane]): the two-piece traceback kernels of the 16-row geometry, which otherwise need 290-330
 * registers and run one wavefront per SIMD.  The values are touched once per row and step, so the LDS traffic is a few
 * per cent of the step; the kernel then fits two wavefronts.  With a single wavefront on a SIMD the loads' latency is
 * exposed and the register form is faster, so both are built and the launcher picks by the number of tasks. */
#define K2A_PK_LDSROWS(G, C, DUAL, MODE, NOMAX) ((G) == 64 && (C) == 16 && (DUAL) && (MODE) != K2A_MODE_SCORE && !(NOMAX))
#define K2A_PK_LDSROW_WORDS(C) (3 * (C) * 64)        /* per wavefront */
/* Codes only (LDSROW_ = 2): the exact score-only kernels held 178-224 registers, two wavefronts per SIMD, with the rows' target
 * codes in registers; with them (round 5: ONE selector per row) in LDS the 16-row geometry fits four, the (8, 18) geometry of
 * short reads three.  Nothing else changes: the selectors are read-only between strip starts, one LDS load per row and step. */
#define K2A_PK_LDSCODES(G, C, DUAL, MODE, NOMAX) ((((G) == 64 && (C) == 16) || ((G) == 8 && (C) == 18) || ((G) == 16 && (C) == 8)) && !(DUAL) && (MODE) == K2A_MODE_SCORE)
#define K2A_PK_LDSCODE_WORDS(C) ((C) * 64)           /* per wavefront */

/* DEFER (exact score-only kernels): the fill tracks every row's maximum but not its column -- three of the fifteen instructions
 * of a row pair, 11 of 51 cycles.  Columns are needed for max_q (the row that holds the alignment's maximum), for mte_q (the last
 * target row) and for the skew term of a Z-drop test; so the fill also streams the two values every lane receives from the strip
 * above ({H, E} of its top neighbour, 8 bytes per lane and step, 512 contiguous bytes per wavefront and step) into a checkpoint
 * block, a second small kernel (k2a_argmax_kernel) re-runs exactly the one or two strips per alignment whose columns are asked for --
 * same lane code, arg-max on, inputs from the checkpoint --, and an alignment in which a Z-drop cannot be ruled out without the
 * skew term (max - H > zdrop somewhere) has its book frozen at that row; a third kernel (k2a_zscan_kernel, round 5) re-runs the
 * strips from there on, sixteen at a time, and folds their rows into the book with the reference's exact test until the drop or
 * the last row.  (Rounds 3-4 reported such an alignment as "inexact" and the host ran it again through the ordinary kernels.) */
struct K2aCkHead { int32_t baseA, baseB; uint32_t hd0, pad; };          /* per strip: what do_init derived from the neighbour lane */
#define K2A_CK_STEP_BYTES 512                                            /* 64 lanes x { hin, ein } */

/* TN: the build for wavefront-tasks whose targets hold a wildcard code (K2aScoring.pk_tn1; the kernels look at their targets first and
 * take this build or the plain one as a whole -- a test inside the plain step costs every batch 1-3 %, 10 % at one wavefront per SIMD:
 * profiles/r6_ab_tn.txt) */
template<int G, int C, bool DUAL, int MODE = K2A_MODE_SCORE, bool RB = false, bool NOMAX = false, int LDSROW_ = 0, bool DEFER = false, bool TN = false>
struct K2aLanePk {
	enum { NIB = K2A_PK_NIBBLES(C, DUAL), TBWORDS = NIB ? C / 4 : C / 2 };
	/* NIB (single gap, 16 rows): direction flags of 4 bits (k2a_dir_flags: bit 0 E wins, 1 F wins, 2 / 3 = the E / F gap leaving the
	 * cell is an extension) -- four consecutive rows per 16-bit half (row 4g at bits 3-0 ... row 4g+3 at bits 15-12), word g of a
	 * lane-step = { alignment A, alignment B }: half the traceback bytes of the byte layout */
	/* group-uniform (both alignments share the shape) */
	int qlen, tlen, tlen_full, w, nstrips;
	const uint8_t *qa, *qbp, *ta, *tbq;   /* query / target codes of alignment A and of alignment B */
	/* schedule, identical to K2aLane */
	int gl, S, i0, je, koff, Snext, knext, koff_next;
	int kfin, kd, rows_m1, wup;         /* last step of the strip, koff + i0, live-row clamp (-1 = no strip), band reach upwards */
	/* systolic ports */
	k2a_pk hout, eout, e2out, hd0, hu_prev;
	int baseA, baseB;                   /* RB: absolute (row-biased) score that the strip's packed values are relative to */
	k2a_pk delta;                       /* RB: base of the strip above minus this strip's base, added to incoming ports */
	uint32_t qb;                        /* { query code A, query code B } of this step's column */
	/* TN builds: hasn = this lane's strip holds a target wildcard row; wn = some lane of the wavefront does (the kernels refresh it
	 * where strips start and end: wavefront-uniform, so that step() tests a scalar) */
	uint32_t hasn;
	bool wn;
	const uint32_t *cptab;              /* the column profiles (K2aScoring.cp), in LDS on the device */
	uint32_t cpA, cpB;                  /* cp[query code] of this step's column for alignment A / B: penalties against target codes 0..3 */
	/* the step's query codes and their column profiles.  The kernels call it for step k + 1 right behind step k (the table
	 * look-ups are then a step old when the first row needs them), and again behind a do_init that changed the lane's codes */
	K2A_FN void set_qb(uint32_t q) { qb = q; cpA = cptab[q & 7u]; cpB = cptab[(q >> 16) & 7u]; }
	/* rows */
	enum { LDSROW = LDSROW_ == 1,       /* row maxima, arg-max columns and target selectors in LDS */
	       LDSTC = LDSROW_ != 0 };      /* 2: only the target selectors */
	k2a_pk hl[C], f[C], f2[DUAL ? C : 1], rmax_[(NOMAX || LDSROW) ? 1 : C], rmj_[(NOMAX || LDSROW || DEFER) ? 1 : C];       /* hl, f, f2, rmax and the ports above: offset form */
	k2a_pk tc_[LDSTC ? 1 : C];                                   /* per row: the v_perm_b32 selector of the rows' target codes {A, B} (K2A_TSEL_BASE) */
	uint32_t *lrow;                                              /* LDSROW: this lane's column of the wavefront's [3][C][64] block; selectors only: [C][64] */
	enum { TCROW = LDSROW ? 2 : 0 };
	K2A_FN k2a_pk rmax(int c) const { return LDSROW ? lrow[(0 * C + c) * 64] : rmax_[(NOMAX || LDSROW) ? 0 : c]; }
	K2A_FN k2a_pk rmj(int c) const { return DEFER ? 0u : LDSROW ? lrow[(1 * C + c) * 64] : rmj_[(NOMAX || LDSROW) ? 0 : c]; }
	K2A_FN k2a_pk tc(int c) const { return LDSTC ? lrow[(TCROW * C + c) * 64] : tc_[LDSTC ? 0 : c]; }
	K2A_FN void set_rmax(int c, k2a_pk v) { if (LDSROW) lrow[(0 * C + c) * 64] = v; else rmax_[(NOMAX || LDSROW) ? 0 : c] = v; }
	K2A_FN void set_rmj(int c, k2a_pk v) { if (DEFER) return; if (LDSROW) lrow[(1 * C + c) * 64] = v; else rmj_[(NOMAX || LDSROW) ? 0 : c] = v; }
	K2A_FN void set_tc(int c, k2a_pk v) { if (LDSTC) lrow[(TCROW * C + c) * 64] = v; else tc_[LDSTC ? 0 : c] = v; }

	K2A_FN static int first_col(int S_, int w_) { return k2a_max(0, S_ * C - w_); }

	K2A_FN void schedule_next()
	{
		koff_next = Snext;
		knext = Snext < nstrips ? koff_next + first_col(Snext, w) : K2A_KNONE;
	}

	K2A_FN void setup(const K2aPair &pr, const K2aPair &prB, const uint8_t *seq, int lane_in_group, bool valid, const uint32_t *cptab_)
	{
		cptab = cptab_; cpA = cpB = 0;
		qlen = pr.qlen; tlen = pr.tlen; tlen_full = pr.tlen_full; w = pr.w;
		qa = seq + pr.qoff; ta = seq + pr.toff; qbp = seq + prB.qoff; tbq = seq + prB.toff;
		nstrips = valid ? (tlen + C - 1) / C : 0;
		gl = lane_in_group;
		S = -1; i0 = 0; je = -1; koff = 0; kfin = K2A_KNONE; kd = 0; rows_m1 = -1; wup = w;
		Snext = gl;
		schedule_next();
		const k2a_pk neg = k2a_pku(K2A_NEG16);
		hout = eout = e2out = hd0 = hu_prev = neg; qb = 0; qwA = qwB = 0; seen = 0;
		hasn = 0; wn = false;
		baseA = baseB = 0; delta = 0;
		local_reset();
#pragma unroll
		for (int c = 0; c < C; ++c) { hl[c] = f[c] = neg; set_tc(c, K2A_TSEL_BASE); if (!NOMAX) { set_rmax(c, neg); set_rmj(c, 0); } if (DUAL) f2[c] = neg; }
		if (NOMAX || LDSROW) rmax_[0] = neg;
		if (NOMAX || LDSROW || DEFER) rmj_[0] = 0;
		if (LDSTC) tc_[0] = 0;
		if (!DUAL) f2[0] = 0;
	}

	K2A_FN int last_step() const { return nstrips > 0 ? (nstrips - 1) + k2a_min(qlen - 1, tlen - 1 + w) : -1; }
	K2A_FN bool need_init(int k) const { return k == knext; }
	K2A_FN bool need_fin(int k) const { return k == kfin; }

	/* bsA / bsB: bases of the lane that owns the strip above (RB only; rotated in by the kernel) */
	/* forced (k2a_argmax_kernel): the strip's bases and first diagonal input come from its checkpoint header instead of the lane above */
	K2A_FN void do_init(const K2aScoring &sc, int bsA = 0, int bsB = 0, const K2aCkHead *forced = 0)
	{
		S = Snext; i0 = S * C; koff = koff_next;
		je = k2a_min(qlen - 1, k2a_min(i0 + C - 1, tlen - 1) + w);
		kfin = koff + je;
		kd = koff + i0;
		rows_m1 = k2a_min(C - 1, tlen - 1 - i0);
		wup = w + (S == 0 ? 1 : 0);                        /* the virtual row -1 reaches one column further (E(0,w) exists) */
		const int js = k2a_max(0, i0 - w);
		const k2a_pk neg = k2a_pku(K2A_NEG16);
		/* target codes of the strip's rows, four rows per (unaligned) dword load and alignment; the arena is padded past the last
		 * row.  Not prefetched: one L2 round trip per strip is noise next to the strip's ~2w+C steps.  Per row one v_perm_b32 pairs
		 * alignment A's code with alignment B's and one add turns the pair into the row's selector -- this code runs with one lane
		 * per group active, every 17-19 steps: it was 11 % of the kernel when it went byte by byte. */
		const uint8_t *tpa = ta + (size_t)S * C, *tpb = tbq + (size_t)S * C;
		hasn = 0;
#pragma unroll
		for (int c4 = 0; c4 < C; c4 += 4) {
			uint32_t da, db;
			__builtin_memcpy(&da, tpa + c4, 4); __builtin_memcpy(&db, tpb + c4, 4);
			if (c4 + 4 > C) { da &= 0xffffu; db &= 0xffffu; }     /* (C = 18: the last two rows' dword reaches into the next strip, which reports them itself) */
			note_codes(da, db);
			if (TN && sc.pk_tn1 && ((da | db) & 0x04040404u)) {       /* a wildcard among these rows' target codes (rare; lane by lane) */
				hasn = 1;
#pragma unroll
				for (int r = 0; r < 4 && c4 + r < C; ++r) set_tc(c4 + r, k2a_tsel_wild(k2a_byte_pair(da, db, r)));
			} else {
#pragma unroll
				for (int r = 0; r < 4 && c4 + r < C; ++r) set_tc(c4 + r, k2a_byte_pair(da, db, r) + K2A_TSEL_BASE);
			}
		}
#pragma unroll
		for (int c = 0; c < C; ++c) {
			hl[c] = neg; f[c] = neg; if (DUAL) f2[c] = neg;
			if (!NOMAX) set_rmax(c, neg);                        /* the arg-max column is written with the row's first live cell */
		}
		const int hcorner = k2a_border<DUAL>(sc, i0) + sc.e * (i0 - 1);   /* H(i0-1,-1), carrying the bias of row i0-1 */
		if (forced) { baseA = forced->baseA; baseB = forced->baseB; delta = 0; }
		else if (RB) {
			/* new base = the diagonal input of the strip's first cell; hu_prev is still relative to the base above */
			const int nbA = js == 0 ? hcorner : bsA + k2a_pk_lo(k2a_ofs_off(hu_prev)), nbB = js == 0 ? hcorner : bsB + k2a_pk_hi(k2a_ofs_off(hu_prev));
			delta = S == 0 ? 0u : k2a_pair16((uint32_t)(bsA - nbA) & 0xffffu, (uint32_t)(bsB - nbB) & 0xffffu);
			baseA = nbA; baseB = nbB;
		}
		if (i0 <= w) {                                          /* some rows start at column 0: virtual column -1 */
#pragma unroll
			for (int c = 0; c < C; ++c) {                        /* ksw2_extz.c:43-44, ksw2_extd.c:49-52; row bias e*i */
				const int hb = k2a_border<DUAL>(sc, i0 + c + 1) + sc.e * (i0 + c);
				if (i0 + c <= w) {
					hl[c] = k2a_ofs_on(k2a_pair16((uint32_t)(hb - baseA) & 0xffffu, (uint32_t)(hb - baseB) & 0xffffu));
					f[c] = k2a_pk_sub(hl[c], k2a_pk2(sc.q + sc.e));
					if (DUAL) f2[c] = k2a_pk_sub(hl[c], k2a_pk2(sc.q2 + sc.e2));
				}
			}
		}
		if (forced) hd0 = forced->hd0;
		else if (RB) hd0 = K2A_OFS;                             /* 0 by construction of the base */
		else if (js == 0) hd0 = k2a_pku(hcorner);
		else hd0 = hu_prev;
		Snext += G;
		schedule_next();
	}

	/* first strip only: the cells above row 0 are the virtual row -1 (ksw2_extz.c:32-35, ksw2_extd.c:33-41) */
	K2A_FN void top_inputs(const K2aScoring &sc, int k, k2a_pk &hin, k2a_pk &ein, k2a_pk &e2in) const
	{
		if (S == 0) {
			const int hb = k2a_border<DUAL>(sc, k - koff + 1);
			const k2a_pk h0 = k2a_ofs_on(k2a_pair16((uint32_t)(hb - baseA) & 0xffffu, (uint32_t)(hb - baseB) & 0xffffu));
			hin = k2a_pk_sub(h0, k2a_pk2(sc.e));               /* row -1 carries bias -e, E(0,.) and E~(0,.) bias 0 */
			ein = k2a_pk_sub(h0, k2a_pk2(sc.q + sc.e)); e2in = k2a_pk_sub(h0, k2a_pk2(sc.q2 + sc.e2));
		}
	}

	/* One column for the C rows of both alignments.  hin/ein/e2in = bottom row of the strip above at this column.
	 * Phase 1 forms every row's diagonal candidate H(i-1,j-1) + s(i,j) while the old H row is still intact, phase 2 runs
	 * the E chain down the rows and writes the new H row in place (no register shuffling at the loop back-edge).
	 * With MODE != SCORE every comparison of the reference's direction logic (ksw2_extz.c:72-86 / 98-112, ksw2_extd.c:88-114 /
	 * 126-152) becomes sign(difference) in packed arithmetic; tbw receives C/2 words of direction bytes.
	 * Returns true when the lane computed live cells. */
	K2A_FN bool step(const K2aScoring &sc, int k, k2a_pk hin, k2a_pk ein, k2a_pk e2in, uint32_t *tbw)
	{
		const int dd = k - kd;                                 /* jj - i0 */
		const k2a_pk neg = k2a_pku(K2A_NEG16);
		const k2a_pk gq = k2a_pk2(sc.q), ge = k2a_pk2(sc.e), gq2 = k2a_pk2(sc.q2), ge2 = k2a_pk2(sc.e2), de2 = k2a_pk2(sc.e2 - sc.e);
		const k2a_pk bias = k2a_pk2(sc.pk_smax + sc.e);        /* largest score + row-bias step; the rows subtract their penalties from it */
		k2a_pk e = ein, e2 = e2in;
		if (dd >= wup) { e = neg; e2 = neg; }                  /* the cell above is outside the band */
		/* live rows lo..hi of this strip at this column (none while the lane owns no strip: rows_m1 = -1) */
		const int lo = k2a_max(0, dd - w);
		const int hi = k2a_min(rows_m1, dd + w);
		const int cnt = k2a_max(hi - lo + 1, 0);
		const uint32_t live = ((1u << cnt) - 1u) << (lo & 31);       /* lo >= 32 only with cnt = 0 */
		const k2a_pk jjpk = k2a_pk2(k - koff);
		/* rows in chunks of CH: phase 1 of a chunk (its diagonal candidates, from the old H row) right before its phase 2, so only
		 * CH candidates are alive at a time instead of C (16-18 registers: what keeps these kernels a wavefront short).  The one
		 * old H a chunk needs from the chunk above -- the row just above its first row -- is saved before that row is rewritten. */
		constexpr int CH = (C % 6 == 0) ? 6 : (C % 4 == 0) ? 4 : C;
		k2a_pk dprev = 0, uprev = 0, above_old = hd0;
#pragma unroll
		for (int c0 = 0; c0 < C; c0 += CH) {
			k2a_pk cand[CH];
			const k2a_pk last_old = hl[c0 + CH - 1];
#pragma unroll
			for (int r = 0; r < CH; ++r) {
				const int c = c0 + r;
				const k2a_pk up = r == 0 ? above_old : hl[c - 1];
				/* H(i-1,j-1) + s(i,j) + e: the row's penalty bytes { smax - s(tA, qA), 0, smax - s(tB, qB), 0 } out of the column profiles */
				cand[r] = k2a_sub32(k2a_add32(up, bias), k2a_perm(cpB, cpA, tc(c)));
			}
			if (TN && wn) {                                      /* a target wildcard row somewhere in the wavefront (scalar test): its constant penalty */
#pragma unroll
				for (int r = 0; r < CH; ++r) cand[r] = k2a_tn_fix(cand[r], tc(c0 + r), sc.pk_tn1);
			}
			above_old = last_old;
			if (CH < C) K2A_SCHED_FENCE();
#pragma unroll
			for (int r = 0; r < CH; ++r) {
				const int c = c0 + r;
				const k2a_pk fc = f[c];
				k2a_pk h = cand[r], s1 = 0, s2 = 0, s3 = 0, s4 = 0, x1 = 0, x2 = 0, x3 = 0, x4 = 0;      /* the differences whose signs are the direction flags (k2a_dir_flags) */
				if (MODE == K2A_MODE_SCORE) {
					h = k2a_pk_max3u(h, e, fc);                    /* one v_pk_maximum3_f16: see "Number format" above */
					if (DUAL) h = k2a_pk_max3u(h, e2, f2[c]);
				} else if (MODE == K2A_MODE_LEFT) {            /* winner changes only on a strictly larger gap state: negative = the gap state wins */
					s1 = k2a_pk_sub(h, e);  h = k2a_pk_maxu(h, e);
					s2 = k2a_pk_sub(h, fc); h = k2a_pk_maxu(h, fc);
					if (DUAL) { s3 = k2a_pk_sub(h, e2); h = k2a_pk_maxu(h, e2); s4 = k2a_pk_sub(h, f2[c]); h = k2a_pk_maxu(h, f2[c]); }
				} else {                                       /* right-aligned: a tie already moves to the gap state: negative = it does NOT win */
					s1 = k2a_pk_sub(e, h);  h = k2a_pk_maxu(h, e);
					s2 = k2a_pk_sub(fc, h); h = k2a_pk_maxu(h, fc);
					if (DUAL) { s3 = k2a_pk_sub(e2, h); h = k2a_pk_maxu(h, e2); s4 = k2a_pk_sub(f2[c], h); h = k2a_pk_maxu(h, f2[c]); }
				}
				h = k2a_pk_sel(k2a_bit_mask(live, c), h, neg);
				/* running row maximum: ties to the last column (keep the old arg-max only where h < max), except
				 * extz + RIGHT + CIGAR where the first column wins (take the new one only where max < h); SURVEY 8a rule 3 */
				if (!NOMAX && DEFER) set_rmax(c, k2a_pk_maxu(rmax(c), h));       /* the column comes from k2a_argmax_kernel */
				else if (!NOMAX) {
					const k2a_pk rm = rmax(c), rj = rmj(c);
					if (!DUAL && MODE == K2A_MODE_RIGHT) set_rmj(c, k2a_pk_selv(k2a_pk_sign(k2a_pk_sub(rm, h)), jjpk, rj));
					else set_rmj(c, k2a_pk_selv(k2a_pk_sign(k2a_pk_sub(h, rm)), rj, jjpk));
					set_rmax(c, k2a_pk_maxu(rm, h));
				}
				/* gaps leaving the cell, all in row-biased form: opening = H' - q; the extension cost cancels against the bias
				 * for E (next row), stays e for F (same row), becomes e2 - e for E~ and stays e2 for F~.  "extension beats
				 * opening" (ksw2_extz.c:79-86 / 105-112) compares the gap state with the opening value directly. */
				const k2a_pk t = k2a_sub32(h, gq);
				if (MODE == K2A_MODE_LEFT) { x1 = k2a_pk_sub(t, e); x2 = k2a_pk_sub(t, fc); }              /* extension strictly better than opening */
				else if (MODE == K2A_MODE_RIGHT) { x1 = k2a_pk_sub(e, t); x2 = k2a_pk_sub(fc, t); }        /* extension at least as good as opening */
				e = k2a_pk_maxu(e, t);
				f[c] = k2a_sub32(k2a_pk_maxu(fc, t), ge);
				if (DUAL) {
					const k2a_pk t2 = k2a_sub32(h, gq2);
					if (MODE == K2A_MODE_LEFT) { x3 = k2a_pk_sub(t2, e2); x4 = k2a_pk_sub(t2, f2[c]); }
					else if (MODE == K2A_MODE_RIGHT) { x3 = k2a_pk_sub(e2, t2); x4 = k2a_pk_sub(f2[c], t2); }
					e2 = k2a_pk_sub(k2a_pk_maxu(e2, t2), de2);           /* e2 - e may be negative: packed subtract */
					f2[c] = k2a_sub32(k2a_pk_maxu(f2[c], t2), ge2);
				}
				if (MODE != K2A_MODE_SCORE) {
					const uint32_t fl = k2a_dir_flags<DUAL, NIB != 0, MODE == K2A_MODE_RIGHT>(s1, s2, s3, s4, x1, x2, x3, x4);      /* byte 0 = alignment A, byte 1 = B */
					if (!(c & 1)) dprev = fl;
					else if (NIB) {
						const uint32_t u = k2a_pk_selv(0x0f0f0f0fu, dprev, fl << 4);      /* rows c - 1 | c: two four-bit codes per byte */
						if ((c & 3) == 3) tbw[c >> 2] = k2a_perm(u, uprev, 0x05010400u);  /* { A rows 4g..4g+3 from bit 0 up, B likewise } */
						else uprev = u;
					} else tbw[c >> 1] = k2a_perm(fl, dprev, 0x05040100u);               /* bytes {A(c-1), B(c-1), A(c), B(c)} */
				}
				hl[c] = h;
			}
			if (CH < C) K2A_SCHED_FENCE();
		}
		hd0 = hin;
		hout = hl[C - 1]; eout = e; e2out = e2;
		return live != 0;
	}

	/* query codes of the column this lane sees at step k+1; idle lanes read a clamped (valid, unused) column */
	K2A_FN uint32_t next_query_codes(int k) const
	{
		const int j = k + 1 - ((k + 1 == knext) ? koff_next : koff);
		const int jc = k2a_min(k2a_max(j, 0), qlen - 1);
		return k2a_pair16(qa[jc], qbp[jc]);
	}

	/* The same, four steps at a time (k2a_fill_pk_kernel): qwA / qwB hold the codes of this lane's columns at the steps
	 * kg .. kg + 3 of the current group (kg = k & ~3), one byte per step, fetched with ONE unaligned dword load per alignment
	 * and group instead of a byte load, an address and a clamp per step; query_pick is one v_perm_b32 with a scalar selector.
	 * The kernel asks for a group at the top of the group before it (k2a_load_early: four steps in flight) and takes it over
	 * below that group's last step.  A strip that starts inside a group re-loads the group under its own column offset
	 * (reload_query_group from the init branch, which does wait).  Bytes of columns outside the query are garbage that only dead cells see (the arena is padded). */
	uint32_t qwA, qwB;
	/* OR of every TARGET code dword this lane fetched (k2a_fill_pk_kernel).  A row's profile has four bytes, target codes 0..3, so
	 * the host keeps pairs whose target holds a wildcard code (>= 4) out of the packed kernels -- by scanning the sequences while it
	 * copies them.  A flat batch (ksw2amd_plan_create_flat) is uploaded as it lies in the caller's arena, unscanned: the kernel
	 * then reports "a code >= 4 was among the target bytes I read" (K2aResult.pad[0]) and the host re-runs that pair through the
	 * int32 kernels.  Bytes that belong to a neighbouring sequence can only cause a needless re-run.  (Query codes are table
	 * indices, 0..4 by the caller's contract `codes < m`; nothing to report.) */
	uint32_t seen;
	K2A_FN void load_query_group(int kg, int koff_use, uint32_t &a, uint32_t &b)            /* the wait comes with the first use of a / b */
	{
		const int jc = k2a_min(k2a_max(kg - koff_use, 0), qlen - 1);    /* column at step kg; a negative one belongs to a lane without a strip */
		a = k2a_load_early(qa + jc); b = k2a_load_early(qbp + jc);
	}
	K2A_FN void note_codes(uint32_t a, uint32_t b) { seen |= a | b; }
	K2A_FN bool saw_wildcard() const { return (seen & 0xfcfcfcfcu) != 0; }      /* (with K2aScoring.pk_tn1 the kernels' own look at the targets decides: k2a_scan_codes) */
	K2A_FN void reload_query_group(int k)                      /* from the init branch: this strip started at step k, inside a group */
	{
		const int j = (k & ~3) - koff;                             /* < 0: the strip's column 0 comes -j steps into the group */
		const int sh = 8 * k2a_min(k2a_max(-j, 0), 3);
		uint32_t va, vb;
		load_query_group(k & ~3, koff, va, vb);
		qwA = va << sh; qwB = vb << sh;
	}
	K2A_FN static uint32_t query_pick(uint32_t a, uint32_t b, int kk)      /* { code A, code B } of step kg + kk */
	{
#if defined(__HIP_DEVICE_COMPILE__)
		return __builtin_amdgcn_perm(b, a, 0x0c040c00u + (uint32_t)kk * 0x00010001u);      /* kk is wave-uniform: a scalar selector */
#else
		return k2a_byte_pair(a, b, kk);
#endif
	}

	/* Strip epilogues.  Both forms first stage the strip's rows {H(i, last column), row max, arg-max} in an LDS row
	 * buffer (K2A_PK_STAGE(C) words per lane group) and then walk them in a ROLLED loop: unrolled, hipcc materialises every row's
	 * constants and unpacked halves at once and the kernel loses a wave of occupancy for code that runs once per strip. */
	K2A_FN void stage_rows(uint32_t *rowbuf) const
	{
#pragma unroll
		for (int c = 0; c < C; ++c) { rowbuf[c] = k2a_ofs_off(hl[c]); rowbuf[C + c] = k2a_ofs_off(rmax(c)); rowbuf[2 * C + c] = rmj(c); }   /* plain int16 halves */
		rowbuf[3 * C] = (uint32_t)i0;
		if (RB) { rowbuf[3 * C + 1] = (uint32_t)baseA; rowbuf[3 * C + 2] = (uint32_t)baseB; }
	}

	/* Sequential form (needed as soon as a Z-drop test is active): the scalar reference's per-row epilogue
	 * (K2aLane::do_fin), once per alignment. */
	K2A_FN void do_fin_seq(const K2aScoring &sc, K2aBook *bA, K2aBook *bB, int zdropA, int zdropB, const uint32_t *rowbuf)
	{
		const int zslope = DUAL ? sc.e2 : sc.e;
#pragma nounroll
		for (int half = 0; half < 2; ++half) {
			K2aBook *b = half ? bB : bA;
			const int zdrop = half ? zdropB : zdropA;
			const int sh = half ? 16 : 0;
			int bmax = b->max, bmax_t = b->max_t, bmax_q = b->max_q, bmqe = b->mqe, bmqe_t = b->mqe_t;
			int bmte = b->mte, bmte_q = b->mte_q, bscore = b->score, bdrop = b->dropped, brows = b->rows;
#pragma nounroll
			for (int c = 0; c < C; ++c) {
				const int i = i0 + c;
				if (i < tlen && !bdrop) {
					const bool reach = i + w >= qlen - 1;
					const int unb = (half ? baseB : baseA) - sc.e * i;                                       /* plus base, minus row bias */
					const int hend = (int)(int16_t)(rowbuf[c] >> sh) + unb, H = (int)(int16_t)(rowbuf[C + c] >> sh) + unb;
					const int j = (int)(uint16_t)(rowbuf[2 * C + c] >> sh);             /* column index: unsigned, reads up to 65 000 */
					if (reach && hend > bmqe) { bmqe = hend; bmqe_t = i; }
					if (i == tlen_full - 1) { bmte = H; bmte_q = j; }
					if (H > bmax) { bmax = H; bmax_t = i; bmax_q = j; }
					else if (DEFER) {
						/* without the row's arg-max column neither "j >= max_q" nor the skew can be evaluated: a drop is impossible
						 * while max - H <= zdrop; otherwise the book is FROZEN here (dropped = 1, inexact = 1, rows = this row + 1) and
						 * the third pass (k2a_zscan_kernel) takes the rows from this one on, with their columns */
						if (zdrop >= 0 && bmax - H > zdrop) { bdrop = 1; b->inexact = 1; }
					} else if (i >= bmax_t && j >= bmax_q) {
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
		}
		end_strip();
	}

	/* Shortcut in front of the sequential form, in registers and for both alignments at once.  For a full strip that
	 * neither holds the last target row nor reaches the query end, all the sequential scan can do is (a) move the
	 * running maximum to the first row of the strip that beats it and (b) Z-drop.  With M = max(book max, strip max) and
	 * m = the smallest row maximum of the strip, no row can drop when M - m <= zdrop (its own test sees at most M - m on the
	 * left and at least zdrop on the right), and then (a) is a plain first-occurrence arg-max over the C rows.
	 * Returns false -- having written nothing -- whenever any of this does not hold; the caller then runs do_fin_seq. */
	K2A_FN bool fin_fast(const K2aScoring &sc, K2aBook *bA, K2aBook *bB, int zdropA, int zdropB)
	{
		if (i0 + C >= tlen || i0 + C - 1 + w >= qlen - 1) return false;
		/* rows compare without their bias: v_c = rmax[c] - e*c = H(row) + (e*i0 - base) */
		k2a_pk m = k2a_ofs_off(rmax(0)), mn = m, arg = 0, argj = rmj(0);
#pragma unroll
		for (int c = 1; c < (NOMAX ? 1 : C); ++c) {
			const k2a_pk v = k2a_pk_sub(rmax(c), k2a_pk2(sc.e * c + K2A_OFS16));
			const k2a_pk gt = k2a_pk_sign(k2a_pk_sub(m, v));                        /* strictly larger: first row keeps a tie */
			arg = k2a_pk_sel(gt, k2a_pk2(c), arg);
			argj = k2a_pk_sel(gt, rmj(c), argj);
			m = k2a_pk_max(m, v);
			mn = k2a_pk_min(mn, v);
		}
		const int offA = (RB ? baseA : 0) - sc.e * i0, offB = (RB ? baseB : 0) - sc.e * i0;
		const int MA = k2a_pk_lo(m) + offA, MB = k2a_pk_hi(m) + offB, mA = k2a_pk_lo(mn) + offA, mB = k2a_pk_hi(mn) + offB;
		const int bmA = bA->max, bmB = bB->max;
		const bool deadA = bA->dropped != 0, deadB = bB->dropped != 0;
		if (!deadA && zdropA >= 0 && k2a_max(bmA, MA) - mA > zdropA) return false;
		if (!deadB && zdropB >= 0 && k2a_max(bmB, MB) - mB > zdropB) return false;
		if (!deadA) {
			if (MA > bmA) { bA->max = MA; bA->max_t = i0 + k2a_pk_lo(arg); bA->max_q = (int)(argj & 0xffffu); }
			bA->rows = i0 + C;
		}
		if (!deadB) {
			if (MB > bmB) { bB->max = MB; bB->max_t = i0 + k2a_pk_hi(arg); bB->max_q = (int)(argj >> 16); }
			bB->rows = i0 + C;
		}
		end_strip();
		return true;
	}

	/* Local form (no Z-drop test anywhere in the wavefront, so nothing can stop early): every lane keeps its own best
	 * (max, row, column) and best end-of-query (mqe, row) for both alignments in packed registers, and ALL lanes of the
	 * group share the rows of the strip that just ended (row c goes to lane c mod G), so an epilogue costs ceil(C / G)
	 * row updates of latency instead of C.  "H > max" in row order == first row reaching the maximum; every lane sees
	 * its rows in increasing order (inside a strip and from strip to strip), and the final merge across lanes
	 * (k2a_merge_local) breaks ties towards the smaller row, so the result is the sequential one. */
	k2a_pk lmax, lmax_t, lmax_q, lmqe, lmqe_t;
	k2a_pk last_h, last_m, last_j;         /* last target row: H(tlen-1, last column), row max, arg-max */

	K2A_FN void local_reset()
	{
		lmax = 0; lmax_t = lmax_q = 0xffffffffu; lmqe = k2a_pk2(K2A_NEG16); lmqe_t = 0xffffffffu;
		last_h = last_m = last_j = 0;
	}

	K2A_FN void fin_local_rows(const K2aScoring &sc, const uint32_t *rowbuf)
	{
		const int fi0 = (int)rowbuf[3 * C];                                    /* first row of the strip that ended */
#pragma nounroll
		for (int c = gl; c < C; c += G) {
			const int i = fi0 + c;
			const k2a_pk bias = k2a_pk2(sc.e * i);
			const k2a_pk pj = rowbuf[2 * C + c], ipk = k2a_pk2(i);
			/* un-bias; rows past the target end hold -inf, which must stay below every real score */
			const k2a_pk ph = (i < tlen) ? k2a_pk_sub(rowbuf[c], bias) : k2a_pk2(K2A_NEG16);
			const k2a_pk pm = (i < tlen) ? k2a_pk_sub(rowbuf[C + c], bias) : k2a_pk2(K2A_NEG16);
			/* rows past the target end kept rmax = hl = -inf and can never win */
			const k2a_pk up = k2a_pk_sign(k2a_pk_sub(lmax, pm));                /* max < H */
			lmax_t = k2a_pk_sel(up, ipk, lmax_t);
			lmax_q = k2a_pk_sel(up, pj, lmax_q);
			lmax = k2a_pk_max(lmax, pm);
			const k2a_pk reach = (i + w >= qlen - 1) ? 0xffffffffu : 0u;        /* the row's last cell is column qlen-1 */
			const k2a_pk uq = k2a_pk_sign(k2a_pk_sub(lmqe, ph)) & reach;
			lmqe_t = k2a_pk_sel(uq, ipk, lmqe_t);
			lmqe = k2a_pk_sel(uq, ph, lmqe);
			if (i == tlen_full - 1 && tlen == tlen_full) { last_h = ph; last_m = pm; last_j = pj; }   /* mte / mte_q / score */
		}
	}
	K2A_FN void end_strip() { S = -1; je = -1; kfin = K2A_KNONE; rows_m1 = -1; hasn = 0; }

	/* NOMAX epilogue: the only thing a finished strip contributes is H(tlen-1, qlen-1), if it holds the last target row and
	 * that row reaches the last column (ksw2_extz2_sse.c:284-285) */
	K2A_FN void fin_score_only(const K2aScoring &sc, K2aBook *bA, K2aBook *bB)
	{
		const int last = tlen_full - 1;
		if (tlen == tlen_full && last >= i0 && last < i0 + C && last + w >= qlen - 1) {
			k2a_pk v = 0;
#pragma unroll
			for (int c = 0; c < C; ++c) if (i0 + c == last) v = k2a_ofs_off(hl[c]);
			bA->score = k2a_pk_lo(v) + (RB ? baseA : 0) - sc.e * last;
			bB->score = k2a_pk_hi(v) + (RB ? baseB : 0) - sc.e * last;
		}
		end_strip();
	}
};

/* The rows of ONE finished strip folded into the book of ONE alignment (half = 0 / 1) of a single-gap task: the exact branch of
 * K2aLanePk::do_fin_seq -- the scalar reference's per-row epilogue (ksw2_extz.c:116-124, ksw2.h:191-207) -- on rows another lane
 * staged (stage_rows).  k2a_zscan_kernel folds the strips it re-ran with it, in row order. */
template<int C>
K2A_FN void k2a_fin_rows_half(const K2aScoring &sc, K2aBook *b, int zdrop, const uint32_t *rowbuf, int half, bool rb, int qlen, int tlen, int tlen_full, int w)
{
	const int i0 = (int)rowbuf[3 * C], base = rb ? (int)rowbuf[3 * C + 1 + half] : 0, sh = half ? 16 : 0;
	int bmax = b->max, bmax_t = b->max_t, bmax_q = b->max_q, bmqe = b->mqe, bmqe_t = b->mqe_t;
	int bmte = b->mte, bmte_q = b->mte_q, bscore = b->score, bdrop = b->dropped, brows = b->rows;
#pragma nounroll
	for (int c = 0; c < C; ++c) {
		const int i = i0 + c;
		if (i < tlen && !bdrop) {
			const bool reach = i + w >= qlen - 1;
			const int unb = base - sc.e * i;                                                        /* plus base, minus row bias */
			const int hend = (int)(int16_t)(rowbuf[c] >> sh) + unb, H = (int)(int16_t)(rowbuf[C + c] >> sh) + unb;
			const int j = (int)(uint16_t)(rowbuf[2 * C + c] >> sh);
			if (reach && hend > bmqe) { bmqe = hend; bmqe_t = i; }
			if (i == tlen_full - 1) { bmte = H; bmte_q = j; }
			if (H > bmax) { bmax = H; bmax_t = i; bmax_q = j; }
			else if (i >= bmax_t && j >= bmax_q) {
				const int dt = i - bmax_t, dq = j - bmax_q;
				const int skew = dt > dq ? dt - dq : dq - dt;
				if (zdrop >= 0 && bmax - H > zdrop + skew * sc.e) bdrop = 1;
			}
			if (!bdrop && i == tlen_full - 1 && reach) bscore = hend;
			brows = i + 1;
		}
	}
	b->max = bmax; b->max_t = bmax_t; b->max_q = bmax_q; b->mqe = bmqe; b->mqe_t = bmqe_t;
	b->mte = bmte; b->mte_q = bmte_q; b->score = bscore; b->dropped = bdrop; b->rows = brows;
}

/* Traceback walk for one alignment (half = 0/1) of a packed task: direction bytes in the reference layout at byte
 * 2c + half of the (step, lane) word (K2aWalk layout 2). */
template<int G, int C, bool DUAL = true, bool MP = false>
K2A_FN int k2a_trace_pair_pk(const uint8_t *tb, int half, int i, int j, uint32_t *out, int qlen, int tlen, int w, uint8_t *win)
{
	return k2a_trace_walk<G, C, K2A_PK_NIBBLES(C, DUAL) ? 3 : 2, MP>(tb, half, i, j, out, qlen, tlen, w, win);
}

/* merge the lane-local bests of one alignment (half = 0/1) of a lane group: loc[l*5 + {0..4}] = lane l's
 * {lmax, lmax_t, lmax_q, lmqe, lmqe_t}.  Ties go to the smaller row, as in the sequential scan. */
K2A_FN void k2a_merge_local(const uint32_t *loc, int G, int half, K2aBook *b)
{
	int bmax = 0, bmax_t = -1, bmax_q = -1, bmqe = K2A_NEG, bmqe_t = -1;
	for (int l = 0; l < G; ++l) {
		const uint32_t *r = loc + l * 5;
		const int m = half ? k2a_pk_hi(r[0]) : k2a_pk_lo(r[0]);
		const int mt = half ? k2a_pk_hi(r[1]) : k2a_pk_lo(r[1]);
		const int mq = half ? k2a_pk_hi(r[2]) : k2a_pk_lo(r[2]);
		const int qe = half ? k2a_pk_hi(r[3]) : k2a_pk_lo(r[3]);
		const int qt = half ? k2a_pk_hi(r[4]) : k2a_pk_lo(r[4]);
		if (mt >= 0 && (m > bmax || (m == bmax && bmax_t >= 0 && mt < bmax_t))) { bmax = m; bmax_t = mt; bmax_q = mq; }
		if (qt >= 0 && (qe > bmqe || (qe == bmqe && qt < bmqe_t))) { bmqe = qe; bmqe_t = qt; }
	}
	b->max = bmax; b->max_t = bmax_t; b->max_q = bmax_q; b->mqe = bmqe; b->mqe_t = bmqe_t;
}

#endif
