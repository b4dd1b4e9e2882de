/* lane_ops_probe.h -- the packed kernels' register primitives (ksw2_amd/csrc/ksw2_lane_pk.h) behind one switch, so that the SAME
 * source line is compiled twice: by hipcc for gfx950 (the inline-asm / builtin forms) and by g++ for the host (the C twins the
 * lock-step simulator tests/sim runs).  tools/probe/lane_ops_probe.hip runs both on the same inputs and compares bit for bit:
 * an instruction that drifted from its twin (a v_perm_b32 selector, a v_bitop3_b32 truth table, v_pk_mad_i16's wrap-around)
 * would pass every CPU test and fail only on the GPU tier -- this names the instruction. */
#ifndef LANE_OPS_PROBE_H_
#define LANE_OPS_PROBE_H_
#include <cstddef>
#include <cstdint>
#include "../../ksw2_amd/csrc/ksw2_lane_pk.h"

enum { OP_PERM, OP_BYTE_PAIR, OP_MAX3U, OP_MAD, OP_SIGN, OP_SELV, OP_PACK_DIRS, OP_BIT_MASK, OP_SHL, OP_ADD, OP_SUB, OP_MAX, OP_MIN, OP_MAXU, OP_SEL,
       OP_QUERY_PICK, OP_COUNT };
static const char *const lane_op_name[OP_COUNT] = { "k2a_perm (v_perm_b32, register selector)", "k2a_byte_pair (v_perm_b32, constant selectors)",
	"k2a_pk_max3u (v_pk_maximum3_f16 on offset-form patterns)", "k2a_pk_mad (v_pk_mad_i16)", "k2a_pk_sign (v_pk_ashrrev_i16 15)",
	"k2a_pk_selv (v_bitop3_b32 0xe4)", "k2a_dir_flags (v_perm_b32 sign selectors + v_bitop3_b32 merges)", "k2a_bit_mask (v_bfe_i32)", "k2a_pk_shl (v_pk_lshlrev_b16)",
	"k2a_pk_add", "k2a_pk_sub", "k2a_pk_max", "k2a_pk_min", "k2a_pk_maxu", "k2a_pk_sel", "query_pick (v_perm_b32, scalar selector)" };

K2A_FN uint32_t lane_op(int op, uint32_t a, uint32_t b, uint32_t c)
{
	switch (op) {
	case OP_PERM: return k2a_perm(a, b, c);
	case OP_BYTE_PAIR: return (c & 3) == 0 ? k2a_byte_pair(a, b, 0) : (c & 3) == 1 ? k2a_byte_pair(a, b, 1) : (c & 3) == 2 ? k2a_byte_pair(a, b, 2) : k2a_byte_pair(a, b, 3);
	case OP_MAX3U: return k2a_pk_max3u(a, b, c);
	case OP_MAD: return k2a_pk_mad(a, b, c);
	case OP_SIGN: return k2a_pk_sign(a);
	case OP_SELV: return k2a_pk_selv(a, b, c);
	case OP_PACK_DIRS: return (c & 1) ? k2a_dir_flags<true, false, false>(a, b, c, a ^ c, b ^ c, ~a, ~b, a + b) ^ k2a_dir_flags<false, true, true>(a, b, 0, 0, c, ~c, 0, 0)
	                                  : k2a_dir_flags<false, false, false>(a, b, 0, 0, c, a - c, 0, 0);
	case OP_BIT_MASK: return k2a_bit_mask(a, (int)(c & 31));
	case OP_SHL: return (c & 1) ? k2a_pk_shl(a, 4) : k2a_pk_shl(a, 1);
	case OP_ADD: return k2a_pk_add(a, b);
	case OP_SUB: return k2a_pk_sub(a, b);
	case OP_MAX: return k2a_pk_max(a, b);
	case OP_MIN: return k2a_pk_min(a, b);
	case OP_MAXU: return k2a_pk_maxu(a, b);
	case OP_SEL: return k2a_pk_sel(a, b, c);
	case OP_QUERY_PICK: return K2aLanePk<64, 16, false>::query_pick(a, b, (int)(c & 3));
	}
	return 0;
}
#endif
