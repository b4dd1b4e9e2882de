/*
 * ksw2_types.h -- plain-C data layout shared by the C host (ksw2_host_*.c), the device shim
 * (ksw2_shim_hip.hip) and the per-lane kernel code (ksw2_lane.h).
 */
#ifndef KSW2_TYPES_H_
#define KSW2_TYPES_H_

#include <stdint.h>

#define K2A_NEG      (-0x40000000)
#define K2A_KNONE    0x7fffffff
#define K2A_MAXM     127             /* residue types (int8_t m, ksw2.h:61) */

/* flag bits the device code looks at (numerically the KSW_EZ_* values, ksw2.h:8-18) */
#define K2A_F_SCORE_ONLY 0x01
#define K2A_F_EXTZ_ONLY 0x40
#define K2A_F_REV_CIGAR 0x80
#define K2A_F_SPLICE_FOR   0x100        /* splice-aware plans: K2aPair.flag carries the KSW_EZ_SPLICE_* bits for k2a_splice_const */
#define K2A_F_SPLICE_REV   0x200
#define K2A_F_SPLICE_FLANK 0x400
#define K2A_F_HAS_JUNC     0x10000000   /* ... and whether annotation bytes follow the target in the arena */

/* kernel MODE */
#define K2A_MODE_SCORE 0     /* no traceback                                      */
#define K2A_MODE_LEFT  1     /* traceback bits, gaps left-aligned  (default)       */
#define K2A_MODE_RIGHT 2     /* traceback bits, gaps right-aligned (KSW_EZ_RIGHT)  */

/* traceback blocks: every lane's run of lane-step words is padded to a multiple of K2A_TB_PAD steps, so that the fill
 * kernels can hand whole 128-byte lines to the memory system (staged through LDS, ksw2_shim_hip.hip) */
#define K2A_TB_PAD 32
#define K2A_TB_PADDED(steps) (((steps) + (K2A_TB_PAD - 1)) / K2A_TB_PAD * K2A_TB_PAD)

/* packed traceback words: single-gap kernels of the 16-row geometries write 4 bits per cell (2-bit winner + E / F continuation,
 * the int32 kernels' code), everything else one byte per cell in the reference's layout (ksw2.h:125-128) */
/* number format of the packed-int16 kernels (ksw2_lane_pk.h): a value v is held as the 16-bit pattern v + K2A_OFS16, and every
 * pattern the fill loops compare lies in 0x0000 .. 0x7BFF (the non-negative finite IEEE halves, so that v_pk_maximum3_f16 is an
 * integer maximum): -inf = K2A_NEG16 -> 0x0C00 with K2A_PK_SLACK units below it, largest value K2A_PK_VMAX -> 0x7BFF */
#define K2A_NEG16    (-16384)
#define K2A_OFS16    0x4C00
#define K2A_PK_VMAX  (0x7BFF - K2A_OFS16)       /* 12287 */
#define K2A_PK_SLACK (K2A_OFS16 + K2A_NEG16)    /* 3072: room below -inf for the base shifts and gap costs applied to it */
/* packed generation-serial class (ksw2_lane_pkmp.h): sliding base */
#ifndef K2A_PKMP_T
#define K2A_PKMP_T    64            /* steps between re-bases (a power of two) */
#endif
#define K2A_PKMP_DEAD (-8192)       /* relative values below this are -inf */
#define K2A_PKMP_RMAX_LIMIT 10000   /* a window's row maximum further above the base than this is merged into its key; it may drift
                                     * K2A_PKMP_T steps further before the next check and must still fit K2A_PK_VMAX */
#define K2A_PK_NIBBLES(C, dual) (!(dual) && (C) == 16)
#define K2A_PK_TB_BYTES(C, dual) (K2A_PK_NIBBLES(C, dual) ? (C) : 2 * (C))        /* per lane-step: C rows x 2 alignments */

/* batch-uniform scoring, passed by value to the kernel */
typedef struct K2aScoring {
	int32_t q, e, q2, e2;        /* gap open / extend; (q2,e2) only for the two-piece model, q+e <= q2+e2 */
	uint32_t prof[5];            /* prof[t] = bytes { s(t,0), s(t,1), s(t,2), s(t,3) } for target code t   */
	int32_t colw[5];             /* colw[t] = s(t, 4): score against the query wildcard (code 4)          */
	/* packed-int16 kernels (ksw2_lane_pk.h, "column profiles"): any matrix over codes 0..3 x 0..4 as penalties below its largest
	 * entry.  cp[q] = bytes { smax - s(0,q), smax - s(1,q), smax - s(2,q), smax - s(3,q) } for query code q (4 = the query's
	 * wildcard; 5..7 unused, zero); the diagonal candidate is H + (pk_smax + e) - byte t of cp[q]: one v_perm_b32 per row. */
	uint32_t cp[8];
	int32_t pk_smax;
	/* TARGET wildcard (code 4) in the packed kernels (round 6): where its scores do not depend on the query code -- s(4, q) = sN for q =
	 * 0..4, as in every match / mismatch / N matrix -- a row whose target code is 4 takes penalty 0 out of the profile (selector byte
	 * 0x0c) and has pk_tn1 - 1 = smax - sN taken off its candidate in a branch only wavefronts that hold such a row enter
	 * (K2aLanePk::step).  0: off -- such pairs leave the packed kernels (scanned plans) or are reported and re-run (unscanned ones). */
	int32_t pk_tn1;
	int32_t m;                   /* residue types; m > 5: scores come from `mat` (staged in LDS), prof/colw unused */
	const int8_t *mat;           /* device copy of the effective m x m matrix, mat[target*m + query]       */
} K2aScoring;

/* register window classes of the diagonal-major kernel: K slots of 64 target positions hold diagonals of up to
 * K*64 - 64 cells (the window starts one cell below the diagonal and is 64-aligned).  A 24-slot class (1 wavefront per
 * SIMD, spills) measured slower than the HBM-state kernel and was dropped. */
#define K2A_DM_SLOTS_S 8
#define K2A_DM_SLOTS   16
#define K2A_DM_DIAG(K) ((K) * 64 - 64)

/* batch-uniform parameters of the splice-aware extension (ksw2_lane_dm.h), passed by value */
typedef struct K2aSplice {
	int32_t q, e, q2, long_thres;
	int32_t m;
	const int8_t *mat;           /* device copy of the effective m x m matrix, mat[target*m + query] */
} K2aSplice;

/* batch-uniform parameters of the gap-linear X-drop extension (ksw2_lane_extf.h), passed by value */
typedef struct K2aExtf {
	int32_t mch, mis, e;         /* mis <= 0 (ksw2_extf2_sse.c:20) */
	int32_t ring;                /* one-extension-per-lane form: 0 = state arrays in HBM scratch, else rows (dwords per lane and array) of the LDS ring */
} K2aExtf;
/* rows of four positions a lane of the one-extension-per-lane form touches on one anti-diagonal, at most: from the dword of
 * position blo - 1 (the carry into the band, the followed cell) to the one of hi + 15 (the padded block, the S refresh), with
 * hi - lo <= min(w, qlen, tlen) and blo >= lo - 15 (ksw2_lane_extf.h) */
#define K2A_EXTF_RING_ROWS(span) (((span) + 30) / 4 + 3)

/* batch-uniform parameters of the SSE-compatible mode (ksw2_lane_ssec.h), passed by value */
typedef struct K2aSsec {
	int32_t q, e, q2, e2;            /* extd2: pieces already ordered q + e <= q2 + e2 (ksw2_extd2_sse.c:78) */
	int32_t qe_first;                /* extd2's scalar q + e from BEFORE that swap, used for the very first cell only (:67,:353,:377) */
	int32_t long_thres, long_diff;   /* ksw2_extd2_sse.c:102-105 */
	int32_t m, sc_mch, sc_mis, sc_N; /* residue types; match / mismatch / wildcard score of the simple scoring (:66-69 / :85-88) */
	const int8_t *mat;               /* KSW_EZ_GENERIC_SC: device copy of the caller's matrix, mat[target * m + query] */
} K2aSsec;
#define K2A_SSEC_APPROX      1       /* bits of K2aPair.pad in that mode: KSW_EZ_APPROX_MAX ... */
#define K2A_SSEC_APPROX_DROP 2       /* ... with KSW_EZ_APPROX_DROP */
#define K2A_SSEC_GENERIC     4       /* KSW_EZ_GENERIC_SC */

/* widest band (positions on one anti-diagonal) the K-slot register window of the X-drop kernel holds */
#define K2A_EXTF_WIN_SPAN(K) (64 * (K) - 109)

/* one alignment, device-resident */
typedef struct K2aPair {
	uint32_t qoff, toff;         /* byte offsets of query / target in the sequence arena                  */
	int32_t qlen, tlen;          /* tlen = rows that own at least one in-band cell (<= true target length) */
	int32_t tlen_full;           /* true target length (mte / score need the real last row)               */
	int32_t w;                   /* effective band, 1 <= w <= max(qlen, tlen)                             */
	int32_t zdrop, end_bonus, flag;
	uint32_t cig_off;            /* dword offset of this pair's CIGAR scratch                              */
	uint64_t tb_off;             /* byte offset of this pair's traceback block                            */
	uint32_t bnd_off;            /* int32 offset of this pair's generation boundary rows (3 x qlen ints)   */
	uint32_t pad;
} K2aPair;

/* what the reference keeps in ksw_extz_t (ksw2.h:33-42), one per alignment */
typedef struct K2aResult {
	int32_t max, zdropped, max_q, max_t, mqe, mqe_t, mte, mte_q, score, reach_end, n_cigar, rows_done;
	int32_t ti, tj;                  /* traceback start cell chosen by k2a_finish(); -1 = no CIGAR          */
	int32_t pad[2];                  /* [0]: a packed kernel read a wildcard code in a target (unscanned plans); [1]: deferred arg-max -- 0 = exact (or settled on the device by the third pass, which rewrites the record through k2a_finish), 1 = the fill froze the book at a row whose Z-drop test needs columns and nothing settled it: the host re-runs the pair (needs_rerun).  No other value is written */
} K2aResult;

/* Streamed launches (ksw2_host_plan.c, "streamed plans"): ONE launch of a packed fill kernel over the whole batch, started under the
 * batch's upload.  Its wavefronts take their wavefront-tasks (the 64 / G consecutive tasks one wavefront runs) by position in the
 * grid, longest first, and each waits in front of its task until the upload pieces its sequences lie in have landed: the host
 * uploads the sequence arena in pieces on another stream and behind each piece copies a block of K2A_WM_BYTES filled with the piece's
 * number onto the plan's watermark block (a DMA copy like the piece itself, ordered behind it by the stream: word 0 of the block =
 * pieces that have landed).  A wavefront whose inputs do not arrive within `timeout_ticks` (100 MHz) sets `abort` and the wavefronts
 * that start later leave at once; the host then runs the plan again, unstreamed.  Device-resident, one per launch, uploaded with the
 * task lists. */
#define K2A_WM_BYTES 65536            /* large enough that the runtime moves it with the DMA engines like the pieces themselves, never with a kernel
                                       * (a launch that fills the device leaves a copy kernel no wavefront slot: tools/probe/stream_publish_probe.hip) */
/* the 2-bit wire format's escape entries (ksw2_lane.h: k2a_wire2_expand; ksw2_host_pool.c: pack2_esc) */
#define K2A_WIRE2_ESC 7                                   /* entries per pair */
#define K2A_WIRE2_SLOT (4 * K2A_WIRE2_ESC)                /* upload bytes: the last ones of the pair's region */
#define K2A_WIRE2_PAD (4 * K2A_WIRE2_SLOT)                /* arena bytes of extra target padding that make room for them */

typedef struct K2aQueueDesc {
	uint32_t next;                    /* wavefront-tasks started (atomic); == nwt after a complete run */
	uint32_t abort;                   /* (the host zeroes these two words before every run) */
	uint32_t nwt;                     /* wavefront-tasks of the launch */
	uint32_t unp_fmt;                 /* wire format of `unp_src`: bits 31-30 = 1: four bits per code, 2: two bits per code + escapes (ksw2_lane.h, K2A_WIRE2_*); bits 29-0: the pairs' stride in the arena */
	const uint32_t *need;             /* [nwt] pieces that must have landed before the wavefront-task may start (0 = none) */
	const uint32_t *wm;               /* the plan's watermark block */
	uint64_t timeout_ticks;
	/* uniform plans on a wire format (ksw2_host_plan.c): the upload carries two or four residue codes per byte into `unp_src`; a
	 * wavefront-task first expands its own pairs' bytes -- unp_bytes per wavefront-task, the arena's first unp_total bytes in all --
	 * into the arena the kernels read (unp_dst).  unp_bytes = 0: the upload is the arena itself. */
	const uint8_t *unp_src;
	uint8_t *unp_dst;
	uint32_t unp_bytes, unp_total;
} K2aQueueDesc;                       /* 64 bytes */

/* deferred arg-max classes: the list of alignments whose book the fill froze (k2a_argmax_kernel -> k2a_zscan_kernel), uint32
 * { count, entries (task * 2 + half) ... }, lies in the traceback arena right in front of the class's first checkpoint block */
#define K2A_ZLIST_WORDS(ntasks) ((2 * (size_t)(ntasks) + 16 + 63) & ~(size_t)63)

/* Uniform plans (ksw2_host_plan.c "uniform batches"): a score-only batch whose pairs all have ONE shape and ONE set of parameters is
 * laid out by rule -- pair i's query at i * stride, its target at i * stride + qpad -- so its K2aPair records, its task list (pairs
 * 2t and 2t + 1 form task t) and a streamed launch's per-wavefront-task piece counts are functions of i: a small kernel
 * (k2a_uniform_layout_kernel) writes them where the fill kernels read them, and the host neither builds nor uploads them. */
#define K2A_UNI_MAXPIECES 48
typedef struct K2aUniform {
	K2aPair tmpl;                    /* everything but qoff / toff / tb_off */
	uint32_t n, ntasks, ng;          /* pairs; tasks (n / 2); tasks per wavefront-task (64 / G) */
	uint32_t stride, qpad;           /* bytes from one pair's query to the next pair's; from a pair's query to its target */
	uint32_t defer;                  /* deferred arg-max: one checkpoint block per wavefront-task */
	uint64_t blk_base, blk_bytes;    /* ... block wt at blk_base + wt * blk_bytes (K2aPair.tb_off of its pairs; bnd_off / cig_off come with the template) */
	uint32_t npieces, margin;        /* streamed launches: upload pieces; bytes past a target's end the kernels may touch */
	uint64_t seq_bytes;
	uint64_t pb[K2A_UNI_MAXPIECES + 1];    /* piece k = arena bytes [pb[k], pb[k + 1]) */
} K2aUniform;

/* per-group bookkeeping state (LDS on the GPU): the scalar reference's ez fields while rows complete */
typedef struct K2aBook {
	int32_t max, max_t, max_q, mqe, mqe_t, mte, mte_q, score, dropped, rows;
	int32_t inexact;             /* deferred arg-max kernels: a Z-drop could not be ruled out without the columns -> K2aResult.pad[1] */
} K2aBook;


#endif
