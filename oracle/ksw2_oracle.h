/*
 * ksw2_oracle.h -- CPU restatement of the ksw2 extension/global alignment contract.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is linked into, imported by, or executed from
 * the product library (ksw2_amd/).  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may use it, and only as the checker.
 *
 * Parity status: PINNED.  tests/test_oracle_golden.py checks every function here against golden
 * vectors produced by the unmodified reference compiled from /root/reference (oracle/_ref, recipe in
 * oracle/Makefile, generator oracle/gen_golden.py) -- the known-answer table of SURVEY.md section 4.2
 * (test/t1.fa x test/q1.fa, MT-human x MT-orang under 8 settings) plus >3000 seeded random cases --
 * and tests/test_oracle_vs_ref.py re-checks it live against oracle/_ref when that artefact exists.
 *
 * What is restated (reference file:line it follows):
 *   kso_extz   <- ksw_extz   ksw2_extz.c:6-135      single affine gap, exact band, row-wise Z-drop
 *   kso_extd   <- ksw_extd   ksw2_extd.c:6-175      two-piece affine gap
 *   kso_gg     <- ksw_gg     ksw2_gg.c:6-102        global score + CIGAR
 *   kso_extz2 / kso_extd2 / kso_gg2
 *              <- the *calling contract* of ksw_extz2_sse / ksw_extd2_sse / ksw_gg2_sse
 *                 (ksw2_extz2_sse.c:56-82,292-301; ksw2_extd2_sse.c:75-100,389-406; ksw2_gg2_sse.c:11-126)
 *                 evaluated with the scalar cell semantics above: this is the contract the
 *                 MI355X kernels implement (SURVEY.md section 8a rules 1-10).
 *   kso_exts2  <- ksw_exts2_sse ksw2_exts2_sse.c:33-415 (splice-aware; ksw2_oracle_exts.c; the SSE code is the only definition)
 *   kso_extf2  <- ksw_extf2_sse ksw2_extf2_sse.c:11-98 (gap-linear X-drop extension; ksw2_oracle_extf.c; follows the SSE memory image)
 *   kso_extz2_sse / kso_extd2_sse <- ksw_extz2_sse / ksw_extd2_sse as they are (ksw2_extz2_sse.c:23-304, ksw2_extd2_sse.c:34-409):
 *                 the SSE memory image, for the opt-in SSE-compatible mode of the product (ksw2_oracle_sse.c)
 *   helpers    <- ksw2.h:113-123 (CIGAR push), :129-161 (traceback state machine, row-major case),
 *                 :163-182 (EQX rewrite), :184-189 (reset), :191-207 (Z-drop test)
 */
#ifndef KSW2_ORACLE_H_
#define KSW2_ORACLE_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KSO_NEG_INF (-0x40000000)

/* flag bits, numerically identical to KSW_EZ_* (ksw2.h:8-18) */
#define KSO_SCORE_ONLY  0x01
#define KSO_RIGHT       0x02
#define KSO_GENERIC_SC  0x04
#define KSO_APPROX_MAX  0x08
#define KSO_APPROX_DROP 0x10
#define KSO_EXTZ_ONLY   0x40
#define KSO_REV_CIGAR   0x80
#define KSO_SPLICE_FOR   0x100
#define KSO_SPLICE_REV   0x200
#define KSO_SPLICE_FLANK 0x400
#define KSO_EQX         0x800

/* Same memory layout as ksw_extz_t (ksw2.h:33-42): 56 bytes, cigar pointer at offset 48. */
typedef struct {
	uint32_t max:31, zdropped:1;
	int max_q, max_t;
	int mqe, mqe_t;
	int mte, mte_q;
	int score;
	int m_cigar, n_cigar;
	int reach_end;
	uint32_t *cigar;      /* malloc/realloc'ed; caller frees with free() */
} kso_extz_t;

/* scalar contract (bit-for-bit what ksw_extz / ksw_extd / ksw_gg return for valid inputs) */
void kso_extz(int qlen, const uint8_t *query, int tlen, const uint8_t *target, int8_t m, const int8_t *mat,
              int8_t q, int8_t e, int w, int zdrop, int flag, kso_extz_t *ez);
void kso_extd(int qlen, const uint8_t *query, int tlen, const uint8_t *target, int8_t m, const int8_t *mat,
              int8_t q, int8_t e, int8_t q2, int8_t e2, int w, int zdrop, int flag, kso_extz_t *ez);
int  kso_gg(int qlen, const uint8_t *query, int tlen, const uint8_t *target, int8_t m, const int8_t *mat,
            int8_t q, int8_t e, int w, int *m_cigar, int *n_cigar, uint32_t **cigar);

/* "...2_sse" calling contract with scalar cell semantics == what the GPU path must produce */
void kso_extz2(int qlen, const uint8_t *query, int tlen, const uint8_t *target, int8_t m, const int8_t *mat,
               int8_t q, int8_t e, int w, int zdrop, int end_bonus, int flag, kso_extz_t *ez);
void kso_extd2(int qlen, const uint8_t *query, int tlen, const uint8_t *target, int8_t m, const int8_t *mat,
               int8_t q, int8_t e, int8_t q2, int8_t e2, int w, int zdrop, int end_bonus, int flag, kso_extz_t *ez);
int  kso_gg2(int qlen, const uint8_t *query, int tlen, const uint8_t *target, int8_t m, const int8_t *mat,
             int8_t q, int8_t e, int w, int *m_cigar, int *n_cigar, uint32_t **cigar);

/* what ksw_extz2_sse / ksw_extd2_sse themselves return -- 16-position blocks ("leaky band"), anti-diagonal Z-drop, padded mte_q,
 * KSW_EZ_APPROX_MAX / APPROX_DROP heuristics (ksw2_oracle_sse.c; follows the SSE memory image) */
void kso_extz2_sse(int qlen, const uint8_t *query, int tlen, const uint8_t *target, int8_t m, const int8_t *mat,
                   int8_t q, int8_t e, int w, int zdrop, int end_bonus, int flag, kso_extz_t *ez);
void kso_extd2_sse(int qlen, const uint8_t *query, int tlen, const uint8_t *target, int8_t m, const int8_t *mat,
                   int8_t q, int8_t e, int8_t q2, int8_t e2, int w, int zdrop, int end_bonus, int flag, kso_extz_t *ez);

/* gap-linear X-drop extension, score only (ksw2_extf2_sse.c:11); mch / mis / e as in the reference's signature */
void kso_extf2(int qlen, const uint8_t *query, int tlen, const uint8_t *target, int8_t mch, int8_t mis, int8_t e, int w, int xdrop,
               kso_extz_t *ez);

/* number of DP cells inside the exact band |i-j|<=w (SURVEY.md section 8d metric definition) */
int64_t kso_band_cells(int qlen, int tlen, int w);
/* Score of a CIGAR (M / I / D runs from the start of both sequences, in the caller's order: ksw2.h:22-27 encoding, len << 4 | op) under
 * the affine (q2 < 0) or two-piece affine gap model: sum of mat[t * m + q] over M columns minus min(q + l e, q2 + l e2) per gap run.
 * *qused / *tused = bases consumed.  A size-independent property check for batches far too big for the oracle itself: an optimal
 * alignment's CIGAR re-scores to the reported score (ksw2_extz.c:127-133 start cells).  Returns KSO_NEG_INF on a malformed CIGAR. */
int kso_cigar_score(int n_cigar, const uint32_t *cigar, int qlen, const uint8_t *query, int tlen, const uint8_t *target, int m, const int8_t *mat,
                    int q, int e, int q2, int e2, int *qused, int *tused);

/* splice-aware extension (ksw2_oracle_exts.c): contract of ksw_exts2_sse, ksw2.h:73-74 */
void kso_exts2(int qlen, const uint8_t *query, int tlen, const uint8_t *target, int8_t m, const int8_t *mat,
               int8_t q, int8_t e, int8_t q2, int8_t noncan, int zdrop, int8_t junc_bonus, int flag, const uint8_t *junc,
               kso_extz_t *ez);
void kso_splice_signals(int tlen, const uint8_t *target, int noncan, int junc_bonus, int flag, const uint8_t *junc,
                        int8_t *donor, int8_t *acceptor);
int kso_long_thres(int q, int e, int q2);

#ifdef __cplusplus
}
#endif
#endif
