/* A ksw2 caller exactly as it would be written against the reference: it includes the REFERENCE's own ksw2.h
 * (-I/root/reference at compile time; the file is not copied) and is linked against libksw2_amd (here: the simulator
 * build, so the test runs without a GPU).  Reads pairs from stdin: "<algo> <w> <zdrop> <flag> <query> <target>" with
 * ACGTN strings, prints the ksw_extz_t fields and the CIGAR; tests/test_dropin_header.py compares with the oracle. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "ksw2.h"

static uint8_t code(char c) { return c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : c == 'T' ? 3 : 4; }

int main(void)
{
	char algo[32], *qs = (char*)malloc(1 << 20), *ts = (char*)malloc(1 << 20);
	int w, zdrop, flag, i;
	int8_t mat[25];
	ksw_extz_t ez;
	for (i = 0; i < 25; ++i) mat[i] = (int8_t)((i / 5 == 4 || i % 5 == 4) ? -1 : i / 5 == i % 5 ? 2 : -4);
#ifdef HAVE_KALLOC
	void *km = km_init();                             /* the reference's pool allocator: the library must grow ez.cigar in it */
#else
	void *km = 0;
#endif
	memset(&ez, 0, sizeof(ez));                       /* once; the callee reuses ez.cigar (README of the reference) */
	while (scanf("%31s %d %d %d %1048575s %1048575s", algo, &w, &zdrop, &flag, qs, ts) == 6) {
		const int ql = (int)strlen(qs), tl = (int)strlen(ts);
		uint8_t *q = (uint8_t*)malloc((size_t)ql), *t = (uint8_t*)malloc((size_t)tl);
		for (i = 0; i < ql; ++i) q[i] = code(qs[i]);
		for (i = 0; i < tl; ++i) t[i] = code(ts[i]);
		if (strcmp(algo, "extz2") == 0) ksw_extz2_sse(km, ql, q, tl, t, 5, mat, 4, 2, w, zdrop, 10, flag, &ez);
		else if (strcmp(algo, "extd2") == 0) ksw_extd2_sse(km, ql, q, tl, t, 5, mat, 4, 2, 24, 1, w, zdrop, 10, flag, &ez);
		else if (strcmp(algo, "exts2") == 0) ksw_exts2_sse(km, ql, q, tl, t, 5, mat, 4, 2, 32, 4, zdrop, 0, flag | KSW_EZ_SPLICE_FOR, 0, &ez);
		else if (strcmp(algo, "extf2") == 0) ksw_extf2_sse(km, ql, q, tl, t, 2, -4, 2, w, zdrop, &ez);
		else if (strcmp(algo, "gg2") == 0) {
			ez.score = ksw_gg2_sse(km, ql, q, tl, t, 5, mat, 4, 2, w, &ez.m_cigar, &ez.n_cigar, &ez.cigar);
			ez.max = 0; ez.zdropped = 0; ez.max_q = ez.max_t = ez.mqe_t = ez.mte_q = -1; ez.mqe = ez.mte = KSW_NEG_INF; ez.reach_end = 0;
		}
		printf("%d %u %d %d %d %d %d %d %u %d %d", ez.score, (unsigned)ez.max, ez.max_t, ez.max_q, ez.mqe, ez.mqe_t, ez.mte, ez.mte_q,
		       (unsigned)ez.zdropped, ez.reach_end, ez.n_cigar);
		for (i = 0; i < ez.n_cigar; ++i) printf(" %u", ez.cigar[i]);
		printf("\n");
		free(q); free(t);
	}
	kfree(km, ez.cigar);                              /* kalloc's kfree, or free() through the macro of ksw2.h:110 */
#ifdef HAVE_KALLOC
	km_destroy(km);
#endif
	free(qs); free(ts);
	return 0;
}
