/*
 * coalesce_bench.c -- an UNCHANGED minimap2-style caller: T host threads, each calling ksw_extz2_sse / ksw_extd2_sse for one
 * pair at a time with its own ksw_extz_t (cli.c:50-132, README.md:54-87 pattern).  Measures calls/s through libksw2_amd.so
 * (which coalesces concurrent calls into device batches, ksw2_host_single.c::queue_one) and checks every result against the batch
 * entry point's result for the same pair (which the parity tests pin to the oracle).
 *
 *   coalesce_bench [threads=64] [calls per thread=2000] [len=512] [band=64] [cigar=0|1]
 * Prints one JSON line: {"threads":..,"calls":..,"seconds":..,"calls_per_s":..,"ms_per_call_per_thread":..,"mismatches":..,
 *                        "coalesced_calls":..,"coalesced_batches":..}
 */
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include "../include/ksw2_amd.h"

static int T = 64, K = 2000, L = 512, W = 64, CIG = 0, NP = 4096;
static uint8_t *Q, *Tg;
static int8_t mat[25];
static ksw_extz_t *expect;
static long mism;

static uint64_t rs = 88172645463325252ull;
static uint64_t rnd(void) { rs ^= rs << 13; rs ^= rs >> 7; rs ^= rs << 17; return rs; }
static double now(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + ts.tv_nsec * 1e-9; }

static int same(const ksw_extz_t *a, const ksw_extz_t *b)
{
	if (a->max != b->max || a->zdropped != b->zdropped || a->max_q != b->max_q || a->max_t != b->max_t || a->mqe != b->mqe || a->mqe_t != b->mqe_t ||
	    a->mte != b->mte || a->mte_q != b->mte_q || a->score != b->score || a->n_cigar != b->n_cigar || a->reach_end != b->reach_end) return 0;
	return a->n_cigar == 0 || memcmp(a->cigar, b->cigar, 4u * (size_t)a->n_cigar) == 0;
}

static void *worker(void *arg)
{
	const long me = (long)arg;
	ksw_extz_t ez;
	long bad = 0;
	int k;
	memset(&ez, 0, sizeof(ez));
	for (k = 0; k < K; ++k) {
		const int i = (int)((me * 7919 + k) % NP);
		if (CIG) ksw_extd2_sse(0, L, Q + (size_t)i * L, L, Tg + (size_t)i * L, 5, mat, 4, 2, 24, 1, W, 400, 0, 0, &ez);
		else ksw_extz2_sse(0, L, Q + (size_t)i * L, L, Tg + (size_t)i * L, 5, mat, 4, 2, W, -1, 0, KSW_EZ_SCORE_ONLY, &ez);
		bad += !same(&ez, &expect[i]);
	}
	free(ez.cigar);
	__sync_fetch_and_add(&mism, bad);
	return 0;
}

int main(int argc, char **argv)
{
	pthread_t *th;
	ksw2amd_scoring_t sc;
	ksw2amd_pair_t *pairs;
	int64_t st0[4], st1[4];
	double t0, dt;
	int i, j;
	if (argc > 1) T = atoi(argv[1]);
	if (argc > 2) K = atoi(argv[2]);
	if (argc > 3) L = atoi(argv[3]);
	if (argc > 4) W = atoi(argv[4]);
	if (argc > 5) CIG = atoi(argv[5]);
	Q = (uint8_t*)malloc((size_t)NP * L); Tg = (uint8_t*)malloc((size_t)NP * L);
	for (i = 0; i < NP; ++i)
		for (j = 0; j < L; ++j) {
			const uint8_t b = (uint8_t)(rnd() & 3);
			Tg[(size_t)i * L + j] = b;
			Q[(size_t)i * L + j] = rnd() % 100 < 8 ? (uint8_t)(rnd() & 3) : b;      /* substitutions only: both stay L long */
		}
	for (i = 0; i < 5; ++i) for (j = 0; j < 5; ++j) mat[i * 5 + j] = (int8_t)(i == 4 || j == 4 ? -1 : i == j ? 2 : -4);
	/* expected results: the batch entry point on the same pairs */
	sc.m = 5; sc.mat = mat; sc.q = 4; sc.e = 2; sc.q2 = 24; sc.e2 = 1;
	pairs = (ksw2amd_pair_t*)calloc((size_t)NP, sizeof(*pairs));
	expect = (ksw_extz_t*)calloc((size_t)NP, sizeof(*expect));
	for (i = 0; i < NP; ++i) {
		pairs[i].query = Q + (size_t)i * L; pairs[i].target = Tg + (size_t)i * L; pairs[i].qlen = pairs[i].tlen = L;
		pairs[i].w = W; pairs[i].zdrop = CIG ? 400 : -1; pairs[i].end_bonus = 0; pairs[i].flag = CIG ? 0 : KSW_EZ_SCORE_ONLY;
	}
	if ((CIG ? ksw2amd_extd_batch(0, &sc, NP, pairs, expect) : ksw2amd_extz_batch(0, &sc, NP, pairs, expect)) != KSW2AMD_OK) {
		fprintf(stderr, "batch failed: %s\n", ksw2amd_last_error());
		return 1;
	}
	th = (pthread_t*)malloc(sizeof(pthread_t) * (size_t)T);
	ksw2amd_host_stats(st0);
	t0 = now();
	for (i = 0; i < T; ++i) pthread_create(&th[i], 0, worker, (void*)(long)i);
	for (i = 0; i < T; ++i) pthread_join(th[i], 0);
	dt = now() - t0;
	ksw2amd_host_stats(st1);
	printf("{\"threads\": %d, \"calls\": %ld, \"len\": %d, \"band\": %d, \"cigar\": %d, \"seconds\": %.4f, \"calls_per_s\": %.1f, \"ms_per_call_per_thread\": %.4f, "
	       "\"mismatches\": %ld, \"coalesced_calls\": %lld, \"coalesced_batches\": %lld}\n", T, (long)T * K, L, W, CIG, dt, (double)T * K / dt, dt / K * 1e3, mism,
	       (long long)(st1[2] - st0[2]), (long long)(st1[3] - st0[3]));
	return mism != 0;
}
