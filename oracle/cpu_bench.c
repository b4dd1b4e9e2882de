/*
 * cpu_bench.c -- TEST/BENCH INFRASTRUCTURE: times a CPU implementation of ksw_extz2_sse / ksw_extd2_sse on
 * T host threads (pthreads, one ksw_extz_t per thread, km = NULL, pairs pulled from an atomic counter), as
 * BASELINE.md section 3 prescribes.  The implementation is passed in as function pointers, so the same loop
 * times the compiled reference (oracle/_ref/libksw2ref.so, "reference") or the oracle port ("port").
 * Used only by bench.py's cpu_baseline leg.
 */
#define _GNU_SOURCE
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include "ksw2_oracle.h"

typedef void (*extz2_fn)(void *km, int qlen, const uint8_t *query, int tlen, const uint8_t *target, int8_t m, const int8_t *mat,
                         int8_t q, int8_t e, int w, int zdrop, int end_bonus, int flag, kso_extz_t *ez);
typedef void (*extd2_fn)(void *km, int qlen, const uint8_t *query, int tlen, const uint8_t *target, int8_t m, const int8_t *mat,
                         int8_t q, int8_t e, int8_t q2, int8_t e2, int w, int zdrop, int end_bonus, int flag, kso_extz_t *ez);

typedef void (*exts2_fn)(void *km, int qlen, const uint8_t *query, int tlen, const uint8_t *target, int8_t m, const int8_t *mat,
                         int8_t q, int8_t e, int8_t q2, int8_t noncan, int zdrop, int8_t junc_bonus, int flag, const uint8_t *junc, kso_extz_t *ez);

typedef struct {
	void *fn; int dual, with_km;                   /* dual: 0 extz2, 1 extd2, 2 exts2 (gq2 = long-gap open, ge2 = noncan) */
	const uint8_t *const *q, *const *t; const int32_t *qlen, *tlen; int n;       /* n pairs of any shapes; index i mod n is aligned next */
	int8_t m; const int8_t *mat; int8_t gq, ge, gq2, ge2; int w, zdrop, flag;
	double seconds;
	volatile long next;
	long done;
	pthread_mutex_t mu;
} job_t;

typedef void (*extf2_fn)(void *km, int qlen, const uint8_t *query, int tlen, const uint8_t *target, int8_t mch, int8_t mis, int8_t e, int w,
                         int xdrop, kso_extz_t *ez);

static double now(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + ts.tv_nsec * 1e-9; }

static void *worker(void *arg)
{
	job_t *J = (job_t*)arg;
	kso_extz_t ez;
	long mine = 0;
	double t0 = now();
	memset(&ez, 0, sizeof(ez));
	while (now() - t0 < J->seconds) {
		long i = __sync_fetch_and_add(&J->next, 1) % J->n;
		const uint8_t *q = J->q[i], *t = J->t[i];
		const int ql = J->qlen[i], tl = J->tlen[i];
		if (J->dual == 3) ((extf2_fn)J->fn)(0, ql, q, tl, t, J->gq, J->ge, J->gq2, J->w, J->zdrop, &ez);      /* mch, mis, e */
		else if (J->dual == 2) ((exts2_fn)J->fn)(0, ql, q, tl, t, J->m, J->mat, J->gq, J->ge, J->gq2, J->ge2, J->zdrop, 0, J->flag, 0, &ez);
		else if (J->dual) ((extd2_fn)J->fn)(0, ql, q, tl, t, J->m, J->mat, J->gq, J->ge, J->gq2, J->ge2, J->w, J->zdrop, 0, J->flag, &ez);
		else ((extz2_fn)J->fn)(0, ql, q, tl, t, J->m, J->mat, J->gq, J->ge, J->w, J->zdrop, 0, J->flag, &ez);
		++mine;
	}
	free(ez.cigar);
	pthread_mutex_lock(&J->mu); J->done += mine; pthread_mutex_unlock(&J->mu);
	return 0;
}

/* returns pairs completed (indices 0, 1, ... mod n, each one finished); *elapsed = wall seconds.  `reserved` must be NULL. */
long kso_cpu_bench(void *fn, int dual, int threads, double seconds, int n, const uint8_t *const *q, const uint8_t *const *t,
                   const int32_t *qlen, const int32_t *tlen, const void *reserved, int8_t m, const int8_t *mat, int8_t gq, int8_t ge, int8_t gq2, int8_t ge2, int w, int zdrop, int flag, double *elapsed)
{
	job_t J;
	pthread_t *th = (pthread_t*)malloc(sizeof(pthread_t) * (size_t)threads);
	int i;
	double t0;
	(void)reserved;
	memset(&J, 0, sizeof(J));
	J.fn = fn; J.dual = dual; J.q = q; J.t = t; J.n = n; J.qlen = qlen; J.tlen = tlen; J.m = m; J.mat = mat;
	J.gq = gq; J.ge = ge; J.gq2 = gq2; J.ge2 = ge2; J.w = w; J.zdrop = zdrop; J.flag = flag; J.seconds = seconds;
	pthread_mutex_init(&J.mu, 0);
	t0 = now();
	for (i = 0; i < threads; ++i) pthread_create(&th[i], 0, worker, &J);
	for (i = 0; i < threads; ++i) pthread_join(th[i], 0);
	*elapsed = now() - t0;
	free(th);
	return J.done;
}

/* adapters so the oracle port can be timed through the same signatures */
void kso_extz2_km(void *km, int qlen, const uint8_t *query, int tlen, const uint8_t *target, int8_t m, const int8_t *mat,
                  int8_t q, int8_t e, int w, int zdrop, int end_bonus, int flag, kso_extz_t *ez)
{ (void)km; kso_extz2(qlen, query, tlen, target, m, mat, q, e, w, zdrop, end_bonus, flag, ez); }
void kso_extd2_km(void *km, int qlen, const uint8_t *query, int tlen, const uint8_t *target, int8_t m, const int8_t *mat,
                  int8_t q, int8_t e, int8_t q2, int8_t e2, int w, int zdrop, int end_bonus, int flag, kso_extz_t *ez)
{ (void)km; kso_extd2(qlen, query, tlen, target, m, mat, q, e, q2, e2, w, zdrop, end_bonus, flag, ez); }
void kso_exts2_km(void *km, int qlen, const uint8_t *query, int tlen, const uint8_t *target, int8_t m, const int8_t *mat,
                  int8_t q, int8_t e, int8_t q2, int8_t noncan, int zdrop, int8_t junc_bonus, int flag, const uint8_t *junc, kso_extz_t *ez)
{ (void)km; kso_exts2(qlen, query, tlen, target, m, mat, q, e, q2, noncan, zdrop, junc_bonus, flag, junc, ez); }
void kso_extf2_km(void *km, int qlen, const uint8_t *query, int tlen, const uint8_t *target, int8_t mch, int8_t mis, int8_t e, int w, int xdrop,
                  kso_extz_t *ez)
{ (void)km; kso_extf2(qlen, query, tlen, target, mch, mis, e, w, xdrop, ez); }
