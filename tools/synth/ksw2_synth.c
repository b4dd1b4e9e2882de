/*
 * ksw2_synth.c -- BENCH / TEST INFRASTRUCTURE: seeded synthetic read / reference pairs, fast enough
 * to generate the benchmark batches (hundreds of MB) in seconds.  Not part of libksw2_amd.so.
 *
 * Every pair has its own xorshift64* stream seeded from (seed, pair index), so any process regenerates any pair and the
 * work splits over threads without changing the data.
 *   target: i.i.d. uniform over {0,1,2,3};
 *   channel: per source base -- substitution with probability `sub`, deletion (geometric length, mean 1.5) with `ind`/2,
 *            insertion of uniform bases (geometric length, mean 1.5) after the base with `ind`/2;
 *   fixed shapes: query = channel(target) trimmed / padded with random bases to qlen; a fraction `tail_pairs` of the pairs
 *            gets the last `tail_frac` of the query replaced by random bases (so that Z-drop fires, config 3);
 *   ragged (config 5): qlen uniform in [lo, hi], query random, target = channel(query); redrawn while |tlen - qlen| > maxdiff.
 */
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct { uint64_t s; } rng_t;
static uint64_t splitmix(uint64_t x) { x += 0x9E3779B97F4A7C15ull; x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull; x = (x ^ (x >> 27)) * 0x94D049BB133111EBull; return x ^ (x >> 31); }
static void rng_seed(rng_t *r, uint64_t seed, uint64_t pair, uint64_t attempt) { r->s = splitmix(splitmix(seed) ^ splitmix(pair * 2 + 1) ^ (attempt << 48)); if (!r->s) r->s = 88172645463325252ull; }
static inline uint64_t rng_next(rng_t *r) { uint64_t x = r->s; x ^= x >> 12; x ^= x << 25; x ^= x >> 27; r->s = x; return x * 0x2545F4914F6CDD1Dull; }
static inline double rng_unit(rng_t *r) { return (double)(rng_next(r) >> 11) * (1.0 / 9007199254740992.0); }
static inline int rng_geo(rng_t *r) { int k = 1; while (rng_unit(r) > 2.0 / 3.0) ++k; return k; }      /* mean 1.5 */

static void fill_random(rng_t *r, uint8_t *dst, int64_t n)
{
	int64_t i = 0;
	while (i < n) {
		uint64_t x = rng_next(r);
		int k;
		for (k = 0; k < 32 && i < n; ++k, x >>= 2) dst[i++] = (uint8_t)(x & 3);
	}
}

/* src[0..L) through the channel; writes at most cap bases to dst (dst may be NULL: length only); returns the natural length */
static int64_t channel(rng_t *r, const uint8_t *src, int64_t L, double sub, double ind, uint8_t *dst, int64_t cap)
{
	int64_t i = 0, o = 0;
	const double pd = sub + ind / 2, pi = sub + ind;
	while (i < L) {
		const double u = rng_unit(r);
		if (u < sub) { const uint8_t b = (uint8_t)((src[i] + 1 + rng_next(r) % 3) & 3); if (dst && o < cap) dst[o] = b; ++o; ++i; }
		else if (u < pd) i += rng_geo(r);
		else if (u < pi) {
			int k = rng_geo(r);
			if (dst && o < cap) dst[o] = src[i];
			++o; ++i;
			for (; k > 0; --k) { const uint8_t b = (uint8_t)(rng_next(r) & 3); if (dst && o < cap) dst[o] = b; ++o; }
		} else { if (dst && o < cap) dst[o] = src[i]; ++o; ++i; }
	}
	return o;
}

typedef struct {
	int kind;                 /* 0 fixed, 1 ragged lengths, 2 ragged fill */
	uint64_t seed; int64_t first; int n;
	int qlen, tlen, lo, hi, maxdiff;
	double sub, ind, tail_pairs, tail_frac;
	uint8_t *q, *t;
	int32_t *ql, *tl; const int64_t *qoff, *toff;
	volatile int next;
} job_t;

static void one_fixed(const job_t *J, int i)
{
	rng_t r;
	uint8_t *q = J->q + (size_t)i * J->qlen, *t = J->t + (size_t)i * J->tlen;
	int64_t o;
	rng_seed(&r, J->seed, (uint64_t)(J->first + i), 0);
	fill_random(&r, t, J->tlen);
	o = channel(&r, t, J->tlen, J->sub, J->ind, q, J->qlen);
	if (o < J->qlen) fill_random(&r, q + o, J->qlen - o);
	if (J->tail_pairs > 0 && rng_unit(&r) < J->tail_pairs) {
		const int k = (int)(J->qlen * J->tail_frac);
		fill_random(&r, q + J->qlen - k, k);
	}
}

static void one_ragged(const job_t *J, int i, uint8_t *tmp)
{
	uint64_t attempt;
	for (attempt = 0;; ++attempt) {
		rng_t r, r2;
		int ql;
		int64_t tl;
		rng_seed(&r, J->seed, (uint64_t)(J->first + i), attempt);
		ql = J->lo + (int)(rng_next(&r) % (uint64_t)(J->hi - J->lo + 1));
		fill_random(&r, tmp, ql);
		r2 = r;
		tl = channel(&r2, tmp, ql, J->sub, J->ind, 0, 0);
		if (tl < 1 || (tl > ql ? tl - ql : ql - tl) > J->maxdiff) continue;
		if (J->kind == 1) { J->ql[i] = ql; J->tl[i] = (int32_t)tl; }
		else {
			memcpy(J->q + J->qoff[i], tmp, (size_t)ql);
			channel(&r, tmp, ql, J->sub, J->ind, J->t + J->toff[i], tl);
		}
		return;
	}
}

static void *worker(void *arg)
{
	job_t *J = (job_t*)arg;
	uint8_t *tmp = J->kind ? (uint8_t*)malloc((size_t)J->hi + 64) : 0;
	for (;;) {
		const int i = __sync_fetch_and_add(&J->next, 1);
		if (i >= J->n) break;
		if (J->kind == 0) one_fixed(J, i); else one_ragged(J, i, tmp);
	}
	free(tmp);
	return 0;
}

static void run(job_t *J, int threads)
{
	pthread_t th[64];
	int i;
	if (threads < 1) threads = 1;
	if (threads > 64) threads = 64;
	J->next = 0;
	for (i = 1; i < threads; ++i) pthread_create(&th[i], 0, worker, J);
	worker(J);
	for (i = 1; i < threads; ++i) pthread_join(th[i], 0);
}

void k2s_fixed(uint64_t seed, int64_t first, int n, int qlen, int tlen, double sub, double ind, double tail_pairs, double tail_frac,
               uint8_t *q, uint8_t *t, int threads)
{
	job_t J;
	memset(&J, 0, sizeof(J));
	J.kind = 0; J.seed = seed; J.first = first; J.n = n; J.qlen = qlen; J.tlen = tlen; J.sub = sub; J.ind = ind;
	J.tail_pairs = tail_pairs; J.tail_frac = tail_frac; J.q = q; J.t = t;
	run(&J, threads);
}

void k2s_ragged_lengths(uint64_t seed, int64_t first, int n, int lo, int hi, double sub, double ind, int maxdiff, int32_t *ql, int32_t *tl, int threads)
{
	job_t J;
	memset(&J, 0, sizeof(J));
	J.kind = 1; J.seed = seed; J.first = first; J.n = n; J.lo = lo; J.hi = hi; J.sub = sub; J.ind = ind; J.maxdiff = maxdiff; J.ql = ql; J.tl = tl;
	run(&J, threads);
}

void k2s_ragged_fill(uint64_t seed, int64_t first, int n, int lo, int hi, double sub, double ind, int maxdiff, const int64_t *qoff, const int64_t *toff,
                     uint8_t *q, uint8_t *t, int threads)
{
	job_t J;
	memset(&J, 0, sizeof(J));
	J.kind = 2; J.seed = seed; J.first = first; J.n = n; J.lo = lo; J.hi = hi; J.sub = sub; J.ind = ind; J.maxdiff = maxdiff; J.qoff = qoff; J.toff = toff;
	J.q = q; J.t = t;
	run(&J, threads);
}
