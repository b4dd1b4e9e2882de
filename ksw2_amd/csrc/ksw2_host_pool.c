/*
 * ksw2_host_pool.c -- worker pool: chunked, pipelined, multi-device batches; gathers; the batch entry points; flat batches.
 */
#include "ksw2_host_int.h"
#if defined(__SSE2__)
#include <emmintrin.h>                    /* pack4: the 4-bit wire format's packing loop; a host without SSE2 takes its scalar tail loop for everything */
#endif

/* ---------------------------------------------------------------- worker pool: chunked, pipelined, multi-device batches
 * A large batch handed to one ksw2amd_ext?_batch call is cut into chunks of consecutive pairs that a few persistent worker
 * threads pull from a shared counter.  Every worker packs, uploads, computes and fetches on a stream (and with pinned staging
 * and device buffers) of its own, so chunk i+1 is packed and uploaded while chunk i computes and chunk i-1's results come
 * back: one calling thread gets the device-bound rate instead of the sum of the phases.  With ksw2amd_set_devices() the
 * workers belong to several GPUs and the same counter shards the batch over them (pairs are independent: no collective). */
typedef struct {
	chunk_fn fn; void *ctx;
	int nchunks; const int *cbeg;               /* chunk c = pairs [cbeg[c], cbeg[c + 1]) */
	int next;                                   /* next chunk, atomic */
	int ndev, dev[POOL_MAXDEV], share;          /* devices of the job, worker threads per device */
	int flush;                                  /* instead of chunks: every worker returns its cached buffers */
	int quiet;                                  /* not a batch: keep it out of the host statistics */
	int rc; char err[512];                      /* first failure */
	int pending;                                /* participating workers still busy */
} job_t;
static struct {
	pthread_mutex_t mu;
	pthread_cond_t work, done, idle;
	int nw, gen, busy;
	int dev[POOL_MAXW], rank[POOL_MAXW];        /* per worker: its device, its index among that device's workers */
	job_t *job;
} g_pool = { PTHREAD_MUTEX_INITIALIZER, PTHREAD_COND_INITIALIZER, PTHREAD_COND_INITIALIZER, PTHREAD_COND_INITIALIZER, 0, 0, 0, {0}, {0}, 0 };
int64_t g_stat[4];                          /* pooled batches, their chunks, coalesced single calls, the batches they formed */
int g_ndev_set, g_dev_set[POOL_MAXDEV];     /* ksw2amd_set_devices(); 0 = the calling thread's device */
__thread int g_is_worker;

static int job_has_dev(const job_t *j, int dev)
{
	int i;
	for (i = 0; i < j->ndev; ++i) if (j->dev[i] == dev) return 1;
	return 0;
}

typedef struct { int idx, dev, seen; } worker_arg_t;

/* The pool's threads run on the cores next to their GPU: its PCI function's local_cpulist under /sys (the NUMA node the device hangs
 * off).  What they do all day is copy sequences into page-locked staging that the device's DMA engines read: from the other socket
 * every byte crosses the inter-socket link twice.  Intersected with the affinity the process already has (a launcher's or bench.py's
 * per-rank pinning stays in force); no /sys entry, an empty intersection or KSW2AMD_PIN=0: nothing is changed. */
static void pin_worker_to_device_node(void)
{
	char bdf[64], path[160], line[4096];
	cpu_set_t have, want;
	FILE *fp;
	char *c;
	int n = 0;
	if (ENV(PIN) && atoi(ENV(PIN)) == 0) return;
	if (k2a_shim_pci_bus_id(bdf, (int)sizeof(bdf)) || sched_getaffinity(0, sizeof(have), &have)) return;
	snprintf(path, sizeof(path), "/sys/bus/pci/devices/%s/local_cpulist", bdf);
	fp = fopen(path, "r");
	if (!fp) return;
	if (!fgets(line, sizeof(line), fp)) { fclose(fp); return; }
	fclose(fp);
	CPU_ZERO(&want);
	for (c = line; *c && *c != '\n'; ) {                    /* "0-63,128-191" */
		char *e;
		long a = strtol(c, &e, 10), b = a;
		if (e == c) break;
		if (*e == '-') { c = e + 1; b = strtol(c, &e, 10); }
		for (; a <= b && a < CPU_SETSIZE; ++a) if (CPU_ISSET((int)a, &have)) { CPU_SET((int)a, &want); ++n; }
		c = *e == ',' ? e + 1 : e;
	}
	if (n >= 2) sched_setaffinity(0, sizeof(want), &want);
}

static void *pool_worker(void *arg_)
{
	worker_arg_t *arg = (worker_arg_t*)arg_;
	const int dev = arg->dev, rank = arg->idx;           /* rank among the workers of this device */
	int seen = arg->seen;
	free(arg);
	g_is_worker = 1;
	k2a_shim_set_device(dev);
	pin_worker_to_device_node();
	/* every second worker's kernels on a high-priority stream: the runtime deals streams onto four hardware queues PER PRIORITY, so eight
	 * workers' streams of one priority share queues in pairs and one chunk's traceback walk holds up the next chunk's fill behind it
	 * (config 3: 19.9 -> 18.8 ms per batch, 10 k with CIGAR 32.4 -> 30.3; KSW2AMD_WORKER_PRIO=0: all ordinary) */
	{ extern __thread int g_stream_high; const char *e = ENV(WORKER_PRIO); g_stream_high = !(e && *e && atoi(e) == 0) && (rank & 1); }
	pthread_mutex_lock(&g_pool.mu);
	for (;;) {
		job_t *j;
		while (g_pool.gen == seen) pthread_cond_wait(&g_pool.work, &g_pool.mu);
		seen = g_pool.gen; j = g_pool.job;
		if (!j || !job_has_dev(j, dev)) continue;
		if (!j->flush && rank >= j->share) continue;        /* not one of this job's workers: it neither works nor is waited for (pool_start counted `share` per device) */
		pthread_mutex_unlock(&g_pool.mu);
		if (j->flush) release_thread_cache();
		else if (rank < j->share) {                         /* a batch of few chunks goes to the same workers every time: their buffer
		                                                     * caches fit it, the others' need not be filled (10 k with CIGAR inside the
		                                                     * default bench run: 680 GCUPS while all six workers took turns, 1 265 alone) */
			pend_t pd = { 0, 0 };
			for (;;) {
				const int c = __sync_fetch_and_add(&j->next, 1);
				const int last = c >= j->nchunks || j->rc;
				const int rc = last ? j->fn(j->ctx, -1, -1, j->share, &pd) : j->fn(j->ctx, j->cbeg[c], j->cbeg[c + 1], j->share, &pd);
				if (rc) {
					pthread_mutex_lock(&g_pool.mu);
					if (!j->rc) { j->rc = rc; snprintf(j->err, sizeof(j->err), "%s", g_err); }
					pthread_mutex_unlock(&g_pool.mu);
				}
				if (last) break;
			}
		}
		pthread_mutex_lock(&g_pool.mu);
		if (--j->pending == 0) pthread_cond_broadcast(&g_pool.done);
	}
	return 0;
}

/* > 0: the batch being cut wants this many workers per device instead of the default (run_batch: batches with CIGARs) */
static __thread int g_job_threads;
int pool_threads_per_device(void)
{
	const char *e = ENV(THREADS);
	int t = e ? atoi(e) : g_job_threads > 0 ? g_job_threads : 6;
	return t < 0 ? 0 : t > 16 ? 16 : t;
}

/* pool_start + pool_wait = pool_run in two halves: the submitting thread does something else while the workers run the job (the
 * gather of a streamed plan: plan_create_ex goes on to lay the plan out and launch it).  Between the two the pool is taken: other
 * submitters, this thread included, run their work inline. */
static int pool_start(job_t *j)
{
	int i, d, have;
	if (g_is_worker) return -1;
	pthread_mutex_lock(&g_pool.mu);
	if (g_pool.busy) { pthread_mutex_unlock(&g_pool.mu); return -1; }      /* busy with another caller's batch (or this thread's gather): that caller runs inline */
	g_pool.busy = 1;                        /* (a flag under `mu`, not a mutex held across calls: the plan that owns a gather may be fetched by another thread) */
	for (d = 0; d < j->ndev; ++d) {
		for (i = 0, have = 0; i < g_pool.nw; ++i) have += g_pool.dev[i] == j->dev[d];
		for (; have < j->share && g_pool.nw < POOL_MAXW; ++have) {
			pthread_t th;
			pthread_attr_t at;
			worker_arg_t *wa = (worker_arg_t*)malloc(sizeof(*wa));
			if (!wa) break;
			wa->idx = have; wa->dev = j->dev[d]; wa->seen = g_pool.gen;
			pthread_attr_init(&at);
			pthread_attr_setdetachstate(&at, PTHREAD_CREATE_DETACHED);
			if (pthread_create(&th, &at, pool_worker, wa)) { free(wa); pthread_attr_destroy(&at); break; }
			pthread_attr_destroy(&at);
			g_pool.rank[g_pool.nw] = have; g_pool.dev[g_pool.nw++] = j->dev[d];
		}
	}
	/* the job's workers: the first `share` of every device of the job (all of them for a cache flush).  A pool that a gather grew to 17
	 * or 24 threads used to make EVERY later job wait for all of them to wake up and check in -- eight-chunk batches of config 2
	 * behind a uniform plan's gather: 1 400 -> 870 GCUPS through the flat entry (round 5) */
	for (i = 0, j->pending = 0; i < g_pool.nw; ++i) j->pending += job_has_dev(j, g_pool.dev[i]) && (j->flush || g_pool.rank[i] < j->share);
	if (j->pending == 0) { g_pool.busy = 0; pthread_mutex_unlock(&g_pool.mu); return -1; }
	g_pool.job = j; ++g_pool.gen;
	if (!j->flush && !j->quiet) { g_stat[0] += 1; g_stat[1] += j->nchunks; }
	pthread_cond_broadcast(&g_pool.work);
	pthread_mutex_unlock(&g_pool.mu);
	return 0;
}
static void pool_wait(job_t *j)
{
	pthread_mutex_lock(&g_pool.mu);
	while (j->pending > 0) pthread_cond_wait(&g_pool.done, &g_pool.mu);
	g_pool.job = 0; g_pool.busy = 0;
	pthread_cond_broadcast(&g_pool.idle);
	pthread_mutex_unlock(&g_pool.mu);
}
/* run `j` on the pool (workers for its devices are created on first use); returns -1 if the pool cannot take it now */
static int pool_run(job_t *j)
{
	if (pool_start(j)) return -1;
	pool_wait(j);
	return 0;
}

void ksw2amd_host_stats(int64_t out[4])
{
	int i;
	for (i = 0; i < 4; ++i) out[i] = g_stat[i];
}

int ksw2amd_set_devices(int n, const int *devices)
{
	int i;
	if (n < 0 || n > POOL_MAXDEV || (n > 0 && !devices)) return fail(KSW2AMD_E_PARAM, "set_devices: bad arguments%s", 0);
	for (i = 0; i < n; ++i)
		if (devices[i] < 0 || devices[i] >= k2a_shim_device_count()) return fail(KSW2AMD_E_NODEVICE, "set_devices: no such device%s", 0);
	pthread_mutex_lock(&g_pool.mu);
	while (g_pool.busy) pthread_cond_wait(&g_pool.idle, &g_pool.mu);       /* not under a running batch */
	for (i = 0; i < n; ++i) g_dev_set[i] = devices[i];
	g_ndev_set = n;
	pthread_mutex_unlock(&g_pool.mu);
	return KSW2AMD_OK;
}

void ksw2amd_release_cache(void)
{
	release_thread_cache();
	if (!g_is_worker && g_pool.nw > 0) {       /* and the pool's threads */
		job_t j;
		int i;
		memset(&j, 0, sizeof(j));
		j.flush = 1;
		pthread_mutex_lock(&g_pool.mu);
		for (i = 0; i < g_pool.nw && j.ndev < POOL_MAXDEV; ++i) if (!job_has_dev(&j, g_pool.dev[i])) j.dev[j.ndev++] = g_pool.dev[i];
		pthread_mutex_unlock(&g_pool.mu);
		j.share = 0;
		pool_run(&j);
	}
}

/* cut [0, n) into at most `nchunks` chunks of consecutive pairs of about equal cost; cost[i] >= 1.  Returns the chunk count, cbeg[0..count] */
static int make_chunks(int n, const double *cost, double total, int nchunks, int workers, int chunk_pairs, int *cbeg)
{
	if (chunk_pairs > 0) {                                /* batches of one shape: whole device fills (uniform_chunks) */
		int c = 0, b;
		for (b = 0; b < n && c < nchunks; b += chunk_pairs) cbeg[c++] = b;
		cbeg[c] = n;
		return c;
	}
	if (chunk_pairs < 0) {                                /* ... growing: two chunks of that size, then chunks of twice the size (uniform_chunks) */
		const int cp = -chunk_pairs;
		int c = 0, b = 0;
		while (b < n) { cbeg[c] = b; b += c < 2 ? cp : 2 * cp; ++c; }
		cbeg[c] = n;
		return c;
	}
	const int nc = nchunks;
	double acc = 0, edge = 0;
	int i, c = 0;
	(void)workers;
	cbeg[0] = 0;
	edge = total / nc;
	for (i = 0; i < n; ++i) {
		acc += cost[i];
		if (c + 1 < nc && acc >= edge && i + 1 < n) {
			cbeg[++c] = i + 1;
			edge += total / nc;
		}
	}
	cbeg[++c] = n;
	return c;
}

void copy_range(const copy_ctx_t *c, int beg, int end)
{
	int i;
	for (i = beg; i < end; ++i) {
		const ksw2amd_pair_t *a = &c->pairs[i];
		if (a->qlen <= 0 || a->tlen <= 0) continue;
		/* the scan looks for wildcard codes in the TARGET: the packed kernels' row profiles hold target codes 0..3 (pk_scoring); a
		 * query's wildcard is an entry of their column-profile table like any other code */
		memcpy(c->h_seq + c->hp[i].qoff, a->query, (size_t)a->qlen);
		if (c->wild) c->wild[i] = (uint8_t)copy_scan(c->h_seq + c->hp[i].toff, a->target, a->tlen);
		else memcpy(c->h_seq + c->hp[i].toff, a->target, (size_t)a->tlen);      /* unscanned (streamed plans) */
		memset(c->h_seq + c->hp[i].toff + a->tlen, 0, 64);                                          /* rows read past the target end */
	}
}
/* The 4-bit wire format of uniform plans: two residue codes per byte into the staging buffer (half the bytes for the gather to write
 * and for the DMA engines to move: config 2's step waits for its upload), expanded on the device by the wavefront that needs them
 * (k2a_queue_wait).  Returns the OR of all source bytes: a code above 15 does not fit and sends the batch to the general path. */
static unsigned pack4(uint8_t *dst, const uint8_t *src, int n)
{
	unsigned bad = 0;
	int k = 0;
#if defined(__SSE2__)
	const __m128i lo = _mm_set1_epi16(0x00ff);
	__m128i acc = _mm_setzero_si128();
	for (; k + 32 <= n; k += 32) {
		const __m128i a = _mm_loadu_si128((const __m128i*)(src + k)), b = _mm_loadu_si128((const __m128i*)(src + k + 16));
		/* 16-bit lanes { even byte, odd byte } -> even | odd << 4 in the low byte */
		const __m128i pa = _mm_and_si128(_mm_or_si128(a, _mm_srli_epi16(a, 4)), lo), pb = _mm_and_si128(_mm_or_si128(b, _mm_srli_epi16(b, 4)), lo);
		acc = _mm_or_si128(acc, _mm_or_si128(a, b));
		_mm_storeu_si128((__m128i*)(dst + (k >> 1)), _mm_packus_epi16(pa, pb));
	}
	{
		uint8_t t[16];
		int x;
		_mm_storeu_si128((__m128i*)t, acc);
		for (x = 0; x < 16; ++x) bad |= t[x];
	}
#endif      /* (any other host: the byte loop below takes the whole range) */
	for (; k + 2 <= n; k += 2) { bad |= src[k] | src[k + 1]; dst[k >> 1] = (uint8_t)((src[k] & 15) | (src[k + 1] << 4)); }
	if (k < n) { bad |= src[k]; dst[k >> 1] = (uint8_t)(src[k] & 15); }
	return bad;
}
/* The 2-bit wire format (ksw2_lane.h, K2A_WIRE2_*): four codes per byte.  pack2: the plain loop for a sequence without a code above 3,
 * returning the OR of its bytes; the caller packs a sequence with such a code again through pack2_esc, which writes 0 for it and adds
 * an escape entry per run -- { offset inside the pair's region + run length + code } -- to the pair's slot.  Returns the number of entries
 * the pair holds afterwards, or -1 when they do not fit (or a code above 15): the batch then takes the general path. */
static unsigned pack2(uint8_t *dst, const uint8_t *src, int n)
{
	unsigned bad = 0;
	int k = 0;
#if defined(__SSE2__)
	const __m128i lo = _mm_set1_epi16(0x00ff);
	__m128i acc = _mm_setzero_si128();
	for (; k + 64 <= n; k += 64) {
		__m128i v[4], h[2];
		int x;
		for (x = 0; x < 4; ++x) {
			v[x] = _mm_loadu_si128((const __m128i*)(src + k + 16 * x));
			acc = _mm_or_si128(acc, v[x]);
			v[x] = _mm_and_si128(_mm_or_si128(v[x], _mm_srli_epi16(v[x], 6)), lo);      /* 16-bit lanes: { even code | odd code << 2 } in the low byte */
		}
		h[0] = _mm_packus_epi16(v[0], v[1]); h[1] = _mm_packus_epi16(v[2], v[3]);     /* bytes of two codes each */
		h[0] = _mm_and_si128(_mm_or_si128(h[0], _mm_srli_epi16(h[0], 4)), lo);
		h[1] = _mm_and_si128(_mm_or_si128(h[1], _mm_srli_epi16(h[1], 4)), lo);
		_mm_storeu_si128((__m128i*)(dst + (k >> 2)), _mm_packus_epi16(h[0], h[1]));     /* bytes of four codes */
	}
	{
		uint8_t t[16];
		int x;
		_mm_storeu_si128((__m128i*)t, acc);
		for (x = 0; x < 16; ++x) bad |= t[x];
	}
#endif
	for (; k + 4 <= n; k += 4) { bad |= src[k] | src[k + 1] | src[k + 2] | src[k + 3]; dst[k >> 2] = (uint8_t)((src[k] & 3) | ((src[k + 1] & 3) << 2) | ((src[k + 2] & 3) << 4) | (src[k + 3] << 6)); }
	if (k < n) {
		unsigned b = 0;
		int x;
		for (x = 0; k + x < n; ++x) { bad |= src[k + x]; b |= (unsigned)(src[k + x] & 3) << (2 * x); }
		dst[k >> 2] = (uint8_t)b;
	}
	return bad;
}
static int pack2_esc(uint8_t *dst, const uint8_t *src, int n, uint32_t region_off, uint8_t *slot, int used)
{
	int k = 0;
	memset(dst, 0, (size_t)(n + 3) >> 2);
	while (k < n) {
		const unsigned c = src[k];
		if (c < 4) { dst[k >> 2] |= (uint8_t)(c << (2 * (k & 3))); ++k; continue; }
		{
			int len = 1;
			uint32_t ent;
			while (k + len < n && src[k + len] == c && len < 255) ++len;
			if (c > 15 || used >= K2A_WIRE2_ESC) return -1;
			ent = (region_off + (uint32_t)k) | ((uint32_t)len << 20) | ((uint32_t)c << 28);
			memcpy(slot + 4 * used, &ent, 4);
			++used;
			k += len;
		}
	}
	return used;
}
static void pack_range(const copy_ctx_t *c, int beg, int end)
{
	unsigned bad = 0;
	int i;
	if (c->su->wire4 == 2) {
		const uint32_t stride = c->su->wire_stride;
		for (i = beg; i < end; ++i) {
			const ksw2amd_pair_t *a = &c->pairs[i];
			uint8_t *q, *t, *slot;
			int used = 0;
			if (a->qlen <= 0 || a->tlen <= 0) continue;
			q = c->h_seq + (c->hp[i].qoff >> 2); t = c->h_seq + (c->hp[i].toff >> 2);
			slot = c->h_seq + (((size_t)c->hp[i].qoff + stride) >> 2) - K2A_WIRE2_SLOT;
			memset(slot, 0, K2A_WIRE2_SLOT);
			if (pack2(q, a->query, a->qlen) > 3) used = pack2_esc(q, a->query, a->qlen, 0, slot, used);
			if (used >= 0 && pack2(t, a->target, a->tlen) > 3) used = pack2_esc(t, a->target, a->tlen, c->hp[i].toff - c->hp[i].qoff, slot, used);
			if (used < 0) bad = 16;
			memset(t + ((a->tlen + 3) >> 2), 0, 16);                                 /* rows read past the target end (copy_range) */
		}
		if (bad > 15) __sync_fetch_and_or(&c->su->wire_bad, 1);
		return;
	}
	for (i = beg; i < end; ++i) {
		const ksw2amd_pair_t *a = &c->pairs[i];
		uint8_t *t;
		if (a->qlen <= 0 || a->tlen <= 0) continue;
		bad |= pack4(c->h_seq + (c->hp[i].qoff >> 1), a->query, a->qlen);
		t = c->h_seq + (c->hp[i].toff >> 1);
		bad |= pack4(t, a->target, a->tlen);
		memset(t + ((a->tlen + 1) >> 1), 0, 32);                                 /* rows read past the target end (copy_range) */
	}
	if (bad > 15) __sync_fetch_and_or(&c->su->wire_bad, 1);
}
void copy_or_pack_range(const copy_ctx_t *c, int beg, int end) { if (c->su && c->su->wire4) pack_range(c, beg, end); else copy_range(c, beg, end); }
static int copy_chunk(void *ctx, int beg, int end, int share, pend_t *pd)
{
	const copy_ctx_t *c = (const copy_ctx_t*)ctx;
	(void)share; (void)pd;
	if (beg >= 0) {
		if (c->uni_plan)                                    /* a uniform plan: the host's per-pair arrays of this range first (copy_range reads the offsets) */
			uni_fill_range(c->uni_plan->uni, c->uni_plan->h_pairs, c->uni_plan->h_cls, c->uni_plan->h_flag, c->uni_plan->h_order, c->uni_cls, c->uni_flag, beg, end);
		copy_or_pack_range(c, beg, end);
		if (c->su) {                                        /* a streamed plan: this chunk is part of a piece of the upload; the piece's last chunk issues what is ready */
			int k = 0;
			while (k + 1 < c->su->np && c->su->pfirst[k + 1] <= beg) ++k;
			if (__sync_sub_and_fetch(&c->su->left[k], 1) == 0) stream_issue(c->su, k);
		}
	}
	return KSW2AMD_OK;
}
/* 1 = the pool's threads did the copy.  Only for plans of 32 MB and more created outside the pool (a single-plan batch, a
 * caller's own ksw2amd_plan_create): config 5's 166 MB took 16 of the 22 ms of its plan creation on the calling thread */
int parallel_copy(copy_ctx_t *c, int n, size_t bytes)
{
	const int tpd = pool_threads_per_device();
	job_t j;
	int cbeg[POOL_MAXW + 2], k, i, nch;
	if (g_is_worker || tpd < 2 || n < 2 * tpd || bytes < ((size_t)32 << 20) || ENV(NO_PARCOPY)) return 0;
	nch = imin(tpd, POOL_MAXW);
	if (c->su) {                                           /* streamed plans: the upload's pieces are the work units, taken in order */
		nch = imin(c->su->np, POOL_MAXW);
		for (k = 0; k <= nch; ++k) cbeg[k] = c->su->pfirst[k];
		for (k = 0; k < nch; ++k) c->su->left[k] = 1;
	} else {
	for (k = 0, i = 0; k < nch; ++k) {                     /* equal byte ranges of the arena (the pairs lie in it in order) */
		const size_t edge = bytes / (size_t)nch * (size_t)k;
		while (i < n && (c->pairs[i].qlen <= 0 || c->pairs[i].tlen <= 0 || c->hp[i].qoff < edge)) ++i;
		cbeg[k] = k ? i : 0;
	}
	cbeg[nch] = n;
	}
	memset(&j, 0, sizeof(j));
	j.fn = copy_chunk; j.ctx = c; j.cbeg = cbeg; j.nchunks = nch; j.quiet = 1;
	j.ndev = 1;                                            /* the creating thread's device's workers */
	j.dev[0] = k2a_shim_get_device(); if (j.dev[0] < 0) j.dev[0] = 0;
	j.share = imin(nch, tpd);
	if (pool_run(&j)) return 0;
	return j.rc == 0;
}

/* the record assembly of a big plan on the pool's threads (plan_fetch_ex); 1 = done */
static int asm_chunk(void *ctx, int beg, int end, int share, pend_t *pd)
{
	(void)share; (void)pd;
	if (beg >= 0) assemble_range((asm_ctx_t*)ctx, beg, end);
	return KSW2AMD_OK;
}
int assemble_parallel(asm_ctx_t *c)
{
	const int tpd = pool_threads_per_device(), n = c->p->n;
	job_t j;
	int cbeg[POOL_MAXW + 2], k, nch;
	if (g_is_worker || tpd < 2) return 0;
	nch = imin(tpd, POOL_MAXW);
	for (k = 0; k <= nch; ++k) cbeg[k] = (int)((int64_t)n * k / nch);
	memset(&j, 0, sizeof(j));
	j.fn = asm_chunk; j.ctx = c; j.cbeg = cbeg; j.nchunks = nch; j.quiet = 1;
	j.ndev = 1;
	j.dev[0] = k2a_shim_get_device(); if (j.dev[0] < 0) j.dev[0] = 0;
	j.share = nch;
	if (pool_run(&j)) return 0;
	return 1;
}

/* Pairs whose device result cannot be used (needs_rerun) go through the ordinary gather path again -- as ONE batch: a batch of long
 * reads of which a tenth may drop would otherwise pay a plan, a launch and a fetch per pair (round 3: one by one).  The coalesced
 * single calls (one km per pair) keep the pair-by-pair form; they are single pairs to begin with. */
int rerun_pairs(ksw2amd_plan_t *p, int nrerun, void *km, ksw_extz_t *ez, ksw_extz_t **ezp, void **kmp)
{
	ksw2amd_pair_t *a;
	ksw_extz_t *zz;
	uint8_t *tmp = 0;
	size_t tmp_bytes = 0, at = 0;
	int i, k = 0, rc = KSW2AMD_OK;
	if (kmp || nrerun == 1) {
		for (i = 0; i < p->n && rc == KSW2AMD_OK; ++i)
			if (needs_rerun(p, i)) rc = pair_rerun(p, i, kmp ? kmp[i] : km, ezp ? ezp[i] : &ez[i]);
		return rc;
	}
	a = (ksw2amd_pair_t*)malloc(sizeof(*a) * (size_t)nrerun);
	zz = (ksw_extz_t*)malloc(sizeof(*zz) * (size_t)nrerun);
	if (!a || !zz) { free(a); free(zz); return fail(KSW2AMD_E_NOMEM, "plan_fetch: host allocation failed%s", 0); }
	for (i = 0; i < p->n && k < nrerun; ++i) {
		if (!needs_rerun(p, i)) continue;
		if (p->flat) a[k] = p->src_pairs[i];
		else {                                             /* from the staging copy and the resolved parameters */
			const K2aPair *d = &p->h_pairs[i];
			a[k].query = p->h_seq + d->qoff; a[k].target = p->h_seq + d->toff; a[k].qlen = d->qlen; a[k].tlen = d->tlen_full;
			a[k].w = d->w; a[k].zdrop = d->zdrop; a[k].end_bonus = p->scalar ? 0 : d->end_bonus; a[k].flag = p->h_flag[i] & ~F_SCALAR_CONTRACT;
		}
		tmp_bytes += (size_t)a[k].qlen + (size_t)a[k].tlen;
		zz[k] = ezp ? *ezp[i] : ez[i];
		++k;
	}
	nrerun = k;
	if (plan_wire4(p)) {                                   /* the staging copy holds two codes per byte: one per byte for the re-run */
		tmp = (uint8_t*)malloc(tmp_bytes + 1);
		if (!tmp) { free(a); free(zz); return fail(KSW2AMD_E_NOMEM, "plan_fetch: host allocation failed%s", 0); }
		for (i = 0, k = 0; i < p->n && k < nrerun; ++i) {
			if (!needs_rerun(p, i)) continue;
			wire4_pair(p, i, tmp + at);
			a[k].query = tmp + at; a[k].target = tmp + at + a[k].qlen;
			at += (size_t)a[k].qlen + (size_t)a[k].tlen;
			++k;
		}
	} else
	if (p->flat_device) {                                  /* the sequences are in device memory only: bring these pairs' back */
		tmp = (uint8_t*)malloc(tmp_bytes + 1);
		if (!tmp) { free(a); free(zz); return fail(KSW2AMD_E_NOMEM, "plan_fetch: host allocation failed%s", 0); }
		for (k = 0; k < nrerun && rc == KSW2AMD_OK; ++k) {
			if (k2a_shim_d2h(tmp + at, a[k].query, (size_t)a[k].qlen, p->stream) || k2a_shim_d2h(tmp + at + a[k].qlen, a[k].target, (size_t)a[k].tlen, p->stream))
				rc = fail(KSW2AMD_E_NODEVICE, "plan_fetch: %s", k2a_shim_last_error());
			a[k].query = tmp + at; a[k].target = tmp + at + a[k].qlen;
			at += (size_t)a[k].qlen + (size_t)a[k].tlen;
		}
		if (k2a_shim_stream_sync(p->stream) && rc == KSW2AMD_OK) rc = fail(KSW2AMD_E_NODEVICE, "plan_fetch: %s", k2a_shim_last_error());
	}
	if (rc == KSW2AMD_OK) {
		++g_no_defer;
		rc = run_serial(p->dual, p->scalar, km, &p->src_sc, nrerun, a, zz, 1, 0, 0);
		--g_no_defer;
		__sync_fetch_and_add(&g_reruns, nrerun);
	}
	for (i = 0, k = 0; i < p->n && k < nrerun; ++i)
		if (needs_rerun(p, i)) { if (ezp) *ezp[i] = zz[k]; else ez[i] = zz[k]; ++k; }      /* (CIGAR buffers may have moved: always copy back) */
	free(a); free(zz); free(tmp);
	return rc;
}

/* The gather of a streamed plan on the pool's threads, asynchronously: chunk k = piece k of the upload, copied (not scanned) into the
 * pinned arena and issued by whichever worker closes the gap (copy_chunk -> stream_issue).  More threads than a batch's chunks get:
 * the copy is memory-bound and a thread moves 4-5 GB/s (config 2: 67 MB in 2.8 ms on six).  0 = started; the plan owns the job until
 * gather_wait(), which also records the event that marks the end of the plan's upload. */
#define K2A_GATHER_SUB 8                /* copy chunks per upload piece: the first piece is complete after an eighth of a piece's copy time, not a whole one */
struct gather_s { job_t j; copy_ctx_t cc; int cbeg[K2A_MAXPIECES * K2A_GATHER_SUB + 2]; };
static int gather_start_ex(ksw2amd_plan_t *p, stream_up_t *su, const ksw2amd_pair_t *pairs, int n, int uniform, int cls0, int flag0)
{
	const int tpd = pool_threads_per_device();
	struct gather_s *g;
	int k, nth;
	(void)n;
	if (g_is_worker || tpd < 2 || ENV(NO_PARCOPY)) return -1;
	g = (struct gather_s*)calloc(1, sizeof(*g));
	if (!g) return -1;
	nth = (int)(p->seq_bytes >> 22);                       /* a thread per 4 MB, between the batch workers' count and 24 */
	nth = imax(tpd, imin(nth, 24)); nth = imin(nth, su->np * K2A_GATHER_SUB);
	g->cc.h_seq = p->h_seq; g->cc.hp = p->h_pairs; g->cc.pairs = pairs; g->cc.wild = 0; g->cc.su = su;
	g->cc.uni_plan = uniform ? p : 0; g->cc.uni_cls = cls0; g->cc.uni_flag = flag0;
	{	/* the workers take the chunks in order (job_t.next), so the pieces complete roughly in order, the first one early */
		int nc = 0, x;
		for (k = 0; k < su->np; ++k) {
			const int lo = su->pfirst[k], hi = su->pfirst[k + 1], sub = imax(1, imin(K2A_GATHER_SUB, hi - lo));
			su->left[k] = sub;
			for (x = 0; x < sub; ++x) g->cbeg[nc++] = lo + (int)((int64_t)(hi - lo) * x / sub);
		}
		g->cbeg[nc] = su->pfirst[su->np];
		g->j.nchunks = nc;
	}
	g->j.fn = copy_chunk; g->j.ctx = &g->cc; g->j.cbeg = g->cbeg; g->j.quiet = 1;
	g->j.ndev = 1;
	g->j.dev[0] = k2a_shim_get_device(); if (g->j.dev[0] < 0) g->j.dev[0] = 0;
	g->j.share = nth;
	if (pool_start(&g->j)) { free(g); return -1; }
	p->gather = g;
	return 0;
}
int gather_start(ksw2amd_plan_t *p, stream_up_t *su, const ksw2amd_pair_t *pairs, int n) { return gather_start_ex(p, su, pairs, n, 0, 0, 0); }
int gather_start_uniform(ksw2amd_plan_t *p, stream_up_t *su, const ksw2amd_pair_t *pairs, int n, int cls0, int flag0) { return gather_start_ex(p, su, pairs, n, 1, cls0, flag0); }
int gather_wait(ksw2amd_plan_t *p)
{
	struct gather_s *g = p->gather;
	int rc = 0;
	if (!g) return 0;
	pool_wait(&g->j);
	p->gather = 0;
	if (trace_level() >= 2 && p->up_state) fprintf(stderr, "[ksw2_amd]   gather of %zu MB on %d threads: first piece issued +%.2f ms, last piece +%.2f ms, waited for at +%.2f ms of the plan\n", p->seq_bytes >> 20, g->j.share, p->up_state->t_first - p->up_state->t0, p->up_state->t_last - p->up_state->t0, now_ms() - p->up_state->t0);
	if (g->j.rc || (p->up_state && p->up_state->rc)) rc = -1;
	if (p->up_state) {                                     /* everything has been issued by now: mark the end of the upload */
		stream_issue(p->up_state, -1);
		if (p->up_ev && k2a_shim_event_record(p->up_ev, p->up_state->up)) rc = -1;
	}
	free(g);
	return rc;
}

int pool_min_pairs(void)
{
	const char *e = ENV(POOL_MIN);           /* tests: pool batches of this many pairs or more, whatever their size */
	return e && atoi(e) > 0 ? atoi(e) : 0;
}

/* critical path of a fill (seconds) from which a batch counts as "long alignments" in plan_chunks; KSW2AMD_LONG_MS overrides */
static double long_path_s(void)
{
	const char *e = ENV(LONG_MS);
	return e && atof(e) > 0 ? atof(e) * 1e-3 : 0.010;
}

/* pairs that put one wavefront on every SIMD, for a batch of this pair's shape: the first packed geometry that holds the band
 * (as the classification in plan_create_ex picks it) runs 2 x 64 / G alignments per wavefront.  0 = no resident geometry (the
 * generation-serial classes) or no device figure: such batches keep the cost-balanced chunks. */
int unit_pairs(const ksw2amd_pair_t *a)
{
	const int simds = k2a_shim_simd_count(), tl = imax(a->tlen, 1), mx = imax(a->qlen, tl);
	const int w = (a->w < 0 || a->w > mx) ? mx : a->w;
	int pc;
	if (simds <= 0) return 0;
	for (pc = (a->flag & KSW_EZ_SCORE_ONLY) ? 0 : 1; pc < K2A_PKCFG_MP; ++pc)
		if (geom_fits(k2a_pkcfg_G[pc], k2a_pkcfg_C[pc], tl, w)) return simds * 2 * (64 / k2a_pkcfg_G[pc]);
	return 0;
}

/* Batches whose pairs all have one shape (the configurations of BASELINE.json; reads trimmed to one length): every wavefront of
 * a fill lasts equally long, so a kernel takes as long as the SIMD that holds the most of them, and a chunk whose wavefronts do
 * not tile the SIMDs wastes the difference (config 3, 16 384 pairs: 8 chunks of 2 048 pairs = 1 024 wavefronts 759 GCUPS end to
 * end, 6 chunks of 2 731 485, 12 of 1 365 416; 10 k x 10 k with CIGAR, 4 096 pairs: 2 chunks 1 286, 3 chunks 585, one plan 1 075;
 * config 2: 8 chunks of half a fill 1 118, 6 chunks 933; profiles/r2_chunk_units.txt).  `unit` = pairs of one wavefront per SIMD
 * (unit_pairs).  Chunks are 2^j x the smallest useful size -- half a unit for short score-only reads, one unit with CIGARs, two
 * units for long score-only reads (the 10 k headline: 12 chunks of 4 096 pairs 3 276, 24 of 2 048 3 211, 6 of 8 192 3 095) -- with
 * at most two chunks per worker.  Returns the chunk count (0 = one plan on the calling thread) and the chunk size. */
int uniform_chunks(int n, int unit, double bytes, double cells, int workers, int ndev, int with_cigar, double path_steps, int *chunk_pairs)
{
	const char *e1 = ENV(CHUNK_MB);
	const double cap_b = (e1 && atof(e1) > 0 ? atof(e1) : 128.0) * 1048576.0;
	const double path_s = path_steps * (with_cigar ? 4.5e-6 : 2.5e-6);
	double cu = with_cigar ? 1.0 : path_s >= long_path_s() ? 2.0 : 0.5, kmax = 2.0 * workers;
	const double units = (double)n / unit;
	int k;
	if (workers <= 0 || n < 512 || (bytes < 4.0 * 1048576.0 && cells < 2e9)) return 0;      /* as plan_chunks: too small to be worth the hand-off */
	if (ndev > 1) kmax = 3.0 * workers;
	if (bytes / cap_b > kmax) kmax = bytes / cap_b;
	while (units / cu > kmax) cu *= 2;
	*chunk_pairs = (int)(cu * unit);
	k = (n + *chunk_pairs - 1) / *chunk_pairs;
	/* Long score-only reads, six chunks and more: the first two chunks at this size (two wavefronts per SIMD each: the device starts
	 * after one such chunk's packing and upload), the rest twice as big -- a kernel that fills the device by itself loses nothing
	 * when it is the last one running, where a two-per-SIMD kernel alone runs at two thirds of the rate (MI355X, 10 k headline end to
	 * end: 4 290 -> 4 370 GCUPS through the pointer entry, 4 330 -> 4 500 through the flat entry; KSW2AMD_GROW=0: all chunks equal) */
	if (!with_cigar && cu == 2.0 && path_s >= long_path_s() && k >= 6 && env_flag(ENV(GROW), 1)) {
		*chunk_pairs = -*chunk_pairs;
		k = 2 + (k - 2 + 1) / 2;
	}
	return k < 2 ? 0 : k;
}

static int plan_chunks(int n, double bytes, double cells, int workers, int ndev, int with_cigar, double path_steps)
{
	const char *e1 = ENV(CHUNK_MB), *e2 = ENV(CHUNK_GCELLS);
	const double cap_b = (e1 && atof(e1) > 0 ? atof(e1) : 128.0) * 1048576.0, cap_c = (e2 && atof(e2) > 0 ? atof(e2) : 40.0) * 1e9;
	const int forced = pool_min_pairs(), min_chunk = forced ? imax(forced / 4, 1) : 256;
	double k;
	if (workers <= 0) return 0;
	if (forced) { if (n < forced) return 0; k = workers; }
	else {
		if (n < 512 || (bytes < 4.0 * 1048576.0 && cells < 2e9)) return 0;
		/* score-only batches: one chunk per worker (packing in parallel; fewer, larger kernels).  With CIGARs the download and the
		 * ksw_extz_t assembly of a chunk cost as much as its kernels: two chunks per worker, double-buffered (ext_chunk), so
		 * that while a worker fetches, kernels keep the device busy (config 3: one chunk per worker left it idle a third of the call) */
		k = bytes / (2.0 * 1048576.0);
		if (k > (with_cigar ? 2 : 1) * workers) k = (with_cigar ? 2 : 1) * workers;
	}
	if (bytes / cap_b > k) k = bytes / cap_b;
	/* the cell cap never cuts a chunk below 4096 pairs: 2048 packed wavefronts, two per SIMD -- kernels of fewer wavefronts leave
	 * SIMDs idle (config 4 at 40 G cells per chunk was 146 pairs per kernel); what really limits such batches is traceback
	 * memory, and ext_chunk splits by that */
	{
		double kc = cells / cap_c;
		if (kc > (double)n / 4096.0) kc = (double)n / 4096.0;
		if (kc > k) k = kc;
	}
	/* A fill kernel lasts at least its longest alignment's step count (2.5 / 4.5 us per step without / with traceback, however
	 * few wavefronts it has), and kernels of long alignments do not really overlap: each has enough workgroups to hold every
	 * SIMD, and its longest tasks are dispatched first (r2_pipeline_traces.txt: 18 chunks of config 5 = 18 x the 250 ms of a
	 * 20 k read instead of 190 ms for everything).  So for long alignments the chunk count is what the batch's cells pay for:
	 * nchunks x path <= cells / rate.  Short alignments (path below 10 ms) keep one (two with CIGARs) chunk per worker. */
	{
		const double total_s = cells / (with_cigar ? 1e12 : 2e12), path_s = path_steps * (with_cigar ? 4.5e-6 : 2.5e-6);
		if (path_s >= long_path_s()) {
			if (with_cigar) { if (k > total_s / path_s) k = total_s / path_s; }
			else {
				/* score only: a kernel of exactly two wavefronts per SIMD (4096 pairs, two per wavefront) has no tail, and such
				 * kernels follow each other without a gap -- 12 chunks of 4096 pairs: 3 050 GCUPS, 9 of 5461: 2 675 */
				const double units = (double)(n / 4096);
				k = (double)(int)(k + 0.999);              /* 11.99 chunks by the cell cap are 12, not 11 of 4 468 pairs */
				if (k > units) k = units;
			}
		}
		else if (k > (with_cigar ? 2 : 1) * workers && bytes / cap_b <= (with_cigar ? 2 : 1) * workers) k = (with_cigar ? 2 : 1) * workers;
	}
	if (ndev > 1 && k < 3 * workers) k = 3 * workers;      /* several devices: finer grains balance them */
	if (k > n / min_chunk) k = n / min_chunk;
	return k < 2 ? 0 : (int)(k + 0.999);
}

typedef struct { int dual, scalar; void *km; const ksw2amd_scoring_t *sc; const ksw2amd_pair_t *pairs; ksw_extz_t *ez; const flat_src_t *flat; } ext_ctx_t;
double now_ms(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6; }
int64_t now_ns(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return (int64_t)ts.tv_sec * 1000000000 + ts.tv_nsec; }
int trace_on(void) { return ENV(TRACE) != 0; }

static double g_batch_t0;                  /* KSW2AMD_TRACE: start of the current pooled batch, for the timeline */
static int ext_finish(ext_ctx_t *c, pend_t *pd)
{
	int rc = KSW2AMD_OK;
	if (pd->p) {
		const double t0 = now_ms();
		rc = ksw2amd_plan_fetch(pd->p, c->km, c->ez + pd->beg);
		if (trace_on()) fprintf(stderr, "[ksw2_amd] chunk @%d n=%d: wait+fetch %.2f ms (from +%.2f to +%.2f ms of the batch)\n", pd->beg, pd->p->n, now_ms() - t0, t0 - g_batch_t0, now_ms() - g_batch_t0);
		ksw2amd_plan_destroy(pd->p);
		pd->p = 0;
	}
	return rc;
}

/* One chunk on a pool worker: pack, upload, run, fetch; the workers are what overlaps the phases of different chunks.  (A worker
 * that queued chunk k + 1 before it waited for chunk k -- rounds 2 and 3, behind a switch -- lost on every configuration once the
 * chunks tiled the SIMDs, and degraded over many batches: removed in round 4; tried once more behind the in-order upload stream
 * for small chunks: config 2 1 061 -> 973, 10 k with CIGAR 1 318 -> 1 186, the others unchanged.)  A chunk that does not fit one plan (traceback
 * memory) takes the serial path. */
static int ext_chunk(void *ctx_, int beg, int end, int share, pend_t *pd)
{
	ext_ctx_t *c = (ext_ctx_t*)ctx_;
	size_t bytes = 0, free_b = 0, total_b = 0, budget = (size_t)1 << 30;
	const char *env = ENV(MAX_BYTES);
	ksw2amd_plan_t *p;
	double t0, t1;
	int i, rc, rc2;
	if (beg < 0) return ext_finish(c, pd);
	for (i = beg; i < end; ++i) bytes += pair_device_bytes(c->dual, &c->pairs[i]);
	if (env && atoll(env) > 0) budget = (size_t)atoll(env);
	else if (bytes > ((size_t)256 << 20)) {
		if (k2a_shim_mem_info(&free_b, &total_b)) return fail(KSW2AMD_E_NODEVICE, "mem_info: %s", k2a_shim_last_error());
		/* what is free now, plus what this worker's cache will hand back */
		budget = device_budget(free_b, total_b, share);
	}
	if (bytes > budget) {
		rc = ext_finish(c, pd);
		return rc ? rc : run_serial(c->dual, c->scalar, c->km, c->sc, end - beg, c->pairs + beg, c->ez + beg, share, c->flat, 0);
	}
	t0 = now_ms();
	p = plan_create_ex(c->dual, c->scalar, c->sc, end - beg, c->pairs + beg, c->flat, 0);
	if (!p) {                                       /* out of device memory: go serial (plans sized to what is free) */
		if (!strstr(g_err, "alloc")) return strstr(g_err, "device") ? KSW2AMD_E_NODEVICE : KSW2AMD_E_PARAM;
		rc = ext_finish(c, pd);
		return rc ? rc : run_serial(c->dual, c->scalar, c->km, c->sc, end - beg, c->pairs + beg, c->ez + beg, share, c->flat, 0);
	}
	t1 = now_ms();
	rc = ksw2amd_plan_run(p, thread_stream());
	if (trace_on()) fprintf(stderr, "[ksw2_amd] chunk @%d n=%d: pack+upload %.2f ms, launch %.2f ms, %zu device bytes (from +%.2f ms of the batch)\n", beg, end - beg, t1 - t0, now_ms() - t1, bytes, t0 - g_batch_t0);
	rc2 = ext_finish(c, pd);
	if (rc) { ksw2amd_plan_destroy(p); return rc; }
	pd->p = p; pd->beg = beg;
	rc = ext_finish(c, pd); if (!rc2) rc2 = rc;                                        /* finish this chunk before taking the next */
	return rc2;
}

/* devices of a pooled job: ksw2amd_set_devices() or the calling thread's current device */
static void job_devices(job_t *j)
{
	int i;
	if (g_ndev_set > 0) { j->ndev = g_ndev_set; for (i = 0; i < g_ndev_set; ++i) j->dev[i] = g_dev_set[i]; }
	else { j->ndev = 1; j->dev[0] = k2a_shim_get_device(); if (j->dev[0] < 0) j->dev[0] = 0; }
}

/* run the chunks of a batch on the pool; 1 = done (rc in *rc), 0 = the caller must run the batch inline */
int run_pooled(chunk_fn fn, void *ctx, int n, const double *cost, double total, int nchunks, int chunk_pairs, int *rc)
{
	job_t j;
	const int tpd = pool_threads_per_device(), workers = tpd * (g_ndev_set > 0 ? g_ndev_set : 1);
	int *cbeg = (int*)malloc(sizeof(int) * ((size_t)nchunks + 2 * (size_t)workers + 2));
	if (!cbeg) return 0;
	memset(&j, 0, sizeof(j));
	j.fn = fn; j.ctx = ctx; j.cbeg = cbeg;
	j.nchunks = make_chunks(n, cost, total, nchunks, workers, chunk_pairs, cbeg);
	job_devices(&j);
	j.share = imax(1, imin(tpd, (j.nchunks + j.ndev - 1) / j.ndev));      /* plans alive per device at a time: the memory budget's divisor */
	if (trace_on()) g_batch_t0 = now_ms();
	if (pool_run(&j)) { free(cbeg); return 0; }
	if (trace_on()) fprintf(stderr, "[ksw2_amd] pooled batch n=%d: %d chunks on %d workers, %.2f ms\n", n, j.nchunks, j.share * j.ndev, now_ms() - g_batch_t0);
	free(cbeg);
	if (j.rc) snprintf(g_err, sizeof(g_err), "%s", j.err);
	*rc = j.rc;
	return 1;
}

static int run_batch_(int dual, int scalar, void *km, const ksw2amd_scoring_t *sc, int n, const ksw2amd_pair_t *pairs, ksw_extz_t *ez, const flat_src_t *flat);
static int run_batch(int dual, int scalar, void *km, const ksw2amd_scoring_t *sc, int n, const ksw2amd_pair_t *pairs, ksw_extz_t *ez, const flat_src_t *flat)
{
	/* Batches with CIGARs: eight workers per device instead of six.  A worker of such a batch spends most of a chunk's time waiting
	 * for kernels and then bringing CIGARs back, the one-shape rule cuts 2^j chunks (config 3: eight), and eight chunks on six workers
	 * are two rounds whose second runs on a third of the device (round 5, same box: config 3 717 -> 765-791 GCUPS end to end, config 5
	 * 926 -> 1 028, 10 k with CIGAR unchanged; profiles/r5_e2e_threads_ab.txt).  Score-only batches keep six: their workers pack and
	 * upload most of the time and more of them only share the host's memory bandwidth (config 2's chunks: 935 -> 580). */
	int rc;
	g_job_threads = n > 0 && pairs && !(pairs[0].flag & KSW_EZ_SCORE_ONLY) ? 8 : 0;
	rc = run_batch_(dual, scalar, km, sc, n, pairs, ez, flat);
	g_job_threads = 0;
	return rc;
}
/* plans created on the calling thread run on ITS current device: only where that is the device the caller selected
 * (ksw2amd_set_devices(1, &d) with another d: the pool's workers, which switch to d, take the batch) */
static int calling_thread_on_set_device(void)
{
	return g_ndev_set == 0 || (g_ndev_set == 1 && g_dev_set[0] == k2a_shim_get_device());
}
static int run_batch_(int dual, int scalar, void *km, const ksw2amd_scoring_t *sc, int n, const ksw2amd_pair_t *pairs, ksw_extz_t *ez, const flat_src_t *flat)
{
	const int tpd = pool_threads_per_device();
	if (n <= 0) return KSW2AMD_OK;
	if (k2a_shim_device_count() <= 0) return fail(KSW2AMD_E_NODEVICE, "no usable %s device", k2a_shim_backend());
	if (n >= (pool_min_pairs() ? pool_min_pairs() : 512) && tpd > 0 && !g_is_worker) {
		const int workers = tpd * (g_ndev_set > 0 ? g_ndev_set : 1);
		double *cost, bytes = 0, cells = 0, total = 0, path = 0, dev_bytes = 0;
		int i, nchunks, rc = 0, uniform = 1, chunk_pairs = 0, same = 1;      /* same: every pair equals pair 0 in every parameter, none is empty (plan_create_uniform) */
		/* one shape for the whole batch (the BASELINE configurations, reads trimmed to one length)?  Then the sums below are n x
		 * the first pair's terms and the per-pair costs are never looked at (uniform_chunks cuts at fixed sizes): this loop was
		 * 1.5-2 ms of the calling thread's time on config 2's 65 536 pairs, in front of a 1.4 ms kernel */
		for (i = 1; i < n; ++i) {
			if (pairs[i].qlen != pairs[0].qlen || pairs[i].tlen != pairs[0].tlen || pairs[i].w != pairs[0].w || ((pairs[i].flag ^ pairs[0].flag) & KSW_EZ_SCORE_ONLY)) { uniform = 0; break; }
			if (pairs[i].zdrop != pairs[0].zdrop || pairs[i].end_bonus != pairs[0].end_bonus || pairs[i].flag != pairs[0].flag || !pairs[i].query || !pairs[i].target) same = 0;
		}
		if (!uniform || pairs[0].qlen <= 0 || pairs[0].tlen <= 0) same = 0;
		if (uniform) dev_bytes = (double)n * (double)pair_device_bytes(dual, &pairs[0]);
		else for (i = 0; i < n; ++i) dev_bytes += (double)pair_device_bytes(dual, &pairs[i]);
		/* One-shape score-only batches that fit the device: ONE streamed plan (section "streamed plans") instead of chunks -- a
		 * single launch over the whole batch, started under the upload, whose wavefronts wait for their pieces, longest task first, at full
		 * occupancy.  Where it pays is where the kernels are long against the host's per-pair work: the 10 k headline (MI355X, round 4,
		 * same box: 4 060 against 3 800 GCUPS through the pointer entry, 4 510 against 4 270 through the flat one), not 512-base reads,
		 * whose plan creation on one thread costs what six workers' chunks cost together (config 2: 3.9 against 3.4-4.0 ms) -- so by
		 * default batches of at least 1 M cells per pair.  KSW2AMD_STREAM=1: every one-shape score-only batch; =0: chunks. */
		if (uniform && (pairs[0].flag & KSW_EZ_SCORE_ONLY) && stream_env() != 0 && calling_thread_on_set_device() && !pool_min_pairs() &&
		    ((double)n * ((double)imax(pairs[0].qlen, 0) + imax(pairs[0].tlen, 0)) >= 4.0 * 1048576.0 || (same && ENV(UNIFORM) && atoi(ENV(UNIFORM)) == 1))) {      /* (KSW2AMD_UNIFORM=1: tests, small batches) */
			const int mx0 = imax(pairs[0].qlen, pairs[0].tlen);
			const int64_t c0 = pairs[0].qlen > 0 && pairs[0].tlen > 0 ? band_cells(pairs[0].qlen, pairs[0].tlen, (pairs[0].w < 0 || pairs[0].w > mx0) ? mx0 : pairs[0].w) : 0;
			size_t free_b = 0, total_b = 0;
			const int fits = dev_bytes <= 256e6 || (k2a_shim_mem_info(&free_b, &total_b) == 0 && dev_bytes <= (double)device_budget(free_b, total_b, 1));
			/* one shape AND one set of parameters, from ordinary host memory: a uniform plan (ksw2_host_plan.c "uniform batches") -- no
			 * per-pair work on this thread, the records built on the device, ONE streamed launch -- whatever the reads' length (round 5:
			 * config 2, whose plan creation used to cost more than its kernel, goes this way too) */
			/* (a flat batch in HOST memory goes the same way: its pairs point into the arena, and the gather that packs two codes per byte
			 * moves half the bytes over the link that uploading the arena as it lies would -- config 2: 1 870 against 900 GCUPS) */
			if (same && fits && (!flat || !flat->on_device)) {
				const double tu0 = now_ms();
				ksw2amd_plan_t *up = plan_create_uniform_entry(dual, scalar, sc, n, pairs);
				if (up) {
					const double tu1 = now_ms();
					double tu2;
					rc = ksw2amd_plan_run(up, g_plan_stream ? g_plan_stream : thread_stream());
					tu2 = now_ms();
					if (rc == KSW2AMD_OK) rc = ksw2amd_plan_fetch(up, km, ez);
					phase_add(tu1 - tu0, tu2 - tu1, now_ms() - tu2);
					{
						const int refit = up->up_state && up->up_state->wire_bad;      /* a residue code above 15: not through the 4-bit wire format */
						ksw2amd_plan_destroy(up);
						if (!refit) return rc;
						for (i = 0; i < n; ++i) ez_reset(&ez[i]);
					}
				}
				else if (g_err[0]) return strstr(g_err, "alloc") ? KSW2AMD_E_NOMEM : strstr(g_err, "device") || strstr(g_err, "upload") ? KSW2AMD_E_NODEVICE : KSW2AMD_E_PARAM;
			}
			if ((stream_env() == 1 || c0 >= stream_min_cells()) && fits)
				return run_serial(dual, scalar, km, sc, n, pairs, ez, 1, flat, 1);
		}
		cost = (double*)malloc(sizeof(double) * (size_t)n);
		if (cost) {
			if (dev_bytes > 64e9 && !pool_min_pairs()) {
				/* traceback memory is what splits this batch: one plan at a time with the whole device, not a slice per worker */
				size_t free_b = 0, total_b = 0;
				if (k2a_shim_mem_info(&free_b, &total_b) == 0 && dev_bytes > 0.5 * (double)total_b && calling_thread_on_set_device()) { free(cost); return run_serial(dual, scalar, km, sc, n, pairs, ez, 1, flat, 0); }
			}
			for (i = 0; i < (uniform ? 1 : n); ++i) {
				const int ql = imax(pairs[i].qlen, 0), tl = imax(pairs[i].tlen, 0), mx = imax(ql, tl);
				const double b = (double)ql + tl, c = ql && tl ? (double)band_cells(ql, tl, (pairs[i].w < 0 || pairs[i].w > mx) ? mx : pairs[i].w) : 0;
				bytes += b; cells += c;
				{	/* steps of the pair's fill: columns + strips; wide bands on long targets run as generations of 1024 rows, four at a time */
					const int wq = (pairs[i].w < 0 || pairs[i].w > mx) ? mx : pairs[i].w;
					const double st = (wq > 1040 && tl > 2048) ? (double)((tl + 4095) / 4096) * (ql + 64) * 1.4 : (double)ql + tl / 8.0;
					if (st > path) path = st;
				}
				cost[i] = 1.0 + c + 64.0 * b;              /* a byte costs the host about as much as 64 cells cost the device */
				total += cost[i];
			}
			if (uniform) { bytes *= n; cells *= n; for (i = 1; i < n; ++i) cost[i] = cost[0]; total = cost[0] * n; }
			{
				const int unit = uniform && !pool_min_pairs() ? unit_pairs(&pairs[0]) : 0;
				if (unit > 0) nchunks = uniform_chunks(n, unit, bytes, cells, workers, g_ndev_set, !(pairs[0].flag & KSW_EZ_SCORE_ONLY), path, &chunk_pairs);
				else nchunks = plan_chunks(n, bytes, cells, workers, g_ndev_set, !(pairs[0].flag & KSW_EZ_SCORE_ONLY), path);
			}
			if (nchunks >= 2) {
				ext_ctx_t ctx;
				ctx.dual = dual; ctx.scalar = scalar; ctx.km = km; ctx.sc = sc; ctx.pairs = pairs; ctx.ez = ez; ctx.flat = flat;
				if (run_pooled(ext_chunk, &ctx, n, cost, total, nchunks, chunk_pairs, &rc)) { free(cost); return rc; }
			}
			free(cost);
		}
	}
	return run_serial(dual, scalar, km, sc, n, pairs, ez, 1, flat, 0);
}


/* pairs that ask for the SSE kernels' own results (wants_ssec) run through the SSE-compatible plans, the others through the
 * exact-contract kernels; a mixed batch is split and its results put back in place */
static int route_batch(int dual, void *km, const ksw2amd_scoring_t *sc, int n, const ksw2amd_pair_t *pairs, ksw_extz_t *ez)
{
	int i, nc = 0, rc = KSW2AMD_OK, part;
	for (i = 0; i < n; ++i) nc += wants_ssec(pairs[i].flag);
	if (nc == 0) return run_batch(dual, 0, km, sc, n, pairs, ez, 0);
	if (nc == n) return ssec_run(dual, km, sc, n, pairs, ez);
	for (part = 0; part < 2 && rc == KSW2AMD_OK; ++part) {
		const int cnt = part ? nc : n - nc;
		ksw2amd_pair_t *pp = (ksw2amd_pair_t*)malloc(sizeof(*pp) * (size_t)cnt);
		ksw_extz_t *zz = (ksw_extz_t*)malloc(sizeof(*zz) * (size_t)cnt);
		int k = 0;
		if (!pp || !zz) { free(pp); free(zz); return fail(KSW2AMD_E_NOMEM, "batch: host allocation failed%s", 0); }
		for (i = 0; i < n; ++i) if (wants_ssec(pairs[i].flag) == part) { pp[k] = pairs[i]; zz[k] = ez[i]; ++k; }
		rc = part ? ssec_run(dual, km, sc, cnt, pp, zz) : run_batch(dual, 0, km, sc, cnt, pp, zz, 0);
		for (i = 0, k = 0; i < n; ++i) if (wants_ssec(pairs[i].flag) == part) ez[i] = zz[k++];      /* CIGAR buffers may have moved: always copy back */
		free(pp); free(zz);
	}
	return rc;
}

int ksw2amd_extz_batch(void *km, const ksw2amd_scoring_t *sc, int n, const ksw2amd_pair_t *pairs, ksw_extz_t *ez)
{
	return route_batch(0, km, sc, n, pairs, ez);
}

int ksw2amd_extd_batch(void *km, const ksw2amd_scoring_t *sc, int n, const ksw2amd_pair_t *pairs, ksw_extz_t *ez)
{
	return route_batch(1, km, sc, n, pairs, ez);
}

/* ---------------------------------------------------------------- flat batches: one arena + offsets (include/ksw2_amd.h) */

static ksw2amd_pair_t *flat_pairs(int n, const ksw2amd_flat_t *in)
{
	ksw2amd_pair_t *pp;
	int i;
	if (n < 0 || !in || (n > 0 && (!in->base || !in->qoff || !in->toff || !in->qlen || !in->tlen))) { fail(KSW2AMD_E_PARAM, "flat batch: bad arguments%s", 0); return 0; }
	pp = (ksw2amd_pair_t*)malloc(sizeof(*pp) * ((size_t)n + 1));
	if (!pp) { fail(KSW2AMD_E_NOMEM, "flat batch: host allocation failed%s", 0); return 0; }
	for (i = 0; i < n; ++i) {
		pp[i].query = in->base + in->qoff[i]; pp[i].target = in->base + in->toff[i];
		pp[i].qlen = in->qlen[i]; pp[i].tlen = in->tlen[i];
		pp[i].w = in->w ? in->w[i] : in->w_all; pp[i].zdrop = in->zdrop ? in->zdrop[i] : in->zdrop_all;
		pp[i].end_bonus = in->end_bonus ? in->end_bonus[i] : in->end_bonus_all; pp[i].flag = in->flag ? in->flag[i] : in->flag_all;
	}
	return pp;
}

static int flat_batch(int dual, void *km, const ksw2amd_scoring_t *sc, int n, const ksw2amd_flat_t *in, ksw_extz_t *ez)
{
	ksw2amd_pair_t *pp = flat_pairs(n, in);
	flat_src_t fs;
	int i, rc, plain = 1;
	if (!pp) return g_err[0] && strstr(g_err, "alloc") ? KSW2AMD_E_NOMEM : KSW2AMD_E_PARAM;
	fs.on_device = in->on_device != 0;
	for (i = 0; i < n && plain; ++i) plain = !wants_ssec(pp[i].flag);
	if (!plain) {                                      /* SSE-compatible pairs keep their own plans: the ordinary entry point sorts them out */
		uint8_t *host = 0;
		if (fs.on_device) {
			/* ... from host memory: bring the span of a device arena back first (a sharded run whose flags ask for the SSE kernels'
			 * results -- KSW2AMD_EZ_SSE_COMPAT, APPROX_MAX | APPROX_DROP, the process-wide switch -- reaches this on every receiving rank) */
			const uint8_t *lo = 0, *hi = 0;
			void *st = thread_stream();
			for (i = 0; i < n; ++i) {
				if (pp[i].qlen <= 0 || pp[i].tlen <= 0) continue;
				if (!lo || pp[i].query < lo) lo = pp[i].query;
				if (pp[i].target < lo) lo = pp[i].target;
				if (pp[i].query + pp[i].qlen > hi) hi = pp[i].query + pp[i].qlen;
				if (pp[i].target + pp[i].tlen > hi) hi = pp[i].target + pp[i].tlen;
			}
			if (lo) {
				host = (uint8_t*)malloc((size_t)(hi - lo) + 1);
				if (!host) { free(pp); return fail(KSW2AMD_E_NOMEM, "flat batch: host allocation failed%s", 0); }
				if (!st || k2a_shim_d2h(host, lo, (size_t)(hi - lo), st) || k2a_shim_stream_sync(st)) { free(host); free(pp); return fail(KSW2AMD_E_NODEVICE, "flat batch: %s", k2a_shim_last_error()); }
				for (i = 0; i < n; ++i) { pp[i].query = host + (pp[i].query - lo); pp[i].target = host + (pp[i].target - lo); }
			}
		}
		rc = route_batch(dual, km, sc, n, pp, ez);
		free(host);
	} else rc = run_batch(dual, 0, km, sc, n, pp, ez, &fs);
	free(pp);
	return rc;
}

int ksw2amd_extz_batch_flat(void *km, const ksw2amd_scoring_t *sc, int n, const ksw2amd_flat_t *in, ksw_extz_t *ez) { return flat_batch(0, km, sc, n, in, ez); }
int ksw2amd_extd_batch_flat(void *km, const ksw2amd_scoring_t *sc, int n, const ksw2amd_flat_t *in, ksw_extz_t *ez) { return flat_batch(1, km, sc, n, in, ez); }

ksw2amd_plan_t *ksw2amd_plan_create_flat(int dual, const ksw2amd_scoring_t *sc, int n, const ksw2amd_flat_t *in)
{
	ksw2amd_pair_t *pp = flat_pairs(n, in);
	ksw2amd_plan_t *p;
	flat_src_t fs;
	if (!pp) return 0;
	fs.on_device = in->on_device != 0;
	p = plan_create_ex(dual, 0, sc, n, pp, &fs, 0);
	free(pp);
	/* the batch entry points let the upload run on while they pack the next chunk; here it is complete on return.  The plan still
	 * BORROWS the arena until ksw2amd_plan_destroy (include/ksw2_amd.h): =/X rewrites and re-runs of a fetch read the sequences there */
	if (p && p->gather) gather_wait(p);
	if (p && p->up_ev) { k2a_shim_event_sync(p->up_ev); }
	return p;
}

/* device memory for callers that build a device-resident arena without linking the HIP runtime themselves */
void *ksw2amd_device_alloc(size_t bytes) { void *d = k2a_shim_malloc(bytes); if (!d) fail(KSW2AMD_E_NOMEM, "device_alloc: %s", k2a_shim_last_error()); return d; }
void ksw2amd_device_free(void *d) { k2a_shim_free(d); }
int ksw2amd_device_upload(void *dst, const void *src, size_t bytes)
{
	void *st = thread_stream();
	if (k2a_shim_h2d(dst, src, bytes, st) || k2a_shim_stream_sync(st)) return fail(KSW2AMD_E_NODEVICE, "device_upload: %s", k2a_shim_last_error());
	return KSW2AMD_OK;
}
int ksw2amd_device_download(void *dst, const void *src, size_t bytes)
{
	void *st = thread_stream();
	if (k2a_shim_d2h(dst, src, bytes, st) || k2a_shim_stream_sync(st)) return fail(KSW2AMD_E_NODEVICE, "device_download: %s", k2a_shim_last_error());
	return KSW2AMD_OK;
}

int ksw2amd_host_register(const void *p, size_t bytes)
{
	if (k2a_shim_host_register((void*)p, bytes)) return fail(KSW2AMD_E_NODEVICE, "host_register: %s", k2a_shim_last_error());
	return KSW2AMD_OK;
}
int ksw2amd_host_unregister(const void *p)
{
	if (k2a_shim_host_unregister((void*)p)) return fail(KSW2AMD_E_NODEVICE, "host_unregister: %s", k2a_shim_last_error());
	return KSW2AMD_OK;
}

