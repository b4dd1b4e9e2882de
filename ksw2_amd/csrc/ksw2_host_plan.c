/*
 * ksw2_host_plan.c -- C host side of libksw2_amd.so (the drop-in boundary, include/ksw2_amd.h), part 1 of 4: environment switches,
 * buffer cache, plans -- creation (classification = THE LAUNCH RULES, layout, upload), run, fetch, CIGAR assembly -- and serial batches.
 * (ksw2_host_pool.c: worker pool and batch entry points; ksw2_host_single.c: the ksw2-named calls; ksw2_host_ext.c: exts / extf / SSE mode.)
 *
 * What lives in the host objects: argument checks and early rejects of the "...2_sse" signatures
 * (ksw2_extz2_sse.c:56-82, ksw2_extd2_sse.c:75-100), the implicit match/mismatch/wildcard scoring
 * (ksw2_extz2_sse.c:66-69,125-140), packing of a batch into device arenas, the choice of kernel
 * geometry per pair, and the assembly of ksw_extz_t results including CIGAR buffer growth with the
 * reference's doubling rule (ksw2.h:113-123) through libc or the caller's kalloc (ksw2.h:103-111).
 * All DP work happens in the kernels behind ksw2_shim.h; there is no CPU alignment code in these files
 * (the opt-in scalar routine for tiny single calls, ksw2_host_single.c: small_pair, is the one exception, off by default).
 */
#include "ksw2_host_int.h"


__thread char g_err[512];
__thread int g_no_defer;
/* > 0: plan_create_ex is asked for the CLASS only -- the kernel family, geometry and form a batch of this many tasks of the given
 * pairs' shape would run as, by the very rules every plan is made by -- and returns before anything is copied, allocated on the
 * device or uploaded (plan_create_uniform) */
static __thread int g_probe_tasks;
size_t thread_cached_device_bytes(void);            /* set around the re-run of a pair the deferred arg-max kernels handed back as inexact */

const char *ksw2amd_last_error(void) { return g_err; }
const char *ksw2amd_backend(void) { return k2a_shim_backend(); }
int ksw2amd_device_count(void) { return k2a_shim_device_count(); }


void ksw2amd_release_cache(void);

/* ---------------------------------------------------------------- environment switches
 * Every KSW2AMD_* switch (A/B runs, tests, tuning) is read ONCE per process into a table -- the call paths, the coalesced
 * single-pair calls above all, never touch getenv().  ksw2amd_reload_env() reads them again (tests that flip a switch inside
 * one process; the Python binding calls it before every plan / batch). */
static const char *const g_env_name[ENV_COUNT] = {
#define X(n) "KSW2AMD_" #n,
	K2A_ENV_LIST
#undef X
};
__thread void *g_plan_stream;                 /* ... and this is the stream it will run on: its uploads go there too (in order: no event, no second queue) */
__thread int g_latency_plan;                  /* this thread is creating the plan of a single-pair call (or of a coalesced batch of them) */
const char *g_env[ENV_COUNT];
volatile int g_env_ready;
static int g_env_gen;                                /* bumped by every (re)load: function-local caches key on it */
static pthread_mutex_t g_env_mu = PTHREAD_MUTEX_INITIALIZER;
static int env_switch(const char *v);
/* KSW2AMD_BACKTRACE=1 (debugging on boxes without a debugger): the library's frames of a crash on stderr, then the default action */
static void crash_handler(int sig)
{
	void *fr[48];
	const int n = backtrace(fr, 48);
	static const char msg[] = "[ksw2_amd] fatal signal; frames:\n";
	if (write(2, msg, sizeof(msg) - 1) < 0) { }
	backtrace_symbols_fd(fr, n, 2);
	signal(sig, SIG_DFL);
	raise(sig);
}
void env_load(void)
{
	int i;
	pthread_mutex_lock(&g_env_mu);
	for (i = 0; i < ENV_COUNT; ++i) {
		const char *v = getenv(g_env_name[i]);
		/* values are kept for the life of the process: another thread may still hold the previous pointer */
		if (v && (!g_env[i] || strcmp(g_env[i], v))) g_env[i] = strdup(v);
		else if (!v) g_env[i] = 0;
	}
	if (g_env[ENV_BACKTRACE] && atoi(g_env[ENV_BACKTRACE])) { signal(SIGSEGV, crash_handler); signal(SIGBUS, crash_handler); signal(SIGABRT, crash_handler); }
	k2a_shim_set_option(K2A_OPT_LDSCODES, env_switch(g_env[ENV_LDSCODES]));
	k2a_shim_set_option(K2A_OPT_LDSROWS, env_switch(g_env[ENV_LDSROWS]));
	++g_env_gen;
	g_env_ready = 1;
	pthread_mutex_unlock(&g_env_mu);
}
void ksw2amd_reload_env(void) { env_load(); }

int ksw2amd_set_device(int device)
{
	release_thread_cache();                                 /* cached buffers belong to the previous device */
	if (k2a_shim_set_device(device)) return fail(KSW2AMD_E_NODEVICE, "set_device: %s", k2a_shim_last_error());
	return KSW2AMD_OK;
}

/* ---------------------------------------------------------------- buffer cache
 * Device allocations and page-locking cost milliseconds; a minimap2-style caller issues many batches (or single-pair
 * calls) from the same thread.  Each thread therefore keeps the buffers of its last plan (one per kind) and hands them to
 * the next plan when they are large enough.  ksw2amd_release_cache() returns them; switching device flushes them. */
static __thread struct { void *p; size_t cap; } g_cache[BUF_KINDS][CACHE_DEPTH];
static __thread void *g_ev_cache[3];
/* Every host thread uploads and (in the one-shot entry points) computes on a stream of its own, so concurrent callers --
 * a minimap2-style thread pool -- overlap their copies and kernels instead of queueing on the device's default stream. */
static __thread void *g_stream;
/* ... and uploads on a second one: a plan is packed and uploaded while the thread's previous plan still computes */
static __thread void *g_up_stream;
/* a worker thread that exits gives its cached buffers and stream back (pthread key destructor) */
static pthread_key_t g_exit_key;
static pthread_once_t g_exit_once = PTHREAD_ONCE_INIT;
/* ... unless the process is already on its way out: the HIP runtime must not be called while it unloads.  The atexit
 * hook waits for destructors that are in flight and turns the later ones into no-ops. */
static pthread_mutex_t g_exit_mu = PTHREAD_MUTEX_INITIALIZER;
static int g_exiting;
static void process_exit_cb(void) { pthread_mutex_lock(&g_exit_mu); g_exiting = 1; pthread_mutex_unlock(&g_exit_mu); }
static void thread_exit_cb(void *unused)
{
	(void)unused;
	pthread_mutex_lock(&g_exit_mu);
	if (!g_exiting) release_thread_cache();
	pthread_mutex_unlock(&g_exit_mu);
}
static void thread_exit_init(void) { pthread_key_create(&g_exit_key, thread_exit_cb); atexit(process_exit_cb); }
static void thread_owns_cache(void)
{
	pthread_once(&g_exit_once, thread_exit_init);
	if (!pthread_getspecific(g_exit_key)) pthread_setspecific(g_exit_key, (void*)1);
}
__thread int g_stream_high;              /* pool workers with an odd rank (KSW2AMD_WORKER_PRIO=1): their kernels' stream from the high-priority pool of hardware queues */
void *thread_stream(void) { if (!g_stream) { thread_owns_cache(); g_stream = g_stream_high ? k2a_shim_stream_create_high() : k2a_shim_stream_create(); } return g_stream; }
void *thread_upload_stream(void) { if (!g_up_stream) { thread_owns_cache(); g_up_stream = ENV(PLAIN_UP_STREAMS) ? k2a_shim_stream_create() : k2a_shim_stream_create_low(); } return g_up_stream; }      /* copies only: see shared_upload_stream */
/* Flat plans upload on ONE stream per device, shared by all host threads: their arena spans go up at link rate one after the
 * other, in the order the plans were created, so the first chunk of a pooled batch is on the device after 1 / nchunks of the
 * batch's upload time and its kernels run under the remaining uploads.  (Six workers uploading on six streams share the link:
 * every chunk arrives at the END of the total upload time -- config 2: 0.9 ms for each 8 MB chunk, then the kernels.) */
static void *g_shared_up[SHARED_UP_MAXDEV];
static pthread_mutex_t g_shared_up_mu = PTHREAD_MUTEX_INITIALIZER;
static pthread_mutex_t g_shared_issue_mu = PTHREAD_MUTEX_INITIALIZER;      /* held while ONE plan's copies go into the shared stream */
void *shared_upload_stream(void)
{
	const int dev = k2a_shim_get_device();
	void *s;
	if (dev < 0 || dev >= SHARED_UP_MAXDEV) return 0;
	pthread_mutex_lock(&g_shared_up_mu);
	if (!g_shared_up[dev]) g_shared_up[dev] = ENV(PLAIN_UP_STREAMS) ? k2a_shim_stream_create() : k2a_shim_stream_create_low();
	s = g_shared_up[dev];
	pthread_mutex_unlock(&g_shared_up_mu);
	return s;
}

/* the second copy lane of streamed plans (stream_issue): its own stream, i.e. its own DMA engine */
static void *g_shared_up2[SHARED_UP_MAXDEV];
static void *shared_upload_stream2(void)
{
	const int dev = k2a_shim_get_device();
	void *s;
	if (dev < 0 || dev >= SHARED_UP_MAXDEV) return 0;
	pthread_mutex_lock(&g_shared_up_mu);
	if (!g_shared_up2[dev]) g_shared_up2[dev] = ENV(PLAIN_UP_STREAMS) ? k2a_shim_stream_create() : k2a_shim_stream_create_low();
	s = g_shared_up2[dev];
	pthread_mutex_unlock(&g_shared_up_mu);
	return s;
}
/* KSW2AMD_STREAM_LANES=1 / 2 forces; unset: two lanes for flat arenas (nothing but the copies to wait for: +15-20 % on config 2), one
 * where a gather fills the staging buffer at the same time (two lose there: profiles/r5_stream_lanes_ab.txt) */
static int stream_lanes(int flat) { const char *e = ENV(STREAM_LANES); return e && *e ? (atoi(e) >= 2 ? 2 : 1) : flat ? 2 : 1; }

/* side streams + events for plans with several kernel classes: the classes are independent, and a class of a few long
 * alignments would otherwise hold the whole device for the duration of one alignment while the next class waits */
#define NSIDE 3
static __thread void *g_side[NSIDE], *g_side_ev[NSIDE + 1];
static int side_streams(void)
{
	int i;
	if (g_side[0]) return 0;
	thread_owns_cache();
	for (i = 0; i < NSIDE; ++i) { g_side[i] = k2a_shim_stream_create(); if (!g_side[i]) return -1; }
	for (i = 0; i <= NSIDE; ++i) { g_side_ev[i] = k2a_shim_event_create(); if (!g_side_ev[i]) return -1; }
	return 0;
}

static void cache_free_raw(int kind, void *p) { if (BUF_IS_HOST(kind)) k2a_shim_host_free(p); else k2a_shim_free(p); }

void *cache_get(int kind, size_t bytes, size_t *cap)
{
	void *p;
	int d, best = -1;
	for (d = 0; d < CACHE_DEPTH; ++d)          /* the smallest cached buffer that is large enough */
		if (g_cache[kind][d].p && g_cache[kind][d].cap >= bytes && (best < 0 || g_cache[kind][d].cap < g_cache[kind][best].cap)) best = d;
	if (best >= 0) {
		p = g_cache[kind][best].p; *cap = g_cache[kind][best].cap;
		g_cache[kind][best].p = 0; g_cache[kind][best].cap = 0;
		return p;
	}
	*cap = bytes + bytes / 8 + 256;                       /* a little slack so slightly larger follow-up batches still fit */
	if (g_env_ready && g_env[ENV_TRACE] && atoi(g_env[ENV_TRACE]) >= 2) {
		struct timespec a, b; void *q;
		clock_gettime(CLOCK_MONOTONIC, &a);
		q = BUF_IS_HOST(kind) ? k2a_shim_host_malloc(*cap) : k2a_shim_malloc(*cap);
		clock_gettime(CLOCK_MONOTONIC, &b);
		fprintf(stderr, "[ksw2_amd] buffer cache miss: kind %d, %zu bytes (%s), %.2f ms; cached of that kind: %zu / %zu\n", kind, *cap, BUF_IS_HOST(kind) ? "pinned host" : "device",
		        (b.tv_sec - a.tv_sec) * 1e3 + (b.tv_nsec - a.tv_nsec) * 1e-6, g_cache[kind][0].cap, g_cache[kind][1].cap);
		return q;
	}
	return BUF_IS_HOST(kind) ? k2a_shim_host_malloc(*cap) : k2a_shim_malloc(*cap);
}

void cache_put(int kind, void *p, size_t cap)
{
	int d, small = 0;
	if (!p) return;
	thread_owns_cache();
	for (d = 0; d < CACHE_DEPTH; ++d) {
		if (!g_cache[kind][d].p) { g_cache[kind][d].p = p; g_cache[kind][d].cap = cap; return; }
		if (g_cache[kind][d].cap < g_cache[kind][small].cap) small = d;
	}
	if (g_cache[kind][small].cap < cap) {                 /* keep the larger ones */
		cache_free_raw(kind, g_cache[kind][small].p);
		g_cache[kind][small].p = p; g_cache[kind][small].cap = cap;
	} else cache_free_raw(kind, p);
}

void release_thread_cache(void)
{
	int k;
	for (k = 0; k < BUF_KINDS; ++k) {
		int d;
		for (d = 0; d < CACHE_DEPTH; ++d) { if (g_cache[k][d].p) cache_free_raw(k, g_cache[k][d].p); g_cache[k][d].p = 0; g_cache[k][d].cap = 0; }
	}
	for (k = 0; k < 3; ++k) { if (g_ev_cache[k]) k2a_shim_event_destroy(g_ev_cache[k]); g_ev_cache[k] = 0; }
	if (g_stream) { k2a_shim_stream_sync(g_stream); k2a_shim_stream_destroy(g_stream); g_stream = 0; }
	if (g_up_stream) { k2a_shim_stream_sync(g_up_stream); k2a_shim_stream_destroy(g_up_stream); g_up_stream = 0; }
	for (k = 0; k < NSIDE; ++k) if (g_side[k]) { k2a_shim_stream_sync(g_side[k]); k2a_shim_stream_destroy(g_side[k]); g_side[k] = 0; }
	for (k = 0; k <= NSIDE; ++k) if (g_side_ev[k]) { k2a_shim_event_destroy(g_side_ev[k]); g_side_ev[k] = 0; }
}

/* ---------------------------------------------------------------- streamed plans
 * A batch entry point used to cut a batch into chunks so that uploads overlap kernels -- and paid for it on short reads: eight
 * kernels of half a wavefront per SIMD take 3 ms of device time for what one full launch does in 1.4 (config 2, round 3,
 * profiles/r3_cfg2_phases.txt).  A streamed plan is ONE plan for the whole batch: its sequence arena goes up in pieces on the
 * device's upload stream, behind every piece a block filled with the piece's number is copied onto the plan's watermark block
 * (K2A_WM_BYTES: a size the runtime moves with the DMA engines -- smaller copies, hipStreamWriteValue32 and one-thread "publish"
 * kernels all need a wavefront slot and do not get one while a launch of waiting wavefronts holds the device: tools/probe/
 * stream_publish_probe.hip, profiles/r4_stream_publish_probe.txt), and every score-only packed class runs as ONE launch over the
 * whole batch, started under the upload, whose wavefronts -- dispatched in task order, longest first -- each wait in front of their
 * task until the watermark says its sequences have landed (K2aQueueDesc, k2a_queue_wait).  The wait is bounded (KSW2AMD_STREAM_TIMEOUT_MS, default 2000): a wavefront that gives
 * up raises the launch's abort word, fetch sees it, waits for the upload and runs the plan again as an ordinary one.
 * KSW2AMD_STREAM=0 never, =1 every plan that can (tests), unset: the batch entry points' one-shape score-only batches. */
int stream_env(void) { return env_switch(ENV(STREAM)); }
int64_t stream_min_cells(void) { const char *e = ENV(STREAM_MIN_CELLS); return e && atoll(e) >= 0 ? atoll(e) : 1000000; }      /* cells per pair from which one-shape batches are streamed by default */
static int64_t g_stream_stat[2];           /* streamed plans run, runs that were aborted and repeated unstreamed */
void ksw2amd_stream_stats(int64_t out[2]) { out[0] = g_stream_stat[0]; out[1] = g_stream_stat[1]; }
/* the watermark source: page-locked, block k filled with k + 1, one per device for the life of the process */
static uint32_t *g_wm_src[SHARED_UP_MAXDEV];
static pthread_mutex_t g_wm_mu = PTHREAD_MUTEX_INITIALIZER;
const uint32_t *k2a_wm_source(void);
static const uint32_t *wm_source(void) { return k2a_wm_source(); }
const uint32_t *k2a_wm_source(void)
{
	const int dev = k2a_shim_get_device();
	uint32_t *b;
	if (dev < 0 || dev >= SHARED_UP_MAXDEV) return 0;
	pthread_mutex_lock(&g_wm_mu);
	if (!g_wm_src[dev]) {
		b = (uint32_t*)k2a_shim_host_malloc((size_t)(K2A_MAXPIECES + 1) * K2A_WM_BYTES);
		if (b) {
			size_t k, i;
			for (k = 0; k <= K2A_MAXPIECES; ++k)                    /* (the last block: zeros, what a plan's watermark starts from) */
				for (i = 0; i < K2A_WM_BYTES / 4; ++i) b[k * (K2A_WM_BYTES / 4) + i] = k < K2A_MAXPIECES ? (uint32_t)k + 1 : 0u;
			g_wm_src[dev] = b;
		}
	}
	b = g_wm_src[dev];
	pthread_mutex_unlock(&g_wm_mu);
	return b;
}
/* issue every piece that is ready and allowed, in order; called with piece `k` just completed (k < 0: only look again) */
void stream_issue(stream_up_t *u, int k)
{
	const double t0 = now_ms();
	pthread_mutex_lock(&u->mu);
	if (k >= 0) u->done[k] = 1;
	while (u->next < u->np && u->next < u->hold && (u->all_ready || u->done[u->next]) && !u->rc) {
		const int c = u->next++;
		if (c == 0) u->t_first = t0;
		u->t_last = t0;
		const size_t lo = u->pb[c], hi = u->pb[c + 1], mid = hi < u->src_bytes ? hi : u->src_bytes > lo ? u->src_bytes : lo;
		if (u->sleep_us > 0) {                           /* tests: the kernel must really wait for its pieces */
			struct timespec ts; ts.tv_sec = 0; ts.tv_nsec = (long)u->sleep_us * 1000L;
			k2a_shim_stream_sync(u->up); nanosleep(&ts, 0);
		}
		{
			/* two lanes: the odd pieces' bytes travel on the second stream (another DMA engine, at the same time as the even pieces' on
			 * the first), an event behind each; the first stream waits for that event in front of the piece's watermark -- the
			 * watermarks stay on ONE stream, in order, each behind all the bytes it vouches for */
			void *lane = (u->up2 && (c & 1)) ? u->up2 : u->up;
			if (mid > lo && (u->src_on_device ? k2a_shim_d2d(u->d_seq + lo, u->src + lo, mid - lo, lane) : k2a_shim_h2d(u->d_seq + lo, u->src + lo, mid - lo, lane))) u->rc = -1;
			if (hi > mid && u->tail && k2a_shim_h2d(u->d_seq + mid, u->tail + (mid - u->src_bytes), hi - mid, lane)) u->rc = -1;
			if (lane != u->up) {
				if (!u->ev2[c]) u->ev2[c] = k2a_shim_event_create();
				if (!u->ev2[c] || k2a_shim_event_record(u->ev2[c], lane) || k2a_shim_stream_wait_event(u->up, u->ev2[c])) u->rc = -1;
			}
		}
		if (u->fault && c >= u->np / 2) continue;        /* tests: the watermarks of the second half never arrive -> the launch times out and aborts */
		if (k2a_shim_h2d(u->d_wm, (const uint8_t*)u->wm_src + (size_t)c * K2A_WM_BYTES, K2A_WM_BYTES, u->up)) u->rc = -1;
	}
	u->issue_ms += now_ms() - t0;
	pthread_mutex_unlock(&u->mu);
}
/* ---------------------------------------------------------------- CIGAR memory */

typedef void *(*krealloc_fn)(void *km, void *p, size_t size);

static void *cigar_realloc(void *km, void *p, size_t size)
{
	static krealloc_fn kr = 0;
	if (km == 0) return realloc(p, size);
	if (kr == 0) kr = (krealloc_fn)dlsym(RTLD_DEFAULT, "krealloc");
	if (kr == 0) {
		fprintf(stderr, "[ksw2_amd] km != NULL but the process exports no krealloc() (kalloc.h:14)\n");
		abort();
	}
	return kr(km, p, size);
}

void ez_reset(ksw_extz_t *ez)           /* ksw2.h:184-189; cigar and m_cigar survive */
{
	ez->max_q = ez->max_t = ez->mqe_t = ez->mte_q = -1;
	ez->max = 0; ez->score = ez->mqe = ez->mte = KSW_NEG_INF;
	ez->n_cigar = 0; ez->zdropped = 0; ez->reach_end = 0;
}

void ez_reserve(void *km, ksw_extz_t *ez, int n)   /* capacity sequence of ksw_push_cigar, ksw2.h:116-119 */
{
	int m = ez->m_cigar;
	if (n <= m) return;
	while (m < n) m = m ? m << 1 : 4;
	ez->cigar = (uint32_t*)cigar_realloc(km, ez->cigar, (size_t)m << 2);
	ez->m_cigar = m;
}

/* KSW_EZ_APPROX_MAX without KSW_EZ_APPROX_DROP on the "...2_sse" entry points: the reference then tracks one cell per
 * diagonal only to deliver the final score (ksw2_extz2_sse.c:270-286, ksw2_extd2_sse.c:366-382, ksw2_exts2_sse.c:386-404) -- no
 * max / mqe / mte, no Z-drop -- and returns { score, CIGAR from the corner unless EXTZ_ONLY }, everything else left reset.
 * Reproduced as such; with APPROX_DROP the reference's drop heuristic depends on its padded band and the exact
 * computation is returned instead. */

/* ---------------------------------------------------------------- plan */




/* in-band cells of the exact band |i-j| <= w: sum over target rows i of min(qlen-1, i+w) - max(0, i-w) + 1, closed form */
int64_t band_cells(int qlen, int tlen, int w)
{
	const int64_t T = (int64_t)qlen + w < tlen ? (int64_t)qlen + w : tlen;    /* rows that own a cell */
	const int64_t a = (int64_t)qlen - 1 - w;                                   /* last row whose right end is i+w */
	const int64_t na = a < 0 ? 0 : (a + 1 < T ? a + 1 : T);                    /* rows 0..na-1: en = i+w, then en = qlen-1 */
	const int64_t nb = (int64_t)w + 1 < T ? (int64_t)w + 1 : T;                /* rows 0..nb-1: st = 0, then st = i-w */
	const int64_t sum_en = na * (na - 1) / 2 + na * w + (T - na) * ((int64_t)qlen - 1);
	const int64_t sum_st = (T - nb) * (T - 1 + nb) / 2 - (T - nb) * (int64_t)w;
	return T <= 0 ? 0 : sum_en - sum_st + T;
}

/* steps of the generation-serial schedule; must match k2a_gen_cols() in ksw2_lane.h */
static size_t mp_total_steps(int G, int C, int qlen, int tlen, int w)
{
	const int R = G * C, ngen = (tlen + R - 1) / R;
	size_t tot = 0;
	int g;
	for (g = 0; g < ngen; ++g) {
		const int lo = imax(0, g * R - w), hi = imin(qlen - 1, imin(g * R + R - 1, tlen - 1) + w);
		const int nl = imin(G, (tlen - g * R + C - 1) / C);
		if (hi >= lo) tot += ((size_t)(hi - lo + 1) + (size_t)(nl - 1) + 7) & ~(size_t)7;      /* k2a_gen_pad */
		(void)0;
	}
	return tot;
}

/* does a (G,C) systolic array hold the band?  all strips resident at once, or a lane is done with
 * strip S before strip S+G starts */
int geom_fits(int G, int C, int tlen_eff, int w)
{
	const int nstrips = (tlen_eff + C - 1) / C;
	return nstrips <= G || 2 * (int64_t)w < (int64_t)G * (C + 1) - C + 1;
}

static int cfg_fits(int cfg, int tlen_eff, int w)
{
	if (cfg == K2A_CFG_MP) return 1;                       /* generation-serial: any band */
	return geom_fits(k2a_cfg_G[cfg], k2a_cfg_C[cfg], tlen_eff, w);
}

/* the scores the kernels use: the caller's matrix with KSW_EZ_GENERIC_SC, else match / mismatch / wildcard built from
 * mat[0], mat[1] and the last entry (ksw2_extz2_sse.c:66-69,125-140; ksw2_extd2_sse.c:85-88,166-180) */
void build_eff(int dual, int m, const int8_t *mat, int e, int e2, int generic, int8_t *eff)
{
	int a, b;
	if (generic) memcpy(eff, mat, (size_t)m * m);
	else {
		int scN = mat[m * m - 1] == 0 ? -(dual ? e2 : e) : mat[m * m - 1];
		for (a = 0; a < m; ++a)
			for (b = 0; b < m; ++b)
				eff[a * m + b] = (int8_t)((a == m - 1 || b == m - 1) ? scN : a == b ? mat[0] : mat[1]);
	}
}

static void build_scoring(int dual, int m, const int8_t *mat, int q, int e, int q2, int e2, int generic, K2aScoring *sc)
{
	int8_t eff[K2A_MAXM * K2A_MAXM];
	int a, b;
	memset(sc, 0, sizeof(*sc));
	sc->q = q; sc->e = e; sc->q2 = dual ? q2 : 0; sc->e2 = dual ? e2 : 0;
	sc->m = m;
	build_eff(dual, m, mat, e, e2, generic, eff);
	if (m > 5) return;                                     /* wide alphabets read the matrix itself (sc->mat, set once it is uploaded) */
	for (a = 0; a < m; ++a) {
		uint32_t p = 0;
		for (b = 0; b < 4 && b < m; ++b) p |= (uint32_t)(uint8_t)eff[a * m + b] << (8 * b);
		sc->prof[a] = p;
		sc->colw[a] = m == 5 ? eff[a * m + 4] : 0;
	}
}

/* Packed-int16 class (ksw2_lane_pk.h): ANY scoring matrix over at most five residue codes -- the reference's kernels take any `mat`
 * at their full rate (ksw2_extz2_sse.c:142-143, ksw2_extd2_sse.c:182-183), and since round 5 so do these: the kernels hold a score
 * as its penalty below the matrix's largest entry and look it up in per-query-code column profiles (K2aScoring.cp).  A profile
 * has four bytes, target codes 0..3: pairs whose TARGET holds code 4 leave this class (plan_create_ex), a query's code 4 is table
 * entry 4.  What must hold ... */
typedef struct { int ok, smax, smin, qemax, qemin, q, e, tn1; uint32_t cp[8]; } pkinfo_t;

static void pk_scoring(int dual, int m, const int8_t *mat, int q, int e, int q2, int e2, int generic, pkinfo_t *o)
{
	int8_t eff[25];
	int x, y, ok = (m >= 1 && m <= 5);
	memset(o, 0, sizeof(*o));
	if (ok) {
		build_eff(dual, m, mat, e, e2, generic, eff);
		o->smax = o->smin = eff[0];
		for (x = 0; x < m * m; ++x) { o->smax = imax(o->smax, eff[x]); o->smin = imin(o->smin, eff[x]); }
		for (y = 0; y < m; ++y)                              /* column profile of query code y: penalties against target codes 0..3 */
			for (x = 0; x < 4 && x < m; ++x) o->cp[y] |= (uint32_t)(o->smax - eff[x * m + y]) << (8 * x);
		/* a TARGET wildcard (code 4) that scores the same against every query code -- every match / mismatch / N matrix -- stays
		 * in the packed kernels as a row with penalty 0 out of the profile and a constant taken off its candidate
		 * (K2aScoring.pk_tn1, K2aLanePk::step); KSW2AMD_TN=0: as before round 6 (such pairs take the int32 kernels or are re-run) */
		if (m == 5 && !(ENV(TN) && ENV(TN)[0] == '0')) {
			for (y = 1; y < 5 && eff[4 * 5 + y] == eff[4 * 5]; ++y) {}
			if (y == 5) o->tn1 = o->smax - eff[4 * 5] + 1;
		}
	}
	o->q = q; o->e = e;
	o->qemax = dual ? imax(q + e, q2 + e2) : q + e;
	o->qemin = dual ? imin(q + e, q2 + e2) : q + e;
	/* the fill loop adds these as unsigned 32-bit constants to both halves at once (ksw2_lane_pk.h, offset form) */
	if (q < 0 || e < 0 || (dual && (q2 < 0 || e2 < 0)) || o->smax + e < 0) ok = 0;
	o->ok = ok;
}

/* ... and every in-band H, E, F must provably stay inside (K2A_NEG16 + qemax, K2A_PK_VMAX - qemax) = (-16384 + qemax, 12287 - qemax):
 *   H(i,j) <= smax * min(qlen, tlen);   H(i,j) >= Hb(|i-j|) + (min(i,j)+1) * smin >= -(q + e*w) + min(qlen,tlen) * min(smin,0)
 * (gap along the border, then the diagonal: a path that stays inside the band). */
static int pk_eligible(const pkinfo_t *k, int qlen, int tlen, int w)
{
	const int64_t L = imin(qlen, tlen);
	int64_t hmax, hmin;
	if (k->ok <= 0 || qlen > 32000 || tlen > 32000) return 0;   /* (the score bound below is far tighter) */
	hmax = (int64_t)imax(k->smax, 0) * L + (int64_t)k->e * tlen;     /* + row bias e*i carried by the packed kernels */
	hmin = -((int64_t)k->q + (int64_t)k->e * (w + 1)) + (int64_t)imin(k->smin, 0) * L;
	return hmax < K2A_PK_VMAX - 2 * k->qemax - 8 && hmin > K2A_NEG16 + 2 * k->qemax + 8;      /* the offset form's range: ksw2_types.h */
}

/* Re-based packed kernels: a strip's values are relative to the H diagonally above its first cell, so what has to fit is
 * the spread over the cells a strip holds at once: at most 2w + 2C + 2 unit steps away from that corner, each step
 * changing H by at most D = max(smax + qemin, -smin) (the usual difference bounds of the affine recurrence with qemin =
 * the cheapest one-residue gap; they also hold at the band edges) plus e of row bias; E, F sit at most qemax + D below
 * their H.  The -inf sentinel is -16384, and
 * up to two base shifts (<= 2C * D each) plus one score are added to it before it is clamped again. */
static int pk_window_ok(const pkinfo_t *k, int qlen, int tlen, int w, int C)
{
	const int64_t D = imax(imax(k->smax, 0) + k->qemin, -k->smin) + k->e;
	if (k->ok <= 0 || qlen > 65000 || tlen > 65000) return 0;          /* column / row indices travel as unsigned 16-bit halves */
	/* ... and what is added to or taken from -inf before the band mask clamps it again (two base shifts, a score, a gap
	 * cost) must stay inside the K2A_PK_SLACK units the offset form keeps below it (ksw2_types.h) */
	if ((int64_t)4 * C * D + 2 * k->qemax + imax(k->smax, 0) + (k->smax - k->smin) + 64 > K2A_PK_SLACK) return 0;      /* (smax - smin: the largest penalty a candidate loses) */
	return ((int64_t)2 * w + 2 * C + 2) * D + 2 * k->qemax + (int64_t)4 * C * D + 64 <= 12000;
}

/* Packed generation-serial class (ksw2_lane_pkmp.h): the base slides, so the read length does not matter; what must fit between
 * the -inf sentinel's guard band (K2A_PKMP_DEAD = -8192) and K2A_PK_VMAX is what a lane holds at one column (C rows) plus the drift
 * of K2A_PKMP_T steps until the next re-base, each unit step changing H by at most D (as in pk_window_ok), E / F up to
 * qemax + D below their H. */
static int pk_slide_ok(const pkinfo_t *k, int qlen, int tlen)
{
	const int64_t D = imax(imax(k->smax, 0) + k->qemin, -k->smin) + k->e;
	if (k->ok <= 0 || qlen > 65000 || tlen > 65000) return 0;
	if (K2A_PKMP_RMAX_LIMIT + (K2A_PKMP_T + 4) * D + 2 * k->qemax + 64 > K2A_PK_VMAX) return 0;      /* a row maximum between two checks */
	return (K2A_PKMP_T + 2 * 16 + 4) * D + 2 * k->qemax + 64 <= 6000;
}

/* per byte of a word: 0x01 where the byte is above 4 (an OR over words cannot tell 4 | 1 from 5) */
static inline uint64_t above4(uint64_t v) { return ((v >> 3 | v >> 4 | v >> 5 | v >> 6 | v >> 7) | ((v >> 2) & (v | (v >> 1)))) & 0x0101010101010101ull; }

/* copy a sequence into the staging arena and report the residue codes >= 4 it holds: 0 = none, 1 = the wildcard of a 5-letter
 * alphabet (code 4) and nothing above, 2 = a code above 4.  One pass over the bytes instead of a scan plus a memcpy */
int copy_scan(uint8_t *dst, const uint8_t *src, int n)
{
	int i = 0;
	uint64_t acc = 0, hi = 0, v0, v1, v2, v3;
	for (; i + 32 <= n; i += 32) {
		memcpy(&v0, src + i, 8); memcpy(&v1, src + i + 8, 8); memcpy(&v2, src + i + 16, 8); memcpy(&v3, src + i + 24, 8);
		memcpy(dst + i, &v0, 8); memcpy(dst + i + 8, &v1, 8); memcpy(dst + i + 16, &v2, 8); memcpy(dst + i + 24, &v3, 8);
		acc |= (v0 | v1) | (v2 | v3);
		if (((v0 | v1) | (v2 | v3)) & 0x0404040404040404ull) hi |= (above4(v0) | above4(v1)) | (above4(v2) | above4(v3));      /* (rare: a word with bit 2 somewhere) */
	}
	for (; i < n; ++i) { dst[i] = src[i]; acc |= src[i]; if (src[i] > 4) hi = 1; }
	return (acc & 0xf8f8f8f8f8f8f8f8ull) || hi ? 2 : (acc & 0x0404040404040404ull) ? 1 : 0;
}

void ksw2amd_plan_destroy(ksw2amd_plan_t *p)
{
	int i;
	if (!p) return;
	if (p->gather) gather_wait(p);
	if (p->up_ev) { k2a_shim_event_sync(p->up_ev); k2a_shim_event_destroy(p->up_ev); p->up_ev = 0; }      /* the upload reads host blocks freed below */
	if (p->up_state) {
		int k;
		for (k = 0; k < K2A_MAXPIECES; ++k) if (p->up_state->ev2[k]) k2a_shim_event_destroy(p->up_state->ev2[k]);
		pthread_mutex_destroy(&p->up_state->mu); free(p->up_state); p->up_state = 0;
	}
	if (p->wm_ev) { k2a_shim_event_destroy(p->wm_ev); p->wm_ev = 0; }
	if (p->meta_ev) { k2a_shim_event_sync(p->meta_ev); k2a_shim_event_destroy(p->meta_ev); p->meta_ev = 0; }      /* (its copies read the page-locked staging recycled below) */
	if (p->stream_used) k2a_shim_stream_sync(p->stream);     /* nothing may still be running on buffers that get recycled */
	cache_put(BUF_SEQ, p->d_seq, p->cap[BUF_SEQ]); cache_put(BUF_TB, p->d_tb, p->cap[BUF_TB]);
	if (p->meta_folded != 2) cache_put(BUF_PAIRS, p->d_pairs, p->cap[BUF_PAIRS]);
	cache_put(BUF_RES, p->d_res, p->cap[BUF_RES]);
	if (!p->meta_folded) cache_put(BUF_ORDER, p->d_order, p->cap[BUF_ORDER]);
	cache_put(BUF_CIG, p->d_cig, p->cap[BUF_CIG]);
	cache_put(BUF_BND, p->d_bnd, p->cap[BUF_BND]); cache_put(BUF_WM, p->d_wm, p->cap[BUF_WM]); cache_put(BUF_HMETA, p->h_meta, p->cap[BUF_HMETA]);
	cache_put(BUF_PK4, p->d_pk4, p->cap[BUF_PK4]);
	free(p->h_qd);
	for (i = 0; i < 3; ++i) if (p->ev[i]) { if (!g_ev_cache[i]) g_ev_cache[i] = p->ev[i]; else k2a_shim_event_destroy(p->ev[i]); }
	free(p->h_pairs); free(p->h_cls); free(p->h_half); free(p->h_flag); free(p->h_order);
	cache_put(BUF_HRES, p->h_res, p->cap[BUF_HRES]);          /* pinned: the results come back with one asynchronous copy */
	if (!p->flat) cache_put(BUF_HSEQ, p->h_seq, p->cap[BUF_HSEQ]);      /* (a flat plan's h_seq is the caller's arena) */
	free(p->src_pairs); free(p->src_mat); free(p->flat_tail); free(p->uni);
	free(p);
}

int cmp_cost_desc(const void *a, const void *b)
{
	const sort_t *x = (const sort_t*)a, *y = (const sort_t*)b;
	if (x->cost != y->cost) return x->cost > y->cost ? -1 : 1;
	if (x->tf != y->tf) return x->tf > y->tf ? -1 : 1;
	return x->idx < y->idx ? -1 : x->idx > y->idx;
}

/* `scalar`: the call came through ksw_extz / ksw_extd / ksw_gg* (matrix used as given, no end bonus, no mismatch-vs-gap
 * reject, gap pieces kept in the caller's order).  Decided by the entry point, never by a bit in the caller's flags. */
/* the sequence copy of a plan (pass 1 of plan_create_ex): the bytes into the pinned arena, a wildcard flag per pair.  A big plan
 * created outside the worker pool has the pool's threads share the copy (parallel_copy, behind the pool) */

int trace_level(void) { const char *e = ENV(TRACE); return e ? atoi(e) : 0; }

/* What every plan creator (extz / extd, splice-aware, X-drop, SSE-compatible) starts with: the plan record and its per-pair host
 * arrays.  `with_order`: the task list is as long as the batch (one entry per pair) and allocated here. */
ksw2amd_plan_t *plan_new(const char *who, int n, int with_order)      /* with_order < 0: extz / extd plans -- they initialise every record they use themselves */
{
	const int raw = with_order < 0;
	ksw2amd_plan_t *p;
	if (k2a_shim_device_count() <= 0) { fail(KSW2AMD_E_NODEVICE, "no usable %s device", k2a_shim_backend()); return 0; }
	p = (ksw2amd_plan_t*)calloc(1, sizeof(*p));
	if (!p) { fail(KSW2AMD_E_NOMEM, "%s: host allocation failed", who); return 0; }
	p->n = n;
	p->h_cls = (int8_t*)malloc((size_t)n + 1);
	p->h_half = raw ? (uint8_t*)malloc((size_t)n + 1) : (uint8_t*)calloc((size_t)n + 1, 1);
	p->h_flag = (int32_t*)malloc(sizeof(int32_t) * ((size_t)n + 1));
	/* (65 536 records are 3.6 MB: clearing them was 0.15 ms of a 1 ms plan creation on config 2) */
	p->h_pairs = raw ? (K2aPair*)malloc(((size_t)n + 1) * sizeof(K2aPair)) : (K2aPair*)calloc((size_t)n + 1, sizeof(K2aPair));
	p->h_res = (K2aResult*)cache_get(BUF_HRES, sizeof(K2aResult) * ((size_t)n + 1), &p->cap[BUF_HRES]);
	/* (not cleared: a fetch overwrites all n records before anything reads one, and a plan that launches nothing never looks at them) */
	if (with_order > 0) p->h_order = (uint32_t*)malloc(sizeof(uint32_t) * ((size_t)n + 1));
	if (!p->h_cls || !p->h_half || !p->h_flag || !p->h_pairs || !p->h_res || (with_order > 0 && !p->h_order)) {
		fail(KSW2AMD_E_NOMEM, "%s: host allocation failed", who);
		ksw2amd_plan_destroy(p);
		return 0;
	}
	if (!raw) memset(p->h_cls, -1, (size_t)n + 1);
	return p;
}
/* ... and ends with: the timing events from the thread's cache; the plan no longer refers to the creating thread's upload stream
 * (which may be gone -- thread exit, ksw2amd_release_cache, ksw2amd_set_device -- before the plan runs or is destroyed) */
void plan_ready(ksw2amd_plan_t *p)
{
	int i;
	for (i = 0; i < 3; ++i) { p->ev[i] = g_ev_cache[i] ? g_ev_cache[i] : k2a_shim_event_create(); g_ev_cache[i] = 0; }
	p->stream = 0; p->stream_used = 0;
}

/* `flat`: the pairs' query / target pointers all lie in ONE arena, in host memory or (flat->on_device) in device memory.  The plan
 * then uploads (or copies on the device) the arena's span as it is and addresses the sequences where they lie: no per-pair gather,
 * no staging copy, no host pass over the bytes.  What the gather pass also did was to look for wildcard codes (the packed kernels
 * cannot score them): flat plans leave that to the packed kernels themselves (K2aLanePk::seen) and re-run what they report. */
/* room for a plan's small arrays behind its sequences: K2aPair per pair, the task lists (two entries per packed task at most), slack */
static uint64_t or_bytes(const uint8_t *p, int n)
{
	uint64_t acc = 0, v0, v1, v2, v3;
	int i = 0;
	for (; i + 32 <= n; i += 32) {
		memcpy(&v0, p + i, 8); memcpy(&v1, p + i + 8, 8); memcpy(&v2, p + i + 16, 8); memcpy(&v3, p + i + 24, 8);
		acc |= (v0 | v1) | (v2 | v3);
	}
	for (; i < n; ++i) acc |= p[i];
	return acc;
}
/* codes >= 4 in the TARGET, as copy_scan reports them (0 none, 1 the wildcard code 4 only, 2 a code above 4): what keeps a pair out of
 * the packed kernels -- level 1 only where the scoring has no constant wildcard row (pk_scoring, pkinfo_t.tn1) */
static int pair_wild_level(const ksw2amd_pair_t *a)
{
	const uint64_t acc = or_bytes(a->target, a->tlen);
	int i;
	if (acc & 0xf8f8f8f8f8f8f8f8ull) return 2;
	if (!(acc & 0x0404040404040404ull)) return 0;
	for (i = 0; i < a->tlen; ++i) if (a->target[i] > 4) return 2;      /* (rare: only targets that hold a wildcard get here) */
	return 1;
}

ksw2amd_plan_t *plan_create_ex(int dual, int scalar, const ksw2amd_scoring_t *sc, int n, const ksw2amd_pair_t *pairs, const flat_src_t *flat, int want_stream)
{
	stream_up_t *su = 0;                   /* streamed plans: the piece-wise upload */
	int tmpl = -1, pcur = 0, nfull = 0, ninvalid = 0, uni = 0;   /* one-shape batches: the pair whose classification the others take over; piece cursor;
	                                        * nfull: pairs classified in full; uni: ONE shape, no empty pair -- every per-pair array is one value */
	int64_t tmpl_cells = 0;
	double tph[6] = { 0, 0, 0, 0, 0, 0 };
	const int tlev = trace_level() >= 2;
	ksw2amd_plan_t *p;
	int i, k, q, e, q2, e2, m, lo, ci;
	size_t off, mat_off = 0, flat_span = 0;
	const uint8_t *flat_lo = 0;
	uint8_t *flat_tmp = 0;
	int shared_up = 0;
	void *up;
	sort_t *srt = 0;
	pkinfo_t pkinfo[2];
	uint8_t *pk_ok = 0, *solo_ok = 0;
	/* KSW2AMD_SOLO: unset = alignments without a partner of identical shape take the solo kernel, and so do classes of so few long
	 * reads that each can have a SIMD of its own; 1 = only the former, all = every eligible alignment (tests), 0 = never */
	const char *solo_env = ENV(SOLO);
	const int solo_mode = !solo_env ? 3 : !strcmp(solo_env, "0") ? 0 : !strcmp(solo_env, "all") ? 2 : 1;
	const int use_pk = !ENV(NO_PK), use_rb = !ENV(NO_RB);
	const int use_pkmp = !ENV(NO_PKMP);                          /* A/B runs and tests: wide bands through the int32 generation-serial kernels */
	/* A/B runs: skip the smaller packed geometries.  Single-pair calls and their coalesced batches (g_latency_plan) skip them by
	 * themselves: 8 lanes x 18 rows per alignment is the geometry that fills a device, but a lane then walks 4 strips of 18 rows one
	 * after the other -- one 512 x 512, w = 64 pair takes 0.455 ms that way and 0.33 ms with 8 rows per lane, and a caller that waits
	 * for ONE pair, or 64 threads that wait for their 64, wait for exactly that.  (16 lanes x 8 rows, 64 x 8, the solo kernel and 64 x 16
	 * are within 10 % of each other, the first ahead at every batch size up to 256: profiles/r4_latency_probe.txt.) */
	const int pk_first = ENV(PK_FIRST) ? atoi(ENV(PK_FIRST)) : (g_latency_plan && n <= 256) ? 1 : 0;

	if (tlev) tph[0] = now_ms();
	g_err[0] = 0;
	if (n < 0 || (n > 0 && !pairs) || !sc) { fail(KSW2AMD_E_PARAM, "plan_create: bad arguments%s", 0); return 0; }
	p = plan_new("plan_create", n, -1);
	if (!p) return 0;
	p->dual = !!dual; p->m = m = sc->m;
	q = sc->q; e = sc->e; q2 = sc->q2; e2 = sc->e2;
	for (i = 0; i < n; ++i) {
		p->h_cls[i] = -1; p->h_flag[i] = (pairs[i].flag & ~F_SCALAR_CONTRACT) | (scalar ? F_SCALAR_CONTRACT : 0);
		if (pairs[i].qlen <= 0 || pairs[i].tlen <= 0) memset(&p->h_pairs[i], 0, sizeof(K2aPair));     /* never aligned; the others are set field by field below */
	}
	p->h_cls[n] = -1; memset(&p->h_pairs[n], 0, sizeof(K2aPair)); p->h_half[n] = 0;

	/* batch-level early rejects of the "...2_sse" signatures; the scalar-contract entry points skip the
	 * mismatch-vs-gap test (ksw_extz has none) but still need a usable matrix */
	{
		if (m <= 0 || (dual && m <= 1) || !sc->mat) p->reject_all = 1;
		else if (m > K2A_MAXM) { fail(KSW2AMD_E_PARAM, "more than 127 residue types (int8_t m, ksw2.h:61)%s", 0); goto err; }
		else {
			/* ksw2_extd2_sse.c:78: cheaper-to-open piece first (the scalar ksw_extd keeps the caller's order) */
			if (dual && !scalar && q2 + e2 < q + e) { int t = q; q = q2; q2 = t; t = e; e = e2; e2 = t; }
			for (k = 1, lo = sc->mat[m * m > 1 ? 1 : 0]; k < m * m; ++k) lo = imin(lo, sc->mat[k]);
			if (!scalar && -lo > 2 * (q + e)) p->reject_all = 1;                                     /* ksw2_extz2_sse.c:78-82 */
		}
	}
	if (p->reject_all || n == 0) return p;
	/* the scoring, for pairs that fetch runs again (pair_rerun) */
	p->scalar = scalar;
	p->src_mat = (int8_t*)malloc((size_t)m * m);
	if (!p->src_mat) { fail(KSW2AMD_E_NOMEM, "plan_create: host allocation failed%s", 0); goto err; }
	memcpy(p->src_mat, sc->mat, (size_t)m * m);
	p->src_sc = *sc; p->src_sc.mat = p->src_mat;

	/* pass 0: sequence arena (query 4-aligned, target 16-aligned and readable one strip past its end) in pinned staging */
	off = 0;
	if (flat) {
		/* the span of the caller's arena that holds this plan's sequences; offsets are taken from its first byte.  (The kernels
		 * read sequences with byte and unaligned dword loads; what they read past a sequence's end is only ever seen by cells
		 * outside the target / the band.) */
		const uint8_t *lo = 0, *hi = 0;
		size_t sum = 0;
		for (i = 0; i < n; ++i) {
			const ksw2amd_pair_t *a = &pairs[i];
			if (a->qlen <= 0 || a->tlen <= 0) continue;
			if (!a->query || !a->target) { fail(KSW2AMD_E_PARAM, "plan_create: NULL sequence%s", 0); goto err; }
			if (!lo || a->query < lo) lo = a->query;
			if (a->target < lo) lo = a->target;
			if (a->query + a->qlen > hi) hi = a->query + a->qlen;
			if (a->target + a->tlen > hi) hi = a->target + a->tlen;
			sum += (size_t)a->qlen + (size_t)a->tlen;
		}
		off = lo ? (size_t)(hi - lo) : 0;
		flat_lo = lo;
		/* pairs scattered over the arena (the span goes up whole), or a span beyond the 32-bit offsets: gather instead */
		if (off > 0xfff00000u || off > 4 * sum + ((size_t)1 << 20)) {
			if (flat->on_device) { fail(KSW2AMD_E_PARAM, "plan_create_flat: the pairs of one plan span more than 4 GiB (or lie scattered) in the device arena%s", 0); goto err; }
			flat = 0; off = 0;
		}
	}
	if (flat) {
		const uint8_t *lo = flat_lo;
		for (i = 0; i < n; ++i) {
			const ksw2amd_pair_t *a = &pairs[i];
			if (a->qlen <= 0 || a->tlen <= 0) continue;
			p->h_pairs[i].qoff = (uint32_t)(a->query - lo); p->h_pairs[i].toff = (uint32_t)(a->target - lo);
		}
		flat_span = off;
		p->flat = 1; p->flat_device = flat->on_device;
		p->src_pairs = (ksw2amd_pair_t*)malloc(sizeof(*pairs) * ((size_t)n + 1));
		if (!p->src_pairs) { fail(KSW2AMD_E_NOMEM, "plan_create: host allocation failed%s", 0); goto err; }
		memcpy(p->src_pairs, pairs, sizeof(*pairs) * (size_t)n);
	} else
	for (i = 0; i < n; ++i) {
		const ksw2amd_pair_t *a = &pairs[i];
		if (a->qlen <= 0 || a->tlen <= 0) continue;                                                 /* ksw2_extz2_sse.c:57 */
		if (!a->query || !a->target) { fail(KSW2AMD_E_PARAM, "plan_create: NULL sequence%s", 0); goto err; }
		off = align_up(off, 4); p->h_pairs[i].qoff = (uint32_t)off; off += (size_t)a->qlen;
		off = align_up(off, 16); p->h_pairs[i].toff = (uint32_t)off; off += (size_t)a->tlen + 64;
		if (off > 0xfff00000u) { fail(KSW2AMD_E_PARAM, "plan_create: more than 4 GiB of sequence in one plan%s", 0); goto err; }
	}
	off = align_up(off + 65536, 256);               /* idle lanes may prefetch codes a few hundred bytes past the last pair */
	if (m > 5) { mat_off = off; off = align_up(off + 2 * (size_t)m * m, 256); }   /* wide alphabets: effective matrices, simple | generic */
	p->seq_bytes = off;
	if (flat) p->h_seq = flat->on_device ? 0 : (uint8_t*)flat_lo;      /* borrowed: EQX rewrites and re-runs read the sequences there */
	else if (!g_probe_tasks) {
		p->h_seq = (uint8_t*)cache_get(BUF_HSEQ, p->seq_bytes + META_ROOM(n), &p->cap[BUF_HSEQ]);      /* (+ the small arrays: one upload per plan) */
		if (!p->h_seq) { fail(KSW2AMD_E_NOMEM, "pinned staging allocation failed: %s", k2a_shim_last_error()); goto err; }
	}

	if (tlev) tph[1] = now_ms();
	/* pass 1: copy the codes; geometry class, traceback / CIGAR / boundary space, packed-int16 eligibility per pair */
	pkinfo[0].ok = pkinfo[1].ok = -1;
	pk_ok = (uint8_t*)calloc((size_t)n + 1, 1);
	solo_ok = (uint8_t*)calloc((size_t)n + 1, 1);
	if (!pk_ok || !solo_ok) { fail(KSW2AMD_E_NOMEM, "plan_create: host allocation failed%s", 0); goto err; }
	/* A streamed plan (section "streamed plans"): the arena goes up in pieces, each followed by its watermark, on the device's
	 * upload stream -- a flat arena's first pieces right here, before anything is classified; a gathered one piece by piece as the
	 * pool's threads complete them (gather_start: the copy runs on while this thread lays the plan out and launches it; no scan for
	 * wildcard codes -- the kernels report them, as in flat plans).  Only two pieces go up before the plan's small arrays (they would
	 * queue behind the whole arena otherwise), the rest behind them.  Whether the plan then RUNS streamed is decided once its classes
	 * are known; the pieces go up either way. */
	if (stream_env() == 0) want_stream = 0; else if (stream_env() == 1) want_stream = 1;
	if (flat && flat->on_device) want_stream = 0;        /* a device-resident arena: nothing to overlap (one device-to-device copy at HBM rate), and that copy is a
	                                                      * KERNEL, which a launch of waiting wavefronts that fills the device would starve (tools/probe/stream_publish_probe.hip) */
	if (g_probe_tasks) want_stream = 0;
	if (g_no_defer) want_stream = 0;                     /* a fetch's re-run of pairs the kernels handed back: through the SCANNED gather path, whose
	                                                      * wildcard flags send a pair to the int32 kernels -- unscanned it would come back again */
	if (want_stream && n > 0 && (p->seq_bytes >= ((size_t)1 << 20) || stream_env() == 1) && (su = (stream_up_t*)calloc(1, sizeof(*su))) != 0) {
		const char *pk_ = ENV(STREAM_PIECE_KB);
		size_t pbytes = pk_ && atol(pk_) > 0 ? (size_t)atol(pk_) << 10 : p->seq_bytes / (flat ? 12 : 24);
		if (!(pk_ && atol(pk_) > 0)) { if (pbytes < ((size_t)1 << 20)) pbytes = (size_t)1 << 20; if (pbytes > ((size_t)32 << 20)) pbytes = (size_t)32 << 20; }
		if (pbytes * K2A_MAXPIECES < p->seq_bytes) pbytes = p->seq_bytes / K2A_MAXPIECES + 1;
		pbytes = align_up(pbytes, 256);
		p->up_state = su;
		su->t0 = now_ms();
		pthread_mutex_init(&su->mu, 0);
		su->wm_src = wm_source();
		su->up = shared_upload_stream();
		su->up2 = stream_lanes(flat != 0) == 2 ? shared_upload_stream2() : 0;
		su->fault = env_flag(ENV(STREAM_FAULT), 0); su->sleep_us = ENV(STREAM_SLEEP_US) ? atoi(ENV(STREAM_SLEEP_US)) : 0;
		su->hold = 2;
		p->d_seq = (uint8_t*)cache_get(BUF_SEQ, p->seq_bytes, &p->cap[BUF_SEQ]);
		p->d_wm = (uint8_t*)cache_get(BUF_WM, K2A_WM_BYTES + NCLS_ENTRIES * sizeof(K2aQueueDesc), &p->cap[BUF_WM]);
		if (!su->wm_src || !su->up || !p->d_seq || !p->d_wm) { fail(KSW2AMD_E_NOMEM, "device allocation failed: %s", k2a_shim_last_error()); goto err; }
		su->d_seq = p->d_seq; su->d_wm = p->d_wm;
		/* the watermark starts at zero: on the thread's own stream and waited for, so it is there before the first piece's watermark */
		/* the watermark starts at zero: a DMA copy of a zero block at the head of the upload stream, in front of the first piece's
		 * watermark; the stream the plan runs on waits for the event behind it.  (Not a memset: hipMemsetAsync + a wait on the thread's
		 * own stream took 15-25 ms here, in steps of 5 -- a blit kernel behind the previous plan's DMA copies, round 4.) */
		p->wm_ev = k2a_shim_event_create();
		if (!p->wm_ev || k2a_shim_h2d(p->d_wm, (const uint8_t*)su->wm_src + (size_t)K2A_MAXPIECES * K2A_WM_BYTES, K2A_WM_BYTES, su->up) ||
		    k2a_shim_event_record(p->wm_ev, su->up)) { fail(KSW2AMD_E_NODEVICE, "upload failed: %s", k2a_shim_last_error()); goto err; }
		if (m > 5 && !flat) {
			build_eff(dual, m, sc->mat, e, e2, 0, (int8_t*)p->h_seq + mat_off);
			build_eff(dual, m, sc->mat, e, e2, 1, (int8_t*)p->h_seq + mat_off + (size_t)m * m);
		}
		if (flat) {
			const size_t tail = p->seq_bytes - flat_span;
			flat_tmp = (uint8_t*)calloc(tail ? tail : 1, 1);
			if (!flat_tmp) { fail(KSW2AMD_E_NOMEM, "plan_create: host allocation failed%s", 0); goto err; }
			if (m > 5) {
				build_eff(dual, m, sc->mat, e, e2, 0, (int8_t*)flat_tmp + (mat_off - flat_span));
				build_eff(dual, m, sc->mat, e, e2, 1, (int8_t*)flat_tmp + (mat_off - flat_span) + (size_t)m * m);
			}
			su->src = flat_lo; su->src_bytes = flat_span; su->tail = flat_tmp; su->src_on_device = flat->on_device; su->all_ready = 1;
			su->np = (int)((p->seq_bytes + pbytes - 1) / pbytes);
			for (k = 0; k <= su->np; ++k) su->pb[k] = (size_t)k * pbytes < p->seq_bytes ? (size_t)k * pbytes : p->seq_bytes;
		} else {
			/* gathered arenas: pieces start at pair boundaries (the copy's work units); the pairs lie in the arena in index order */
			su->src = p->h_seq; su->src_bytes = p->seq_bytes;
			su->np = 0; su->pb[0] = 0; su->pfirst[0] = 0;
			for (i = 0; i < n; ++i)
				if (pairs[i].qlen > 0 && pairs[i].tlen > 0 && p->h_pairs[i].qoff >= su->pb[su->np] + pbytes && su->np + 1 < K2A_MAXPIECES) {
					++su->np; su->pb[su->np] = p->h_pairs[i].qoff; su->pfirst[su->np] = i;
				}
			++su->np; su->pb[su->np] = p->seq_bytes; su->pfirst[su->np] = n;
			p->unscanned = 1;
		}
		p->npieces = su->np;
		p->stream = su->up; p->stream_used = 1; shared_up = 1;
		if (flat) stream_issue(su, -1);
		else if (gather_start(p, su, pairs, n)) {           /* the pool cannot take it (a worker's own plan, another caller's batch): copy here, piece by piece */
			copy_ctx_t cc = { 0 };
			cc.h_seq = p->h_seq; cc.hp = p->h_pairs; cc.pairs = pairs; cc.wild = 0; cc.su = su;
			su->hold = su->np;
			for (k = 0; k < su->np; ++k) { copy_range(&cc, su->pfirst[k], su->pfirst[k + 1]); stream_issue(su, k); }      /* (cc.su is not consulted by copy_range itself) */
		}
		if (su->rc) { fail(KSW2AMD_E_NODEVICE, "upload failed: %s", k2a_shim_last_error()); goto err; }
	} else {
		copy_ctx_t cc = { 0 };
		su = 0;
		cc.h_seq = p->h_seq; cc.hp = p->h_pairs; cc.pairs = pairs; cc.wild = solo_ok;              /* (solo_ok doubles as the wildcard flags until the loop below sets it) */
		cc.su = 0;
		if (!flat && !g_probe_tasks && !parallel_copy(&cc, n, p->seq_bytes)) copy_range(&cc, 0, n);  /* flat: nothing is copied, nothing scanned (wild = 0) */
	}
	for (i = 0; i < n; ++i) {
		const ksw2amd_pair_t *a = &pairs[i];
		K2aPair *d = &p->h_pairs[i];
		const int fl = p->h_flag[i];
		int w = a->w, cfg, mode, generic, mx, wild;
		if (a->qlen <= 0 || a->tlen <= 0) { ++ninvalid; continue; }
		wild = solo_ok[i]; solo_ok[i] = 0;                    /* copy_scan's level: 0, 1 = code 4 only, 2 = above */
		if (wild == 1) {                                      /* the target's wildcard: a packed row like any other where the scoring allows (pkinfo_t.tn1) */
			const int gen_ = (fl & (KSW_EZ_GENERIC_SC | F_SCALAR_CONTRACT)) ? 1 : 0;
			if (pkinfo[gen_].ok < 0) pk_scoring(dual, m, sc->mat, q, e, q2, e2, gen_, &pkinfo[gen_]);
			if (pkinfo[gen_].tn1) wild = 0;
		}
		if (su) {                                             /* the upload piece this pair's last byte (+ what the kernels may touch behind it) lies in */
			const size_t qe = (size_t)d->qoff + (size_t)a->qlen, te = (size_t)d->toff + (size_t)a->tlen, end = qe > te ? qe : te;
			const size_t lim = end + K2A_STREAM_MARGIN < p->seq_bytes ? end + K2A_STREAM_MARGIN : p->seq_bytes;
			while (pcur + 1 < su->np && su->pb[pcur + 1] < lim) ++pcur;
			while (pcur > 0 && su->pb[pcur] >= lim) --pcur;
			p->h_half[i] = (uint8_t)(pcur + 1);
		}
		/* one-shape batches (and runs of one shape inside ragged ones): everything below depends on the pair's shape, parameters and
		 * flags only -- take it over from the last pair that was classified in full (a pair with a wildcard code never is a template,
		 * nor one of the generation-serial class, whose boundary rows are per pair) */
		if (tmpl >= 0 && !wild && a->qlen == pairs[tmpl].qlen && a->tlen == pairs[tmpl].tlen && a->w == pairs[tmpl].w && a->zdrop == pairs[tmpl].zdrop &&
		    a->end_bonus == pairs[tmpl].end_bonus && fl == p->h_flag[tmpl]) {
			const K2aPair *t = &p->h_pairs[tmpl];
			d->qlen = t->qlen; d->tlen = t->tlen; d->tlen_full = t->tlen_full; d->w = t->w; d->zdrop = t->zdrop; d->end_bonus = t->end_bonus; d->flag = t->flag;
			d->cig_off = 0; d->tb_off = 0; d->bnd_off = 0; d->pad = 0;
			p->h_cls[i] = p->h_cls[tmpl]; pk_ok[i] = pk_ok[tmpl]; solo_ok[i] = solo_ok[tmpl];
			p->cells += tmpl_cells;
			continue;
		}
		mx = imax(a->qlen, a->tlen);
		if (w < 0 || w > mx) w = mx;                                                               /* ksw2_extz2_sse.c:72 */
		d->qlen = a->qlen; d->tlen_full = a->tlen; d->w = w;
		d->cig_off = 0; d->tb_off = 0; d->bnd_off = 0; d->pad = 0;
		++nfull;
		d->tlen = (int64_t)a->qlen + w < a->tlen ? a->qlen + w : a->tlen;      /* rows i with i-w <= qlen-1 */
		d->zdrop = a->zdrop;
		d->end_bonus = scalar ? K2A_NEG : a->end_bonus;
		d->flag = fl & (KSW_EZ_EXTZ_ONLY | KSW_EZ_REV_CIGAR | KSW_EZ_SCORE_ONLY);
		if (is_approx(fl)) {                                   /* only the score and the corner CIGAR exist in this mode */
			d->zdrop = -1;
			if (fl & KSW_EZ_EXTZ_ONLY) d->flag |= KSW_EZ_SCORE_ONLY;
		}
		for (cfg = 0; cfg < K2A_NCFG; ++cfg) if (cfg_fits(cfg, d->tlen, w)) break;
		mode = (d->flag & KSW_EZ_SCORE_ONLY) ? K2A_MODE_SCORE : (fl & KSW_EZ_RIGHT) ? K2A_MODE_RIGHT : K2A_MODE_LEFT;
		generic = (fl & (KSW_EZ_GENERIC_SC | F_SCALAR_CONTRACT)) ? 1 : 0;
		ci = (cfg * 3 + mode) * 2 + generic;
		p->h_cls[i] = (int8_t)ci;
		tmpl_cells = band_cells(a->qlen, a->tlen, w);
		p->cells += tmpl_cells;
		tmpl = (!wild && cfg != K2A_CFG_MP) ? i : -1;
		if (pkinfo[generic].ok < 0) pk_scoring(dual, m, sc->mat, q, e, q2, e2, generic, &pkinfo[generic]);
		if (use_pk && pkinfo[generic].ok > 0 && a->qlen <= 65000 && a->tlen <= 65000 && !wild) {
			/* packed class: first geometry that holds the band, 1-based; scores that fit 16 bits outright use the plain
			 * kernels, longer reads the re-based ones as long as the band window fits */
			const int plain = pk_eligible(&pkinfo[generic], a->qlen, d->tlen, w);
			int pc;
			/* (8 lanes x 18 rows) needs every register with traceback on: score-only pairs only */
			for (pc = imax(mode == K2A_MODE_SCORE ? 0 : 1, pk_first); pc < K2A_PKCFG_MP; ++pc)
				if (geom_fits(k2a_pkcfg_G[pc], k2a_pkcfg_C[pc], d->tlen, w) &&
				    (plain || (use_rb && pk_window_ok(&pkinfo[generic], a->qlen, d->tlen, w, k2a_pkcfg_C[pc])))) break;
			/* no resident geometry holds the band: the packed generation-serial class (sliding base), exact modes only */
			if (pc == K2A_PKCFG_MP && !(cfg == K2A_CFG_MP && use_rb && use_pkmp && !is_approx(fl) && pk_slide_ok(&pkinfo[generic], a->qlen, d->tlen))) pc = K2A_NPKCFG;
			/* (flat plans: the generation-serial kernels do not report wildcard codes, so a pair goes there only after a look at its
			 * bytes -- which a device arena does not allow; the packed and the solo kernels report them and the host re-runs the pair) */
			if (pc == K2A_PKCFG_MP && (flat || p->unscanned) && ((flat && flat->on_device) || pair_wild_level(a) > (pkinfo[generic].tn1 ? 1 : 0))) pc = K2A_NPKCFG;
			if (pc < K2A_NPKCFG) pk_ok[i] = (uint8_t)(1 + pc + ((plain && pc != K2A_PKCFG_MP) ? 0 : K2A_NPKCFG) + (is_approx(fl) ? 2 * K2A_NPKCFG : 0));
			/* solo kernel: two strips of SC rows per lane, each with its own base (the window of an SC-row strip); a lane must finish
			 * a double strip before its next one starts: 2 * 64 steps + 2 * SC * 64 columns later, against 2 * w + 2 * SC columns */
			const int SC = K2A_SOLO_ROWS(mode == K2A_MODE_SCORE);
			if (solo_mode && !is_approx(fl) && pk_window_ok(&pkinfo[generic], a->qlen, d->tlen, w, SC) &&
			    ((d->tlen + 2 * SC - 1) / (2 * SC) <= 64 || w < 64 * (SC + 1) - SC)) {
				solo_ok[i] = 1;
				if (solo_mode == 2) pk_ok[i] = PASS_SOLO;
			}
		}
		if (cfg == K2A_CFG_MP) {                              /* boundary rows H, E, E~ between generations */
			d->bnd_off = (uint32_t)p->bnd_words;
			p->bnd_words += 3 * (size_t)a->qlen + 16;
			if (p->bnd_words > 0xfff00000u) { fail(KSW2AMD_E_PARAM, "plan_create: boundary rows over 16 GiB in one plan%s", 0); goto err; }
		}
	}

	if (tlev) tph[2] = now_ms();
	/* Alignments without a partner of identical shape would be paired with themselves.  With traceback on a
	 * one-alignment-per-wavefront geometry that is slower than the int32 kernel (tools/scripts/ragged_probe.py: 10 k reads of
	 * unique lengths, CIGAR: 707 vs 825 GCUPS) and writes twice the direction bytes, so the odd one of every shape goes back.
	 * Parity of every (class, shape) key in one pass over an open-addressing table. */
	uni = n > 1 && nfull == 1 && ninvalid == 0;          /* (nothing below has changed a pair's class yet) */
	if (uni) {
		/* one shape: the table below has one key; its odd one out is the last pair */
		if ((n & 1) && p->h_cls[0] >= 0 && pk_ok[0] && pk_ok[0] != PASS_SOLO && ((p->h_cls[0] / 2) % 3 != K2A_MODE_SCORE || (solo_mode && solo_ok[0])) &&
		    k2a_pkcfg_G[(pk_ok[0] - 1) % K2A_NPKCFG] == 64) { pk_ok[n - 1] = (uint8_t)(solo_mode && solo_ok[n - 1] ? PASS_SOLO : 0); uni = 0; }
	} else
	if (n > 0) {
		size_t cap = 16, h;
		struct slot { uint64_t k1, k2; int32_t last, odd; } *tab;
		int any = 0;
		for (i = 0; i < n; ++i)
			if (p->h_cls[i] >= 0 && pk_ok[i] && pk_ok[i] != PASS_SOLO && ((p->h_cls[i] / 2) % 3 != K2A_MODE_SCORE || (solo_mode && solo_ok[i])) && k2a_pkcfg_G[(pk_ok[i] - 1) % K2A_NPKCFG] == 64) ++any;
		if (any) {
			while (cap < 2 * (size_t)any) cap <<= 1;
			tab = (struct slot*)calloc(cap, sizeof(*tab));
			if (!tab) { fail(KSW2AMD_E_NOMEM, "plan_create: host allocation failed%s", 0); goto err; }
			for (i = 0; i < n; ++i) {
				uint64_t k1, k2;
				if (!(p->h_cls[i] >= 0 && pk_ok[i] && pk_ok[i] != PASS_SOLO && ((p->h_cls[i] / 2) % 3 != K2A_MODE_SCORE || (solo_mode && solo_ok[i])) && k2a_pkcfg_G[(pk_ok[i] - 1) % K2A_NPKCFG] == 64)) continue;
				k1 = ((uint64_t)(uint32_t)p->h_pairs[i].qlen << 32) | (uint32_t)p->h_pairs[i].tlen_full;      /* with w this fixes the rows too */
				k2 = ((uint64_t)(uint32_t)p->h_pairs[i].w << 32) | ((uint32_t)p->h_cls[i] << 8) | pk_ok[i] | 0x80000000u;     /* never 0 */
				for (h = (size_t)((k1 * 0x9E3779B97F4A7C15ull ^ k2 * 0xC2B2AE3D27D4EB4Full) >> 20) & (cap - 1); tab[h].k2 && (tab[h].k1 != k1 || tab[h].k2 != k2);
				     h = (h + 1) & (cap - 1)) {}
				tab[h].k1 = k1; tab[h].k2 = k2; tab[h].last = i; tab[h].odd ^= 1;
			}
			for (h = 0; h < cap; ++h)
				if (tab[h].k2 && tab[h].odd) {
					const int last = tab[h].last;
					const int solo = solo_mode && solo_ok[last];
					pk_ok[last] = (uint8_t)(solo ? PASS_SOLO : 0);
				}
			free(tab);
		}
	}

	/* Few long reads.  A packed launch has half the wavefronts of a launch with one read per wavefront, and a wavefront alone on
	 * its SIMD runs at little more than half the SIMD's rate (it cannot issue faster than one instruction per ~6 cycles and nobody
	 * covers its waits: profiles/r3_single_wave_issue.txt, r3_solo_experiments.txt).  So a one-alignment-per-wavefront class of at
	 * most as many reads as the device has SIMDs goes to the solo kernel, read by read: every read gets a SIMD of its own and both
	 * register halves (MI355X, 10 k x 10 k, w = 500, profiles/r3_solo_crossover.txt: 1 024 reads score only 6.2 ms solo, 7.6 ms in
	 * pairs, 8.5 ms int32; with CIGAR 15.4 / 17.8 / 17.4 ms; dual gap with CIGAR 19.3 / 24.5 / 25.4 ms; from 1 536 reads on pairs
	 * win, 17.2 against 21.2 ms).  What cannot go there (approximate modes, a window the solo halves cannot hold) falls back to the int32 kernels below 0.4 packed wavefronts per SIMD as before (round 2: 512
	 * packed wavefronts 11.5 ms, 1 024 int32 wavefronts 8.8 ms).  KSW2AMD_SIMDS overrides the device's SIMD count, 0 = both off. */
	{
		const char *ev = ENV(SIMDS);
		const int simds = ev ? atoi(ev) : k2a_shim_simd_count();
		if (simds > 0) {
			int cnt[NCLS_MAX * NPASS], b;
			memset(cnt, 0, sizeof(cnt));
			if (uni) { if (p->h_cls[0] >= 0 && pk_ok[0] && pk_ok[0] != PASS_SOLO) cnt[p->h_cls[0] * NPASS + pk_ok[0]] = g_probe_tasks ? 2 * g_probe_tasks : n; }
			else
			for (i = 0; i < n; ++i) if (p->h_cls[i] >= 0 && pk_ok[i] && pk_ok[i] != PASS_SOLO) ++cnt[p->h_cls[i] * NPASS + pk_ok[i]];
			for (b = 0; b < NCLS_MAX * NPASS; ++b)
				if (cnt[b]) {
					const int pcb = (b % NPASS - 1) % K2A_NPKCFG, G = k2a_pkcfg_G[pcb];
					const int64_t waves = ((int64_t)(cnt[b] + 1) / 2 * G + 63) / 64 * (pcb == K2A_PKCFG_MP ? 4 : 1);     /* that class: four wavefronts per task */
					/* one-alignment-per-wavefront classes only: for the short shapes of the multi-group geometries the gain is a
					 * fraction of a millisecond per call and costs 2-3 x the SIMD time, which concurrent callers would rather keep */
					cnt[b] = G != 64 ? 0 : (solo_mode == 3 && pcb != K2A_PKCFG_MP && cnt[b] <= simds ? 2 : 0) | (waves * 10 < (int64_t)simds * 4 ? 1 : 0);   /* 2 = solo, 1 = int32 */
				}
			if (uni && !(p->h_cls[0] >= 0 && pk_ok[0] && pk_ok[0] != PASS_SOLO && cnt[p->h_cls[0] * NPASS + pk_ok[0]])) { /* one shape, nothing to demote */ }
			else
			for (i = 0; i < n; ++i)
				if (p->h_cls[i] >= 0 && pk_ok[i] && pk_ok[i] != PASS_SOLO) {
					const int what = cnt[p->h_cls[i] * NPASS + pk_ok[i]];
					if ((what & 2) && solo_ok[i]) pk_ok[i] = PASS_SOLO;
					else if (what & 1) pk_ok[i] = 0;
				}
		}
	}
	/* (Reads without a partner of their shape take the solo kernel whatever their number: with round 3's kernel it is ahead of the
	 * int32 kernels and of pairing a read with itself at every batch size -- unique 8-12 k reads, 256 to 8 192 of them: score only
	 * 1.3-1.4 x int32, dual gap with CIGAR 1.4-2.1 x, r3_solo_crossover.txt.  Round 2 sent them back below four per SIMD.) */

	if (tlev) tph[3] = now_ms();
	/* the sequence arena goes up while the host sorts out the task lists (pinned staging: the copy is asynchronous) */
	if (m > 5 && !flat && !su) {
		build_eff(dual, m, sc->mat, e, e2, 0, (int8_t*)p->h_seq + mat_off);
		build_eff(dual, m, sc->mat, e, e2, 1, (int8_t*)p->h_seq + mat_off + (size_t)m * m);
	}
	if (!su && !g_probe_tasks) p->d_seq = (uint8_t*)cache_get(BUF_SEQ, p->seq_bytes + (flat ? 0 : META_ROOM(n)), &p->cap[BUF_SEQ]);
	if (!p->d_seq && !g_probe_tasks) { fail(KSW2AMD_E_NOMEM, "device allocation failed: %s", k2a_shim_last_error()); goto err; }
	/* Big uploads go through ONE stream per device, whoever issues them: the chunks of a big batch are packed by several worker
	 * threads at once, and six 80 MB copies on six streams share the link -- all of them arrive after 9-13 ms and the device idles
	 * until then (KSW2AMD_TRACE=2 timeline of the 10 k headline); in one queue the first chunk's bytes are there 1.6 ms after its
	 * packing ends and its kernel starts while the others still travel (pointer entry, MI355X: headline 4 024 -> 4 336 GCUPS end to
	 * end, config 2 968 -> 1 165, 10 k with CIGAR 1 290 -> 1 321; config 3's 8 MB chunks and config 5 unchanged within noise).  Plans
	 * under 16 MB (single calls, coalesced batches, small chunks) keep the calling thread's own stream and wait for it: an event per
	 * call would only add latency there.  KSW2AMD_NO_SHARED_UP=1: the old behaviour, for A/B runs. */
	up = su ? su->up : (flat && !flat->on_device) || (!flat && p->seq_bytes >= ((size_t)(ENV(SHARED_UP_MIN_MB) ? imax(atoi(ENV(SHARED_UP_MIN_MB)), 0) : 4) << 20) && !ENV(NO_SHARED_UP)) ? shared_upload_stream() : 0;
	if (up) shared_up = 1; else up = g_plan_stream ? g_plan_stream : thread_upload_stream();
	p->stream = up; p->stream_used = !g_probe_tasks;  /* plan_destroy waits for it before the buffers are recycled */
	if (su) { /* the pieces are on their way (or there) already */ }
	else if (flat) {
		/* the arena's span as it lies there (an upload from caller memory: asynchronous if the caller page-locked it,
		 * ksw2amd_host_register); the padding behind it and the matrices of a wide alphabet from a small staging block */
		const size_t tail = p->seq_bytes - flat_span;
		flat_tmp = (uint8_t*)calloc(tail ? tail : 1, 1);           /* freed behind the stream synchronisation that ends plan creation */
		if (!flat_tmp) { fail(KSW2AMD_E_NOMEM, "plan_create: host allocation failed%s", 0); goto err; }
		if (m > 5) {
			build_eff(dual, m, sc->mat, e, e2, 0, (int8_t*)flat_tmp + (mat_off - flat_span));
			build_eff(dual, m, sc->mat, e, e2, 1, (int8_t*)flat_tmp + (mat_off - flat_span) + (size_t)m * m);
		}
	}
	/* (A plan's copies are issued TOGETHER, below, once its small arrays exist: sequences, small arrays, event -- one copy for a gathered plan.
	 * Issued here, the sequences were followed by the other workers' sequences before this plan's small arrays got into the queue:
	 * the first chunk of config 2 had its 10 MB on the device after 0.16 ms and its kernel started 1.06 ms into the batch, when the
	 * sixth chunk's bytes had arrived too -- round 4, rocprofv3 timeline of the pooled batch, tools/scripts/timeline.py.  The small
	 * arrays on a stream of their own are no way out: their copies are blit kernels that queue up behind another chunk's fill on
	 * whichever hardware queue the stream shares -- config 2 1 160 -> 900 GCUPS, 10 k with CIGAR 1 280 -> 1 040.) */

	/* pass 2: task lists per class, most expensive first (similar shapes end up in the same wavefront).  Packed-int16
	 * candidates of a class are paired up with a neighbour of identical (qlen, tlen, rows inside the band, w); a leftover is
	 * paired with itself. */
	{
		enum { NB = NCLS_MAX * NPASS };
		int bcnt[NB], bpos[NB], b;
		p->ncls = 0; p->ntasks = 0;
		p->h_order = (uint32_t*)malloc(sizeof(uint32_t) * (3 * (size_t)n + 4));      /* task lists (two entries per packed task) + streamed plans' per-wavefront-task piece counts */
		srt = (sort_t*)malloc(sizeof(sort_t) * ((size_t)n + 1));
		if (!p->h_order || !srt) { fail(KSW2AMD_E_NOMEM, "plan_create: host allocation failed%s", 0); goto err; }
		memset(bcnt, 0, sizeof(bcnt));
		if (uni) bcnt[p->h_cls[0] * NPASS + pk_ok[0]] = n;        /* one shape, one class: the task list is the batch in its own order */
		else
		for (i = 0; i < n; ++i) if (p->h_cls[i] >= 0) ++bcnt[p->h_cls[i] * NPASS + pk_ok[i]];
		for (b = 0, k = 0; b < NB; ++b) { bpos[b] = k; k += bcnt[b]; }
		if (uni) { for (b = 0; b < NB; ++b) bpos[b] += bcnt[b]; }
		else
		for (i = 0; i < n; ++i)
			if (p->h_cls[i] >= 0) {
				sort_t *e_ = &srt[bpos[p->h_cls[i] * NPASS + pk_ok[i]]++];
				e_->idx = (uint32_t)i;
				e_->cost = ((int64_t)p->h_pairs[i].qlen << 40) + ((int64_t)p->h_pairs[i].tlen << 16) + p->h_pairs[i].w;
				e_->tf = (uint32_t)p->h_pairs[i].tlen_full;
			}
		for (b = 0, k = 0; b < NB; ++b) {
			const int cnt = bcnt[b], pass = b % NPASS;   /* 0: one alignment per lane group, 1 + pc (+ NPKCFG): packed class pc */
			sort_t *g = srt + (bpos[b] - cnt);
			int ntask = 0;
			cls_t *c;
			if (cnt == 0) continue;
			ci = b / NPASS;
			if (!uni) {
			for (i = 1; i < cnt && cmp_cost_desc(&g[i - 1], &g[i]) <= 0; ++i) {}       /* one shape: already in order */
			if (i < cnt) qsort(g, (size_t)cnt, sizeof(sort_t), cmp_cost_desc);
			}
			c = &p->cls[p->ncls++];
			c->solo = pass == PASS_SOLO;
			c->cfg = c->solo ? 0 : pass ? (pass - 1) % K2A_NPKCFG : ci / 6; c->rb = pass && !c->solo ? ((pass - 1) / K2A_NPKCFG) & 1 : 0;
			c->nomax = !c->solo && pass > 2 * K2A_NPKCFG; c->mode = (ci / 2) % 3; c->generic = ci & 1; c->pk = pass != 0 && !c->solo; c->first = k;
			build_scoring(dual, m, sc->mat, q, e, q2, e2, c->generic, &c->sc);
			if (pkinfo[c->generic].ok > 0) { memcpy(c->sc.cp, pkinfo[c->generic].cp, sizeof(c->sc.cp)); c->sc.pk_smax = pkinfo[c->generic].smax; c->sc.pk_tn1 = pkinfo[c->generic].tn1; }
			if (uni) {
				if (!pass || c->solo) { for (i = 0; i < cnt; ++i) p->h_order[k++] = (uint32_t)i; ntask = cnt; }
				else for (i = 0; i < cnt; i += 2, ++ntask) { p->h_order[k++] = (uint32_t)i; p->h_order[k++] = (uint32_t)(i + 1 < cnt ? i + 1 : i); }
			} else
			if (!pass || c->solo) {
				for (i = 0; i < cnt; ++i) p->h_order[k++] = g[i].idx;
				ntask = cnt;
			} else {
				for (i = 0; i < cnt; ++ntask) {
					const uint32_t ia = g[i].idx;
					uint32_t ib = ia;
					/* same (qlen, rows, w) AND same true target length: a target cut off by the band (rows < tlen) has no last row */
					if (i + 1 < cnt && g[i + 1].cost == g[i].cost && g[i + 1].tf == g[i].tf) { ib = g[i + 1].idx; i += 2; } else i += 1;
					p->h_order[k++] = ia; p->h_order[k++] = ib;
				}
			}
			c->count = ntask;
			p->ntasks += ntask;
		}
		p->norder = k;
		free(srt); srt = 0;
	}

	if (tlev) tph[4] = now_ms();
	/* packed generation-serial tasks: boundary entries + the four wavefronts' row-maximum keys (ksw2_shim.h), shared by the two alignments */
	for (k = 0; k < p->ncls; ++k) {
		const cls_t *c = &p->cls[k];
		if (!c->pk || c->cfg != K2A_PKCFG_MP) continue;
		for (i = 0; i < c->count; ++i) {
			K2aPair *da = &p->h_pairs[p->h_order[c->first + 2 * i]], *db = &p->h_pairs[p->h_order[c->first + 2 * i + 1]];
			p->bnd_words = align_up(p->bnd_words, 4);
			da->bnd_off = db->bnd_off = (uint32_t)p->bnd_words;
			p->bnd_words += align_up((size_t)da->qlen * (dual ? 5 : 4) + 16, 4) + K2A_PKMP_WAVES * (size_t)(64 * 16 * 2 * 2);      /* K2A_PKMP_BND_WORDS + K2A_PKMP_WAVES x K2A_PKMP_SPILL_WORDS(16) */
			if (p->bnd_words > 0xfff00000u) { fail(KSW2AMD_E_PARAM, "plan_create: boundary rows over 16 GiB in one plan%s", 0); goto err; }
		}
	}

	/* Deferred arg-max (K2aLanePk, DEFER): the exact score-only single-gap packed classes track row maxima without their columns and
	 * stream a checkpoint per wavefront and step into the traceback arena (unused by score-only tasks); a second kernel re-runs the
	 * strips whose columns the results need.  3 of 15 instructions per row pair (11 of 51 cycles) for 512 bytes of coalesced writes per
	 * wavefront and step: the 10 k x 10 k headline 3 985 -> 4 692 GCUPS (round 3, same box).  Every pair of a wavefront gets the wavefront's block: tb_off = byte offset, bnd_off = steps of the stream, cig_off =
	 * strips per group in the header table (ksw2_shim.h).  KSW2AMD_DEFER=0 / 1 forces it; by default classes of at least 32 tasks
	 * take it if they keep 1.5 wavefronts on every SIMD, unless the checkpoints of the plan would not fit beside everything else. */
	{
		const char *ev = ENV(DEFER);
		const int forced = ev && *ev ? (atoi(ev) != 0) : -1;
		size_t ck_total = 0;
		for (k = 0; k < p->ncls; ++k) {
			cls_t *c = &p->cls[k];
			c->defer = c->pk && !c->solo && c->cfg != K2A_PKCFG_MP && !dual && c->mode == K2A_MODE_SCORE && !c->nomax && !g_no_defer &&
			           /* by default the one-alignment-per-wavefront geometries only: config 2's (8, 18) measured 2 749 against 2 827 GCUPS
			            * with it (two wavefronts per SIMD either way, a vector wavefront index in the store address), round 3 */
			           /* ... and only where the launch keeps at least 1.5 wavefronts on every SIMD: the deferred kernels of the 16-row
			            * geometry have their code planes in LDS, whose latency a lone wavefront cannot hide (1 024 pairs of 10 k x 10 k =
			            * 512 wavefronts: 1 063 GCUPS deferred against 1 241 from registers) */
			           (forced < 0 ? k2a_pkcfg_G[c->cfg] == 64 && (k2a_shim_simd_count() > 0 ? 2 * (int64_t)(g_probe_tasks ? g_probe_tasks : c->count) >= 3 * (int64_t)k2a_shim_simd_count()
			                                                                                      : (g_probe_tasks ? g_probe_tasks : c->count) >= 32) : forced);
		}
		if (g_probe_tasks) { free(pk_ok); free(solo_ok); return p; }      /* the class is known: nothing was copied, allocated on the device or uploaded */
		for (lo = 0; lo < 2; ++lo) {                       /* 0: size it, 1: lay it out */
			size_t at = p->tb_bytes;
			for (k = 0; k < p->ncls; ++k) {
				const cls_t *c = &p->cls[k];
				const int G = c->pk ? k2a_pkcfg_G[c->cfg] : 64, C = c->pk ? k2a_pkcfg_C[c->cfg] : 16, NG = 64 / G;
				int t0;
				if (!c->defer) continue;
				at += align_up(4 * K2A_ZLIST_WORDS(c->count), 256);      /* the class's list of frozen books, right in front of its first checkpoint block (ksw2_types.h) */
				for (t0 = 0; t0 < c->count; t0 += NG) {
					uint32_t steps = 0, hs = 0;
					int t;
					size_t bytes;
					for (t = t0; t < imin(c->count, t0 + NG); ++t) {
						const K2aPair *d = &p->h_pairs[p->h_order[c->first + 2 * t]];
						const uint32_t ns = (uint32_t)((d->tlen + C - 1) / C);
						steps = (uint32_t)imax((int)steps, (int)ns - 1 + imin(d->qlen - 1, d->tlen - 1 + d->w) + 1);
						hs = (uint32_t)imax((int)hs, (int)ns);
					}
					bytes = align_up((size_t)steps * 512 + (size_t)NG * hs * 16, 256);
					if (lo) for (t = t0; t < imin(c->count, t0 + NG); ++t) {
						K2aPair *da = &p->h_pairs[p->h_order[c->first + 2 * t]], *db = &p->h_pairs[p->h_order[c->first + 2 * t + 1]];
						da->tb_off = db->tb_off = at; da->bnd_off = db->bnd_off = steps; da->cig_off = db->cig_off = hs;
					}
					at += bytes;
				}
			}
			if (!lo) {
				ck_total = at - p->tb_bytes;
				if (ck_total > ((size_t)1 << 30) && forced < 0) {            /* big: only if it fits beside the rest of the device's tenants */
					size_t free_b = 0, total_b = 0;
					if (k2a_shim_mem_info(&free_b, &total_b) || ck_total > (free_b + thread_cached_device_bytes()) / 10 * 6) {
						for (k = 0; k < p->ncls; ++k) p->cls[k].defer = 0;
						ck_total = 0;
						break;
					}
				}
				if (ck_total == 0) break;
			} else p->tb_bytes = at;
		}
	}

	/* pass 3: traceback blocks (one per task: the two alignments of a packed task share theirs) and CIGAR scratch */
	for (k = 0; k < p->ncls; ++k) {
		const cls_t *c = &p->cls[k];
		if (c->mode == K2A_MODE_SCORE) continue;
		for (i = 0; i < c->count; ++i) {
			const uint32_t ia = p->h_order[c->first + (c->pk ? 2 * i : i)];
			const uint32_t ib = c->pk ? p->h_order[c->first + 2 * i + 1] : ia;
			K2aPair *da = &p->h_pairs[ia], *db = &p->h_pairs[ib];
			const int G = c->pk ? k2a_pkcfg_G[c->cfg] : k2a_cfg_G[c->cfg], C = c->pk ? k2a_pkcfg_C[c->cfg] : k2a_cfg_C[c->cfg];
			const int nstrips = (da->tlen + C - 1) / C;
			size_t steps = (size_t)(nstrips - 1) + (size_t)imin(da->qlen - 1, da->tlen - 1 + da->w) + 1;
			const size_t wb = c->pk ? (size_t)K2A_PK_TB_BYTES(C, dual) : (size_t)C * (dual ? 8 : 4) / 8;
			if (!c->solo && (c->pk ? c->cfg == K2A_PKCFG_MP : c->cfg == K2A_CFG_MP)) steps = mp_total_steps(G, C, da->qlen, da->tlen, da->w);
			da->tb_off = db->tb_off = p->tb_bytes;
			if (c->solo) {        /* k2a_solo_steps: 2 * (double strips - 1) + 2 + last column; 64 lanes x 2 * K2A_SOLO_C bytes per step */
				const int nds = (da->tlen + 2 * K2A_SOLO_C - 1) / (2 * K2A_SOLO_C);
				p->tb_bytes += align_up(K2A_TB_PADDED((size_t)(2 * (nds - 1) + 2) + (size_t)imin(da->qlen - 1, da->tlen - 1 + da->w)) * 64 * 2 * K2A_SOLO_C, 256);
			} else
			p->tb_bytes += align_up(K2A_TB_PADDED(steps) * G * wb, 256);          /* lane runs padded: k2a_tb_word */
			da->cig_off = (uint32_t)p->cig_words;
			p->cig_words += (size_t)da->qlen + da->tlen_full + 2;
			if (ib != ia) { db->cig_off = (uint32_t)p->cig_words; p->cig_words += (size_t)db->qlen + db->tlen_full + 2; }
			if (p->cig_words > 0xfff00000u) { fail(KSW2AMD_E_PARAM, "plan_create: CIGAR scratch over 16 GiB in one plan%s", 0); goto err; }
		}
	}

	/* streamed plans: which classes run as queues (the resident packed kernels), their descriptors, and per wavefront-task the
	 * pieces it has to wait for */
	for (k = 0; k < p->ncls; ++k) p->cls[k].qd = -1;
	if (su) {
		size_t at = 0;
		const char *te = ENV(STREAM_TIMEOUT_MS);
		const uint64_t ticks = (uint64_t)(te && atoi(te) > 0 ? atoi(te) : 2000) * 100000u;      /* 100 MHz wall clock */
		for (k = 0; k < p->ncls; ++k) {
			cls_t *c = &p->cls[k];
			if (c->pk && !c->solo && c->cfg != K2A_PKCFG_MP && c->mode == K2A_MODE_SCORE) c->qd = p->nqd++;      /* (the QUEUE builds of the kernels: score-only) */
		}
		if (p->nqd) {
			p->h_qd = (K2aQueueDesc*)calloc((size_t)p->nqd, sizeof(K2aQueueDesc));
			if (!p->h_qd) { fail(KSW2AMD_E_NOMEM, "plan_create: host allocation failed%s", 0); goto err; }
			for (k = 0; k < p->ncls; ++k) {
				const cls_t *c = &p->cls[k];
				const int NG = c->qd >= 0 ? 64 / k2a_pkcfg_G[c->cfg] : 1, nwt = (c->count + NG - 1) / NG;
				uint32_t *need = p->h_order + p->norder + at;
				int wt, t;
				if (c->qd < 0) continue;
				for (wt = 0; wt < nwt; ++wt) {
					uint32_t nd = 0;
					for (t = wt * NG; t < imin(c->count, (wt + 1) * NG); ++t) {
						const uint32_t na = p->h_half[p->h_order[c->first + 2 * t]], nb = p->h_half[p->h_order[c->first + 2 * t + 1]];
						if (na > nd) nd = na;
						if (nb > nd) nd = nb;
					}
					need[wt] = nd;
				}
				p->h_qd[c->qd].nwt = (uint32_t)nwt; p->h_qd[c->qd].timeout_ticks = ticks;
				p->h_qd[c->qd].unp_fmt = (uint32_t)at;         /* (host side only, until the loop below: where this class's piece counts start) */
				at += (size_t)nwt;
			}
			p->need_words = at;
			p->streamed = 1;
		}
	}

	if (tlev) tph[5] = now_ms();
	/* upload the rest */
	/* the small arrays: behind the sequences in the same buffer (gathered plans: ONE upload), or the task lists behind the pairs (flat
	 * plans, whose sequences come from the caller's arena); a streamed plan keeps them apart (its pieces are on their way already) */
	if (su) {
		p->d_pairs = (K2aPair*)cache_get(BUF_PAIRS, sizeof(K2aPair) * ((size_t)n + 1), &p->cap[BUF_PAIRS]);
		p->d_order = (uint32_t*)cache_get(BUF_ORDER, sizeof(uint32_t) * ((size_t)p->norder + p->need_words + 1), &p->cap[BUF_ORDER]);
	} else if (flat) {
		p->d_pairs = (K2aPair*)cache_get(BUF_PAIRS, META_ROOM(n), &p->cap[BUF_PAIRS]);
		p->d_order = p->d_pairs ? (uint32_t*)((uint8_t*)p->d_pairs + align_up(sizeof(K2aPair) * (size_t)n, 256)) : 0;
		p->meta_folded = 1;
	} else {
		p->d_pairs = (K2aPair*)(p->d_seq + align_up(p->seq_bytes, 256));
		p->d_order = (uint32_t*)((uint8_t*)p->d_pairs + align_up(sizeof(K2aPair) * (size_t)n, 256));
		p->meta_folded = 2;
	}
	p->d_res = (K2aResult*)cache_get(BUF_RES, sizeof(K2aResult) * ((size_t)n + 1), &p->cap[BUF_RES]);
	p->d_tb = p->tb_bytes ? (uint8_t*)cache_get(BUF_TB, p->tb_bytes, &p->cap[BUF_TB]) : 0;
	p->d_cig = p->cig_words ? (uint32_t*)cache_get(BUF_CIG, p->cig_words * 4, &p->cap[BUF_CIG]) : 0;
	p->d_bnd = p->bnd_words ? (int32_t*)cache_get(BUF_BND, p->bnd_words * 4, &p->cap[BUF_BND]) : 0;
	if (!p->d_pairs || !p->d_res || !p->d_order || (p->tb_bytes && !p->d_tb) || (p->cig_words && !p->d_cig) ||
	    (p->bnd_words && !p->d_bnd)) {
		fail(KSW2AMD_E_NOMEM, "device allocation failed: %s", k2a_shim_last_error());
		goto err;
	}
	if (su) p->up_ev = k2a_shim_event_create();
	if (su) {
		/* a streamed plan's small arrays go up on the upload stream itself, between the second piece and the third (su->hold): nothing
		 * of the plan waits on the host for them -- the stream the plan runs on waits for the event behind them (meta_ev), then the
		 * streamed launches start.  (On a stream of their own with a host-side wait they took 15-25 ms whenever the DMA engines
		 * were busy with the pieces: round 4, every process but the first on a box.)  up_ev marks the end of the pieces: unstreamed
		 * classes of the plan, a repeated run after an abort and plan_destroy wait for it. */
		for (k = 0; k < p->nqd; ++k) {
			p->h_qd[k].need = p->d_order + p->norder + p->h_qd[k].unp_fmt; p->h_qd[k].unp_fmt = 0;
			p->h_qd[k].wm = (const uint32_t*)p->d_wm;
		}
		p->meta_ev = k2a_shim_event_create();
		/* ... from page-locked staging: copies the DMA engines do by themselves, like the pieces.  (From pageable memory the runtime
		 * stages them with the caller waiting; a memset is a kernel.  The result records need no clearing when every class of the plan
		 * is a queue class: k2a_finish writes all of a record.) */
		{
			const size_t b_pairs = align_up(sizeof(K2aPair) * (size_t)n, 256), b_order = align_up(sizeof(uint32_t) * ((size_t)p->norder + p->need_words), 256),
			             b_qd = sizeof(K2aQueueDesc) * (size_t)p->nqd;
			int all_queues = p->nqd > 0;
			for (k = 0; k < p->ncls; ++k) if (p->cls[k].qd < 0) all_queues = 0;
			/* (the clears are kernels: on the stream the plan first runs on, never on the shared upload stream -- a kernel there gets no
			 * wavefront slot while a device-filling launch waits for pieces queued behind it) */
			p->clear_res = !all_queues; p->clear_bnd = p->bnd_words > 0;
			p->h_meta = (uint8_t*)cache_get(BUF_HMETA, b_pairs + b_order + b_qd + 256, &p->cap[BUF_HMETA]);
			if (!p->h_meta) { fail(KSW2AMD_E_NOMEM, "pinned staging allocation failed: %s", k2a_shim_last_error()); goto err; }
			memcpy(p->h_meta, p->h_pairs, sizeof(K2aPair) * (size_t)n);
			memcpy(p->h_meta + b_pairs, p->h_order, sizeof(uint32_t) * ((size_t)p->norder + p->need_words));
			if (b_qd) memcpy(p->h_meta + b_pairs + b_order, p->h_qd, b_qd);
		pthread_mutex_lock(&su->mu);                        /* (the gather's workers issue pieces on the same stream: keep the order) */
		if (!p->meta_ev ||
		    k2a_shim_h2d(p->d_pairs, p->h_meta, sizeof(K2aPair) * (size_t)n, up) ||
		    k2a_shim_h2d(p->d_order, p->h_meta + b_pairs, sizeof(uint32_t) * ((size_t)p->norder + p->need_words), up) ||
		    (p->nqd && k2a_shim_h2d(p->d_wm + K2A_WM_BYTES, p->h_meta + b_pairs + b_order, b_qd, up)) ||
		    k2a_shim_event_record(p->meta_ev, up)) {
			pthread_mutex_unlock(&su->mu);
			fail(KSW2AMD_E_NODEVICE, "upload failed: %s", k2a_shim_last_error());
			goto err;
		}
		su->hold = su->np;
		pthread_mutex_unlock(&su->mu);
		}
		/* now the rest of the arena (what the gather has not completed yet follows as its workers get there) */
		stream_issue(su, -1);
		if (su->rc || !p->up_ev) { fail(KSW2AMD_E_NODEVICE, "upload failed: %s", k2a_shim_last_error()); goto err; }
		if (!p->gather && k2a_shim_event_record(p->up_ev, up)) { fail(KSW2AMD_E_NODEVICE, "upload failed: %s", k2a_shim_last_error()); goto err; }   /* (with a gather in flight: recorded by gather_wait) */
	} else {
		/* every other plan: the same staging and the same rule -- nobody waits here.  The small arrays leave from page-locked staging, an
		 * event behind them is what the stream the plan runs on waits for (ksw2amd_plan_run), and the creating thread goes on to the
		 * launches (a coalesced batch of single-pair calls: one host wait per batch, in fetch) or to packing its next chunk.  The result
		 * records are cleared only where something reads a record no kernel writes: the CIGAR compaction walks every pair of the plan,
		 * the invalid ones too (k2a_finish writes all of a record for every pair that is in a class). */
		const size_t b_pairs = align_up(sizeof(K2aPair) * (size_t)n, 256), b_meta = b_pairs + sizeof(uint32_t) * (size_t)p->norder;
		const size_t meta_off = align_up(p->seq_bytes, 256), tail = flat ? p->seq_bytes - flat_span : 0;
		const int need_clear = ninvalid > 0 && p->cig_words > 0;
		uint8_t *hm;
		if (flat) {
			/* pairs + task lists + the arena's padding and the matrices of a wide alphabet (flat_tmp), from one page-locked block */
			p->h_meta = (uint8_t*)cache_get(BUF_HMETA, b_meta + tail + 256, &p->cap[BUF_HMETA]);
			if (!p->h_meta) { fail(KSW2AMD_E_NOMEM, "pinned staging allocation failed: %s", k2a_shim_last_error()); goto err; }
			hm = p->h_meta;
			memcpy(hm + align_up(b_meta, 256), flat_tmp, tail);
		} else hm = p->h_seq + meta_off;
		if (b_meta + 256 > META_ROOM(n)) { fail(KSW2AMD_E_PARAM, "plan_create: task lists larger than planned%s", 0); goto err; }
		memcpy(hm, p->h_pairs, sizeof(K2aPair) * (size_t)n);
		memcpy(hm + b_pairs, p->h_order, sizeof(uint32_t) * (size_t)p->norder);
		p->up_ev = k2a_shim_event_create();
		p->clear_res = need_clear; p->clear_bnd = p->bnd_words > 0;   /* on the stream of the first run (see the streamed branch) */
		if (shared_up) pthread_mutex_lock(&g_shared_issue_mu);      /* one plan's copies in one piece */
		if (!p->up_ev ||
		    (flat && ((flat->on_device ? k2a_shim_d2d(p->d_seq, flat_lo, flat_span, up) : k2a_shim_h2d(p->d_seq, flat_lo, flat_span, up)) ||
		              k2a_shim_h2d(p->d_seq + flat_span, hm + align_up(b_meta, 256), tail, up) ||
		              k2a_shim_h2d(p->d_pairs, hm, b_meta, up))) ||
		    (!flat && k2a_shim_h2d(p->d_seq, p->h_seq, meta_off + b_meta, up)) ||
		    k2a_shim_event_record(p->up_ev, up)) {
			if (shared_up) pthread_mutex_unlock(&g_shared_issue_mu);
			fail(KSW2AMD_E_NODEVICE, "upload failed: %s", k2a_shim_last_error());
			goto err;
		}
		if (shared_up) pthread_mutex_unlock(&g_shared_issue_mu);
	}
	if (m > 5) for (k = 0; k < p->ncls; ++k) p->cls[k].sc.mat = (const int8_t*)p->d_seq + mat_off + (p->cls[k].generic ? (size_t)m * m : 0);
	free(pk_ok); free(solo_ok);
	p->flat_tail = flat_tmp; flat_tmp = 0;                          /* still being read by the upload */
	free(flat_tmp);
	plan_ready(p);                                                  /* the uploads are complete (or fenced by up_ev) */
	if (tlev) { const double t6 = now_ms(); char tmsg[96]; tmsg[0] = 0; if (su) snprintf(tmsg, sizeof(tmsg), "; streamed: %d pieces, %.3f ms in the upload calls so far", su->np, su->issue_ms); fprintf(stderr, "[ksw2_amd] plan_create n=%d: host arrays + arena layout %.3f, copy + classify %.3f, shape parity + demotions %.3f, sequence upload call + task lists %.3f, traceback layout %.3f, uploads + sync %.3f ms%s\n", n, tph[1] - tph[0], tph[2] - tph[1], tph[3] - tph[2], tph[4] - tph[3] , tph[5] - tph[4], t6 - tph[5], tmsg); }
	return p;
err:
	if (p && p->gather) gather_wait(p);
	if ((flat_tmp || su) && p && p->stream_used && p->stream) k2a_shim_stream_sync(p->stream);
	free(srt); free(pk_ok); free(solo_ok); free(flat_tmp);
	ksw2amd_plan_destroy(p);
	return 0;
}

/* ---------------------------------------------------------------- uniform batches
 * A score-only batch whose pairs all have one shape and one set of parameters (config 2; the 10 k headline) is the case where the
 * host's per-pair work is pure overhead: plan_create_ex walks the batch half a dozen times on ONE thread -- offsets, classification,
 * task list, piece counts, a copy of the records into page-locked staging -- about 40 ns per pair, 2.5 ms for config 2's 65 536 pairs
 * next to a 1.3 ms kernel (KSW2AMD_TRACE=2, round 5), which is why such batches went through eight chunk plans on six workers
 * instead of one streamed launch.  Here nothing per pair is left on the creating thread:
 *   - the class comes from plan_create_ex itself, asked about TWO of the pairs as if they were n (g_probe_tasks): one set of rules;
 *   - the arena is laid out by rule (pair i at i * stride) and the device writes its own records, task list and piece counts
 *     (K2aUniform, k2a_uniform_layout_kernel, launched by ksw2amd_plan_run in front of the fill): nothing of them is uploaded;
 *   - the pool's threads copy the sequences piece by piece into page-locked staging, as for every streamed plan, and fill the host's
 *     per-pair arrays of their ranges on the way (what a fetch looks at: class, flags, offsets for a re-run).
 * Returns NULL with g_err empty when the batch is not of this kind (the caller takes the general path), NULL with a message on a
 * real failure.  KSW2AMD_UNIFORM=0: never.  The caller guarantees that every pair equals pairs[0] in qlen, tlen, w, zdrop,
 * end_bonus and flag and that none is empty. */
#define K2A_UNI_MIN_PAIRS 2048
void uni_fill_range(const K2aUniform *u, K2aPair *hp, int8_t *h_cls, int32_t *h_flag, uint32_t *h_order, int cls0, int flag0, int beg, int end)
{
	int i;
	for (i = beg; i < end; ++i) {
		K2aPair d = u->tmpl;
		d.qoff = (uint32_t)i * u->stride; d.toff = (uint32_t)i * u->stride + u->qpad;
		if (u->defer) d.tb_off = u->blk_base + (uint64_t)(((uint32_t)i >> 1) / u->ng) * u->blk_bytes;
		hp[i] = d; h_cls[i] = (int8_t)cls0; h_flag[i] = flag0; h_order[i] = (uint32_t)i;
	}
}

static ksw2amd_plan_t *plan_create_uniform(int dual, int scalar, const ksw2amd_scoring_t *sc, int n, const ksw2amd_pair_t *pairs)
{
	ksw2amd_plan_t *t, *p = 0;
	stream_up_t *su = 0;
	K2aUniform *u = 0;
	cls_t c0;
	K2aPair tm;
	int cls0, flag0, k, G, C, NG, nwt, mx, w, wire4;
	size_t pbytes, ppp;
	uint8_t *d_pk = 0;
	const char *ev = ENV(UNIFORM);
	g_err[0] = 0;
	if ((ev && atoi(ev) == 0) || n < K2A_UNI_MIN_PAIRS || (n & 1) || !pairs || !sc || sc->m > 5 || sc->m <= 0 || !sc->mat) return 0;
	if (!(pairs[0].flag & KSW_EZ_SCORE_ONLY) || pairs[0].qlen <= 0 || pairs[0].tlen <= 0 || !pairs[0].query || !pairs[0].target) return 0;
	if (stream_env() == 0 || g_no_defer || g_probe_tasks || 0) return 0;
	/* the class, by the rules of every plan */
	g_probe_tasks = n / 2;
	t = plan_create_ex(dual, scalar, sc, 2, pairs, 0, 0);
	g_probe_tasks = 0;
	if (!t) return 0;                                            /* (a real failure: g_err says which) */
	if (t->reject_all || t->ncls != 1 || !t->cls[0].pk || t->cls[0].solo || t->cls[0].cfg == K2A_PKCFG_MP || t->cls[0].mode != K2A_MODE_SCORE || t->cls[0].count != 1) {
		ksw2amd_plan_destroy(t); g_err[0] = 0; return 0;
	}
	c0 = t->cls[0]; tm = t->h_pairs[0]; cls0 = t->h_cls[0]; flag0 = t->h_flag[0];
	p = plan_new("plan_create", n, -1);
	if (!p) { ksw2amd_plan_destroy(t); return 0; }
	p->dual = !!dual; p->m = sc->m; p->scalar = scalar;
	p->src_mat = t->src_mat; t->src_mat = 0;
	p->src_sc = *sc; p->src_sc.mat = p->src_mat;
	ksw2amd_plan_destroy(t);
	p->h_cls[n] = -1; memset(&p->h_pairs[n], 0, sizeof(K2aPair)); p->h_half[n] = 0;
	u = (K2aUniform*)calloc(1, sizeof(*u));
	p->h_order = (uint32_t*)malloc(sizeof(uint32_t) * ((size_t)n + 4));
	if (!u || !p->h_order) { free(u); fail(KSW2AMD_E_NOMEM, "plan_create: host allocation failed%s", 0); goto err; }
	p->uni = u;
	mx = imax(pairs[0].qlen, pairs[0].tlen); w = pairs[0].w; if (w < 0 || w > mx) w = mx;
	p->cells = (int64_t)n * band_cells(pairs[0].qlen, pairs[0].tlen, w);
	/* the arena: query 16-aligned, target 16-aligned and readable one strip past its end, like every gathered arena */
	u->n = (uint32_t)n; u->ntasks = (uint32_t)(n / 2);
	/* the wire format (KSW2AMD_WIRE4=0: none; KSW2AMD_WIRE2=1: two bits per code): staging and upload hold two or four residue codes
	 * per byte -- a half or a quarter of the bytes for the gather to write and for the DMA engines to move -- and every wavefront-task
	 * expands its own pairs into the arena before it reads them (K2aQueueDesc.unp_*, k2a_queue_wait).  The pieces keep their pair
	 * boundaries; all offsets of the upload side shift.  Two bits per code (round 6): codes above 3 travel as escape entries in the
	 * last upload bytes of the pair's region (ksw2_lane.h, K2A_WIRE2_*), for which the target's padding grows by K2A_WIRE2_PAD.
	 * OPT-IN: on config 2 it moved the step by +3 % inside a +-7 % spread (profiles/r6_wire2_ab_ssec_rows.txt: what the step waits
	 * for after round 5's 4-bit format is the gather and the last task's latency, no longer the link), and a pair with more wildcard
	 * runs than its slot holds sends the whole batch to the general path -- a cliff the 4-bit format only has for codes above 15. */
	wire4 = (ENV(WIRE4) && atoi(ENV(WIRE4)) == 0) ? 0 : (ENV(WIRE2) && atoi(ENV(WIRE2)) == 1) ? 2 : 1;
	u->qpad = (uint32_t)align_up((size_t)pairs[0].qlen, 16); u->stride = u->qpad + (uint32_t)align_up((size_t)pairs[0].tlen + 64 + (wire4 == 2 ? K2A_WIRE2_PAD : 0), 16);
	if (wire4 == 2 && u->stride >= (1u << 20)) { wire4 = 1; u->stride = u->qpad + (uint32_t)align_up((size_t)pairs[0].tlen + 64, 16); }      /* (an escape's offset has 20 bits) */
	if ((uint64_t)n * u->stride > 0xfff00000u - 65536u) { g_err[0] = 0; goto na; }
	p->seq_bytes = align_up((size_t)n * u->stride + 65536, 256);
	u->seq_bytes = p->seq_bytes; u->margin = K2A_STREAM_MARGIN;
	tm.cig_off = 0; tm.tb_off = 0; tm.bnd_off = 0; tm.pad = 0;
	/* the one class */
	G = k2a_pkcfg_G[c0.cfg]; C = k2a_pkcfg_C[c0.cfg]; NG = 64 / G; nwt = (n / 2 + NG - 1) / NG;
	u->ng = (uint32_t)NG;
	p->ncls = 1; p->cls[0] = c0; p->cls[0].first = 0; p->cls[0].count = n / 2; p->cls[0].qd = 0; p->ntasks = n / 2; p->norder = n;
	if (c0.defer) {                                              /* checkpoint blocks, one per wavefront-task (plan_create_ex, "Deferred arg-max") */
		const uint32_t ns = (uint32_t)((tm.tlen + C - 1) / C), steps = ns - 1 + (uint32_t)imin(tm.qlen - 1, tm.tlen - 1 + tm.w) + 1;
		const size_t bytes = align_up((size_t)steps * 512 + (size_t)NG * ns * 16, 256), zl = align_up(4 * K2A_ZLIST_WORDS(n / 2), 256);
		size_t free_b = 0, total_b = 0;
		if (zl + (size_t)nwt * bytes > ((size_t)1 << 30) && !(ENV(DEFER) && *ENV(DEFER)) &&
		    (k2a_shim_mem_info(&free_b, &total_b) || zl + (size_t)nwt * bytes > (free_b + thread_cached_device_bytes()) / 10 * 6)) { g_err[0] = 0; goto na; }   /* (the general path decides what then) */
		u->defer = 1; u->blk_base = zl; u->blk_bytes = bytes;
		tm.bnd_off = steps; tm.cig_off = ns;
		p->tb_bytes = zl + (size_t)nwt * bytes;
	}
	u->tmpl = tm;
	/* page-locked staging, device arena, watermark: as every streamed plan */
	p->h_seq = (uint8_t*)cache_get(BUF_HSEQ, (p->seq_bytes >> wire4) + 256, &p->cap[BUF_HSEQ]);
	su = (stream_up_t*)calloc(1, sizeof(*su));
	if (!p->h_seq || !su) { free(su); su = 0; fail(KSW2AMD_E_NOMEM, "pinned staging allocation failed: %s", k2a_shim_last_error()); goto err; }
	p->up_state = su;
	su->t0 = now_ms();
	pthread_mutex_init(&su->mu, 0);
	{
		const char *pk_ = ENV(STREAM_PIECE_KB);
		pbytes = pk_ && atol(pk_) > 0 ? (size_t)atol(pk_) << 10 : p->seq_bytes / 24;
		if (!(pk_ && atol(pk_) > 0)) { if (pbytes < ((size_t)1 << 20)) pbytes = (size_t)1 << 20; if (pbytes > ((size_t)32 << 20)) pbytes = (size_t)32 << 20; }
		if (wire4 && pbytes < ((size_t)K2A_WM_BYTES << wire4)) pbytes = (size_t)K2A_WM_BYTES << wire4;      /* a half / a quarter of it travels: never a copy so small that the runtime moves it with a kernel (K2A_WM_BYTES) */
		if (pbytes * (K2A_MAXPIECES - 1) < p->seq_bytes) pbytes = p->seq_bytes / (K2A_MAXPIECES - 1) + 1;
		ppp = (pbytes + u->stride - 1) / u->stride;              /* pairs per piece: pieces start at pair boundaries (the copy's work units) */
		if (ppp < 1) ppp = 1;
		while (((size_t)n + ppp - 1) / ppp > K2A_MAXPIECES) ++ppp;
	}
	su->np = (int)(((size_t)n + ppp - 1) / ppp);
	for (k = 0; k <= su->np; ++k) {
		const size_t first = (size_t)k * ppp < (size_t)n ? (size_t)k * ppp : (size_t)n;
		su->pfirst[k] = (int)first; su->pb[k] = k < su->np ? first * u->stride : p->seq_bytes;
		u->pb[k] = su->pb[k];
		su->pb[k] >>= wire4;                                      /* (the device counts pieces in arena offsets, the upload moves a half / a quarter of them) */
	}
	su->wire4 = wire4; su->wire_stride = u->stride;
	u->npieces = (uint32_t)su->np;
	su->wm_src = 0;
	{
		extern const uint32_t *k2a_wm_source(void);
		su->wm_src = k2a_wm_source();
	}
	su->up = shared_upload_stream();
	su->up2 = stream_lanes(0) == 2 ? shared_upload_stream2() : 0;
	su->fault = env_flag(ENV(STREAM_FAULT), 0); su->sleep_us = ENV(STREAM_SLEEP_US) ? atoi(ENV(STREAM_SLEEP_US)) : 0;
	su->src = p->h_seq; su->src_bytes = p->seq_bytes >> wire4;
	p->d_seq = (uint8_t*)cache_get(BUF_SEQ, p->seq_bytes, &p->cap[BUF_SEQ]);
	if (wire4) d_pk = (uint8_t*)cache_get(BUF_PK4, p->seq_bytes >> wire4, &p->cap[BUF_PK4]);
	p->d_wm = (uint8_t*)cache_get(BUF_WM, K2A_WM_BYTES + NCLS_ENTRIES * sizeof(K2aQueueDesc), &p->cap[BUF_WM]);
	p->d_pairs = (K2aPair*)cache_get(BUF_PAIRS, sizeof(K2aPair) * ((size_t)n + 1), &p->cap[BUF_PAIRS]);
	p->d_order = (uint32_t*)cache_get(BUF_ORDER, sizeof(uint32_t) * ((size_t)n + (size_t)nwt + 1), &p->cap[BUF_ORDER]);
	p->d_res = (K2aResult*)cache_get(BUF_RES, sizeof(K2aResult) * ((size_t)n + 1), &p->cap[BUF_RES]);
	p->d_tb = p->tb_bytes ? (uint8_t*)cache_get(BUF_TB, p->tb_bytes, &p->cap[BUF_TB]) : 0;
	p->h_meta = (uint8_t*)cache_get(BUF_HMETA, 256, &p->cap[BUF_HMETA]);
	p->h_qd = (K2aQueueDesc*)calloc(1, sizeof(K2aQueueDesc));
	p->wm_ev = k2a_shim_event_create(); p->meta_ev = k2a_shim_event_create(); p->up_ev = k2a_shim_event_create();
	if (!su->wm_src || !su->up || !p->d_seq || !p->d_wm || !p->d_pairs || !p->d_order || !p->d_res || (p->tb_bytes && !p->d_tb) || !p->h_meta || !p->h_qd ||
	    !p->wm_ev || !p->meta_ev || !p->up_ev || (wire4 && !d_pk)) { fail(KSW2AMD_E_NOMEM, "device allocation failed: %s", k2a_shim_last_error()); goto err; }
	su->d_seq = wire4 ? d_pk : p->d_seq; su->d_wm = p->d_wm;
	p->d_pk4 = d_pk;
	p->unscanned = 1; p->npieces = su->np; p->stream = su->up; p->stream_used = 1;
	/* the launch's descriptor: the one thing that is uploaded besides the sequences (64 bytes, behind the zeroed watermark) */
	{
		const char *te = ENV(STREAM_TIMEOUT_MS);
		p->nqd = 1; p->need_words = (size_t)nwt;
		p->h_qd[0].nwt = (uint32_t)nwt; p->h_qd[0].timeout_ticks = (uint64_t)(te && atoi(te) > 0 ? atoi(te) : 2000) * 100000u;
		p->h_qd[0].need = p->d_order + p->norder; p->h_qd[0].wm = (const uint32_t*)p->d_wm;
		if (wire4) {
			p->h_qd[0].unp_src = d_pk; p->h_qd[0].unp_dst = p->d_seq;
			p->h_qd[0].unp_bytes = 2u * (uint32_t)NG * u->stride; p->h_qd[0].unp_total = (uint32_t)n * u->stride;
			p->h_qd[0].unp_fmt = ((uint32_t)wire4 << 30) | u->stride;
		}
		memcpy(p->h_meta, p->h_qd, sizeof(K2aQueueDesc));
	}
	if (k2a_shim_h2d(p->d_wm, (const uint8_t*)su->wm_src + (size_t)K2A_MAXPIECES * K2A_WM_BYTES, K2A_WM_BYTES, su->up) || k2a_shim_event_record(p->wm_ev, su->up) ||
	    k2a_shim_h2d(p->d_wm + K2A_WM_BYTES, p->h_meta, sizeof(K2aQueueDesc), su->up) || k2a_shim_event_record(p->meta_ev, su->up)) {
		fail(KSW2AMD_E_NODEVICE, "upload failed: %s", k2a_shim_last_error()); goto err;
	}
	su->hold = su->np;                                            /* nothing else has to get in front of the pieces */
	p->streamed = 1;
	/* the copy (and the host's per-pair arrays) on the pool's threads, piece by piece, each piece issued as it completes */
	{
		extern int gather_start_uniform(ksw2amd_plan_t *p, stream_up_t *su, const ksw2amd_pair_t *pairs, int n, int cls0, int flag0);
		if (gather_start_uniform(p, su, pairs, n, cls0, flag0)) {      /* the pool cannot take it: here, piece by piece */
			copy_ctx_t cc = { 0 };
			memset(&cc, 0, sizeof(cc));
			cc.h_seq = p->h_seq; cc.hp = p->h_pairs; cc.pairs = pairs; cc.su = su;
			for (k = 0; k < su->np; ++k) {
				extern void copy_or_pack_range(const copy_ctx_t *c, int beg, int end);
				uni_fill_range(u, p->h_pairs, p->h_cls, p->h_flag, p->h_order, cls0, flag0, su->pfirst[k], su->pfirst[k + 1]);
				copy_or_pack_range(&cc, su->pfirst[k], su->pfirst[k + 1]);
				stream_issue(su, k);
			}
			if (su->rc || k2a_shim_event_record(p->up_ev, su->up)) { fail(KSW2AMD_E_NODEVICE, "upload failed: %s", k2a_shim_last_error()); goto err; }
		}
	}
	plan_ready(p);
	p->stream = su->up; p->stream_used = 1;
	return p;
na:
err:
	if (p && p->gather) gather_wait(p);
	if (p && p->stream_used && p->stream) k2a_shim_stream_sync(p->stream);
	{
		char keep[sizeof(g_err)];
		memcpy(keep, g_err, sizeof(keep));
		ksw2amd_plan_destroy(p);
		memcpy(g_err, keep, sizeof(keep));
	}
	return 0;
}
ksw2amd_plan_t *plan_create_uniform_entry(int dual, int scalar, const ksw2amd_scoring_t *sc, int n, const ksw2amd_pair_t *pairs) { return plan_create_uniform(dual, scalar, sc, n, pairs); }

ksw2amd_plan_t *ksw2amd_plan_create(int dual, const ksw2amd_scoring_t *sc, int n, const ksw2amd_pair_t *pairs)
{
	/* (KSW2AMD_STREAM=1 streams these too: tests, A/B runs.)  create = pack + upload: the caller may free its inputs on return, so a
	 * gather still reading them on the pool's threads is waited for here (the batch entries call plan_create_ex and keep the overlap) */
	ksw2amd_plan_t *p = plan_create_ex(dual, 0, sc, n, pairs, 0, 0);
	if (p && p->gather && gather_wait(p)) {              /* a piece's copy or upload failed: gather_wait is the only consumer of that verdict (it clears p->gather) */
		fail(KSW2AMD_E_NODEVICE, "plan_create: upload failed: %s", k2a_shim_last_error());
		if (p->stream_used && p->stream) k2a_shim_stream_sync(p->stream);
		{
			char keep[sizeof(g_err)];
			memcpy(keep, g_err, sizeof(keep));
			ksw2amd_plan_destroy(p);
			memcpy(g_err, keep, sizeof(keep));
		}
		return 0;
	}
	return p;
}


int ksw2amd_plan_run(ksw2amd_plan_t *p, void *stream)
{
	int c, streaming = 0, nrest = 0;
	if (!p) return fail(KSW2AMD_E_PARAM, "plan_run: NULL plan%s", 0);
	if (p->splice == 3) return ssec_plan_run(p, stream);
	if (p->splice == 2) return extf_plan_run(p, stream);
	if (p->splice) return exts_plan_run(p, stream);
	p->stream = stream; p->ran = 1; p->stream_used = 1;
	if (p->gather && (!k2a_shim_async_launches() || !p->streamed) && gather_wait(p))      /* an ordinary launch, or one that runs inside the call, needs the whole arena */
		return fail(KSW2AMD_E_NODEVICE, "plan_run: upload failed: %s", k2a_shim_last_error());
	streaming = p->streamed && p->nqd > 0;
	if (p->up_ev && !streaming && k2a_shim_stream_wait_event(stream, p->up_ev)) goto err;      /* the plan's upload (shared stream) before its kernels */
	if (p->reject_all || p->ntasks == 0) return KSW2AMD_OK;
	/* first run: the records no kernel writes (invalid pairs, read by the CIGAR compaction) and the boundary scratch's -inf pattern */
	if ((p->clear_res && k2a_shim_memset(p->d_res, 0, sizeof(K2aResult) * (size_t)p->n, stream)) ||
	    (p->clear_bnd && k2a_shim_memset(p->d_bnd, 0xC0, p->bnd_words * 4, stream))) goto err;
	p->clear_res = p->clear_bnd = 0;
	/* uniform plans: the records, the task list and the piece counts are written on the device by rule (K2aUniform), in front of the
	 * launches that read them; nothing of them was built or uploaded by the host */
	if (p->uni && k2a_shim_launch_uniform_layout(p->uni, p->d_pairs, p->d_order, p->nqd ? p->d_order + p->norder : 0, stream)) goto err;
	if (k2a_shim_event_record(p->ev[0], stream)) goto err;
	if (streaming && ((p->wm_ev && k2a_shim_stream_wait_event(stream, p->wm_ev)) || (p->meta_ev && k2a_shim_stream_wait_event(stream, p->meta_ev)))) goto err;
	if (streaming) {
		/* streamed plan: every score-only packed class as ONE launch that starts now, under the upload, each wavefront waiting for the
		 * pieces of its own task (k2a_queue_wait); then, behind the whole upload, whatever else the plan holds */
		K2aQueueDesc *d_qd = (K2aQueueDesc*)(p->d_wm + K2A_WM_BYTES);
		__sync_fetch_and_add(&g_stream_stat[0], 1);
		for (c = 0; c < p->ncls; ++c) {
			const cls_t *k = &p->cls[c];
			if (k->qd < 0) { ++nrest; continue; }
			if (k2a_shim_memset(d_qd + k->qd, 0, 8, stream) ||           /* next = abort = 0 */
			    k2a_shim_launch_fill_pk(k->cfg, p->dual, k->mode, k->rb, k->nomax, k->defer, &k->sc, p->d_pairs, p->d_order + k->first, k->count, p->d_seq,
			                            p->d_tb, p->d_res, d_qd + k->qd, stream)) goto err;
			if (k->mode != K2A_MODE_SCORE &&
			    k2a_shim_launch_trace_pk(k->cfg, p->dual, p->d_pairs, p->d_order + k->first, k->count, p->d_tb, p->d_res, p->d_cig, stream)) goto err;
		}
		if (nrest == 0) {
			if (k2a_shim_event_record(p->ev[1], stream) || k2a_shim_event_record(p->ev[2], stream)) goto err;
			return KSW2AMD_OK;
		}
		/* the plan's other classes are ordinary launches: behind the WHOLE arena.  The event that marks its end is recorded when the
		 * gather's last piece has been issued (gather_wait) -- waiting for an event nobody has recorded yet is no wait at all, and the
		 * int32 / solo classes of a mixed plan then read sequences that are not there (found by the fuzz script's streamed entries) */
		if (p->gather && gather_wait(p)) goto err;
		if (p->up_ev && k2a_shim_stream_wait_event(stream, p->up_ev)) goto err;
	} else nrest = p->ncls;
	if (nrest > 1 && !ENV(SERIAL) && side_streams() == 0) {
		/* several classes: fork them over the caller's stream and the side streams (fill, then that class's traceback, in
		 * stream order), join on the caller's stream.  The fill / traceback split of plan_timing is then meaningless:
		 * both report the whole run (KSW2AMD_SERIAL=1 restores the two-phase order for profiling). */
		int used = 0, ord[NCLS_ENTRIES], x, y;
		/* small classes first: their few wavefronts get their slots at once and run beside the big launches instead of after them */
		int nord = 0;
		for (c = 0; c < p->ncls; ++c) if (!(streaming && p->cls[c].qd >= 0)) ord[nord++] = c;      /* (streamed classes are running already) */
		for (x = 1; x < nord; ++x)
			for (y = x; y > 0 && p->cls[ord[y]].count < p->cls[ord[y - 1]].count; --y) { const int t = ord[y]; ord[y] = ord[y - 1]; ord[y - 1] = t; }
		if (k2a_shim_event_record(g_side_ev[NSIDE], stream)) goto err;
		for (x = 0; x < nord; ++x) {
			const cls_t *k = &p->cls[ord[x]];
			const int lane = x % (NSIDE + 1);
			void *s = lane == 0 ? stream : g_side[lane - 1];
			if (lane > 0 && !(used & (1 << lane))) { if (k2a_shim_stream_wait_event(s, g_side_ev[NSIDE])) goto err; used |= 1 << lane; }
			if (k->solo) {
				if (k2a_shim_launch_fill_solo(p->dual, k->mode, &k->sc, p->d_pairs, p->d_order + k->first, k->count, p->d_seq, p->d_tb, p->d_res, s)) goto err;
				if (k->mode != K2A_MODE_SCORE &&
				    k2a_shim_launch_trace_solo(p->d_pairs, p->d_order + k->first, k->count, p->d_tb, p->d_res, p->d_cig, s)) goto err;
			} else if (k->pk) {
				if (k->cfg == K2A_PKCFG_MP ? k2a_shim_launch_fill_pkmp(p->dual, k->mode, &k->sc, p->d_pairs, p->d_order + k->first, k->count, p->d_seq, p->d_tb,
				                                                     (uint32_t*)p->d_bnd, p->d_res, s)
				    : k2a_shim_launch_fill_pk(k->cfg, p->dual, k->mode, k->rb, k->nomax, k->defer, &k->sc, p->d_pairs, p->d_order + k->first, k->count, p->d_seq,
				                            p->d_tb, p->d_res, 0, s)) goto err;
				if (k->mode != K2A_MODE_SCORE &&
				    k2a_shim_launch_trace_pk(k->cfg, p->dual, p->d_pairs, p->d_order + k->first, k->count, p->d_tb, p->d_res, p->d_cig, s)) goto err;
			} else {
				if (k2a_shim_launch_fill(k->cfg, p->dual, k->mode, &k->sc, p->d_pairs, p->d_order + k->first, k->count, p->d_seq, p->d_tb,
				                         p->d_bnd, p->d_res, s)) goto err;
				if (k->mode != K2A_MODE_SCORE &&
				    k2a_shim_launch_trace(k->cfg, p->dual, p->d_pairs, p->d_order + k->first, k->count, p->d_tb, p->d_res, p->d_cig, s)) goto err;
			}
		}
		for (c = 1; c <= NSIDE; ++c)
			if (used & (1 << c)) {
				if (k2a_shim_event_record(g_side_ev[c - 1], g_side[c - 1]) || k2a_shim_stream_wait_event(stream, g_side_ev[c - 1])) goto err;
			}
		if (k2a_shim_event_record(p->ev[1], stream) || k2a_shim_event_record(p->ev[2], stream)) goto err;
		return KSW2AMD_OK;
	}
	for (c = 0; c < p->ncls; ++c) {
		const cls_t *k = &p->cls[c];
		if (streaming && k->qd >= 0) continue;
		if (k->solo) {
			if (k2a_shim_launch_fill_solo(p->dual, k->mode, &k->sc, p->d_pairs, p->d_order + k->first, k->count, p->d_seq, p->d_tb, p->d_res, stream)) goto err;
		} else if (k->pk) {
			if (k->cfg == K2A_PKCFG_MP ? k2a_shim_launch_fill_pkmp(p->dual, k->mode, &k->sc, p->d_pairs, p->d_order + k->first, k->count, p->d_seq, p->d_tb,
			                                                     (uint32_t*)p->d_bnd, p->d_res, stream)
			    : k2a_shim_launch_fill_pk(k->cfg, p->dual, k->mode, k->rb, k->nomax, k->defer, &k->sc, p->d_pairs, p->d_order + k->first, k->count, p->d_seq, p->d_tb,
			                            p->d_res, 0, stream)) goto err;
		} else if (k2a_shim_launch_fill(k->cfg, p->dual, k->mode, &k->sc, p->d_pairs, p->d_order + k->first, k->count, p->d_seq, p->d_tb,
		                                p->d_bnd, p->d_res, stream)) goto err;
	}
	if (k2a_shim_event_record(p->ev[1], stream)) goto err;
	for (c = 0; c < p->ncls; ++c) {
		const cls_t *k = &p->cls[c];
		if (k->mode == K2A_MODE_SCORE || (streaming && k->qd >= 0)) continue;
		if (k->solo) {
			if (k2a_shim_launch_trace_solo(p->d_pairs, p->d_order + k->first, k->count, p->d_tb, p->d_res, p->d_cig, stream)) goto err;
		} else if (k->pk) {
			if (k2a_shim_launch_trace_pk(k->cfg, p->dual, p->d_pairs, p->d_order + k->first, k->count, p->d_tb, p->d_res, p->d_cig, stream)) goto err;
		} else if (k2a_shim_launch_trace(k->cfg, p->dual, p->d_pairs, p->d_order + k->first, k->count, p->d_tb, p->d_res, p->d_cig, stream)) goto err;
	}
	if (k2a_shim_event_record(p->ev[2], stream)) goto err;
	return KSW2AMD_OK;
err:
	return fail(KSW2AMD_E_NODEVICE, "plan_run: %s", k2a_shim_last_error());
}

int ksw2amd_plan_timing(ksw2amd_plan_t *p, float *fill_ms, float *total_ms)
{
	if (!p || !p->ran) return fail(KSW2AMD_E_PARAM, "plan_timing: plan has not run%s", 0);
	if (p->reject_all || p->ntasks == 0) { if (fill_ms) *fill_ms = 0; if (total_ms) *total_ms = 0; return KSW2AMD_OK; }
	if (fill_ms) *fill_ms = k2a_shim_event_ms(p->ev[0], p->ev[1]);
	if (total_ms) *total_ms = k2a_shim_event_ms(p->ev[0], p->ev[2]);
	return KSW2AMD_OK;
}

int64_t ksw2amd_plan_cells(const ksw2amd_plan_t *p) { return p ? p->cells : 0; }
int64_t ksw2amd_plan_packed_pairs(const ksw2amd_plan_t *p)
{
	int64_t n = 0;
	int c, i;
	if (!p) return 0;
	for (c = 0; c < p->ncls; ++c)
		if (p->cls[c].pk)
			for (i = 0; i < p->cls[c].count; ++i)
				n += p->h_order[p->cls[c].first + 2 * i] == p->h_order[p->cls[c].first + 2 * i + 1] ? 1 : 2;
		else if (p->cls[c].solo) n += p->cls[c].count;
	return n;
}
/* one line per kernel class of an extz / extd plan, as the next ksw2amd_plan_run would launch it */
int ksw2amd_plan_describe(const ksw2amd_plan_t *p, char *buf, int cap)
{
	static const char *const mode_name[3] = { "score", "left", "right" }, *const form_name[4] = { "registers", "ldsrows", "ldscodes", "defer" };
	int c, len = 0;
	if (!p || !buf || cap <= 0) return 0;
	buf[0] = 0;
	if (p->splice == 2 && !p->reject_all) {          /* ksw_extf2_sse plans: one line per kernel class in use */
		static const char *const fkind[10] = { "extf-lds", "extf-lds", "extf-lds", "extf-hbm", "extf-win4", "extf-win8", "extf-lane", "extf-grp", "extf-grp32", "extf-grp64" };
		int nl = 0;
		for (c = 0; c < 10 && len < cap - 1; ++c)
			if (p->f_count[c]) {
				len += snprintf(buf + len, (size_t)(cap - len), "kernel=%s form=%s ring=%d tasks=%d\n", fkind[c], c == 6 && p->f_par.ring ? "ldsring" : c == 6 ? "hbm" : "-",
				                c == 6 ? p->f_par.ring : 0, p->f_count[c]);
				++nl;
			}
		return nl;
	}
	if (p->splice == 1 && !p->reject_all) {          /* ksw_exts2_sse plans: register windows of 8 / 16 slots, or the HBM-state kernel */
		static const char *const skind[3] = { "exts-win8", "exts-win16", "exts-hbm" };
		int nl = 0, mode, g, wn;
		for (mode = 0; mode < 3; ++mode) for (g = 0; g < 2; ++g) for (wn = 0; wn < 3 && len < cap - 1; ++wn)
			if (p->s_count[mode][g][wn]) {
				len += snprintf(buf + len, (size_t)(cap - len), "kernel=%s mode=%s generic=%d tasks=%d\n", skind[wn], mode_name[mode], g, p->s_count[mode][g][wn]);
				++nl;
			}
		return nl;
	}
	if (p->splice == 3 && !p->reject_all) {          /* SSE-compatible plans: state arrays in HBM scratch, in LDS or in registers */
		static const char *const form_name[3] = { "hbm", "lds", "blk" };
		int nl = 0, mode, lds;
		for (mode = 0; mode < 3; ++mode) for (lds = 0; lds < 3 && len < cap - 1; ++lds)
			if (p->s_count[mode][0][lds]) {
				len += snprintf(buf + len, (size_t)(cap - len), "kernel=ssec gaps=%d mode=%s form=%s tasks=%d\n", p->dual ? 2 : 1, mode_name[mode], form_name[lds], p->s_count[mode][0][lds]);
				++nl;
			}
		return nl;
	}
	if (p->splice || p->reject_all) return 0;
	for (c = 0; c < p->ncls && len < cap - 1; ++c) {
		const cls_t *k = &p->cls[c];
		const char *kind = k->solo ? "solo" : k->pk ? (k->cfg == K2A_PKCFG_MP ? "pkmp" : "pk") : (k->cfg == K2A_CFG_MP ? "mp" : "int32");
		const int G = k->solo ? 64 : k->pk ? k2a_pkcfg_G[k->cfg] : k2a_cfg_G[k->cfg], C = k->solo ? 2 * K2A_SOLO_ROWS(k->mode == K2A_MODE_SCORE) : k->pk ? k2a_pkcfg_C[k->cfg] : k2a_cfg_C[k->cfg];
		const int form = k->defer ? 3 : k->solo ? 0 : k->pk ? (k->cfg == K2A_PKCFG_MP ? 0 : k2a_shim_pk_form(k->cfg, p->dual, k->mode, k->nomax, k->count))
		                                     : (k->cfg == K2A_CFG_MP ? k2a_shim_mp_form(p->dual, k->mode, k->count) : 0);
		len += snprintf(buf + len, (size_t)(cap - len), "kernel=%s G=%d C=%d gaps=%d mode=%s rebased=%d nomax=%d generic=%d form=%s tasks=%d uniform=%d wire4=%d tn=%d\n",
		                kind, G, C, p->dual ? 2 : 1, mode_name[k->mode], k->rb, k->nomax, k->generic, form_name[form], k->count, p->uni ? 1 : 0,
		                p->up_state ? p->up_state->wire4 : 0, (k->pk || k->solo) && k->sc.pk_tn1 ? 1 : 0);      /* tn: target wildcards are rows of this packed class (K2aScoring.pk_tn1) */
	}
	return p->ncls;
}
int64_t ksw2amd_plan_device_bytes(const ksw2amd_plan_t *p)
{
	if (!p) return 0;
	return (int64_t)(p->seq_bytes + p->tb_bytes + p->cig_words * 4 + p->bnd_words * 4 + (sizeof(K2aPair) + sizeof(K2aResult)) * (size_t)p->n + 4 * (size_t)p->norder);
}

static int fetch_results(ksw2amd_plan_t *p)
{
	if (!p || !p->ran) return fail(KSW2AMD_E_PARAM, "plan_fetch: plan has not run%s", 0);
	if (p->gather && gather_wait(p)) return fail(KSW2AMD_E_NODEVICE, "plan_fetch: upload failed: %s", k2a_shim_last_error());
	if (p->reject_all || p->ntasks == 0) return KSW2AMD_OK;
	if (p->streamed && p->nqd > 0) {
		if (k2a_shim_stream_sync(p->stream)) return fail(KSW2AMD_E_NODEVICE, "plan_fetch: %s", k2a_shim_last_error());
		/* did every streamed launch get its inputs?  A wavefront that waited longer than the launch's timeout raised `abort` and the
		 * queue was left unfinished: wait for the upload, then run the whole plan again the ordinary way (bounded, never a hang) */
		K2aQueueDesc back[NCLS_ENTRIES];
		int k, aborted = 0;
		if (k2a_shim_d2h(back, p->d_wm + K2A_WM_BYTES, sizeof(K2aQueueDesc) * (size_t)p->nqd, p->stream) || k2a_shim_stream_sync(p->stream))
			return fail(KSW2AMD_E_NODEVICE, "plan_fetch: %s", k2a_shim_last_error());
		for (k = 0; k < p->nqd; ++k) aborted |= back[k].abort != 0 || back[k].next < back[k].nwt;
		if (aborted) {
			void *st = p->stream;
			__sync_fetch_and_add(&g_stream_stat[1], 1);
			if (trace_level()) fprintf(stderr, "[ksw2_amd] streamed plan n=%d: a launch gave up waiting for its inputs (abort %u, started %u of %u wavefront-tasks, %d pieces); running the plan again behind its upload\n", p->n, back[0].abort, back[0].next, back[0].nwt, p->npieces);
			p->streamed = 0;
			/* (4-bit wire format: the wavefront-tasks that never started have not expanded their pairs -- the whole arena, now) */
			if ((p->up_ev && k2a_shim_event_sync(p->up_ev)) || k2a_shim_memset(p->d_res, 0, sizeof(K2aResult) * (size_t)p->n, st) ||
			    (plan_wire4(p) && k2a_shim_launch_wire_expand(p->d_pk4, p->d_seq, (size_t)p->uni->n * p->uni->stride, plan_wire4(p), p->uni->stride, st)) ||
			    ksw2amd_plan_run(p, st) || k2a_shim_stream_sync(st))
				return fail(KSW2AMD_E_NODEVICE, "plan_fetch: %s", k2a_shim_last_error());
		}
	}
	if (k2a_shim_d2h(p->h_res, p->d_res, sizeof(K2aResult) * (size_t)p->n, p->stream) ||
	    k2a_shim_stream_sync(p->stream))
		return fail(KSW2AMD_E_NODEVICE, "plan_fetch: %s", k2a_shim_last_error());
	return KSW2AMD_OK;
}

int64_t g_reruns;                   /* pairs that fetch ran again (diagnostics: ksw2amd_rerun_count) */
int64_t ksw2amd_rerun_count(void) { return g_reruns; }
int ksw2amd_plan_fetch_raw(ksw2amd_plan_t *p, int32_t *out16)
{
	int i, rc = fetch_results(p);
	if (rc) return rc;
	memset(out16, 0, sizeof(int32_t) * 16 * (size_t)p->n);
	for (i = 0; i < p->n; ++i)
		if (!p->reject_all && needs_rerun(p, i)) {                                     /* see pair_rerun */
			ksw_extz_t z;
			int32_t *o = out16 + 16 * (size_t)i;
			memset(&z, 0, sizeof(z));
			rc = pair_rerun(p, i, 0, &z);
			if (rc) { free(z.cigar); return rc; }
			o[0] = (int32_t)z.max; o[1] = (int32_t)z.zdropped; o[2] = z.max_q; o[3] = z.max_t; o[4] = z.mqe; o[5] = z.mqe_t; o[6] = z.mte; o[7] = z.mte_q;
			o[8] = z.score; o[9] = z.reach_end; o[10] = z.n_cigar; o[11] = p->h_res[i].rows_done; o[12] = o[13] = -1;
			free(z.cigar);
		} else if (p->h_cls[i] >= 0 && !p->reject_all) memcpy(out16 + 16 * (size_t)i, &p->h_res[i], sizeof(K2aResult));
		else {
			int32_t *o = out16 + 16 * (size_t)i;
			o[2] = o[3] = o[5] = o[7] = -1; o[4] = o[6] = o[8] = KSW_NEG_INF; o[12] = o[13] = -1;
		}
	return KSW2AMD_OK;
}

/* M runs -> =/X runs (KSW_EZ_EQX, ksw2_extd2_sse.c:399-406 / ksw2.h:163-182) */
/* `stride` = 1 for plain sequences, 2 for the byte-interleaved copies of a packed task (pointer already at the right half) */
void eqx_rewrite(void *km, const uint8_t *query, const uint8_t *target, int stride, ksw_extz_t *ez)
{
	int n0 = ez->n_cigar, k, i, x = 0, y = 0, n = 0;
	uint32_t *old = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)(n0 + 1));
	memcpy(old, ez->cigar, sizeof(uint32_t) * (size_t)n0);
	ez->n_cigar = 0;
	for (k = 0; k < n0; ++k) {
		uint32_t op = old[k] & 0xf, len = old[k] >> 4;
		if (op == KSW_CIGAR_MATCH) {
			for (i = 0; i < (int)len; ++i) {
				uint32_t o = target[(size_t)(x + i) * stride] == query[(size_t)(y + i) * stride] ? KSW_CIGAR_EQ : KSW_CIGAR_X;
				if (n > 0 && (ez->cigar[n - 1] & 0xf) == o) ez->cigar[n - 1] += 1u << 4;
				else { ez_reserve(km, ez, n + 1); ez->cigar[n++] = 1u << 4 | o; }
			}
			x += (int)len; y += (int)len;
		} else {
			if (n > 0 && (ez->cigar[n - 1] & 0xf) == op) ez->cigar[n - 1] += len << 4;
			else { ez_reserve(km, ez, n + 1); ez->cigar[n++] = len << 4 | op; }
			if (op == KSW_CIGAR_DEL || op == KSW_CIGAR_N_SKIP) x += (int)len;
			else if (op == KSW_CIGAR_INS) y += (int)len;
		}
	}
	ez->n_cigar = n;
	free(old);
}

/* A kalloc pool has no locks (kalloc.c:24-28), and the library's worker threads assemble CIGARs into the caller's pools: every use
 * of a pool is bracketed by ONE OF 64 mutexes chosen by the pool's address -- callers with a pool per thread (the minimap2 pattern)
 * no longer serialise on one process-wide lock while their CIGARs are assembled (round 3: g_km_mu).  km == NULL is libc's realloc,
 * which needs none. */
#define KM_LOCKS 64
static pthread_mutex_t g_km_mu[KM_LOCKS];
static pthread_once_t g_km_once = PTHREAD_ONCE_INIT;
static void km_init(void) { int i; for (i = 0; i < KM_LOCKS; ++i) pthread_mutex_init(&g_km_mu[i], 0); }
static pthread_mutex_t *km_mutex(const void *km)
{
	uint64_t h = (uint64_t)(uintptr_t)km;
	pthread_once(&g_km_once, km_init);
	h ^= h >> 33; h *= 0xff51afd7ed558ccdull; h ^= h >> 29;
	return &g_km_mu[h & (KM_LOCKS - 1)];
}
void km_lock(const void *km) { if (km) pthread_mutex_lock(km_mutex(km)); }
void km_unlock(const void *km) { if (km) pthread_mutex_unlock(km_mutex(km)); }

/* Pairs whose device result cannot be used are run again through the ordinary gather path, one by one:
 *   K2aResult.pad[0] -- flat plans: a packed kernel met a wildcard code (K2aLanePk::seen); the gather path's scan sends the pair to
 *                       the int32 kernels.  Host arenas read the sequences where they lie, device arenas bring them back first;
 *   K2aResult.pad[1] == 1 -- deferred arg-max: a Z-drop could not be ruled out without the arg-max columns, and the second pass (k2a_argmax_kernel)
 *                       found that the reference's test does NOT hold at the row the fill stopped at: the re-run keeps the columns.
 *                       (0 = exact, or settled on the device by the third pass, k2a_zscan_kernel: the record is final.) */
int needs_rerun(const ksw2amd_plan_t *p, int i) { return p->h_cls[i] >= 0 && !p->splice && (((p->flat || p->unscanned) && p->h_res[i].pad[0]) || p->h_res[i].pad[1] == 1); }
/* uniform plans on the 4-bit wire format keep two codes per byte in their staging copy: pair i's query and target, one code per byte,
 * into out[0 .. qlen) and out[qlen .. qlen + tlen) */
void wire4_pair(const ksw2amd_plan_t *p, int i, uint8_t *out)
{
	const K2aPair *d = &p->h_pairs[i];
	int x;
	if (plan_wire4(p) == 2) {                              /* four codes per byte + the pair's escape entries (ksw2_lane.h) */
		const uint32_t stride = p->up_state->wire_stride;
		const uint8_t *slot = p->h_seq + (((size_t)d->qoff + stride) >> 2) - K2A_WIRE2_SLOT;
		int e;
		for (x = 0; x < d->qlen; ++x) out[x] = (p->h_seq[((size_t)d->qoff + (size_t)x) >> 2] >> (2 * (x & 3))) & 3;
		for (x = 0; x < d->tlen_full; ++x) out[d->qlen + x] = (p->h_seq[((size_t)d->toff + (size_t)x) >> 2] >> (2 * (x & 3))) & 3;
		for (e = 0; e < K2A_WIRE2_ESC; ++e) {
			uint32_t ent, off, len, y;
			memcpy(&ent, slot + 4 * e, 4);
			if (!ent) continue;
			off = ent & 0xfffffu; len = (ent >> 20) & 0xffu;
			for (y = 0; y < len; ++y) {                     /* an offset inside the pair's region: the query's bytes, or the target's */
				const uint32_t o = off + y;
				if (o < (uint32_t)d->qlen) out[o] = (uint8_t)(ent >> 28);
				else if (o >= d->toff - d->qoff && o < d->toff - d->qoff + (uint32_t)d->tlen_full) out[d->qlen + (o - (d->toff - d->qoff))] = (uint8_t)(ent >> 28);
			}
		}
		return;
	}
	for (x = 0; x < d->qlen; ++x) { const uint8_t b = p->h_seq[((size_t)d->qoff + (size_t)x) >> 1]; out[x] = (x & 1) ? b >> 4 : b & 15; }
	for (x = 0; x < d->tlen_full; ++x) { const uint8_t b = p->h_seq[((size_t)d->toff + (size_t)x) >> 1]; out[d->qlen + x] = (x & 1) ? b >> 4 : b & 15; }
}
int plan_wire4(const ksw2amd_plan_t *p) { return p->up_state ? p->up_state->wire4 : 0; }      /* 0: none, 1: four bits per code, 2: two bits + escapes */

int pair_rerun(ksw2amd_plan_t *p, int i, void *km, ksw_extz_t *z)
{
	ksw2amd_pair_t a;
	uint8_t *tmp = 0;
	int rc;
	if (p->flat) a = p->src_pairs[i];
	else {                                             /* from the staging copy and the resolved parameters */
		const K2aPair *d = &p->h_pairs[i];
		a.query = p->h_seq + d->qoff; a.target = p->h_seq + d->toff; a.qlen = d->qlen; a.tlen = d->tlen_full;
		a.w = d->w; a.zdrop = d->zdrop; a.end_bonus = p->scalar ? 0 : d->end_bonus; a.flag = p->h_flag[i] & ~F_SCALAR_CONTRACT;
	}
	if (plan_wire4(p)) {
		tmp = (uint8_t*)malloc((size_t)a.qlen + (size_t)a.tlen + 1);
		if (!tmp) return fail(KSW2AMD_E_NOMEM, "plan_fetch: host allocation failed%s", 0);
		wire4_pair(p, i, tmp);
		a.query = tmp; a.target = tmp + a.qlen;
	} else if (p->flat_device) {
		tmp = (uint8_t*)malloc((size_t)a.qlen + (size_t)a.tlen + 1);
		if (!tmp) return fail(KSW2AMD_E_NOMEM, "plan_fetch: host allocation failed%s", 0);
		if (k2a_shim_d2h(tmp, a.query, (size_t)a.qlen, p->stream) || k2a_shim_d2h(tmp + a.qlen, a.target, (size_t)a.tlen, p->stream) ||
		    k2a_shim_stream_sync(p->stream)) { free(tmp); return fail(KSW2AMD_E_NODEVICE, "plan_fetch: %s", k2a_shim_last_error()); }
		a.query = tmp; a.target = tmp + a.qlen;
	}
	++g_no_defer;
	rc = run_serial(p->dual, p->scalar, km, &p->src_sc, 1, &a, z, 1, 0, 0);
	--g_no_defer;
	__sync_fetch_and_add(&g_reruns, 1);
	free(tmp);
	return rc;
}

/* results of pairs [beg, end) into the caller's records: ez[i] with CIGAR memory from `km`, or -- the coalesced single calls -- *ezp[i]
 * with memory from kmp[i] */
void assemble_range(asm_ctx_t *c, int beg, int end)
{
	ksw2amd_plan_t *p = c->p;
	ksw_extz_t *ez = c->ez, **ezp = c->ezp;
	void *km = c->km, **kmp = c->kmp;
	const uint32_t *pool = c->pool;
	const size_t *pos = c->pos;
	int i, nrerun = 0, rc = KSW2AMD_OK;
	for (i = beg; i < end; ++i) {
		ksw_extz_t *z = ezp ? ezp[i] : &ez[i];
		const K2aResult *r = &p->h_res[i];
		if (kmp) km = kmp[i];
		ez_reset(z);
		if (p->splice == 2 && p->h_cls[i] < 0) {          /* an empty sequence: ksw2_extf2_sse.c:33 runs no anti-diagonal, :37 leaves at the first */
			if (imax(p->h_pairs[i].qlen, 0) + imax(p->h_pairs[i].tlen, 0) == 1) z->score = 0;
			else z->zdropped = 1;
			continue;
		}
		if (p->reject_all || p->h_cls[i] < 0) continue;
		if (needs_rerun(p, i)) { ++nrerun; continue; }              /* pair_rerun, below */
		z->max = (uint32_t)r->max; z->zdropped = (uint32_t)r->zdropped;
		z->max_q = r->max_q; z->max_t = r->max_t; z->mqe = r->mqe; z->mqe_t = r->mqe_t;
		z->mte = r->mte; z->mte_q = r->mte_q; z->score = r->score; z->reach_end = r->reach_end;
		if (p->splice != 3 && is_approx(p->h_flag[i])) {      /* (the SSE-compatible kernels produce that mode's fields themselves) */
			z->max = 0; z->max_q = z->max_t = z->mqe_t = z->mte_q = -1; z->mqe = z->mte = KSW_NEG_INF; z->reach_end = 0;
			if (r->zdropped || (p->h_flag[i] & KSW_EZ_EXTZ_ONLY)) continue;    /* no start cell without a maximum */
		}
		if (r->n_cigar > 0) {
			const uint32_t *src = pool + pos[i];
			int nc = r->n_cigar;
			if (kmp) km_lock(km);
			ez_reserve(km, z, nc);
			memcpy(z->cigar, src, sizeof(uint32_t) * (size_t)nc);               /* already in the caller's order (k2a_compact_kernel; ksw2.h:157-159) */
			z->n_cigar = nc;
			if (p->dual && (p->h_flag[i] & KSW_EZ_EQX) && !(p->h_flag[i] & F_SCALAR_CONTRACT)) {
				if (p->flat_device) {                          /* the sequences are in device memory only: bring this pair's back */
					const ksw2amd_pair_t *a = &p->src_pairs[i];
					uint8_t *tmp = (uint8_t*)malloc((size_t)a->qlen + (size_t)a->tlen + 1);
					if (tmp && !k2a_shim_d2h(tmp, a->query, (size_t)a->qlen, p->stream) && !k2a_shim_d2h(tmp + a->qlen, a->target, (size_t)a->tlen, p->stream) &&
					    !k2a_shim_stream_sync(p->stream)) eqx_rewrite(km, tmp, tmp + a->qlen, 1, z);
					else rc = fail(KSW2AMD_E_NODEVICE, "plan_fetch: %s", tmp ? k2a_shim_last_error() : "host allocation failed");
					free(tmp);
				} else eqx_rewrite(km, p->h_seq + p->h_pairs[i].qoff, p->h_seq + p->h_pairs[i].toff, 1, z);
			}
			if (kmp) km_unlock(km);
		}
	}
	if (nrerun) __sync_fetch_and_add(&c->nrerun, nrerun);
	if (rc) c->rc = rc;
}

int plan_fetch_ex(ksw2amd_plan_t *p, void *km, ksw_extz_t *ez, ksw_extz_t **ezp, void **kmp)
{
	int i, rc = fetch_results(p), nrerun = 0;
	uint32_t *pool = 0;
	size_t total = 0, *pos = 0, cap_hpool = 0;
	if (rc) return rc;
	if (!p->reject_all && p->cig_words) {
		/* bring every CIGAR back with one D2H: prefix-sum the counts, compact on the device, download the pool */
		uint32_t *hpos = (uint32_t*)malloc(sizeof(uint32_t) * ((size_t)p->n + 1)), *d_pos = 0, *d_pool = 0;
		size_t cap_pos = 0, cap_pool = 0;
		int bad = 0;
		pos = (size_t*)malloc(sizeof(size_t) * ((size_t)p->n + 1));
		if (!hpos || !pos) { free(hpos); free(pos); return fail(KSW2AMD_E_NOMEM, "plan_fetch: host allocation failed%s", 0); }
		for (i = 0; i < p->n; ++i) {
			if (p->h_cls[i] < 0) p->h_res[i].n_cigar = 0;
			pos[i] = total; hpos[i] = (uint32_t)total;
			total += (size_t)p->h_res[i].n_cigar;
		}
		/* pinned (from the thread's cache): a download into pageable memory is staged by the runtime at a third of the link's
		 * rate, and config 5's CIGARs are 200 MB per batch */
		pool = (uint32_t*)cache_get(BUF_HPOOL, sizeof(uint32_t) * (total + 1), &cap_hpool);
		if (!pool) { free(hpos); free(pos); return fail(KSW2AMD_E_NOMEM, "plan_fetch: host allocation failed%s", 0); }
		if (total > 0) {
			/* device scratch from the thread's buffer cache: an allocation costs milliseconds and synchronises the device */
			d_pos = (uint32_t*)cache_get(BUF_POS, sizeof(uint32_t) * (size_t)p->n, &cap_pos);
			d_pool = (uint32_t*)cache_get(BUF_POOL, sizeof(uint32_t) * total, &cap_pool);
			bad = !d_pos || !d_pool || total > 0xfff00000u ||
			      k2a_shim_h2d(d_pos, hpos, sizeof(uint32_t) * (size_t)p->n, p->stream) ||
			      k2a_shim_launch_compact(p->d_pairs, p->d_res, d_pos, p->n, p->d_cig, d_pool, p->stream) ||
			      k2a_shim_d2h(pool, d_pool, sizeof(uint32_t) * total, p->stream) || k2a_shim_stream_sync(p->stream);
			if (bad) k2a_shim_stream_sync(p->stream);
			cache_put(BUF_POS, d_pos, cap_pos); cache_put(BUF_POOL, d_pool, cap_pool);
		}
		free(hpos);
		if (bad) { free(pos); cache_put(BUF_HPOOL, pool, cap_hpool); return fail(KSW2AMD_E_NODEVICE, "plan_fetch: %s", k2a_shim_last_error()); }
	}
	{
		asm_ctx_t ac;
		ac.p = p; ac.km = km; ac.ez = ez; ac.ezp = ezp; ac.kmp = kmp; ac.pool = pool; ac.pos = pos; ac.nrerun = 0; ac.rc = KSW2AMD_OK;
		/* a big score-only plan (a streamed batch: one plan for everything): its records are assembled by the pool's threads, as the
		 * chunks of the same batch were before -- 65 536 ksw_extz_t of config 2 are 1.3 ms on one thread, next to a 1.4 ms kernel */
		if (!(!p->cig_words && !km && !kmp && p->n >= 16384 && assemble_parallel(&ac))) {
			if (!kmp) km_lock(km);                            /* one pool for the batch: once around the loop; a pool per pair (coalesced calls): pair by pair, inside */
			assemble_range(&ac, 0, p->n);
			if (!kmp) km_unlock(km);
		}
		nrerun = ac.nrerun; if (ac.rc) rc = ac.rc;
	}
	free(pos); cache_put(BUF_HPOOL, pool, cap_hpool);
	if (nrerun > 0 && rc == KSW2AMD_OK) rc = rerun_pairs(p, nrerun, km, ez, ezp, kmp);
	return rc;
}

int ksw2amd_plan_fetch(ksw2amd_plan_t *p, void *km, ksw_extz_t *ez)
{
	return plan_fetch_ex(p, km, ez, 0, 0);
}

/* ---------------------------------------------------------------- batch entry points */

/* Device memory one plan of a worker may take: its fair share of the device (70 % over the `share` workers of this device), but no
 * more than it can get right now -- the free memory plus what its own buffer cache hands back.  (Free memory alone is the wrong
 * measure: after the first batch the workers' caches hold most of it, and a chunk that fitted before would be cut into slivers.) */
size_t thread_cached_device_bytes(void)
{
	size_t b = 0;
	int k;
	for (k = 0; k < BUF_KINDS; ++k) { int d; for (d = 0; d < CACHE_DEPTH; ++d) if (!BUF_IS_HOST(k) && g_cache[k][d].p) b += g_cache[k][d].cap; }
	return b;
}
size_t device_budget(size_t free_b, size_t total_b, int share)
{
	const size_t fair = total_b / 10 * 7 / (size_t)(share > 0 ? share : 1), have = (free_b + thread_cached_device_bytes()) / 10 * 9;
	return fair < have ? fair : have;
}

size_t pair_device_bytes(int dual, const ksw2amd_pair_t *a)
{
	/* upper bound of what plan_create allocates for this pair */
	size_t b = (size_t)imax(a->qlen, 0) + (size_t)imax(a->tlen, 0) + 96 + sizeof(K2aPair) + sizeof(K2aResult) + 4;
	if (a->qlen > 0 && a->tlen > 0 && !(a->flag & KSW_EZ_SCORE_ONLY)) {
		int mx = imax(a->qlen, a->tlen), w = (a->w < 0 || a->w > mx) ? mx : a->w;
		size_t steps = (size_t)a->qlen + (size_t)a->tlen / 8 + 2 + K2A_TB_PAD;      /* lane runs are padded */
		size_t lanes = (size_t)imin(64, (2 * w + 16) / 9 + 2);
		(void)lanes;
		if (w <= 1040 || a->tlen <= 2048)
			b += steps * 64 * (dual ? 32 : 16) / (w <= 68 ? 8 : w <= 284 ? 4 : w <= 536 ? 2 : 1) + 256;
		else   /* generation-serial: one (qlen + 63)-step sweep per 1024 rows, 64 lanes x 16 rows per step */
			/* int32 and dual-gap packed classes: 16 bytes per lane-step and pair; single-gap packed class (4-bit codes): 8.  A
			 * single-gap pair that ends up in the int32 class (wildcards, generic matrix, too few tasks) needs twice this: the
			 * callers halve a plan whose allocation fails */
			b += (((size_t)a->tlen / 1024 + 1) * ((size_t)a->qlen + 72) + K2A_TB_PAD) * 64 * (dual ? 16 : 8) + 24 * (size_t)a->qlen + 320 + 131072;
		b += ((size_t)a->qlen + a->tlen + 2) * 4;
	} else if (!dual && a->qlen > 0 && a->tlen > 0 && !g_no_defer && !(ENV(DEFER) && atoi(ENV(DEFER)) == 0)) {
		/* score only, single gap: the deferred arg-max kernels' checkpoint stream of the one-alignment-per-wavefront geometries
		 * (plan_create_ex): 512 bytes per step of a wavefront that two pairs share */
		const int mx = imax(a->qlen, a->tlen), w = (a->w < 0 || a->w > mx) ? mx : a->w;
		if (w > 68 && w <= 536 && !(a->flag & KSW_EZ_APPROX_MAX)) b += ((size_t)a->qlen + (size_t)a->tlen / 8 + 64) * 256 + (size_t)a->tlen;
	}
	return b;
}

/* One slice of a batch on the calling thread: plan(s) sized to `1 / share` of the device's free memory (share = threads that
 * work on this device at the same time), each created, run on the thread's own stream, fetched and destroyed. */
int64_t g_phase_us[4];                   /* ksw2amd_host_phase_us */
void phase_add(double create_ms, double launch_ms, double fetch_ms)
{
	__sync_fetch_and_add(&g_phase_us[0], (int64_t)(create_ms * 1000.0)); __sync_fetch_and_add(&g_phase_us[1], (int64_t)(launch_ms * 1000.0));
	__sync_fetch_and_add(&g_phase_us[2], (int64_t)(fetch_ms * 1000.0)); __sync_fetch_and_add(&g_phase_us[3], 1);
}
void ksw2amd_host_phase_us(int64_t out[4])
{
	int i;
	for (i = 0; i < 4; ++i) out[i] = g_phase_us[i];
}

int run_serial(int dual, int scalar, void *km, const ksw2amd_scoring_t *sc, int n, const ksw2amd_pair_t *pairs, ksw_extz_t *ez, int share, const flat_src_t *flat, int want_stream)
{
	size_t budget, free_b = 0, total_b = 0, acc;
	const char *env = ENV(MAX_BYTES);
	int beg = 0, end, unit;
	double t0;
	if (n <= 0) return KSW2AMD_OK;
	if (share < 1) share = 1;
	unit = unit_pairs(&pairs[0]);
	if (unit <= 0) unit = k2a_shim_simd_count();          /* generation-serial classes: two tasks of two pairs per CU */
	if (env && atoll(env) > 0) budget = (size_t)atoll(env);
	else {
		/* small batches (the single-pair entry points above all) skip the free-memory query: it costs ~0.1 ms */
		for (end = 0, acc = 0; end < n && acc <= ((size_t)256 << 20); ++end) acc += pair_device_bytes(dual, &pairs[end]);
		if (acc <= ((size_t)256 << 20)) budget = (size_t)1 << 30;
		else {
			if (k2a_shim_mem_info(&free_b, &total_b)) return fail(KSW2AMD_E_NODEVICE, "mem_info: %s", k2a_shim_last_error());
			budget = device_budget(free_b, total_b, share);
		}
	}
	while (beg < n) {
		ksw2amd_plan_t *p;
		int rc, limit = n - beg;
		size_t seq = 0, cig = 0;
		/* a plan addresses its sequence arena and CIGAR scratch with 32-bit offsets: stay below 3 G bytes / words each */
		for (end = beg, acc = 0; end < n; ++end) {
			const size_t b = pair_device_bytes(dual, &pairs[end]);
			const size_t sq = (size_t)imax(pairs[end].qlen, 0) + (size_t)imax(pairs[end].tlen, 0) + 96;
			const size_t cg = (pairs[end].flag & KSW_EZ_SCORE_ONLY) ? 0 : sq;
			if (end > beg && (acc + b > budget || seq + sq > 3000000000u || cig + cg > 3000000000u || end - beg >= (1 << 22))) break;
			acc += b; seq += sq; cig += cg;
		}
		/* a batch that is split anyway: whole device fills per plan (see uniform_chunks; config 4 through 249-pair plans fell
		 * back to the int32 class, one wavefront on a quarter of the SIMDs: 254 GCUPS end to end) */
		if (end < n && unit > 0 && end - beg > unit) end = beg + (end - beg) / unit * unit;
		/* the footprint estimate is an upper bound in practice; should the device still run out, retry with half the pairs */
		t0 = now_ms();
		for (p = 0; p == 0; ) {
			if (end - beg > limit) end = beg + limit;
			p = plan_create_ex(dual, scalar, sc, end - beg, pairs + beg, flat, want_stream && beg == 0 && end == n);
			if (p) break;
			if (!strstr(g_err, "alloc") || end - beg <= 1) return strstr(g_err, "alloc") ? KSW2AMD_E_NOMEM : g_err[0] && strstr(g_err, "device") ? KSW2AMD_E_NODEVICE : KSW2AMD_E_PARAM;
			release_thread_cache();
			limit = (end - beg) / 2;
		}
		{
			const double t1 = now_ms();
			double t2, t3;
			rc = ksw2amd_plan_run(p, g_plan_stream ? g_plan_stream : thread_stream());
			t2 = now_ms();
			if (rc == KSW2AMD_OK) rc = ksw2amd_plan_fetch(p, km, ez + beg);
			t3 = now_ms();
			phase_add(t1 - t0, t2 - t1, t3 - t2);
			if (!trace_on()) { ksw2amd_plan_destroy(p); p = 0; }
			if (trace_on()) {
				float dev_ms = -1.0f;
				if (rc == KSW2AMD_OK && p->ran && !p->reject_all && p->ntasks > 0) dev_ms = k2a_shim_event_ms(p->ev[0], p->ev[2]);
				ksw2amd_plan_destroy(p); p = 0;
				fprintf(stderr, "[ksw2_amd] serial plan @%d n=%d: create %.2f ms, launch %.2f ms, wait+fetch %.2f ms (device: %.2f ms from the first launch to the last kernel's end), destroy %.2f ms, budget %zu\n", beg, end - beg, t1 - t0, t2 - t1, t3 - t2, dev_ms, now_ms() - t3, budget);
			}
		}
		if (rc) return rc;
		beg = end;
	}
	return KSW2AMD_OK;
}

