/*
 * ksw2_host.c -- C host side of libksw2_amd.so (the drop-in boundary, include/ksw2_amd.h).
 *
 * What lives here: argument checks and early rejects of the "...2_sse" signatures
 * (ksw2_extz2_sse.c:56-82, ksw2_extd2_sse.c:75-100), the implicit match/mismatch/wildcard scoring
 * (ksw2_extz2_sse.c:66-69,125-140), packing of a batch into device arenas, the choice of kernel
 * geometry per pair, and the assembly of ksw_extz_t results including CIGAR buffer growth with the
 * reference's doubling rule (ksw2.h:113-123) through libc or the caller's kalloc (ksw2.h:103-111).
 * All DP work happens in the kernels behind ksw2_shim.h; there is no CPU alignment code in this file.
 */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <execinfo.h>
#include <signal.h>
#include <unistd.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#include <pthread.h>
#include <string.h>
#include <time.h>
#include "../../include/ksw2_amd.h"
#include "ksw2_shim.h"

#define F_SCALAR_CONTRACT 0x40000000   /* internal: call came through ksw_extz / ksw_extd / ksw_gg* */
#define NCLS_MAX (K2A_NCFG * 3 * 2)
#define NPASS (2 + 4 * K2A_NPKCFG)     /* per class: one alignment per lane group | packed class pc (x re-based) (x no maximum tracking) | solo */
#define PASS_SOLO (NPASS - 1)           /* both halves of the packed registers for one alignment (ksw2_lane_solo.h) */
#define NCLS_ENTRIES (NCLS_MAX * NPASS)

static __thread char g_err[512];
static __thread int g_no_defer;
static size_t thread_cached_device_bytes(void);            /* set around the re-run of a pair the deferred arg-max kernels handed back as inexact */

const char *ksw2amd_last_error(void) { return g_err; }
const char *ksw2amd_backend(void) { return k2a_shim_backend(); }
int ksw2amd_device_count(void) { return k2a_shim_device_count(); }

static int fail(int code, const char *fmt, const char *detail)
{
	snprintf(g_err, sizeof(g_err), fmt, detail ? detail : "");
	return code;
}

void ksw2amd_release_cache(void);
static void release_thread_cache(void);

/* ---------------------------------------------------------------- environment switches
 * Every KSW2AMD_* switch (A/B runs, tests, tuning) is read ONCE per process into a table -- the call paths, the coalesced
 * single-pair calls above all, never touch getenv().  ksw2amd_reload_env() reads them again (tests that flip a switch inside
 * one process; the Python binding calls it before every plan / batch). */
#define K2A_ENV_LIST \
	X(ABORT_ON_ERROR) \
	X(BACKTRACE) \
	X(APPROX_DROP_EXACT) \
	X(CHUNK_GCELLS) \
	X(CHUNK_MB) \
	X(COALESCE_SLOTS) \
	X(COALESCE_WINDOW_US) \
	X(COALESCE_PLAIN_STREAMS) \
	X(SHARED_UP_MIN_MB) \
	X(DEFER) \
	X(EXTF_HBM) \
	X(EXTF_LANE) \
	X(EXTF_RING) \
	X(GROW) \
	X(EXTF_LDS) \
	X(EXTF_WIN) \
	X(EXTS_BIG) \
	X(EXTS_REG) \
	X(LDSCODES) \
	X(LDSROWS) \
	X(LONG_MS) \
	X(MAX_BYTES) \
	X(NO_PARCOPY) \
	X(NO_PK) \
	X(NO_PKMP) \
	X(NO_RB) \
	X(NO_SHARED_UP) \
	X(PK_FIRST) \
	X(POOL_MIN) \
	X(SERIAL) \
	X(SIMDS) \
	X(SMALL_CELLS) \
	X(SOLO) \
	X(SSEC_HBM) \
	X(SSE_COMPAT) \
	X(STREAM) \
	X(STREAM_FAULT) \
	X(STREAM_MIN_CELLS) \
	X(STREAM_PIECE_KB) \
	X(STREAM_SLEEP_US) \
	X(STREAM_TIMEOUT_MS) \
	X(THREADS) \
	X(TRACE)
enum {
#define X(n) ENV_##n,
	K2A_ENV_LIST
#undef X
	ENV_COUNT
};
static const char *const g_env_name[ENV_COUNT] = {
#define X(n) "KSW2AMD_" #n,
	K2A_ENV_LIST
#undef X
};
static __thread void *g_plan_stream;                 /* ... and this is the stream it will run on: its uploads go there too (in order: no event, no second queue) */
static __thread int g_latency_plan;                  /* this thread is creating the plan of a single-pair call (or of a coalesced batch of them) */
static const char *g_env[ENV_COUNT];
static volatile int g_env_ready;
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
static void env_load(void)
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
static int env_switch(const char *v) { return v && *v ? (atoi(v) != 0) : -1; }      /* -1 = automatic */
#define ENV(n) (g_env_ready ? g_env[ENV_##n] : (env_load(), g_env[ENV_##n]))
static int env_flag(const char *e, int dflt) { return e && *e ? atoi(e) != 0 : dflt; }
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
enum { BUF_HSEQ, BUF_SEQ, BUF_PAIRS, BUF_RES, BUF_ORDER, BUF_TB, BUF_CIG, BUF_BND, BUF_POS, BUF_POOL, BUF_HPOOL, BUF_HRES, BUF_WM, BUF_HMETA, BUF_KINDS };
#define BUF_IS_HOST(k) ((k) == BUF_HSEQ || (k) == BUF_HPOOL || (k) == BUF_HRES || (k) == BUF_HMETA)      /* pinned host staging; everything else is device memory */
#define CACHE_DEPTH 2                  /* a worker that queues its next chunk before it fetches the current one holds two plans */
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
static void *thread_stream(void) { if (!g_stream) { thread_owns_cache(); g_stream = k2a_shim_stream_create(); } return g_stream; }
static void *thread_upload_stream(void) { if (!g_up_stream) { thread_owns_cache(); g_up_stream = k2a_shim_stream_create(); } return g_up_stream; }
/* Flat plans upload on ONE stream per device, shared by all host threads: their arena spans go up at link rate one after the
 * other, in the order the plans were created, so the first chunk of a pooled batch is on the device after 1 / nchunks of the
 * batch's upload time and its kernels run under the remaining uploads.  (Six workers uploading on six streams share the link:
 * every chunk arrives at the END of the total upload time -- config 2: 0.9 ms for each 8 MB chunk, then the kernels.) */
#define SHARED_UP_MAXDEV 16
static void *g_shared_up[SHARED_UP_MAXDEV];
static pthread_mutex_t g_shared_up_mu = PTHREAD_MUTEX_INITIALIZER;
static pthread_mutex_t g_shared_issue_mu = PTHREAD_MUTEX_INITIALIZER;      /* held while ONE plan's copies go into the shared stream */
static void *shared_upload_stream(void)
{
	const int dev = k2a_shim_get_device();
	void *s;
	if (dev < 0 || dev >= SHARED_UP_MAXDEV) return 0;
	pthread_mutex_lock(&g_shared_up_mu);
	if (!g_shared_up[dev]) g_shared_up[dev] = k2a_shim_stream_create();
	s = g_shared_up[dev];
	pthread_mutex_unlock(&g_shared_up_mu);
	return s;
}

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

static void *cache_get(int kind, size_t bytes, size_t *cap)
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

static void cache_put(int kind, void *p, size_t cap)
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

static void release_thread_cache(void)
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
#define K2A_MAXPIECES 48
#define K2A_STREAM_MARGIN 256          /* bytes past a sequence's end that the kernels may touch (dword query loads, one strip of target codes) */
static int stream_env(void) { return env_switch(ENV(STREAM)); }
static int64_t stream_min_cells(void) { const char *e = ENV(STREAM_MIN_CELLS); return e && atoll(e) >= 0 ? atoll(e) : 1000000; }      /* cells per pair from which one-shape batches are streamed by default */
static int64_t g_stream_stat[2];           /* streamed plans run, runs that were aborted and repeated unstreamed */
void ksw2amd_stream_stats(int64_t out[2]) { out[0] = g_stream_stat[0]; out[1] = g_stream_stat[1]; }
/* the watermark source: page-locked, block k filled with k + 1, one per device for the life of the process */
static uint32_t *g_wm_src[SHARED_UP_MAXDEV];
static pthread_mutex_t g_wm_mu = PTHREAD_MUTEX_INITIALIZER;
static const uint32_t *wm_source(void)
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
/* the upload side of a streamed plan: pieces [pb[k], pb[k + 1]) of the arena, issued in order by whoever finishes the gap */
typedef struct {
	int np, next, fault, sleep_us;         /* next: first piece not yet issued; fault / sleep_us: test hooks (KSW2AMD_STREAM_FAULT / _SLEEP_US) */
	double issue_ms, t0, t_first, t_last;  /* KSW2AMD_TRACE: host time spent in the upload calls; creation, first and last piece issued (now_ms) */
	int hold, all_ready;                   /* hold: pieces that may go up for now (the plan's small arrays must not queue behind the whole arena:
	                                        * two pieces, the arrays, then the rest); all_ready: a flat arena -- nothing to wait for */
	size_t pb[K2A_MAXPIECES + 1];
	int pfirst[K2A_MAXPIECES + 1];         /* gather plans: first pair of each piece (the copy's work units) */
	uint8_t done[K2A_MAXPIECES];
	int left[K2A_MAXPIECES];               /* gather plans: copy chunks of the piece still outstanding (the copy's work units are finer than the pieces) */
	const uint8_t *src; size_t src_bytes;  /* host (or device, flat device arenas) bytes of [0, src_bytes); the rest of the last piece comes from `tail` */
	const uint8_t *tail;
	int src_on_device;
	uint8_t *d_seq, *d_wm;
	const uint32_t *wm_src;
	void *up;
	int rc;
	pthread_mutex_t mu;
} stream_up_t;
/* issue every piece that is ready and allowed, in order; called with piece `k` just completed (k < 0: only look again) */
static double now_ms(void);
static void stream_issue(stream_up_t *u, int k)
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
		if (mid > lo && (u->src_on_device ? k2a_shim_d2d(u->d_seq + lo, u->src + lo, mid - lo, u->up) : k2a_shim_h2d(u->d_seq + lo, u->src + lo, mid - lo, u->up))) u->rc = -1;
		if (hi > mid && u->tail && k2a_shim_h2d(u->d_seq + mid, u->tail + (mid - u->src_bytes), hi - mid, u->up)) u->rc = -1;
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

static void ez_reset(ksw_extz_t *ez)           /* ksw2.h:184-189; cigar and m_cigar survive */
{
	ez->max_q = ez->max_t = ez->mqe_t = ez->mte_q = -1;
	ez->max = 0; ez->score = ez->mqe = ez->mte = KSW_NEG_INF;
	ez->n_cigar = 0; ez->zdropped = 0; ez->reach_end = 0;
}

static void ez_reserve(void *km, ksw_extz_t *ez, int n)   /* capacity sequence of ksw_push_cigar, ksw2.h:116-119 */
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
static int is_approx(int flag)
{
	return !(flag & F_SCALAR_CONTRACT) && (flag & KSW_EZ_APPROX_MAX) && !(flag & KSW_EZ_APPROX_DROP);
}

/* ---------------------------------------------------------------- plan */

typedef struct {
	int qd;                                /* >= 0: index of the class's K2aQueueDesc (streamed plans) */
	int cfg, mode, generic, pk, rb, nomax, solo, defer, first, count;   /* defer: arg-max columns by a second pass (K2aLanePk, DEFER); pk: packed-int16 tasks, two h_order entries per task; rb: per-strip bases;
	                                                        * nomax: KSW_EZ_APPROX_MAX launches without row maxima */
	K2aScoring sc;
} cls_t;

struct ksw2amd_plan_s {
	int dual, n, reject_all, ran, ncls;
	int m;
	K2aPair *h_pairs;
	int8_t *h_cls;                 /* class index per pair, -1 = rejected before the device */
	uint8_t *h_half;               /* streamed plans: upload pieces that must have landed before the pair's sequences are complete on the device */
	int32_t *h_flag;               /* caller's flag per pair */
	uint32_t *h_order;
	int ntasks;
	cls_t cls[NCLS_ENTRIES];
	int norder;
	uint8_t *h_seq;
	size_t seq_bytes, tb_bytes, cig_words, bnd_words;
	size_t cap[BUF_KINDS];         /* capacities of the (possibly recycled) buffers */
	uint8_t *d_seq, *d_tb;
	int32_t *d_bnd;
	K2aPair *d_pairs;
	K2aResult *d_res, *h_res;
	uint32_t *d_order, *d_cig;
	void *ev[3];
	void *stream;
	int stream_used;
	int64_t cells;
	/* splice-aware plans (ksw2amd_exts_plan_create): tasks of h_order grouped by kernel mode x matrix variant */
	int splice, s_first[3][2][3], s_count[3][2][3];   /* [mode][matrix variant][window class: 8 slots, 16 slots, state in HBM] */
	K2aSplice s_par[2];
	/* gap-linear X-drop plans (ksw2amd_extf_plan_create, splice == 2): tasks grouped by where the state arrays live */
	int f_first[7], f_count[7];          /* [6]: one extension per lane, groups of 64 with interleaved sequences (ksw2_lane_extf.h) */
	size_t f_state_bytes;                /* that class: zeroed state rows at the start of d_tb, re-zeroed by every run */
	K2aExtf f_par;
	/* SSE-compatible plans (ksw2amd_sse_plan_create, splice == 3): tasks grouped by kernel mode in s_first / s_count[mode][0][0] */
	K2aSsec c_par;
	size_t c_lds[3];               /* SSE-compatible plans: per-wavefront LDS bytes of the tasks whose state fits LDS, per mode (0 = none) */
	/* flat plans (ksw2amd_plan_create_flat): the sequences went up as they lie in the caller's arena -- no staging copy, no host
	 * scan for wildcard codes; the packed kernels report such codes and fetch re-runs those pairs (pair_rerun) */
	int flat, flat_device, scalar;         /* flat: h_seq (host arenas) is the caller's memory, not a staging buffer */
	void *up_ev;                           /* flat plans from host arenas do not wait for their upload: the run's stream waits for this event */
	uint8_t *flat_tail;                    /* ... and the staging block of the arena's padding lives as long as the plan */
	ksw2amd_pair_t *src_pairs;             /* the caller's pairs (pointers into the arena), kept for the re-runs */
	ksw2amd_scoring_t src_sc; int8_t *src_mat;
	/* streamed plans (section "streamed plans" below): the sequence arena goes up in pieces, one launch per score-only packed class
	 * starts under the upload, every wavefront waiting for its own task's pieces */
	int streamed;                          /* classes with cls_t.qd >= 0 exist and the next run launches them as queues */
	int meta_folded;                       /* 1: d_order lies inside d_pairs' buffer; 2: both lie inside d_seq's (one upload per plan) */
	int unscanned;                         /* a streamed plan's gathered arena is copied, not scanned: wildcard pairs are reported by the kernels like in flat plans */
	stream_up_t *up_state;                 /* the piece-wise upload (lives as long as the plan: the gather's workers issue pieces) */
	struct gather_s *gather;               /* the gather of a streamed plan, running on the pool's threads until gather_wait() */
	int npieces;
	void *wm_ev;                           /* behind the copy that zeroes the watermark block (upload stream) */
	uint8_t *h_meta;                       /* page-locked staging of the small arrays (streamed plans) */
	void *meta_ev;                         /* behind the plan's small arrays (upload stream): what a streamed run waits for before its first launch */
	uint8_t *d_wm;                         /* watermark block (K2A_WM_BYTES) followed by the K2aQueueDesc array of the streamed classes */
	K2aQueueDesc *h_qd; int nqd;
	size_t need_words;                     /* per-wavefront-task piece counts, behind the task lists in d_order */
};

static size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }
static int imin(int a, int b) { return a < b ? a : b; }
static int imax(int a, int b) { return a > b ? a : b; }

/* in-band cells of the exact band |i-j| <= w: sum over target rows i of min(qlen-1, i+w) - max(0, i-w) + 1, closed form */
static int64_t band_cells(int qlen, int tlen, int w)
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
static int geom_fits(int G, int C, int tlen_eff, int w)
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
static void build_eff(int dual, int m, const int8_t *mat, int e, int e2, int generic, int8_t *eff)
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

/* Packed-int16 class (ksw2_lane_pk.h): the scoring must be match / mismatch / wildcard on a 5-letter alphabet ... */
typedef struct { int ok, a, b, n, smax, smin, qemax, qemin, q, e; } pkinfo_t;

static void pk_scoring(int dual, int m, const int8_t *mat, int q, int e, int q2, int e2, int generic, pkinfo_t *o)
{
	K2aScoring t;
	int x, y, ok = (m == 5);
	memset(o, 0, sizeof(*o));
	if (ok) {
		int8_t eff[25];
		build_scoring(dual, m, mat, q, e, q2, e2, generic, &t);
		for (x = 0; x < 5; ++x)
			for (y = 0; y < 5; ++y)
				eff[x * 5 + y] = y < 4 ? (int8_t)(t.prof[x] >> (8 * y)) : (int8_t)t.colw[x];
		o->a = eff[0]; o->b = eff[1]; o->n = eff[24];
		o->smax = o->smin = eff[0];
		for (x = 0; x < 5; ++x)
			for (y = 0; y < 5; ++y) {
				const int want = (x == 4 || y == 4) ? o->n : x == y ? o->a : o->b;
				if (eff[x * 5 + y] != want) ok = 0;
				o->smax = imax(o->smax, eff[x * 5 + y]); o->smin = imin(o->smin, eff[x * 5 + y]);
			}
	}
	o->q = q; o->e = e;
	o->qemax = dual ? imax(q + e, q2 + e2) : q + e;
	o->qemin = dual ? imin(q + e, q2 + e2) : q + e;
	/* the fill loop adds these as unsigned 32-bit constants to both halves at once (ksw2_lane_pk.h, offset form) */
	if (q < 0 || e < 0 || (dual && (q2 < 0 || e2 < 0)) || o->a + e < 0 || o->a - o->b < 0) ok = 0;
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
	if ((int64_t)4 * C * D + 2 * k->qemax + imax(k->smax, 0) + 64 > K2A_PK_SLACK) return 0;
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

/* copy a sequence into the staging arena and report whether it holds a residue code >= 4 (the wildcard of a 5-letter
 * alphabet): one pass over the bytes instead of a scan plus a memcpy */
static int copy_scan(uint8_t *dst, const uint8_t *src, int n)
{
	int i = 0;
	uint64_t acc = 0, v0, v1, v2, v3;
	for (; i + 32 <= n; i += 32) {
		memcpy(&v0, src + i, 8); memcpy(&v1, src + i + 8, 8); memcpy(&v2, src + i + 16, 8); memcpy(&v3, src + i + 24, 8);
		memcpy(dst + i, &v0, 8); memcpy(dst + i + 8, &v1, 8); memcpy(dst + i + 16, &v2, 8); memcpy(dst + i + 24, &v3, 8);
		acc |= (v0 | v1) | (v2 | v3);
	}
	for (; i < n; ++i) { dst[i] = src[i]; acc |= src[i]; }
	return (acc & 0xfcfcfcfcfcfcfcfcull) != 0;
}

static int gather_wait(ksw2amd_plan_t *p);
void ksw2amd_plan_destroy(ksw2amd_plan_t *p)
{
	int i;
	if (!p) return;
	if (p->gather) gather_wait(p);
	if (p->up_ev) { k2a_shim_event_sync(p->up_ev); k2a_shim_event_destroy(p->up_ev); p->up_ev = 0; }      /* the upload reads host blocks freed below */
	if (p->up_state) { pthread_mutex_destroy(&p->up_state->mu); free(p->up_state); p->up_state = 0; }
	if (p->wm_ev) { k2a_shim_event_destroy(p->wm_ev); p->wm_ev = 0; }
	if (p->meta_ev) { k2a_shim_event_sync(p->meta_ev); k2a_shim_event_destroy(p->meta_ev); p->meta_ev = 0; }      /* (its copies read the page-locked staging recycled below) */
	if (p->stream_used) k2a_shim_stream_sync(p->stream);     /* nothing may still be running on buffers that get recycled */
	cache_put(BUF_SEQ, p->d_seq, p->cap[BUF_SEQ]); cache_put(BUF_TB, p->d_tb, p->cap[BUF_TB]);
	if (p->meta_folded != 2) cache_put(BUF_PAIRS, p->d_pairs, p->cap[BUF_PAIRS]);
	cache_put(BUF_RES, p->d_res, p->cap[BUF_RES]);
	if (!p->meta_folded) cache_put(BUF_ORDER, p->d_order, p->cap[BUF_ORDER]);
	cache_put(BUF_CIG, p->d_cig, p->cap[BUF_CIG]);
	cache_put(BUF_BND, p->d_bnd, p->cap[BUF_BND]); cache_put(BUF_WM, p->d_wm, p->cap[BUF_WM]); cache_put(BUF_HMETA, p->h_meta, p->cap[BUF_HMETA]);
	free(p->h_qd);
	for (i = 0; i < 3; ++i) if (p->ev[i]) { if (!g_ev_cache[i]) g_ev_cache[i] = p->ev[i]; else k2a_shim_event_destroy(p->ev[i]); }
	free(p->h_pairs); free(p->h_cls); free(p->h_half); free(p->h_flag); free(p->h_order);
	cache_put(BUF_HRES, p->h_res, p->cap[BUF_HRES]);          /* pinned: the results come back with one asynchronous copy */
	if (!p->flat) cache_put(BUF_HSEQ, p->h_seq, p->cap[BUF_HSEQ]);      /* (a flat plan's h_seq is the caller's arena) */
	free(p->src_pairs); free(p->src_mat); free(p->flat_tail);
	free(p);
}

typedef struct { int64_t cost; uint32_t idx, tf; } sort_t;      /* tf = true target length: part of a packed pair's shape */
static int cmp_cost_desc(const void *a, const void *b)
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
typedef struct { uint8_t *h_seq; const K2aPair *hp; const ksw2amd_pair_t *pairs; uint8_t *wild; stream_up_t *su; } copy_ctx_t;   /* su: streamed plans -- chunk k of the copy is piece k of the upload */
static void copy_range(const copy_ctx_t *c, int beg, int end);
static int parallel_copy(copy_ctx_t *c, int n, size_t bytes);

static int gather_start(ksw2amd_plan_t *p, stream_up_t *su, const ksw2amd_pair_t *pairs, int n);
static double now_ms(void);
static int trace_level(void) { const char *e = ENV(TRACE); return e ? atoi(e) : 0; }

/* What every plan creator (extz / extd, splice-aware, X-drop, SSE-compatible) starts with: the plan record and its per-pair host
 * arrays.  `with_order`: the task list is as long as the batch (one entry per pair) and allocated here. */
static ksw2amd_plan_t *plan_new(const char *who, int n, int with_order)      /* with_order < 0: extz / extd plans -- they initialise every record they use themselves */
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
static void plan_ready(ksw2amd_plan_t *p)
{
	int i;
	for (i = 0; i < 3; ++i) { p->ev[i] = g_ev_cache[i] ? g_ev_cache[i] : k2a_shim_event_create(); g_ev_cache[i] = 0; }
	p->stream = 0; p->stream_used = 0;
}

/* `flat`: the pairs' query / target pointers all lie in ONE arena, in host memory or (flat->on_device) in device memory.  The plan
 * then uploads (or copies on the device) the arena's span as it is and addresses the sequences where they lie: no per-pair gather,
 * no staging copy, no host pass over the bytes.  What the gather pass also did was to look for wildcard codes (the packed kernels
 * cannot score them): flat plans leave that to the packed kernels themselves (K2aLanePk::seen) and re-run what they report. */
typedef struct { int on_device; } flat_src_t;
/* room for a plan's small arrays behind its sequences: K2aPair per pair, the task lists (two entries per packed task at most), slack */
#define META_ROOM(n) (align_up(sizeof(K2aPair) * ((size_t)(n) + 1), 256) + sizeof(uint32_t) * (3 * (size_t)(n) + 8) + 512)
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
static int pair_has_wild(const ksw2amd_pair_t *a)
{
	return ((or_bytes(a->query, a->qlen) | or_bytes(a->target, a->tlen)) & 0xfcfcfcfcfcfcfcfcull) != 0;
}

static ksw2amd_plan_t *plan_create_ex(int dual, int scalar, const ksw2amd_scoring_t *sc, int n, const ksw2amd_pair_t *pairs, const flat_src_t *flat, int want_stream)
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
	 * themselves: 8 or 16 lanes x 18 / 8 rows per alignment is the geometry that fills a device, but a lane then walks 4 strips of
	 * 18 rows one after the other -- one 512 x 512, w = 64 pair takes 0.455 ms that way and 0.33 ms with 64 lanes x 8 rows (or the
	 * solo kernel), and a caller that waits for ONE pair, or 64 threads that wait for their 64, wait for exactly that. */
	const int pk_first = ENV(PK_FIRST) ? atoi(ENV(PK_FIRST)) : (g_latency_plan && n <= 256) ? 2 : 0;

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
	else {
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
			copy_ctx_t cc;
			cc.h_seq = p->h_seq; cc.hp = p->h_pairs; cc.pairs = pairs; cc.wild = 0; cc.su = su;
			su->hold = su->np;
			for (k = 0; k < su->np; ++k) { copy_range(&cc, su->pfirst[k], su->pfirst[k + 1]); stream_issue(su, k); }      /* (cc.su is not consulted by copy_range itself) */
		}
		if (su->rc) { fail(KSW2AMD_E_NODEVICE, "upload failed: %s", k2a_shim_last_error()); goto err; }
	} else {
		copy_ctx_t cc;
		su = 0;
		cc.h_seq = p->h_seq; cc.hp = p->h_pairs; cc.pairs = pairs; cc.wild = solo_ok;              /* (solo_ok doubles as the wildcard flags until the loop below sets it) */
		cc.su = 0;
		if (!flat && !parallel_copy(&cc, n, p->seq_bytes)) copy_range(&cc, 0, n);                   /* flat: nothing is copied, nothing scanned (wild = 0) */
	}
	for (i = 0; i < n; ++i) {
		const ksw2amd_pair_t *a = &pairs[i];
		K2aPair *d = &p->h_pairs[i];
		const int fl = p->h_flag[i];
		int w = a->w, cfg, mode, generic, mx, wild;
		if (a->qlen <= 0 || a->tlen <= 0) { ++ninvalid; continue; }
		wild = solo_ok[i]; solo_ok[i] = 0;
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
			if (pc == K2A_PKCFG_MP && (flat || p->unscanned) && ((flat && flat->on_device) || pair_has_wild(a))) pc = K2A_NPKCFG;
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
			if (uni) { if (p->h_cls[0] >= 0 && pk_ok[0] && pk_ok[0] != PASS_SOLO) cnt[p->h_cls[0] * NPASS + pk_ok[0]] = n; }
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
	if (!su) p->d_seq = (uint8_t*)cache_get(BUF_SEQ, p->seq_bytes + (flat ? 0 : META_ROOM(n)), &p->cap[BUF_SEQ]);
	if (!p->d_seq) { fail(KSW2AMD_E_NOMEM, "device allocation failed: %s", k2a_shim_last_error()); goto err; }
	/* Big uploads go through ONE stream per device, whoever issues them: the chunks of a big batch are packed by several worker
	 * threads at once, and six 80 MB copies on six streams share the link -- all of them arrive after 9-13 ms and the device idles
	 * until then (KSW2AMD_TRACE=2 timeline of the 10 k headline); in one queue the first chunk's bytes are there 1.6 ms after its
	 * packing ends and its kernel starts while the others still travel (pointer entry, MI355X: headline 4 024 -> 4 336 GCUPS end to
	 * end, config 2 968 -> 1 165, 10 k with CIGAR 1 290 -> 1 321; config 3's 8 MB chunks and config 5 unchanged within noise).  Plans
	 * under 16 MB (single calls, coalesced batches, small chunks) keep the calling thread's own stream and wait for it: an event per
	 * call would only add latency there.  KSW2AMD_NO_SHARED_UP=1: the old behaviour, for A/B runs. */
	up = su ? su->up : (flat && !flat->on_device) || (!flat && p->seq_bytes >= ((size_t)(ENV(SHARED_UP_MIN_MB) ? imax(atoi(ENV(SHARED_UP_MIN_MB)), 0) : 4) << 20) && !ENV(NO_SHARED_UP)) ? shared_upload_stream() : 0;
	if (up) shared_up = 1; else up = g_plan_stream ? g_plan_stream : thread_upload_stream();
	p->stream = up; p->stream_used = 1;              /* plan_destroy waits for it before the buffers are recycled */
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
			c->sc.pk_a = pkinfo[c->generic].a; c->sc.pk_b = pkinfo[c->generic].b; c->sc.pk_n = pkinfo[c->generic].n;
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
			           (forced < 0 ? k2a_pkcfg_G[c->cfg] == 64 && (k2a_shim_simd_count() > 0 ? 2 * (int64_t)c->count >= 3 * (int64_t)k2a_shim_simd_count() : c->count >= 32) : forced);
		}
		for (lo = 0; lo < 2; ++lo) {                       /* 0: size it, 1: lay it out */
			size_t at = p->tb_bytes;
			for (k = 0; k < p->ncls; ++k) {
				const cls_t *c = &p->cls[k];
				const int G = c->pk ? k2a_pkcfg_G[c->cfg] : 64, C = c->pk ? k2a_pkcfg_C[c->cfg] : 16, NG = 64 / G;
				int t0;
				if (!c->defer) continue;
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
				p->h_qd[c->qd].pad = (uint32_t)at;             /* (host side only: where this class's piece counts start) */
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
			p->h_qd[k].need = p->d_order + p->norder + p->h_qd[k].pad; p->h_qd[k].pad = 0;
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
		    (!all_queues && k2a_shim_memset(p->d_res, 0, sizeof(K2aResult) * (size_t)n, up)) ||
		    (p->bnd_words && k2a_shim_memset(p->d_bnd, 0xC0, p->bnd_words * 4, up)) ||
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
		if (shared_up) pthread_mutex_lock(&g_shared_issue_mu);      /* one plan's copies in one piece */
		if (!p->up_ev ||
		    (flat && ((flat->on_device ? k2a_shim_d2d(p->d_seq, flat_lo, flat_span, up) : k2a_shim_h2d(p->d_seq, flat_lo, flat_span, up)) ||
		              k2a_shim_h2d(p->d_seq + flat_span, hm + align_up(b_meta, 256), tail, up) ||
		              k2a_shim_h2d(p->d_pairs, hm, b_meta, up))) ||
		    (!flat && k2a_shim_h2d(p->d_seq, p->h_seq, meta_off + b_meta, up)) ||
		    (need_clear && k2a_shim_memset(p->d_res, 0, sizeof(K2aResult) * (size_t)n, up)) ||
		    (p->bnd_words && k2a_shim_memset(p->d_bnd, 0xC0, p->bnd_words * 4, up)) ||
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

ksw2amd_plan_t *ksw2amd_plan_create(int dual, const ksw2amd_scoring_t *sc, int n, const ksw2amd_pair_t *pairs)
{
	return plan_create_ex(dual, 0, sc, n, pairs, 0, 0);      /* (KSW2AMD_STREAM=1 streams these too: tests, A/B runs) */
}

static int exts_plan_run(ksw2amd_plan_t *p, void *stream);
static int extf_plan_run(ksw2amd_plan_t *p, void *stream);
static int ssec_plan_run(ksw2amd_plan_t *p, void *stream);

int ksw2amd_plan_run(ksw2amd_plan_t *p, void *stream)
{
	int c, streaming = 0, nrest = 0;
	if (!p) return fail(KSW2AMD_E_PARAM, "plan_run: NULL plan%s", 0);
	if (p->splice == 3) return ssec_plan_run(p, stream);
	if (p->splice == 2) return extf_plan_run(p, stream);
	if (p->splice) return exts_plan_run(p, stream);
	p->stream = stream; p->ran = 1; p->stream_used = 1;
	if (p->gather && (!k2a_shim_async_launches() || !p->streamed)) gather_wait(p);      /* an ordinary launch, or one that runs inside the call, needs the whole arena */
	streaming = p->streamed && p->nqd > 0;
	if (p->up_ev && !streaming && k2a_shim_stream_wait_event(stream, p->up_ev)) goto err;      /* the plan's upload (shared stream) before its kernels */
	if (p->reject_all || p->ntasks == 0) return KSW2AMD_OK;
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
		static const char *const fkind[7] = { "extf-lds", "extf-lds", "extf-lds", "extf-hbm", "extf-win4", "extf-win8", "extf-lane" };
		int nl = 0;
		for (c = 0; c < 7 && len < cap - 1; ++c)
			if (p->f_count[c]) {
				len += snprintf(buf + len, (size_t)(cap - len), "kernel=%s form=%s ring=%d tasks=%d\n", fkind[c], c == 6 && p->f_par.ring ? "ldsring" : c == 6 ? "hbm" : "-",
				                c == 6 ? p->f_par.ring : 0, p->f_count[c]);
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
		len += snprintf(buf + len, (size_t)(cap - len), "kernel=%s G=%d C=%d gaps=%d mode=%s rebased=%d nomax=%d generic=%d form=%s tasks=%d\n",
		                kind, G, C, p->dual ? 2 : 1, mode_name[k->mode], k->rb, k->nomax, k->generic, form_name[form], k->count);
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
			if (trace_level()) fprintf(stderr, "[ksw2_amd] streamed plan n=%d: a launch gave up waiting for its inputs; running the plan again behind its upload\n", p->n);
			p->streamed = 0;
			if ((p->up_ev && k2a_shim_event_sync(p->up_ev)) || k2a_shim_memset(p->d_res, 0, sizeof(K2aResult) * (size_t)p->n, st) ||
			    ksw2amd_plan_run(p, st) || k2a_shim_stream_sync(st))
				return fail(KSW2AMD_E_NODEVICE, "plan_fetch: %s", k2a_shim_last_error());
		}
	}
	if (k2a_shim_d2h(p->h_res, p->d_res, sizeof(K2aResult) * (size_t)p->n, p->stream) ||
	    k2a_shim_stream_sync(p->stream))
		return fail(KSW2AMD_E_NODEVICE, "plan_fetch: %s", k2a_shim_last_error());
	return KSW2AMD_OK;
}

static int needs_rerun(const ksw2amd_plan_t *p, int i);
static int pair_rerun(ksw2amd_plan_t *p, int i, void *km, ksw_extz_t *z);
static int64_t g_reruns;                   /* pairs that fetch ran again (diagnostics: ksw2amd_rerun_count) */
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
static void eqx_rewrite(void *km, const uint8_t *query, const uint8_t *target, int stride, ksw_extz_t *ez)
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
static void km_lock(const void *km) { if (km) pthread_mutex_lock(km_mutex(km)); }
static void km_unlock(const void *km) { if (km) pthread_mutex_unlock(km_mutex(km)); }

/* Pairs whose device result cannot be used are run again through the ordinary gather path, one by one:
 *   K2aResult.pad[0] -- flat plans: a packed kernel met a wildcard code (K2aLanePk::seen); the gather path's scan sends the pair to
 *                       the int32 kernels.  Host arenas read the sequences where they lie, device arenas bring them back first;
 *   K2aResult.pad[1] -- deferred arg-max: a Z-drop could not be ruled out without the arg-max columns; the re-run keeps them. */
static int run_serial(int dual, int scalar, void *km, const ksw2amd_scoring_t *sc, int n, const ksw2amd_pair_t *pairs, ksw_extz_t *ez, int share, const flat_src_t *flat, int want_stream);
static int needs_rerun(const ksw2amd_plan_t *p, int i) { return p->h_cls[i] >= 0 && !p->splice && (((p->flat || p->unscanned) && p->h_res[i].pad[0]) || p->h_res[i].pad[1]); }
static int pair_rerun(ksw2amd_plan_t *p, int i, void *km, ksw_extz_t *z)
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
	if (p->flat_device) {
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
typedef struct { ksw2amd_plan_t *p; void *km; ksw_extz_t *ez, **ezp; void **kmp; const uint32_t *pool; const size_t *pos; int nrerun, rc; } asm_ctx_t;
static void assemble_range(asm_ctx_t *c, int beg, int end)
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
static int assemble_parallel(asm_ctx_t *c);
static int rerun_pairs(ksw2amd_plan_t *p, int nrerun, void *km, ksw_extz_t *ez, ksw_extz_t **ezp, void **kmp);

static int plan_fetch_ex(ksw2amd_plan_t *p, void *km, ksw_extz_t *ez, ksw_extz_t **ezp, void **kmp)
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
static size_t thread_cached_device_bytes(void)
{
	size_t b = 0;
	int k;
	for (k = 0; k < BUF_KINDS; ++k) { int d; for (d = 0; d < CACHE_DEPTH; ++d) if (!BUF_IS_HOST(k) && g_cache[k][d].p) b += g_cache[k][d].cap; }
	return b;
}
static size_t device_budget(size_t free_b, size_t total_b, int share)
{
	const size_t fair = total_b / 10 * 7 / (size_t)(share > 0 ? share : 1), have = (free_b + thread_cached_device_bytes()) / 10 * 9;
	return fair < have ? fair : have;
}

static size_t pair_device_bytes(int dual, const ksw2amd_pair_t *a)
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
static double now_ms(void);
static int trace_on(void);
static int unit_pairs(const ksw2amd_pair_t *a);
static int run_serial(int dual, int scalar, void *km, const ksw2amd_scoring_t *sc, int n, const ksw2amd_pair_t *pairs, ksw_extz_t *ez, int share, const flat_src_t *flat, int want_stream)
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

/* ---------------------------------------------------------------- worker pool: chunked, pipelined, multi-device batches
 * A large batch handed to one ksw2amd_ext?_batch call is cut into chunks of consecutive pairs that a few persistent worker
 * threads pull from a shared counter.  Every worker packs, uploads, computes and fetches on a stream (and with pinned staging
 * and device buffers) of its own, so chunk i+1 is packed and uploaded while chunk i computes and chunk i-1's results come
 * back: one calling thread gets the device-bound rate instead of the sum of the phases.  With ksw2amd_set_devices() the
 * workers belong to several GPUs and the same counter shards the batch over them (pairs are independent: no collective). */
#define POOL_MAXW 64
#define POOL_MAXDEV 16
typedef struct { ksw2amd_plan_t *p; int beg; } pend_t;      /* a worker's plan that is computing while the worker packs the next chunk */
typedef int (*chunk_fn)(void *ctx, int beg, int end, int share, pend_t *pd);   /* beg < 0: finish what is pending */
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
	int dev[POOL_MAXW];
	job_t *job;
} g_pool = { PTHREAD_MUTEX_INITIALIZER, PTHREAD_COND_INITIALIZER, PTHREAD_COND_INITIALIZER, PTHREAD_COND_INITIALIZER, 0, 0, 0, {0}, 0 };
static int64_t g_stat[4];                          /* pooled batches, their chunks, coalesced single calls, the batches they formed */
static int g_ndev_set, g_dev_set[POOL_MAXDEV];     /* ksw2amd_set_devices(); 0 = the calling thread's device */
static __thread int g_is_worker;

static int job_has_dev(const job_t *j, int dev)
{
	int i;
	for (i = 0; i < j->ndev; ++i) if (j->dev[i] == dev) return 1;
	return 0;
}

typedef struct { int idx, dev, seen; } worker_arg_t;

static void *pool_worker(void *arg_)
{
	worker_arg_t *arg = (worker_arg_t*)arg_;
	const int dev = arg->dev, rank = arg->idx;           /* rank among the workers of this device */
	int seen = arg->seen;
	free(arg);
	g_is_worker = 1;
	k2a_shim_set_device(dev);
	pthread_mutex_lock(&g_pool.mu);
	for (;;) {
		job_t *j;
		while (g_pool.gen == seen) pthread_cond_wait(&g_pool.work, &g_pool.mu);
		seen = g_pool.gen; j = g_pool.job;
		if (!j || !job_has_dev(j, dev)) continue;
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

static int pool_threads_per_device(void)
{
	const char *e = ENV(THREADS);
	int t = e ? atoi(e) : 6;
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
			g_pool.dev[g_pool.nw++] = j->dev[d];
		}
	}
	for (i = 0, j->pending = 0; i < g_pool.nw; ++i) j->pending += job_has_dev(j, g_pool.dev[i]);
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

static void copy_range(const copy_ctx_t *c, int beg, int end)
{
	int i;
	for (i = beg; i < end; ++i) {
		const ksw2amd_pair_t *a = &c->pairs[i];
		if (a->qlen <= 0 || a->tlen <= 0) continue;
		if (c->wild) c->wild[i] = (uint8_t)(copy_scan(c->h_seq + c->hp[i].qoff, a->query, a->qlen) | copy_scan(c->h_seq + c->hp[i].toff, a->target, a->tlen));
		else { memcpy(c->h_seq + c->hp[i].qoff, a->query, (size_t)a->qlen); memcpy(c->h_seq + c->hp[i].toff, a->target, (size_t)a->tlen); }   /* unscanned (streamed plans) */
		memset(c->h_seq + c->hp[i].toff + a->tlen, 0, 64);                                          /* rows read past the target end */
	}
}
static int copy_chunk(void *ctx, int beg, int end, int share, pend_t *pd)
{
	const copy_ctx_t *c = (const copy_ctx_t*)ctx;
	(void)share; (void)pd;
	if (beg >= 0) {
		copy_range(c, beg, end);
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
static int parallel_copy(copy_ctx_t *c, int n, size_t bytes)
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
static int assemble_parallel(asm_ctx_t *c)
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
static int rerun_pairs(ksw2amd_plan_t *p, int nrerun, void *km, ksw_extz_t *ez, ksw_extz_t **ezp, void **kmp)
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
static int gather_start(ksw2amd_plan_t *p, stream_up_t *su, const ksw2amd_pair_t *pairs, int n)
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
static int gather_wait(ksw2amd_plan_t *p)
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

static int pool_min_pairs(void)
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
static int unit_pairs(const ksw2amd_pair_t *a)
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
static int uniform_chunks(int n, int unit, double bytes, double cells, int workers, int ndev, int with_cigar, double path_steps, int *chunk_pairs)
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
static double now_ms(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6; }
static int64_t now_ns(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return (int64_t)ts.tv_sec * 1000000000 + ts.tv_nsec; }
static int trace_on(void) { return ENV(TRACE) != 0; }

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
static int run_pooled(chunk_fn fn, void *ctx, int n, const double *cost, double total, int nchunks, int chunk_pairs, int *rc)
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

static int run_batch(int dual, int scalar, void *km, const ksw2amd_scoring_t *sc, int n, const ksw2amd_pair_t *pairs, ksw_extz_t *ez, const flat_src_t *flat)
{
	const int tpd = pool_threads_per_device();
	if (n <= 0) return KSW2AMD_OK;
	if (k2a_shim_device_count() <= 0) return fail(KSW2AMD_E_NODEVICE, "no usable %s device", k2a_shim_backend());
	if (n >= (pool_min_pairs() ? pool_min_pairs() : 512) && tpd > 0 && !g_is_worker) {
		const int workers = tpd * (g_ndev_set > 0 ? g_ndev_set : 1);
		double *cost, bytes = 0, cells = 0, total = 0, path = 0, dev_bytes = 0;
		int i, nchunks, rc = 0, uniform = 1, chunk_pairs = 0;
		/* one shape for the whole batch (the BASELINE configurations, reads trimmed to one length)?  Then the sums below are n x
		 * the first pair's terms and the per-pair costs are never looked at (uniform_chunks cuts at fixed sizes): this loop was
		 * 1.5-2 ms of the calling thread's time on config 2's 65 536 pairs, in front of a 1.4 ms kernel */
		for (i = 1; i < n; ++i)
			if (pairs[i].qlen != pairs[0].qlen || pairs[i].tlen != pairs[0].tlen || pairs[i].w != pairs[0].w || ((pairs[i].flag ^ pairs[0].flag) & KSW_EZ_SCORE_ONLY)) { uniform = 0; break; }
		if (uniform) dev_bytes = (double)n * (double)pair_device_bytes(dual, &pairs[0]);
		else for (i = 0; i < n; ++i) dev_bytes += (double)pair_device_bytes(dual, &pairs[i]);
		/* One-shape score-only batches that fit the device: ONE streamed plan (section "streamed plans") instead of chunks -- a
		 * single launch over the whole batch, started under the upload, whose wavefronts wait for their pieces, longest task first, at full
		 * occupancy.  Where it pays is where the kernels are long against the host's per-pair work: the 10 k headline (MI355X, round 4,
		 * same box: 4 060 against 3 800 GCUPS through the pointer entry, 4 510 against 4 270 through the flat one), not 512-base reads,
		 * whose plan creation on one thread costs what six workers' chunks cost together (config 2: 3.9 against 3.4-4.0 ms) -- so by
		 * default batches of at least 1 M cells per pair.  KSW2AMD_STREAM=1: every one-shape score-only batch; =0: chunks. */
		if (uniform && (pairs[0].flag & KSW_EZ_SCORE_ONLY) && stream_env() != 0 && g_ndev_set <= 1 && !pool_min_pairs() &&
		    (double)n * ((double)imax(pairs[0].qlen, 0) + imax(pairs[0].tlen, 0)) >= 4.0 * 1048576.0) {
			const int mx0 = imax(pairs[0].qlen, pairs[0].tlen);
			const int64_t c0 = pairs[0].qlen > 0 && pairs[0].tlen > 0 ? band_cells(pairs[0].qlen, pairs[0].tlen, (pairs[0].w < 0 || pairs[0].w > mx0) ? mx0 : pairs[0].w) : 0;
			size_t free_b = 0, total_b = 0;
			if ((stream_env() == 1 || c0 >= stream_min_cells()) &&
			    (dev_bytes <= 256e6 || (k2a_shim_mem_info(&free_b, &total_b) == 0 && dev_bytes <= (double)device_budget(free_b, total_b, 1))))
				return run_serial(dual, scalar, km, sc, n, pairs, ez, 1, flat, 1);
		}
		cost = (double*)malloc(sizeof(double) * (size_t)n);
		if (cost) {
			if (dev_bytes > 64e9 && !pool_min_pairs()) {
				/* traceback memory is what splits this batch: one plan at a time with the whole device, not a slice per worker */
				size_t free_b = 0, total_b = 0;
				if (k2a_shim_mem_info(&free_b, &total_b) == 0 && dev_bytes > 0.5 * (double)total_b && g_ndev_set <= 1) { free(cost); return run_serial(dual, scalar, km, sc, n, pairs, ez, 1, flat, 0); }
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

static int wants_ssec(int flag);
static int ssec_run(int dual, void *km, const ksw2amd_scoring_t *sc, int n, const ksw2amd_pair_t *pairs, ksw_extz_t *ez);

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

/* ---------------------------------------------------------------- the ksw2-named single-pair calls */

/* The ksw2 signatures return void: a HIP failure (no device, out of memory) has no channel, and there is no CPU fallback to hide
 * it behind.  A failing call never crosses the C boundary with an abort or a made-up result: it returns with `ez` reset (score =
 * KSW_NEG_INF, no CIGAR -- "no alignment" in the reference's own terms, ksw2.h:184-189), counts itself (ksw2amd_error_count),
 * leaves its message in ksw2amd_last_error() and says so on stderr (the first few times, then every 1000th).  A caller that wants
 * to know at once installs a handler (function name, KSW2AMD_E_* code, message); KSW2AMD_ABORT_ON_ERROR=1 aborts instead. */
static long g_small_calls;
long ksw2amd_small_call_count(void) { return g_small_calls; }
static ksw2amd_error_fn g_err_fn;
static void *g_err_user;
static long g_err_count;

void ksw2amd_set_error_handler(ksw2amd_error_fn fn, void *user) { g_err_fn = fn; g_err_user = user; }
long ksw2amd_error_count(void) { return g_err_count; }

static void call_failed(const char *fn, int code, ksw_extz_t *ez)
{
	__sync_fetch_and_add(&g_err_count, 1);
	if (ez) ez_reset(ez);
	if (g_err_fn) { g_err_fn(fn, code, g_err, g_err_user); return; }
	if (g_err_count <= 8 || g_err_count % 1000 == 0)
		fprintf(stderr, "[ksw2_amd] %s failed (%ld so far): %s -- libksw2_amd has no CPU fallback; *ez is reset (score = KSW_NEG_INF)\n", fn, g_err_count, g_err);
	if (env_flag(ENV(ABORT_ON_ERROR), 0)) abort();
}

static int queue_one(const char *fn, int dual, void *km, const ksw2amd_scoring_t *sc, const ksw2amd_pair_t *pr, ksw_extz_t *ez);

/* ---------------------------------------------------------------- opt-in host path for tiny single calls
 * One pair per call from one thread (cli.c:50-132, README.md:54-87 of the reference) costs a launch and two PCIe round trips
 * here -- ~0.5 ms whatever the size -- where the reference's SSE kernel needs tens of microseconds for a few hundred bases, and
 * coalescing concurrent callers (below) only helps callers that are concurrent.  A caller that has such calls can hand the
 * small ones to this code instead: ksw2amd_set_small_call_cells(c) / KSW2AMD_SMALL_CELLS=c sends every call of the ksw2-named
 * single-pair entry points whose exact band has at most c cells through the scalar routine below, on the calling thread.
 * OFF by default (c = 0), never taken because something failed, never used by the batch entry points, and only in a process
 * whose device works (the first call still initialises it: without a GPU the library fails as loudly as before).
 * It is the product's own restatement of the result contract (ksw2_extz.c:38-133, ksw2_extd.c:42-173 are
 * the loops it replaces), written against the same rules as the kernels -- cell update, direction byte and continuation bits,
 * row maxima with their tie rules, per-row bookkeeping and Z-drop, start of the traceback, =/X rewrite -- and tested like them:
 * against the oracle and the golden vectors (tests/test_small_calls.py), never through the oracle. */
static volatile int64_t g_small_cells = -1;              /* -1: not set by the API, KSW2AMD_SMALL_CELLS decides */
void ksw2amd_set_small_call_cells(int64_t cells) { g_small_cells = cells < 0 ? 0 : cells; }
static int64_t small_cells_limit(void)
{
	if (g_small_cells >= 0) return g_small_cells;
	return ENV(SMALL_CELLS) ? atoll(ENV(SMALL_CELLS)) : 0;
}
static int small_border(int dual, int q, int e, int q2, int e2, int k)       /* H on the virtual row / column -1 at distance k */
{
	int a = -(q + k * e);
	if (dual) { const int b = -(q2 + k * e2); if (b > a) a = b; }
	return k <= 0 ? 0 : a;
}
/* returns a KSW2AMD_* code; *z is complete on KSW2AMD_OK */
static int small_pair(int dual, int scalar, void *km, const ksw2amd_scoring_t *sc, const ksw2amd_pair_t *a, ksw_extz_t *z)
{
	const int m = sc->m, qlen = a->qlen, tlen_full = a->tlen, fl = a->flag;
	int q = sc->q, e = sc->e, q2 = sc->q2, e2 = sc->e2, w = a->w, i, j, k, lo;
	const int generic = (fl & KSW_EZ_GENERIC_SC) || scalar, right = !!(fl & KSW_EZ_RIGHT), score_only = !!(fl & KSW_EZ_SCORE_ONLY);
	const int with_tb = !score_only, firstj = !dual && right && with_tb;          /* extz + RIGHT + CIGAR: row-maximum ties to the first column */
	int8_t eff[K2A_MAXM * K2A_MAXM];
	int32_t *H, *E, *E2;
	uint8_t *dir = 0;
	int tlen, bw, bmax = 0, bmax_t = -1, bmax_q = -1, bmqe = K2A_NEG, bmqe_t = -1, bmte = K2A_NEG, bmte_q = -1, bscore = K2A_NEG, bdrop = 0;
	int ti = -1, tj = -1, reach_end = 0, zslope;
	ez_reset(z);
	if (m <= 0 || (dual && m <= 1) || !sc->mat || qlen <= 0 || tlen_full <= 0) return KSW2AMD_OK;     /* ksw2_extz2_sse.c:57, ksw2_extd2_sse.c:76 */
	if (m > K2A_MAXM) return fail(KSW2AMD_E_PARAM, "more than 127 residue types (int8_t m, ksw2.h:61)%s", 0);
	if (!a->query || !a->target) return fail(KSW2AMD_E_PARAM, "NULL sequence%s", 0);
	if (dual && !scalar && q2 + e2 < q + e) { int t = q; q = q2; q2 = t; t = e; e = e2; e2 = t; }      /* ksw2_extd2_sse.c:78 */
	for (k = 1, lo = sc->mat[m * m > 1 ? 1 : 0]; k < m * m; ++k) lo = imin(lo, sc->mat[k]);
	if (!scalar && -lo > 2 * (q + e)) return KSW2AMD_OK;                                              /* ksw2_extz2_sse.c:78-82 */
	build_eff(dual, m, sc->mat, e, e2, generic, eff);
	{ const int mx = imax(qlen, tlen_full); if (w < 0 || w > mx) w = mx; }                            /* ksw2_extz2_sse.c:72 */
	tlen = (int64_t)qlen + w < tlen_full ? qlen + w : tlen_full;                                      /* rows i with i - w <= qlen - 1 */
	bw = imin(qlen, 2 * w + 1);
	zslope = dual ? e2 : e;
	H = (int32_t*)malloc(sizeof(int32_t) * 3 * ((size_t)qlen + 2));
	if (with_tb) dir = (uint8_t*)malloc((size_t)tlen * (size_t)bw + 1);
	if (!H || (with_tb && !dir)) { free(H); free(dir); return fail(KSW2AMD_E_NOMEM, "small call: host allocation failed%s", 0); }
	E = H + qlen + 2; E2 = E + qlen + 2;
	/* row -1: H(-1, j) and the gap states it opens into row 0 (ksw2_extz.c:32-35, ksw2_extd.c:33-41) */
	for (j = 0; j < qlen; ++j) {
		if (j <= w) { const int hb = small_border(dual, q, e, q2, e2, j + 1); H[j] = hb; E[j] = hb - (q + e); E2[j] = hb - (q2 + e2); }
		else H[j] = E[j] = E2[j] = K2A_NEG;
	}
	for (i = 0; i < tlen && !bdrop; ++i) {
		const int st = imax(0, i - w), en = imin(qlen - 1, i + w), reach = i + w >= qlen - 1;
		const uint8_t tc = a->target[i];
		const int8_t *srow = eff + (size_t)tc * m;
		int hdiag = st == 0 ? small_border(dual, q, e, q2, e2, i) : H[st - 1];         /* H(i-1, st-1) */
		int f, f2, rm = K2A_NEG, rj = -1, hend = K2A_NEG;
		uint8_t *drow = with_tb ? dir + (size_t)i * bw : 0;
		if (st == 0) { const int hb = small_border(dual, q, e, q2, e2, i + 1); f = hb - (q + e); f2 = hb - (q2 + e2); }      /* column -1 (ksw2_extz.c:43-44) */
		else f = f2 = K2A_NEG;
		for (j = st; j <= en; ++j) {
			int h = hdiag + srow[a->query[j]], ee = E[j], ee2 = dual ? E2[j] : K2A_NEG, t, ex, fx;
			unsigned d = 0;
			if (i > 0 && j - i >= w) { ee = K2A_NEG; ee2 = K2A_NEG; }                   /* the cell above is outside the band */
			hdiag = H[j];                                                             /* H(i-1, j): the next column's diagonal */
			if (!right) {                                                             /* ksw2_extz.c:72-75, ksw2_extd.c:88-95 */
				d = h >= ee ? 0u : 1u; h = imax(h, ee);
				d = h >= f ? d : 2u;   h = imax(h, f);
				if (dual) { d = h >= ee2 ? d : 3u; h = imax(h, ee2); d = h >= f2 ? d : 4u; h = imax(h, f2); }
			} else {                                                                  /* ksw2_extz.c:98-101, ksw2_extd.c:126-133 */
				d = h > ee ? 0u : 1u;  h = imax(h, ee);
				d = h > f ? d : 2u;    h = imax(h, f);
				if (dual) { d = h > ee2 ? d : 3u; h = imax(h, ee2); d = h > f2 ? d : 4u; h = imax(h, f2); }
			}
			if (firstj ? h > rm : h >= rm) rj = j;                                     /* SURVEY 8a rule 3 */
			rm = imax(rm, h);
			t = h - (q + e); ex = ee - e; fx = f - e;
			if (!right) { d |= (ex > t ? 1u : 0u) << 3; d |= (fx > t ? 1u : 0u) << 4; }
			else { d |= (ex >= t ? 1u : 0u) << 3; d |= (fx >= t ? 1u : 0u) << 4; }
			E[j] = imax(ex, t); f = imax(fx, t);
			if (dual) {
				const int t2 = h - (q2 + e2), ex2 = ee2 - e2, fx2 = f2 - e2;
				if (!right) { d |= (ex2 > t2 ? 1u : 0u) << 5; d |= (fx2 > t2 ? 1u : 0u) << 6; }
				else { d |= (ex2 >= t2 ? 1u : 0u) << 5; d |= (fx2 >= t2 ? 1u : 0u) << 6; }
				E2[j] = imax(ex2, t2); f2 = imax(fx2, t2);
			}
			H[j] = h;
			if (drow) drow[j - st] = (uint8_t)d;
			hend = h;
		}
		if (st > 0) H[st - 1] = K2A_NEG;                                                 /* left the band */
		/* the row's epilogue (ksw2_extz.c:116-124, ksw2_extd.c:156-164; ksw2.h:191-207 with is_rot = 0) */
		if (reach && hend > bmqe) { bmqe = hend; bmqe_t = i; }
		if (i == tlen_full - 1) { bmte = rm; bmte_q = rj; }
		if (rm > bmax) { bmax = rm; bmax_t = i; bmax_q = rj; }
		else if (i >= bmax_t && rj >= bmax_q) {
			const int dt = i - bmax_t, dq = rj - bmax_q, skew = dt > dq ? dt - dq : dq - dt;
			if (a->zdrop >= 0 && bmax - rm > a->zdrop + skew * zslope) bdrop = 1;
		}
		if (!bdrop && i == tlen_full - 1 && reach) bscore = hend;
	}
	/* rows, or the corner column, that the band cannot reach: stop like the SSE kernels do (ksw2_extz2_sse.c:111-114) */
	if (!bdrop && (tlen < tlen_full || (tlen_full - 1) + w < qlen - 1)) bdrop = 1;
	z->max = (uint32_t)bmax; z->zdropped = (uint32_t)bdrop; z->max_q = bmax_q; z->max_t = bmax_t;
	z->mqe = bmqe; z->mqe_t = bmqe_t; z->mte = bmte; z->mte_q = bmte_q; z->score = bscore;
	/* start of the traceback (ksw2_extz2_sse.c:292-301 / ksw2_extz.c:127-133) */
	if (score_only) { }
	else if (!bdrop && !(fl & KSW_EZ_EXTZ_ONLY)) { ti = tlen_full - 1; tj = qlen - 1; }
	else if (!bdrop && (fl & KSW_EZ_EXTZ_ONLY) && bmqe + (scalar ? K2A_NEG : a->end_bonus) > bmax) { reach_end = 1; ti = bmqe_t; tj = qlen - 1; }
	else if (bmax_t >= 0 && bmax_q >= 0) { ti = bmax_t; tj = bmax_q; }
	z->reach_end = reach_end;
	if (ti >= 0 && tj >= 0) {                                                           /* ksw_backtrack, ksw2.h:129-161 */
		uint32_t *cg = (uint32_t*)malloc(sizeof(uint32_t) * ((size_t)qlen + tlen_full + 2));
		uint32_t last_op = 0xffffffffu, run = 0;
		int n = 0, state = 0;
		if (!cg) { free(H); free(dir); return fail(KSW2AMD_E_NOMEM, "small call: host allocation failed%s", 0); }
		i = ti; j = tj;
		while (i >= 0 && j >= 0) {
			const unsigned d = dir[(size_t)i * bw + (j - imax(0, i - w))];
			uint32_t op;
			if (state == 0) state = d & 7;
			else if (!((d >> (state + 2)) & 1)) state = 0;
			if (state == 0) state = d & 7;
			if (state == 0) { op = 0; --i; --j; }
			else if (state == 1 || state == 3) { op = 2; --i; }
			else { op = 1; --j; }
			if (op == last_op) ++run;
			else { if (run) cg[n++] = run << 4 | last_op; last_op = op; run = 1; }
		}
		if (i >= 0) { if (last_op == 2) run += (uint32_t)i + 1; else { if (run) cg[n++] = run << 4 | last_op; last_op = 2; run = (uint32_t)i + 1; } }
		if (j >= 0) { if (last_op == 1) run += (uint32_t)j + 1; else { if (run) cg[n++] = run << 4 | last_op; last_op = 1; run = (uint32_t)j + 1; } }
		if (run) cg[n++] = run << 4 | last_op;
		if (n > 0) {
			km_lock(km);
			ez_reserve(km, z, n);
			for (k = 0; k < n; ++k) z->cigar[k] = cg[(fl & KSW_EZ_REV_CIGAR) ? k : n - 1 - k];           /* ksw2.h:157-159 */
			z->n_cigar = n;
			if (dual && (fl & KSW_EZ_EQX) && !scalar) eqx_rewrite(km, a->query, a->target, 1, z);
			km_unlock(km);
		}
		free(cg);
	}
	free(H); free(dir);
	return KSW2AMD_OK;
}


static void one_pair(const char *fn, int dual, int scalar, void *km, int qlen, const uint8_t *query, int tlen, const uint8_t *target, int8_t m,
                     const int8_t *mat, int8_t q, int8_t e, int8_t q2, int8_t e2, int w, int zdrop, int end_bonus, int flag,
                     ksw_extz_t *ez)
{
	ksw2amd_scoring_t sc;
	ksw2amd_pair_t pr;
	int rc;
	sc.m = m; sc.mat = mat; sc.q = q; sc.e = e; sc.q2 = q2; sc.e2 = e2;
	pr.query = query; pr.target = target; pr.qlen = qlen; pr.tlen = tlen;
	pr.w = w; pr.zdrop = zdrop; pr.end_bonus = end_bonus; pr.flag = flag & ~F_SCALAR_CONTRACT;
	if (!scalar && wants_ssec(pr.flag)) {                             /* the SSE kernels' own results (ksw2_lane_ssec.h) */
		rc = ssec_run(dual, km, &sc, 1, &pr, ez);
		if (rc != KSW2AMD_OK) call_failed(fn, rc, ez);
		return;
	}
	{	/* opt-in: tiny pairs on the calling thread (small_pair); the device is brought up first all the same */
		const int64_t lim = small_cells_limit();
		const int wn = (w < 0 || w > imax(qlen, tlen)) ? imax(qlen, tlen) : w;
		if (lim > 0 && !is_approx(pr.flag | (scalar ? F_SCALAR_CONTRACT : 0)) && qlen > 0 && tlen > 0 && band_cells(qlen, tlen, wn) <= lim && thread_stream()) {
			rc = small_pair(dual, scalar, km, &sc, &pr, ez);
			if (rc != KSW2AMD_OK) call_failed(fn, rc, ez);
			else __sync_fetch_and_add(&g_small_calls, 1);
			return;
		}
	}
	if (!scalar && queue_one(fn, dual, km, &sc, &pr, ez)) return;     /* coalesced with other threads' calls */
	g_latency_plan = 1;
	rc = run_serial(dual, scalar, km, &sc, 1, &pr, ez, 1, 0, 0);
	g_latency_plan = 0;
	if (rc != KSW2AMD_OK) call_failed(fn, rc, ez);
}

void ksw_extz2_sse(void *km, int qlen, const uint8_t *query, int tlen, const uint8_t *target, int8_t m, const int8_t *mat,
                   int8_t q, int8_t e, int w, int zdrop, int end_bonus, int flag, ksw_extz_t *ez)
{
	one_pair("ksw_extz2_sse", 0, 0, km, qlen, query, tlen, target, m, mat, q, e, 0, 0, w, zdrop, end_bonus, flag, ez);
}

void ksw_extd2_sse(void *km, int qlen, const uint8_t *query, int tlen, const uint8_t *target, int8_t m, const int8_t *mat,
                   int8_t q, int8_t e, int8_t q2, int8_t e2, int w, int zdrop, int end_bonus, int flag, ksw_extz_t *ez)
{
	one_pair("ksw_extd2_sse", 1, 0, km, qlen, query, tlen, target, m, mat, q, e, q2, e2, w, zdrop, end_bonus, flag, ez);
}

void ksw_extz2_sse41(void *km, int qlen, const uint8_t *query, int tlen, const uint8_t *target, int8_t m, const int8_t *mat,
                     int8_t q, int8_t e, int w, int zdrop, int end_bonus, int flag, ksw_extz_t *ez)
{ ksw_extz2_sse(km, qlen, query, tlen, target, m, mat, q, e, w, zdrop, end_bonus, flag, ez); }
void ksw_extz2_sse2(void *km, int qlen, const uint8_t *query, int tlen, const uint8_t *target, int8_t m, const int8_t *mat,
                    int8_t q, int8_t e, int w, int zdrop, int end_bonus, int flag, ksw_extz_t *ez)
{ ksw_extz2_sse(km, qlen, query, tlen, target, m, mat, q, e, w, zdrop, end_bonus, flag, ez); }
void ksw_extd2_sse41(void *km, int qlen, const uint8_t *query, int tlen, const uint8_t *target, int8_t m, const int8_t *mat,
                     int8_t q, int8_t e, int8_t q2, int8_t e2, int w, int zdrop, int end_bonus, int flag, ksw_extz_t *ez)
{ ksw_extd2_sse(km, qlen, query, tlen, target, m, mat, q, e, q2, e2, w, zdrop, end_bonus, flag, ez); }
void ksw_extd2_sse2(void *km, int qlen, const uint8_t *query, int tlen, const uint8_t *target, int8_t m, const int8_t *mat,
                    int8_t q, int8_t e, int8_t q2, int8_t e2, int w, int zdrop, int end_bonus, int flag, ksw_extz_t *ez)
{ ksw_extd2_sse(km, qlen, query, tlen, target, m, mat, q, e, q2, e2, w, zdrop, end_bonus, flag, ez); }

/* scalar-named entry points: matrix always used as given, no end bonus, no mismatch-vs-gap reject */
void ksw_extz(void *km, int qlen, const uint8_t *query, int tlen, const uint8_t *target, int8_t m, const int8_t *mat,
              int8_t q, int8_t e, int w, int zdrop, int flag, ksw_extz_t *ez)
{
	one_pair("ksw_extz", 0, 1, km, qlen, query, tlen, target, m, mat, q, e, 0, 0, w, zdrop, 0, flag, ez);
}

void ksw_extd(void *km, int qlen, const uint8_t *query, int tlen, const uint8_t *target, int8_t m, const int8_t *mat,
              int8_t q, int8_t e, int8_t q2, int8_t e2, int w, int zdrop, int flag, ksw_extz_t *ez)
{
	one_pair("ksw_extd", 1, 1, km, qlen, query, tlen, target, m, mat, q, e, q2, e2, w, zdrop, 0, flag, ez);
}

static int global_align(const char *fn, void *km, int qlen, const uint8_t *query, int tlen, const uint8_t *target, int8_t m,
                        const int8_t *mat, int8_t q, int8_t e, int w, int *m_cigar_, int *n_cigar_, uint32_t **cigar_)
{
	/* ksw2_gg.c:6-102 == extension kernel with Z-drop off, matrix scoring, left-aligned gaps, corner start */
	ksw_extz_t ez;
	const int with_cigar = m_cigar_ && n_cigar_ && cigar_;
	memset(&ez, 0, sizeof(ez));
	if (with_cigar) { ez.cigar = *cigar_; ez.m_cigar = *m_cigar_; *n_cigar_ = 0; }
	one_pair(fn, 0, 1, km, qlen, query, tlen, target, m, mat, q, e, 0, 0, w, -1, 0, with_cigar ? 0 : KSW_EZ_SCORE_ONLY, &ez);
	if (with_cigar) {
		*cigar_ = ez.cigar; *m_cigar_ = ez.m_cigar;
		*n_cigar_ = ez.zdropped ? 0 : ez.n_cigar;
	}
	return ez.zdropped ? KSW_NEG_INF : ez.score;
}

int ksw_gg(void *km, int qlen, const uint8_t *query, int tlen, const uint8_t *target, int8_t m, const int8_t *mat,
           int8_t q, int8_t e, int w, int *m_cigar_, int *n_cigar_, uint32_t **cigar_)
{ return global_align("ksw_gg", km, qlen, query, tlen, target, m, mat, q, e, w, m_cigar_, n_cigar_, cigar_); }
int ksw_gg2(void *km, int qlen, const uint8_t *query, int tlen, const uint8_t *target, int8_t m, const int8_t *mat,
            int8_t q, int8_t e, int w, int *m_cigar_, int *n_cigar_, uint32_t **cigar_)
{ return global_align("ksw_gg2", km, qlen, query, tlen, target, m, mat, q, e, w, m_cigar_, n_cigar_, cigar_); }
int ksw_gg2_sse(void *km, int qlen, const uint8_t *query, int tlen, const uint8_t *target, int8_t m, const int8_t *mat,
                int8_t q, int8_t e, int w, int *m_cigar_, int *n_cigar_, uint32_t **cigar_)
{ return global_align("ksw_gg2_sse", km, qlen, query, tlen, target, m, mat, q, e, w, m_cigar_, n_cigar_, cigar_); }

/* ---------------------------------------------------------------- coalescing of concurrent single-pair calls
 * A minimap2-style caller runs a pool of host threads that each call ksw_extz2_sse / ksw_extd2_sse for one pair at a time.
 * One pair is far too little work for a launch: the call costs ~0.45 ms of fixed latency, and the runtime serialises the
 * threads' API calls.  So at most KSW2AMD_COALESCE_SLOTS (default 4, 0 = off) device batches of single-pair calls are in
 * flight at a time.  A call that finds a free slot and no crowd runs at once, alone, exactly as before.  Otherwise it pushes
 * its request on a lock-free list; whoever pushed onto the EMPTY list is the leader of that list: it waits for a slot -- while
 * every other arriving call joins the list -- takes the list, and runs everybody's pairs as ONE batch per (function, scoring)
 * group, each result into the caller's own ksw_extz_t with CIGAR memory from the caller's own km.  Few threads: no added
 * latency.  Many threads: one batch per round trip of the pool.  Results are identical either way.
 *
 * Round 4 (64 threads, 512 x 512: 33 k -> see INTEGRATION.md section 1): what bounded the rate was not the device but the
 * hand-overs.  (1) A pool whose calls come back together calls again together: the first one back found the slots free,
 * ran ALONE, and the other 63 waited for its 0.5 ms launch before their batch could start -- a one-pair plan in front of every
 * batch.  Calls are counted per millisecond; eight or more in this or the last one is a crowd, and in a crowd nobody runs
 * alone: the leader collects for at most KSW2AMD_COALESCE_WINDOW_US (default 200), or until as many calls have arrived as the
 * last batches held (the usual end: a few microseconds).  (2) One mutex and one condition variable woke 63 followers through
 * 63 serial hand-overs of that mutex, and the same threads queued up on it again to enter their next call.  The list is a
 * compare-and-swap push, a follower sleeps on one process-wide futex word that a finished batch bumps once (one system call
 * wakes everybody; a follower of another batch looks at its own flag and sleeps again), and nothing is locked anywhere. */
#include <linux/futex.h>
#include <sys/syscall.h>
#include <limits.h>
#include <sched.h>
#define COAL_MAXQ 512
#define COAL_IDS 64                     /* lists alive at a time: one per slot and the one being collected; a request record is aligned to this */
typedef struct creq_s {
	struct creq_s *next;
	int dual, rc, taken;
	volatile int done;
	const ksw2amd_scoring_t *sc;
	const ksw2amd_pair_t *pr;
	void *km;
	ksw_extz_t *ez;
	char err[200];
} creq_t;
static struct {
	uintptr_t head;                        /* lock-free LIFO of waiting requests: the top record's address | the list's number (COAL_IDS - 1 low bits' worth);
	                                        * pushing onto 0 makes the pusher the leader of a new list */
	int count;                             /* requests on the list (approximate while a leader takes it) */
	int busy;                              /* bit k: slot k has a batch on the device; also the futex word a leader without a slot sleeps on */
	int gen[COAL_IDS];                     /* gen[list number]: bumped when that list's batch is done -- the futex word its followers sleep on */
	int next_id;
	int expect;                            /* the size the last batches had */
	int win_n, win_prev;                   /* calls in the current / the previous millisecond */
	int64_t win_t0;
} g_coal;

#if defined(__x86_64__) || defined(__i386__)
#define cpu_relax() __builtin_ia32_pause()
#else
#define cpu_relax() ((void)0)
#endif
static long futex_call(int *addr, int op, int val, const struct timespec *to) { return syscall(SYS_futex, addr, op, val, to, 0, 0); }

static int same_scoring(const creq_t *a, const creq_t *b)
{
	const ksw2amd_scoring_t *x = a->sc, *y = b->sc;
	if (a->dual != b->dual || x->m != y->m || x->q != y->q || x->e != y->e || (a->dual && (x->q2 != y->q2 || x->e2 != y->e2))) return 0;
	if (x->mat == y->mat) return 1;
	if (!x->mat || !y->mat || x->m <= 0) return 0;
	return memcmp(x->mat, y->mat, (size_t)x->m * x->m) == 0;
}

static void coal_process(creq_t *list, void *stream)
{
	creq_t *r, *g;
	for (g = list; g; g = g->next) {
		ksw2amd_pair_t pairs[COAL_MAXQ];
		ksw_extz_t *ezp[COAL_MAXQ];
		void *kmp[COAL_MAXQ];
		creq_t *mem[COAL_MAXQ];
		ksw2amd_plan_t *p;
		size_t bytes = 0;
		int n = 0, rc = KSW2AMD_OK, i;
		if (g->taken) continue;
		for (r = g; r && n < COAL_MAXQ; r = r->next) {
			size_t b;
			if (r->taken || !same_scoring(g, r)) continue;
			b = pair_device_bytes(g->dual, r->pr);
			if (n > 0 && bytes + b > ((size_t)8 << 30)) continue;       /* stays for a later group */
			bytes += b;
			r->taken = 1; mem[n] = r; pairs[n] = *r->pr; ezp[n] = r->ez; kmp[n] = r->km; ++n;
		}
		__sync_fetch_and_add(&g_stat[2], n); __sync_fetch_and_add(&g_stat[3], 1);
		g_latency_plan = 1; g_plan_stream = stream;
		p = plan_create_ex(g->dual, 0, g->sc, n, pairs, 0, 0);
		g_latency_plan = 0; g_plan_stream = 0;
		if (!p) rc = strstr(g_err, "alloc") ? KSW2AMD_E_NOMEM : strstr(g_err, "device") ? KSW2AMD_E_NODEVICE : KSW2AMD_E_PARAM;
		else {
			rc = ksw2amd_plan_run(p, stream ? stream : thread_stream());
			if (rc == KSW2AMD_OK) rc = plan_fetch_ex(p, 0, 0, ezp, kmp);
			ksw2amd_plan_destroy(p);
		}
		for (i = 0; i < n; ++i) { mem[i]->rc = rc; if (rc) snprintf(mem[i]->err, sizeof(mem[i]->err), "%.190s", g_err); }
	}
}

/* Slots are numbered, and each one owns a stream per device.  The device has FOUR hardware queues and the runtime deals its streams
 * onto them in turn: 64 caller threads with a stream each share them 16 to a queue, and two batches whose leaders' streams meet on
 * one queue run one after the other -- a one-pair plan's device time read 0.33 ms alone and 0.66 ms next to another thread's batch
 * (tools/probe/concurrent_small_kernels_probe.hip: four streams overlap perfectly, 0.32 ms per 0.30 ms kernel; eight take 0.59).
 * A batch therefore runs on its slot's stream, uploads included (in order: no event, no second queue), whoever its leader is. */
#define COAL_MAXSLOTS 8
static void *g_coal_stream[COAL_MAXSLOTS][SHARED_UP_MAXDEV];
static int coal_try_slot(int slots)
{
	int b = __atomic_load_n(&g_coal.busy, __ATOMIC_RELAXED), k;
	for (;;) {
		for (k = 0; k < slots && (b >> k & 1); ++k) {}
		if (k >= slots) return -1;
		if (__atomic_compare_exchange_n(&g_coal.busy, &b, b | 1 << k, 0, __ATOMIC_ACQUIRE, __ATOMIC_RELAXED)) return k;
	}
}
static void coal_free_slot(int k)
{
	__atomic_fetch_and(&g_coal.busy, ~(1 << k), __ATOMIC_RELEASE);
	futex_call(&g_coal.busy, FUTEX_WAKE_PRIVATE, 1, 0);                 /* a leader without a slot */
}
static void *coal_slot_stream(int k)                                      /* (only the slot's holder gets here) */
{
	const int dev = k2a_shim_get_device();
	if (dev < 0 || dev >= SHARED_UP_MAXDEV) return 0;
	if (!g_coal_stream[k][dev]) g_coal_stream[k][dev] = ENV(COALESCE_PLAIN_STREAMS) ? k2a_shim_stream_create() : k2a_shim_stream_create_high();
	return g_coal_stream[k][dev];
}

/* 1 = handled (result or failure delivered), 0 = coalescing is off: the caller runs the pair itself */
static int queue_one(const char *fn, int dual, void *km, const ksw2amd_scoring_t *sc, const ksw2amd_pair_t *pr, ksw_extz_t *ez)
{
	creq_t me __attribute__((aligned(COAL_IDS)));
	const char *se = ENV(COALESCE_SLOTS), *we = ENV(COALESCE_WINDOW_US);
	const int slots = imin(se ? imax(atoi(se), 0) : 4, COAL_MAXSLOTS);
	const int64_t window_ns = (we ? imax(atoi(we), 0) : 200) * (int64_t)1000;
	int64_t t_in, t0;
	int crowd, wn, slot, id = 0;
	uintptr_t old;
	if (slots == 0 || g_is_worker) return 0;
	me.next = 0; me.dual = dual; me.rc = 0; me.taken = 0; me.done = 0; me.sc = sc; me.pr = pr; me.km = km; me.ez = ez; me.err[0] = 0;
	/* how many callers are there?  (counters without a lock: a lost update changes nothing that matters) */
	t_in = now_ns();
	t0 = __atomic_load_n(&g_coal.win_t0, __ATOMIC_RELAXED);
	if (t_in - t0 > 1000000) {
		__atomic_store_n(&g_coal.win_prev, t_in - t0 > 2000000 ? 0 : __atomic_load_n(&g_coal.win_n, __ATOMIC_RELAXED), __ATOMIC_RELAXED);
		__atomic_store_n(&g_coal.win_n, 0, __ATOMIC_RELAXED);
		__atomic_store_n(&g_coal.win_t0, t_in, __ATOMIC_RELAXED);
	}
	wn = __atomic_add_fetch(&g_coal.win_n, 1, __ATOMIC_RELAXED);
	crowd = window_ns > 0 && (wn >= 8 || __atomic_load_n(&g_coal.win_prev, __ATOMIC_RELAXED) >= 8);
	if (!crowd && __atomic_load_n(&g_coal.head, __ATOMIC_RELAXED) == 0 && (slot = coal_try_slot(slots)) >= 0) {      /* a free slot, nobody waiting, no crowd: run alone, now */
		g_latency_plan = 1; g_plan_stream = coal_slot_stream(slot);
		me.rc = run_serial(dual, 0, km, sc, 1, pr, ez, 1, 0, 0);
		g_latency_plan = 0; g_plan_stream = 0;
		if (me.rc) snprintf(me.err, sizeof(me.err), "%.190s", g_err);
		coal_free_slot(slot);
	} else {
		/* the list's number travels in the low bits of the head word, so a pusher learns it with its push: followers of list i sleep on
		 * gen[i], and a finished batch wakes its own followers only (one word for everybody: four pools woke each other four times per
		 * round trip, and every leader paid for waking all of them) */
		const int fresh = __atomic_fetch_add(&g_coal.next_id, 1, __ATOMIC_RELAXED) & (COAL_IDS - 1);
		old = __atomic_load_n(&g_coal.head, __ATOMIC_RELAXED);
		do {
			me.next = (creq_t*)(old & ~(uintptr_t)(COAL_IDS - 1));
			id = old ? (int)(old & (COAL_IDS - 1)) : fresh;
		} while (!__atomic_compare_exchange_n(&g_coal.head, &old, (uintptr_t)&me | (uintptr_t)id, 0, __ATOMIC_RELEASE, __ATOMIC_RELAXED));
		__atomic_fetch_add(&g_coal.count, 1, __ATOMIC_RELAXED);
		if (old == 0) {                                                       /* onto the empty list: the leader of whatever it holds when taken */
			creq_t *list, *r, *nx;
			const int expect = __atomic_load_n(&g_coal.expect, __ATOMIC_RELAXED);
			int n = 0;
			const int tl = trace_level();
			int64_t t_slot, t_col, t_done, t_woken;
			while ((slot = coal_try_slot(slots)) < 0) {                         /* others keep joining the list meanwhile */
				const int b = __atomic_load_n(&g_coal.busy, __ATOMIC_RELAXED);
				if (b == (1 << slots) - 1) futex_call(&g_coal.busy, FUTEX_WAIT_PRIVATE, b, 0);
			}
			t_slot = tl ? now_ns() : 0;
			if (crowd) {                                                        /* the collection window (one thread polls; nobody is woken for it) */
				const int64_t t_dl = t_in + window_ns;
				int spins = 0;
				while (__atomic_load_n(&g_coal.count, __ATOMIC_RELAXED) < imin(expect, COAL_MAXQ) && now_ns() < t_dl)
					if (++spins & 15) cpu_relax(); else sched_yield();
			}
			t_col = tl ? now_ns() : 0;
			list = (creq_t*)(__atomic_exchange_n(&g_coal.head, (uintptr_t)0, __ATOMIC_ACQUIRE) & ~(uintptr_t)(COAL_IDS - 1));       /* (the next push starts a new list with a leader of its own) */
			for (r = list; r; r = r->next) ++n;
			__atomic_fetch_sub(&g_coal.count, n, __ATOMIC_RELAXED);
			__atomic_store_n(&g_coal.expect, n >= expect ? n : expect - (expect - n + 1) / 2, __ATOMIC_RELAXED);      /* (follows a shrinking pool in a few batches) */
			coal_process(list, coal_slot_stream(slot));
			t_done = tl ? now_ns() : 0;
			coal_free_slot(slot);
			for (r = list; r; r = nx) { nx = r->next; if (r != &me) __atomic_store_n(&r->done, 1, __ATOMIC_RELEASE); }      /* (a follower's record dies as soon as it sees this) */
			__atomic_fetch_add(&g_coal.gen[id], 1, __ATOMIC_RELEASE);
			if (n > 1) futex_call(&g_coal.gen[id], FUTEX_WAKE_PRIVATE, INT_MAX, 0);
			if (tl) {
				t_woken = now_ns();
				fprintf(stderr, "[ksw2_amd] coalesced batch of %d (expected %d): waited %.3f ms for a slot, collected for %.3f, plan + run + fetch %.3f, wake-up calls %.3f ms\n",
				        n, expect, (t_slot - t_in) * 1e-6, (t_col - t_slot) * 1e-6, (t_done - t_col) * 1e-6, (t_woken - t_done) * 1e-6);
			}
		} else {
			int *word = &g_coal.gen[id];
			for (;;) {
				const int g = __atomic_load_n(word, __ATOMIC_ACQUIRE);
				if (__atomic_load_n(&me.done, __ATOMIC_ACQUIRE)) break;
				futex_call(word, FUTEX_WAIT_PRIVATE, g, 0);
			}
		}
	}
	if (me.rc) { snprintf(g_err, sizeof(g_err), "%s", me.err); call_failed(fn, me.rc, ez); }
	return 1;
}

/* ---------------------------------------------------------------- splice-aware extension (ksw_exts2_sse) */

static int exts_long_thres(int q, int e, int q2)           /* ksw2_exts2_sse.c:102-104 */
{
	int lt = (q2 - q) / e - 1;
	if (q2 > q + e + lt * e) ++lt;
	return lt;
}

ksw2amd_plan_t *ksw2amd_exts_plan_create(const ksw2amd_splice_t *sc, int n, const ksw2amd_spair_t *pairs)
{
	ksw2amd_plan_t *p;
	const int m = sc ? sc->m : 0;
	int i, k, g, lo;
	size_t off, mat_off, up_bytes = 0, cst_words = 0;
	void *up;
	uint32_t fill[3][2][3];
	int wn;

	g_err[0] = 0;
	if (n < 0 || (n > 0 && !pairs) || !sc) { fail(KSW2AMD_E_PARAM, "exts: bad arguments%s", 0); return 0; }
	p = plan_new("exts", n, 1);
	if (!p) return 0;
	p->splice = 1; p->m = m;
	for (i = 0; i < n; ++i) { p->h_cls[i] = -1; p->h_flag[i] = pairs[i].flag & ~F_SCALAR_CONTRACT; }
	/* ksw2_exts2_sse.c:74,91: unusable model or a mismatch no gap pair could undercut -> results stay reset */
	if (m <= 1 || !sc->mat || sc->q2 <= sc->q + sc->e) p->reject_all = 1;
	else if (m > K2A_MAXM || sc->e <= 0) { fail(KSW2AMD_E_PARAM, "exts: m > 127 or gap extension <= 0%s", 0); goto err; }
	else {
		for (k = 1, lo = sc->mat[1]; k < m * m; ++k) lo = imin(lo, sc->mat[k]);
		if (-lo > 2 * (sc->q + sc->e)) p->reject_all = 1;
	}
	if (p->reject_all || n == 0) return p;

	/* arena: query bytes (4-aligned) + one dword of constants per target position; the two effective matrices at the end */
	memset(p->s_count, 0, sizeof(p->s_count));
	off = 0;
	for (i = 0; i < n; ++i) {
		const ksw2amd_spair_t *a = &pairs[i];
		K2aPair *d = &p->h_pairs[i];
		const int fl = p->h_flag[i];
		int mode, generic;
		if (a->qlen <= 0 || a->tlen <= 0) continue;
		if (!a->query || !a->target) { fail(KSW2AMD_E_PARAM, "exts: NULL sequence%s", 0); goto err; }
		mode = (fl & KSW_EZ_SCORE_ONLY) ? K2A_MODE_SCORE : (fl & KSW_EZ_RIGHT) ? K2A_MODE_RIGHT : K2A_MODE_LEFT;
		if (is_approx(fl) && (fl & KSW_EZ_EXTZ_ONLY)) mode = K2A_MODE_SCORE;      /* no start cell in that mode: no CIGAR */
		generic = (fl & KSW_EZ_GENERIC_SC) ? 1 : 0;
		/* register windows wherever the diagonal fits one: faster than the HBM-state kernel in every mode
		 * (tools/scripts/exts_classes.py) */
		wn = imin(a->qlen, a->tlen) <= K2A_DM_DIAG(K2A_DM_SLOTS_S) ? 0 : imin(a->qlen, a->tlen) <= K2A_DM_DIAG(K2A_DM_SLOTS) ? 1 : 2;
		if (ENV(EXTS_BIG)) wn = 2;        /* tests: every pair through the HBM-state kernel */
		else if (ENV(EXTS_REG) && imin(a->qlen, a->tlen) <= K2A_DM_DIAG(K2A_DM_SLOTS)) wn = imin(wn, 1);   /* tests: 16 slots with traceback */
		if (wn == 2) {                                 /* 9 ints of state per target position, 16-byte granules */
			d->pad = (uint32_t)(p->bnd_words / 4);
			p->bnd_words += align_up(9 * (size_t)a->tlen, 4);
			if (p->bnd_words / 4 > 0xfff00000u) { fail(KSW2AMD_E_PARAM, "exts: state scratch over 64 GiB in one plan%s", 0); goto err; }
		}
		p->h_cls[i] = (int8_t)((mode * 2 + generic) * 3 + wn);
		++p->s_count[mode][generic][wn];
		d->qlen = a->qlen; d->tlen = d->tlen_full = a->tlen;
		d->w = imax(a->qlen, a->tlen);                 /* no band: k2a_finish must never see an unreachable corner */
		d->zdrop = a->zdrop; d->end_bonus = K2A_NEG;   /* no end bonus in this function */
		d->flag = fl & (KSW_EZ_EXTZ_ONLY | KSW_EZ_REV_CIGAR | KSW_EZ_SCORE_ONLY | KSW_EZ_SPLICE_FOR | KSW_EZ_SPLICE_REV | KSW_EZ_SPLICE_FLANK);   /* the splice bits: k2a_splice_const */
		if (a->junc) d->flag |= K2A_F_HAS_JUNC;
		if (is_approx(fl)) {
			d->zdrop = -1;
			if (fl & KSW_EZ_EXTZ_ONLY) d->flag |= KSW_EZ_SCORE_ONLY;
		}
		/* uploaded: query, target, annotation bytes; the per-position dwords (bnd_off) are built on the device behind them */
		off = align_up(off, 4); d->qoff = (uint32_t)off; off += (size_t)a->qlen;
		off = align_up(off, 4); d->toff = (uint32_t)off; off += align_up((size_t)a->tlen, 4) + (a->junc ? align_up((size_t)a->tlen, 4) : 0) + 4;
		cst_words += (size_t)a->tlen;
		if (off + 4 * cst_words > 0xfff00000u) { fail(KSW2AMD_E_PARAM, "exts: more than 4 GiB of sequence in one plan%s", 0); goto err; }
		p->cells += (int64_t)a->qlen * a->tlen;
		if (mode != K2A_MODE_SCORE) {
			d->tb_off = p->tb_bytes;
			p->tb_bytes += align_up((size_t)(a->qlen + a->tlen - 1) * (size_t)imin(a->qlen, a->tlen), 256);
			d->cig_off = (uint32_t)p->cig_words;
			p->cig_words += (size_t)a->qlen + a->tlen + 2;
			if (p->cig_words > 0xfff00000u) { fail(KSW2AMD_E_PARAM, "exts: CIGAR scratch over 16 GiB in one plan%s", 0); goto err; }
		}
	}
	off = align_up(off + 256, 256);
	mat_off = off; off = align_up(off + 2 * (size_t)m * m, 256);
	up_bytes = off;                                   /* what goes over the link; the constants follow on the device only */
	for (i = 0; i < n; ++i)
		if (p->h_cls[i] >= 0) { p->h_pairs[i].bnd_off = (uint32_t)(off / 4); off += 4 * (size_t)p->h_pairs[i].tlen; }
	p->seq_bytes = align_up(off + 256, 256);
	p->h_seq = (uint8_t*)cache_get(BUF_HSEQ, up_bytes, &p->cap[BUF_HSEQ]);
	if (!p->h_seq) { fail(KSW2AMD_E_NOMEM, "pinned staging allocation failed: %s", k2a_shim_last_error()); goto err; }
	for (k = 0, i = 0; i < 3; ++i)
		for (g = 0; g < 2; ++g)
			for (wn = 0; wn < 3; ++wn) { p->s_first[i][g][wn] = k; fill[i][g][wn] = (uint32_t)k; k += p->s_count[i][g][wn]; }
	p->ntasks = p->norder = k;
	for (i = 0; i < n; ++i) {
		const ksw2amd_spair_t *a = &pairs[i];
		if (p->h_cls[i] < 0) continue;
		memcpy(p->h_seq + p->h_pairs[i].qoff, a->query, (size_t)a->qlen);
		memcpy(p->h_seq + p->h_pairs[i].toff, a->target, (size_t)a->tlen);
		if (a->junc) memcpy(p->h_seq + p->h_pairs[i].toff + align_up((size_t)a->tlen, 4), a->junc, (size_t)a->tlen);
		p->h_order[fill[p->h_cls[i] / 6][(p->h_cls[i] / 3) & 1][p->h_cls[i] % 3]++] = (uint32_t)i;
	}
	build_eff(0, m, sc->mat, sc->e, 0, 0, (int8_t*)p->h_seq + mat_off);
	build_eff(0, m, sc->mat, sc->e, 0, 1, (int8_t*)p->h_seq + mat_off + (size_t)m * m);

	p->d_seq = (uint8_t*)cache_get(BUF_SEQ, p->seq_bytes, &p->cap[BUF_SEQ]);
	p->d_pairs = (K2aPair*)cache_get(BUF_PAIRS, sizeof(K2aPair) * ((size_t)n + 1), &p->cap[BUF_PAIRS]);
	p->d_res = (K2aResult*)cache_get(BUF_RES, sizeof(K2aResult) * ((size_t)n + 1), &p->cap[BUF_RES]);
	p->d_order = (uint32_t*)cache_get(BUF_ORDER, sizeof(uint32_t) * ((size_t)p->norder + 1), &p->cap[BUF_ORDER]);
	p->d_tb = p->tb_bytes ? (uint8_t*)cache_get(BUF_TB, p->tb_bytes, &p->cap[BUF_TB]) : 0;
	p->d_cig = p->cig_words ? (uint32_t*)cache_get(BUF_CIG, p->cig_words * 4, &p->cap[BUF_CIG]) : 0;
	p->d_bnd = p->bnd_words ? (int32_t*)cache_get(BUF_BND, p->bnd_words * 4, &p->cap[BUF_BND]) : 0;
	if (!p->d_seq || !p->d_pairs || !p->d_res || !p->d_order || (p->tb_bytes && !p->d_tb) || (p->cig_words && !p->d_cig) ||
	    (p->bnd_words && !p->d_bnd)) {
		fail(KSW2AMD_E_NOMEM, "device allocation failed: %s", k2a_shim_last_error());
		goto err;
	}
	up = thread_upload_stream();
	p->stream = up; p->stream_used = 1;
	if (k2a_shim_h2d(p->d_seq, p->h_seq, up_bytes, up) || k2a_shim_h2d(p->d_pairs, p->h_pairs, sizeof(K2aPair) * (size_t)n, up) ||
	    k2a_shim_h2d(p->d_order, p->h_order, sizeof(uint32_t) * (size_t)p->norder, up) ||
	    k2a_shim_memset(p->d_res, 0, sizeof(K2aResult) * (size_t)n, up) ||
	    k2a_shim_launch_splice_const(p->d_pairs, n, p->d_seq, sc->noncan, sc->junc_bonus, up) || k2a_shim_stream_sync(up)) {
		fail(KSW2AMD_E_NODEVICE, "upload failed: %s", k2a_shim_last_error());
		goto err;
	}
	for (g = 0; g < 2; ++g) {
		p->s_par[g].q = sc->q; p->s_par[g].e = sc->e; p->s_par[g].q2 = sc->q2; p->s_par[g].m = m;
		p->s_par[g].long_thres = exts_long_thres(sc->q, sc->e, sc->q2);
		p->s_par[g].mat = (const int8_t*)p->d_seq + mat_off + (g ? (size_t)m * m : 0);
	}
	plan_ready(p);                                  /* uploads complete */
	return p;
err:
	ksw2amd_plan_destroy(p);
	return 0;
}

static int exts_plan_run(ksw2amd_plan_t *p, void *stream)
{
	int mode, g, wn;
	p->stream = stream; p->ran = 1; p->stream_used = 1;
	if (p->up_ev && k2a_shim_stream_wait_event(stream, p->up_ev)) goto err;      /* the plan's upload (shared stream) before its kernels */
	if (p->reject_all || p->ntasks == 0) return KSW2AMD_OK;
	if (k2a_shim_event_record(p->ev[0], stream)) goto err;
	for (mode = 0; mode < 3; ++mode)
		for (g = 0; g < 2; ++g)
			for (wn = 0; wn < 3; ++wn)
				if (p->s_count[mode][g][wn] &&
				    k2a_shim_launch_exts(mode, wn, &p->s_par[g], p->d_pairs, p->d_order + p->s_first[mode][g][wn], p->s_count[mode][g][wn], p->d_seq,
				                         p->d_tb, p->d_bnd, p->d_res, stream)) goto err;
	if (k2a_shim_event_record(p->ev[1], stream)) goto err;
	for (mode = 1; mode < 3; ++mode)
		for (g = 0; g < 2; ++g)
			for (wn = 0; wn < 3; ++wn)
				if (p->s_count[mode][g][wn] &&
				    k2a_shim_launch_exts_trace(&p->s_par[g], p->d_pairs, p->d_order + p->s_first[mode][g][wn], p->s_count[mode][g][wn], p->d_tb,
				                               p->d_res, p->d_cig, stream)) goto err;
	if (k2a_shim_event_record(p->ev[2], stream)) goto err;
	return KSW2AMD_OK;
err:
	return fail(KSW2AMD_E_NODEVICE, "exts run: %s", k2a_shim_last_error());
}

static int exts_serial(void *km, const ksw2amd_splice_t *sc, int n, const ksw2amd_spair_t *pairs, ksw_extz_t *ez, int share)
{
	int beg = 0;
	size_t budget, free_b = 0, total_b = 0;
	const char *env = ENV(MAX_BYTES);
	if (n <= 0) return KSW2AMD_OK;
	if (k2a_shim_device_count() <= 0) return fail(KSW2AMD_E_NODEVICE, "no usable %s device", k2a_shim_backend());
	if (env && atoll(env) > 0) budget = (size_t)atoll(env);
	else if (n == 1) budget = (size_t)1 << 34;
	else {
		if (k2a_shim_mem_info(&free_b, &total_b)) return fail(KSW2AMD_E_NODEVICE, "mem_info: %s", k2a_shim_last_error());
		budget = device_budget(free_b, total_b, share);
	}
	while (beg < n) {
		ksw2amd_plan_t *p;
		size_t acc = 0, seq = 0;
		int end, rc;
		for (end = beg; end < n; ++end) {
			const size_t ql = (size_t)imax(pairs[end].qlen, 0), tl = (size_t)imax(pairs[end].tlen, 0);
			const size_t b = ql + 6 * tl + 256 + ((pairs[end].flag & KSW_EZ_SCORE_ONLY) ? 0 : (ql + tl) * (ql < tl ? ql : tl) + 4 * (ql + tl) + 512);      /* query, target, annotation, 4 bytes of constants per position */
			if (end > beg && (acc + b > budget || seq + ql + 6 * tl > 3000000000u || end - beg >= (1 << 22))) break;
			acc += b; seq += ql + 6 * tl + 32;
		}
		{
			const double t0 = now_ms();
			double t1, t2;
			p = ksw2amd_exts_plan_create(sc, end - beg, pairs + beg);
			if (!p) return strstr(g_err, "alloc") ? KSW2AMD_E_NOMEM : strstr(g_err, "device") ? KSW2AMD_E_NODEVICE : KSW2AMD_E_PARAM;
			t1 = now_ms();
			rc = ksw2amd_plan_run(p, thread_stream());
			t2 = now_ms();
			if (rc == KSW2AMD_OK) rc = ksw2amd_plan_fetch(p, km, ez + beg);
			if (trace_on()) fprintf(stderr, "[ksw2_amd] exts plan @%d n=%d: pack+upload %.2f ms, launch %.2f ms, wait+fetch %.2f ms\n", beg, end - beg, t1 - t0, t2 - t1, now_ms() - t2);
		}
		ksw2amd_plan_destroy(p);
		if (rc) return rc;
		beg = end;
	}
	return KSW2AMD_OK;
}

/* the splice-aware and the X-drop batches through the same worker pool as the extz / extd batches (run_pooled): one chunk per
 * worker, each packed, run and fetched on the worker's own streams -- the packing (per-position splice constants, interleaved
 * lane blocks) is what bounds these functions end to end */
/* chunk count and size for these one-alignment-per-wavefront classes: one chunk per worker; a batch of one shape is cut at
 * multiples of a device fill (one wavefront per SIMD) like the extz / extd batches (uniform_chunks) */
static int wave_chunks(int n, int workers, int uniform, int *chunk_pairs)
{
	const int simds = k2a_shim_simd_count();
	int k = imin(workers, n / 256);
	*chunk_pairs = 0;
	if (uniform && simds > 0 && k >= 2 && n >= 2 * simds) {
		int cp = (n + k - 1) / k;
		cp = (cp + simds - 1) / simds * simds;
		*chunk_pairs = cp;
		k = (n + cp - 1) / cp;
	}
	return k;
}

typedef struct { void *km; const ksw2amd_splice_t *sc; const ksw2amd_spair_t *pairs; ksw_extz_t *ez; } exts_ctx_t;
static int exts_chunk(void *ctx_, int beg, int end, int share, pend_t *pd)
{
	exts_ctx_t *c = (exts_ctx_t*)ctx_;
	(void)pd;
	if (beg < 0) return KSW2AMD_OK;
	return exts_serial(c->km, c->sc, end - beg, c->pairs + beg, c->ez + beg, share);
}

int ksw2amd_exts_batch(void *km, const ksw2amd_splice_t *sc, int n, const ksw2amd_spair_t *pairs, ksw_extz_t *ez)
{
	const int tpd = pool_threads_per_device();
	if (n >= (pool_min_pairs() ? pool_min_pairs() : 2048) && tpd > 0 && !g_is_worker && k2a_shim_device_count() > 0) {
		const int workers = tpd * (g_ndev_set > 0 ? g_ndev_set : 1);
		double *cost = (double*)malloc(sizeof(double) * (size_t)n), total = 0;
		int i, rc = 0, uniform = 1, chunk_pairs = 0, nchunks;
		for (i = 1; i < n && uniform; ++i) uniform = pairs[i].qlen == pairs[0].qlen && pairs[i].tlen == pairs[0].tlen;
		nchunks = wave_chunks(n, workers, uniform, &chunk_pairs);
		if (cost && nchunks >= 2) {
			exts_ctx_t ctx;
			for (i = 0; i < n; ++i) { cost[i] = 1.0 + (double)imax(pairs[i].qlen, 0) * imax(pairs[i].tlen, 0); total += cost[i]; }
			ctx.km = km; ctx.sc = sc; ctx.pairs = pairs; ctx.ez = ez;
			if (run_pooled(exts_chunk, &ctx, n, cost, total, nchunks, chunk_pairs, &rc)) { free(cost); return rc; }
		}
		free(cost);
	}
	return exts_serial(km, sc, n, pairs, ez, 1);
}

void ksw_exts2_sse(void *km, int qlen, const uint8_t *query, int tlen, const uint8_t *target, int8_t m, const int8_t *mat,
                   int8_t q, int8_t e, int8_t q2, int8_t noncan, int zdrop, int8_t junc_bonus, int flag, const uint8_t *junc, ksw_extz_t *ez)
{
	ksw2amd_splice_t sc;
	ksw2amd_spair_t pr;
	sc.m = m; sc.mat = mat; sc.q = q; sc.e = e; sc.q2 = q2; sc.noncan = noncan; sc.junc_bonus = junc_bonus;
	pr.query = query; pr.target = target; pr.junc = junc; pr.qlen = qlen; pr.tlen = tlen; pr.zdrop = zdrop; pr.flag = flag;
	{
		const int rc = ksw2amd_exts_batch(km, &sc, 1, &pr, ez);
		if (rc != KSW2AMD_OK) call_failed("ksw_exts2_sse", rc, ez);
	}
}
void ksw_exts2_sse41(void *km, int qlen, const uint8_t *query, int tlen, const uint8_t *target, int8_t m, const int8_t *mat,
                     int8_t q, int8_t e, int8_t q2, int8_t noncan, int zdrop, int8_t junc_bonus, int flag, const uint8_t *junc, ksw_extz_t *ez)
{ ksw_exts2_sse(km, qlen, query, tlen, target, m, mat, q, e, q2, noncan, zdrop, junc_bonus, flag, junc, ez); }
void ksw_exts2_sse2(void *km, int qlen, const uint8_t *query, int tlen, const uint8_t *target, int8_t m, const int8_t *mat,
                    int8_t q, int8_t e, int8_t q2, int8_t noncan, int zdrop, int8_t junc_bonus, int flag, const uint8_t *junc, ksw_extz_t *ez)
{ ksw_exts2_sse(km, qlen, query, tlen, target, m, mat, q, e, q2, noncan, zdrop, junc_bonus, flag, junc, ez); }

/* ---------------------------------------------------------------- gap-linear X-drop extension (ksw_extf2_sse) */

#define EXTF_LDS_T0 1024
#define EXTF_LDS_T1 4096
#define EXTF_LDS_T2 21504          /* 3 x 21504 bytes = 63 KiB of LDS for one wavefront */

ksw2amd_plan_t *ksw2amd_extf_plan_create(int8_t mch, int8_t mis, int8_t e, int n, const ksw2amd_fpair_t *pairs)
{
	ksw2amd_plan_t *p;
	int i, c, span, nlane = 0, use_lane;
	size_t off = 0;
	uint32_t fill[7];
	void *up;
	sort_t *srt = 0;

	g_err[0] = 0;
	if (n < 0 || (n > 0 && !pairs)) { fail(KSW2AMD_E_PARAM, "extf: bad arguments%s", 0); return 0; }
	p = plan_new("extf", n, 1);
	if (!p) return 0;
	p->splice = 2;
	memset(p->h_flag, 0, sizeof(int32_t) * ((size_t)n + 1));
	p->f_par.mch = mch; p->f_par.mis = mis < 0 ? mis : -mis; p->f_par.e = e;      /* ksw2_extf2_sse.c:18-20 */
	/* One extension per lane instead of per wavefront: an order of magnitude fewer instructions per cell, but a wavefront then
	 * holds 64 extensions and every lane walks its band serially: for big batches of narrow bands (KSW2AMD_EXTF_LANE=1 / 0 forces) */
	{
		/* measured (profiles/r2_extf_lane.txt): the lane form reaches ~600 GCUPS from 4 wavefronts per SIMD (262 144 extensions) and
		 * scales down linearly below 131 072; the position-per-lane forms do 180 / 460 / 760 GCUPS at 30 / 100 / 300 positions in
		 * the band whatever the batch size.  Take the lane form where it is ahead by 15 %. */
		const char *ev = ENV(EXTF_LANE);
		double span_sum = 0, lane_rate, wave_rate;
		int nv = 0;
		for (i = 0; i < n; ++i)
			if (pairs[i].qlen > 0 && pairs[i].tlen > 0) {
				const int wq = pairs[i].w < 0 ? imax(pairs[i].qlen, pairs[i].tlen) : pairs[i].w;
				span_sum += imin(imin(pairs[i].qlen, pairs[i].tlen), wq < 0x7ffffff0 ? wq + 1 : wq); ++nv;
			}
		lane_rate = 600.0 * (n >= 131072 ? 1.0 : (double)n / 131072.0);
		wave_rate = nv ? 60.0 + 4.0 * span_sum / nv : 0.0;
		if (wave_rate > 760.0) wave_rate = 760.0;
		use_lane = ev && *ev ? atoi(ev) != 0 : lane_rate > 1.15 * wave_rate;
		if (ENV(EXTF_LDS) || ENV(EXTF_WIN) || ENV(EXTF_HBM)) use_lane = ev && *ev ? atoi(ev) != 0 : 0;
	}
	for (i = 0; i < n; ++i) {
		const ksw2amd_fpair_t *a = &pairs[i];
		K2aPair *d = &p->h_pairs[i];
		p->h_cls[i] = -1;
		d->qlen = a->qlen; d->tlen = d->tlen_full = a->tlen;
		if (a->qlen <= 0 || a->tlen <= 0) continue;
		if (!a->query || !a->target) { fail(KSW2AMD_E_PARAM, "extf: NULL sequence%s", 0); goto err; }
		d->w = a->w < 0 ? imax(a->qlen, a->tlen) : a->w;                             /* ksw2_extf2_sse.c:23 */
		d->zdrop = a->xdrop;
		c = a->tlen <= EXTF_LDS_T0 ? 0 : a->tlen <= EXTF_LDS_T1 ? 1 : a->tlen <= EXTF_LDS_T2 ? 2 : 3;
		/* narrow bands run from registers: at most min(w + 1, qlen, tlen) positions of an anti-diagonal are inside the band */
		span = imin(imin(a->qlen, a->tlen), d->w < 0x7ffffff0 ? d->w + 1 : d->w);
		/* (tools/scripts/extf_classes.py: equal to the LDS form up to ~128 positions on short targets -- both are bound by
		 * the per-anti-diagonal bookkeeping -- and 1.5-1.8 x faster on wider bands and wherever the LDS form needs 12 KiB or more) */
		if (!ENV(EXTF_LDS) && (span > 128 || c > 0)) c = span <= K2A_EXTF_WIN_SPAN(4) ? 4 : span <= K2A_EXTF_WIN_SPAN(8) ? 5 : c;
		if (ENV(EXTF_WIN)) c = span <= K2A_EXTF_WIN_SPAN(4) ? 4 : span <= K2A_EXTF_WIN_SPAN(8) ? 5 : c;   /* tests: the window wherever it fits */
		if (ENV(EXTF_HBM)) c = 3;            /* tests: every pair through the HBM-state kernel */
		p->cells += band_cells(a->qlen, a->tlen, d->w);
		if (use_lane) { p->h_cls[i] = 6; ++p->f_count[6]; ++nlane; continue; }          /* sequences and state: grouped below */
		if (c == 3) { d->tb_off = p->tb_bytes; p->tb_bytes += align_up(3 * align_up((size_t)a->tlen, 16), 256); }
		p->h_cls[i] = (int8_t)c; ++p->f_count[c];
		off = align_up(off, 4); d->qoff = (uint32_t)off; off += (size_t)a->qlen;
		off = align_up(off, 4); d->toff = (uint32_t)off; off += (size_t)a->tlen;
		if (off > 0xfff00000u) { fail(KSW2AMD_E_PARAM, "extf: more than 4 GiB of sequence in one plan%s", 0); goto err; }
	}
	for (c = 0, i = 0; c < 7; ++c) { p->f_first[c] = i; fill[c] = (uint32_t)i; i += p->f_count[c]; }
	if (nlane) {
		/* groups of 64 pairs of similar shape (sorted by target, query, band); per group: target codes and the reversed query
		 * interleaved by lane in the sequence arena, three state arrays of `rows` dwords per lane in the scratch block */
		int k = 0, g;
		srt = (sort_t*)malloc(sizeof(sort_t) * (size_t)nlane);
		if (!srt) { fail(KSW2AMD_E_NOMEM, "extf: host allocation failed%s", 0); goto err; }
		for (k = 0, c = 0; c < n; ++c)
			if (p->h_cls[c] == 6) {
				srt[k].idx = (uint32_t)c; srt[k].tf = (uint32_t)p->h_pairs[c].w;
				srt[k].cost = ((int64_t)p->h_pairs[c].tlen << 32) + p->h_pairs[c].qlen; ++k;
			}
		qsort(srt, (size_t)nlane, sizeof(sort_t), cmp_cost_desc);
		for (g = 0; g < nlane; g += 64) {
			const int cnt = imin(64, nlane - g);
			int tmax = 0, qmax = 0, j;
			size_t trows, qrows, toff_g, qoff_g;
			for (j = 0; j < cnt; ++j) { tmax = imax(tmax, p->h_pairs[srt[g + j].idx].tlen); qmax = imax(qmax, p->h_pairs[srt[g + j].idx].qlen); }
			trows = (align_up((size_t)tmax, 16) + 16) / 4;            /* dwords per lane: the padded target + one block (the followed cell's neighbour) */
			qrows = (align_up((size_t)qmax, 4) + 64) / 4;             /* the reversed query + the zeros the score runs read past it */
			off = align_up(off, 256); toff_g = off; off += trows * 256;
			qoff_g = off; off += qrows * 256;
			if (off > 0xfff00000u) { fail(KSW2AMD_E_PARAM, "extf: more than 4 GiB of sequence in one plan%s", 0); goto err; }
			for (j = 0; j < cnt; ++j) {
				K2aPair *d = &p->h_pairs[srt[g + j].idx];
				d->toff = (uint32_t)toff_g; d->qoff = (uint32_t)qoff_g; d->pad = (uint32_t)trows;
				d->tb_off = p->tb_bytes;
				p->h_order[fill[6]++] = srt[g + j].idx;
			}
			p->tb_bytes += 3 * trows * 256;
		}
		p->f_state_bytes = p->tb_bytes;                       /* (class 3 blocks, if any, were laid out before: none when this class is on) */
		/* the state arrays as LDS rings where every lane's band fits one (k2a_extf_lane_ring_kernel; KSW2AMD_EXTF_RING=0: HBM scratch) */
		{
			int need = 0;
			for (k = 0; k < nlane; ++k) {
				const K2aPair *d = &p->h_pairs[srt[k].idx];
				need = imax(need, K2A_EXTF_RING_ROWS(imin(imin(d->qlen, d->tlen) - 1, d->w)));
			}
			/* 3 x rows x 256 bytes of LDS per wavefront: at most 64 rows (48 KB).  Measured (tools/scripts/extf_lane_probe.py,
			 * profiles/r3_extf_lane_ring.txt): the rings win while the launch has few wavefronts per SIMD (65 536 x 1 000^2, band 100:
			 * 382 against 300 GCUPS; band 30: 312 against 272), the HBM form with its four and more wavefronts per SIMD wins on big
			 * launches (262 144 x 1 000^2: 601 against 343) -- the lane's loop is serial and wants the wavefronts more than the
			 * bandwidth.  Unset: rings up to 1.5 wavefronts per SIMD; 1 = wherever they fit, 0 = never. */
			need = (need + 3) & ~3;
			if (need < 16) need = 16;
			p->f_par.ring = need > 64 ? 0 : ENV(EXTF_RING) ? (atoi(ENV(EXTF_RING)) ? need : 0)
			              : (k2a_shim_simd_count() > 0 && (int64_t)(nlane + 63) / 64 * 2 <= 3 * (int64_t)k2a_shim_simd_count() ? need : 0);
		}
	}
	p->ntasks = p->norder = i;
	if (p->ntasks == 0) return p;
	p->seq_bytes = align_up(off + 256, 256);
	p->h_seq = (uint8_t*)cache_get(BUF_HSEQ, p->seq_bytes, &p->cap[BUF_HSEQ]);
	if (!p->h_seq) { fail(KSW2AMD_E_NOMEM, "pinned staging allocation failed: %s", k2a_shim_last_error()); goto err; }
	if (nlane) memset(p->h_seq, 0, p->seq_bytes);             /* the interleaved blocks read zero past every sequence's end */
	for (i = 0; i < n; ++i) {
		if (p->h_cls[i] < 0 || p->h_cls[i] == 6) continue;
		memcpy(p->h_seq + p->h_pairs[i].qoff, pairs[i].query, (size_t)pairs[i].qlen);
		memcpy(p->h_seq + p->h_pairs[i].toff, pairs[i].target, (size_t)pairs[i].tlen);
		p->h_order[fill[p->h_cls[i]]++] = (uint32_t)i;
	}
	for (i = 0; i < nlane; ++i) {                             /* lane = position in the class's task list, byte x of lane l at (x / 4 * 64 + l) * 4 + x % 4 */
		const uint32_t pi = p->h_order[p->f_first[6] + i];
		const K2aPair *d = &p->h_pairs[pi];
		const int lane = i & 63, ql = d->qlen, tl = d->tlen;
		uint8_t *T = p->h_seq + d->toff + 4 * lane, *Q = p->h_seq + d->qoff + 4 * lane;
		const uint8_t *ts = pairs[pi].target, *qs = pairs[pi].query;
		int x;
		for (x = 0; x < tl; ++x) T[(size_t)(x >> 2) * 256 + (x & 3)] = ts[x];
		for (x = 0; x < ql; ++x) Q[(size_t)(x >> 2) * 256 + (x & 3)] = qs[ql - 1 - x];      /* reversed, like ksw2_extf2_sse.c:31 */
	}
	free(srt); srt = 0;
	p->d_seq = (uint8_t*)cache_get(BUF_SEQ, p->seq_bytes, &p->cap[BUF_SEQ]);
	p->d_pairs = (K2aPair*)cache_get(BUF_PAIRS, sizeof(K2aPair) * ((size_t)n + 1), &p->cap[BUF_PAIRS]);
	p->d_res = (K2aResult*)cache_get(BUF_RES, sizeof(K2aResult) * ((size_t)n + 1), &p->cap[BUF_RES]);
	p->d_order = (uint32_t*)cache_get(BUF_ORDER, sizeof(uint32_t) * ((size_t)p->norder + 1), &p->cap[BUF_ORDER]);
	p->d_tb = p->tb_bytes ? (uint8_t*)cache_get(BUF_TB, p->tb_bytes, &p->cap[BUF_TB]) : 0;
	if (!p->d_seq || !p->d_pairs || !p->d_res || !p->d_order || (p->tb_bytes && !p->d_tb)) {
		fail(KSW2AMD_E_NOMEM, "device allocation failed: %s", k2a_shim_last_error());
		goto err;
	}
	up = thread_upload_stream();
	p->stream = up; p->stream_used = 1;
	if (k2a_shim_h2d(p->d_seq, p->h_seq, p->seq_bytes, up) || k2a_shim_h2d(p->d_pairs, p->h_pairs, sizeof(K2aPair) * (size_t)n, up) ||
	    k2a_shim_h2d(p->d_order, p->h_order, sizeof(uint32_t) * (size_t)p->norder, up) ||
	    k2a_shim_memset(p->d_res, 0, sizeof(K2aResult) * (size_t)n, up) || k2a_shim_stream_sync(up)) {
		fail(KSW2AMD_E_NODEVICE, "upload failed: %s", k2a_shim_last_error());
		goto err;
	}
	plan_ready(p);                                  /* uploads complete */
	return p;
err:
	free(srt);
	ksw2amd_plan_destroy(p);
	return 0;
}

static int extf_plan_run(ksw2amd_plan_t *p, void *stream)
{
	int c;
	p->stream = stream; p->ran = 1; p->stream_used = 1;
	if (p->ntasks == 0) return KSW2AMD_OK;
	if (k2a_shim_event_record(p->ev[0], stream)) goto err;
	if (p->f_count[6] && k2a_shim_memset(p->d_tb, 0, p->f_state_bytes, stream)) goto err;     /* the reference's zeroed arrays (ksw2_extf2_sse.c:25) */
	for (c = 6; c >= 0; --c)
		if (p->f_count[c] && k2a_shim_launch_extf(c, &p->f_par, p->d_pairs, p->d_order + p->f_first[c], p->f_count[c], p->d_seq, p->d_tb, p->d_res, stream))
			goto err;
	if (k2a_shim_event_record(p->ev[1], stream) || k2a_shim_event_record(p->ev[2], stream)) goto err;
	return KSW2AMD_OK;
err:
	return fail(KSW2AMD_E_NODEVICE, "extf run: %s", k2a_shim_last_error());
}


/* ---------------------------------------------------------------- SSE-compatible mode (ksw2_lane_ssec.h) */

/* Process-wide default (ksw2amd_set_sse_compat, KSW2AMD_SSE_COMPAT=1 at load): every ksw_extz2_sse / ksw_extd2_sse call and
 * batch returns what the reference's SSE kernels return.  Per pair: KSW2AMD_EZ_SSE_COMPAT in the flags.  Independent of
 * both, KSW_EZ_APPROX_MAX | KSW_EZ_APPROX_DROP takes this path, because that heuristic follows one cell through the SSE
 * kernels' padded blocks and has no meaning outside them (KSW2AMD_APPROX_DROP_EXACT=1: the exact computation instead). */
static int g_sse_compat = -1;
void ksw2amd_set_sse_compat(int on) { g_sse_compat = on ? 1 : 0; }
static int wants_ssec(int flag)
{
	static int drop_exact = -1;
	if (g_sse_compat < 0) g_sse_compat = env_flag(ENV(SSE_COMPAT), 0);
	if (drop_exact < 0) drop_exact = env_flag(ENV(APPROX_DROP_EXACT), 0);
	if (g_sse_compat || (flag & KSW2AMD_EZ_SSE_COMPAT)) return 1;
	return (flag & KSW_EZ_APPROX_MAX) && (flag & KSW_EZ_APPROX_DROP) && !drop_exact;
}

/* state bytes per alignment up to which the SSE-compatible kernel keeps them in LDS.  What the LDS form gains in latency it loses
 * in wavefronts per CU: 512-base reads (4.6 KB) 106 -> 133 GCUPS, 2 048-base reads (18-22 KB: seven wavefronts per CU) 126 -> 79 */
#define SSEC_LDS_MAX ((size_t)8 * 1024)
static int ssec_ncol(int qlen, int tlen, int w)            /* = k2a_ssec_ncol (ksw2_lane_ssec.h) */
{
	const int n = imin(imin(qlen, tlen), w + 1);
	return ((n + 15) / 16 + 1) * 16;
}

static size_t ssec_pair_bytes(int dual, const ksw2amd_pair_t *a)
{
	const size_t ql = (size_t)imax(a->qlen, 0), tl = (size_t)imax(a->tlen, 0);
	const int mx = imax(a->qlen, a->tlen), w = (a->w < 0 || a->w > mx) ? mx : a->w;
	size_t b = ql + tl + 64 + (size_t)(dual ? 11 : 9) * (tl + 16) + sizeof(K2aPair) + sizeof(K2aResult);
	if (ql && tl && !(a->flag & KSW_EZ_SCORE_ONLY)) b += (ql + tl) * (size_t)ssec_ncol(a->qlen, a->tlen, w) + 4 * (ql + tl) + 512;
	return b;
}

ksw2amd_plan_t *ksw2amd_sse_plan_create(int dual, const ksw2amd_scoring_t *sc, int n, const ksw2amd_pair_t *pairs)
{
	ksw2amd_plan_t *p;
	int i, k, mode, m, q, e, q2, e2, lo;
	size_t off, mat_off;
	void *up;
	uint32_t fill[6];

	g_err[0] = 0;
	if (n < 0 || (n > 0 && !pairs) || !sc) { fail(KSW2AMD_E_PARAM, "sse plan: bad arguments%s", 0); return 0; }
	p = plan_new("sse plan", n, 1);
	if (!p) return 0;
	p->splice = 3; p->dual = !!dual; p->m = m = sc->m;
	q = sc->q; e = sc->e; q2 = dual ? sc->q2 : 0; e2 = dual ? sc->e2 : 0;
	for (i = 0; i < n; ++i) { p->h_cls[i] = -1; p->h_flag[i] = pairs[i].flag & ~F_SCALAR_CONTRACT; }
	/* ksw2_extz2_sse.c:57,78-82 / ksw2_extd2_sse.c:76,96-100 */
	if (m <= (dual ? 1 : 0) || !sc->mat) p->reject_all = 1;
	else if (m > K2A_MAXM) { fail(KSW2AMD_E_PARAM, "more than 127 residue types (int8_t m, ksw2.h:61)%s", 0); goto err; }
	else {
		p->c_par.qe_first = q + e;
		if (dual && q2 + e2 < q + e) { int t = q; q = q2; q2 = t; t = e; e = e2; e2 = t; }
		for (k = 1, lo = sc->mat[m * m > 1 ? 1 : 0]; k < m * m; ++k) lo = imin(lo, sc->mat[k]);
		if (-lo > 2 * (q + e)) p->reject_all = 1;
	}
	if (p->reject_all || n == 0) return p;

	memset(p->s_count, 0, sizeof(p->s_count));
	off = 0;
	for (i = 0; i < n; ++i) {
		const ksw2amd_pair_t *a = &pairs[i];
		K2aPair *d = &p->h_pairs[i];
		const int fl = p->h_flag[i];
		int w = a->w, mx, T16;
		if (a->qlen <= 0 || a->tlen <= 0) continue;
		if (!a->query || !a->target) { fail(KSW2AMD_E_PARAM, "sse plan: NULL sequence%s", 0); goto err; }
		mx = imax(a->qlen, a->tlen);
		if (w < 0 || w > mx) w = mx;                                       /* a wider band than the sequences changes nothing (ksw2_extz2_sse.c:72) */
		mode = (fl & KSW_EZ_SCORE_ONLY) ? K2A_MODE_SCORE : (fl & KSW_EZ_RIGHT) ? K2A_MODE_RIGHT : K2A_MODE_LEFT;
		{	/* state arrays of up to SSEC_LDS_MAX bytes live in LDS (k2a_ssec_kernel<.., LDS = true>); KSW2AMD_SSEC_HBM=1: never (tests) */
			const size_t sb = (size_t)(dual ? 11 : 9) * (size_t)((a->tlen + 15) / 16 * 16);
			const int lds = sb <= SSEC_LDS_MAX && !ENV(SSEC_HBM);
			p->h_cls[i] = (int8_t)(mode + 3 * lds);
			++p->s_count[mode][0][lds];
			if (lds && sb > p->c_lds[mode]) p->c_lds[mode] = sb;
		}
		d->qlen = a->qlen; d->tlen = d->tlen_full = a->tlen; d->w = w;
		d->zdrop = a->zdrop; d->end_bonus = a->end_bonus;
		d->flag = fl & (KSW_EZ_EXTZ_ONLY | KSW_EZ_REV_CIGAR | KSW_EZ_SCORE_ONLY);
		d->pad = ((fl & KSW_EZ_APPROX_MAX) ? K2A_SSEC_APPROX : 0) | (((fl & KSW_EZ_APPROX_MAX) && (fl & KSW_EZ_APPROX_DROP)) ? K2A_SSEC_APPROX_DROP : 0) |
		         ((fl & KSW_EZ_GENERIC_SC) ? K2A_SSEC_GENERIC : 0);
		off = align_up(off, 4); d->qoff = (uint32_t)off; off += (size_t)a->qlen;
		off = align_up(off, 4); d->toff = (uint32_t)off; off += (size_t)a->tlen;
		if (off > 0xfff00000u) { fail(KSW2AMD_E_PARAM, "sse plan: more than 4 GiB of sequence in one plan%s", 0); goto err; }
		T16 = (a->tlen + 15) / 16 * 16;
		d->bnd_off = (uint32_t)(p->bnd_words / 4);                           /* 16-byte units */
		p->bnd_words += (size_t)(dual ? 11 : 9) * (size_t)T16 / 4;
		if (p->bnd_words / 4 > 0xfff00000u) { fail(KSW2AMD_E_PARAM, "sse plan: state scratch over 64 GiB in one plan%s", 0); goto err; }
		p->cells += band_cells(a->qlen, a->tlen, w);
		if (mode != K2A_MODE_SCORE) {
			d->tb_off = p->tb_bytes;
			p->tb_bytes += align_up((size_t)(a->qlen + a->tlen - 1) * (size_t)ssec_ncol(a->qlen, a->tlen, w), 256);
			d->cig_off = (uint32_t)p->cig_words;
			p->cig_words += (size_t)a->qlen + a->tlen + 2;
			if (p->cig_words > 0xfff00000u) { fail(KSW2AMD_E_PARAM, "sse plan: CIGAR scratch over 16 GiB in one plan%s", 0); goto err; }
		}
	}
	off = align_up(off + 256, 256);
	mat_off = off; off = align_up(off + (size_t)m * m, 256);
	p->seq_bytes = off;
	p->h_seq = (uint8_t*)cache_get(BUF_HSEQ, p->seq_bytes, &p->cap[BUF_HSEQ]);
	if (!p->h_seq) { fail(KSW2AMD_E_NOMEM, "pinned staging allocation failed: %s", k2a_shim_last_error()); goto err; }
	for (k = 0, mode = 0; mode < 6; ++mode) { p->s_first[mode % 3][0][mode / 3] = k; fill[mode] = (uint32_t)k; k += p->s_count[mode % 3][0][mode / 3]; }
	p->ntasks = p->norder = k;
	for (i = 0; i < n; ++i) {
		if (p->h_cls[i] < 0) continue;
		memcpy(p->h_seq + p->h_pairs[i].qoff, pairs[i].query, (size_t)pairs[i].qlen);
		memcpy(p->h_seq + p->h_pairs[i].toff, pairs[i].target, (size_t)pairs[i].tlen);
		p->h_order[fill[p->h_cls[i]]++] = (uint32_t)i;
	}
	memcpy(p->h_seq + mat_off, sc->mat, (size_t)m * m);

	p->d_seq = (uint8_t*)cache_get(BUF_SEQ, p->seq_bytes, &p->cap[BUF_SEQ]);
	p->d_pairs = (K2aPair*)cache_get(BUF_PAIRS, sizeof(K2aPair) * ((size_t)n + 1), &p->cap[BUF_PAIRS]);
	p->d_res = (K2aResult*)cache_get(BUF_RES, sizeof(K2aResult) * ((size_t)n + 1), &p->cap[BUF_RES]);
	p->d_order = (uint32_t*)cache_get(BUF_ORDER, sizeof(uint32_t) * ((size_t)p->norder + 1), &p->cap[BUF_ORDER]);
	p->d_tb = p->tb_bytes ? (uint8_t*)cache_get(BUF_TB, p->tb_bytes, &p->cap[BUF_TB]) : 0;
	p->d_cig = p->cig_words ? (uint32_t*)cache_get(BUF_CIG, p->cig_words * 4, &p->cap[BUF_CIG]) : 0;
	p->d_bnd = p->bnd_words ? (int32_t*)cache_get(BUF_BND, p->bnd_words * 4, &p->cap[BUF_BND]) : 0;
	if (!p->d_seq || !p->d_pairs || !p->d_res || !p->d_order || (p->tb_bytes && !p->d_tb) || (p->cig_words && !p->d_cig) ||
	    (p->bnd_words && !p->d_bnd)) {
		fail(KSW2AMD_E_NOMEM, "device allocation failed: %s", k2a_shim_last_error());
		goto err;
	}
	up = thread_upload_stream();
	p->stream = up; p->stream_used = 1;
	if (k2a_shim_h2d(p->d_seq, p->h_seq, p->seq_bytes, up) || k2a_shim_h2d(p->d_pairs, p->h_pairs, sizeof(K2aPair) * (size_t)n, up) ||
	    k2a_shim_h2d(p->d_order, p->h_order, sizeof(uint32_t) * (size_t)p->norder, up) ||
	    k2a_shim_memset(p->d_res, 0, sizeof(K2aResult) * (size_t)n, up) || k2a_shim_stream_sync(up)) {
		fail(KSW2AMD_E_NODEVICE, "upload failed: %s", k2a_shim_last_error());
		goto err;
	}
	p->c_par.q = q; p->c_par.e = e; p->c_par.q2 = q2; p->c_par.e2 = e2; p->c_par.m = m;
	p->c_par.sc_mch = sc->mat[0]; p->c_par.sc_mis = sc->mat[m * m > 1 ? 1 : 0];
	p->c_par.sc_N = sc->mat[m * m - 1] == 0 ? -(dual ? e2 : e) : sc->mat[m * m - 1];     /* ksw2_extz2_sse.c:68, ksw2_extd2_sse.c:87 */
	if (dual) {                                                                             /* ksw2_extd2_sse.c:102-105 */
		int lt = e != e2 ? (q2 - q) / (e - e2) - 1 : 0;
		if (q2 + e2 + lt * e2 > q + e + lt * e) ++lt;
		p->c_par.long_thres = lt; p->c_par.long_diff = lt * (e - e2) - (q2 - q) - e2;
	}
	p->c_par.mat = (const int8_t*)p->d_seq + mat_off;
	plan_ready(p);                                  /* uploads complete */
	return p;
err:
	ksw2amd_plan_destroy(p);
	return 0;
}

static int ssec_plan_run(ksw2amd_plan_t *p, void *stream)
{
	int mode;
	p->stream = stream; p->ran = 1; p->stream_used = 1;
	if (p->up_ev && k2a_shim_stream_wait_event(stream, p->up_ev)) goto err;      /* the plan's upload (shared stream) before its kernels */
	if (p->reject_all || p->ntasks == 0) return KSW2AMD_OK;
	if (k2a_shim_event_record(p->ev[0], stream)) goto err;
	for (mode = 0; mode < 6; ++mode)
		if (p->s_count[mode % 3][0][mode / 3] &&
		    k2a_shim_launch_ssec(p->dual, mode % 3, mode / 3 ? p->c_lds[mode % 3] : 0, &p->c_par, p->d_pairs, p->d_order + p->s_first[mode % 3][0][mode / 3],
		                         p->s_count[mode % 3][0][mode / 3], p->d_seq, p->d_tb, (uint8_t*)p->d_bnd, p->d_res, stream)) goto err;
	if (k2a_shim_event_record(p->ev[1], stream)) goto err;
	for (mode = 0; mode < 6; ++mode)
		if (mode % 3 && p->s_count[mode % 3][0][mode / 3] &&
		    k2a_shim_launch_ssec_trace(p->d_pairs, p->d_order + p->s_first[mode % 3][0][mode / 3], p->s_count[mode % 3][0][mode / 3], p->d_tb, p->d_res, p->d_cig, stream)) goto err;
	if (k2a_shim_event_record(p->ev[2], stream)) goto err;
	return KSW2AMD_OK;
err:
	return fail(KSW2AMD_E_NODEVICE, "sse-compatible run: %s", k2a_shim_last_error());
}

/* n pairs through SSE-compatible plans sized to the device's free memory */
static int ssec_run(int dual, void *km, const ksw2amd_scoring_t *sc, int n, const ksw2amd_pair_t *pairs, ksw_extz_t *ez)
{
	int beg = 0;
	size_t budget, free_b = 0, total_b = 0;
	const char *env = ENV(MAX_BYTES);
	if (n <= 0) return KSW2AMD_OK;
	if (k2a_shim_device_count() <= 0) return fail(KSW2AMD_E_NODEVICE, "no usable %s device", k2a_shim_backend());
	if (env && atoll(env) > 0) budget = (size_t)atoll(env);
	else if (n == 1) budget = (size_t)1 << 36;
	else {
		if (k2a_shim_mem_info(&free_b, &total_b)) return fail(KSW2AMD_E_NODEVICE, "mem_info: %s", k2a_shim_last_error());
		budget = free_b / 10 * 7;
	}
	while (beg < n) {
		ksw2amd_plan_t *p;
		size_t acc = 0, seq = 0;
		int end, rc;
		for (end = beg; end < n; ++end) {
			const size_t b = ssec_pair_bytes(dual, &pairs[end]), sq = (size_t)imax(pairs[end].qlen, 0) + (size_t)imax(pairs[end].tlen, 0) + 8;
			if (end > beg && (acc + b > budget || seq + sq > 3000000000u || end - beg >= (1 << 22))) break;
			acc += b; seq += sq;
		}
		p = ksw2amd_sse_plan_create(dual, sc, end - beg, pairs + beg);
		if (!p) return strstr(g_err, "alloc") ? KSW2AMD_E_NOMEM : strstr(g_err, "device") ? KSW2AMD_E_NODEVICE : KSW2AMD_E_PARAM;
		rc = ksw2amd_plan_run(p, thread_stream());
		if (rc == KSW2AMD_OK) rc = ksw2amd_plan_fetch(p, km, ez + beg);
		ksw2amd_plan_destroy(p);
		if (rc) return rc;
		beg = end;
	}
	return KSW2AMD_OK;
}

static int extf_serial(void *km, int8_t mch, int8_t mis, int8_t e, int n, const ksw2amd_fpair_t *pairs, ksw_extz_t *ez)
{
	int beg = 0;
	if (n <= 0) return KSW2AMD_OK;
	if (k2a_shim_device_count() <= 0) return fail(KSW2AMD_E_NODEVICE, "no usable %s device", k2a_shim_backend());
	while (beg < n) {
		ksw2amd_plan_t *p;
		size_t seq = 0;
		int end, rc;
		for (end = beg; end < n; ++end) {
			const size_t b = (size_t)imax(pairs[end].qlen, 0) + 4 * (size_t)imax(pairs[end].tlen, 0) + 64;
			if (end > beg && (seq + b > 3000000000u || end - beg >= (1 << 22))) break;
			seq += b;
		}
		p = ksw2amd_extf_plan_create(mch, mis, e, end - beg, pairs + beg);
		if (!p) return strstr(g_err, "alloc") ? KSW2AMD_E_NOMEM : strstr(g_err, "device") ? KSW2AMD_E_NODEVICE : KSW2AMD_E_PARAM;
		rc = ksw2amd_plan_run(p, thread_stream());
		if (rc == KSW2AMD_OK) rc = ksw2amd_plan_fetch(p, km, ez + beg);
		ksw2amd_plan_destroy(p);
		if (rc) return rc;
		beg = end;
	}
	return KSW2AMD_OK;
}

typedef struct { void *km; int8_t mch, mis, e; const ksw2amd_fpair_t *pairs; ksw_extz_t *ez; } extf_ctx_t;
static int extf_chunk(void *ctx_, int beg, int end, int share, pend_t *pd)
{
	extf_ctx_t *c = (extf_ctx_t*)ctx_;
	(void)pd; (void)share;
	if (beg < 0) return KSW2AMD_OK;
	return extf_serial(c->km, c->mch, c->mis, c->e, end - beg, c->pairs + beg, c->ez + beg);
}

int ksw2amd_extf_batch(void *km, int8_t mch, int8_t mis, int8_t e, int n, const ksw2amd_fpair_t *pairs, ksw_extz_t *ez)
{
	const int tpd = pool_threads_per_device();
	/* (batches big enough for the one-extension-per-lane form stay whole: it needs every wavefront it can get) */
	if (n >= (pool_min_pairs() ? pool_min_pairs() : 2048) && n < 131072 && tpd > 0 && !g_is_worker && k2a_shim_device_count() > 0) {
		const int workers = tpd * (g_ndev_set > 0 ? g_ndev_set : 1);
		double *cost = (double*)malloc(sizeof(double) * (size_t)n), total = 0;
		int i, rc = 0, uniform = 1, chunk_pairs = 0, nchunks;
		for (i = 1; i < n && uniform; ++i) uniform = pairs[i].qlen == pairs[0].qlen && pairs[i].tlen == pairs[0].tlen && pairs[i].w == pairs[0].w;
		nchunks = wave_chunks(n, workers, uniform, &chunk_pairs);
		if (cost && nchunks >= 2) {
			extf_ctx_t ctx;
			for (i = 0; i < n; ++i) { cost[i] = 1.0 + (double)imax(pairs[i].qlen, 0) + imax(pairs[i].tlen, 0); total += cost[i]; }
			ctx.km = km; ctx.mch = mch; ctx.mis = mis; ctx.e = e; ctx.pairs = pairs; ctx.ez = ez;
			if (run_pooled(extf_chunk, &ctx, n, cost, total, nchunks, chunk_pairs, &rc)) { free(cost); return rc; }
		}
		free(cost);
	}
	return extf_serial(km, mch, mis, e, n, pairs, ez);
}

void ksw_extf2_sse(void *km, int qlen, const uint8_t *query, int tlen, const uint8_t *target, int8_t mch, int8_t mis, int8_t e, int w, int xdrop,
                   ksw_extz_t *ez)
{
	ksw2amd_fpair_t pr;
	pr.query = query; pr.target = target; pr.qlen = qlen; pr.tlen = tlen; pr.w = w; pr.xdrop = xdrop;
	{
		const int rc = ksw2amd_extf_batch(km, mch, mis, e, 1, &pr, ez);
		if (rc != KSW2AMD_OK) call_failed("ksw_extf2_sse", rc, ez);
	}
}
