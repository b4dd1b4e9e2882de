/*
 * ksw2_host_int.h -- what the host objects of libksw2_amd.so share: ksw2_host_plan.c (switches, buffer cache, plans), ksw2_host_pool.c
 * (worker pool, batch entry points, flat batches), ksw2_host_single.c (the ksw2-named calls, coalescing), ksw2_host_ext.c (exts, extf, SSE-compatible mode).
 * Nothing here is part of the drop-in boundary (include/ksw2_amd.h); every symbol declared below is hidden in the shared library.
 */
#ifndef KSW2_HOST_INT_H_
#define KSW2_HOST_INT_H_
#define _GNU_SOURCE
#include <dlfcn.h>
#include <execinfo.h>
#include <signal.h>
#include <unistd.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#include <pthread.h>
#include <sched.h>
#include <string.h>
#include <time.h>
#include "../../include/ksw2_amd.h"
#include "ksw2_shim.h"


#define F_SCALAR_CONTRACT 0x40000000   /* internal: call came through ksw_extz / ksw_extd / ksw_gg* */
#define NCLS_MAX (K2A_NCFG * 3 * 2)
#define NPASS (2 + 4 * K2A_NPKCFG)     /* per class: one alignment per lane group | packed class pc (x re-based) (x no maximum tracking) | solo */
#define PASS_SOLO (NPASS - 1)           /* both halves of the packed registers for one alignment (ksw2_lane_solo.h) */
#define NCLS_ENTRIES (NCLS_MAX * NPASS)

#pragma GCC visibility push(hidden)
#define K2A_ENV_LIST \
	X(ABORT_ON_ERROR) \
	X(BACKTRACE) \
	X(APPROX_DROP_EXACT) \
	X(CHUNK_GCELLS) \
	X(CHUNK_MB) \
	X(COALESCE_SLOTS) \
	X(COALESCE_WINDOW_US) \
	X(COALESCE_PLAIN_STREAMS) \
	X(COALESCE_CROWD) \
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
	X(PIN) \
	X(PK_FIRST) \
	X(POOL_MIN) \
	X(SERIAL) \
	X(SIMDS) \
	X(SMALL_CELLS) \
	X(SOLO) \
	X(SSEC_HBM) \
	X(SSEC_BLK) \
	X(EXTF_GRP) \
	X(STREAM_LANES) \
	X(WIRE4) \
	X(WIRE2) \
	X(PLAIN_UP_STREAMS) \
	X(WORKER_PRIO) \
	X(SSE_COMPAT) \
	X(STREAM) \
	X(STREAM_FAULT) \
	X(STREAM_MIN_CELLS) \
	X(STREAM_PIECE_KB) \
	X(UNIFORM) \
	X(STREAM_SLEEP_US) \
	X(STREAM_TIMEOUT_MS) \
	X(THREADS) \
	X(TN) \
	X(TRACE)
enum {
#define X(n) ENV_##n,
	K2A_ENV_LIST
#undef X
	ENV_COUNT
};
enum { BUF_HSEQ, BUF_SEQ, BUF_PAIRS, BUF_RES, BUF_ORDER, BUF_TB, BUF_CIG, BUF_BND, BUF_POS, BUF_POOL, BUF_HPOOL, BUF_HRES, BUF_WM, BUF_HMETA, BUF_PK4, BUF_KINDS };
#define BUF_IS_HOST(k) ((k) == BUF_HSEQ || (k) == BUF_HPOOL || (k) == BUF_HRES || (k) == BUF_HMETA)      /* pinned host staging; everything else is device memory */
#define CACHE_DEPTH 2                  /* a worker that queues its next chunk before it fetches the current one holds two plans */
#define SHARED_UP_MAXDEV 16
#define POOL_MAXW 64
#define POOL_MAXDEV 16

extern __thread void *g_plan_stream;
extern __thread int g_latency_plan;
extern int64_t g_reruns;
extern int64_t g_stat[4];
extern __thread int g_is_worker;
extern int g_ndev_set, g_dev_set[POOL_MAXDEV];
void wire4_pair(const ksw2amd_plan_t *p, int i, uint8_t *out);
int plan_wire4(const ksw2amd_plan_t *p);
void phase_add(double create_ms, double launch_ms, double fetch_ms);      /* ksw2amd_host_phase_us */
extern const char *g_env[ENV_COUNT];
extern volatile int g_env_ready;
extern __thread char g_err[512];
extern __thread int g_no_defer;

static int fail(int code, const char *fmt, const char *detail)
{
	snprintf(g_err, sizeof(g_err), fmt, detail ? detail : "");
	return code;
}

#define ENV(n) (g_env_ready ? g_env[ENV_##n] : (env_load(), g_env[ENV_##n]))
static inline int env_flag(const char *e, int dflt) { return e && *e ? atoi(e) != 0 : dflt; }
static inline int env_switch(const char *v) { return v && *v ? (atoi(v) != 0) : -1; }      /* -1 = automatic */

#define K2A_MAXPIECES 48
#define K2A_STREAM_MARGIN 256          /* bytes past a sequence's end that the kernels may touch (dword query loads, one strip of target codes) */
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
	int wire4, wire_bad;                   /* uniform plans: the staging buffer and the upload hold 2^wire4 codes per byte (1: the 4-bit wire format, 2: the 2-bit one with escape entries, ksw2_lane.h); wire_bad: a source byte above 15, or more escapes in a pair than its slot holds (the batch is repeated on the general path) */
	uint32_t wire_stride;                  /* ... the pairs' stride in the arena (the 2-bit format's escape slots end a pair's region) */
	void *up2, *ev2[K2A_MAXPIECES];        /* two copy lanes (KSW2AMD_STREAM_LANES): odd pieces travel on a second stream, an event behind each (stream_issue) */
	int rc;
	pthread_mutex_t mu;
} stream_up_t;

static inline int is_approx(int flag)
{
	return !(flag & F_SCALAR_CONTRACT) && (flag & KSW_EZ_APPROX_MAX) && !(flag & KSW_EZ_APPROX_DROP);
}
static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }
static inline int imin(int a, int b) { return a < b ? a : b; }
static inline int imax(int a, int b) { return a > b ? a : b; }

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
	int f_first[10], f_count[10];        /* [6]: one extension per lane, groups of 64 with interleaved sequences (ksw2_lane_extf.h); [7..9]: four / two / one per wavefront in registers (ksw2_lane_extfb.h) */
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
	K2aUniform *uni;                       /* uniform plans (plan_create_uniform): records, task list and piece counts are written on the device by rule */
	uint8_t *d_pk4;                  /* uniform plans on the 4-bit wire format: the device copy of the upload (the arena the kernels read is d_seq) */
	int clear_res, clear_bnd;        /* the first run clears the result records / fills the boundary scratch on its own stream (never the shared upload stream) */
	size_t need_words;                     /* per-wavefront-task piece counts, behind the task lists in d_order */
};
typedef struct { int64_t cost; uint32_t idx, tf; } sort_t;      /* tf = true target length: part of a packed pair's shape */
typedef struct { uint8_t *h_seq; const K2aPair *hp; const ksw2amd_pair_t *pairs; uint8_t *wild; stream_up_t *su;      /* su: streamed plans -- chunk k of the copy is piece k of the upload */
                 ksw2amd_plan_t *uni_plan; int uni_cls, uni_flag; } copy_ctx_t;                                              /* uniform plans: the copy's workers fill the host's per-pair arrays of their range first */
typedef struct { int on_device; } flat_src_t;
#define META_ROOM(n) (align_up(sizeof(K2aPair) * ((size_t)(n) + 1), 256) + sizeof(uint32_t) * (3 * (size_t)(n) + 8) + 512)
typedef struct { ksw2amd_plan_t *p; void *km; ksw_extz_t *ez, **ezp; void **kmp; const uint32_t *pool; const size_t *pos; int nrerun, rc; } asm_ctx_t;
typedef struct { ksw2amd_plan_t *p; int beg; } pend_t;      /* a worker's plan that is computing while the worker packs the next chunk */
typedef int (*chunk_fn)(void *ctx, int beg, int end, int share, pend_t *pd);   /* beg < 0: finish what is pending */

int exts_plan_run(ksw2amd_plan_t *p, void *stream);
int extf_plan_run(ksw2amd_plan_t *p, void *stream);
int wants_ssec(int flag);
int ssec_plan_run(ksw2amd_plan_t *p, void *stream);
int ssec_run(int dual, void *km, const ksw2amd_scoring_t *sc, int n, const ksw2amd_pair_t *pairs, ksw_extz_t *ez);
void *thread_stream(void);
void *thread_upload_stream(void);
void *cache_get(int kind, size_t bytes, size_t *cap);
void cache_put(int kind, void *p, size_t cap);
void release_thread_cache(void);
int stream_env(void);
int64_t stream_min_cells(void);
void stream_issue(stream_up_t *u, int k);
void ez_reset(ksw_extz_t *ez);
void ez_reserve(void *km, ksw_extz_t *ez, int n);
int64_t band_cells(int qlen, int tlen, int w);
int geom_fits(int G, int C, int tlen_eff, int w);
void build_eff(int dual, int m, const int8_t *mat, int e, int e2, int generic, int8_t *eff);
int copy_scan(uint8_t *dst, const uint8_t *src, int n);
int cmp_cost_desc(const void *a, const void *b);
int trace_level(void);
ksw2amd_plan_t *plan_new(const char *who, int n, int with_order);
void plan_ready(ksw2amd_plan_t *p);
ksw2amd_plan_t *plan_create_ex(int dual, int scalar, const ksw2amd_scoring_t *sc, int n, const ksw2amd_pair_t *pairs, const flat_src_t *flat, int want_stream);
void eqx_rewrite(void *km, const uint8_t *query, const uint8_t *target, int stride, ksw_extz_t *ez);
void km_lock(const void *km);
void km_unlock(const void *km);
int needs_rerun(const ksw2amd_plan_t *p, int i);
int pair_rerun(ksw2amd_plan_t *p, int i, void *km, ksw_extz_t *z);
void assemble_range(asm_ctx_t *c, int beg, int end);
int plan_fetch_ex(ksw2amd_plan_t *p, void *km, ksw_extz_t *ez, ksw_extz_t **ezp, void **kmp);
size_t device_budget(size_t free_b, size_t total_b, int share);
size_t pair_device_bytes(int dual, const ksw2amd_pair_t *a);
int run_serial(int dual, int scalar, void *km, const ksw2amd_scoring_t *sc, int n, const ksw2amd_pair_t *pairs, ksw_extz_t *ez, int share, const flat_src_t *flat, int want_stream);
int pool_threads_per_device(void);
void copy_range(const copy_ctx_t *c, int beg, int end);
int parallel_copy(copy_ctx_t *c, int n, size_t bytes);
int assemble_parallel(asm_ctx_t *c);
int rerun_pairs(ksw2amd_plan_t *p, int nrerun, void *km, ksw_extz_t *ez, ksw_extz_t **ezp, void **kmp);
int gather_start(ksw2amd_plan_t *p, stream_up_t *su, const ksw2amd_pair_t *pairs, int n);
int gather_wait(ksw2amd_plan_t *p);
int gather_start_uniform(ksw2amd_plan_t *p, stream_up_t *su, const ksw2amd_pair_t *pairs, int n, int cls0, int flag0);
void uni_fill_range(const K2aUniform *u, K2aPair *hp, int8_t *h_cls, int32_t *h_flag, uint32_t *h_order, int cls0, int flag0, int beg, int end);
ksw2amd_plan_t *plan_create_uniform_entry(int dual, int scalar, const ksw2amd_scoring_t *sc, int n, const ksw2amd_pair_t *pairs);
int pool_min_pairs(void);
int unit_pairs(const ksw2amd_pair_t *a);
int uniform_chunks(int n, int unit, double bytes, double cells, int workers, int ndev, int with_cigar, double path_steps, int *chunk_pairs);
double now_ms(void);
int64_t now_ns(void);
int trace_on(void);
int run_pooled(chunk_fn fn, void *ctx, int n, const double *cost, double total, int nchunks, int chunk_pairs, int *rc);
void call_failed(const char *fn, int code, ksw_extz_t *ez);
void env_load(void);
size_t thread_cached_device_bytes(void);
void *shared_upload_stream(void);
int exts_chunk(void *ctx_, int beg, int end, int share, pend_t *pd);
int extf_chunk(void *ctx_, int beg, int end, int share, pend_t *pd);
#pragma GCC visibility pop
#endif
