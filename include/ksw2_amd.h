/*
 * ksw2_amd.h -- C-ABI of libksw2_amd.so: MI355X (gfx950) implementation of ksw2's banded
 * extension / global alignment hot path.
 *
 * Part 1 is the ksw2 calling surface (same symbol names, argument order and meaning, ksw_extz_t layout
 * and CIGAR encoding as the reference's ksw2.h), so a minimap2-style caller that was compiled against
 * the reference header links against this library unchanged.  Part 2 is the batched front-end the
 * reference does not have: thousands of independent (query, target, band) pairs per launch.
 *
 * Result contract: every function evaluates the alignment on the GPU with the exact
 * band |i-j| <= w and the row-wise Z-drop of the reference's scalar ksw_extz / ksw_extd, and returns
 * score / max / max_q / max_t / mqe / mqe_t / mte / mte_q / zdropped / reach_end / CIGAR bit-identical to THOSE.
 * Where the reference's SSE functions differ from their scalar twins, the default results therefore differ from an SSE
 * build's: mte_q (the SSE kernels report it from their 16-padded range: ~94 % of calls), scores at the edge of narrow
 * bands (their 16-position blocks leak), Z-drop (theirs is per anti-diagonal), max_t / max_q on ties, ksw_gg2_sse on
 * narrow bands; two reference bugs are not reproduced (ksw_extd2_sse with e == e2; its first cell when it swaps the
 * gap pieces).  The SSE-compatible mode (KSW2AMD_EZ_SSE_COMPAT, below) returns the SSE functions' own results instead,
 * every field and the CIGAR, and KSW_EZ_APPROX_MAX | KSW_EZ_APPROX_DROP always does.
 * There is no CPU fallback: without a usable gfx950 device the ksw2-named entry points (which return void) reset *ez
 * (score = KSW_NEG_INF, no CIGAR), count the failure (ksw2amd_error_count), keep its message (ksw2amd_last_error) and report it
 * on stderr or through ksw2amd_set_error_handler -- they never abort across the C boundary unless KSW2AMD_ABORT_ON_ERROR=1 asks
 * for it; the ksw2amd_* entry points return a negative code.
 *
 * Reference interface each declaration replaces is cited as (ksw2.h:LINE).
 */
#ifndef KSW2_AMD_H_
#define KSW2_AMD_H_

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ Part 1: the ksw2 surface */
#ifndef KSW2_H_          /* a caller may include the reference's ksw2.h instead: identical ABI */

#define KSW_NEG_INF        (-0x40000000)          /* (ksw2.h:6)  */

#define KSW_EZ_SCORE_ONLY  0x01                   /* (ksw2.h:8)  no CIGAR */
#define KSW_EZ_RIGHT       0x02                   /* (ksw2.h:9)  right-align gaps */
#define KSW_EZ_GENERIC_SC  0x04                   /* (ksw2.h:10) use mat[] for every residue pair */
#define KSW_EZ_APPROX_MAX  0x08                   /* (ksw2.h:11) alone: like the reference only score + corner CIGAR are returned */
#define KSW_EZ_APPROX_DROP 0x10                   /* (ksw2.h:12) with APPROX_MAX: routed to the SSE-compatible kernels, which reproduce the reference's drop heuristic (see KSW2AMD_EZ_SSE_COMPAT below) */
#define KSW_EZ_EXTZ_ONLY   0x40                   /* (ksw2.h:13) extension only */
#define KSW_EZ_REV_CIGAR   0x80                   /* (ksw2.h:14) CIGAR in end->start order */
#define KSW_EZ_SPLICE_FOR   0x100                  /* (ksw2.h:15) exts2: GT..AG signals (forward transcript strand) */
#define KSW_EZ_SPLICE_REV   0x200                  /* (ksw2.h:16) exts2: CT..AC signals (reverse strand) */
#define KSW_EZ_SPLICE_FLANK 0x400                  /* (ksw2.h:17) exts2: half penalty for GT / AG without the preferred flank */
#define KSW_EZ_EQX         0x800                  /* (ksw2.h:18) =/X instead of M (extd2 only) */

#define KSW_CIGAR_MATCH  0                        /* (ksw2.h:22-27) */
#define KSW_CIGAR_INS    1
#define KSW_CIGAR_DEL    2
#define KSW_CIGAR_N_SKIP 3
#define KSW_CIGAR_EQ     7
#define KSW_CIGAR_X      8

/* (ksw2.h:33-42) 56 bytes; cigar[k] = len<<4 | op; the callee reuses and grows `cigar` (capacity
 * m_cigar), the caller zero-initialises the struct once and finally frees `cigar`. */
typedef struct {
	uint32_t max:31, zdropped:1;
	int max_q, max_t;
	int mqe, mqe_t;
	int mte, mte_q;
	int score;
	int m_cigar, n_cigar;
	int reach_end;
	uint32_t *cigar;
} ksw_extz_t;

#endif /* KSW2_H_ */

/* `km`: NULL -> CIGAR memory comes from libc realloc (caller frees with free()); non-NULL -> the
 * caller's kalloc pool: the library calls the process's krealloc(km, ..) (kalloc.h:14), resolved at run time. */

/* (ksw2.h:64-65) single affine gap, replaces ksw_extz2_sse (ksw2_extz2_sse.c:23-304) */
void ksw_extz2_sse(void *km, int qlen, const uint8_t *query, int tlen, const uint8_t *target, int8_t m, const int8_t *mat,
                   int8_t q, int8_t e, int w, int zdrop, int end_bonus, int flag, ksw_extz_t *ez);
/* (ksw2.h:70-71) two-piece affine gap, replaces ksw_extd2_sse (ksw2_extd2_sse.c:34-409) */
void ksw_extd2_sse(void *km, int qlen, const uint8_t *query, int tlen, const uint8_t *target, int8_t m, const int8_t *mat,
                   int8_t gapo, int8_t gape, int8_t gapo2, int8_t gape2, int w, int zdrop, int end_bonus, int flag, ksw_extz_t *ez);
/* (ksw2.h:89-90) global alignment, replace ksw_gg2 (ksw2_gg2.c:4-114) and ksw_gg2_sse (ksw2_gg2_sse.c:11-126).
 * ksw_gg2: all three CIGAR pointers NULL -> score only.  Band that cannot reach the corner
 * (w < |tlen-qlen|): returns KSW_NEG_INF with *n_cigar_ = 0 (undefined in the reference). */
int ksw_gg2(void *km, int qlen, const uint8_t *query, int tlen, const uint8_t *target, int8_t m, const int8_t *mat,
            int8_t gapo, int8_t gape, int w, int *m_cigar_, int *n_cigar_, uint32_t **cigar_);
int ksw_gg2_sse(void *km, int qlen, const uint8_t *query, int tlen, const uint8_t *target, int8_t m, const int8_t *mat,
                int8_t gapo, int8_t gape, int w, int *m_cigar_, int *n_cigar_, uint32_t **cigar_);
/* (ksw2.h:61-62, 67-68, 88) the scalar entry points, evaluated by the same GPU kernels
 * (they are the functions whose results define the contract) */
void ksw_extz(void *km, int qlen, const uint8_t *query, int tlen, const uint8_t *target, int8_t m, const int8_t *mat,
              int8_t q, int8_t e, int w, int zdrop, int flag, ksw_extz_t *ez);
void ksw_extd(void *km, int qlen, const uint8_t *query, int tlen, const uint8_t *target, int8_t m, const int8_t *mat,
              int8_t gapo, int8_t gape, int8_t gapo2, int8_t gape2, int w, int zdrop, int flag, ksw_extz_t *ez);
int ksw_gg(void *km, int qlen, const uint8_t *query, int tlen, const uint8_t *target, int8_t m, const int8_t *mat,
           int8_t gapo, int8_t gape, int w, int *m_cigar_, int *n_cigar_, uint32_t **cigar_);
/* multi-ISA builds of the reference export these names under KSW_CPU_DISPATCH
 * (ksw2_extz2_sse.c:16-24, ksw2_extd2_sse.c:25-36); same functions */
void ksw_extz2_sse41(void *km, int qlen, const uint8_t *query, int tlen, const uint8_t *target, int8_t m, const int8_t *mat,
                     int8_t q, int8_t e, int w, int zdrop, int end_bonus, int flag, ksw_extz_t *ez);
void ksw_extz2_sse2(void *km, int qlen, const uint8_t *query, int tlen, const uint8_t *target, int8_t m, const int8_t *mat,
                    int8_t q, int8_t e, int w, int zdrop, int end_bonus, int flag, ksw_extz_t *ez);
void ksw_extd2_sse41(void *km, int qlen, const uint8_t *query, int tlen, const uint8_t *target, int8_t m, const int8_t *mat,
                     int8_t gapo, int8_t gape, int8_t gapo2, int8_t gape2, int w, int zdrop, int end_bonus, int flag, ksw_extz_t *ez);
void ksw_extd2_sse2(void *km, int qlen, const uint8_t *query, int tlen, const uint8_t *target, int8_t m, const int8_t *mat,
                    int8_t gapo, int8_t gape, int8_t gapo2, int8_t gape2, int w, int zdrop, int end_bonus, int flag, ksw_extz_t *ez);

/* (ksw2.h:73-74) splice-aware extension, replaces ksw_exts2_sse (ksw2_exts2_sse.c:33-415): affine gap (q, e) plus a long
 * target-consuming gap ("intron") with open cost q2, no extension cost, penalty `noncan` at non-canonical donor /
 * acceptor sites (flag KSW_EZ_SPLICE_FOR / _REV / _FLANK), junc[t] bits 1/2 (forward donor/acceptor), 8/4 (reverse)
 * rewarded with junc_bonus.  Unbanded; Z-drop, max and mqe / mte per anti-diagonal exactly like the reference,
 * CIGAR with N for introns.  Bit-exact against the reference's SSE kernel in exact-max mode.
 * Diagonals of up to 1472 cells (min(qlen, tlen)) run from registers, longer ones from an HBM-resident state (slower). */
void ksw_exts2_sse(void *km, int qlen, const uint8_t *query, int tlen, const uint8_t *target, int8_t m, const int8_t *mat,
                   int8_t gapo, int8_t gape, int8_t gapo2, int8_t noncan, int zdrop, int8_t junc_bonus, int flag, const uint8_t *junc, ksw_extz_t *ez);
void ksw_exts2_sse41(void *km, int qlen, const uint8_t *query, int tlen, const uint8_t *target, int8_t m, const int8_t *mat,
                     int8_t gapo, int8_t gape, int8_t gapo2, int8_t noncan, int zdrop, int8_t junc_bonus, int flag, const uint8_t *junc, ksw_extz_t *ez);
void ksw_exts2_sse2(void *km, int qlen, const uint8_t *query, int tlen, const uint8_t *target, int8_t m, const int8_t *mat,
                    int8_t gapo, int8_t gape, int8_t gapo2, int8_t noncan, int zdrop, int8_t junc_bonus, int flag, const uint8_t *junc, ksw_extz_t *ez);

/* (ksw2.h:76) gap-linear X-drop extension, score only; replaces ksw_extf2_sse (ksw2_extf2_sse.c:11-98): match `mch`,
 * mismatch -|mis|, gap cost e per base, band w (< 0: none), X-drop on the one cell per anti-diagonal the reference follows.
 * Fills ez->max, max_t, max_q, score (when every anti-diagonal ran) and zdropped; the other fields stay reset.  The
 * reference's results depend on the 16-byte blocking of its SSE loops (cells outside the band are updated too and feed
 * the band's edge); reproduced bit for bit for its SSE4.1 build.  Targets up to 21504 residues keep their state in LDS,
 * longer ones in HBM (slower). */
void ksw_extf2_sse(void *km, int qlen, const uint8_t *query, int tlen, const uint8_t *target, int8_t mch, int8_t mis, int8_t e, int w, int xdrop,
                   ksw_extz_t *ez);

/* ------------------------------------------------------------------ Part 2: batched front-end */

/* scoring shared by every pair of a batch (the arguments m, mat, q, e[, q2, e2] of the calls above) */
typedef struct {
	int32_t m;                 /* residue codes 0..m-1, last one is the wildcard (m <= 127) */
	const int8_t *mat;         /* m*m, mat[target*m + query] */
	int8_t q, e, q2, e2;       /* q2/e2 only read by the two-piece (extd) entry points */
} ksw2amd_scoring_t;

/* one alignment = the per-call arguments of ksw_extz2_sse / ksw_extd2_sse */
typedef struct {
	const uint8_t *query, *target;
	int32_t qlen, tlen;
	int32_t w, zdrop, end_bonus, flag;
} ksw2amd_pair_t;

#define KSW2AMD_OK            0
#define KSW2AMD_E_NODEVICE   (-1)      /* no usable gfx950 device / HIP runtime error */
#define KSW2AMD_E_PARAM      (-2)      /* argument outside what this release supports (see last_error) */
#define KSW2AMD_E_NOMEM      (-3)

const char *ksw2amd_last_error(void);  /* message of the calling thread's last failing call */
const char *ksw2amd_backend(void);     /* "hip:gfx950" */
int ksw2amd_device_count(void);
int ksw2amd_set_device(int device);    /* device used by the calling thread's subsequent calls */
/* each thread keeps the device / pinned buffers of its last batch for reuse (allocation costs milliseconds); this returns the
 * calling thread's and those of the library's worker threads */
void ksw2amd_release_cache(void);
/* Devices the batch entry points below shard their work over (process-wide; pairs are independent, so this is host-side
 * sharding without any collective: the library's worker threads -- KSW2AMD_THREADS per device, default 6 -- pull chunks of
 * the batch from a shared counter).  n = 0 (the default): the calling thread's current device only.
 * Replaces nothing in the reference (ksw2.h has no device notion); SURVEY.md section 8b "device selection / multi-GPU inside". */
int ksw2amd_set_devices(int n, const int *devices);
/* The ksw2-named functions return void (ksw2.h:61-76), so a device failure has no return channel.  The failing call returns with
 * *ez reset (ksw2.h:184-189: score = KSW_NEG_INF, no CIGAR), ksw2amd_error_count() goes up, ksw2amd_last_error() holds the message;
 * by default it is also printed on stderr (first occurrences), with a handler installed the handler gets (function name,
 * KSW2AMD_E_* code, message) instead.  KSW2AMD_ABORT_ON_ERROR=1: abort() after the report. */
typedef void (*ksw2amd_error_fn)(const char *func, int code, const char *msg, void *user);
void ksw2amd_set_error_handler(ksw2amd_error_fn fn, void *user);
long ksw2amd_error_count(void);        /* failed ksw2-named calls so far */

/* Opt-in host path for tiny single calls (replaces nothing of the reference; its one-pair-per-call pattern is cli.c:50-132 /
 * README.md:54-87).  A ksw2-named single-pair call costs a launch and two PCIe round trips here, ~0.5 ms whatever its size.  With
 * cells > 0, every such call whose exact band has at most `cells` DP cells is computed on the calling thread by the library's own
 * scalar code instead -- same result contract, bit for bit (tests/test_small_calls.py) -- and ksw2amd_small_call_count() counts
 * them.  Default 0 = never (KSW2AMD_SMALL_CELLS sets the default).  Not a fallback: it is never taken because something failed,
 * the batch entry points never use it, and the device is initialised by the first call all the same.  KSW_EZ_APPROX_MAX requests
 * and the SSE-compatible mode always go to the device.  Measured crossover and rates: INTEGRATION.md. */
void ksw2amd_set_small_call_cells(int64_t cells);
long ksw2amd_small_call_count(void);
/* diagnostics: { batches run on the worker pool, chunks they were cut into, single-pair calls that were coalesced with other
 * threads' calls, device batches those formed } since the library was loaded */
void ksw2amd_host_stats(int64_t out[4]);
/* where the batch entry points' host threads spent their time, in microseconds summed over threads since the library was loaded:
 * { plan creation (pack / gather into page-locked staging, uploads issued), launch calls, waiting for the device + fetch + ksw_extz_t
 * assembly, plans run }.  For a multi-rank job's per-rank report (bench.py config.per_rank): which side a slow rank was slow on. */
void ksw2amd_host_phase_us(int64_t out[4]);

/* n independent alignments; ez[i] ends up exactly as after
 *   ksw_extz2_sse(km, pairs[i].qlen, pairs[i].query, ..., sc->m, sc->mat, sc->q, sc->e, w, zdrop, end_bonus, flag, &ez[i])
 * (resp. ksw_extd2_sse).  ez[] must be zero-initialised (or hold reusable cigar buffers) like for the
 * single calls.  Large batches are split internally to fit device memory. */
int ksw2amd_extz_batch(void *km, const ksw2amd_scoring_t *sc, int n, const ksw2amd_pair_t *pairs, ksw_extz_t *ez);
int ksw2amd_extd_batch(void *km, const ksw2amd_scoring_t *sc, int n, const ksw2amd_pair_t *pairs, ksw_extz_t *ez);

/* Flat batches (new; no reference counterpart): every sequence of the batch lies in ONE arena and a pair is two offsets and two
 * lengths -- what a caller that reads its sequences into a buffer has anyway.  The library then uploads the arena's span as it
 * is (one copy instead of 2 n gathers into staging; asynchronous and at link rate if the arena is page-locked, ksw2amd_host_register)
 * and never walks over the bytes on the host: a pair with a wildcard code in a packed-int16 kernel is reported by the kernel
 * and run again through the int32 kernels.  `on_device` != 0: `base` is device memory (a hipMalloc'ed arena of the calling
 * thread's device, e.g. a shard that RCCL delivered): no upload at all (pairs that ask for the SSE kernels' own results, which keep
 * their state on the host side of the plan, have the arena's span brought back first).  Per-pair arrays w / zdrop / end_bonus / flag may be NULL:
 * the *_all value then applies to every pair.  Results are those of the ordinary entry points on the same pairs. */
typedef struct {
	const uint8_t *base;                  /* the arena */
	const uint64_t *qoff, *toff;          /* [n] byte offsets of query / target in the arena (one plan spans at most 4 GiB of it) */
	const int32_t *qlen, *tlen;           /* [n] */
	const int32_t *w, *zdrop, *end_bonus, *flag;      /* [n] or NULL */
	int32_t w_all, zdrop_all, end_bonus_all, flag_all;
	int32_t on_device;
} ksw2amd_flat_t;
int ksw2amd_extz_batch_flat(void *km, const ksw2amd_scoring_t *sc, int n, const ksw2amd_flat_t *in, ksw_extz_t *ez);
int ksw2amd_extd_batch_flat(void *km, const ksw2amd_scoring_t *sc, int n, const ksw2amd_flat_t *in, ksw_extz_t *ez);
/* device memory of the calling thread's device for a device-resident arena, without linking the HIP runtime (synchronous copies) */
void *ksw2amd_device_alloc(size_t bytes);
void  ksw2amd_device_free(void *d);
int   ksw2amd_device_upload(void *dst, const void *src, size_t bytes);
int   ksw2amd_device_download(void *dst, const void *src, size_t bytes);
/* page-lock / release caller memory (an arena that is reused from batch to batch): uploads from it need no staging copy */
int ksw2amd_host_register(const void *p, size_t bytes);
int ksw2amd_host_unregister(const void *p);

/* SSE-compatible mode (opt-in).  By default the "...2_sse" functions above return the exact-band, row-wise results of the
 * scalar ksw_extz / ksw_extd (the contract).  The reference's SSE kernels differ from that where their
 * 16-position blocks leak across the band edge, in their anti-diagonal Z-drop, in mte_q, in tie order.  A
 * caller that must reproduce an SSE build's output bit for bit -- every ksw_extz_t field and the CIGAR -- asks for it
 *   per pair:      KSW2AMD_EZ_SSE_COMPAT or-ed into the flags of the single calls / of ksw2amd_pair_t, or
 *   process-wide:  ksw2amd_set_sse_compat(1), or KSW2AMD_SSE_COMPAT=1 in the environment of an unchanged caller.
 * Those pairs run through kernels that keep the SSE data flow (ksw2_extz2_sse.c:101-301, ksw2_extd2_sse.c:131-398; one
 * alignment per wavefront, slower than the default path).  KSW_EZ_APPROX_MAX | KSW_EZ_APPROX_DROP always takes this path:
 * that heuristic is defined by the SSE data flow (KSW2AMD_APPROX_DROP_EXACT=1 computes exactly instead). */
#define KSW2AMD_EZ_SSE_COMPAT 0x20000000
void ksw2amd_set_sse_compat(int on);

/* splice-aware batches: the arguments of ksw_exts2_sse, scoring shared by the batch */
typedef struct {
	int32_t m;
	const int8_t *mat;
	int8_t q, e, q2, noncan, junc_bonus;
} ksw2amd_splice_t;
typedef struct {
	const uint8_t *query, *target, *junc;        /* junc may be NULL */
	int32_t qlen, tlen;
	int32_t zdrop, flag;
} ksw2amd_spair_t;
int ksw2amd_exts_batch(void *km, const ksw2amd_splice_t *sc, int n, const ksw2amd_spair_t *pairs, ksw_extz_t *ez);

/* gap-linear X-drop batches: the arguments of ksw_extf2_sse, scoring shared by the batch.  ez[i] ends up exactly as after
 *   ksw_extf2_sse(km, pairs[i].qlen, pairs[i].query, pairs[i].tlen, pairs[i].target, mch, mis, e, pairs[i].w, pairs[i].xdrop, &ez[i]) */
typedef struct {
	const uint8_t *query, *target;
	int32_t qlen, tlen;
	int32_t w, xdrop;
} ksw2amd_fpair_t;
int ksw2amd_extf_batch(void *km, int8_t mch, int8_t mis, int8_t e, int n, const ksw2amd_fpair_t *pairs, ksw_extz_t *ez);
/* The same two with the sequences (and junction arrays) in DEVICE memory: pairs[].query / target / junc are device pointers -- a
 * shard of a multi-GPU job that RCCL delivered into HBM (ksw2_amd/parallel.py).  One kernel gathers them into the plan's arena;
 * nothing of them crosses the link.  (Replaces nothing in the reference: ksw2.h has no notion of a device.) */
int ksw2amd_exts_batch_device(void *km, const ksw2amd_splice_t *sc, int n, const ksw2amd_spair_t *pairs, ksw_extz_t *ez);
int ksw2amd_extf_batch_device(void *km, int8_t mch, int8_t mis, int8_t e, int n, const ksw2amd_fpair_t *pairs, ksw_extz_t *ez);

/* The same in three phases, for callers that keep batches resident in HBM (and for benchmarking the
 * device part alone): create = pack + upload, run = kernels only (asynchronous on `stream`, a hipStream_t
 * or NULL), fetch = wait + download + fill ez[]. */
typedef struct ksw2amd_plan_s ksw2amd_plan_t;
ksw2amd_plan_t *ksw2amd_plan_create(int dual, const ksw2amd_scoring_t *sc, int n, const ksw2amd_pair_t *pairs);
int  ksw2amd_plan_run(ksw2amd_plan_t *plan, void *stream);
int  ksw2amd_plan_fetch(ksw2amd_plan_t *plan, void *km, ksw_extz_t *ez);
void ksw2amd_plan_destroy(ksw2amd_plan_t *plan);
/* a resident plan from a flat batch (host or device arena); run / fetch / timing / cells / destroy as below.  The upload is complete
 * when this returns, but the plan BORROWS the arena: its bytes must stay valid and unchanged until ksw2amd_plan_destroy -- a fetch
 * reads the sequences there for =/X CIGARs (KSW_EZ_EQX) and for the pairs it runs again (wildcard codes met by a packed kernel,
 * alignments the deferred arg-max kernels handed back).  The batch entry points above borrow the arena for the duration of the call only. */
ksw2amd_plan_t *ksw2amd_plan_create_flat(int dual, const ksw2amd_scoring_t *sc, int n, const ksw2amd_flat_t *in);
/* device time of the last ksw2amd_plan_run (HIP events on its stream), fill kernels only / fill + traceback; ms */
int  ksw2amd_plan_timing(ksw2amd_plan_t *plan, float *fill_ms, float *total_ms);
/* in-band DP cells of the plan (exact band, all rows counted even if Z-drop stops early) and device bytes held */
int64_t ksw2amd_plan_cells(const ksw2amd_plan_t *plan);
int64_t ksw2amd_plan_device_bytes(const ksw2amd_plan_t *plan);
/* alignments routed to the packed-int16 kernels (two same-shape alignments per lane group) */
int64_t ksw2amd_plan_packed_pairs(const ksw2amd_plan_t *plan);
/* diagnostics: one text line per kernel class the plan's next run launches --
 *   "kernel=pk G=64 C=16 gaps=1 mode=score rebased=1 nomax=0 generic=0 form=ldscodes tasks=1536"
 * (kernel: int32 / mp / pk / pkmp / solo; form: registers / ldsrows / ldscodes / defer, the launch-time choice);
 * ksw_extf2_sse plans: "kernel=extf-lane form=ldsring ring=36 tasks=65536" (extf-lds / extf-hbm / extf-win4 / extf-win8 / extf-lane;
 * form of the lane class: hbm / ldsring with its rows).  Returns the number of lines; extz / extd / extf plans.  Tests use it to
 * assert which kernel an unforced launch took. */
int ksw2amd_plan_describe(const ksw2amd_plan_t *plan, char *buf, int cap);
/* The KSW2AMD_* environment switches (tuning, A/B runs, tests) are read once per process; this reads them again. */
void ksw2amd_reload_env(void);
/* pairs that a fetch ran a second time through the ordinary kernels since the library was loaded: flat batches' wildcard pairs, and
 * alignments in which the deferred arg-max kernels could not rule out a Z-drop without the arg-max columns */
int64_t ksw2amd_rerun_count(void);
/* Streamed plans (new; replaces nothing in the reference): a one-shape score-only batch handed to the batch entry points runs as ONE
 * plan whose sequence arena goes up in pieces while a single launch per kernel class runs under the upload, each wavefront starting its
 * task when the task's pieces have landed.  out[0] = streamed plans run since the library was loaded, out[1] = runs in which a
 * launch gave up waiting for its inputs (bounded wait) and the plan was run again behind its upload.  KSW2AMD_STREAM=0 / 1:
 * never / every plan that can. */
void ksw2amd_stream_stats(int64_t out[2]);
/* a resident plan of SSE-compatible alignments (every pair, whatever its flags); run / fetch / timing / cells / destroy as above */
ksw2amd_plan_t *ksw2amd_sse_plan_create(int dual, const ksw2amd_scoring_t *sc, int n, const ksw2amd_pair_t *pairs);
/* a resident plan of splice-aware extensions; run / fetch / timing / cells / destroy as above */
ksw2amd_plan_t *ksw2amd_exts_plan_create(const ksw2amd_splice_t *sc, int n, const ksw2amd_spair_t *pairs);
/* a resident plan of gap-linear X-drop extensions; run / fetch / timing / cells / destroy as above */
ksw2amd_plan_t *ksw2amd_extf_plan_create(int8_t mch, int8_t mis, int8_t e, int n, const ksw2amd_fpair_t *pairs);
/* raw device results without the host-side ez[] assembly: 16 int32 per pair
 * {max, zdropped, max_q, max_t, mqe, mqe_t, mte, mte_q, score, reach_end, n_cigar, rows_done, ti, tj, 0, 0} */
int  ksw2amd_plan_fetch_raw(ksw2amd_plan_t *plan, int32_t *out16);

#ifdef __cplusplus
}
#endif
#endif
