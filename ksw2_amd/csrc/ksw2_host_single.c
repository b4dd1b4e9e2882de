/*
 * ksw2_host_single.c -- the ksw2-named single-pair calls, the opt-in host path for tiny calls, coalescing of concurrent calls, the global functions.
 */
#include "ksw2_host_int.h"

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

void call_failed(const char *fn, int code, ksw_extz_t *ez)
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
	const int crowd_min = ENV(COALESCE_CROWD) ? imax(atoi(ENV(COALESCE_CROWD)), 1) : 8;      /* calls per millisecond that make a crowd (tests: 1 = always) */
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
	crowd = window_ns > 0 && (wn >= crowd_min || __atomic_load_n(&g_coal.win_prev, __ATOMIC_RELAXED) >= crowd_min);
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

