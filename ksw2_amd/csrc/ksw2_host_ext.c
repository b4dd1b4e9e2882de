/*
 * ksw2_host_ext.c -- host side of ksw_exts2_sse, ksw_extf2_sse and the SSE-compatible mode.
 */
#include "ksw2_host_int.h"

/* ---------------------------------------------------------------- splice-aware extension (ksw_exts2_sse) */

static int exts_long_thres(int q, int e, int q2)           /* ksw2_exts2_sse.c:102-104 */
{
	int lt = (q2 - q) / e - 1;
	if (q2 > q + e + lt * e) ++lt;
	return lt;
}

/* Device-resident sources (ksw2amd_exts_batch_device / ksw2amd_extf_batch_device: a shard that RCCL delivered into device memory):
 * pairs[].query / target / junc are DEVICE pointers.  The plan's arena is laid out as ever; instead of the copy into page-locked
 * staging, one kernel gathers the sequences from where they lie (k2a_gather_kernel: a table of { source, arena offset, length },
 * one workgroup per entry) behind the upload of the rest (zeros, matrices).  Nothing of the sequences crosses the link. */
static __thread int g_src_device;
typedef struct { uint64_t src; uint32_t dst, len; } gather_ent_t;      /* = K2aGather (ksw2_shim.h) */
static int gather_device_sources(ksw2amd_plan_t *p, const gather_ent_t *tab, int nent, void *up)
{
	size_t cap = 0;
	void *d_tab;
	int rc;
	if (nent == 0) return 0;
	d_tab = cache_get(BUF_POS, sizeof(gather_ent_t) * (size_t)nent, &cap);
	if (!d_tab) return -1;
	rc = k2a_shim_h2d(d_tab, tab, sizeof(gather_ent_t) * (size_t)nent, up) || k2a_shim_launch_gather((const K2aGather*)d_tab, nent, p->d_seq, up) ||
	     k2a_shim_stream_sync(up);
	cache_put(BUF_POS, d_tab, cap);
	return rc;
}

ksw2amd_plan_t *ksw2amd_exts_plan_create(const ksw2amd_splice_t *sc, int n, const ksw2amd_spair_t *pairs)
{
	ksw2amd_plan_t *p;
	const int m = sc ? sc->m : 0;
	int i, k, g, lo;
	size_t off, mat_off, up_bytes = 0, cst_words = 0;
	void *up;
	uint32_t fill[3][2][3];
	int wn, ngt = 0;
	gather_ent_t *gt = 0;

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
	if (g_src_device) {
		memset(p->h_seq, 0, up_bytes);
		gt = (gather_ent_t*)malloc(sizeof(gather_ent_t) * (3 * (size_t)n + 1));
		if (!gt) { fail(KSW2AMD_E_NOMEM, "exts: host allocation failed%s", 0); goto err; }
	}
	for (k = 0, i = 0; i < 3; ++i)
		for (g = 0; g < 2; ++g)
			for (wn = 0; wn < 3; ++wn) { p->s_first[i][g][wn] = k; fill[i][g][wn] = (uint32_t)k; k += p->s_count[i][g][wn]; }
	p->ntasks = p->norder = k;
	for (i = 0; i < n; ++i) {
		const ksw2amd_spair_t *a = &pairs[i];
		if (p->h_cls[i] < 0) continue;
		if (g_src_device) {
			gt[ngt].src = (uint64_t)(uintptr_t)a->query; gt[ngt].dst = p->h_pairs[i].qoff; gt[ngt++].len = (uint32_t)a->qlen;
			gt[ngt].src = (uint64_t)(uintptr_t)a->target; gt[ngt].dst = p->h_pairs[i].toff; gt[ngt++].len = (uint32_t)a->tlen;
			if (a->junc) { gt[ngt].src = (uint64_t)(uintptr_t)a->junc; gt[ngt].dst = p->h_pairs[i].toff + (uint32_t)align_up((size_t)a->tlen, 4); gt[ngt++].len = (uint32_t)a->tlen; }
		} else {
		memcpy(p->h_seq + p->h_pairs[i].qoff, a->query, (size_t)a->qlen);
		memcpy(p->h_seq + p->h_pairs[i].toff, a->target, (size_t)a->tlen);
		if (a->junc) memcpy(p->h_seq + p->h_pairs[i].toff + align_up((size_t)a->tlen, 4), a->junc, (size_t)a->tlen);
		}
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
	    (gt && gather_device_sources(p, gt, ngt, up)) ||
	    k2a_shim_launch_splice_const(p->d_pairs, n, p->d_seq, sc->noncan, sc->junc_bonus, up) || k2a_shim_stream_sync(up)) {
		fail(KSW2AMD_E_NODEVICE, "upload failed: %s", k2a_shim_last_error());
		goto err;
	}
	for (g = 0; g < 2; ++g) {
		p->s_par[g].q = sc->q; p->s_par[g].e = sc->e; p->s_par[g].q2 = sc->q2; p->s_par[g].m = m;
		p->s_par[g].long_thres = exts_long_thres(sc->q, sc->e, sc->q2);
		p->s_par[g].mat = (const int8_t*)p->d_seq + mat_off + (g ? (size_t)m * m : 0);
	}
	free(gt);
	plan_ready(p);                                  /* uploads complete */
	return p;
err:
	free(gt);
	ksw2amd_plan_destroy(p);
	return 0;
}

int exts_plan_run(ksw2amd_plan_t *p, void *stream)
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
static int wave_chunks(int n, int workers, int uniform, int per_wave, int *chunk_pairs)      /* per_wave: alignments one wavefront of the batch's kernel holds */
{
	const int simds = k2a_shim_simd_count() * imax(per_wave, 1);
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

typedef struct { void *km; const ksw2amd_splice_t *sc; const ksw2amd_spair_t *pairs; ksw_extz_t *ez; int src_device; } exts_ctx_t;
int exts_chunk(void *ctx_, int beg, int end, int share, pend_t *pd)
{
	exts_ctx_t *c = (exts_ctx_t*)ctx_;
	(void)pd;
	if (beg < 0) return KSW2AMD_OK;
	{
		int rc;
		g_src_device = c->src_device;
		rc = exts_serial(c->km, c->sc, end - beg, c->pairs + beg, c->ez + beg, share);
		g_src_device = 0;
		return rc;
	}
}

int ksw2amd_exts_batch(void *km, const ksw2amd_splice_t *sc, int n, const ksw2amd_spair_t *pairs, ksw_extz_t *ez)
{
	const int tpd = pool_threads_per_device();
	if (n >= (pool_min_pairs() ? pool_min_pairs() : 2048) && tpd > 0 && !g_is_worker && k2a_shim_device_count() > 0) {
		const int workers = tpd * (g_ndev_set > 0 ? g_ndev_set : 1);
		double *cost = (double*)malloc(sizeof(double) * (size_t)n), total = 0;
		int i, rc = 0, uniform = 1, chunk_pairs = 0, nchunks;
		for (i = 1; i < n && uniform; ++i) uniform = pairs[i].qlen == pairs[0].qlen && pairs[i].tlen == pairs[0].tlen;
		nchunks = wave_chunks(n, workers, uniform, 1, &chunk_pairs);
		if (cost && nchunks >= 2) {
			exts_ctx_t ctx;
			for (i = 0; i < n; ++i) { cost[i] = 1.0 + (double)imax(pairs[i].qlen, 0) * imax(pairs[i].tlen, 0); total += cost[i]; }
			ctx.km = km; ctx.sc = sc; ctx.pairs = pairs; ctx.ez = ez; ctx.src_device = g_src_device;
			if (run_pooled(exts_chunk, &ctx, n, cost, total, nchunks, chunk_pairs, &rc)) { free(cost); return rc; }
		}
		free(cost);
	}
	return exts_serial(km, sc, n, pairs, ez, 1);
}

int ksw2amd_exts_batch_device(void *km, const ksw2amd_splice_t *sc, int n, const ksw2amd_spair_t *pairs, ksw_extz_t *ez)
{
	int rc;
	g_src_device = 1;
	rc = ksw2amd_exts_batch(km, sc, n, pairs, ez);
	g_src_device = 0;
	return rc;
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
#define EXTFB_SPAN 160              /* = K2A_EXTFB_SPAN(16 / 32 / 64) (ksw2_lane_extfb.h) */
#define EXTFB_SPAN32 416
#define EXTFB_SPAN64 928

ksw2amd_plan_t *ksw2amd_extf_plan_create(int8_t mch, int8_t mis, int8_t e, int n, const ksw2amd_fpair_t *pairs)
{
	ksw2amd_plan_t *p;
	int i, c, span, nlane = 0, use_lane, ngt = 0;
	size_t off = 0;
	uint32_t fill[10];
	void *up;
	sort_t *srt = 0;
	gather_ent_t *gt = 0;

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
		if (g_src_device) use_lane = 0;                                      /* (that form's arena is interleaved by the host) */
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
		/* narrow bands: four extensions per wavefront (k2a_extf_grp_kernel; KSW2AMD_EXTF_GRP=0 never, =1 wherever the band fits) */
		{
			const char *gv = ENV(EXTF_GRP);
			const int forced = ENV(EXTF_LDS) || ENV(EXTF_WIN) || ENV(EXTF_HBM);
			if (span <= EXTFB_SPAN && (gv && *gv ? atoi(gv) != 0 : !forced)) c = 7;
			/* ... two / one extension per wavefront for wider bands (G = 32 / 64 lanes each), in front of the register windows and the
			 * LDS forms (KSW2AMD_EXTF_GRP=1: only the four-per-wavefront form) */
			else if (span <= EXTFB_SPAN64 && (gv && *gv ? atoi(gv) >= 2 : !forced)) c = span <= EXTFB_SPAN32 ? 8 : 9;
		}
		p->cells += band_cells(a->qlen, a->tlen, d->w);
		if (use_lane) { p->h_cls[i] = 6; ++p->f_count[6]; ++nlane; continue; }          /* sequences and state: grouped below */
		if (c == 3) { d->tb_off = p->tb_bytes; p->tb_bytes += align_up(3 * align_up((size_t)a->tlen, 16), 256); }
		p->h_cls[i] = (int8_t)c; ++p->f_count[c];
		off = align_up(off, 4); d->qoff = (uint32_t)off; off += (size_t)a->qlen;
		off = align_up(off, 4); d->toff = (uint32_t)off; off += (size_t)a->tlen;
		if (off > 0xfff00000u) { fail(KSW2AMD_E_PARAM, "extf: more than 4 GiB of sequence in one plan%s", 0); goto err; }
	}
	for (c = 0, i = 0; c < 10; ++c) { p->f_first[c] = i; fill[c] = (uint32_t)i; i += p->f_count[c]; }
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
	if (g_src_device) {
		memset(p->h_seq, 0, p->seq_bytes);
		gt = (gather_ent_t*)malloc(sizeof(gather_ent_t) * (2 * (size_t)n + 1));
		if (!gt) { fail(KSW2AMD_E_NOMEM, "extf: host allocation failed%s", 0); goto err; }
	}
	for (i = 0; i < n; ++i) {
		if (p->h_cls[i] < 0 || p->h_cls[i] == 6) continue;
		if (gt) {
			gt[ngt].src = (uint64_t)(uintptr_t)pairs[i].query; gt[ngt].dst = p->h_pairs[i].qoff; gt[ngt++].len = (uint32_t)pairs[i].qlen;
			gt[ngt].src = (uint64_t)(uintptr_t)pairs[i].target; gt[ngt].dst = p->h_pairs[i].toff; gt[ngt++].len = (uint32_t)pairs[i].tlen;
		} else {
		memcpy(p->h_seq + p->h_pairs[i].qoff, pairs[i].query, (size_t)pairs[i].qlen);
		memcpy(p->h_seq + p->h_pairs[i].toff, pairs[i].target, (size_t)pairs[i].tlen);
		}
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
	    k2a_shim_memset(p->d_res, 0, sizeof(K2aResult) * (size_t)n, up) || (gt && gather_device_sources(p, gt, ngt, up)) || k2a_shim_stream_sync(up)) {
		fail(KSW2AMD_E_NODEVICE, "upload failed: %s", k2a_shim_last_error());
		goto err;
	}
	free(gt);
	plan_ready(p);                                  /* uploads complete */
	return p;
err:
	free(srt); free(gt);
	ksw2amd_plan_destroy(p);
	return 0;
}

int extf_plan_run(ksw2amd_plan_t *p, void *stream)
{
	int c;
	p->stream = stream; p->ran = 1; p->stream_used = 1;
	if (p->ntasks == 0) return KSW2AMD_OK;
	if (k2a_shim_event_record(p->ev[0], stream)) goto err;
	if (p->f_count[6] && k2a_shim_memset(p->d_tb, 0, p->f_state_bytes, stream)) goto err;     /* the reference's zeroed arrays (ksw2_extf2_sse.c:25) */
	for (c = 9; c >= 0; --c)
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
int wants_ssec(int flag)
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
#define SSECB_SPAN 960                /* = K2A_SSECB_SPAN (ksw2_lane_ssecb.h) */
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
	uint32_t fill[9];

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
		{	/* kernel form.  2: state in registers (k2a_ssec_blk_kernel: simple scoring over at most five codes, bands up to SSECB_SPAN positions;
			 * KSW2AMD_SSEC_BLK=0: never); 1: state arrays of up to SSEC_LDS_MAX bytes in LDS (k2a_ssec_kernel<.., LDS = true>);
			 * 0: in HBM scratch (KSW2AMD_SSEC_HBM=1: always, tests) */
			const size_t sb = (size_t)(dual ? 11 : 9) * (size_t)((a->tlen + 15) / 16 * 16);
			const char *blk = ENV(SSEC_BLK);
			int form = sb <= SSEC_LDS_MAX;
			if (!(fl & KSW_EZ_GENERIC_SC) && m <= 5 && imin(imin(a->qlen, a->tlen), w + 1) <= SSECB_SPAN && !(blk && blk[0] == '0')) form = 2;      /* (m <= 5: the register form's score profiles hold query codes 0..3 + the wildcard) */
			if (ENV(SSEC_HBM)) form = 0;
			p->h_cls[i] = (int8_t)(mode + 3 * form);
			++p->s_count[mode][0][form];
			if (form == 1 && sb > p->c_lds[mode]) p->c_lds[mode] = sb;
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
	for (k = 0, mode = 0; mode < 9; ++mode) { p->s_first[mode % 3][0][mode / 3] = k; fill[mode] = (uint32_t)k; k += p->s_count[mode % 3][0][mode / 3]; }
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

int ssec_plan_run(ksw2amd_plan_t *p, void *stream)
{
	int mode;
	p->stream = stream; p->ran = 1; p->stream_used = 1;
	if (p->up_ev && k2a_shim_stream_wait_event(stream, p->up_ev)) goto err;      /* the plan's upload (shared stream) before its kernels */
	if (p->reject_all || p->ntasks == 0) return KSW2AMD_OK;
	if (k2a_shim_event_record(p->ev[0], stream)) goto err;
	for (mode = 0; mode < 3; ++mode)
		if (p->s_count[mode][0][2] &&
		    k2a_shim_launch_ssec_blk(p->dual, mode, &p->c_par, p->d_pairs, p->d_order + p->s_first[mode][0][2], p->s_count[mode][0][2], p->d_seq, p->d_tb, p->d_res, stream)) goto err;
	for (mode = 0; mode < 6; ++mode)
		if (p->s_count[mode % 3][0][mode / 3] &&
		    k2a_shim_launch_ssec(p->dual, mode % 3, mode / 3 ? p->c_lds[mode % 3] : 0, &p->c_par, p->d_pairs, p->d_order + p->s_first[mode % 3][0][mode / 3],
		                         p->s_count[mode % 3][0][mode / 3], p->d_seq, p->d_tb, (uint8_t*)p->d_bnd, p->d_res, stream)) goto err;
	if (k2a_shim_event_record(p->ev[1], stream)) goto err;
	for (mode = 0; mode < 9; ++mode)
		if (mode % 3 && p->s_count[mode % 3][0][mode / 3] &&
		    k2a_shim_launch_ssec_trace(p->d_pairs, p->d_order + p->s_first[mode % 3][0][mode / 3], p->s_count[mode % 3][0][mode / 3], p->d_tb, p->d_res, p->d_cig, stream)) goto err;
	if (k2a_shim_event_record(p->ev[2], stream)) goto err;
	return KSW2AMD_OK;
err:
	return fail(KSW2AMD_E_NODEVICE, "sse-compatible run: %s", k2a_shim_last_error());
}

/* n pairs through SSE-compatible plans sized to the device's free memory */
int ssec_run(int dual, void *km, const ksw2amd_scoring_t *sc, int n, const ksw2amd_pair_t *pairs, ksw_extz_t *ez)
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

typedef struct { void *km; int8_t mch, mis, e; const ksw2amd_fpair_t *pairs; ksw_extz_t *ez; int src_device; } extf_ctx_t;
int extf_chunk(void *ctx_, int beg, int end, int share, pend_t *pd)
{
	extf_ctx_t *c = (extf_ctx_t*)ctx_;
	(void)pd; (void)share;
	if (beg < 0) return KSW2AMD_OK;
	{
		int rc;
		g_src_device = c->src_device;
		rc = extf_serial(c->km, c->mch, c->mis, c->e, end - beg, c->pairs + beg, c->ez + beg);
		g_src_device = 0;
		return rc;
	}
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
		{	/* narrow bands run four extensions per wavefront (k2a_extf_grp_kernel): a chunk of whole device fills is four times the pairs
			 * (16 384 x 1 000^2, band 100: four chunks of 4 096 run at 680-715 GCUPS end to end, six of 3 072 at 460-490) */
			const int wq = pairs[0].w < 0 ? imax(pairs[0].qlen, pairs[0].tlen) : pairs[0].w;
			const int span0 = imin(imin(pairs[0].qlen, pairs[0].tlen), wq < 0x7ffffff0 ? wq + 1 : wq);
			const char *gv = ENV(EXTF_GRP);
			const int grp = uniform && !(gv && *gv && atoi(gv) == 0) && !(ENV(EXTF_LDS) || ENV(EXTF_WIN) || ENV(EXTF_HBM));
			const int wide = !(gv && *gv && atoi(gv) == 1);
			nchunks = wave_chunks(n, workers, uniform, !grp ? 1 : span0 <= EXTFB_SPAN ? 4 : (wide && span0 <= EXTFB_SPAN32) ? 2 : 1, &chunk_pairs);
		}
		if (cost && nchunks >= 2) {
			extf_ctx_t ctx;
			for (i = 0; i < n; ++i) { cost[i] = 1.0 + (double)imax(pairs[i].qlen, 0) + imax(pairs[i].tlen, 0); total += cost[i]; }
			ctx.km = km; ctx.mch = mch; ctx.mis = mis; ctx.e = e; ctx.pairs = pairs; ctx.ez = ez; ctx.src_device = g_src_device;
			if (run_pooled(extf_chunk, &ctx, n, cost, total, nchunks, chunk_pairs, &rc)) { free(cost); return rc; }
		}
		free(cost);
	}
	return extf_serial(km, mch, mis, e, n, pairs, ez);
}

int ksw2amd_extf_batch_device(void *km, int8_t mch, int8_t mis, int8_t e, int n, const ksw2amd_fpair_t *pairs, ksw_extz_t *ez)
{
	int rc;
	g_src_device = 1;
	rc = ksw2amd_extf_batch(km, mch, mis, e, n, pairs, ez);
	g_src_device = 0;
	return rc;
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

