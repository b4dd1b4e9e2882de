/*
 * ksw2_shim_sim.cpp -- TEST INFRASTRUCTURE: a host-memory implementation of ksw2_shim.h that runs the
 * very same per-lane code (ksw2_amd/csrc/ksw2_lane.h) for 64 "lanes" in lock step, mirroring the control
 * flow of k2a_fill_kernel in ksw2_shim_hip.hip line by line.  Linked with ksw2_host.c into
 * tests/sim/libksw2_amd_sim.so so the packing / geometry / scheduling / bookkeeping logic can be checked
 * against the oracle in the CPU test tier.  It is never part of the product library libksw2_amd.so.
 */
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include "../../ksw2_amd/csrc/ksw2_shim.h"
#include "../../ksw2_amd/csrc/ksw2_lane.h"

static char g_err[256] = "";

template<int G, int C, bool DUAL, int MODE>
static void sim_fill(const K2aScoring sc, const K2aPair *pairs, const uint32_t *order, int ntasks, const uint8_t *seq, uint8_t *tb,
                     K2aResult *res)
{
	constexpr int NG = 64 / G;
	typedef K2aLane<G, C, DUAL, MODE> Lane;
	const int nwaves = (ntasks + NG - 1) / NG;
	for (int wv = 0; wv < nwaves; ++wv) {
		static Lane L[64];
		K2aBook book[NG];
		K2aPair pr[64];
		uint32_t pi[64];
		bool valid[64], gdone[64];
		int klast[64], kmax = -1;
		uint8_t *tbp[64];
		for (int lane = 0; lane < 64; ++lane) {
			const int grp = lane / G, gl = lane % G, task = wv * NG + grp;
			valid[lane] = task < ntasks;
			pi[lane] = order[valid[lane] ? task : 0];
			pr[lane] = pairs[pi[lane]];
			L[lane].setup(pr[lane], seq, gl, valid[lane]);
			if (gl == 0) k2a_book_reset(&book[grp]);
			klast[lane] = L[lane].last_step();
			if (klast[lane] > kmax) kmax = klast[lane];
			tbp[lane] = tb + pr[lane].tb_off + (size_t)gl * (Lane::TBWORDS * 4);
			gdone[lane] = !valid[lane];
			L[lane].qb = L[lane].next_query_code(-1);
		}
		for (int k = 0; k <= kmax; ++k) {
			int hin[64], ein[64], e2in[64], qnext[64];
			for (int lane = 0; lane < 64; ++lane) {          /* DPP rotate inside each group */
				const int grp = lane / G, gl = lane % G, src = grp * G + (gl + G - 1) % G;
				hin[lane] = L[src].hout; ein[lane] = L[src].eout; e2in[lane] = DUAL ? L[src].e2out : 0;
			}
			bool anyfin = false;
			bool nfin[64];
			for (int lane = 0; lane < 64; ++lane) {
				if (L[lane].need_init(k)) L[lane].do_init(sc);
				qnext[lane] = L[lane].next_query_code(k);
				uint32_t tw[Lane::TBWORDS];
				const bool live = L[lane].step(sc, k, hin[lane], ein[lane], e2in[lane], tw);
				if (MODE != K2A_MODE_SCORE && live)
					memcpy(tbp[lane] + (size_t)k * (G * Lane::TBWORDS * 4), tw, sizeof(tw));
				nfin[lane] = L[lane].need_fin(k);
				anyfin |= nfin[lane];
			}
			if (anyfin) {
				for (int lane = 0; lane < 64; ++lane)
					if (nfin[lane]) L[lane].do_fin(sc, &book[lane / G], pr[lane].zdrop);
				for (int lane = 0; lane < 64; ++lane)
					if (book[lane / G].dropped) gdone[lane] = true;
			}
			bool all_done = true;
			for (int lane = 0; lane < 64; ++lane) {
				L[lane].qb = qnext[lane];
				if (!(gdone[lane] || k >= klast[lane])) all_done = false;
			}
			if (all_done) break;
		}
		for (int lane = 0; lane < 64; ++lane)
			if (valid[lane] && lane % G == 0) k2a_finish(pr[lane], book[lane / G], &res[pi[lane]]);
	}
}

template<int G, int C, bool DUAL>
static void sim_trace(const K2aPair *pairs, const uint32_t *order, int ntasks, const uint8_t *tb, K2aResult *res, uint32_t *cig)
{
	for (int t = 0; t < ntasks; ++t) {
		const uint32_t pi = order[t];
		const K2aPair pr = pairs[pi];
		int n = 0;
		if (res[pi].ti >= 0 && res[pi].tj >= 0) n = k2a_trace_pair<G, C, DUAL>(tb + pr.tb_off, res[pi].ti, res[pi].tj, cig + pr.cig_off);
		res[pi].n_cigar = n;
	}
}

typedef void (*fill_fn)(const K2aScoring, const K2aPair*, const uint32_t*, int, const uint8_t*, uint8_t*, K2aResult*);
typedef void (*trace_fn)(const K2aPair*, const uint32_t*, int, const uint8_t*, K2aResult*, uint32_t*);
#define FILL_ROW(G, C) { { sim_fill<G, C, false, 0>, sim_fill<G, C, false, 1>, sim_fill<G, C, false, 2> }, \
                         { sim_fill<G, C, true, 0>,  sim_fill<G, C, true, 1>,  sim_fill<G, C, true, 2> } }
static const fill_fn g_fill[K2A_NCFG][2][3] = { FILL_ROW(16, 8), FILL_ROW(64, 8), FILL_ROW(64, 16), FILL_ROW(64, 32) };
#define TRACE_ROW(G, C) { sim_trace<G, C, false>, sim_trace<G, C, true> }
static const trace_fn g_trace[K2A_NCFG][2] = { TRACE_ROW(16, 8), TRACE_ROW(64, 8), TRACE_ROW(64, 16), TRACE_ROW(64, 32) };

extern "C" {

const char *k2a_shim_backend(void) { return "sim"; }
const char *k2a_shim_last_error(void) { return g_err; }
int k2a_shim_device_count(void) { return 1; }
int k2a_shim_set_device(int dev) { return dev == 0 ? 0 : -1; }
int k2a_shim_mem_info(size_t *free_b, size_t *total_b) { *free_b = (size_t)8 << 30; *total_b = (size_t)8 << 30; return 0; }
void *k2a_shim_malloc(size_t bytes) { return calloc(bytes ? bytes : 16, 1); }
void k2a_shim_free(void *p) { free(p); }
void *k2a_shim_host_malloc(size_t bytes) { return malloc(bytes ? bytes : 16); }
void k2a_shim_host_free(void *p) { free(p); }
int k2a_shim_h2d(void *dst, const void *src, size_t bytes, void *) { memcpy(dst, src, bytes); return 0; }
int k2a_shim_d2h(void *dst, const void *src, size_t bytes, void *) { memcpy(dst, src, bytes); return 0; }
int k2a_shim_memset(void *dst, int v, size_t bytes, void *) { memset(dst, v, bytes); return 0; }
void *k2a_shim_stream_create(void) { return (void*)1; }
void k2a_shim_stream_destroy(void *) {}
int k2a_shim_stream_sync(void *) { return 0; }
void *k2a_shim_event_create(void) { return calloc(1, sizeof(double)); }
void k2a_shim_event_destroy(void *ev) { free(ev); }
int k2a_shim_event_record(void *ev, void *)
{
	*(double*)ev = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
	return 0;
}
float k2a_shim_event_ms(void *a, void *b) { return (float)(*(double*)b - *(double*)a); }

int k2a_shim_launch_fill(int cfg, int dual, int mode, const K2aScoring *sc, const K2aPair *pairs, const uint32_t *order,
                         int ntasks, const uint8_t *seq, uint8_t *tb, K2aResult *res, void *)
{
	if (ntasks > 0) g_fill[cfg][dual ? 1 : 0][mode](*sc, pairs, order, ntasks, seq, tb, res);
	return 0;
}
int k2a_shim_launch_trace(int cfg, int dual, const K2aPair *pairs, const uint32_t *order, int ntasks, const uint8_t *tb,
                          K2aResult *res, uint32_t *cig, void *)
{
	if (ntasks > 0) g_trace[cfg][dual ? 1 : 0](pairs, order, ntasks, tb, res, cig);
	return 0;
}
int k2a_shim_launch_compact(const K2aPair *pairs, const K2aResult *res, const uint32_t *pos, int n, const uint32_t *cig,
                            uint32_t *pool, void *)
{
	for (int i = 0; i < n; ++i)
		for (int k = 0; k < res[i].n_cigar; ++k) pool[pos[i] + k] = cig[pairs[i].cig_off + k];
	return 0;
}

}
