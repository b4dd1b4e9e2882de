/*
 * ksw2_shim.h -- the thin C-ABI between the C host (ksw2_host_*.c) and the device runtime.
 *
 * The product library links ksw2_shim_hip.hip (HIP runtime + gfx950 kernels).  tests/sim/ links the very
 * same host code against a host-memory, lock-step wave simulator of the same interface so that the
 * packing / scheduling logic can be checked without a GPU; that simulator is test infrastructure and is
 * never part of libksw2_amd.so.
 */
#ifndef KSW2_SHIM_H_
#define KSW2_SHIM_H_

#include <stddef.h>
#include "ksw2_types.h"

#ifdef __cplusplus
extern "C" {
#endif

/* kernel geometry classes: lanes per alignment (G) x target rows per lane (C) */
#define K2A_NCFG 5
#define K2A_CFG_MP 4      /* generation-serial class: any band, boundary rows through HBM */
static const int k2a_cfg_G[K2A_NCFG] = { 16, 64, 64, 64, 64 };
static const int k2a_cfg_C[K2A_NCFG] = {  8,  8, 16, 32, 16 };
/* geometry classes of the packed-int16 kernels (two same-shape alignments per lane group) */
#ifndef K2A_PKMP_WAVES
#define K2A_PKMP_WAVES 4      /* packed generation-serial class: wavefronts (generations in flight) per task */
#endif
#define K2A_NPKCFG 5
#define K2A_PKCFG_MP 4    /* packed generation-serial class (ksw2_lane_pkmp.h): any band, a workgroup of 4 wavefronts per task */
static const int k2a_pkcfg_G[K2A_NPKCFG] = {  8, 16, 64, 64, 64 };
static const int k2a_pkcfg_C[K2A_NPKCFG] = { 18,  8,  8, 16, 16 };

const char *k2a_shim_backend(void);                /* "hip:gfx950" or "sim" */
const char *k2a_shim_last_error(void);
int   k2a_shim_async_launches(void);         /* 1: launches run on the device behind the call (hip); 0: inside the call (the simulator): what a
                                              * launch waits for on the device must then be complete before the call */
int   k2a_shim_device_count(void);
int   k2a_shim_simd_count(void);               /* SIMDs (wavefront slots side by side) of the current device; 0 = unknown */
int   k2a_shim_set_device(int dev);
int   k2a_shim_pci_bus_id(char *buf, int cap);  /* the current device's directory name under /sys/bus/pci/devices; -1 = unknown */
int   k2a_shim_get_device(void);             /* device of the calling thread; -1 = none */
int   k2a_shim_mem_info(size_t *free_b, size_t *total_b);

/* launch-time kernel forms: every choice the launcher makes can be forced (-1 = automatic, 0 / 1) and is reported */
#define K2A_OPT_LDSCODES 0     /* exact score-only packed (64,16) kernels: target-code planes in LDS */
#define K2A_OPT_LDSROWS  1     /* two-piece traceback packed (64,16) and int32 generation-serial traceback kernels: row state in LDS */
#define K2A_NOPT 2
void  k2a_shim_set_option(int opt, int value);
int   k2a_shim_pk_form(int cfg, int dual, int mode, int nomax, int ntasks);   /* what k2a_shim_launch_fill_pk takes: 0 registers, 1 row state in LDS, 2 code planes in LDS */
int   k2a_shim_mp_form(int dual, int mode, int ntasks);                        /* k2a_shim_launch_fill, class K2A_CFG_MP: 1 = row state in LDS */

void *k2a_shim_malloc(size_t bytes);               /* device memory */
void  k2a_shim_free(void *p);
void *k2a_shim_host_malloc(size_t bytes);          /* pinned host staging */
void  k2a_shim_host_free(void *p);
int   k2a_shim_h2d(void *dst, const void *src, size_t bytes, void *stream);
int   k2a_shim_d2h(void *dst, const void *src, size_t bytes, void *stream);
int   k2a_shim_d2d(void *dst, const void *src, size_t bytes, void *stream);
int   k2a_shim_host_register(void *p, size_t bytes);   /* page-lock caller memory so that uploads from it are asynchronous and at link rate */
int   k2a_shim_host_unregister(void *p);
int   k2a_shim_memset(void *dst, int v, size_t bytes, void *stream);

void *k2a_shim_stream_create(void);
void *k2a_shim_stream_create_high(void);      /* highest priority: hardware queues apart from the ordinary streams' */
void *k2a_shim_stream_create_low(void);       /* lowest priority: the upload streams (copies only), apart from every stream that runs kernels */
void  k2a_shim_stream_destroy(void *stream);
int   k2a_shim_stream_sync(void *stream);
void *k2a_shim_event_create(void);
void  k2a_shim_event_destroy(void *ev);
int   k2a_shim_event_record(void *ev, void *stream);
int   k2a_shim_stream_wait_event(void *stream, void *ev);   /* later work on `stream` waits for `ev` */
float k2a_shim_event_ms(void *start, void *stop);  /* blocks on `stop` */
int   k2a_shim_event_sync(void *ev);               /* blocks until the work recorded before `ev` is done */

/*
 * Fill kernel: ntasks alignments (order[t] = index into pairs/res) on geometry class `cfg`.
 *   dual: 0 single affine gap (extz2 / gg2), 1 two-piece (extd2);  mode: K2A_MODE_*
 * Trace kernel: walks the traceback blocks written by a fill launch of the same class and writes each
 * CIGAR (end -> start order) into cig[pairs[i].cig_off ...], count in res[i].n_cigar.
 */
int k2a_shim_launch_fill(int cfg, int dual, int mode, const K2aScoring *sc, const K2aPair *pairs, const uint32_t *order,
                         int ntasks, const uint8_t *seq, uint8_t *tb, int32_t *bnd, K2aResult *res, void *stream);
int k2a_shim_launch_trace(int cfg, int dual, const K2aPair *pairs, const uint32_t *order, int ntasks, const uint8_t *tb,
                          K2aResult *res, uint32_t *cig, void *stream);

/* Packed-int16 fill: ntasks tasks of TWO same-shape alignments each, order2[2t], order2[2t+1] = their indices (equal
 * for an unpaired leftover); both K2aPair entries point at the task's byte-interleaved sequences and, with
 * mode != SCORE, at the task's shared traceback block (2*C bytes per lane-step).  cfg indexes the k2a_pkcfg_* table.
 * rebased: per-strip score bases (reads of any length whose band window fits 16 bits).
 * nomax: final score and direction bytes only (KSW_EZ_APPROX_MAX launches).
 * The packed trace kernel walks 2*ntasks alignments of such a launch. */
/* defer (exact score-only single-gap classes): the fill tracks row maxima without their columns and streams a checkpoint of every
 * lane's top inputs into `tb`; a second kernel of the same launch call re-runs the strips whose columns the results need
 * (K2aLanePk, DEFER).  Every pair of a wavefront carries the wavefront's checkpoint block in tb_off (byte offset in tb), its
 * stream length in steps in bnd_off and the strips-per-group stride of its header table in cig_off:
 *   block = [bnd_off steps][64 lanes] x 8 bytes, then [64 / G groups][cig_off strips] x 16 bytes (K2aCkHead). */
/* qd != NULL (device pointer, K2aQueueDesc in ksw2_types.h): a streamed launch -- the ordinary full grid, every wavefront's task given
 * by its position in the grid, started under the upload: a wavefront first waits (k2a_queue_wait) until the watermark says that the
 * qd->need[] pieces its task lies in have landed, or gives up after the timeout and sets qd->abort; qd->next only counts the
 * wavefront-tasks that started (the fetch compares it with the task count).  Because the grid can fill the device with waiting
 * wavefronts, nothing the pieces depend on may be a kernel queued behind it.  Score-only classes only (mode == K2A_MODE_SCORE):
 * those are what the batch entry points stream. */
int k2a_shim_launch_fill_pk(int cfg, int dual, int mode, int rebased, int nomax, int defer, const K2aScoring *sc, const K2aPair *pairs, const uint32_t *order2,
                            int ntasks, const uint8_t *seq, uint8_t *tb, K2aResult *res, K2aQueueDesc *qd, void *stream);
int k2a_shim_launch_trace_pk(int cfg, int dual, const K2aPair *pairs, const uint32_t *order2, int ntasks, const uint8_t *tb,
                             K2aResult *res, uint32_t *cig, void *stream);

/* uniform plans (K2aUniform, ksw2_types.h): write pairs[n], order2[2 * ntasks] and -- need != NULL -- the streamed launch's
 * need[ceil(ntasks / ng)] on the device */
/* device-resident sources of the exts / extf batch entries: entry k copies len bytes from the device address src to dst + dst_off */
typedef struct K2aGather { uint64_t src; uint32_t dst, len; } K2aGather;
int k2a_shim_launch_gather(const K2aGather *tab, int n, uint8_t *dst, void *stream);
/* 4-bit wire format of uniform plans: bytes / 2 upload bytes at src -> bytes arena bytes at dst (bytes a multiple of 8) */
int k2a_shim_launch_wire_expand(const uint8_t *src, uint8_t *dst, size_t bytes, int fmt, uint32_t stride, void *stream);      /* fmt 1: four bits per code, 2: two bits + escapes */
int k2a_shim_launch_uniform_layout(const K2aUniform *u, K2aPair *pairs, uint32_t *order2, uint32_t *need, void *stream);

/* Packed generation-serial fill (class K2A_PKCFG_MP): one task (two same-shape alignments) per workgroup of four wavefronts that
 * pipeline the task's generations of 1024 target rows.  Both K2aPair entries of a task share bnd_off: K2A_PKMP_BND_WORDS
 * (ksw2_lane_pkmp.h) uint32 of boundary entries followed by K2A_PKMP_WAVES x K2A_PKMP_SPILL_WORDS(16) of row-maximum keys, 16-byte aligned;
 * traceback block as for the int32 generation-serial class but 32 bytes per lane-step.  The packed trace launch with
 * cfg = K2A_PKCFG_MP walks it. */
int k2a_shim_launch_fill_pkmp(int dual, int mode, const K2aScoring *sc, const K2aPair *pairs, const uint32_t *order2, int ntasks,
                              const uint8_t *seq, uint8_t *tb, uint32_t *bnd, K2aResult *res, void *stream);

/* Solo packed fill (ksw2_lane_solo.h): ONE alignment per wavefront using both 16-bit halves (rows [i0,i0+C) and [i0+C,i0+2C) of
 * a double strip), for alignments without a partner of identical shape.  K2A_SOLO_C rows per half; order[t] = pair index. */
#define K2A_SOLO_C 8       /* tasks with a traceback */
/* score-only tasks: 16 rows per half fit the registers (208 VGPRs) and halve the per-step overhead per cell, but 64 lanes x 32
 * rows = 2 048 rows in flight want a band that wide -- at w = 500 half the lanes idle (measured, 1 024 x 10 k x 10 k: 9.5 ms
 * against 7.1 ms with 8 rows, profiles/r3_solo_experiments.txt).  The steps of a task are its columns whatever the height. */
#define K2A_SOLO_CS 8
#define K2A_SOLO_ROWS(score_only) ((score_only) ? K2A_SOLO_CS : K2A_SOLO_C)
int k2a_shim_launch_fill_solo(int dual, int mode, const K2aScoring *sc, const K2aPair *pairs, const uint32_t *order, int ntasks,
                              const uint8_t *seq, uint8_t *tb, K2aResult *res, void *stream);
int k2a_shim_launch_trace_solo(const K2aPair *pairs, const uint32_t *order, int ntasks, const uint8_t *tb, K2aResult *res,
                               uint32_t *cig, void *stream);

/* Splice-aware extension (ksw2_lane_dm.h): one alignment per wavefront, diagonal-major.  pairs[i].bnd_off = dword offset
 * (in seq) of the alignment's packed per-target-position constants, tb_off = its direction bytes ((qlen+tlen-1) rows of
 * min(qlen,tlen) bytes), w / end_bonus set so that k2a_finish applies the plain start-cell rule.  The trace launch walks
 * them with the intron state.  order[t] = index into pairs / res, as for the fill kernels.
 * win = register window class 0 / 1: diagonals up to K2A_DM_DIAG(K2A_DM_SLOTS_S / K2A_DM_SLOTS) cells; win = 2: any length,
 * state in `scratch` (9 * tlen ints per alignment at 4 * pairs[i].pad). */
int k2a_shim_launch_splice_const(const K2aPair *pairs, int n, uint8_t *seq, int noncan, int junc_bonus, void *stream);   /* per-position constants from the uploaded targets */
int k2a_shim_launch_exts(int mode, int win, const K2aSplice *sp, const K2aPair *pairs, const uint32_t *order, int ntasks, const uint8_t *seq,
                         uint8_t *tb, int32_t *scratch, K2aResult *res, void *stream);
int k2a_shim_launch_exts_trace(const K2aSplice *sp, const K2aPair *pairs, const uint32_t *order, int ntasks, const uint8_t *tb,
                               K2aResult *res, uint32_t *cig, void *stream);

/* Gap-linear X-drop extension (ksw2_lane_extf.h): one alignment per wavefront; class 0..2 keep the three state arrays in LDS
 * (targets up to 1024 / 4096 / 21504 residues), class 3 in `scratch` (3 x 16-padded target length bytes at pairs[i].tb_off).
 * Class 4 / 5: register window of 4 / 8 slots, for alignments whose band never holds more than K2A_EXTF_WIN_SPAN(K) positions.
 * pairs[i].zdrop = the X-drop threshold, pairs[i].w the band resolved as in ksw2_extf2_sse.c:23. */
int k2a_shim_launch_extf(int cls, const K2aExtf *par, const K2aPair *pairs, const uint32_t *order, int ntasks, const uint8_t *seq,
                         uint8_t *scratch, K2aResult *res, void *stream);

/* SSE-compatible mode (ksw2_lane_ssec.h): ksw_extz2_sse / ksw_extd2_sse as the reference's SSE kernels return them.  One
 * alignment per wavefront; pairs[i].bnd_off = offset of its state arrays in `scratch` in 16-byte units ((5 or 7) + 4 bytes per
 * 16-padded target position), tb_off = its direction matrix ((qlen + tlen - 1) * k2a_ssec_ncol bytes), pad = K2A_SSEC_* bits,
 * tlen = tlen_full, w = the band resolved as in ksw2_extz2_sse.c:72. */
/* lds_bytes > 0: the tasks' state arrays fit that many bytes each and live in LDS (one wavefront per workgroup) instead of `scratch` */
int k2a_shim_launch_ssec(int dual, int mode, size_t lds_bytes, const K2aSsec *par, const K2aPair *pairs, const uint32_t *order, int ntasks, const uint8_t *seq,
                         uint8_t *tb, uint8_t *scratch, K2aResult *res, void *stream);
int k2a_shim_launch_ssec_blk(int dual, int mode, const K2aSsec *par, const K2aPair *pairs, const uint32_t *order, int ntasks, const uint8_t *seq, uint8_t *tb, K2aResult *res, void *stream);
int k2a_shim_launch_ssec_trace(const K2aPair *pairs, const uint32_t *order, int ntasks, const uint8_t *tb, K2aResult *res, uint32_t *cig, void *stream);

/* Compaction: pool[pos[i] .. pos[i]+res[i].n_cigar) = cig[pairs[i].cig_off ..) for the n pairs of a plan
 * (pos = exclusive prefix sum of n_cigar, computed by the host), so that one D2H brings every CIGAR back. */
int k2a_shim_launch_compact(const K2aPair *pairs, const K2aResult *res, const uint32_t *pos, int n, const uint32_t *cig,
                            uint32_t *pool, void *stream);

#ifdef __cplusplus
}
#endif
#endif
