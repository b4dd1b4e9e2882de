/*
 * ksw2-test-amd -- command-line driver with the option set and TSV output of the reference's test program
 * (cli.c:159-259: getopt "t:w:R:rsgz:A:B:O:E:Ka", output cli.c:134-145), running every algorithm through
 * libksw2_amd.so (MI355X).  Lets a user A/B the two implementations byte for byte on stdout:
 *
 *     ksw2-test-amd [-t extz|extz2_sse|extd|extd2_sse|exts2_sse|extf2_sse|gg|gg2|gg2_sse] [-w band] [-z zdrop] [-r] [-s] [-g]
 *                   [-A match] [-B mismatch] [-O gapo[,gapo2]] [-E gape[,gape2]] [-R rep] [-a] [-b] <target.fa> <query.fa>
 *
 * Own code: FASTA reader on zlib's gzFile (plain or .gz input, as cli.c:210-211 through kseq.h; one record per '>' header),
 * ACGT -> 0..3, other -> 4 (cli.c:17-34,58-65).  If a file cannot be opened the argument itself is taken as the sequence
 * (cli.c:212-215).
 * -b (new): align all pairs in one batched call (ksw2amd_ext?_batch) instead of one call per pair.
 * -K (cli.c:177,206): results are allocated from a caller-side pool passed as `km`.  The reference's kalloc is out of scope
 * (SURVEY section 2), so the pool here is this program's own: it exports krealloc / kfree (kalloc.h:12-15), and the library
 * finds them exactly as it finds the real kalloc's in a minimap2-style caller.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include <zlib.h>
#include "../include/ksw2_amd.h"

/* the caller-side pool behind -K: a counting wrapper around the C allocator with the kalloc entry points' signatures */
typedef struct { long n_realloc, n_free; } pool_t;
void *krealloc(void *km, void *ptr, size_t size) { if (km) ++((pool_t*)km)->n_realloc; return realloc(ptr, size); }
void *kmalloc(void *km, size_t size) { return krealloc(km, 0, size); }
void kfree(void *km, void *ptr) { if (km) ++((pool_t*)km)->n_free; free(ptr); }

typedef struct { char *name, *seq; int len; } rec_t;
typedef struct { rec_t *a; int n, m; } recs_t;

static uint8_t code_of(int c)
{
	switch (c) {
	case 'A': case 'a': return 0;
	case 'C': case 'c': return 1;
	case 'G': case 'g': return 2;
	case 'T': case 't': return 3;
	default: return 4;
	}
}

static void push_rec(recs_t *r, const char *name, const char *seq, int len)
{
	if (r->n == r->m) { r->m = r->m ? r->m * 2 : 16; r->a = (rec_t*)realloc(r->a, sizeof(rec_t) * (size_t)r->m); }
	r->a[r->n].name = strdup(name);
	r->a[r->n].seq = (char*)malloc((size_t)len + 1);
	memcpy(r->a[r->n].seq, seq, (size_t)len);
	r->a[r->n].seq[len] = 0;
	r->a[r->n].len = len;
	++r->n;
}

/* one line of any length from a gzFile (plain files pass through zlib unchanged); returns the length or -1 at the end */
static ssize_t gz_getline(gzFile fp, char **line, size_t *cap)
{
	size_t n = 0;
	for (;;) {
		if (*cap < n + 4096) { *cap = (n + 4096) * 2; *line = (char*)realloc(*line, *cap); }
		if (!gzgets(fp, *line + n, (int)(*cap - n))) return n ? (ssize_t)n : -1;
		n += strlen(*line + n);
		if (n && (*line)[n - 1] == '\n') return (ssize_t)n;
	}
}

static int read_fasta(const char *fn, recs_t *out)
{
	gzFile fp = gzopen(fn, "r");
	char *line = 0, *seq = 0, name[1024] = "";
	size_t cap = 0, scap = 0;
	ssize_t got;
	int slen = 0, have = 0;
	if (!fp) return -1;
	while ((got = gz_getline(fp, &line, &cap)) >= 0) {
		while (got > 0 && (line[got - 1] == '\n' || line[got - 1] == '\r')) line[--got] = 0;
		if (line[0] == '>') {
			if (have) push_rec(out, name, seq ? seq : "", slen);
			sscanf(line + 1, "%1023s", name);
			slen = 0; have = 1;
		} else if (have && got > 0) {
			if ((size_t)slen + (size_t)got + 1 > scap) { scap = ((size_t)slen + (size_t)got + 1) * 2; seq = (char*)realloc(seq, scap); }
			memcpy(seq + slen, line, (size_t)got);
			slen += (int)got;
		}
	}
	if (have) push_rec(out, name, seq ? seq : "", slen);
	free(line); free(seq); gzclose(fp);
	return 0;
}

static void gen_simple_mat(int m, int8_t *mat, int8_t a, int8_t b)     /* cli.c:36-48: wildcard row/column = 0 */
{
	int i, j;
	a = a < 0 ? -a : a; b = b > 0 ? -b : b;
	for (i = 0; i < m; ++i)
		for (j = 0; j < m; ++j)
			mat[i * m + j] = (i == m - 1 || j == m - 1) ? 0 : i == j ? a : b;
}

static void print_aln(const char *tn, const char *qn, const ksw_extz_t *ez)
{
	int i;
	printf("%s\t%s\t%d\t%d\t%d\t%d", tn, qn, ez->score, ez->max, ez->max_t, ez->max_q);
	if (ez->n_cigar > 0) {
		putchar('\t');
		for (i = 0; i < ez->n_cigar; ++i) printf("%d%c", ez->cigar[i] >> 4, "MID"[ez->cigar[i] & 0xf]);
	}
	putchar('\n');
}

static uint8_t *encode(const rec_t *r)
{
	uint8_t *s = (uint8_t*)calloc((size_t)r->len + 1, 1);
	int i;
	for (i = 0; i < r->len; ++i) s[i] = code_of((unsigned char)r->seq[i]);
	return s;
}

int main(int argc, char *argv[])
{
	int8_t a = 2, b = 4, q = 4, e = 2, q2 = 13, e2 = 1, mat[25];
	int c, i, w = -1, flag = 0, rep = 1, zdrop = -1, pair = 1, batch = 0, njob = 0, *jobs = 0;
	const char *algo = "extd";              /* cli.c:164 */
	char *s;
	recs_t T = {0, 0, 0}, Q = {0, 0, 0};
	ksw_extz_t ez;
	pool_t pool = {0, 0};
	void *km = 0;

	while ((c = getopt(argc, argv, "t:w:R:rsgz:A:B:O:E:Kab")) >= 0) {
		if (c == 't') algo = optarg;
		else if (c == 'w') w = atoi(optarg);
		else if (c == 'R') rep = atoi(optarg);
		else if (c == 'z') zdrop = atoi(optarg);
		else if (c == 'r') flag |= KSW_EZ_RIGHT;
		else if (c == 's') flag |= KSW_EZ_SCORE_ONLY;
		else if (c == 'g') flag |= KSW_EZ_APPROX_MAX | KSW_EZ_APPROX_DROP;
		else if (c == 'a') pair = 0;
		else if (c == 'b') batch = 1;
		else if (c == 'K') km = &pool;
		else if (c == 'A') a = (int8_t)atoi(optarg);
		else if (c == 'B') b = (int8_t)atoi(optarg);
		else if (c == 'O') { q = q2 = (int8_t)strtol(optarg, &s, 10); if (*s == ',') q2 = (int8_t)strtol(s + 1, &s, 10); }
		else if (c == 'E') { e = e2 = (int8_t)strtol(optarg, &s, 10); if (*s == ',') e2 = (int8_t)strtol(s + 1, &s, 10); }
	}
	if (argc - optind < 2) {
		fprintf(stderr, "Usage: ksw2-test-amd [-t algo] [-w band] [-z zdrop] [-rsgab] [-A a] [-B b] [-O o1[,o2]] [-E e1[,e2]] [-R rep] <target.fa> <query.fa>\n");
		fprintf(stderr, "  algorithms: extz extz2_sse extd extd2_sse exts2_sse extf2_sse gg gg2 gg2_sse (all evaluated on the GPU by libksw2_amd, backend %s)\n", ksw2amd_backend());
		return 1;
	}
	gen_simple_mat(5, mat, a, b);
	if (read_fasta(argv[optind], &T) < 0) push_rec(&T, "target", argv[optind], (int)strlen(argv[optind]));
	if (read_fasta(argv[optind + 1], &Q) < 0) push_rec(&Q, "query", argv[optind + 1], (int)strlen(argv[optind + 1]));

	/* alignment jobs in the reference's output order: lock-step pairs, or for every query all targets (cli.c:220-243) */
	{
		int j;
		njob = 0;
		jobs = (int*)malloc(sizeof(int) * 2 * ((size_t)T.n * (size_t)(pair ? 1 : Q.n) + 1));
		if (pair) { for (i = 0; i < T.n && i < Q.n; ++i) { jobs[2 * njob] = i; jobs[2 * njob + 1] = i; ++njob; } }
		else for (j = 0; j < Q.n; ++j) for (i = 0; i < T.n; ++i) { jobs[2 * njob] = i; jobs[2 * njob + 1] = j; ++njob; }
	}
	memset(&ez, 0, sizeof(ez));
	if (batch && (strcmp(algo, "extz2_sse") == 0 || strcmp(algo, "extd2_sse") == 0)) {
		const int dual = strcmp(algo, "extd2_sse") == 0;
		const int n = njob;
		ksw2amd_pair_t *p = (ksw2amd_pair_t*)calloc((size_t)n + 1, sizeof(*p));
		ksw_extz_t *res = (ksw_extz_t*)calloc((size_t)n + 1, sizeof(*res));
		uint8_t **te = (uint8_t**)calloc((size_t)T.n + 1, sizeof(*te)), **qe = (uint8_t**)calloc((size_t)Q.n + 1, sizeof(*qe));
		ksw2amd_scoring_t sc;
		int k, rc = 0;
		for (i = 0; i < T.n; ++i) te[i] = encode(&T.a[i]);
		for (i = 0; i < Q.n; ++i) qe[i] = encode(&Q.a[i]);
		for (k = 0; k < n; ++k) {
			const int ti = jobs[2 * k], qj = jobs[2 * k + 1];
			p[k].query = qe[qj]; p[k].target = te[ti]; p[k].qlen = Q.a[qj].len; p[k].tlen = T.a[ti].len;
			p[k].w = w; p[k].zdrop = zdrop; p[k].end_bonus = 0; p[k].flag = flag;
		}
		sc.m = 5; sc.mat = mat; sc.q = q; sc.e = e; sc.q2 = q2; sc.e2 = e2;
		for (i = 0; i < rep && rc == 0; ++i)
			rc = dual ? ksw2amd_extd_batch(km, &sc, n, p, res) : ksw2amd_extz_batch(km, &sc, n, p, res);
		if (rc) { fprintf(stderr, "ERROR: %s\n", ksw2amd_last_error()); return 1; }
		for (k = 0; k < n; ++k) print_aln(T.a[jobs[2 * k]].name, Q.a[jobs[2 * k + 1]].name, &res[k]);
		return 0;
	}
	for (i = 0; i < njob; ++i) {
		const rec_t *tr = &T.a[jobs[2 * i]], *qr = &Q.a[jobs[2 * i + 1]];
		uint8_t *ts = encode(tr), *qs = encode(qr);
		const int ql = qr->len, tl = tr->len;
		int r;
		for (r = 0; r < rep; ++r) {
			/* cli.c:53-55 resets these per call; the callee resets the rest */
			ez.max_q = ez.max_t = ez.mqe_t = ez.mte_q = -1; ez.max = 0; ez.mqe = ez.mte = KSW_NEG_INF; ez.n_cigar = 0;
			if (strcmp(algo, "gg") == 0 || strcmp(algo, "gg2") == 0 || strcmp(algo, "gg2_sse") == 0) {
				int (*f)(void*, int, const uint8_t*, int, const uint8_t*, int8_t, const int8_t*, int8_t, int8_t, int, int*, int*, uint32_t**) =
					strcmp(algo, "gg") == 0 ? ksw_gg : strcmp(algo, "gg2") == 0 ? ksw_gg2 : ksw_gg2_sse;
				if ((flag & KSW_EZ_SCORE_ONLY) && strcmp(algo, "gg2_sse") != 0) ez.score = f(km, ql, qs, tl, ts, 5, mat, q, e, w, 0, 0, 0);
				else ez.score = f(km, ql, qs, tl, ts, 5, mat, q, e, w, &ez.m_cigar, &ez.n_cigar, &ez.cigar);
			} else if (strcmp(algo, "extz") == 0) ksw_extz(km, ql, qs, tl, ts, 5, mat, q, e, w, zdrop, flag, &ez);
			else if (strcmp(algo, "extz2_sse") == 0) ksw_extz2_sse(km, ql, qs, tl, ts, 5, mat, q, e, w, zdrop, 0, flag, &ez);
			else if (strcmp(algo, "extd") == 0) ksw_extd(km, ql, qs, tl, ts, 5, mat, q, e, q2, e2, w, zdrop, flag, &ez);
			else if (strcmp(algo, "extd2_sse") == 0) ksw_extd2_sse(km, ql, qs, tl, ts, 5, mat, q, e, q2, e2, w, zdrop, 0, flag, &ez);
			else if (strcmp(algo, "exts2_sse") == 0) {     /* cli.c:79-83: its own 1 / -2 matrix, q=2 e=1 q2=32 noncan=4, forward signals */
				int8_t smat[25];
				int x, y;
				for (x = 0; x < 5; ++x)
					for (y = 0; y < 5; ++y) smat[x * 5 + y] = (int8_t)((x == 4 || y == 4) ? 0 : x == y ? 1 : -2);
				ksw_exts2_sse(km, ql, qs, tl, ts, 5, smat, 2, 1, 32, 4, zdrop, 0, flag | KSW_EZ_SPLICE_FOR, 0, &ez);
			}
			else if (strcmp(algo, "extf2_sse") == 0) ksw_extf2_sse(km, ql, qs, tl, ts, mat[0], mat[1], e, w, zdrop, &ez);     /* cli.c:78 */
			else { fprintf(stderr, "ERROR: can't find algorithm '%s'\n", algo); return 1; }
		}
		print_aln(tr->name, qr->name, &ez);
		free(qs); free(ts);
	}
	kfree(km, ez.cigar);
	if (km && getenv("KSW2_TEST_POOL_STATS")) fprintf(stderr, "[ksw2-test-amd] pool: %ld krealloc, %ld kfree\n", pool.n_realloc, pool.n_free);
	return 0;
}
