/*
 * pemap_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement of the reference PEMapper hot path (wingolab-org/pecaller, src/pemapper.c:907-2289),
 * used as the checker in tests/, in __graft_entry__.smoke() and as bench.py's cpu_baseline leg.
 * Nothing under pecaller_amd/ may include, link or call this.
 *
 * Parity pin: checked against the compiled reference (oracle/_ref/pemapper, pemapper_tsw) on the
 * fixtures of tests/golden/ (.mfile words, decompressed pileup records, summary counters, per-site
 * insertion multisets); see tests/test_oracle_golden.py and tests/golden/make_golden.sh.
 */
#ifndef PEMAP_ORACLE_H
#define PEMAP_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORA_MAX_HITS 200        /* pemapper.c:162 */
#define ORA_TOO_MANY 100        /* pemapper.c:163 */
#define ORA_SLOP 10             /* pemapper.c:47  */
#define ORA_MAX_READ 300        /* pemapper.c:155 */

typedef struct ora_index
{
  /* direct mode (the reference's layout, pemapper.c:2129-2165) -- used when pos_index != NULL */
  const uint32_t *pos_index;    /* 2^32 + 1 */
  const uint32_t *mers;
  /* compact mode: the same slices through a sorted table of the distinct 16-mers that occur */
  const uint32_t *ukmer;        /* n_ukmer distinct k-mers, ascending */
  const uint32_t *ustart;       /* n_ukmer + 1 offsets into mers */
  uint64_t n_ukmer;
  uint64_t n_mers;
  const char *genome;           /* upper-cased .seq stream */
  uint64_t genome_size;
  const uint32_t *contig_starts;        /* n_contigs + 1, compressed (len-15) prefix sums, pemapper.c:434-448 */
  int n_contigs;
  int idepth;                   /* 16 */
} ora_index;

typedef struct ora_params
{
  int paired;
  int min_dist, max_dist;
  double min_align;             /* MIN_ALIGN, pemapper.c:244/292 */
  int bisulfite;
} ora_params;

/* per-end debug record: what initial_map returned and what each SW call produced */
typedef struct ora_end_dbg
{
  int n_hits;
  uint32_t spot[ORA_MAX_HITS];
  uint8_t orient[ORA_MAX_HITS];
  int32_t win_start_lo[ORA_MAX_HITS];   /* low 32 bits of window start (real coords) */
  int32_t win_len[ORA_MAX_HITS];
  double score[ORA_MAX_HITS];
  int32_t start[ORA_MAX_HITS][3];
} ora_end_dbg;

typedef struct ora_ins
{
  uint32_t pos;                 /* 0-based index into .seq */
  uint16_t len;
  char seq[ORA_MAX_READ];
} ora_ins;

typedef struct ora_state ora_state;

ora_state *ora_create (const ora_index * idx, const ora_params * prm);
void ora_destroy (ora_state * st);

/* Map n reads (pairs).  reads are `stride`-spaced, NUL terminated or len-delimited.  Thread count >= 1
 * (plain pthreads over contiguous ranges; pileup counters are updated with atomic adds, which is what
 * the reference's per-100-base mutexes amount to).  dbg1/dbg2 may be NULL. */
int ora_map_batch (ora_state * st, const char *reads1, const int *len1, const char *reads2, const int *len2,
                   long n, int stride, uint32_t * m1, uint32_t * m2, int *mapping_type,
                   ora_end_dbg * dbg1, ora_end_dbg * dbg2, int threads);

/* counters: [genome_size][6] u16 = A,C,G,T,Del,Ins  (pemapper.c:49-60) */
const uint16_t *ora_counts (ora_state * st);
long ora_n_ins (ora_state * st);
const ora_ins *ora_ins_log (ora_state * st);

/* summary counters (pemapper.c:1238-1265): out[0]=total_reads out[1]=total_bases out[2]=total_dist out[3]=no_dists,
 * out[4..12] = mate_counts[0..8] */
void ora_summary (ora_state * st, long *out13);

/* building blocks exposed for unit tests */
int ora_initial_map (const ora_index * idx, int bisulfite, const char *fwd, const char *rev, int len,
                     uint32_t * spots, uint8_t * orients);
void ora_revcomp (const char *in, char *out, int n);
int ora_find_chrom (const uint32_t * pos, int n_contigs, uint32_t x);
double ora_sw (const char *ref, int nn, const char *seq, int mm, int bisulfite, int *start3, double *planes /* 3*(nn+1)*(mm+1) or NULL */ );
uint32_t ora_kmer (const char *s);
void ora_neighbours (uint32_t kmer, uint32_t * out49);

#ifdef __cplusplus
}
#endif
#endif
