/*
 * pecaller_main.c -- host program (plain C) with the command line and on-disk formats of the reference's pecaller,
 * calling the MI355X per-site caller through the C-ABI of include/pemap_hip.h (pecall_dev_call_sites).
 *
 *   pecaller_hip pileup_ext sdx no_files outfile Prob_to_call theta haploid[y,n] no_threads use_pedfile[y,n] [pedfile denovo_rate] [guide.bed]
 *
 * (src/pecaller.c:227-257.)  It runs in the directory that holds the binary pileups, like the reference: every file
 * whose name contains `pileup_ext` is a sample, in directory order, named by its file name up to the first '.'
 * (pecaller.c:495-515).  Outputs, in the reference's formats: <outfile>.base.gz (a call and a posterior per sample and
 * column), <outfile>.snp and <outfile>.piles.gz (the variant columns), <outfile>.dist (coverage statistics).
 *
 * What replaces what: the dispatcher loop of main (pecaller.c:865-923: the 64-way merge of the pileup streams by
 * position, the reference base and contig of the column) stays here on the host and fills tiles of columns; the worker
 * threads' call_single_base (1207-1691) is one pecall_dev_call_sites per tile; the worker's sprintf block (1564-1690) is
 * emit_rows below.  Rows are written in genome order (the reference's order depends on thread timing).
 *
 * With a BED guide file (the last argument, pecaller.c:925-1068) every position of the listed intervals is called, covered
 * or not, and columns on chrY / chrMT are called with HAPLOID forced (955-957).
 *
 * Not supported (an error, not a silent difference): more than 512 samples (up to 64 is the device caller's fast case).  `no_threads` - 1 threads (as many as the host has CPUs, at most 128) walk the pileup streams, format the rows and deflate <outfile>.base.gz.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <ctype.h>
#include <stdint.h>
#include <dirent.h>
#include <errno.h>
#include <math.h>
#include <zlib.h>
#include <unistd.h>
#include <time.h>
#include "../../include/pemap_hip.h"
#include "host_io.h"

#define MAX_SAMPLES 512         /* PCS_MAXN of the device caller */
#define NA 6
#define MAX_DIST 501            /* pecaller.c:222 */
/* columns per device call, and genome positions per range of the stream merge.  A call has a fixed part (two kernel launches, the
   transfers' latencies), so tiles are large; PECALLER_TILE_LOG2 (10..22) overrides the exponent (tests: several ranges on a small
   fixture) */
static size_t TILE = (size_t) 1 << 20;

static void
die (const char *fmt, const char *arg)
{
  printf (fmt, arg);
  printf ("\n");
  exit (1);
}

static const char GEN[16] = "ACGTDIMRWSYKEHN";  /* int_to_gen, pecaller.c:2910-2943 */
static const char *TYPE_NAME[7] = { "", "SNP", "DEL", "INS", "LOW", "MULTIALLELIC", "MESS" };  /* pecaller.c:523-528 */

static int
gen_to_int (char c)             /* pecaller.c:2869-2907: an unknown letter ends the run */
{
  const char *p = c ? strchr (GEN, c) : NULL;
  if (!p)
    {
      printf ("\n This is impossible\n Illegal character in gen_to_int %c\n\n", c);
      exit (1);
    }
  return (int) (p - GEN);
}

/* pecaller's own find_chrom, pecaller.c:1793-1816 (not pemapper's) */
static int
find_chrom (const unsigned int *pos, int first, int last, int try, unsigned this)
{
  if (first == last)
    return first;
  if (first >= try)
    return this > pos[first] ? first + 1 : first;
  if (last <= try)
    return last;
  if (pos[try] < this)
    return find_chrom (pos, try, last, (last + try) / 2, this);
  if (pos[try] > this)
    return find_chrom (pos, first, try, (try + first) / 2, this);
  return try + 1;
}

typedef struct
{
  zreader f;                    /* the stream is inflated by a thread of its own (host_io.h) */
  unsigned int cur;             /* position of the pending record, 0 = exhausted (pecaller.c:840-849) */
  unsigned short data[NA];
  char name[256];
  /* .dist statistics, pecaller.c:884-889, 1077-1140 */
  double mean;
  unsigned int base_count, max_coverage, counts[MAX_DIST];
} sample_t;

/* the pending record consumed: read the next one (pecaller.c:891-907) */
static void
advance (sample_t * s, int *running)
{
  /* (gzeof / gzread of 4 then 12 bytes in the reference: the end of the stream is a read of nothing) */
  zreader *z = &s->f;
  if (z->pos + 16 <= z->cur_len)
    {
      /* the whole record lies in the block at hand: 64 of these per column are the merge's inner loop */
      const char *q = z->ring[z->head] + z->pos;
      memcpy (&s->cur, q, sizeof (unsigned int));
      memcpy (s->data, q + 4, sizeof (unsigned short) * NA);
      z->pos += 16;
      return;
    }
  if (zr_read (&s->f, &s->cur, sizeof (unsigned int)) != 0)
    zr_read (&s->f, s->data, sizeof (unsigned short) * NA);
  else
    {
      s->cur = 0;
      (*running)--;
    }
}

typedef struct
{
  /* one tile of columns */
  uint16_t *reads;              /* [TILE][indiv][6] */
  uint8_t *ref_base, *chrom;
  char *ref_char;
  int *contig;
  unsigned int *pos;
  int8_t *call, *type;
  /* the posteriors that are not 1: the columns that have one (ascending) and their samples' posteriors (pecall_dev_call_sites_sparse) */
  uint32_t *post_site;
  double *post_rows;
  uint64_t n_post, post_cap;
  int32_t *ac, *denovo;
  long n;
} tile_t;

/* The rows of <outfile>.base.gz are put together in memory and handed to the parallel gz writer tile by tile (host_io.h: gzip
   members, the same bytes after inflation as gzprintf would have produced, pecaller.c:1760-1775).  A posterior of exactly 1 --
   nearly all of them -- prints as "1" under %g and is written without going through printf. */
typedef struct
{
  char *p;
  size_t n, cap;
} sbuf;

static inline char *
sb_room (sbuf * b, size_t k)
{
  if (b->n + k > b->cap)
    {
      b->cap = 2 * (b->n + k) + (1 << 20);
      b->p = (char *) realloc (b->p, b->cap);
      if (!b->p)
        die ("\n pecaller_hip: out of memory for %s", "the output rows");
    }
  return b->p + b->n;
}

/* rows of columns [s0, s1): <outfile>.base.gz text into ob, <outfile>.snp text into sb, <outfile>.piles.gz text into pb */
static void
emit_rows (const tile_t * t, long s0, long s1, int indiv, char **contig_names, sbuf * ob, sbuf * sb, sbuf * pb)
{
  char minor[80], am_count[80], tmp[64];
  /* the first listed column at or behind s0 */
  uint64_t lp = 0, hi = t->n_post;
  while (lp < hi)
    {
      const uint64_t mid = (lp + hi) / 2;
      if ((long) t->post_site[mid] < s0)
        lp = mid + 1;
      else
        hi = mid;
    }
  for (long s = s0; s < s1; s++)
    {
      /* the column's posteriors: a row of the list, or all 1 (p = NULL) */
      const double *p = NULL;
      if (lp < t->n_post && (long) t->post_site[lp] == s)
        p = t->post_rows + (size_t) lp++ * indiv;
      if (t->type[s] < 0)
        continue;               /* reference base not A/C/G/T: the worker skips the column (pecaller.c:1208, 1718) */
      const char *frag = contig_names[t->contig[s]];
      const int8_t *call = t->call + s * indiv;
      const size_t fl = strlen (frag);
      {
        char *w = sb_room (ob, fl + 32 + (size_t) indiv * 32);
        char *w0 = w;
        *w++ = '\n';
        memcpy (w, frag, fl);
        w += fl;
        *w++ = '\t';
        w += sprintf (w, "%d", (int) t->pos[s]);
        *w++ = '\t';
        *w++ = t->ref_char[s];
        for (int i = 0; i < indiv; i++)
          {
            *w++ = '\t';
            if (call[i] < 14)
              {
                *w++ = GEN[call[i]];
                *w++ = '\t';
                if (!p || p[i] == 1.0)
                  *w++ = '1';
                else
                  w += sprintf (w, "%g", p[i]);
              }
            else
              {
                *w++ = 'N';
                *w++ = '\t';
                *w++ = '1';
              }
          }
        ob->n += (size_t) (w - w0);
      }
      if (t->type[s] == 0)
        continue;
      minor[0] = am_count[0] = '\0';
      for (int a = 0; a < NA; a++)
        if (t->ac[s * NA + a] > 0)
          {
            sprintf (tmp, "%c,", GEN[a]);
            strcat (minor, tmp);
            sprintf (tmp, "%d,", t->ac[s * NA + a]);
            strcat (am_count, tmp);
          }
      if (minor[0])
        {
          minor[strlen (minor) - 1] = '\0';
          am_count[strlen (am_count) - 1] = '\0';
        }
      sb->n += (size_t) sprintf (sb_room (sb, fl + 256), "\n%s\t%d\t%c\t%s\t%s\t%s%s", frag, (int) t->pos[s], t->ref_char[s], minor, am_count,
                                 t->denovo[s] > 0 ? "DENOVO_" : "", TYPE_NAME[t->type[s]]);
      pb->n += (size_t) sprintf (sb_room (pb, fl + 64), "\n%s\t%d\t%c", frag, (int) t->pos[s], t->ref_char[s]);
      for (int i = 0; i < indiv; i++)
        {
          sb->n += (size_t) sprintf (sb_room (sb, 64), "\t%c\t%g", GEN[call[i]], p ? p[i] : 1.0);
          const uint16_t *r = t->reads + ((size_t) s * indiv + i) * NA;
          for (int a = 0; a < NA; a++)
            pb->n += (size_t) sprintf (sb_room (pb, 16), "\t%d", (int) r[a]);
        }
    }
}

typedef struct
{
  const tile_t *t;
  long s0, s1;
  int indiv;
  char **contig_names;
  sbuf ob, sb, pb;
} emit_job;

static void *
emit_thread (void *arg)
{
  emit_job *j = (emit_job *) arg;
  emit_rows (j->t, j->s0, j->s1, j->indiv, j->contig_names, &j->ob, &j->sb, &j->pb);
  return NULL;
}

/* the rows of a tile, formatted by `threads` threads (contiguous runs of columns, put together in column order) */
static void
emit_tile (const tile_t * t, int indiv, char **contig_names, int threads, sbuf * ob, FILE * snpfile, gzFile pilefile)
{
  enum { MAXT = 128 };
  static emit_job jobs[MAXT];   /* (their buffers are kept from tile to tile) */
  pthread_t th[MAXT];
  if (threads > MAXT)
    threads = MAXT;
  if (threads < 1 || t->n < 4096)
    threads = 1;
  for (int k = 0; k < threads; k++)
    {
      jobs[k].t = t;
      jobs[k].s0 = t->n * k / threads;
      jobs[k].s1 = t->n * (k + 1) / threads;
      jobs[k].indiv = indiv;
      jobs[k].contig_names = contig_names;
      jobs[k].ob.n = jobs[k].sb.n = jobs[k].pb.n = 0;
    }
  for (int k = 1; k < threads; k++)
    if (pthread_create (&th[k], NULL, emit_thread, &jobs[k]))
      die ("\n pecaller_hip: can not start %s", "a formatting thread");
  emit_thread (&jobs[0]);
  for (int k = 1; k < threads; k++)
    pthread_join (th[k], NULL);
  for (int k = 0; k < threads; k++)
    {
      memcpy (sb_room (ob, jobs[k].ob.n), jobs[k].ob.p, jobs[k].ob.n);
      ob->n += jobs[k].ob.n;
      if (jobs[k].sb.n)
        fwrite (jobs[k].sb.p, 1, jobs[k].sb.n, snpfile);
      if (jobs[k].pb.n)
        gzwrite (pilefile, jobs[k].pb.p, (unsigned) jobs[k].pb.n);
    }
}

/* ---- the merge of the pileup streams without a guide file, a range of genome positions at a time.  The reference's dispatcher
        (find_lowest / the per-column loop, pecaller.c:891-1039, 1820-1833) takes the lowest pending position of all streams, makes
        a column of it from the streams that have a record there, and advances those: for streams in ascending order (as pemapper
        writes them) the columns are the union of the positions, each sample's counts where it has a record and zeros where it
        has none.  Here every stream is walked on its own over the positions [p0, p0 + TILE) into a plane of its own (a thread
        takes several streams; the statistics of <outfile>.dist are per stream and see the same records in the same order), and
        the planes are put together column by column, positions without any record left out. */
typedef struct
{
  sample_t *sm;
  int no_files, indiv, T, k;
  unsigned int p0;
  unsigned long long p1;
  uint16_t *planes;             /* [indiv][TILE][NA] */
  uint8_t *marks;               /* [T][TILE]: a stream of thread k has a record at the slot */
  /* second part */
  tile_t *t;
  long *chunk_base;             /* column of the first marked slot of each chunk of MG_CHUNK slots */
  const unsigned int *frag_pos;
  const uint8_t *chrom_type;
  const char *genome;
  unsigned int gsize;
  int no_contigs, start_chrom;
  /* guide mode (a stretch of a BED interval, pecaller.c:941-1039): EVERY position of [p0, p1) is a column, covered or not, on contig
     gwhich; the columns are appended to the tile from col0 on */
  int guide, gwhich;
  long col0;
  /* out: the last slot at which one of this thread's streams, live when the walk began, came to its end (-1: none did) */
  long end_slot;
} merge_ctx;

/* set by the parallel stream walk when a stream is not in ascending order: the run is abandoned and repeated with the serial merge */
static volatile int g_unordered = 0;
#define RC_UNORDERED 77

static size_t MG_CHUNK = 65536;        /* slots per work item of the column pass (<= TILE) */

static void
advance_nr (sample_t * s)
{
  zreader *z = &s->f;
  if (z->pos + 16 <= z->cur_len)
    {
      const char *q = z->ring[z->head] + z->pos;
      memcpy (&s->cur, q, sizeof (unsigned int));
      memcpy (s->data, q + 4, sizeof (unsigned short) * NA);
      z->pos += 16;
      return;
    }
  if (zr_read (&s->f, &s->cur, sizeof (unsigned int)) != 0)
    zr_read (&s->f, s->data, sizeof (unsigned short) * NA);
  else
    s->cur = 0;
}

static void *
merge_streams (void *arg)
{
  merge_ctx *c = (merge_ctx *) arg;
  uint8_t *mark = c->marks + (size_t) c->k * TILE;
  memset (mark, 0, TILE);
  if (c->guide && c->k == 0)
    memset (mark, 1, (size_t) (c->p1 - c->p0));
  c->end_slot = -1;
  for (int i = c->k; i < c->no_files; i += c->T)
    {
      sample_t *s = &c->sm[i];
      uint16_t *plane = c->planes + (size_t) i * TILE * NA;
      size_t done = 0;          /* slots of the plane written so far */
      const int live = s->cur != 0;
      if (c->guide)
        {
          /* records in front of the interval are passed over (pecaller.c:975-976); every position of the stretch counts as seen */
          while (s->cur != 0 && s->cur < c->p0)
            advance_nr (s);
          s->base_count += (unsigned int) (c->p1 - c->p0);
        }
      /* The records of the range.  The pending one first (s->cur / s->data); then straight out of the inflater's block as long as whole
         records lie in it -- 64 of these loops are the merge's time: one load of the position, the six counters copied as 8 + 4
         bytes, their sum for the stream's statistics.  (The coverage total is a sum of integers: it is kept in an integer and added
         to the double once per range; every partial sum is far below 2^53, so the double is the same.) */
      unsigned long long cov_sum = 0;
      unsigned int cov_max = s->max_coverage;
      unsigned int n_rec = 0;
      zreader *z = &s->f;
      while (s->cur != 0 && (unsigned long long) s->cur < c->p1)
        {
          unsigned int pos = s->cur;
          const char *rec = NULL;       /* NULL: the counters are in s->data */
          for (;;)
            {
              if ((unsigned long long) pos < (unsigned long long) c->p0 + done)
                {
                  /* a record at or in front of the stream's previous one: the parallel walk rests on ascending streams (as pemapper
                     writes them); the run starts over with the reference's own dispatcher, which takes what comes (run_once) */
                  g_unordered = 1;
                  return NULL;
                }
              const size_t slot = (size_t) (pos - c->p0);
              if (slot != done)
                memset (plane + done * NA, 0, (slot - done) * NA * sizeof (uint16_t));
              uint16_t *dst = plane + slot * NA;
              memcpy (dst, rec ? (const void *) (rec + 4) : (const void *) s->data, NA * sizeof (uint16_t));
              const unsigned int cov = (unsigned int) dst[0] + dst[1] + dst[2] + dst[3] + dst[4] + dst[5];
              cov_sum += cov;
              if (cov > cov_max)
                cov_max = cov;
              s->counts[cov < MAX_DIST - 1 ? cov : MAX_DIST - 1]++;
              n_rec++;
              mark[slot] = 1;
              done = slot + 1;
              /* the next record, if it lies whole in the block at hand and belongs to the range */
              if (z->pos + 16 > z->cur_len)
                break;
              rec = z->ring[z->head] + z->pos;
              memcpy (&pos, rec, sizeof pos);
              if (pos == 0 || (unsigned long long) pos >= c->p1)
                break;          /* (left where it is: advance_nr below reads it as the pending record) */
              z->pos += 16;
            }
          advance_nr (s);
        }
      s->mean += (double) cov_sum;
      s->max_coverage = cov_max;
      if (!c->guide)
        s->base_count += n_rec;
      memset (plane + done * NA, 0, ((size_t) TILE - done) * NA * sizeof (uint16_t));
      /* the stream ended inside this range: at the slot of its last record, or -- all its records lying in front of the range -- at
         the range's first position, where the reference reads past them (pecaller.c:975-993) */
      if (live && s->cur == 0)
        {
          const long at = done ? (long) done - 1 : 0;
          if (at > c->end_slot)
            c->end_slot = at;
        }
    }
  return NULL;
}

static int
slot_marked (const merge_ctx * c, size_t slot)
{
  for (int k = 0; k < c->T; k++)
    if (c->marks[(size_t) k * TILE + slot])
      return 1;
  return 0;
}

static void *
merge_count (void *arg)
{
  merge_ctx *c = (merge_ctx *) arg;
  for (size_t ch = (size_t) c->k; ch < TILE / MG_CHUNK; ch += (size_t) c->T)
    {
      long n = 0;
      for (size_t slot = ch * MG_CHUNK; slot < (ch + 1) * MG_CHUNK; slot++)
        n += slot_marked (c, slot);
      c->chunk_base[ch] = n;
    }
  return NULL;
}

static void *
merge_columns (void *arg)
{
  merge_ctx *c = (merge_ctx *) arg;
  tile_t *t = c->t;
  for (size_t ch = (size_t) c->k; ch < TILE / MG_CHUNK; ch += (size_t) c->T)
    {
      long col = c->col0 + c->chunk_base[ch];
      for (size_t slot = ch * MG_CHUNK; slot < (ch + 1) * MG_CHUNK; slot++)
        if (slot_marked (c, slot))
          {
            const unsigned int lowest = c->p0 + (unsigned int) slot;
            const int which = c->guide ? c->gwhich : find_chrom (c->frag_pos, 0, c->no_contigs - 1, c->start_chrom, lowest);
            const char ref = lowest < c->gsize ? c->genome[lowest] : '\0';
            t->ref_char[col] = ref;
            t->ref_base[col] = (uint8_t) gen_to_int (ref);
            t->contig[col] = which;
            t->pos[col] = 1 + lowest - c->frag_pos[which - 1];
            t->chrom[col] = c->chrom_type[which] | ((c->guide && (c->chrom_type[which] == 2 || c->chrom_type[which] == 3)) ? 16 : 0);
            uint16_t *dst = t->reads + (size_t) col * c->indiv * NA;
            for (int i = 0; i < c->no_files; i++)
              memcpy (dst + (size_t) i * NA, c->planes + ((size_t) i * TILE + slot) * NA, NA * sizeof (uint16_t));
            col++;
          }
    }
  return NULL;
}

static void
run_threads (void *(*fn) (void *), merge_ctx * ctx, int T)
{
  pthread_t th[128];
  for (int k = 1; k < T; k++)
    if (pthread_create (&th[k], NULL, fn, &ctx[k]))
      die ("\n pecaller_hip: can not start %s", "a merge thread");
  fn (&ctx[0]);
  for (int k = 1; k < T; k++)
    pthread_join (th[k], NULL);
}

/* ---- the device call of a tile and its text run on a thread each while the main thread merges the next tile: three sets of tile
        arrays go round (merge -> device -> rows -> free) */
#define N_TILES 3
typedef struct
{
  pthread_mutex_t mu;
  pthread_cond_t cv;
  tile_t free_tile[N_TILES];
  int n_free;
} tile_pool;

static void
pool_put (tile_pool * p, tile_t t)
{
  pthread_mutex_lock (&p->mu);
  p->free_tile[p->n_free++] = t;
  pthread_cond_broadcast (&p->cv);
  pthread_mutex_unlock (&p->mu);
}

static tile_t
pool_get (tile_pool * p)
{
  pthread_mutex_lock (&p->mu);
  while (p->n_free == 0)
    pthread_cond_wait (&p->cv, &p->mu);
  tile_t t = p->free_tile[--p->n_free];
  pthread_mutex_unlock (&p->mu);
  t.n = 0;
  return t;
}

typedef struct consumer_s
{
  pthread_t th;
  pthread_mutex_t mu;
  pthread_cond_t cv;
  int has_job, busy, stop;
  tile_t job;
  int role;                     /* 0: the device call, then on to `next`; 1: the rows, then back to the pool */
  struct consumer_s *next;
  tile_pool *pool;
  /* what the work needs */
  pecall_dev *pc;
  int indiv, haploid, threads;
  double threshold, theta;
  char **contig_names;
  sbuf *ob;
  FILE *snpfile;
  gzFile pilefile;
  pgz *outfile;
  const char *outname;
  double sec_dev, sec_text;
  long tot_cols;
} consumer_t;

static void consumer_wait_idle (consumer_t * c);

/* hand a tile to a stage (waits until the stage has given its previous one away) */
static void
consumer_give (consumer_t * c, tile_t t)
{
  consumer_wait_idle (c);
  pthread_mutex_lock (&c->mu);
  c->job = t;
  c->has_job = 1;
  pthread_cond_broadcast (&c->cv);
  pthread_mutex_unlock (&c->mu);
}

static void *
consumer_main (void *arg)
{
  consumer_t *c = (consumer_t *) arg;
  for (;;)
    {
      pthread_mutex_lock (&c->mu);
      while (!c->has_job && !c->stop)
        pthread_cond_wait (&c->cv, &c->mu);
      if (!c->has_job && c->stop)
        {
          pthread_mutex_unlock (&c->mu);
          return NULL;
        }
      c->has_job = 0;
      c->busy = 1;
      pthread_mutex_unlock (&c->mu);
      tile_t *t = &c->job;
      struct timespec a, b;
      clock_gettime (CLOCK_MONOTONIC, &a);
      if (c->role == 0)
        {
          int rc = pecall_dev_call_sites_sparse (c->pc, t->reads, t->ref_base, t->chrom, t->n, c->indiv, c->haploid, c->threshold, c->theta, t->call,
                                                 t->post_site, t->post_rows, t->post_cap, &t->n_post, t->type, t->ac, NULL, t->denovo);
          if (rc && t->n_post > t->post_cap)
            {
              /* more columns with a posterior that is not 1 than the list holds (one per 8 columns to begin with): a list of the size
                 the call asked for, and once more */
              t->post_cap = t->n_post + t->n_post / 8 + 1024;
              t->post_site = (uint32_t *) realloc (t->post_site, t->post_cap * sizeof (uint32_t));
              t->post_rows = (double *) realloc (t->post_rows, t->post_cap * (size_t) c->indiv * sizeof (double));
              if (!t->post_site || !t->post_rows)
                die ("\n pecaller_hip: out of memory for %s", "the list of posteriors");
              rc = pecall_dev_call_sites_sparse (c->pc, t->reads, t->ref_base, t->chrom, t->n, c->indiv, c->haploid, c->threshold, c->theta, t->call,
                                                 t->post_site, t->post_rows, t->post_cap, &t->n_post, t->type, t->ac, NULL, t->denovo);
            }
          if (rc)
            die ("\n pecaller_hip: %s", pecall_dev_last_error (c->pc));
          clock_gettime (CLOCK_MONOTONIC, &b);
          c->sec_dev += (double) (b.tv_sec - a.tv_sec) + 1e-9 * (double) (b.tv_nsec - a.tv_nsec);
          consumer_give (c->next, *t);
        }
      else
        {
          emit_tile (t, c->indiv, c->contig_names, c->threads, c->ob, c->snpfile, c->pilefile);
          if (pgz_write (c->outfile, c->ob->p, c->ob->n))
            die ("\n pecaller_hip: write to %s.base.gz failed", c->outname);
          c->ob->n = 0;
          clock_gettime (CLOCK_MONOTONIC, &b);
          c->sec_text += (double) (b.tv_sec - a.tv_sec) + 1e-9 * (double) (b.tv_nsec - a.tv_nsec);
          c->tot_cols += t->n;
          pool_put (c->pool, *t);
        }
      pthread_mutex_lock (&c->mu);
      c->busy = 0;
      pthread_cond_broadcast (&c->cv);
      pthread_mutex_unlock (&c->mu);
    }
}

static void
consumer_wait_idle (consumer_t * c)
{
  pthread_mutex_lock (&c->mu);
  while (c->has_job || c->busy)
    pthread_cond_wait (&c->cv, &c->mu);
  pthread_mutex_unlock (&c->mu);
}

static void
tile_alloc (tile_t * t, int indiv)
{
  t->reads = (uint16_t *) malloc ((size_t) TILE * indiv * NA * sizeof (uint16_t));
  t->ref_base = (uint8_t *) malloc (TILE);
  t->chrom = (uint8_t *) malloc (TILE);
  t->denovo = (int32_t *) malloc (TILE * sizeof (int32_t));
  t->ref_char = (char *) malloc (TILE);
  t->contig = (int *) malloc (TILE * sizeof (int));
  t->pos = (unsigned int *) malloc (TILE * sizeof (unsigned int));
  t->call = (int8_t *) malloc ((size_t) TILE * indiv);
  t->post_cap = TILE / 8 > 1024 ? TILE / 8 : 1024;
  {
    /* (tests: a list that is too short for the first tiles, so that the second call with the size asked for is taken) */
    const char *e = getenv ("PECALLER_POST_CAP");
    if (e && atol (e) >= 1)
      t->post_cap = (uint64_t) atol (e);
  }
  t->post_site = (uint32_t *) malloc (t->post_cap * sizeof (uint32_t));
  t->post_rows = (double *) malloc (t->post_cap * (size_t) indiv * sizeof (double));
  t->n_post = 0;
  t->type = (int8_t *) malloc (TILE);
  t->ac = (int32_t *) malloc ((size_t) TILE * NA * sizeof (int32_t));
  t->n = 0;
  if (!t->reads || !t->ref_base || !t->chrom || !t->denovo || !t->ref_char || !t->contig || !t->pos || !t->call || !t->post_site || !t->post_rows || !t->type || !t->ac)
    die ("\n pecaller_hip: out of memory for %s", "a tile");
}

static unsigned long long GUIDE_RANGE_MIN = 4096;      /* positions of a guide interval left at which the streams are walked in parallel (PECALLER_GUIDE_RANGE_MIN; tests: 64) */
/* the next line of the BED guide file (pecaller.c:1041-1066): contig, first and last position, 1-based -> 0 at its end */
static int
next_guide_interval (FILE * guide_file, char **contig_names, int no_contigs, const unsigned int *frag_pos, int *gwhich, unsigned int *lowest,
                     unsigned int *gend)
{
  char line[4096];
  line[0] = '\0';
  if (!feof (guide_file))
    fgets (line, 4095, guide_file);
  if (strlen (line) < 5)
    return 0;
  char *tok = strtok (line, "\t \n");
  *gwhich = -1;
  for (int i = 0; i < no_contigs; i++)
    if (strcmp (tok, contig_names[i]) == 0)
      {
        *gwhich = i;
        break;
      }
  if (*gwhich < 0)
    {
      printf ("\n For line chrom %s \n", tok);
      exit (1);
    }
  *lowest = frag_pos[*gwhich - 1] + (unsigned int) atoi (strtok (NULL, "\t \n")) - 1;
  *gend = frag_pos[*gwhich - 1] + (unsigned int) atoi (strtok (NULL, "\t \n")) - 1;
  return 1;
}


/* serial_merge: the pileup streams are merged the reference's way, one column at a time from the lowest pending position of all
   streams (find_lowest, pecaller.c:865-923, 1820-1833) -- whatever order the records come in; otherwise by the parallel walk, and
   RC_UNORDERED is returned as soon as that meets a record out of order (what has been written by then is written over by the repeat) */
static int
run_once (int argc, char *argv[], int serial_merge)
{
  char ss[4096], sdxname[4096];
  g_unordered = 0;
  if (argc < 10 || argc > 13)
    {
      printf
        ("\n Usage %s pileup_extension sdx_file no_files outfile Prob_to_call theta haploid[y,n] no_threads use_pedfile[y,n] [pedfilename] [denovo_mutation_rate] [guide_file_bed_format]\n",
         argv[0]);
      exit (1);
    }
  int no_threads = atoi (argv[8]);
  if (no_threads < 2 || no_threads > 200)
    {
      printf ("\n Number of threads is limited to 2 to 200.   You entered %d \n\n", no_threads);
      exit (1);
    }
  const double threshold = atof (argv[5]), theta = atof (argv[6]);
  if (theta < 1e-10 || theta > 0.5)
    {
      printf ("\n Encountered impossible value for theta = %g \n", theta);
      exit (1);
    }
  const int use_ped = (strchr (argv[9], 'Y') || strchr (argv[9], 'y')) ? 1 : 0;
  double denovo_rate = 0;
  if (use_ped)
    {
      if (argc < 12)
        die ("\n pecaller_hip: use_pedfile = %s needs the ped file name and the de-novo mutation rate", argv[9]);
      denovo_rate = atof (argv[11]);
      if (denovo_rate < 1e-30 || denovo_rate > theta)
        {
          printf ("\n Encounted impossible denovo mutation rate of %g with a theta of %g", denovo_rate, theta);
          exit (1);
        }
    }
  if (argc != (use_ped ? 12 : 10) && argc != (use_ped ? 13 : 11))
    die ("\n pecaller_hip: unexpected number of arguments (last: %s)", argv[argc - 1]);
  FILE *guide_file = NULL;
  if (argc == (use_ped ? 13 : 11) && !(guide_file = fopen (argv[argc - 1], "r")))
    die ("\n Can not open file %s for writing which should contain the guide_file", argv[argc - 1]);
  const int haploid = (strchr (argv[7], 'Y') || strchr (argv[7], 'y')) ? 1 : 0;

  gzFile pilefile;
  pgz outfile;
  sbuf ob = { NULL, 0, 0 };
  FILE *snpfile, *distfile;
  sprintf (ss, "%s.base.gz", argv[4]);
  const int pgz_rc = pgz_open (&outfile, ss, no_threads > 64 ? 64 : no_threads);
  /* the rows are ~280 bytes of text per column and 64 samples: at zlib's default level their deflate is the largest single item of the
     run's CPU time (12 of ~30 core-seconds per 8 M columns); level 2 takes half of that for a file 1.4 times the size */
  if (!getenv ("PEMAP_GZ_LEVEL"))
    outfile.level = 2;
  if (pgz_rc)
    die ("\n Can not open file %s", ss);
  sprintf (ss, "%s.snp", argv[4]);
  if (!(snpfile = fopen (ss, "w")))
    die ("\n Can not open file %s for writing", ss);
  sprintf (ss, "%s.dist", argv[4]);
  if (!(distfile = fopen (ss, "w")))
    die ("\n Can not open file %s for writing", ss);
  sprintf (ss, "%s.piles.gz", argv[4]);
  if (!(pilefile = gzopen (ss, "w")))
    die ("\n Can not open file %s for writing", ss);
  gzbuffer (pilefile, 131072);

  /* ---- .sdx: contig ends in .seq coordinates (length + 15 each), names, the chrY flag (pecaller.c:447-483) */
  strcpy (sdxname, argv[2]);
  FILE *sfile = fopen (sdxname, "r");
  if (!sfile)
    die ("\n Can not open file %s", sdxname);
  if (strstr (sdxname, ".sdx"))
    for (int i = (int) strlen (sdxname) - 1; i > 0; i--)
      if (sdxname[i] == '.')
        {
          sdxname[i] = '\0';
          break;
        }
  if (!fgets (ss, 256, sfile))
    die ("\n Empty file %s", argv[2]);
  const int no_contigs = atoi (ss);
  unsigned int *frag_store = (unsigned int *) calloc (no_contigs + 2, sizeof (unsigned int)), *frag_pos = frag_store + 1;
  char **contig_names = (char **) calloc (no_contigs + 1, sizeof (char *));
  uint8_t *chrom_type = (uint8_t *) calloc (no_contigs + 1, 1);   /* AUTO 0, CHRX 1, CHRY 2, CHRMT 3 (pecaller.c:98-101) */
  frag_pos[-1] = 0;
  for (int i = 0; i < no_contigs; i++)
    {
      if (!fgets (ss, 1024, sfile))
        die ("\n Short file %s", argv[2]);
      char *tok = strtok (ss, "\t \n");
      frag_pos[i] = (unsigned int) atoi (tok) + 15 + frag_pos[i - 1];
      tok = strtok (NULL, "\t \n");
      contig_names[i] = strdup (tok);
      char low[1024];
      strcpy (low, tok);
      char *pre = strtok (low, ":_- \n");
      for (char *q = pre; q && *q; q++)
        *q = (char) tolower (*q);
      chrom_type[i] = !pre ? 0 : !strcmp (pre, "chrx") ? 1 : !strcmp (pre, "chry") ? 2 : !strcmp (pre, "chrmt") ? 3 : 0;
    }
  fclose (sfile);

  /* ---- the reference letters: the whole .seq in memory (the reference pages 50 MB windows through gzseek, 1753-1789) */
  sprintf (ss, "%s.seq", sdxname);
  gzFile reffile = gzopen (ss, "r");
  if (!reffile)
    die ("\n Can not open file %s for reading", ss);
  gzbuffer (reffile, 1 << 22);
  const size_t gsize = frag_pos[no_contigs - 1];
  char *genome = (char *) calloc (gsize + 1, 1);
  for (size_t got = 0; got < gsize;)
    {
      int n = gzread (reffile, genome + got, (unsigned) ((gsize - got) > (1u << 30) ? (1u << 30) : (gsize - got)));
      if (n <= 0)
        break;
      got += (size_t) n;
    }
  gzclose (reffile);

  /* ---- the samples: directory order (pecaller.c:486-520) */
  int no_files = atoi (argv[3]);
  sample_t *sm = (sample_t *) calloc (no_files + 1, sizeof (sample_t));
  DIR *dir = opendir (".");
  if (!dir)
    {
      fprintf (stderr, "%s %d: opendir() failed (%s)\n", __FILE__, __LINE__, strerror (errno));
      exit (-1);
    }
  int found = 0;
  for (struct dirent * de = readdir (dir); de != NULL && found <= no_files; de = readdir (dir))
    if (strstr (de->d_name, argv[1]) != NULL)
      {
        if (found == no_files)
          {
            found++;
            break;
          }
        if (zr_open (&sm[found].f, de->d_name))
          die ("\n Can not open file %s which should contain pileup information", de->d_name);
        strncpy (ss, de->d_name, sizeof ss - 1);
        char *tok = strtok (ss, "\n.\t ");
        strncpy (sm[found].name, tok ? tok : "", sizeof sm[found].name - 1);
        found++;
      }
  closedir (dir);
  if (found > no_files)
    die ("%s", "\n Found more files than you specified \n");
  no_files = found;
  const int indiv = no_files;
  printf ("\n Found a total of %d individuals\n\n", indiv);
  if (indiv < 1 || indiv > MAX_SAMPLES)
    die ("\n pecaller_hip: %s samples; the device caller takes 1 to 512 (64 and fewer are its fast case)", argv[3]);

  pecall_dev *pc;
  if (pecall_dev_create (&pc, getenv ("PEMAP_DEVICE") ? atoi (getenv ("PEMAP_DEVICE")) : 0))
    die ("\n pecaller_hip: %s", pecall_dev_last_error (NULL));
  if (use_ped)
    {
      /* the ped file: family, individual, father, mother, sex per line (pecaller.c:561-604); parents that are not among the
         samples are ignored; a parent's kids are numbered in the order of the lines */
      static int dad[MAX_SAMPLES], mom[MAX_SAMPLES], sex[MAX_SAMPLES], nk[MAX_SAMPLES], kid[MAX_SAMPLES][2 * MAX_SAMPLES], off[MAX_SAMPLES + 1],
        list[2 * MAX_SAMPLES];
      for (int i = 0; i < MAX_SAMPLES; i++)
        {
          dad[i] = mom[i] = -1;
          sex[i] = nk[i] = 0;
        }
      FILE *pedfile = fopen (argv[10], "r");
      if (!pedfile)
        die ("\n Could Not open %s", argv[10]);
      char line[8192];
      while (fgets (line, sizeof line, pedfile) && strlen (line) > 5)
        {
          strtok (line, "\n\t ");
          char *ind = strtok (NULL, "\n\t "), *tf = strtok (NULL, "\n\t "), *tm = strtok (NULL, "\n\t "), *ts = strtok (NULL, "\n\t ");
          if (!ind || !tf || !tm || !ts)
            die ("\n pecaller_hip: short line in %s", argv[10]);
          for (int i = 0; i < indiv; i++)
            if (strcmp (ind, sm[i].name) == 0)
              {
                if (strcmp (tf, "0") != 0)
                  for (int j = 0; j < indiv; j++)
                    if (strcmp (tf, sm[j].name) == 0)
                      {
                        dad[i] = j;
                        kid[j][nk[j]++] = i;
                        break;
                      }
                if (strcmp (tm, "0") != 0)
                  for (int j = 0; j < indiv; j++)
                    if (strcmp (tm, sm[j].name) == 0)
                      {
                        mom[i] = j;
                        kid[j][nk[j]++] = i;
                        break;
                      }
                sex[i] = atoi (ts);
              }
        }
      fclose (pedfile);
      off[0] = 0;
      for (int i = 0; i < indiv; i++)
        {
          off[i + 1] = off[i] + nk[i];
          if (off[i + 1] > 2 * MAX_SAMPLES)
            die ("\n pecaller_hip: too many parent-child links in %s", argv[10]);
          for (int k = 0; k < nk[i]; k++)
            list[off[i] + k] = kid[i][k];
        }
      if (pecall_dev_set_pedigree (pc, indiv, dad, mom, sex, off, list, denovo_rate))
        die ("\n pecaller_hip: %s", pecall_dev_last_error (pc));
    }

  int running = no_files;
  for (int i = 0; i < no_files; i++)
    {
      if (zr_read (&sm[i].f, &sm[i].cur, sizeof (unsigned int)) != 0)
        zr_read (&sm[i].f, sm[i].data, sizeof (unsigned short) * NA);
      else
        sm[i].cur = 0;
      if (sm[i].cur == 0)
        running--;
    }
  fprintf (snpfile, "Fragment\tPosition\tReference\tAlleles\tAllele_Counts\tType");
  ob.n += (size_t) sprintf (sb_room (&ob, 64), "Fragment\tPosition\tReference");
  gzprintf (pilefile, "Fragment\tPosition\tReference");
  for (int i = 0; i < indiv; i++)
    {
      fprintf (snpfile, "\t%s\t", sm[i].name);
      ob.n += (size_t) sprintf (sb_room (&ob, strlen (sm[i].name) + 8), "\t%s\t", sm[i].name);
      gzprintf (pilefile, "\t%s\t\t\t\t\t", sm[i].name);
    }

  struct timespec tstart, tc0, tc1;
  double sec_dev = 0, sec_text = 0, sec_merge = 0, sec_wait = 0;
  long tot_cols = 0;
  clock_gettime (CLOCK_MONOTONIC, &tstart);
  tc0 = tc1 = tstart;
  {
    const char *tl = getenv ("PECALLER_TILE_LOG2");
    if (tl && atoi (tl) >= 10 && atoi (tl) <= 22)
      TILE = (size_t) 1 << atoi (tl);
    else
      {
        /* the host arrays hold ~54 bytes per (column, sample) -- the merge's planes, three tiles of reads and calls, the lists of the
           posteriors that are not 1: 2^20 columns are 3.6 GB with 64 samples; with more samples the tile shrinks so that columns x samples stays at that product */
        while (TILE > ((size_t) 1 << 16) && TILE * (size_t) no_files > ((size_t) 1 << 26))
          TILE >>= 1;
      }
    if (MG_CHUNK > TILE)
      MG_CHUNK = TILE;
    const char *gr = getenv ("PECALLER_GUIDE_RANGE_MIN");
    if (gr && atol (gr) >= 1)
      GUIDE_RANGE_MIN = (unsigned long long) atol (gr);
  }
  tile_pool pool;
  memset (&pool, 0, sizeof pool);
  pthread_mutex_init (&pool.mu, NULL);
  pthread_cond_init (&pool.cv, NULL);
  /* the tiles are handed to pecall_dev_call_sites again and again: page-locked once, their columns and results move by DMA
     straight from and to them (a refusal only means staged copies) */
  for (int k = 0; k < N_TILES; k++)
    {
      tile_t one;
      tile_alloc (&one, indiv);
      tile_t *tt = &one;
      (void) pecall_dev_pin_host (pc, tt->reads, (uint64_t) TILE * indiv * NA * sizeof (uint16_t));
      (void) pecall_dev_pin_host (pc, tt->ref_base, (uint64_t) TILE);
      (void) pecall_dev_pin_host (pc, tt->chrom, (uint64_t) TILE);
      (void) pecall_dev_pin_host (pc, tt->call, (uint64_t) TILE * indiv);
      /* (the list of posteriors is filled by the library with plain copies: not page-locked, so that it can be re-allocated freely) */
      (void) pecall_dev_pin_host (pc, tt->type, (uint64_t) TILE);
      (void) pecall_dev_pin_host (pc, tt->ac, (uint64_t) TILE * NA * sizeof (int32_t));
      (void) pecall_dev_pin_host (pc, tt->denovo, (uint64_t) TILE * sizeof (int32_t));
      pool_put (&pool, one);
    }
  tile_t t = pool_get (&pool);
  /* the threads of the merge and of the row formatting: the reference's worker threads minus its dispatcher, as many as the host has CPUs and the run has streams, at most 128 */
  int MT = no_threads - 1;
  {
    const long ncpu = sysconf (_SC_NPROCESSORS_ONLN);
    if (ncpu > 0 && MT > (int) ncpu)
      MT = (int) ncpu;
    if (MT > 128)
      MT = 128;
    if (MT > no_files)
      MT = no_files;
    if (MT < 1)
      MT = 1;
  }
  merge_ctx mc[128];
  long *chunk_base = (long *) calloc (TILE / MG_CHUNK, sizeof (long));
  uint16_t *planes = NULL;
  uint8_t *marks = NULL;
  /* (with a guide file too: its long intervals go through the same merge) */
    {
      planes = (uint16_t *) malloc ((size_t) no_files * TILE * NA * sizeof (uint16_t));
      marks = (uint8_t *) malloc ((size_t) MT * TILE);
      if (!planes || !marks)
        die ("\n pecaller_hip: out of memory for %s", "the merge planes");
    }
  for (int k = 0; k < MT; k++)
    {
      memset (&mc[k], 0, sizeof mc[k]);
      mc[k].sm = sm;
      mc[k].no_files = no_files;
      mc[k].indiv = indiv;
      mc[k].T = MT;
      mc[k].k = k;
      mc[k].planes = planes;
      mc[k].marks = marks;
      mc[k].t = &t;
      mc[k].chunk_base = chunk_base;
      mc[k].frag_pos = frag_pos;
      mc[k].chrom_type = chrom_type;
      mc[k].genome = genome;
      mc[k].gsize = gsize;
      mc[k].no_contigs = no_contigs;
    }
  consumer_t cons, rows;
  memset (&cons, 0, sizeof cons);
  pthread_mutex_init (&cons.mu, NULL);
  pthread_cond_init (&cons.cv, NULL);
  cons.pc = pc;
  cons.indiv = indiv;
  cons.haploid = haploid;
  cons.threads = MT;
  cons.threshold = threshold;
  cons.theta = theta;
  cons.contig_names = contig_names;
  cons.ob = &ob;
  cons.snpfile = snpfile;
  cons.pilefile = pilefile;
  cons.outfile = &outfile;
  cons.outname = argv[4];
  rows = cons;
  pthread_mutex_init (&rows.mu, NULL);
  pthread_cond_init (&rows.cv, NULL);
  rows.role = 1;
  rows.pool = &pool;
  cons.role = 0;
  cons.next = &rows;
  if (pthread_create (&rows.th, NULL, consumer_main, &rows) || pthread_create (&cons.th, NULL, consumer_main, &cons))
    die ("\n pecaller_hip: can not start %s", "the device and the text threads");
  clock_gettime (CLOCK_MONOTONIC, &tc0);
  t.n = 0;
  unsigned int tot_bases = 0;
  const int start_chrom = (no_contigs - 1) / 2 > 0 ? (no_contigs - 1) / 2 : 0;
  for (int k = 0; k < MT; k++)
    mc[k].start_chrom = start_chrom;
  if (serial_merge)
    GUIDE_RANGE_MIN = ~0ull;    /* (every guide position through the per-column scan, which is the reference's) */
  /* guide mode state: the current interval [lowest, end] of contig `gwhich` (pecaller.c:927-953, 1040-1066) */
  unsigned int lowest = 0, gend = 0;
  int gwhich = -1;
  if (guide_file)
    {
      char line[4096];
      if (!fgets (line, 4095, guide_file))
        running = 0;
      else
        {
          char *tok = strtok (line, "\t \n");
          for (int i = 0; i < no_contigs && tok; i++)
            if (strcmp (tok, contig_names[i]) == 0)
              {
                gwhich = i;
                break;
              }
          if (gwhich < 0)
            {
              printf ("\n For line chrom %s \n", tok ? tok : "");
              exit (1);
            }
          lowest = frag_pos[gwhich - 1] + (unsigned int) atoi (strtok (NULL, "\t \n")) - 1;
          gend = frag_pos[gwhich - 1] + (unsigned int) atoi (strtok (NULL, "\t \n")) - 1;
        }
    }
  while (running > 0 || t.n > 0)
    {
      int tile_done = 0;
      if (running > 0 && !guide_file && serial_merge)
        {
          /* one column, the reference's way (pecaller.c:865-923): the lowest pending position of all streams, the streams that have a
             record there, zeros for the others; a stream's records are taken in the order they come */
          unsigned int lowest = 0;
          for (int i = 0; i < no_files; i++)
            if (sm[i].cur > 0 && (lowest == 0 || sm[i].cur < lowest))
              lowest = sm[i].cur;
          const int which = find_chrom (frag_pos, 0, no_contigs - 1, start_chrom, lowest);
          const char ref = lowest < gsize ? genome[lowest] : '\0';
          const long sl = t.n++;
          t.ref_char[sl] = ref;
          t.ref_base[sl] = (uint8_t) gen_to_int (ref);
          t.contig[sl] = which;
          t.pos[sl] = 1 + lowest - frag_pos[which - 1];
          t.chrom[sl] = chrom_type[which];
          tot_bases++;
          uint16_t *col = t.reads + (size_t) sl * indiv * NA;
          for (int i = 0; i < no_files; i++)
            if (sm[i].cur == lowest)
              {
                unsigned int cov = 0;
                for (int a = 0; a < NA; a++)
                  {
                    col[i * NA + a] = sm[i].data[a];
                    cov += sm[i].data[a];
                  }
                sm[i].mean += (double) cov;
                if (cov > sm[i].max_coverage)
                  sm[i].max_coverage = cov;
                sm[i].counts[cov < MAX_DIST - 1 ? cov : MAX_DIST - 1]++;
                sm[i].base_count++;
                advance (&sm[i], &running);
              }
            else
              for (int a = 0; a < NA; a++)
                col[i * NA + a] = 0;
        }
      else if (running > 0 && !guide_file)
        {
          /* the next range of positions: from the lowest pending position of all streams (find_lowest, pecaller.c:1820-1833) */
          unsigned int p0 = 0;
          for (int i = 0; i < no_files; i++)
            if (sm[i].cur > 0 && (p0 == 0 || sm[i].cur < p0))
              p0 = sm[i].cur;
          for (int k = 0; k < MT; k++)
            {
              mc[k].p0 = p0;
              mc[k].p1 = (unsigned long long) p0 + TILE;
            }
          run_threads (merge_streams, mc, MT);
          if (g_unordered)
            {
              running = 0;      /* (the run is abandoned: what is in flight is finished and closed, run_once returns RC_UNORDERED) */
              t.n = 0;
              continue;
            }
          run_threads (merge_count, mc, MT);
          long ncol = 0;
          for (size_t ch = 0; ch < TILE / MG_CHUNK; ch++)
            {
              const long n = chunk_base[ch];
              chunk_base[ch] = ncol;
              ncol += n;
            }
          run_threads (merge_columns, mc, MT);
          t.n = ncol;
          tot_bases += (unsigned int) ncol;
          running = 0;
          for (int i = 0; i < no_files; i++)
            running += sm[i].cur != 0;
          tile_done = 1;
        }
      else if (running > 0 && (unsigned long long) gend + 1 - lowest >= GUIDE_RANGE_MIN && (size_t) t.n < TILE)
        {
          /* a long stretch of the guide interval: every stream is walked over it on its own, as without a guide file (the per-column
             scan of all streams below costs 1.4 us a column) */
          unsigned long long n = (unsigned long long) gend + 1 - lowest;
          if (n > TILE - (size_t) t.n)
            n = TILE - (size_t) t.n;
          for (int k = 0; k < MT; k++)
            {
              mc[k].p0 = lowest;
              mc[k].p1 = (unsigned long long) lowest + n;
              mc[k].guide = 1;
              mc[k].gwhich = gwhich;
              mc[k].col0 = t.n;
            }
          run_threads (merge_streams, mc, MT);
          if (g_unordered)
            {
              running = 0;
              t.n = 0;
              continue;
            }
          /* The reference's loop runs while a stream is open (pecaller.c:952): the column at which the last stream ends is the last one.
             The walk above went over the whole stretch: cut it there, and take the positions behind the cut out of every stream's
             count of positions seen again. */
          {
            int still = 0;
            for (int i = 0; i < no_files; i++)
              still += sm[i].cur != 0;
            if (still == 0)
              {
                long last = 0;
                for (int k = 0; k < MT; k++)
                  if (mc[k].end_slot > last)
                    last = mc[k].end_slot;
                const unsigned long long keep = (unsigned long long) last + 1;
                if (keep < n)
                  {
                    memset (marks + keep, 0, (size_t) (n - keep));
                    for (int i = 0; i < no_files; i++)
                      sm[i].base_count -= (unsigned int) (n - keep);
                    n = keep;
                  }
                running = 0;
              }
          }
          run_threads (merge_count, mc, MT);
          long ncol = 0;
          for (size_t ch = 0; ch < TILE / MG_CHUNK; ch++)
            {
              const long nn = chunk_base[ch];
              chunk_base[ch] = ncol;
              ncol += nn;
            }
          run_threads (merge_columns, mc, MT);
          t.n += ncol;
          tot_bases += (unsigned int) ncol;
          lowest += (unsigned int) n;
          if (running > 0 && lowest > gend && !next_guide_interval (guide_file, contig_names, no_contigs, frag_pos, &gwhich, &lowest, &gend))
            running = 0;
        }
      else if (running > 0)
        {
          /* one position of the guide interval (pecaller.c:941-1039) */
          const char ref = lowest < gsize ? genome[lowest] : '\0';
          const long s = t.n++;
          t.ref_char[s] = ref;
          t.ref_base[s] = (uint8_t) gen_to_int (ref);
          t.contig[s] = gwhich;
          t.pos[s] = 1 + lowest - frag_pos[gwhich - 1];
          t.chrom[s] = chrom_type[gwhich] | ((chrom_type[gwhich] == 2 || chrom_type[gwhich] == 3) ? 16 : 0);
          tot_bases++;
          uint16_t *col = t.reads + (size_t) s * indiv * NA;
          for (int i = 0; i < no_files; i++)
            {
              while (sm[i].cur < lowest && sm[i].cur > 0)
                advance (&sm[i], &running);
              if (sm[i].cur == lowest)
                {
                  unsigned int cov = 0;
                  for (int a = 0; a < NA; a++)
                    {
                      col[i * NA + a] = sm[i].data[a];
                      cov += sm[i].data[a];
                    }
                  sm[i].mean += (double) cov;
                  if (cov > sm[i].max_coverage)
                    sm[i].max_coverage = cov;
                  sm[i].counts[cov < MAX_DIST - 1 ? cov : MAX_DIST - 1]++;
                  sm[i].base_count++;
                  advance (&sm[i], &running);
                }
              else
                {
                  for (int a = 0; a < NA; a++)
                    col[i * NA + a] = 0;
                  sm[i].base_count++;
                }
            }
          lowest++;
          if (lowest > gend && !next_guide_interval (guide_file, contig_names, no_contigs, frag_pos, &gwhich, &lowest, &gend))
            running = 0;
        }
      if (tile_done || (size_t) t.n == TILE || (running <= 0 && t.n > 0))
        {
          clock_gettime (CLOCK_MONOTONIC, &tc1);
          sec_merge += (double) (tc1.tv_sec - tc0.tv_sec) + 1e-9 * (double) (tc1.tv_nsec - tc0.tv_nsec);
          /* hand the tile to the device thread and go on with a free set of arrays */
          consumer_give (&cons, t);
          t = pool_get (&pool);
          t.n = 0;
          clock_gettime (CLOCK_MONOTONIC, &tc0);
          sec_wait += (double) (tc0.tv_sec - tc1.tv_sec) + 1e-9 * (double) (tc0.tv_nsec - tc1.tv_nsec);
        }
    }
  for (int st = 0; st < 2; st++)
    {
      consumer_t *cc = st ? &rows : &cons;      /* (the device thread has passed its last tile on before it is idle) */
      consumer_wait_idle (cc);
      pthread_mutex_lock (&cc->mu);
      cc->stop = 1;
      pthread_cond_broadcast (&cc->cv);
      pthread_mutex_unlock (&cc->mu);
      pthread_join (cc->th, NULL);
    }
  sec_dev = cons.sec_dev;
  sec_text = rows.sec_text;
  tot_cols = rows.tot_cols;

  /* ---- <outfile>.dist, pecaller.c:1077-1140 */
  unsigned int *tot_1x = (unsigned int *) calloc (no_files, sizeof (unsigned int)), *tot_8x = (unsigned int *) calloc (no_files, sizeof (unsigned int));
  int *median = (int *) calloc (no_files, sizeof (int));
  for (int i = 0; i < no_files; i++)
    {
      if (sm[i].base_count > 0)
        sm[i].mean /= (double) sm[i].base_count;
      for (int j = 8; j < MAX_DIST; j++)
        tot_8x[i] += sm[i].counts[j];
      tot_1x[i] = tot_8x[i];
      for (int j = 1; j < 8; j++)
        tot_1x[i] += sm[i].counts[j];
      sm[i].counts[0] = tot_bases - tot_1x[i];
      long median_count = sm[i].counts[0];
      const long stop = tot_bases / 2;
      for (int j = 1; j < MAX_DIST; j++)
        {
          if (median_count > stop)
            break;
          median_count += sm[i].counts[++median[i]];
        }
    }
  fprintf (distfile, "Category");
  for (int i = 0; i < no_files; i++)
    fprintf (distfile, "\t%s", sm[i].name);
  fprintf (distfile, "\nTotal Number of bases in target");
  for (int i = 0; i < no_files; i++)
    fprintf (distfile, "\t%u", tot_bases);
  fprintf (distfile, "\nTotal Number of bases with at least 1x coverage");
  for (int i = 0; i < no_files; i++)
    fprintf (distfile, "\t%u", tot_1x[i]);
  fprintf (distfile, "\nTotal Number of bases with at least 8x coverage");
  for (int i = 0; i < no_files; i++)
    fprintf (distfile, "\t%u", tot_8x[i]);
  fprintf (distfile, "\nMean depth of coverage");
  for (int i = 0; i < no_files; i++)
    fprintf (distfile, "\t%g", sm[i].mean);
  fprintf (distfile, "\nMedian depth of coverage");
  for (int i = 0; i < no_files; i++)
    fprintf (distfile, "\t%d", median[i]);
  fprintf (distfile, "\nMaximum depth of coverage");
  for (int i = 0; i < no_files; i++)
    fprintf (distfile, "\t%d", (int) sm[i].max_coverage);
  fprintf (distfile, "\n\nDepth");
  for (int j = 0; j < MAX_DIST - 1; j++)
    {
      fprintf (distfile, "\n%d", j);
      for (int i = 0; i < no_files; i++)
        fprintf (distfile, "\t%u", sm[i].counts[j]);
    }
  fprintf (distfile, "\n%d+", MAX_DIST - 1);
  for (int i = 0; i < no_files; i++)
    fprintf (distfile, "\t%u", sm[i].counts[MAX_DIST - 1]);
  fprintf (distfile, "\n");
  fclose (distfile);
  fclose (snpfile);
  if (ob.n && pgz_write (&outfile, ob.p, ob.n))         /* (the header line, when there was no column at all) */
    die ("\n pecaller_hip: write to %s.base.gz failed", argv[4]);
  if (pgz_close (&outfile))
    die ("\n pecaller_hip: closing %s.base.gz failed", argv[4]);
  gzclose (pilefile);
  clock_gettime (CLOCK_MONOTONIC, &tc1);
  {
    const double sec = (double) (tc1.tv_sec - tstart.tv_sec) + 1e-9 * (double) (tc1.tv_nsec - tstart.tv_nsec);
    printf ("\n pecaller_hip: %ld columns x %d samples merged, called and written in %.3f s (%.3f M columns/s; stream merge %.3f s + %.3f s waiting for the other thread: device calls %.3f s, rows and gz %.3f s) \n",
            tot_cols, indiv, sec, (double) tot_cols / (sec > 0 ? sec : 1) / 1e6, sec_merge, sec_wait, sec_dev, sec_text);
  }
  for (int i = 0; i < no_files; i++)
    zr_close (&sm[i].f);
  pecall_dev_destroy (pc);
  if (guide_file)
    fclose (guide_file);
  return g_unordered ? RC_UNORDERED : 0;
}

int
main (int argc, char *argv[])
{
  /* PECALLER_SERIAL_MERGE=1: the serial merge from the start (pileup files known to be out of order) */
  const char *e = getenv ("PECALLER_SERIAL_MERGE");
  int rc = run_once (argc, argv, e && atoi (e));
  if (rc == RC_UNORDERED)
    {
      printf ("\n pecaller_hip: a pileup stream is not in ascending order: starting over with the serial merge (the reference's dispatcher) \n");
      fflush (stdout);
      rc = run_once (argc, argv, 1);
    }
  return rc;
}
