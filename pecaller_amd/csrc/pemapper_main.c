/*
 * pemapper_main.c -- host program (plain C) with the command line and on-disk formats of the reference's pemapper
 * and pemapper_tsw, calling the MI355X hot path through the C-ABI of include/pemap_hip.h.
 *
 *   pemapper_hip out sdx s|sa file1 is_bisulfite min_match threads max_reads [trim_start trim_end]
 *   pemapper_hip out sdx p|pa file1 file2 max_dist min_dist is_bisulfite min_match threads max_reads [trim_start trim_end]
 *
 * (src/pemapper.c:226-358; with the two trailing arguments it behaves as src/pemapper_tsw.c:234-311: reads are trimmed
 * and array-file lines may carry an output base name in a second column, pemapper_tsw.c:266-280, 636-674.)
 * Outputs: <out>.pileup.gz (16-byte records), <out>.indel.txt.gz, <out>.summary.txt, <fastq>.mfile -- the reference's
 * formats (pemapper.c:775-781, 819-900).
 *
 * Differences from the reference, all outside the hot path: `threads` bounds the host threads that deflate the pileup (the
 * batch itself goes to the GPU) and, over three, the files of an array that are read side by side (a worker per file pair, at most 64:
 * PEMAPPER_FILE_WORKERS); batches are handed over with pemap_dev_submit_batch, so the next one is parsed while the
 * GPU maps; <out>.pileup.gz is a sequence of gzip members (it inflates to the reference's bytes); the index arrays are rebuilt on the GPU from <sdx>.seq instead of inflating the 16 GiB <sdx>.idx (the device
 * builder is verified to produce the reference builder's arrays; set PEMAP_INDEX_FROM_FILES=1 to load .idx/.mdx);
 * reads longer than PEMAP_MAX_READ or shorter than PEMAP_MIN_READ are an error instead of undefined behaviour.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <ctype.h>
#include <stdint.h>
#include <zlib.h>
#include <pthread.h>
#include <time.h>
#include <unistd.h>
#include "../../include/pemap_hip.h"
#include "host_io.h"

#define MAX_FILES 2000
#define BATCH_PAIRS (1 << 20)
#define ROW_STRIDE 288

static void
die (const char *msg, const char *arg)
{
  printf (msg, arg);            /* the reference prints to stdout and exits with 1 (pemapper.c:2808-2816) */
  printf ("\n");
  exit (1);
}

static void
ck (pemap_dev * dev, int rc)
{
  if (rc)
    {
      printf ("\n pemap_hip: %s\n", pemap_dev_last_error (dev));
      exit (1);
    }
}

/* ---- line reader over a gz stream: my_gzgets (pemapper.c:2447-2483) without its 1.2 GB slab.  The stream is inflated by
        a thread of its own into a ring of blocks, so the two mate files inflate in parallel with each other, with the line
        scan and with the GPU batch in flight (the reference inflates on the main thread, pemapper.c:626) */
#define LR_BLOCK (1 << 22)
#define LR_RING 8
typedef struct
{
  gzsrc src;
  char *own[LR_RING];           /* the ring's own buffers (zlib mode only) */
  char *buf;
  size_t cap, len, pos;
  int eof;
  pthread_t th;
  pthread_mutex_t mu;
  pthread_cond_t cv;
  char *ring[LR_RING];
  int ring_len[LR_RING];
  int head, count, done, stop;
} lreader;

static void *
lr_inflate (void *arg)
{
  lreader *r = (lreader *) arg;
  int tail = 0;
  for (;;)
    {
      pthread_mutex_lock (&r->mu);
      while (r->count == LR_RING && !r->stop)
        pthread_cond_wait (&r->cv, &r->mu);
      const int stop = r->stop;
      if (stop)
        {
          r->done = 1;
          pthread_cond_broadcast (&r->cv);
        }
      pthread_mutex_unlock (&r->mu);
      if (stop)
        return NULL;
      /* the reference's my_gzgets ends the file at the first failed read; so does this reader, but not silently */
      char *blk = NULL;
      int got = 0;
      const int rc = gzsrc_next (&r->src, r->own[tail], &blk, &got);
      if (rc < 0)
        fprintf (stderr, "\n Warning: a read file is truncated or not a valid gzip stream (%s); the reads before the damage are mapped \n",
                 r->src.err ? r->src.err : "zlib");
      if (rc <= 0)
        got = 0;
      pthread_mutex_lock (&r->mu);
      if (got <= 0)
        r->done = 1;
      else
        {
          r->ring[tail] = blk;
          r->ring_len[tail] = got;
          r->count++;
        }
      pthread_cond_broadcast (&r->cv);
      pthread_mutex_unlock (&r->mu);
      if (got <= 0)
        return NULL;
      tail = (tail + 1) % LR_RING;
    }
}

static void
lr_open (lreader * r, const char *path)
{
  if (gzsrc_open (&r->src, path, LR_BLOCK, LR_RING))
    die ("\n Can not open file %s for reading", path);
  r->cap = 2 * LR_BLOCK;
  r->buf = (char *) malloc (r->cap + 1);
  r->len = r->pos = 0;
  r->eof = 0;
  r->head = r->count = r->done = r->stop = 0;
  for (int i = 0; i < LR_RING; i++)
    {
      r->ring[i] = NULL;
      r->own[i] = r->src.mode == 0 ? (char *) malloc (LR_BLOCK) : NULL;
    }
  pthread_mutex_init (&r->mu, NULL);
  pthread_cond_init (&r->cv, NULL);
  if (pthread_create (&r->th, NULL, lr_inflate, r))
    die ("\n Can not start the reader thread for %s", path);
}

static void
lr_close (lreader * r)
{
  /* the caller may stop before the end of the file (max_reads) */
  pthread_mutex_lock (&r->mu);
  r->stop = 1;
  pthread_cond_broadcast (&r->cv);
  pthread_mutex_unlock (&r->mu);
  pthread_join (r->th, NULL);
  gzsrc_close (&r->src);
  for (int i = 0; i < LR_RING; i++)
    free (r->own[i]);
  pthread_mutex_destroy (&r->mu);
  pthread_cond_destroy (&r->cv);
  free (r->buf);
}

/* next line (its '\n' not counted; a '\r' stays, as in the reference; NOT 0-terminated unless it was put together in r->buf) and its
   length, NULL at end of data.  A line that lies inside one
   block of the ring is returned where it is (valid until the next call); only a line that straddles blocks is put together in
   r->buf.  (The reference's my_gzgets copies every line, pemapper.c:2447-2483.) */
static char *
lr_gets (lreader * r, size_t *len_out)
{
  size_t have = 0;              /* bytes of a straddling line collected in r->buf so far */
  for (;;)
    {
      if (r->pos < r->len)
        {
          char *base = r->ring[r->head] + r->pos;
          char *nl = (char *) memchr (base, '\n', r->len - r->pos);
          const size_t n = nl ? (size_t) (nl - base) : r->len - r->pos;
          if (nl && have == 0)
            {
              /* (the block is left as it is: it may be the input file's mapping, or the window the inflater copies matches from) */
              r->pos += n + 1;
              if (len_out)
                *len_out = n;
              return base;
            }
          if (have + n + 1 > r->cap)
            {
              r->cap = 2 * (have + n + 1);
              r->buf = (char *) realloc (r->buf, r->cap + 1);
            }
          memcpy (r->buf + have, base, n);
          have += n;
          r->pos += n + (nl ? 1 : 0);
          if (nl)
            {
              r->buf[have] = '\0';
              if (len_out)
                *len_out = have;
              return r->buf;
            }
        }
      if (r->eof)
        return NULL;            /* an unterminated last line is dropped, pemapper.c:2466-2481 */
      /* the block is used up: hand it back and take the next one */
      pthread_mutex_lock (&r->mu);
      if (r->len > 0)
        {
          r->head = (r->head + 1) % LR_RING;
          r->count--;
          r->len = r->pos = 0;
          pthread_cond_broadcast (&r->cv);
        }
      while (r->count == 0 && !r->done)
        pthread_cond_wait (&r->cv, &r->mu);
      if (r->count == 0)
        r->eof = 1;
      else
        {
          r->len = (size_t) r->ring_len[r->head];
          r->pos = 0;
        }
      pthread_mutex_unlock (&r->mu);
    }
}

/* ---- one mate file's share of a batch: the records of the reference's read loop (pemapper.c:649-748: the first record is the line
        after the first line; after a sequence two lines are skipped, then lines up to one that starts with '@', and the line after it
        is the next sequence), trimmed (pemapper_tsw.c:693-704), copied into the batch's rows.  The two mate files are scanned by two
        threads side by side; the main loop takes the shorter of the two counts, as the reference's loop ends with the first file
        that runs out. */
typedef struct
{
  lreader *in;
  char *rows;
  int *lens;
  int want;                     /* records to take at most */
  int trim_s, trim_e;
  int first_mate;               /* the `sl1 <= 12` end-of-input rule looks at the first mate only (pemapper.c:663) */
  int started;                  /* the stream's first record has been taken */
  int got;
  int end;                      /* 0 the batch is full, 1 the stream ended, 2 a first-mate read of 12 bases or fewer, 3 a length out of range */
  int bad_len;
} fill_job;

static void *
fill_rows (void *arg)
{
  fill_job *j = (fill_job *) arg;
  lreader *r = j->in;
  j->got = 0;
  j->end = 0;
  while (j->got < j->want)
    {
      char *s;
      size_t full = 0;
      if (!j->started)
        {
          lr_gets (r, NULL);
          s = lr_gets (r, &full);
          j->started = 1;
        }
      else
        {
          lr_gets (r, NULL);
          lr_gets (r, NULL);
          s = lr_gets (r, &full);
          int not_there = 1;
          while (s != NULL && not_there)
            {
              if (s[0] == '@')
                not_there = 0;
              s = lr_gets (r, &full);
            }
          if (not_there)
            s = NULL;
        }
      if (s == NULL)
        {
          j->end = 1;
          break;
        }
      int sl = (int) full - j->trim_s;
      s += (sl >= 0) ? j->trim_s : (int) full;
      sl -= j->trim_e;
      if (sl < 0)
        sl = 0;
      if (j->first_mate && sl <= 12)
        {
          j->end = 2;
          break;
        }
      if (sl > PEMAP_MAX_READ || sl < PEMAP_MIN_READ)
        {
          j->end = 3;
          j->bad_len = sl;
          break;
        }
      memcpy (j->rows + (size_t) j->got * ROW_STRIDE, s, (size_t) sl);
      j->lens[j->got] = sl;
      j->got++;
    }
  return NULL;
}

/* find_chrom, pemapper.c:2168-2186, on the real (un-compressed) contig starts for the indel table (856) */
static int
find_chrom (const uint32_t * pos, int n, int first, int last, int try, uint32_t this)
{
  if (first == last)
    return first;
  uint32_t a = (try >= 0 && try <= n) ? pos[try] : 0xFFFFFFFFu;
  uint32_t b = (try + 1 >= 0 && try + 1 <= n) ? pos[try + 1] : 0xFFFFFFFFu;
  if (a <= this && b >= this)
    return try;
  if (a > this)
    last = try - 1;
  else
    first = try + 1;
  try = (last + first) / 2;
  return find_chrom (pos, n, first, last, try, this);
}

/* ---- insertion strings collected from the device log */
typedef struct
{
  uint32_t pos;
  uint32_t off;
  int len;
} ins_rec;
static ins_rec *g_ins;
static size_t g_nins, g_capins;
static char *g_inschars;
static size_t g_nchars, g_capchars;

static void
ins_cb (void *user, uint32_t pos, const char *seq, int len)
{
  (void) user;
  if (g_nins == g_capins)
    {
      g_capins = g_capins ? 2 * g_capins : 4096;
      g_ins = (ins_rec *) realloc (g_ins, g_capins * sizeof (ins_rec));
    }
  if (g_nchars + (size_t) len + 1 > g_capchars)
    {
      g_capchars = g_capchars ? 2 * g_capchars + (size_t) len : 1 << 16;
      g_inschars = (char *) realloc (g_inschars, g_capchars);
    }
  g_ins[g_nins].pos = pos;
  g_ins[g_nins].off = (uint32_t) g_nchars;
  g_ins[g_nins].len = len;
  g_nins++;
  memcpy (g_inschars + g_nchars, seq, (size_t) len);
  g_inschars[g_nchars + len] = 0;
  g_nchars += (size_t) len + 1;
}

static int
cmp_ins (const void *a, const void *b)
{
  const ins_rec *x = (const ins_rec *) a, *y = (const ins_rec *) b;
  if (x->pos != y->pos)
    return x->pos < y->pos ? -1 : 1;
  return x->off < y->off ? -1 : (x->off > y->off);
}

typedef struct
{
  pemap_dev *dev;
  const char *genome;
  uint64_t gsize;
  uint32_t *real_starts;        /* contig starts in .seq coordinates */
  char **contig_names;
  int n_contigs;
  int paired;
  int io_threads;
  char mate_names[9][80];
} ctx_t;

#pragma pack(push, 1)
typedef struct
{
  uint32_t pos;
  uint16_t c[6];
} pile_rec;
#pragma pack(pop)

/* the final genome walk and the three writers, pemapper.c:788-900 / pemapper_tsw.c dump_output 848-965 */
static void
dump_output (ctx_t * c, const char *basename, long tot_pairs)
{
  char path[4200];
  long S[13];
  ck (c->dev, pemap_dev_summary (c->dev, S));
  const long total_reads = S[0], total_bases = S[1], total_dist = S[2], no_dists = S[3];
  const long *mate_counts = &S[4];
  snprintf (path, sizeof path, "%s.summary.txt", basename);
  FILE *summaryfile = fopen (path, "w");
  if (!summaryfile)
    die ("\n Can not open file %s for writing", path);
  snprintf (path, sizeof path, "%s.pileup.gz", basename);
  pgz pileupfile;
  if (pgz_open (&pileupfile, path, c->io_threads))
    die ("\n Can not open file %s for writing", path);
  snprintf (path, sizeof path, "%s.indel.txt.gz", basename);
  gzFile indelfile = gzopen (path, "w");
  if (!indelfile)
    die ("\n Can not open file %s for writing", path);
  gzbuffer (indelfile, 33554432);

  if (total_bases <= 0)
    {
      fprintf (summaryfile, "\n================================================================");
      fprintf (summaryfile, "\n================= Summary ======================================");
      fprintf (summaryfile, "\n================================================================");
      fprintf (summaryfile, "\n================================================================");
      fprintf (summaryfile,
               "\n\nTotal Number of Mapping reads of Any Kind\t0\tWith average Length\t0\tAverage Depth\t0\tAverage Insert Size\t0");
      fprintf (summaryfile, "\n\nMapping Type\tCount\tFraction");
      fprintf (summaryfile, "\nAll\t%ld\t1", tot_pairs);
      for (int i = 0; i <= 8; i++)
        if (strstr (c->mate_names[i], "Not Used") == NULL)
          fprintf (summaryfile, "\n%s\t%ld\t%g", c->mate_names[i], mate_counts[i], (double) mate_counts[i] / (double) tot_pairs);
      fprintf (summaryfile, "\n");
      fclose (summaryfile);
      pgz_close (&pileupfile);
      gzclose (indelfile);
      return;
    }
  double avg_readlen = (double) total_bases;
  if (total_reads > 0)
    avg_readlen /= (double) total_reads;
  double avg_dist = (double) total_dist;
  if (no_dists > 0)
    avg_dist /= (double) no_dists;

  /* insertion strings, grouped by site */
  g_nins = 0;
  g_nchars = 0;
  ck (c->dev, pemap_dev_fetch_pileup (c->dev, NULL, ins_cb, NULL));
  qsort (g_ins, g_nins, sizeof (ins_rec), cmp_ins);

  gzprintf (indelfile,
            "Fragment\tPositions\tReference Base\tTotal Coverage\tReference Reads\tNo Deletions\tNo Insertions\tInsertion Sequence");
  const uint64_t chunk = 1ull << 26;
  pile_rec *recs = (pile_rec *) malloc (chunk * sizeof (pile_rec));
  size_t ip = 0;
  for (uint64_t first = 0; first < c->gsize; first += chunk)
    {
      uint64_t cnt = c->gsize - first < chunk ? c->gsize - first : chunk, n = 0;
      ck (c->dev, pemap_dev_fetch_records (c->dev, first, cnt, recs, chunk, &n));
      if (n && pgz_write (&pileupfile, recs, (size_t) n * sizeof (pile_rec)))
        die ("\n Can not write %s", "the pileup");
      for (uint64_t r = 0; r < n; r++)
        if (recs[r].c[5] > 0)
          {
            const uint32_t pos = recs[r].pos;
            const char ref = c->genome[pos];
            int tot_c = 0;
            for (int k = 0; k < 6; k++)
              tot_c += recs[r].c[k];
            int ref_reads = ref == 'A' ? recs[r].c[0] : ref == 'C' ? recs[r].c[1] : ref == 'G' ? recs[r].c[2] : recs[r].c[3];
            int which = find_chrom (c->real_starts, c->n_contigs, 0, c->n_contigs - 1, 7, pos);
            int contig_pos = 1 + (int) (pos - c->real_starts[which]);
            gzprintf (indelfile, "\n%s\t%d\t%c\t%d\t%d\t%d\t%d", c->contig_names[which], contig_pos, ref, tot_c, ref_reads,
                      recs[r].c[4], recs[r].c[5]);
            while (ip < g_nins && g_ins[ip].pos < pos)
              ip++;
            /* the counter is a wrapping u16, the strings are all kept: print as many as the counter says */
            for (int j = 0; j < recs[r].c[5] && ip < g_nins && g_ins[ip].pos == pos; j++, ip++)
              gzprintf (indelfile, "\t%s", g_inschars + g_ins[ip].off);
          }
    }
  free (recs);
  if (pgz_close (&pileupfile))
    die ("\n Can not write %s", "the pileup");
  gzclose (indelfile);
  double avg_reads = (double) total_bases / (double) c->gsize;
  fprintf (summaryfile, "\n================================================================");
  fprintf (summaryfile, "\n================= Summary ======================================");
  fprintf (summaryfile, "\n================================================================");
  fprintf (summaryfile, "\n================================================================");
  fprintf (summaryfile,
           "\n\nTotal Number of Mapping reads of Any Kind\t%ld\tWith average Length\t%g\tAverage Depth\t%g\tAverage Insert Size\t%g",
           total_reads, avg_readlen, avg_reads, avg_dist);
  fprintf (summaryfile, "\n\nMapping Type\tCount\tFraction");
  fprintf (summaryfile, "\nAll\t%ld\t1", tot_pairs);
  for (int i = 0; i <= 8; i++)
    if (strstr (c->mate_names[i], "Not Used") == NULL)
      fprintf (summaryfile, "\n%s\t%ld\t%g", c->mate_names[i], mate_counts[i], (double) mate_counts[i] / (double) tot_pairs);
  fprintf (summaryfile, "\n");
  fclose (summaryfile);
  ck (c->dev, pemap_dev_reset_pileup (c->dev));  /* counters and totals start over for the next output name (tsw 917, 948-954) */
}

static int
read_name_list (const char *path, char **names, char **outs)
{
  FILE *f = fopen (path, "r");
  if (!f)
    die ("\n Can not open file %s for reading", path);
  char line[1100];
  int n = 0;
  while (n < MAX_FILES && fgets (line, 1023, f))
    {
      char *tok = strtok (line, "\t \n");
      if (!tok || strlen (tok) <= 2)
        break;                  /* pemapper.c:258: a name of <= 2 characters ends the list */
      names[n] = strdup (tok);
      char *t2 = strtok (NULL, "\t \n");
      outs[n] = t2 ? strdup (t2) : NULL;
      n++;
    }
  fclose (f);
  return n;
}

/* ---- the files of an output set are read and mapped by a few workers side by side (the reference's usage is array mode over many
        file pairs, pemapper.c:307-348, map_directory_array.pl:92-100): a gz member inflates on one core at ~2 M reads per second,
        the device maps thirty times that.  A worker owns two sets of batch buffers -- one is filled from its file's streams while
        the GPU maps the other (the reference's reader thread fills a free PTHREAD_DATA_NODE while its workers map the others,
        pemapper.c:663-703), pinned once like pd_node_alloc -- and hands its batches to the ONE device object (submit / wait are
        made for several threads); every file's coordinates go to its own .mfile, pileup counters and summary are sums. */
typedef struct
{
  pemap_dev *dev;
  int paired, trim_s, trim_e, batch_pairs;
  long max_reads;
  char **names1, **names2;
  int next, last;               /* files [next, last) of the output set are still to be taken */
  long tot_pairs;
  pthread_mutex_t mu;
} file_pool;

typedef struct
{
  file_pool *P;
  char *r1s[2], *r2s[2];
  int *l1s[2], *l2s[2], *mts[2];
} file_worker;

static void
worker_alloc (file_worker * w, file_pool * P)
{
  w->P = P;
  for (int k = 0; k < 2; k++)
    {
      w->r1s[k] = (char *) malloc ((size_t) P->batch_pairs * ROW_STRIDE);
      w->r2s[k] = P->paired ? (char *) malloc ((size_t) P->batch_pairs * ROW_STRIDE) : NULL;
      w->l1s[k] = (int *) malloc (sizeof (int) * (size_t) P->batch_pairs);
      w->l2s[k] = (int *) malloc (sizeof (int) * (size_t) P->batch_pairs);
      w->mts[k] = (int *) malloc (sizeof (int) * (size_t) P->batch_pairs);
      if (!w->r1s[k] || (P->paired && !w->r2s[k]) || !w->l1s[k] || !w->l2s[k] || !w->mts[k])
        die ("\n pemapper_hip: out of memory for %s", "the batch buffers");
      (void) pemap_dev_pin_host (P->dev, w->r1s[k], (uint64_t) P->batch_pairs * ROW_STRIDE);   /* (a refusal only means staged copies) */
      if (P->paired)
        (void) pemap_dev_pin_host (P->dev, w->r2s[k], (uint64_t) P->batch_pairs * ROW_STRIDE);
    }
}

static void
map_one_file (file_worker * w, int iter)
{
  file_pool *P = w->P;
  pemap_dev *dev = P->dev;
  const int paired = P->paired;
  char path[4200];
  int cur_set = 0, have_pending = 0;
  uint64_t pending = 0;
  char *r1 = w->r1s[0], *r2 = w->r2s[0];
  int *l1 = w->l1s[0], *l2 = w->l2s[0], *mt = w->mts[0];
  char **r1s = w->r1s, **r2s = w->r2s;
  int **l1s = w->l1s, **l2s = w->l2s, **mts = w->mts;
    lreader in1, in2;
    lr_open (&in1, P->names1[iter]);
    if (paired)
      lr_open (&in2, P->names2[iter]);
    size_t cap = 1 << 20, current_read = 0;
    uint32_t *maps1 = (uint32_t *) calloc (cap, sizeof (uint32_t)), *maps2 = paired ? (uint32_t *) calloc (cap, sizeof (uint32_t)) : NULL;
    struct timespec ts0, ts1;
    clock_gettime (CLOCK_MONOTONIC, &ts0);
    fill_job j1, j2;
    memset (&j1, 0, sizeof j1);
    memset (&j2, 0, sizeof j2);
    j1.in = &in1;
    j1.trim_s = P->trim_s;
    j1.trim_e = P->trim_e;
    j1.first_mate = 1;
    j2 = j1;
    j2.in = &in2;
    j2.first_mate = 0;
    int not_done = 1;
    int nb = 0;
    printf ("\n Ready to map \n");
    while (not_done)
      {
        long want = P->batch_pairs;
        if ((long) current_read + want > P->max_reads)
          want = P->max_reads - (long) current_read;
        j1.rows = r1;
        j1.lens = l1;
        j1.want = (int) want;
        j2.rows = r2;
        j2.lens = l2;
        j2.want = (int) want;
        pthread_t t2;
        int threaded = 0;
        if (paired)
          threaded = pthread_create (&t2, NULL, fill_rows, &j2) == 0;
        fill_rows (&j1);
        if (paired)
          {
            if (threaded)
              pthread_join (t2, NULL);
            else
              fill_rows (&j2);
          }
        /* the reference's loop checks the first mate's length for the end-of-input rule before either length for the range, and
           stops at the first record either file cannot supply */
        nb = j1.got;
        if (paired && j2.got < nb)
          nb = j2.got;
        if (j1.end == 3 && j1.got == nb)
          {
            printf ("\n Read %ld of %s has length %d: supported range is %d..%d \n", (long) current_read + nb, P->names1[iter], j1.bad_len, PEMAP_MIN_READ,
                    PEMAP_MAX_READ);
            exit (1);
          }
        if (paired && j2.end == 3 && j2.got == nb && !(j1.end == 2 && j1.got == nb))
          {
            printf ("\n Read %ld of %s has length %d: supported range is %d..%d \n", (long) current_read + nb, P->names2[iter], j2.bad_len, PEMAP_MIN_READ,
                    PEMAP_MAX_READ);
            exit (1);
          }
        current_read += (size_t) nb;
        if (nb < want || (long) current_read >= P->max_reads)
          not_done = 0;
        if (nb > 0)
          {
            if (current_read > cap)
              {
                /* the batch in flight writes into the arrays about to move */
                if (have_pending)
                  ck (dev, pemap_dev_wait_batch (dev, pending));
                have_pending = 0;
                size_t ncap = cap;
                while (ncap < current_read)
                  ncap *= 2;
                maps1 = (uint32_t *) realloc (maps1, ncap * sizeof (uint32_t));
                if (paired)
                  maps2 = (uint32_t *) realloc (maps2, ncap * sizeof (uint32_t));
                cap = ncap;
              }
            uint64_t ticket = 0;
            ck (dev, pemap_dev_submit_batch (dev, r1, l1, r2, l2, nb, ROW_STRIDE, maps1 + (current_read - (size_t) nb),
                                             paired ? maps2 + (current_read - (size_t) nb) : NULL, mt, &ticket));
            /* the other set's batch must be back before that set is filled again */
            if (have_pending)
              ck (dev, pemap_dev_wait_batch (dev, pending));
            pending = ticket;
            have_pending = 1;
            cur_set ^= 1;
            r1 = r1s[cur_set];
            r2 = r2s[cur_set];
            l1 = l1s[cur_set];
            l2 = l2s[cur_set];
            mt = mts[cur_set];
            printf ("\n We have read %ld reads \n\n", (long) current_read);
            nb = 0;
          }
      }
    if (have_pending)
      ck (dev, pemap_dev_wait_batch (dev, pending));
    have_pending = 0;
    if (nb > 0)               /* loop left through the length test */
      {
        if (current_read > cap)
          {
            maps1 = (uint32_t *) realloc (maps1, current_read * sizeof (uint32_t));
            if (paired)
              maps2 = (uint32_t *) realloc (maps2, current_read * sizeof (uint32_t));
          }
        ck (dev, pemap_dev_map_batch (dev, r1, l1, r2, l2, nb, ROW_STRIDE, maps1 + (current_read - (size_t) nb),
                                      paired ? maps2 + (current_read - (size_t) nb) : NULL, mt));
      }
    clock_gettime (CLOCK_MONOTONIC, &ts1);
    {
      const double sec = (double) (ts1.tv_sec - ts0.tv_sec) + 1e-9 * (double) (ts1.tv_nsec - ts0.tv_nsec);
      printf ("\n pemapper_hip: %ld %s read and mapped in %.3f s (%.2f M reads/s, input parsing included) \n", (long) current_read,
              paired ? "pairs" : "reads", sec, (paired ? 2.0 : 1.0) * (double) current_read / (sec > 0 ? sec : 1) / 1e6);
    }
    printf ("\n Made it out alive, and have started cleanup \n\n");
    snprintf (path, sizeof path, "%s.mfile", P->names1[iter]);
    FILE *m = fopen (path, "wb");
    if (!m)
      die ("\n Can not open file %s", path);
    fwrite (maps1, sizeof (uint32_t), current_read, m);
    fclose (m);
    if (paired)
      {
        snprintf (path, sizeof path, "%s.mfile", P->names2[iter]);
        m = fopen (path, "wb");
        if (!m)
          die ("\n Can not open file %s", path);
        fwrite (maps2, sizeof (uint32_t), current_read, m);
        fclose (m);
        lr_close (&in2);
      }
    lr_close (&in1);
    free (maps1);
    free (maps2);
  pthread_mutex_lock (&P->mu);
  P->tot_pairs += (long) current_read;
  pthread_mutex_unlock (&P->mu);
}

static void *
file_worker_main (void *arg)
{
  file_worker *w = (file_worker *) arg;
  file_pool *P = w->P;
  for (;;)
    {
      pthread_mutex_lock (&P->mu);
      const int iter = P->next < P->last ? P->next++ : -1;
      pthread_mutex_unlock (&P->mu);
      if (iter < 0)
        return NULL;
      map_one_file (w, iter);
    }
}

int
main (int argc, char *argv[])
{
  if (argc < 4)
    die ("\nUsage: %s out_file sdx_file paired_or_single_or_array[p,s,pa,ps] file1 [file2] [max_dist] [min_dist] is_bisulfite[y,n] min_match_percentage max_threads max_reads [trim_from_start trim_from_end]", argv[0]);
  const char c_end = (char) toupper (argv[3][0]);
  const char c_array = (char) toupper (argv[3][1]);
  int paired, min_dist = 0, max_dist = 0, bis = 0, max_threads, trim_s = 0, trim_e = 0;
  double min_align;
  long max_reads;
  const char *f1arg, *f2arg = NULL;
  if (c_end == 'S')
    {
      if (argc != 9 && argc != 11)
        die ("\nUsage: %s out_file sdx_file [s,sa] file1 is_bisulfite[y,n] min_match_percentage max_threads max_reads [trim_from_start trim_from_end]", argv[0]);
      paired = 0;
      f1arg = argv[4];
      bis = (strchr (argv[5], 'Y') || strchr (argv[5], 'y'));
      min_align = atof (argv[6]);
      max_threads = atoi (argv[7]);
      max_reads = (long) atoi (argv[8]);
      if (argc == 11)
        {
          trim_s = atoi (argv[9]);
          trim_e = atoi (argv[10]);
        }
    }
  else if (c_end == 'P')
    {
      if (argc != 12 && argc != 14)
        die ("\nUsage: %s out_file sdx_file [p,pa] file1 file2 max_dist min_dist is_bisulfite[y,n] min_match_percentage max_threads max_reads [trim_from_start trim_from_end]", argv[0]);
      paired = 1;
      f1arg = argv[4];
      f2arg = argv[5];
      max_dist = atoi (argv[6]);
      min_dist = atoi (argv[7]);
      bis = (strchr (argv[8], 'Y') || strchr (argv[8], 'y'));
      min_align = atof (argv[9]);
      max_threads = atoi (argv[10]);
      max_reads = atol (argv[11]);
      if (argc == 14)
        {
          trim_s = atoi (argv[12]);
          trim_e = atoi (argv[13]);
        }
    }
  else
    die ("\nUsage: %s out_file sdx_file paired_or_single_or_array[p,s,pa,ps] file1 [file2] [max_dist] [min_dist] is_bisulfite[y,n] min_match_percentage max_threads max_reads", argv[0]);
  if ((max_threads < 2) || (max_threads > 10000))
    {
      printf ("\n Max_threads is not a sensible number (2,10000).  You gave %d \n", max_threads);
      exit (1);
    }
  char **names1 = (char **) calloc (MAX_FILES + 1, sizeof (char *));
  char **names2 = (char **) calloc (MAX_FILES + 1, sizeof (char *));
  char **outs = (char **) calloc (MAX_FILES + 1, sizeof (char *));
  char **outs2 = (char **) calloc (MAX_FILES + 1, sizeof (char *));
  int file_num = 1;
  if (c_array == 'A')
    {
      file_num = read_name_list (f1arg, names1, outs);
      if (paired && read_name_list (f2arg, names2, outs2) != file_num)
        die ("\n Mismatch in number of files in the two arrays %s", "");
      if (file_num >= MAX_FILES)
        die ("\n Too many files found in array... 2000 is the limit %s", "");
    }
  else
    {
      names1[0] = strdup (f1arg);
      if (paired)
        names2[0] = strdup (f2arg);
    }

  /* ---- .sdx: count, then "len-15 <tab> name" per contig, then idepth (pemapper.c:394-448) */
  char sdxname[1024], path[1100], line[1100];
  strncpy (sdxname, argv[2], 1000);
  sdxname[1000] = 0;
  FILE *sfile = fopen (sdxname, "r");
  if (!sfile)
    die ("\n Can not open file %s", sdxname);
  if (strstr (sdxname, ".sdx") != NULL)
    for (int i = (int) strlen (sdxname) - 1; i > 0; i--)
      if (sdxname[i] == '.')
        {
          sdxname[i] = '\0';
          i = 0;
        }
  if (!fgets (line, 256, sfile))
    die ("\n Empty file %s", argv[2]);
  const int n_contigs = atoi (line);
  if (n_contigs < 1)
    die ("\n No contigs in %s", argv[2]);
  ctx_t c;
  memset (&c, 0, sizeof c);
  c.n_contigs = n_contigs;
  c.paired = paired;
  c.contig_names = (char **) calloc ((size_t) n_contigs + 1, sizeof (char *));
  uint32_t *contig_len = (uint32_t *) calloc ((size_t) n_contigs + 1, sizeof (uint32_t));
  uint32_t *contig_starts = (uint32_t *) calloc ((size_t) n_contigs + 2, sizeof (uint32_t));
  c.real_starts = (uint32_t *) calloc ((size_t) n_contigs + 2, sizeof (uint32_t));
  for (int i = 0; i < n_contigs; i++)
    {
      if (!fgets (line, 1024, sfile))
        die ("\n Truncated file %s", argv[2]);
      char *tok = strtok (line, "\t \n");
      contig_len[i] = (uint32_t) atoi (tok) + 15u;
      tok = strtok (NULL, "\t \n");
      c.contig_names[i] = strdup (tok ? tok : "");
      contig_starts[i + 1] = contig_starts[i] + (contig_len[i] - 15u);
      c.real_starts[i + 1] = c.real_starts[i] + contig_len[i];
    }
  int idepth = 16;
  if (fgets (line, 1024, sfile))
    idepth = atoi (line);
  fclose (sfile);
  c.gsize = c.real_starts[n_contigs];
  printf ("\n Genome size is %ld \n\n", (long) c.gsize);

  /* ---- genome letters */
  snprintf (path, sizeof path, "%s.seq", sdxname);
  gzFile reffile = gzopen (path, "r");
  if (!reffile)
    die ("\n Can not open file %s for reading", path);
  gzbuffer (reffile, 33554432);
  char *genome = (char *) malloc (c.gsize + 1);
  printf ("\n About to read genome \n\n");
  for (uint64_t got = 0; got < c.gsize;)
    {
      unsigned want = (unsigned) ((c.gsize - got) < (1u << 30) ? (c.gsize - got) : (1u << 30));
      int r = gzread (reffile, genome + got, want);
      if (r <= 0)
        die ("\n Short read on %s", path);
      got += (uint64_t) r;
    }
  gzclose (reffile);
  c.genome = genome;

  pemap_dev *dev = NULL;
  const char *devs = getenv ("PEMAP_DEVICE");
  if (pemap_dev_create (&dev, devs ? atoi (devs) : 0))
    {
      printf ("\n pemap_hip: %s\n", pemap_dev_last_error (NULL));
      exit (1);
    }
  c.dev = dev;
  {
    long ncpu = sysconf (_SC_NPROCESSORS_ONLN);
    c.io_threads = max_threads - 1;     /* one of the reference's threads is its reader (pemapper.c:360-365) */
    if (ncpu > 0 && c.io_threads > (int) ncpu)
      c.io_threads = (int) ncpu;
    if (c.io_threads > 192)
      c.io_threads = 192;
  }
  printf ("\n About to read kmers index \n\n");
  const char *from_files = getenv ("PEMAP_INDEX_FROM_FILES");
  if (from_files && atoi (from_files))
    {
      /* init_index_buffer, pemapper.c:2129-2155 */
      const uint64_t NI = (1ull << 32) + 1ull;
      uint32_t *pos_index = (uint32_t *) malloc (NI * sizeof (uint32_t));
      snprintf (path, sizeof path, "%s.idx", sdxname);
      gzFile ifile = gzopen (path, "r");
      if (!pos_index || !ifile)
        die ("\nCould Not Open file %s", path);
      gzbuffer (ifile, 33554432);
      for (uint64_t got = 0; got < NI * 4;)
        {
          unsigned want = (unsigned) ((NI * 4 - got) < (1u << 30) ? (NI * 4 - got) : (1u << 30));
          int r = gzread (ifile, (char *) pos_index + got, want);
          if (r <= 0)
            die ("\n Short read on %s", path);
          got += (uint64_t) r;
        }
      gzclose (ifile);
      uint64_t n_mers = pos_index[NI - 1];
      uint32_t *mers = (uint32_t *) malloc ((n_mers + 1) * sizeof (uint32_t));
      snprintf (path, sizeof path, "%s.mdx", sdxname);
      FILE *mf = fopen (path, "r");
      if (!mf || fread (mers, sizeof (uint32_t), n_mers, mf) != n_mers)
        die ("\nCould Not read file %s", path);
      fclose (mf);
      ck (dev, pemap_dev_load_index (dev, pos_index, mers, n_mers, genome, c.gsize, contig_starts, n_contigs, idepth));
      free (pos_index);
      free (mers);
    }
  else
    ck (dev, pemap_dev_build_index (dev, genome, c.gsize, contig_len, n_contigs, bis));
  ck (dev, pemap_dev_set_params (dev, paired, min_dist, max_dist, min_align, bis));

  if (paired)
    {
      const char *nm[9] = { "Unique Mate-Paired", "Unique Mate-Paired with slip", "Unique Single End", "Unique Mis-size",
        "Non-Unique Mate-Paired", "Non-Unique Mis-size", "Fragment Mismatch", "Non-unique with no map", "Neither Map"
      };
      for (int i = 0; i < 9; i++)
        strcpy (c.mate_names[i], nm[i]);
    }
  else
    {
      const char *nm[9] = { "Not Used", "Not Used", "Unique Mapping", "Not Used", "Not Used", "Not Used", "Not Used",
        "Non-Unique Mapping, discarded", "No mapping reaches threshold"
      };
      for (int i = 0; i < 9; i++)
        strcpy (c.mate_names[i], nm[i]);
    }

  char basename[1024];
  snprintf (basename, sizeof basename, "%s", argv[1]);
  /* workers: a file pair keeps ~4 host threads busy (two inflate threads, two line scans); as many files at a time as a third of the threads (<= the host's CPUs), at most 64 */
  int n_workers = c.io_threads / 3;
  if (n_workers > 64)
    n_workers = 64;
  if (n_workers > file_num)
    n_workers = file_num;
  if (n_workers < 1)
    n_workers = 1;
  {
    const char *e = getenv ("PEMAPPER_FILE_WORKERS");
    if (e && atoi (e) >= 1 && atoi (e) <= 64)
      n_workers = atoi (e) < file_num ? atoi (e) : file_num;
  }
  file_pool pool;
  memset (&pool, 0, sizeof pool);
  pool.dev = dev;
  pool.paired = paired;
  pool.trim_s = trim_s;
  pool.trim_e = trim_e;
  pool.max_reads = max_reads;
  pool.names1 = names1;
  pool.names2 = names2;
  /* one file at a time: batches of 2^20 pairs; several: 2^18, the device pipeline's own chunk (three batches are in flight at most) */
  pool.batch_pairs = n_workers > 1 ? BATCH_PAIRS / 4 : BATCH_PAIRS;
  pthread_mutex_init (&pool.mu, NULL);
  file_worker *workers = (file_worker *) calloc ((size_t) n_workers, sizeof (file_worker));
  for (int k = 0; k < n_workers; k++)
    worker_alloc (&workers[k], &pool);
  printf ("\n About to start mapping everything \n\n");
  for (int iter = 0; iter < file_num;)
    {
      printf ("\n About to open new set of files \n");
      /* pemapper_tsw.c:636-648: a new output name closes the previous output set */
      if (outs[iter] && strcmp (basename, outs[iter]) != 0)
        {
          if (iter > 0)
            {
              dump_output (&c, basename, pool.tot_pairs);
              pool.tot_pairs = 0;
            }
          snprintf (basename, sizeof basename, "%s", outs[iter]);
        }
      /* the files that follow with the same output name (all of them, for the plain pemapper): one output set */
      int last = iter + 1;
      while (last < file_num && !(outs[last] && strcmp (basename, outs[last]) != 0))
        last++;
      pool.next = iter;
      pool.last = last;
      const int T = last - iter < n_workers ? last - iter : n_workers;
      struct timespec set0, set1;
      const long pairs_before = pool.tot_pairs;
      clock_gettime (CLOCK_MONOTONIC, &set0);
      pthread_t th[64];
      int started = 0;
      for (int k = 1; k < T; k++)
        {
          if (pthread_create (&th[k], NULL, file_worker_main, &workers[k]) != 0)
            break;              /* (the workers that did start take the files between them) */
          started = k;
        }
      file_worker_main (&workers[0]);
      for (int k = 1; k <= started; k++)
        pthread_join (th[k], NULL);
      clock_gettime (CLOCK_MONOTONIC, &set1);
      if (last - iter > 1)
        {
          const double sec = (double) (set1.tv_sec - set0.tv_sec) + 1e-9 * (double) (set1.tv_nsec - set0.tv_nsec);
          printf ("\n pemapper_hip: %d files by %d workers: %ld %s read and mapped in %.3f s (%.2f M reads/s, input parsing included) \n", last - iter, started + 1,
                  pool.tot_pairs - pairs_before, paired ? "pairs" : "reads", sec,
                  (paired ? 2.0 : 1.0) * (double) (pool.tot_pairs - pairs_before) / (sec > 0 ? sec : 1) / 1e6);
        }
      iter = last;
    }
  dump_output (&c, basename, pool.tot_pairs);
  pemap_dev_destroy (dev);
  return 0;
}
