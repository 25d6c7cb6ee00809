/*
 * pemap_oracle.c -- TEST INFRASTRUCTURE ONLY (see pemap_oracle.h).
 *
 * CPU restatement of the reference PEMapper seed-and-extend loop, written against flat arrays.
 * Every function cites the reference lines it follows (paths relative to /root/reference/src).
 * The arithmetic types, comparison directions and evaluation order of the reference are kept,
 * including its quirks, because mapping coordinates and pileup counts must be bit-identical.
 *
 * Built with -ffp-contract=off.  The SW recurrence uses fp64 add/sub/compare only (no FMA shapes).
 */
#include "pemap_oracle.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <pthread.h>

#define minim(a,b) ((a<b)?a:b)  /* pemapper.c:35 */
#define maxim(a,b) ((a>b)?a:b)  /* pemapper.c:36: on ties takes b */

enum
{ UNIQUE_MATE, UNIQUE_SLIP, UNIQUE_SINGLE, UNIQUE_MIS, NON_MATE, NON_MIS, FRAG_MIS, NON_NO, NEITHER_MAP };    /* pemapper.c:37-45 */

struct ora_state
{
  ora_index idx;
  ora_params prm;
  uint16_t *counts;             /* [genome_size][6] */
  ora_ins *ins;
  long n_ins, cap_ins;
  pthread_mutex_t ins_mutex;
  long total_reads, total_bases, total_dist, no_dists, mate_counts[9];
  double match[256][256];       /* forward == reverse bonus matrix, pemapper.c:2006-2035 */
};

/* ---------------------------------------------------------------------------------------------- */
/* 2-bit packing: fill_cv_mat / convert_seq_int, pemapper.c:2375-2423.  A=0 C=1 G=2 T=3 (either case),
 * every other byte (N included) = 0; first base in the top two bits. */
static inline unsigned
cv (char c)
{
  switch (c)
    {
    case 'c': case 'C': return 1;
    case 'g': case 'G': return 2;
    case 't': case 'T': return 3;
    default: return 0;
    }
}

uint32_t
ora_kmer (const char *s)
{
  uint32_t m = 0;
  for (int i = 0; i < 16; i++)
    m = (m << 2) + cv (s[i]);
  return m;
}

/* fill_mers, pemapper.c:1969-2003 with the mismatch table of pemapper.c:546-565: exact k-mer first, then bytes 0..3
 * (byte 0 = last four bases), inside a byte the 2-bit fields from the low end, alternatives ascending. */
void
ora_neighbours (uint32_t kmer, uint32_t * out)
{
  int m = 0;
  out[m++] = kmer;
  for (int field = 0; field < 16; field++)
    {
      unsigned sh = 2 * field;
      uint32_t cur = (kmer >> sh) & 3u;
      uint32_t base = kmer & ~(3u << sh);
      for (uint32_t k = 0; k < 4; k++)
        if (k != cur)
          out[m++] = base + (k << sh);
    }
}

/* reverse_transcribe, pemapper.c:2303-2337 */
void
ora_revcomp (const char *in, char *out, int n)
{
  for (int i = n - 1; i > -1; i--)
    {
      char c;
      switch (in[i])
        {
        case 'A': c = 'T'; break;
        case 'C': c = 'G'; break;
        case 'G': c = 'C'; break;
        case 'T': c = 'A'; break;
        case 'W': c = 'W'; break;
        case 'S': c = 'S'; break;
        case 'K': c = 'M'; break;
        case 'M': c = 'K'; break;
        case 'Y': c = 'R'; break;
        case 'R': c = 'Y'; break;
        default: c = 'N';
        }
      *out++ = c;
    }
  *out = '\0';
}

/* get_mers, pemapper.c:2158-2165.  `which + 1` is evaluated in unsigned int, so the all-T k-mer reads
 * pos_index[0] - pos_index[0xFFFFFFFF]: a huge unsigned length that always trips too_many_spots. */
static inline const uint32_t *
get_mers (const ora_index * ix, uint32_t which, uint32_t * decode_length)
{
  if (ix->pos_index)
    {
      *decode_length = ix->pos_index[(long) (uint32_t) (which + 1)] - ix->pos_index[which];
      return &ix->mers[ix->pos_index[which]];
    }
  /* compact mode: pos_index[k] == number of indexed positions whose k-mer is < k */
  uint64_t lo = 0, hi = ix->n_ukmer;
  while (lo < hi)
    {
      uint64_t mid = (lo + hi) >> 1;
      if (ix->ukmer[mid] < which)
        lo = mid + 1;
      else
        hi = mid;
    }
  uint32_t p0 = ix->ustart[lo];         /* pos_index[which] */
  uint32_t p1;
  if (which == 0xFFFFFFFFu)
    p1 = 0;                     /* pos_index[0] */
  else if (lo < ix->n_ukmer && ix->ukmer[lo] == which)
    p1 = ix->ustart[lo + 1];
  else
    p1 = p0;
  *decode_length = p1 - p0;
  return &ix->mers[p0];
}

static int
cmp_u32 (const void *a, const void *b)  /* sort_unsigned_int, pemapper.c:2426-2435 */
{
  uint32_t x = *(const uint32_t *) a, y = *(const uint32_t *) b;
  return (x < y) ? -1 : (x > y);
}

/* find_matches, pemapper.c:2189-2289.  mers[s][0] = count, mers[s][1..] ascending positions. */
static void
find_matches (uint32_t ** mers, int max_depth, const int *offsets, int idepth, int *min_match, uint32_t * hits,
              int *hits_off, int *tot_hits, uint8_t * orient, uint8_t or)
{
  unsigned int tot_found;
  unsigned int mer_pos[64];
  unsigned int i, j, k, loop, max_off = maxim (2, idepth - 4);
  int start, end;
  unsigned int min_spots = 10000;

  for (i = 0; i <= (unsigned) max_depth; i++)
    min_spots = minim (min_spots, mers[i][0]);
  if (min_spots > ORA_MAX_HITS)
    {
      *tot_hits = 0;
      return;
    }
  for (loop = 0; loop <= (unsigned int) (1 + max_depth - (*min_match)); loop++)
    {
      start = -(offsets[loop] + (int) max_off);
      end = max_off;
      for (j = loop + 1; j <= (unsigned) max_depth; j++)
        end = maxim (end, (int) (max_off + offsets[j] - offsets[loop]));
      for (i = loop; i <= (unsigned) max_depth; i++)
        mer_pos[i] = 1;
      for (i = 1; i <= mers[loop][0]; i++)
        {
          long this_start = (long) mers[loop][i] + (long) start;
          long this_end = (long) mers[loop][i] + (long) end;
          this_start = maxim (this_start, 0);
          this_end = maxim (this_end, 0);
          for (j = loop + 1; j <= (unsigned) max_depth; j++)
            while ((mer_pos[j] < mers[j][0]) && (mers[j][mer_pos[j]] < this_start))
              mer_pos[j]++;
          tot_found = 1;
          for (j = loop + 1; j <= (unsigned) max_depth; j++)
            for (k = mer_pos[j]; (k <= mers[j][0] && mers[j][k] <= this_end); k++)
              {
                /* abs() of an unsigned difference converted to int, compared with unsigned max_off (2244) */
                int d = (int) ((mers[loop][i] - mers[j][k]) - (unsigned) (offsets[loop] - offsets[j]));
                unsigned ad = (unsigned) abs (d);
                if (ad < max_off)
                  {
                    tot_found++;
                    k = mers[j][0] + 1;
                  }
              }
          if (tot_found > (unsigned) *min_match)
            {
              *min_match = tot_found;
              *tot_hits = 0;
              hits[*tot_hits] = mers[loop][i];
              hits_off[*tot_hits] = offsets[loop];
              orient[*tot_hits] = or;
              (*tot_hits)++;
            }
          else if (tot_found == (unsigned) *min_match)
            {
              if (*tot_hits < ORA_MAX_HITS)
                {
                  int new = 1;
                  for (k = 0; k < (unsigned) *tot_hits; k++)
                    if (hits[k] - hits_off[k] == mers[loop][i] - offsets[loop])
                      {
                        k = *tot_hits;
                        new = 0;
                      }
                  if (new)
                    {
                      hits[*tot_hits] = mers[loop][i];
                      hits_off[*tot_hits] = offsets[loop];
                      orient[*tot_hits] = or;
                      (*tot_hits)++;
                    }
                }
              else
                return;
            }
        }
    }
}

/* initial_map, pemapper.c:1539-1690 */
int
ora_initial_map (const ora_index * ix, int bisulfite, const char *fwd, const char *rev, int seq_len,
                 uint32_t * match_returns, uint8_t * orient_return)
{
  int idepth = ix->idepth;
  int max_mers = ORA_TOO_MANY * 50;
  int i, j;
  int N_limit = 1 + seq_len / 10;
  int n_count = 0;
  for (i = 0; i < seq_len; i++)
    if (fwd[i] == 'N')
      n_count++;
  if (n_count >= N_limit)
    return 0;

  char seqs[2][ORA_MAX_READ + 1];
  memcpy (seqs[0], fwd, seq_len);
  seqs[0][seq_len] = 0;
  memcpy (seqs[1], rev, seq_len);
  seqs[1][seq_len] = 0;
  if (bisulfite)                /* convert_ct, pemapper.c:2292-2300 */
    for (j = 0; j < 2; j++)
      for (i = 0; i < seq_len; i++)
        if (seqs[j][i] == 'C')
          seqs[j][i] = 'T';

  int total_cuts = seq_len / idepth;
  if (seq_len % idepth == 0)
    total_cuts--;
  int offsets[64];
  offsets[0] = 0;
  i = 1;
  while (i < total_cuts)
    {
      offsets[i] = offsets[i - 1] + idepth;
      i++;
    }
  if (i == total_cuts)
    offsets[i] = seq_len - idepth;

  uint32_t *pool = (uint32_t *) malloc (sizeof (uint32_t) * 2 * (total_cuts + 1) * (size_t) (max_mers + 1));
  uint32_t *lists[2][64];
  for (int s = 0; s < 2; s++)
    for (i = 0; i <= total_cuts; i++)
      lists[s][i] = pool + ((size_t) s * (total_cuts + 1) + i) * (max_mers + 1);

  for (int s = 0; s < 2; s++)
    for (i = 0; i <= total_cuts; i++)
      {
        uint32_t nb[49];
        uint32_t this_tot = 0;
        uint32_t *L = lists[s][i];
        ora_neighbours (ora_kmer (&seqs[s][offsets[i]]), nb);
        L[0] = 0;
        for (j = 0; j <= 48; j++)
          {
            const uint32_t *tmer = get_mers (ix, nb[j], &this_tot);
            if (this_tot >= ORA_TOO_MANY)
              {
                L[0] = 0;
                j = 48;
              }
            else
              {
                memcpy (&L[L[0] + 1], tmer, this_tot * sizeof (uint32_t));
                L[0] += this_tot;
              }
          }
        if (L[0] > 1)
          qsort (&L[1], L[0], sizeof (uint32_t), cmp_u32);
      }

  int min_match = maxim (1, total_cuts);
  if (total_cuts > 4)
    min_match = (4 * (total_cuts)) / 5;
  min_match = minim (min_match, 4);

  uint32_t hits[ORA_MAX_HITS + 1];
  int hits_off[ORA_MAX_HITS + 1];
  uint8_t orient[ORA_MAX_HITS + 1];
  int tot1 = 0, tot2;
  find_matches (lists[0], total_cuts, offsets, idepth, &min_match, hits, hits_off, &tot1, orient, 0);
  tot2 = tot1;
  if (tot2 < ORA_MAX_HITS)
    find_matches (lists[1], total_cuts, offsets, idepth, &min_match, hits, hits_off, &tot2, orient, 1);
  for (i = 0; i < tot2; i++)
    {
      orient_return[i] = orient[i];
      long temp = (long) hits[i] - (long) hits_off[i];
      match_returns[i] = maxim (0, temp);
    }
  free (pool);
  return tot2;
}

/* find_chrom, pemapper.c:2168-2186, first probe 7 as at every call site (1052, 1070, 856).  The reference reads
 * pos[try], pos[try+1] beyond the table when there are 2..7 contigs (undefined there); entries past the table
 * read as 0xFFFFFFFF here, which steers the bisection back into range.  Parity is claimed for 1 or >= 8 contigs. */
static int
find_chrom_rec (const uint32_t * pos, int n, int first, int last, int try, uint32_t this)
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
  return find_chrom_rec (pos, n, first, last, try, this);
}

int
ora_find_chrom (const uint32_t * pos, int n_contigs, uint32_t x)
{
  return find_chrom_rec (pos, n_contigs, 0, n_contigs - 1, 7, x);
}

/* ---------------------------------------------------------------------------------------------- */
/* One SW problem: planes S[k][i][j], i = 0..nn (reference), j = 0..mm (read); row stride mm+1. */
typedef struct
{
  double *S;                    /* 3 planes */
  int nn, mm;
} sw_planes;

#define SP(p,k,i,j) ((p)->S[((size_t)(k) * ((p)->nn + 1) + (i)) * ((p)->mm + 1) + (j)])

static const double GO = 2.0;   /* gap_open, pemapper.c:2039 */
static const double GE = 1.0 / 36.0;    /* gap_extend, pemapper.c:2040 */

/* init_penalty_matrices, pemapper.c:2051-2095 */
static void
sw_borders (sw_planes * p)
{
  SP (p, 0, 0, 0) = 0.0;
  SP (p, 1, 0, 0) = 0.0;
  SP (p, 2, 0, 0) = -1.0 * GO;
  for (int j = 1; j <= p->mm; j++)
    SP (p, 0, 0, j) = SP (p, 1, 0, j) = SP (p, 2, 0, j) = -(GO + (double) (j - 1) * GE);
  for (int i = 1; i <= p->nn; i++)
    {
      SP (p, 0, i, 0) = SP (p, 0, 0, 0);
      SP (p, 1, i, 0) = SP (p, 1, 0, 0);
      SP (p, 2, i, 0) = SP (p, 2, 0, 0);
    }
}

/* smith_waterman_align, pemapper.c:1694-1748 */
static double
sw_align (const ora_state * st, const char *base, int nn, const char *seq, int mm, sw_planes * p, int *start)
{
  int maxi, maxj, maxk, i, j, i1, j1;
  double bump;
  maxk = 0;
  maxi = 0;
  maxj = mm;
  for (i = 1; i <= nn; i++)
    for (j = 1; j <= mm; j++)
      {
        i1 = i - 1;
        j1 = j - 1;
        SP (p, 2, i, j) = maxim (SP (p, 0, i, j1) - GO, SP (p, 2, i, j1) - GE);
        SP (p, 1, i, j) = maxim (SP (p, 0, i1, j) - GO, SP (p, 1, i1, j) - GE);
        bump = st->match[(unsigned char) base[i1]][(unsigned char) seq[j1]];
        SP (p, 0, i, j) = maxim (maxim (SP (p, 0, i1, j1) + bump, SP (p, 1, i1, j1) + bump), SP (p, 2, i1, j1) + bump);
      }
  j = mm;
  for (i = 1; i <= nn; i++)
    {
      if (SP (p, 0, i, j) > SP (p, maxk, maxi, maxj))
        {
          maxk = 0;
          maxi = i;
          maxj = j;
        }
      if (SP (p, 1, i, j) > SP (p, maxk, maxi, maxj))
        {
          maxk = 1;
          maxi = i;
          maxj = j;
        }
      if (SP (p, 2, i, j) > SP (p, maxk, maxi, maxj))
        {
          maxk = 2;
          maxi = i;
          maxj = j;
        }
    }
  start[0] = maxk;
  start[1] = maxi;
  start[2] = maxj;
  return SP (p, maxk, maxi, maxj);
}

static void
log_ins (ora_state * st, uint32_t pos, const char *ins_string, int ins_len)
{
  pthread_mutex_lock (&st->ins_mutex);
  if (st->n_ins == st->cap_ins)
    {
      st->cap_ins = st->cap_ins ? 2 * st->cap_ins : 1024;
      st->ins = (ora_ins *) realloc (st->ins, sizeof (ora_ins) * st->cap_ins);
    }
  ora_ins *e = &st->ins[st->n_ins++];
  e->pos = pos;
  e->len = (uint16_t) ins_len;
  for (int m = 0; m < ins_len; m++)     /* stored back in read order, pemapper.c:1892-1893 */
    e->seq[m] = ins_string[ins_len - (m + 1)];
  e->seq[ins_len] = 0;
  pthread_mutex_unlock (&st->ins_mutex);
}

/* smith_waterman_backtrack, pemapper.c:1752-1965.  gpos0 = index into .seq of base[0]. */
static void
sw_backtrack (ora_state * st, uint64_t gpos0, int nn, const char *seq, int mm, sw_planes * p, const int *start)
{
  double smax;
  char ins_string[ORA_MAX_READ + 1];
  int ins_len = 0;
  int maxi, maxj, maxk, i, j, k, i1 = 0, j1 = 0;
  k = start[0];
  i = start[1];
  j = start[2];
  while (i > 0 && j > 0)
    {
      i1 = i - 1;
      j1 = j - 1;
      if (k == 0)
        {
          maxi = i1;
          maxj = j1;
          maxk = 0;
          smax = SP (p, 0, i1, j1);
          if (SP (p, 1, maxi, maxj) > smax)
            {
              maxk = 1;
              smax = SP (p, maxk, maxi, maxj);
            }
          if (SP (p, 2, maxi, maxj) > smax)
            maxk = 2;
        }
      else if (k == 2)
        {
          maxk = 0;
          maxi = i;
          maxj = j1;
          smax = SP (p, maxk, maxi, maxj) - GO;
          if (SP (p, 2, maxi, maxj) - GE > smax)
            maxk = 2;
        }
      else
        {
          maxk = 0;
          maxi = i1;
          maxj = j;
          smax = SP (p, maxk, maxi, maxj) - GO;
          if (SP (p, 1, maxi, maxj) - GE > smax)
            maxk = 1;
        }
      uint16_t *c = &st->counts[(gpos0 + i1) * 6];
      if (maxi != i)
        {
          if (maxj != j)
            {
              if (seq[j1] == 'A')
                __atomic_fetch_add (&c[0], 1, __ATOMIC_RELAXED);
              else if (seq[j1] == 'T')
                __atomic_fetch_add (&c[3], 1, __ATOMIC_RELAXED);
              else if (seq[j1] == 'G')
                __atomic_fetch_add (&c[2], 1, __ATOMIC_RELAXED);
              else if (seq[j1] == 'C')
                __atomic_fetch_add (&c[1], 1, __ATOMIC_RELAXED);
            }
          else
            __atomic_fetch_add (&c[4], 1, __ATOMIC_RELAXED);
          if (ins_len > 0)
            {
              log_ins (st, (uint32_t) (gpos0 + i1), ins_string, ins_len);
              __atomic_fetch_add (&c[5], 1, __ATOMIC_RELAXED);
            }
          ins_len = 0;
        }
      else
        {
          ins_string[ins_len] = seq[j1];
          ins_len++;
        }
      i = maxi;
      j = maxj;
      k = maxk;
    }
  if (ins_len > 0 && i >= 1)    /* pemapper.c:1918-1958: attaches to base[i1] of the LAST loop iteration */
    {
      uint16_t *c = &st->counts[(gpos0 + i1) * 6];
      log_ins (st, (uint32_t) (gpos0 + i1), ins_string, ins_len);
      __atomic_fetch_add (&c[5], 1, __ATOMIC_RELAXED);
    }
}

double
ora_sw (const char *ref, int nn, const char *seq, int mm, int bisulfite, int *start3, double *planes)
{
  /* only the match table of a state is needed */
  static ora_state *cache[2];
  if (!cache[bisulfite != 0])
    {
      ora_index ix;
      memset (&ix, 0, sizeof (ix));
      ora_params pr;
      memset (&pr, 0, sizeof (pr));
      pr.bisulfite = bisulfite;
      cache[bisulfite != 0] = ora_create (&ix, &pr);
    }
  sw_planes p;
  p.nn = nn;
  p.mm = mm;
  p.S = planes ? planes : (double *) malloc (sizeof (double) * 3 * (size_t) (nn + 1) * (mm + 1));
  sw_borders (&p);
  double s = sw_align (cache[bisulfite != 0], ref, nn, seq, mm, &p, start3);
  if (!planes)
    free (p.S);
  return s;
}

/* ---------------------------------------------------------------------------------------------- */
typedef struct
{
  sw_planes pl[ORA_MAX_HITS];
  uint64_t gpos[ORA_MAX_HITS];  /* window start, index into .seq */
  int pass_len[ORA_MAX_HITS];
  uint32_t spots[ORA_MAX_HITS + 1];
  uint8_t orients[ORA_MAX_HITS + 1];
  int start_t[ORA_MAX_HITS][3];
  double smax[ORA_MAX_HITS + 1];
  int hits;
} end_ws;

static void
planes_for (sw_planes * p, int nn, int mm)
{
  size_t need = 3 * (size_t) (nn + 1) * (mm + 1);
  if (!p->S || (size_t) 3 * (p->nn + 1) * (p->mm + 1) < need)
    {
      free (p->S);
      p->S = (double *) malloc (sizeof (double) * need);
    }
  p->nn = nn;
  p->mm = mm;
  sw_borders (p);
}

/* window set-up, pemapper.c:1047-1081 */
static void
set_windows (const ora_state * st, end_ws * w, int len)
{
  const uint32_t *cs = st->idx.contig_starts;
  for (int i = 0; i < w->hits; i++)
    {
      int this_chrom = ora_find_chrom (cs, st->idx.n_contigs, w->spots[i]);
      unsigned int extra = 15 * this_chrom;
      long ttemp = maxim (0, (long) extra + (long) w->spots[i] - (long) ORA_SLOP);
      unsigned int start_match = maxim (cs[this_chrom] + extra, ttemp);
      unsigned int end_match = minim (cs[this_chrom + 1] + extra, extra + w->spots[i] + len + ORA_SLOP);
      int blen = 1 + end_match - start_match;
      w->pass_len[i] = blen;
      w->gpos[i] = start_match;
    }
}

static double
align_hit (const ora_state * st, end_ws * w, int i, char seqs[2][ORA_MAX_READ + 1], int len)
{
  planes_for (&w->pl[i], w->pass_len[i], len);
  return sw_align (st, st->idx.genome + w->gpos[i], w->pass_len[i], seqs[w->orients[i]], len, &w->pl[i], w->start_t[i]);
}

/* single-end selection, pemapper.c:1084-1174.  Returns the class; *bsm = winning hit when UNIQUE_SINGLE. */
static int
single_select (const ora_state * st, end_ws * w, char seqs[2][ORA_MAX_READ + 1], int len, int *bsm)
{
  double good_score = len * st->prm.min_align * 1.0;
  double top_score = -GO * len;
  int top_score_count = 0;
  double this_score = 0;
  for (int i = 0; i < w->hits; i++)
    {
      this_score = align_hit (st, w, i, seqs, len);
      w->smax[i] = this_score;
      if (this_score > top_score && this_score >= good_score)
        {
          top_score = this_score;
          top_score_count = 1;
          *bsm = i;
        }
      else if ((fabs (this_score - top_score) < 0.0001) && (top_score_count > 0))
        top_score_count++;
    }
  if (top_score_count == 0)
    return NEITHER_MAP;
  if (top_score_count == 1)
    return UNIQUE_SINGLE;
  return NON_NO;
}

/* find_mate_pairs, pemapper.c:1313-1536.  use1/use2 = which hit to backtrack, -1 for none. */
static int
find_mate_pairs (const ora_state * st, end_ws * w1, end_ws * w2, char s1[2][ORA_MAX_READ + 1], int l1,
                 char s2[2][ORA_MAX_READ + 1], int l3, int *use1, int *use2)
{
  int n1 = w1->hits, n2 = w2->hits;
  int perfect = 0, which1, which2, i;
  long temp_dist;
  double tot_best = -1e5;
  double *smax1 = w1->smax, *smax2 = w2->smax;
  unsigned int start_match1 = 0xFFFFFFFFu, start_match2 = 0xFFFFFFFFu;   /* overwritten before first use (inc > 0.001 always fires first) */
  *use1 = *use2 = -1;
  for (i = 0; i <= ORA_MAX_HITS; i++)
    smax1[i] = smax2[i] = -1.0;
  double good_score1 = l1 * st->prm.min_align * 1.0;
  double good_score2 = l3 * st->prm.min_align * 1.0;
  for (which1 = 0; which1 < n1; which1++)
    smax1[which1] = align_hit (st, w1, which1, s1, l1);
  for (which2 = 0; which2 < n2; which2++)
    smax2[which2] = align_hit (st, w2, which2, s2, l3);
  int slip_count = 0;
  for (which1 = 0; which1 < n1; which1++)
    if (smax1[which1] >= good_score1)
      for (which2 = 0; which2 < n2; which2++)
        if (smax2[which2] >= good_score2)
          {
            unsigned int p1 = w1->spots[which1];
            unsigned int p2 = w2->spots[which2];
            temp_dist = labs ((long) p1 - (long) p2);
            int or1 = w1->orients[which1];
            int or2 = w2->orients[which2];
            int is_perfect = ((temp_dist >= st->prm.min_dist) && (temp_dist <= st->prm.max_dist) && (or1 != or2));
            if (is_perfect)
              {
                double this_1 = smax1[which1];
                double this_2 = smax2[which2];
                double inc = smax1[which1] + smax2[which2] - tot_best;
                if (inc > 0.001)
                  {
                    perfect = 1;
                    start_match1 = which1;
                    start_match2 = which2;
                    tot_best = this_1 + this_2;
                    slip_count = 1;
                  }
                else if (inc > -0.001)
                  {
                    if ((start_match1 == (unsigned) which1) || (start_match2 == (unsigned) which2))
                      slip_count++;
                    perfect++;
                  }
              }
          }
  int exit_code = NEITHER_MAP;
  if (perfect > 0)
    {
      *use1 = start_match1;
      *use2 = start_match2;
      if (perfect == 1)
        exit_code = UNIQUE_MATE;
      else if (slip_count == perfect)
        exit_code = UNIQUE_SLIP;
      else
        {
          exit_code = NON_MATE;
          *use1 = *use2 = -1;
        }
    }
  else
    {
      int best1 = 0, best2 = 0, m1_c = 0, m2_c = 0;
      for (i = 1; i < n1; i++)
        if (smax1[i] > smax1[best1])
          {
            best1 = i;
            m1_c = 1;
          }
        else if (smax1[i] - smax1[best1] > -0.0001)
          m1_c++;
      for (i = 1; i < n2; i++)
        if (smax2[i] > smax2[best2])
          {
            best2 = i;
            m2_c = 1;
          }
        else if (smax2[i] - smax2[best1] > -0.0001)     /* sic: best1, pemapper.c:1468 */
          m2_c++;
      int ok2 = (smax2[best2] >= good_score2) && (m2_c < 2);
      if (smax1[best1] >= good_score1)
        {
          if (m1_c < 2)
            {
              *use1 = best1;
              if (ok2)
                {
                  *use2 = best2;
                  exit_code = UNIQUE_MIS;
                }
              else
                exit_code = UNIQUE_SINGLE;
            }
          else
            {
              if (ok2)
                {
                  *use2 = best2;
                  exit_code = UNIQUE_SINGLE;
                }
              else
                exit_code = NON_MIS;
            }
        }
      else
        {
          if (ok2)
            {
              *use2 = best2;
              exit_code = UNIQUE_SINGLE;
            }
          else
            exit_code = NON_MIS;
        }
    }
  return exit_code;
}

typedef struct
{
  ora_state *st;
  const char *reads1, *reads2;
  const int *len1, *len2;
  long lo, hi;
  int stride;
  uint32_t *m1, *m2;
  int *mapping_type;
  ora_end_dbg *dbg1, *dbg2;
} job_t;

static void
fill_dbg (ora_end_dbg * d, const end_ws * w, int scored)
{
  d->n_hits = w->hits;
  for (int i = 0; i < w->hits; i++)
    {
      d->spot[i] = w->spots[i];
      d->orient[i] = w->orients[i];
      d->win_start_lo[i] = (int32_t) w->gpos[i];
      d->win_len[i] = w->pass_len[i];
      d->score[i] = scored ? w->smax[i] : 0.0;
      for (int k = 0; k < 3; k++)
        d->start[i][k] = scored ? w->start_t[i][k] : 0;
    }
}

/* the per-read body of map_everything, pemapper.c:1010-1235 */
static void *
map_range (void *arg)
{
  job_t *jb = (job_t *) arg;
  ora_state *st = jb->st;
  end_ws *w1 = (end_ws *) calloc (1, sizeof (end_ws));
  end_ws *w2 = (end_ws *) calloc (1, sizeof (end_ws));
  char s1[2][ORA_MAX_READ + 1], s2[2][ORA_MAX_READ + 1];
  for (long it = jb->lo; it < jb->hi; it++)
    {
      int l1 = jb->len1[it], l3 = 0;
      int use1 = -1, use2 = -1, first_call;
      memcpy (s1[0], jb->reads1 + (size_t) it * jb->stride, l1);
      s1[0][l1] = 0;
      ora_revcomp (s1[0], s1[1], l1);
      w1->hits = ora_initial_map (&st->idx, st->prm.bisulfite, s1[0], s1[1], l1, w1->spots, w1->orients);
      w2->hits = 0;
      if (st->prm.paired)
        {
          l3 = jb->len2[it];
          memcpy (s2[0], jb->reads2 + (size_t) it * jb->stride, l3);
          s2[0][l3] = 0;
          ora_revcomp (s2[0], s2[1], l3);
          w2->hits = ora_initial_map (&st->idx, st->prm.bisulfite, s2[0], s2[1], l3, w2->spots, w2->orients);
        }
      set_windows (st, w1, l1);
      set_windows (st, w2, l3);
      int scored1 = 0, scored2 = 0;
      if (w1->hits > 0 && w2->hits == 0)
        {
          int bsm = 0;
          first_call = single_select (st, w1, s1, l1, &bsm);
          if (first_call == UNIQUE_SINGLE)
            use1 = bsm;
          scored1 = 1;
        }
      else if (w2->hits > 0 && w1->hits == 0)
        {
          int bsm = 0;
          first_call = single_select (st, w2, s2, l3, &bsm);
          if (first_call == UNIQUE_SINGLE)
            use2 = bsm;
          scored2 = 1;
        }
      else if (w1->hits > 0 && w2->hits > 0)
        {
          first_call = find_mate_pairs (st, w1, w2, s1, l1, s2, l3, &use1, &use2);
          scored1 = scored2 = 1;
        }
      else
        first_call = NEITHER_MAP;       /* pemapper.c:1186-1192 */

      uint32_t temp_spot = 0;
      if (use1 >= 0)
        {
          sw_backtrack (st, w1->gpos[use1], w1->pass_len[use1], s1[w1->orients[use1]], l1, &w1->pl[use1], w1->start_t[use1]);
          temp_spot = (uint32_t) (w1->gpos[use1] + w1->start_t[use1][1]) + 1;   /* bn[start1[1]].pos + 1, pemapper.c:1208 */
        }
      jb->m1[it] = temp_spot;
      temp_spot = 0;
      if (use2 >= 0)
        {
          sw_backtrack (st, w2->gpos[use2], w2->pass_len[use2], s2[w2->orients[use2]], l3, &w2->pl[use2], w2->start_t[use2]);
          temp_spot = (uint32_t) (w2->gpos[use2] + w2->start_t[use2][1]) + 1;
        }
      if (jb->m2)
        jb->m2[it] = temp_spot;
      jb->mapping_type[it] = first_call;
      if (jb->dbg1)
        fill_dbg (&jb->dbg1[it], w1, scored1);
      if (jb->dbg2)
        fill_dbg (&jb->dbg2[it], w2, scored2);
    }
  for (int i = 0; i < ORA_MAX_HITS; i++)
    {
      free (w1->pl[i].S);
      free (w2->pl[i].S);
    }
  free (w1);
  free (w2);
  return NULL;
}

int
ora_map_batch (ora_state * st, const char *reads1, const int *len1, const char *reads2, const int *len2, long n,
               int stride, uint32_t * m1, uint32_t * m2, int *mapping_type, ora_end_dbg * dbg1, ora_end_dbg * dbg2,
               int threads)
{
  if (threads < 1)
    threads = 1;
  if (threads > n)
    threads = n > 0 ? (int) n : 1;
  job_t *jobs = (job_t *) calloc (threads, sizeof (job_t));
  pthread_t *th = (pthread_t *) calloc (threads, sizeof (pthread_t));
  for (int t = 0; t < threads; t++)
    {
      job_t *jb = &jobs[t];
      jb->st = st;
      jb->reads1 = reads1;
      jb->reads2 = reads2;
      jb->len1 = len1;
      jb->len2 = len2;
      jb->stride = stride;
      jb->lo = n * t / threads;
      jb->hi = n * (t + 1) / threads;
      jb->m1 = m1;
      jb->m2 = m2;
      jb->mapping_type = mapping_type;
      jb->dbg1 = dbg1;
      jb->dbg2 = dbg2;
      if (threads == 1)
        map_range (jb);
      else
        pthread_create (&th[t], NULL, map_range, jb);
    }
  if (threads > 1)
    for (int t = 0; t < threads; t++)
      pthread_join (th[t], NULL);
  /* result fold, pemapper.c:1238-1265 */
  for (long j = 0; j < n; j++)
    {
      uint32_t a = m1[j], b = m2 ? m2[j] : 0;
      st->mate_counts[mapping_type[j]]++;
      if (a)
        {
          st->total_reads++;
          st->total_bases += len1[j];
          if (b)
            {
              st->total_reads++;
              st->total_bases += len2[j];
              long test = labs ((long) (uint32_t) (a - b));   /* unsigned difference widened, pemapper.c:1250 */
              if (test < st->prm.max_dist * 4)
                {
                  st->total_dist += test;
                  st->no_dists++;
                }
            }
        }
      else if (b)
        {
          st->total_reads++;
          st->total_bases += len2[j];
        }
    }
  free (jobs);
  free (th);
  return 0;
}

ora_state *
ora_create (const ora_index * idx, const ora_params * prm)
{
  ora_state *st = (ora_state *) calloc (1, sizeof (ora_state));
  st->idx = *idx;
  st->prm = *prm;
  if (idx->genome_size)
    st->counts = (uint16_t *) calloc (idx->genome_size * 6 + 6, sizeof (uint16_t));
  pthread_mutex_init (&st->ins_mutex, NULL);
  /* init_bonus_matrices, pemapper.c:2006-2035, replayed literally on its 301 x 301 table: the row fill of iteration i
   * runs BEFORE that iteration's four 'N'/'n' assignments, so row 'N' (78) is wiped at i = 78 and only regains 1.0 for
   * columns >= 78; a reference 'N' therefore matches read 'N', 'T' and lower case, but NOT read 'A', 'C', 'G', while a
   * read 'N' (column 'N') matches every reference byte. */
  {
    enum { MS = 300 };
    static double M[MS + 1][MS + 1];
    double match_bonus = 1.0;
    double mtemp = -1.0 / ((double) 3.0 * match_bonus);
    for (int i = 0; i <= MS; i++)
      {
        for (int j = 0; j <= MS; j++)
          M[i][j] = (i == j) ? match_bonus : mtemp;
        M[i]['N'] = M['N'][i] = M[i]['n'] = M['n'][i] = match_bonus;
        if (prm->bisulfite)
          M['C']['T'] = M['C']['t'] = M['c']['T'] = M['c']['t'] = match_bonus;
      }
    for (int i = 0; i < 256; i++)
      for (int j = 0; j < 256; j++)
        st->match[i][j] = M[i][j];
  }
  return st;
}

void
ora_destroy (ora_state * st)
{
  if (!st)
    return;
  free (st->counts);
  free (st->ins);
  free (st);
}

const uint16_t *
ora_counts (ora_state * st)
{
  return st->counts;
}

long
ora_n_ins (ora_state * st)
{
  return st->n_ins;
}

const ora_ins *
ora_ins_log (ora_state * st)
{
  return st->ins;
}

void
ora_summary (ora_state * st, long *out)
{
  out[0] = st->total_reads;
  out[1] = st->total_bases;
  out[2] = st->total_dist;
  out[3] = st->no_dists;
  for (int i = 0; i < 9; i++)
    out[4 + i] = st->mate_counts[i];
}
