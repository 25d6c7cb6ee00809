// pemap_kernels.hip.h -- device code of the PEMapper hot path for gfx950 (CDNA4, wave64).
//
// Written for MI355X only: 64-lane wavefronts, LDS-staged read/k-mer lists, DPP/ds_bpermute lane exchange,
// fp64 VALU for the Smith-Waterman planes (no MFMA: the recurrence is add/sub/max/compare, not a contraction).
// Compiled with -ffp-contract=off: every fp64 operation below is a single IEEE add/sub/compare, which is what
// the reference's C does on x86-64, so scores and tie-breaks are bit-identical.
//
// Reference (wingolab-org/pecaller) line numbers cited as pemapper.c:NNN refer to src/pemapper.c.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define PM_MAX_HITS 200            // max_hits, pemapper.c:162
#define PM_TOO_MANY 100            // too_many_spots, pemapper.c:163
#define PM_SLOP 10                 // MISALIGN_SLOP, pemapper.c:47
#define PM_MAX_SEG 19              // segments of a 299-base read
#define PM_SEED_CAP 1024           // positions per strand kept in LDS; larger lists go to the per-block global scratch
#define PM_SEG_LIST_MAX 4851       // 48*99 + 99 positions per segment at most (every bucket < too_many_spots)
#define PM_LPA 8                   // lanes per alignment in the SW kernels
// register budget of the SW kernels: 4 VGPRs of state per owned column (two doubles) plus temporaries
#define PM_WAVES_PER_EU(W) ((W) <= 19 ? 4 : (W) <= 26 ? 3 : 2)

struct PmIndex
{
  const uint32_t *pos_index;       // [2^32 + 1]
  const uint32_t *mers;            // [n_mers]
  const uint8_t *genome;           // [gsize]
  const uint32_t *contig_starts;   // [n_contigs + 1] compressed (len-15) prefix sums
  uint64_t n_mers;
  uint64_t gsize;
  int n_contigs;
  int idepth;
};

// counters shared by the kernels of one run (zeroed at the start of every run)
struct PmCounters
{
  unsigned int n_tasks_s;          // SW problems of read-ends with exactly one hit (scored WITH direction nibbles)
  unsigned int n_tasks_m;          // SW problems of read-ends with several hits (score only)
  unsigned int n_slots;            // direction slabs handed out
  unsigned int n_redo;             // winners of multi-hit ends: scored again with direction nibbles
  unsigned int n_wins;             // alignments to walk back
  unsigned int pad0;
  unsigned long long positions;    // P: entries copied out of .mdx
  unsigned long long cells_score;  // DP cells computed without / with direction nibbles
  unsigned long long cells_dirs;
  unsigned long long pile_incs;
  unsigned long long n_ins;
};

// insertion log cursor: survives runs until the host drains the log
struct PmInsCursor
{
  unsigned int ins_bytes;
  unsigned int ins_overflow;
};

// per read-end hit record arrays, all [n_ends][PM_MAX_HITS]
struct PmHits
{
  int *n_hits;                     // [n_ends]
  uint32_t *spot;                  // hit - offset, compressed coordinates (pemapper.c:1664-1669)
  uint32_t *gpos;                  // window start, index into .seq (pemapper.c:1055)
  int16_t *nn;                     // window length (pemapper.c:1058)
  uint8_t *orient;
  double *score;
  int16_t *sti;                    // start[1]
  uint8_t *stk;                    // start[0]
  int *slot;                       // [n_ends] direction slab of the end's traceable alignment, -1 = none
};

struct PmBatch
{
  const uint8_t *reads1, *reads2;  // stride-spaced rows
  const int *len1, *len2;
  int n;                           // pairs (or single reads)
  int stride;
  int paired;
  int n_ends;
};

struct PmParams
{
  int min_dist, max_dist;
  double min_align;
  int bisulfite;
};

// ------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned pm_code (uint8_t c, int bis)
{
  // fill_cv_mat, pemapper.c:2375-2383: c/C=1 g/G=2 t/T=3, everything else (N included) 0;
  // convert_ct (2292-2300) turns an upper-case 'C' of the seed copy into 'T' first when mapping bisulfite reads.
  if (bis && c == 'C')
    return 3;
  if (c == 'C' || c == 'c')
    return 1;
  if (c == 'G' || c == 'g')
    return 2;
  if (c == 'T' || c == 't')
    return 3;
  return 0;
}

__device__ __forceinline__ uint8_t pm_rc (uint8_t c)
{
  // reverse_transcribe, pemapper.c:2303-2337
  switch (c)
    {
    case 'A': return 'T';
    case 'C': return 'G';
    case 'G': return 'C';
    case 'T': return 'A';
    case 'W': return 'W';
    case 'S': return 'S';
    case 'K': return 'M';
    case 'M': return 'K';
    case 'Y': return 'R';
    case 'R': return 'Y';
    default: return 'N';
    }
}

__device__ __forceinline__ const uint8_t *pm_read_ptr (const PmBatch & b, int end, int *len)
{
  if (b.paired)
    {
      int pr = end >> 1;
      if (end & 1)
        {
          *len = b.len2[pr];
          return b.reads2 + (size_t) pr * b.stride;
        }
      *len = b.len1[pr];
      return b.reads1 + (size_t) pr * b.stride;
    }
  *len = b.len1[end];
  return b.reads1 + (size_t) end * b.stride;
}

// find_chrom, pemapper.c:2168-2186 with its fixed first probe 7 (call sites 1052, 1070).  Entries past the table read as
// 0xFFFFFFFF (the reference reads beyond its allocation when there are 2..7 contigs; parity is claimed for 1 or >= 8).
__device__ __forceinline__ int pm_find_chrom (const uint32_t * pos, int n, uint32_t x)
{
  int first = 0, last = n - 1, tr = 7;
  for (int it = 0; it < 64; it++)
    {
      if (first == last)
        return first;
      uint32_t a = (tr >= 0 && tr <= n) ? pos[tr] : 0xFFFFFFFFu;
      uint32_t b = (tr + 1 >= 0 && tr + 1 <= n) ? pos[tr + 1] : 0xFFFFFFFFu;
      if (a <= x && b >= x)
        return tr;
      if (a > x)
        last = tr - 1;
      else
        first = tr + 1;
      tr = (last + first) / 2;
    }
  return first < 0 ? 0 : (first > n - 1 ? n - 1 : first);
}

// j-th entry of fill_mers' list (pemapper.c:1969-2003 with the table of 546-565): 0 = the k-mer itself, then the 48
// single-substitution neighbours, 2-bit fields from the low end (last base first), alternatives ascending.
__device__ __forceinline__ uint32_t pm_neighbour (uint32_t k, int j)
{
  if (j == 0)
    return k;
  int f = (j - 1) / 3, a = (j - 1) % 3;
  unsigned sh = 2u * f;
  uint32_t cur = (k >> sh) & 3u;
  uint32_t alt = (uint32_t) a + ((uint32_t) a >= cur ? 1u : 0u);
  return (k & ~(3u << sh)) + (alt << sh);
}

// ============================================================================================================
// K1/K2: seed gather + diagonal vote.  One wave (64-thread block) per read-end, persistent over ends.
// ============================================================================================================
struct __align__ (8) PmSeedShared
{
  uint32_t raw[PM_SEED_CAP];
  uint32_t sorted[PM_SEED_CAP];
  uint32_t it_start[PM_MAX_SEG * 49];
  uint16_t it_len[PM_MAX_SEG * 49];      // 0xFFFF = bucket >= too_many_spots
  uint16_t it_off[PM_MAX_SEG * 49];
  uint32_t kmer[PM_MAX_SEG + 1];
  int seg_cnt[PM_MAX_SEG + 1];
  int seg_base[PM_MAX_SEG + 2];
  int offsets[PM_MAX_SEG + 1];
  uint32_t hits[PM_MAX_HITS];
  uint16_t hits_off[PM_MAX_HITS];
  uint8_t hits_or[PM_MAX_HITS];
  uint8_t seq[2][320];
};

// number of list elements < v  (list ascending)
__device__ __forceinline__ int pm_lower_bound (const uint32_t * lst, int n, int64_t v)
{
  int lo = 0, hi = n;
  while (lo < hi)
    {
      int mid = (lo + hi) >> 1;
      if ((int64_t) lst[mid] < v)
        lo = mid + 1;
      else
        hi = mid;
    }
  return lo;
}

// find_matches, pemapper.c:2189-2289, for one strand.  The reference walks anchors (segment `loop`, position i) in
// order and keeps a running best `min_match`; per anchor it counts the later segments that have a position within
// max_off of the anchor's diagonal.  That count (tot_found) does not depend on the walk, so 64 anchors are counted in
// parallel (binary search into each later segment's sorted list) and the walk's state machine -- reset on '>', append
// on '==' if the diagonal is new, stop when max_hits tied hits are held -- is replayed in anchor order over the lanes
// whose count can still matter.  Returns false when the reference `return`s early with a full list (2283-2284).
__device__ bool pm_find_matches (PmSeedShared & sh, const uint32_t * lists, int max_depth, int idepth, int &min_match,
                                 int &tot_hits, uint8_t orient, int lane)
{
  unsigned min_spots = 10000;
  for (int s = 0; s <= max_depth; s++)
    min_spots = min (min_spots, (unsigned) sh.seg_cnt[s]);
  if (min_spots > PM_MAX_HITS)
    {
      tot_hits = 0;                 // pemapper.c:2203-2207
      return true;
    }
  const int max_off = max (2, idepth - 4);
  for (int loop = 0; loop <= 1 + max_depth - min_match; loop++)
    {
      const int n = sh.seg_cnt[loop];
      const uint32_t *la = lists + sh.seg_base[loop];
      const int off_a = sh.offsets[loop];
      for (int i0 = 0; i0 < n; i0 += 64)
        {
          int i = i0 + lane;
          bool act = i < n;
          uint32_t m = 0;
          int tf = 0;
          if (act)
            {
              m = la[i];
              tf = 1;
              for (int j = loop + 1; j <= max_depth; j++)
                {
                  // |(m - m_jk) - (off_a - off_j)| < max_off   (pemapper.c:2244; int wrap cannot occur below 2^32 - 400 positions)
                  int64_t t = (int64_t) m - (int64_t) (off_a - sh.offsets[j]);
                  const uint32_t *lj = lists + sh.seg_base[j];
                  int nj = sh.seg_cnt[j];
                  int lo = pm_lower_bound (lj, nj, t - (max_off - 1));
                  if (lo < nj && (int64_t) lj[lo] <= t + (max_off - 1))
                    tf++;
                }
            }
          unsigned long long cand = __ballot (act && tf >= min_match);
          while (cand)
            {
              int l = __ffsll ((long long) cand) - 1;
              cand &= cand - 1;
              int tfl = __shfl (tf, l);
              uint32_t ml = __shfl (m, l);
              if (tfl > min_match)
                {
                  min_match = tfl;
                  tot_hits = 0;
                  if (lane == 0)
                    {
                      sh.hits[0] = ml;
                      sh.hits_off[0] = (uint16_t) off_a;
                      sh.hits_or[0] = orient;
                    }
                  tot_hits = 1;
                  __syncthreads ();
                }
              else if (tfl == min_match)
                {
                  if (tot_hits < PM_MAX_HITS)
                    {
                      uint32_t diag = ml - (uint32_t) off_a;      // unsigned, pemapper.c:2268
                      bool dup = false;
                      for (int k = lane; k < tot_hits; k += 64)
                        if (sh.hits[k] - (uint32_t) sh.hits_off[k] == diag)
                          dup = true;
                      if (!__any (dup))
                        {
                          if (lane == 0)
                            {
                              sh.hits[tot_hits] = ml;
                              sh.hits_off[tot_hits] = (uint16_t) off_a;
                              sh.hits_or[tot_hits] = orient;
                            }
                          tot_hits++;
                          __syncthreads ();
                        }
                    }
                  else
                    return false;
                }
            }
        }
    }
  return true;
}

__global__ __launch_bounds__ (64) void pm_seed_kernel (PmIndex ix, PmBatch b, PmParams prm, PmHits h, uint32_t * tasks_s,
                                                      uint32_t * tasks_m, PmCounters * ctr, uint32_t * gscratch)
{
  __shared__ PmSeedShared sh;
  const int lane = threadIdx.x;
  const int idepth = ix.idepth;
  uint32_t *g_raw = gscratch + (size_t) blockIdx.x * 2 * PM_MAX_SEG * PM_SEG_LIST_MAX;
  uint32_t *g_sorted = g_raw + (size_t) PM_MAX_SEG * PM_SEG_LIST_MAX;

  for (int e = blockIdx.x; e < b.n_ends; e += gridDim.x)
    {
      __syncthreads ();
      int len;
      const uint8_t *src = pm_read_ptr (b, e, &len);
      // ---- read + reverse complement into LDS; N filter (pemapper.c:1552-1559: upper-case 'N' only)
      int n_count = 0;
      for (int i = lane; i < len; i += 64)
        {
          uint8_t c = src[i];
          sh.seq[0][i] = c;
          sh.seq[1][len - 1 - i] = pm_rc (c);
          n_count += (c == 'N');
        }
      for (int o = 32; o > 0; o >>= 1)
        n_count += __shfl_xor (n_count, o);
      __syncthreads ();
      int tot = 0;
      if (n_count < 1 + len / 10)
        {
          // ---- segment offsets (pemapper.c:1573-1587)
          int total_cuts = len / idepth;
          if (len % idepth == 0)
            total_cuts--;
          if (lane <= total_cuts)
            sh.offsets[lane] = (lane < total_cuts || total_cuts == 0) ? lane * idepth : len - idepth;
          const int S = total_cuts + 1;
          int min_match = max (1, total_cuts);       // pemapper.c:1642-1645
          if (total_cuts > 4)
            min_match = (4 * total_cuts) / 5;
          min_match = min (min_match, 4);
          __syncthreads ();
          bool go_on = true;
          for (int strand = 0; strand < 2 && go_on; strand++)
            {
              // ---- 16-mers of the segments (convert_seq_int, pemapper.c:2408-2423)
              if (lane < S)
                {
                  const uint8_t *p = &sh.seq[strand][sh.offsets[lane]];
                  uint32_t k = 0;
                  for (int i = 0; i < 16; i++)
                    k = (k << 2) + pm_code (p[i], prm.bisulfite);
                  sh.kmer[lane] = k;
                }
              __syncthreads ();
              // ---- 49 bucket look-ups per segment (get_mers, pemapper.c:2158-2165; `which + 1` wraps in 32 bits)
              for (int x = lane; x < S * 49; x += 64)
                {
                  int seg = x / 49, j = x - seg * 49;
                  uint32_t nb = pm_neighbour (sh.kmer[seg], j);
                  uint32_t i0 = ix.pos_index[nb];
                  uint32_t i1 = ix.pos_index[(uint32_t) (nb + 1u)];
                  uint32_t ln = i1 - i0;
                  sh.it_start[x] = i0;
                  sh.it_len[x] = (ln >= PM_TOO_MANY) ? 0xFFFF : (uint16_t) ln;
                }
              __syncthreads ();
              // ---- a segment with any bucket >= too_many_spots is emptied (pemapper.c:1602-1606)
              if (lane < S)
                {
                  int sum = 0;
                  bool bad = false;
                  for (int j = 0; j < 49; j++)
                    {
                      uint16_t ln = sh.it_len[lane * 49 + j];
                      sh.it_off[lane * 49 + j] = (uint16_t) sum;
                      if (ln == 0xFFFF)
                        bad = true;
                      else
                        sum += ln;
                    }
                  sh.seg_cnt[lane] = bad ? 0 : sum;
                }
              __syncthreads ();
              if (lane == 0)
                {
                  int acc = 0;
                  for (int s = 0; s < S; s++)
                    {
                      sh.seg_base[s] = acc;
                      acc += sh.seg_cnt[s];
                    }
                  sh.seg_base[S] = acc;
                  atomicAdd (&ctr->positions, (unsigned long long) acc);
                }
              __syncthreads ();
              const int T = sh.seg_base[S];
              uint32_t *raw = (T <= PM_SEED_CAP) ? sh.raw : g_raw;
              uint32_t *sorted = (T <= PM_SEED_CAP) ? sh.sorted : g_sorted;
              // ---- copy the bucket slices (each slice is ascending in .mdx)
              for (int x = lane; x < S * 49; x += 64)
                {
                  int seg = x / 49;
                  int ln = sh.it_len[x];
                  if (sh.seg_cnt[seg] > 0 && ln != 0xFFFF)
                    {
                      uint32_t *dst = raw + sh.seg_base[seg] + sh.it_off[x];
                      const uint32_t *s = ix.mers + sh.it_start[x];
                      for (int t = 0; t < ln; t++)
                        dst[t] = s[t];
                    }
                }
              __threadfence_block ();
              __syncthreads ();
              // ---- sort each segment ascending (qsort, pemapper.c:1613-1614).  A position occurs in one bucket only, so
              //      keys are distinct and the rank of an element is the number of smaller ones.
              for (int x = lane; x < T; x += 64)
                {
                  int seg = 0;
                  while (x >= sh.seg_base[seg + 1])
                    seg++;
                  const int sb = sh.seg_base[seg], sc = sh.seg_cnt[seg];
                  uint32_t v = raw[x];
                  int rank = 0;
                  if (sc <= 64)
                    {
                      for (int y = 0; y < sc; y++)
                        rank += (raw[sb + y] < v);
                    }
                  else
                    {
                      for (int j = 0; j < 49; j++)
                        {
                          int ln = sh.it_len[seg * 49 + j];
                          if (ln > 0)
                            rank += pm_lower_bound (raw + sb + sh.it_off[seg * 49 + j], ln, (int64_t) v);
                        }
                    }
                  sorted[sb + rank] = v;
                }
              __threadfence_block ();
              __syncthreads ();
              // ---- diagonal vote; the reverse strand is skipped when the forward one filled the list (pemapper.c:1656-1660)
              go_on = pm_find_matches (sh, sorted, total_cuts, idepth, min_match, tot, (uint8_t) strand, lane);
              if (tot >= PM_MAX_HITS)
                go_on = false;
              __syncthreads ();
            }
        }
      // ---- hits -> spots and SW windows (pemapper.c:1664-1669, 1047-1081)
      // an end with one hit is scored once, with direction nibbles, into its own slab; ends with several hits are
      // scored without, and only the winner is scored again (pm_select_kernel)
      unsigned tbase = 0;
      uint32_t *tasks = (tot == 1) ? tasks_s : tasks_m;
      if (lane == 0)
        {
          h.n_hits[e] = tot;
          h.slot[e] = (tot == 1) ? (int) atomicAdd (&ctr->n_slots, 1u) : -1;
          if (tot == 1)
            tbase = atomicAdd (&ctr->n_tasks_s, 1u);
          else if (tot > 1)
            tbase = atomicAdd (&ctr->n_tasks_m, (unsigned) tot);
        }
      tbase = __shfl (tbase, 0);
      for (int t = lane; t < tot; t += 64)
        {
          long temp = (long) sh.hits[t] - (long) sh.hits_off[t];
          uint32_t spot = (uint32_t) (temp > 0 ? temp : 0);
          int chrom = pm_find_chrom (ix.contig_starts, ix.n_contigs, spot);
          unsigned extra = 15u * (unsigned) chrom;
          long tt = (long) extra + (long) spot - (long) PM_SLOP;
          if (tt < 0)
            tt = 0;
          unsigned cs0 = ix.contig_starts[chrom] + extra;
          unsigned start_match = ((long) cs0 > tt) ? cs0 : (unsigned) tt;
          unsigned e1 = ix.contig_starts[chrom + 1] + extra;
          unsigned e2 = extra + spot + (unsigned) len + PM_SLOP;
          unsigned end_match = e1 < e2 ? e1 : e2;
          int blen = (int) (1u + end_match - start_match);
          size_t o = (size_t) e * PM_MAX_HITS + t;
          h.spot[o] = spot;
          h.orient[o] = sh.hits_or[t];
          h.gpos[o] = start_match;
          h.nn[o] = (int16_t) blen;
          tasks[tbase + t] = (uint32_t) o;
        }
    }
}


#include "pemap_sw.hip.h"
