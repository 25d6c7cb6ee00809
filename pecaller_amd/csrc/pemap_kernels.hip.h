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
#define PM_TMAX 308                // SW steps: window rows (<= 299) + PM_LPA - 1, rounded up
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

// counters shared by the kernels of one run
struct PmCounters
{
  unsigned int n_tasks;            // SW score problems (H)
  unsigned int n_trace;            // SW trace problems
  unsigned int ins_bytes;          // insertion log cursor
  unsigned int ins_overflow;
  unsigned long long positions;    // P: entries copied out of .mdx
  unsigned long long cells_score;
  unsigned long long cells_trace;
  unsigned long long pile_incs;
  unsigned long long n_ins;
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

__global__ __launch_bounds__ (64) void pm_seed_kernel (PmIndex ix, PmBatch b, PmParams prm, PmHits h, uint32_t * tasks,
                                                      PmCounters * ctr, uint32_t * gscratch)
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
      unsigned tbase = 0;
      if (lane == 0)
        {
          h.n_hits[e] = tot;
          if (tot > 0)
            tbase = atomicAdd (&ctr->n_tasks, (unsigned) tot);
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

// ============================================================================================================
// K3: Smith-Waterman forward (smith_waterman_align, pemapper.c:1694-1748).
//
// PM_LPA = 8 lanes share one alignment, 8 alignments per wave.  Lane g owns read columns j = g*W+1 .. g*W+W and
// walks the reference rows skewed by g (row i = step - g), so that the only lane-to-lane traffic per step is two
// doubles from lane g-1: S2 of the cell to the left and max3 of that cell (next row's diagonal).  Per own column a lane
// keeps two doubles in registers: U1 = S1 of the row below and D = max(S0,S1,S2) of the cell.
//
//   S2[i][j] = max (S0[i][j-1]-go, S2[i][j-1]-ge)      S1[i][j] = max (S0[i-1][j]-go, S1[i-1][j]-ge)
//   S0[i][j] = max3 (S.[i-1][j-1]) + match             (the reference adds the bonus to each plane before the max;
//                                                        rounding is monotone, so max-then-add gives the same double)
//
// With DIRS the four comparisons the traceback will make at a cell (pemapper.c:1799-1831) are stored as one nibble:
//   bit0 S1>S0   bit1 S2>max(S0,S1)   bit2 S1-ge > S0-go   bit3 S2-ge > S0-go
// ============================================================================================================
#define PM_GO 2.0
#define PM_GE (1.0 / 36.0)
#define PM_MISS (-1.0 / 3.0)

__device__ __forceinline__ double pm_border (int j)     // S[k][0][j], j >= 1, pemapper.c:2077-2078
{
  return -(PM_GO + (double) (j - 1) * PM_GE);
}

__device__ __forceinline__ double pm_max (double a, double b)   // maxim(a,b), pemapper.c:36
{
  // (a > b) ? a : b.  No NaN and no negative zero can arise in this recurrence (finite constants, add/sub only), so the
  // IEEE maximum is the same double; it is one v_max_f64 instead of a compare and two selects.
  return __builtin_fmax (a, b);
}

// init_bonus_matrices, pemapper.c:2006-2035, as a predicate.  The row fill of iteration i precedes that iteration's
// 'N'/'n' assignments, so row 'N' keeps 1.0 only for columns >= 'N': a reference N matches read N, T and lower case
// but not read A/C/G; a read N matches everything.
__device__ __forceinline__ bool pm_match (uint8_t r, uint8_t q, int bis)
{
  bool m = (r == q) | (q == 'N') | (q == 'n');
  m |= (r == 'N') & (q >= 'N');
  m |= (r == 'n') & (q >= 'n');
  m |= (bis != 0) & ((r == 'C') | (r == 'c')) & ((q == 'T') | (q == 't'));
  return m;
}

struct PmSwTask
{
  const uint8_t *read;             // forward read bytes
  const uint8_t *ref;              // genome + window start
  int mm, nn;
  int orient;
  bool valid;
};

__device__ __forceinline__ uint8_t pm_oriented (const PmSwTask & tk, int jz)
{
  return tk.orient ? pm_rc (tk.read[tk.mm - 1 - jz]) : tk.read[jz];
}

// one DP cell; everything by value/reference so that after unrolling all state lives in registers
template < bool DIRS > __device__ __forceinline__ void pm_cell (double &dg, double &s2, double &U1c, double &Dc, uint32_t mword, int bit,
                                                                uint32_t & dword, int nibpos, double &o0, double &o1, double &o2)
{
  // bump = match ? 1.0 : -1/3 assembled from the mask bit without a branch (v_bfe_i32 + 2 x v_bfi_b32)
  const int t = __builtin_amdgcn_sbfe ((int) mword, bit, 1);            // 0 or -1
  const uint32_t hi = ((uint32_t) t & 0x3FF00000u) | (~(uint32_t) t & 0xBFD55555u);
  const uint32_t lo = ~(uint32_t) t & 0x55555555u;
  const double bump = __hiloint2double ((int) hi, (int) lo);
  const double s1 = U1c;
  const double s0 = dg + bump;
  dg = Dc;
  const double a0 = s0 - PM_GO;
  const double x1 = s1 - PM_GE;
  const double x2 = s2 - PM_GE;
  const double m01 = pm_max (s0, s1);
  U1c = pm_max (a0, x1);
  Dc = pm_max (m01, s2);
  if (DIRS)
    {
      uint32_t nib = (s1 > s0 ? 1u : 0u) | (s2 > m01 ? 2u : 0u) | (x1 > a0 ? 4u : 0u) | (x2 > a0 ? 8u : 0u);
      dword |= nib << nibpos;
    }
  o0 = s0;
  o1 = s1;
  o2 = s2;
  s2 = pm_max (a0, x2);
}

// UNI: every alignment of the wave has the same read length, so the last-column tracker's position is wave-uniform and
// the test `c == c_last` is a scalar branch.  Otherwise it is a per-lane predicate (mixed read lengths; slower).
template < int W, bool DIRS, bool UNI >
__device__ __forceinline__ void pm_sw_forward (const PmSwTask & tk, int bis, int lane, int nn_max, uint32_t * dirbuf,
                                               double &best, int &bk, int &bi)
{
  constexpr int DW = (W * 4 + 31) / 32;
  int g = lane & (PM_LPA - 1);
  // opaque to the optimiser: otherwise the 2 W border doubles below are computed once per kernel, kept live across the
  // persistent task loop and double the register footprint
  asm volatile ("":"+v" (g));
  const int mm = tk.mm, nn = tk.valid ? tk.nn : 0;
  // ---- match masks of this lane's W read bytes against reference A, C, G, T, N (bit c = column c of the lane)
  uint64_t mk[5] = { 0, 0, 0, 0, 0 };
#pragma unroll 1
  for (int c = 0; c < W; c++)
    {
      const int jz = g * W + c;    // 0-based read position
      uint8_t q = 0;
      if (tk.valid && jz < mm)
        q = pm_oriented (tk, jz);
      const uint64_t bitc = 1ull << c;
      if (q != 0)
        {
          mk[0] |= pm_match ('A', q, bis) ? bitc : 0ull;
          mk[1] |= pm_match ('C', q, bis) ? bitc : 0ull;
          mk[2] |= pm_match ('G', q, bis) ? bitc : 0ull;
          mk[3] |= pm_match ('T', q, bis) ? bitc : 0ull;
          mk[4] |= pm_match ('N', q, bis) ? bitc : 0ull;
        }
    }
  double U1[W], D[W];
#pragma unroll
  for (int c = 0; c < W; c++)
    {
      const double bj = pm_border (g * W + c + 1);
      D[c] = bj;
      U1[c] = pm_max (bj - PM_GO, bj - PM_GE);
    }
  double Dprev = (g == 0) ? 0.0 : pm_border (g * W);   // max3 of cell (0, j0-1)
  double R2out = 0.0, Dout = 0.0;
  int g_last = (mm - 1) / W, c_last = (mm - 1) - g_last * W;
  if (UNI)
    {
      g_last = __builtin_amdgcn_readfirstlane (g_last);
      c_last = __builtin_amdgcn_readfirstlane (c_last);
    }
  double bst = pm_border (mm);     // S[0][0][mm], pemapper.c:1701-1703
  int k_b = 0, i_b = 0;
  uint8_t r_next = (tk.valid && g == 0 && nn >= 1) ? tk.ref[0] : 0;

  for (int t = 1; t <= nn_max + PM_LPA - 1; t++)
    {
      const int i = t - g;
      double R2in = __shfl_up (R2out, 1, PM_LPA);
      double Dimp = __shfl_up (Dout, 1, PM_LPA);
      if (g == 0)
        {
          R2in = pm_max (0.0 - PM_GO, -PM_GO - PM_GE);      // S2[i][1] from the column-0 border (pemapper.c:2079-2081)
          Dimp = 0.0;
        }
      const bool act = (i >= 1) && (i <= nn);
      const uint8_t r = r_next;
      if (tk.valid && (i + 1 >= 1) && (i + 1 <= nn))
        r_next = tk.ref[i];        // next step's reference byte
      if (act)
        {
          uint64_t msel = (r == 'A') ? mk[0] : (r == 'C') ? mk[1] : (r == 'G') ? mk[2] : (r == 'T') ? mk[3] : mk[4];
          if (r != 'A' && r != 'C' && r != 'G' && r != 'T' && r != 'N')
            {
              // any other reference byte (IUPAC codes, lower case): evaluate the predicate column by column
              msel = 0;
#pragma unroll 1
              for (int c = 0; c < W; c++)
                {
                  const int jz = g * W + c;
                  if (jz < mm && pm_match (r, pm_oriented (tk, jz), bis))
                    msel |= 1ull << c;
                }
            }
          const uint32_t m0 = (uint32_t) msel, m1 = (uint32_t) (msel >> 32);
          double dg = Dprev;
          double s2 = R2in;
          uint32_t dw[DW];
#pragma unroll
          for (int d = 0; d < DW; d++)
            dw[d] = 0;
          double t0 = 0.0, t1 = 0.0, t2 = 0.0;
          bool have = false;
#pragma unroll
          for (int c = 0; c < W; c++)
            {
              double o0, o1, o2;
              pm_cell < DIRS > (dg, s2, U1[c], D[c], (c < 32) ? m0 : m1, c & 31, dw[c >> 3], (c & 7) * 4, o0, o1, o2);
              if (UNI)
                {
                  if (c == c_last)
                    {
                      t0 = o0;
                      t1 = o1;
                      t2 = o2;
                      have = true;
                    }
                }
              else if (c == c_last)
                {
                  t0 = o0;
                  t1 = o1;
                  t2 = o2;
                  have = true;
                }
            }
          if (have && g == g_last)
            {
              // last read column, rows ascending, planes 0,1,2, strict '>' (pemapper.c:1724-1741)
              if (t0 > bst) { bst = t0; k_b = 0; i_b = i; }
              if (t1 > bst) { bst = t1; k_b = 1; i_b = i; }
              if (t2 > bst) { bst = t2; k_b = 2; i_b = i; }
            }
          R2out = s2;
          Dout = D[W - 1];
          Dprev = Dimp;
          if (DIRS)
            {
              uint32_t *dst = dirbuf + ((size_t) (t - 1) * 64 + lane) * DW;
#pragma unroll
              for (int d = 0; d < DW; d++)
                dst[d] = dw[d];
            }
        }
    }
  // hand the tracker's result to lane g == 0 of the group
  const int srcl = (lane & ~(PM_LPA - 1)) + ((mm - 1) / W);
  best = __shfl (bst, srcl);
  bk = __shfl (k_b, srcl);
  bi = __shfl (i_b, srcl);
}

__device__ __forceinline__ int pm_wave_max (int v)
{
  for (int o = 32; o > 0; o >>= 1)
    v = max (v, __shfl_xor (v, o));
  return v;
}

template < int W, bool UNI > __global__ __launch_bounds__ (64, PM_WAVES_PER_EU (W)) void pm_sw_score_kernel (PmIndex ix, PmBatch b, PmParams prm, PmHits h,
                                                                                         const uint32_t * tasks, PmCounters * ctr,
                                                                                         int mm_uniform)
{
  const int lane = threadIdx.x;
  const int q = lane >> 3;
  const unsigned n_tasks = ctr->n_tasks;
  for (unsigned base = blockIdx.x * 8u; base < n_tasks; base += gridDim.x * 8u)
    {
      PmSwTask tk;
      tk.valid = (base + q) < n_tasks;
      size_t o = 0;
      tk.mm = UNI ? mm_uniform : 16;
      tk.nn = 0;
      tk.orient = 0;
      tk.read = nullptr;
      tk.ref = nullptr;
      if (tk.valid)
        {
          o = tasks[base + q];
          int end = (int) (o / PM_MAX_HITS);
          tk.read = pm_read_ptr (b, end, &tk.mm);
          tk.nn = h.nn[o];
          if (tk.nn < 0)
            tk.nn = 0;
          tk.orient = h.orient[o];
          tk.ref = ix.genome + h.gpos[o];
        }
      int nn_max = pm_wave_max (tk.nn);
      double best;
      int bk, bi;
      pm_sw_forward < W, false, UNI > (tk, prm.bisulfite, lane, nn_max, nullptr, best, bk, bi);
      if (tk.valid && (lane & 7) == 0)
        {
          h.score[o] = best;
          h.stk[o] = (uint8_t) bk;
          h.sti[o] = (int16_t) bi;
          atomicAdd (&ctr->cells_score, (unsigned long long) tk.nn * tk.mm);
        }
    }
}

// ============================================================================================================
// K4: pair / single-end selection (pemapper.c:1084-1185 and find_mate_pairs 1313-1536), one lane per read (pair).
// Output: the hit to trace per end (or -1), the class, and m1/m2 = window start + start[1] + 1 (pemapper.c:1208, 1228).
// ============================================================================================================
__device__ int pm_single_select (const double *sc, int n, int len, double min_align, int *bsm)
{
  double good_score = len * min_align * 1.0;
  double top_score = -PM_GO * len;
  int top_score_count = 0;
  for (int i = 0; i < n; i++)
    {
      double this_score = sc[i];
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
    return 8;                   // NEITHER_MAP
  if (top_score_count == 1)
    return 2;                   // UNIQUE_SINGLE
  return 7;                     // NON_NO
}

__global__ void pm_select_kernel (PmBatch b, PmParams prm, PmHits h, uint32_t * trace, PmCounters * ctr, uint32_t * m1,
                                  uint32_t * m2, int *mtype)
{
  int it = blockIdx.x * blockDim.x + threadIdx.x;
  if (it >= b.n)
    return;
  const int e1 = b.paired ? 2 * it : it;
  const int n1 = h.n_hits[e1];
  const int n2 = b.paired ? h.n_hits[e1 + 1] : 0;
  const size_t o1 = (size_t) e1 * PM_MAX_HITS, o2 = o1 + PM_MAX_HITS;
  const double *s1 = h.score + o1, *s2 = h.score + o2;
  const int l1 = b.len1[it], l3 = b.paired ? b.len2[it] : 0;
  int use1 = -1, use2 = -1, code;
  if (n1 > 0 && n2 == 0)
    {
      int bsm = 0;
      code = pm_single_select (s1, n1, l1, prm.min_align, &bsm);
      if (code == 2)
        use1 = bsm;
    }
  else if (n2 > 0 && n1 == 0)
    {
      int bsm = 0;
      code = pm_single_select (s2, n2, l3, prm.min_align, &bsm);
      if (code == 2)
        use2 = bsm;
    }
  else if (n1 > 0 && n2 > 0)
    {
      const double good1 = l1 * prm.min_align * 1.0, good2 = l3 * prm.min_align * 1.0;
      int perfect = 0, slip_count = 0;
      unsigned sm1 = 0xFFFFFFFFu, sm2 = 0xFFFFFFFFu;
      double tot_best = -1e5;
      for (int w1 = 0; w1 < n1; w1++)
        if (s1[w1] >= good1)
          for (int w2 = 0; w2 < n2; w2++)
            if (s2[w2] >= good2)
              {
                long p1 = (long) h.spot[o1 + w1], p2 = (long) h.spot[o2 + w2];
                long temp_dist = labs (p1 - p2);
                int or1 = h.orient[o1 + w1], or2 = h.orient[o2 + w2];
                if ((temp_dist >= prm.min_dist) && (temp_dist <= prm.max_dist) && (or1 != or2))
                  {
                    double inc = s1[w1] + s2[w2] - tot_best;
                    if (inc > 0.001)
                      {
                        perfect = 1;
                        sm1 = w1;
                        sm2 = w2;
                        tot_best = s1[w1] + s2[w2];
                        slip_count = 1;
                      }
                    else if (inc > -0.001)
                      {
                        if (sm1 == (unsigned) w1 || sm2 == (unsigned) w2)
                          slip_count++;
                        perfect++;
                      }
                  }
              }
      code = 8;
      if (perfect > 0)
        {
          use1 = (int) sm1;
          use2 = (int) sm2;
          if (perfect == 1)
            code = 0;
          else if (slip_count == perfect)
            code = 1;
          else
            {
              code = 4;
              use1 = use2 = -1;
            }
        }
      else
        {
          int best1 = 0, best2 = 0, m1_c = 0, m2_c = 0;
          for (int i = 1; i < n1; i++)
            if (s1[i] > s1[best1])
              {
                best1 = i;
                m1_c = 1;
              }
            else if (s1[i] - s1[best1] > -0.0001)
              m1_c++;
          // the reference indexes smax2 with best1 here (pemapper.c:1468); smax2[k] for k >= n2 reads its -1.0 fill (1348-1351)
          const double s2b1 = (best1 < n2) ? s2[best1] : -1.0;
          for (int i = 1; i < n2; i++)
            if (s2[i] > s2[best2])
              {
                best2 = i;
                m2_c = 1;
              }
            else if (s2[i] - s2b1 > -0.0001)
              m2_c++;
          const bool ok2 = (s2[best2] >= good2) && (m2_c < 2);
          if (s1[best1] >= good1)
            {
              if (m1_c < 2)
                {
                  use1 = best1;
                  if (ok2)
                    {
                      use2 = best2;
                      code = 3;
                    }
                  else
                    code = 2;
                }
              else if (ok2)
                {
                  use2 = best2;
                  code = 2;
                }
              else
                code = 5;
            }
          else if (ok2)
            {
              use2 = best2;
              code = 2;
            }
          else
            code = 5;
        }
    }
  else
    code = 8;
  uint32_t r1 = 0, r2 = 0;
  if (use1 >= 0)
    {
      size_t o = o1 + use1;
      r1 = (uint32_t) (h.gpos[o] + (uint32_t) h.sti[o]) + 1u;
      trace[atomicAdd (&ctr->n_trace, 1u)] = (uint32_t) o;
    }
  if (use2 >= 0)
    {
      size_t o = o2 + use2;
      r2 = (uint32_t) (h.gpos[o] + (uint32_t) h.sti[o]) + 1u;
      trace[atomicAdd (&ctr->n_trace, 1u)] = (uint32_t) o;
    }
  m1[it] = r1;
  if (m2)
    m2[it] = r2;
  mtype[it] = code;
}

// ============================================================================================================
// K5: traceback + pileup (smith_waterman_backtrack, pemapper.c:1752-1965).  The winning alignments are scored again
// with DIRS (nibbles to the wave's slab of the direction buffer), then lane 0 of each 8-lane group walks its path.
// Pileup counters are u32 in HBM updated with no-return atomics (the reference's u16 counters wrap; the fetch
// truncates, which is the same arithmetic).  Insertions go to a byte log through an atomic cursor.
// ============================================================================================================
template < int W, bool UNI > __global__ __launch_bounds__ (64, PM_WAVES_PER_EU (W)) void pm_sw_trace_kernel (PmIndex ix, PmBatch b, PmParams prm, PmHits h,
                                                                                         const uint32_t * trace, PmCounters * ctr,
                                                                                         uint32_t * dirbuf_all, uint32_t * counts,
                                                                                         uint8_t * ins_log, unsigned ins_cap, int mm_uniform)
{
  constexpr int DW = (W * 4 + 31) / 32;
  const int lane = threadIdx.x;
  const int q = lane >> 3;
  const unsigned n_trace = ctr->n_trace;
  uint32_t *dirbuf = dirbuf_all + (size_t) blockIdx.x * PM_TMAX * 64 * DW;
  for (unsigned base = blockIdx.x * 8u; base < n_trace; base += gridDim.x * 8u)
    {
      PmSwTask tk;
      tk.valid = (base + q) < n_trace;
      size_t o = 0;
      tk.mm = UNI ? mm_uniform : 16;
      tk.nn = 0;
      tk.orient = 0;
      tk.read = nullptr;
      tk.ref = nullptr;
      uint32_t gpos = 0;
      if (tk.valid)
        {
          o = trace[base + q];
          int end = (int) (o / PM_MAX_HITS);
          tk.read = pm_read_ptr (b, end, &tk.mm);
          tk.nn = h.nn[o];
          if (tk.nn < 0)
            tk.nn = 0;
          tk.orient = h.orient[o];
          gpos = h.gpos[o];
          tk.ref = ix.genome + gpos;
        }
      int nn_max = pm_wave_max (tk.nn);
      double best;
      int bk, bi;
      __syncthreads ();
      pm_sw_forward < W, true, UNI > (tk, prm.bisulfite, lane, nn_max, dirbuf, best, bk, bi);
      __threadfence ();
      __syncthreads ();
      if (tk.valid && (lane & 7) == 0)
        {
          const int mm = tk.mm;
          int k = bk, i = bi, j = mm;
          int i1 = 0, j1 = 0, ins_len = 0;
          unsigned long long incs = 0, nins = 0;
          const int gl = lane;      // lane of group member 0
          while (i > 0 && j > 0)
            {
              i1 = i - 1;
              j1 = j - 1;
              int maxi, maxj, maxk;
              // cell whose comparisons decide the predecessor plane
              int ci, cj;
              if (k == 0) { maxi = i1; maxj = j1; ci = i1; cj = j1; }
              else if (k == 2) { maxi = i; maxj = j1; ci = i; cj = j1; }
              else { maxi = i1; maxj = j; ci = i1; cj = j; }
              maxk = 0;
              if (ci >= 1 && cj >= 1)
                {
                  int gg = (cj - 1) / W, c = (cj - 1) - gg * W;
                  uint32_t wv = dirbuf[((size_t) (ci + gg - 1) * 64 + (gl + gg)) * DW + (c >> 3)];
                  uint32_t nib = (wv >> ((c & 7) * 4)) & 0xFu;
                  if (k == 0)
                    maxk = (nib & 2u) ? 2 : ((nib & 1u) ? 1 : 0);
                  else if (k == 2)
                    maxk = (nib & 8u) ? 2 : 0;
                  else
                    maxk = (nib & 4u) ? 1 : 0;
                }
              // border cells: the walk ends after this step (i or j becomes 0), maxk is never used
              uint32_t *cnt = counts + ((size_t) gpos + (size_t) i1) * 6;
              if (maxi != i)
                {
                  if (maxj != j)
                    {
                      uint8_t ch = tk.orient ? pm_rc (tk.read[mm - 1 - j1]) : tk.read[j1];
                      int slot = (ch == 'A') ? 0 : (ch == 'C') ? 1 : (ch == 'G') ? 2 : (ch == 'T') ? 3 : -1;
                      if (slot >= 0)
                        {
                          atomicAdd (&cnt[slot], 1u);
                          incs++;
                        }
                    }
                  else
                    {
                      atomicAdd (&cnt[4], 1u);
                      incs++;
                    }
                  if (ins_len > 0)
                    {
                      // inserted bases = oriented read [j, j + ins_len) (collected right to left, stored back in read order, 1892-1893)
                      unsigned need = 8u + (((unsigned) ins_len + 3u) & ~3u);
                      unsigned at = atomicAdd (&ctr->ins_bytes, need);
                      if (at + need <= ins_cap)
                        {
                          *(uint32_t *) (ins_log + at) = gpos + (uint32_t) i1;
                          *(uint32_t *) (ins_log + at + 4) = (uint32_t) ins_len;
                          for (int m = 0; m < ins_len; m++)
                            {
                              int jz = j + m;
                              ins_log[at + 8 + m] = tk.orient ? pm_rc (tk.read[mm - 1 - jz]) : tk.read[jz];
                            }
                        }
                      else
                        atomicExch (&ctr->ins_overflow, 1u);
                      atomicAdd (&cnt[5], 1u);
                      incs++;
                      nins++;
                    }
                  ins_len = 0;
                }
              else
                ins_len++;
              i = maxi;
              j = maxj;
              k = maxk;
            }
          if (ins_len > 0 && i >= 1)        // pemapper.c:1918-1958: attached to base[i1] of the last step
            {
              uint32_t *cnt = counts + ((size_t) gpos + (size_t) i1) * 6;
              unsigned need = 8u + (((unsigned) ins_len + 3u) & ~3u);
              unsigned at = atomicAdd (&ctr->ins_bytes, need);
              if (at + need <= ins_cap)
                {
                  *(uint32_t *) (ins_log + at) = gpos + (uint32_t) i1;
                  *(uint32_t *) (ins_log + at + 4) = (uint32_t) ins_len;
                  for (int m = 0; m < ins_len; m++)
                    {
                      int jz = j + m;
                      ins_log[at + 8 + m] = tk.orient ? pm_rc (tk.read[mm - 1 - jz]) : tk.read[jz];
                    }
                }
              else
                atomicExch (&ctr->ins_overflow, 1u);
              atomicAdd (&cnt[5], 1u);
              incs++;
              nins++;
            }
          atomicAdd (&ctr->cells_trace, (unsigned long long) tk.nn * tk.mm);
          atomicAdd (&ctr->pile_incs, incs);
          if (nins)
            atomicAdd (&ctr->n_ins, nins);
        }
    }
}
