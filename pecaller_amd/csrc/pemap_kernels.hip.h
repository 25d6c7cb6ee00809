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
// lanes per alignment in the SW kernels are a template parameter: 8 (half a DPP row) or 16 (a whole one)
// register budget of the SW kernels: 4 VGPRs of state per owned column (two doubles) plus temporaries
#define PM_WAVES_PER_EU(W) ((W) <= 13 ? 4 : (W) <= 19 ? 3 : (W) <= 32 ? 2 : 1)

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
  // look-up replicas (pemap_aux.hip.h): n_rep = 8 tables of 2^32 self-contained entries, replica p at rep + p * 2^32, or 0
  const uint32_t *rep;
  const uint32_t *multi;           // records {count, positions...} of the buckets of 2..99 positions, 16-byte units
  uint32_t multi_base;             // entries >= multi_base (and < 0xFFFFFFFE) point into multi
  int n_rep;
};

// The pileup counters (the reference's all_base_list: six unsigned short columns per genome position, pemapper.c:156, 1840-1870).
// Plane-major, 16 bits each, two positions to a word: plane q of position pos is half (pos & 1) of word q * plane_words + (pos >> 1).
// WHICH plane holds column col of a position depends on the reference letter there: for A / C / G / T (codes 0..3) the four base
// columns are rotated so that the column of the reference base itself is plane 0, q = (col - code) & 3; for every other letter, and
// for the Del / Ins columns 4 and 5, q = col.  Reads mostly agree with the reference, so the ~150 increments of a read fall into
// the ~5 lines of plane 0 under it plus a line per mismatch -- 7 lines instead of ~23 with a plane per base (and 57 with six u32
// counters per position side by side) -- and every touched line goes back to HBM once.  The arithmetic is the reference's: 16
// bits, wrapping.  Both halves are incremented with 32-bit atomics; the high half wraps by itself, the low half's wrap would carry
// into its neighbour, so increments of a low half return the old value and take the carry back when they see 0xFFFF.  A rotation
// per position is a bijection on that position's counters: sums over ranks, totals and resets do not see it; the export undoes it.
struct PmPile
{
  uint32_t *w;
  size_t plane_words;
  const uint8_t *genome;        // the letters the rotation follows
};

__device__ __forceinline__ int pm_pile_plane (uint8_t ref, int col)
{
  const int rc = (ref == 'A') ? 0 : (ref == 'C') ? 1 : (ref == 'G') ? 2 : (ref == 'T') ? 3 : -1;
  return (col > 3 || rc < 0) ? col : ((col - rc) & 3);
}

// word of the counter of (pos, plane q), and what a 32-bit add must add to it
__device__ __forceinline__ uint32_t *pm_pile_word (const PmPile & p, size_t pos, int q)
{
  return p.w + (size_t) q * p.plane_words + (pos >> 1);
}

// `add` = 1 (the low half), 0x10000 (the high half) or 0x10001 (both halves of a word: two neighbouring positions)
__device__ __forceinline__ void pm_pile_add (uint32_t * q, uint32_t add)
{
  if (!(add & 1u))
    atomicAdd (q, add);
  else
    {
      const uint32_t old = atomicAdd (q, add);
      if ((old & 0xFFFFu) == 0xFFFFu)
        atomicSub (q, 0x10000u);
    }
}

__device__ __forceinline__ void pm_pile_inc (const PmPile & p, size_t pos, int col)
{
  const int q = col > 3 ? col : pm_pile_plane (p.genome[pos], col);
  pm_pile_add (pm_pile_word (p, pos, q), (pos & 1) ? 0x10000u : 1u);
}

__device__ __forceinline__ uint16_t pm_pile_get (const PmPile & p, size_t pos, int col)
{
  const int q = col > 3 ? col : pm_pile_plane (p.genome[pos], col);
  return (uint16_t) (p.w[(size_t) q * p.plane_words + (pos >> 1)] >> (16 * (pos & 1)));
}

// counters shared by the kernels of one run (zeroed at the start of every run)
struct PmCounters
{
  unsigned int n_tasks_s;          // SW problems of read-ends with exactly one hit (scored WITH direction nibbles)
  unsigned int n_tasks_m;          // SW problems of read-ends with several hits (score only)
  unsigned int n_slots;            // direction slabs handed out
  unsigned int n_redo;             // winners of multi-hit ends: scored again with direction nibbles
  unsigned int n_wins;             // alignments to walk back
  unsigned int n_tasks_dp;         // single-hit problems the gapless rule (pm_gapless_kernel) leaves to the DP
  unsigned long long positions;    // P: entries copied out of .mdx
  unsigned long long cells_score;  // DP cells computed without / with direction nibbles
  unsigned long long cells_dirs;
  unsigned long long pile_incs;
  unsigned long long n_ins;
  unsigned int sw_next[4];         // work counters of the three SW launches of a chunk (single-hit, multi-hit, redo)
  unsigned int n_band[2];          // problems left to the banded DP (pm_band_kernel): single-hit ends, multi-hit ends
  unsigned int band_next[2];       // their work counters
  unsigned long long cells_band;   // band cells computed
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

// the look-up replicas (pemap_aux.hip.h): replica p holds the entry of k-mer k at k with its 4-bit fields 0 and p swapped
__device__ __forceinline__ uint32_t pm_swap_fields (uint32_t k, int p)
{
  const unsigned sh = 4u * (unsigned) p;
  const uint32_t f0 = k & 15u, fp = (k >> sh) & 15u;
  return (k & ~(15u | (15u << sh))) | (f0 << sh) | fp;        // p = 0: k itself
}

// neighbour j of k-mer k through the replicas: the entry (0xFFFFFFFF empty, 0xFFFFFFFE too many, < multi_base the bucket's only
// position, else the code of its record), read from the replica in which k and its neighbours at that pair of bases share a line
__device__ __forceinline__ uint32_t pm_rep_entry (const PmIndex & ix, uint32_t k, int j)
{
  const int p = j > 0 ? (((j - 1) / 3) >> 1) & 7 : 0;
  return ix.rep[((size_t) p << 32) + (size_t) pm_swap_fields (pm_neighbour (k, j), p)];
}

// s_setprio takes an immediate: wave issue priority 0 (default) .. 3 among the waves of a SIMD
__device__ __forceinline__ void pm_set_prio (int p)
{
  if (p == 1)
    __builtin_amdgcn_s_setprio (1);
  else if (p == 2)
    __builtin_amdgcn_s_setprio (2);
  else if (p == 3)
    __builtin_amdgcn_s_setprio (3);
}

#include "pemap_seed.hip.h"
#include "pemap_seed2.hip.h"
#include "pemap_seed4.hip.h"
#include "pemap_sw.hip.h"
#include "pemap_band.hip.h"
