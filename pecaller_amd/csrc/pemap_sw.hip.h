// pemap_sw.hip.h -- Smith-Waterman scoring, pair selection and traceback/pileup kernels (gfx950, wave64).
// Included at the end of pemap_kernels.hip.h.
#pragma once

// ============================================================================================================
// K3: Smith-Waterman forward (smith_waterman_align, pemapper.c:1694-1748).
//
// PM_LPA = 8 lanes share one alignment, 8 alignments per wave.  The read is RIGHT-aligned on the 8 x W columns of
// the lane group: read column j (1-based) sits at global column J = j - 1 + pad, pad = 8 W - mm; lane g owns
// J = g W .. g W + W - 1 and walks the reference rows skewed by g (row i = step - g).  The only lane-to-lane traffic
// per step is two doubles from lane g-1: S2 of the cell to the left and max3 of that cell (next row's diagonal).
// Per owned column a lane keeps two doubles in registers: U1 = S1 of the row below and D = max(S0,S1,S2) of the cell.
//
//   S2[i][j] = max (S0[i][j-1]-go, S2[i][j-1]-ge)      S1[i][j] = max (S0[i-1][j]-go, S1[i-1][j]-ge)
//   S0[i][j] = max3 (S.[i-1][j-1]) + match             (the reference adds the bonus to each plane before the max;
//                                                        rounding is monotone, so max-then-add gives the same double)
//
// Right alignment puts the last read column -- the only one whose cells compete for the best score
// (pemapper.c:1717-1742) -- at lane 7, column W-1 for every read length, so the tracker costs nothing per cell.  The
// pad columns to the left of read column 1 are made to reproduce the column-0 border (S0 = 0, S1 = 0, S2 = -go,
// pemapper.c:2062-2081) exactly: they hold a wildcard (bonus +1 against every reference byte), the pad column k to the
// left of column 0 starts at D = -k, U1 = -1e30, and lane 0 is fed D = -pad, S2 = -1e30.  By induction S0 = -k there,
// so column 0 presents max3 = 0 and hands S2 = max (0-go, .) = -go to column 1: small integers, exact in fp64.
//
// With DIRS the four comparisons the traceback will make at a cell (pemapper.c:1799-1831) are stored as one nibble:
//   bit0 S1>S0   bit1 S2>max(S0,S1)   bit2 S1-ge > S0-go   bit3 S2-ge > S0-go
// in the alignment's direction slab: dword ((g * tstride + step-1) * DW + c/8), first cell of a dword in its highest
// nibble -- lane-major, so that the diagonal walk of the traceback reads consecutive addresses.
// ============================================================================================================
#define PM_GO 2.0
#define PM_GE (1.0 / 36.0)
#define PM_NEGBIG (-1.0e30)

__device__ __forceinline__ double pm_border (int j)     // S[k][0][j], j >= 1, pemapper.c:2077-2078
{
  return -(PM_GO + (double) (j - 1) * PM_GE);
}

// max3 of row 0 at read column j, extended to the pad columns j <= 0 (see above)
__device__ __forceinline__ double pm_top (int j)
{
  return (j >= 1) ? pm_border (j) : (double) j;
}

__device__ __forceinline__ double pm_max (double a, double b)   // maxim(a,b), pemapper.c:36
{
  // (a > b) ? a : b.  No NaN and no negative zero can arise in this recurrence (finite constants, add/sub only), so the
  // IEEE maximum is the same double: one v_max_f64 (built with -fno-honor-nans -mno-amdgpu-ieee: no canonicalising copy).
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

__device__ __forceinline__ uint8_t pm_oriented (const uint8_t * read, int mm, int orient, int jz)
{
  return orient ? pm_rc (read[mm - 1 - jz]) : read[jz];
}

// Bonus constants as bit patterns: 1.0 = 0x3FF00000'00000000, -1/3 = 0xBFD55555'55555555 (pemapper.c:2011-2019).
struct PmBumpK
{
  uint32_t hi_match, hi_miss, lo_miss;
};

// acc = 2 * acc + (a > b): the compare writes a lane mask to an SGPR pair and v_addc_co_u32 shifts it in as the carry --
// two instructions per stored comparison instead of compare + select + shift/or.
__device__ __forceinline__ unsigned long long pm_gt (double a, double b)
{
  return __builtin_amdgcn_fcmp (a, b, 2);       // FCMP_OGT
}

__device__ __forceinline__ void pm_push (uint32_t & acc, unsigned long long m)
{
  unsigned long long carry_out;
  asm ("v_addc_co_u32 %0, %1, %0, %0, %2":"+v" (acc), "=s" (carry_out):"s" (m));
}

// bonus = match ? 1.0 : -1/3, assembled from bit `bit` of the lane's match mask: v_bfe_i32 (0 / -1) and two v_bfi_b32
__device__ __forceinline__ double pm_bump (uint32_t mword, const int bit, const PmBumpK & bk)
{
  int t;
  uint32_t hi, lo;
  asm ("v_bfe_i32 %0, %1, %2, 1":"=v" (t):"v" (mword), "n" (bit));
  asm ("v_bfi_b32 %0, %1, %2, %3":"=v" (hi):"v" (t), "v" (bk.hi_match), "v" (bk.hi_miss));
  asm ("v_bfi_b32 %0, %1, 0, %2":"=v" (lo):"v" (t), "v" (bk.lo_miss));
  return __hiloint2double ((int) hi, (int) lo);
}

// One DP cell; after unrolling all state lives in registers.  s0 = S0 of this cell (diagonal + bonus) comes in already
// added: the caller forms the NEXT cell's s0 from the old Dc before this cell overwrites it, and every read of an old
// value precedes the write of the new one, so no register copies are needed.
template < bool DIRS > __device__ __forceinline__ void pm_cell (const double s0, double &s2, double &U1c, double &Dc, uint32_t & dword)
{
  const double s1 = U1c;
  const double a0 = s0 - PM_GO;
  const double x1 = s1 - PM_GE;
  const double x2 = s2 - PM_GE;
  const double m01 = pm_max (s0, s1);
  unsigned long long k3 = 0, k2 = 0, k1 = 0, k0 = 0;
  if (DIRS)
    {
      // nibble, most significant bit first: S2-ge > S0-go, S1-ge > S0-go, S2 > max(S0,S1), S1 > S0
      k3 = pm_gt (x2, a0);
      k2 = pm_gt (x1, a0);
      k1 = pm_gt (s2, m01);
      k0 = pm_gt (s1, s0);
    }
  Dc = pm_max (m01, s2);
  U1c = pm_max (a0, x1);
  s2 = pm_max (a0, x2);
  if (DIRS)
    {
      pm_push (dword, k3);
      pm_push (dword, k2);
      pm_push (dword, k1);
      pm_push (dword, k0);
    }
}

// value of lane - 1.  Alignment groups of 8 or 16 lanes never straddle a DPP row: row_shr:1 (the first lane of a row reads
// 0).  The first lane of every group overrides what it receives.
template < int LPA > __device__ __forceinline__ uint32_t pm_from_left (uint32_t v)
{
  static_assert (LPA == 8 || LPA == 16, "alignment groups are half a DPP row or a whole one");
  return (uint32_t) __builtin_amdgcn_update_dpp (0, (int) v, 0x111, 0xF, 0xF, true);
}

template < int LPA > __device__ __forceinline__ double pm_from_left (double v)
{
  const uint32_t lo = pm_from_left < LPA > ((uint32_t) __double2loint (v)), hi = pm_from_left < LPA > ((uint32_t) __double2hiint (v));
  return __hiloint2double ((int) hi, (int) lo);
}

typedef uint64_t pm_u64_unaligned __attribute__ ((aligned (1)));

// reference bytes 8 k .. 8 k + 7 of a window of nn bytes, byte 8 k in the low byte; bytes past the window read as 0
__device__ __forceinline__ uint64_t pm_ref_chunk (const uint8_t * ref, int nn, int k)
{
  const int s = 8 * k;
  const int last = nn > 8 ? nn - 8 : 0;
  const int sc = s < last ? s : last;
  const uint64_t v = *(const pm_u64_unaligned *) (ref + sc);
  const int shift = s - sc;
  return shift >= 8 ? 0ull : (v >> (8 * shift));
}

template < int W > struct PmSwGeom
{
  static constexpr int DW = (W * 4 + 31) / 32;   // dwords of direction nibbles per lane per row
  // Cells are shifted into their dword first-in-highest: cell c of a lane sits in dword c / 8 at this bit offset
  __host__ __device__ static constexpr int nib_shift (int c)
  {
    return 4 * (((W - (c / 8) * 8) < 8 ? (W - (c / 8) * 8) : 8) - 1 - (c % 8));
  }
};

// match mask of one lane's columns against an arbitrary reference byte (IUPAC codes, lower case): rare, kept out of line
template < int W > __device__ __noinline__ uint64_t pm_slow_mask (const uint8_t * read, int mm, int orient, int g, int pad, uint8_t r, int bis)
{
  uint64_t msel = 0;
  for (int c = 0; c < W; c++)
    {
      const int jz = g * W + c - pad;
      if (jz < 0 || pm_match (r, pm_oriented (read, mm, orient, jz), bis))
        msel |= 1ull << c;
    }
  return msel;
}

// Direction nibbles leave the wave through LDS: a lane produces DW dwords per step, which in HBM are consecutive for
// consecutive steps of that lane.  Written directly, every store instruction would touch 64 different lines with 8-20
// bytes each and each of them reaches HBM as its own partial-line write.  Instead PM_STAGE steps are collected in LDS
// ([lane][step][DW]) and flushed with 16-byte-per-lane stores in which 4 (or more) adjacent lanes cover one lane's
// chunk, i.e. whole 64-byte lines (chunks are 64-byte multiples because tstride is a multiple of 16).
// steps collected per flush: the smallest count that makes a lane's chunk a whole number of 64-byte lines
#define PM_STAGE_OF(DW) ((DW) == 2 ? 8 : 16)

template < int W, int PM_LPA > __device__ __forceinline__ void pm_flush_dirs (const uint32_t * stage, uint32_t * const *slab_of_group, int lane, int tstride,
                                                                  int t0)
{
  constexpr int DW = PmSwGeom < W >::DW;
  constexpr int PM_STAGE = PM_STAGE_OF (DW);
  constexpr int CHUNK = PM_STAGE * DW;            // dwords one lane owns per flush
  constexpr int NINSTR = (64 * CHUNK) / (64 * 4);   // 16-byte stores per lane
#pragma unroll 2
  for (int k = 0; k < NINSTR; k++)
    {
      const int o = (k * 64 + lane) * 4;          // dword offset in the staged [64][CHUNK] image
      const int L = o / CHUNK, within = o - L * CHUNK;
      const uint4 v = *(const uint4 *) (stage + o);
      uint32_t *dst = slab_of_group[L / PM_LPA] + ((size_t) (L % PM_LPA) * tstride + t0) * DW + within;
      *(uint4 *) dst = v;
    }
}

template < int W, int PM_LPA, bool DIRS >
__device__ __forceinline__ void pm_sw_forward (const PmSwTask & tk, int bis, int lane, int nn_max, uint32_t * stage,
                                               uint32_t * const *slab_of_group, int tstride, double &best, int &bk, int &bi)
{
  constexpr int DW = PmSwGeom < W >::DW;
  constexpr int PM_STAGE = PM_STAGE_OF (DW);
  int g = lane % PM_LPA;
  // opaque to the optimiser: otherwise the 2 W border doubles below are computed once per kernel, kept live across the
  // persistent task loop and double the register footprint
  asm volatile ("":"+v" (g));
  const int mm = tk.mm, nn = tk.valid ? tk.nn : 0;
  const int pad = PM_LPA * W - mm;
  // ---- match masks of this lane's W columns against reference A, C, G, T, N (bit c = column c of the lane)
  uint64_t mk[5] = { 0, 0, 0, 0, 0 };
  // the lane's W read bytes: all loads issued before the first is used
  uint8_t qraw[W];
#pragma unroll
  for (int c = 0; c < W; c++)
    {
      const int jz = g * W + c - pad;
      const int src = tk.orient ? (mm - 1 - jz) : jz;
      qraw[c] = (jz >= 0 && tk.valid) ? tk.read[src] : (uint8_t) 0;
    }
#pragma unroll
  for (int c = 0; c < W; c++)
    {
      const int jz = g * W + c - pad;      // 0-based read position, negative in the pad
      const uint64_t bitc = 1ull << c;
      if (jz < 0)
        {
          mk[0] |= bitc;
          mk[1] |= bitc;
          mk[2] |= bitc;
          mk[3] |= bitc;
          mk[4] |= bitc;
        }
      else if (tk.valid)
        {
          const uint8_t q = tk.orient ? pm_rc (qraw[c]) : qraw[c];
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
      const int j = g * W + c - pad + 1;   // 1-based read column, <= 0 in the pad
      const double bj = pm_top (j);
      D[c] = bj;
      U1[c] = (j >= 1) ? pm_max (bj - PM_GO, bj - PM_GE) : PM_NEGBIG;
    }
  PmBumpK bumpk;
  bumpk.hi_match = 0x3FF00000u;
  bumpk.hi_miss = 0xBFD55555u;
  bumpk.lo_miss = 0x55555555u;
  double Dprev = pm_top (g * W - pad);     // max3 of the cell left of this lane's first column, row 0
  const double R2in0 = (pad > 0) ? PM_NEGBIG : pm_max (0.0 - PM_GO, -PM_GO - PM_GE);        // S2[i][1] from the border (2079-2081)
  const double Dimp0 = pm_top (-pad);
  double R2out = 0.0, Dout = 0.0;
  double bst = pm_border (mm);     // S[0][0][mm], pemapper.c:1701-1703
  int k_b = 0, i_b = 0;
  // The reference byte of row i is needed by lane g at step i + g: it enters at lane 0 of the group and moves one lane
  // per step with the DP values.  Lane 0 takes it from an 8-byte chunk loaded 8 steps ahead.
  uint32_t r_last = 0;
  uint64_t rw = 0, rw_next = pm_ref_chunk (tk.ref, nn, 0);

  for (int t = 1; t <= nn_max + PM_LPA - 1; t++)
    {
      const int i = t - g;
      double R2in = pm_from_left < PM_LPA > (R2out);
      double Dimp = pm_from_left < PM_LPA > (Dout);
      const uint32_t r_in = pm_from_left < PM_LPA > (r_last);
      if (((t - 1) & 7) == 0)
        {
          rw = rw_next;
          rw_next = pm_ref_chunk (tk.ref, nn, ((t - 1) >> 3) + 1);
        }
      const uint32_t r = (g == 0) ? ((uint32_t) rw & 0xFFu) : r_in;
      rw >>= 8;
      r_last = r;
      R2in = (g == 0) ? R2in0 : R2in;
      Dimp = (g == 0) ? Dimp0 : Dimp;
      const bool act = (i >= 1) && (i <= nn);
      if (act)
        {
          uint64_t msel = (r == 'A') ? mk[0] : (r == 'C') ? mk[1] : (r == 'G') ? mk[2] : (r == 'T') ? mk[3] : mk[4];
          if (__builtin_expect (r != 'A' && r != 'C' && r != 'G' && r != 'T' && r != 'N', 0))
            msel = pm_slow_mask < W > (tk.read, mm, tk.orient, g, pad, (uint8_t) r, bis);
          const uint32_t m0 = (uint32_t) msel, m1 = (uint32_t) (msel >> 32);
          double s2 = R2in;
          uint32_t dw[DW];
#pragma unroll
          for (int d = 0; d < DW; d++)
            dw[d] = 0;
          double o0 = 0.0, o1 = 0.0, o2 = 0.0;
          double s0 = Dprev + pm_bump (m0, 0, bumpk);
#pragma unroll
          for (int c = 0; c < W; c++)
            {
              double s0n = 0.0;
              if (c + 1 < W)
                s0n = D[c] + pm_bump ((c + 1 < 32) ? m0 : m1, (c + 1) & 31, bumpk);   // the old D[c] is the next cell's diagonal
              if (c == W - 1)
                {
                  o0 = s0;
                  o1 = U1[c];
                  o2 = s2;
                }
              pm_cell < DIRS > (s0, s2, U1[c], D[c], dw[c >> 3]);
              s0 = s0n;
            }
          // The last column of lane 7 is read column mm: rows ascending, planes 0,1,2, strict '>' (pemapper.c:1724-1741).
          // Every lane runs the selects (no branch); only lane 7's result is read.
          const bool u0 = o0 > bst;
          bst = u0 ? o0 : bst;
          const bool u1 = o1 > bst;
          bst = u1 ? o1 : bst;
          const bool u2 = o2 > bst;
          bst = u2 ? o2 : bst;
          k_b = u2 ? 2 : (u1 ? 1 : (u0 ? 0 : k_b));
          i_b = (u0 | u1 | u2) ? i : i_b;
          R2out = s2;
          Dout = D[W - 1];
          Dprev = Dimp;
          if (DIRS)
            {
              uint32_t *dst = stage + (lane * PM_STAGE + ((t - 1) & (PM_STAGE - 1))) * DW;
#pragma unroll
              for (int d = 0; d < DW; d++)
                dst[d] = dw[d];
            }
        }
      if (DIRS && (t & (PM_STAGE - 1)) == 0)
        {
          __builtin_amdgcn_fence (__ATOMIC_SEQ_CST, "wavefront");
          __builtin_amdgcn_wave_barrier ();
          pm_flush_dirs < W, PM_LPA > (stage, slab_of_group, lane, tstride, t - PM_STAGE);
          __builtin_amdgcn_fence (__ATOMIC_SEQ_CST, "wavefront");
          __builtin_amdgcn_wave_barrier ();
        }
    }
  if (DIRS)
    {
      const int t_end = nn_max + PM_LPA - 1;
      if (t_end & (PM_STAGE - 1))
        {
          __builtin_amdgcn_fence (__ATOMIC_SEQ_CST, "wavefront");
          __builtin_amdgcn_wave_barrier ();
          pm_flush_dirs < W, PM_LPA > (stage, slab_of_group, lane, tstride, t_end & ~(PM_STAGE - 1));
          __builtin_amdgcn_fence (__ATOMIC_SEQ_CST, "wavefront");
          __builtin_amdgcn_wave_barrier ();
        }
    }
  // hand the tracker's result to lane g == 0 of the group
  const int srcl = min ((lane / PM_LPA) * PM_LPA + PM_LPA - 1, 63);
  best = __shfl (bst, srcl);
  bk = __shfl (k_b, srcl);
  bi = __shfl (i_b, srcl);
}

// y + 1.0 + 1.0 + ... (n times), every addition rounded as the DP rounds it -- without n additions.  While y >= 1 stays inside its
// binade [2^e, 2^(e+1)) an addition of 1.0 is exact (y and 1.0 are multiples of ulp (y), the sum is below 2^(e+1)), so k such steps
// are ONE exact addition of k; only the step that crosses into the next binade rounds, and it is taken on its own with the same
// operands the step-by-step fold has there.  Below 1 (the first steps of a fold that starts at -1/3 or 2/3) every step is taken
// singly.  At most two additions per binade instead of one per read base.
__device__ __forceinline__ double pm_add_ones (double y, int n)
{
  while (n > 0 && y < 1.0)
    {
      y = y + 1.0;
      n--;
    }
  while (n > 0)
    {
      // y >= 1: the steps that stay below the next power of two
      const int e = (int) ((__double2hiint (y) >> 20) & 0x7FF) - 1023;          // 2^e <= y < 2^(e+1); e <= 9 for these scores
      const int top = (2 << e) - 1;                                             // the largest integer part inside the binade
      const int k = min (n, top - (int) y);
      if (k > 0)
        {
          y = y + (double) k;
          n -= k;
        }
      if (n > 0)
        {
          y = y + 1.0;          // into the next binade: the one step that rounds
          n--;
        }
    }
  return y;
}

// ============================================================================================================
// K3a: the gapless rule.  For most reads the affine-gap DP only confirms what a comparison along the window's
// nn - mm + 1 diagonals already shows, and those cases can be decided exactly without it.  Scores (pemapper.c:2006-2095):
// match +1, mismatch -1/3, gap open 2, gap extend 1/36; the read is aligned globally.  A gapless alignment on diagonal d
// with x mismatches scores mm - 4x/3.  An alignment with gaps scores at most: mm - 4 with two or more gaps; mm - 3 with one
// insertion (a read base is lost as well); mm - 2 - (b-1)/36 - 4y/3 with one deletion of b reference bases and y mismatches.
// Let xmin be the smallest x over the diagonals.
//   (1) xmin <= 1.  mm - 4/3 > mm - 2: only diagonals with x <= 1 can hold the best cell of the last read column
//       (pemapper.c:1717-1742), in plane 0, and the DP's value there is the left fold  S0 = S0 + bonus  from S0[.][0] = 0
//       (2062-2081) along the diagonal: any other path into a cell of such a diagonal carries a gap (<= c - 2 at column c)
//       against the fold's >= c - 4/3.  The traceback (1799-1831) stays in plane 0: on the diagonal S0 >= c - 4/3 while
//       S1 <= c - 2 and S2 <= c - 3, so `S1 > S0` and `S2 > max (S0, S1)` are false down to column 0: mm diagonal steps.
//   (2) xmin = 2.  mm - 8/3 is beaten exactly by the alignments with ONE deletion and NO mismatch (mm - 2 ... mm - 2.56):
//       a prefix [0, p) perfect on a diagonal d1 and the suffix [p, mm) perfect on a later one d2.  With pre[d] the
//       length of d's perfect prefix and suf[d] of its perfect suffix, one exists iff pre[d1] + suf[d2] >= mm for some
//       d1 < d2.  If none does, (1)'s statements hold for the diagonals with x = 2: a path with a gap reaches a cell of
//       such a diagonal with more than the fold's c - 8/3 only as (perfect prefix elsewhere, deletion, perfect stretch
//       here past both mismatches), which continued to the end IS such an alignment; the same for S1 on the diagonal.
// In both cases the winner is the first maximum in ascending row order under strict '>' (1724-1741) among the folds,
// which are reproduced addition by addition, so the score is the DP's double to the last bit.  Such a problem gets its
// score, start cell (plane 0, row d + mm) and the PM_GAPLESS flag; the walk kernel then emits mm diagonal steps without
// a direction slab.  Everything else is appended to tasks_dp for pm_sw_kernel.  With 1 % substitutions 81 % of the
// 150-base reads have at most two.  One half-wave per problem, lane = diagonal; wrong diagonals drop out after a few
// bases.
// ============================================================================================================
// ---- eight bases at a time (the gapless rule's comparisons were one byte per trip round a loop: half of the kernel's instructions)
// bit k = byte k of x is not zero (k = 0 the lowest byte).  Per 32-bit word the usual carry trick leaves a byte's top bit set where
// the byte is non-zero; the two words' marks are packed into nibbles and gathered by ONE multiplication: source bits 0, 4, 8 ... 28
// times 2^21 + 2^14 + 2^7 + 1 land on 32 distinct positions (no carries), those wanted on bits 21..28.
__device__ __forceinline__ unsigned pm_nonzero_bytes (uint64_t x)
{
  const uint32_t lo = (uint32_t) x, hi = (uint32_t) (x >> 32);
  const uint32_t a = ((lo & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | lo, b = ((hi & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | hi;
  const uint32_t w = ((a >> 7) & 0x01010101u) | ((b >> 3) & 0x10101010u);
  return ((w * 0x00204081u) >> 21) & 0xFFu;
}

// does a word hold an 'N' or an 'n'?  (non-zero = yes; the bytes above a true hit may be marked as well, which an "is there one" test does not mind)
__device__ __forceinline__ uint32_t pm_has_n (uint32_t v)
{
  const uint32_t y = (v | 0x20202020u) ^ 0x6E6E6E6Eu;
  return (y - 0x01010101u) & ~y & 0x80808080u;
}

// complements of four bases at once: bits 1..2 of 'A' 'C' 'T' 'G' are 0 1 2 3, and v_perm_b32 looks the four complements up in a
// four-byte table.  Any other byte comes out as SOME base, so complementing twice gives the word back iff all four bytes are plain
// upper-case bases: the test that guards the fast path (pm_rc_flat's 'N' for the rest is the slow one's business).
__device__ __forceinline__ uint32_t pm_comp4 (uint32_t v)
{
  return __builtin_amdgcn_perm (0u, 0x43414754u, (v >> 1) & 0x03030303u);     // table bytes 0..3 = 'T' 'G' 'A' 'C'
}

// reverse_transcribe (pemapper.c:2303-2337) of eight read bytes: byte k of the result = complement of byte 7 - k
__device__ __forceinline__ uint64_t pm_rc8 (uint64_t v, int vacated_bits)
{
  // (bytes vacated by the caller's shift hold zeros and are masked out of the result by the caller: as 'A's they pass the fast path's test)
  const uint64_t vv = v | (0x4141414141414141ull & ~(~0ull << vacated_bits));
  const uint32_t lo = (uint32_t) vv, hi = (uint32_t) (vv >> 32);
  const uint32_t clo = pm_comp4 (lo), chi = pm_comp4 (hi);
  if (pm_comp4 (clo) == lo && pm_comp4 (chi) == hi)
    return ((uint64_t) __builtin_amdgcn_perm (0u, clo, 0x00010203u) << 32) | (uint64_t) __builtin_amdgcn_perm (0u, chi, 0x00010203u);
  uint64_t w = 0;
#pragma unroll
  for (int k = 0; k < 8; k++)
    w |= (uint64_t) pm_rc_flat ((uint8_t) (v >> (8 * (7 - k)))) << (8 * k);
  return w;
}

#define PM_GAPLESS 4            // flag in PmHits::stk beside the plane number
#define PM_GL_QUEUE 32           // problems a wave of pm_gapless_kernel collects before it fetches a list's counter

// problems the rule leaves open go to tasks_band (pm_band_kernel, pemap_band.hip.h) when the best diagonal has at most
// PM_BAND_MAXX_ mismatches -- the condition under which the banded DP is exact -- and to tasks_dp (the full DP) otherwise
// (the band's half-width K, its PM_BAND_W = 21 + 2 K + 1 diagonals and the mismatch bound are derived in pemap_band.hip.h:
// K + 3 + K/36 > 4 x / 3  <=>  x <= 6, 5, 4, 3 for K = 5, 4, 3, 2)
#ifndef PM_BAND_K
#define PM_BAND_K 5
#endif
#define PM_BAND_W (22 + 2 * PM_BAND_K)
#define PM_BAND_MAXX_ (PM_BAND_K >= 5 ? 6 : PM_BAND_K + 1)
static_assert (PM_BAND_K >= 2 && PM_BAND_K <= 5, "the band holds at most 32 diagonals");
// (one-wave workgroups: a wave goes wherever a SIMD has room beside the seed kernel's waves; four to a workgroup waited for room for four)
#define PM_GL_BLOCK 64
#define PM_GL_PER_BLOCK (PM_GL_BLOCK / 32)     // problems a workgroup works on at a time: one per half-wave
__global__ __launch_bounds__ (PM_GL_BLOCK, 6) void pm_gapless_kernel (PmIndex ix, PmBatch b, PmParams prm, PmHits h, const uint32_t * tasks,
                                                          const unsigned *n_tasks_p, uint32_t * tasks_dp, unsigned *n_tasks_dp, int max_x,
                                                          uint32_t * tasks_band, unsigned *n_tasks_band)
{
  __shared__ __align__ (8) uint8_t rd[PM_GL_PER_BLOCK][320];
  __shared__ __align__ (8) uint8_t win[PM_GL_PER_BLOCK][352];         // the window: nn <= 299 bytes, read in 8-byte pieces up to 8 bytes past 8-byte boundaries
  __shared__ uint32_t q_band[PM_GL_BLOCK / 64][PM_GL_QUEUE], q_dp[PM_GL_BLOCK / 64][PM_GL_QUEUE];       // per wave: problems on their way to the two DP lists
  const int wv = threadIdx.x >> 6;
  int n_qb = 0, n_qd = 0;
  auto flush_queue = [&] (const uint32_t * q, int &n, uint32_t * dst, unsigned *counter)
  {
    if (n == 0)
      return;
    const unsigned base = (unsigned) __builtin_amdgcn_readfirstlane ((int) ((threadIdx.x & 63) == 0 ? atomicAdd (counter, (unsigned) n) : 0u));
    __builtin_amdgcn_fence (__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier ();
    if ((int) (threadIdx.x & 63) < n)
      dst[base + (threadIdx.x & 63)] = q[threadIdx.x & 63];
    n = 0;
    __builtin_amdgcn_fence (__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier ();
  };
  const int lane = threadIdx.x & 63, l = lane & 31, half = lane & 32;
  const int slot = threadIdx.x >> 5;            // half-wave of the block
  const unsigned n_tasks = *n_tasks_p;
  const int bis = prm.bisulfite;
  const double miss = __hiloint2double ((int) 0xBFD55555u, (int) 0x55555555u);     // -1/3 as the reference's double (pemapper.c:2011-2019)
  // A problem is a chain of dependent trips to memory: its task word, the hit's record (length, window, strand), the read and the
  // window.  The first two are fetched one and two rounds ahead (registers), the last two arrive together and are compared out of
  // LDS: one exposed trip per round instead of four plus one per eight bases of the window.
  const unsigned stride = gridDim.x * (unsigned) PM_GL_PER_BLOCK;
  struct Meta
  {
    size_t o;
    int mm, nn, orient;
    const uint8_t *read, *ref;
  };
  // (no branches around these loads: past the list's end they read its last entry again.  A load under a condition makes the
  // compiler wait for every outstanding load where the paths join, which is at once -- and the round ahead is gone)
  auto load_meta = [&] (uint32_t task_word)->Meta
  {
    Meta m;
    m.o = task_word;
    m.read = pm_read_ptr (b, (int) (m.o / PM_MAX_HITS), &m.mm);
    m.nn = h.nn[m.o];
    m.orient = h.orient[m.o];
    m.ref = ix.genome + h.gpos[m.o];
    return m;
  };
  auto load_task = [&] (unsigned tt)->uint32_t
  {
    return tasks[tt < n_tasks ? tt : n_tasks - 1u];
  };
  if (blockIdx.x * (unsigned) PM_GL_PER_BLOCK >= n_tasks)
    return;
  unsigned t_cur = blockIdx.x * (unsigned) PM_GL_PER_BLOCK + (unsigned) slot;
  Meta nx = load_meta (load_task (t_cur));
  uint32_t task2 = load_task (t_cur + stride);
  for (unsigned t0 = blockIdx.x * (unsigned) PM_GL_PER_BLOCK; t0 < n_tasks; t0 += stride)
    {
      const unsigned t = t0 + (unsigned) slot;
      const bool valid = t < n_tasks;
      const Meta cur = nx;
      const size_t o = cur.o;
      const int mm = cur.mm, nn = cur.nn, orient = cur.orient;
      const uint8_t *read = cur.read, *ref = cur.ref;
      // the next round's record and the task word of the round after it: in flight while this round is compared
      nx = load_meta (task2);
      task2 = load_task (t + 2u * stride);
      // the oriented read, padded with zeros to a multiple of 8, and the window, both in 8-byte pieces (two of each per lane at most:
      // 8 x 64 = 512 bytes) whose loads are ALL issued before the first is waited for.  (The genome buffer and the read rows are
      // padded: reading a few bytes past a window or a read is safe.)  A piece of the reverse strand is the piece of the read that
      // mirrors it, bytes swapped end for end and complemented (reverse_transcribe, pemapper.c:2303-2337); where that piece would
      // start before the read (its last one, when the length is no multiple of 8) it starts at the read and is shifted up.
      {
        uint64_t rq[2] = { 0ull, 0ull }, wq[2] = { 0ull, 0ull };
        int sh_up[2] = { 0, 0 };
#pragma unroll
        for (int u = 0; u < 2; u++)
          {
            const int c = l + 32 * u;
            if (8 * c < mm)
              {
                int s0 = orient ? mm - 8 - 8 * c : 8 * c;
                if (s0 < 0)
                  {
                    sh_up[u] = -8 * s0;
                    s0 = 0;
                  }
                rq[u] = *(const pm_u64_unaligned *) (read + s0);
              }
            if (valid && 8 * c < nn + 16)
              wq[u] = *(const pm_u64_unaligned *) (ref + 8 * c);
          }
#pragma unroll
        for (int u = 0; u < 2; u++)
          {
            const int c = l + 32 * u;
            if (8 * c < ((mm + 7) & ~7))
              {
                uint64_t v = rq[u];
                const int nb = mm - 8 * c < 8 ? mm - 8 * c : 8;
                if (orient)
                  {
                    v <<= sh_up[u];
                    v = pm_rc8 (v, sh_up[u]);
                  }
                if (nb < 8)
                  v &= (1ull << (8 * nb)) - 1ull;
                *(uint64_t *) &rd[slot][8 * c] = v;
              }
            if (valid && 8 * c < nn + 16)
              *(uint64_t *) &win[slot][8 * c] = wq[u];
          }
      }
      __builtin_amdgcn_fence (__ATOMIC_SEQ_CST, "wavefront");
      __builtin_amdgcn_wave_barrier ();
      const int ndiag = valid ? nn - mm + 1 : 0;         // diagonals on which the whole read lies inside the window
      const bool mine = l < ndiag;
      // Which bases of an 8-base piece of the read mismatch the window on diagonal d: piece c = read bases 8 c .. 8 c + 7 against
      // window bytes d + 8 c ..., two aligned pieces of the window shifted together.  Bit k = base 8 c + k mismatches (pm_match's
      // rule: bytes that differ are looked at one by one).
      const int nch = (mm + 7) >> 3;
      auto piece_mask = [&] (int d, int c)->unsigned
      {
        const uint64_t q = *(const uint64_t *) &rd[slot][8 * c];
        const int off = d + 8 * c, sh = 8 * (off & 7);
        const uint64_t *wq = (const uint64_t *) &win[slot][off & ~7];
        const uint64_t lo = wq[0], hi = wq[1];
        const uint64_t r = sh ? ((lo >> sh) | (hi << (64 - sh))) : lo;
        const int nb = mm - 8 * c < 8 ? mm - 8 * c : 8;
        uint64_t x = q ^ r;
        if (nb < 8)
          x &= (1ull << (8 * nb)) - 1ull;
        if (x == 0ull)
          return 0u;
        // bytes that differ mismatch unless pm_match's exceptions apply: an 'N' or 'n' on either side, or bisulfite
        unsigned m = pm_nonzero_bytes (x);
        if (bis | (int) (pm_has_n ((uint32_t) q) | pm_has_n ((uint32_t) (q >> 32)) | pm_has_n ((uint32_t) r) | pm_has_n ((uint32_t) (r >> 32))))
          {
            m = 0u;
            for (int k = 0; k < nb; k++)
              if (((x >> (8 * k)) & 0xFFull) && !pm_match ((uint8_t) (r >> (8 * k)), (uint8_t) (q >> (8 * k)), bis))
                m |= 1u << k;
          }
        return m;
      };
      // Forward: the mismatches of every diagonal, exactly up to PM_BAND_MAXX_ (three decide the rule, PM_BAND_MAXX_ the banded DP),
      // the positions of the first two and of the last one.  Lane = diagonal walking its ~19 pieces, as round 2 had it, was the
      // kernel's instruction count: a wrong diagonal is out after a piece or two, the right one kept the whole half-wave in its loop.
      // Now every diagonal looks at its first two pieces (16 bases; a read has at least 16); the few that have at most
      // PM_BAND_MAXX_ mismatches there are finished one at a time with lane = PIECE, all their pieces in one step.
      int mism = 99, m1 = mm, m2 = mm, mlast = -1;      // mlast: the last mismatch's position, -1 = none; known for the finished ones
      bool finished = false;
      unsigned head = 0u;
      if (mine)
        {
          head = piece_mask (l, 0) | (nch > 1 ? piece_mask (l, 1) << 8 : 0u);
          mism = __popc (head);
          if (head)
            {
              m1 = __ffs ((int) head) - 1;
              mlast = 31 - __clz ((int) head);
              const unsigned h2 = head & (head - 1u);
              if (h2)
                m2 = __ffs ((int) h2) - 1;
            }
        }
      unsigned todo = (unsigned) (__ballot (mine && mism <= PM_BAND_MAXX_) >> half);
      while (__any (todo != 0u))
        {
          const bool go = todo != 0u;           // (per half-wave)
          const int d = go ? __ffs ((int) todo) - 1 : 0;
          todo &= todo - 1u;
          // what diagonal d's lane knows from its first 16 bases, in every lane of the half
          const int d_cnt = __shfl (mism, half + d), d_m1 = __shfl (m1, half + d), d_m2 = __shfl (m2, half + d), d_last = __shfl (mlast, half + d);
          int tot = d_cnt, f1 = d_m1, f2 = d_m2, flast = d_last;
          for (int cb = 2; cb < nch; cb += 32)
            {
              const int c = cb + l;
              const unsigned mk = (go && c < nch) ? piece_mask (d, c) : 0u;
              const int cnt = __popc (mk);
              // inclusive prefix count over the lanes of the half-wave (DPP: row shifts, then row 0 -> 1 and row 2 -> 3)
              int inc = cnt;
              inc += pm_dpp_or < 0x111, 0xF > (0, inc);
              inc += pm_dpp_or < 0x112, 0xF > (0, inc);
              inc += pm_dpp_or < 0x114, 0xF > (0, inc);
              inc += pm_dpp_or < 0x118, 0xF > (0, inc);
              inc += pm_dpp_or < 0x142, 0xA > (0, inc);
              const int exc = inc - cnt;
              const int here = __shfl (inc, half + 31);
              // the first and the second mismatch of the read, if they lie in these pieces: the lane whose pieces hold number
              // (1 or 2) - (mismatches before these pieces) of them
              const int want1 = 1 - tot, want2 = 2 - tot;        // 1-based rank among this pass's mismatches
              const unsigned own1 = (unsigned) (__ballot (want1 >= 1 && exc < want1 && inc >= want1) >> half);
              const unsigned own2 = (unsigned) (__ballot (want2 >= 1 && exc < want2 && inc >= want2) >> half);
              auto nth = [&] (int rank)->int      // position of the rank-th (1-based, <= cnt) set bit of mk
              {
                unsigned t = mk;
                for (int i = 1; i < rank; i++)
                  t &= t - 1u;
                return 8 * c + __ffs ((int) t) - 1;
              };
              const int v1 = nth (want1 - exc > 0 ? want1 - exc : 1), v2 = nth (want2 - exc > 0 ? want2 - exc : 1);
              if (own1)
                f1 = __shfl (v1, half + __ffs ((int) own1) - 1);
              if (own2)
                f2 = __shfl (v2, half + __ffs ((int) own2) - 1);
              const unsigned any = (unsigned) (__ballot (cnt > 0) >> half);
              if (any)
                {
                  const int top = 31 - __clz ((int) any);
                  const int vl = 8 * c + 31 - __clz ((int) (mk ? mk : 1u));
                  flast = __shfl (vl, half + top);
                }
              tot += here;
            }
          if (go && l == d)
            {
              mism = tot;
              m1 = f1;
              m2 = f2;
              mlast = flast;
              finished = true;
            }
        }
      const unsigned c1 = (unsigned) (__ballot (mism <= 1) >> half);      // this half-wave's diagonals with x <= 1
      unsigned cmask = c1;
      if (c1 == 0u && max_x >= 2)
        {
          const unsigned c2 = (unsigned) (__ballot (mism == 2) >> half);
          if (c2 != 0u)
            {
              // case (2): is there a one-deletion alignment without mismatch?  suf[d] by a scan from the read's end
              int suf = 0;
              if (mine && finished)
                suf = mlast < 0 ? mm : mm - 1 - mlast;
              else if (mine)
                {
                  // (a diagonal with seven or more mismatches in its first 16 bases: its perfect suffix is scanned from the end)
                  suf = mm;
                  for (int c = nch - 1; c >= 0 && suf == mm; c--)
                    {
                      const unsigned mk = piece_mask (l, c);
                      if (mk)
                        suf = mm - 1 - (8 * c + 31 - __clz ((int) mk));
                    }
                }
              // pmax = the longest perfect prefix among the earlier diagonals
              const int pre = mine ? m1 : 0;
              int pmax = pre;
              for (int s = 1; s < 32; s <<= 1)
                {
                  const int v = __shfl_up (pmax, s);
                  if (l >= s)
                    pmax = max (pmax, v);
                }
              int pbefore = __shfl_up (pmax, 1);
              if (l == 0)
                pbefore = 0;
              const unsigned del = (unsigned) (__ballot (mine && pbefore + suf >= mm) >> half);
              cmask = del ? 0u : c2;
            }
        }
      // the DP's value at the end of a candidate diagonal: the left fold of the bonuses
      double sc = 0.0;
      if ((cmask >> l) & 1u)
        {
          if (mism >= 1)
            {
              sc = (double) m1 + miss;
              if (mism == 2)
                {
                  sc = pm_add_ones (sc, m2 - m1 - 1);
                  sc = sc + miss;
                  sc = pm_add_ones (sc, mm - m2 - 1);
                }
              else
                sc = pm_add_ones (sc, mm - m1 - 1);
            }
          else
            sc = (double) mm;
        }
      // what the rule leaves open goes to the banded DP's list or to the full DP's, through a queue of the wave in LDS: one fetch of
      // the list's counter per PM_GL_QUEUE problems.  (One per problem -- 70 K same-address atomics per launch, each a round trip
      // that the next one waits behind -- was the kernel's time: the single-address rate is ~85 M per second.)
      {
        int xmin = mism;
        for (int s = 16; s; s >>= 1)
          xmin = min (xmin, __shfl_xor (xmin, s));      // (within the half-wave)
        const bool open = valid && cmask == 0u && l == 0;
        const bool to_band = open && tasks_band && xmin <= PM_BAND_MAXX_, to_dp = open && !to_band;
        const unsigned long long mb = __ballot (to_band), md = __ballot (to_dp);
        if (to_band)
          q_band[wv][n_qb + (int) __popcll (mb & ((1ull << lane) - 1ull))] = (uint32_t) o;
        if (to_dp)
          q_dp[wv][n_qd + (int) __popcll (md & ((1ull << lane) - 1ull))] = (uint32_t) o;
        n_qb += (int) __popcll (mb);
        n_qd += (int) __popcll (md);
        if (n_qb > PM_GL_QUEUE - 2)
          flush_queue (q_band[wv], n_qb, tasks_band, n_tasks_band);
        if (n_qd > PM_GL_QUEUE - 2)
          flush_queue (q_dp[wv], n_qd, tasks_dp, n_tasks_dp);
      }
      if (valid)
        {
          if (cmask != 0u)
            {
              // ascending rows, strict '>' (pemapper.c:1724-1741)
              double best = 0.0;
              int bd = -1;
              unsigned m = cmask;
              while (m)
                {
                  const int dl = __ffs ((int) m) - 1;
                  m &= m - 1u;
                  const double s = __shfl (sc, half + dl);
                  if (bd < 0 || s > best)
                    {
                      best = s;
                      bd = dl;
                    }
                }
              if (l == 0)
                {
                  h.score[o] = best;
                  h.stk[o] = (uint8_t) PM_GAPLESS;
                  h.sti[o] = (int16_t) (bd + mm);
                }
            }
        }
      __builtin_amdgcn_fence (__ATOMIC_SEQ_CST, "wavefront");
      __builtin_amdgcn_wave_barrier ();
    }
  flush_queue (q_band[wv], n_qb, tasks_band, n_tasks_band);
  flush_queue (q_dp[wv], n_qd, tasks_dp, n_tasks_dp);
}

__device__ __forceinline__ int pm_wave_max (int v)
{
  for (int o = 32; o > 0; o >>= 1)
    v = max (v, __shfl_xor (v, o));
  return v;
}

// One persistent wave per 64 / PM_LPA problems.  DIRS: the problems are traceable alignments (the only hit of an end, or the winner
// of a multi-hit end) and write their nibbles to the end's slab.
template < int W, int PM_LPA, bool DIRS > __global__ __launch_bounds__ (64, PM_WAVES_PER_EU (W)) void pm_sw_kernel (PmIndex ix, PmBatch b, PmParams prm, PmHits h,
                                                                                              const uint32_t * tasks,
                                                                                              const unsigned *n_tasks_p, PmCounters * ctr,
                                                                                              uint32_t * dirbuf, uint32_t * dump_slab, int tstride,
                                                                                              int mm_fill, int prio, unsigned *next_task)
{
  constexpr int TPW = 64 / PM_LPA;        // tasks per wave
  __shared__ uint32_t stage[DIRS ? 64 * PM_STAGE_OF (PmSwGeom < W >::DW) * PmSwGeom < W >::DW : 4];
  __shared__ uint32_t *slab_of_group[TPW + 1];       // + 1: the lanes left over when PM_LPA does not divide 64
  // issue priority among the waves sharing the SIMD (PEMAP_SW_PRIO)
  pm_set_prio (prio);
  const int lane = threadIdx.x;
  const int q = lane / PM_LPA;
  const unsigned n_tasks = *n_tasks_p;
  // (a wave without a first task group leaves before it touches the work counter: the launches for the multi-hit ends and
  // the re-scored winners are often all but empty, and 4096 waves adding to one counter cost 0.2 ms each time)
  if (blockIdx.x * (unsigned) TPW >= n_tasks)
    return;
  const size_t slab_dwords = (size_t) PM_LPA * tstride * PmSwGeom < W >::DW;
  // task groups are handed out through a counter, fetched one group ahead (persistent waves start at different times)
  unsigned base_next = gridDim.x * (unsigned) TPW
    + (unsigned) __builtin_amdgcn_readfirstlane ((int) (threadIdx.x == 0 ? atomicAdd (next_task, (unsigned) TPW) : 0u));
  for (unsigned base = blockIdx.x * (unsigned) TPW; base < n_tasks;)
    {
      const unsigned base_cur = base;
      base = base_next;
      if (base < n_tasks)
        base_next = gridDim.x * (unsigned) TPW + (unsigned) __builtin_amdgcn_readfirstlane ((int) (threadIdx.x == 0 ? atomicAdd (next_task, (unsigned) TPW) : 0u));
      __builtin_amdgcn_fence (__ATOMIC_SEQ_CST, "wavefront");
      __builtin_amdgcn_wave_barrier ();
      PmSwTask tk;
      tk.valid = q < TPW && (base_cur + q) < n_tasks;
      size_t o = 0;
      tk.mm = mm_fill;
      tk.nn = 0;
      tk.orient = 0;
      tk.read = b.reads1;
      tk.ref = ix.genome;
      uint32_t *slab = nullptr;
      if (tk.valid)
        {
          o = tasks[base_cur + q];
          int end = (int) (o / PM_MAX_HITS);
          tk.read = pm_read_ptr (b, end, &tk.mm);
          tk.nn = h.nn[o];
          if (tk.nn < 0)
            tk.nn = 0;
          tk.orient = h.orient[o];
          tk.ref = ix.genome + h.gpos[o];
          if (DIRS)
            slab = dirbuf + (size_t) h.slot[end] * slab_dwords;
        }
      if (DIRS)
        {
          // groups without a task flush into the wave's last slab region that is valid: point them at group 0's slab
          // rows beyond the window (never read); simpler: give them the dump slab at the end of the direction buffer
          if ((lane % PM_LPA) == 0)
            slab_of_group[q] = tk.valid ? slab : dump_slab;
          __builtin_amdgcn_fence (__ATOMIC_SEQ_CST, "wavefront");
          __builtin_amdgcn_wave_barrier ();
        }
      int nn_max = pm_wave_max (tk.nn);
      double best;
      int bk, bi;
      pm_sw_forward < W, PM_LPA, DIRS > (tk, prm.bisulfite, lane, nn_max, stage, slab_of_group, tstride, best, bk, bi);
      if (tk.valid && (lane % PM_LPA) == 0)
        {
          h.score[o] = best;
          h.stk[o] = (uint8_t) bk;
          h.sti[o] = (int16_t) bi;
          atomicAdd (DIRS ? &ctr->cells_dirs : &ctr->cells_score, (unsigned long long) tk.nn * tk.mm);
        }
    }
}

// ============================================================================================================
// K4: pair / single-end selection (pemapper.c:1084-1185 and find_mate_pairs 1313-1536), one lane per read (pair).
// Output: the hit to trace per end (or none), the class, and m1/m2 = window start + start[1] + 1 (pemapper.c:1208, 1228).
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

__global__ __launch_bounds__ (64) void pm_select_kernel (PmBatch b, PmParams prm, PmHits h, uint32_t * redo, uint32_t * wins, PmCounters * ctr, uint32_t * m1,
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
                long temp_dist = p1 > p2 ? p1 - p2 : p2 - p1;
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
  for (int which = 0; which < 2; which++)
    {
      const int use = which ? use2 : use1;
      if (use < 0)
        continue;
      const int e = e1 + which;
      const size_t o = (which ? o2 : o1) + use;
      const uint32_t r = (uint32_t) (h.gpos[o] + (uint32_t) h.sti[o]) + 1u;
      if (which)
        r2 = r;
      else
        r1 = r;
      if (h.slot[e] < 0 && !(h.stk[o] & PM_GAPLESS))
        {
          // winner of a multi-hit end (not decided by the gapless rule, whose traceback needs no nibbles): give it a slab
          // and have it scored again with direction nibbles
          h.slot[e] = (int) atomicAdd (&ctr->n_slots, 1u);
          redo[atomicAdd (&ctr->n_redo, 1u)] = (uint32_t) o;
        }
      wins[atomicAdd (&ctr->n_wins, 1u)] = (uint32_t) o;
    }
  m1[it] = r1;
  if (m2)
    m2[it] = r2;
  mtype[it] = code;
}

// ============================================================================================================
// K5: traceback + pileup (smith_waterman_backtrack, pemapper.c:1752-1965).  One lane per winning alignment walks the
// nibbles of its slab from the start cell to the first row or column.  Pileup counters are u32 in HBM updated with
// no-return atomics (the reference's u16 counters wrap; the fetch truncates, which is the same arithmetic).
// Insertions go to a byte log through an atomic cursor.
// ============================================================================================================
__device__ __forceinline__ void pm_log_insertion (uint8_t * ins_log, unsigned ins_cap, PmInsCursor * cur, uint32_t pos, const uint8_t * read,
                                                  int mm, int orient, int j, int ins_len)
{
  // inserted bases = oriented read [j, j + ins_len): collected right to left, stored back in read order (pemapper.c:1892-1893)
  unsigned need = 8u + (((unsigned) ins_len + 3u) & ~3u);
  unsigned at = atomicAdd (&cur->ins_bytes, need);
  if (at + need <= ins_cap)
    {
      *(uint32_t *) (ins_log + at) = pos;
      *(uint32_t *) (ins_log + at + 4) = (uint32_t) ins_len;
      for (int m = 0; m < ins_len; m++)
        ins_log[at + 8 + m] = pm_oriented (read, mm, orient, j + m);
    }
  else
    atomicExch (&cur->ins_overflow, 1u);
}

// The traceback in two kernels.  pm_walk_kernel only follows the direction nibbles and records the walk as 2-bit steps
// (0 diagonal, 1 vertical = deletion, 2 horizontal = inserted base), 32 per 64-bit word, PM_PATH_WORDS(L) words per
// alignment; insertions -- rare -- are logged there and then.  pm_pile_kernel replays the steps with one WAVE per
// alignment, lane = step: row and read column of every step come from prefix popcounts, the pileup increments of an
// alignment go out as a few wave-wide atomic instructions over consecutive positions.  A lane-per-alignment kernel that
// did both had each of its half million walkers hold a direction line, a read line and a pileup line in L2 at once;
// they did not fit, every step refetched its lines and the kernel ran at the HBM limit for random 64-byte lines.
#define PM_PATH_WORDS(L) ((((2 * (L) + 21 + 31) / 32) + 1) & ~1)

template < int W, int PM_LPA > __global__ __launch_bounds__ (64) void pm_walk_kernel (PmBatch b, PmHits h, const uint32_t * wins, PmCounters * ctr,
                                                                           PmInsCursor * cur, const uint32_t * dirbuf, int tstride,
                                                                           PmPile counts, uint8_t * ins_log, unsigned ins_cap,
                                                                           unsigned long long *path, int path_words, uint16_t * n_steps)
{
  constexpr int DW = PmSwGeom < W >::DW;
  const unsigned n_wins = ctr->n_wins;
  const size_t slab_dwords = (size_t) PM_LPA * tstride * DW;
  unsigned long long incs = 0, nins = 0;
  for (unsigned w = blockIdx.x * blockDim.x + threadIdx.x; w < n_wins; w += gridDim.x * blockDim.x)
    {
      const size_t o = wins[w];
      const int end = (int) (o / PM_MAX_HITS);
      int mm;
      const uint8_t *read = pm_read_ptr (b, end, &mm);
      const int orient = h.orient[o];
      const uint32_t gpos = h.gpos[o];
      const uint32_t *slab = dirbuf + (size_t) h.slot[end] * slab_dwords;
      const int pad = PM_LPA * W - mm;
      int k = h.stk[o], i = h.sti[o], j = mm;
      const bool banded = (k & 8) != 0;          // PM_BANDED: the slab holds pm_band_kernel's column-major nibbles
      if (banded)
        k &= 3;
      int i1 = 0, ins_len = 0;
      unsigned long long *pw = path + (size_t) w * path_words;
      if (k & PM_GAPLESS)
        continue;               // decided by pm_gapless_kernel: mm diagonal steps from (plane 0, row i, column mm), which pm_pile_kernel knows
      unsigned long long acc = 0;
      int ns = 0;
      while (i > 0 && j > 0)
        {
          i1 = i - 1;
          const int j1 = j - 1;
          int maxi, maxj, ci, cj;
          // (ci, cj): the cell whose comparisons decide the predecessor plane
          if (k == 0) { maxi = i1; maxj = j1; ci = i1; cj = j1; }
          else if (k == 2) { maxi = i; maxj = j1; ci = i; cj = j1; }
          else { maxi = i1; maxj = j; ci = i1; cj = j; }
          int maxk = 0;
          if (ci >= 1 && cj >= 1)
            {
              uint32_t nib;
              if (banded)
                {
                  const int bb = ci - cj + PM_BAND_K;   // diagonal + K: the optimal path stays inside the band (pemap_band.hip.h)
                  nib = (slab[(size_t) cj * 4 + (bb >> 3)] >> (4 * (7 - (bb & 7)))) & 0xFu;
                }
              else
                {
                  const int J = cj - 1 + pad, gg = J / W, c = J - gg * W;
                  const uint32_t wv = slab[((size_t) gg * tstride + (ci + gg - 1)) * DW + (c >> 3)];
                  const int nd = (W - (c >> 3) * 8) < 8 ? (W - (c >> 3) * 8) : 8;      // cells in this dword
                  nib = (wv >> (4 * (nd - 1 - (c & 7)))) & 0xFu;
                }
              if (k == 0)
                maxk = (nib & 2u) ? 2 : ((nib & 1u) ? 1 : 0);
              else if (k == 2)
                maxk = (nib & 8u) ? 2 : 0;
              else
                maxk = (nib & 4u) ? 1 : 0;
            }
          // on the borders the walk ends after this step (i or j becomes 0) and maxk is never used
          const unsigned long long code = (maxi != i) ? ((maxj != j) ? 0ull : 1ull) : 2ull;
          acc |= code << (2 * (ns & 31));
          ns++;
          if ((ns & 31) == 0)
            {
              pw[(ns >> 5) - 1] = acc;
              acc = 0;
            }
          if (maxi != i)
            {
              if (ins_len > 0)
                {
                  pm_log_insertion (ins_log, ins_cap, cur, gpos + (uint32_t) i1, read, mm, orient, j, ins_len);
                  pm_pile_inc (counts, (size_t) gpos + (size_t) i1, 5);
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
      if (ns & 31)
        pw[ns >> 5] = acc;
      n_steps[w] = (uint16_t) ns;
      if (ins_len > 0 && i >= 1)    // pemapper.c:1918-1958: attached to base[i1] of the last step
        {
          pm_log_insertion (ins_log, ins_cap, cur, gpos + (uint32_t) i1, read, mm, orient, j, ins_len);
          pm_pile_inc (counts, (size_t) gpos + (size_t) i1, 5);
          incs++;
          nins++;
        }
    }
  if (incs)
    atomicAdd (&ctr->pile_incs, incs);
  if (nins)
    atomicAdd (&ctr->n_ins, nins);
}

// second half: the recorded steps of every alignment applied to the pileup (pemapper.c:1840-1870), one wave per alignment.
// An alignment decided by pm_gapless_kernel has no recorded steps: it is mm diagonal steps from (row sti, column mm).
// Lane = step, so neighbouring lanes usually hold neighbouring positions of the same plane (PmPile: the plane of the reference
// base), i.e. the two halves of one word: the even position's lane then adds to both halves at once and its neighbour stays out.
__global__ __launch_bounds__ (64) void pm_pile_kernel (PmBatch b, PmHits h, const uint32_t * wins, PmCounters * ctr, PmPile counts,
                                                       const unsigned long long *path, int path_words, const uint16_t * n_steps)
{
  const int lane = threadIdx.x & 63;
  const unsigned n_wins = ctr->n_wins;
  const unsigned wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = (gridDim.x * blockDim.x) >> 6;
  const unsigned long long below = (1ull << lane) - 1ull;
  unsigned long long incs = 0;
  // an alignment's record (winner -> read, strand, window, start cell, steps) is fetched one round ahead, the winner's word two
  // rounds ahead: the round itself waits for the steps, then the read and reference bytes, then nothing (the adds return nothing)
  struct Rec
  {
    int mm, orient, ns, sti;
    uint32_t gpos;
    bool gapless;
    const uint8_t *read;
  };
  // (no branches around these loads -- past the end they read the last winner again: a load under a condition makes the compiler wait
  // for every outstanding load where the paths join)
  if (n_wins == 0u)
    return;
  auto load_win = [&] (unsigned ww)->uint32_t
  {
    return wins[ww < n_wins ? ww : n_wins - 1u];
  };
  auto load_rec = [&] (unsigned ww, uint32_t win_word)->Rec
  {
    Rec r;
    const size_t o = win_word;
    r.read = pm_read_ptr (b, (int) (o / PM_MAX_HITS), &r.mm);
    r.orient = h.orient[o];
    r.gpos = h.gpos[o];
    r.gapless = (h.stk[o] & PM_GAPLESS) != 0;
    r.ns = n_steps[ww < n_wins ? ww : n_wins - 1u];
    r.sti = h.sti[o];
    return r;
  };
  Rec nx = load_rec (wave, load_win (wave));
  uint32_t win2 = load_win (wave + n_waves);
  for (unsigned w = wave; w < n_wins; w += n_waves)
    {
      const Rec cur = nx;
      nx = load_rec (w + n_waves, win2);
      win2 = load_win (w + 2u * n_waves);
      const int mm = cur.mm, orient = cur.orient;
      const uint8_t *read = cur.read;
      const uint32_t gpos = cur.gpos;
      const bool gapless = cur.gapless;
      const int ns = gapless ? mm : cur.ns;
      const unsigned long long *pw = path + (size_t) w * path_words;
      int i = cur.sti, j = mm;
      for (int s0 = 0; s0 < ns; s0 += 64)
        {
          const int s = s0 + lane;
          int code = 3;
          if (s < ns)
            code = gapless ? 0 : (int) ((pw[s >> 5] >> (2 * (s & 31))) & 3ull);
          const unsigned long long mi = __ballot (code == 0 || code == 1), mj = __ballot (code == 0 || code == 2);
          const int ib = i - __popcll (mi & below), jb = j - __popcll (mj & below);      // row and column before this step
          uint32_t *q = nullptr;
          const size_t pos = (size_t) gpos + (size_t) (ib - 1);
          if (code == 0)
            {
              const uint8_t ch = pm_oriented (read, mm, orient, jb - 1);
              const int slot = (ch == 'A') ? 0 : (ch == 'C') ? 1 : (ch == 'G') ? 2 : (ch == 'T') ? 3 : -1;       // pemapper.c:1850-1857
              if (slot >= 0)
                q = pm_pile_word (counts, pos, pm_pile_plane (counts.genome[pos], slot));
            }
          else if (code == 1)
            q = pm_pile_word (counts, pos, 4);
          incs += q != nullptr;
          // the walk runs towards lower positions: lane + 1 holds pos - 1 after a diagonal or vertical step; with pos odd the two
          // share a word, and this lane (the high half) leaves its increment to that one (the low half's, which reads the old value)
          const unsigned long long qn = __shfl_down ((unsigned long long) (uintptr_t) q, 1);
          const unsigned long long qp = __shfl_up ((unsigned long long) (uintptr_t) q, 1);
          const int odd_n = __shfl_down ((int) (pos & 1), 1), odd_p = __shfl_up ((int) (pos & 1), 1);
          const bool give = q != nullptr && (pos & 1) && lane < 63 && qn == (unsigned long long) (uintptr_t) q && !odd_n;
          const bool take = q != nullptr && !(pos & 1) && lane > 0 && qp == (unsigned long long) (uintptr_t) q && odd_p;
          if (q != nullptr && !give)
            pm_pile_add (q, take ? 0x10001u : (pos & 1) ? 0x10000u : 1u);
          i -= __popcll (mi);
          j -= __popcll (mj);
        }
    }
  for (int o = 32; o; o >>= 1)
    incs += __shfl_xor (incs, o);
  if (lane == 0 && incs)
    atomicAdd (&ctr->pile_incs, incs);
}
