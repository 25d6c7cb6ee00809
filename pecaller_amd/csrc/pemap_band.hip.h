// pemap_band.hip.h -- the Smith-Waterman DP of smith_waterman_align (pemapper.c:1694-1748) restricted to a band of diagonals,
// for the problems whose result provably lies inside it.  FOUR LANES per alignment (16 alignments per wave): lane q of a quad owns
// the band's diagonals 8 q .. 8 q + 7 and walks the read's columns one column behind lane q - 1, its part of the band's state (two
// doubles per diagonal) in registers; two doubles cross between neighbouring lanes per column (DPP quad permutes).
//
// Why a band is exact.  Rows i = 0 .. nn (reference window), columns j = 0 .. mm (read), diagonal d = i - j, D = nn - mm (<= 21:
// the window is the hit +/- MISALIGN_SLOP, pemapper.c:47, 1047-1081).  The read is aligned globally, so a path ends in column mm
// at a row <= nn, i.e. on a diagonal <= D; it starts in column 0 (any row, free: S[0][i][0] = S[1][i][0] = 0, pemapper.c:2062-2081)
// on a diagonal >= 0, or on the top border (0, j0) at the price of an insertion of j0 read bases (S[k][0][j0] = -(2 + (j0-1)/36)).
// A path that visits a cell with d < -K has, up to there, inserted more than K read bases (insertions are the only moves that
// lower d); a path at a cell with d > D + K must still insert more than K to end on a diagonal <= D.  More than K inserted read
// bases cost at least one gap of K + 1 (2 + K/36) and forgo K + 1 match bonuses: such a path scores at most
// mm - (K + 1) - 2 - K/36.  If some diagonal of the window aligns the read with x mismatches, the best score is at least that
// gapless fold, mm - 4x/3 (match +1, mismatch -1/3, pemapper.c:2011-2019).  With K = 5 the first is mm - 8.139 and the second is
// at least mm - 8 for x <= 6: every path that leaves the band [-K, D + K] is strictly worse than the optimum, by 0.139 -- ~10^10
// times the rounding of these sums -- so
//   * the maximum of the last column (pemapper.c:1717-1742: rows ascending, planes 0, 1, 2, strict '>') is attained only at cells of
//     the band and only by paths inside it: the first cell that attains it is the same with the other cells at -infinity;
//   * rounding is monotone, so a cell's DP value is the maximum over the paths into it of the path's own fold, and two
//     alternatives the traceback compares at a cell (1799-1831) share everything after that cell: an alternative whose value
//     comes from a path outside the band would, continued along the optimal suffix, be a full path outside the band, i.e.
//     strictly below the optimum, so it is strictly below the alternative on the optimal path -- as is the same alternative
//     restricted to the band (it can only be lower); alternatives that TIE with the one on the optimal path are attained inside
//     the band (else, again, a path outside would tie the optimum).  Every comparison the traceback makes along the optimal path
//     comes out the same.
// Cells outside the band that this kernel happens to compute (its rectangle of 32 diagonals is a superset of [-K, D + K] when the
// window is clipped) only add real paths; cells it does not compute count as -infinity.  pm_gapless_kernel sends a problem here
// when its rule cannot decide it, the whole read lies inside the window on at least one diagonal, and the best diagonal has at
// most PM_BAND_MAXX mismatches; everything else (reads with a real indel, mostly) goes to the full DP (pm_sw_kernel).
//
// State per diagonal b = d + K at the column just finished: M3[b] = max (S0, S1, S2) of its cell (the diagonal predecessor of
// the next column's cell), E2[b] = max (S0 - go, S2 - ge) of its cell = S2 of the cell to its right (one diagonal down).  S1
// runs down the column.  Per cell: 3 subtractions, 1 addition, 4 maxima in fp64 as in pm_cell, plus the four comparisons of the
// direction nibble (same meaning and bit order as pm_sw_kernel's) packed 8 cells to a dword, 4 dwords = 16 bytes per column,
// written column-major into the end's direction slab: slab[4 j + b / 8], first cell in the highest nibble.
//
// Why four lanes.  Round 2's form gave every alignment ONE lane: 64 alignments per wave, 150 columns x 32 cells in a row = 105 K
// instructions per wave, and a launch's ~62 K problems were fewer than one wave per SIMD -- 0.56 ms per launch alone and 2.0 ms
// beside the seed kernel for work the chip's fp64 rate does in 0.15.  Lane q of a quad now owns cells 8 q .. 8 q + 7 of a column
// and runs one column behind lane q - 1 (at step t it computes column t - q): the cell above its first one (S1 down the column)
// was finished by lane q - 1 a step earlier, the cell to the right of its last one (E2 of diagonal 8 q + 8 in the previous
// column) is the FIRST cell lane q + 1 computes in the same step -- it crosses after that cell, before this lane's last one.
// A quarter of the chain per wave, four times the waves (~4 per SIMD), 70 registers instead of 256.  The arithmetic of a cell,
// the order of the last column's scan (rows ascending: lane 0's cells, then lane 1's ...) and the slab's layout are unchanged.
#pragma once

// PM_BAND_K, PM_BAND_W = 21 + 2 K + 1 and the mismatch bound PM_BAND_MAXX_ are defined beside pm_gapless_kernel (pemap_sw.hip.h)
#define PM_BANDED 8             // flag in PmHits::stk beside the plane number: the direction slab has the band's layout
static_assert (PM_BAND_W == 32, "four lanes x eight diagonals");

#ifndef PM_BAND_WAVES_PER_EU
#define PM_BAND_WAVES_PER_EU 4
#endif

// value of the next / previous lane of the quad (lane 3 / lane 0 read themselves): DPP quad_perm, no LDS
__device__ __forceinline__ int pm_quad_next (int v)
{
  return __builtin_amdgcn_mov_dpp (v, 0xF9, 0xF, 0xF, true);    // quad_perm:[1,2,3,3]
}

__device__ __forceinline__ int pm_quad_prev (int v)
{
  return __builtin_amdgcn_mov_dpp (v, 0x90, 0xF, 0xF, true);    // quad_perm:[0,0,1,2]
}

__device__ __forceinline__ double pm_quad_next (double v)
{
  return __hiloint2double (pm_quad_next (__double2hiint (v)), pm_quad_next (__double2loint (v)));
}

__device__ __forceinline__ double pm_quad_prev (double v)
{
  return __hiloint2double (pm_quad_prev (__double2hiint (v)), pm_quad_prev (__double2loint (v)));
}

template < bool DIRS > __global__ __launch_bounds__ (64, PM_BAND_WAVES_PER_EU) void pm_band_kernel (PmIndex ix, PmBatch b, PmParams prm, PmHits h, const uint32_t * tasks,
                                                                               const unsigned *n_tasks_p, PmCounters * ctr, uint32_t * dirbuf,
                                                                               size_t slab_dwords, unsigned *next_task)
{
  constexpr int K = PM_BAND_K;
  // [problem][group of four columns & 1][column & 3]: the direction nibbles of a 64-byte line of the slab, two lines in the making
  // (lane 0 is three columns ahead of lane 3)
  __shared__ uint4 stage[16 * 2 * 4];
  const int lane = threadIdx.x, q = lane & 3, prob = lane >> 2;
  const unsigned n_tasks = *n_tasks_p;
  if (blockIdx.x * 16u >= n_tasks)
    return;
  const int bis = prm.bisulfite;
  const double miss = __hiloint2double ((int) 0xBFD55555u, (int) 0x55555555u);     // -1/3 as the reference's double
  unsigned base_next = gridDim.x * 16u + (unsigned) __builtin_amdgcn_readfirstlane ((int) (lane == 0 ? atomicAdd (next_task, 16u) : 0u));
  for (unsigned base = blockIdx.x * 16u; base < n_tasks;)
    {
      const unsigned base_cur = base;
      base = base_next;
      if (base < n_tasks)
        base_next = gridDim.x * 16u + (unsigned) __builtin_amdgcn_readfirstlane ((int) (lane == 0 ? atomicAdd (next_task, 16u) : 0u));
      const bool valid = base_cur + (unsigned) prob < n_tasks;
      size_t o = 0;
      int mm = 0, nn = 0, orient = 0;
      const uint8_t *read = b.reads1, *ref = ix.genome;
      uint4 *slab = nullptr;
      if (valid)
        {
          o = tasks[base_cur + prob];
          const int end = (int) (o / PM_MAX_HITS);
          read = pm_read_ptr (b, end, &mm);
          nn = h.nn[o];
          orient = h.orient[o];
          ref = ix.genome + h.gpos[o];
          if (DIRS)
            slab = (uint4 *) (dirbuf + (size_t) h.slot[end] * slab_dwords);
        }
      // column 0: rows 0 .. nn exist, S0 = S1 = 0, S2 = -go (pemapper.c:2062-2081)
      double M3[8], E2[8];
#pragma unroll
      for (int c = 0; c < 8; c++)
        {
          const int i = 8 * q + c - K;
          const bool ex = valid && i >= 0 && i <= nn;
          M3[c] = ex ? 0.0 : PM_NEGBIG;
          E2[c] = ex ? pm_max (0.0 - PM_GO, -PM_GO - PM_GE) : PM_NEGBIG;
        }
      double bst = pm_border (mm > 0 ? mm : 1);        // S[0][0][mm], pemapper.c:1701-1703
      int k_b = 0, i_b = 0;
      double U_last = PM_NEGBIG;        // S1 below this lane's last cell of the column it finished last
      const int mm_max = pm_wave_max (mm);
      // (columns 4 g .. 4 g + 3 of a problem's slab = one 64-byte line: lane q stores column 4 g + q.  A problem whose read ended
      // before the group has nothing to store, one whose read ends inside it also stores what the stage holds for the columns behind
      // its last one: slots of its own slab nobody reads)
      auto flush_cols = [&] (int g)
      {
        pm_wave_sync ();
        if (valid && 4 * g <= mm)
          slab[4 * g + q] = stage[(prob * 2 + (g & 1)) * 4 + q];
      };
      // the read letter and the 8 window bytes of a step are requested a step ahead (no branch around the loads: columns outside
      // 1 .. mm read column 1's or mm's bytes again, unused) -- loaded where they are needed, every step began with a trip to L2
      auto fetch = [&] (int jj, uint8_t & qr_out, uint64_t & w_out)
      {
        const int jc = jj < 1 ? 1 : (jj > mm ? (mm > 0 ? mm : 1) : jj);
        qr_out = read[orient ? (mm - jc) : (jc - 1)];
        w_out = *(const pm_u64_unaligned *) (ref + (jc - K - 1 + 8 * q));
      };
      uint8_t qr_nx;
      uint64_t w_nx;
      fetch (1 - q, qr_nx, w_nx);
      for (int t = 1; t <= mm_max + 3; t++)
        {
          const int j = t - q;                  // this lane's column
          const bool act = valid && j >= 1 && j <= mm;
          const uint8_t qr = qr_nx;
          const uint64_t v = w_nx;
          fetch (j + 1, qr_nx, w_nx);
          uint8_t qc = (uint8_t) 'A';
          uint32_t w_lo = 0u, w_hi = 0u;
          if (act)
            {
              qc = orient ? pm_rc (qr) : qr;
              // the window's 8 bytes under this lane's cells: byte c = reference base of row i = j + 8 q + c - K.  (In the first
              // columns the address starts up to K + 1 bytes before the window, in the last ones it ends behind it: the genome buffer is
              // padded on both sides, pemap_dev_index_alloc; what is computed from those bytes is never used, see below.)
              w_lo = (uint32_t) v;
              w_hi = (uint32_t) (v >> 32);
              // For a plain read letter without bisulfite, pm_match (r, q) is r == q except that a reference N matches T
              // (init_bonus_matrices' N row, pemapper.c:2013-2023).  Every other column is rewritten so that byte equality with q
              // says what pm_match says: byte = q where the letters match, ~q where they do not.
              uint32_t has_n = 0;
              {
                const uint32_t v0 = w_lo ^ 0x4E4E4E4Eu, v1 = w_hi ^ 0x4E4E4E4Eu;
                has_n = ((v0 - 0x01010101u) & ~v0 & 0x80808080u) | ((v1 - 0x01010101u) & ~v1 & 0x80808080u);
              }
              const bool q_plain = qc == 'A' || qc == 'C' || qc == 'G' || qc == 'T';
              if (!q_plain || bis || has_n)
                {
                  uint32_t a0 = 0, a1 = 0;
#pragma unroll
                  for (int c = 0; c < 4; c++)
                    {
                      const uint8_t r0 = (uint8_t) (w_lo >> (8 * c)), r1 = (uint8_t) (w_hi >> (8 * c));
                      a0 |= (uint32_t) (pm_match (r0, qc, bis) ? qc : (uint8_t) ~ qc) << (8 * c);
                      a1 |= (uint32_t) (pm_match (r1, qc, bis) ? qc : (uint8_t) ~ qc) << (8 * c);
                    }
                  w_lo = a0;
                  w_hi = a1;
                }
            }
          // Cells of the band outside the matrix need no mask.  Above it (i < 0) a diagonal still holds column 0's -1e30, which
          // absorbs the bonuses; the top border (i == 0, the first K columns only: diagonal K - j, one of lane 0's) is written over the
          // cell computed there; below it (i > nn) a cell is fed by, and feeds, only rows >= its own (the three moves never go up), so
          // whatever is computed there stays there, and the scan of the last column skips those rows.
          const double bj = pm_border (j >= 1 ? j : 1);       // S[k][0][j] = -(2 + (j - 1)/36) in all three planes
          const double bj_u = pm_max (bj - PM_GO, bj - PM_GE);
          const int b_top = K - j;
          // S1 above this lane's first cell: what lane q - 1 left below its last cell of the same column, a step ago
          const double u_in = pm_quad_prev (U_last);
          // the running best of the last column's scan arrives from the lane above the same way
          const double bst_in = pm_quad_prev (bst);
          const int kb_in = pm_quad_prev (k_b), ib_in = pm_quad_prev (i_b);
          uint32_t dw = 0u;
          double U = q == 0 ? PM_NEGBIG : u_in;        // nothing above the band
          auto cell = [&] (const int c, const double s2)
          {
            const uint32_t r = ((c < 4 ? w_lo : w_hi) >> (8 * (c & 3))) & 0xFFu;
            const double s0 = M3[c] + ((r == (uint32_t) qc) ? 1.0 : miss);
            const double s1 = U;
            const double a0 = s0 - PM_GO, x1 = s1 - PM_GE, x2 = s2 - PM_GE;
            const double m01 = pm_max (s0, s1);
            // the traceback's four comparisons; the two low bits also name the plane that holds the cell's maximum (below)
            pm_push (dw, pm_gt (x2, a0));
            pm_push (dw, pm_gt (x1, a0));
            pm_push (dw, pm_gt (s2, m01));
            pm_push (dw, pm_gt (s1, s0));
            M3[c] = pm_max (m01, s2);
            E2[c] = pm_max (a0, x2);
            U = pm_max (a0, x1);
            if (c < K)
              {
                const bool top = 8 * q + c == b_top;
                M3[c] = top ? bj : M3[c];
                U = top ? bj_u : U;
              }
          };
          // the first cell; then the E2 it leaves crosses to the lane above, whose last cell of this step needs it
          // (E2[1] is read before the cell overwrites anything it depends on: the old E2[1] belongs to the previous column)
          if (act)
            cell (0, E2[1]);
          const double e2_in = pm_quad_next (E2[0]);
          if (act)
            {
#pragma unroll
              for (int c = 1; c < 7; c++)
                {
                  cell (c, E2[c + 1]);
                  __builtin_amdgcn_sched_barrier (0);
                }
              cell (7, q == 3 ? PM_NEGBIG : e2_in);
              U_last = U;
              if (j == mm)
                {
                  // The scan of the last column (pemapper.c:1724-1741: rows ascending, planes 0, 1, 2, strict '>'), continued from
                  // the lane above.  A cell takes the lead iff its maximum is above the running best, and the plane that is left
                  // holding it is the first one that attains the cell's maximum: plane 2 if S2 > max (S0, S1), else plane 1 if
                  // S1 > S0, else plane 0 -- the cell's two low bits.
                  if (q > 0)
                    {
                      bst = bst_in;
                      k_b = kb_in;
                      i_b = ib_in;
                    }
                  int c_w = -1;
#pragma unroll
                  for (int c = 0; c < 8; c++)
                    {
                      const int i = j + 8 * q + c - K;
                      const bool u = i >= 1 && i <= nn && M3[c] > bst;
                      bst = u ? M3[c] : bst;
                      c_w = u ? c : c_w;
                    }
                  if (c_w >= 0)
                    {
                      const uint32_t nib = (dw >> (4 * (7 - c_w))) & 0xFu;
                      k_b = (nib & 2u) ? 2 : (int) (nib & 1u);
                      i_b = j + 8 * q + c_w - K;
                    }
                }
              if (DIRS)
                ((uint32_t *) stage)[((prob * 2 + ((j >> 2) & 1)) * 4 + (j & 3)) * 4 + q] = dw;
            }
          // four columns = one 64-byte line of the slab: staged in LDS and stored together, so that the line is complete in L2 within
          // a few cycles.  Lane 3 finishes column 4 g + 3 at step 4 g + 6.
          if (DIRS && t >= 6 && (t & 3) == 2)
            flush_cols ((t - 6) >> 2);
        }
      if (DIRS)
        {
          // the group the loop has not stored (the last step that stores is 4 g + 6 <= mm_max + 3: at most one group is left)
          const int g_rest = mm_max >= 3 ? ((mm_max - 3) >> 2) + 1 : 0;
          if (4 * g_rest <= mm_max)
            flush_cols (g_rest);
        }
      if (valid && q == 3)
        {
          h.score[o] = bst;
          h.stk[o] = (uint8_t) (k_b | PM_BANDED);
          h.sti[o] = (int16_t) i_b;
        }
      unsigned long long cells = (valid && q == 0) ? (unsigned long long) mm * PM_BAND_W : 0ull;
      for (int s = 32; s; s >>= 1)
        cells += __shfl_xor (cells, s);
      if (lane == 0)
        atomicAdd (&ctr->cells_band, cells);
      pm_wave_sync ();
    }
}
