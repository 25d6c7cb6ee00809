// pemap_band.hip.h -- the Smith-Waterman DP of smith_waterman_align (pemapper.c:1694-1748) restricted to a band of diagonals,
// for the problems whose result provably lies inside it.  ONE LANE per alignment (64 alignments per wave), no lane-to-lane
// traffic: the lane walks the read's columns and, per column, the band's 32 diagonals in ascending row order, with the band's
// state (two doubles per diagonal) in registers.
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
// runs down the column in a scalar.  Per cell: 3 subtractions, 1 addition, 4 maxima in fp64 as in pm_cell, plus the four
// comparisons of the direction nibble (same meaning and bit order as pm_sw_kernel's) packed 8 cells to a dword, 4 dwords = 16
// bytes per column, written column-major into the end's direction slab: slab[4 j + b / 8], first cell in the highest nibble.
#pragma once

// PM_BAND_K, PM_BAND_W = 21 + 2 K + 1 and the mismatch bound PM_BAND_MAXX_ are defined beside pm_gapless_kernel (pemap_sw.hip.h)
#define PM_BANDED 8             // flag in PmHits::stk beside the plane number: the direction slab has the band's layout

#ifndef PM_BAND_WAVES_PER_EU
#define PM_BAND_WAVES_PER_EU 2
#endif
template < bool DIRS > __global__ __launch_bounds__ (64, PM_BAND_WAVES_PER_EU) void pm_band_kernel (PmIndex ix, PmBatch b, PmParams prm, PmHits h, const uint32_t * tasks,
                                                                               const unsigned *n_tasks_p, PmCounters * ctr, uint32_t * dirbuf,
                                                                               size_t slab_dwords, unsigned *next_task)
{
  constexpr int K = PM_BAND_K, BW = PM_BAND_W;
  __shared__ uint4 stage[64 * 4];       // [lane][column & 3]: the direction nibbles of four columns
  const int lane = threadIdx.x;
  const unsigned n_tasks = *n_tasks_p;
  if (blockIdx.x * 64u >= n_tasks)
    return;
  const int bis = prm.bisulfite;
  const double miss = __hiloint2double ((int) 0xBFD55555u, (int) 0x55555555u);     // -1/3 as the reference's double
  unsigned base_next = gridDim.x * 64u + (unsigned) __builtin_amdgcn_readfirstlane ((int) (lane == 0 ? atomicAdd (next_task, 64u) : 0u));
  for (unsigned base = blockIdx.x * 64u; base < n_tasks;)
    {
      const unsigned base_cur = base;
      base = base_next;
      if (base < n_tasks)
        base_next = gridDim.x * 64u + (unsigned) __builtin_amdgcn_readfirstlane ((int) (lane == 0 ? atomicAdd (next_task, 64u) : 0u));
      const bool valid = base_cur + (unsigned) lane < n_tasks;
      size_t o = 0;
      int mm = 0, nn = 0, orient = 0;
      const uint8_t *read = b.reads1, *ref = ix.genome;
      uint4 *slab = nullptr;
      if (valid)
        {
          o = tasks[base_cur + lane];
          const int end = (int) (o / PM_MAX_HITS);
          read = pm_read_ptr (b, end, &mm);
          nn = h.nn[o];
          orient = h.orient[o];
          ref = ix.genome + h.gpos[o];
          if (DIRS)
            slab = (uint4 *) (dirbuf + (size_t) h.slot[end] * slab_dwords);
        }
      // (columns 4 g .. 4 g + 3 of this lane's slab; a lane whose read ended before the group has nothing to store, one whose read
      // ends inside it also stores what the stage holds for the columns behind its last one: slots of its own slab nobody reads)
      auto flush_cols = [&] (int g)
      {
        if (valid && 4 * g <= mm)
          {
#pragma unroll
            for (int c = 0; c < 4; c++)
              slab[4 * g + c] = stage[lane * 4 + c];
          }
      };
      // column 0: rows 0 .. nn exist, S0 = S1 = 0, S2 = -go (pemapper.c:2062-2081)
      double M3[BW], E2[BW];
#pragma unroll
      for (int bb = 0; bb < BW; bb++)
        {
          const int i = bb - K;
          const bool ex = valid && i >= 0 && i <= nn;
          M3[bb] = ex ? 0.0 : PM_NEGBIG;
          E2[bb] = ex ? pm_max (0.0 - PM_GO, -PM_GO - PM_GE) : PM_NEGBIG;
        }
      double bst = pm_border (mm > 0 ? mm : 1);        // S[0][0][mm], pemapper.c:1701-1703
      int k_b = 0, i_b = 0;
      const int mm_max = pm_wave_max (mm);
      for (int j = 1; j <= mm_max; j++)
        {
          const bool act = valid && j <= mm;
          const uint8_t qr = act ? read[orient ? (mm - j) : (j - 1)] : (uint8_t) 'A';
          const uint8_t q = orient ? pm_rc (qr) : qr;
          // the window's 32 bytes under this column's band: byte bb = reference base of row i = j + bb - K.  (In the first K columns
          // the address starts up to K + 1 bytes before the window, in the last ones it ends behind it: the genome buffer is padded
          // on both sides, pemap_dev_index_alloc; what is computed from those bytes is never used, see below.)
          const int w0 = j - K - 1;           // window byte of diagonal 0
          uint32_t wnd[8] = { 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u };      // 32 bytes are loaded, the first BW are used
          if (act)
            {
#pragma unroll
              for (int k = 0; k < 4; k++)
                {
                  const uint64_t v = *(const pm_u64_unaligned *) (ref + w0 + 8 * k);
                  wnd[2 * k] = (uint32_t) v;
                  wnd[2 * k + 1] = (uint32_t) (v >> 32);
                }
              // For a plain read letter q without bisulfite, pm_match (r, q) is r == q except that a reference N matches T
              // (init_bonus_matrices' N row, pemapper.c:2013-2023).  Every other column is rewritten so that byte equality with q
              // says what pm_match says: byte = q where the letters match, ~q where they do not.
              uint32_t has_n = 0;
#pragma unroll
              for (int k = 0; k < (BW + 3) / 4; k++)
                {
                  const uint32_t v = wnd[k] ^ 0x4E4E4E4Eu;
                  has_n |= (v - 0x01010101u) & ~v & 0x80808080u;
                }
              const bool q_plain = q == 'A' || q == 'C' || q == 'G' || q == 'T';
              if (!q_plain || bis || has_n)
                {
#pragma unroll
                  for (int k = 0; k < (BW + 3) / 4; k++)
                    {
                      uint32_t w = 0;
#pragma unroll
                      for (int c = 0; c < 4; c++)
                        {
                          const uint8_t r = (uint8_t) (wnd[k] >> (8 * c));
                          w |= (uint32_t) (pm_match (r, q, bis) ? q : (uint8_t) ~ q) << (8 * c);
                        }
                      wnd[k] = w;
                    }
                }
            }
          // Cells of the band outside the matrix need no mask.  Above it (i < 0) a diagonal still holds column 0's -1e30, which
          // absorbs the bonuses; the top border (i == 0, the first K columns only: diagonal K - j) is written over the cell computed
          // there; below it (i > nn) a cell is fed by, and feeds, only rows >= its own (the three moves never go up), so whatever is
          // computed there stays there, and the scan of the last column skips those rows.  A lane past its last column computes on,
          // unread.
          const double bj = pm_border (j);       // S[k][0][j] = -(2 + (j - 1)/36) in all three planes
          const double bj_u = pm_max (bj - PM_GO, bj - PM_GE);
          const int b_top = K - j;              // uniform
          const bool any_last = __ballot (act && j == mm) != 0ull;
          const bool is_last = act && j == mm;
          uint32_t dw[4] = { 0u, 0u, 0u, 0u };
          double U = PM_NEGBIG;         // S1 running down the column: nothing above the band
#pragma unroll
          for (int bb = 0; bb < BW; bb++)
            {
              const uint32_t r = (wnd[bb >> 2] >> (8 * (bb & 3))) & 0xFFu;
              const double s0 = M3[bb] + ((r == (uint32_t) q) ? 1.0 : miss);
              const double s2 = (bb + 1 < BW) ? E2[bb + 1] : PM_NEGBIG;
              const double s1 = U;
              const double a0 = s0 - PM_GO, x1 = s1 - PM_GE, x2 = s2 - PM_GE;
              const double m01 = pm_max (s0, s1);
              // the traceback's four comparisons; the two low bits also name the plane that holds the cell's maximum (below)
              pm_push (dw[bb >> 3], pm_gt (x2, a0));
              pm_push (dw[bb >> 3], pm_gt (x1, a0));
              pm_push (dw[bb >> 3], pm_gt (s2, m01));
              pm_push (dw[bb >> 3], pm_gt (s1, s0));
              M3[bb] = pm_max (m01, s2);
              E2[bb] = pm_max (a0, x2);
              U = pm_max (a0, x1);
              if (bb < K)
                {
                  const bool top = bb == b_top;
                  M3[bb] = top ? bj : M3[bb];
                  U = top ? bj_u : U;
                }
              // (keeps the scheduler from starting many cells' independent halves at once: their doubles in flight cost more registers
              // than the band itself)
              __builtin_amdgcn_sched_barrier (0);
            }
          if constexpr ((BW & 7) != 0)
            dw[(BW >> 3) & 3] <<= 4 * (8 - (BW & 7));        // the last dword's first cell in its highest nibble, like the others
          if (any_last)
            {
              // The scan of the last column (pemapper.c:1724-1741: rows ascending, planes 0, 1, 2, strict '>').  A cell takes the
              // lead iff its maximum is above the running best, and the plane that is left holding it is the first one that attains
              // the cell's maximum: plane 2 if S2 > max (S0, S1), else plane 1 if S1 > S0, else plane 0 -- the cell's two low bits.
              int b_w = -1;
#pragma unroll
              for (int bb = 0; bb < BW; bb++)
                {
                  const int i = j + bb - K;
                  const bool u = is_last && i >= 1 && i <= nn && M3[bb] > bst;
                  bst = u ? M3[bb] : bst;
                  b_w = u ? bb : b_w;
                }
              if (b_w >= 0)
                {
                  const int g = b_w >> 3;
                  const uint32_t d = g == 0 ? dw[0] : g == 1 ? dw[1] : g == 2 ? dw[2] : dw[3];
                  const uint32_t nib = (d >> (4 * (7 - (b_w & 7)))) & 0xFu;
                  k_b = (nib & 2u) ? 2 : (int) (nib & 1u);
                  i_b = j + b_w - K;
                }
            }
          if (DIRS)
            {
              // four columns = one 64-byte line of the slab: staged in LDS and stored together, so that the line is complete in
              // L2 within a few cycles (stored column by column, a line was written back partial, fetched and written again:
              // 1.3 GB fetched + 1.5 GB written per step against 0.6 GB of nibbles)
              stage[lane * 4 + (j & 3)] = make_uint4 (dw[0], dw[1], dw[2], dw[3]);
              if ((j & 3) == 3)
                flush_cols (j >> 2);
            }
        }
      if (DIRS && (mm_max & 3) != 3)
        flush_cols (mm_max >> 2);
      if (valid)
        {
          h.score[o] = bst;
          h.stk[o] = (uint8_t) (k_b | PM_BANDED);
          h.sti[o] = (int16_t) i_b;
        }
      unsigned long long cells = valid ? (unsigned long long) mm * BW : 0ull;
      for (int s = 32; s; s >>= 1)
        cells += __shfl_xor (cells, s);
      if (lane == 0)
        atomicAdd (&ctr->cells_band, cells);
    }
}
