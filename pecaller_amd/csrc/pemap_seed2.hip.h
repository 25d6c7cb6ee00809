// pemap_seed2.hip.h -- the seed stage split in two kernels so that its HBM-random half and its LDS/VALU half can run
// concurrently (on two HIP streams) with each other and with the fp64 SW kernel of the previous chunk:
//
//   pm_lookup_kernel   one workgroup per read-end: look-ups in pos_index, slice gather from .mdx -> per-end lists of
//                      (diagonal key, segment) in HBM.  Tiny register/LDS footprint, 32 waves per CU: HBM-random bound.
//   pm_vote_kernel     one workgroup per read-end: lists -> LDS, find_matches (pm_vote_strand) -> hits, windows, SW tasks.
//
// Read-ends whose strand holds more than PM_SEED_CAP positions (repeats) are appended to a list and handled afterwards
// by the monolithic pm_seed_kernel (list mode), which has the global spill path.
#pragma once

enum { PM_KIND_NORMAL = 0, PM_KIND_SKIP = 1, PM_KIND_BIG = 2 };

struct PmEndHeader
{
  uint16_t T[2];                       // positions per strand
  uint16_t seg_base[2][PM_MAX_SEG + 1];
  uint8_t kind;
  uint8_t pad[3];
};

struct PmLists
{
  PmEndHeader *hdr;                    // [n_ends]
  uint32_t *key;                       // [n_ends][2][PM_SEED_CAP]
  uint8_t *seg;                        // [n_ends][2][PM_SEED_CAP]
  uint32_t *big_list;                  // ends left to the monolithic kernel
  unsigned *n_big;
  unsigned long long *positions;       // P counter
};

template < int SMAX > struct __align__ (8) PmLookupShared
{
  static constexpr int NITEMS = 2 * SMAX * 49;
  uint32_t it_start[NITEMS];
  uint16_t it_len[NITEMS];
  uint16_t it_off[NITEMS];
  uint32_t kmer[2 * SMAX];
  int seg_cnt[2 * SMAX];
  int seg_base[2][SMAX + 1];
  int offsets[SMAX + 1];
  int ncount;
  uint8_t seq[2][320];
};

template < int SMAX > __global__ __launch_bounds__ (PM_SEED_THREADS, 8) void pm_lookup_kernel (PmIndex ix, PmBatch b, PmParams prm, PmLists out)
{
  typedef PmLookupShared < SMAX > SH;
  __shared__ SH sh;
  constexpr int NI = (SH::NITEMS + PM_SEED_THREADS - 1) / PM_SEED_THREADS;
  const int tid = threadIdx.x;
  const int idepth = ix.idepth;
  const int e = blockIdx.x;
  if (e >= b.n_ends)
    return;
  int len;
  const uint8_t *src = pm_read_ptr (b, e, &len);
  if (tid == 0)
    sh.ncount = 0;
  __syncthreads ();
  // ---- read + reverse complement; N filter (pemapper.c:1552-1559: upper-case 'N' only)
  int isn = 0;
  for (int i = tid; i < len; i += PM_SEED_THREADS)
    {
      uint8_t c = src[i];
      sh.seq[0][i] = c;
      sh.seq[1][len - 1 - i] = pm_rc (c);
      isn += (c == 'N');
    }
  int total_cuts = len / idepth;        // pemapper.c:1573-1587
  if (len % idepth == 0)
    total_cuts--;
  if (total_cuts > SMAX - 1)
    total_cuts = SMAX - 1;
  const int S = total_cuts + 1;
  if (tid <= total_cuts)
    sh.offsets[tid] = (tid < total_cuts || total_cuts == 0) ? tid * idepth : len - idepth;
  if (isn)
    atomicAdd (&sh.ncount, isn);
  __syncthreads ();
  PmEndHeader *hd = &out.hdr[e];
  if (sh.ncount >= 1 + len / 10)
    {
      if (tid == 0)
        hd->kind = PM_KIND_SKIP;
      return;
    }
  if (tid < 2 * S)
    {
      const int strand = tid / S, seg = tid - strand * S;
      const uint8_t *p = &sh.seq[strand][sh.offsets[seg]];
      uint32_t k = 0;
      for (int i = 0; i < 16; i++)
        k = (k << 2) + pm_code (p[i], prm.bisulfite);
      sh.kmer[tid] = k;
    }
  __syncthreads ();
  // ---- 2 x S x 49 bucket look-ups, 8-byte gathers, all in flight (get_mers, pemapper.c:2158-2165)
  {
    uint32_t v0[NI], v1[NI], nbv[NI];
    const uint32_t pos_index_0 = ix.pos_index[0];
#pragma unroll
    for (int r = 0; r < NI; r++)
      {
        const int x = tid + r * PM_SEED_THREADS;
        v0[r] = v1[r] = nbv[r] = 0;
        if (x < 2 * S * 49)
          {
            const int sg = x / 49, j = x - sg * 49;
            const uint32_t nb = pm_neighbour (sh.kmer[sg], j);
            const pm_u32x2 pr = *(const pm_u32x2 *) (ix.pos_index + nb);
            v0[r] = pr.x;
            v1[r] = pr.y;
            nbv[r] = nb;
          }
      }
#pragma unroll
    for (int r = 0; r < NI; r++)
      {
        const int x = tid + r * PM_SEED_THREADS;
        if (x < 2 * S * 49)
          {
            // the all-T k-mer's successor is entry 0: `which + 1` is evaluated in 32 bits (pemapper.c:2163)
            const uint32_t ln = ((nbv[r] == 0xFFFFFFFFu) ? pos_index_0 : v1[r]) - v0[r];
            sh.it_start[x] = v0[r];
            sh.it_len[x] = (ln >= PM_TOO_MANY) ? 0xFFFF : (uint16_t) ln;
          }
      }
  }
  __syncthreads ();
  // ---- a segment with any bucket >= too_many_spots is emptied (pemapper.c:1602-1606)
  if (tid < 2 * S)
    {
      int sum = 0;
      bool bad = false;
      for (int j = 0; j < 49; j++)
        {
          const uint16_t ln = sh.it_len[tid * 49 + j];
          sh.it_off[tid * 49 + j] = (uint16_t) sum;
          if (ln == 0xFFFF)
            bad = true;
          else
            sum += ln;
        }
      sh.seg_cnt[tid] = bad ? 0 : sum;
    }
  __syncthreads ();
  if (tid < 2)
    {
      int acc = 0;
      for (int s = 0; s < S; s++)
        {
          sh.seg_base[tid][s] = acc;
          acc += sh.seg_cnt[tid * S + s];
        }
      sh.seg_base[tid][S] = acc;
    }
  __syncthreads ();
  const int T0 = sh.seg_base[0][S], T1 = sh.seg_base[1][S];
  if (T0 > PM_SEED_CAP || T1 > PM_SEED_CAP)
    {
      if (tid == 0)
        {
          hd->kind = PM_KIND_BIG;
          out.big_list[atomicAdd (out.n_big, 1u)] = (uint32_t) e;
        }
      return;
    }
  if (tid == 0)
    {
      hd->kind = PM_KIND_NORMAL;
      hd->T[0] = (uint16_t) T0;
      hd->T[1] = (uint16_t) T1;
      atomicAdd (out.positions, (unsigned long long) (T0 + T1));
    }
  if (tid < 2 * (S + 1))
    {
      const int st = tid / (S + 1), k = tid - st * (S + 1);
      hd->seg_base[st][k] = (uint16_t) sh.seg_base[st][k];
    }
  // ---- bucket slices -> (diagonal key, segment) lists, one position per thread and round
  uint32_t *okey = out.key + (size_t) e * 2 * PM_SEED_CAP;
  uint8_t *oseg = out.seg + (size_t) e * 2 * PM_SEED_CAP;
  for (int pp = tid; pp < T0 + T1; pp += PM_SEED_THREADS)
    {
      const int strand = pp < T0 ? 0 : 1;
      const int p = strand ? pp - T0 : pp;
      int seg = 0;
      while (p >= sh.seg_base[strand][seg + 1])
        seg++;
      const int q = p - sh.seg_base[strand][seg];
      const int x0 = (strand * S + seg) * 49;
      int lo = 0, hi = 48;      // largest j with it_off[j] <= q: that slice holds position q
      while (lo < hi)
        {
          const int mid = (lo + hi + 1) >> 1;
          if ((int) sh.it_off[x0 + mid] <= q)
            lo = mid;
          else
            hi = mid - 1;
        }
      const uint32_t m = ix.mers[sh.it_start[x0 + lo] + (uint32_t) (q - (int) sh.it_off[x0 + lo])];
      okey[strand * PM_SEED_CAP + p] = m + (uint32_t) (PM_DIAG_BIAS - sh.offsets[seg]);
      oseg[strand * PM_SEED_CAP + p] = (uint8_t) seg;
    }
}

// ---- the same work with ONE WAVE per read-end, persistent.  Random 64-byte lines are a DRAM-side limit (about 47 G/s on
// MI355X however the requests are issued), reached with a few hundred lines in flight per CU; a wave that keeps its whole
// end's look-ups in flight (NI per lane) needs no help from occupancy, so a handful of these waves per CU saturate the
// limit and the rest of the CU stays free for the vote / SW / walk kernels running beside them.  No workgroup barriers:
// the wave's LDS traffic is ordered by the hardware, pm_wave_sync only stops the compiler from reordering it.
__device__ __forceinline__ void pm_wave_sync ()
{
  __builtin_amdgcn_fence (__ATOMIC_SEQ_CST, "wavefront");
  __builtin_amdgcn_wave_barrier ();
}

template < int SMAX > struct __align__ (8) PmLookupWaveShared
{
  static constexpr int NITEMS = 2 * SMAX * 49;
  uint32_t it_start[NITEMS];
  uint16_t it_off[NITEMS];            // slice lengths (0xFFFF = too many), then their exclusive prefix inside the segment
  uint32_t kmer[2 * SMAX];
  int seg_base[2][SMAX + 1];
  int offsets[SMAX + 1];
  uint8_t seq[2][320];
};

template < int SMAX, int PM_LW_BATCH > __global__ __launch_bounds__ (64) void pm_lookup_wave_kernel (PmIndex ix, PmBatch b, PmParams prm, PmLists out)
{
  typedef PmLookupWaveShared < SMAX > SH;
  __shared__ SH sh;
  const int lane = threadIdx.x;
  const int idepth = ix.idepth;
  const uint32_t pos_index_0 = ix.pos_index[0];
  unsigned long long n_pos = 0;
  for (int e = blockIdx.x; e < b.n_ends; e += gridDim.x)
    {
      int len;
      const uint8_t *src = pm_read_ptr (b, e, &len);
      // ---- read + reverse complement; N filter (pemapper.c:1552-1559: upper-case 'N' only)
      int isn = 0;
      for (int i = lane; i < len; i += 64)
        {
          const uint8_t c = src[i];
          sh.seq[0][i] = c;
          sh.seq[1][len - 1 - i] = pm_rc (c);
          isn += (c == 'N');
        }
      for (int o = 32; o; o >>= 1)
        isn += __shfl_xor (isn, o);
      int total_cuts = len / idepth;    // pemapper.c:1573-1587
      if (len % idepth == 0)
        total_cuts--;
      if (total_cuts > SMAX - 1)
        total_cuts = SMAX - 1;
      const int S = total_cuts + 1;
      PmEndHeader *hd = &out.hdr[e];
      if (isn >= 1 + len / 10)
        {
          if (lane == 0)
            hd->kind = PM_KIND_SKIP;
          continue;
        }
      if (lane <= total_cuts)
        sh.offsets[lane] = (lane < total_cuts || total_cuts == 0) ? lane * idepth : len - idepth;
      pm_wave_sync ();
      if (lane < 2 * S)
        {
          const int strand = lane / S, seg = lane - strand * S;
          const uint8_t *p = &sh.seq[strand][sh.offsets[seg]];
          uint32_t k = 0;
#pragma unroll 4
          for (int i = 0; i < 16; i++)
            k = (k << 2) + pm_code (p[i], prm.bisulfite);
          sh.kmer[lane] = k;
        }
      pm_wave_sync ();
      // ---- 2 x S x 49 bucket look-ups (get_mers, pemapper.c:2158-2165): 8-byte gathers, PM_LW_BATCH rounds (x 64 lanes) in
      //      flight at a time -- a few such waves per CU already hold more lines in flight than DRAM turns around, and the
      //      small register footprint is what lets the fp64 SW waves stay resident beside them
#pragma unroll 1
      for (int r0 = 0; r0 * 64 < 2 * S * 49; r0 += PM_LW_BATCH)
        {
          uint32_t v0[PM_LW_BATCH], v1[PM_LW_BATCH];
#pragma unroll
          for (int r = 0; r < PM_LW_BATCH; r++)
            {
              const int x = lane + (r0 + r) * 64;
              v0[r] = v1[r] = 0;
              if (x < 2 * S * 49)
                {
                  const int sg = x / 49, j = x - sg * 49;
                  const pm_u32x2 pr = *(const pm_u32x2 *) (ix.pos_index + pm_neighbour (sh.kmer[sg], j));
                  v0[r] = pr.x;
                  v1[r] = pr.y;
                }
            }
#pragma unroll
          for (int r = 0; r < PM_LW_BATCH; r++)
            {
              const int x = lane + (r0 + r) * 64;
              if (x < 2 * S * 49)
                {
                  // the all-T k-mer's successor is entry 0: `which + 1` is evaluated in 32 bits (pemapper.c:2163)
                  const int sg = x / 49, j = x - sg * 49;
                  const uint32_t nb = pm_neighbour (sh.kmer[sg], j);
                  const uint32_t ln = ((nb == 0xFFFFFFFFu) ? pos_index_0 : v1[r]) - v0[r];
                  sh.it_start[x] = v0[r];
                  sh.it_off[x] = (ln >= PM_TOO_MANY) ? 0xFFFF : (uint16_t) ln;
                }
            }
        }
      pm_wave_sync ();
      // ---- a segment with any bucket >= too_many_spots is emptied (pemapper.c:1602-1606); one lane per segment
      int cnt = 0;
      if (lane < 2 * S)
        {
          int sum = 0;
          bool bad = false;
#pragma unroll 7
          for (int j = 0; j < 49; j++)
            {
              const uint16_t ln = sh.it_off[lane * 49 + j];
              sh.it_off[lane * 49 + j] = (uint16_t) sum;
              if (ln == 0xFFFF)
                bad = true;
              else
                sum += ln;
            }
          cnt = bad ? 0 : sum;
        }
      // exclusive prefix of the segment counts inside each strand (lanes 0..S-1 and S..2S-1)
      int inc = cnt;
      for (int o = 1; o < 64; o <<= 1)
        {
          const int t = __shfl_up (inc, o);
          if (lane >= o)
            inc += t;
        }
      const int T0 = __shfl (inc, S - 1), TT = __shfl (inc, 2 * S - 1);
      const int T1 = TT - T0;
      if (lane < 2 * S)
        {
          const int strand = lane / S, seg = lane - strand * S;
          sh.seg_base[strand][seg] = inc - cnt - (strand ? T0 : 0);
        }
      if (lane == 0)
        {
          sh.seg_base[0][S] = T0;
          sh.seg_base[1][S] = T1;
        }
      pm_wave_sync ();
      if (T0 > PM_SEED_CAP || T1 > PM_SEED_CAP)
        {
          if (lane == 0)
            {
              hd->kind = PM_KIND_BIG;
              out.big_list[atomicAdd (out.n_big, 1u)] = (uint32_t) e;
            }
          continue;
        }
      if (lane == 0)
        {
          hd->kind = PM_KIND_NORMAL;
          hd->T[0] = (uint16_t) T0;
          hd->T[1] = (uint16_t) T1;
        }
      n_pos += (unsigned long long) TT;
      if (lane < 2 * (S + 1))
        {
          const int st = lane / (S + 1), k = lane - st * (S + 1);
          hd->seg_base[st][k] = (uint16_t) sh.seg_base[st][k];
        }
      // ---- bucket slices -> (diagonal key, segment) lists, one position per lane and round
      uint32_t *okey = out.key + (size_t) e * 2 * PM_SEED_CAP;
      uint8_t *oseg = out.seg + (size_t) e * 2 * PM_SEED_CAP;
#pragma unroll 2
      for (int pp = lane; pp < TT; pp += 64)
        {
          const int strand = pp < T0 ? 0 : 1;
          const int p = strand ? pp - T0 : pp;
          int seg = 0;
          while (p >= sh.seg_base[strand][seg + 1])
            seg++;
          const int q = p - sh.seg_base[strand][seg];
          const int x0 = (strand * S + seg) * 49;
          int lo = 0, hi = 48;  // largest j with it_off[j] <= q: that slice holds position q
          while (lo < hi)
            {
              const int mid = (lo + hi + 1) >> 1;
              if ((int) sh.it_off[x0 + mid] <= q)
                lo = mid;
              else
                hi = mid - 1;
            }
          const uint32_t m = ix.mers[sh.it_start[x0 + lo] + (uint32_t) (q - (int) sh.it_off[x0 + lo])];
          okey[strand * PM_SEED_CAP + p] = m + (uint32_t) (PM_DIAG_BIAS - sh.offsets[seg]);
          oseg[strand * PM_SEED_CAP + p] = (uint8_t) seg;
        }
      pm_wave_sync ();
    }
  if (lane == 0 && n_pos)
    atomicAdd (out.positions, n_pos);
}

// LDS of the vote kernel: one strand at a time (the lists come from HBM), no look-up items
template < int SMAX > struct __align__ (8) PmVoteShared
{
  union
  {
    uint32_t table[PM_SEED_TABLE];
  } u;
  uint32_t ekey[1][PM_SEED_CAP];
  uint32_t bkey[PM_SEED_CAP];
  uint32_t hits[PM_MAX_HITS];
  uint32_t wsum[4];
  int seg_cnt[2 * SMAX];
  int seg_base[2][SMAX + 1];
  int offsets[1][SMAX + 1];
  int state[4];
  unsigned n_surv;
  uint16_t surv[PM_SEED_CAP];
  uint16_t order[PM_SEED_CAP];
  uint16_t hits_off[PM_MAX_HITS];
  uint8_t eseg[1][PM_SEED_CAP];
  uint8_t bseg[PM_SEED_CAP];
  uint8_t tfs[PM_SEED_CAP];
  uint8_t hits_or[PM_MAX_HITS];
};

template < int SMAX > __global__ __launch_bounds__ (PM_SEED_THREADS) void pm_vote_kernel (PmIndex ix, PmBatch b, PmParams prm, PmHits h,
                                                                                          uint32_t * tasks_s, uint32_t * tasks_m,
                                                                                          PmCounters * ctr, PmLists in, int probe)
{
  typedef PmVoteShared < SMAX > SH;
  __shared__ SH sh;
  const int tid = threadIdx.x;
  const int idepth = ix.idepth;
  const int max_off = max (2, idepth - 4);
  const int e = blockIdx.x;
  if (e >= b.n_ends)
    return;
  const PmEndHeader *hd = &in.hdr[e];
  const int kind = hd->kind;
  if (kind == PM_KIND_BIG)
    return;                     // left to pm_seed_kernel in list mode
  int len;
  (void) pm_read_ptr (b, e, &len);
  int total_cuts = len / idepth;
  if (len % idepth == 0)
    total_cuts--;
  if (total_cuts > SMAX - 1)
    total_cuts = SMAX - 1;
  const int S = total_cuts + 1;
  int tot = 0;
  if (kind == PM_KIND_NORMAL)
    {
      const int T0 = hd->T[0], T1 = hd->T[1];
      if (tid <= total_cuts)
        sh.offsets[0][tid] = (tid < total_cuts || total_cuts == 0) ? tid * idepth : len - idepth;
      if (tid < 2 * (S + 1))
        {
          const int st = tid / (S + 1), k = tid - st * (S + 1);
          sh.seg_base[st][k] = hd->seg_base[st][k];
        }
      const uint32_t *ikey = in.key + (size_t) e * 2 * PM_SEED_CAP;
      const uint8_t *iseg = in.seg + (size_t) e * 2 * PM_SEED_CAP;
      // strand 1's list is prefetched into registers while strand 0 is voted on
      constexpr int NP = PM_SEED_CAP / PM_SEED_THREADS;
      uint32_t k1[NP];
      uint8_t s1[NP];
#pragma unroll
      for (int r = 0; r < NP; r++)
        {
          const int p = tid + r * PM_SEED_THREADS;
          k1[r] = p < T1 ? ikey[PM_SEED_CAP + p] : 0u;
          s1[r] = p < T1 ? iseg[PM_SEED_CAP + p] : (uint8_t) 0;
        }
      for (int p = tid; p < T0; p += PM_SEED_THREADS)
        {
          sh.ekey[0][p] = ikey[p];
          sh.eseg[0][p] = iseg[p];
        }
      __syncthreads ();
      if (tid < 2 * S)
        {
          const int st = tid / S, k = tid - st * S;
          sh.seg_cnt[tid] = sh.seg_base[st][k + 1] - sh.seg_base[st][k];
        }
      __syncthreads ();
      int min_match = max (1, total_cuts);   // pemapper.c:1642-1645
      if (total_cuts > 4)
        min_match = (4 * total_cuts) / 5;
      min_match = min (min_match, 4);
      bool go_on = true;
      for (int strand = 0; strand < 2 && go_on && probe != 1; strand++)
        {
          if (strand == 1)
            {
#pragma unroll
              for (int r = 0; r < NP; r++)
                {
                  const int p = tid + r * PM_SEED_THREADS;
                  if (p < T1)
                    {
                      sh.ekey[0][p] = k1[r];
                      sh.eseg[0][p] = s1[r];
                    }
                }
              pm_lds_barrier ();
            }
          pm_vote_strand < SH, uint16_t > (sh, sh.ekey[0], sh.eseg[0], sh.bkey, sh.bseg, sh.surv, sh.order, sh.tfs, strand ? T1 : T0,
                                           &sh.seg_cnt[strand * S], sh.offsets[0], total_cuts, max_off, min_match, tot, go_on, (uint8_t) strand, probe);
          if (tot >= PM_MAX_HITS)
            go_on = false;
          pm_lds_barrier ();
        }
    }
  pm_seed_emit (sh, h, e, tot);
}
