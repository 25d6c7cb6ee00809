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
  unsigned *next_end;                  // work counter of the persistent look-up waves (ends beyond the first grid-ful)
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

// one read-end of the wave-per-end look-up kernel
template < int SMAX, int PM_LW_BATCH, class SH >
__device__ __forceinline__ void pm_lookup_one_end (SH & sh, const PmIndex & ix, const PmBatch & b, const PmParams & prm, const PmLists & out, const int e,
                                                   const int lane, const int idepth, const uint32_t pos_index_0, unsigned long long &n_pos)
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
      return;
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
      return;
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

template < int SMAX, int PM_LW_BATCH > __global__ __launch_bounds__ (64) void pm_lookup_wave_kernel (PmIndex ix, PmBatch b, PmParams prm, PmLists out, int prio)
{
  pm_set_prio (prio);
  typedef PmLookupWaveShared < SMAX > SH;
  __shared__ SH sh;
  const int lane = threadIdx.x;
  const int idepth = ix.idepth;
  const uint32_t pos_index_0 = ix.pos_index[0];
  unsigned long long n_pos = 0;
  // Ends are handed out through a counter, fetched one end ahead: persistent waves start whenever the other stream's kernels
  // leave them a slot, and a static stride would make the latest starter the whole kernel's tail.
  int e_next = (int) gridDim.x + (int) __builtin_amdgcn_readfirstlane ((int) (lane == 0 ? atomicAdd (out.next_end, 1u) : 0u));
  for (int e = blockIdx.x; e < b.n_ends;)
    {
      const int e_cur = e;
      e = e_next;
      if (e < b.n_ends)
        e_next = (int) gridDim.x + (int) __builtin_amdgcn_readfirstlane ((int) (lane == 0 ? atomicAdd (out.next_end, 1u) : 0u));
      pm_lookup_one_end < SMAX, PM_LW_BATCH > (sh, ix, b, prm, out, e_cur, lane, idepth, pos_index_0, n_pos);
    }
  if (lane == 0 && n_pos)
    atomicAdd (out.positions, n_pos);
}

// ---- the look-ups against the 8 replicas of the table (pemap_aux.hip.h): ONE WAVE per read-end, persistent, same output
// as pm_lookup_wave_kernel.  A (strand, segment)'s k-mer and its 48 neighbours lie in 8 lines, one per replica: the
// wave reads each line whole -- 4 adjacent lanes x 16 bytes, 16 lines per wave instruction, every line of the read-end
// in flight at once -- into LDS, decodes the 2 x S x 49 entries from there, then fetches the record headers of the
// buckets of 2..99 positions (the only second-level requests) and copies the positions out.  2 x S x 8 + (multi
// buckets) fabric requests per read-end against 2 x S x 43 + (non-empty buckets) for the reference's layout.
template < int SMAX > struct __align__ (16) PmLookupRepShared
{
  static constexpr int NITEMS = 2 * SMAX * 49;
  static constexpr int NLINES = 2 * SMAX * 8;
  uint32_t lines[NLINES * 16];
  uint32_t it_start[NITEMS];          // the entry: a position, or the code of a record
  uint16_t it_off[NITEMS];            // bucket sizes (0xFFFF = too many), then their exclusive prefix inside the segment
  uint16_t mlist[NITEMS];             // items whose entry points to a record
  uint32_t kmer[2 * SMAX];
  int seg_base[2][SMAX + 1];
  int offsets[SMAX + 1];
  uint8_t seq[2][320];
};

template < int SMAX, int PM_LR_BATCH, class SH >
__device__ __forceinline__ void pm_lookup_rep_one_end (SH & sh, const PmIndex & ix, const PmBatch & b, const PmParams & prm, const PmLists & out, const int e,
                                                       const int lane, const int idepth, unsigned long long &n_pos)
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
      return;
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
  // ---- the 2 x S x 8 lines, 16 per wave instruction, PM_LR_BATCH instructions (x 16 lines) in flight at a time
  {
    const int n_lines = 2 * S * 8;
    const int quarter = lane & 3;
#pragma unroll 1
    for (int r0 = 0; r0 * 16 < n_lines; r0 += PM_LR_BATCH)
      {
        uint4 v[PM_LR_BATCH];
#pragma unroll
        for (int r = 0; r < PM_LR_BATCH; r++)
          {
            const int li = (r0 + r) * 16 + (lane >> 2);
            v[r] = make_uint4 (0u, 0u, 0u, 0u);
            if (li < n_lines)
              {
                const int sg = li >> 3, p = li & 7;
                const uint32_t idx = pm_swap_fields (sh.kmer[sg], p);
                v[r] = *(const uint4 *) (ix.rep + ((size_t) p << 32) + (size_t) (idx & ~15u) + (size_t) (quarter * 4));
              }
          }
#pragma unroll
        for (int r = 0; r < PM_LR_BATCH; r++)
          {
            const int li = (r0 + r) * 16 + (lane >> 2);
            if (li < n_lines)
              *(uint4 *) (&sh.lines[li * 16 + quarter * 4]) = v[r];
          }
      }
  }
  pm_wave_sync ();
  // ---- the 2 x S x 49 entries (get_mers, pemapper.c:2158-2165, in fill_mers' order): one (strand, segment) per round, lane j =
  //      neighbour j, so that which field a lane replaces, by which alternative, and in which replica's line the entry lies
  //      are constants of the lane (pm_neighbour written out)
  int n_multi = 0;
  const uint32_t multi_base = ix.multi_base;
  {
    const int nb_f = lane > 0 ? (lane - 1) / 3 : 0;
    const uint32_t nb_a = lane > 0 ? (uint32_t) ((lane - 1) % 3) : 0u;
    const uint32_t nb_sh = 2u * (uint32_t) (nb_f & 15);
    const uint32_t nb_keep = lane > 0 ? ~(3u << nb_sh) : 0xFFFFFFFFu;  // lane 0: the k-mer itself
    const uint32_t nb_alt_on = lane > 0 ? 0xFFFFFFFFu : 0u;
    const int nb_p = (nb_f >> 1) & 7;
    const uint32_t nb_p4 = 4u * (uint32_t) nb_p, nb_pw = 16u * (uint32_t) nb_p;
    for (int sg = 0; sg < 2 * S; sg++)
      {
        const int x = sg * 49 + lane;
        bool is_multi = false;
        if (lane < 49)
          {
            const uint32_t k = sh.kmer[sg];
            const uint32_t cur = (k >> nb_sh) & 3u;
            const uint32_t alt = nb_a + (nb_a >= cur ? 1u : 0u);
            const uint32_t nbk = (k & nb_keep) | ((alt << nb_sh) & nb_alt_on);
            const uint32_t ent = sh.lines[sg * 128 + nb_pw + ((nbk >> nb_p4) & 15u)];
            sh.it_start[x] = ent;
            is_multi = ent >= multi_base && ent < 0xFFFFFFFEu;
            sh.it_off[x] = (ent == 0xFFFFFFFFu) ? (uint16_t) 0 : (ent == 0xFFFFFFFEu) ? (uint16_t) 0xFFFF : (uint16_t) 1;
          }
        const unsigned long long bal = __ballot (is_multi);
        if (is_multi)
          sh.mlist[n_multi + __popcll (bal & ((1ull << lane) - 1ull))] = (uint16_t) x;
        n_multi += __popcll (bal);
      }
  }
  pm_wave_sync ();
  // ---- sizes of the buckets that have a record: its first 16 bytes, 4 rounds in flight
#pragma unroll 1
  for (int i0 = 0; i0 < n_multi; i0 += 256)
    {
      uint32_t cnt[4];
#pragma unroll
      for (int r = 0; r < 4; r++)
        {
          const int i = i0 + r * 64 + lane;
          cnt[r] = 0;
          if (i < n_multi)
            cnt[r] = ix.multi[(size_t) (sh.it_start[sh.mlist[i]] - multi_base) * 4];
        }
#pragma unroll
      for (int r = 0; r < 4; r++)
        {
          const int i = i0 + r * 64 + lane;
          if (i < n_multi)
            sh.it_off[sh.mlist[i]] = (uint16_t) cnt[r];
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
      return;
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
  // ---- positions -> (diagonal key, segment) lists, one position per lane and round
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
      int lo = 0, hi = 48;  // largest j with it_off[j] <= q: that bucket holds position q
      while (lo < hi)
        {
          const int mid = (lo + hi + 1) >> 1;
          if ((int) sh.it_off[x0 + mid] <= q)
            lo = mid;
          else
            hi = mid - 1;
        }
      const uint32_t ent = sh.it_start[x0 + lo];
      const uint32_t m = (ent < multi_base) ? ent : ix.multi[(size_t) (ent - multi_base) * 4 + 1u + (uint32_t) (q - (int) sh.it_off[x0 + lo])];
      okey[strand * PM_SEED_CAP + p] = m + (uint32_t) (PM_DIAG_BIAS - sh.offsets[seg]);
      oseg[strand * PM_SEED_CAP + p] = (uint8_t) seg;
    }
  pm_wave_sync ();
}

template < int SMAX, int PM_LR_BATCH > __global__ __launch_bounds__ (64) void pm_lookup_rep_kernel (PmIndex ix, PmBatch b, PmParams prm, PmLists out, int prio)
{
  pm_set_prio (prio);
  typedef PmLookupRepShared < SMAX > SH;
  __shared__ SH sh;
  const int lane = threadIdx.x;
  const int idepth = ix.idepth;
  unsigned long long n_pos = 0;
  int e_next = (int) gridDim.x + (int) __builtin_amdgcn_readfirstlane ((int) (lane == 0 ? atomicAdd (out.next_end, 1u) : 0u));
  for (int e = blockIdx.x; e < b.n_ends;)
    {
      const int e_cur = e;
      e = e_next;
      if (e < b.n_ends)
        e_next = (int) gridDim.x + (int) __builtin_amdgcn_readfirstlane ((int) (lane == 0 ? atomicAdd (out.next_end, 1u) : 0u));
      pm_lookup_rep_one_end < SMAX, PM_LR_BATCH > (sh, ix, b, prm, out, e_cur, lane, idepth, n_pos);
    }
  if (lane == 0 && n_pos)
    atomicAdd (out.positions, n_pos);
}

// ---- the same look-ups as a three-stage pipeline inside the wave.  pm_lookup_rep_kernel spends its time waiting: per
// read-end the read, the lines, the record headers and the record positions are four dependent trips to HBM, and its
// rate is the number of resident waves over the sum of those latencies.  Here a wave works on three read-ends at once:
//   R (end k+2)  the read's bytes requested into registers
//   P (end k+1)  read -> LDS, N filter, k-mers, the 2 x S x 8 line requests issued as LDS-DMA (global_load_lds_dwordx4:
//                16 bytes per lane straight into LDS, no registers held while in flight)
//   Q (end k)    entries decoded from the lines that landed during the previous Q, record headers, prefix, positions out
// so that a read-end costs one exposed HBM latency (the record headers) instead of four, and two or three waves per CU
// do what six did -- which leaves the CU to the fp64 SW kernel of the other stream.  Lane j handles neighbour j of every
// (strand, segment), one segment per round, and keeps the entries in registers from decode to output.  The order of the
// positions inside a segment's list is immaterial (the reference sorts them, the vote kernel hashes them), so a segment's
// list is laid out as [buckets of one position, in neighbour order][records of 2..99 positions]: the offsets of the first
// kind are ballots and population counts, only the records need a prefix sum.
template < int SMAX > struct __align__ (16) PmLookupRep2Shared
{
  static constexpr int NITEMS = 2 * SMAX * 49;
  static constexpr int NROUNDS = SMAX;                          // 2 x SMAX x 8 lines, 16 per wave instruction
  uint32_t lines[2][NROUNDS * 256];
  uint32_t ment[NITEMS];              // the entries that point to a record, in (segment, neighbour) order
  uint32_t mp[NITEMS + 1];            // exclusive prefix of their counts
  uint16_t mlist[NITEMS];             // segment * 64 + neighbour of each
  uint32_t kmer[2][2 * SMAX];
  int offsets[2][SMAX + 1];
  int seg_ms[2 * SMAX + 2];           // where the positions of a segment's records start in the end's lists, minus mp[] of its first record
  int seg_base[2][SMAX + 1];
  uint8_t seq[2][320];
};

typedef __attribute__ ((address_space (3))) void *pm_lptr_t;
#define PM_SEG_DEAD ((int) 0x80000000)

// pm_rc without branches (five copies of the switch's decision tree per stage were most of the kernel's code): the
// complements of 'A'..'Z' as bytes of four 64-bit constants, everything else 'N'
__device__ __forceinline__ uint8_t pm_rc_flat (uint8_t c)
{
  const unsigned idx = (unsigned) c - (unsigned) 'A';
  const unsigned w = idx >> 3;
  const unsigned long long t = (w == 0) ? 0x4e434e4e4e474e54ull : (w == 1) ? 0x4e4e4e4b4e4d4e4eull : (w == 2) ? 0x4e574e4e4153594eull : 0x4e4e4e4e4e4e4e52ull;
  const uint8_t r = (uint8_t) (t >> (8u * (idx & 7u)));
  return idx < 26u ? r : (uint8_t) 'N';
}

// pm_code without branches
__device__ __forceinline__ unsigned pm_code_flat (uint8_t c, int bis)
{
  const unsigned lc = (unsigned) c | 0x20u;
  unsigned code = (lc == 'c') ? 1u : (lc == 'g') ? 2u : (lc == 't') ? 3u : 0u;
  code = (bis && c == 'C') ? 3u : code;
  return code;
}

// LDS-DMA as an asm statement (cdna_hip_programming.md, inline-asm notes): 16 bytes per lane from each lane's own address to
// lds_byte_addr + 16 * lane.  hipcc does not count it, which is the point: issued through the builtin, every later LDS
// read of the kernel waits for vmcnt(0), i.e. for the NEXT read-end's lines as well.  The kernel waits for it itself.
__device__ __forceinline__ void pm_glds16 (const void *gsrc, uint32_t lds_byte_addr)
{
  unsigned keep;
  asm volatile ("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0":"=&s" (keep):"v" (gsrc), "s" (lds_byte_addr):"memory");
}

// LDS is passed as dynamic shared memory (sizeof (PmLookupRep2Shared < SMAX >)): with a static 34 KB the compiler concludes
// that one wave per SIMD is all the kernel will ever get and spends 256 VGPRs on it, whatever the launch bounds say.
extern __shared__ __align__ (16) uint8_t pm_lookup_rep2_lds[];

template < int SMAX > __global__ __launch_bounds__ (64, SMAX <= 10 ? 3 : 2) void pm_lookup_rep2_kernel (PmIndex ix, PmBatch b, PmParams prm, PmLists out, int prio)
{
  pm_set_prio (prio);
  typedef PmLookupRep2Shared < SMAX > SH;
  SH & sh = *reinterpret_cast < SH * >(pm_lookup_rep2_lds);
  const int lane = threadIdx.x;
  const int idepth = ix.idepth;
  const uint32_t multi_base = ix.multi_base;
  const int n_ends = b.n_ends;
  const int nb = b.stride < 320 ? b.stride : 320;
  unsigned long long n_pos = 0;
  // ends are handed out through a counter; two values are always in flight ahead of their use
  int eQ = n_ends, eP = blockIdx.x, eR;
  int eF = (int) gridDim.x + (int) __builtin_amdgcn_readfirstlane ((int) (lane == 0 ? atomicAdd (out.next_end, 1u) : 0u));
  eR = eF;
  eF = (int) gridDim.x + (int) __builtin_amdgcn_readfirstlane ((int) (lane == 0 ? atomicAdd (out.next_end, 1u) : 0u));
  // stage registers: the bytes of eP
  uint8_t rb[5];
  int rlen = 0;
  int SQ = 0;
  bool skipQ = true;
  uint32_t kQ = 0;
  int buf = 0;
  // lane j looks at neighbour j of every segment (fill_mers' order, pm_neighbour): the 2-bit field it replaces, the
  // alternative's rank, the replica (= 4-bit field) whose line holds it
  const int nb_f = lane > 0 ? (lane - 1) / 3 : 0;
  const uint32_t nb_a = lane > 0 ? (uint32_t) ((lane - 1) % 3) : 0u;
  const uint32_t nb_sh = 2u * (uint32_t) (nb_f & 15);
  const uint32_t nb_keep = lane > 0 ? ~(3u << nb_sh) : 0xFFFFFFFFu;      // lane 0: the k-mer itself
  const uint32_t nb_alt_on = lane > 0 ? 0xFFFFFFFFu : 0u;
  const int nb_p = (nb_f >> 1) & 7;
  const uint32_t nb_p4 = 4u * (uint32_t) nb_p, nb_pw = 16u * (uint32_t) nb_p;
  {
    const uint8_t *src = pm_read_ptr (b, eP, &rlen);
#pragma unroll
    for (int t = 0; t < 5; t++)
      {
        const int i = lane + 64 * t;
        rb[t] = (i < nb) ? src[i] : (uint8_t) 0;
      }
  }
  while (eQ < n_ends || eP < n_ends)
    {
      // the lines of eQ (requested during the previous iteration) and the bytes of eP have landed after this
      asm volatile ("s_waitcnt vmcnt(0)":::"memory");
      int SP = 0;
      bool skipP = true;
      uint32_t kP = 0;          // lane sg: the k-mer of (strand, segment) sg of eP
      if (eP < n_ends)
        {
          // ---- P: the read in registers -> LDS, N filter (pemapper.c:1552-1559), k-mers, line requests into lines[buf ^ 1]
          const int bf = buf ^ 1;
          const int len = rlen;
          int isn = 0;
#pragma unroll
          for (int t = 0; t < 5; t++)
            {
              const int i = lane + 64 * t;
              if (i < len)
                {
                  // the 2-bit codes of both strands (fill_cv_mat / convert_ct, pemapper.c:2375-2383, 2292-2300), not the letters
                  const uint8_t c = rb[t];
                  sh.seq[0][i] = (uint8_t) pm_code_flat (c, prm.bisulfite);
                  sh.seq[1][len - 1 - i] = (uint8_t) pm_code_flat (pm_rc_flat (c), prm.bisulfite);
                  isn += (c == 'N');
                }
            }
          for (int o = 32; o; o >>= 1)
            isn += __shfl_xor (isn, o);
          int cuts = len / idepth;      // pemapper.c:1573-1587
          if (len % idepth == 0)
            cuts--;
          if (cuts > SMAX - 1)
            cuts = SMAX - 1;
          SP = cuts + 1;
          skipP = isn >= 1 + len / 10;
          if (!skipP)
            {
              if (lane <= cuts)
                sh.offsets[bf][lane] = (lane < cuts || cuts == 0) ? lane * idepth : len - idepth;
              pm_wave_sync ();
              if (lane < 2 * SP)
                {
                  const int strand = lane / SP, seg = lane - strand * SP;
                  const uint8_t *p = &sh.seq[strand][sh.offsets[bf][seg]];
                  uint32_t k = 0;
#pragma unroll
                  for (int i = 0; i < 16; i++)
                    k = (k << 2) + p[i];
                  sh.kmer[bf][lane] = k;
                  kP = k;
                }
              pm_wave_sync ();
              const int n_lines = 2 * SP * 8;
              const uint32_t lds0 = (uint32_t) __builtin_amdgcn_readfirstlane ((int) (uint32_t) (size_t) (pm_lptr_t) & sh.lines[bf][0]);
#pragma unroll
              for (int r = 0; r < SMAX; r++)
                {
                  const int li = r * 16 + (lane >> 2);
                  if (li < n_lines)
                    {
                      const int p = li & 7;
                      const uint32_t idx = pm_swap_fields (sh.kmer[bf][li >> 3], p);
                      const uint32_t *g = ix.rep + ((size_t) p << 32) + (size_t) (idx & ~15u) + (size_t) ((lane & 3) * 4);
                      pm_glds16 (g, lds0 + (uint32_t) (r * 1024));
                    }
                }
            }
        }
      // ---- R: request the bytes of the end after (whole row up to the stride: no dependence on its length)
      const int eN = eR;
      if (eN < n_ends)
        {
          eR = eF;
          eF = (int) gridDim.x + (int) __builtin_amdgcn_readfirstlane ((int) (lane == 0 ? atomicAdd (out.next_end, 1u) : 0u));
          const uint8_t *src = pm_read_ptr (b, eN, &rlen);
#pragma unroll
          for (int t = 0; t < 5; t++)
            {
              const int i = lane + 64 * t;
              rb[t] = (i < nb) ? src[i] : (uint8_t) 0;
            }
        }
      // ---- Q
      if (eQ < n_ends)
        {
          const int e = eQ, S = SQ;
          PmEndHeader *hd = &out.hdr[e];
          if (skipQ)
            {
              if (lane == 0)
                hd->kind = PM_KIND_SKIP;
            }
          else
            {
              const uint32_t *lines = &sh.lines[buf][0];
              // ---- C: one (strand, segment) per round, lane j = neighbour j (get_mers, pemapper.c:2158-2165, fill_mers' order).
              //      First every read of the landed lines (nothing between them that could order them), then the bookkeeping.
              uint32_t ent[2 * SMAX];
#pragma unroll
              for (int sg = 0; sg < 2 * SMAX; sg++)
                {
                  const uint32_t k = (uint32_t) __builtin_amdgcn_readlane ((int) kQ, sg);
                  const uint32_t cur = (k >> nb_sh) & 3u;
                  const uint32_t alt = nb_a + (nb_a >= cur ? 1u : 0u);
                  const uint32_t nbk = (k & nb_keep) | ((alt << nb_sh) & nb_alt_on);
                  ent[sg] = lines[sg * 128 + nb_pw + ((nbk >> nb_p4) & 15u)];
                }
              int n_multi = 0;
              int my_ns = 0, my_mfirst = 0;     // lane sg: segment sg's buckets of one position; its first record in mlist
              int my_badi = 0;                  // lane sg: a bucket of the segment reaches too_many_spots (pemapper.c:1602-1606)
#pragma unroll
              for (int sg = 0; sg < 2 * SMAX; sg++)
                {
                  const uint32_t en = (lane < 49 && sg < 2 * S) ? ent[sg] : 0xFFFFFFFFu;
                  ent[sg] = en;
                  const bool is_multi = en >= multi_base && en < 0xFFFFFFFEu;
                  const unsigned long long bs = __ballot (en < multi_base), bm = __ballot (is_multi), bt = __ballot (en == 0xFFFFFFFEu);
                  if (is_multi)
                    {
                      const int slot = n_multi + (int) __builtin_amdgcn_mbcnt_hi ((unsigned) (bm >> 32), __builtin_amdgcn_mbcnt_lo ((unsigned) bm, 0u));
                      sh.mlist[slot] = (uint16_t) (sg * 64 + lane);
                      sh.ment[slot] = en;
                    }
                  my_ns = (lane == sg) ? (int) __popcll (bs) : my_ns;
                  my_mfirst = (lane == sg) ? n_multi : my_mfirst;
                  my_badi = (lane == sg) ? (bt != 0ull ? 1 : 0) : my_badi;
                  n_multi += __popcll (bm);
                }
              const bool my_bad = my_badi != 0;
              pm_wave_sync ();
              // ---- D: the counts of the records (their first word), 4 rounds in flight, and the exclusive prefix of the counts
              {
                uint32_t carry = 0;
#pragma unroll 1
                for (int i0 = 0; i0 < n_multi; i0 += 256)
                  {
                    uint32_t cnt[4];
#pragma unroll
                    for (int r = 0; r < 4; r++)
                      {
                        const int i = i0 + r * 64 + lane;
                        cnt[r] = 0;
                        if (i < n_multi)
                          cnt[r] = ix.multi[(size_t) (sh.ment[i] - multi_base) * 4];
                      }
#pragma unroll
                    for (int r = 0; r < 4; r++)
                      {
                        const int i = i0 + r * 64 + lane;
                        if (i0 + r * 64 < n_multi)
                          {
                            uint32_t incl = cnt[r];
                            for (int o = 1; o < 64; o <<= 1)
                              {
                                const uint32_t t = __shfl_up (incl, o);
                                if (lane >= o)
                                  incl += t;
                              }
                            if (i < n_multi)
                              sh.mp[i] = carry + incl - cnt[r];
                            carry += __shfl (incl, 63);
                          }
                      }
                  }
                if (lane == 0)
                  sh.mp[n_multi] = carry;
              }
              pm_wave_sync ();
              // ---- E: segment sizes, their prefix inside each strand, where each segment's two kinds of positions start
              int mf_next = __shfl_down (my_mfirst, 1);
              if (lane == 2 * S - 1)
                mf_next = n_multi;
              int cnt = 0;
              uint32_t mp_first = 0;
              if (lane < 2 * S)
                {
                  mp_first = sh.mp[my_mfirst];
                  cnt = my_bad ? 0 : my_ns + (int) (sh.mp[mf_next] - mp_first);
                }
              int inc = cnt;
              for (int o = 1; o < 64; o <<= 1)
                {
                  const int t = __shfl_up (inc, o);
                  if (lane >= o)
                    inc += t;
                }
              const int T0 = __shfl (inc, S - 1), TT = __shfl (inc, 2 * S - 1);
              const int T1 = TT - T0;
              int my_dst = PM_SEG_DEAD;         // lane sg: list index of the segment's first position
              uint32_t my_bias = 0;
              if (lane < 2 * S)
                {
                  const int strand = lane / S, seg = lane - strand * S;
                  const int sb = inc - cnt - (strand ? T0 : 0);
                  sh.seg_base[strand][seg] = sb;
                  my_bias = (uint32_t) (PM_DIAG_BIAS - sh.offsets[buf][seg]);
                  if (!my_bad)
                    my_dst = strand * PM_SEED_CAP + sb;
                  sh.seg_ms[lane] = my_bad ? PM_SEG_DEAD : my_dst + my_ns - (int) mp_first;
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
                }
              else
                {
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
                  uint32_t *okey = out.key + (size_t) e * 2 * PM_SEED_CAP;
                  uint8_t *oseg = out.seg + (size_t) e * 2 * PM_SEED_CAP;
                  // ---- F: the buckets of one position, from registers
#pragma unroll
                  for (int sg = 0; sg < 2 * SMAX; sg++)
                    if (sg < 2 * S)
                      {
                        const int dd = __builtin_amdgcn_readlane (my_dst, sg);
                        const uint32_t bias = (uint32_t) __builtin_amdgcn_readlane ((int) my_bias, sg);
                        const uint32_t en = ent[sg];
                        const bool single = en < multi_base;
                        const unsigned long long bs = __ballot (single);
                        if (single && dd != PM_SEG_DEAD)
                          {
                            const uint32_t at = (uint32_t) dd + __builtin_amdgcn_mbcnt_hi ((unsigned) (bs >> 32), __builtin_amdgcn_mbcnt_lo ((unsigned) bs, 0u));
                            *(uint32_t *) ((uint8_t *) okey + (at << 2)) = en + bias;        // (32-bit byte offset: at < 2 * PM_SEED_CAP)
                            oseg[at] = (uint8_t) (sg >= S ? sg - S : sg);
                          }
                      }
                  // ---- G: the records' positions (the first three come with the count)
#pragma unroll 1
                  for (int i0 = 0; i0 < n_multi; i0 += 64)
                    {
                      const int i = i0 + lane;
                      if (i < n_multi)
                        {
                          const int sg = sh.mlist[i] >> 6;
                          const int ms = sh.seg_ms[sg];
                          if (ms != PM_SEG_DEAD)
                            {
                              const uint32_t dst = (uint32_t) ms + sh.mp[i];
                              const uint32_t *rec = ix.multi + (size_t) (sh.ment[i] - multi_base) * 4;
                              const uint4 h = *(const uint4 *) rec;
                              const int seg = sg >= S ? sg - S : sg;
                              const uint32_t bias = (uint32_t) (PM_DIAG_BIAS - sh.offsets[buf][seg]);
                              const uint32_t c = h.x;
                              okey[dst] = h.y + bias;
                              oseg[dst] = (uint8_t) seg;
                              okey[dst + 1] = h.z + bias;
                              oseg[dst + 1] = (uint8_t) seg;
                              if (c > 2)
                                {
                                  okey[dst + 2] = h.w + bias;
                                  oseg[dst + 2] = (uint8_t) seg;
                                }
#pragma unroll 1
                              for (uint32_t t = 3; t < c; t++)
                                {
                                  okey[dst + t] = rec[t + 1] + bias;
                                  oseg[dst + t] = (uint8_t) seg;
                                }
                            }
                        }
                    }
                }
              pm_wave_sync ();
            }
        }
      eQ = eP;
      eP = eN;
      kQ = kP;
      SQ = SP;
      skipQ = skipP;
      buf ^= 1;
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
      for (int strand = 0; strand < 2 && go_on && PM_PROBE (probe) != 1; strand++)
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

// ---- find_matches with ONE WAVE per read-end (no workgroup barriers), persistent.  Same method as pm_vote_strand
// (pemap_seed.hip.h): counting sort of the strand's positions into buckets of hash(diagonal / 16), exact tot_found of
// every anchor from the three buckets around its bin, the anchors that reach the running best ranked into the walk's order,
// and the reference's walk replayed on them.  Differences: anchors are visited in bucket order (their order is immaterial),
// so the unsorted list never has to sit in LDS -- it is read twice from HBM (count, scatter); 512 buckets; at most
// PM_VW_SURV ranked anchors, an end with more is appended to the big-end list and left to pm_seed_kernel.
#define PM_VW_NB 512
#define PM_VW_SURV 512

__device__ __forceinline__ unsigned pm_bin_hash9 (uint32_t bin)
{
  return (bin * 2654435761u) >> 23;     // 9 bits
}

struct __align__ (16) PmVoteWaveShared
{
  uint32_t table[PM_VW_NB];
  uint32_t bkey[PM_SEED_CAP];           // the strand's diagonal keys, bucketed
  uint32_t hits[PM_MAX_HITS];
  uint16_t surv[PM_VW_SURV];            // anchors that reach the running best (indices into bkey), arrival order
  uint16_t order[PM_VW_SURV];           // the same in walk order (indices into surv)
  uint16_t hits_off[PM_MAX_HITS];
  uint8_t bseg[PM_SEED_CAP];
  uint8_t tfs[PM_VW_SURV];
  uint8_t hits_or[PM_MAX_HITS];
  unsigned n_surv;
};

// the first PM_VW_PRE x 64 positions of both strands' lists and the header of a read-end, loaded into registers one end
// ahead (unconditionally: the lists are allocated at full capacity, what lies beyond T is ignored)
#define PM_VW_PRE 8
struct PmVotePre
{
  uint32_t k[2][PM_VW_PRE];
  uint32_t s[2][PM_VW_PRE];
  uint32_t hw;                  // dword `lane` of the end's PmEndHeader
};

__device__ __forceinline__ void pm_vote_prefetch (PmVotePre & r, const PmLists & in, int e, int n_ends, int lane)
{
  static_assert (sizeof (PmEndHeader) == 4 * 22, "header decode below assumes 22 dwords");
  if (e >= n_ends)
    return;
  r.hw = (lane < 22) ? ((const uint32_t *) &in.hdr[e])[lane] : 0u;
#pragma unroll
  for (int st = 0; st < 2; st++)
#pragma unroll
    for (int i = 0; i < PM_VW_PRE; i++)
      {
        const size_t o = ((size_t) e * 2 + st) * PM_SEED_CAP + i * 64 + lane;
        r.k[st][i] = in.key[o];
        r.s[st][i] = in.seg[o];
      }
}

// PERSIST: the wave strides over the ends and loads the next end's lists while it votes; otherwise one end per wave (the
// grid covers the ends) and no second register set
template < int SMAX, bool PERSIST > __global__ __launch_bounds__ (64) void pm_vote_wave_kernel (PmIndex ix, PmBatch b, PmParams prm, PmHits h, PmLists in, int prio)
{
  pm_set_prio (prio);
  __shared__ PmVoteWaveShared sh;
  const int lane = threadIdx.x;
  const int idepth = ix.idepth;
  const int max_off = max (2, idepth - 4);
  const uint32_t span = (uint32_t) (2 * (max_off - 1));
  PmVotePre nxt;
  pm_vote_prefetch (nxt, in, blockIdx.x, b.n_ends, lane);
  for (int e = blockIdx.x; e < b.n_ends; e += PERSIST ? (int) gridDim.x : b.n_ends)
    {
      const PmVotePre cur = nxt;
      if (PERSIST)
        pm_vote_prefetch (nxt, in, e + gridDim.x, b.n_ends, lane);
      const int kind = (int) (__shfl (cur.hw, 21) & 0xFFu);
      if (kind == PM_KIND_BIG)
        continue;               // left to pm_seed_kernel in list mode
      int len;
      (void) pm_read_ptr (b, e, &len);
      int total_cuts = len / idepth;
      if (len % idepth == 0)
        total_cuts--;
      if (total_cuts > SMAX - 1)
        total_cuts = SMAX - 1;
      const int S = total_cuts + 1;
      const int last_off = len - idepth;
      int tot = 0;
      bool overflow = false;
      if (kind == PM_KIND_NORMAL)
        {
          int min_match = max (1, total_cuts);       // pemapper.c:1642-1645
          if (total_cuts > 4)
            min_match = (4 * total_cuts) / 5;
          min_match = min (min_match, 4);
          bool go_on = true;
          for (int strand = 0; strand < 2 && go_on && !overflow; strand++)
            {
              const int T = (int) ((__shfl (cur.hw, 0) >> (16 * strand)) & 0xFFFFu);
              // pemapper.c:2200-2207: nothing is searched (and earlier hits are dropped) when every segment holds more than max_hits positions
              int cnt = 10000;
              {
                // seg_base[strand][lane] is u16 number 2 + strand * (PM_MAX_SEG + 1) + lane of the header
                const int i0 = 2 + strand * (PM_MAX_SEG + 1) + (lane < S ? lane : 0), i1 = i0 + 1;
                const int b0 = (int) ((__shfl (cur.hw, i0 >> 1) >> (16 * (i0 & 1))) & 0xFFFFu);
                const int b1 = (int) ((__shfl (cur.hw, i1 >> 1) >> (16 * (i1 & 1))) & 0xFFFFu);
                if (lane < S)
                  cnt = b1 - b0;
              }
              for (int o = 32; o; o >>= 1)
                cnt = min (cnt, __shfl_xor (cnt, o));
              if (cnt > PM_MAX_HITS)
                {
                  tot = 0;
                  continue;
                }
              const uint32_t *ikey = in.key + ((size_t) e * 2 + strand) * PM_SEED_CAP;
              const uint8_t *iseg = in.seg + ((size_t) e * 2 + strand) * PM_SEED_CAP;
              // ---- counting sort by bucket: clear, count, exclusive scan, scatter
              constexpr int PER = PM_VW_NB / 64;
#pragma unroll
              for (int i = 0; i < PER; i++)
                sh.table[lane + i * 64] = 0;
              if (lane == 0)
                sh.n_surv = 0;
              pm_wave_sync ();
#pragma unroll
              for (int i = 0; i < PM_VW_PRE; i++)
                if (i * 64 + lane < T)
                  atomicAdd (&sh.table[pm_bin_hash9 ((strand ? cur.k[1][i] : cur.k[0][i]) >> 4)], 1u);
              for (int p = PM_VW_PRE * 64 + lane; p < T; p += 64)
                atomicAdd (&sh.table[pm_bin_hash9 (ikey[p] >> 4)], 1u);
              pm_wave_sync ();
              {
                uint32_t v[PER], run = 0;
#pragma unroll
                for (int i = 0; i < PER; i++)
                  {
                    v[i] = run;
                    run += sh.table[lane * PER + i];
                  }
                uint32_t inc = run;
                for (int o = 1; o < 64; o <<= 1)
                  {
                    const uint32_t t = __shfl_up (inc, o);
                    if (lane >= o)
                      inc += t;
                  }
                const uint32_t base = inc - run;
                pm_wave_sync ();
#pragma unroll
                for (int i = 0; i < PER; i++)
                  sh.table[lane * PER + i] = base + v[i];
              }
              pm_wave_sync ();
#pragma unroll
              for (int i = 0; i < PM_VW_PRE; i++)
                if (i * 64 + lane < T)
                  {
                    const uint32_t k = strand ? cur.k[1][i] : cur.k[0][i];
                    const uint32_t pos = atomicAdd (&sh.table[pm_bin_hash9 (k >> 4)], 1u);
                    sh.bkey[pos] = k;
                    sh.bseg[pos] = (uint8_t) (strand ? cur.s[1][i] : cur.s[0][i]);
                  }
              for (int p = PM_VW_PRE * 64 + lane; p < T; p += 64)
                {
                  const uint32_t k = ikey[p];
                  const uint32_t pos = atomicAdd (&sh.table[pm_bin_hash9 (k >> 4)], 1u);
                  sh.bkey[pos] = k;
                  sh.bseg[pos] = iseg[p];
                }
              pm_wave_sync ();  // table[h] is now the END of bucket h
              // ---- tot_found of every anchor the walk can reach (pemapper.c:2216, 2241-2249)
              const int loop_max = 1 + total_cuts - min_match;
              for (int x = lane; x < T; x += 64)
                {
                  const int sa = sh.bseg[x];
                  if (sa > loop_max)
                    continue;
                  const uint32_t ka = sh.bkey[x];
                  const uint32_t bin = ka >> 4;
                  const uint32_t later = (total_cuts >= 31 ? 0xFFFFFFFFu : ((1u << (total_cuts + 1)) - 1u)) & ~((2u << sa) - 1u);
                  uint32_t bits = 0;
                  for (int db = -1; db <= 1 && bits != later; db++)
                    {
                      const unsigned hh = pm_bin_hash9 (bin + (uint32_t) db);
                      const uint32_t lo = hh ? sh.table[hh - 1] : 0u, hi = sh.table[hh];
                      for (uint32_t y = lo; y < hi; y++)
                        {
                          // |diag_y - diag_a| < max_off, in wrapping 32-bit arithmetic (keys stay below 2^32 - 100)
                          const uint32_t dd = sh.bkey[y] - ka + (uint32_t) (max_off - 1);
                          const int sy = sh.bseg[y];
                          if (dd <= span && sy > sa)
                            bits |= 1u << sy;
                        }
                    }
                  const int tf = 1 + __popc (bits);
                  if (tf >= min_match)
                    {
                      const unsigned slot = atomicAdd (&sh.n_surv, 1u);
                      if (slot < PM_VW_SURV)
                        {
                          sh.surv[slot] = (uint16_t) x;
                          sh.tfs[slot] = (uint8_t) tf;
                        }
                    }
                }
              pm_wave_sync ();
              const int ns = (int) sh.n_surv;
              if (ns > PM_VW_SURV)
                {
                  overflow = true;
                  break;
                }
              // ---- walk order: segment ascending, position ascending inside a segment (same offset, so diagonal ascending)
              for (int sv = lane; sv < ns; sv += 64)
                {
                  const int a = (int) sh.surv[sv];
                  const uint64_t ck = ((uint64_t) sh.bseg[a] << 32) | sh.bkey[a];
                  int rank = 0;
                  for (int y = 0; y < ns; y++)
                    {
                      const int bq = (int) sh.surv[y];
                      rank += ((((uint64_t) sh.bseg[bq] << 32) | sh.bkey[bq]) < ck);
                    }
                  sh.order[rank] = (uint16_t) sv;
                }
              pm_wave_sync ();
              // ---- the walk's state machine on the ranked anchors (pemapper.c:2251-2284)
              bool more = true, done = false;
              int cur_loop = -1;
              for (int i0 = 0; i0 < ns && !done; i0 += 64)
                {
                  const int i = i0 + lane;
                  const bool act = i < ns;
                  const int sv = act ? (int) sh.order[i] : 0;
                  const int tf = act ? (int) sh.tfs[sv] : 0;
                  const int a = act ? (int) sh.surv[sv] : 0;
                  const int my_loop = act ? (int) sh.bseg[a] : 0;
                  const int my_off = (my_loop < total_cuts || total_cuts == 0) ? my_loop * idepth : last_off;
                  const uint32_t my_ml = (act ? sh.bkey[a] : 0u) - (uint32_t) (PM_DIAG_BIAS - my_off);        // the position itself
                  unsigned long long cand = __ballot (act && tf >= min_match);
                  while (cand)
                    {
                      const int l = __ffsll ((long long) cand) - 1;
                      cand &= cand - 1;
                      const int tfl = __shfl (tf, l);
                      const int loop = __shfl (my_loop, l);
                      if (loop != cur_loop)
                        {
                          // the walk's loop bound is tested when a segment is entered, not inside it (pemapper.c:2216)
                          if (loop > 1 + total_cuts - min_match)
                            {
                              done = true;
                              break;
                            }
                          cur_loop = loop;
                        }
                      const int off_a = __shfl (my_off, l);
                      const uint32_t ml = __shfl (my_ml, l);
                      if (tfl > min_match)
                        {
                          min_match = tfl;
                          if (lane == 0)
                            {
                              sh.hits[0] = ml;
                              sh.hits_off[0] = (uint16_t) off_a;
                              sh.hits_or[0] = (uint8_t) strand;
                            }
                          tot = 1;
                          pm_wave_sync ();
                          cand &= __ballot (tf >= min_match);     // candidates below the new best would fall through both tests
                        }
                      else if (tfl == min_match)
                        {
                          if (tot < PM_MAX_HITS)
                            {
                              const uint32_t diag = ml - (uint32_t) off_a;        // unsigned, pemapper.c:2268
                              bool dup = false;
                              for (int k = lane; k < tot; k += 64)
                                if (sh.hits[k] - (uint32_t) sh.hits_off[k] == diag)
                                  dup = true;
                              if (!__any (dup))
                                {
                                  if (lane == 0)
                                    {
                                      sh.hits[tot] = ml;
                                      sh.hits_off[tot] = (uint16_t) off_a;
                                      sh.hits_or[tot] = (uint8_t) strand;
                                    }
                                  tot++;
                                  pm_wave_sync ();
                                }
                            }
                          else
                            {
                              more = false; // the reference returns with a full list (pemapper.c:2283-2284)
                              done = true;
                              break;
                            }
                        }
                    }
                }
              if (tot >= PM_MAX_HITS)
                more = false;
              go_on = more;
              pm_wave_sync ();
            }
        }
      if (overflow)
        {
          if (lane == 0)
            in.big_list[atomicAdd (in.n_big, 1u)] = (uint32_t) e;
          continue;
        }
      // ---- raw hits out (pm_seed_emit's format); pm_emit_kernel makes windows and task lists of them
      if (lane == 0)
        h.n_hits[e] = tot;
      for (int t = lane; t < tot; t += 64)
        {
          const size_t o = (size_t) e * PM_MAX_HITS + t;
          h.spot[o] = sh.hits[t];
          h.nn[o] = (int16_t) sh.hits_off[t];
          h.orient[o] = sh.hits_or[t];
        }
      pm_wave_sync ();
    }
}
