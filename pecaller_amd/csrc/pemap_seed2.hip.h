// pemap_seed2.hip.h -- the seed stage for the reference's table layout, split in two kernels (the form that serves when the
// look-up replicas are not in use: a device without the room for them, PEMAP_REPLICAS=0, the two-rank rehearsal on one GPU):
//
//   pm_lookup_wave_kernel   one WAVE per read-end, persistent: look-ups in pos_index, slice gather from .mdx -> per-end lists of
//                           (diagonal key, segment) in HBM.
//   pm_vote_wave_kernel     one WAVE per read-end: lists -> find_matches -> raw hits (pm_emit_kernel makes windows and SW tasks).
//
// Read-ends whose strand holds more than PM_SEED_CAP positions (repeats) are appended to a list and handled afterwards
// by the monolithic pm_seed_kernel (list mode), which has the global spill path.  With the replicas the fused kernel of
// pemap_seed3.hip.h does both halves in one wave.  (Round 1's and round 2's other forms -- a workgroup per read-end for either
// half, the look-ups against the replicas as a kernel of their own, plain and software-pipelined -- were measured slower, had no
// test that selected them and are gone; DESIGN.md section 5 keeps their numbers.)
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

// ---- the look-ups with ONE WAVE per read-end, persistent.  Random 64-byte lines are a DRAM-side limit (about 47 G/s on
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

// the wave strides over the ends and loads the next end's lists while it votes
template < int SMAX > __global__ __launch_bounds__ (64) void pm_vote_wave_kernel (PmIndex ix, PmBatch b, PmParams prm, PmHits h, PmLists in, int prio)
{
  pm_set_prio (prio);
  __shared__ PmVoteWaveShared sh;
  const int lane = threadIdx.x;
  const int idepth = ix.idepth;
  const int max_off = max (2, idepth - 4);
  const uint32_t span = (uint32_t) (2 * (max_off - 1));
  PmVotePre nxt;
  pm_vote_prefetch (nxt, in, blockIdx.x, b.n_ends, lane);
  for (int e = blockIdx.x; e < b.n_ends; e += (int) gridDim.x)
    {
      const PmVotePre cur = nxt;
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
