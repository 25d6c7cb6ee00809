// pemap_seed.hip.h -- K1/K2: seed gather and diagonal vote (initial_map / find_matches) on gfx950.
// Included from pemap_kernels.hip.h.
#pragma once

// One 256-thread workgroup per read-end, persistent over ends.  Template SMAX bounds the number of 16-base segments
// and sizes the LDS arrays.
//
// Phases per read-end:
//   1. both strands' 2 x S x 49 bucket look-ups in pos_index, all in flight at once (the HBM-random phase);
//   2. per segment: emptied if any bucket >= too_many_spots, else offsets of its 49 slices;
//   3. both strands' bucket slices copied from .mdx, one position per thread and round; every position is stored as its
//      DIAGONAL key  m + 300 - offset(segment)  (read start implied by that seed) together with its segment number;
//   4. per strand, find_matches (pemapper.c:2189-2289) WITHOUT sorting the lists: the reference counts, per anchor, the
//      later segments that hold a position within max_off of the anchor's diagonal (tot_found) and only acts on anchors
//      whose count reaches the running best.  The positions are counting-sorted into 2048 buckets by a hash of
//      diagonal/16, every anchor gets its exact count from the three buckets around its bin, the few anchors that reach
//      the running best are put in the walk's order (segment, position) and ONE wave replays the walk's state machine on
//      them -- reset on '>', append on '==' if the diagonal is new, stop at max_hits tied hits.
#define PM_SEED_THREADS 256
typedef uint32_t pm_u32x2 __attribute__ ((ext_vector_type (2), aligned (4)));
#define PM_SEED_TABLE 2048
#define PM_DIAG_BIAS 300
// timing probes (kernels cut short after a phase: results are wrong) exist only in builds with -DPEMAP_TIMING_PROBES; the
// product build compiles them out, whatever a caller passes
#ifdef PEMAP_TIMING_PROBES
#define PM_PROBE(x) (x)
#else
#define PM_PROBE(x) 0
#endif

template < int SMAX > struct __align__ (8) PmSeedShared
{
  static constexpr int NITEMS = 2 * SMAX * 49;            // both strands
  struct Items
  {
    uint32_t it_start[NITEMS];          // pos_index[k-mer]
    uint16_t it_len[NITEMS];            // 0xFFFF = bucket >= too_many_spots
    uint16_t it_off[NITEMS];            // exclusive prefix of the lengths inside the segment
  };
  union
  {
    Items it;                           // phases 1-3
    uint32_t table[PM_SEED_TABLE];      // phase 4: segment masks per diagonal bin
  } u;
  uint32_t ekey[2][PM_SEED_CAP];        // diagonal keys of the gathered positions, per strand
  uint32_t bkey[PM_SEED_CAP];           // the strand being voted on, bucketed by hashed diagonal bin
  uint32_t hits[PM_MAX_HITS];
  uint32_t wsum[4];
  uint32_t kmer[2][2 * SMAX];           // [pipeline buffer][strand * S + segment]
  int seg_cnt[2 * SMAX];
  int seg_base[2][SMAX + 1];
  int offsets[2][SMAX + 1];             // [pipeline buffer][segment]
  int skip[2];                          // N filter verdict of the buffered read
  int ncount[2];
  int state[4];                         // min_match, tot_hits, go_on, task base
  unsigned n_surv;
  uint16_t surv[PM_SEED_CAP];           // surviving anchors (indices into ekey), then the same in walk order
  uint16_t order[PM_SEED_CAP];
  uint16_t hits_off[PM_MAX_HITS];
  uint8_t eseg[2][PM_SEED_CAP];
  uint8_t bseg[PM_SEED_CAP];
  uint8_t tfs[PM_SEED_CAP];             // tot_found of the candidates
  uint8_t hits_or[PM_MAX_HITS];
  uint8_t seq[2][2][320];               // [pipeline buffer][strand]
};

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains vmcnt, which would make the look-ups
// that were issued for the workgroup's NEXT read-end (software pipeline) complete at the first barrier of the vote.
__device__ __forceinline__ void pm_lds_barrier ()
{
  asm volatile ("s_waitcnt lgkmcnt(0)\n\ts_barrier":::"memory");
}

template < bool LDS_ONLY > __device__ __forceinline__ void pm_barrier ()
{
  if (LDS_ONLY)
    pm_lds_barrier ();
  else
    __syncthreads ();
}

__device__ __forceinline__ unsigned pm_bin_hash (uint32_t bin)
{
  return (bin * 2654435761u) >> 21;     // 11 bits
}

// find_matches for one strand on unsorted diagonal keys; see the header comment.  All 256 threads enter; returns through
// sh.state.  IdxT = uint16_t (LDS arrays) or uint32_t (global spill arrays of a strand with more than PM_SEED_CAP positions).
//
// The positions are bucketed by a hash of diagonal/16 (counting sort through the LDS table: count, exclusive scan, scatter);
// an anchor's tot_found (pemapper.c:2241-2249) then only has to look at the buckets of its own and the two neighbouring
// bins -- a handful of entries -- instead of at every position of the strand.  Buckets may mix bins (hash collisions) and
// may be visited twice; the test on the real diagonals makes that harmless.
template < class SH, class IdxT >
__device__ void pm_vote_strand (SH & sh, const uint32_t * ekey, const uint8_t * eseg, uint32_t * bkey, uint8_t * bseg, IdxT * surv, IdxT * order,
                                uint8_t * tfs, int T, const int *seg_cnt, const int *offsets, int total_cuts, int max_off, int &min_match,
                                int &tot, bool & go_on, uint8_t strand, int probe = 0)
{
  constexpr bool LDSP = sizeof (IdxT) == 2;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  // pemapper.c:2200-2207: nothing is searched (and earlier hits are dropped) when every segment holds more than max_hits positions
  unsigned min_spots = 10000;
  for (int s = 0; s <= total_cuts; s++)
    min_spots = min (min_spots, (unsigned) seg_cnt[s]);
  if (min_spots > PM_MAX_HITS)
    {
      tot = 0;
      return;
    }
  constexpr int PER = PM_SEED_TABLE / PM_SEED_THREADS;   // table entries per thread
  for (int i = 0; i < PER; i++)
    sh.u.table[tid * PER + i] = 0;
  if (tid == 0)
    sh.n_surv = 0;
  pm_barrier < LDSP > ();
  // ---- counting sort by bucket: count
  for (int p = tid; p < T; p += PM_SEED_THREADS)
    atomicAdd (&sh.u.table[pm_bin_hash (ekey[p] >> 4)], 1u);
  pm_barrier < LDSP > ();
  // ---- exclusive scan of the 2048 counts
  {
    uint32_t v[PER], run = 0;
    for (int i = 0; i < PER; i++)
      {
        v[i] = run;
        run += sh.u.table[tid * PER + i];
      }
    uint32_t inc = run;
    for (int o = 1; o < 64; o <<= 1)
      {
        const uint32_t t = __shfl_up (inc, o);
        if (lane >= o)
          inc += t;
      }
    if (lane == 63)
      sh.wsum[wv] = inc;
    pm_barrier < LDSP > ();
    uint32_t base = inc - run;
    for (int w = 0; w < wv; w++)
      base += sh.wsum[w];
    for (int i = 0; i < PER; i++)
      sh.u.table[tid * PER + i] = base + v[i];
  }
  pm_barrier < LDSP > ();
  // ---- scatter; afterwards table[h] is the END of bucket h (= start of bucket h + 1)
  for (int p = tid; p < T; p += PM_SEED_THREADS)
    {
      const uint32_t k = ekey[p];
      const uint32_t pos = atomicAdd (&sh.u.table[pm_bin_hash (k >> 4)], 1u);
      bkey[pos] = k;
      bseg[pos] = eseg[p];
    }
  pm_barrier < LDSP > ();
  if (PM_PROBE (probe) == 2)
    return;
  // ---- tot_found of every anchor the walk can reach: 1 + number of LATER segments holding a position whose diagonal
  //      differs by less than max_off.  An anchor of segment `loop` is only visited while loop <= 1 + max_depth - min_match
  //      (pemapper.c:2216; the bound only shrinks), and only acts if its count reaches min_match (which only grows).
  const int loop_max = 1 + total_cuts - min_match;
  const uint32_t span = (uint32_t) (2 * (max_off - 1));
  for (int p = tid; p < T; p += PM_SEED_THREADS)
    {
      const int sa = eseg[p];
      if (sa > loop_max)
        continue;
      const uint32_t ka = ekey[p];
      const uint32_t bin = ka >> 4;
      const uint32_t later = (total_cuts >= 31 ? 0xFFFFFFFFu : ((1u << (total_cuts + 1)) - 1u)) & ~((2u << sa) - 1u);
      uint32_t bits = 0;
      for (int db = -1; db <= 1 && bits != later; db++)
        {
          const unsigned hh = pm_bin_hash (bin + (uint32_t) db);
          const uint32_t lo = hh ? sh.u.table[hh - 1] : 0u, hi = sh.u.table[hh];
          for (uint32_t x = lo; x < hi; x++)
            {
              // |diag_x - diag_a| < max_off, in wrapping 32-bit arithmetic (keys stay below 2^32 - 100)
              const uint32_t d = bkey[x] - ka + (uint32_t) (max_off - 1);
              const int sx = bseg[x];
              if (d <= span && sx > sa)
                bits |= 1u << sx;
            }
        }
      const int tf = 1 + __popc (bits);
      if (tf >= min_match)
        {
          const unsigned slot = atomicAdd (&sh.n_surv, 1u);
          surv[slot] = (IdxT) p;
          tfs[slot] = (uint8_t) tf;
        }
    }
  pm_barrier < LDSP > ();
  if (PM_PROBE (probe) == 3)
    return;
  const int ns = (int) sh.n_surv;
  // walk order: segment ascending, position ascending inside a segment (same offset, so diagonal ascending)
  for (int sv = tid; sv < ns; sv += PM_SEED_THREADS)
    {
      const int a = (int) surv[sv];
      const uint64_t ck = ((uint64_t) eseg[a] << 32) | ekey[a];
      int rank = 0;
      for (int y = 0; y < ns; y++)
        {
          const int bq = (int) surv[y];
          rank += ((((uint64_t) eseg[bq] << 32) | ekey[bq]) < ck);
        }
      order[rank] = (IdxT) sv;
    }
  pm_barrier < LDSP > ();
  if (PM_PROBE (probe) == 4)
    return;
  if (tid < 64)
    {
      bool more = true, done = false;
      int cur_loop = -1;
      for (int i0 = 0; i0 < ns && !done; i0 += 64)
        {
          const int i = i0 + lane;
          const bool act = i < ns;
          const int sv = act ? (int) order[i] : 0;
          const int tf = act ? (int) tfs[sv] : 0;
          const int a = act ? (int) surv[sv] : 0;
          // every lane fetches its own candidate's segment, offset and position up front; the serial part below only shuffles
          const int my_loop = act ? (int) eseg[a] : 0;
          const int my_off = offsets[my_loop];
          const uint32_t my_ml = (act ? ekey[a] : 0u) - (uint32_t) (PM_DIAG_BIAS - my_off);        // the position itself
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
                      sh.hits_or[0] = strand;
                    }
                  tot = 1;
                  __builtin_amdgcn_fence (__ATOMIC_SEQ_CST, "wavefront");
                  __builtin_amdgcn_wave_barrier ();
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
                              sh.hits_or[tot] = strand;
                            }
                          tot++;
                          __builtin_amdgcn_fence (__ATOMIC_SEQ_CST, "wavefront");
                          __builtin_amdgcn_wave_barrier ();
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
      if (lane == 0)
        {
          sh.state[0] = min_match;
          sh.state[1] = tot;
          sh.state[2] = more ? 1 : 0;
        }
    }
  pm_barrier < LDSP > ();
  min_match = sh.state[0];
  tot = sh.state[1];
  go_on = sh.state[2] != 0;
}

// The vote's result for one read-end leaves the workgroup raw: the hit count and, per hit, the matched position, the
// segment offset it was found with and the strand.  pm_emit_kernel turns them into SW windows and task lists afterwards
// with one thread per end, so that no workgroup waits on contig look-ups or on the shared task counters.
template < class SH > __device__ __forceinline__ void pm_seed_emit (SH & sh, const PmHits & h, int e, int tot)
{
  const int tid = threadIdx.x;
  if (tid == 0)
    h.n_hits[e] = tot;
  for (int t = tid; t < tot; t += PM_SEED_THREADS)
    {
      const size_t o = (size_t) e * PM_MAX_HITS + t;
      h.spot[o] = sh.hits[t];
      h.nn[o] = (int16_t) sh.hits_off[t];
      h.orient[o] = sh.hits_or[t];
    }
}

// hit -> spot and SW window (pemapper.c:1664-1669, 1047-1081), in place
__device__ __forceinline__ void pm_emit_hit (const PmIndex & ix, const PmHits & h, size_t o, int len)
{
  long temp = (long) h.spot[o] - (long) h.nn[o];
  uint32_t spot = (uint32_t) (temp > 0 ? temp : 0);
  int chrom = pm_find_chrom (ix.contig_starts, ix.n_contigs, spot);
  unsigned extra = 15u * (unsigned) chrom;
  long tt = (long) extra + (long) spot - (long) PM_SLOP;
  if (tt < 0)
    tt = 0;
  unsigned cs0 = ix.contig_starts[chrom] + extra;
  unsigned start_match = ((long) cs0 > tt) ? cs0 : (unsigned) tt;
  unsigned e1 = ix.contig_starts[chrom + 1] + extra;
  unsigned e2w = extra + spot + (unsigned) len + PM_SLOP;
  unsigned end_match = e1 < e2w ? e1 : e2w;
  int blen = (int) (1u + end_match - start_match);
  h.spot[o] = spot;
  h.gpos[o] = start_match;
  h.nn[o] = (int16_t) blen;
}

// One thread per read-end.  An end with one hit is scored once, with direction nibbles, into its own slab; ends with
// several hits are scored without, and only the winner is scored again (pm_select_kernel).  Slab numbers and task-list
// ranges are handed out with one atomic per wave.
// (one-wave workgroups, like every kernel that runs beside the seed kernel: a wave starts wherever a SIMD has room, a workgroup of
// four waited until a CU had room for all four)
__global__ __launch_bounds__ (64) void pm_emit_kernel (PmIndex ix, PmBatch b, PmHits h, uint32_t * tasks_s, uint32_t * tasks_m, PmCounters * ctr)
{
  const int e = blockIdx.x * 64 + threadIdx.x;
  const int lane = threadIdx.x & 63;
  const int tot = e < b.n_ends ? h.n_hits[e] : 0;
  const unsigned long long below = (1ull << lane) - 1ull;
  // ---- single hits
  const unsigned long long ms = __ballot (tot == 1);
  unsigned slot0 = 0, task0 = 0;
  if (ms)
    {
      if (lane == 0)
        {
          slot0 = atomicAdd (&ctr->n_slots, (unsigned) __popcll (ms));
          task0 = atomicAdd (&ctr->n_tasks_s, (unsigned) __popcll (ms));
        }
      slot0 = __shfl (slot0, 0);
      task0 = __shfl (task0, 0);
    }
  // ---- several hits: exclusive prefix of the counts inside the wave
  const unsigned mine = tot > 1 ? (unsigned) tot : 0u;
  unsigned inc = mine;
  for (int o = 1; o < 64; o <<= 1)
    {
      const unsigned t = __shfl_up (inc, o);
      if (lane >= o)
        inc += t;
    }
  const unsigned wave_total = __shfl (inc, 63);
  unsigned mbase = 0;
  if (wave_total)
    {
      if (lane == 0)
        mbase = atomicAdd (&ctr->n_tasks_m, wave_total);
      mbase = __shfl (mbase, 0);
    }
  if (e >= b.n_ends)
    return;
  int len;
  (void) pm_read_ptr (b, e, &len);
  const size_t o0 = (size_t) e * PM_MAX_HITS;
  if (tot == 1)
    {
      const unsigned r = (unsigned) __popcll (ms & below);
      h.slot[e] = (int) (slot0 + r);
      pm_emit_hit (ix, h, o0, len);
      tasks_s[task0 + r] = (uint32_t) o0;
    }
  else
    {
      h.slot[e] = -1;
      const unsigned tb = mbase + inc - mine;
      for (int t = 0; t < tot; t++)
        {
          pm_emit_hit (ix, h, o0 + t, len);
          tasks_m[tb + t] = (uint32_t) (o0 + t);
        }
    }
}

// A look-up item (bucket) of the monolithic kernel: it_start / it_len, and the q-th position of its slice.  On the reference's
// layout it_start indexes `mers`; through the replicas a bucket of one position carries it in it_start (it_len 1: records hold
// 2..99), a longer one the word of its record's first position in `multi` (the record's count is read here: one dependent load).
__device__ __forceinline__ void pm_item_of_entry (const PmIndex & ix, uint32_t ent, uint32_t & it_start, uint16_t & it_len)
{
  if (ent == 0xFFFFFFFFu)
    {
      it_start = 0;
      it_len = 0;
    }
  else if (ent == 0xFFFFFFFEu)
    {
      it_start = 0;
      it_len = 0xFFFF;
    }
  else if (ent < ix.multi_base)
    {
      it_start = ent;
      it_len = 1;
    }
  else
    {
      const uint32_t w = (ent - ix.multi_base) * 4u;
      it_start = w + 1u;
      it_len = (uint16_t) ix.multi[w];
    }
}

__device__ __forceinline__ uint32_t pm_item_pos (const PmIndex & ix, uint32_t it_start, uint16_t it_len, uint32_t q)
{
  if (ix.n_rep != 8)
    return ix.mers[it_start + q];
  return it_len == 1 ? it_start : ix.multi[it_start + q];
}

// Stage A of a read-end (software-pipelined one end ahead of the vote): the read and its reverse complement into LDS
// buffer `buf`, the N filter, the segment offsets and 16-mers, and the 2 x S x 49 look-ups ISSUED into registers.
// They complete while the workgroup votes on the previous end.
template < int SMAX, int NI, class SH >
__device__ __forceinline__ void pm_seed_stage_a (SH & sh, const PmIndex & ix, const PmBatch & b, int bis, int e, int buf, uint32_t (&v0)[NI],
                                                 uint32_t (&v1)[NI])
{
  const int tid = threadIdx.x;
  const int idepth = ix.idepth;
  int len;
  const uint8_t *src = pm_read_ptr (b, e, &len);
  // ---- read + reverse complement; N filter (pemapper.c:1552-1559: upper-case 'N' only)
  int isn = 0;
  for (int i = tid; i < len; i += PM_SEED_THREADS)
    {
      uint8_t c = src[i];
      sh.seq[buf][0][i] = c;
      sh.seq[buf][1][len - 1 - i] = pm_rc (c);
      isn += (c == 'N');
    }
  // ---- segment offsets (pemapper.c:1573-1587)
  int total_cuts = len / idepth;
  if (len % idepth == 0)
    total_cuts--;
  if (total_cuts > SMAX - 1)
    total_cuts = SMAX - 1;      // cannot happen: the host picks SMAX from the longest staged read
  const int S = total_cuts + 1;
  if (tid <= total_cuts)
    sh.offsets[buf][tid] = (tid < total_cuts || total_cuts == 0) ? tid * idepth : len - idepth;
  if (isn)
    atomicAdd (&sh.ncount[buf], isn);
  pm_lds_barrier ();
  const int n_count = sh.ncount[buf];
  const bool skip = n_count >= 1 + len / 10;
  if (tid == 0)
    sh.skip[buf] = skip ? 1 : 0;
  // ---- 16-mers of the segments of both strands (convert_seq_int, pemapper.c:2408-2423)
  if (tid < 2 * S)
    {
      const int strand = tid / S, seg = tid - strand * S;
      const uint8_t *p = &sh.seq[buf][strand][sh.offsets[buf][seg]];
      uint32_t k = 0;
      for (int i = 0; i < 16; i++)
        k = (k << 2) + pm_code (p[i], bis);
      sh.kmer[buf][tid] = k;
    }
  pm_lds_barrier ();
  // ---- 49 bucket look-ups per segment, both strands, all in flight at once (get_mers, pemapper.c:2158-2165)
#pragma unroll
  for (int r = 0; r < NI; r++)
    {
      const int x = tid + r * PM_SEED_THREADS;
      v0[r] = v1[r] = 0;
      if (!skip && x < 2 * S * 49)
        {
          const int sg = x / 49, j = x - sg * 49;
          if (ix.n_rep == 8)
            {
              // with the look-up replicas: the bucket's entry, 8 lines per segment instead of 43 (the record header follows at the
              // commit)
              v0[r] = pm_rep_entry (ix, sh.kmer[buf][sg], j);
            }
          else
            {
              const uint32_t nb = pm_neighbour (sh.kmer[buf][sg], j);
              // one 8-byte gather for the pair (dword aligned), nothing depends on it until the commit one iteration later
              const pm_u32x2 pr = *(const pm_u32x2 *) (ix.pos_index + nb);
              v0[r] = pr.x;
              v1[r] = pr.y;
            }
        }
    }
}

template < int SMAX > __global__ __launch_bounds__ (PM_SEED_THREADS, 4) void pm_seed_kernel (PmIndex ix, PmBatch b, PmParams prm, PmHits h,
                                                                                          uint32_t * tasks_s, uint32_t * tasks_m,
                                                                                          PmCounters * ctr, uint32_t * gscratch, int scratch_blocks,
                                                                                          int phase_limit,
                                                                                          const uint32_t * end_list, const unsigned *n_list)
{
  // end_list != NULL: only the listed read-ends are processed (the ends the split look-up / vote kernels passed over)
  typedef PmSeedShared < SMAX > SH;
  __shared__ SH sh;
  constexpr int NI = (SH::NITEMS + PM_SEED_THREADS - 1) / PM_SEED_THREADS;   // look-ups per thread
  const int tid = threadIdx.x;
  const int idepth = ix.idepth;
  const int max_off = max (2, idepth - 4);
  // per-workgroup spill area for the rare strand whose positions exceed the LDS capacity: keys, survivors, order (u32 each),
  // segment numbers and counts (u8 each), PM_MAX_SEG * PM_SEG_LIST_MAX entries each
  // gscratch holds scratch_blocks such areas: a grid larger than that would write past it (the host never launches one)
  constexpr size_t GN = (size_t) PM_MAX_SEG * PM_SEG_LIST_MAX;
  if ((int) gridDim.x > scratch_blocks)
    return;
  uint32_t *g_key = gscratch + (size_t) blockIdx.x * 6 * GN;
  uint32_t *g_bkey = g_key + GN;
  uint32_t *g_surv = g_bkey + GN;
  uint32_t *g_order = g_surv + GN;
  uint8_t *g_seg = (uint8_t *) (g_order + GN);
  uint8_t *g_bseg = g_seg + GN;
  uint8_t *g_tfs = g_bseg + GN;

  uint32_t v0[NI], v1[NI];
  const uint32_t pos_index_0 = ix.pos_index[0];
  const int n_iter = end_list ? (int) *n_list : b.n_ends;
  int it = blockIdx.x;
  int buf = 0;
  if (tid < 2)
    sh.ncount[tid] = 0;
  __syncthreads ();
  if (it < n_iter)
    pm_seed_stage_a < SMAX, NI > (sh, ix, b, prm.bisulfite, end_list ? (int) end_list[it] : it, 0, v0, v1);
  for (; it < n_iter; it += gridDim.x, buf ^= 1)
    {
      const int e = end_list ? (int) end_list[it] : it;
      int len;
      (void) pm_read_ptr (b, e, &len);
      int total_cuts = len / idepth;
      if (len % idepth == 0)
        total_cuts--;
      if (total_cuts > SMAX - 1)
        total_cuts = SMAX - 1;
      const int S = total_cuts + 1;
      pm_lds_barrier ();        // the previous end's vote is done with the table that aliases the items
      const bool skip = sh.skip[buf] != 0;
      if (tid == 0)
        sh.ncount[buf ^ 1] = 0; // N counter of the buffer the next stage A fills
      // ---- commit this end's look-ups (issued one iteration ago) to LDS
#pragma unroll
      for (int r = 0; r < NI; r++)
        {
          const int x = tid + r * PM_SEED_THREADS;
          if (x < 2 * S * 49)
            {
              // the all-T k-mer's successor is entry 0: `which + 1` is evaluated in 32 bits (pemapper.c:2163)
              const int sg = x / 49, j = x - sg * 49;
              if (ix.n_rep == 8)
                pm_item_of_entry (ix, v0[r], sh.u.it.it_start[x], sh.u.it.it_len[x]);
              else
                {
                  const uint32_t nb = pm_neighbour (sh.kmer[buf][sg], j);
                  const uint32_t ln = ((nb == 0xFFFFFFFFu) ? pos_index_0 : v1[r]) - v0[r];
                  sh.u.it.it_start[x] = v0[r];
                  sh.u.it.it_len[x] = (ln >= PM_TOO_MANY) ? 0xFFFF : (uint16_t) ln;
                }
            }
        }
      pm_lds_barrier ();
      int tot = 0;
      int T0 = 0, T1 = 0;
      if (!skip && PM_PROBE (phase_limit) != 1)
        {
          // ---- a segment with any bucket >= too_many_spots is emptied (pemapper.c:1602-1606)
          if (tid < 2 * S)
            {
              int sum = 0;
              bool bad = false;
              for (int j = 0; j < 49; j++)
                {
                  const uint16_t ln = sh.u.it.it_len[tid * 49 + j];
                  sh.u.it.it_off[tid * 49 + j] = (uint16_t) sum;
                  if (ln == 0xFFFF)
                    bad = true;
                  else
                    sum += ln;
                }
              sh.seg_cnt[tid] = bad ? 0 : sum;
            }
          pm_lds_barrier ();
          if (tid < 2)
            {
              int acc = 0;
              for (int s = 0; s < S; s++)
                {
                  sh.seg_base[tid][s] = acc;
                  acc += sh.seg_cnt[tid * S + s];
                }
              sh.seg_base[tid][S] = acc;
              atomicAdd (&ctr->positions, (unsigned long long) acc);
            }
          pm_lds_barrier ();
          T0 = sh.seg_base[0][S];
          T1 = sh.seg_base[1][S];
        }
      // a strand with more positions than the LDS arrays hold goes to the workgroup's global spill area (gathered just
      // before it is voted on)
      const bool lds0 = T0 <= PM_SEED_CAP, lds1 = T1 <= PM_SEED_CAP;
      // ---- both strands' bucket slices, one position per thread and round (each slice is ascending in .mdx)
      for (int pp = tid; pp < (lds0 ? T0 : 0) + (lds1 ? T1 : 0); pp += PM_SEED_THREADS)
        {
          const int strand = (lds0 && pp < T0) ? 0 : 1;
          const int p = (strand == 1 && lds0) ? pp - T0 : pp;
          int seg = 0;
          while (p >= sh.seg_base[strand][seg + 1])
            seg++;
          const int q = p - sh.seg_base[strand][seg];
          const int x0 = (strand * S + seg) * 49;
          int lo = 0, hi = 48;  // largest j with it_off[j] <= q: that slice holds position q
          while (lo < hi)
            {
              const int mid = (lo + hi + 1) >> 1;
              if ((int) sh.u.it.it_off[x0 + mid] <= q)
                lo = mid;
              else
                hi = mid - 1;
            }
          const uint32_t m = pm_item_pos (ix, sh.u.it.it_start[x0 + lo], sh.u.it.it_len[x0 + lo], (uint32_t) (q - (int) sh.u.it.it_off[x0 + lo]));
          sh.ekey[strand][p] = m + (uint32_t) (PM_DIAG_BIAS - sh.offsets[buf][seg]);
          sh.eseg[strand][p] = (uint8_t) seg;
        }
      // ---- stage A of the workgroup's NEXT end: its look-ups fly while this end is voted on
      const int it2 = it + gridDim.x;
      if (it2 < n_iter)
        pm_seed_stage_a < SMAX, NI > (sh, ix, b, prm.bisulfite, end_list ? (int) end_list[it2] : it2, buf ^ 1, v0, v1);
      if (!skip && PM_PROBE (phase_limit) != 1 && PM_PROBE (phase_limit) != 2)
        {
          int min_match = max (1, total_cuts);       // pemapper.c:1642-1645
          if (total_cuts > 4)
            min_match = (4 * total_cuts) / 5;
          min_match = min (min_match, 4);
          bool go_on = true;
          for (int strand = 0; strand < 2 && go_on; strand++)
            {
              const int T = strand ? T1 : T0;
              const bool in_lds = strand ? lds1 : lds0;
              if (!in_lds)
                {
                  // spill path.  The items (look-up results) alias the vote table and the next end's look-ups are in
                  // registers, so this strand's look-ups are simply done again before it is gathered.
                  __syncthreads ();
                  for (int x = tid; x < S * 49; x += PM_SEED_THREADS)
                    {
                      const int sg = strand * S + x / 49, j = x % 49;
                      if (ix.n_rep == 8)
                        pm_item_of_entry (ix, pm_rep_entry (ix, sh.kmer[buf][sg], j), sh.u.it.it_start[strand * S * 49 + x], sh.u.it.it_len[strand * S * 49 + x]);
                      else
                        {
                          const uint32_t nb = pm_neighbour (sh.kmer[buf][sg], j);
                          const uint32_t i0 = ix.pos_index[nb];
                          const uint32_t ln = ix.pos_index[(uint32_t) (nb + 1u)] - i0;
                          sh.u.it.it_start[strand * S * 49 + x] = i0;
                          sh.u.it.it_len[strand * S * 49 + x] = (ln >= PM_TOO_MANY) ? 0xFFFF : (uint16_t) ln;
                        }
                    }
                  __syncthreads ();
                  if (tid < S)
                    {
                      int sum = 0;
                      for (int j = 0; j < 49; j++)
                        {
                          const uint16_t ln = sh.u.it.it_len[(strand * S + tid) * 49 + j];
                          sh.u.it.it_off[(strand * S + tid) * 49 + j] = (uint16_t) sum;
                          if (ln != 0xFFFF)
                            sum += ln;
                        }
                    }
                  __syncthreads ();
                  for (int p = tid; p < T; p += PM_SEED_THREADS)
                    {
                      int seg = 0;
                      while (p >= sh.seg_base[strand][seg + 1])
                        seg++;
                      const int q = p - sh.seg_base[strand][seg];
                      const int x0 = (strand * S + seg) * 49;
                      int lo = 0, hi = 48;
                      while (lo < hi)
                        {
                          const int mid = (lo + hi + 1) >> 1;
                          if ((int) sh.u.it.it_off[x0 + mid] <= q)
                            lo = mid;
                          else
                            hi = mid - 1;
                        }
                      const uint32_t m = pm_item_pos (ix, sh.u.it.it_start[x0 + lo], sh.u.it.it_len[x0 + lo], (uint32_t) (q - (int) sh.u.it.it_off[x0 + lo]));
                      g_key[p] = m + (uint32_t) (PM_DIAG_BIAS - sh.offsets[buf][seg]);
                      g_seg[p] = (uint8_t) seg;
                    }
                  __syncthreads ();
                }
              else
                pm_lds_barrier ();
              if (in_lds)
                pm_vote_strand < SH, uint16_t > (sh, sh.ekey[strand], sh.eseg[strand], sh.bkey, sh.bseg, sh.surv, sh.order, sh.tfs, T,
                                                 &sh.seg_cnt[strand * S], sh.offsets[buf], total_cuts, max_off, min_match, tot, go_on,
                                                 (uint8_t) strand);
              else
                pm_vote_strand < SH, uint32_t > (sh, g_key, g_seg, g_bkey, g_bseg, g_surv, g_order, g_tfs, T, &sh.seg_cnt[strand * S],
                                                 sh.offsets[buf], total_cuts, max_off, min_match, tot, go_on, (uint8_t) strand);
              if (tot >= PM_MAX_HITS)
                go_on = false;
            }
        }
      pm_lds_barrier ();
      pm_seed_emit (sh, h, e, tot);
    }
}
