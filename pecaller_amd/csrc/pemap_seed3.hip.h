// pemap_seed3.hip.h -- the seed stage of a read-end in ONE kernel: bucket look-ups against the 8 table replicas
// (fill_mers / get_mers, pemapper.c:1969-2003, 2158-2165; initial_map 1539-1690) and the diagonal vote (find_matches,
// pemapper.c:2189-2289), ONE WAVE per read-end, persistent, the positions never leaving LDS.
//
// Why a third form.  Round 1's split (pm_lookup_rep_kernel -> (key, segment) lists in HBM -> pm_vote_wave_kernel) spent
// 2,350 + 1,750 wave instructions per read-end, moved the lists through HBM twice (9 bytes per position each way) and left
// the step's length to how the runtime happened to map three streams onto hardware queues (look-ups and vote on one queue:
// 47.7 ms per step; on two: 41.7 .. 46; profiles/r02_queue_mapping.txt).  Here
//   * the look-up half decodes one (strand, segment) per round with lane = neighbour, writes the single-position buckets
//     straight into the strand's list in LDS (offsets from ballots) and resolves the record buckets from the first one or two
//     16-byte units of their records -- no per-position binary search, no prefix pass over the 49 buckets of a segment;
//   * the vote half does not sort.  A histogram of the diagonal bins of both strands finds the few positions that can be
//     anchors at all (a necessary condition: at least min_match positions in the anchor's three bins), only the positions
//     next to those are kept, and the exact count of later segments on the anchor's diagonal (tot_found, pemapper.c:2241-2249)
//     is an all-pairs test among these few.  The walk over the ranked anchors is the reference's state machine unchanged;
//   * the wave keeps three read-ends in flight: the bytes of end k+2 and the table lines of end k+1 travel while end k is
//     decoded and voted on, so one exposed trip to HBM per end (the record headers) is left of four.
// Output: the raw hit lists of pm_vote_wave_kernel (h.n_hits / spot / nn / orient), consumed by pm_emit_kernel.  Ends with
// more than PM_SEED_CAP positions on a strand, or with more than PM_S3_RCAP positions near candidate anchors (repeats), go
// to the big-end list and through pm_seed_kernel in list mode, as before.
#pragma once
#include <type_traits>
#include "pemap_wave.hip.h"

#ifndef PM_S3_NH_LOG2
#define PM_S3_NH_LOG2 11
#endif
#define PM_S3_NH (1 << PM_S3_NH_LOG2)   // cells of the bin table: hash of (diagonal bin, strand)
#ifndef PM_S3_WAVES_PER_EU
#define PM_S3_WAVES_PER_EU 3
#endif
#ifndef PM_S3_RCAP
#define PM_S3_RCAP 160          // positions next to candidate anchors, both strands together
#endif
#ifndef PM_S3_GRAB
// Read-ends are handed out through ONE counter in HBM.  Fetched once per end, that single address bounded the kernel: 85 M fetches per
// second is what it delivers, and the launch took 6.1 ms with 5, 6, 7 or 8 waves per CU alike.  A wave now takes 4 consecutive ends per
// fetch: 4.3 ms per launch with 8 waves per CU alone on the GPU (6.65 with 5: per wave the kernel is latency-bound as before).
#define PM_S3_GRAB 4            // consecutive read-ends a wave takes per fetch of the work counter
#endif
#ifndef PM_S3_CAP
#define PM_S3_CAP PM_SEED_CAP   // positions a strand's list holds for reads of up to 160 bases
#endif
#ifndef PM_S3_BIGCAP
#define PM_S3_BIGCAP 0          // 1: reads of up to 160 bases get the long reads' list capacity too (2,048 per strand, 256 near candidates)
#endif

template < int SMAX > struct __align__ (16) PmSeed3Shared
{
  static constexpr int NSEG = 2 * SMAX;
  // positions a strand's list holds: 1,024 for reads of up to 160 bases (10 segments x 49 look-ups, most of them empty or single);
  // twice that for longer reads, whose 13..19 segments gather in proportion (2 x 250 bp: 5 % of the ends passed 1,024 and took the slow
  // monolithic kernel, half of that configuration's seed time)
  static constexpr int CAP = (SMAX <= 10 && !PM_S3_BIGCAP) ? PM_S3_CAP : 2 * PM_SEED_CAP;
  // cells of the bin table and positions kept next to candidate anchors: more of both for the longer reads' fuller lists
  static constexpr int NH_LOG2 = SMAX <= 10 ? PM_S3_NH_LOG2 : PM_S3_NH_LOG2 + 1;
  static constexpr int NH = 1 << NH_LOG2;
  static constexpr int RCAP = (SMAX <= 10 && !PM_S3_BIGCAP) ? PM_S3_RCAP : 256;
  union
  {
    uint32_t lines[SMAX * 128];         // ONE strand's SMAX x 8 lines of 16 entries (the other strand's wait in registers) ...
    struct                              // ... the vote's tables afterwards (the next end's lines wait in registers too)
    {
      // bit s of cell h: a position of segment s has its diagonal in a bin that hashes to h.  16-bit cells, two to a word, while
      // the segments fit (reads of up to 256 bases)
      uint32_t segmask[SMAX <= 16 ? NH / 2 : NH];
      uint32_t candbit[NH / 32];
      uint2 r[RCAP];                    // x = key, y = segment | strand << 5 | candidate << 6
      uint2 sv[RCAP];                   // the surviving anchors: x = key, y = segment | tot_found << 8 (strand 0 from the front, 1 from the back)
      uint16_t order[RCAP];
    } v;
    struct                              // ... and in between, while the records are read: the work list of their tails
    {
      uint32_t src[128];                // per record (64 of strand 0, 64 of strand 1) with more than 3 positions: word in `multi` ...
      int32_t dst[128];                 // ... and slot in its strand's list of its 4th position, both minus the record's place in the flattened tail
      uint8_t sg[128];                  // segment | strand << 7
      uint8_t mark[2 * CAP];            // flattened tail: record number + 1 at the record's first element, 0 elsewhere
    } x;
  } a;
  // diagonal keys m + PM_DIAG_BIAS - offset(segment) per strand, from the front; while the entries are decoded the entries that
  // point to a record wait at the back (a strand has at most 49 x S <= 931 look-ups, each of them one or the other)
  uint32_t key[2][CAP];
  uint8_t tag[2][CAP];                  // segment of the position; bit 7: candidate anchor
  uint32_t hits[PM_MAX_HITS];
  uint16_t hits_off[PM_MAX_HITS];
  uint8_t hits_or[PM_MAX_HITS];
  int seg_cnt[NSEG];
  uint8_t seq[2][320];                  // 2-bit codes of the end whose k-mers are being formed
};

template < int NH_LOG2 > __device__ __forceinline__ unsigned pm_s3_hash_t (uint32_t bin, unsigned strand)
{
  return ((bin * 2u + strand) * 2654435761u) >> (32 - NH_LOG2);
}

// the bin table's cells: 16 bits each (two to a word) for up to 16 segments, 32 bits beyond
template < int SMAX > __device__ __forceinline__ void pm_s3_mask_or (uint32_t * tab, unsigned h, unsigned seg)
{
  if (SMAX <= 16)
    atomicOr (&tab[h >> 1], (1u << seg) << (16u * (h & 1u)));
  else
    atomicOr (&tab[h], 1u << seg);
}

template < int SMAX > __device__ __forceinline__ uint32_t pm_s3_mask_get (const uint32_t * tab, unsigned h)
{
  if (SMAX <= 16)
    return (tab[h >> 1] >> (16u * (h & 1u))) & 0xFFFFu;
  return tab[h];
}

// LDS is passed as dynamic shared memory (sizeof (PmSeed3Shared < SMAX >)): from a static 30 KB the compiler concludes that two waves
// per SIMD are all the kernel will ever get and spends 185 VGPRs on it, whatever the launch bounds say -- and a SIMD that hosts such
// a wave has room for one wave of the fp64 SW kernel (168 VGPRs) instead of two.
extern __shared__ __align__ (16) uint8_t pm_seed3_lds[];

// Cycle probes of the kernel's phases (a library built with -DPEMAP_TIMING_PROBES only): per wave the core-clock cycles between
// the marks below, summed over the grid into pm_s3_probe[]; pemap_capi.hip prints and clears them after every run.
#ifdef PEMAP_TIMING_PROBES
__device__ unsigned long long pm_s3_probe[32];      // 16 phases; 16..18: segments decoded, dropped (a too-many bucket), dropped by the k-mer's own bucket;
                                                    // 20: ends decoded; 21..23: left to the big-end kernel by the records' room, the list capacity, the candidates' capacity;
                                                    // 24..27: records of 2-3, 4-7, 8-15, 16+ positions; 28..31: ends with a record of 8+, of 16+, of 4+, with any record
#define PM_S3_T(i) do { const unsigned long long t_ = __builtin_readcyclecounter (); pacc[i] += t_ - plast; plast = t_; } while (0)
#else
#define PM_S3_T(i) do { } while (0)
#endif

template < int SMAX > __global__ __launch_bounds__ (64, SMAX <= 10 ? PM_S3_WAVES_PER_EU : 2) void pm_seed3_kernel (PmIndex ix, PmBatch b, PmParams prm, PmHits h, PmLists out, int prio)
{
  pm_set_prio (prio);
  typedef PmSeed3Shared < SMAX > SH;
  SH & sh = *reinterpret_cast < SH * >(pm_seed3_lds);
  const int lane = threadIdx.x;
  const int idepth = ix.idepth;
  const int max_off = max (2, idepth - 4);
  const uint32_t span = (uint32_t) (2 * (max_off - 1));
  const uint32_t multi_base = ix.multi_base;
  const int n_ends = b.n_ends;
  const int nb = b.stride < 320 ? b.stride : 320;
  unsigned long long n_pos = 0;
  // lane j looks at neighbour j of every segment (fill_mers' order, pm_neighbour): the 2-bit field it replaces, the
  // alternative's rank, the replica (= 4-bit field) whose line holds the entry
  const int nb_f = lane > 0 ? (lane - 1) / 3 : 0;
  const uint32_t nb_a = lane > 0 ? (uint32_t) ((lane - 1) % 3) : 0u;
  const uint32_t nb_sh = 2u * (uint32_t) (nb_f & 15);
  const uint32_t nb_keep = lane > 0 ? ~(3u << nb_sh) : 0xFFFFFFFFu;      // lane 0: the k-mer itself
  const uint32_t nb_alt_on = lane > 0 ? 0xFFFFFFFFu : 0u;
  const int nb_p = (nb_f >> 1) & 7;
  const uint32_t nb_p4 = 4u * (uint32_t) nb_p, nb_pw = 16u * (uint32_t) nb_p;

  // ends are handed out through a counter, two values ahead of their use (the first grid-ful by block index)
  // (the grid size is read from the dispatch packet with a vector load: brought to a scalar register here, before the loop, or the
  // loop's first use of it carries a "wait for every outstanding load" into every iteration)
  const int grid_n = __builtin_amdgcn_readfirstlane ((int) gridDim.x);
  // (PM_S3_GRAB ends per fetch: the counter counts blocks of that many consecutive ends; block b = ends [b * GRAB, (b + 1) * GRAB))
  int blkR = blockIdx.x, idxR = 0;
  auto next_end = [&] ()->int
  {
    const int e = blkR * PM_S3_GRAB + idxR;
    if (++idxR == PM_S3_GRAB)
      {
        idxR = 0;
        blkR = grid_n + (int) __builtin_amdgcn_readfirstlane ((int) (lane == 0 ? atomicAdd (out.next_end, 1u) : 0u));
      }
    return e;
  };
  int eQ = next_end ();
  int eP = next_end ();
  int eR = blkR * PM_S3_GRAB + idxR;          // (the end after the next two: its block is known, the fetch of the block after it is the loop's)

  uint8_t rb[5];                // the bytes of the end whose k-mers are formed next
  int rlen = 0;
  uint4 ln[SMAX];               // the table lines of the end decoded next: 4 lanes x 16 bytes per line, 16 lines per register
  auto load_bytes = [&] (int e)
  {
    const uint8_t *src = pm_read_ptr (b, e, &rlen);
#pragma unroll
    for (int t = 0; t < 5; t++)
      {
        const int i = lane + 64 * t;
        rb[t] = (i < nb) ? src[i] : (uint8_t) 0;
      }
  };
  // read in registers -> 2-bit codes of both strands, N filter (pemapper.c:1552-1559), segment count, the 2 x S k-mers into
  // kmer[bf], and the 2 x S x 8 line requests into ln[].  -> S, or 0 when the N filter drops the read
  // (the length is a per-lane load of one address: told to be wave-uniform here, where it is first needed, so that everything
  // derived from it -- segment counts, loop bounds, the branches on them -- lives in scalar registers)
  // (`between` runs after the read's bytes have been consumed and before the line requests are issued: the memory counter is in
  // order, so whatever is issued BEFORE the point that waits for the bytes is waited for as well, and the compiler cannot count
  // conditional stores -- the loop puts the previous end's output and the work counter's atomic there)
  auto stage_p = [&] (uint32_t & kmer_out, int &len_out, auto between)->int
  {
    const int len = __builtin_amdgcn_readfirstlane (rlen);
    len_out = len;
    int isn = 0;                // (counted with ballots: scalar)
#pragma unroll
    for (int t = 0; t < 5; t++)
      {
        const int i = lane + 64 * t;
        const uint8_t c = rb[t];
        if (i < len)
          {
            // fill_cv_mat / convert_ct (pemapper.c:2375-2383, 2292-2300) of the read and of its reverse complement
            sh.seq[0][i] = (uint8_t) pm_code_flat (c, prm.bisulfite);
            sh.seq[1][len - 1 - i] = (uint8_t) pm_code_flat (pm_rc_flat (c), prm.bisulfite);
          }
        isn += (int) __popcll (__ballot (i < len && c == 'N'));
      }
    int cuts = len / idepth;    // pemapper.c:1573-1587
    if (len % idepth == 0)
      cuts--;
    if (cuts > SMAX - 1)
      cuts = SMAX - 1;
    cuts = __builtin_amdgcn_readfirstlane (cuts);       // (integer division is done by the vector unit)
    const int S = cuts + 1;
    pm_wave_sync ();
    between ();
    if (isn >= __builtin_amdgcn_readfirstlane (1 + len / 10))
      return 0;
    if (lane < 2 * S)
      {
        const int strand = lane >= S ? 1 : 0, seg = lane - strand * S;
        const int off = (seg < cuts || cuts == 0) ? seg * idepth : len - idepth;
        const uint8_t *p = &sh.seq[strand][off];
        uint32_t k = 0;
#pragma unroll
        for (int i = 0; i < 16; i++)
          k = (k << 2) + p[i];
        kmer_out = k;
      }
    const int n_lines = 2 * S * 8;
#pragma unroll
    for (int r = 0; r < SMAX; r++)
      {
        const int li = r * 16 + (lane >> 2);
        ln[r] = make_uint4 (0u, 0u, 0u, 0u);
        // (li >> 3 = 2 r + (lane >> 5): the two k-mers a register's lines belong to come over the scalar side)
        const uint32_t k_lo = (uint32_t) __builtin_amdgcn_readlane ((int) kmer_out, (2 * r) & 63), k_hi = (uint32_t) __builtin_amdgcn_readlane ((int) kmer_out, (2 * r + 1) & 63);
        const uint32_t ksrc = lane < 32 ? k_lo : k_hi;
        if (li < n_lines)
          {
            const int p = li & 7;
            const uint32_t idx = pm_swap_fields (ksrc, p);
            ln[r] = *(const uint4 *) (ix.rep + ((size_t) p << 32) + (size_t) (idx & ~15u) + (size_t) ((lane & 3) * 4));
          }
      }
    return S;
  };

  int SQ = 0, lenQ = 0;
  uint32_t kQ = 0;              // lane sg: the k-mer of (strand, segment) sg of the end being decoded
  if (eQ < n_ends)
    {
      load_bytes (eQ);
      SQ = stage_p (kQ, lenQ, [] () { });
    }
  if (eP < n_ends)
    load_bytes (eP);

  // the end whose hits are still in LDS (written out one iteration later, see O below)
  int e_out = -1, tot_out = 0;
  bool big_out = false;
  auto flush_out = [&] ()
  {
    if (e_out < 0)
      return;
    if (big_out)
      {
        if (lane == 0)
          out.big_list[atomicAdd (out.n_big, 1u)] = (uint32_t) e_out;
      }
    else
      {
        // ---- raw hits out (pm_seed_emit's format); pm_emit_kernel makes windows and task lists of them
        if (lane == 0)
          h.n_hits[e_out] = tot_out;
        for (int t = lane; t < tot_out; t += 64)
          {
            const size_t o = (size_t) e_out * PM_MAX_HITS + t;
            h.spot[o] = sh.hits[t];
            h.nn[o] = (int16_t) sh.hits_off[t];
            h.orient[o] = sh.hits_or[t];
          }
      }
    e_out = -1;
  };
#ifdef PEMAP_TIMING_PROBES
  unsigned long long pacc[32];
  for (int i_ = 0; i_ < 32; i_++)
    pacc[i_] = 0ull;
  bool p_any4 = false, p_any8 = false, p_any16 = false;
  unsigned long long plast = __builtin_readcyclecounter ();
#endif
  while (eQ < n_ends)
    {
      // (loop-carried and wave-uniform: said so, or the compiler keeps them, and every branch on them, in vector registers)
      const int e = eQ, S = __builtin_amdgcn_readfirstlane (SQ), len = __builtin_amdgcn_readfirstlane (lenQ);
      int cuts = S - 1;
      const int last_off = len - idepth;
      int tot = 0;
      bool big = false;
      int T0 = 0, T1 = 0, cmin0 = 0, cmin1 = 0;
      if (S > 0)
        {
          // ---- A: the lines of this end (requested one iteration ago) go from registers to LDS one strand at a time (lines
          //      [strand * S * 8, (strand + 1) * S * 8) of the 2 x S x 8), so that the buffer is half the size
          auto lines_to_lds = [&] (int strand)
          {
            const int first = strand * S * 8, last = first + S * 8;
#pragma unroll
            for (int r = 0; r < SMAX; r++)
              {
                const int li = r * 16 + (lane >> 2);
                if (li >= first && li < last)
                  *(uint4 *) (&sh.a.lines[(li - first) * 16 + (lane & 3) * 4]) = ln[r];
              }
          };
          lines_to_lds (0);
          if (lane < 2 * SMAX)
            sh.seg_cnt[lane] = 0;
          pm_wave_sync ();
          PM_S3_T (0);
          // ---- B: one (strand, segment) per round, lane j = neighbour j.  A segment with a bucket of too_many_spots or more is
          //      dropped whole (pemapper.c:1602-1606: the entry itself says so); buckets of one position go straight to the
          //      front of the strand's list; entries that point to a record are parked at its back
          // (the two strands are two instances of the same code: their counters stay in scalar registers)
          int nf0 = 0, nf1 = 0, nm0 = 0, nm1 = 0;
          auto decode_strand = [&] (auto ST)
          {
            constexpr int strand = decltype (ST)::value;
            int &nf = strand ? nf1 : nf0, &nm = strand ? nm1 : nm0;
            // neighbour `lane` of segment sg: its entry, from the segment's 8 lines
            auto entry_of = [&] (int sg)->uint32_t
            {
              const uint32_t k = (uint32_t) __builtin_amdgcn_readlane ((int) kQ, sg);
              const uint32_t cur = (k >> nb_sh) & 3u;
              const uint32_t alt = nb_a + (nb_a >= cur ? 1u : 0u);
              const uint32_t nbk = (k & nb_keep) | ((alt << nb_sh) & nb_alt_on);
              return lane < 49 ? sh.a.lines[(sg - strand * S) * 128 + nb_pw + ((nbk >> nb_p4) & 15u)] : 0xFFFFFFFFu;
            };
            uint32_t ent_next = entry_of (strand * S);
#pragma unroll 1
            for (int seg = 0; seg < S; seg++)
              {
                const int sg = strand * S + seg;
                const uint32_t ent = ent_next;
                if (seg + 1 < S)
                  ent_next = entry_of (sg + 1);         // (its LDS read flies while this segment is filed)
#ifdef PEMAP_TIMING_PROBES
                {
                  const unsigned long long tm_ = __ballot (ent == 0xFFFFFFFEu);
                  pacc[16] += 1ull;
                  pacc[17] += tm_ != 0ull;
                  pacc[18] += (tm_ & 1ull) != 0ull;
                }
#endif
                if (__ballot (ent == 0xFFFFFFFEu) != 0ull)
                  continue;
                const bool single = ent < multi_base, multi = ent >= multi_base && ent != 0xFFFFFFFFu;
                const unsigned long long bs = __ballot (single), bm = __ballot (multi);
                const int off = (seg < cuts || cuts == 0) ? seg * idepth : last_off;
                if (single)
                  {
                    const int at = nf + pm_lanes_below (bs);
                    sh.key[strand][at] = ent + (uint32_t) (PM_DIAG_BIAS - off);
                    sh.tag[strand][at] = (uint8_t) seg;
                  }
                if (multi)
                  {
                    const int at = SH::CAP - 1 - (nm + pm_lanes_below (bm));
                    sh.key[strand][at] = ent - multi_base;      // the record's first 16-byte unit
                    sh.tag[strand][at] = (uint8_t) seg;
                  }
                const int ns = (int) __popcll (bs);
                nf += ns;
                nm += (int) __popcll (bm);
                if (lane == 0)
                  sh.seg_cnt[sg] = ns;
              }
          };
          decode_strand (std::integral_constant < int, 0 > { });
          pm_wave_sync ();
          PM_S3_T (1);
          lines_to_lds (1);
          pm_wave_sync ();
          PM_S3_T (2);
          decode_strand (std::integral_constant < int, 1 > { });
          pm_wave_sync ();
          PM_S3_T (3);
          // ---- C: the records: {count, positions...} in 16-byte units; the first unit answers for buckets of up to 3 positions,
          //      the second for up to 7, longer ones are copied by the whole wave.  The first 64 records of BOTH strands are
          //      requested before either strand is filed (one exposed HBM latency instead of two); what a strand has beyond them
          //      follows in rounds of 2 x 64.  The positions go to the front of the list whose back still holds the records not
          //      yet read: an end whose list would reach them is left to the monolithic kernel (only ends close to the capacity
          //      anyway).
          // one round: lane = record (count in hdr.x, its first three positions behind it), `limit` = where this strand's front
          // must stop
          auto rec_round = [&] (auto ST, const uint4 & hdr, const int seg, const uint32_t unit_, const bool valid, const int limit, auto more, uint32_t & c_out, int &dst_out, bool & wr_out)
          {
            constexpr int strand = decltype (ST)::value;
            int &nf = strand ? nf1 : nf0;
            const uint32_t c = valid ? hdr.x : 0u;
#ifdef PEMAP_TIMING_PROBES
            pacc[24] += (unsigned long long) __popcll (__ballot (valid && c <= 3u));
            pacc[25] += (unsigned long long) __popcll (__ballot (valid && c > 3u && c <= 7u));
            pacc[26] += (unsigned long long) __popcll (__ballot (valid && c > 7u && c <= 15u));
            pacc[27] += (unsigned long long) __popcll (__ballot (valid && c > 15u));
            p_any4 = p_any4 || __ballot (valid && c > 3u) != 0ull;
            p_any8 = p_any8 || __ballot (valid && c > 7u) != 0ull;
            p_any16 = p_any16 || __ballot (valid && c > 15u) != 0ull;
#endif
            const uint32_t incl = pm_wave_incl_sum (c);
            const int dst = nf + (int) (incl - c);
            nf += __builtin_amdgcn_readlane ((int) incl, 63);
            if (nf > limit)
              {
#ifdef PEMAP_TIMING_PROBES
                pacc[21] += big ? 0ull : 1ull;
#endif
                big = true;     // (wave-uniform) nothing more is written for this end: it goes to the big-end list
              }
            const bool wr = valid && !big;
            const int off = (seg < cuts || cuts == 0) ? seg * idepth : last_off;
            const uint32_t bias = (uint32_t) (PM_DIAG_BIAS - off);
            if (wr)
              {
                atomicAdd (&sh.seg_cnt[strand * S + seg], (int) c);
                uint32_t *kk = &sh.key[strand][dst];
                uint8_t *tt = &sh.tag[strand][dst];
                kk[0] = hdr.y + bias;
                tt[0] = (uint8_t) seg;
                kk[1] = hdr.z + bias;
                tt[1] = (uint8_t) seg;
                if (c > 2)
                  {
                    kk[2] = hdr.w + bias;
                    tt[2] = (uint8_t) seg;
                  }
              }
            PM_S3_T (13);
            c_out = c;
            dst_out = dst;
            wr_out = wr;
            if (!decltype (more)::value)
              return;
            const bool need2 = wr && c > 3;
            if (__ballot (need2) != 0ull)
              {
                if (need2)
                  {
                    const uint4 h2 = *(const uint4 *) (ix.multi + (size_t) unit_ * 4 + 4);
                    uint32_t *kk = &sh.key[strand][dst];
                    uint8_t *tt = &sh.tag[strand][dst];
                    kk[3] = h2.x + bias;
                    tt[3] = (uint8_t) seg;
                    if (c > 4)
                      {
                        kk[4] = h2.y + bias;
                        tt[4] = (uint8_t) seg;
                      }
                    if (c > 5)
                      {
                        kk[5] = h2.z + bias;
                        tt[5] = (uint8_t) seg;
                      }
                    if (c > 6)
                      {
                        kk[6] = h2.w + bias;
                        tt[6] = (uint8_t) seg;
                      }
                  }
              }
            PM_S3_T (14);
            unsigned long long bl = __ballot (wr && c > 7);
            while (bl)
              {
                const int l = __ffsll ((long long) bl) - 1;
                bl &= bl - 1;
                const int cc = __builtin_amdgcn_readlane ((int) c, l), dd = __builtin_amdgcn_readlane (dst, l), sgl = __builtin_amdgcn_readlane (seg, l);
                const uint32_t bl_bias = (uint32_t) __builtin_amdgcn_readlane ((int) bias, l);
                const uint32_t *rec = ix.multi + (size_t) (uint32_t) __builtin_amdgcn_readlane ((int) unit_, l) * 4 + 1;
                for (int q = 7 + lane; q < cc; q += 64)
                  {
                    sh.key[strand][dd + q] = rec[q] + bl_bias;
                    sh.tag[strand][dd + q] = (uint8_t) sgl;
                  }
              }
            PM_S3_T (15);
          };
          // a strand's records from the 65th on
          auto records_rest = [&] (auto ST)
          {
            constexpr int strand = decltype (ST)::value;
            const int nm = strand ? nm1 : nm0;
#pragma unroll 1
            for (int i0 = 64; i0 < nm && !big; i0 += 128)
              {
                uint4 hd[2];
                int sgv[2];
                uint32_t unit[2];
#pragma unroll
                for (int r = 0; r < 2; r++)
                  {
                    const int i = i0 + r * 64 + lane;
                    hd[r] = make_uint4 (0u, 0u, 0u, 0u);
                    sgv[r] = 0;
                    unit[r] = 0;
                    if (i < nm)
                      {
                        unit[r] = sh.key[strand][SH::CAP - 1 - i];
                        sgv[r] = sh.tag[strand][SH::CAP - 1 - i];
                        hd[r] = *(const uint4 *) (ix.multi + (size_t) unit[r] * 4);
                      }
                  }
                const int limit = (i0 + 128 < nm) ? SH::CAP - nm : SH::CAP;
#pragma unroll
                for (int r = 0; r < 2; r++)
                  if (i0 + r * 64 < nm && !big)
                    {
                      uint32_t c_;
                      int d_;
                      bool w_;
                      rec_round (ST, hd[r], sgv[r], unit[r], i0 + r * 64 + lane < nm, limit, std::true_type { }, c_, d_, w_);
                    }
              }
          };
          {
            uint4 hA = make_uint4 (0u, 0u, 0u, 0u), hB = make_uint4 (0u, 0u, 0u, 0u);
            int sgA = 0, sgB = 0;
            uint32_t uA = 0, uB = 0;
            if (lane < nm0)
              {
                uA = sh.key[0][SH::CAP - 1 - lane];
                sgA = sh.tag[0][SH::CAP - 1 - lane];
                hA = *(const uint4 *) (ix.multi + (size_t) uA * 4);
              }
            if (lane < nm1)
              {
                uB = sh.key[1][SH::CAP - 1 - lane];
                sgB = sh.tag[1][SH::CAP - 1 - lane];
                hB = *(const uint4 *) (ix.multi + (size_t) uB * 4);
              }
#ifdef PEMAP_TIMING_PROBES
            { const uint32_t w_ = hA.x + hB.x; asm volatile ("" :: "v" (w_)); }      // (the loads have landed)
            PM_S3_T (12);
#endif
            // the first three positions of each record; the 4th and later ones of all 2 x 64 records as ONE flattened list, lane =
            // element, so that their loads are in flight together whatever records they belong to (record by record, every record
            // of more than 7 positions cost a memory round trip of its own): the record of element t is the largest record number
            // marked at or before t (a prefix maximum), its source word and list slot are affine in t
            uint32_t cA = 0, cB = 0;
            int dA = 0, dB = 0;
            bool wA = false, wB = false;
            if (nm0 > 0)
              rec_round (std::integral_constant < int, 0 > { }, hA, sgA, uA, lane < nm0, nm0 > 64 ? SH::CAP - nm0 : SH::CAP, std::false_type { }, cA, dA, wA);
            if (nm1 > 0 && !big)
              rec_round (std::integral_constant < int, 1 > { }, hB, sgB, uB, lane < nm1, nm1 > 64 ? SH::CAP - nm1 : SH::CAP, std::false_type { }, cB, dB, wB);
            if (!big)
              {
                const uint32_t restA = (wA && cA > 3u) ? cA - 3u : 0u, restB = (wB && cB > 3u) ? cB - 3u : 0u;
                const uint32_t inA = pm_wave_incl_sum (restA), inB = pm_wave_incl_sum (restB);
                const int totA = __builtin_amdgcn_readlane ((int) inA, 63), M = totA + __builtin_amdgcn_readlane ((int) inB, 63);
                if (M > 0)
                  {
                    for (int i = lane; i < (M + 15) / 16; i += 64)
                      ((uint4 *) sh.a.x.mark)[i] = make_uint4 (0u, 0u, 0u, 0u);
                    pm_wave_sync ();
                    if (restA)
                      {
                        const int ro = (int) (inA - restA);
                        sh.a.x.mark[ro] = (uint8_t) (lane + 1);
                        sh.a.x.src[lane] = uA * 4u + 4u - (uint32_t) ro;
                        sh.a.x.dst[lane] = dA + 3 - ro;
                        sh.a.x.sg[lane] = (uint8_t) sgA;
                      }
                    if (restB)
                      {
                        const int ro = totA + (int) (inB - restB);
                        sh.a.x.mark[ro] = (uint8_t) (lane + 65);
                        sh.a.x.src[64 + lane] = uB * 4u + 4u - (uint32_t) ro;
                        sh.a.x.dst[64 + lane] = dB + 3 - ro;
                        sh.a.x.sg[64 + lane] = (uint8_t) (sgB | 0x80);
                      }
                    pm_wave_sync ();
                    int carry = 0;
#pragma unroll 1
                    for (int t0 = 0; t0 < M; t0 += 256)
                      {
                        uint32_t val[4];
                        int own[4];
#pragma unroll
                        for (int j = 0; j < 4; j++)
                          {
                            const int t = t0 + 64 * j + lane;
                            const int m = t < M ? (int) sh.a.x.mark[t] : 0;
                            const int o = max (pm_wave_incl_max (m), carry);
                            carry = __builtin_amdgcn_readlane (o, 63);
                            own[j] = o - 1;
                            val[j] = 0u;
                            if (t < M)
                              val[j] = ix.multi[(size_t) (sh.a.x.src[o - 1] + (uint32_t) t)];
                          }
#pragma unroll
                        for (int j = 0; j < 4; j++)
                          {
                            const int t = t0 + 64 * j + lane;
                            if (t < M)
                              {
                                const int o = own[j];
                                const int d = sh.a.x.dst[o] + t;
                                const int sgs = (int) sh.a.x.sg[o];
                                const int seg = sgs & 31, st = sgs >> 7;
                                const int off = (seg < cuts || cuts == 0) ? seg * idepth : last_off;
                                sh.key[st][d] = val[j] + (uint32_t) (PM_DIAG_BIAS - off);
                                sh.tag[st][d] = (uint8_t) seg;
                              }
                          }
                      }
                  }
                PM_S3_T (15);
                records_rest (std::integral_constant < int, 0 > { });
                if (!big)
                  records_rest (std::integral_constant < int, 1 > { });
              }
          }
          pm_wave_sync ();
          PM_S3_T (4);
          T0 = nf0;
          T1 = nf1;
#ifdef PEMAP_TIMING_PROBES
          pacc[20] += 1ull;
          pacc[22] += (!big && (T0 > SH::CAP || T1 > SH::CAP)) ? 1ull : 0ull;
          pacc[28] += p_any8 ? 1ull : 0ull;
          pacc[29] += p_any16 ? 1ull : 0ull;
          pacc[30] += p_any4 ? 1ull : 0ull;
          pacc[31] += (nm0 + nm1 > 0) ? 1ull : 0ull;
          p_any4 = p_any8 = p_any16 = false;
#endif
          big = big || T0 > SH::CAP || T1 > SH::CAP;
          // pemapper.c:2200-2207: a strand is not searched when every one of its segments holds more than max_hits positions
          {
            const int c = lane < 2 * S ? sh.seg_cnt[lane] : 10000;
            cmin0 = pm_wave_min (lane < S ? c : 10000);
            cmin1 = pm_wave_min (lane >= S ? c : 10000);
          }
        }
      // ---- O: the PREVIOUS end's hits leave LDS (its vote wrote them, this end's vote has not run yet), and the counter hands out the
      //      end after the next two.  Both are issued here, ahead of the loads below and with this end's vote between them and the
      //      next wait on the memory counter: at the bottom of the loop they made every iteration wait for an atomic's round trip.
      // (atomicInc: the compiler's wave-aggregation of atomicAdd reads the result back at once; only lane 0's value is ever read)
      uint32_t raw_next = 0u;
      auto out_and_next = [&] ()
      {
        flush_out ();
        if (lane == 0 && idxR == PM_S3_GRAB - 1)
          raw_next = atomicInc (out.next_end, 0xFFFFFFFFu);
      };
      // ---- P: the next end's k-mers and line requests (its bytes arrived during the previous iteration); R: the bytes of the end after
      int SP = 0, lenP = 0;
      uint32_t kP = 0;
      if (eP < n_ends)
        SP = stage_p (kP, lenP, out_and_next);
      else
        out_and_next ();
      if (eR < n_ends)
        load_bytes (eR);
      PM_S3_T (5);
      // ---- V: find_matches (pemapper.c:2189-2289) on the lists in LDS
      if (S > 0 && !big)
        {
          n_pos += (unsigned long long) (T0 + T1);
          int min_match = max (1, cuts);        // pemapper.c:1642-1645
          if (cuts > 4)
            min_match = (4 * cuts) / 5;
          min_match = min (min_match, 4);
          const int mm0 = min_match;
          const int loop_max0 = 1 + cuts - mm0;
          const bool use0 = cmin0 <= PM_MAX_HITS, use1 = cmin1 <= PM_MAX_HITS;
          // which segments have a position in each diagonal bin (16 diagonals wide), both strands
#pragma unroll
          for (int i = 0; i < (int) (sizeof (sh.a.v.segmask) / 1024); i++)
            ((uint4 *) sh.a.v.segmask)[lane + 64 * i] = make_uint4 (0u, 0u, 0u, 0u);
          for (int i = lane; i < SH::NH / 32; i += 64)
            sh.a.v.candbit[i] = 0u;
          pm_wave_sync ();
          // (each pass takes the positions four rounds at a time: the LDS reads of a batch are issued together)
          for (int st = 0; st < 2; st++)
            if (st ? use1 : use0)
              {
                const int T = st ? T1 : T0;
                for (int i0 = 0; i0 < T; i0 += 256)
                  {
                    uint32_t kk[4], tg[4];
#pragma unroll
                    for (int j = 0; j < 4; j++)
                      {
                        const int i = i0 + 64 * j + lane;
                        kk[j] = i < T ? sh.key[st][i] : 0u;
                        tg[j] = i < T ? (uint32_t) sh.tag[st][i] : 0u;
                      }
#pragma unroll
                    for (int j = 0; j < 4; j++)
                      if (i0 + 64 * j + lane < T)
                        pm_s3_mask_or < SMAX > (sh.a.v.segmask, pm_s3_hash_t < SH::NH_LOG2 > (kk[j] >> 4, (unsigned) st), tg[j]);
                  }
              }
          pm_wave_sync ();
          PM_S3_T (6);
          unsigned long long any_cand = 0ull;
          // candidate anchors: positions of a segment the walk can reach whose three bins hold at least min_match - 1 LATER segments
          // (everything within max_off - 1 <= 15 diagonals of an anchor lies in those bins; colliding bins only add candidates)
          for (int st = 0; st < 2; st++)
            if (st ? use1 : use0)
              {
                const int T = st ? T1 : T0;
                for (int i0 = 0; i0 < T; i0 += 256)
                  {
                    uint32_t kk[4], tg[4];
#pragma unroll
                    for (int j = 0; j < 4; j++)
                      {
                        const int i = i0 + 64 * j + lane;
                        kk[j] = i < T ? sh.key[st][i] : 0u;
                        tg[j] = i < T ? (uint32_t) sh.tag[st][i] : 0u;
                      }
                    uint32_t mk[4];
#pragma unroll
                    for (int j = 0; j < 4; j++)
                      {
                        const uint32_t bin = kk[j] >> 4;
                        mk[j] = pm_s3_mask_get < SMAX > (sh.a.v.segmask, pm_s3_hash_t < SH::NH_LOG2 > (bin - 1u, (unsigned) st)) | pm_s3_mask_get < SMAX > (sh.a.v.segmask, pm_s3_hash_t < SH::NH_LOG2 > (bin, (unsigned) st))
                          | pm_s3_mask_get < SMAX > (sh.a.v.segmask, pm_s3_hash_t < SH::NH_LOG2 > (bin + 1u, (unsigned) st));
                      }
#pragma unroll
                    for (int j = 0; j < 4; j++)
                      {
                        const int i = i0 + 64 * j + lane;
                        const int sa = (int) tg[j];
                        const bool is_cand = i < T && sa <= loop_max0 && 1 + __popc (mk[j] & ~((2u << sa) - 1u)) >= mm0;
                        any_cand |= __ballot (is_cand);
                        if (is_cand)
                          {
                            const unsigned hc = pm_s3_hash_t < SH::NH_LOG2 > (kk[j] >> 4, (unsigned) st);
                            sh.tag[st][i] = (uint8_t) (sa | 0x80);
                            atomicOr (&sh.a.v.candbit[hc >> 5], 1u << (hc & 31u));
                          }
                      }
                  }
              }
          pm_wave_sync ();
          PM_S3_T (7);
          // the positions next to a candidate (same or adjacent bin), both strands, compacted
          int nR = 0;
          for (int st = 0; st < 2 && any_cand != 0ull; st++)      // (most ends that do not map have no candidate at all)
            if (st ? use1 : use0)
              {
                const int T = st ? T1 : T0;
                for (int i0 = 0; i0 < T; i0 += 256)
                  {
                    uint32_t kk[4], tg[4], cb[4];
#pragma unroll
                    for (int j = 0; j < 4; j++)
                      {
                        const int i = i0 + 64 * j + lane;
                        kk[j] = i < T ? sh.key[st][i] : 0u;
                        tg[j] = i < T ? (uint32_t) sh.tag[st][i] : 0u;
                      }
#pragma unroll
                    for (int j = 0; j < 4; j++)
                      {
                        const uint32_t bin = kk[j] >> 4;
                        const unsigned h0 = pm_s3_hash_t < SH::NH_LOG2 > (bin - 1u, (unsigned) st), h1 = pm_s3_hash_t < SH::NH_LOG2 > (bin, (unsigned) st), h2 = pm_s3_hash_t < SH::NH_LOG2 > (bin + 1u, (unsigned) st);
                        cb[j] = ((sh.a.v.candbit[h0 >> 5] >> (h0 & 31u)) | (sh.a.v.candbit[h1 >> 5] >> (h1 & 31u)) | (sh.a.v.candbit[h2 >> 5] >> (h2 & 31u))) & 1u;
                      }
#pragma unroll
                    for (int j = 0; j < 4; j++)
                      if (i0 + 64 * j < T)
                        {
                          const bool rel = i0 + 64 * j + lane < T && cb[j] != 0u;
                          const unsigned long long br = __ballot (rel);
                          if (rel)
                            {
                              const int at = nR + pm_lanes_below (br);
                              if (at < SH::RCAP)
                                sh.a.v.r[at] = make_uint2 (kk[j], (tg[j] & 31u) | ((uint32_t) st << 5) | ((tg[j] & 0x80u) >> 1));
                            }
                          nR += (int) __popcll (br);
                        }
                  }
              }
          pm_wave_sync ();
          PM_S3_T (8);
          if (nR > SH::RCAP)
            {
#ifdef PEMAP_TIMING_PROBES
              pacc[23] += 1ull;
#endif
              big = true;       // a repeat: left to the monolithic kernel
            }
          else
            {
              // tot_found of every candidate (pemapper.c:2241-2249): 1 + the later segments with a position within max_off of its
              // diagonal; the anchors that reach min_match, per strand, compacted: sv[] (strand 0 from the front, strand 1 from the back)
              int ns0 = 0, ns1 = 0;
              for (int i0 = 0; i0 < nR; i0 += 64)
                {
                  const int i = i0 + lane;
                  const uint2 me = i < nR ? sh.a.v.r[i] : make_uint2 (0u, 0u);
                  const bool cand = (me.y & 0x40u) != 0u;
                  uint32_t bits = 0;
                  if (__ballot (cand) != 0ull)
                    for (int c0 = 0; c0 < nR; c0 += 64)
                      {
                        // the others, 64 at a time in registers: handed round by readlane, no LDS trip per pair
                        const uint2 oth = c0 + lane < nR ? sh.a.v.r[c0 + lane] : make_uint2 (0u, 0u);
                        const int ny = min (64, nR - c0);
                        for (int y = 0; y < ny; y++)
                          {
                            const uint32_t ox = (uint32_t) __builtin_amdgcn_readlane ((int) oth.x, y), oy = (uint32_t) __builtin_amdgcn_readlane ((int) oth.y, y);
                            const uint32_t dd = ox - me.x + (uint32_t) (max_off - 1);   // |diag_y - diag_a| < max_off in wrapping arithmetic
                            if (dd <= span && ((oy ^ me.y) & 0x20u) == 0u && (oy & 31u) > (me.y & 31u))
                              bits |= 1u << (oy & 31u);
                          }
                      }
                  const int tf = 1 + __popc (bits);
                  const bool surv = cand && tf >= mm0;
                  const bool s1 = surv && (me.y & 0x20u) != 0u, s0 = surv && (me.y & 0x20u) == 0u;
                  const unsigned long long b0 = __ballot (s0), b1 = __ballot (s1);
                  if (s0)
                    sh.a.v.sv[ns0 + pm_lanes_below (b0)] = make_uint2 (me.x, (me.y & 31u) | ((uint32_t) tf << 8));
                  if (s1)
                    sh.a.v.sv[SH::RCAP - 1 - (ns1 + pm_lanes_below (b1))] = make_uint2 (me.x, (me.y & 31u) | ((uint32_t) tf << 8));
                  ns0 += (int) __popcll (b0);
                  ns1 += (int) __popcll (b1);
                }
              pm_wave_sync ();
              PM_S3_T (9);
              bool go_on = true;
              for (int strand = 0; strand < 2 && go_on; strand++)
                {
                  if (!(strand ? use1 : use0))
                    {
                      tot = 0;
                      continue;
                    }
                  // ---- walk order of this strand's surviving anchors: segment ascending, position ascending inside a segment
                  const int ns = strand ? ns1 : ns0;
                  const uint2 *svp = strand ? &sh.a.v.sv[SH::RCAP - ns1] : &sh.a.v.sv[0];
                  for (int i0 = 0; i0 < ns; i0 += 64)
                    {
                      const int i = i0 + lane;
                      const uint2 me = i < ns ? svp[i] : make_uint2 (0u, 0u);
                      const uint64_t ck = ((uint64_t) (me.y & 31u) << 32) | me.x;
                      int rank = 0;
                      for (int c0 = 0; c0 < ns; c0 += 64)
                        {
                          const uint2 oth = c0 + lane < ns ? svp[c0 + lane] : make_uint2 (0u, 0u);
                          const int ny = min (64, ns - c0);
                          for (int y = 0; y < ny; y++)
                            {
                              const uint32_t ox = (uint32_t) __builtin_amdgcn_readlane ((int) oth.x, y), oy = (uint32_t) __builtin_amdgcn_readlane ((int) oth.y, y);
                              rank += (((((uint64_t) (oy & 31u)) << 32) | ox) < ck) ? 1 : 0;
                            }
                        }
                      if (i < ns)
                        sh.a.v.order[rank] = (uint16_t) i;
                    }
                  pm_wave_sync ();
                  // ---- the walk's state machine on the ranked anchors (pemapper.c:2251-2284)
                  bool more = true, done = false;
                  int cur_loop = -1;
                  for (int i0 = 0; i0 < ns && !done; i0 += 64)
                    {
                      const int i = i0 + lane;
                      const bool act = i < ns;
                      const int ri = act ? (int) sh.a.v.order[i] : 0;
                      const uint2 a = svp[ri];
                      const int tf = act ? (int) (a.y >> 8) : 0;
                      const int my_loop = act ? (int) (a.y & 31u) : 0;
                      const int my_off = (my_loop < cuts || cuts == 0) ? my_loop * idepth : last_off;
                      const uint32_t my_ml = (act ? a.x : 0u) - (uint32_t) (PM_DIAG_BIAS - my_off);        // the position itself
                      unsigned long long cnd = __ballot (act && tf >= min_match);
                      while (cnd)
                        {
                          const int l = __ffsll ((long long) cnd) - 1;
                          cnd &= cnd - 1;
                          const int tfl = __builtin_amdgcn_readlane (tf, l);
                          const int loop = __builtin_amdgcn_readlane (my_loop, l);
                          if (loop != cur_loop)
                            {
                              // the walk's loop bound is tested when a segment is entered, not inside it (pemapper.c:2216)
                              if (loop > 1 + cuts - min_match)
                                {
                                  done = true;
                                  break;
                                }
                              cur_loop = loop;
                            }
                          const int off_a = __builtin_amdgcn_readlane (my_off, l);
                          const uint32_t ml = (uint32_t) __builtin_amdgcn_readlane ((int) my_ml, l);
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
                              cnd &= __ballot (tf >= min_match);     // candidates below the new best would fall through both tests
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
        }
      PM_S3_T (10);
      e_out = e;
      tot_out = tot;
      big_out = big;
      pm_wave_sync ();
      PM_S3_T (11);
      eQ = eP;
      eP = eR;
      if (idxR == PM_S3_GRAB - 1)
        {
          const int raw_s = __builtin_amdgcn_readfirstlane ((int) raw_next);
          blkR = grid_n + raw_s;
          idxR = 0;
          // (pinned to a scalar register HERE: left alone the compiler defers the read of the atomic's result to eR's first use, which
          // comes right behind the next line requests -- and the in-order memory counter then waits for those as well)
          asm volatile ("; work counter read %0"::"s" (raw_s));
        }
      else
        idxR++;
      eR = blkR * PM_S3_GRAB + idxR;
      SQ = SP;
      lenQ = lenP;
      kQ = kP;
    }
  flush_out ();
#ifdef PEMAP_TIMING_PROBES
  if (lane == 0)
    for (int i = 0; i < 32; i++)
      atomicAdd (&pm_s3_probe[i], pacc[i]);
#endif
  if (lane == 0 && n_pos)
    atomicAdd (out.positions, n_pos);
}
