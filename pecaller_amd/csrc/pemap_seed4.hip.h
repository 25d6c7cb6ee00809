// pemap_seed4.hip.h -- the seed stage of a read-end in ONE kernel, fourth form: bucket look-ups against the 8 table replicas
// (fill_mers / get_mers, pemapper.c:1969-2003, 2158-2165; initial_map 1539-1690) and the diagonal vote (find_matches,
// pemapper.c:2189-2289), ONE WAVE per read-end, persistent, the positions never leaving LDS.
//
// What round 4 measured about the third form (pm_seed3_kernel): alone on the GPU its launch shrinks in
// proportion to its waves per CU (a wave waits in half of its cycles), but 19.6 KB of LDS admit 8 waves and 168 VGPRs leave a SIMD that
// hosts two of them room for ONE wave of the other stream's DP kernels (120-128 VGPRs each) -- the two streams traded time one for
// one.  This form is built for footprint first:
//   * ONE list for both strands (the strand is a bit of the position's tag): 1,536 positions for reads of up to 160 bases where the
//     third form held 2 x 1,024, and a read-end overflows when the two strands TOGETHER do -- a repeat usually fills one strand;
//   * the vote's tables overlay one another (the survivors' list lies in the bin table it was made from), 2-bit bases packed;
//     13.6 KB per wave where the third form took 19.6;
//   * at most 128 VGPRs (__launch_bounds__ (64, 4)) and no scratch: three seed waves and a DP wave fit a SIMD;
//   * the bin table is indexed by the diagonal bin itself (mod its size; strand 1 half a table further), not by a hash of it:
//     the three bins around a position are three neighbouring cells, read with one LDS instruction where the third form made three
//     hashes and three reads -- twice per position (candidates, positions next to candidates);
//   * table lines of a read-end are requested under a wave-uniform test (register r holds lines of segments 2r, 2r + 1 of the
//     2 S: wanted iff r < S), the per-segment counts of the `min_spots` rule travel in a register, not through LDS, round by round.
// The method is the third form's, restated:
//   look-ups  one (strand, segment) per round, lane j = neighbour j of fill_mers' order; a segment with a too-many bucket is
//             dropped whole (pemapper.c:1602-1606); buckets of one position go straight to the front of the list, entries that
//             point to a record are parked at its back and resolved from their records {count, p1, p2, p3 | p4 ...}: the first units
//             of the first 128 records requested together, the 4th and later positions of all of them as ONE flattened list;
//   vote      no sort.  A table of bins (16 diagonals wide) holds per bin the SET of segments with a position there; a position can
//             only be an anchor with tot_found >= min_match if the three bins around its diagonal hold min_match - 1 LATER
//             segments; the positions in or next to a bin with such a candidate are compacted (at most RCAP; more = a repeat, left
//             to the monolithic kernel) and the exact tot_found (pemapper.c:2241-2249) is an all-pairs test among them; the
//             surviving anchors are ranked (segment, position) per strand and the reference's walk (2251-2284) is replayed;
//   pipeline  three read-ends in flight per wave: the bytes of end k+2 and the table lines of end k+1 travel while end k is decoded
//             and voted on; ends are handed out through a counter, PM_S4_GRAB at a time.
// Output: raw hit lists (h.n_hits / spot / nn = segment offset / orient) for pm_emit_kernel; ends that do not fit go to the big-end list.
// LIST mode (second tier): the same kernel over a list of read-ends -- the big-end list of the first tier -- with a larger list.
#pragma once
#include <type_traits>
#include "pemap_wave.hip.h"

#ifndef PM_S4_KCAP
#define PM_S4_KCAP 1536         // positions of BOTH strands the list holds (reads of up to 160 bases, first tier)
#endif
#ifndef PM_S4_KCAP_LONG
#define PM_S4_KCAP_LONG 2048    // ... for reads of more than 160 bases (3,072: 28 KB of LDS and 5 waves per CU at 16 segments; 2,048: 22.9 KB and 7 -- 2 x 245 bases 44.4 -> 41.3 ms per step, a quarter more ends for the second tier)
#endif
#ifndef PM_S4_NH_LOG2_LONG
#define PM_S4_NH_LOG2_LONG 12   // log2 of the bins of the vote's table for those (11 for reads of up to 160 bases)
#endif
#ifndef PM_S4_WAVES_PER_EU
#define PM_S4_WAVES_PER_EU 4
#endif
#ifndef PM_S4_GRAB
#define PM_S4_GRAB 4            // consecutive read-ends a wave takes per fetch of the work counter (a single address returns ~85 M atomics/s)
#endif

// tag of a position: segment (5 bits) | strand << 5 | candidate anchor << 7
#define PM_S4_SEG(t) ((t) & 31u)
#define PM_S4_STRAND(t) (((t) >> 5) & 1u)

template < int SMAX, int TIER > struct __align__ (16) PmSeed4Shared
{
  static constexpr int NSEG = 2 * SMAX;
  // the list: positions of both strands.  Second tier (the first tier's big-end list): twice / four times that
  static constexpr int KCAP = TIER ? (SMAX <= 10 ? 4 * PM_S4_KCAP : 6144) : (SMAX <= 10 ? PM_S4_KCAP : PM_S4_KCAP_LONG);
  static constexpr bool CELL16 = SMAX <= 16;                                    // a bin's segment set: 16 bits while the segments fit
  static constexpr int NH_LOG2 = TIER ? (SMAX <= 10 ? 12 : CELL16 ? 13 : 12) : (SMAX <= 10 ? 11 : PM_S4_NH_LOG2_LONG);
  static constexpr int NH = 1 << NH_LOG2;
  static constexpr int RCAP = (SMAX <= 10 ? 160 : 256) * (TIER ? 2 : 1);
  static constexpr int CELL_WORDS = ((CELL16 ? NH / 2 : NH) + 2 + 3) & ~3;       // cell c is stored at index c + 1: a pad cell either side
  static constexpr int CAND_WORDS = (NH / 32 + 1 + 3) & ~3;
  static_assert (KCAP >= 2 * 49 * SMAX + 64, "the list holds an end's look-ups while they are decoded");
  union
  {
    uint32_t lines[SMAX * 128];         // ONE strand's SMAX x 8 lines of 16 entries (the other strand's wait in registers) ...
    struct                              // ... the vote's tables afterwards
    {
      union
      {
        uint32_t cell[CELL_WORDS];      // bit s of cell c: a position of segment s has its diagonal bin = c (mod NH; strand 1: + NH / 2)
        struct                          // once the anchors' counts are known the cells are dead: the surviving anchors lie here
        {
          uint2 sv[RCAP];               // x = key, y = segment | tot_found << 8 (strand 0 from the front, 1 from the back)
          uint16_t order[RCAP];
        } w;
      } c;
      uint32_t candbit[CAND_WORDS];     // bit c + 1: a candidate anchor's bin
      uint2 r[RCAP];                    // positions next to candidates: x = key, y = segment | strand << 5 | candidate << 6
    } v;
    struct                              // ... and in between, while the records are read: the work list of their tails
    {
      uint32_t src[128];                // per record with more than 3 positions: word in `multi` ...
      int32_t dst[128];                 // ... and list slot of its 4th position, both minus the record's place in the flattened tail
      uint8_t sg[128];                  // its tag (segment | strand << 5)
      uint8_t mark[KCAP];               // flattened tail: record number + 1 at the record's first element, 0 elsewhere
    } x;
  } a;
  // diagonal keys m + PM_DIAG_BIAS - offset(segment), from the front; while the entries are decoded the entries that point to a
  // record wait at the back (an end has at most 2 x 49 x S look-ups, each of them one or the other)
  uint32_t key[KCAP];
  uint8_t tag[KCAP];
  uint32_t hits[PM_MAX_HITS];
  uint16_t hits_off[PM_MAX_HITS];       // segment offset | strand << 15
  int seg_cnt[NSEG + 2];
  uint8_t seq[2][16 * SMAX + 8];        // 2-bit codes of the end whose k-mers are being formed, a byte each
};

static_assert (sizeof (PmSeed4Shared < 19, 1 >) <= 65536, "dynamic LDS of a launch without an attribute");
extern __shared__ __align__ (16) uint8_t pm_seed4_lds[];

#ifdef PEMAP_TIMING_PROBES
__device__ unsigned long long pm_s4_probe[32];      // 0..15 phases; 16..18 segments decoded / dropped / dropped by the k-mer's own bucket; 20 ends;
                                                    // 21..23 to the big-end list by the list's room / (unused) / the candidates' capacity
#define PM_S4_T(i) do { const unsigned long long t_ = __builtin_readcyclecounter (); pacc[i] += t_ - plast; plast = t_; } while (0)
#else
#define PM_S4_T(i) do { } while (0)
#endif

template < int SMAX, int TIER > __global__ __launch_bounds__ (64, TIER ? 2 : (SMAX <= 10 ? PM_S4_WAVES_PER_EU : 2))
void pm_seed4_kernel (PmIndex ix, PmBatch b, PmParams prm, PmHits h, PmLists out, const uint32_t * elist, const unsigned *n_elist, int prio)
{
  pm_set_prio (prio);
  typedef PmSeed4Shared < SMAX, TIER > SH;
  SH & sh = *reinterpret_cast < SH * >(pm_seed4_lds);
  const int lane = threadIdx.x;
  const int idepth = ix.idepth;
  const int max_off = max (2, idepth - 4);
  const uint32_t span = (uint32_t) (2 * (max_off - 1));
  const uint32_t multi_base = ix.multi_base;
  // first tier: ends [0, n_ends); second tier: the ends elist[0 .. *n_elist)
  const int n_work = TIER ? (int) __builtin_amdgcn_readfirstlane ((int) *n_elist) : b.n_ends;
  constexpr int NBT = (16 * SMAX + 63) / 64;    // registers of read bytes per lane (a read of S <= SMAX segments has at most 16 SMAX bases)
  const int nb = b.stride < 16 * SMAX ? b.stride : 16 * SMAX;
  unsigned long long n_pos = 0;
  // lane j looks at neighbour j of every segment (fill_mers' order, pm_neighbour): the 2-bit field it replaces, the
  // alternative's rank, the replica (= 4-bit field) whose line holds the entry
  const int nb_f = lane > 0 ? (lane - 1) / 3 : 0;
  const uint32_t nb_a = lane > 0 ? (uint32_t) ((lane - 1) % 3) : 0u;
  const uint32_t nb_sh = 2u * (uint32_t) (nb_f & 15);
  const uint32_t nb_keep = lane > 0 ? ~(3u << nb_sh) : 0xFFFFFFFFu;      // lane 0: the k-mer itself
  const uint32_t nb_alt_on = lane > 0 ? 0xFFFFFFFFu : 0u;
  const uint32_t nb_p4 = 4u * (uint32_t) ((nb_f >> 1) & 7), nb_pw = 16u * (uint32_t) ((nb_f >> 1) & 7);

  // ends are handed out through a counter, two values ahead of their use (the first grid-ful by block index)
  const int grid_n = __builtin_amdgcn_readfirstlane ((int) gridDim.x);
  int blkR = blockIdx.x, idxR = 0;
  auto next_end = [&] ()->int
  {
    const int e = blkR * PM_S4_GRAB + idxR;
    if (++idxR == PM_S4_GRAB)
      {
        idxR = 0;
        blkR = grid_n + (int) __builtin_amdgcn_readfirstlane ((int) (lane == 0 ? atomicAdd (out.next_end, 1u) : 0u));
      }
    return e;
  };
  int eQ = next_end ();
  int eP = next_end ();
  int eR = blkR * PM_S4_GRAB + idxR;

  uint8_t rb[NBT];              // the bytes of the end whose k-mers are formed next
  int rlen = 0, rend = 0;       // ... its length and its number (second tier: from the list)
  uint4 ln[SMAX];               // the table lines of the end decoded next: 4 lanes x 16 bytes per line, 16 lines per register
#pragma unroll
  for (int r = 0; r < SMAX; r++)
    ln[r] = make_uint4 (0u, 0u, 0u, 0u);
  auto load_bytes = [&] (int w)
  {
    const int e = TIER ? (int) elist[w] : w;
    rend = e;
    const uint8_t *src = pm_read_ptr (b, e, &rlen);
#pragma unroll
    for (int t = 0; t < NBT; t++)
      {
        const int i = lane + 64 * t;
        rb[t] = (i < nb) ? src[i] : (uint8_t) 0;
      }
  };
  // read in registers -> 2-bit codes of both strands, N filter (pemapper.c:1552-1559), segment count, the 2 x S k-mers into
  // kmer_out (lane = (strand, segment)), and the 2 x S x 8 line requests into ln[].  -> S, or 0 when the N filter drops the read.
  // (`between` runs after the read's bytes have been consumed and before the line requests are issued: the memory counter is in
  // order, so whatever is issued BEFORE the point that waits for the bytes is waited for as well -- the loop puts the previous end's
  // output and the work counter's atomic there)
  auto stage_p = [&] (uint32_t & kmer_out, int &len_out, int &end_out, auto between)->int
  {
    const int len = __builtin_amdgcn_readfirstlane (rlen);
    len_out = len;
    end_out = __builtin_amdgcn_readfirstlane (rend);
    int isn = 0;
#pragma unroll
    for (int t = 0; t < NBT; t++)
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
    // register r holds the lines of (strand, segment) 2 r (lanes 0..31) and 2 r + 1 (lanes 32..63), 8 each: wanted iff r < S
    const int p = (lane >> 2) & 7;
    const uint32_t *rep_lane = ix.rep + ((size_t) p << 32) + (size_t) ((lane & 3) * 4);
#pragma unroll
    for (int r = 0; r < SMAX; r++)
      if (r < S)
        {
          const uint32_t k_lo = (uint32_t) __builtin_amdgcn_readlane ((int) kmer_out, (2 * r) & 63), k_hi = (uint32_t) __builtin_amdgcn_readlane ((int) kmer_out, (2 * r + 1) & 63);
          const uint32_t ksrc = lane < 32 ? k_lo : k_hi;
          const uint32_t idx = pm_swap_fields (ksrc, p);
          ln[r] = *(const uint4 *) (rep_lane + (size_t) (idx & ~15u));
        }
      else
        ln[r] = make_uint4 (0u, 0u, 0u, 0u);
    return S;
  };

  int SQ = 0, lenQ = 0, endQ = 0;
  uint32_t kQ = 0;              // lane sg: the k-mer of (strand, segment) sg of the end being decoded
  if (eQ < n_work)
    {
      load_bytes (eQ);
      SQ = stage_p (kQ, lenQ, endQ, [] () { });
    }
  if (eP < n_work)
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
        if (lane == 0)
          h.n_hits[e_out] = tot_out;
        for (int t = lane; t < tot_out; t += 64)
          {
            // (a 32-bit index -- task ids are end * 200 + hit in 32 bits everywhere -- added to the arrays' scalar bases: with a 64-bit
            // one the compiler kept per-lane base addresses alive around the whole loop, in scratch)
            const uint32_t o = (uint32_t) e_out * (uint32_t) PM_MAX_HITS + (uint32_t) t;
            const uint32_t m = sh.hits_off[t];
            h.spot[o] = sh.hits[t];
            h.nn[o] = (int16_t) (m & 0x7FFFu);
            h.orient[o] = (uint8_t) (m >> 15);
          }
      }
    e_out = -1;
  };
#ifdef PEMAP_TIMING_PROBES
  unsigned long long pacc[32];
  for (int i_ = 0; i_ < 32; i_++)
    pacc[i_] = 0ull;
  unsigned long long plast = __builtin_readcyclecounter ();
#endif
  while (eQ < n_work)
    {
      // (loop-carried and wave-uniform: said so, or the compiler keeps them, and every branch on them, in vector registers)
      const int e = __builtin_amdgcn_readfirstlane (endQ), S = __builtin_amdgcn_readfirstlane (SQ), len = __builtin_amdgcn_readfirstlane (lenQ);
      const int cuts = S - 1;
      const int last_off = len - idepth;
      int tot = 0;
      bool big = false;
      int T = 0, cmin0 = 0, cmin1 = 0;
      if (S > 0)
        {
          // ---- A: the lines of this end (requested one iteration ago) go from registers to LDS one strand at a time
          auto lines_to_lds = [&] (int strand)
          {
            const int sgA = strand * S, sgB = sgA + S;
#pragma unroll
            for (int r = 0; r < SMAX; r++)
              if (2 * r + 1 >= sgA && 2 * r < sgB)      // (uniform)
                {
                  const int sg = 2 * r + (lane >> 5);
                  if (sg >= sgA && sg < sgB)
                    *(uint4 *) (&sh.a.lines[(sg - sgA) * 128 + (lane & 31) * 4]) = ln[r];
                }
          };
          lines_to_lds (0);
          pm_wave_sync ();
          PM_S4_T (0);
          // ---- B: one (strand, segment) per round, lane j = neighbour j.  A segment with a bucket of too_many_spots or more is
          //      dropped whole (pemapper.c:1602-1606: the entry itself says so); buckets of one position go straight to the
          //      front of the list; entries that point to a record are parked at its back
          int nf = 0, nm = 0;
          int cntv = 0;         // lane sg: positions of (strand, segment) sg found in single-position buckets
          auto decode_strand = [&] (auto ST)
          {
            constexpr int strand = decltype (ST)::value;
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
                // (one compare each: an entry is the bucket's only position, or -- 0xFFFFFFFE being too many, seen above, and 0xFFFFFFFF
                // empty -- a record's unit number above multi_base; the two kinds share one pair of stores, front and back of the list)
                const uint32_t unit = ent - multi_base;
                const bool single = ent < multi_base, multi = unit < 0xFFFFFFFEu - multi_base;
                const unsigned long long bs = __ballot (single), bm = __ballot (multi);
                const int off = (seg < cuts || cuts == 0) ? seg * idepth : last_off;
                const uint32_t tg = (uint32_t) (seg | (strand << 5));
                // (both ranks and both values computed by every lane, then selected: no divergent arms)
                const int rs = pm_lanes_below_here (bs), rm = pm_lanes_below_here (bm);
                int at = nf + rs;
                at = multi ? SH::KCAP - 1 - nm - rm : at;
                uint32_t val = ent + (uint32_t) (PM_DIAG_BIAS - off);
                val = multi ? unit : val;                       // (a record: its first 16-byte unit)
                if (single || multi)
                  {
                    sh.key[at] = val;
                    sh.tag[at] = (uint8_t) tg;
                  }
                const int ns = (int) __popcll (bs);
                nf += ns;
                nm += (int) __popcll (bm);
                cntv = lane == sg ? ns : cntv;
              }
          };
          decode_strand (std::integral_constant < int, 0 > { });
          pm_wave_sync ();
          PM_S4_T (1);
          lines_to_lds (1);
          pm_wave_sync ();
          PM_S4_T (2);
          decode_strand (std::integral_constant < int, 1 > { });
          if (lane < SH::NSEG + 2)
            sh.seg_cnt[lane] = cntv;
          pm_wave_sync ();
          PM_S4_T (3);
          // ---- C: the records: {count, positions...} in 16-byte units; the first unit answers for buckets of up to 3 positions.
          //      The first 128 records are requested together (one exposed HBM latency); what an end has beyond them follows in
          //      rounds of 128.  The positions go to the front of the list whose back still holds the records not yet read: an end
          //      whose list would reach them is left to the next tier.
          // one round: lane = record (count in hdr.x, its first three positions behind it), `limit` = where the front must stop
          auto rec_round = [&] (const uint4 & hdr, const uint32_t tg, const bool valid, const int limit, uint32_t & c_out, int &dst_out, bool & wr_out)
          {
            const uint32_t c = valid ? hdr.x : 0u;
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
            const int seg = (int) PM_S4_SEG (tg);
            const int off = (seg < cuts || cuts == 0) ? seg * idepth : last_off;
            const uint32_t bias = (uint32_t) (PM_DIAG_BIAS - off);
            if (wr)
              {
                atomicAdd (&sh.seg_cnt[(int) PM_S4_STRAND (tg) * S + seg], (int) c);
                uint32_t *kk = &sh.key[dst];
                uint8_t *tt = &sh.tag[dst];
                kk[0] = hdr.y + bias;
                tt[0] = (uint8_t) tg;
                kk[1] = hdr.z + bias;
                tt[1] = (uint8_t) tg;
                if (c > 2)
                  {
                    kk[2] = hdr.w + bias;
                    tt[2] = (uint8_t) tg;
                  }
              }
            c_out = c;
            dst_out = dst;
            wr_out = wr;
          };
          // the 4th and later positions of the records of one or two rounds as ONE flattened list, lane = element, so that their loads
          // are in flight together whatever records they belong to: the record of element t is the largest record number marked at or
          // before t (a prefix maximum), its source word and list slot are affine in t
          auto tails = [&] (const uint32_t uA, const uint32_t tgA, const uint32_t cA, const int dA, const bool wA, const uint32_t uB, const uint32_t tgB,
                            const uint32_t cB, const int dB, const bool wB)
          {
            const uint32_t restA = (wA && cA > 3u) ? cA - 3u : 0u, restB = (wB && cB > 3u) ? cB - 3u : 0u;
            const uint32_t inA = pm_wave_incl_sum (restA), inB = pm_wave_incl_sum (restB);
            const int totA = __builtin_amdgcn_readlane ((int) inA, 63), M = totA + __builtin_amdgcn_readlane ((int) inB, 63);
            if (M <= 0)
              return;
            for (int i = lane; i < (M + 15) / 16; i += 64)
              ((uint4 *) sh.a.x.mark)[i] = make_uint4 (0u, 0u, 0u, 0u);
            pm_wave_sync ();
            if (restA)
              {
                const int ro = (int) (inA - restA);
                sh.a.x.mark[ro] = (uint8_t) (lane + 1);
                sh.a.x.src[lane] = uA * 4u + 4u - (uint32_t) ro;
                sh.a.x.dst[lane] = dA + 3 - ro;
                sh.a.x.sg[lane] = (uint8_t) tgA;
              }
            if (restB)
              {
                const int ro = totA + (int) (inB - restB);
                sh.a.x.mark[ro] = (uint8_t) (lane + 65);
                sh.a.x.src[64 + lane] = uB * 4u + 4u - (uint32_t) ro;
                sh.a.x.dst[64 + lane] = dB + 3 - ro;
                sh.a.x.sg[64 + lane] = (uint8_t) tgB;
              }
            pm_wave_sync ();
            int carry = 0;
#pragma unroll 1
            for (int t0 = 0; t0 < M; t0 += 128)
              {
                uint32_t val[2];
                int own[2];
#pragma unroll
                for (int j = 0; j < 2; j++)
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
                for (int j = 0; j < 2; j++)
                  {
                    const int t = t0 + 64 * j + lane;
                    if (t < M)
                      {
                        const int o = own[j];
                        const int d = sh.a.x.dst[o] + t;
                        const uint32_t tg = (uint32_t) sh.a.x.sg[o];
                        const int seg = (int) PM_S4_SEG (tg);
                        const int off = (seg < cuts || cuts == 0) ? seg * idepth : last_off;
                        sh.key[d] = val[j] + (uint32_t) (PM_DIAG_BIAS - off);
                        sh.tag[d] = (uint8_t) tg;
                      }
                  }
              }
          };
#pragma unroll 1
          for (int i0 = 0; i0 < nm && !big; i0 += 128)
            {
              uint4 hA = make_uint4 (0u, 0u, 0u, 0u), hB = make_uint4 (0u, 0u, 0u, 0u);
              uint32_t uA = 0, uB = 0, tgA = 0, tgB = 0;
              const bool vA = i0 + lane < nm, vB = i0 + 64 + lane < nm;
              if (vA)
                {
                  uA = sh.key[SH::KCAP - 1 - (i0 + lane)];
                  tgA = sh.tag[SH::KCAP - 1 - (i0 + lane)];
                  hA = *(const uint4 *) (ix.multi + (size_t) uA * 4);
                }
              if (vB)
                {
                  uB = sh.key[SH::KCAP - 1 - (i0 + 64 + lane)];
                  tgB = sh.tag[SH::KCAP - 1 - (i0 + 64 + lane)];
                  hB = *(const uint4 *) (ix.multi + (size_t) uB * 4);
                }
#ifdef PEMAP_TIMING_PROBES
              { const uint32_t w_ = hA.x + hB.x; asm volatile ("" :: "v" (w_)); }      // (the loads have landed)
              PM_S4_T (12);
#endif
              // (the records of this round are in registers: their slots at the back are free; those of later rounds are not)
              const int limit = (i0 + 128 < nm) ? SH::KCAP - nm : SH::KCAP;
              uint32_t cA = 0, cB = 0;
              int dA = 0, dB = 0;
              bool wA = false, wB = false;
              rec_round (hA, tgA, vA, limit, cA, dA, wA);
              if (i0 + 64 < nm && !big)
                rec_round (hB, tgB, vB, limit, cB, dB, wB);
              PM_S4_T (13);
              if (!big)
                tails (uA, tgA, cA, dA, wA, uB, tgB, cB, dB, wB);
              PM_S4_T (15);
            }
          pm_wave_sync ();
          PM_S4_T (4);
          T = nf;
#ifdef PEMAP_TIMING_PROBES
          pacc[20] += 1ull;
#endif
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
        if (lane == 0 && idxR == PM_S4_GRAB - 1)
          raw_next = atomicInc (out.next_end, 0xFFFFFFFFu);
      };
      // ---- P: the next end's k-mers and line requests (its bytes arrived during the previous iteration); R: the bytes of the end after
      int SP = 0, lenP = 0, endP = 0;
      uint32_t kP = 0;
      // Every load issued so far has landed or is about to be needed (the next end's bytes were requested an iteration ago, behind
      // this end's lines; the records were consumed above): said HERE, on every path, so that the compiler's wait-count model is
      // clean before the line requests go out.  Without it the zeroing of rb[] in load_bytes -- registers a load of the previous
      // iteration MAY still own on the path that skipped stage_p -- carries `s_waitcnt vmcnt(1)` right behind the line requests
      // (conditional, so uncounted), and the wave sat out their whole HBM latency: 31 % of its cycles (the third form had the same
      // wait; the lines were never in flight across the vote as its comments said)
      __builtin_amdgcn_s_waitcnt (0x0F70);      // vmcnt(0)
      if (eP < n_work)
        SP = stage_p (kP, lenP, endP, out_and_next);
      else
        out_and_next ();
      if (eR < n_work)
        load_bytes (eR);
      PM_S4_T (5);
      // ---- V: find_matches (pemapper.c:2189-2289) on the list in LDS
      if (S > 0 && !big)
        {
          n_pos += (unsigned long long) T;
          int min_match = max (1, cuts);        // pemapper.c:1642-1645
          if (cuts > 4)
            min_match = (4 * cuts) / 5;
          min_match = min (min_match, 4);
          const int mm0 = min_match;
          const int loop_max0 = 1 + cuts - mm0;
          const bool use0 = cmin0 <= PM_MAX_HITS, use1 = cmin1 <= PM_MAX_HITS;
          const uint32_t use_bits = (use0 ? 1u : 0u) | (use1 ? 2u : 0u);
          // cell c (stored at index c + 1) of a key: its diagonal bin mod NH, strand 1 half a table further
          auto cell_of = [&] (uint32_t key, uint32_t tg)->uint32_t
          {
            return ((key >> 4) + (PM_S4_STRAND (tg) << (SH::NH_LOG2 - 1))) & (uint32_t) (SH::NH - 1);
          };
          // which segments have a position in each diagonal bin (16 diagonals wide), both strands
#pragma unroll
          for (int i = 0; i < SH::CELL_WORDS / 4; i += 64)
            if (i + lane < SH::CELL_WORDS / 4)
              ((uint4 *) sh.a.v.c.cell)[i + lane] = make_uint4 (0u, 0u, 0u, 0u);
          if (lane < SH::CAND_WORDS / 4)
            ((uint4 *) sh.a.v.candbit)[lane] = make_uint4 (0u, 0u, 0u, 0u);
          pm_wave_sync ();
          // (each pass takes the positions two rounds at a time: the LDS reads of a batch are issued together)
          for (int i0 = 0; i0 < T; i0 += 128)
            {
              uint32_t kk[2], tg[2];
#pragma unroll
              for (int j = 0; j < 2; j++)
                {
                  const int i = i0 + 64 * j + lane;
                  kk[j] = i < T ? sh.key[i] : 0u;
                  tg[j] = i < T ? (uint32_t) sh.tag[i] : 0u;
                }
#pragma unroll
              for (int j = 0; j < 2; j++)
                if (i0 + 64 * j + lane < T && ((use_bits >> PM_S4_STRAND (tg[j])) & 1u))
                  {
                    const uint32_t c1 = cell_of (kk[j], tg[j]) + 1u;
                    if constexpr (SH::CELL16)
                      atomicOr (&sh.a.v.c.cell[c1 >> 1], (1u << PM_S4_SEG (tg[j])) << (16u * (c1 & 1u)));
                    else
                      atomicOr (&sh.a.v.c.cell[c1], 1u << PM_S4_SEG (tg[j]));
                  }
            }
          pm_wave_sync ();
          // the pad cells either side mirror the table's other end (the bins are taken mod NH)
          if (lane == 0)
            {
              if constexpr (SH::CELL16)
                {
                  const uint32_t a0 = sh.a.v.c.cell[0], b0 = sh.a.v.c.cell[SH::NH / 2];
                  const uint32_t m = (a0 & 0xFFFF0000u) | (b0 & 0xFFFFu);
                  sh.a.v.c.cell[0] = m;
                  sh.a.v.c.cell[SH::NH / 2] = m;
                }
              else
                {
                  sh.a.v.c.cell[0] = sh.a.v.c.cell[SH::NH];
                  sh.a.v.c.cell[SH::NH + 1] = sh.a.v.c.cell[1];
                }
            }
          pm_wave_sync ();
          PM_S4_T (6);
          // the segments with a position in the three bins around cell c
          auto around = [&] (uint32_t c)->uint32_t
          {
            if constexpr (SH::CELL16)
              {
                const uint32_t d0 = sh.a.v.c.cell[c >> 1], d1 = sh.a.v.c.cell[(c >> 1) + 1];
                return ((d0 >> 16) | d1 | ((c & 1u) ? (d1 >> 16) : d0)) & 0xFFFFu;
              }
            return sh.a.v.c.cell[c] | sh.a.v.c.cell[c + 1] | sh.a.v.c.cell[c + 2];
          };
          unsigned long long any_cand = 0ull;
          // candidate anchors: positions of a segment the walk can reach whose three bins hold at least min_match - 1 LATER segments
          // (everything within max_off - 1 <= 15 diagonals of an anchor lies in those bins; colliding bins only add candidates)
          for (int i0 = 0; i0 < T; i0 += 128)
            {
              uint32_t kk[2], tg[2], mk[2];
#pragma unroll
              for (int j = 0; j < 2; j++)
                {
                  const int i = i0 + 64 * j + lane;
                  kk[j] = i < T ? sh.key[i] : 0u;
                  tg[j] = i < T ? (uint32_t) sh.tag[i] : 0u;
                }
#pragma unroll
              for (int j = 0; j < 2; j++)
                mk[j] = around (cell_of (kk[j], tg[j]));
#pragma unroll
              for (int j = 0; j < 2; j++)
                {
                  const int i = i0 + 64 * j + lane;
                  const int sa = (int) PM_S4_SEG (tg[j]);
                  const bool is_cand = i < T && ((use_bits >> PM_S4_STRAND (tg[j])) & 1u) && sa <= loop_max0 && 1 + __popc (mk[j] & ~((2u << sa) - 1u)) >= mm0;
                  any_cand |= __ballot (is_cand);
                  if (is_cand)
                    {
                      const uint32_t c1 = cell_of (kk[j], tg[j]) + 1u;
                      sh.tag[i] = (uint8_t) (tg[j] | 0x80u);
                      atomicOr (&sh.a.v.candbit[c1 >> 5], 1u << (c1 & 31u));
                    }
                }
            }
          pm_wave_sync ();
          if (lane == 0 && any_cand != 0ull)
            {
              const uint32_t a0 = sh.a.v.candbit[0], b0 = sh.a.v.candbit[SH::NH / 32];
              sh.a.v.candbit[0] = a0 | (b0 & 1u);
              sh.a.v.candbit[SH::NH / 32] = b0 | (a0 & 2u);
            }
          pm_wave_sync ();
          PM_S4_T (7);
          // the positions next to a candidate (same or adjacent bin), compacted
          int nR = 0;
          if (any_cand != 0ull)         // (most ends that do not map have no candidate at all)
            for (int i0 = 0; i0 < T; i0 += 128)
              {
                uint32_t kk[2], tg[2], cb[2];
#pragma unroll
                for (int j = 0; j < 2; j++)
                  {
                    const int i = i0 + 64 * j + lane;
                    kk[j] = i < T ? sh.key[i] : 0u;
                    tg[j] = i < T ? (uint32_t) sh.tag[i] : 0u;
                  }
#pragma unroll
                for (int j = 0; j < 2; j++)
                  {
                    const uint32_t c = cell_of (kk[j], tg[j]);
                    const uint32_t d0 = sh.a.v.candbit[c >> 5], d1 = sh.a.v.candbit[(c >> 5) + 1];
                    cb[j] = __builtin_amdgcn_alignbit (d1, d0, c & 31u) & 7u;
                  }
#pragma unroll
                for (int j = 0; j < 2; j++)
                  if (i0 + 64 * j < T)
                    {
                      const bool rel = i0 + 64 * j + lane < T && cb[j] != 0u && ((use_bits >> PM_S4_STRAND (tg[j])) & 1u);
                      const unsigned long long br = __ballot (rel);
                      if (rel)
                        {
                          const int at = nR + pm_lanes_below (br);
                          if (at < SH::RCAP)
                            sh.a.v.r[at] = make_uint2 (kk[j], (tg[j] & 63u) | ((tg[j] & 0x80u) >> 1));
                        }
                      nR += (int) __popcll (br);
                    }
              }
          pm_wave_sync ();
          PM_S4_T (8);
          if (nR > SH::RCAP)
            {
#ifdef PEMAP_TIMING_PROBES
              pacc[23] += 1ull;
#endif
              big = true;       // a repeat: left to the next tier
            }
          else
            {
              // tot_found of every candidate (pemapper.c:2241-2249): 1 + the later segments with a position within max_off of its
              // diagonal; the anchors that reach min_match, per strand, compacted: sv[] (strand 0 from the front, strand 1 from the back)
              // (sv lies in the cells, dead since the candidates were found; r[] is apart)
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
                    sh.a.v.c.w.sv[ns0 + pm_lanes_below (b0)] = make_uint2 (me.x, (me.y & 31u) | ((uint32_t) tf << 8));
                  if (s1)
                    sh.a.v.c.w.sv[SH::RCAP - 1 - (ns1 + pm_lanes_below (b1))] = make_uint2 (me.x, (me.y & 31u) | ((uint32_t) tf << 8));
                  ns0 += (int) __popcll (b0);
                  ns1 += (int) __popcll (b1);
                }
              pm_wave_sync ();
              PM_S4_T (9);
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
                  const uint2 *svp = strand ? &sh.a.v.c.w.sv[SH::RCAP - ns1] : &sh.a.v.c.w.sv[0];
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
                        sh.a.v.c.w.order[rank] = (uint16_t) i;
                    }
                  pm_wave_sync ();
                  // ---- the walk's state machine on the ranked anchors (pemapper.c:2251-2284)
                  bool more = true, done = false;
                  int cur_loop = -1;
                  for (int i0 = 0; i0 < ns && !done; i0 += 64)
                    {
                      const int i = i0 + lane;
                      const bool act = i < ns;
                      const int ri = act ? (int) sh.a.v.c.w.order[i] : 0;
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
                                  sh.hits_off[0] = (uint16_t) (off_a | (strand << 15));
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
                                    if (sh.hits[k] - (uint32_t) (sh.hits_off[k] & 0x7FFFu) == diag)
                                      dup = true;
                                  if (!__any (dup))
                                    {
                                      if (lane == 0)
                                        {
                                          sh.hits[tot] = ml;
                                          sh.hits_off[tot] = (uint16_t) (off_a | (strand << 15));
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
      PM_S4_T (10);
      e_out = e;
      tot_out = tot;
      big_out = big;
      pm_wave_sync ();
      PM_S4_T (11);
      eQ = eP;
      eP = eR;
      if (idxR == PM_S4_GRAB - 1)
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
      eR = blkR * PM_S4_GRAB + idxR;
      SQ = SP;
      lenQ = lenP;
      endQ = endP;
      kQ = kP;
    }
  flush_out ();
#ifdef PEMAP_TIMING_PROBES
  if (lane == 0)
    for (int i = 0; i < 32; i++)
      atomicAdd (&pm_s4_probe[i], pacc[i]);
#endif
  if (lane == 0 && n_pos)
    atomicAdd (out.positions, n_pos);
}
