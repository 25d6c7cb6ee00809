// pemap_aux.hip.h -- device code around the hot path: index construction (the arrays index_genome_whole.c writes),
// pileup export, and the synthetic workload generator used by bench.py and the scale tests.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// ------------------------------------------------------------------------------------------------------------
// Index build.  index_genome_whole.c:206-299: a rolling 32-bit 2-bit-per-base register over each contig; 'N' resets
// the run; a position is indexed once 16 non-N letters have been seen; letters other than C/G/T code as 0 (169-177,
// 264); the stored position is gpos + newpos with newpos starting at 1 - 16 per contig, i.e. the k-mer's first base in
// "len-15" compressed coordinates (215, 271).  .mdx = positions grouped by k-mer ascending, genome order inside a
// k-mer (334-341); .idx = exclusive prefix sums of the bucket sizes over all 2^32 k-mers plus the total (337, 342).
// ------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t ix_code (uint8_t c, int bis)
{
  // bit_mat, index_genome_whole.c:169-177: only upper-case letters reach it (toupper at 252)
  if (c == 'G')
    return 2;
  if (c == 'T')
    return 3;
  if (c == 'C')
    return bis ? 3 : 1;
  return 0;
}

__device__ __forceinline__ int ix_contig_of (const uint64_t * real_starts, int n_contigs, uint64_t p)
{
  int lo = 0, hi = n_contigs;   // largest c with real_starts[c] <= p
  while (hi - lo > 1)
    {
      int mid = (lo + hi) >> 1;
      if (real_starts[mid] <= p)
        lo = mid;
      else
        hi = mid;
    }
  return lo;
}

// valid k-mer starting at real position p ?  returns key and compressed position
__device__ __forceinline__ bool ix_kmer_at (const uint8_t * genome, const uint64_t * real_starts, int n_contigs, uint64_t gsize,
                                            uint64_t p, int bis, uint32_t * key, uint32_t * val)
{
  if (p + 16 > gsize)
    return false;
  int c = ix_contig_of (real_starts, n_contigs, p);
  if (p + 16 > real_starts[c + 1])
    return false;
  uint32_t k = 0;
  bool ok = true;
  for (int i = 0; i < 16; i++)
    {
      uint8_t ch = genome[p + i];
      if (ch == 'N')
        ok = false;
      k = (k << 2) + ix_code (ch, bis);
    }
  *key = k;
  *val = (uint32_t) (p - 15ull * (uint64_t) c);
  return ok;
}

#define IX_BLOCK 256
#define IX_PER_BLOCK 4096

// pass 1: number of indexed positions per 4096-position tile
__global__ __launch_bounds__ (IX_BLOCK) void ix_count_kernel (const uint8_t * genome, const uint64_t * real_starts, int n_contigs,
                                                              uint64_t gsize, int bis, uint32_t * tile_count)
{
  __shared__ unsigned s_cnt;
  if (threadIdx.x == 0)
    s_cnt = 0;
  __syncthreads ();
  uint64_t base = (uint64_t) blockIdx.x * IX_PER_BLOCK;
  unsigned mine = 0;
  for (int r = 0; r < IX_PER_BLOCK / IX_BLOCK; r++)
    {
      uint64_t p = base + (uint64_t) r * IX_BLOCK + threadIdx.x;
      uint32_t k, v;
      if (p < gsize && ix_kmer_at (genome, real_starts, n_contigs, gsize, p, bis, &k, &v))
        mine++;
    }
  atomicAdd (&s_cnt, mine);
  __syncthreads ();
  if (threadIdx.x == 0)
    tile_count[blockIdx.x] = s_cnt;
}

// pass 2: write (key, pos) in genome order at the tile's offset
__global__ __launch_bounds__ (IX_BLOCK) void ix_emit_kernel (const uint8_t * genome, const uint64_t * real_starts, int n_contigs,
                                                             uint64_t gsize, int bis, const uint64_t * tile_offset,
                                                             uint32_t * keys, uint32_t * vals)
{
  __shared__ unsigned s_wave[IX_BLOCK / 64];
  __shared__ unsigned s_run;
  if (threadIdx.x == 0)
    s_run = 0;
  __syncthreads ();
  const uint64_t base = (uint64_t) blockIdx.x * IX_PER_BLOCK;
  const uint64_t out0 = tile_offset[blockIdx.x];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  for (int r = 0; r < IX_PER_BLOCK / IX_BLOCK; r++)
    {
      uint64_t p = base + (uint64_t) r * IX_BLOCK + threadIdx.x;
      uint32_t k = 0, v = 0;
      bool ok = p < gsize && ix_kmer_at (genome, real_starts, n_contigs, gsize, p, bis, &k, &v);
      unsigned long long bal = __ballot (ok);
      unsigned before = __popcll (bal & ((1ull << lane) - 1ull));
      if (lane == 0)
        s_wave[wv] = __popcll (bal);
      __syncthreads ();
      unsigned woff = 0, tot = 0;
      for (int w = 0; w < IX_BLOCK / 64; w++)
        {
          if (w < wv)
            woff += s_wave[w];
          tot += s_wave[w];
        }
      unsigned run = s_run;
      if (ok)
        {
          uint64_t at = out0 + run + woff + before;
          keys[at] = k;
          vals[at] = v;
        }
      __syncthreads ();
      if (threadIdx.x == 0)
        s_run = run + tot;
      __syncthreads ();
    }
}

// single-block exclusive scan of u32 tile counts into u64 offsets (n up to a few million: trivial next to the sort)
__global__ __launch_bounds__ (1024) void ix_scan_tiles_kernel (const uint32_t * cnt, uint64_t * off, uint64_t n, uint64_t * total)
{
  __shared__ uint64_t s_part[1024];
  const uint64_t per = (n + 1023) / 1024;
  const uint64_t lo = threadIdx.x * per, hi = (lo + per < n) ? lo + per : n;
  uint64_t sum = 0;
  for (uint64_t i = lo; i < hi; i++)
    sum += cnt[i];
  s_part[threadIdx.x] = sum;
  __syncthreads ();
  if (threadIdx.x == 0)
    {
      uint64_t acc = 0;
      for (int i = 0; i < 1024; i++)
        {
          uint64_t t = s_part[i];
          s_part[i] = acc;
          acc += t;
        }
      *total = acc;
    }
  __syncthreads ();
  uint64_t acc = s_part[threadIdx.x];
  for (uint64_t i = lo; i < hi; i++)
    {
      off[i] = acc;
      acc += cnt[i];
    }
}

// after the stable sort by key: the last entry e of each run of equal keys writes e+1 to pos_index[key+1]
// (the inclusive end of bucket `key`); a running maximum over the 2^32+1 table then yields the prefix table.
__global__ void ix_run_ends_kernel (const uint32_t * keys, uint64_t n, uint32_t * pos_index)
{
  uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n)
    return;
  uint32_t k = keys[i];
  if (i + 1 == n || keys[i + 1] != k)
    pos_index[(uint64_t) k + 1ull] = (uint32_t) (i + 1);
}

#define SC_BLOCK 256
#define SC_ITEMS 32
#define SC_TILE (SC_BLOCK * SC_ITEMS)

__global__ __launch_bounds__ (SC_BLOCK) void ix_maxscan_reduce_kernel (const uint32_t * a, uint64_t n, uint32_t * tile_max)
{
  __shared__ uint32_t s[SC_BLOCK];
  uint64_t base = (uint64_t) blockIdx.x * SC_TILE;
  uint32_t m = 0;
  for (int r = 0; r < SC_ITEMS; r++)
    {
      uint64_t i = base + (uint64_t) r * SC_BLOCK + threadIdx.x;
      if (i < n)
        m = max (m, a[i]);
    }
  s[threadIdx.x] = m;
  __syncthreads ();
  for (int o = SC_BLOCK / 2; o > 0; o >>= 1)
    {
      if (threadIdx.x < o)
        s[threadIdx.x] = max (s[threadIdx.x], s[threadIdx.x + o]);
      __syncthreads ();
    }
  if (threadIdx.x == 0)
    tile_max[blockIdx.x] = s[0];
}

// exclusive running max over the tile maxima, single block
__global__ __launch_bounds__ (1024) void ix_maxscan_tiles_kernel (uint32_t * tile_max, uint64_t n)
{
  __shared__ uint32_t s_part[1024];
  const uint64_t per = (n + 1023) / 1024;
  const uint64_t lo = threadIdx.x * per, hi = (lo + per < n) ? lo + per : n;
  uint32_t m = 0;
  for (uint64_t i = lo; i < hi; i++)
    m = max (m, tile_max[i]);
  s_part[threadIdx.x] = m;
  __syncthreads ();
  if (threadIdx.x == 0)
    {
      uint32_t acc = 0;
      for (int i = 0; i < 1024; i++)
        {
          uint32_t t = s_part[i];
          s_part[i] = acc;
          acc = max (acc, t);
        }
    }
  __syncthreads ();
  uint32_t acc = s_part[threadIdx.x];
  for (uint64_t i = lo; i < hi; i++)
    {
      uint32_t t = tile_max[i];
      tile_max[i] = acc;
      acc = max (acc, t);
    }
}

// in-place inclusive running max inside each tile, seeded with the tile's carry-in
__global__ __launch_bounds__ (SC_BLOCK) void ix_maxscan_apply_kernel (uint32_t * a, uint64_t n, const uint32_t * tile_carry)
{
  __shared__ uint32_t s[SC_BLOCK];
  // thread t owns SC_ITEMS consecutive elements so that the scan order is the array order
  uint64_t base = (uint64_t) blockIdx.x * SC_TILE + (uint64_t) threadIdx.x * SC_ITEMS;
  uint32_t v[SC_ITEMS];
  uint32_t m = 0;
#pragma unroll
  for (int r = 0; r < SC_ITEMS; r++)
    {
      uint64_t i = base + r;
      v[r] = (i < n) ? a[i] : 0u;
      m = max (m, v[r]);
      v[r] = m;
    }
  s[threadIdx.x] = m;
  __syncthreads ();
  // exclusive running max over threads (Hillis-Steele on 256 values)
  uint32_t mine = m;
  for (int o = 1; o < SC_BLOCK; o <<= 1)
    {
      uint32_t t = (threadIdx.x >= (unsigned) o) ? s[threadIdx.x - o] : 0u;
      __syncthreads ();
      mine = max (mine, t);
      s[threadIdx.x] = mine;
      __syncthreads ();
    }
  uint32_t carry = tile_carry[blockIdx.x];
  if (threadIdx.x > 0)
    carry = max (carry, s[threadIdx.x - 1]);
#pragma unroll
  for (int r = 0; r < SC_ITEMS; r++)
    {
      uint64_t i = base + r;
      if (i < n)
        a[i] = max (v[r], carry);
    }
}

// ------------------------------------------------------------------------------------------------------------
// Look-up replicas.  The reference's table costs one 64-byte fabric request per 8-byte look-up, and a read-end makes
// 2 x S x 49 of them: MI355X serves ~48 G random requests per second whatever their size up to 128 bytes
// (tools/micro/line_gather.hip), so the look-up phase is bound by the NUMBER of lines it touches.  288 GB of HBM buy
// that number down: the 2^32-entry table is kept 8 times, replica p ordered by the k-mer with its 4-bit fields 0 and p
// swapped, so that the 16 k-mers that differ only in bases 2p, 2p+1 (counted from the read's last base) share one
// 64-byte line of replica p.  A k-mer and its 48 single-substitution neighbours (fill_mers, pemapper.c:1969-2003) then
// sit in 8 lines, one per replica, instead of 43.  An entry is self-contained:
//     0xFFFFFFFF          empty bucket
//     0xFFFFFFFE          bucket of too_many_spots (100) or more positions: the segment is dropped (pemapper.c:1602-1606)
//     v < multi_base      the bucket's only position
//     otherwise           2..99 positions: record {count, positions...} at multi + 4 * (v - multi_base) words (16-byte units)
// so the common buckets (empty, one position) need no second request.  Bucket sizes follow get_mers (pemapper.c:2158-2165)
// including its 32-bit `which + 1` for the all-T k-mer.  Built from pos_index / mers, which stay resident (big read-ends,
// the export entry points and the broadcast use them).
// ------------------------------------------------------------------------------------------------------------
#define IX_REP_EMPTY 0xFFFFFFFFu
#define IX_REP_TOOMANY 0xFFFFFFFEu
#define IX_REP_TOO_MANY_SPOTS 100u

__device__ __forceinline__ uint32_t ix_bucket_len (const uint32_t * pos_index, uint64_t k, uint32_t pos_index_0)
{
  const uint32_t nxt = (k == 0xFFFFFFFFull) ? pos_index_0 : pos_index[k + 1];
  return nxt - pos_index[k];
}

// 16-byte units of the bucket's record (0 unless it holds 2..99 positions)
__global__ void ix_rep_units_kernel (const uint32_t * pos_index, uint32_t * units)
{
  const uint32_t p0 = pos_index[0];
  for (uint64_t k = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; k < (1ull << 32); k += (uint64_t) gridDim.x * blockDim.x)
    {
      const uint32_t ln = ix_bucket_len (pos_index, k, p0);
      units[k] = (ln >= 2 && ln < IX_REP_TOO_MANY_SPOTS) ? (ln + 4u) / 4u : 0u;
    }
}

// exclusive prefix sums over 2^32 u32 values, in place: tile sums, a one-block scan of them, the tiles themselves
__global__ __launch_bounds__ (SC_BLOCK) void ix_sumscan_reduce_kernel (const uint32_t * a, uint64_t n, uint32_t * tile_sum)
{
  __shared__ uint32_t s[SC_BLOCK];
  uint64_t base = (uint64_t) blockIdx.x * SC_TILE;
  uint32_t m = 0;
  for (int r = 0; r < SC_ITEMS; r++)
    {
      uint64_t i = base + (uint64_t) r * SC_BLOCK + threadIdx.x;
      if (i < n)
        m += a[i];
    }
  s[threadIdx.x] = m;
  __syncthreads ();
  for (int o = SC_BLOCK / 2; o > 0; o >>= 1)
    {
      if (threadIdx.x < o)
        s[threadIdx.x] += s[threadIdx.x + o];
      __syncthreads ();
    }
  if (threadIdx.x == 0)
    tile_sum[blockIdx.x] = s[0];
}

__global__ __launch_bounds__ (1024) void ix_sumscan_tiles_kernel (uint32_t * tile_sum, uint64_t n, unsigned long long *total)
{
  __shared__ unsigned long long s_part[1024];
  const uint64_t per = (n + 1023) / 1024;
  const uint64_t lo = threadIdx.x * per, hi = (lo + per < n) ? lo + per : n;
  unsigned long long m = 0;
  for (uint64_t i = lo; i < hi; i++)
    m += tile_sum[i];
  s_part[threadIdx.x] = m;
  __syncthreads ();
  if (threadIdx.x == 0)
    {
      unsigned long long acc = 0;
      for (int i = 0; i < 1024; i++)
        {
          unsigned long long t = s_part[i];
          s_part[i] = acc;
          acc += t;
        }
      *total = acc;
    }
  __syncthreads ();
  unsigned long long acc = s_part[threadIdx.x];
  for (uint64_t i = lo; i < hi; i++)
    {
      uint32_t t = tile_sum[i];
      tile_sum[i] = (uint32_t) acc;        // meaningful only while the total fits 32 bits (checked by the caller)
      acc += t;
    }
}

__global__ __launch_bounds__ (SC_BLOCK) void ix_sumscan_apply_kernel (uint32_t * a, uint64_t n, const uint32_t * tile_carry)
{
  __shared__ uint32_t s[SC_BLOCK];
  uint64_t base = (uint64_t) blockIdx.x * SC_TILE + (uint64_t) threadIdx.x * SC_ITEMS;
  uint32_t v[SC_ITEMS];
  uint32_t m = 0;
#pragma unroll
  for (int r = 0; r < SC_ITEMS; r++)
    {
      uint64_t i = base + r;
      const uint32_t t = (i < n) ? a[i] : 0u;
      v[r] = m;                 // exclusive inside the thread
      m += t;
    }
  s[threadIdx.x] = m;
  __syncthreads ();
  uint32_t mine = m;
  for (int o = 1; o < SC_BLOCK; o <<= 1)
    {
      uint32_t t = (threadIdx.x >= (unsigned) o) ? s[threadIdx.x - o] : 0u;
      __syncthreads ();
      mine += t;
      s[threadIdx.x] = mine;
      __syncthreads ();
    }
  uint32_t carry = tile_carry[blockIdx.x];
  if (threadIdx.x > 0)
    carry += s[threadIdx.x - 1];
#pragma unroll
  for (int r = 0; r < SC_ITEMS; r++)
    {
      uint64_t i = base + r;
      if (i < n)
        a[i] = v[r] + carry;
    }
}

// replica 0 (the reference's k-mer order): the entry of every bucket, and the records of the buckets of 2..99 positions
__global__ void ix_rep_encode_kernel (const uint32_t * pos_index, const uint32_t * mers, const uint32_t * unit_off, uint32_t multi_base,
                                      uint32_t * rep0, uint32_t * multi)
{
  const uint32_t p0 = pos_index[0];
  for (uint64_t k = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; k < (1ull << 32); k += (uint64_t) gridDim.x * blockDim.x)
    {
      const uint32_t st = pos_index[k];
      const uint32_t ln = ix_bucket_len (pos_index, k, p0);
      uint32_t e;
      if (ln == 0)
        e = IX_REP_EMPTY;
      else if (ln >= IX_REP_TOO_MANY_SPOTS)
        e = IX_REP_TOOMANY;
      else if (ln == 1)
        e = mers[st];
      else
        {
          const uint32_t off = unit_off[k];
          e = multi_base + off;
          uint32_t *rec = multi + (size_t) off * 4;
          rec[0] = ln;
          const uint32_t words = ((ln + 4u) / 4u) * 4u;
          for (uint32_t i = 1; i < words; i++)
            rec[i] = (i <= ln) ? mers[st + i - 1] : 0u;
        }
      rep0[k] = e;
    }
}

// replica p from replica 0: dst[k with 4-bit fields 0 and p swapped] = src[k].  One 256-thread block moves the 16 x 16
// entries that share all other fields: 16 whole lines in, 16 whole lines out.
__global__ __launch_bounds__ (256) void ix_rep_permute_kernel (const uint32_t * src, uint32_t * dst, int p)
{
  __shared__ uint32_t tile[16][17];
  const unsigned sh = 4u * (unsigned) p;
  const int a = threadIdx.x >> 4, b = threadIdx.x & 15;
  for (uint64_t t = blockIdx.x; t < (1ull << 24); t += gridDim.x)
    {
      // t = the 24 bits of the other six fields: fields 1..p-1 below field p, fields p+1..7 above it
      const uint64_t low = t & ((1ull << (sh - 4)) - 1ull), high = t >> (sh - 4);
      const uint64_t base = (high << (sh + 4)) | (low << 4);
      tile[a][b] = src[base | ((uint64_t) a << sh) | (uint64_t) b];     // field p = a, field 0 = b
      __syncthreads ();
      dst[base | ((uint64_t) a << sh) | (uint64_t) b] = tile[b][a];     // entry whose field p = b, field 0 = a, stored at (a, b)
      __syncthreads ();
    }
}

// ------------------------------------------------------------------------------------------------------------
// Pileup export (the final genome walk, pemapper.c:828-843): the device's counter planes (PmPile) -> the reference's u16 columns.
// ------------------------------------------------------------------------------------------------------------
// out[(pos - first) * 6 + col] for `count` positions from `first`: the planes transposed back to the reference's six columns per position
__global__ void pile_to_u16_kernel (PmPile counts, uint64_t first, uint64_t count, uint16_t * out)
{
  uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
  if (i < count * 6)
    {
      const uint64_t pos = i / 6;
      out[i] = pm_pile_get (counts, first + pos, (int) (i - pos * 6));
    }
}

struct __attribute__ ((packed)) PileRec
{
  uint32_t pos;
  uint16_t c[6];
};

// ordered compaction of the non-zero sites of [first, first+count): tile counts, scan, emit
#define PR_BLOCK 256
__device__ __forceinline__ bool pile_nonzero (const PmPile & counts, uint64_t p, uint16_t * c)
{
  unsigned tot = 0;
  for (int k = 0; k < 6; k++)
    {
      c[k] = pm_pile_get (counts, p, k);
      tot += c[k];              // int sum of the six u16 columns, pemapper.c:829-831
    }
  return tot > 0;
}

__global__ __launch_bounds__ (PR_BLOCK) void pile_count_kernel (PmPile counts, uint64_t first, uint64_t count,
                                                                uint32_t * tile_count)
{
  __shared__ unsigned s_cnt;
  if (threadIdx.x == 0)
    s_cnt = 0;
  __syncthreads ();
  uint64_t i = (uint64_t) blockIdx.x * PR_BLOCK + threadIdx.x;
  uint16_t c[6];
  bool nz = i < count && pile_nonzero (counts, first + i, c);
  if (nz)
    atomicAdd (&s_cnt, 1u);
  __syncthreads ();
  if (threadIdx.x == 0)
    tile_count[blockIdx.x] = s_cnt;
}

__global__ __launch_bounds__ (PR_BLOCK) void pile_emit_kernel (PmPile counts, uint64_t first, uint64_t count,
                                                               const uint64_t * tile_offset, PileRec * out, uint64_t cap)
{
  __shared__ unsigned s_wave[PR_BLOCK / 64];
  uint64_t i = (uint64_t) blockIdx.x * PR_BLOCK + threadIdx.x;
  uint16_t c[6];
  bool nz = i < count && pile_nonzero (counts, first + i, c);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  unsigned long long bal = __ballot (nz);
  unsigned before = __popcll (bal & ((1ull << lane) - 1ull));
  if (lane == 0)
    s_wave[wv] = __popcll (bal);
  __syncthreads ();
  unsigned woff = 0;
  for (int w = 0; w < wv; w++)
    woff += s_wave[w];
  if (nz)
    {
      uint64_t at = tile_offset[blockIdx.x] + woff + before;
      if (at < cap)
        {
          PileRec r;
          r.pos = (uint32_t) (first + i);
          for (int k = 0; k < 6; k++)
            r.c[k] = c[k];
          out[at] = r;
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// Synthetic workload.  Everything is a pure function of (seed, position / read number).
// ------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t sy_mix (uint64_t x)
{
  x += 0x9E3779B97F4A7C15ull;   // splitmix64
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}

__device__ __forceinline__ uint64_t sy_hash (uint64_t seed, uint64_t a, uint64_t b)
{
  return sy_mix (sy_mix (seed ^ (a * 0xD6E8FEB86659FD93ull)) + b);
}

#define SY_TILE 320             // repeat copies are 320-base tiles
#define SY_FAMILIES 48

// genome letter at real position p of contig c (contig-local position lp, contig length cl)
__device__ __forceinline__ uint8_t sy_base (uint64_t seed, uint64_t p, uint64_t lp, uint64_t cl, unsigned rep_thresh)
{
  const uint8_t L[4] = { 'A', 'C', 'G', 'T' };
  // N runs: 10 kb at both contig ends, and a centromere-like block of 1 % in the middle
  uint64_t tel = cl > 400000 ? 10000 : cl / 40;
  if (lp < tel || lp + tel >= cl)
    return 'N';
  uint64_t cen0 = cl / 2, cen1 = cen0 + cl / 100;
  if (lp >= cen0 && lp < cen1)
    return 'N';
  uint64_t tile = p / SY_TILE, tp = p % SY_TILE;
  uint64_t th = sy_hash (seed, 0x11, tile);
  if ((unsigned) (th & 0xFFFFFF) < rep_thresh)
    {
      // repeat copy: the family level is geometric (P(level k) = 2^-(k+1)), four families per level, so that copy
      // numbers span about 10^3 .. 10^6 on a 3 Gbp genome: high-copy families overflow too_many_spots, low-copy
      // ones produce multi-hit ends
      unsigned u = (unsigned) ((th >> 24) & 0xFFFF);
      int lvl = u ? (__clz ((int) u) - 16) : 15;
      if (lvl > 11)
        lvl = 11;
      int fam = lvl * 4 + (int) ((th >> 40) & 3);
      unsigned div = 2 + (unsigned) (fam % 7) * 3;        // 2 .. 20 % divergence from the family consensus
      uint64_t ch = sy_hash (seed, 0x22 + fam, tp);
      uint64_t mh = sy_hash (seed, 0x33, p);
      unsigned b = (unsigned) (ch & 3);
      if ((unsigned) (mh % 100) < div)
        b = (b + 1 + (unsigned) ((mh >> 8) % 3)) & 3;
      return L[b];
    }
  return L[sy_hash (seed, 0x44, p) & 3];
}

__global__ void sy_genome_kernel (uint64_t seed, uint8_t * genome, uint64_t gsize, const uint64_t * real_starts, int n_contigs,
                                  unsigned rep_thresh)
{
  uint64_t p = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= gsize)
    return;
  int c = ix_contig_of (real_starts, n_contigs, p);
  genome[p] = sy_base (seed, p, p - real_starts[c], real_starts[c + 1] - real_starts[c], rep_thresh);
}

__device__ __forceinline__ uint8_t sy_comp (uint8_t c)
{
  return c == 'A' ? 'T' : c == 'C' ? 'G' : c == 'G' ? 'C' : c == 'T' ? 'A' : 'N';
}

// one thread per read end
__global__ void sy_reads_kernel (uint64_t seed, const uint8_t * genome, uint64_t gsize, int n, int read_len, int paired,
                                 unsigned sub_thresh, unsigned indel_thresh, uint64_t first_read, uint8_t * reads1, int *len1,
                                 uint8_t * reads2, int *len2, int stride, unsigned one_indel_thresh)
{
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  int ends = paired ? 2 * n : n;
  if (t >= ends)
    return;
  int pr = paired ? t >> 1 : t, which = paired ? (t & 1) : 0;
  uint64_t rn = first_read + (uint64_t) pr;
  uint64_t h0 = sy_hash (seed, 0x55, rn);
  uint64_t fl = 300 + (h0 >> 40) % 201;
  uint64_t s = sy_hash (seed, 0x66, rn) % (gsize - 700);
  bool swap = (h0 >> 20) & 1;
  // end A reads the fragment forward from s; end B reads it backward (complemented) from s + fl - 1
  bool endB = paired ? ((which == 1) != swap) : false;
  uint8_t *dst = (which == 0 ? reads1 : reads2) + (size_t) pr * stride;
  const uint8_t L[4] = { 'A', 'C', 'G', 'T' };
  int64_t cur = endB ? (int64_t) (s + fl - 1) : (int64_t) s;
  const int64_t dir = endB ? -1 : 1;
  int k = 0;
  uint64_t salt = (uint64_t) t * 1315423911ull;
  // one_indel_thresh (BASELINE config "5 % indel-enriched reads"): that share of the read-ends carries ONE insertion or deletion
  // of 1..10 bases somewhere between base 20 and base read_len - 20, instead of the per-base indel rate
  const uint64_t ih = sy_hash (seed ^ salt, 0x99, rn);
  bool one_pending = one_indel_thresh && read_len > 60 && (unsigned) (ih & 0xFFFFFF) < one_indel_thresh;
  const int one_pos = 20 + (int) ((ih >> 24) % (uint64_t) (read_len > 60 ? read_len - 40 : 1));
  const int one_len = 1 + (int) ((ih >> 44) % 10);
  const bool one_ins = (ih >> 60) & 1;
  while (k < read_len)
    {
      uint64_t mh = sy_hash (seed ^ salt, 0x77 + which, (uint64_t) k * 4 + (uint64_t) (cur & 3));
      unsigned r24 = (unsigned) (mh & 0xFFFFFF);
      if (one_pending && k == one_pos)
        {
          one_pending = false;
          if (one_ins)
            for (int q = 0; q < one_len && k < read_len; q++)
              dst[k++] = L[(sy_hash (seed ^ salt, 0x9a, (uint64_t) q) >> 30) & 3];
          else
            cur += dir * one_len;
          continue;
        }
      if (r24 < indel_thresh / 2)
        {
          dst[k++] = L[(mh >> 30) & 3];   // insertion
          continue;
        }
      if (r24 < indel_thresh)
        cur += dir;             // deletion
      uint8_t g = (cur >= 0 && (uint64_t) cur < gsize) ? genome[cur] : 'N';
      if (endB)
        g = sy_comp (g);
      cur += dir;
      if (g != 'N' && (unsigned) ((mh >> 24) & 0xFFFFFF) < sub_thresh)
        {
          unsigned b = g == 'A' ? 0 : g == 'C' ? 1 : g == 'G' ? 2 : 3;
          g = L[(b + 1 + (unsigned) ((mh >> 50) % 3)) & 3];
        }
      dst[k++] = g;
    }
  (which == 0 ? len1 : len2)[pr] = read_len;
}
