// pecall_site.hip.h -- PECaller's per-site caller (the body of call_single_base, src/pecaller.c:1207-1691, with or without
// a pedigree) on gfx950: ONE WAVE per pileup column, persistent over columns.
//
//   lane = sample (INDIV <= 64) for everything that is per sample: set-up (1230-1260), fill_sample_like (2448-2507),
//          marginal posteriors and calls (1443-1468), classification counts (1575-1597);
//   lane = joint configuration (or candidate configuration) for the beam: fill_config_like (2347-2360),
//          fill_config_probs (2511-2788: every (configuration, genotype) candidate of a sample is priced by its own lane,
//          then the reference's running best_like / best_post acceptance rule is replayed over them in order),
//          clean_config_probs (2248-2344: stable rank sort by posterior, cut at 2.3 nats / 514, homozygous fallback);
//   lane = (genotype, allele) cell for the Dirichlet re-estimation (1475-1553) and = genotype row for
//          check_alpha_sanity (2076-2188).
//
// Every floating-point sum is taken in the reference's order by a single lane, so results agree with the CPU to the
// last bit wherever libm agrees (exp / pow / log of the device are used in the normalisation, the re-estimation and
// ln n! above 10000 only; the ln n! table and the Hardy-Weinberg table come from the host, built with its libm).
//
// Configurations live in two pools (current list / list being built).  Almost every column needs one to three
// configurations, so the pools sit in LDS (32 entries); a column whose list outgrows that moves to per-wave pools in HBM
// (flat pointers: the code is the same) for the rest of its passes.
#pragma once
#include <type_traits>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "pecall_kernels.hip.h"

#define PCS_MAXN 512            // samples per call at most: one lane per sample and chunk of 64 (NCH = 1, 2, 4 or 8 chunks; 8: 124 KB of LDS, a wave per CU)
#define PCS_ROW(NCH) (64 * (NCH))
#define PCS_NA 6
#define PCS_NG 14
#define PCS_MAXCFG 514          // max_configs, pecaller.c:1180
#define PCS_SMALL 32            // list length a pool in LDS takes
#define PCS_SMALLCAP 40         // its slots (+1 for the homozygous fallback, padded)
#define PCS_BIGCAP 7232         // >= max_gen * (max_configs + 1) + 1 (pecaller.c:1194), multiple of 8

struct PcsParams
{
  int indiv, max_gen, min_depth, haploid;
  double threshold, ln_theta;
  const double *tab;            // ln n!, n <= 10000
  const double *hw;             // ln_HW[n][minor][hets] flattened; hw_off[n] = start of the (2n+1) x (n+1) matrix
  const int *hw_off;
  // pedigree (use_ped = y): parents (-1 = none), sex, each sample's kids in ped-file order, the de-novo tables of main
  int use_ped;
  double ln_denovo;
  const int16_t *dad, *mom;             // [PCS_MAXN]
  const int8_t *sex;                    // [PCS_MAXN]
  const uint16_t *kid_off, *kid_list;   // [PCS_MAXN + 1], [2 * PCS_MAXN]
  const short *dyad;            // [4][15][15]
  const short *trio;            // [4][15][15][15]
};

struct PcsPool
{
  double *like, *prior, *post;
  int16_t *acount;              // [cap][6]
  int16_t *hets;
  int16_t *nden;                // no_denovo
  int8_t *nall;
  int8_t *calls;                // [cap][64]
  uint16_t *ord;                // list position -> slot
  uint16_t *ord2;               // scratch of the sort
  int cap;                      // longest list
  int row;                      // bytes of a calls row: 64 per chunk of samples
};

// phase probes of the beam search (only in a library built with -DPECALL_TIMING_PROBES, tools/variants.sh): wave cycles per phase, summed
// over the waves of a launch into pcs_probe[], printed by pecall_dev_sites_run under PECALL_LIST_STATS
#ifdef PECALL_TIMING_PROBES
__device__ unsigned long long pcs_probe[16];
struct PcsProbe
{
  unsigned long long acc[16], last;
};
#define PCS_T(pp, i) do { const unsigned long long t_ = __builtin_readcyclecounter (); (pp).acc[i] += t_ - (pp).last; (pp).last = t_; } while (0)
#else
struct PcsProbe
{
};
#define PCS_T(pp, i) do { } while (0)
#endif

template < int NCH > struct __align__ (16) PcsSmallPool
{
  double like[PCS_SMALLCAP], prior[PCS_SMALLCAP], post[PCS_SMALLCAP];
  int8_t calls[PCS_SMALLCAP][PCS_ROW (NCH)];
  int16_t acount[PCS_SMALLCAP][PCS_NA];
  int16_t hets[PCS_SMALLCAP], nden[PCS_SMALLCAP];
  uint16_t ord[PCS_SMALLCAP], ord2[PCS_SMALLCAP];
  int8_t nall[PCS_SMALLCAP];
};

// bytes of one per-wave pool in HBM
#define PCS_BIG_BYTES_OF(ROW) ((size_t) PCS_BIGCAP * (3 * 8 + (ROW) + 2 * PCS_NA + 2 + 2 + 2 + 2 + 1 + 5))

template < int NCH > struct __align__ (16) PcsShared
{
  PcsSmallPool < NCH > pool[2];
  double like[PCS_ROW (NCH)][PCS_NG + 1];       // per-sample genotype log-likelihoods of the pass
  double mean[PCS_NG][PCS_NA], var[PCS_NG][PCS_NA], wt[PCS_NG][PCS_NA], fr[PCS_NG][PCS_NA];
  int al[PCS_NG][PCS_NA], first[PCS_NG][PCS_NA];
  int reads[PCS_ROW (NCH)][PCS_NA];
  int tot[PCS_ROW (NCH)];
  typename std::conditional < (NCH > 4), uint16_t, uint8_t >::type sord[PCS_ROW (NCH)];          // samples by margin, descending (a byte each while they are at most 256)
  uint8_t dup[PCS_SMALLCAP];
};

// one bit per sample, 64 to a word: the samples above the depth floor, the settled ones ...
template < int NCH > struct PcsMask
{
  unsigned long long w[NCH];
  __device__ __forceinline__ bool test (int i) const
  {
    unsigned long long v = w[0];
#pragma unroll
    for (int c = 1; c < NCH; c++)
      v = ((i >> 6) == c) ? w[c] : v;
    return (v >> (i & 63)) & 1ull;
  }
  __device__ __forceinline__ int count () const
  {
    int n = 0;
#pragma unroll
    for (int c = 0; c < NCH; c++)
      n += (int) __popcll (w[c]);
    return n;
  }
};

// element (chunk i >> 6) of a per-chunk register array, i the same in every lane
template < int NCH, class T > __device__ __forceinline__ T pcs_chunk_of (const T (&x)[NCH], int i)
{
  T v = x[0];
#pragma unroll
  for (int c = 1; c < NCH; c++)
    v = ((i >> 6) == c) ? x[c] : v;
  return v;
}

__device__ __forceinline__ void pcs_sync ()
{
  __builtin_amdgcn_fence (__ATOMIC_SEQ_CST, "wavefront");
  __builtin_amdgcn_wave_barrier ();
}

// value of lane l, l the same in every lane: v_readlane_b32 (an SGPR broadcast, no LDS round trip)
__device__ __forceinline__ int pcs_bcast (int v, int l)
{
  return __builtin_amdgcn_readlane (v, l);
}

__device__ __forceinline__ double pcs_bcast (double v, int l)
{
  return __hiloint2double (__builtin_amdgcn_readlane (__double2hiint (v), l), __builtin_amdgcn_readlane (__double2loint (v), l));
}

template < int NCH > __device__ __forceinline__ void pcs_pool_small (PcsPool & p, PcsSmallPool < NCH > *s)
{
  p.like = s->like;
  p.prior = s->prior;
  p.post = s->post;
  p.acount = &s->acount[0][0];
  p.hets = s->hets;
  p.nden = s->nden;
  p.nall = s->nall;
  p.calls = &s->calls[0][0];
  p.ord = s->ord;
  p.ord2 = s->ord2;
  p.cap = PCS_SMALL;
  p.row = PCS_ROW (NCH);
}

__device__ __forceinline__ void pcs_pool_big (PcsPool & p, char *base, int row)
{
  p.like = (double *) base;
  p.prior = p.like + PCS_BIGCAP;
  p.post = p.prior + PCS_BIGCAP;
  p.calls = (int8_t *) (p.post + PCS_BIGCAP);
  p.acount = (int16_t *) (p.calls + (size_t) PCS_BIGCAP * row);
  p.hets = p.acount + (size_t) PCS_BIGCAP * PCS_NA;
  p.ord = (uint16_t *) (p.hets + PCS_BIGCAP);
  p.ord2 = p.ord + PCS_BIGCAP;
  p.nden = (int16_t *) (p.ord2 + PCS_BIGCAP);
  p.nall = (int8_t *) (p.nden + PCS_BIGCAP);
  p.cap = PCS_BIGCAP - 8;
  p.row = row;
}

// get_het_alleles, pecaller.c:2191-2245
__host__ __device__ __forceinline__ void pcs_het (int g, int &a, int &b, int ref)
{
  if (g < PCS_NA)
    a = b = g;
  else if (g < 12)
    {
      const int k = g - 6;      // 01 02 03 12 13 23
      a = k < 3 ? 0 : (k < 5 ? 1 : 2);
      b = k < 3 ? k + 1 : (k < 5 ? k - 1 : 3);
    }
  else
    {
      a = ref;
      b = g == 12 ? 4 : 5;
    }
}

// allele_counts[ref][g][k], pecaller.c:725-737
__device__ __forceinline__ int pcs_ac (int ref, int g, int k, int haploid)
{
  int a, b;
  pcs_het (g, a, b, ref);
  return (a == k ? 1 : 0) + ((!haploid && b == k) ? 1 : 0);
}

// genotype_order[ref][jj], pecaller.c:617-722 (diploid, then haploid)
__constant__ unsigned char pcs_order_tab[2][4][PCS_NG] = {
  {{0, 7, 6, 8, 12, 13, 1, 2, 3, 4, 5, 9, 10, 11}, {1, 10, 6, 9, 12, 13, 0, 2, 3, 4, 5, 7, 8, 11},
   {2, 7, 9, 11, 12, 13, 0, 1, 3, 4, 5, 6, 8, 10}, {3, 10, 8, 11, 12, 13, 1, 0, 2, 4, 5, 6, 7, 9}},
  {{0, 2, 1, 3, 4, 5}, {1, 3, 0, 2, 4, 5}, {2, 0, 1, 3, 4, 5}, {3, 1, 0, 2, 4, 5}}
};

__device__ __forceinline__ int pcs_order (int ref, int jj, int haploid)
{
  return pcs_order_tab[haploid ? 1 : 0][ref][jj];
}

// fill_alpha_prior, pecaller.c:3043-3139: the integer Dirichlet pseudo-counts of genotype row g for reference base dom
// (normal_factor 300: hom = 300, het = 150)
__host__ __device__ __forceinline__ void pcs_alpha_prior_row (int g, int dom, int (&row)[PCS_NA])
{
  const int normal_factor = 300;
  const int hom = normal_factor, het = normal_factor / 2;
  const int hom_err = hom / 300 > 1 ? hom / 300 : 1, err = (2 * het) / 300 > 1 ? (2 * het) / 300 : 1;
  for (int k = 0; k < PCS_NA; k++)
    row[k] = err;
  if (g < 4)
    for (int k = 0; k < PCS_NA; k++)
      row[k] = (g == k) ? hom : hom_err;
  else if (g == 4 || g == 5)
    {
      for (int k = 0; k < 4; k++)
        row[k] = (k == dom) ? (g == 4 ? hom / 5 : hom) : err;
      row[4] = g == 4 ? (4 * hom) / 5 : err;
      row[5] = g == 4 ? err : (4 * hom) / 5;
    }
  else if (g < 12)
    {
      int a, b;
      pcs_het (g, a, b, dom);
      if (a == dom || b == dom)
        {
          for (int k = 0; k < PCS_NA; k++)
            row[k] = (k == dom) ? (51 * het) / 50 : (k == (a == dom ? b : a)) ? (49 * het) / 50 : (k == 4) ? (het / 20 > 1 ? het / 20 : 1) : err;
        }
      else
        for (int k = 0; k < PCS_NA; k++)
          row[k] = (k == a || k == b) ? het : err;
    }
  else if (g == 12)
    for (int k = 0; k < PCS_NA; k++)
      row[k] = (k == 4) ? (4 * het) / 5 : (k == dom) ? (6 * het) / 5 : err;
  else
    for (int k = 0; k < PCS_NA; k++)
      row[k] = (k == 5) ? (2 * het) / 5 : (k == dom) ? (8 * het) / 5 : err;
}

// config_alloc, pecaller.c:2987-3027, into slot s: every sample above the depth floor called homozygous `dom`
template < int NCH > __device__ __forceinline__ void pcs_cfg_init (const PcsPool & p, int s, int dom, const PcsMask < NCH > &deep, int haploid, int lane)
{
#pragma unroll
  for (int c = 0; c < NCH; c++)
    p.calls[(size_t) s * p.row + 64 * c + lane] = ((deep.w[c] >> lane) & 1ull) ? (int8_t) dom : (int8_t) PCS_NG;
  if (lane < PCS_NA)
    p.acount[s * PCS_NA + lane] = (lane == dom) ? (int16_t) (deep.count () * (haploid ? 1 : 2)) : (int16_t) 0;
  if (lane == 0)
    {
      p.hets[s] = 0;
      p.nden[s] = 0;
      p.nall[s] = deep.count () ? 1 : 0;
      p.like[s] = 0;
      p.prior[s] = 0;
      p.post[s] = 1;
    }
}

// fill_config_like for slot s, by one lane
template < int NCH > __device__ __forceinline__ void pcs_cfg_like (const PcsPool & p, int s, const PcsShared < NCH > &sh, const PcsMask < NCH > &deep, int N)
{
  double l = 0;
  for (int i = 0; i < N; i++)
    if (deep.test (i))
      l += sh.like[i][p.calls[(size_t) s * p.row + i]];
  p.like[s] = l;
  p.post[s] = l + p.prior[s];
}

// the same by the whole wave: lane = sample fetches its term (the slot's call of the sample -- a byte of a pool that may lie in HBM -- and the
// likelihood it selects), then the terms are added in sample order as the reference adds them.  One lane walking the samples waited for a
// trip to memory per sample: pcs_clean's homozygous fallback (a list without a homozygous configuration -- every cut of a variant column
// with many carriers) was 86-91 % of the beam search's cycles at 256 and 512 samples and half of them at 64 (profiles/r04_ab_sweeps.txt).
template < int NCH > __device__ __forceinline__ double pcs_cfg_like_wave (const PcsPool & p, int s, const PcsShared < NCH > &sh, const PcsMask < NCH > &deep, int N,
                                                                          int lane)
{
  double term[NCH];
#pragma unroll
  for (int c = 0; c < NCH; c++)
    {
      const int i = 64 * c + lane;
      term[c] = (i < N && ((deep.w[c] >> lane) & 1ull)) ? sh.like[i][p.calls[(size_t) s * p.row + i]] : 0.0;
    }
  // (sample order = chunk by chunk, the set bits of the chunk's mask in ascending order: a scalar bit scan and two v_readlane per sample --
  // a loop over the sample number that selects the chunk's register and tests the chunk's mask per sample was ~60 instructions a sample)
  double l = 0;
#pragma unroll
  for (int c = 0; c < NCH; c++)
    {
      const double tc = term[c];
      unsigned long long m = deep.w[c];
      while (m)
        {
          const int k = __ffsll ((long long) m) - 1;
          m &= m - 1;
          l += pcs_bcast (tc, k);
        }
    }
  if (lane == 0)
    {
      p.like[s] = l;
      p.post[s] = l + p.prior[s];
    }
  return l;                     // (the same in every lane)
}

// the likelihood of the configuration "every sample above the depth floor homozygous for allele a", a = 0 .. 3, once it has been summed in
// a pass: it depends on the pass's likelihoods only, and pcs_clean's fallback asks for it after nearly every expansion of a variant column
struct PcsHomCache
{
  double v[4];
  unsigned have;
};

// stable sort of the list's first n positions by posterior, descending (qsort + sort_configs; glibc's merge sort is stable)
__device__ __forceinline__ void pcs_sort (PcsPool & p, int n, int lane)
{
  for (int pos = lane; pos < n; pos += 64)
    {
      const int s = p.ord[pos];
      const double pe = p.post[s];
      int rank = 0;
      for (int j = 0; j < n; j++)
        {
          const double pj = p.post[p.ord[j]];
          rank += (pj > pe) || (pj == pe && j < pos);
        }
      p.ord2[rank] = (uint16_t) s;
    }
  pcs_sync ();
  for (int pos = lane; pos < n; pos += 64)
    p.ord[pos] = p.ord2[pos];
  pcs_sync ();
}

// clean_config_probs, pecaller.c:2248-2344; returns the new list length
template < int NCH > __device__ __forceinline__ int pcs_clean (PcsPool & p, int n, int ref, double ct, const PcsShared < NCH > &sh, const PcsMask < NCH > &deep,
                                                              const PcsParams & P, int site_hap, int lane, PcsProbe & pp, PcsHomCache & hc)
{
  pcs_sort (p, n, lane);
  PCS_T (pp, 4);
  int mx = min (PCS_MAXCFG, n);
  {
    const double p0 = p.post[p.ord[0]];
    // the first position whose posterior is more than ct below the best (the list is sorted)
    int cut = mx;
    for (int i0 = 1; i0 < mx && cut == mx; i0 += 64)
      {
        const int i = i0 + lane;
        const unsigned long long m = __ballot (i < mx && p0 > p.post[p.ord[i < mx ? i : 0]] + ct);
        if (m)
          cut = i0 + __ffsll ((long long) m) - 1;
      }
    mx = cut;
  }
  bool found_hom = false;
  for (int i0 = 0; i0 < mx && !found_hom; i0 += 64)
    {
      const int i = i0 + lane;
      found_hom = __any (i < mx && p.nall[p.ord[i < mx ? i : 0]] == 1);
    }
  PCS_T (pp, 10);
  if (!found_hom)
    {
      const int s0 = p.ord[0];
      int best_hom = 0;
      for (int i = 1; i < PCS_NA; i++)
        if (p.acount[s0 * PCS_NA + i] > p.acount[s0 * PCS_NA + best_hom])
          best_hom = i;
      if (best_hom > 3)
        best_hom = ref;
      const int s = n;          // slots 0 .. n-1 hold the list, slot n is free
      pcs_cfg_init (p, s, best_hom, deep, site_hap, lane);
      pcs_sync ();
      double lh = 0.0;
      if ((hc.have >> best_hom) & 1u)
        {
#pragma unroll
          for (int a = 0; a < 4; a++)
            lh = (a == best_hom) ? hc.v[a] : lh;
        }
      else
        {
          lh = pcs_cfg_like_wave (p, s, sh, deep, P.indiv, lane);
#pragma unroll
          for (int a = 0; a < 4; a++)
            hc.v[a] = (a == best_hom) ? lh : hc.v[a];
          hc.have |= 1u << best_hom;
        }
      if (lane == 0)
        {
          p.like[s] = lh;
          p.post[s] = lh;
          p.ord[mx] = (uint16_t) s;
        }
      pcs_sync ();
      PCS_T (pp, 11);
      if (p.post[s] > p.post[p.ord[mx - 1]])
        pcs_sort (p, mx + 1, lane);
      PCS_T (pp, 12);
      mx++;
    }
  PCS_T (pp, 5);
  return mx;
}

// copy the first n slots of a pool (dense list, ord[i] = i) to another pool
__device__ __forceinline__ void pcs_migrate (const PcsPool & from, const PcsPool & to, int n, int lane)
{
  for (int s = lane; s < n; s += 64)
    {
      to.like[s] = from.like[s];
      to.prior[s] = from.prior[s];
      to.post[s] = from.post[s];
      to.hets[s] = from.hets[s];
      to.nden[s] = from.nden[s];
      to.nall[s] = from.nall[s];
      to.ord[s] = from.ord[s];
      for (int k = 0; k < PCS_NA; k++)
        to.acount[s * PCS_NA + k] = from.acount[s * PCS_NA + k];
    }
  for (int s = 0; s < n; s++)
    for (int c = 0; c < from.row; c += 64)
      to.calls[(size_t) s * to.row + c + lane] = from.calls[(size_t) s * from.row + c + lane];
  pcs_sync ();
}

// add_denovo, pecaller.c:2396-2445; genotype 14 = not called; chrom 0 autosome, 1 X, 2 Y, 3 MT
__device__ __forceinline__ int pcs_add_denovo (const PcsParams & P, int kid, int dad, int mom, int sex, int chrom, int ref)
{
  const short *dy = P.dyad + (size_t) ref * 225, *tr = P.trio + (size_t) ref * 3375;
  if (dad < PCS_NG)
    {
      if (mom < PCS_NG)
        {
          if (chrom == 0)
            return tr[(dad * 15 + mom) * 15 + kid];
          if (chrom == 1)
            return sex == 1 ? dy[mom * 15 + kid] : tr[(dad * 15 + mom) * 15 + kid];
          if (chrom == 2)
            return sex == 1 ? dy[dad * 15 + kid] : 0;
          if (chrom == 3)
            return dy[mom * 15 + kid];
          return 0;
        }
      if (chrom == 0 || (chrom == 1 && sex == 2) || (chrom == 2 && sex == 1))
        return dy[dad * 15 + kid];
      return 0;
    }
  if (mom < PCS_NG && chrom != 2)
    return dy[mom * 15 + kid];
  return 0;
}

// the de-novo events sample `who` takes part in under one configuration's calls with who's own call replaced by `mine`
// (pecaller.c:2578-2603): as a child, and as a parent of each of its kids; dg / mg carry over from one kid to the next when
// a kid lacks that parent, as in the reference (they are initialised once, before the loop)
__device__ __forceinline__ int pcs_denovo_around (const PcsParams & P, const int8_t * calls, int who, int mine, int chrom, int ref)
{
#define PCS_CALL(x) ((x) == who ? mine : (int) calls[x])
  int n = 0;
  const int d = P.dad[who], m = P.mom[who];
  if (d >= 0)
    n += pcs_add_denovo (P, mine, PCS_CALL (d), m >= 0 ? PCS_CALL (m) : PCS_NG, P.sex[who], chrom, ref);
  else if (m >= 0)
    n += pcs_add_denovo (P, mine, PCS_NG, PCS_CALL (m), P.sex[who], chrom, ref);
  int dg = PCS_NG, mg = PCS_NG;
  for (int q = P.kid_off[who]; q < P.kid_off[who + 1]; q++)
    {
      const int kid = P.kid_list[q];
      const int kd = P.dad[kid], km = P.mom[kid];
      if (kd >= 0)
        dg = PCS_CALL (kd);
      if (km >= 0)
        mg = PCS_CALL (km);
      n += pcs_add_denovo (P, PCS_CALL (kid), dg, mg, P.sex[kid], chrom, ref);
    }
#undef PCS_CALL
  return n;
}

// fill_config_probs, pecaller.c:2511-2788: the configurations of `cur` re-decided for sample `who`.
// dupbuf: one byte per list position (LDS for short lists, HBM for long ones).  Returns the length of the list built in nw;
// nw may be switched to `big_nw` (the wave's pool in HBM) when it outgrows LDS: *went_big is set.
template < int NCH > __device__ int pcs_expand (const PcsPool & cur, PcsPool & nw, const PcsPool & big_nw, bool &went_big, int n, int who, int ref, double thres,
                                                const PcsShared < NCH > &sh, uint8_t * dupbuf, int r4, int r5, int chrom, int site_hap, const PcsParams & P, int lane,
                                                PcsProbe & pp)
{
  const int G = P.max_gen, N = P.indiv;
  // ---- a configuration equal to an earlier one on every other sample is skipped (pecaller.c:2542-2558)
  const int ww = who >> 3;
  const unsigned long long wmask = ~(0xFFull << (8 * (who & 7)));
  for (int i = lane; i < n; i += 64)
    {
      const unsigned long long *ci = (const unsigned long long *) (cur.calls + (size_t) cur.ord[i] * cur.row);
      int dup = 0;
      for (int ii = 0; ii < i && !dup; ii++)
        {
          const unsigned long long *cj = (const unsigned long long *) (cur.calls + (size_t) cur.ord[ii] * cur.row);
          int same = 1;
          for (int w = 0; w < (N + 7) / 8 && same; w++)
            {
              unsigned long long x = ci[w] ^ cj[w];
              if (w == ww)
                x &= wmask;
              if (w == (N - 1) / 8 && (N & 7))
                x &= (1ull << (8 * (N & 7))) - 1ull;
              same = (x == 0);
            }
          dup = same;
        }
      dupbuf[i] = (uint8_t) dup;
    }
  pcs_sync ();
  PCS_T (pp, 1);
  double best_post = cur.post[cur.ord[0]], best_like = cur.like[cur.ord[0]];
  int newcount = 0;
  const long total = (long) n * G;
  for (long c0 = 0; c0 < total; c0 += 64)
    {
      // ---- every candidate (configuration, genotype) of this chunk priced by its own lane
      const long c = c0 + lane;
      const bool in = c < total;
      const int pos = in ? (int) (c / G) : 0, jj = in ? (int) (c - (long) pos * G) : 0;
      const int s = cur.ord[pos];
      const bool valid = in && !dupbuf[pos];
      const int j = pcs_order (ref, jj, P.haploid);
      const int g_old = cur.calls[(size_t) s * cur.row + who];
      double base = cur.like[s];
      if (g_old < PCS_NG)
        base -= sh.like[who][g_old];
      double templ = base + sh.like[who][j];
      // an indel genotype needs three supporting reads (pecaller.c:2622-2625)
      if ((j == 4 || j == 12) && r4 < 3)
        templ -= 1e10;
      if ((j == 13 || j == 5) && r5 < 3)
        templ -= 1e10;
      // allele counts with the sample's old genotype taken out and the candidate put in (allele_counts[ref][g][.]: one count
      // for each allele of the genotype, the second only when diploid)
      int ac[PCS_NA], nall = 0;
      int oa = -1, ob = -1, na, nb;
      if (g_old < PCS_NG)
        pcs_het (g_old, oa, ob, ref);
      pcs_het (j, na, nb, ref);
      if (P.haploid)
        {
          ob = -1;
          nb = -1;
        }
#pragma unroll
      for (int k = 0; k < PCS_NA; k++)
        {
          ac[k] = cur.acount[s * PCS_NA + k] - (k == oa) - (k == ob) + (k == na) + (k == nb);
          nall += ac[k] > 0;
        }
      const int hets = cur.hets[s] - ((g_old < PCS_NG && g_old >= PCS_NA) ? 1 : 0) + (j >= PCS_NA ? 1 : 0);
      int nden = 0;
      if (P.use_ped && valid)
        {
          const int8_t *row = cur.calls + (size_t) s * cur.row;
          nden = cur.nden[s] - (g_old < PCS_NG ? pcs_denovo_around (P, row, who, g_old, chrom, ref) : 0) + pcs_denovo_around (P, row, who, j, chrom, ref);
        }
      double prior = 0;
      if (nall > 1)
        prior = (nall - 1) * P.ln_theta;
      if (nden > 0)
        prior += nden * P.ln_denovo;
      if (!site_hap && nall > 1)
        {
          int major = 0, minor = 0;
#pragma unroll
          for (int k = 1; k < PCS_NA; k++)
            if (ac[k] > ac[major])
              major = k;
#pragma unroll
          for (int k = 0; k < PCS_NA; k++)
            if (k != major)
              minor += ac[k];
          int mj = 0;
#pragma unroll
          for (int k = 0; k < PCS_NA; k++)
            mj = (k == major) ? ac[k] : mj;
          major = mj;
          if (minor > major)
            {
              const int sw = major;
              major = minor;
              minor = sw;
            }
          const int hh = min (minor, hets);
          const int tot_n = (minor + major) / 2;
          if ((minor - hh) % 2 == 1)
            {
              minor++;
              major++;
            }
          if (valid)
            prior += P.hw[P.hw_off[tot_n] + (long) minor * (tot_n + 1) + hh];
        }
      const double post = prior + templ;
      // ---- the reference's acceptance rule, in candidate order (pecaller.c:2628, 2738-2758)
      // best_like / best_post only grow, so a candidate that fails the first test against their values at the start of the
      // chunk fails it later too: only the others take part in the ordered replay
      unsigned long long m = __ballot (valid && ((templ + thres > best_post) || (templ + 0.01 > best_like)));
      PCS_T (pp, 2);
      while (m)
        {
          const int k = __ffsll ((long long) m) - 1;
          m &= m - 1;
          const double t = pcs_bcast (templ, k), po = pcs_bcast (post, k);
          if (!((t + thres > best_post) || (t + 0.01 > best_like)))
            continue;
          best_like = (t > best_like) ? t : best_like;
          best_post = (po > best_post) ? po : best_post;
          if (!(po + thres > best_post))
            continue;
          if (newcount == nw.cap && !went_big)
            {
              pcs_migrate (nw, big_nw, newcount, lane);
              nw = big_nw;
              went_big = true;
            }
          const int sk = pcs_bcast (s, k), jk = pcs_bcast (j, k);
          const int d = newcount;
#pragma unroll
          for (int c = 0; c < NCH; c++)
            nw.calls[(size_t) d * nw.row + 64 * c + lane] = (64 * c + lane == who) ? (int8_t) jk : cur.calls[(size_t) sk * cur.row + 64 * c + lane];
          const double pr = pcs_bcast (prior, k);
          const int hk = pcs_bcast (hets, k), nk = pcs_bcast (nall, k), dk = pcs_bcast (nden, k);
          int ak = 0;
#pragma unroll
          for (int q = 0; q < PCS_NA; q++)
            {
              const int v = pcs_bcast (ac[q], k);
              ak = (lane == q) ? v : ak;
            }
          if (lane < PCS_NA)
            nw.acount[d * PCS_NA + lane] = (int16_t) ak;
          if (lane == 0)
            {
              nw.like[d] = t;
              nw.prior[d] = pr;
              nw.post[d] = po;
              nw.hets[d] = (int16_t) hk;
              nw.nden[d] = (int16_t) dk;
              nw.nall[d] = (int8_t) nk;
              nw.ord[d] = (uint16_t) d;
            }
          newcount++;
        }
      PCS_T (pp, 3);
    }
  pcs_sync ();
  return newcount;
}

// calls and site classification of one column (pecaller.c:1565-1636, de-novo rows 1650-1671), lane = sample (of each chunk of 64):
// what the reference prints for it -- calls and posteriors, Allele_Counts, the row's type, the passes it took
// (RD: the reads of chunk c as six ints and their depth -- the caller's registers, or the column's row read again where the registers do not hold NCH x 6)
template < int NCH, class RD > __device__ __forceinline__ void pcs_write_site_rd (const PcsParams & P, long site, int lane, int dom, int chrom, RD rd,
                                                                              double average_depth, const int (&final_call)[NCH],
                                                                              const double (&final_p)[NCH], int pass, int8_t * call, double *post_out,
                                                                              int8_t * type_out, int32_t * allele_count, int8_t * n_pass, int32_t * denovo_out)
{
  const int N = P.indiv, md = P.min_depth;
  // ---- calls and site classification (pecaller.c:1565-1636)
  const double low_base = (8 > 0.4 * average_depth) ? 8 : 0.4 * average_depth;
  int ac6[PCS_NA] = { 0, 0, 0, 0, 0, 0 }, on_target = 0, off_target = 0, not_low = 0;
  bool called[NCH];
#pragma unroll
  for (int c = 0; c < NCH; c++)
    {
      int rc[PCS_NA], totc;
      rd (c, rc, totc);
      called[c] = 64 * c + lane < N && totc > md;
      if (called[c] && final_p[c] >= P.threshold)
        {
#pragma unroll
          for (int a = 0; a < PCS_NA; a++)
            {
              const int k = pcs_ac (dom, final_call[c], a, P.haploid);
              if (k)
                {
                  ac6[a] += k;
                  on_target += rc[a];
                }
              else if (a != dom || final_call[c] != PCS_NA - 1)
                off_target += rc[a];
            }
          if (totc > low_base && final_call[c] != dom)
            not_low += 1;
        }
    }
  for (int o = 32; o; o >>= 1)
    {
#pragma unroll
      for (int a = 0; a < PCS_NA; a++)
        ac6[a] += __shfl_xor (ac6[a], o);
      on_target += __shfl_xor (on_target, o);
      off_target += __shfl_xor (off_target, o);
      not_low += __shfl_xor (not_low, o);
    }
#pragma unroll
  for (int c = 0; c < NCH; c++)
    if (64 * c + lane < N)
      {
        call[site * N + 64 * c + lane] = called[c] ? (int8_t) final_call[c] : (int8_t) PCS_NG;
        post_out[site * N + 64 * c + lane] = called[c] ? final_p[c] : 1.0;
      }
  int n_all = 0, isdel = 0, isins = 0, type = 0;
#pragma unroll
  for (int a = 0; a < PCS_NA; a++)
    if (ac6[a] > 0)
      {
        n_all++;
        if (a == 4)
          isdel = 1;
        else if (a == 5)
          isins = 1;
        else if (a != dom)
          type = 1;
      }
  int dom_count = 0;
#pragma unroll
  for (int a = 0; a < PCS_NA; a++)
    dom_count = (a == dom) ? ac6[a] : dom_count;
  if (n_all > 1 || (n_all > 0 && dom_count < 1))
    {
      if ((double) off_target / (double) (on_target + off_target) > 0.15)
        type = 6;
      else if (n_all > 2)
        type = 5;
      else if (not_low > 0)
        type = isdel ? 2 : isins ? 3 : 1;
      else
        type = 4;
    }
  int mine = 0;
#pragma unroll
  for (int a = 0; a < PCS_NA; a++)
    mine = (lane == a) ? ac6[a] : mine;
  if (lane < PCS_NA)
    allele_count[site * PCS_NA + lane] = mine;
  // ---- de-novo events among the confident calls (pecaller.c:1650-1671): the row's type gets a DENOVO_ prefix
  int dcount = 0;
  if (type && P.use_ped)
    {
      int fc[NCH];
      double fp[NCH];
#pragma unroll
      for (int c = 0; c < NCH; c++)
        {
          fc[c] = called[c] ? final_call[c] : PCS_NG;
          fp[c] = called[c] ? final_p[c] : 1.0;
        }
      // a parent's call and posterior: the parent's lane of the parent's chunk
      auto from_sample = [&] (int who, int &c_out, double &p_out)
      {
        const int wl = who >= 0 ? (who & 63) : 0, wc = who >= 0 ? (who >> 6) : 0;
        c_out = PCS_NG;
        p_out = 1.0;
#pragma unroll
        for (int c = 0; c < NCH; c++)
          {
            const int vc = __shfl (fc[c], wl);
            const double vp = __hiloint2double (__shfl (__double2hiint (fp[c]), wl), __shfl (__double2loint (fp[c]), wl));
            if (wc == c)
              {
                c_out = vc;
                p_out = vp;
              }
          }
      };
#pragma unroll
      for (int c = 0; c < NCH; c++)
        {
          const int smp = 64 * c + lane;
          const int d = smp < N ? P.dad[smp] : -1, m = smp < N ? P.mom[smp] : -1;
          int dfc, mfc;
          double dfp, mfp;
          from_sample (d, dfc, dfp);
          from_sample (m, mfc, mfp);
          if (smp < N && fp[c] >= P.threshold)
            dcount += pcs_add_denovo (P, fc[c], (d >= 0 && dfp >= P.threshold) ? dfc : PCS_NG, (m >= 0 && mfp >= P.threshold) ? mfc : PCS_NG, P.sex[smp],
                                      chrom, dom);
        }
      for (int o = 32; o; o >>= 1)
        dcount += __shfl_xor (dcount, o);
    }
  if (lane == 0)
    {
      type_out[site] = (int8_t) type;
      n_pass[site] = (int8_t) pass;
      denovo_out[site] = dcount;
    }
}

template < int NCH > __device__ __forceinline__ void pcs_write_site (const PcsParams & P, long site, int lane, int dom, int chrom, const int (&r)[NCH][PCS_NA],
                                                                    const int (&tot)[NCH], double average_depth, const int (&final_call)[NCH],
                                                                    const double (&final_p)[NCH], int pass, int8_t * call, double *post_out,
                                                                    int8_t * type_out, int32_t * allele_count, int8_t * n_pass, int32_t * denovo_out)
{
  auto rd = [&] (const int c, int (&rc)[PCS_NA], int &totc)
  {
#pragma unroll
    for (int a = 0; a < PCS_NA; a++)
      rc[a] = r[c][a];
    totc = tot[c];              // (the caller's: 0 in a column the site filters drop)
  };
  pcs_write_site_rd < NCH > (P, site, lane, dom, chrom, rd, average_depth, final_call, final_p, pass, call, post_out, type_out, allele_count, n_pass, denovo_out);
}

// (one chunk of samples, as the shortcut kernel has them)
__device__ __forceinline__ void pcs_write_site (const PcsParams & P, long site, int lane, int dom, int chrom, const int (&r)[PCS_NA], int tot,
                                                double average_depth, int final_call, double final_p, int pass, int8_t * call, double *post_out,
                                                int8_t * type_out, int32_t * allele_count, int8_t * n_pass, int32_t * denovo_out)
{
  int r1[1][PCS_NA];
#pragma unroll
  for (int a = 0; a < PCS_NA; a++)
    r1[0][a] = r[a];
  const int t1[1] = { tot }, c1[1] = { final_call };
  const double p1[1] = { final_p };
  pcs_write_site < 1 > (P, site, lane, dom, chrom, r1, t1, average_depth, c1, p1, pass, call, post_out, type_out, allele_count, n_pass, denovo_out);
}

#define PCS_BUCKETS 4           // parts of the list of columns left to the beam search (pcs_fast_kernel files, pcs_call_kernel walks them)
// One wave per site.  Outputs are what the reference prints per row: call 0..13 or 14 (N), posterior,
// site type (0 REF, 1 SNP, 2 DEL, 3 INS, 4 LOW, 5 MULTIALLELIC, 6 MESS; -1 = reference base not A/C/G/T, skipped),
// Allele_Counts, passes.
// NCH = chunks of 64 samples (1, 2, 4 or 8): a lane stands for sample 64 c + lane of every chunk c
extern __shared__ __align__ (16) uint8_t pcs_call_lds[];
template < int NCH > __global__ __launch_bounds__ (64) void pcs_call_kernel (PcsParams P, const uint16_t * reads, const uint8_t * dom_of, const uint8_t * chrom_of,
                                                                          long n_sites, int8_t * call, double *post_out, int8_t * type_out,
                                                                          int32_t * allele_count, int8_t * n_pass, int32_t * denovo_out, char *scratch,
                                                                          unsigned long long *next_site, const unsigned *site_list, const unsigned *n_list)
{
  PcsShared < NCH > &sh = *reinterpret_cast < PcsShared < NCH > *>(pcs_call_lds);
  constexpr int ROW = PCS_ROW (NCH);
  const int lane = threadIdx.x;
  const int N = P.indiv, G = P.max_gen, md = P.min_depth;
  char *my = scratch + (size_t) blockIdx.x * (2 * PCS_BIG_BYTES_OF (ROW) + PCS_BIGCAP);
  PcsPool bigp[2];
  pcs_pool_big (bigp[0], my, ROW);
  pcs_pool_big (bigp[1], my + PCS_BIG_BYTES_OF (ROW), ROW);
  uint8_t *big_dup = (uint8_t *) (my + 2 * PCS_BIG_BYTES_OF (ROW));
  PcsProbe pp;
#ifdef PECALL_TIMING_PROBES
  for (int i_ = 0; i_ < 16; i_++)
    pp.acc[i_] = 0ull;
  pp.last = __builtin_readcyclecounter ();
#endif
  // columns are handed out through a counter (the first grid-ful by block index): a column that needs the whole beam search
  // takes ~50 times as long as one settled by the shortcut below, so a fixed stride would leave most waves waiting for a few
  // (site_list: only the listed columns -- the ones pcs_fast_kernel could not settle)
  // The list comes in PCS_BUCKETS parts of n_sites slots each, by the number of samples that disagreed with the shortcut (n_list[b]
  // columns in part b): the parts are walked from the last -- the columns likely to hold the largest beams -- to the first, so
  // that the launch does not end waiting for a heavy column that was handed out late.
  long cnt[PCS_BUCKETS] = { 0, 0, 0, 0 };
  long n_iter = n_sites;
  if (site_list)
    {
      n_iter = 0;
      for (int b = 0; b < PCS_BUCKETS; b++)
        {
          cnt[b] = (long) n_list[b];
          n_iter += cnt[b];
        }
    }
  for (long it = blockIdx.x; it < n_iter; it = (long) gridDim.x + (long) __shfl ((long long) (lane == 0 ? atomicAdd (next_site, 1ull) : 0ull), 0))
    {
      long site = it;
      if (site_list)
        {
          long k = it;
          int b = PCS_BUCKETS - 1;
          while (b > 0 && k >= cnt[b])
            {
              k -= cnt[b];
              b--;
            }
          site = (long) site_list[(size_t) b * (size_t) n_sites + (size_t) k];
        }
      const int dom = dom_of[site];
      const int chrom = chrom_of[site] & 3;
      // bit 4: HAPLOID forced for this column (BED guide mode on chrY / chrMT, pecaller.c:955-957): only the initial allele
      // counts and the Hardy-Weinberg term see it
      const int site_hap = P.haploid | ((chrom_of[site] >> 4) & 1);
      if (dom > 3)
        {
#pragma unroll
          for (int c = 0; c < NCH; c++)
            if (64 * c + lane < N)
              {
                call[site * N + 64 * c + lane] = PCS_NG;
                post_out[site * N + 64 * c + lane] = 1.0;
              }
          if (lane < PCS_NA)
            allele_count[site * PCS_NA + lane] = 0;
          if (lane == 0)
            {
              type_out[site] = -1;
              n_pass[site] = 0;
              denovo_out[site] = 0;
            }
          continue;
        }
      // ---- per-sample set-up (pecaller.c:1230-1260)
      int r[NCH][PCS_NA], tot[NCH];
      double coef[NCH];
      int initial_call[NCH], final_call[NCH];
      double final_p[NCH];
      int tsum = 0, sample_count = 0;
#pragma unroll
      for (int c = 0; c < NCH; c++)
        {
          const int smp = 64 * c + lane;
#pragma unroll
          for (int a = 0; a < PCS_NA; a++)
            r[c][a] = smp < N ? (int) reads[(site * N + smp) * PCS_NA + a] : 0;
          tot[c] = r[c][0] + r[c][1] + r[c][2] + r[c][3] + r[c][4];
          coef[c] = pc_factln (P.tab, tot[c]);
#pragma unroll
          for (int a = 0; a < PCS_NA; a++)
            coef[c] -= pc_factln (P.tab, r[c][a]);
          initial_call[c] = tot[c] > md ? dom : PCS_NG;
          final_call[c] = initial_call[c];
          final_p[c] = 1.0;
          // average depth: the reference adds the integer totals into a double in sample order; integers add exactly
          tsum += tot[c];
          sample_count += (int) __popcll (__ballot (smp < N && tot[c] >= 8));
        }
      for (int o = 32; o; o >>= 1)
        tsum += __shfl_xor (tsum, o);
      const double average_depth = (double) tsum / (double) N;
      bool bad_base = average_depth < 8;
      if (sample_count < (double) 0.5 * N && chrom != 2)
        bad_base = true;
      PcsMask < NCH > deep;
#pragma unroll
      for (int c = 0; c < NCH; c++)
        {
          if (bad_base)
            tot[c] = 0;
          deep.w[c] = __ballot (64 * c + lane < N && tot[c] > md);
#pragma unroll
          for (int a = 0; a < PCS_NA; a++)
            sh.reads[64 * c + lane][a] = r[c][a];
          sh.tot[64 * c + lane] = tot[c];
        }
      PcsPool pool[2];
      pcs_pool_small (pool[0], &sh.pool[0]);
      pcs_pool_small (pool[1], &sh.pool[1]);
      int ci = 0;               // pool[ci] = current list
      bool big = false;
      int total = 1, pass = 0;
      bool calls_changed = !bad_base;
      const int normal_factor = 300;
      if (!bad_base)
        {
          pcs_cfg_init (pool[0], 0, dom, deep, site_hap, lane);
          if (lane == 0)
            pool[0].ord[0] = 0;
          // ---- fill_alpha_prior (pecaller.c:3043-3139): lane = genotype row
          if (lane < G)
            {
              int row[PCS_NA];
              pcs_alpha_prior_row (lane, dom, row);
              for (int k = 0; k < PCS_NA; k++)
                sh.al[lane][k] = row[k];
            }
        }
      pcs_sync ();
      const double ct = 2.3;    // starting_threshold
      while (calls_changed && pass < 5)
        {
          pass++;
          PcsHomCache hom_cache;        // (of this pass's likelihoods)
          hom_cache.have = 0u;
#pragma unroll
          for (int a = 0; a < 4; a++)
            hom_cache.v[a] = 0.0;
          // ---- Dirichlet means of the pass (pecaller.c:1354-1364)
          if (lane < G)
            {
              int myt = 0;
              for (int a = 0; a < PCS_NA; a++)
                {
                  myt += sh.al[lane][a];
                  sh.first[lane][a] = sh.al[lane][a];
                }
              for (int a = 0; a < PCS_NA; a++)
                sh.mean[lane][a] = (double) sh.al[lane][a] / (double) myt;
            }
          pcs_sync ();
          // ---- fill_sample_like (pecaller.c:2448-2507), lane = sample
          double norm = 1.0;
          for (int i = 2; i <= pass; i++)
            norm *= 2.5;        // new_norm[pass], pecaller.c:1339-1344
          double initial_p[NCH];
#pragma unroll
          for (int c = 0; c < NCH; c++)
            {
              const int smp = 64 * c + lane;
              initial_p[c] = 0.0;
              initial_call[c] = PCS_NG;
              if (smp < N && tot[c] > md)
                {
                  const double sc0 = (double) min (tot[c], 100) * norm;
                  const double sc1 = (10 > sc0) ? 10 : sc0;
                  const double scale = (1000 < sc1) ? 1000 : sc1;
                  double mx = -1e100;
                  int best = PCS_NG;
                  for (int g = 0; g < G; g++)
                    {
                      int tot_a = 0, tot_tot = 0;
                      double cf = coef[c], lk = 0.0;
                      for (int a = 0; a < PCS_NA; a++)
                        {
                          const double cv = ceil (scale * sh.mean[g][a]);
                          const int ta = (int) ((1 > cv) ? 1 : cv);
                          tot_a += ta;
                          tot_tot += ta + r[c][a];
                          cf -= pc_factln (P.tab, ta - 1);
                          lk += pc_factln (P.tab, ta + r[c][a] - 1);
                        }
                      cf += pc_factln (P.tab, tot_a - 1);
                      lk += cf;
                      lk -= pc_factln (P.tab, tot_tot - 1);
                      sh.like[smp][g] = lk;
                      if (lk > mx)
                        {
                          best = g;
                          mx = lk;
                        }
                    }
                  initial_p[c] = 1e100;
                  initial_call[c] = best;
                  for (int g = 0; g < G; g++)
                    if (g != best)
                      {
                        const double dlt = mx - sh.like[smp][g];
                        initial_p[c] = (dlt < initial_p[c]) ? dlt : initial_p[c];
                      }
                }
            }
          // ---- shortcut for the column every sample agrees on (the overwhelming majority).  If, in the first pass, every sample
          //      above the depth floor has the reference homozygote as its best genotype with a margin of more than 2.31 nats, the beam
          //      never holds more than the all-reference configuration: for each sample in turn the candidate "reference homozygote"
          //      comes first (genotype_order starts with it) and is kept; every other candidate j has
          //      templ = (L - l_ref) + l_j <= L - 2.31 + rounding, which fails both acceptance tests of pecaller.c:2628 / 2750
          //      (templ + 2.3 > best_post, templ + 0.01 > best_like; best_post = best_like = L for a configuration with one allele,
          //      whose prior is 0) whatever its prior is -- unsupported indel genotypes (2622-2625) only sink further.  The list
          //      stays one configuration through clean_config_probs (it is homozygous), its posterior normalises to exactly 1, each
          //      sample's call is the reference base with final_p = 1, nothing changed, and the pass loop ends after pass 1
          //      (pecaller.c:1454-1471).  The 0.01 nat between 2.3 and 2.31 is ~10^10 times the rounding of the sums involved.
          if (pass == 1)
            {
              bool ok = true;
#pragma unroll
              for (int c = 0; c < NCH; c++)
                ok = ok && (!(64 * c + lane < N && tot[c] > md) || (initial_call[c] == dom && initial_p[c] > 2.31));
              if (__all (ok))
                {
#pragma unroll
                  for (int c = 0; c < NCH; c++)
                    {
                      final_call[c] = (64 * c + lane < N && tot[c] > md) ? dom : PCS_NG;
                      final_p[c] = 1.0;
                    }
                  break;
                }
            }
          // samples by margin, descending, stable (sort_compare_sample_pointer)
          {
            int rank[NCH];
#pragma unroll
            for (int c = 0; c < NCH; c++)
              rank[c] = 0;
            for (int jn = 0; jn < N; jn++)
              {
                const double pj = pcs_bcast (pcs_chunk_of < NCH > (initial_p, jn), jn & 63);
#pragma unroll
                for (int c = 0; c < NCH; c++)
                  rank[c] += (pj > initial_p[c]) || (pj == initial_p[c] && jn < 64 * c + lane);
              }
#pragma unroll
            for (int c = 0; c < NCH; c++)
              if (64 * c + lane < N)
                sh.sord[rank[c]] = (typename std::conditional < (NCH > 4), uint16_t, uint8_t >::type) (64 * c + lane);
          }
          pcs_sync ();
          // (a pass starts from one configuration: the wave adds its samples' terms; a longer list a lane per configuration)
          if (total <= 4)
            for (int i = 0; i < total; i++)
              pcs_cfg_like_wave (pool[ci], pool[ci].ord[i], sh, deep, N, lane);
          else
            for (int i = lane; i < total; i += 64)
              pcs_cfg_like (pool[ci], pool[ci].ord[i], sh, deep, N);
          pcs_sync ();
          PCS_T (pp, 0);
          total = pcs_clean (pool[ci], total, dom, ct, sh, deep, P, site_hap, lane, pp, hom_cache);
          // ---- the settled samples at the head of the order, in one go.  While the list is the single all-`dom` configuration, a
          //      sample whose best genotype is `dom` by more than 2.31 nats leaves it that: pcs_expand prices the candidate `dom`
          //      first (templ = (like - l_dom) + l_dom, prior 0: it is kept and becomes the list), and every other candidate has
          //      templ <= that - 2.31 + rounding, which fails both acceptance tests (templ + 2.3 > best_post, templ + 0.01 >
          //      best_like, pecaller.c:2628 / 2750); pcs_clean then finds the one homozygous configuration and changes nothing.  The
          //      only thing that moves is the configuration's likelihood, by the rounding of that subtraction and addition: it is
          //      replayed here sample by sample, in the same order, instead of a whole expansion + clean-up per sample (the samples
          //      are ordered by margin, descending, so in a typical column all but the last one or two are of this kind).
          int k_first = 0;
          if (total == 1 && pool[ci].ord[0] == 0 && pool[ci].nall[0] == 1 && pool[ci].prior[0] == 0.0 && pool[ci].nden[0] == 0
              && pool[ci].acount[dom] == (int16_t) (deep.count () * (site_hap ? 1 : 2)))
            {
              PcsMask < NCH > settled;
#pragma unroll
              for (int c = 0; c < NCH; c++)
                settled.w[c] = __ballot (64 * c + lane < N && tot[c] > md && initial_call[c] == dom && initial_p[c] > 2.31);
              while (k_first < N && settled.test (sh.sord[k_first]))
                k_first++;
              if (k_first > 0)
                {
                  double L = pool[ci].like[0];
                  for (int k = 0; k < k_first; k++)
                    {
                      const double l = sh.like[sh.sord[k]][dom];
                      L = (L - l) + l;
                    }
                  pcs_sync ();
                  if (lane == 0)
                    {
                      pool[ci].like[0] = L;
                      pool[ci].post[0] = 0.0 + L;
                    }
                  pcs_sync ();
                }
            }
          for (int k = k_first; k < N; k++)
            {
              const int ind = sh.sord[k];
              if (deep.test (ind))
                {
                  const int ni = ci ^ 1;
                  if (big)
                    pool[ni] = bigp[ni];
                  bool went_big = big;
                  const int r4 = sh.reads[ind][4], r5 = sh.reads[ind][5];
                  const int cnt = pcs_expand (pool[ci], pool[ni], bigp[ni], went_big, total, ind, dom, ct, sh, (total <= PCS_SMALLCAP) ? sh.dup : big_dup,
                                              r4, r5, chrom, site_hap, P, lane, pp);
                  big = went_big;
                  ci = ni;
                  total = pcs_clean (pool[ci], cnt, dom, ct, sh, deep, P, site_hap, lane, pp, hom_cache);
                }
              else
                {
                  // a sample under the depth floor is 'N' in every configuration (pecaller.c:1408-1417)
                  for (int i = lane; i < total; i += 64)
                    pool[ci].calls[(size_t) pool[ci].ord[i] * ROW + ind] = PCS_NG;
#pragma unroll
                  for (int c = 0; c < NCH; c++)
                    if (64 * c + lane == ind)
                      {
                        final_call[c] = PCS_NG;
                        final_p[c] = 1.0;
                      }
                  pcs_sync ();
                }
            }
          PCS_T (pp, 6);
          // ---- posteriors of the configurations (pecaller.c:1423-1441): exp of the difference to the best, normalised
          const PcsPool & cp = pool[ci];
          {
            const double max_post = cp.post[cp.ord[0]];
            for (int i = lane; i < total; i += 64)
              {
                const int s = cp.ord[i];
                const double dlt = cp.post[s] - max_post;
                cp.post[s] = dlt > -40 ? exp (dlt) : 0;
              }
            pcs_sync ();
            double tot_post = 0;
            for (int i = 0; i < total; i++)
              tot_post += cp.post[cp.ord[i]];
            pcs_sync ();
            for (int i = lane; i < total; i += 64)
              cp.post[cp.ord[i]] /= tot_post;
            pcs_sync ();
          }
          // ---- marginal posteriors and calls (pecaller.c:1443-1468), lane = sample
          calls_changed = false;
#pragma unroll
          for (int c = 0; c < NCH; c++)
            if (64 * c + lane < N && tot[c] > md)
              {
                // post_prob[g] = sum of the posteriors of the configurations that call this sample g, in list order; the
                // first largest one is the call
                int besti = 0;
                double bestp = 0;
                for (int g = 0; g < G; g++)
                  {
                    double acc = 0;
                    for (int i = 0; i < total; i++)
                      {
                        const int s = cp.ord[i];
                        if (cp.calls[(size_t) s * ROW + 64 * c + lane] == g)
                          acc += cp.post[s];
                      }
                    if (g == 0 || acc > bestp)
                      {
                        besti = g;
                        bestp = acc;
                      }
                  }
                final_p[c] = bestp;
                final_call[c] = besti;
                if (final_call[c] != initial_call[c] || final_p[c] < P.threshold)
                  calls_changed = true;
              }
          calls_changed = __any (calls_changed);
          PCS_T (pp, 7);
          if (N < 4 || pass == 5)
            calls_changed = false;
          if (calls_changed)
            {
              // ---- moment-matched re-estimation of the Dirichlet parameters (pecaller.c:1475-1553), lane = (genotype, allele)
              for (int it = lane; it < G * PCS_NA; it += 64)
                {
                  const int g = it / PCS_NA, a = it - g * PCS_NA;
                  double m = 0, v = 0, w = 0;
                  for (int i = 0; i < total; i++)
                    {
                      const int s = cp.ord[i];
                      const double po = cp.post[s];
                      for (int ind = 0; ind < N; ind++)
                        if (deep.test (ind) && cp.calls[(size_t) s * ROW + ind] == g)
                          {
                            const double f = (double) sh.reads[ind][a] / (double) sh.tot[ind];
                            m += f * po;
                            v += (f * f) * po;
                            w += po;
                          }
                    }
                  if (w > 1e-9)
                    {
                      m /= w;
                      v /= w;
                      v -= m * m;
                    }
                  sh.mean[g][a] = m;
                  sh.var[g][a] = v;
                  sh.wt[g][a] = w;
                }
              pcs_sync ();
              if (lane < G)
                {
                  const int g = lane;
                  const double var_eps = 1e-6;
                  int non_zero_var = 0, this_min = 0, little_up = 0;
                  for (int a = 1; a < PCS_NA; a++)
                    if (sh.mean[g][a] > sh.mean[g][little_up])
                      little_up = a;
                  for (int a = 0; a < PCS_NA; a++)
                    {
                      if (sh.wt[g][a] >= 1.5 && sh.var[g][a] > var_eps * sh.mean[g][a])
                        non_zero_var++;
                      if (sh.mean[g][a] < sh.mean[g][this_min])
                        this_min = a;
                      if (sh.mean[g][a] > var_eps && sh.mean[g][a] < sh.mean[g][little_up])
                        little_up = a;
                    }
                  bool use_first = true;
                  if (non_zero_var > 1)
                    {
                      double s0 = 1.0;
                      for (int a = 0; a < PCS_NA; a++)
                        if (a != this_min && sh.var[g][a] > var_eps * sh.mean[g][a])
                          s0 *= sh.mean[g][a] * (1.0 - sh.mean[g][a]) / sh.var[g][a];
                      s0 = pow (s0 - 1.0, (double) 1.0 / (double) (non_zero_var - 1.0));
                      const double lu = 1.0 / sh.mean[g][little_up];
                      s0 = (s0 > lu) ? s0 : lu;
                      if (s0 > 3.0)
                        {
                          use_first = false;
                          for (int a = 0; a < PCS_NA; a++)
                            {
                              const int cv = (int) ceil (sh.mean[g][a] * s0);
                              sh.al[g][a] = (1 > cv) ? 1 : cv;
                            }
                        }
                    }
                  if (use_first)
                    for (int a = 0; a < PCS_NA; a++)
                      sh.al[g][a] = sh.first[g][a];
                }
              pcs_sync ();
              // ---- check_alpha_sanity (pecaller.c:2076-2188), lane = genotype row; rows only read the reference base's row
              //      of the fractions, which no row modifies
              if (lane < G)
                {
                  int tt = 0;
                  for (int a = 0; a < PCS_NA; a++)
                    tt += sh.al[lane][a];
                  for (int a = 0; a < PCS_NA; a++)
                    sh.fr[lane][a] = (double) sh.al[lane][a] / (double) tt;
                }
              pcs_sync ();
              if (lane < G)
                {
                  const int i = lane;
                  bool reset = false;
                  if (i < 4)
                    {
                      int mxi = 0;
                      for (int a = 1; a < PCS_NA; a++)
                        if (sh.al[i][a] > sh.al[i][mxi])
                          mxi = a;
                      if (mxi != i)
                        reset = true;
                      else
                        for (int a = 0; a < PCS_NA; a++)
                          if (a != i && sh.fr[i][a] > 0.3)
                            reset = true;
                    }
                  else if (i == 4)
                    reset = sh.fr[4][4] - sh.fr[dom][4] < 0.5;
                  else if (i == 5)
                    reset = sh.fr[5][5] - sh.fr[dom][5] < -0.1;
                  else
                    {
                      int a, b;
                      pcs_het (i, a, b, dom);
                      if (b == dom)
                        {
                          const int t = a;
                          a = b;
                          b = t;
                        }
                      if (sh.fr[i][b] - sh.fr[dom][b] < 0.25)
                        reset = true;
                      else
                        {
                          double fa = sh.fr[i][a], fb = sh.fr[i][b];
                          if (dom == a)
                            fa -= 0.05;
                          else
                            fa -= (sh.fr[dom][a] > 0.05) ? sh.fr[dom][a] : 0.05;
                          fb -= (0.05 > sh.fr[dom][b]) ? 0.05 : sh.fr[dom][b];
                          for (int q = 0; q < PCS_NA && !reset; q++)
                            if (q != a && q != b)
                              if (sh.fr[i][q] > fa || sh.fr[i][q] > fb)
                                reset = true;
                        }
                    }
                  if (reset)
                    for (int a = 0; a < PCS_NA; a++)
                      sh.al[i][a] = sh.first[i][a];
                  double scale = sh.al[i][0];
                  for (int a = 1; a < PCS_NA; a++)
                    scale += sh.al[i][a];
                  scale = (double) normal_factor / scale;
                  for (int a = 0; a < PCS_NA; a++)
                    if (sh.al[i][a] > 1)
                      {
                        const int cv = (int) ceil (scale * (double) sh.al[i][a]);
                        sh.al[i][a] = (1 > cv) ? 1 : cv;
                      }
                }
              pcs_sync ();
            }
#pragma unroll
          for (int c = 0; c < NCH; c++)
            initial_call[c] = final_call[c];
        }
      PCS_T (pp, 8);
      pcs_write_site < NCH > (P, site, lane, dom, chrom, r, tot, average_depth, final_call, final_p, pass, call, post_out, type_out, allele_count, n_pass, denovo_out);
      pcs_sync ();
      PCS_T (pp, 9);
    }
#ifdef PECALL_TIMING_PROBES
  if (lane == 0)
    for (int i_ = 0; i_ < 16; i_++)
      if (pp.acc[i_])
        atomicAdd (&pcs_probe[i_], pp.acc[i_]);
#endif
}

// ---- The columns with a few unsettled samples, without the pools.  In a column of BASELINE config 4 (64 samples, 30x) one or two
// samples usually carry two or three error reads and miss the shortcut's 2.31-nat margin: 1.03 M of 2 M columns, each of which cost
// pcs_call_kernel a whole wave with 36 KB of LDS for ~70 K cycles.  What the reference computes for such a column is small: the
// samples are taken in descending order of margin, so the settled ones come first and leave the single all-reference configuration
// as it is (the argument at pcs_call_kernel's replay: only its likelihood moves, by the rounding of (L - l) + l, sample by sample);
// then each unsettled sample expands a list of one to four configurations by its 14 genotypes (fill_config_probs, pecaller.c:2511-
// 2788) and the list is sorted and cut (clean_config_probs, 2248-2344); posteriors, marginals and calls follow (1423-1468).  Here the
// wave that computed the column's likelihoods in pcs_fast_kernel does that on the spot: a configuration lives in a LANE (likelihood,
// prior, posterior, allele counts, heterozygotes, the genotypes of the unsettled samples packed four bits each), a candidate
// (configuration, genotype) is priced by its own lane exactly as pcs_expand prices it, the acceptance rule is replayed in candidate
// order, the sort is a rank + ds_permute.  Every floating-point sum runs in the reference's order in one lane or in wave-uniform
// registers.  The column is handed to pcs_call_kernel instead (-> false, nothing written) whenever the small form does not apply:
// a pedigree, more than PCS_MINI_U unsettled samples, an unsettled sample whose margin is not below every settled one's (a real
// variant carrier: it would come first in the order), a list that grows past four configurations before the last expansion, no
// homozygous configuration left after a cut (the fallback of 2286-2333), or calls that change (a second pass).
#define PCS_MINI_U 4
// allele counts of a configuration packed into ints: a byte each while the column has at most 64 samples (255 >= 2 x 64 alleles), 16 bits
// each beyond (NCH chunks of 64 samples)
template < int NCH > struct PcsAcPack
{
  static constexpr int BITS = NCH == 1 ? 8 : 16, PER = NCH == 1 ? 3 : 2, WORDS = PCS_NA / PER, MASK = (1 << BITS) - 1;
  static __device__ __forceinline__ int pack (const int (&ac)[PCS_NA], int w)
  {
    int v = 0;
#pragma unroll
    for (int q = 0; q < PER; q++)
      v |= ac[w * PER + q] << (BITS * q);
    return v;
  }
  static __device__ __forceinline__ int get (const int (&word)[WORDS], int k)     // (k a compile-time constant at every use)
  {
    return (word[k / PER] >> (BITS * (k % PER))) & MASK;
  }
};

// The small beam's sizes.  Up to 128 samples (NCH <= 2, all likelihoods in the caller's registers): four unsettled samples, their genotypes
// in an int above the configuration's heterozygote and allele counts.  Beyond (pcs_fast_kernel's form for 4 or 8 chunks, the unsettled
// samples' likelihoods parked in LDS): sixteen, in 64 bits of their own (SPLIT) -- at 512 samples a column has eight on average.
template < int NCH > struct PcsMini
{
  typedef typename std::conditional < (NCH > 2), unsigned long long, int >::type Misc;
  typedef typename std::conditional < (NCH > 4), uint16_t, uint8_t >::type Sord;
  static constexpr int MAXU = NCH > 2 ? 16 : PCS_MINI_U;
  static constexpr bool SPLIT = NCH > 2;
  static constexpr int GSHIFT = SPLIT ? 0 : 12;         // where the genotypes start in Misc
};

__device__ __forceinline__ unsigned long long pcs_bcast (unsigned long long v, int l)
{
  return ((unsigned long long) (unsigned) __builtin_amdgcn_readlane ((int) (v >> 32), l) << 32) | (unsigned long long) (unsigned) __builtin_amdgcn_readlane ((int) v, l);
}

__device__ __forceinline__ int pcs_permute (int dst, int v)
{
  return __builtin_amdgcn_ds_permute (dst, v);
}

__device__ __forceinline__ unsigned long long pcs_permute (int dst, unsigned long long v)
{
  return ((unsigned long long) (unsigned) __builtin_amdgcn_ds_permute (dst, (int) (v >> 32)) << 32) | (unsigned long long) (unsigned) __builtin_amdgcn_ds_permute (dst, (int) v);
}

__device__ __forceinline__ int pcs_shfl (int v, int src)
{
  return __shfl (v, src);
}

__device__ __forceinline__ unsigned long long pcs_shfl (unsigned long long v, int src)
{
  return ((unsigned long long) (unsigned) __shfl ((int) (v >> 32), src) << 32) | (unsigned long long) (unsigned) __shfl ((int) v, src);
}

// lk_dom: every sample's likelihood of the reference homozygote; stage (who, r4, r5) -> the 14 likelihoods of sample `who` where every lane
// can index them (LDS), and its deletion / insertion reads
template < int NCH, class STAGE > __device__ bool pcs_mini_core (const PcsParams & P, const double (&lk_dom)[NCH], const bool (&deep)[NCH],
                                                                const bool (&ok)[NCH], const int (&best)[NCH], const double (&margin)[NCH], int dom, int site_hap,
                                                                typename PcsMini < NCH >::Sord * w_sord, STAGE stage, int lane, int (&final_call)[NCH],
                                                                double (&final_p)[NCH])
{
  typedef PcsAcPack < NCH > AC;
  typedef typename PcsMini < NCH >::Misc Misc;
  typedef typename PcsMini < NCH >::Sord Sord;
  const int N = P.indiv, G = P.max_gen;
  unsigned long long deep_m[NCH], unset_m[NCH];
  int nu = 0, n_deep = 0, n_set = 0;
#pragma unroll
  for (int c = 0; c < NCH; c++)
    {
      deep_m[c] = __ballot (deep[c]);
      unset_m[c] = __ballot (deep[c] && !ok[c]);
      nu += (int) __popcll (unset_m[c]);
      n_deep += (int) __popcll (deep_m[c]);
      n_set += (int) __popcll (deep_m[c] & ~unset_m[c]);
    }
  if (P.use_ped || nu > PcsMini < NCH >::MAXU || nu == 0)
    return false;
  if (n_deep * (site_hap ? 1 : 2) > AC::MASK)
    return false;               // (allele counts are packed here)
  // ---- the order of the samples: margin descending, sample index ascending on ties (sort_compare_sample_pointer; the sort is stable)
  double ip[NCH];               // (the samples under the depth floor last: they take no part in the beam)
  int rank[NCH];
#pragma unroll
  for (int c = 0; c < NCH; c++)
    {
      ip[c] = deep[c] ? margin[c] : -1.0;
      rank[c] = 0;
    }
  for (int jn = 0; jn < N; jn++)
    {
      const double pj = pcs_bcast (pcs_chunk_of < NCH > (ip, jn), jn & 63);
#pragma unroll
      for (int c = 0; c < NCH; c++)
        rank[c] += (pj > ip[c]) || (pj == ip[c] && jn < 64 * c + lane);
    }
  // the unsettled samples must follow every settled one (the leading run of settled samples is then all of them)
  {
    bool early = false;
#pragma unroll
    for (int c = 0; c < NCH; c++)
      early = early || (deep[c] && !ok[c] && rank[c] < n_set);
    if (__any (early))
      return false;
  }
#pragma unroll
  for (int c = 0; c < NCH; c++)
    if (64 * c + lane < N)
      w_sord[rank[c]] = (Sord) (64 * c + lane);
  pcs_sync ();
  int sord[NCH];
#pragma unroll
  for (int c = 0; c < NCH; c++)
    sord[c] = (int) w_sord[64 * c + lane];
  auto sample_at = [&] (int k)->int     // (k the same in every lane)
  {
    return pcs_bcast (pcs_chunk_of < NCH > (sord, k), k & 63);
  };
  // ---- fill_config_like of the all-reference configuration (pecaller.c:2347-2360): the deep samples' likelihoods of `dom`, in
  //      sample order; then the settled samples' expansions, each of which leaves (L - l) + l
  double L = 0.0;
  if constexpr (NCH > 2)
    {
      // (chunk by chunk, the set bits of the chunk's mask in ascending order: the same order without a register select per sample)
#pragma unroll
      for (int c = 0; c < NCH; c++)
        {
          const double tc = lk_dom[c];
          unsigned long long m = deep_m[c];
          while (m)
            {
              const int k = __ffsll ((long long) m) - 1;
              m &= m - 1;
              L += pcs_bcast (tc, k);
            }
        }
    }
  else
    for (int i = 0; i < N; i++)
      if ((pcs_chunk_of < NCH > (deep_m, i) >> (i & 63)) & 1ull)
        L += pcs_bcast (pcs_chunk_of < NCH > (lk_dom, i), i & 63);
  for (int k = 0; k < n_set; k++)
    {
      const int who = sample_at (k);
      const double l = pcs_bcast (pcs_chunk_of < NCH > (lk_dom, who), who & 63);
      L = (L - l) + l;
    }
  // ---- the list: configuration i in lane i
  double c_like = L, c_prior = 0.0, c_post = 0.0 + L;
  constexpr bool SPLIT = PcsMini < NCH >::SPLIT;
  constexpr int GSHIFT = PcsMini < NCH >::GSHIFT;
  int c_ac[AC::WORDS];
  Misc c_misc = SPLIT ? 0 : 1 << 8;     // misc: hets | nall << 8 | genotypes of the unsettled samples << 12 (SPLIT: the genotypes alone ...
  int c_hn = 1 << 8;                    // ... hets | nall << 8 here)
  {
    int ac[PCS_NA] = { 0, 0, 0, 0, 0, 0 };
#pragma unroll
    for (int k = 0; k < 4; k++)
      ac[k] = (k == dom) ? n_deep * (site_hap ? 1 : 2) : 0;
#pragma unroll
    for (int w = 0; w < AC::WORDS; w++)
      c_ac[w] = AC::pack (ac, w);
  }
  (void) c_prior;
  int n = 1;
  const double thres = 2.3;
  for (int t = 0; t < nu; t++)
    {
      if (n > 4)
        return false;
      const int who = sample_at (n_set + t);
      // the sample's likelihoods, where every lane can index them
      int r4, r5;
      const double *w_like = stage (who, r4, r5);
      double best_post = pcs_bcast (c_post, 0), best_like = pcs_bcast (c_like, 0);
      // ---- every candidate (configuration, genotype) priced by its own lane (pcs_expand; the sample's old genotype is `dom`)
      const int total = n * G;
      const bool in = lane < total;
      const int pos = in ? lane / G : 0, jj = in ? lane - pos * G : 0;
      const int j = pcs_order (dom, jj, P.haploid);
      const double s_like = __hiloint2double (__shfl (__double2hiint (c_like), pos), __shfl (__double2loint (c_like), pos));
      int s_ac[AC::WORDS];
#pragma unroll
      for (int w = 0; w < AC::WORDS; w++)
        s_ac[w] = __shfl (c_ac[w], pos);
      const Misc s_misc = pcs_shfl (c_misc, pos);
      const int s_hn = SPLIT ? __shfl (c_hn, pos) : (int) (s_misc & 0xFFF);
      double base = s_like;
      base -= w_like[dom];
      double templ = base + w_like[j];
      if ((j == 4 || j == 12) && r4 < 3)
        templ -= 1e10;
      if ((j == 13 || j == 5) && r5 < 3)
        templ -= 1e10;
      int ac[PCS_NA], nall = 0;
      int oa = dom, ob = dom, na, nb;
      pcs_het (j, na, nb, dom);
      if (P.haploid)
        {
          ob = -1;
          nb = -1;
        }
#pragma unroll
      for (int k = 0; k < PCS_NA; k++)
        {
          const int old = AC::get (s_ac, k);
          ac[k] = old - (k == oa) - (k == ob) + (k == na) + (k == nb);
          nall += ac[k] > 0;
        }
      const int hets = (s_hn & 0xFF) + (j >= PCS_NA ? 1 : 0);
      double prior = 0;
      if (nall > 1)
        prior = (nall - 1) * P.ln_theta;
      if (!site_hap && nall > 1)
        {
          int major = 0, minor = 0;
#pragma unroll
          for (int k = 1; k < PCS_NA; k++)
            if (ac[k] > ac[major])
              major = k;
#pragma unroll
          for (int k = 0; k < PCS_NA; k++)
            if (k != major)
              minor += ac[k];
          int mj = 0;
#pragma unroll
          for (int k = 0; k < PCS_NA; k++)
            mj = (k == major) ? ac[k] : mj;
          major = mj;
          if (minor > major)
            {
              const int sw = major;
              major = minor;
              minor = sw;
            }
          const int hh = min (minor, hets);
          const int tot_n = (minor + major) / 2;
          if ((minor - hh) % 2 == 1)
            {
              minor++;
              major++;
            }
          if (in)
            prior += P.hw[P.hw_off[tot_n] + (long) minor * (tot_n + 1) + hh];
        }
      const double post = prior + templ;
      const int hn_new = hets | (nall << 8);
      const Misc misc_new = (SPLIT ? s_misc : (Misc) hn_new | (s_misc & ~(Misc) 0xFFF)) | ((Misc) j << (GSHIFT + 4 * t));
      int ac_new[AC::WORDS];
#pragma unroll
      for (int w = 0; w < AC::WORDS; w++)
        ac_new[w] = AC::pack (ac, w);
      // ---- the acceptance rule in candidate order (pecaller.c:2628, 2738-2758); the kept ones become the new list, in that order
      double n_like = 0, n_prior = 0, n_post = 0;
      int n_ac[AC::WORDS], newcount = 0, n_hn = 0;
      Misc n_misc = 0;
#pragma unroll
      for (int w = 0; w < AC::WORDS; w++)
        n_ac[w] = 0;
      unsigned long long m = __ballot (in && ((templ + thres > best_post) || (templ + 0.01 > best_like)));
      while (m)
        {
          const int k = __ffsll ((long long) m) - 1;
          m &= m - 1;
          const double tt = pcs_bcast (templ, k), po = pcs_bcast (post, k);
          if (!((tt + thres > best_post) || (tt + 0.01 > best_like)))
            continue;
          best_like = (tt > best_like) ? tt : best_like;
          best_post = (po > best_post) ? po : best_post;
          if (!(po + thres > best_post))
            continue;
          const double pr = pcs_bcast (prior, k);
          const Misc mi = pcs_bcast (misc_new, k);
          const int hn = SPLIT ? pcs_bcast (hn_new, k) : 0;
          int av[AC::WORDS];
#pragma unroll
          for (int w = 0; w < AC::WORDS; w++)
            av[w] = pcs_bcast (ac_new[w], k);
          if (lane == newcount)
            {
              n_like = tt;
              n_prior = pr;
              n_post = po;
#pragma unroll
              for (int w = 0; w < AC::WORDS; w++)
                n_ac[w] = av[w];
              n_misc = mi;
              n_hn = hn;
            }
          newcount++;
        }
      // ---- clean_config_probs (pecaller.c:2248-2344): stable sort by posterior, descending; cut 2.3 nats below the best
      int rk = 0;
      for (int q = 0; q < newcount; q++)
        {
          const double pq = pcs_bcast (n_post, q);
          rk += (pq > n_post) || (pq == n_post && q < lane);
        }
      const int dst = (lane < newcount ? rk : lane) * 4;        // (lanes beyond the list keep their place: a permutation of all 64)
#define PCS_PUSH(x) __builtin_amdgcn_ds_permute (dst, (x))
      c_like = __hiloint2double (PCS_PUSH (__double2hiint (n_like)), PCS_PUSH (__double2loint (n_like)));
      c_prior = __hiloint2double (PCS_PUSH (__double2hiint (n_prior)), PCS_PUSH (__double2loint (n_prior)));
      c_post = __hiloint2double (PCS_PUSH (__double2hiint (n_post)), PCS_PUSH (__double2loint (n_post)));
#pragma unroll
      for (int w = 0; w < AC::WORDS; w++)
        c_ac[w] = PCS_PUSH (n_ac[w]);
      c_misc = pcs_permute (dst, n_misc);
      if (SPLIT)
        c_hn = pcs_permute (dst, n_hn);
      else
        c_hn = (int) (c_misc & 0xFFF);
#undef PCS_PUSH
      int mx = newcount;        // (at most 56: max_configs = 514 is out of reach)
      {
        const double p0 = pcs_bcast (c_post, 0);
        const unsigned long long cutm = __ballot (lane >= 1 && lane < mx && p0 > c_post + thres);
        if (cutm)
          mx = __ffsll ((long long) cutm) - 1;
      }
      if (!__any (lane < mx && ((c_hn >> 8) & 0xF) == 1))
        return false;           // no homozygous configuration in the list: the reference adds one (2286-2333) -- pcs_call_kernel's case
      n = mx;
    }
  // ---- posteriors of the configurations (pecaller.c:1423-1441)
  double e;
  {
    const double max_post = pcs_bcast (c_post, 0);
    const double dlt = c_post - max_post;
    e = (lane < n && dlt > -40) ? exp (dlt) : 0;
    double tot_post = 0;
    for (int i = 0; i < n; i++)
      tot_post += pcs_bcast (e, i);
    e /= tot_post;
  }
  // ---- marginal posteriors and calls (pecaller.c:1443-1468), lane = sample: a settled sample is `dom` in every configuration
  bool changed = false;
#pragma unroll
  for (int c = 0; c < NCH; c++)
    {
      int my_t = -1;
      for (int t = 0; t < nu; t++)
        my_t = (sample_at (n_set + t) == 64 * c + lane) ? t : my_t;
      int besti = 0;
      double bestp = 0;
      for (int g = 0; g < G; g++)
        {
          double acc = 0;
          for (int i = 0; i < n; i++)
            {
              const double pi = pcs_bcast (e, i);
              const Misc gi = pcs_bcast (c_misc, i);
              const int mine = my_t >= 0 ? (int) ((gi >> (GSHIFT + 4 * my_t)) & 0xF) : dom;
              if (mine == g)
                acc += pi;
            }
          if (g == 0 || acc > bestp)
            {
              besti = g;
              bestp = acc;
            }
        }
      if (deep[c])
        {
          final_p[c] = bestp;
          final_call[c] = besti;
          changed = changed || final_call[c] != best[c] || final_p[c] < P.threshold;
        }
      else
        {
          final_call[c] = PCS_NG;
          final_p[c] = 1.0;
        }
    }
  if (__any (changed) && N >= 4)
    return false;               // a second pass (pecaller.c:1454-1471): the whole machinery
  return true;
}

// up to 128 samples: the likelihoods of every sample are in the caller's registers
template < int NCH > __device__ bool pcs_mini_beam (const PcsParams & P, const double (&lk)[NCH][PCS_NG], const int (&r)[NCH][PCS_NA], const bool (&deep)[NCH],
                                                    const bool (&ok)[NCH], const int (&best)[NCH], const double (&margin)[NCH], int dom, int site_hap,
                                                    uint8_t * w_sord, double *w_like, int lane, int (&final_call)[NCH], double (&final_p)[NCH])
{
  double lk_dom[NCH];
#pragma unroll
  for (int c = 0; c < NCH; c++)
    {
      lk_dom[c] = 0.0;
#pragma unroll
      for (int g = 0; g < 4; g++)
        lk_dom[c] = (g == dom) ? lk[c][g] : lk_dom[c];
    }
  auto stage = [&] (int who, int &r4, int &r5)->const double *
  {
    pcs_sync ();
#pragma unroll
    for (int c = 0; c < NCH; c++)
      if (64 * c + lane == who)
        {
#pragma unroll
          for (int g = 0; g < PCS_NG; g++)
            w_like[g] = lk[c][g];
        }
    pcs_sync ();
    int r4v[NCH], r5v[NCH];
#pragma unroll
    for (int c = 0; c < NCH; c++)
      {
        r4v[c] = r[c][4];
        r5v[c] = r[c][5];
      }
    r4 = pcs_bcast (pcs_chunk_of < NCH > (r4v, who), who & 63);
    r5 = pcs_bcast (pcs_chunk_of < NCH > (r5v, who), who & 63);
    return w_like;
  };
  return pcs_mini_core < NCH > (P, lk_dom, deep, ok, best, margin, dom, site_hap, w_sord, stage, lane, final_call, final_p);
}


// ---- the columns every sample agrees on, without the beam.  One wave per column (lane = sample), eight columns per
// workgroup round, the ln n! table (80 KB) and the four first-pass Dirichlet mean tables in the workgroup's LDS.  Per column:
// the set-up and site filters of pecaller.c:1230-1304, the first pass of fill_sample_like (2448-2507), and the test of the
// shortcut argued in pcs_call_kernel: every sample above the depth floor has the reference homozygote as its best genotype
// by more than 2.31 nats.  Then the column's result is known -- every such sample is called the reference base with
// posterior 1 after one pass, site type REF -- and is written here; so are the columns the site filters drop (every call N,
// zero passes) and the ones whose reference base is not A/C/G/T.  Everything else goes to slow_list for pcs_call_kernel.
// The ln n! table in LDS comes in two sizes.  Pass 1 looks up n <= 600 + the sample's depth (ta <= 100 per allele): for a column with
// no sample deeper than ~1,400 reads 2,048 entries serve, a workgroup is 4 waves with 30 KB and 12 waves per CU run (the registers'
// limit) beside whatever else is resident; the form with the whole table (80 KB, workgroups of 8 waves, one per CU, as before round 3)
// takes the columns the first form found too deep and listed, in a launch of its own when there are any.
#define PCS_FAST_TAB 2048
#define PCS_FAST_GRAB 8                // columns a wave of the shortcut kernel takes per fetch of the work counter
#define PCS_FAST_BLOCK_OF(TABN) ((TABN) == PCS_FAST_TAB ? 256 : 512)
// pass 1's integer Dirichlet parameters ta = max (1, ceil (scale * mean[g][a])) (pecaller.c:2478) depend on the reference base, the
// genotype, the allele and on scale = min (depth, 100) clamped to 10 .. 100 only: 4 x 91 x 14 x 6 bytes, built by the host with the
// kernel's arithmetic (pcs_ta_table) and staged in LDS -- a third of the shortcut kernel's instructions were these ceilings
#define PCS_TA_SCALES 91
#define PCS_TA_ROWS (4 * PCS_TA_SCALES * PCS_NG)
#define PCS_TA_BYTES (PCS_TA_ROWS * 4 + PCS_TA_ROWS * 2)        // a0..a3 packed in a word per row, a4 a5 in a half-word
// per wave: the samples' order and one sample's likelihoods (pcs_mini_beam); beyond 128 samples 16 = PcsMini::MAXU unsettled samples' likelihoods
// (128 bytes each) and the order (two bytes a sample beyond 256)
#define PCS_FAST_WAVE_BYTES_OF(NCH) ((NCH) <= 2 ? 64 * (NCH) + PCS_NG * 8 + 16 : 16 * 128 + ((NCH) > 4 ? 2 : 1) * 64 * (NCH))
#define PCS_FAST_LDS_BYTES_OF2(TABN, NCH) (((TABN) + 1) * 8 + PCS_TA_BYTES + (PCS_FAST_BLOCK_OF (TABN) / 64) * PCS_FAST_WAVE_BYTES_OF ((NCH) <= 2 ? 2 : (NCH)))
#define PCS_FAST_LDS_BYTES_OF(TABN) PCS_FAST_LDS_BYTES_OF2 (TABN, 2)

// the ta table as words: PCS_TA_ROWS words (alleles 0..3) followed by PCS_TA_ROWS half-words (alleles 4, 5); every ta is at most 100
static void pcs_ta_table (uint32_t * out)
{
  uint16_t *hi = (uint16_t *) (out + PCS_TA_ROWS);
  for (int dom = 0; dom < 4; dom++)
    for (int g = 0; g < PCS_NG; g++)
      {
        // fill_alpha_prior's row and d_alpha_mean of pass 1 (pecaller.c:3043-3139, 1354-1364)
        int row[PCS_NA];
        pcs_alpha_prior_row (g, dom, row);
        int myt = 0;
        for (int k = 0; k < PCS_NA; k++)
          myt += row[k];
        for (int sc = 0; sc < PCS_TA_SCALES; sc++)
          {
            const double scale = (double) (sc + 10);
            uint32_t w4 = 0, w2 = 0;
            for (int a = 0; a < PCS_NA; a++)
              {
                const double mean = (double) row[a] / (double) myt;
                const double cv = ceil (scale * mean);
                const uint32_t ta = (uint32_t) ((1 > cv) ? 1 : cv);
                if (a < 4)
                  w4 |= ta << (8 * a);
                else
                  w2 |= ta << (8 * (a - 4));
              }
            const int i = (dom * PCS_TA_SCALES + sc) * PCS_NG + g;
            out[i] = w4;
            hi[i] = (uint16_t) w2;
          }
      }
}

// NCH chunks of 64 samples per lane (1: up to 64 samples, the form everything above describes; 2: up to 128 -- every per-sample value an
// array of two, the wave-wide counts sums of two ballots, the small beam with 16-bit allele counts: round 4)
template < int TABN, int NCH > __global__ __launch_bounds__ (PCS_FAST_BLOCK_OF (TABN), (TABN == PCS_FAST_TAB && NCH == 1) ? 3 : 2)
void pcs_fast_kernel (PcsParams P, const uint16_t * reads, const uint8_t * dom_of, const uint8_t * chrom_of,
                      long n_sites, int8_t * call, double *post_out, int8_t * type_out,
                      int32_t * allele_count, int8_t * n_pass, int32_t * denovo_out,
                      unsigned *slow_list, unsigned *n_slow, unsigned *deep_list, unsigned *n_deep, unsigned *next_piece,
                      const uint32_t * ta_table, const uint8_t * skip)
{
  constexpr int PCS_FAST_BLOCK = PCS_FAST_BLOCK_OF (TABN);
  // Two forms.  TABN = PCS_FAST_TAB: the head of the ln n! table in LDS, every column of the range; a column in which a sample is too
  // deep for the head (every n it looks up is <= 6 x 100 + depth) is put on deep_list and left alone.  TABN = PC_TABLE: the whole
  // table, the columns of deep_list only (a second launch, made when the host has seen that the list is not empty).
  constexpr bool LISTED = TABN != PCS_FAST_TAB;
  const long n_items = LISTED ? (long) *n_deep : n_sites;
  extern __shared__ double pcs_fast_lds[];
  double *tab = pcs_fast_lds;
  uint32_t *ta_lo = (uint32_t *) (pcs_fast_lds + TABN + 1);            // [4][PCS_TA_SCALES][PCS_NG]: ta of alleles 0..3, a byte each
  uint16_t *ta_hi = (uint16_t *) (ta_lo + PCS_TA_ROWS);                 // ... of alleles 4, 5
  for (int i = threadIdx.x; i < TABN; i += PCS_FAST_BLOCK)
    tab[i] = P.tab[i];
  for (int i = threadIdx.x; i < PCS_TA_ROWS; i += PCS_FAST_BLOCK)
    ta_lo[i] = ta_table[i];
  for (int i = threadIdx.x; i < PCS_TA_ROWS / 2; i += PCS_FAST_BLOCK)
    ((uint32_t *) ta_hi)[i] = ta_table[PCS_TA_ROWS + i];
  const int N = P.indiv, G = P.max_gen, md = P.min_depth;
  __syncthreads ();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint8_t *w_wave = (uint8_t *) (ta_hi + PCS_TA_ROWS) + (size_t) wave * PCS_FAST_WAVE_BYTES_OF (NCH);
  uint8_t *w_sord = w_wave;             // (up to 128 samples; beyond that the wave's bytes are laid out where they are used)
  double *w_like = (double *) (w_sord + 64 * NCH);
  // Columns are handed out PCS_FAST_GRAB at a time through a counter (next_piece; the first grid-ful of pieces by wave index).
  // With a fixed stride the launch took as long as its unluckiest workgroup: beside the beam searches of earlier chunks, whose
  // waves hold 40 KB of LDS each for milliseconds, a CU now and then has room for two of these workgroups instead of three, the third
  // starts when another one ends -- and then still had its full share to do (2.2 ms per chunk alone, 3.5 ms beside them).
  const long n_pieces = (n_items + PCS_FAST_GRAB - 1) / PCS_FAST_GRAB;
  const long first_free = (long) gridDim.x * (PCS_FAST_BLOCK / 64);
  for (long piece = (long) blockIdx.x * (PCS_FAST_BLOCK / 64) + wave; piece < n_pieces;)
    {
      // (the next piece's number is asked for now and looked at when this piece is done)
      unsigned nx = 0u;
      if (lane == 0)
        nx = atomicAdd (next_piece, 1u);
      const long item_end = min ((piece + 1) * PCS_FAST_GRAB, n_items);
      for (long item = piece * PCS_FAST_GRAB; item < item_end; item++)
    {
      const long site = LISTED ? (long) deep_list[item] : item;
      if (skip && skip[site])
        continue;               // (pcs_heavy_kernel listed the column for the beam search before this kernel started)
      const int dom = dom_of[site];
      const int chrom = chrom_of[site] & 3;
      int decided = 1;          // 1: written here, 0: left to the beam
      int n_unset = 0;          // samples the shortcut could not settle (left to the beam: a measure of its work)
      int my_call[NCH], npass = 0, ac_dom = 0, type = 0;
#pragma unroll
      for (int c = 0; c < NCH; c++)
        my_call[c] = PCS_NG;
      auto likelihoods = [&] (const double *tb, const int (&rr)[PCS_NA], const int tt, double (&lkc)[PCS_NG], int &bestc, double &marginc)
      {
        double coef = pc_factln (tb, tt);
#pragma unroll
        for (int a = 0; a < PCS_NA; a++)
          coef -= pc_factln (tb, rr[a]);
        // scale = min (depth, 100) * norm (1 in pass 1), at least 10: an integer 10 .. 100, the row of the ta table
        const int sci = (tt < 10 ? 10 : (tt > 100 ? 100 : tt)) - 10;
        const uint32_t *tl = ta_lo + (dom * PCS_TA_SCALES + sci) * PCS_NG;
        const uint16_t *th = ta_hi + (dom * PCS_TA_SCALES + sci) * PCS_NG;
        double mx = -1e100;
#pragma unroll
        for (int g = 0; g < PCS_NG; g++)
          {
            lkc[g] = 0.0;
            if (g < G)
              {
                const uint32_t w4 = tl[g], w2 = th[g];
                int tot_a = 0, tot_tot = 0;
                double cf = coef, l = 0.0;
#pragma unroll
                for (int a = 0; a < PCS_NA; a++)
                  {
                    const int ta = (int) ((a < 4 ? (w4 >> (8 * a)) : (w2 >> (8 * (a - 4)))) & 0xFFu);
                    tot_a += ta;
                    tot_tot += ta + rr[a];
                    cf -= pc_factln (tb, ta - 1);
                    l += pc_factln (tb, ta + rr[a] - 1);
                  }
                cf += pc_factln (tb, tot_a - 1);
                l += cf;
                l -= pc_factln (tb, tot_tot - 1);
                lkc[g] = l;
                if (l > mx)
                  {
                    bestc = g;
                    mx = l;
                  }
              }
          }
        // initial_p: the margin of the best genotype over every other one (pecaller.c:2494-2504)
        marginc = 1e100;
#pragma unroll
        for (int g = 0; g < PCS_NG; g++)
          if (g < G && g != bestc)
            {
              const double dlt = mx - lkc[g];
              marginc = (dlt < marginc) ? dlt : marginc;
            }
      };
      if (dom > 3)
        type = -1;              // the reference skips the column (pecaller.c:1208, 1718)
      else if constexpr (NCH > 2)
        {
          // More than 128 samples: the same test and the same small beam, a chunk of 64 samples at a time.  The likelihoods of four or
          // eight samples per lane do not fit the registers, and the small beam needs few of them: every sample's likelihood of the
          // reference homozygote, its margin and its verdict stay in registers, the 14 likelihoods of an UNSETTLED sample (and its
          // deletion / insertion reads) are parked in the wave's LDS as they are computed -- up to PcsMini::MAXU of them; a column with
          // more, like any column the small beam hands back and a column too deep for the table's head, is listed for pcs_call_kernel,
          // which is the whole caller.
          typedef PcsMini < NCH > MINI;
          static_assert (PCS_FAST_WAVE_BYTES_OF (NCH) == MINI::MAXU * 128 + (int) sizeof (typename MINI::Sord) * 64 * NCH, "the wave's bytes: the parked likelihoods, the order");
          bool too_deep = false;
          int tsum = 0, sample_count = 0;
#pragma unroll
          for (int c = 0; c < NCH; c++)
            {
              const bool have = 64 * c + lane < N;
              int tt = 0, r5 = 0;
              if (have)
                {
                  const uint16_t *p = reads + (site * N + 64 * c + lane) * PCS_NA;
                  tt = (int) p[0] + (int) p[1] + (int) p[2] + (int) p[3] + (int) p[4];
                  r5 = (int) p[5];
                }
              too_deep = too_deep || (tt + r5 + 6 * 100 + 1 >= TABN);
              tsum += tt;
              sample_count += (int) __popcll (__ballot (have && tt >= 8));
            }
          for (int o = 32; o; o >>= 1)
            tsum += __shfl_xor (tsum, o);
          const double average_depth = (double) tsum / (double) N;
          bool bad_base = average_depth < 8;
          if (sample_count < (double) 0.5 * N && chrom != 2)
            bad_base = true;
          if (!bad_base)
            {
              if (__any (too_deep))
                {
                  decided = 0;
                  n_unset = 64;
                }
              else
                {
                  double *stash = (double *) w_wave;            // [MAXU][16]: 14 likelihoods, {r4, r5}, {sample, -}
                  typename MINI::Sord * w_sord = (typename MINI::Sord *) (w_wave + MINI::MAXU * 128);
                  double lk_dom[NCH], margin[NCH];
                  bool deep[NCH], ok[NCH];
                  int best[NCH], n_deep_s = 0;
#pragma unroll
                  for (int c = 0; c < NCH; c++)
                    {
                      lk_dom[c] = 0.0;
                      margin[c] = 0.0;
                      deep[c] = false;
                      ok[c] = true;
                      best[c] = PCS_NG;
                    }
                  pcs_sync ();          // (the previous column's beam has read its stash)
                  // (a loop that is not unrolled -- eight copies of the likelihoods would not fit the instruction cache -- whose chunk's
                  // results go to their registers through selects: an array indexed by the loop variable would live in scratch)
#pragma unroll 1
                  for (int c = 0; c < NCH; c++)
                    {
                      int rr[PCS_NA];
                      const bool have = 64 * c + lane < N;
#pragma unroll
                      for (int a = 0; a < PCS_NA; a++)
                        rr[a] = have ? (int) reads[(site * N + 64 * c + lane) * PCS_NA + a] : 0;
                      const int tt = rr[0] + rr[1] + rr[2] + rr[3] + rr[4];
                      const bool dp = have && tt > md;
                      bool okc = true;
                      double lkc[PCS_NG], mg = 0.0, lkd = 0.0;
                      int bc = PCS_NG;
#pragma unroll
                      for (int g = 0; g < PCS_NG; g++)
                        lkc[g] = 0.0;
                      if (dp)
                        {
                          likelihoods (tab, rr, tt, lkc, bc, mg);
                          okc = bc == dom && mg > 2.31;
#pragma unroll
                          for (int g = 0; g < 4; g++)
                            lkd = (g == dom) ? lkc[g] : lkd;
                        }
                      const unsigned long long um = __ballot (dp && !okc);
                      if (dp && !okc)
                        {
                          const int slot = n_unset + (int) __popcll (um & ((1ull << lane) - 1ull));
                          if (slot < MINI::MAXU)
                            {
                              double *sl = stash + 16 * slot;
#pragma unroll
                              for (int g = 0; g < PCS_NG; g++)
                                sl[g] = lkc[g];
                              ((int *) (sl + 14))[0] = rr[4];
                              ((int *) (sl + 14))[1] = rr[5];
                              ((int *) (sl + 15))[0] = 64 * c + lane;
                            }
                        }
#pragma unroll
                      for (int cc = 0; cc < NCH; cc++)
                        {
                          const bool here = cc == c;
                          my_call[cc] = (here && dp) ? dom : my_call[cc];
                          lk_dom[cc] = here ? lkd : lk_dom[cc];
                          margin[cc] = here ? mg : margin[cc];
                          deep[cc] = here ? dp : deep[cc];
                          ok[cc] = here ? okc : ok[cc];
                          best[cc] = here ? bc : best[cc];
                        }
                      n_unset += (int) __popcll (um);
                      n_deep_s += (int) __popcll (__ballot (dp));
                    }
                  if (n_unset == 0)
                    {
                      npass = 1;
                      if (1.0 >= P.threshold)
                        ac_dom = n_deep_s * (P.haploid ? 1 : 2);
                    }
                  else
                    {
                      if (n_unset <= MINI::MAXU)
                        {
                          pcs_sync ();
                          const int nst = n_unset;
                          auto stage = [&] (int who, int &r4, int &r5)->const double *
                          {
                            int slot = 0;
                            for (int q = 1; q < nst; q++)
                              slot = (((const int *) (stash + 16 * q + 15))[0] == who) ? q : slot;
                            const int *ri = (const int *) (stash + 16 * slot + 14);
                            r4 = ri[0];
                            r5 = ri[1];
                            return stash + 16 * slot;
                          };
                          int fc[NCH];
                          double fp[NCH];
#pragma unroll
                          for (int c = 0; c < NCH; c++)
                            {
                              fc[c] = PCS_NG;
                              fp[c] = 1.0;
                            }
                          const int site_hap = P.haploid | ((chrom_of[site] >> 4) & 1);
                          if (pcs_mini_core < NCH > (P, lk_dom, deep, ok, best, margin, dom, site_hap, w_sord, stage, lane, fc, fp))
                            {
                              auto rd = [&] (const int c, int (&rc)[PCS_NA], int &totc)
                              {
                                const bool have = 64 * c + lane < N;
#pragma unroll
                                for (int a = 0; a < PCS_NA; a++)
                                  rc[a] = have ? (int) reads[(site * N + 64 * c + lane) * PCS_NA + a] : 0;
                                totc = rc[0] + rc[1] + rc[2] + rc[3] + rc[4];
                              };
                              pcs_write_site_rd < NCH > (P, site, lane, dom, chrom, rd, average_depth, fc, fp, 1, call, post_out, type_out, allele_count, n_pass,
                                                         denovo_out);
                              continue;
                            }
                        }
                      decided = 0;
                    }
                }
            }
        }
      else
        {
          int r[NCH][PCS_NA], tot[NCH];
          bool too_deep = false;
          int tsum = 0, sample_count = 0;
#pragma unroll
          for (int c = 0; c < NCH; c++)
            {
              const bool have = 64 * c + lane < N;
#pragma unroll
              for (int a = 0; a < PCS_NA; a++)
                r[c][a] = have ? (int) reads[(site * N + 64 * c + lane) * PCS_NA + a] : 0;
              tot[c] = r[c][0] + r[c][1] + r[c][2] + r[c][3] + r[c][4];
              too_deep = too_deep || (tot[c] + r[c][5] + 6 * 100 + 1 >= TABN);
              tsum += tot[c];
              sample_count += (int) __popcll (__ballot (have && tot[c] >= 8));
            }
          if (!LISTED && __any (too_deep))
            {
              // (a sample too deep for the table's head: the column waits for the form with the whole table; its posteriors read 1
              // until then, so that pcs_sparse_kernel does not take what an earlier call left there)
              if (lane == 0)
                deep_list[atomicAdd (n_deep, 1u)] = (unsigned) site;
#pragma unroll
              for (int c = 0; c < NCH; c++)
                if (64 * c + lane < N)
                  post_out[site * N + 64 * c + lane] = 1.0;
              continue;
            }
          for (int o = 32; o; o >>= 1)
            tsum += __shfl_xor (tsum, o);
          const double average_depth = (double) tsum / (double) N;
          bool bad_base = average_depth < 8;
          if (sample_count < (double) 0.5 * N && chrom != 2)
            bad_base = true;
          if (!bad_base)
            {
              bool deep[NCH], ok[NCH];
              double lk[NCH][PCS_NG], margin[NCH];
              int best[NCH];
              bool all_ok = true;
              int n_deep_s = 0;
#pragma unroll
              for (int c = 0; c < NCH; c++)
                {
                  deep[c] = 64 * c + lane < N && tot[c] > md;
                  ok[c] = true;
                  margin[c] = 0.0;
                  best[c] = PCS_NG;
#pragma unroll
                  for (int g = 0; g < PCS_NG; g++)
                    lk[c][g] = 0.0;
                  if (deep[c])
                    {
                      likelihoods (tab, r[c], tot[c], lk[c], best[c], margin[c]);
                      // best genotype the reference homozygote, and its margin above 2.31
                      ok[c] = best[c] == dom && margin[c] > 2.31;
                      my_call[c] = dom;
                    }
                  all_ok = all_ok && __all (ok[c]);
                  n_deep_s += (int) __popcll (__ballot (deep[c]));
                }
              if (all_ok)
                {
                  npass = 1;
                  // Allele_Counts of the row (pecaller.c:1575-1597): every confident call adds its alleles; the posterior is 1
                  if (1.0 >= P.threshold)
                    ac_dom = n_deep_s * (P.haploid ? 1 : 2);
                }
              else
                {
                  // a few unsettled samples: the small beam, here and now (pcs_mini_beam); anything else is listed for pcs_call_kernel
                  int fc[NCH];
                  double fp[NCH];
#pragma unroll
                  for (int c = 0; c < NCH; c++)
                    {
                      fc[c] = PCS_NG;
                      fp[c] = 1.0;
                    }
                  const int site_hap = P.haploid | ((chrom_of[site] >> 4) & 1);
                  if (pcs_mini_beam < NCH > (P, lk, r, deep, ok, best, margin, dom, site_hap, w_sord, w_like, lane, fc, fp))
                    {
                      pcs_write_site < NCH > (P, site, lane, dom, chrom, r, tot, average_depth, fc, fp, 1, call, post_out, type_out, allele_count, n_pass, denovo_out);
                      continue;
                    }
                  decided = 0;
#pragma unroll
                  for (int c = 0; c < NCH; c++)
                    n_unset += (int) __popcll (__ballot (!ok[c]));
                }
            }
        }
      if (decided)
        {
#pragma unroll
          for (int c = 0; c < NCH; c++)
            if (64 * c + lane < N)
              {
                call[site * N + 64 * c + lane] = (int8_t) my_call[c];
                post_out[site * N + 64 * c + lane] = 1.0;
              }
          if (lane < PCS_NA)
            allele_count[site * PCS_NA + lane] = (type == 0 && lane == dom) ? ac_dom : 0;
          if (lane == 0)
            {
              type_out[site] = (int8_t) type;
              n_pass[site] = (int8_t) npass;
              denovo_out[site] = 0;
            }
        }
      else
        {
          // listed for the beam search, in the part of the list that goes with the number of samples that were not settled
          const int ns = n_unset;
          const int b = ns >= 20 ? 3 : ns >= 8 ? 2 : ns >= 3 ? 1 : 0;
          if (lane == 0)
            slow_list[(size_t) b * (size_t) n_sites + atomicAdd (&n_slow[b], 1u)] = (unsigned) site;
        }
    }
      piece = first_free + (long) (unsigned) __shfl ((int) nx, 0);
    }
}

// The columns whose beam search is long, found BEFORE anything else runs.  A launch of the beam search ends with its slowest column, a
// variant column in which twenty or more samples carry the variant takes one wave 4 to 7 ms, and such a column of the LAST chunk used to
// be listed when the last shortcut kernel ended: the run's last 8 ms were a few waves finishing them on an otherwise idle chip
// (profiles/r03_pecall_timeline.txt).  This pass reads every column once (lane = sample, 12 bytes each), counts the samples with three or
// more reads that are not the reference base's -- and at least an eighth of their depth -- and lists the columns with `thr` or more of
// them by that count (the list's PCS_BUCKETS parts, as the shortcut kernel files its own); flag[site] = 1 tells the shortcut kernel to
// leave the column alone.  pcs_call_kernel is the whole caller and takes any column, so the choice only moves work: the listed columns'
// beam searches start at once, beside the shortcut kernels of all chunks.  (Any sample count; beyond 128 samples the threshold is
// PECALL_HEAVY_MIN_WIDE: the small beam settles the columns whose variant reads are errors, and a beam search is expensive there.)
__global__ __launch_bounds__ (256) void pcs_heavy_kernel (const uint16_t * reads, const uint8_t * dom_of, long n_sites, int N, int thr, unsigned *list,
                                                          unsigned *n_list, uint8_t * flag)
{
  const int lane = threadIdx.x & 63;
  const long wave = ((long) blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = ((long) gridDim.x * blockDim.x) >> 6;
  for (long site = wave; site < n_sites; site += n_waves)
    {
      const int dom = dom_of[site];
      if (dom > 3)
        continue;
      int c = 0;
      for (int s0 = 0; s0 < N; s0 += 64)         // (a lane per sample, 64 at a time)
        {
          int alt = 0, tot = 0;
          if (s0 + lane < N)
            {
              const uint16_t *r = reads + (site * N + s0 + lane) * PCS_NA;
              int ref = 0;
#pragma unroll
              for (int a = 0; a < PCS_NA; a++)
                {
                  const int v = (int) r[a];
                  tot += v;
                  ref = a == dom ? v : ref;
                }
              alt = tot - ref;
            }
          c += (int) __popcll (__ballot (alt >= 3 && 8 * alt >= tot));
        }
      if (c < thr)
        continue;
      if (lane == 0)
        {
          const int b = c < 12 ? 0 : c < 20 ? 1 : c < 32 ? 2 : 3;
          list[(size_t) b * (size_t) n_sites + atomicAdd (&n_list[b], 1u)] = (unsigned) site;
          flag[site] = 1;
        }
    }
}

// The posteriors that are not exactly 1, column by column: nearly every column of real data is settled by the shortcut with posterior
// 1 for every sample, and a caller that prints them (%g) or thresholds them needs the others only -- 512 of the 606 bytes a 64-sample
// column sends back are these doubles.  One wave per column of [0, n_items) (or per column of `list`): if any sample's posterior
// differs from 1, the column takes the next free row of `rows` (its samples' posteriors) and its number goes to `cols`, in no
// particular order (the host sorts the few thousand column numbers).  `bias` is added to the column number (the chunk's offset).
__global__ __launch_bounds__ (256) void pcs_sparse_kernel (const double *post, long n_items, const unsigned *list, const unsigned *n_list, int N, unsigned bias,
                                                           unsigned *cols, double *rows, unsigned long long cap, unsigned long long *n_rows)
{
  const int lane = threadIdx.x & 63;
  const long wave = ((long) blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = ((long) gridDim.x * blockDim.x) >> 6;
  const long n = list ? (long) *n_list : n_items;
  for (long it = wave; it < n; it += n_waves)
    {
      const long site = list ? (long) list[it] : it;
      bool any = false;
      for (int i = lane; i < N; i += 64)
        any = any || post[site * N + i] != 1.0;
      if (!__any (any))
        continue;
      unsigned long long row = 0ull;
      if (lane == 0)
        row = atomicAdd (n_rows, 1ull);
      row = (unsigned long long) __shfl ((long long) row, 0);
      if (row >= cap)
        continue;               // (counted all the same: the host sees that the list was too short)
      if (lane == 0)
        cols[row] = (unsigned) site + bias;
      for (int i = lane; i < N; i += 64)
        rows[row * (unsigned long long) N + i] = post[site * N + i];
    }
}
