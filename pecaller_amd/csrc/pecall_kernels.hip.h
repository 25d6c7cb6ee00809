// pecall_kernels.hip.h -- PECaller's per-(site, sample) Dirichlet-multinomial genotype log-likelihood on gfx950.
//
// fill_sample_like, src/pecaller.c:2448-2507, with the per-sample set-up of src/pecaller.c:1230-1260 (tot over
// A,C,G,T,Del only; multinomial coefficient over all six counts).  fp64, -ffp-contract=off, the reference's order of
// additions.  ln n! comes from the reference's own table formula (factln/exactfactln/gammln, pecaller.c:3163-3214),
// built on the host with libm and staged in LDS (10001 doubles = 80 KB of the CU's 160 KB).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define PC_ALLELES 6
#define PC_MAX_GEN 14
#define PC_TABLE 10001
#define PC_BLOCK 1024

__device__ __forceinline__ double pc_gammln (double xx)        // pecaller.c:3163-3181
{
  const double cof[6] = { 76.18009173, -86.50532033, 24.01409822, -1.231739516, 0.120858003e-2, -0.536382e-5 };
  double x = xx - 1.0;
  double tmp = x + 5.5;
  tmp -= (x + 0.5) * log (tmp);
  double ser = 1.0;
  for (int j = 0; j <= 5; j++)
    {
      x += 1.0;
      ser += cof[j] / x;
    }
  return -tmp + log (2.50662827465 * ser);
}

__device__ __forceinline__ double pc_factln (const double *tab, int n)        // pecaller.c:3198-3214
{
  if (n <= 1)
    return 0.0;
  if (n <= 10000)
    return tab[n];
  return pc_gammln (n + 1.0);
}

// one thread per (site, sample); like[.][14], best genotype and the margin to the runner-up
__global__ __launch_bounds__ (PC_BLOCK) void pc_site_like_kernel (const uint16_t * reads, const double *alpha_mean, const double *g_tab,
                                                                  long n_items, int indiv, int max_gen, int min_depth, double norm,
                                                                  double *like, int8_t * best_out, double *margin_out)
{
  extern __shared__ double tab[];
  for (int i = threadIdx.x; i < PC_TABLE; i += blockDim.x)
    tab[i] = g_tab[i];
  __syncthreads ();
  for (long it = (long) blockIdx.x * blockDim.x + threadIdx.x; it < n_items; it += (long) gridDim.x * blockDim.x)
    {
      const long site = it / indiv;
      const uint16_t *r16 = reads + it * PC_ALLELES;
      int r[PC_ALLELES];
      for (int a = 0; a < PC_ALLELES; a++)
        r[a] = r16[a];
      int tot = r[0];
      for (int a = 1; a < PC_ALLELES - 1; a++)
        tot += r[a];
      double coef0 = pc_factln (tab, tot);
      for (int a = 0; a < PC_ALLELES; a++)
        coef0 -= pc_factln (tab, r[a]);
      double *out = like + it * PC_MAX_GEN;
      if (tot > min_depth)
        {
          // scale = minim (1000, maxim (10, minim (tot, 100) * norm)), pecaller.c:2466
          int t100 = (tot < 100) ? tot : 100;
          double sc = (double) t100 * norm;
          sc = (10 > sc) ? 10 : sc;
          sc = (1000 < sc) ? 1000 : sc;
          const double *al = alpha_mean + site * (PC_MAX_GEN * PC_ALLELES);
          double mx = -1e100;
          int best = PC_MAX_GEN;
          double lk[PC_MAX_GEN];
          for (int j = 0; j < max_gen; j++)
            {
              int tot_a = 0, tot_tot = 0;
              double l = 0.0;
              double coef = coef0;
              for (int a = 0; a < PC_ALLELES; a++)
                {
                  double ca = ceil (sc * al[j * PC_ALLELES + a]);
                  int this_alpha = (int) ((1 > ca) ? 1 : ca);
                  tot_a += this_alpha;
                  tot_tot += this_alpha + r[a];
                  coef -= pc_factln (tab, this_alpha - 1);
                  l += pc_factln (tab, this_alpha + r[a] - 1);
                }
              coef += pc_factln (tab, tot_a - 1);
              l += coef;
              l -= pc_factln (tab, tot_tot - 1);
              lk[j] = l;
              out[j] = l;
              if (l > mx)
                {
                  best = j;
                  mx = l;
                }
            }
          for (int j = max_gen; j < PC_MAX_GEN; j++)
            out[j] = 0.0;
          double ip = 1e100;
          for (int j = 0; j < max_gen; j++)
            if (j != best)
              ip = ((mx - lk[j]) < ip) ? (mx - lk[j]) : ip;
          if (best_out)
            best_out[it] = (int8_t) best;
          if (margin_out)
            margin_out[it] = ip;
        }
      else
        {
          for (int j = 0; j < PC_MAX_GEN; j++)
            out[j] = 0.0;
          if (best_out)
            best_out[it] = PC_MAX_GEN;
          if (margin_out)
            margin_out[it] = 0.0;
        }
    }
}
