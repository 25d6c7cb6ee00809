/*
 * pecall_oracle.c -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement of PECaller's per-(site, sample) genotype log-likelihood: fill_sample_like (src/pecaller.c:2448-2507),
 * the per-sample set-up it depends on (src/pecaller.c:1230-1260) and the reference's ln n! (factln / exactfactln /
 * gammln, src/pecaller.c:3163-3214).  Used by tests/ and bench.py's cpu_baseline only; nothing under pecaller_amd/
 * links it.  Built with -ffp-contract=off.
 *
 * Parity pin: tests/golden/pecall_like.npz holds the inputs and outputs of the reference's own fill_sample_like for
 * 4032 (site, pass) records of a 6-sample run, captured from an instrumented scratch build of the reference
 * (tests/golden/make_golden_pecall.py).  tests/test_pecall.py::test_oracle_matches_reference_dump requires equality to the
 * last bit on this host (same formula, same libm).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

#define NO_ALLELES 6
#define MAX_GENOTYPES 14
#define minim(a,b) ((a<b)?a:b)
#define maxim(a,b) ((a>b)?a:b)

static double
gammln (double xx)
{
  double x, tmp, ser;
  static double cof[6] = { 76.18009173, -86.50532033, 24.01409822,
    -1.231739516, 0.120858003e-2, -0.536382e-5
  };
  int j;
  x = xx - 1.0;
  tmp = x + 5.5;
  tmp -= (x + 0.5) * log (tmp);
  ser = 1.0;
  for (j = 0; j <= 5; j++)
    {
      x += 1.0;
      ser += cof[j] / x;
    }
  return -tmp + log (2.50662827465 * ser);
}

static double
exactfactln (int n)
{
  double x = 1.0;
  for (int i = 2; i <= n; i++)
    x *= (double) i;
  return log (x);
}

double
ora_factln (int n)
{
  if (n <= 1)
    return 0.0;
  if (n <= 40)
    return exactfactln (n);
  return gammln (n + 1.0);
}

/* reads[n_sites][indiv][6], alpha[n_sites][14][6] -> like[n_sites][indiv][14], best, margin */
void
ora_site_like (const uint16_t * reads, const double *alpha_all, long n_sites, int indiv, int max_gen, int min_depth_needed,
               double norm, double *like_out, int8_t * best_out, double *margin_out)
{
  for (long s = 0; s < n_sites; s++)
    {
      const double *alpha = alpha_all + s * MAX_GENOTYPES * NO_ALLELES;
      for (int i = 0; i < indiv; i++)
        {
          long it = s * indiv + i;
          int r[NO_ALLELES];
          for (int a = 0; a < NO_ALLELES; a++)
            r[a] = reads[it * NO_ALLELES + a];
          int tot = r[0];
          for (int a = 1; a < NO_ALLELES - 1; a++)
            tot += r[a];
          double sn_coef = ora_factln (tot);
          for (int a = 0; a < NO_ALLELES; a++)
            sn_coef -= ora_factln (r[a]);
          double *like = like_out + it * MAX_GENOTYPES;
          for (int j = 0; j < MAX_GENOTYPES; j++)
            like[j] = 0.0;
          double max = -1e100, scale, coef;
          int best = MAX_GENOTYPES;
          scale = minim (1000, maxim (10, minim (tot, 100) * norm));
          if (tot > min_depth_needed)
            {
              for (int j = 0; j < max_gen; j++)
                {
                  int tot_a = 0, tot_tot = 0;
                  like[j] = 0.0;
                  coef = sn_coef;
                  for (int ii = 0; ii < NO_ALLELES; ii++)
                    {
                      int this_alpha = maxim (1, ceil (scale * alpha[j * NO_ALLELES + ii]));
                      tot_a += this_alpha;
                      tot_tot += this_alpha + r[ii];
                      coef -= ora_factln (this_alpha - 1);
                      like[j] += ora_factln (this_alpha + r[ii] - 1);
                    }
                  coef += ora_factln (tot_a - 1);
                  like[j] += coef;
                  like[j] -= ora_factln (tot_tot - 1);
                  if (like[j] > max)
                    {
                      best = j;
                      max = like[j];
                    }
                }
              double ip = 1e100;
              for (int j = 0; j < max_gen; j++)
                if (j != best)
                  ip = minim (max - like[j], ip);
              if (best_out)
                best_out[it] = (int8_t) best;
              if (margin_out)
                margin_out[it] = ip;
            }
          else
            {
              if (best_out)
                best_out[it] = MAX_GENOTYPES;
              if (margin_out)
                margin_out[it] = 0.0;
            }
        }
    }
}
