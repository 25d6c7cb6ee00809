/*
 * pecall_site_oracle.c -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement of PECaller's per-site caller, i.e. the body of call_single_base (src/pecaller.c:1207-1691) for one
 * pileup column of INDIV samples, with or without a pedigree (add_denovo 2396-2445, the trio / dyad tables of main 312-374):
 *
 *   site set-up and filters                      src/pecaller.c:1230-1337
 *   pass loop (at most 5 passes)                 src/pecaller.c:1349-1559
 *     Dirichlet means from the integer alphas    1354-1364
 *     per-sample likelihoods, sample order       fill_sample_like, 2448-2507
 *     beam over joint configurations             fill_config_like 2347-2360, clean_config_probs 2248-2344,
 *                                                fill_config_probs 2511-2788 (Hardy-Weinberg prior: fill_hardy_weinberg 2791-2866)
 *     posteriors, calls, stop rule               1423-1471
 *     alpha re-estimation                        1473-1553, check_alpha_sanity 2076-2188, fill_alpha_prior 3043-3139
 *   site classification                          src/pecaller.c:1565-1636
 *
 * Used by tests/ only; nothing under pecaller_amd/ links it.  Built with -ffp-contract=off.
 *
 * Parity pin: tests/golden/pecall_sites.* -- binary pileups of 8 samples and the sorted .base.gz / .snp text the
 * reference itself (oracle/_ref/pecaller, built at -O1 from /root/reference/src/pecaller.c, one worker thread) printed
 * for them (tests/golden/make_golden_pecall_sites.py).  tests/test_pecall_sites.py formats this oracle's calls and
 * posteriors with the reference's "%c\t%g" and requires the same text, site by site.
 *
 * Sorting: the reference sorts configurations and samples with qsort and a comparator that reports ties as equal;
 * glibc 2.35's qsort is a merge sort for these sizes, i.e. stable, and the fixtures were produced with it.  A stable
 * sort is used here.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define NA 6                    /* NO_ALLELES: A C G T Del Ins */
#define NG 14                   /* MAX_GENOTYPES */
#define MAX_CONFIGS 514         /* pecaller.c:1180 */
#define minim(a,b) ((a<b)?a:b)
#define maxim(a,b) ((a>b)?a:b)

double ora_factln (int n);      /* pecall_oracle.c */

typedef struct
{
  double prior, like, post;
  int gcount[NG];
  int acount[NA];
  int hets, homs, no_alleles, no_denovo;
  int8_t *calls;                /* [indiv] */
} Cfg;

typedef struct
{
  int indiv, max_gen, min_depth, haploid;
  double threshold, ln_theta;
  double **hw;                  /* hw[n] = (2n+1) x (n+1) matrix, row-major, n = 1..indiv */
  int acounts[4][NG][NA];       /* allele_counts[ref][genotype][allele], pecaller.c:725-737 */
  int order[4][NG];             /* genotype_order[ref][.], pecaller.c:617-722 */
  /* per-site scratch */
  Cfg *cur, *nxt;
  int8_t *calls_cur, *calls_nxt;
  int cap_cur, cap_nxt;
  int *idx, *tmp;
  int max_list;                 /* longest list fill_config_probs has built (test coverage statistic) */
  int site_hap;                 /* HAPLOID of the column: the run's, or forced for chrY / chrMT in BED guide mode (pecaller.c:955-957) */
  /* pedigree (use_ped = y) */
  int use_ped;
  double ln_denovo;
  int *dad, *mom, *sex;         /* [indiv], -1 = absent */
  int *kid_off, *kid_list;      /* kids of sample i = kid_list[kid_off[i] .. kid_off[i + 1]), in ped-file order */
  short dyad[4][NG + 1][NG + 1];
  short trio[4][NG + 1][NG + 1][NG + 1];
} Caller;

#define AUTO 0
#define CHRX 1
#define CHRY 2
#define CHRMT 3

static void
het_alleles (int g, int *a, int *b, int ref)    /* get_het_alleles, pecaller.c:2191-2245 */
{
  static const int ha[6] = { 0, 0, 0, 1, 1, 2 }, hb[6] = { 1, 2, 3, 2, 3, 3 };
  if (g < NA)
    *a = *b = g;
  else if (g < 12)
    {
      *a = ha[g - 6];
      *b = hb[g - 6];
    }
  else
    {
      *a = ref;
      *b = g == 12 ? 4 : 5;
    }
}

/* ln of the exact Hardy-Weinberg probability of `hets` heterozygotes given i minor alleles among n diploids */
static double *
hardy_weinberg (int n)
{
  const int asize = 2 * n, cols = n + 1;
  double *m = calloc ((size_t) (asize + 1) * cols, sizeof (double));
  for (int i = 1; i <= asize; i++)
    {
      double *row = m + (size_t) i * cols;
      const int Na = 2 * n - i, Nb = i;
      const double p = (double) i / (double) (Na + Nb);
      const int expect = ceil (i * (1.0 - p));
      const int start = ((expect - i) % 2 == 0) ? expect : expect - 1;  /* same parity as i */
      double sum = row[start] = 1.0;
      int nbb = (Nb - start) / 2, naa = (Na - start) / 2;
      for (int nab = start + 2; naa > 0 && nbb > 0; nab += 2, naa--, nbb--)
        {
          row[nab] = row[nab - 2] * 4.0 * ((double) naa * (double) nbb) / ((double) (nab) * (double) (nab - 1.0));
          sum += row[nab];
        }
      nbb = (Nb - start) / 2;
      naa = (Na - start) / 2;
      for (int nab = start - 2; nab >= 0; nab -= 2, naa++, nbb++)
        {
          row[nab] = row[nab + 2] * ((double) (nab + 2.0) * (double) (nab + 1.0)) / ((double) 4.0 * ((double) (naa + 1.0) * (nbb + 1.0)));
          sum += row[nab];
        }
      for (int j = 0; j <= n; j++)
        row[j] /= sum;
    }
  for (size_t x = 0; x < (size_t) (asize + 1) * cols; x++)
    m[x] = m[x] > 1e-50 ? log (m[x]) : -5000;
  return m;
}

void *
ora_caller_create (int indiv, int haploid, double threshold, double theta)
{
  static const int dip[4][NG] = {
    {0, 7, 6, 8, 12, 13, 1, 2, 3, 4, 5, 9, 10, 11},
    {1, 10, 6, 9, 12, 13, 0, 2, 3, 4, 5, 7, 8, 11},
    {2, 7, 9, 11, 12, 13, 0, 1, 3, 4, 5, 6, 8, 10},
    {3, 10, 8, 11, 12, 13, 1, 0, 2, 4, 5, 6, 7, 9}
  };
  static const int hap[4][6] = { {0, 2, 1, 3, 4, 5}, {1, 3, 0, 2, 4, 5}, {2, 0, 1, 3, 4, 5}, {3, 1, 0, 2, 4, 5} };
  Caller *c = calloc (1, sizeof (Caller));
  c->indiv = indiv;
  c->haploid = haploid;
  c->max_gen = haploid ? 6 : NG;        /* pecaller.c:326-336 */
  c->min_depth = haploid ? 1 : 2;
  c->threshold = threshold;
  c->ln_theta = log (theta);
  for (int r = 0; r < 4; r++)
    {
      for (int j = 0; j < c->max_gen; j++)
        c->order[r][j] = haploid ? hap[r][j] : dip[r][j];
      for (int g = 0; g < NG; g++)
        {
          int a, b;
          het_alleles (g, &a, &b, r);
          c->acounts[r][g][a]++;
          if (!haploid)
            c->acounts[r][g][b]++;
        }
    }
  if (!haploid)
    {
      c->hw = calloc (indiv + 1, sizeof (double *));
      for (int n = 1; n <= indiv; n++)
        c->hw[n] = hardy_weinberg (n);
    }
  c->cap_cur = c->max_gen * (MAX_CONFIGS + 1) + 1;
  c->cap_nxt = (c->max_gen + 1) * (MAX_CONFIGS + 2);
  c->cur = calloc (c->cap_cur, sizeof (Cfg));
  c->nxt = calloc (c->cap_nxt, sizeof (Cfg));
  c->calls_cur = calloc ((size_t) c->cap_cur * indiv, 1);
  c->calls_nxt = calloc ((size_t) c->cap_nxt * indiv, 1);
  for (int i = 0; i < c->cap_cur; i++)
    c->cur[i].calls = c->calls_cur + (size_t) i * indiv;
  for (int i = 0; i < c->cap_nxt; i++)
    c->nxt[i].calls = c->calls_nxt + (size_t) i * indiv;
  c->idx = calloc (c->cap_nxt, sizeof (int));
  c->tmp = calloc (c->cap_nxt, sizeof (int));
  return c;
}

void
ora_caller_destroy (void *p)
{
  Caller *c = p;
  if (c->hw)
    for (int n = 1; n <= c->indiv; n++)
      free (c->hw[n]);
  free (c->hw);
  free (c->cur);
  free (c->nxt);
  free (c->calls_cur);
  free (c->calls_nxt);
  free (c->idx);
  free (c->tmp);
  free (c->dad);
  free (c->mom);
  free (c->sex);
  free (c->kid_off);
  free (c->kid_list);
  free (c);
}

int
ora_caller_max_list (void *p)
{
  return ((Caller *) p)->max_list;
}

/* use_ped = y: parents (-1 = none), sex and each sample's kids in ped-file order; denovo_rate = argv[11].
 * The de-novo tables as main builds them (pecaller.c:312-374). */
void
ora_caller_set_ped (void *p, const int *dad, const int *mom, const int *sex, const int *kid_off, const int *kid_list, double denovo_rate)
{
  Caller *c = p;
  const int N = c->indiv;
  c->use_ped = 1;
  c->ln_denovo = log (denovo_rate);
  c->dad = malloc (N * sizeof (int));
  c->mom = malloc (N * sizeof (int));
  c->sex = malloc (N * sizeof (int));
  c->kid_off = malloc ((N + 1) * sizeof (int));
  memcpy (c->dad, dad, N * sizeof (int));
  memcpy (c->mom, mom, N * sizeof (int));
  memcpy (c->sex, sex, N * sizeof (int));
  memcpy (c->kid_off, kid_off, (N + 1) * sizeof (int));
  c->kid_list = malloc ((kid_off[N] + 1) * sizeof (int));
  memcpy (c->kid_list, kid_list, kid_off[N] * sizeof (int));
  memset (c->dyad, 0, sizeof c->dyad);
  memset (c->trio, 0, sizeof c->trio);
  for (int r = 0; r < 4; r++)
    for (int i = 0; i < c->max_gen; i++)
      for (int j = 0; j < c->max_gen; j++)
        {
          if (c->haploid)
            {
              c->dyad[r][i][j] = i != j;
              continue;
            }
          int da, db, ka, kb;
          het_alleles (i, &da, &db, r);
          het_alleles (j, &ka, &kb, r);
          if (ka != da && ka != db && kb != da && kb != db)
            c->dyad[r][i][j] = 1;
          for (int k = 0; k < c->max_gen; k++)
            {
              int ma, mb;
              het_alleles (k, &ma, &mb, r);
              /* one allele from each parent: nothing new; no allele explained by either: two events; else one */
              if ((ka == ma && (kb == da || kb == db)) || (ka == mb && (kb == da || kb == db)) || (kb == ma && (ka == da || ka == db))
                  || (kb == mb && (ka == da || ka == db)))
                c->trio[r][i][k][j] = 0;
              else if (ka != ma && kb != db && kb != ma && ka != db && ka != mb && kb != da && kb != mb && ka != da)
                c->trio[r][i][k][j] = 2;
              else
                c->trio[r][i][k][j] = 1;
            }
        }
}

/* add_denovo, pecaller.c:2396-2445; genotype NG = not called */
static int
add_denovo (const Caller * c, int kid, int dad, int mom, int sex, int chrom, int ref)
{
  if (dad < NG)
    {
      if (mom < NG)
        {
          if (chrom == AUTO)
            return c->trio[ref][dad][mom][kid];
          if (chrom == CHRX)
            return sex == 1 ? c->dyad[ref][mom][kid] : c->trio[ref][dad][mom][kid];
          if (chrom == CHRY)
            return sex == 1 ? c->dyad[ref][dad][kid] : 0;
          if (chrom == CHRMT)
            return c->dyad[ref][mom][kid];
          return 0;
        }
      if (chrom == AUTO || (chrom == CHRX && sex == 2) || (chrom == CHRY && sex == 1))
        return c->dyad[ref][dad][kid];
      return 0;
    }
  if (mom < NG && chrom != CHRY)
    return c->dyad[ref][mom][kid];
  return 0;
}

/* the de-novo events sample `who` takes part in under the calls of one configuration (pecaller.c:2578-2603): as a child,
 * and as a parent of each of its kids.  The reference carries dg / mg over from one kid to the next when a kid lacks that
 * parent (they are initialised once, before the loop). */
static int
denovo_around (const Caller * c, const int8_t * calls, int who, int chrom, int ref)
{
  int n = 0;
  const int j = calls[who];
  if (c->dad[who] >= 0)
    n += add_denovo (c, j, calls[c->dad[who]], c->mom[who] >= 0 ? calls[c->mom[who]] : NG, c->sex[who], chrom, ref);
  else if (c->mom[who] >= 0)
    n += add_denovo (c, j, NG, calls[c->mom[who]], c->sex[who], chrom, ref);
  int dg = NG, mg = NG;
  for (int q = c->kid_off[who]; q < c->kid_off[who + 1]; q++)
    {
      const int kid = c->kid_list[q];
      if (c->dad[kid] >= 0)
        dg = calls[c->dad[kid]];
      if (c->mom[kid] >= 0)
        mg = calls[c->mom[kid]];
      n += add_denovo (c, calls[kid], dg, mg, c->sex[kid], chrom, ref);
    }
  return n;
}

/* the caller's tables, for handing to the device implementation under test */
const double *
ora_caller_hw (void *p, int n)
{
  return ((Caller *) p)->hw[n];
}

static void
cfg_copy (Cfg * d, const Cfg * s, int indiv)
{
  int8_t *keep = d->calls;
  *d = *s;
  d->calls = keep;
  memcpy (d->calls, s->calls, indiv);
}

/* config_alloc, pecaller.c:2987-3027: every sample above the depth floor called homozygous `dom` */
static void
cfg_init (const Caller * c, Cfg * t, int dom, const int *tot)
{
  memset (t->gcount, 0, sizeof t->gcount);
  memset (t->acount, 0, sizeof t->acount);
  t->homs = t->hets = t->no_alleles = 0;
  for (int i = 0; i < c->indiv; i++)
    if (tot[i] > c->min_depth)
      {
        t->calls[i] = (int8_t) dom;
        t->gcount[dom]++;
        t->acount[dom] += c->site_hap ? 1 : 2;
        t->homs++;
        t->no_alleles = 1;
      }
    else
      t->calls[i] = NG;
  t->like = 0;
  t->prior = 0;
  t->post = 1;
  t->no_denovo = 0;
}

static void
cfg_like (const Caller * c, Cfg * t, const int *tot, const double (*like)[NG + 1])      /* fill_config_like */
{
  t->like = 0;
  for (int i = 0; i < c->indiv; i++)
    if (tot[i] > c->min_depth)
      t->like += like[i][t->calls[i]];
  t->post = t->like + t->prior;
}

/* stable sort of cfg[0..n) by post, descending (sort_configs, pecaller.c:2363-2377) */
static void
sort_cfgs (Caller * c, Cfg * cfg, int n, Cfg * scratch)
{
  int *a = c->idx, *b = c->tmp;
  for (int i = 0; i < n; i++)
    a[i] = i;
  for (int w = 1; w < n; w *= 2)
    {
      for (int lo = 0; lo < n; lo += 2 * w)
        {
          int mid = minim (lo + w, n), hi = minim (lo + 2 * w, n), i = lo, j = mid, k = lo;
          while (i < mid && j < hi)
            b[k++] = (cfg[a[j]].post > cfg[a[i]].post) ? a[j++] : a[i++];
          while (i < mid)
            b[k++] = a[i++];
          while (j < hi)
            b[k++] = a[j++];
        }
      int *t = a;
      a = b;
      b = t;
    }
  /* apply the permutation through the scratch array */
  for (int i = 0; i < n; i++)
    cfg_copy (&scratch[i], &cfg[a[i]], c->indiv);
  for (int i = 0; i < n; i++)
    cfg_copy (&cfg[i], &scratch[i], c->indiv);
}

/* clean_config_probs, pecaller.c:2248-2344 */
static int
clean_cfgs (Caller * c, int n, int ref, double ct, const int *tot, const double (*like)[NG + 1])
{
  Cfg *cn = c->cur;
  sort_cfgs (c, cn, n, c->nxt);
  int max = minim (MAX_CONFIGS, n);
  for (int i = 1; i < max; i++)
    if (cn[0].post > cn[i].post + ct)
      max = i;
  int found_hom = 0;
  for (int i = 0; i < max && !found_hom; i++)
    if (cn[i].no_alleles == 1)
      found_hom = 1;
  if (!found_hom)
    {
      int best_hom = 0;
      for (int i = 1; i < NA; i++)
        if (cn[0].acount[i] > cn[0].acount[best_hom])
          best_hom = i;
      if (best_hom > 3)
        best_hom = ref;
      cfg_init (c, &cn[max], best_hom, tot);
      cn[max].prior = 0.0;
      cfg_like (c, &cn[max], tot, like);
      cn[max].post = cn[max].like;
      if (cn[max].post > cn[max - 1].post)
        sort_cfgs (c, cn, max + 1, c->nxt);
      max++;
    }
  return max;
}

/* fill_config_probs, pecaller.c:2511-2788, without the pedigree terms */
static int
expand_cfgs (Caller * c, int n, int who, int ref, int chrom, double thres, const int *reads, const double *slike)
{
  Cfg *cn = c->cur, *nw = c->nxt;
  double best_post = cn[0].post, best_like = cn[0].like;
  int newcount = 0;
  for (int i = 0; i < n; i++)
    {
      int dup = 0;
      for (int ii = 0; ii < i && !dup; ii++)
        {
          dup = 1;
          for (int jj = 0; jj < c->indiv && dup; jj++)
            if (jj != who && cn[i].calls[jj] != cn[ii].calls[jj])
              dup = 0;
        }
      if (dup)
        continue;
      Cfg *old = &cn[i];
      int j = old->calls[who];
      if (j < NG)
        {
          for (int k = 0; k < NA; k++)
            old->acount[k] -= c->acounts[ref][j][k];
          if (j >= NA)
            old->hets--;
          else
            old->homs--;
          if (c->use_ped)
            old->no_denovo -= denovo_around (c, old->calls, who, chrom, ref);
          old->like -= slike[j];
          old->gcount[j]--;
        }
      for (int jj = 0; jj < c->max_gen; jj++)
        {
          j = c->order[ref][jj];
          double templ = old->like + slike[j];
          /* an indel genotype needs three supporting reads (pecaller.c:2622-2625) */
          if ((j == 4 || j == 12) && reads[4] < 3)
            templ -= 1e10;
          if ((j == 13 || j == 5) && reads[5] < 3)
            templ -= 1e10;
          if (!((templ + thres > best_post) || (templ + 0.01 > best_like)))
            continue;
          Cfg *t = &nw[newcount];
          cfg_copy (t, old, c->indiv);
          t->like = templ;
          t->gcount[j]++;
          t->calls[who] = (int8_t) j;
          if (j >= NA)
            t->hets++;
          else
            t->homs++;
          for (int k = 0; k < NA; k++)
            t->acount[k] += c->acounts[ref][j][k];
          t->no_alleles = 0;
          for (int k = 0; k < NA; k++)
            if (t->acount[k] > 0)
              t->no_alleles++;
          if (c->use_ped)
            t->no_denovo += denovo_around (c, t->calls, who, chrom, ref);
          t->prior = 0;
          if (t->no_alleles > 1)
            t->prior = (t->no_alleles - 1) * c->ln_theta;
          if (t->no_denovo > 0)
            t->prior += t->no_denovo * c->ln_denovo;
          if (!c->site_hap && t->no_alleles > 1)
            {
              int major = 0, minor = 0;
              for (int k = 1; k < NA; k++)
                if (t->acount[k] > t->acount[major])
                  major = k;
              for (int k = 0; k < NA; k++)
                if (k != major)
                  minor += t->acount[k];
              major = t->acount[major];
              if (minor > major)
                {
                  int sw = major;
                  major = minor;
                  minor = sw;
                }
              int hets = minim (minor, t->hets);
              int tot_n = (minor + major) / 2;
              if ((minor - hets) % 2 == 1)
                {
                  minor++;
                  major++;
                }
              t->prior += c->hw[tot_n][(size_t) minor * (tot_n + 1) + hets];
            }
          t->post = t->prior + t->like;
          best_like = maxim (t->like, best_like);
          best_post = maxim (t->post, best_post);
          if (t->post + thres > best_post)
            newcount++;
        }
    }
  for (int i = 0; i < newcount; i++)
    cfg_copy (&cn[i], &nw[i], c->indiv);
  if (newcount > c->max_list)
    c->max_list = newcount;
  return newcount;
}

/* fill_alpha_prior, pecaller.c:3043-3139 */
static void
alpha_prior_init (int (*al)[NA], int max_gen, int hom, int het, int ref)
{
  const int hom_err = maxim (1, hom / 300), err = maxim (1, (2 * het) / 300);
  for (int g = 0; g < max_gen; g++)
    {
      if (g < 4)
        for (int j = 0; j < NA; j++)
          al[g][j] = (g == j) ? hom : hom_err;
      else if (g == 4 || g == 5)
        {
          for (int k = 0; k < 4; k++)
            al[g][k] = (k == ref) ? (g == 4 ? hom / 5 : hom) : err;
          al[g][4] = g == 4 ? (4 * hom) / 5 : err;
          al[g][5] = g == 4 ? err : (4 * hom) / 5;
        }
      else if (g < 12)
        {
          int a, b;
          het_alleles (g, &a, &b, ref);
          for (int k = 0; k < NA; k++)
            al[g][k] = err;
          if (a == ref || b == ref)
            {
              al[g][ref] = (51 * het) / 50;
              al[g][a == ref ? b : a] = (49 * het) / 50;
              al[g][4] = maxim (1, het / 20);
            }
          else
            al[g][a] = al[g][b] = het;
        }
      else if (g == 12)
        {
          for (int k = 0; k < NA; k++)
            al[g][k] = err;
          al[g][4] = (4 * het) / 5;
          al[g][ref] = (6 * het) / 5;
        }
      else
        {
          for (int k = 0; k < NA; k++)
            al[g][k] = err;
          al[g][5] = (2 * het) / 5;
          al[g][ref] = (8 * het) / 5;
        }
    }
}

/* check_alpha_sanity, pecaller.c:2076-2188 */
static void
alpha_sanity (int (*al)[NA], int max_gen, int (*first)[NA], int ref, int normal_factor)
{
  double frac[NG][NA];
  for (int i = 0; i < max_gen; i++)
    {
      int tot = 0;
      for (int j = 0; j < NA; j++)
        tot += al[i][j];
      for (int j = 0; j < NA; j++)
        frac[i][j] = (double) al[i][j] / (double) tot;
    }
  for (int i = 0; i < 4; i++)
    {
      int mx = 0, bad = 0;
      for (int j = 1; j < NA; j++)
        if (al[i][j] > al[i][mx])
          mx = j;
      if (mx != i)
        bad = 1;
      else
        for (int j = 0; j < NA; j++)
          if (j != i && frac[i][j] > 0.3)
            bad = 1;
      if (bad)
        memcpy (al[i], first[i], sizeof al[i]);
    }
  if (frac[4][4] - frac[ref][4] < 0.5)
    memcpy (al[4], first[4], sizeof al[4]);
  if (frac[5][5] - frac[ref][5] < -0.1)
    memcpy (al[5], first[5], sizeof al[5]);
  for (int i = NA; i < max_gen; i++)
    {
      int a, b;
      het_alleles (i, &a, &b, ref);
      if (b == ref)
        {
          int t = a;
          a = b;
          b = t;
        }
      if (frac[i][b] - frac[ref][b] < 0.25)
        memcpy (al[i], first[i], sizeof al[i]);
      else
        {
          int bad = 0;
          if (ref == a)
            frac[i][a] -= 0.05;
          else
            frac[i][a] -= maxim (frac[ref][a], 0.05);
          frac[i][b] -= maxim (0.05, frac[ref][b]);
          for (int j = 0; j < NA && !bad; j++)
            if (j != a && j != b)
              if (frac[i][j] > frac[i][a] || frac[i][j] > frac[i][b])
                bad = 1;
          if (bad)
            memcpy (al[i], first[i], sizeof al[i]);
        }
    }
  for (int i = 0; i < max_gen; i++)
    {
      double scale = al[i][0];
      for (int j = 1; j < NA; j++)
        scale += al[i][j];
      scale = (double) normal_factor / scale;
      for (int j = 0; j < NA; j++)
        if (al[i][j] > 1)
          al[i][j] = maxim (1, (int) ceil (scale * (double) al[i][j]));
    }
}

/* One site.  reads[indiv][6]; dom = reference base 0..3; chrom = AUTO / CHRX / CHRY / CHRMT of the contig (pecaller.c:474-482).
 * Out: call[indiv] (0..13, 14 = 'N'), p[indiv], allele_count[6] (Allele_Counts of the .snp row), returns the site type
 * (0 REF, 1 SNP, 2 DEL, 3 INS, 4 LOW, 5 MULTIALLELIC, 6 MESS); *n_pass = passes run; *denovo = d_count of the row. */
static int
call_site (Caller * c, const uint16_t * rd, int dom, int chrom_in, int8_t * call, double *p_out, int *allele_count, int *n_pass, int *denovo)
{
  /* bit 4 of the chromosome byte: the column is called with HAPLOID forced (only the initial allele counts and the
     Hardy-Weinberg term see it: max_gen, the depth floor and allele_counts stay the run's) */
  const int chrom = chrom_in & 3;
  c->site_hap = c->haploid || (chrom_in & 0x10);
  const int N = c->indiv, G = c->max_gen, md = c->min_depth;
  int reads[N][NA], tot[N], initial_call[N], final_call[N];
  double frac[N][NA], coef[N], like[N][NG + 1], initial_p[N], final_p[N], post_prob[N][NG + 1];
  int ord[N];
  memset (frac, 0, sizeof frac);
  memset (like, 0, sizeof like);
  double average_depth = 0;
  for (int i = 0; i < N; i++)
    {
      for (int a = 0; a < NA; a++)
        reads[i][a] = rd[i * NA + a];
      tot[i] = reads[i][0];
      for (int a = 1; a < NA - 1; a++)
        tot[i] += reads[i][a];
      if (tot[i] > 0)
        for (int a = 0; a < NA; a++)
          frac[i][a] = (double) reads[i][a] / (double) tot[i];
      for (int g = 0; g <= NG; g++)
        post_prob[i][g] = 0.0;
      coef[i] = ora_factln (tot[i]);
      for (int a = 0; a < NA; a++)
        coef[i] -= ora_factln (reads[i][a]);
      initial_call[i] = final_call[i] = tot[i] > md ? dom : NG;
      final_p[i] = 1.0;
      initial_p[i] = 0.0;
      average_depth += tot[i];
    }
  average_depth /= (double) N;
  int bad_base = dom > 3 || average_depth < 8;
  int sample_count = 0;
  for (int i = 0; i < N; i++)
    if (tot[i] >= 8)
      sample_count++;
  if (sample_count < (double) 0.5 * N && chrom != CHRY)
    bad_base = 1;
  int al[NG][NA], first[NG][NA];
  double mean[NG][NA], var[NG][NA], wt[NG][NA];
  const int normal_factor = 300;
  int calls_changed = 1, pass = 0, total = 1;
  if (!bad_base)
    {
      cfg_init (c, &c->cur[0], dom, tot);
      alpha_prior_init (al, G, normal_factor, normal_factor / 2, dom);
    }
  else
    {
      for (int i = 0; i < N; i++)
        tot[i] = 0;
      calls_changed = 0;
    }
  double new_norm[6];
  new_norm[0] = new_norm[1] = 1;
  for (int i = 2; i <= 5; i++)
    new_norm[i] = new_norm[i - 1] * 2.5;
  const double ct = 2.3;        /* starting_threshold */
  while (calls_changed && pass < 5)
    {
      pass++;
      for (int g = 0; g < G; g++)
        {
          int myt = 0;
          for (int a = 0; a < NA; a++)
            {
              myt += al[g][a];
              first[g][a] = al[g][a];
            }
          for (int a = 0; a < NA; a++)
            mean[g][a] = (double) al[g][a] / (double) myt;
        }
      /* ---- fill_sample_like */
      const double norm = new_norm[pass];
      for (int i = 0; i < N; i++)
        {
          ord[i] = i;
          double scale = minim (1000, maxim (10, minim (tot[i], 100) * norm));
          if (tot[i] > md)
            {
              double mx = -1e100;
              int best = NG;
              for (int g = 0; g < G; g++)
                {
                  int tot_a = 0, tot_tot = 0;
                  double cf = coef[i];
                  like[i][g] = 0.0;
                  for (int a = 0; a < NA; a++)
                    {
                      int ta = maxim (1, ceil (scale * mean[g][a]));
                      tot_a += ta;
                      tot_tot += ta + reads[i][a];
                      cf -= ora_factln (ta - 1);
                      like[i][g] += ora_factln (ta + reads[i][a] - 1);
                    }
                  cf += ora_factln (tot_a - 1);
                  like[i][g] += cf;
                  like[i][g] -= ora_factln (tot_tot - 1);
                  if (like[i][g] > mx)
                    {
                      best = g;
                      mx = like[i][g];
                    }
                }
              initial_p[i] = 1e100;
              initial_call[i] = best;
              for (int g = 0; g < G; g++)
                if (g != best)
                  initial_p[i] = minim (mx - like[i][g], initial_p[i]);
            }
          else
            {
              initial_p[i] = 0.0;
              initial_call[i] = NG;
            }
        }
      /* samples by margin, descending, stable (sort_compare_sample_pointer) */
      for (int i = 1; i < N; i++)
        {
          int v = ord[i], k = i;
          while (k > 0 && initial_p[v] > initial_p[ord[k - 1]])
            {
              ord[k] = ord[k - 1];
              k--;
            }
          ord[k] = v;
        }
      for (int i = 0; i < total; i++)
        cfg_like (c, &c->cur[i], tot, (const double (*)[NG + 1]) like);
      total = clean_cfgs (c, total, dom, ct, tot, (const double (*)[NG + 1]) like);
      for (int k = 0; k < N; k++)
        {
          const int ind = ord[k];
          if (tot[ind] > md)
            {
              total = expand_cfgs (c, total, ind, dom, chrom, ct, reads[ind], like[ind]);
              total = clean_cfgs (c, total, dom, ct, tot, (const double (*)[NG + 1]) like);
            }
          else
            {
              final_call[ind] = NG;
              for (int g = 0; g < G; g++)
                post_prob[ind][g] = 0.0;
              post_prob[ind][NG] = 1.0;
              for (int i = 0; i < total; i++)
                c->cur[i].calls[ind] = NG;
              final_p[ind] = 1.0;
            }
        }
      Cfg *cn = c->cur;
      const double max_post = cn[0].post;
      double tot_post = 0;
      for (int i = 0; i < total; i++)
        {
          cn[i].post -= max_post;
          cn[i].post = cn[i].post > -40 ? exp (cn[i].post) : 0;
          tot_post += cn[i].post;
        }
      for (int i = 0; i < total; i++)
        cn[i].post /= tot_post;
      for (int ind = 0; ind < N; ind++)
        for (int g = 0; g < G; g++)
          post_prob[ind][g] = 0;
      for (int ind = 0; ind < N; ind++)
        if (tot[ind] > md)
          for (int i = 0; i < total; i++)
            post_prob[ind][cn[i].calls[ind]] += cn[i].post;
      calls_changed = 0;
      for (int ind = 0; ind < N; ind++)
        if (tot[ind] > md)
          {
            int besti = 0;
            for (int g = 1; g < G; g++)
              if (post_prob[ind][g] > post_prob[ind][besti])
                besti = g;
            final_p[ind] = post_prob[ind][besti];
            final_call[ind] = besti;
            if (final_call[ind] != initial_call[ind] || final_p[ind] < c->threshold)
              calls_changed = 1;
          }
      if (N < 4 || pass == 5)
        calls_changed = 0;
      if (calls_changed)
        {
          /* ---- moment-matched re-estimation of the Dirichlet parameters (pecaller.c:1475-1553) */
          memset (mean, 0, sizeof mean);
          memset (var, 0, sizeof var);
          memset (wt, 0, sizeof wt);
          for (int i = 0; i < total; i++)
            for (int ind = 0; ind < N; ind++)
              if (tot[ind] > md)
                for (int a = 0; a < NA; a++)
                  {
                    const int g = cn[i].calls[ind];
                    mean[g][a] += frac[ind][a] * cn[i].post;
                    var[g][a] += (frac[ind][a] * frac[ind][a]) * cn[i].post;
                    wt[g][a] += cn[i].post;
                  }
          for (int g = 0; g < G; g++)
            for (int a = 0; a < NA; a++)
              if (wt[g][a] > 1e-9)
                {
                  mean[g][a] /= wt[g][a];
                  var[g][a] /= wt[g][a];
                  var[g][a] -= mean[g][a] * mean[g][a];
                }
          const double var_eps = 1e-6;
          for (int g = 0; g < G; g++)
            {
              int non_zero_var = 0, this_min = 0, little_up = 0;
              for (int a = 1; a < NA; a++)
                if (mean[g][a] > mean[g][little_up])
                  little_up = a;
              for (int a = 0; a < NA; a++)
                {
                  if (wt[g][a] >= 1.5 && var[g][a] > var_eps * mean[g][a])
                    non_zero_var++;
                  if (mean[g][a] < mean[g][this_min])
                    this_min = a;
                  if (mean[g][a] > var_eps && mean[g][a] < mean[g][little_up])
                    little_up = a;
                }
              int use_first = 1;
              if (non_zero_var > 1)
                {
                  double s0 = 1.0;
                  for (int a = 0; a < NA; a++)
                    if (a != this_min && var[g][a] > var_eps * mean[g][a])
                      s0 *= mean[g][a] * (1.0 - mean[g][a]) / var[g][a];
                  s0 = pow (s0 - 1.0, (double) 1.0 / (double) (non_zero_var - 1.0));
                  s0 = maxim (s0, 1.0 / mean[g][little_up]);
                  if (s0 > 3.0)
                    {
                      use_first = 0;
                      for (int a = 0; a < NA; a++)
                        al[g][a] = maxim (1, (int) ceil (mean[g][a] * s0));
                    }
                }
              if (use_first)
                for (int a = 0; a < NA; a++)
                  al[g][a] = first[g][a];
            }
          alpha_sanity (al, G, first, dom, normal_factor);
        }
      for (int ind = 0; ind < N; ind++)
        initial_call[ind] = final_call[ind];
    }
  (void) initial_p;
  /* ---- calls and site classification (pecaller.c:1565-1636) */
  int this_allele_count[NA] = { 0, 0, 0, 0, 0, 0 };
  const double low_base = maxim (8, 0.4 * average_depth);
  int on_target = 0, off_target = 0, not_low = 0;
  for (int ind = 0; ind < N; ind++)
    if (tot[ind] > md)
      {
        call[ind] = (int8_t) final_call[ind];
        p_out[ind] = final_p[ind];
        if (final_p[ind] >= c->threshold)
          {
            for (int a = 0; a < NA; a++)
              if (c->acounts[dom][final_call[ind]][a])
                {
                  this_allele_count[a] += c->acounts[dom][final_call[ind]][a];
                  on_target += reads[ind][a];
                }
              else if (a != dom || final_call[ind] != NA - 1)
                off_target += reads[ind][a];
            if (tot[ind] > low_base && final_call[ind] != dom)
              not_low++;
          }
      }
    else
      {
        call[ind] = NG;
        p_out[ind] = 1.0;
      }
  int n_all = 0, isdel = 0, isins = 0, type = 0;
  for (int a = 0; a < NA; a++)
    {
      allele_count[a] = this_allele_count[a];
      if (this_allele_count[a] > 0)
        {
          n_all++;
          if (a == 4)
            isdel = 1;
          else if (a == 5)
            isins = 1;
          else if (a != dom)
            type = 1;
        }
    }
  if (n_all > 1 || (n_all > 0 && this_allele_count[dom > 3 ? 0 : dom] < 1))
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
  /* ---- de-novo events among the confident calls (pecaller.c:1650-1671): the row type gets a DENOVO_ prefix */
  *denovo = 0;
  if (type && c->use_ped)
    for (int i = 0; i < N; i++)
      {
        const double fp = tot[i] > md ? final_p[i] : 1.0;
        const int fc = tot[i] > md ? final_call[i] : NG;
        if (fp >= c->threshold)
          {
            int dc = NG, mc = NG;
            const int d = c->dad[i], m = c->mom[i];
            if (d >= 0 && (tot[d] > md ? final_p[d] : 1.0) >= c->threshold)
              dc = tot[d] > md ? final_call[d] : NG;
            if (m >= 0 && (tot[m] > md ? final_p[m] : 1.0) >= c->threshold)
              mc = tot[m] > md ? final_call[m] : NG;
            *denovo += add_denovo (c, fc, dc, mc, c->sex[i], chrom, dom);
          }
      }
  *n_pass = pass;
  return type;
}

/* reads[n_sites][indiv][6], dom[n_sites] (0..3 = A C G T; anything else is a site the reference skips: outputs 'N'/1, type -1),
 * chrom[n_sites] (0 autosome, 1 chrX, 2 chrY, 3 chrMT, + 16 = HAPLOID forced for the column; NULL = all autosomal) -> call[n_sites][indiv], p[n_sites][indiv], type[n_sites], allele_count[n_sites][6], n_pass[n_sites] */
void
ora_call_sites (void *p, const uint16_t * reads, const uint8_t * dom, const uint8_t * chrom, long n_sites, int8_t * call, double *post,
                int8_t * type, int32_t * allele_count, int8_t * n_pass, int32_t * denovo)
{
  Caller *c = p;
  for (long s = 0; s < n_sites; s++)
    {
      int ac[NA] = { 0, 0, 0, 0, 0, 0 }, np = 0, dn = 0;
      if (dom[s] > 3)
        {
          for (int i = 0; i < c->indiv; i++)
            {
              call[s * c->indiv + i] = NG;
              post[s * c->indiv + i] = 1.0;
            }
          type[s] = -1;
        }
      else
        type[s] = (int8_t) call_site (c, reads + (size_t) s * c->indiv * NA, dom[s], chrom ? chrom[s] : 0, call + s * c->indiv,
                                      post + s * c->indiv, ac, &np, &dn);
      for (int a = 0; a < NA; a++)
        allele_count[s * NA + a] = ac[a];
      n_pass[s] = (int8_t) np;
      if (denovo)
        denovo[s] = dn;
    }
}
