// pecall_capi.hip -- C-ABI of the PECaller likelihood kernel (include/pemap_hip.h, pecall_dev_*).
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdarg.h>
#include <math.h>
#include "../../include/pemap_hip.h"
#include "pecall_kernels.hip.h"
#include "pecall_site.hip.h"

static char g_pc_err[512] = "";

struct pecall_dev
{
  int device;
  hipStream_t stream;
  char err[512];
  double *d_tab;
  uint16_t *d_reads;
  double *d_alpha, *d_like, *d_margin;
  int8_t *d_best;
  long cap_items, cap_sites;
  int grid;
  // per-site caller
  double *d_hw;
  int *d_hw_off;
  int hw_indiv;
  char *d_scratch;
  int site_grid;
  uint16_t *d_sreads;
  uint8_t *d_dom, *d_chromy;
  int8_t *d_call, *d_type, *d_npass;
  double *d_post;
  int32_t *d_ac, *d_den;
  long cap_ssites, cap_sitems;
  // pedigree
  int ped_indiv, ped_haploid;
  double denovo_rate;
  int8_t h_dad[PCS_MAXN], h_mom[PCS_MAXN], h_sex[PCS_MAXN];
  uint8_t h_kid_off[PCS_MAXN + 1], h_kid_list[2 * PCS_MAXN];
  int8_t *d_ped;                // dad[64] mom[64] sex[64] kid_off[65 -> 72] kid_list[128]
  short *d_dyad, *d_trio;
  long staged_sites;
  int staged_indiv;
  hipEvent_t ev_site[2];
  unsigned long long *d_next_site;      // work counter of the per-site kernel; behind it the PCS_BUCKETS counts of listed columns
  unsigned *d_slow;             // columns left to the beam search
};

static int pc_fail (pecall_dev * d, const char *fmt, ...)
{
  va_list ap;
  va_start (ap, fmt);
  vsnprintf (d ? d->err : g_pc_err, 512, fmt, ap);
  va_end (ap);
  return 1;
}

#define PCCHK(d, call) do { hipError_t e_ = (call); if (e_ != hipSuccess) return pc_fail (d, "%s:%d: %s -> %s", __FILE__, __LINE__, #call, hipGetErrorString (e_)); } while (0)

// the reference's ln n! table, pecaller.c:3163-3214, evaluated with the host libm as the reference does
static double h_gammln (double xx)
{
  static const double cof[6] = { 76.18009173, -86.50532033, 24.01409822, -1.231739516, 0.120858003e-2, -0.536382e-5 };
  double x = xx - 1.0, tmp = x + 5.5, ser = 1.0;
  tmp -= (x + 0.5) * log (tmp);
  for (int j = 0; j <= 5; j++)
    {
      x += 1.0;
      ser += cof[j] / x;
    }
  return -tmp + log (2.50662827465 * ser);
}

static double h_factln (int n)
{
  if (n <= 1)
    return 0.0;
  if (n <= 40)
    {
      double x = 1.0;
      for (int i = 2; i <= n; i++)
        x *= (double) i;
      return log (x);
    }
  return h_gammln (n + 1.0);
}

extern "C" const char *pecall_dev_last_error (const pecall_dev * d)
{
  return d ? d->err : g_pc_err;
}

extern "C" int pecall_dev_create (pecall_dev ** out, int device_id)
{
  *out = nullptr;
  int n = 0;
  if (hipGetDeviceCount (&n) != hipSuccess || n <= 0)
    return pc_fail (nullptr, "no HIP device visible: this library has no CPU path");
  if (device_id < 0 || device_id >= n)
    return pc_fail (nullptr, "device %d out of range", device_id);
  pecall_dev *d = (pecall_dev *) calloc (1, sizeof (pecall_dev));
  d->device = device_id;
  PCCHK (nullptr, hipSetDevice (device_id));
  hipDeviceProp_t prop;
  PCCHK (nullptr, hipGetDeviceProperties (&prop, device_id));
  if (strncmp (prop.gcnArchName, "gfx950", 6) != 0)
    {
      free (d);
      return pc_fail (nullptr, "device %d is %s: built for gfx950 only", device_id, prop.gcnArchName);
    }
  d->grid = (prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256) * 2;
  PCCHK (nullptr, hipStreamCreateWithFlags (&d->stream, hipStreamNonBlocking));
  double *tab = (double *) malloc (sizeof (double) * PC_TABLE);
  for (int i = 0; i < PC_TABLE; i++)
    tab[i] = h_factln (i);
  PCCHK (nullptr, hipMalloc ((void **) &d->d_tab, sizeof (double) * PC_TABLE));
  PCCHK (nullptr, hipMemcpy (d->d_tab, tab, sizeof (double) * PC_TABLE, hipMemcpyHostToDevice));
  free (tab);
  PCCHK (nullptr, hipFuncSetAttribute ((const void *) pc_site_like_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, PC_TABLE * 8));
  *out = d;
  return 0;
}

extern "C" void pecall_dev_destroy (pecall_dev * d)
{
  if (!d)
    return;
  hipSetDevice (d->device);
  hipStreamSynchronize (d->stream);
  hipFree (d->d_tab);
  hipFree (d->d_reads);
  hipFree (d->d_alpha);
  hipFree (d->d_like);
  hipFree (d->d_margin);
  hipFree (d->d_best);
  hipFree (d->d_hw);
  hipFree (d->d_hw_off);
  hipFree (d->d_scratch);
  hipFree (d->d_sreads);
  hipFree (d->d_dom);
  hipFree (d->d_chromy);
  hipFree (d->d_call);
  hipFree (d->d_type);
  hipFree (d->d_npass);
  hipFree (d->d_post);
  hipFree (d->d_ac);
  hipFree (d->d_den);
  hipFree (d->d_ped);
  hipFree (d->d_dyad);
  hipFree (d->d_trio);
  hipFree (d->d_slow);
  hipFree (d->d_next_site);
  if (d->ev_site[0])
    {
      hipEventDestroy (d->ev_site[0]);
      hipEventDestroy (d->ev_site[1]);
    }
  hipStreamDestroy (d->stream);
  free (d);
}

static int pc_ensure (pecall_dev * d, long n_sites, long n_items)
{
  if (n_items > d->cap_items)
    {
      hipFree (d->d_reads);
      hipFree (d->d_like);
      hipFree (d->d_margin);
      hipFree (d->d_best);
      PCCHK (d, hipMalloc ((void **) &d->d_reads, n_items * PC_ALLELES * sizeof (uint16_t)));
      PCCHK (d, hipMalloc ((void **) &d->d_like, n_items * PC_MAX_GEN * sizeof (double)));
      PCCHK (d, hipMalloc ((void **) &d->d_margin, n_items * sizeof (double)));
      PCCHK (d, hipMalloc ((void **) &d->d_best, n_items));
      d->cap_items = n_items;
    }
  if (n_sites > d->cap_sites)
    {
      hipFree (d->d_alpha);
      PCCHK (d, hipMalloc ((void **) &d->d_alpha, n_sites * PC_MAX_GEN * PC_ALLELES * sizeof (double)));
      d->cap_sites = n_sites;
    }
  return 0;
}

extern "C" int pecall_dev_stage (pecall_dev * d, const uint16_t * reads, const double *alpha_mean, int n_sites, int indiv)
{
  PCCHK (d, hipSetDevice (d->device));
  if (n_sites <= 0 || indiv <= 0)
    return pc_fail (d, "stage: n_sites %d indiv %d", n_sites, indiv);
  long n_items = (long) n_sites * indiv;
  int rc = pc_ensure (d, n_sites, n_items);
  if (rc)
    return rc;
  PCCHK (d, hipMemcpy (d->d_reads, reads, n_items * PC_ALLELES * sizeof (uint16_t), hipMemcpyHostToDevice));
  PCCHK (d, hipMemcpy (d->d_alpha, alpha_mean, (long) n_sites * PC_MAX_GEN * PC_ALLELES * sizeof (double), hipMemcpyHostToDevice));
  return 0;
}

extern "C" int pecall_dev_run (pecall_dev * d, int n_sites, int indiv, int max_gen, int min_depth, double norm, int sync)
{
  PCCHK (d, hipSetDevice (d->device));
  long n_items = (long) n_sites * indiv;
  if (n_items > d->cap_items || n_sites > d->cap_sites)
    return pc_fail (d, "run: more items than staged");
  if (max_gen < 1 || max_gen > PC_MAX_GEN)
    return pc_fail (d, "run: max_gen %d", max_gen);
  hipLaunchKernelGGL (pc_site_like_kernel, dim3 (d->grid), dim3 (PC_BLOCK), PC_TABLE * sizeof (double), d->stream, d->d_reads, d->d_alpha,
                      d->d_tab, n_items, indiv, max_gen, min_depth, norm, d->d_like, d->d_best, d->d_margin);
  PCCHK (d, hipGetLastError ());
  if (sync)
    PCCHK (d, hipStreamSynchronize (d->stream));
  return 0;
}

extern "C" int pecall_dev_collect (pecall_dev * d, int n_sites, int indiv, double *like, int8_t * best, double *margin)
{
  PCCHK (d, hipSetDevice (d->device));
  long n_items = (long) n_sites * indiv;
  PCCHK (d, hipStreamSynchronize (d->stream));
  if (like)
    PCCHK (d, hipMemcpy (like, d->d_like, n_items * PC_MAX_GEN * sizeof (double), hipMemcpyDeviceToHost));
  if (best)
    PCCHK (d, hipMemcpy (best, d->d_best, n_items, hipMemcpyDeviceToHost));
  if (margin)
    PCCHK (d, hipMemcpy (margin, d->d_margin, n_items * sizeof (double), hipMemcpyDeviceToHost));
  return 0;
}

extern "C" int pecall_dev_site_like (pecall_dev * d, const uint16_t * reads, const double *alpha_mean, int n_sites, int indiv,
                                     int max_gen, int min_depth, double norm, double *like, int8_t * best, double *margin)
{
  int rc = pecall_dev_stage (d, reads, alpha_mean, n_sites, indiv);
  if (rc)
    return rc;
  rc = pecall_dev_run (d, n_sites, indiv, max_gen, min_depth, norm, 1);
  if (rc)
    return rc;
  return pecall_dev_collect (d, n_sites, indiv, like, best, margin);
}

// ---- per-site caller (pecall_site.hip.h)

// ln of the exact Hardy-Weinberg probabilities (fill_hardy_weinberg, pecaller.c:2791-2866) for n diploids: row = number of
// minor alleles (0 .. 2n), column = heterozygotes; built with the host libm like the reference's table
static void h_hardy_weinberg (int n, double *m)
{
  const int asize = 2 * n, cols = n + 1;
  for (long x = 0; x < (long) (asize + 1) * cols; x++)
    m[x] = 0.0;
  for (int i = 1; i <= asize; i++)
    {
      double *row = m + (long) i * cols;
      const int Na = 2 * n - i, Nb = i;
      const double p = (double) i / (double) (Na + Nb);
      const int expect = (int) ceil (i * (1.0 - p));
      const int start = ((expect & 1) == (i & 1)) ? expect : expect - 1;
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
  for (long x = 0; x < (long) (asize + 1) * cols; x++)
    m[x] = m[x] > 1e-50 ? log (m[x]) : -5000;
}

static int pcs_ensure (pecall_dev * d, long n_sites, int indiv)
{
  if (indiv != d->hw_indiv)
    {
      hipFree (d->d_hw);
      hipFree (d->d_hw_off);
      d->d_hw = nullptr;
      d->d_hw_off = nullptr;
      int *off = (int *) calloc (indiv + 2, sizeof (int));
      long tot = 0;
      for (int n = 1; n <= indiv; n++)
        {
          off[n] = (int) tot;
          tot += (long) (2 * n + 1) * (n + 1);
        }
      double *hw = (double *) malloc (sizeof (double) * tot);
      for (int n = 1; n <= indiv; n++)
        h_hardy_weinberg (n, hw + off[n]);
      PCCHK (d, hipMalloc ((void **) &d->d_hw, sizeof (double) * tot));
      PCCHK (d, hipMalloc ((void **) &d->d_hw_off, sizeof (int) * (indiv + 2)));
      PCCHK (d, hipMemcpy (d->d_hw, hw, sizeof (double) * tot, hipMemcpyHostToDevice));
      PCCHK (d, hipMemcpy (d->d_hw_off, off, sizeof (int) * (indiv + 2), hipMemcpyHostToDevice));
      free (hw);
      free (off);
      d->hw_indiv = indiv;
    }
  if (!d->d_scratch)
    {
      d->site_grid = d->grid * 4;       // 8 waves per CU; LDS admits 4 resident, the rest queue
      PCCHK (d, hipMalloc ((void **) &d->d_scratch, (size_t) d->site_grid * (2 * PCS_BIG_BYTES + PCS_BIGCAP)));
    }
  long items = n_sites * indiv;
  if (n_sites > d->cap_ssites || items > d->cap_sitems)
    {
      hipFree (d->d_sreads); hipFree (d->d_dom); hipFree (d->d_chromy); hipFree (d->d_call); hipFree (d->d_type);
      hipFree (d->d_npass); hipFree (d->d_post); hipFree (d->d_ac); hipFree (d->d_den); hipFree (d->d_slow);
      PCCHK (d, hipMalloc ((void **) &d->d_slow, (size_t) PCS_BUCKETS * n_sites * sizeof (unsigned)));
      PCCHK (d, hipMalloc ((void **) &d->d_sreads, items * PCS_NA * sizeof (uint16_t)));
      PCCHK (d, hipMalloc ((void **) &d->d_dom, n_sites));
      PCCHK (d, hipMalloc ((void **) &d->d_chromy, n_sites));
      PCCHK (d, hipMalloc ((void **) &d->d_call, items));
      PCCHK (d, hipMalloc ((void **) &d->d_post, items * sizeof (double)));
      PCCHK (d, hipMalloc ((void **) &d->d_type, n_sites));
      PCCHK (d, hipMalloc ((void **) &d->d_npass, n_sites));
      PCCHK (d, hipMalloc ((void **) &d->d_ac, n_sites * PCS_NA * sizeof (int32_t)));
      PCCHK (d, hipMalloc ((void **) &d->d_den, n_sites * sizeof (int32_t)));
      d->cap_ssites = n_sites;
      d->cap_sitems = items;
    }
  return 0;
}

// get_het_alleles, pecaller.c:2191-2245
static void h_het (int g, int *a, int *b, int ref)
{
  static const int ha[6] = { 0, 0, 0, 1, 1, 2 }, hb[6] = { 1, 2, 3, 2, 3, 3 };
  if (g < PCS_NA)
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

// dyad_denovo / trio_denovo as main fills them (pecaller.c:312-374), for the four reference bases
static void h_denovo_tables (int haploid, short *dyad, short *trio)
{
  memset (dyad, 0, sizeof (short) * 4 * 225);
  memset (trio, 0, sizeof (short) * 4 * 3375);
  const int G = haploid ? 6 : PCS_NG;
  for (int r = 0; r < 4; r++)
    for (int i = 0; i < G; i++)
      for (int j = 0; j < G; j++)
        {
          if (haploid)
            {
              dyad[r * 225 + i * 15 + j] = i != j;
              continue;
            }
          int da, db, ka, kb;
          h_het (i, &da, &db, r);
          h_het (j, &ka, &kb, r);
          if (ka != da && ka != db && kb != da && kb != db)
            dyad[r * 225 + i * 15 + j] = 1;
          for (int k = 0; k < G; k++)
            {
              int ma, mb;
              h_het (k, &ma, &mb, r);
              short v;
              if ((ka == ma && (kb == da || kb == db)) || (ka == mb && (kb == da || kb == db)) || (kb == ma && (ka == da || ka == db))
                  || (kb == mb && (ka == da || ka == db)))
                v = 0;          // one allele from each parent
              else if (ka != ma && kb != db && kb != ma && ka != db && ka != mb && kb != da && kb != mb && ka != da)
                v = 2;
              else
                v = 1;
              trio[r * 3375 + (i * 15 + k) * 15 + j] = v;       // [dad][mom][kid]
            }
        }
}

extern "C" int pecall_dev_set_pedigree (pecall_dev * d, int indiv, const int *dad, const int *mom, const int *sex, const int *kid_off,
                                        const int *kid_list, double denovo_rate)
{
  if (!dad)
    {
      d->ped_indiv = 0;
      return 0;
    }
  if (indiv < 1 || indiv > PCS_MAXN)
    return pc_fail (d, "set_pedigree: %d samples (1..%d)", indiv, PCS_MAXN);
  if (!(denovo_rate >= 1e-30))
    return pc_fail (d, "set_pedigree: de-novo mutation rate %g (pecaller.c:381-385)", denovo_rate);
  if (!kid_off || !kid_list || kid_off[0] != 0 || kid_off[indiv] < 0 || kid_off[indiv] > 2 * PCS_MAXN)
    return pc_fail (d, "set_pedigree: kid_off must start at 0 and end at no more than %d parent-child links", 2 * PCS_MAXN);
  for (int i = 0; i < indiv; i++)
    if (kid_off[i + 1] < kid_off[i])
      return pc_fail (d, "set_pedigree: kid_off is not ascending at sample %d", i);
  for (int i = 0; i < kid_off[indiv]; i++)
    if (kid_list[i] < 0 || kid_list[i] >= indiv)
      return pc_fail (d, "set_pedigree: kid_list[%d] = %d is not a sample index (0..%d)", i, kid_list[i], indiv - 1);
  for (int i = 0; i < indiv; i++)
    {
      if (dad[i] >= indiv || mom[i] >= indiv)
        return pc_fail (d, "set_pedigree: parent index out of range for sample %d", i);
      d->h_dad[i] = (int8_t) (dad[i] < 0 ? -1 : dad[i]);
      d->h_mom[i] = (int8_t) (mom[i] < 0 ? -1 : mom[i]);
      d->h_sex[i] = (int8_t) sex[i];
    }
  for (int i = 0; i <= indiv; i++)
    d->h_kid_off[i] = (uint8_t) kid_off[i];
  for (int i = 0; i < kid_off[indiv]; i++)
    d->h_kid_list[i] = (uint8_t) kid_list[i];
  d->ped_indiv = indiv;
  d->denovo_rate = denovo_rate;
  d->ped_haploid = -1;          // tables are made at the next call, for its ploidy
  return 0;
}

static int pcs_ensure_ped (pecall_dev * d, int haploid)
{
  if (!d->d_ped)
    {
      PCCHK (d, hipMalloc ((void **) &d->d_ped, 3 * PCS_MAXN + 72 + 2 * PCS_MAXN));
      PCCHK (d, hipMalloc ((void **) &d->d_dyad, sizeof (short) * 4 * 225));
      PCCHK (d, hipMalloc ((void **) &d->d_trio, sizeof (short) * 4 * 3375));
    }
  if (d->ped_haploid != haploid)
    {
      short *dy = (short *) malloc (sizeof (short) * 4 * 225), *tr = (short *) malloc (sizeof (short) * 4 * 3375);
      h_denovo_tables (haploid, dy, tr);
      PCCHK (d, hipMemcpy (d->d_dyad, dy, sizeof (short) * 4 * 225, hipMemcpyHostToDevice));
      PCCHK (d, hipMemcpy (d->d_trio, tr, sizeof (short) * 4 * 3375, hipMemcpyHostToDevice));
      free (dy);
      free (tr);
      PCCHK (d, hipMemcpy (d->d_ped, d->h_dad, PCS_MAXN, hipMemcpyHostToDevice));
      PCCHK (d, hipMemcpy (d->d_ped + PCS_MAXN, d->h_mom, PCS_MAXN, hipMemcpyHostToDevice));
      PCCHK (d, hipMemcpy (d->d_ped + 2 * PCS_MAXN, d->h_sex, PCS_MAXN, hipMemcpyHostToDevice));
      PCCHK (d, hipMemcpy (d->d_ped + 3 * PCS_MAXN, d->h_kid_off, PCS_MAXN + 1, hipMemcpyHostToDevice));
      PCCHK (d, hipMemcpy (d->d_ped + 3 * PCS_MAXN + 72, d->h_kid_list, 2 * PCS_MAXN, hipMemcpyHostToDevice));
      d->ped_haploid = haploid;
    }
  return 0;
}

// the per-site caller in three steps (host -> device, kernel, device -> host), so that the kernel can be timed on resident
// columns; pecall_dev_call_sites is the three in a row
extern "C" int pecall_dev_sites_stage (pecall_dev * d, const uint16_t * reads, const uint8_t * ref_base, const uint8_t * chrom_type, long n_sites,
                                       int indiv)
{
  PCCHK (d, hipSetDevice (d->device));
  if (n_sites <= 0 || indiv <= 0 || indiv > PCS_MAXN)
    return pc_fail (d, "call_sites: n_sites %ld, indiv %d (1..%d samples per call)", n_sites, indiv, PCS_MAXN);
  int rc = pcs_ensure (d, n_sites, indiv);
  if (rc)
    return rc;
  long items = n_sites * indiv;
  PCCHK (d, hipMemcpyAsync (d->d_sreads, reads, items * PCS_NA * sizeof (uint16_t), hipMemcpyHostToDevice, d->stream));
  PCCHK (d, hipMemcpyAsync (d->d_dom, ref_base, n_sites, hipMemcpyHostToDevice, d->stream));
  if (chrom_type)
    PCCHK (d, hipMemcpyAsync (d->d_chromy, chrom_type, n_sites, hipMemcpyHostToDevice, d->stream));
  else
    PCCHK (d, hipMemsetAsync (d->d_chromy, 0, n_sites, d->stream));
  PCCHK (d, hipStreamSynchronize (d->stream));
  d->staged_sites = n_sites;
  d->staged_indiv = indiv;
  return 0;
}

extern "C" int pecall_dev_sites_run (pecall_dev * d, int haploid, double threshold, double theta, float *kernel_ms)
{
  PCCHK (d, hipSetDevice (d->device));
  const long n_sites = d->staged_sites;
  const int indiv = d->staged_indiv;
  if (n_sites <= 0)
    return pc_fail (d, "sites_run: nothing staged");
  if (!(theta >= 1e-10 && theta <= 0.5))
    return pc_fail (d, "call_sites: theta %g outside [1e-10, 0.5] (pecaller.c:305-309)", theta);
  int rc;
  if (d->ped_indiv && d->ped_indiv != indiv)
    return pc_fail (d, "call_sites: the pedigree was set for %d samples, this call has %d", d->ped_indiv, indiv);
  if (d->ped_indiv && d->denovo_rate > theta)
    return pc_fail (d, "call_sites: de-novo mutation rate %g above theta %g (pecaller.c:381-385)", d->denovo_rate, theta);
  if (d->ped_indiv && (rc = pcs_ensure_ped (d, haploid ? 1 : 0)))
    return rc;
  PcsParams P;
  P.indiv = indiv;
  P.haploid = haploid ? 1 : 0;
  P.max_gen = haploid ? 6 : PCS_NG;     // pecaller.c:326-336
  P.min_depth = haploid ? 1 : 2;
  P.threshold = threshold;
  P.ln_theta = log (theta);
  P.tab = d->d_tab;
  P.hw = d->d_hw;
  P.hw_off = d->d_hw_off;
  P.use_ped = d->ped_indiv ? 1 : 0;
  P.ln_denovo = d->ped_indiv ? log (d->denovo_rate) : 0.0;
  P.dad = d->d_ped;
  P.mom = d->d_ped + PCS_MAXN;
  P.sex = d->d_ped + 2 * PCS_MAXN;
  P.kid_off = (const uint8_t *) d->d_ped + 3 * PCS_MAXN;
  P.kid_list = (const uint8_t *) d->d_ped + 3 * PCS_MAXN + 72;
  P.dyad = d->d_dyad;
  P.trio = d->d_trio;
  long grid = n_sites < d->site_grid ? n_sites : d->site_grid;
  if (!d->ev_site[0])
    {
      PCCHK (d, hipEventCreate (&d->ev_site[0]));
      PCCHK (d, hipEventCreate (&d->ev_site[1]));
    }
  if (!d->d_next_site)
    {
      PCCHK (d, hipMalloc ((void **) &d->d_next_site, 4 * sizeof (unsigned long long)));
      PCCHK (d, hipFuncSetAttribute ((const void *) pcs_fast_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, PCS_FAST_LDS_BYTES));
    }
  PCCHK (d, hipMemsetAsync (d->d_next_site, 0, 4 * sizeof (unsigned long long), d->stream));
  PCCHK (d, hipEventRecord (d->ev_site[0], d->stream));
  // the columns every sample agrees on are settled by pcs_fast_kernel; the rest, listed by it, go through the beam search
  unsigned *n_slow = (unsigned *) (d->d_next_site + 1);
  long fgrid = (n_sites + PCS_FAST_BLOCK / 64 - 1) / (PCS_FAST_BLOCK / 64);
  if (fgrid > d->grid / 2)
    fgrid = d->grid / 2;        // one workgroup per CU: the ln n! table takes half its LDS
  hipLaunchKernelGGL (pcs_fast_kernel, dim3 ((unsigned) fgrid), dim3 (PCS_FAST_BLOCK), PCS_FAST_LDS_BYTES, d->stream, P, d->d_sreads, d->d_dom, d->d_chromy,
                      n_sites, d->d_call, d->d_post, d->d_type, d->d_ac, d->d_npass, d->d_den, d->d_slow, n_slow);
  hipLaunchKernelGGL (pcs_call_kernel, dim3 ((unsigned) grid), dim3 (64), 0, d->stream, P, d->d_sreads, d->d_dom, d->d_chromy, n_sites, d->d_call,
                      d->d_post, d->d_type, d->d_ac, d->d_npass, d->d_den, d->d_scratch, d->d_next_site, d->d_slow, n_slow);
  PCCHK (d, hipGetLastError ());
  PCCHK (d, hipEventRecord (d->ev_site[1], d->stream));
  PCCHK (d, hipStreamSynchronize (d->stream));
  if (kernel_ms)
    PCCHK (d, hipEventElapsedTime (kernel_ms, d->ev_site[0], d->ev_site[1]));
  if (getenv ("PECALL_LIST_STATS"))
    {
      // how many columns the shortcut left to the beam search, by part of the list
      unsigned c[PCS_BUCKETS];
      if (hipMemcpy (c, n_slow, sizeof c, hipMemcpyDeviceToHost) == hipSuccess)
        fprintf (stderr, "[pecall] %ld columns, listed for the beam by unsettled samples <3 / <8 / <20 / more: %u %u %u %u\n", n_sites, c[0], c[1], c[2], c[3]);
    }
  return 0;
}

extern "C" int pecall_dev_sites_collect (pecall_dev * d, int8_t * call, double *posterior, int8_t * site_type, int32_t * allele_count,
                                         int8_t * n_pass, int32_t * denovo)
{
  PCCHK (d, hipSetDevice (d->device));
  const long n_sites = d->staged_sites;
  const long items = n_sites * d->staged_indiv;
  if (n_sites <= 0)
    return pc_fail (d, "sites_collect: nothing staged");
  PCCHK (d, hipMemcpyAsync (call, d->d_call, items, hipMemcpyDeviceToHost, d->stream));
  PCCHK (d, hipMemcpyAsync (posterior, d->d_post, items * sizeof (double), hipMemcpyDeviceToHost, d->stream));
  if (site_type)
    PCCHK (d, hipMemcpyAsync (site_type, d->d_type, n_sites, hipMemcpyDeviceToHost, d->stream));
  if (allele_count)
    PCCHK (d, hipMemcpyAsync (allele_count, d->d_ac, n_sites * PCS_NA * sizeof (int32_t), hipMemcpyDeviceToHost, d->stream));
  if (n_pass)
    PCCHK (d, hipMemcpyAsync (n_pass, d->d_npass, n_sites, hipMemcpyDeviceToHost, d->stream));
  if (denovo)
    PCCHK (d, hipMemcpyAsync (denovo, d->d_den, n_sites * sizeof (int32_t), hipMemcpyDeviceToHost, d->stream));
  PCCHK (d, hipStreamSynchronize (d->stream));
  return 0;
}

extern "C" int pecall_dev_call_sites (pecall_dev * d, const uint16_t * reads, const uint8_t * ref_base, const uint8_t * chrom_type, long n_sites,
                                      int indiv, int haploid, double threshold, double theta, int8_t * call, double *posterior,
                                      int8_t * site_type, int32_t * allele_count, int8_t * n_pass, int32_t * denovo)
{
  if (!(theta >= 1e-10 && theta <= 0.5))
    return pc_fail (d, "call_sites: theta %g outside [1e-10, 0.5] (pecaller.c:305-309)", theta);
  int rc = pecall_dev_sites_stage (d, reads, ref_base, chrom_type, n_sites, indiv);
  if (!rc)
    rc = pecall_dev_sites_run (d, haploid, threshold, theta, nullptr);
  if (!rc)
    rc = pecall_dev_sites_collect (d, call, posterior, site_type, allele_count, n_pass, denovo);
  return rc;
}
