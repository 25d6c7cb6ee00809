// pecall_capi.hip -- C-ABI of the PECaller likelihood kernel (include/pemap_hip.h, pecall_dev_*).
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdarg.h>
#include <math.h>
#include <chrono>
#include <vector>
#include <algorithm>
#include "../../include/pemap_hip.h"
#include "pecall_kernels.hip.h"
#include "pecall_site.hip.h"

static char g_pc_err[512] = "";
#define PCS_SLOTS 3             // staging buffers of the seam's pipeline (chunks in flight between the two host copies)
#define PCS_CTRS 6               // 64-bit counters per chunk: [0] the beam search's work counter, [1..2] the PCS_BUCKETS counts of listed columns, [3] columns on the deep list, [4] the shortcut kernel's work counter
#define PCS_CALL_STREAMS 4       // streams the beam searches of consecutive chunks alternate on (each with a quarter of the waves and of the scratch)

// host ranges page-locked by the caller (one table for the library: pemap_capi.hip)
bool pm_host_pin_lookup (const void *p, size_t bytes);
bool pm_host_pin_range (const void *p, size_t bytes, hipStream_t copy_stream);
int pm_host_unpin (const void *host_ptr);
void pm_par_memcpy (char *dst, const char *src, size_t bytes);

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
  int16_t h_dad[PCS_MAXN], h_mom[PCS_MAXN];
  int8_t h_sex[PCS_MAXN];
  uint16_t h_kid_off[PCS_MAXN + 1], h_kid_list[2 * PCS_MAXN];
  int16_t *d_ped;               // int16: dad[MAXN] mom[MAXN] kid_off[MAXN + 8] kid_list[2 MAXN]; then bytes: sex[MAXN]
  int scratch_row;              // calls-row width (64 per chunk of samples) d_scratch was sized for
  uint32_t *d_ta;               // pass 1's integer Dirichlet parameters (pcs_ta_table)
  short *d_dyad, *d_trio;
  long staged_sites;
  int staged_indiv;
  hipEvent_t ev_site[2];
  unsigned long long *d_next_site;      // per chunk: work counter of the per-site kernel; behind it the PCS_BUCKETS counts of listed columns
  int cap_chunks;
  unsigned *d_slow;             // columns left to the beam search
  // the columns pcs_heavy_kernel lists for the beam search before the shortcut kernels start (up to 64 samples): flags, the list in
  // PCS_BUCKETS parts, its counters (the layout of a chunk's), a stream and a share of the scratch of its own
  uint8_t *d_heavy_flag;
  unsigned *d_heavy_list;
  unsigned long long *d_heavy_ctr;      // [1 + cap_chunks][PCS_CTRS]: slot 0 for a whole run's list (resident columns), slot 1 + k for chunk k's (the seam)
  hipStream_t stream_heavy;
  hipEvent_t ev_heavy[2], *ev_heavy_k;  // list made / its beam search done; per chunk: done
  int heavy_min, heavy_min_wide, heavy_grid;
  unsigned *d_deep;             // columns too deep for the head of the ln n! table (per chunk, at the chunk's offset)
  // pecall_dev_call_sites_sparse: the columns with a posterior that is not 1 (pcs_sparse_kernel)
  unsigned *d_sp_cols;
  double *d_sp_rows;
  unsigned long long *d_sp_n;
  unsigned long long *h_ctrs;   // page-locked: the chunks' counters and the list's length on their way to the host ([cap_chunks * PCS_CTRS + 1])
  unsigned long long sp_cap;
  int sp_indiv;
  // the caller in chunks of columns (pcs_run_chunk): the shortcut kernel of chunk k + 1 runs beside the beam search of chunk k, and at
  // the seam (pecall_dev_call_sites) beside the copies of the chunks around them
  long chunk_sites;
  hipStream_t stream_call[PCS_CALL_STREAMS], stream_h2d, stream_d2h;
  hipEvent_t *ev_h2d, *ev_fast, *ev_call, *ev_d2h;      // [cap_chunks]
  char *h_in[PCS_SLOTS], *h_out[PCS_SLOTS];            // pinned staging for callers whose buffers are not pinned
  size_t h_in_bytes, h_out_bytes;
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
  {
    uint32_t *ta = (uint32_t *) malloc (PCS_TA_BYTES);
    pcs_ta_table (ta);
    PCCHK (nullptr, hipMalloc ((void **) &d->d_ta, PCS_TA_BYTES));
    PCCHK (nullptr, hipMemcpy (d->d_ta, ta, PCS_TA_BYTES, hipMemcpyHostToDevice));
    free (ta);
  }
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
  hipFree (d->d_ta);
  hipFree (d->d_dyad);
  hipFree (d->d_trio);
  hipFree (d->d_slow);
  hipFree (d->d_heavy_flag);
  hipFree (d->d_heavy_list);
  hipFree (d->d_heavy_ctr);
  hipFree (d->d_deep);
  hipFree (d->d_sp_cols);
  hipFree (d->d_sp_rows);
  hipFree (d->d_sp_n);
  if (d->h_ctrs)
    hipHostFree (d->h_ctrs);
  hipFree (d->d_next_site);
  if (d->ev_site[0])
    {
      hipEventDestroy (d->ev_site[0]);
      hipEventDestroy (d->ev_site[1]);
    }
  for (int k = 0; k < d->cap_chunks; k++)
    {
      hipEventDestroy (d->ev_h2d[k]);
      hipEventDestroy (d->ev_fast[k]);
      hipEventDestroy (d->ev_call[k]);
      hipEventDestroy (d->ev_heavy_k[k]);
      hipEventDestroy (d->ev_d2h[k]);
    }
  free (d->ev_h2d);
  free (d->ev_fast);
  free (d->ev_call);
  free (d->ev_heavy_k);
  free (d->ev_d2h);
  for (int i = 0; i < PCS_SLOTS; i++)
    {
      if (d->h_in[i])
        hipHostFree (d->h_in[i]);
      if (d->h_out[i])
        hipHostFree (d->h_out[i]);
    }
  if (d->stream_h2d)
    {
      for (int i = 0; i < PCS_CALL_STREAMS; i++)
        hipStreamDestroy (d->stream_call[i]);
      hipStreamDestroy (d->stream_h2d);
      hipStreamDestroy (d->stream_d2h);
      hipStreamDestroy (d->stream_heavy);
      hipEventDestroy (d->ev_heavy[0]);
      hipEventDestroy (d->ev_heavy[1]);
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
  const int row = 64 * ((indiv + 63) / 64 > 4 ? 8 : (indiv + 63) / 64 > 2 ? 4 : (indiv + 63) / 64);
  if (!d->d_scratch || row > d->scratch_row)
    {
      PCCHK (d, hipDeviceSynchronize ());
      hipFree (d->d_scratch);
      d->d_scratch = nullptr;
      d->site_grid = d->grid * 4;       // 8 waves per CU; LDS admits 4 resident, the rest queue
      d->heavy_grid = d->grid;          // (the early beam search of the listed heavy columns: 2 waves per CU, behind the site_grid shares)
      {
        const char *e = getenv ("PECALL_HEAVY_WAVES");  // waves per CU of that launch (1 .. 4)
        const int w = (e && *e) ? atoi (e) : 3;
        d->heavy_grid = d->grid / 2 * (w < 1 ? 1 : w > 4 ? 4 : w);
      }
      PCCHK (d, hipMalloc ((void **) &d->d_scratch, (size_t) (d->site_grid + d->heavy_grid) * (2 * PCS_BIG_BYTES_OF (row) + PCS_BIGCAP)));
      d->scratch_row = row;
    }
  long items = n_sites * indiv;
  if (n_sites > d->cap_ssites || items > d->cap_sitems)
    {
      hipFree (d->d_sreads); hipFree (d->d_dom); hipFree (d->d_chromy); hipFree (d->d_call); hipFree (d->d_type);
      hipFree (d->d_npass); hipFree (d->d_post); hipFree (d->d_ac); hipFree (d->d_den); hipFree (d->d_slow); hipFree (d->d_deep);
      hipFree (d->d_heavy_flag); hipFree (d->d_heavy_list);
      d->d_heavy_flag = nullptr; d->d_heavy_list = nullptr;
      // (nothing dangles if one of the allocations below fails: the next call allocates again, destroy frees nullptr)
      d->d_sreads = nullptr; d->d_dom = nullptr; d->d_chromy = nullptr; d->d_call = nullptr; d->d_type = nullptr;
      d->d_npass = nullptr; d->d_post = nullptr; d->d_ac = nullptr; d->d_den = nullptr; d->d_slow = nullptr; d->d_deep = nullptr;
      d->cap_ssites = 0;
      d->cap_sitems = 0;
      PCCHK (d, hipMalloc ((void **) &d->d_slow, (size_t) PCS_BUCKETS * n_sites * sizeof (unsigned)));
      PCCHK (d, hipMalloc ((void **) &d->d_deep, (size_t) n_sites * sizeof (unsigned)));
      PCCHK (d, hipMalloc ((void **) &d->d_heavy_flag, (size_t) n_sites));
      PCCHK (d, hipMalloc ((void **) &d->d_heavy_list, (size_t) PCS_BUCKETS * n_sites * sizeof (unsigned)));
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
      d->h_dad[i] = (int16_t) (dad[i] < 0 ? -1 : dad[i]);
      d->h_mom[i] = (int16_t) (mom[i] < 0 ? -1 : mom[i]);
      d->h_sex[i] = (int8_t) sex[i];
    }
  for (int i = 0; i <= indiv; i++)
    d->h_kid_off[i] = (uint16_t) kid_off[i];
  for (int i = 0; i < kid_off[indiv]; i++)
    d->h_kid_list[i] = (uint16_t) kid_list[i];
  d->ped_indiv = indiv;
  d->denovo_rate = denovo_rate;
  d->ped_haploid = -1;          // tables are made at the next call, for its ploidy
  return 0;
}

static int pcs_ensure_ped (pecall_dev * d, int haploid)
{
  if (!d->d_ped)
    {
      PCCHK (d, hipMalloc ((void **) &d->d_ped, sizeof (int16_t) * (5 * PCS_MAXN + 8) + PCS_MAXN));
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
      PCCHK (d, hipMemcpy (d->d_ped, d->h_dad, sizeof (int16_t) * PCS_MAXN, hipMemcpyHostToDevice));
      PCCHK (d, hipMemcpy (d->d_ped + PCS_MAXN, d->h_mom, sizeof (int16_t) * PCS_MAXN, hipMemcpyHostToDevice));
      PCCHK (d, hipMemcpy (d->d_ped + 2 * PCS_MAXN, d->h_kid_off, sizeof (uint16_t) * (PCS_MAXN + 1), hipMemcpyHostToDevice));
      PCCHK (d, hipMemcpy (d->d_ped + 3 * PCS_MAXN + 8, d->h_kid_list, sizeof (uint16_t) * 2 * PCS_MAXN, hipMemcpyHostToDevice));
      PCCHK (d, hipMemcpy (d->d_ped + 5 * PCS_MAXN + 8, d->h_sex, PCS_MAXN, hipMemcpyHostToDevice));
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

// the parameter block of a call, after the checks the reference makes on its command line
static int pcs_params (pecall_dev * d, int indiv, int haploid, double threshold, double theta, PcsParams & P)
{
  if (!(theta >= 1e-10 && theta <= 0.5))
    return pc_fail (d, "call_sites: theta %g outside [1e-10, 0.5] (pecaller.c:305-309)", theta);
  int rc;
  if (d->ped_indiv && d->ped_indiv != indiv)
    return pc_fail (d, "call_sites: the pedigree was set for %d samples, this call has %d", d->ped_indiv, indiv);
  if (d->ped_indiv && d->denovo_rate > theta)
    return pc_fail (d, "call_sites: de-novo mutation rate %g above theta %g (pecaller.c:381-385)", d->denovo_rate, theta);
  if (d->ped_indiv && (rc = pcs_ensure_ped (d, haploid ? 1 : 0)))
    return rc;
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
  P.kid_off = (const uint16_t *) (d->d_ped + 2 * PCS_MAXN);
  P.kid_list = (const uint16_t *) (d->d_ped + 3 * PCS_MAXN + 8);
  P.sex = (const int8_t *) (d->d_ped + 5 * PCS_MAXN + 8);
  P.dyad = d->d_dyad;
  P.trio = d->d_trio;
  return 0;
}

// streams, per-chunk events and counters for n_sites columns
static int pcs_ensure_chunks (pecall_dev * d, long n_sites)
{
  if (!d->chunk_sites)
    {
      const char *e = getenv ("PECALL_CHUNK_LOG2");
      int lg = (e && *e) ? atoi (e) : 18;
      if (lg < 8) lg = 8;
      if (lg > 24) lg = 24;
      d->chunk_sites = 1L << lg;
    }
  if (!d->stream_h2d)
    {
      for (int i = 0; i < PCS_CALL_STREAMS; i++)
        // (a stream priority for the beam searches was tried: 64 -> 60 M columns/s, the shortcut kernels wait for them then)
        PCCHK (d, hipStreamCreateWithFlags (&d->stream_call[i], hipStreamNonBlocking));
      // Which streams share one of the runtime's 4 hardware queues (per priority level) is settled when they are made and depends on every
      // stream the process made before: with the early beam search's stream on the queue of the shortcut kernels' stream a resident run of
      // 2 M columns took 23.4 ms instead of 19.2, with a copy stream on a beam-search stream's queue the seam did 25 M columns/s instead
      // of 43 -- the FIRST object of a process had the one, every later object the other (profiles/r04_ab_sweeps.txt; the order the
      // streams are first used in changes nothing).  Streams of another priority take queues of their own: the copies' streams high,
      // the early beam search's low, the shortcut kernels' and the chunks' beam searches' streams at the default.
      int prio_low = 0, prio_high = 0;
      PCCHK (d, hipDeviceGetStreamPriorityRange (&prio_low, &prio_high));
      if (getenv ("PECALL_FLAT_PRIORITIES"))    // (diagnostic: every stream at the default priority, as before round 4)
        prio_low = prio_high = 0;
      PCCHK (d, hipStreamCreateWithPriority (&d->stream_h2d, hipStreamNonBlocking, prio_high));
      PCCHK (d, hipStreamCreateWithPriority (&d->stream_d2h, hipStreamNonBlocking, prio_high));
      PCCHK (d, hipStreamCreateWithPriority (&d->stream_heavy, hipStreamNonBlocking, prio_low));
      PCCHK (d, hipEventCreateWithFlags (&d->ev_heavy[0], hipEventDisableTiming));
      PCCHK (d, hipEventCreateWithFlags (&d->ev_heavy[1], hipEventDisableTiming));
      {
        // PECALL_HEAVY_MIN: samples with variant reads from which a column's beam search is started ahead of the shortcut kernels; 0 = never
        const char *e = getenv ("PECALL_HEAVY_MIN");
        d->heavy_min = (e && *e) ? atoi (e) : 1;
        // ... beyond 128 samples (PECALL_HEAVY_MIN_WIDE): a column's beam search costs tens of milliseconds there and the small beam settles
        // most columns whose only variant reads are errors, so the early start takes the columns with several such samples only
        e = getenv ("PECALL_HEAVY_MIN_WIDE");
        d->heavy_min_wide = (e && *e) ? atoi (e) : 3;
      }
      PCCHK (d, hipFuncSetAttribute ((const void *) pcs_fast_kernel < PC_TABLE, 1 >, hipFuncAttributeMaxDynamicSharedMemorySize, PCS_FAST_LDS_BYTES_OF (PC_TABLE)));
      PCCHK (d, hipFuncSetAttribute ((const void *) pcs_fast_kernel < PC_TABLE, 2 >, hipFuncAttributeMaxDynamicSharedMemorySize, PCS_FAST_LDS_BYTES_OF (PC_TABLE)));
      PCCHK (d, hipFuncSetAttribute ((const void *) pcs_call_kernel < 4 >, hipFuncAttributeMaxDynamicSharedMemorySize, (int) sizeof (PcsShared < 4 >)));
      PCCHK (d, hipFuncSetAttribute ((const void *) pcs_call_kernel < 8 >, hipFuncAttributeMaxDynamicSharedMemorySize, (int) sizeof (PcsShared < 8 >)));
    }
  if (!d->ev_site[0])
    {
      PCCHK (d, hipEventCreate (&d->ev_site[0]));
      PCCHK (d, hipEventCreate (&d->ev_site[1]));
    }
  const int nch = (int) ((n_sites + d->chunk_sites - 1) / d->chunk_sites);
  if (nch > d->cap_chunks)
    {
      PCCHK (d, hipDeviceSynchronize ());
      hipFree (d->d_next_site);
      d->d_next_site = nullptr;
      PCCHK (d, hipMalloc ((void **) &d->d_next_site, (size_t) nch * PCS_CTRS * sizeof (unsigned long long)));
      hipFree (d->d_heavy_ctr);
      d->d_heavy_ctr = nullptr;
      PCCHK (d, hipMalloc ((void **) &d->d_heavy_ctr, (size_t) (nch + 1) * PCS_CTRS * sizeof (unsigned long long)));
      if (d->h_ctrs)
        hipHostFree (d->h_ctrs);
      d->h_ctrs = nullptr;
      PCCHK (d, hipHostMalloc ((void **) &d->h_ctrs, ((size_t) nch * PCS_CTRS + 1) * sizeof (unsigned long long), hipHostMallocDefault));
      d->ev_h2d = (hipEvent_t *) realloc (d->ev_h2d, sizeof (hipEvent_t) * nch);
      d->ev_fast = (hipEvent_t *) realloc (d->ev_fast, sizeof (hipEvent_t) * nch);
      d->ev_call = (hipEvent_t *) realloc (d->ev_call, sizeof (hipEvent_t) * nch);
      d->ev_d2h = (hipEvent_t *) realloc (d->ev_d2h, sizeof (hipEvent_t) * nch);
      d->ev_heavy_k = (hipEvent_t *) realloc (d->ev_heavy_k, sizeof (hipEvent_t) * nch);
      for (int k = d->cap_chunks; k < nch; k++)
        {
          PCCHK (d, hipEventCreateWithFlags (&d->ev_h2d[k], hipEventDisableTiming));
          PCCHK (d, hipEventCreateWithFlags (&d->ev_fast[k], hipEventDisableTiming));
          PCCHK (d, hipEventCreateWithFlags (&d->ev_call[k], hipEventDisableTiming));
          PCCHK (d, hipEventCreateWithFlags (&d->ev_d2h[k], hipEventDisableTiming));
          PCCHK (d, hipEventCreateWithFlags (&d->ev_heavy_k[k], hipEventDisableTiming));
        }
      d->cap_chunks = nch;
    }
  return 0;
}

// Chunk k = columns [off, off + m): its counters, the shortcut kernel (with the small beam) on the object's stream, and the beam search
// of the columns it lists on one of PCS_CALL_STREAMS streams, so that it runs beside the next chunks' shortcut kernels.
// The shortcut kernel has two forms: the head of the ln n! table in LDS (three workgroups per CU), or the whole table (one per CU).
// Which one a column needs depends on its deepest sample.  Nothing here waits for the device: the form with the table's head runs
// over every chunk and puts the columns that are too deep for it -- beyond ~1,400 reads in one sample, rare -- on a list of the chunk;
// the host looks at the lists' lengths once, when all chunks are through, and gives the chunks with a list a second pass
// (whole_table = true: the other form over the listed columns, and the beam search of what that lists).
// (History: a kernel of its own looked for the deepest sample of a chunk first, 0.12 ms and the chunk's 200 MB a second time per
// chunk; before that the host waited for each chunk's depth, which cost the seam half its rate -- the 4-byte copy queued behind
// the chunks' 150 MB copies.)
static int pcs_chunk_reset (pecall_dev * d, const PcsParams & P, int k, long off, long m)
{
  // (the chunk's counters)
  (void) P;
  (void) off;
  (void) m;
  PCCHK (d, hipMemsetAsync (d->d_next_site + (size_t) k * PCS_CTRS, 0, PCS_CTRS * sizeof (unsigned long long), d->stream));
  return 0;
}

static int pcs_heavy_start (pecall_dev * d, const PcsParams & P, long off, long m, int slot, hipEvent_t done, hipStream_t on = nullptr, char *scratch_at = nullptr,
                            long waves = 0);
// heavy: 0 no early beam search; 1 the run made its list already (pcs_heavy_start over all columns): the shortcut kernel passes the flagged
// columns over; 2 the chunk makes its own list here (the seam: columns arrive chunk by chunk) and its results wait for that beam search too
static int pcs_chunk_kernels (pecall_dev * d, const PcsParams & P, int k, long off, long m, bool whole_table, bool sparse = false, int heavy = 0)
{
  const int N = P.indiv;
  unsigned long long *ctr = d->d_next_site + (size_t) k * PCS_CTRS;
  unsigned *n_slow = (unsigned *) (ctr + 1);
  unsigned *slow = d->d_slow + (size_t) PCS_BUCKETS * off;
  unsigned *n_deep = (unsigned *) (ctr + 3), *next_piece = (unsigned *) (ctr + 4);
  unsigned *deep_list = d->d_deep + off;
  const int nch = N <= 64 ? 1 : N <= 128 ? 2 : N <= 256 ? 4 : 8;
  // (second pass: behind the first pass's beam search of the chunk, which shares these counters; the deep list's length stays)
  if (whole_table)
    {
      PCCHK (d, hipStreamWaitEvent (d->stream, d->ev_call[k], 0));
      PCCHK (d, hipMemsetAsync (ctr, 0, 3 * sizeof (unsigned long long), d->stream));
      PCCHK (d, hipMemsetAsync (ctr + 4, 0, sizeof (unsigned long long), d->stream));
    }
  if (heavy == 2 && !whole_table)
    {
      // (on the chunk's own beam-search stream, in front of the search of what the shortcut kernel lists: the chunks' early searches
      // then alternate on PCS_CALL_STREAMS streams like the others -- on the one stream of the resident form they ran one behind the
      // other, 8 x 5 ms, and every chunk's results waited for its own)
      const int rc = pcs_heavy_start (d, P, off, m, 1 + k, d->ev_heavy_k[k], d->stream_call[k % PCS_CALL_STREAMS],
                                      d->d_scratch + (size_t) (k % PCS_CALL_STREAMS) * (size_t) (d->site_grid / PCS_CALL_STREAMS) * (2 * PCS_BIG_BYTES_OF (64 * nch) + PCS_BIGCAP),
                                      d->site_grid / PCS_CALL_STREAMS);
      if (rc)
        return rc;
    }
  if (nch <= 2 || !whole_table)
    {
      // the shortcut kernel: a lane per sample up to 64 samples, two samples per lane up to 128 (round 4); beyond that a chunk of 64
      // samples at a time, the unsettled samples' likelihoods parked in LDS for the small beam: what it cannot write goes to the beam
      // search's list (no second pass with the whole table: too deep = listed)
#define PCS_FAST(TAB_, NCH_, GRID_) hipLaunchKernelGGL (HIP_KERNEL_NAME (pcs_fast_kernel < TAB_, NCH_ >), dim3 ((unsigned) (GRID_)), dim3 (PCS_FAST_BLOCK_OF (TAB_)), \
    PCS_FAST_LDS_BYTES_OF2 (TAB_, NCH_), d->stream, P, d->d_sreads + off * N * PCS_NA, d->d_dom + off, d->d_chromy + off, m, d->d_call + off * N, d->d_post + off * N, \
    d->d_type + off, d->d_ac + off * PCS_NA, d->d_npass + off, d->d_den + off, slow, n_slow, deep_list, n_deep, next_piece, d->d_ta, \
    heavy ? (const uint8_t *) d->d_heavy_flag + off : (const uint8_t *) nullptr)
      if (!whole_table)
        {
          constexpr int B = PCS_FAST_BLOCK_OF (PCS_FAST_TAB);
          long fgrid = (m + B / 64 - 1) / (B / 64);
          // three workgroups of 4 waves per CU (two with two samples per lane: the registers)
          const long cap = nch == 1 ? (long) d->grid / 2 * 3 : (long) d->grid;
          if (fgrid > cap)
            fgrid = cap;
          if (nch == 1)
            PCS_FAST (PCS_FAST_TAB, 1, fgrid);
          else if (nch == 2)
            PCS_FAST (PCS_FAST_TAB, 2, fgrid);
          else if (nch == 4)
            PCS_FAST (PCS_FAST_TAB, 4, fgrid);
          else
            PCS_FAST (PCS_FAST_TAB, 8, fgrid);
        }
      else
        {
          constexpr int B = PCS_FAST_BLOCK_OF (PC_TABLE);
          long fgrid = (m + B / 64 - 1) / (B / 64);
          if (fgrid > d->grid / 2)
            fgrid = d->grid / 2;        // one workgroup per CU: the whole ln n! table takes half its LDS
          if (nch == 1)
            PCS_FAST (PC_TABLE, 1, fgrid);
          else
            PCS_FAST (PC_TABLE, 2, fgrid);
        }
#undef PCS_FAST
    }
  else if (whole_table)
    return 0;                   // (more than 128 samples: no second pass, a column too deep for the table's head was listed for the beam search)
  // (the beam search's kernel takes the columns the shortcut kernel listed, a lane standing for a sample of each chunk of 64)
  PCCHK (d, hipEventRecord (d->ev_fast[k], d->stream));
  // (a chunk lists a few hundred columns for the beam search, a handful of them heavy -- milliseconds on one wave: behind each other
  // on one stream the chunks' searches were the caller's time, 8 x 5.5 ms.  They alternate on PCS_CALL_STREAMS streams.)
  hipStream_t sc = d->stream_call[k % PCS_CALL_STREAMS];
  const long cgrid = d->site_grid / PCS_CALL_STREAMS;
  PCCHK (d, hipStreamWaitEvent (sc, d->ev_fast[k], 0));
  const long grid = m < cgrid ? m : cgrid;
  const int row = 64 * nch;
  char *scratch = d->d_scratch + (size_t) (k % PCS_CALL_STREAMS) * (size_t) cgrid * (2 * PCS_BIG_BYTES_OF (row) + PCS_BIGCAP);
#define PCS_CALL(NCH_, LIST, NLIST) hipLaunchKernelGGL (HIP_KERNEL_NAME (pcs_call_kernel < NCH_ >), dim3 ((unsigned) grid), dim3 (64), sizeof (PcsShared < NCH_ >), sc, P, \
    d->d_sreads + off * N * PCS_NA, d->d_dom + off, d->d_chromy + off, m, d->d_call + off * N, d->d_post + off * N, d->d_type + off, d->d_ac + off * PCS_NA, \
    d->d_npass + off, d->d_den + off, scratch, ctr, LIST, NLIST)
  if (nch == 1)
    PCS_CALL (1, slow, n_slow);
  else if (nch == 2)
    PCS_CALL (2, slow, n_slow);
  else if (nch == 4)
    PCS_CALL (4, slow, n_slow);
  else
    PCS_CALL (8, slow, n_slow); // (257 .. 512 samples: 124 KB of LDS, one wave per CU at a time)
#undef PCS_CALL
  if (heavy == 2)
    PCCHK (d, hipStreamWaitEvent (sc, d->ev_heavy_k[k], 0));    // (the chunk's results are whole when its early beam search is through too)
  if (sparse)
    {
      // behind the beam search, on its stream: the chunk's columns with a posterior that is not 1 (second pass: of the deep list only)
      const long sgrid = whole_table ? d->grid : ((m + 3) / 4 < (long) d->grid * 4 ? (m + 3) / 4 : (long) d->grid * 4);
      hipLaunchKernelGGL (pcs_sparse_kernel, dim3 ((unsigned) (sgrid > 0 ? sgrid : 1)), dim3 (256), 0, sc, d->d_post + off * N, m,
                          whole_table ? (const unsigned *) deep_list : (const unsigned *) nullptr, (const unsigned *) n_deep, N, (unsigned) off,
                          d->d_sp_cols, d->d_sp_rows, d->sp_cap, d->d_sp_n);
    }
  PCCHK (d, hipGetLastError ());
  PCCHK (d, hipEventRecord (d->ev_call[k], sc));
  return 0;
}

// pcs_heavy_kernel over the columns [off, off + m) on the object's stream, then -- on a stream of its own, with scratch of its own --
// the beam search of what it listed, heaviest part first; `done` closes it.  slot: the list's counters (0: a whole run, 1 + k: chunk k).
static int pcs_heavy_start (pecall_dev * d, const PcsParams & P, long off, long m, int slot, hipEvent_t done, hipStream_t on, char *scratch_at, long waves)
{
  hipStream_t hs = on ? on : d->stream_heavy;
  const int N = P.indiv;
  unsigned long long *ctr = d->d_heavy_ctr + (size_t) slot * PCS_CTRS;
  unsigned *n_list = (unsigned *) (ctr + 1);
  unsigned *list = d->d_heavy_list + (size_t) PCS_BUCKETS * off;
  PCCHK (d, hipMemsetAsync (ctr, 0, PCS_CTRS * sizeof (unsigned long long), d->stream));
  PCCHK (d, hipMemsetAsync (d->d_heavy_flag + off, 0, (size_t) m, d->stream));
  long hgrid = (m + 3) / 4;
  if (hgrid > (long) d->grid * 4)
    hgrid = (long) d->grid * 4;
  hipLaunchKernelGGL (pcs_heavy_kernel, dim3 ((unsigned) hgrid), dim3 (256), 0, d->stream, d->d_sreads + off * N * PCS_NA, d->d_dom + off, m, N, N <= 128 ? d->heavy_min : d->heavy_min_wide, list,
                      n_list, d->d_heavy_flag + off);
  PCCHK (d, hipEventRecord (d->ev_heavy[0], d->stream));
  PCCHK (d, hipStreamWaitEvent (hs, d->ev_heavy[0], 0));
  const int row = N <= 64 ? 64 : N <= 128 ? 128 : N <= 256 ? 256 : 512;
  char *scratch = scratch_at ? scratch_at : d->d_scratch + (size_t) d->site_grid * (2 * PCS_BIG_BYTES_OF (row) + PCS_BIGCAP);
  const long hw = waves > 0 ? waves : (long) d->heavy_grid;
  const long grid = m < hw ? m : hw;
#define PCS_HEAVY(NCH_) hipLaunchKernelGGL (HIP_KERNEL_NAME (pcs_call_kernel < NCH_ >), dim3 ((unsigned) grid), dim3 (64), sizeof (PcsShared < NCH_ >), hs, P, \
                      d->d_sreads + off * N * PCS_NA, d->d_dom + off, d->d_chromy + off, m, d->d_call + off * N, d->d_post + off * N, d->d_type + off, \
                      d->d_ac + off * PCS_NA, d->d_npass + off, d->d_den + off, scratch, ctr, (const unsigned *) list, (const unsigned *) n_list)
  if (N <= 64)
    PCS_HEAVY (1);
  else if (N <= 128)
    PCS_HEAVY (2);
  else if (N <= 256)
    PCS_HEAVY (4);
  else
    PCS_HEAVY (8);
#undef PCS_HEAVY
  PCCHK (d, hipGetLastError ());
  PCCHK (d, hipEventRecord (done, hs));
  return 0;
}

// the chunks whose depth asks for the whole table (to be called when the first pass is through on the device): 1 in deep[k]
static int pcs_deep_chunks (pecall_dev * d, const PcsParams & P, int nch, std::vector < char >&deep, int *n_deep)
{
  *n_deep = 0;
  deep.assign ((size_t) nch, 0);
  if (P.indiv > 128)
    return 0;               // (no shortcut kernel, no deep list)
  // (into page-locked memory of the object's own: a pageable target that shares a page with an array the caller registered is refused)
  unsigned long long *c = d->h_ctrs;
  PCCHK (d, hipMemcpy (c, d->d_next_site, (size_t) nch * PCS_CTRS * sizeof (unsigned long long), hipMemcpyDeviceToHost));
  for (int k = 0; k < nch; k++)
    if ((unsigned) c[(size_t) k * PCS_CTRS + 3] > 0u)
      {
        deep[k] = 1;
        (*n_deep)++;
      }
  return 0;
}

extern "C" int pecall_dev_sites_run (pecall_dev * d, int haploid, double threshold, double theta, float *kernel_ms)
{
  PCCHK (d, hipSetDevice (d->device));
  const long n_sites = d->staged_sites;
  const int indiv = d->staged_indiv;
  if (n_sites <= 0)
    return pc_fail (d, "sites_run: nothing staged");
  PcsParams P;
  int rc = pcs_params (d, indiv, haploid, threshold, theta, P);
  if (rc)
    return rc;
  if ((rc = pcs_ensure_chunks (d, n_sites)))
    return rc;
  PCCHK (d, hipEventRecord (d->ev_site[0], d->stream));
  const long C = d->chunk_sites;
  const int k = (int) ((n_sites + C - 1) / C);
  std::vector < char >deep ((size_t) k, 0);
  // the long beam searches first (pcs_heavy_kernel): listed from the resident columns, started on a stream of their own beside everything
  // that follows -- a launch of the beam search ends with its slowest column, and the last chunk's used to be the run's last 8 ms
  const bool heavy = (indiv <= 128 ? d->heavy_min : d->heavy_min_wide) > 0;
  if (heavy)
    if ((rc = pcs_heavy_start (d, P, 0, n_sites, 0, d->ev_heavy[1])))
      return rc;
  for (int pass = 0; pass < 2; pass++)
    {
      // (the counters of all chunks first)
      for (int q = 0; q < k && pass == 0; q++)
        if ((rc = pcs_chunk_reset (d, P, q, (long) q * C, n_sites - (long) q * C < C ? n_sites - (long) q * C : C)))
          return rc;
      for (int q = 0; q < k; q++)
        if (pass == 0 || deep[q])
          if ((rc = pcs_chunk_kernels (d, P, q, (long) q * C, n_sites - (long) q * C < C ? n_sites - (long) q * C : C, pass == 1, false, heavy ? 1 : 0)))
            return rc;
      // (the object's stream ends behind the beam searches: ev_site[1] closes the interval of all streams)
      for (int q = 0; q < k; q++)
        PCCHK (d, hipStreamWaitEvent (d->stream, d->ev_call[q], 0));
      if (heavy)
        PCCHK (d, hipStreamWaitEvent (d->stream, d->ev_heavy[1], 0));
      if (pass == 0)
        {
          PCCHK (d, hipEventRecord (d->ev_site[1], d->stream));
          PCCHK (d, hipStreamSynchronize (d->stream));
          int n_deep = 0;
          if ((rc = pcs_deep_chunks (d, P, k, deep, &n_deep)))
            return rc;
          if (n_deep == 0)
            break;
        }
      else
        {
          // (deep chunks: the second pass is part of the run; its interval ends here)
          PCCHK (d, hipEventRecord (d->ev_site[1], d->stream));
          PCCHK (d, hipStreamSynchronize (d->stream));
        }
    }
  if (kernel_ms)
    PCCHK (d, hipEventElapsedTime (kernel_ms, d->ev_site[0], d->ev_site[1]));
  if (getenv ("PECALL_LIST_STATS"))
    {
      // how many columns the shortcut left to the beam search, by part of the list
      unsigned long long tot[PCS_BUCKETS] = { 0ull, 0ull, 0ull, 0ull };
      for (int q = 0; q < k; q++)
        {
          unsigned c[PCS_BUCKETS];
          if (hipMemcpy (c, (unsigned *) (d->d_next_site + (size_t) q * PCS_CTRS + 1), sizeof c, hipMemcpyDeviceToHost) == hipSuccess)
            for (int b = 0; b < PCS_BUCKETS; b++)
              tot[b] += c[b];
        }
      fprintf (stderr, "[pecall] %ld columns, listed for the beam by unsettled samples <3 / <8 / <20 / more: %llu %llu %llu %llu\n", n_sites, tot[0], tot[1], tot[2], tot[3]);
#ifdef PECALL_TIMING_PROBES
      {
        // the beam search's phase probes (pecall_site.hip.h): wave cycles summed over the run's launches
        (void) hipDeviceSynchronize ();
        unsigned long long pr[16], z[16] = { 0ull };
        if (hipMemcpyFromSymbol (pr, HIP_SYMBOL (pcs_probe), sizeof pr) == hipSuccess)
          {
            static const char *nm[13] = { "set-up, likelihoods, re-estimation", "expand: duplicates", "expand: pricing", "expand: acceptance + rows", "clean: sort", "clean: end",
              "below-floor samples", "posteriors + marginals", "before write", "write", "clean: cut + homozygous?", "clean: fallback's configuration", "clean: its sort" };
            double tot_c = 0;
            for (int i = 0; i < 13; i++)
              tot_c += (double) pr[i];
            fprintf (stderr, "[pcs_probe]");
            for (int i = 0; i < 13; i++)
              fprintf (stderr, " %s %.1f%%", nm[i], tot_c > 0 ? 100.0 * (double) pr[i] / tot_c : 0.0);
            fprintf (stderr, " | total %.3f G wave-cycles\n", tot_c / 1e9);
            (void) hipMemcpyToSymbol (HIP_SYMBOL (pcs_probe), z, sizeof z);
          }
      }
#endif
      unsigned hc[PCS_BUCKETS] = { 0u, 0u, 0u, 0u };
      if (heavy && hipMemcpy (hc, (unsigned *) (d->d_heavy_ctr + 1), sizeof hc, hipMemcpyDeviceToHost) == hipSuccess)
        fprintf (stderr, "[pecall] started ahead of the shortcut kernels, by samples with variant reads <12 / <20 / <32 / more: %u %u %u %u\n", hc[0], hc[1], hc[2], hc[3]);
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

extern "C" int pecall_dev_pin_host (pecall_dev * d, const void *host_ptr, uint64_t n_bytes)
{
  PCCHK (d, hipSetDevice (d->device));
  if (!host_ptr || !n_bytes)
    return pc_fail (d, "pin_host: empty range");
  if (!pm_host_pin_range (host_ptr, (size_t) n_bytes, d->stream_h2d))
    return pc_fail (d, "pin_host: hipHostRegister of %llu bytes failed", (unsigned long long) n_bytes);
  return 0;
}

extern "C" int pecall_dev_unpin_host (pecall_dev * d, const void *host_ptr)
{
  PCCHK (d, hipSetDevice (d->device));
  if (d->stream_h2d)
    {
      PCCHK (d, hipStreamSynchronize (d->stream_h2d));
      PCCHK (d, hipStreamSynchronize (d->stream_d2h));
    }
  const int rc = pm_host_unpin (host_ptr);
  if (rc == 1)
    return pc_fail (d, "unpin_host: %p is not inside a range pinned through pecall_dev_pin_host", host_ptr);
  if (rc == 2)
    return pc_fail (d, "unpin_host: hipHostUnregister failed");
  return 0;
}

// The seam: host columns in, calls and posteriors out.  The columns travel in chunks: the host-to-device copy of chunk k + 1, the
// kernels of chunk k and the device-to-host copy of chunk k - 1 run side by side (three streams behind each other through events).
// Buffers the caller pinned (pecall_dev_pin_host) are copied from and to directly; others pass through PCS_SLOTS pinned staging
// buffers, filled and emptied by a few host threads (one core moves ~10 GB/s; a 64-sample column is 768 bytes in, ~620 out).
// sparse = the posteriors come back as the list of the columns in which one differs from 1 (post_site / post_rows / post_cap / n_post)
// instead of the dense array `posterior`
static int pcs_call_sites_impl (pecall_dev * d, const uint16_t * reads, const uint8_t * ref_base, const uint8_t * chrom_type, long n_sites,
                                int indiv, int haploid, double threshold, double theta, int8_t * call, double *posterior,
                                int8_t * site_type, int32_t * allele_count, int8_t * n_pass, int32_t * denovo,
                                bool sparse, uint32_t * post_site, double *post_rows, uint64_t post_cap, uint64_t * n_post)
{
  PCCHK (d, hipSetDevice (d->device));
  if (n_sites <= 0 || indiv <= 0 || indiv > PCS_MAXN)
    return pc_fail (d, "call_sites: n_sites %ld, indiv %d (1..%d samples per call)", n_sites, indiv, PCS_MAXN);
  if (!reads || !ref_base || !call || (!sparse && !posterior))
    return pc_fail (d, "call_sites: a required pointer is NULL");
  if (sparse && (!post_site || !post_rows || !n_post || post_cap == 0))
    return pc_fail (d, "call_sites_sparse: the list of posteriors needs post_site, post_rows, a capacity and n_post");
  if (sparse)
    {
      *n_post = 0;
      if (d->sp_cap < post_cap || d->sp_indiv < indiv)
        {
          PCCHK (d, hipDeviceSynchronize ());
          hipFree (d->d_sp_cols);
          hipFree (d->d_sp_rows);
          d->d_sp_cols = nullptr;
          d->d_sp_rows = nullptr;
          d->sp_cap = 0;
          const unsigned long long cap = post_cap > d->sp_cap ? post_cap : d->sp_cap;
          const int wide = indiv > d->sp_indiv ? indiv : d->sp_indiv;
          PCCHK (d, hipMalloc ((void **) &d->d_sp_cols, (size_t) cap * sizeof (unsigned)));
          PCCHK (d, hipMalloc ((void **) &d->d_sp_rows, (size_t) cap * (size_t) wide * sizeof (double)));
          if (!d->d_sp_n)
            PCCHK (d, hipMalloc ((void **) &d->d_sp_n, sizeof (unsigned long long)));
          d->sp_cap = cap;
          d->sp_indiv = wide;
        }
    }
  // (the tables first: the parameter block carries their device addresses)
  int rc = pcs_ensure (d, n_sites, indiv);
  if (rc)
    return rc;
  PcsParams P;
  if ((rc = pcs_params (d, indiv, haploid, threshold, theta, P)))
    return rc;
  if ((rc = pcs_ensure_chunks (d, n_sites)))
    return rc;
  d->staged_sites = n_sites;
  d->staged_indiv = indiv;
  const long C = d->chunk_sites;
  const int nch = (int) ((n_sites + C - 1) / C);
  const size_t N = (size_t) indiv;
  // per column: in = reads + reference base + chromosome class; out = calls + posteriors + type + passes + allele counts + de-novo count
  const size_t in_col = N * PCS_NA * 2 + 2, out_col = N * (sparse ? 1 : 9) + 2 + PCS_NA * 4 + 4;
  const bool in_direct = pm_host_pin_lookup (reads, (size_t) n_sites * N * PCS_NA * 2) && pm_host_pin_lookup (ref_base, (size_t) n_sites)
    && (!chrom_type || pm_host_pin_lookup (chrom_type, (size_t) n_sites));
  const bool out_direct = pm_host_pin_lookup (call, (size_t) n_sites * N) && (sparse || pm_host_pin_lookup (posterior, (size_t) n_sites * N * 8))
    && (!site_type || pm_host_pin_lookup (site_type, (size_t) n_sites)) && (!allele_count || pm_host_pin_lookup (allele_count, (size_t) n_sites * PCS_NA * 4))
    && (!n_pass || pm_host_pin_lookup (n_pass, (size_t) n_sites)) && (!denovo || pm_host_pin_lookup (denovo, (size_t) n_sites * 4));
  const long cmax = n_sites < C ? n_sites : C;
  if (!in_direct && d->h_in_bytes < (size_t) cmax * in_col)
    for (int i = 0; i < PCS_SLOTS; i++)
      {
        if (d->h_in[i])
          hipHostFree (d->h_in[i]);
        d->h_in[i] = nullptr;
        d->h_in_bytes = 0;
        PCCHK (d, hipHostMalloc ((void **) &d->h_in[i], (size_t) C * in_col, hipHostMallocDefault));
        if (i == PCS_SLOTS - 1)
          d->h_in_bytes = (size_t) C * in_col;
      }
  if (!out_direct && d->h_out_bytes < (size_t) cmax * out_col)
    for (int i = 0; i < PCS_SLOTS; i++)
      {
        if (d->h_out[i])
          hipHostFree (d->h_out[i]);
        d->h_out[i] = nullptr;
        d->h_out_bytes = 0;
        PCCHK (d, hipHostMalloc ((void **) &d->h_out[i], (size_t) C * out_col, hipHostMallocDefault));
        if (i == PCS_SLOTS - 1)
          d->h_out_bytes = (size_t) C * out_col;
      }
  // staging layout of a chunk of m columns: [reads][ref][chrom] and [call][posterior][type][passes][allele counts][de-novo]
  auto finish = [&] (int j) -> int
  {
    // chunk j's results are on the host: from the staging slot to the caller's arrays
    PCCHK (d, hipEventSynchronize (d->ev_d2h[j]));
    if (out_direct)
      return 0;
    const long off = (long) j * C, m = n_sites - off < C ? n_sites - off : C;
    const char *o = d->h_out[j % PCS_SLOTS];
    pm_par_memcpy ((char *) (call + off * N), o, (size_t) m * N);
    o += (size_t) m * N;
    if (!sparse)
      {
        pm_par_memcpy ((char *) (posterior + off * N), o, (size_t) m * N * 8);
        o += (size_t) m * N * 8;
      }
    if (site_type)
      memcpy (site_type + off, o, (size_t) m);
    o += m;
    if (n_pass)
      memcpy (n_pass + off, o, (size_t) m);
    o += m;
    if (allele_count)
      memcpy (allele_count + off * PCS_NA, o, (size_t) m * PCS_NA * 4);
    o += (size_t) m * PCS_NA * 4;
    if (denovo)
      memcpy (denovo + off, o, (size_t) m * 4);
    return 0;
  };
  // chunk j's kernels, and its results on their way out behind its beam search (which follows its shortcut kernel)
  auto kernels_and_out = [&] (int j, bool whole_table) -> int
  {
    const long off = (long) j * C, m = n_sites - off < C ? n_sites - off : C;
    int rc2 = whole_table ? 0 : pcs_chunk_reset (d, P, j, off, m);
    if (rc2 || (rc2 = pcs_chunk_kernels (d, P, j, off, m, whole_table, sparse, (N <= 128 ? d->heavy_min : d->heavy_min_wide) > 0 ? 2 : 0)))
      return rc2;
    PCCHK (d, hipStreamWaitEvent (d->stream_d2h, d->ev_call[j], 0));
    char *o = out_direct ? nullptr : d->h_out[j % PCS_SLOTS];
#define PCS_OUT(dst_host, dev_ptr, bytes) do { void *dst_ = out_direct ? (void *) (dst_host) : (void *) o; if (out_direct ? (dst_host) != nullptr : true) \
    PCCHK (d, hipMemcpyAsync (dst_, dev_ptr, bytes, hipMemcpyDeviceToHost, d->stream_d2h)); if (!out_direct) o += (bytes); } while (0)
    PCS_OUT (call + off * N, d->d_call + off * N, (size_t) m * N);
    if (!sparse)
      PCS_OUT (posterior + off * N, d->d_post + off * N, (size_t) m * N * 8);
    PCS_OUT (site_type ? site_type + off : nullptr, d->d_type + off, (size_t) m);
    PCS_OUT (n_pass ? n_pass + off : nullptr, d->d_npass + off, (size_t) m);
    PCS_OUT (allele_count ? allele_count + off * PCS_NA : nullptr, d->d_ac + off * PCS_NA, (size_t) m * PCS_NA * 4);
    PCS_OUT (denovo ? denovo + off : nullptr, d->d_den + off, (size_t) m * 4);
#undef PCS_OUT
    PCCHK (d, hipEventRecord (d->ev_d2h[j], d->stream_d2h));
    return 0;
  };
  if (sparse)
    PCCHK (d, hipMemsetAsync (d->d_sp_n, 0, sizeof (unsigned long long), d->stream));
  const bool trace = getenv ("PECALL_SEAM_TRACE") != nullptr;
  const auto t_start = std::chrono::steady_clock::now ();
  auto since = [&] () { return std::chrono::duration < double, std::milli > (std::chrono::steady_clock::now () - t_start).count (); };
  for (int k = 0; k < nch; k++)
    {
      const long off = (long) k * C, m = n_sites - off < C ? n_sites - off : C;
      const double t0 = since ();
      // ---- in: (a staging slot is free again when the chunk that used it PCS_SLOTS chunks ago has been copied to the device; the
      //      result slot of the same number when that chunk's results have been handed over: finish (k - PCS_SLOTS))
      const uint16_t *src_r = reads + off * N * PCS_NA;
      const uint8_t *src_b = ref_base + off, *src_c = chrom_type ? chrom_type + off : nullptr;
      if (!in_direct)
        {
          if (k >= PCS_SLOTS)
            PCCHK (d, hipEventSynchronize (d->ev_h2d[k - PCS_SLOTS]));
          char *st = d->h_in[k % PCS_SLOTS];
          pm_par_memcpy (st, (const char *) src_r, (size_t) m * N * PCS_NA * 2);
          src_r = (const uint16_t *) st;
          st += (size_t) m * N * PCS_NA * 2;
          memcpy (st, src_b, (size_t) m);
          src_b = (const uint8_t *) st;
          st += m;
          if (src_c)
            {
              memcpy (st, src_c, (size_t) m);
              src_c = (const uint8_t *) st;
            }
        }
      PCCHK (d, hipMemcpyAsync (d->d_sreads + off * N * PCS_NA, src_r, (size_t) m * N * PCS_NA * 2, hipMemcpyHostToDevice, d->stream_h2d));
      PCCHK (d, hipMemcpyAsync (d->d_dom + off, src_b, (size_t) m, hipMemcpyHostToDevice, d->stream_h2d));
      if (src_c)
        PCCHK (d, hipMemcpyAsync (d->d_chromy + off, src_c, (size_t) m, hipMemcpyHostToDevice, d->stream_h2d));
      else
        PCCHK (d, hipMemsetAsync (d->d_chromy + off, 0, (size_t) m, d->stream_h2d));
      PCCHK (d, hipEventRecord (d->ev_h2d[k], d->stream_h2d));
      // ---- the chunk's kernels behind its copy, its results behind its kernels: all queued, the host goes on to the next chunk
      PCCHK (d, hipStreamWaitEvent (d->stream, d->ev_h2d[k], 0));
      if (!out_direct && k >= PCS_SLOTS && (rc = finish (k - PCS_SLOTS)))
        return rc;
      if ((rc = kernels_and_out (k, false)))
        return rc;
      if (trace)
        fprintf (stderr, "[pecall seam] chunk %d: host at %.2f ms, enqueued by %.2f ms (direct in %d out %d)\n", k, t0, since (), (int) in_direct, (int) out_direct);
    }
  for (int j = (out_direct || nch < PCS_SLOTS) ? 0 : nch - PCS_SLOTS; j < nch; j++)
    if ((rc = finish (j)))
      return rc;
  if (trace)
    fprintf (stderr, "[pecall seam] first pass on the host by %.2f ms\n", since ());
  // ---- chunks too deep for the table's head: once more, with the whole table (their columns are on the device)
  {
    std::vector < char >deep;
    int n_deep = 0;
    PCCHK (d, hipStreamSynchronize (d->stream));
    if ((rc = pcs_deep_chunks (d, P, nch, deep, &n_deep)))
      return rc;
    for (int j = 0; j < nch && n_deep; j++)
      if (deep[j])
        {
          if ((rc = kernels_and_out (j, true)) || (rc = finish (j)))
            return rc;
        }
  }
  PCCHK (d, hipStreamSynchronize (d->stream));
  for (int i = 0; i < PCS_CALL_STREAMS; i++)
    PCCHK (d, hipStreamSynchronize (d->stream_call[i]));
  if (sparse)
    {
      // the listed columns, in ascending order of their numbers (the kernels appended them as they came)
      PCCHK (d, hipMemcpy (d->h_ctrs + (size_t) d->cap_chunks * PCS_CTRS, d->d_sp_n, sizeof (unsigned long long), hipMemcpyDeviceToHost));
      const unsigned long long n = d->h_ctrs[(size_t) d->cap_chunks * PCS_CTRS];
      *n_post = n;
      if (n > post_cap)
        return pc_fail (d, "call_sites_sparse: %llu columns have a posterior that is not 1, the list holds %llu (n_post says how many are needed)", n,
                        (unsigned long long) post_cap);
      if (n > 0)
        {
          // (through a page-locked block of this call's own: a copy into pageable memory that shares a page with one of the caller's
          // registered arrays is refused by the runtime -- tools/micro/hostreg.hip -- and a small heap block may well do that)
          char *blk = nullptr;
          const size_t rows_bytes = (size_t) n * N * sizeof (double), cols_bytes = ((size_t) n * sizeof (unsigned) + 63) & ~(size_t) 63;
          PCCHK (d, hipHostMalloc ((void **) &blk, rows_bytes + cols_bytes, hipHostMallocDefault));
          const double *rows = (const double *) blk;
          const unsigned *cols = (const unsigned *) (blk + rows_bytes);
          hipError_t e1 = hipMemcpy ((void *) cols, d->d_sp_cols, (size_t) n * sizeof (unsigned), hipMemcpyDeviceToHost);
          hipError_t e2 = hipMemcpy ((void *) rows, d->d_sp_rows, rows_bytes, hipMemcpyDeviceToHost);
          if (e1 != hipSuccess || e2 != hipSuccess)
            {
              hipHostFree (blk);
              return pc_fail (d, "call_sites_sparse: copying the list back: %s", hipGetErrorString (e1 != hipSuccess ? e1 : e2));
            }
          std::vector < unsigned >order ((size_t) n);
          for (size_t i = 0; i < (size_t) n; i++)
            order[i] = (unsigned) i;
          std::sort (order.begin (), order.end (), [&] (unsigned a, unsigned b) { return cols[a] < cols[b]; });
          for (size_t i = 0; i < (size_t) n; i++)
            {
              post_site[i] = cols[order[i]];
              memcpy (post_rows + i * N, rows + (size_t) order[i] * N, N * sizeof (double));
            }
          hipHostFree (blk);
        }
    }
  if (trace)
    fprintf (stderr, "[pecall seam] done at %.2f ms\n", since ());
  return 0;
}

extern "C" int pecall_dev_call_sites (pecall_dev * d, const uint16_t * reads, const uint8_t * ref_base, const uint8_t * chrom_type, long n_sites,
                                      int indiv, int haploid, double threshold, double theta, int8_t * call, double *posterior,
                                      int8_t * site_type, int32_t * allele_count, int8_t * n_pass, int32_t * denovo)
{
  return pcs_call_sites_impl (d, reads, ref_base, chrom_type, n_sites, indiv, haploid, threshold, theta, call, posterior, site_type, allele_count, n_pass,
                              denovo, false, nullptr, nullptr, 0, nullptr);
}

extern "C" int pecall_dev_call_sites_sparse (pecall_dev * d, const uint16_t * reads, const uint8_t * ref_base, const uint8_t * chrom_type, long n_sites,
                                             int indiv, int haploid, double threshold, double theta, int8_t * call, uint32_t * post_site,
                                             double *post_rows, uint64_t post_cap, uint64_t * n_post, int8_t * site_type, int32_t * allele_count,
                                             int8_t * n_pass, int32_t * denovo)
{
  return pcs_call_sites_impl (d, reads, ref_base, chrom_type, n_sites, indiv, haploid, threshold, theta, call, nullptr, site_type, allele_count, n_pass,
                              denovo, true, post_site, post_rows, post_cap, n_post);
}
