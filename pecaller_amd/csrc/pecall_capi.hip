// pecall_capi.hip -- C-ABI of the PECaller likelihood kernel (include/pemap_hip.h, pecall_dev_*).
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdarg.h>
#include <math.h>
#include "../../include/pemap_hip.h"
#include "pecall_kernels.hip.h"

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
