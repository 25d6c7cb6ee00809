// hostreg.hip -- what hipHostRegister / hipHostUnregister accept on this stack (ROCm 7.2, MI355X): unaligned ranges, adjacent
// ranges that share a page, nested ranges.  Answers shaped pemap_capi.hip's pin_range.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#define SHOW(call) do { hipError_t e_ = (call); printf ("%-70s -> %s\n", #call, hipGetErrorString (e_)); (void) hipGetLastError (); } while (0)
int main ()
{
  const size_t N = 8u << 20;
  char *a = (char *) malloc (N + 8192), *b = (char *) malloc (N + 8192);
  memset (a, 1, N + 8192);
  memset (b, 2, N + 8192);
  char *d;
  hipMalloc ((void **) &d, N);
  hipStream_t st;
  hipStreamCreateWithFlags (&st, hipStreamNonBlocking);
  printf ("a = %p b = %p\n", a, b);
  SHOW (hipHostRegister (a + 16, N, hipHostRegisterDefault));
  SHOW (hipMemcpyAsync (d, a + 16 + 304 * 1000, 304 * 3000, hipMemcpyHostToDevice, st));
  SHOW (hipStreamSynchronize (st));
  SHOW (hipHostUnregister (a + 16));
  puts ("-- adjacent unaligned ranges sharing a page");
  SHOW (hipHostRegister (b + 16, 304 * 3000, hipHostRegisterDefault));
  SHOW (hipHostRegister (b + 16 + 304 * 3000, 304 * 1, hipHostRegisterDefault));
  SHOW (hipHostRegister (b + 16 + 304 * 3001, 304 * 5999, hipHostRegisterDefault));
  SHOW (hipMemcpyAsync (d, b + 16 + 304 * 2990, 304 * 20, hipMemcpyHostToDevice, st));
  SHOW (hipStreamSynchronize (st));
  SHOW (hipHostUnregister (b + 16 + 304 * 3000));
  SHOW (hipHostUnregister (b + 16));
  SHOW (hipHostUnregister (b + 16 + 304 * 3001));
  puts ("-- a range registered while another object's range is live, then unregistered in the other order");
  SHOW (hipHostRegister (a + 16, N, hipHostRegisterDefault));
  SHOW (hipHostRegister (b + 16, 304 * 3000, hipHostRegisterDefault));
  SHOW (hipHostRegister (b + 16 + 304 * 3000, 304 * 1, hipHostRegisterDefault));
  SHOW (hipMemcpyAsync (d, a + 16, 304 * 3000, hipMemcpyHostToDevice, st));
  SHOW (hipMemcpyAsync (d, b + 16, 304 * 3000, hipMemcpyHostToDevice, st));
  SHOW (hipStreamSynchronize (st));
  SHOW (hipHostUnregister (a + 16));
  SHOW (hipHostUnregister (b + 16));
  SHOW (hipHostUnregister (b + 16 + 304 * 3000));
  puts ("-- nested");
  SHOW (hipHostRegister (a, N, hipHostRegisterDefault));
  SHOW (hipHostRegister (a + 4096, 8192, hipHostRegisterDefault));
  SHOW (hipHostUnregister (a + 4096));
  SHOW (hipHostUnregister (a));
  puts ("-- page-aligned union");
  char *pa = (char *) ((uintptr_t) (b + 16) & ~(uintptr_t) 4095);
  SHOW (hipHostRegister (pa, 4096 * 300, hipHostRegisterDefault));
  SHOW (hipHostUnregister (pa));
  SHOW (hipHostRegister (pa, 4096 * 600, hipHostRegisterDefault));
  SHOW (hipHostUnregister (pa));
  return 0;
}
