// micro-benchmark: issue rate of fp64 VALU ops on gfx950 (cycles per wave64 instruction per SIMD)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
template <int OP> __global__ __launch_bounds__(64) void k(double* out, int iters, double seed)
{
  double a[8];
  for (int i = 0; i < 8; i++) a[i] = seed + threadIdx.x * 0.001 + i;
  unsigned acc = 0;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < 8; i++) {
      if (OP == 0) a[i] = a[i] + 1.0000001;
      if (OP == 1) a[i] = __builtin_fmax(a[i], a[(i + 1) & 7] - 3.0) ;
      if (OP == 2) { unsigned long long m = __builtin_amdgcn_fcmp(a[i], a[(i+3)&7], 2); unsigned long long j; asm volatile("v_addc_co_u32 %0, %1, %0, %0, %2" : "+v"(acc), "=s"(j) : "s"(m)); }
      if (OP == 3) asm volatile("v_max_f64 %0, %0, %1" : "+v"(a[i]) : "v"(a[(i+1)&7]));
      if (OP == 4) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[i]) : "v"(a[(i+1)&7]));
      if (OP == 5) { int t; asm volatile("v_bfe_i32 %0, %1, 3, 1" : "=v"(t) : "v"(acc)); acc += t; }
    }
  }
  double s = 0; for (int i = 0; i < 8; i++) s += a[i];
  out[blockIdx.x * 64 + threadIdx.x] = s + acc;
}
template <int OP> void run(const char* name, int waves_per_simd)
{
  int blocks = 256 * 4 * waves_per_simd, iters = 20000;
  double* d; hipMalloc(&d, blocks * 64 * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<OP><<<blocks, 64>>>(d, 100, 1.0); hipDeviceSynchronize();
  hipEventRecord(e0); k<OP><<<blocks, 64>>>(d, iters, 1.0); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double n_per_simd = (double) waves_per_simd * iters * 8 * (OP == 2 ? 2 : 1) * (OP == 5 ? 2 : 1);
  printf("%-22s waves/SIMD=%d  %.3f ms  -> %.2f cycles per wave-instruction per SIMD at 2.4 GHz\n", name, waves_per_simd, ms, ms * 1e-3 * 2.4e9 / n_per_simd);
  hipFree(d);
}
int main() {
  for (int w : {1, 2, 4}) {
    run<0>("v_add_f64 (C)", w); run<4>("v_add_f64 (asm)", w); run<3>("v_max_f64 (asm)", w); run<1>("fmax+sub (2 ops as 1)", w);
    run<2>("v_cmp_f64+v_addc", w); run<5>("v_bfe_i32+v_add_u32", w);
  }
  return 0;
}
