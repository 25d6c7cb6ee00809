// FETCH_SIZE calibration for THIS path's access pattern: N random 8-byte gathers (dword-aligned, like pos_index pairs)
// into a 16 GiB table.  Run under `rocprofv3 --pmc FETCH_SIZE`; known request count: N gathers, each a distinct random line
// with probability ~1 (2^28 lines, N = 2^28 gathers -> 63 % distinct lines; the kernel prints the exact distinct count is not
// needed: L2 32 MiB and the 256 MiB Infinity Cache cover < 2 % of the table, so nearly every gather misses).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef uint32_t u32x2 __attribute__((ext_vector_type(2), aligned(4)));
__device__ __forceinline__ uint64_t mix(uint64_t x) { x += 0x9E3779B97F4A7C15ull; x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull; x = (x ^ (x >> 27)) * 0x94D049BB133111EBull; return x ^ (x >> 31); }
__global__ void gather8(const uint32_t* tab, uint64_t n_entries, uint64_t n_gathers, uint32_t* out)
{
  uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t acc = 0;
  for (; i < n_gathers; i += (uint64_t) gridDim.x * blockDim.x) {
    uint64_t k = mix(i) % (n_entries - 1);
    u32x2 v = *(const u32x2*) (tab + k);
    acc += v.x ^ v.y;
  }
  if (acc == 0x12345678u) out[0] = acc;
}
__global__ void stream4(const uint32_t* tab, uint64_t n, uint32_t* out)
{
  uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; uint32_t acc = 0;
  for (; i < n; i += (uint64_t) gridDim.x * blockDim.x) acc += tab[i];
  if (acc == 0x12345678u) out[0] = acc;
}
int main() {
  const uint64_t n_entries = (1ull << 32) + 1; uint32_t *tab, *out;
  if (hipMalloc(&tab, n_entries * 4) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipMalloc(&out, 64); hipMemset(tab, 1, n_entries * 4);
  const uint64_t n_g = 1ull << 30;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  gather8<<<256 * 32, 256>>>(tab, n_entries, 1 << 20, out); hipDeviceSynchronize();
  hipEventRecord(e0); gather8<<<256 * 32, 256>>>(tab, n_entries, n_g, out); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("gather8: %llu gathers in %.2f ms = %.1f G gathers/s; 64 B per gather = %.1f GB (%.2f TB/s)\n", (unsigned long long) n_g, ms, n_g / ms / 1e6, n_g * 64 / 1e9, n_g * 64 / ms / 1e9);
  hipEventRecord(e0); stream4<<<256 * 32, 256>>>(tab, 1ull << 32, out); hipEventRecord(e1); hipEventSynchronize(e1);
  hipEventElapsedTime(&ms, e0, e1);
  printf("stream4: 17.18 GB in %.2f ms = %.2f TB/s\n", ms, 17.18 / ms);
  return 0;
}
