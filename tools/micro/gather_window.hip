// Does locality of random 8-byte gathers matter on MI355X?  Each wave instruction gathers 64 random entries from ONE window
// of `win` entries placed at random in a 16 GiB table.  win = table size is the fully random pattern of the look-up kernel.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef uint32_t u32x2 __attribute__((ext_vector_type(2), aligned(4)));
__device__ __forceinline__ uint64_t mix(uint64_t x) { x += 0x9E3779B97F4A7C15ull; x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull; x = (x ^ (x >> 27)) * 0x94D049BB133111EBull; return x ^ (x >> 31); }
__global__ void gatherw(const uint32_t* tab, uint64_t n_entries, uint64_t win, int per_win, uint64_t n_gathers, uint32_t* out)
{
  uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t acc = 0;
  for (; i < n_gathers; i += (uint64_t) gridDim.x * blockDim.x) {
    // per_win consecutive gather indices share a window
    uint64_t wbase = (mix((i / per_win) * 2 + 1) % (n_entries / win)) * win;
    uint64_t k = wbase + mix(i * 2) % (win - 1);
    u32x2 v = *(const u32x2*) (tab + k);
    acc += v.x ^ v.y;
  }
  if (acc == 0x12345678u) out[0] = acc;
}
int main() {
  const uint64_t n_entries = (1ull << 32); uint32_t *tab, *out;
  if (hipMalloc(&tab, (n_entries + 16) * 4) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipMalloc(&out, 64); hipMemset(tab, 1, n_entries * 4);
  const uint64_t n_g = 1ull << 29;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  gatherw<<<256 * 32, 256>>>(tab, n_entries, n_entries, 64, 1 << 20, out); hipDeviceSynchronize();
  const uint64_t wins[] = { 1ull << 32, 1ull << 28, 1ull << 24, 1ull << 19, 1ull << 16, 1ull << 14, 1ull << 12 };
  const int pers[] = { 64, 25, 4096 };
  for (int pi = 0; pi < 3; pi++)
    for (int wi = 0; wi < 7; wi++) {
      float ms;
      hipEventRecord(e0); gatherw<<<256 * 32, 256>>>(tab, n_entries, wins[wi], pers[pi], n_g, out); hipEventRecord(e1); hipEventSynchronize(e1);
      hipEventElapsedTime(&ms, e0, e1);
      printf("window %10.0f KB, %4d gathers per window: %.1f G gathers/s\n", wins[wi] * 4.0 / 1024, pers[pi], n_g / ms / 1e6);
    }
  return 0;
}
