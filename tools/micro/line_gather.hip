// Rate of random aligned blocks of 64 / 128 / 256 bytes read whole (16 bytes per lane, 4 / 8 / 16 adjacent lanes per block)
// against the rate of random 8-byte gathers, in a 16 GiB table and in a 128 GiB one (8 replicas of the look-up table).
// Answers: does a 64-byte line read whole cost the same as an 8-byte gather (one fabric request either way)?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef uint32_t u32x2 __attribute__((ext_vector_type(2), aligned(4)));
__device__ __forceinline__ uint64_t mix(uint64_t x) { x += 0x9E3779B97F4A7C15ull; x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull; x = (x ^ (x >> 27)) * 0x94D049BB133111EBull; return x ^ (x >> 31); }

__global__ void gather8(const uint32_t* tab, uint64_t n_entries, uint64_t n, uint32_t* out)
{
  uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t acc = 0;
  for (; i < n; i += (uint64_t) gridDim.x * blockDim.x) {
    uint64_t k = mix(i * 2) % (n_entries - 1);
    u32x2 v = *(const u32x2*) (tab + k);
    acc += v.x ^ v.y;
  }
  if (acc == 0x12345678u) out[0] = acc;
}

// LANES adjacent lanes read one random block of LANES * 16 bytes; n = number of 16-byte pieces
template <int LANES> __global__ void gatherblk(const uint4* tab, uint64_t n_blocks, uint64_t n, uint32_t* out)
{
  uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t acc = 0;
  for (; i < n; i += (uint64_t) gridDim.x * blockDim.x) {
    uint64_t blk = mix((i / LANES) * 2 + 1) % n_blocks;
    uint4 v = tab[blk * LANES + (i % LANES)];
    acc += v.x ^ v.y ^ v.z ^ v.w;
  }
  if (acc == 0x12345678u) out[0] = acc;
}

// one lane reads a whole random 64-byte line (4 x 16 bytes)
__global__ void gatherline1(const uint4* tab, uint64_t n_lines, uint64_t n, uint32_t* out)
{
  uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t acc = 0;
  for (; i < n; i += (uint64_t) gridDim.x * blockDim.x) {
    uint64_t ln = mix(i * 2 + 1) % n_lines;
    const uint4* p = tab + ln * 4;
    uint4 a = p[0], b = p[1], c = p[2], d = p[3];
    acc += a.x ^ b.y ^ c.z ^ d.w;
  }
  if (acc == 0x12345678u) out[0] = acc;
}

int main() {
  size_t fr = 0, tot = 0; hipMemGetInfo(&fr, &tot);
  printf("hipMemGetInfo: free %.2f GB (%.2f GiB), total %.2f GB (%.2f GiB)\n", fr / 1e9, fr / 1073741824.0, tot / 1e9, tot / 1073741824.0);
  uint32_t* out; hipMalloc(&out, 64);
  const uint64_t sizes[] = { 16ull << 30, 128ull << 30 };
  for (int si = 0; si < 2; si++) {
    const uint64_t bytes = sizes[si];
    uint32_t* tab;
    if (hipMalloc(&tab, bytes + 64) != hipSuccess) { printf("alloc of %.0f GiB failed\n", bytes / 1073741824.0); continue; }
    hipMemset(tab, 1, bytes);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms;
    const uint64_t n_g = 1ull << 30;
    const dim3 grid(256 * 32), blk(256);
    gather8<<<grid, blk>>>(tab, bytes / 4, 1 << 20, out); hipDeviceSynchronize();
    hipEventRecord(e0); gather8<<<grid, blk>>>(tab, bytes / 4, n_g, out); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    printf("table %4.0f GiB: 8-byte gathers           %.1f G/s\n", bytes / 1073741824.0, n_g / ms / 1e6);
    hipEventRecord(e0); gatherblk<4><<<grid, blk>>>((const uint4*) tab, bytes / 64, n_g * 4, out); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    printf("table %4.0f GiB: 64-byte lines, 4 lanes   %.1f G lines/s  (%.2f TB/s)\n", bytes / 1073741824.0, n_g / ms / 1e6, n_g * 64 / ms / 1e9);
    hipEventRecord(e0); gatherline1<<<grid, blk>>>((const uint4*) tab, bytes / 64, n_g, out); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    printf("table %4.0f GiB: 64-byte lines, 1 lane    %.1f G lines/s  (%.2f TB/s)\n", bytes / 1073741824.0, n_g / ms / 1e6, n_g * 64 / ms / 1e9);
    hipEventRecord(e0); gatherblk<8><<<grid, blk>>>((const uint4*) tab, bytes / 128, n_g * 4, out); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    printf("table %4.0f GiB: 128-byte blocks, 8 lanes %.1f G blocks/s (%.2f TB/s)\n", bytes / 1073741824.0, n_g / 2 / ms / 1e6, n_g * 64 / ms / 1e9);
    hipEventRecord(e0); gatherblk<16><<<grid, blk>>>((const uint4*) tab, bytes / 256, n_g * 4, out); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    printf("table %4.0f GiB: 256-byte blocks, 16 lanes %.1f G blocks/s (%.2f TB/s)\n", bytes / 1073741824.0, n_g / 4 / ms / 1e6, n_g * 64 / ms / 1e9);
    hipFree(tab);
  }
  return 0;
}
