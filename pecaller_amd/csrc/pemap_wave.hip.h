// pemap_wave.hip.h -- wave64 scans and reductions on the DPP path (gfx950), shared by the seed kernels.
#pragma once

__device__ __forceinline__ int pm_lanes_below (unsigned long long m)
{
  return (int) __builtin_amdgcn_mbcnt_hi ((unsigned) (m >> 32), __builtin_amdgcn_mbcnt_lo ((unsigned) m, 0u));
}

// the same, opaque to the optimiser: where two ranks feed a select, the compiler otherwise sinks each mbcnt pair into an arm of its own and
// makes divergent control flow of the select (exec flipped twice per decoded segment in the seed kernel)
__device__ __forceinline__ int pm_lanes_below_here (unsigned long long m)
{
  int r;
  const unsigned lo = (unsigned) m, hi = (unsigned) (m >> 32);
  asm volatile ("v_mbcnt_lo_u32_b32 %0, %1, 0\n\tv_mbcnt_hi_u32_b32 %0, %2, %0" : "=&v" (r) : "s" (lo), "s" (hi));
  return r;
}

// Wave-wide scan / reduction on the DPP path (row shifts inside the 16-lane rows, then the row broadcasts of gfx9): six VALU
// instructions, against six round trips through the LDS crossbar for the __shfl forms.  Lanes without a source lane take `ident`.
template < int CTRL, int ROW_MASK > __device__ __forceinline__ int pm_dpp_or (int ident, int v)
{
  return __builtin_amdgcn_update_dpp (ident, v, CTRL, ROW_MASK, 0xF, false);
}

__device__ __forceinline__ uint32_t pm_wave_incl_sum (uint32_t v)       // inclusive prefix sum over the lanes; lane 63 = the total
{
  v += (uint32_t) pm_dpp_or < 0x111, 0xF > (0, (int) v);        // row_shr:1
  v += (uint32_t) pm_dpp_or < 0x112, 0xF > (0, (int) v);        // row_shr:2
  v += (uint32_t) pm_dpp_or < 0x114, 0xF > (0, (int) v);        // row_shr:4
  v += (uint32_t) pm_dpp_or < 0x118, 0xF > (0, (int) v);        // row_shr:8
  v += (uint32_t) pm_dpp_or < 0x142, 0xA > (0, (int) v);        // row_bcast:15 into rows 1 and 3
  v += (uint32_t) pm_dpp_or < 0x143, 0xC > (0, (int) v);        // row_bcast:31 into rows 2 and 3
  return v;
}

__device__ __forceinline__ int pm_wave_incl_max (int v)       // inclusive prefix maximum of non-negative values
{
  v = max (v, pm_dpp_or < 0x111, 0xF > (0, v));
  v = max (v, pm_dpp_or < 0x112, 0xF > (0, v));
  v = max (v, pm_dpp_or < 0x114, 0xF > (0, v));
  v = max (v, pm_dpp_or < 0x118, 0xF > (0, v));
  v = max (v, pm_dpp_or < 0x142, 0xA > (0, v));
  v = max (v, pm_dpp_or < 0x143, 0xC > (0, v));
  return v;
}

__device__ __forceinline__ int pm_wave_min (int v)      // the minimum over the wave (uniform)
{
  const int big = 0x7FFFFFFF;
  v = min (v, pm_dpp_or < 0x111, 0xF > (big, v));
  v = min (v, pm_dpp_or < 0x112, 0xF > (big, v));
  v = min (v, pm_dpp_or < 0x114, 0xF > (big, v));
  v = min (v, pm_dpp_or < 0x118, 0xF > (big, v));
  v = min (v, pm_dpp_or < 0x142, 0xA > (big, v));
  v = min (v, pm_dpp_or < 0x143, 0xC > (big, v));
  return __builtin_amdgcn_readlane (v, 63);
}

