// DIAGNOSTIC BUILD ONLY (`make -C geo-trax_amd stamp` -> build/libgtx_stamp.so, read by tools/clock_probe.py); force-included
// in front of conv_igemm_split.hip, never part of libgtx.so. Shader-clock and 100 MHz wall-clock ticks
// spent inside the K loop of the convolution, summed over workgroups, to read the clock the chip
// holds under the kernel (MI355X_MICROARCH.md, DVFS give-back item 6) and the cycles a K loop takes. The sums live in a buffer
// of their own; no output value depends on them. GTX_DIAG_FN names the translation unit's read-out function.
#pragma once
#include <hip/hip_runtime.h>

namespace gtx {
namespace {
__device__ unsigned long long g_clock_stamp[8];   // [0] loop cycles, [1] 100 MHz ticks, [2] K loops stamped, [3] cycles entry -> loop, [4] loop end -> stores retired
}
}

// -DGTXS_DIAG_FRONT_ONLY (conv_igemm_split.hip only): stamp nothing but the front launch (tools/front_probe.py)
#ifdef GTXS_DIAG_FRONT_ONLY
#define GTXS_DIAG_ON (FRONT)
#else
#define GTXS_DIAG_ON (true)
#endif

#ifndef GTXS_DIAG_NO_ENTRY
#define GTXS_DIAG_ENTRY() const unsigned long long st_e0__ = __builtin_amdgcn_s_memtime();
#define GTXS_DIAG_EXIT()                                                                                    \
  {                                                                                                         \
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       /* the workgroup's stores have left */            \
    const unsigned long long st_e1__ = __builtin_amdgcn_s_memtime();                                        \
    if (GTXS_DIAG_ON && threadIdx.x == 0) {                                                                                 \
      atomicAdd(&gtx::g_clock_stamp[3], st_c0__ - st_e0__);     /* entry -> K loop */                        \
      atomicAdd(&gtx::g_clock_stamp[4], st_e1__ - st_c1s__);    /* K loop end -> stores retired */           \
    }                                                                                                       \
  }
#endif
#define GTXS_DIAG_LOOP_BEGIN() \
  const unsigned long long st_c0__ = __builtin_amdgcn_s_memtime(), st_r0__ = __builtin_amdgcn_s_memrealtime(); \
  [[maybe_unused]] unsigned long long st_c1s__ = 0, st_ph__ = st_c0__;
// cycles since the previous phase stamp, summed into g_clock_stamp[K] (7: a K chunk's weight commit + matrix phase)
#define GTXS_DIAG_PHASE(K)                                                                                  \
  {                                                                                                         \
    const unsigned long long st_p__ = __builtin_amdgcn_s_memtime();                                         \
    if (GTXS_DIAG_ON && threadIdx.x == 0) atomicAdd(&gtx::g_clock_stamp[K], st_p__ - st_ph__);                              \
    st_ph__ = st_p__;                                                                                       \
  }
#define GTXS_DIAG_LOOP_END()                                                                                \
  {                                                                                                         \
    const unsigned long long st_c1__ = __builtin_amdgcn_s_memtime(), st_r1__ = __builtin_amdgcn_s_memrealtime(); \
    st_c1s__ = st_c1__;                                                                                     \
    if (GTXS_DIAG_ON && threadIdx.x == 0) {                                                                                 \
      atomicAdd(&gtx::g_clock_stamp[0], st_c1__ - st_c0__);                                                 \
      atomicAdd(&gtx::g_clock_stamp[1], st_r1__ - st_r0__);                                                 \
      atomicAdd(&gtx::g_clock_stamp[2], 1ull);                                                              \
    }                                                                                                       \
  }

// out = {shader-clock ticks, 100 MHz ticks, K loops, 0...} summed since the last call; clears the sums.
extern "C" int GTX_DIAG_FN(unsigned long long out[8]) {
  if (hipDeviceSynchronize() != hipSuccess) return -1;
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(gtx::g_clock_stamp), 8 * sizeof(unsigned long long)) != hipSuccess) return -1;
  const unsigned long long zero[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  return hipMemcpyToSymbol(HIP_SYMBOL(gtx::g_clock_stamp), zero, sizeof zero) == hipSuccess ? 0 : -1;
}
