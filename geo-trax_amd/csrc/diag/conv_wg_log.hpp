// DIAGNOSTIC BUILD ONLY (`make -C geo-trax_amd wglog` -> build/libgtx_wglog.so, read by tools/wg_turnover.py); force-included in
// front of conv_igemm_split.hip, never part of libgtx.so. Every workgroup of a split-f16x3 convolution launch records where it
// ran (HW_ID, XCC_ID) and when (s_memrealtime, the 100 MHz counter all XCDs share) at its first and its last instruction:
// how long a CU's workgroup slot stays empty between two workgroups of one launch, and how full the slots are over a launch.
#pragma once
#include <hip/hip_runtime.h>

namespace gtx {
namespace {
constexpr int kWgLogMax = 16384;
__device__ unsigned long long g_wg_log[3 * kWgLogMax];      // per hardware block: (xcc << 32 | hw_id), t_first, t_last
}
}

#define GTXS_DIAG_ENTRY() const unsigned long long wl_t0__ = __builtin_amdgcn_s_memrealtime();
#define GTXS_DIAG_LOOP_BEGIN()
#define GTXS_DIAG_LOOP_END()
#define GTXS_DIAG_PHASE(K)
#define GTXS_DIAG_EXIT()                                                                                     \
  if (threadIdx.x == 0 && blockIdx.x < gtx::kWgLogMax) {                                                     \
    const unsigned long long wl_t1__ = __builtin_amdgcn_s_memrealtime();                                     \
    const unsigned hw__ = __builtin_amdgcn_s_getreg(4 | (31 << 11)), xcc__ = __builtin_amdgcn_s_getreg(20 | (31 << 11)); \
    gtx::g_wg_log[3 * blockIdx.x + 0] = ((unsigned long long)xcc__ << 32) | hw__;                           \
    gtx::g_wg_log[3 * blockIdx.x + 1] = wl_t0__;                                                             \
    gtx::g_wg_log[3 * blockIdx.x + 2] = wl_t1__;                                                             \
  }

// out[3 * n] <- the log's first n entries; clears them
extern "C" int gtx_debug_wg_log(unsigned long long* out, int n) {
  if (n < 0 || n > gtx::kWgLogMax) return -1;
  if (hipDeviceSynchronize() != hipSuccess) return -1;
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(gtx::g_wg_log), 3ull * n * sizeof(unsigned long long)) != hipSuccess) return -1;
  return 0;
}
