// L2 2-nearest-neighbour search over unit-norm 128-d descriptors (match_l2.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstddef>

#include "common.hpp"

namespace gtx {

// Bytes of scratch match2nn() needs for nq queries against nt train rows.
size_t match2nn_workspace_bytes(int nq, int nt);
int match2nn_splits(int nq, int nt);

// fp32 [n] -> fp16 [n] on the stream.
void descriptors_to_half(const float* src, void* dst, size_t n, hipStream_t s);

// For every query row: the two train rows with the smallest L2 distance (searched by fp16 dot
// products, distances re-measured in fp32), idx = -1 / d = 3e38 when the train set has fewer rows.
// All pointers are device pointers; *_f16 are the fp16 copies of the fp32 [n][128] descriptors.
void match2nn(const void* q_f16, const float* q_f32, int nq, const void* t_f16, const float* t_f32, int nt, void* workspace,
              int* idx1, int* idx2, float* d1, float* d2, hipStream_t s);

double match2nn_flops(int nq, int nt);

}  // namespace gtx
