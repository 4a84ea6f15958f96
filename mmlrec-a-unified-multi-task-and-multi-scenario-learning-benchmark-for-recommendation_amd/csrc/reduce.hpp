// Fixed-order reduction of per-workgroup partial results ("slabs"): out[i] (+)= sum_s slab[s][i].
// Shared by the wgrad, gate and head kernels; bitwise reproducible because the order over s is fixed.
#pragma once
#include "common.hpp"

namespace mml {

struct ReduceSeg {
  const float* slab;  // [S][n] (S = this segment's own partial count)
  float* out;         // rows x cols with leading dimension ldo
  int64_t n;          // rows*cols
  int64_t sstride;    // floats between consecutive partials of this segment
  int32_t cols;
  int32_t accumulate;
  int64_t ldo;
  int64_t start;      // prefix of n over segments
  int32_t S;
  int32_t pad_;
};
constexpr int MAX_REDUCE_SEGS = 40;
struct ReduceLaunch {
  ReduceSeg seg[MAX_REDUCE_SEGS];
  int32_t n;
  int32_t pad_;
  int64_t total;
};

int launch_slab_reduce(const ReduceLaunch& R, hipStream_t st, const char* who);
// mml_rows_reduce_batch: while a collector is set (this thread), launch_slab_reduce appends the segments it is given to the
// collector instead of launching them; the batch entry launches the collected list once.
void reduce_collect(ReduceLaunch* into);

}  // namespace mml
