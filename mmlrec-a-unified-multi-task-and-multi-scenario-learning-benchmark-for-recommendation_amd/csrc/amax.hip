// Operand magnitudes for the two-plane fp16 GEMM arithmetic (include/mmlrec.h, "operand magnitudes"): the largest |x|
// of a tensor as a float bit pattern (for non-negative floats the unsigned order of the bits is the order of the
// values), raised with atomic max.  The GEMM launches produce the magnitudes of their own outputs; this file serves
// tensors that come from elsewhere (weights at the start of a step, outputs of the row kernels).  HBM-bound: one read.
#include "common.hpp"

namespace mml {

struct AmaxLaunch {
  mml_amax_desc t[MML_MAX_AMAX];
  int32_t blk0[MML_MAX_AMAX + 1];  // first workgroup of tensor i in the 1-D grid
  int32_t n;
};

__global__ __launch_bounds__(256) void amax_kernel(const AmaxLaunch L) {
  int ti = 0;
  while (ti + 1 < L.n && (int)blockIdx.x >= L.blk0[ti + 1]) ++ti;
  const mml_amax_desc& T = L.t[ti];
  const int bx = (int)blockIdx.x - L.blk0[ti], nb = L.blk0[ti + 1] - L.blk0[ti];
  float amf = 0.f;
  const bool vec = (T.cols % 4 == 0) && (T.ld % 4 == 0) && aligned16(T.x);
  if (vec) {
    // eight / four independent 16-byte loads in flight per thread (a plain grid-stride loop issues them one at a time)
    const int c4 = T.cols / 4;
    const int64_t total = T.rows * c4;
    const int64_t step = (int64_t)nb * 256;
    const bool flat = T.ld == T.cols;  // contiguous rows: no index arithmetic (a 64-bit division per load otherwise)
    auto at = [&](int64_t i) __attribute__((always_inline)) {
      if (flat) return reinterpret_cast<const float4*>(T.x)[i];
      const int64_t r = (total < 0x7fffffff) ? (int64_t)((uint32_t)i / (uint32_t)c4) : i / c4;
      return *reinterpret_cast<const float4*>(T.x + r * T.ld + 4 * (i - r * c4));
    };
    int64_t i = (int64_t)bx * 256 + threadIdx.x;
    for (; i + 7 * step < total; i += 8 * step) {
      const float4 v0 = at(i), v1 = at(i + step), v2 = at(i + 2 * step), v3 = at(i + 3 * step);
      const float4 v4 = at(i + 4 * step), v5 = at(i + 5 * step), v6 = at(i + 6 * step), v7 = at(i + 7 * step);
      amax_acc(amf, v0); amax_acc(amf, v1); amax_acc(amf, v2); amax_acc(amf, v3);
      amax_acc(amf, v4); amax_acc(amf, v5); amax_acc(amf, v6); amax_acc(amf, v7);
    }
    for (; i + 3 * step < total; i += 4 * step) {
      const float4 v0 = at(i), v1 = at(i + step), v2 = at(i + 2 * step), v3 = at(i + 3 * step);
      amax_acc(amf, v0); amax_acc(amf, v1); amax_acc(amf, v2); amax_acc(amf, v3);
    }
    for (; i < total; i += step) amax_acc(amf, at(i));
  } else {
    const int64_t total = T.rows * T.cols;
    for (int64_t i = (int64_t)bx * 256 + threadIdx.x; i < total; i += (int64_t)nb * 256) {
      const int64_t r = i / T.cols;
      amax_acc(amf, T.x[r * T.ld + (i - r * T.cols)]);
    }
  }
  // (Inf / NaN: the integer pattern of an Inf is what the consumers test for; a NaN does not register -- see amax_acc)
  amax_flush<true>(amf, T.slot);
}

__global__ __launch_bounds__(256) void amax_reset_kernel(uint32_t* slots, int64_t words) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < words; i += (int64_t)gridDim.x * 256) slots[i] = 0u;
}

}  // namespace mml

using namespace mml;

extern "C" int mml_amax_batch(const mml_amax_desc* d, int32_t n, mml_stream_t stream) {
  MML_REQUIRE(n >= 0 && (n == 0 || d), "mml_amax_batch: bad descriptor array");
  int i = 0;
  while (i < n) {
    AmaxLaunch L{};
    int total = 0;
    while (i < n && L.n < MML_MAX_AMAX) {
      const mml_amax_desc& q = d[i];
      MML_REQUIRE(q.slot && q.rows >= 0 && q.cols >= 0 && q.ld >= q.cols && (q.x || q.rows * q.cols == 0),
                  "mml_amax_batch: tensor %d malformed", i);
      ++i;
      if (q.rows * q.cols == 0) continue;
      // 16 elements (four 16-byte loads, all in flight at once) per thread where the tensor is large enough to give
      // every CU work that way; beyond 8192 workgroups the threads loop
      int64_t nb = cdiv(q.rows * (int64_t)q.cols, 256 * 16);
      static int cap = 0;
      if (!cap) {
        const char* e = getenv("MMLREC_AMAX_BLOCKS");  // lab knob
        cap = e ? atoi(e) : 1024;
      }
      // One atomic per workgroup ended the kernel, all of them on the slot's one 32-byte line (~8 ns each, serialised):
      // the input of the first layer ([65 536, 240]) took 54 us with 4 096 workgroups, 35 with 2 048, 20-21 with 128 .. 384
      // (warm caches, eager launch included; tools/lab/bench_amax.py).  Since the workgroups look at the slot before the
      // atomic (amax_flush<true>) the count no longer matters: 20 us at 256, 2 048 and 4 096.
      if (nb > cap) nb = cap;
      // Round 6: tensors of at most 32 MB (PepNet's [65 536, 80] gate inputs, 21 MB) on at most 256 workgroups: their pass is
      // its fixed costs -- launch, one look at / atomic on the slot's line per workgroup -- not bandwidth: 17-18 -> 12-13 us
      // per launch (amax_kernel 35 -> 25 us per PepNet step; 128 the same, 512 30 us).
      static const bool cap_env = getenv("MMLREC_AMAX_BLOCKS") != nullptr;
      if (!cap_env && q.rows * (int64_t)q.cols <= ((int64_t)8 << 20) && nb > 256) nb = 256;
      L.blk0[L.n] = total;
      L.t[L.n++] = q;
      total += (int)nb;
    }
    L.blk0[L.n] = total;
    if (total == 0) continue;
    MML_LAUNCH(amax_kernel, dim3((unsigned)total), dim3(256), 0, to_stream(stream), L);
    int rc = check_launch("mml_amax_batch");
    if (rc) return rc;
  }
  return MML_OK;
}

extern "C" int mml_amax_reset(uint32_t* slots, int64_t n_slots, mml_stream_t stream) {
  MML_REQUIRE(n_slots >= 0 && (n_slots == 0 || slots), "mml_amax_reset: bad arguments");
  if (n_slots == 0) return MML_OK;
  // a kernel, not hipMemsetAsync: the step is replayed from HIP graphs, and a memset NODE captured in front of the
  // kernels that raise the slots did not keep its place in the replayed order with the runtime of this image (the
  // consumers then read zeros -> scale 2^110 -> Inf)
  const int64_t words = n_slots * MML_AMAX_WORDS;
  int64_t nb = cdiv(words, 256);
  if (nb > 1024) nb = 1024;
  MML_LAUNCH(amax_reset_kernel, dim3((unsigned)nb), dim3(256), 0, to_stream(stream), slots, words);
  return check_launch("mml_amax_reset");
}
