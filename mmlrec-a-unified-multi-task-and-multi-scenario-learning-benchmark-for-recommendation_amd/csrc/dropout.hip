// Dropout of a DNN layer's output (reference model/utils.py:121, :159: nn.Dropout(dropout_rate) after the activation):
// out = x * keep / (1 - p), keep ~ Bernoulli(1 - p).  HBM-bound: one read, one write.
//
// The mask is never stored: it is a pure function of (seed, step, site, row0 + row, column / 4) through Philox4x32-10 (the
// counter-based generator of Salmon et al., SC'11: ten rounds of two 32x32 -> 64-bit multiplications), so the backward
// launch regenerates exactly the mask the forward applied (same call on the gradient), a replayed HIP graph draws a
// fresh mask every step (the step counter is read from device memory), and a CPU restatement can reproduce it bit for
// bit (oracle/mmlrec_oracle.py: philox4x32).  It is NOT torch's generator stream: the reference's mask for a given
// torch seed cannot be reproduced, only its distribution and its arithmetic (x * (1 / (1 - p)) for kept elements,
// x * 0 for dropped ones).
#include "common.hpp"

namespace mml {

__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                                              uint32_t out[4]) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

struct DropArgs {
  const float* x;
  float* out;
  int64_t ldx, ldo, rows, row0;
  int32_t cols, c4;       // c4 = ceil(cols / 4): one Philox block covers four consecutive columns of a row
  uint32_t thr;           // dropped iff word < thr, thr = floor(p * 2^32)
  float scale;            // 1 / (1 - p), rounded once in fp32 like torch's noise.div_(1 - p)
  uint32_t k0, k1, site;
  const int32_t* step_dev;
  int32_t step;
  int32_t accumulate;
  int32_t vec;
};

__global__ __launch_bounds__(256) void dropout_kernel(const DropArgs a) {
  const uint32_t step = a.step_dev ? (uint32_t)*a.step_dev : (uint32_t)a.step;
  const int64_t total = a.rows * a.c4;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t r = (total < 0x7fffffff) ? (int64_t)((uint32_t)i / (uint32_t)a.c4) : i / a.c4;
    const int32_t q = (int32_t)(i - r * a.c4);
    uint32_t w[4];
    philox4x32_10((uint32_t)(a.row0 + r), (uint32_t)q, step, a.site, a.k0, a.k1, w);
    const float* xp = a.x + r * a.ldx + 4 * q;
    float* op = a.out + r * a.ldo + 4 * q;
    if (a.vec) {
      const float4 v = *reinterpret_cast<const float4*>(xp);
      float4 o;
      o.x = v.x * (w[0] < a.thr ? 0.f : a.scale);
      o.y = v.y * (w[1] < a.thr ? 0.f : a.scale);
      o.z = v.z * (w[2] < a.thr ? 0.f : a.scale);
      o.w = v.w * (w[3] < a.thr ? 0.f : a.scale);
      if (a.accumulate) {
        const float4 p = *reinterpret_cast<const float4*>(op);
        o.x += p.x; o.y += p.y; o.z += p.z; o.w += p.w;
      }
      *reinterpret_cast<float4*>(op) = o;
    } else {
      const int n = (a.cols - 4 * q) < 4 ? (a.cols - 4 * q) : 4;
      for (int j = 0; j < n; ++j) {
        float o = xp[j] * (w[j] < a.thr ? 0.f : a.scale);
        if (a.accumulate) o += op[j];
        op[j] = o;
      }
    }
  }
}

}  // namespace mml

using namespace mml;

extern "C" int mml_dropout(const float* x, int64_t ldx, float* out, int64_t ldo, int64_t rows, int32_t cols, int64_t row0,
                           float p, uint64_t seed, uint32_t site, const int32_t* step_dev, int32_t step,
                           int32_t accumulate, mml_stream_t stream) {
  MML_REQUIRE(rows >= 0 && cols >= 0, "mml_dropout: negative size");
  MML_REQUIRE(p >= 0.f && p < 1.f, "mml_dropout: p must be in [0, 1)");  // (nn.Dropout accepts p = 1: all zeros; the
  // reference documents [0, 1), model/utils.py:109)
  if (rows == 0 || cols == 0) return MML_OK;
  MML_REQUIRE(x && out && ldx >= cols && ldo >= cols, "mml_dropout: bad arguments");
  MML_REQUIRE(row0 >= 0 && row0 + rows <= 0xffffffffLL, "mml_dropout: row counter beyond 2^32");
  DropArgs a;
  a.x = x; a.out = out; a.ldx = ldx; a.ldo = ldo; a.rows = rows; a.row0 = row0; a.cols = cols; a.c4 = (cols + 3) / 4;
  const double t = (double)p * 4294967296.0;
  a.thr = t >= 4294967295.0 ? 0xffffffffu : (uint32_t)t;
  a.scale = 1.0f / (1.0f - p);
  a.k0 = (uint32_t)seed; a.k1 = (uint32_t)(seed >> 32); a.site = site;
  a.step_dev = step_dev; a.step = step; a.accumulate = accumulate;
  a.vec = (cols % 4 == 0) && (ldx % 4 == 0) && (ldo % 4 == 0) && aligned16(x) && aligned16(out);
  int64_t nb = cdiv(rows * a.c4, 256);
  if (nb > 8192) nb = 8192;
  MML_LAUNCH(dropout_kernel, dim3((unsigned)nb), dim3(256), 0, to_stream(stream), a);
  return check_launch("mml_dropout");
}
