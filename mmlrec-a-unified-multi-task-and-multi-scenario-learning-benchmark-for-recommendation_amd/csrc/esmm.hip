// ESMM output stage (reference model/esmm.py:58-62): out = [ctr, ctr * cvr] from the two sigmoid heads, with the summed
// BCE of both outputs (model/basemodel.py:294-296, log terms clamped at -100) and its gradient w.r.t. the two head
// probabilities.  B x 2 floats: one workgroup walks the batch, so the loss sum has a fixed order and needs no zeroing
// pass or atomics.
#include "common.hpp"

namespace mml {

constexpr int ESMM_THREADS = 1024;

__global__ __launch_bounds__(ESMM_THREADS) void esmm_combine_kernel(const float* praw, int64_t ldr, const float* y,
                                                                    int64_t ldy, const float* dout, int64_t lddo,
                                                                    float* pout, int64_t ldo, float* draw, int64_t lddr,
                                                                    float* loss, int64_t B) {
  __shared__ float red[ESMM_THREADS / 64];
  float acc = 0.f;
  for (int64_t b = threadIdx.x; b < B; b += ESMM_THREADS) {
    const float c = praw[b * ldr], v = praw[b * ldr + 1];
    const float p0 = c, p1 = c * v;
    pout[b * ldo] = p0;
    pout[b * ldo + 1] = p1;
    float g0 = 0.f, g1 = 0.f;
    if (y) {
      const float y0 = y[b * ldy], y1 = y[b * ldy + 1];
      acc += -(y0 * fmaxf(logf(p0), -100.f) + (1.f - y0) * fmaxf(log1pf(-p0), -100.f));
      acc += -(y1 * fmaxf(logf(p1), -100.f) + (1.f - y1) * fmaxf(log1pf(-p1), -100.f));
      g0 = (p0 - y0) / fmaxf((1.f - p0) * p0, 1e-12f);
      g1 = (p1 - y1) / fmaxf((1.f - p1) * p1, 1e-12f);
    } else if (dout) {
      g0 = dout[b * lddo];
      g1 = dout[b * lddo + 1];
    }
    if (draw) {
      draw[b * lddr] = g0 + g1 * v;  // d/d ctr
      draw[b * lddr + 1] = g1 * c;   // d/d cvr
    }
  }
  if (!loss) return;
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    float s = 0.f;
    for (int w = 0; w < ESMM_THREADS / 64; ++w) s += red[w];
    loss[0] = s;
  }
}

}  // namespace mml

using namespace mml;

extern "C" int mml_esmm_combine(const float* p_raw, int64_t ldr, const float* y, int64_t ldy, const float* d_out,
                                int64_t lddo, float* p_out, int64_t ldo, float* d_raw, int64_t lddr, float* loss,
                                int64_t B, mml_stream_t stream) {
  MML_REQUIRE(B >= 0, "mml_esmm_combine: negative batch");
  if (B == 0) return MML_OK;
  MML_REQUIRE(p_raw && p_out && ldr >= 2 && ldo >= 2, "mml_esmm_combine: null probabilities or leading dimension < 2");
  MML_REQUIRE(!y || ldy >= 2, "mml_esmm_combine: ldy < 2");
  MML_REQUIRE(!d_out || lddo >= 2, "mml_esmm_combine: lddo < 2");
  MML_REQUIRE(!d_raw || lddr >= 2, "mml_esmm_combine: lddr < 2");
  MML_REQUIRE(!loss || y, "mml_esmm_combine: a loss needs labels");
  MML_LAUNCH(esmm_combine_kernel, dim3(1), dim3(ESMM_THREADS), 0, to_stream(stream), p_raw, ldr, y, ldy, d_out, lddo,
             p_out, ldo, d_raw, lddr, loss, B);
  return check_launch("mml_esmm_combine");
}
