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
      acc += -(y0 * bce_log_clamp(logf(p0)) + (1.f - y0) * bce_log_clamp(log1pf(-p0)));
      acc += -(y1 * bce_log_clamp(logf(p1)) + (1.f - y1) * bce_log_clamp(log1pf(-p1)));
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

// ESCM output stage and loss (reference model/escm.py:74-112 + the loss branch of BaseModel.fit,
// model/basemodel.py:284-292): out = [ctr, cvr, ctr * cvr];
//   loss = BCE_sum(ctr, y0) + cf_w * mean_b( L1 * ips_b * y0_b ) + global_w * BCE_sum(ctr * cvr, y1)
//   L1 = BCE_sum(cvr, y1)  (a scalar),  ips_b = clip(1 / max(ctr_b * N, 1e-6), -15, 15) * B,  N = sum_b y0_b
// (counterfact_ipw, escm.py:98-112; `ips.stop_gradient = True` there is a no-op attribute in torch, so the gradient DOES
// flow through ips -- reproduced).  mean_b(L1 ips_b y0_b) = L1 * S with S = sum_b y0_b clip(...).  One workgroup:
// the three reductions (N, then L0 / L1 / L2 / S, then the gradients that need L1 and S) run in a fixed order.
__device__ __forceinline__ float block_sum(float v, float* red) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
  __syncthreads();  // (red may still be read by the previous call's consumers)
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  float s = 0.f;
  for (int w = 0; w < ESMM_THREADS / 64; ++w) s += red[w];
  return s;
}

__device__ __forceinline__ float bce_term(float p, float y) {
  return -(y * bce_log_clamp(logf(p)) + (1.f - y) * bce_log_clamp(log1pf(-p)));
}

__global__ __launch_bounds__(ESMM_THREADS) void escm_combine_kernel(const float* praw, int64_t ldr, const float* y,
                                                                    int64_t ldy, const float* dout, int64_t lddo,
                                                                    float* pout, int64_t ldo, float* draw, int64_t lddr,
                                                                    float* loss, int64_t B, float cf_w, float global_w) {
  __shared__ float red[ESMM_THREADS / 64];
  for (int64_t b = threadIdx.x; b < B; b += ESMM_THREADS) {
    const float c = praw[b * ldr], v = praw[b * ldr + 1];
    pout[b * ldo] = c;
    pout[b * ldo + 1] = v;
    pout[b * ldo + 2] = c * v;
    if (!y && dout && draw) {  // autograd: chain rule of the three outputs
      const float d0 = dout[b * lddo], d1 = dout[b * lddo + 1], d2 = dout[b * lddo + 2];
      draw[b * lddr] = d0 + d2 * v;
      draw[b * lddr + 1] = d1 + d2 * c;
    }
  }
  if (!y) return;
  float n_part = 0.f;
  for (int64_t b = threadIdx.x; b < B; b += ESMM_THREADS) n_part += y[b * ldy];
  const float N = block_sum(n_part, red);
  float l0 = 0.f, l1 = 0.f, l2 = 0.f, sp = 0.f;
  for (int64_t b = threadIdx.x; b < B; b += ESMM_THREADS) {
    const float c = praw[b * ldr], v = praw[b * ldr + 1];
    const float y0 = y[b * ldy], y1 = y[b * ldy + 1];
    l0 += bce_term(c, y0);
    l1 += bce_term(v, y1);
    l2 += bce_term(c * v, y1);
    const float ps = fmaxf(c * N, 1e-6f);
    sp += y0 * fminf(fmaxf(1.f / ps, -15.f), 15.f);
  }
  const float L0 = block_sum(l0, red), L1 = block_sum(l1, red), L2 = block_sum(l2, red), S = block_sum(sp, red);
  if (threadIdx.x == 0 && loss) loss[0] = L0 + cf_w * (L1 * S) + global_w * L2;
  if (!draw) return;
  for (int64_t b = threadIdx.x; b < B; b += ESMM_THREADS) {
    const float c = praw[b * ldr], v = praw[b * ldr + 1];
    const float y0 = y[b * ldy], y1 = y[b * ldy + 1];
    const float p2 = c * v;
    const float g0 = (c - y0) / fmaxf((1.f - c) * c, 1e-12f);
    const float g1 = (v - y1) / fmaxf((1.f - v) * v, 1e-12f);
    const float g2 = global_w * (p2 - y1) / fmaxf((1.f - p2) * p2, 1e-12f);
    // d clip(1 / max(c N, 1e-6)) / dc: -N / (c N)^2 where neither the maximum nor the clip is active
    const float ps = c * N;
    float dips = 0.f;
    if (ps > 1e-6f) {
      const float r = 1.f / ps;
      if (r >= -15.f && r <= 15.f) dips = -N * r * r;
    }
    draw[b * lddr] = g0 + cf_w * L1 * y0 * dips + g2 * v;
    draw[b * lddr + 1] = cf_w * S * g1 + g2 * c;
  }
}

}  // namespace mml

using namespace mml;

extern "C" int mml_escm_combine(const float* p_raw, int64_t ldr, const float* y, int64_t ldy, const float* d_out,
                                int64_t lddo, float* p_out, int64_t ldo, float* d_raw, int64_t lddr, float* loss,
                                int64_t B, float cf_w, float global_w, mml_stream_t stream) {
  MML_REQUIRE(B >= 0, "mml_escm_combine: negative batch");
  if (B == 0) return MML_OK;
  MML_REQUIRE(p_raw && p_out && ldr >= 2 && ldo >= 3, "mml_escm_combine: null probabilities or leading dimension too small");
  MML_REQUIRE(!y || ldy >= 2, "mml_escm_combine: ldy < 2");
  MML_REQUIRE(!d_out || lddo >= 3, "mml_escm_combine: lddo < 3");
  MML_REQUIRE(!d_raw || lddr >= 2, "mml_escm_combine: lddr < 2");
  MML_REQUIRE(!loss || y, "mml_escm_combine: a loss needs labels");
  MML_LAUNCH(escm_combine_kernel, dim3(1), dim3(ESMM_THREADS), 0, to_stream(stream), p_raw, ldr, y, ldy, d_out, lddo,
             p_out, ldo, d_raw, lddr, loss, B, cf_w, global_w);
  return check_launch("mml_escm_combine");
}

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
