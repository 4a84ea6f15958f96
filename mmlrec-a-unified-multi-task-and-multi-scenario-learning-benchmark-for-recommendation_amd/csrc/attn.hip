// Two-token attention of AITM (reference model/aitm.py:84-93): for every sample, tokens t in {0, 1} (the transferred
// feature of the previous task and the task's own feature) carry V_t, K_t, Q_t in R^H;
//     s_t = <K_t, Q_t> / sqrt(H),   a = softmax(s_0, s_1),   out = a_0 V_0 + a_1 V_1.
// One wavefront per sample, lanes stride over the H columns (any H, any alignment); the two dot products finish with a
// 6-step butterfly.  Bandwidth-bound row kernel (7 rows of H floats in, 1 out; backward 8 in, 6 out).
#include "common.hpp"

namespace mml {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

__global__ __launch_bounds__(256) void attn2_fwd_kernel(const mml_attn2_desc d) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nw = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t b = wave; b < d.B; b += nw) {
    float s0 = 0.f, s1 = 0.f;
    for (int c = lane; c < d.H; c += 64) {
      s0 += d.K[0][b * d.ldk[0] + c] * d.Q[0][b * d.ldq[0] + c];
      s1 += d.K[1][b * d.ldk[1] + c] * d.Q[1][b * d.ldq[1] + c];
    }
    s0 = wave_sum(s0) / d.sqrt_h;
    s1 = wave_sum(s1) / d.sqrt_h;
    const float m = fmaxf(s0, s1);
    const float e0 = expf(s0 - m), e1 = expf(s1 - m);
    const float a0 = e0 / (e0 + e1), a1 = e1 / (e0 + e1);
    if (lane == 0 && d.A) {
      d.A[b * 2] = a0;
      d.A[b * 2 + 1] = a1;
    }
    for (int c = lane; c < d.H; c += 64)
      d.out[b * d.ldo + c] = a0 * d.V[0][b * d.ldv[0] + c] + a1 * d.V[1][b * d.ldv[1] + c];
  }
}

__global__ __launch_bounds__(256) void attn2_bwd_kernel(const mml_attn2_desc d) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nw = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t b = wave; b < d.B; b += nw) {
    const float a0 = d.A[b * 2], a1 = d.A[b * 2 + 1];
    float g0 = 0.f, g1 = 0.f;
    for (int c = lane; c < d.H; c += 64) {
      const float go = d.dout[b * d.lddo + c];
      g0 += go * d.V[0][b * d.ldv[0] + c];
      g1 += go * d.V[1][b * d.ldv[1] + c];
    }
    g0 = wave_sum(g0);
    g1 = wave_sum(g1);
    const float dot = a0 * g0 + a1 * g1;
    const float ds0 = a0 * (g0 - dot) / d.sqrt_h, ds1 = a1 * (g1 - dot) / d.sqrt_h;
    for (int c = lane; c < d.H; c += 64) {
      const float go = d.dout[b * d.lddo + c];
      d.dV[0][b * d.lddv[0] + c] = a0 * go;
      d.dV[1][b * d.lddv[1] + c] = a1 * go;
      d.dK[0][b * d.lddk[0] + c] = ds0 * d.Q[0][b * d.ldq[0] + c];
      d.dQ[0][b * d.lddq[0] + c] = ds0 * d.K[0][b * d.ldk[0] + c];
      d.dK[1][b * d.lddk[1] + c] = ds1 * d.Q[1][b * d.ldq[1] + c];
      d.dQ[1][b * d.lddq[1] + c] = ds1 * d.K[1][b * d.ldk[1] + c];
    }
  }
}

static int check_desc(const mml_attn2_desc* d, bool bwd, const char* who) {
  MML_REQUIRE(d, "%s: null descriptor", who);
  MML_REQUIRE(d->B >= 0 && d->H >= 1 && d->sqrt_h > 0.f, "%s: bad extents", who);
  for (int t = 0; t < 2; ++t) {
    MML_REQUIRE(d->V[t] && d->K[t] && d->Q[t] && d->ldv[t] >= d->H && d->ldk[t] >= d->H && d->ldq[t] >= d->H,
                "%s: token %d: null operand or leading dimension < H", who, t);
    if (bwd)
      MML_REQUIRE(d->dV[t] && d->dK[t] && d->dQ[t] && d->lddv[t] >= d->H && d->lddk[t] >= d->H && d->lddq[t] >= d->H,
                  "%s: token %d: null gradient or leading dimension < H", who, t);
  }
  if (bwd) MML_REQUIRE(d->A && d->dout && d->lddo >= d->H, "%s: backward needs A and dout", who);
  else MML_REQUIRE(d->out && d->ldo >= d->H, "%s: null output", who);
  return MML_OK;
}

static unsigned attn_grid(int64_t B) {
  int64_t blocks = cdiv(B, 4);  // four waves (samples) per workgroup
  if (blocks > 256 * 8) blocks = 256 * 8;
  return (unsigned)(blocks < 1 ? 1 : blocks);
}

}  // namespace mml

using namespace mml;

extern "C" int mml_attn2_fwd(const mml_attn2_desc* d, mml_stream_t stream) {
  int rc = check_desc(d, false, "mml_attn2_fwd");
  if (rc) return rc;
  if (d->B == 0) return MML_OK;
  MML_LAUNCH(attn2_fwd_kernel, dim3(attn_grid(d->B)), dim3(256), 0, to_stream(stream), *d);
  return check_launch("mml_attn2_fwd");
}

extern "C" int mml_attn2_bwd(const mml_attn2_desc* d, mml_stream_t stream) {
  int rc = check_desc(d, true, "mml_attn2_bwd");
  if (rc) return rc;
  if (d->B == 0) return MML_OK;
  MML_LAUNCH(attn2_bwd_kernel, dim3(attn_grid(d->B)), dim3(256), 0, to_stream(stream), *d);
  return check_launch("mml_attn2_bwd");
}
