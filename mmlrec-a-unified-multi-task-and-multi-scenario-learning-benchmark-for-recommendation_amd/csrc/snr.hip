// Sub-network-routing gate of SNR-trans (reference model/snr_trans.py:8-50): every (output o, input j) pair owns a fixed
// [units, units] transformation M_oj (plain Python lists in the reference: never registered, never optimised) scaled by
// a learned hard-concrete coefficient
//     s = sigmoid(log u - log(1 - u) + log(alpha) / beta),  z = clamp(s * (eps - gamma) + gamma, 0, 1).
// out_o = sum_j z_oj (x_j @ M_oj) = [x_0 | x_1 | ...] @ [z_o0 M_o0 ; z_o1 M_o1 ; ...]: the scaled blocks are
// materialised once per step (fwd kernel below) and the routing becomes ONE [K,N]-layout GEMM per output; the
// gradient of z_oj is the Frobenius product <dW'_oj, M_oj> of the weight gradient that GEMM's wgrad produces (bwd).
// A few hundred kB of parameters per step: one workgroup per block forward; one workgroup in total backward, so the
// sums (and d alpha, which collects every block) have a fixed order.
#include "common.hpp"

namespace mml {

struct SnrConsts {
  float beta, gamma, eps;
};

__device__ __forceinline__ float snr_z(float u, float alpha, const SnrConsts& c, float* dz_du, float* dz_dalpha) {
  const float logit = logf(u) - logf(1.f - u) + logf(alpha) / c.beta;
  const float s = 1.f / (1.f + expf(-logit));
  const float sb = s * (c.eps - c.gamma) + c.gamma;
  const bool live = sb > 0.f && sb <= 1.f;  // the reference's (s_ > 0) * s_, then (z > 1) + (z <= 1) * z
  const float z = sb <= 0.f ? 0.f : (sb > 1.f ? 1.f : sb);
  if (dz_du) {
    const float ds = live ? (c.eps - c.gamma) * s * (1.f - s) : 0.f;
    *dz_du = ds * (1.f / u + 1.f / (1.f - u));
    *dz_dalpha = ds / (alpha * c.beta);
  }
  return z;
}

// zw = 1: one coefficient per block (SNR-trans); zw = units: one per OUTPUT COLUMN of the block (MSSM, whose u is a
// [units] vector per pair, model/mssm.py:26-29): element e of a row-major [in, out] block belongs to column e % zw
__global__ __launch_bounds__(256) void snr_weights_fwd_kernel(const float* u, const float* alpha, const float* M, float* W,
                                                              int64_t block, int zw, SnrConsts c) {
  const float* m = M + (int64_t)blockIdx.x * block;
  float* w = W + (int64_t)blockIdx.x * block;
  if (zw == 1) {
    const float z = snr_z(u[blockIdx.x], alpha[0], c, nullptr, nullptr);
    for (int64_t i = threadIdx.x; i < block; i += 256) w[i] = z * m[i];
  } else {
    const float* ub = u + (int64_t)blockIdx.x * zw;
    for (int64_t i = threadIdx.x; i < block; i += 256) w[i] = snr_z(ub[i % zw], alpha[0], c, nullptr, nullptr) * m[i];
  }
}

// backward, stage 1: one workgroup per block.  zw == 1: dz = <dW_b, M_b>; zw > 1: one dz per output column (thread t owns
// columns t, t + 256, ...).  Writes du (when u learns) and the block's contribution to d alpha into part[b].
__global__ __launch_bounds__(256) void snr_weights_bwd_kernel(const float* dW, const float* M, const float* u,
                                                              const float* alpha, float* du, float* part, int acc_u,
                                                              int64_t block, int zw, SnrConsts c) {
  __shared__ float red[4];
  const int b = blockIdx.x;
  const float* g = dW + (int64_t)b * block;
  const float* m = M + (int64_t)b * block;
  float acc = 0.f;  // zw == 1: the block's dot product; zw > 1: this thread's share of d alpha
  if (zw == 1) {
    for (int64_t i = threadIdx.x; i < block; i += 256) acc += g[i] * m[i];
  } else {
    const int rows = (int)(block / zw);
    for (int col = threadIdx.x; col < zw; col += 256) {
      float dz = 0.f;
      for (int k = 0; k < rows; ++k) dz += g[(int64_t)k * zw + col] * m[(int64_t)k * zw + col];
      float dzu, dza;
      snr_z(u[(int64_t)b * zw + col], alpha[0], c, &dzu, &dza);
      if (du) {
        const float v = dz * dzu;
        du[(int64_t)b * zw + col] = acc_u ? du[(int64_t)b * zw + col] + v : v;
      }
      acc += dz * dza;
    }
  }
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float t = red[0] + red[1] + red[2] + red[3];
    if (zw == 1) {
      float dzu, dza;
      snr_z(u[b], alpha[0], c, &dzu, &dza);
      const float v = t * dzu;
      du[b] = acc_u ? du[b] + v : v;
      part[b] = t * dza;
    } else {
      part[b] = t;
    }
  }
}

// stage 2: d alpha = sum of the per-block contributions, in block order
__global__ void snr_alpha_kernel(const float* part, float* dalpha, int nblocks, int acc_alpha) {
  float s = 0.f;
  for (int b = 0; b < nblocks; ++b) s += part[b];
  dalpha[0] = acc_alpha ? dalpha[0] + s : s;
}

}  // namespace mml

using namespace mml;

extern "C" int mml_snr_gate_weights_fwd(const float* u, const float* alpha, const float* M, float* W, int32_t n_blocks,
                                        int64_t block, int32_t zw, float beta, float gamma, float eps,
                                        mml_stream_t stream) {
  MML_REQUIRE(n_blocks >= 0 && block >= 0, "mml_snr_gate_weights_fwd: negative extent");
  if (n_blocks == 0 || block == 0) return MML_OK;
  MML_REQUIRE(u && alpha && M && W && beta != 0.f, "mml_snr_gate_weights_fwd: null argument");
  MML_REQUIRE(zw >= 1 && block % zw == 0, "mml_snr_gate_weights_fwd: zw must divide the block size");
  MML_LAUNCH(snr_weights_fwd_kernel, dim3((unsigned)n_blocks), dim3(256), 0, to_stream(stream), u, alpha, M, W, block,
             (int)zw, SnrConsts{beta, gamma, eps});
  return check_launch("mml_snr_gate_weights_fwd");
}

extern "C" int mml_snr_gate_weights_bwd(const float* dW, const float* M, const float* u, const float* alpha, float* du,
                                        float* dalpha, int32_t acc_u, int32_t acc_alpha, int32_t n_blocks, int64_t block,
                                        int32_t zw, float beta, float gamma, float eps, float* workspace,
                                        mml_stream_t stream) {
  MML_REQUIRE(n_blocks >= 0 && block >= 0, "mml_snr_gate_weights_bwd: negative extent");
  if (n_blocks == 0) return MML_OK;
  MML_REQUIRE(dW && M && u && alpha && dalpha && workspace && beta != 0.f, "mml_snr_gate_weights_bwd: null argument");
  MML_REQUIRE(zw >= 1 && block % zw == 0, "mml_snr_gate_weights_bwd: zw must divide the block size");
  MML_REQUIRE(du || zw > 1, "mml_snr_gate_weights_bwd: du may only be null for frozen per-column coefficients (zw > 1)");
  MML_LAUNCH(snr_weights_bwd_kernel, dim3((unsigned)n_blocks), dim3(256), 0, to_stream(stream), dW, M, u, alpha, du,
             workspace, (int)acc_u, block, (int)zw, SnrConsts{beta, gamma, eps});
  int rc = check_launch("mml_snr_gate_weights_bwd");
  if (rc) return rc;
  MML_LAUNCH(snr_alpha_kernel, dim3(1), dim3(1), 0, to_stream(stream), workspace, dalpha, (int)n_blocks, (int)acc_alpha);
  return check_launch("mml_snr_gate_weights_bwd");
}
