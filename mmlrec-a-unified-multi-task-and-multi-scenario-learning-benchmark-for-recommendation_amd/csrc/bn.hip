// BatchNorm1d inside DNN (reference model/utils.py:132-134, :153-154: fc -> bn -> activation), training and eval mode.
//   training: mu = mean_b z, var = biased variance over the batch, y = act(gamma (z - mu) rstd + beta), rstd = 1/sqrt(var + eps);
//             running_mean = (1 - m) running_mean + m mu, running_var = (1 - m) running_var + m var B/(B-1),
//             num_batches_tracked += 1  (torch.nn.BatchNorm1d defaults: eps 1e-5, momentum 0.1)
//   eval    : the same affine map with the running statistics.
// Column statistics over the batch in two stages (row chunks -> per-chunk partial sums in a workspace -> one ordered final
// sum in double), so the result does not depend on the launch geometry; backward likewise:
//   dbeta = sum dy, dgamma = sum dy xhat, dz = gamma rstd (dy - (dbeta + xhat dgamma) / B).
// Streaming kernels: lanes run over columns (coalesced rows), HBM-bound.
#include "common.hpp"

namespace mml {

constexpr int BN_CHUNK = 256;  // rows per partial

// partial column sums of a [B, n] matrix: P0 = sum a, P1 = sum a * b   (b = a for the forward; a = dy, b = xhat backward)
// MODE 0: b = a;  MODE 1: b = (z - mean) * rstd
template <int MODE>
__global__ __launch_bounds__(256) void bn_partial_kernel(const float* a, int64_t lda, const float* z, int64_t ldz,
                                                         const float* mean, const float* rstd, int64_t B, int n,
                                                         float* part) {
  const int64_t r0 = (int64_t)blockIdx.x * BN_CHUNK;
  const int64_t r1 = (r0 + BN_CHUNK < B) ? r0 + BN_CHUNK : B;
  for (int c = threadIdx.x; c < n; c += 256) {
    float s0 = 0.f, s1 = 0.f;
    const float mu = MODE ? mean[c] : 0.f, rs = MODE ? rstd[c] : 0.f;
    for (int64_t r = r0; r < r1; ++r) {
      const float av = a[r * lda + c];
      const float bv = MODE ? (z[r * ldz + c] - mu) * rs : av;
      s0 += av;
      s1 += av * bv;
    }
    part[((int64_t)blockIdx.x * 2) * n + c] = s0;
    part[((int64_t)blockIdx.x * 2 + 1) * n + c] = s1;
  }
}

__global__ __launch_bounds__(256) void bn_stats_final_kernel(const float* part, int nchunks, int64_t B, int n, float eps,
                                                             float momentum, float* mean, float* rstd,
                                                             float* running_mean, float* running_var, int64_t* nbt) {
  for (int c = blockIdx.x * 256 + threadIdx.x; c < n; c += gridDim.x * 256) {
    double s0 = 0.0, s1 = 0.0;
    for (int k = 0; k < nchunks; ++k) {
      s0 += part[((int64_t)k * 2) * n + c];
      s1 += part[((int64_t)k * 2 + 1) * n + c];
    }
    const double mu = s0 / (double)B;
    double var = s1 / (double)B - mu * mu;
    if (var < 0.0) var = 0.0;
    mean[c] = (float)mu;
    rstd[c] = (float)(1.0 / sqrt(var + (double)eps));
    if (running_mean) {
      const double unb = B > 1 ? var * (double)B / (double)(B - 1) : var;
      running_mean[c] = (float)((1.0 - momentum) * running_mean[c] + momentum * mu);
      running_var[c] = (float)((1.0 - momentum) * running_var[c] + momentum * unb);
    }
  }
  if (nbt && blockIdx.x == 0 && threadIdx.x == 0) nbt[0] += 1;
}

__global__ __launch_bounds__(256) void bn_eval_stats_kernel(const float* running_mean, const float* running_var, int n,
                                                            float eps, float* mean, float* rstd) {
  for (int c = blockIdx.x * 256 + threadIdx.x; c < n; c += gridDim.x * 256) {
    mean[c] = running_mean[c];
    rstd[c] = 1.f / sqrtf(running_var[c] + eps);
  }
}

__global__ __launch_bounds__(256) void bn_apply_kernel(const float* z, int64_t ldz, const float* mean, const float* rstd,
                                                       const float* gamma, const float* beta, float* y, int64_t ldy,
                                                       int64_t B, int n, int act) {
  const int64_t total = B * n;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int64_t r = i / n;
    const int c = (int)(i - r * n);
    float v = (z[r * ldz + c] - mean[c]) * rstd[c] * gamma[c] + beta[c];
    if (act == MML_ACT_RELU) v = v > 0.f ? v : 0.f;
    else if (act == MML_ACT_SIGMOID) v = 1.f / (1.f + expf(-v));
    else if (act == MML_ACT_SIGMOID2) v = 2.f / (1.f + expf(-v));
    y[r * ldy + c] = v;
  }
}

__global__ __launch_bounds__(256) void bn_bwd_final_kernel(const float* part, int nchunks, int n, float* dgamma,
                                                           float* dbeta, int acc, float* sums) {
  for (int c = blockIdx.x * 256 + threadIdx.x; c < n; c += gridDim.x * 256) {
    double s0 = 0.0, s1 = 0.0;
    for (int k = 0; k < nchunks; ++k) {
      s0 += part[((int64_t)k * 2) * n + c];
      s1 += part[((int64_t)k * 2 + 1) * n + c];
    }
    sums[c] = (float)s0;      // sum dy
    sums[n + c] = (float)s1;  // sum dy * xhat
    dbeta[c] = acc ? dbeta[c] + (float)s0 : (float)s0;
    dgamma[c] = acc ? dgamma[c] + (float)s1 : (float)s1;
  }
}

__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* dy, int64_t lddy, const float* z, int64_t ldz,
                                                           const float* mean, const float* rstd, const float* gamma,
                                                           const float* sums, float* dz, int64_t lddz, int64_t B, int n) {
  const int64_t total = B * n;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const float invB = 1.f / (float)B;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int64_t r = i / n;
    const int c = (int)(i - r * n);
    const float xh = (z[r * ldz + c] - mean[c]) * rstd[c];
    dz[r * lddz + c] = gamma[c] * rstd[c] * (dy[r * lddy + c] - (sums[c] + xh * sums[n + c]) * invB);
  }
}

static unsigned bn_grid(int64_t n) {
  int64_t b = cdiv(n, 256);
  if (b > 256 * 8) b = 256 * 8;
  return (unsigned)(b < 1 ? 1 : b);
}

}  // namespace mml

using namespace mml;

extern "C" int64_t mml_bn_workspace_bytes(int64_t B, int32_t n) {
  if (B <= 0 || n <= 0) return 0;
  return (cdiv(B, (int64_t)BN_CHUNK) * 2 * n + 2 * (int64_t)n) * 4;
}

extern "C" int mml_bn_fwd(const float* z, int64_t ldz, const float* gamma, const float* beta, float* running_mean,
                          float* running_var, int64_t* num_batches_tracked, float* mean, float* rstd, float* y,
                          int64_t ldy, int64_t B, int32_t n, int32_t act, int32_t training, float eps, float momentum,
                          void* workspace, int64_t workspace_bytes, mml_stream_t stream) {
  MML_REQUIRE(B >= 0 && n >= 0, "mml_bn_fwd: negative extent");
  if (B == 0 || n == 0) return MML_OK;
  MML_REQUIRE(z && gamma && beta && mean && rstd && y && ldz >= n && ldy >= n, "mml_bn_fwd: null argument or ld < n");
  MML_REQUIRE(running_mean && running_var, "mml_bn_fwd: running statistics are required");
  MML_REQUIRE(act >= MML_ACT_NONE && act <= MML_ACT_SIGMOID2, "mml_bn_fwd: unknown activation");
  hipStream_t st = to_stream(stream);
  if (training) {
    MML_REQUIRE(workspace && workspace_bytes >= mml_bn_workspace_bytes(B, n), "mml_bn_fwd: workspace too small");
    const int nch = (int)cdiv(B, (int64_t)BN_CHUNK);
    float* part = static_cast<float*>(workspace);
    MML_LAUNCH(bn_partial_kernel<0>, dim3((unsigned)nch), dim3(256), 0, st, z, ldz, (const float*)nullptr, (int64_t)0,
               (const float*)nullptr, (const float*)nullptr, B, (int)n, part);
    MML_LAUNCH(bn_stats_final_kernel, dim3(bn_grid(n)), dim3(256), 0, st, part, nch, B, (int)n, eps, momentum, mean, rstd,
               running_mean, running_var, num_batches_tracked);
  } else {
    MML_LAUNCH(bn_eval_stats_kernel, dim3(bn_grid(n)), dim3(256), 0, st, running_mean, running_var, (int)n, eps, mean, rstd);
  }
  MML_LAUNCH(bn_apply_kernel, dim3(bn_grid(B * n)), dim3(256), 0, st, z, ldz, mean, rstd, gamma, beta, y, ldy, B, (int)n,
             (int)act);
  return check_launch("mml_bn_fwd");
}

extern "C" int mml_bn_bwd(const float* dy, int64_t lddy, const float* z, int64_t ldz, const float* gamma, const float* mean,
                          const float* rstd, float* dz, int64_t lddz, float* dgamma, float* dbeta, int32_t accumulate,
                          int64_t B, int32_t n, void* workspace, int64_t workspace_bytes, mml_stream_t stream) {
  MML_REQUIRE(B >= 0 && n >= 0, "mml_bn_bwd: negative extent");
  if (B == 0 || n == 0) return MML_OK;
  MML_REQUIRE(dy && z && gamma && mean && rstd && dz && dgamma && dbeta && lddy >= n && ldz >= n && lddz >= n,
              "mml_bn_bwd: null argument or ld < n");
  MML_REQUIRE(workspace && workspace_bytes >= mml_bn_workspace_bytes(B, n), "mml_bn_bwd: workspace too small");
  hipStream_t st = to_stream(stream);
  const int nch = (int)cdiv(B, (int64_t)BN_CHUNK);
  float* part = static_cast<float*>(workspace);
  float* sums = part + (int64_t)nch * 2 * n;
  MML_LAUNCH(bn_partial_kernel<1>, dim3((unsigned)nch), dim3(256), 0, st, dy, lddy, z, ldz, mean, rstd, B, (int)n, part);
  MML_LAUNCH(bn_bwd_final_kernel, dim3(bn_grid(n)), dim3(256), 0, st, part, nch, (int)n, dgamma, dbeta, (int)accumulate, sums);
  MML_LAUNCH(bn_bwd_apply_kernel, dim3(bn_grid(B * n)), dim3(256), 0, st, dy, lddy, z, ldz, mean, rstd, gamma, sums, dz,
             lddz, B, (int)n);
  return check_launch("mml_bn_bwd");
}
