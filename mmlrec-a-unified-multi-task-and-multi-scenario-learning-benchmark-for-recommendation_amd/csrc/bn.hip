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

// partial column statistics of a [B, n] matrix over one chunk of rows.
// MODE 0 (forward): P0 = mean of the chunk, P1 = sum of squared deviations from that mean (Welford in double: the
//         E[z^2] - mu^2 form cancels catastrophically when |mean| >> std; torch uses Welford too);
// MODE 1 (backward): P0 = sum a, P1 = sum a * xhat with xhat = (z - mean) * rstd   (a = dy).
template <int MODE>
__global__ __launch_bounds__(256) void bn_partial_kernel(const float* a, int64_t lda, const float* z, int64_t ldz,
                                                         const float* mean, const float* rstd, int64_t B, int n,
                                                         float* part) {
  const int64_t r0 = (int64_t)blockIdx.x * BN_CHUNK;
  const int64_t r1 = (r0 + BN_CHUNK < B) ? r0 + BN_CHUNK : B;
  for (int c = threadIdx.x; c < n; c += 256) {
    if (MODE == 0) {
      double mu = 0.0, m2 = 0.0;
      int cnt = 0;
      for (int64_t r = r0; r < r1; ++r) {
        const double x = a[r * lda + c];
        ++cnt;
        const double d = x - mu;
        mu += d / cnt;
        m2 += d * (x - mu);
      }
      part[((int64_t)blockIdx.x * 2) * n + c] = (float)mu;
      part[((int64_t)blockIdx.x * 2 + 1) * n + c] = (float)m2;
    } else {
      float s0 = 0.f, s1 = 0.f;
      const float mu = mean[c], rs = rstd[c];
      for (int64_t r = r0; r < r1; ++r) {
        const float av = a[r * lda + c];
        s0 += av;
        s1 += av * ((z[r * ldz + c] - mu) * rs);
      }
      part[((int64_t)blockIdx.x * 2) * n + c] = s0;
      part[((int64_t)blockIdx.x * 2 + 1) * n + c] = s1;
    }
  }
}

// chunk statistics -> batch statistics by Chan's parallel-variance merge, in chunk order, in double
__global__ __launch_bounds__(256) void bn_stats_final_kernel(const float* part, int nchunks, int64_t B, int n, float eps,
                                                             float momentum, float* mean, float* rstd,
                                                             float* running_mean, float* running_var, int64_t* nbt) {
  for (int c = blockIdx.x * 256 + threadIdx.x; c < n; c += gridDim.x * 256) {
    double mu = 0.0, m2 = 0.0, cnt = 0.0;
    for (int k = 0; k < nchunks; ++k) {
      const int64_t r0 = (int64_t)k * BN_CHUNK;
      const double nk = (double)((r0 + BN_CHUNK < B ? r0 + BN_CHUNK : B) - r0);
      const double mk = part[((int64_t)k * 2) * n + c], m2k = part[((int64_t)k * 2 + 1) * n + c];
      const double d = mk - mu, tot = cnt + nk;
      mu += d * nk / tot;
      m2 += m2k + d * d * cnt * nk / tot;
      cnt = tot;
    }
    const double var = m2 / (double)B;
    mean[c] = (float)mu;
    rstd[c] = (float)(1.0 / sqrt(var + (double)eps));
    if (running_mean) {
      const double unb = B > 1 ? m2 / (double)(B - 1) : var;
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

// ---- DomainBatchNorm (reference model/utils.py:553-636; STAR's per-domain BN, model/star.py:32-33, :50-51) ----------
// What the reference computes (probed): gamma / beta are unregistered lists frozen at (1, 0).
//   training: every domain's branch normalises with the statistics of the WHOLE batch (F.batch_norm(..., training=True)
//     ignores the tensors passed as running statistics), so  out = sum_d mask[:, d] * BN_batch(x) = BN_batch(x) for
//     one-hot masks (mml_bn_fwd with gamma 1 / beta 0); side effect: for every domain d
//         pop_mean[d] = decay pop_mean[d] + (1 - decay) mean(x[domain == d]),  pop_var likewise with the UNBIASED variance
//     -- evaluated for EVERY domain (both arguments of the reference's torch.where are computed), so a domain with no
//     sample in the batch turns its population statistics into NaN, and one with a single sample its variance.
//   eval: out[b] = sum_d mask[b, d] * (x[b] - pop_mean[d]) / sqrt(pop_var[d] + 1e-5).
// blockIdx.y = domain; a thread owns a column and walks the batch (Welford in double).
__global__ __launch_bounds__(256) void domain_bn_update_kernel(const float* x, int64_t ldx, const float* mask,
                                                               int64_t ldm, int64_t B, int n, int D, float* pop_mean,
                                                               float* pop_var, float decay) {
  const int d = blockIdx.y;
  for (int c = blockIdx.x * 256 + threadIdx.x; c < n; c += gridDim.x * 256) {
    double mu = 0.0, m2 = 0.0;
    int64_t cnt = 0;
    for (int64_t b = 0; b < B; ++b) {
      int arg = 0;  // torch.argmax: first maximal entry
      float best = mask[b * ldm];
      for (int j = 1; j < D; ++j) {
        const float v = mask[b * ldm + j];
        if (v > best) { best = v; arg = j; }
      }
      if (arg != d) continue;
      const double v = x[b * ldx + c];
      ++cnt;
      const double dl = v - mu;
      mu += dl / (double)cnt;
      m2 += dl * (v - mu);
    }
    const float nanf_ = __int_as_float(0x7fc00000);
    const float bm = cnt > 0 ? (float)mu : nanf_;
    const float bv = cnt > 1 ? (float)(m2 / (double)(cnt - 1)) : nanf_;
    pop_mean[(int64_t)d * n + c] = pop_mean[(int64_t)d * n + c] * decay + bm * (1.f - decay);
    pop_var[(int64_t)d * n + c] = pop_var[(int64_t)d * n + c] * decay + bv * (1.f - decay);
  }
}

__global__ __launch_bounds__(256) void domain_bn_eval_kernel(const float* x, int64_t ldx, const float* mask, int64_t ldm,
                                                             const float* pop_mean, const float* pop_var, float* y,
                                                             int64_t ldy, int64_t B, int n, int D, float eps) {
  const int64_t total = B * n;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int64_t b = i / n;
    const int c = (int)(i - b * n);
    const float v = x[b * ldx + c];
    float acc = 0.f;
    for (int d = 0; d < D; ++d) {
      const float m = mask[b * ldm + d];
      acc += m * ((v - pop_mean[(int64_t)d * n + c]) / sqrtf(pop_var[(int64_t)d * n + c] + eps));
    }
    y[b * ldy + c] = acc;
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
    // torch: "Expected more than 1 value per channel when training" (a ragged last batch of one sample)
    MML_REQUIRE(B > 1, "mml_bn_fwd: training-mode BatchNorm needs more than 1 sample per batch (got %lld)", (long long)B);
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

extern "C" int mml_domain_bn_update(const float* x, int64_t ldx, const float* mask, int64_t ldm, int64_t B, int32_t n,
                                    int32_t D, float* pop_mean, float* pop_var, float decay, mml_stream_t stream) {
  MML_REQUIRE(B >= 0 && n >= 0 && D > 0, "mml_domain_bn_update: bad sizes");
  if (n == 0) return MML_OK;
  MML_REQUIRE(x && mask && pop_mean && pop_var && ldx >= n && ldm >= D, "mml_domain_bn_update: null argument or ld too small");
  MML_LAUNCH(domain_bn_update_kernel, dim3((unsigned)cdiv(n, 256), (unsigned)D), dim3(256), 0, to_stream(stream), x, ldx,
             mask, ldm, B, (int)n, (int)D, pop_mean, pop_var, decay);
  return check_launch("mml_domain_bn_update");
}

extern "C" int mml_domain_bn_eval(const float* x, int64_t ldx, const float* mask, int64_t ldm, const float* pop_mean,
                                  const float* pop_var, float* y, int64_t ldy, int64_t B, int32_t n, int32_t D, float eps,
                                  mml_stream_t stream) {
  MML_REQUIRE(B >= 0 && n >= 0 && D > 0, "mml_domain_bn_eval: bad sizes");
  if (B == 0 || n == 0) return MML_OK;
  MML_REQUIRE(x && mask && pop_mean && pop_var && y && ldx >= n && ldy >= n && ldm >= D,
              "mml_domain_bn_eval: null argument or ld too small");
  MML_LAUNCH(domain_bn_eval_kernel, dim3(bn_grid(B * n)), dim3(256), 0, to_stream(stream), x, ldx, mask, ldm, pop_mean,
             pop_var, y, ldy, B, (int)n, (int)D, eps);
  return check_launch("mml_domain_bn_eval");
}
