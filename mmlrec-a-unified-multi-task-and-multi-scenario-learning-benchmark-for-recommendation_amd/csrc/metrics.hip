// Per-batch AUC of an epoch's predictions, on the device.
//
// The reference computes sklearn.metrics.roc_auc_score on the host for EVERY training step (model/basemodel.py:316-331:
// y_pred.cpu() -> numpy -> metric, then the epoch log averages the per-step values, :335-337).  Here the predictions of
// the whole epoch stay in HBM and one launch produces the AUC of every (batch, column) pair: one workgroup sorts one
// batch column in LDS (bitonic, 64-bit keys = order-preserving image of the fp32 prediction with the label in the
// lowest bit) and evaluates the Mann-Whitney statistic with sklearn's tie handling -- a tie group contributes half of
// its negative count to each of its positives -- in exact integer arithmetic:
//     AUC = sum over positives p of (2 * #neg(pred < p) + #neg(pred == p)) / (2 * n_pos * n_neg)
// which is what the trapezoidal area under sklearn's ROC curve evaluates to.  Integer / compare work on data that is
// already resident: bounded by LDS sort passes, not by HBM.
#include "common.hpp"

namespace mml {

constexpr int AUC_THREADS = 256;
constexpr int AUC_MAX_SEG = 4096;

__device__ __forceinline__ uint32_t orderable(float p) {
  if (p == 0.f) p = 0.f;  // -0 and +0 compare equal on the host
  const uint32_t u = __float_as_uint(p);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// inclusive scan over AUC_THREADS per-thread partials (Hillis-Steele in LDS); OP: 0 = add, 1 = max
template <int OP>
__device__ __forceinline__ uint32_t block_scan_excl(uint32_t v, uint32_t* tmp) {
  const int t = threadIdx.x;
  tmp[t] = v;
  __syncthreads();
  for (int d = 1; d < AUC_THREADS; d <<= 1) {
    uint32_t o = (t >= d) ? tmp[t - d] : 0u;
    __syncthreads();
    if (t >= d) tmp[t] = OP ? (tmp[t] > o ? tmp[t] : o) : tmp[t] + o;
    __syncthreads();
  }
  const uint32_t ex = t ? tmp[t - 1] : 0u;
  __syncthreads();
  return ex;
}

__global__ __launch_bounds__(AUC_THREADS) void auc_segments_kernel(const float* pred, int64_t ldp, const float* y,
                                                                   int64_t ldy, int64_t n, int cols, int seg, int M,
                                                                   double* auc) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  uint64_t* key = reinterpret_cast<uint64_t*>(smem_raw);            // [M]
  uint16_t* cneg = reinterpret_cast<uint16_t*>(key + M);            // [M] negatives in [0, i)
  uint16_t* gstart = cneg + M;                                      // [M] first index of i's tie group
  __shared__ uint32_t tmp[AUC_THREADS];
  __shared__ unsigned long long total2;
  __shared__ uint32_t npos_s;
  const int s = blockIdx.x / cols, c = blockIdx.x - s * cols;
  const int64_t r0 = (int64_t)s * seg;
  const int len = (int)((n - r0 < seg) ? (n - r0) : seg);
  const int t = threadIdx.x;
  if (t == 0) { total2 = 0ull; npos_s = 0u; }
  for (int i = t; i < M; i += AUC_THREADS) {
    uint64_t k = ~0ull;  // padding sorts last
    if (i < len) {
      const float p = pred[(r0 + i) * ldp + c];
      const uint32_t lab = y[(r0 + i) * ldy + c] > 0.5f ? 1u : 0u;
      k = ((uint64_t)orderable(p) << 1) | lab;
    }
    key[i] = k;
  }
  __syncthreads();
  // bitonic sort, ascending
  for (int k2 = 2; k2 <= M; k2 <<= 1) {
    for (int j = k2 >> 1; j > 0; j >>= 1) {
      for (int i = t; i < M; i += AUC_THREADS) {
        const int l = i ^ j;
        if (l > i) {
          const uint64_t a = key[i], b = key[l];
          const bool up = (i & k2) == 0;
          if ((a > b) == up) { key[i] = b; key[l] = a; }
        }
      }
      __syncthreads();
    }
  }
  // each thread owns CH consecutive sorted positions
  const int CH = M / AUC_THREADS > 0 ? M / AUC_THREADS : 1;
  const int i0 = t * CH;
  const bool own = i0 < M;
  uint32_t nneg = 0, gmax = 0;
  if (own)
    for (int i = i0; i < i0 + CH && i < M; ++i) {
      if (i < len && !(key[i] & 1ull)) ++nneg;
      if (i > 0 && (key[i] >> 1) != (key[i - 1] >> 1)) gmax = (uint32_t)i;
    }
  const uint32_t neg_before = block_scan_excl<0>(own ? nneg : 0u, tmp);
  const uint32_t g_before = block_scan_excl<1>(own ? gmax : 0u, tmp);
  if (own) {
    uint32_t cn = neg_before, gs = g_before;
    for (int i = i0; i < i0 + CH && i < M; ++i) {
      if (i > 0 && (key[i] >> 1) != (key[i - 1] >> 1)) gs = (uint32_t)i;
      cneg[i] = (uint16_t)cn;
      gstart[i] = (uint16_t)gs;
      if (i < len && !(key[i] & 1ull)) ++cn;
    }
  }
  __syncthreads();
  unsigned long long part = 0ull;
  uint32_t np = 0;
  for (int i = t; i < len; i += AUC_THREADS) {
    if (key[i] & 1ull) {
      const uint32_t less = cneg[gstart[i]];
      const uint32_t eq = cneg[i] - less;  // negatives of a tie group sort in front of its positives
      part += 2ull * less + eq;
      ++np;
    }
  }
  atomicAdd(&total2, part);  // integer sums: order-independent
  atomicAdd(&npos_s, np);
  __syncthreads();
  if (t == 0) {
    const double npos = (double)npos_s, nn = (double)(len - (int)npos_s);
    auc[(int64_t)s * cols + c] = (npos_s == 0 || (int)npos_s == len) ? __longlong_as_double(0x7ff8000000000000ll)
                                                                    : (double)total2 / (2.0 * npos * nn);
  }
}

}  // namespace mml

using namespace mml;

extern "C" int mml_auc_segments(const float* pred, int64_t ldp, const float* y, int64_t ldy, int64_t n, int32_t cols,
                                int32_t seg, double* auc, mml_stream_t stream) {
  MML_REQUIRE(n >= 0 && cols >= 1 && seg >= 1, "mml_auc_segments: bad extents n=%lld cols=%d seg=%d", (long long)n, cols,
              seg);
  if (n == 0) return MML_OK;
  MML_REQUIRE(pred && y && auc && ldp >= cols && ldy >= cols, "mml_auc_segments: null pointer or leading dimension < cols");
  if (seg > AUC_MAX_SEG) {
    set_error("mml_auc_segments: segments of more than %d rows are not supported (got %d)", AUC_MAX_SEG, seg);
    return MML_ERR_UNSUPPORTED;
  }
  int M = AUC_THREADS;
  while (M < seg) M <<= 1;
  const int64_t nseg = cdiv(n, (int64_t)seg);
  MML_REQUIRE(nseg * cols <= 0x7fffffff, "mml_auc_segments: grid too large");
  const size_t lds = (size_t)M * (8 + 2 + 2);
  MML_LAUNCH(auc_segments_kernel, dim3((unsigned)(nseg * cols)), dim3(AUC_THREADS), lds, to_stream(stream), pred, ldp, y,
             ldy, n, (int)cols, (int)seg, M, auc);
  return check_launch("mml_auc_segments");
}
