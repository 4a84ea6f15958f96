/* C/OpenMP restatement of the oracle's three memory-bound loops (TEST INFRASTRUCTURE, same status as
 * mmlrec_oracle.py: only tests, smoke() and bench.py's cpu_baseline leg may use it).  Semantics are those of the numpy
 * functions of the same name in mmlrec_oracle.py; tests/test_oracle_fast.py checks the two against each other.
 *   gather_fields : input_from_feature_columns + combined_dnn_input (model/basemodel.py:461-487, model/utils.py:434-446)
 *   scatter_fields: embedding_dense_backward, duplicates accumulate in batch order (model/basemodel.py:122)
 *   adam_dense / adagrad_dense: torch.optim defaults over a whole tensor (model/basemodel.py:313, :569-584)
 */
#include <math.h>
#include <stdint.h>

void gather_fields(const float* const* tabs, const int64_t* idx, int64_t B, int F, int E, float* out, int64_t ldo) {
#pragma omp parallel for schedule(static)
  for (int64_t b = 0; b < B; ++b)
    for (int f = 0; f < F; ++f) {
      const float* src = tabs[f] + idx[b * F + f] * E;
      float* dst = out + b * ldo + (int64_t)f * E;
      for (int e = 0; e < E; ++e) dst[e] = src[e];
    }
}

/* one thread per field: inside a field rows are added in batch order (bitwise the np.add.at result) */
void scatter_fields(float* const* gtabs, const int64_t* idx, int64_t B, int F, int E, const float* d, int64_t ldd) {
#pragma omp parallel for schedule(dynamic, 1)
  for (int f = 0; f < F; ++f) {
    float* g = gtabs[f];
    for (int64_t b = 0; b < B; ++b) {
      float* dst = g + idx[b * F + f] * E;
      const float* src = d + b * ldd + (int64_t)f * E;
      for (int e = 0; e < E; ++e) dst[e] += src[e];
    }
  }
}

void adam_dense(float* p, const float* g, float* m, float* v, int64_t n, float lr, float b1, float b2, float eps,
                int step) {
  const double bc1 = 1.0 - pow((double)b1, (double)step), bc2 = 1.0 - pow((double)b2, (double)step);
  const float step_size = (float)((double)lr / bc1), bc2s = (float)sqrt(bc2);
  const float omb1 = (float)(1.0 - (double)b1), omb2 = (float)(1.0 - (double)b2);
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; ++i) {
    const float gi = g[i];
    const float mi = m[i] * b1 + omb1 * gi;
    const float vi = v[i] * b2 + omb2 * gi * gi;
    m[i] = mi;
    v[i] = vi;
    p[i] -= step_size * (mi / (sqrtf(vi) / bc2s + eps));
  }
}

void adagrad_dense(float* p, const float* g, float* s, int64_t n, float lr, float eps) {
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; ++i) {
    const float gi = g[i];
    const float si = s[i] + gi * gi;
    s[i] = si;
    p[i] -= lr * gi / (sqrtf(si) + eps);
  }
}
