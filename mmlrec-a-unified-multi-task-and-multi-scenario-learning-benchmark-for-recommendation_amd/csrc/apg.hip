// APG (reference model/apg.py:9-118, the branch main.py builds: use_uv_shared=True, use_mf_p=False): the per-sample
// generated [k,k] weight  W_b = reshape(Linear_kk(scene_b))  and bias  c_b = Linear_bias(scene_b)  applied to the
// low-rank activation  o1_b  (apg.py:77-80, :100-104):
//     o2_b[j] = sum_i o1_b[i] W_b[i,j] + c_b[j]
//             = sum_{i,e} o1_b[i] s_b[e] Wkk[(i k + j), e]  +  sum_i o1_b[i] bkk[i k + j]  +  sum_e s_b[e] Wb[j, e]  +  bb[j]
// is ONE ordinary GEMM  o2 = z W_cat + bb  on the feature row  z_b = [o1_b (x) s_b | o1_b | s_b | 0-pad]  against a
// re-laid-out copy W_cat [Kf, k] of the three parameter tensors -- no per-sample weight is ever materialised
// (the reference builds B x k x k of them).  This file holds the two elementwise pieces around that GEMM:
//   features fwd / bwd  (the scene embedding is detached in the reference, apg.py:152-153: no gradient to s)
//   weight pack / gradient unpack (pure re-layouts).
#include "common.hpp"

namespace mml {

__global__ __launch_bounds__(256) void apg_features_fwd_kernel(const float* __restrict__ o1, int64_t ldo1,
                                                               const float* __restrict__ s, int64_t lds,
                                                               float* __restrict__ z, int64_t ldz, int64_t B, int k,
                                                               int E, int Kf) {
  const int64_t total = B * Kf;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int kE = k * E;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) {
    const int64_t b = t / Kf;
    const int c = (int)(t - b * Kf);
    float v;
    if (c < kE) {
      const int i = c / E, e = c - i * E;
      v = o1[b * ldo1 + i] * s[b * lds + e];
    } else if (c < kE + k) {
      v = o1[b * ldo1 + (c - kE)];
    } else if (c < kE + k + E) {
      v = s[b * lds + (c - kE - k)];
    } else {
      v = 0.f;
    }
    z[b * ldz + c] = v;
  }
}

// do1[b, i] (+)= sum_e dz[b, iE + e] s[b, e] + dz[b, kE + i]
__global__ __launch_bounds__(256) void apg_features_bwd_kernel(const float* __restrict__ dz, int64_t lddz,
                                                               const float* __restrict__ s, int64_t lds,
                                                               float* __restrict__ do1, int64_t lddo1, int64_t B, int k,
                                                               int E, int accumulate) {
  const int64_t total = B * k;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) {
    const int64_t b = t / k;
    const int i = (int)(t - b * k);
    float acc = dz[b * lddz + k * E + i];
    for (int e = 0; e < E; ++e) acc += dz[b * lddz + i * E + e] * s[b * lds + e];
    float* d = do1 + b * lddo1 + i;
    *d = accumulate ? *d + acc : acc;
  }
}

// dir 0: W_cat <- (Wkk, bkk, Wb);  dir 1: (dWkk, dbkk, dWb) (+)= dW_cat
__global__ __launch_bounds__(256) void apg_weights_kernel(float* Wkk, float* bkk, float* Wb, float* Wcat, int64_t ldw,
                                                          int k, int E, int dir, int acc_kk, int acc_bkk, int acc_wb) {
  const int rows = k * E + k + E;
  const int total = rows * k;
  for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < total; t += gridDim.x * blockDim.x) {
    const int r = t / k, j = t - r * k;
    float* p;
    int acc;
    if (r < k * E) {
      const int i = r / E, e = r - i * E;
      p = Wkk + (int64_t)(i * k + j) * E + e;
      acc = acc_kk;
    } else if (r < k * E + k) {
      p = bkk + (r - k * E) * k + j;
      acc = acc_bkk;
    } else {
      p = Wb + (int64_t)j * E + (r - k * E - k);
      acc = acc_wb;
    }
    float* c = Wcat + (int64_t)r * ldw + j;
    if (dir == 0) *c = *p;
    else *p = acc ? *p + *c : *c;
  }
}

}  // namespace mml

using namespace mml;

static unsigned apg_grid(int64_t n) {
  int64_t b = cdiv(n, 256);
  if (b > 256 * 8) b = 256 * 8;
  return (unsigned)(b < 1 ? 1 : b);
}

extern "C" int mml_apg_features_fwd(const float* o1, int64_t ldo1, const float* s, int64_t lds, float* z, int64_t ldz,
                                    int64_t B, int32_t k, int32_t E, int32_t Kf, mml_stream_t stream) {
  MML_REQUIRE(B >= 0 && k > 0 && E > 0 && Kf >= k * E + k + E && ldz >= Kf, "mml_apg_features_fwd: bad sizes");
  if (B == 0) return MML_OK;
  MML_REQUIRE(o1 && s && z, "mml_apg_features_fwd: null argument");
  MML_LAUNCH(apg_features_fwd_kernel, dim3(apg_grid(B * Kf)), dim3(256), 0, to_stream(stream), o1, ldo1, s, lds, z, ldz,
             B, k, E, Kf);
  return check_launch("mml_apg_features_fwd");
}

extern "C" int mml_apg_features_bwd(const float* dz, int64_t lddz, const float* s, int64_t lds, float* do1,
                                    int64_t lddo1, int64_t B, int32_t k, int32_t E, int32_t accumulate,
                                    mml_stream_t stream) {
  MML_REQUIRE(B >= 0 && k > 0 && E > 0 && lddz >= k * E + k, "mml_apg_features_bwd: bad sizes");
  if (B == 0) return MML_OK;
  MML_REQUIRE(dz && s && do1, "mml_apg_features_bwd: null argument");
  MML_LAUNCH(apg_features_bwd_kernel, dim3(apg_grid(B * k)), dim3(256), 0, to_stream(stream), dz, lddz, s, lds, do1,
             lddo1, B, k, E, accumulate);
  return check_launch("mml_apg_features_bwd");
}

extern "C" int mml_apg_weights(float* Wkk, float* bkk, float* Wb, float* Wcat, int64_t ldw, int32_t k, int32_t E,
                               int32_t dir, int32_t acc_kk, int32_t acc_bkk, int32_t acc_wb, mml_stream_t stream) {
  MML_REQUIRE(Wkk && bkk && Wb && Wcat && k > 0 && E > 0 && ldw >= k && (dir == 0 || dir == 1),
              "mml_apg_weights: bad arguments");
  MML_LAUNCH(apg_weights_kernel, dim3(apg_grid((int64_t)(k * E + k + E) * k)), dim3(256), 0, to_stream(stream), Wkk, bkk,
             Wb, Wcat, ldw, k, E, dir, acc_kk, acc_bkk, acc_wb);
  return check_launch("mml_apg_weights");
}
