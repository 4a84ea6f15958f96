/* CPU restatement of the hot-path subset of include/mmlrec.h under the SAME symbols and signatures -- TEST
 * INFRASTRUCTURE, NOT PRODUCT CODE (same status as mmlrec_oracle.py: only tests may load the library built from this
 * file; the shipped package never does and fails loudly without the HIP library).
 *
 * SURVEY 8(b), last sentence: "The same symbols are provided by the CPU restatement library so tests run in both
 * containers."  Plain C loops over HOST memory (the mml_stream_t argument is ignored), float64 accumulation inside
 * the dot products, no tiling, no atomics; every function restates the reference lines its HIP counterpart names:
 *   mml_gather_fwd            BaseModel.input_from_feature_columns + combined_dnn_input (model/basemodel.py:461-487, utils.py:434-446)
 *   mml_scatter_bwd           aten::embedding_dense_backward, sparse=False (model/basemodel.py:122), batch order
 *   mml_gemm_grouped_*        DNN.forward Linear + activation and its autograd pair (model/utils.py:146-161)
 *   mml_gate_mix_fwd / _bwd   gate softmax + expert mix (model/mmoe.py:80-88, model/ple.py:127-152)
 *   mml_head_fwd / mml_head_bce_fwd_bwd   final layer + PredictionLayer + summed BCE (model/utils.py:242-248, basemodel.py:294-296)
 *   mml_opt_step_dense        torch.optim.{SGD,Adam,Adagrad,RMSprop}.step over dense tensors (model/basemodel.py:313, :569-584)
 *   mml_amax_batch / _reset   operand magnitudes (a contract of ours, see include/mmlrec.h)
 *   mml_dropout               nn.Dropout after a DNN layer (model/utils.py:121, :159) under this build's Philox mask stream
 *   mml_gemm_planes_cut       the pre-cut weight planes of the two-plane GEMM arithmetic (a contract of ours; the CPU GEMMs
 *                             here ignore w_planes / w_kexp and read the float weights)
 *   mml_cast16_batch / mml_gather16_fwd / mml_g16_tn / mml_g16_wgrad   the bf16-STORAGE layer family (K3' of the header: the
 *                             same DNN layers, model/utils.py:146-161, on operands stored as bf16 -- round to nearest even --
 *                             with float64 accumulation here)
 * tests/test_cabi_cpu.py drives one full MMoE training step of a reference-made golden fixture through these entry
 * points (the call sequence of mmlrec_amd/engine.py) in the CPU container.
 * Build: oracle/build_fast.py (gcc -O2 -shared -fPIC), output oracle/_build/libmmlrec_cpu.so. */
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../include/mmlrec.h"

static __thread char g_err[256] = "";
static int fail(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof g_err, fmt, ap);
  va_end(ap);
  return MML_ERR_ARG;
}
#define REQUIRE(c, ...) do { if (!(c)) return fail(__VA_ARGS__); } while (0)

const char* mml_last_error(void) { return g_err; }
int mml_version(void) { return 100; }

static float act_fwd(float v, int act) {
  if (act == MML_ACT_RELU) return v > 0.f ? v : 0.f;
  if (act == MML_ACT_SIGMOID) return 1.f / (1.f + expf(-v));
  if (act == MML_ACT_SIGMOID2) return 2.f / (1.f + expf(-v));
  return v;
}
static float act_bwd_from_output(float y, int act) {
  if (act == MML_ACT_RELU) return y > 0.f ? 1.f : 0.f;
  if (act == MML_ACT_SIGMOID) return y * (1.f - y);
  if (act == MML_ACT_SIGMOID2) return y * (1.f - 0.5f * y);
  return 1.f;
}
static void amax_raise(uint32_t* slot, float v) {
  union { float f; uint32_t u; } c;
  c.f = fabsf(v);
  if (slot && c.u > slot[0] && c.u <= 0x7f800000u) slot[0] = c.u;
}

/* ------------------------------------------------------------------------------------------------ K1 / K2 */
int mml_gather_fwd(const float* const* tables, const int64_t* vocab, const int32_t* col, int32_t F, int32_t E,
                   const float* X, int64_t ldX, int32_t dense_col0, int32_t Nd, int64_t B, float* out, int64_t ldo,
                   int32_t* status, mml_stream_t stream) {
  (void)stream;
  REQUIRE(F >= 0 && F <= MML_MAX_FIELDS && E > 0 && B >= 0 && Nd >= 0, "mml_gather_fwd: bad sizes");
  REQUIRE(B == 0 || (X && out && tables && vocab), "mml_gather_fwd: null argument");
  REQUIRE(ldo >= (int64_t)F * E + Nd, "mml_gather_fwd: ldo too small");
  for (int64_t b = 0; b < B; ++b) {
    for (int f = 0; f < F; ++f) {
      int64_t i = (int64_t)X[b * ldX + (col ? col[f] : f)]; /* .long(): truncation (model/basemodel.py:476) */
      if (i < 0) { if (status) *status |= 1; i = 0; }
      if (i >= vocab[f]) { if (status) *status |= 2; i = vocab[f] - 1; }
      memcpy(out + b * ldo + (int64_t)f * E, tables[f] + i * E, sizeof(float) * E);
    }
    for (int j = 0; j < Nd; ++j) out[b * ldo + (int64_t)F * E + j] = X[b * ldX + dense_col0 + j];
  }
  return MML_OK;
}

int mml_scatter_bwd(float* const* grad_tables, const int64_t* vocab, const int32_t* col, int32_t F, int32_t E,
                    const float* X, int64_t ldX, int64_t B, const float* dOut, int64_t ldo, uint32_t* const* seen,
                    const int64_t* rowbase, int32_t* touched, int32_t* touched_count, int32_t touched_cap,
                    uint8_t* row_marks, int32_t* status, mml_stream_t stream) {
  (void)stream;
  REQUIRE(F >= 0 && F <= MML_MAX_FIELDS && E > 0 && B >= 0, "mml_scatter_bwd: bad sizes");
  REQUIRE(B == 0 || (X && dOut && grad_tables && vocab), "mml_scatter_bwd: null argument");
  REQUIRE(!touched || (seen && rowbase && touched_count && touched_cap > 0), "mml_scatter_bwd: touched list malformed");
  if (touched) *touched_count = 0;
  int64_t markbase = 0;
  for (int f = 0; f < F; ++f) {
    for (int64_t b = 0; b < B; ++b) {
      const int64_t i = (int64_t)X[b * ldX + (col ? col[f] : f)];
      if (i < 0) { if (status) *status |= 1; continue; }
      if (i >= vocab[f]) { if (status) *status |= 2; continue; }
      for (int e = 0; e < E; ++e) grad_tables[f][i * E + e] += dOut[b * ldo + (int64_t)f * E + e];
      if (row_marks && !touched) row_marks[markbase + i] = 1;
      if (touched && !((seen[f][i >> 5] >> (i & 31)) & 1u)) {
        seen[f][i >> 5] |= 1u << (i & 31);
        if (*touched_count < touched_cap) touched[*touched_count] = (int32_t)(rowbase[f] + i);
        ++*touched_count;
      }
    }
    markbase += (vocab[f] + 31) / 32 * 32;
  }
  return MML_OK;
}

/* ------------------------------------------------------------------------------------------------ operand magnitudes */
int mml_amax_reset(uint32_t* slots, int64_t n_slots, mml_stream_t stream) {
  (void)stream;
  REQUIRE(n_slots >= 0 && (n_slots == 0 || slots), "mml_amax_reset: bad arguments");
  memset(slots, 0, (size_t)n_slots * MML_AMAX_WORDS * sizeof(uint32_t));
  return MML_OK;
}
int mml_amax_batch(const mml_amax_desc* d, int32_t n, mml_stream_t stream) {
  (void)stream;
  REQUIRE(n >= 0 && (n == 0 || d), "mml_amax_batch: bad descriptor array");
  for (int i = 0; i < n; ++i) {
    REQUIRE(d[i].slot && d[i].ld >= d[i].cols, "mml_amax_batch: tensor %d malformed", i);
    for (int64_t r = 0; r < d[i].rows; ++r)
      for (int c = 0; c < d[i].cols; ++c) amax_raise(d[i].slot, d[i].x[r * d[i].ld + c]);
  }
  return MML_OK;
}

/* ------------------------------------------------------------------------------------------------ K3 */
/* Philox4x32-10 (Salmon et al., SC'11), one block; the mask contract of mml_dropout in include/mmlrec.h */
static void philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1) {
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
    c[0] = n0; c[1] = (uint32_t)p1; c[2] = n2; c[3] = (uint32_t)p0;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
}

int mml_dropout(const float* x, int64_t ldx, float* out, int64_t ldo, int64_t rows, int32_t cols, int64_t row0, float p,
                uint64_t seed, uint32_t site, const int32_t* step_dev, int32_t step, int32_t accumulate,
                mml_stream_t stream) {
  (void)stream;
  REQUIRE(rows >= 0 && cols >= 0, "mml_dropout: negative size");
  REQUIRE(p >= 0.f && p < 1.f, "mml_dropout: p must be in [0, 1)");
  if (rows == 0 || cols == 0) return MML_OK;
  REQUIRE(x && out && ldx >= cols && ldo >= cols, "mml_dropout: bad arguments");
  const double t = (double)p * 4294967296.0;
  const uint32_t thr = t >= 4294967295.0 ? 0xffffffffu : (uint32_t)t;
  const float scale = 1.0f / (1.0f - p);
  const uint32_t st = step_dev ? (uint32_t)*step_dev : (uint32_t)step;
  for (int64_t r = 0; r < rows; ++r)
    for (int32_t q = 0; 4 * q < cols; ++q) {
      uint32_t w[4] = {(uint32_t)(row0 + r), (uint32_t)q, st, site};
      philox4x32_10(w, (uint32_t)seed, (uint32_t)(seed >> 32));
      for (int j = 0; j < 4 && 4 * q + j < cols; ++j) {
        float o = x[r * ldx + 4 * q + j] * (w[j] < thr ? 0.f : scale);
        if (accumulate) o += out[r * ldo + 4 * q + j];
        out[r * ldo + 4 * q + j] = o;
      }
    }
  return MML_OK;
}

/* mml_gemm_planes_cut: the two fp16 planes of a weight matrix in the operand-image layout of include/mmlrec.h.  The
 * conversions are spelled out (round to nearest even, like the device's v_cvt_pk_f16_f32). */
static int amax_exp(const uint32_t* slot) {
  uint32_t m = 0;
  for (int i = 0; i < MML_AMAX_WORDS; ++i) m = slot[i] > m ? slot[i] : m;
  const int e = (int)((m >> 23) & 0xffu);
  if (e == 255) return 0;
  const int k = 141 - e;
  return k > 110 ? 110 : (k < -110 ? -110 : k);
}
/* fp32 -> fp16 bit pattern, round to nearest even (the gcc of this image has no _Float16 on x86-64), and back */
static uint32_t half_bits(float xf) {
  uint32_t f;
  memcpy(&f, &xf, 4);
  const uint32_t sign = (f >> 16) & 0x8000u, x = f & 0x7fffffffu;
  if (x >= 0x7f800000u) return sign | 0x7c00u | (x > 0x7f800000u ? 0x200u : 0u);  /* Inf / NaN */
  if (x >= 0x477ff000u) return sign | 0x7c00u;                                     /* >= 65520 rounds to Inf */
  if (x < 0x38800000u) {                                                           /* below 2^-14: subnormal or zero */
    float a;
    memcpy(&a, &x, 4);
    return sign | (uint32_t)lrintf(a * 16777216.0f);  /* |x| 2^24, ties to even (1024 = the smallest normal) */
  }
  uint32_t h = ((((x >> 23) - 127u + 15u)) << 10) | ((x & 0x7fffffu) >> 13);
  const uint32_t rem = x & 0x1fffu;
  if (rem > 0x1000u || (rem == 0x1000u && (h & 1u))) ++h;
  return sign | h;
}
static float half_value(uint32_t h) {
  const uint32_t e = (h >> 10) & 31u, m = h & 0x3ffu;
  float v;
  if (e == 0) v = ldexpf((float)m, -24);
  else if (e == 31) v = m ? NAN : INFINITY;
  else {
    const uint32_t b = ((e - 15u + 127u) << 23) | (m << 13);
    memcpy(&v, &b, 4);
  }
  return (h & 0x8000u) ? -v : v;
}
int mml_gemm_planes_cut(const mml_planes_desc* d, int32_t n, mml_stream_t stream) {
  (void)stream;
  REQUIRE(n >= 0 && (n == 0 || d), "mml_gemm_planes_cut: bad descriptor array");
  for (int i = 0; i < n; ++i) {
    const mml_planes_desc* q = &d[i];
    const int64_t ldp = q->ldp ? q->ldp : q->ld;
    REQUIRE(q->W && q->planes && q->kexp && q->n_amax >= 1 && q->n_amax <= MML_MAX_SRC, "mml_gemm_planes_cut: matrix %d malformed", i);
    REQUIRE(q->layout == MML_PLANES_ROWS ? ldp >= (q->cols + 15) / 16 * 16 : (q->rows % 16 == 0 && ldp >= q->cols),
            "mml_gemm_planes_cut: matrix %d: pitch / reduction extent", i);
    REQUIRE(!q->W2 || (2 * q->n_amax <= MML_MAX_SRC && q->ld2 >= q->cols), "mml_gemm_planes_cut: matrix %d: product factors", i);
    int k = 110;
    for (int a = 0; a < q->n_amax; ++a) {
      int ka;
      if (q->W2) { /* K6: the bound of a product is the product of the factors' bounds (fp32 product, as on the device) */
        uint32_t ma = 0, mb = 0, pb[MML_AMAX_WORDS] = {0};
        for (int w = 0; w < MML_AMAX_WORDS; ++w) {
          ma = q->amax[a][w] > ma ? q->amax[a][w] : ma;
          mb = q->amax[q->n_amax + a][w] > mb ? q->amax[q->n_amax + a][w] : mb;
        }
        float fa, fb;
        memcpy(&fa, &ma, 4);
        memcpy(&fb, &mb, 4);
        const float pr = fa * fb;
        memcpy(&pb[0], &pr, 4);
        ka = amax_exp(pb);
      } else {
        ka = amax_exp(q->amax[a]);
      }
      k = ka < k ? ka : k;
    }
    *q->kexp = k;
    const float s = ldexpf(1.f, k);
    const int rows_k = q->layout == MML_PLANES_ROWS;  /* reduction along a row */
    const int64_t outer = rows_k ? q->rows : q->cols, blocks = rows_k ? (q->cols + 15) / 16 : q->rows / 16;
    for (int64_t o = 0; o < outer; ++o)
      for (int64_t b = 0; b < blocks; ++b) {
        uint32_t hp[16], lp[16];
        for (int e = 0; e < 16; ++e) {
          const int64_t kk = 16 * b + e;
          const int live = rows_k ? kk < q->cols : 1;
          float x = live ? (rows_k ? q->W[o * q->ld + kk] : q->W[kk * q->ld + o]) : 0.f;
          if (q->W2 && live) x = x * (rows_k ? q->W2[o * q->ld2 + kk] : q->W2[kk * q->ld2 + o]);
          const float y = x * s;
          hp[e] = half_bits(y);
          lp[e] = half_bits(y - half_value(hp[e]));
        }
        for (int h = 0; h < 2; ++h)
          for (int j = 0; j < 4; ++j) {
            const int k0 = 4 * h + ((2 * j) & 3) + 8 * ((2 * j) >> 2), k1 = 4 * h + ((2 * j + 1) & 3) + 8 * ((2 * j + 1) >> 2);
            const int64_t wh = 16 * b + 4 * h + j, wl = 16 * b + 8 + 4 * h + j;
            if (rows_k) {
              q->planes[o * ldp + wh] = hp[k0] | (hp[k1] << 16);
              q->planes[o * ldp + wl] = lp[k0] | (lp[k1] << 16);
            } else {
              q->planes[wh * ldp + o] = hp[k0] | (hp[k1] << 16);
              q->planes[wl * ldp + o] = lp[k0] | (lp[k1] << 16);
            }
          }
      }
  }
  return MML_OK;
}

int mml_gemm_set_mode(int32_t mode) { (void)mode; return MML_OK; }
int mml_gemm_get_mode(void) { return 0; }
const char* mml_gemm_last_kernel(void) { return "cpu"; }
const char* mml_gather_last_kernel(void) { return "cpu"; }

int mml_gemm_grouped_fwd(const mml_gemm_fwd_desc* d, int32_t n, mml_stream_t stream) {
  (void)stream;
  REQUIRE(n >= 0 && (n == 0 || d), "mml_gemm_grouped_fwd: bad descriptor array");
  for (int p = 0; p < n; ++p) {
    const mml_gemm_fwd_desc* q = d + p;
    REQUIRE(q->A && q->W && q->C && q->M >= 0 && q->N > 0 && q->K > 0, "mml_gemm_grouped_fwd: problem %d malformed", p);
    for (int64_t m = 0; m < q->M; ++m)
      for (int nn = 0; nn < q->N; ++nn) {
        double s = q->bias ? q->bias[nn] : 0.0;
        for (int k = 0; k < q->K; ++k)
          s += (double)q->A[m * q->lda + k] * (q->w_kn ? q->W[(int64_t)k * q->ldw + nn] : q->W[(int64_t)nn * q->ldw + k]);
        const float v = act_fwd((float)s, q->act);
        q->C[m * q->ldc + nn] = v;
        amax_raise(q->amax_out, v);
        if (q->mul && q->prod) { /* K7: the gated layer input leaves with the gate */
          const float pv = v * q->mul[m * q->ldmul + nn];
          q->prod[m * q->ldprod + nn] = pv;
          amax_raise(q->amax_prod, pv);
        }
        if (q->relu_mask && q->act == MML_ACT_RELU) {
          uint32_t* w = q->relu_mask + m * q->ldmask + (nn >> 5);
          if (v > 0.f) *w |= 1u << (nn & 31); else *w &= ~(1u << (nn & 31));
        }
      }
  }
  return MML_OK;
}

int mml_gemm_grouped_dgrad(const mml_gemm_dgrad_desc* d, int32_t n, mml_stream_t stream) {
  (void)stream;
  REQUIRE(n >= 0 && (n == 0 || d), "mml_gemm_grouped_dgrad: bad descriptor array");
  for (int p = 0; p < n; ++p) {
    const mml_gemm_dgrad_desc* q = d + p;
    REQUIRE((q->dA || q->gate_h) && q->n_src >= 1 && q->n_src <= MML_MAX_SRC, "mml_gemm_grouped_dgrad: problem %d malformed", p);
    REQUIRE(q->gate_h || q->act == MML_ACT_NONE || q->Y || q->relu_mask,
            "mml_gemm_grouped_dgrad: act set but Y null in problem %d", p);
    REQUIRE(!q->gate_h || (q->gate_g && q->d_h && q->d_g), "mml_gemm_grouped_dgrad: gate-mode problem %d lacks a factor", p);
    for (int64_t m = 0; m < q->M; ++m)
      for (int k = 0; k < q->K; ++k) {
        double s = 0.0;
        for (int sidx = 0; sidx < q->n_src; ++sidx)
          for (int nn = 0; nn < q->N[sidx]; ++nn)
            s += (double)q->dC[sidx][m * q->lddc[sidx] + nn] *
                 (q->w_kn[sidx] ? q->W[sidx][(int64_t)k * q->ldw[sidx] + nn] : q->W[sidx][(int64_t)nn * q->ldw[sidx] + k]);
        float v = (float)s;
        if (q->gate_h) { /* K7 backward: the gradients of the two factors of x = h (.) g */
          const float hv = q->gate_h[m * q->ld_h + k], gv = q->gate_g[m * q->ld_g + k];
          float* ph = q->d_h + m * q->ld_dh + k;
          float* pg = q->d_g + m * q->ld_dg + k;
          const float vh = v * gv * (q->act_h != MML_ACT_NONE ? act_bwd_from_output(hv, q->act_h) : 1.f) + (q->acc_h ? *ph : 0.f);
          const float vg = v * hv * (q->act_g != MML_ACT_NONE ? act_bwd_from_output(gv, q->act_g) : 1.f) + (q->acc_g ? *pg : 0.f);
          *ph = vh;
          *pg = vg;
          amax_raise(q->amax_dh, vh);
          amax_raise(q->amax_dg, vg);
          continue;
        }
        if (q->relu_mask) {
          if (!((q->relu_mask[m * q->ldmask + (k >> 5)] >> (k & 31)) & 1u)) v = 0.f;
        } else if (q->act != MML_ACT_NONE) {
          v *= act_bwd_from_output(q->Y[m * q->ldy + k], q->act);
        }
        if (q->accumulate) v += q->dA[m * q->ldda + k];
        q->dA[m * q->ldda + k] = v;
        amax_raise(q->amax_out, v);
      }
  }
  return MML_OK;
}

/* SURVEY 8(b) names of K7 / K6 (include/mmlrec.h): the grouped GEMMs restricted to the descriptors they are named for */
int mml_pep_gate_fwd(const mml_gemm_fwd_desc* d, int32_t n, mml_stream_t stream) {
  REQUIRE(n >= 0 && (n == 0 || d), "mml_pep_gate_fwd: bad descriptor array");
  for (int i = 0; i < n; ++i) REQUIRE(d[i].mul && d[i].prod, "mml_pep_gate_fwd: problem %d carries no mul / prod", i);
  return mml_gemm_grouped_fwd(d, n, stream);
}
int mml_pep_gate_bwd(const mml_gemm_dgrad_desc* d, int32_t n, mml_stream_t stream) {
  REQUIRE(n >= 0 && (n == 0 || d), "mml_pep_gate_bwd: bad descriptor array");
  for (int i = 0; i < n; ++i) REQUIRE(d[i].gate_h, "mml_pep_gate_bwd: problem %d is not in gate mode", i);
  return mml_gemm_grouped_dgrad(d, n, stream);
}
int mml_star_linear_fwd(const mml_gemm_fwd_desc* d, int32_t n, mml_stream_t stream) {
  REQUIRE(n >= 0 && (n == 0 || d), "mml_star_linear_fwd: bad descriptor array");
  for (int i = 0; i < n; ++i)
    REQUIRE(d[i].w_kn == 1 && d[i].w_planes && d[i].w_kexp,
            "mml_star_linear_fwd: problem %d is not a [K, N]-layout layer with pre-cut planes", i);
  return mml_gemm_grouped_fwd(d, n, stream);
}
int mml_star_linear_bwd(const mml_gemm_dgrad_desc* d, int32_t n, mml_stream_t stream) {
  REQUIRE(n >= 0 && (n == 0 || d), "mml_star_linear_bwd: bad descriptor array");
  for (int i = 0; i < n; ++i)
    for (int s2 = 0; s2 < d[i].n_src && s2 < MML_MAX_SRC; ++s2)
      REQUIRE(d[i].w_kn[s2] == 1 && d[i].w_planes[s2] && d[i].w_kexp[s2],
              "mml_star_linear_bwd: source %d of problem %d is not a [K, N]-layout layer with pre-cut planes", s2, i);
  return mml_gemm_grouped_dgrad(d, n, stream);
}

int64_t mml_gemm_grouped_wgrad_workspace_bytes(const mml_gemm_wgrad_desc* d, int32_t n) { (void)d; (void)n; return 256; }
int mml_gemm_grouped_wgrad_phase(const mml_gemm_wgrad_desc* d, int32_t n, void* workspace, int64_t workspace_bytes,
                                 int32_t phase, mml_stream_t stream) {
  (void)workspace; (void)workspace_bytes; (void)stream;
  REQUIRE(n >= 0 && (n == 0 || d), "mml_gemm_grouped_wgrad: bad descriptor array");
  REQUIRE(phase >= 0 && phase <= 2, "mml_gemm_grouped_wgrad_phase: bad phase");
  if (phase == 2) return MML_OK; /* (the partial-product phase already wrote the results: there are no slabs here) */
  for (int p = 0; p < n; ++p) {
    const mml_gemm_wgrad_desc* q = d + p;
    REQUIRE(q->dC && q->A && q->dW, "mml_gemm_grouped_wgrad: null pointer in problem %d", p);
    for (int nn = 0; nn < q->N; ++nn) {
      for (int k = 0; k < q->K; ++k) {
        double s = 0.0;
        for (int64_t m = 0; m < q->M; ++m) s += (double)q->dC[m * q->lddc + nn] * q->A[m * q->lda + k];
        float* dst = q->w_kn ? q->dW + (int64_t)k * q->lddw + nn : q->dW + (int64_t)nn * q->lddw + k;
        *dst = (q->accumulate ? *dst : 0.f) + (float)s;
      }
      if (q->dbias) {
        double s = 0.0;
        for (int64_t m = 0; m < q->M; ++m) s += q->dC[m * q->lddc + nn];
        q->dbias[nn] = (q->accumulate ? q->dbias[nn] : 0.f) + (float)s;
      }
    }
  }
  return MML_OK;
}
int mml_gemm_grouped_wgrad(const mml_gemm_wgrad_desc* d, int32_t n, void* workspace, int64_t workspace_bytes,
                           mml_stream_t stream) {
  return mml_gemm_grouped_wgrad_phase(d, n, workspace, workspace_bytes, 0, stream);
}

/* ------------------------------------------------------------------------------------------------ K4 */
int mml_gate_mix_fwd(const mml_gate_group* g, mml_stream_t stream) {
  (void)stream;
  REQUIRE(g && g->n_experts >= 1 && g->n_gates >= 1 && g->H > 0, "mml_gate_mix_fwd: bad group");
  if (g->out_bf16) return MML_ERR_UNSUPPORTED;  /* (bf16 outputs exist in the HIP library's fast row kernels only) */
  for (int64_t b = 0; b < g->B; ++b)
    for (int gi = 0; gi < g->n_gates; ++gi) {
      const mml_gate_desc* d = &g->gate[gi];
      double logit[MML_MAX_EXPERTS], mx = -1e300, den = 0.0;
      for (int e = 0; e < d->ne; ++e) {
        double s = 0.0;
        for (int k = 0; k < d->Gd; ++k) s += (double)d->G[b * d->ldg + k] * d->Wg[(int64_t)e * d->Gd + k];
        logit[e] = (float)s;
        if (logit[e] > mx) mx = logit[e];
      }
      for (int e = 0; e < d->ne; ++e) { logit[e] = expf((float)(logit[e] - mx)); den += logit[e]; }
      for (int e = 0; e < d->ne; ++e) d->P[b * d->ldp + e] = (float)(logit[e] / den);
      for (int h = 0; h < g->H; ++h) {
        double s = 0.0;
        for (int e = 0; e < d->ne; ++e) s += (double)d->P[b * d->ldp + e] * g->E[d->expert[e]][b * g->lde[d->expert[e]] + h];
        d->mix[b * d->ldmix + h] = (float)s;
        amax_raise(g->amax_mix, (float)s);
      }
    }
  return MML_OK;
}

int64_t mml_gate_mix_bwd_workspace_bytes(const mml_gate_group* g) { (void)g; return 256; }
int mml_gate_mix_bwd(const mml_gate_group* g, void* workspace, int64_t workspace_bytes, mml_stream_t stream) {
  (void)workspace; (void)workspace_bytes; (void)stream;
  REQUIRE(g && g->n_experts >= 1 && g->n_gates >= 1 && g->H > 0, "mml_gate_mix_bwd: bad group");
  if (g->out_bf16) return MML_ERR_UNSUPPORTED;
  for (int gi = 0; gi < g->n_gates; ++gi)
    if (g->gate[gi].active)
      memset(g->gate[gi].dWg, 0, sizeof(float) * (size_t)g->gate[gi].ne * g->gate[gi].Gd);
  for (int64_t b = 0; b < g->B; ++b) {
    for (int x = 0; x < g->n_experts; ++x)
      for (int h = 0; h < g->H; ++h) g->dE[x][b * g->ldde[x] + h] = 0.f;
    for (int gi = 0; gi < g->n_gates; ++gi) {
      const mml_gate_desc* d = &g->gate[gi];
      if (!d->active) continue;
      double dp[MML_MAX_EXPERTS], dot = 0.0;
      for (int e = 0; e < d->ne; ++e) {
        const int x = d->expert[e];
        double s = 0.0;
        for (int h = 0; h < g->H; ++h) s += (double)d->dmix[b * d->lddmix + h] * g->E[x][b * g->lde[x] + h];
        dp[e] = s;
        dot += d->P[b * d->ldp + e] * s;
      }
      for (int e = 0; e < d->ne; ++e) dp[e] = d->P[b * d->ldp + e] * (dp[e] - dot); /* dlogit */
      for (int k = 0; k < d->Gd; ++k) {
        const float gk = d->G[b * d->ldg + k];
        double dg = 0.0;
        for (int e = 0; e < d->ne; ++e) {
          dg += dp[e] * d->Wg[(int64_t)e * d->Gd + k];
          d->dWg[(int64_t)e * d->Gd + k] += (float)(dp[e] * gk);
        }
        if (d->g_relu && !(gk > 0.f)) dg = 0.0;
        d->dG[b * d->lddg + k] = (float)dg;
        amax_raise(g->amax_dG, (float)dg);
      }
      for (int e = 0; e < d->ne; ++e) {
        const int x = d->expert[e];
        for (int h = 0; h < g->H; ++h) g->dE[x][b * g->ldde[x] + h] += d->P[b * d->ldp + e] * d->dmix[b * d->lddmix + h];
      }
    }
    for (int x = 0; x < g->n_experts; ++x)
      for (int h = 0; h < g->H; ++h) {
        float* v = &g->dE[x][b * g->ldde[x] + h];
        if (g->e_relu && !(g->E[x][b * g->lde[x] + h] > 0.f)) *v = 0.f;
        amax_raise(g->amax_dE, *v);
      }
  }
  return MML_OK;
}

/* ------------------------------------------------------------------------------------------------ K5 */
static int heads(const mml_head_group* g, int train) {
  REQUIRE(g && g->n_heads >= 1 && g->n_heads <= MML_MAX_HEADS && g->prob, "mml_head_*: bad group");
  if (g->dh_bf16) return MML_ERR_UNSUPPORTED;   /* (bf16 outputs exist in the HIP library's fast row kernels only) */
  double loss = 0.0;
  if (train)
    for (int t = 0; t < g->n_heads; ++t) {
      const mml_head_desc* d = &g->head[t];
      if (d->dw) memset(d->dw, 0, sizeof(float) * (size_t)d->H);
      if (d->dbias) d->dbias[0] = 0.f;
    }
  for (int64_t b = 0; b < g->B; ++b)
    for (int t = 0; t < g->n_heads; ++t) {
      const mml_head_desc* d = &g->head[t];
      double s = d->bias[0];
      for (int i = 0; i < d->n_bias2; ++i) s += d->bias2[i];
      for (int h = 0; h < d->H; ++h)
        s += (double)d->Hin[b * d->ldh + h] * (d->gate ? d->gate[b * d->ldgate + h] : 1.f) * d->w[h] * (d->w2 ? d->w2[h] : 1.f);
      const float p = 1.f / (1.f + expf(-(float)s));
      const float m = (d->mask_col >= 0 && g->mask) ? g->mask[b * g->ldmask + d->mask_col] : 1.f;
      const float pm = p * m;
      g->prob[b * g->ldprob + t] = pm;
      if (!train) continue;
      float dpm;
      if (g->y) {
        const float y = g->y[b * g->ldy + t];
        /* F.binary_cross_entropy: torch.clamp(log p, min = -100) -- a NaN stays a NaN (fmaxf alone would drop it) */
        const float lg = logf(pm), lg1 = log1pf(-pm);
        const float lp = lg != lg ? lg : fmaxf(lg, -100.f), l1p = lg1 != lg1 ? lg1 : fmaxf(lg1, -100.f);
        loss += -(y * lp + (1.f - y) * l1p);
        dpm = (pm - y) / fmaxf((1.f - pm) * pm, 1e-12f);
      } else {
        dpm = g->dprob[b * g->lddprob + t];
      }
      const float dlogit = dpm * m * p * (1.f - p);
      if (d->dbias) d->dbias[0] += dlogit;
      for (int h = 0; h < d->H; ++h) {
        const float hv = d->Hin[b * d->ldh + h];
        const float gv = d->gate ? d->gate[b * d->ldgate + h] : 1.f;   /* gated head: the input is hv * gv (pepnet.py:72-78) */
        if (d->dw) d->dw[h] += dlogit * (hv * gv);
        float dh = dlogit * d->w[h] * (d->w2 ? d->w2[h] : 1.f);
        if (d->gate && d->dgate) {
          float dg = dh * hv;
          if (d->gate_act == MML_ACT_SIGMOID) dg *= gv * (1.f - gv);
          else if (d->gate_act == MML_ACT_SIGMOID2) dg *= 2.f * (0.5f * gv) * (1.f - 0.5f * gv);
          d->dgate[b * d->lddgate + h] = dg;
          amax_raise(g->amax_dG, dg);
        }
        dh *= gv;
        if (d->h_relu && !(hv > 0.f)) dh = 0.f;
        if (d->dH) { d->dH[b * d->lddh + h] = dh; amax_raise(g->amax_dH, dh); }
      }
    }
  if (train && g->loss) g->loss[0] = (float)loss;
  return MML_OK;
}
int64_t mml_head_workspace_bytes(const mml_head_group* g) { (void)g; return 256; }
int mml_head_fwd(const mml_head_group* g, mml_stream_t stream) { (void)stream; return heads(g, 0); }
int mml_head_bce_fwd_bwd(const mml_head_group* g, void* workspace, int64_t workspace_bytes, mml_stream_t stream) {
  (void)workspace; (void)workspace_bytes; (void)stream;
  return heads(g, 1);
}
/* the two-launch forms of include/mmlrec.h: here phase 1 (and 0) does everything, phase 2 has nothing left to reduce */
int mml_head_bce_fwd_bwd_phase(const mml_head_group* g, void* workspace, int64_t workspace_bytes, int32_t phase,
                               mml_stream_t stream) {
  REQUIRE(phase >= 0 && phase <= 2, "mml_head_bce_fwd_bwd_phase: bad phase");
  return phase == 2 ? MML_OK : mml_head_bce_fwd_bwd(g, workspace, workspace_bytes, stream);
}
int mml_gate_mix_bwd_phase(const mml_gate_group* g, void* workspace, int64_t workspace_bytes, int32_t phase,
                           mml_stream_t stream) {
  REQUIRE(phase >= 0 && phase <= 2, "mml_gate_mix_bwd_phase: bad phase");
  return phase == 2 ? MML_OK : mml_gate_mix_bwd(g, workspace, workspace_bytes, stream);
}
/* (phase 1 of the CPU restatement leaves finished results: nothing to reduce) */
int mml_rows_reduce_batch(const mml_rows_reduce_item* items, int32_t n, mml_stream_t stream) {
  (void)stream;
  REQUIRE(n >= 0 && (n == 0 || items), "mml_rows_reduce_batch: bad item array");
  for (int i = 0; i < n; ++i)
    REQUIRE((items[i].kind == MML_ROWS_REDUCE_HEAD || items[i].kind == MML_ROWS_REDUCE_GATE) && items[i].group,
            "mml_rows_reduce_batch: bad item");
  return MML_OK;
}

/* ------------------------------------------------------------------------------------------------ K8 */
int mml_opt_step_dense(const mml_opt_tensor* t, int32_t n, const mml_opt_hyper* h, mml_stream_t stream) {
  (void)stream;
  REQUIRE(h && h->kind >= MML_OPT_SGD && h->kind <= MML_OPT_RMSPROP, "mml_opt_step_dense: bad hyper");
  REQUIRE(n >= 0 && (n == 0 || t), "mml_opt_step_dense: bad tensor array");
  const int step = h->step_dev ? ((const int32_t*)h->step_dev)[0] : h->step;
  const double bc1 = 1.0 - pow(h->beta1, step), bc2 = 1.0 - pow(h->beta2, step);
  for (int i = 0; i < n; ++i) {
    REQUIRE(t[i].param && t[i].grad, "mml_opt_step_dense: tensor %d malformed", i);
    for (int64_t j = 0; j < t[i].n; ++j) {
      float p = t[i].param[j], gr = t[i].grad[j];
      if (t[i].l2 != 0.f) gr += 2.f * t[i].l2 * p;
      if (t[i].l1 != 0.f) gr += t[i].l1 * (p > 0.f ? 1.f : (p < 0.f ? -1.f : 0.f));
      if (h->kind == MML_OPT_SGD) {
        p -= h->lr * gr;
      } else if (h->kind == MML_OPT_ADAM) { /* torch.optim.Adam: betas (0.9, 0.999), eps 1e-8 */
        float* m = t[i].state1 + j; float* v = t[i].state2 + j;
        *m = h->beta1 * *m + (1.f - h->beta1) * gr;
        *v = h->beta2 * *v + (1.f - h->beta2) * gr * gr;
        const float denom = sqrtf(*v) / (float)sqrt(bc2) + h->eps;
        p -= (float)(h->lr / bc1) * (*m / denom);
      } else if (h->kind == MML_OPT_ADAGRAD) {
        float* s = t[i].state1 + j;
        *s += gr * gr;
        p -= h->lr * gr / (sqrtf(*s) + h->eps);
      } else { /* RMSprop: alpha 0.99 */
        float* s = t[i].state1 + j;
        *s = h->alpha * *s + (1.f - h->alpha) * gr * gr;
        p -= h->lr * gr / (sqrtf(*s) + h->eps);
      }
      t[i].param[j] = p;
      if (h->zero_grad || t[i].zero_grads) ((float*)t[i].grad)[j] = 0.f;
    }
    if (t[i].grad_marks && t[i].row_elems > 0) memset(t[i].grad_marks, 0, (size_t)(t[i].n / t[i].row_elems));
  }
  return MML_OK;
}

/* ------------------------------------------------------------------------------------------------ K3' (bf16 storage) */
static uint16_t to_bf16(float f) {  /* round to nearest even; NaN stays NaN */
  union { float f; uint32_t u; } c;
  c.f = f;
  if ((c.u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((c.u >> 16) | 0x40u);
  c.u += 0x7fffu + ((c.u >> 16) & 1u);
  return (uint16_t)(c.u >> 16);
}
static float from_bf16(uint16_t h) {
  union { float f; uint32_t u; } c;
  c.u = (uint32_t)h << 16;
  return c.f;
}
const char* mml_g16_last_kernel(void) { return "cpu"; }
int mml_gemm_set_nt(int32_t on) { (void)on; return MML_OK; }

int mml_cast16_batch(const mml_cast16_desc* d, int32_t n, mml_stream_t stream) {
  (void)stream;
  REQUIRE(d || n == 0, "mml_cast16_batch: descriptor array is null");
  for (int i = 0; i < n; ++i) {
    const mml_cast16_desc* q = d + i;
    REQUIRE(q->src && q->dst, "mml_cast16_batch: null matrix (item %d)", i);
    for (int64_t r = 0; r < q->rows; ++r)
      for (int c = 0; c < q->cols; ++c) {
        const uint16_t v = to_bf16(q->src[r * q->lds + c]);
        if (q->transpose) q->dst[(int64_t)c * q->ldd + r] = v;
        else q->dst[r * q->ldd + c] = v;
      }
  }
  return MML_OK;
}

int mml_gather16_fwd(const float* const* tables, const int64_t* vocab, const int32_t* col, int32_t F, int32_t E,
                     const float* X, int64_t ldX, int32_t dense_col0, int32_t Nd, int64_t B, uint16_t* out, int64_t ldo,
                     int32_t* status, mml_stream_t stream) {
  (void)stream;
  REQUIRE(tables && vocab && col && X && out, "mml_gather16_fwd: null argument");
  for (int64_t b = 0; b < B; ++b) {
    for (int f = 0; f < F; ++f) {
      int64_t i = (int64_t)X[b * ldX + col[f]];
      if (i < 0) { if (status) *status |= 1; i = 0; }
      else if (i >= vocab[f]) { if (status) *status |= 2; i = vocab[f] - 1; }
      for (int e = 0; e < E; ++e) out[b * ldo + (int64_t)f * E + e] = to_bf16(tables[f][i * E + e]);
    }
    for (int j = 0; j < Nd; ++j) out[b * ldo + (int64_t)F * E + j] = to_bf16(X[b * ldX + dense_col0 + j]);
  }
  return MML_OK;
}

int mml_g16_tn(const mml_g16_tn_desc* d, int32_t n, mml_stream_t stream) {
  (void)stream;
  REQUIRE(d, "mml_g16_tn: descriptor array is null");
  for (int i = 0; i < n; ++i) {
    const mml_g16_tn_desc* q = d + i;
    REQUIRE(q->M % 128 == 0 && q->N % 64 == 0 && q->n_src >= 1 && q->n_src <= MML_MAX_SRC, "mml_g16_tn: extents (problem %d)", i);
    REQUIRE(!(q->c_bf16 && q->accumulate), "mml_g16_tn: accumulation needs an fp32 output");
    for (int64_t m = 0; m < q->M; ++m)
      for (int c = 0; c < q->N; ++c) {
        double acc = 0.0;
        for (int s = 0; s < q->n_src; ++s) {
          REQUIRE(q->K[s] % 64 == 0, "mml_g16_tn: K %% 64 (problem %d)", i);
          const uint16_t* a = q->A[s] + m * q->lda[s];
          const uint16_t* b = q->B[s] + (int64_t)c * q->ldb[s];
          for (int k = 0; k < q->K[s]; ++k) acc += (double)from_bf16(a[k]) * (double)from_bf16(b[k]);
        }
        float v = (float)acc + (q->bias ? q->bias[c] : 0.f);
        if (q->act == MML_ACT_RELU) v = v > 0.f ? v : 0.f;
        if (q->mask_out) {
          uint32_t* w = q->mask_out + m * q->ldmask + (c >> 5);
          if (v > 0.f) *w |= 1u << (c & 31); else *w &= ~(1u << (c & 31));
        }
        if (q->mask_in && !((q->mask_in[m * q->ldmask + (c >> 5)] >> (c & 31)) & 1u)) v = 0.f;
        if (q->c_bf16) ((uint16_t*)q->C)[m * q->ldc + c] = to_bf16(v);
        else {
          float* o = (float*)q->C + m * q->ldc + c;
          *o = q->accumulate ? *o + v : v;
        }
      }
  }
  return MML_OK;
}

int64_t mml_g16_wgrad_workspace_bytes(const mml_g16_wgrad_desc* d, int32_t n) { (void)d; (void)n; return 256; }
int mml_g16_wgrad(const mml_g16_wgrad_desc* d, int32_t n, void* workspace, int64_t workspace_bytes, int32_t phase,
                  mml_stream_t stream) {
  (void)workspace; (void)workspace_bytes; (void)stream;
  REQUIRE(d, "mml_g16_wgrad: descriptor array is null");
  REQUIRE(phase >= 0 && phase <= 2, "mml_g16_wgrad: phase must be 0, 1 or 2");
  if (phase == 1) return MML_OK;  /* (everything happens in the reduction phase here) */
  for (int i = 0; i < n; ++i) {
    const mml_g16_wgrad_desc* q = d + i;
    REQUIRE(q->M % 64 == 0 && q->N % 128 == 0 && q->K % 128 == 0, "mml_g16_wgrad: extents (problem %d)", i);
    for (int r = 0; r < q->N; ++r) {
      double bsum = 0.0;
      for (int64_t m = 0; m < q->M; ++m) bsum += (double)from_bf16(q->dC[m * q->lddc + r]);
      if (q->dbias) q->dbias[r] = (q->accumulate ? q->dbias[r] : 0.f) + (float)bsum;
      for (int k = 0; k < q->K; ++k) {
        double acc = 0.0;
        for (int64_t m = 0; m < q->M; ++m)
          acc += (double)from_bf16(q->dC[m * q->lddc + r]) * (double)from_bf16(q->A[m * q->lda + k]);
        float* o = q->dW + (int64_t)r * q->lddw + k;
        *o = (q->accumulate ? *o : 0.f) + (float)acc;
      }
    }
  }
  return MML_OK;
}
