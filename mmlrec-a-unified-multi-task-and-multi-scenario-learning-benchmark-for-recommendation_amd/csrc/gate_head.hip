// K4 gate (skinny linear + softmax + expert mix) and K5 prediction heads (+BCE) -- forward and backward.
//
// K4 restates gate_dnn_final_layer -> softmax(1) -> matmul([B,1,Ne],[B,Ne,H]) of MMoE (model/mmoe.py:80-88) and
// of PLE's CGC layer (model/ple.py:127-152).  K5 restates tower_dnn_final_layer (Linear(H->1,bias=False)),
// PredictionLayer (model/utils.py:242-248), the domain-mask product (model/mmoe.py:101-106) and the summed
// F.binary_cross_entropy with its backward (model/basemodel.py:294-296, :312).
//
// Both are bandwidth-bound row kernels: ONE WAVEFRONT PER SAMPLE, lanes strided over the feature axis so every
// wave-instruction touches one contiguous 256-byte run of a row; dot products finish with a 64-lane butterfly.
// Parameter gradients (dWg, dw, dbias) and the loss are accumulated per workgroup (LDS float atomics / registers),
// written once per workgroup to a slab and summed in fixed order by the shared reduce kernel.
#include "common.hpp"
#include "reduce.hpp"
#include "rows_fast.hpp"

namespace mml {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

constexpr int ROW_BLOCK = 256;               // 4 waves = 4 samples in flight per workgroup
constexpr int ROW_WAVES = ROW_BLOCK / kWave;

// ------------------------------------------------------------------------------------------------
// gate forward
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(ROW_BLOCK) void gate_mix_fwd_kernel(const mml_gate_group g) {
  const int lane = threadIdx.x & 63;
  const int64_t wave0 = (int64_t)blockIdx.x * ROW_WAVES + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * ROW_WAVES;
  float am_mix = 0.f;
  for (int64_t b = wave0; b < g.B; b += nwaves) {
    for (int gi = 0; gi < g.n_gates; ++gi) {
      const mml_gate_desc& d = g.gate[gi];
      const float* Grow = d.G + b * d.ldg;
      float logit[MML_MAX_EXPERTS];
#pragma unroll
      for (int e = 0; e < MML_MAX_EXPERTS; ++e) {
        logit[e] = -INFINITY;
        if (e < d.ne) {
          float part = 0.f;
          for (int k = lane; k < d.Gd; k += 64) part += Grow[k] * d.Wg[e * d.Gd + k];
          logit[e] = wave_sum(part);
        }
      }
      float m = logit[0];
#pragma unroll
      for (int e = 1; e < MML_MAX_EXPERTS; ++e) m = fmaxf(m, logit[e]);
      float den = 0.f;
#pragma unroll
      for (int e = 0; e < MML_MAX_EXPERTS; ++e) {
        logit[e] = (e < d.ne) ? expf(logit[e] - m) : 0.f;
        den += logit[e];
      }
      const float inv = 1.f / den;
      float mine = 0.f;
#pragma unroll
      for (int e = 0; e < MML_MAX_EXPERTS; ++e) {
        logit[e] *= inv;
        if (lane == e) mine = logit[e];
      }
      if (lane < d.ne) d.P[b * d.ldp + lane] = mine;
      for (int h = lane; h < g.H; h += 64) {
        float acc = 0.f;
#pragma unroll
        for (int e = 0; e < MML_MAX_EXPERTS; ++e)
          if (e < d.ne) {
            const int x = d.expert[e];
            acc += logit[e] * g.E[x][b * g.lde[x] + h];
          }
        d.mix[b * d.ldmix + h] = acc;
        amax_acc(am_mix, acc);
      }
    }
  }
  amax_flush(am_mix, g.amax_mix);
}

// ------------------------------------------------------------------------------------------------
// gate backward
// LDS: per-workgroup dWg accumulators [sum_g ne_g * Gd_g] floats + per-wave coefficient tables [gates][experts].
// ------------------------------------------------------------------------------------------------
struct GateBwdAux {
  int32_t wg_off[MML_MAX_GATES];  // float offset of gate g's dWg block inside the per-workgroup slab
  int32_t wg_total;
  int32_t pad_;
  float* slab;                    // [gridDim.x][wg_total]
};

__global__ __launch_bounds__(ROW_BLOCK) void gate_mix_bwd_kernel(const mml_gate_group g, const GateBwdAux aux) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* wacc = smem;                                                          // [wg_total]
  float* coef_all = smem + aux.wg_total;                                       // [ROW_WAVES][gates*experts]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float* coef = coef_all + wave * (MML_MAX_GATES * MML_MAX_EXPERTS);
  for (int i = threadIdx.x; i < aux.wg_total; i += ROW_BLOCK) wacc[i] = 0.f;
  __syncthreads();

  const int64_t wave0 = (int64_t)blockIdx.x * ROW_WAVES + wave;
  const int64_t nwaves = (int64_t)gridDim.x * ROW_WAVES;
  float am_dg = 0.f, am_de = 0.f;
  for (int64_t b = wave0; b < g.B; b += nwaves) {
    for (int i = lane; i < MML_MAX_GATES * MML_MAX_EXPERTS; i += 64) coef[i] = 0.f;
    // phase A: per gate softmax backward, dG, dWg
    for (int gi = 0; gi < g.n_gates; ++gi) {
      const mml_gate_desc& d = g.gate[gi];
      if (!d.active) continue;
      const float* dm = d.dmix + b * d.lddmix;
      float dp[MML_MAX_EXPERTS];
#pragma unroll
      for (int e = 0; e < MML_MAX_EXPERTS; ++e) {
        dp[e] = 0.f;
        if (e < d.ne) {
          const int x = d.expert[e];
          const float* Er = g.E[x] + b * g.lde[x];
          float part = 0.f;
          for (int h = lane; h < g.H; h += 64) part += dm[h] * Er[h];
          dp[e] = wave_sum(part);
        }
      }
      float p[MML_MAX_EXPERTS];
      float dot = 0.f;
#pragma unroll
      for (int e = 0; e < MML_MAX_EXPERTS; ++e) {
        p[e] = (e < d.ne) ? d.P[b * d.ldp + e] : 0.f;
        dot += p[e] * dp[e];
      }
#pragma unroll
      for (int e = 0; e < MML_MAX_EXPERTS; ++e) {
        dp[e] = p[e] * (dp[e] - dot);  // dlogit_e
        if (e < d.ne && lane == 0) coef[gi * MML_MAX_EXPERTS + d.expert[e]] = p[e];
      }
      const float* Grow = d.G + b * d.ldg;
      for (int k = lane; k < d.Gd; k += 64) {
        const float gk = Grow[k];
        float dg = 0.f;
#pragma unroll
        for (int e = 0; e < MML_MAX_EXPERTS; ++e)
          if (e < d.ne) {
            dg += dp[e] * d.Wg[e * d.Gd + k];
            atomicAdd(&wacc[aux.wg_off[gi] + e * d.Gd + k], dp[e] * gk);  // ds_add_f32
          }
        if (d.g_relu && !(gk > 0.f)) dg = 0.f;
        d.dG[b * d.lddg + k] = dg;
        amax_acc(am_dg, dg);
      }
    }
    __builtin_amdgcn_wave_barrier();
    // phase B: dE_x = relu'(E_x) * sum_g coef[g][x] * dmix_g
    for (int h = lane; h < g.H; h += 64) {
      float dmv[MML_MAX_GATES];
#pragma unroll
      for (int gi = 0; gi < MML_MAX_GATES; ++gi)
        dmv[gi] = (gi < g.n_gates && g.gate[gi].active) ? g.gate[gi].dmix[b * g.gate[gi].lddmix + h] : 0.f;
#pragma unroll
      for (int x = 0; x < MML_MAX_EXPERTS; ++x)
        if (x < g.n_experts) {
          float acc = 0.f;
#pragma unroll
          for (int gi = 0; gi < MML_MAX_GATES; ++gi) acc += coef[gi * MML_MAX_EXPERTS + x] * dmv[gi];
          if (g.e_relu && !(g.E[x][b * g.lde[x] + h] > 0.f)) acc = 0.f;
          g.dE[x][b * g.ldde[x] + h] = acc;
          amax_acc(am_de, acc);
        }
    }
    __builtin_amdgcn_wave_barrier();
  }
  amax_flush(am_dg, g.amax_dG);
  amax_flush(am_de, g.amax_dE);
  __syncthreads();
  float* out = aux.slab + (int64_t)blockIdx.x * aux.wg_total;
  for (int i = threadIdx.x; i < aux.wg_total; i += ROW_BLOCK) out[i] = wacc[i];
}

// ------------------------------------------------------------------------------------------------
// heads
// ------------------------------------------------------------------------------------------------
constexpr int HEAD_SLOTS = 4;  // H <= 256

struct HeadAux {
  float* slab;       // [gridDim.x][n_heads*(H_max+1) + 1]  (dw | dbias per head, then loss)
  int32_t stride;    // floats per workgroup
  int32_t hmax;
  int32_t train;
  int32_t pad_;
};

__global__ __launch_bounds__(ROW_BLOCK) void head_kernel(const mml_head_group g, const HeadAux aux) {
  __shared__ float red[ROW_WAVES][MML_MAX_HEADS * (HEAD_SLOTS * 64 + 1) + 1];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float dwacc[MML_MAX_HEADS][HEAD_SLOTS];
  float dbacc[MML_MAX_HEADS];
  float lossacc = 0.f;
  float am_dh = 0.f;
#pragma unroll
  for (int t = 0; t < MML_MAX_HEADS; ++t) {
    dbacc[t] = 0.f;
#pragma unroll
    for (int s = 0; s < HEAD_SLOTS; ++s) dwacc[t][s] = 0.f;
  }
  const int64_t wave0 = (int64_t)blockIdx.x * ROW_WAVES + wave;
  const int64_t nwaves = (int64_t)gridDim.x * ROW_WAVES;
  for (int64_t b = wave0; b < g.B; b += nwaves) {
#pragma unroll
    for (int t = 0; t < MML_MAX_HEADS; ++t) {
      if (t >= g.n_heads) continue;
      const mml_head_desc& d = g.head[t];
      const float* hr = d.Hin + b * d.ldh;
      float hv[HEAD_SLOTS], wv[HEAD_SLOTS];
      float part = 0.f;
#pragma unroll
      for (int s = 0; s < HEAD_SLOTS; ++s) {
        const int h = lane + 64 * s;
        hv[s] = 0.f;
        wv[s] = 0.f;
        if (h < d.H) {
          hv[s] = hr[h];
          wv[s] = d.w[h];
          if (d.w2) wv[s] *= d.w2[h];
          part += hv[s] * wv[s];
        }
      }
      float logit = wave_sum(part) + d.bias[0];
      for (int i = 0; i < d.n_bias2; ++i) logit += d.bias2[i];
      const float p = 1.f / (1.f + expf(-logit));
      const float m = (d.mask_col >= 0 && g.mask) ? g.mask[b * g.ldmask + d.mask_col] : 1.f;
      const float pm = p * m;
      if (lane == 0) g.prob[b * g.ldprob + t] = pm;
      if (aux.train) {
        float dpm;
        if (g.y) {
          const float y = g.y[b * g.ldy + t];
          // F.binary_cross_entropy: log terms clamped at -100; backward divides by max(p(1-p), 1e-12)
          const float lp = bce_log_clamp(logf(pm));
          const float l1p = bce_log_clamp(log1pf(-pm));
          if (lane == 0) lossacc += -(y * lp + (1.f - y) * l1p);
          dpm = (pm - y) / fmaxf((1.f - pm) * pm, 1e-12f);
        } else {
          dpm = g.dprob[b * g.lddprob + t];
        }
        const float dlogit = dpm * m * p * (1.f - p);
        if (lane == 0) dbacc[t] += dlogit;
#pragma unroll
        for (int s = 0; s < HEAD_SLOTS; ++s) {
          const int h = lane + 64 * s;
          if (h < d.H) {
            dwacc[t][s] += dlogit * hv[s];
            float dh = dlogit * wv[s];
            if (d.h_relu && !(hv[s] > 0.f)) dh = 0.f;
            d.dH[b * d.lddh + h] = dh;
            amax_acc(am_dh, dh);
          }
        }
      }
    }
  }
  if (!aux.train) return;
  amax_flush(am_dh, g.amax_dH);
  // workgroup reduction: waves -> LDS -> fixed-order sum by wave 0 -> slab
  const int per_head = HEAD_SLOTS * 64 + 1;
#pragma unroll
  for (int t = 0; t < MML_MAX_HEADS; ++t) {
    if (t >= g.n_heads) continue;
#pragma unroll
    for (int s = 0; s < HEAD_SLOTS; ++s) red[wave][t * per_head + s * 64 + lane] = dwacc[t][s];
    if (lane == 0) red[wave][t * per_head + HEAD_SLOTS * 64] = dbacc[t];
  }
  if (lane == 0) red[wave][MML_MAX_HEADS * per_head] = lossacc;
  __syncthreads();
  float* out = aux.slab + (int64_t)blockIdx.x * aux.stride;
  for (int i = threadIdx.x; i < g.n_heads * (aux.hmax + 1) + 1; i += ROW_BLOCK) {
    int src;
    if (i == g.n_heads * (aux.hmax + 1)) {
      src = MML_MAX_HEADS * per_head;
    } else {
      const int t = i / (aux.hmax + 1), h = i % (aux.hmax + 1);
      src = t * per_head + (h == aux.hmax ? HEAD_SLOTS * 64 : h);
    }
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < ROW_WAVES; ++w) s += red[w][src];
    out[i] = s;
  }
}

static int row_grid(int64_t B) {
  int64_t blocks = cdiv(B, ROW_WAVES);
  if (blocks > 256 * 8) blocks = 256 * 8;  // persistent rows beyond 8 workgroups per CU
  if (blocks < 1) blocks = 1;
  return (int)blocks;
}

static int check_gate_group(const mml_gate_group* g, bool bwd, const char* who) {
  MML_REQUIRE(g, "%s: null group", who);
  MML_REQUIRE(g->n_experts >= 1 && g->n_experts <= MML_MAX_EXPERTS, "%s: n_experts=%d", who, g->n_experts);
  MML_REQUIRE(g->n_gates >= 1 && g->n_gates <= MML_MAX_GATES, "%s: n_gates=%d", who, g->n_gates);
  MML_REQUIRE(g->H > 0 && g->B >= 0, "%s: bad H/B", who);
  for (int x = 0; x < g->n_experts; ++x) {
    MML_REQUIRE(g->E[x] && g->lde[x] >= g->H, "%s: expert %d null or lde too small", who, x);
    MML_REQUIRE(!bwd || (g->dE[x] && g->ldde[x] >= g->H), "%s: dE[%d] null or ldde too small", who, x);
  }
  for (int i = 0; i < g->n_gates; ++i) {
    const mml_gate_desc& d = g->gate[i];
    MML_REQUIRE(d.ne >= 1 && d.ne <= MML_MAX_EXPERTS && d.Gd > 0, "%s: gate %d bad ne/Gd", who, i);
    MML_REQUIRE(d.G && d.Wg && d.P && d.ldg >= d.Gd && d.ldp >= d.ne, "%s: gate %d null pointer / ld", who, i);
    for (int e = 0; e < d.ne; ++e)
      MML_REQUIRE(d.expert[e] >= 0 && d.expert[e] < g->n_experts, "%s: gate %d expert index out of range", who, i);
    if (!bwd) MML_REQUIRE(d.mix && d.ldmix >= g->H, "%s: gate %d mix null / ld", who, i);
    if (bwd && d.active)
      MML_REQUIRE(d.dmix && d.dG && d.dWg && d.lddmix >= g->H && d.lddg >= d.Gd, "%s: gate %d backward buffers", who, i);
  }
  return MML_OK;
}

}  // namespace mml

using namespace mml;

extern "C" int mml_gate_mix_fwd(const mml_gate_group* grp, mml_stream_t stream) {
  int rc = check_gate_group(grp, false, "mml_gate_mix_fwd");
  if (rc) return rc;
  if (grp->B == 0) return MML_OK;
  rc = gate_fwd_fast(grp, to_stream(stream));
  if (rc <= 0) return rc;  // handled by the aligned fast path (or failed there)
  if (grp->out_bf16) {     // (bf16 outputs exist in the fast row kernels only: include/mmlrec.h)
    set_error("mml_gate_mix_fwd: out_bf16 on a shape the fast row kernel does not serve");
    return MML_ERR_UNSUPPORTED;
  }
  MML_LAUNCH(gate_mix_fwd_kernel, dim3(row_grid(grp->B)), dim3(ROW_BLOCK), 0, to_stream(stream), *grp);
  return check_launch("mml_gate_mix_fwd");
}

static int gate_bwd_layout(const mml_gate_group* g, GateBwdAux& aux) {
  int off = 0;
  for (int i = 0; i < g->n_gates; ++i) {
    aux.wg_off[i] = off;
    if (g->gate[i].active) off += g->gate[i].ne * g->gate[i].Gd;
  }
  aux.wg_total = off;
  return off;
}

extern "C" int64_t mml_gate_mix_bwd_workspace_bytes(const mml_gate_group* grp) {
  if (!grp) return 0;
  int64_t tot = 0;
  for (int i = 0; i < grp->n_gates && i < MML_MAX_GATES; ++i) tot += (int64_t)grp->gate[i].ne * grp->gate[i].Gd;
  return (int64_t)256 * 8 * tot * 4 + 256;  // either path uses at most 2048 workgroups
}

extern "C" int mml_gate_mix_bwd(const mml_gate_group* grp, void* workspace, int64_t workspace_bytes,
                                mml_stream_t stream) {
  return mml_gate_mix_bwd_phase(grp, workspace, workspace_bytes, 0, stream);
}

// phase 1: the row kernel (input gradients stored, per-workgroup partial sums of dWg left in the workspace); phase 2: the
// reduction of those partial sums into dWg; 0: both.  (Which kernel phase 1 ran is a function of the group alone, so
// phase 2 finds the same layout.)
extern "C" int mml_gate_mix_bwd_phase(const mml_gate_group* grp, void* workspace, int64_t workspace_bytes, int32_t phase,
                                      mml_stream_t stream) {
  int rc = check_gate_group(grp, true, "mml_gate_mix_bwd");
  if (rc) return rc;
  MML_REQUIRE(phase >= 0 && phase <= 2, "mml_gate_mix_bwd_phase: bad phase");
  if (grp->B == 0) return MML_OK;
  {
    GateFastAux fa{};
    if (gate_fast_config(grp, true, fa)) {
      MML_REQUIRE(workspace && (int64_t)fa.grid * fa.wg_total * 4 <= workspace_bytes,
                  "mml_gate_mix_bwd: workspace too small");
      fa.slab = static_cast<float*>(workspace);
      rc = 0;
      if (phase != 2) rc = gate_bwd_fast(grp, fa, to_stream(stream));
      else if (!gate_bwd_fast_serves(grp, fa)) rc = 1;
      if (rc < 0) return rc;
      if (rc == 0) {
        if (phase == 1) return MML_OK;
        ReduceLaunch R{};
        int64_t start = 0;
        for (int i = 0; i < grp->n_gates; ++i) {
          const mml_gate_desc& d = grp->gate[i];
          if (!d.active) continue;
          ReduceSeg& s = R.seg[R.n++];
          s.slab = fa.slab + fa.wg_off[i]; s.out = d.dWg; s.n = (int64_t)d.ne * d.Gd; s.sstride = fa.wg_total;
          s.cols = d.Gd; s.ldo = d.Gd; s.accumulate = 0; s.start = start; s.S = fa.grid;
          start += s.n;
        }
        R.total = start;
        return launch_slab_reduce(R, to_stream(stream), "mml_gate_mix_bwd(reduce)");
      }
    }
  }
  if (grp->out_bf16) {  // (bf16 outputs exist in the fast row kernels only: include/mmlrec.h)
    set_error("mml_gate_mix_bwd: out_bf16 on a shape the fast row kernel does not serve");
    return MML_ERR_UNSUPPORTED;
  }
  GateBwdAux aux{};
  const int tot = gate_bwd_layout(grp, aux);
  const int grid = row_grid(grp->B);
  MML_REQUIRE(workspace && (int64_t)grid * tot * 4 <= workspace_bytes, "mml_gate_mix_bwd: workspace too small");
  const size_t lds = (size_t)(tot + ROW_WAVES * MML_MAX_GATES * MML_MAX_EXPERTS) * 4;
  MML_REQUIRE(lds <= 64 * 1024, "mml_gate_mix_bwd: gate weights too large for the LDS accumulators (%zu B)", lds);
  aux.slab = static_cast<float*>(workspace);
  if (phase != 2) {
    MML_LAUNCH(gate_mix_bwd_kernel, dim3(grid), dim3(ROW_BLOCK), lds, to_stream(stream), *grp, aux);
    rc = check_launch("mml_gate_mix_bwd");
    if (rc) return rc;
  }
  if (phase == 1) return MML_OK;
  ReduceLaunch R{};
  int64_t start = 0;
  for (int i = 0; i < grp->n_gates; ++i) {
    const mml_gate_desc& d = grp->gate[i];
    if (!d.active) continue;
    ReduceSeg& s = R.seg[R.n++];
    s.slab = aux.slab + aux.wg_off[i]; s.out = d.dWg; s.n = (int64_t)d.ne * d.Gd; s.sstride = tot;
    s.cols = d.Gd; s.ldo = d.Gd; s.accumulate = 0; s.start = start; s.S = grid;
    start += s.n;
  }
  R.total = start;
  return launch_slab_reduce(R, to_stream(stream), "mml_gate_mix_bwd(reduce)");
}

static bool head_group_gated(const mml_head_group* g) {
  for (int t = 0; t < g->n_heads; ++t)
    if (g->head[t].gate) return true;
  return false;
}

static int check_head_group(const mml_head_group* g, bool train, const char* who, int& hmax) {
  MML_REQUIRE(g, "%s: null group", who);
  MML_REQUIRE(g->n_heads >= 1 && g->n_heads <= MML_MAX_HEADS && g->B >= 0, "%s: n_heads=%d", who, g->n_heads);
  MML_REQUIRE(g->prob && g->ldprob >= g->n_heads, "%s: prob null / ldprob", who);
  MML_REQUIRE(!train || (g->y && g->ldy >= g->n_heads) || (g->dprob && g->lddprob >= g->n_heads),
              "%s: training needs y or dprob", who);
  hmax = 0;
  for (int t = 0; t < g->n_heads; ++t) {
    const mml_head_desc& d = g->head[t];
    MML_REQUIRE(d.Hin && d.w && d.bias && d.H >= 1 && d.H <= HEAD_SLOTS * 64 && d.ldh >= d.H,
                "%s: head %d malformed (H=%d, max %d)", who, t, d.H, HEAD_SLOTS * 64);
    MML_REQUIRE(d.n_bias2 == 0 || d.bias2, "%s: head %d bias2 null", who, t);
    MML_REQUIRE(!train || (d.dH && d.dw && d.dbias && d.lddh >= d.H), "%s: head %d backward buffers", who, t);
    MML_REQUIRE(d.mask_col < 0 || g->mask, "%s: head %d wants a mask column but mask is null", who, t);
    MML_REQUIRE(!d.gate || (d.ldgate >= d.H && (!train || (d.dgate && d.lddgate >= d.H))),
                "%s: gated head %d: gate pitch / dgate", who, t);
    MML_REQUIRE(!d.gate || d.gate_act == MML_ACT_NONE || d.gate_act == MML_ACT_SIGMOID || d.gate_act == MML_ACT_SIGMOID2,
                "%s: gated head %d: gate_act must be none, sigmoid or 2 sigmoid", who, t);
    if (d.H > hmax) hmax = d.H;
  }
  return MML_OK;
}

extern "C" int64_t mml_head_workspace_bytes(const mml_head_group* grp) {
  if (!grp) return 0;
  int hmax = 0;
  for (int t = 0; t < grp->n_heads && t < MML_MAX_HEADS; ++t)
    if (grp->head[t].H > hmax) hmax = grp->head[t].H;
  return (int64_t)256 * 8 * (grp->n_heads * (hmax + 1) + 1) * 4 + 256;  // either path uses at most 2048 workgroups
}

extern "C" int mml_head_fwd(const mml_head_group* grp, mml_stream_t stream) {
  int hmax;
  int rc = check_head_group(grp, false, "mml_head_fwd", hmax);
  if (rc) return rc;
  if (grp->B == 0) return MML_OK;
  {
    HeadFastAux fa{};
    if (head_fast_config(grp, false, hmax, fa)) return head_fast(grp, fa, to_stream(stream));
  }
  if (head_group_gated(grp)) {
    set_error("mml_head_fwd: gated heads on a shape the fast row kernel does not serve");
    return MML_ERR_UNSUPPORTED;
  }
  HeadAux aux{};
  aux.hmax = hmax;
  MML_LAUNCH(head_kernel, dim3(row_grid(grp->B)), dim3(ROW_BLOCK), 0, to_stream(stream), *grp, aux);
  return check_launch("mml_head_fwd");
}

extern "C" int mml_head_bce_fwd_bwd(const mml_head_group* grp, void* workspace, int64_t workspace_bytes,
                                    mml_stream_t stream) {
  return mml_head_bce_fwd_bwd_phase(grp, workspace, workspace_bytes, 0, stream);
}

// phase 1: the row kernel (probabilities, input gradients; per-workgroup partial sums of dw / dbias / loss left in the
// workspace); phase 2: their reduction; 0: both
extern "C" int mml_head_bce_fwd_bwd_phase(const mml_head_group* grp, void* workspace, int64_t workspace_bytes, int32_t phase,
                                          mml_stream_t stream) {
  MML_REQUIRE(phase >= 0 && phase <= 2, "mml_head_bce_fwd_bwd_phase: bad phase");
  int hmax;
  int rc = check_head_group(grp, true, "mml_head_bce_fwd_bwd", hmax);
  if (rc) return rc;
  if (grp->B == 0) return MML_OK;
  int grid = row_grid(grp->B);
  HeadAux aux{};
  aux.hmax = hmax;
  aux.stride = grp->n_heads * (hmax + 1) + 1;
  aux.train = 1;
  HeadFastAux fa{};
  const bool fast = head_fast_config(grp, true, hmax, fa) != 0;
  if (fast) grid = fa.grid;
  MML_REQUIRE(workspace && (int64_t)grid * aux.stride * 4 <= workspace_bytes, "mml_head_bce_fwd_bwd: workspace too small");
  aux.slab = static_cast<float*>(workspace);
  if (phase != 2) {
    if (fast) {
      fa.slab = aux.slab; fa.stride = aux.stride; fa.train = 1;
      rc = head_fast(grp, fa, to_stream(stream));
    } else {
      if (grp->dh_bf16 || head_group_gated(grp)) {
        set_error("mml_head_bce_fwd_bwd: dh_bf16 / gated heads on a shape the fast row kernel does not serve");
        return MML_ERR_UNSUPPORTED;
      }
      MML_LAUNCH(head_kernel, dim3(grid), dim3(ROW_BLOCK), 0, to_stream(stream), *grp, aux);
      rc = check_launch("mml_head_bce_fwd_bwd");
    }
    if (rc) return rc;
  }
  if (phase == 1) return MML_OK;
  ReduceLaunch R{};
  int64_t start = 0;
  for (int t = 0; t < grp->n_heads; ++t) {
    const mml_head_desc& d = grp->head[t];
    ReduceSeg& w = R.seg[R.n++];
    w.slab = aux.slab + t * (hmax + 1); w.out = d.dw; w.n = d.H; w.sstride = aux.stride; w.cols = d.H; w.ldo = d.H;
    w.start = start; w.S = grid;
    start += d.H;
    ReduceSeg& bb = R.seg[R.n++];
    bb.slab = aux.slab + t * (hmax + 1) + hmax; bb.out = d.dbias; bb.n = 1; bb.sstride = aux.stride; bb.cols = 1;
    bb.ldo = 1; bb.start = start; bb.S = grid;
    start += 1;
  }
  if (grp->loss) {
    ReduceSeg& l = R.seg[R.n++];
    l.slab = aux.slab + grp->n_heads * (hmax + 1); l.out = grp->loss; l.n = 1; l.sstride = aux.stride; l.cols = 1;
    l.ldo = 1; l.start = start; l.S = grid;
    start += 1;
  }
  R.total = start;
  return launch_slab_reduce(R, to_stream(stream), "mml_head_bce_fwd_bwd(reduce)");
}

// The reductions (phase 2) of several head / gate groups in ONE launch: nothing but the optimizer (and the host, for the
// loss) reads dw / dbias / loss / dWg, and inside a step's graph every launch takes >= 4.6 us from start to end.
extern "C" int mml_rows_reduce_batch(const mml_rows_reduce_item* items, int32_t n, mml_stream_t stream) {
  MML_REQUIRE(n >= 0 && (n == 0 || items), "mml_rows_reduce_batch: bad item array");
  ReduceLaunch all{};
  reduce_collect(&all);
  int rc = MML_OK;
  for (int i = 0; i < n && rc == MML_OK; ++i) {
    const mml_rows_reduce_item& it = items[i];
    if (it.kind == MML_ROWS_REDUCE_HEAD)
      rc = mml_head_bce_fwd_bwd_phase(static_cast<const mml_head_group*>(it.group), it.workspace, it.workspace_bytes, 2, stream);
    else if (it.kind == MML_ROWS_REDUCE_GATE)
      rc = mml_gate_mix_bwd_phase(static_cast<const mml_gate_group*>(it.group), it.workspace, it.workspace_bytes, 2, stream);
    else if (it.kind == MML_ROWS_REDUCE_TOWER_HEAD)
      rc = mml_tower_head_fwd_bwd(static_cast<const mml_tower_head_group*>(it.group), it.workspace, it.workspace_bytes, 2, stream);
    else {
      set_error("mml_rows_reduce_batch: item %d: kind %d", i, it.kind);
      rc = MML_ERR_ARG;
    }
  }
  reduce_collect(nullptr);
  if (rc != MML_OK) return rc;
  return launch_slab_reduce(all, to_stream(stream), "mml_rows_reduce_batch");
}

