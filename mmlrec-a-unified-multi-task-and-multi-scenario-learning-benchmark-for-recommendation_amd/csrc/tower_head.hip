// K5' (round 6): the TOP of a multi-task network in one launch -- the last tower layer (Linear + ReLU), the prediction head
// (tower_dnn_final_layer Linear(H -> 1, bias = False) + PredictionLayer bias + sigmoid), the optional domain-mask product,
// the summed BCE, and their whole backward down to dL/d(tower input) (reference model/mmoe.py:93-108, model/utils.py:146-161,
// :242-248, model/basemodel.py:294-296).  Nothing below the loss depends on anything but the tower's input, so forward and
// backward of these layers need no trip through memory between them: the tower output h never leaves the registers.
//
// Unfused (round 5) the headline step spent three launches here -- towers forward 26 us, head + BCE 28 us, towers' input gradient
// 24 us at B = 65 536 -- moving the tower outputs and their gradients four times.  Here a wave owns 32-row blocks of the batch
// like gemm_ws_kernel's (same operand images: the tower weight's pre-cut fp16 planes in BOTH layouts stay in LDS, the input
// rows come from HBM as MFMA fragments):
//   rows -> h = relu(x W^T + b) [3 x NS MFMAs per k-step] -> logit = h . w + bias (lane sums + one half-wave exchange) ->
//   p, masked p, clamped-log BCE, dlogit (the head kernel's expressions) -> dH = dlogit w relu'(h) -> cut into planes IN THE
//   ACCUMULATOR LAYOUT (a lane's eight values of a 16-column block are exactly an A fragment: the planes' k order was chosen
//   for this) -> dX = dH W [3 x NS2 MFMAs per k-step] -> row-major turn -> stores.
// dH is scaled per 32-row block (its own largest magnitude: the rows of a block only meet their own products), so no magnitude
// has to travel between the two GEMMs.  Written: prob, dH (the tower's weight gradient reads it), dX; per-workgroup partial
// sums of the head's dw / dbias and the loss (fixed-order reduction: phase 2).
#include "common.hpp"
#include "lds_async.hpp"
#include "reduce.hpp"

#include <stdlib.h>

namespace mml {

using thf32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int TH_MAX = MML_MAX_HEADS;
constexpr int TH_W_BYTES = 64 * 1024;                 // each of the two plane images
constexpr int TH_BIAS_OFF = 2 * TH_W_BYTES;           // <= 128 tower biases, then <= 128 head weights
constexpr int TH_RED_OFF = TH_BIAS_OFF + 1024;        // 8 waves x (128 dw + dbias + loss) partial sums
constexpr int TH_AMAX_OFF = TH_RED_OFF + 8 * 136 * 4; // two magnitude words
constexpr int TH_TURN_OFF = TH_AMAX_OFF + 64;         // eight waves x 2 KiB
constexpr int TH_LDS_BYTES = TH_TURN_OFF + 8 * 2048;
static_assert(TH_LDS_BYTES <= 160 * 1024, "tower + head kernel LDS budget");

struct ThProblem {
  const float* A;            // [M, K] tower input
  const uint32_t* amaxA;
  const uint32_t* planesF;   // W [N, K], MML_PLANES_ROWS, pitch ldpf words
  const uint32_t* planesB;   // W [N, K], MML_PLANES_COLS, pitch ldpb words
  const int32_t* kexpF;
  const int32_t* kexpB;
  const float* bias1;        // [N] or null
  const float* w;            // [N] head weight
  const float* hbias;        // [1]
  const float* hbias2;       // [n_hbias2] or null
  float* dH;                 // [M, N]
  float* dA;                 // [M, K]
  uint32_t* amax_dH;
  uint32_t* amax_dA;
  float* slab;               // [wg_per_prob][N + 1]: dw, dbias partial sums of this problem
  float* loss_slab;          // [wg_per_prob] loss partial sums of this problem (the problems' arrays stand back to back)
  int64_t lda, ldpf, ldpb, lddh, ldda;
  int32_t n_hbias2, mask_col, head, pad_;
};
struct ThLaunch {
  int64_t M;
  float* prob;
  const float* y;
  const float* mask;
  int64_t ldprob, ldy, ldmask;
  int32_t n_prob, wg_per_prob;
  ThProblem p[TH_MAX];
};
static_assert(sizeof(ThLaunch) <= 4096, "ThLaunch must fit the kernel-argument block");

__device__ __forceinline__ uint32_t th_amax_load(const uint32_t* p) {
  uint32_t m = 0;
#pragma unroll
  for (int i = 0; i < MML_AMAX_WORDS; ++i) m = p[i] > m ? p[i] : m;
  return m;
}
__device__ __forceinline__ int th_scale_exp(uint32_t bits) {  // (gemm.hip's rule: |x| 2^k < 2^15)
  const int e = (int)((bits >> 23) & 0xffu);
  if (e == 255) return 0;
  const int k = 141 - e;
  return k > 110 ? 110 : (k < -110 ? -110 : k);
}
__device__ __forceinline__ float th_pow2(int k) { return __uint_as_float((uint32_t)(127 + k) << 23); }

// NS: tower width / 32 (the head's H), NS2: tower input width / 32
template <int NS, int NS2>
__global__ __launch_bounds__(512, 2) void tower_head_kernel(const ThLaunch L) {
  constexpr int N = NS * 32, K = NS2 * 32;
  constexpr int KBF = K / 16;   // k-steps of the forward GEMM
  constexpr int KBB = N / 16;   // k-steps of the input-gradient GEMM
  constexpr int D = 4, G = KBF / D;
  static_assert(KBF % D == 0, "tower input width must be a multiple of 64");
  static_assert(KBF * 2 * NS * 1024 <= TH_W_BYTES && KBB * 2 * NS2 * 1024 <= TH_W_BYTES, "plane images");
  __shared__ __attribute__((aligned(16))) float lds[TH_LDS_BYTES / 4];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, h = lane >> 5;
  const int pi = (int)blockIdx.x / L.wg_per_prob;
  const int wl = (int)blockIdx.x - pi * L.wg_per_prob;
  const ThProblem& P = L.p[pi];
  const int64_t M = L.M;

  // ---- both plane images -> LDS, fragment order [k-step][plane][32-column sub-tile][lane] x 16 bytes (gemm_ws_kernel's) ----
  {
    uint32_t* const lw = reinterpret_cast<uint32_t*>(lds);
    {  // forward: MML_PLANES_ROWS, row = output column, 16 words per k-step
      const int total = NS * 32 * KBF * 4;
      for (int idx = tid; idx < total; idx += 512) {
        const int c = idx & 3, b = (idx >> 2) % KBF, n = idx / (4 * KBF);
        const uint4 v = *reinterpret_cast<const uint4*>(P.planesF + (int64_t)n * P.ldpf + 16 * b + 4 * c);
        *reinterpret_cast<uint4*>(lw + ((((b * 2 + (c >> 1)) * NS + (n >> 5)) * 64 + (c & 1) * 32 + (n & 31)) * 4)) = v;
      }
    }
    {  // input gradient: MML_PLANES_COLS, row = reduction index (tower column), word e = 8 plane + 4 lane half + i
      uint32_t* const lb = lw + TH_W_BYTES / 4;
      constexpr int NO = NS2 * 32;
      const int total = KBB * 16 * NO;
      for (int idx = tid; idx < total; idx += 512) {
        const int c = idx % NO, r = idx / NO;
        const int b = r >> 4, e = r & 15;
        lb[(((b * 2 + (e >> 3)) * NS2 + (c >> 5)) * 64 + ((e >> 2) & 1) * 32 + (c & 31)) * 4 + (e & 3)] =
            P.planesB[(int64_t)r * P.ldpb + c];
      }
    }
    if (tid < N) {
      lds[TH_BIAS_OFF / 4 + tid] = P.bias1 ? P.bias1[tid] : 0.f;
      lds[TH_BIAS_OFF / 4 + 128 + tid] = P.w[tid];
    }
    if (tid < 2) reinterpret_cast<uint32_t*>(lds)[TH_AMAX_OFF / 4 + tid] = 0u;
  }
  __syncthreads();

  const int kA = __builtin_amdgcn_readfirstlane(th_scale_exp(th_amax_load(P.amaxA)));
  const int kF = __builtin_amdgcn_readfirstlane(*P.kexpF);
  const int kB = __builtin_amdgcn_readfirstlane(*P.kexpB);
  const float sA = th_pow2(kA);
  const float invF = th_pow2(-kA) * th_pow2(-kF);
  float hb = P.hbias[0];
  for (int i = 0; i < P.n_hbias2; ++i) hb += P.hbias2[i];

  const int64_t nrb = (M + 31) >> 5;
  const int stride = L.wg_per_prob * 8;
  int64_t rb = wl * 8 + wave;
  float am_dh = 0.f, am_da = 0.f, lossacc = 0.f, dbacc = 0.f;
  float dwacc[NS][16];
#pragma unroll
  for (int ni = 0; ni < NS; ++ni)
#pragma unroll
    for (int r = 0; r < 16; ++r) dwacc[ni][r] = 0.f;

  auto arow = [&](const int64_t rb_) __attribute__((always_inline)) {
    int64_t row = rb_ * 32 + l31;
    row = row < M ? row : M - 1;
    return P.A + row * P.lda + 4 * h;
  };
  const f32x4_t* const wfragF = reinterpret_cast<const f32x4_t*>(lds) + lane;
  const f32x4_t* const wfragB = reinterpret_cast<const f32x4_t*>(lds) + TH_W_BYTES / 16 + lane;
  float* const turn = lds + TH_TURN_OFF / 4 + wave * 512;

  if (rb < nrb) {
    float4 r0[D], r1[D];
    const float* ap = arow(rb);
#pragma unroll
    for (int j = 0; j < D; ++j) {
      r0[j] = *reinterpret_cast<const float4*>(ap + 16 * j);
      r1[j] = *reinterpret_cast<const float4*>(ap + 16 * j + 8);
    }
    for (; rb < nrb; rb += stride) {
      const float* const ap_next = arow(rb + stride < nrb ? rb + stride : rb);
      const int64_t row = rb * 32 + l31;
      const bool ok = row < M;
      const int64_t lrow = ok ? row : M - 1;
      // the row's label and mask, requested before the GEMM that hides them
      const float yv = L.y[lrow * L.ldy + P.head];
      const float mv = (P.mask_col >= 0 && L.mask) ? L.mask[lrow * L.ldmask + P.mask_col] : 1.f;
      // ---- forward GEMM: acc[ni] = W planes x row fragments
      thf32x16 acc[NS];
#pragma unroll
      for (int ni = 0; ni < NS; ++ni)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[ni][r] = 0.f;
#pragma unroll 1
      for (int g = 0; g < G; ++g) {
        const float* const pf = (g + 1 < G) ? ap + 16 * D * (g + 1) : ap_next;
        const f32x4_t* const wg_ = wfragF + g * 2 * D * NS * 64;
#pragma unroll
        for (int j = 0; j < D; ++j) {
          const float x[8] = {r0[j].x, r0[j].y, r0[j].z, r0[j].w, r1[j].x, r1[j].y, r1[j].z, r1[j].w};
          F16Cut c;
          f16_cut_a2(x, sA, c, 0);
          f16_cut_a2(x, sA, c, 2);
          f16_cut_b2(x, sA, c, 0);
          f16_cut_b2(x, sA, c, 2);
          f16_cut_l(c, 0);
          f16_cut_l(c, 1);
          f16_cut_l(c, 2);
          f16_cut_l(c, 3);
          f16x8 Ah, Al;
          f16_cut_done(c, Ah, Al);
          r0[j] = *reinterpret_cast<const float4*>(pf + 16 * j);
          r1[j] = *reinterpret_cast<const float4*>(pf + 16 * j + 8);
          f16x8 bh[NS], bl[NS];
#pragma unroll
          for (int ni = 0; ni < NS; ++ni) {
            bh[ni] = __builtin_bit_cast(f16x8, wg_[((j * 2 + 0) * NS + ni) * 64]);
            bl[ni] = __builtin_bit_cast(f16x8, wg_[((j * 2 + 1) * NS + ni) * 64]);
          }
#pragma unroll
          for (int ni = 0; ni < NS; ++ni) acc[ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl[ni], Ah, acc[ni], 0, 0, 0);
#pragma unroll
          for (int ni = 0; ni < NS; ++ni) acc[ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[ni], Al, acc[ni], 0, 0, 0);
#pragma unroll
          for (int ni = 0; ni < NS; ++ni) acc[ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[ni], Ah, acc[ni], 0, 0, 0);
        }
      }
      // ---- h = relu(. + bias) in the accumulator layout (lane = row, register r = column 8 (r >> 2) + 4 h + (r & 3) of
      // sub-tile ni); the head's logit: this lane's 16 NS columns, then the other half of the row
      float part = 0.f;
#pragma unroll
      for (int ni = 0; ni < NS; ++ni)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const float4 b4 = *reinterpret_cast<const float4*>(lds + TH_BIAS_OFF / 4 + ni * 32 + 8 * g + 4 * h);
          const float4 w4 = *reinterpret_cast<const float4*>(lds + TH_BIAS_OFF / 4 + 128 + ni * 32 + 8 * g + 4 * h);
          const float v0 = __builtin_fmaxf(acc[ni][4 * g + 0] * invF + b4.x, 0.f);
          const float v1 = __builtin_fmaxf(acc[ni][4 * g + 1] * invF + b4.y, 0.f);
          const float v2 = __builtin_fmaxf(acc[ni][4 * g + 2] * invF + b4.z, 0.f);
          const float v3 = __builtin_fmaxf(acc[ni][4 * g + 3] * invF + b4.w, 0.f);
          acc[ni][4 * g + 0] = v0; acc[ni][4 * g + 1] = v1; acc[ni][4 * g + 2] = v2; acc[ni][4 * g + 3] = v3;
          part += v0 * w4.x + v1 * w4.y + v2 * w4.z + v3 * w4.w;
        }
      const float logit = part + __shfl_xor(part, 32, 64) + hb;
      // ---- PredictionLayer, mask, summed BCE and its derivative (csrc/rows_fast.hip: head_fast_kernel's expressions)
      const float pj = 1.f / (1.f + expf(-logit));
      const float pm = pj * mv;
      if (ok && h == 0) L.prob[row * L.ldprob + P.head] = pm;
      const float lp = bce_log_clamp(logf(pm));
      const float l1p = bce_log_clamp(log1pf(-pm));
      if (ok && h == 0) lossacc += -(yv * lp + (1.f - yv) * l1p);
      const float dpm = (pm - yv) / fmaxf((1.f - pm) * pm, 1e-12f);
      const float dlogit = ok ? dpm * mv * pj * (1.f - pj) : 0.f;
      if (h == 0) dbacc += dlogit;
      // ---- dH = dlogit w relu'(h); the head's dw; the block's largest |dH|
      float amb = 0.f;
#pragma unroll
      for (int ni = 0; ni < NS; ++ni)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const float4 w4 = *reinterpret_cast<const float4*>(lds + TH_BIAS_OFF / 4 + 128 + ni * 32 + 8 * g + 4 * h);
          const float wv[4] = {w4.x, w4.y, w4.z, w4.w};
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float hv = acc[ni][4 * g + j];
            dwacc[ni][4 * g + j] += dlogit * hv;
            const float d = hv > 0.f ? dlogit * wv[j] : 0.f;
            acc[ni][4 * g + j] = d;
            amb = fmaxf(amb, fabsf(d));
          }
        }
      am_dh = fmaxf(am_dh, amb);
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) amb = fmaxf(amb, __shfl_xor(amb, o, 64));
      const int kD = __builtin_amdgcn_readfirstlane(th_scale_exp(__float_as_uint(amb)));
      const float sD = th_pow2(kD);
      const float invB = th_pow2(-kD) * th_pow2(-kB);
      // ---- dH to memory (the tower's weight gradient reads it): row-major through the wave's turn area
      const int tR = lane >> 2, tc = lane & 3;
      const int64_t orow = rb * 32 + tR;
      const bool ok0 = orow < M, ok1 = orow + 16 < M;
      float* const tw = turn + l31 * 16;
      auto turn_store = [&](const float4 va, const float4 vb, float* const dst, const int64_t ldd, const int col0)
                            __attribute__((always_inline)) {
        *reinterpret_cast<float4*>(tw + 4 * ((0 + h) ^ ((l31 >> 2) & 3))) = va;
        *reinterpret_cast<float4*>(tw + 4 * ((2 + h) ^ ((l31 >> 2) & 3))) = vb;
        asm volatile("" ::: "memory");
        const float4 q0 = *reinterpret_cast<const float4*>(turn + tR * 16 + 4 * (tc ^ ((tR >> 2) & 3)));
        const float4 q1 = *reinterpret_cast<const float4*>(turn + (tR + 16) * 16 + 4 * (tc ^ (((tR + 16) >> 2) & 3)));
        float* const d0 = dst + orow * ldd + col0 + 4 * tc;
        if (ok0) *reinterpret_cast<float4*>(d0) = q0;
        if (ok1) *reinterpret_cast<float4*>(d0 + 16 * ldd) = q1;
      };
#pragma unroll
      for (int ni = 0; ni < NS; ++ni) {
        turn_store(make_float4(acc[ni][0], acc[ni][1], acc[ni][2], acc[ni][3]),
                   make_float4(acc[ni][4], acc[ni][5], acc[ni][6], acc[ni][7]), P.dH, P.lddh, ni * 32);
        turn_store(make_float4(acc[ni][8], acc[ni][9], acc[ni][10], acc[ni][11]),
                   make_float4(acc[ni][12], acc[ni][13], acc[ni][14], acc[ni][15]), P.dH, P.lddh, ni * 32 + 16);
      }
      // ---- input-gradient GEMM: the lane's eight dH values of a 16-column block ARE an A fragment (k order of the planes)
      thf32x16 acc2[NS2];
#pragma unroll
      for (int n2 = 0; n2 < NS2; ++n2)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc2[n2][r] = 0.f;
#pragma unroll
      for (int s = 0; s < KBB; ++s) {
        const int ni = s >> 1, r8 = 8 * (s & 1);
        const float x[8] = {acc[ni][r8 + 0], acc[ni][r8 + 1], acc[ni][r8 + 2], acc[ni][r8 + 3],
                            acc[ni][r8 + 4], acc[ni][r8 + 5], acc[ni][r8 + 6], acc[ni][r8 + 7]};
        F16Cut c;
        f16_cut_a2(x, sD, c, 0);
        f16_cut_a2(x, sD, c, 2);
        f16_cut_b2(x, sD, c, 0);
        f16_cut_b2(x, sD, c, 2);
        f16_cut_l(c, 0);
        f16_cut_l(c, 1);
        f16_cut_l(c, 2);
        f16_cut_l(c, 3);
        f16x8 Ah, Al;
        f16_cut_done(c, Ah, Al);
        f16x8 bh[NS2], bl[NS2];
#pragma unroll
        for (int n2 = 0; n2 < NS2; ++n2) {
          bh[n2] = __builtin_bit_cast(f16x8, wfragB[((s * 2 + 0) * NS2 + n2) * 64]);
          bl[n2] = __builtin_bit_cast(f16x8, wfragB[((s * 2 + 1) * NS2 + n2) * 64]);
        }
#pragma unroll
        for (int n2 = 0; n2 < NS2; ++n2) acc2[n2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl[n2], Ah, acc2[n2], 0, 0, 0);
#pragma unroll
        for (int n2 = 0; n2 < NS2; ++n2) acc2[n2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[n2], Al, acc2[n2], 0, 0, 0);
#pragma unroll
        for (int n2 = 0; n2 < NS2; ++n2) acc2[n2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[n2], Ah, acc2[n2], 0, 0, 0);
      }
#pragma unroll
      for (int n2 = 0; n2 < NS2; ++n2) {
        float4 vv[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          vv[g] = make_float4(acc2[n2][4 * g] * invB, acc2[n2][4 * g + 1] * invB, acc2[n2][4 * g + 2] * invB,
                              acc2[n2][4 * g + 3] * invB);
          amax_acc(am_da, vv[g]);
        }
        turn_store(vv[0], vv[1], P.dA, P.ldda, n2 * 32);
        turn_store(vv[2], vv[3], P.dA, P.ldda, n2 * 32 + 16);
      }
      ap = ap_next;
    }
  }

  // ---- magnitudes of what was stored (one atomic per workgroup and slot)
  {
    uint32_t* const words = reinterpret_cast<uint32_t*>(lds) + TH_AMAX_OFF / 4;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      am_dh = fmaxf(am_dh, __shfl_xor(am_dh, o, 64));
      am_da = fmaxf(am_da, __shfl_xor(am_da, o, 64));
    }
    if (lane == 0) {
      atomicMax(words, __float_as_uint(am_dh));
      atomicMax(words + 1, __float_as_uint(am_da));
    }
  }
  // ---- the head's dw / dbias / loss: rows of a wave (32 lanes of each half) -> one value per column; waves -> workgroup
  float* const red = lds + TH_RED_OFF / 4 + wave * 136;
#pragma unroll
  for (int ni = 0; ni < NS; ++ni)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float v = dwacc[ni][r];
#pragma unroll
      for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);   // over the 32 rows of this half
      if (l31 == 0) red[ni * 32 + 8 * (r >> 2) + 4 * h + (r & 3)] = v;
    }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    dbacc += __shfl_xor(dbacc, o, 64);
    lossacc += __shfl_xor(lossacc, o, 64);
  }
  if (lane == 0) {
    red[128] = dbacc;
    red[129] = lossacc;
  }
  __syncthreads();
  {
    const uint32_t* const words = reinterpret_cast<const uint32_t*>(lds) + TH_AMAX_OFF / 4;
    if (tid == 0 && P.amax_dH && words[0]) atomicMax(P.amax_dH + (blockIdx.x & (MML_AMAX_WORDS - 1)), words[0]);
    if (tid == 1 && P.amax_dA && words[1]) atomicMax(P.amax_dA + (blockIdx.x & (MML_AMAX_WORDS - 1)), words[1]);
  }
  float* const out = P.slab + (int64_t)wl * (N + 1);
  for (int i = tid; i < N + 2; i += 512) {
    const int src = i < N ? i : 128 + (i - N);
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < 8; ++w) s += lds[TH_RED_OFF / 4 + w * 136 + src];
    if (i <= N) out[i] = s;
    else P.loss_slab[wl] = s;
  }
}

static int th_cus() {
  static int cus = 0;
  if (cus == 0) {
    int dev = 0, nn = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&nn, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || nn <= 0)
      nn = 256;
    cus = nn;
  }
  return cus;
}

static int th_wg_per_prob(const mml_tower_head_group* g) {
  int w = th_cus() / g->n;
  const int64_t blocks = (g->M + 255) / 256;  // (one 32-row block per wave at least)
  if (w > blocks) w = (int)blocks;
  return w < 1 ? 1 : w;
}

}  // namespace mml

using namespace mml;

extern "C" int mml_tower_head_serves(const mml_tower_head_group* g) {
  if (!g || g->n < 1 || g->n > MML_MAX_HEADS || g->M < 1 || !g->prob || !g->y) return 0;
  for (int t = 0; t < g->n; ++t) {
    const mml_tower_head_desc& d = g->t[t];
    if (!((d.N == 64 && d.K == 128) || (d.N == 64 && d.K == 64))) return 0;  // (128-wide towers spill: not instantiated)
    if (d.N != g->t[0].N || d.K != g->t[0].K) return 0;
    if (!d.A || !d.amax_a || !d.w_planes_fwd || !d.w_planes_bwd || !d.kexp_fwd || !d.kexp_bwd || !d.w || !d.hbias) return 0;
    if (!d.dH || !d.dA || !d.dw || !d.dhbias) return 0;
    if (!aligned16(d.A) || d.lda % 4 != 0 || d.lda < d.K || !aligned16(d.dH) || d.lddh % 4 != 0 || d.lddh < d.N ||
        !aligned16(d.dA) || d.ldda % 4 != 0 || d.ldda < d.K)
      return 0;
    if (!aligned16(d.w_planes_fwd) || d.ldpf % 4 != 0 || d.ldpf < d.K || d.ldpb < d.K) return 0;
    if (d.n_hbias2 < 0 || (d.n_hbias2 > 0 && !d.hbias2)) return 0;
    if (d.mask_col >= 0 && !g->mask) return 0;
  }
  return 1;
}

extern "C" int64_t mml_tower_head_workspace_bytes(const mml_tower_head_group* g) {
  if (!g || g->n < 1 || g->n > MML_MAX_HEADS) return -1;
  return (int64_t)g->n * th_cus() * (g->t[0].N + 2) * 4 + 256;  // dw + dbias rows, then the loss arrays
}

// phase 1: the launch (prob, dH, dA; per-workgroup partial sums of dw / dbias / loss in the workspace); 2: their reduction; 0: both
extern "C" int mml_tower_head_fwd_bwd(const mml_tower_head_group* g, void* workspace, int64_t workspace_bytes, int32_t phase,
                                      mml_stream_t stream) {
  MML_REQUIRE(g != nullptr && phase >= 0 && phase <= 2, "mml_tower_head_fwd_bwd: null group or bad phase");
  if (!mml_tower_head_serves(g)) {
    set_error("mml_tower_head_fwd_bwd: not a group this kernel serves (widths 64 <- 128 / 64 <- 64, fp32, 16-byte "
              "aligned rows, pre-cut planes in both layouts, the input's magnitude)");
    return MML_ERR_UNSUPPORTED;
  }
  MML_REQUIRE(workspace && workspace_bytes >= mml_tower_head_workspace_bytes(g), "mml_tower_head_fwd_bwd: workspace too small");
  const int N = g->t[0].N, K = g->t[0].K;
  ThLaunch L{};
  L.M = g->M; L.prob = g->prob; L.y = g->y; L.mask = g->mask; L.ldprob = g->ldprob; L.ldy = g->ldy; L.ldmask = g->ldmask;
  L.n_prob = g->n;
  L.wg_per_prob = th_wg_per_prob(g);
  float* ws = static_cast<float*>(workspace);
  for (int t = 0; t < g->n; ++t) {
    const mml_tower_head_desc& d = g->t[t];
    ThProblem& P = L.p[t];
    P.A = d.A; P.amaxA = d.amax_a; P.planesF = d.w_planes_fwd; P.planesB = d.w_planes_bwd; P.kexpF = d.kexp_fwd;
    P.kexpB = d.kexp_bwd; P.bias1 = d.bias1; P.w = d.w; P.hbias = d.hbias; P.hbias2 = d.hbias2; P.dH = d.dH; P.dA = d.dA;
    P.amax_dH = d.amax_dH; P.amax_dA = d.amax_dA; P.lda = d.lda; P.ldpf = d.ldpf; P.ldpb = d.ldpb; P.lddh = d.lddh;
    P.ldda = d.ldda; P.n_hbias2 = d.n_hbias2; P.mask_col = d.mask_col; P.head = d.head;
    P.slab = ws + (int64_t)t * L.wg_per_prob * (N + 1);
    P.loss_slab = ws + (int64_t)g->n * L.wg_per_prob * (N + 1) + (int64_t)t * L.wg_per_prob;
  }
  if (phase != 2) {
    const dim3 grid((unsigned)(L.wg_per_prob * L.n_prob)), block(512);
    if (K == 128) MML_LAUNCH((tower_head_kernel<2, 4>), grid, block, 0, to_stream(stream), L);
    else MML_LAUNCH((tower_head_kernel<2, 2>), grid, block, 0, to_stream(stream), L);
    const int rc = check_launch("mml_tower_head_fwd_bwd");
    if (rc != MML_OK || phase == 1) return rc;
  }
  ReduceLaunch R{};
  int64_t start = 0;
  for (int t = 0; t < g->n; ++t) {
    const mml_tower_head_desc& d = g->t[t];
    const float* slab = L.p[t].slab;
    ReduceSeg& w = R.seg[R.n++];
    w.slab = slab; w.out = d.dw; w.n = N; w.sstride = N + 1; w.cols = N; w.ldo = N; w.start = start; w.S = L.wg_per_prob;
    start += N;
    ReduceSeg& b = R.seg[R.n++];
    b.slab = slab + N; b.out = d.dhbias; b.n = 1; b.sstride = N + 1; b.cols = 1; b.ldo = 1; b.start = start; b.S = L.wg_per_prob;
    start += 1;
  }
  if (g->loss) {  // ONE loss over all heads: their partial-sum arrays stand back to back -> one segment of n x wg_per_prob terms
    ReduceSeg& l = R.seg[R.n++];
    l.slab = L.p[0].loss_slab; l.out = g->loss; l.n = 1; l.sstride = 1; l.cols = 1; l.ldo = 1; l.start = start;
    l.S = g->n * L.wg_per_prob;
    start += 1;
  }
  R.total = start;
  return launch_slab_reduce(R, to_stream(stream), "mml_tower_head_fwd_bwd(reduce)");
}
