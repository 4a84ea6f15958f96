// K3 (round 4): weight-stationary streaming GEMM for the layers whose WHOLE weight fits the LDS.
//
// The second expert layer, the towers and their input gradients (reference model/mmoe.py:69-119: expert_dnn / tower_dnn
// called per expert / task on its OWN input; model/utils.py:146-161 is the Linear -> ReLU layer) are streaming problems:
// every problem reads its own [M, K] activations once and writes [M, N] once, K and N <= 256 -- at M = 65 536 about
// 400 MB of traffic against 50 GFLOP of plane products.  gemm_pipe_kernel (gemm.hip) runs them as 128 x 128 tiles of 8
// or 16 k-steps: a tile is mostly its own prologue (pipeline fill, barriers) and epilogue, two of them fit a CU, and
// the launch ends up bound by neither HBM (2.9 TB/s) nor the MFMA pipe (15 %).  Here:
//   * a persistent workgroup (eight waves, two per SIMD) serves ONE problem and keeps that problem's pre-cut weight
//     planes (mml_gemm_planes_cut; <= 128 KiB) in LDS for its whole life, in fragment order: a wave's ds_read_b128 of a
//     32-column x 16-k fragment is 1 KiB of consecutive bytes;
//   * a wave owns 32-row blocks of the batch and never meets another wave again: no barrier, no tile prologue.  Its
//     activations come straight from global memory as MFMA fragments (two 16-byte loads per lane and k-step, four
//     k-steps ahead in a register ring, across block boundaries), are cut into their two fp16 planes in registers and
//     meet the weight fragments from LDS in 3 x NS v_mfma_f32_32x32x16_f16 per k-step; the epilogue (unscale, bias /
//     ReLU + sign mask, or the mask / accumulation of an input gradient) stores 16 bytes per lane (lane = batch row);
//     the other wave of the SIMD computes meanwhile;
//   * up to 256 output columns (eight 32-column sub-tiles of accumulators) per wave: the input gradient of a 256-wide
//     layer reads its rows once.
// Products, their order (W_l A_h, W_h A_l, W_h A_h per 16-k block) and the k order are those of gemm_pipe_kernel<..,
// EMU = 2, BPL>: same bits (tests/test_gemm_ws_gpu.py).
//
// K7 (round 6; PepNet, reference model/pepnet.py:64-78, :139-140: every PPNet layer's input is h (.) 2 sigmoid(gate(..))):
//   * forward, template K7 of MODE 0: the gate network's output layer stores g = act(z) AND prod = g (.) mul from the same
//     row-major turn -- the product makes no pass of its own (it cost a 12 B / element launch);
//   * backward, K7 of MODE 1 ("gate mode" of mml_gemm_grouped_dgrad): the input gradient v of the layer that reads
//     h (.) g is not stored; the turn reads h and g once and stores dH (+)= v g act_h'(h) and dG (+)= v h act_g'(g)
//     (the expressions of gemm_pipe_kernel's gate mode) with one magnitude slot each.
#include "common.hpp"
#include "lds_async.hpp"

#include <stdlib.h>

namespace mml {

using wf32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int WS_W_BYTES = 128 * 1024;              // the weight image
constexpr int WS_BIAS_OFF = WS_W_BYTES;             // <= 256 bias values
constexpr int WS_AMAX_OFF = WS_BIAS_OFF + 1024;     // the workgroup's magnitude word
constexpr int WS_TURN_OFF = WS_AMAX_OFF + 64;       // eight waves x 2 KiB: 32 rows x 64 bytes turned row-major for the stores
constexpr int WS_LDS_BYTES = WS_TURN_OFF + 8 * 2048;
static_assert(WS_LDS_BYTES <= 160 * 1024, "weight-stationary kernel LDS budget");
constexpr int WS_MIN_ROWS = 8192;                   // below: the tile kernel (a persistent grid would idle)
__device__ __forceinline__ float ws_act_bwd(const float y, const int act) {  // (gemm.hip: act_bwd)
  if (act == MML_ACT_RELU) return y > 0.f ? 1.f : 0.f;
  if (act == MML_ACT_SIGMOID) return y * (1.f - y);
  if (act == MML_ACT_SIGMOID2) {
    const float s_ = 0.5f * y;
    return 2.f * s_ * (1.f - s_);
  }
  return 1.f;
}
struct WsProblem {
  const float* A;          // [M, Kred] activations (forward) / output gradients (input gradient)
  const uint32_t* amaxA;   // magnitude slot of A
  const uint32_t* planes;  // pre-cut weight, pitch ldp words
  const int32_t* kexp;     // exponent the planes were cut with
  float* C;                // [M, Nout]
  const float* bias;       // forward: [Nout] or null
  uint32_t* mask;          // forward: sign mask written; input gradient: sign mask read
  uint32_t* amax_out;
  int64_t lda, ldp, ldc, ldmask;
  int32_t layout;          // MML_PLANES_ROWS: planes[out col][k]; MML_PLANES_COLS: planes[k][out col]
  int32_t act;             // forward: MML_ACT_NONE / RELU / SIGMOID / SIGMOID2
  int32_t accumulate;      // input gradient: C +=
  int32_t G;               // groups of D k-steps (Kred = 16 D G)
  // K7: forward: x1 = mul, c2 = prod; gate mode: x1 = h, x2 = g, C = dH (act1 = act_h, accumulate = acc_h), c2 = dG
  const float* x1;
  const float* x2;
  float* c2;
  uint32_t* amax_out2;     // magnitude of what c2 receives
  int64_t ld1, ld2, ldc2;
  int32_t act1, act2, acc2, pad_;
};

struct WsLaunch {
  int32_t M, n_prob, wg_per_prob, pad_;
  WsProblem p[MML_MAX_GROUP];
};
static_assert(sizeof(WsLaunch) <= 4096, "WsLaunch must fit the kernel-argument block");

__device__ __forceinline__ uint32_t ws_amax_load(const uint32_t* p) {
  uint32_t m = 0;
#pragma unroll
  for (int i = 0; i < MML_AMAX_WORDS; ++i) m = p[i] > m ? p[i] : m;
  return m;
}
// (the rule of gemm.hip: |x| 2^k < 2^15 for every |x| <= the slot's value; Inf / NaN: scale 1)
__device__ __forceinline__ int ws_scale_exp(uint32_t bits) {
  const int e = (int)((bits >> 23) & 0xffu);
  if (e == 255) return 0;
  const int k = 141 - e;
  return k > 110 ? 110 : (k < -110 ? -110 : k);
}
__device__ __forceinline__ float ws_pow2(int k) { return __uint_as_float((uint32_t)(127 + k) << 23); }

// MODE 0: forward (bias, ReLU, sign mask out when MASKS); MODE 1: input gradient (sign mask in when MASKS, accumulation)
// K7: forward with the product epilogue / input gradient in gate mode (see the head of this file)
template <int NS, int MODE, bool MASKS, int D, bool K7 = false>
__global__ __launch_bounds__(512, 2) void gemm_ws_kernel(const WsLaunch L) {
  constexpr int NSTOT = NS;
  __shared__ __attribute__((aligned(16))) float lds[WS_LDS_BYTES / 4];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, h = lane >> 5;
  const int pi = (int)blockIdx.x / L.wg_per_prob;
  const int wl = (int)blockIdx.x - pi * L.wg_per_prob;
  const WsProblem& P = L.p[pi];
  const int G = P.G, KB = D * G, M = L.M;  // D: k-steps per group (4, or 5 for reductions of 80, 160, 240)

  // ---- the problem's weight planes -> LDS, fragment order: [k-step][plane][32-column sub-tile][lane] x 16 bytes ----
  {
    uint32_t* const lw = reinterpret_cast<uint32_t*>(lds);
    if (P.layout == MML_PLANES_ROWS) {  // row = output column, 16 words per k-step: chunk c = 2 plane + lane half
      const int total = NSTOT * 32 * KB * 4;
      for (int idx = tid; idx < total; idx += 512) {
        const int c = idx & 3, b = (idx >> 2) % KB, n = idx / (4 * KB);
        const uint4 v = *reinterpret_cast<const uint4*>(P.planes + (int64_t)n * P.ldp + 16 * b + 4 * c);
        *reinterpret_cast<uint4*>(lw + ((((b * 2 + (c >> 1)) * NSTOT + (n >> 5)) * 64 + (c & 1) * 32 + (n & 31)) * 4)) = v;
      }
    } else {                            // row = k, word e = 8 plane + 4 lane half + i of its block of 16
      constexpr int NO = NSTOT * 32;
      const int total = KB * 16 * NO;
      for (int idx = tid; idx < total; idx += 512) {
        const int c = idx % NO, r = idx / NO;
        const int b = r >> 4, e = r & 15;
        lw[(((b * 2 + (e >> 3)) * NSTOT + (c >> 5)) * 64 + ((e >> 2) & 1) * 32 + (c & 31)) * 4 + (e & 3)] =
            P.planes[(int64_t)r * P.ldp + c];
      }
    }
    if (MODE == 0 && tid < NSTOT * 32) lds[WS_BIAS_OFF / 4 + tid] = P.bias ? P.bias[tid] : 0.f;
    if (tid < 2) reinterpret_cast<uint32_t*>(lds)[WS_AMAX_OFF / 4 + tid] = 0u;
  }
  __syncthreads();

  const int kA = __builtin_amdgcn_readfirstlane(ws_scale_exp(ws_amax_load(P.amaxA)));
  const int kB = __builtin_amdgcn_readfirstlane(*P.kexp);
  const float sA = ws_pow2(kA);
  const float inv = ws_pow2(-kA) * ws_pow2(-kB);

  const int nrb = (M + 31) >> 5;
  const int stride = L.wg_per_prob * 8;
  int rb = wl * 8 + wave;
  float am = 0.f, am2 = 0.f;

  auto arow = [&](const int rb_) __attribute__((always_inline)) {
    int row = rb_ * 32 + l31;
    row = row < M ? row : M - 1;
    return P.A + (int64_t)row * P.lda + 4 * h;
  };
  const f32x4_t* const wfrag = reinterpret_cast<const f32x4_t*>(lds) + lane;  // + 64 x (fragment number)
  float* const turn = lds + WS_TURN_OFF / 4 + wave * 512;                       // this wave's row-major turn area

  if (rb < nrb) {
    // The raw fragments of the next four k-steps (plain loads: hipcc places the waits.  It gathers a group's refills behind
    // the group's last MFMA and waits for them at the top of the next group, so a wave overlaps nothing by itself -- the
    // seven other waves of the CU do.  Hand-counted waits with a look-ahead of eight k-steps, refills in bursts of 512
    // bytes per row and twelve waves per CU were all built and measured level with this form: DESIGN 9.)
    float4 r0[D], r1[D];
    const float* ap = arow(rb);
#pragma unroll
    for (int j = 0; j < D; ++j) {
      r0[j] = *reinterpret_cast<const float4*>(ap + 16 * j);
      r1[j] = *reinterpret_cast<const float4*>(ap + 16 * j + 8);
    }
    for (; rb < nrb; rb += stride) {
      const float* const ap_next = arow(rb + stride < nrb ? rb + stride : rb);
      {
        wf32x16 acc[NS];
#pragma unroll
        for (int ni = 0; ni < NS; ++ni)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[ni][r] = 0.f;
        const float* const unit_next = ap_next;
#pragma unroll 1
        for (int g = 0; g < G; ++g) {
          const float* const pf = (g + 1 < G) ? ap + 16 * D * (g + 1) : unit_next;
          const f32x4_t* const wg_ = wfrag + g * 2 * D * NSTOT * 64;
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
              bh[ni] = __builtin_bit_cast(f16x8, wg_[((j * 2 + 0) * NSTOT + ni) * 64]);
              bl[ni] = __builtin_bit_cast(f16x8, wg_[((j * 2 + 1) * NSTOT + ni) * 64]);
            }
#pragma unroll
            for (int ni = 0; ni < NS; ++ni) acc[ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl[ni], Ah, acc[ni], 0, 0, 0);
#pragma unroll
            for (int ni = 0; ni < NS; ++ni) acc[ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[ni], Al, acc[ni], 0, 0, 0);
#pragma unroll
            for (int ni = 0; ni < NS; ++ni) acc[ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[ni], Ah, acc[ni], 0, 0, 0);
          }
        }
        // ---- the block's outputs.  Accumulators: lane = batch row, registers = four runs of four columns (8 g + 4 h + j).
        // Stored like that a wave-instruction writes 32 bytes of each of 32 rows -- partial lines the L2 has to merge: the
        // second expert layer ran 89 us that way.  So every 16-column half of a sub-tile is turned row-major through the
        // wave's 2 KiB of LDS: a lane then moves 16 bytes of a row and an instruction covers 16 rows x 64 contiguous bytes:
        // 81 us (16-byte chunk c of row r sits at chunk c ^ ((r >> 2) & 3): conflict-free both ways; a wave's LDS
        // operations execute in order, so no barrier stands between the two directions).  Measured beside it: whole
        // 128-byte lines through 4 KiB per wave, with the bias read from global memory to make room: 105 us; nontemporal
        // stores 161 (32-byte pieces) / 95 (64) / 96 (128).
        const int row = rb * 32 + l31;
        const bool ok = row < M;
        float* const tw = turn + l31 * 16;                     // write side: row l31 (floats)
        const int tR = lane >> 2, tc = lane & 3;               // read side: rows tR, tR + 16, logical chunk tc
        const int orow = rb * 32 + tR;
        float* const cst = P.C + (int64_t)orow * P.ldc + 4 * tc;
        const int64_t c16 = 16 * P.ldc;
        const bool ok0 = orow < M, ok1 = orow + 16 < M;
        const int lrow0 = ok0 ? orow : M - 1, lrow1 = ok1 ? orow + 16 : M - 1;  // (K7: rows the extra operands are read from)
        auto turn_store = [&](const float4 va, const float4 vb, const int ni, const int half) __attribute__((always_inline)) {
          // va / vb: this lane's runs g = 2 half, 2 half + 1 (columns 16 half + 4 h + {0..3}, + 8)
          *reinterpret_cast<float4*>(tw + 4 * ((0 + h) ^ ((l31 >> 2) & 3))) = va;
          *reinterpret_cast<float4*>(tw + 4 * ((2 + h) ^ ((l31 >> 2) & 3))) = vb;
          asm volatile("" ::: "memory");
          float4 q0 = *reinterpret_cast<const float4*>(turn + tR * 16 + 4 * (tc ^ ((tR >> 2) & 3)));
          float4 q1 = *reinterpret_cast<const float4*>(turn + (tR + 16) * 16 + 4 * (tc ^ (((tR + 16) >> 2) & 3)));
          float* const d0 = cst + ni * 32 + 16 * half;
          if constexpr (K7) {
            // (rows past the batch read the last row: always a legal address, their results are never stored)
            const int cofs = 4 * tc + ni * 32 + 16 * half;
            const int64_t oc2 = (int64_t)orow * P.ldc2 + cofs;
            const float4 m0 = *reinterpret_cast<const float4*>(P.x1 + (int64_t)lrow0 * P.ld1 + cofs);
            const float4 m1 = *reinterpret_cast<const float4*>(P.x1 + (int64_t)lrow1 * P.ld1 + cofs);
            if constexpr (MODE == 0) {  // prod = g (.) mul; g itself is stored below
              const float4 p0 = make_float4(q0.x * m0.x, q0.y * m0.y, q0.z * m0.z, q0.w * m0.w);
              const float4 p1 = make_float4(q1.x * m1.x, q1.y * m1.y, q1.z * m1.z, q1.w * m1.w);
              if (ok0) *reinterpret_cast<float4*>(P.c2 + oc2) = p0;
              if (ok1) *reinterpret_cast<float4*>(P.c2 + oc2 + 16 * P.ldc2) = p1;
              amax_acc(am2, p0);
              amax_acc(am2, p1);
            } else {                    // q = the raw input gradient v of h (.) g: dG = v h act_g'(g), dH = v g act_h'(h)
              const float4 g0 = *reinterpret_cast<const float4*>(P.x2 + (int64_t)lrow0 * P.ld2 + cofs);
              const float4 g1 = *reinterpret_cast<const float4*>(P.x2 + (int64_t)lrow1 * P.ld2 + cofs);
              const int a1 = P.act1, a2 = P.act2;
              float4 e0 = make_float4(q0.x * m0.x, q0.y * m0.y, q0.z * m0.z, q0.w * m0.w);
              float4 e1 = make_float4(q1.x * m1.x, q1.y * m1.y, q1.z * m1.z, q1.w * m1.w);
              if (a2 != MML_ACT_NONE) {
                e0.x *= ws_act_bwd(g0.x, a2); e0.y *= ws_act_bwd(g0.y, a2); e0.z *= ws_act_bwd(g0.z, a2); e0.w *= ws_act_bwd(g0.w, a2);
                e1.x *= ws_act_bwd(g1.x, a2); e1.y *= ws_act_bwd(g1.y, a2); e1.z *= ws_act_bwd(g1.z, a2); e1.w *= ws_act_bwd(g1.w, a2);
              }
              q0.x *= g0.x * ws_act_bwd(m0.x, a1); q0.y *= g0.y * ws_act_bwd(m0.y, a1);
              q0.z *= g0.z * ws_act_bwd(m0.z, a1); q0.w *= g0.w * ws_act_bwd(m0.w, a1);
              q1.x *= g1.x * ws_act_bwd(m1.x, a1); q1.y *= g1.y * ws_act_bwd(m1.y, a1);
              q1.z *= g1.z * ws_act_bwd(m1.z, a1); q1.w *= g1.w * ws_act_bwd(m1.w, a1);
              if (P.acc2) {
                if (ok0) {
                  const float4 o = *reinterpret_cast<const float4*>(P.c2 + oc2);
                  e0.x += o.x; e0.y += o.y; e0.z += o.z; e0.w += o.w;
                }
                if (ok1) {
                  const float4 o = *reinterpret_cast<const float4*>(P.c2 + oc2 + 16 * P.ldc2);
                  e1.x += o.x; e1.y += o.y; e1.z += o.z; e1.w += o.w;
                }
              }
              if (ok0) *reinterpret_cast<float4*>(P.c2 + oc2) = e0;
              if (ok1) *reinterpret_cast<float4*>(P.c2 + oc2 + 16 * P.ldc2) = e1;
              amax_acc(am2, e0);
              amax_acc(am2, e1);
              if (!P.accumulate) {  // (accumulating: the magnitude of the sums, below)
                amax_acc(am, q0);
                amax_acc(am, q1);
              }
            }
          }
          if (MODE == 1 && P.accumulate) {
            if (ok0) {
              const float4 o = *reinterpret_cast<const float4*>(d0);
              q0.x += o.x; q0.y += o.y; q0.z += o.z; q0.w += o.w;
            }
            if (ok1) {
              const float4 o = *reinterpret_cast<const float4*>(d0 + c16);
              q1.x += o.x; q1.y += o.y; q1.z += o.z; q1.w += o.w;
            }
            amax_acc(am, q0);
            amax_acc(am, q1);
          }
          if (ok0) *reinterpret_cast<float4*>(d0) = q0;
          if (ok1) *reinterpret_cast<float4*>(d0 + c16) = q1;
        };
        if constexpr (MODE == 0) {
          const int act = P.act;  // (uniform)
#pragma unroll
          for (int ni = 0; ni < NS; ++ni) {
            uint32_t bits = 0u;
            float4 vv[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              const float4 b4 = *reinterpret_cast<const float4*>(lds + WS_BIAS_OFF / 4 + 4 * h + ni * 32 + 8 * g);
              float4 v;
              v.x = acc[ni][4 * g] * inv + b4.x;
              v.y = acc[ni][4 * g + 1] * inv + b4.y;
              v.z = acc[ni][4 * g + 2] * inv + b4.z;
              v.w = acc[ni][4 * g + 3] * inv + b4.w;
              if (act == MML_ACT_RELU) {
                v.x = __builtin_fmaxf(v.x, 0.f);
                v.y = __builtin_fmaxf(v.y, 0.f);
                v.z = __builtin_fmaxf(v.z, 0.f);
                v.w = __builtin_fmaxf(v.w, 0.f);
              } else if (act != MML_ACT_NONE) {  // sigmoid, or PepNet's 2 sigmoid: the tile kernel's expressions
                const float two = act == MML_ACT_SIGMOID2 ? 2.f : 1.f;
                v.x = two / (1.f + __expf(-v.x));
                v.y = two / (1.f + __expf(-v.y));
                v.z = two / (1.f + __expf(-v.z));
                v.w = two / (1.f + __expf(-v.w));
              }
              amax_acc(am, v);
              if (MASKS)
                bits |= ((v.x > 0.f ? 1u : 0u) | (v.y > 0.f ? 2u : 0u) | (v.z > 0.f ? 4u : 0u) | (v.w > 0.f ? 8u : 0u)) << (8 * g);
              vv[g] = v;
            }
            turn_store(vv[0], vv[1], ni, 0);
            turn_store(vv[2], vv[3], ni, 1);
            if (MASKS) {  // the two lanes of a row hold the interleaved halves of its 32-column word
              const uint32_t mine = bits << (4 * h);
              const auto sw = __builtin_amdgcn_permlane32_swap(mine, mine, false, false);
              const uint32_t word = sw[0] | sw[1];
              if (ok && h == 0) P.mask[(int64_t)row * P.ldmask + ni] = word;
            }
          }
        } else {
          const bool accumulate = P.accumulate != 0;
          uint32_t mw[NS];
          if (MASKS) {
#pragma unroll
            for (int ni = 0; ni < NS; ++ni) mw[ni] = ok ? P.mask[(int64_t)row * P.ldmask + ni] : 0u;
          }
#pragma unroll
          for (int ni = 0; ni < NS; ++ni) {
            float4 vv[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              float4 v;
              v.x = acc[ni][4 * g] * inv;
              v.y = acc[ni][4 * g + 1] * inv;
              v.z = acc[ni][4 * g + 2] * inv;
              v.w = acc[ni][4 * g + 3] * inv;
              if (MASKS) {
                const uint32_t nib = mw[ni] >> (8 * g + 4 * h);
                if (!(nib & 1u)) v.x = 0.f;
                if (!(nib & 2u)) v.y = 0.f;
                if (!(nib & 4u)) v.z = 0.f;
                if (!(nib & 8u)) v.w = 0.f;
              }
              if (!accumulate && !K7) amax_acc(am, v);  // (accumulating / gate mode: the magnitude of what is stored, taken behind the turn)
              vv[g] = v;
            }
            turn_store(vv[0], vv[1], ni, 0);
            turn_store(vv[2], vv[3], ni, 1);
          }
        }
      }
      ap = ap_next;
    }
  }

  // the workgroup's largest |output| -> the problem's slot (one atomic per workgroup)
  if (P.amax_out) {  // (uniform)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) am = fmaxf(am, __shfl_xor(am, o, 64));
    uint32_t* const word = reinterpret_cast<uint32_t*>(lds) + WS_AMAX_OFF / 4;
    if (lane == 0) atomicMax(word, __float_as_uint(am));
    __syncthreads();
    if (tid == 0 && *word) atomicMax(P.amax_out + (blockIdx.x & (MML_AMAX_WORDS - 1)), *word);
  }
  if constexpr (K7) {
    if (P.amax_out2) {  // (uniform) the second output's magnitude: prod, or dG
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) am2 = fmaxf(am2, __shfl_xor(am2, o, 64));
      uint32_t* const word2 = reinterpret_cast<uint32_t*>(lds) + WS_AMAX_OFF / 4 + 1;
      if (lane == 0) atomicMax(word2, __float_as_uint(am2));
      __syncthreads();
      if (tid == 0 && *word2) atomicMax(P.amax_out2 + (blockIdx.x & (MML_AMAX_WORDS - 1)), *word2);
    }
  }
}

static int g_ws_on = -1;
static int ws_enabled() {
  if (g_ws_on < 0) {
    const char* e = getenv("MMLREC_GEMM_WS");
    g_ws_on = (e && e[0] == '0') ? 0 : 1;
  }
  return g_ws_on;
}
static int ws_cus() {
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

// symbol of the calling thread's most recent weight-stationary launch, as rocprofv3 prints it (mml_gemm_last_kernel)
static thread_local char g_ws_last[64] = "gemm_ws_kernel";
const char* mml_gemm_ws_last_symbol() { return g_ws_last; }

template <int MODE>
static int ws_launch(const WsLaunch& L, const int nout, const bool masks, const int dgroup, const bool k7, hipStream_t st) {
  const dim3 g((unsigned)(L.wg_per_prob * L.n_prob)), b(512);
  snprintf(g_ws_last, sizeof(g_ws_last), "gemm_ws_kernel<%d, %d, %s, %d, %s>", nout / 32, MODE, (masks && !k7) ? "true" : "false",
           dgroup, k7 ? "true" : "false");
#define WS_GO2(NS_, D_)                                                                 \
  do {                                                                                  \
    if (k7) MML_LAUNCH((gemm_ws_kernel<NS_, MODE, false, D_, true>), g, b, 0, st, L);   \
    else if (masks) MML_LAUNCH((gemm_ws_kernel<NS_, MODE, true, D_>), g, b, 0, st, L);  \
    else MML_LAUNCH((gemm_ws_kernel<NS_, MODE, false, D_>), g, b, 0, st, L);            \
  } while (0)
#define WS_GO8()                                                                              \
  do {                                                                                        \
    if (dgroup == 5) {                                                                        \
      if (masks) MML_LAUNCH((gemm_ws_kernel<8, MODE, true, 5>), g, b, 0, st, L);              \
      else MML_LAUNCH((gemm_ws_kernel<8, MODE, false, 5>), g, b, 0, st, L);                   \
    } else {                                                                                  \
      if (masks) MML_LAUNCH((gemm_ws_kernel<8, MODE, true, 4>), g, b, 0, st, L);              \
      else MML_LAUNCH((gemm_ws_kernel<8, MODE, false, 4>), g, b, 0, st, L);                   \
    }                                                                                         \
  } while (0)
#define WS_GO(NS_)                 \
  do {                             \
    if (dgroup == 5) WS_GO2(NS_, 5); \
    else WS_GO2(NS_, 4);           \
  } while (0)
  if (nout == 64) WS_GO(2);
  else if (nout == 128) WS_GO(4);
  else if (k7) return MML_ERR_UNSUPPORTED;  // (never reached: ws_fwd_ok / ws_dgrad_ok refuse 256-wide K7 problems --
                                            //  eight sub-tiles of accumulators plus the factors' registers would spill)
  else WS_GO8();  // (256 output columns of an input gradient: eight sub-tiles at once -- swept in two passes of four the rows
                  //  were read twice from HBM, PMC 300 MB against 138 algorithmic: 106 -> 93 us)
#undef WS_GO
#undef WS_GO8
#undef WS_GO2
  return check_launch(MODE == 0 ? "mml_gemm_grouped_fwd(ws)" : "mml_gemm_grouped_dgrad(ws)");
}

// k-steps per group for a reduction of `kred` values: 4 (multiples of 64), 5 (other multiples of 80), 0 = not served
static int ws_dgroup(const int kred) { return kred <= 0 ? 0 : (kred % 64 == 0 ? 4 : (kred % 80 == 0 ? 5 : 0)); }

}  // namespace mml

using namespace mml;

extern "C" int mml_gemm_set_ws(int32_t on) {
  g_ws_on = on ? 1 : 0;
  return MML_OK;
}

// Both return MML_OK when the weight-stationary kernel took the launch and MML_ERR_UNSUPPORTED (no error text) when the
// launch is not one it serves: the caller then runs the tile kernel.  A launch whose problems differ in shape (PepNet's
// and PLE's layers: 128- and 64-wide siblings in one call) is served as one kernel launch per (output width, sign masks)
// class -- all of its problems must qualify.
static bool ws_fwd_ok(const mml_gemm_fwd_desc& q, const mml_gemm_fwd_desc& d0) {
  if (q.M != d0.M || ws_dgroup(q.K) == 0) return false;
  if (q.N != 256 && q.N != 128 && q.N != 64) return false;  // (the instantiated output widths)
  if ((int64_t)q.N * q.K * 4 > WS_W_BYTES) return false;
  if (!q.A || !q.C || !q.w_planes || !q.w_kexp || !q.amax_a) return false;
  if (q.mul || q.prod) {  // K7: prod = act(..) (.) mul from the same turn
    if (!q.mul || !q.prod || q.ldmul < q.N || q.ldprod < q.N) return false;
    if (!aligned16(q.mul) || q.ldmul % 4 != 0 || !aligned16(q.prod) || q.ldprod % 4 != 0) return false;
    if (q.act == MML_ACT_RELU && q.relu_mask) return false;  // (no instantiation with sign masks: gates end in a sigmoid)
    if (q.N == 256) return false;                            // (nor with eight sub-tiles: registers)
  }
  if (q.act != MML_ACT_RELU && q.act != MML_ACT_NONE && q.act != MML_ACT_SIGMOID && q.act != MML_ACT_SIGMOID2) return false;
  if (!aligned16(q.A) || q.lda % 4 != 0 || !aligned16(q.C) || q.ldc % 4 != 0) return false;
  if (!q.w_kn && (!aligned16(q.w_planes) || q.ldw % 4 != 0)) return false;
  if (q.act == MML_ACT_RELU && q.relu_mask && q.ldmask * 32 < q.N) return false;
  return true;
}

int mml_gemm_ws_try_fwd(const mml_gemm_fwd_desc* d, int32_t n, hipStream_t st) {
  if (!ws_enabled() || n < 1 || n > MML_MAX_GROUP) return MML_ERR_UNSUPPORTED;
  const mml_gemm_fwd_desc& d0 = d[0];
  if (d0.M < WS_MIN_ROWS) return MML_ERR_UNSUPPORTED;
  for (int i = 0; i < n; ++i)
    if (!ws_fwd_ok(d[i], d0)) return MML_ERR_UNSUPPORTED;
  if (n > ws_cus()) return MML_ERR_UNSUPPORTED;  // (decided before the first launch, never between two of them)
  bool done[MML_MAX_GROUP] = {};
  for (int i = 0; i < n; ++i) {
    if (done[i]) continue;
    const bool mi = d[i].act == MML_ACT_RELU && d[i].relu_mask != nullptr;
    const bool ki = d[i].mul != nullptr;
    WsLaunch L{};
    for (int j = i; j < n; ++j) {
      const mml_gemm_fwd_desc& q = d[j];
      const bool m = q.act == MML_ACT_RELU && q.relu_mask != nullptr;
      if (done[j] || q.N != d[i].N || m != mi || (q.mul != nullptr) != ki || ws_dgroup(q.K) != ws_dgroup(d[i].K)) continue;
      done[j] = true;
      WsProblem& P = L.p[L.n_prob++];
      P.A = q.A;
      P.amaxA = q.amax_a;
      P.planes = q.w_planes;
      P.kexp = q.w_kexp;
      P.C = q.C;
      P.bias = q.bias;
      P.mask = m ? q.relu_mask : nullptr;
      P.amax_out = q.amax_out;
      P.lda = q.lda;
      P.ldp = q.ldw;
      P.ldc = q.ldc;
      P.ldmask = q.ldmask;
      P.layout = q.w_kn ? MML_PLANES_COLS : MML_PLANES_ROWS;  // ([K, N]: the reduction runs down the rows)
      P.act = q.act;
      P.G = q.K / (16 * ws_dgroup(q.K));
      if (ki) {
        P.x1 = q.mul; P.ld1 = q.ldmul; P.c2 = q.prod; P.ldc2 = q.ldprod; P.amax_out2 = q.amax_prod;
      }
    }
    L.M = d0.M;
    L.wg_per_prob = ws_cus() / L.n_prob;
    const int rc = ws_launch<0>(L, d[i].N, mi, ws_dgroup(d[i].K), ki, st);
    if (rc != MML_OK) return rc;
  }
  return MML_OK;
}

static bool ws_dgrad_ok(const mml_gemm_dgrad_desc& q, const mml_gemm_dgrad_desc& d0) {
  if (q.n_src != 1 || q.M != d0.M) return false;
  if (q.gate_h) {  // K7 gate mode: dH / dG instead of dA
    if (!q.gate_g || !q.d_h || !q.d_g || q.ld_h < q.K || q.ld_g < q.K || q.ld_dh < q.K || q.ld_dg < q.K) return false;
    if (!aligned16(q.gate_h) || !aligned16(q.gate_g) || !aligned16(q.d_h) || !aligned16(q.d_g) || q.ld_h % 4 != 0 ||
        q.ld_g % 4 != 0 || q.ld_dh % 4 != 0 || q.ld_dg % 4 != 0)
      return false;
  } else if (!q.dA) {
    return false;
  }
  const int32_t kred = q.N[0];
  if (ws_dgroup(kred) == 0) return false;
  if (q.K != 64 && q.K != 128 && q.K != 256) return false;  // (the instantiated output widths)
  if ((int64_t)q.K * kred * 4 > WS_W_BYTES) return false;
  if (!q.dC[0] || !q.w_planes[0] || !q.w_kexp[0] || !q.amax_dc[0]) return false;
  const bool m = q.act == MML_ACT_RELU && q.relu_mask != nullptr;
  if (q.gate_h) {
    if (m || q.Y || q.K == 256) return false;  // (gate mode: no derivative of the product itself; no 256-wide instantiation)
    if (!aligned16(q.dC[0]) || q.lddc[0] % 4 != 0) return false;
    if (q.w_kn[0] && (!aligned16(q.w_planes[0]) || q.ldw[0] % 4 != 0)) return false;
    return true;
  }
  if (q.act != MML_ACT_NONE && !m) return false;  // (derivatives from the stored outputs: the tile kernel)
  if (m && q.ldmask * 32 < q.K) return false;
  if (!aligned16(q.dC[0]) || q.lddc[0] % 4 != 0 || !aligned16(q.dA) || q.ldda % 4 != 0) return false;
  if (q.w_kn[0] && (!aligned16(q.w_planes[0]) || q.ldw[0] % 4 != 0)) return false;
  return true;
}

int mml_gemm_ws_try_dgrad(const mml_gemm_dgrad_desc* d, int32_t n, hipStream_t st) {
  if (!ws_enabled() || n < 1 || n > MML_MAX_GROUP) return MML_ERR_UNSUPPORTED;
  const mml_gemm_dgrad_desc& d0 = d[0];
  if (d0.M < WS_MIN_ROWS) return MML_ERR_UNSUPPORTED;
  for (int i = 0; i < n; ++i)
    if (!ws_dgrad_ok(d[i], d0)) return MML_ERR_UNSUPPORTED;
  // Two problems of one call that write the same dA (an overwriting and an accumulating one) need an ORDER: the
  // problems of a class run in one kernel launch on disjoint workgroup ranges, i.e. concurrently, and classes are
  // launched in the order of their first members -- neither keeps the call's order.  Left to the tile kernel (the
  // engine never builds such a call: chunks of one input go into different launches).
  for (int i = 0; i < n; ++i)
    for (int j = i + 1; j < n; ++j) {
      const float* ti[2] = {d[i].gate_h ? d[i].d_h : d[i].dA, d[i].gate_h ? d[i].d_g : nullptr};
      const float* tj[2] = {d[j].gate_h ? d[j].d_h : d[j].dA, d[j].gate_h ? d[j].d_g : nullptr};
      for (int a = 0; a < 2; ++a)
        for (int b = 0; b < 2; ++b)
          if (ti[a] && ti[a] == tj[b]) return MML_ERR_UNSUPPORTED;
    }
  if (n > ws_cus()) return MML_ERR_UNSUPPORTED;  // (every class gets >= 1 workgroup per problem: decided BEFORE the first launch)
  bool done[MML_MAX_GROUP] = {};
  for (int i = 0; i < n; ++i) {
    if (done[i]) continue;
    const bool mi = d[i].act == MML_ACT_RELU && d[i].relu_mask != nullptr;
    const bool ki = d[i].gate_h != nullptr;
    WsLaunch L{};
    for (int j = i; j < n; ++j) {
      const mml_gemm_dgrad_desc& q = d[j];
      const bool m = q.act == MML_ACT_RELU && q.relu_mask != nullptr;
      if (done[j] || q.K != d[i].K || m != mi || (q.gate_h != nullptr) != ki || ws_dgroup(q.N[0]) != ws_dgroup(d[i].N[0]))
        continue;
      done[j] = true;
      WsProblem& P = L.p[L.n_prob++];
      P.A = q.dC[0];
      P.amaxA = q.amax_dc[0];
      P.planes = q.w_planes[0];
      P.kexp = q.w_kexp[0];
      P.C = q.dA;
      P.mask = m ? const_cast<uint32_t*>(q.relu_mask) : nullptr;
      P.amax_out = q.amax_out;
      P.lda = q.lddc[0];
      P.ldp = q.ldw[0];
      P.ldc = q.ldda;
      P.ldmask = q.ldmask;
      P.layout = q.w_kn[0] ? MML_PLANES_ROWS : MML_PLANES_COLS;  // ([N, K]: the reduction runs down the rows)
      P.accumulate = q.accumulate;
      P.G = q.N[0] / (16 * ws_dgroup(q.N[0]));
      if (ki) {
        P.C = q.d_h; P.ldc = q.ld_dh; P.accumulate = q.acc_h; P.amax_out = q.amax_dh; P.mask = nullptr;
        P.x1 = q.gate_h; P.ld1 = q.ld_h; P.act1 = q.act_h;
        P.x2 = q.gate_g; P.ld2 = q.ld_g; P.act2 = q.act_g;
        P.c2 = q.d_g; P.ldc2 = q.ld_dg; P.acc2 = q.acc_g; P.amax_out2 = q.amax_dg;
      }
    }
    L.M = d0.M;
    L.wg_per_prob = ws_cus() / L.n_prob;  // (>= 1: n <= ws_cus() was checked before the first launch)
    const int rc = ws_launch<1>(L, d[i].K, mi, ws_dgroup(d[i].N[0]), ki, st);
    if (rc != MML_OK) return rc;
  }
  return MML_OK;
}
