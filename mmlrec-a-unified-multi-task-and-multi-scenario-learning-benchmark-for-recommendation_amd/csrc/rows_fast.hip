// Fast paths of the row kernels (K4 gate, K5 head) for the common aligned case: every feature axis (H, Gd) is a
// multiple of 4 floats with 16-byte aligned rows and at most 256 wide.
//
// Layout of the work: a wavefront is split into 64/LPS lane groups, ONE SAMPLE PER GROUP, and every lane owns one
// 16-byte column slot of each row (LPS = 16, 32 or 64 lanes chosen so that one float4 per lane covers the widest
// row).  All global traffic is therefore dwordx4 per lane and contiguous per sample; dot products finish with a
// log2(LPS)-step butterfly inside the group.  Parameter gradients accumulate in registers over all samples a lane
// group visits and are combined once per workgroup in a fixed order (wave shuffles -> LDS -> slab), so results are
// bitwise reproducible.  Semantics are those of the generic kernels in gate_head.hip (which remain the fallback).
#include "common.hpp"
#include "reduce.hpp"
#include "rows_fast.hpp"

#include <stdlib.h>

namespace mml {

// Sum over the LPS lanes of a lane group, result in every lane.  Inside a row of 16 lanes the exchange is a DPP modifier
// (quad permutes, then the two mirrors: VALU only); only the steps across rows go through the LDS crossbar
// (ds_bpermute).  As five __shfl_xor per sum the PLE gate backward issued 384 LDS instructions and 2 055 VALU
// instructions per pair of samples and was bound by exactly those (PMC: DESIGN 10.3).
__device__ __forceinline__ float dpp_add(float v, const int ctrl_sel) {
  // (ctrl as a switch on a compile-time constant: the builtin wants an immediate)
  const int x = __builtin_bit_cast(int, v);
  int y = x;
  if (ctrl_sel == 0) y = __builtin_amdgcn_update_dpp(x, x, 0xB1, 0xf, 0xf, false);        // quad_perm [1,0,3,2]
  else if (ctrl_sel == 1) y = __builtin_amdgcn_update_dpp(x, x, 0x4E, 0xf, 0xf, false);   // quad_perm [2,3,0,1]
  else if (ctrl_sel == 2) y = __builtin_amdgcn_update_dpp(x, x, 0x141, 0xf, 0xf, false);  // row_half_mirror
  else y = __builtin_amdgcn_update_dpp(x, x, 0x140, 0xf, 0xf, false);                     // row_mirror
  return v + __builtin_bit_cast(float, y);
}
template <int LPS>
__device__ __forceinline__ float group_sum(float v) {
  if (LPS >= 2) v = dpp_add(v, 0);
  if (LPS >= 4) v = dpp_add(v, 1);
  if (LPS >= 8) v = dpp_add(v, 2);
  if (LPS >= 16) v = dpp_add(v, 3);
#pragma unroll
  for (int off = 16; off < LPS; off <<= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// sum over the 64/LPS lane groups of a wave (same sub-lane in every group)
template <int LPS>
__device__ __forceinline__ float cross_group_sum(float v) {
#pragma unroll
  for (int off = LPS; off < 64; off <<= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// Round 6: EIGHT sums over a lane group at once (MMoE's 4 experts x 2 gates: the logits of the forward, the dmix . E dots
// of the backward).  Eight butterflies of log2(LPS) exchange-adds each were 48 DPP / LDS-crossbar steps per sample; here every
// exchange also HALVES what a lane carries -- the partner keeps the other half -- so the first three steps (all DPP) take
// 4 + 2 + 1 adds and leave lane l with the block-of-eight partial sum of value (l & 7); the remaining steps add ONE value
// across the 8-lane blocks of the group (row_ror:8, then the crossbar for 32 / 64 lanes).  Result: the total of value
// (sub & 7) in every lane (each total lives in LPS / 8 lanes).  Partners and the bit that decides which half a lane keeps:
// row_half_mirror (lane ^ 7) by bit 2, quad_perm [2,3,0,1] (lane ^ 2) by bit 1, quad_perm [1,0,3,2] (lane ^ 1) by bit 0.
template <int CTRL>
__device__ __forceinline__ float dpp_get(float v) {
  const int x = __builtin_bit_cast(int, v);
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(x, x, CTRL, 0xf, 0xf, false));
}
template <int LPS>
__device__ __forceinline__ float pack_sum8(const float (&v)[8], const int lane) {
  const bool b2 = (lane & 4) != 0, b1 = (lane & 2) != 0, b0 = (lane & 1) != 0;
  float a[4], c[2];
#pragma unroll
  for (int j = 0; j < 4; ++j) a[j] = (b2 ? v[4 + j] : v[j]) + dpp_get<0x141>(b2 ? v[j] : v[4 + j]);
#pragma unroll
  for (int j = 0; j < 2; ++j) c[j] = (b1 ? a[2 + j] : a[j]) + dpp_get<0x4E>(b1 ? a[j] : a[2 + j]);
  float r = (b0 ? c[1] : c[0]) + dpp_get<0xB1>(b0 ? c[0] : c[1]);
  if (LPS >= 16) r += dpp_get<0x128>(r);  // row_ror:8 = lane ^ 8 inside a row of 16
#pragma unroll
  for (int off = 16; off < LPS; off <<= 1) r += __shfl_xor(r, off, 64);
  return r;
}
__device__ __forceinline__ float quad_max(float v) {
  v = fmaxf(v, dpp_get<0xB1>(v));
  return fmaxf(v, dpp_get<0x4E>(v));
}
__device__ __forceinline__ float quad_sum(float v) {
  v += dpp_get<0xB1>(v);
  return v + dpp_get<0x4E>(v);
}
// (lab knob MMLREC_GATE_PACK=0: the eight separate butterflies of round 5)
static int gate_pack_on() {
  static int on = -1;
  if (on < 0) {
    const char* e = getenv("MMLREC_GATE_PACK");
    on = (e && atoi(e) == 0) ? 0 : 1;
  }
  return on;
}

__device__ __forceinline__ float dot4(const float4& a, const float4& b) {
  return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w;
}
__device__ __forceinline__ float4 mul4(const float4& a, const float4& b) {
  return make_float4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w);
}
__device__ __forceinline__ float act_bwd_rows(const float y, const int act) {  // (gemm.hip: act_bwd)
  if (act == MML_ACT_RELU) return y > 0.f ? 1.f : 0.f;
  if (act == MML_ACT_SIGMOID) return y * (1.f - y);
  if (act == MML_ACT_SIGMOID2) {
    const float s_ = 0.5f * y;
    return 2.f * s_ * (1.f - s_);
  }
  return 1.f;
}
__device__ __forceinline__ void fma4(float4& acc, float s, const float4& v) {
  acc.x += s * v.x; acc.y += s * v.y; acc.z += s * v.z; acc.w += s * v.w;
}
__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, const float4& v) { *reinterpret_cast<float4*>(p) = v; }
// bf16-storage path: the expert outputs may ARRIVE as bf16 (mml_gate_group.out_bf16 & MML_GATE_E_BF16): four values = 8
// bytes at element offset `off`, widened exactly (a bf16 is the upper half of an fp32)
__device__ __forceinline__ float4 ld4i(const float* base, int64_t off, bool as16) {
  if (as16) {
    const uint2 q = *reinterpret_cast<const uint2*>(reinterpret_cast<const uint16_t*>(base) + off);
    return make_float4(__uint_as_float(q.x << 16), __uint_as_float(q.x & 0xffff0000u), __uint_as_float(q.y << 16),
                       __uint_as_float(q.y & 0xffff0000u));
  }
  return *reinterpret_cast<const float4*>(base + off);
}
// bf16-storage path (include/mmlrec.h, K3'): a tensor only GEMMs read is written as four bf16 values (8 bytes, round to
// nearest even) at ELEMENT offset `off` of the same base pointer seen as a bf16 buffer; `as16` is launch-uniform
__device__ __forceinline__ void st4o(float* base, int64_t off, const float4& v, bool as16) {
  if (as16) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    typedef __bf16 b2 __attribute__((ext_vector_type(2)));
    const f2 lo = {v.x, v.y}, hi = {v.z, v.w};
    *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(base) + off) =
        make_uint2(__builtin_bit_cast(uint32_t, __builtin_convertvector(lo, b2)),
                   __builtin_bit_cast(uint32_t, __builtin_convertvector(hi, b2)));
  } else {
    *reinterpret_cast<float4*>(base + off) = v;
  }
}

constexpr int FB = 256;          // threads per workgroup
constexpr int FW = FB / 64;      // waves per workgroup

// ------------------------------------------------------------------------------------------------ gate forward
template <int LPS, int NE>
__global__ __launch_bounds__(FB) void gate_fwd_fast_kernel(const mml_gate_group g, const GateFastAux aux) {
  extern __shared__ __attribute__((aligned(16))) float smem[];  // gate weights, all gates back to back
  for (int i = threadIdx.x; i < aux.wg_total; i += FB) {
    int gi = 0;
    while (gi + 1 < g.n_gates && i >= aux.wg_off[gi + 1]) ++gi;
    smem[i] = g.gate[gi].Wg[i - aux.wg_off[gi]];
  }
  __syncthreads();
  constexpr int SPW = 64 / LPS;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sub = lane % LPS, grp = lane / LPS;
  const int64_t stride = (int64_t)gridDim.x * FW * SPW;
  const int64_t iters = (g.B + stride - 1) / stride;
  const bool hcol = 4 * sub < g.H;
  float am_mix = 0.f;
  for (int64_t it = 0; it < iters; ++it) {
    int64_t b = it * stride + ((int64_t)blockIdx.x * FW + wave) * SPW + grp;
    const bool valid = b < g.B;
    if (!valid) b = g.B - 1;  // keep every lane in the shuffles; stores are predicated
    for (int gi = 0; gi < g.n_gates; ++gi) {
      const mml_gate_desc& d = g.gate[gi];
      const float* W = smem + aux.wg_off[gi];
      float p[NE];
      const bool gcol = 4 * sub < d.Gd;
      float4 Gv = gcol ? ld4(d.G + b * d.ldg + 4 * sub) : make_float4(0, 0, 0, 0);
      // the expert rows of this gate, requested before the logits / softmax arithmetic that hides their latency:
      // straight-line loads with a clamped expert slot (a load inside `if (e < ne)` sits in its own basic block and
      // is waited for one at a time)
      float4 Ev[NE];
#pragma unroll
      for (int e = 0; e < NE; ++e) {
        const int x = d.expert[e < d.ne ? e : d.ne - 1];
        Ev[e] = hcol ? ld4i(g.E[x], b * g.lde[x] + 4 * sub, (g.out_bf16 & MML_GATE_E_BF16) != 0) : make_float4(0, 0, 0, 0);
      }
#pragma unroll
      for (int e = 0; e < NE; ++e) {
        p[e] = -INFINITY;
        if (e < d.ne) {
          float part = gcol ? dot4(Gv, ld4(W + e * d.Gd + 4 * sub)) : 0.f;
          p[e] = group_sum<LPS>(part);
        }
      }
      float m = p[0];
#pragma unroll
      for (int e = 1; e < NE; ++e) m = fmaxf(m, p[e]);
      float den = 0.f;
#pragma unroll
      for (int e = 0; e < NE; ++e) {
        p[e] = (e < d.ne) ? expf(p[e] - m) : 0.f;
        den += p[e];
      }
      const float inv = 1.f / den;
      float mine = 0.f;
#pragma unroll
      for (int e = 0; e < NE; ++e) {
        p[e] *= inv;
        if (sub == e) mine = p[e];
      }
      if (valid && sub < d.ne) d.P[b * d.ldp + sub] = mine;
      if (hcol) {
        float4 acc = make_float4(0, 0, 0, 0);
#pragma unroll
        for (int e = 0; e < NE; ++e) fma4(acc, p[e], Ev[e]);  // p[e] == 0 beyond the gate's expert count
        if (valid) {
          st4o(d.mix, b * d.ldmix + 4 * sub, acc, (g.out_bf16 & MML_GATE_MIX_BF16) != 0);
          amax_acc(am_mix, acc);
        }
      }
    }
  }
  amax_flush(am_mix, g.amax_mix);
}

// Load-once forward (at most NE experts in the group, any membership): the softmax of every gate first, its
// probabilities scattered BY EXPERT INDEX into a per-lane-group LDS row, then every expert row is read once and feeds all
// gates' mixtures.  (The per-gate kernel above re-reads shared experts for every gate that mixes them: 2x the expert
// traffic for MMoE, 2.25x for a PLE level -- and the second read misses L2: PMC 296 MB read vs 167 MB algorithmic.)
// HV = 2 (round 6): bf16 expert rows of up to 256 columns on 32-lane groups, EIGHT columns per lane -- one 16-byte load per
// expert row and lane instead of an 8-byte one (the vector-memory path serves 8-byte accesses at 0.54-0.70 of the 16-byte
// rate) and two samples per wave and trip instead of one; the gate inputs (<= 128 columns) keep four columns per lane.
template <int LPS, int NE, int NG, bool PACK = false, int HV = 1>
__global__ __launch_bounds__(FB) void gate_fwd_once_kernel(const mml_gate_group g, const GateFastAux aux) {
  static_assert(!PACK || (NE == 4 && NG == 2), "the packed sums carry eight values: 4 experts x 2 gates");
  static_assert(HV == 1 || HV == 2, "one or two 16-byte pieces of a row per lane");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int SPW = 64 / LPS;
  float* coef_all = smem + aux.wg_total;  // [FW*SPW][NG*MML_MAX_EXPERTS]
  int* emap = reinterpret_cast<int*>(coef_all + FW * SPW * NG * MML_MAX_EXPERTS);  // [NG][NE] expert of gate slot e
  for (int i = threadIdx.x; i < aux.wg_total; i += FB) {
    int gi = 0;
    while (gi + 1 < g.n_gates && i >= aux.wg_off[gi + 1]) ++gi;
    smem[i] = g.gate[gi].Wg[i - aux.wg_off[gi]];
  }
  for (int i = threadIdx.x; i < NG * NE; i += FB) {
    const int gi = i / NE, e = i - gi * NE;
    emap[i] = (gi < g.n_gates && e < g.gate[gi].ne) ? g.gate[gi].expert[e] : 0;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sub = lane % LPS, grp = lane / LPS;
  float* coef = coef_all + (wave * SPW + grp) * NG * MML_MAX_EXPERTS;
  const int64_t stride = (int64_t)gridDim.x * FW * SPW;
  const int64_t iters = (g.B + stride - 1) / stride;
  const bool hcol = 4 * HV * sub < g.H;
  float am_mix = 0.f;
  // The rows of trip it + 1 are requested before trip it is worked on (round 5): a trip is a chain of a global load, two
  // LDS round trips, the softmax and the stores -- 11 us per trip at B = 65 536 with six workgroups per CU, and a wave had
  // nothing in flight while it worked through it.
  float4 Evn[NE][HV], Gvn[NG];
  auto request = [&](const int64_t it) __attribute__((always_inline)) {
    int64_t b = it * stride + ((int64_t)blockIdx.x * FW + wave) * SPW + grp;
    if (b >= g.B) b = g.B - 1;
#pragma unroll
    for (int x = 0; x < NE; ++x) {
      if constexpr (HV == 2) {
        if (g.out_bf16 & MML_GATE_E_BF16) {  // (launch-uniform) bf16 rows: eight values = one 16-byte load, widened exactly
          uint4 q = make_uint4(0, 0, 0, 0);
          if (hcol && x < g.n_experts) q = *reinterpret_cast<const uint4*>(reinterpret_cast<const uint16_t*>(g.E[x]) + b * g.lde[x] + 8 * sub);
          Evn[x][0] = make_float4(__uint_as_float(q.x << 16), __uint_as_float(q.x & 0xffff0000u), __uint_as_float(q.y << 16), __uint_as_float(q.y & 0xffff0000u));
          Evn[x][1] = make_float4(__uint_as_float(q.z << 16), __uint_as_float(q.z & 0xffff0000u), __uint_as_float(q.w << 16), __uint_as_float(q.w & 0xffff0000u));
        } else {                             // fp32 rows: two 16-byte loads
          const bool on = hcol && x < g.n_experts;
          Evn[x][0] = on ? ld4(g.E[x] + b * g.lde[x] + 8 * sub) : make_float4(0, 0, 0, 0);
          Evn[x][1] = on ? ld4(g.E[x] + b * g.lde[x] + 8 * sub + 4) : make_float4(0, 0, 0, 0);
        }
      } else {
        Evn[x][0] = (hcol && x < g.n_experts) ? ld4i(g.E[x], b * g.lde[x] + 4 * sub, (g.out_bf16 & MML_GATE_E_BF16) != 0) : make_float4(0, 0, 0, 0);
      }
    }
#pragma unroll
    for (int gi = 0; gi < NG; ++gi)
      Gvn[gi] = (gi < g.n_gates && 4 * sub < g.gate[gi].Gd) ? ld4(g.gate[gi].G + b * g.gate[gi].ldg + 4 * sub)
                                                             : make_float4(0, 0, 0, 0);
  };
  if (iters > 0) request(0);
  for (int64_t it = 0; it < iters; ++it) {
    int64_t b = it * stride + ((int64_t)blockIdx.x * FW + wave) * SPW + grp;
    const bool valid = b < g.B;
    if (!valid) b = g.B - 1;
    for (int i = sub; i < NG * MML_MAX_EXPERTS; i += LPS) coef[i] = 0.f;
    float4 Ev[NE][HV], Gv[NG];
#pragma unroll
    for (int x = 0; x < NE; ++x)
#pragma unroll
      for (int h = 0; h < HV; ++h) Ev[x][h] = Evn[x][h];
#pragma unroll
    for (int gi = 0; gi < NG; ++gi) Gv[gi] = Gvn[gi];
    if (it + 1 < iters) request(it + 1);
    __builtin_amdgcn_wave_barrier();  // coef zeroed before the scattered writes below
    if constexpr (PACK) {
      // all eight logits' partial dots, ONE packed reduction; lane l then owns the logit of gate (l >> 2) & 1, expert slot
      // l & 3: its softmax is a max and a sum over the lane's quad and ONE expf per lane (every lane computed all eight before)
      float d8[8];
#pragma unroll
      for (int gi = 0; gi < 2; ++gi) {
        const mml_gate_desc& d = g.gate[gi < g.n_gates ? gi : 0];
        const float* W = smem + aux.wg_off[gi < g.n_gates ? gi : 0];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const bool on = gi < g.n_gates && e < d.ne && 4 * sub < d.Gd;
          d8[gi * 4 + e] = on ? dot4(Gv[gi], ld4(W + e * d.Gd + 4 * sub)) : 0.f;
        }
      }
      const float logit = pack_sum8<LPS>(d8, lane);
      const int gl = (sub >> 2) & 1, el = sub & 3;
      const bool has = gl < g.n_gates;
      const mml_gate_desc& dl_ = g.gate[has ? gl : 0];  // (per-lane choice between the two gates: selects, no branch)
      const bool live = has && el < dl_.ne;
      const float xl = live ? logit : -INFINITY;
      const float m = quad_max(xl);
      const float ex = live ? expf(xl - m) : 0.f;
      const float den = quad_sum(ex);
      const float pr = ex * (1.f / den);
      if (sub < 8 && live) {
        if (valid) dl_.P[b * dl_.ldp + el] = pr;
        coef[gl * MML_MAX_EXPERTS + emap[gl * NE + el]] = pr;
      }
    } else {
#pragma unroll
    for (int gi = 0; gi < NG; ++gi) {
      if (gi >= g.n_gates) continue;
      const mml_gate_desc& d = g.gate[gi];
      const float* W = smem + aux.wg_off[gi];
      const bool gcol = 4 * sub < d.Gd;
      float p[NE];
#pragma unroll
      for (int e = 0; e < NE; ++e) {
        p[e] = -INFINITY;
        if (e < d.ne) p[e] = group_sum<LPS>(gcol ? dot4(Gv[gi], ld4(W + e * d.Gd + 4 * sub)) : 0.f);
      }
      float m = p[0];
#pragma unroll
      for (int e = 1; e < NE; ++e) m = fmaxf(m, p[e]);
      float den = 0.f;
#pragma unroll
      for (int e = 0; e < NE; ++e) {
        p[e] = (e < d.ne) ? expf(p[e] - m) : 0.f;
        den += p[e];
      }
      const float inv = 1.f / den;
      float mine = 0.f;
#pragma unroll
      for (int e = 0; e < NE; ++e) {
        p[e] *= inv;
        if (sub == e) mine = p[e];
      }
      if (sub < d.ne) {
        if (valid) d.P[b * d.ldp + sub] = mine;
        coef[gi * MML_MAX_EXPERTS + emap[gi * NE + sub]] = mine;
      }
    }
    }
    __builtin_amdgcn_wave_barrier();
    if (hcol) {
#pragma unroll
      for (int gi = 0; gi < NG; ++gi) {
        if (gi >= g.n_gates) continue;
        float4 acc[HV];
#pragma unroll
        for (int h = 0; h < HV; ++h) acc[h] = make_float4(0, 0, 0, 0);
#pragma unroll
        for (int x = 0; x < NE; ++x) {
          const float cf = coef[gi * MML_MAX_EXPERTS + x];
#pragma unroll
          for (int h = 0; h < HV; ++h) fma4(acc[h], cf, Ev[x][h]);
        }
        if (valid) {
          const bool m16 = (g.out_bf16 & MML_GATE_MIX_BF16) != 0;
          if constexpr (HV == 2) {
            if (m16) {  // eight bf16 values: one 16-byte store
              typedef float f2 __attribute__((ext_vector_type(2)));
              typedef __bf16 b2 __attribute__((ext_vector_type(2)));
              const f2 p0 = {acc[0].x, acc[0].y}, p1 = {acc[0].z, acc[0].w}, p2 = {acc[1].x, acc[1].y}, p3 = {acc[1].z, acc[1].w};
              *reinterpret_cast<uint4*>(reinterpret_cast<uint16_t*>(g.gate[gi].mix) + b * g.gate[gi].ldmix + 8 * sub) =
                  make_uint4(__builtin_bit_cast(uint32_t, __builtin_convertvector(p0, b2)),
                             __builtin_bit_cast(uint32_t, __builtin_convertvector(p1, b2)),
                             __builtin_bit_cast(uint32_t, __builtin_convertvector(p2, b2)),
                             __builtin_bit_cast(uint32_t, __builtin_convertvector(p3, b2)));
            } else {
              st4(g.gate[gi].mix + b * g.gate[gi].ldmix + 8 * sub, acc[0]);
              st4(g.gate[gi].mix + b * g.gate[gi].ldmix + 8 * sub + 4, acc[1]);
            }
          } else {
            st4o(g.gate[gi].mix, b * g.gate[gi].ldmix + 4 * sub, acc[0], m16);
          }
#pragma unroll
          for (int h = 0; h < HV; ++h) amax_acc(am_mix, acc[h]);
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
  amax_flush(am_mix, g.amax_mix);
}

// ------------------------------------------------------------------------------------------------ gate backward
// MODE 0: any expert lists; 1: every gate mixes experts 0..ne-1 in order (MMoE); 2: any membership, but at most NE
// experts in the group -- every expert row is loaded ONCE per sample and serves all gates (PLE levels)
// HV = 2 (round 6, MODE 1 only): bf16 expert rows of up to 256 columns on 32-lane groups, eight row columns per lane -- see
// gate_fwd_once_kernel; the upstream gradient rows (fp32) are two 16-byte loads per lane, dE leaves as one 16-byte store of
// eight bf16 values (or two fp32 stores), the gate-input side (<= 128 columns) keeps four columns per lane.
template <int LPS, int NE, int NG, int MODE, int HV = 1>
__global__ __launch_bounds__(FB) void gate_bwd_fast_kernel(const mml_gate_group g, const GateFastAux aux) {
  static_assert(HV == 1 || (HV == 2 && MODE == 1), "eight columns per lane: the MMoE form only");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int SPW = 64 / LPS;
  float* Wsm = smem;                                   // [wg_total] gate weights
  float* coef_all = smem + aux.wg_total;               // [FW*SPW][NG*MML_MAX_EXPERTS] softmax coefficients per sample
  // dWg partial sums live in LDS, one [wg_total] region per LANE GROUP (= per sample slot of a wave): the group's lanes
  // read-modify-write their own 16 bytes, no atomics, fixed order.  (In registers they cost NG*NE float4 per lane --
  // 128 VGPRs for a PLE level with 3 gates over 8 experts, one or two waves per SIMD for a kernel that lives on
  // memory-level parallelism.)
  float* red = coef_all + FW * SPW * NG * MML_MAX_EXPERTS;  // [FW*SPW][wg_total]
  for (int i = threadIdx.x; i < aux.wg_total; i += FB) {
    int gi = 0;
    while (gi + 1 < g.n_gates && i >= aux.wg_off[gi + 1]) ++gi;
    Wsm[i] = g.gate[gi].active ? g.gate[gi].Wg[i - aux.wg_off[gi]] : 0.f;
  }
  for (int i = threadIdx.x; i < FW * SPW * aux.wg_total; i += FB) red[i] = 0.f;
  int* smap = reinterpret_cast<int*>(red + FW * SPW * aux.wg_total);  // [NG][NE] slot of expert x in gate gi, or -1
  if constexpr (MODE == 2) {
    for (int i = threadIdx.x; i < NG * NE; i += FB) {
      const int gi = i / NE, x = i - gi * NE;
      int slot = -1;
      if (gi < g.n_gates && g.gate[gi].active)
        for (int k = 0; k < g.gate[gi].ne; ++k)
          if (g.gate[gi].expert[k] == x) slot = k;
      smap[i] = slot;
    }
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sub = lane % LPS, grp = lane / LPS;
  float* coef = coef_all + (wave * SPW + grp) * NG * MML_MAX_EXPERTS;
  const int64_t stride = (int64_t)gridDim.x * FW * SPW;
  const int64_t iters = (g.B + stride - 1) / stride;
  const bool hcol = 4 * HV * sub < g.H;

  float* myred = red + (wave * SPW + grp) * aux.wg_total;
  float am_dg = 0.f, am_de = 0.f;  // operand magnitudes of everything this lane stores (mml_gate_group.amax_dG / amax_dE)
  // MODE 2: the (gate, expert) -> slot map in SGPRs (read through LDS + readfirstlane on every use it was 48 LDS reads
  // per pair of samples)
  int smap_s[NG * NE];
  if constexpr (MODE == 2) {
#pragma unroll
    for (int i = 0; i < NG * NE; ++i) smap_s[i] = __builtin_amdgcn_readfirstlane(smap[i]);
  }

  if constexpr (MODE == 1) {
  // every gate mixes experts 0..ne-1 in order (MMoE): one set of expert rows serves all gates and the final dE loop
  // (requesting the next trip's rows ahead, as the forward kernel does, costs 47 more VGPRs here -- 165, three waves per
  // SIMD: 92.7 us with two workgroups per CU, 96.6 with three, 110.6 with four, against 95.6 without it: not taken)
  for (int64_t it = 0; it < iters; ++it) {
    int64_t b = it * stride + ((int64_t)blockIdx.x * FW + wave) * SPW + grp;
    const bool valid = b < g.B;
    if (!valid) b = g.B - 1;
    for (int i = sub; i < NG * MML_MAX_EXPERTS; i += LPS) coef[i] = 0.f;
    // every load of the sample first (a store between two loads would serialise their latencies): upstream
    // gradients, expert rows, softmax probabilities, gate inputs
    float4 dmv[NG][HV], Gv[NG], Ev[NE][HV];
    float pv[NG][NE];
#pragma unroll
    for (int e = 0; e < NE; ++e) {
      if constexpr (HV == 2) {
        if (g.out_bf16 & MML_GATE_E_BF16) {
          uint4 q = make_uint4(0, 0, 0, 0);
          if (hcol && e < g.n_experts) q = *reinterpret_cast<const uint4*>(reinterpret_cast<const uint16_t*>(g.E[e]) + b * g.lde[e] + 8 * sub);
          Ev[e][0] = make_float4(__uint_as_float(q.x << 16), __uint_as_float(q.x & 0xffff0000u), __uint_as_float(q.y << 16), __uint_as_float(q.y & 0xffff0000u));
          Ev[e][1] = make_float4(__uint_as_float(q.z << 16), __uint_as_float(q.z & 0xffff0000u), __uint_as_float(q.w << 16), __uint_as_float(q.w & 0xffff0000u));
        } else {
          const bool on = hcol && e < g.n_experts;
          Ev[e][0] = on ? ld4(g.E[e] + b * g.lde[e] + 8 * sub) : make_float4(0, 0, 0, 0);
          Ev[e][1] = on ? ld4(g.E[e] + b * g.lde[e] + 8 * sub + 4) : make_float4(0, 0, 0, 0);
        }
      } else {
        Ev[e][0] = (hcol && e < g.n_experts) ? ld4i(g.E[e], b * g.lde[e] + 4 * sub, (g.out_bf16 & MML_GATE_E_BF16) != 0) : make_float4(0, 0, 0, 0);
      }
    }
#pragma unroll
    for (int gi = 0; gi < NG; ++gi) {
#pragma unroll
      for (int h = 0; h < HV; ++h) dmv[gi][h] = make_float4(0, 0, 0, 0);
      Gv[gi] = make_float4(0, 0, 0, 0);
#pragma unroll
      for (int e = 0; e < NE; ++e) pv[gi][e] = 0.f;
      if (gi >= g.n_gates) continue;
      const mml_gate_desc& d = g.gate[gi];
      if (!d.active) continue;
      if (hcol) {
#pragma unroll
        for (int h = 0; h < HV; ++h) dmv[gi][h] = ld4(d.dmix + b * d.lddmix + 4 * HV * sub + 4 * h);
      }
      if (4 * sub < d.Gd) Gv[gi] = ld4(d.G + b * d.ldg + 4 * sub);
#pragma unroll
      for (int e = 0; e < NE; ++e)
        if (e < d.ne) pv[gi][e] = d.P[b * d.ldp + e];
    }
#pragma unroll
    for (int gi = 0; gi < NG; ++gi) {
      if (gi >= g.n_gates) continue;
      const mml_gate_desc& d = g.gate[gi];
      if (!d.active) continue;
      float dl[NE];
      float dot = 0.f;
#pragma unroll
      for (int e = 0; e < NE; ++e) {
        dl[e] = 0.f;
        if (e < d.ne) {
          float part = 0.f;
          if (hcol) {
#pragma unroll
            for (int h = 0; h < HV; ++h) part += dot4(dmv[gi][h], Ev[e][h]);
          }
          dl[e] = group_sum<LPS>(part);
          dot += pv[gi][e] * dl[e];
          if (sub == 0) coef[gi * MML_MAX_EXPERTS + e] = pv[gi][e];
        }
      }
#pragma unroll
      for (int e = 0; e < NE; ++e) dl[e] = valid ? pv[gi][e] * (dl[e] - dot) : 0.f;  // dlogit (0 for padding samples)
      if (4 * sub < d.Gd) {
        const float* W = Wsm + aux.wg_off[gi];
        float4 dg = make_float4(0, 0, 0, 0);
#pragma unroll
        for (int e = 0; e < NE; ++e)
          if (e < d.ne) {
            fma4(dg, dl[e], ld4(W + e * d.Gd + 4 * sub));
            float* r = myred + aux.wg_off[gi] + e * d.Gd + 4 * sub;
            float4 t = ld4(r);
            fma4(t, dl[e], Gv[gi]);
            st4(r, t);
          }
        if (d.g_relu) {
          if (!(Gv[gi].x > 0.f)) dg.x = 0.f;
          if (!(Gv[gi].y > 0.f)) dg.y = 0.f;
          if (!(Gv[gi].z > 0.f)) dg.z = 0.f;
          if (!(Gv[gi].w > 0.f)) dg.w = 0.f;
        }
        if (valid) {
          st4o(d.dG, b * d.lddg + 4 * sub, dg, (g.out_bf16 & MML_GATE_DG_BF16) != 0);
          amax_acc(am_dg, dg);
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
    if (hcol) {
#pragma unroll
      for (int x = 0; x < NE; ++x) {
        if (x >= g.n_experts) continue;
        float4 acc[HV];
#pragma unroll
        for (int h = 0; h < HV; ++h) acc[h] = make_float4(0, 0, 0, 0);
#pragma unroll
        for (int gi = 0; gi < NG; ++gi) {
          const float cf = coef[gi * MML_MAX_EXPERTS + x];
#pragma unroll
          for (int h = 0; h < HV; ++h) fma4(acc[h], cf, dmv[gi][h]);
        }
        if (g.e_relu) {
#pragma unroll
          for (int h = 0; h < HV; ++h) {
            if (!(Ev[x][h].x > 0.f)) acc[h].x = 0.f;
            if (!(Ev[x][h].y > 0.f)) acc[h].y = 0.f;
            if (!(Ev[x][h].z > 0.f)) acc[h].z = 0.f;
            if (!(Ev[x][h].w > 0.f)) acc[h].w = 0.f;
          }
        }
        if (valid) {
          const bool e16 = (g.out_bf16 & MML_GATE_DE_BF16) != 0;
          if constexpr (HV == 2) {
            if (e16) {
              typedef float f2 __attribute__((ext_vector_type(2)));
              typedef __bf16 b2 __attribute__((ext_vector_type(2)));
              const f2 p0 = {acc[0].x, acc[0].y}, p1 = {acc[0].z, acc[0].w}, p2 = {acc[1].x, acc[1].y}, p3 = {acc[1].z, acc[1].w};
              *reinterpret_cast<uint4*>(reinterpret_cast<uint16_t*>(g.dE[x]) + b * g.ldde[x] + 8 * sub) =
                  make_uint4(__builtin_bit_cast(uint32_t, __builtin_convertvector(p0, b2)),
                             __builtin_bit_cast(uint32_t, __builtin_convertvector(p1, b2)),
                             __builtin_bit_cast(uint32_t, __builtin_convertvector(p2, b2)),
                             __builtin_bit_cast(uint32_t, __builtin_convertvector(p3, b2)));
            } else {
              st4(g.dE[x] + b * g.ldde[x] + 8 * sub, acc[0]);
              st4(g.dE[x] + b * g.ldde[x] + 8 * sub + 4, acc[1]);
            }
          } else {
            st4o(g.dE[x], b * g.ldde[x] + 4 * sub, acc[0], e16);
          }
#pragma unroll
          for (int h = 0; h < HV; ++h) amax_acc(am_de, acc[h]);
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
  } else if constexpr (MODE == 2) {
  for (int64_t it = 0; it < iters; ++it) {
    int64_t b = it * stride + ((int64_t)blockIdx.x * FW + wave) * SPW + grp;
    const bool valid = b < g.B;
    if (!valid) b = g.B - 1;
    for (int i = sub; i < NG * MML_MAX_EXPERTS; i += LPS) coef[i] = 0.f;
    // all loads of the sample first: every expert row once, the upstream gradients and gate inputs of every gate
    float4 Ev[NE], dmv[NG], Gv[NG];
#pragma unroll
    for (int x = 0; x < NE; ++x)
      Ev[x] = (hcol && x < g.n_experts) ? ld4i(g.E[x], b * g.lde[x] + 4 * sub, (g.out_bf16 & MML_GATE_E_BF16) != 0) : make_float4(0, 0, 0, 0);
#pragma unroll
    for (int gi = 0; gi < NG; ++gi) {
      dmv[gi] = make_float4(0, 0, 0, 0);
      Gv[gi] = make_float4(0, 0, 0, 0);
      if (gi >= g.n_gates) continue;
      const mml_gate_desc& d = g.gate[gi];
      if (!d.active) continue;
      if (hcol) dmv[gi] = ld4(d.dmix + b * d.lddmix + 4 * sub);
      if (4 * sub < d.Gd) Gv[gi] = ld4(d.G + b * d.ldg + 4 * sub);
      // softmax probabilities, by EXPERT index, through LDS: lane x of the group fetches the one of expert x
      if (sub < NE) {
        const int slot = smap[gi * NE + sub];
        if (slot >= 0) coef[gi * MML_MAX_EXPERTS + sub] = d.P[b * d.ldp + slot];
      }
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int gi = 0; gi < NG; ++gi) {
      if (gi >= g.n_gates) continue;
      const mml_gate_desc& d = g.gate[gi];
      if (!d.active) continue;
      float dl[NE];
      float dot = 0.f;
#pragma unroll
      for (int x = 0; x < NE; ++x) {
        dl[x] = 0.f;
        const int slot = smap_s[gi * NE + x];
        if (slot >= 0) {
          dl[x] = group_sum<LPS>(hcol ? dot4(dmv[gi], Ev[x]) : 0.f);
          dot += coef[gi * MML_MAX_EXPERTS + x] * dl[x];
        }
      }
#pragma unroll
      for (int x = 0; x < NE; ++x) dl[x] = valid ? coef[gi * MML_MAX_EXPERTS + x] * (dl[x] - dot) : 0.f;  // dlogit
      if (4 * sub < d.Gd) {
        const float* W = Wsm + aux.wg_off[gi];
        float4 dg = make_float4(0, 0, 0, 0);
#pragma unroll
        for (int x = 0; x < NE; ++x) {
          const int slot = smap_s[gi * NE + x];
          if (slot >= 0) {
            fma4(dg, dl[x], ld4(W + slot * d.Gd + 4 * sub));
            float* r = myred + aux.wg_off[gi] + slot * d.Gd + 4 * sub;
            float4 t = ld4(r);
            fma4(t, dl[x], Gv[gi]);
            st4(r, t);
          }
        }
        if (d.g_relu) {
          if (!(Gv[gi].x > 0.f)) dg.x = 0.f;
          if (!(Gv[gi].y > 0.f)) dg.y = 0.f;
          if (!(Gv[gi].z > 0.f)) dg.z = 0.f;
          if (!(Gv[gi].w > 0.f)) dg.w = 0.f;
        }
        if (valid) {
          st4o(d.dG, b * d.lddg + 4 * sub, dg, (g.out_bf16 & MML_GATE_DG_BF16) != 0);
          amax_acc(am_dg, dg);
        }
      }
    }
    if (hcol) {
#pragma unroll
      for (int x = 0; x < NE; ++x) {
        if (x >= g.n_experts) continue;
        float4 acc = make_float4(0, 0, 0, 0);
#pragma unroll
        for (int gi = 0; gi < NG; ++gi) fma4(acc, coef[gi * MML_MAX_EXPERTS + x], dmv[gi]);
        if (g.e_relu) {
          if (!(Ev[x].x > 0.f)) acc.x = 0.f;
          if (!(Ev[x].y > 0.f)) acc.y = 0.f;
          if (!(Ev[x].z > 0.f)) acc.z = 0.f;
          if (!(Ev[x].w > 0.f)) acc.w = 0.f;
        }
        if (valid) {
          st4o(g.dE[x], b * g.ldde[x] + 4 * sub, acc, (g.out_bf16 & MML_GATE_DE_BF16) != 0);
          amax_acc(am_de, acc);
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
  } else {
  for (int64_t it = 0; it < iters; ++it) {
    int64_t b = it * stride + ((int64_t)blockIdx.x * FW + wave) * SPW + grp;
    const bool valid = b < g.B;
    if (!valid) b = g.B - 1;
    for (int i = sub; i < NG * MML_MAX_EXPERTS; i += LPS) coef[i] = 0.f;
    float4 dmv[NG];
#pragma unroll
    for (int gi = 0; gi < NG; ++gi) {
      dmv[gi] = make_float4(0, 0, 0, 0);
      if (gi >= g.n_gates) continue;
      const mml_gate_desc& d = g.gate[gi];
      if (!d.active) continue;
      if (hcol) dmv[gi] = ld4(d.dmix + b * d.lddmix + 4 * sub);
      float dl[NE], p[NE];
      float dot = 0.f;
      {
        // all loads of the gate first, in straight-line code (clamped expert slot), then the reductions
        float4 Ev[NE];
#pragma unroll
        for (int e = 0; e < NE; ++e) {
          const int ec = e < d.ne ? e : d.ne - 1;
          const int x = d.expert[ec];
          Ev[e] = hcol ? ld4i(g.E[x], b * g.lde[x] + 4 * sub, (g.out_bf16 & MML_GATE_E_BF16) != 0) : make_float4(0, 0, 0, 0);
          p[e] = d.P[b * d.ldp + ec];
        }
#pragma unroll
        for (int e = 0; e < NE; ++e) {
          dl[e] = 0.f;
          if (e < d.ne) {
            dl[e] = group_sum<LPS>(dot4(dmv[gi], Ev[e]));
            dot += p[e] * dl[e];
            if (sub == 0) coef[gi * MML_MAX_EXPERTS + d.expert[e]] = p[e];
          } else {
            p[e] = 0.f;
          }
        }
      }
#pragma unroll
      for (int e = 0; e < NE; ++e) dl[e] = valid ? p[e] * (dl[e] - dot) : 0.f;  // dlogit (0 for padding samples)
      if (4 * sub < d.Gd) {
        const float4 Gv = ld4(d.G + b * d.ldg + 4 * sub);
        const float* W = Wsm + aux.wg_off[gi];
        float4 dg = make_float4(0, 0, 0, 0);
#pragma unroll
        for (int e = 0; e < NE; ++e)
          if (e < d.ne) {
            fma4(dg, dl[e], ld4(W + e * d.Gd + 4 * sub));
            float* r = myred + aux.wg_off[gi] + e * d.Gd + 4 * sub;
            float4 t = ld4(r);
            fma4(t, dl[e], Gv);
            st4(r, t);
          }
        if (d.g_relu) {
          if (!(Gv.x > 0.f)) dg.x = 0.f;
          if (!(Gv.y > 0.f)) dg.y = 0.f;
          if (!(Gv.z > 0.f)) dg.z = 0.f;
          if (!(Gv.w > 0.f)) dg.w = 0.f;
        }
        if (valid) {
          st4o(d.dG, b * d.lddg + 4 * sub, dg, (g.out_bf16 & MML_GATE_DG_BF16) != 0);
          amax_acc(am_dg, dg);
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
    if (hcol) {
#pragma unroll 4
      for (int x = 0; x < g.n_experts; ++x) {
        float4 acc = make_float4(0, 0, 0, 0);
#pragma unroll
        for (int gi = 0; gi < NG; ++gi) fma4(acc, coef[gi * MML_MAX_EXPERTS + x], dmv[gi]);
        if (g.e_relu) {
          const float4 Ev = ld4i(g.E[x], b * g.lde[x] + 4 * sub, (g.out_bf16 & MML_GATE_E_BF16) != 0);
          if (!(Ev.x > 0.f)) acc.x = 0.f;
          if (!(Ev.y > 0.f)) acc.y = 0.f;
          if (!(Ev.z > 0.f)) acc.z = 0.f;
          if (!(Ev.w > 0.f)) acc.w = 0.f;
        }
        if (valid) {
          st4o(g.dE[x], b * g.ldde[x] + 4 * sub, acc, (g.out_bf16 & MML_GATE_DE_BF16) != 0);
          amax_acc(am_de, acc);
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
  }
  amax_flush(am_dg, g.amax_dG);
  amax_flush(am_de, g.amax_dE);
  // combine: fixed-order sum over the lane-group regions -> slab
  __syncthreads();
  float* out = aux.slab + (int64_t)blockIdx.x * aux.wg_total;
  for (int i = threadIdx.x; i < aux.wg_total; i += FB) {
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < FW * SPW; ++w) s += red[w * aux.wg_total + i];
    out[i] = s;
  }
}

// ------------------------------------------------------------------------------------------------ heads
// GATED (round 6): heads whose input is Hin (.) gate (mml_head_desc.gate; PepNet's last PPNet layer): the product is formed
// in registers from the two rows, the backward writes dH = dlogit w gate relu'(Hin) and dgate = dlogit w Hin act'(gate) --
// the product, its gradient and the element-wise launches around them never touch memory.
// samples per lane group and trip (see the loop below).  Round 6: four for up to four plain heads as well (STAR's four heads:
// 90 -> 84 us at B = 65 536, 206 VGPRs at two workgroups per CU); gated heads hold two rows per sample and head: two.
#ifndef MML_HEAD_U
#define MML_HEAD_U (NT <= 2 ? 4 : ((NT <= 4 && !GATED) ? 4 : 2))
#endif
template <int LPS, int NT, bool GATED>
__global__ __launch_bounds__(FB) void head_fast_kernel(const mml_head_group g, const HeadFastAux aux) {
  __shared__ float red[FW][NT * (4 * LPS + 1) + 1];
  constexpr int SPW = 64 / LPS;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sub = lane % LPS, grp = lane / LPS;
  const int64_t stride = (int64_t)gridDim.x * FW * SPW;
  const int64_t iters = (g.B + stride - 1) / stride;
  float4 dwacc[NT];
  float dbacc[NT];
  float lossacc = 0.f;
  float am_dh = 0.f, am_dg = 0.f;
  float4 wv[NT];
  float bias[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    dwacc[t] = make_float4(0, 0, 0, 0);
    dbacc[t] = 0.f;
    wv[t] = make_float4(0, 0, 0, 0);
    bias[t] = 0.f;
    if (t < g.n_heads) {
      const mml_head_desc& d = g.head[t];
      if (4 * sub < d.H) {
        wv[t] = ld4(d.w + 4 * sub);
        if (d.w2) {
          const float4 w2 = ld4(d.w2 + 4 * sub);
          wv[t].x *= w2.x; wv[t].y *= w2.y; wv[t].z *= w2.z; wv[t].w *= w2.w;
        }
      }
      bias[t] = d.bias[0];
      for (int i = 0; i < d.n_bias2; ++i) bias[t] += d.bias2[i];
    }
  }
  // U samples per lane group and trip: every load of the trip (U x n_heads rows, labels, masks) is issued before the
  // first logit is formed.  One sample per trip (until round 5) left a wave with two 16-byte loads in flight in front of
  // a chain of DPP sums, exp, log and log1p: 28 us for 67 MB at B = 65 536 (2.3 TB/s).  The samples of a lane group
  // are visited in the same order as before: the partial sums of dw / dbias / loss are the same bits.
  constexpr int U = MML_HEAD_U;
  for (int64_t it0 = 0; it0 < iters; it0 += U) {
    int64_t bb[U];
    bool vld[U];
    float4 hvu[U][NT];
    float4 gvu[GATED ? U : 1][GATED ? NT : 1];
    float yu[U][NT], mu[U][NT];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      bb[u] = (it0 + u) * stride + ((int64_t)blockIdx.x * FW + wave) * SPW + grp;
      vld[u] = bb[u] < g.B;
      if (!vld[u]) bb[u] = g.B - 1;
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        hvu[u][t] = make_float4(0, 0, 0, 0);
        if (GATED) gvu[u][t] = make_float4(1.f, 1.f, 1.f, 1.f);
        yu[u][t] = 0.f;
        mu[u][t] = 1.f;
        if (t >= g.n_heads || it0 + u >= iters) continue;  // (a trip past the last one: uniform, nothing loaded)
        const mml_head_desc& d = g.head[t];
        if (4 * sub < d.H) hvu[u][t] = ld4(d.Hin + bb[u] * d.ldh + 4 * sub);
        if (GATED && d.gate && 4 * sub < d.H) gvu[u][t] = ld4(d.gate + bb[u] * d.ldgate + 4 * sub);
        if (d.mask_col >= 0 && g.mask) mu[u][t] = g.mask[bb[u] * g.ldmask + d.mask_col];
        if (aux.train) yu[u][t] = g.y ? g.y[bb[u] * g.ldy + t] : g.dprob[bb[u] * g.lddprob + t];
      }
    }
    // the logits of the trip: after the lane-group sums every lane of a group holds all U x NT of them
    constexpr int NI = U * NT;  // (<= 16 <= LPS)
    float lg[U][NT];
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int t = 0; t < NT; ++t)
        lg[u][t] = (t < g.n_heads && it0 + u < iters)
                       ? group_sum<LPS>(dot4(GATED ? mul4(hvu[u][t], gvu[u][t]) : hvu[u][t], wv[t])) + bias[t] : 0.f;
    // The per-sample scalar chain (sigmoid, the two clamped logarithms of the BCE, its derivative) ONCE per trip: lane
    // j of a group takes item j = (u, t) of the trip instead of every lane repeating all NI chains -- the chain is ~150
    // instructions with three transcendentals and two divisions, and it, not the 67 MB, was the kernel's time
    // (28 us at B = 65 536 for AE-30's two heads, 92 us for PepNet's four).  Same instructions on the same values:
    // probabilities, gradients and the dw / dbias sums are the bits of the former form; the loss is now summed per lane
    // first (fixed order, deterministic).
    const int j = sub & (NI - 1);
    const int tj = j % NT;
    float lgj = 0.f, mj = 1.f, yj = 0.f;
    bool vj = false;
    int64_t bj = 0;
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int t = 0; t < NT; ++t)
        if (j == u * NT + t) {
          lgj = lg[u][t];
          mj = mu[u][t];
          yj = yu[u][t];
          vj = vld[u] && t < g.n_heads && it0 + u < iters;
          bj = bb[u];
        }
    const bool own = vj && sub < NI;
    const float pj = 1.f / (1.f + expf(-lgj));
    const float pmj = pj * mj;
    if (own) g.prob[bj * g.ldprob + tj] = pmj;
    float dlj = 0.f;
    if (aux.train) {
      float dpm;
      if (g.y) {
        const float lp = bce_log_clamp(logf(pmj));
        const float l1p = bce_log_clamp(log1pf(-pmj));
        if (own) lossacc += -(yj * lp + (1.f - yj) * l1p);
        dpm = (pmj - yj) / fmaxf((1.f - pmj) * pmj, 1e-12f);
      } else {
        dpm = yj;
      }
      dlj = vj ? dpm * mj * pj * (1.f - pj) : 0.f;
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (it0 + u >= iters) break;
        const int64_t b = bb[u];
        const bool valid = vld[u];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          if (t >= g.n_heads) continue;
          const mml_head_desc& d = g.head[t];
          const bool col = 4 * sub < d.H;
          const float4 hv = hvu[u][t];
          const float dlogit = __shfl(dlj, (lane & ~(LPS - 1)) + u * NT + t, 64);
          if (sub == 0) dbacc[t] += dlogit;
          if (col) {
            float4 dh = make_float4(dlogit * wv[t].x, dlogit * wv[t].y, dlogit * wv[t].z, dlogit * wv[t].w);
            if (GATED && d.gate) {  // (uniform per head) input = hv (.) gv
              const float4 gv = gvu[u][t];
              fma4(dwacc[t], dlogit, mul4(hv, gv));
              float4 dg = mul4(dh, hv);
              if (d.gate_act != MML_ACT_NONE) {
                dg.x *= act_bwd_rows(gv.x, d.gate_act); dg.y *= act_bwd_rows(gv.y, d.gate_act);
                dg.z *= act_bwd_rows(gv.z, d.gate_act); dg.w *= act_bwd_rows(gv.w, d.gate_act);
              }
              dh = mul4(dh, gv);
              if (valid) {
                st4o(d.dgate, b * d.lddgate + 4 * sub, dg, false);
                amax_acc(am_dg, dg);
              }
            } else {
              fma4(dwacc[t], dlogit, hv);
            }
            if (d.h_relu) {
              if (!(hv.x > 0.f)) dh.x = 0.f;
              if (!(hv.y > 0.f)) dh.y = 0.f;
              if (!(hv.z > 0.f)) dh.z = 0.f;
              if (!(hv.w > 0.f)) dh.w = 0.f;
            }
            if (valid) {
              st4o(d.dH, b * d.lddh + 4 * sub, dh, g.dh_bf16 != 0);
              amax_acc(am_dh, dh);
            }
          }
        }
      }
    }
  }
  if (!aux.train) return;
  amax_flush(am_dh, g.amax_dH);
  if (GATED) amax_flush(am_dg, g.amax_dG);
  constexpr int PH = 4 * LPS + 1;  // per head: dw (4*LPS slots) + dbias
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    if (t >= g.n_heads) continue;
    float4 v = dwacc[t];
    v.x = cross_group_sum<LPS>(v.x);
    v.y = cross_group_sum<LPS>(v.y);
    v.z = cross_group_sum<LPS>(v.z);
    v.w = cross_group_sum<LPS>(v.w);
    const float db = cross_group_sum<LPS>(dbacc[t]);
    if (grp == 0) {
      red[wave][t * PH + 4 * sub + 0] = v.x;
      red[wave][t * PH + 4 * sub + 1] = v.y;
      red[wave][t * PH + 4 * sub + 2] = v.z;
      red[wave][t * PH + 4 * sub + 3] = v.w;
      if (sub == 0) red[wave][t * PH + 4 * LPS] = db;
    }
  }
  const float ls = cross_group_sum<LPS>(group_sum<LPS>(lossacc));
  if (lane == 0) red[wave][NT * PH] = ls;
  __syncthreads();
  float* out = aux.slab + (int64_t)blockIdx.x * aux.stride;
  for (int i = threadIdx.x; i < g.n_heads * (aux.hmax + 1) + 1; i += FB) {
    int src;
    if (i == g.n_heads * (aux.hmax + 1)) {
      src = NT * PH;
    } else {
      const int t = i / (aux.hmax + 1), h = i % (aux.hmax + 1);
      src = t * PH + (h == aux.hmax ? 4 * LPS : h);
    }
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < FW; ++w) s += red[w][src];
    out[i] = s;
  }
}

// ------------------------------------------------------------------------------------------------ host side
static int pick_lps(int width) { return width <= 64 ? 16 : (width <= 128 ? 32 : (width <= 256 ? 64 : 0)); }

int fast_row_grid(int64_t B, int lps, int per_cu = 4) {
  const int spb = FW * (64 / lps);
  int64_t blocks = cdiv(B, spb);
  if (blocks > 256 * per_cu) blocks = 256 * per_cu;  // persistent: one partial slab row per workgroup goes to the reducer
  if (blocks < 1) blocks = 1;
  return (int)blocks;
}

static bool ok4(const void* p, int64_t ld) { return aligned16(p) && (ld % 4 == 0); }

// does the load-once forward take this group on lane groups of `lps` lanes?  (MMLREC_GATE_FWD_MODE=0: per-gate kernel
// everywhere, a measurement knob)
static size_t gate_fwd_once_lds(const GateFastAux& aux, int lps) {
  return ((size_t)aux.wg_total + (size_t)FW * (64 / lps) * aux.ng * MML_MAX_EXPERTS + (size_t)aux.ng * aux.ne) * 4;
}
static bool gate_fwd_once_ok(const mml_gate_group& g, const GateFastAux& aux, int lps) {
  static int forced = -2;
  if (forced == -2) {
    const char* e = getenv("MMLREC_GATE_FWD_MODE");
    forced = e ? atoi(e) : -1;
  }
  return forced != 0 && g.n_experts <= aux.ne && aux.ne * aux.ng <= 32 && gate_fwd_once_lds(aux, lps) <= 60 * 1024;
}

int gate_fast_config(const mml_gate_group* g, bool bwd, GateFastAux& aux) {
  if (g->H % 4 || g->H > 256) return 0;
  int width = g->H, nemax = 0, off = 0;
  for (int x = 0; x < g->n_experts; ++x) {
    if (!ok4(g->E[x], g->lde[x])) return 0;
    if (bwd && !ok4(g->dE[x], g->ldde[x])) return 0;
  }
  for (int i = 0; i < g->n_gates; ++i) {
    const mml_gate_desc& d = g->gate[i];
    if (d.Gd % 4 || d.Gd > 256 || !ok4(d.G, d.ldg) || !aligned16(d.Wg)) return 0;
    if (!bwd && !ok4(d.mix, d.ldmix)) return 0;
    if (bwd && d.active && (!ok4(d.dmix, d.lddmix) || !ok4(d.dG, d.lddg))) return 0;
    if (d.Gd > width) width = d.Gd;
    if (d.ne > nemax) nemax = d.ne;
    aux.wg_off[i] = off;
    off += d.ne * d.Gd;
  }
  aux.wg_total = off;
  aux.ident = 1;
  for (int i = 0; i < g->n_gates; ++i) {
    const mml_gate_desc& d = g->gate[i];
    if (d.ne != g->n_experts) aux.ident = 0;
    for (int e = 0; e < d.ne; ++e)
      if (d.expert[e] != e) aux.ident = 0;
  }
  aux.lps = pick_lps(width);
  aux.ne = nemax <= 4 ? 4 : (nemax <= 8 ? 8 : 16);
  aux.ng = g->n_gates <= 2 ? 2 : (g->n_gates == 3 ? 3 : (g->n_gates <= 4 ? 4 : 8));  // 3: a PLE level at T = 2
  aux.hv = 1;
  // forward, bf16 expert rows wider than 128 columns under gate inputs of at most 128 (KuaiRec-32 in the bf16-storage mode):
  // 32-lane groups with eight row columns per lane (gate_fwd_once_kernel<.., HV = 2>; MMLREC_GATE_HV=0: one sample per wave)
  static int bwd_forced = -2;
  if (bwd_forced == -2) {
    const char* e = getenv("MMLREC_GATE_BWD_MODE");
    bwd_forced = e ? atoi(e) : -1;
  }
  // (lab, MMLREC_GATE_HV=2: the same form on 16-lane groups for fp32 rows of 65..128 columns under gate inputs of at most 64 --
  //  AE-30's MMoE: four samples per wave and trip)
  static int hv16 = -1;
  if (hv16 < 0) {
    const char* e = getenv("MMLREC_GATE_HV");
    hv16 = (e && atoi(e) == 2) ? 1 : 0;
  }
  const bool wide16 = (g->out_bf16 & MML_GATE_E_BF16) && g->H > 128;
  const bool narrow32 = hv16 && !(g->out_bf16 & MML_GATE_E_BF16) && g->H > 64 && g->H <= 128;
  if ((wide16 || narrow32) && g->H % 8 == 0 && aux.ne == 4 && aux.ng == 2 && gate_pack_on() &&
      (!bwd || (aux.ident && bwd_forced == -1))) {
    const int lps_hv = wide16 ? 32 : 16, gd_max = wide16 ? 128 : 64;
    static int hv_on = -1;
    if (hv_on < 0) {
      const char* e = getenv("MMLREC_GATE_HV");
      hv_on = (e && atoi(e) == 0) ? 0 : 1;
    }
    bool ok = hv_on != 0 && g->n_experts <= 4 && (bwd || gate_fwd_once_ok(*g, aux, lps_hv));  // (only these kernels have the form)
    for (int x = 0; x < g->n_experts && ok; ++x)
      ok = g->lde[x] % 8 == 0 && (!bwd || g->ldde[x] % ((g->out_bf16 & MML_GATE_DE_BF16) ? 8 : 4) == 0);
    for (int i = 0; i < g->n_gates && ok; ++i)
      ok = g->gate[i].Gd <= gd_max && (bwd || g->gate[i].ldmix % ((g->out_bf16 & MML_GATE_MIX_BF16) ? 8 : 4) == 0);
    if (ok) {
      aux.lps = lps_hv;
      aux.hv = 2;
    }
  }
  if (bwd && aux.ne * aux.ng > 32) return 0;  // register budget (upstream gradients, coefficients)
  static int fwd_per_cu = -1;
  if (fwd_per_cu < 0) {
    const char* e = getenv("MMLREC_GATE_FWD_WGS");
    fwd_per_cu = e ? atoi(e) : 4;
  }
  // the forward kernel writes no per-workgroup partials.  Round 5: with the next trip's rows requested ahead (91 VGPRs for
  // MMoE's 4 experts x 2 gates) FOUR workgroups per CU are the best grid -- AE-30 at B = 65 536, device time of the call:
  // 45.7 us (5.2 TB/s of its own bytes) against 48.4 with five, 54.5 with six, 52.8 with eight; before the prefetch six
  // were one round of resident workgroups (64 us; eight 67-68, twelve 72).
  static int bwd_per_cu = -1;
  if (bwd_per_cu < 0) {
    const char* e = getenv("MMLREC_GATE_BWD_WGS");
    bwd_per_cu = e ? atoi(e) : 4;
  }
  // (the forward of a PLE level -- 8 experts x 3 gates, 148 VGPRs: three waves per SIMD -- is one round of resident
  // workgroups at three per CU: 119.9 us against 126.6 with four, 132.0 with two, B = 65 536)
  const int fwd_here = (!getenv("MMLREC_GATE_FWD_WGS") && aux.ne * aux.ng > 16) ? 3 : fwd_per_cu;
  aux.grid = fast_row_grid(g->B, aux.lps, bwd ? bwd_per_cu : fwd_here);
  return aux.lps;
}

template <int LPS>
static void launch_gate_fwd(const mml_gate_group& g, const GateFastAux& aux, hipStream_t st) {
  dim3 gr(aux.grid), bl(FB);
  const size_t lds1 = gate_fwd_once_lds(aux, LPS);
  if (gate_fwd_once_ok(g, aux, LPS)) {
#define MML_GF(NE_, NG_) MML_LAUNCH((gate_fwd_once_kernel<LPS, NE_, NG_>), gr, bl, lds1, st, g, aux)
    if (aux.ne == 4 && aux.ng == 2) {
      if constexpr (LPS == 32 || LPS == 16) {
        if (aux.hv == 2) {
          MML_LAUNCH((gate_fwd_once_kernel<LPS, 4, 2, true, 2>), gr, bl, lds1, st, g, aux);
          return;
        }
      }
      if (gate_pack_on()) MML_LAUNCH((gate_fwd_once_kernel<LPS, 4, 2, true>), gr, bl, lds1, st, g, aux);
      else MML_GF(4, 2);
      return;
    }
    if (aux.ne == 4 && aux.ng == 3) { MML_GF(4, 3); return; }
    if (aux.ne == 4 && aux.ng == 4) { MML_GF(4, 4); return; }
    if (aux.ne == 4 && aux.ng == 8) { MML_GF(4, 8); return; }
    if (aux.ne == 8 && aux.ng == 2) { MML_GF(8, 2); return; }
    if (aux.ne == 8 && aux.ng == 3) { MML_GF(8, 3); return; }
    if (aux.ne == 8 && aux.ng == 4) { MML_GF(8, 4); return; }
    if (aux.ne == 16 && aux.ng == 2) { MML_GF(16, 2); return; }
#undef MML_GF
  }
  const size_t lds = (size_t)aux.wg_total * 4;
  if (aux.ne == 4) MML_LAUNCH((gate_fwd_fast_kernel<LPS, 4>), gr, bl, lds, st, g, aux);
  else if (aux.ne == 8) MML_LAUNCH((gate_fwd_fast_kernel<LPS, 8>), gr, bl, lds, st, g, aux);
  else MML_LAUNCH((gate_fwd_fast_kernel<LPS, 16>), gr, bl, lds, st, g, aux);
}

int gate_fwd_fast(const mml_gate_group* g, hipStream_t st) {
  GateFastAux aux{};
  if (!gate_fast_config(g, false, aux)) return 1;  // not handled
  if ((size_t)aux.wg_total * 4 > 48 * 1024) return 1;
  if (aux.lps == 16) launch_gate_fwd<16>(*g, aux, st);
  else if (aux.lps == 32) launch_gate_fwd<32>(*g, aux, st);
  else launch_gate_fwd<64>(*g, aux, st);
  return check_launch("mml_gate_mix_fwd(fast)");
}

size_t gate_bwd_fast_lds(const GateFastAux& aux) {
  const int spw = 64 / aux.lps;
  return ((size_t)aux.wg_total + (size_t)FW * spw * aux.ng * MML_MAX_EXPERTS + (size_t)FW * spw * aux.wg_total +
          (size_t)aux.ng * aux.ne) * 4;
}

template <int LPS>
static int launch_gate_bwd(const mml_gate_group& g, const GateFastAux& aux, hipStream_t st) {
  const size_t lds = gate_bwd_fast_lds(aux);
  dim3 gr(aux.grid), bl(FB);
  static int forced = -2;  // MMLREC_GATE_BWD_MODE=0|1|2: measurement knob (default: best applicable mode)
  if (forced == -2) {
    const char* e = getenv("MMLREC_GATE_BWD_MODE");
    forced = e ? atoi(e) : -1;
  }
#define MML_GB(NE_, NG_)                                                                       \
  do {                                                                                         \
    if (aux.ident && NE_ * NG_ <= 8 && forced != 2 && forced != 0) MML_LAUNCH((gate_bwd_fast_kernel<LPS, NE_, NG_, 1>), gr, bl, lds, st, g, aux); \
    else if (g.n_experts <= NE_ && NE_ * NG_ <= 32 && forced != 0) MML_LAUNCH((gate_bwd_fast_kernel<LPS, NE_, NG_, 2>), gr, bl, lds, st, g, aux); \
    else MML_LAUNCH((gate_bwd_fast_kernel<LPS, NE_, NG_, 0>), gr, bl, lds, st, g, aux);   \
  } while (0)
  if (aux.hv == 2) {  // (gate_fast_config: the MMoE form on 32-lane groups)
    if constexpr (LPS == 32 || LPS == 16) MML_LAUNCH((gate_bwd_fast_kernel<LPS, 4, 2, 1, 2>), gr, bl, lds, st, g, aux);
    else return 1;
  } else if (aux.ne == 4 && aux.ng == 2) MML_GB(4, 2);
  else if (aux.ne == 4 && aux.ng == 4) MML_GB(4, 4);
  else if (aux.ne == 4 && aux.ng == 8) MML_GB(4, 8);
  else if (aux.ne == 8 && aux.ng == 2) MML_GB(8, 2);
  else if (aux.ne == 8 && aux.ng == 3) MML_GB(8, 3);
  else if (aux.ne == 4 && aux.ng == 3) MML_GB(4, 3);
  else if (aux.ne == 8 && aux.ng == 4) MML_GB(8, 4);
  else if (aux.ne == 16 && aux.ng == 2) MML_GB(16, 2);
  else return 1;
#undef MML_GB
  return check_launch("mml_gate_mix_bwd(fast)");
}

bool gate_bwd_fast_serves(const mml_gate_group* g, const GateFastAux& aux) {
  (void)g;
  return gate_bwd_fast_lds(aux) <= 60 * 1024;
}

int gate_bwd_fast(const mml_gate_group* g, GateFastAux& aux, hipStream_t st) {
  if (!gate_bwd_fast_serves(g, aux)) return 1;
  if (aux.lps == 16) return launch_gate_bwd<16>(*g, aux, st);
  if (aux.lps == 32) return launch_gate_bwd<32>(*g, aux, st);
  return launch_gate_bwd<64>(*g, aux, st);
}

int head_fast_config(const mml_head_group* g, bool train, int hmax, HeadFastAux& aux) {
  if (hmax % 4 || hmax > 256) return 0;
  for (int t = 0; t < g->n_heads; ++t) {
    const mml_head_desc& d = g->head[t];
    if (d.H % 4 || !ok4(d.Hin, d.ldh) || !aligned16(d.w) || (d.w2 && !aligned16(d.w2))) return 0;
    if (train && !ok4(d.dH, d.lddh)) return 0;
    if (d.gate && (!ok4(d.gate, d.ldgate) || (train && !ok4(d.dgate, d.lddgate)) || g->dh_bf16)) return 0;
    if (d.gate) aux.gated = 1;
  }
  aux.lps = pick_lps(hmax);
  aux.nt = g->n_heads <= 2 ? 2 : (g->n_heads <= 4 ? 4 : 8);
  aux.hmax = hmax;
  static int head_per_cu = -1;
  if (head_per_cu < 0) {
    const char* e = getenv("MMLREC_HEAD_WGS");
    head_per_cu = e ? atoi(e) : 2;
  }
  // (two workgroups per CU since the trips hold U samples per lane group: the call -- kernel + the reduction of one
  // partial row per workgroup -- takes 29.4 us on AE-30 / 77.9 on PepNet at B = 65 536 against 34.3 / 83.0 with four,
  // 47.8 / 84.0 with eight, 31.5 / 104.8 with one)
  aux.grid = fast_row_grid(g->B, aux.lps, head_per_cu);
  return aux.lps;
}

template <int LPS>
static void launch_head(const mml_head_group& g, const HeadFastAux& aux, hipStream_t st) {
  dim3 gr(aux.grid), bl(FB);
  if (aux.gated) {
    if (aux.nt == 2) MML_LAUNCH((head_fast_kernel<LPS, 2, true>), gr, bl, 0, st, g, aux);
    else if (aux.nt == 4) MML_LAUNCH((head_fast_kernel<LPS, 4, true>), gr, bl, 0, st, g, aux);
    else MML_LAUNCH((head_fast_kernel<LPS, 8, true>), gr, bl, 0, st, g, aux);
    return;
  }
  if (aux.nt == 2) MML_LAUNCH((head_fast_kernel<LPS, 2, false>), gr, bl, 0, st, g, aux);
  else if (aux.nt == 4) MML_LAUNCH((head_fast_kernel<LPS, 4, false>), gr, bl, 0, st, g, aux);
  else MML_LAUNCH((head_fast_kernel<LPS, 8, false>), gr, bl, 0, st, g, aux);
}

int head_fast(const mml_head_group* g, const HeadFastAux& aux, hipStream_t st) {
  if (aux.lps == 16) launch_head<16>(*g, aux, st);
  else if (aux.lps == 32) launch_head<32>(*g, aux, st);
  else launch_head<64>(*g, aux, st);
  return check_launch("mml_head(fast)");
}

}  // namespace mml
