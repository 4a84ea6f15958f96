// K8 optimizers (torch.optim.{SGD,Adam,Adagrad,RMSprop} with their defaults, as built by
// BaseModel._get_optim, model/basemodel.py:569-584, and stepped at :313) and the K6/K7 elementwise helpers for
// STAR (model/utils.py:214-218) and PepNet (model/pepnet.py:31-32, :72-78, :139-140).
// All of these are pure streaming kernels: 16-byte accesses per lane, grid-stride, HBM-bound.
#include "common.hpp"

#include <stdlib.h>

namespace mml {

struct OptLaunch {
  mml_opt_tensor t[MML_MAX_OPT_TENSORS];
  int32_t n;
  int32_t variant;  // streaming form of the vector loop (tuning knob MMLREC_OPT_VARIANT): bit 0 = nontemporal, bit 1 = 2x unroll; bit 2 = capped grid (4 chunks in flight per thread)
  mml_opt_hyper h;
  int64_t chunk0[MML_MAX_OPT_TENSORS + 1];  // flat kernel: first 4-element chunk of tensor i in the concatenation
  // streaming kernel over MANY tensors (round 6: every table of the model in one marked launch): a 1-D grid whose
  // workgroups are dealt to the tensors in proportion to their sizes; blk0[i] = first workgroup of tensor i (prop != 0)
  int32_t blk0[MML_MAX_OPT_TENSORS + 1];
  int32_t prop;
};

struct StepConsts {
  float step_size;  // Adam: lr / (1 - beta1^t)
  float inv_bc2s;   // Adam: 1 / sqrt(1 - beta2^t)
};

// Computed by ONE lane per workgroup (double-precision pow is hundreds of instructions) and broadcast through LDS.
__device__ __forceinline__ StepConsts step_consts(const mml_opt_hyper& h) {
  __shared__ StepConsts sc;
  if (threadIdx.x == 0) {
    StepConsts c{h.lr, 1.f};
    if (h.kind == MML_OPT_ADAM) {
      const int t = h.step_dev ? *h.step_dev : h.step;
      // torch computes the bias corrections in double precision on the host (python floats)
      const double bc1 = 1.0 - pow((double)h.beta1, (double)t);
      const double bc2 = 1.0 - pow((double)h.beta2, (double)t);
      c.step_size = (float)((double)h.lr / bc1);
      c.inv_bc2s = (float)(1.0 / sqrt(bc2));
    }
    sc = c;
  }
  __syncthreads();
  return sc;
}

// One element of torch.optim's single-tensor update (the _single_tensor_* functions with default flags).
__device__ __forceinline__ void opt_update(const mml_opt_hyper& h, const StepConsts& c, float& p, float g, float& s1,
                                           float& s2) {
  switch (h.kind) {
    case MML_OPT_SGD:
      p -= h.lr * g;
      break;
    case MML_OPT_ADAM: {
      s1 = s1 + (1.f - h.beta1) * (g - s1);               // exp_avg.lerp_(grad, 1-beta1)
      s2 = h.beta2 * s2 + (1.f - h.beta2) * g * g;        // exp_avg_sq.mul_(beta2).addcmul_(g, g, 1-beta2)
      const float denom = sqrtf(s2) * c.inv_bc2s + h.eps; // (sqrt(v) / sqrt(bc2)).add_(eps)
      p -= c.step_size * (s1 / denom);                    // param.addcdiv_(exp_avg, denom, value=-step_size)
      break;
    }
    case MML_OPT_ADAGRAD:
      s1 += g * g;                                        // state_sum.addcmul_(g, g)
      p -= h.lr * (g / (sqrtf(s1) + h.eps));              // param.addcdiv_(g, sqrt(sum)+eps, value=-lr)
      break;
    case MML_OPT_RMSPROP:
      s1 = h.alpha * s1 + (1.f - h.alpha) * g * g;        // square_avg.mul_(alpha).addcmul_(g, g, 1-alpha)
      p -= h.lr * (g / (sqrtf(s1) + h.eps));
      break;
  }
}

// Gradient of the regulariser folded into the update (model/basemodel.py:524-540): d/dp [l1 |p| + l2 p^2].
__device__ __forceinline__ float reg_grad(float g, float p, float l1, float l2) {
  if (l2 != 0.f) g += (2.f * l2) * p;
  if (l1 != 0.f) g += l1 * (p > 0.f ? 1.f : (p < 0.f ? -1.f : 0.f));
  return g;
}

// blockIdx.y = tensor, blockIdx.x strides over that tensor in float4 chunks.
// STREAM only names the launch (profilers see two symbols): true = a launch that streams >= 2^24 parameters through
// HBM (the dense table update), false = everything else (MLP parameters, small tables).  Same code.
// PATH = the form of the vector loop, one instantiation each so that each keeps ITS register count (all forms in one
// kernel: 170 VGPRs, two waves per SIMD -- and the plain loop, which needs 40, ran at that occupancy too):
//   0 plain grid-stride loop (also the remainder loop of every other form)   1 two chunks per iteration
//   2 split update, untouched rows, U chunks in flight per thread            3 marked gradients, U chunks in flight
enum { OPT_PLAIN = 0, OPT_UNROLL2 = 1, OPT_SKIP_U = 2, OPT_MARK_U = 3 };
template <bool STREAM, int PATH, int U>
__global__ __launch_bounds__(256) void opt_dense_kernel(const OptLaunch L) {
  // (workgroup-uniform: scalar loads and compares)
  int ti = blockIdx.y, nblk = gridDim.x, lblk = blockIdx.x;
  if (L.prop) {
    ti = 0;
    while (ti + 1 < L.n && (int)blockIdx.x >= L.blk0[ti + 1]) ++ti;
    nblk = L.blk0[ti + 1] - L.blk0[ti];
    lblk = (int)blockIdx.x - L.blk0[ti];
  }
  const mml_opt_tensor& T = L.t[ti];
  const mml_opt_hyper& h = L.h;
  const StepConsts c = step_consts(h);
  const int64_t n4 = T.n >> 2;
  const bool vec = aligned16(T.param) && aligned16(T.grad) && (!T.state1 || aligned16(T.state1)) &&
                   (!T.state2 || aligned16(T.state2));
  const int64_t stride = (int64_t)nblk * blockDim.x;
  const int64_t tid = (int64_t)lblk * blockDim.x + threadIdx.x;
  float* gw = const_cast<float*>(T.grad);
  const bool reg = T.l1 != 0.f || T.l2 != 0.f;
  if (vec) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    const bool nt = (L.variant & 1) != 0;
    f4* P = reinterpret_cast<f4*>(T.param);
    f4* G = reinterpret_cast<f4*>(gw);
    f4* S1 = reinterpret_cast<f4*>(T.state1);
    f4* S2 = reinterpret_cast<f4*>(T.state2);
    const f4 zero = {0.f, 0.f, 0.f, 0.f};
    auto ld = [&](const f4* q) { return nt ? __builtin_nontemporal_load(q) : *q; };
    auto st = [&](f4* q, const f4& v) {
      if (nt) __builtin_nontemporal_store(v, q);
      else *q = v;
    };
    auto one = [&](f4& p, const f4& g, f4& a, f4& b) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float pe = p[e], ae = a[e], be = b[e];
        opt_update(h, c, pe, reg ? reg_grad(g[e], pe, T.l1, T.l2) : g[e], ae, be);
        p[e] = pe;
        a[e] = ae;
        b[e] = be;
      }
    };
    const uint32_t* skip = T.skip_rows;
    const int re = T.row_elems > 0 ? T.row_elems : 1;
    // row of chunk c = elements [4c, 4c + 4) (inside ONE row when row_elems % 4 == 0): a shift for the usual power-of-two
    // embedding widths (a 64-bit division is ~100 VALU instructions per chunk, issue slots the GEMM waves beside this
    // kernel want)
    const int rsh = (re & (re - 1)) == 0 ? __builtin_ctz((unsigned)re) : -1;
    auto row_of = [&](int64_t c) { return rsh >= 0 ? ((c << 2) >> rsh) : ((c << 2) / re); };
    auto skipped = [&](int64_t c) {
      const int64_t row = row_of(c);
      return (skip[row >> 5] >> (row & 31)) & 1u;
    };
    const bool zg = T.zero_grads != 0;
    // marked gradients: a row whose byte is 0 has a zero gradient, not read.  The chunks of a row sit in adjacent lanes
    // of ONE wave (row_elems / 4 is a power of two <= 64, chunk index = lane + multiple of 64): every lane of the row
    // reads the byte in the same wave-instruction, before the first chunk's lane clears it further down.
    uint8_t* const gm = T.grad_marks;
    int64_t i = tid;
    if (PATH == OPT_UNROLL2 && !skip && !gm) {
      for (; i + stride < n4; i += 2 * stride) {  // eight 16-byte loads in flight per thread
        const int64_t j = i + stride;
        f4 p0 = ld(P + i), p1 = ld(P + j);
        const f4 g0 = ld(G + i), g1 = ld(G + j);
        f4 a0 = S1 ? ld(S1 + i) : zero, a1 = S1 ? ld(S1 + j) : zero;
        f4 b0 = S2 ? ld(S2 + i) : zero, b1 = S2 ? ld(S2 + j) : zero;
        one(p0, g0, a0, b0);
        one(p1, g1, a1, b1);
        st(P + i, p0);
        st(P + j, p1);
        if (S1) { st(S1 + i, a0); st(S1 + j, a1); }
        if (S2) { st(S2 + i, b0); st(S2 + j, b1); }
        if (h.zero_grad && (g0.x != 0.f || g0.y != 0.f || g0.z != 0.f || g0.w != 0.f)) G[i] = zero;
        if (h.zero_grad && (g1.x != 0.f || g1.y != 0.f || g1.z != 0.f || g1.w != 0.f)) G[j] = zero;
      }
    }
    if (PATH == OPT_SKIP_U && skip) {
      // early half of the split table update under a capped grid (mml_opt_hyper.max_blocks): it runs BESIDE other
      // kernels with few waves, so the memory-level parallelism has to come from the thread: U independent chunks (3U
      // 16-byte loads) in flight.  (With the full grid the plain loop below is faster: fewer registers, more waves.)
      for (; i + (U - 1) * stride < n4; i += U * stride) {
        f4 p[U], a[U], b[U];
        uint32_t w[U];
#pragma unroll
        for (int k = 0; k < U; ++k) {  // the U bitmap words first ...
          const int64_t row = row_of(i + k * stride);
          w[k] = (skip[row >> 5] >> (row & 31)) & 1u;
        }
#pragma unroll
        for (int k = 0; k < U; ++k) {  // ... then every load of the live chunks
          const int64_t j = i + k * stride;
          if (!w[k]) {
            p[k] = ld(P + j);
            a[k] = S1 ? ld(S1 + j) : zero;
            b[k] = S2 ? ld(S2 + j) : zero;
          }
        }
#pragma unroll
        for (int k = 0; k < U; ++k) {
          if (w[k]) continue;
          const int64_t j = i + k * stride;
          const f4 g = zg ? zero : ld(G + j);
          one(p[k], g, a[k], b[k]);
          st(P + j, p[k]);
          if (S1) st(S1 + j, a[k]);
          if (S2) st(S2 + j, b[k]);
          if (h.zero_grad && (g.x != 0.f || g.y != 0.f || g.z != 0.f || g.w != 0.f)) G[j] = zero;
        }
      }
    }
    if (PATH == OPT_MARK_U && gm && !skip) {
      // marked single-launch update under a capped grid (mml_opt_hyper.max_blocks: it runs beside the weight-gradient
      // GEMMs and leaves them their wave slots): the memory-level parallelism comes from the thread -- U chunks, 3U
      // 16-byte loads in flight
      for (; i + (U - 1) * stride < n4; i += U * stride) {
        f4 p[U], a[U], b[U], g[U];
        bool lv[U];
        int64_t rw[U];
#pragma unroll
        for (int k = 0; k < U; ++k) {
          rw[k] = row_of(i + k * stride);
          lv[k] = gm[rw[k]] != 0;
        }
#pragma unroll
        for (int k = 0; k < U; ++k) {
          const int64_t j = i + k * stride;
          p[k] = ld(P + j);
          a[k] = S1 ? ld(S1 + j) : zero;
          b[k] = S2 ? ld(S2 + j) : zero;
          g[k] = (zg || !lv[k]) ? zero : ld(G + j);
        }
#pragma unroll
        for (int k = 0; k < U; ++k) {
          const int64_t j = i + k * stride;
          one(p[k], g[k], a[k], b[k]);
          st(P + j, p[k]);
          if (S1) st(S1 + j, a[k]);
          if (S2) st(S2 + j, b[k]);
          if (h.zero_grad && (g[k].x != 0.f || g[k].y != 0.f || g[k].z != 0.f || g[k].w != 0.f)) G[j] = zero;
          if (lv[k] && (j << 2) == rw[k] * re) gm[rw[k]] = 0;
        }
      }
    }
    for (; i < n4; i += stride) {
      if (skip && skipped(i)) continue;
      bool live = true;
      int64_t row = 0;
      if (gm) {
        row = row_of(i);
        live = gm[row] != 0;
      }
      f4 p = ld(P + i);
      const f4 g = (zg || !live) ? zero : ld(G + i);
      f4 a = S1 ? ld(S1 + i) : zero;
      f4 b = S2 ? ld(S2 + i) : zero;
      one(p, g, a, b);
      st(P + i, p);
      if (S1) st(S1 + i, a);
      if (S2) st(S2 + i, b);
      // re-zero only what the scatter touched (~1 % of the rows): saves the 4 B/element store of a blind memset
      if (h.zero_grad && (g.x != 0.f || g.y != 0.f || g.z != 0.f || g.w != 0.f)) G[i] = zero;
      if (gm && live && (i << 2) == row * re) gm[row] = 0;
    }
  }
  const int64_t tail0 = vec ? (n4 << 2) : 0;
  for (int64_t i = tail0 + tid; i < T.n; i += stride) {
    if (T.skip_rows) {
      const int64_t row = i / (T.row_elems > 0 ? T.row_elems : 1);
      if ((T.skip_rows[row >> 5] >> (row & 31)) & 1u) continue;
    }
    float p = T.param[i], a = T.state1 ? T.state1[i] : 0.f, b = T.state2 ? T.state2[i] : 0.f;
    opt_update(h, c, p, reg_grad(T.zero_grads ? 0.f : T.grad[i], p, T.l1, T.l2), a, b);
    T.param[i] = p;
    if (T.state1) T.state1[i] = a;
    if (T.state2) T.state2[i] = b;
    if (h.zero_grad && T.grad[i] != 0.f) gw[i] = 0.f;
  }
}

// Many tensors of very different sizes (MLP weights and biases, the small tables) in ONE balanced launch: the tensors
// are concatenated in units of 4 elements and a thread strides over that index space; the tensor of a chunk is found
// by bisection of the prefix table (kept in LDS with the descriptors: a per-lane index into the kernel-argument block
// would send it to scratch).
__global__ __launch_bounds__(256) void opt_flat_kernel(const OptLaunch L) {
  __shared__ mml_opt_tensor ts[MML_MAX_OPT_TENSORS];
  __shared__ int64_t pre[MML_MAX_OPT_TENSORS + 1];
  for (int i = threadIdx.x; i < L.n; i += 256) ts[i] = L.t[i];
  for (int i = threadIdx.x; i <= L.n; i += 256) pre[i] = L.chunk0[i];
  const mml_opt_hyper& h = L.h;
  const StepConsts c = step_consts(h);  // (contains the __syncthreads that also publishes ts / pre)
  const int64_t total = pre[L.n];
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t ch = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; ch < total; ch += stride) {
    int lo = 0, hi = L.n;  // largest t with pre[t] <= ch
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (pre[mid] <= ch) lo = mid;
      else hi = mid;
    }
    const mml_opt_tensor T = ts[lo];
    const int64_t e0 = (ch - pre[lo]) << 2;
    float* gw = const_cast<float*>(T.grad);
    const int re = T.row_elems > 0 ? T.row_elems : 1;
    if (T.skip_rows && re % 4 == 0) {  // chunk inside one row
      const int64_t row = e0 / re;
      if ((T.skip_rows[row >> 5] >> (row & 31)) & 1u) continue;
    }
    const bool vec = (e0 + 4 <= T.n) && aligned16(T.param) && aligned16(T.grad) && (!T.state1 || aligned16(T.state1)) &&
                     (!T.state2 || aligned16(T.state2));
    if (vec) {
      float4 p = *reinterpret_cast<float4*>(T.param + e0);
      const float4 g = T.zero_grads ? make_float4(0, 0, 0, 0) : *reinterpret_cast<const float4*>(T.grad + e0);
      float4 a = T.state1 ? *reinterpret_cast<float4*>(T.state1 + e0) : make_float4(0, 0, 0, 0);
      float4 b = T.state2 ? *reinterpret_cast<float4*>(T.state2 + e0) : make_float4(0, 0, 0, 0);
      opt_update(h, c, p.x, reg_grad(g.x, p.x, T.l1, T.l2), a.x, b.x);
      opt_update(h, c, p.y, reg_grad(g.y, p.y, T.l1, T.l2), a.y, b.y);
      opt_update(h, c, p.z, reg_grad(g.z, p.z, T.l1, T.l2), a.z, b.z);
      opt_update(h, c, p.w, reg_grad(g.w, p.w, T.l1, T.l2), a.w, b.w);
      *reinterpret_cast<float4*>(T.param + e0) = p;
      if (T.state1) *reinterpret_cast<float4*>(T.state1 + e0) = a;
      if (T.state2) *reinterpret_cast<float4*>(T.state2 + e0) = b;
      if (h.zero_grad && (g.x != 0.f || g.y != 0.f || g.z != 0.f || g.w != 0.f))
        *reinterpret_cast<float4*>(gw + e0) = make_float4(0, 0, 0, 0);
    } else {
      for (int64_t i = e0; i < e0 + 4 && i < T.n; ++i) {
        if (T.skip_rows) {
          const int64_t row = i / re;
          if ((T.skip_rows[row >> 5] >> (row & 31)) & 1u) continue;
        }
        float p = T.param[i], a = T.state1 ? T.state1[i] : 0.f, b = T.state2 ? T.state2[i] : 0.f;
        const float g = T.zero_grads ? 0.f : T.grad[i];
        opt_update(h, c, p, reg_grad(g, p, T.l1, T.l2), a, b);
        T.param[i] = p;
        if (T.state1) T.state1[i] = a;
        if (T.state2) T.state2[i] = b;
        if (h.zero_grad && g != 0.f) gw[i] = 0.f;
      }
    }
  }
}

// sparse rows: one E-float row per group of lanes, rows taken from the touched list
struct RowsLaunch {
  float* tab[MML_MAX_FIELDS];
  float* grad[MML_MAX_FIELDS];
  float* s1[MML_MAX_FIELDS];
  float* s2[MML_MAX_FIELDS];
  uint32_t* seen[MML_MAX_FIELDS];
  int64_t rowbase[MML_MAX_FIELDS + 1];
  int32_t F, E;
  int32_t* last[MML_MAX_FIELDS];  // lazy-exact mode: per-row "current as of step" words (or all null)
  const int32_t* touched;
  const int32_t* touched_count;
  int32_t cap;
  int32_t pad_;
  mml_opt_hyper h;
};

// field of a global row id: bisection of an LDS copy of rowbase (a per-lane scan of the kernel-argument array is a
// chain of up to F dependent loads: it was most of the touched-row kernels' time)
__device__ __forceinline__ int field_of_row(const int64_t* rb_lds, int F, int64_t grow) {
  int lo = 0, hi = F;  // largest f with rb[f] <= grow
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (rb_lds[mid] <= grow) lo = mid;
    else hi = mid;
  }
  return lo;
}

// One lane = one 16-byte piece of a touched row (E % 4 == 0; else one float): four independent element updates per
// lane, and the row / field decode once per piece instead of once per float.
template <int VEC>
__global__ __launch_bounds__(256) void opt_rows_kernel(const RowsLaunch L) {
  const mml_opt_hyper& h = L.h;
  // per-field descriptors in LDS: a per-lane index into the kernel-argument arrays is a memory load per use
  __shared__ int64_t rb_lds[MML_MAX_FIELDS + 1];
  __shared__ float* tab_l[MML_MAX_FIELDS];
  __shared__ float* grad_l[MML_MAX_FIELDS];
  __shared__ float* s1_l[MML_MAX_FIELDS];
  __shared__ float* s2_l[MML_MAX_FIELDS];
  __shared__ uint32_t* seen_l[MML_MAX_FIELDS];
  __shared__ int32_t* last_l[MML_MAX_FIELDS];
  for (int i = threadIdx.x; i <= L.F; i += 256) rb_lds[i] = L.rowbase[i];
  for (int i = threadIdx.x; i < L.F; i += 256) {
    tab_l[i] = L.tab[i]; grad_l[i] = L.grad[i]; s1_l[i] = L.s1[i]; s2_l[i] = L.s2[i]; seen_l[i] = L.seen[i];
    last_l[i] = L.last[i];
  }
  const StepConsts c = step_consts(h);  // (contains the __syncthreads that publishes rb_lds)
  int32_t cnt = *L.touched_count;
  if (cnt > L.cap) cnt = L.cap;
  const int per_row = L.E / VEC;
  const int64_t total = (int64_t)cnt * per_row;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t item = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; item < total; item += stride) {
    const int32_t li = (int32_t)(item / per_row);
    const int e = (int)(item - (int64_t)li * per_row) * VEC;
    const int64_t grow = L.touched[li];
    const int f = field_of_row(rb_lds, L.F, grow);
    const int64_t row = grow - rb_lds[f];
    const int64_t o = row * L.E + e;
    float* const T_ = tab_l[f];
    float* const G_ = grad_l[f];
    float* const S1 = s1_l[f];
    float* const S2 = s2_l[f];
    float p[VEC], g[VEC], a[VEC], b[VEC];
    if (VEC == 4) {
      *reinterpret_cast<float4*>(p) = *reinterpret_cast<const float4*>(T_ + o);
      *reinterpret_cast<float4*>(g) = *reinterpret_cast<const float4*>(G_ + o);
      if (S1) *reinterpret_cast<float4*>(a) = *reinterpret_cast<const float4*>(S1 + o);
      if (S2) *reinterpret_cast<float4*>(b) = *reinterpret_cast<const float4*>(S2 + o);
    } else {
      p[0] = T_[o];
      g[0] = G_[o];
      if (S1) a[0] = S1[o];
      if (S2) b[0] = S2[o];
    }
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
      if (!S1) a[k] = 0.f;
      if (!S2) b[k] = 0.f;
      opt_update(h, c, p[k], g[k], a[k], b[k]);
    }
    if (VEC == 4) {
      *reinterpret_cast<float4*>(T_ + o) = *reinterpret_cast<float4*>(p);
      if (S1) *reinterpret_cast<float4*>(S1 + o) = *reinterpret_cast<float4*>(a);
      if (S2) *reinterpret_cast<float4*>(S2 + o) = *reinterpret_cast<float4*>(b);
      *reinterpret_cast<float4*>(G_ + o) = make_float4(0.f, 0.f, 0.f, 0.f);
    } else {
      T_[o] = p[0];
      if (S1) S1[o] = a[0];
      if (S2) S2[o] = b[0];
      G_[o] = 0.f;
    }
    if (e == 0) {
      atomicAnd(seen_l[f] + (row >> 5), ~(1u << (row & 31)));
      if (last_l[f]) last_l[f][row] = h.step_dev ? *h.step_dev : h.step;
    }
  }
}

// ---- lazy-exact catch-up ---------------------------------------------------------------------------
// Replays the zero-gradient steps (from, to] of the dense optimizer for one element, with the per-step arithmetic of
// opt_update (g = 0).  Bias corrections follow torch (double precision powers), advanced by recurrence.
__device__ __forceinline__ void catchup_element(const mml_opt_hyper& h, int from, int to, float& p, float& m, float& v) {
  if (to <= from) return;
  if (h.kind == MML_OPT_ADAM) {
    if (m == 0.f) {  // never had a gradient (or fully decayed): p cannot move, v only decays
      v *= powf(h.beta2, (float)(to - from));
      return;
    }
    // bias-correction powers by float recurrence (one powf to start): ~1e-7 relative on the step size, far below
    // the 1e-4 parity budget, and an order of magnitude cheaper per replayed step than double-precision pow/div/sqrt
    float b1p = powf(h.beta1, (float)from), b2p = powf(h.beta2, (float)from);
    int j = from;
    while (j < to) {
      ++j;
      b1p *= h.beta1;
      b2p *= h.beta2;
      StepConsts c;
      c.step_size = h.lr / (1.f - b1p);
      c.inv_bc2s = rsqrtf(1.f - b2p);
      const float before = p;
      opt_update(h, c, p, 0.f, m, v);
      // the per-step move shrinks by ~0.9 per step once the bias-correction growth has died out (j > 16): when it
      // no longer changes p in fp32 it never will again, and only the moments keep decaying
      if (p == before && j - from > 16) {
        const float rem = (float)(to - j);
        m *= powf(h.beta1, rem);
        v *= powf(h.beta2, rem);
        return;
      }
    }
  } else if (h.kind == MML_OPT_RMSPROP) {
    m *= powf(h.alpha, (float)(to - from));  // `m` carries state1 = square_avg: p does not move, the average decays
  }
  // SGD / Adagrad: a zero gradient changes nothing
}

struct CatchupLaunch {
  float* tab[MML_MAX_FIELDS];
  float* s1[MML_MAX_FIELDS];
  float* s2[MML_MAX_FIELDS];
  int32_t* last[MML_MAX_FIELDS];
  int64_t rowbase[MML_MAX_FIELDS + 1];
  int32_t F, E;
  const int32_t* touched;
  const int32_t* touched_count;
  int32_t cap;
  int32_t dense;      // 1: every row of table 0 (V rows), no list
  int64_t V;
  mml_opt_hyper h;
};

__global__ __launch_bounds__(256) void opt_catchup_kernel(const CatchupLaunch L) {
  __shared__ int64_t rb_lds[MML_MAX_FIELDS + 1];
  if (!L.dense) {
    for (int i = threadIdx.x; i <= L.F; i += 256) rb_lds[i] = L.rowbase[i];
    __syncthreads();
  }
  const mml_opt_hyper& h = L.h;
  const int target = (h.step_dev ? *h.step_dev : h.step) - (L.dense ? 0 : 1);
  int64_t nrows;
  if (L.dense) {
    nrows = L.V;
  } else {
    int32_t cnt = *L.touched_count;
    nrows = cnt > L.cap ? L.cap : cnt;
  }
  const int64_t total = nrows * L.E;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t item = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; item < total; item += stride) {
    const int64_t li = item / L.E;
    const int e = (int)(item - li * L.E);
    int f = 0;
    int64_t row = li;
    if (!L.dense) {
      const int64_t grow = L.touched[li];
      f = field_of_row(rb_lds, L.F, grow);
      row = grow - rb_lds[f];
    }
    const int from = L.last[f][row];
    if (from >= target) continue;
    const int64_t o = row * L.E + e;
    float p = L.tab[f][o];
    float m = L.s1[f] ? L.s1[f][o] : 0.f, v = L.s2[f] ? L.s2[f][o] : 0.f;
    catchup_element(h, from, target, p, m, v);
    L.tab[f][o] = p;
    if (L.s1[f]) L.s1[f][o] = m;
    if (L.s2[f]) L.s2[f][o] = v;
    // every lane of the row has read `from` above (same wave, in order), so lane 0 may now publish the new step
    __builtin_amdgcn_wave_barrier();
    if (e == 0) L.last[f][row] = target;
  }
}

__global__ void counter_kernel(int32_t* c, int32_t delta, int32_t reset) {
  if (threadIdx.x == 0 && blockIdx.x == 0) *c = reset ? 0 : (*c + delta);
}

// ---- elementwise -------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ew_mul_kernel(const float* a, const float* b, float* out, int64_t n) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = a[i] * b[i];
}

// act_a / act_b: the operand is the OUTPUT of that activation and this product is its only consumer, so the
// activation's derivative (from the output value, like mml_act_bwd) is folded into the gradient written here
__device__ __forceinline__ float act_deriv_from_output(float y, int act) {
  if (act == MML_ACT_RELU) return y > 0.f ? 1.f : 0.f;
  if (act == MML_ACT_SIGMOID) return y * (1.f - y);
  if (act == MML_ACT_SIGMOID2) return y * (1.f - 0.5f * y);
  return 1.f;
}
__global__ __launch_bounds__(256) void ew_mul_bwd_kernel(const float* dout, const float* a, const float* b, float* da,
                                                         float* db, int acc_a, int acc_b, int64_t n, int act_a,
                                                         int act_b) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const float d = dout[i];
    const float av = (db || act_a) ? a[i] : 0.f;
    const float bv = (da || act_b) ? b[i] : 0.f;
    if (da) {
      const float v = d * bv * act_deriv_from_output(av, act_a);
      da[i] = acc_a ? da[i] + v : v;
    }
    if (db) {
      const float v = d * av * act_deriv_from_output(bv, act_b);
      db[i] = acc_b ? db[i] + v : v;
    }
  }
}

struct AddN {
  const float* in[16];
  int32_t n_in;
};
__global__ __launch_bounds__(256) void ew_add_n_kernel(const AddN A, float* out, int64_t n) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    float s = 0.f;
    for (int k = 0; k < A.n_in; ++k) s += A.in[k][i];
    out[i] = s;
  }
}

// n independent items in ONE launch: out[i] (+)= sum_k x_k[i] * (y_k ? y_k[i] : 1).  STAR's derived parameters
// (W_spec * W_shared, b_spec + b_shared for every head and layer, model/utils.py:214-216) and their gradients were
// ~60 launches of a few KB each per step; they are two launches each way now.
struct SumProdBatch {
  mml_sumprod_desc d[MML_SUMPROD_BATCH];
};
// (act_deriv_from_output is defined further up with the mul_bwd kernel; deriv_of = the OUTPUT of the activation whose
// derivative multiplies the sum -- PepNet's gate products, where the factor is 2*sigmoid(.) or relu(.) and this product
// its only consumer)
__global__ __launch_bounds__(256) void sumprod_batch_kernel(const SumProdBatch Bt) {
  const mml_sumprod_desc& D = Bt.d[blockIdx.y];
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  bool vec = (D.n & 3) == 0 && aligned16(D.out) && (!D.deriv_of || aligned16(D.deriv_of));
  for (int k = 0; k < D.n_terms; ++k) vec = vec && aligned16(D.x[k]) && (!D.y[k] || aligned16(D.y[k]));
  float am = 0.f;  // operand magnitude of what this thread stores (D.amax_out)
  if (vec) {  // 16 bytes per lane and operand
    const int64_t n4 = D.n >> 2;
    for (int64_t i = tid; i < n4; i += stride) {
      float4 s = D.accumulate ? reinterpret_cast<const float4*>(D.out)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
      for (int k = 0; k < D.n_terms; ++k) {
        const float4 x = reinterpret_cast<const float4*>(D.x[k])[i];
        if (D.y[k]) {
          const float4 y = reinterpret_cast<const float4*>(D.y[k])[i];
          s.x += x.x * y.x; s.y += x.y * y.y; s.z += x.z * y.z; s.w += x.w * y.w;
        } else {
          s.x += x.x; s.y += x.y; s.z += x.z; s.w += x.w;
        }
      }
      if (D.act != MML_ACT_NONE) {
        const float4 o = reinterpret_cast<const float4*>(D.deriv_of)[i];
        s.x *= act_deriv_from_output(o.x, D.act); s.y *= act_deriv_from_output(o.y, D.act);
        s.z *= act_deriv_from_output(o.z, D.act); s.w *= act_deriv_from_output(o.w, D.act);
      }
      reinterpret_cast<float4*>(D.out)[i] = s;
      amax_acc(am, s);
    }
    amax_flush(am, D.amax_out);
    return;
  }
  for (int64_t i = tid; i < D.n; i += stride) {
    float s = D.accumulate ? D.out[i] : 0.f;
    for (int k = 0; k < D.n_terms; ++k) s += D.x[k][i] * (D.y[k] ? D.y[k][i] : 1.f);
    if (D.act != MML_ACT_NONE) s *= act_deriv_from_output(D.deriv_of[i], D.act);
    D.out[i] = s;
    amax_acc(am, s);
  }
  amax_flush(am, D.amax_out);
}

__global__ __launch_bounds__(256) void copy2d_kernel(const float* src, int64_t lds_, float* dst, int64_t ldd, int64_t rows,
                                                     int cols, int accumulate) {
  const int64_t total = rows * cols;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int64_t r = i / cols;
    const int c = (int)(i - r * cols);
    const float v = src[r * lds_ + c];
    float* d = dst + r * ldd + c;
    *d = accumulate ? *d + v : v;
  }
}

constexpr int COPY2D_BATCH = 32;
struct Copy2dBatch {
  mml_copy2d_desc d[COPY2D_BATCH];
};
// blockIdx.y = item; blockIdx.x strides over that item's elements.  Round 6: 16-byte pieces, four in flight per thread, where
// the item allows it (PepNet's [65 536, 64] concat blocks: one scalar element and a 64-bit division per trip before), and
// an item that raises a magnitude slot is walked by at most 256 workgroups -- every workgroup ends with a look at (and
// possibly an atomic on) the slot's one line, and 2 048 of them finishing together cost more than the magnitude pass the
// slot replaces (measured: +25 us per copy launch).
__global__ __launch_bounds__(256) void copy2d_batch_kernel(const Copy2dBatch Bt) {
  const mml_copy2d_desc& D = Bt.d[blockIdx.y];
  const int nb = D.amax_out ? ((int)gridDim.x < 256 ? (int)gridDim.x : 256) : (int)gridDim.x;
  const bool mine = (int)blockIdx.x < nb;  // (uniform)
  float amf = 0.f;
  if (mine) {
    const int64_t stride = (int64_t)nb * 256;
    const int64_t tid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool vec = D.cols % 4 == 0 && D.lds % 4 == 0 && D.ldd % 4 == 0 && aligned16(D.src) && aligned16(D.dst);
    if (vec) {
      const int c4 = D.cols >> 2;
      const int64_t total = D.rows * c4;
      const bool small = total < 0x7fffffff;
      auto off = [&](int64_t i, int64_t& so, int64_t& dof) __attribute__((always_inline)) {
        const int64_t r = small ? (int64_t)((uint32_t)i / (uint32_t)c4) : i / c4;
        const int64_t c = (i - r * c4) << 2;
        so = r * D.lds + c;
        dof = r * D.ldd + c;
      };
      auto fin = [&](float4 v, int64_t dof) __attribute__((always_inline)) {
        float4* q = reinterpret_cast<float4*>(D.dst + dof);
        if (D.accumulate) {
          const float4 o = *q;
          v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
        }
        *q = v;
        amax_acc(amf, v);
      };
      int64_t i = tid;
      for (; i + 3 * stride < total; i += 4 * stride) {
        int64_t s0, s1, s2, s3, d0, d1, d2, d3;
        off(i, s0, d0); off(i + stride, s1, d1); off(i + 2 * stride, s2, d2); off(i + 3 * stride, s3, d3);
        const float4 v0 = *reinterpret_cast<const float4*>(D.src + s0), v1 = *reinterpret_cast<const float4*>(D.src + s1);
        const float4 v2 = *reinterpret_cast<const float4*>(D.src + s2), v3 = *reinterpret_cast<const float4*>(D.src + s3);
        fin(v0, d0); fin(v1, d1); fin(v2, d2); fin(v3, d3);
      }
      for (; i < total; i += stride) {
        int64_t s0, d0;
        off(i, s0, d0);
        fin(*reinterpret_cast<const float4*>(D.src + s0), d0);
      }
    } else {
      const int64_t total = D.rows * D.cols;
      for (int64_t i = tid; i < total; i += stride) {
        const int64_t r = i / D.cols;
        const int c = (int)(i - r * D.cols);
        float v = D.src[r * D.lds + c];
        float* q = D.dst + r * D.ldd + c;
        if (D.accumulate) v += *q;
        *q = v;
        amax_acc(amf, v);
      }
    }
  }
  amax_flush<true>(amf, mine ? D.amax_out : nullptr);  // (uniform per workgroup; a null slot returns at once)
}

struct ColSegs {
  const float* src[MML_MAX_FIELDS];
  float* dst[MML_MAX_FIELDS];
  int64_t lds_[MML_MAX_FIELDS], ldd[MML_MAX_FIELDS];
  int32_t width[MML_MAX_FIELDS];
  int32_t start[MML_MAX_FIELDS + 1];  // prefix sum of widths
  int32_t n;
};
// One thread per (row, packed column): consecutive lanes walk the packed columns of one row, so each segment is
// read and written in runs of `width` contiguous floats.
__global__ __launch_bounds__(256) void copy_cols_kernel(const ColSegs S, int64_t rows, int accumulate) {
  const int W = S.start[S.n];
  const int64_t total = rows * W;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int64_t r = i / W;
    const int c = (int)(i - r * W);
    int s = 0;
    while (c >= S.start[s + 1]) ++s;
    const int cc = c - S.start[s];
    const float v = S.src[s][r * S.lds_[s] + cc];
    float* d = S.dst[s] + r * S.ldd[s] + cc;
    *d = accumulate ? *d + v : v;
  }
}

// The same copy in 16-byte pieces (every segment start, width, leading dimension and base pointer a multiple of four
// floats -- the row / row-gradient exchange buffers with E % 4 == 0): segment tables and the column -> segment map are
// staged in LDS once per workgroup, so the inner loop is one LDS lookup + one float4 load + one float4 store.
constexpr int COPY_COLS_MAX_W4 = 1024;
__global__ __launch_bounds__(256) void copy_cols_vec4_kernel(const ColSegs S, int64_t rows, int accumulate) {
  typedef float f4 __attribute__((ext_vector_type(4)));
  __shared__ const float* s_src[MML_MAX_FIELDS];
  __shared__ float* s_dst[MML_MAX_FIELDS];
  __shared__ int64_t s_lds[MML_MAX_FIELDS], s_ldd[MML_MAX_FIELDS];
  __shared__ int32_t s_start[MML_MAX_FIELDS + 1];
  __shared__ uint8_t seg_of[COPY_COLS_MAX_W4];
  const int W4 = S.start[S.n] >> 2;
  for (int k = threadIdx.x; k < S.n; k += 256) {
    s_src[k] = S.src[k]; s_dst[k] = S.dst[k]; s_lds[k] = S.lds_[k]; s_ldd[k] = S.ldd[k];
  }
  for (int k = threadIdx.x; k <= S.n; k += 256) s_start[k] = S.start[k];
  __syncthreads();
  for (int c = threadIdx.x; c < W4; c += 256) {
    int lo = 0, hi = S.n - 1;  // last segment whose start <= 4c
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (s_start[mid] <= 4 * c) lo = mid; else hi = mid - 1;
    }
    seg_of[c] = (uint8_t)lo;
  }
  __syncthreads();
  const int64_t total = rows * W4;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    int64_t r;
    int c;
    if (total < 0x7fffffff) {  // 32-bit division when it fits
      const uint32_t ri = (uint32_t)i / (uint32_t)W4;
      r = ri;
      c = (int)((uint32_t)i - ri * (uint32_t)W4);
    } else {
      r = i / W4;
      c = (int)(i - r * W4);
    }
    const int sg = seg_of[c];
    const int cc = 4 * c - s_start[sg];
    const f4 v = *reinterpret_cast<const f4*>(s_src[sg] + r * s_lds[sg] + cc);
    f4* d = reinterpret_cast<f4*>(s_dst[sg] + r * s_ldd[sg] + cc);
    *d = accumulate ? *d + v : v;
  }
}

__global__ __launch_bounds__(256) void act_bwd_kernel(const float* y, const float* dy, float* dst, int64_t n, int act) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const float v = y[i];
    float d = 1.f;
    if (act == MML_ACT_RELU) d = v > 0.f ? 1.f : 0.f;
    else if (act == MML_ACT_SIGMOID) d = v * (1.f - v);
    else if (act == MML_ACT_SIGMOID2) d = v * (1.f - 0.5f * v);  // y = 2s: dy/dx = 2 s (1-s) = y (1 - y/2)
    dst[i] = d * dy[i];
  }
}

static unsigned ew_grid(int64_t n) {
  int64_t b = cdiv(n, 256);
  if (b > 256 * 8) b = 256 * 8;
  if (b < 1) b = 1;
  return (unsigned)b;
}

}  // namespace mml

using namespace mml;

static int check_hyper(const mml_opt_hyper* h, const char* who) {
  MML_REQUIRE(h, "%s: null hyper", who);
  MML_REQUIRE(h->kind >= MML_OPT_SGD && h->kind <= MML_OPT_RMSPROP, "%s: unknown optimizer kind %d", who, h->kind);
  MML_REQUIRE(h->kind != MML_OPT_ADAM || h->step_dev || h->step >= 1, "%s: Adam needs step >= 1", who);
  return MML_OK;
}

extern "C" int mml_opt_step_dense(const mml_opt_tensor* tensors, int32_t n, const mml_opt_hyper* hyper,
                                  mml_stream_t stream) {
  int rc = check_hyper(hyper, "mml_opt_step_dense");
  if (rc) return rc;
  MML_REQUIRE(n >= 0 && (n == 0 || tensors), "mml_opt_step_dense: bad tensor array");
  int i = 0;
  while (i < n) {
    OptLaunch L{};
    L.h = *hyper;
    int64_t total = 0, chunks = 0;
    while (i < n && L.n < MML_MAX_OPT_TENSORS) {
      const mml_opt_tensor& t = tensors[i];
      MML_REQUIRE(t.param && t.grad && t.n >= 0, "mml_opt_step_dense: tensor %d malformed", i);
      MML_REQUIRE(hyper->kind == MML_OPT_SGD || t.state1, "mml_opt_step_dense: tensor %d needs state1", i);
      MML_REQUIRE(hyper->kind != MML_OPT_ADAM || t.state2, "mml_opt_step_dense: tensor %d needs state2 (Adam)", i);
      // (both kernels decide per 16-byte chunk by the row of its first element: a chunk must not straddle two rows)
      MML_REQUIRE(!t.skip_rows || (t.row_elems > 0 && t.row_elems % 4 == 0 && t.n % t.row_elems == 0 &&
                                   aligned16(t.param)),
                  "mml_opt_step_dense: tensor %d: skip_rows needs row_elems %% 4 == 0 dividing n and a 16-byte aligned "
                  "parameter (16-byte chunks must lie inside one row)", i);
      if (t.grad_marks) {
        const int cpr = t.row_elems / 4;
        MML_REQUIRE(t.row_elems > 0 && t.row_elems % 4 == 0 && cpr <= 64 && (cpr & (cpr - 1)) == 0 &&
                        t.n % t.row_elems == 0 && !t.skip_rows && aligned16(t.param) && aligned16(t.grad) &&
                        (!t.state1 || aligned16(t.state1)) && (!t.state2 || aligned16(t.state2)),
                    "mml_opt_step_dense: tensor %d: grad_marks needs 16-byte aligned [rows, row_elems] tensors with "
                    "row_elems / 4 a power of two <= 64 and no skip_rows", i);
      }
      L.chunk0[L.n] = chunks;
      L.t[L.n++] = t;
      total += t.n;
      chunks += cdiv(t.n, 4);
      ++i;
    }
    L.chunk0[L.n] = chunks;
    if (total == 0) continue;
    static int variant = -1;
    if (variant < 0) {
      const char* e = getenv("MMLREC_OPT_VARIANT");
      variant = e ? atoi(e) : 0;
    }
    L.variant = variant | (hyper->max_blocks > 0 ? 4 : 0);
    // A group dominated by one huge tensor (the dense table update) streams with the per-tensor kernel (blockIdx.y =
    // tensor: no index search on the 2.7 GB stream); everything else (dozens of tensors from 64 B to a few MB) goes
    // through the flat kernel, where every workgroup has the same amount of work.
    int64_t nmax = 0;
    for (int k = 0; k < L.n; ++k) nmax = L.t[k].n > nmax ? L.t[k].n : nmax;
    // the loop form (see opt_dense_kernel): decided for the launch, so every tensor of it must qualify
    bool all_gm = true, all_skip = true, none = true;
    for (int k = 0; k < L.n; ++k) {
      all_gm = all_gm && L.t[k].grad_marks && !L.t[k].skip_rows;
      all_skip = all_skip && L.t[k].skip_rows;
      none = none && !L.t[k].grad_marks && !L.t[k].skip_rows;
    }
    // Round 6: a model's tables in ONE marked streaming launch (AE-30: 4 huge + 26 small; the small ones were a second
    // launch of the flat kernel, 31-34 us behind the stream): workgroups dealt in proportion to the tensors' sizes.
    const bool many = L.n > 4 && all_gm;
    if (total >= ((int64_t)1 << 24) && (L.n <= 4 || many)) {
      int64_t bx = cdiv(cdiv(nmax, 4), 256);
      if (bx > 256 * 8) bx = 256 * 8;
      if (hyper->max_blocks > 0 && bx * L.n > hyper->max_blocks) bx = cdiv(hyper->max_blocks, L.n);
      dim3 grid((unsigned)bx, (unsigned)L.n);
      if (many) {
        // every tensor gets what a launch of its own would give it (one workgroup per 256 chunks, at most 2 048): the huge
        // tables stream exactly as in their four-tensor launch, the small ones add a few hundred workgroups (the caller
        // lists them FIRST: they start with the launch instead of trailing it).  Under a workgroup cap: in proportion.
        int64_t want = 0;
        for (int k = 0; k < L.n; ++k) {
          const int64_t need = cdiv(cdiv(L.t[k].n, 4), 256);
          want += need > 256 * 8 ? 256 * 8 : (need < 1 ? 1 : need);
        }
        const bool capped = hyper->max_blocks > 0 && want > hyper->max_blocks;
        int64_t tb = capped ? hyper->max_blocks : want;
        if (tb < L.n) tb = L.n;
        int64_t at = 0;
        for (int k = 0; k < L.n; ++k) {
          L.blk0[k] = (int32_t)at;
          const int64_t need = cdiv(cdiv(L.t[k].n, 4), 256);
          int64_t nb = capped ? tb * L.t[k].n / total : (need > 256 * 8 ? 256 * 8 : need);
          if (nb > need) nb = need;
          if (nb < 1) nb = 1;
          at += nb;
        }
        L.blk0[L.n] = (int32_t)at;
        L.prop = 1;
        grid = dim3((unsigned)at, 1);
      }
      static int u_mark = -1;
      if (u_mark < 0) {
        const char* e = getenv("MMLREC_OPT_U");  // lab knob: chunks in flight per thread of the marked form (2, 4, 8)
        u_mark = e ? atoi(e) : 2;
      }
      hipStream_t st = to_stream(stream);
      if ((L.variant & 4) && all_gm) {
        if (u_mark == 4) MML_LAUNCH((opt_dense_kernel<true, OPT_MARK_U, 4>), grid, dim3(256), 0, st, L);
        else if (u_mark == 8) MML_LAUNCH((opt_dense_kernel<true, OPT_MARK_U, 8>), grid, dim3(256), 0, st, L);
        else MML_LAUNCH((opt_dense_kernel<true, OPT_MARK_U, 2>), grid, dim3(256), 0, st, L);
      } else if ((L.variant & 4) && all_skip) {
        MML_LAUNCH((opt_dense_kernel<true, OPT_SKIP_U, 4>), grid, dim3(256), 0, st, L);
      } else if ((L.variant & 2) && none) {
        MML_LAUNCH((opt_dense_kernel<true, OPT_UNROLL2, 1>), grid, dim3(256), 0, st, L);
      } else {
        MML_LAUNCH((opt_dense_kernel<true, OPT_PLAIN, 1>), grid, dim3(256), 0, st, L);
      }
    } else {
      for (int k = 0; k < L.n; ++k)
        MML_REQUIRE(!L.t[k].grad_marks, "mml_opt_step_dense: grad_marks is only honoured by the streaming launch "
                    "(>= 2^24 parameters in <= 4 tensors)");
      int64_t bx = cdiv(chunks, 256);
      if (bx > 256 * 8) bx = 256 * 8;
      if (hyper->max_blocks > 0 && bx > hyper->max_blocks) bx = hyper->max_blocks;
      MML_LAUNCH(opt_flat_kernel, dim3((unsigned)bx), dim3(256), 0, to_stream(stream), L);
    }
    rc = check_launch("mml_opt_step_dense");
    if (rc) return rc;
  }
  return MML_OK;
}

static int launch_catchup(CatchupLaunch& L, int64_t max_items, hipStream_t st, const char* who) {
  int64_t blocks = cdiv(max_items, 256);
  if (blocks > 256 * 8) blocks = 256 * 8;
  if (blocks < 1) blocks = 1;
  MML_LAUNCH(opt_catchup_kernel, dim3((unsigned)blocks), dim3(256), 0, st, L);
  return check_launch(who);
}

extern "C" int mml_opt_catchup_rows(float* const* tables, float* const* state1, float* const* state2,
                                    int32_t* const* last, const int64_t* rowbase, int32_t F, int32_t E,
                                    const int32_t* touched, const int32_t* touched_count, int32_t touched_cap,
                                    const mml_opt_hyper* hyper, mml_stream_t stream) {
  int rc = check_hyper(hyper, "mml_opt_catchup_rows");
  if (rc) return rc;
  MML_REQUIRE(F >= 1 && F <= MML_MAX_FIELDS && E > 0 && tables && last && rowbase && touched && touched_count &&
                  touched_cap > 0, "mml_opt_catchup_rows: bad arguments");
  if (hyper->kind == MML_OPT_SGD || hyper->kind == MML_OPT_ADAGRAD) return MML_OK;  // zero gradients change nothing
  CatchupLaunch L{};
  for (int f = 0; f < F; ++f) {
    MML_REQUIRE(tables[f] && last[f] && state1 && state1[f], "mml_opt_catchup_rows: field %d null", f);
    MML_REQUIRE(hyper->kind != MML_OPT_ADAM || (state2 && state2[f]), "mml_opt_catchup_rows: field %d needs state2", f);
    L.tab[f] = tables[f]; L.s1[f] = state1[f]; L.s2[f] = state2 ? state2[f] : nullptr; L.last[f] = last[f];
    L.rowbase[f] = rowbase[f];
  }
  L.rowbase[F] = rowbase[F];
  L.F = F; L.E = E; L.touched = touched; L.touched_count = touched_count; L.cap = touched_cap; L.h = *hyper;
  return launch_catchup(L, (int64_t)touched_cap * E, to_stream(stream), "mml_opt_catchup_rows");
}

extern "C" int mml_opt_catchup_dense(float* table, float* state1, float* state2, int32_t* last, int64_t V, int32_t E,
                                     const mml_opt_hyper* hyper, mml_stream_t stream) {
  int rc = check_hyper(hyper, "mml_opt_catchup_dense");
  if (rc) return rc;
  MML_REQUIRE(table && last && V >= 0 && E > 0, "mml_opt_catchup_dense: bad arguments");
  if (V == 0 || hyper->kind == MML_OPT_SGD || hyper->kind == MML_OPT_ADAGRAD) return MML_OK;
  MML_REQUIRE(state1 && (hyper->kind != MML_OPT_ADAM || state2), "mml_opt_catchup_dense: optimizer state missing");
  CatchupLaunch L{};
  L.tab[0] = table; L.s1[0] = state1; L.s2[0] = state2; L.last[0] = last;
  L.F = 1; L.E = E; L.dense = 1; L.V = V; L.h = *hyper;
  return launch_catchup(L, V * E, to_stream(stream), "mml_opt_catchup_dense");
}

extern "C" int mml_opt_step_rows(float* const* tables, float* const* grad_tables, float* const* state1,
                                 float* const* state2, uint32_t* const* seen, const int64_t* rowbase, int32_t F,
                                 int32_t E, const int32_t* touched, const int32_t* touched_count, int32_t touched_cap,
                                 int32_t* const* last, const mml_opt_hyper* hyper, mml_stream_t stream) {
  int rc = check_hyper(hyper, "mml_opt_step_rows");
  if (rc) return rc;
  MML_REQUIRE(F >= 1 && F <= MML_MAX_FIELDS && E > 0, "mml_opt_step_rows: bad F/E");
  MML_REQUIRE(tables && grad_tables && seen && rowbase && touched && touched_count && touched_cap > 0,
              "mml_opt_step_rows: null argument");
  RowsLaunch L{};
  for (int f = 0; f < F; ++f) {
    MML_REQUIRE(tables[f] && grad_tables[f] && seen[f], "mml_opt_step_rows: field %d null", f);
    L.tab[f] = tables[f]; L.grad[f] = grad_tables[f]; L.seen[f] = seen[f];
    L.s1[f] = state1 ? state1[f] : nullptr;
    L.s2[f] = state2 ? state2[f] : nullptr;
    MML_REQUIRE(hyper->kind == MML_OPT_SGD || L.s1[f], "mml_opt_step_rows: field %d needs state1", f);
    MML_REQUIRE(hyper->kind != MML_OPT_ADAM || L.s2[f], "mml_opt_step_rows: field %d needs state2", f);
    L.rowbase[f] = rowbase[f];
    L.last[f] = last ? last[f] : nullptr;
  }
  L.rowbase[F] = rowbase[F];
  L.F = F; L.E = E; L.touched = touched; L.touched_count = touched_count; L.cap = touched_cap; L.h = *hyper;
  // the row count lives on the device: size the grid for the capacity, surplus workgroups exit at once
  bool vec = (E % 4 == 0);
  for (int f = 0; f < F && vec; ++f)
    vec = aligned16(L.tab[f]) && aligned16(L.grad[f]) && (!L.s1[f] || aligned16(L.s1[f])) && (!L.s2[f] || aligned16(L.s2[f]));
  int64_t blocks = cdiv((int64_t)touched_cap * (vec ? E / 4 : E), 256);
  if (blocks > 256 * 8) blocks = 256 * 8;
  if (vec) MML_LAUNCH(opt_rows_kernel<4>, dim3((unsigned)blocks), dim3(256), 0, to_stream(stream), L);
  else MML_LAUNCH(opt_rows_kernel<1>, dim3((unsigned)blocks), dim3(256), 0, to_stream(stream), L);
  return check_launch("mml_opt_step_rows");
}

extern "C" int mml_counter_update(int32_t* counter, int32_t delta, int32_t reset, mml_stream_t stream) {
  MML_REQUIRE(counter, "mml_counter_update: null counter");
  MML_LAUNCH(counter_kernel, dim3(1), dim3(64), 0, to_stream(stream), counter, delta, reset);
  return check_launch("mml_counter_update");
}

extern "C" int mml_ew_mul(const float* a, const float* b, float* out, int64_t n, mml_stream_t stream) {
  MML_REQUIRE(n >= 0 && (n == 0 || (a && b && out)), "mml_ew_mul: null argument");
  if (n == 0) return MML_OK;
  MML_LAUNCH(ew_mul_kernel, dim3(ew_grid(n)), dim3(256), 0, to_stream(stream), a, b, out, n);
  return check_launch("mml_ew_mul");
}

extern "C" int mml_ew_mul_bwd_act(const float* dout, const float* a, const float* b, float* da, float* db,
                                  int32_t acc_a, int32_t acc_b, int64_t n, int32_t act_a, int32_t act_b,
                                  mml_stream_t stream) {
  MML_REQUIRE(n >= 0 && (n == 0 || dout), "mml_ew_mul_bwd: null dout");
  MML_REQUIRE(!da || b, "mml_ew_mul_bwd: da needs b");
  MML_REQUIRE(!db || a, "mml_ew_mul_bwd: db needs a");
  MML_REQUIRE(act_a >= MML_ACT_NONE && act_a <= MML_ACT_SIGMOID2 && act_b >= MML_ACT_NONE && act_b <= MML_ACT_SIGMOID2,
              "mml_ew_mul_bwd: unknown activation");
  MML_REQUIRE((!act_a || (a && da && !acc_a)) && (!act_b || (b && db && !acc_b)),
              "mml_ew_mul_bwd: a folded activation derivative needs the operand, its gradient, and no accumulation");
  if (n == 0) return MML_OK;
  MML_LAUNCH(ew_mul_bwd_kernel, dim3(ew_grid(n)), dim3(256), 0, to_stream(stream), dout, a, b, da, db, acc_a,
                     acc_b, n, act_a, act_b);
  return check_launch("mml_ew_mul_bwd");
}

extern "C" int mml_ew_mul_bwd(const float* dout, const float* a, const float* b, float* da, float* db, int32_t acc_a,
                              int32_t acc_b, int64_t n, mml_stream_t stream) {
  return mml_ew_mul_bwd_act(dout, a, b, da, db, acc_a, acc_b, n, MML_ACT_NONE, MML_ACT_NONE, stream);
}

extern "C" int mml_ew_add_n(const float* const* in, int32_t n_in, float* out, int64_t n, mml_stream_t stream) {
  MML_REQUIRE(in && n_in >= 1 && n_in <= 16 && out && n >= 0, "mml_ew_add_n: bad arguments");
  if (n == 0) return MML_OK;
  AddN A{};
  A.n_in = n_in;
  for (int i = 0; i < n_in; ++i) {
    MML_REQUIRE(in[i], "mml_ew_add_n: input %d null", i);
    A.in[i] = in[i];
  }
  MML_LAUNCH(ew_add_n_kernel, dim3(ew_grid(n)), dim3(256), 0, to_stream(stream), A, out, n);
  return check_launch("mml_ew_add_n");
}

extern "C" int mml_copy2d(const float* src, int64_t lds_, float* dst, int64_t ldd, int64_t rows, int32_t cols,
                          int32_t accumulate, mml_stream_t stream) {
  MML_REQUIRE(rows >= 0 && cols >= 0, "mml_copy2d: negative extent");
  if (rows == 0 || cols == 0) return MML_OK;
  MML_REQUIRE(src && dst && lds_ >= cols && ldd >= cols, "mml_copy2d: null pointer or leading dimension < cols");
  MML_LAUNCH(copy2d_kernel, dim3(ew_grid(rows * cols)), dim3(256), 0, to_stream(stream), src, lds_, dst, ldd,
                     rows, cols, accumulate);
  return check_launch("mml_copy2d");
}

extern "C" int mml_copy2d_batch(const mml_copy2d_desc* d, int32_t n, mml_stream_t stream) {
  MML_REQUIRE(n >= 0 && (n == 0 || d), "mml_copy2d_batch: null descriptor array");
  for (int i0 = 0; i0 < n; i0 += COPY2D_BATCH) {
    Copy2dBatch Bt{};
    int m = 0;
    int64_t most = 0;
    for (int i = i0; i < n && m < COPY2D_BATCH; ++i) {
      const mml_copy2d_desc& D = d[i];
      MML_REQUIRE(D.rows >= 0 && D.cols >= 0, "mml_copy2d_batch: item %d has a negative extent", i);
      if (D.rows == 0 || D.cols == 0) continue;
      MML_REQUIRE(D.src && D.dst && D.lds >= D.cols && D.ldd >= D.cols,
                  "mml_copy2d_batch: item %d: null pointer or leading dimension < cols", i);
      Bt.d[m++] = D;
      most = D.rows * D.cols > most ? D.rows * D.cols : most;
    }
    if (m == 0) continue;
    MML_LAUNCH(copy2d_batch_kernel, dim3(ew_grid(most), (unsigned)m), dim3(256), 0, to_stream(stream), Bt);
    int rc = check_launch("mml_copy2d_batch");
    if (rc) return rc;
  }
  return MML_OK;
}

extern "C" int mml_copy_cols(const float* const* src, const int64_t* lds_, float* const* dst, const int64_t* ldd,
                             const int32_t* width, int32_t n_seg, int64_t rows, int32_t accumulate,
                             mml_stream_t stream) {
  MML_REQUIRE(n_seg >= 0 && rows >= 0, "mml_copy_cols: negative extent");
  if (n_seg == 0 || rows == 0) return MML_OK;
  MML_REQUIRE(src && lds_ && dst && ldd && width, "mml_copy_cols: null array");
  for (int i0 = 0; i0 < n_seg; i0 += MML_MAX_FIELDS) {
    ColSegs S{};
    S.n = (n_seg - i0 < MML_MAX_FIELDS) ? (n_seg - i0) : MML_MAX_FIELDS;
    int acc = 0;
    for (int s = 0; s < S.n; ++s) {
      const int k = i0 + s;
      MML_REQUIRE(src[k] && dst[k] && width[k] > 0 && lds_[k] >= width[k] && ldd[k] >= width[k],
                  "mml_copy_cols: segment %d malformed", k);
      S.src[s] = src[k]; S.dst[s] = dst[k]; S.lds_[s] = lds_[k]; S.ldd[s] = ldd[k]; S.width[s] = width[k];
      S.start[s] = acc;
      acc += width[k];
    }
    S.start[S.n] = acc;
    bool v4 = (acc >> 2) <= COPY_COLS_MAX_W4;
    for (int s = 0; s < S.n && v4; ++s)
      v4 = S.width[s] % 4 == 0 && S.lds_[s] % 4 == 0 && S.ldd[s] % 4 == 0 && aligned16(S.src[s]) && aligned16(S.dst[s]);
    if (v4)
      MML_LAUNCH(copy_cols_vec4_kernel, dim3(ew_grid(rows * (acc >> 2))), dim3(256), 0, to_stream(stream), S, rows,
                 accumulate);
    else
      MML_LAUNCH(copy_cols_kernel, dim3(ew_grid(rows * acc)), dim3(256), 0, to_stream(stream), S, rows, accumulate);
    int rc = check_launch("mml_copy_cols");
    if (rc) return rc;
  }
  return MML_OK;
}

extern "C" int mml_act_bwd(const float* y, const float* dy, float* dst, int64_t n, int32_t act, mml_stream_t stream) {
  MML_REQUIRE(n >= 0 && (n == 0 || (y && dy && dst)), "mml_act_bwd: null argument");
  MML_REQUIRE(act >= MML_ACT_NONE && act <= MML_ACT_SIGMOID2, "mml_act_bwd: unknown activation %d", act);
  if (n == 0) return MML_OK;
  MML_LAUNCH(act_bwd_kernel, dim3(ew_grid(n)), dim3(256), 0, to_stream(stream), y, dy, dst, n, act);
  return check_launch("mml_act_bwd");
}

extern "C" int mml_sumprod_batch(const mml_sumprod_desc* d, int32_t n, mml_stream_t stream) {
  MML_REQUIRE(n >= 0 && (n == 0 || d), "mml_sumprod_batch: bad descriptor array");
  for (int i0 = 0; i0 < n; i0 += MML_SUMPROD_BATCH) {
    SumProdBatch Bt{};
    const int m = (n - i0 < MML_SUMPROD_BATCH) ? n - i0 : MML_SUMPROD_BATCH;
    int64_t nmax = 0;
    for (int i = 0; i < m; ++i) {
      const mml_sumprod_desc& D = d[i0 + i];
      MML_REQUIRE(D.out && D.n >= 0 && D.n_terms >= 0 && D.n_terms <= MML_SUMPROD_TERMS, "mml_sumprod_batch: item %d malformed", i0 + i);
      for (int k = 0; k < D.n_terms; ++k) MML_REQUIRE(D.x[k], "mml_sumprod_batch: item %d term %d is null", i0 + i, k);
      MML_REQUIRE(D.act >= MML_ACT_NONE && D.act <= MML_ACT_SIGMOID2 && (D.act == MML_ACT_NONE || (D.deriv_of && !D.accumulate)),
                  "mml_sumprod_batch: item %d: a folded activation derivative needs deriv_of and no accumulation", i0 + i);
      Bt.d[i] = D;
      nmax = D.n > nmax ? D.n : nmax;
    }
    if (nmax == 0) continue;
    // about 2048 workgroups in all (8 per CU: every wave slot) walking their item grid-stride: items that publish their
    // magnitude end with one atomic per workgroup, and 30 000 of those on eight words cost more than the pass itself
    unsigned gx = ew_grid(nmax);
    const unsigned cap = (unsigned)(2048 / m > 64 ? 2048 / m : 64);
    if (gx > cap) gx = cap;
    MML_LAUNCH(sumprod_batch_kernel, dim3(gx, (unsigned)m), dim3(256), 0, to_stream(stream), Bt);
    int rc = check_launch("mml_sumprod_batch");
    if (rc) return rc;
  }
  return MML_OK;
}
