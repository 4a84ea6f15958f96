// K3 (round 4): activation-stationary forward GEMM for the layers that SHARE their input.
//
// The first DNN layer of every expert and gate tower reads the same dnn_input (reference model/mmoe.py:69-79: each
// expert_dnn / gate_dnn is called on the one combined input; model/utils.py:146-161 is the Linear -> ReLU layer itself).
// gemm_pipe_kernel (gemm.hip) treats the N-tiles of such a launch as unrelated tiles: the 128 x K activation panel is
// fetched, staged and cut into its two fp16 planes once per N-TILE -- nine times for AE-30's 4 x 256 + 2 x 64 output
// columns -- and every tile pays its own prologue.  Here a persistent workgroup owns a 128-row panel:
//   * the panel (K <= 240: <= 120 KB) is moved into LDS ONCE, cut in place into the two fp16 planes of the scaled values
//     (same bits as the in-register cut of gemm.hip: h = rne16(x s), l = rne16(x s - h)), and stays there while the
//     workgroup sweeps every N-tile of the row block;
//   * the weights arrive pre-cut (mml_gemm_planes_cut, MML_PLANES_ROWS) through a three-stage LDS-DMA ring, so the k-loop
//     holds no VALU work at all: LDS-DMA issue, fragment reads one step ahead, twelve v_mfma_f32_32x32x16_f16 per wave;
//   * one wave per SIMD (256 threads, up to 512 VGPRs): the finished tile leaves the k-loop in a second register set
//     (the last MFMA of every product block writes there) and its epilogue -- unscale, bias, ReLU, sign mask, row-major
//     turn through LDS, whole-line stores -- is dealt in eight pieces into the k-steps of the NEXT tile, so stores
//     trickle out beside the MFMAs instead of arriving as one burst per tile;
//   * a tile is a PAIR of 64-column half tiles, each with its own problem: two 64-wide gate layers share one tile.
// Results are bitwise those of gemm_pipe_kernel<.., EMU = 2, BPL> on the same operands (same planes, same product
// order hl, lh, hh per 16-k block, same k order): tests/test_gemm_panel_gpu.py.
#include "common.hpp"
#include "lds_async.hpp"

#include <stdlib.h>

#include <type_traits>

namespace mml {

using pf32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int PN_BM = 128;               // panel rows
constexpr int PN_KB_MAX = 15;            // 16-k blocks of a panel (K <= 240)
constexpr int PN_KB_MIN = 10;            // (eight epilogue pieces need eight middle k-steps)
constexpr int PN_STAGE = 8192;           // bytes of one 128-row x 16-k image
constexpr int PN_NST = 3;                // weight ring stages
constexpr int PN_SCR = 2048;             // bytes of a wave's row-major turn area (16 rows x 128 B)
constexpr int PN_MAXH = 64;              // half tiles per launch
constexpr int PN_PANEL_OFF = 0;
constexpr int PN_RING_OFF = PN_KB_MAX * PN_STAGE;
constexpr int PN_SCR_OFF = PN_RING_OFF + PN_NST * PN_STAGE;
constexpr int PN_BIAS_OFF = PN_SCR_OFF + 4 * PN_SCR;            // two buffers x four waves x 64 floats
constexpr int PN_AMAX_OFF = PN_BIAS_OFF + 2 * 4 * 256;          // one word per problem
constexpr int PN_INVB_OFF = PN_AMAX_OFF + MML_MAX_GROUP * 4;    // 2^-kB per problem (read once from the exponent words)
constexpr int PN_LDS_BYTES = PN_INVB_OFF + MML_MAX_GROUP * 4;
static_assert(PN_LDS_BYTES <= 160 * 1024, "panel kernel LDS budget");

struct PanelProblem {
  float* C;
  const float* bias;
  uint32_t* mask;
  const uint32_t* planes;  // MML_PLANES_ROWS image of W [N, K], pitch ldp words
  const int32_t* kexp;     // exponent the planes were cut with
  uint32_t* amax_out;
  int64_t ldc, ldmask, ldp;
  int32_t N, relu;
};

struct PanelLaunch {
  const float* A;
  const uint32_t* amaxA;
  int64_t lda;
  int32_t M, K;
  int32_t n_prob, n_half;
  int32_t store_masks;  // every problem writes its relu sign mask (3 stores per epilogue piece instead of 2)
  int32_t pad_;
  PanelProblem p[MML_MAX_GROUP];
  uint16_t half_prob[PN_MAXH];
  uint16_t half_col0[PN_MAXH];
};
static_assert(sizeof(PanelLaunch) <= 4096, "PanelLaunch must fit the kernel-argument block");

__device__ __forceinline__ uint32_t pn_amax_load(const uint32_t* p) {
  uint32_t m = 0;
#pragma unroll
  for (int i = 0; i < MML_AMAX_WORDS; ++i) m = p[i] > m ? p[i] : m;
  return m;
}
// (the rule of gemm.hip: |x| 2^k < 2^15 for every |x| <= the slot's value; Inf / NaN: scale 1)
__device__ __forceinline__ int pn_scale_exp(uint32_t bits) {
  const int e = (int)((bits >> 23) & 0xffu);
  if (e == 255) return 0;
  const int k = 141 - e;
  return k > 110 ? 110 : (k < -110 ? -110 : k);
}
__device__ __forceinline__ float pn_pow2(int k) { return __uint_as_float((uint32_t)(127 + k) << 23); }

__device__ __forceinline__ void pn_wait_vm(const int n) {
  switch (n) {
    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
    case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
    case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
    case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
    case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
  }
}

enum { PN_FIRST = 0, PN_MID = 1, PN_LAST = 2 };

// the raw LDS fragments of one k-step: a lane's 8 halves of the h and the l plane for two 32-row / 32-column sub-tiles
struct PnFrag {
  f32x4_t ah[2], al[2], bh[2], bl[2];
  __device__ __forceinline__ void landed() {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      lds_landed(ah[i]);
      lds_landed(al[i]);
      lds_landed(bh[i]);
      lds_landed(bl[i]);
    }
  }
};

template <bool MASKS>
__global__ __launch_bounds__(256, 1) void gemm_panel_kernel(const PanelLaunch Larg) {
  typedef const __attribute__((address_space(4))) PanelLaunch KLaunch;
  KLaunch& L = *(KLaunch*)__builtin_amdgcn_kernarg_segment_ptr();
  __shared__ __attribute__((aligned(16))) float lds[PN_LDS_BYTES / 4];
  constexpr int NSTORE = MASKS ? 3 : 2;  // VMEM operations of one epilogue piece (two 16-byte row stores + mask words)

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane_ = lane;
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, h = lane >> 5;
  const uint32_t lds0 = lds_byte_addr(lds);
  const int KB = L.K >> 4;
  const int M = L.M;
  const int npairs = (L.n_half + 1) >> 1;
  const int npanels = (M + PN_BM - 1) / PN_BM;

  if (tid < MML_MAX_GROUP) {
    const uint32_t a = lds0 + PN_AMAX_OFF + tid * 4, z = 0u;
    asm volatile("ds_write_b32 %0, %1" ::"v"(a), "v"(z) : "memory");
    // the weights' exponents, once: a global load inside the tile loop would make hipcc drain the LDS-DMA ring there
    const float ib = tid < L.n_prob ? pn_pow2(-*L.p[tid].kexp) : 1.f;
    const uint32_t b = lds0 + PN_INVB_OFF + tid * 4;
    asm volatile("ds_write_b32 %0, %1" ::"v"(b), "v"(ib) : "memory");
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  // power-of-two scale of the activations (one per launch: every problem reads the same panel)
  const int kA = __builtin_amdgcn_readfirstlane(pn_scale_exp(pn_amax_load(L.amaxA)));
  const float sA = pn_pow2(kA), invA = pn_pow2(-kA);

  // ---- fragment read addresses (the image of gemm.hip: 64-byte rows, 16-byte chunk c of row r at c ^ ((r >> 2) & 3)) ----
  const int swz = (l31 >> 2) & 3;
  const uint32_t aAh = lds0 + PN_PANEL_OFF + (wm * 64 + l31) * 64 + ((h ^ swz) * 16);
  const uint32_t aAl = lds0 + PN_PANEL_OFF + (wm * 64 + l31) * 64 + (((2 + h) ^ swz) * 16);
  const uint32_t aBh = lds0 + PN_RING_OFF + (wn * 64 + l31) * 64 + ((h ^ swz) * 16);
  const uint32_t aBl = lds0 + PN_RING_OFF + (wn * 64 + l31) * 64 + (((2 + h) ^ swz) * 16);
  auto read_frags = [&](PnFrag& f, const int kb, const int stage) __attribute__((always_inline)) {
    const uint32_t ka = (uint32_t)kb * PN_STAGE, sb = (uint32_t)stage * PN_STAGE;
    f.bh[0] = ds_read128<0>(aBh + sb);
    f.bl[0] = ds_read128<0>(aBl + sb);
    f.ah[0] = ds_read128<0>(aAh + ka);
    f.al[0] = ds_read128<0>(aAl + ka);
    f.bh[1] = ds_read128<2048>(aBh + sb);
    f.bl[1] = ds_read128<2048>(aBl + sb);
    f.ah[1] = ds_read128<2048>(aAh + ka);
    f.al[1] = ds_read128<2048>(aAl + ka);
  };

  // ---- DMA lane geometry: wave-instruction t of a 128-row image covers rows 16 t .. 16 t + 15, lane -> (row, phys chunk) ----
  const int drow = lane >> 2;                          // row inside the 16-row group
  const int dchunk = (lane & 3) ^ ((drow >> 2) & 3);   // logical chunk that lands at physical chunk lane & 3

  // ---- per-tile context of this wave's half tile ----
  struct Ctx {
    float* C;           // &C[0][col0]
    uint32_t* mask;     // &mask[0][col0 / 32] or null
    int64_t ldc, ldmask;
    float inv;          // 2^-(kA + kB)
    int pi;
    bool valid, amax;
  };
  auto load_ctx = [&](Ctx& c, const int pair) __attribute__((always_inline)) {
    int hi = 2 * pair + wn;
    c.valid = hi < L.n_half;
    hi = c.valid ? hi : 2 * pair;
    const int pi = L.half_prob[hi], col0 = L.half_col0[hi];
    c.pi = pi;
    c.ldc = L.p[pi].ldc;
    c.ldmask = L.p[pi].ldmask;
    c.C = L.p[pi].C + col0;
    uint32_t* const mk = L.p[pi].mask;
    c.mask = mk ? mk + (col0 >> 5) : nullptr;
    float ib;
    {
      const uint32_t b = lds0 + PN_INVB_OFF + (uint32_t)pi * 4u;
      asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(ib) : "v"(b) : "memory");
    }
    c.inv = invA * __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, ib)));
    c.amax = L.p[pi].amax_out != nullptr;
  };

  // weight ring: source pointers of this wave's two DMA instructions per stage (j = half tile j of the pair)
  const float* pb[2];
  auto setup_b = [&](const int pair) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      int hi = 2 * pair + j;
      hi = hi < L.n_half ? hi : 2 * pair;
      const int pi = L.half_prob[hi], col0 = L.half_col0[hi];
      const float* base = reinterpret_cast<const float*>(L.p[pi].planes);
      pb[j] = base + (int64_t)(col0 + 16 * wave + drow) * L.p[pi].ldp + 4 * dchunk;
    }
  };
  auto issue_b = [&](const int stage) __attribute__((always_inline)) {
    float* sb = lds + (PN_RING_OFF + stage * PN_STAGE) / 4;
#pragma unroll
    for (int j = 0; j < 2; ++j) dma16(pb[j], sb + (wave + 4 * j) * 256);
#pragma unroll
    for (int j = 0; j < 2; ++j) pb[j] += 16;
  };
  auto bias_dma = [&](const int pair) __attribute__((always_inline)) {
    int hi = 2 * pair + wn;
    hi = hi < L.n_half ? hi : 2 * pair;
    const int pi = L.half_prob[hi], col0 = L.half_col0[hi];
    const float* bias = L.p[pi].bias;
    const float* src = bias ? bias + col0 + lane : L.A;  // (always exactly one VMEM operation: the waits count it)
    __builtin_amdgcn_sched_barrier(0);
    dma4(src, lds + (PN_BIAS_OFF + (pair & 1) * 1024 + wave * 256) / 4);
    __builtin_amdgcn_sched_barrier(0);
  };

  pf32x16 acc[2][2], epi[2][2];
  PnFrag F[2];
  Ctx cctx, ectx;
  float am_f = 0.f;
  int row0 = 0;
  bool full = true;
  bool ectx_bias = false;
  int epair = 0;

  // ---- one eighth of a tile's epilogue: sub-tile (mi, ni) = U >> 1, rows 16 (U & 1) .. + 15 of it ----
  auto unit = [&](auto uc) __attribute__((always_inline)) {
    constexpr int U = decltype(uc)::value;
    constexpr int MI = U >> 2, NI = (U >> 1) & 1, RH = U & 1;
    const pf32x16& a = epi[MI][NI];
    // every address of a piece is derived from an OPAQUE copy of the lane number: hipcc otherwise hoists the address
    // arithmetic of all 8 x 2 x 2 instantiations out of the panel loop and spills (256 VGPRs + scratch)
    int lane = lane_;
    asm volatile("" : "+v"(lane));
    const int l31 = lane & 31, h = lane >> 5;
    const uint32_t scr = lds0 + PN_SCR_OFF + wave * PN_SCR;
    const int rr = l31 & 15;
    if ((l31 >> 4) == RH) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4_t v = {a[4 * g], a[4 * g + 1], a[4 * g + 2], a[4 * g + 3]};
        ds_write128(scr + rr * 128 + (((2 * g + h) ^ (rr & 7)) * 16), v);
      }
    }
    const int R = lane >> 3, cc = lane & 7;
    f32x4_t b4 = {0.f, 0.f, 0.f, 0.f};
    if (ectx_bias) b4 = ds_read128<0>(lds0 + PN_BIAS_OFF + (epair & 1) * 1024 + wave * 256 + (NI * 32 + 4 * cc) * 4);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (ectx_bias) lds_landed(b4);
    __builtin_amdgcn_sched_barrier(0);
    f32x4_t v[2];
#pragma unroll
    for (int p = 0; p < 2; ++p) v[p] = ds_read128<0>(scr + (R + 8 * p) * 128 + ((cc ^ ((R + 8 * p) & 7)) * 16));
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int p = 0; p < 2; ++p) lds_landed(v[p]);
    __builtin_amdgcn_sched_barrier(0);
    const float inv = ectx.inv;
    const bool relu = L.p[ectx.pi].relu != 0;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      f32x4_t& x = v[p];
      x.x = x.x * inv + b4.x;
      x.y = x.y * inv + b4.y;
      x.z = x.z * inv + b4.z;
      x.w = x.w * inv + b4.w;
      if (relu) {
        x.x = x.x > 0.f ? x.x : 0.f;
        x.y = x.y > 0.f ? x.y : 0.f;
        x.z = x.z > 0.f ? x.z : 0.f;
        x.w = x.w > 0.f ? x.w : 0.f;
      }
    }
    const int rowb = row0 + wm * 64 + MI * 32 + RH * 16 + R;  // rows rowb, rowb + 8
    const int colq = NI * 32 + 4 * cc;
    if (MASKS) {  // the eight lanes cc = 0..7 of a row hold the eight nibbles of its 32-column word
      uint32_t wsel = 0u;
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        const f32x4_t& x = v[p];
        uint32_t w = ((x.x > 0.f ? 1u : 0u) | (x.y > 0.f ? 2u : 0u) | (x.z > 0.f ? 4u : 0u) | (x.w > 0.f ? 8u : 0u))
                     << (4 * cc);
        w |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)w, 0x104, 0xf, 0xf, true);
        w |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)w, 0x102, 0xf, 0xf, true);
        w |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)w, 0x101, 0xf, 0xf, true);
        uint32_t moved = w;
        if (p == 1) moved = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)w, 0x111, 0xf, 0xf, true);
        if (cc == p) wsel = moved;
      }
      const int mrow = rowb + 8 * cc;  // lane cc < 2 owns row R + 8 cc
      if (cc < 2 && mrow < M) ectx.mask[(int64_t)mrow * ectx.ldmask + NI] = wsel;
    }
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const int row = rowb + 8 * p;
      if (row < M) {
        const float4 x = make_float4(v[p].x, v[p].y, v[p].z, v[p].w);
        *reinterpret_cast<float4*>(ectx.C + (int64_t)row * ectx.ldc + colq) = x;
        am_f = fmaxf(fmaxf(am_f, fabsf(x.x)), fmaxf(fabsf(x.y), fmaxf(fabsf(x.z), fabsf(x.w))));
      }
    }
    if (U == 7 && ectx.amax) {  // the tile's largest |output| -> the workgroup's word of its problem
      const uint32_t aw = lds0 + PN_AMAX_OFF + (uint32_t)ectx.pi * 4u;
      const uint32_t am_bits = __float_as_uint(am_f);
      asm volatile("ds_max_u32 %0, %1" ::"v"(aw), "v"(am_bits) : "memory");
      am_f = 0.f;
    }
  };

  int e2 = 0, e1 = 0;   // VMEM operations issued after the weight DMA of the step before last / of the last step
  bool have_epi = false;

  // one k-step.  Q: fragment register set of this step; KIND: first / middle / last step of the tile; U: epilogue piece
  // of the PREVIOUS tile dealt into this step (8 = none)
  auto step = [&](auto qc, auto kindc, auto uc, const int kb, const int gs, const bool more_dma) __attribute__((always_inline)) {
    constexpr int Q = decltype(qc)::value, KIND = decltype(kindc)::value, U = decltype(uc)::value;
    // the weights of step gs + 1 were issued two steps ago; younger: what followed them in that step, the DMA of the
    // step before this one and what followed it
    pn_wait_vm(more_dma ? 2 + e2 + e1 : 0);
    __builtin_amdgcn_s_barrier();
    int ecur = 0;
    if (more_dma) issue_b((gs + 3) % PN_NST);
    // next step's fragments -> the other register set
    const int kbn = (kb + 1 == KB) ? 0 : kb + 1;
    read_frags(F[Q ^ 1], kbn, (gs + 1) % PN_NST);
    __builtin_amdgcn_sched_barrier(0);
    const PnFrag& f = F[Q];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) {
        const f16x8 ah = __builtin_bit_cast(f16x8, f.ah[mi]), al = __builtin_bit_cast(f16x8, f.al[mi]);
        const f16x8 bh = __builtin_bit_cast(f16x8, f.bh[ni]), bl = __builtin_bit_cast(f16x8, f.bl[ni]);
        pf32x16 c0;
        if (KIND == PN_FIRST) {
#pragma unroll
          for (int r = 0; r < 16; ++r) c0[r] = 0.f;
        } else {
          c0 = acc[mi][ni];
        }
        pf32x16 t = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl, ah, c0, 0, 0, 0);
        t = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh, al, t, 0, 0, 0);
        t = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh, ah, t, 0, 0, 0);
        if (KIND == PN_LAST) epi[mi][ni] = t;
        else acc[mi][ni] = t;
      }
    if constexpr (U < 8) {
      if (have_epi && ectx.valid) {
        unit(uc);
        ecur = NSTORE;
        if (!full) {  // an edge panel's guarded stores may issue fewer operations than counted: drain
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    F[Q ^ 1].landed();
    __builtin_amdgcn_sched_barrier(0);
    e2 = e1;
    e1 = ecur;
  };

  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using KF = std::integral_constant<int, PN_FIRST>;
  using KM = std::integral_constant<int, PN_MID>;
  using KL = std::integral_constant<int, PN_LAST>;
  using UN = std::integral_constant<int, 8>;

  for (int panel = blockIdx.x; panel < npanels; panel += gridDim.x) {
    row0 = panel * PN_BM;
    full = row0 + PN_BM <= M;
    // ---- the activation panel: fp32 image by LDS-DMA, then cut in place into the planes of the scaled values ----
    {
      const float* pa[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        int row = row0 + 16 * (wave + 4 * j) + drow;
        row = row < M ? row : M - 1;
        pa[j] = L.A + (int64_t)row * L.lda + 4 * dchunk;
      }
      for (int kb = 0; kb < KB; ++kb) {
        float* sa = lds + (PN_PANEL_OFF + kb * PN_STAGE) / 4;
#pragma unroll
        for (int j = 0; j < 2; ++j) dma16(pa[j] + 16 * kb, sa + (wave + 4 * j) * 256);
      }
    }
    // the weight ring of the first tile starts while the panel lands
    setup_b(0);
    int dpair = 0, dkb = 0;  // cursor of the weight DMA
    const int total = npairs * KB;
    int issued = 0;
#pragma unroll
    for (int st = 0; st < PN_NST; ++st)
      if (issued < total) {
        issue_b(st);
        ++issued;
        if (++dkb == KB) {
          dkb = 0;
          ++dpair;
          if (dpair < npairs) setup_b(dpair);
        }
      }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    for (int kb = 0; kb < KB; ++kb) {
      // thread = (row 32 wave' + l31, half h) for wave' = wave: rows 32 wave .. 32 wave + 31, all k-blocks
      const int r = 32 * wave + l31;
      const uint32_t a0 = lds0 + PN_PANEL_OFF + kb * PN_STAGE + r * 64 + ((h ^ swz) * 16);
      const uint32_t a1 = lds0 + PN_PANEL_OFF + kb * PN_STAGE + r * 64 + (((2 + h) ^ swz) * 16);
      f32x4_t q0 = ds_read128<0>(a0), q1 = ds_read128<0>(a1);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      lds_landed(q0);
      lds_landed(q1);
      __builtin_amdgcn_sched_barrier(0);
      const float x[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
      uint32_t hw[4], lw[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float y0 = x[2 * i] * sA, y1 = x[2 * i + 1] * sA;  // exact: sA is a power of two
        const _Float16 h0 = (_Float16)y0, h1 = (_Float16)y1;     // round to nearest even
        const float r0 = y0 - (float)h0, r1 = y1 - (float)h1;    // exact
        const _Float16 l0 = (_Float16)r0, l1 = (_Float16)r1;
        hw[i] = (uint32_t)__builtin_bit_cast(uint16_t, h0) | ((uint32_t)__builtin_bit_cast(uint16_t, h1) << 16);
        lw[i] = (uint32_t)__builtin_bit_cast(uint16_t, l0) | ((uint32_t)__builtin_bit_cast(uint16_t, l1) << 16);
      }
      const f32x4_t ph = {__uint_as_float(hw[0]), __uint_as_float(hw[1]), __uint_as_float(hw[2]), __uint_as_float(hw[3])};
      const f32x4_t pl = {__uint_as_float(lw[0]), __uint_as_float(lw[1]), __uint_as_float(lw[2]), __uint_as_float(lw[3])};
      ds_write128(a0, ph);
      ds_write128(a1, pl);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    // stage 0 -> registers
    read_frags(F[0], 0, 0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    F[0].landed();
    __builtin_amdgcn_sched_barrier(0);
    e2 = e1 = 0;
    have_epi = false;
    int gs = 0;  // step counter of this panel (its parity picks the fragment register set)
    auto dma_advance = [&]() __attribute__((always_inline)) {
      ++issued;
      if (++dkb == KB) {
        dkb = 0;
        ++dpair;
        if (dpair < npairs) setup_b(dpair);
      }
    };
    for (int pair = 0; pair < npairs; ++pair) {
      load_ctx(cctx, pair);
      for (int kb = 0; kb < KB; ++kb) {
        const bool more = issued < total;
        const int u = (kb >= 1 && kb <= 8) ? kb - 1 : 8;
        const int q = gs & 1;
        if (kb == 0) {
          if (q == 0) step(I0{}, KF{}, UN{}, kb, gs, more);
          else step(I1{}, KF{}, UN{}, kb, gs, more);
          bias_dma(pair);
          e1 += 1;
        } else if (kb == KB - 1) {
          if (q == 0) step(I0{}, KL{}, UN{}, kb, gs, more);
          else step(I1{}, KL{}, UN{}, kb, gs, more);
        } else {
#define PN_CASE(U_)                                                                  \
  case U_:                                                                           \
    if (q == 0) step(I0{}, KM{}, std::integral_constant<int, U_>{}, kb, gs, more);   \
    else step(I1{}, KM{}, std::integral_constant<int, U_>{}, kb, gs, more);          \
    break;
          switch (u) {
            PN_CASE(0)
            PN_CASE(1)
            PN_CASE(2)
            PN_CASE(3)
            PN_CASE(4)
            PN_CASE(5)
            PN_CASE(6)
            PN_CASE(7)
            default:
              if (q == 0) step(I0{}, KM{}, UN{}, kb, gs, more);
              else step(I1{}, KM{}, UN{}, kb, gs, more);
              break;
          }
#undef PN_CASE
        }
        if (more) dma_advance();
        ++gs;
      }
      // the finished tile sits in `epi`: its pieces go into the next tile's steps (or the flush below)
      ectx = cctx;
      ectx_bias = L.p[cctx.pi].bias != nullptr;
      epair = pair;
      have_epi = true;
    }
    // ---- the last tile of the panel: its epilogue, un-overlapped ----
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (its bias DMA included)
    if (ectx.valid) {
      unit(std::integral_constant<int, 0>{});
      unit(std::integral_constant<int, 1>{});
      unit(std::integral_constant<int, 2>{});
      unit(std::integral_constant<int, 3>{});
      unit(std::integral_constant<int, 4>{});
      unit(std::integral_constant<int, 5>{});
      unit(std::integral_constant<int, 6>{});
      unit(std::integral_constant<int, 7>{});
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // every wave has read its last fragments: the panel and the ring may be overwritten
  }

  // the workgroup's magnitudes -> the slots
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (tid < L.n_prob) {
    uint32_t* const amo = L.p[tid].amax_out;
    if (amo) {
      const uint32_t a = lds0 + PN_AMAX_OFF + tid * 4;
      uint32_t v;
      asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
      if (v) atomicMax(amo + (blockIdx.x & (MML_AMAX_WORDS - 1)), v);
    }
  }
}

static int g_panel_on = -1;
static int pn_enabled() {
  if (g_panel_on < 0) {
    const char* e = getenv("MMLREC_GEMM_PANEL");
    g_panel_on = (e && e[0] == '0') ? 0 : 1;
  }
  return g_panel_on;
}

}  // namespace mml

using namespace mml;

extern "C" int mml_gemm_set_panel(int32_t on) {
  g_panel_on = on ? 1 : 0;
  return MML_OK;
}

// Returns MML_OK when the panel kernel took the launch, MML_ERR_UNSUPPORTED (without setting an error) when the launch
// is not one it serves -- the caller (mml_gemm_grouped_fwd) then runs the tile kernel.
int mml_gemm_panel_try_fwd(const mml_gemm_fwd_desc* d, int32_t n, hipStream_t st) {
  if (!pn_enabled() || n < 1 || n > MML_MAX_GROUP) return MML_ERR_UNSUPPORTED;
  const mml_gemm_fwd_desc& d0 = d[0];
  if (d0.K % 16 != 0 || d0.K < 16 * PN_KB_MIN || d0.K > 16 * PN_KB_MAX) return MML_ERR_UNSUPPORTED;
  if (d0.M < PN_BM * 64) return MML_ERR_UNSUPPORTED;  // (small batches: the tile kernel fills the chip better)
  if (!d0.amax_a || !aligned16(d0.A) || d0.lda % 4 != 0) return MML_ERR_UNSUPPORTED;
  int halves = 0, masks = 0, relus = 0;
  for (int i = 0; i < n; ++i) {
    const mml_gemm_fwd_desc& q = d[i];
    if (q.A != d0.A || q.lda != d0.lda || q.M != d0.M || q.K != d0.K || q.amax_a != d0.amax_a) return MML_ERR_UNSUPPORTED;
    if (q.w_kn != 0 || !q.w_planes || !q.w_kexp || !aligned16(q.w_planes) || q.ldw % 4 != 0) return MML_ERR_UNSUPPORTED;
    if (q.N % 64 != 0 || q.N < 64) return MML_ERR_UNSUPPORTED;
    if (q.act != MML_ACT_RELU && q.act != MML_ACT_NONE) return MML_ERR_UNSUPPORTED;
    if (!aligned16(q.C) || q.ldc % 4 != 0 || (q.bias && !aligned16(q.bias))) return MML_ERR_UNSUPPORTED;
    const bool m = q.act == MML_ACT_RELU && q.relu_mask != nullptr;
    if (m && q.ldmask * 32 < q.N) return MML_ERR_UNSUPPORTED;
    masks += m ? 1 : 0;
    relus += 1;
    halves += q.N / 64;
  }
  if (halves > PN_MAXH) return MML_ERR_UNSUPPORTED;
  if (masks != 0 && masks != relus) return MML_ERR_UNSUPPORTED;  // (the waits count the stores of a piece: all or none)
  PanelLaunch L{};
  L.A = d0.A;
  L.amaxA = d0.amax_a;
  L.lda = d0.lda;
  L.M = d0.M;
  L.K = d0.K;
  L.n_prob = n;
  L.store_masks = masks != 0;
  int hidx = 0;
  for (int i = 0; i < n; ++i) {
    const mml_gemm_fwd_desc& q = d[i];
    PanelProblem& P = L.p[i];
    P.C = q.C;
    P.bias = q.bias;
    P.mask = (q.act == MML_ACT_RELU) ? q.relu_mask : nullptr;
    P.planes = q.w_planes;
    P.kexp = q.w_kexp;
    P.amax_out = q.amax_out;
    P.ldc = q.ldc;
    P.ldmask = q.ldmask;
    P.ldp = q.ldw;
    P.N = q.N;
    P.relu = q.act == MML_ACT_RELU;
    for (int c = 0; c < q.N; c += 64) {
      L.half_prob[hidx] = (uint16_t)i;
      L.half_col0[hidx] = (uint16_t)c;
      ++hidx;
    }
  }
  L.n_half = hidx;
  static int cus = 0;
  if (cus == 0) {
    int dev = 0, nn = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&nn, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || nn <= 0)
      nn = 256;
    cus = nn;
  }
  const int npanels = (int)cdiv(L.M, PN_BM);
  const int grid = npanels < cus ? npanels : cus;
  if (L.store_masks) MML_LAUNCH((gemm_panel_kernel<true>), dim3((unsigned)grid), dim3(256), 0, st, L);
  else MML_LAUNCH((gemm_panel_kernel<false>), dim3((unsigned)grid), dim3(256), 0, st, L);
  return check_launch("mml_gemm_grouped_fwd(panel)");
}
