// K3 (round 4): activation-stationary forward GEMM for the layers that SHARE their input.
//
// The first DNN layer of every expert and gate tower reads the same dnn_input (reference model/mmoe.py:69-79: each
// expert_dnn / gate_dnn is called on the one combined input; model/utils.py:146-161 is the Linear -> ReLU layer itself).
// gemm_pipe_kernel (gemm.hip) treats the N-tiles of such a launch as unrelated tiles: the 128 x K activation panel is
// fetched, staged and cut into its two fp16 planes once per N-TILE -- nine times for AE-30's 4 x 256 + 2 x 64 output
// columns -- and every tile pays its own prologue.  Here a persistent workgroup owns a 128-row panel:
//   * one wave per SIMD (256 threads, 512 VGPRs each): a wave keeps the MFMA fragments of ITS 32 rows of the panel -- all
//     K <= 320 of them, already cut into the two fp16 planes of the scaled values (same bits as the in-register cut of
//     gemm.hip: h = rne16(x s), l = rne16(x s - h)) -- in 8 K / 16 registers (160 of the 512 at K = 320) while the workgroup sweeps every N-tile of
//     the row block: the activations are read and cut ONCE per panel and never touch LDS;
//   * the whole LDS is a twelve-stage ring for the pre-cut weights (mml_gemm_planes_cut, MML_PLANES_ROWS): ten k-steps
//     of LDS-DMA in flight.  The depth is not a luxury: vector-memory operations of a wave retire IN ORDER, so a weight
//     tile issued behind an epilogue store cannot count as landed before that store has been acknowledged (~1.5 us under
//     load); the first form of this kernel (panel in LDS, three stages) ran 1.7x SLOWER than the tile kernel for it;
//   * the k-loop holds no VALU work: LDS-DMA issue, weight fragment reads one step ahead, twelve
//     v_mfma_f32_32x32x16_f16 per wave and step; the finished tile leaves the loop in a second register set (the last
//     MFMA of every product block writes there) and its epilogue -- unscale, bias, ReLU, sign mask, row-major turn
//     through LDS, whole-line stores -- is dealt in eight pieces into k-steps 1..8 of the NEXT tile, so the stores
//     trickle out beside the MFMAs instead of arriving as one burst per tile;
//   * a tile is a PAIR of 64-column half tiles, each with its own problem: two 64-wide gate layers share one tile.
// Results are bitwise those of gemm_pipe_kernel<.., EMU = 2, BPL> on the same operands (same planes, same product
// order hl, lh, hh per 16-k block, same k order): tests/test_gemm_panel_gpu.py.
#include "common.hpp"
#include "lds_async.hpp"

#include <stdlib.h>

#include <type_traits>
#include <utility>

namespace mml {

using pf32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int PN_BM = 128;               // panel rows (four waves x 32)
constexpr int PN_STAGE = 8192;           // bytes of one 128-column x 16-k weight image
#ifndef PN_D_LAB
#define PN_D_LAB 12
#endif
constexpr int PN_D = PN_D_LAB;           // weight ring stages
constexpr int PN_W = PN_D - 2;           // k-steps between the issue of a stage and the wait for it
constexpr int PN_SCR = 4096;             // bytes of a wave's row-major turn area (32 rows x 128 B)
constexpr int PN_MAXH = 64;              // half tiles per launch
constexpr int PN_RING_OFF = 0;
constexpr int PN_SCR_OFF = PN_RING_OFF + PN_D * PN_STAGE;
constexpr int PN_BIAS_OFF = PN_SCR_OFF + 4 * PN_SCR;            // two buffers x 128 floats
constexpr int PN_AMAX_OFF = PN_BIAS_OFF + 2 * 512;              // one word per problem, then dummy words
constexpr int PN_INVB_OFF = PN_AMAX_OFF + 2 * MML_MAX_GROUP * 4;  // 2^-kB per problem (read once from the exponent words)
constexpr int PN_ZERO_OFF = PN_INVB_OFF + MML_MAX_GROUP * 4;    // 128 zeros: the "bias" of a problem without one
constexpr int PN_LDS_BYTES = PN_ZERO_OFF + 512;
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
  int32_t half[PN_MAXH];  // problem << 16 | first column of the half tile (32-bit words: scalar loads, not VMEM)
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

// s_waitcnt vmcnt(n) for a run-time n (the immediate is six bits wide)
__device__ __forceinline__ void pn_wait_vm(const int n) {
#define PN_W1(N_) case N_: asm volatile("s_waitcnt vmcnt(" #N_ ")" ::: "memory"); break;
  switch (n) {
    PN_W1(0) PN_W1(1) PN_W1(2) PN_W1(3) PN_W1(4) PN_W1(5) PN_W1(6) PN_W1(7) PN_W1(8) PN_W1(9)
    PN_W1(10) PN_W1(11) PN_W1(12) PN_W1(13) PN_W1(14) PN_W1(15) PN_W1(16) PN_W1(17) PN_W1(18) PN_W1(19)
    PN_W1(20) PN_W1(21) PN_W1(22) PN_W1(23) PN_W1(24) PN_W1(25) PN_W1(26) PN_W1(27) PN_W1(28) PN_W1(29)
    PN_W1(30) PN_W1(31) PN_W1(32) PN_W1(33) PN_W1(34) PN_W1(35) PN_W1(36) PN_W1(37) PN_W1(38) PN_W1(39)
    PN_W1(40) PN_W1(41) PN_W1(42) PN_W1(43) PN_W1(44) PN_W1(45) PN_W1(46) PN_W1(47) PN_W1(48) PN_W1(49)
    PN_W1(50) PN_W1(51) PN_W1(52) PN_W1(53) PN_W1(54) PN_W1(55) PN_W1(56) PN_W1(57) PN_W1(58) PN_W1(59)
    PN_W1(60) PN_W1(61) PN_W1(62)
    default: asm volatile("s_waitcnt vmcnt(63)" ::: "memory"); break;
  }
#undef PN_W1
}

template <int I, int N, typename F>
__device__ __forceinline__ void pn_static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    pn_static_for<I + 1, N>(f);
  }
}

// ---- the hand-placed schedule of a tile ----
// One wave per SIMD means nothing overlaps by itself: a wave issues in program order, so whatever is to run beside the
// MFMAs has to stand BETWEEN them in the instruction stream (~5 single-issue instructions hide in the 32-cycle shadow
// of a v_mfma_f32_32x32x16), and every scalar branch is a bubble nothing covers (the first hand-placed form tested its
// run-time switches per gap: 124 us for the empty loop skeleton alone).  Hence: no branch inside a k-step -- what varies
// (epilogue running or not, the tile before stored or not) is a template argument of the tile body, what is optional
// per problem (bias, magnitude slot) is an address that points at a zero / dummy area -- and launches with anything
// irregular (ragged last panel, odd number of half tiles, an activation other than ReLU) go to the tile kernel.
// A k-step is twelve MFMAs = twelve gaps:
//   gap 0, 1   the eight fragment reads of the next step
//   gap 2, 4   the wave's two LDS-DMA instructions of the step PN_D - 1 ahead (+ cursor bookkeeping)
//   the other eight gaps: PN_SPG slices each of the PREVIOUS tile's epilogue
// The epilogue of a 32 x 32 sub-tile (lane = row, registers = four runs of four columns) is PN_NSL slices of about five
// instructions: bias reads, then per run of four columns unscale + bias / ReLU + magnitude / sign bits / row-major turn
// through LDS, then the mask word, the row-major reads and four whole-line stores.
#ifndef PN_LAB
#define PN_LAB 0  // lab builds (tools/lab/panel_lab.sh): 1 no epilogue slices, 2 no weight DMA / waits, 8 no MFMAs,
#endif            // 16 the panel is loaded once, 32 no epilogue stores, 64 / 512.. no epilogue arithmetic, 128 no vmcnt waits,
                  // 256 no row-major turn, 16384 no fragment reads (all give garbage results), 65536 plain instead of nontemporal stores
constexpr int PN_NSL = 28;                    // slices per sub-tile
constexpr int PN_EGAPS = 8;                   // epilogue gaps per k-step
constexpr int pn_spg(int KB) { return (4 * PN_NSL + KB * PN_EGAPS - 1) / (KB * PN_EGAPS); }  // slices per gap
// epilogue gap index (0..7) of MFMA gap m, or -1
constexpr int pn_egap(int m) { return m == 3 ? 0 : (m >= 5 ? m - 4 : -1); }
// VMEM operations slice S of a sub-tile issues
constexpr int pn_slice_ops(int S, bool masks) { return S == 22 ? (masks ? 1 : 0) : ((S == 24 || S == 25) ? 2 : 0); }
// VMEM operations the epilogue slices issue in k-step kb of a tile, from epilogue gap eg0 on (compile-time schedule)
constexpr int pn_step_ops(int KB, int kb, bool masks, int eg0) {
  int n = 0;
  const int spg = pn_spg(KB);
  for (int eg = eg0; eg < PN_EGAPS; ++eg)
    for (int q = 0; q < spg; ++q) {
      const int e = (kb * PN_EGAPS + eg) * spg + q;
      if (e < 4 * PN_NSL) n += pn_slice_ops(e % PN_NSL, masks);
    }
  return n;
}
// What a wave has issued after the SECOND weight DMA (gap 4) of the pair it waits for at the top of k-step kb: the
// PN_W - 1 later pairs and the epilogue stores of this tile's slices (cur) / of the tile before (prev) -- of the awaited
// pair's own step only those behind gap 4 (epilogue gap 0 = MFMA gap 3 stands BEFORE it: counting its stores would let
// the wait pass with that DMA still in flight).  Not counted: the bias DMA of waves 0 / 1 (they wait for one
// operation more than they must).
constexpr int pn_younger(int KB, int kb, bool masks, bool cur, bool prev) {
  int n = 2 * (PN_W - 1);
  for (int d = 1; d <= PN_W; ++d) {  // step kb - d
    const int k = kb - d;
    if (k >= 0 ? cur : prev) n += pn_step_ops(KB, k >= 0 ? k : k + KB, masks, d == PN_W ? 1 : 0);  // (PN_W <= KB)
  }
  return n;
}

template <int KB, bool MASKS>
__global__ __launch_bounds__(256, 1) void gemm_panel_kernel(const PanelLaunch Larg) {
  static_assert(KB >= PN_W && KB >= 10 && KB <= 20, "the panel kernel holds 10..20 k-blocks per row in registers");
  constexpr int SPG = pn_spg(KB);
  static_assert(KB * PN_EGAPS * SPG >= 4 * PN_NSL, "the epilogue must end inside the tile");
  typedef const __attribute__((address_space(4))) PanelLaunch KLaunch;
  KLaunch& L = *(KLaunch*)__builtin_amdgcn_kernarg_segment_ptr();
  __shared__ __attribute__((aligned(16))) float lds[PN_LDS_BYTES / 4];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, h = lane >> 5;
  const uint32_t lds0 = lds_byte_addr(lds);
  const int M = L.M;
  const int npairs = L.n_half >> 1;
  const int npanels = M / PN_BM;

  if (tid < 2 * MML_MAX_GROUP) {  // magnitude words (+ the dummy word behind them) and 1 / scale of the weights
    const uint32_t a = lds0 + PN_AMAX_OFF + tid * 4, z = 0u;
    asm volatile("ds_write_b32 %0, %1" ::"v"(a), "v"(z) : "memory");
  }
  if (tid < MML_MAX_GROUP) {
    // the weights' exponents, once: a global load inside the tile loop would make hipcc drain the LDS-DMA ring there
    const float ib = tid < L.n_prob ? pn_pow2(-*L.p[tid].kexp) : 1.f;
    const uint32_t b = lds0 + PN_INVB_OFF + tid * 4;
    asm volatile("ds_write_b32 %0, %1" ::"v"(b), "v"(ib) : "memory");
  }
  if (tid < 128) {  // the zero area a problem without bias reads its bias from
    const uint32_t a = lds0 + PN_ZERO_OFF + tid * 4, z = 0u;
    asm volatile("ds_write_b32 %0, %1" ::"v"(a), "v"(z) : "memory");
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  // power-of-two scale of the activations (one per launch: every problem reads the same panel)
  const int kA = __builtin_amdgcn_readfirstlane(pn_scale_exp(pn_amax_load(L.amaxA)));
  const float sA = pn_pow2(kA), invA = pn_pow2(-kA);

  // ---- weight fragment reads (the image of gemm.hip: 64-byte rows, 16-byte chunk c of row r at c ^ ((r >> 2) & 3)) ----
  const int swz = (l31 >> 2) & 3;
  const uint32_t aBh = lds0 + PN_RING_OFF + l31 * 64 + ((h ^ swz) * 16);
  const uint32_t aBl = lds0 + PN_RING_OFF + l31 * 64 + (((2 + h) ^ swz) * 16);
  struct FragB {
    f32x4_t bh[4], bl[4];
  };
  auto landed_b = [&](FragB& f) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      lds_landed(f.bh[i]);
      lds_landed(f.bl[i]);
    }
  };

  // ---- DMA lane geometry: wave-instruction t of a 128-row image covers rows 16 t .. 16 t + 15, lane -> (row, phys chunk) ----
  const int drow = lane >> 2;                          // row inside the 16-row group
  const int dchunk = (lane & 3) ^ ((drow >> 2) & 3);   // logical chunk that lands at physical chunk lane & 3

  // ---- context of a half tile whose epilogue is in progress (all wave-uniform) ----
  struct Ctx {
    float* C;           // &C[0][col0]
    uint32_t* mask;     // &mask[0][col0 / 32]
    int ldc, ldmask;
    float inv;          // 2^-(kA + kB)
    uint32_t bias_at;   // LDS byte address of the tile pair's 128 bias values -- or of the zero area
    uint32_t amax_at;   // LDS byte address of the problem's magnitude word -- or of the dummy word
  };
  auto load_ctx = [&](Ctx& c, const int pair, const int j, const int par) __attribute__((always_inline)) {
    const int hw_ = L.half[2 * pair + j], pi = hw_ >> 16, col0 = hw_ & 0xffff;
    c.ldc = (int)L.p[pi].ldc;
    c.ldmask = (int)L.p[pi].ldmask;
    c.C = L.p[pi].C + col0;
    c.mask = L.p[pi].mask + (col0 >> 5);
    float ib;
    {
      const uint32_t b = lds0 + PN_INVB_OFF + (uint32_t)pi * 4u;
      asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(ib) : "v"(b) : "memory");
    }
    c.inv = invA * __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, ib)));
    c.bias_at = lds0 + (L.p[pi].bias ? PN_BIAS_OFF + par * 512 : PN_ZERO_OFF);
    c.amax_at = lds0 + PN_AMAX_OFF + (L.p[pi].amax_out ? pi : MML_MAX_GROUP) * 4;
  };

  // weight ring: source pointers of this wave's two DMA instructions per stage (j = half tile j of the pair)
  const float* pb[2];
  auto setup_b = [&](const int pair) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int hw_ = L.half[2 * pair + j], pi = hw_ >> 16, col0 = hw_ & 0xffff;
      const float* base = reinterpret_cast<const float*>(L.p[pi].planes);
      pb[j] = base + (int64_t)(col0 + 16 * wave + drow) * L.p[pi].ldp + 4 * dchunk;
    }
  };
  // The 128 bias values of a tile pair travel to LDS by one 4-byte LDS-DMA per lane of waves 0 / 1 (half tile 0 / 1), in
  // k-step 1 of the tile: its epilogue reads them from step 0 of the NEXT tile on, behind PN_W later weight-DMA waits of
  // the issuing wave and as many barriers (issued in the tile's last step -- the first hand-placed form -- they were
  // read one step later: wrong bias in a few panels under load).  Two buffers, alternating by TILE (not by pair: the
  // last pair of a panel and the first of the next may have the same parity).
  auto bias_dma = [&](const int pair, const int par) __attribute__((always_inline)) {
    if (wave < 2) {
      const int hw_ = L.half[2 * pair + wave], pi = hw_ >> 16, col0 = hw_ & 0xffff;
      const float* bias = L.p[pi].bias;
      const float* src = bias ? bias + col0 + lane : L.A;
      dma4(src, lds + (PN_BIAS_OFF + par * 512 + wave * 256) / 4);
    }
  };

  // ---- the wave's 32 panel rows as MFMA fragments: KB x (h plane, l plane) ----
  f16x8 Ah[KB], Al[KB];
  auto load_panel = [&](const int row0) __attribute__((always_inline)) {
    const int row = row0 + 32 * wave + l31;
    const float* ar = L.A + (int64_t)row * L.lda + 4 * h;
    float4 q0[KB], q1[KB];
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
      q0[kb] = *reinterpret_cast<const float4*>(ar + 16 * kb);
      q1[kb] = *reinterpret_cast<const float4*>(ar + 16 * kb + 8);
    }
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
      const float x[8] = {q0[kb].x, q0[kb].y, q0[kb].z, q0[kb].w, q1[kb].x, q1[kb].y, q1[kb].z, q1[kb].w};
      f16x8 hh, ll;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float y = x[e] * sA;          // exact: sA is a power of two
        const _Float16 hv = (_Float16)y;    // round to nearest even
        const float r = y - (float)hv;      // exact
        hh[e] = hv;
        ll[e] = (_Float16)r;
      }
      Ah[kb] = hh;
      Al[kb] = ll;
    }
    // (all of it HERE: hipcc would otherwise sink the cut of block kb -- and the wait for its loads -- into k-step kb)
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
      asm volatile("" : "+v"(Ah[kb]));
      asm volatile("" : "+v"(Al[kb]));
    }
  };

  pf32x16 acc[4], epi[4];
  Ctx ectx[2];
  float am_f[2] = {0.f, 0.f};
  int row0 = 0;       // of the panel being computed
  uint32_t erow0 = 0; // of the panel the tile in `epi` belongs to

  // ---- epilogue state that lives across slices ----
  f32x4_t eb[4];      // bias of the four column runs of the sub-tile
  f32x4_t ev[4];      // a run of four outputs (pre-turn), then the four row-major pieces
  uint32_t emw = 0u;  // sign bits of this lane's 16 outputs
  // per-lane addresses of the epilogue, computed once (a handful of registers; the first form recomputed them from an
  // opaque lane number in every slice, ~8 instructions each time)
  const uint32_t scr = lds0 + PN_SCR_OFF + wave * PN_SCR;
  uint32_t e_wr[4];   // turn area, write side: 16-byte chunk c = 2 g + h of row r at c ^ (r & 7)
#pragma unroll
  for (int g = 0; g < 4; ++g) e_wr[g] = scr + l31 * 128 + (((2 * g + h) ^ (l31 & 7)) * 16);
  const int eR = lane >> 3, ecc = lane & 7;
  const uint32_t e_rd = scr + eR * 128 + ((ecc ^ eR) * 16);  // read side: lane (R, cc) takes chunk cc of rows R + 8 p
  const uint32_t e_bias = (uint32_t)(4 * h) * 4u;             // + bias_at + NI * 128: the lane's columns 8 g + 4 h + j
  const uint32_t e_row = (uint32_t)(32 * wave + eR);          // row of the panel this lane stores (+ 8 p)
  const uint32_t e_mrow = (uint32_t)(32 * wave + l31);        // row of the panel whose mask word this lane holds
  const uint32_t e_sh = (uint32_t)(4 * h);
  // One slice (S of PN_NSL) of the epilogue of sub-tile NI of the tile in `epi`: at most ~5 instructions, no branch.
  //   0 bias reads | 1 wait | 2 + 5 g + {0, 1: unscale + bias, 2: ReLU, 3: magnitude + sign tests, 4: sign bits + turn}
  //   22 mask word | 23 row-major reads | 24, 25 stores | 26 magnitude word
  auto eslice = [&](auto nic, auto sc) __attribute__((always_inline)) {
    constexpr int NI = decltype(nic)::value, S = decltype(sc)::value, HT = NI >> 1;
    const Ctx& ec = ectx[HT];
    pf32x16& a = epi[NI];
    if constexpr (S == 0) {
      const uint32_t ab = ec.bias_at + e_bias;
      eb[0] = ds_read128<NI * 128>(ab);
      eb[1] = ds_read128<NI * 128 + 32>(ab);
      eb[2] = ds_read128<NI * 128 + 64>(ab);
      eb[3] = ds_read128<NI * 128 + 96>(ab);
      emw = 0u;
    } else if constexpr (S == 1) {
      if constexpr (!(PN_LAB & 256)) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int g = 0; g < 4; ++g) lds_landed(eb[g]);
    } else if constexpr (S >= 2 && S <= 21) {
      constexpr int g = (S - 2) / 5, part = (S - 2) % 5;
      f32x4_t& x = ev[g];
      if constexpr (((PN_LAB & 64) && part != 4) || ((PN_LAB & 512) && part <= 1) || ((PN_LAB & 1024) && part == 2) ||
                    ((PN_LAB & 2048) && part == 3)) {
      } else if constexpr (part == 0) {  // unscale + bias (one exact multiplication by a power of two inside the FMA)
        x.x = a[4 * g] * ec.inv + eb[g].x;
        x.y = a[4 * g + 1] * ec.inv + eb[g].y;
      } else if constexpr (part == 1) {
        x.z = a[4 * g + 2] * ec.inv + eb[g].z;
        x.w = a[4 * g + 3] * ec.inv + eb[g].w;
      } else if constexpr (part == 2) {  // ReLU (v > 0 ? v : 0 -- one v_max_f32: a NaN gives 0 like the comparison)
        asm("v_max_f32 %0, 0, %0" : "+v"(x.x));
        asm("v_max_f32 %0, 0, %0" : "+v"(x.y));
        asm("v_max_f32 %0, 0, %0" : "+v"(x.z));
        asm("v_max_f32 %0, 0, %0" : "+v"(x.w));
      } else if constexpr (part == 3) {  // magnitude (the values are >= 0 now)
        asm("v_max3_f32 %0, %0, %1, %2" : "+v"(am_f[HT]) : "v"(x.x), "v"(x.y));
        asm("v_max3_f32 %0, %0, %1, %2" : "+v"(am_f[HT]) : "v"(x.z), "v"(x.w));
      } else {  // sign bits of columns 8 g + j (v > 0 <=> its bit pattern != 0; shifted by 4 h at the end); the turn
        if (MASKS && !(PN_LAB & 64) && !(PN_LAB & 4096)) {
          const uint32_t u0 = __float_as_uint(x.x), u1 = __float_as_uint(x.y), u2 = __float_as_uint(x.z), u3 = __float_as_uint(x.w);
          uint32_t t0, t1, t2, t3;
          asm("v_min_u32 %0, 1, %1" : "=v"(t0) : "v"(u0));
          asm("v_min_u32 %0, 1, %1" : "=v"(t1) : "v"(u1));
          asm("v_min_u32 %0, 1, %1" : "=v"(t2) : "v"(u2));
          asm("v_min_u32 %0, 1, %1" : "=v"(t3) : "v"(u3));
          asm("v_lshl_or_b32 %0, %1, %2, %0" : "+v"(emw) : "v"(t0), "n"(8 * g));
          asm("v_lshl_or_b32 %0, %1, %2, %0" : "+v"(emw) : "v"(t1), "n"(8 * g + 1));
          asm("v_lshl_or_b32 %0, %1, %2, %0" : "+v"(emw) : "v"(t2), "n"(8 * g + 2));
          asm("v_lshl_or_b32 %0, %1, %2, %0" : "+v"(emw) : "v"(t3), "n"(8 * g + 3));
        }
        if constexpr (!(PN_LAB & 256)) ds_write128(e_wr[g], x);
        else asm volatile("" ::"v"(x));
      }
    } else if constexpr (S == 22) {  // the row's 32-column word: lanes r and r + 32 hold its two interleaved halves
      if (MASKS) {
        const uint32_t mine = emw << e_sh;
        const auto sw = __builtin_amdgcn_permlane32_swap(mine, mine, false, false);
        const uint32_t word = sw[0] | sw[1];
        const uint32_t at = (erow0 + e_mrow) * (uint32_t)ec.ldmask + (NI & 1);
        if constexpr (!(PN_LAB & 32)) ec.mask[at] = word;  // (both lanes of a row store the same word)
        else asm volatile("" ::"v"(word), "v"(at));
      }
    } else if constexpr (S == 23) {
      if constexpr (!(PN_LAB & 256)) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        ev[0] = ds_read128<0>(e_rd);
        ev[1] = ds_read128<1024>(e_rd);
        ev[2] = ds_read128<2048>(e_rd);
        ev[3] = ds_read128<3072>(e_rd);
      }
    } else if constexpr (S == 24 || S == 25) {
      if constexpr (S == 24 && !(PN_LAB & 256)) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int p = 0; p < 4; ++p) lds_landed(ev[p]);
      }
      float* const cp = ec.C + (NI & 1) * 32;
#pragma unroll
      for (int p = 2 * (S - 24); p < 2 * (S - 24) + 2; ++p) {
        const uint32_t at = (erow0 + e_row + 8 * p) * (uint32_t)ec.ldc + 4 * ecc;  // (< 2^32 bytes: checked on the host)
        // (nontemporal: the outputs are read next by another kernel, long after this line has left the L2 -- 180 -> 172 us)
        if constexpr (PN_LAB & 65536) *reinterpret_cast<float4*>(cp + at) = make_float4(ev[p].x, ev[p].y, ev[p].z, ev[p].w);
        else if constexpr (!(PN_LAB & 32)) __builtin_nontemporal_store(ev[p], reinterpret_cast<f32x4_t*>(cp + at));
        else asm volatile("" ::"v"(ev[p]), "v"(at));
      }
    } else if constexpr (S == 26) {
      if constexpr ((NI & 1) == 1) {  // the half tile's largest |output| -> the workgroup's word of its problem (or the dummy)
        const uint32_t am_bits = __float_as_uint(am_f[HT]);
        asm volatile("ds_max_u32 %0, %1" ::"v"(ec.amax_at), "v"(am_bits) : "memory");
        am_f[HT] = 0.f;
      }
    }
  };
  // epilogue slice number e of the tile (0 .. 4 PN_NSL - 1)
  auto eslice_at = [&](auto ec_) __attribute__((always_inline)) {
    constexpr int e = decltype(ec_)::value;
    if constexpr (e < 4 * PN_NSL) eslice(std::integral_constant<int, e / PN_NSL>{}, std::integral_constant<int, e % PN_NSL>{});
  };

  // ---- the weight stream: one cursor over (pair, k-block), cyclic over the pairs; it does not know about panels and
  // runs PN_D - 1 steps past the end (a few KB of weights nobody reads: no branch in the loop for it) ----
  const int my_panels = (npanels - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
  if (my_panels <= 0) return;
  int dpair = 0, dstage = 0;
  auto dma_one = [&](const int j) __attribute__((always_inline)) {
    dma16(pb[j], lds + (PN_RING_OFF + dstage * PN_STAGE) / 4 + (wave + 4 * j) * 256);
    pb[j] += 16;
  };
  auto next_pair = [&]() __attribute__((always_inline)) {
    dpair = dpair + 1 == npairs ? 0 : dpair + 1;
    setup_b(dpair);
  };
  setup_b(0);
  {
    int dkb = 0;
    for (int st = 0; st < PN_D - 1; ++st) {
      dma_one(0);
      dma_one(1);
      dstage = dstage + 1;
      if (++dkb == KB) {
        dkb = 0;
        next_pair();
      }
    }
  }

  FragB fb;            // the weight fragments of the step about to run
  int rstage = 0;      // ring stage of the step about to run

  // ---- one tile: KB k-steps.  EPI: the slices of the tile in `epi` run in its gaps; PREV: the tile before ran slices too ----
  auto tile = [&](auto epic, auto prevc, const int pair, const int par) __attribute__((always_inline)) {
    constexpr bool EPI = decltype(epic)::value, PREV = decltype(prevc)::value;
    pn_static_for<0, KB>([&](auto kbc) __attribute__((always_inline)) {
      constexpr int kb = decltype(kbc)::value;
      // ---- the weights of the next step (issued PN_W steps ago) have landed; then the workgroup's barrier ----
      if constexpr (!(PN_LAB & 2) && !(PN_LAB & 128)) {
        constexpr int N = pn_younger(KB, kb, MASKS, EPI && !(PN_LAB & 1), PREV && !(PN_LAB & 1));
        static_assert(N <= 63, "vmcnt is six bits wide");
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
      }
      if constexpr (!(PN_LAB & 4)) __builtin_amdgcn_s_barrier();
      FragB nf;
      const int nstage = rstage + 1 == PN_D ? 0 : rstage + 1;
      const uint32_t sb = (uint32_t)nstage * PN_STAGE;
      __builtin_amdgcn_sched_barrier(0);
      pn_static_for<0, 12>([&](auto mc) __attribute__((always_inline)) {
        // (the three dependent MFMAs of a product block stand four apart: with one wave per SIMD a dependent MFMA issued
        // right behind its producer waits for the producer's whole latency -- measured: the loop ran at half speed)
        constexpr int m = decltype(mc)::value, ni = m % 4, j = m / 4;
        if constexpr (!(PN_LAB & 8)) {
          const f16x8 bh = __builtin_bit_cast(f16x8, fb.bh[ni]), bl = __builtin_bit_cast(f16x8, fb.bl[ni]);
          if constexpr (j == 0) {
            pf32x16 c0;
            if constexpr (kb == 0) {
#pragma unroll
              for (int r = 0; r < 16; ++r) c0[r] = 0.f;
            } else {
              c0 = acc[ni];
            }
            acc[ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl, Ah[kb], c0, 0, 0, 0);
          } else if constexpr (j == 1) {
            acc[ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh, Al[kb], acc[ni], 0, 0, 0);
          } else {
            if constexpr (kb == KB - 1) epi[ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh, Ah[kb], acc[ni], 0, 0, 0);
            else acc[ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh, Ah[kb], acc[ni], 0, 0, 0);
          }
        }
        // ---- what stands in this MFMA's shadow ----
        if constexpr ((PN_LAB & 16384) && (m == 0 || m == 1)) {
          if constexpr (m == 0) nf = fb;
        } else if constexpr (m == 0) {
          nf.bh[0] = ds_read128<0>(aBh + sb);
          nf.bl[0] = ds_read128<0>(aBl + sb);
          nf.bh[1] = ds_read128<2048>(aBh + sb);
          nf.bl[1] = ds_read128<2048>(aBl + sb);
        } else if constexpr (m == 1) {
          nf.bh[2] = ds_read128<4096>(aBh + sb);
          nf.bl[2] = ds_read128<4096>(aBl + sb);
          nf.bh[3] = ds_read128<6144>(aBh + sb);
          nf.bl[3] = ds_read128<6144>(aBl + sb);
        } else if constexpr (m == 2) {
          if constexpr (!(PN_LAB & 2)) dma_one(0);
          if constexpr (kb == 1) bias_dma(pair, par);
        } else if constexpr (m == 4) {
          if constexpr (!(PN_LAB & 2)) {
            dma_one(1);
            dstage = dstage + 1 == PN_D ? 0 : dstage + 1;
            // (the cursor is PN_D - 1 steps ahead: it leaves its tile at a step known at compile time)
            if constexpr ((kb + PN_D - 1) % KB == KB - 1) next_pair();
          }
        } else {
          constexpr int eg = pn_egap(m);
          if constexpr (EPI && !(PN_LAB & 1)) {
            pn_static_for<0, SPG>([&](auto qc) __attribute__((always_inline)) {
              eslice_at(std::integral_constant<int, (kb * PN_EGAPS + eg) * SPG + decltype(qc)::value>{});
            });
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      });
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      landed_b(nf);
      __builtin_amdgcn_sched_barrier(0);
      fb = nf;
      rstage = nstage;
    });
    if constexpr (PN_LAB != 0) {  // (lab builds: whatever is switched off, the products stay live)
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) asm volatile("" ::"v"(epi[ni]));
    }
    // the finished tile sits in `epi`: its slices go into the next tile's gaps (or the flush at the end)
    load_ctx(ectx[0], pair, 0, par);
    load_ctx(ectx[1], pair, 1, par);
    erow0 = (uint32_t)row0;
  };
  using TT = std::true_type;
  using FF = std::false_type;

  int tiles_done = 0;
  for (int panel = blockIdx.x; panel < npanels; panel += gridDim.x) {
    row0 = panel * PN_BM;
    if (!(PN_LAB & 16) || tiles_done == 0) load_panel(row0);
    if (tiles_done == 0) {  // the first fragments of the workgroup (later ones are read one step ahead, across panels too)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      fb.bh[0] = ds_read128<0>(aBh);
      fb.bl[0] = ds_read128<0>(aBl);
      fb.bh[1] = ds_read128<2048>(aBh);
      fb.bl[1] = ds_read128<2048>(aBl);
      fb.bh[2] = ds_read128<4096>(aBh);
      fb.bl[2] = ds_read128<4096>(aBl);
      fb.bh[3] = ds_read128<6144>(aBh);
      fb.bl[3] = ds_read128<6144>(aBl);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      landed_b(fb);
      __builtin_amdgcn_sched_barrier(0);
    }
    // (at a panel switch hipcc has waited with vmcnt(0) for the panel's loads: counts that still include operations of
    // the panel before only over-estimate what is in flight by operations that are done)
    for (int pair = 0; pair < npairs; ++pair) {
      if (tiles_done == 0) tile(FF{}, FF{}, pair, 0);
      else if (tiles_done == 1) tile(TT{}, FF{}, pair, 1);
      else tile(TT{}, TT{}, pair, tiles_done & 1);
      ++tiles_done;
    }
  }
  // the last tile's epilogue, with nothing beside it
  if constexpr (!(PN_LAB & 1))
    pn_static_for<0, 4 * PN_NSL>([&](auto ec_) __attribute__((always_inline)) {
      eslice_at(ec_);
      __builtin_amdgcn_sched_barrier(0);
    });

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
  // the instantiated panel widths: every K = 16 k from 160 to 320 (the caller pads other reduction lengths with zero
  // columns on both operands -- engine.py, Val.kpad: AE with its 63 dense columns, K0 = 303, arrives here as 304)
  if (d0.K % 16 != 0 || d0.K < 160 || d0.K > 320) return MML_ERR_UNSUPPORTED;
  if (d0.M < PN_BM * 64) return MML_ERR_UNSUPPORTED;  // (small batches: the tile kernel)
  if (d0.M % PN_BM != 0) {
    // ragged batch: the whole panels here, the last M % 128 rows through the tile kernel (a row's result does not depend
    // on the tile it is computed in: same planes, same product and k order -- tests/test_gemm_panel_gpu.py)
    mml_gemm_fwd_desc head[MML_MAX_GROUP], tail[MML_MAX_GROUP];
    const int64_t mh = d0.M - d0.M % PN_BM;
    for (int i = 0; i < n; ++i) {
      head[i] = tail[i] = d[i];
      head[i].M = (int32_t)mh;
      tail[i].M = d[i].M - (int32_t)mh;
      tail[i].A = d[i].A + mh * d[i].lda;
      tail[i].C = d[i].C + mh * d[i].ldc;
      if (d[i].relu_mask) tail[i].relu_mask = d[i].relu_mask + mh * d[i].ldmask;
      if (d[i].M != d0.M) return MML_ERR_UNSUPPORTED;
    }
    const int rc = mml_gemm_panel_try_fwd(head, n, st);
    if (rc != MML_OK) return rc;
    return mml_gemm_grouped_fwd(tail, n, (void*)st);  // (M < 128: never comes back here)
  }
  if (!d0.amax_a || !aligned16(d0.A) || d0.lda % 4 != 0) return MML_ERR_UNSUPPORTED;
  int halves = 0, masks = 0, relus = 0;
  for (int i = 0; i < n; ++i) {
    const mml_gemm_fwd_desc& q = d[i];
    if (q.mul || q.prod) return MML_ERR_UNSUPPORTED;  // (K7 products: the tile kernel's epilogue)
    if (q.A != d0.A || q.lda != d0.lda || q.M != d0.M || q.K != d0.K || q.amax_a != d0.amax_a) return MML_ERR_UNSUPPORTED;
    if (q.w_kn != 0 || !q.w_planes || !q.w_kexp || !aligned16(q.w_planes) || q.ldw % 4 != 0) return MML_ERR_UNSUPPORTED;
    if (q.N % 64 != 0 || q.N < 64) return MML_ERR_UNSUPPORTED;
    if (q.act != MML_ACT_RELU) return MML_ERR_UNSUPPORTED;
    if (!aligned16(q.C) || q.ldc % 4 != 0 || (q.bias && !aligned16(q.bias))) return MML_ERR_UNSUPPORTED;
    if ((int64_t)q.M * q.ldc >= (1ll << 30) || (int64_t)q.M * q.ldmask >= (1ll << 30)) return MML_ERR_UNSUPPORTED;  // (32-bit offsets)
    const bool m = q.act == MML_ACT_RELU && q.relu_mask != nullptr;
    if (m && q.ldmask * 32 < q.N) return MML_ERR_UNSUPPORTED;
    masks += m ? 1 : 0;
    relus += 1;
    halves += q.N / 64;
  }
  if (halves > PN_MAXH || halves % 2 != 0) return MML_ERR_UNSUPPORTED;
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
      L.half[hidx] = (i << 16) | c;
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
  const dim3 g((unsigned)(npanels < cus ? npanels : cus)), b(256);
#define PN_GO(KB_)                                                                    \
  do {                                                                                \
    if (L.store_masks) MML_LAUNCH((gemm_panel_kernel<KB_, true>), g, b, 0, st, L);    \
    else MML_LAUNCH((gemm_panel_kernel<KB_, false>), g, b, 0, st, L);                 \
  } while (0)
  switch (L.K / 16) {
    case 10: PN_GO(10); break;
    case 11: PN_GO(11); break;
    case 12: PN_GO(12); break;
    case 13: PN_GO(13); break;
    case 14: PN_GO(14); break;
    case 15: PN_GO(15); break;
    case 16: PN_GO(16); break;
    case 17: PN_GO(17); break;
    case 18: PN_GO(18); break;
    case 19: PN_GO(19); break;
    default: PN_GO(20); break;
  }
#undef PN_GO
  return check_launch("mml_gemm_grouped_fwd(panel)");
}
