// K3p: Linear(+activation) forward on operands that arrive ALREADY cut into bf16 planes.
//
// gemm.hip emulates the fp32 product with three bf16 planes per operand (x = h + m + l, exact) and six
// v_mfma_f32_32x32x16_bf16 per 16-k block; it cuts every fragment in registers, in every wave that uses it -- in the
// first expert layer an activation is cut 16 times and a weight a thousand times, and the cuts (not the MFMAs) bound
// the kernel (9.6 VALU per MFMA, MFMA pipe 47 % busy).  Here the cut happens ONCE, where the value is
// produced (mml_planes_cut; weights once per step), and the GEMM only moves planes: global -> LDS by DMA, LDS ->
// registers with one ds_read_b128 per plane and fragment, six MFMAs per block -- no VALU work in the loop but addresses.
//
// Geometry: planes are 1.5x the bytes of fp32, so the LDS budget is spent differently from gemm_pipe_kernel:
// ONE workgroup of 8 waves per CU (two waves per SIMD, as there), tile 256 x 128: the 128-column weight tile is shared
// by 4 row-waves instead of 2.  Stage = A 3 x 8 KiB + B 3 x 4 KiB = 36 KiB, four stages = 144 KiB of the 160 KiB.
//   Memory layout of a cut matrix ("plane panels"): [K/8 panels][rows][3 planes][8 bf16] -- a panel holds eight
//   consecutive columns of every row, the h, m, l chunks of a row side by side (48 bytes), rows back to back.  The
//   256 x 8 piece of a tile is then 12 KiB of CONTIGUOUS memory and every DMA wave-instruction reads 1 KiB = 8 whole
//   cache lines.  This is what the kernel's speed hangs on: the CU <-> L2 path moves ~2.5 clocks per 128-byte line
//   touched, used or not (tools/lab/micro/dma_rate.hip: 113 GB/s per CU for contiguous lanes, 13 GB/s for one 16-byte
//   piece per line).  Plane-major storage [3][rows][K] (64 lines per instruction) ran the first AE-30 layer in 426 us,
//   row-interleaved [rows][K/8][3][8] (~30 lines) in 349 us, against 189 us with the DMA removed.
//   LDS image of one operand: [2 k-groups][ROWS][3 planes][16 B]: a lane's MFMA fragment of one plane -- 8 consecutive
//   k of one row -- is one 16-byte chunk at row * 48 + plane * 16; 16 consecutive rows hit 16 distinct 16-byte bank
//   groups (3 is coprime with 16): conflict-free ds_read_b128.  Written by the DMA in lane order (lane l of DMA
//   instruction t fetches chunk 64 t + l of the k-group: row = chunk / 3, plane = chunk % 3).
// Schedule (as gemm_pipe_kernel): DMA of step i+3 issued at the top of step i, fragments of step i+1 read at the top
// of step i into the other register set, one s_barrier per step, counted s_waitcnt vmcnt.
// Results agree with gemm_pipe_kernel's three-plane form to fp32 accumulation-order noise (same planes, same six
// products, same k-step order; the k <-> lane-half assignment inside a 16-k block differs).
#include "lds_async.hpp"

#include <stdlib.h>

#include <type_traits>

namespace mml {

using f32x16q = __attribute__((ext_vector_type(16))) float;

constexpr int QM = 256, QN = 128, QK = 16, QSTAGES = 4;
constexpr int QA_KG = QM * 48;                         // bytes of one k-group of the A stage: 256 rows x 3 planes x 16 B
constexpr int QB_KG = QN * 48;
constexpr int QA_STAGE = 2 * QA_KG;
constexpr int QSTAGE = QA_STAGE + 2 * QB_KG;           // 36 KiB
constexpr int QRING = QSTAGES * QSTAGE;                // 144 KiB
constexpr int QLDS = QRING + 8 * 64 * 4;               // + one 64-float bias slot per wave (dynamic LDS)
constexpr int QLOADS = 5;                              // DMA wave-instructions per wave and k-step (3 A + 2 B)
constexpr int QNSTORE = 16;                            // 16-byte output stores per wave of an interior tile

struct QProblem {
  const uint16_t* A;   // interleaved planes of the [M, K] row operand
  const uint16_t* W;   // interleaved planes of the [N, K] column operand
  int64_t pa_step, pw_step;  // elements between consecutive panels: 24 x rows
  float* C;
  int64_t ldc;
  const float* bias;
  uint32_t* mask;      // relu sign bits out (or null)
  int64_t ldmask;
  int32_t M, N, K16;   // K16: reduction extent rounded up to the 16-wide k-step (pad columns are zero in both operands)
  int32_t act;
  int32_t tiles_n, tile0;
};

struct QLaunch {
  QProblem p[MML_MAX_GROUP];
  int32_t n, total_ntiles, tiles_m;
};

struct QFrag {
  f32x4_t p[3];
  __device__ __forceinline__ void landed() { lds_landed(p[0]); lds_landed(p[1]); lds_landed(p[2]); }
};

#ifdef QLAB_TIMES  // per-phase cycle sums of wave 0 of workgroup 0: [0] vmcnt wait [1] barrier [2] body [3] lgkm wait
                   // [4] epilogue [5] steps [6] tiles
__device__ unsigned long long g_qlab_t[8];
#define QT(var) unsigned long long var = __builtin_readcyclecounter()
#else
#define QT(var)
#endif

template <int ACT>
__device__ __forceinline__ float qact(float v) {
  if (ACT == MML_ACT_RELU) return v > 0.f ? v : 0.f;
  if (ACT == MML_ACT_SIGMOID) return 1.f / (1.f + __expf(-v));
  if (ACT == MML_ACT_SIGMOID2) return 2.f / (1.f + __expf(-v));
  return v;
}

__global__ __launch_bounds__(512) void gemm_planes_kernel(const QLaunch Larg) {
  typedef const __attribute__((address_space(4))) QLaunch KLaunch;
  KLaunch& L = *(KLaunch*)__builtin_amdgcn_kernarg_segment_ptr();  // descriptors stay in the kernel-argument block
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x;
  // wave as a SCALAR (readfirstlane once, here): everything derived from it -- the LDS destinations of the DMA above all
  // -- then lives in SGPRs.  As a VGPR expression every DMA needed a v_readfirstlane -> s_mov m0 in the loop, and that
  // VALU -> SALU hand-over waits for the wave's MFMAs in flight (~100 cycles per DMA, 5 per step).
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, h = lane >> 5;
  const int64_t total = (int64_t)L.tiles_m * L.total_ntiles;

  struct Cursor {
    int64_t vid;
    int pi, row0, col0, k0, kend, M, N;
    int left;  // k-steps of this tile still to go (the k-loop starts at a per-column-tile offset and wraps, see decode)
    bool ok;
    bool short_tile;  // fewer than three k-steps: the bias DMA may still be in flight at the epilogue
    bool counted;     // interior tile: its epilogue leaves exactly QNSTORE (+ mask) stores in flight
  };
  auto decode = [&](Cursor& c) __attribute__((always_inline)) {
    c.ok = c.vid < total;
    if (!c.ok) return;
    const int outer = (int)(c.vid / L.total_ntiles);
    int j = (int)(c.vid - (int64_t)outer * L.total_ntiles);
    const int jj = j;
    int pi = 0;
    while (pi + 1 < L.n && j >= L.p[pi + 1].tile0) ++pi;
    j -= L.p[pi].tile0;
    c.pi = pi;
    c.M = L.p[pi].M;
    c.N = L.p[pi].N;
    c.row0 = outer * QM;
    c.col0 = j * QN;
    c.kend = L.p[pi].K16;
    c.left = c.kend / QK;
    // The column tiles of one row panel run side by side on one XCD and read the same A lines; started at the same k
    // they would all wait on the same L2 misses, step after step.  Tile jj starts its (cyclic) k-loop jj / ntiles of
    // the way in: one tile takes the miss, the others find the line in L2.  (The fp32 sum of a column tile is then
    // taken in a rotated order: deterministic, different from tile to tile.)
#ifdef QLAB_NO_STAGGER
    c.k0 = 0;
    (void)jj;
#else
    c.k0 = (int)((int64_t)jj * c.left / L.total_ntiles) * QK;
#endif
    c.short_tile = c.kend < 3 * QK;
    c.counted = c.row0 + QM <= c.M && c.col0 + QN <= c.N;
  };
  auto advance = [&](Cursor& c) __attribute__((always_inline)) -> bool {  // true: the source pointers must be set up again
    if (--c.left > 0) {
      c.k0 += QK;
      if (c.k0 < c.kend) return false;
      c.k0 = 0;  // wrapped
      return true;
    }
    c.vid += gridDim.x;
    decode(c);
    return true;
  };

  // ---- LDS-DMA.  A k-group of the A stage is 768 chunks = 12 wave-instructions, 24 per stage: wave w issues numbers
  // w, w + 8, w + 16.  The B stage is 2 x 6 instructions: waves 0..3 issue numbers w and w + 8, waves 4..7 number w and
  // the same one again (every wave issues the same number of VMEM operations: the counted waits rely on it). ----
  const uint16_t* pa[3];
  const uint16_t* pb[2];
  int dstA[3], dstB[2];  // byte offsets inside a stage (wave-uniform)
  int64_t incA = 0, incB = 0;
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const int t = wave + 8 * j;
    dstA[j] = (t / 12) * QA_KG + (t % 12) * 1024;
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int u = (j == 0 || wave >= 4) ? wave : wave + 8;
    dstB[j] = QA_STAGE + (u / 6) * QB_KG + (u % 6) * 1024;
  }
  auto setup_ptrs = [&](const Cursor& c) __attribute__((always_inline)) {
    if (!c.ok) return;
    const uint16_t* A = L.p[c.pi].A;
    const uint16_t* W = L.p[c.pi].W;
    incA = 2 * L.p[c.pi].pa_step;  // one k-step = two panels
    incB = 2 * L.p[c.pi].pw_step;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int t = wave + 8 * j;
      const int q = (t % 12) * 64 + lane;
      int row = c.row0 + q / 3;
      row = row < c.M ? row : c.M - 1;
      pa[j] = A + (int64_t)(c.k0 / 8 + t / 12) * L.p[c.pi].pa_step + (int64_t)row * 24 + (q % 3) * 8;
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int u = (j == 0 || wave >= 4) ? wave : wave + 8;
      const int q = (u % 6) * 64 + lane;
      int col = c.col0 + q / 3;
      col = col < c.N ? col : c.N - 1;
      pb[j] = W + (int64_t)(c.k0 / 8 + u / 6) * L.p[c.pi].pw_step + (int64_t)col * 24 + (q % 3) * 8;
    }
  };
  char* const smem_b = reinterpret_cast<char*>(smem);
  // one of the QLOADS DMA instructions of a stage (J = 0..2: A, 3..4: B); the step spreads them between its MFMA blocks:
  // issued back to back after the barrier, the 40 instructions of the 8 waves queue up in the address unit and every
  // wave sits in its last issue for most of a microsecond before its first MFMA (measured: no DMA / MFMA overlap)
  auto issue_one = [&](const int stage, auto jc) __attribute__((always_inline)) {
    constexpr int J = decltype(jc)::value;
    char* dst = smem_b + stage * QSTAGE;
    __builtin_amdgcn_sched_barrier(0);
#ifndef QLAB_NO_DMA
    if constexpr (J < 3) dma16(reinterpret_cast<const float*>(pa[J]), reinterpret_cast<float*>(dst + dstA[J]));
    else dma16(reinterpret_cast<const float*>(pb[J - 3]), reinterpret_cast<float*>(dst + dstB[J - 3]));
#endif
    if constexpr (J < 3) pa[J] += incA;
    else pb[J - 3] += incB;
    __builtin_amdgcn_sched_barrier(0);
  };
  auto issue = [&](const int stage) __attribute__((always_inline)) {
    issue_one(stage, std::integral_constant<int, 0>{});
    issue_one(stage, std::integral_constant<int, 1>{});
    issue_one(stage, std::integral_constant<int, 2>{});
    issue_one(stage, std::integral_constant<int, 3>{});
    issue_one(stage, std::integral_constant<int, 4>{});
  };

  // The bias values of this wave's 64 columns travel to LDS by one 4-byte LDS-DMA per lane when the compute cursor
  // enters a tile (always exactly one VMEM operation -- a dummy address without a bias: the counted waits rely on it).
  float* const lds_bias = smem + QRING / 4 + wave * 64;
  auto bias_dma = [&](const Cursor& c) __attribute__((always_inline)) {
    if (!c.ok) return;
    const float* bias = L.p[c.pi].bias;
    int col = c.col0 + wn * 64 + lane;
    col = col < c.N ? col : c.N - 1;
    const float* src = bias ? bias + col : L.p[c.pi].C;
    __builtin_amdgcn_sched_barrier(0);
    dma4(src, lds_bias);
    __builtin_amdgcn_sched_barrier(0);
  };

  f32x16q acc[2][2];
  auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
  };
  zero_acc();

  const uint32_t lds0 = lds_byte_addr(smem);
  const uint32_t aA = lds0 + h * QA_KG + (wm * 64 + l31) * 48;
  const uint32_t aB = lds0 + QA_STAGE + h * QB_KG + (wn * 64 + l31) * 48;
  auto read_a = [&](const uint32_t so, auto mic, QFrag& f) __attribute__((always_inline)) {
    constexpr int MI = decltype(mic)::value;
    f.p[0] = ds_read128<MI * 1536>(aA + so);
    f.p[1] = ds_read128<MI * 1536 + 16>(aA + so);
    f.p[2] = ds_read128<MI * 1536 + 32>(aA + so);
  };
  auto read_b = [&](const uint32_t so, auto nic, QFrag& f) __attribute__((always_inline)) {
    constexpr int NI = decltype(nic)::value;
    f.p[0] = ds_read128<NI * 1536>(aB + so);
    f.p[1] = ds_read128<NI * 1536 + 16>(aB + so);
    f.p[2] = ds_read128<NI * 1536 + 32>(aB + so);
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;

  // ---- epilogue: each 32 x 32 sub-tile is turned row-major through this wave's 4 KiB of the stage the step just
  // consumed (idle until the DMA of the step after next), then a lane moves 16 bytes of a row: whole 128-byte lines
  // per 8 lanes.  16-byte chunk c of row r sits at chunk c ^ (r & 7) (conflict-free both ways). ----
  auto epilogue_act = [&](const Cursor& c, const uint32_t so_epi, auto actc) __attribute__((always_inline)) {
    constexpr int ACT = decltype(actc)::value;
    const int pi = c.pi;
    const int row0 = c.row0, col0 = c.col0, PM = c.M, PN = c.N;
    float* const C = L.p[pi].C;
    const int64_t ldc = L.p[pi].ldc;
    const bool has_bias = L.p[pi].bias != nullptr;
    uint32_t* const mask = (ACT == MML_ACT_RELU) ? L.p[pi].mask : nullptr;
    const int64_t ldmask = L.p[pi].ldmask;
    const uint32_t tb = lds0 + so_epi + wave * 4096;
    const int R = lane >> 3, cc = lane & 7;
    if (c.short_tile) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // bias DMA landed?
#pragma unroll
    for (int sidx = 0; sidx < 4; ++sidx) {
      const int mi = sidx >> 1, ni = sidx & 1;
      const int colg = col0 + wn * 64 + ni * 32 + 4 * cc;  // this lane's 4 columns
      const int rowb = row0 + wm * 64 + mi * 32 + R;       // ... of rows rowb + 8p
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4_t v = {acc[mi][ni][4 * g], acc[mi][ni][4 * g + 1], acc[mi][ni][4 * g + 2], acc[mi][ni][4 * g + 3]};
        ds_write128(tb + l31 * 128 + (((2 * g + h) ^ (l31 & 7)) * 16), v);
      }
      f32x4_t b4 = {0.f, 0.f, 0.f, 0.f};
      if (has_bias) b4 = ds_read128<0>(lds_byte_addr(lds_bias) + (ni * 32 + 4 * cc) * 4);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (has_bias) lds_landed(b4);
      __builtin_amdgcn_sched_barrier(0);
      f32x4_t v[4];
#pragma unroll
      for (int p = 0; p < 4; ++p) v[p] = ds_read128<0>(tb + (R + 8 * p) * 128 + ((cc ^ ((R + 8 * p) & 7)) * 16));
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int p = 0; p < 4; ++p) lds_landed(v[p]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        f32x4_t& x = v[p];
        x.x = qact<ACT>(x.x + b4.x);
        x.y = qact<ACT>(x.y + b4.y);
        x.z = qact<ACT>(x.z + b4.z);
        x.w = qact<ACT>(x.w + b4.w);
        asm volatile("" : "+v"(x));  // computed HERE, for every lane: not sunk into the guarded stores below
      }
      if (mask) {  // eight lanes (cc = 0..7) hold the eight nibbles of a row's 32-column word
        uint32_t wsel = 0u;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
          const f32x4_t& x = v[p];
          uint32_t w = ((x.x > 0.f ? 1u : 0u) | (x.y > 0.f ? 2u : 0u) | (x.z > 0.f ? 4u : 0u) | (x.w > 0.f ? 8u : 0u))
                       << (4 * cc);
          // OR the eight nibbles into the lane with cc == 0 (row shifts by 4, 2, 1 inside the group of 8) ...
          w |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)w, 0x104, 0xf, 0xf, true);
          w |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)w, 0x102, 0xf, 0xf, true);
          w |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)w, 0x101, 0xf, 0xf, true);
          // ... and hand the word of row R + 8p to the lane with cc == p, so that ONE store writes four rows' words
          uint32_t moved = w;
          if (p == 1) moved = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)w, 0x111, 0xf, 0xf, true);
          if (p == 2) moved = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)w, 0x112, 0xf, 0xf, true);
          if (p == 3) moved = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)w, 0x113, 0xf, 0xf, true);
          if (cc == p) wsel = moved;
        }
        const int mrow = rowb + 8 * cc;  // lane cc < 4 owns row R + 8 cc
        const int cg = col0 + wn * 64 + ni * 32;
        if (cc < 4 && mrow < PM && cg < PN) mask[(int64_t)mrow * ldmask + (cg >> 5)] = wsel;
      }
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        const int row = rowb + 8 * p;
        if (row >= PM || colg >= PN) continue;
        *reinterpret_cast<float4*>(C + (int64_t)row * ldc + colg) = make_float4(v[p].x, v[p].y, v[p].z, v[p].w);
      }
    }
  };
  auto epilogue = [&](const Cursor& c, const uint32_t so_epi) __attribute__((always_inline)) {
    switch (L.p[c.pi].act) {
      case MML_ACT_RELU: epilogue_act(c, so_epi, std::integral_constant<int, MML_ACT_RELU>{}); break;
      case MML_ACT_SIGMOID: epilogue_act(c, so_epi, std::integral_constant<int, MML_ACT_SIGMOID>{}); break;
      case MML_ACT_SIGMOID2: epilogue_act(c, so_epi, std::integral_constant<int, MML_ACT_SIGMOID2>{}); break;
      default: epilogue_act(c, so_epi, std::integral_constant<int, MML_ACT_NONE>{}); break;
    }
  };

  // ---- pipeline state: two fragment register sets, indexed by step parity ----
  QFrag FA0[2], FA1[2], FB0[2], FB1[2];

  Cursor cur;
  cur.vid = xcd_remap(blockIdx.x, gridDim.x);
  decode(cur);
  Cursor pf = cur;
  setup_ptrs(pf);
  bias_dma(cur);
  int issued = 0, i = 0;
  int epi_left = 0;  // steps (0..2) during which the stores of a counted epilogue may still be in flight
#pragma unroll
  for (int st = 0; st < 3; ++st)
    if (pf.ok) {
      issue(st);
      if (advance(pf)) setup_ptrs(pf);
      ++issued;
    }
  if (issued >= 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * QLOADS) : "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  read_a(0u, I0{}, FA0[0]);
  read_a(0u, I1{}, FA1[0]);
  read_b(0u, I0{}, FB0[0]);
  read_b(0u, I1{}, FB1[0]);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  FA0[0].landed();
  FA1[0].landed();
  FB0[0].landed();
  FB1[0].landed();
  __builtin_amdgcn_sched_barrier(0);

  auto step = [&](auto par_c) __attribute__((always_inline)) {
    constexpr int P = decltype(par_c)::value, Q = P ^ 1;
    const int sidx = i & (QSTAGES - 1);
    const uint32_t so_cur = (uint32_t)sidx * QSTAGE;
    const uint32_t so_next = (uint32_t)((sidx + 1) & (QSTAGES - 1)) * QSTAGE;
    // Stage i+1 (read below) was issued two steps ago; in steady state only the QLOADS loads of step i+2 are younger.
        // For two steps after a counted epilogue its stores and the next tile's bias DMA are younger too.
    QT(t0);
    if (__builtin_expect((epi_left | (pf.ok ? 0 : 1)) == 0, 1)) {
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(QLOADS) : "memory");
    } else if (pf.ok && epi_left > 0) {
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(QLOADS + QNSTORE + 1) : "memory");
      --epi_left;
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      epi_left = 0;
    }
    QT(t1);
    __builtin_amdgcn_s_barrier();
    QT(t2);
    const bool dma = pf.ok;
    const int st_dma = (sidx + 3) & (QSTAGES - 1);
    // One instruction stream per wave, pinned in this order: 24 MFMAs with the 12 fragment reads of the next step and
    // the 5 DMA instructions of step i+3 dealt between them.  Issued in a bunch after the barrier, the 96 reads (and 40
    // DMAs) of the 8 waves queue up in the LDS / address units and every wave sits in its last issue -- in-order --
    // before its first MFMA: measured, nothing overlapped (DMA alone 173 us, MFMA alone 189 us, together 290 us).
    const uint32_t ra = aA + so_next, rb = aB + so_next;
#ifdef QLAB_NO_MFMA
#define QMMA(MI_, NI_, PB_, PA_) \
  acc[MI_][NI_][(PB_ + PA_) & 15] += (NI_ ? FB1[P] : FB0[P]).p[PB_].x * (MI_ ? FA1[P] : FA0[P]).p[PA_].y
#else
#define QMMA(MI_, NI_, PB_, PA_)                                                                              \
  acc[MI_][NI_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, (NI_ ? FB1[P] : FB0[P]).p[PB_]), \
                                                          __builtin_bit_cast(bf16x8, (MI_ ? FA1[P] : FA0[P]).p[PA_]), \
                                                          acc[MI_][NI_], 0, 0, 0)
#endif
#define QFENCE() __builtin_amdgcn_sched_barrier(0)
    // rows 0..31 of the wave (A0) against both column halves; plane products smallest first: (l,h) (m,m) (h,l) (m,h) (h,m) (h,h)
    // DMA slots every ~5 MFMAs: a wave's LDS-DMA instructions do not pipeline (tools/lab/micro/dma_lat.hip: each
    // further one adds ~80 ns), and a wave that issues the next one too early sits in that issue instead of its MFMAs
#define QDMA(J_) if (dma) issue_one(st_dma, std::integral_constant<int, J_>{})
    QMMA(0, 0, 2, 0); FB0[Q].p[0] = ds_read128<0>(rb); QFENCE();
    QMMA(0, 1, 2, 0); FB0[Q].p[1] = ds_read128<16>(rb); QFENCE();
    QDMA(0);
    QMMA(0, 0, 1, 1); FB0[Q].p[2] = ds_read128<32>(rb); QFENCE();
    QMMA(0, 1, 1, 1); FA0[Q].p[0] = ds_read128<0>(ra); QFENCE();
    QMMA(0, 0, 0, 2); FA0[Q].p[1] = ds_read128<16>(ra); QFENCE();
    QMMA(0, 1, 0, 2); FA0[Q].p[2] = ds_read128<32>(ra); QFENCE();
    QMMA(0, 0, 1, 0); FB1[Q].p[0] = ds_read128<1536>(rb); QFENCE();
    QDMA(1);
    QMMA(0, 1, 1, 0); FB1[Q].p[1] = ds_read128<1536 + 16>(rb); QFENCE();
    QMMA(0, 0, 0, 1); FB1[Q].p[2] = ds_read128<1536 + 32>(rb); QFENCE();
    QMMA(0, 1, 0, 1); FA1[Q].p[0] = ds_read128<1536>(ra); QFENCE();
    QMMA(0, 0, 0, 0); FA1[Q].p[1] = ds_read128<1536 + 16>(ra); QFENCE();
    QMMA(0, 1, 0, 0); FA1[Q].p[2] = ds_read128<1536 + 32>(ra); QFENCE();
    QDMA(2);
    // rows 32..63 (A1)
    QMMA(1, 0, 2, 0); QFENCE();
    QMMA(1, 1, 2, 0); QFENCE();
    QMMA(1, 0, 1, 1); QFENCE();
    QMMA(1, 1, 1, 1); QFENCE();
    QMMA(1, 0, 0, 2); QFENCE();
    QDMA(3);
    QMMA(1, 1, 0, 2); QFENCE();
    QMMA(1, 0, 1, 0); QFENCE();
    QMMA(1, 1, 1, 0); QFENCE();
    QMMA(1, 0, 0, 1); QFENCE();
    QMMA(1, 1, 0, 1); QFENCE();
    if (dma) {
      issue_one(st_dma, std::integral_constant<int, 4>{});
      if (advance(pf)) setup_ptrs(pf);
    }
    QMMA(1, 0, 0, 0); QFENCE();
    QMMA(1, 1, 0, 0); QFENCE();
#undef QDMA
#undef QMMA
#undef QFENCE
    QT(t3);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    QT(t4);
    FB0[Q].landed();
    FA0[Q].landed();
    FB1[Q].landed();
    FA1[Q].landed();
    __builtin_amdgcn_sched_barrier(0);
    const bool tile_end = cur.left == 1;
    if (__builtin_expect(tile_end, 0)) {
      epilogue(cur, so_cur);
      // every LDS read of the epilogue was consumed before its store issued: a counted tile leaves exactly QNSTORE
      // (or, with a sign mask, more) stores in flight and nothing else
      epi_left = cur.counted ? 2 : 0;
      if (!cur.counted) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      zero_acc();
    }
    advance(cur);
    if (__builtin_expect(tile_end, 0)) bias_dma(cur);
    ++i;
#ifdef QLAB_TIMES
    {
      __builtin_amdgcn_sched_barrier(0);
      unsigned long long t5 = __builtin_readcyclecounter();
      if (blockIdx.x == 0 && tid == 0) {
        g_qlab_t[0] += t1 - t0; g_qlab_t[1] += t2 - t1; g_qlab_t[2] += t3 - t2; g_qlab_t[3] += t4 - t3;
        g_qlab_t[4] += t5 - t4; g_qlab_t[5] += 1; g_qlab_t[6] += tile_end ? 1 : 0;
      }
    }
#endif
  };

  while (true) {
    if (!cur.ok) break;
    step(I0{});
    if (!cur.ok) break;
    step(I1{});
  }
}

// ---- fp32 -> planes -------------------------------------------------------------------------------------------------
struct CutLaunch {
  mml_planes_cut_desc d[MML_MAX_GROUP];
  int64_t chunk0[MML_MAX_GROUP + 1];  // first 8-column output chunk of descriptor i in the launch
  int32_t n;
};

// work item = 8 consecutive columns of one OUTPUT row (one 16-byte store per plane)
__global__ __launch_bounds__(256) void planes_cut_kernel(const CutLaunch L) {
  __shared__ int64_t pre[MML_MAX_GROUP + 1];
  for (int i = threadIdx.x; i <= L.n; i += 256) pre[i] = L.chunk0[i];
  __syncthreads();
  const int64_t total = pre[L.n];
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t ch = (int64_t)blockIdx.x * 256 + threadIdx.x; ch < total; ch += stride) {
    int di = 0;
    while (di + 1 < L.n && ch >= pre[di + 1]) ++di;
    const mml_planes_cut_desc& D = L.d[di];
    const int64_t cpr = D.ldp >> 3;
    const int64_t e = ch - pre[di];
    const int64_t orow = e / cpr;
    const int c0 = (int)(e - orow * cpr) * 8;
    float x[8];
    if (!D.transpose) {
      const float* s = D.src + orow * D.ld + c0;
      if (c0 + 8 <= D.cols && aligned16(s)) {
        const float4 q0 = *reinterpret_cast<const float4*>(s), q1 = *reinterpret_cast<const float4*>(s + 4);
        x[0] = q0.x; x[1] = q0.y; x[2] = q0.z; x[3] = q0.w; x[4] = q1.x; x[5] = q1.y; x[6] = q1.z; x[7] = q1.w;
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = (c0 + j < D.cols) ? s[j] : 0.f;
      }
    } else {  // output row = source column `orow`, output columns = source rows c0 .. c0 + 7
#pragma unroll
      for (int j = 0; j < 8; ++j) x[j] = (c0 + j < D.rows) ? D.src[(int64_t)(c0 + j) * D.ld + orow] : 0.f;
    }
    bf16x8 pl[3];
    split_planes<3>(x, pl);
    const int64_t orows = D.transpose ? D.cols : D.rows;
    uint16_t* o = D.planes + ((int64_t)(c0 >> 3) * orows + orow) * 24;  // [panel][row][plane][8]
#pragma unroll
    for (int p = 0; p < 3; ++p) *reinterpret_cast<bf16x8*>(o + p * 8) = pl[p];
  }
}

}  // namespace mml

using namespace mml;

extern "C" int mml_planes_cut(const mml_planes_cut_desc* d, int32_t n, mml_stream_t stream) {
  MML_REQUIRE(n >= 0 && (n == 0 || d), "mml_planes_cut: bad descriptor array");
  int i = 0;
  while (i < n) {
    CutLaunch L{};
    int64_t chunks = 0;
    while (i < n && L.n < MML_MAX_GROUP) {
      const mml_planes_cut_desc& q = d[i];
      MML_REQUIRE(q.src && q.planes && q.rows >= 0 && q.cols >= 0, "mml_planes_cut: descriptor %d malformed", i);
      const int64_t orows = q.transpose ? q.cols : q.rows, ocols = q.transpose ? q.rows : q.cols;
      MML_REQUIRE(q.ldp % 16 == 0 && q.ldp >= ocols && aligned16(q.planes),
                  "mml_planes_cut: descriptor %d: ldp must be a multiple of 16 covering the row, planes 16-byte aligned", i);
      MML_REQUIRE(q.ld >= q.cols, "mml_planes_cut: descriptor %d: ld < cols", i);
      L.chunk0[L.n] = chunks;
      L.d[L.n++] = q;
      chunks += orows * (q.ldp / 8);
      ++i;
    }
    L.chunk0[L.n] = chunks;
    if (chunks == 0) continue;
    int64_t blocks = cdiv(chunks, 256);
    if (blocks > 256 * 16) blocks = 256 * 16;
    MML_LAUNCH(planes_cut_kernel, dim3((unsigned)blocks), dim3(256), 0, to_stream(stream), L);
    int rc = check_launch("mml_planes_cut");
    if (rc) return rc;
  }
  return MML_OK;
}

extern "C" int mml_gemm_planes_fwd(const mml_gemm_planes_fwd_desc* d, int32_t n, mml_stream_t stream) {
  MML_REQUIRE(n >= 0 && (n == 0 || d), "mml_gemm_planes_fwd: bad descriptor array");
  static int lds_ok = 0;
  if (!lds_ok) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_planes_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, QLDS);
    if (e != hipSuccess) {
      set_error("mml_gemm_planes_fwd: cannot reserve %d bytes of LDS: %s", QLDS, hipGetErrorString(e));
      return MML_ERR_HIP;
    }
    lds_ok = 1;
  }
  static int cus = 0;
  if (cus == 0) {
    int dev = 0, c = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || c <= 0)
      c = 256;
    cus = c;
  }
  int i = 0;
  while (i < n) {
    QLaunch L{};
    int j = i, t = 0;
    while (j < n && j - i < MML_MAX_GROUP && d[j].M == d[i].M) {
      const mml_gemm_planes_fwd_desc& q = d[j];
      MML_REQUIRE(q.A && q.W && q.C, "mml_gemm_planes_fwd: null pointer in problem %d", j);
      MML_REQUIRE(q.M >= 1 && q.N >= 4 && q.K >= 1 && q.N % 4 == 0, "mml_gemm_planes_fwd: bad sizes in problem %d", j);
      const int k16 = (q.K + 15) / 16 * 16;
      MML_REQUIRE(q.ldpa >= k16 && q.ldpw >= k16 && q.ldpa % 16 == 0 && q.ldpw % 16 == 0 && aligned16(q.A) &&
                      aligned16(q.W),
                  "mml_gemm_planes_fwd: problem %d: planes need 16-byte aligned rows of ldp >= ceil16(K), ldp %% 16 == 0", j);
      MML_REQUIRE(aligned16(q.C) && q.ldc % 4 == 0 && q.ldc >= q.N && (!q.bias || aligned16(q.bias)),
                  "mml_gemm_planes_fwd: problem %d: C / bias must allow 16-byte accesses", j);
      MML_REQUIRE(!q.relu_mask || q.ldmask * 32 >= q.N, "mml_gemm_planes_fwd: ldmask too small in problem %d", j);
      QProblem& P = L.p[j - i];
      P.A = q.A; P.W = q.W; P.pa_step = 24 * (int64_t)q.M; P.pw_step = 24 * (int64_t)q.N;
      P.C = q.C; P.ldc = q.ldc; P.bias = q.bias;
      P.mask = (q.act == MML_ACT_RELU) ? q.relu_mask : nullptr; P.ldmask = q.ldmask;
      P.M = q.M; P.N = q.N; P.K16 = k16; P.act = q.act;
      P.tiles_n = (int)cdiv(q.N, QN);
      P.tile0 = t;
      t += P.tiles_n;
      ++j;
    }
    L.n = j - i;
    L.total_ntiles = t;
    L.tiles_m = (int)cdiv(d[i].M, QM);
    int64_t blocks = (int64_t)L.tiles_m * t;
    if (blocks > cus) blocks = cus;  // persistent: one workgroup per CU
    MML_LAUNCH(gemm_planes_kernel, dim3((unsigned)blocks), dim3(512), QLDS, to_stream(stream), L);
    int rc = check_launch("mml_gemm_planes_fwd");
    if (rc) return rc;
    i = j;
  }
  return MML_OK;
}

#ifdef QLAB_TIMES
extern "C" int mml_lab_planes_times(unsigned long long* out, int reset) {
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_qlab_t), sizeof(g_qlab_t)) != hipSuccess) return -1;
  if (reset) {
    unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_qlab_t), z, sizeof(z)) != hipSuccess) return -1;
  }
  return 0;
}
#endif
