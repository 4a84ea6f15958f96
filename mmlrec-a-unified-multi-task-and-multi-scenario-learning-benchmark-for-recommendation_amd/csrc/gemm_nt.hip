// Weight gradients of the fp32-equivalent GEMM family, cut ONCE per workgroup and transposed by the LDS
// (round 5; reference: autograd's mm backward of nn.Linear, model/utils.py:146-161 -- dW = dC^T A, db = column sums of dC).
//
// Both operands of a weight gradient are activations [batch, columns]: the reduction runs down the SLOW dimension of
// both, and both need their two fp16 planes cut (include/mmlrec.h, mml_gemm_set_mode: h = rne16(x s), l = rne16(x s - h),
// three v_mfma_f32_32x32x16_f16 per product block).  The tile kernel (gemm.hip, gemm_pipe_kernel<false, false, ...>) lets
// every wave cut the fragments it reads -- a dnn_input fragment is cut by 9 x 2 waves, a gradient fragment by 2 x 2 --
// and assembles k-contiguous fragments from a batch-major image with 4-byte LDS reads: 12 VALU instructions per MFMA,
// matrix pipe 36 % busy (profiles/r04_mfma.csv).  Here
//   * a workgroup (4 waves, 2 x 2) owns a 128 x 128 tile of dW for a slab of the batch and walks it in steps of 32 rows;
//   * the step's rows travel HBM -> registers as they lie in memory (16 bytes per lane, whole 512-byte row pieces),
//     every element is cut ONCE (2.5 VALU instructions) and its planes are written to LDS row-major as 8-byte pieces;
//   * the MFMA fragments -- eight consecutive BATCH rows of one column per lane -- come out of the transposing LDS read
//     of CDNA4, ds_read_b64_tr_b16 (4 rows x 16 columns per 16-lane group, delivered column-major): no strided reads, no
//     register transposition;
//   * the loads of step s + 1 are issued before the MFMAs of step s and cut + stored behind them (two LDS stages, one
//     barrier per step);
//   * the bias gradient is a fourth and fifth MFMA per block row against an all-ones column (k-tile 0 only);
//   * partial tiles go to the caller's workspace, scaled back exactly, and are summed in slab order (bitwise
//     reproducible) by nt_reduce_kernel.
// Serves launches whose problems carry both magnitudes, nn.Linear weight layout, M % 32 == 0, M >= 16 384, N % 32 == 0,
// K % 4 == 0 and 16-byte aligned rows; everything else stays on the tile kernel (mml_gemm_nt_try_wgrad returns
// MML_ERR_UNSUPPORTED).
#include "common.hpp"
#include "lds_async.hpp"

namespace mml {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) s16x4 lds_s16x4_t;

// Problems per launch.  Round 6: 3 x MML_MAX_GROUP -- PepNet's 40-odd weight gradients were three launch + reduction pairs
// (377 us of its 1.67 ms step); the launch descriptor is 3.9 KB of kernel arguments (limit 4 KB), read from device memory.
constexpr int NT_MAX_GROUP = 48;
constexpr int NT_STEP = 32;          // batch rows per step
constexpr int NT_PLANE = NT_STEP * 128;   // fp16 elements of one plane image ([32 rows][128 columns])
constexpr int NT_STAGE = 4 * NT_PLANE;    // dC h, dC l, A h, A l

struct NtProblem {
  const float* dC;
  const float* A;
  float* ws;       // [slabs][N][K]
  float* ws_bias;  // [slabs][N] or null
  const uint32_t* amax_dc;
  const uint32_t* amax_a;
  int64_t lddc, lda;
  int32_t N, K, ntiles, ktiles;
};
struct NtLaunch {
  NtProblem p[NT_MAX_GROUP];
  int32_t n_prob, steps, slabs, tiles;
};

__device__ __forceinline__ uint32_t nt_amax_load(const uint32_t* p) {
  uint32_t m = 0;
#pragma unroll
  for (int i = 0; i < MML_AMAX_WORDS; ++i) m = p[i] > m ? p[i] : m;
  return m;
}
// power-of-two exponent k with |x| 2^k < 2^15 for every |x| <= the slot's value (the rule of gemm.hip)
__device__ __forceinline__ int nt_scale_exp(uint32_t bits) {
  const int e = (int)((bits >> 23) & 0xffu);
  if (e == 255) return 0;
  const int k = 141 - e;
  return k > 110 ? 110 : (k < -110 ? -110 : k);
}
__device__ __forceinline__ float nt_pow2(int k) { return __uint_as_float((uint32_t)(127 + k) << 23); }

// image (b) of the CDNA programming guide, T10: 256-byte rows, chunk c of row r at position c ^ (((r & 3) << 2) | ((r >> 2) & 3))
__device__ __forceinline__ int nt_swz32(int row) { return ((row & 3) << 2) | ((row >> 2) & 3); }

// One fragment = two transposing reads (batch rows 8 h + q and 8 h + 4 + q of the 16-row sub-step).  `lo_off` / `hi_off` are
// the lane's element offsets of the two reads inside a plane image for sub-step 0 (computed once per kernel); the stage, the
// plane and the sub-step (16 rows = 2 048 elements, swizzle-neutral: the row's low four bits do not change) are compile-time
// offsets that fold into the instruction's immediate.
__device__ __forceinline__ f16x8 nt_frag32(const uint16_t* img, int lo_off, int hi_off) {
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(img + lo_off));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(img + hi_off));
  const s16x8 v = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
  return __builtin_bit_cast(f16x8, v);
}
__device__ __forceinline__ int nt_frag_off(int chunk0, int lane, int t) {
  const int g = lane >> 4, hh = g >> 1, q = (lane & 15) >> 2, p = lane & 3;
  const int ch = chunk0 + 2 * (g & 1) + (p >> 1);
  const int row = 8 * hh + 4 * t + q;
  return row * 128 + ((ch ^ nt_swz32(row)) << 3) + 4 * (p & 1);
}

// four values -> their two planes (8 bytes each): h = rne16(x s), l = rne16(x s - h); x s is exact (s a power of two).
// Ten VALU instructions: 2 v_pk_mul_f32 (round 6; four v_mul before), 2 + 2 v_cvt_pk_f16_f32, 4 v_fma_mix_f32 that read the
// fp16 halves of h directly.
__device__ __forceinline__ void nt_cut4(const float4& x, const float s, uint2& h, uint2& l) {
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  float r0, r1, r2, r3;
  uint32_t h0, h1, l0, l1;
  // (the two scalings of a register pair in one packed multiply: a loaded float4 IS two aligned pairs)
  const f32x2 x01 = {x.x, x.y}, x23 = {x.z, x.w}, ss = {s, s};
  f32x2 y01, y23;
  asm("v_pk_mul_f32 %0, %1, %2" : "=v"(y01) : "v"(x01), "s"(ss));
  asm("v_pk_mul_f32 %0, %1, %2" : "=v"(y23) : "v"(x23), "s"(ss));
  asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(h0) : "v"(y01.x), "v"(y01.y));
  asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(h1) : "v"(y23.x), "v"(y23.y));
  asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(r0) : "v"(x.x), "s"(s), "v"(h0));
  asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(r1) : "v"(x.y), "s"(s), "v"(h0));
  asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(r2) : "v"(x.z), "s"(s), "v"(h1));
  asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(r3) : "v"(x.w), "s"(s), "v"(h1));
  asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(l0) : "v"(r0), "v"(r1));
  asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(l1) : "v"(r2), "v"(r3));
  h = make_uint2(h0, h1);
  l = make_uint2(l0, l1);
}

__global__ __launch_bounds__(256, 2) void gemm_nt_kernel(const NtLaunch L) {
  __shared__ __attribute__((aligned(16))) uint16_t lds[2 * NT_STAGE];  // 64 KiB: two stages of four plane images
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);  // (wave-uniform: scalar branches)
  const int wn = w >> 1, wk = w & 1;
  const int id = xcd_remap(blockIdx.x, gridDim.x);
  const int slab = id / L.tiles;
  int t = id - slab * L.tiles;
  int pi = 0;
  while (t >= L.p[pi].ntiles * L.p[pi].ktiles) t -= L.p[pi].ntiles * L.p[pi].ktiles, ++pi;  // (uniform)
  const NtProblem& P = L.p[pi];
  const int nt = t / P.ktiles, kt = t - nt * P.ktiles;
  const int n0 = nt * 128, k0 = kt * 128;
  const int per = (L.steps + L.slabs - 1) / L.slabs;
  const int s_begin = slab * per, s_end = min(L.steps, s_begin + per);
  const bool want_bias = P.ws_bias != nullptr && kt == 0;  // (uniform)

  // scales (one pair per problem); 1 / (sC sA) must be one fp32 number: where the exponents add up beyond its range both
  // give way equally (gemm.hip: problem_scales)
  int kC = nt_scale_exp(nt_amax_load(P.amax_dc)), kA = nt_scale_exp(nt_amax_load(P.amax_a));
  {
    const int over = kC + kA - 126, under = -126 - (kC + kA);
    if (over > 0) { kC -= (over + 1) >> 1; kA -= over >> 1; }
    if (under > 0) { kC += (under + 1) >> 1; kA += under >> 1; }
  }
  kC = __builtin_amdgcn_readfirstlane(kC);
  kA = __builtin_amdgcn_readfirstlane(kA);
  const float sC = nt_pow2(kC), sA = nt_pow2(kA), inv = nt_pow2(-(kC + kA));

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  // bias gradient (k-tile 0): every thread sums the four dC columns it stages, in fp32; the eight row groups of a column
  // meet in LDS at the end (no accumulator registers, no extra MFMAs)
  float4 bsum = make_float4(0.f, 0.f, 0.f, 0.f);

  // staging map: 32 lanes cover one 128-column row (16 bytes each), 8 rows per pass, 4 passes per operand
  const int c4 = tid & 31, r8 = tid >> 5;
  // (columns beyond the problem: the lane re-reads the problem's first columns -- finite values whose products land in
  //  accumulator columns / rows the epilogue never stores)
  const bool colC = n0 + 4 * c4 < P.N, colA = k0 + 4 * c4 < P.K;
  // Addresses: a step's row base is workgroup-uniform (scalar registers, scalar arithmetic); what a thread adds -- its row of
  // the eight-row pass and its 16-byte column -- is ONE 32-bit byte offset per operand, computed here.  (Per-lane 64-bit row
  // arithmetic cost four VALU instructions per load: 64 of the 352 per trip of two steps, on an issue port the 48 MFMAs share.)
  const uint32_t tC = (uint32_t)(((int64_t)r8 * P.lddc + (colC ? n0 + 4 * c4 : 0)) * 4);
  const uint32_t tA = (uint32_t)(((int64_t)r8 * P.lda + (colA ? k0 + 4 * c4 : 0)) * 4);
  // TWO steps of rows in flight (two register sets, the trip below unrolled by two): one step's MFMAs are ~0.7 us, a load's
  // round trip 1-2 us -- with one step ahead the first-layer launch took 258 us, with two 198, with three 201 (and a
  // branch-free body with three sets spills: 508 bytes of scratch).  The body of a step
  // is ONE basic block (no branch: loads past the slab re-read its last step, sub-tiles outside the problem multiply zeros),
  // so that the compiler interleaves the next rows' cut and LDS stores with this step's MFMAs.
  float4 rc0[4], ra0[4], rc1[4], ra1[4];
  const int last = s_end - 1;
  auto load_step = [&](int st, float4 (&rc)[4], float4 (&ra)[4]) __attribute__((always_inline)) {
    st = st > last ? last : st;
    const char* const bC = reinterpret_cast<const char*>(P.dC + (int64_t)st * NT_STEP * P.lddc);
    const char* const bA = reinterpret_cast<const char*>(P.A + (int64_t)st * NT_STEP * P.lda);
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      rc[p] = *reinterpret_cast<const float4*>(bC + (int64_t)(8 * p) * P.lddc * 4 + tC);
      ra[p] = *reinterpret_cast<const float4*>(bA + (int64_t)(8 * p) * P.lda * 4 + tA);
    }
  };
  auto store_step = [&](uint16_t* stage, const float4 (&rc)[4], const float4 (&ra)[4], const float count) __attribute__((always_inline)) {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int row = r8 + 8 * p;
      const int off = row * 128 + (((c4 >> 1) ^ nt_swz32(row)) << 3) + 4 * (c4 & 1);
      uint2 h, l;
      nt_cut4(rc[p], sC, h, l);
      *reinterpret_cast<uint2*>(stage + off) = h;
      *reinterpret_cast<uint2*>(stage + NT_PLANE + off) = l;
      nt_cut4(ra[p], sA, h, l);
      *reinterpret_cast<uint2*>(stage + 2 * NT_PLANE + off) = h;
      *reinterpret_cast<uint2*>(stage + 3 * NT_PLANE + off) = l;
      // (count = 1 for a real step of a workgroup that owns the bias sums, else 0: an fma, no select)
      bsum.x = fmaf(rc[p].x, count, bsum.x); bsum.y = fmaf(rc[p].y, count, bsum.y);
      bsum.z = fmaf(rc[p].z, count, bsum.z); bsum.w = fmaf(rc[p].w, count, bsum.w);
    }
  };
  // per-lane offsets of the fragment reads (sub-step 0): [n / k sub-tile][low / high half of the fragment]
  int offC[2][2], offA[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int t2 = 0; t2 < 2; ++t2) {
      offC[i][t2] = nt_frag_off(wn * 8 + i * 4, lane, t2);
      offA[i][t2] = nt_frag_off(wk * 8 + i * 4, lane, t2);
    }
  auto compute = [&](const uint16_t* cur) __attribute__((always_inline)) {
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
      const uint16_t* const b0 = cur + sub * 16 * 128;
      f16x8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        ah[i] = nt_frag32(b0, offC[i][0], offC[i][1]);
        al[i] = nt_frag32(b0 + NT_PLANE, offC[i][0], offC[i][1]);
        bh[i] = nt_frag32(b0 + 2 * NT_PLANE, offA[i][0], offA[i][1]);
        bl[i] = nt_frag32(b0 + 3 * NT_PLANE, offA[i][0], offA[i][1]);
      }
      // product by product over the four accumulators: consecutive MFMAs are independent
#pragma unroll
      for (int pr = 0; pr < 3; ++pr)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(pr == 2 ? al[i] : ah[i], pr == 1 ? bl[j] : bh[j], acc[i][j], 0, 0, 0);
    }
  };

  // step q computes from LDS stage q & 1; its rows were cut into that stage one step earlier and loaded two steps earlier
  const int ns = s_end - s_begin;
  if (ns > 0) {
    load_step(s_begin, rc0, ra0);
    load_step(s_begin + 1, rc1, ra1);
    store_step(lds, rc0, ra0, want_bias ? 1.f : 0.f);
  }
  __syncthreads();
  // trip of two steps: set 0 was stored for step q already and takes the rows of step q + 2; set 1 (step q + 1) is cut into
  // the other stage behind the MFMAs of step q (past the end of the slab: a harmless re-cut of the last step into the stage
  // nobody reads again; the bias sums count real steps only)
#define NT_TRIP(Q, RCL, RAL, RCS, RAS)                                                        \
  {                                                                                           \
    const int q_ = (Q);                                                                       \
    load_step(s_begin + q_ + 2, RCL, RAL);                                                    \
    compute(lds + (q_ & 1) * NT_STAGE);                                                       \
    store_step(lds + ((q_ + 1) & 1) * NT_STAGE, RCS, RAS, (want_bias && q_ + 1 < ns) ? 1.f : 0.f); \
    __syncthreads();                                                                          \
  }
  // (no exit between the two halves of the pair: with one, the accumulators were copied -- 32 v_mov_b64 per pair -- to meet
  //  in the same registers on both exits; an odd step count ends with one trip of its own)
  int q = 0;
  for (; q + 1 < ns; q += 2) {
    NT_TRIP(q, rc0, ra0, rc1, ra1)
    NT_TRIP(q + 1, rc1, ra1, rc0, ra0)
  }
  if (q < ns) NT_TRIP(q, rc0, ra0, rc1, ra1)
#undef NT_TRIP
  // sub-tiles of this wave that lie inside the problem (whole 32-column blocks)
  bool vi[2], vj[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    vi[i] = n0 + wn * 64 + i * 32 < P.N;
    vj[i] = k0 + wk * 64 + i * 32 < P.K;
  }

  const int h = lane >> 5, c31 = lane & 31;
  float* const ws = P.ws + (int64_t)slab * P.N * P.K;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      if (!vi[i] || !vj[j]) continue;
      const int k = k0 + wk * 64 + j * 32 + c31;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = n0 + wn * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (k < P.K) ws[(int64_t)n * P.K + k] = acc[i][j][r] * inv;
      }
    }
  if (want_bias) {  // (uniform) the eight row groups of every column meet in LDS; the operand images are idle by now
    float* const red = reinterpret_cast<float*>(lds);
    *reinterpret_cast<float4*>(red + (r8 * 32 + c4) * 4) = bsum;
    __syncthreads();
    if (tid < 128 && n0 + tid < P.N) {
      float sum = 0.f;
#pragma unroll
      for (int g = 0; g < 8; ++g) sum += red[(g * 32 + (tid >> 2)) * 4 + (tid & 3)];
      P.ws_bias[(int64_t)slab * P.N + n0 + tid] = sum;
    }
  }
}

// ---- lab variant (round 6, MMLREC_NT_DIRECT=1): the dC fragments straight from global memory ----------------------------------
// gemm_nt_kernel's LDS carries, per wave and 32-row step, 32 transposing reads (2 cycles each) and 8 16-byte stores (13 cycles
// each: the store path, not the array, sets their price) against 24 MFMAs -- with eight waves per CU as many LDS cycles as MFMA
// cycles.  An MFMA A fragment of dC^T is eight consecutive BATCH rows of one column per lane, 32 lanes = 32 consecutive columns:
// a lane can fetch it itself with eight dword loads whose 32 lanes cover one 128-byte line.  This variant does that for dC (cut
// in registers, once per wave: the two k-halves of the workgroup cut it twice) and keeps the LDS path for the activation operand
// only: half the stores, half the fragment reads, 32 KiB of LDS per workgroup -- for 32 dword loads per lane and step on the
// vector-memory path instead of 4 dwordx4.  Same products, scales and k order as gemm_nt_kernel: its bits, except the bias
// gradient's summation order.
// MEASURED (tools/lab/nt_time.py, stand-alone, one box, profiles/r06_nt_direct.txt): AE-30 first layers 180-185 -> 213-220 us,
// KuaiRec-32 first layers 611-632 -> 754 us, second layers level: with half the LDS traffic the kernel is SLOWER -- the 36
// vector-memory instructions per lane and step (8 in gemm_nt_kernel) are the new bound, so the LDS was not the only one at its
// limit.  Kept as mode 3 of mml_gemm_set_nt (tests run it), not a default.
constexpr int NTD_STAGE = 2 * NT_PLANE;  // A h, A l

__global__ __launch_bounds__(256, 2) void gemm_ntd_kernel(const NtLaunch L) {
  __shared__ __attribute__((aligned(16))) uint16_t lds[2 * NTD_STAGE];  // 32 KiB
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = w >> 1, wk = w & 1;
  const int id = xcd_remap(blockIdx.x, gridDim.x);
  const int slab = id / L.tiles;
  int t = id - slab * L.tiles;
  int pi = 0;
  while (t >= L.p[pi].ntiles * L.p[pi].ktiles) t -= L.p[pi].ntiles * L.p[pi].ktiles, ++pi;  // (uniform)
  const NtProblem& P = L.p[pi];
  const int nt = t / P.ktiles, kt = t - nt * P.ktiles;
  const int n0 = nt * 128, k0 = kt * 128;
  const int per = (L.steps + L.slabs - 1) / L.slabs;
  const int s_begin = slab * per, s_end = min(L.steps, s_begin + per);
  const bool want_bias = P.ws_bias != nullptr && kt == 0;  // (uniform)

  int kC = nt_scale_exp(nt_amax_load(P.amax_dc)), kA = nt_scale_exp(nt_amax_load(P.amax_a));
  {
    const int over = kC + kA - 126, under = -126 - (kC + kA);
    if (over > 0) { kC -= (over + 1) >> 1; kA -= over >> 1; }
    if (under > 0) { kC += (under + 1) >> 1; kA += under >> 1; }
  }
  kC = __builtin_amdgcn_readfirstlane(kC);
  kA = __builtin_amdgcn_readfirstlane(kA);
  const float sC = nt_pow2(kC), sA = nt_pow2(kA), inv = nt_pow2(-(kC + kA));

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int c4 = tid & 31, r8 = tid >> 5;
  const bool colA = k0 + 4 * c4 < P.K;
  const uint32_t tA = (uint32_t)(((int64_t)r8 * P.lda + (colA ? k0 + 4 * c4 : 0)) * 4);
  const int hh = lane >> 5, c31 = lane & 31;
  bool vi[2], vj[2];
  uint32_t tD[2];  // the lane's byte offset into a step's rows: its half's first row, its column of block i
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    vi[i] = n0 + wn * 64 + i * 32 < P.N;
    vj[i] = k0 + wk * 64 + i * 32 < P.K;
    // (a block outside the problem re-reads the problem's first 32 columns: finite values for accumulators nobody stores)
    tD[i] = (uint32_t)(((int64_t)(8 * hh) * P.lddc + (vi[i] ? n0 + wn * 64 + i * 32 : 0) + c31) * 4);
  }
  float bs[2] = {0.f, 0.f};

  float dc0[2][2][8], dc1[2][2][8];  // [sub-step][block][batch row of the fragment]
  float4 ra0[4], ra1[4];
  const int last = s_end - 1;
  auto load_step = [&](int st, float (&dc)[2][2][8], float4 (&ra)[4]) __attribute__((always_inline)) {
    st = st > last ? last : st;
    const char* const bC = reinterpret_cast<const char*>(P.dC + (int64_t)st * NT_STEP * P.lddc);
    const char* const bA = reinterpret_cast<const char*>(P.A + (int64_t)st * NT_STEP * P.lda);
#pragma unroll
    for (int sub = 0; sub < 2; ++sub)
#pragma unroll
      for (int e = 0; e < 8; ++e)
#pragma unroll
        for (int i = 0; i < 2; ++i)
          dc[sub][i][e] = *reinterpret_cast<const float*>(bC + (int64_t)(sub * 16 + e) * P.lddc * 4 + tD[i]);
#pragma unroll
    for (int p = 0; p < 4; ++p) ra[p] = *reinterpret_cast<const float4*>(bA + (int64_t)(8 * p) * P.lda * 4 + tA);
  };
  auto store_step = [&](uint16_t* stage, const float4 (&ra)[4]) __attribute__((always_inline)) {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int row = r8 + 8 * p;
      const int off = row * 128 + (((c4 >> 1) ^ nt_swz32(row)) << 3) + 4 * (c4 & 1);
      uint2 h, l;
      nt_cut4(ra[p], sA, h, l);
      *reinterpret_cast<uint2*>(stage + off) = h;
      *reinterpret_cast<uint2*>(stage + NT_PLANE + off) = l;
    }
  };
  int offA[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int t2 = 0; t2 < 2; ++t2) offA[i][t2] = nt_frag_off(wk * 8 + i * 4, lane, t2);
  auto compute = [&](const uint16_t* cur, const float (&dc)[2][2][8]) __attribute__((always_inline)) {
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
      const uint16_t* const b0 = cur + sub * 16 * 128;
      f16x8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        bh[i] = nt_frag32(b0, offA[i][0], offA[i][1]);
        bl[i] = nt_frag32(b0 + NT_PLANE, offA[i][0], offA[i][1]);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const float4 x0 = make_float4(dc[sub][i][0], dc[sub][i][1], dc[sub][i][2], dc[sub][i][3]);
        const float4 x1 = make_float4(dc[sub][i][4], dc[sub][i][5], dc[sub][i][6], dc[sub][i][7]);
        uint2 h0, l0, h1, l1;
        nt_cut4(x0, sC, h0, l0);
        nt_cut4(x1, sC, h1, l1);
        const uint4 hv = make_uint4(h0.x, h0.y, h1.x, h1.y), lv = make_uint4(l0.x, l0.y, l1.x, l1.y);
        ah[i] = __builtin_bit_cast(f16x8, hv);
        al[i] = __builtin_bit_cast(f16x8, lv);
        bs[i] += ((x0.x + x0.y) + (x0.z + x0.w)) + ((x1.x + x1.y) + (x1.z + x1.w));
      }
#pragma unroll
      for (int pr = 0; pr < 3; ++pr)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(pr == 2 ? al[i] : ah[i], pr == 1 ? bl[j] : bh[j], acc[i][j], 0, 0, 0);
    }
  };

  const int ns = s_end - s_begin;
  if (ns > 0) {
    load_step(s_begin, dc0, ra0);
    load_step(s_begin + 1, dc1, ra1);
    store_step(lds, ra0);
  }
  __syncthreads();
  // step q: dC from the register set loaded two trips earlier (cut right in front of its MFMAs), the activation fragments from
  // LDS stage q & 1; the set is refilled for step q + 2 BEHIND the MFMAs that read it
#define NTD_TRIP(Q, DCQ, RAQ, RAS)                                                            \
  {                                                                                           \
    const int q_ = (Q);                                                                       \
    compute(lds + (q_ & 1) * NTD_STAGE, DCQ);                                                 \
    store_step(lds + ((q_ + 1) & 1) * NTD_STAGE, RAS);                                        \
    load_step(s_begin + q_ + 2, DCQ, RAQ);                                                    \
    __syncthreads();                                                                          \
  }
  int q = 0;
  for (; q + 1 < ns; q += 2) {
    NTD_TRIP(q, dc0, ra0, ra1)
    NTD_TRIP(q + 1, dc1, ra1, ra0)
  }
  if (q < ns) NTD_TRIP(q, dc0, ra0, ra1)
#undef NTD_TRIP

  float* const ws = P.ws + (int64_t)slab * P.N * P.K;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      if (!vi[i] || !vj[j]) continue;
      const int k = k0 + wk * 64 + j * 32 + c31;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = n0 + wn * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
        if (k < P.K) ws[(int64_t)n * P.K + k] = acc[i][j][r] * inv;
      }
    }
  if (want_bias) {  // (uniform) the two halves of a column's rows meet by one exchange; the k-half 0 waves own the sums
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const float sum = bs[i] + __shfl_xor(bs[i], 32, 64);
      if (wk == 0 && hh == 0 && vi[i]) P.ws_bias[(int64_t)slab * P.N + n0 + wn * 64 + i * 32 + c31] = sum;
    }
  }
}

struct NtRedProblem {
  const float* ws;
  const float* ws_bias;
  float* dW;
  float* dbias;
  int64_t lddw;
  int32_t N, K, accumulate;
};
struct NtRedLaunch {
  NtRedProblem p[NT_MAX_GROUP];
  int32_t n_prob, slabs;
};
// blockIdx.y = problem.  A workgroup sums 64 four-element pieces: its four waves take a quarter of the slabs each (eight
// loads in flight per lane), the quarters meet in LDS and are added in their fixed order -- bitwise reproducible, and four
// times the loads in flight of a thread-per-piece loop (second layers, 64 slabs: 42 -> see profiles/r05_nt_time.txt).
__global__ __launch_bounds__(256) void nt_reduce_kernel(const NtRedLaunch L) {
  __shared__ float4 part[4][64];
  const NtRedProblem& P = L.p[blockIdx.y];
  const int k4 = P.K >> 2;
  const int64_t total = (int64_t)P.N * k4;
  const int64_t plane = (int64_t)P.N * P.K;
  const int e = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int per = (L.slabs + 3) >> 2;
  const int s0 = g * per, s1 = min(L.slabs, s0 + per);
  for (int64_t base = (int64_t)blockIdx.x * 64; base < total; base += (int64_t)gridDim.x * 64) {
    const int64_t it = base + e;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    int64_t n = 0;
    int k = 0;
    if (it < total) {
      n = it / k4;
      k = (int)(it - n * k4) * 4;
      const float* const src = P.ws + n * P.K + k;
      int sl = s0;
      for (; sl + 8 <= s1; sl += 8) {
        float4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const float4*>(src + (sl + u) * plane);
#pragma unroll
        for (int u = 0; u < 8; ++u) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
      }
      for (; sl < s1; ++sl) {
        const float4 v = *reinterpret_cast<const float4*>(src + sl * plane);
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
      }
    }
    part[g][e] = s;
    __syncthreads();
    if (g == 0 && it < total) {
      float4 t = part[0][e];
#pragma unroll
      for (int q = 1; q < 4; ++q) { t.x += part[q][e].x; t.y += part[q][e].y; t.z += part[q][e].z; t.w += part[q][e].w; }
      float* const o = P.dW + n * P.lddw + k;
      if (P.accumulate) {
        t.x += o[0]; t.y += o[1]; t.z += o[2]; t.w += o[3];
      }
      o[0] = t.x; o[1] = t.y; o[2] = t.z; o[3] = t.w;
    }
    __syncthreads();
  }
  if (P.dbias && blockIdx.x == 0) {
    for (int n = threadIdx.x; n < P.N; n += blockDim.x) {
      // (eight slabs' loads in flight: a dependent load per slab made this loop -- 64 slabs, a few KB -- the longest part
      //  of the launch, ~30 us)
      float s = P.ws_bias[n];
      int sl = 1;
      for (; sl + 8 <= L.slabs; sl += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = P.ws_bias[(int64_t)(sl + u) * P.N + n];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += v[u];
      }
      for (; sl < L.slabs; ++sl) s += P.ws_bias[(int64_t)sl * P.N + n];
      P.dbias[n] = P.accumulate ? P.dbias[n] + s : s;
    }
  }
}

// Which launches: mml_gemm_set_nt / MMLREC_GEMM_NT = 0 none, 1 (DEFAULT) every launch that qualifies, 2 only those of at
// least NT_MIN_TILES output tiles.  Measured on one box, B = 65 536:
//   * stand-alone, cold caches, partial-product launch + reduction (tools/lab/nt_time.py, profiles/r05_nt_time.txt): AE-30
//     first layers (20 tiles) 196 + 15 us against the tile kernel's 212 + 16; KuaiRec-32 first layers (72 tiles) 665 + 20
//     against 681 + 24; second layers (8 tiles) 135 + 22 against 137 + 17; towers (2 half-empty tiles) 50 + 13 against 49 + 11;
//   * in the replayed AE-30 step, interleaved (tools/lab/ab_nt.sh, profiles/r05_ab_nt.txt): mode 0 1.673 / 1.695 ms, mode 2
//     1.644 / 1.602, mode 1 **1.598 / 1.609** (weight-gradient launches 0.287 -> 0.265 ms in the instrumented pass; the
//     step gains more than that); KuaiRec-32 fp32 3.12 -> 3.03 ms.
// How it got there: one step of rows in flight 258 us (first layers), two 198, three 201 (and spills in a branch-free body);
// then the step body as ONE basic block -- no branch around the MFMAs of sub-tiles outside the problem, no conditional loads
// -- so that the compiler interleaves the next rows' cut and LDS stores with the MFMAs: 184-196; the bias gradient from the
// staging threads' own sums instead of two more MFMAs and 32 accumulator registers; a reduction whose bias loop kept one
// dependent load in flight cost 30 us by itself.  Counters (tools/lab/nt_pmc.sh, profiles/r05_nt_pmc.txt, before the last two
// steps): 8.8 VALU instructions per MFMA against 11.9, MFMA pipe busy 0.32 against 0.34, no LDS bank conflicts either way,
// the same L2 requests and hit rate (70 %): neither kernel is bound by its cuts or by HBM; both wait on the short
// dependent chain read -> MFMA -> next read at two to three waves per SIMD.
constexpr int NT_MIN_TILES = 16;
static int g_nt_on = -1;
static int nt_mode() {
  if (g_nt_on < 0) {
    const char* e = getenv("MMLREC_GEMM_NT");
    g_nt_on = e ? atoi(e) : 1;
    if (g_nt_on < 0 || g_nt_on > 3) g_nt_on = 1;
  }
  return g_nt_on;
}
// Workgroups per CU.  Two fill the register file (2 x 4 waves x 248 VGPRs): nothing else becomes resident beside them, so a
// launch forked beside this one (the table scatter, the table optimizer: HBM-bound, idle matrix pipe) waits for it.  One
// workgroup per CU leaves half of every SIMD's registers and 76 KiB of LDS to the other branch.
static int g_nt_per_cu = 0;
static int nt_per_cu() {
  if (!g_nt_per_cu) {
    const char* e = getenv("MMLREC_NT_PER_CU");
    g_nt_per_cu = (e && atoi(e) == 1) ? 1 : 2;
  }
  return g_nt_per_cu;
}
static int nt_cus() {
  static int n = 0;
  if (!n) {
    int dev = 0;
    hipDeviceProp_t p;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess) n = p.multiProcessorCount;
    if (n <= 0) n = 256;
  }
  return n;
}

}  // namespace mml

using namespace mml;

// the per-problem conditions (mml_gemm_nt_serves: the grouping pass of the host side asks the same question)
static bool nt_serves(const mml_gemm_wgrad_desc& q) {
  if (q.M < 16384 || q.M % NT_STEP != 0) return false;
  if (q.w_kn || !q.dC || !q.A || !q.dW || !q.amax_dc || !q.amax_a) return false;
  if (q.N <= 0 || q.N % 32 != 0 || q.K <= 0 || q.K % 4 != 0) return false;
  if (!aligned16(q.dC) || !aligned16(q.A) || q.lddc % 4 != 0 || q.lda % 4 != 0 || q.lddc < q.N || q.lda < q.K ||
      q.lddw < q.K)
    return false;
  return true;
}

extern "C" int mml_gemm_nt_serves(const mml_gemm_wgrad_desc* desc) { return (desc && nt_serves(*desc)) ? 1 : 0; }

extern "C" int mml_gemm_set_nt(int32_t on) {
  MML_REQUIRE(on >= 0 && on <= 3, "mml_gemm_set_nt: 0 (off), 1 (every qualifying launch: default), 2 (launches of >= 16 tiles) "
              "or 3 (lab: every qualifying launch on gemm_ntd_kernel)");
  g_nt_on = on;
  return MML_OK;
}

// MML_OK: the launch (phase 1 = partial tiles, 2 = their reduction, 0 = both) was served; MML_ERR_UNSUPPORTED (no error
// text): not a launch this kernel serves -- the caller runs the tile kernel.  The decision depends on the descriptors and
// the workspace size only, so phase 2 of a call pair decides like its phase 1.
int mml_gemm_nt_try_wgrad(const mml_gemm_wgrad_desc* d, int32_t n, void* workspace, int64_t workspace_bytes, int32_t phase,
                          hipStream_t st) {
  if (nt_mode() == 0 || n < 1 || n > NT_MAX_GROUP || !workspace) return MML_ERR_UNSUPPORTED;
  const int32_t M = d[0].M;
  if (M < 16384 || M % NT_STEP != 0) return MML_ERR_UNSUPPORTED;
  int64_t tiles = 0, elems = 0;
  for (int i = 0; i < n; ++i) {
    const mml_gemm_wgrad_desc& q = d[i];
    if (q.M != M || !nt_serves(q)) return MML_ERR_UNSUPPORTED;
    tiles += cdiv(q.N, 128) * cdiv(q.K, 128);
    elems += (int64_t)q.N * q.K + q.N;
  }
  if (nt_mode() == 2 && tiles < NT_MIN_TILES) return MML_ERR_UNSUPPORTED;
  const int steps = M / NT_STEP;
  const int per_cu = nt_per_cu();
  int64_t slabs = (per_cu * (int64_t)nt_cus()) / tiles;   // two workgroups per CU (64 KiB of LDS each), or one (see nt_per_cu)
  if (slabs > steps / 8) slabs = steps / 8;           // at least 256 batch rows per slab
  if (slabs > 64) slabs = 64;
  const int64_t fit = workspace_bytes / (elems * 4);  // (the caller sized the workspace for the tile kernel's split)
  if (slabs > fit) slabs = fit;
  if (slabs < 1) return MML_ERR_UNSUPPORTED;

  NtLaunch L{};
  NtRedLaunch R{};
  L.n_prob = R.n_prob = n;
  L.steps = steps;
  L.slabs = R.slabs = (int)slabs;
  float* ws = reinterpret_cast<float*>(workspace);
  int maxred = 1;
  for (int i = 0; i < n; ++i) {
    const mml_gemm_wgrad_desc& q = d[i];
    NtProblem& P = L.p[i];
    P.dC = q.dC; P.A = q.A; P.lddc = q.lddc; P.lda = q.lda; P.N = q.N; P.K = q.K;
    P.amax_dc = q.amax_dc; P.amax_a = q.amax_a;
    P.ntiles = (int)cdiv(q.N, 128); P.ktiles = (int)cdiv(q.K, 128);
    P.ws = ws;
    ws += slabs * q.N * q.K;
    P.ws_bias = q.dbias ? ws : nullptr;
    ws += slabs * q.N;
    L.tiles += P.ntiles * P.ktiles;
    NtRedProblem& Q = R.p[i];
    Q.ws = P.ws; Q.ws_bias = P.ws_bias; Q.dW = q.dW; Q.dbias = q.dbias; Q.lddw = q.lddw; Q.N = q.N; Q.K = q.K;
    Q.accumulate = q.accumulate;
    const int blocks = (int)cdiv((int64_t)q.N * q.K / 4, 64);
    maxred = blocks > maxred ? blocks : maxred;
  }
  if (phase != 2) {
    // (mode 3: the lab variant gemm_ntd_kernel, dC fragments straight from global memory -- measured 19-21 % slower)
    // (one per CU: 20 KiB of unused dynamic LDS on top of the 64 KiB make a second workgroup not fit)
    if (nt_mode() == 3) MML_LAUNCH(gemm_ntd_kernel, dim3((unsigned)(L.tiles * slabs)), dim3(256), 0, st, L);
    else MML_LAUNCH(gemm_nt_kernel, dim3((unsigned)(L.tiles * slabs)), dim3(256), per_cu == 1 ? 20480 : 0, st, L);
    const int rc = check_launch("mml_gemm_grouped_wgrad(nt)");
    if (rc != MML_OK) return rc;
  }
  if (phase != 1) {
    if (maxred > 2048) maxred = 2048;
    MML_LAUNCH(nt_reduce_kernel, dim3((unsigned)maxred, (unsigned)n), dim3(256), 0, st, R);
    return check_launch("mml_gemm_grouped_wgrad(nt reduce)");
  }
  return MML_OK;
}
