// K3' -- the bf16-STORAGE GEMM family (include/mmlrec.h): the DNN layers of the reference (model/utils.py:146-161:
// addmm + relu_, and autograd's mm / mm backward pair) for BASELINE.json configs[1] (MMoE on KuaiRec-shaped batches, bf16),
// with activations and their gradients stored as bf16 wherever producer and consumers are GEMMs.
//
// Written for gfx950 only.  What differs from the fp32-equivalent kernels of gemm.hip / gemm_panel.hip / gemm_ws.hip:
//   * no operand is converted inside a GEMM: bf16 tiles travel HBM -> LDS with global_load_lds_dwordx4 (16 bytes per
//     lane, the XOR swizzle applied to the SOURCE address so that the linear LDS image is conflict-free for the fragment
//     reads) and LDS -> MFMA fragments as they are, one v_mfma_f32_32x32x16_bf16 per 16-k block;
//   * forward and input gradient are ONE kernel (g16_tn_kernel): both operands reduction-contiguous -- the weights are
//     cast to bf16 once per step in both orientations (cast16_kernel), so that the input gradient reads W^T rows;
//   * the weight gradient (g16_nt_kernel) reduces over the BATCH, the slow dimension of both of its operands: the tiles
//     are staged as they lie in memory ([batch row][column]) and the MFMA fragments are read with the transposing LDS
//     read of CDNA4, ds_read_b64_tr_b16 (4 batch rows x 16 columns per 16-lane group, delivered column-major) -- no
//     transposition in registers, no strided 2-byte LDS reads;
//   * 128 x 128 (x 64-k) tiles, 4 waves as 2 x 2, ~32 KiB of LDS and < 128 VGPRs per workgroup: three to four workgroups
//     per CU overlap each other's load, MFMA and epilogue phases (the layers' reductions are 4-36 k-steps long, far too
//     short for a deep software pipeline inside one workgroup to pay).  Measured at the end of round 5: the same kernel with
//     two LDS buffers (the DMA of k-step t + 1 issued before k-step t is worked on, fragments by asm ds_read_b128, 64 KiB
//     of LDS -> two workgroups per CU) ran KuaiRec-32's step at 1.720 ms against 1.610 (three interleaved pairs); not kept.
#include "common.hpp"
#include "lds_async.hpp"

#include <stdlib.h>

namespace mml {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) s16x4 lds_s16x4_t;

#ifndef G16_TN_WAVES
#define G16_TN_WAVES 4   // waves per SIMD the forward / input-gradient kernel is compiled for: 128 VGPRs, 20 bytes of
                         // scratch in the epilogue; same-box A/B on KuaiRec-32 at 65 536: 1.68 against 1.70 ms with 3 (140 VGPRs)
#endif
constexpr int G16_MAX_GROUP = 8;  // problems per launch (the launch struct travels in the 4 KiB kernel-argument block)

__device__ __forceinline__ uint32_t pack_bf16(float a, float b) {  // {bf16(a), bf16(b)}, round to nearest even
  typedef float f2 __attribute__((ext_vector_type(2)));
  const f2 v = {a, b};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}
__device__ __forceinline__ uint16_t to_bf16(float a) { return (uint16_t)(pack_bf16(a, 0.f) & 0xffffu); }

// ------------------------------------------------------------------------------------------------ casts
struct Cast16Launch {
  mml_cast16_desc d[MML_MAX_PLANES];
  int32_t n;
};
// blockIdx.y = matrix; grid-stride over 4-element pieces of the OUTPUT rows
__global__ __launch_bounds__(256) void cast16_kernel(const Cast16Launch L) {
  const mml_cast16_desc& d = L.d[blockIdx.y];
  if (d.transpose) {
    // 32 x 32 tiles through LDS: 128-byte row pieces are read, 64-byte row pieces of the transposed copy written (the
    // element-wise form below reads a COLUMN of the source per output row: 36 us for the 3.5 M weights of KuaiRec MMoE)
    __shared__ float tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int tr = (int)((d.rows + 31) >> 5), tc = (d.cols + 31) >> 5;
    for (int t = blockIdx.x; t < tr * tc; t += gridDim.x) {
      const int64_t r0 = (int64_t)(t / tc) * 32;
      const int c0 = (t % tc) * 32;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int64_t r = r0 + ty + 8 * k;
        tile[ty + 8 * k][tx] = (r < d.rows && c0 + tx < d.cols) ? d.src[r * d.lds + c0 + tx] : 0.f;
      }
      __syncthreads();
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int c = c0 + ty + 8 * k;          // row of the transposed copy
        const int64_t r = r0 + tx;              // its column
        if (c < d.cols && r < d.rows) d.dst[(int64_t)c * d.ldd + r] = to_bf16(tile[tx][ty + 8 * k]);
      }
      __syncthreads();
    }
    return;
  }
  const int64_t orows = d.transpose ? d.cols : d.rows;
  const int ocols = d.transpose ? (int)d.rows : d.cols;
  const int pieces = (ocols + 3) >> 2;
  const int64_t total = orows * pieces;
  for (int64_t it = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; it < total; it += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = it / pieces;
    const int c = (int)(it - r * pieces) * 4;
    float v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int cc = c + j;
      v[j] = cc < ocols ? (d.transpose ? d.src[(int64_t)cc * d.lds + r] : d.src[r * d.lds + cc]) : 0.f;
    }
    uint16_t* o = d.dst + r * d.ldd + c;
    if (c + 3 < ocols && ((d.ldd & 3) == 0) && ((reinterpret_cast<uintptr_t>(d.dst) & 7u) == 0)) {
      *reinterpret_cast<uint2*>(o) = make_uint2(pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3]));
    } else {
      for (int j = 0; j < 4 && c + j < ocols; ++j) o[j] = to_bf16(v[j]);
    }
  }
}

// ------------------------------------------------------------------------------------------------ gather -> bf16
struct Gather16Args {
  const float* tab[MML_MAX_FIELDS];
  int64_t vocab[MML_MAX_FIELDS];
  int32_t col[MML_MAX_FIELDS];
  const float* X;
  int64_t ldX, B, ldo;
  uint16_t* out;
  int32_t* status;
  int32_t F, E, dense_col0, Nd;
};
// One thread = four consecutive values of one sample's output row (E % 4 == 0), or one dense scalar -- the layout of
// gather_vec4_kernel (gather_scatter.hip), 8-byte stores.  Index semantics of X[:, c].long(): truncation, model/basemodel.py:476.
__global__ __launch_bounds__(256) void gather16_kernel(const Gather16Args a) {
  const int e4 = a.E >> 2;
  const int nvec = a.F * e4;
  const int per_sample = nvec + a.Nd;
  const int64_t total = a.B * per_sample;
  int bad = 0;
  for (int64_t it = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; it < total; it += (int64_t)gridDim.x * blockDim.x) {
    const int64_t b = it / per_sample;
    const int c = (int)(it - b * per_sample);
    if (c < nvec) {
      const int f = c / e4, part = c - f * e4;
      int64_t i = (int64_t)a.X[b * a.ldX + a.col[f]];
      if (i < 0) {
        bad |= 1;
        i = 0;
      } else if (i >= a.vocab[f]) {
        bad |= 2;
        i = a.vocab[f] - 1;
      }
      const float4 v = *reinterpret_cast<const float4*>(a.tab[f] + i * a.E + part * 4);
      *reinterpret_cast<uint2*>(a.out + b * a.ldo + (int64_t)c * 4) = make_uint2(pack_bf16(v.x, v.y), pack_bf16(v.z, v.w));
    } else {
      const int j = c - nvec;
      a.out[b * a.ldo + (int64_t)a.F * a.E + j] = to_bf16(a.X[b * a.ldX + a.dense_col0 + j]);
    }
  }
  if (bad && a.status) atomicOr(a.status, bad);
}

// ------------------------------------------------------------------------------------------------ forward / input gradient
struct G16TnProblem {
  const uint16_t* A[MML_MAX_SRC];
  const uint16_t* B[MML_MAX_SRC];
  int64_t lda[MML_MAX_SRC], ldb[MML_MAX_SRC];
  int32_t K[MML_MAX_SRC];
  const float* bias;
  void* C;
  int64_t ldc;
  uint32_t* mask_out;
  const uint32_t* mask_in;
  int64_t ldmask;
  int32_t N, n_src, act, c_bf16, accumulate, ntiles;
};
struct G16TnLaunch {
  G16TnProblem p[G16_MAX_GROUP];
  int32_t n_prob, M, tiles_per_mblock;
};

// LDS image of a [rows][64 k] bf16 tile: 128-byte rows, the eight 16-byte chunks of row r stored at chunk position
// c ^ ((r >> 1) & 7): the ds_read_b128 of a 32x32x16 fragment (lane = row, all lanes the same logical chunk) then takes
// 16 distinct 16-byte slots of the 256-byte bank row in each of its 16-lane groups ({0-3, 12-15, 20-27}, ...: MI355X
// microarchitecture guide, LDS) -- conflict-free.  The DMA writes LDS linearly, so the swizzle is applied to the SOURCE.
__device__ __forceinline__ int tn_swz(int row) { return (row >> 1) & 7; }

// BN = 256 (round 6): a wave's tile is 64 x 128 -- per 16-k block it reads 2 + 4 fragments for 8 MFMAs (0.75 reads per
// MFMA) where the 64 x 64 wave tile of BN = 128 reads 2 + 2 for 4 (1.0).  A CU's LDS delivers one 1 KiB fragment read in 8
// clocks to ONE of its four SIMDs, an MFMA occupies its SIMD for 32: at one read per MFMA the LDS pipe is exactly as busy as
// the matrix pipes and every hiccup of it stalls them (KuaiRec-32's 512-wide layers ran at 0.17-0.24 of the bf16 peak).  128
// accumulator registers -> two workgroups per CU (48 KiB of LDS each).
template <int BN>
__global__ __launch_bounds__(256, BN == 256 ? 2 : G16_TN_WAVES) void g16_tn_kernel(const G16TnLaunch L) {
  constexpr int NJ = BN / 64;  // 32-column subtiles per wave along N (waves 2 x 2: wave tile 64 x BN/2)
  __shared__ __attribute__((aligned(16))) uint16_t lds[(128 + BN) * 64];
  uint16_t* const sA = lds;
  uint16_t* const sB = lds + 128 * 64;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int wm = w >> 1, wn = w & 1;
  const int id = xcd_remap(blockIdx.x, gridDim.x);
  const int mb = id / L.tiles_per_mblock;
  int t = id - mb * L.tiles_per_mblock;
  int pi = 0;
  while (t >= L.p[pi].ntiles) t -= L.p[pi++].ntiles;  // (uniform)
  const G16TnProblem& P = L.p[pi];
  const int64_t m0 = (int64_t)mb * 128;
  const int n0 = t * BN;

  f32x16 acc[2][NJ];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // per-lane pieces of the staging addresses (row inside an 8-row DMA instruction, swizzled source chunk)
  const int lrow = lane >> 3, lpos = lane & 7;
  for (int s = 0; s < P.n_src; ++s) {
    const uint16_t* const A = P.A[s] + m0 * P.lda[s];
    const uint16_t* const B = P.B[s] + (int64_t)n0 * P.ldb[s];
    const int64_t lda = P.lda[s], ldb = P.ldb[s];
    // the lane's source pointers, once per source: a k-step moves them on by 64 elements (recomputed per k-step they were
    // five VALU instructions per MFMA: 64-bit row x pitch products for twelve DMA instructions against sixteen MFMAs)
    const uint16_t* ga[4];
    const uint16_t* gb[BN / 32];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = w * 32 + i * 8 + lrow;
      ga[i] = A + (int64_t)row * lda + ((lpos ^ tn_swz(row)) << 3);
    }
#pragma unroll
    for (int i = 0; i < BN / 32; ++i) {
      const int row = w * (BN / 4) + i * 8 + lrow;
      gb[i] = B + (int64_t)row * ldb + ((lpos ^ tn_swz(row)) << 3);
    }
    for (int k0 = 0; k0 < P.K[s]; k0 += 64) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {  // A tile: 128 rows, 4 DMA instructions of 8 rows per wave
        __builtin_amdgcn_global_load_lds(reinterpret_cast<const float*>(ga[i]),
                                         reinterpret_cast<float*>(sA + (w * 32 + i * 8) * 64), 16, 0, 0);
        ga[i] += 64;
      }
#pragma unroll
      for (int i = 0; i < BN / 32; ++i) {  // B tile: BN rows
        __builtin_amdgcn_global_load_lds(reinterpret_cast<const float*>(gb[i]),
                                         reinterpret_cast<float*>(sB + (w * (BN / 4) + i * 8) * 64), 16, 0, 0);
        gb[i] += 64;
      }
      __syncthreads();  // (waits for the DMA: vmcnt(0) + barrier)
#pragma unroll
      for (int sub = 0; sub < 4; ++sub) {
        const int ch = 2 * sub + (lane >> 5);
        bf16x8 a[2], b[NJ];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int row = wm * 64 + i * 32 + (lane & 31);
          a[i] = *reinterpret_cast<const bf16x8*>(sA + row * 64 + ((ch ^ tn_swz(row)) << 3));
        }
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          const int row = wn * (BN / 2) + j * 32 + (lane & 31);
          b[j] = *reinterpret_cast<const bf16x8*>(sB + row * 64 + ((ch ^ tn_swz(row)) << 3));
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
      }
      __syncthreads();  // (the tile is consumed: the next DMA may overwrite it)
    }
  }

  // Epilogue.  Element (reg r of subtile i, j) is row (r & 3) + 8 (r >> 2) + 4 (lane >> 5), column lane & 31 of its 32 x 32
  // subtile: stored from the registers a wave would write 64-byte (bf16) row pieces with 2-byte stores and fetch one mask
  // word per element.  Instead every wave turns its 64 x BN/2 tile row-major through its quarter of the (now idle)
  // operand LDS and stores 16 bytes per lane -- whole 128-byte lines; the ReLU sign masks are formed from / applied to the
  // eight (bf16) or four (fp32) consecutive values a lane then holds (a quad of lanes = one 32-column mask word).
  // A wave's tile goes through the turn in passes of at most 64 columns (NJP sub-tiles): BN = 256 makes two passes over
  // the same LDS region (its 12 KiB per wave hold 64 rows x 64 columns of bf16, or 32 x 64 of fp32, as at BN = 128).
  constexpr int NJP = NJ < 2 ? NJ : 2;            // sub-tiles per pass
  constexpr int WTN = NJP * 32;                   // columns of a pass
  constexpr int REGION = (128 + BN) * 64 / 4;     // bf16 elements of LDS per wave (12 KiB at BN = 256, 8 at 128, 6 at 64)
  uint16_t* const stage = lds + w * REGION;
  const int h = lane >> 5, c31 = lane & 31;
  const int64_t mrow0 = m0 + wm * 64;
  const bool relu = P.act == MML_ACT_RELU;
#pragma unroll
  for (int pass = 0; pass < NJ / NJP; ++pass) {
  if (pass) {  // (the previous pass's row-major reads are done before its region is overwritten)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
  const int ncol0 = n0 + wn * (BN / 2) + pass * WTN;
  float bias[NJP];
  f32x16 (&accp)[2][NJ] = acc;
#pragma unroll
  for (int j = 0; j < NJP; ++j) bias[j] = P.bias ? P.bias[ncol0 + j * 32 + c31] : 0.f;
#define ACC(i_, j_) accp[i_][pass * NJP + (j_)]

  if (P.c_bf16) {
    // write: pairs of columns packed; even lanes store the pair of reg r, odd lanes the pair of reg r + 1
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < NJP; ++j)
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
          float va = ACC(i, j)[r] + bias[j], vb = ACC(i, j)[r + 1] + bias[j];
          if (relu) {
            va = fmaxf(va, 0.f);
            vb = fmaxf(vb, 0.f);
          }
          const float na = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(va), 0xB1, 0xF, 0xF, true));  // lane ^ 1
          const float nb = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(vb), 0xB1, 0xF, 0xF, true));
          const bool odd = lane & 1;
          const uint32_t word = odd ? pack_bf16(nb, vb) : pack_bf16(va, na);
          const int rr = odd ? r + 1 : r;
          const int row = i * 32 + (rr & 3) + 8 * (rr >> 2) + 4 * h;
          const int col = j * 32 + (c31 & ~1);
          *reinterpret_cast<uint32_t*>(stage + row * WTN + col) = word;
        }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    constexpr int CPR = WTN / 8;         // 16-byte chunks per row
    constexpr int RPI = 64 / CPR;        // rows per iteration
    const int ch = lane % CPR, rsub = lane / CPR;
    uint16_t* const C = reinterpret_cast<uint16_t*>(P.C);
#pragma unroll
    for (int it = 0; it < 64 / RPI; ++it) {
      const int row = it * RPI + rsub;
      const int64_t m = mrow0 + row;
      uint4 q = *reinterpret_cast<const uint4*>(stage + row * WTN + ch * 8);
      uint32_t* qq = reinterpret_cast<uint32_t*>(&q);
      const int widx = (ncol0 >> 5) + (ch >> 2);
      if (P.mask_in) {   // (uniform) ReLU derivative: the forward's sign bits of these eight columns
        const uint32_t bits = P.mask_in[m * P.ldmask + widx] >> (8 * (ch & 3));
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (!((bits >> (2 * e)) & 1u)) qq[e] &= 0xffff0000u;
          if (!((bits >> (2 * e + 1)) & 1u)) qq[e] &= 0x0000ffffu;
        }
      }
      if (P.mask_out) {  // (uniform) sign bits of the stored values (ReLU outputs: positive = any magnitude bit)
        uint32_t bits = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          bits |= ((qq[e] & 0x00007fffu) ? 1u : 0u) << (2 * e);
          bits |= ((qq[e] & 0x7fff0000u) ? 1u : 0u) << (2 * e + 1);
        }
        uint32_t wd = bits << (8 * (ch & 3));
        wd |= (uint32_t)__builtin_amdgcn_mov_dpp((int)wd, 0xB1, 0xF, 0xF, true);  // lane ^ 1
        wd |= (uint32_t)__builtin_amdgcn_mov_dpp((int)wd, 0x4E, 0xF, 0xF, true);  // lane ^ 2
        if ((ch & 3) == 0) P.mask_out[m * P.ldmask + widx] = wd;
      }
      *reinterpret_cast<uint4*>(C + m * P.ldc + ncol0 + ch * 8) = q;
    }
  } else {
    float* const st32 = reinterpret_cast<float*>(stage);
    float* const C = reinterpret_cast<float*>(P.C);
    constexpr int CPR = WTN / 4;         // 16-byte chunks (four floats) per row
    constexpr int RPI = 64 / CPR;        // rows per iteration
    const int ch = lane % CPR, rsub = lane / CPR;
#pragma unroll
    for (int i = 0; i < 2; ++i) {        // two passes of 32 rows: a pass fills the wave's LDS region
      if (i) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
      }
#pragma unroll
      for (int j = 0; j < NJP; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float v = ACC(i, j)[r] + bias[j];
          if (relu) v = fmaxf(v, 0.f);
          st32[((r & 3) + 8 * (r >> 2) + 4 * h) * WTN + j * 32 + c31] = v;
        }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int it = 0; it < 32 / RPI; ++it) {
        const int row = it * RPI + rsub;
        const int64_t m = mrow0 + i * 32 + row;
        float4 q = *reinterpret_cast<const float4*>(st32 + row * WTN + ch * 4);
        const int widx = (ncol0 >> 5) + (ch >> 3);
        if (P.mask_in) {
          const uint32_t bits = P.mask_in[m * P.ldmask + widx] >> (4 * (ch & 7));
          if (!(bits & 1u)) q.x = 0.f;
          if (!(bits & 2u)) q.y = 0.f;
          if (!(bits & 4u)) q.z = 0.f;
          if (!(bits & 8u)) q.w = 0.f;
        }
        if (P.mask_out) {
          uint32_t wd = ((q.x > 0.f ? 1u : 0u) | (q.y > 0.f ? 2u : 0u) | (q.z > 0.f ? 4u : 0u) | (q.w > 0.f ? 8u : 0u))
                        << (4 * (ch & 7));
          wd |= (uint32_t)__builtin_amdgcn_mov_dpp((int)wd, 0xB1, 0xF, 0xF, true);  // lane ^ 1
          wd |= (uint32_t)__builtin_amdgcn_mov_dpp((int)wd, 0x4E, 0xF, 0xF, true);  // lane ^ 2
          wd |= (uint32_t)__shfl_xor((int)wd, 4, 64);
          if ((ch & 7) == 0) P.mask_out[m * P.ldmask + widx] = wd;
        }
        float* const o = C + m * P.ldc + ncol0 + ch * 4;
        if (P.accumulate) {
          const float4 old = *reinterpret_cast<const float4*>(o);
          q.x += old.x; q.y += old.y; q.z += old.z; q.w += old.w;
        }
        *reinterpret_cast<float4*>(o) = q;
      }
    }
  }
  }  // pass
#undef ACC
}

// ------------------------------------------------------------------------------------------------ weight gradient
struct G16NtProblem {
  const uint16_t* dC;
  const uint16_t* A;
  float* ws;       // [slabs][N][K] partial tiles
  float* ws_bias;  // [slabs][N] partial column sums of dC, or null
  int64_t lddc, lda;
  int32_t N, K, ntiles, ktiles;
};
struct G16NtLaunch {
  G16NtProblem p[G16_MAX_GROUP];
  int32_t n_prob, steps, slabs, tiles;  // steps: 64-row steps of the batch; tiles: output tiles of all problems
};

// LDS image of a [64 batch rows][128 columns] bf16 tile for ds_read_b64_tr_b16: 256-byte rows, the sixteen 16-byte chunks
// of row r stored at chunk position c ^ (((r & 3) << 2) | ((r >> 2) & 3)) -- image (b) of the CDNA programming guide, T10:
// the transposed reads of a 32x32x16 operand are conflict-free.
__device__ __forceinline__ int nt_swz(int row) { return ((row & 3) << 2) | ((row >> 2) & 3); }

__device__ __forceinline__ bf16x8 nt_fragment(const uint16_t* tile, int ms, int chunk0, int lane) {
  // rows ms + 8 h + 4 t + q (t = 0, 1), columns 8 chunk0 + 16 (g & 1) + (0 .. 15), for the lane's 16-lane group g
  const int g = lane >> 4, hh = g >> 1, q = (lane & 15) >> 2, p = lane & 3;
  const int ch = chunk0 + 2 * (g & 1) + (p >> 1);
  s16x4 lo, hi;
  {
    const int row = ms + 8 * hh + q;
    const uint16_t* a = tile + row * 128 + ((ch ^ nt_swz(row)) << 3) + 4 * (p & 1);
    lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)a);
  }
  {
    const int row = ms + 8 * hh + 4 + q;
    const uint16_t* a = tile + row * 128 + ((ch ^ nt_swz(row)) << 3) + 4 * (p & 1);
    hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)a);
  }
  const s16x8 v = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
  return __builtin_bit_cast(bf16x8, v);
}

// KW = 256 (round 6): the workgroup's tile is 128 (N) x 256 (K), a wave's 64 x 128 -- per 16-row sub-step it reads 2 + 4
// fragments for 8 MFMAs (0.75 fragment reads per MFMA) where the 64 x 64 wave tile of KW = 128 reads 2 + 2 for 4 (1.0: the
// LDS pipe exactly as busy as the matrix pipes, see g16_tn_kernel<256>).  The weight gradient's reduction is the batch --
// hundreds of steps per workgroup -- so the lower occupancy (128 accumulator registers: two workgroups per CU) costs no
// prologue / epilogue overlap here.  The A tile is two 64 x 128 images (one per k-half of the tile, the layout nt_fragment
// reads); wave (wn, wk) takes image wk.
template <int KW>
__global__ __launch_bounds__(256, KW == 256 ? 2 : 3) void g16_nt_kernel(const G16NtLaunch L) {
  constexpr int NIMG = KW / 128;   // A images
  constexpr int NJ = KW / 64;      // 32-column sub-tiles of a wave along K
  __shared__ __attribute__((aligned(16))) uint16_t lds[(1 + NIMG) * 64 * 128];
  uint16_t* const sC = lds;
  uint16_t* const sA = lds + 64 * 128;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int wn = w >> 1, wk = w & 1;
  // (slab-major: the workgroups of one slab read the same batch rows at about the same time)
  const int id = xcd_remap(blockIdx.x, gridDim.x);
  const int slab = id / L.tiles;
  int t = id - slab * L.tiles;
  int pi = 0;
  while (t >= L.p[pi].ntiles * L.p[pi].ktiles) t -= L.p[pi].ntiles * L.p[pi].ktiles, ++pi;  // (uniform)
  const G16NtProblem& P = L.p[pi];
  const int nt = t / P.ktiles, kt = t - nt * P.ktiles;
  const int n0 = nt * 128, k0 = kt * KW;
  const int per = (L.steps + L.slabs - 1) / L.slabs;
  const int s_begin = slab * per, s_end = min(L.steps, s_begin + per);
  const bool want_bias = P.ws_bias != nullptr && kt == 0;  // (uniform)

  f32x16 acc[2][NJ], accb[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
#pragma unroll
    for (int r = 0; r < 16; ++r) accb[i][r] = 0.f;
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  }
  // the all-ones column: B[k][col] = (col == 0) -> D[n][0] = sum_k A[n][k], the column sums of dC (bias gradient)
  s16x8 ones_v;
#pragma unroll
  for (int e = 0; e < 8; ++e) ones_v[e] = (lane & 31) == 0 ? (short)0x3F80 : (short)0;
  const bf16x8 ones = __builtin_bit_cast(bf16x8, ones_v);
  // this wave's A image and first chunk inside it
  const uint16_t* const myA = sA + (NIMG == 2 ? wk : 0) * 64 * 128;
  const int chunkA0 = NIMG == 2 ? 0 : wk * 8;

  const int lrow = lane >> 4, lpos = lane & 15;
  for (int st = s_begin; st < s_end; ++st) {
    const int64_t m0 = (int64_t)st * 64;
#pragma unroll
    for (int i = 0; i < 4; ++i) {  // each image: 64 rows of 256 bytes, 4 DMA instructions of 4 rows per wave
      const int row = (w * 4 + i) * 4 + lrow;
      const int ch = lpos ^ nt_swz(row);
      __builtin_amdgcn_global_load_lds(reinterpret_cast<const float*>(P.dC + (m0 + row) * P.lddc + n0 + (ch << 3)),
                                       reinterpret_cast<float*>(sC + (w * 4 + i) * 4 * 128), 16, 0, 0);
#pragma unroll
      for (int im = 0; im < NIMG; ++im)
        __builtin_amdgcn_global_load_lds(reinterpret_cast<const float*>(P.A + (m0 + row) * P.lda + k0 + im * 128 + (ch << 3)),
                                         reinterpret_cast<float*>(sA + im * 64 * 128 + (w * 4 + i) * 4 * 128), 16, 0, 0);
    }
    __syncthreads();
#pragma unroll
    for (int sub = 0; sub < 4; ++sub) {
      bf16x8 a[2], b[NJ];
#pragma unroll
      for (int i = 0; i < 2; ++i) a[i] = nt_fragment(sC, sub * 16, wn * 8 + i * 4, lane);
#pragma unroll
      for (int j = 0; j < NJ; ++j) b[j] = nt_fragment(myA, sub * 16, chunkA0 + j * 4, lane);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
      if (want_bias && wk == 0) {
#pragma unroll
        for (int i = 0; i < 2; ++i) accb[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], ones, accb[i], 0, 0, 0);
      }
    }
    __syncthreads();
  }

  const int h = lane >> 5, c31 = lane & 31;
  float* const ws = P.ws + (int64_t)slab * P.N * P.K;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = n0 + wn * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        const int k = k0 + wk * (KW / 2) + j * 32 + c31;
        ws[(int64_t)n * P.K + k] = acc[i][j][r];
      }
  if (want_bias && wk == 0 && c31 == 0) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = n0 + wn * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        P.ws_bias[(int64_t)slab * P.N + n] = accb[i][r];
      }
  }
}

struct G16RedProblem {
  const float* ws;
  const float* ws_bias;
  float* dW;
  float* dbias;
  int64_t lddw;
  int32_t N, K, accumulate;
};
struct G16RedLaunch {
  G16RedProblem p[G16_MAX_GROUP];
  int32_t n_prob, slabs;
};
// blockIdx.y = problem; the slabs are added in their fixed order (bitwise reproducible)
__global__ __launch_bounds__(256) void g16_reduce_kernel(const G16RedLaunch L) {
  const G16RedProblem& P = L.p[blockIdx.y];
  const int k4 = P.K >> 2;
  const int64_t total = (int64_t)P.N * k4;
  const int64_t plane = (int64_t)P.N * P.K;
  for (int64_t it = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; it < total; it += (int64_t)gridDim.x * blockDim.x) {
    const int64_t n = it / k4;
    const int k = (int)(it - n * k4) * 4;
    // (eight slabs' loads in flight per thread -- one load per iteration is latency-bound; additions in slab order)
    const float* const src = P.ws + n * P.K + k;
    float4 s = *reinterpret_cast<const float4*>(src);
    int sl = 1;
    for (; sl + 8 <= L.slabs; sl += 8) {
      float4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const float4*>(src + (sl + u) * plane);
#pragma unroll
      for (int u = 0; u < 8; ++u) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
    }
    for (; sl < L.slabs; ++sl) {
      const float4 v = *reinterpret_cast<const float4*>(src + sl * plane);
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    float* const o = P.dW + n * P.lddw + k;
    if (P.accumulate) {
      s.x += o[0]; s.y += o[1]; s.z += o[2]; s.w += o[3];
    }
    o[0] = s.x; o[1] = s.y; o[2] = s.z; o[3] = s.w;
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

static thread_local const char* g16_last = "";
static int g16_cus() {
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

extern "C" const char* mml_g16_last_kernel(void) { return g16_last; }

extern "C" int mml_cast16_batch(const mml_cast16_desc* d, int32_t n, mml_stream_t stream) {
  MML_REQUIRE(d != nullptr || n == 0, "mml_cast16_batch: descriptor array is null");
  MML_REQUIRE(n >= 0, "mml_cast16_batch: n < 0");
  for (int base = 0; base < n; base += MML_MAX_PLANES) {
    Cast16Launch L{};
    L.n = n - base < MML_MAX_PLANES ? n - base : MML_MAX_PLANES;
    int64_t most = 1;
    for (int i = 0; i < L.n; ++i) {
      const mml_cast16_desc& q = d[base + i];
      MML_REQUIRE(q.src && q.dst && q.rows >= 0 && q.cols >= 0, "mml_cast16_batch: null matrix or negative extent (item %d)", base + i);
      MML_REQUIRE(q.lds >= q.cols && q.ldd >= (q.transpose ? q.rows : q.cols), "mml_cast16_batch: row pitch below the row length (item %d)", base + i);
      L.d[i] = q;
      const int64_t pieces = (q.transpose ? (int64_t)q.cols * ((q.rows + 3) / 4) : q.rows * ((q.cols + 3) / 4));
      most = pieces > most ? pieces : most;
    }
    int64_t bx = cdiv(most, 256);
    if (bx > 1024) bx = 1024;
    MML_LAUNCH(cast16_kernel, dim3((unsigned)bx, (unsigned)L.n), dim3(256), 0, to_stream(stream), L);
    const int rc = check_launch("mml_cast16_batch");
    if (rc != MML_OK) return rc;
  }
  g16_last = "cast16_kernel";
  return MML_OK;
}

extern "C" int mml_gather16_fwd(const float* const* tables, const int64_t* vocab, const int32_t* col, int32_t F, int32_t E,
                                const float* X, int64_t ldX, int32_t dense_col0, int32_t Nd, int64_t B, uint16_t* out,
                                int64_t ldo, int32_t* status, mml_stream_t stream) {
  MML_REQUIRE(tables && vocab && col && X && out, "mml_gather16_fwd: null argument");
  MML_REQUIRE(F >= 1 && F <= MML_MAX_FIELDS, "mml_gather16_fwd: F out of range");
  MML_REQUIRE(E >= 4 && E % 4 == 0, "mml_gather16_fwd: embedding width must be a multiple of 4");
  MML_REQUIRE(Nd >= 0 && B >= 0 && ldo >= (int64_t)F * E + Nd && ldo % 4 == 0, "mml_gather16_fwd: bad output pitch");
  MML_REQUIRE((reinterpret_cast<uintptr_t>(out) & 7u) == 0, "mml_gather16_fwd: output must be 8-byte aligned");
  if (B == 0) return MML_OK;
  Gather16Args a{};
  for (int f = 0; f < F; ++f) {
    MML_REQUIRE(tables[f] && aligned16(tables[f]) && vocab[f] >= 1, "mml_gather16_fwd: table %d null / misaligned / empty", f);
    a.tab[f] = tables[f];
    a.vocab[f] = vocab[f];
    a.col[f] = col[f];
  }
  a.X = X; a.ldX = ldX; a.B = B; a.ldo = ldo; a.out = out; a.status = status;
  a.F = F; a.E = E; a.dense_col0 = dense_col0; a.Nd = Nd;
  const int64_t total = B * ((int64_t)F * (E / 4) + Nd);
  int64_t blocks = cdiv(total, 256);
  if (blocks > 0x7fffffff) blocks = 0x7fffffff;
  MML_LAUNCH(gather16_kernel, dim3((unsigned)blocks), dim3(256), 0, to_stream(stream), a);
  g16_last = "gather16_kernel";
  return check_launch("mml_gather16_fwd");
}

// one kernel launch over problems that share a tile width
static int g16_tn_launch(const mml_g16_tn_desc* d, const int* idx, int n, int bn, hipStream_t st) {
  G16TnLaunch L{};
  L.n_prob = n;
  L.M = d[idx[0]].M;
  for (int k = 0; k < n; ++k) {
    const int i = idx[k];
    const mml_g16_tn_desc& q = d[i];
    G16TnProblem& P = L.p[k];
    MML_REQUIRE(q.M == L.M, "mml_g16_tn: the problems of a launch share M");
    MML_REQUIRE(q.N > 0 && q.N % 64 == 0, "mml_g16_tn: N must be a positive multiple of 64 (problem %d)", i);
    MML_REQUIRE(q.n_src >= 1 && q.n_src <= MML_MAX_SRC, "mml_g16_tn: n_src out of range (problem %d)", i);
    MML_REQUIRE(q.act == MML_ACT_NONE || q.act == MML_ACT_RELU, "mml_g16_tn: activation must be none or relu");
    MML_REQUIRE(q.C != nullptr && q.ldc >= q.N, "mml_g16_tn: null output or pitch below N (problem %d)", i);
    MML_REQUIRE(aligned16(q.C) && q.ldc % (q.c_bf16 ? 8 : 4) == 0, "mml_g16_tn: output rows must be 16-byte aligned (problem %d)", i);
    MML_REQUIRE(!(q.c_bf16 && q.accumulate), "mml_g16_tn: accumulation needs an fp32 output");
    MML_REQUIRE(!((q.mask_out || q.mask_in) && q.ldmask * 32 < q.N), "mml_g16_tn: mask pitch below ceil(N / 32) words");
    MML_REQUIRE(!(q.mask_out && q.act != MML_ACT_RELU), "mml_g16_tn: sign masks belong to a ReLU output");
    for (int s = 0; s < q.n_src; ++s) {
      MML_REQUIRE(q.A[s] && q.B[s] && q.K[s] > 0 && q.K[s] % 64 == 0, "mml_g16_tn: source %d of problem %d: null operand or K %% 64 != 0", s, i);
      MML_REQUIRE(aligned16(q.A[s]) && aligned16(q.B[s]) && q.lda[s] % 8 == 0 && q.ldb[s] % 8 == 0 && q.lda[s] >= q.K[s] &&
                      q.ldb[s] >= q.K[s],
                  "mml_g16_tn: operands must be 16-byte aligned with row pitches that are multiples of 8 (problem %d)", i);
      P.A[s] = q.A[s]; P.B[s] = q.B[s]; P.lda[s] = q.lda[s]; P.ldb[s] = q.ldb[s]; P.K[s] = q.K[s];
    }
    P.bias = q.bias; P.C = q.C; P.ldc = q.ldc; P.mask_out = q.mask_out; P.mask_in = q.mask_in; P.ldmask = q.ldmask;
    P.N = q.N; P.n_src = q.n_src; P.act = q.act; P.c_bf16 = q.c_bf16; P.accumulate = q.accumulate;
    P.ntiles = q.N / bn;
    L.tiles_per_mblock += P.ntiles;
  }
  const int64_t grid = (int64_t)(L.M / 128) * L.tiles_per_mblock;
  MML_REQUIRE(grid <= 0x7fffffff, "mml_g16_tn: too many tiles");
  if (bn == 256) {
    MML_LAUNCH(g16_tn_kernel<256>, dim3((unsigned)grid), dim3(256), 0, st, L);
    g16_last = "g16_tn_kernel<256>";
  } else if (bn == 128) {
    MML_LAUNCH(g16_tn_kernel<128>, dim3((unsigned)grid), dim3(256), 0, st, L);
    g16_last = "g16_tn_kernel<128>";
  } else {
    MML_LAUNCH(g16_tn_kernel<64>, dim3((unsigned)grid), dim3(256), 0, st, L);
    g16_last = "g16_tn_kernel<64>";
  }
  return check_launch("mml_g16_tn");
}

// Widest tile the launches of mml_g16_tn may use (process-wide; MMLREC_G16_BN = 64 / 128 / 256, default 256): a lab knob.
static int g16_bn_max() {
  static int v = 0;
  if (v == 0) {
    const char* e = getenv("MMLREC_G16_BN");
    v = (e && atoi(e) > 0) ? atoi(e) : 256;
  }
  return v;
}

extern "C" int mml_g16_tn(const mml_g16_tn_desc* d, int32_t n, mml_stream_t stream) {
  MML_REQUIRE(d != nullptr, "mml_g16_tn: descriptor array is null");
  MML_REQUIRE(n >= 1 && n <= G16_MAX_GROUP, "mml_g16_tn: 1 .. %d problems per launch", G16_MAX_GROUP);
  MML_REQUIRE(d[0].M > 0 && d[0].M % 128 == 0, "mml_g16_tn: M must be a positive multiple of 128");
  // The problems whose width is a multiple of 256 (at batches that still fill the chip with 128 x 256 tiles) run 128 x 256
  // tiles in a launch of their own; the others share 128 x 128 tiles, or 128 x 64 as soon as one width is not a multiple of
  // 128 (the rule of round 5).  Problems of one call are independent: two launches in a row.
  // Measured (KuaiRec-32 at B = 65 536, three interleaved pairs, tools/lab/ab_g16_bn.sh): the wide tiles win where the
  // reduction is LONG -- the first layers' input gradient, one problem of 2 304 reduction columns: 184 -> 167 us -- and
  // lose where it is short (4 x 512 <- 256: 160 -> 188 us; 4 x 512 -> 256 over 512: 133 -> 140; the first-layer group,
  // which they split into two launches: 205 -> 228): two workgroups per CU instead of four cover each other's load /
  // MFMA / epilogue phases less well than the fewer LDS reads return on 4-8 k-steps.  So: from 1 536 reduction columns on.
  int wide[G16_MAX_GROUP], rest[G16_MAX_GROUP], nw = 0, nr = 0;
  for (int i = 0; i < n; ++i) {
    int64_t ksum = 0;
    for (int s = 0; s < d[i].n_src && s < MML_MAX_SRC; ++s) ksum += d[i].K[s];
    const int64_t kmin = g16_bn_max() > 256 ? 0 : 1536;  // (MMLREC_G16_BN=512: wide tiles for every qualifying problem -- lab)
    if (g16_bn_max() >= 256 && d[i].N > 0 && d[i].N % 256 == 0 && d[i].M >= 8192 && ksum >= kmin) wide[nw++] = i;
    else rest[nr++] = i;
  }
  if (nw) {
    const int rc = g16_tn_launch(d, wide, nw, 256, to_stream(stream));
    if (rc != MML_OK) return rc;
  }
  if (nr) {
    bool all128 = g16_bn_max() >= 128;
    for (int k = 0; k < nr; ++k) all128 = all128 && d[rest[k]].N % 128 == 0;
    return g16_tn_launch(d, rest, nr, all128 ? 128 : 64, to_stream(stream));
  }
  return MML_OK;
}

// 128 x 256 tiles (g16_nt_kernel<256>) for a launch whose problems all have K % 256 == 0 (MMLREC_G16_NT_KW=128 keeps the
// narrow tiles: lab knob, read once)
static bool g16_nt_wide(const mml_g16_wgrad_desc* d, int32_t n) {
  static int kw = 0;
  if (kw == 0) {
    const char* e = getenv("MMLREC_G16_NT_KW");
    kw = (e && atoi(e) > 0) ? atoi(e) : 256;
  }
  if (kw < 256 || d[0].M < 8192) return false;
  for (int i = 0; i < n; ++i)
    if (d[i].K % 256 != 0) return false;
  return true;
}

static int g16_wgrad_plan(const mml_g16_wgrad_desc* d, int32_t n, int* slabs_out, int64_t* bytes_out) {
  MML_REQUIRE(d != nullptr, "mml_g16_wgrad: descriptor array is null");
  MML_REQUIRE(n >= 1 && n <= G16_MAX_GROUP, "mml_g16_wgrad: 1 .. %d problems per launch", G16_MAX_GROUP);
  int64_t tiles = 0, elems = 0;
  for (int i = 0; i < n; ++i) {
    const mml_g16_wgrad_desc& q = d[i];
    MML_REQUIRE(q.M == d[0].M && q.M > 0 && q.M % 64 == 0, "mml_g16_wgrad: the problems share M, a positive multiple of 64");
    MML_REQUIRE(q.N > 0 && q.N % 128 == 0 && q.K > 0 && q.K % 128 == 0, "mml_g16_wgrad: N and K must be multiples of 128 (problem %d)", i);
    MML_REQUIRE(q.dC && q.A && q.dW, "mml_g16_wgrad: null operand (problem %d)", i);
    // (dW / dbias may sit anywhere in a flat gradient arena: the reduction stores them element by element)
    MML_REQUIRE(aligned16(q.dC) && aligned16(q.A) && q.lddc % 8 == 0 && q.lda % 8 == 0 && q.lddc >= q.N && q.lda >= q.K &&
                    q.lddw >= q.K,
                "mml_g16_wgrad: dC and A must be 16-byte aligned with pitches that are multiples of 8 (problem %d)", i);
    tiles += (int64_t)(q.N / 128) * (q.K / 128);
    elems += (int64_t)q.N * q.K + q.N;
  }
  if (g16_nt_wide(d, n)) tiles /= 2;  // (K % 256 == 0 for every problem: exact)
  const int steps = d[0].M / 64;
  // slabs: enough workgroups for two per CU, at least four 64-row steps each, at most 32 (the partial tiles are
  // written and read back: 64 KiB per tile and slab)
  int64_t slabs = (2 * (int64_t)g16_cus()) / tiles;
  if (slabs > steps / 4) slabs = steps / 4;
  if (slabs > 32) slabs = 32;
  if (slabs < 1) slabs = 1;
  *slabs_out = (int)slabs;
  *bytes_out = slabs * elems * 4;
  return MML_OK;
}

extern "C" int64_t mml_g16_wgrad_workspace_bytes(const mml_g16_wgrad_desc* d, int32_t n) {
  int slabs = 0;
  int64_t bytes = 0;
  if (g16_wgrad_plan(d, n, &slabs, &bytes) != MML_OK) return -1;
  return bytes;
}

extern "C" int mml_g16_wgrad(const mml_g16_wgrad_desc* d, int32_t n, void* workspace, int64_t workspace_bytes, int32_t phase,
                             mml_stream_t stream) {
  int slabs = 0;
  int64_t bytes = 0;
  const int rc0 = g16_wgrad_plan(d, n, &slabs, &bytes);
  if (rc0 != MML_OK) return rc0;
  MML_REQUIRE(workspace != nullptr && workspace_bytes >= bytes && aligned16(workspace), "mml_g16_wgrad: workspace too small or misaligned (%lld bytes needed)", (long long)bytes);
  MML_REQUIRE(phase >= 0 && phase <= 2, "mml_g16_wgrad: phase must be 0, 1 or 2");
  G16NtLaunch L{};
  G16RedLaunch R{};
  const bool wide = g16_nt_wide(d, n);
  L.n_prob = R.n_prob = n;
  L.steps = d[0].M / 64;
  L.slabs = R.slabs = slabs;
  float* ws = reinterpret_cast<float*>(workspace);
  int maxred = 1;
  for (int i = 0; i < n; ++i) {
    const mml_g16_wgrad_desc& q = d[i];
    G16NtProblem& P = L.p[i];
    P.dC = q.dC; P.A = q.A; P.lddc = q.lddc; P.lda = q.lda; P.N = q.N; P.K = q.K;
    P.ntiles = q.N / 128; P.ktiles = q.K / (wide ? 256 : 128);
    P.ws = ws;
    ws += (int64_t)slabs * q.N * q.K;
    P.ws_bias = q.dbias ? ws : nullptr;
    ws += (int64_t)slabs * q.N;
    L.tiles += P.ntiles * P.ktiles;
    G16RedProblem& Q = R.p[i];
    Q.ws = P.ws; Q.ws_bias = P.ws_bias; Q.dW = q.dW; Q.dbias = q.dbias; Q.lddw = q.lddw; Q.N = q.N; Q.K = q.K;
    Q.accumulate = q.accumulate;
    const int blocks = (int)cdiv((int64_t)q.N * q.K / 4, 256);
    maxred = blocks > maxred ? blocks : maxred;
  }
  if (phase != 2) {
    if (wide) {
      MML_LAUNCH(g16_nt_kernel<256>, dim3((unsigned)(L.tiles * slabs)), dim3(256), 0, to_stream(stream), L);
      g16_last = "g16_nt_kernel<256>";
    } else {
      MML_LAUNCH(g16_nt_kernel<128>, dim3((unsigned)(L.tiles * slabs)), dim3(256), 0, to_stream(stream), L);
      g16_last = "g16_nt_kernel<128>";
    }
    const int rc = check_launch("mml_g16_wgrad");
    if (rc != MML_OK) return rc;
  }
  if (phase != 1) {
    if (maxred > 512) maxred = 512;
    MML_LAUNCH(g16_reduce_kernel, dim3((unsigned)maxred, (unsigned)n), dim3(256), 0, to_stream(stream), R);
    if (phase == 2) g16_last = "g16_reduce_kernel";
    return check_launch("mml_g16_wgrad(reduce)");
  }
  return MML_OK;
}
