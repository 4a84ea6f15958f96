// K3: grouped GEMM family for the expert / gate / tower MLPs on the fp32 MFMA pipe of gfx950.
//
// Restates DNN.forward (model/utils.py:146-161: Linear -> ReLU per layer) and the two autograd GEMMs of each
// layer's backward.  v_mfma_f32_32x32x2_f32 gives exact fp32 products/accumulation (bitwise an fmaf chain), so
// forward logits stay inside the 1e-4 tolerance the north star asks for without any reduced-precision step.
//
// One templated tile engine serves all three roles:
//   fwd   : C[M,N]  = act(A[M,K] W^T + b)                 rows<-A (reduction-contiguous), cols<-W
//   dgrad : dA[M,K] = (sum_s dC_s[M,N_s] W_s) * act'(Y)   rows<-dC_s, cols<-W_s, reduction = N_s, summed over s
//   wgrad : dW[N,K] = dC[M,N]^T A[M,K]                    rows<-dC^T, cols<-A, reduction = batch, split across
//                                                          workgroups into slabs + fixed-order reduce kernel
// Block tile 128 x BN (BN = 128 or 64), 4 waves as 2x2, each wave (64 x BN/2) = MI x NI MFMA 32x32 tiles.
// Operand tiles are staged global -> registers -> LDS with the next tile's global loads in flight during the
// MFMAs.  An operand whose reduction index is contiguous in memory ("RC") is kept as [row][k] (pitch 36 floats:
// conflict-free ds_read_b128 per 16-lane group); otherwise as [k][row] (pitch 132) and read with ds_read_b32.
// k <-> (lane half h, step j) mapping inside an 8-deep group: k = 8q + 4h + j for BOTH operands.
#include "common.hpp"
#include "lds_async.hpp"
#include "reduce.hpp"

#include <stdlib.h>

#include <type_traits>

namespace mml {

using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int BM = 128;
constexpr int BK = 32;
constexpr int PITCH_RC = BK + 4;    // [row][k] layout
constexpr int PITCH_NRC = 128 + 4;  // [k][row] layout (row extent always 128 slots; BN=64 uses half)

struct Source {
  const float* A;  // row operand
  const float* B;  // col operand
  int64_t lda, ldb;
  int32_t Kred;    // reduction extent of this source
  int32_t vecA, vecB;  // 16-byte vector loads legal
  // magnitudes of the two operands (MML_AMAX_WORDS words each: the largest is the bit pattern of an upper bound of
  // max |x|), or null: the two-plane fp16 arithmetic needs both for every source of a launch
  const uint32_t* amaxA;
  const uint32_t* amaxB;
};

enum { EPI_FWD = 0, EPI_DGRAD = 1, EPI_SLAB = 2 };

constexpr int MAX_SOURCES = 24;  // per launch, over all problems (kernel-argument block stays < 4 KiB)

struct Problem {
  int32_t src0, nsrc;  // sources [src0, src0 + nsrc) of Launch::src
  int32_t M, N;        // output extent
  float* C;
  int64_t ldc;
  const float* bias;   // fwd
  const float* Y;      // dgrad derivative source
  int64_t ldy;
  uint32_t* mask;      // relu sign bits: written by fwd, read by dgrad instead of Y (or null)
  int64_t ldmask;      // words per row
  int32_t act, accumulate;
  int32_t vec_out;     // C (and Y, when read) 16-byte aligned, ld % 4 == 0, N % 4 == 0: 16-byte epilogue accesses
  int32_t tiles_n;     // ceil(N / BN)
  int32_t tile0;       // first n-tile (fwd/dgrad) or first output tile (wgrad) of this problem in the launch
  // wgrad only
  int32_t tiles_m;
  int32_t bias_cols;   // 0: bias partials are sums of the ROW operand (rows = n); 1: of the COL operand (w_kn)
  float* bias_slab;    // [S][n] partial sums over the batch of dC, or null
  int64_t slab_off;    // float offset of this problem's slab in the workspace
  uint32_t* amax_out;  // fwd / dgrad: MML_AMAX_WORDS words that receive max |C| (atomic max of bit patterns), or null
  // K7 (PepNet gate products fused into the epilogues; the kernel-argument block has no room for nine more fields per
  // problem, so the operands travel in fields the launch kind does not use):
  //   fwd, bias_slab != null: prod = C * mul is stored as well -- mul = Y / ldy, prod = bias_slab / slab_off (pitch),
  //        amax_out2 = magnitude of prod;
  //   dgrad, bias_slab != null ("gate mode"): the input gradient v of h (.) g is not stored, its two factors' gradients
  //        are: dH (+)= v g act_h'(h) -> C / ldc (accumulate, amax_out), dG (+)= v h act_g'(g) -> bias_slab / slab_off
  //        (accumulate = bias_cols, amax_out2); h = Y / ldy with act_h = act, g = bias / ldmask with act_g = tiles_m.
  uint32_t* amax_out2;
};

// ---- operand magnitudes (mml_amax_*, the amax fields of the GEMM descriptors) ----
// A magnitude slot is MML_AMAX_WORDS consecutive words; producers atomicMax the bit pattern of |x| into ANY of them
// (which one is picked from the workgroup / wave number, so that thousands of same-address atomics do not serialise
// at the memory side), consumers take the largest.
__device__ __forceinline__ uint32_t amax_load(const uint32_t* p) {
  if (!p) return 0x3f800000u;  // (no magnitude given: scale 2^14; the host only selects the fp16 form when all are there)
  uint32_t m = 0;
#pragma unroll
  for (int i = 0; i < MML_AMAX_WORDS; ++i) m = p[i] > m ? p[i] : m;
  return m;
}
// power-of-two exponent k with |x| 2^k < 2^15 for every |x| <= the slot's value (fp16: largest finite 65504)
__device__ __forceinline__ int amax_scale_exp(uint32_t bits) {
  int e = (int)((bits >> 23) & 0xffu);  // |x| < 2^(e - 126)
  if (e == 255) return 0;               // Inf / NaN in the operand: they propagate whatever the scale
  int k = 141 - e;
  return k > 110 ? 110 : (k < -110 ? -110 : k);
}
__device__ __forceinline__ float pow2f(int k) { return __uint_as_float((uint32_t)(127 + k) << 23); }
// wave-wide maximum of a per-lane bit pattern, then ONE atomic per wave into the slot
__device__ __forceinline__ void amax_publish(uint32_t am, uint32_t* slot, int salt) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const uint32_t t = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(((threadIdx.x & 63) ^ o) << 2), (int)am);
    am = t > am ? t : am;
  }
  if ((threadIdx.x & 63) == 0 && am) atomicMax(slot + (salt & (MML_AMAX_WORDS - 1)), am);
}

// ---- pre-cut weights (mml_gemm_planes_cut): the two fp16 planes of a weight matrix, laid out so that they travel through
// the SAME global -> LDS image and fragment reads as the floats (see planes_of) ----
constexpr int PLANES_PER_LAUNCH = 27;  // (kernel-argument block: 27 x 144 B + 116 B < 4 KiB)
struct PlanesLaunch {
  mml_planes_desc d[PLANES_PER_LAUNCH];
  int32_t item0[PLANES_PER_LAUNCH + 1];  // first work item of matrix i (one item = one block of 16 along the reduction)
  int32_t n;
};
static_assert(sizeof(PlanesLaunch) <= 4096, "PlanesLaunch must fit the kernel-argument block");
// the lane's eight k of a 16-block, in fragment order: S_h = {4h .. 4h+3, 8+4h .. 8+4h+3} (RawFrag::get)
__device__ __forceinline__ int planes_k(int h, int e) { return 4 * h + (e & 3) + 8 * (e >> 2); }
__global__ __launch_bounds__(256) void planes_cut_kernel(const PlanesLaunch L) {
  const mml_planes_desc& D = L.d[blockIdx.y];
  // the group's exponent, once per workgroup: lane w of the first n_amax * MML_AMAX_WORDS threads reads one slot word
  __shared__ uint32_t mx;
  if (threadIdx.x == 0) mx = 0u;
  __syncthreads();
  if (D.W2 == nullptr) {
    if ((int)threadIdx.x < D.n_amax * MML_AMAX_WORDS) {
      const uint32_t* sl = D.amax[threadIdx.x / MML_AMAX_WORDS];
      atomicMax(&mx, sl[threadIdx.x % MML_AMAX_WORDS]);
    }
  } else if ((int)threadIdx.x < D.n_amax) {
    // K6: the matrix is a product of two factors: its magnitude bound is the product of theirs (a non-finite factor
    // bound stays non-finite: exponent 0 below)
    const float a = __uint_as_float(amax_load(D.amax[threadIdx.x])), b = __uint_as_float(amax_load(D.amax[D.n_amax + threadIdx.x]));
    atomicMax(&mx, __float_as_uint(a * b));
  }
  __syncthreads();
  // (amax_scale_exp is monotone: the exponent of the largest magnitude is the smallest exponent of the group; an
  // Inf / NaN pattern gives 0 like the in-kernel cut of the operand that holds it)
  int k = amax_scale_exp(mx);
  if (D.n_amax > 1 && (mx >> 23) >= 255u) {  // a non-finite slot must not hide the finite ones' exponents: redo per slot
    k = 110;
    for (int a = 0; a < D.n_amax; ++a) {
      uint32_t bits = amax_load(D.amax[a]);
      if (D.W2) bits = __float_as_uint(__uint_as_float(bits) * __uint_as_float(amax_load(D.amax[D.n_amax + a])));
      const int ka = amax_scale_exp(bits);
      k = ka < k ? ka : k;
    }
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) *D.kexp = k;
  const float s = pow2f(k);
  const int64_t total = L.item0[blockIdx.y + 1] - L.item0[blockIdx.y];
  for (int64_t loc = (int64_t)blockIdx.x * 256 + threadIdx.x; loc < total; loc += (int64_t)gridDim.x * 256) {
    float x[16];
    int64_t base, stride, pbase, pstride;  // element e of the block: W[base + e stride], word e: planes[pbase + e pstride]
    int live = 16;                           // (ROWS with a partial last block: the missing columns count as zeros)
    const int64_t ldp = D.ldp ? D.ldp : D.ld;
    if (D.layout == MML_PLANES_ROWS) {
      const int nb = (D.cols + 15) / 16;
      const int64_t r = loc / nb;
      const int c0 = 16 * (int)(loc - r * nb);
      base = r * D.ld + c0;
      pbase = r * ldp + c0;
      stride = pstride = 1;
      live = D.cols - c0 < 16 ? D.cols - c0 : 16;
    } else {
      const int64_t rb = loc / D.cols;
      base = 16 * rb * D.ld + (loc - rb * D.cols);
      pbase = 16 * rb * ldp + (loc - rb * D.cols);
      stride = D.ld;
      pstride = ldp;
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) x[e] = e < live ? D.W[base + e * stride] : 0.f;
    if (D.W2) {  // K6: the derived weight W (.) W2 (the fp32 product the element-wise kernel would have stored)
      int64_t base2, stride2;
      if (D.layout == MML_PLANES_ROWS) {
        const int nb = (D.cols + 15) / 16;
        const int64_t r = loc / nb;
        base2 = r * D.ld2 + 16 * (int)(loc - r * nb);
        stride2 = 1;
      } else {
        const int64_t rb = loc / D.cols;
        base2 = 16 * rb * D.ld2 + (loc - rb * D.cols);
        stride2 = D.ld2;
      }
#pragma unroll
      for (int e = 0; e < 16; ++e) x[e] *= e < live ? D.W2[base2 + e * stride2] : 0.f;
    }
    uint32_t hp[16], lp[16];  // half bit patterns
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const float y = x[e] * s;            // exact: s is a power of two
      const _Float16 h = (_Float16)y;      // round to nearest even
      const float r = y - (float)h;        // exact
      const _Float16 l = (_Float16)r;
      hp[e] = (uint32_t)__builtin_bit_cast(uint16_t, h);
      lp[e] = (uint32_t)__builtin_bit_cast(uint16_t, l);
    }
    // word 4h + i = halves (S_h[2i], S_h[2i+1]) of the h plane, word 8 + 4h + i the same of the l plane
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int k0 = planes_k(h, 2 * i), k1 = planes_k(h, 2 * i + 1);
        D.planes[pbase + (4 * h + i) * pstride] = hp[k0] | (hp[k1] << 16);
        D.planes[pbase + (8 + 4 * h + i) * pstride] = lp[k0] | (lp[k1] << 16);
      }
  }
}

struct Launch {
  Source src[MAX_SOURCES];
  Problem p[MML_MAX_GROUP];
  int32_t n;
  int32_t total_ntiles;  // fwd/dgrad: sum of tiles_n ; wgrad: sum of tiles_m*tiles_n
  int32_t tiles_m;       // fwd/dgrad: common M tiles
  int32_t splits;        // wgrad: S
  int32_t chunk;         // wgrad: batch rows per split (multiple of BK)
  float* slab;           // wgrad workspace
};

__device__ __forceinline__ float act_fwd(float v, int act) {
  if (act == MML_ACT_RELU) return v > 0.f ? v : 0.f;
  if (act == MML_ACT_SIGMOID) return 1.f / (1.f + __expf(-v));
  if (act == MML_ACT_SIGMOID2) return 2.f / (1.f + __expf(-v));
  return v;
}

__device__ __forceinline__ float act_bwd(float y, int act) {
  if (act == MML_ACT_RELU) return y > 0.f ? 1.f : 0.f;
  if (act == MML_ACT_SIGMOID) return y * (1.f - y);
  if (act == MML_ACT_SIGMOID2) {
    const float s = 0.5f * y;
    return 2.f * s * (1.f - s);
  }
  return 1.f;
}

// ---- global -> register tile fetch ---------------------------------------------------------------
// RC: tile is [ROWS rows][BK k], memory k-contiguous. thread t, piece c: row = (t + 256c) / 8, k4 = (t + 256c) % 8.
// !RC: tile is [BK k][ROWS rows], memory row-contiguous. row4 = idx % (ROWS/4), k = idx / (ROWS/4).
template <bool RC, int ROWS>
__device__ __forceinline__ void fetch_tile(float4 (&r)[ROWS / 32], const float* __restrict__ base, int64_t ld,
                                           int row0, int nrows, int k0, int kend, bool vec, int tid) {
  constexpr int PIECES = ROWS / 32;  // float4 per thread: ROWS*BK/4/256
#pragma unroll
  for (int c = 0; c < PIECES; ++c) {
    const int idx = tid + 256 * c;
    int row, k;
    if (RC) {
      row = row0 + idx / (BK / 4);
      k = k0 + (idx % (BK / 4)) * 4;
    } else {
      // thread = (row quad, PIECES consecutive k): lanes walk the row quads (512 contiguous bytes per k)
      row = row0 + (tid % (ROWS / 4)) * 4;
      k = k0 + (tid / (ROWS / 4)) * PIECES + c;
    }
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (RC) {
      if (row < nrows && k < kend) {
        const float* p = base + (int64_t)row * ld + k;
        if (vec && k + 3 < kend) {
          v = *reinterpret_cast<const float4*>(p);
        } else {
          v.x = p[0];
          if (k + 1 < kend) v.y = p[1];
          if (k + 2 < kend) v.z = p[2];
          if (k + 3 < kend) v.w = p[3];
        }
      }
    } else {
      if (k < kend && row < nrows) {
        const float* p = base + (int64_t)k * ld + row;
        if (vec && row + 3 < nrows) {
          v = *reinterpret_cast<const float4*>(p);
        } else {
          v.x = p[0];
          if (row + 1 < nrows) v.y = p[1];
          if (row + 2 < nrows) v.z = p[2];
          if (row + 3 < nrows) v.w = p[3];
        }
      }
    }
    r[c] = v;
  }
}

template <bool RC, int ROWS>
__device__ __forceinline__ void stash_tile(const float4 (&r)[ROWS / 32], float* lds, int tid) {
  constexpr int PIECES = ROWS / 32;
#pragma unroll
  for (int c = 0; c < PIECES; ++c) {
    const int idx = tid + 256 * c;
    if (RC) {
      const int row = idx / (BK / 4), k4 = idx % (BK / 4);
      *reinterpret_cast<float4*>(lds + row * PITCH_RC + k4 * 4) = r[c];
    } else {
      const int row4 = tid % (ROWS / 4), k = (tid / (ROWS / 4)) * PIECES + c;
      *reinterpret_cast<float4*>(lds + k * PITCH_NRC + row4 * 4) = r[c];
    }
  }
}


// fragment for 8-deep group q: 4 values (j = 0..3) for row `row` (tile-local), lane half h.
template <bool RC>
__device__ __forceinline__ float4 read_frag(const float* lds, int row, int q, int h) {
  if (RC) {
    return *reinterpret_cast<const float4*>(lds + row * PITCH_RC + q * 8 + h * 4);
  } else {
    const float* p = lds + (q * 8 + h * 4) * PITCH_NRC + row;
    return make_float4(p[0], p[PITCH_NRC], p[2 * PITCH_NRC], p[3 * PITCH_NRC]);
  }
}

// ---- the tile engine -------------------------------------------------------------------------------
// Accumulates sum_s sum_{k in [kbeg_s, kend_s)} rowop_s[row0+i][k] * colop_s[col0+j][k] into acc.
template <bool ARC, bool BRC, int BN>
__device__ __forceinline__ void tile_mainloop(f32x16 (&acc)[2][BN / 64], const Launch& L, const Problem& P, int row0,
                                              int col0, int kbeg, int klen_limit, float* ldsA, float* ldsB,
                                              float* bias_part /* per-thread partial batch sum of dC, or null */) {
  constexpr int NI = BN / 64;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, h = lane >> 5;
  float4 ra[BM / 32], rb[BN / 32];

  for (int s = 0; s < P.nsrc; ++s) {
    const Source& S = L.src[P.src0 + s];
    int k0 = kbeg;
    int kend = S.Kred;
    if (klen_limit > 0 && kbeg + klen_limit < kend) kend = kbeg + klen_limit;
    if (k0 >= kend) continue;
    fetch_tile<ARC, BM>(ra, S.A, S.lda, row0, P.M, k0, kend, S.vecA, tid);
    fetch_tile<BRC, BN>(rb, S.B, S.ldb, col0, P.N, k0, kend, S.vecB, tid);
    for (; k0 < kend; k0 += BK) {
      __syncthreads();  // previous step's LDS reads are done
      stash_tile<ARC, BM>(ra, ldsA, tid);
      stash_tile<BRC, BN>(rb, ldsB, tid);
      __syncthreads();
      if (k0 + BK < kend) {  // next tile's loads fly during the MFMAs
        fetch_tile<ARC, BM>(ra, S.A, S.lda, row0, P.M, k0 + BK, kend, S.vecA, tid);
        fetch_tile<BRC, BN>(rb, S.B, S.ldb, col0, P.N, k0 + BK, kend, S.vecB, tid);
      }
      if (bias_part != nullptr) {
        // batch sums of dC over this k-slab (wgrad bias gradient); both operands are in [k][row] layout here
        if (tid < (P.bias_cols ? BN : BM)) {
          float sacc = 0.f;
          const float* src = P.bias_cols ? ldsB : ldsA;
#pragma unroll 8
          for (int k = 0; k < BK; ++k) sacc += src[k * PITCH_NRC + tid];
          *bias_part += sacc;
        }
      }
#pragma unroll
      for (int q = 0; q < BK / 8; ++q) {
        float4 fa[2], fb[NI];
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) fa[mi] = read_frag<ARC>(ldsA, wm * 64 + mi * 32 + l31, q, h);
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) fb[ni] = read_frag<BRC>(ldsB, wn * (BN / 2) + ni * 32 + l31, q, h);
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
          for (int ni = 0; ni < NI; ++ni) {
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[mi].x, fb[ni].x, acc[mi][ni], 0, 0, 0);
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[mi].y, fb[ni].y, acc[mi][ni], 0, 0, 0);
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[mi].z, fb[ni].z, acc[mi][ni], 0, 0, 0);
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[mi].w, fb[ni].w, acc[mi][ni], 0, 0, 0);
          }
      }
    }
  }
}


// BN = 64 tiles need <= 128 VGPRs, so four workgroups (one wave each per SIMD) fit a CU; the images take
// 36 KiB of LDS per workgroup (4 x 36 = 144 KiB <= 160 KiB).
template <bool ARC, bool BRC, int BN, int EPI>
__global__ __launch_bounds__(256, (BN == 64 ? 4 : 2)) void gemm_kernel(const Launch L) {
  constexpr int NI = BN / 64;
  constexpr int TILE_FLOATS = 128 * PITCH_RC;  // f32 images: 128*36 = 4608 floats (>= 32*132)
  __shared__ __attribute__((aligned(16))) float lds[2 * TILE_FLOATS];
  float* ldsA = lds;
  float* ldsB = lds + TILE_FLOATS;

  const int vid = xcd_remap(blockIdx.x, gridDim.x);
  int pi = 0, row0, col0, kbeg = 0, klen = 0, split = 0;
  if (EPI != EPI_SLAB) {
    const int mt = vid / L.total_ntiles;
    int j = vid - mt * L.total_ntiles;
    while (pi + 1 < L.n && j >= L.p[pi + 1].tile0) ++pi;
    j -= L.p[pi].tile0;
    row0 = mt * BM;
    col0 = j * BN;
  } else {
    split = vid / L.total_ntiles;
    int j = vid - split * L.total_ntiles;
    while (pi + 1 < L.n && j >= L.p[pi + 1].tile0) ++pi;
    j -= L.p[pi].tile0;
    row0 = (j / L.p[pi].tiles_n) * BM;
    col0 = (j % L.p[pi].tiles_n) * BN;
    kbeg = split * L.chunk;
    klen = L.chunk;
  }
  const Problem& P = L.p[pi];

  f32x16 acc[2][NI];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

  float bsum = 0.f;
  const bool want_bias = (EPI == EPI_SLAB) && P.bias_slab != nullptr && (P.bias_cols ? row0 == 0 : col0 == 0);
  tile_mainloop<ARC, BRC, BN>(acc, L, P, row0, col0, kbeg, klen, ldsA, ldsB, want_bias ? &bsum : nullptr);

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, h = lane >> 5;

  if (EPI == EPI_SLAB) {
    if (want_bias) {
      if (!P.bias_cols) {
        if (tid < BM && row0 + tid < P.M) P.bias_slab[(int64_t)split * P.M + row0 + tid] = bsum;
      } else {
        if (tid < BN && col0 + tid < P.N) P.bias_slab[(int64_t)split * P.N + col0 + tid] = bsum;
      }
    }
    float* slab = L.slab + P.slab_off + (int64_t)split * P.M * P.N;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        const int col = col0 + wn * (BN / 2) + ni * 32 + l31;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = row0 + wm * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
          if (row < P.M && col < P.N) slab[(int64_t)row * P.N + col] = acc[mi][ni][r];
        }
      }
    return;
  }

  uint32_t am = 0;
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      const int col = col0 + wn * (BN / 2) + ni * 32 + l31;
      if (col >= P.N) continue;
      float b = 0.f;
      if (EPI == EPI_FWD && P.bias) b = P.bias[col];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = row0 + wm * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (row >= P.M) continue;
        float v = acc[mi][ni][r];
        float* dst = P.C + (int64_t)row * P.ldc + col;
        if (EPI == EPI_FWD) {
          v = act_fwd(v + b, P.act);
          if (P.mask) {  // relu sign bits: the 32 lanes of a half-wave are the 32 columns of one mask word
            const unsigned long long bal = __ballot(v > 0.f);
            if (l31 == 0) P.mask[(int64_t)row * P.ldmask + (col >> 5)] = (uint32_t)(h ? (bal >> 32) : bal);
          }
        } else {
          if (P.mask) {
            if (!((P.mask[(int64_t)row * P.ldmask + (col >> 5)] >> l31) & 1u)) v = 0.f;
          } else if (P.act != MML_ACT_NONE) {
            v *= act_bwd(P.Y[(int64_t)row * P.ldy + col], P.act);
          }
          if (P.accumulate) v += *dst;
        }
        *dst = v;
        am = max(am, __float_as_uint(v) & 0x7fffffffu);
      }
    }
  if (P.amax_out) amax_publish(am, P.amax_out, blockIdx.x * 4 + wave);
}

// ====================================================================================================
// LDS-DMA operand staging (used by gemm_pipe_kernel below)
//
// Same tile geometry and MFMA map as gemm_kernel, different staging: operand tiles go global -> LDS with
// global_load_lds_dwordx4 (no staging VGPRs, no ds_write, no per-lane guards), K step 16, ahead of the MFMAs, with ONE
// raw s_barrier per k-step and counted s_waitcnt vmcnt(N).
//   * reduction-contiguous operand ([rows][16 k], 64-B rows): the 16-B chunk a lane fetches is XOR-swizzled on the
//     SOURCE side (phys = chunk ^ ((row >> 2) & 3)) because the LDS destination of a wave-instruction is lane-linear;
//     reading (row, chunk) at phys is conflict-free for the ds_read_b128 lane groups;
//   * row-contiguous operand ([16 k][rows]): plain image, fragments read with 2 x ds_read2st64_b32 (lanes =
//     consecutive rows -> conflict-free).
// Rows past the end of an operand are clamped to the last valid row / chunk (their products only reach outputs the
// epilogue discards), so no zero-fill is needed.  Eligibility (checked on the host, else gemm_kernel runs): every
// reduction extent a multiple of 16, 16-byte aligned operands with ld % 4 == 0, row-contiguous extents % 4 == 0.
// ====================================================================================================
constexpr int GK = 16;
constexpr int GA = 128 * GK;  // floats per A stage


// The lane's 8 k-values of one 32-row sub-tile for one k-step, as the LDS reads deliver them:
//   reduction-contiguous image: two ds_read_b128 (k-groups 0 and 1);
//   row-contiguous image      : four ds_read2st64_b32 (consecutive k are ROWS*4 B apart = 1 or 2 units of 256 B).
template <bool RC>
struct RawFrag {
  f32x4_t q0, q1;
  __device__ __forceinline__ void landed() { lds_landed(q0); lds_landed(q1); }
  __device__ __forceinline__ void get(float (&x)[8]) const {
    x[0] = q0.x; x[1] = q0.y; x[2] = q0.z; x[3] = q0.w; x[4] = q1.x; x[5] = q1.y; x[6] = q1.z; x[7] = q1.w;
  }
};
template <>
struct RawFrag<false> {
  f32x2_t p0, p1, p2, p3;
  __device__ __forceinline__ void landed() { lds_landed(p0); lds_landed(p1); lds_landed(p2); lds_landed(p3); }
  __device__ __forceinline__ void get(float (&x)[8]) const {
    x[0] = p0.x; x[1] = p0.y; x[2] = p1.x; x[3] = p1.y; x[4] = p2.x; x[5] = p2.y; x[6] = p3.x; x[7] = p3.y;
  }
};
// RC : a0 / a1 = byte address of (row, phys chunk of k-group 0 / 1); OFF = stage / +32-row immediate
// NRC: a = byte address of (k = 4h, row); U = 256-B unit offset of the stage (+ sub-tile handled by the caller's address)
template <int OFF>
__device__ __forceinline__ void read_frag_rc(uint32_t a0, uint32_t a1, RawFrag<true>& r) {
  r.q0 = ds_read128<OFF>(a0);
  r.q1 = ds_read128<OFF>(a1);
}
template <int ROWS, int SOFF>
__device__ __forceinline__ void read_frag_nrc(uint32_t a, RawFrag<false>& r) {
  constexpr int SU = ROWS * 4 / 256;
  constexpr int U0 = SOFF / 256, U1 = (SOFF + 8 * ROWS * 4) / 256;
  r.p0 = ds_read2st64<U0, U0 + SU>(a);
  r.p1 = ds_read2st64<U0 + 2 * SU, U0 + 3 * SU>(a);
  r.p2 = ds_read2st64<U1, U1 + SU>(a);
  r.p3 = ds_read2st64<U1 + 2 * SU, U1 + 3 * SU>(a);
}


// ====================================================================================================
// Software-pipelined direct-to-LDS kernel ("pipe"): same tiles, operand images and MFMA map as gemm_glds_kernel,
// different schedule.
//   * four LDS stages; the LDS-DMA of step i+3 is issued at the top of step i (two full steps of flight, as before);
//   * the fragment reads of step i+1 are issued at the top of step i into a second register set, so no MFMA ever
//     waits on LDS latency;
//   * the fp32 -> bf16-plane cuts of the NEXT operands are placed in the shadow of the current MFMA blocks
//     (sched_group_barrier: 1 MFMA : a few VALU), instead of one VALU phase followed by one MFMA phase;
//   * DMA source pointers are per-lane running pointers (set up once per tile / source, one 64-bit add per load);
//   * one copy of the epilogue (the step body is dispatched on i % 4 from a plain loop).
// Per step and wave (BN = 128): carry-in = prepared A0, B0 and raw A1, B1 of this step;
//   block 1: acc00 += A0 x B0   ||  prepare B1          block 3: acc10 += A1 x B0  ||  prepare next B0
//   block 2: acc01 += A0 x B1   ||  prepare A1          block 4: acc11 += A1 x B1  ||  prepare next A0
// ====================================================================================================
constexpr int PSTAGES = 4;


template <int EMU>
struct Prep {
  bf16x8 p[EMU == 0 ? 1 : EMU];
};
template <>
struct Prep<0> {
  float x[8];
};
// EMU 2: two fp16 planes of the scaled operand (three v_mfma_f32_32x32x16_f16 per 16-k block: hh, hl, lh)
template <>
struct Prep<2> {
  f16x8 h, l;
};

template <int EMU, bool RC>
__device__ __forceinline__ void prep_frag(const RawFrag<RC>& r, Prep<EMU>& o, const float scale = 1.f) {
  if constexpr (EMU == 0) {
    r.get(o.x);
  } else if constexpr (EMU == 2) {
    float x[8];
    r.get(x);
    split_f16_planes(x, scale, o.h, o.l);
  } else {
    float x[8];
    r.get(x);
    split_planes<EMU>(x, o.p);
  }
}

template <int EMU, bool RC>
__device__ __forceinline__ void prep_frag_b(const RawFrag<RC>& r, Prep<EMU>& o, const float scale = 1.f) {
  if constexpr (EMU == 2) {
    prep_frag<EMU>(r, o, scale);
    return;
  }
  prep_frag<EMU>(r, o);
}

// BPL: a fragment of the pre-cut column operand IS its two planes (mml_gemm_planes_cut wrote the 8 halves of h where the
// first four floats of the lane's 8 values would be, the 8 halves of l in place of the last four)
__device__ __forceinline__ void planes_of(const RawFrag<true>& r, Prep<2>& o) {
  o.h = __builtin_bit_cast(f16x8, r.q0);
  o.l = __builtin_bit_cast(f16x8, r.q1);
}
__device__ __forceinline__ void planes_of(const RawFrag<false>& r, Prep<2>& o) {
  const f32x4_t hh = __builtin_shufflevector(r.p0, r.p1, 0, 1, 2, 3);
  const f32x4_t ll = __builtin_shufflevector(r.p2, r.p3, 0, 1, 2, 3);
  o.h = __builtin_bit_cast(f16x8, hh);
  o.l = __builtin_bit_cast(f16x8, ll);
}

// acc += rows(a) x cols(b); the COLUMN operand is the MFMA's A input (a lane then owns an output row, see epilogue)
template <int EMU>
__device__ __forceinline__ void mma_block(f32x16& acc, const Prep<EMU>& a, const Prep<EMU>& b) {
  if constexpr (EMU == 0) {
#pragma unroll
    for (int k = 0; k < 8; ++k) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(b.x[k], a.x[k], acc, 0, 0, 0);
  } else if constexpr (EMU == 2) {  // smallest products first; l x l (<= 2^-22 of the product) is dropped
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(b.l, a.h, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(b.h, a.l, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(b.h, a.h, acc, 0, 0, 0);
  } else {
#pragma unroll
    for (int lvl = EMU - 1; lvl >= 0; --lvl)  // smallest products first
#pragma unroll
      for (int ia = 0; ia <= lvl; ++ia) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b.p[lvl - ia], a.p[ia], acc, 0, 0, 0);
      }
  }
}

// EMU 2: one product block (three dependent MFMAs, ~32 cycles each) with the cut of ONE raw fragment dealt into their
// shadows by hand: 10 + 10 + 4 VALU instructions.
template <bool RC>
__device__ __forceinline__ void mma_prep_f16(f32x16& acc, const Prep<2>& a, const Prep<2>& b, const RawFrag<RC>& r,
                                             const float scale, Prep<2>& o) {
  float x[8];
  r.get(x);
  F16Cut c;
  __builtin_amdgcn_sched_barrier(0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(b.l, a.h, acc, 0, 0, 0);
  __builtin_amdgcn_sched_barrier(0);
  f16_cut_hr2(x, scale, c, 0);
  __builtin_amdgcn_sched_barrier(0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(b.h, a.l, acc, 0, 0, 0);
  __builtin_amdgcn_sched_barrier(0);
  f16_cut_hr2(x, scale, c, 2);
  __builtin_amdgcn_sched_barrier(0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(b.h, a.h, acc, 0, 0, 0);
  __builtin_amdgcn_sched_barrier(0);
  f16_cut_l(c, 0);
  f16_cut_l(c, 1);
  f16_cut_l(c, 2);
  f16_cut_l(c, 3);
  f16_cut_done(c, o.h, o.l);
  __builtin_amdgcn_sched_barrier(0);
}

// BPL (the column operand arrives pre-cut): the cut of the ONE fragment a pair of product blocks has to prepare, dealt over
// the six MFMAs of the pair: first block 6 + 4 + 6, second block 4 + 4 + 0 instructions.
template <bool RC>
__device__ __forceinline__ void mma_cut_first(f32x16& acc, const Prep<2>& a, const Prep<2>& b, const RawFrag<RC>& r,
                                              const float scale, F16Cut& c) {
  float x[8];
  r.get(x);
  __builtin_amdgcn_sched_barrier(0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(b.l, a.h, acc, 0, 0, 0);
  __builtin_amdgcn_sched_barrier(0);
  f16_cut_a2(x, scale, c, 0);
  __builtin_amdgcn_sched_barrier(0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(b.h, a.l, acc, 0, 0, 0);
  __builtin_amdgcn_sched_barrier(0);
  f16_cut_b2(x, scale, c, 0);
  __builtin_amdgcn_sched_barrier(0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(b.h, a.h, acc, 0, 0, 0);
  __builtin_amdgcn_sched_barrier(0);
  f16_cut_a2(x, scale, c, 2);
  __builtin_amdgcn_sched_barrier(0);
}
template <bool RC>
__device__ __forceinline__ void mma_cut_second(f32x16& acc, const Prep<2>& a, const Prep<2>& b, const RawFrag<RC>& r,
                                               const float scale, F16Cut& c, Prep<2>& o) {
  float x[8];
  r.get(x);
  __builtin_amdgcn_sched_barrier(0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(b.l, a.h, acc, 0, 0, 0);
  __builtin_amdgcn_sched_barrier(0);
  f16_cut_b2(x, scale, c, 2);
  __builtin_amdgcn_sched_barrier(0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(b.h, a.l, acc, 0, 0, 0);
  __builtin_amdgcn_sched_barrier(0);
  f16_cut_l(c, 0);
  f16_cut_l(c, 1);
  f16_cut_l(c, 2);
  f16_cut_l(c, 3);
  f16_cut_done(c, o.h, o.l);
  __builtin_amdgcn_sched_barrier(0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(b.h, a.h, acc, 0, 0, 0);
  __builtin_amdgcn_sched_barrier(0);
}

// Keeps a prepared operand's conversion where it was written: without a use in the same block hipcc sinks the VALU
// work of an operand that is only consumed by the NEXT step out of the MFMA shadow it was placed in.
template <int EMU>
__device__ __forceinline__ void pin_prep(Prep<EMU>& o) {
  if constexpr (EMU == 0) {
    asm volatile("" : "+v"(o.x[0]), "+v"(o.x[1]), "+v"(o.x[2]), "+v"(o.x[3]));
    asm volatile("" : "+v"(o.x[4]), "+v"(o.x[5]), "+v"(o.x[6]), "+v"(o.x[7]));
  } else if constexpr (EMU == 2) {
    asm volatile("" : "+v"(o.h));
    asm volatile("" : "+v"(o.l));
  } else {
#pragma unroll
    for (int p = 0; p < EMU; ++p) asm volatile("" : "+v"(o.p[p]));
  }
}

// interleave hint for one block: NM MFMAs, each followed by NV VALU instructions of the neighbouring prepare
template <int NM, int NV>
__device__ __forceinline__ void interleave_hint() {
#pragma unroll
  for (int t = 0; t < NM; ++t) {
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    if (NV > 0) __builtin_amdgcn_sched_group_barrier(0x002, NV, 0);
  }
}

template <int ACT>
__device__ __forceinline__ float act_fwd_t(float v) {
  if (ACT == MML_ACT_RELU) return v > 0.f ? v : 0.f;
  if (ACT == MML_ACT_SIGMOID) return 1.f / (1.f + __expf(-v));
  if (ACT == MML_ACT_SIGMOID2) return 2.f / (1.f + __expf(-v));
  return v;
}
template <int ACT>
__device__ __forceinline__ float act_bwd_t(float y) {
  if (ACT == MML_ACT_RELU) return y > 0.f ? 1.f : 0.f;
  if (ACT == MML_ACT_SIGMOID) return y * (1.f - y);
  if (ACT == MML_ACT_SIGMOID2) {
    const float s = 0.5f * y;
    return 2.f * s * (1.f - s);
  }
  return 1.f;
}

// BCOLS (weight-gradient launches only): the bias partials are batch sums of the COLUMN operand ([K,N] weights) instead
// of the row operand; uniform per launch (the host groups problems by layout), so the per-step sums carry no branch.
// BPL (EMU 2, forward / input-gradient launches): the column operand (the weights) arrives PRE-CUT -- Source::B points at
// the plane image mml_gemm_planes_cut wrote (same shape, pitch and LDS image as the floats), Source::amaxB at the
// exponent it was scaled with -- so its fragments are used as they land, without the in-register cut.
// K7 (forward / input-gradient launches): the epilogue also handles the PepNet gate products (Problem: second output).
// Its own instantiations: the extra operands cost registers (the plain kernels spilled with the code merely present).
template <bool ARC, bool BRC, int BN, int EPI, int EMU, bool BCOLS = false, bool BPL = false, bool K7 = false>
// (128 x 64 input-gradient kernels: two workgroups per CU like the wide tiles -- their epilogue (mask words, Y, the
// accumulate target) does not fit the 168 registers three would leave)
__global__ __launch_bounds__(256, ((BN == 64 && EPI != EPI_DGRAD && !K7) ? 3 : 2)) void gemm_pipe_kernel(const Launch Larg) {
  typedef const __attribute__((address_space(4))) Launch KLaunch;
  KLaunch& L = *(KLaunch*)__builtin_amdgcn_kernarg_segment_ptr();  // see gemm_glds_kernel
  constexpr int NI = BN / 64;
  constexpr int GB = BN * GK;
  constexpr int STG = GA + GB;       // floats per stage
  constexpr int LOADS = 2 + NI;      // LDS-DMA instructions per wave per k-step
  constexpr int NSTORE = 8 * NI;     // 16-byte epilogue stores per wave of an interior tile
  constexpr int NBIAS = (EPI == EPI_FWD) ? 1 : 0;  // the next tile's bias DMA follows the stores
  constexpr int AFTER_EPI = (LOADS + NSTORE + NBIAS < 63) ? LOADS + NSTORE + NBIAS : 63;
  constexpr int NMFMA = EMU == 0 ? 8 : (EMU == 1 ? 1 : (EMU == 2 ? 3 : 6));  // MFMAs per block
  constexpr int NVALU = EMU == 0 ? 0 : 8;  // VALU slots per MFMA of a block with one prepare (EMU 1: 4 cvt_pk)
  // + one 64-float bias slot per wave + one word per problem: the largest |output| this workgroup stored for it
  __shared__ __attribute__((aligned(16))) float lds[PSTAGES * STG + 256 + 2 * MML_MAX_GROUP];
  const int tid = threadIdx.x;
  // wave as a SCALAR: the LDS destinations of the DMA (wave-dependent) then live in SGPRs; as a VGPR expression every
  // DMA needed v_readfirstlane -> s_mov m0 inside the loop, a VALU -> SALU hand-over that waits for the wave's MFMAs
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, h = lane >> 5;

  const int64_t total = (int64_t)(EPI == EPI_SLAB ? L.splits : L.tiles_m) * L.total_ntiles;
  struct Cursor {
    int64_t vid;
    int pi, row0, col0, split;
    int s, k0, kend;
    int M, N, nsrc, src0;
    bool ok;
    bool want_bias, bias_cols;  // wgrad: this tile also sums the bias partials (of the row / column operand)
    bool short_tile;            // fewer than three k-steps: the bias DMA may still be in flight at the epilogue
    bool counted;               // interior tile with 16-byte stores: its epilogue leaves exactly NSTORE stores in flight
    float sA, sB, inv;          // EMU 2: power-of-two scales of the row / column operand of this problem, 1 / (sA sB)
  };
  // EMU 2: ONE scale pair per problem (the smallest over its sources: accumulation across sources needs a common unit)
  auto problem_scales = [&](const int pi, float& sA, float& sB, float& inv) __attribute__((always_inline)) {
    int kA = 110, kB = 110;
    const int s0 = L.p[pi].src0, ns = L.p[pi].nsrc;
    for (int s2 = 0; s2 < ns; ++s2) {
      const int ka = amax_scale_exp(amax_load(L.src[s0 + s2].amaxA));
      kA = ka < kA ? ka : kA;
      if constexpr (!BPL) {
        const int kb = amax_scale_exp(amax_load(L.src[s0 + s2].amaxB));
        kB = kb < kB ? kb : kB;
      }
    }
    if constexpr (BPL) {  // the exponent the planes were cut with (one per problem: its sources were cut as a group)
      kB = *reinterpret_cast<const int32_t*>(L.src[s0].amaxB);
      // (the planes are what they are: where the exponents add up beyond fp32's range the row operand alone gives way)
      if (kA + kB > 126) kA = 126 - kB;
      if (kA + kB < -126) kA = -126 - kB;
    }
    // 1 / (sA sB) must be ONE fp32 number (the epilogue multiplies once): where the exponents add up beyond its range
    // -- two operands below 2^-48, whose products fp32 can barely hold -- both scales give way equally
    const int over = kA + kB - 126, under = -126 - (kA + kB);
    if (over > 0) { kA -= (over + 1) >> 1; kB -= over >> 1; }
    if (under > 0) { kA += (under + 1) >> 1; kB += under >> 1; }
    kA = __builtin_amdgcn_readfirstlane(kA);
    kB = __builtin_amdgcn_readfirstlane(kB);
    sA = pow2f(kA);
    sB = pow2f(kB);
    inv = pow2f(-(kA + kB));
  };
  // ... of the tile a virtual id names (the look-ahead of the last step of a tile)
  auto scales_of_vid = [&](const int64_t vid, float& sA, float& sB) __attribute__((always_inline)) {
    if (vid >= total) return;
    const int outer = (int)(vid / L.total_ntiles);
    const int j = (int)(vid - (int64_t)outer * L.total_ntiles);
    int pi = 0;
    while (pi + 1 < L.n && j >= L.p[pi + 1].tile0) ++pi;
    float inv_unused;
    problem_scales(pi, sA, sB, inv_unused);
  };
  auto decode = [&](Cursor& c) __attribute__((always_inline)) {
    c.ok = c.vid < total;
    if (!c.ok) return;
    const int outer = (int)(c.vid / L.total_ntiles);
    int j = (int)(c.vid - (int64_t)outer * L.total_ntiles);
    int pi = 0;
    while (pi + 1 < L.n && j >= L.p[pi + 1].tile0) ++pi;
    j -= L.p[pi].tile0;
    c.pi = pi;
    c.s = 0;
    c.M = L.p[pi].M;
    c.N = L.p[pi].N;
    c.nsrc = L.p[pi].nsrc;
    c.src0 = L.p[pi].src0;
    if (EPI != EPI_SLAB) {
      c.row0 = outer * BM;
      c.col0 = j * BN;
      c.split = 0;
      c.k0 = 0;
      c.kend = L.src[c.src0].Kred;
    } else {
      const int tn = L.p[pi].tiles_n;
      c.split = outer;
      c.row0 = (j / tn) * BM;
      c.col0 = (j % tn) * BN;
      c.k0 = outer * L.chunk;
      const int kr = L.src[c.src0].Kred;
      c.kend = (c.k0 + L.chunk < kr) ? c.k0 + L.chunk : kr;
      if (c.k0 >= c.kend) c.kend = c.k0 + GK;
    }
    c.bias_cols = (EPI == EPI_SLAB) && BCOLS;
    c.want_bias = (EPI == EPI_SLAB) && L.p[pi].bias_slab != nullptr && (c.bias_cols ? c.row0 == 0 : c.col0 == 0);
    c.short_tile = (EPI != EPI_SLAB) && (c.nsrc == 1) && (c.kend - c.k0 < 3 * GK);
    c.counted = c.row0 + BM <= c.M && c.col0 + BN <= c.N && (EPI == EPI_SLAB || L.p[pi].vec_out != 0);
    if (K7 && EPI != EPI_SLAB && L.p[pi].bias_slab != nullptr) c.counted = false;  // (K7 tiles store a second output: drain)
    if constexpr (EMU == 2) problem_scales(pi, c.sA, c.sB, c.inv);
    else c.sA = c.sB = c.inv = 1.f;
  };
  auto last_step = [&](const Cursor& c) __attribute__((always_inline)) {
    return c.k0 + GK >= c.kend && (EPI == EPI_SLAB || c.s + 1 >= c.nsrc);
  };
  // returns true when the cursor moved to another (tile, source): running pointers must be set up again
  auto advance = [&](Cursor& c) __attribute__((always_inline)) -> bool {
    c.k0 += GK;
    if (c.k0 < c.kend) return false;
    if (EPI != EPI_SLAB && c.s + 1 < c.nsrc) {
      ++c.s;
      c.k0 = 0;
      c.kend = L.src[c.src0 + c.s].Kred;
      return true;
    }
    c.vid += gridDim.x;
    decode(c);
    return true;
  };

  // ---- LDS-DMA with running per-lane source pointers ----
  const float* pa[2];
  const float* pb[NI];
  int64_t incA = 0, incB = 0;
  auto setup_ptrs = [&](const Cursor& c) __attribute__((always_inline)) {
    if (!c.ok) return;
    const int si = c.src0 + c.s;
    const float* A = L.src[si].A;
    const float* B = L.src[si].B;
    const int64_t lda = L.src[si].lda, ldb = L.src[si].ldb;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int t = wave + 4 * j;
      if (ARC) {
        const int rl = 16 * t + (lane >> 2);
        int row = c.row0 + rl;
        row = row < c.M ? row : c.M - 1;
        pa[j] = A + (int64_t)row * lda + c.k0 + 4 * ((lane & 3) ^ ((rl >> 2) & 3));
      } else {
        const int kr = t * 2 + lane / 32;  // 2 k-rows of 128 floats per wave-instruction
        int col = c.row0 + 4 * (lane % 32);
        col = (col + 4 <= c.M) ? col : c.M - 4;
        pa[j] = A + (int64_t)(c.k0 + kr) * lda + col;
      }
    }
    incA = ARC ? GK : GK * lda;
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int t = wave + 4 * j;
      if (BRC) {
        const int rl = 16 * t + (lane >> 2);
        int row = c.col0 + rl;
        row = row < c.N ? row : c.N - 1;
        pb[j] = B + (int64_t)row * ldb + c.k0 + 4 * ((lane & 3) ^ ((rl >> 2) & 3));
      } else {
        constexpr int KR = 256 / BN, CH = BN / 4;
        const int kr = t * KR + lane / CH;
        int col = c.col0 + 4 * (lane % CH);
        col = (col + 4 <= c.N) ? col : c.N - 4;
        pb[j] = B + (int64_t)(c.k0 + kr) * ldb + col;
      }
    }
    incB = BRC ? GK : GK * ldb;
  };
  auto issue = [&](const int stage) __attribute__((always_inline)) {
    float* sa = lds + stage * STG;
#pragma unroll
    for (int j = 0; j < 2; ++j)
      dma16(pa[j], sa + (wave + 4 * j) * 256);
#pragma unroll
    for (int j = 0; j < NI; ++j)
      dma16(pb[j], sa + GA + (wave + 4 * j) * 256);
#pragma unroll
    for (int j = 0; j < 2; ++j) pa[j] += incA;
#pragma unroll
    for (int j = 0; j < NI; ++j) pb[j] += incB;
  };

  // fwd: the bias values of this wave's 64 (BN/2) columns travel to LDS by one 4-byte LDS-DMA per lane when the compute
  // cursor enters a tile, so the epilogue reads them with ds_read_b128 instead of eight dependent global loads.  Always
  // exactly one VMEM operation (a dummy address when there is no bias): the counted waits rely on it.
  float* const lds_bias = lds + PSTAGES * STG + wave * 64;
  auto bias_dma = [&](const Cursor& c) __attribute__((always_inline)) {
    if (EPI != EPI_FWD || !c.ok) return;
    const float* bias = L.p[c.pi].bias;
    int col = c.col0 + wn * (BN / 2) + lane;
    col = col < c.N ? col : c.N - 1;
    const float* src = bias ? bias + col : L.p[c.pi].C;
    __builtin_amdgcn_sched_barrier(0);
    dma4(src, lds_bias);
    __builtin_amdgcn_sched_barrier(0);
  };

  f32x16 acc[2][NI];
  auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
  };
  zero_acc();
  // wgrad bias partials: batch sums of the row (or, for [K,N] weights, the column) operand, taken from the raw
  // fragments as they pass through the registers (16 adds per step) instead of re-reading the LDS image
  float bs_cur[2] = {0.f, 0.f}, bs_next[2] = {0.f, 0.f};
  auto sum8 = [](const auto& r) __attribute__((always_inline)) {
    float x[8];
    r.get(x);
    return ((x[0] + x[1]) + (x[2] + x[3])) + ((x[4] + x[5]) + (x[6] + x[7]));
  };

  const uint32_t lds0 = lds_byte_addr(lds);
  uint32_t aA0, aA1, aB0, aB1;
  {
    const int ra = wm * 64 + l31, rb = wn * (BN / 2) + l31;
    if (ARC) {
      aA0 = lds0 + ra * GK * 4 + ((h ^ ((ra >> 2) & 3)) * 16);
      aA1 = lds0 + ra * GK * 4 + (((2 + h) ^ ((ra >> 2) & 3)) * 16);
    } else {
      aA0 = lds0 + (4 * h * BM + ra) * 4;
      aA1 = aA0 + 32 * 4;
    }
    if (BRC) {
      aB0 = lds0 + rb * GK * 4 + ((h ^ ((rb >> 2) & 3)) * 16);
      aB1 = lds0 + rb * GK * 4 + (((2 + h) ^ ((rb >> 2) & 3)) * 16);
    } else {
      aB0 = lds0 + (4 * h * BN + rb) * 4;
      aB1 = aB0 + 32 * 4;
    }
  }

  // ---- epilogue: lane = output row, registers = four runs of 4 consecutive columns.  The activation is dispatched ONCE
  // per tile (a per-element `switch` made this 10 000 instructions of branchy code that cost as much as 14 k-steps).
  // EMU 2: the accumulators hold sum (a sA)(b sB); one exact power-of-two multiplication brings them back
  auto unscale = [&](const float v, const Cursor& c) __attribute__((always_inline)) -> float {
    if constexpr (EMU == 2) return v * c.inv;
    else return v;
  };
  // largest |stored output| of this lane's part of the current tile (fwd / dgrad with amax_out): v_max_f32 with an |x|
  // source modifier, one instruction per element (a NaN does not register; it reaches the consumer as a NaN anyway)
  float am_f = 0.f;
  auto epilogue_slab = [&](const Cursor& c) __attribute__((always_inline)) {
    const int pi = c.pi;
    const int row0 = c.row0, col0 = c.col0, PM = c.M, PN = c.N;
    if (c.want_bias) {
      float* bs = L.p[pi].bias_slab;
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        // lanes l and l + 32 hold the two k-halves of one row's sum
        float v = bs_cur[t];
        v += __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute((lane ^ 32) << 2, __builtin_bit_cast(int, v)));
        if (!BCOLS) {
          const int row = row0 + wm * 64 + t * 32 + l31;
          if (wn == 0 && h == 0 && row < PM) bs[(int64_t)c.split * PM + row] = v;
        } else if (t < NI) {
          const int col = col0 + wn * (BN / 2) + t * 32 + l31;
          if (wm == 0 && h == 0 && col < PN) bs[(int64_t)c.split * PN + col] = v;
        }
      }
    }
    float* slab = L.slab + L.p[pi].slab_off + (int64_t)c.split * PM * PN;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
      const int row = row0 + wm * 64 + mi * 32 + l31;
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int col = col0 + wn * (BN / 2) + ni * 32 + 8 * g + 4 * h;
          if (row < PM && col < PN)
            *reinterpret_cast<float4*>(slab + (int64_t)row * PN + col) =
                make_float4(unscale(acc[mi][ni][4 * g], c), unscale(acc[mi][ni][4 * g + 1], c),
                            unscale(acc[mi][ni][4 * g + 2], c), unscale(acc[mi][ni][4 * g + 3], c));
        }
    }
  };
  auto epilogue_act = [&](const Cursor& c, const uint32_t so_epi, auto actc) __attribute__((always_inline)) {
    constexpr int ACT = decltype(actc)::value;
    const int pi = c.pi;
    const int row0 = c.row0, col0 = c.col0, PM = c.M, PN = c.N;
    float* const C = L.p[pi].C;
    const int64_t ldc = L.p[pi].ldc;
    const float* const bias = L.p[pi].bias;
    const float* const Y = L.p[pi].Y;
    const int64_t ldy = L.p[pi].ldy;
    const bool accumulate = L.p[pi].accumulate != 0;
    const bool vec = L.p[pi].vec_out != 0;
    // relu sign bits (1 bit per output, word [row][col / 32]): written by the forward launch, read by dgrad in place of
    // Y.  The host only passes a mask together with act == RELU.
    uint32_t* const mask = (ACT == MML_ACT_RELU) ? L.p[pi].mask : nullptr;
    const int64_t ldmask = L.p[pi].ldmask;
    const bool use_mask = (EPI == EPI_DGRAD) && mask != nullptr;
    auto xor_lane = [&](uint32_t v, int m) __attribute__((always_inline)) {
      return (uint32_t)__builtin_amdgcn_ds_bpermute((lane ^ m) << 2, (int)v);
    };
    if (vec && BN == 64) {
      // 128 x 64 tiles: a stage is 12 KiB, too small for four 4-KiB transposition areas -> per-lane 16-byte accesses
      // (lane = row), all loads of a 32-row half before its stores
      if (EPI == EPI_FWD && c.short_tile) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      f32x4_t b4[NI][4];
      if (EPI == EPI_FWD && bias) {
        const uint32_t ab = lds_byte_addr(lds_bias) + 16 * h;
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
          for (int g = 0; g < 4; ++g) b4[ni][g] = ds_read128<0>(ab + (ni * 32 + 8 * g) * 4);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
          for (int g = 0; g < 4; ++g) lds_landed(b4[ni][g]);
        __builtin_amdgcn_sched_barrier(0);
      }
      // K7: a second output (see Problem)
      float* const C2 = K7 ? L.p[pi].bias_slab : nullptr;
      const int64_t ldc2 = L.p[pi].slab_off;
      const bool mulf = K7 && (EPI == EPI_FWD) && C2 != nullptr;   // prod = C * mul
      const bool gate = K7 && (EPI == EPI_DGRAD) && C2 != nullptr; // dH, dG instead of dA
      const float* const Gm = gate ? L.p[pi].bias : nullptr;
      const int act_g = L.p[pi].tiles_m;
      const bool acc2 = L.p[pi].bias_cols != 0;
      float am2 = 0.f;
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) {
        const int row = row0 + wm * 64 + mi * 32 + l31;
        const bool row_ok = row < PM;
        float4 y4[NI][4];
        float4 g4[K7 ? NI : 1][4];
        uint32_t mw[NI];  // this row's mask word per 32-column group: read (dgrad) or built (fwd)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) mw[ni] = 0u;
        if (use_mask) {
#pragma unroll
          for (int ni = 0; ni < NI; ++ni) {
            const int cg = col0 + wn * (BN / 2) + ni * 32;
            if (row_ok && cg < PN) mw[ni] = mask[(int64_t)row * ldmask + (cg >> 5)];
          }
        } else if ((EPI == EPI_DGRAD && ACT != MML_ACT_NONE) || mulf || gate) {
#pragma unroll
          for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              const int col = col0 + wn * (BN / 2) + ni * 32 + 8 * g + 4 * h;
              y4[ni][g] = (row_ok && col < PN) ? *reinterpret_cast<const float4*>(Y + (int64_t)row * ldy + col)
                                               : make_float4(0, 0, 0, 0);
            }
        }
        if (gate) {
#pragma unroll
          for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              const int col = col0 + wn * (BN / 2) + ni * 32 + 8 * g + 4 * h;
              g4[K7 ? ni : 0][g] = (row_ok && col < PN) ? *reinterpret_cast<const float4*>(Gm + (int64_t)row * ldmask + col)
                                                        : make_float4(0, 0, 0, 0);
            }
        }
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int col = col0 + wn * (BN / 2) + ni * 32 + 8 * g + 4 * h;
            if (!row_ok || col >= PN) continue;
            float4 v = make_float4(unscale(acc[mi][ni][4 * g], c), unscale(acc[mi][ni][4 * g + 1], c),
                                   unscale(acc[mi][ni][4 * g + 2], c), unscale(acc[mi][ni][4 * g + 3], c));
            float* dst = C + (int64_t)row * ldc + col;
            if (EPI == EPI_FWD) {
              if (bias) {
                v.x += b4[ni][g].x; v.y += b4[ni][g].y; v.z += b4[ni][g].z; v.w += b4[ni][g].w;
              }
              v.x = act_fwd_t<ACT>(v.x); v.y = act_fwd_t<ACT>(v.y); v.z = act_fwd_t<ACT>(v.z); v.w = act_fwd_t<ACT>(v.w);
              if (mask) {
                const uint32_t nib = (v.x > 0.f ? 1u : 0u) | (v.y > 0.f ? 2u : 0u) | (v.z > 0.f ? 4u : 0u) | (v.w > 0.f ? 8u : 0u);
                mw[ni] |= nib << (8 * g + 4 * h);
              }
            } else {
              const float4 raw = v;  // (gate mode: the input gradient of h (.) g before h's derivative)
              if (use_mask) {
                const uint32_t nib = mw[ni] >> (8 * g + 4 * h);
                if (!(nib & 1u)) v.x = 0.f;
                if (!(nib & 2u)) v.y = 0.f;
                if (!(nib & 4u)) v.z = 0.f;
                if (!(nib & 8u)) v.w = 0.f;
              } else if (ACT != MML_ACT_NONE) {
                v.x *= act_bwd_t<ACT>(y4[ni][g].x); v.y *= act_bwd_t<ACT>(y4[ni][g].y);
                v.z *= act_bwd_t<ACT>(y4[ni][g].z); v.w *= act_bwd_t<ACT>(y4[ni][g].w);
              }
              if (gate) {  // v = raw act_h'(h) so far: dH = v g; dG = raw h act_g'(g)
                const float4 gq = g4[K7 ? ni : 0][g], hq = y4[ni][g];
                float4 dg = make_float4(raw.x * hq.x, raw.y * hq.y, raw.z * hq.z, raw.w * hq.w);
                if (act_g != MML_ACT_NONE) {
                  dg.x *= act_bwd(gq.x, act_g); dg.y *= act_bwd(gq.y, act_g);
                  dg.z *= act_bwd(gq.z, act_g); dg.w *= act_bwd(gq.w, act_g);
                }
                v.x *= gq.x; v.y *= gq.y; v.z *= gq.z; v.w *= gq.w;
                float* const d2 = C2 + (int64_t)row * ldc2 + col;
                if (acc2) {
                  const float4 o = *reinterpret_cast<const float4*>(d2);
                  dg.x += o.x; dg.y += o.y; dg.z += o.z; dg.w += o.w;
                }
                *reinterpret_cast<float4*>(d2) = dg;
                am2 = fmaxf(fmaxf(am2, fabsf(dg.x)), fmaxf(fabsf(dg.y), fmaxf(fabsf(dg.z), fabsf(dg.w))));
              }
              if (accumulate) {
                const float4 o = *reinterpret_cast<const float4*>(dst);
                v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
              }
            }
            *reinterpret_cast<float4*>(dst) = v;
            am_f = fmaxf(fmaxf(am_f, fabsf(v.x)), fmaxf(fabsf(v.y), fmaxf(fabsf(v.z), fabsf(v.w))));
            if (mulf) {  // prod = C * mul
              const float4 mq = y4[ni][g];
              const float4 pr = make_float4(v.x * mq.x, v.y * mq.y, v.z * mq.z, v.w * mq.w);
              *reinterpret_cast<float4*>(C2 + (int64_t)row * ldc2 + col) = pr;
              am2 = fmaxf(fmaxf(am2, fabsf(pr.x)), fmaxf(fabsf(pr.y), fmaxf(fabsf(pr.z), fabsf(pr.w))));
            }
          }
        if (EPI == EPI_FWD && mask) {  // the two half-waves hold the odd / even 4-column groups of the same rows
#pragma unroll
          for (int ni = 0; ni < NI; ++ni) {
            const uint32_t w = mw[ni] | xor_lane(mw[ni], 32);
            const int cg = col0 + wn * (BN / 2) + ni * 32;
            if (h == 0 && row_ok && cg < PN) mask[(int64_t)row * ldmask + (cg >> 5)] = w;
          }
        }
      }
      if ((mulf || gate) && L.p[pi].amax_out2) {  // (like am_f: the workgroup's word of the problem, second set)
        const uint32_t a2 = lds0 + (uint32_t)(PSTAGES * STG + 256 + MML_MAX_GROUP + pi) * 4u;
        const uint32_t am2_bits = __float_as_uint(am2);
        asm volatile("ds_max_u32 %0, %1" ::"v"(a2), "v"(am2_bits) : "memory");
      }
    } else if (vec) {
      // Each 32 x 32 sub-tile is turned row-major through this wave's 4 KiB of the stage buffer the step just consumed
      // (idle until the next step's DMA, which every wave issues after the next barrier): a lane then moves 16 bytes
      // of a row and a wave-instruction covers 8 rows x 128 B -- whole cache lines for the stores and for dgrad's
      // reads of Y, instead of 32 rows x 32 B.  16-byte chunk c of row r sits at chunk c ^ (r & 7): conflict-free for
      // the column-wise writes and the row-wise reads.
      const uint32_t tb = lds0 + so_epi + wave * 4096;
      const int R = lane >> 3, cc = lane & 7;
      if (EPI == EPI_FWD && c.short_tile) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // bias DMA landed?
      float4 y4[4];  // dgrad: Y values of the sub-tile being written (row-major, like the stores)
      auto load_y = [&](const int sidx, float4 (&y)[4]) __attribute__((always_inline)) {
        const int mi = sidx / NI, ni = sidx % NI;
        const int colg = col0 + wn * (BN / 2) + ni * 32 + 4 * cc;
        const int rowb = row0 + wm * 64 + mi * 32 + R;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
          const bool ok = rowb + 8 * p < PM && colg < PN;
          y[p] = ok ? *reinterpret_cast<const float4*>(Y + (int64_t)(rowb + 8 * p) * ldy + colg) : make_float4(0, 0, 0, 0);
        }
      };
      // K7: a second output (see Problem)
      float* const C2 = (K7 && EPI != EPI_SLAB) ? L.p[pi].bias_slab : nullptr;
      const int64_t ldc2 = L.p[pi].slab_off;
      const bool mulf = K7 && (EPI == EPI_FWD) && C2 != nullptr;   // prod = C * mul
      const bool gate = K7 && (EPI == EPI_DGRAD) && C2 != nullptr; // dH, dG instead of dA
      const float* const Gm = gate ? L.p[pi].bias : nullptr;
      const int64_t ldg = ldmask;
      const int act_g = L.p[pi].tiles_m;
      const bool acc2 = L.p[pi].bias_cols != 0;
      float am2 = 0.f;
      const bool USE_Y = ((EPI == EPI_DGRAD) && (ACT != MML_ACT_NONE) && !use_mask) || gate;
      float4 g4[4];
#pragma unroll
      for (int sidx = 0; sidx < 2 * NI; ++sidx) {
        const int mi = sidx / NI, ni = sidx % NI;
        {
          if (USE_Y) load_y(sidx, y4);  // in flight during the LDS round trip
          if (gate) {
            const int colg_ = col0 + wn * (BN / 2) + ni * 32 + 4 * cc, rowb_ = row0 + wm * 64 + mi * 32 + R;
#pragma unroll
            for (int p = 0; p < 4; ++p)
              g4[p] = (rowb_ + 8 * p < PM && colg_ < PN)
                          ? *reinterpret_cast<const float4*>(Gm + (int64_t)(rowb_ + 8 * p) * ldg + colg_)
                          : make_float4(0, 0, 0, 0);
          }
          const int colg = col0 + wn * (BN / 2) + ni * 32 + 4 * cc;  // this lane's 4 columns
          const int rowb = row0 + wm * 64 + mi * 32 + R;             // ... of rows rowb + 8p
          uint32_t mw[4];  // mask word of row rowb + 8p (all eight lanes of a row read / build the same word)
          if (use_mask) {
#pragma unroll
            for (int p = 0; p < 4; ++p)
              mw[p] = (rowb + 8 * p < PM && colg < PN) ? mask[(int64_t)(rowb + 8 * p) * ldmask + (colg >> 5)] : 0u;
          }
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const f32x4_t v = {unscale(acc[mi][ni][4 * g], c), unscale(acc[mi][ni][4 * g + 1], c),
                               unscale(acc[mi][ni][4 * g + 2], c), unscale(acc[mi][ni][4 * g + 3], c)};
            ds_write128(tb + l31 * 128 + (((2 * g + h) ^ (l31 & 7)) * 16), v);
          }
          f32x4_t b4 = {0.f, 0.f, 0.f, 0.f};
          if (EPI == EPI_FWD && bias) b4 = ds_read128<0>(lds_byte_addr(lds_bias) + (ni * 32 + 4 * cc) * 4);
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          if (EPI == EPI_FWD && bias) lds_landed(b4);
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
            if (gate) {
              // (gate mode: both gradients are formed from the raw input gradient in the store loop below)
            } else if (EPI == EPI_FWD) {
              x.x = act_fwd_t<ACT>(x.x + b4.x); x.y = act_fwd_t<ACT>(x.y + b4.y);
              x.z = act_fwd_t<ACT>(x.z + b4.z); x.w = act_fwd_t<ACT>(x.w + b4.w);
            } else if (use_mask) {
              const uint32_t nib = mw[p] >> (4 * cc);
              if (!(nib & 1u)) x.x = 0.f;
              if (!(nib & 2u)) x.y = 0.f;
              if (!(nib & 4u)) x.z = 0.f;
              if (!(nib & 8u)) x.w = 0.f;
            } else if (ACT != MML_ACT_NONE) {
              x.x *= act_bwd_t<ACT>(y4[p].x); x.y *= act_bwd_t<ACT>(y4[p].y);
              x.z *= act_bwd_t<ACT>(y4[p].z); x.w *= act_bwd_t<ACT>(y4[p].w);
            }
          }
          if (EPI == EPI_FWD && mask) {  // eight lanes (cc = 0..7) hold the eight nibbles of a row's word
            uint32_t wsel = 0u;
#pragma unroll
            for (int p = 0; p < 4; ++p) {
              const f32x4_t& x = v[p];
              uint32_t w = ((x.x > 0.f ? 1u : 0u) | (x.y > 0.f ? 2u : 0u) | (x.z > 0.f ? 4u : 0u) | (x.w > 0.f ? 8u : 0u))
                           << (4 * cc);
              // OR the eight nibbles into lane cc == 0 with three DPP row shifts (lane i takes lane i + 4, + 2, + 1:
              // lanes 0..3 of every group of 8 only ever read inside their group; the other lanes' results are unused)
              w |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)w, 0x104, 0xf, 0xf, true);
              w |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)w, 0x102, 0xf, 0xf, true);
              w |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)w, 0x101, 0xf, 0xf, true);
              // ... and hand the word of row R + 8p to lane cc == p (row_shr:p), so that ONE store instruction writes the
              // four words of this lane group
              uint32_t moved = w;
              if (p == 1) moved = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)w, 0x111, 0xf, 0xf, true);
              if (p == 2) moved = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)w, 0x112, 0xf, 0xf, true);
              if (p == 3) moved = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)w, 0x113, 0xf, 0xf, true);
              if (cc == p) wsel = moved;
            }
            const int mrow = rowb + 8 * cc;  // lane cc < 4 owns row R + 8 cc
            const int cg = col0 + wn * (BN / 2) + ni * 32;
            if (cc < 4 && mrow < PM && cg < PN) mask[(int64_t)mrow * ldmask + (cg >> 5)] = wsel;
          }
#pragma unroll
          for (int p = 0; p < 4; ++p) {
            const int row = rowb + 8 * p;
            if (row >= PM || colg >= PN) continue;
            float4 x = make_float4(v[p].x, v[p].y, v[p].z, v[p].w);
            if (EPI == EPI_DGRAD && accumulate && !gate) {  // (rare: outputs summed over more than MML_MAX_SRC sources)
              const float4 o = *reinterpret_cast<const float4*>(C + (int64_t)row * ldc + colg);
              x.x += o.x; x.y += o.y; x.z += o.z; x.w += o.w;
            }
            if (gate) {  // x = the raw input gradient v of h (.) g: dG = v h act_g'(g), dH = v g act_h'(h)
              const float4 gq = g4[p], hq = y4[p];
              float4 dg = make_float4(x.x * hq.x, x.y * hq.y, x.z * hq.z, x.w * hq.w);
              if (act_g != MML_ACT_NONE) {
                dg.x *= act_bwd(gq.x, act_g); dg.y *= act_bwd(gq.y, act_g);
                dg.z *= act_bwd(gq.z, act_g); dg.w *= act_bwd(gq.w, act_g);
              }
              x.x *= gq.x * act_bwd_t<ACT>(hq.x); x.y *= gq.y * act_bwd_t<ACT>(hq.y);
              x.z *= gq.z * act_bwd_t<ACT>(hq.z); x.w *= gq.w * act_bwd_t<ACT>(hq.w);
              if (accumulate) {
                const float4 o = *reinterpret_cast<const float4*>(C + (int64_t)row * ldc + colg);
                x.x += o.x; x.y += o.y; x.z += o.z; x.w += o.w;
              }
              float* const d2 = C2 + (int64_t)row * ldc2 + colg;
              if (acc2) {
                const float4 o = *reinterpret_cast<const float4*>(d2);
                dg.x += o.x; dg.y += o.y; dg.z += o.z; dg.w += o.w;
              }
              *reinterpret_cast<float4*>(d2) = dg;
              am2 = fmaxf(fmaxf(am2, fabsf(dg.x)), fmaxf(fabsf(dg.y), fmaxf(fabsf(dg.z), fabsf(dg.w))));
            }
            // (nontemporal, MMLREC_BUILD_NTSTORE=1 at build time: the output is read next by another kernel, long after the
            // line has left the L2 -- the panel kernel gained 4 % from it)
#ifdef MML_GEMM_NTSTORE
            __builtin_nontemporal_store(f32x4_t{x.x, x.y, x.z, x.w}, reinterpret_cast<f32x4_t*>(C + (int64_t)row * ldc + colg));
#else
            *reinterpret_cast<float4*>(C + (int64_t)row * ldc + colg) = x;
#endif
            am_f = fmaxf(fmaxf(am_f, fabsf(x.x)), fmaxf(fabsf(x.y), fmaxf(fabsf(x.z), fabsf(x.w))));
            if (mulf) {  // prod = C * mul (mul is read here, piece by piece: its registers are not held across the turn)
              const float4 mq = *reinterpret_cast<const float4*>(Y + (int64_t)row * ldy + colg);
              const float4 pr = make_float4(x.x * mq.x, x.y * mq.y, x.z * mq.z, x.w * mq.w);
              *reinterpret_cast<float4*>(C2 + (int64_t)row * ldc2 + colg) = pr;
              am2 = fmaxf(fmaxf(am2, fabsf(pr.x)), fmaxf(fabsf(pr.y), fmaxf(fabsf(pr.z), fabsf(pr.w))));
            }
          }
        }
      }
      if ((mulf || gate) && L.p[pi].amax_out2) {  // (like am_f below: the workgroup's word of the problem, second set)
        const uint32_t a2 = lds0 + (uint32_t)(PSTAGES * STG + 256 + MML_MAX_GROUP + pi) * 4u;
        const uint32_t am2_bits = __float_as_uint(am2);
        asm volatile("ds_max_u32 %0, %1" ::"v"(a2), "v"(am2_bits) : "memory");
      }
    } else {  // unaligned / odd-width outputs: element-wise
#pragma unroll  // (full unroll: run-time indices would put the accumulators in scratch)
      for (int mi = 0; mi < 2; ++mi) {
        const int row = row0 + wm * 64 + mi * 32 + l31;
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
          const int cg = col0 + wn * (BN / 2) + ni * 32;
          uint32_t mw = 0u;
          if (use_mask && row < PM && cg < PN) mw = mask[(int64_t)row * ldmask + (cg >> 5)];
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int bit = 8 * (r >> 2) + 4 * h + (r & 3);
            const int col = cg + bit;
            if (row >= PM || col >= PN) continue;
            float x = unscale(acc[mi][ni][r], c);
            float* dst = C + (int64_t)row * ldc + col;
            if (EPI == EPI_FWD) {
              x = act_fwd_t<ACT>(x + (bias ? bias[col] : 0.f));
              if (mask && x > 0.f) mw |= 1u << bit;
            } else {
              if (use_mask) {
                if (!((mw >> bit) & 1u)) x = 0.f;
              } else if (ACT != MML_ACT_NONE) {
                x *= act_bwd_t<ACT>(Y[(int64_t)row * ldy + col]);
              }
              if (accumulate) x += *dst;
            }
            *dst = x;
            am_f = fmaxf(am_f, fabsf(x));
          }
          if (EPI == EPI_FWD && mask) {
            const uint32_t w = mw | xor_lane(mw, 32);
            if (h == 0 && row < PM && cg < PN) mask[(int64_t)row * ldmask + (cg >> 5)] = w;
          }
        }
      }
    }
  };
  auto epilogue = [&](const Cursor& c, const uint32_t so_epi) __attribute__((always_inline)) {
    if (EPI == EPI_SLAB) {
      epilogue_slab(c);
      return;
    }
    switch (L.p[c.pi].act) {
      case MML_ACT_RELU: epilogue_act(c, so_epi, std::integral_constant<int, MML_ACT_RELU>{}); break;
      case MML_ACT_SIGMOID: epilogue_act(c, so_epi, std::integral_constant<int, MML_ACT_SIGMOID>{}); break;
      case MML_ACT_SIGMOID2: epilogue_act(c, so_epi, std::integral_constant<int, MML_ACT_SIGMOID2>{}); break;
      default: epilogue_act(c, so_epi, std::integral_constant<int, MML_ACT_NONE>{}); break;
    }
    // The output's magnitude for the GEMMs that will read it: wave maximum -> the workgroup's word of this problem in
    // LDS (ds_max_u32, asm like every LDS access of this kernel); ONE global atomic per workgroup and problem follows at
    // the end of the kernel.  (A global atomic per tile stalled the pipeline: the counted waits make every younger
    // load wait for it -- +0.1 ms per launch at M = 65 536.)
    // (every lane issues the LDS atomic on the same word: ~2 cycles per lane once per tile, and no registers for a wave
    // reduction -- the 128 x 64 input-gradient kernels sit at their 168-VGPR limit)
    if (L.p[c.pi].amax_out) {
      const uint32_t a = lds0 + (uint32_t)(PSTAGES * STG + 256 + c.pi) * 4u;
      const uint32_t am_bits = __float_as_uint(am_f);
      asm volatile("ds_max_u32 %0, %1" ::"v"(a), "v"(am_bits) : "memory");
    }
    am_f = 0.f;
  };

  // ---- fragment reads of one stage: the stage's byte offset is a run-time VGPR add (the reads are inline asm, so the
  // compiler's waitcnt pass never sees them next to the in-flight LDS-DMA) ----
  auto read_a = [&](const uint32_t so, auto mic, RawFrag<ARC>& r) __attribute__((always_inline)) {
    constexpr int MI = decltype(mic)::value;
    if constexpr (ARC) read_frag_rc<MI * 32 * GK * 4>(aA0 + so, aA1 + so, r);
    else read_frag_nrc<BM, 0>((MI == 0 ? aA0 : aA1) + so, r);
  };
  auto read_b = [&](const uint32_t so, auto nic, RawFrag<BRC>& r) __attribute__((always_inline)) {
    constexpr int NIX = decltype(nic)::value;
    constexpr int SB = GA * 4;
    if constexpr (BRC) read_frag_rc<SB + NIX * 32 * GK * 4>(aB0 + so, aB1 + so, r);
    else read_frag_nrc<BN, SB>((NIX == 0 ? aB0 : aB1) + so, r);
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;

  // ---- pipeline state: two register sets, indexed by step parity (the loop body is two steps) ----
  Prep<EMU> PA0[2], PB0[2];
  RawFrag<ARC> RA1[2];
  RawFrag<BRC> RB1[2];

  if (EPI != EPI_SLAB && tid < 2 * MML_MAX_GROUP) {  // (ordered before the first epilogue by the barriers of the k-steps)
    const uint32_t a = lds0 + (uint32_t)(PSTAGES * STG + 256 + tid) * 4u;
    const uint32_t z = 0u;
    asm volatile("ds_write_b32 %0, %1" ::"v"(a), "v"(z) : "memory");
  }
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
  // stage 0 -> registers
  if (issued >= 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * LOADS) : "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  {
    RawFrag<ARC> a0;
    RawFrag<BRC> b0;
    read_a(0u, I0{}, a0);
    read_a(0u, I1{}, RA1[0]);
    read_b(0u, I0{}, b0);
    if (NI == 2) read_b(0u, I1{}, RB1[0]);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    a0.landed();
    RA1[0].landed();
    b0.landed();
    if (NI == 2) RB1[0].landed();
    __builtin_amdgcn_sched_barrier(0);
    if (EPI == EPI_SLAB) {
      if (!BCOLS) {
        bs_cur[0] = sum8(a0);
        bs_cur[1] = sum8(RA1[0]);
      } else {
        bs_cur[0] = sum8(b0);
        if (NI == 2) bs_cur[1] = sum8(RB1[0]);
      }
    }
    prep_frag<EMU>(a0, PA0[0], cur.sA);
    if constexpr (BPL) planes_of(b0, PB0[0]);
    else prep_frag_b<EMU>(b0, PB0[0], cur.sB);
  }

  auto step = [&](auto par_c) __attribute__((always_inline)) {
    constexpr int P = decltype(par_c)::value, Q = P ^ 1;
    // The operands of step i + 1 are prepared in the shadow of this step's MFMAs; at the end of a tile they belong to
    // the NEXT tile (possibly another problem, with other scales), so the cursor is advanced HERE and installed below.
    const bool tile_end = last_step(cur);
    float nsA = cur.sA, nsB = cur.sB;
    if constexpr (EMU == 2) {
      if (__builtin_expect(tile_end, 0)) scales_of_vid(cur.vid + gridDim.x, nsA, nsB);
    }
    const int sidx = i & (PSTAGES - 1);
    const uint32_t so_cur = (uint32_t)sidx * (STG * 4);
    const uint32_t so_next = (uint32_t)((sidx + 1) & (PSTAGES - 1)) * (STG * 4);
    // Stage i+1 (read below) was issued two steps ago: in steady state at most the loads of step i+2 are younger.
    // Off the hot path: for two steps after a counted epilogue its stores are younger too; once the prefetch cursor
    // has run out of work, drain.
    if (__builtin_expect((epi_left | (pf.ok ? 0 : 1)) == 0, 1)) {
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LOADS) : "memory");
    } else if (pf.ok && epi_left > 0) {
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(AFTER_EPI) : "memory");
      --epi_left;
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      epi_left = 0;
    }
    __builtin_amdgcn_s_barrier();
    if (pf.ok) {
      issue((sidx + 3) & (PSTAGES - 1));
      if (advance(pf)) setup_ptrs(pf);
      ++issued;
    }
    // next step's fragments -> the other register set (in flight during blocks 1-2)
    RawFrag<ARC> na0;
    RawFrag<BRC> nb0;
    read_b(so_next, I0{}, nb0);
    read_a(so_next, I0{}, na0);
    read_a(so_next, I1{}, RA1[Q]);
    if (NI == 2) read_b(so_next, I1{}, RB1[Q]);
    __builtin_amdgcn_sched_barrier(0);
    Prep<EMU> PA1, PB1;
    if (NI == 2) {
      if constexpr (EMU == 2 && BPL) {
        planes_of(RB1[P], PB1);
        F16Cut c1;
        mma_cut_first(acc[0][0], PA0[P], PB0[P], RA1[P], cur.sA, c1);
        mma_cut_second(acc[0][NI - 1], PA0[P], PB1, RA1[P], cur.sA, c1, PA1);
      } else if constexpr (EMU == 2) {
        mma_prep_f16(acc[0][0], PA0[P], PB0[P], RB1[P], cur.sB, PB1);
        mma_prep_f16(acc[0][NI - 1], PA0[P], PB1, RA1[P], cur.sA, PA1);
      } else {
        prep_frag_b<EMU>(RB1[P], PB1, cur.sB);
        mma_block<EMU>(acc[0][0], PA0[P], PB0[P]);
        interleave_hint<NMFMA, NVALU>();
        __builtin_amdgcn_sched_barrier(0);
        prep_frag<EMU>(RA1[P], PA1, cur.sA);
        mma_block<EMU>(acc[0][NI - 1], PA0[P], PB1);
        interleave_hint<NMFMA, NVALU>();
        __builtin_amdgcn_sched_barrier(0);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      nb0.landed();
      na0.landed();
      RA1[Q].landed();
      RB1[Q].landed();
      __builtin_amdgcn_sched_barrier(0);
      if (EPI == EPI_SLAB) {
        if (!BCOLS) {
          bs_next[0] = sum8(na0);
          bs_next[1] = sum8(RA1[Q]);
        } else {
          bs_next[0] = sum8(nb0);
          bs_next[1] = sum8(RB1[Q]);
        }
      }
      if constexpr (EMU == 2 && BPL) {
        planes_of(nb0, PB0[Q]);
        F16Cut c2;
        mma_cut_first(acc[1][0], PA1, PB0[P], na0, nsA, c2);
        mma_cut_second(acc[1][NI - 1], PA1, PB1, na0, nsA, c2, PA0[Q]);
      } else if constexpr (EMU == 2) {
        mma_prep_f16(acc[1][0], PA1, PB0[P], nb0, nsB, PB0[Q]);
        mma_prep_f16(acc[1][NI - 1], PA1, PB1, na0, nsA, PA0[Q]);
      } else {
        prep_frag_b<EMU>(nb0, PB0[Q], nsB);
        mma_block<EMU>(acc[1][0], PA1, PB0[P]);
        pin_prep<EMU>(PB0[Q]);
        interleave_hint<NMFMA, NVALU>();
        __builtin_amdgcn_sched_barrier(0);
        prep_frag<EMU>(na0, PA0[Q], nsA);
        mma_block<EMU>(acc[1][NI - 1], PA1, PB1);
        pin_prep<EMU>(PA0[Q]);
        interleave_hint<NMFMA, NVALU>();
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
      prep_frag<EMU>(RA1[P], PA1, cur.sA);
      mma_block<EMU>(acc[0][0], PA0[P], PB0[P]);
      interleave_hint<NMFMA, NVALU>();
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      nb0.landed();
      na0.landed();
      RA1[Q].landed();
      __builtin_amdgcn_sched_barrier(0);
      if (EPI == EPI_SLAB) {
        if (!BCOLS) {
          bs_next[0] = sum8(na0);
          bs_next[1] = sum8(RA1[Q]);
        } else {
          bs_next[0] = sum8(nb0);
        }
      }
      if constexpr (BPL) planes_of(nb0, PB0[Q]);
      else prep_frag_b<EMU>(nb0, PB0[Q], nsB);
      prep_frag<EMU>(na0, PA0[Q], nsA);
      mma_block<EMU>(acc[1][0], PA1, PB0[P]);
      pin_prep<EMU>(PB0[Q]);
      pin_prep<EMU>(PA0[Q]);
      interleave_hint<NMFMA, 2 * NVALU>();
      __builtin_amdgcn_sched_barrier(0);
    }
    if (__builtin_expect(tile_end, 0)) {
      epilogue(cur, so_cur);
      // every load of the epilogue (bias, Y, accumulate target) was consumed before its store issued, so a counted
      // tile leaves exactly NSTORE (or more: bias partials) stores in flight and nothing else
      epi_left = cur.counted ? 2 : 0;
      if (!cur.counted) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      zero_acc();
      bs_cur[0] = 0.f;
      bs_cur[1] = 0.f;
    }
    if (EPI == EPI_SLAB) {  // the fragments read in this step belong to the next one
      bs_cur[0] += bs_next[0];
      bs_cur[1] += bs_next[1];
    }
    advance(cur);
    if (__builtin_expect(tile_end, 0)) bias_dma(cur);
    ++i;
  };

  while (true) {
    if (!cur.ok) break;
    step(I0{});
    if (!cur.ok) break;
    step(I1{});
  }
  if (EPI != EPI_SLAB) {  // the workgroup's magnitudes -> the slots (every wave left the loop at the same step)
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (tid < L.n) {
      uint32_t* const amo = L.p[tid].amax_out;
      if (amo) {
        const uint32_t a = lds0 + (uint32_t)(PSTAGES * STG + 256 + tid) * 4u;
        uint32_t v;
        asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
        if (v) atomicMax(amo + (blockIdx.x & (MML_AMAX_WORDS - 1)), v);
      }
      uint32_t* const amo2 = K7 ? L.p[tid].amax_out2 : nullptr;  // (K7: the second output's magnitude)
      if (amo2) {
        const uint32_t a = lds0 + (uint32_t)(PSTAGES * STG + 256 + MML_MAX_GROUP + tid) * 4u;
        uint32_t v;
        asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
        if (v) atomicMax(amo2 + (blockIdx.x & (MML_AMAX_WORDS - 1)), v);
      }
    }
  }
}

// host-side eligibility of a whole launch for the direct-to-LDS path
static bool pipe_ok(const Launch& L, bool arc, bool brc, int bn, int epi) {
  static int enabled = -1;
  if (enabled < 0) {
    const char* e = getenv("MMLREC_GEMM_GLDS");
    enabled = (e && e[0] == '0') ? 0 : 1;
  }
  if (!enabled) return false;
  for (int i = 0; i < L.n; ++i) {
    const Problem& P = L.p[i];
    if (!arc && (P.M % 4 != 0 || P.M < 4)) return false;
    if (!brc && (P.N % 4 != 0 || P.N < 4)) return false;
    if (P.M < 1 || P.N < 1) return false;
    for (int s = 0; s < P.nsrc; ++s) {
      const Source& S = L.src[P.src0 + s];
      if (S.Kred % GK != 0 || S.Kred < GK || !S.vecA || !S.vecB) return false;
    }
  }
  if (epi == EPI_SLAB && L.chunk % GK != 0) return false;
  return true;
}

static inline int32_t vec_ok(const float* p, int64_t ld) { return (aligned16(p) && (ld % 4 == 0)) ? 1 : 0; }

// 0 = fp32 MFMA (v_mfma_f32_32x32x2_f32) on every launch
// 1 = bf16 operands (round-to-nearest-even in registers), fp32 accumulation: ONE v_mfma_f32_32x32x16_bf16 per 16-k
//     block.  Opt-in reduced precision (~3e-3 relative per product; BASELINE's KuaiRec configuration names bf16).
// 3 = three bf16 planes per operand (six v_mfma_f32_32x32x16_bf16 per 16-k block) on every LDS-DMA launch
// 4 = auto (default): the three-plane form where it is faster, the fp32 MFMA elsewhere.  Both give fp32-equivalent
//     results (max-norm error vs float64 4.7e-7 vs 4.3e-7, tools/bench_gemm.py).
static int g_gemm_mode = -1;
static int g_wgrad_pad = -1;  // unused dynamic LDS requested by wgrad launches (caps their residency, see launch_tiles)
static int gemm_mode() {
  if (g_gemm_mode < 0) {
    const char* e = getenv("MMLREC_GEMM_MODE");
    g_gemm_mode = (e && (e[0] == '0' || e[0] == '1' || e[0] == '2' || e[0] == '3')) ? e[0] - '0' : 4;
  }
  return g_gemm_mode;
}

struct TileChoice {
  int bn, emu;
};

static thread_local char g_last_kernel[96] = "";
static void note_kernel(const char* fam, bool arc, bool brc, int bn, int epi, int mode, int bcols = -1,
                        bool bpl = false) {
  if (bcols < 0)
    snprintf(g_last_kernel, sizeof(g_last_kernel), "%s<%s, %s, %d, %d>", fam, arc ? "true" : "false",
             brc ? "true" : "false", bn, epi);
  else  // (the symbol as a profiler prints it: every template argument, the defaulted ones too)
    snprintf(g_last_kernel, sizeof(g_last_kernel), "%s<%s, %s, %d, %d, %d, %s, %s>", fam, arc ? "true" : "false",
             brc ? "true" : "false", bn, epi, mode, bcols ? "true" : "false", bpl ? "true" : "false");
}

// planes / kexps (or null): per source of the launch, the pre-cut image of its column operand and the exponent slot
// (mml_gemm_planes_cut); used when EVERY source has them and the launch runs the two-plane arithmetic on the LDS-DMA path
template <int EPI>
static int launch_tiles(const Launch& Lin, bool arc, bool brc, TileChoice tc, int64_t nblocks, hipStream_t st,
                        const char* who, const uint32_t* const* planes = nullptr, const int32_t* const* kexps = nullptr,
                        bool k7 = false) {
  Launch Lcopy;
  const Launch* Lp = &Lin;
  const int bn = tc.bn;
  if (nblocks <= 0) return MML_OK;
  if (nblocks > 0x7fffffff) {
    set_error("%s: grid too large", who);
    return MML_ERR_ARG;
  }
  dim3 g((unsigned)nblocks), b(256);
  int emu = tc.emu;  // 0 = fp32 MFMA, 1 = bf16 operands, 2 = two scaled fp16 planes, 3 = three bf16 planes
  if (emu == 3 && gemm_mode() != 3) {  // every operand of the launch comes with its magnitude: two fp16 planes
    bool all = true;
    for (int i = 0; i < Lin.n && all; ++i)
      for (int s = 0; s < Lin.p[i].nsrc && all; ++s)
        all = Lin.src[Lin.p[i].src0 + s].amaxA != nullptr && Lin.src[Lin.p[i].src0 + s].amaxB != nullptr;
    if (all) emu = 2;
  }
  bool bpl = false;
  if (EPI != EPI_SLAB && emu == 2 && planes && kexps && pipe_ok(Lin, arc, brc, bn, EPI)) {
    static int use = -1;
    if (use < 0) {
      const char* e = getenv("MMLREC_GEMM_PLANES");  // 0: ignore pre-cut weights (A/B switch)
      use = e ? atoi(e) : 1;
    }
    bpl = use != 0;
    for (int i = 0; i < Lin.n && bpl; ++i)
      for (int s = 0; s < Lin.p[i].nsrc && bpl; ++s) {
        const int si = Lin.p[i].src0 + s;
        // (one exponent per problem: the kernel reads the first source's)
        bpl = planes[si] && kexps[si] && aligned16(planes[si]) && kexps[si] == kexps[Lin.p[i].src0];
      }
    if (bpl) {
      Lcopy = Lin;
      for (int i = 0; i < Lin.n; ++i)
        for (int s = 0; s < Lin.p[i].nsrc; ++s) {
          const int si = Lin.p[i].src0 + s;
          Lcopy.src[si].B = reinterpret_cast<const float*>(planes[si]);
          Lcopy.src[si].amaxB = reinterpret_cast<const uint32_t*>(kexps[si]);
        }
      Lp = &Lcopy;
    }
  }
  const Launch& L = *Lp;
  const bool bcols = (EPI == EPI_SLAB) && L.n > 0 && L.p[0].bias_cols != 0;  // (uniform per launch: see the wgrad entry)
  (void)bcols;
  // The weight-gradient GEMMs run on a side stream next to the HBM-bound table optimizer (trainer.py).  An unused
  // dynamic-LDS request (mml_gemm_set_wgrad_lds_pad) lowers their residency so that the optimizer's waves co-reside.
  if (g_wgrad_pad < 0) {
    const char* e = getenv("MMLREC_WGRAD_LDS_PAD");
    g_wgrad_pad = e ? atoi(e) : 0;
  }
  const size_t dyn = (EPI == EPI_SLAB) ? (size_t)g_wgrad_pad : 0;
  if (pipe_ok(L, arc, brc, bn, EPI)) {
    note_kernel("gemm_pipe_kernel", arc, brc, bn, EPI, emu, bcols ? 1 : 0, bpl);
    // persistent workgroups: one per resident slot (128 x 128 tiles: 2 per CU; 128 x 64: 3), each loops over tiles
    static int cus = 0;
    if (cus == 0) {
      int dev = 0, n = 0;
      if (hipGetDevice(&dev) != hipSuccess ||
          hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
        n = 256;
      cus = n;
    }
    const int64_t slots = (int64_t)cus * ((bn == 64 && EPI != EPI_DGRAD && !k7) ? 3 : 2);
    if (nblocks > slots) g = dim3((unsigned)slots);
#ifdef MML_LAB  // ablation builds (tools/lab): one tile width / arithmetic only, to keep compile times short
#define MML_GL(A_, B_)                                                                                            \
  do {                                                                                                            \
    if (bpl) MML_LAUNCH((gemm_pipe_kernel<A_, B_, MML_LAB_BN, EPI, MML_LAB_EMU, false, (EPI != EPI_SLAB && MML_LAB_EMU == 2)>), \
                        g, b, dyn, st, L);                                                                        \
    else MML_LAUNCH((gemm_pipe_kernel<A_, B_, MML_LAB_BN, EPI, MML_LAB_EMU, false>), g, b, dyn, st, L);           \
  } while (0)
#else
#define MML_GL3(A_, B_, N_, C_)                                                              \
  do {                                                                                       \
    /* (nn.Linear weights [N, K] only; gate mode on 128 x 64 tiles only: h, g and both gradients of a 128-wide */ \
    /* tile do not fit the registers next to the accumulators -- the host picks the narrow tiles for it)       */ \
    if constexpr (EPI != EPI_SLAB && ((EPI == EPI_FWD) == (B_)) && (EPI == EPI_FWD || N_ == 64)) {                 \
      if (k7) { /* (the two-plane and three-plane forms only: what the engine runs) */            \
        if (emu == 2 && bpl) MML_LAUNCH((gemm_pipe_kernel<A_, B_, N_, EPI, 2, false, true, true>), g, b, dyn, st, L);  \
        else if (emu == 2) MML_LAUNCH((gemm_pipe_kernel<A_, B_, N_, EPI, 2, false, false, true>), g, b, dyn, st, L);   \
        else if (emu == 3) MML_LAUNCH((gemm_pipe_kernel<A_, B_, N_, EPI, 3, false, false, true>), g, b, dyn, st, L);   \
        else { set_error("%s: the K7 products run on the plane-emulated GEMM forms only", who); return MML_ERR_UNSUPPORTED; } \
        break;                                                                                    \
      }                                                                                           \
    }                                                                                             \
    if (k7) { set_error("%s: the K7 products need nn.Linear weights ([N, K])", who); return MML_ERR_UNSUPPORTED; }  \
    if (emu == 0) MML_LAUNCH((gemm_pipe_kernel<A_, B_, N_, EPI, 0, C_>), g, b, dyn, st, L);       \
    else if (emu == 1) MML_LAUNCH((gemm_pipe_kernel<A_, B_, N_, EPI, 1, C_>), g, b, dyn, st, L);  \
    else if (emu == 2 && bpl)                                                                     \
      MML_LAUNCH((gemm_pipe_kernel<A_, B_, N_, EPI, 2, C_, (EPI != EPI_SLAB)>), g, b, dyn, st, L); \
    else if (emu == 2) MML_LAUNCH((gemm_pipe_kernel<A_, B_, N_, EPI, 2, C_>), g, b, dyn, st, L);  \
    else MML_LAUNCH((gemm_pipe_kernel<A_, B_, N_, EPI, 3, C_>), g, b, dyn, st, L);                \
  } while (0)
#define MML_GL2(A_, B_, N_)                                  \
  do {                                                       \
    if (EPI == EPI_SLAB && bcols) MML_GL3(A_, B_, N_, (EPI == EPI_SLAB)); \
    else MML_GL3(A_, B_, N_, false);                         \
  } while (0)
#define MML_GL(A_, B_)                  \
  do {                                  \
    if (bn == 64) MML_GL2(A_, B_, 64);  \
    else MML_GL2(A_, B_, 128);          \
  } while (0)
#endif
    // (weight-gradient launches are always row-contiguous on both operands, the others reduction-contiguous on A)
    if constexpr (EPI == EPI_SLAB) {
      if (!arc && !brc) MML_GL(false, false);
      else { set_error("%s: unsupported operand layout", who); return MML_ERR_UNSUPPORTED; }
    } else {
      if (arc && brc) MML_GL(true, true);
      else if (arc && !brc) MML_GL(true, false);
      else { set_error("%s: unsupported operand layout", who); return MML_ERR_UNSUPPORTED; }
    }
#undef MML_GL
#ifndef MML_LAB
#undef MML_GL2
#undef MML_GL3
#endif
    return check_launch(who);
  }
#ifdef MML_LAB
  set_error("%s: lab build has only the direct-to-LDS kernel", who);
  return MML_ERR_UNSUPPORTED;
#else
  // shapes the LDS-DMA path cannot take (reduction extents not a multiple of 16, unaligned operands): register-staged
  // fp32 MFMA kernel
  note_kernel("gemm_kernel", arc, brc, bn, EPI, 0);
#define MML_GO(A_, B_, N_) MML_LAUNCH((gemm_kernel<A_, B_, N_, EPI>), g, b, dyn, st, L)
  if (arc && brc) { if (bn == 128) MML_GO(true, true, 128); else MML_GO(true, true, 64); }
  else if (arc && !brc) { if (bn == 128) MML_GO(true, false, 128); else MML_GO(true, false, 64); }
  else if (!arc && !brc) { if (bn == 128) MML_GO(false, false, 128); else MML_GO(false, false, 64); }
  else {
    set_error("%s: unsupported operand layout", who);
    return MML_ERR_UNSUPPORTED;
  }
#undef MML_GO
  return check_launch(who);
#endif
}

// Tile width and arithmetic of one grouped launch.  kind: 0 fwd, 1 dgrad, 2 wgrad; kred = longest reduction extent.
// Measured on MI355X at M = 65536 with gemm_pipe_kernel (tools/gsweep.sh, us):
//                                  fp32 MFMA 128x64 | 128x128     three planes 128x64 | 128x128
//   L1 4x(240->256)+2x(240->64)  fwd 346|345 wgrad 618|400 dgrad 384|330     fwd 295|262 wgrad 488|286 dgrad 304|241
//   L2 4x(256->128)              fwd 179|152 wgrad 267|201 dgrad 228|234     fwd 145|113 wgrad 209|154 dgrad 202|197
//   towers 2x(128->64)           fwd  40| 46 wgrad  93| 70 dgrad  44| 34     fwd  32| 36 wgrad  75| 59 dgrad  34| 27
// 128x128 tiles run two workgroups per CU (<= 256 VGPRs), 128x64 three.
static TileChoice pick_tiles(const int32_t* Ns, int n, int kind, int64_t kred, int64_t row_tiles) {
  static int force = -1;
  if (force < 0) {
    const char* e = getenv("MMLREC_GEMM_BN");
    force = e ? atoi(e) : 0;
  }
  const int mode = gemm_mode();
  TileChoice tc{64, 0};
#ifdef MML_LAB
  return TileChoice{MML_LAB_BN, MML_LAB_EMU};
#endif
  int64_t pad64 = 0, pad128 = 0;
  for (int i = 0; i < n; ++i) {
    pad64 += cdiv(Ns[i], 64) * 64;
    pad128 += cdiv(Ns[i], 128) * 128;
  }
  tc.bn = (pad128 * 100 <= pad64 * 150) ? 128 : 64;  // wide tiles unless > 1/3 of their columns would be padding
  if (row_tiles * (pad128 / 128) < 1024) tc.bn = 64;  // ... or they would not fill the 512 resident slots twice
  if (force == 64 || force == 128) tc.bn = force;
  tc.emu = (mode == 0) ? 0 : (mode == 1 ? 1 : 3);  // (3 becomes 2 in launch_tiles when the magnitudes are there)
  (void)kind;
  (void)kred;
  return tc;
}

}  // namespace mml

using namespace mml;

extern "C" int mml_gemm_set_mode(int32_t mode) {
  MML_REQUIRE(mode == 0 || mode == 1 || mode == 2 || mode == 3 || mode == 4,
              "mml_gemm_set_mode: mode must be 0 (fp32 MFMA), 1 (bf16 operands, reduced precision), 2 / 4 (auto: fp32 "
              "emulated from two scaled fp16 planes where the operand magnitudes are given, else from three bf16 "
              "planes) or 3 (three bf16 planes on every LDS-DMA launch)");
  g_gemm_mode = mode;
  return MML_OK;
}

extern "C" int mml_gemm_get_mode(void) { return gemm_mode(); }

extern "C" int mml_gemm_planes_cut(const mml_planes_desc* d, int32_t n, mml_stream_t stream) {
  MML_REQUIRE(n >= 0 && (n == 0 || d), "mml_gemm_planes_cut: bad descriptor array");
  int i = 0;
  while (i < n) {
    PlanesLaunch L{};
    int64_t items = 0;
    while (i < n && L.n < PLANES_PER_LAUNCH) {
      const mml_planes_desc& q = d[i];
      MML_REQUIRE(q.W && q.planes && q.kexp && q.rows >= 0 && q.cols >= 0 && q.ld >= q.cols,
                  "mml_gemm_planes_cut: matrix %d malformed", i);
      MML_REQUIRE(q.layout == MML_PLANES_ROWS || q.layout == MML_PLANES_COLS, "mml_gemm_planes_cut: layout of matrix %d", i);
      const int64_t ldp = q.ldp ? q.ldp : q.ld;
      MML_REQUIRE(q.layout == MML_PLANES_ROWS ? ldp >= (q.cols + 15) / 16 * 16 : (q.rows % 16 == 0 && ldp >= q.cols),
                  "mml_gemm_planes_cut: matrix %d: the planes pitch cannot hold the reduction extent rounded up to 16 "
                  "(ROWS), or the row count is not a multiple of 16 (COLS)", i);
      MML_REQUIRE(q.n_amax >= 1 && q.n_amax * (q.W2 ? 2 : 1) <= MML_MAX_SRC,
                  "mml_gemm_planes_cut: matrix %d needs 1..%d magnitude slots (half as many pairs with W2)", i, MML_MAX_SRC);
      MML_REQUIRE(!q.W2 || q.ld2 >= q.cols, "mml_gemm_planes_cut: matrix %d: ld2 < cols", i);
      for (int a = 0; a < q.n_amax * (q.W2 ? 2 : 1); ++a)
        MML_REQUIRE(q.amax[a], "mml_gemm_planes_cut: null magnitude slot (matrix %d)", i);
      const int64_t it = q.layout == MML_PLANES_ROWS ? q.rows * ((q.cols + 15) / 16) : (q.rows / 16) * q.cols;
      MML_REQUIRE(items + it < 0x7fffffff, "mml_gemm_planes_cut: too many blocks");
      L.item0[L.n] = (int32_t)items;
      L.d[L.n++] = q;
      items += it;
      ++i;
    }
    L.item0[L.n] = (int32_t)items;
    if (items == 0) continue;
    int64_t most = 0;
    for (int k = 0; k < L.n; ++k) most = (L.item0[k + 1] - L.item0[k]) > most ? (L.item0[k + 1] - L.item0[k]) : most;
    int64_t nb = cdiv(most, 256);
    if (nb > 256) nb = 256;
    if (nb < 1) nb = 1;
    MML_LAUNCH(planes_cut_kernel, dim3((unsigned)nb, (unsigned)L.n), dim3(256), 0, to_stream(stream), L);
    int rc = check_launch("mml_gemm_planes_cut");
    if (rc) return rc;
  }
  return MML_OK;
}

extern "C" const char* mml_gemm_last_kernel(void) { return g_last_kernel; }


extern "C" int mml_gemm_set_wgrad_lds_pad(int32_t bytes) {
  MML_REQUIRE(bytes >= 0 && bytes <= 64 * 1024, "mml_gemm_set_wgrad_lds_pad: bytes outside [0, 65536]");
  g_wgrad_pad = bytes;
  return MML_OK;
}

// gemm_panel.hip: the activation-stationary kernel for launches whose problems all read ONE input (first DNN layers)
int mml_gemm_panel_try_fwd(const mml_gemm_fwd_desc* d, int32_t n, hipStream_t st);

// gemm_ws.hip: the weight-stationary kernel for launches whose problems each stream their own input past a weight that
// fits the LDS (second expert layers, towers, and their input gradients)
int mml_gemm_ws_try_fwd(const mml_gemm_fwd_desc* d, int32_t n, hipStream_t st);
int mml_gemm_ws_try_dgrad(const mml_gemm_dgrad_desc* d, int32_t n, hipStream_t st);
namespace mml { const char* mml_gemm_ws_last_symbol(); }

extern "C" int mml_gemm_grouped_fwd(const mml_gemm_fwd_desc* d, int32_t n, mml_stream_t stream) {
  MML_REQUIRE(n >= 0 && (n == 0 || d), "mml_gemm_grouped_fwd: bad descriptor array");
  if (n > 0 && gemm_mode() >= 2 && gemm_mode() != 3) {  // (auto / two-plane arithmetic only)
    int rc = mml_gemm_panel_try_fwd(d, n, to_stream(stream));
    if (rc != MML_ERR_UNSUPPORTED) {
      if (rc == MML_OK) snprintf(g_last_kernel, sizeof(g_last_kernel), "gemm_panel_kernel");
      return rc;
    }
    rc = mml_gemm_ws_try_fwd(d, n, to_stream(stream));
    if (rc != MML_ERR_UNSUPPORTED) {
      if (rc == MML_OK) snprintf(g_last_kernel, sizeof(g_last_kernel), "%s", mml::mml_gemm_ws_last_symbol());
      return rc;
    }
  }
  // split into runs of <= MML_MAX_GROUP problems with a uniform weight layout
  int i = 0;
  while (i < n) {
    int j = i;
    int32_t Ns[MML_MAX_GROUP];
    Launch L{};
    bool any_k7 = false;
    const uint32_t* planes[MAX_SOURCES] = {};
    const int32_t* kexps[MAX_SOURCES] = {};
    while (j < n && j - i < MML_MAX_GROUP && d[j].w_kn == d[i].w_kn && d[j].M == d[i].M) {
      const mml_gemm_fwd_desc& q = d[j];
      MML_REQUIRE(q.A && q.W && q.C, "mml_gemm_grouped_fwd: null pointer in problem %d", j);
      MML_REQUIRE(q.M >= 0 && q.N > 0 && q.K > 0, "mml_gemm_grouped_fwd: bad sizes in problem %d", j);
      MML_REQUIRE(q.lda >= q.K && q.ldc >= q.N && q.ldw >= (q.w_kn ? q.N : q.K),
                  "mml_gemm_grouped_fwd: leading dimension too small in problem %d", j);
      Problem& P = L.p[j - i];
      P.nsrc = 1;
      P.src0 = j - i;
      Source& S0 = L.src[j - i];
      S0.A = q.A; S0.lda = q.lda; S0.B = q.W; S0.ldb = q.ldw; S0.Kred = q.K;
      S0.vecA = vec_ok(q.A, q.lda); S0.vecB = vec_ok(q.W, q.ldw);
      S0.amaxA = q.amax_a; S0.amaxB = q.amax_w;
      planes[j - i] = q.w_planes;  // ([K, N] weights: the MML_PLANES_COLS image, K6)
      kexps[j - i] = q.w_kexp;
      P.amax_out = q.amax_out;
      P.M = q.M; P.N = q.N; P.C = q.C; P.ldc = q.ldc; P.bias = q.bias; P.act = q.act;
      MML_REQUIRE(!q.relu_mask || q.ldmask * 32 >= q.N, "mml_gemm_grouped_fwd: ldmask too small in problem %d", j);
      P.mask = (q.act == MML_ACT_RELU) ? q.relu_mask : nullptr; P.ldmask = q.ldmask;
      P.vec_out = (vec_ok(q.C, q.ldc) && q.N % 4 == 0 && (!q.bias || aligned16(q.bias))) ? 1 : 0;
      if (q.mul || q.prod) {  // K7: prod = C * mul from the same epilogue (see Problem)
        MML_REQUIRE(q.mul && q.prod && q.ldmul >= q.N && q.ldprod >= q.N,
                    "mml_gemm_grouped_fwd: problem %d needs both mul and prod, with pitches >= N", j);
        MML_REQUIRE(P.vec_out && vec_ok(q.mul, q.ldmul) && vec_ok(q.prod, q.ldprod),
                    "mml_gemm_grouped_fwd: mul / prod of problem %d need 16-byte aligned operands and N %% 4 == 0", j);
        P.Y = q.mul; P.ldy = q.ldmul; P.bias_slab = q.prod; P.slab_off = q.ldprod; P.amax_out2 = q.amax_prod;
        any_k7 = true;
      }
      Ns[j - i] = q.N;
      ++j;
    }
    L.n = j - i;
    int64_t kred = 0;
    for (int k = 0; k < L.n; ++k) kred = L.src[k].Kred > kred ? L.src[k].Kred : kred;
    const TileChoice tc = pick_tiles(Ns, L.n, 0, kred, cdiv(d[i].M, BM));
    const int bn = tc.bn;
    int t = 0;
    for (int k = 0; k < L.n; ++k) {
      L.p[k].tiles_n = (int)cdiv(L.p[k].N, bn);
      L.p[k].tile0 = t;
      t += L.p[k].tiles_n;
    }
    L.total_ntiles = t;
    L.tiles_m = (int)cdiv(d[i].M, BM);
    if (any_k7 && !pipe_ok(L, true, d[i].w_kn == 0, bn, EPI_FWD)) {
      set_error("mml_gemm_grouped_fwd: mul / prod need the LDS-DMA kernel (K %% 16 == 0, 16-byte aligned operands)");
      return MML_ERR_UNSUPPORTED;
    }
    int rc = launch_tiles<EPI_FWD>(L, true, d[i].w_kn == 0, tc, (int64_t)L.tiles_m * t, to_stream(stream),
                                   "mml_gemm_grouped_fwd", planes, kexps, any_k7);
    if (rc) return rc;
    i = j;
  }
  return MML_OK;
}

// gemm_os.hip: the output-stationary kernel for ONE wide input gradient summed over many sources (d(dnn_input): every
// expert's and gate's first layer feeds it)
int mml_gemm_os_try_dgrad(const mml_gemm_dgrad_desc* d, int32_t n, hipStream_t st);

extern "C" int mml_gemm_grouped_dgrad(const mml_gemm_dgrad_desc* d, int32_t n, mml_stream_t stream) {
  MML_REQUIRE(n >= 0 && (n == 0 || d), "mml_gemm_grouped_dgrad: bad descriptor array");
  if (n > 0 && gemm_mode() >= 2 && gemm_mode() != 3) {
    int rc = mml_gemm_os_try_dgrad(d, n, to_stream(stream));
    if (rc != MML_ERR_UNSUPPORTED) {
      if (rc == MML_OK) snprintf(g_last_kernel, sizeof(g_last_kernel), "gemm_os_kernel");
      return rc;
    }
    rc = mml_gemm_ws_try_dgrad(d, n, to_stream(stream));
    if (rc != MML_ERR_UNSUPPORTED) {
      if (rc == MML_OK) snprintf(g_last_kernel, sizeof(g_last_kernel), "%s", mml::mml_gemm_ws_last_symbol());
      return rc;
    }
  }
  int i = 0;
  while (i < n) {
    int j = i;
    int32_t Ns[MML_MAX_GROUP];
    Launch L{};
    const int32_t lay = d[i].n_src > 0 ? d[i].w_kn[0] : 0;
    int nsources = 0;
    bool any_k7 = false;
    const uint32_t* planes[MAX_SOURCES] = {};
    const int32_t* kexps[MAX_SOURCES] = {};
    while (j < n && j - i < MML_MAX_GROUP && d[j].M == d[i].M) {
      const mml_gemm_dgrad_desc& q = d[j];
      MML_REQUIRE((q.dA || q.gate_h) && q.n_src >= 1 && q.n_src <= MML_MAX_SRC, "mml_gemm_grouped_dgrad: problem %d malformed", j);
      MML_REQUIRE(q.M >= 0 && q.K > 0 && q.ldda >= q.K, "mml_gemm_grouped_dgrad: bad sizes in problem %d", j);
      MML_REQUIRE(q.act == MML_ACT_NONE || q.Y || q.relu_mask, "mml_gemm_grouped_dgrad: act set but Y null in problem %d", j);
      bool same = true;
      for (int s = 0; s < q.n_src; ++s) same = same && (q.w_kn[s] == lay);
      if ((!same || nsources + q.n_src > MAX_SOURCES) && j > i) break;
      MML_REQUIRE(same, "mml_gemm_grouped_dgrad: mixed weight layouts inside problem %d", j);
      Problem& P = L.p[j - i];
      P.nsrc = q.n_src;
      P.src0 = nsources;
      for (int s = 0; s < q.n_src; ++s) {
        MML_REQUIRE(q.dC[s] && q.W[s] && q.N[s] > 0, "mml_gemm_grouped_dgrad: null source %d in problem %d", s, j);
        Source& S = L.src[nsources++];
        S.A = q.dC[s]; S.lda = q.lddc[s]; S.B = q.W[s]; S.ldb = q.ldw[s];
        S.Kred = q.N[s];
        S.vecA = vec_ok(q.dC[s], q.lddc[s]); S.vecB = vec_ok(q.W[s], q.ldw[s]);
        S.amaxA = q.amax_dc[s]; S.amaxB = q.amax_w[s];
        planes[nsources - 1] = q.w_planes[s];  // ([K, N] weights: the MML_PLANES_ROWS image, K6)
        kexps[nsources - 1] = q.w_kexp[s];
      }
      P.amax_out = q.amax_out;
      P.M = q.M; P.N = q.K; P.C = q.dA; P.ldc = q.ldda; P.Y = q.Y; P.ldy = q.ldy; P.act = q.act;
      MML_REQUIRE(!q.relu_mask || (q.act == MML_ACT_RELU && q.ldmask * 32 >= q.K),
                  "mml_gemm_grouped_dgrad: relu_mask needs act == RELU and ldmask >= K/32 (problem %d)", j);
      P.mask = const_cast<uint32_t*>(q.relu_mask); P.ldmask = q.ldmask;
      P.vec_out = (vec_ok(q.dA, q.ldda) && q.K % 4 == 0 && (!q.Y || vec_ok(q.Y, q.ldy))) ? 1 : 0;
      P.accumulate = q.accumulate;
      if (q.gate_h) {  // K7 backward: the gradients of the two factors of the gated input (see Problem)
        MML_REQUIRE(q.gate_g && q.d_h && q.d_g && q.ld_h >= q.K && q.ld_g >= q.K && q.ld_dh >= q.K && q.ld_dg >= q.K,
                    "mml_gemm_grouped_dgrad: gate mode of problem %d needs gate_g, d_h, d_g and pitches >= K", j);
        MML_REQUIRE(q.K % 4 == 0 && vec_ok(q.gate_h, q.ld_h) && vec_ok(q.gate_g, q.ld_g) && vec_ok(q.d_h, q.ld_dh) &&
                        vec_ok(q.d_g, q.ld_dg),
                    "mml_gemm_grouped_dgrad: gate mode of problem %d needs 16-byte aligned operands and K %% 4 == 0", j);
        P.C = q.d_h; P.ldc = q.ld_dh; P.Y = q.gate_h; P.ldy = q.ld_h; P.act = q.act_h; P.accumulate = q.acc_h;
        P.mask = nullptr; P.ldmask = q.ld_g;  // (ldmask carries the pitch of g in gate mode)
        P.bias = q.gate_g; P.bias_slab = q.d_g; P.slab_off = q.ld_dg; P.tiles_m = q.act_g; P.bias_cols = q.acc_g;
        P.amax_out = q.amax_dh; P.amax_out2 = q.amax_dg;
        P.vec_out = 1;
        any_k7 = true;
      } else {
        MML_REQUIRE(q.dA, "mml_gemm_grouped_dgrad: problem %d has no dA", j);
      }
      Ns[j - i] = q.K;
      ++j;
    }
    L.n = j - i;
    int64_t kred = 0;
    for (int k = 0; k < L.n; ++k) {
      int64_t sum = 0;
      for (int s2 = 0; s2 < L.p[k].nsrc; ++s2) sum += L.src[L.p[k].src0 + s2].Kred;
      kred = sum > kred ? sum : kred;
    }
    TileChoice tc = pick_tiles(Ns, L.n, 1, kred, cdiv(d[i].M, BM));
    if (any_k7) tc.bn = 64;  // (gate mode: see launch_tiles)
    const int bn = tc.bn;
    int t = 0;
    for (int k = 0; k < L.n; ++k) {
      L.p[k].tiles_n = (int)cdiv(L.p[k].N, bn);
      L.p[k].tile0 = t;
      t += L.p[k].tiles_n;
    }
    L.total_ntiles = t;
    L.tiles_m = (int)cdiv(d[i].M, BM);
    if (any_k7 && !pipe_ok(L, true, lay == 1, bn, EPI_DGRAD)) {
      set_error("mml_gemm_grouped_dgrad: gate mode needs the LDS-DMA kernel (N_s %% 16 == 0, 16-byte aligned operands)");
      return MML_ERR_UNSUPPORTED;
    }
    // col operand = W: reduction index is W's row for [N,K] (not contiguous) and contiguous for [K,N]
    int rc = launch_tiles<EPI_DGRAD>(L, true, lay == 1, tc, (int64_t)L.tiles_m * t, to_stream(stream),
                                     "mml_gemm_grouped_dgrad", planes, kexps, any_k7);
    if (rc) return rc;
    i = j;
  }
  return MML_OK;
}

extern "C" int mml_pep_gate_fwd(const mml_gemm_fwd_desc* d, int32_t n, mml_stream_t stream) {
  MML_REQUIRE(n >= 0 && (n == 0 || d), "mml_pep_gate_fwd: bad descriptor array");
  for (int i = 0; i < n; ++i) MML_REQUIRE(d[i].mul && d[i].prod, "mml_pep_gate_fwd: problem %d carries no mul / prod", i);
  return mml_gemm_grouped_fwd(d, n, stream);
}

extern "C" int mml_star_linear_fwd(const mml_gemm_fwd_desc* d, int32_t n, mml_stream_t stream) {
  MML_REQUIRE(n >= 0 && (n == 0 || d), "mml_star_linear_fwd: bad descriptor array");
  for (int i = 0; i < n; ++i)
    MML_REQUIRE(d[i].w_kn == 1 && d[i].w_planes && d[i].w_kexp,
                "mml_star_linear_fwd: problem %d is not a [K, N]-layout layer with pre-cut planes", i);
  return mml_gemm_grouped_fwd(d, n, stream);
}

extern "C" int mml_star_linear_bwd(const mml_gemm_dgrad_desc* d, int32_t n, mml_stream_t stream) {
  MML_REQUIRE(n >= 0 && (n == 0 || d), "mml_star_linear_bwd: bad descriptor array");
  for (int i = 0; i < n; ++i)
    for (int s2 = 0; s2 < d[i].n_src && s2 < MML_MAX_SRC; ++s2)
      MML_REQUIRE(d[i].w_kn[s2] == 1 && d[i].w_planes[s2] && d[i].w_kexp[s2],
                  "mml_star_linear_bwd: source %d of problem %d is not a [K, N]-layout layer with pre-cut planes", s2, i);
  return mml_gemm_grouped_dgrad(d, n, stream);
}

extern "C" int mml_pep_gate_bwd(const mml_gemm_dgrad_desc* d, int32_t n, mml_stream_t stream) {
  MML_REQUIRE(n >= 0 && (n == 0 || d), "mml_pep_gate_bwd: bad descriptor array");
  for (int i = 0; i < n; ++i) MML_REQUIRE(d[i].gate_h, "mml_pep_gate_bwd: problem %d is not in gate mode", i);
  return mml_gemm_grouped_dgrad(d, n, stream);
}

// wgrad planning shared by the workspace query and the launch
namespace {
struct WgradPlan {
  int bn, emu, S, chunk;
  int64_t total_tiles;
  int64_t slab_floats;  // all problems, all splits, incl. bias slabs
};
WgradPlan plan_wgrad(const mml_gemm_wgrad_desc* d, int i, int j) {
  WgradPlan w{};
  int32_t cols[MML_MAX_GROUP];
  for (int k = i; k < j; ++k) cols[k - i] = d[k].w_kn ? d[k].N : d[k].K;
  int64_t rt = 0;
  for (int k = i; k < j; ++k) rt = cdiv(d[k].w_kn ? d[k].K : d[k].N, BM) > rt ? cdiv(d[k].w_kn ? d[k].K : d[k].N, BM) : rt;
  const TileChoice tc = pick_tiles(cols, j - i, 2, d[i].M, rt * cdiv(d[i].M, 8 * BK));
  w.bn = tc.bn;
  w.emu = tc.emu;
  int64_t tiles = 0, out_elems = 0, bias_elems = 0;
  for (int k = i; k < j; ++k) {
    const int rows = d[k].w_kn ? d[k].K : d[k].N, cls = d[k].w_kn ? d[k].N : d[k].K;
    tiles += cdiv(rows, BM) * cdiv(cls, w.bn);
    out_elems += (int64_t)d[k].N * d[k].K;
    if (d[k].dbias) bias_elems += d[k].N;
  }
  w.total_tiles = tiles;
  const int64_t M = d[i].M;
  int64_t S = tiles > 0 ? (w.bn == 128 ? 512 : 768) / tiles : 1;  // one batch chunk per resident (persistent) workgroup
  const int64_t maxS = cdiv(M, 8 * BK);      // at least 256 batch rows per split
  if (S > maxS) S = maxS;
  if (S < 1) S = 1;
  int64_t chunk = cdiv(cdiv(M, S), BK) * BK;
  if (chunk < BK) chunk = BK;
  S = cdiv(M, chunk);
  if (S < 1) S = 1;
  w.S = (int)S;
  w.chunk = (int)chunk;
  w.slab_floats = S * (out_elems + bias_elems);
  return w;
}
}  // namespace

extern "C" int64_t mml_gemm_grouped_wgrad_workspace_bytes(const mml_gemm_wgrad_desc* d, int32_t n) {
  if (n <= 0 || !d) return 0;
  int64_t best = 0;
  int i = 0;
  while (i < n) {
    int j = i;
    while (j < n && j - i < MML_MAX_GROUP && d[j].M == d[i].M && d[j].w_kn == d[i].w_kn) ++j;
    WgradPlan w = plan_wgrad(d, i, j);
    best += ((w.slab_floats * 4 + 255) / 256) * 256;  // groups are laid out one after another (phased launches)
    i = j;
  }
  return best + 256;
}

// gemm_nt.hip: weight gradients cut once per workgroup, fragments through the transposing LDS read (large batches, both
// operand magnitudes known)
int mml_gemm_nt_try_wgrad(const mml_gemm_wgrad_desc* d, int32_t n, void* workspace, int64_t workspace_bytes, int32_t phase,
                          hipStream_t st);

extern "C" int mml_gemm_grouped_wgrad_phase(const mml_gemm_wgrad_desc* d, int32_t n, void* workspace,
                                            int64_t workspace_bytes, int32_t phase, mml_stream_t stream) {
  MML_REQUIRE(n >= 0 && (n == 0 || d), "mml_gemm_grouped_wgrad: bad descriptor array");
  MML_REQUIRE(phase >= 0 && phase <= 2, "mml_gemm_grouped_wgrad_phase: phase must be 0 (both), 1 (partials) or 2 (reduce)");
  if (n == 0) return MML_OK;
  MML_REQUIRE(workspace && aligned16(workspace), "mml_gemm_grouped_wgrad: workspace null or misaligned");
  if (gemm_mode() >= 2 && gemm_mode() != 3) {  // (auto / two-plane arithmetic only)
    const int rc = mml_gemm_nt_try_wgrad(d, n, workspace, workspace_bytes, phase, to_stream(stream));
    if (rc != MML_ERR_UNSUPPORTED) {
      if (rc == MML_OK && phase != 2) snprintf(g_last_kernel, sizeof(g_last_kernel), "gemm_nt_kernel");
      return rc;
    }
  }
  int i = 0;
  int64_t ws_off = 0;  // bytes: every group of problems owns its own piece of the workspace
  while (i < n) {
    int j = i;
    while (j < n && j - i < MML_MAX_GROUP && d[j].M == d[i].M && d[j].w_kn == d[i].w_kn) ++j;
    WgradPlan w = plan_wgrad(d, i, j);
    MML_REQUIRE(ws_off + w.slab_floats * 4 <= workspace_bytes, "mml_gemm_grouped_wgrad: workspace %lld < %lld bytes",
                (long long)workspace_bytes, (long long)(ws_off + w.slab_floats * 4));
    Launch L{};
    ReduceLaunch R{};
    L.n = j - i;
    L.splits = w.S;
    L.chunk = w.chunk;
    L.slab = reinterpret_cast<float*>(static_cast<char*>(workspace) + ws_off);
    ws_off += ((w.slab_floats * 4 + 255) / 256) * 256;
    int64_t off = 0, rstart = 0;
    int t = 0;
    for (int k = i; k < j; ++k) {
      const mml_gemm_wgrad_desc& q = d[k];
      MML_REQUIRE(q.dC && q.A && q.dW, "mml_gemm_grouped_wgrad: null pointer in problem %d", k);
      MML_REQUIRE(q.M >= 0 && q.N > 0 && q.K > 0, "mml_gemm_grouped_wgrad: bad sizes in problem %d", k);
      MML_REQUIRE(q.lddc >= q.N && q.lda >= q.K && q.lddw >= (q.w_kn ? q.N : q.K),
                  "mml_gemm_grouped_wgrad: leading dimension too small in problem %d", k);
      Problem& P = L.p[k - i];
      P.nsrc = 1;
      P.src0 = k - i;
      Source& S = L.src[k - i];
      // rows of the output tile come from dC^T (n) unless the weight is stored [K,N]; both operands are
      // batch-major in memory, i.e. NOT reduction-contiguous.
      if (!q.w_kn) {
        S.A = q.dC; S.lda = q.lddc; S.B = q.A; S.ldb = q.lda;
        S.amaxA = q.amax_dc; S.amaxB = q.amax_a;
        P.M = q.N; P.N = q.K;
      } else {
        S.A = q.A; S.lda = q.lda; S.B = q.dC; S.ldb = q.lddc;
        S.amaxA = q.amax_a; S.amaxB = q.amax_dc;
        P.M = q.K; P.N = q.N;
      }
      S.Kred = q.M;
      S.vecA = vec_ok(S.A, S.lda); S.vecB = vec_ok(S.B, S.ldb);
      P.tiles_m = (int)cdiv(P.M, BM);
      P.tiles_n = (int)cdiv(P.N, w.bn);
      P.tile0 = t;
      t += P.tiles_m * P.tiles_n;
      P.slab_off = off;
      P.bias_cols = q.w_kn ? 1 : 0;
      const int64_t elems = (int64_t)q.N * q.K;
      ReduceSeg& g = R.seg[R.n++];
      g.slab = L.slab + off; g.out = q.dW; g.n = elems; g.cols = q.w_kn ? q.N : q.K; g.ldo = q.lddw;
      g.S = w.S; g.sstride = elems;
      g.accumulate = q.accumulate; g.start = rstart;
      rstart += elems;
      off += (int64_t)w.S * elems;
      P.bias_slab = nullptr;
      if (q.dbias) {
        P.bias_slab = L.slab + off;
        ReduceSeg& b = R.seg[R.n++];
        b.slab = L.slab + off; b.out = q.dbias; b.n = q.N; b.cols = q.N; b.ldo = q.N;
        b.S = w.S; b.sstride = q.N;
        b.accumulate = q.accumulate; b.start = rstart;
        rstart += q.N;
        off += (int64_t)w.S * q.N;
      }
    }
    L.total_ntiles = t;
    R.total = rstart;
    int rc = MML_OK;
    if (phase != 2)
      rc = launch_tiles<EPI_SLAB>(L, false, false, TileChoice{w.bn, w.emu}, (int64_t)t * w.S, to_stream(stream),
                                  "mml_gemm_grouped_wgrad");
    if (rc) return rc;
    if (phase != 1) rc = launch_slab_reduce(R, to_stream(stream), "mml_gemm_grouped_wgrad(reduce)");
    if (rc) return rc;
    i = j;
  }
  return MML_OK;
}

extern "C" int mml_gemm_grouped_wgrad(const mml_gemm_wgrad_desc* d, int32_t n, void* workspace,
                                      int64_t workspace_bytes, mml_stream_t stream) {
  return mml_gemm_grouped_wgrad_phase(d, n, workspace, workspace_bytes, 0, stream);
}
