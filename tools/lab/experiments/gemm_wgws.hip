// EXPERIMENT (round 4, not part of the library: build it with tools/lab/wg_build.sh): cut-once weight-gradient kernel for
// layers whose whole [N, K] gradient tile fits a workgroup's registers.  Correct (float64 error 3-6e-7 incl. the bias
// gradient) and SLOWER than the tile kernel it was meant to replace, in every form tried (stand-alone with cold caches,
// AE-30 shapes at B = 65 536, tools/lab/wg_time.py; tile kernel: second expert layers 143-147 us, towers 54-56):
//   every wave loads + cuts + multiplies, cut phase in front of the MFMAs ............ 172.7 / 54.8 us
//   the cut of step s + 1 in the same basic block as the MFMAs of step s ............ 175.6 / 55.2
//   four consumer waves + four loader waves (one of each per SIMD) ................... 175.4 / 55.2
//   + three register stages of rows (look-ahead of three steps; spills at two tasks per lane) ... 355.9 / 55.8
// and with parts compiled out (first form): no MFMAs 170, no cut and no loads 87, rows loaded once 110, nothing but the
// barriers, the slab stores and the reduction 55 (33 MB of partial tiles written and read back: a third of the time, the
// same for the tile kernel).  PMC of the first form: 6.5 VALU instructions per MFMA (tile kernel: 12), waves waiting 60 %
// of their cycles.  What it would need next is not known; the in-step tile kernel (108 us) stays.
//
// dW[n][k] = sum_b dC[b][n] A[b][k] (reference model/utils.py:146-161 under autograd: the Linear layers of the expert, gate
// and tower stacks).  The tile kernel (gemm.hip, EPI_SLAB) treats it as a GEMM with the batch as the reduction: both
// operands are activations, both are cut into their fp16 planes in registers by EVERY wave that reads them (a fragment of
// dnn_input by 9 x 2 waves, a fragment of dC by 2 x 2: 12 VALU instructions per MFMA, matrix pipe 0.36 busy) after a
// transposing read from an fp32 LDS image.  Here a workgroup (eight waves) owns ONE problem's whole [N, K] tile for a chunk
// of the batch:
//   * every 32 rows of the chunk, the workgroup's lanes load dC and A row-major (a lane: four columns x eight rows, a wave
//     instruction: 512 contiguous bytes of two rows), cut each element ONCE and write the planes to LDS already transposed
//     -- one ds_write_b128 per column and plane, 16 lanes covering 256 contiguous bytes;
//   * LDS image per k-step (16 rows) and plane: [32-column block][position][lane half] x 16 bytes, where a block's column
//     4 g + j sits at position 8 j + g: the writes of one instruction are contiguous, and so is a wave's fragment read (1 KiB);
//     lane r of a fragment therefore stands for column 4 (r % 8) + r / 8 of its block, on both operands (the output map
//     follows);
//   * two buffers, ONE barrier per 32 rows; every wave keeps its share of the tile ((N / 32 / WN) x (K / 32 / WK)
//     sub-tiles of 32 x 32) in accumulators for the whole chunk and stores it once into the partial-sum slab the existing
//     reduction (launch_slab_reduce) sums; the bias gradient (column sums of dC) rides with the loaders.
// Not bitwise the tile kernel (another summation order over the batch); float64 error like the other two-plane kernels.
#include "common.hpp"
#include "lds_async.hpp"
#include "reduce.hpp"

#include <stdlib.h>

#include <type_traits>

namespace mml {

using gf32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int WG_MIN_ROWS = 8192;
#ifndef WG_LAB
#define WG_LAB 0  // lab builds (results are garbage): 1 no MFMAs, 2 no cut / LDS writes, 4 rows loaded once, 8 no fragment reads
#endif

struct WgProblem {
  const float* dC;          // [M, N]
  const float* A;           // [M, K]
  const uint32_t* amax_dc;  // magnitude slots
  const uint32_t* amax_a;
  float* slab;              // [S][N * K] partial tiles, [n][k] row-major
  float* bias_slab;         // [S][N] or null
  int64_t lddc, lda;
  int32_t N, K;
};

struct WgLaunch {
  int32_t M, n_prob, S, chunk;  // S workgroups per problem, each `chunk` rows (a multiple of 32)
  WgProblem p[MML_MAX_GROUP];
};
static_assert(sizeof(WgLaunch) <= 4096, "WgLaunch must fit the kernel-argument block");

__device__ __forceinline__ uint32_t wg_amax_load(const uint32_t* p) {
  uint32_t m = 0;
#pragma unroll
  for (int i = 0; i < MML_AMAX_WORDS; ++i) m = p[i] > m ? p[i] : m;
  return m;
}
__device__ __forceinline__ int wg_scale_exp(uint32_t bits) {
  const int e = (int)((bits >> 23) & 0xffu);
  if (e == 255) return 0;
  const int k = 141 - e;
  return k > 110 ? 110 : (k < -110 ? -110 : k);
}
__device__ __forceinline__ float wg_pow2(int k) { return __uint_as_float((uint32_t)(127 + k) << 23); }

// The cut of lds_async.hpp (h = rne16(x s), l = rne16(x s - h); 24 instructions per eight values) with the scale in a
// VECTOR register: the lanes of one wave may cut different operands here (dC and A have their own scales).
__device__ __forceinline__ void wg_cut(const float (&x)[8], const float s, f16x8& hi, f16x8& lo) {
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  uint32_t hw[4], lw[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    float y0, y1, r0, r1;
    uint32_t h, l;
    asm("v_mul_f32 %0, %1, %2" : "=v"(y0) : "v"(x[2 * j]), "v"(s));
    asm("v_mul_f32 %0, %1, %2" : "=v"(y1) : "v"(x[2 * j + 1]), "v"(s));
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(h) : "v"(y0), "v"(y1));
    asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(r0) : "v"(x[2 * j]), "v"(s), "v"(h));
    asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(r1) : "v"(x[2 * j + 1]), "v"(s), "v"(h));
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(l) : "v"(r0), "v"(r1));
    hw[j] = h;
    lw[j] = l;
  }
  const u32x4 a = {hw[0], hw[1], hw[2], hw[3]}, b = {lw[0], lw[1], lw[2], lw[3]};
  hi = __builtin_bit_cast(f16x8, a);
  lo = __builtin_bit_cast(f16x8, b);
}

// NT / KT: 32-column blocks of dC (N = 32 NT) and of A (K <= 32 KT, columns beyond K are zeros).  Waves 0..3 CONSUME (one
// per SIMD: WN x WK = 4 of them split the tile's sub-tiles), waves 4..7 LOAD and CUT (one per SIMD): a SIMD's two waves
// run the VALU work of the cut and the MFMAs side by side without any instruction interleaving -- with every wave in
// both roles the barrier kept the two waves of a SIMD in lockstep, cut phase then MFMA phase: 175 us for the second
// expert layers against 146 of the tile kernel, 87 with the cut compiled out.
template <int NT, int KT, int WN, int WK>
__global__ __launch_bounds__(512, 2) void gemm_wgws_kernel(const WgLaunch L) {
  static_assert(WN * WK == 4 && NT % WN == 0 && KT % WK == 0, "four consumer waves split the tile");
  constexpr int SN = NT / WN, SK = KT / WK;   // sub-tiles of a consumer wave
  static_assert(SN * SK <= 8, "at most 128 accumulator registers per lane");
  constexpr int NB = NT + KT;                 // column blocks of the staged image (dC first, then A)
  constexpr int CG = 8 * NB;                  // 4-column groups
  constexpr int TASKS = 4 * CG;               // (column group, k-step, lane half): four columns x eight rows each
  constexpr int TPL = (TASKS + 255) / 256;    // tasks per loader lane
  constexpr int PLANE = NB * 1024;            // bytes of one (k-step, plane) image
  constexpr int BUF = 4 * PLANE;              // two k-steps x two planes
  static_assert(2 * BUF + 4096 <= 160 * 1024, "LDS budget");
  __shared__ __attribute__((aligned(16))) char lds[2 * BUF + 4096];
  float* const lbias = reinterpret_cast<float*>(lds + 2 * BUF);  // [4][256]: column sums of dC per (k-step, lane half)

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int pi = (int)blockIdx.x / L.S, split = (int)blockIdx.x - pi * L.S;
  const WgProblem& P = L.p[pi];
  const int N = P.N, K = P.K;
  const int row0 = split * L.chunk;
  int nds = (L.M - row0) / 32;   // 32-row steps of this chunk
  nds = nds > L.chunk / 32 ? L.chunk / 32 : nds;
  lbias[tid] = 0.f;
  lbias[tid + 512] = 0.f;
  const int kA = __builtin_amdgcn_readfirstlane(wg_scale_exp(wg_amax_load(P.amax_dc)));
  const int kB = __builtin_amdgcn_readfirstlane(wg_scale_exp(wg_amax_load(P.amax_a)));
  __syncthreads();

  if (wave >= 4) {
    // ---- loaders: task t -> (hh = t & 1: which eight rows of the k-step, q = t / 2 -> k-step ks = q / CG, column group cg) ----
    const int lt = tid - 256;
    const float* src[TPL];
    int64_t ld[TPL];
    int wbase[TPL];
    float sc[TPL], isdcf[TPL];
    int bcol[TPL];   // (the task's slot of lbias: part (k-step, half) x column)
    bool live[TPL], used[TPL];
#pragma unroll
    for (int u = 0; u < TPL; ++u) {
      const int t = lt + 256 * u;
      used[u] = t < TASKS;
      const int hh = t & 1, q = t >> 1;
      const int ks = q / CG, cg = q - ks * CG;
      const bool isdc = cg < 8 * NT;
      const int col = isdc ? 4 * cg : 4 * (cg - 8 * NT);   // first of the task's four columns in its operand
      live[u] = used[u] && (isdc ? col < N : col < K);     // (columns beyond K: zeros)
      ld[u] = isdc ? P.lddc : P.lda;
      src[u] = (isdc ? P.dC : P.A) + (int64_t)(row0 + 16 * ks + 8 * hh) * ld[u] + col;
      sc[u] = isdc ? wg_pow2(kA) : wg_pow2(kB);
      isdcf[u] = (isdc && live[u] && P.bias_slab) ? 1.f : 0.f;
      bcol[u] = (ks * 2 + hh) * 256 + col;
      // LDS byte offset of the unit for column j: block cg / 8, position 8 j + cg % 8, half hh
      wbase[u] = ks * 2 * PLANE + (cg >> 3) * 1024 + (cg & 7) * 32 + hh * 16;
    }
    // Three register stages of raw rows: the rows of step s are requested three steps ahead (one step of look-ahead left the
    // loaders waiting a full HBM latency at every barrier: 175 us for the second expert layers, 110 with the rows loaded once).
    float4 raw[3][TPL][8];
    float4 bsum[TPL];
#pragma unroll
    for (int u = 0; u < TPL; ++u) bsum[u] = make_float4(0.f, 0.f, 0.f, 0.f);
    auto load_rows = [&](auto stc, const int ds) __attribute__((always_inline)) {
      constexpr int ST = decltype(stc)::value;
      if (ds >= nds) return;
#pragma unroll
      for (int u = 0; u < TPL; ++u) {
        if (live[u]) {
          const float* p = src[u] + (int64_t)ds * 32 * ld[u];
#pragma unroll
          for (int i = 0; i < 8; ++i) raw[ST][u][i] = *reinterpret_cast<const float4*>(p + i * ld[u]);
        } else {
#pragma unroll
          for (int i = 0; i < 8; ++i) raw[ST][u][i] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
      }
    };
    auto cut_store = [&](auto stc, const int buf) __attribute__((always_inline)) {
      constexpr int ST = decltype(stc)::value;
#pragma unroll
      for (int u = 0; u < TPL; ++u) {
        if (!used[u]) continue;
        char* const wb = lds + buf * BUF + wbase[u];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          bsum[u].x += isdcf[u] * raw[ST][u][i].x;
          bsum[u].y += isdcf[u] * raw[ST][u][i].y;
          bsum[u].z += isdcf[u] * raw[ST][u][i].z;
          bsum[u].w += isdcf[u] * raw[ST][u][i].w;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float x[8];
#pragma unroll
          for (int i = 0; i < 8; ++i)
            x[i] = j == 0 ? raw[ST][u][i].x : (j == 1 ? raw[ST][u][i].y : (j == 2 ? raw[ST][u][i].z : raw[ST][u][i].w));
          f16x8 hp, lp;
          wg_cut(x, sc[u], hp, lp);
          *reinterpret_cast<f16x8*>(wb + j * 256) = hp;
          *reinterpret_cast<f16x8*>(wb + PLANE + j * 256) = lp;
        }
      }
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    // step s lives in stage s % 3; step ds + 1 is cut into the other LDS buffer while the consumers work on step ds, its
    // stage then takes the rows of step ds + 4; ONE barrier per step
    load_rows(I0{}, 0);
    load_rows(I1{}, 1);
    load_rows(I2{}, 2);
    if (nds > 0) cut_store(I0{}, 0);
    load_rows(I0{}, 3);
    __syncthreads();
    for (int ds0 = 0; ds0 < nds; ds0 += 3) {
      if (ds0 < nds) {          // ds = ds0: step ds0 + 1 sits in stage 1
        if (ds0 + 1 < nds) cut_store(I1{}, (ds0 + 1) & 1);
        load_rows(I1{}, ds0 + 4);
        __syncthreads();
      }
      if (ds0 + 1 < nds) {      // ds = ds0 + 1: step ds0 + 2 in stage 2
        if (ds0 + 2 < nds) cut_store(I2{}, (ds0 + 2) & 1);
        load_rows(I2{}, ds0 + 5);
        __syncthreads();
      }
      if (ds0 + 2 < nds) {      // ds = ds0 + 2: step ds0 + 3 in stage 0
        if (ds0 + 3 < nds) cut_store(I0{}, (ds0 + 3) & 1);
        load_rows(I0{}, ds0 + 6);
        __syncthreads();
      }
    }
    // ---- bias gradient: the column sums of dC (four tasks per column: two k-steps x two halves) ----
    if (P.bias_slab) {
#pragma unroll
      for (int u = 0; u < TPL; ++u) {
        if (isdcf[u] != 0.f) {
          lbias[bcol[u]] = bsum[u].x;       // (one task per slot: no atomics, the four parts are added in a fixed order)
          lbias[bcol[u] + 1] = bsum[u].y;
          lbias[bcol[u] + 2] = bsum[u].z;
          lbias[bcol[u] + 3] = bsum[u].w;
        }
      }
    }
  } else {
    // ---- consumers: wave (wn, wk) owns sub-tiles [wn SN, +SN) x [wk SK, +SK) ----
    const int wn = wave % WN, wk = wave / WN;
    gf32x16 acc[SN][SK];
#pragma unroll
    for (int a = 0; a < SN; ++a)
#pragma unroll
      for (int b = 0; b < SK; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    // (fragment lane r = lane % 32, half lane / 32 -> the unit at position r, half: 32 r + 16 (lane / 32) bytes into its block)
    const int rdl = (lane & 31) * 32 + (lane >> 5) * 16;
    const int rdn = (wn * SN) * 1024 + rdl;          // + sub-tile * 1024: the dC fragment of this lane
    const int rdk = (NT + wk * SK) * 1024 + rdl;     // the A fragments
    __syncthreads();
    for (int ds = 0; ds < nds; ++ds) {
      const char* const rb = lds + (ds & 1) * BUF;
#pragma unroll
      for (int s = 0; s < 2; ++s) {   // the two k-steps of the 32 rows
        f16x8 dh[SN], dl[SN], ah[SK], al[SK];
#pragma unroll
        for (int a = 0; a < SN; ++a) {
          dh[a] = *reinterpret_cast<const f16x8*>(rb + s * 2 * PLANE + rdn + a * 1024);
          dl[a] = *reinterpret_cast<const f16x8*>(rb + s * 2 * PLANE + PLANE + rdn + a * 1024);
        }
#pragma unroll
        for (int b = 0; b < SK; ++b) {
          ah[b] = *reinterpret_cast<const f16x8*>(rb + s * 2 * PLANE + rdk + b * 1024);
          al[b] = *reinterpret_cast<const f16x8*>(rb + s * 2 * PLANE + PLANE + rdk + b * 1024);
        }
        // rows of the product = A's columns (k), lanes = dC's columns (n); smallest products first
#pragma unroll
        for (int a = 0; a < SN; ++a)
#pragma unroll
          for (int b = 0; b < SK; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[b], dh[a], acc[a][b], 0, 0, 0);
#pragma unroll
        for (int a = 0; a < SN; ++a)
#pragma unroll
          for (int b = 0; b < SK; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[b], dl[a], acc[a][b], 0, 0, 0);
#pragma unroll
        for (int a = 0; a < SN; ++a)
#pragma unroll
          for (int b = 0; b < SK; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[b], dh[a], acc[a][b], 0, 0, 0);
      }
      __syncthreads();
    }
    // ---- the chunk's partial tile -> slab[split]; lane r of a fragment stands for column 4 (r % 8) + r / 8 of its block ----
    const float inv = wg_pow2(-kA) * wg_pow2(-kB);
    float* const slab = P.slab + (int64_t)split * N * K;
    const int l31 = lane & 31, h2 = lane >> 5;
    const int nloc = 4 * (l31 & 7) + (l31 >> 3);
#pragma unroll
    for (int a = 0; a < SN; ++a) {
      const int n = 32 * (wn * SN + a) + nloc;
#pragma unroll
      for (int b = 0; b < SK; ++b) {
        const int kblk = 32 * (wk * SK + b);
#pragma unroll
        for (int m = 0; m < 16; ++m) {
          const int i = 8 * (m >> 2) + 4 * h2 + (m & 3);   // row of the product = fragment lane of A
          const int k = kblk + 4 * (i & 7) + (i >> 3);
          if (n < N && k < K) slab[(int64_t)n * K + k] = acc[a][b][m] * inv;
        }
      }
    }
  }
  if (P.bias_slab) {
    __syncthreads();
    if (tid < N) P.bias_slab[(int64_t)split * N + tid] = (lbias[tid] + lbias[256 + tid]) + (lbias[512 + tid] + lbias[768 + tid]);
  }
}

static int g_wgws_on = -1;
static int wgws_enabled() {
  if (g_wgws_on < 0) {
    const char* e = getenv("MMLREC_GEMM_WGWS");
    g_wgws_on = (e && e[0] == '0') ? 0 : 1;
  }
  return g_wgws_on;
}
static int wgws_cus() {
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

// shape class of a problem: 0 = not served
static int wgws_class(const mml_gemm_wgrad_desc& q) {
  if (q.w_kn || q.K <= 0 || q.K % 4 != 0) return 0;
  if (q.N == 128 && q.K <= 256 && q.K > 128) return 1;   // <4, 8, 2, 2>
  if (q.N == 64 && q.K <= 128) return 2;                 // <2, 4, 2, 2>
  if (q.N == 64 && q.K <= 256 && q.K > 128) return 4;    // <2, 8, 2, 2>
  if (q.N == 128 && q.K <= 128) return 5;                // <4, 4, 2, 2>
  return 0;
}

}  // namespace mml

using namespace mml;

extern "C" int mml_gemm_set_wgws(int32_t on) {
  g_wgws_on = on ? 1 : 0;
  return MML_OK;
}

// Workgroups (= partial tiles) per problem for a launch of n problems with M rows
static int wgws_splits(int32_t n, int64_t M) {
  int64_t S = wgws_cus() / (n > 0 ? n : 1);
  const int64_t maxS = M / 256;  // at least 256 rows per chunk
  if (S > maxS) S = maxS;
  return S < 1 ? 1 : (int)S;
}

// Whether the kernel serves the whole launch
static bool wgws_serves(const mml_gemm_wgrad_desc* d, int32_t n) {
  if (n < 1 || n > MML_MAX_GROUP) return false;
  if (d[0].M < WG_MIN_ROWS || d[0].M % 32 != 0) return false;
  for (int i = 0; i < n; ++i) {
    const mml_gemm_wgrad_desc& q = d[i];
    if (q.M != d[0].M || wgws_class(q) == 0) return false;
    if (!q.dC || !q.A || !q.dW || !q.amax_dc || !q.amax_a) return false;
    if (!aligned16(q.dC) || !aligned16(q.A) || q.lddc % 4 != 0 || q.lda % 4 != 0 || q.lddw < q.K) return false;
  }
  return true;
}

// Bytes of workspace the kernel needs for the launch, 0 when it does not serve it (flags aside: the caller sizes the
// workspace for both kernels, whichever runs)
int64_t mml_gemm_wgws_workspace_bytes(const mml_gemm_wgrad_desc* d, int32_t n) {
  if (!wgws_serves(d, n)) return 0;
  const int S = wgws_splits(n, d[0].M);
  int64_t fl = 0;
  for (int i = 0; i < n; ++i) fl += (int64_t)S * ((int64_t)d[i].N * d[i].K + (d[i].dbias ? d[i].N : 0));
  return fl * 4 + 256;
}

// The weight gradient of a launch the kernel serves: phase 1 = partial tiles into the workspace (per problem S x N K floats,
// then S x N bias partials), phase 2 = their fixed-order reduction into dW / dbias, 0 = both.  MML_ERR_UNSUPPORTED (no
// error text) when the launch is not served: the caller runs the tile kernel.
int mml_gemm_wgws_try(const mml_gemm_wgrad_desc* d, int32_t n, void* workspace, int64_t workspace_bytes, int32_t phase,
                      hipStream_t st) {
  if (!wgws_enabled() || !wgws_serves(d, n)) return MML_ERR_UNSUPPORTED;
  if (mml_gemm_wgws_workspace_bytes(d, n) > workspace_bytes + 256) return MML_ERR_UNSUPPORTED;  // (sized for the tile kernel only)
  const int64_t M = d[0].M;
  const int S = wgws_splits(n, M);
  const int chunk = (int)(cdiv(cdiv(M, (int64_t)S), 32) * 32);
  float* base = static_cast<float*>(workspace);
  float* slabs[MML_MAX_GROUP];
  float* bslabs[MML_MAX_GROUP];
  ReduceLaunch R{};
  int64_t off = 0, rstart = 0;
  for (int i = 0; i < n; ++i) {
    const mml_gemm_wgrad_desc& q = d[i];
    const int64_t elems = (int64_t)q.N * q.K;
    slabs[i] = base + off;
    ReduceSeg& g = R.seg[R.n++];
    g.slab = base + off; g.out = q.dW; g.n = elems; g.cols = q.K; g.ldo = q.lddw; g.S = S; g.sstride = elems;
    g.accumulate = q.accumulate; g.start = rstart;
    rstart += elems;
    off += (int64_t)S * elems;
    bslabs[i] = nullptr;
    if (q.dbias) {
      bslabs[i] = base + off;
      ReduceSeg& b = R.seg[R.n++];
      b.slab = base + off; b.out = q.dbias; b.n = q.N; b.cols = q.N; b.ldo = q.N; b.S = S; b.sstride = q.N;
      b.accumulate = q.accumulate; b.start = rstart;
      rstart += q.N;
      off += (int64_t)S * q.N;
    }
  }
  R.total = rstart;
  if (phase != 2) {
    bool done[MML_MAX_GROUP] = {};
    for (int i = 0; i < n; ++i) {
      if (done[i]) continue;
      const int cls = wgws_class(d[i]);
      WgLaunch L{};
      for (int j = i; j < n; ++j) {
        if (done[j] || wgws_class(d[j]) != cls) continue;
        done[j] = true;
        WgProblem& P = L.p[L.n_prob++];
        P.dC = d[j].dC; P.A = d[j].A; P.amax_dc = d[j].amax_dc; P.amax_a = d[j].amax_a;
        P.slab = slabs[j]; P.bias_slab = bslabs[j];
        P.lddc = d[j].lddc; P.lda = d[j].lda; P.N = d[j].N; P.K = d[j].K;
      }
      L.M = (int32_t)M;
      L.S = S;
      L.chunk = chunk;
      const dim3 g((unsigned)(L.n_prob * S)), b(512);
      if (cls == 1) MML_LAUNCH((gemm_wgws_kernel<4, 8, 2, 2>), g, b, 0, st, L);
      else if (cls == 2) MML_LAUNCH((gemm_wgws_kernel<2, 4, 2, 2>), g, b, 0, st, L);
      else if (cls == 3) MML_LAUNCH((gemm_wgws_kernel<4, 8, 2, 2>), g, b, 0, st, L);  /* (class 3 is not served) */
      else if (cls == 4) MML_LAUNCH((gemm_wgws_kernel<2, 8, 2, 2>), g, b, 0, st, L);
      else MML_LAUNCH((gemm_wgws_kernel<4, 4, 2, 2>), g, b, 0, st, L);
      const int rc = check_launch("mml_gemm_grouped_wgrad(wgws)");
      if (rc) return rc;
    }
  }
  if (phase != 1) return launch_slab_reduce(R, st, "mml_gemm_grouped_wgrad(wgws reduce)");
  return MML_OK;
}

// (lab / test entry: the kernel alone, without the dispatch of mml_gemm_grouped_wgrad)
extern "C" int mml_gemm_wgws_wgrad(const mml_gemm_wgrad_desc* d, int32_t n, void* workspace, int64_t workspace_bytes,
                                   mml_stream_t stream) {
  const int rc = mml_gemm_wgws_try(d, n, workspace, workspace_bytes, 0, to_stream(stream));
  if (rc == MML_ERR_UNSUPPORTED) set_error("mml_gemm_wgws_wgrad: launch not served (shape, alignment, magnitudes or workspace)");
  return rc;
}
extern "C" int64_t mml_gemm_wgws_wgrad_workspace_bytes(const mml_gemm_wgrad_desc* d, int32_t n) {
  return mml_gemm_wgws_workspace_bytes(d, n);
}
