// K3 (round 5): output-stationary input-gradient GEMM for ONE wide gradient summed over many sources.
//
// d(dnn_input) is the sum over every layer that reads the combined input -- each expert's and each gate's first Linear
// (reference model/mmoe.py:69-79 calls them all on the one dnn_input; autograd adds their `mm` backward products,
// model/utils.py:146-161 is the layer): dA[M, K] = sum_s dC_s[M, N_s] W_s[N_s, K], a reduction of sum N_s = 1 152 for
// AE-30 against K = 240 output columns.  gemm_pipe_kernel (gemm.hip) runs it as 128 x 128 tiles: the 302 MB of dC are
// fetched, staged and cut into their fp16 planes once per N-TILE (twice), and the weights' planes (1.1 MB) are re-read by
// every one of the 1 024 tiles.  Here a workgroup owns 256 rows x ALL K <= 256 columns:
//   * eight waves (two per SIMD: one wave's cuts and LDS reads run under the other's MFMAs without a hand-placed
//     schedule), wave (wr, wc) holds rows 64 wr .. + 63 x columns 128 wc .. + 127 as 2 x 4 accumulator tiles
//     (128 VGPRs) for the whole reduction: the gradient is written ONCE, at the end;
//   * dC travels HBM -> LDS by LDS-DMA in whole 64-byte row pieces (a k-step = 16 reduction elements of 256 rows = 16 KiB),
//     is read as MFMA fragments and cut ONCE per element: h = rne16(x s), l = rne16(x s - h) -- the bits of gemm.hip's cut;
//   * the weights arrive pre-cut (mml_gemm_planes_cut, MML_PLANES_COLS: 16 word rows x K columns per k-step = 16 KiB
//     by LDS-DMA); a lane's fragment is four words of a column, read as two ds_read2st64_b32 per plane;
//   * two rings, five stages for the gradients (HBM: four k-steps in flight) and three for the planes (L2: two), ONE
//     barrier per k-step.
// Same planes, same product order per 16-block (A_h B_l, A_l B_h, A_h B_h), same order of sources and of k as
// gemm_pipe_kernel<.., EMU = 2, BPL>: the results are its bits (tests/test_gemm_os_gpu.py).
#include "common.hpp"
#include "lds_async.hpp"

#include <stdlib.h>

#include <type_traits>

namespace mml {

using of32x16 = __attribute__((ext_vector_type(16))) float;
typedef uint32_t ou32x4 __attribute__((ext_vector_type(4)));

constexpr int OS_BM = 256;                   // rows of a workgroup's panel
constexpr int OS_NC = 256;                   // output columns a workgroup holds
constexpr int OS_DA = 5;                     // ring stages of the gradients (four k-steps in flight)
constexpr int OS_DB = 3;                     // ... of the weights' planes (two in flight)
constexpr int OS_A_STAGE = OS_BM * 64;       // bytes: 256 rows x 16 floats
constexpr int OS_B_STAGE = 16 * OS_NC * 4;   // bytes: 16 word rows x 256 columns
constexpr int OS_A_OFF = 0;
constexpr int OS_B_OFF = OS_A_OFF + OS_DA * OS_A_STAGE;
constexpr int OS_LDS_BYTES = OS_B_OFF + OS_DB * OS_B_STAGE;
static_assert(OS_LDS_BYTES <= 160 * 1024 - 1024, "gemm_os LDS budget");

struct OsSource {
  const float* dC;
  const uint32_t* planes;  // MML_PLANES_COLS image of W [N, K], pitch ldp words
  const uint32_t* amax_dc;
  int64_t lddc, ldp;
  int32_t N, pad_;
};
struct OsLaunch {
  float* dA;
  const int32_t* kexp;  // exponent the planes of ALL sources were cut with (one group)
  uint32_t* amax_out;
  int64_t ldda;
  int32_t M, K, nsrc, accumulate, ksteps, pad_;
  OsSource s[MML_MAX_SRC];
};
static_assert(sizeof(OsLaunch) <= 4096, "OsLaunch must fit the kernel-argument block");

__device__ __forceinline__ uint32_t os_amax_load(const uint32_t* p) {
  uint32_t m = 0;
#pragma unroll
  for (int i = 0; i < MML_AMAX_WORDS; ++i) m = p[i] > m ? p[i] : m;
  return m;
}
// (the rule of gemm.hip: |x| 2^k < 2^15 for every |x| <= the slot's value; Inf / NaN: scale 1)
__device__ __forceinline__ int os_scale_exp(uint32_t bits) {
  const int e = (int)((bits >> 23) & 0xffu);
  if (e == 255) return 0;
  const int k = 141 - e;
  return k > 110 ? 110 : (k < -110 ? -110 : k);
}
__device__ __forceinline__ float os_pow2(int k) { return __uint_as_float((uint32_t)(127 + k) << 23); }

__device__ __forceinline__ void os_wait_vm(const int n) {  // n in {0, 2, 4, 6}
  if (n >= 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  else if (n >= 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else if (n >= 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// LAB (lab builds only, -DMML_OS_LAB: tools/lab/os_lab.sh, MMLREC_OS_LAB): 1 no MFMAs, 2 no cut, 4 no weight fragment reads, 8 no gradient fragment reads,
// 16 no DMA / waits, 32 no barrier (all give garbage results)
template <int LAB>
__global__ __launch_bounds__(512, 1) void gemm_os_kernel(const OsLaunch Larg) {
  typedef const __attribute__((address_space(4))) OsLaunch KLaunch;
  KLaunch& L = *(KLaunch*)__builtin_amdgcn_kernarg_segment_ptr();
  __shared__ __attribute__((aligned(16))) float lds[OS_LDS_BYTES / 4];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const int l31 = lane & 31, h = lane >> 5;
  const uint32_t lds0 = lds_byte_addr(lds);
  const int M = L.M, K = L.K, T = L.ksteps, nsrc = L.nsrc;
  const int npanels = (M + OS_BM - 1) / OS_BM;

  // ONE scale pair for the problem (gemm.hip, problem_scales with pre-cut weights): the smallest exponent over the
  // sources' gradients, the exponent the planes were cut with; where the two add up beyond fp32's range the row operand
  // gives way
  int kA = 110;
  for (int s = 0; s < nsrc; ++s) {
    const int ka = os_scale_exp(os_amax_load(L.s[s].amax_dc));
    kA = ka < kA ? ka : kA;
  }
  const int kB = *L.kexp;
  if (kA + kB > 126) kA = 126 - kB;
  if (kA + kB < -126) kA = -126 - kB;
  kA = __builtin_amdgcn_readfirstlane(kA);
  const float sA = os_pow2(kA);
  const float inv = os_pow2(-(kA + __builtin_amdgcn_readfirstlane(kB)));

  // ---- DMA lane geometry ----
  // dC: wave-instruction i (of 16 per stage) covers rows 16 i .. 16 i + 15 of the panel: lane -> (row, physical chunk);
  // the 16-byte chunk c of row r lands at position c ^ ((r >> 2) & 3) of the row's 64 bytes (the image of gemm.hip)
  const int drow = lane >> 2;
  const int dchunk = (lane & 3) ^ ((drow >> 2) & 3);
  // planes: wave-instruction j (of 16) is word row j of the k-step's block: lane -> columns 4 lane .. 4 lane + 3
  const int bcol = 4 * (lane < (K >> 2) ? lane : (K >> 2) - 1);

  // ---- fragment read addresses (inside a stage) ----
  const int swz = (l31 >> 2) & 3;
  uint32_t a_lo[2], a_hi[2];  // the lane's two chunks of its row: k = 4 h .. 4 h + 3 and 8 + 4 h .. 8 + 4 h + 3
#pragma unroll
  for (int mi = 0; mi < 2; ++mi) {
    const uint32_t row = (uint32_t)(64 * wr + 32 * mi + l31);
    a_lo[mi] = lds0 + OS_A_OFF + row * 64u + (uint32_t)((h ^ swz) << 4);
    a_hi[mi] = lds0 + OS_A_OFF + row * 64u + (uint32_t)(((2 + h) ^ swz) << 4);
  }
  // word rows 4 h + i (h plane) and 8 + 4 h + i (l plane) of column 128 wc + 32 ni + l31
  const uint32_t b_at = lds0 + OS_B_OFF + (uint32_t)(4 * h) * 1024u + (uint32_t)(128 * wc + l31) * 4u;

  float am = 0.f;
  for (int panel = blockIdx.x; panel < npanels; panel += gridDim.x) {
    const int row0 = panel * OS_BM;
    // ---- the two streams of k-steps, each a cursor over (source, 16-block): the gradients come from HBM (~4 us under
    // load: four k-steps ahead), the planes from L2 (two ahead) ----
    int csA = 0, ckbA = 0, nkbA = 0, issuedA = 0;
    int csB = 0, ckbB = 0, nkbB = 0, issuedB = 0;
    const float* pa[2];
    const float* pb[2];
    int64_t bstep = 0;
    auto set_a = [&]() __attribute__((always_inline)) {
      const int64_t lddc = L.s[csA].lddc;
      const float* dC = L.s[csA].dC;
      nkbA = L.s[csA].N >> 4;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        int r = row0 + 16 * (2 * wave + j) + drow;
        r = r < M ? r : M - 1;  // (ragged last panel: the rows beyond the batch repeat the last one and are not stored)
        pa[j] = dC + (int64_t)r * lddc + 4 * dchunk;
      }
    };
    auto set_b = [&]() __attribute__((always_inline)) {
      const int64_t ldp = L.s[csB].ldp;
      const float* pl = reinterpret_cast<const float*>(L.s[csB].planes);
      nkbB = L.s[csB].N >> 4;
      bstep = 16 * ldp;
#pragma unroll
      for (int j = 0; j < 2; ++j) pb[j] = pl + (int64_t)(2 * wave + j) * ldp + bcol;
    };
    auto issue_a = [&]() __attribute__((always_inline)) {
      const int q = issuedA % OS_DA;
      float* As = lds + (OS_A_OFF + q * OS_A_STAGE) / 4 + (2 * wave) * 256;
      dma16(pa[0], As);
      dma16(pa[1], As + 256);
      pa[0] += 16;
      pa[1] += 16;
      ++issuedA;
      if (++ckbA == nkbA) {
        ckbA = 0;
        if (++csA < nsrc) set_a();
      }
    };
    auto issue_b = [&]() __attribute__((always_inline)) {
      const int q = issuedB % OS_DB;
      float* Bs = lds + (OS_B_OFF + q * OS_B_STAGE) / 4 + (2 * wave) * 256;
      dma16(pb[0], Bs);
      dma16(pb[1], Bs + 256);
      pb[0] += bstep;
      pb[1] += bstep;
      ++issuedB;
      if (++ckbB == nkbB) {
        ckbB = 0;
        if (++csB < nsrc) set_b();
      }
    };
    set_a();
    set_b();
    if constexpr (!(LAB & 16)) {
      // the order of the steady state (iteration t issues B(t + 2), then A(t + 4)), wound back: A0 A1 | B0 A2 | B1 A3
      for (int i = 0; i < OS_DA - OS_DB && issuedA < T; ++i) issue_a();
      for (int i = 0; i < OS_DB - 1; ++i) {
        if (issuedB < T) issue_b();
        if (issuedA < T) issue_a();
      }
    }

    of32x16 acc[2][4];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int ni = 0; ni < 4; ++ni)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    f16x8 Ah[2], Al[2], Bh[4], Bl[4];  // the fragments of a k-step
    auto sync_top = [&](const int t) __attribute__((always_inline)) {
      // this wave's DMAs of k-step t have landed -- B(t) and, issued long before it, A(t); what was issued behind B(t) may
      // be in flight: B(t + 1 ..) and A(t + 2 ..) -- then everybody's; the barrier also says that every wave is done with
      // k-step t - 1, whose two slots the next issues overwrite
      if constexpr (!(LAB & 16)) {
        const int nb = issuedB - (t + 1), na = issuedA - (t + 2);
        os_wait_vm(2 * ((nb > 0 ? nb : 0) + (na > 0 ? na : 0)));
      }
      if constexpr (!(LAB & 32)) __builtin_amdgcn_s_barrier();
      if constexpr (!(LAB & 16)) {
        if (issuedB < T) issue_b();
        if (issuedA < T) issue_a();
      }
    };
    f32x4_t xa[2][2];  // a k-step's fragments as they leave the LDS: gradient rows (fp32), weight planes (words)
    f32x2_t wb[4][4];
    auto reads_a = [&](const int t) __attribute__((always_inline)) {
      const uint32_t qa = (uint32_t)((t % OS_DA) * OS_A_STAGE);
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) {
        if constexpr (LAB & 8) {
          xa[mi][0] = xa[mi][1] = f32x4_t{(float)t, 1.f, 2.f, (float)lane};
        } else {
          xa[mi][0] = ds_read128<0>(a_lo[mi] + qa);
          xa[mi][1] = ds_read128<0>(a_hi[mi] + qa);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    };
    auto reads_b = [&](const int t) __attribute__((always_inline)) {
      const uint32_t qb = (uint32_t)((t % OS_DB) * OS_B_STAGE);
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) {
        const uint32_t at = b_at + qb + (uint32_t)(128 * ni);
        if constexpr (LAB & 4) {
          wb[ni][0] = wb[ni][1] = wb[ni][2] = wb[ni][3] = f32x2_t{(float)t, (float)lane};
        } else {
          wb[ni][0] = ds_read2st64<0, 4>(at);
          wb[ni][1] = ds_read2st64<8, 12>(at);
          wb[ni][2] = ds_read2st64<32, 36>(at);
          wb[ni][3] = ds_read2st64<40, 44>(at);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    };
    auto frags_finish = [&]() __attribute__((always_inline)) {
      // (LDS operations return in order: with at most 15 of the 4 + 16 outstanding the four row reads have landed)
      if constexpr (LAB & 4) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt lgkmcnt(15)" ::: "memory");
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) {
        lds_landed(xa[mi][0]);
        lds_landed(xa[mi][1]);
      }
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) {
        const float x[8] = {xa[mi][0].x, xa[mi][0].y, xa[mi][0].z, xa[mi][0].w,
                            xa[mi][1].x, xa[mi][1].y, xa[mi][1].z, xa[mi][1].w};
        if constexpr (LAB & 2) {
          Ah[mi] = __builtin_bit_cast(f16x8, xa[mi][0]);
          Al[mi] = __builtin_bit_cast(f16x8, xa[mi][1]);
        } else {
          F16Cut c;
          f16_cut_hr2(x, sA, c, 0);
          f16_cut_hr2(x, sA, c, 2);
          f16_cut_l(c, 0);
          f16_cut_l(c, 1);
          f16_cut_l(c, 2);
          f16_cut_l(c, 3);
          f16_cut_done(c, Ah[mi], Al[mi]);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int ni = 0; ni < 4; ++ni)
#pragma unroll
        for (int i = 0; i < 4; ++i) lds_landed(wb[ni][i]);
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) {
        const ou32x4 hw = {__float_as_uint(wb[ni][0].x), __float_as_uint(wb[ni][0].y), __float_as_uint(wb[ni][1].x),
                           __float_as_uint(wb[ni][1].y)};
        const ou32x4 lw = {__float_as_uint(wb[ni][2].x), __float_as_uint(wb[ni][2].y), __float_as_uint(wb[ni][3].x),
                           __float_as_uint(wb[ni][3].y)};
        Bh[ni] = __builtin_bit_cast(f16x8, hw);
        Bl[ni] = __builtin_bit_cast(f16x8, lw);
      }
    };
    auto mfmas = [&]() __attribute__((always_inline)) {
      // (the three products of a block eight MFMAs apart: a dependent MFMA issued behind its producer waits for it)
      if constexpr (LAB & 1) {
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
          for (int ni = 0; ni < 4; ++ni) {
            acc[mi][ni][0] += (float)Ah[mi][0] + (float)Bl[ni][1];
            acc[mi][ni][1] += (float)Al[mi][2] + (float)Bh[ni][3];
          }
        return;
      }
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Bl[ni], Ah[mi], acc[mi][ni], 0, 0, 0);
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Bh[ni], Al[mi], acc[mi][ni], 0, 0, 0);
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Bh[ni], Ah[mi], acc[mi][ni], 0, 0, 0);
    };
    // The MFMAs of k-step t - 1 stand between the issue of k-step t's row reads and the wait for them; the weight words
    // are requested behind the MFMAs (into the registers their fragments have just left: 256 VGPRs per wave do not hold
    // both), and the rows are cut while those arrive and the matrix pipe works through the 24 products.
    // (Two wave groups half a k-step apart -- one reading and cutting while the other issues its MFMAs -- were built and
    // measured: 165.7 us against 163.8 in phase; the phases of a k-step do not simply add up, see profiles/r05_os_lab.txt.
    // Four waves of 128 x 128 (one per SIMD, 64 KB of fragment reads per k-step instead of 96) were built too: hipcc spills
    // 208 bytes per lane at 256 VGPRs + 256 AGPRs in that form -- not run.)
    sync_top(0);
    reads_a(0);
    reads_b(0);
    frags_finish();
    for (int t = 1; t < T; ++t) {
      sync_top(t);
      reads_a(t);
      mfmas();
      __builtin_amdgcn_sched_barrier(0);
      reads_b(t);
      frags_finish();
    }
    mfmas();

    // ---- the panel's gradient: lane = batch row l31 of the sub-tile, registers = four runs of four columns ----
    float* const dA = L.dA;
    const int64_t ldda = L.ldda;
    const bool accum = L.accumulate != 0;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
      const int row = row0 + 64 * wr + 32 * mi + l31;
#pragma unroll
      for (int ni = 0; ni < 4; ++ni)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int col = 128 * wc + 32 * ni + 8 * g + 4 * h;
          if (row >= M || col >= K) continue;
          float4 x = make_float4(acc[mi][ni][4 * g] * inv, acc[mi][ni][4 * g + 1] * inv, acc[mi][ni][4 * g + 2] * inv,
                                 acc[mi][ni][4 * g + 3] * inv);
          float* const at = dA + (int64_t)row * ldda + col;
          if (accum) {
            const float4 o = *reinterpret_cast<const float4*>(at);
            x.x += o.x; x.y += o.y; x.z += o.z; x.w += o.w;
          }
          *reinterpret_cast<float4*>(at) = x;
          amax_acc(am, x);
        }
    }
    // (the next panel's first DMAs overwrite slots other waves may still read; its waits count only DMAs)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  if (L.amax_out) amax_flush(am, L.amax_out);
}

}  // namespace mml

using namespace mml;

// MML_OK: served; MML_ERR_UNSUPPORTED (no error text): not a launch of this kernel -- the caller goes on to the
// weight-stationary and the tile kernel.  MMLREC_GEMM_OS=0 (read on every call: a test switches it inside one process)
// turns the kernel off.
int mml_gemm_os_try_dgrad(const mml_gemm_dgrad_desc* d, int32_t n, hipStream_t st) {
  const char* e = getenv("MMLREC_GEMM_OS");
  if ((e && e[0] == '0') || n != 1) return MML_ERR_UNSUPPORTED;
  const mml_gemm_dgrad_desc& q = d[0];
  if (q.gate_h || q.Y || q.relu_mask || q.act != MML_ACT_NONE || !q.dA) return MML_ERR_UNSUPPORTED;
  if (q.n_src < 2 || q.n_src > MML_MAX_SRC) return MML_ERR_UNSUPPORTED;  // (one source: the weight-stationary kernel's)
  // (every workgroup computes OS_NC = 256 columns: a narrower gradient wastes the difference -- PLE's K = 128 launches
  // measured 1 % slower per step with this kernel than with the tile kernel, so they stay there)
  if (q.M < 16384 || q.K < 192 || q.K > OS_NC || q.K % 4 != 0) return MML_ERR_UNSUPPORTED;
  if (!aligned16(q.dA) || q.ldda % 4 != 0 || q.ldda < q.K) return MML_ERR_UNSUPPORTED;
  if ((int64_t)q.M * q.ldda >= (1ll << 40)) return MML_ERR_UNSUPPORTED;
  OsLaunch L{};
  int64_t ntot = 0;
  for (int s = 0; s < q.n_src; ++s) {
    if (q.w_kn[s] != 0 || !q.dC[s] || !q.w_planes[s] || !q.w_kexp[s] || !q.amax_dc[s]) return MML_ERR_UNSUPPORTED;
    if (q.w_kexp[s] != q.w_kexp[0]) return MML_ERR_UNSUPPORTED;  // (one group, one exponent)
    if (q.N[s] <= 0 || q.N[s] % 16 != 0) return MML_ERR_UNSUPPORTED;
    if (!aligned16(q.dC[s]) || q.lddc[s] % 4 != 0 || q.lddc[s] < q.N[s]) return MML_ERR_UNSUPPORTED;
    if (!aligned16(q.w_planes[s]) || q.ldw[s] % 4 != 0 || q.ldw[s] < q.K) return MML_ERR_UNSUPPORTED;
    OsSource& S = L.s[s];
    S.dC = q.dC[s];
    S.planes = q.w_planes[s];
    S.amax_dc = q.amax_dc[s];
    S.lddc = q.lddc[s];
    S.ldp = q.ldw[s];
    S.N = q.N[s];
    ntot += q.N[s];
  }
  if (ntot < 256) return MML_ERR_UNSUPPORTED;  // (short reductions: the ring would not fill)
  L.dA = q.dA;
  L.kexp = q.w_kexp[0];
  L.amax_out = q.amax_out;
  L.ldda = q.ldda;
  L.M = q.M;
  L.K = q.K;
  L.nsrc = q.n_src;
  L.accumulate = q.accumulate;
  L.ksteps = (int32_t)(ntot / 16);
  static int cus = 0;
  if (cus == 0) {
    int dev = 0, nn = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&nn, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || nn <= 0)
      nn = 256;
    cus = nn;
  }
  const int npanels = (int)cdiv(L.M, OS_BM);
  const dim3 grid((unsigned)(npanels < cus ? npanels : cus)), block(512);
#ifdef MML_OS_LAB  // lab build only (MMLREC_BUILD_OS_LAB=1 python -m ...build; tools/lab/os_lab.sh): garbage-result variants
  const char* lab = getenv("MMLREC_OS_LAB");
  switch (lab ? atoi(lab) : 0) {
    case 1: MML_LAUNCH(gemm_os_kernel<1>, grid, block, 0, st, L); break;
    case 2: MML_LAUNCH(gemm_os_kernel<2>, grid, block, 0, st, L); break;
    case 4: MML_LAUNCH(gemm_os_kernel<4>, grid, block, 0, st, L); break;
    case 8: MML_LAUNCH(gemm_os_kernel<8>, grid, block, 0, st, L); break;
    case 12: MML_LAUNCH(gemm_os_kernel<12>, grid, block, 0, st, L); break;
    case 14: MML_LAUNCH(gemm_os_kernel<14>, grid, block, 0, st, L); break;
    case 16: MML_LAUNCH(gemm_os_kernel<16>, grid, block, 0, st, L); break;
    case 32: MML_LAUNCH(gemm_os_kernel<32>, grid, block, 0, st, L); break;
    case 48: MML_LAUNCH(gemm_os_kernel<48>, grid, block, 0, st, L); break;
    case 62: MML_LAUNCH(gemm_os_kernel<62>, grid, block, 0, st, L); break;
    case 15: MML_LAUNCH(gemm_os_kernel<15>, grid, block, 0, st, L); break;
    default: MML_LAUNCH(gemm_os_kernel<0>, grid, block, 0, st, L); break;
  }
#else
  MML_LAUNCH(gemm_os_kernel<0>, grid, block, 0, st, L);
#endif
  return check_launch("mml_gemm_grouped_dgrad(os)");
}
