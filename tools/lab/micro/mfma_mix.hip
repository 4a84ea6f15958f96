// Microbenchmark: what the instruction mix of one k-step of gemm_pipe_kernel<.., EMU 2> costs when nothing else is in
// the way -- 12 MFMAs (four product blocks of three DEPENDENT v_mfma_f32_32x32x16_f16), the cut of four raw
// fragments (80 VALU: v_fma_mixlo/hi_f16, v_fma_mix_f32, v_cvt_pk_f16_f32) dealt 8 + 8 + 4 into the MFMA shadows as
// the kernel does, optionally the eight ds_read_b128 of the next step's fragments and the four LDS-DMA instructions
// of a later stage (L2-resident source) with the counted wait and the barrier.  Cycles per step and wave, with two
// workgroups of four waves per CU (two waves per SIMD) like the kernel, and with one.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_mix mfma_mix.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

struct Cut { uint32_t h[4], l[4]; float r0[4], r1[4]; };
__device__ __forceinline__ void cut_hr(const float (&x)[8], const float s, Cut& c, const int j) {
  asm volatile("v_fma_mixlo_f16 %0, %1, %2, 0 op_sel:[0,0,0] op_sel_hi:[0,0,0]" : "=v"(c.h[j]) : "v"(x[2 * j]), "v"(s));
  asm volatile("v_fma_mixhi_f16 %0, %1, %2, 0 op_sel:[0,0,0] op_sel_hi:[0,0,0]" : "+v"(c.h[j]) : "v"(x[2 * j + 1]), "v"(s));
  asm volatile("v_fma_mix_f32 %0, %1, %2, -%3 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(c.r0[j]) : "v"(x[2 * j]), "v"(s), "v"(c.h[j]));
  asm volatile("v_fma_mix_f32 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(c.r1[j]) : "v"(x[2 * j + 1]), "v"(s), "v"(c.h[j]));
}
// the same count of plain fp32 VALU work (what the cut would cost if the mixed-precision forms ran at another rate)
__device__ __forceinline__ void plain_hr(const float (&x)[8], const float s, Cut& c, const int j) {
  float a, b;
  asm volatile("v_mul_f32 %0, %1, %2" : "=v"(a) : "v"(x[2 * j]), "v"(s));
  asm volatile("v_mul_f32 %0, %1, %2" : "=v"(b) : "v"(x[2 * j + 1]), "v"(s));
  asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(c.r0[j]) : "v"(x[2 * j]), "v"(s), "v"(a));
  asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(c.r1[j]) : "v"(x[2 * j + 1]), "v"(s), "v"(b));
  c.h[j] = __float_as_uint(a) ^ __float_as_uint(b);
}
// the cut with the two half-rate v_fma_mixlo/hi_f16 replaced by two v_mul_f32 + one v_cvt_pk_f16_f32 (6 instead of 5
// instructions per pair, all of them full or near-full rate)
__device__ __forceinline__ void cut6_hr(const float (&x)[8], const float s, Cut& c, const int j) {
  float y0, y1;
  asm volatile("v_mul_f32 %0, %1, %2" : "=v"(y0) : "v"(x[2 * j]), "v"(s));
  asm volatile("v_mul_f32 %0, %1, %2" : "=v"(y1) : "v"(x[2 * j + 1]), "v"(s));
  asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(c.h[j]) : "v"(y0), "v"(y1));
  asm volatile("v_fma_mix_f32 %0, %1, %2, -%3 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(c.r0[j]) : "v"(x[2 * j]), "v"(s), "v"(c.h[j]));
  asm volatile("v_fma_mix_f32 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(c.r1[j]) : "v"(x[2 * j + 1]), "v"(s), "v"(c.h[j]));
}
// ... and with the two pairs of a shadow interleaved (distance 3-4 between dependent instructions instead of 1-2)
__device__ __forceinline__ void cut6i_hr2(const float (&x)[8], const float s, Cut& c, const int j) {
  float y0, y1, y2, y3;
  asm volatile("v_mul_f32 %0, %1, %2" : "=v"(y0) : "v"(x[2 * j]), "v"(s));
  asm volatile("v_mul_f32 %0, %1, %2" : "=v"(y1) : "v"(x[2 * j + 1]), "v"(s));
  asm volatile("v_mul_f32 %0, %1, %2" : "=v"(y2) : "v"(x[2 * j + 2]), "v"(s));
  asm volatile("v_mul_f32 %0, %1, %2" : "=v"(y3) : "v"(x[2 * j + 3]), "v"(s));
  asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(c.h[j]) : "v"(y0), "v"(y1));
  asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(c.h[j + 1]) : "v"(y2), "v"(y3));
  asm volatile("v_fma_mix_f32 %0, %1, %2, -%3 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(c.r0[j]) : "v"(x[2 * j]), "v"(s), "v"(c.h[j]));
  asm volatile("v_fma_mix_f32 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(c.r1[j]) : "v"(x[2 * j + 1]), "v"(s), "v"(c.h[j]));
  asm volatile("v_fma_mix_f32 %0, %1, %2, -%3 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(c.r0[j + 1]) : "v"(x[2 * j + 2]), "v"(s), "v"(c.h[j + 1]));
  asm volatile("v_fma_mix_f32 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(c.r1[j + 1]) : "v"(x[2 * j + 3]), "v"(s), "v"(c.h[j + 1]));
}
// the old mixed-precision forms, interleaved the same way
__device__ __forceinline__ void cuti_hr2(const float (&x)[8], const float s, Cut& c, const int j) {
  asm volatile("v_fma_mixlo_f16 %0, %1, %2, 0 op_sel:[0,0,0] op_sel_hi:[0,0,0]" : "=v"(c.h[j]) : "v"(x[2 * j]), "v"(s));
  asm volatile("v_fma_mixlo_f16 %0, %1, %2, 0 op_sel:[0,0,0] op_sel_hi:[0,0,0]" : "=v"(c.h[j + 1]) : "v"(x[2 * j + 2]), "v"(s));
  asm volatile("v_fma_mixhi_f16 %0, %1, %2, 0 op_sel:[0,0,0] op_sel_hi:[0,0,0]" : "+v"(c.h[j]) : "v"(x[2 * j + 1]), "v"(s));
  asm volatile("v_fma_mixhi_f16 %0, %1, %2, 0 op_sel:[0,0,0] op_sel_hi:[0,0,0]" : "+v"(c.h[j + 1]) : "v"(x[2 * j + 3]), "v"(s));
  asm volatile("v_fma_mix_f32 %0, %1, %2, -%3 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(c.r0[j]) : "v"(x[2 * j]), "v"(s), "v"(c.h[j]));
  asm volatile("v_fma_mix_f32 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(c.r1[j]) : "v"(x[2 * j + 1]), "v"(s), "v"(c.h[j]));
  asm volatile("v_fma_mix_f32 %0, %1, %2, -%3 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(c.r0[j + 1]) : "v"(x[2 * j + 2]), "v"(s), "v"(c.h[j + 1]));
  asm volatile("v_fma_mix_f32 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(c.r1[j + 1]) : "v"(x[2 * j + 3]), "v"(s), "v"(c.h[j + 1]));
}
__device__ __forceinline__ void cut_l(Cut& c, const int j) {
  asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(c.l[j]) : "v"(c.r0[j]), "v"(c.r1[j]));
}
__device__ __forceinline__ void plain_l(Cut& c, const int j) {
  float t;
  asm volatile("v_add_f32 %0, %1, %2" : "=v"(t) : "v"(c.r0[j]), "v"(c.r1[j]));
  c.l[j] = __float_as_uint(t);
}

// MODE bits: 1 MFMA, 2 cut (mixed-precision forms), 4 cut (plain fp32 forms, same count), 8 LDS fragment reads,
//            16 LDS-DMA + counted wait, 32 barrier per step, 64 the three MFMAs of a block on DIFFERENT accumulators,
//            128 cut with v_mul + v_cvt_pk instead of v_fma_mixlo/hi (6 instructions per pair)
template <int MODE>
__global__ __launch_bounds__(256, 2) void k(const float* __restrict__ src, int iters, float* out, unsigned long long* cyc) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  f32x16 acc[6];
#pragma unroll
  for (int a = 0; a < 6; ++a)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
  float x[4][8];
#pragma unroll
  for (int f = 0; f < 4; ++f)
#pragma unroll
    for (int e = 0; e < 8; ++e) x[f][e] = (float)(tid + f * 8 + e) * 1e-3f;
  f16x8 H[4], Lo[4];
#pragma unroll
  for (int f = 0; f < 4; ++f) {
    const u32x4 z = {0x3c003c00u + f, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u};
    H[f] = __builtin_bit_cast(f16x8, z);
    Lo[f] = __builtin_bit_cast(f16x8, z);
  }
  const float s = 1.5f;
  const uint32_t lds0 = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) float*)sm;
  const uint32_t ra = lds0 + (uint32_t)(lane & 31) * 64u + (uint32_t)(lane >> 5) * 16u + (uint32_t)(wave >> 1) * 4096u;
  const float* g = src + ((size_t)blockIdx.x * 4 + wave) * 4096 + lane * 4;
  f32x4_t q[8];
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    if (MODE & 16) {
      asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    }
    if (MODE & 32) __builtin_amdgcn_s_barrier();
    if (MODE & 16) {
      const int st = (it + 3) & 3;
#pragma unroll
      for (int d = 0; d < 4; ++d)
        __builtin_amdgcn_global_load_lds(g + ((it * 4 + d) & 3) * 256, sm + st * 4096 + (wave + 4 * d) * 256, 16, 0, 0);
    }
    if (MODE & 8) {
      const uint32_t a = ra + (uint32_t)((it + 1) & 3) * 16384u;
#pragma unroll
      for (int d = 0; d < 8; ++d)
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(q[d]) : "v"(a), "n"(0) : "memory");
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int blk = 0; blk < 4; ++blk) {
      if (blk == 2 && (MODE & 8)) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int d = 0; d < 8; ++d) asm volatile("" : "+v"(q[d]));
        // (the reads feed the next step's raw fragments)
        x[0][0] += q[0].x; x[1][0] += q[2].x; x[2][0] += q[4].x; x[3][0] += q[6].x;
        x[0][1] += q[1].x; x[1][1] += q[3].x; x[2][1] += q[5].x; x[3][1] += q[7].x;
      }
      Cut c;
      f32x16& a0 = acc[blk];
      f32x16& a1 = (MODE & 64) ? acc[4] : acc[blk];
      f32x16& a2 = (MODE & 64) ? acc[5] : acc[blk];
      const int fa = blk >> 1, fb = 2 + (blk & 1);
      __builtin_amdgcn_sched_barrier(0);
      if (MODE & 1) a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(Lo[fb], H[fa], a0, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (MODE & 2) { cut_hr(x[blk], s, c, 0); cut_hr(x[blk], s, c, 1); }
      if (MODE & 128) { cut6_hr(x[blk], s, c, 0); cut6_hr(x[blk], s, c, 1); }
      if (MODE & 256) cut6i_hr2(x[blk], s, c, 0);
      if (MODE & 512) cuti_hr2(x[blk], s, c, 0);
      if (MODE & 4) { plain_hr(x[blk], s, c, 0); plain_hr(x[blk], s, c, 1); }
      __builtin_amdgcn_sched_barrier(0);
      if (MODE & 1) a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(H[fb], Lo[fa], a1, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (MODE & 2) { cut_hr(x[blk], s, c, 2); cut_hr(x[blk], s, c, 3); }
      if (MODE & 128) { cut6_hr(x[blk], s, c, 2); cut6_hr(x[blk], s, c, 3); }
      if (MODE & 256) cut6i_hr2(x[blk], s, c, 2);
      if (MODE & 512) cuti_hr2(x[blk], s, c, 2);
      if (MODE & 4) { plain_hr(x[blk], s, c, 2); plain_hr(x[blk], s, c, 3); }
      __builtin_amdgcn_sched_barrier(0);
      if (MODE & 1) a2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(H[fb], H[fa], a2, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (MODE & (2 | 128 | 256 | 512)) { cut_l(c, 0); cut_l(c, 1); cut_l(c, 2); cut_l(c, 3); }
      if (MODE & 4) { plain_l(c, 0); plain_l(c, 1); plain_l(c, 2); plain_l(c, 3); }
      if (MODE & (6 | 128 | 256 | 512)) {
        const u32x4 hh = {c.h[0], c.h[1], c.h[2], c.h[3]}, ll = {c.l[0], c.l[1], c.l[2], c.l[3]};
        H[blk] = __builtin_bit_cast(f16x8, hh);
        Lo[blk] = __builtin_bit_cast(f16x8, ll);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  unsigned long long t1 = __builtin_readcyclecounter();
  if (tid == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
  float r = 0.f;
#pragma unroll
  for (int a = 0; a < 6; ++a) r += acc[a][0] + acc[a][7];
  r += (float)H[0][0] + (float)Lo[3][1];
  if (r == 12345.678f) out[0] = r;
}

template <int MODE>
void run(const float* src, int wgs_per_cu, const char* what, float* out, unsigned long long* cyc) {
  const int iters = 4000;
  CK(hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  hipLaunchKernelGGL((k<MODE>), dim3(256 * wgs_per_cu), dim3(256), 64 * 1024, 0, src, 10, out, cyc);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  hipLaunchKernelGGL((k<MODE>), dim3(256 * wgs_per_cu), dim3(256), 64 * 1024, 0, src, iters, out, cyc);
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  unsigned long long c; CK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
  printf("%-58s %d WG/CU: %7.1f ns per step  (counter: %6.0f ticks per step)\n", what, wgs_per_cu, ms * 1e6 / iters, (double)c / iters);
}
int main() {
  float* src; float* out; unsigned long long* cyc;
  CK(hipMalloc(&src, 512u * 4 * 4096 * 4 + 65536)); CK(hipMemset(src, 0, 512u * 4 * 4096 * 4 + 65536));
  CK(hipMalloc(&out, 4)); CK(hipMalloc(&cyc, 8));
  for (int w : {1, 2}) {
    run<1>(src, w, "12 MFMA (4 chains of 3 dependent)", out, cyc);
    run<1 | 64>(src, w, "12 MFMA (independent accumulators inside a block)", out, cyc);
    run<2>(src, w, "80 cut VALU (mixed-precision forms)", out, cyc);
    run<4>(src, w, "80 plain fp32 VALU", out, cyc);
    run<128>(src, w, "96 cut VALU (v_mul + v_cvt_pk form)", out, cyc);
    run<1 | 128>(src, w, "12 MFMA + cut (v_mul + v_cvt_pk form)", out, cyc);
    run<256>(src, w, "96 cut VALU (v_mul + v_cvt_pk form, interleaved pairs)", out, cyc);
    run<1 | 256>(src, w, "12 MFMA + cut (v_mul + v_cvt_pk, interleaved pairs)", out, cyc);
    run<512>(src, w, "80 cut VALU (mixed forms, interleaved pairs)", out, cyc);
    run<1 | 512>(src, w, "12 MFMA + cut (mixed forms, interleaved pairs)", out, cyc);
    run<1 | 256 | 8 | 16 | 32>(src, w, "12 MFMA + cut6i + reads + DMA + wait + barrier", out, cyc);
    run<1 | 128 | 8 | 16 | 32>(src, w, "12 MFMA + cut6 + reads + DMA + wait + barrier", out, cyc);
    run<1 | 2>(src, w, "12 MFMA + cut", out, cyc);
    run<1 | 4>(src, w, "12 MFMA + plain VALU", out, cyc);
    run<1 | 2 | 64>(src, w, "12 MFMA (indep.) + cut", out, cyc);
    run<1 | 2 | 8>(src, w, "12 MFMA + cut + 8 ds_read_b128", out, cyc);
    run<1 | 2 | 8 | 32>(src, w, "12 MFMA + cut + reads + barrier", out, cyc);
    run<1 | 2 | 8 | 16 | 32>(src, w, "12 MFMA + cut + reads + 4 LDS-DMA (L2) + wait + barrier", out, cyc);
    run<8 | 16 | 32>(src, w, "reads + 4 LDS-DMA + wait + barrier (no arithmetic)", out, cyc);
    run<1 | 8 | 16 | 32>(src, w, "12 MFMA + reads + DMA + wait + barrier (no cut)", out, cyc);
  }
  return 0;
}
