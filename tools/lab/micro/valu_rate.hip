// Microbenchmark: issue cost of the VALU instructions a fp32 -> two-fp16-plane cut can be built from, in clocks per
// wave-instruction, with one and two waves per SIMD (8 independent chains per wave, so latency is hidden), alone and
// next to a stream of MFMAs from the same waves (does the instruction overlap with the matrix pipe or add to it?).
//   hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

#define OPS(X) \
  X(0, "v_mul_f32", "v_mul_f32 %0, %1, %2") \
  X(1, "v_fma_f32", "v_fma_f32 %0, %1, %2, %0") \
  X(2, "v_and_b32", "v_and_b32 %0, %1, %2") \
  X(3, "v_perm_b32", "v_perm_b32 %0, %1, %2, %0") \
  X(4, "v_fma_mixlo_f16", "v_fma_mixlo_f16 %0, %1, %2, 0 op_sel:[0,0,0] op_sel_hi:[0,0,0]") \
  X(5, "v_fma_mix_f32 (f16 addend)", "v_fma_mix_f32 %0, %1, %2, -%0 op_sel:[0,0,0] op_sel_hi:[0,0,1]") \
  X(6, "v_cvt_pk_f16_f32", "v_cvt_pk_f16_f32 %0, %1, %2") \
  X(7, "v_cvt_f32_f16", "v_cvt_f32_f16 %0, %1") \
  X(8, "v_cvt_f16_f32", "v_cvt_f16_f32 %0, %1") \
  X(9, "v_pk_mul_f32", "v_pk_mul_f32 %0, %1, %2") \
  X(10, "v_pk_fma_f32", "v_pk_fma_f32 %0, %1, %2, %0") \
  X(11, "v_sub_f32", "v_sub_f32 %0, %1, %2") \
  X(12, "v_pk_mul_f16", "v_pk_mul_f16 %0, %1, %2") \
  X(13, "v_cvt_pk_bf16_f32", "v_cvt_pk_bf16_f32 %0, %1, %2") \
  X(14, "v_lshlrev_b32", "v_lshlrev_b32 %0, 3, %1") \
  X(15, "v_add_u32", "v_add_u32 %0, %1, %2") \
  X(16, "v_bfi_b32", "v_bfi_b32 %0, %1, %2, %0") \
  X(17, "v_mov_b32", "v_mov_b32 %0, %1") \
  X(18, "v_pk_add_f16", "v_pk_add_f16 %0, %1, %2") \
  X(19, "v_mad_mix? v_fma_mix_f32 all f32", "v_fma_mix_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[0,0,0]") \
  X(20, "v_cvt_pkrtz_f16_f32", "v_cvt_pkrtz_f16_f32 %0, %1, %2")

template <int OP>
__device__ __forceinline__ void one(float& d, const float a, const float b) {
#define X(ID, NAME, ASM) if constexpr (OP == ID) { if constexpr (ID == 7 || ID == 8 || ID == 14 || ID == 17) asm volatile(ASM : "+v"(d) : "v"(a)); else asm volatile(ASM : "+v"(d) : "v"(a), "v"(b)); }
  OPS(X)
#undef X
}
template <>
__device__ __forceinline__ void one<9>(float& d, const float a, const float b) {}
template <>
__device__ __forceinline__ void one<10>(float& d, const float a, const float b) {}

typedef float f2 __attribute__((ext_vector_type(2)));
template <int OP>
__device__ __forceinline__ void one2(f2& d, const f2 a, const f2 b) {
  if constexpr (OP == 9) asm volatile("v_pk_mul_f32 %0, %1, %2" : "+v"(d) : "v"(a), "v"(b));
  if constexpr (OP == 10) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(d) : "v"(a), "v"(b));
}

// NV VALU instructions of kind OP after every MFMA (MF = 1) or alone (MF = 0)
template <int OP, int MF, int NV>
__global__ __launch_bounds__(256, 2) void k(int iters, float* out) {
  extern __shared__ float sm[];
  f32x16 acc[2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
  const u32x4 z = {0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u};
  const f16x8 H = __builtin_bit_cast(f16x8, z);
  float d[8];
  f2 d2[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { d[i] = (float)(threadIdx.x + i); d2[i] = f2{d[i], d[i] + 1.f}; }
  const float a = 1.0001f, b = 0.9999f;
  const f2 a2 = {a, b}, b2 = {b, a};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      if (MF) acc[m & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(H, H, acc[m & 1], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        if constexpr (OP == 9 || OP == 10) one2<OP>(d2[v & 7], a2, b2);
        else one<OP>(d[v & 7], a, b);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  float r = acc[0][0] + acc[1][3];
#pragma unroll
  for (int i = 0; i < 8; ++i) r += d[i] + d2[i].x + d2[i].y;
  if (r == 12345.678f) out[0] = r;
}

template <int OP, int MF, int NV>
float run(int wgs_per_cu, float* out) {
  const int iters = 2000;
  CK(hipFuncSetAttribute((const void*)k<OP, MF, NV>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  hipLaunchKernelGGL((k<OP, MF, NV>), dim3(256 * wgs_per_cu), dim3(256), 64 * 1024, 0, 10, out);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  hipLaunchKernelGGL((k<OP, MF, NV>), dim3(256 * wgs_per_cu), dim3(256), 64 * 1024, 0, iters, out);
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  return ms * 1e6f / (iters * 8);  // ns per (MFMA + NV VALU) group and wave-slot
}
template <int OP>
void op(const char* name, float* out) {
  constexpr int NV = 8;
  const float ghz = 2.4f;  // nominal; the MFMA-only line below calibrates
  for (int w : {1, 2}) {
    const float v = run<OP, 0, NV>(w, out), mv = run<OP, 1, NV>(w, out), m = run<OP, 1, 0>(w, out);
    printf("%-34s %d wave/SIMD: VALU alone %5.2f ns/instr | MFMA alone %6.1f ns | MFMA + %d VALU %6.1f ns  (sum %6.1f, max %6.1f)\n",
           name, w, v / NV, m, NV, mv, m + v, m > v ? m : v);
  }
  (void)ghz;
}
int main() {
  float* out; CK(hipMalloc(&out, 4));
#define X(ID, NAME, ASM) op<ID>(NAME, out);
  OPS(X)
#undef X
  return 0;
}
