// Pieces shared by the LDS-DMA GEMM kernels (gemm.hip, gemm_planes.hip): asynchronous LDS accesses as inline asm,
// the global -> LDS DMA wave-instructions, the XCD-aware tile remap.
#pragma once
#include "common.hpp"

namespace mml {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));

// XCD-aware remap: blocks b and b+8 share an XCD (round-robin dispatch), so give each XCD a contiguous run of
// virtual tile ids: the n-tiles that re-read one 128-row A panel then hit the same 4 MiB L2.  Bijective for any
// grid size (cdna guide T1).
__device__ __forceinline__ int xcd_remap(int bid, int nblocks) {
  const int q = nblocks >> 3, r = nblocks & 7;
  const int xcd = bid & 7, slot = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
}

// LDS accesses as inline asm: hipcc treats an LDS-DMA in flight as a pending LDS store and would put
// s_waitcnt vmcnt(0) in front of every C++-level ds_read (draining the prefetch); asm accesses are invisible to that
// pass, so the counted vmcnt(N) of the kernel is the only VMEM wait in the loop.  lgkmcnt waits are placed by hand.
//
// The price: to the compiler an asm ds_read has produced its value at once, so it feels free to COPY the destination
// register before the data has landed (it did, to form register pairs for v_pk_add_f32).  Rule used throughout: a
// value read this way is kept in a native vector type exactly as the instruction wrote it, is not touched until the
// wait, and is passed through lds_landed() right after the wait; every real use hangs off lds_landed()'s output.
typedef __attribute__((address_space(3))) float lds_f32_t;
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t lds_byte_addr(const float* p) {
  return (uint32_t)(uintptr_t)(const lds_f32_t*)p;
}
template <int OFF>
__device__ __forceinline__ f32x4_t ds_read128(uint32_t addr) {
  f32x4_t v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
  return v;
}
__device__ __forceinline__ void ds_write128(uint32_t addr, const f32x4_t& v) {
  asm volatile("ds_write_b128 %0, %1" ::"v"(addr), "v"(v) : "memory");
}
template <int U0, int U1>
__device__ __forceinline__ f32x2_t ds_read2st64(uint32_t addr) {  // two dwords at addr + U0*256 B and addr + U1*256 B
  f32x2_t v;
  asm volatile("ds_read2st64_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(v) : "v"(addr), "n"(U0), "n"(U1) : "memory");
  return v;
}
__device__ __forceinline__ void lds_landed(f32x4_t& v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ void lds_landed(f32x2_t& v) { asm volatile("" : "+v"(v)); }

// one LDS-DMA wave-instruction: every lane moves 16 bytes from its own global address to lds_dst + 16 * lane
__device__ __forceinline__ void dma16(const float* g, float* lds_dst) {
  __builtin_amdgcn_global_load_lds(g, lds_dst, 16, 0, 0);
}
// ... and the 4-byte form: lds_dst + 4 * lane
__device__ __forceinline__ void dma4(const float* g, float* lds_dst) {
  __builtin_amdgcn_global_load_lds(g, lds_dst, 4, 0, 0);
}

// fp32 value -> PL bf16 planes by mantissa slicing (h = top 8 significant bits, m = the next 8, l = the last 8; every
// subtraction is exact, so h + m + l == x bit for bit when PL == 3).  Eight k-values of one operand row per lane,
// packed as the 32x32x16 bf16 MFMA wants them.
template <int PL>
__device__ __forceinline__ void split_planes(const float (&x)[8], bf16x8 (&out)[PL]) {
  uint32_t w[PL][4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const uint32_t u0 = __float_as_uint(x[2 * j]), u1 = __float_as_uint(x[2 * j + 1]);
    if (PL == 1) {  // plain bf16 operands: round to nearest even (v_cvt_pk_bf16_f32)
      typedef float f2 __attribute__((ext_vector_type(2)));
      const f2 xx = {x[2 * j], x[2 * j + 1]};
      w[0][j] = __builtin_bit_cast(uint32_t, __builtin_convertvector(xx, bf16x2_t));
      continue;
    }
    w[0][j] = __builtin_amdgcn_perm(u1, u0, 0x07060302u);  // {bf16 bits of x0, bf16 bits of x1}, truncated
    if (PL > 1) {
      const float r0 = x[2 * j] - __uint_as_float(u0 & 0xffff0000u);
      const float r1 = x[2 * j + 1] - __uint_as_float(u1 & 0xffff0000u);
      const uint32_t q0 = __float_as_uint(r0), q1 = __float_as_uint(r1);
      if (PL == 2) {
        typedef float f2 __attribute__((ext_vector_type(2)));
        f2 r = {r0, r1};
        w[1][j] = __builtin_bit_cast(uint32_t, __builtin_convertvector(r, bf16x2_t));  // RNE: 2^-17 |x| left over
      } else {
        w[1][j] = __builtin_amdgcn_perm(q1, q0, 0x07060302u);
        const float s0 = r0 - __uint_as_float(q0 & 0xffff0000u);
        const float s1 = r1 - __uint_as_float(q1 & 0xffff0000u);
        w[PL - 1][j] = __builtin_amdgcn_perm(__float_as_uint(s1), __float_as_uint(s0), 0x07060302u);
      }
    }
  }
#pragma unroll
  for (int p = 0; p < PL; ++p) {
    typedef uint32_t u4 __attribute__((ext_vector_type(4)));
    u4 t = {w[p][0], w[p][1], w[p][2], w[p][3]};
    out[p] = __builtin_bit_cast(bf16x8, t);
  }
}

// fp32 value -> TWO fp16 planes of the scaled value: h = rne16(x s), l = rne16(x s - h).  The residual is exact in fp32
// (h agrees with x s in its leading 11 bits), so h + l carries 22-23 significant bits of x s as long as l stays a
// normal fp16 number; s is a power of two chosen per operand tensor from its largest magnitude (see gemm.hip).
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ void split_f16_planes(const float (&x)[8], float s, f16x8& hi, f16x8& lo) {
  // Five VALU instructions per PAIR of values, the scale folded in (mixed-precision FMAs: fp32 sources, fp16 result /
  // an fp16 addend): h = rne16(x s) into the low / high half, r = x s - h exactly, l = rne16(r) packed.  Written as asm
  // because hipcc forms v_pk_mul_f32 + v_cvt_pk + 2 v_cvt_f32_f16 + v_pk_fma_f32 + v_cvt_pk from the C expression
  // (8 issue slots per pair: packed fp32 ops run at half rate), which made the kernel VALU-bound again.
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  uint32_t hw[4], lw[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    uint32_t h, l;
    float r0, r1;
    asm("v_fma_mixlo_f16 %0, %1, %2, 0 op_sel:[0,0,0] op_sel_hi:[0,0,0]" : "=v"(h) : "v"(x[2 * j]), "v"(s));
    asm("v_fma_mixhi_f16 %0, %1, %2, 0 op_sel:[0,0,0] op_sel_hi:[0,0,0]" : "+v"(h) : "v"(x[2 * j + 1]), "v"(s));
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

// The same cut, piecewise: the pipelined GEMM places the pieces by hand in the shadows of the three MFMAs of a product
// block (asm statements are invisible to sched_group_barrier, which arranges the bf16 forms)
struct F16Cut {
  uint32_t h[4], l[4];
  float r0[4], r1[4];
};
__device__ __forceinline__ void f16_cut_hr(const float (&x)[8], const float s, F16Cut& c, const int j) {
  asm("v_fma_mixlo_f16 %0, %1, %2, 0 op_sel:[0,0,0] op_sel_hi:[0,0,0]" : "=v"(c.h[j]) : "v"(x[2 * j]), "v"(s));
  asm("v_fma_mixhi_f16 %0, %1, %2, 0 op_sel:[0,0,0] op_sel_hi:[0,0,0]" : "+v"(c.h[j]) : "v"(x[2 * j + 1]), "v"(s));
  asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(c.r0[j]) : "v"(x[2 * j]), "v"(s), "v"(c.h[j]));
  asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
      : "=v"(c.r1[j])
      : "v"(x[2 * j + 1]), "v"(s), "v"(c.h[j]));
}
// Two pairs (j, j + 1) at once, the form the pipelined GEMM uses since round 3's instruction-rate measurements
// (tools/lab/micro/valu_rate.hip, mfma_mix.hip): v_fma_mixlo/hi_f16 issue at a quarter of the plain fp32 rate (7.1 ns
// per instruction pair of two waves against 2.3 for v_mul_f32 and 3.7 for v_cvt_pk_f16_f32) and do not overlap the
// MFMA pipe, so h = rne16(x s) is two multiplications and one packed conversion instead (x s is exact: s is a power of
// two); and the instructions of the two pairs alternate, so that a wave does not issue an instruction right behind the
// one it depends on (a SIMD holds two waves of this kernel: dependent issue is not hidden).
__device__ __forceinline__ void f16_cut_hr2(const float (&x)[8], const float s, F16Cut& c, const int j) {
  float y0, y1, y2, y3;
  asm("v_mul_f32 %0, %1, %2" : "=v"(y0) : "v"(x[2 * j]), "s"(s));
  asm("v_mul_f32 %0, %1, %2" : "=v"(y1) : "v"(x[2 * j + 1]), "s"(s));
  asm("v_mul_f32 %0, %1, %2" : "=v"(y2) : "v"(x[2 * j + 2]), "s"(s));
  asm("v_mul_f32 %0, %1, %2" : "=v"(y3) : "v"(x[2 * j + 3]), "s"(s));
  asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(c.h[j]) : "v"(y0), "v"(y1));
  asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(c.h[j + 1]) : "v"(y2), "v"(y3));
  asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(c.r0[j]) : "v"(x[2 * j]), "s"(s), "v"(c.h[j]));
  asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
      : "=v"(c.r1[j])
      : "v"(x[2 * j + 1]), "s"(s), "v"(c.h[j]));
  asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel:[0,0,0] op_sel_hi:[0,0,1]"
      : "=v"(c.r0[j + 1])
      : "v"(x[2 * j + 2]), "s"(s), "v"(c.h[j + 1]));
  asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
      : "=v"(c.r1[j + 1])
      : "v"(x[2 * j + 3]), "s"(s), "v"(c.h[j + 1]));
}
// The same two pairs in two pieces, for launches whose column operand arrives pre-cut (gemm.hip: BPL): the cut of the
// one remaining fragment is spread over the six MFMAs of TWO product blocks, 6 + 4 + 6 | 4 + 4 instructions -- about what
// one wave can issue in the shadow of an MFMA (one VALU instruction per ~6 clocks, tools/lab/micro/valu_rate.hip).
__device__ __forceinline__ void f16_cut_a2(const float (&x)[8], const float s, F16Cut& c, const int j) {
  float y0, y1, y2, y3;
  asm("v_mul_f32 %0, %1, %2" : "=v"(y0) : "v"(x[2 * j]), "s"(s));
  asm("v_mul_f32 %0, %1, %2" : "=v"(y1) : "v"(x[2 * j + 1]), "s"(s));
  asm("v_mul_f32 %0, %1, %2" : "=v"(y2) : "v"(x[2 * j + 2]), "s"(s));
  asm("v_mul_f32 %0, %1, %2" : "=v"(y3) : "v"(x[2 * j + 3]), "s"(s));
  asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(c.h[j]) : "v"(y0), "v"(y1));
  asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(c.h[j + 1]) : "v"(y2), "v"(y3));
}
__device__ __forceinline__ void f16_cut_b2(const float (&x)[8], const float s, F16Cut& c, const int j) {
  asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(c.r0[j]) : "v"(x[2 * j]), "s"(s), "v"(c.h[j]));
  asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
      : "=v"(c.r1[j])
      : "v"(x[2 * j + 1]), "s"(s), "v"(c.h[j]));
  asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel:[0,0,0] op_sel_hi:[0,0,1]"
      : "=v"(c.r0[j + 1])
      : "v"(x[2 * j + 2]), "s"(s), "v"(c.h[j + 1]));
  asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
      : "=v"(c.r1[j + 1])
      : "v"(x[2 * j + 3]), "s"(s), "v"(c.h[j + 1]));
}
__device__ __forceinline__ void f16_cut_l(F16Cut& c, const int j) {
  asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(c.l[j]) : "v"(c.r0[j]), "v"(c.r1[j]));
}
__device__ __forceinline__ void f16_cut_done(const F16Cut& c, f16x8& hi, f16x8& lo) {
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  const u32x4 a = {c.h[0], c.h[1], c.h[2], c.h[3]}, b = {c.l[0], c.l[1], c.l[2], c.l[3]};
  hi = __builtin_bit_cast(f16x8, a);
  lo = __builtin_bit_cast(f16x8, b);
}

}  // namespace mml
