// Internal interface between gate_head.hip (C-ABI entry points, generic kernels) and rows_fast.hip (aligned fast paths).
#pragma once
#include "common.hpp"

namespace mml {

struct GateFastAux {
  int32_t wg_off[MML_MAX_GATES];  // float offset of gate i's weight block (all gates, active or not)
  int32_t wg_total;
  int32_t lps, ne, ng, grid;
  int32_t ident;                  // every gate mixes experts 0..ne-1 in order (MMoE): backward reuses its expert-row loads
  int32_t hv;                     // 16-byte pieces of an expert / mixture row per lane: 1, or 2 (bf16 rows of 129..256 columns
                                  // on 32-lane groups -- two samples per wave and trip, 16-byte accesses; forward only)
  float* slab;                    // backward: [grid][wg_total] per-workgroup dWg partials
};

struct HeadFastAux {
  float* slab;                    // [grid][stride]
  int32_t stride, hmax, train;
  int32_t lps, nt, grid;
  int32_t gated;                  // some head reads Hin (.) gate (mml_head_desc.gate)
};

// return the lane-group width (16/32/64) when the fast path applies, 0 otherwise
int gate_fast_config(const mml_gate_group* g, bool bwd, GateFastAux& aux);
int head_fast_config(const mml_head_group* g, bool train, int hmax, HeadFastAux& aux);
// return 1 = not handled (caller falls back), MML_OK, or a negative error
int gate_fwd_fast(const mml_gate_group* g, hipStream_t st);
int gate_bwd_fast(const mml_gate_group* g, GateFastAux& aux, hipStream_t st);
// whether gate_bwd_fast takes the group (the reduction phase of mml_gate_mix_bwd_phase must find the layout phase 1 used)
bool gate_bwd_fast_serves(const mml_gate_group* g, const GateFastAux& aux);
int head_fast(const mml_head_group* g, const HeadFastAux& aux, hipStream_t st);

}  // namespace mml
