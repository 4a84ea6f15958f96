// K1 fused multi-field gather(+dense copy) and K2 sparse row-scatter backward.
//
// K1 restates BaseModel.input_from_feature_columns + combined_dnn_input of the reference
// (model/basemodel.py:461-487, model/utils.py:434-446) as ONE launch for all F fields: the output row
// dnn_input[b, :] is F*E contiguous floats, so consecutive lanes write consecutive 16-byte pieces (fully
// coalesced stores) while each lane pulls its piece of a table row with one 16-byte load.  HBM-bound.
//
// K2 restates aten::embedding_dense_backward (sparse=False, model/basemodel.py:122): row-granular float
// atomics into dense [V,E] accumulators, shaped as E contiguous floats per row per wave-instruction.
#include "common.hpp"

#include <stdlib.h>

#include <atomic>

namespace mml {

struct FieldTable {
  const float* tab[MML_MAX_FIELDS];
  int64_t vocab[MML_MAX_FIELDS];
  int32_t col[MML_MAX_FIELDS];
};

struct GatherArgs {
  const float* X;      // fp32-encoded indices (+dense values) or null
  const int32_t* idx;  // native indices or null
  const float* dense;  // dense values for the idx32 variant
  int64_t ldX, ldi, ldd;
  int32_t F, E, dense_col0, Nd;
  int64_t B;
  float* out;
  int64_t ldo;
  int32_t* status;
  uint8_t* marks;  // optional: a byte per table row (layout of mml_scatter_bwd's row_marks), set for every row read
  int64_t markbase[MML_MAX_FIELDS];
  float* wgmax;    // optional (mml_gather_fwd_wgmax): workgroup w stores the largest |value| it wrote to wgmax[w]
};

// Reference semantics of X[:, c].long(): truncation toward zero (model/basemodel.py:476).
__device__ __forceinline__ int64_t load_index(const GatherArgs& a, int64_t b, int f, const FieldTable& ft,
                                              int& bad) {
  int64_t i;
  if (a.idx) {
    i = a.idx[b * a.ldi + f];
  } else {
    float v = a.X[b * a.ldX + ft.col[f]];
    i = (int64_t)v;  // v_cvt: truncates toward zero
  }
  const int64_t V = ft.vocab[f];
  if (i < 0) {
    bad |= 1;
    i = 0;
  } else if (i >= V) {
    bad |= 2;
    i = V - 1;
  }
  return i;
}

// One thread = one 16-byte piece of the output row (E % 4 == 0): four values of an embedding row, or four dense
// features (the last piece of a row: Nd % 4 of them, stored as scalars -- the columns behind them are not this kernel's).
// (Until round 5 a dense feature was a thread of its own: AE's 63 dense columns, reference configs_msl/config_AE.json,
// doubled the threads of the launch and stored 4 bytes each -- 36 -> 66 us at B = 65 536.)
// ITEMS (index -> row -> store) chains per thread; launched with ITEMS = 1 (see launch_gather).
template <int ITEMS, bool WGMAX = false>
__global__ __launch_bounds__(256) void gather_vec4_kernel(const FieldTable ft, const GatherArgs a) {
  float am = 0.f;  // (WGMAX) the largest |value| this thread stored
  const int e4 = a.E >> 2;
  const int nvec = a.F * e4;             // 16-byte pieces per sample
  const int per_sample = nvec + ((a.Nd + 3) >> 2);  // + pieces of four dense features
  // A thread keeps ONE output column slot c and walks ITEMS consecutive samples: the (sample, slot) split costs one
  // 32-bit division per ITEMS items instead of a 64-bit one per item, the field / piece decode happens once per
  // thread, and neighbouring lanes still write neighbouring 16-byte pieces of the same row.
  const int64_t groups = (a.B + ITEMS - 1) / ITEMS;          // sample groups
  const int64_t total = groups * per_sample;                 // threads' work items
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  int bad = 0;
  for (int64_t base = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; base < total; base += stride) {
    int64_t grp;
    int c;
    if (total < 0x7fffffff) {
      const uint32_t q = (uint32_t)base / (uint32_t)per_sample;
      grp = q;
      c = (int)((uint32_t)base - q * (uint32_t)per_sample);
    } else {
      grp = base / per_sample;
      c = (int)(base - grp * per_sample);
    }
    const int64_t b0 = grp * ITEMS;
    if (c < nvec) {
      const int f = c / e4;
      const int part = c - f * e4;
      const float* tab = ft.tab[f] + part * 4;
      float4 v[ITEMS];
#pragma unroll
      for (int i = 0; i < ITEMS; ++i)
        if (b0 + i < a.B) {
          const int64_t row = load_index(a, b0 + i, f, ft, bad);
          v[i] = *reinterpret_cast<const float4*>(tab + row * a.E);
          // the split dense table update needs the batch's row set before its early pass starts: the gather has every
          // index in a register anyway (one byte store per lookup instead of a separate pass over X)
          if (a.marks && part == 0) a.marks[a.markbase[f] + row] = 1;
        }
#pragma unroll
      for (int i = 0; i < ITEMS; ++i)
        if (b0 + i < a.B) {
          *reinterpret_cast<float4*>(a.out + (b0 + i) * a.ldo + (int64_t)c * 4) = v[i];
          if (WGMAX) amax_acc(am, v[i]);
        }
    } else {
      const int j = 4 * (c - nvec);
      const int nj = a.Nd - j;  // >= 1
#pragma unroll
      for (int i = 0; i < ITEMS; ++i)
        if (b0 + i < a.B) {
          const int64_t b = b0 + i;
          const float* src = a.X ? a.X + b * a.ldX + a.dense_col0 + j : a.dense + b * a.ldd + j;  // (4-byte aligned only)
          float* dst = a.out + b * a.ldo + (int64_t)a.F * a.E + j;                                // (16-byte aligned)
          float4 d = make_float4(src[0], 0.f, 0.f, 0.f);
          if (nj > 1) d.y = src[1];
          if (nj > 2) d.z = src[2];
          if (nj > 3) d.w = src[3];
          if (nj > 3) {
            *reinterpret_cast<float4*>(dst) = d;
          } else {
            dst[0] = d.x;
            if (nj > 1) dst[1] = d.y;
            if (nj > 2) dst[2] = d.z;
          }
          if (WGMAX) amax_acc(am, d);
        }
    }
  }
  if (bad && a.status) atomicOr(a.status, bad);
  if (WGMAX) {
    // The magnitude of the gathered input without a pass over it and without atomics (15 000 workgroups raising one slot
    // cost +0.1 ms, common.hpp): ONE plain store per workgroup; the consumer's magnitude launch reads these 60 KB instead
    // of the 63 MB of the output.
    __shared__ float wmax[4];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) am = fmaxf(am, __shfl_xor(am, o, 64));
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = am;
    __syncthreads();
    if (threadIdx.x == 0) a.wgmax[blockIdx.x] = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
  }
}

// north_star's "LDS-staged index dedup", measured rather than argued: the tables with at most kLdsRows rows
// (AE-30: 6 x 100 + 2 rows = 19 KiB at E = 8) are copied into LDS once per PERSISTENT workgroup and their lookups served
// from there; the other fields read HBM / L2 as in gather_vec4_kernel.  Opt-in (MMLREC_GATHER_LDS=1); not the default:
// see DESIGN 10.14 for the numbers (the one-item-per-thread kernel above wins at every batch size).
constexpr int kLdsRows = 128;
constexpr int kLdsFloats = 12 * 1024;  // 48 KiB
struct SmallTabs {
  int32_t off[MML_MAX_FIELDS];  // float offset of field f's copy in LDS, or -1
  int32_t total;                // floats
};
__global__ __launch_bounds__(256) void gather_lds_kernel(const FieldTable ft, const GatherArgs a, const SmallTabs st) {
  __shared__ __attribute__((aligned(16))) float small[kLdsFloats];
  for (int f = 0; f < a.F; ++f) {
    if (st.off[f] < 0) continue;
    const int n4 = (int)(ft.vocab[f] * a.E) >> 2;
    for (int i = threadIdx.x; i < n4; i += 256)
      reinterpret_cast<float4*>(small + st.off[f])[i] = reinterpret_cast<const float4*>(ft.tab[f])[i];
  }
  __syncthreads();
  const int e4 = a.E >> 2;
  const int nvec = a.F * e4;
  const int per_sample = nvec + a.Nd;
  const int64_t total = a.B * per_sample;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  int bad = 0;
  for (int64_t base = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; base < total; base += stride) {
    const int64_t b = base / per_sample;
    const int c = (int)(base - b * per_sample);
    if (c < nvec) {
      const int f = c / e4;
      const int part = c - f * e4;
      const int64_t row = load_index(a, b, f, ft, bad);
      float4 v;
      if (st.off[f] >= 0) v = *reinterpret_cast<const float4*>(small + st.off[f] + row * a.E + part * 4);
      else v = *reinterpret_cast<const float4*>(ft.tab[f] + row * a.E + part * 4);
      *reinterpret_cast<float4*>(a.out + b * a.ldo + (int64_t)c * 4) = v;
    } else {
      const int j = c - nvec;
      a.out[b * a.ldo + (int64_t)a.F * a.E + j] = a.X ? a.X[b * a.ldX + a.dense_col0 + j] : a.dense[b * a.ldd + j];
    }
  }
  if (bad && a.status) atomicOr(a.status, bad);
}

// Generic path (E not a multiple of 4, or misaligned buffers): one thread per output float.
__global__ __launch_bounds__(256) void gather_scalar_kernel(const FieldTable ft, const GatherArgs a) {
  const int per_sample = a.F * a.E + a.Nd;
  const int64_t total = a.B * per_sample;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  int bad = 0;
  for (int64_t item = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; item < total; item += stride) {
    const int64_t b = item / per_sample;
    const int c = (int)(item - b * per_sample);
    float v;
    if (c < a.F * a.E) {
      const int f = c / a.E;
      const int e = c - f * a.E;
      const int64_t row = load_index(a, b, f, ft, bad);
      v = ft.tab[f][row * a.E + e];
      if (a.marks && e == 0) a.marks[a.markbase[f] + row] = 1;
    } else {
      const int j = c - a.F * a.E;
      v = a.X ? a.X[b * a.ldX + a.dense_col0 + j] : a.dense[b * a.ldd + j];
    }
    a.out[b * a.ldo + c] = v;
  }
  if (bad && a.status) atomicOr(a.status, bad);
}

static thread_local const char* g_last_gather = "";
extern "C" const char* mml_gather_last_kernel(void) { return g_last_gather; }

static int launch_gather(const FieldTable& ft, const GatherArgs& a, hipStream_t stream) {
  if (a.B == 0) return MML_OK;
  bool vec = (a.E % 4 == 0) && (a.ldo % 4 == 0) && aligned16(a.out);
  for (int f = 0; f < a.F && vec; ++f) vec = aligned16(ft.tab[f]);
  const int threads = 256;
  if (vec) {
    // ONE item per thread.  Round 1 gave a thread four consecutive samples ("four independent chains"); in the compiled
    // kernel the chains run one after the other (every index load is followed by its range checks), so a thread's
    // latency was four chains long and the grid four times smaller: 240 workgroups at B = 4 096.  Graph-replayed device
    // times on AE-30 (tools/bench_rows.py --graph, Zipf), four items -> one: 30.8 -> 4.1 us at B = 4 096, 35 -> 8 us at
    // 16 384, 35-46 -> 22.6 us at 65 536 (5.9 TB/s algorithmic), 541 -> 513 us at 1 048 576.  MMLREC_GATHER_ITEMS=4
    // brings the old form back (lab).
    static int items = -1;
    if (items < 0) {
      const char* e = getenv("MMLREC_GATHER_ITEMS");
      items = (e && atoi(e) == 4) ? 4 : 1;
    }
    // n > 0: the persistent LDS-staged variant with n workgroups per CU (measurement knob; read on every call so that a
    // test can switch it inside one process -- a getenv is nanoseconds next to a launch)
    const char* e_lds = getenv("MMLREC_GATHER_LDS");
    const int use_lds = e_lds ? atoi(e_lds) : 0;
    if (use_lds > 0 && use_lds <= 16 && !a.marks && !a.wgmax) {
      SmallTabs st{};
      int off = 0;
      for (int f = 0; f < a.F; ++f) {
        st.off[f] = -1;
        const int64_t fl = ft.vocab[f] * a.E;
        if (ft.vocab[f] <= kLdsRows && off + fl <= kLdsFloats) {
          st.off[f] = off;
          off += (int)((fl + 3) / 4 * 4);
        }
      }
      st.total = off;
      MML_LAUNCH(gather_lds_kernel, dim3(256u * (unsigned)use_lds), dim3(threads), 0, stream, ft, a, st);
      g_last_gather = "gather_lds_kernel";
      return check_launch("mml_gather_fwd(lds)");
    }
    const int64_t per_sample = (int64_t)a.F * (a.E / 4) + (a.Nd + 3) / 4;
    const int64_t total = cdiv(a.B, (int64_t)items) * per_sample;
    int64_t blocks = cdiv(total, (int64_t)threads);
    if (blocks > 0x7fffffff) blocks = 0x7fffffff;  // (grid-stride beyond that)
    if (a.wgmax) MML_LAUNCH((gather_vec4_kernel<1, true>), dim3((unsigned)blocks), dim3(threads), 0, stream, ft, a);
    else if (items == 4) MML_LAUNCH(gather_vec4_kernel<4>, dim3((unsigned)blocks), dim3(threads), 0, stream, ft, a);
    else MML_LAUNCH(gather_vec4_kernel<1>, dim3((unsigned)blocks), dim3(threads), 0, stream, ft, a);
    g_last_gather = "gather_vec4_kernel";
  } else {
    MML_REQUIRE(!a.wgmax, "mml_gather_fwd_wgmax: needs E %% 4 == 0, ldo %% 4 == 0 and 16-byte aligned tables and output");
    const int64_t total = a.B * ((int64_t)a.F * a.E + a.Nd);
    int64_t blocks = cdiv(total, threads);
    if (blocks > 256 * 32) blocks = 256 * 32;
    MML_LAUNCH(gather_scalar_kernel, dim3((unsigned)blocks), dim3(threads), 0, stream, ft, a);
    g_last_gather = "gather_scalar_kernel";
  }
  return check_launch("mml_gather_fwd");
}

// ------------------------------------------------------------------------------------------------
// K2 scatter
// ------------------------------------------------------------------------------------------------
struct ScatterArgs {
  float* gtab[MML_MAX_FIELDS];
  uint32_t* seen[MML_MAX_FIELDS];
  int64_t rowbase[MML_MAX_FIELDS];
  const float* X;
  int64_t ldX;
  const int32_t* idx;  // native indices idx[b * ldi + f] (used instead of X when non-null)
  int64_t ldi;
  int64_t B;
  const float* dOut;
  int64_t ldo;
  int32_t F, E;
  int32_t* touched;
  int32_t* touched_count;
  int32_t touched_cap;
  uint8_t* marks;  // optional byte per row (32-byte aligned per field, see mml_scatter_bwd): plain-store row marking
  int64_t markbase[MML_MAX_FIELDS];
  int32_t* status;
  // deterministic mode (mml_scatter_bwd_det): ONE fixed-point unit for the whole launch, taken from the magnitude slot
  // of dOut, and 64-bit integer accumulators per table row -- integer sums do not depend on the order of the addends
  long long* acc64[MML_MAX_FIELDS];
  const uint32_t* amax_dout;
  int32_t fix_shift;
};

// One lane = one gradient float: lanes run over e fastest, so every wave-instruction reads 256 contiguous
// bytes of dOut and issues its float atomics as E-float contiguous row segments.
__global__ __launch_bounds__(256) void scatter_atomic_kernel(const FieldTable ft, const ScatterArgs a) {
  const int FE = a.F * a.E;
  const int64_t total = a.B * FE;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  int bad = 0;
  for (int64_t item = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; item < total; item += stride) {
    const int64_t b = item / FE;
    const int c = (int)(item - b * FE);
    const int f = c / a.E;
    const int e = c - f * a.E;
    const int64_t row = a.idx ? (int64_t)a.idx[b * a.ldi + f] : (int64_t)a.X[b * a.ldX + ft.col[f]];
    const int64_t V = ft.vocab[f];
    if (row < 0) {
      bad |= 1;
      continue;
    }
    if (row >= V) {
      bad |= 2;
      continue;
    }
    const float g = a.dOut[b * a.ldo + c];
    atomicAdd(a.gtab[f] + row * a.E + e, g);
    if (a.touched && e == 0) {
      const uint32_t bit = 1u << (row & 31);
      const uint32_t old = atomicOr(a.seen[f] + (row >> 5), bit);
      if (!(old & bit)) {
        const int slot = atomicAdd(a.touched_count, 1);
        if (slot < a.touched_cap) a.touched[slot] = (int32_t)(a.rowbase[f] + row);
      }
    }
  }
  if (bad && a.status) atomicOr(a.status, bad);
}

// LDS-staged index dedup: one workgroup takes ONE field and a chunk of samples, folds duplicate rows in an
// insert-only open-addressing hash table held in LDS (keys claimed with ds_cmpst, gradients added with ds_add_f32),
// then flushes every occupied slot with E contiguous float atomics.  Hot rows of skewed / tiny-vocabulary fields
// reach HBM once per chunk instead of once per sample, which removes the same-address serialisation of the atomics.
template <int SLOTS>
__global__ __launch_bounds__(256) void scatter_hash_kernel(const FieldTable ft, const ScatterArgs a, int chunk) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  int* keys = reinterpret_cast<int*>(smem);
  float* acc = smem + SLOTS;
  const int E = a.E;
  // The F workgroups of one sample chunk read neighbouring 4E-byte pieces of the same dOut rows (and neighbouring X
  // columns): consecutive workgroup ids are dealt round-robin to the 8 XCDs, so the chunk is chosen from the XCD slot
  // (id % 8) and the field from the position inside the XCD -- the F readers of a 128-byte line then share ONE L2
  // instead of fetching the line into up to four of them.
  const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
  const int f = j % a.F;
  const int64_t c = (int64_t)(j / a.F) * 8 + xcd;
  if (c * chunk >= a.B) return;
  for (int i = threadIdx.x; i < SLOTS; i += 256) keys[i] = -1;
  for (int i = threadIdx.x; i < SLOTS * E; i += 256) acc[i] = 0.f;
  __syncthreads();
  const int64_t b0 = c * chunk;
  const int nb = (int)((a.B - b0 < chunk) ? (a.B - b0) : chunk);
  const int64_t V = ft.vocab[f];
  const int colf = ft.col[f];
  const bool POW2E = (E & (E - 1)) == 0 && E <= 64;  // the E lanes of a sample are an aligned lane group
  int bad = 0;
  for (int item = threadIdx.x; item < nb * E; item += 256) {
    const int s = item / E, e = item - s * E;
    const int64_t b = b0 + s;
    const int64_t row = a.idx ? (int64_t)a.idx[b * a.ldi + f] : (int64_t)a.X[b * a.ldX + colf];
    if (row < 0) { bad |= 1; continue; }
    if (row >= V) { bad |= 2; continue; }
    const float g = a.dOut ? a.dOut[b * a.ldo + f * E + e] : 0.f;
    unsigned slot = ((unsigned)row * 2654435761u) >> 16;
    slot &= (SLOTS - 1);
    const int key = (int)row;
    if (!POW2E || e == 0) {  // one lane per sample claims the slot ...
      while (true) {
        const int old = atomicCAS(&keys[slot], -1, key);
        if (old == -1 || old == key) break;
        slot = (slot + 1) & (SLOTS - 1);
      }
    }
    if (POW2E) slot = (unsigned)__shfl((int)slot, (int)(threadIdx.x & 63) & ~(E - 1));  // ... its E lanes follow
    if (a.dOut) atomicAdd(&acc[slot * E + e], g);
  }
  __syncthreads();
  float* gt = a.gtab[f];
  // flush: E contiguous float atomics per occupied slot.  Rows this launch sees for the first time (per-table `seen`
  // bitmap) are collected in an LDS list and appended to the global touched list with ONE counter atomic per
  // workgroup (a per-row atomic on the single counter serialises ~600 k requests at B = 65 536: 1.5 ms).
  __shared__ int n_new, base_out;
  int* newrows = reinterpret_cast<int*>(smem + SLOTS * (1 + E));  // [SLOTS], only allocated when a.touched != null
  if (threadIdx.x == 0) n_new = 0;
  __syncthreads();
  for (int item = threadIdx.x; item < SLOTS * E; item += 256) {
    const int slot = item / E, e = item - slot * E;
    const int key = keys[slot];
    if (key < 0) continue;
    if (a.dOut) atomicAdd(gt + (int64_t)key * E + e, acc[item]);
    if (a.touched && e == 0) {
      const uint32_t bit = 1u << (key & 31);
      const uint32_t old = atomicOr(a.seen[f] + (key >> 5), bit);
      if (!(old & bit)) newrows[atomicAdd(&n_new, 1)] = key;
    }
  }
  if (a.touched) {
    __syncthreads();
    if (threadIdx.x == 0) base_out = n_new ? atomicAdd(a.touched_count, n_new) : 0;
    __syncthreads();
    const int base = base_out;
    for (int i = threadIdx.x; i < n_new; i += 256)
      if (base + i < a.touched_cap) a.touched[base + i] = (int32_t)(a.rowbase[f] + newrows[i]);
  }
  if (bad && a.status) atomicOr(a.status, bad);
}


// ------------------------------------------------------------------------------------------------
// K2, restructured (round 2).  Same idea as scatter_hash_kernel -- fold the duplicate rows of a sample chunk in LDS,
// then E contiguous float atomics per distinct row -- rebuilt around what the MI355X measurements said
// (tools/lab/micro/lds_atomic.hip, per wave-instruction and CU):  ds_add_f32 = 195 cycles whatever the address
// pattern, ds_add_u64 = 15, ds_add_u32 = 9, ds_cmpst_rtn_b32 = 15; a same-address integer atomic costs ~4 cycles per
// lane.  The fp32 LDS atomics were 94 us of the old kernel's 168 us, the single-address "occupied slots" counter
// most of another 54 us.  So:
//   * the duplicates of a chunk are summed in 64-bit FIXED POINT with integer LDS atomics: every gradient value is
//     scaled by 2^(178 - emax), emax = the largest exponent the workgroup holds for this field, so the largest values
//     keep their 24 mantissa bits 28 bits above the unit and 11 bits of headroom are left for 2048 addends.  Values
//     down to 2^-28 of the largest are summed EXACTLY, smaller ones are rounded to 2^-52 of the largest -- the
//     chunk sum is at least as accurate as an fp32 accumulation of the same rows, and independent of the order;
//   * a lane owns one 16-byte piece of a gradient row (E/4 lanes per sample) and issues the index and gradient loads
//     of ALL its samples before the first LDS operation (the old loop chained 16 dependent load -> LDS steps);
//   * the occupied slots are listed with ONE counter atomic per wave (ballot), and the flush walks that list instead
//     of all SLOTS x E cells;
//   * tables with V <= SLOTS rows are direct-mapped (slot = row: no compare-and-swap, no probing) and one workgroup
//     folds kDirectReps chunks before it flushes.
// LDS accumulators are plane-major, acc[e][slot] with a (SLOTS + 1) pitch: the lanes of an insert differ in slot
// (random banks) and the lanes of a flush in e (consecutive banks).
// ------------------------------------------------------------------------------------------------
struct ScatterPlan {
  int32_t reps[MML_MAX_FIELDS];          // chunks folded per workgroup (direct-mapped fields: kDirectReps, others 1)
  int32_t grp_base[MML_MAX_FIELDS + 1];  // prefix sum of per-field group counts (per XCD slot)
};
constexpr int kDirectReps = 4;

__device__ __forceinline__ long long to_fixed(float x, int emax, const int shift = 28) {
  const unsigned u = __float_as_uint(x);
  int e = (int)((u >> 23) & 0xffu);
  if (e == 255) return 0;             // Inf / NaN: added to the table row directly (scatter_fold_kernel), not folded
  unsigned m = u & 0x7fffffu;
  if (e) m |= 0x800000u; else e = 1;  // subnormal
  const int sh = e - emax + shift;    // <= shift: x = m * 2^(e - 150) in units of 2^(emax - 150 - shift)
  long long v;
  if (sh >= 0) v = (long long)m << sh;
  else if (sh > -25) v = ((long long)m + (1ll << (-sh - 1))) >> (-sh);
  else v = 0;
  return (u >> 31) ? -v : v;
}

__device__ __forceinline__ float from_fixed(long long v, int emax, const int shift = 28) {
  return (float)ldexp((double)v, emax - 150 - shift);  // int64 -> f64 is exact below 2^53 and rounds once above; one rounding to f32
}

// NT threads per workgroup: the insert is a chain of LDS round trips (claim -> list -> shuffle -> add), so the kernel
// wants every wave slot of the CU: 2 workgroups x 1024 threads at 76 KiB of LDS each (256 threads: 98 us, 512: 89 us,
// 1024: 83 us on Zipf AE-30, B = 65 536)
// DET: deterministic mode -- the fixed-point unit comes from the launch-wide magnitude of dOut (a.amax_dout) instead of
// the workgroup's own maximum, and the flush adds the 64-bit chunk sums to 64-bit row accumulators (a.acc64) with INTEGER
// atomics: every row total is an exact integer sum, whatever the order -- bitwise repeatable, and independent of the
// order of the samples in the batch (scatter_det_finalize_kernel turns the totals into the fp32 gradient).
template <int SLOTS, int E, int NT, bool DET = false>
__global__ __launch_bounds__(NT) void scatter_fold_kernel(const FieldTable ft, const ScatterArgs a,
                                                           const ScatterPlan sp) {
  constexpr int LPS = E / 4;                   // lanes per sample
  constexpr int CHUNK = SLOTS / 2;             // samples per chunk (hash load factor <= 0.5)
  constexpr int PER_PASS = NT / LPS;
  static_assert(CHUNK % PER_PASS == 0 && CHUNK >= PER_PASS, "workgroup size does not divide the chunk");          // samples per pass of the workgroup
  constexpr int PASSES = CHUNK / PER_PASS;
  constexpr int PITCH = SLOTS + 1;
  extern __shared__ __attribute__((aligned(16))) long long smem64[];
  long long* acc = smem64;                                   // [E][PITCH] fixed-point sums
  int* keys = reinterpret_cast<int*>(acc + E * PITCH);       // [SLOTS]
  unsigned short* occ = reinterpret_cast<unsigned short*>(keys + SLOTS);  // [SLOTS] occupied slots, first-claim order
  __shared__ int n_occ;
  __shared__ unsigned mx_bits;
  if (blockIdx.x == 0 && threadIdx.x == 0 && a.touched) *a.touched_count = 0;  // (the compaction that follows appends)
  // chunk group from the XCD slot, field from the position inside the XCD (see scatter_hash_kernel)
  const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
  int f = 0;
  while (f + 1 < a.F && j >= sp.grp_base[f + 1]) ++f;
  const int q = j - sp.grp_base[f];
  const int reps = sp.reps[f];
  const int64_t V = ft.vocab[f];
  const bool direct = reps > 1;
  const int64_t nchunks = (a.B + CHUNK - 1) / CHUNK;
  if ((int64_t)xcd + 8 * (int64_t)q * reps >= nchunks) return;
  const int colf = ft.col[f];
  const int part = threadIdx.x % LPS;
  const int s_in = threadIdx.x / LPS;
  const int lane = threadIdx.x & 63;
  int bad = 0;
  // ---- every load of this lane in flight before the first LDS operation
  int rows[kDirectReps][PASSES];
  float4 g[kDirectReps][PASSES];
  unsigned mx = 0;
#pragma unroll
  for (int r = 0; r < kDirectReps; ++r) {
    const int64_t c = xcd + 8 * ((int64_t)q * reps + r);
#pragma unroll
    for (int k = 0; k < PASSES; ++k) {
      rows[r][k] = -1;
      g[r][k] = make_float4(0.f, 0.f, 0.f, 0.f);
      const int64_t b = c * CHUNK + k * PER_PASS + s_in;
      if (r < reps && c < nchunks && b < a.B) {
        // (the gradient load must not wait for the index: both are issued back to back, validity is applied later)
        if (a.dOut) g[r][k] = *reinterpret_cast<const float4*>(a.dOut + b * a.ldo + f * E + part * 4);
        const int64_t row = a.idx ? (int64_t)a.idx[b * a.ldi + f] : (int64_t)a.X[b * a.ldX + colf];
        if (row < 0) bad |= 1;
        else if (row >= V) bad |= 2;
        else rows[r][k] = (int)row;
      }
    }
  }
  if (a.dOut)
    for (int i = threadIdx.x; i < E * PITCH / 2; i += NT)
      *reinterpret_cast<float4*>(acc + i * 2) = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int i = threadIdx.x; i < SLOTS; i += NT) keys[i] = -1;
  if (threadIdx.x == 0) { n_occ = 0; mx_bits = 0; }
  __syncthreads();
  if (DET) {
    if (threadIdx.x == 0) {
      uint32_t m = 0;
      for (int i = 0; i < MML_AMAX_WORDS; ++i) m = a.amax_dout[i] > m ? a.amax_dout[i] : m;
      mx_bits = m;
    }
    __syncthreads();
  } else if (a.dOut) {  // largest magnitude (as bit pattern) of the workgroup's gradient values -> the fixed-point scale
#pragma unroll
    for (int r = 0; r < kDirectReps; ++r)
#pragma unroll
      for (int k = 0; k < PASSES; ++k) {
        const unsigned m0 = __float_as_uint(g[r][k].x) & 0x7fffffffu, m1 = __float_as_uint(g[r][k].y) & 0x7fffffffu;
        const unsigned m2 = __float_as_uint(g[r][k].z) & 0x7fffffffu, m3 = __float_as_uint(g[r][k].w) & 0x7fffffffu;
        mx = max(mx, max(max(m0, m1), max(m2, m3)));
      }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = max(mx, (unsigned)__shfl_xor((int)mx, o));
    if (lane == 0) atomicMax(&mx_bits, mx);
    __syncthreads();
  }
  int emax = (int)(mx_bits >> 23);
  // A diverged backward (Inf / NaN in dOut) must reach the table like it does in the reference (index_add of a NaN is
  // NaN), not be turned into a finite fixed-point value: such elements skip the fold and go to HBM as float atomics.
  const bool nonfinite = emax >= 255;  // workgroup-uniform
  emax = emax < 1 ? 1 : (emax > 254 ? 254 : emax);
  const int shift = DET ? a.fix_shift : 28;
  // ---- insert
#pragma unroll
  for (int r = 0; r < kDirectReps; ++r) {
    if (r >= reps) break;
#pragma unroll
    for (int k = 0; k < PASSES; ++k) {
      const int key = rows[r][k];
      unsigned slot = 0;
      bool is_new = false;
      if (direct) {
        if (key >= 0) {
          slot = (unsigned)key;
          if (part == 0 && keys[slot] != key) keys[slot] = key;  // racing writers store the same value
        }
      } else {
        if (key >= 0 && part == 0) {  // one lane per sample claims the slot ...
          slot = (((unsigned)key * 2654435761u) >> 16) & (SLOTS - 1);
          while (true) {
            const int old = atomicCAS(&keys[slot], -1, key);
            if (old == -1) { is_new = true; break; }
            if (old == key) break;
            slot = (slot + 1) & (SLOTS - 1);
          }
        }
        // new slots of this wave join the occupied list with ONE counter atomic
        const unsigned long long nm = __ballot(is_new);
        if (nm) {
          const int leader = __ffsll((long long)nm) - 1;
          int base = 0;
          if (lane == leader) base = atomicAdd(&n_occ, __popcll(nm));
          base = __shfl(base, leader);
          if (is_new) occ[base + __popcll(nm & ((1ull << lane) - 1ull))] = (unsigned short)slot;
        }
        if (LPS > 1) slot = (unsigned)__shfl((int)slot, lane & ~(LPS - 1));  // ... its lanes follow
      }
      if (a.dOut && key >= 0) {
        unsigned long long* p = reinterpret_cast<unsigned long long*>(acc) + (part * 4) * PITCH + slot;
        atomicAdd(p, (unsigned long long)to_fixed(g[r][k].x, emax, shift));
        atomicAdd(p + PITCH, (unsigned long long)to_fixed(g[r][k].y, emax, shift));
        atomicAdd(p + 2 * PITCH, (unsigned long long)to_fixed(g[r][k].z, emax, shift));
        atomicAdd(p + 3 * PITCH, (unsigned long long)to_fixed(g[r][k].w, emax, shift));
        if (nonfinite) {
          float* dst = a.gtab[f] + (int64_t)key * E + part * 4;
          const float gv[4] = {g[r][k].x, g[r][k].y, g[r][k].z, g[r][k].w};
#pragma unroll
          for (int i = 0; i < 4; ++i)
            if ((__float_as_uint(gv[i]) & 0x7f800000u) == 0x7f800000u) atomicAdd(dst + i, gv[i]);
        }
      }
    }
  }
  __syncthreads();
  float* gt = a.gtab[f];
  const int n_items = (direct ? (int)V : n_occ) * E;
  for (int item = threadIdx.x; item < n_items; item += NT) {
    const int i = item / E, e = item - i * E;
    const int slot = direct ? i : (int)occ[i];
    const int key = keys[slot];
    if (key < 0) continue;  // (direct-mapped: a row no sample of these chunks touched)
    if (DET) {
      const long long v = acc[e * PITCH + slot];
      if (v) atomicAdd(reinterpret_cast<unsigned long long*>(a.acc64[f]) + (int64_t)key * E + e, (unsigned long long)v);
    } else if (a.dOut) {
      atomicAdd(gt + (int64_t)key * E + e, from_fixed(acc[e * PITCH + slot], emax));
    }
    // touched-row bookkeeping: only MARK the row here (a non-returning atomic: they run at the float-atomic request
    // rate, while the returning form that told "first time seen" cost the index-only pass 0.5 ms); the list is built
    // from the bitmaps by rows_compact_kernel
    if (e == 0) {
      if (a.marks) a.marks[a.markbase[f] + key] = 1;  // plain store: hot rows cost nothing (see mark_rows_kernel)
      else if (a.touched) atomicOr(a.seen[f] + (key >> 5), 1u << (key & 31));
    }
  }
  if (bad && a.status) atomicOr(a.status, bad);
}

// Index-only pass (mml_index_unique) with a mark map: one byte store per lookup, no LDS, no atomics.  Hot rows are
// written by thousands of lanes -- plain same-value stores merge in L2, where same-address atomics serialise at the
// memory side (the atomicOr form of this pass took 350 us at B = 65 536, 15 tiny tables hammering a few words each).
__global__ __launch_bounds__(256) void mark_rows_kernel(const FieldTable ft, const ScatterArgs a) {
  if (blockIdx.x == 0 && threadIdx.x == 0 && a.touched_count) *a.touched_count = 0;  // (the compaction that follows appends)
  const int64_t total = a.B * a.F;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  int bad = 0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int64_t b = i / a.F;
    const int f = (int)(i - b * a.F);
    const int64_t row = a.idx ? (int64_t)a.idx[b * a.ldi + f] : (int64_t)a.X[b * a.ldX + ft.col[f]];
    if (row < 0) bad |= 1;
    else if (row >= ft.vocab[f]) bad |= 2;
    else a.marks[a.markbase[f] + row] = 1;
  }
  if (bad && a.status) atomicOr(a.status, bad);
}

// touched list := every row whose bit is set in the per-field `seen` bitmaps (global row id rowbase[f] + row),
// *count := their number.  One word per lane, one counter atomic per wave; 12.49 M rows = 1.56 MB of bitmap.
struct CompactArgs {
  uint32_t* seen[MML_MAX_FIELDS];
  uint8_t* marks;  // or null: the bitmaps already hold the marks
  int64_t wordbase[MML_MAX_FIELDS + 1];
  int64_t rowbase[MML_MAX_FIELDS];
  int64_t vocab[MML_MAX_FIELDS];
  int32_t blk0[MML_MAX_FIELDS + 1];  // first workgroup of field f in the 1-D grid
  int32_t F;
  int32_t* touched;
  int32_t* count;
  int32_t cap;
};

__global__ __launch_bounds__(1024) void rows_compact_kernel(const CompactArgs a) {
  // One field per workgroup (uniform: the per-field pointers and bases stay in scalar registers; a per-lane search of
  // the field tables made this pass 70 us).  The grid is 1-D with exactly the workgroups each field needs: as a
  // (longest field, F) rectangle it launched 9 180 workgroups of 1 024 threads on AE-30, 95 % of them for nothing --
  // 16 us of a 300 us small-batch step.
  int f = 0;
  while (f + 1 < a.F && (int)blockIdx.x >= a.blk0[f + 1]) ++f;
  const int bx = (int)blockIdx.x - a.blk0[f];
  const int64_t W = a.wordbase[f + 1] - a.wordbase[f];
  const int64_t stride = (int64_t)(a.blk0[f + 1] - a.blk0[f]) * blockDim.x;
  const int lane = threadIdx.x & 63;
  uint32_t* seen = a.seen[f];
  uint8_t* marks = a.marks ? a.marks + a.wordbase[f] * 32 : nullptr;
  const int64_t rowbase = a.rowbase[f];
  for (int64_t w0 = (int64_t)bx * blockDim.x; w0 < W; w0 += stride) {  // (whole waves stay in the loop)
    const int64_t wi = w0 + threadIdx.x;
    uint32_t bits = 0;
    if (wi < W) {
      if (marks) {  // 32 mark bytes -> one bitmap word (this lane is the word's only writer), marks cleared
        uint4* m = reinterpret_cast<uint4*>(marks + wi * 32);
        const uint4 lo = m[0], hi = m[1];
        if (lo.x | lo.y | lo.z | lo.w | hi.x | hi.y | hi.z | hi.w) {
          const uint32_t q[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
#pragma unroll
          for (int k = 0; k < 8; ++k)
#pragma unroll
            for (int j = 0; j < 4; ++j)
              if ((q[k] >> (8 * j)) & 0xffu) bits |= 1u << (4 * k + j);
          m[0] = make_uint4(0, 0, 0, 0);
          m[1] = make_uint4(0, 0, 0, 0);
          bits |= seen[wi];
          seen[wi] = bits;
        } else {
          bits = seen[wi];
        }
      } else {
        bits = seen[wi];
      }
    }
    const int n = __popc(bits);
    int pre = n;  // inclusive prefix sum over the wave
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int t = __shfl_up(pre, o);
      if (lane >= o) pre += t;
    }
    // ... over the workgroup, then ONE counter atomic per 1024 words (a returning atomic on a single address runs at
    // ~88 per microsecond chip-wide: one per wave made this pass 70 us)
    __shared__ int wsum[16], wbase[16], gbase;
    const int wv = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 63) wsum[wv] = pre;
    __syncthreads();
    if (threadIdx.x == 0) {
      int run = 0;
      for (int k = 0; k < 16; ++k) { wbase[k] = run; run += wsum[k]; }
      gbase = run ? atomicAdd(a.count, run) : 0;
    }
    __syncthreads();
    int pos = gbase + wbase[wv] + pre - n;
    while (bits) {
      const int b = __ffs((int)bits) - 1;
      bits &= bits - 1;
      if (pos < a.cap) a.touched[pos] = (int32_t)(rowbase + wi * 32 + b);
      ++pos;
    }
  }
}

// count_zeroed: the kernel launched just before (mark_rows_kernel, scatter_fold_kernel) has already reset *touched_count
// -- one fill launch less per step, ~5 us of a 300 us small-batch step
static int launch_compact(const FieldTable& ft, const ScatterArgs& a, hipStream_t stream, const char* who,
                          bool count_zeroed = false) {
  CompactArgs c{};
  int64_t words = 0;
  for (int f = 0; f < a.F; ++f) {
    c.seen[f] = a.seen[f];
    c.wordbase[f] = words;  // (= a.markbase[f] / 32)
    c.rowbase[f] = a.rowbase[f];
    words += (ft.vocab[f] + 31) / 32;
  }
  c.wordbase[a.F] = words;
  c.marks = a.marks;
  c.F = a.F; c.touched = a.touched; c.count = a.touched_count; c.cap = a.touched_cap;
  if (!count_zeroed) {
    hipError_t e = hipMemsetAsync(a.touched_count, 0, sizeof(int32_t), stream);
    if (e != hipSuccess) {
      set_error("%s: hipMemsetAsync: %s", who, hipGetErrorString(e));
      return MML_ERR_HIP;
    }
  }
  int total = 0;
  for (int f = 0; f < a.F; ++f) {
    int64_t nb = cdiv(c.wordbase[f + 1] - c.wordbase[f], 1024);
    if (nb > 256 * 4) nb = 256 * 4;
    if (nb < 1) nb = 1;
    c.blk0[f] = total;
    total += (int)nb;
  }
  c.blk0[a.F] = total;
  MML_LAUNCH(rows_compact_kernel, dim3((unsigned)total), dim3(1024), 0, stream, c);
  return check_launch(who);
}

template <int SLOTS, int E, int NT, bool DET = false>
static int launch_fold(const FieldTable& ft, const ScatterArgs& a, hipStream_t stream, const char* who) {
  constexpr int CHUNK = SLOTS / 2;
  ScatterPlan sp{};
  const int64_t nchunks = cdiv(a.B, (int64_t)CHUNK);
  const int64_t per_xcd = cdiv(nchunks, 8);
  int64_t total = 0;
  for (int f = 0; f < a.F; ++f) {
    sp.reps[f] = (ft.vocab[f] <= SLOTS) ? kDirectReps : 1;
    sp.grp_base[f] = (int32_t)total;
    total += cdiv(per_xcd, sp.reps[f]);
  }
  sp.grp_base[a.F] = (int32_t)total;
  if (total * 8 > 0x7fffffff) return 1;  // caller falls back
  const size_t lds = (size_t)E * (SLOTS + 1) * 8 + (size_t)SLOTS * 4 + (size_t)SLOTS * 2;
  // more than the 64 KiB a kernel may use by default; the attribute is per DEVICE (a process may drive several)
  static std::atomic<uint64_t> attr_set{0};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev < 0 || dev >= 64 || !((attr_set.load(std::memory_order_relaxed) >> dev) & 1ull)) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&scatter_fold_kernel<SLOTS, E, NT, DET>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    if (e != hipSuccess) {
      set_error("%s: hipFuncSetAttribute: %s", who, hipGetErrorString(e));
      return MML_ERR_HIP;
    }
    if (dev >= 0 && dev < 64) attr_set.fetch_or(1ull << dev, std::memory_order_relaxed);
  }
  MML_LAUNCH((scatter_fold_kernel<SLOTS, E, NT, DET>), dim3((unsigned)(total * 8)), dim3(NT), lds, stream, ft, a, sp);
  int rc = check_launch(who);
  if (rc || !a.touched) return rc;
  return launch_compact(ft, a, stream, who, true);
}

// E in {4, 8, 16}, 16-byte aligned gradient rows: the restructured kernel; returns 1 when the shape is not covered
static int try_fold(const FieldTable& ft, const ScatterArgs& a, hipStream_t stream, const char* who) {
  if (a.dOut && (a.ldo % 4 != 0 || !aligned16(a.dOut))) return 1;
  for (int f = 0; f < a.F; ++f)
    if (ft.vocab[f] > 0x7fffffff) return 1;
  if (a.E == 8) return launch_fold<1024, 8, 1024>(ft, a, stream, who);
  if (a.E == 4) return launch_fold<1024, 4, 512>(ft, a, stream, who);
  if (a.E == 16) return launch_fold<512, 16, 1024>(ft, a, stream, who);
  return 1;
}

// Deterministic mode, second launch: every marked row's 64-bit totals -> fp32, added to the gradient accumulator (one
// thread owns a row: no atomics), the totals zeroed again.  One field per workgroup range like rows_compact_kernel.
struct DetFinalArgs {
  float* gtab[MML_MAX_FIELDS];
  long long* acc64[MML_MAX_FIELDS];
  int64_t vocab[MML_MAX_FIELDS];
  int64_t markbase[MML_MAX_FIELDS];
  int32_t blk0[MML_MAX_FIELDS + 1];
  uint8_t* marks;
  const uint32_t* amax_dout;
  int32_t F, E, fix_shift, clear_marks;
};
__global__ __launch_bounds__(256) void scatter_det_finalize_kernel(const DetFinalArgs a) {
  int f = 0;
  while (f + 1 < a.F && (int)blockIdx.x >= a.blk0[f + 1]) ++f;
  const int bx = (int)blockIdx.x - a.blk0[f], nb = a.blk0[f + 1] - a.blk0[f];
  uint32_t m = 0;
  for (int i = 0; i < MML_AMAX_WORDS; ++i) m = a.amax_dout[i] > m ? a.amax_dout[i] : m;
  int emax = (int)(m >> 23);
  emax = emax < 1 ? 1 : (emax > 254 ? 254 : emax);  // (as scatter_fold_kernel clamps it; Inf / NaN went to the table directly)
  uint8_t* marks = a.marks + a.markbase[f];
  for (int64_t row = (int64_t)bx * 256 + threadIdx.x; row < a.vocab[f]; row += (int64_t)nb * 256) {
    if (!marks[row]) continue;
    long long* src = a.acc64[f] + row * a.E;
    float* dst = a.gtab[f] + row * a.E;
    for (int e = 0; e < a.E; ++e) {
      const long long v = src[e];
      if (v) {
        dst[e] += from_fixed(v, emax, a.fix_shift);
        src[e] = 0;
      }
    }
    if (a.clear_marks) marks[row] = 0;
  }
}

}  // namespace mml

using namespace mml;

static int fill_fields(FieldTable& ft, const float* const* tables, const int64_t* vocab, const int32_t* col,
                       int32_t F, const char* who) {
  MML_REQUIRE(F >= 0 && F <= MML_MAX_FIELDS, "%s: F=%d outside [0,%d]", who, F, MML_MAX_FIELDS);
  MML_REQUIRE(F == 0 || (tables && vocab), "%s: null tables/vocab", who);
  for (int f = 0; f < F; ++f) {
    MML_REQUIRE(tables[f] != nullptr, "%s: table %d is null", who, f);
    MML_REQUIRE(vocab[f] > 0, "%s: vocab[%d]=%lld", who, f, (long long)vocab[f]);
    ft.tab[f] = tables[f];
    ft.vocab[f] = vocab[f];
    ft.col[f] = col ? col[f] : f;
  }
  return MML_OK;
}

extern "C" int mml_gather_fwd(const float* const* tables, const int64_t* vocab, const int32_t* col, int32_t F,
                              int32_t E, const float* X, int64_t ldX, int32_t dense_col0, int32_t Nd, int64_t B,
                              float* out, int64_t ldo, int32_t* status, mml_stream_t stream) {
  FieldTable ft;
  int rc = fill_fields(ft, tables, vocab, col, F, "mml_gather_fwd");
  if (rc) return rc;
  MML_REQUIRE(B >= 0 && E > 0 && Nd >= 0, "mml_gather_fwd: bad sizes B=%lld E=%d Nd=%d", (long long)B, E, Nd);
  MML_REQUIRE(B == 0 || (X && out), "mml_gather_fwd: null X/out");
  MML_REQUIRE(ldo >= (int64_t)F * E + Nd, "mml_gather_fwd: ldo=%lld < F*E+Nd", (long long)ldo);
  GatherArgs a{};
  a.X = X; a.ldX = ldX; a.F = F; a.E = E; a.dense_col0 = dense_col0; a.Nd = Nd; a.B = B;
  a.out = out; a.ldo = ldo; a.status = status;
  return launch_gather(ft, a, to_stream(stream));
}

// number of workgroups (= partial maxima) mml_gather_fwd_wgmax writes for these sizes
extern "C" int64_t mml_gather_wgmax_len(int32_t F, int32_t E, int32_t Nd, int64_t B) {
  if (F <= 0 || E <= 0 || E % 4 != 0 || Nd < 0 || B <= 0) return 0;
  const int64_t per_sample = (int64_t)F * (E / 4) + (Nd + 3) / 4;  // (threads per sample of gather_vec4_kernel)
  const int64_t blocks = cdiv(B * per_sample, (int64_t)256);
  return blocks > 0x7fffffff ? 0 : blocks;
}

extern "C" int mml_gather_fwd_wgmax(const float* const* tables, const int64_t* vocab, const int32_t* col, int32_t F,
                                    int32_t E, const float* X, int64_t ldX, int32_t dense_col0, int32_t Nd, int64_t B,
                                    float* out, int64_t ldo, float* wg_max, int64_t wg_max_len, int32_t* status,
                                    mml_stream_t stream) {
  FieldTable ft;
  int rc = fill_fields(ft, tables, vocab, col, F, "mml_gather_fwd_wgmax");
  if (rc) return rc;
  MML_REQUIRE(B > 0 && E > 0 && Nd >= 0 && X && out && wg_max, "mml_gather_fwd_wgmax: bad arguments");
  MML_REQUIRE(ldo >= (int64_t)F * E + Nd, "mml_gather_fwd_wgmax: ldo=%lld < F*E+Nd", (long long)ldo);
  const int64_t need = mml_gather_wgmax_len(F, E, Nd, B);
  MML_REQUIRE(need > 0 && wg_max_len == need, "mml_gather_fwd_wgmax: wg_max_len must be mml_gather_wgmax_len() = %lld",
              (long long)need);
  GatherArgs a{};
  a.X = X; a.ldX = ldX; a.F = F; a.E = E; a.dense_col0 = dense_col0; a.Nd = Nd; a.B = B;
  a.out = out; a.ldo = ldo; a.status = status; a.wgmax = wg_max;
  return launch_gather(ft, a, to_stream(stream));
}

extern "C" int mml_gather_fwd_mark(const float* const* tables, const int64_t* vocab, const int32_t* col, int32_t F,
                                   int32_t E, const float* X, int64_t ldX, int32_t dense_col0, int32_t Nd, int64_t B,
                                   float* out, int64_t ldo, uint8_t* row_marks, int32_t* status, mml_stream_t stream) {
  FieldTable ft;
  int rc = fill_fields(ft, tables, vocab, col, F, "mml_gather_fwd_mark");
  if (rc) return rc;
  MML_REQUIRE(B >= 0 && E > 0 && Nd >= 0, "mml_gather_fwd_mark: bad sizes");
  MML_REQUIRE(B == 0 || (X && out), "mml_gather_fwd_mark: null X/out");
  MML_REQUIRE(ldo >= (int64_t)F * E + Nd, "mml_gather_fwd_mark: ldo too small");
  GatherArgs a{};
  a.X = X; a.ldX = ldX; a.F = F; a.E = E; a.dense_col0 = dense_col0; a.Nd = Nd; a.B = B;
  a.out = out; a.ldo = ldo; a.status = status; a.marks = row_marks;
  int64_t words = 0;
  for (int f = 0; f < F; ++f) {
    a.markbase[f] = words * 32;
    words += (vocab[f] + 31) / 32;
  }
  return launch_gather(ft, a, to_stream(stream));
}

extern "C" int mml_gather_fwd_idx32(const float* const* tables, const int64_t* vocab, int32_t F, int32_t E,
                                    const int32_t* idx, int64_t ldi, const float* dense, int64_t ldd, int32_t Nd,
                                    int64_t B, float* out, int64_t ldo, int32_t* status, mml_stream_t stream) {
  FieldTable ft;
  int rc = fill_fields(ft, tables, vocab, nullptr, F, "mml_gather_fwd_idx32");
  if (rc) return rc;
  MML_REQUIRE(B >= 0 && E > 0 && Nd >= 0, "mml_gather_fwd_idx32: bad sizes");
  MML_REQUIRE(B == 0 || (idx && out), "mml_gather_fwd_idx32: null idx/out");
  MML_REQUIRE(Nd == 0 || dense, "mml_gather_fwd_idx32: Nd>0 but dense is null");
  MML_REQUIRE(ldo >= (int64_t)F * E + Nd, "mml_gather_fwd_idx32: ldo too small");
  GatherArgs a{};
  a.idx = idx; a.ldi = ldi; a.dense = dense; a.ldd = ldd; a.F = F; a.E = E; a.Nd = Nd; a.B = B;
  a.out = out; a.ldo = ldo; a.status = status;
  return launch_gather(ft, a, to_stream(stream));
}

// mark map layout: field f owns bytes [32 * wordbase_f, 32 * wordbase_f + V_f), wordbase = prefix sum of ceil(V_f / 32)
static void set_marks(ScatterArgs& a, const FieldTable& ft, uint8_t* row_marks) {
  a.marks = row_marks;
  int64_t words = 0;
  for (int f = 0; f < a.F; ++f) {
    a.markbase[f] = words * 32;
    words += (ft.vocab[f] + 31) / 32;
  }
}

static int scatter_impl(float* const* grad_tables, const int64_t* vocab, const int32_t* col, int32_t F,
                        int32_t E, const float* X, int64_t ldX, const int32_t* idx, int64_t ldi, int64_t B,
                        const float* dOut, int64_t ldo, uint32_t* const* seen, const int64_t* rowbase,
                        int32_t* touched, int32_t* touched_count, int32_t touched_cap, uint8_t* row_marks,
                        int32_t* status, mml_stream_t stream) {
  FieldTable ft;
  int rc = fill_fields(ft, (const float* const*)grad_tables, vocab, col, F, "mml_scatter_bwd");
  if (rc) return rc;
  MML_REQUIRE(B >= 0 && E > 0, "mml_scatter_bwd: bad sizes");
  MML_REQUIRE(B == 0 || ((X || idx) && dOut), "mml_scatter_bwd: null X/dOut");
  MML_REQUIRE(!touched || (seen && rowbase && touched_count && touched_cap > 0),
              "mml_scatter_bwd: touched list needs seen/rowbase/touched_count/cap");
  if (B == 0 || F == 0) {  // an empty batch touches no row: the list of the previous call must not survive
    if (touched) {
      hipError_t e = hipMemsetAsync(touched_count, 0, sizeof(int32_t), to_stream(stream));
      if (e != hipSuccess) {
        set_error("mml_scatter_bwd: hipMemsetAsync: %s", hipGetErrorString(e));
        return MML_ERR_HIP;
      }
    }
    return MML_OK;
  }
  ScatterArgs a{};
  for (int f = 0; f < F; ++f) {
    a.gtab[f] = grad_tables[f];
    a.seen[f] = touched ? seen[f] : nullptr;
    a.rowbase[f] = touched ? rowbase[f] : 0;
    MML_REQUIRE(!touched || seen[f], "mml_scatter_bwd: seen[%d] is null", f);
  }
  a.X = X; a.ldX = ldX; a.idx = idx; a.ldi = ldi; a.B = B; a.dOut = dOut; a.ldo = ldo; a.F = F; a.E = E;
  a.touched = touched; a.touched_count = touched_count; a.touched_cap = touched_cap; a.status = status;
  set_marks(a, ft, row_marks);
  const int threads = 256;
  if (!getenv("MMLREC_SCATTER_OLD")) {
    rc = try_fold(ft, a, to_stream(stream), "mml_scatter_bwd");
    if (rc <= 0) return rc;
  }
  // only the fold kernel writes marks: a caller that asked for marks alone (mml_opt_tensor.grad_marks: "unmarked rows
  // have a zero gradient") must not be served by a kernel that leaves them unset
  MML_REQUIRE(!row_marks || touched, "mml_scatter_bwd: row_marks without a touched list needs E in {4, 8, 16} and a "
              "16-byte aligned dOut with ldo %% 4 == 0 (the LDS-fold kernel)");
  if (!touched) set_marks(a, ft, nullptr);
  if (touched) {  // the fold path resets the list itself; these kernels only append
    hipError_t e = hipMemsetAsync(touched_count, 0, sizeof(int32_t), to_stream(stream));
    if (e != hipSuccess) {
      set_error("mml_scatter_bwd: hipMemsetAsync: %s", hipGetErrorString(e));
      return MML_ERR_HIP;
    }
  }
  if (E <= 16) {
    // dynamic LDS stays under the 64 KiB default limit: SLOTS * (1 + E) * 4 bytes = 36 KiB (4 workgroups per CU)
    const int slots = (E <= 8) ? 1024 : 512;
    const int chunk = slots / 2;  // load factor <= 0.5
    const int64_t nblocks = (int64_t)F * 8 * cdiv(cdiv(B, chunk), 8);  // 8 XCD slots x F fields x chunk groups
    const size_t lds = (size_t)slots * (1 + E) * 4 + (touched ? (size_t)slots * 4 : 0);
    if (nblocks <= 0x7fffffff) {
      if (slots == 1024)
        MML_LAUNCH(scatter_hash_kernel<1024>, dim3((unsigned)nblocks), dim3(threads), lds, to_stream(stream), ft, a, chunk);
      else
        MML_LAUNCH(scatter_hash_kernel<512>, dim3((unsigned)nblocks), dim3(threads), lds, to_stream(stream), ft, a, chunk);
      return check_launch("mml_scatter_bwd");
    }
  }
  int64_t blocks = cdiv(B * (int64_t)F * E, threads);
  if (blocks > 256 * 16) blocks = 256 * 16;
  MML_LAUNCH(scatter_atomic_kernel, dim3((unsigned)blocks), dim3(threads), 0, to_stream(stream), ft, a);
  return check_launch("mml_scatter_bwd");
}

extern "C" int mml_scatter_bwd(float* const* grad_tables, const int64_t* vocab, const int32_t* col, int32_t F,
                               int32_t E, const float* X, int64_t ldX, int64_t B, const float* dOut, int64_t ldo,
                               uint32_t* const* seen, const int64_t* rowbase, int32_t* touched,
                               int32_t* touched_count, int32_t touched_cap, uint8_t* row_marks, int32_t* status,
                               mml_stream_t stream) {
  return scatter_impl(grad_tables, vocab, col, F, E, X, ldX, nullptr, 0, B, dOut, ldo, seen, rowbase, touched,
                      touched_count, touched_cap, row_marks, status, stream);
}

extern "C" int mml_scatter_bwd_idx32(float* const* grad_tables, const int64_t* vocab, int32_t F, int32_t E,
                                     const int32_t* idx, int64_t ldi, int64_t B, const float* dOut, int64_t ldo,
                                     uint32_t* const* seen, const int64_t* rowbase, int32_t* touched,
                                     int32_t* touched_count, int32_t touched_cap, uint8_t* row_marks, int32_t* status,
                                     mml_stream_t stream) {
  return scatter_impl(grad_tables, vocab, nullptr, F, E, nullptr, 0, idx, ldi, B, dOut, ldo, seen, rowbase, touched,
                      touched_count, touched_cap, row_marks, status, stream);
}

static int unique_impl(const int64_t* vocab, const int32_t* col, int32_t F, int32_t E, const float* X, int64_t ldX,
                       const int32_t* idx, int64_t ldi, int64_t B, uint32_t* const* seen, const int64_t* rowbase,
                       int32_t* touched, int32_t* touched_count, int32_t touched_cap, uint8_t* row_marks,
                       int32_t* status, mml_stream_t stream) {
  MML_REQUIRE(F >= 0 && F <= MML_MAX_FIELDS && vocab && seen && rowbase && touched && touched_count && touched_cap > 0,
              "mml_index_unique: bad arguments");
  MML_REQUIRE(E > 0 && E <= 16, "mml_index_unique: E must be in [1,16]");
  if (B == 0 || F == 0) return MML_OK;
  MML_REQUIRE(X || idx, "mml_index_unique: null X");
  FieldTable ft;
  ScatterArgs a{};
  for (int f = 0; f < F; ++f) {
    MML_REQUIRE(vocab[f] > 0 && seen[f], "mml_index_unique: field %d malformed", f);
    ft.tab[f] = nullptr;
    ft.vocab[f] = vocab[f];
    ft.col[f] = col ? col[f] : f;
    a.seen[f] = seen[f];
    a.rowbase[f] = rowbase[f];
  }
  a.X = X; a.ldX = ldX; a.idx = idx; a.ldi = ldi; a.B = B; a.dOut = nullptr; a.F = F; a.E = E;
  a.touched = touched; a.touched_count = touched_count; a.touched_cap = touched_cap; a.status = status;
  set_marks(a, ft, row_marks);
  if (row_marks && !getenv("MMLREC_SCATTER_OLD")) {  // byte marks + bitmap compaction: no LDS, no atomics
    int64_t blocks = cdiv(B * (int64_t)F, 256);
    if (blocks > 256 * 16) blocks = 256 * 16;
    MML_LAUNCH(mark_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, to_stream(stream), ft, a);
    int rc2 = check_launch("mml_index_unique");
    if (rc2) return rc2;
    return launch_compact(ft, a, to_stream(stream), "mml_index_unique", true);
  }
  if (!getenv("MMLREC_SCATTER_OLD")) {
    const int rc = try_fold(ft, a, to_stream(stream), "mml_index_unique");
    if (rc <= 0) return rc;
  }
  const int slots = (E <= 8) ? 1024 : 512;
  const int chunk = slots / 2;
  const int64_t nblocks = (int64_t)F * 8 * cdiv(cdiv(B, chunk), 8);  // 8 XCD slots x F fields x chunk groups
  MML_REQUIRE(nblocks <= 0x7fffffff, "mml_index_unique: grid too large");
  const size_t lds = (size_t)slots * (1 + E) * 4 + (size_t)slots * 4;
  if (slots == 1024)
    MML_LAUNCH(scatter_hash_kernel<1024>, dim3((unsigned)nblocks), dim3(256), lds, to_stream(stream), ft, a, chunk);
  else
    MML_LAUNCH(scatter_hash_kernel<512>, dim3((unsigned)nblocks), dim3(256), lds, to_stream(stream), ft, a, chunk);
  return check_launch("mml_index_unique");
}

extern "C" int mml_index_unique(const int64_t* vocab, const int32_t* col, int32_t F, int32_t E, const float* X,
                                int64_t ldX, int64_t B, uint32_t* const* seen, const int64_t* rowbase,
                                int32_t* touched, int32_t* touched_count, int32_t touched_cap, uint8_t* row_marks,
                                int32_t* status, mml_stream_t stream) {
  return unique_impl(vocab, col, F, E, X, ldX, nullptr, 0, B, seen, rowbase, touched, touched_count, touched_cap,
                     row_marks, status, stream);
}

extern "C" int mml_index_unique_idx32(const int64_t* vocab, int32_t F, int32_t E, const int32_t* idx, int64_t ldi,
                                      int64_t B, uint32_t* const* seen, const int64_t* rowbase, int32_t* touched,
                                      int32_t* touched_count, int32_t touched_cap, uint8_t* row_marks, int32_t* status,
                                      mml_stream_t stream) {
  return unique_impl(vocab, nullptr, F, E, nullptr, 0, idx, ldi, B, seen, rowbase, touched, touched_count,
                     touched_cap, row_marks, status, stream);
}

extern "C" int mml_rows_compact(uint32_t* const* seen, const int64_t* vocab, const int64_t* rowbase, int32_t F,
                                int32_t* touched, int32_t* touched_count, int32_t touched_cap, uint8_t* row_marks,
                                mml_stream_t stream) {
  MML_REQUIRE(F >= 1 && F <= MML_MAX_FIELDS && seen && vocab && rowbase && touched && touched_count && touched_cap > 0,
              "mml_rows_compact: bad arguments");
  FieldTable ft;
  ScatterArgs a{};
  a.F = F;
  for (int f = 0; f < F; ++f) {
    MML_REQUIRE(vocab[f] > 0 && seen[f], "mml_rows_compact: field %d malformed", f);
    ft.vocab[f] = vocab[f];
    a.seen[f] = seen[f];
    a.rowbase[f] = rowbase[f];
  }
  a.touched = touched; a.touched_count = touched_count; a.touched_cap = touched_cap;
  set_marks(a, ft, row_marks);
  return launch_compact(ft, a, to_stream(stream), "mml_rows_compact");
}

extern "C" int mml_scatter_bwd_det(float* const* grad_tables, const int64_t* vocab, const int32_t* col, int32_t F,
                                   int32_t E, const float* X, int64_t ldX, int64_t B, const float* dOut, int64_t ldo,
                                   int64_t* const* acc64, uint32_t* amax_slot, uint8_t* row_marks, int32_t clear_marks,
                                   int32_t* status, mml_stream_t stream) {
  FieldTable ft;
  int rc = fill_fields(ft, (const float* const*)grad_tables, vocab, col, F, "mml_scatter_bwd_det");
  if (rc) return rc;
  MML_REQUIRE(B >= 0 && (E == 4 || E == 8 || E == 16), "mml_scatter_bwd_det: E must be 4, 8 or 16");
  MML_REQUIRE(B == 0 || (X && dOut), "mml_scatter_bwd_det: null X/dOut");
  MML_REQUIRE(acc64 && amax_slot && row_marks, "mml_scatter_bwd_det: acc64, amax_slot and row_marks are required");
  MML_REQUIRE(ldo % 4 == 0 && aligned16(dOut), "mml_scatter_bwd_det: dOut must be 16-byte aligned with ldo %% 4 == 0");
  if (B == 0 || F == 0) return MML_OK;
  hipStream_t st = to_stream(stream);
  // the launch-wide magnitude of dOut (a maximum does not depend on the order either)
  rc = mml_amax_reset(amax_slot, 1, stream);
  if (rc) return rc;
  mml_amax_desc ad{};
  ad.x = dOut; ad.rows = B; ad.ld = ldo; ad.cols = F * E; ad.slot = amax_slot;
  rc = mml_amax_batch(&ad, 1, stream);
  if (rc) return rc;
  ScatterArgs a{};
  DetFinalArgs fa{};
  for (int f = 0; f < F; ++f) {
    MML_REQUIRE(acc64[f], "mml_scatter_bwd_det: acc64[%d] is null", f);
    MML_REQUIRE(vocab[f] <= 0x7fffffff, "mml_scatter_bwd_det: table %d too large", f);
    a.gtab[f] = grad_tables[f];
    a.acc64[f] = reinterpret_cast<long long*>(acc64[f]);
    fa.gtab[f] = grad_tables[f];
    fa.acc64[f] = a.acc64[f];
    fa.vocab[f] = vocab[f];
  }
  a.X = X; a.ldX = ldX; a.B = B; a.dOut = dOut; a.ldo = ldo; a.F = F; a.E = E; a.status = status;
  a.amax_dout = amax_slot;
  // headroom: a row may receive all B addends at the launch-wide maximum: 24 significand bits + shift + log2(B) < 63
  int lg = 1;
  while (((int64_t)1 << lg) < B) ++lg;
  int shift = 62 - 24 - lg;
  shift = shift > 28 ? 28 : (shift < 4 ? 4 : shift);
  a.fix_shift = shift;
  set_marks(a, ft, row_marks);
  if (E == 8) rc = launch_fold<1024, 8, 1024, true>(ft, a, st, "mml_scatter_bwd_det");
  else if (E == 4) rc = launch_fold<1024, 4, 512, true>(ft, a, st, "mml_scatter_bwd_det");
  else rc = launch_fold<512, 16, 1024, true>(ft, a, st, "mml_scatter_bwd_det");
  if (rc) return rc < 0 ? rc : MML_ERR_UNSUPPORTED;
  int total = 0;
  for (int f = 0; f < F; ++f) {
    fa.markbase[f] = a.markbase[f];
    int64_t nb = cdiv(vocab[f], 256 * 8);
    if (nb > 1024) nb = 1024;
    if (nb < 1) nb = 1;
    fa.blk0[f] = total;
    total += (int)nb;
  }
  fa.blk0[F] = total;
  fa.marks = row_marks; fa.amax_dout = amax_slot; fa.F = F; fa.E = E; fa.fix_shift = shift; fa.clear_marks = clear_marks;
  MML_LAUNCH(scatter_det_finalize_kernel, dim3((unsigned)total), dim3(256), 0, st, fa);
  return check_launch("mml_scatter_bwd_det");
}
