// Row-wise sharded embedding tables (no reference counterpart: the reference's --is_parallel path is a
// broken stub, main.py:81-83): requester-side routing of a batch's lookups to the ranks that own the rows.
//
// Layout contract (parallel.RowSharding): row r of field f lives on rank  owner = (r + f) mod world  (the rotation by
// the field number keeps the hot low-rank rows of all Zipf-distributed fields off a single rank) at local row r / world
// of that field's shard; every rank keeps its F shards back to back in ONE flat [R, E] buffer, so a lookup travels as
// a single int32 key = keybase[f] + r / world into the owner's flat row space.
//
//   mml_route_count : per-owner lookup counts of a batch                  -> counters[0 .. world)
//   mml_route_place : keys grouped by owner (the all-to-all send layout)  -> send_keys, and for every (sample, field)
//                     the position of its key in that layout              -> pos [B, F]
// Owners answer with rows in the order they received the keys, so `pos` also addresses the returned rows: the
// expansion into dnn_input is mml_gather_fwd_idx32 on the received row block, and mml_rows_permute is the inverse
// (pack d(dnn_input) pieces into the send order of the gradient exchange).  Integer work, HBM-bound, bit-exact.
#include "common.hpp"

namespace mml {

struct RouteArgs {
  const float* X;      // fp32-encoded indices (reference layout, model/basemodel.py:476) or null
  const int32_t* idx;  // native indices or null
  int64_t ldX, ldi;
  int64_t B;
  int32_t F, world;
  int32_t col[MML_MAX_FIELDS];
  int64_t vocab[MML_MAX_FIELDS];
  int32_t keybase[MML_MAX_FIELDS];
  int32_t* counters;   // [2 * world]: lookups per owner | placement cursors
  int32_t* send_keys;  // [B * F]
  int32_t* pos;        // [B, F]
  int32_t* status;
};

constexpr int kRouteItems = 4;

template <bool PLACE>
__global__ __launch_bounds__(256) void route_kernel(const RouteArgs a) {
  __shared__ int cnt_l[64];
  __shared__ int base_l[64];
  if (threadIdx.x < 64) cnt_l[threadIdx.x] = 0;
  __syncthreads();
  const int64_t total = a.B * a.F;
  const int64_t i0 = (int64_t)blockIdx.x * (256 * kRouteItems) + threadIdx.x;
  int own[kRouteItems], key[kRouteItems], slot[kRouteItems];
  int bad = 0;
#pragma unroll
  for (int k = 0; k < kRouteItems; ++k) {
    const int64_t i = i0 + k * 256;
    own[k] = -1;
    if (i >= total) continue;
    const int64_t b = i / a.F;
    const int f = (int)(i - b * a.F);
    int64_t r = a.idx ? (int64_t)a.idx[b * a.ldi + f] : (int64_t)a.X[b * a.ldX + a.col[f]];  // truncation toward zero
    if (r < 0) {
      bad |= 1;
      r = 0;
    } else if (r >= a.vocab[f]) {
      bad |= 2;
      r = a.vocab[f] - 1;
    }
    own[k] = (int)((r + f) % a.world);
    key[k] = a.keybase[f] + (int)(r / a.world);
    slot[k] = atomicAdd(&cnt_l[own[k]], 1);
  }
  __syncthreads();
  if (!PLACE) {
    if (threadIdx.x < a.world && cnt_l[threadIdx.x]) atomicAdd(a.counters + threadIdx.x, cnt_l[threadIdx.x]);
  } else {
    if (threadIdx.x < a.world) {
      int off = 0;  // start of this owner's segment = lookups bound for the lower ranks
      for (int j = 0; j < (int)threadIdx.x; ++j) off += a.counters[j];
      const int n = cnt_l[threadIdx.x];
      base_l[threadIdx.x] = off + (n ? atomicAdd(a.counters + a.world + threadIdx.x, n) : 0);
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kRouteItems; ++k) {
      if (own[k] < 0) continue;
      const int p = base_l[own[k]] + slot[k];
      a.send_keys[p] = key[k];
      a.pos[i0 + k * 256] = p;
    }
  }
  if (bad && a.status) atomicOr(a.status, bad);
}

// dst[pos[b, f], :] = src[b, f*E : (f+1)*E]  (every position is hit exactly once: plain stores)
__global__ __launch_bounds__(256) void rows_permute_kernel(const float* __restrict__ src, int64_t lds,
                                                           const int32_t* __restrict__ pos, int32_t F, int32_t E,
                                                           int64_t B, float* __restrict__ dst, bool vec) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  if (vec) {
    const int e4 = E >> 2;
    const int64_t total = B * F * e4;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
      const int64_t bf = i / e4;
      const int part = (int)(i - bf * e4);
      const int64_t b = bf / F;
      const int f = (int)(bf - b * F);
      const float4 v = *reinterpret_cast<const float4*>(src + b * lds + (int64_t)f * E + part * 4);
      *reinterpret_cast<float4*>(dst + (int64_t)pos[bf] * E + part * 4) = v;
    }
  } else {
    const int64_t total = B * F * E;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
      const int64_t bf = i / E;
      const int e = (int)(i - bf * E);
      const int64_t b = bf / F;
      const int f = (int)(bf - b * F);
      dst[(int64_t)pos[bf] * E + e] = src[b * lds + (int64_t)f * E + e];
    }
  }
}

// Shard <-> full table conversion: shard[l, :] = table[l * world + first, :] for l < rows_local (rows beyond the table
// end are zero-filled), or the inverse copy.  first = (rank - f) mod world.
__global__ __launch_bounds__(256) void shard_rows_kernel(float* __restrict__ table, int64_t V, float* __restrict__ shard,
                                                         int64_t rows_local, int32_t E, int32_t world, int32_t first,
                                                         int to_table) {
  const int64_t total = rows_local * E;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int64_t l = i / E;
    const int e = (int)(i - l * E);
    const int64_t r = l * world + first;
    if (to_table) {
      if (r < V) table[r * E + e] = shard[i];
    } else {
      shard[i] = r < V ? table[r * E + e] : 0.f;
    }
  }
}

// ---- requester-side de-duplication (the exchange carries every distinct row once) -----------------------------------
// The batch's distinct rows (the touched list mml_index_unique builds: global row ids rowbase[f] + r) are routed
// instead of the B*F lookups: under Zipf a 65 536-sample AE-30 batch holds 209 k distinct rows against 1.97 M lookups.
//   route_list_kernel<false> : per-owner counts of the list            -> counters[0 .. world)
//   route_list_kernel<true>  : send_keys grouped by owner, slot_of[global row id] = position in that order
//   lookup_slots_kernel      : pos[b, f] = slot_of[rowbase[f] + X[b, f]]   (then the usual expansion / packing by pos)
//   rows_clear_kernel        : clears the list's `seen` bits (the requester-side bitmaps have no optimizer pass that would)
struct RouteListArgs {
  const int32_t* list;
  const int32_t* count;  // device: entries in list
  int32_t cap;
  int32_t F, world;
  int64_t rowbase[MML_MAX_FIELDS + 1];
  int32_t keybase[MML_MAX_FIELDS];
  int32_t* counters;
  int32_t* send_keys;
  int32_t* slot_of;
};

__device__ __forceinline__ int field_of(const int64_t* rb, int F, int64_t g) {
  int lo = 0, hi = F;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (rb[mid] <= g) lo = mid;
    else hi = mid;
  }
  return lo;
}

template <bool PLACE>
__global__ __launch_bounds__(256) void route_list_kernel(const RouteListArgs a) {
  __shared__ int cnt_l[64];
  __shared__ int base_l[64];
  __shared__ int64_t rb[MML_MAX_FIELDS + 1];
  if (threadIdx.x < 64) cnt_l[threadIdx.x] = 0;
  for (int i = threadIdx.x; i <= a.F; i += 256) rb[i] = a.rowbase[i];
  __syncthreads();
  int n = *a.count;
  if (n > a.cap) n = a.cap;
  const int64_t i0 = (int64_t)blockIdx.x * (256 * kRouteItems) + threadIdx.x;
  if ((int64_t)blockIdx.x * (256 * kRouteItems) >= n) return;  // (uniform per workgroup)
  int own[kRouteItems], key[kRouteItems], slot[kRouteItems], gid[kRouteItems];
#pragma unroll
  for (int k = 0; k < kRouteItems; ++k) {
    const int64_t i = i0 + k * 256;
    own[k] = -1;
    if (i >= n) continue;
    const int64_t g = a.list[i];
    const int f = field_of(rb, a.F, g);
    const int64_t r = g - rb[f];
    gid[k] = (int)g;
    own[k] = (int)((r + f) % a.world);
    key[k] = a.keybase[f] + (int)(r / a.world);
    slot[k] = atomicAdd(&cnt_l[own[k]], 1);
  }
  __syncthreads();
  if (!PLACE) {
    if (threadIdx.x < a.world && cnt_l[threadIdx.x]) atomicAdd(a.counters + threadIdx.x, cnt_l[threadIdx.x]);
  } else {
    if (threadIdx.x < a.world) {
      int off = 0;
      for (int j = 0; j < (int)threadIdx.x; ++j) off += a.counters[j];
      const int m = cnt_l[threadIdx.x];
      base_l[threadIdx.x] = off + (m ? atomicAdd(a.counters + a.world + threadIdx.x, m) : 0);
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kRouteItems; ++k) {
      if (own[k] < 0) continue;
      const int p = base_l[own[k]] + slot[k];
      a.send_keys[p] = key[k];
      a.slot_of[gid[k]] = p;
    }
  }
}

struct LookupArgs {
  const float* X;
  int64_t ldX, B;
  int32_t F;
  int32_t col[MML_MAX_FIELDS];
  int64_t vocab[MML_MAX_FIELDS];
  int64_t rowbase[MML_MAX_FIELDS];
  const int32_t* slot_of;
  int32_t* pos;
  int32_t* status;
};

__global__ __launch_bounds__(256) void lookup_slots_kernel(const LookupArgs a) {
  const int64_t total = a.B * a.F;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  int bad = 0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int64_t b = i / a.F;
    const int f = (int)(i - b * a.F);
    int64_t r = (int64_t)a.X[b * a.ldX + a.col[f]];
    if (r < 0) { bad |= 1; r = 0; }
    else if (r >= a.vocab[f]) { bad |= 2; r = a.vocab[f] - 1; }
    a.pos[i] = a.slot_of[a.rowbase[f] + r];
  }
  if (bad && a.status) atomicOr(a.status, bad);
}

struct ClearArgs {
  const int32_t* list;
  const int32_t* count;
  int32_t cap, F;
  int64_t rowbase[MML_MAX_FIELDS + 1];
  uint32_t* seen[MML_MAX_FIELDS];
};

__global__ __launch_bounds__(256) void rows_clear_kernel(const ClearArgs a) {
  __shared__ int64_t rb[MML_MAX_FIELDS + 1];
  __shared__ uint32_t* seen_l[MML_MAX_FIELDS];
  for (int i = threadIdx.x; i <= a.F; i += 256) rb[i] = a.rowbase[i];
  for (int i = threadIdx.x; i < a.F; i += 256) seen_l[i] = a.seen[i];
  __syncthreads();
  int n = *a.count;
  if (n > a.cap) n = a.cap;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const int64_t g = a.list[i];
    const int f = field_of(rb, a.F, g);
    const int64_t r = g - rb[f];
    seen_l[f][r >> 5] = 0u;  // every row of the word that is set is in the list: racing writers store the same 0
  }
}

}  // namespace mml

using namespace mml;

static int fill_route(RouteArgs& a, const float* X, int64_t ldX, const int32_t* idx, int64_t ldi, const int32_t* col,
                      const int64_t* vocab, const int64_t* keybase, int32_t F, int64_t B, int32_t world,
                      int32_t* counters, int32_t* status, const char* who) {
  MML_REQUIRE(F > 0 && F <= MML_MAX_FIELDS, "%s: F=%d outside [1,%d]", who, F, MML_MAX_FIELDS);
  MML_REQUIRE(world >= 1 && world <= 64, "%s: world=%d outside [1,64]", who, world);
  MML_REQUIRE(B >= 0 && B * (int64_t)F < 0x7fffffff, "%s: B*F must fit int32", who);
  MML_REQUIRE((X || idx) && vocab && counters, "%s: null argument", who);
  a.X = X; a.idx = idx; a.ldX = ldX; a.ldi = ldi; a.B = B; a.F = F; a.world = world;
  for (int f = 0; f < F; ++f) {
    MML_REQUIRE(vocab[f] > 0, "%s: vocab[%d]=%lld", who, f, (long long)vocab[f]);
    a.col[f] = col ? col[f] : f;
    a.vocab[f] = vocab[f];
    if (keybase) {
      MML_REQUIRE(keybase[f] >= 0 && keybase[f] + (vocab[f] + world - 1) / world <= 0x7fffffff,
                  "%s: flat row space exceeds int32", who);
      a.keybase[f] = (int32_t)keybase[f];
    }
  }
  a.counters = counters;
  a.status = status;
  return MML_OK;
}

extern "C" int mml_route_count(const float* X, int64_t ldX, const int32_t* idx, int64_t ldi, const int32_t* col,
                               const int64_t* vocab, int32_t F, int64_t B, int32_t world, int32_t* counters,
                               int32_t* status, mml_stream_t stream) {
  RouteArgs a{};
  int rc = fill_route(a, X, ldX, idx, ldi, col, vocab, nullptr, F, B, world, counters, status, "mml_route_count");
  if (rc) return rc;
  hipError_t e = hipMemsetAsync(counters, 0, sizeof(int32_t) * 2 * world, to_stream(stream));
  if (e != hipSuccess) {
    set_error("mml_route_count: hipMemsetAsync: %s", hipGetErrorString(e));
    return MML_ERR_HIP;
  }
  if (B == 0) return MML_OK;
  const int64_t blocks = cdiv(B * F, 256 * kRouteItems);
  MML_LAUNCH(route_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, to_stream(stream), a);
  return check_launch("mml_route_count");
}

extern "C" int mml_route_place(const float* X, int64_t ldX, const int32_t* idx, int64_t ldi, const int32_t* col,
                               const int64_t* vocab, const int64_t* keybase, int32_t F, int64_t B, int32_t world,
                               int32_t* counters, int32_t* send_keys, int32_t* pos, int32_t* status,
                               mml_stream_t stream) {
  RouteArgs a{};
  MML_REQUIRE(keybase && send_keys && pos, "mml_route_place: null argument");
  int rc = fill_route(a, X, ldX, idx, ldi, col, vocab, keybase, F, B, world, counters, status, "mml_route_place");
  if (rc) return rc;
  if (B == 0) return MML_OK;
  a.send_keys = send_keys;
  a.pos = pos;
  const int64_t blocks = cdiv(B * F, 256 * kRouteItems);
  MML_LAUNCH(route_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, to_stream(stream), a);
  return check_launch("mml_route_place");
}

extern "C" int mml_rows_permute(const float* src, int64_t lds, const int32_t* pos, int32_t F, int32_t E, int64_t B,
                                float* dst, mml_stream_t stream) {
  MML_REQUIRE(F > 0 && E > 0 && B >= 0, "mml_rows_permute: bad sizes");
  if (B == 0) return MML_OK;
  MML_REQUIRE(src && pos && dst, "mml_rows_permute: null argument");
  const bool vec = (E % 4 == 0) && (lds % 4 == 0) && aligned16(src) && aligned16(dst);
  const int64_t total = B * F * (vec ? E / 4 : E);
  int64_t blocks = cdiv(total, 256);
  if (blocks > 256 * 16) blocks = 256 * 16;
  MML_LAUNCH(rows_permute_kernel, dim3((unsigned)blocks), dim3(256), 0, to_stream(stream), src, lds, pos, F, E, B, dst,
             vec);
  return check_launch("mml_rows_permute");
}

extern "C" int mml_shard_rows(float* table, int64_t V, float* shard, int64_t rows_local, int32_t E, int32_t world,
                              int32_t first, int32_t to_table, mml_stream_t stream) {
  MML_REQUIRE(table && shard && V > 0 && rows_local >= 0 && E > 0 && world >= 1 && first >= 0 && first < world,
              "mml_shard_rows: bad arguments");
  if (rows_local == 0) return MML_OK;
  int64_t blocks = cdiv(rows_local * E, 256);
  if (blocks > 256 * 16) blocks = 256 * 16;
  MML_LAUNCH(shard_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, to_stream(stream), table, V, shard, rows_local, E,
             world, first, to_table);
  return check_launch("mml_shard_rows");
}

static int fill_route_list(RouteListArgs& a, const int32_t* list, const int32_t* count, int32_t cap,
                           const int64_t* vocab, const int64_t* rowbase, const int64_t* keybase, int32_t F, int32_t world,
                           int32_t* counters, const char* who) {
  MML_REQUIRE(F > 0 && F <= MML_MAX_FIELDS && world >= 1 && world <= 64 && cap > 0, "%s: bad sizes", who);
  MML_REQUIRE(list && count && vocab && rowbase && counters, "%s: null argument", who);
  a.list = list; a.count = count; a.cap = cap; a.F = F; a.world = world; a.counters = counters;
  for (int f = 0; f <= F; ++f) a.rowbase[f] = rowbase[f];
  MML_REQUIRE(rowbase[F] <= 0x7fffffff, "%s: global row ids must fit int32", who);
  if (keybase)
    for (int f = 0; f < F; ++f) a.keybase[f] = (int32_t)keybase[f];
  return MML_OK;
}

extern "C" int mml_route_list_count(const int32_t* list, const int32_t* count, int32_t cap, const int64_t* vocab,
                                    const int64_t* rowbase, int32_t F, int32_t world, int32_t* counters,
                                    mml_stream_t stream) {
  RouteListArgs a{};
  int rc = fill_route_list(a, list, count, cap, vocab, rowbase, nullptr, F, world, counters, "mml_route_list_count");
  if (rc) return rc;
  hipError_t e = hipMemsetAsync(counters, 0, sizeof(int32_t) * 2 * world, to_stream(stream));
  if (e != hipSuccess) {
    set_error("mml_route_list_count: hipMemsetAsync: %s", hipGetErrorString(e));
    return MML_ERR_HIP;
  }
  MML_LAUNCH(route_list_kernel<false>, dim3((unsigned)cdiv(cap, 256 * kRouteItems)), dim3(256), 0, to_stream(stream), a);
  return check_launch("mml_route_list_count");
}

extern "C" int mml_route_list_place(const int32_t* list, const int32_t* count, int32_t cap, const int64_t* vocab,
                                    const int64_t* rowbase, const int64_t* keybase, int32_t F, int32_t world,
                                    int32_t* counters, int32_t* send_keys, int32_t* slot_of, mml_stream_t stream) {
  RouteListArgs a{};
  MML_REQUIRE(keybase && send_keys && slot_of, "mml_route_list_place: null argument");
  int rc = fill_route_list(a, list, count, cap, vocab, rowbase, keybase, F, world, counters, "mml_route_list_place");
  if (rc) return rc;
  a.send_keys = send_keys;
  a.slot_of = slot_of;
  MML_LAUNCH(route_list_kernel<true>, dim3((unsigned)cdiv(cap, 256 * kRouteItems)), dim3(256), 0, to_stream(stream), a);
  return check_launch("mml_route_list_place");
}

extern "C" int mml_lookup_slots(const float* X, int64_t ldX, const int32_t* col, const int64_t* vocab,
                                const int64_t* rowbase, int32_t F, int64_t B, const int32_t* slot_of, int32_t* pos,
                                int32_t* status, mml_stream_t stream) {
  MML_REQUIRE(F > 0 && F <= MML_MAX_FIELDS && B >= 0, "mml_lookup_slots: bad sizes");
  if (B == 0) return MML_OK;
  MML_REQUIRE(X && vocab && rowbase && slot_of && pos, "mml_lookup_slots: null argument");
  LookupArgs a{};
  a.X = X; a.ldX = ldX; a.B = B; a.F = F; a.slot_of = slot_of; a.pos = pos; a.status = status;
  for (int f = 0; f < F; ++f) {
    a.col[f] = col ? col[f] : f;
    a.vocab[f] = vocab[f];
    a.rowbase[f] = rowbase[f];
  }
  int64_t blocks = cdiv(B * F, 256);
  if (blocks > 256 * 16) blocks = 256 * 16;
  MML_LAUNCH(lookup_slots_kernel, dim3((unsigned)blocks), dim3(256), 0, to_stream(stream), a);
  return check_launch("mml_lookup_slots");
}

extern "C" int mml_rows_clear(const int32_t* list, const int32_t* count, int32_t cap, const int64_t* rowbase,
                              uint32_t* const* seen, int32_t F, mml_stream_t stream) {
  MML_REQUIRE(F > 0 && F <= MML_MAX_FIELDS && cap > 0 && list && count && rowbase && seen, "mml_rows_clear: bad arguments");
  ClearArgs a{};
  a.list = list; a.count = count; a.cap = cap; a.F = F;
  for (int f = 0; f <= F; ++f) a.rowbase[f] = rowbase[f];
  for (int f = 0; f < F; ++f) a.seen[f] = seen[f];
  int64_t blocks = cdiv(cap, 256);
  if (blocks > 256 * 8) blocks = 256 * 8;
  MML_LAUNCH(rows_clear_kernel, dim3((unsigned)blocks), dim3(256), 0, to_stream(stream), a);
  return check_launch("mml_rows_clear");
}
