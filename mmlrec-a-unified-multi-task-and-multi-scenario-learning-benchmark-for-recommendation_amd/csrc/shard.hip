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
