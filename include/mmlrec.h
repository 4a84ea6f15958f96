/*
 * mmlrec.h -- C ABI of the MI355X-native MMLRec hot path (libmmlrec_hip.so).
 *
 * The reference (alipay/MMLRec) is pure Python/PyTorch and has no FFI of its own: the boundary it
 * exposes is the nn.Module contract (forward(X, domain_mask) in model/{sharedbottom,mmoe,ple,star,pepnet}.py
 * driven by BaseModel.fit, model/basemodel.py:261-313).  Everything below that contract is ATen.  This header
 * declares the entry points that replace those ATen call sites; each one cites the reference lines whose
 * arithmetic it takes over.  INTEGRATION.md shows the ctypes stub a reference maintainer would add.
 *
 * Conventions
 *   - plain C: raw device pointers, sizes, a hipStream_t passed as void*; no torch types.
 *   - every function returns MML_OK (0) or a negative error code and never throws; mml_last_error()
 *     returns a thread-local message for the last failure.
 *   - caller owns every buffer.  Nothing is allocated inside a call; kernels that need scratch take an
 *     explicit workspace pointer + size and a *_workspace_bytes() query tells how much they need.
 *   - descriptor arrays (mml_*_desc) live in HOST memory; they are copied into the kernel argument
 *     block, so no hidden H2D traffic and the calls are hipGraph-capturable.
 *   - all matrices are fp32 row-major with an explicit leading dimension (elements, not bytes).
 *   - re-entrant on different streams; no global state besides the thread-local error string.
 *   - device-side problems (an index outside its table) set bits in an optional int32 status word
 *     instead of trapping: bit 0 = index < 0, bit 1 = index >= vocab (nn.Embedding raises IndexError
 *     on CPU for these; the host wrapper turns a non-zero status into the same exception).
 */
#ifndef MMLREC_H
#define MMLREC_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MML_OK 0
#define MML_ERR_ARG (-1)         /* bad argument (null pointer, size out of range, misaligned buffer) */
#define MML_ERR_HIP (-2)         /* a HIP runtime call failed; see mml_last_error() */
#define MML_ERR_UNSUPPORTED (-3) /* valid request the library does not implement */

#define MML_MAX_FIELDS 64   /* sparse fields per gather/scatter launch (more: call again with the next slice) */
#define MML_MAX_GROUP 16    /* GEMM problems per grouped launch */
#define MML_MAX_SRC 8       /* accumulated sources per dgrad output */
#define MML_MAX_EXPERTS 16  /* experts visible to one gate group */
#define MML_MAX_GATES 8     /* gates per gate group */
#define MML_MAX_HEADS 8     /* prediction heads per head launch */
#define MML_MAX_OPT_TENSORS 32
#define MML_AMAX_WORDS 8    /* words of one operand-magnitude slot (see "operand magnitudes" at the GEMM family) */
#define MML_MAX_AMAX 64     /* tensors per mml_amax_batch launch (16 until round 6: PepNet's 42 stable weights took three launches) */

typedef void* mml_stream_t; /* hipStream_t */

/* activation / derivative codes */
#define MML_ACT_NONE 0
#define MML_ACT_RELU 1
#define MML_ACT_SIGMOID 2
#define MML_ACT_SIGMOID2 3 /* 2*sigmoid(x): PepNet GateNN (model/pepnet.py:31-32) */

/* optimizer kinds: torch.optim defaults as used by BaseModel._get_optim (model/basemodel.py:569-584) */
#define MML_OPT_SGD 0
#define MML_OPT_ADAM 1
#define MML_OPT_ADAGRAD 2
#define MML_OPT_RMSPROP 3

int mml_version(void);
const char* mml_last_error(void);
/* out[0]=CU count, [1]=LDS bytes/CU, [2]=total HBM bytes, [3]=wavefront size, [4]=max clock kHz, [5]=gfx arch number */
int mml_device_caps(int device, int64_t* out6);
/* A HIP stream whose kernels may only run on compute units [cu_lo, cu_hi) of `device` (bit i of the queue's CU mask;
 * on gfx950 the driver deals mask bits round-robin over the 8 XCDs, so a contiguous range takes the same number of CUs
 * from every XCD).  Lets an HBM-bound stream (the dense table optimizer) and an MFMA-bound one (the weight-gradient
 * GEMMs) run side by side instead of alternating workgroup by workgroup.  mml_stream_destroy releases it. */
int mml_stream_create_cu_range(int device, int cu_lo, int cu_hi, mml_stream_t* stream_out);
int mml_stream_destroy(mml_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * K1  fused multi-field gather + concat.
 * Replaces BaseModel.input_from_feature_columns (model/basemodel.py:461-487: per field
 * nn.Embedding(X[:, c].long())) and combined_dnn_input (model/utils.py:434-446: cat + flatten + cat).
 *   out[b, f*E + e]   = tables[f][trunc(X[b, col[f]]) * E + e]     f < F, e < E   (bit-exact copy)
 *   out[b, F*E + j]   = X[b, dense_col0 + j]                       j < Nd
 * X carries indices as fp32 (model/basemodel.py:262, :476) -> exact below 2^24 rows.
 * tables/vocab/col are HOST arrays of length F (tables holds DEVICE pointers).
 * ---------------------------------------------------------------------------------------------- */
int mml_gather_fwd(const float* const* tables, const int64_t* vocab, const int32_t* col, int32_t F, int32_t E,
                   const float* X, int64_t ldX, int32_t dense_col0, int32_t Nd, int64_t B,
                   float* out, int64_t ldo, int32_t* status, mml_stream_t stream);
/* mml_gather_fwd that also leaves the magnitude of what it wrote: workgroup w stores the largest |value| of its part
 * of `out` to wg_max[w] (plain stores, no atomics; wg_max_len = mml_gather_wgmax_len(F, E, Nd, B) values, all of them
 * written by every call).  Feeding wg_max -- viewed as one row of floats -- to mml_amax_batch gives the magnitude slot of
 * `out` from a few KB instead of a pass over the whole output.  Needs E % 4 == 0, ldo % 4 == 0 and 16-byte aligned
 * tables and output (MML_ERR_ARG otherwise). */
int64_t mml_gather_wgmax_len(int32_t F, int32_t E, int32_t Nd, int64_t B);
/* Kernel symbol of the calling thread's most recent gather launch ("gather_vec4_kernel", "gather_lds_kernel" -- the
 * LDS-staged variant for tables of at most 128 rows, north_star's "LDS-staged index dedup", which runs when the
 * environment holds MMLREC_GATHER_LDS = n > 0 workgroups per CU at the time of the call and the launch neither marks rows
 * nor leaves workgroup maxima -- or "gather_scalar_kernel"); "" before the first one.  For tests / benchmark labels. */
const char* mml_gather_last_kernel(void);
int mml_gather_fwd_wgmax(const float* const* tables, const int64_t* vocab, const int32_t* col, int32_t F, int32_t E,
                         const float* X, int64_t ldX, int32_t dense_col0, int32_t Nd, int64_t B, float* out,
                         int64_t ldo, float* wg_max, int64_t wg_max_len, int32_t* status, mml_stream_t stream);
/* mml_gather_fwd that also marks every row it reads in row_marks (a byte per table row, layout and contract of
 * mml_scatter_bwd's row_marks): the split dense table update (mml_opt_tensor.skip_rows) gets the batch's row set from
 * the gather itself; mml_rows_compact then turns the marks into the `seen` bitmaps and the touched-row list
 * (touched[] = every marked or already-seen row, *touched_count = their number; the marks are cleared). */
int mml_gather_fwd_mark(const float* const* tables, const int64_t* vocab, const int32_t* col, int32_t F, int32_t E,
                        const float* X, int64_t ldX, int32_t dense_col0, int32_t Nd, int64_t B, float* out,
                        int64_t ldo, uint8_t* row_marks, int32_t* status, mml_stream_t stream);
int mml_rows_compact(uint32_t* const* seen, const int64_t* vocab, const int64_t* rowbase, int32_t F, int32_t* touched,
                     int32_t* touched_count, int32_t touched_cap, uint8_t* row_marks, mml_stream_t stream);
/* same with native int32 indices idx[b*ldi + f] (additive API for vocabularies >= 2^24);
 * dense values come from dense[b*ldd + j] (may be null when Nd == 0). */
int mml_gather_fwd_idx32(const float* const* tables, const int64_t* vocab, int32_t F, int32_t E,
                         const int32_t* idx, int64_t ldi, const float* dense, int64_t ldd, int32_t Nd, int64_t B,
                         float* out, int64_t ldo, int32_t* status, mml_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * K2  sparse row-scatter backward of K1.
 * Replaces aten::embedding_dense_backward triggered by sparse=False (model/basemodel.py:122,
 * model/utils.py:476):  grad_tables[f][idx[b,f], :] += dOut[b, f*E:(f+1)*E]  (duplicates accumulate).
 * grad_tables are dense [V_f, E] accumulators owned by the caller (zeroed by the caller, or kept zero
 * between steps by mml_opt_step_rows).  Duplicate rows inside a workgroup are pre-reduced in LDS before
 * one float atomic per (row, e) reaches HBM, so the result is order-dependent in the last bits.
 * If `touched` is non-null the first writer of a row (per-table bitmap `seen[f]`, V_f bits rounded up to
 * 32) appends the global row id rowbase[f] + idx to touched[0..cap) and bumps *touched_count: the compact row
 * list the sparse-row optimizer walks.  rowbase is a HOST array of F+1 int64 (exclusive prefix sum of vocab).
 * ---------------------------------------------------------------------------------------------- */
int mml_scatter_bwd(float* const* grad_tables, const int64_t* vocab, const int32_t* col, int32_t F, int32_t E,
                    const float* X, int64_t ldX, int64_t B, const float* dOut, int64_t ldo,
                    uint32_t* const* seen, const int64_t* rowbase, int32_t* touched, int32_t* touched_count,
                    int32_t touched_cap, uint8_t* row_marks, int32_t* status, mml_stream_t stream);
/* row_marks (optional, may be NULL): a DEVICE scratch map of 32 * sum_f ceil(V_f / 32) bytes, all-zero between calls
 * (field f owns the bytes from 32 * sum_{g<f} ceil(V_g / 32)).  With it the rows are marked by plain byte stores and
 * the list / bitmaps are rebuilt from the marks by a compaction pass (touched[] = every marked or already-seen row,
 * *touched_count = their number) instead of one same-address atomic per hot row: 350 us -> 30 us for the index-only
 * pass at B = 65 536 on the AliExpress-shaped tables.
 * row_marks WITHOUT a touched list (touched == NULL): the scatter only marks every row it adds to and leaves the bytes
 * set -- the consumer is mml_opt_step_dense with mml_opt_tensor.grad_marks, which skips the gradient read of unmarked
 * rows and clears the marks.  Needs the LDS-fold kernel (E in {4, 8, 16}, 16-byte aligned dOut, ldo % 4 == 0); other
 * shapes are rejected (MML_ERR_ARG) rather than served by a kernel that would leave the marks unset. */

/* Unique (field, row) list of a batch WITHOUT gradients: the scatter's LDS dedup run on the indices alone.  Appends
 * rowbase[f] + row for every distinct row of X[:, col[f]] to touched[] (first-seen order, `seen` bitmaps as above).
 * Used by the lazy-exact table optimizer, which must bring exactly these rows up to date BEFORE the gather reads
 * them. */
int mml_index_unique(const int64_t* vocab, const int32_t* col, int32_t F, int32_t E, const float* X, int64_t ldX,
                     int64_t B, uint32_t* const* seen, const int64_t* rowbase, int32_t* touched,
                     int32_t* touched_count, int32_t touched_cap, uint8_t* row_marks, int32_t* status,
                     mml_stream_t stream);

/* Deterministic form of mml_scatter_bwd ("mode: sorted" of SURVEY 8(b), without the sort): the same LDS fold, but
 * every addend is turned into 64-bit fixed point with ONE unit for the whole launch (taken from max |dOut|, which the call
 * measures into amax_slot: MML_AMAX_WORDS words) and the chunk sums are added to 64-bit integer row accumulators
 * acc64[f] ([V_f, E] int64, all zero between calls) with integer atomics; a second launch turns the totals of the
 * marked rows into fp32, adds them to grad_tables and zeroes them again.  Integer sums do not depend on the order of
 * the addends, so the result is BITWISE repeatable and independent of the order of the samples in the batch (a sorted
 * fp32 sum would still depend on it); accuracy: addends down to 2^-(62 - 24 - ceil(log2 B)) of the largest |dOut| are
 * summed exactly.  row_marks: the byte map of mml_scatter_bwd (required: the second launch walks it); clear_marks = 0
 * leaves the marks for mml_opt_tensor.grad_marks / mml_rows_compact.  E in {4, 8, 16}, dOut 16-byte aligned. */
int mml_scatter_bwd_det(float* const* grad_tables, const int64_t* vocab, const int32_t* col, int32_t F, int32_t E,
                        const float* X, int64_t ldX, int64_t B, const float* dOut, int64_t ldo, int64_t* const* acc64,
                        uint32_t* amax_slot, uint8_t* row_marks, int32_t clear_marks, int32_t* status,
                        mml_stream_t stream);
/* Native-index variants (int32 idx[b*ldi + f], fields in array order): vocabularies >= 2^24 and the
 * owner side of row-sharded tables, which sees lookups as keys into its flat row space (F = 1). */
int mml_scatter_bwd_idx32(float* const* grad_tables, const int64_t* vocab, int32_t F, int32_t E, const int32_t* idx,
                          int64_t ldi, int64_t B, const float* dOut, int64_t ldo, uint32_t* const* seen,
                          const int64_t* rowbase, int32_t* touched, int32_t* touched_count, int32_t touched_cap,
                          uint8_t* row_marks, int32_t* status, mml_stream_t stream);
int mml_index_unique_idx32(const int64_t* vocab, int32_t F, int32_t E, const int32_t* idx, int64_t ldi, int64_t B,
                           uint32_t* const* seen, const int64_t* rowbase, int32_t* touched, int32_t* touched_count,
                           int32_t touched_cap, uint8_t* row_marks, int32_t* status, mml_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Row-wise sharded tables (replaces the reference's dead --is_parallel stub, main.py:81-83,
 * model/basemodel.py:235-238).  Row r of field f lives on rank (r + f) mod world at local row r / world; a rank keeps
 * its F shards back to back in one flat [R, E] buffer and a lookup travels as the int32 key keybase[f] + r / world.
 *   mml_route_count  counters[0..world) = lookups of the batch bound for each owner (counters[world..2*world) is
 *                    cleared for mml_route_place); indices are validated / clamped like mml_gather_fwd (status bits).
 *   mml_route_place  send_keys[B*F] = keys grouped by owner (owner segments in rank order: the send layout of ONE
 *                    all-to-all over all fields), pos[b*F + f] = position of (b, f)'s key in send_keys.  Needs the
 *                    counters of mml_route_count on the same batch.  Order inside an owner segment is unspecified.
 *   mml_rows_permute dst[pos[b*F+f], :] = src[b, f*E:(f+1)*E]: packs d(dnn_input) into the send order of the gradient
 *                    exchange (the inverse direction is mml_gather_fwd_idx32 on the received row block).
 *   mml_shard_rows   shard[l, :] = table[l*world + first, :] (to_table = 0; rows past the table end are zero) or the
 *                    inverse copy (to_table = 1); first = (rank - f) mod world.
 * X / idx: exactly one is non-null (fp32-encoded indices with column map col[], or native int32 [B, ldi]).
 * col, vocab, keybase are HOST arrays of F entries; counters, send_keys, pos, status are DEVICE buffers.
 * ---------------------------------------------------------------------------------------------- */
int mml_route_count(const float* X, int64_t ldX, const int32_t* idx, int64_t ldi, const int32_t* col,
                    const int64_t* vocab, int32_t F, int64_t B, int32_t world, int32_t* counters, int32_t* status,
                    mml_stream_t stream);
int mml_route_place(const float* X, int64_t ldX, const int32_t* idx, int64_t ldi, const int32_t* col,
                    const int64_t* vocab, const int64_t* keybase, int32_t F, int64_t B, int32_t world,
                    int32_t* counters, int32_t* send_keys, int32_t* pos, int32_t* status, mml_stream_t stream);
int mml_rows_permute(const float* src, int64_t lds, const int32_t* pos, int32_t F, int32_t E, int64_t B, float* dst,
                     mml_stream_t stream);
int mml_shard_rows(float* table, int64_t V, float* shard, int64_t rows_local, int32_t E, int32_t world, int32_t first,
                   int32_t to_table, mml_stream_t stream);
/* Requester-side de-duplication: route the batch's DISTINCT rows (the touched list of mml_index_unique: global row ids
 * rowbase[f] + r, *count of them on the device, capacity cap) instead of its B*F lookups -- under Zipf a 65 536-sample
 * AliExpress-shaped batch holds 209 k distinct rows against 1.97 M lookups.
 *   mml_route_list_count / _place   as mml_route_count / _place, over the list; _place also writes
 *                                   slot_of[global row id] = position of the row's key in send_keys
 *   mml_lookup_slots                pos[b*F + f] = slot_of[rowbase[f] + trunc(X[b, col[f]])]  (indices validated / clamped)
 *   mml_rows_clear                  clears the `seen` words of the listed rows (requester-side bitmaps)
 * slot_of is a DEVICE int32 array of rowbase[F] entries (only the listed rows are written / read). */
int mml_route_list_count(const int32_t* list, const int32_t* count, int32_t cap, const int64_t* vocab,
                         const int64_t* rowbase, int32_t F, int32_t world, int32_t* counters, mml_stream_t stream);
int mml_route_list_place(const int32_t* list, const int32_t* count, int32_t cap, const int64_t* vocab,
                         const int64_t* rowbase, const int64_t* keybase, int32_t F, int32_t world, int32_t* counters,
                         int32_t* send_keys, int32_t* slot_of, mml_stream_t stream);
int mml_lookup_slots(const float* X, int64_t ldX, const int32_t* col, const int64_t* vocab, const int64_t* rowbase,
                     int32_t F, int64_t B, const int32_t* slot_of, int32_t* pos, int32_t* status, mml_stream_t stream);
int mml_rows_clear(const int32_t* list, const int32_t* count, int32_t cap, const int64_t* rowbase,
                   uint32_t* const* seen, int32_t F, mml_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * K3  grouped GEMM family on the fp32 MFMA pipe (v_mfma_f32_32x32x2_f32: exact fp32 products and sums).
 * Replaces nn.Linear + activation inside DNN.forward (model/utils.py:146-161: addmm, relu_) and the
 * autograd mm/mm backward pair of each layer.
 * ---------------------------------------------------------------------------------------------- */
/* Arithmetic of the GEMM family (process-wide):
 *   0 = fp32 MFMA on every launch (v_mfma_f32_32x32x2_f32: bitwise an fmaf chain);
 *   4 = auto (DEFAULT): fp32-equivalent.  Where it is faster, every fp32 operand value is cut in registers into three
 *       bf16 planes h + m + l == x (exact mantissa slices) and a 16-k product block costs six
 *       v_mfma_f32_32x32x16_bf16 (hh, hm, mh, mm, hl, lh; the dropped terms are <= 2^-24 |a b|) with fp32
 *       accumulation.  Max-norm error against float64 equals the fp32 MFMA's (4.7e-7 vs 4.3e-7 on a
 *       8192 x 256 x 240 product, tools/bench_gemm.py); the other launches use the fp32 MFMA;
 *       When every operand of a launch comes with its magnitude (the amax fields of the descriptors below), auto runs
 *       the TWO-PLANE fp16 form instead: each operand is scaled by a power of two that puts its largest magnitude below
 *       2^15, cut into h = rne16(x s), l = rne16(x s - h) (x s = h + l to 22-23 significant bits), a 16-k block costs
 *       three v_mfma_f32_32x32x16_f16 (hh, hl, lh) and the fp32 accumulators are scaled back exactly in the epilogue.
 *       Max-norm error against float64 3.4e-7 (three bf16 planes 4.7e-7, fp32 MFMA 4.3e-7), 25-30 % faster;
 *   2 = same as auto;
 *   3 = the three-plane bf16 form on every launch of the LDS-DMA kernel (magnitudes ignored);
 *   1 = REDUCED precision, opt-in: operands rounded to bf16 in registers, one v_mfma_f32_32x32x16_bf16 per 16-k block,
 *       fp32 accumulation (~3e-3 relative per product; not covered by the 1e-4 parity contract).
 * Environment MMLREC_GEMM_MODE overrides the default. */
int mml_gemm_set_mode(int32_t mode);
int mml_gemm_get_mode(void);
/* Activation-stationary forward (csrc/gemm_panel.hip).  A forward launch whose problems all read ONE input A [M, K] (the
 * first DNN layer of every expert / gate tower: reference model/mmoe.py:69-79) with K in {160, 208, 240}, M >= 8 192 and
 * M % 128 == 0, N % 64 == 0 (an even number of 64-column half tiles over the launch), nn.Linear weights with pre-cut
 * planes, the magnitude of A, activation relu, 16-byte aligned operands, and relu sign masks on all problems or on
 * none, is served by a persistent workgroup per 128-row block that keeps the block's fp16 planes as MFMA fragments in
 * registers and sweeps every N-tile past them (bitwise the results of the tile kernel).  on = 0 switches it off
 * (process-wide; default on; environment MMLREC_GEMM_PANEL=0 does the same). */
int mml_gemm_set_panel(int32_t on);
/* Weight-stationary streaming GEMM (csrc/gemm_ws.hip).  A forward launch whose problems ALL have K % 64 == 0 or K % 80 == 0, N in {64,
 * 128, 256}, N K <= 32 768 (the two fp16 planes of the weight fit 128 KiB of LDS), one M >= 8 192, pre-cut planes (either
 * weight layout), the magnitude of A and activation relu, none, sigmoid or 2 sigmoid -- and an input-gradient launch of
 * single-source problems with K in {64, 128, 256} output columns, N % 64 == 0 or N % 80 == 0, N K <= 32 768, activation none or relu by sign
 * mask -- is served by persistent workgroups that keep one problem's weight planes in LDS while their waves stream 32-row
 * blocks of the batch past them (second expert layers, towers, PepNet's and STAR's layers: reference model/mmoe.py:69-119,
 * model/pepnet.py:64-78, model/utils.py:171-218); problems of different widths in one call become one kernel launch per
 * width.  Bitwise the results of the tile kernel.  on = 0 switches it off (default on; environment MMLREC_GEMM_WS=0 does
 * the same). */
int mml_gemm_set_ws(int32_t on);
/* Weight gradients cut once per workgroup (csrc/gemm_nt.hip).  A weight-gradient launch of <= 16 (MML_MAX_GROUP) problems in nn.Linear
 * layout that all carry both operand magnitudes, with M % 32 == 0, M >= 16 384, N % 32 == 0, K % 4 == 0 and 16-byte
 * aligned operand rows (the DNN layers of a large batch: reference model/utils.py:146-161, autograd's mm backward) is
 * served by workgroups that own a 128 x 128 tile of dW for a slab of the batch: the rows of both operands are cut into
 * their fp16 planes ONCE, stored to LDS as they lie in memory, and the MFMA fragments -- consecutive batch rows of a column
 * -- are read with the transposing LDS read ds_read_b64_tr_b16; partial tiles go to the same workspace, summed in slab order.
 * Same products and fp32 accumulation as the tile kernel, another summation order (not bitwise equal to it; bitwise
 * repeatable).  on = 0 never, 1 (default) every qualifying launch, 2 launches of at least 16 output tiles, 3 a measured
 * lab variant (the dC fragments loaded straight from global memory, half the LDS traffic: 19-21 % slower, kept for the
 * record); environment MMLREC_GEMM_NT sets the same (csrc/gemm_nt.hip holds the numbers: AE-30 step 1.68 -> 1.60 ms on
 * one box). */
int mml_gemm_set_nt(int32_t on);
/* Kernel symbol (as rocprofv3 prints it, without the mml:: prefix) of the calling thread's most recent GEMM launch;
 * "" before the first one.  For profilers / benchmark labels. */
const char* mml_gemm_last_kernel(void);
/* Unused dynamic LDS requested by the weight-gradient launches (process-wide, default 0).  A trainer that runs the
 * wgrad GEMMs on a side stream next to an HBM-bound kernel (the dense table optimizer) sets ~17 KiB so that only three
 * wgrad workgroups fit a CU and the other kernel's waves can co-reside. */
int mml_gemm_set_wgrad_lds_pad(int32_t bytes);

/* Operand magnitudes.  A magnitude slot is MML_AMAX_WORDS consecutive uint32 words on the device; its value is the
 * LARGEST of them, read as the bit pattern of a non-negative float: an upper bound of max |x| over a tensor (0 = the
 * tensor is all zero).  Producers raise a slot with atomic max on any of its words (mml_amax_batch; the GEMM launches
 * through amax_out: the magnitude of what they stored), nobody lowers it: mml_amax_reset zeroes slots at the start of a
 * step.  A GEMM launch whose descriptors all carry the magnitudes of BOTH operands may run the two-plane fp16
 * arithmetic (mml_gemm_set_mode); a stale-high bound costs precision only at 2^-39 of the bound, a too-LOW bound
 * overflows fp16 -- so a slot must cover everything the tensor holds when the GEMM runs. */
typedef struct {
  const float* x;   /* [rows, cols], row pitch ld */
  int64_t rows, ld;
  int32_t cols, pad_;
  uint32_t* slot;   /* MML_AMAX_WORDS words */
} mml_amax_desc;
int mml_amax_batch(const mml_amax_desc* descs, int32_t n, mml_stream_t stream);
int mml_amax_reset(uint32_t* slots, int64_t n_slots, mml_stream_t stream);

typedef struct {
  const float* A;    /* [M, K] input activations                                  */
  const float* W;    /* [N, K] nn.Linear weight layout (model/utils.py:130)        */
  const float* bias; /* [N] or NULL                                                */
  float* C;          /* [M, N] = act(A W^T + bias)                                 */
  int64_t lda, ldw, ldc;
  int32_t M, N, K;
  int32_t act;       /* MML_ACT_*                                                  */
  int32_t w_kn;      /* 0: W is [N,K] (nn.Linear); 1: W is [K,N] (STAR SharedSpecificLinear layout, model/utils.py:171) */
  int32_t pad_;
  /* Optional (training): when non-NULL and act == MML_ACT_RELU, bit (c & 31) of relu_mask[r * ldmask + (c >> 5)] is
   * set to (C[r][c] > 0) -- 1 bit per output instead of the 4 bytes mml_gemm_grouped_dgrad would otherwise re-read
   * to apply relu'.  ldmask >= ceil(N / 32) words per row. */
  uint32_t* relu_mask;
  int64_t ldmask;
  /* Optional operand magnitudes (see above): of A, of W; amax_out receives the magnitude of C. */
  const uint32_t* amax_a;
  const uint32_t* amax_w;
  uint32_t* amax_out;
  /* Optional pre-cut weight (mml_gemm_planes_cut, layout MML_PLANES_ROWS for w_kn = 0): the two fp16 planes of W in a
   * buffer of W's own shape and pitch, and the exponent they were scaled with.  When every problem of a launch that runs
   * the two-plane arithmetic carries them, the kernel reads the planes instead of cutting W's fragments (same bits). */
  const uint32_t* w_planes;
  const int32_t* w_kexp;
  /* K7, optional (PepNet: x = h (.) GateNN(.), reference model/pepnet.py:31-32, :72-78, :139-140): with mul != NULL the
   * launch ALSO stores prod[r][c] = C[r][c] * mul[r][c] -- the gated layer input leaves the gate network's last Linear
   * (act = MML_ACT_SIGMOID2) without a pass of its own over memory.  mul, prod: [M, N] at pitches ldmul / ldprod,
   * 16-byte aligned, N % 4 == 0, C 16-byte aligned too; amax_prod receives the magnitude of prod.  Only on the LDS-DMA
   * kernel (K % 16 == 0, aligned operands): other shapes return MML_ERR_UNSUPPORTED. */
  const float* mul;
  float* prod;
  int64_t ldmul, ldprod;
  uint32_t* amax_prod;
} mml_gemm_fwd_desc;
int mml_gemm_grouped_fwd(const mml_gemm_fwd_desc* descs, int32_t n, mml_stream_t stream);
/* SURVEY 8(b) name of the K7 forward: mml_gemm_grouped_fwd restricted to descriptors that carry mul / prod. */
int mml_pep_gate_fwd(const mml_gemm_fwd_desc* descs, int32_t n, mml_stream_t stream);


typedef struct {
  float* dA;           /* [M, K] = sum_s dC_s W_s  (then * act'(Y) if act != NONE)            */
  const float* Y;      /* [M, K] forward OUTPUT of the layer that produced A (relu/sigmoid derivative source) or NULL */
  int64_t ldda, ldy;
  int32_t M, K;
  int32_t act;         /* activation whose derivative is applied in the epilogue             */
  int32_t n_src;
  int32_t accumulate;  /* 1: dA += result (after the derivative), 0: overwrite                */
  int32_t pad_;
  const float* dC[MML_MAX_SRC]; /* [M, N_s] gradients w.r.t. the pre-activation outputs      */
  const float* W[MML_MAX_SRC];  /* [N_s, K] (w_kn=0) or [K, N_s] (w_kn=1)                     */
  int64_t lddc[MML_MAX_SRC], ldw[MML_MAX_SRC];
  int32_t N[MML_MAX_SRC];
  int32_t w_kn[MML_MAX_SRC];
  /* Optional: the sign mask mml_gemm_grouped_fwd wrote for Y (act must be MML_ACT_RELU); used instead of Y. */
  const uint32_t* relu_mask;
  int64_t ldmask;
  /* Optional operand magnitudes: of dC[s] and W[s] for every source; amax_out receives the magnitude of dA as stored
   * (after the derivative and the accumulation). */
  const uint32_t* amax_dc[MML_MAX_SRC];
  const uint32_t* amax_w[MML_MAX_SRC];
  uint32_t* amax_out;
  /* Optional pre-cut weights of the sources (mml_gemm_planes_cut, layout MML_PLANES_COLS for w_kn = 0), all sources of
   * a problem cut as ONE group (one common exponent): honoured when every source of every problem of the launch has them. */
  const uint32_t* w_planes[MML_MAX_SRC];
  const int32_t* w_kexp[MML_MAX_SRC];
  /* K7 backward, optional ("gate mode", gate_h != NULL): the problem's input is a product x = h (.) g; the input gradient
   * v = sum_s dC_s W_s is NOT stored, the epilogue forms the gradients of the two factors from it:
   *     d_h (+)= v * g * act_h'(h)          d_g (+)= v * h * act_g'(g)
   * act_h / act_g: the activations that PRODUCED h / g (derivative from the outputs h, g; MML_ACT_NONE: factor 1), acc_h /
   * acc_g: 1 = add to what d_h / d_g hold.  dA, Y, act, accumulate, relu_mask and amax_out are ignored.  gate_h, gate_g,
   * d_h, d_g: [M, K] at pitches ld_h, ld_g, ld_dh, ld_dg, 16-byte aligned, K % 4 == 0; amax_dh / amax_dg receive the
   * magnitudes of what was stored.  Only on the LDS-DMA kernel, like mul / prod of the forward. */
  const float* gate_h;
  const float* gate_g;
  float* d_h;
  float* d_g;
  int64_t ld_h, ld_g, ld_dh, ld_dg;
  int32_t act_h, act_g, acc_h, acc_g;
  uint32_t* amax_dh;
  uint32_t* amax_dg;
} mml_gemm_dgrad_desc;
int mml_gemm_grouped_dgrad(const mml_gemm_dgrad_desc* descs, int32_t n, mml_stream_t stream);
/* SURVEY 8(b) name of the K7 backward: mml_gemm_grouped_dgrad restricted to gate-mode descriptors. */
int mml_pep_gate_bwd(const mml_gemm_dgrad_desc* descs, int32_t n, mml_stream_t stream);

/* Pre-cut weights for the two-plane fp16 arithmetic.  The GEMM kernels cut every fp32 operand fragment into its planes
 * h = rne16(x 2^k), l = rne16(x 2^k - h) in registers, once per wave that reads it; a weight matrix is read by every row
 * tile of the batch, so it can be cut ONCE per step instead: `planes` has W's shape and pitch and holds, per aligned
 * block of 16 values along the reduction, the 32 bytes of h and the 32 bytes of l in the order the kernel's fragment
 * reads deliver them (so the planes travel through the same LDS image as the floats would).
 *   MML_PLANES_ROWS: the reduction runs along a ROW of W ([N, K] read by the forward: K % 16 == 0, or any K with ldp;
 *                    a [K, N] matrix (w_kn = 1) read by the input gradient);
 *   MML_PLANES_COLS: the reduction runs down the ROWS of W ([N, K] read by the input gradient: N % 16 == 0; a [K, N]
 *                    matrix read by the forward).
 * k = the largest exponent that keeps every |w| 2^k below 2^15 for the LARGEST of the n_amax magnitude slots given
 * (the weights that feed one input-gradient problem share their exponent); it is written to *kexp.  Bit-identical to
 * the in-kernel cut with the same magnitudes. */
#define MML_PLANES_ROWS 0
#define MML_PLANES_COLS 1
#define MML_MAX_PLANES 64
typedef struct {
  const float* W;     /* [rows, cols], row pitch ld (floats) */
  uint32_t* planes;   /* [rows, cols] 4-byte words at row pitch ldp */
  int64_t rows, ld;
  int32_t cols, layout;
  int32_t n_amax, pad_;
  const uint32_t* amax[MML_MAX_SRC];
  int32_t* kexp;
  /* Row pitch of `planes` in words; 0 = ld.  A wider pitch serves the zero-padded operand a GEMM reads when the
   * reduction extent is not a multiple of 16 (K0 = 303 -> 304): with MML_PLANES_ROWS, cols may then be ANY length and
   * the last, partial block is cut as if the missing columns were zero (ldp >= cols rounded up to 16; the words of the
   * missing columns are written as zero planes); with MML_PLANES_COLS the columns beyond `cols` are not touched (the
   * caller zeroes the buffer once). */
  int64_t ldp;
  /* K6, optional (STAR, reference model/utils.py:214-218: the layer's weight is W_specific (.) W_shared): with W2 != NULL
   * the planes are cut from the element-wise PRODUCT W[r][c] * W2[r][c] (W2: same shape, pitch ld2) -- the derived weight
   * reaches the GEMMs pre-cut without ever being read back.  The magnitude bound of the product is the product of the
   * factors' bounds: amax[0 .. n_amax) belong to the W factors of the group, amax[n_amax .. 2 n_amax) to their W2
   * factors (2 n_amax <= MML_MAX_SRC). */
  const float* W2;
  int64_t ld2;
} mml_planes_desc;
int mml_gemm_planes_cut(const mml_planes_desc* descs, int32_t n, mml_stream_t stream);
/* SURVEY 8(b) names of K6: the grouped forward / input-gradient GEMMs restricted to layers in STAR's [K, N] weight layout
 * (w_kn = 1) that carry pre-cut planes of the derived weight (mml_gemm_planes_cut with W2; layout MML_PLANES_COLS for the
 * forward -- the reduction runs down the rows of a [K, N] matrix --, MML_PLANES_ROWS for the input gradient). */
int mml_star_linear_fwd(const mml_gemm_fwd_desc* descs, int32_t n, mml_stream_t stream);
int mml_star_linear_bwd(const mml_gemm_dgrad_desc* descs, int32_t n, mml_stream_t stream);

typedef struct {
  const float* dC;  /* [M, N] gradient w.r.t. pre-activation output */
  const float* A;   /* [M, K] layer input                            */
  float* dW;        /* [N, K] (w_kn=0) or [K, N] (w_kn=1) = dC^T A   */
  float* dbias;     /* [N] = column sums of dC, or NULL              */
  int64_t lddc, lda, lddw;
  int32_t M, N, K;
  int32_t accumulate; /* 1: dW/dbias += ; 0: overwrite               */
  int32_t w_kn;
  int32_t pad_;
  /* Optional operand magnitudes of dC and A. */
  const uint32_t* amax_dc;
  const uint32_t* amax_a;
} mml_gemm_wgrad_desc;
/* The reduction over the batch is split across workgroups; partial tiles go to `workspace` and are
 * summed in a fixed order by a second kernel (bitwise reproducible). */
int64_t mml_gemm_grouped_wgrad_workspace_bytes(const mml_gemm_wgrad_desc* descs, int32_t n);
int mml_gemm_grouped_wgrad(const mml_gemm_wgrad_desc* descs, int32_t n, void* workspace, int64_t workspace_bytes,
                           mml_stream_t stream);
/* The same in two launches: phase 1 writes the per-chunk partial products to the workspace, phase 2 reduces them into
 * dW / dbias (phase 0 = both, as mml_gemm_grouped_wgrad).  A trainer that runs several weight-gradient GEMMs next to
 * an HBM-bound kernel issues all partials first and the (small, bandwidth-hungry) reductions last; each call then
 * needs its own workspace. */
int mml_gemm_grouped_wgrad_phase(const mml_gemm_wgrad_desc* descs, int32_t n, void* workspace,
                                 int64_t workspace_bytes, int32_t phase, mml_stream_t stream);
/* 1 when ONE weight-gradient problem satisfies every per-problem condition of the cut-once weight-gradient kernel
 * (mml_gemm_set_nt, csrc/gemm_nt.hip: layout, both magnitudes, M, N, K, alignment and row pitches), else 0.  A launch is
 * served when all of its problems are, they share M and there are at most 48 of them: a caller that groups problems into
 * launches (engine.Plan.merge_wgrad) asks here, so that its grouping and the library's decision cannot drift apart.  No GPU
 * work, no error text. */
int mml_gemm_nt_serves(const mml_gemm_wgrad_desc* desc);

/* ------------------------------------------------------------------------------------------------
 * K4  gate: skinny linear [Gd -> ne] (no bias) + softmax over experts + expert mix.
 * Replaces gate_dnn_final_layer + softmax + matmul([B,1,Ne],[B,Ne,H]) (model/mmoe.py:80-88,
 * model/ple.py:127-152).  A group = a list of expert outputs (each [B,H]) and up to MML_MAX_GATES gates, each
 * mixing a subset of them (MMoE: every gate sees every expert; PLE CGC: task gate t sees its S specific
 * experts + the shared ones, the shared gate sees all).
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
  const float* G;   /* [B, Gd] gate-DNN output (or dnn_input when gate_dnn_hidden_units == []) */
  const float* Wg;  /* [ne, Gd] nn.Linear(bias=False) weight                                   */
  float* P;         /* [B, ne] softmax(G Wg^T)                                                 */
  float* mix;       /* [B, H]  sum_e P[:,e] * E_{expert[e]}                                    */
  /* backward only: */
  const float* dmix; /* [B, H] gradient of mix                                                 */
  float* dG;         /* [B, Gd] = (dlogits Wg) (* relu'(G) if g_relu)                          */
  float* dWg;        /* [ne, Gd] = dlogits^T G                                                 */
  int64_t ldg, ldp, ldmix, lddmix, lddg;
  int32_t Gd, ne;
  int32_t g_relu;    /* 1: G is a ReLU output, fold its derivative into dG                     */
  int32_t active;    /* backward: 0 = this gate's mix was never consumed (PLE last-level shared gate, model/ple.py:146-152) */
  int32_t expert[MML_MAX_EXPERTS]; /* indices into the group's expert list                    */
} mml_gate_desc;
typedef struct {
  const float* E[MML_MAX_EXPERTS];  /* expert outputs, each [B, H]                            */
  float* dE[MML_MAX_EXPERTS];       /* backward: gradient w.r.t. each expert's PRE-activation (relu' folded in when e_relu) */
  int64_t lde[MML_MAX_EXPERTS], ldde[MML_MAX_EXPERTS];
  int32_t n_experts, n_gates, H;
  int32_t e_relu;                   /* experts end in ReLU (always true for DNN, model/utils.py:155-156) */
  int64_t B;
  mml_gate_desc gate[MML_MAX_GATES];
  /* Optional operand-magnitude slots (see "operand magnitudes" at the GEMM family), each shared by all tensors of its
   * kind: raised with max |mix| over every gate's mixture (forward), max |dE| over every expert gradient and max |dG|
   * over every gate-input gradient (backward).  NULL = not wanted. */
  uint32_t* amax_mix;
  uint32_t* amax_dE;
  uint32_t* amax_dG;
  /* bf16-storage path (K3'): bit 0 = every gate's `mix` is a bf16 [B, H] buffer (ldmix in bf16 elements), bit 1 = every
   * dE, bit 2 = every dG -- the tensors that only GEMMs read are written as their operands (round to nearest even);
   * bit 3 = every expert output E ARRIVES as bf16 (lde in bf16 elements; written that way by the layer that produced it:
   * mml_g16_tn_desc.c_bf16) and is widened exactly; everything else the kernels read stays fp32.  Only the fast row
   * kernels honour it: other shapes return MML_ERR_UNSUPPORTED when a bit is set.  With bit 3, 4 experts x <= 2 gates,
   * 128 < H <= 256, H % 8 == 0 and gate inputs of at most 128 columns the kernels run eight row columns per lane on
   * 32-lane groups (16-byte accesses to the bf16 rows, two samples per wave and trip: round 6). */
  int32_t out_bf16;
  int32_t pad_;
} mml_gate_group;
#define MML_GATE_MIX_BF16 1
#define MML_GATE_DE_BF16 2
#define MML_GATE_DG_BF16 4
#define MML_GATE_E_BF16 8
int mml_gate_mix_fwd(const mml_gate_group* grp, mml_stream_t stream);
int64_t mml_gate_mix_bwd_workspace_bytes(const mml_gate_group* grp);
int mml_gate_mix_bwd(const mml_gate_group* grp, void* workspace, int64_t workspace_bytes, mml_stream_t stream);
/* The same in two launches, like mml_gemm_grouped_wgrad_phase: phase 1 runs the row kernel (input gradients stored, the
 * per-workgroup partial sums of dWg left in the workspace), phase 2 reduces them into dWg (phase 0 = both).  Only the
 * optimizer reads dWg, so a trainer issues phase 2 off its backward chain (beside the weight-gradient GEMMs); the
 * workspace must stay untouched between the two. */
int mml_gate_mix_bwd_phase(const mml_gate_group* grp, void* workspace, int64_t workspace_bytes, int32_t phase,
                           mml_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * K5  prediction heads + loss.
 * Replaces tower_dnn_final_layer (Linear(H->1, bias=False)), PredictionLayer (x + bias, sigmoid;
 * model/utils.py:242-248), the optional domain-mask product (model/mmoe.py:101-106) and
 * sum_t F.binary_cross_entropy(reduction='sum') (model/basemodel.py:294-296) with its backward.
 * Forward-only (predict): y == NULL.
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
  const float* Hin;  /* [B, H] tower output (or the stream itself when tower_dnn_hidden_units == []) */
  const float* w;    /* [H] final-layer weight ([1,H] row, or STAR's [H,1] column: same memory)       */
  const float* w2;   /* optional second factor multiplied elementwise into w (STAR: specific * shared) or NULL */
  const float* bias; /* [1] PredictionLayer bias (out.<t>.bias)                                        */
  const float* bias2;/* optional extra bias terms summed in (STAR shared+specific final biases) or NULL; [n_bias2] */
  float* dH;         /* [B, H] backward: dlogit * w (* relu'(Hin) if h_relu)                           */
  float* dw;         /* [H] backward: sum_b dlogit * Hin  (gradient of the EFFECTIVE weight w*w2)      */
  float* dbias;      /* [1] backward: sum_b dlogit                                                     */
  int64_t ldh, lddh;
  int32_t H;
  int32_t h_relu;
  int32_t n_bias2;
  int32_t mask_col;  /* column of `mask` multiplied into the probability, or -1                        */
  /* Gated head (round 6; PepNet, reference model/pepnet.py:72-78: the last PPNet layer reads h (.) 2 sigmoid(gate)):
   * gate != NULL -> the head's input is Hin (.) gate, formed in registers (the product never goes to memory);
   * backward: dH = dlogit w gate (* relu'(Hin) if h_relu), dgate = dlogit w Hin act'(gate) with act = gate_act
   * (MML_ACT_NONE / SIGMOID / SIGMOID2, the derivative taken from the stored gate value), dw = sum_b dlogit Hin gate.
   * Served by the fast row kernel only (H % 4 == 0, H <= 256, 16-byte aligned rows; fp32 dH): anything else is
   * MML_ERR_UNSUPPORTED. */
  const float* gate; /* [B, H] or NULL                                                                 */
  float* dgate;      /* [B, H] backward (required when gate != NULL and the group trains)              */
  int64_t ldgate, lddgate;
  int32_t gate_act;
  int32_t pad_;
} mml_head_desc;
typedef struct {
  int32_t n_heads;
  int32_t dh_bf16;    /* bf16-storage path (K3'): 1 = every head's dH is a bf16 [B, H] buffer (lddh in bf16 elements); fast kernel only */
  int64_t B;
  float* prob;        /* [B, ldprob] probabilities, head t in column t                                 */
  int64_t ldprob;
  const float* y;     /* [B, ldy] labels or NULL (forward only)                                        */
  int64_t ldy;
  const float* mask;  /* [B, ldmask] domain mask or NULL                                               */
  int64_t ldmask;
  float* loss;        /* [1] device scalar: sum of BCE over heads and samples (overwritten), or NULL   */
  const float* dprob; /* [B, lddprob] upstream dL/dprob used INSTEAD of the BCE gradient when y == NULL (autograd path) */
  int64_t lddprob;
  mml_head_desc head[MML_MAX_HEADS];
  uint32_t* amax_dH;  /* optional operand-magnitude slot raised with max |dH| over all heads (training), or NULL */
  uint32_t* amax_dG;  /* the same for max |dgate| over all gated heads, or NULL                                   */
} mml_head_group;
int64_t mml_head_workspace_bytes(const mml_head_group* grp);
int mml_head_fwd(const mml_head_group* grp, mml_stream_t stream);
/* forward + loss + backward of the heads in one pass (training).  With y != NULL the loss is the summed BCE;
 * with y == NULL and dprob != NULL the heads are differentiated against the given upstream gradient. */
int mml_head_bce_fwd_bwd(const mml_head_group* grp, void* workspace, int64_t workspace_bytes, mml_stream_t stream);
/* The same in two launches: phase 1 = the row kernel (probabilities, dH), phase 2 = the reduction of the per-workgroup
 * partial sums into dw / dbias / loss (phase 0 = both); see mml_gate_mix_bwd_phase. */
int mml_head_bce_fwd_bwd_phase(const mml_head_group* grp, void* workspace, int64_t workspace_bytes, int32_t phase,
                               mml_stream_t stream);
/* The reductions (phase 2) of several head / gate groups in ONE launch (round 5): items[i] names a group whose phase 1 has
 * run into `workspace`; the result is what the phase-2 calls give one by one (dw / dbias / loss of a head group, dWg of a
 * gate group) -- in the last bits it may differ where the merged list is long enough for another reduction kernel (the
 * order of the partial sums is fixed either way).  At most 40 result tensors per call. */
#define MML_ROWS_REDUCE_HEAD 0
#define MML_ROWS_REDUCE_GATE 1
#define MML_ROWS_REDUCE_TOWER_HEAD 2   /* group = const mml_tower_head_group* (K5') */
typedef struct {
  int32_t kind; /* MML_ROWS_REDUCE_HEAD: group = const mml_head_group*; MML_ROWS_REDUCE_GATE: const mml_gate_group* */
  int32_t pad_;
  const void* group;
  void* workspace;
  int64_t workspace_bytes;
} mml_rows_reduce_item;
int mml_rows_reduce_batch(const mml_rows_reduce_item* items, int32_t n, mml_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * K5' (round 6)  the top of the network in ONE launch: last tower layer (Linear(K -> N) + ReLU) -> head (Linear(N -> 1,
 * no bias) + PredictionLayer bias + sigmoid) -> optional domain-mask product -> summed BCE -> backward of all of it down to
 * dL/d(tower input).  Replaces, for every task t, tower_dnn[t] (its last layer), tower_dnn_final_layer[t], out[t] and the
 * loss terms of model/mmoe.py:93-108, model/utils.py:146-161, :242-248, model/basemodel.py:294-296 -- i.e. one
 * mml_gemm_grouped_fwd, one mml_head_bce_fwd_bwd and one mml_gemm_grouped_dgrad launch.  The tower outputs stay in
 * registers; written: prob, dH = dL/d(tower pre-activation) (the tower's weight gradient reads it), dA = dL/d(tower input)
 * (overwritten), and the head's dw / dhbias and the loss through per-workgroup partial sums (phase 2: fixed order).
 * Arithmetic: the two-plane fp16 products of the GEMM family (the tower weight's pre-cut planes in BOTH layouts,
 * mml_gemm_planes_cut; the input's magnitude slot; dH is scaled per 32-row block by its own largest magnitude), the head
 * kernel's expressions for probability, clamped-log BCE and its derivative.  Training only (y required).
 * Served shapes: mml_tower_head_serves (all tasks of one (N, K) in {(64, 128), (64, 64)}, 16-byte aligned rows).
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
  const float* A;               /* [M, K] tower input                                                     */
  int64_t lda;
  const uint32_t* amax_a;       /* magnitude slot of A                                                    */
  const uint32_t* w_planes_fwd; /* planes of the tower weight W [N, K], MML_PLANES_ROWS, pitch ldpf words */
  const uint32_t* w_planes_bwd; /* planes of the same W, MML_PLANES_COLS, pitch ldpb words                */
  int64_t ldpf, ldpb;
  const int32_t* kexp_fwd;      /* exponents the two images were cut with                                 */
  const int32_t* kexp_bwd;
  const float* bias1;           /* [N] tower bias or NULL                                                 */
  const float* w;               /* [N] head weight                                                        */
  const float* hbias;           /* [1] PredictionLayer bias                                               */
  const float* hbias2;          /* [n_hbias2] further bias terms summed in, or NULL                       */
  float* dH;                    /* [M, N]                                                                 */
  float* dA;                    /* [M, K]                                                                 */
  int64_t lddh, ldda;
  float* dw;                    /* [N] head weight gradient                                               */
  float* dhbias;                /* [1]                                                                    */
  uint32_t* amax_dH;            /* optional magnitude slots raised with what was stored, or NULL          */
  uint32_t* amax_dA;
  int32_t K, N, n_hbias2;
  int32_t mask_col;             /* column of `mask` multiplied into the probability, or -1                */
  int32_t head;                 /* column of prob / y this task uses                                      */
  int32_t pad_;
} mml_tower_head_desc;
typedef struct {
  int32_t n;                    /* tasks                                                                  */
  int32_t pad_;
  int64_t M;
  float* prob;                  /* [M, ldprob]                                                            */
  int64_t ldprob;
  const float* y;               /* [M, ldy] labels                                                        */
  int64_t ldy;
  const float* mask;            /* [M, ldmask] domain mask or NULL                                        */
  int64_t ldmask;
  float* loss;                  /* [1] sum of BCE over tasks and samples (overwritten), or NULL           */
  mml_tower_head_desc t[MML_MAX_HEADS];
} mml_tower_head_group;
int mml_tower_head_serves(const mml_tower_head_group* grp);            /* 1 / 0, no error text                   */
int64_t mml_tower_head_workspace_bytes(const mml_tower_head_group* grp);
/* phase 1: the launch; 2: the reduction of the partial sums; 0: both */
int mml_tower_head_fwd_bwd(const mml_tower_head_group* grp, void* workspace, int64_t workspace_bytes, int32_t phase,
                           mml_stream_t stream);


/* ------------------------------------------------------------------------------------------------
 * K3'  bf16-STORAGE GEMM family (round 5; csrc/gemm16.hip) -- BASELINE.json configs[1] ("MMoE ... KuaiRec-shaped ...
 * bf16"): the DNN layers of model/utils.py:146-161 with activations and their gradients STORED as bf16 wherever
 * producer and consumers are GEMMs (and, round 6, the expert outputs of an MMoE whose gate kernels read bf16 rows
 * sixteen bytes per lane: mml_gate_group.out_bf16 bit 3 -- one rounding of those outputs more than operand rounding at
 * the GEMMs).  Arithmetic: one v_mfma_f32_32x32x16_bf16 per 16-k block, fp32 accumulation --
 * exactly the products of mml_gemm_set_mode(1) (operands rounded to bf16, there in registers, here when they are
 * stored), with half the activation traffic and no conversion work in the kernels: bf16 tiles go HBM -> LDS by DMA and
 * LDS -> MFMA fragments as they are (ds_read_b128; the batch-reduction of the weight gradient through the transposing
 * LDS read ds_read_b64_tr_b16).  Opt-in like mode 1, outside the 1e-4 fp32 parity contract; bf16 values are
 * uint16_t bit patterns (round to nearest even).
 *   mml_cast16_batch      bf16 copies of fp32 matrices, optionally transposed: the weights of a step, once per step
 *                         ([N, K] for the forward, [K, N] for the input gradient -- both read reduction-contiguous)
 *   mml_gather16_fwd      mml_gather_fwd writing dnn_input as bf16 (the table rows rounded on the way)
 *   mml_g16_tn            C[M, N] = epilogue(sum_s A_s[M, K_s] B_s[N, K_s]^T): forward (one source, + bias, ReLU, ReLU
 *                         sign masks like mml_gemm_fwd_desc.relu_mask) and input gradient (the sources of
 *                         mml_gemm_dgrad_desc with the TRANSPOSED bf16 weights, ReLU derivative from the sign masks);
 *                         C bf16 or fp32.  M % 128 == 0, N % 64 == 0, every K_s % 64 == 0, 16-byte aligned rows.
 *   mml_g16_wgrad         dW[N, K] (+)= dC[M, N]^T A[M, K], dbias[N] (+)= column sums of dC; dC, A bf16, dW / dbias
 *                         fp32; the batch is cut into slabs, partial tiles go to the workspace and are summed in a fixed
 *                         order (bitwise reproducible).  N % 128 == 0, K % 128 == 0, M % 64 == 0.
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
  const float* src;   /* fp32 [rows, cols], row pitch lds (elements) */
  uint16_t* dst;      /* bf16 [rows, cols] (transpose = 0) or [cols, rows] (transpose = 1), row pitch ldd */
  int64_t rows, lds, ldd;
  int32_t cols, transpose;
} mml_cast16_desc;
int mml_cast16_batch(const mml_cast16_desc* descs, int32_t n, mml_stream_t stream);
int mml_gather16_fwd(const float* const* tables, const int64_t* vocab, const int32_t* col, int32_t F, int32_t E,
                     const float* X, int64_t ldX, int32_t dense_col0, int32_t Nd, int64_t B, uint16_t* out, int64_t ldo,
                     int32_t* status, mml_stream_t stream);
typedef struct {
  int32_t M, N;
  int32_t n_src;      /* 1 .. MML_MAX_SRC: the reduction runs over the sources one after the other */
  int32_t act;        /* MML_ACT_NONE / MML_ACT_RELU applied to (sum + bias) -- forward */
  const uint16_t* A[MML_MAX_SRC];   /* bf16 [M, K_s] */
  const uint16_t* B[MML_MAX_SRC];   /* bf16 [N, K_s]: the weight with the reduction along its rows */
  int64_t lda[MML_MAX_SRC], ldb[MML_MAX_SRC];
  int32_t K[MML_MAX_SRC];
  const float* bias;  /* [N] or NULL */
  void* C;            /* [M, N]: bf16 (c_bf16 = 1) or fp32 */
  int64_t ldc;
  int32_t c_bf16;
  int32_t accumulate; /* fp32 C only: C += result */
  uint32_t* mask_out;       /* forward, act = RELU: bit (c & 31) of mask_out[r * ldmask + (c >> 5)] = (C[r][c] > 0), or NULL */
  const uint32_t* mask_in;  /* input gradient: the result is multiplied by that bit (ReLU derivative), or NULL */
  int64_t ldmask;
} mml_g16_tn_desc;
int mml_g16_tn(const mml_g16_tn_desc* descs, int32_t n, mml_stream_t stream);
typedef struct {
  const uint16_t* dC; /* bf16 [M, N] */
  const uint16_t* A;  /* bf16 [M, K] */
  float* dW;          /* fp32 [N, K] */
  float* dbias;       /* fp32 [N] or NULL */
  int64_t lddc, lda, lddw;
  int32_t M, N, K;
  int32_t accumulate;
} mml_g16_wgrad_desc;
int64_t mml_g16_wgrad_workspace_bytes(const mml_g16_wgrad_desc* descs, int32_t n);
/* phase 0 = both launches, 1 = the partial products, 2 = their reduction (like mml_gemm_grouped_wgrad_phase) */
int mml_g16_wgrad(const mml_g16_wgrad_desc* descs, int32_t n, void* workspace, int64_t workspace_bytes, int32_t phase,
                  mml_stream_t stream);
/* Kernel symbol of the calling thread's most recent launch of this family ("" before the first). */
const char* mml_g16_last_kernel(void);

/* ------------------------------------------------------------------------------------------------
 * K6/K7  elementwise helpers for STAR / PepNet.
 *   mul:      out = a * b                                  (STAR W_spec * W_shared, model/utils.py:215;
 *                                                           PepNet hidden * gate, model/pepnet.py:77, :140)
 *   mul_bwd:  da (+)= dout * b ; db (+)= dout * a          (either may be NULL = stop-gradient)
 *   add_n:    out = sum_i in[i]                            (bias sums, gradient fan-in)
 * ---------------------------------------------------------------------------------------------- */
int mml_ew_mul(const float* a, const float* b, float* out, int64_t n, mml_stream_t stream);
int mml_ew_mul_bwd(const float* dout, const float* a, const float* b, float* da, float* db, int32_t acc_a,
                   int32_t acc_b, int64_t n, mml_stream_t stream);
/* mul_bwd with the derivative of the activation that PRODUCED an operand folded in (act_* != MML_ACT_NONE: the operand
 * holds the activation's output and this product is its only consumer; the gradient written is then the one w.r.t. the
 * pre-activation): PepNet's h * 2*sigmoid(gate) products, model/pepnet.py:72-78 -- saves the separate act' pass */
int mml_ew_mul_bwd_act(const float* dout, const float* a, const float* b, float* da, float* db, int32_t acc_a,
                       int32_t acc_b, int64_t n, int32_t act_a, int32_t act_b, mml_stream_t stream);
int mml_ew_add_n(const float* const* in, int32_t n_in, float* out, int64_t n, mml_stream_t stream);
/* n independent items in one launch (16 per kernel-argument block): out[i] (+)= sum_k x_k[i] * (y_k ? y_k[i] : 1).
 * STAR's derived parameters -- W_spec[d] * W_shared and b_spec[d] + b_shared of every head and layer
 * (SharedSpecificLinear, model/utils.py:214-216) -- and their gradients (d W_shared = sum_d dW_eff[d] * W_spec[d], ...)
 * as one call each way instead of one launch per tensor.  `d` is a HOST array. */
#define MML_SUMPROD_TERMS 8
#define MML_SUMPROD_BATCH 16
typedef struct mml_sumprod_desc {
  float* out;
  const float* x[MML_SUMPROD_TERMS];
  const float* y[MML_SUMPROD_TERMS]; /* NULL = 1 */
  int64_t n;
  int32_t n_terms;
  int32_t accumulate;
  /* act != MML_ACT_NONE: the sum is multiplied by act'(.) taken from deriv_of[i], the OUTPUT of that activation (needs
   * accumulate == 0): the gradient w.r.t. the pre-activation of a factor whose only consumer is this product --
   * PepNet's h * 2*sigmoid(gate) products of all tasks of a layer in one launch each way (model/pepnet.py:72-78) */
  const float* deriv_of;
  int32_t act;
  int32_t pad_;
  uint32_t* amax_out; /* optional operand-magnitude slot (see the GEMM family) raised with max |out|, or NULL */
} mml_sumprod_desc;
int mml_sumprod_batch(const mml_sumprod_desc* d, int32_t n, mml_stream_t stream);
/* Dropout after a DNN layer's activation (reference model/utils.py:121 `self.dropout = nn.Dropout(dropout_rate)`,
 * :159 `fc = self.dropout(fc)`; training mode only -- evaluation is the identity and launches nothing):
 *   out[r, c] (+)= x[r, c] * (keep(r, c) ? 1 / (1 - p) : 0),  r < rows, c < cols (row pitches ldx / ldo; out may be x).
 * keep() is a pure function of (seed, step, site, row0 + r, c): word (c mod 4) of Philox4x32-10 with counter
 * (row0 + r, c / 4, step, site) and key (seed low, seed high), kept iff word >= floor(p * 2^32).  row0 = the position
 * of this buffer's first row in the GLOBAL batch (rank * local batch on a data-parallel rank, else 0): N ranks on a
 * split batch draw the mask one rank draws on the whole batch.  No mask is stored: the
 * backward is the SAME call on dL/d(out) (same seed / step / site).  step_dev (device int32, read by the kernel: a
 * replayed HIP graph draws a new mask every step) overrides step when not NULL.  `site` tells the layers of a model
 * apart.  The stream of masks is this library's, not torch's generator: a reference run with the same torch seed drops
 * different elements (same distribution, same arithmetic). */
int mml_dropout(const float* x, int64_t ldx, float* out, int64_t ldo, int64_t rows, int32_t cols, int64_t row0, float p,
                uint64_t seed, uint32_t site, const int32_t* step_dev, int32_t step, int32_t accumulate,
                mml_stream_t stream);
/* strided 2-D copy / accumulate: dst[r, c] (+)= src[r, c], r < rows, c < cols (concat / split of feature blocks,
 * model/pepnet.py:72, :139) */
int mml_copy2d(const float* src, int64_t lds, float* dst, int64_t ldd, int64_t rows, int32_t cols, int32_t accumulate,
               mml_stream_t stream);
/* n independent strided 2-D copies in ONE launch (item i: dst[r, c] (+)= src[r, c], r < rows, c < cols): the engine's
 * zero-padded weight copies for layers whose reduction length is not a multiple of the GEMM k-step (K0 = 303 with the
 * 63 dense AliExpress columns, configs_msl/config_AE.json:18-23), one launch per layer group instead of one per weight.
 * `d` is a HOST array. */
typedef struct mml_copy2d_desc {
  const float* src;
  int64_t lds;
  float* dst;
  int64_t ldd;
  int64_t rows;
  int32_t cols;
  int32_t accumulate;
  /* optional operand-magnitude slot (MML_AMAX_WORDS words, mml_amax_batch's format): raised with max |x| of what this item
   * stores -- the copies that assemble a GEMM operand measure it on the way (PepNet's gate inputs, model/pepnet.py:72,
   * :139: torch.cat of a detached input and the scene embedding) instead of a pass of mml_amax_batch over the result. */
  uint32_t* amax_out;
} mml_copy2d_desc;
int mml_copy2d_batch(const mml_copy2d_desc* d, int32_t n, mml_stream_t stream);
/* Up to MML_MAX_FIELDS strided column-block copies in ONE launch: for segment s, dst[s][r*ldd[s] + c] (+)= src[s][r*lds[s] + c],
 * r < rows, c < width[s].  Packs / unpacks the per-field pieces of index, row and gradient blocks around the
 * all-to-all exchange of table-sharded runs (no reference counterpart: the reference is single-process).
 * All arrays are HOST arrays of n_seg entries holding device pointers / element strides. */
int mml_copy_cols(const float* const* src, const int64_t* lds, float* const* dst, const int64_t* ldd,
                  const int32_t* width, int32_t n_seg, int64_t rows, int32_t accumulate, mml_stream_t stream);
/* BatchNorm1d inside DNN (model/utils.py:132-134, :153-154: fc -> bn -> activation; torch defaults eps 1e-5, momentum
 * 0.1).  fwd, training != 0: batch statistics (mean / rstd are written for the backward), running statistics and
 * num_batches_tracked updated in place; training == 0: running statistics.  y = act(gamma (z - mean) rstd + beta).
 * bwd: dy is the gradient w.r.t. the BN output (activation derivative already applied by the caller);
 * dgamma / dbeta are written (or accumulated), dz = gamma rstd (dy - (sum dy + xhat sum dy xhat) / B).
 * Workspace: mml_bn_workspace_bytes(B, n) for either call. */
int64_t mml_bn_workspace_bytes(int64_t B, int32_t n);
int mml_bn_fwd(const float* z, int64_t ldz, const float* gamma, const float* beta, float* running_mean, float* running_var,
               int64_t* num_batches_tracked, float* mean, float* rstd, float* y, int64_t ldy, int64_t B, int32_t n,
               int32_t act, int32_t training, float eps, float momentum, void* workspace, int64_t workspace_bytes,
               mml_stream_t stream);
int mml_bn_bwd(const float* dy, int64_t lddy, const float* z, int64_t ldz, const float* gamma, const float* mean,
               const float* rstd, float* dz, int64_t lddz, float* dgamma, float* dbeta, int32_t accumulate, int64_t B,
               int32_t n, void* workspace, int64_t workspace_bytes, mml_stream_t stream);
/* DomainBatchNorm (model/utils.py:553-636; STAR applies it after its first star layer when forward() is given a domain
 * mask, model/star.py:50-51).  gamma / beta are unregistered there, frozen at (1, 0).  Training mode normalises with
 * the statistics of the WHOLE batch for every domain (= mml_bn_fwd with gamma 1 / beta 0; the caller issues that) and
 * mml_domain_bn_update moves the per-domain population statistics:
 *     pop_mean[d] = decay pop_mean[d] + (1 - decay) mean(x[argmax(mask) == d]),  pop_var[d] with the unbiased variance
 * for EVERY domain, like the reference (a domain without samples gets NaN, one with a single sample a NaN variance).
 * Eval mode: y[b] = sum_d mask[b, d] (x[b] - pop_mean[d]) / sqrt(pop_var[d] + eps)  (mml_domain_bn_eval).
 * pop_mean / pop_var are [D, n] device arrays. */
int mml_domain_bn_update(const float* x, int64_t ldx, const float* mask, int64_t ldm, int64_t B, int32_t n, int32_t D,
                         float* pop_mean, float* pop_var, float decay, mml_stream_t stream);
int mml_domain_bn_eval(const float* x, int64_t ldx, const float* mask, int64_t ldm, const float* pop_mean,
                       const float* pop_var, float* y, int64_t ldy, int64_t B, int32_t n, int32_t D, float eps,
                       mml_stream_t stream);
/* SNR-trans routing weights (model/snr_trans.py:38-50).  n_blocks = outputs x inputs blocks of `block` floats each:
 * fwd  W[b] = z(u[b], alpha) * M[b]  with the hard-concrete z = clamp(sigmoid(log u - log(1-u) + log(alpha)/beta)
 *      * (eps - gamma) + gamma, 0, 1);  the routing is then one [K,N] GEMM per output on the concatenated inputs;
 * bwd  dz[b] = <dW[b], M[b]>, du[b] (+)= dz[b] dz/du, dalpha (+)= sum_b dz[b] dz/dalpha (fixed order).
 * M is the reference's unregistered (hence frozen) trans_matrix stack.
 * zw = 1: one coefficient per block (u has n_blocks entries); zw = units: one per output column of a block (MSSM,
 * model/mssm.py:26-29: u has n_blocks * zw entries and is itself unregistered there -> du may be null). */
int mml_snr_gate_weights_fwd(const float* u, const float* alpha, const float* M, float* W, int32_t n_blocks,
                             int64_t block, int32_t zw, float beta, float gamma, float eps, mml_stream_t stream);
int mml_snr_gate_weights_bwd(const float* dW, const float* M, const float* u, const float* alpha, float* du,
                             float* dalpha, int32_t acc_u, int32_t acc_alpha, int32_t n_blocks, int64_t block,
                             int32_t zw, float beta, float gamma, float eps, float* workspace /* n_blocks floats */,
                             mml_stream_t stream);
/* Two-token attention of AITM (model/aitm.py:84-93): per sample, tokens t = 0, 1 with V_t, K_t, Q_t in R^H:
 * s_t = <K_t, Q_t> / sqrt_h, a = softmax(s_0, s_1), out = a_0 V_0 + a_1 V_1.  fwd writes out and (if non-null) the
 * weights A [B,2]; bwd reads A and dout and OVERWRITES dV, dK, dQ of both tokens.  All pointers are device pointers,
 * leading dimensions in elements. */
typedef struct mml_attn2_desc {
  const float* V[2];
  const float* K[2];
  const float* Q[2];
  int64_t ldv[2], ldk[2], ldq[2];
  float* out;
  int64_t ldo;
  float* A;
  const float* dout;
  int64_t lddo;
  float* dV[2];
  float* dK[2];
  float* dQ[2];
  int64_t lddv[2], lddk[2], lddq[2];
  int64_t B;
  int32_t H;
  float sqrt_h; /* the divisor: (float)sqrt(H) in the reference */
} mml_attn2_desc;
int mml_attn2_fwd(const mml_attn2_desc* d, mml_stream_t stream);
int mml_attn2_bwd(const mml_attn2_desc* d, mml_stream_t stream);
/* ESMM output stage (model/esmm.py:58-62): p_out[b] = (ctr, ctr * cvr) from the two head probabilities p_raw[b] = (ctr,
 * cvr).  With labels y: loss[0] = summed BCE of both outputs (model/basemodel.py:294-296) and d_raw = dLoss / d(ctr, cvr);
 * without labels but with d_out (= dL / d p_out from autograd): d_raw by the chain rule.  Feed d_raw to
 * mml_head_bce_fwd_bwd as `dprob`.  Any of y, d_out, d_raw, loss may be null. */
int mml_esmm_combine(const float* p_raw, int64_t ldr, const float* y, int64_t ldy, const float* d_out, int64_t lddo,
                     float* p_out, int64_t ldo, float* d_raw, int64_t lddr, float* loss, int64_t B, mml_stream_t stream);
/* APG (model/apg.py:9-118, the use_uv_shared / no-mf_p branch main.py builds).  The per-sample generated [k,k] weight
 * W_b = reshape(Linear_kk(scene_b)) and bias c_b = Linear_bias(scene_b) applied to the low-rank activation o1_b
 * (apg.py:77-80, :100-104) are ONE ordinary GEMM  o2 = z W_cat + bb  on
 *     z_b = [ o1_b (x) s_b  (k*E, index i*E+e) | o1_b (k) | s_b (E) | zeros up to Kf ]
 * against W_cat [Kf, k] ([K,N] layout): rows i*E+e = Wkk[(i*k+j), e], rows k*E+i = bkk[i*k+j], rows k*E+k+e = Wb[j, e].
 *   mml_apg_features_fwd  writes z;  mml_apg_features_bwd  do1 (+)= dz contracted with s (the scene embedding is
 *   detached in the reference: no gradient to s);  mml_apg_weights  dir 0: packs W_cat from (Wkk [k*k,E], bkk [k*k],
 *   Wb [k,E]); dir 1: unpacks dW_cat into their gradients (accumulating where acc_* != 0). */
int mml_apg_features_fwd(const float* o1, int64_t ldo1, const float* s, int64_t lds, float* z, int64_t ldz, int64_t B,
                         int32_t k, int32_t E, int32_t Kf, mml_stream_t stream);
int mml_apg_features_bwd(const float* dz, int64_t lddz, const float* s, int64_t lds, float* do1, int64_t lddo1,
                         int64_t B, int32_t k, int32_t E, int32_t accumulate, mml_stream_t stream);
int mml_apg_weights(float* Wkk, float* bkk, float* Wb, float* Wcat, int64_t ldw, int32_t k, int32_t E, int32_t dir,
                    int32_t acc_kk, int32_t acc_bkk, int32_t acc_wb, mml_stream_t stream);
/* ESCM output stage and loss (model/escm.py:74-112 and the loss branch of BaseModel.fit, model/basemodel.py:284-292):
 * p_out[b] = (ctr, cvr, ctr * cvr) from the two head probabilities p_raw[b] = (ctr, cvr).  With labels y [B,2]:
 *   loss = BCE_sum(ctr, y0) + cf_w * L1 * S + global_w * BCE_sum(ctr * cvr, y1),   L1 = BCE_sum(cvr, y1),
 *   S = sum_b y0_b * clip(1 / max(ctr_b * N, 1e-6), -15, 15),  N = sum_b y0_b      (counterfact_ipw, escm.py:98-112,
 *   the mean over the batch cancels the factor batch_size; the gradient flows through the inverse propensity, as it
 *   does in the reference), and d_raw = dLoss / d(ctr, cvr).  Without labels but with d_out (= dL / d p_out [B,3] from
 *   autograd): d_raw by the chain rule.  Feed d_raw to mml_head_bce_fwd_bwd as `dprob`. */
int mml_escm_combine(const float* p_raw, int64_t ldr, const float* y, int64_t ldy, const float* d_out, int64_t lddo,
                     float* p_out, int64_t ldo, float* d_raw, int64_t lddr, float* loss, int64_t B, float cf_w,
                     float global_w, mml_stream_t stream);
/* ----------------------------------------------------------------------------------------------
 * Per-batch AUC on the device.  Replaces sklearn.metrics.roc_auc_score run on the host for every
 * training step (model/basemodel.py:316-331; the epoch log averages the per-step values, :335-337): for every segment
 * of `seg` consecutive rows (the last one may be shorter) and every column c < cols,
 *     auc[s * cols + c] = roc_auc_score(y[rows, c] > 0.5, pred[rows, c])
 * with sklearn's tie handling (Mann-Whitney with half credit inside a tie group, evaluated in integer arithmetic), NaN
 * when the segment holds a single class (sklearn raises ValueError there).  seg <= 4096 (one workgroup sorts one
 * segment column in LDS); larger segments return MML_ERR_UNSUPPORTED.  `auc` is a DEVICE array of ceil(n/seg)*cols doubles.
 * ---------------------------------------------------------------------------------------------- */
int mml_auc_segments(const float* pred, int64_t ldp, const float* y, int64_t ldy, int64_t n, int32_t cols, int32_t seg,
                     double* auc, mml_stream_t stream);
/* dst = act'(y) * dy for MML_ACT_SIGMOID2 / SIGMOID / RELU given the forward OUTPUT y (GateNN backward) */
int mml_act_bwd(const float* y, const float* dy, float* dst, int64_t n, int32_t act, mml_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * K8  optimizers (torch.optim.{SGD,Adam,Adagrad,RMSprop} defaults; model/basemodel.py:313, :569-584).
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
  float* param; const float* grad; /* [n] */
  float* state1;                   /* Adam m | Adagrad sum | RMSprop square_avg | SGD unused (NULL) */
  float* state2;                   /* Adam v | others NULL                                          */
  int64_t n;
  /* regulariser of this tensor (BaseModel.get_regularization_loss, model/basemodel.py:524-540: total_loss +=
   * sum(l1 |p|) + sum(l2 p^2)): the update sees grad + l1 * sign(p) + 2 * l2 * p.  0 = none. */
  float l1, l2;
  /* Split dense table update (the reference's dense optimizer, model/basemodel.py:313, with its 2.7 GB table stream
   * taken off the critical path): when skip_rows != NULL the tensor is a [rows, row_elems] table and every row whose
   * bit is set in skip_rows (the `seen` bitmap mml_index_unique builds from the batch's indices BEFORE the forward)
   * is left untouched -- those rows get their update, with their gradient, from mml_opt_step_rows after the scatter.
   * All other rows have a zero gradient this step; with zero_grads != 0 the kernel does not even read `grad` for them
   * (the accumulators are all-zero between steps).  Together the two launches are exactly one dense step. */
  const uint32_t* skip_rows;
  int32_t row_elems;
  int32_t zero_grads;
  /* Marked gradients (single-launch dense table update): grad_marks != NULL makes the tensor a [rows, row_elems] table
   * with one byte per row; a row whose byte is 0 has an all-zero gradient (the accumulators are all-zero between steps
   * and mml_scatter_bwd marks every row it adds to, see its row_marks) and `grad` is not read for it -- 4 of the 28
   * bytes per parameter of a dense Adam step, for 99 % of the rows.  The kernel clears the bytes it finds set.
   * Needs row_elems % 4 == 0 with row_elems / 4 a power of two <= 64, 16-byte aligned tensors, no skip_rows. */
  uint8_t* grad_marks;
} mml_opt_tensor;
typedef struct {
  int32_t kind;      /* MML_OPT_* */
  int32_t step;      /* 1-based step number used for Adam bias correction when step_dev == NULL */
  const int32_t* step_dev; /* optional device counter (hipGraph replay): the kernel reads *step_dev instead */
  float lr, beta1, beta2, eps, alpha;
  int32_t zero_grad; /* 1: write zeros over grad after use (keeps dense table-gradient accumulators clean) */
  /* > 0: cap on the workgroups of a dense launch.  A streaming launch normally fills every wave slot of the chip (8
   * workgroups per CU); the early half of the split table update runs BESIDE the forward / backward kernels and must
   * leave them slots (uncapped, a B = 4 096 step's GEMMs waited 0.3 ms for a slot), and it has the whole forward +
   * backward to finish in. */
  int32_t max_blocks;
} mml_opt_hyper;
/* dense update of up to MML_MAX_OPT_TENSORS whole tensors in one launch (tables included: the reference's
 * optimizer touches every row of every table every step) */
int mml_opt_step_dense(const mml_opt_tensor* tensors, int32_t n, const mml_opt_hyper* hyper, mml_stream_t stream);
/* sparse-row update: only rows listed in touched[0 .. *touched_count) (global row ids over the concatenated
 * tables; table f owns [rowbase[f], rowbase[f+1])).  Exactly equal to the dense update for SGD and Adagrad
 * (rows with zero gradient do not move); for Adam/RMSprop it is the "lazy" variant, NOT the reference's result.
 * Re-zeroes the gradient rows and clears their `seen` bits. */
int mml_opt_step_rows(float* const* tables, float* const* grad_tables, float* const* state1, float* const* state2,
                      uint32_t* const* seen, const int64_t* rowbase, int32_t F, int32_t E,
                      const int32_t* touched, const int32_t* touched_count, int32_t touched_cap,
                      int32_t* const* last /* per-table [V] "row is current as of step" words, or NULL */,
                      const mml_opt_hyper* hyper, mml_stream_t stream);
/* Lazy-EXACT dense Adam/RMSprop for the tables.  The reference's optimizer updates every row every step: a row with
 * zero gradient still decays its moments and moves by lr_t * m_t / (sqrt(v_t / bc2_t) + eps).  Those zero-gradient
 * steps are a deterministic function of (p, m, v, step range), so they can be replayed when a row is next READ:
 *   mml_opt_catchup_rows  replays steps last[row]+1 .. target for the listed rows (target = *step_dev - 1 or
 *                         hyper->step - 1: the state the reference has before the current step), with the same
 *                         per-step arithmetic as mml_opt_step_dense and an early exit once the update falls below
 *                         half an ulp of p (from then on only the moments decay, in closed form);
 *   mml_opt_catchup_dense does the same for EVERY row of one table (before evaluation / checkpointing).
 * Together with mml_opt_step_rows(..., last, ...) this reproduces the dense trajectory while touching only the rows
 * of the current batch. */
int mml_opt_catchup_rows(float* const* tables, float* const* state1, float* const* state2, int32_t* const* last,
                         const int64_t* rowbase, int32_t F, int32_t E, const int32_t* touched,
                         const int32_t* touched_count, int32_t touched_cap, const mml_opt_hyper* hyper,
                         mml_stream_t stream);
int mml_opt_catchup_dense(float* table, float* state1, float* state2, int32_t* last, int64_t V, int32_t E,
                          const mml_opt_hyper* hyper, mml_stream_t stream);
/* *counter += delta (single thread): Adam step counter, touched-list reset (delta = -*counter when reset != 0) */
int mml_counter_update(int32_t* counter, int32_t delta, int32_t reset, mml_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* MMLREC_H */
